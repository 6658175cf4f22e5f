"""GPU parity for simple_world_comm (the last stock scenario): simple_tag physics with a
speaking leader, forest visibility masks, food rewards and integer collision info, through the C ABI
against the golden vectors generated from the reference and the CPU oracle.
f64 <= 1e-9 over free-running 25-step episodes (discrete outputs equal); f32 teacher-forced <= 1e-4,
discrete outputs equal except where the reference itself is within 1e-5 of a threshold."""
import os

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.gpu_util import HipSim, dev, golden_soa, lib, set_force_tile, stats_rows

CASES = ['g13_simple_world_comm', 'g13_simple_world_comm_dense']


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _maxerr(got, want):
    return float(np.max(np.abs(got - want))) if got.size else 0.0


@pytest.fixture(params=[0, 1], ids=['auto_kernel', 'tile_kernel'])
def force_tile(request):
    set_force_tile(request.param)
    yield request.param
    set_force_tile(0)


def _random_state(spec, B, seed, spread=0.6):
    """Crowded worlds: plenty of collisions, agents in forests and on food."""
    rng = np.random.RandomState(seed)
    pos = rng.uniform(-spread, spread, (spec.n_entities, 2, B))
    pos[:, :, ::9] *= 2.2                                                  # some envs reach the boundary penalty
    vel = rng.uniform(-0.5, 0.5, (spec.n_agents, 2, B))
    return rng, pos, vel


@pytest.mark.parametrize('name', CASES)
def test_f64_trajectory_matches_reference(golden_dir, name, force_tile):
    g = _load(golden_dir, name)
    spec = O.simple_world_comm_spec()
    S, T = g['act'].shape[:2]
    sim = HipSim(spec, S, torch.float64)
    sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'))
    sim.comm[:, :, :S] = dev(golden_soa(g, 'comm0'), torch.float64)
    sim.observe(rew=False, info=False, comm=sim.comm)
    assert _maxerr(sim.host()[2], g['obs0'].T) == 0.0
    worst = 0.0
    for t in range(T):
        sim.env_step(dev(g['act'][:, t].T, torch.float64), comm=sim.comm)
        pos, vel, obs, rew, info = sim.host()
        errs = (_maxerr(pos, golden_soa(g, 'pos', t)), _maxerr(vel, golden_soa(g, 'vel', t)),
                _maxerr(obs, g['obs'][:, t].T), _maxerr(rew, golden_soa(g, 'rew', t)))
        worst = max(worst, *errs)
        assert max(errs) <= 1e-9, (name, t, errs)
        assert np.array_equal(info, golden_soa(g, 'info', t))
        assert np.array_equal(sim.comm[..., :S].cpu().numpy(), golden_soa(g, 'comm', t))
    print('%s force_tile=%d worst |err| = %.3e' % (name, force_tile, worst))


@pytest.mark.parametrize('name', CASES)
def test_f32_teacher_forced_matches_reference(golden_dir, name, force_tile):
    g = _load(golden_dir, name)
    spec = O.simple_world_comm_spec()
    S, T = g['act'].shape[:2]
    A, E = spec.n_agents, spec.n_entities
    sim = HipSim(spec, S, torch.float32)
    sizes = np.asarray(spec.size)
    kept = 0.0
    for t in range(T):
        p0 = golden_soa(g, 'pos0') if t == 0 else golden_soa(g, 'pos', t - 1)
        v0 = golden_soa(g, 'vel0') if t == 0 else golden_soa(g, 'vel', t - 1)
        sim.set_state(p0, v0)
        sim.env_step(dev(g['act'][:, t].T, torch.float32), comm=sim.comm)
        pos, vel, obs, rew, info = sim.host()
        np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(vel, golden_soa(g, 'vel', t), rtol=1e-4, atol=1e-4)
        # thresholds: agent-entity contact distances (collisions, food, forests), boundary penalty knees
        P = golden_soa(g, 'pos', t)                                       # [E,2,S]
        dist = np.sqrt(((P[:A, None] - P[None]) ** 2).sum(2))              # [A,E,S]
        dmin = np.add.outer(sizes[:A], sizes)[..., None]
        near = (np.abs(dist - dmin) < 1e-5).any((0, 1))
        ax = np.abs(P[:A])
        near |= (np.abs(ax - 0.9) < 1e-5).any((0, 1)) | (np.abs(ax - 1.0) < 1e-5).any((0, 1))
        ok = ~near
        kept += ok.mean() / T
        np.testing.assert_allclose(obs[:, ok], g['obs'][:, t].T[:, ok], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(rew[:, ok], golden_soa(g, 'rew', t)[:, ok], rtol=1e-4, atol=2e-4)
        assert np.array_equal(info[..., ok], golden_soa(g, 'info', t)[..., ok])
    assert kept > 0.95


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('mk', [O.simple_world_comm_spec, lambda: O.simple_world_comm_spec(5, 3, 2, 3, 2)],
                         ids=['stock', '5_3_2_3_2'])
def test_split_entry_points_equal_fused_and_random_batch_matches_oracle(mk, dtype, force_tile):
    """mpe_world_step + mpe_observe (leader's state.c from the comm buffer) == mpe_env_step (from the
    actions), bit for bit, ragged batch with ld > B; f64 also against the batched oracle; statistics."""
    spec = mk()
    A = spec.n_agents
    B, ld, steps = 3000 + 17, 3072, 3
    rng, pos, vel = _random_state(spec, B, 3)
    a, b = HipSim(spec, B, dtype, ld), HipSim(spec, B, dtype, ld)
    a.set_state(pos, vel)
    b.set_state(pos, vel)
    rows, width = stats_rows(a.desc, B, 8 if dtype == torch.float64 else 4), lib.mpe_stats_width(a.desc)
    stats = torch.zeros(rows, width, dtype=torch.float64, device='cuda')
    bo = O.BatchedOracle(spec)
    want = np.zeros(width)
    seen_hidden = seen_food = 0
    for t in range(steps):
        an = rng.rand(sum(spec.act_dims), B)
        act = torch.zeros(sum(spec.act_dims), ld, dtype=dtype, device='cuda')
        act[:, :B] = dev(an, dtype)
        a.env_step(act, comm=a.comm, stats=stats)
        b.world_step(act, comm=b.comm)
        b.observe(comm=b.comm)
        for x, y in zip(a.host(), b.host()):
            assert np.array_equal(x, y, equal_nan=True)
        assert torch.equal(a.comm[..., :B], b.comm[..., :B])
        if dtype == torch.float64:
            pos, vel, obs, rew, info = bo.env_step(pos, vel, an)
            gp, gv, go, gr, gi = a.host()
            assert max(_maxerr(gp, pos), _maxerr(gv, vel), _maxerr(go, np.concatenate(obs, 0)), _maxerr(gr, rew)) <= 1e-9
            assert np.array_equal(gi, info)
            assert _maxerr(a.comm[..., :B].cpu().numpy(), bo.comm_state(an)) == 0.0
            want[:A] += rew.sum(1)
            want[A:2 * A] += info[:, 0].sum(1)
            want[2 * A] += info[:, 0].sum()
            want[2 * A + 1] += B
            seen_hidden += int((obs[1][4 + 2 * spec.n_landmarks:4 + 2 * spec.n_landmarks + 2] == 0).all(0).sum())
            seen_food += int((rew[A - 1] > 1.0).sum())
    for t_ in (a.obs, a.rew, a.info, a.comm):
        assert torch.isnan(t_[..., B:]).all()                              # padding never written
    if dtype == torch.float64:
        assert seen_hidden > 0 and seen_food > 0
        out = torch.empty(width, dtype=torch.float64, device='cuda')
        assert lib.mpe_stats_reduce(stats.data_ptr(), rows, width, out.data_ptr(), a.stream) == 0
        np.testing.assert_allclose(out.cpu().numpy(), want, rtol=1e-9)


def test_register_and_tile_kernels_bit_identical():
    spec = O.simple_world_comm_spec()
    B = 4096 + 11
    rng, pos, vel = _random_state(spec, B, 8)
    acts = rng.rand(4, sum(spec.act_dims), B)
    for dtype in (torch.float64, torch.float32):
        res = []
        for ft in (0, 1):
            set_force_tile(ft)
            sim = HipSim(spec, B, dtype)
            sim.set_state(pos, vel)
            sim.comm[:] = 0.25
            sim.observe(comm=sim.comm)
            out = [t.copy() for t in sim.host()]
            for t in range(4):
                sim.env_step(dev(acts[t], dtype), comm=sim.comm)
            res.append(out + list(sim.host()) + [sim.comm.cpu().numpy()])
        set_force_tile(0)
        for x, y in zip(*res):
            assert np.array_equal(x, y, equal_nan=True)


def test_api_episode_loop_matches_oracle():
    """make_env('simple_world_comm', benchmark=True): MultiDiscrete leader + Discrete(5) spaces
    (environment.py:42-76), reset, 25 steps and the collision info against the oracle (f64)."""
    from marl_amd import make_env, spaces
    spec = O.simple_world_comm_spec()
    B, T = 256, 25
    env = make_env('simple_world_comm', benchmark=True, num_envs=B, dtype=torch.float64, seed=5)
    assert env.action_space == [spaces.MultiDiscrete([[0, 4], [0, 3]])] + [spaces.Discrete(5)] * 5
    assert [s.shape[0] for s in env.observation_space] == spec.obs_dims
    obs_n = env.reset()
    w = env.world
    ids = np.arange(B, dtype=np.uint64)
    assert np.array_equal(w.pos.cpu().numpy(), O.reset_positions(spec, 5, w.episode - 1, ids))
    npy = lambda t: t.detach().cpu().numpy().astype(np.float64)
    bo = O.BatchedOracle(spec)
    pos, vel = npy(w.pos), npy(w.vel)
    obs0, _, _ = bo.observe(pos, vel)
    for i in range(env.n):
        assert np.array_equal(npy(obs_n[i]).T, obs0[i])
    gen = torch.Generator(device='cuda').manual_seed(2)
    for t in range(T):
        action_n = [torch.rand(B, d, generator=gen, device='cuda', dtype=torch.float64) for d in spec.act_dims]
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        act = np.concatenate([npy(a).T for a in action_n], 0)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act)
        np.testing.assert_allclose(npy(w.pos), pos, rtol=0, atol=1e-9)
        for i, a in enumerate(w.agents):
            np.testing.assert_allclose(npy(new_obs_n[i]).T, obs[i], rtol=0, atol=1e-9)
            assert np.array_equal(npy(info_n['n'][i]), info[i, 0])
        assert np.array_equal(npy(w.agents[0].state.c).T, act[5:9])
        np.testing.assert_allclose(np.stack([npy(x) for x in rew_n]), rew, rtol=0, atol=1e-9)
