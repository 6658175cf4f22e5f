"""GPU parity for the speaking stock scenarios of SURVEY 8f row f1: simple_speaker_listener,
simple_reference (ragged actions, agent.state.c observed by the other agent, per-agent goal_b) and
simple_crypto (nobody moves, messages scored against the goal colour, benchmark info)
through the C ABI, against the golden vectors generated from the reference and the CPU oracle.
f64 <= 1e-9 over free-running 25-step episodes; f32 <= 1e-4 (no thresholds in these scenarios)."""
import os

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.gpu_util import HipSim, dev, golden_soa, lib, set_force_tile, stats_rows

CASES = [('g10_simple_speaker_listener', O.speaker_listener_spec), ('g11_simple_reference', O.simple_reference_spec),
         ('g12_simple_crypto', O.simple_crypto_spec)]
SPECS = pytest.mark.parametrize('mk', [O.speaker_listener_spec, O.simple_reference_spec, O.simple_crypto_spec],
                                ids=['speaker_listener', 'reference', 'crypto'])


def _messages(rng, spec, B):
    """Random action rows; every 7th env sends an all-zero message from agent (env // 7) % A (the
    switch simple_crypto's rewards have, simple_crypto.py:104,109,119)."""
    an = rng.rand(sum(spec.act_dims), B)
    envs = np.arange(0, B, 7)
    offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
    for a in range(spec.n_agents):
        if spec.silent[a]:
            continue
        sel = envs[(envs // 7) % spec.n_agents == a]
        r0 = offs[a] + spec.comm_row0(a)
        an[r0:r0 + spec.dim_c, sel] = 0.0
    return an


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _maxerr(got, want):
    return float(np.max(np.abs(got - want))) if got.size else 0.0


@pytest.fixture(params=[0, 1], ids=['auto_kernel', 'tile_kernel'])
def force_tile(request):
    """The stock shapes run the register kernel (mpe_comm_kernel.h); the hook routes them through the
    generic LDS-tile kernel too."""
    set_force_tile(request.param)
    yield request.param
    set_force_tile(0)


@SPECS
def test_register_and_tile_kernels_bit_identical(mk):
    spec = mk()
    B = 4096 + 11
    rng = np.random.RandomState(8)
    pos = rng.uniform(-1, 1, (spec.n_entities, 2, B))
    vel = rng.uniform(-0.5, 0.5, (spec.n_agents, 2, B))
    goal = rng.randint(0, spec.n_landmarks, (spec.goal_rows, B))
    acts = np.stack([_messages(rng, spec, B) for _ in range(4)])
    for dtype in (torch.float64, torch.float32):
        res = []
        for ft in (0, 1):
            set_force_tile(ft)
            sim = HipSim(spec, B, dtype)
            sim.set_state(pos, vel, goal)
            sim.comm[:] = 0.25
            sim.observe(comm=sim.comm)
            out = [t.copy() for t in sim.host()]
            for t in range(4):
                sim.env_step(dev(acts[t], dtype), comm=sim.comm)
            res.append(out + list(sim.host()) + [sim.comm.cpu().numpy()])
        set_force_tile(0)
        for x, y in zip(*res):
            assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize('name,mk', CASES)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_trajectory_matches_reference(golden_dir, name, mk, dtype, tol, force_tile):
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    goal = np.maximum(g['goal'].T[:spec.goal_rows], 0)
    sim = HipSim(spec, S, dtype)
    sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), goal)
    sim.comm[:, :, :S] = dev(golden_soa(g, 'comm0'), dtype)
    sim.observe(rew=False, info=False, comm=sim.comm)
    assert _maxerr(sim.host()[2], g['obs0'].T) <= (0.0 if dtype == torch.float64 else tol)
    worst = 0.0
    for t in range(T):
        if dtype == torch.float32 and t:                                  # teacher-forced
            sim.set_state(golden_soa(g, 'pos', t - 1), golden_soa(g, 'vel', t - 1), goal)
        sim.env_step(dev(g['act'][:, t].T, dtype), comm=sim.comm)
        pos, vel, obs, rew, info = sim.host()
        errs = (_maxerr(pos, golden_soa(g, 'pos', t)), _maxerr(vel, golden_soa(g, 'vel', t)),
                _maxerr(sim.comm[..., :S].double().cpu().numpy(), golden_soa(g, 'comm', t)),
                _maxerr(obs, g['obs'][:, t].T),
                _maxerr(info, golden_soa(g, 'info', t)) if 'info' in g.files else 0.0)
        rerr = np.abs(rew - golden_soa(g, 'rew', t)) / np.maximum(1.0, np.abs(golden_soa(g, 'rew', t)))
        worst = max(worst, *errs, float(rerr.max()))
        assert max(errs) <= tol and rerr.max() <= tol, (name, t, errs, rerr.max())
    print('%s %s worst |err| = %.3e' % (name, dtype, worst))


@SPECS
@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_split_entry_points_equal_fused_and_random_batch_matches_oracle(mk, dtype, force_tile):
    """mpe_world_step + mpe_observe (which reads agent.state.c from the comm buffer) == mpe_env_step
    (which reads it from the actions), bit for bit, on a ragged batch with ld > B; f64 also against
    the batched oracle; statistics rows hold the reward sums."""
    spec = mk()
    B, ld, steps = 5000 + 19, 5120, 3
    rng = np.random.RandomState(5)
    pos = rng.uniform(-1, 1, (spec.n_entities, 2, B))
    vel = rng.uniform(-0.5, 0.5, (spec.n_agents, 2, B))
    goal = rng.randint(0, spec.n_landmarks, (spec.goal_rows, B))
    a, b = HipSim(spec, B, dtype, ld), HipSim(spec, B, dtype, ld)
    a.set_state(pos, vel, goal)
    b.set_state(pos, vel, goal)
    rows, width = stats_rows(a.desc, B, 8 if dtype == torch.float64 else 4), lib.mpe_stats_width(a.desc)
    stats = torch.zeros(rows, width, dtype=torch.float64, device='cuda')
    bo = O.BatchedOracle(spec)
    want_sum = np.zeros(spec.n_agents)
    for t in range(steps):
        an = _messages(rng, spec, B)
        act = torch.zeros(sum(spec.act_dims), ld, dtype=dtype, device='cuda')
        act[:, :B] = dev(an, dtype)
        a.env_step(act, comm=a.comm, stats=stats)
        b.world_step(act, comm=b.comm)
        b.observe(comm=b.comm)
        for x, y in zip(a.host(), b.host()):
            assert np.array_equal(x, y, equal_nan=True)
        assert torch.equal(a.comm[..., :B], b.comm[..., :B])
        if dtype == torch.float64:
            pos, vel, obs, rew, info = bo.env_step(pos, vel, an, goal=goal)
            gp, gv, go, gr, gi = a.host()
            assert max(_maxerr(gp, pos), _maxerr(gv, vel), _maxerr(go, np.concatenate(obs, 0)), _maxerr(gr, rew)) <= 1e-9
            if spec.info_width:
                assert _maxerr(gi, info) == 0.0
            assert _maxerr(a.comm[..., :B].cpu().numpy(), bo.comm_state(an)) == 0.0
            want_sum += rew.sum(1)
    for t_ in (a.obs, a.rew):
        assert torch.isnan(t_[..., B:]).all()                              # padding never written
    assert torch.isnan(a.comm[..., B:]).all()
    if dtype == torch.float64:
        out = torch.empty(width, dtype=torch.float64, device='cuda')
        assert lib.mpe_stats_reduce(stats.data_ptr(), rows, width, out.data_ptr(), a.stream) == 0
        np.testing.assert_allclose(out.cpu().numpy()[:spec.n_agents], want_sum, rtol=1e-9)
        assert out[2 * spec.n_agents + 1].item() == steps * B


@SPECS
def test_reset_draws_goals_from_the_philox_oracle(mk):
    spec = mk()
    B = 2048
    sim = HipSim(spec, B, torch.float64)
    sim.reset(seed=11, episode=4, env_id0=100)
    ids = np.arange(B, dtype=np.uint64) + 100
    assert np.array_equal(sim.goal[:spec.goal_rows].cpu().numpy().reshape(spec.goal_rows, B),
                          O.reset_goals(spec, 11, 4, ids).reshape(spec.goal_rows, B))
    assert np.array_equal(sim.pos.cpu().numpy(), O.reset_positions(spec, 11, 4, ids))
    g = sim.goal[:spec.goal_rows].cpu().numpy()
    assert g.min() == 0 and g.max() == spec.n_landmarks - 1 and (spec.goal_rows == 1 or (g[0] != g[1]).mean() > 0.4)


CNOISE_CASES = [('g14_simple_reference_cnoise', O.simple_reference_spec), ('g14_simple_crypto_cnoise', O.simple_crypto_spec),
                ('g14_simple_world_comm_cnoise', O.simple_world_comm_spec)]


@pytest.mark.parametrize('name,mk', CNOISE_CASES)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_c_noise_trajectory_matches_reference(golden_dir, name, mk, dtype, tol):
    """agent.c_noise (core.py:229): state.c = action.c + c_noise * randn, with the reference's recorded draws
    passed in as the cnoise buffer; the noisy state.c is what gets observed (and scored, simple_crypto).
    A world with comm noise runs the generic tile kernel; split entry points equal the fused one."""
    import dataclasses
    g = _load(golden_dir, name)
    spec = mk()
    spec = dataclasses.replace(spec, c_noise=[float(g['spec_c_noise'])] * spec.n_agents)
    S, T = g['act'].shape[:2]
    goal = np.maximum(g['goal'].T[:spec.goal_rows], 0) if 'goal' in g.files else None
    sim, two = HipSim(spec, S, dtype), HipSim(spec, S, dtype)
    for x in (sim, two):
        x.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), goal)
    for t in range(T):
        if dtype == torch.float32 and t:                                  # teacher-forced
            for x in (sim, two):
                x.set_state(golden_soa(g, 'pos', t - 1), golden_soa(g, 'vel', t - 1), goal)
        act, cn = dev(g['act'][:, t].T, dtype), dev(golden_soa(g, 'cnoise', t), dtype)
        sim.env_step(act, comm=sim.comm, cnoise=cn)
        two.world_step(act, comm=two.comm, cnoise=cn)
        two.observe(comm=two.comm)
        for x, y in zip(sim.host(), two.host()):
            assert np.array_equal(x, y, equal_nan=True)
        pos, vel, obs, rew, info = sim.host()
        assert _maxerr(pos, golden_soa(g, 'pos', t)) <= tol
        assert _maxerr(sim.comm[..., :S].double().cpu().numpy(), golden_soa(g, 'comm', t)) <= tol
        if spec.kind == O.WORLD_COMM and dtype == torch.float32:
            continue                                                      # thresholds: covered by test_gpu_world_comm
        assert _maxerr(obs, g['obs'][:, t].T) <= tol
        rerr = np.abs(rew - golden_soa(g, 'rew', t)) / np.maximum(1.0, np.abs(golden_soa(g, 'rew', t)))
        assert rerr.max() <= tol, (name, t, rerr.max())
    with_noise = sim.comm[0, :, :S].double().cpu().numpy()
    offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
    clean = g['act'][:, T - 1, offs[0] + spec.comm_row0(0):offs[1]].T
    assert np.abs(with_noise - clean).max() > 0.01                        # the noise really went in


def test_c_noise_through_the_python_api():
    """agent.c_noise set on a world from make_env: fresh normals every step (or world.cnoise from the caller);
    silent agents stay at zero; without c_noise nothing changes."""
    from marl_amd import make_env
    B = 512
    env = make_env('simple_reference', num_envs=B, dtype=torch.float64, seed=1)
    env.reset()
    w = env.world
    for a in w.agents:
        a.c_noise = 0.5
    w.invalidate_descriptor()
    gen = torch.Generator(device='cuda').manual_seed(0)
    action_n = [torch.rand(B, 15, generator=gen, device='cuda', dtype=torch.float64) for _ in range(2)]
    w.cnoise = torch.full((2, 10, B), 2.0, dtype=torch.float64, device='cuda')
    obs_n, _, _, _ = env.step(action_n)
    assert torch.allclose(w.agents[0].state.c, action_n[0][:, 5:] + 1.0, rtol=0, atol=1e-15)
    assert torch.equal(obs_n[1][:, -10:], w.agents[0].state.c)            # the other agent hears the noisy message
    w.cnoise = None
    env.step(action_n)
    noise = (w.agents[1].state.c - action_n[1][:, 5:]) / 0.5
    assert abs(float(noise.mean())) < 0.05 and abs(float(noise.std()) - 1.0) < 0.05



@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('flags', [0, 1], ids=['register_kernel', 'tile_kernel'])
@pytest.mark.parametrize('name', ['h_S1L2F2_10Targets_wide', 'h_L1F2_3Targets_100observe', 'm_S1L2F4_MovingTarget',
                                  'm_L1F4_MovingTarget', 'g13_simple_world_comm', 'g10_simple_speaker_listener',
                                  'g11_simple_reference'])
def test_silent_comm_zero_flag_leaves_silent_rows_alone(golden_dir, name, flags, dtype):
    """MPE_FLAG_SILENT_COMM_ZERO: agent.state.c of a silent agent is zeros(dim_c) after every step (core.py:226-227); with
    the flag the caller vouches that those rows hold zeros already and the step does not touch them (here: a sentinel
    survives), the speaking agents' rows and every other output are bit-identical to a step without the flag, which
    writes the zeros whatever the rows held."""
    from tests.gpu_util import set_call_flags
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    spec = (O.hier_spec_from_golden(g) if name[0] in 'hm' else
            {'g13': O.simple_world_comm_spec, 'g10': O.speaker_listener_spec, 'g11': O.simple_reference_spec}[name[:3]]())
    S = g['act'].shape[0]
    A, C = spec.n_agents, spec.dim_c
    act = dev(np.ascontiguousarray(g['act'][:, 0].T), dtype)
    outs = []
    for extra in (0, 8):
        set_call_flags(flags | extra)
        try:
            sim = HipSim(spec, S, dtype)
            goal = g['goal'].T[:spec.goal_rows] if 'goal' in g.files and spec.goal_rows else None
            sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), goal,
                          golden_soa(g, 'lforce0') if 'lforce0' in g.files else None)
            comm = torch.full((A, C, S), 7.5, dtype=dtype, device='cuda')
            sim.env_step(act, comm=comm)
            torch.cuda.synchronize()
            outs.append((comm.clone(), sim.pos.clone(), sim.vel.clone(), sim.obs.clone(), sim.rew.clone(), sim.info.clone(),
                         None if sim.lforce is None else sim.lforce.clone()))
        finally:
            set_call_flags(0)
    (c0, *rest0), (c1, *rest1) = outs
    for i in range(A):
        if spec.silent[i]:
            assert (c0[i] == 0).all() and (c1[i] == 7.5).all(), i
        else:
            assert torch.equal(c0[i], c1[i]) and not (c0[i] == 7.5).all(), i
    for x, y in zip(rest0, rest1):
        assert (x is None and y is None) or torch.equal(x.nan_to_num(nan=123.0), y.nan_to_num(nan=123.0))
    assert lib.mpe_stats_rows(sim.desc, S, 4, 32) == -6          # an unknown flag bit: MPE_ERR_ARG


def test_world_keeps_silent_comm_rows_zero_without_rewriting_them():
    """The Python World passes the flag (its comm buffer starts as zeros); a write through AgentState.c's setter on a
    silent agent is undone in front of the next step, like the reference's state.c = zeros (core.py:226-227), and a
    graph replay does the same."""
    from marl_amd import _hip, make_env
    from marl_amd.graph import CapturedSteps
    env = make_env('S1L2F2_10Targets_wide', benchmark=True, num_envs=300, dtype=torch.float64, seed=3)
    env.reset()
    w = env.world
    assert w._buf.flags & _hip.FLAG_SILENT_COMM_ZERO
    w.set_flags(_hip.FLAG_FORCE_TILE)
    assert w._buf.flags == _hip.FLAG_FORCE_TILE | _hip.FLAG_SILENT_COMM_ZERO
    w.set_flags(0)
    acts = [w.new_action_buffer() for _ in range(3)]
    for a in acts:
        a.copy_(torch.rand(a.shape, dtype=torch.float64, device='cuda'))
    follower, leader = w.agents[0], w.agents[2]
    assert follower.silent and not leader.silent
    follower.state.c = np.arange(1.0, 8.0)
    assert (follower.state.c == torch.arange(1.0, 8.0, dtype=torch.float64, device='cuda')).all()
    env.step(acts[0])
    assert (follower.state.c == 0).all()
    assert torch.equal(leader.state.c.t(), acts[0][w.act_row0[2] + 5:w.act_row0[2] + 12])
    cap = CapturedSteps(env, acts)
    follower.state.c = np.ones(7)
    cap.replay()
    torch.cuda.synchronize()
    assert (follower.state.c == 0).all()
    assert torch.equal(leader.state.c.t(), acts[2][w.act_row0[2] + 5:w.act_row0[2] + 12])
