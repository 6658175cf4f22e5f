"""GPU parity tests (run on a real MI355X: pytest -m gpu).  Every test calls the HIP path through
the C ABI of libmpe_hip.so and checks it against the committed golden vectors (generated from the
reference) or against the CPU oracle on the same seeded inputs.

Bars (BASELINE.json north_star):
  f64: |hip - reference| <= 1e-9 per step over whole 25-step episodes (free-running trajectories)
  f32: <= 1e-4 relative (atol 1e-4 for quantities near zero); rewards / collision counts equal
       except on envs within 1e-5 of a collision / bound() threshold.
"""
import dataclasses
import os

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from tests.gpu_util import HipSim, dev, golden_soa, lib, set_call_flags, set_force_tile, stats_rows

F64_TOL = 1e-9

CASES = [
    ('g1_simple_tag', lambda: O.simple_tag_spec()),
    ('g2_simple_spread', lambda: O.simple_spread_spec()),
    ('g3_scaled_tag', lambda: O.simple_tag_spec(48, 16, 32)),
    ('g3d_scaled_tag_dense', lambda: O.simple_tag_spec(48, 16, 32)),
    ('g4_simple_tag_unoise', lambda: O.simple_tag_spec(u_noise=0.3)),
    ('g5_simple_tag_dense', lambda: O.simple_tag_spec()),
    ('g5_simple_spread_dense', lambda: O.simple_spread_spec()),
    ('g6_simple_tag_oob', lambda: O.simple_tag_spec()),
    ('g7_simple', lambda: O.simple_spec()),
    ('g8_simple_adversary', lambda: O.simple_adversary_spec()),
    ('g8_simple_adversary_dense', lambda: O.simple_adversary_spec()),
    ('g9_simple_push', lambda: O.simple_push_spec()),
    ('g9_simple_push_dense', lambda: O.simple_push_spec()),
]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


# every test that takes this fixture runs on each kernel that can serve the world: the one-env-per-lane register kernel
# (MPE_FLAG_WIDE), the one-agent-per-wavefront form of simple_tag / simple_spread (MPE_FLAG_SPLIT; other worlds: the
# same kernel as WIDE) and the generic LDS-tile kernel (MPE_FLAG_FORCE_TILE)
@pytest.fixture(params=[2, 4, 1], ids=['wide_kernel', 'split_kernel', 'tile_kernel'])
def force_tile(request):
    set_call_flags(request.param)
    yield request.param
    set_call_flags(0)


def _maxerr(got, want):
    return float(np.max(np.abs(got - want))) if got.size else 0.0


@pytest.mark.parametrize('name,mk', CASES)
def test_f64_trajectory_matches_reference(golden_dir, name, mk, force_tile):
    """Free-running episodes from the golden initial state; compare every output of every step."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    sim = HipSim(spec, S, torch.float64)
    sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), g['goal'])
    sim.observe(rew=False, info=False)
    assert _maxerr(sim.host()[2], g['obs0'].T) == 0.0          # pure gathers/subtractions: exact
    worst = 0.0
    for t in range(T):
        act = dev(golden_soa(g, 'act', t), torch.float64)
        noise = dev(golden_soa(g, 'noise', t), torch.float64) if 'noise' in g.files else None
        sim.env_step(act, noise)
        pos, vel, obs, rew, info = sim.host()
        errs = (_maxerr(pos, golden_soa(g, 'pos', t)), _maxerr(vel, golden_soa(g, 'vel', t)),
                _maxerr(obs, g['obs'][:, t].T), _maxerr(rew, golden_soa(g, 'rew', t)),
                _maxerr(info, golden_soa(g, 'info', t)))
        worst = max(worst, *errs)
        assert max(errs) <= F64_TOL, (name, t, errs)
    print('%s force_tile=%d worst |err| = %.3e' % (name, force_tile, worst))


@pytest.mark.parametrize('name,mk', CASES)
def test_f32_teacher_forced_matches_reference(golden_dir, name, mk, force_tile):
    """Each step starts from the golden state of the previous step (no error accumulation):
    f32 outputs within 1e-4 relative of the float64 reference; discrete outputs equal except where
    the reference itself is within 1e-5 of a threshold."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    sim = HipSim(spec, S, torch.float32)
    A = spec.n_agents
    for t in range(T):
        p0 = golden_soa(g, 'pos0') if t == 0 else golden_soa(g, 'pos', t - 1)
        v0 = golden_soa(g, 'vel0') if t == 0 else golden_soa(g, 'vel', t - 1)
        sim.set_state(p0, v0, g['goal'])
        act = dev(golden_soa(g, 'act', t), torch.float32)
        noise = dev(golden_soa(g, 'noise', t), torch.float32) if 'noise' in g.files else None
        sim.env_step(act, noise)
        pos, vel, obs, rew, info = sim.host()
        np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(vel, golden_soa(g, 'vel', t), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(obs, g['obs'][:, t].T, rtol=1e-4, atol=1e-4)
        # discrete parts: flag envs where any pair distance / |pos| is near a threshold
        P = golden_soa(g, 'pos', t)                                   # [E,2,S]
        dist = np.sqrt(((P[:A, None] - P[None, :A]) ** 2).sum(2))      # [A,A,S]
        dmin = np.add.outer(np.asarray(spec.size[:A]), np.asarray(spec.size[:A]))[..., None]
        near = (np.abs(dist - dmin) < 1e-5).any((0, 1))
        if spec.kind == O.SIMPLE_TAG:
            ax = np.abs(P[:A])
            near |= (np.abs(ax - 0.9) < 1e-5).any((0, 1)) | (np.abs(ax - 1.0) < 1e-5).any((0, 1))
        elif spec.kind == O.SIMPLE_SPREAD:
            dl = np.sqrt(((P[:A, None] - P[None, A:]) ** 2).sum(2))
            near |= (np.abs(dl - 0.1) < 1e-5).any((0, 1))
        else:
            near[:] = False                      # no thresholds in simple / simple_adversary / simple_push
        ok = ~near
        np.testing.assert_allclose(rew[:, ok], golden_soa(g, 'rew', t)[:, ok], rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(info[..., ok], golden_soa(g, 'info', t)[..., ok], rtol=1e-4, atol=2e-4)
        assert ok.mean() > 0.9


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('mk', [lambda: O.simple_tag_spec(), lambda: O.simple_spread_spec(),
                                lambda: O.simple_tag_spec(5, 3, 4), lambda: O.simple_spread_spec(5, 7),
                                lambda: O.simple_spec(), lambda: O.simple_adversary_spec(),
                                lambda: O.simple_adversary_spec(6), lambda: O.simple_push_spec(),
                                lambda: O.simple_push_spec(5, 3, 2)],
                         ids=['tag', 'spread', 'tag_5_3_4', 'spread_5_7', 'simple', 'adversary', 'adversary_6',
                              'push', 'push_5_3_2'])
def test_split_entry_points_equal_fused(mk, dtype, force_tile):
    """mpe_world_step + mpe_observe == mpe_env_step, bit for bit; ragged batch (B not a tile multiple)
    inside a larger allocation (ld > B): rows beyond B must stay untouched."""
    spec = mk()
    B, ld = 1000 + 37, 1200
    rng = np.random.RandomState(1)
    pos = rng.uniform(-0.5, 0.5, (spec.n_entities, 2, B))
    vel = rng.uniform(-0.5, 0.5, (spec.n_agents, 2, B))
    act = torch.zeros(spec.n_agents, 5, ld, dtype=dtype, device='cuda')
    act[:, :, :B] = dev(rng.rand(spec.n_agents, 5, B), dtype)
    a, b = HipSim(spec, B, dtype, ld), HipSim(spec, B, dtype, ld)
    goal = rng.randint(0, max(spec.n_landmarks, 1), B)
    a.set_state(pos, vel, goal)
    b.set_state(pos, vel, goal)
    comm = torch.full((spec.n_agents, spec.dim_c, ld), 7.0, dtype=dtype, device='cuda')
    for _ in range(3):
        a.env_step(act)
        b.world_step(act, comm=comm)
        b.observe()
    for x, y in zip(a.host(), b.host()):
        assert np.array_equal(x, y, equal_nan=True)
    assert (comm[:, :, :B] == 0).all() and (comm[:, :, B:] == 7).all()     # core.py:226-227
    for t in (a.obs, a.rew, a.info):
        assert torch.isnan(t[..., B:]).all()                               # padding never written
    assert (a.pos[..., B:] == 0).all() and (a.vel[..., B:] == 0).all()


@pytest.mark.parametrize('mk', [lambda: O.simple_tag_spec(), lambda: O.simple_spread_spec(), lambda: O.simple_spec(),
                                lambda: O.simple_adversary_spec(), lambda: O.simple_push_spec()],
                         ids=['tag', 'spread', 'simple', 'adversary', 'push'])
def test_small_and_tile_kernels_bit_identical(mk):
    """The register kernels (one env per lane; one agent per wavefront) and the LDS-tile kernel implement the same
    operation order."""
    spec = mk()
    B = 4096 + 11
    rng = np.random.RandomState(2)
    pos = rng.uniform(-0.4, 0.4, (spec.n_entities, 2, B))
    vel = rng.uniform(-1, 1, (spec.n_agents, 2, B))
    for dtype in (torch.float64, torch.float32):
        act = dev(rng.rand(spec.n_agents, 5, B), dtype)
        res = []
        for flags in (2, 4, 1):          # wide, split (tag / spread only), tile
            set_call_flags(flags)
            sim = HipSim(spec, B, dtype)
            sim.set_state(pos, vel, np.arange(B) % max(spec.n_landmarks, 1))
            for _ in range(5):
                sim.env_step(act)
            res.append(sim.host())
        set_call_flags(0)
        for other in res[1:]:
            for x, y in zip(res[0], other):
                assert np.array_equal(x, y, equal_nan=True)


def test_config2_simple_spread_65536_f64_vs_oracle():
    """BASELINE config 2: simple_spread N=3, 65,536 envs, fp64, 25-step episode, every output of
    every step within 1e-9 of the CPU oracle fed the same initial state and actions."""
    spec = O.simple_spread_spec()
    B, T = 65536, 25
    rng = np.random.RandomState(3)
    pos = rng.uniform(-1, 1, (6, 2, B))
    vel = np.zeros((3, 2, B))
    bo = O.BatchedOracle(spec)
    sim = HipSim(spec, B, torch.float64)
    sim.set_state(pos, vel)
    worst = 0.0
    for t in range(T):
        logits = rng.randn(3, 5, B)
        act = np.exp(logits) / np.exp(logits).sum(1, keepdims=True)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act)
        sim.env_step(dev(act, torch.float64))
        hp, hv, ho, hr, hi = sim.host()
        errs = (_maxerr(hp, pos), _maxerr(hv, vel), _maxerr(ho, np.concatenate(obs, 0)), _maxerr(hr, rew),
                _maxerr(hi, info))
        worst = max(worst, *errs)
        assert max(errs) <= F64_TOL, (t, errs)
    print('config2 worst |err| = %.3e' % worst)


def test_config3_simple_tag_1M_f32_properties():
    """BASELINE config 3 size (1,048,576 envs, f32): a 4,096-env sample against the oracle (1e-4),
    plus size-independent properties on the full batch: run-to-run determinism, adversary reward ==
    10 * sum of adversary collision counts, observation rows consistent with the state."""
    spec = O.simple_tag_spec()
    B, T = 1 << 20, 25
    g = torch.Generator(device='cuda').manual_seed(5)
    pos0 = torch.rand(6, 2, B, generator=g, device='cuda') * 1.2 - 0.6
    vel0 = torch.zeros(4, 2, B, device='cuda')
    acts = [torch.softmax(torch.randn(4, 5, B, generator=g, device='cuda'), dim=1) for _ in range(T)]
    runs = []
    for rep in range(2):
        sim = HipSim(spec, B, torch.float32)
        sim.pos.copy_(pos0)
        sim.vel.copy_(vel0)
        for t in range(T):
            sim.env_step(acts[t])
        runs.append(sim)
    a, b = runs
    for x, y in ((a.pos, b.pos), (a.vel, b.vel), (a.obs, b.obs), (a.rew, b.rew), (a.info, b.info)):
        assert torch.equal(x, y)
    assert torch.isfinite(a.obs).all() and torch.isfinite(a.rew).all()
    # reward / info consistency (simple_tag.py:57-66,115-129)
    assert torch.equal(a.rew[0], 10 * a.info[:3, 0].sum(0)) and torch.equal(a.rew[0], a.rew[1])
    assert (a.info[3] == 0).all() and (a.info[:3].sum() > 0)
    # obs rows vs state (simple_tag.py:131-147): agent 3 (good): [vel, pos, lm-pos x2, others-pos x3]
    o3 = a.obs[48:62]
    assert torch.equal(o3[0:2], a.vel[3]) and torch.equal(o3[2:4], a.pos[3])
    assert torch.equal(o3[4:6], a.pos[4] - a.pos[3]) and torch.equal(o3[8:10], a.pos[0] - a.pos[3])
    assert torch.equal(a.obs[14:16], a.vel[3])                          # adversary 0 sees the prey's velocity
    # sample vs oracle (f64 oracle on the f32 inputs), free-running 25 steps
    idx = torch.arange(0, B, B // 4096, device='cuda')[:4096]
    bo = O.BatchedOracle(spec)
    p, v = pos0[:, :, idx].double().cpu().numpy(), vel0[:, :, idx].double().cpu().numpy()
    for t in range(T):
        p, v, obs, rew, info = bo.env_step(p, v, acts[t][:, :, idx].double().cpu().numpy())
    hp = a.pos[:, :, idx].double().cpu().numpy()
    hv = a.vel[:, :, idx].double().cpu().numpy()
    # chaotic contacts amplify f32 rounding over 25 free-running steps: compare the bulk tightly and
    # allow a small fraction of envs (those that went through a hard contact) a looser bound
    err = np.maximum(np.abs(hp - p).max((0, 1)), np.abs(hv - v).max((0, 1)))
    assert np.mean(err < 1e-4) > 0.97, np.mean(err < 1e-4)
    assert np.median(err) < 5e-6


def test_config5_scaled_tag_f64_and_f32_vs_oracle():
    """BASELINE config 5 shape (48 adversaries, 16 good, 32 landmarks): f64 spot check on 256 envs
    vs the oracle (1e-9), f32 teacher-forced (1e-4), dense starts so contacts fire."""
    spec = O.simple_tag_spec(48, 16, 32)
    B = 256 + 3
    rng = np.random.RandomState(7)
    pos = rng.uniform(-1.0, 1.0, (96, 2, B))
    vel = rng.uniform(-0.3, 0.3, (64, 2, B))
    bo = O.BatchedOracle(spec)
    s64, s32 = HipSim(spec, B, torch.float64), HipSim(spec, B, torch.float32)
    s64.set_state(pos, vel)
    for t in range(4):
        act = rng.rand(64, 5, B)
        s32.set_state(pos, vel)
        s32.env_step(dev(act, torch.float32))
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act)
        s64.env_step(dev(act, torch.float64))
        hp, hv, ho, hr, hi = s64.host()
        errs = (_maxerr(hp, pos), _maxerr(hv, vel), _maxerr(ho, np.concatenate(obs, 0)), _maxerr(hr, rew),
                _maxerr(hi, info))
        assert max(errs) <= F64_TOL, (t, errs)
        fp, fv, fo, fr, fi = s32.host()
        np.testing.assert_allclose(fp, pos, rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(fv, vel, rtol=1e-4, atol=2e-4)
        np.testing.assert_allclose(fo, np.concatenate(obs, 0), rtol=1e-4, atol=2e-4)
        # 768 (good, adversary) pairs per env: an f32 flip of one strict '<' moves a count by 1, so
        # compare per element and allow the few near-threshold envs
        assert np.mean(np.abs(fr - rew) <= 1e-3) > 0.98 and np.mean(fi == info) > 0.98
    assert (rew[:48] > 0).any() and (rew[48:] <= -10).any()


def test_config5_scaled_tag_262144_f32_properties():
    """BASELINE config 5 at its full size (262,144 envs of 48 + 16 agents / 32 landmarks, f32, the LDS-tile kernel with
    the 16-byte observation stores): size-independent properties on the whole batch -- run-to-run determinism, every
    observation row of four agents rebuilt exactly from the state (simple_tag.py:131-147), reward / benchmark
    consistency (:57-66, 115-129) -- and a 256-env sample of the last step against the oracle."""
    spec = O.simple_tag_spec(48, 16, 32)
    A, L, E, B, T = 64, 32, 96, 1 << 18, 3
    g = torch.Generator(device='cuda').manual_seed(9)
    pos0 = torch.rand(E, 2, B, generator=g, device='cuda') * 1.6 - 0.8
    vel0 = torch.rand(A, 2, B, generator=g, device='cuda') * 0.4 - 0.2
    acts = [torch.softmax(torch.randn(A, 5, B, generator=g, device='cuda'), dim=1) for _ in range(T)]
    sim = HipSim(spec, B, torch.float32)
    first = None
    for rep in range(2):
        sim.pos.copy_(pos0)
        sim.vel.copy_(vel0)
        for t in range(T):
            if t == T - 1:
                before = (sim.pos.clone(), sim.vel.clone())
            sim.env_step(acts[t])
        if first is None:
            first = tuple(x.clone() for x in (sim.pos, sim.vel, sim.rew, sim.info))
            obs_sum = sim.obs.sum(dtype=torch.float64)
            obs_first = sim.obs[:452].clone()                            # agents 0 and 1 (15 GB: not cloned whole)
    for x, y in zip(first, (sim.pos, sim.vel, sim.rew, sim.info)):
        assert torch.equal(x, y)
    assert torch.equal(obs_first, sim.obs[:452]) and obs_sum == sim.obs.sum(dtype=torch.float64)
    assert torch.isfinite(sim.rew).all()
    # observation rows vs state, exact (f32 subtractions of the same operands)
    row0 = lambda i: 226 * i if i < 48 else 226 * 48 + 224 * (i - 48)
    for i in (0, 47, 48, 63):
        o = sim.obs[row0(i):row0(i) + (226 if i < 48 else 224)]
        others = [j for j in range(A) if j != i]
        want = torch.cat([sim.vel[i], sim.pos[i], (sim.pos[A:] - sim.pos[i]).reshape(2 * L, B),
                          (sim.pos[others] - sim.pos[i]).reshape(2 * (A - 1), B),
                          sim.vel[[j for j in range(48, A) if j != i]].reshape(-1, B)])
        assert want.shape == o.shape and torch.equal(o, want), i
    # every adversary gets 10 x (all good-adversary contacts); its benchmark count is its own contacts; good agents report 0
    total = sim.info[:48, 0].sum(0)
    assert (total > 0).any() and (sim.info[48:] == 0).all()
    for a in (0, 17, 47):
        assert torch.equal(sim.rew[a], 10 * total)
    assert (sim.rew[48:] <= 0).all()
    # the last step of a 256-env sample, from the same pre-step state, against the oracle (f32 vs f64: 1e-4)
    idx = torch.arange(0, B, B // 256, device='cuda')[:256]
    p, v = (x[:, :, idx].double().cpu().numpy() for x in before)
    p, v, obs, rew, info = O.BatchedOracle(spec).env_step(p, v, acts[T - 1][:, :, idx].double().cpu().numpy())
    np.testing.assert_allclose(sim.pos[:, :, idx].double().cpu().numpy(), p, rtol=1e-4, atol=1e-4)
    np.testing.assert_allclose(sim.vel[:, :, idx].double().cpu().numpy(), v, rtol=1e-4, atol=2e-4)
    np.testing.assert_allclose(sim.obs[:, idx].double().cpu().numpy(), np.concatenate(obs, 0), rtol=1e-4, atol=2e-4)
    assert np.mean(np.abs(sim.rew[:, idx].double().cpu().numpy() - rew) <= 1e-3) > 0.98


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_reset_matches_philox_oracle(dtype, force_tile):
    """mpe_reset_* draws exactly the oracle's Philox stream; masked envs untouched; the result does
    not depend on how envs are split into calls (env_id0)."""
    spec = O.simple_tag_spec()
    B = 5000
    sim = HipSim(spec, B, dtype)
    sim.vel.fill_(3.0)
    sim.reset(seed=0x1234567890ABCDEF, episode=7)
    want = O.reset_positions(spec, 0x1234567890ABCDEF, 7, np.arange(B, dtype=np.uint64),
                             np.float64 if dtype == torch.float64 else np.float32)
    assert np.array_equal(sim.pos.cpu().numpy(), want)
    assert (sim.vel == 0).all()
    # shard independence
    lo = HipSim(spec, 2000, dtype)
    lo.reset(seed=0x1234567890ABCDEF, episode=7, env_id0=3000)
    assert torch.equal(lo.pos, sim.pos[:, :, 3000:5000])
    big = HipSim(spec, 16, dtype)
    big.reset(seed=1, episode=0, env_id0=(1 << 40) + 5)
    want = O.reset_positions(spec, 1, 0, np.arange(16, dtype=np.uint64) + np.uint64((1 << 40) + 5),
                             np.float64 if dtype == torch.float64 else np.float32)
    assert np.array_equal(big.pos.cpu().numpy(), want)
    # goal landmark draw (simple_adversary / simple_push): same stream, entity word 0xFFFFFFFF
    adv = O.simple_adversary_spec(5)
    ga = HipSim(adv, 3000, dtype)
    ga.reset(seed=99, episode=2, env_id0=17)
    want_goal = O.reset_goals(adv, 99, 2, np.arange(3000, dtype=np.uint64) + np.uint64(17))
    assert np.array_equal(ga.goal[0].cpu().numpy(), want_goal)
    assert set(np.unique(want_goal)) == {0, 1, 2, 3}
    # mask
    before = sim.pos.clone()
    mask = (torch.arange(B, device='cuda') % 3 == 0).to(torch.uint8)
    sim.vel.fill_(1.0)
    sim.reset(seed=5, episode=8, mask=mask)
    m = mask.bool()
    assert torch.equal(sim.pos[:, :, ~m], before[:, :, ~m]) and not torch.equal(sim.pos[:, :, m], before[:, :, m])
    assert (sim.vel[:, :, m] == 0).all() and (sim.vel[:, :, ~m] == 1).all()


@pytest.mark.parametrize('mk', [lambda: O.simple_tag_spec(), lambda: O.simple_spread_spec(),
                                lambda: O.simple_tag_spec(6, 2, 3)], ids=['tag', 'spread', 'tag_6_2_3'])
def test_episode_stats_rows_and_reduce(mk, force_tile):
    """Per-workgroup statistics rows + deterministic reduce == sums of the per-env outputs."""
    spec = mk()
    A = spec.n_agents
    B = 3000 + 13
    rng = np.random.RandomState(11)
    sim = HipSim(spec, B, torch.float32)
    sim.set_state(rng.uniform(-0.3, 0.3, (spec.n_entities, 2, B)), np.zeros((A, 2, B)))
    rows = stats_rows(sim.desc, B, 4)
    W = lib.mpe_stats_width(sim.desc)
    assert W == 2 * A + 2 and rows >= 1
    stats = torch.zeros(rows, W, dtype=torch.float64, device='cuda')
    want = torch.zeros(W, dtype=torch.float64, device='cuda')
    for t in range(6):
        sim.env_step(dev(rng.rand(A, 5, B), torch.float32), stats=stats)
        want[:A] += sim.rew[:, :B].double().sum(1)
        col = 0 if spec.kind == O.SIMPLE_TAG else 1
        want[A:2 * A] += sim.info[:, col, :B].double().sum(1)
        if spec.kind == O.SIMPLE_TAG:
            want[2 * A] += sim.info[:, 0, :B].double().sum()
        else:
            want[2 * A] += sim.info[0, 3, :B].double().sum()
        want[2 * A + 1] += B
    out = torch.empty(W, dtype=torch.float64, device='cuda')
    assert lib.mpe_stats_reduce(stats.data_ptr(), rows, W, out.data_ptr(), sim.stream) == 0
    torch.testing.assert_close(out, want, rtol=1e-12, atol=1e-9)
    out2 = torch.empty_like(out)
    lib.mpe_stats_reduce(stats.data_ptr(), rows, W, out2.data_ptr(), sim.stream)
    assert torch.equal(out, out2)
    assert out[2 * A + 1].item() == 6 * B and out[A:2 * A].sum().item() > 0


def test_f64_contact_force_relative_accuracy(force_tile):
    """The f64 pair force uses the library's own exp / log1p / sqrt / quotient forms (mpe_device.h: exp_neg_f64,
    log1p_unit_f64, sqrt_rcp_f64, quot) instead of ocml's.  Their error is relative, so it is checked as such over the
    whole range of get_collision_force (core.py:233-249): separations from 1e-7 (deep overlap, x = +150) through the
    contact threshold to the softplus tail, where the force is 1e-300 and below (KAT-4 of SURVEY 8c: the tail is not an
    exact zero) and finally underflows.  Bar: 8 ulp of the oracle's float64 value, exact zeros where it has them."""
    spec = O.simple_tag_spec()
    B = 60000
    rng = np.random.RandomState(11)
    sep = np.concatenate([10.0 ** rng.uniform(-7, -1, B // 4),        # overlapping
                          0.15 + rng.uniform(-0.05, 0.05, B // 4),      # around dist_min = 0.075 + 0.075
                          0.15 + rng.uniform(0.0, 0.8, B // 4),         # the tail, down to the smallest normals
                          0.15 + rng.uniform(0.7, 3.0, B - 3 * (B // 4))])  # subnormal results, then exact zeros
    ang = rng.uniform(0, 2 * np.pi, B)
    pos = np.zeros((6, 2, B))
    pos[0] = rng.uniform(-1, 1, (2, B))
    pos[1, 0] = pos[0, 0] + sep * np.cos(ang)
    pos[1, 1] = pos[0, 1] + sep * np.sin(ang)
    for e in range(2, 6):                                               # everybody else: exact-zero forces
        pos[e, 0] = 20.0 * e
    vel = np.zeros((4, 2, B))
    act = np.zeros((4, 5, B))
    sim = HipSim(spec, B, torch.float64)
    sim.set_state(pos, vel)
    sim.world_step(dev(act, torch.float64))
    _, want = O.BatchedOracle(spec).world_step(pos, vel, act)
    got = sim.host()[1]
    w, g = want[0], got[0]                                              # agent 0 is the one pushed (one-sided)
    assert np.isfinite(w).all() and np.isfinite(g).all()
    # subnormal results: the penalty is rounded to a multiple of 4.9e-324 before it is scaled by contact_force * d / dist
    # (up to 100) and dt, so one subnormal step of difference in exp() shows as up to ~10 steps here
    tiny = np.abs(w) < 2.3e-308
    sub = float(np.max(np.abs(g - w)[tiny])) / 4.95e-324
    print('f64 pair force, subnormal results: max difference %.0f subnormal steps' % sub)
    assert sub <= 64
    rel = np.abs(g - w)[~tiny] / np.abs(w)[~tiny]
    print('f64 pair force: max relative error %.2e (%.1f ulp), %d exact zeros, %d subnormal'
          % (rel.max(), rel.max() / 2.2e-16, int((w == 0).sum()), int((tiny & (w != 0)).sum())))
    assert rel.max() <= 8 * 2.2e-16
    assert (w == 0).sum() > 1000 and (np.abs(w[~tiny]) < 1e-290).sum() > 100   # the sweep reached both ends
    assert np.array_equal(got[1:], want[1:])                            # nobody else moves


def test_coincident_entities_give_nan_like_the_reference():
    """dist == 0 -> delta/dist = NaN in the reference (core.py:246); parity means not 'fixing' it."""
    spec = O.simple_tag_spec()
    pos = np.zeros((6, 2, 2))
    pos[:, 0, :] = np.arange(6)[:, None] * 1.0
    pos[1, :, 0] = pos[0, :, 0]                 # env 0: agents 0 and 1 coincide
    sim = HipSim(spec, 2, torch.float64)
    sim.set_state(pos, np.zeros((4, 2, 2)))
    sim.env_step(dev(np.zeros((4, 5, 2)), torch.float64))
    p, v = O.BatchedOracle(spec).world_step(pos, np.zeros((4, 2, 2)), np.zeros((4, 5, 2)))
    hp, hv = sim.host()[:2]
    assert np.isnan(p[0, :, 0]).all() and np.isnan(hp[0, :, 0]).all()
    assert np.array_equal(np.isnan(hp), np.isnan(p)) and np.array_equal(np.isnan(hv), np.isnan(v))


def test_c_abi_rejects_bad_arguments():
    spec = O.simple_tag_spec()
    sim = HipSim(spec, 8, torch.float64)
    act = torch.zeros(4, 5, 8, dtype=torch.float64, device='cuda')
    f = lib.mpe_env_step_f64
    good = dict(act=act, obs=sim.obs, rew=sim.rew, info=sim.info)
    assert f(sim.desc, 8, 8, sim.buffers(**good), sim.stream) == 0
    assert f(sim.desc, 0, 8, sim.buffers(**good), sim.stream) == -3
    assert f(sim.desc, 8, 4, sim.buffers(**good), sim.stream) == -3
    assert f(sim.desc, 8, 8, sim.buffers(**dict(good, act=None)), sim.stream) == -1
    assert f(sim.desc, 8, 8, None, sim.stream) == -1
    b = sim.buffers(**good)
    b.pos = sim.pos.data_ptr() + 4
    assert f(sim.desc, 8, 8, b, sim.stream) == -4
    d = type(sim.desc).from_buffer_copy(sim.desc)
    d.n_agents = 100
    assert f(d, 8, 8, sim.buffers(**good), sim.stream) == -2
    d = type(sim.desc).from_buffer_copy(sim.desc)
    d.movable[5] = 1                                                      # a movable landmark needs buffers.lvel
    assert f(d, 8, 8, sim.buffers(**good), sim.stream) == -1
    torch.cuda.synchronize()


@pytest.mark.parametrize('mk', [lambda: O.simple_spec(), lambda: O.simple_adversary_spec(), lambda: O.simple_push_spec()],
                         ids=['simple', 'adversary', 'push'])
def test_episode_stats_of_the_goal_scenarios(mk, force_tile):
    """Statistics rows of simple / simple_adversary / simple_push: reward sums per agent and the env-step
    count (the count once came out as one per wavefront: a ballot evaluated under `if (lane == 0)`)."""
    spec = mk()
    A, B = spec.n_agents, 3000 + 13
    rng = np.random.RandomState(12)
    sim = HipSim(spec, B, torch.float32)
    sim.set_state(rng.uniform(-0.5, 0.5, (spec.n_entities, 2, B)), np.zeros((A, 2, B)),
                  rng.randint(0, spec.n_landmarks, B))
    rows, W = stats_rows(sim.desc, B, 4), lib.mpe_stats_width(sim.desc)
    stats = torch.zeros(rows, W, dtype=torch.float64, device='cuda')
    want = torch.zeros(A, dtype=torch.float64, device='cuda')
    for t in range(4):
        sim.env_step(dev(rng.rand(A, 5, B), torch.float32), stats=stats)
        want += sim.rew[:, :B].double().sum(1)
    out = torch.empty(W, dtype=torch.float64, device='cuda')
    assert lib.mpe_stats_reduce(stats.data_ptr(), rows, W, out.data_ptr(), sim.stream) == 0
    torch.testing.assert_close(out[:A], want, rtol=1e-12, atol=1e-9)
    assert out[2 * A + 1].item() == 4 * B


MOVABLE = [('g19_simple_tag_movable_landmarks',
            lambda: dataclasses.replace(O.simple_tag_spec(), movable=[True] * 6, mass=[1, 1, 1, 1, 1, 2.5],
                                        landmark_max_speed=[None, 0.08])),
           ('g19_simple_spread_movable_landmarks',
            lambda: dataclasses.replace(O.simple_spread_spec(), movable=[True] * 3 + [True, False, True], collide=[True] * 6,
                                        size=[0.15] * 3 + [0.12] * 3, landmark_max_speed=[None, None, 0.3]))]


@pytest.mark.parametrize('name,mk', MOVABLE)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_movable_landmarks_match_reference(golden_dir, name, mk, dtype, tol):
    """core.py:211-222 integrates EVERY movable entity: a landmark made movable (no scenario file does; fixtures G19
    generated from the reference with Entity.movable / initial_mass / max_speed set on landmarks) gets a velocity, is
    damped, pushed by the colliding landmarks behind it (one-sided response), speed-clamped and moved.  Such worlds run
    the generic tile kernel.  f64 free-running within 1e-9, f32 teacher-forced within 1e-4; split entry points == fused."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    sim, two = HipSim(spec, S, dtype), HipSim(spec, S, dtype)
    for s_ in (sim, two):
        s_.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'))
        s_.lvel.copy_(dev(golden_soa(g, 'lvel0'), dtype))
    for t in range(T):
        if dtype == torch.float32 and t:
            for s_ in (sim, two):
                s_.set_state(golden_soa(g, 'pos', t - 1), golden_soa(g, 'vel', t - 1))
                s_.lvel.copy_(dev(golden_soa(g, 'lvel', t - 1), dtype))
        act = dev(golden_soa(g, 'act', t), dtype)
        sim.env_step(act)
        two.world_step(act)
        two.observe()
        pos, vel, obs, rew, info = sim.host()
        np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(vel, golden_soa(g, 'vel', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(sim.lvel.double().cpu().numpy(), golden_soa(g, 'lvel', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(obs, g['obs'][:, t].T, rtol=tol, atol=tol)
        if dtype == torch.float64:
            np.testing.assert_allclose(rew, golden_soa(g, 'rew', t), rtol=0, atol=tol)
        for a, b in zip(sim.host(), two.host()):
            assert np.array_equal(a, b, equal_nan=True)
        assert torch.equal(sim.lvel, two.lvel)
    # reset clears the landmark velocities (simple_tag.py:54), untouched rows of an immovable landmark stay zero
    sim.reset(3, 0)
    assert not sim.lvel.any() and not sim.vel.any()


def test_movable_landmark_through_the_public_api(golden_dir):
    """landmark.movable = True after make_env (+ world.invalidate_descriptor()): landmark.state.p_vel becomes a view of
    world.lvel, the trajectory is the reference's."""
    from marl_amd import make_env
    g = _load(golden_dir, 'g19_simple_tag_movable_landmarks')
    S, T = g['act'].shape[:2]
    env = make_env('simple_tag', num_envs=S, dtype=torch.float64, seed=2)
    w = env.world
    assert w.landmarks[0].state.p_vel.shape == (S, 2) and w.lvel is None
    w.landmarks[0].movable = w.landmarks[1].movable = True
    w.landmarks[1].initial_mass, w.landmarks[1].max_speed = 2.5, 0.08
    w.invalidate_descriptor()
    env.reset()
    assert w.lvel is not None and not w.lvel.any()
    w.pos.copy_(dev(golden_soa(g, 'pos0'), torch.float64))
    w.vel.copy_(dev(golden_soa(g, 'vel0'), torch.float64))
    for i, lm in enumerate(w.landmarks):
        lm.state.p_vel = dev(g['lvel0'][:, i], torch.float64)
    w.mark_dirty()
    for t in range(T):
        obs_n, rew_n, _, _ = env.step([dev(g['act'][:, t, i], torch.float64) for i in range(env.n)])
        np.testing.assert_allclose(w.pos.cpu().numpy(), golden_soa(g, 'pos', t), rtol=0, atol=1e-9)
        np.testing.assert_allclose(torch.stack([lm.state.p_vel for lm in w.landmarks]).cpu().numpy(),
                                   np.moveaxis(g['lvel'][:, t], 0, 1), rtol=0, atol=1e-9)
        np.testing.assert_allclose(torch.cat(obs_n, 1).cpu().numpy(), g['obs'][:, t], rtol=0, atol=1e-9)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('mk', ['tag', 'world_comm', 'adversary', 'spread'])
def test_residency_variants_of_the_stock_kernels_bit_identical(mk, dtype):
    """The fused one-env-per-lane kernels stream their landmark rows past the Infinity Cache when the state outgrows it
    (mpe_residency_plan follows the row stride): a small batch inside multi-million-element rows runs that variant, with
    every output equal bit for bit to the run at ld = B."""
    from tests.gpu_util import set_call_flags
    spec = {'tag': O.simple_tag_spec, 'world_comm': O.simple_world_comm_spec, 'adversary': O.simple_adversary_spec,
            'spread': O.simple_spread_spec}[mk]()
    A, E, B = spec.n_agents, spec.n_entities, 515
    nbytes = 8 if dtype == torch.float64 else 4
    rng = np.random.RandomState(17)
    pos, vel = rng.uniform(-0.4, 0.4, (E, 2, B)), rng.uniform(-0.5, 0.5, (A, 2, B))
    goal = rng.randint(0, spec.n_landmarks, B)
    acts = rng.rand(3, sum(spec.act_dims), B)
    res, plans = [], []
    try:
        set_call_flags(2)                                                # MPE_FLAG_WIDE: one env per lane at this batch size too
        for ld in (B, (6 if nbytes == 8 else 12) * 1048576 + 128):
            sim = HipSim(spec, B, dtype, ld)
            plans.append(lib.mpe_residency_plan(sim.desc, ld, nbytes))
            sim.set_state(pos, vel, goal if spec.goal_rows else None)
            for t in range(3):
                act = torch.zeros(sum(spec.act_dims), ld, dtype=dtype, device='cuda')
                act[:, :B] = dev(acts[t], dtype)
                sim.env_step(act.view(A, 5, ld) if all(spec.silent) else act)
            res.append(sim.host())
            del sim, act
            torch.cuda.empty_cache()
    finally:
        set_call_flags(0)
    assert plans[0] == 0 and plans[1] & 1
    for x, y in zip(*res):
        assert np.array_equal(x, y, equal_nan=True)
