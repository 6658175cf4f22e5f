"""GPU parity for SURVEY 8f row f2: the fork's follower / leader / super-leader pursuit scenarios
(multiagent/scenarios/L1F*.py, S1L*.py) -- leader force (core.py:153-194), masked and
distance-sorted observations, hierarchical rewards -- through the C ABI, against the golden vectors
generated from the reference and against the CPU oracle.

f64: <= 1e-9 per step over free-running 25-step episodes.  f32: <= 1e-4 relative, teacher-forced;
envs where the float64 reference is within 1e-5 of a visibility / contact / bound() threshold or
of a tie in the target ordering are excluded from the f32 comparison of the affected outputs.
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
BASES = ['L1F2_3Targets_100observe', 'S1L2F2_3Targets_100observe', 'L1F2_3Targets_075observe',
         'L1F1_10Targets_wide', 'L1F4_3Targets_wide', 'S1L2F2_10Targets_wide',
         'S1L2F2_10Targets_wide_rewardFixed',
         # round 2: the remaining files (HIER_3_10_2 / HIER_3_10_1 register kernels, the other zoom rates)
         'L1F2_10Targets_wide', 'S1L1F1_10Targets_wide', 'L1F2_3Targets_090observe', 'L1F2_3Targets_095observe',
         'L1F2_3Targets_105observe', 'S1L2F2_3Targets_075observe', 'S1L2F2_3Targets_090observe',
         'S1L2F2_3Targets_095observe', 'S1L2F2_3Targets_105observe']
# the two *_MovingTarget files (kind MPE_HIER_MOVING): a moving prey instead of landmarks
MOVING = ['L1F4_MovingTarget', 'S1L2F4_MovingTarget']
HIER_CASES = (['h_' + b + suf for b in BASES for suf in ('', '_dense')]
              + ['m_' + b + suf for b in MOVING for suf in ('', '_dense', '_crowd')])


def _spec_of(golden_dir, base):
    return O.hier_spec_from_golden(_load(golden_dir, ('m_' if base in MOVING else 'h_') + base))


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


@pytest.fixture(params=[0, 1, 2], ids=['auto_kernel', 'tile_kernel', 'wide_kernel'])
def force_tile(request):
    """The stock shapes run a register kernel by default: at these batch sizes (<= MPE_HIER_SPLIT_MAX_ENVS) the
    one-agent-per-wavefront form (mpe_hsplit_kernel.h), beyond them the one-env-per-lane form (mpe_hier_kernel.h), which
    MPE_FLAG_WIDE selects here; the other hook routes them through the generic LDS-tile kernel."""
    set_call_flags({0: 0, 1: 1, 2: 2}[request.param])
    yield request.param
    set_call_flags(0)


def _maxerr(got, want):
    return float(np.max(np.abs(got - want))) if got.size else 0.0


@pytest.mark.parametrize('name', HIER_CASES)
def test_f64_trajectory_matches_reference(golden_dir, name, force_tile):
    g = _load(golden_dir, name)
    spec = O.hier_spec_from_golden(g)
    S, T = g['act'].shape[:2]
    sim = HipSim(spec, S, torch.float64)
    sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), lforce=golden_soa(g, 'lforce0'))
    sim.observe(rew=False, info=False)
    assert _maxerr(sim.host()[2], g['obs0'].T) == 0.0
    worst = 0.0
    for t in range(T):
        act = dev(g['act'][:, t].T, torch.float64)                   # [act_total, S]
        sim.env_step(act)
        pos, vel, obs, rew, info = sim.host()
        lf = sim.lforce.cpu().numpy()
        errs = (_maxerr(pos, golden_soa(g, 'pos', t)), _maxerr(vel, golden_soa(g, 'vel', t)),
                _maxerr(lf, golden_soa(g, 'lforce', t)), _maxerr(obs, g['obs'][:, t].T),
                _maxerr(rew, golden_soa(g, 'rew', t)), _maxerr(info, golden_soa(g, 'info', t)))
        worst = max(worst, *errs)
        assert max(errs) <= F64_TOL, (name, t, errs)
    print('%s worst |err| = %.3e' % (name, worst))


def _near_threshold(spec, P, eps=1e-5):
    """Envs whose float64 state sits within eps of a discrete decision of the gathers."""
    A, L = spec.n_agents, spec.n_landmarks
    size = np.asarray(spec.size)
    dist = np.sqrt(((P[:A, None] - P[None, :]) ** 2).sum(2))            # [A,E,S]
    near = np.zeros(P.shape[-1], bool)
    rad = np.asarray(spec.obs_r)[:, None, None]
    off = ~np.eye(A, A + L, dtype=bool)[..., None]
    near |= ((np.abs(dist - rad) < eps) & off).any((0, 1))              # visibility radius
    dmin = (size[:A, None] + size[None, :])[..., None]
    near |= ((np.abs(dist - dmin) < eps) & off).any((0, 1))             # contacts
    ax = np.abs(P[:A])
    near |= (np.abs(ax - 0.9) < eps).any((0, 1)) | (np.abs(ax - 1.0) < eps).any((0, 1))
    dl = np.sort(dist[:, A:], axis=1)                                   # target ordering ties
    if L > 1:
        near |= (np.diff(dl, axis=1) < eps).any((0, 1))
    return near


@pytest.mark.parametrize('name', HIER_CASES)
def test_f32_teacher_forced_matches_reference(golden_dir, name, force_tile):
    g = _load(golden_dir, name)
    spec = O.hier_spec_from_golden(g)
    S, T = g['act'].shape[:2]
    sim = HipSim(spec, S, torch.float32)
    kept = 0.0
    for t in range(T):
        p0 = golden_soa(g, 'pos0') if t == 0 else golden_soa(g, 'pos', t - 1)
        v0 = golden_soa(g, 'vel0') if t == 0 else golden_soa(g, 'vel', t - 1)
        sim.set_state(p0, v0)
        sim.env_step(dev(g['act'][:, t].T, torch.float32))
        pos, vel, obs, rew, info = sim.host()
        lf = sim.lforce.double().cpu().numpy()
        np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(vel, golden_soa(g, 'vel', t), rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(lf, golden_soa(g, 'lforce', t), rtol=1e-4, atol=1e-4)
        ok = ~_near_threshold(spec, golden_soa(g, 'pos', t))
        np.testing.assert_allclose(obs[:, ok], g['obs'][:, t].T[:, ok], rtol=1e-4, atol=1e-4)
        np.testing.assert_allclose(rew[:, ok], golden_soa(g, 'rew', t)[:, ok], rtol=1e-4, atol=2e-4)
        np.testing.assert_array_equal(info[..., ok], golden_soa(g, 'info', t)[..., ok])
        kept += ok.mean()
    assert kept / T > 0.9


def _custom_spec():
    """A world the presets do not build: 3 commands (core.py:169-171 mapping), two leaders commanding
    the same followers, a silent leader (its c stays zero), a good agent and a leader under the
    super-leader, non-colliding entities, mixed masses / speed limits."""
    A, L = 7, 5
    return O.ScenarioSpec(
        kind=O.HIER_PURSUIT, n_agents=A, n_landmarks=L, adversary=[True, True, False, True, True, True, True],
        size=[0.075, 0.06, 0.05, 0.075, 0.075, 0.08, 0.07] + [0.05, 0.07, 0.05, 0.1, 0.05],
        accel=[3.0, 3.5, 4.0, 3.0, None, 3.0, 3.0], max_speed=[1.0, None, 1.3, 1.0, 1.0, 0.8, 1.0],
        collide=[True, True, False, True, True, True, True] + [True, True, False, True, True],
        movable=[True] * A + [False] * L, u_noise=[None] * A,
        mass=[1.0, 2.0, 1.0, 1.0, 0.5, 1.0, 1.0] + [1.0] * L,
        agent_pos_range=1.0, landmark_pos_range=1.0, dim_c=3,
        silent=[True, True, True, False, False, False, True],
        force_id=[-1, -1, -1, 1, 1, 3, 1], forced_id=[1, 1, 3, 3, -1, -1, -1],
        advleader=[False, False, False, True, True, True, True],
        advsuperleader=[False, False, False, False, False, True, False],
        obs_r=[0.3, 0.25, 0.5, 0.6, 0.4, 0.7, 0.2], reward_variant=0, n_target_obs=3)


def _rand_rollout(spec, B, dtype, steps, seed, reward_variant=None):
    if reward_variant is not None:
        spec = dataclasses.replace(spec, reward_variant=reward_variant)
    rng = np.random.RandomState(seed)
    E, A = spec.n_entities, spec.n_agents
    pos = rng.uniform(-1.1, 1.1, (E, 2, B))
    pos[:, :, : B // 2] *= 0.25                                          # dense half: contacts, visible targets
    pos[:, :, : B // 16] *= 0.1                                          # a crowd: the count >= 4 bonus of MovingTarget
    vel = rng.uniform(-0.5, 0.5, (A, 2, B))
    lf0 = rng.uniform(-1, 1, (A, 2, B))
    acts = rng.rand(steps, sum(spec.act_dims), B)
    return spec, pos, vel, lf0, acts


@pytest.mark.parametrize('variant', [0, 1])
@pytest.mark.parametrize('which', ['custom', 'S1L2F2_10Targets_wide', 'L1F4_3Targets_wide'] + MOVING)
def test_f64_random_batch_matches_oracle(golden_dir, which, variant, force_tile):
    """20 011 envs (ragged last tile) against the batched CPU oracle: every branch of the command
    decode and of both reward texts."""
    base = _custom_spec() if which == 'custom' else _spec_of(golden_dir, which)
    B, steps = 20011, 4
    spec, pos, vel, lf0, acts = _rand_rollout(base, B, torch.float64, steps, 7, variant)
    sim = HipSim(spec, B, torch.float64)
    sim.set_state(pos, vel, lforce=lf0)
    bo = O.BatchedOracle(spec)
    sim.observe()
    obs, rew, info = bo.observe(pos, vel, lforce=lf0)
    got = sim.host()
    assert _maxerr(got[2], np.concatenate(obs, 0)) == 0.0 and _maxerr(got[3], rew) <= F64_TOL
    for t in range(steps):
        sim.env_step(dev(acts[t], torch.float64))
        pos, vel, obs, rew, info, lf = bo.env_step(pos, vel, acts[t], return_lforce=True)
        gp, gv, go, gr, gi = sim.host()
        errs = (_maxerr(gp, pos), _maxerr(gv, vel), _maxerr(sim.lforce.cpu().numpy(), lf),
                _maxerr(go, np.concatenate(obs, 0)), _maxerr(gr, rew), _maxerr(gi, info))
        assert max(errs) <= F64_TOL, (which, variant, t, errs)
    assert (rew >= 1).any() and (info > 0).any() if which == 'custom' else (rew >= 1).any()


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_split_entry_points_equal_fused(golden_dir, dtype, force_tile):
    """mpe_world_step + mpe_observe == mpe_env_step bit for bit (ld > B); comm = the leaders' own
    command entries, zero for silent agents (core.py:226-230); padding untouched."""
    spec = O.hier_spec_from_golden(_load(golden_dir, 'h_S1L2F2_3Targets_100observe'))
    B, ld = 1000 + 37, 1200
    _, pos, vel, lf0, acts = _rand_rollout(spec, B, dtype, 3, 3)
    a, b = HipSim(spec, B, dtype, ld), HipSim(spec, B, dtype, ld)
    a.set_state(pos, vel, lforce=lf0)
    b.set_state(pos, vel, lforce=lf0)
    comm = torch.full((spec.n_agents, spec.dim_c, ld), 7.0, dtype=dtype, device='cuda')
    rows = np.concatenate([[0], np.cumsum(spec.act_dims)])
    for t in range(3):
        act = torch.zeros(sum(spec.act_dims), ld, dtype=dtype, device='cuda')
        act[:, :B] = dev(acts[t], dtype)
        a.env_step(act)
        b.world_step(act, comm=comm)
        b.observe()
    for x, y in zip(a.host() + (a.lforce,), b.host() + (b.lforce,)):
        x, y = (t.cpu().numpy() if torch.is_tensor(t) else t for t in (x, y))
        assert np.array_equal(x, y, equal_nan=True)
    for i in range(spec.n_agents):
        want = 0.0 if spec.silent[i] else act[rows[i] + 5:rows[i] + 5 + spec.dim_c, :B]
        assert (comm[i, :, :B] == want).all()
    assert (comm[:, :, B:] == 7).all()
    for t in (a.obs, a.rew, a.info):
        assert torch.isnan(t[..., B:]).all()
    assert (a.pos[..., B:] == 0).all() and (a.vel[..., B:] == 0).all() and (a.lforce[..., B:] == 0).all()


@pytest.mark.parametrize('base', BASES + MOVING)
def test_register_and_tile_kernels_bit_identical(golden_dir, base):
    """Both kernels implement the same operation order: every output equal bit for bit, f64 and f32,
    on states that exercise contacts, visibility and all seven commands."""
    spec = _spec_of(golden_dir, base)
    B = 4096 + 11
    for dtype in (torch.float64, torch.float32):
        _, pos, vel, lf0, acts = _rand_rollout(spec, B, dtype, 4, 21)
        res = []
        for ft in (2, 1):                                                # MPE_FLAG_WIDE, MPE_FLAG_FORCE_TILE
            set_call_flags(ft)
            sim = HipSim(spec, B, dtype)
            sim.set_state(pos, vel, lforce=lf0)
            comm = torch.zeros(spec.n_agents, spec.dim_c, B, dtype=dtype, device='cuda')
            sim.observe()
            out = [t.copy() for t in sim.host()]
            for t in range(4):
                sim.env_step(dev(acts[t], dtype), comm=comm)
            res.append(out + list(sim.host()) + [sim.lforce.cpu().numpy(), comm.cpu().numpy()])
        set_force_tile(0)
        for x, y in zip(*res):
            assert np.array_equal(x, y, equal_nan=True)


# one file per register-kernel shape (mpe_launch.h: HIER_* / MOVING_*) and both reward texts
SHAPES = ['L1F2_3Targets_100observe', 'L1F4_3Targets_wide', 'L1F1_10Targets_wide', 'L1F2_10Targets_wide',
          'S1L1F1_10Targets_wide', 'S1L2F2_3Targets_100observe', 'S1L2F2_10Targets_wide',
          'S1L2F2_10Targets_wide_rewardFixed'] + MOVING
WIDE, SPLIT = 2, 4


def test_default_form_by_batch_size(golden_dir):
    """B <= MPE_HIER_SPLIT_MAX_ENVS runs one agent per wavefront (statistics rows per 64 envs), larger batches one env
    per lane (per 256); the two flags override; the tile kernel ignores both."""
    from marl_amd import _hip
    n = _hip.HIER_SPLIT_MAX_ENVS
    for base in ('S1L2F2_10Targets_wide', 'S1L2F4_MovingTarget', 'L1F2_3Targets_100observe'):
        d = HipSim(_spec_of(golden_dir, base), 8, torch.float32).desc
        for nbytes in (4, 8):
            rows = lambda B, f: lib.mpe_stats_rows(d, B, nbytes, f)
            assert rows(n, 0) == n // 64 and rows(n + 1, 0) == -(-(n + 1) // 256)
            assert rows(n, WIDE) == n // 256 and rows(4 * n, SPLIT) == 4 * n // 64
            assert rows(1000, 1) == rows(1000, 1 | WIDE) == rows(1000, 1 | SPLIT)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('base', SHAPES)
def test_agent_per_wavefront_form_equals_env_per_lane_form(golden_dir, base, dtype):
    """mpe_hsplit_kernel.h (one agent per wavefront, MPE_FLAG_SPLIT; the default for the f64 runs of the two largest
    worlds) against mpe_hier_kernel.h (MPE_FLAG_WIDE): the same bits for state, leader_force, comm, observations,
    rewards and info from every entry point, on a ragged batch inside a padded allocation; the statistics rows group
    64 * EW envs instead of 256 (integers equal, reward sums to the last bits)."""
    spec = _spec_of(golden_dir, base)
    A = spec.n_agents
    B, ld = 128 * 11 + 77, 1664
    _, pos, vel, lf0, acts = _rand_rollout(spec, B, dtype, 4, 31)
    pos[0, 0, 100:160] = 1.07                                            # bound() penalties
    pos[A - 1, 1, 140:200] = -1.4
    acts = [torch.zeros(sum(spec.act_dims), ld, dtype=dtype, device='cuda') for _ in range(4)]
    rng = np.random.RandomState(9)
    for a in acts:
        a[:, :B] = dev(rng.rand(sum(spec.act_dims), B), dtype)
    W = 2 * A + 2
    out = {}
    try:
        for flags in (WIDE, SPLIT):
            set_call_flags(flags)
            fused, parts = HipSim(spec, B, dtype, ld), HipSim(spec, B, dtype, ld)
            rows = stats_rows(fused.desc, B, fused.pos.element_size())
            assert rows == -(-B // (256 if flags == WIDE else 64))
            stats = torch.zeros(rows, W, dtype=torch.float64, device='cuda')
            comm = torch.full((A, spec.dim_c, ld), 7.0, dtype=dtype, device='cuda')
            for s in (fused, parts):
                s.set_state(pos, vel, lforce=lf0)
            fused.observe()
            first = [t.copy() for t in fused.host()]
            for a in acts:
                fused.env_step(a, stats=stats, comm=comm)
                parts.world_step(a)
                parts.observe()
            names = ('pos', 'vel', 'obs', 'rew', 'info', 'lforce')
            for nm, x, y in zip(names, fused.host() + (fused.lforce.cpu().numpy(),),
                                parts.host() + (parts.lforce.cpu().numpy(),)):
                bad = ~((x == y) | (np.isnan(x) & np.isnan(y)))
                assert not bad.any(), (flags, nm, np.unique(np.nonzero(bad)[0])[:20], np.nonzero(bad)[-1][:10])
            assert (comm[:, :, B:] == 7).all()
            for t in (fused.obs, fused.rew, fused.info):
                assert torch.isnan(t[..., B:]).all()                      # padding never written
            assert (fused.pos[..., B:] == 0).all() and (fused.lforce[..., B:] == 0).all()
            red = torch.empty(W, dtype=torch.float64, device='cuda')
            assert lib.mpe_stats_reduce(stats.data_ptr(), rows, W, red.data_ptr(), fused.stream) == 0
            out[flags] = (first + list(fused.host()) + [fused.lforce.cpu().numpy(), comm.cpu().numpy()],
                          red.cpu().numpy())
    finally:
        set_call_flags(0)
    for x, y in zip(out[WIDE][0], out[SPLIT][0]):
        assert np.array_equal(x, y, equal_nan=True)
    sw, ss = out[WIDE][1], out[SPLIT][1]
    assert np.array_equal(sw[A:], ss[A:])                                # contact counters, env count: integers
    np.testing.assert_allclose(ss[:A], sw[:A], rtol=1e-12)
    assert sw[2 * A + 1] == 4 * B and sw[A:2 * A + 1].sum() > 0           # target hits / prey contacts happened


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('base', ['S1L2F2_10Targets_wide', 'S1L2F4_MovingTarget', 'L1F2_3Targets_100observe'])
def test_residency_variants_bit_identical(golden_dir, base, dtype):
    """The fused one-env-per-lane kernels have one variant per Infinity Cache plan (mpe_residency_plan: which state rows are
    read and written with nontemporal accesses).  The plan follows the row stride, so a small batch inside wide rows runs
    the variants of multi-million-env launches: every output equal bit for bit to the run at ld = B."""
    spec = _spec_of(golden_dir, base)
    A, B = spec.n_agents, 777
    nbytes = 8 if dtype == torch.float64 else 4
    _, pos, vel, lf0, acts = _rand_rollout(spec, B, dtype, 3, 41)
    res, plans = [], []
    try:
        set_call_flags(WIDE)
        for ld in [B] + [int(k * (1 if dtype == torch.float64 else 2) * 1048576) + 128 for k in (0.5, 0.65, 1, 2)]:
            sim = HipSim(spec, B, dtype, ld)
            plans.append(lib.mpe_residency_plan(sim.desc, ld, nbytes))
            sim.set_state(pos, vel, lforce=lf0)
            for t in range(3):
                act = torch.zeros(sum(spec.act_dims), ld, dtype=dtype, device='cuda')
                act[:, :B] = dev(acts[t], dtype)
                sim.env_step(act)
            res.append(list(sim.host()) + [sim.lforce[..., :B].cpu().numpy()])
            assert torch.isnan(sim.obs[..., B:B + 64]).all() and (sim.pos[..., B:B + 64] == 0).all()
            del sim, act
            torch.cuda.empty_cache()
    finally:
        set_call_flags(0)
    assert plans[0] == 0 and plans[-1] != 0 and (A < 5 or plans[-1] & 2)     # cached ... velocities streamed too (5 agents and more)
    assert A < 5 or len(set(plans)) >= 3                                     # S1L2F2_10Targets: 0, 8, 12, 13, 15 = all five variants
    if 'S1L2F2_10Targets' in base:
        assert plans == [0, 8, 12, 13, 15]
    for other in res[1:]:
        for x, y in zip(res[0], other):
            assert np.array_equal(x, y, equal_nan=True)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('base', ['S1L2F2_10Targets_wide', 'S1L2F2_3Targets_100observe', 'L1F2_3Targets_075observe',
                                  'L1F1_10Targets_wide'])
def test_equidistant_targets_keep_their_index_order(golden_dir, base, dtype, force_tile):
    """The visible targets are listed nearest first by a STABLE sort (L1F2_3Targets_100observe.py:242-245): targets at
    exactly the same squared distance stay in index order.  Entities on a grid of eighths: every distance is exact in
    f32 and ties are everywhere, also behind a nearer target that arrives later (the case the sorted-list insertion
    of the register kernels got wrong until the f32 kernel-form fuzz met one).  Observations and counters exact."""
    spec = _spec_of(golden_dir, base)
    spec = dataclasses.replace(spec, obs_r=[0.75, 0.5, 0.625, 0.75, 1.0][:spec.n_agents])   # (exact squares; wide enough to see a few targets)
    A, E, B = spec.n_agents, spec.n_entities, 4096 + 37
    rng = np.random.RandomState(77)
    pos = rng.randint(-5, 6, (E, 2, B)) / 8.0
    pos[A:, :, : B // 2] = rng.randint(-2, 3, (E - A, 2, B // 2)) / 8.0    # a tight cluster of targets: many equal distances
    vel = rng.randint(-4, 5, (A, 2, B)) / 8.0
    lf0 = rng.randint(-8, 9, (A, 2, B)) / 8.0
    npdt = np.float64 if dtype == torch.float64 else np.float32
    obs, _, info = O.BatchedOracle(spec, npdt).observe(pos, vel, lforce=lf0)
    d2 = ((pos[A:, None] - pos[None, :A]) ** 2).sum(2)                      # [L, A, B]
    ds = np.sort(np.where(d2 < (np.asarray(spec.obs_r) ** 2)[None, :, None], d2, np.inf), 0)[:spec.n_target_obs + 1]
    ties = ((ds[1:] == ds[:-1]) & np.isfinite(ds[1:])).any(0)             # [A, B]: equal distances among the listed targets
    assert ties.sum() > B // 4 and ties.any(1).all()
    sim = HipSim(spec, B, dtype)
    sim.set_state(pos, vel, lforce=lf0)
    sim.observe()
    got = sim.host()
    assert np.array_equal(got[2], np.concatenate(obs, 0).astype(npdt))
    assert np.array_equal(got[4], np.asarray(info).reshape(got[4].shape))


@pytest.mark.parametrize('base', MOVING)
def test_moving_world_in_which_nobody_is_commanded(golden_dir, base, force_tile):
    """forced_id = -1 everywhere: no agent has a leader_force (the world keeps no such rows at all), yet the observations
    of the *_MovingTarget files still end with it for the agents whose text says so: zeros (core.py:154).  Found by the
    stock-shape fuzz: the tile kernel wrote velocity rows there."""
    spec = _spec_of(golden_dir, base)
    spec = dataclasses.replace(spec, forced_id=[-1] * spec.n_agents)
    assert not spec.has_leader_force
    B, steps = 1500 + 7, 3
    _, pos, vel, _, acts = _rand_rollout(spec, B, torch.float64, steps, 5)
    sim = HipSim(spec, B, torch.float64)
    sim.set_state(pos, vel)
    bo = O.BatchedOracle(spec)
    for t in range(steps):
        sim.env_step(dev(acts[t], torch.float64))
        pos, vel, obs, rew, info = bo.env_step(pos, vel, acts[t])[:5]
        gp, gv, go, gr, gi = sim.host()
        errs = (_maxerr(gp, pos), _maxerr(gv, vel), _maxerr(go, np.concatenate(obs, 0)), _maxerr(gr, rew), _maxerr(gi, info))
        assert max(errs) <= F64_TOL, (base, t, errs)


@pytest.mark.parametrize('base', ['S1L2F2_10Targets_wide', 'S1L2F4_MovingTarget'])
def test_stats_rows_accumulate_rewards_and_contacts(golden_dir, base, force_tile):
    spec = _spec_of(golden_dir, base)
    B, steps = 5000 + 3, 3
    _, pos, vel, lf0, acts = _rand_rollout(spec, B, torch.float64, steps, 11)
    from marl_amd import _hip
    sim = HipSim(spec, B, torch.float64)
    sim.set_state(pos, vel, lforce=lf0)
    A = spec.n_agents
    rows, width = stats_rows(sim.desc, B, 8), lib.mpe_stats_width(sim.desc)
    stats = torch.zeros(rows, width, dtype=torch.float64, device='cuda')
    want = np.zeros(width)
    bo = O.BatchedOracle(spec)
    for t in range(steps):
        sim.env_step(dev(acts[t], torch.float64), stats=stats)
        pos, vel, obs, rew, info, lf = bo.env_step(pos, vel, acts[t], return_lforce=True)
        want[:A] += rew.sum(1)
        want[A:2 * A] += info[:, 0].sum(1)
        want[2 * A + 1] += B
    out = torch.empty(width, dtype=torch.float64, device='cuda')
    assert lib.mpe_stats_reduce(stats.data_ptr(), rows, width, out.data_ptr(), sim.stream) == 0
    got = out.cpu().numpy()
    np.testing.assert_allclose(got[:2 * A], want[:2 * A], rtol=1e-9, atol=1e-6)
    assert got[2 * A + 1] == want[2 * A + 1] and got[2 * A] >= 0
    if base in MOVING:                                                   # [2A]: (prey, adversary) contacts
        assert got[2 * A] == want[2 * A - 1] > 0
    assert _hip.lib.mpe_info_width(sim.desc) == 1


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_target_move_matches_oracle(golden_dir, dtype):
    """SURVEY 8f row f3 (experiments/train_and_eval.py:207-222, parity unpinned: see the oracle's
    header): the kernel respawns exactly the landmarks the oracle does, at the oracle's Philox
    positions (bit for bit: decisions are evaluated in the kernel's own dtype), and leaves every
    other entity, the padding and the velocities alone; the draw does not depend on sharding."""
    spec = O.hier_spec_from_golden(_load(golden_dir, 'h_S1L2F2_10Targets_wide'))
    B, ld = 3000 + 5, 3200
    npdt = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.RandomState(4)
    pos = (rng.uniform(-1, 1, (spec.n_entities, 2, B)) * np.where(np.arange(B) % 2, 0.15, 1.0)).astype(npdt)
    sim = HipSim(spec, B, dtype, ld)
    sim.set_state(pos, np.zeros((spec.n_agents, 2, B)))
    count = torch.full((ld,), -7, dtype=torch.int32, device='cuda')
    fn = getattr(lib, 'mpe_target_move_' + sim.suf)
    assert fn(sim.desc, B, ld, 99, 3, 1000, sim.buffers(), count.data_ptr(), sim.stream) == 0
    want_pos, want_cnt = O.target_move(spec, pos, seed=99, counter=3, env_ids=np.arange(B, dtype=np.uint64) + 1000,
                                       dtype=npdt)
    got = sim.pos[..., :B].cpu().numpy()
    assert np.array_equal(got, want_pos)
    assert np.array_equal(count[:B].cpu().numpy(), want_cnt) and (count[B:] == -7).all()
    assert want_cnt.sum() > 100 and (want_cnt[::2] == 0).mean() > 0.5
    assert (sim.pos[..., B:] == 0).all() and (sim.vel == 0).all()
    # second shard geometry: envs [1500, B) alone give the same rows
    sim2 = HipSim(spec, B - 1500, dtype)
    sim2.set_state(pos[..., 1500:], np.zeros((spec.n_agents, 2, B - 1500)))
    assert fn(sim2.desc, B - 1500, B - 1500, 99, 3, 2500, sim2.buffers(), None, sim2.stream) == 0
    assert np.array_equal(sim2.pos.cpu().numpy(), want_pos[..., 1500:])
    assert fn(sim.desc, B, ld, 99, 3, 1000, None, None, sim.stream) == -1
