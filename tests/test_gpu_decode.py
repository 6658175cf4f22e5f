"""GPU tests of the two non-default decodes of MultiAgentEnv._set_action (multiagent/environment.py:176-187):
env.discrete_action_input (one integer per agent) and env.force_discrete_action (argmax -> one-hot), as the passes
mpe_action_from_index_* / mpe_action_force_discrete_* over the SoA action rows in front of the step kernels.
Checked through the C ABI against fixtures generated from the reference (tests/golden/gen_golden.py discrete_cases),
against the oracle on random batches, and through the public env API."""
import os

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from marl_amd import _hip
    from tests.gpu_util import HipSim, dev, golden_soa, lib

CASES = [('g15_simple_tag_index_actions', O.simple_tag_spec, 'index'),
         ('g15_simple_spread_index_actions', O.simple_spread_spec, 'index'),
         ('g16_simple_tag_force_discrete', O.simple_tag_spec, 'force'),
         ('g16_simple_push_force_discrete', O.simple_push_spec, 'force')]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


def _decode(sim, mode, act, idx=None):
    """Run the decode pass on the [A*5, ld] action rows `act` (index mode: from int32 idx [A, ld])."""
    if mode == 'index':
        rc = getattr(lib, 'mpe_action_from_index_' + sim.suf)(sim.desc, sim.B, sim.ld, idx.shape[1], idx.data_ptr(),
                                                             act.data_ptr(), sim.stream)
    else:
        rc = getattr(lib, 'mpe_action_force_discrete_' + sim.suf)(sim.desc, sim.B, sim.ld, act.data_ptr(), sim.stream)
    assert rc == 0, lib.mpe_strerror(rc)


@pytest.mark.parametrize('name,mk,mode', CASES)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_decode_passes_match_reference_trajectories(golden_dir, name, mk, mode, dtype, tol):
    """f64: free-running 25-step episodes within 1e-9 of the reference; f32: teacher-forced per step, 1e-4."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    A = spec.n_agents
    sim = HipSim(spec, S, dtype)
    sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), g['goal'])
    for t in range(T):
        if dtype == torch.float32 and t:
            sim.set_state(golden_soa(g, 'pos', t - 1), golden_soa(g, 'vel', t - 1), g['goal'])
        a = golden_soa(g, 'act', t)                                   # [A, 5, S]: what env.step received
        if mode == 'index':
            idx = torch.as_tensor(np.ascontiguousarray(a[:, 0]), dtype=torch.int32).cuda()
            act = torch.full((A * 5, S), float('nan'), dtype=dtype, device='cuda')    # every row must be written
            _decode(sim, mode, act, idx)
            # the rows written are the 5-vector whose default decode gives the integer's force
            u0 = (act[1::5] - act[2::5]).cpu().numpy()
            k = a[:, 0]
            assert np.array_equal(u0, np.where(k == 1, -1.0, np.where(k == 2, 1.0, 0.0)))
        else:
            act = dev(a.reshape(A * 5, S), dtype)
            _decode(sim, mode, act)
            want = (np.arange(5)[None, :, None] == np.argmax(a, axis=1)[:, None, :]).astype(np.float64)
            assert np.array_equal(act.double().cpu().numpy().reshape(A, 5, S), want)      # first maximum on ties
        sim.env_step(act.view(A, 5, S))
        pos, vel, obs, rew, info = sim.host()
        np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(vel, golden_soa(g, 'vel', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(obs, g['obs'][:, t].T, rtol=tol, atol=tol)
        if dtype == torch.float64:
            np.testing.assert_allclose(rew, golden_soa(g, 'rew', t), rtol=0, atol=tol)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_decode_passes_on_a_padded_batch_with_nan_and_ties(dtype):
    """20 011 envs in rows of stride 20 096: np.argmax's rules (first maximum, a NaN wins, the first NaN among
    several), -inf rows, integers outside 0..4 and negative ones; the padding columns stay untouched."""
    B, ld = 20011, 20096
    spec = O.simple_tag_spec()
    A = spec.n_agents
    sim = HipSim(spec, B, dtype, ld=ld)
    rng = np.random.RandomState(5)
    a = rng.randint(0, 3, size=(A, 5, ld)) / 2.0
    a[:, :, 7::11] = np.where(rng.rand(A, 5, len(range(7, ld, 11))) < 0.3, np.nan, a[:, :, 7::11])
    a[:, :, 3::13] = -np.inf
    a[0, 2, 5::17] = np.inf
    act = dev(a.reshape(A * 5, ld), dtype)
    _decode(sim, 'force', act)
    got = act.double().cpu().numpy().reshape(A, 5, ld)
    want = (np.arange(5)[None, :, None] == np.argmax(a, axis=1)[:, None, :]).astype(np.float64)
    assert np.array_equal(got[:, :, :B], want[:, :, :B])
    pad = dict(nan=7.0, neginf=-9.0, posinf=9.0)
    assert np.array_equal(np.nan_to_num(got[:, :, B:], **pad), np.nan_to_num(a[:, :, B:], **pad))

    k = rng.randint(-3, 9, size=(A, ld)).astype(np.int32)
    idx = torch.as_tensor(k).cuda()
    act = torch.full((A * 5, ld), 3.0, dtype=dtype, device='cuda')
    _decode(sim, 'index', act, idx)
    got = act.double().cpu().numpy().reshape(A, 5, ld)
    bo = O.BatchedOracle(spec, discrete_action_input=True)
    for i in range(A):
        want = bo.physical_action(np.broadcast_to(k[i].astype(np.float64), (5, ld)))
        assert np.array_equal(got[i, :, :B], want[:, :B]), i
    assert (got[:, :, B:] == 3.0).all()


def test_force_discrete_leaves_comm_entries_and_immovable_agents_alone():
    """environment.py:174,184-187: only a movable agent's first 5 entries are rewritten.  simple_speaker_listener:
    rows 0-2 are the speaker's comm entries (not movable: no physical part), rows 3-7 the listener's physical ones;
    S1L2F2 leaders: 5 physical rows then 7 comm rows."""
    B = 1000
    for spec in (O.speaker_listener_spec(),):
        sim = HipSim(spec, B, torch.float64)
        R = sum(spec.act_dims)
        act = torch.rand(R, B, dtype=torch.float64, device='cuda')
        before = act.clone()
        _decode(sim, 'force', act)
        offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
        for i in range(spec.n_agents):
            rows = slice(int(offs[i]), int(offs[i + 1]))
            if spec.movable[i]:
                phys = act[rows][:5]
                want = torch.nn.functional.one_hot(before[rows][:5].argmax(0), 5).t().double()
                assert torch.equal(phys, want)
                assert torch.equal(act[rows][5:], before[rows][5:])
            else:
                assert torch.equal(act[rows], before[rows])


def test_decode_passes_refuse_bad_arguments():
    spec = O.simple_tag_spec()
    sim = HipSim(spec, 64, torch.float64)
    act = torch.zeros(20, 64, dtype=torch.float64, device='cuda')
    idx = torch.zeros(4, 64, dtype=torch.int32, device='cuda')
    f = lib.mpe_action_from_index_f64
    assert f(sim.desc, 64, 64, 64, None, act.data_ptr(), sim.stream) == -1
    assert f(sim.desc, 64, 64, 64, idx.data_ptr(), None, sim.stream) == -1
    assert f(sim.desc, 64, 64, 32, idx.data_ptr(), act.data_ptr(), sim.stream) == -3
    assert f(sim.desc, 64, 32, 64, idx.data_ptr(), act.data_ptr(), sim.stream) != 0
    assert lib.mpe_action_force_discrete_f64(sim.desc, 64, 64, None, sim.stream) == -1
    # a world with a speaking agent: the reference's integer decode fails its own assert (environment.py:208)
    ssim = HipSim(O.simple_world_comm_spec(), 64, torch.float64)
    big = torch.zeros(ssim.spec.n_agents * 9, 64, dtype=torch.float64, device='cuda')
    idx6 = torch.zeros(6, 64, dtype=torch.int32, device='cuda')
    assert f(ssim.desc, 64, 64, 64, idx6.data_ptr(), big.data_ptr(), ssim.stream) == -5
    torch.cuda.synchronize()


@pytest.mark.parametrize('name,mk,mode', CASES)
def test_env_api_discrete_modes_match_reference(golden_dir, name, mk, mode):
    """The public switches, set the way the reference's users set them (env.discrete_action_input = True after
    make_env; world.discrete_action = True -> env.force_discrete_action, environment.py:33): 25 steps from the
    fixture's initial state, f64 within 1e-9 of the reference."""
    from marl_amd import make_env
    g = _load(golden_dir, name)
    S, T = g['act'].shape[:2]
    env = make_env(name.split('_', 1)[1].rsplit('_', 2)[0], benchmark=False, num_envs=S, dtype=torch.float64, seed=1)
    env.reset()
    w = env.world
    w.pos.copy_(dev(golden_soa(g, 'pos0'), torch.float64))
    w.vel.copy_(dev(golden_soa(g, 'vel0'), torch.float64))
    if w.goal is not None and 'goal' in g.files and (g['goal'] >= 0).all():
        w.goal.view(-1)[:S].copy_(torch.as_tensor(g['goal'], dtype=torch.int32).cuda().view(-1))
    w.mark_dirty()
    if mode == 'index':
        env.discrete_action_input = True
    else:
        env.force_discrete_action = True
    for t in range(T):
        a = g['act'][:, t]                                            # [S, A, 5]
        if mode == 'index':
            forms = [[torch.as_tensor(a[:, i, 0], dtype=torch.int64).cuda() for i in range(env.n)],
                     torch.as_tensor(a[:, :, 0].T.copy(), dtype=torch.int32).cuda(),
                     [a[:, i, 0].astype(np.int64) for i in range(env.n)]]
            action_n = forms[t % 3]
        else:
            action_n = [dev(a[:, i], torch.float64) for i in range(env.n)]
        obs_n, rew_n, done_n, info_n = env.step(action_n)
        np.testing.assert_allclose(w.pos.cpu().numpy(), golden_soa(g, 'pos', t), rtol=0, atol=1e-9)
        np.testing.assert_allclose(torch.cat(obs_n, 1).cpu().numpy(), g['obs'][:, t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(torch.stack(rew_n, 1).cpu().numpy(), g['rew'][:, t], rtol=0, atol=1e-9)


def test_env_api_integer_actions_refused_for_speaking_agents():
    from marl_amd import make_env
    env = make_env('simple_reference', num_envs=8, dtype=torch.float64)
    env.reset()
    env.discrete_action_input = True
    with pytest.raises(AssertionError):                 # environment.py:208
        env.step([torch.zeros(8, dtype=torch.int64, device='cuda')] * 2)


@pytest.mark.parametrize('flags', [0, 1], ids=['register_kernel', 'tile_kernel'])
def test_force_discrete_on_multidiscrete_actions_matches_reference(golden_dir, flags):
    """S1L2F2_3Targets_100observe with env.force_discrete_action: the decode pass rewrites the 5 physical rows of every
    agent and leaves the leaders' 7 command rows alone; 25 free-running steps within 1e-9 of the reference (f64)."""
    from tests.gpu_util import set_call_flags
    g = _load(golden_dir, 'g16_h_S1L2F2_3Targets_100observe_force_discrete')
    spec = O.hier_spec_from_golden(g)
    S, T = g['act'].shape[:2]
    set_call_flags(flags)
    try:
        sim = HipSim(spec, S, torch.float64)
        sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'), lforce=golden_soa(g, 'lforce0'))
        offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
        for t in range(T):
            act = dev(np.ascontiguousarray(g['act'][:, t].T), torch.float64)
            before = act.clone()
            _decode(sim, 'force', act)
            for i in range(spec.n_agents):
                assert torch.equal(act[offs[i] + 5:offs[i + 1]], before[offs[i] + 5:offs[i + 1]])      # command rows
                assert ((act[offs[i]:offs[i] + 5] == 1).sum(0) == 1).all() and (act[offs[i]:offs[i] + 5].sum(0) == 1).all()
            sim.env_step(act)
            pos, vel, obs, rew, info = sim.host()
            np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=0, atol=1e-9)
            np.testing.assert_allclose(sim.lforce.cpu().numpy(), golden_soa(g, 'lforce', t), rtol=0, atol=1e-9)
            np.testing.assert_allclose(obs, g['obs'][:, t].T, rtol=0, atol=1e-9)
            np.testing.assert_allclose(rew, golden_soa(g, 'rew', t), rtol=0, atol=1e-9)
    finally:
        set_call_flags(0)


CONTINUOUS = [('g18_simple_tag_continuous', O.simple_tag_spec, False),
              ('g18_simple_spread_continuous', O.simple_spread_spec, False),
              ('g18_simple_tag_continuous_force', O.simple_tag_spec, True)]


@pytest.mark.parametrize('name,mk,forced', CONTINUOUS)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_continuous_action_space_matches_reference_trajectories(golden_dir, name, mk, forced, dtype, tol):
    """env.discrete_action_space = False (multiagent/environment.py:191-192): mpe_action_from_force_* writes the rows
    (0, u_x, 0, u_y, 0) whose default decode is u, bit for bit; f64 free-running within 1e-9 of the reference's
    trajectories, f32 teacher-forced within 1e-4."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    A = spec.n_agents
    sim = HipSim(spec, S, dtype)
    sim.set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'))
    for t in range(T):
        if dtype == torch.float32 and t:
            sim.set_state(golden_soa(g, 'pos', t - 1), golden_soa(g, 'vel', t - 1))
        a = golden_soa(g, 'act', t)                                   # [A, 5, S]; columns 0, 1 = u
        u = dev(np.ascontiguousarray(a[:, :2]), dtype)                # [A, 2, S]
        act = torch.full((A * 5, S), float('nan'), dtype=dtype, device='cuda')
        rc = getattr(lib, 'mpe_action_from_force_' + sim.suf)(sim.desc, S, S, S, u.data_ptr(), int(forced), act.data_ptr(),
                                                              sim.stream)
        assert rc == 0, lib.mpe_strerror(rc)
        rows = act.double().cpu().numpy().reshape(A, 5, S)
        want = a[:, :2].astype(np.float32 if dtype == torch.float32 else np.float64).astype(np.float64)
        if forced:
            second = np.argmax(a[:, :2], axis=1) == 1
            want = np.stack([np.where(second, 0.0, 1.0), np.where(second, 1.0, 0.0)], 1)
        assert np.array_equal(rows[:, 1], want[:, 0]) and np.array_equal(rows[:, 3], want[:, 1])
        assert not rows[:, 0].any() and not rows[:, 2].any() and not rows[:, 4].any()
        sim.env_step(act.view(A, 5, S))
        pos, vel, obs, rew, info = sim.host()
        np.testing.assert_allclose(pos, golden_soa(g, 'pos', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(vel, golden_soa(g, 'vel', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(obs, g['obs'][:, t].T, rtol=tol, atol=tol)
        if dtype == torch.float64:
            np.testing.assert_allclose(rew, golden_soa(g, 'rew', t), rtol=0, atol=tol)


@pytest.mark.parametrize('name,mk,forced', CONTINUOUS)
def test_env_api_continuous_action_space_matches_reference(golden_dir, name, mk, forced):
    """The public switch, flipped after make_env like the reference's users must (the constructor hard-codes True,
    environment.py:29): list of [B, 2] tensors, one [B, n, 2] tensor, numpy arrays."""
    from marl_amd import make_env
    g = _load(golden_dir, name)
    S, T = g['act'].shape[:2]
    env = make_env('simple_tag' if 'tag' in name else 'simple_spread', benchmark=False, num_envs=S, dtype=torch.float64, seed=1)
    env.reset()
    w = env.world
    w.pos.copy_(dev(golden_soa(g, 'pos0'), torch.float64))
    w.vel.copy_(dev(golden_soa(g, 'vel0'), torch.float64))
    w.mark_dirty()
    env.discrete_action_space = False
    env.force_discrete_action = forced
    for t in range(T):
        u = g['act'][:, t, :, :2]                                      # [S, A, 2]
        forms = [[dev(u[:, i], torch.float64) for i in range(env.n)], dev(u, torch.float64), [u[:, i].copy() for i in range(env.n)]]
        passed = forms[t % 3]
        obs_n, rew_n, done_n, info_n = env.step(passed)
        if t % 3 == 0:                                                 # the caller's tensors are read, not scaled in place
            assert all(torch.equal(p, dev(u[:, i], torch.float64)) for i, p in enumerate(passed))
        np.testing.assert_allclose(w.pos.cpu().numpy(), golden_soa(g, 'pos', t), rtol=0, atol=1e-9)
        np.testing.assert_allclose(torch.cat(obs_n, 1).cpu().numpy(), g['obs'][:, t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(torch.stack(rew_n, 1).cpu().numpy(), g['rew'][:, t], rtol=0, atol=1e-9)


def test_continuous_action_space_refused_for_movable_speaking_agents_and_bad_arguments():
    from marl_amd import make_env
    env = make_env('simple_reference', num_envs=8, dtype=torch.float64)
    env.reset()
    env.discrete_action_space = False
    with pytest.raises(ValueError):                     # core.py:216 in the reference: 2-vector += 5-vector
        env.step([torch.zeros(8, 2, dtype=torch.float64, device='cuda')] * 2)
    ssim = HipSim(O.simple_world_comm_spec(), 64, torch.float64)
    u = torch.zeros(6, 2, 64, dtype=torch.float64, device='cuda')
    big = torch.zeros(6 * 9, 64, dtype=torch.float64, device='cuda')
    f = lib.mpe_action_from_force_f64
    assert f(ssim.desc, 64, 64, 64, u.data_ptr(), 0, big.data_ptr(), ssim.stream) == -5
    sim = HipSim(O.simple_tag_spec(), 64, torch.float64)
    act = torch.zeros(20, 64, dtype=torch.float64, device='cuda')
    assert f(sim.desc, 64, 64, 64, None, 0, act.data_ptr(), sim.stream) == -1
    assert f(sim.desc, 64, 64, 64, u.data_ptr(), 0, None, sim.stream) == -1
    assert f(sim.desc, 64, 64, 32, u.data_ptr(), 0, act.data_ptr(), sim.stream) == -3
    assert f(sim.desc, 64, 64, 64, u.data_ptr() + 4, 0, act.data_ptr(), sim.stream) == -4
    # a speaker that cannot move has no physical rows: nothing is written for it, nothing is refused
    sl = HipSim(O.speaker_listener_spec(), 64, torch.float64)
    rows = torch.full((8, 64), 7.0, dtype=torch.float64, device='cuda')
    u2 = torch.ones(2, 2, 64, dtype=torch.float64, device='cuda')
    assert f(sl.desc, 64, 64, 64, u2.data_ptr(), 0, rows.data_ptr(), sl.stream) == 0
    assert (rows[:3] == 7.0).all() and torch.equal(rows[3:, 0].cpu(), torch.tensor([0., 1., 0., 1., 0.], dtype=torch.float64))
    torch.cuda.synchronize()
