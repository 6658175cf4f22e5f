"""GPU tests of the drop-in Python boundary (make_env / MultiAgentEnv.step / reset / Scenario),
written like the loops that drive the reference (experiments/simple_train.py:257-300)."""
import os

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu


def _np(t):
    return t.double().cpu().numpy()


@pytest.mark.parametrize('name,spec', [('simple_tag', O.simple_tag_spec()), ('simple_spread', O.simple_spread_spec())])
def test_make_env_spaces_and_shapes(name, spec):
    """SURVEY KAT-3: n, Discrete(5) action spaces, observation dims."""
    from marl_amd import make_env
    env = make_env(name, benchmark=True, num_envs=7, dtype=torch.float64)
    assert env.n == spec.n_agents
    assert [s.n for s in env.action_space] == [5] * env.n
    assert [s.shape[0] for s in env.observation_space] == spec.obs_dims
    obs_n = env.reset()
    assert [tuple(o.shape) for o in obs_n] == [(7, d) for d in spec.obs_dims]
    obs_n, rew_n, done_n, info_n = env.step([torch.rand(7, 5, device='cuda', dtype=torch.float64) for _ in range(env.n)])
    assert [tuple(o.shape) for o in obs_n] == [(7, d) for d in spec.obs_dims]
    assert all(tuple(r.shape) == (7,) for r in rew_n)
    assert all(d.dtype == torch.bool and not d.any() for d in done_n)
    assert len(info_n['n']) == env.n
    env2 = make_env(name, benchmark=False, num_envs=3)
    env2.reset()
    assert env2.step(torch.rand(3, env2.n, 5, device='cuda', dtype=torch.float64))[3] == {'n': [{}] * env2.n}


@pytest.mark.parametrize('name,spec', [('simple_tag', O.simple_tag_spec()), ('simple_spread', O.simple_spread_spec())])
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)])
def test_episode_loop_matches_oracle(name, spec, dtype, tol):
    """reset() + 25 x step() driven through the reference's loop shape; oracle replays the same
    initial state and actions.  f32 is compared teacher-forced per step."""
    from marl_amd import make_env
    B, T = 512, 25
    env = make_env(name, benchmark=True, num_envs=B, dtype=dtype, seed=3)
    obs_n = env.reset()
    w = env.world
    bo = O.BatchedOracle(spec)
    # reset state == the oracle's Philox restatement
    want = O.reset_positions(spec, 3, w.episode - 1, np.arange(B, dtype=np.uint64),
                             np.float64 if dtype == torch.float64 else np.float32)
    assert np.array_equal(w.pos.cpu().numpy(), want) and (w.vel == 0).all()
    obs0, _, _ = bo.observe(_np(w.pos), _np(w.vel))
    for i in range(env.n):
        np.testing.assert_allclose(_np(obs_n[i]).T, obs0[i], rtol=0, atol=tol)
    pos, vel = _np(w.pos), _np(w.vel)
    gen = torch.Generator(device='cuda').manual_seed(9)
    for t in range(T):
        action_n = [torch.softmax(torch.randn(B, 5, generator=gen, device='cuda', dtype=dtype), -1) for _ in range(env.n)]
        if dtype == torch.float32:
            pos, vel = _np(w.pos), _np(w.vel)
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        act = np.stack([_np(a).T for a in action_n])
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act)
        np.testing.assert_allclose(_np(w.pos), pos, rtol=tol, atol=tol)
        np.testing.assert_allclose(_np(w.vel), vel, rtol=tol, atol=tol)
        for i in range(env.n):
            np.testing.assert_allclose(_np(new_obs_n[i]).T, obs[i], rtol=tol, atol=tol)
        r = np.stack([_np(x) for x in rew_n])
        if dtype == torch.float64:
            np.testing.assert_allclose(r, rew, rtol=0, atol=tol)
            if spec.kind == O.SIMPLE_TAG:
                assert np.array_equal(np.stack([_np(x) for x in info_n['n']]), info[:, 0])
            else:
                got = np.stack([np.stack([_np(c) for c in tup]) for tup in info_n['n']])
                np.testing.assert_allclose(got, info, rtol=0, atol=tol)
        else:
            assert np.mean(np.abs(r - rew) <= 1e-3) > 0.99


def test_action_views_are_zero_copy_and_all_input_forms_agree():
    from marl_amd import make_env
    B = 300
    outs = []
    act = torch.softmax(torch.randn(B, 4, 5, device='cuda', dtype=torch.float64), -1)
    for form in ('tensor', 'list', 'views', 'numpy'):
        env = make_env('simple_tag', benchmark=True, num_envs=B, seed=11)
        env.reset()
        if form == 'tensor':
            a = act
        elif form == 'list':
            a = [act[:, i] for i in range(4)]
        elif form == 'numpy':
            a = [act[:, i].cpu().numpy() for i in range(4)]
        else:
            for i in range(4):
                env.action_views[i].copy_(act[:, i])
            a = env.action_views
        obs_n, rew_n, _, _ = env.step(a)
        outs.append(torch.cat([o.clone() for o in obs_n], 1))
    for o in outs[1:]:
        assert torch.equal(o, outs[0])


def test_scenario_callbacks_and_state_setters():
    """The plug-in callbacks work stand-alone (make_env.py:40-43 wiring), entity.state setters feed
    the kernels (KAT-1 / KAT-2 of SURVEY 8c through the public API)."""
    from marl_amd import make_env
    env = make_env('simple_tag', benchmark=True, num_envs=2)
    env.reset()
    w = env.world
    far = [(5.0, 5.0), (-5.0, 5.0), (5.0, -5.0), (-5.0, -5.0)]
    for i, e in enumerate(w.entities):
        e.state.p_pos = np.array(far[i % 4]) * (1 + i)
    w.agents[0].state.p_pos = np.array([0.0, 0.0])
    w.agents[1].state.p_pos = np.array([0.1, 0.0])
    env.step([np.zeros(5)] * 4)
    np.testing.assert_allclose(_np(w.agents[0].state.p_vel), [[-0.5, 0.0]] * 2, atol=1e-12)
    np.testing.assert_allclose(_np(w.agents[0].state.p_pos), [[-0.05, 0.0]] * 2, atol=1e-12)
    assert (w.agents[1].state.p_vel.abs() < 1e-100).all()                 # one-sided (core.py:203-207)

    env = make_env('simple_spread', benchmark=True, num_envs=3)
    env.reset()
    w = env.world
    for i in range(3):
        w.landmarks[i].state.p_pos = np.array([float(i), 0.0])
        w.agents[i].state.p_pos = np.array([float(i), 0.0])
    scen = env.reward_callback.__self__
    for a in w.agents:
        assert (scen.reward(a, w) == -1.0).all()                          # self-collision term
        rew, coll, md, occ = scen.benchmark_data(a, w)
        assert (coll == 1).all() and (occ == 3).all() and (md == 0).all()
        assert scen.observation(a, w).shape == (3, 18)


def test_user_scenario_callbacks_run_on_world_step():
    """A scenario with its own torch callbacks (the reference's plug-in contract) still gets the
    HIP World.step; obs/reward come from the callbacks as in environment.py:105-110."""
    from marl_amd import make_env
    from marl_amd.environment import MultiAgentEnv
    base = make_env('simple_tag', num_envs=64, seed=2)
    scen = base.reset_callback.__self__
    w = base.world

    def my_obs(agent, world):
        return torch.cat([agent.state.p_vel, agent.state.p_pos], 1)

    def my_rew(agent, world):
        return -(agent.state.p_pos ** 2).sum(1)

    env = MultiAgentEnv(w, scen.reset_world, my_rew, my_obs)
    assert not env._fused and [s.shape[0] for s in env.observation_space] == [4] * 4
    env.reset()
    ref = make_env('simple_tag', num_envs=64, seed=2)
    ref.world.episode = w.episode - 1
    ref.reset()
    assert torch.equal(ref.world.pos, w.pos)
    act = torch.rand(64, 4, 5, device='cuda', dtype=torch.float64)
    o1, r1, _, _ = env.step(act)
    o2, r2, _, _ = ref.step(act)
    assert torch.equal(ref.world.pos, w.pos) and torch.equal(ref.world.vel, w.vel)
    for i in range(4):
        assert torch.equal(o1[i], o2[i][:, :4])
        assert torch.equal(r1[i], -(w.pos[i] ** 2).sum(0))


def test_episode_stats_all_reduce_single_process():
    from marl_amd import make_env
    from marl_amd.distributed import EpisodeStats
    env = make_env('simple_tag', benchmark=True, num_envs=2048, dtype=torch.float32, seed=4)
    env.reset()
    st = EpisodeStats(env.world)
    tot_r = torch.zeros(4, dtype=torch.float64, device='cuda')
    tot_c = torch.zeros(4, dtype=torch.float64, device='cuda')
    for t in range(25):
        _, rew_n, _, info_n = env.step(torch.softmax(torch.randn(2048, 4, 5, device='cuda'), -1))
        tot_r += torch.stack(rew_n).double().sum(1)
        tot_c += torch.stack(info_n['n']).double().sum(1)
    s = st.summary(25)
    assert s['env_steps'] == 25 * 2048 and s['episodes'] == 2048
    np.testing.assert_allclose(s['mean_episode_reward'], (tot_r / 2048).cpu().numpy(), rtol=1e-9, atol=1e-9)
    np.testing.assert_allclose(s['collisions_per_episode'], (tot_c / 2048).cpu().numpy(), rtol=1e-12)


@pytest.mark.parametrize('name,spec', [('simple', O.simple_spec()), ('simple_adversary', O.simple_adversary_spec()),
                                       ('simple_push', O.simple_push_spec())])
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)])
def test_f1_scenarios_episode_loop_matches_oracle(name, spec, dtype, tol):
    """SURVEY 8f row f1 through the public API: reset (Philox positions + goal) and 25 steps."""
    from marl_amd import make_env
    B, T = 384, 25
    env = make_env(name, benchmark=(name == 'simple_adversary'), num_envs=B, dtype=dtype, seed=5)
    assert [s.shape[0] for s in env.observation_space] == spec.obs_dims
    obs_n = env.reset()
    w = env.world
    ids = np.arange(B, dtype=np.uint64)
    npdt = np.float64 if dtype == torch.float64 else np.float32
    assert np.array_equal(w.pos.cpu().numpy(), O.reset_positions(spec, 5, w.episode - 1, ids, npdt))
    goal = w.goal.cpu().numpy()
    if spec.kind in O.GOAL_KINDS:
        assert np.array_equal(goal, O.reset_goals(spec, 5, w.episode - 1, ids))
    bo = O.BatchedOracle(spec)
    pos, vel = _np(w.pos), _np(w.vel)
    obs0, _, _ = bo.observe(pos, vel, goal=goal)
    for i in range(env.n):
        np.testing.assert_allclose(_np(obs_n[i]).T, obs0[i], rtol=0, atol=tol)
    gen = torch.Generator(device='cuda').manual_seed(3)
    for t in range(T):
        action_n = [torch.softmax(torch.randn(B, 5, generator=gen, device='cuda', dtype=dtype), -1) for _ in range(env.n)]
        if dtype == torch.float32:
            pos, vel = _np(w.pos), _np(w.vel)
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, np.stack([_np(a).T for a in action_n]), goal=goal)
        np.testing.assert_allclose(_np(w.pos), pos, rtol=tol, atol=tol)
        np.testing.assert_allclose(_np(w.vel), vel, rtol=tol, atol=tol)
        for i in range(env.n):
            np.testing.assert_allclose(_np(new_obs_n[i]).T, obs[i], rtol=tol, atol=tol)
        np.testing.assert_allclose(np.stack([_np(x) for x in rew_n]), rew, rtol=tol, atol=tol * 10)
        if name == 'simple_adversary':
            assert tuple(info_n['n'][0].shape) == (B,)                        # adversary: scalar per env
            np.testing.assert_allclose(_np(info_n['n'][0]), info[0, 0], rtol=tol, atol=tol)
            got = np.stack([_np(c) for c in info_n['n'][1]])                  # good agent: L+1 tuple
            np.testing.assert_allclose(got, info[1], rtol=tol, atol=tol)
        else:
            assert info_n == {'n': [{}] * env.n}
    if name != 'simple_adversary':
        with pytest.raises(AttributeError):                                   # like the reference
            make_env(name, benchmark=True, num_envs=4)


@pytest.mark.parametrize('name', ['L1F2_3Targets_100observe', 'S1L2F2_10Targets_wide', 'S1L2F2_10Targets_wide_rewardFixed',
                                  'S1L1F1_10Targets_wide'])
def test_hierarchy_scenarios_episode_loop_matches_oracle(name):
    """SURVEY 8f row f2 through the public API: make_env by the reference's file name, MultiDiscrete
    action spaces for the leaders (environment.py:61-69), 25 steps against the oracle (f64)."""
    from marl_amd import make_env, spaces
    from marl_amd.scenarios import hierarchy
    B, T, tol = 256, 25, 1e-9
    env = make_env(name, benchmark=True, num_envs=B, dtype=torch.float64, seed=8)
    w = env.world
    p = hierarchy.PRESETS[name]
    A = p['n_follower'] + p['n_leader'] + p['n_super']
    z = p['zoom']
    spec = O.ScenarioSpec(
        kind=O.HIER_PURSUIT, n_agents=A, n_landmarks=p['n_landmarks'], adversary=[True] * A,
        size=[a.size for a in w.agents] + [l.size for l in w.landmarks], accel=[3.0] * A, max_speed=[1.0] * A,
        collide=[True] * (A + p['n_landmarks']), movable=[True] * A + [False] * p['n_landmarks'], u_noise=[None] * A,
        mass=[1.0] * (A + p['n_landmarks']), agent_pos_range=1.0, landmark_pos_range=1.0, dim_c=7,
        silent=[a.silent for a in w.agents], force_id=[a.force_id for a in w.agents],
        forced_id=[a.forced_id for a in w.agents], advleader=[a.advleader for a in w.agents],
        advsuperleader=[a.advsuperleader for a in w.agents], obs_r=[a.obs_r for a in w.agents],
        reward_variant=p['reward_variant'])
    assert abs(spec.obs_r[0] - 0.2 * z) < 1e-15
    for sp, a in zip(env.action_space, w.agents):
        assert sp == (spaces.Discrete(5) if a.silent else spaces.MultiDiscrete([[0, 4], [0, 6]]))
    assert [s.shape[0] for s in env.observation_space] == spec.obs_dims
    assert [tuple(v.shape) for v in env.action_views] == [(B, d) for d in spec.act_dims]
    obs_n = env.reset()
    ids = np.arange(B, dtype=np.uint64)
    assert np.array_equal(w.pos.cpu().numpy(), O.reset_positions(spec, 8, w.episode - 1, ids, np.float64))
    bo = O.BatchedOracle(spec)
    pos, vel = _np(w.pos), _np(w.vel)
    obs0, _, _ = bo.observe(pos, vel)
    for i in range(env.n):
        assert np.array_equal(_np(obs_n[i]).T, obs0[i])
    gen = torch.Generator(device='cuda').manual_seed(3)
    for t in range(T):
        action_n = [torch.softmax(torch.randn(B, d, generator=gen, device='cuda', dtype=torch.float64), -1)
                    for d in spec.act_dims]
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        act = np.concatenate([_np(a).T for a in action_n], 0)
        pos, vel, obs, rew, info, lf = bo.env_step(pos, vel, act, return_lforce=True)
        np.testing.assert_allclose(_np(w.pos), pos, rtol=0, atol=tol)
        np.testing.assert_allclose(_np(w.lforce), lf, rtol=0, atol=tol)
        for i, a in enumerate(w.agents):
            np.testing.assert_allclose(_np(new_obs_n[i]).T, obs[i], rtol=0, atol=tol)
            np.testing.assert_allclose(_np(a.leader_force).T, lf[i], rtol=0, atol=tol)
            want_c = 0.0 if a.silent else act[w.act_row0[i] + 5:w.act_row0[i] + 12].T
            assert np.array_equal(_np(a.state.c), np.broadcast_to(want_c, (B, 7)))       # core.py:226-230
        np.testing.assert_allclose(np.stack([_np(x) for x in rew_n]), rew, rtol=0, atol=tol)
        assert np.array_equal(np.stack([_np(x) for x in info_n['n']]), info[:, 0])
    # the scenario callbacks stand alone too (make_env.py:40-43 wiring)
    scen = env.reward_callback.__self__
    assert torch.equal(scen.reward(w.agents[0], w), rew_n[0]) and torch.equal(scen.observation(w.agents[-1], w), new_obs_n[-1])
    with pytest.raises(ValueError):
        env.step([torch.zeros(B, 5, device='cuda', dtype=torch.float64)] * env.n)    # leaders need 12 entries


def test_training_loop_with_target_move_and_benchmark_reset():
    """The fork's loop shape (experiments/train_and_eval.py:313-314, :103-126): env.step, then
    target_move(env); reset_world_for_benchmark as the evaluation env's reset callback."""
    from marl_amd import make_env
    from marl_amd.environment import MultiAgentEnv
    from marl_amd.experiments import target_move, reset_world_for_benchmark
    B = 512
    env = make_env('L1F2_3Targets_100observe', benchmark=True, num_envs=B, seed=6)
    env.reset()
    w = env.world
    spec_sizes = [e.size for e in w.entities]
    total = 0
    gen = torch.Generator(device='cuda').manual_seed(1)
    for t in range(25):
        env.step([torch.rand(B, d, generator=gen, device='cuda', dtype=torch.float64) for d in w.act_dims])
        before = w.pos.clone()
        cnt = target_move(env)
        assert cnt.dtype == torch.int32 and tuple(cnt.shape) == (B,)
        moved = (w.pos != before).any(1)                                  # [E, B]
        assert not moved[:3].any() and torch.equal(moved[3:].sum(0).int(), cnt)
        # a landmark moved iff a follower (agents 0, 1) touched it
        d = (before[3:, None] - before[None, :2]).pow(2).sum(2).sqrt()     # [L, 2, B]
        dmin = torch.tensor([[spec_sizes[3 + l] + spec_sizes[a] for a in range(2)] for l in range(3)],
                            device='cuda', dtype=torch.float64)[..., None]
        assert torch.equal((d < dmin).any(1), moved[3:])
        total += int(cnt.sum())
    assert total > 0 and w.move_counter == 25
    scen = env.reward_callback.__self__
    ev = MultiAgentEnv(w, reset_world_for_benchmark, scen.reward, scen.observation, scen.benchmark_data)
    lm = w.pos[3:].clone()
    obs_n = ev.reset()
    want = torch.tensor([[-0.5, -0.5], [-0.5, 0.5], [0.5, -0.5]], device='cuda', dtype=torch.float64)
    for i in range(3):
        assert torch.equal(w.agents[i].state.p_pos, want[i].expand(B, 2)) and (w.agents[i].state.p_vel == 0).all()
        assert torch.equal(obs_n[i][:, 2:4], want[i].expand(B, 2))
    assert torch.equal(w.pos[3:], lm)                                     # landmarks keep their positions


@pytest.mark.parametrize('name,mk', [('simple_speaker_listener', O.speaker_listener_spec), ('simple_reference', O.simple_reference_spec)])
def test_speaking_scenarios_episode_loop_matches_oracle(name, mk):
    """SURVEY 8f row f1 (speaking half) through the public API: action spaces as the reference builds
    them (environment.py:42-76), reset (Philox positions + goals), 25 steps against the oracle (f64)."""
    from marl_amd import make_env, spaces
    spec = mk()
    B, T = 320, 25
    env = make_env(name, num_envs=B, dtype=torch.float64, seed=9)
    want_spaces = {'simple_speaker_listener': [spaces.Discrete(3), spaces.Discrete(5)],
                   'simple_reference': [spaces.MultiDiscrete([[0, 4], [0, 9]])] * 2}[name]
    assert env.action_space == want_spaces
    assert [s.shape[0] for s in env.observation_space] == spec.obs_dims
    obs_n = env.reset()
    w = env.world
    ids = np.arange(B, dtype=np.uint64)
    assert np.array_equal(w.pos.cpu().numpy(), O.reset_positions(spec, 9, w.episode - 1, ids))
    goal = w.goal.cpu().numpy().reshape(spec.goal_rows, B)
    assert np.array_equal(goal, O.reset_goals(spec, 9, w.episode - 1, ids).reshape(spec.goal_rows, B))
    bo = O.BatchedOracle(spec)
    pos, vel = _np(w.pos), _np(w.vel)
    obs0, _, _ = bo.observe(pos, vel, goal=goal)
    for i in range(env.n):
        assert np.array_equal(_np(obs_n[i]).T, obs0[i])
    gen = torch.Generator(device='cuda').manual_seed(2)
    for t in range(T):
        action_n = [torch.rand(B, d, generator=gen, device='cuda', dtype=torch.float64) for d in spec.act_dims]
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        act = np.concatenate([_np(a).T for a in action_n], 0)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act, goal=goal)
        np.testing.assert_allclose(_np(w.pos), pos, rtol=0, atol=1e-9)
        for i, a in enumerate(w.agents):
            np.testing.assert_allclose(_np(new_obs_n[i]).T, obs[i], rtol=0, atol=1e-9)
            assert np.array_equal(_np(a.state.c).T, bo.comm_state(act)[i])
        np.testing.assert_allclose(np.stack([_np(x) for x in rew_n]), rew, rtol=0, atol=1e-9)
        assert info_n == {'n': [{}] * env.n}
    if name == 'simple_reference':                                        # no benchmark_data attribute at all
        with pytest.raises(AttributeError):
            make_env(name, benchmark=True, num_envs=4)
    else:                                                                 # defined but broken: fails when called (:62-64)
        env = make_env(name, benchmark=True, num_envs=4)
        env.reset()
        with pytest.raises(NameError):
            env.step([torch.rand(4, d, device='cuda', dtype=torch.float64) for d in spec.act_dims])


def test_simple_crypto_episode_loop_matches_oracle():
    """simple_crypto through the public API with benchmark=True: Discrete(4) spaces for three agents that
    only speak, info = (state.c, goal colour) (simple_crypto.py:78-80), rewards with the all-zero-message
    switch (:104,109,119)."""
    from marl_amd import make_env, spaces
    spec = O.simple_crypto_spec()
    B, T = 320, 25
    env = make_env('simple_crypto', benchmark=True, num_envs=B, dtype=torch.float64, seed=4)
    assert env.action_space == [spaces.Discrete(4)] * 3
    assert [s.shape[0] for s in env.observation_space] == [4, 8, 8]
    obs_n = env.reset()
    w = env.world
    ids = np.arange(B, dtype=np.uint64)
    goal = w.goal.cpu().numpy().reshape(2, B)
    assert np.array_equal(goal, O.reset_goals(spec, 4, w.episode - 1, ids).reshape(2, B))
    assert set(np.unique(goal)) == {0, 1}
    bo = O.BatchedOracle(spec)
    pos, vel = _np(w.pos), _np(w.vel)
    obs0, _, _ = bo.observe(pos, vel, goal=goal)
    for i in range(3):
        assert np.array_equal(_np(obs_n[i]).T, obs0[i])
    gen = torch.Generator(device='cuda').manual_seed(6)
    onehot = np.eye(4)[goal[0]]                                           # [B, 4] goal colour
    for t in range(T):
        action_n = [torch.rand(B, 4, generator=gen, device='cuda', dtype=torch.float64) for _ in range(3)]
        action_n[t % 3][::5] = 0.0                                        # some agents stay silent
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        act = np.concatenate([_np(a).T for a in action_n], 0)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act, goal=goal)
        assert np.array_equal(_np(w.pos), pos)                            # nobody moves
        for i, a in enumerate(w.agents):
            assert np.array_equal(_np(new_obs_n[i]).T, obs[i])
            assert np.array_equal(_np(a.state.c), _np(action_n[i]))
            c, colour = info_n['n'][i]
            assert np.array_equal(_np(c), _np(action_n[i])) and np.array_equal(_np(colour), onehot)
        np.testing.assert_allclose(np.stack([_np(x) for x in rew_n]), rew, rtol=0, atol=1e-9)
        silent = ~_np(action_n[0]).any(1)
        assert (np.stack([_np(x) for x in rew_n])[0][silent] == 0).all()


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_scripted_prey_matches_oracle_and_drives_the_env(dtype):
    """SURVEY 8f row f3 (experiments/simple_train.py:157-205, parity unpinned: see the oracle's header):
    good_agent_action / good_agent_victim_action for every env in one launch, against the vectorised
    oracle (f64: bit for bit, same fma norm), then used as action_n[-1] the way the training loop does."""
    from marl_amd import make_env
    from marl_amd.experiments import good_agent_action, good_agent_victim_action
    B = 3000 + 7
    env = make_env('L1F4_MovingTarget', num_envs=B, dtype=dtype, seed=3)
    env.reset()
    w = env.world
    spec = O.hier_spec_from_golden(np.load(os.path.join(os.path.dirname(__file__), 'golden', 'm_L1F4_MovingTarget.npz')))
    w.pos[:, :, ::3] *= 2.5                                                # some envs far outside: the cap
    w.mark_dirty()
    prey = w.agents[-1]
    act, edge = good_agent_action(prey, w)
    assert tuple(act.shape) == (B, 5) and tuple(edge.shape) == (B, 2)
    want_act, want_edge = O.scripted_action(spec, _np(w.pos), -1, 0, dtype=np.float64 if dtype == torch.float64 else np.float32)
    if dtype == torch.float64:
        assert np.array_equal(_np(act).T, want_act) and np.array_equal(_np(edge).T, want_edge)
    else:
        same = (_np(edge).T == want_edge).all(0)                           # f32 sums may pick another corner on near-ties
        assert same.mean() > 0.999
        np.testing.assert_allclose(_np(act).T[:, same], want_act[:, same], rtol=1e-6, atol=1e-7)
    assert float(act.max()) == pytest.approx(1.3) and (act[:, 0] == 0).all()
    vact = good_agent_victim_action(prey, w, w.agents[2])
    want_v, _ = O.scripted_action(spec, _np(w.pos), -1, 1, 2, dtype=np.float64 if dtype == torch.float64 else np.float32)
    assert np.array_equal(_np(vact).T, want_v.astype(np.float64))
    # the loop: action_n[-1] = good_act (simple_train.py:269-273)
    gen = torch.Generator(device='cuda').manual_seed(0)
    action_n = [torch.rand(B, d, generator=gen, device='cuda', dtype=dtype) for d in w.act_dims]
    action_n[-1] = act
    before = w.pos[-1].clone()
    env.step(action_n)
    moved = w.pos[-1] - before                                             # the prey accelerates towards its corner
    pull = torch.sign(edge.t() - before)
    far = (edge.t() - before).abs() > 0.2
    assert ((torch.sign(moved) == pull) | ~far).float().mean() > 0.9


def test_previous_observations_survive_the_next_step():
    """The reference's loops read obs_n after the next env.step (experience(obs_n[i], ..., new_obs_n[i]),
    simple_train.py:284-287).  The returned tensors are views of kernel output rows; two alternating buffers
    keep the previous step's views intact (callback path and fused path, reset included)."""
    from marl_amd import make_env
    B = 257
    for kwargs in (dict(), dict(benchmark=True)):
        env = make_env('simple_tag', num_envs=B, dtype=torch.float64, seed=0, **kwargs)
        obs0 = env.reset()
        keep0 = [o.clone() for o in obs0]
        gen = torch.Generator(device='cuda').manual_seed(0)
        act = [torch.rand(B, 5, generator=gen, device='cuda', dtype=torch.float64) for _ in range(env.n)]
        obs1, _, _, _ = env.step(act)
        keep1 = [o.clone() for o in obs1]
        for a, b in zip(obs0, keep0):
            assert torch.equal(a, b)
        obs2, _, _, _ = env.step(act)
        for a, b, c in zip(obs1, keep1, obs2):
            assert torch.equal(a, b) and not torch.equal(a, c)
        assert obs2[0].data_ptr() == obs0[0].data_ptr()                    # two buffers, alternating
        env.world.obs_buffers = 1                                          # opt out: one buffer, half the memory
        env.world._alloc_outputs()
        o1, _, _, _ = env.step(act)
        o2, _, _, _ = env.step(act)
        assert o1[0].data_ptr() == o2[0].data_ptr()



def test_comm_state_setter_invalidates_cached_observations():
    """agent.state.c = ... must mark the world dirty like the p_pos / p_vel setters: scenarios that observe
    other.state.c (simple_reference.py:79-89) would otherwise return the gather cached before the write."""
    from marl_amd import make_env
    env = make_env('simple_reference', num_envs=64, dtype=torch.float64, seed=3)
    env.reset()
    sc_obs = env.observation_callback
    w = env.world
    a0, a1 = w.agents
    before = sc_obs(a1, w).clone()
    a0.state.c = np.arange(1, w.dim_c + 1, dtype=np.float64)
    after = sc_obs(a1, w)                                    # agent 1 sees agent 0's message in its last dim_c entries
    assert torch.equal(after[:, -w.dim_c:], torch.arange(1, w.dim_c + 1, dtype=torch.float64, device='cuda').expand(64, -1))
    assert torch.equal(after[:, :-w.dim_c], before[:, :-w.dim_c])


def test_library_calls_run_on_the_worlds_device():
    """_hip.on_device: a no-op when the world's device is current, otherwise it switches for the call and back.
    (One GPU on the test box: the switch itself is exercised with the same index.)"""
    from marl_amd import _hip, make_env
    cur = torch.cuda.current_device()
    with _hip.on_device(torch.device('cuda', cur)) as g:
        assert g.prev is None and torch.cuda.current_device() == cur
    with _hip.on_device('cuda'):
        assert torch.cuda.current_device() == cur
    env = make_env('simple_tag', num_envs=32, device='cuda:%d' % cur, dtype=torch.float32)
    env.reset()
    env.step(torch.rand(32, 4, 5, device='cuda'))
    assert torch.cuda.current_device() == cur


@pytest.mark.parametrize('scenario,noise', [('simple_tag', True), ('S1L2F2_10Targets_wide', False), ('simple_world_comm', False)])
def test_captured_steps_replay_equals_eager_stepping(scenario, noise):
    """marl_amd.graph.CapturedSteps: K env.step launches recorded in a HIP graph and replayed with one host call leave
    the world, the outputs and the Python-side bookkeeping exactly as K eager steps do (ragged action rows, recorded
    u_noise draws, statistics rows included); a second replay continues from there."""
    from marl_amd import make_env
    from marl_amd.graph import CapturedSteps
    B, K = 3001, 7

    def build():
        env = make_env(scenario, benchmark=True, num_envs=B, dtype=torch.float32, seed=11)
        w = env.world
        if noise:
            for a in w.agents:
                a.u_noise = 0.3
            w.invalidate_descriptor()
            w.noise = torch.randn(len(w.agents), 2, B, generator=torch.Generator(device='cuda').manual_seed(3), device='cuda')
        w.enable_stats()
        g = torch.Generator(device='cuda').manual_seed(5)
        acts = []
        for _ in range(K):
            a = w.new_action_buffer()
            a.copy_(torch.rand(a.shape, generator=g, device='cuda'))
            acts.append(a)
        env.reset()
        return env, acts
    e1, a1 = build()
    e2, a2 = build()
    cap = CapturedSteps(e2, a2)
    assert torch.equal(e1.world.pos, e2.world.pos) and torch.equal(e1.world.stats, e2.world.stats)   # capture ran nothing
    for rnd in range(2):
        for a in a1:
            out1 = e1.step(a)
        out2 = cap.replay()
        torch.cuda.synchronize()
        for x, y in zip(out1[0] + out1[1], out2[0] + out2[1]):                      # observations and rewards
            assert torch.equal(x, y)
        w1, w2 = e1.world, e2.world
        assert torch.equal(w1.pos, w2.pos) and torch.equal(w1.vel, w2.vel) and torch.equal(w1.stats, w2.stats)
        assert torch.equal(w1.info, w2.info) and (w1.lforce is None or torch.equal(w1.lforce, w2.lforce))
        assert w2.obs_views[0].data_ptr() == out2[0][0].data_ptr()           # world.obs points at the replay's last buffer
        # scenario callbacks after a replay serve the replay's last outputs without a new launch
        assert torch.equal(e2.reward_callback(w2.agents[0], w2), out2[1][0])
    with pytest.raises(ValueError):                                              # not the world's row stride: refused
        CapturedSteps(e2, [torch.zeros(tuple(a2[0].shape), device='cuda')])


def test_ambiguous_action_tensor_shape_is_decided_by_the_row_stride():
    """With 5 agents and 5 envs an [n, 5, B] SoA tensor and a reference-like [B, n, 5] tensor have the same shape:
    only a buffer with the world's own row stride (world.new_action_buffer()) is read as SoA."""
    from marl_amd import make_env
    env = make_env('simple_tag', num_envs=5, dtype=torch.float64, seed=2, num_adversaries=4, num_good_agents=1)
    assert env.n == 5 and env.num_envs == 5
    act = torch.softmax(torch.randn(5, 5, 5, device='cuda', dtype=torch.float64), -1)        # [B, n, 5]
    env.reset()
    state = (env.world.pos.clone(), env.world.vel.clone())
    o1 = [o.clone() for o in env.step(act)[0]]
    env.world.pos.copy_(state[0]); env.world.vel.copy_(state[1]); env.world.mark_dirty()
    o2 = [o.clone() for o in env.step([act[:, i] for i in range(5)])[0]]                    # per-agent list: unambiguous
    env.world.pos.copy_(state[0]); env.world.vel.copy_(state[1]); env.world.mark_dirty()
    soa = env.world.new_action_buffer()
    soa.copy_(act.permute(1, 2, 0))                                                          # [n, 5, B], world stride
    o3 = [o.clone() for o in env.step(soa)[0]]
    for a, b, c in zip(o1, o2, o3):
        assert torch.equal(a, b) and torch.equal(a, c)


def test_batch_multi_agent_env_concatenates_member_envs():
    """environment.py:326-373: n, spaces, step and reset of the list wrapper, over two batched worlds of different seeds."""
    from marl_amd import make_env
    from marl_amd.environment import BatchMultiAgentEnv
    B = 96
    envs = [make_env('simple_tag', benchmark=True, num_envs=B, dtype=torch.float64, seed=s) for s in (1, 2)]
    ref = [make_env('simple_tag', benchmark=True, num_envs=B, dtype=torch.float64, seed=s) for s in (1, 2)]
    batch = BatchMultiAgentEnv(envs)
    assert batch.n == 8 and batch.action_space is envs[0].action_space and batch.observation_space is envs[0].observation_space
    obs_n = batch.reset()
    want = ref[0].reset() + ref[1].reset()
    assert len(obs_n) == 8 and all(torch.equal(a, b) for a, b in zip(obs_n, want))
    gen = torch.Generator(device='cuda').manual_seed(0)
    action_n = [torch.rand(B, 5, generator=gen, device='cuda', dtype=torch.float64) for _ in range(8)]
    obs_n, rew_n, done_n, info_n = batch.step(action_n, 0)
    o0, r0, d0, _ = ref[0].step(action_n[:4])
    o1, r1, d1, _ = ref[1].step(action_n[4:])
    assert all(torch.equal(a, b) for a, b in zip(obs_n, o0 + o1)) and all(torch.equal(a, b) for a, b in zip(rew_n, r0 + r1))
    assert len(done_n) == 8 and info_n == {'n': []}
    assert not torch.equal(obs_n[0], obs_n[4])                       # two different worlds


SCRIPTED_CASES = ['s1_simple_tag_scripted_prey', 's1_simple_tag_scripted_prey_dense',
                  's2_simple_tag_scripted_chasers', 's2_simple_tag_scripted_chasers_dense']


@pytest.mark.parametrize('name', SCRIPTED_CASES)
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 1e-9), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_scripted_agents_match_reference(golden_dir, name, dtype, tol):
    """core.py:126-127 / environment.py:21-22 through the public API, built the way the reference's users build it
    (scenario.make_world(), agent.action_callback = ..., MultiAgentEnv(world, ...)): the callbacks of
    tests/golden/gen_golden.py:SCRIPTED restated on the batched state; 25 steps against the reference's trajectories
    (f64 free-running <= 1e-9, f32 teacher-forced <= 1e-4), incl. what the callbacks returned and u_noise on top of it."""
    import marl_amd.scenarios as scenarios
    from marl_amd.core import Action
    from marl_amd.environment import MultiAgentEnv
    from marl_amd.graph import CapturedSteps
    from tests.gpu_util import dev, golden_soa
    g = np.load(os.path.join(golden_dir, name + '.npz'))
    S, T = g['act'].shape[:2]
    un = float(g['u_noise'])

    def toward(target_of, gain, cap):
        def cb(agent, world):
            a = Action()
            a.u = torch.clamp(gain * (target_of(world).state.p_pos - agent.state.p_pos), -cap, cap)
            a.c = torch.zeros(world.num_envs, world.dim_c, dtype=world.dtype, device=world.device)
            return a
        return cb
    cbs = ({3: toward(lambda w: w.landmarks[0], 2.0, 4.0)} if name.startswith('s1_') else
           {0: toward(lambda w: w.agents[3], 4.0, 3.0), 2: toward(lambda w: w.agents[3], 1.5, 3.0)})
    assert sorted(cbs) == list(g['scripted'])
    scen = scenarios.load('simple_tag.py').Scenario(num_envs=S, dtype=dtype, seed=1)
    world = scen.make_world()
    for i, cb in cbs.items():
        world.agents[i].action_callback = cb
    if un:
        for a in world.agents:
            a.u_noise = un
        world.invalidate_descriptor()
    env = MultiAgentEnv(world, scen.reset_world, scen.reward, scen.observation, scen.benchmark_data)
    policy = [i for i in range(4) if i not in cbs]
    assert env.n == len(policy) == len(env.action_space) == len(env.observation_space) == len(env.action_views)
    assert [a._index for a in env.agents] == policy and [a._index for a in world.scripted_agents] == sorted(cbs)
    assert len(env.reset()) == env.n

    def set_state(p, v):
        world.pos.copy_(dev(p, dtype))
        world.vel.copy_(dev(v, dtype))
        world.mark_dirty()
    set_state(golden_soa(g, 'pos0'), golden_soa(g, 'vel0'))
    obs0 = torch.cat([env._get_obs(a) for a in env.agents], 1)
    np.testing.assert_allclose(_np(obs0), g['obs0'], rtol=0, atol=0 if dtype == torch.float64 else 1e-6)
    for t in range(T):
        if dtype == torch.float32 and t:
            set_state(golden_soa(g, 'pos', t - 1), golden_soa(g, 'vel', t - 1))
        if un:
            world.noise = dev(golden_soa(g, 'noise', t), dtype)
        obs_n, rew_n, done_n, info_n = env.step([dev(g['act'][:, t, i], dtype) for i in policy])
        assert len(obs_n) == len(rew_n) == len(done_n) == len(info_n['n']) == env.n
        for i in cbs:
            u = _np(world.agents[i].action.u)
            if dtype == torch.float64:
                assert np.array_equal(u, g['u_scripted'][:, t, i])           # same IEEE operations as the reference
            else:
                np.testing.assert_allclose(u, g['u_scripted'][:, t, i], rtol=1e-5, atol=1e-5)
        np.testing.assert_allclose(_np(world.pos), golden_soa(g, 'pos', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(_np(world.vel), golden_soa(g, 'vel', t), rtol=tol, atol=tol)
        np.testing.assert_allclose(_np(torch.cat(obs_n, 1)), g['obs'][:, t], rtol=tol, atol=tol)
        if dtype == torch.float64:
            np.testing.assert_allclose(_np(torch.stack(rew_n, 1)), g['rew'][:, t], rtol=0, atol=tol)
            assert np.array_equal(_np(torch.stack(info_n['n'], 1)), g['info'][:, t, :, 0])
    with pytest.raises(ValueError):                     # the SoA / [B, n, 5] forms would have to name the scripted rows
        env.step(torch.zeros(S, env.n, 5, dtype=dtype, device='cuda'))
    with pytest.raises(RuntimeError):
        CapturedSteps(env, [world.new_action_buffer()])
    # integer actions (environment.py:176-182) for the policy agents, the callbacks for the others: same as the one-hots
    state = (world.pos.clone(), world.vel.clone())
    world.noise = torch.zeros_like(world.vel) if un else None
    k = torch.randint(0, 5, (env.n, S), device='cuda')
    onehot = [torch.zeros(S, 5, dtype=dtype, device='cuda') for _ in range(env.n)]
    for j in range(env.n):
        for kk, slot in ((1, 2), (2, 1), (3, 4), (4, 3)):
            onehot[j][k[j] == kk, slot] = 1.0
    o1 = [o.clone() for o in env.step(onehot)[0]]
    world.pos.copy_(state[0]); world.vel.copy_(state[1]); world.mark_dirty()
    env.discrete_action_input = True
    o2 = env.step([k[j] for j in range(env.n)])[0]
    for a, b in zip(o1, o2):
        assert torch.equal(a, b)


def test_scripted_speaking_leader_in_a_command_hierarchy():
    """A scripted agent that speaks (core.py:126-127 with action.c): in S1L2F2_3Targets_100observe the first leader gets
    its physical force AND its 7-entry command from a callback; the command pushes its followers (core.py:153-194) and
    lands in state.c.  Against the oracle, which takes the callback's u as the force and reads the command from the
    leader's comm rows."""
    import marl_amd.scenarios as scenarios
    from marl_amd.core import Action
    from marl_amd.environment import MultiAgentEnv
    B, T = 300, 6
    g = np.load(os.path.join(os.path.dirname(__file__), 'golden', 'h_S1L2F2_3Targets_100observe.npz'))
    spec = O.hier_spec_from_golden(g)
    scen = scenarios.load('S1L2F2_3Targets_100observe.py').Scenario(num_envs=B, dtype=torch.float64, seed=4)
    world = scen.make_world()
    lead = next(i for i, a in enumerate(world.agents) if not a.silent and not a.advsuperleader)
    cmd = torch.tensor([0.1, 0.2, 0.05, 0.3, 0.15, 0.9, 0.4], dtype=torch.float64, device='cuda')   # argmax 5: "towards me"

    def cb(agent, w):
        a = Action()
        a.u = 0.5 * torch.sign(w.landmarks[0].state.p_pos - agent.state.p_pos)
        a.c = cmd.expand(w.num_envs, 7) * (1.0 + agent.state.p_pos[:, :1].abs())      # per-env scaling, same argmax
        return a
    world.agents[lead].action_callback = cb
    env = MultiAgentEnv(world, scen.reset_world, scen.reward, scen.observation, scen.benchmark_data)
    policy = [i for i in range(spec.n_agents) if i != lead]
    assert env.n == spec.n_agents - 1 and [a._index for a in env.agents] == policy
    env.reset()
    bo = O.BatchedOracle(spec)
    pos, vel = _np(world.pos), _np(world.vel)
    gen = torch.Generator(device='cuda').manual_seed(5)
    offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
    for t in range(T):
        action_n = [torch.rand(B, spec.act_dims[i], generator=gen, device='cuda', dtype=torch.float64) for i in policy]
        u = 0.5 * np.sign(pos[spec.n_agents] - pos[lead])                       # landmark 0 = entity n_agents
        c = _np(cmd)[:, None] * (1.0 + np.abs(pos[lead][:1]))
        obs_n, rew_n, _, info_n = env.step(action_n)
        act = np.zeros((sum(spec.act_dims), B))
        for i, a in zip(policy, action_n):
            act[offs[i]:offs[i + 1]] = _np(a).T
        act[offs[lead] + 5:offs[lead + 1]] = c
        pos, vel, obs, rew, info, lf = bo.env_step(pos, vel, act, return_lforce=True, scripted_u={lead: u})
        np.testing.assert_allclose(_np(world.pos), pos, rtol=0, atol=1e-9)
        np.testing.assert_allclose(_np(world.lforce), lf, rtol=0, atol=1e-9)
        np.testing.assert_allclose(_np(torch.cat(obs_n, 1)).T, np.concatenate([obs[i] for i in policy], 0), rtol=0, atol=1e-9)
        np.testing.assert_allclose(np.stack([_np(r) for r in rew_n]), rew[policy], rtol=0, atol=1e-9)
        assert np.array_equal(_np(world.agents[lead].state.c).T, c)
    assert np.abs(lf).max() > 0                                                 # somebody was pushed by the scripted command


@pytest.mark.parametrize('with_done', [True, False], ids=['callback_path', 'fused_path'])
def test_shared_reward_and_done_callback_match_reference(golden_dir, with_done):
    """environment.py:33,113-115 (world.collaborative: every agent receives the sum of the rewards) and :146-149 (a
    done_callback), against a fixture generated from the reference with both set on simple_spread.  A done_callback takes
    the env off the fused launch (World.step + the callbacks, like environment.py:105-110); without it the same rewards
    come from the fused path."""
    import marl_amd.scenarios as scenarios
    from marl_amd.environment import MultiAgentEnv
    from tests.gpu_util import dev, golden_soa
    g = np.load(os.path.join(golden_dir, 'g17_simple_spread_shared_reward_done.npz'))
    S, T = g['act'].shape[:2]
    scen = scenarios.load('simple_spread.py').Scenario(num_envs=S, dtype=torch.float64, seed=1)
    world = scen.make_world()
    world.collaborative = True

    def done(agent, world):
        d = torch.stack([(agent.state.p_pos - l.state.p_pos).pow(2).sum(1).sqrt() for l in world.landmarks]).min(0).values
        return d < 0.25
    env = MultiAgentEnv(world, scen.reset_world, scen.reward, scen.observation, scen.benchmark_data,
                        done_callback=done if with_done else None)
    assert env.shared_reward and env._fused == (not with_done)
    env.reset()
    world.pos.copy_(dev(golden_soa(g, 'pos0'), torch.float64))
    world.vel.copy_(dev(golden_soa(g, 'vel0'), torch.float64))
    world.mark_dirty()
    for t in range(T):
        obs_n, rew_n, done_n, info_n = env.step([dev(g['act'][:, t, i], torch.float64) for i in range(env.n)])
        np.testing.assert_allclose(_np(world.pos), golden_soa(g, 'pos', t), rtol=0, atol=1e-9)
        np.testing.assert_allclose(_np(torch.cat(obs_n, 1)), g['obs'][:, t], rtol=0, atol=1e-9)
        np.testing.assert_allclose(_np(torch.stack(rew_n, 1)), g['rew'][:, t], rtol=0, atol=1e-9)
        assert rew_n[0] is rew_n[1] is rew_n[2]                              # [reward] * n (environment.py:115)
        if with_done:
            assert np.array_equal(torch.stack(done_n, 1).cpu().numpy(), g['done'][:, t])
        else:
            assert not any(d.any() for d in done_n)
    assert g['done'].any() and not g['done'].all()
