#!/usr/bin/env python3
"""Golden-vector generator (build container only; never runs on the GPU box).

Imports the *reference* implementation from /root/reference (read-only), rolls
out episodes and dumps inputs + expected outputs as small .npz fixtures next to
this file.  Nothing of the reference is copied: this script only contains the
harness scaffolding from SURVEY.md Appendix A (an in-memory ``gym`` stand-in so
``import multiagent`` works, and the ``forced_id/force_id = -1`` attributes the
fork's ``World.step`` needs, reference core.py:157-159).

Cases:
  G1  simple_tag      64 seeds x 25 steps, benchmark=True
  G2  simple_spread   64 seeds x 25 steps, benchmark=True (info 4-tuple)
  G3  scaled tag 48 adv / 16 good / 32 landmarks, 2 seeds x 12 steps
  G4  simple_tag with u_noise = 0.3 (core.py:147), 16 seeds x 25 steps
  G5  dense starts U(-0.3, 0.3): collisions / max_speed clamp / +-10 rewards fire
      (tag and spread)
  G6  out-of-bounds starts: all four branches of simple_tag bound()
  G7-G9 simple / simple_adversary / simple_push, default and dense starts
  H*   L1F*/S1L* leader-follower pursuit scenarios of the fork
  KAT known-answer cases 1, 2, 4 of SURVEY.md section 8c

Run:  python tests/golden/gen_golden.py
"""
import importlib.util
import os
import sys
import types

import numpy as np

REF = '/root/reference/multiagent-particle-envs'
OUT = os.path.dirname(os.path.abspath(__file__))


def _install_gym_stub():
    gym = types.ModuleType('gym')
    gym.Env = type('Env', (), {})
    gym.Space = type('Space', (), {})
    sp = types.ModuleType('gym.spaces')

    class Discrete(gym.Space):
        def __init__(self, n):
            self.n = n

    class Box(gym.Space):
        def __init__(self, low, high, shape=None, dtype=None):
            self.low, self.high, self.shape, self.dtype = low, high, shape, dtype

    class Tuple(gym.Space):
        def __init__(self, spaces):
            self.spaces = spaces

    sp.Discrete, sp.Box, sp.Tuple = Discrete, Box, Tuple
    sp.prng = types.SimpleNamespace(np_random=np.random)
    envs = types.ModuleType('gym.envs')
    reg = types.ModuleType('gym.envs.registration')
    reg.register = lambda **kw: None
    reg.EnvSpec = object
    gym.spaces, gym.envs, envs.registration = sp, envs, reg
    sys.modules.update({'gym': gym, 'gym.spaces': sp, 'gym.envs': envs,
                        'gym.envs.registration': reg})


def _load(path, name):
    spec = importlib.util.spec_from_file_location(name, path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


_install_gym_stub()
sys.path.insert(0, REF)
from multiagent.environment import MultiAgentEnv  # noqa: E402  (reference)
from multiagent.core import World, Agent, Landmark  # noqa: E402  (reference)

TAG = _load(REF + '/build/lib/multiagent/scenarios/simple_tag.py', 'ref_simple_tag')
SPREAD = _load(REF + '/build/lib/multiagent/scenarios/simple_spread.py', 'ref_simple_spread')
OTHER = {k: _load(REF + '/build/lib/multiagent/scenarios/%s.py' % k, 'ref_' + k)
         for k in ('simple', 'simple_adversary', 'simple_push')}


HIER_DIR = REF + '/multiagent/scenarios/'          # the fork's live scenarios
HIER = {}


def _hier(name):
    if name not in HIER:
        HIER[name] = _load(HIER_DIR + name + '.py', 'ref_' + name)
    return HIER[name]


def build_hier_env(name, seed):
    """L1F*/S1L* scenarios set force_id / forced_id themselves: no shim needed."""
    np.random.seed(seed)
    S = _hier(name).Scenario()
    world = S.make_world()
    env = MultiAgentEnv(world, S.reset_world, S.reward, S.observation, S.benchmark_data)
    return env, world


def rollout_hier(name, seeds, T, init=None, actions='rand', force_discrete=False):
    """Like rollout(); actions are ragged (5 physical + dim_c comm entries for non-silent agents,
    environment.py:161-168) and stored concatenated agent by agent; agent.leader_force recorded."""
    keys = ('pos0', 'vel0', 'obs0', 'lforce0', 'act', 'pos', 'vel', 'lforce', 'obs', 'rew', 'info')
    out = {k: [] for k in keys}
    act_dims = None
    for seed in seeds:
        env, world = build_hier_env(name, seed)
        A = len(world.agents)
        act_dims = [5 + (0 if a.silent else world.dim_c) for a in world.agents]
        obs0 = env.reset()
        if init is not None:
            init(world, np.random.RandomState(10_000 + seed))
            obs0 = [env._get_obs(a) for a in env.agents]
        pos0, vel0 = _state(world)
        # the prey of the *_MovingTarget files has no leader_force until the first World.step (core.py:156)
        out['lforce0'].append(np.stack([np.asarray(getattr(a, 'leader_force', np.zeros(2)), float) for a in world.agents]))
        arng = np.random.RandomState(20_000 + seed)
        ep = {k: [] for k in ('act', 'pos', 'vel', 'lforce', 'obs', 'rew', 'info')}
        for t in range(T):
            parts = []
            for n in act_dims:
                if actions == 'rand':
                    parts.append(arng.rand(n))
                else:                                  # softmax per sub-space (what MADDPG emits)
                    v = []
                    for lo, hi in ((0, 5), (5, n)):
                        if hi > lo:
                            z = arng.randn(hi - lo) * 2.0
                            e = np.exp(z - z.max())
                            v.append(e / e.sum())
                    parts.append(np.concatenate(v))
            if force_discrete:      # environment.py:184-187 on a MultiDiscrete action: only action[0], the physical part
                env.force_discrete_action = True
                parts = [np.concatenate([np.round(p[:5] * 3) / 3, p[5:]]) for p in parts]     # coarse grid: ties
            obs_n, rew_n, done_n, info_n = env.step([p.copy() for p in parts])
            pos, vel = _state(world)
            ep['act'].append(np.concatenate(parts))
            ep['pos'].append(pos.copy())
            ep['vel'].append(vel.copy())
            ep['lforce'].append(np.stack([np.asarray(a.leader_force, float).copy() for a in world.agents]))
            ep['obs'].append(np.concatenate(obs_n))
            ep['rew'].append(np.asarray(rew_n, dtype=np.float64))
            ep['info'].append(np.stack([_info_row(x) for x in info_n['n']]))
        out['pos0'].append(pos0)
        out['vel0'].append(vel0)
        out['obs0'].append(np.concatenate(obs0))
        for k in ep:
            out[k].append(np.stack(ep[k]))
    res = {k: np.stack(v) for k, v in out.items()}
    res['seeds'] = np.asarray(list(seeds))
    res['act_dims'] = np.asarray(act_dims)
    # what make_world() set up, read back from the reference's world (the oracle spec is built from these)
    ents, ags = world.entities, world.agents
    res['spec_n_agents'] = np.int64(len(ags))
    res['spec_dim_c'] = np.int64(world.dim_c)
    res['spec_size'] = np.array([e.size for e in ents], float)
    res['spec_collide'] = np.array([bool(e.collide) for e in ents])
    res['spec_movable'] = np.array([bool(e.movable) for e in ents])
    res['spec_accel'] = np.array([a.accel for a in ags], float)
    res['spec_max_speed'] = np.array([a.max_speed for a in ags], float)
    res['spec_silent'] = np.array([bool(a.silent) for a in ags])
    res['spec_adversary'] = np.array([bool(a.adversary) for a in ags])
    res['spec_force_id'] = np.array([a.force_id for a in ags], np.int64)
    res['spec_forced_id'] = np.array([a.forced_id for a in ags], np.int64)
    res['spec_advleader'] = np.array([bool(a.advleader) for a in ags])
    res['spec_advsuperleader'] = np.array([bool(a.advsuperleader) for a in ags])
    res['spec_obs_r'] = np.array([a.obs_r for a in ags], float)
    res['spec_reward_variant'] = np.int64(1 if 'rewardFixed' in name or name == 'S1L2F4_MovingTarget' else 0)
    res['spec_moving_target'] = np.bool_('MovingTarget' in name)
    return res


def _shim(world):
    for a in world.agents:
        a.forced_id = -1
        a.force_id = -1
        a.advsuperleader = False


def build_env(kind, seed, scaled=None, u_noise=None):
    """kind in {'simple_tag', 'simple_spread'}; scaled=(n_adv, n_good, n_landmarks)."""
    np.random.seed(seed)
    if kind == 'simple_tag':
        S = TAG.Scenario()
        if scaled is None:
            world = S.make_world()
        else:
            n_adv, n_good, n_lm = scaled
            world = World()
            world.dim_c = 2
            world.agents = [Agent() for _ in range(n_adv + n_good)]
            for i, ag in enumerate(world.agents):
                ag.name = 'agent %d' % i
                ag.collide = True
                ag.silent = True
                ag.adversary = i < n_adv
                ag.size = 0.075 if ag.adversary else 0.05
                ag.accel = 3.0 if ag.adversary else 4.0
                ag.max_speed = 1.0 if ag.adversary else 1.3
            world.landmarks = [Landmark() for _ in range(n_lm)]
            for i, lm in enumerate(world.landmarks):
                lm.name = 'landmark %d' % i
                lm.collide = True
                lm.movable = False
                lm.size = 0.2
                lm.boundary = False
            S.reset_world(world)
    elif kind == 'simple_spread':
        S = SPREAD.Scenario()
        world = S.make_world()
    else:
        S = OTHER[kind].Scenario()
        world = S.make_world()
    _shim(world)
    if u_noise is not None:
        for a in world.agents:
            a.u_noise = u_noise
    env = MultiAgentEnv(world, S.reset_world, S.reward, S.observation, getattr(S, 'benchmark_data', None))
    return env, world


def _state(world):
    pos = np.stack([e.state.p_pos for e in world.entities]).astype(np.float64)
    vel = np.stack([a.state.p_vel for a in world.agents]).astype(np.float64)
    return pos, vel


def _info_row(x, width=None):
    if isinstance(x, dict):                       # no benchmark_data callback (environment.py:133-134)
        return np.zeros(0)
    r = np.atleast_1d(np.asarray(x, dtype=np.float64))
    if width is not None and r.shape[0] < width:  # simple_adversary: scalar for the adversary, tuple for the others
        r = np.concatenate([r, np.zeros(width - r.shape[0])])
    return r


def _goal_index(world):
    g = getattr(world.agents[0], 'goal_a', None)
    return -1 if g is None else world.landmarks.index(g)


def rollout(kind, seeds, T, scaled=None, u_noise=None, init=None, actions='rand'):
    """Returns dict of stacked arrays over seeds."""
    keys = ('pos0', 'vel0', 'obs0', 'goal', 'act', 'noise', 'pos', 'vel', 'obs', 'rew', 'info')
    out = {k: [] for k in keys}
    for seed in seeds:
        env, world = build_env(kind, seed, scaled, u_noise)
        A = len(world.agents)
        obs0 = env.reset()
        if init is not None:
            init(world, np.random.RandomState(10_000 + seed))
            obs0 = [env._get_obs(a) for a in env.agents]
        pos0, vel0 = _state(world)
        out['goal'].append(_goal_index(world))
        arng = np.random.RandomState(20_000 + seed)
        ep = {k: [] for k in ('act', 'noise', 'pos', 'vel', 'obs', 'rew', 'info')}
        for t in range(T):
            if actions == 'rand':
                act = arng.rand(A, 5)
            elif actions == 'softmax':
                logits = arng.randn(A, 5)
                e = np.exp(logits - logits.max(axis=1, keepdims=True))
                act = e / e.sum(axis=1, keepdims=True)
            elif actions == 'index':
                # env.discrete_action_input (environment.py:176-182): one integer per agent; 5 and 7 are outside
                # Discrete(5) and decode to no force at all
                act = arng.randint(0, 8, size=(A, 1)).astype(np.float64) * np.ones((1, 5))
            elif actions == 'force':
                # env.force_discrete_action (environment.py:184-187): values on a coarse grid, so that ties occur and
                # np.argmax's first-maximum rule matters
                act = arng.randint(0, 4, size=(A, 5)) / 4.0
            elif actions in ('continuous', 'continuous_force'):
                # env.discrete_action_space = False (environment.py:191-192): the action IS u (dim_p = 2 entries), kept in
                # columns 0, 1 of 'act'; 'continuous_force': values on a coarse grid (ties) + force_discrete_action
                act = np.zeros((A, 5))
                act[:, :2] = (arng.randint(-2, 3, size=(A, 2)) / 2.0 if actions == 'continuous_force'
                              else arng.uniform(-1.0, 1.0, size=(A, 2)))
            else:
                act = np.zeros((A, 5))
            s_t = 30_000 + 1000 * seed + t
            np.random.seed(s_t)
            noise = np.random.RandomState(s_t).randn(A, 2)
            if actions == 'index':
                env.discrete_action_input = True
                obs_n, rew_n, done_n, info_n = env.step([int(act[i, 0]) for i in range(A)])
            elif actions in ('continuous', 'continuous_force'):
                env.discrete_action_space = False         # flipped after construction: the spaces stay Discrete(5)
                env.force_discrete_action = actions == 'continuous_force'
                obs_n, rew_n, done_n, info_n = env.step([act[i, :2].copy() for i in range(A)])   # scaled in place
            else:
                env.force_discrete_action = actions == 'force'
                obs_n, rew_n, done_n, info_n = env.step([act[i].copy() for i in range(A)])
            assert not any(done_n)
            pos, vel = _state(world)
            ep['act'].append(act)
            ep['noise'].append(noise)
            ep['pos'].append(pos.copy())
            ep['vel'].append(vel.copy())
            ep['obs'].append(np.concatenate(obs_n))
            ep['rew'].append(np.asarray(rew_n, dtype=np.float64))
            rows = [_info_row(x) for x in info_n['n']]
            width = max(r.shape[0] for r in rows)
            ep['info'].append(np.stack([_info_row(x, width) for x in info_n['n']]))
        out['pos0'].append(pos0)
        out['vel0'].append(vel0)
        out['obs0'].append(np.concatenate(obs0))
        for k in ep:
            out[k].append(np.stack(ep[k]))
    res = {k: (np.asarray(v) if k == 'goal' else np.stack(v)) for k, v in out.items()}
    res['seeds'] = np.asarray(list(seeds))
    res['u_noise'] = np.float64(0.0 if u_noise is None else u_noise)
    if u_noise is None:
        res.pop('noise')
    return res


def _toward(target_of, gain, cap):
    """A scripted agent's action_callback (core.py:83,126-127): a capped proportional controller towards the entity
    target_of(world) returns.  The returned Action's u is used as the force as it stands (core.py:143-149: no
    sensitivity, that is applied by the env's decode of POLICY agents only, environment.py:193-196)."""
    from multiagent.core import Action

    def cb(agent, world):
        a = Action()
        a.u = np.clip(gain * (target_of(world).state.p_pos - agent.state.p_pos), -cap, cap)
        a.c = np.zeros(world.dim_c)
        return a
    return cb


SCRIPTED = {
    # name: (kind, {agent index: callback}, u_noise, init)
    's1_simple_tag_scripted_prey': ('simple_tag', lambda: {3: _toward(lambda w: w.landmarks[0], 2.0, 4.0)}, None),
    's2_simple_tag_scripted_chasers': ('simple_tag', lambda: {0: _toward(lambda w: w.agents[3], 4.0, 3.0),
                                                             2: _toward(lambda w: w.agents[3], 1.5, 3.0)}, 0.3),
}


def rollout_scripted(name, seeds, T, init=None):
    """World.step's scripted-agent loop (core.py:126-127): some agents carry an action_callback, the env then only
    knows the others (world.policy_agents: n, spaces, action_n, obs_n, reward_n, info_n all skip the scripted ones,
    environment.py:21-22,96-110).  'act' holds the policy agents' actions at their WORLD index (zeros elsewhere),
    'u_scripted' what the callbacks returned; obs / rew / info are the env's lists, i.e. policy agents only."""
    kind, mk, u_noise = SCRIPTED[name]
    keys = ('pos0', 'vel0', 'obs0', 'act', 'noise', 'u_scripted', 'pos', 'vel', 'obs', 'rew', 'info')
    out = {k: [] for k in keys}
    for seed in seeds:
        np.random.seed(seed)
        S = TAG.Scenario()
        world = S.make_world()
        _shim(world)
        cbs = mk()
        for i, cb in cbs.items():
            world.agents[i].action_callback = cb
        if u_noise is not None:
            for a in world.agents:
                a.u_noise = u_noise
        env = MultiAgentEnv(world, S.reset_world, S.reward, S.observation, S.benchmark_data)
        A = len(world.agents)
        policy = [i for i in range(A) if i not in cbs]
        assert env.n == len(policy) and len(env.action_space) == len(policy)
        obs0 = env.reset()
        if init is not None:
            init(world, np.random.RandomState(10_000 + seed))
            obs0 = [env._get_obs(a) for a in env.agents]
        pos0, vel0 = _state(world)
        arng = np.random.RandomState(20_000 + seed)
        ep = {k: [] for k in ('act', 'noise', 'u_scripted', 'pos', 'vel', 'obs', 'rew', 'info')}
        for t in range(T):
            logits = arng.randn(A, 5)
            e = np.exp(logits - logits.max(axis=1, keepdims=True))
            act = e / e.sum(axis=1, keepdims=True)
            act[list(cbs)] = 0.0
            s_t = 30_000 + 1000 * seed + t
            np.random.seed(s_t)
            noise = np.random.RandomState(s_t).randn(A, 2)
            obs_n, rew_n, done_n, info_n = env.step([act[i].copy() for i in policy])
            assert len(obs_n) == len(rew_n) == len(info_n['n']) == len(policy)
            pos, vel = _state(world)
            u = np.zeros((A, 2))
            for i in cbs:
                u[i] = world.agents[i].action.u        # what World.step used (before noise)
            ep['act'].append(act)
            ep['noise'].append(noise)
            ep['u_scripted'].append(u)
            ep['pos'].append(pos.copy())
            ep['vel'].append(vel.copy())
            ep['obs'].append(np.concatenate(obs_n))
            ep['rew'].append(np.asarray(rew_n, dtype=np.float64))
            ep['info'].append(np.stack([_info_row(x) for x in info_n['n']]))
        out['pos0'].append(pos0)
        out['vel0'].append(vel0)
        out['obs0'].append(np.concatenate(obs0))
        for k in ep:
            out[k].append(np.stack(ep[k]))
    res = {k: np.stack(v) for k, v in out.items()}
    res['seeds'] = np.asarray(list(seeds))
    res['scripted'] = np.asarray(sorted(cbs), np.int64)
    res['u_noise'] = np.float64(0.0 if u_noise is None else u_noise)
    return res


def rollout_shared_done(seeds, T):
    """environment.py:33,113-115 + :146-149: simple_spread with world.collaborative = True (upstream's simple_spread sets it,
    this fork's copy does not: every agent then receives the SUM of the rewards) and a done_callback (an agent is done
    once it is within 0.25 of some landmark).  Dense starts, so that both fire."""
    keys = ('pos0', 'vel0', 'act', 'pos', 'vel', 'obs', 'rew', 'done')
    out = {k: [] for k in keys}
    for seed in seeds:
        np.random.seed(seed)
        S = SPREAD.Scenario()
        world = S.make_world()
        _shim(world)
        world.collaborative = True

        def done(agent, world):
            return bool(min(np.sqrt(np.sum(np.square(agent.state.p_pos - l.state.p_pos))) for l in world.landmarks) < 0.25)
        env = MultiAgentEnv(world, S.reset_world, S.reward, S.observation, S.benchmark_data, done_callback=done)
        assert env.shared_reward
        A = len(world.agents)
        env.reset()
        dense_init(world, np.random.RandomState(10_000 + seed))
        pos0, vel0 = _state(world)
        arng = np.random.RandomState(20_000 + seed)
        ep = {k: [] for k in ('act', 'pos', 'vel', 'obs', 'rew', 'done')}
        for t in range(T):
            act = arng.rand(A, 5)
            obs_n, rew_n, done_n, info_n = env.step([act[i].copy() for i in range(A)])
            pos, vel = _state(world)
            ep['act'].append(act)
            ep['pos'].append(pos.copy())
            ep['vel'].append(vel.copy())
            ep['obs'].append(np.concatenate(obs_n))
            ep['rew'].append(np.asarray(rew_n, dtype=np.float64))
            ep['done'].append(np.asarray(done_n, dtype=bool))
        out['pos0'].append(pos0)
        out['vel0'].append(vel0)
        for k in ep:
            out[k].append(np.stack(ep[k]))
    res = {k: np.stack(v) for k, v in out.items()}
    res['seeds'] = np.asarray(list(seeds))
    return res


def scripted_cases():
    save('g17_simple_spread_shared_reward_done', **rollout_shared_done(range(16), 25))
    for name in SCRIPTED:
        save(name, **rollout_scripted(name, range(16), 25))
        save(name + '_dense', **rollout_scripted(name, range(16), 25, init=dense_init))


COMM = {k: _load(REF + '/build/lib/multiagent/scenarios/%s.py' % k, 'ref_' + k)
        for k in ('simple_speaker_listener', 'simple_reference', 'simple_crypto',
                  'simple_world_comm')}                                                    # SURVEY 8f row f1, non-silent half


def rollout_comm(kind, seeds, T, init=None, c_noise=None):
    """simple_speaker_listener / simple_reference: agents that speak (comm entries in the action,
    environment.py:198-206; a non-movable agent has no physical entries, :174) and per-agent goal
    landmarks (agent.goal_b).  Actions are stored concatenated agent by agent (act_dims); `comm` is
    agent.state.c after each step; `goal` the landmark index of every agent's goal_b (-1: none)."""
    keys = ('pos0', 'vel0', 'comm0', 'obs0', 'goal', 'act', 'pos', 'vel', 'comm', 'obs', 'rew', 'info', 'cnoise')
    out = {k: [] for k in keys}
    crypto = kind == 'simple_crypto'       # goal rows: (goal_a landmark, landmark whose colour is agent 2's key)
    wcomm = kind == 'simple_world_comm'    # simple_tag physics + food / forests + a speaking leader; integer info
    for seed in seeds:
        np.random.seed(seed)
        S = COMM[kind].Scenario()
        world = S.make_world()
        _shim(world)
        if c_noise is not None:                # the scenario files carry it commented out (simple_reference.py:16)
            for a in world.agents:
                a.c_noise = c_noise
        # benchmark_data is broken (speaker_listener) / absent (reference); simple_crypto has one
        env = MultiAgentEnv(world, S.reset_world, S.reward, S.observation, S.benchmark_data if crypto or wcomm else None)
        obs0 = env.reset()
        if init is not None:
            init(world, np.random.RandomState(10_000 + seed))
            obs0 = [env._get_obs(a) for a in env.agents]
        act_dims = [(5 if a.movable else 0) + (0 if a.silent else world.dim_c) for a in world.agents]
        pos0, vel0 = _state(world)
        if crypto:
            out['goal'].append([world.landmarks.index(world.agents[0].goal_a), int(np.argmax(world.agents[2].key))])
        else:
            out['goal'].append([-1 if getattr(a, 'goal_b', None) is None else world.landmarks.index(a.goal_b)
                                for a in world.agents])
        out['comm0'].append(np.stack([np.asarray(a.state.c, float) for a in world.agents]))
        arng = np.random.RandomState(40_000 + seed)
        ep = {k: [] for k in ('act', 'pos', 'vel', 'comm', 'obs', 'rew', 'info', 'cnoise')}
        for t in range(T):
            parts = [arng.rand(n) for n in act_dims]
            # core.py:229 draws randn(dim_c) per speaking agent, in agent order, from the global stream
            s_t = 50_000 + 1000 * seed + t
            np.random.seed(s_t)
            speaking = [i for i, a in enumerate(world.agents) if not a.silent]
            cn = np.zeros((len(world.agents), world.dim_c))
            if c_noise is not None:
                cn[speaking] = np.random.RandomState(s_t).randn(len(speaking), world.dim_c)
            ep['cnoise'].append(cn)
            if crypto and t % 5 == 4:          # an all-zero message switches the reward terms off (:104,109,119)
                parts[(t // 5) % 2] = np.zeros(act_dims[0])
            if force_discrete:      # environment.py:184-187 on a MultiDiscrete action: only action[0], the physical part
                env.force_discrete_action = True
                parts = [np.concatenate([np.round(p[:5] * 3) / 3, p[5:]]) for p in parts]     # coarse grid: ties
            obs_n, rew_n, done_n, info_n = env.step([p.copy() for p in parts])
            pos, vel = _state(world)
            ep['act'].append(np.concatenate(parts))
            ep['pos'].append(pos.copy())
            ep['vel'].append(vel.copy())
            ep['comm'].append(np.stack([np.asarray(a.state.c, float).copy() for a in world.agents]))
            ep['obs'].append(np.concatenate(obs_n))
            ep['rew'].append(np.asarray(rew_n, dtype=np.float64))
            ep['info'].append(np.stack([np.concatenate(x) for x in info_n['n']]) if crypto
                              else np.asarray(info_n['n'], float).reshape(-1, 1) if wcomm else np.zeros((len(act_dims), 0)))
        out['pos0'].append(pos0)
        out['vel0'].append(vel0)
        out['obs0'].append(np.concatenate(obs0))
        for k in ep:
            out[k].append(np.stack(ep[k]))
    res = {k: np.asarray(v) if k == 'goal' else np.stack(v) for k, v in out.items()}
    if not (crypto or wcomm):
        res.pop('info')
    if c_noise is None:
        res.pop('cnoise')
    res['spec_c_noise'] = np.float64(0.0 if c_noise is None else c_noise)
    if wcomm:
        res.pop('goal')
        res['spec_n_food'], res['spec_n_forests'] = np.int64(len(world.food)), np.int64(len(world.forests))
    res['seeds'] = np.asarray(list(seeds))
    res['act_dims'] = np.asarray(act_dims)
    res['spec_dim_c'] = np.int64(world.dim_c)
    res['spec_size'] = np.array([e.size for e in world.entities], float)
    res['spec_movable'] = np.array([bool(e.movable) for e in world.entities])
    res['spec_collide'] = np.array([bool(e.collide) for e in world.entities])
    res['spec_silent'] = np.array([bool(a.silent) for a in world.agents])
    res['spec_shared_reward'] = np.bool_(env.shared_reward)
    return res


def dense_init(world, rng):
    for e in world.entities:
        e.state.p_pos = rng.uniform(-0.3, 0.3, 2)
    for a in world.agents:
        a.state.p_vel = rng.uniform(-0.5, 0.5, 2)


def crowd_init(world, rng):
    """Everybody on top of everybody: fires the count >= 4 bonus of the *_MovingTarget reward."""
    for e in world.entities:
        e.state.p_pos = rng.uniform(-0.03, 0.03, 2)


def oob_init(world, rng):
    mags = [0.85, 0.95, 1.5, 2.5]
    for a in world.agents:
        sx, sy = rng.choice([-1.0, 1.0], 2)
        mx, my = rng.choice(mags, 2)
        a.state.p_pos = np.array([sx * mx, sy * my]) + rng.uniform(-0.02, 0.02, 2)


def save(name, **arrs):
    path = os.path.join(OUT, name + '.npz')
    np.savez_compressed(path, **arrs)
    print('%-28s %8.1f KiB' % (name, os.path.getsize(path) / 1024))


def kat_cases():
    # KAT-1: one-sided contact (core.py:203-207)
    env, world = build_env('simple_tag', 0)
    env.reset()
    far = [(5.0, 5.0), (-5.0, 5.0), (5.0, -5.0), (-5.0, -5.0)]
    for i, e in enumerate(world.entities):
        e.state.p_pos = np.array(far[i % 4]) * (1 + i)
    world.agents[0].state.p_pos = np.array([0.0, 0.0])
    world.agents[1].state.p_pos = np.array([0.1, 0.0])
    pos0, vel0 = _state(world)
    z = [np.zeros(5) for _ in world.agents]
    obs_n, rew_n, _, info_n = env.step(z)
    pos1, vel1 = _state(world)
    save('kat1_one_sided', pos0=pos0, vel0=vel0, pos=pos1, vel=vel1,
         obs=np.concatenate(obs_n), rew=np.asarray(rew_n, float),
         info=np.stack([_info_row(x) for x in info_n['n']]))
    # KAT-2: spread self-collision term (simple_spread.py:78-80)
    env, world = build_env('simple_spread', 0)
    env.reset()
    for i in range(3):
        world.landmarks[i].state.p_pos = np.array([float(i), 0.0])
        world.agents[i].state.p_pos = np.array([float(i), 0.0])
    pos0, vel0 = _state(world)
    rew = np.array([env._get_reward(a) for a in env.agents], float)
    info = np.stack([_info_row(env._get_info(a)) for a in env.agents])
    save('kat2_spread_self', pos0=pos0, vel0=vel0, rew=rew, info=info)
    # KAT-4: far-apart entities, zero action: pos bitwise unchanged, tiny vel
    env, world = build_env('simple_tag', 0)
    env.reset()
    for i, e in enumerate(world.entities):
        e.state.p_pos = np.array([-0.9 + 0.6 * (i % 3), -0.5 + 1.0 * (i // 3)])
    pos0, vel0 = _state(world)
    env.step([np.zeros(5) for _ in world.agents])
    pos1, vel1 = _state(world)
    save('kat4_far_zero', pos0=pos0, vel0=vel0, pos=pos1, vel=vel1)


def main():
    save('g1_simple_tag', **rollout('simple_tag', range(64), 25))
    save('g2_simple_spread', **rollout('simple_spread', range(64), 25))
    save('g3_scaled_tag', **rollout('simple_tag', range(2), 12, scaled=(48, 16, 32),
                                    init=None, actions='softmax'))
    save('g3d_scaled_tag_dense', **rollout('simple_tag', range(1), 6, scaled=(48, 16, 32),
                                           init=dense_init, actions='softmax'))
    save('g4_simple_tag_unoise', **rollout('simple_tag', range(16), 25, u_noise=0.3))
    save('g5_simple_tag_dense', **rollout('simple_tag', range(32), 25, init=dense_init,
                                          actions='softmax'))
    save('g5_simple_spread_dense', **rollout('simple_spread', range(32), 25, init=dense_init,
                                             actions='softmax'))
    save('g6_simple_tag_oob', **rollout('simple_tag', range(24), 10, init=oob_init,
                                        actions='rand'))
    # the remaining all-silent stock scenarios (same physics, new gathers)
    save('g7_simple', **rollout('simple', range(32), 25))
    save('g8_simple_adversary', **rollout('simple_adversary', range(48), 25))
    save('g8_simple_adversary_dense', **rollout('simple_adversary', range(16), 25, init=dense_init,
                                                actions='softmax'))
    save('g9_simple_push', **rollout('simple_push', range(24), 25))
    save('g9_simple_push_dense', **rollout('simple_push', range(16), 25, init=dense_init,
                                           actions='softmax'))
    save('g10_simple_speaker_listener', **rollout_comm('simple_speaker_listener', range(48), 25))
    save('g11_simple_reference', **rollout_comm('simple_reference', range(48), 25))
    save('g12_simple_crypto', **rollout_comm('simple_crypto', range(48), 25))
    # c_noise (core.py:229): no scenario file sets it (simple_reference.py:16 has it commented out)
    save('g14_simple_reference_cnoise', **rollout_comm('simple_reference', range(16), 25, c_noise=0.1))
    save('g14_simple_crypto_cnoise', **rollout_comm('simple_crypto', range(16), 25, c_noise=0.25))
    save('g14_simple_world_comm_cnoise', **rollout_comm('simple_world_comm', range(16), 25, c_noise=0.1))
    save('g13_simple_world_comm', **rollout_comm('simple_world_comm', range(48), 25))
    save('g13_simple_world_comm_dense', **rollout_comm('simple_world_comm', range(24), 25, init=dense_init))
    # the fork's leader / follower / super-leader pursuit family
    for name, n in (('L1F2_3Targets_100observe', 32), ('S1L2F2_3Targets_100observe', 32),
                    ('L1F2_3Targets_075observe', 8), ('L1F1_10Targets_wide', 16), ('L1F4_3Targets_wide', 16),
                    ('S1L2F2_10Targets_wide', 16), ('S1L2F2_10Targets_wide_rewardFixed', 16)):
        save('h_' + name, **rollout_hier(name, range(n), 25, actions='softmax'))
        save('h_' + name + '_dense', **rollout_hier(name, range(n // 2), 25, init=dense_init))
    # the two *_MovingTarget files: a moving prey instead of landmarks, own reward / observation text
    for name in ('L1F4_MovingTarget', 'S1L2F4_MovingTarget'):
        save('m_' + name, **rollout_hier(name, range(24), 25, actions='softmax'))
        save('m_' + name + '_dense', **rollout_hier(name, range(24), 25, init=dense_init))
        save('m_' + name + '_crowd', **rollout_hier(name, range(16), 6, init=crowd_init))
    hier_rest()
    movable_cases()
    discrete_cases()
    scripted_cases()
    kat_cases()


# The nine hierarchy files that round 1 covered only through hierarchy.PRESETS (they differ from the files above in
# counts, zoom rate and observation radius): short fixtures, enough to pin the world constants read back from the
# reference's make_world() and a few steps of the dynamics, rewards and masked / sorted observations.
HIER_REST = (('L1F2_10Targets_wide', 8, 25), ('S1L1F1_10Targets_wide', 8, 25),
             ('L1F2_3Targets_090observe', 4, 5), ('L1F2_3Targets_095observe', 4, 5), ('L1F2_3Targets_105observe', 4, 5),
             ('S1L2F2_3Targets_075observe', 4, 5), ('S1L2F2_3Targets_090observe', 4, 5),
             ('S1L2F2_3Targets_095observe', 4, 5), ('S1L2F2_3Targets_105observe', 4, 5))


def discrete_cases():
    """The two non-default decodes of MultiAgentEnv._set_action.  'act' holds what was passed to env.step: for the
    integer mode the integer repeated over the 5 columns, for force_discrete_action the vector before the argmax."""
    save('g15_simple_tag_index_actions', **rollout('simple_tag', range(16), 25, actions='index'))
    save('g15_simple_spread_index_actions', **rollout('simple_spread', range(8), 25, init=dense_init, actions='index'))
    save('g16_simple_tag_force_discrete', **rollout('simple_tag', range(16), 25, actions='force'))
    save('g16_simple_push_force_discrete', **rollout('simple_push', range(8), 25, init=dense_init, actions='force'))
    save('g16_h_S1L2F2_3Targets_100observe_force_discrete',
         **rollout_hier('S1L2F2_3Targets_100observe', range(8), 25, init=dense_init, force_discrete=True))
    continuous_cases()


def rollout_movable_landmarks(kind, seeds, T, tweak, init=None):
    """A stock world in which `tweak(world)` made some landmarks movable (Entity.movable / mass / max_speed, core.py:36-53;
    no scenario file does): core.py:211-222 integrates every movable entity, so the landmarks get a velocity ('lvel',
    recorded per step), are damped, and are pushed by the landmarks behind them (one-sided response, core.py:199-208)."""
    keys = ('pos0', 'vel0', 'lvel0', 'act', 'pos', 'vel', 'lvel', 'obs', 'rew')
    out = {k: [] for k in keys}
    for seed in seeds:
        env, world = build_env(kind, seed)
        tweak(world)
        A = len(world.agents)
        env.reset()
        if init is not None:
            init(world, np.random.RandomState(10_000 + seed))
        lv = np.random.RandomState(40_000 + seed).uniform(-0.5, 0.5, size=(len(world.landmarks), 2))
        for lm, v in zip(world.landmarks, lv):
            lm.state.p_vel = v.copy() if lm.movable else np.zeros(2)
        pos0, vel0 = _state(world)
        out['lvel0'].append(np.stack([lm.state.p_vel for lm in world.landmarks]).astype(np.float64))
        arng = np.random.RandomState(20_000 + seed)
        ep = {k: [] for k in ('act', 'pos', 'vel', 'lvel', 'obs', 'rew')}
        for t in range(T):
            act = arng.rand(A, 5)
            obs_n, rew_n, done_n, info_n = env.step([act[i].copy() for i in range(A)])
            pos, vel = _state(world)
            ep['act'].append(act)
            ep['pos'].append(pos.copy())
            ep['vel'].append(vel.copy())
            ep['lvel'].append(np.stack([lm.state.p_vel for lm in world.landmarks]).astype(np.float64))
            ep['obs'].append(np.concatenate(obs_n))
            ep['rew'].append(np.asarray(rew_n, dtype=np.float64))
        out['pos0'].append(pos0)
        out['vel0'].append(vel0)
        for k in ep:
            out[k].append(np.stack(ep[k]))
    res = {k: np.stack(v) for k, v in out.items()}
    res['seeds'] = np.asarray(list(seeds))
    return res


def _tag_movable(world):          # both landmarks move; the second one is heavy and speed-limited
    world.landmarks[0].movable = True
    world.landmarks[1].movable = True
    world.landmarks[1].initial_mass = 2.5
    world.landmarks[1].max_speed = 0.08


def _spread_movable(world):       # landmarks 0 and 2 move and (unlike the stock file) collide: 0 is pushed by 1 and 2
    for i in (0, 2):
        world.landmarks[i].movable = True
    for lm in world.landmarks:
        lm.collide = True
        lm.size = 0.12
    world.landmarks[2].max_speed = 0.3


def movable_cases():
    save('g19_simple_tag_movable_landmarks', **rollout_movable_landmarks('simple_tag', range(16), 25, _tag_movable, init=dense_init))
    save('g19_simple_spread_movable_landmarks', **rollout_movable_landmarks('simple_spread', range(16), 25, _spread_movable,
                                                                           init=dense_init))


def continuous_cases():
    """env.discrete_action_space = False (environment.py:191-192, "NEVER do here" in the fork's comments but a live
    switch): 'act'[:, :, :, 0:2] holds the force command u that was passed to env.step."""
    save('g18_simple_tag_continuous', **rollout('simple_tag', range(16), 25, actions='continuous'))
    save('g18_simple_spread_continuous', **rollout('simple_spread', range(8), 25, init=dense_init, actions='continuous'))
    save('g18_simple_tag_continuous_force', **rollout('simple_tag', range(8), 25, actions='continuous_force'))


def hier_rest():
    for name, n, T in HIER_REST:
        save('h_' + name, **rollout_hier(name, range(n), T, actions='softmax'))
        save('h_' + name + '_dense', **rollout_hier(name, range(max(2, n // 2)), T, init=dense_init))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'hier_rest':     # only the fixtures added in round 2
        hier_rest()
    elif len(sys.argv) > 1 and sys.argv[1] == 'discrete':
        discrete_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == 'continuous':
        continuous_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == 'movable':
        movable_cases()
    elif len(sys.argv) > 1 and sys.argv[1] == 'scripted':
        scripted_cases()
    else:
        main()
