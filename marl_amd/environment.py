"""Batched MultiAgentEnv: the reference's gym-style env API (multiagent/environment.py:9-154) over
``num_envs`` instances living on one MI355X.

    env = make_env('simple_tag', benchmark=True, num_envs=1 << 20, dtype=torch.float32)
    obs_n = env.reset()                       # list of n tensors [B, obs_dim_i]
    obs_n, reward_n, done_n, info_n = env.step(action_n)

Agents with an ``action_callback`` are scripted (core.py:83,126-127): ``World.step`` asks the callback for their action
and the env's ``n``, spaces, ``action_n``, ``obs_n``, ``reward_n`` and ``info_n`` cover the other agents only
(``world.policy_agents``); set the callbacks before constructing the env.

``action_n``: list of n tensors ``[B, 5]`` (soft one-hot Discrete(5), environment.py:188-190; ``[B, 5 +
dim_c]`` for a non-silent agent, whose space is MultiDiscrete: 5 physical entries then the dim_c comm
entries, environment.py:161-168), one tensor ``[B, n, 5]``, or one contiguous tensor ``[n, 5, B]`` /
``[act_total, B]`` (the kernel's SoA layout; used in place when it has the world's padded row stride,
i.e. comes from ``env.world.new_action_buffer()``).  Policies can write straight into
``env.action_views[i]`` (``[B, act_dim_i]`` views of the kernel's SoA action buffer) and pass that
list: then step() copies nothing.
Returned tensors are views of buffers that the next step()/reset() overwrites.

When the callbacks are the methods of a built-in scenario (simple_tag / simple_spread) the whole
step is one fused HIP launch; with user callbacks World.step() is one launch and the callbacks
run as given, agent by agent, exactly like environment.py:105-110.
"""
import numpy as np
import torch

from . import _hip, spaces
from .scenario import HipScenario


class MultiAgentEnv(object):
    metadata = {'render.modes': []}

    def __init__(self, world, reset_callback=None, reward_callback=None,
                 observation_callback=None, info_callback=None,
                 done_callback=None, shared_viewer=True, observation_radius=1):
        self.world = world
        self.agents = self.world.policy_agents
        self.n = len(world.policy_agents)
        self.num_envs = world.num_envs
        self.reset_callback = reset_callback
        self.reward_callback = reward_callback
        self.observation_callback = observation_callback
        self.info_callback = info_callback
        self.done_callback = done_callback
        # environment.py:29-37
        self.discrete_action_space = True
        self.discrete_action_input = False
        self.force_discrete_action = world.discrete_action if hasattr(world, 'discrete_action') else False
        self.shared_reward = world.collaborative if hasattr(world, 'collaborative') else False
        self.time = 0
        self.obs_r = observation_radius
        self._idx = None          # int32 [n, ld]: the integer actions of discrete_action_input
        self._u = None            # [A, 2, ld]: the force commands of discrete_action_space = False

        # fused path: every callback is a built-in scenario's own method
        owners = {getattr(cb, '__self__', None) for cb in (reset_callback, reward_callback, observation_callback)}
        if info_callback is not None:
            owners.add(getattr(info_callback, '__self__', None))
        self._scenario = owners.pop() if len(owners) == 1 else None
        self._fused = (isinstance(self._scenario, HipScenario) and done_callback is None
                       and all(getattr(cb, '__func__', None) is getattr(HipScenario, nm)
                               for cb, nm in ((reward_callback, 'reward'), (observation_callback, 'observation')))
                       and (info_callback is None
                            or getattr(info_callback, '__func__', None) is HipScenario.benchmark_data))

        # configure spaces (environment.py:39-80)
        self.action_space = []
        self.observation_space = []
        for agent in self.agents:
            u_space = spaces.Discrete(world.dim_p * 2 + 1)
            c_space = spaces.Discrete(world.dim_c)
            total = ([u_space] if agent.movable else []) + ([] if agent.silent else [c_space])   # environment.py:43-63
            if len(total) > 1:                        # environment.py:65-73
                self.action_space.append(spaces.MultiDiscrete([[0, sp.n - 1] for sp in total]))
            else:
                self.action_space.append(total[0])    # IndexError for an agent with no action, like the reference
            obs_dim = int(observation_callback(agent, self.world).shape[-1])
            self.observation_space.append(spaces.Box(low=-np.inf, high=+np.inf, shape=(obs_dim,), dtype=np.float32))
        self._own_act = world.act
        self._uniform = world._uniform_actions                              # world.act is [n,5,B]
        # world index of every policy agent (core.py:115-116): scripted agents own rows of the action buffer and of the
        # outputs too, but the env's lists skip them (environment.py:21-22,96-110)
        self._pidx = [a._index for a in self.agents]
        self._all_policy = len(self._pidx) == len(world.agents)
        self.action_views = [world.act_rows[world.act_row0[i]:world.act_row0[i] + world.act_dims[i]].t()
                             for i in self._pidx]                           # [B,act_dim] views, strides (1,B)
        self._done = torch.zeros(self.num_envs, dtype=torch.bool, device=world.device)
        self._empty_info = [{} for _ in range(self.n)]

    # -- environment.py:157-208: the default decode (:188-190) runs inside the step kernels; the two other decodes
    # rewrite the action rows in front of them (mpe_action_from_index_*, mpe_action_force_discrete_*) -----------
    def _set_actions(self, action_n):
        if self.discrete_action_input:               # environment.py:176-182: one integer per agent
            return self._set_index_actions(action_n)
        if not self.discrete_action_space:           # environment.py:191-192: the action is the force command itself
            return self._set_force_actions(action_n)
        self._copy_actions(action_n)
        if self.force_discrete_action:               # environment.py:184-187: argmax -> one-hot, in place like the reference
            w = self.world
            fn = getattr(_hip.lib, 'mpe_action_force_discrete_' + _hip.suffix(w.dtype))
            with _hip.on_device(w.device):
                _hip.check(fn(w.descriptor(), w.num_envs, w.ld, w.act.data_ptr(), _hip.stream_of(w.device)),
                           'mpe_action_force_discrete')

    def _set_index_actions(self, action_n):
        w = self.world
        if any(not a.silent for a in self.agents):
            # environment.py:200-208: the integer decode of a comm action never consumes its entry
            raise AssertionError('discrete_action_input with a speaking agent: the reference fails its own '
                                 '`assert len(action) == 0` (environment.py:208)')
        if self._idx is None:      # one row per WORLD agent (a scripted agent's stays 0; World.step fills its action rows)
            self._idx = torch.zeros(len(w.agents), w.ld, dtype=torch.int32, device=w.device)
        idx = self._idx[:, :self.num_envs]
        if not self._all_policy:
            if torch.is_tensor(action_n) or len(action_n) != self.n:
                raise ValueError('need one action per policy agent (a list)')
            for i, a in zip(self._pidx, action_n):
                idx[i].copy_(torch.as_tensor(a, device=w.device).reshape(-1).expand(self.num_envs))
        elif torch.is_tensor(action_n) and action_n.dim() == 2:
            if tuple(action_n.shape) == (self.num_envs, self.n) and self.num_envs != self.n:
                action_n = action_n.t()
            if tuple(action_n.shape) != (self.n, self.num_envs):
                raise ValueError('integer actions must be [n, num_envs] (or [num_envs, n])')
            idx.copy_(action_n)
        else:
            if len(action_n) != self.n:
                raise ValueError('need one action per agent')
            for i, a in enumerate(action_n):
                idx[i].copy_(torch.as_tensor(a, device=w.device).reshape(-1).expand(self.num_envs))
        if w.act.data_ptr() != self._own_act.data_ptr():
            w.act = self._own_act
        fn = getattr(_hip.lib, 'mpe_action_from_index_' + _hip.suffix(w.dtype))
        with _hip.on_device(w.device):
            _hip.check(fn(w.descriptor(), w.num_envs, w.ld, w.ld, self._idx.data_ptr(), w.act.data_ptr(),
                          _hip.stream_of(w.device)), 'mpe_action_from_index')

    def _set_force_actions(self, action_n):
        """``env.discrete_action_space = False`` (environment.py:29,191-192; the spaces were built Discrete at
        construction, as in the reference, where the switch can only be flipped afterwards): the action of agent i is
        its force command ``u`` -- ``[B, 2]`` per agent (a list), or one ``[B, n, 2]`` / ``[n, 2, B]`` tensor --,
        multiplied by the sensitivity like any other; ``force_discrete_action`` first turns it into
        one_hot(argmax(u)) (:184-187).  The reference scales the CALLER's array in place (``agent.action.u = action[0]``,
        ``u *= sensitivity``); here the caller's tensors are only read."""
        w = self.world
        if any(a.movable and not a.silent for a in w.agents):
            # environment.py:161-168: the first 5 entries of the MultiDiscrete action become u; core.py:216 then adds a
            # 5-vector to the 2-vector velocity
            raise ValueError('discrete_action_space = False with a movable speaking agent: the reference fails in '
                             'integrate_state (core.py:216: operands could not be broadcast together)')
        if self._u is None:
            self._u = torch.zeros(len(w.agents), 2, w.ld, dtype=w.dtype, device=w.device)
        u = self._u[:, :, :self.num_envs]
        if torch.is_tensor(action_n) and action_n.dim() == 3:
            if not self._all_policy:
                raise ValueError('with scripted agents pass one action per POLICY agent (a list, environment.py:96-98)')
            if tuple(action_n.shape) == (self.num_envs, self.n, 2) and action_n.stride() != u.stride():
                action_n = action_n.permute(1, 2, 0)
            if tuple(action_n.shape) != (self.n, 2, self.num_envs):
                raise ValueError('force commands must be [num_envs, n, 2] or [n, 2, num_envs]')
            u.copy_(action_n)
        else:
            if len(action_n) != self.n:
                raise ValueError('need one action per agent')
            for i, a in zip(self._pidx, action_n):
                a = torch.as_tensor(a, dtype=w.dtype, device=w.device)
                if a.shape[-1] != 2:
                    raise ValueError('a force command has dim_p = 2 components, got %s' % (tuple(a.shape),))
                u[i].copy_(a.expand(self.num_envs, 2).t())
        if w.act.data_ptr() != self._own_act.data_ptr():
            w.act = self._own_act
        fn = getattr(_hip.lib, 'mpe_action_from_force_' + _hip.suffix(w.dtype))
        with _hip.on_device(w.device):
            _hip.check(fn(w.descriptor(), w.num_envs, w.ld, w.ld, self._u.data_ptr(), int(bool(self.force_discrete_action)),
                          w.act.data_ptr(), _hip.stream_of(w.device)), 'mpe_action_from_force')

    def _copy_actions(self, action_n):
        act = self.world.act
        if not self._all_policy and torch.is_tensor(action_n):
            raise ValueError('with scripted agents pass one action per POLICY agent (a list, environment.py:96-98)')
        soa = (torch.is_tensor(action_n) and action_n.dim() == act.dim() and tuple(action_n.shape) == tuple(act.shape)
               and action_n.dtype == act.dtype and action_n.device == act.device)
        if soa and act.dim() == 3 and tuple(action_n.shape) == (self.num_envs, self.n, 5):
            # [n, 5, B] and [B, n, 5] coincide (B = n = 5): only a tensor with the world's own row stride
            # (world.new_action_buffer()) is read as the SoA layout, anything else as the reference-like [B, n, 5]
            soa = action_n.stride() == self._own_act.stride()
        if soa:
            # SoA path: an [n, 5, B] ([act_total, B]) tensor is the kernel's own layout.  With the
            # world's row stride (world.new_action_buffer()) it is adopted in place, no copy;
            # env.action_views keep pointing at the env's original buffer.  Any other strides: one
            # row-wise copy into the env's buffer.
            if action_n.stride() == self._own_act.stride():
                self.world.act = action_n
                return
            if act.data_ptr() != self._own_act.data_ptr():
                self.world.act = act = self._own_act
            act.copy_(action_n)
            return
        if act.data_ptr() != self._own_act.data_ptr():
            self.world.act = act = self._own_act
        if torch.is_tensor(action_n) and action_n.dim() == 3:
            if not self._uniform or tuple(action_n.shape) != (self.num_envs, self.n, 5):
                raise ValueError('action tensor must be [num_envs, n, 5] (all agents silent)')
            act.copy_(action_n.permute(1, 2, 0))
            return
        if len(action_n) != self.n:
            raise ValueError('need one action per agent')
        for i, a in enumerate(action_n):
            v = self.action_views[i]
            if torch.is_tensor(a) and a.data_ptr() == v.data_ptr() and a.stride() == v.stride() \
                    and a.shape == v.shape:
                continue                                   # already in place
            a = torch.as_tensor(a, dtype=act.dtype, device=act.device)
            if a.shape[-1] != v.shape[1]:
                raise ValueError('action of agent %d must have %d components (5 physical%s)'
                                 % (i, v.shape[1], '' if v.shape[1] == 5 else ' + dim_c comm'))
            v.copy_(a.expand(self.num_envs, v.shape[1]))

    def step(self, action_n):
        self.agents = self.world.policy_agents
        if len(self.agents) != self.n:
            raise RuntimeError('the set of scripted agents changed: build a new MultiAgentEnv (n and the spaces are fixed '
                               'at construction, environment.py:21-22)')
        self._set_actions(action_n)
        world = self.world
        if self._fused:
            world.env_step(want_info=self.info_callback is not None)
            obs_n = [world.obs_views[i] for i in self._pidx]
            reward_n = [world.rew[i] for i in self._pidx]
            if self.info_callback is not None:
                info_n = {'n': [self._scenario._format_info(world.info[i], a) for i, a in zip(self._pidx, self.agents)]}
            else:
                info_n = {'n': list(self._empty_info)}
        else:
            world.step()
            obs_n, reward_n, info_n = [], [], {'n': []}
            for agent in self.agents:
                obs_n.append(self._get_obs(agent))
                reward_n.append(self._get_reward(agent))
                info_n['n'].append(self._get_info(agent))
        done_n = [self._get_done(agent) for agent in self.agents]
        if self.shared_reward:            # environment.py:113-115
            reward = torch.stack(list(reward_n)).sum(0)
            reward_n = [reward] * self.n
        return obs_n, reward_n, done_n, info_n

    def reset(self):
        self.reset_callback(self.world)
        self.agents = self.world.policy_agents
        if self._fused:
            w = self.world
            w.gather_obs()
            return [w.obs_views[i] for i in self._pidx]
        return [self._get_obs(agent) for agent in self.agents]

    # -- environment.py:131-154 ------------------------------------------------------------------------
    def _get_info(self, agent):
        if self.info_callback is None:
            return {}
        return self.info_callback(agent, self.world)

    def _get_obs(self, agent):
        if self.observation_callback is None:
            return torch.zeros(self.num_envs, 0, dtype=self.world.dtype, device=self.world.device)
        return self.observation_callback(agent, self.world)

    def _get_done(self, agent):
        if self.done_callback is None:
            return self._done
        return self.done_callback(agent, self.world)

    def _get_reward(self, agent):
        if self.reward_callback is None:
            return torch.zeros(self.num_envs, dtype=self.world.dtype, device=self.world.device)
        return self.reward_callback(agent, self.world)

    def render(self, mode='human'):
        raise NotImplementedError('rendering (environment.py:216-298) is out of scope for the batched simulator')


class BatchMultiAgentEnv(object):
    """multiagent/environment.py:326-373: the reference's own wrapper around a list of envs, flat concatenation of the
    per-env lists (``n`` = sum of the envs' ``n``; spaces = the first env's).  Here every member is itself a batched
    MultiAgentEnv, so the wrapper only matters to callers that drive several worlds (different scenarios or seeds)
    through one ``step``.  The reference's ``step(action_n, time)`` forwards ``time`` to ``MultiAgentEnv.step``, which
    does not take it (a TypeError there); it is accepted and ignored here.  Like the reference, ``info_n`` comes back
    empty (:360) and nothing is rendered."""
    metadata = {'runtime.vectorized': True, 'render.modes': []}

    def __init__(self, env_batch):
        self.env_batch = env_batch

    @property
    def n(self):
        return int(np.sum([env.n for env in self.env_batch]))

    @property
    def action_space(self):
        return self.env_batch[0].action_space

    @property
    def observation_space(self):
        return self.env_batch[0].observation_space

    def step(self, action_n, time=None):
        obs_n, reward_n, done_n, info_n = [], [], [], {'n': []}
        i = 0
        for env in self.env_batch:
            obs, reward, done, _ = env.step(action_n[i:(i + env.n)])
            i += env.n
            obs_n += obs
            reward_n += reward
            done_n += done
        return obs_n, reward_n, done_n, info_n

    def reset(self):
        obs_n = []
        for env in self.env_batch:
            obs_n += env.reset()
        return obs_n

    def render(self, mode='human', close=True):
        raise NotImplementedError('rendering (environment.py:216-298) is out of scope for the batched simulator')
