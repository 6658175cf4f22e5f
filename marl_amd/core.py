"""Batched world model: the reference's ``multiagent/core.py`` data model (core.py:4-121) with the
per-entity numpy arrays replaced by struct-of-arrays device tensors shared by all env instances.

    world.pos  [E, 2, B]   entities = agents then landmarks (core.py:109-111)
    world.vel  [A, 2, B]
    world.comm [A, dim_c, B]
    world.act  [A, 5, B]   soft one-hot Discrete(5) actions (environment.py:188-190); with non-silent
                           agents (MultiDiscrete, environment.py:61-69) ragged rows [act_total, B]:
                           agent i owns rows act_row0[i] .. +5 (+dim_c comm entries)
    world.lforce [A, 2, B] agent.leader_force (core.py:156-192), worlds with a command hierarchy only

``entity.state.p_pos`` / ``p_vel`` / ``agent.state.c`` are ``[B, 2]`` views into those buffers
(strides (1, B)), so reference-style scenario code keeps working on whole batches.
``World.step()`` (core.py:124-140) is one HIP kernel launch (libmpe_hip.so: mpe_world_step_*).
"""
import os

import torch

from . import _hip


class EntityState(object):
    """core.py:4-9; arrays are views into the world's SoA buffers."""

    def __init__(self, entity):
        self._e = entity

    def _rows(self, name):
        w, i = self._e._world, self._e._index
        if w is None or w.pos is None:
            raise RuntimeError('entity state is only available after World.allocate()')
        if name == 'pos':
            return w.pos[i]
        if i >= w.vel.shape[0]:    # a landmark: it has velocity rows only in a world with a movable landmark
            if w._landmark_velocities() is None:
                return None
            return w.lvel[i - w.vel.shape[0]]
        return w.vel[i]

    @property
    def p_pos(self):
        return self._rows('pos').t()

    @p_pos.setter
    def p_pos(self, value):
        r = self._rows('pos')
        r.copy_(torch.as_tensor(value, dtype=r.dtype, device=r.device).expand(r.shape[1], 2).t())
        self._e._world.mark_dirty()

    @property
    def p_vel(self):
        r = self._rows('vel')
        if r is None:  # landmarks never move: zeros (simple_tag.py:54)
            w = self._e._world
            return torch.zeros(w.num_envs, 2, dtype=w.dtype, device=w.device)
        return r.t()

    @p_vel.setter
    def p_vel(self, value):
        r = self._rows('vel')
        if r is not None:
            r.copy_(torch.as_tensor(value, dtype=r.dtype, device=r.device).expand(r.shape[1], 2).t())
            self._e._world.mark_dirty()


class AgentState(EntityState):
    """core.py:12-16."""

    @property
    def c(self):
        return self._e._world.comm[self._e._index].t()

    @c.setter
    def c(self, value):
        w = self._e._world
        r = w.comm[self._e._index]
        r.copy_(torch.as_tensor(value, dtype=r.dtype, device=r.device).expand(r.shape[1], r.shape[0]).t())
        w.mark_comm_dirty()     # observations of the speaking scenarios read other.state.c; a silent agent's rows go
                                # back to zeros with the next step (core.py:226-227): World._clean_silent_comm


class Action(object):
    """core.py:19-24.  ``u`` is decoded inside the step kernel; kept for API shape."""

    def __init__(self):
        self.u = None
        self.c = None


class Entity(object):
    """core.py:27-53 (same attribute names and defaults)."""

    def __init__(self):
        self.name = ''
        self.size = 0.050
        self.movable = False
        self.collide = True
        self.density = 25.0
        self.color = None
        self.max_speed = None
        self.accel = None
        self.state = EntityState(self)
        self.initial_mass = 1.0
        self.obs_r = 0.0
        self._world = None
        self._index = -1

    @property
    def mass(self):
        return self.initial_mass


class Landmark(Entity):
    """core.py:56-58."""

    def __init__(self):
        super(Landmark, self).__init__()


class Agent(Entity):
    """core.py:61-85."""

    def __init__(self):
        super(Agent, self).__init__()
        self.movable = True
        self.silent = False
        self.blind = False
        self.pblind = True
        self.u_noise = None
        self.c_noise = None
        self.u_range = 1.0
        self.state = AgentState(self)
        self.action = Action()
        self._action_callback = None
        # fork's command hierarchy (set by the L1F*/S1L* scenarios, L1F2_3Targets_100observe.py:36-70)
        self.adversary = False
        self.advleader = False
        self.advsuperleader = False
        self.force_id = -1
        self.forced_id = -1

    @property
    def action_callback(self):
        """core.py:83: a scripted agent's ``callback(agent, world) -> Action``; World.step asks it for the agent's action
        (core.py:126-127) and the env leaves the agent out of its lists (world.policy_agents).  Batched: ``action.u`` is
        ``[B, 2]`` (or anything that broadcasts to it), the force as it stands; ``action.c`` ``[B, dim_c]`` for a
        speaking agent."""
        return self._action_callback

    @action_callback.setter
    def action_callback(self, cb):
        self._action_callback = cb
        if self._world is not None:
            self._world.invalidate_descriptor()      # a scripted agent's force takes no sensitivity

    @property
    def leader_force(self):
        """[B, 2] view of this agent's row pair in world.lforce (core.py:156)."""
        w = self._world
        if w is None or w.lforce is None:
            return None
        return w.lforce[self._index].t()

    @leader_force.setter
    def leader_force(self, value):
        w = self._world
        if w is not None and w.lforce is not None:
            r = w.lforce[self._index]
            r.copy_(torch.as_tensor(value, dtype=r.dtype, device=r.device).expand(r.shape[1], 2).t())
            w.mark_dirty()


class World(object):
    """core.py:88-140, batched over ``num_envs`` instances."""

    def __init__(self):
        self.agents = []
        self.landmarks = []
        self.dim_c = 0
        self.dim_p = 2
        self.dim_color = 3
        self.dt = 0.1
        self.damping = 0.25
        self.contact_force = 1e+2
        self.contact_margin = 1e-3
        # batched state (allocate())
        self.num_envs = 0
        self.ld = 0                # row stride of every SoA buffer (>= num_envs, see padded_stride)
        self.device = None
        self.dtype = None
        self.pos = self.vel = self.comm = self.act = self.noise = self.lforce = self.lvel = None
        self.cnoise = None         # optional caller-supplied normals for c_noise, [A, dim_c, B]
        # u_noise / c_noise normals (core.py:147-148,229): `world.noise` / `world.cnoise` when the caller supplies them
        # (recorded draws: parity mode); otherwise drawn INSIDE the step kernels, keyed by (seed, global env id,
        # noise_step, agent, component) -- no extra launch, no noise rows in HBM, the same draws however the batch is
        # sharded.  host_noise = True: torch's generator fills a buffer per step instead (the round-2 path).
        self.host_noise = False
        self.noise_step = 0        # steps taken so far (the Philox counter word of the in-kernel draws)
        self.act_rows = None       # [act_total, B]: the action buffer as the kernel indexes it
        self.act_row0 = self.act_dims = None
        self.goal = None           # int32 [B]: goal landmark index (agent.goal_a), goal scenarios only
        # gather outputs (allocated on first use): obs [obs_total,B], rew [A,B], info [A,W,B]
        self.obs = self.rew = self.info = None
        self.obs_views = None
        self.obs_buffers = 2       # observation outputs alternate between this many buffers (_alloc_outputs)
        self.stats = None          # per-workgroup statistics rows (enable_stats())
        self._tick = 0             # bumped whenever the state changes
        self._gather_tick = -1     # tick the gather buffers (obs, rew, info) correspond to
        self._obs_tick = -1        # tick the obs buffer alone corresponds to
        # device reset RNG: Philox key / per-reset counter / global id of env 0 (multi-GPU shards)
        self.seed = 0
        self.move_counter = 0      # target_move() calls so far (RNG counter of the respawns)
        self.episode = 0
        self.env_id0 = 0
        # scenario kind + reset ranges for the C ABI descriptor (set by the built-in scenarios)
        self.hip_kind = None
        self.agent_pos_range = 1.0
        self.landmark_pos_range = 1.0
        self._desc = None

    # -- core.py:109-121 -----------------------------------------------------------------------------
    @property
    def entities(self):
        return self.agents + self.landmarks

    @property
    def policy_agents(self):
        return [agent for agent in self.agents if agent.action_callback is None]

    @property
    def scripted_agents(self):
        return [agent for agent in self.agents if agent.action_callback is not None]

    # -- batched storage -----------------------------------------------------------------------------
    @staticmethod
    def padded_stride(num_envs, elem_bytes):
        """Row stride ``ld`` (elements) of the SoA buffers for a batch of ``num_envs``.  Rows are padded
        so that the stride in bytes is 1 KiB past a multiple of 4 KiB: with a power-of-two stride (1 Mi
        f32 envs = exactly 4 MiB per row) the ~126 rows one wavefront streams collide in the DRAM address
        hash; measured on MI355X, simple_tag 1 Mi envs f32: 86.9 us per step at ld = B, 81.2 us at
        ld = B + 256 (B + 64: 85.5, + 2048: 84.7, + 8192: 92.2)."""
        nbytes = -(-int(num_envs) * elem_bytes // 256) * 256
        pad = int(os.environ.get('MARL_AMD_ROW_PAD_BYTES', 1024))     # tuning aid; multiples of 256
        nbytes += (pad - nbytes) % 4096
        return nbytes // elem_bytes

    def rows(self, *shape, dtype=None):
        """A zeroed SoA buffer ``[*shape, num_envs]`` laid out with the world's padded row stride
        (a view of a ``[*shape, ld]`` allocation): what every array handed to the kernels must be,
        since they share one ``ld``."""
        dtype = self.dtype if dtype is None else dtype
        return torch.zeros(*shape, self.ld, dtype=dtype, device=self.device)[..., :self.num_envs]

    def set_goal_rows(self, n):
        """Scenarios with one goal landmark per agent (simple_reference) keep ``world.goal`` as [n, B]."""
        self.goal = self.rows(n, dtype=torch.int32)
        self._buf.set(goal=self.goal)

    def new_action_buffer(self):
        """A tensor shaped and strided like ``world.act`` that MultiAgentEnv.step() adopts without a copy."""
        t = self.rows(sum(self.act_dims))
        return t.view(len(self.agents), 5, self.num_envs) if self.act.dim() == 3 else t

    @property
    def _uniform_actions(self):
        """Every agent's action is exactly the 5 physical entries ([A, 5, B] layout)."""
        return all(a.silent and a.movable for a in self.agents)

    def allocate(self, num_envs, device=None, dtype=torch.float64, ld=None):
        if device is None:
            device = torch.device('cuda', torch.cuda.current_device())
        device = torch.device(device)
        if device.type != 'cuda':
            raise RuntimeError('marl_amd runs on an AMD GPU (torch device "cuda"); there is no CPU path')
        _hip.suffix(dtype)
        A, E = len(self.agents), len(self.entities)
        B = int(num_envs)
        self.num_envs, self.device, self.dtype = B, device, dtype
        self.ld = int(ld) if ld is not None else self.padded_stride(B, 4 if dtype == torch.float32 else 8)
        if self.ld < B:
            raise ValueError('ld must be >= num_envs')
        self.pos = self.rows(E, 2)
        self.vel = self.rows(A, 2)
        self.comm = self.rows(A, max(self.dim_c, 1))[:, :self.dim_c]
        # action rows (environment.py:42-76): 5 physical entries, + dim_c comm entries if not silent
        self.act_dims = [(5 if a.movable else 0) + (0 if a.silent else self.dim_c) for a in self.agents]
        self.act_row0 = [sum(self.act_dims[:i]) for i in range(A)]
        self.act_rows = self.rows(sum(self.act_dims))
        self.act = self.act_rows.view(A, 5, B) if self._uniform_actions else self.act_rows
        self.lforce = self.rows(A, 2) if any(a.forced_id >= 0 for a in self.agents) else None
        self.lvel = None           # [L, 2, B] landmark velocities: allocated once a landmark is movable
        self.goal = self.rows(dtype=torch.int32)   # [B]; [goal_rows, B] for scenarios with several goals (set_goal_rows)
        self.noise = None
        self._noise_buf = None
        self._cnoise_buf = None
        # the part of the noise step counter that lives in device memory (a launch recorded in a HIP graph bakes its
        # scalar arguments: graph.py moves this word between replays); _noise_dev_val is its host shadow
        self._noise_step_dev = torch.zeros(1, dtype=torch.int32, device=device)
        self._noise_dev_val = 0
        self.obs = self.rew = self.info = self.obs_views = self.stats = None
        for i, e in enumerate(self.entities):
            e._world, e._index = self, i
        self._desc = None
        # agent.state.c is only maintained on device when somebody speaks
        self._comm_arg = self.comm if (self.dim_c and not all(a.silent for a in self.agents)) else None
        # world.comm was allocated as zeros and only the step kernels (and AgentState.c's setter, which says so) write
        # it: the rows of silent agents hold their zeros(dim_c) already, the step need not rewrite them every time
        self._silent_comm_dirty = False
        self._buf = _hip.Buffers(flags=_hip.FLAG_SILENT_COMM_ZERO, pos=self.pos, vel=self.vel, lforce=self.lforce,
                                 goal=self.goal, comm=self._comm_arg)
        self.mark_dirty()
        return self

    def set_flags(self, flags):
        """Per-call hints passed in ``mpe_buffers.flags`` (e.g. ``_hip.FLAG_FORCE_TILE``: run the generic LDS-tile
        kernel for a world that has a register-resident one).  The statistics rows depend on the kernel chosen."""
        self._buf.flags = (int(flags) & ~_hip.FLAG_DEVICE_NOISE) | _hip.FLAG_SILENT_COMM_ZERO
        if self.stats is not None:
            self.stats = None
            self.enable_stats()

    def _run_scripted(self):
        """core.py:126-127: ``agent.action = agent.action_callback(agent, self)`` for every scripted agent, in agent
        order, in front of the step kernels; the action lands in the agent's own rows of the action buffer."""
        rows = None
        for a in self.agents:
            cb = a.action_callback
            if cb is None:
                continue
            if rows is None:
                rows = self.act.view(-1, self.num_envs) if self.act.dim() == 3 else self.act
            a.action = action = cb(a, self)
            r0 = self.act_row0[a._index]
            if a.movable:
                u = torch.as_tensor(action.u, dtype=self.dtype, device=self.device).expand(self.num_envs, 2)
                rows[r0:r0 + 5].zero_()
                rows[r0 + 1].copy_(u[:, 0])
                rows[r0 + 3].copy_(u[:, 1])
                r0 += 5
            if not a.silent:
                c = torch.as_tensor(action.c, dtype=self.dtype, device=self.device).expand(self.num_envs, self.dim_c)
                rows[r0:r0 + self.dim_c].copy_(c.t())

    def _landmark_velocities(self):
        """``world.lvel`` [L, 2, B] when some landmark is movable (no scenario file has one, but core.py:211-222
        integrates every movable entity; set ``landmark.movable = True`` and call ``world.invalidate_descriptor()``),
        else None.  Allocated on first need."""
        if self.lvel is None and any(l.movable for l in self.landmarks):
            self.lvel = self.rows(len(self.landmarks), 2)
            self._buf.set(lvel=self.lvel)
        return self.lvel

    def _clean_silent_comm(self):
        """core.py:226-227: a step leaves agent.state.c of a silent agent at zeros(dim_c).  The kernels are told not to
        rewrite those rows (FLAG_SILENT_COMM_ZERO); after a write through AgentState.c's setter they are zeroed here, in
        front of the next step."""
        if self._silent_comm_dirty:
            for i, a in enumerate(self.agents):
                if a.silent and self.dim_c:
                    self.comm[i].zero_()
            self._silent_comm_dirty = False

    def mark_dirty(self):
        """Call after writing world.pos / world.vel directly (setters on entity.state do it)."""
        self._tick += 1

    def mark_comm_dirty(self):
        """Call after writing ``world.comm`` directly (``agent.state.c = ...`` does it): cached observations are
        dropped, and the rows of silent agents are zeroed again in front of the next step, which no longer rewrites
        them itself (FLAG_SILENT_COMM_ZERO; core.py:226-227 leaves a silent agent's state.c at zeros)."""
        self._tick += 1
        self._silent_comm_dirty = True

    def _alloc_outputs(self):
        d = self.descriptor()
        A, B = len(self.agents), self.num_envs
        total = _hip.lib.mpe_obs_total(d)
        # Observations are double-buffered: the tensors a step returns are views of the kernel's output rows,
        # and the reference's loops still read obs_n after the next env.step (experience(obs_n[i], ...,
        # new_obs_n[i]), simple_train.py:284-287).  Alternating two buffers keeps the previous step's views
        # intact for exactly that long at no extra traffic; world.obs_buffers = 1 halves the memory.
        self._obs_pair, self._views_pair = [], []
        for _ in range(max(1, int(self.obs_buffers))):
            obs = self.rows(total)
            views = []
            for i in range(A):
                r0, n = _hip.lib.mpe_obs_row0(d, i), _hip.lib.mpe_obs_dim(d, i)
                views.append(obs[r0:r0 + n].t())                # [B, obs_dim_i], strides (1, ld)
            self._obs_pair.append(obs)
            self._views_pair.append(views)
        self._obs_slot = 0
        self.obs, self.obs_views = self._obs_pair[0], self._views_pair[0]
        self.rew = self.rows(A)
        self.info = self.rows(A, _hip.lib.mpe_info_width(d))
        self._buf.set(obs=self.obs, rew=self.rew)

    def _flip_obs(self):
        """Point the next observation write at the other buffer (see _alloc_outputs)."""
        self._obs_slot = (self._obs_slot + 1) % len(self._obs_pair)
        self.obs, self.obs_views = self._obs_pair[self._obs_slot], self._views_pair[self._obs_slot]
        self._buf.set(obs=self.obs)

    def gather_obs(self):
        """Observations only (what MultiAgentEnv.reset() returns, environment.py:124-129)."""
        if self.obs is None:
            self._alloc_outputs()
        if self._gather_tick == self._tick or self._obs_tick == self._tick:
            return
        self._flip_obs()
        d = self.descriptor()
        fn = getattr(_hip.lib, 'mpe_observe_' + _hip.suffix(self.dtype))
        b = _hip.Buffers(flags=self._buf.flags, pos=self.pos, vel=self.vel, lforce=self.lforce, goal=self.goal,
                         obs=self.obs, comm=self._comm_arg)
        with _hip.on_device(self.device):
            _hip.check(fn(d, self.num_envs, self.ld, b, _hip.stream_of(self.device)), 'mpe_observe')
        self._obs_tick = self._tick

    def gather(self, force=False):
        """Observation / reward / benchmark_data of every agent for the current state: one launch
        of mpe_observe_* (scenario callbacks simple_tag.py:57-147 / simple_spread.py:46-99)."""
        if self.obs is None:
            self._alloc_outputs()
        if self._gather_tick == self._tick and not force:
            return
        if self._obs_tick != self._tick:         # a new state: keep the previous observations intact
            self._flip_obs()
            self._obs_tick = self._tick
        d = self.descriptor()
        fn = getattr(_hip.lib, 'mpe_observe_' + _hip.suffix(self.dtype))
        b = self._buf
        b.info = self.info.data_ptr() if self.info.numel() else None
        with _hip.on_device(self.device):
            _hip.check(fn(d, self.num_envs, self.ld, b, _hip.stream_of(self.device)), 'mpe_observe')
        self._gather_tick = self._tick

    def enable_stats(self):
        """Allocate (or zero) the per-workgroup statistics rows env_step accumulates into."""
        d = self.descriptor()
        rows = _hip.lib.mpe_stats_rows(d, self.num_envs, 4 if self.dtype == torch.float32 else 8, self._buf.flags)
        width = _hip.lib.mpe_stats_width(d)
        if self.stats is None or tuple(self.stats.shape) != (rows, width):
            self.stats = torch.zeros(rows, width, dtype=torch.float64, device=self.device)
        else:
            self.stats.zero_()
        return self.stats

    def read_stats(self, out=None):
        """Deterministic reduction of the statistics rows -> [2A+2] float64 device vector."""
        if out is None:
            out = torch.empty(self.stats.shape[1], dtype=torch.float64, device=self.device)
        with _hip.on_device(self.device):
            _hip.check(_hip.lib.mpe_stats_reduce(self.stats.data_ptr(), self.stats.shape[0], self.stats.shape[1],
                                                 out.data_ptr(), _hip.stream_of(self.device)), 'mpe_stats_reduce')
        return out

    def env_step(self, want_info=True):
        """MultiAgentEnv.step's device work in ONE launch (mpe_env_step_*): action decode + World.step
        + all observation / reward / info gathers, from the actions in ``world.act``."""
        if self.obs is None:
            self._alloc_outputs()
        d = self.descriptor()
        fn = getattr(_hip.lib, 'mpe_env_step_' + _hip.suffix(self.dtype))
        self._run_scripted()
        b = self._buf
        b.act = self.act.data_ptr()
        keep = self._bind_noise(b)
        b.info = self.info.data_ptr() if (want_info and self.info.numel()) else None
        b.stats = _hip.ptr(self.stats)
        self._clean_silent_comm()
        self._flip_obs()
        with _hip.on_device(self.device):
            _hip.check(fn(d, self.num_envs, self.ld, b, _hip.stream_of(self.device)), 'mpe_env_step')
        self._tick += 1
        self._obs_tick = self._tick
        if want_info:
            self._gather_tick = self._tick

    def descriptor(self):
        """mpe_scenario_desc for this world (rebuilt when entity properties may have changed)."""
        if self._desc is not None:
            return self._desc
        if self.hip_kind is None:
            raise RuntimeError('world.hip_kind is not set (built-in scenarios set it in make_world)')
        d = _hip.ScenarioDesc()
        A, L = len(self.agents), len(self.landmarks)
        if A > _hip.MPE_MAX_AGENTS or A + L > _hip.MPE_MAX_ENTITIES:
            raise ValueError('world too large for libmpe_hip (max %d agents / %d entities)'
                             % (_hip.MPE_MAX_AGENTS, _hip.MPE_MAX_ENTITIES))
        d.kind, d.n_agents, d.n_landmarks, d.dim_c = self.hip_kind, A, L, self.dim_c
        self._landmark_velocities()
        d.dt, d.damping = self.dt, self.damping
        d.contact_force, d.contact_margin = self.contact_force, self.contact_margin
        d.agent_pos_range, d.landmark_pos_range = self.agent_pos_range, self.landmark_pos_range
        for i, e in enumerate(self.entities):
            d.size[i], d.mass[i] = e.size, e.mass
            d.movable[i], d.collide[i] = int(bool(e.movable)), int(bool(e.collide))
            if i >= A:          # Entity.max_speed of a landmark (core.py:42,217): matters when it is movable
                d.lm_max_speed[i] = -1.0 if e.max_speed is None else e.max_speed
        d.reward_variant = int(getattr(self, 'reward_variant', 0))
        d.n_target_obs = int(getattr(self, 'n_target_obs', 3))
        for i, a in enumerate(self.agents):
            d.c_noise[i] = a.c_noise if a.c_noise else 0.0            # core.py:229
            d.accel[i] = 5.0 if a.accel is None else a.accel          # environment.py:193-196
            if a.action_callback is not None:
                # core.py:126-127,143-149: the callback's u IS the force; the sensitivity belongs to the env's decode of
                # policy agents.  _run_scripted() writes u into the agent's action rows as (0, ux, 0, uy, 0), which the
                # kernels' decode (a[1] - a[2], a[3] - a[4]) * accel returns bit for bit with accel = 1
                d.accel[i] = 1.0
            d.max_speed[i] = -1.0 if a.max_speed is None else a.max_speed
            d.u_noise[i] = a.u_noise if a.u_noise else 0.0            # core.py:147
            d.adversary[i] = int(bool(getattr(a, 'adversary', False)))
            d.silent[i] = int(bool(a.silent))
            d.force_id[i], d.forced_id[i] = int(a.force_id), int(a.forced_id)
            d.advleader[i], d.advsuperleader[i] = int(bool(a.advleader)), int(bool(a.advsuperleader))
            d.obs_r[i] = a.obs_r
            d.advfollower[i] = int(bool(getattr(a, 'advfollower', False)))
            d.leader[i] = int(bool(getattr(a, 'leader', False)))
        # simple_world_comm.py:42-57: world.landmarks ends with world.food, then world.forests
        d.n_food, d.n_forests = len(getattr(self, 'food', ())), len(getattr(self, 'forests', ()))
        _hip.check(_hip.lib.mpe_desc_check(d), 'mpe_desc_check')
        self._desc = d
        return d

    def invalidate_descriptor(self):
        self._desc = None
        self._silent_comm_dirty = True      # an agent may have turned silent: its rows must read zeros again

    def _has_c_noise(self):
        return bool(self.dim_c) and any(a.c_noise and not a.silent for a in self.agents)

    def _cnoise_ptr(self):
        """Standard normals for c_noise (core.py:229), [A, dim_c, B]: ``world.cnoise`` if the caller set it
        (recorded draws for parity); None otherwise: the kernels draw them (``host_noise``: torch.randn per step)."""
        if not self._has_c_noise() or (self.cnoise is None and not self.host_noise):
            return None
        if self._cnoise_buf is None:
            self._cnoise_buf = self.rows(len(self.agents), self.dim_c)
        if self.cnoise is None:
            self._cnoise_buf.normal_()
        else:
            self._cnoise_buf.copy_(self.cnoise)
        return self._cnoise_buf.data_ptr()

    def _noise_ptr(self):
        """Standard normals for u_noise (core.py:147).  ``world.noise`` may be set by the caller ([A,2,B], e.g.
        recorded draws for parity); None otherwise: the kernels draw them (``host_noise``: torch.randn per step)."""
        if not any(a.u_noise for a in self.agents) or (self.noise is None and not self.host_noise):
            return None, None
        if self._noise_buf is None:
            self._noise_buf = self.rows(len(self.agents), 2)
        if self.noise is None:
            self._noise_buf.normal_()
        else:
            self._noise_buf.copy_(self.noise)
        return self._noise_buf, self._noise_buf.data_ptr()

    def _bind_noise(self, b):
        """The noise members of mpe_buffers for the step about to be launched; bumps the step counter."""
        keep, b.noise = self._noise_ptr()
        b.cnoise = self._cnoise_ptr()
        if any(a.u_noise for a in self.agents) or self._has_c_noise():
            b.flags |= _hip.FLAG_DEVICE_NOISE           # draws only what the caller did not supply
            b.noise_seed, b.env_id0 = int(self.seed), int(self.env_id0)
            b.noise_step = (self.noise_step - self._noise_dev_val) & 0xFFFFFFFF
            b.noise_step_dev = self._noise_step_dev.data_ptr()
        else:
            b.flags &= ~_hip.FLAG_DEVICE_NOISE
        self.noise_step = (self.noise_step + 1) & 0xFFFFFFFF
        return keep

    def _set_noise_dev(self, value):
        """Moves the device part of the noise step counter (graph replays); eager steps subtract it again."""
        value &= 0xFFFFFFFF
        if value != self._noise_dev_val:
            self._noise_step_dev.fill_(value - (1 << 32) if value >= (1 << 31) else value)
            self._noise_dev_val = value

    # -- core.py:124-140 -----------------------------------------------------------------------------
    def step(self):
        """Advance every env by one tick using the actions in ``world.act``."""
        d = self.descriptor()
        fn = getattr(_hip.lib, 'mpe_world_step_' + _hip.suffix(self.dtype))
        self._run_scripted()
        b = self._buf
        b.act = self.act.data_ptr()
        keep = self._bind_noise(b)
        self._clean_silent_comm()
        with _hip.on_device(self.device):
            _hip.check(fn(d, self.num_envs, self.ld, b, _hip.stream_of(self.device)), 'mpe_world_step')
        self.mark_dirty()
