"""Replay buffer of the learner, kept in HBM (reference: maddpg/maddpg/trainer/replay_buffer.py; one per
agent, maddpg.py:147,154-156).  SURVEY 8f row f4, first slice: observations never leave the GPU.

Same methods as the reference class.  ``add`` takes one step of all B envs at once -- ``[B, d]`` tensors as
``env.step`` returns them (transposed views of the kernels' SoA rows) or any other strides, ``[B]`` rewards
and dones -- and equals B reference ``add()`` calls in env order.  Records are stored transition-major
(``[capacity, width]``, width = obs + act + 1 + obs + 1), so ``sample_index`` is one row gather that returns
``[batch, d]`` tensors ready for the networks.  libmpe_hip.so: mpe_replay_add_*, mpe_replay_gather_*.

``ReplayBuffer(size, compact=True)`` stores ``obs_tp1`` once: it is the next step's ``obs_t`` of the same env, so the
ring keeps (obs_t | act | rew | done) and looks the successor row up when a batch is gathered (40 % fewer bytes per
appended step for simple_tag's 39-column records).  Steps without a successor -- the last one of an episode, which the
loop announces with ``terminal`` exactly as the reference's ``experience(..., terminal)`` receives it
(maddpg.py:154, experiments/simple_train.py:283-287), and the newest one whenever a sample is drawn -- keep their own
copy.  Needs ``size % num_envs == 0``; samples are bit-identical to the full ring's (tests/test_gpu_replay.py)."""
import ctypes as C

import torch

from . import _hip


class ReplayBuffer(object):
    def __init__(self, size, device=None, dtype=torch.float32, compact=False):
        """replay_buffer.py:5-16.  ``size``: max number of transitions; old ones are overwritten."""
        self.compact = bool(compact)
        self._ends = None          # compact: [capacity, od] own obs_tp1 copies (rows of steps flagged in _own)
        self._own = None           # compact: uint8 [capacity / n] device flags, _own_host their host shadow
        self._own_host = None
        self._n = None             # compact: transitions per append (fixed by the first one)
        self._pending = None       # compact: (first ring row, obs_tp1 tensor) of the newest step, added with terminal=False
        self._maxsize = int(size)
        self._next_idx = 0
        self._len = 0
        self.device = torch.device('cuda' if device is None else device)
        if self.device.type != 'cuda':
            raise RuntimeError('marl_amd.ReplayBuffer lives in GPU memory (no CPU fallback)')
        self.dtype = dtype
        self.check_index = True    # validate sample indices like the reference's list indexing (one host sync)
        self._ring = None          # [capacity, width], allocated by the first add()
        self._dims = None          # (obs_dim, act_dim)
        self._cursor_dev = None    # int64[1]: moving part of the ring cursor for appends recorded in a HIP graph

    def __len__(self):             # :18-19
        return self._len

    def clear(self):               # :21-23
        self._len = 0
        self._next_idx = 0
        self._pending = None

    def device_cursor(self):
        """int64 device word that ``add_all(..., cursors=True)`` adds to the append position (mpe_replay_add_multi_cur_*):
        0 outside graph replays (marl_amd.graph.CapturedTrainSteps sets it before each replay)."""
        if self._cursor_dev is None:
            self._cursor_dev = torch.zeros(1, dtype=torch.int64, device=self.device)
        return self._cursor_dev

    def one_launch_capable(self, n):
        """Whether ``add_all(..., cursors=True)`` of n transitions can append to this (already allocated) buffer: the
        one-launch kernel of mpe_replay_add_multi_cur_* holds a 256-transition tile of the record plus its column tables
        in 64 KiB of LDS (records of <= 61 f32 / 30 f64 columns) and needs a full tile."""
        width = max(self._ring.shape[1], self._dims[0] if self.compact else 0)
        esize = self._ring.element_size()
        return width * 257 * esize + 160 * 16 <= 64 * 1024 and min(n, self._maxsize) >= 256

    def _seg(self, t, n):
        """mpe_replay_seg of a [n, d] / [n] tensor with arbitrary strides."""
        if not torch.is_tensor(t):             # e.g. the scalar False of done_n (environment.py:108)
            t = torch.full((n,), float(t), dtype=self.dtype, device=self.device)
        if t.dtype != self.dtype or t.device != self.device:
            t = t.to(device=self.device, dtype=self.dtype)
        if t.dim() == 1:
            t = t.unsqueeze(1)
        if t.shape[0] != n:
            raise ValueError('expected %d transitions, got a tensor of shape %s' % (n, tuple(t.shape)))
        return t, _hip.ReplaySeg(t.data_ptr(), t.shape[1], t.stride(1), t.stride(0))

    def _prepare(self, obs_t, action, reward, obs_tp1, done, terminal=None):
        """The appends of one step (allocates the ring on first use): (n, jobs, keep-alive); a job is
        (width, segment array, number of segments, destination ring).  Full records: one job of five segments.  Compact:
        (obs_t | act | rew | done) into the ring, plus obs_tp1 into ``_ends`` unless ``terminal`` is False."""
        n = reward.shape[0]
        keep, segs = zip(*[self._seg(t, n) for t in (obs_t, action, reward, obs_tp1, done)])
        width = sum(s.rows for s in segs)
        od, ad = segs[0].rows, segs[1].rows
        if self._ring is None:
            self._dims = (od, ad)
            if self.compact:
                if n < 1 or self._maxsize % n != 0 or self._maxsize // n < 2:
                    raise ValueError('a compact replay buffer needs size %% num_envs == 0 and room for two steps '
                                     '(size %d, %d transitions per step)' % (self._maxsize, n))
                self._n = n
                self._ring = torch.zeros(self._maxsize, width - od, dtype=self.dtype, device=self.device)
                self._ends = torch.zeros(self._maxsize, od, dtype=self.dtype, device=self.device)
                self._own = torch.ones(self._maxsize // n, dtype=torch.uint8, device=self.device)
                self._own_host = [1] * (self._maxsize // n)
            else:
                self._ring = torch.zeros(self._maxsize, width, dtype=self.dtype, device=self.device)
        elif (od, ad) != self._dims or segs[3].rows != self._dims[0]:
            raise ValueError('transition shape changed: %s' % ((od, ad),))
        if not self.compact:
            return n, [(width, (_hip.ReplaySeg * 5)(*segs), 5, self._ring)], keep
        if n != self._n:
            raise ValueError('a compact replay buffer takes %d transitions per append, got %d' % (self._n, n))
        if self._pending is not None:
            # the previous step was added with terminal=False: this step's obs_t IS its obs_tp1 (nothing was copied)
            t = keep[0]
            prev = self._pending[1]
            if t.data_ptr() != prev.data_ptr() or t.stride() != prev.stride() or t.shape != prev.shape:
                raise ValueError('compact replay buffer: the previous add() was not flagged terminal, so this obs_t must '
                                 'be the tensor that was its obs_tp1 (pass terminal=True to the add() before a reset)')
        jobs = [(width - od, (_hip.ReplaySeg * 4)(segs[0], segs[1], segs[2], segs[4]), 4, self._ring)]
        if terminal is None or terminal:
            jobs.append((od, (_hip.ReplaySeg * 1)(segs[3]), 1, self._ends))
        return n, jobs, keep

    def _set_own(self, slot, value):
        if self._own_host[slot] != value:
            self._own_host[slot] = value
            self._own[slot] = value

    def _advance(self, n, keep, terminal=None, flags=True):
        self._keep = keep          # sources stay alive until the stream has consumed them
        if self.compact:
            own = terminal is None or bool(terminal)
            if flags:              # (appends recorded in a HIP graph: marl_amd.graph sets the flags after each replay)
                self._set_own(self._next_idx // self._n, 1 if own else 0)
            self._pending = None if own else (self._next_idx, keep[3])
        self._next_idx = (self._next_idx + n) % self._maxsize
        self._len = min(self._len + n, self._maxsize)

    def _flush_pending(self):
        """Compact: a sample is about to be drawn while the newest step has no successor yet: give it its own copy of
        obs_tp1 (one [n, od] copy per draw, i.e. per 100 steps in the reference's loop)."""
        if self._pending is None:
            return
        row0, t = self._pending
        t, seg = self._seg(t, self._n)
        fn = getattr(_hip.lib, 'mpe_replay_add_' + _hip.suffix(self.dtype))
        with _hip.on_device(self.device):
            _hip.check(fn(self._maxsize, row0, self._n, self._dims[0], 1, (_hip.ReplaySeg * 1)(seg), self._ends.data_ptr(),
                          _hip.stream_of(self.device)), 'mpe_replay_add')
        self._set_own(row0 // self._n, 1)
        self._pending = None

    def add(self, obs_t, action, reward, obs_tp1, done, terminal=None):
        """replay_buffer.py:25-32 for one step of B envs (= B add() calls in env order).  ``terminal`` (compact buffers
        only; what the reference's loop hands to ``experience``): False = the next add() continues these episodes, so
        its obs_t is this obs_tp1 and nothing is copied now; True / None = this step keeps its own copy."""
        n, jobs, keep = self._prepare(obs_t, action, reward, obs_tp1, done, terminal)
        fn = getattr(_hip.lib, 'mpe_replay_add_' + _hip.suffix(self.dtype))
        with _hip.on_device(self.device):
            for width, arr, nseg, ring in jobs:
                _hip.check(fn(self._maxsize, self._next_idx, n, width, nseg, arr, ring.data_ptr(),
                              _hip.stream_of(self.device)), 'mpe_replay_add')
        self._advance(n, keep, terminal)

    def _encode_sample(self, idxes):
        """replay_buffer.py:34-44: (obs [b, od], act [b, ad], rew [b], obs_next [b, od], done [b])."""
        if self._ring is None:
            raise IndexError('sample from an empty replay buffer')
        if not torch.is_tensor(idxes):
            idxes = torch.as_tensor(list(idxes), dtype=torch.int64)
        idxes = idxes.to(device=self.device, dtype=torch.int64).contiguous()
        b = idxes.numel()
        if self.check_index and b:
            # the reference indexes a Python list of len(self) transitions: anything outside raises IndexError
            # (negative indices count from the end there; they are not used by any caller and rejected here)
            lo, hi = torch.aminmax(idxes)
            if int(lo) < 0 or int(hi) >= self._len:
                raise IndexError('replay index out of range [0, %d): min %d, max %d' % (self._len, int(lo), int(hi)))
        od, ad = self._dims
        out = torch.empty(b, 2 * od + ad + 2, dtype=self.dtype, device=self.device)
        with _hip.on_device(self.device):
            if self.compact:
                self._flush_pending()
                fn = getattr(_hip.lib, 'mpe_replay_gather_compact_' + _hip.suffix(self.dtype))
                _hip.check(fn(self._maxsize, od, ad, self._n, idxes.data_ptr(), b, self._ring.data_ptr(),
                              self._ends.data_ptr(), self._own.data_ptr(), out.data_ptr(), _hip.stream_of(self.device)),
                           'mpe_replay_gather_compact')
            else:
                fn = getattr(_hip.lib, 'mpe_replay_gather_' + _hip.suffix(self.dtype))
                _hip.check(fn(self._maxsize, out.shape[1], idxes.data_ptr(), b, self._ring.data_ptr(), out.data_ptr(),
                              _hip.stream_of(self.device)), 'mpe_replay_gather')
        return (out[:, :od], out[:, od:od + ad], out[:, od + ad], out[:, od + ad + 1:2 * od + ad + 1],
                out[:, 2 * od + ad + 1])

    def make_index(self, batch_size):
        """replay_buffer.py:46-47: uniform indices over the stored transitions (device tensor)."""
        return torch.randint(0, self._len, (batch_size,), device=self.device, dtype=torch.int64)

    def make_latest_index(self, batch_size):
        """replay_buffer.py:49-52: the latest ``batch_size`` slots, shuffled."""
        if batch_size > self._len:
            raise IndexError('make_latest_index(%d) with only %d transitions stored' % (batch_size, self._len))
        idx = (self._next_idx - 1 - torch.arange(batch_size, device=self.device, dtype=torch.int64)) % self._maxsize
        return idx[torch.randperm(batch_size, device=self.device)]

    def sample_index(self, idxes):      # :54-55
        return self._encode_sample(idxes)

    def sample(self, batch_size):       # :57-83
        if batch_size > 0:
            idxes = self.make_index(batch_size)
        else:
            idxes = torch.arange(self._len, device=self.device, dtype=torch.int64)
        return self._encode_sample(idxes)

    def collect(self):                  # :85-86
        return self.sample(-1)


def add_all(buffers, obs_n, action_n, reward_n, obs_tp1_n, done_n, cursors=False, terminal=None):
    """``buffers[i].add(obs_n[i], action_n[i], reward_n[i], obs_tp1_n[i], done_n[i])`` for every agent (the loop of
    experiments/simple_train.py:284-287) in ONE launch (libmpe_hip.so: mpe_replay_add_multi_*).
    ``cursors``: append at ``_next_idx + *device_cursor()`` (mpe_replay_add_multi_cur_*; for HIP-graph capture).
    ``terminal``: see ``ReplayBuffer.add`` (compact buffers)."""
    b0 = buffers[0]
    if any(b.device != b0.device or b.dtype != b0.dtype for b in buffers):
        raise ValueError('add_all needs buffers on one device with one dtype')
    done_n = list(done_n)
    if all(torch.is_tensor(d) and d is done_n[0] for d in done_n) and done_n[0].dtype != b0.dtype:
        # env.step hands every agent the same bool tensor (environment.py:108): convert it once, not per buffer
        done_n = [done_n[0].to(b0.dtype)] * len(done_n)
    prep = [b._prepare(*args, terminal) for b, args in zip(buffers, zip(obs_n, action_n, reward_n, obs_tp1_n, done_n))]
    n = prep[0][0]
    if any(p[0] != n for p in prep):
        raise ValueError('every agent must add the same number of transitions')
    owners = [b for b, p in zip(buffers, prep) for _ in p[1]]
    flat = [j for p in prep for j in p[1]]
    jobs = (_hip.ReplayJob * len(flat))(*[_hip.ReplayJob(b._maxsize, b._next_idx, width, nseg, arr, ring.data_ptr())
                                          for b, (width, arr, nseg, ring) in zip(owners, flat)])
    with _hip.on_device(b0.device):
        if cursors:
            cur = (C.c_void_p * len(flat))(*[b.device_cursor().data_ptr() for b in owners])
            fn = getattr(_hip.lib, 'mpe_replay_add_multi_cur_' + _hip.suffix(b0.dtype))
            _hip.check(fn(len(flat), jobs, cur, n, _hip.stream_of(b0.device)), 'mpe_replay_add_multi_cur')
        else:
            fn = getattr(_hip.lib, 'mpe_replay_add_multi_' + _hip.suffix(b0.dtype))
            _hip.check(fn(len(flat), jobs, n, _hip.stream_of(b0.device)), 'mpe_replay_add_multi')
    for b, p in zip(buffers, prep):
        b._advance(n, p[2], terminal, flags=not cursors)


class StepStore(object):
    """Replay without an append (libmpe_hip.so: mpe_replay_gather_steps_*; include/mpe_hip.h, STEP STORE).

    Everything a transition holds is already written by the step's own launches -- the actors write the action rows,
    the fused env step the reward rows and the NEXT observation rows -- so the store hands them a different slot of its
    ``[slots, rows, num_envs]`` arrays at every step and the rings fill themselves: no copy, no append launch (the append
    was 0.138 ms of the 0.58 ms training step at 1 Mi envs, and 0.217 ms with full records).  Transition (slot s, env e)
    is ``obs[s] | act[s] | rew[s] | obs[s + 1] | done[s]``; a reset writes its observation into the slot after the
    episode's last ``obs_tp1``, which therefore holds no transition of its own (one slot per episode).

        store = StepStore(env, slots=104)
        for tr, view in zip(trainers, store.views()):
            tr.replay_buffer = view                     # the reference's per-agent buffer interface, read side
        obs_n = env.reset()
        while True:
            action_n = act_all(trainers, obs_n, out=store.action_views())   # or any other writer of those views
            obs_n, rew_n, done_n, info_n = store.step()                    # env.step on the slot's own action rows
            if terminal: obs_n = env.reset()

    ``store.step(action_n)`` copies a caller's action list into the slot first.  ``len(view)`` counts the stored
    transitions; index ``k * num_envs + e`` is env e of the k-th OLDEST stored step (the reference's ring enumerates
    storage positions instead: the same set, sampled uniformly either way; before the first wrap the two orders
    coincide).  Samples are bit-identical to a full ``ReplayBuffer`` fed the same steps (tests/test_gpu_replay.py).
    Needs the fused env step (built-in scenario callbacks), the default action decode and no ``done_callback``."""

    def __init__(self, env, slots):
        w = env.world
        if not env._fused or not env._all_policy:
            raise RuntimeError('StepStore needs the fused env step and no scripted agents')
        if env.discrete_action_input or not env.discrete_action_space or env.force_discrete_action:
            raise RuntimeError('StepStore: the stored action rows are the ones the default decode reads')
        if env.shared_reward:
            raise RuntimeError('StepStore: shared_reward sums the reward rows on the host side (environment.py:113-115)')
        S = int(slots)
        if S < 3:
            raise ValueError('StepStore needs at least 3 slots')
        self.env, self.slots = env, S
        if w.obs is None:
            w._alloc_outputs()
        d = w.descriptor()
        A = len(w.agents)
        self.obs = w.rows(S, w.obs.shape[0])
        self.act = w.rows(S, sum(w.act_dims))
        self.rew = w.rows(S, A)
        cur = w._obs_slot
        self.obs[0].copy_(w._obs_pair[cur])         # whatever observations are current stay current, in slot 0
        w._obs_pair, w._views_pair = [], []
        for s in range(S):
            obs = self.obs[s]
            w._obs_pair.append(obs)
            w._views_pair.append([obs[_hip.lib.mpe_obs_row0(d, i):_hip.lib.mpe_obs_row0(d, i) + _hip.lib.mpe_obs_dim(d, i)].t()
                                  for i in range(A)])
        w._obs_slot = 0
        w.obs, w.obs_views = w._obs_pair[0], w._views_pair[0]
        w._buf.set(obs=w.obs)
        self._act_soa = [self.act[s].view(A, 5, w.num_envs) if w._uniform_actions else self.act[s] for s in range(S)]
        self._act_views = [[self.act[s][w.act_row0[i]:w.act_row0[i] + w.act_dims[i]].t() for i in env._pidx]
                           for s in range(S)]
        self._live = []              # slots that hold a transition, oldest first
        self._live_dev = torch.zeros(S + 1, dtype=torch.int32, device=w.device)   # the table, then the count
        self._live_dirty = True
        self._cursor = 0             # the observation slot after this store's last step
        self._views = [StepStoreView(self, j) for j in range(env.n)]

    def views(self):
        return list(self._views)

    def _sync(self):
        """Observation slots written behind the store's back (a reset, a gather) hold no transition any more."""
        w = self.env.world
        s = w._obs_slot
        j = self._cursor
        while j != s:
            j = (j + 1) % self.slots
            self._drop(j)
        self._cursor = s
        return s

    def _drop(self, slot):
        if slot in self._live:
            self._live.remove(slot)
            self._live_dirty = True

    def action_views(self):
        """``[num_envs, act_dim]`` views of the current slot's action rows, one per policy agent: where the actors write."""
        return self._act_views[self._sync()]

    def step(self, action_n=None):
        """``env.step`` with this slot's action rows (``action_n``: copied into them first), rewards into this slot's
        reward rows, the next observations into the next slot."""
        env, w = self.env, self.env.world
        if w._obs_tick != w._tick:
            raise RuntimeError('StepStore.step: call right after env.reset() or a step (the observations must be current)')
        s = self._sync()
        self._drop(s)
        if action_n is not None:
            for v, a in zip(self._act_views[s], action_n):
                if not (torch.is_tensor(a) and a.data_ptr() == v.data_ptr() and a.stride() == v.stride()):
                    v.copy_(torch.as_tensor(a, dtype=v.dtype, device=v.device).expand(v.shape))
        w.rew = self.rew[s]
        w._buf.set(rew=w.rew)
        out = env.step(self._act_soa[s])
        q = w._obs_slot                 # holds obs_tp1 now: whatever transition started there is gone
        self._drop(q)
        self._live.append(s)
        self._live_dirty = True
        self._cursor = q
        return out

    def _set_slot(self, slot):
        w = self.env.world
        w._obs_slot = slot
        w.obs, w.obs_views = w._obs_pair[slot], w._views_pair[slot]
        w._buf.set(obs=w.obs)

    def align(self, period):
        """Moves the current observations to the next slot that is a multiple of ``period`` (slots in between hold no
        transition afterwards): marl_amd.graph.CapturedStoreSteps records one HIP graph per window of K + 1 slots and
        starts every episode at such a slot."""
        s = self._sync()
        t = (-(-s // period) * period) % self.slots
        if t != s:
            self.obs[t].copy_(self.obs[s])
            j = s
            while j != t:
                j = (j + 1) % self.slots
                self._drop(j)
            self._set_slot(t)
            self._cursor = t
        return t

    def _live_table(self):
        self._sync()
        if self._live_dirty:
            t = torch.tensor(self._live + [0] * (self.slots - len(self._live)) + [len(self._live)], dtype=torch.int32)
            self._live_dev.copy_(t)
            self._live_dirty = False
        return len(self._live)


class StepStoreView(object):
    """One agent's transitions of a ``StepStore`` behind the read side of the reference's ReplayBuffer interface
    (replay_buffer.py:18-19,46-86); ``add`` raises: the store is filled by the steps themselves."""
    compact = False

    def __init__(self, store, j):
        w = store.env.world
        i = store.env._pidx[j]
        d = w.descriptor()
        self.store, self.device, self.dtype = store, w.device, w.dtype
        self._dims = (_hip.lib.mpe_obs_dim(d, i), w.act_dims[i])
        self._rows = (_hip.lib.mpe_obs_row0(d, i), w.act_row0[i], i)
        self.check_index = True

    def __len__(self):
        return self.store._live_table() * self.store.env.num_envs

    def add(self, *args, **kw):
        raise RuntimeError('a StepStore view is filled by StepStore.step(), not by add()')

    def clear(self):
        self.store._live = []
        self.store._live_dirty = True

    def make_index(self, batch_size):
        return torch.randint(0, len(self), (batch_size,), device=self.device, dtype=torch.int64)

    def make_latest_index(self, batch_size):
        n = len(self)
        if batch_size > n:
            raise IndexError('make_latest_index(%d) with only %d transitions stored' % (batch_size, n))
        idx = n - 1 - torch.arange(batch_size, device=self.device, dtype=torch.int64)
        return idx[torch.randperm(batch_size, device=self.device)]

    def sample_index(self, idxes):
        st, w = self.store, self.store.env.world
        n_live = st._live_table()
        if n_live == 0:
            raise IndexError('sample from an empty replay buffer')
        if not torch.is_tensor(idxes):
            idxes = torch.as_tensor(list(idxes), dtype=torch.int64)
        idxes = idxes.to(device=self.device, dtype=torch.int64).contiguous()
        b = idxes.numel()
        if self.check_index and b:
            lo, hi = torch.aminmax(idxes)
            if int(lo) < 0 or int(hi) >= n_live * w.num_envs:
                raise IndexError('replay index out of range [0, %d): min %d, max %d' % (n_live * w.num_envs, int(lo), int(hi)))
        od, ad = self._dims
        out = torch.empty(b, 2 * od + ad + 2, dtype=self.dtype, device=self.device)
        desc = _hip.StepStoreDesc(obs=st.obs.data_ptr(), act=st.act.data_ptr(), rew=st.rew.data_ptr(), done=None,
                                  live=st._live_dev.data_ptr(), obs_slot_stride=st.obs.stride(0),
                                  act_slot_stride=st.act.stride(0), rew_slot_stride=st.rew.stride(0), ld=w.ld,
                                  n=w.num_envs, slots=st.slots, n_live=n_live, obs_row0=self._rows[0],
                                  act_row0=self._rows[1], rew_row=self._rows[2])
        fn = getattr(_hip.lib, 'mpe_replay_gather_steps_' + _hip.suffix(self.dtype))
        with _hip.on_device(self.device):
            _hip.check(fn(C.byref(desc), od, ad, idxes.data_ptr(), b, out.data_ptr(), _hip.stream_of(self.device)),
                       'mpe_replay_gather_steps')
        return (out[:, :od], out[:, od:od + ad], out[:, od + ad], out[:, od + ad + 1:2 * od + ad + 1],
                out[:, 2 * od + ad + 1])

    def sample(self, batch_size):
        if batch_size > 0:
            return self.sample_index(self.make_index(batch_size))
        return self.sample_index(torch.arange(len(self), device=self.device, dtype=torch.int64))

    def collect(self):
        return self.sample(-1)
