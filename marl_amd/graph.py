"""HIP-graph replay of a run of consecutive ``MultiAgentEnv.step()`` calls.

A fused env step of a small batch is a 10-80 us kernel; the Python + ctypes path of one ``env.step`` costs about
as much on the host, so below ~10^5 envs the loop is launch-bound (simple_spread, 65 536 envs, f64: 12 us kernel,
26 us per eager step).  ``CapturedSteps`` records the launches of K steps once (``torch.cuda.graph`` captures what
the C ABI enqueues on torch's current stream) and replays them with one host call: the reference's loop shape
(experiments/simple_train.py:257-300: act, step, store, repeat) with the per-step host work gone.

The graph bakes device pointers: the K action tensors (pass buffers from ``world.new_action_buffer()`` and keep
writing the policy's actions into them), the env's observation / reward / info buffers and the statistics rows.
``replay()`` returns the last step's ``(obs_n, reward_n, done_n, info_n)`` -- views of those fixed buffers."""
import torch

from . import _hip


class CapturedSteps(object):
    def __init__(self, env, action_seq):
        if not env._fused:
            raise RuntimeError('CapturedSteps needs the fused step (built-in scenario callbacks)')
        w = env.world
        if w.scripted_agents:
            raise RuntimeError('CapturedSteps cannot record scripted agents (their action_callback is host code)')
        for a in action_seq:
            if not (torch.is_tensor(a) and tuple(a.shape) == tuple(w.act.shape) and a.stride() == env._own_act.stride()
                    and a.dtype == w.act.dtype and a.device == w.act.device):
                raise ValueError('action_seq must hold tensors from world.new_action_buffer() (adopted in place)')
        self.env, self.actions, self.K = env, list(action_seq), len(action_seq)
        if w.obs is None:
            w._alloc_outputs()
        # everything a step moves: the warm-up step and the capture itself (which executes nothing, but runs the host
        # side of K steps) must leave the world exactly as it was, also when the capture fails
        state = [None if t is None else t.clone()
                 for t in (w.pos, w.vel, w.lforce, w.stats, w.comm if w._comm_arg is not None else None, w.lvel)]
        host = (w._obs_slot, w.act, w.noise_step, w._gather_tick)
        try:
            env.step(self.actions[0])              # every lazy one-time cost (module load, LDS limit) before the capture
            self._restore(state, host)
            # a recorded launch bakes noise_step: what moves from replay to replay is the device word (core.py)
            self._noise_base = (w.noise_step - w._noise_dev_val) & 0xFFFFFFFF
            self.graph = torch.cuda.CUDAGraph()
            # thread_local: another thread's HIP calls (the RCCL watchdog of torch.distributed polls events) must not
            # invalidate the capture; everything enqueued so far is drained first
            torch.cuda.synchronize(w.device)
            with _hip.on_device(w.device):
                with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                    for a in self.actions:
                        self.last = env.step(a)
            self._slot_after = w._obs_slot
        finally:
            self._restore(state, host)

    def _restore(self, state, host):
        w = self.env.world
        for t, saved in zip((w.pos, w.vel, w.lforce, w.stats, w.comm, w.lvel), state):
            if saved is not None and t is not None:
                t.copy_(saved)
        w._obs_slot, w.act, w.noise_step, gather_tick = host
        w.obs, w.obs_views = w._obs_pair[w._obs_slot], w._views_pair[w._obs_slot]
        w._buf.set(obs=w.obs)
        w.mark_dirty()                             # the output buffers hold the warm-up step's results: not current

    def replay(self):
        """Runs the K steps; the world's Python-side bookkeeping ends up as after K ``env.step`` calls."""
        w = self.env.world
        w._clean_silent_comm()
        if w._buf.flags & _hip.FLAG_DEVICE_NOISE:
            w._set_noise_dev(w.noise_step - self._noise_base)
        self.graph.replay()
        w.noise_step = (w.noise_step + self.K) & 0xFFFFFFFF
        w._obs_slot = self._slot_after
        w.obs, w.obs_views = w._obs_pair[w._obs_slot], w._views_pair[w._obs_slot]
        w._buf.set(obs=w.obs)
        w.act = self.actions[-1]
        w._tick += self.K
        w._obs_tick = w._tick
        if self.env.info_callback is not None:
            w._gather_tick = w._tick
        return self.last


class CapturedTrainSteps(object):
    """K iterations of the reference's inner training loop (experiments/simple_train.py:257-300) in ONE HIP graph:

        action_n = act_all(trainers, obs_n, out=env.action_views)      # every actor + Gumbel-softmax sample: 1 launch
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)          # fused env step: 1 launch
        add_all(buffers, obs_n, action_n, rew_n, new_obs_n, done_n)    # every agent's replay append: 1 launch
        obs_n = new_obs_n

    Below ~10^5 envs these three launches take 20-40 us on the GPU and 200-300 us of Python + ctypes on the host; a
    replay of the recorded launches leaves the GPU time.  Episode resets and the learners' updates stay outside (call
    ``env.reset()`` / ``trainer.update`` between replays, as the reference does every max_episode_len / 100 steps).

    A recorded launch bakes its arguments.  What moves from replay to replay lives in device memory instead: the call
    counter that keys the actors' Philox noise (``mpe_actor_forward_multi_ctr_f32``) and every replay ring's append
    cursor (``mpe_replay_add_multi_cur_*``); ``replay()`` writes both words from the host-side bookkeeping before it
    launches the graph, so eager calls may be mixed with replays freely.  The actors' packed weights are refilled in
    place by ``trainer.update``.  Call after at least one eager iteration (the rings and the packed weights must exist)
    and with current observations (right after ``env.reset()`` or ``env.step``)."""

    def __init__(self, env, trainers, K, kernel=0):
        from .maddpg import act_all
        from .replay_buffer import add_all
        if not env._fused:
            raise RuntimeError('CapturedTrainSteps needs the fused step (built-in scenario callbacks)')
        w = env.world
        if w.scripted_agents:
            raise RuntimeError('CapturedTrainSteps cannot record scripted agents (their action_callback is host code)')
        self.env, self.trainers, self.K = env, list(trainers), int(K)
        self.buffers = [tr.replay_buffer for tr in self.trainers]
        if any(tr._packed is None for tr in self.trainers) or any(b._ring is None for b in self.buffers):
            raise RuntimeError('run one eager act_all / env.step / add_all iteration before capturing')
        if w.obs is None or w._obs_tick != w._tick:
            raise RuntimeError('capture right after env.reset() or env.step(): the observations must be current')
        for b in self.buffers:      # what add_all(cursors=True) would refuse in the middle of the capture, refused up front
            if not b.one_launch_capable(w.num_envs):
                raise RuntimeError('CapturedTrainSteps: a replay record of %d columns / a batch of %d envs is outside the '
                                   'one-launch append (mpe_replay_add_multi_cur_*: <= 61 f32 / 30 f64 columns, >= 256 '
                                   'transitions)' % (b._ring.shape[1], w.num_envs))
        self.ctr = torch.zeros(1, dtype=torch.int32, device=w.device)
        for b in self.buffers:
            b.device_cursor().zero_()
        self._base_calls = self.trainers[0]._act_calls
        self._base_idx = [b._next_idx for b in self.buffers]
        self._slot0 = w._obs_slot
        self._noise_base = (w.noise_step - w._noise_dev_val) & 0xFFFFFFFF
        host = ([tr._act_calls for tr in self.trainers],
                [(b._next_idx, b._len, getattr(b, '_keep', None), b._pending) for b in self.buffers],
                (w._tick, w._obs_tick, w._gather_tick, w.act, w.noise_step))
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(w.device)
        try:
            with _hip.on_device(w.device):
                with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                    obs_n = [w.obs_views[i] for i in env._pidx]
                    for k in range(self.K):
                        action_n = act_all(self.trainers, obs_n, out=env.action_views, kernel=kernel, counter_dev=self.ctr)
                        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
                        # compact rings: the last recorded step keeps its own obs_tp1 (an episode reset follows a replay)
                        add_all(self.buffers, obs_n, action_n, rew_n, new_obs_n, done_n, cursors=True,
                                terminal=(k == self.K - 1))
                        obs_n = new_obs_n
                    self.last = (new_obs_n, rew_n, done_n, info_n)
            self._slot_after = w._obs_slot
            self._act_after = w.act
            self._n = w.num_envs
        finally:
            # a capture executes nothing, but the host-side bookkeeping moved: put it back (also after a failed capture)
            for tr, c in zip(self.trainers, host[0]):
                tr._act_calls = c
            for b, (i, l, k, pend) in zip(self.buffers, host[1]):
                b._next_idx, b._len, b._keep, b._pending = i, l, k, pend
            w._tick, w._obs_tick, w._gather_tick, w.act, w.noise_step = host[2]
            self._set_slot(self._slot0)

    def _set_slot(self, slot):
        w = self.env.world
        w._obs_slot = slot
        w.obs, w.obs_views = w._obs_pair[slot], w._views_pair[slot]
        w._buf.set(obs=w.obs)

    def replay(self):
        """Runs the K iterations; returns the last one's ``(obs_n, reward_n, done_n, info_n)``."""
        w = self.env.world
        if w._obs_tick != w._tick:
            raise RuntimeError('replay right after env.reset() or env.step(): the observations must be current')
        if w._obs_slot != self._slot0:      # the current observations sit in the other buffer: move them
            w._obs_pair[self._slot0].copy_(w._obs_pair[w._obs_slot])
            self._set_slot(self._slot0)
        delta = (self.trainers[0]._act_calls - self._base_calls) & 0xFFFFFFFF
        self.ctr.fill_(delta - (1 << 32) if delta >= (1 << 31) else delta)
        for b, base in zip(self.buffers, self._base_idx):
            b.device_cursor().fill_((b._next_idx - base) % b._maxsize)
        w._clean_silent_comm()
        if w._buf.flags & _hip.FLAG_DEVICE_NOISE:
            w._set_noise_dev(w.noise_step - self._noise_base)
        self.graph.replay()
        w.noise_step = (w.noise_step + self.K) & 0xFFFFFFFF
        for b in self.buffers:
            b.device_cursor().zero_()                  # eager add_all(cursors=True) calls see no offset
        for tr in self.trainers:
            tr._act_calls += self.K
        for b in self.buffers:
            if b.compact:          # the recorded appends do not touch the step-slot flags: K - 1 chained steps, then one
                s0, S = b._next_idx // b._n, b._maxsize // b._n     # with its own copy of obs_tp1
                for j in range(self.K):
                    b._set_own((s0 + j) % S, 1 if j == self.K - 1 else 0)
                b._pending = None
            b._next_idx = (b._next_idx + self.K * self._n) % b._maxsize
            b._len = min(b._len + self.K * self._n, b._maxsize)
        self._set_slot(self._slot_after)
        w.act = self._act_after
        w._tick += self.K
        w._obs_tick = w._tick
        if self.env.info_callback is not None:
            w._gather_tick = w._tick
        return self.last


class CapturedStoreSteps(object):
    """``CapturedTrainSteps`` for a ``marl_amd.replay_buffer.StepStore``: K iterations of

        action_n = act_all(trainers, obs_n, out=store.action_views())   # every actor: 1 launch, into the slot's rows
        obs_n, rew_n, done_n, info_n = store.step()                     # fused env step: 1 launch, into the slot's rows

    in one HIP graph -- two launches per iteration, the replay append is gone.  A recorded launch bakes its pointers and
    every step uses another slot, so there is one graph per window of K + 1 slots (K transitions + the slot of the last
    obs_tp1; the reset that follows a replay lands in the first slot of the next window): ``store.slots % (K + 1) == 0``,
    graphs are recorded the first time their window comes up.  ``store.align(K + 1)`` moves a store that is between
    windows to the next one."""

    def __init__(self, env, trainers, store, K, kernel=0):
        if store.env is not env:
            raise ValueError('the store belongs to another env')
        self.env, self.trainers, self.store, self.K, self.kernel = env, list(trainers), store, int(K), kernel
        if store.slots % (self.K + 1) != 0:
            raise ValueError('CapturedStoreSteps: %d slots are not a multiple of K + 1 = %d' % (store.slots, self.K + 1))
        if any(tr._packed is None for tr in self.trainers):
            raise RuntimeError('run one eager act_all before capturing (the packed actor weights must exist)')
        self.ctr = torch.zeros(1, dtype=torch.int32, device=env.world.device)
        self._windows = {}

    def _capture(self, s0):
        from .maddpg import act_all
        env, w, st = self.env, self.env.world, self.store
        base_calls = self.trainers[0]._act_calls
        noise_base = (w.noise_step - w._noise_dev_val) & 0xFFFFFFFF
        host = ([tr._act_calls for tr in self.trainers], (w._tick, w._obs_tick, w._gather_tick, w.act, w.noise_step, w.rew))
        graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(w.device)
        try:
            with _hip.on_device(w.device):
                with torch.cuda.graph(graph, capture_error_mode='thread_local'):
                    obs_n = [w.obs_views[i] for i in env._pidx]
                    for k in range(self.K):
                        s = s0 + k
                        act_all(self.trainers, obs_n, out=st._act_views[s], kernel=self.kernel, counter_dev=self.ctr)
                        w.rew = st.rew[s]
                        w._buf.set(rew=w.rew)
                        obs_n, rew_n, done_n, info_n = env.step(st._act_soa[s])
                    last = (obs_n, rew_n, done_n, info_n)
        finally:
            for tr, c in zip(self.trainers, host[0]):
                tr._act_calls = c
            w._tick, w._obs_tick, w._gather_tick, w.act, w.noise_step, w.rew = host[1]
            w._buf.set(rew=w.rew)
            st._set_slot(s0)
        return graph, last, base_calls, noise_base

    def replay(self):
        """Runs the K iterations from the current slot; returns the last one's ``(obs_n, reward_n, done_n, info_n)``."""
        env, w, st, K = self.env, self.env.world, self.store, self.K
        if w._obs_tick != w._tick:
            raise RuntimeError('replay right after env.reset() or a step: the observations must be current')
        s0 = st._sync()
        if s0 % (K + 1) != 0:
            raise RuntimeError('CapturedStoreSteps.replay: the store is at slot %d, not at the start of a window of %d '
                               '(store.align(%d) moves it there)' % (s0, K + 1, K + 1))
        if s0 not in self._windows:
            self._windows[s0] = self._capture(s0)
        graph, last, base_calls, noise_base = self._windows[s0]
        delta = (self.trainers[0]._act_calls - base_calls) & 0xFFFFFFFF
        self.ctr.fill_(delta - (1 << 32) if delta >= (1 << 31) else delta)
        w._clean_silent_comm()
        if w._buf.flags & _hip.FLAG_DEVICE_NOISE:
            w._set_noise_dev(w.noise_step - noise_base)
        graph.replay()
        w.noise_step = (w.noise_step + K) & 0xFFFFFFFF
        for tr in self.trainers:
            tr._act_calls += K
        for k in range(K + 1):                     # K new transitions; the slot behind them holds the last obs_tp1
            st._drop(s0 + k)
        st._live.extend(range(s0, s0 + K))
        st._live_dirty = True
        st._set_slot(s0 + K)
        st._cursor = s0 + K
        w.act = st._act_soa[s0 + K - 1]
        w.rew = st.rew[s0 + K - 1]
        w._buf.set(rew=w.rew)
        w._tick += K
        w._obs_tick = w._tick
        if env.info_callback is not None:
            w._gather_tick = w._tick
        return last


class CapturedUpdates(object):
    """The learners' updates -- ``for tr in trainers: tr.update(trainers, t)`` at a step where an update is due
    (maddpg.py:161-196, experiments/simple_train.py:295-299) -- as ONE HIP-graph replay.

    An update of one MADDPG learner is ~150 small launches (two 3-layer networks forward and backward, per-variable
    clipping, Adam, polyak); for batches of 1 024 the GPU work is a few hundred microseconds and the host path ~3 ms per
    learner, every 100 steps.  At small env batches that is most of the training loop (``tools/train_demo.py``).

    Construct it at a step where the update is due and every buffer holds at least ``max_replay_buffer_len``
    transitions: the constructor performs that update eagerly (it is the warm-up) and records the same calls;
    ``replay()`` then performs one update of every trainer per call.  The batch indices are drawn outside the graph by
    ``ReplayBuffer.make_index`` (they depend on the rings' fill level) into fixed index tensors; the Gumbel noise of the
    sampled actions comes from torch's graph-aware generator.  Adam runs in its ``capturable`` form (step counters
    on the device).  Returns every trainer's ``[q_loss, p_loss, mean(target_q), mean(rew), mean(target_q_next),
    std(target_q)]`` as device scalars (overwritten by the next replay)."""

    def __init__(self, trainers):
        self.trainers = list(trainers)
        t0 = self.trainers[0]
        dev, bs = t0.device, t0.args.batch_size
        for tr in self.trainers:
            if len(tr.replay_buffer) < tr.max_replay_buffer_len:
                raise RuntimeError('every replay buffer must hold max_replay_buffer_len transitions (maddpg.py:162)')
            for opt in (tr.p_opt, tr.q_opt):
                for g in opt.param_groups:
                    g['capturable'] = True
                for st in opt.state.values():          # states made by earlier eager steps keep their counter on the host
                    if torch.is_tensor(st.get('step')):
                        st['step'] = st['step'].to(device=dev, dtype=torch.float32)
        from .maddpg import pack_actor
        for tr in self.trainers:                       # the recorded update refills the packed actor weights in place:
            if tr._packed is None:                     # they must exist before the capture, or later acting steps
                tr._packed = pack_actor(tr.p)          # would read weights no replay ever refreshes
        self.index = [torch.zeros(bs, dtype=torch.int64, device=dev) for _ in self.trainers]
        self._draw()
        first = self._body()                           # the update that is due now; creates the optimizer states
        self.graph = torch.cuda.CUDAGraph()
        torch.cuda.synchronize(dev)
        checks = [tr.replay_buffer.check_index for tr in self.trainers]
        try:
            for tr in self.trainers:
                tr.replay_buffer.check_index = False   # the range check reads the indices back on the host
            with _hip.on_device(dev):
                with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                    self.last = self._body()
        finally:
            for tr, c in zip(self.trainers, checks):
                tr.replay_buffer.check_index = c
        for outs, eager in zip(self.last, first):      # a capture computes nothing: `last` reports the eager update
            for o, e in zip(outs, eager):
                o.copy_(e)

    def _draw(self):
        for tr, idx in zip(self.trainers, self.index):
            if tr.replay_buffer.compact:               # outside the graph: the recorded gathers then find every step
                tr.replay_buffer._flush_pending()      # either chained to its successor or with its own obs_tp1
            idx.copy_(tr.replay_buffer.make_index(idx.numel()))
            tr.replay_sample_index = idx

    def _body(self):
        return [tr.update_from_index(self.trainers, idx) for tr, idx in zip(self.trainers, self.index)]

    def replay(self):
        self._draw()
        self.graph.replay()
        return self.last
