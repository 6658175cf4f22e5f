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
        for a in action_seq:
            if not (torch.is_tensor(a) and tuple(a.shape) == tuple(w.act.shape) and a.stride() == env._own_act.stride()
                    and a.dtype == w.act.dtype and a.device == w.act.device):
                raise ValueError('action_seq must hold tensors from world.new_action_buffer() (adopted in place)')
        self.env, self.actions, self.K = env, list(action_seq), len(action_seq)
        if w.obs is None:
            w._alloc_outputs()
        state = [t.clone() for t in (w.pos, w.vel)] + [None if t is None else t.clone() for t in (w.lforce, w.stats)]
        env.step(self.actions[0])                  # every lazy one-time cost (module load, LDS limit) before the capture
        self._restore(state)
        self._slot0 = w._obs_slot
        self.graph = torch.cuda.CUDAGraph()
        # thread_local: another thread's HIP calls (the RCCL watchdog of torch.distributed polls events) must not
        # invalidate the capture; everything enqueued so far is drained first
        torch.cuda.synchronize(w.device)
        with _hip.on_device(w.device):
            with torch.cuda.graph(self.graph, capture_error_mode='thread_local'):
                for a in self.actions:
                    self.last = env.step(a)
        self._slot_after = w._obs_slot
        self._restore(state)                       # a capture executes nothing, but the bookkeeping moved: put it back
        w._obs_slot = self._slot0
        w.obs, w.obs_views = w._obs_pair[w._obs_slot], w._views_pair[w._obs_slot]
        w._buf.set(obs=w.obs)

    def _restore(self, state):
        w = self.env.world
        w.pos.copy_(state[0])
        w.vel.copy_(state[1])
        if state[2] is not None:
            w.lforce.copy_(state[2])
        if state[3] is not None and w.stats is not None:
            w.stats.copy_(state[3])
        w.mark_dirty()

    def replay(self):
        """Runs the K steps; the world's Python-side bookkeeping ends up as after K ``env.step`` calls."""
        w = self.env.world
        self.graph.replay()
        w._obs_slot = self._slot_after
        w.obs, w.obs_views = w._obs_pair[w._obs_slot], w._views_pair[w._obs_slot]
        w._buf.set(obs=w.obs)
        w.act = self.actions[-1]
        w._tick += self.K
        w._obs_tick = w._tick
        if self.env.info_callback is not None:
            w._gather_tick = w._tick
        return self.last
