"""Does the replay append of step t (memory-bound) run beside the actors of step t + 1 (issue-bound) when they are put on two
streams?  act -> step -> store loop of tools/train_demo.py without updates, serial vs overlapped:
    python tools/overlap_probe.py [num_envs] [steps]"""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd import make_env  # noqa: E402
from marl_amd.maddpg import MADDPGAgentTrainer, act_all  # noqa: E402
from marl_amd.replay_buffer import add_all  # noqa: E402


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
    args = types.SimpleNamespace(lr=1e-2, gamma=0.95, batch_size=1024, num_units=64, max_episode_len=25)
    env = make_env('simple_tag', num_envs=B, dtype=torch.float32, seed=0)
    w = env.world
    obs_shape_n = [env.observation_space[i].shape for i in range(env.n)]
    trainers = [MADDPGAgentTrainer('agent_%d' % i, obs_shape_n, env.action_space, i, args, replay_size=1 << 22,
                                   seed=0, env_id0=w.env_id0) for i in range(env.n)]
    buffers = [tr.replay_buffer for tr in trainers]
    bufs = [w.new_action_buffer() for _ in range(2)]
    views = [[b[i].t() for i in range(env.n)] for b in bufs]
    side = torch.cuda.Stream()
    main_s = torch.cuda.current_stream()

    def run(overlap, n):
        obs_n = env.reset()
        pending = None
        for t in range(n):
            k = t & 1
            action_n = act_all(trainers, obs_n, out=views[k])
            if pending is not None:
                main_s.wait_event(pending)              # step t overwrites what append t - 1 reads (obs slot, rewards)
            new_obs_n, rew_n, done_n, info_n = env.step(bufs[k])
            if overlap:
                ev = torch.cuda.Event()
                ev.record(main_s)
                side.wait_event(ev)
                with torch.cuda.stream(side):
                    add_all(buffers, obs_n, action_n, rew_n, new_obs_n, done_n)
                    pending = torch.cuda.Event()
                    pending.record(side)
            else:
                add_all(buffers, obs_n, action_n, rew_n, new_obs_n, done_n)
            obs_n = new_obs_n
        torch.cuda.synchronize()

    for overlap in (False, True, False, True):
        run(overlap, 30)
        t0 = time.perf_counter()
        run(overlap, steps)
        dt = time.perf_counter() - t0
        print('%d envs, %s: %.3f ms/step' % (B, 'append beside the next actors' if overlap else 'serial', 1e3 * dt / steps), flush=True)


if __name__ == '__main__':
    main()
