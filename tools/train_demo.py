"""The reference's training loop shape (experiments/simple_train.py:257-300) entirely on one GPU: batched env
step -> per-agent replay rings -> MADDPG updates every 100 steps.  Prints agent-steps/s with the learner in the
loop (run on the GPU box from the repo root):
    python tools/train_demo.py [scenario] [num_envs] [steps] [eager|graph] [compact|full|store] [f16actor]
One episode (25 iterations) is one HIP-graph replay (marl_amd.graph.CapturedTrainSteps) and the learners' updates
another (CapturedUpdates) unless `eager` is given.  `compact`: replay rings that store obs_tp1 once
(marl_amd.replay_buffer.ReplayBuffer(compact=True)).  `store`: no append at all -- the actors and the env step write
into the slots of a marl_amd.replay_buffer.StepStore (104 slots = 4 windows of 26; CapturedStoreSteps).  `f16actor`: the
opt-in actor kernel on two f16 pieces per operand (MPE_ACTOR_MFMA_F16: ~1e-6 instead of ~4e-7 against PyTorch f32)."""
import os
import sys
import time
import types

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from marl_amd import make_env  # noqa: E402
from marl_amd.graph import CapturedStoreSteps, CapturedTrainSteps, CapturedUpdates  # noqa: E402
from marl_amd.maddpg import MADDPGAgentTrainer, act_all  # noqa: E402
from marl_amd.replay_buffer import StepStore, add_all  # noqa: E402


def main():
    scenario = sys.argv[1] if len(sys.argv) > 1 else 'simple_tag'
    B = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 16
    steps = int(sys.argv[3]) if len(sys.argv) > 3 else 400
    args = types.SimpleNamespace(lr=1e-2, gamma=0.95, batch_size=1024, num_units=64, max_episode_len=25)
    env = make_env(scenario, num_envs=B, dtype=torch.float32, seed=0)
    obs_shape_n = [env.observation_space[i].shape for i in range(env.n)]
    compact = len(sys.argv) > 5 and sys.argv[5] == 'compact'
    trainers = [MADDPGAgentTrainer('agent_%d' % i, obs_shape_n, env.action_space, i, args, replay_size=1 << 22,
                                   seed=0, env_id0=env.world.env_id0, replay_compact=compact) for i in range(env.n)]
    eager = len(sys.argv) > 4 and sys.argv[4] == 'eager'
    use_store = len(sys.argv) > 5 and sys.argv[5] == 'store'
    actor_kernel = 3 if 'f16actor' in sys.argv[6:] else 0
    obs_n = env.reset()
    t0 = None
    store = None
    if use_store:
        store = StepStore(env, 4 * (args.max_episode_len + 1))
        for tr, view in zip(trainers, store.views()):
            tr.replay_buffer = view

    def iteration(obs_n, terminal=True):
        if store is not None:
            act_all(trainers, obs_n, out=store.action_views(), kernel=actor_kernel)   # every agent's actor: one launch, into the slot's rows
            return store.step()[0]                                  # the env step writes rewards and next observations there
        action_n = act_all(trainers, obs_n, out=env.action_views, kernel=actor_kernel)   # every agent's actor: one launch, no copies
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        add_all([tr.replay_buffer for tr in trainers], obs_n, action_n, rew_n, new_obs_n, done_n,
                terminal=terminal)                                  # every agent's append: one launch
        return new_obs_n

    # one episode (25 iterations of act -> step -> store) is one HIP-graph replay; resets and updates stay outside
    EP = args.max_episode_len
    obs_n = iteration(obs_n)                            # the rings and the packed actor weights exist after this
    obs_n = env.reset()
    if store is not None:
        store.align(EP + 1)
    episode = None if eager else (CapturedStoreSteps(env, trainers, store, EP, kernel=actor_kernel) if store is not None
                                  else CapturedTrainSteps(env, trainers, EP, kernel=actor_kernel))
    updates = None
    for t in range(EP, steps + 100 + 1, EP):
        if t == 100 + EP:                               # 100 warm-up steps, incl. the first update
            torch.cuda.synchronize()
            t0 = time.perf_counter()
        if episode is not None:
            episode.replay()
        else:
            for k in range(EP):
                obs_n = iteration(obs_n, terminal=k == EP - 1)      # experiments/simple_train.py:283: episode_step >= max_episode_len
        obs_n = env.reset()
        for tr in trainers:
            tr.preupdate()
        due = t % 100 == 0 and all(len(tr.replay_buffer) >= tr.max_replay_buffer_len for tr in trainers)
        if eager or not due:
            for tr in trainers:
                out = tr.update(trainers, t)            # every 100 steps (maddpg.py:161-163), i.e. every 4th episode
        elif updates is None:
            updates = CapturedUpdates(trainers)         # performs this step's update and records it
            out = updates.last[-1]
        else:
            out = updates.replay()[-1]                  # all four learners' updates: one HIP-graph replay
        if out is not None:
            print('step %5d  q_loss %.4f  p_loss %.4f  mean reward %.3f' % (t, float(out[0]), float(out[1]), float(out[3])))
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print('%s: %d envs x %d steps with %d learners in the loop (%s, replay: %s): %.3f ms/step, %.3e agent-steps/s'
          % (scenario, B, steps, env.n, 'eager' if eager else 'one graph replay per episode',
             ('step store (no append)' if use_store else ('compact records' if compact else 'full records')) + ('; f16 actor kernel' if actor_kernel else ''), 1e3 * dt / steps,
             B * env.n * steps / dt))


if __name__ == '__main__':
    main()
