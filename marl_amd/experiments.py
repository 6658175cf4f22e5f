"""Caller-side world mutations of the fork's training / evaluation loop (reference:
experiments/train_and_eval.py), batched.

    obs_n, rew_n, done_n, info_n = env.step(action_n)
    count = target_move(env)                  # train_and_eval.py:313-314
"""
import numpy as np
import torch

from . import _hip


def target_move(env):
    """train_and_eval.py:212-222: respawn every landmark that a follower (agent.advfollower) touches at
    U(-1, 1)^2; returns the number of respawned landmarks per env as an int32 ``[B]`` tensor (the
    reference returns a Python int).  One HIP launch (libmpe_hip.so: mpe_target_move_*); positions come
    from the Philox stream keyed by (seed, global env id, call counter, landmark)."""
    w = env.world
    d = w.descriptor()
    count = torch.empty(w.num_envs, dtype=torch.int32, device=w.device)
    fn = getattr(_hip.lib, 'mpe_target_move_' + _hip.suffix(w.dtype))
    _hip.check(fn(d, w.num_envs, w.ld, w.seed, w.move_counter, w.env_id0, _hip.Buffers(pos=w.pos),
                  count.data_ptr(), _hip.stream_of(w.device)), 'mpe_target_move')
    w.move_counter += 1
    w.mark_dirty()
    return count


_BENCHMARK_STARTS = ((-0.5, -0.5), (-0.5, +0.5), (+0.5, -0.5), (+0.5, +0.5), (+0.0, +0.0))


def reset_world_for_benchmark(world):
    """train_and_eval.py:103-126: fixed start positions for the first five agents (followers 0-2, leader,
    prey), zero velocity and comm for every agent; landmarks keep their positions."""
    for i, agent in enumerate(world.agents):
        if i < len(_BENCHMARK_STARTS):
            agent.state.p_pos = np.array(_BENCHMARK_STARTS[i])
        agent.state.p_vel = np.zeros(world.dim_p)
        if world.dim_c:
            agent.state.c = np.zeros(world.dim_c)
