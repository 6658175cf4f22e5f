"""Caller-side helpers of the fork's training / evaluation loops (reference:
experiments/train_and_eval.py, experiments/simple_train.py), batched.

    good_act, _ = good_agent_action(env.world.agents[-1], env.world)   # simple_train.py:270-273
    action_n[-1] = good_act
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
    with _hip.on_device(w.device):
        _hip.check(fn(d, w.num_envs, w.ld, w.seed, w.move_counter, w.env_id0, _hip.Buffers(pos=w.pos),
                      count.data_ptr(), _hip.stream_of(w.device)), 'mpe_target_move')
    w.move_counter += 1
    w.mark_dirty()
    return count


def _scripted(agent, world, mode, target):
    act = world.rows(5)                      # [5, B] view with the world's row stride
    point = world.rows(2)
    fn = getattr(_hip.lib, 'mpe_scripted_action_' + _hip.suffix(world.dtype))
    with _hip.on_device(world.device):
        _hip.check(fn(world.descriptor(), world.num_envs, world.ld, agent._index, mode, target, PREY_ACTION_CAP,
                      world.pos.data_ptr(), act.data_ptr(), point.data_ptr(), _hip.stream_of(world.device)),
                   'mpe_scripted_action')
    return act.t(), point.t()


PREY_ACTION_CAP = 1.3                        # simple_train.py:181-187 (train_and_eval.py uses agent.max_speed)


def good_agent_action(agent, world):
    """simple_train.py:157-188: the scripted prey heads for the corner of [-0.9, 0.9]^2 that is farthest
    (summed distance) from all other agents.  Returns ``(act [B, 5], target_edge [B, 2])``; ``act`` is a
    valid entry of ``action_n`` (a transposed view of SoA rows, adopted by ``env.step`` without a copy)."""
    return _scripted(agent, world, 0, 0)


def good_agent_victim_action(agent, world, target):
    """simple_train.py:190-205: the curriculum's easy prey walks towards ``target`` (an agent), gain 7.
    Returns ``act [B, 5]``."""
    return _scripted(agent, world, 1, target._index)[0]


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
