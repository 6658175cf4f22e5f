"""world_size-2 rehearsal of the multi-GPU path on CPU (gloo): envs shard by contiguous global id
ranges with no data-path collective; the only exchange is the all-reduce of the episode statistics
vector, and the result equals the single-process run over the concatenated batch."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import mpe_oracle as O

TOTAL, STEPS, SEED = 1000, 5, 42


def _stats_for_range(env_id0, count):
    """What one rank computes: reset by global env id, STEPS env steps, summed statistics
    (layout of mpe_stats_*: [sum rew (A), sum collisions (A), sum pair hits, env-steps])."""
    spec = O.simple_tag_spec()
    A = spec.n_agents
    ids = np.arange(env_id0, env_id0 + count, dtype=np.uint64)
    pos = O.reset_positions(spec, SEED, 0, ids)
    pos[:4] *= 0.3                                  # crowd the agents so collisions happen
    vel = np.zeros((A, 2, count))
    acts = np.random.RandomState(7).rand(STEPS, A, 5, TOTAL)[..., env_id0:env_id0 + count]
    bo = O.BatchedOracle(spec)
    vec = np.zeros(2 * A + 2)
    for t in range(STEPS):
        pos, vel, obs, rew, info = bo.env_step(pos, vel, acts[t])
        vec[:A] += rew.sum(1)
        vec[A:2 * A] += info[:, 0].sum(1)
        vec[2 * A] += info[:, 0].sum()
        vec[2 * A + 1] += count
    return vec


def _worker(rank, world_size, port, out_path):
    os.environ['MASTER_ADDR'] = '127.0.0.1'
    os.environ['MASTER_PORT'] = str(port)
    dist.init_process_group('gloo', rank=rank, world_size=world_size)
    from marl_amd.distributed import all_reduce_stats, shard_range
    env_id0, count = shard_range(TOTAL, rank, world_size)
    vec = torch.from_numpy(_stats_for_range(env_id0, count))
    all_reduce_stats(vec)
    if rank == 0:
        np.save(out_path, vec.numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_sharded_stats_all_reduce_equals_single_process(tmp_path):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    out = str(tmp_path / 'stats.npy')
    mp.spawn(_worker, args=(2, port, out), nprocs=2, join=True)
    got = np.load(out)
    want = _stats_for_range(0, TOTAL)
    np.testing.assert_allclose(got, want, rtol=1e-12, atol=1e-9)
    assert got[-1] == TOTAL * STEPS and got[8] > 0


def test_summarize_stats():
    from marl_amd.distributed import summarize_stats
    v = np.array([50.0, 50.0, 50.0, -100.0, 5, 5, 5, 0, 15, 250.0])
    s = summarize_stats(v, 4, steps_per_episode=25)
    assert s['episodes'] == 10 and s['mean_episode_reward'] == [5.0, 5.0, 5.0, -10.0]
    assert s['collisions_per_episode'] == [0.5, 0.5, 0.5, 0.0] and s['extra_per_episode'] == 1.5
