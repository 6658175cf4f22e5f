"""Multi-GPU: env instances shard trivially (there is no cross-env term anywhere on the path), one
process per GPU.  The only cross-GPU traffic is one all-reduce(sum) of the episode statistics
vector (2A+2 float64) per reporting window -- ``torch.distributed`` backend "nccl" (= RCCL over
xGMI on ROCm); "gloo" works for CPU rehearsal.  (Reference: the host-side sums of rew_n / info_n in
experiments/simple_train.py:284-286,357.)"""
import torch
import torch.distributed as dist


def shard_range(total_envs, rank, world_size):
    """Contiguous split: rank r owns envs [env_id0, env_id0 + count).  Global env ids key the reset
    RNG, so results do not depend on world_size."""
    base, rem = divmod(int(total_envs), int(world_size))
    count = base + (1 if rank < rem else 0)
    env_id0 = rank * base + min(rank, rem)
    return env_id0, count


def all_reduce_stats(vec, group=None):
    """Sum the statistics vector over ranks in place (no-op without an initialised process group)."""
    if dist.is_available() and dist.is_initialized():
        if vec.is_cuda and dist.get_backend(group) == 'gloo':
            # CPU rehearsal backend: stage through the host (RCCL reduces the device tensor directly)
            host = vec.cpu()
            dist.all_reduce(host, op=dist.ReduceOp.SUM, group=group)
            vec.copy_(host)
        else:
            dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=group)
    return vec


def summarize_stats(vec, n_agents, steps_per_episode=25):
    """[2A+2] vector -> dict of per-agent mean episode reward / collisions per episode."""
    v = torch.as_tensor(vec).double().cpu()
    A = n_agents
    env_steps = float(v[2 * A + 1])
    episodes = env_steps / steps_per_episode if env_steps else float('nan')
    return {
        'env_steps': env_steps,
        'episodes': episodes,
        'mean_episode_reward': (v[:A] / episodes).tolist(),
        'collisions_per_episode': (v[A:2 * A] / episodes).tolist(),
        'extra_per_episode': float(v[2 * A]) / episodes,
    }


class EpisodeStats(object):
    """Window of episode statistics for a batched world: device-side accumulation inside the fused
    step kernel, deterministic on-device reduce, one all-reduce when read."""

    def __init__(self, world, group=None):
        self.world = world
        self.group = group
        world.enable_stats()

    def reset(self):
        self.world.enable_stats()

    def read(self):
        vec = self.world.read_stats()
        return all_reduce_stats(vec, self.group)

    def read_async(self):
        """Like ``read`` but does not make the current stream wait for the collective: returns ``(vec, work)``; the
        all-reduce runs on RCCL's own stream beside whatever is enqueued next (the reset and the next episode's steps)
        until ``finish`` is called.  ``work`` is None without a process group or with the gloo rehearsal backend, whose
        host staging is synchronous."""
        vec = self.world.read_stats()
        if (dist.is_available() and dist.is_initialized() and vec.is_cuda
                and dist.get_backend(self.group) != 'gloo'):
            return vec, dist.all_reduce(vec, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        return all_reduce_stats(vec, self.group), None

    @staticmethod
    def finish(pending):
        """Orders the current stream behind the collective of ``read_async`` and returns the reduced vector."""
        vec, work = pending
        if work is not None:
            work.wait()
        return vec

    def summary(self, steps_per_episode=25):
        return summarize_stats(self.read(), len(self.world.agents), steps_per_episode)
