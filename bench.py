#!/usr/bin/env python3
"""Headline benchmark: agent-steps/sec of batched simple_tag on N MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

One "step" = one MultiAgentEnv.step() over the rank's whole batch (action decode + World.step +
observation + reward + benchmark info for all agents, one fused HIP launch), with the periodic
Scenario.reset_world + observation gather every 25 steps inside the timed region, episode
statistics accumulated on device and all-reduced (RCCL) once per episode.  Inputs (state, a
rotating pool of action tensors) are resident in HBM before the timed region.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, fused
env-step kernel, HIP-event brackets around the launches inside the timed region) and `cpu_baseline` (the
oracle's serial numpy loop on one host core, rank 0 at N=1 only).
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0     # MI355X HBM3E peak (MI355X_MICROARCH.md, Chip-level parameters)
EPISODE_LEN = 25

WORKLOADS = {
    # name: (scenario, scenario kwargs, envs per GPU, dtype)
    'simple_tag_1M_f32': ('simple_tag', {}, 1 << 20, torch.float32),            # BASELINE configs[2]/[3]
    'simple_spread_64k_f64': ('simple_spread', {}, 1 << 16, torch.float64),     # BASELINE configs[1]
    'scaled_tag_256k_f32': ('simple_tag', dict(num_adversaries=48, num_good_agents=16, num_landmarks=32),
                            1 << 18, torch.float32),                              # BASELINE configs[4]
    'simple_tag_1M_f64': ('simple_tag', {}, 1 << 20, torch.float64),
    # SURVEY 8f row f1 (same kernels, other gathers)
    'simple_4M_f32': ('simple', {}, 1 << 22, torch.float32),
    'simple_adversary_1M_f32': ('simple_adversary', {}, 1 << 20, torch.float32),
    'simple_push_512k_f32': ('simple_push', {}, 1 << 19, torch.float32),
    # the fork's command-hierarchy pursuit scenarios (reference file names)
    'L1F2_3Targets_1M_f32': ('L1F2_3Targets_100observe', {}, 1 << 20, torch.float32),
    'S1L2F2_10Targets_1M_f32': ('S1L2F2_10Targets_wide', {}, 1 << 20, torch.float32),
    'S1L2F2_10Targets_1M_f64': ('S1L2F2_10Targets_wide', {}, 1 << 20, torch.float64),
    # SURVEY 8f row f1, the speaking stock scenarios
    'simple_speaker_listener_2M_f32': ('simple_speaker_listener', {}, 1 << 21, torch.float32),
    'simple_reference_1M_f32': ('simple_reference', {}, 1 << 20, torch.float32),
    'simple_crypto_2M_f32': ('simple_crypto', {}, 1 << 21, torch.float32),
    'simple_world_comm_1M_f32': ('simple_world_comm', {}, 1 << 20, torch.float32),
    'L1F4_MovingTarget_1M_f32': ('L1F4_MovingTarget', {}, 1 << 20, torch.float32),
    'S1L2F4_MovingTarget_1M_f32': ('S1L2F4_MovingTarget', {}, 1 << 20, torch.float32),
}


def algorithmic_elems(env, with_info=True):
    """compulsory elements moved per env-step:
    act rows (5A, + dim_c per non-silent agent) + 2E pos rd + 2A vel rd + 4A pos,vel wr + sum obs_dim
    + A reward + A*info_w (+ 2A leader_force wr and A*dim_c comm wr in a command hierarchy)."""
    w = env.world
    A, E = len(w.agents), len(w.entities)
    obs = sum(s.shape[0] for s in env.observation_space)
    info_w = w.info.shape[1] if (with_info and w.info is not None and env.info_callback is not None) else 0
    # agent.state.c of every agent is rewritten each step once any agent speaks (core.py:224-230)
    extra = (2 * A if w.lforce is not None else 0) + (A * w.dim_c if any(not a.silent for a in w.agents) else 0)
    moving = sum(1 for a in w.agents if a.movable)           # only movable agents have / update a velocity
    pos_rd = 2 * E if moving else 0                          # simple_crypto: nobody moves, nothing looks at positions
    return sum(w.act_dims) + pos_rd + 2 * moving + 4 * moving + obs + A + A * info_w + extra


def measured_traffic(workload):
    """HBM bytes per fused launch from the committed rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE,
    tools/summarize_pmc.py); None when no pass was recorded for this workload."""
    path = os.path.join(ROOT, 'profiles', 'pmc_%s.json' % workload)
    if not os.path.exists(path):
        return None
    ks = json.load(open(path))['kernels']

    def is_fused(name):      # template tails: small/hier <..., DO_STEP, DO_OBS>, tile <..., DO_STEP, DO_OBS, HIER>
        args = name[name.index('<') + 1:name.rindex('>(')].split(', ')
        return (args[-3:-1] if 'tile_kernel' in name else args[-2:]) == ['true', 'true']
    fused = [v for k, v in ks.items() if 'mpe::' in k and is_fused(k)]
    return fused[0]['traffic_bytes'] if fused else None


def cpu_baseline(seconds=12.0):
    """Reference's serial numpy loop, restated (oracle.SerialEnv: one Python object per entity,
    validated bit-exact against the reference), on ONE host core: simple_tag, 25-step episodes
    incl. reset, random soft actions, benchmark=True -- the configuration of BASELINE.md."""
    import numpy as np
    from oracle import mpe_oracle as O
    spec = O.simple_tag_spec()
    env = O.SerialEnv(spec, benchmark=True)
    rng = np.random.RandomState(0)
    np.random.seed(0)
    env.reset()
    steps = 0
    t0 = time.perf_counter()
    while True:
        for _ in range(EPISODE_LEN):
            env.step([rng.rand(5) for _ in range(env.n)])
        env.reset()
        steps += EPISODE_LEN
        dt = time.perf_counter() - t0
        if dt >= seconds:
            break
    return {'value': steps * env.n / dt, 'unit': 'agent-steps/s', 'cores': 1, 'kind': 'port',
            'sample': '%d env-steps of simple_tag (A=4, L=2), 25-step episodes incl. reset, benchmark=True, '
                      'serial numpy loop (oracle.SerialEnv) on 1 host core of %d' % (steps, os.cpu_count())}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='simple_tag_1M_f32', choices=sorted(WORKLOADS))
    ap.add_argument('--envs', type=int, default=0, help='envs per GPU (default: the workload\'s)')
    ap.add_argument('--action-buffers', type=int, default=4,
                    help='resident action tensors rotated step by step (1 = the same tensor every step)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL over xGMI (one GPU per rank); gloo = rehearsal of the multi-rank path '
                         'with every rank on whatever GPUs exist (ranks then share devices)')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-events', action='store_true', help='skip per-launch HIP events (roofline from wall time)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world_size = int(os.environ.get('WORLD_SIZE', 1))
    if world_size != args.gpus:
        if world_size == 1 and args.gpus > 1:
            raise SystemExit('launch with torch.distributed.run --nproc-per-node %d for --gpus %d' % (args.gpus, args.gpus))
        args.gpus = world_size
    dev_index = local_rank if args.backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    import torch.distributed as dist
    if world_size > 1:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')

    from marl_amd import make_env, _hip
    from marl_amd.distributed import EpisodeStats, shard_range, summarize_stats

    scenario, kwargs, envs_per_gpu, dtype = WORKLOADS[args.workload]
    B = args.envs or envs_per_gpu
    env_id0, _ = shard_range(B * world_size, rank, world_size)      # weak scaling: B per GPU fixed
    # scenarios without a (working) benchmark_data
    has_info = scenario not in ('simple', 'simple_push', 'simple_speaker_listener', 'simple_reference')
    env = make_env(scenario, benchmark=has_info, num_envs=B, device=device, dtype=dtype, seed=0, **kwargs)
    env.world.env_id0 = env_id0
    if os.environ.get('MPE_FORCE_TILE'):      # tuning aid: run the stock worlds on the generic LDS-tile kernel
        env.world.set_flags(_hip.FLAG_FORCE_TILE)
    n = env.n
    gen = torch.Generator(device=device).manual_seed(1234 + rank)

    def random_actions():
        """Soft one-hot actions in the kernel's SoA layout: softmax over each agent's 5 physical entries
        (and, for a non-silent agent, over its dim_c command entries)."""
        w = env.world
        a = w.new_action_buffer()          # the world's padded row stride: env.step adopts it without a copy
        if w.act.dim() == 3:
            a.copy_(torch.softmax(torch.randn(n, 5, B, generator=gen, device=device, dtype=dtype), dim=1))
            return a
        for ag, r0, d in zip(w.agents, w.act_row0, w.act_dims):
            nu = 5 if ag.movable else 0                      # physical part, then the comm part
            for lo, hi in ((r0, r0 + nu), (r0 + nu, r0 + d)):
                if hi > lo:
                    a[lo:hi] = torch.softmax(torch.randn(hi - lo, B, generator=gen, device=device, dtype=dtype), dim=0)
        return a

    pool = [random_actions() for _ in range(args.action_buffers)]
    stats = EpisodeStats(env.world)
    stats_total = torch.zeros(2 * n + 2, dtype=torch.float64, device=device)

    def run(nsteps, events=None, t_base=0):
        """events: list collecting (start, end, n_launches) HIP-event brackets around each run of
        consecutive fused env-step launches (one bracket per episode segment: an event between every
        pair of launches would itself cost ~5 us of dispatch gap per step)."""
        open_bracket = None
        for t in range(nsteps):
            tt = t_base + t
            if tt % EPISODE_LEN == 0:
                if open_bracket is not None:
                    e1 = torch.cuda.Event(enable_timing=True)
                    e1.record()
                    events.append((open_bracket[0], e1, open_bracket[1]))
                    open_bracket = None
                if tt:
                    stats_total.add_(stats.read())      # deterministic reduce + one all-reduce / episode
                    stats.reset()
                env.reset()
            if events is not None and open_bracket is None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
                open_bracket = [e0, 0]
            env.step(pool[tt % len(pool)])
            if open_bracket is not None:
                open_bracket[1] += 1
        if open_bracket is not None:
            e1 = torch.cuda.Event(enable_timing=True)
            e1.record()
            events.append((open_bracket[0], e1, open_bracket[1]))

    def fence():
        if world_size > 1:
            dist.barrier()
        torch.cuda.synchronize()

    # prime every code path once (module load, first-launch costs, allocator) before the W warmup
    # steps, so that the timed region is steady state whatever W is
    env.reset()
    env.step(pool[0])
    stats_total.add_(stats.read())      # also loads torch's f64 add kernel (a ~20 ms one-time cost)
    stats_total.zero_()
    stats.reset()
    torch.cuda.synchronize()

    run(args.warmup)
    events = None if args.no_events else []
    fence()
    t0 = time.perf_counter()
    run(args.steps, events, t_base=args.warmup)
    fence()
    elapsed = time.perf_counter() - t0
    if world_size > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if args.backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    stats_total.add_(stats.read())

    elems = algorithmic_elems(env)
    bytes_per_launch = elems * env.world.pos.element_size() * B
    if events:
        per = sorted(a.elapsed_time(b) / k for a, b, k in events)
        kern_ms = sum(a.elapsed_time(b) for a, b, k in events) / sum(k for a, b, k in events)
        kern_med = per[len(per) // 2]
    else:
        kern_ms = kern_med = elapsed * 1e3 / args.steps
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9

    if rank == 0:
        value = B * world_size * n * args.steps / elapsed
        out = {
            'metric': 'agent-steps/sec (whole node) for batched simple_tag' if scenario == 'simple_tag' and not kwargs
                      else 'agent-steps/sec (whole node), %s' % args.workload,
            'value': value, 'unit': 'agent-steps/s', 'n_gpus': world_size, 'steps': args.steps,
            'warmup': args.warmup, 'ms_per_step': elapsed * 1e3 / args.steps, 'higher_is_better': True,
            'scaling': 'weak', 'vs_baseline': None,
            'dtype': 'f32' if dtype == torch.float32 else 'f64', 'data': 'synthetic',
            'config': {'workload': args.workload, 'scenario': scenario, 'envs_per_gpu': B, 'row_stride': env.world.ld,
                       'agents': n, 'entities': len(env.world.entities), 'episode_len': EPISODE_LEN,
                       'actions': 'softmax(N(0,1)) soft one-hot, %d resident buffers rotated' % len(pool),
                       'reset_in_timed_region': True, 'benchmark_info': has_info,
                       'stats_allreduce': 'per episode' if world_size > 1 else 'per episode (single rank: local reduce)',
                       'parallelism': 'env-sharded dp%d' % world_size, 'backend': args.backend if world_size > 1 else None},
            'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                         'frac': achieved / HBM_PEAK_GBS,
                         'traffic': measured_traffic(args.workload) if B == envs_per_gpu else None,
                         'kernel': 'mpe env_step (fused)', 'algorithmic_bytes_per_launch': bytes_per_launch,
                         'bytes_per_env_step': elems * env.world.pos.element_size(),
                         'avg_launch_ms': kern_ms, 'median_launch_ms': kern_med,
                         'timing': 'hip events around each episode segment of consecutive launches / launches' if events else 'wall/steps'},
            'episode_stats': summarize_stats(stats_total, n, EPISODE_LEN),
        }
        if world_size == 1 and not args.no_cpu_baseline:
            out['cpu_baseline'] = cpu_baseline(args.cpu_baseline_seconds)
        print(json.dumps(out))
    if world_size > 1:
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
