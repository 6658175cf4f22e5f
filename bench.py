#!/usr/bin/env python3
"""Headline benchmark: agent-steps/sec of batched simple_tag on N MI355X (BASELINE.json).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: typed without a launcher, bench.py times the CPU baseline and then starts the N ranks
itself as fresh child processes through torch.distributed.run (`self_launch`); the parent never touches a GPU.

One "step" = one MultiAgentEnv.step() over the rank's whole batch (action decode + World.step +
observation + reward + benchmark info for all agents, one fused HIP launch).  Episodes are 25 steps
(BASELINE.md); at every episode boundary the loop reads the episode statistics (deterministic on-device
reduce + one all-reduce over RCCL) and calls env.reset() (Scenario.reset_world + observation gather).  The
timed region is phased so that it always contains such a boundary (the JSON line says how many).  Inputs
(state, a rotating pool of action tensors) are resident in HBM before the timed region.

The runs of consecutive steps between boundaries are replayed as HIP graphs (marl_amd/graph.py; --no-graph
= one Python call per step): for small batches the host path of a step costs more than its kernel.

Prints ONE JSON line on rank 0 (contract in the task description) with `roofline` (HBM, fused env-step
kernel, HIP-event brackets around the launches inside the timed region), `cpu_baseline` (the oracle's serial
numpy loop on one host core and on all of the box's cores; timed once, by rank 0 or by the self-launching parent,
before that process touches a GPU) and, at N=1 with the default
workload, `other_configs`: short runs of the other single-GPU BASELINE configs, each with its own roofline.
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
POOL = 5                  # resident action tensors, rotated by position in the episode (divides EPISODE_LEN)

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
    # fp64 (parity mode) forms of the larger worlds
    'L1F2_3Targets_1M_f64': ('L1F2_3Targets_100observe', {}, 1 << 20, torch.float64),
    'simple_world_comm_1M_f64': ('simple_world_comm', {}, 1 << 20, torch.float64),
    'S1L2F4_MovingTarget_1M_f64': ('S1L2F4_MovingTarget', {}, 1 << 20, torch.float64),
    'scaled_tag_64k_f64': ('simple_tag', dict(num_adversaries=48, num_good_agents=16, num_landmarks=32),
                           1 << 16, torch.float64),
}
# the other single-GPU configurations of BASELINE.json (+ the fp64 parity mode at the headline size): (name, steps, warmup)
OTHER_CONFIGS = (('simple_spread_64k_f64', 500, 50), ('scaled_tag_256k_f32', 100, 25), ('simple_tag_1M_f64', 250, 25))


def algorithmic_elems(env, with_info=True):
    """compulsory elements moved per env-step:
    act rows (5A, + dim_c per non-silent agent) + 2E pos rd + 2A vel rd + 4A pos,vel wr + sum obs_dim
    + A reward + A*info_w (+ 2A leader_force wr in a command hierarchy and dim_c comm wr per SPEAKING agent: a silent
    agent's state.c stays the zeros it holds, MPE_FLAG_SILENT_COMM_ZERO)."""
    w = env.world
    A, E = len(w.agents), len(w.entities)
    obs = sum(s.shape[0] for s in env.observation_space)
    info_w = w.info.shape[1] if (with_info and w.info is not None and env.info_callback is not None) else 0
    # agent.state.c of a speaking agent is rewritten each step (core.py:224-230); a silent one's rows keep their zeros
    extra = (2 * A if w.lforce is not None else 0) + w.dim_c * sum(1 for a in w.agents if not a.silent)
    moving = sum(1 for a in w.agents if a.movable)           # only movable agents have / update a velocity
    pos_rd = 2 * E if moving else 0                          # simple_crypto: nobody moves, nothing looks at positions
    # u_noise normals supplied as a tensor are 2A more rows to read; drawn inside the kernel they are no traffic at all
    noise_rd = 2 * A if (w.host_noise or w.noise is not None) and any(a.u_noise for a in w.agents) else 0
    return sum(w.act_dims) + pos_rd + 2 * moving + 4 * moving + obs + A + A * info_w + extra + noise_rd


def measured_pmc(workload):
    """The fused launch's entry in the committed rocprofv3 PMC summary of this workload (tools/profile.sh ->
    tools/summarize_pmc.py): `traffic_bytes` = HBM bytes per launch (FETCH_SIZE x2 + WRITE_SIZE, the guide's gfx950
    correction) and the `sq` block's `valu_util` (SQ_ACTIVE_INST_VALU x 4 / (1024 SIMDs x kernel cycles)) and
    `wave_wait_frac` (SQ_WAIT_ANY / SQ_WAVE_CYCLES); {} when no pass was recorded for this workload."""
    path = os.path.join(ROOT, 'profiles', 'pmc_%s.json' % workload)
    if not os.path.exists(path):
        return {}
    ks = json.load(open(path))['kernels']

    def is_fused(name):      # template tails: small/hier <..., DO_STEP, DO_OBS[, RES]>, tile <..., DO_STEP, DO_OBS, HIER[, KIND]>
        args = name[name.index('<') + 1:name.rindex('>(')].split(', ')
        if 'tile_kernel' in name:
            return args[3:5] == ['true', 'true']
        if args[-1].isdigit():                   # the Infinity Cache variant of the fused kernel (mpe_residency_plan)
            args = args[:-1]
        return args[-2:] == ['true', 'true']
    fused = [v for k, v in ks.items() if 'mpe::' in k and is_fused(k)]
    return fused[0] if fused else {}


# ------------------------------------------------------------------------------------------------------------
# CPU baseline: the reference's serial numpy loop, restated (oracle.SerialEnv), on the box's host cores
# ------------------------------------------------------------------------------------------------------------
def _cpu_worker(seconds, seed):
    """simple_tag, 25-step episodes incl. reset, random soft actions, benchmark=True;
    returns (env-steps, seconds)."""
    import numpy as np
    from oracle import mpe_oracle as O
    env = O.SerialEnv(O.simple_tag_spec(), benchmark=True)
    rng = np.random.RandomState(seed)
    np.random.seed(seed)
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
            return steps, dt


def _cpu_model():
    try:
        for ln in open('/proc/cpuinfo'):
            if ln.startswith('model name'):
                return ln.split(':', 1)[1].strip()
    except OSError:
        pass
    return 'unknown'


def host_cores():
    """Cores this job may use: the affinity mask, capped by the box's per-job thread setting (the GPU box shows the
    whole machine in os.cpu_count() and sets OMP_NUM_THREADS to the job's share)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    cap = os.environ.get('OMP_NUM_THREADS')
    if cap and cap.isdigit() and int(cap) > 0:
        n = min(n, int(cap))
    return max(1, n)


def cpu_baseline(seconds=12.0):
    """Two legs, half the time each: ONE core (the reference is a serial single-thread loop), then the same loop
    in `host_cores()` independent processes.  Must run before this process initialises the GPU (the
    workers are forked)."""
    import multiprocessing as mp
    A = 4
    s1, d1 = _cpu_worker(seconds / 2, 0)
    cores = host_cores()
    with mp.get_context('fork').Pool(cores) as pool:
        res = pool.starmap(_cpu_worker, [(seconds / 2, 100 + i) for i in range(cores)])
    multi = sum(s * A / d for s, d in res)
    return {'value': s1 * A / d1, 'unit': 'agent-steps/s', 'cores': 1, 'kind': 'port',
            'sample': '%d env-steps of simple_tag (A=4, L=2), 25-step episodes incl. reset, benchmark=True, '
                      'serial numpy loop (oracle.SerialEnv) on 1 host core' % s1,
            'all_cores': {'value': multi, 'unit': 'agent-steps/s', 'cores': cores,
                          'sample': '%d env-steps, the same loop in %d independent processes'
                                    % (sum(s for s, _ in res), cores)},
            'cpu_model': _cpu_model(), 'os_cpu_count': os.cpu_count()}


# ------------------------------------------------------------------------------------------------------------
# one workload
# ------------------------------------------------------------------------------------------------------------
def plan_segments(pos0, nsteps):
    """The step positions pos0 .. pos0 + nsteps - 1 (position in the episode = index mod EPISODE_LEN) cut at the
    episode boundaries: [(boundary_before, pos, length), ...]."""
    out, t = [], pos0
    end = pos0 + nsteps
    while t < end:
        pos = t % EPISODE_LEN
        n = min(EPISODE_LEN - pos, end - t)
        out.append((pos == 0, pos, n))
        t += n
    return out


def run_workload(name, steps, warmup, B, device, rank, world_size, dist, backend, use_graph=True, use_events=True,
                 actions='random', u_noise=0.0, host_noise=False):
    from marl_amd import make_env, _hip
    from marl_amd.distributed import EpisodeStats, shard_range, summarize_stats
    from marl_amd.graph import CapturedSteps

    scenario, kwargs, envs_per_gpu, dtype = WORKLOADS[name]
    B = B or envs_per_gpu
    env_id0, _ = shard_range(B * world_size, rank, world_size)      # weak scaling: B per GPU fixed
    # scenarios without a (working) benchmark_data
    has_info = scenario not in ('simple', 'simple_push', 'simple_speaker_listener', 'simple_reference')
    env = make_env(scenario, benchmark=has_info, num_envs=B, device=device, dtype=dtype, seed=0, **kwargs)
    env.world.env_id0 = env_id0
    if u_noise:
        # core.py:147-148: every agent's force gets u_noise * N(0, 1) per step.  The normals are drawn inside the step
        # kernel, keyed by (seed, global env id, step, agent) -- or, with --host-noise, by torch's generator into a
        # [A, 2, B] buffer per step (the round-2 path: one more launch, 4A more rows of traffic)
        for a in env.world.agents:
            a.u_noise = float(u_noise)
        env.world.host_noise = bool(host_noise)
        env.world.invalidate_descriptor()
    if os.environ.get('MPE_FLAGS'):           # tuning aid: kernel selection flags (include/mpe_hip.h, MPE_FLAG_*)
        env.world.set_flags(int(os.environ['MPE_FLAGS']))
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

    def actions_by_env_id(k):
        """Pool entry k as a function of the GLOBAL env id only (float64 sin hash -> softmax per sub-space): a batch
        sharded over ranks then sees exactly the actions of the unsharded batch (tests/test_gpu_sharded.py)."""
        w = env.world
        a = w.new_action_buffer()
        rows = a.reshape(-1, a.shape[-1])                    # [act_total, B] view of the SoA rows
        gid = torch.arange(env_id0 + 1, env_id0 + 1 + B, device=device, dtype=torch.float64)
        for ag, r0, d in zip(w.agents, w.act_row0, w.act_dims):
            nu = 5 if ag.movable else 0
            for lo, hi in ((r0, r0 + nu), (r0 + nu, r0 + d)):
                if hi > lo:
                    freq = torch.arange(lo, hi, device=device, dtype=torch.float64) * 0.37 + 1.3 * k + 0.011
                    rows[lo:hi] = torch.softmax(3.0 * torch.sin(gid[None, :] * freq[:, None]), dim=0).to(dtype)
        return a

    pool = [actions_by_env_id(k) if actions == 'by-env-id' else random_actions() for k in range(POOL)]
    stats = EpisodeStats(env.world)
    stats_total = torch.zeros(2 * n + 2, dtype=torch.float64, device=device)

    # phase: the timed region starts max(1, steps // 2) steps before an episode boundary, the warmup before that
    pos_timed = (-max(1, steps // 2)) % EPISODE_LEN
    pos_warm = (pos_timed - warmup) % EPISODE_LEN
    segs_warm, segs_timed = plan_segments(pos_warm, warmup), plan_segments(pos_timed, steps)

    # prime every code path once (module load, first-launch costs, allocator, graph captures) before the W warmup
    # steps, so that the timed region is steady state whatever W is
    env.reset()
    env.step(pool[0])
    stats_total.add_(stats.read())      # also loads torch's f64 add kernel (a ~20 ms one-time cost)
    stats_total.zero_()
    stats.reset()
    graphs, graph_error = {}, None
    if use_graph:
        try:
            for _, pos, length in segs_warm + segs_timed:
                if (pos, length) not in graphs and length > 1:
                    graphs[(pos, length)] = CapturedSteps(env, [pool[(pos + k) % POOL] for k in range(length)])
        except RuntimeError as e:          # a capture that fails must not cost the run: step eagerly instead
            graphs, graph_error = {}, str(e).splitlines()[0][:200]
            torch.cuda.synchronize()
        stats.reset()
    torch.cuda.synchronize()

    counts = {'resets': 0, 'stats_reads': 0, 'graph_replays': 0, 'eager_steps': 0}
    pending = []        # the previous boundary's all-reduce, left running on RCCL's stream beside the next episode

    def run(segs, events=None, count=False):
        """events: list collecting (start, end, n_launches) HIP-event brackets around each run of consecutive
        fused env-step launches (an event between every pair of launches would itself cost ~5 us of dispatch
        gap per step)."""
        for boundary, pos, length in segs:
            if boundary:
                # deterministic reduce + one all-reduce per episode; the collective runs beside the reset and the
                # next episode's steps and is waited for at the next boundary (or at the end of the run)
                if pending:
                    stats_total.add_(EpisodeStats.finish(pending.pop()))
                pending.append(stats.read_async())
                stats.reset()
                env.reset()
                if count:
                    counts['resets'] += 1
                    counts['stats_reads'] += 1
            if events is not None:
                e0 = torch.cuda.Event(enable_timing=True)
                e0.record()
            g = graphs.get((pos, length))
            if g is not None:
                g.replay()
                if count:
                    counts['graph_replays'] += 1
            else:
                for k in range(length):
                    env.step(pool[(pos + k) % POOL])
                if count:
                    counts['eager_steps'] += length
            if events is not None:
                e1 = torch.cuda.Event(enable_timing=True)
                e1.record()
                events.append((e0, e1, length))
        if pending:
            stats_total.add_(EpisodeStats.finish(pending.pop()))

    use_dist = dist.is_initialized()

    def fence():
        if use_dist:
            dist.barrier()
        torch.cuda.synchronize()

    env.reset()
    run(segs_warm)
    stats.reset()                       # the statistics in the JSON line cover exactly the timed steps
    stats_total.zero_()
    events = [] if use_events else None
    fence()
    t0 = time.perf_counter()
    run(segs_timed, events, count=True)
    fence()
    elapsed = time.perf_counter() - t0
    if use_dist:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device if backend == 'nccl' else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    stats_total.add_(stats.read())

    elems = algorithmic_elems(env)
    esize = env.world.pos.element_size()
    bytes_per_launch = elems * esize * B
    if events:
        per = sorted(a.elapsed_time(b) / k for a, b, k in events)
        kern_ms = sum(a.elapsed_time(b) for a, b, k in events) / sum(k for a, b, k in events)
        kern_med = per[len(per) // 2]
    else:
        kern_ms = kern_med = elapsed * 1e3 / steps
    achieved = bytes_per_launch / (kern_ms * 1e-3) / 1e9
    # committed counter passes of exactly this workload at its own batch size (not re-collected by this run)
    pmc = measured_pmc(name) if B == envs_per_gpu else {}
    out = {
        'metric': 'agent-steps/sec (whole node) for batched simple_tag' if scenario == 'simple_tag' and not kwargs
                  else 'agent-steps/sec (whole node), %s' % name,
        'value': B * world_size * n * steps / elapsed, 'unit': 'agent-steps/s', 'n_gpus': world_size, 'steps': steps,
        'warmup': warmup, 'ms_per_step': elapsed * 1e3 / steps, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None,
        'dtype': 'f32' if dtype == torch.float32 else 'f64', 'data': 'synthetic',
        'config': {'workload': name, 'scenario': scenario, 'envs_per_gpu': B, 'row_stride': env.world.ld,
                   'agents': n, 'entities': len(env.world.entities), 'episode_len': EPISODE_LEN,
                   'actions': 'softmax(N(0,1)) soft one-hot, %d resident buffers rotated' % len(pool),
                   'resets_in_timed_region': counts['resets'], 'stats_reads_in_timed_region': counts['stats_reads'],
                   'timed_region_starts_at_episode_step': pos_timed,
                   'launch': 'hip graph per run of steps between episode boundaries' if graphs else 'one host call per step',
                   'graph_replays': counts['graph_replays'], 'eager_steps': counts['eager_steps'],
                   'graph_capture_error': graph_error,
                   'benchmark_info': has_info,
                   'u_noise': u_noise or None,
                   'noise_source': (None if not u_noise else 'torch generator, [A,2,B] buffer per step' if host_noise
                                    else 'in-kernel Philox4x32-10 + Box-Muller keyed by global env id'),
                   'stats_allreduce': 'per episode' if use_dist else 'per episode (single rank: local reduce)',
                   'parallelism': 'env-sharded dp%d' % world_size, 'backend': backend if use_dist else None},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS,
                     'traffic': pmc.get('traffic_bytes'),
                     'valu_util': pmc.get('sq', {}).get('valu_util'),
                     'wave_wait_frac': pmc.get('sq', {}).get('wave_wait_frac'),
                     'counters_from': 'profiles/pmc_%s.json' % name if pmc else None,
                     'kernel': 'mpe env_step (fused)', 'algorithmic_bytes_per_launch': bytes_per_launch,
                     'bytes_per_env_step': elems * esize,
                     'avg_launch_ms': kern_ms, 'median_launch_ms': kern_med,
                     'timing': ('hip events around each run of consecutive launches (one graph replay) / launches'
                                if graphs else 'hip events around each run of consecutive launches / launches')
                               if events else 'wall/steps'},
        'episode_stats': summarize_stats(stats_total, n, EPISODE_LEN),
        'episode_stats_vector': [float(x) for x in stats_total.cpu()],
    }
    del graphs, env, pool, stats
    torch.cuda.empty_cache()
    return out


def self_launch(args, argv):
    """`python bench.py --gpus N` with N > 1 and no launcher environment: time the CPU baseline here, then start the N
    ranks as FRESH child processes (`python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py <same
    arguments>`, one rank per GPU over RCCL), relay rank 0's JSON line with `cpu_baseline` merged in, and return the
    children's exit status.  This process makes no GPU call at any point (nothing is exec'ed after GPU initialisation,
    and the forked CPU workers inherit no GPU state)."""
    import socket
    import subprocess
    if args.backend == 'nccl' and torch.cuda.device_count() < args.gpus:      # device_count() does not initialise HIP
        print('bench.py: --gpus %d with backend nccl (RCCL) needs one GPU per rank; %d visible'
              % (args.gpus, torch.cuda.device_count()), file=sys.stderr)
        return 2
    cpu = None if args.no_cpu_baseline else cpu_baseline(args.cpu_baseline_seconds)
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    for k in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE'):
        env.pop(k, None)
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(args.gpus),
           '--master-addr', '127.0.0.1', '--master-port', str(port), os.path.abspath(__file__)] + list(argv)
    if not args.no_cpu_baseline:
        cmd.append('--no-cpu-baseline')
    child = subprocess.run(cmd, stdout=subprocess.PIPE, text=True, env=env, cwd=ROOT)     # stderr passes through
    lines = [ln for ln in child.stdout.splitlines() if ln.startswith('{')]
    for ln in child.stdout.splitlines():
        if not ln.startswith('{'):
            print(ln, file=sys.stderr)
    if child.returncode != 0:
        print('bench.py: the %d-rank child job ended with status %d' % (args.gpus, child.returncode), file=sys.stderr)
        return child.returncode if 0 < child.returncode < 256 else 1
    if len(lines) != 1:
        print('bench.py: expected one JSON line from rank 0, got %d' % len(lines), file=sys.stderr)
        return 1
    out = json.loads(lines[0])
    out['config']['launched_by'] = 'bench.py (fresh child ranks via torch.distributed.run)'
    if cpu is not None:
        out['cpu_baseline'] = cpu
    print(json.dumps(out))
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=500)
    ap.add_argument('--warmup', type=int, default=50)
    ap.add_argument('--workload', default='simple_tag_1M_f32', choices=sorted(WORKLOADS))
    ap.add_argument('--envs', type=int, default=0, help='envs per GPU (default: the workload\'s)')
    ap.add_argument('--backend', default='nccl', choices=['nccl', 'gloo'],
                    help='nccl = RCCL over xGMI (one GPU per rank); gloo = rehearsal of the multi-rank path '
                         'with every rank on whatever GPUs exist (ranks then share devices)')
    ap.add_argument('--cpu-baseline-seconds', type=float, default=12.0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-other-configs', action='store_true', help='skip the short runs of the other BASELINE configs')
    ap.add_argument('--no-events', action='store_true', help='skip per-launch HIP events (roofline from wall time)')
    ap.add_argument('--no-graph', action='store_true', help='one host call per step instead of HIP-graph replays')
    ap.add_argument('--u-noise', type=float, default=0.0, help='agent.u_noise of every agent (core.py:147); default none')
    ap.add_argument('--host-noise', action='store_true', help='draw u_noise with torch\'s generator instead of in the kernel')
    ap.add_argument('--actions', default='random', choices=['random', 'by-env-id'],
                    help='by-env-id: the action of an env depends on its global id only (sharding-invariant runs)')
    args = ap.parse_args()

    rank = int(os.environ.get('RANK', 0))
    local_rank = int(os.environ.get('LOCAL_RANK', 0))
    world_size = int(os.environ.get('WORLD_SIZE', 1))
    if 'RANK' not in os.environ and args.gpus > 1:
        # `python bench.py --gpus N` as typed: this process becomes the launcher and never touches a GPU
        raise SystemExit(self_launch(args, sys.argv[1:]))
    args.gpus = world_size            # under a launcher the launcher's rank count is the truth

    cpu = None
    if rank == 0 and not args.no_cpu_baseline:
        # forks workers: before anything in this process touches the GPU.  Under a launcher with N > 1 the other
        # ranks wait for rank 0 inside init_process_group meanwhile (its default timeout is minutes, this is seconds).
        cpu = cpu_baseline(args.cpu_baseline_seconds)

    dev_index = local_rank if args.backend == 'nccl' else local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_index)
    device = torch.device('cuda', dev_index)
    import torch.distributed as dist
    # under torch.distributed.run the process group is initialised even for ONE rank, so that `--gpus 1` launched by
    # the driver's own command line runs the same RCCL calls (barrier, all-reduce of the statistics and of the time)
    # as the N > 1 runs
    launched = 'RANK' in os.environ and 'MASTER_ADDR' in os.environ
    if world_size > 1 or launched:
        if args.backend == 'nccl':
            dist.init_process_group('nccl', device_id=device)
        else:
            dist.init_process_group('gloo')

    out = run_workload(args.workload, args.steps, args.warmup, args.envs, device, rank, world_size, dist, args.backend,
                       use_graph=not args.no_graph, use_events=not args.no_events, actions=args.actions,
                       u_noise=args.u_noise, host_noise=args.host_noise)
    if rank == 0:
        if cpu is not None:
            out['cpu_baseline'] = cpu
        if (world_size == 1 and args.workload == 'simple_tag_1M_f32' and not args.envs and not args.no_other_configs
                and not args.u_noise):
            out['other_configs'] = {}
            for name, st, wu in OTHER_CONFIGS:
                r = run_workload(name, st, wu, 0, device, 0, 1, dist, args.backend, use_graph=not args.no_graph,
                                 use_events=not args.no_events)
                out['other_configs'][name] = {k: r[k] for k in ('value', 'unit', 'steps', 'warmup', 'ms_per_step', 'dtype',
                                                                 'config', 'roofline')}
        print(json.dumps(out))
    if dist.is_initialized():
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
