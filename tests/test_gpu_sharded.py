"""SURVEY 8(e) on the product path: a batch sharded over ranks (contiguous global env-id ranges, no data-path
collective, one all-reduce of the episode statistics per episode) gives the statistics of the unsharded batch.

One MI355X is all the test box has, so the two ranks are two fresh child processes that share it (gloo for the
all-reduce, as tests/test_distributed_cpu.py does on the CPU; with one GPU per rank bench.py takes the same path
with backend nccl = RCCL).  Everything that depends on the env goes by GLOBAL env id: the reset RNG (mpe_reset's
env_id0) and, here, the actions (bench.py --actions by-env-id).  The host sums this replaces:
experiments/simple_train.py:284-286,357."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
COMMON = ['--steps', '50', '--warmup', '0', '--actions', 'by-env-id', '--no-cpu-baseline', '--no-other-configs']


def _json(out):
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def _single(envs, *extra):
    return _json(subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--envs', str(envs)] + COMMON + list(extra),
                                capture_output=True, text=True, timeout=300, cwd=ROOT))


def _two_ranks(envs_per_rank, *extra):
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
           '--envs', str(envs_per_rank)] + COMMON + list(extra)
    return _json(subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env))


@pytest.mark.parametrize('workload', ['simple_tag_1M_f32', 'simple_spread_64k_f64'])
def test_two_rank_statistics_equal_the_unsharded_run(workload):
    per_rank = 50_000                                   # not a multiple of the 256-env workgroup: ragged shards
    whole = _single(2 * per_rank, '--workload', workload)
    two = _two_ranks(per_rank, '--workload', workload)
    assert two['n_gpus'] == 2 and two['config']['backend'] == 'gloo' and two['config']['stats_allreduce'] == 'per episode'
    a, b = np.array(whole['episode_stats_vector']), np.array(two['episode_stats_vector'])
    n = whole['config']['agents']
    assert a[2 * n + 1] == b[2 * n + 1] == 2 * per_rank * 50          # env-steps
    assert np.any(a[:n] != 0)
    np.testing.assert_allclose(b, a, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(a[n:], b[n:])                        # collision counters: integers, exact
    # throughput of the job = envs of both ranks over the slower rank's time
    assert two['value'] == pytest.approx(2 * per_rank * n * 50 / (two['ms_per_step'] * 50e-3), rel=1e-6)


def test_graph_replay_equals_eager_stepping():
    """The HIP-graph replays of bench.py and one host call per step produce the same statistics bit for bit."""
    g = _single(30_011)
    e = _single(30_011, '--no-graph')
    assert g['config']['graph_replays'] == 2 and e['config']['eager_steps'] == 50
    assert g['episode_stats_vector'] == e['episode_stats_vector']


def test_single_rank_under_torchrun_runs_the_rccl_path():
    """`torch.distributed.run --nproc-per-node 1 bench.py --gpus 1` (backend nccl = RCCL): the barrier and the all-reduces
    of the statistics vector and of the elapsed time execute on the device communicator, with the result of the plain run.
    (More than one rank needs more than one GPU with RCCL; the two-rank test above uses gloo.)"""
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY='0')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '1', '--envs', '30011'] + COMMON
    one = _json(subprocess.run(cmd, capture_output=True, text=True, timeout=300, cwd=ROOT, env=env))
    assert one['n_gpus'] == 1 and one['config']['backend'] == 'nccl' and one['config']['stats_allreduce'] == 'per episode'
    plain = _single(30_011)
    assert plain['config']['backend'] is None
    assert one['episode_stats_vector'] == plain['episode_stats_vector']


def test_bench_gpus_2_as_typed_launches_its_own_ranks():
    """`python bench.py --gpus 2 ...` with NO launcher (the driver's plain command form): the parent times the CPU
    baseline without touching a GPU, starts the two ranks as fresh child processes, and prints ONE JSON line with
    `n_gpus == 2`, `cpu_baseline`, and the statistics of the unsharded run.  (gloo: the two ranks share the one GPU.)"""
    per_rank = 50_000
    common = [a for a in COMMON if a != '--no-cpu-baseline'] + ['--cpu-baseline-seconds', '2']
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                          '--envs', str(per_rank)] + common, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env)
    two = _json(out)
    assert two['n_gpus'] == 2 and two['config']['backend'] == 'gloo' and two['config']['stats_allreduce'] == 'per episode'
    assert two['config']['parallelism'] == 'env-sharded dp2' and 'launched_by' in two['config']
    c = two['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 1000 and c['all_cores']['value'] > 1000
    whole = _single(2 * per_rank)
    a, b = np.array(whole['episode_stats_vector']), np.array(two['episode_stats_vector'])
    n = whole['config']['agents']
    assert a[2 * n + 1] == b[2 * n + 1] == 2 * per_rank * 50
    np.testing.assert_allclose(b, a, rtol=1e-12, atol=0)
    np.testing.assert_array_equal(a[n:], b[n:])


def test_bench_gpus_n_without_enough_gpus_fails_with_a_message_not_a_traceback():
    """backend nccl needs one GPU per rank; on the one-GPU box `--gpus 2` must end non-zero before any child starts."""
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE')}
    import torch
    if torch.cuda.device_count() >= 2:
        pytest.skip('needs a box with fewer than 2 GPUs')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--no-cpu-baseline'],
                         capture_output=True, text=True, timeout=120, cwd=ROOT, env=env)
    assert out.returncode == 2 and 'one GPU per rank' in out.stderr and 'Traceback' not in out.stderr
