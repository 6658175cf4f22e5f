"""bench.py contract: one JSON line with the driver's keys plus `roofline`, `cpu_baseline` and (default workload)
`other_configs`; what the line says about the timed region is computed from the loop, not asserted."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, timeout=600):
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + list(args), capture_output=True, text=True,
                         timeout=timeout, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    return json.loads(lines[0])


def _check_roofline(r, bytes_per_env_step):
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] < 1
    assert r['bytes_per_env_step'] == bytes_per_env_step


@pytest.mark.parametrize('mode', ['graph', 'eager'])
def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline(mode):
    d = _bench('--envs', '65536', '--steps', '50', '--warmup', '5', '--cpu-baseline-seconds', '2',
               *(['--no-graph'] if mode == 'eager' else []))
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 50 and d['warmup'] == 5 and d['scaling'] == 'weak'
    assert d['unit'] == 'agent-steps/s' and d['dtype'] == 'f32' and d['vs_baseline'] is None
    cfg = d['config']
    assert cfg['workload'] == 'simple_tag_1M_f32' and 'model' not in cfg and 'other_configs' not in d
    # 50 steps starting 25 steps before a boundary: positions 0..24 | 0..24 -> both boundaries fall inside
    assert cfg['timed_region_starts_at_episode_step'] == 0
    assert cfg['resets_in_timed_region'] == 2 and cfg['stats_reads_in_timed_region'] == 2
    if mode == 'graph':
        assert cfg['graph_replays'] == 2 and cfg['eager_steps'] == 0 and 'graph' in cfg['launch']
    else:
        assert cfg['graph_replays'] == 0 and cfg['eager_steps'] == 50
    _check_roofline(d['roofline'], 504)
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 1000 and 'sample' in c
    assert c['all_cores']['cores'] >= 1 and c['all_cores']['value'] > 1000 and c['cpu_model']
    # whole-job value is consistent with its own step time
    assert abs(d['value'] - 65536 * 4 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    assert d['value'] > 1e9          # north-star target already met on one GPU
    # the statistics really were accumulated over the timed region and the warmup before it (episodes of 25 steps)
    assert d['episode_stats']['env_steps'] == 65536 * 50


def test_driver_command_line_has_an_episode_boundary_and_the_other_configs():
    """`bench.py --steps 20 --warmup 5` (what the driver runs): the 20 timed steps straddle an episode boundary, and
    every other single-GPU BASELINE config gets a short run with its own roofline block."""
    d = _bench('--steps', '20', '--warmup', '5', '--cpu-baseline-seconds', '2')
    cfg = d['config']
    assert cfg['envs_per_gpu'] == 1 << 20 and cfg['timed_region_starts_at_episode_step'] == 15
    assert cfg['resets_in_timed_region'] == 1 and cfg['stats_reads_in_timed_region'] == 1
    _check_roofline(d['roofline'], 504)
    assert d['roofline']['frac'] > 0.5
    # north_star: "HBM GB/s and VALU utilisation from rocprof": the committed counter passes ride in the line
    for r in [d['roofline']] + [o['roofline'] for o in d['other_configs'].values()]:
        assert r['counters_from'].startswith('profiles/pmc_') and 0 < r['valu_util'] < 1 and 0 < r['wave_wait_frac'] < 1
        assert 0.8 < r['traffic'] / r['algorithmic_bytes_per_launch'] < 1.2
    oc = d['other_configs']
    assert set(oc) == {'simple_spread_64k_f64', 'scaled_tag_256k_f32', 'simple_tag_1M_f64'}
    _check_roofline(oc['simple_spread_64k_f64']['roofline'], 912)
    _check_roofline(oc['scaled_tag_256k_f32']['roofline'], 61824)
    _check_roofline(oc['simple_tag_1M_f64']['roofline'], 1008)
    for name, r in oc.items():
        assert r['config']['resets_in_timed_region'] >= 1 and r['value'] > 1e9, name
    assert oc['simple_spread_64k_f64']['dtype'] == 'f64' and oc['simple_spread_64k_f64']['config']['envs_per_gpu'] == 1 << 16
