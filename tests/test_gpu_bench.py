"""bench.py contract: one JSON line with the driver's keys plus `roofline` and `cpu_baseline`."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bench_prints_one_json_line_with_roofline_and_cpu_baseline():
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--envs', '65536', '--steps', '50',
                          '--warmup', '5', '--cpu-baseline-seconds', '1'], capture_output=True, text=True,
                         timeout=300, cwd=ROOT)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1
    d = json.loads(lines[0])
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better',
              'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline', 'cpu_baseline'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 50 and d['warmup'] == 5 and d['scaling'] == 'weak'
    assert d['unit'] == 'agent-steps/s' and d['dtype'] == 'f32' and d['vs_baseline'] is None
    assert d['config']['workload'] == 'simple_tag_1M_f32' and 'model' not in d['config']
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0
    assert abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12 and 0 < r['frac'] < 1
    assert r['bytes_per_env_step'] == 504
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['value'] > 1000 and 'sample' in c
    # whole-job value is consistent with its own step time
    assert abs(d['value'] - 65536 * 4 / (d['ms_per_step'] * 1e-3)) / d['value'] < 1e-6
    assert d['value'] > 1e9          # north-star target already met on one GPU
