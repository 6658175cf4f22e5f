"""Buffers beyond 2^31 / 2^32 bytes in the register kernels (tools/big_batch_check.py): 9 Mi envs of simple_tag (f32: 2.3 GB
of observation rows), simple_world_comm (7.3 GB) and S1L2F2_10Targets (5.2 GB) -- one fused step of the whole batch
equals, bit for bit, the same step of three 1 500-env slices run as batches of their own.  (The tile kernel passes
2^32 bytes at BASELINE config 5 already: tests/test_gpu_parity.py.)"""
import importlib.util
import os

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize('which', ['simple_tag', 'simple_world_comm', 'S1L2F2_10Targets_wide'])
def test_nine_million_envs_equal_their_slices(which, golden_dir):
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'big_batch_check.py')
    spec = importlib.util.spec_from_file_location('big_batch_check', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    sc = {'simple_tag': O.simple_tag_spec, 'simple_world_comm': O.simple_world_comm_spec}.get(which)
    sc = sc() if sc else O.hier_spec_from_golden(np.load(os.path.join(golden_dir, 'h_%s.npz' % which)))
    B = 9 * 1048576 + 77
    assert sum(sc.obs_dims) * B * 4 > 2 ** 31
    assert mod.run(sc, B, torch.float32) == 0
    torch.cuda.empty_cache()
