"""Differential fuzz (tools/fuzz_parity.py): worlds no scenario file builds -- random counts, per-entity constants, irregular
collide patterns, u_noise on some agents, random command hierarchies -- stepped by the kernel the library picks and by the
generic tile kernel, every output of 3 steps within 1e-9 of the CPU oracle (f64)."""
import importlib.util
import os

import pytest
import torch

pytestmark = pytest.mark.gpu


def test_random_worlds_match_the_oracle():
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_parity.py')
    spec = importlib.util.spec_from_file_location('fuzz_parity', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, refused, kinds = mod.run(400, 0)
    assert bad == 0 and refused == 0
    bad, _, _ = mod.run(30, 1, perturb=1e-6)            # negative control: a 1e-6 change of the damping is noticed
    assert bad >= 50
    assert set(kinds) == {'tag', 'spread', 'simple', 'adversary', 'push', 'hier', 'wcomm'} and min(kinds.values()) >= 20
