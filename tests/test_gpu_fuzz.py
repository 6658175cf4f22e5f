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
    assert set(kinds) == {'tag', 'spread', 'simple', 'adversary', 'push', 'hier', 'wcomm', 'big', 'stock_hier', 'moving'}
    assert min(kinds.values()) >= 20


def test_stock_shapes_with_random_constants_match_the_oracle():
    """The shapes that have register kernels (nine hierarchy shapes, nine stock scenarios), with random sizes / masses / limits / radii / command links / reward
    text / world constants, on all three kernel forms (one agent per wavefront, one env per lane, tile): <= 1e-9."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_parity.py')
    spec = importlib.util.spec_from_file_location('fuzz_parity', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, refused, kinds = mod.run(300, 3, stock_only=True)
    assert bad == 0 and refused == 0 and set(kinds) == {'stock_hier', 'stock_small'} and min(kinds.values()) >= 100


@pytest.mark.parametrize('stock', [False, True], ids=['random', 'stock'])
def test_worlds_on_a_grid_match_the_oracle(stock):
    """The same worlds with every entity on a grid and sizes / radii on the same grid: centre distances exactly AT the contact
    and visibility thresholds (strict comparisons on both sides), exact ties in the nearest-target order (stable), and
    coincident colliding entities, whose NaN force (core.py:239-243) then runs through three steps of physics and gathers:
    NaN for NaN like the reference (Python min() over a list keeps a leading NaN, skips a later one).  f64 <= 1e-9 vs the
    oracle incl. the observe-only call on the grid state; f32: all kernel forms agree bit for bit."""
    path = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tools', 'fuzz_parity.py')
    spec = importlib.util.spec_from_file_location('fuzz_parity', path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    bad, refused, kinds = mod.run(150, 11 + stock, stock_only=stock, grid=True)
    assert bad == 0 and refused == 0 and min(kinds.values()) >= 5


@pytest.mark.parametrize('mk', ['tag', 'spread', 'hier', 'wcomm'])
def test_batch_size_edges(mk, golden_dir):
    """B = 1, around the wavefront / workgroup / tile sizes and around the small-batch switch (MPE_SPLIT_MAX_ENVS), each in
    a row stride ld > B whose padding must stay untouched: default kernel choice and tile kernel against the oracle
    (f64, 2 steps), f32 default == f32 tile bit for bit."""
    import numpy as np
    from oracle import mpe_oracle as O
    from tests.gpu_util import HipSim, dev, set_call_flags
    spec = {'tag': O.simple_tag_spec, 'spread': O.simple_spread_spec, 'wcomm': O.simple_world_comm_spec,
            'hier': lambda: O.hier_spec_from_golden(np.load(os.path.join(golden_dir, 'h_S1L2F2_3Targets_100observe.npz')))}[mk]()
    A, E = spec.n_agents, spec.n_entities
    bo = O.BatchedOracle(spec)
    rng = np.random.RandomState(11)
    for B in (1, 2, 31, 63, 64, 65, 127, 129, 255, 256, 257, 511, 513, 32767, 32768, 32769):
        ld = B + 64 + (B % 7)
        pos = rng.uniform(-0.5, 0.5, (E, 2, B))
        vel = rng.uniform(-0.5, 0.5, (A, 2, B))
        acts = [rng.rand(sum(spec.act_dims), B) for _ in range(2)]
        outs = {}
        for dtype in (torch.float64, torch.float32):
            for flags in (0, 1):
                set_call_flags(flags)
                try:
                    sim = HipSim(spec, B, dtype, ld=ld)
                    sim.pos.fill_(5.0); sim.vel.fill_(5.0)
                    sim.set_state(pos, vel)
                    for a in acts:
                        act = torch.full((sum(spec.act_dims), ld), float('nan'), dtype=dtype, device='cuda')
                        act[:, :B] = dev(a, dtype)
                        sim.env_step(act.view(A, 5, ld) if sim.spec.act_dims == [5] * A else act)
                    torch.cuda.synchronize()
                    outs[(dtype, flags)] = sim.host()
                    assert (sim.pos[..., B:] == 5.0).all() and (sim.vel[..., B:] == 5.0).all(), (mk, B, flags)
                    for t in (sim.obs, sim.rew, sim.info):
                        assert torch.isnan(t[..., B:]).all(), (mk, B, flags)
                finally:
                    set_call_flags(0)
        p, v = pos, vel
        for a in acts:
            p, v, obs, rew, info = bo.env_step(p, v, a)
        want = (p, v, np.concatenate(obs, 0), rew)
        for flags in (0, 1):
            for x, y in zip(outs[(torch.float64, flags)], want):
                assert np.nanmax(np.abs(x - np.asarray(y).reshape(x.shape)), initial=0.0) <= 1e-9, (mk, B, flags)
        for x, y in zip(outs[(torch.float32, 0)], outs[(torch.float32, 1)]):
            assert np.array_equal(x, y, equal_nan=True), (mk, B)
