"""The small-batch form of the simple_tag / simple_spread step (marl_amd/csrc/mpe_split_kernel.h: one agent per
wavefront) against the one-env-per-lane kernel: same bits for state, observations, rewards and info from every entry
point, with and without u_noise, on ragged batches inside padded allocations; statistics rows per 64 envs."""
import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from marl_amd import _hip, make_env
    from tests.gpu_util import HipSim, dev, lib, set_call_flags, stats_rows

WIDE, SPLIT = 2, 4


@pytest.fixture(autouse=True)
def _default_flags():
    yield
    set_call_flags(0)


SPECS = [('tag', lambda: O.simple_tag_spec()), ('spread', lambda: O.simple_spread_spec()),
         ('tag_unoise', lambda: O.simple_tag_spec(u_noise=0.3)), ('spread_unoise', lambda: O.simple_spread_spec(u_noise=0.2))]


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('name,mk', SPECS, ids=[n for n, _ in SPECS])
def test_split_equals_wide_bit_for_bit(name, mk, dtype):
    spec = mk()
    A, E = spec.n_agents, spec.n_entities
    B, ld = 64 * 9 + 37, 1024                      # a ragged last workgroup, padding beyond B
    rng = np.random.RandomState(5)
    pos = rng.uniform(-0.35, 0.35, (E, 2, B))      # dense: plenty of contacts
    pos[0, 0, :50] = 1.05                          # some agents outside the arena (bound() penalty of the prey...)
    pos[A - 1, 1, 50:90] = -1.3                    # ... and the prey itself
    vel = rng.uniform(-1, 1, (A, 2, B))
    acts = [dev(np.pad(rng.rand(A, 5, B), ((0, 0), (0, 0), (0, ld - B))), dtype) for _ in range(4)]
    noises = [dev(np.pad(rng.randn(A, 2, B), ((0, 0), (0, 0), (0, ld - B))), dtype) for _ in range(4)]
    use_noise = 'unoise' in name
    W = 2 * A + 2
    out = {}
    for flags in (WIDE, SPLIT):
        set_call_flags(flags)
        fused, parts = HipSim(spec, B, dtype, ld), HipSim(spec, B, dtype, ld)
        rows = stats_rows(fused.desc, B, fused.pos.element_size())
        assert rows == -(-B // (64 if flags == SPLIT else 256))
        stats = torch.zeros(rows, W, dtype=torch.float64, device='cuda')
        comm = torch.full((A, spec.dim_c, ld), 7.0, dtype=dtype, device='cuda')
        for s in (fused, parts):
            s.set_state(pos, vel)
        for a, n in zip(acts, noises):
            n = n if use_noise else None
            fused.env_step(a, noise=n, stats=stats, comm=comm)
            parts.world_step(a, noise=n)
            parts.observe()
        for x, y in zip(fused.host(), parts.host()):           # world_step + observe == env_step on this kernel too
            assert np.array_equal(x, y, equal_nan=True)
        assert (comm[:, :, :B] == 0).all() and (comm[:, :, B:] == 7).all()
        for t in (fused.obs, fused.rew, fused.info):
            assert torch.isnan(t[..., B:]).all()               # padding never written
        red = torch.empty(W, dtype=torch.float64, device='cuda')
        assert lib.mpe_stats_reduce(stats.data_ptr(), rows, W, red.data_ptr(), fused.stream) == 0
        out[flags] = (fused.host(), red.cpu().numpy())
    for x, y in zip(out[WIDE][0], out[SPLIT][0]):
        assert np.array_equal(x, y, equal_nan=True)
    sw, ss = out[WIDE][1], out[SPLIT][1]
    assert np.array_equal(sw[A:], ss[A:])                      # contact counters, env count: integers
    np.testing.assert_allclose(ss[:A], sw[:A], rtol=1e-12)     # reward sums: 64-env groups instead of 256
    assert sw[2 * A + 1] == 4 * B and sw[A:2 * A].sum() > 0


def test_default_selection_by_batch_size():
    """B <= MPE_SPLIT_MAX_ENVS runs one agent per wavefront (statistics rows per 64 envs), larger batches one env per
    lane; worlds without a split form ignore the two flags."""
    tag, adv = HipSim(O.simple_tag_spec(), 8, torch.float32).desc, HipSim(O.simple_adversary_spec(), 8, torch.float32).desc
    n = _hip.SPLIT_MAX_ENVS
    rows = lambda d, B, f: lib.mpe_stats_rows(d, B, 4, f)
    assert rows(tag, n, 0) == n // 64 and rows(tag, n + 1, 0) == -(-(n + 1) // 256)
    assert rows(tag, 1 << 20, SPLIT) == (1 << 20) // 64 and rows(tag, 4096, WIDE) == 16
    assert rows(adv, 4096, 0) == rows(adv, 4096, SPLIT) == rows(adv, 4096, WIDE) == 16
    assert rows(tag, 4096, 32) < 0                              # unknown flag bit


@pytest.mark.parametrize('scenario', ['simple_tag', 'simple_spread'])
def test_python_api_small_batch_matches_wide(scenario):
    """make_env at a small batch (the split kernel by default) against the same env forced onto the wide kernel."""
    B = 1000
    res = []
    for flags in (0, WIDE):
        env = make_env(scenario, benchmark=True, num_envs=B, device='cuda', dtype=torch.float64, seed=3)
        env.world.set_flags(flags)
        env.world.enable_stats()
        obs = env.reset()
        g = torch.Generator(device='cuda').manual_seed(9)
        for _ in range(5):
            act = [torch.softmax(torch.randn(B, 5, generator=g, device='cuda', dtype=torch.float64), dim=1) for _ in range(env.n)]
            obs, rew, done, info = env.step(act)
        res.append((torch.stack([o.sum(1) for o in obs]), torch.stack(list(rew)), env.world.pos.clone(),
                    env.world.read_stats()))
    for x, y in zip(res[0][:3], res[1][:3]):
        assert torch.equal(x, y)
    torch.testing.assert_close(res[0][3], res[1][3], rtol=1e-12, atol=0)
