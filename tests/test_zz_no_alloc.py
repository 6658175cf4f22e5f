"""mpe_env_step_stats_static (include/mpe_hip.h; no counterpart in the reference): the allocation-free variant of
mpe_env_step_* + mpe_stats_reduce, per-workgroup statistics rows in the library's file-scope device array.

Checked through the C ABI against the existing entry points, bit for bit (the kernels and the reduce order are the
same), on two calls in a row, the second with a smaller batch (the array keeps the first call's contents), for
both arithmetic types and both kernels (register / LDS-tile); plus the size limit.  Collected last (test_zz_*) so a
failure here cannot hide the parity tests."""
import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from marl_amd import _hip
    from tests.gpu_util import HipSim, dev, lib, stats_rows


def _state(spec, B, seed):
    rng = np.random.RandomState(seed)
    A, E = spec.n_agents, spec.n_entities
    pos = rng.uniform(-0.6, 0.6, (E, 2, B))          # crowded: contacts, rewards and counters are non-zero
    vel = rng.uniform(-0.5, 0.5, (A, 2, B))
    act = rng.dirichlet(np.ones(5), (A, B)).transpose(0, 2, 1)
    return pos, vel, np.ascontiguousarray(act)


def _pair(spec, B, dtype, seed, flags):
    """(existing entry points, new entry point) run from the same state; returns both sims and both stats vectors."""
    pos, vel, act = _state(spec, B, seed)
    nbytes = 4 if dtype == torch.float32 else 8
    a, b = HipSim(spec, B, dtype, ld=B + 37), HipSim(spec, B, dtype, ld=B + 37)
    outs = []
    for sim, static in ((a, False), (b, True)):
        sim.set_state(pos, vel)
        act_d = torch.zeros(spec.n_agents, 5, sim.ld, dtype=dtype, device='cuda')
        act_d[:, :, :B] = dev(act, dtype)
        width = lib.mpe_stats_width(sim.desc)
        out = torch.full((width,), float('nan'), dtype=torch.float64, device='cuda')
        bufs = _hip.Buffers(flags=flags, pos=sim.pos, vel=sim.vel, goal=sim.goal, act=act_d, obs=sim.obs, rew=sim.rew,
                            info=sim.info)
        if static:
            rc = lib.mpe_env_step_stats_static(sim.desc, B, sim.ld, bufs, nbytes, out.data_ptr(), sim.stream)
            assert rc == 0, lib.mpe_strerror(rc)
        else:
            rows = lib.mpe_stats_rows(sim.desc, B, nbytes, flags)
            stats = torch.zeros(rows, width, dtype=torch.float64, device='cuda')
            bufs.stats = stats.data_ptr()
            rc = getattr(lib, 'mpe_env_step_' + sim.suf)(sim.desc, B, sim.ld, bufs, sim.stream)
            assert rc == 0, lib.mpe_strerror(rc)
            rc = lib.mpe_stats_reduce(stats.data_ptr(), rows, width, out.data_ptr(), sim.stream)
            assert rc == 0, lib.mpe_strerror(rc)
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    return a, b, outs[0], outs[1]


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64])
@pytest.mark.parametrize('flags', [0, 1])
def test_two_calls_in_a_row_second_smaller_match_the_existing_entry_points(dtype, flags):
    spec = O.simple_tag_spec()
    for B, seed in ((50_000, 1), (20_011, 2)):       # the second call uses fewer rows of the shared array
        a, b, want, got = _pair(spec, B, dtype, seed, flags)
        assert np.array_equal(want, got), (B, want, got)           # bit-exact: same kernels, same reduce order
        assert want[-1] == B and want[2 * spec.n_agents] > 0       # env-steps counted, contacts happened
        for x, y in zip(a.host(), b.host()):                       # pos, vel, obs, rew, info
            assert np.array_equal(x, y, equal_nan=True)


def test_spread_and_size_limit():
    spec = O.simple_spread_spec()
    a, b, want, got = _pair(spec, 4_099, torch.float64, 3, 0)
    assert np.array_equal(want, got) and want[-1] == 4_099
    # beyond the array: rejected before anything is launched (no pointer is dereferenced)
    sim = HipSim(O.simple_tag_spec(), 8, torch.float32)
    out = torch.zeros(10, dtype=torch.float64, device='cuda')
    big = 4_000_000
    bufs = _hip.Buffers(pos=sim.pos, vel=sim.vel, act=sim.pos, obs=sim.obs, rew=sim.rew)
    assert lib.mpe_stats_rows(sim.desc, big, 4, 0) * 10 * 8 > (1 << 20)
    assert lib.mpe_env_step_stats_static(sim.desc, big, big, bufs, 4, out.data_ptr(), sim.stream) == -7
    assert b'scratch' in lib.mpe_strerror(-7)
    assert lib.mpe_env_step_stats_static(sim.desc, 8, 8, bufs, 2, out.data_ptr(), sim.stream) == -6    # dtype_bytes
    assert lib.mpe_env_step_stats_static(sim.desc, 8, 8, bufs, 4, None, sim.stream) == -1


def test_entry_point_under_hip_graph_capture_and_with_garbage_in_the_ignored_stats_member():
    """What the header advertises: after one eager call on the device (which resolves the scratch array's address) the
    entry point records under HIP graph capture -- it allocates nothing -- and a replay gives the eager call's bits.
    buf->stats is documented as ignored: a misaligned / dangling value there must not be looked at."""
    spec = O.simple_tag_spec()
    B = 30_011
    pos, vel, act = _state(spec, B, 5)
    sims = [HipSim(spec, B, torch.float32, ld=B + 5) for _ in range(2)]
    outs = []
    side = torch.cuda.Stream()
    for sim, captured in zip(sims, (False, True)):
        sim.set_state(pos, vel)
        act_d = torch.zeros(spec.n_agents, 5, sim.ld, dtype=torch.float32, device='cuda')
        act_d[:, :, :B] = dev(act, torch.float32)
        width = lib.mpe_stats_width(sim.desc)
        out = torch.full((width,), float('nan'), dtype=torch.float64, device='cuda')
        bufs = _hip.Buffers(pos=sim.pos, vel=sim.vel, goal=sim.goal, act=act_d, obs=sim.obs, rew=sim.rew, info=sim.info)
        bufs.stats = 0xDEAD0003                                            # ignored: neither aligned nor mapped
        if not captured:
            rc = lib.mpe_env_step_stats_static(sim.desc, B, sim.ld, bufs, 4, out.data_ptr(), sim.stream)
            assert rc == 0, lib.mpe_strerror(rc)
        else:
            g = torch.cuda.CUDAGraph()
            torch.cuda.synchronize()
            with torch.cuda.graph(g, stream=side, capture_error_mode='thread_local'):
                rc = lib.mpe_env_step_stats_static(sim.desc, B, sim.ld, bufs, 4, out.data_ptr(),
                                                   torch.cuda.current_stream().cuda_stream)
            assert rc == 0, lib.mpe_strerror(rc)
            assert torch.isnan(out).all()                                   # a capture executes nothing
            g.replay()
        torch.cuda.synchronize()
        outs.append(out.cpu().numpy())
    assert np.array_equal(outs[0], outs[1]) and outs[0][-1] == B
    for x, y in zip(sims[0].host(), sims[1].host()):
        assert np.array_equal(x, y, equal_nan=True)
