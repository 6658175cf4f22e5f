"""GPU tests of the on-device replay buffer (first slice) against the CPU restatement of
maddpg/maddpg/trainer/replay_buffer.py (oracle/replay_oracle.py; PARITY UNPINNED, see its header): pure data
movement, so every comparison is bit for bit."""
import numpy as np
import pytest
import torch

from oracle.replay_oracle import ReplayBuffer as OracleBuffer

pytestmark = pytest.mark.gpu


def _np(t):
    return t.detach().cpu().numpy()


def _check(rb, orb, idx):
    got = rb.sample_index(idx)
    want = orb.sample_index([int(i) for i in idx])
    for g, w in zip(got, want):
        assert np.array_equal(_np(g), w.astype(_np(g).dtype))


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('capacity,B,od,ad', [(50, 7, 6, 5), (64, 16, 3, 2), (40, 100, 4, 9), (1000, 129, 70, 12),
                                                  (300, 70, 100, 5), (1000, 600, 6, 5), (5000, 777, 20, 9)])
def test_ring_add_and_sample_match_the_oracle(dtype, capacity, B, od, ad):
    """Wrap-around, exact fill, a step larger than the whole ring, records on both kernels (up to 160 columns:
    whole tile in LDS, flat store, 256 transitions per workgroup when a step has that many and the tile fits;
    wider: 64 columns per pass);
    sources in every layout the loop produces: SoA row views (stride (1, ld)), contiguous [B, d], 1-D."""
    from marl_amd.replay_buffer import ReplayBuffer
    npdt = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.RandomState(capacity)
    rb, orb = ReplayBuffer(capacity, dtype=dtype), OracleBuffer(capacity)
    ld = B + 13
    for step in range(9):
        obs, act = rng.rand(B, od).astype(npdt), rng.rand(B, ad).astype(npdt)
        rew, nxt, done = rng.randn(B).astype(npdt), rng.rand(B, od).astype(npdt), rng.rand(B) < 0.2
        soa = torch.zeros(od, ld, dtype=dtype, device='cuda')             # like env.step's observation rows
        soa[:, :B] = torch.as_tensor(obs.T).cuda()
        rb.add(soa[:, :B].t(), torch.as_tensor(act).cuda(), torch.as_tensor(rew).cuda(),
               torch.as_tensor(nxt).cuda(), torch.as_tensor(done).cuda())
        orb.add_batch(obs, act, rew, nxt, done)
        assert len(rb) == len(orb) and rb._next_idx == orb._next_idx
        _check(rb, orb, rng.randint(0, len(orb), 33))
    for g, w in zip(rb.collect(), orb.collect()):                         # the whole ring, storage order
        assert np.array_equal(_np(g), w.astype(npdt))
    lat = _np(rb.make_latest_index(5))
    assert sorted(lat.tolist()) == sorted((orb._next_idx - 1 - i) % capacity for i in range(5))
    idx = rb.make_index(4096)
    assert int(idx.min()) >= 0 and len(rb) // 2 < int(idx.max()) <= len(rb) - 1
    o, a, r, n2, d = rb.sample(64)
    assert tuple(o.shape) == (64, od) and tuple(a.shape) == (64, ad) and tuple(r.shape) == (64,) and tuple(d.shape) == (64,)
    rb.clear()
    assert len(rb) == 0 and rb._next_idx == 0


def test_training_loop_shape_with_the_env():
    """experience() of every agent after env.step (maddpg.py:154-156, simple_train.py:284-287), then one
    shared index list gathered from every agent's buffer (maddpg.py:167-178)."""
    from marl_amd import make_env
    from marl_amd.replay_buffer import ReplayBuffer
    B, T = 300, 6
    env = make_env('simple_tag', num_envs=B, dtype=torch.float32, seed=2)
    obs_n = env.reset()
    bufs = [ReplayBuffer(1024, dtype=torch.float32) for _ in range(env.n)]
    oracles = [OracleBuffer(1024) for _ in range(env.n)]
    gen = torch.Generator(device='cuda').manual_seed(0)
    for t in range(T):
        action_n = [torch.rand(B, 5, generator=gen, device='cuda') for _ in range(env.n)]
        want_prev = [_np(o).copy() for o in obs_n]
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        for i in range(env.n):
            # obs_n (from the previous step) is still intact: the env alternates two observation buffers
            assert np.array_equal(_np(obs_n[i]), want_prev[i]) and not np.array_equal(_np(new_obs_n[i]), want_prev[i])
            bufs[i].add(obs_n[i], action_n[i], rew_n[i], new_obs_n[i], done_n[i])
            d = done_n[i]
            d = _np(d) if torch.is_tensor(d) else np.full(B, float(d))
            oracles[i].add_batch(want_prev[i], _np(action_n[i]), _np(rew_n[i]), _np(new_obs_n[i]), d)
        obs_n = new_obs_n
    assert len(bufs[0]) == 1024 and bufs[0]._next_idx == (B * T) % 1024
    index = bufs[0].make_index(256)
    for i in range(env.n):
        _check(bufs[i], oracles[i], _np(index))


def test_out_of_range_indices_raise_like_the_reference():
    """replay_buffer.py:34-44 indexes a Python list of len(self) transitions: an index beyond it raises IndexError
    (the gather kernel wraps indices so that it cannot fault; the Python class must not hand out
    never-written rows silently)."""
    from marl_amd.replay_buffer import ReplayBuffer
    rb = ReplayBuffer(100, dtype=torch.float32)
    z = lambda *s: torch.rand(*s, device='cuda')
    rb.add(z(10, 3), z(10, 2), z(10), z(10, 3), z(10))
    rb.sample_index(torch.tensor([0, 9], device='cuda'))
    for bad in ([10], [0, 99], [-1], [100], [1 << 40]):
        with pytest.raises(IndexError):
            rb.sample_index(torch.tensor(bad, device='cuda'))
    with pytest.raises(IndexError):
        rb.make_latest_index(11)
    rb.check_index = False                                                # opt out: no host sync, kernel wraps
    rb.sample_index(torch.tensor([150], device='cuda'))


# ---- compact rings: obs_tp1 stored once (mpe_replay_gather_compact_*) -----------------------------------------------
def _episode_loop(rb, orb, rng, B, od, ad, steps, ep_len, dtype, npdt, sample_every=3):
    """The reference's loop shape: obs_n <- new_obs_n between steps, a fresh obs after every ep_len-th step (env.reset());
    the compact ring is told `terminal` like MADDPGAgentTrainer.experience; samples drawn now and then (each flushes the
    newest step).  Every sample is compared with the oracle ring fed the same transitions."""
    ld = B + 9
    bufs = [torch.zeros(od, ld, dtype=dtype, device='cuda') for _ in range(2)]     # the env's two observation buffers
    slot = 0
    obs = rng.rand(B, od).astype(npdt)
    bufs[slot][:, :B] = torch.as_tensor(obs.T).cuda()
    for t in range(steps):
        act = rng.rand(B, ad).astype(npdt)
        rew, nxt, done = rng.randn(B).astype(npdt), rng.rand(B, od).astype(npdt), rng.rand(B) < 0.2
        bufs[1 - slot][:, :B] = torch.as_tensor(nxt.T).cuda()
        terminal = (t + 1) % ep_len == 0
        rb.add(bufs[slot][:, :B].t(), torch.as_tensor(act).cuda(), torch.as_tensor(rew).cuda(),
               bufs[1 - slot][:, :B].t(), torch.as_tensor(done).cuda(), terminal=terminal)
        orb.add_batch(obs, act, rew, nxt, done)
        slot = 1 - slot
        obs = nxt
        if terminal:                                   # env.reset(): the next obs_t is unrelated to this obs_tp1
            obs = rng.rand(B, od).astype(npdt)
            slot = 1 - slot                            # (reset writes the other buffer, like World.gather_obs)
            bufs[slot][:, :B] = torch.as_tensor(obs.T).cuda()
        assert len(rb) == len(orb) and rb._next_idx == orb._next_idx
        if t % sample_every == 0:
            _check(rb, orb, rng.randint(0, len(orb), 64))
    for g, w in zip(rb.collect(), orb.collect()):
        assert np.array_equal(_np(g), w.astype(npdt))


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
@pytest.mark.parametrize('B,slots,od,ad,ep_len', [(7, 5, 6, 5, 4), (300, 4, 16, 5, 25), (64, 2, 3, 2, 3), (1000, 7, 30, 12, 5)])
def test_compact_ring_samples_equal_the_full_ring(dtype, B, slots, od, ad, ep_len):
    """obs_tp1 stored once: every sample, the whole ring and the bookkeeping equal the CPU restatement of the
    reference's buffer over several wraps of the ring, with episode ends, samples drawn right after a non-terminal
    step (the newest step has no successor yet) and right after a terminal one."""
    from marl_amd.replay_buffer import ReplayBuffer
    npdt = np.float64 if dtype == torch.float64 else np.float32
    rng = np.random.RandomState(B)
    rb, orb = ReplayBuffer(B * slots, dtype=dtype, compact=True), OracleBuffer(B * slots)
    _episode_loop(rb, orb, rng, B, od, ad, steps=3 * slots + 5, ep_len=ep_len, dtype=dtype, npdt=npdt)
    assert rb._ring.shape[1] == od + ad + 2            # the ring really is the compact one
    # terminal=None (a caller that does not say): every step keeps its own copy -- still correct
    rb2, orb2 = ReplayBuffer(B * slots, dtype=dtype, compact=True), OracleBuffer(B * slots)
    for t in range(slots + 2):
        o, a = rng.rand(B, od).astype(npdt), rng.rand(B, ad).astype(npdt)
        r, n2, d = rng.randn(B).astype(npdt), rng.rand(B, od).astype(npdt), rng.rand(B) < 0.5
        rb2.add(*[torch.as_tensor(x).cuda() for x in (o, a, r, n2, d)])
        orb2.add_batch(o, a, r, n2, d)
    for g, w in zip(rb2.collect(), orb2.collect()):
        assert np.array_equal(_np(g), w.astype(npdt))


def test_compact_ring_refuses_what_it_cannot_represent():
    from marl_amd.replay_buffer import ReplayBuffer
    z = lambda *s: torch.zeros(*s, device='cuda')
    with pytest.raises(ValueError):                    # size % n != 0
        ReplayBuffer(100, compact=True).add(z(7, 4), z(7, 5), z(7), z(7, 4), z(7))
    with pytest.raises(ValueError):                    # a single step slot: a row would be its own successor
        ReplayBuffer(8, compact=True).add(z(8, 4), z(8, 5), z(8), z(8, 4), z(8))
    rb = ReplayBuffer(64, compact=True)
    nxt = z(8, 4)
    rb.add(z(8, 4), z(8, 5), z(8), nxt, z(8), terminal=False)
    with pytest.raises(ValueError):                    # not flagged terminal, yet the next obs_t is another tensor
        rb.add(z(8, 4), z(8, 5), z(8), z(8, 4), z(8), terminal=False)
    rb.add(nxt, z(8, 5), z(8), z(8, 4), z(8), terminal=True)
    with pytest.raises(ValueError):                    # another batch size
        rb.add(z(4, 4), z(4, 5), z(4), z(4, 4), z(4))
    lib = __import__('marl_amd._hip', fromlist=['lib']).lib
    idx = torch.zeros(4, dtype=torch.int64, device='cuda')
    f = lib.mpe_replay_gather_compact_f32
    out = z(4, 15)
    args = lambda **kw: [kw.get(k, v) for k, v in (('capacity', 64), ('od', 4), ('ad', 5), ('n', 8), ('idx', idx.data_ptr()),
                                                    ('batch', 4), ('ring', rb._ring.data_ptr()), ('ends', rb._ends.data_ptr()),
                                                    ('own', rb._own.data_ptr()), ('out', out.data_ptr()), ('stream', None))]
    assert f(*args()) == 0
    assert f(*args(n=7)) == -6 and f(*args(ends=None)) == -1 and f(*args(own=None)) == -1 and f(*args(od=0)) == -6
    assert f(*args(out=out.data_ptr() + 2)) == -4
    torch.cuda.synchronize()


def test_boundary_record_widths_through_the_one_launch_append():
    """mpe_replay_add_multi(_cur)_*: the widest records whose 256-transition tile fits 64 KiB of LDS TOGETHER WITH the
    kernel's static column tables (61 f32 / 30 f64 columns), one wider (falls back to the single-buffer kernel; refused
    when a device cursor is given), against the single-buffer append, bit for bit."""
    import ctypes as C
    from marl_amd import _hip
    from marl_amd.replay_buffer import ReplayBuffer, add_all
    for dtype, wmax in ((torch.float32, 61), (torch.float64, 30)):
        for width in (wmax, wmax + 1, wmax + 2):
            od = (width - 2 - 3) // 2                   # width = 2 od + ad + 2
            ad = width - 2 - 2 * od
            B = 700
            rng = np.random.RandomState(width)
            one = [ReplayBuffer(2000, dtype=dtype) for _ in range(2)]
            multi = [ReplayBuffer(2000, dtype=dtype) for _ in range(2)]
            for step in range(4):                       # incl. a wrap of the 2 000-row ring
                data = [[torch.as_tensor(x).to(dtype).cuda() for x in
                         (rng.rand(B, od), rng.rand(B, ad), rng.randn(B), rng.rand(B, od), (rng.rand(B) < 0.3).astype(float))]
                        for _ in range(2)]
                for b, dd in zip(one, data):
                    b.add(*dd)
                if width <= wmax:
                    add_all(multi, *zip(*data), cursors=step % 2 == 1)
                else:
                    add_all(multi, *zip(*data))
            for a, b in zip(one, multi):
                assert a._ring.shape[1] == width and torch.equal(a._ring, b._ring)
            if width > wmax:                            # a device cursor there: refused before anything is launched
                before = [b._ring.clone() for b in multi]
                with pytest.raises(_hip.MpeError) as e:
                    add_all(multi, *zip(*data), cursors=True)
                assert e.value.code == -5
                torch.cuda.synchronize()
                assert all(torch.equal(x, b._ring) for x, b in zip(before, multi))
            assert multi[0].one_launch_capable(B) == (width <= wmax)
