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
