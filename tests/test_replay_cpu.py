"""Replay buffer, CPU side (first slice).  PARITY UNPINNED: the reference
(maddpg/maddpg/trainer/replay_buffer.py) has no tests or fixtures for this class and importing it to make
golden vectors was refused here.  These tests pin the restatement in
oracle/replay_oracle.py to hand-worked cases of the reference text, and check the C ABI's argument
validation (no launches)."""
import ctypes as C

import numpy as np

from oracle.replay_oracle import ReplayBuffer as OracleBuffer


def _t(i):
    return (np.array([i, i + 0.5]), np.array([10.0 * i]), float(-i), np.array([i + 1, i + 1.5]), float(i % 2))


def test_oracle_ring_semantics_by_hand():
    rb = OracleBuffer(4)
    assert len(rb) == 0 and rb._next_idx == 0
    for i in range(3):
        rb.add(*_t(i))
    assert len(rb) == 3 and rb._next_idx == 3                 # replay_buffer.py:28-32: append while it grows
    rb.add(*_t(3))
    assert len(rb) == 4 and rb._next_idx == 0                 # wraps at maxsize
    rb.add(*_t(4))                                            # overwrites slot 0
    rb.add(*_t(5))
    assert len(rb) == 4 and rb._next_idx == 2
    obs, act, rew, nxt, done = rb.collect()                   # storage order
    assert rew.tolist() == [-4.0, -5.0, -2.0, -3.0] and act[:, 0].tolist() == [40.0, 50.0, 20.0, 30.0]
    assert obs.shape == (4, 2) and nxt[0].tolist() == [5.0, 5.5] and done.tolist() == [0.0, 1.0, 0.0, 1.0]
    o, a, r, n2, d = rb.sample_index([2, 2, 0])               # duplicates allowed, order kept
    assert r.tolist() == [-2.0, -2.0, -4.0] and o[2].tolist() == [4.0, 4.5]
    lat = rb.make_latest_index(3)                             # :50: slots 1, 0, 3 (shuffled)
    assert sorted(lat) == [0, 1, 3]
    idx = rb.make_index(50)
    assert min(idx) >= 0 and max(idx) <= 3
    rb.clear()
    assert len(rb) == 0 and rb._next_idx == 0


def test_oracle_batch_add_equals_sequential_adds():
    rng = np.random.RandomState(0)
    a, b = OracleBuffer(10), OracleBuffer(10)
    for step in range(5):
        obs, act, rew, nxt, done = rng.rand(7, 3), rng.rand(7, 2), rng.randn(7), rng.rand(7, 3), rng.rand(7) < 0.3
        a.add_batch(obs, act, rew, nxt, done)
        for e in range(7):
            b.add(obs[e], act[e], rew[e], nxt[e], float(done[e]))
    assert a._next_idx == b._next_idx == 5 and len(a) == 10
    for x, y in zip(a.collect(), b.collect()):
        assert np.array_equal(x, y)


def test_replay_c_abi_argument_checks():
    from marl_amd import _hip
    lib = _hip.lib
    seg = (_hip.ReplaySeg * 2)(_hip.ReplaySeg(8, 3, 16, 1), _hip.ReplaySeg(8, 1, 16, 1))
    ok = dict(capacity=10, next_idx=0, n=4, width=4, nseg=2, segs=seg, ring=8, stream=None)
    def add(**kw):
        a = dict(ok)
        a.update(kw)
        return lib.mpe_replay_add_f32(a['capacity'], a['next_idx'], a['n'], a['width'], a['nseg'], a['segs'], a['ring'], a['stream'])
    assert add(ring=None) == -1 and add(segs=None) == -1
    assert add(capacity=0) == -6 and add(next_idx=10) == -6 and add(n=-1) == -6 and add(nseg=9) == -6
    assert add(width=5) == -6                                 # segment rows must add up to the record width
    assert add(ring=6) == -4
    assert add(n=0) == 0                                      # nothing to do, nothing launched
    bad = (_hip.ReplaySeg * 2)(_hip.ReplaySeg(None, 3, 16, 1), _hip.ReplaySeg(8, 1, 16, 1))
    assert add(segs=bad) == -1
    g = lib.mpe_replay_gather_f32
    assert g(10, 4, None, 4, 8, 8, None) == -1 and g(10, 4, 8, 4, None, 8, None) == -1
    assert g(0, 4, 8, 4, 8, 8, None) == -6 and g(10, 0, 8, 4, 8, 8, None) == -6 and g(10, 4, 8, -1, 8, 8, None) == -6
    assert g(10, 4, 4, 4, 8, 8, None) == -4 and g(10, 4, 8, 0, 8, 8, None) == 0
