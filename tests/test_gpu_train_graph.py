"""marl_amd.graph.CapturedTrainSteps: K iterations of act -> step -> store recorded in one HIP graph must leave exactly
what the eager loop leaves -- replay rings, env state, observations, the trainers' and buffers' bookkeeping -- over
several replays with resets, an eager iteration and a weight update in between."""
import types

import pytest
import torch

pytestmark = pytest.mark.gpu


def _setup(B, seed=0, compact=False):
    from marl_amd import make_env
    from marl_amd.maddpg import MADDPGAgentTrainer
    env = make_env('simple_tag', num_envs=B, dtype=torch.float32, seed=seed)
    args = types.SimpleNamespace(lr=1e-2, gamma=0.95, batch_size=256, num_units=64, max_episode_len=25)
    obs_shape_n = [env.observation_space[i].shape for i in range(env.n)]
    torch.manual_seed(123)                                   # identical initial networks in both copies
    trainers = [MADDPGAgentTrainer('agent_%d' % i, obs_shape_n, env.action_space, i, args, replay_size=B * 40,
                                   seed=seed, env_id0=env.world.env_id0, replay_compact=compact)
                for i in range(env.n)]
    return env, trainers


def _iteration(env, trainers, obs_n, terminal=None):
    from marl_amd.maddpg import act_all
    from marl_amd.replay_buffer import add_all
    action_n = act_all(trainers, obs_n, out=env.action_views)
    new_obs_n, rew_n, done_n, info_n = env.step(action_n)
    add_all([tr.replay_buffer for tr in trainers], obs_n, action_n, rew_n, new_obs_n, done_n, terminal=terminal)
    return new_obs_n


def _run(B, K, graph, compact=False):
    from marl_amd.graph import CapturedTrainSteps
    env, trainers = _setup(B, compact=compact)
    obs_n = env.reset()
    obs_n = _iteration(env, trainers, obs_n)                 # rings and packed weights exist after this
    obs_n = env.reset()
    cap = CapturedTrainSteps(env, trainers, K) if graph else None

    def block(obs_n):
        if cap is not None:
            return cap.replay()[0]
        for k in range(K):
            obs_n = _iteration(env, trainers, obs_n, terminal=k == K - 1)
        return obs_n

    obs_n = block(obs_n)
    obs_n = env.reset()
    obs_n = block(obs_n)
    obs_n = _iteration(env, trainers, obs_n)                 # an eager iteration between replays
    torch.manual_seed(7)
    for tr in trainers:                                      # a weight update: the packed actors are refilled in place
        tr.preupdate()
    for tr in trainers:
        tr.update(trainers, 100)
    obs_n = block(obs_n)                                     # K odd: this block starts from the other observation buffer
    torch.cuda.synchronize()
    rings = [tr.replay_buffer._ring.clone() for tr in trainers]
    book = [(tr._act_calls, tr.replay_buffer._next_idx, len(tr.replay_buffer)) for tr in trainers]
    stored = [[x.clone() for x in tr.replay_buffer.collect()] for tr in trainers]     # (obs, act, rew, obs_next, done)
    return rings, book, env.world.pos.clone(), env.world.vel.clone(), [o.clone() for o in obs_n], stored


@pytest.mark.parametrize('K', [4, 5])
def test_captured_train_steps_equal_the_eager_loop(K):
    B = 512
    eager, captured = _run(B, K, False), _run(B, K, True)
    assert eager[1] == captured[1]                           # call counters, ring cursors, lengths
    assert eager[1][0][2] == (3 * K + 2) * B
    for a, b in zip(eager[0], captured[0]):
        assert torch.equal(a, b)                             # every stored transition, bit for bit
    assert torch.equal(eager[2], captured[2]) and torch.equal(eager[3], captured[3])
    for a, b in zip(eager[4], captured[4]):
        assert torch.equal(a, b)


@pytest.mark.parametrize('K', [4, 5])
def test_compact_rings_in_the_eager_and_the_captured_loop_hold_the_full_rings_transitions(K):
    """ReplayBuffer(compact=True) -- obs_tp1 stored once, `terminal` announced by the loop like the reference's
    experience(..., terminal) -- through add_all and through CapturedTrainSteps (whose last recorded step keeps its own
    copy): every stored transition as the trainers will sample it equals the full ring's, bit for bit."""
    B = 512
    full, eager, captured = _run(B, K, False), _run(B, K, False, compact=True), _run(B, K, True, compact=True)
    assert full[1] == eager[1] == captured[1]
    assert eager[0][0].shape[1] == full[0][0].shape[1] - 16      # the adversaries' rings lost their 16 obs_tp1 columns
    for f, e, c in zip(full[5], eager[5], captured[5]):
        for x, y, z in zip(f, e, c):
            assert torch.equal(x, y) and torch.equal(x, z)
    for a, b in zip(eager[0], captured[0]):                       # and the compact rings themselves agree
        assert torch.equal(a, b)
    assert torch.equal(full[2], captured[2]) and torch.equal(full[3], captured[3])


def test_ring_wrap_inside_a_replay():
    """The device cursor is added modulo the capacity: a replay that crosses the end of the ring wraps like the eager loop."""
    from marl_amd.graph import CapturedTrainSteps
    res = []
    for graph in (False, True):
        env, trainers = _setup(512)
        for tr in trainers:
            tr.replay_buffer._maxsize = 512 * 7               # 1 + 4 + 4 iterations of 512: wraps in the second block
        obs_n = _iteration(env, trainers, env.reset())
        cap = CapturedTrainSteps(env, trainers, 4) if graph else None
        for _ in range(2):
            if cap is not None:
                obs_n = cap.replay()[0]
            else:
                for _ in range(4):
                    obs_n = _iteration(env, trainers, obs_n)
        torch.cuda.synchronize()
        res.append(([tr.replay_buffer._ring.clone() for tr in trainers],
                    [(tr.replay_buffer._next_idx, len(tr.replay_buffer)) for tr in trainers]))
    assert res[0][1] == res[1][1] == [(512 * 2, 512 * 7)] * 4
    for a, b in zip(res[0][0], res[1][0]):
        assert torch.equal(a, b)


def test_captured_updates_equal_eager_updates(monkeypatch):
    """marl_amd.graph.CapturedUpdates: the learners' updates of a due step as one HIP-graph replay.  With the batch indices
    and the Gumbel noise made deterministic (both draw from generators whose streams differ between eager calls and a
    replay), three rounds of updates leave the same networks, targets and packed actor weights as the eager
    ``tr.update`` loop, up to Adam's capturable form keeping its step counter in f32 on the device."""
    import marl_amd.maddpg as M
    from marl_amd.graph import CapturedUpdates
    from marl_amd.replay_buffer import ReplayBuffer

    def soft_sample_fixed(flat, sizes, generator=None):                # u = 1/2 everywhere: a constant shift, no noise
        return torch.cat([torch.softmax(p, dim=-1) for p in torch.split(flat, sizes, dim=-1)], dim=-1)
    calls = {'n': 0}

    def make_index_fixed(self, batch_size):
        calls['n'] += 1
        return (torch.arange(batch_size, device=self.device, dtype=torch.int64) * 37 + 11 * (calls['n'] % 4 + 1)) % self._len
    monkeypatch.setattr(M, 'soft_sample', soft_sample_fixed)
    monkeypatch.setattr(ReplayBuffer, 'make_index', make_index_fixed)

    def build():
        env, trainers = _setup(512)
        obs_n = env.reset()
        for _ in range(14):                                            # 14 x 512 transitions >= 256 x 25
            obs_n = _iteration(env, trainers, obs_n)
        return env, trainers
    (_, ta), (_, tb) = build(), build()
    calls['n'] = 0
    outs_a = []
    for t in (100, 200, 300):
        for tr in ta:
            tr.preupdate()
        outs_a.append([tr.update(ta, t) for tr in ta])
    calls['n'] = 0
    cap = CapturedUpdates(tb)                                          # performs the first due update itself
    outs_b = [[[x.clone() for x in o] for o in cap.last]]
    for _ in range(2):
        outs_b.append([[x.clone() for x in o] for o in cap.replay()])
    torch.cuda.synchronize()
    # Adam divides by sqrt(v): rounding-level differences of a gradient (the capturable form's f32 step counter, the order
    # of a reduction) come back as ~1e-4 in a weight after three steps of lr = 1e-2, against steps of ~1e-2 each
    init = _setup(512)[1]
    for a, b, c in zip(ta, tb, init):
        for net in ('p', 'q', 'target_p', 'target_q'):
            for x, y, z in zip(*[[w.detach() for w in getattr(tr, net).parameters()] for tr in (a, b, c)]):
                assert float((x - y).abs().max()) <= 2e-3, (a.name, net)
                if net in ('p', 'q') and x.dim() == 2:
                    assert float((x - z).abs().max()) >= 1e-2 and float((x - y).abs().max()) < 0.1 * float((x - z).abs().max())
        assert float((a._packed - b._packed).abs().max()) <= 2e-3
        assert torch.equal(a.replay_sample_index, b.replay_sample_index)
    for ra, rb in zip(outs_a, outs_b):
        for oa, ob in zip(ra, rb):
            for x, y in zip(oa, ob):
                assert torch.allclose(x, y, rtol=5e-2, atol=1e-3)
    assert not torch.allclose(outs_b[0][0][0], outs_b[2][0][0])        # the replays really train (the loss moves)
    with pytest.raises(RuntimeError):
        CapturedUpdates(_setup(64)[1])                                 # empty buffers: no update is due (maddpg.py:162)
