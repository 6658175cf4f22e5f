"""MADDPG learner (reference maddpg/maddpg/trainer/maddpg.py, common/distributions.py).
PARITY UNPINNED: TensorFlow is not installed and the reference's learner code is not run here (DESIGN.md
section 4).  CPU tests pin the restated pieces to closed forms; GPU tests run the trainer on the env + replay
data path and check that the updates do what the reference's text says."""
import types

import numpy as np
import pytest
import torch

from marl_amd import spaces
from marl_amd.maddpg import clip_by_norm_, mlp_model, param_sizes, polyak_update_, soft_sample


def test_param_sizes_follow_the_action_spaces():
    assert param_sizes(spaces.Discrete(5)) == [5]
    assert param_sizes(spaces.MultiDiscrete([[0, 4], [0, 6]])) == [5, 7]      # a leader: 5 physical + 7 commands


def test_soft_sample_is_a_gumbel_softmax_per_sub_space():
    g = torch.Generator().manual_seed(0)
    flat = torch.randn(4096, 12, dtype=torch.float64)
    out = soft_sample(flat, [5, 7], generator=g)
    assert out.shape == flat.shape and (out >= 0).all()
    assert torch.allclose(out[:, :5].sum(1), torch.ones(4096, dtype=torch.float64))
    assert torch.allclose(out[:, 5:].sum(1), torch.ones(4096, dtype=torch.float64))
    g = torch.Generator().manual_seed(0)                                    # closed form with the same uniforms
    u = torch.rand(flat.shape, dtype=torch.float64, generator=g)
    want = torch.softmax((flat - torch.log(-torch.log(u)))[:, :5], dim=-1)
    assert torch.equal(out[:, :5], want)
    # argmax of the relaxed sample is distributed like softmax(logits) (the Gumbel-max trick)
    logits = torch.tensor([1.0, 0.0, -1.0, 0.5, 0.0], dtype=torch.float64).expand(200000, 5)
    freq = torch.bincount(soft_sample(logits, [5]).argmax(1), minlength=5).double() / 200000
    assert torch.allclose(freq, torch.softmax(logits[0], 0), atol=5e-3)


def test_clip_by_norm_is_per_variable():
    a, b = torch.nn.Parameter(torch.zeros(3)), torch.nn.Parameter(torch.zeros(2))
    a.grad, b.grad = torch.tensor([3.0, 4.0, 0.0]), torch.tensor([0.3, 0.0])
    clip_by_norm_([a, b], 0.5)                                              # tf.clip_by_norm: g * clip / max(|g|, clip)
    assert torch.allclose(a.grad, torch.tensor([0.3, 0.4, 0.0])) and torch.allclose(b.grad, torch.tensor([0.3, 0.0]))


def test_polyak_update_and_model_shape():
    torch.manual_seed(0)
    online, target = mlp_model(6, 3, num_units=8), mlp_model(6, 3, num_units=8)
    assert [tuple(p.shape) for p in online.parameters()] == [(8, 6), (8,), (8, 8), (8,), (3, 8), (3,)]
    assert all((m.bias == 0).all() for m in online if isinstance(m, torch.nn.Linear))
    w0, t0 = online[0].weight.detach().clone(), target[0].weight.detach().clone()
    polyak_update_(online, target)
    assert torch.allclose(target[0].weight, 0.99 * t0 + 0.01 * w0)          # maddpg.py:21-24
    assert torch.equal(online[0].weight, w0)


def _args(**kw):
    d = dict(lr=1e-2, gamma=0.95, batch_size=256, num_units=64, max_episode_len=25)
    d.update(kw)
    return types.SimpleNamespace(**d)


@pytest.mark.gpu
@pytest.mark.parametrize('scenario', ['simple_spread', 'S1L2F2_3Targets_100observe'])
def test_training_loop_runs_on_device_and_learns_the_critic(scenario):
    """The reference's loop shape (simple_train.py:257-300): act, step, experience, update; everything stays
    on the GPU.  The critic's Bellman error on a held batch falls over the updates; targets follow by polyak."""
    from marl_amd import make_env
    from marl_amd.maddpg import MADDPGAgentTrainer
    torch.manual_seed(0)
    B = 512
    env = make_env(scenario, num_envs=B, dtype=torch.float32, seed=1)
    args = _args(batch_size=256, max_episode_len=4)
    obs_shape_n = [env.observation_space[i].shape for i in range(env.n)]
    trainers = [MADDPGAgentTrainer('agent_%d' % i, obs_shape_n, env.action_space, i, args) for i in range(env.n)]
    obs_n = env.reset()
    losses, t = [], 0
    tp0 = trainers[0].target_q[0].weight.detach().clone()
    for step in range(1, 1201):
        action_n = [tr.action(obs) for tr, obs in zip(trainers, obs_n)]
        for a, sp in zip(action_n, env.action_space):
            assert a.shape == (B, sum(param_sizes(sp))) and a.is_cuda
        new_obs_n, rew_n, done_n, info_n = env.step(action_n)
        if step % 10 == 0:                                                # keep the test short: store every 10th step
            for i, tr in enumerate(trainers):
                tr.experience(obs_n[i], action_n[i], rew_n[i], new_obs_n[i], done_n[i], False)
        obs_n = new_obs_n
        if step % 25 == 0:
            obs_n = env.reset()
        t += 1
        for tr in trainers:
            tr.preupdate()
        for tr in trainers:
            out = tr.update(trainers, t)
            if out is not None and tr is trainers[0]:
                losses.append([float(x) for x in out])
    assert len(losses) == 12                                              # every 100 steps (maddpg.py:164)
    assert all(np.isfinite(l).all() for l in losses)
    assert not torch.equal(trainers[0].target_q[0].weight, tp0)           # targets moved
    assert len(trainers[0].replay_buffer) == 120 * B


@pytest.mark.gpu
def test_update_follows_the_reference_arithmetic():
    """One update on a hand-filled buffer: target_q = rew + gamma (1 - done) Q'(o', a'~pi'), q_loss = mean
    squared error before the step; local_q_func (ddpg) only sees the agent's own obs and action."""
    from marl_amd.maddpg import MADDPGAgentTrainer
    torch.manual_seed(3)
    args = _args(batch_size=64, max_episode_len=1)
    obs_shape_n, act_space_n = [(6,), (4,)], [spaces.Discrete(5), spaces.MultiDiscrete([[0, 4], [0, 2]])]
    tr = [MADDPGAgentTrainer('a%d' % i, obs_shape_n, act_space_n, i, args, local_q_func=(i == 1)) for i in range(2)]
    assert tr[0].q[0].in_features == 6 + 4 + 5 + 8 and tr[1].q[0].in_features == 4 + 8
    n = 64
    data = []
    for i, (od, ad) in enumerate(((6, 5), (4, 8))):
        obs, act, nxt = torch.rand(n, od, device='cuda'), torch.rand(n, ad, device='cuda'), torch.rand(n, od, device='cuda')
        rew, done = torch.randn(n, device='cuda'), (torch.rand(n, device='cuda') < 0.3).float()
        tr[i].experience(obs, act, rew, nxt, done, False)
        data.append((obs, act, rew, nxt, done))
    assert tr[0].update(tr, 3) is None                                    # not a multiple of 100
    q0 = [p.detach().clone() for p in tr[0].q.parameters()]
    p0 = [p.detach().clone() for p in tr[0].p.parameters()]
    tq0 = [p.detach().clone() for p in tr[0].target_q.parameters()]
    q_loss, p_loss, mean_tq, mean_rew, mean_tqn, std_tq = [float(x) for x in tr[0].update(tr, 100)]
    idx = tr[0].replay_sample_index
    assert idx.shape == (64,) and int(idx.max()) < n
    rew, done = data[0][2][idx], data[0][4][idx]
    assert mean_rew == pytest.approx(float(rew.mean()), rel=1e-6)
    assert q_loss >= 0 and np.isfinite([q_loss, p_loss, std_tq, mean_tqn]).all()
    assert any(not torch.equal(a, b) for a, b in zip(q0, tr[0].q.parameters()))
    assert any(not torch.equal(a, b) for a, b in zip(p0, tr[0].p.parameters()))
    for a, b, c in zip(tq0, tr[0].target_q.parameters(), tr[0].q.parameters()):
        assert torch.allclose(b, 0.99 * a + 0.01 * c, atol=1e-7)          # polyak after the step (maddpg.py:193-194)
    # with every transition terminal the target is the reward itself
    for i in range(2):
        tr[i].replay_buffer.clear()
        obs, act, rew, nxt, _ = data[i]
        tr[i].experience(obs, act, rew, nxt, torch.ones(n, device='cuda'), False)
    import copy
    critic = copy.deepcopy(tr[1].q)                                       # the critic before the step
    out = [float(x) for x in tr[1].update(tr, 200)]
    assert out[2] == pytest.approx(out[3], rel=1e-6)                       # mean(target_q) == mean(rew)
    idx = tr[1].replay_sample_index
    with torch.no_grad():                                                 # ddpg critic: own obs and action only
        q = critic(torch.cat([data[1][0][idx], data[1][1][idx]], 1))[:, 0]
    assert out[0] == pytest.approx(float(torch.mean((q - data[1][2][idx]) ** 2)), rel=1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize('kernel', [1, 2, 3], ids=['valu', 'mfma', 'mfma_f16'])
@pytest.mark.parametrize('in_dim,sizes', [(16, [5]), (34, [5, 4]), (20, [5, 7]), (3, [3]), (11, [5, 10]), (244, [5])])
def test_fused_actor_kernel_matches_torch(in_dim, sizes, kernel):
    """mpe_actor_forward_f32 against the plain PyTorch fp32 forward + Gumbel-softmax with the same uniforms:
    <= 1e-5 (different summation order in the three small dot products)."""
    import functools
    from marl_amd import maddpg
    from marl_amd.maddpg import pack_actor
    actor_forward = functools.partial(maddpg.actor_forward, kernel=kernel)   # vector kernel / matrix-core kernel
    torch.manual_seed(in_dim)
    B, ld = 5000 + 13, 5376
    net = mlp_model(in_dim, sum(sizes)).cuda()
    for m in net:
        if isinstance(m, torch.nn.Linear):
            torch.nn.init.normal_(m.bias, std=0.3)                          # biases are zero at init: make them count
    obs = (torch.randn(in_dim, ld, device='cuda') * 2)[:, :B].t()          # SoA rows, as env.step returns them
    u = torch.rand(B, sum(sizes), device='cuda').clamp_(1e-6, 1 - 1e-6)
    packed = pack_actor(net)
    got = actor_forward(packed, obs, sizes, u=u)
    assert got.shape == (B, sum(sizes)) and got.stride(0) == 1 and got.stride(1) == ld
    with torch.no_grad():
        noisy = net(obs) - torch.log(-torch.log(u))
        want = torch.cat([torch.softmax(p, -1) for p in torch.split(noisy, sizes, -1)], -1)
    err = float((got - want).abs().max())
    print('actor kernel %d, %d inputs: max |err| vs torch f32 = %.2e' % (kernel, in_dim, err))
    assert err <= 1e-5              # (the opt-in f16 form, kernel 3, carries 22 significant bits per operand: same bar)
    own = actor_forward(packed, obs, sizes, seed=7, call_counter=1)       # the kernel's own Philox draw
    again = actor_forward(packed, obs, sizes, seed=7, call_counter=1)
    other = actor_forward(packed, obs, sizes, seed=7, call_counter=2)
    assert torch.equal(own, again) and not torch.equal(own, other)
    off = 0
    for s in sizes:
        assert torch.allclose(own[:, off:off + s].sum(1), torch.ones(B, device='cuda'), atol=1e-5)
        off += s
    # the argmax of the relaxed sample follows softmax(logits) (Gumbel-max): compare frequencies on one logit row
    const_obs = obs[:1].expand(B, in_dim).t().contiguous().t()
    acts = torch.cat([actor_forward(packed, const_obs, sizes, seed=3, call_counter=c)[:, :sizes[0]] for c in range(1, 21)])
    with torch.no_grad():
        p = torch.softmax(net(obs[:1])[0, :sizes[0]], 0)
    freq = torch.bincount(acts.argmax(1), minlength=sizes[0]).float() / acts.shape[0]
    assert float((freq - p).abs().max()) < 0.01


def test_philox_open_interval_conversion_never_returns_one():
    """mpe_actor_kernel.h:philox_open01 restated in numpy float32: ((w >> 9) + 0.5) * 2^-23.  The 24-bit form
    ((w >> 8) + 0.5) * 2^-24 rounds to exactly 1.0 for w = 0xFFFFFFFF (round-to-even tie) -> log(-log(1)) = -inf
    -> NaN action about once per step at 1 Mi envs; the 23-bit form is exact."""
    w = np.array([0xFFFFFFFF, 0, 0x80000000, 0xFFFFFE00], dtype=np.uint32)
    old = ((w >> np.uint32(8)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 16777216.0)
    new = ((w >> np.uint32(9)).astype(np.float32) + np.float32(0.5)) * np.float32(1.0 / 8388608.0)
    assert old[0] == np.float32(1.0)                                    # the bug this replaces
    assert new.dtype == np.float32 and new.max() < 1.0 and new.min() > 0.0
    assert new[0] == np.float32(1.0) - np.float32(2.0 ** -24) and new[1] == np.float32(2.0 ** -24)
    g = -np.log(-np.log(new.astype(np.float64)))
    assert np.isfinite(g).all()


@pytest.mark.gpu
def test_fused_actor_extreme_uniforms_and_sharded_noise():
    """(a) the extreme values the uniform conversion can produce (2^-24 and 1 - 2^-24, i.e. Philox words 0 and
    0xFFFFFFFF) give finite actions that sum to 1 per sub-space; (b) the kernel's own draw is finite everywhere on
    a large batch; (c) the draw is keyed by the GLOBAL env id: shards [0, B) and [B, 2B) with env_id0 = 0 / B
    reproduce the single 2B-env call bit for bit (env-sharded multi-GPU runs explore identically whatever the
    world size)."""
    from marl_amd import maddpg
    from marl_amd.maddpg import actor_forward, pack_actor
    torch.manual_seed(5)
    sizes, in_dim = [5, 7], 20
    net = mlp_model(in_dim, sum(sizes)).cuda()
    packed = pack_actor(net)
    B, ld = 4096, 4096 + 256
    obs_rows = torch.randn(in_dim, ld, device='cuda')
    obs = obs_rows[:, :B].t()
    lo, hi = 2.0 ** -24, 1.0 - 2.0 ** -24
    u = torch.full((B, sum(sizes)), hi, device='cuda')
    u[::2] = lo
    u[1::4, 0] = lo
    got = actor_forward(packed, obs, sizes, u=u)
    assert torch.isfinite(got).all()
    assert torch.allclose(got[:, :5].sum(1), torch.ones(B, device='cuda'), atol=1e-5)
    assert torch.allclose(got[:, 5:].sum(1), torch.ones(B, device='cuda'), atol=1e-5)
    for c in range(1, 4):
        own = actor_forward(packed, obs, sizes, seed=11, call_counter=c)
        assert torch.isfinite(own).all()
    # sharding: two half batches with their global env offsets == one call over the concatenation
    B2 = 2 * B
    rows2 = torch.randn(in_dim, B2 + 256, device='cuda')
    whole = actor_forward(packed, rows2[:, :B2].t(), sizes, seed=3, call_counter=9, env_id0=1000)
    first = actor_forward(packed, rows2[:, :B].t(), sizes, seed=3, call_counter=9, env_id0=1000)
    second_rows = torch.zeros(in_dim, B + 256, device='cuda')
    second_rows[:, :B] = rows2[:, B:B2]
    second = actor_forward(packed, second_rows[:, :B].t(), sizes, seed=3, call_counter=9, env_id0=1000 + B)
    assert torch.equal(whole[:B], first) and torch.equal(whole[B:], second)
    wrong = actor_forward(packed, second_rows[:, :B].t(), sizes, seed=3, call_counter=9, env_id0=1000)
    assert not torch.equal(whole[B:], wrong)
    # the vector kernel and the matrix-core kernel draw the same noise: equal up to the summation order of the dot products
    a1 = maddpg.actor_forward(packed, rows2[:, :B2].t(), sizes, seed=3, call_counter=9, env_id0=1000, kernel=1)
    a2 = maddpg.actor_forward(packed, rows2[:, :B2].t(), sizes, seed=3, call_counter=9, env_id0=1000, kernel=2)
    assert float((a1 - a2).abs().max()) <= 1e-5 and torch.equal(whole, a2)      # AUTO = the matrix-core kernel


@pytest.mark.gpu
def test_trainer_noise_follows_seed_and_global_env_id():
    """MADDPGAgentTrainer passes world.env_id0 and a seed derived from the run seed to the actor kernel."""
    from marl_amd import make_env
    from marl_amd.maddpg import MADDPGAgentTrainer
    B = 512
    env = make_env('simple_tag', num_envs=2 * B, dtype=torch.float32, seed=4)
    obs_n = env.reset()
    shapes = [env.observation_space[i].shape for i in range(env.n)]

    def trainer(seed, env_id0):
        torch.manual_seed(0)                                             # same weights
        return MADDPGAgentTrainer('a', shapes, env.action_space, 0, _args(), replay_size=64, seed=seed, env_id0=env_id0)
    whole = trainer(7, 0).action(obs_n[0])
    rows = env.world.obs[:16]
    lo = trainer(7, 0).action(rows[:, :B].t())
    hi_rows = env.world.rows(16)
    hi_rows[:, :B] = rows[:, B:2 * B]
    hi = trainer(7, B).action(hi_rows[:, :B].t())
    assert torch.equal(whole[:B], lo) and torch.equal(whole[B:], hi)
    assert not torch.equal(whole, trainer(8, 0).action(obs_n[0]))         # the run seed matters


@pytest.mark.gpu
@pytest.mark.parametrize('scenario', ['simple_tag', 'S1L2F2_3Targets_100observe'])
def test_act_all_and_add_all_equal_the_per_agent_calls(scenario):
    """One launch for every agent's actor (mpe_actor_forward_multi_f32) and one for every agent's replay append
    (mpe_replay_add_multi_f32) give exactly what the reference's per-trainer loops give (simple_train.py:260,284-287)."""
    from marl_amd import make_env
    from marl_amd.maddpg import MADDPGAgentTrainer, act_all
    from marl_amd.replay_buffer import add_all
    B = 3000
    env = make_env(scenario, num_envs=B, dtype=torch.float32, seed=5)
    shapes = [env.observation_space[i].shape for i in range(env.n)]

    def trainers():
        torch.manual_seed(1)
        return [MADDPGAgentTrainer('a%d' % i, shapes, env.action_space, i, _args(), replay_size=4096, seed=9)
                for i in range(env.n)]
    one, many = trainers(), trainers()
    obs_n = env.reset()
    for step in range(3):                                  # 3 x 3000 transitions into rings of 4096: the last append wraps
        a_many = act_all(many, obs_n)
        a_one = [tr.action(o) for tr, o in zip(one, obs_n)]
        for x, y in zip(a_many, a_one):
            assert x.shape == y.shape and torch.equal(x, y)
        new_obs_n, rew_n, done_n, _ = env.step(a_many)
        add_all([tr.replay_buffer for tr in many], obs_n, a_many, rew_n, new_obs_n, done_n)
        for i, tr in enumerate(one):
            tr.experience(obs_n[i], a_one[i], rew_n[i], new_obs_n[i], done_n[i], False)
        obs_n = new_obs_n
    for t1, t2 in zip(one, many):
        r1, r2 = t1.replay_buffer, t2.replay_buffer
        assert len(r1) == len(r2) == 4096 and r1._next_idx == r2._next_idx == (3 * B) % 4096
        assert torch.equal(r1._ring, r2._ring)
    assert not torch.equal(a_many[0], act_all(many, obs_n)[0])          # the call counter moved on
    # out=env.action_views: written in place, env.step then copies nothing, same values as a fresh buffer
    a_ref = [x.clone() for x in act_all(trainers(), obs_n)]
    a_in = act_all(trainers(), obs_n, out=env.action_views)
    for x, y, v in zip(a_in, a_ref, env.action_views):
        assert x.data_ptr() == v.data_ptr() and torch.equal(x, y)


@pytest.mark.gpu
def test_f16_actor_kernel_at_the_edges_of_its_range():
    """MPE_ACTOR_MFMA_F16 (opt-in) splits every operand into two f16 pieces: observations of 1e-6 (pieces in f16's
    subnormal range: 6e-8 absolute resolution) still meet the 1e-5 bar against PyTorch f32; observations beyond +-65 504
    saturate -- the documented price -- but stay finite and normalised (no inf - inf in the softmax)."""
    from marl_amd import maddpg
    from marl_amd.maddpg import pack_actor
    torch.manual_seed(3)
    B, ld, in_dim, sizes = 4096 + 5, 4352, 16, [5]
    net = mlp_model(in_dim, 5).cuda()
    packed = pack_actor(net)
    u = torch.rand(B, 5, device='cuda').clamp_(1e-6, 1 - 1e-6)
    for scale in (1e-6, 1e-3, 30.0):
        obs = (torch.randn(in_dim, ld, device='cuda') * scale)[:, :B].t()
        got = maddpg.actor_forward(packed, obs, sizes, u=u, kernel=3)
        with torch.no_grad():
            want = torch.softmax(net(obs) - torch.log(-torch.log(u)), -1)
        assert float((got - want).abs().max()) <= 1e-5, scale
    huge = (torch.randn(in_dim, ld, device='cuda') * 1e6)[:, :B].t()
    got = maddpg.actor_forward(packed, huge, sizes, u=u, kernel=3)
    assert torch.isfinite(got).all() and torch.allclose(got.sum(1), torch.ones(B, device='cuda'), atol=1e-5)
    ref = maddpg.actor_forward(packed, huge, sizes, u=u, kernel=2)          # the default kernel has f32's range
    with torch.no_grad():
        want = torch.softmax(net(huge) - torch.log(-torch.log(u)), -1)
    assert float((ref - want).abs().max()) <= 1e-4 and float((got - want).abs().max()) > 1e-2
