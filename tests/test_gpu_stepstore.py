"""marl_amd.replay_buffer.StepStore (mpe_replay_gather_steps_*): the replay rings filled by the step's own launches --
no append.  PARITY UNPINNED like the rest of SURVEY 8f row f4 (the reference's replay_buffer.py could not be imported,
DESIGN.md section 4): the store is held, bit for bit, to the full-record ReplayBuffer fed the same steps, which
tests/test_gpu_replay.py holds to the CPU restatement of the reference's buffer."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _run(scenario, B, slots, steps, ep_len, dtype, sample_every=4):
    from marl_amd import make_env
    from marl_amd.replay_buffer import ReplayBuffer, StepStore
    env = make_env(scenario, num_envs=B, dtype=dtype, seed=3)
    obs_n = env.reset()
    store = StepStore(env, slots)
    views = store.views()
    full = [ReplayBuffer(B * (steps + 1), dtype=dtype) for _ in range(env.n)]
    g = torch.Generator(device='cuda').manual_seed(B)
    obs_n = env.reset()
    done_steps = 0
    for t in range(steps):
        av = store.action_views()
        if t % 3 == 2:                       # a caller with its own action tensors: copied into the slot
            action_n = [torch.rand(v.shape, generator=g, device='cuda', dtype=dtype) for v in av]
            new_obs_n, rew_n, done_n, _ = store.step(action_n)
        else:                                # an actor that writes the slot's rows itself
            for v in av:
                v.copy_(torch.rand(v.shape, generator=g, device='cuda', dtype=dtype))
            action_n = av
            new_obs_n, rew_n, done_n, _ = store.step()
        for i in range(env.n):
            full[i].add(obs_n[i], action_n[i], rew_n[i], new_obs_n[i], done_n[i])
        done_steps += 1
        obs_n = new_obs_n
        if (t + 1) % ep_len == 0:
            obs_n = env.reset()
        if t % sample_every == 0 or t == steps - 1:
            L = len(views[0]) // B
            assert 1 <= L <= slots - 1 and all(len(v) == L * B for v in views)
            idx = torch.randint(0, L * B, (257,), generator=g, device='cuda')
            ref_idx = (done_steps - L) * B + idx          # the k-th oldest stored step is step number done_steps - L + k
            for i in range(env.n):
                for x, y in zip(views[i].sample_index(idx), full[i].sample_index(ref_idx)):
                    assert torch.equal(x, y), (scenario, t, i)
    return env, store, views, full, done_steps


@pytest.mark.parametrize('dtype', [torch.float32, torch.float64], ids=['f32', 'f64'])
@pytest.mark.parametrize('scenario,B,slots,ep_len', [('simple_tag', 300, 9, 5), ('simple_tag', 1000, 4, 25),
                                                      ('simple_world_comm', 257, 13, 4), ('simple_speaker_listener', 64, 3, 2),
                                                      ('S1L2F2_10Targets_wide', 500, 8, 7)])
def test_step_store_samples_equal_a_full_ring(scenario, B, slots, ep_len, dtype):
    """Several wraps of the slot ring, episode ends (the slot that only holds an episode's last obs_tp1 is skipped),
    actors that write the slot's action rows and callers that bring their own tensors, uniform and ragged action rows."""
    env, store, views, full, n = _run(scenario, B, slots, 4 * slots + 3, ep_len, dtype)
    L = len(views[0]) // B
    for i in range(env.n):                                   # everything that is stored, in order
        got = views[i].collect()
        want = full[i].sample_index(torch.arange((n - L) * B, n * B, device='cuda'))
        for x, y in zip(got, want):
            assert torch.equal(x, y)
        lat = views[i].make_latest_index(50)
        assert int(lat.min()) == L * B - 50 and int(lat.max()) == L * B - 1


def test_step_store_refuses_what_it_cannot_do():
    from marl_amd import _hip, make_env
    from marl_amd.replay_buffer import StepStore
    env = make_env('simple_tag', num_envs=64, dtype=torch.float32, seed=0)
    env.reset()
    with pytest.raises(ValueError):
        StepStore(env, 2)
    store = StepStore(env, 4)
    v = store.views()[0]
    with pytest.raises(RuntimeError):
        v.add(None, None, None, None, None)
    with pytest.raises(IndexError):
        v.sample_index([0])                                  # nothing stored yet
    for a in store.action_views():
        a.fill_(0.2)
    store.step()
    assert len(v) == 64
    with pytest.raises(IndexError):
        v.sample_index([64])
    with pytest.raises(IndexError):
        v.make_latest_index(65)
    env2 = make_env('simple_tag', num_envs=64, dtype=torch.float32, seed=0)
    env2.discrete_action_input = True
    with pytest.raises(RuntimeError):
        StepStore(env2, 4)
    # the C ABI's own checks
    f = _hip.lib.mpe_replay_gather_steps_f32
    import ctypes as C
    idx = torch.zeros(4, dtype=torch.int64, device='cuda')
    out = torch.zeros(4, 2 * 16 + 5 + 2, device='cuda')

    def desc(**kw):
        d = dict(obs=store.obs.data_ptr(), act=store.act.data_ptr(), rew=store.rew.data_ptr(), done=None,
                 live=store._live_dev.data_ptr(), obs_slot_stride=store.obs.stride(0), act_slot_stride=store.act.stride(0),
                 rew_slot_stride=store.rew.stride(0), ld=env.world.ld, n=64, slots=4, n_live=1, obs_row0=0, act_row0=0, rew_row=0)
        d.update(kw)
        return C.byref(_hip.StepStoreDesc(**d))
    len(v)                                                   # (uploads the table and the count)
    assert f(desc(), 16, 5, idx.data_ptr(), 4, out.data_ptr(), None) == 0
    assert f(desc(obs=None), 16, 5, idx.data_ptr(), 4, out.data_ptr(), None) == -1
    assert f(desc(live=None), 16, 5, idx.data_ptr(), 4, out.data_ptr(), None) == -1
    assert f(desc(n_live=4), 16, 5, idx.data_ptr(), 4, out.data_ptr(), None) == -6      # a full ring has no successor slot
    assert f(desc(obs_slot_stride=15 * env.world.ld), 16, 5, idx.data_ptr(), 4, out.data_ptr(), None) == -6
    assert f(desc(ld=32), 16, 5, idx.data_ptr(), 4, out.data_ptr(), None) == -6
    assert f(desc(), 16, 5, idx.data_ptr(), 4, out.data_ptr() + 2, None) == -4
    torch.cuda.synchronize()


def _train_setup(B, seed=0, u_noise=None):
    import types
    from marl_amd import make_env
    from marl_amd.maddpg import MADDPGAgentTrainer
    env = make_env('simple_tag', num_envs=B, dtype=torch.float32, seed=seed)
    if u_noise:                                              # drawn inside the step kernels, keyed by the step counter
        for a in env.world.agents:
            a.u_noise = u_noise
        env.world.invalidate_descriptor()
    args = types.SimpleNamespace(lr=1e-2, gamma=0.95, batch_size=256, num_units=64, max_episode_len=25)
    obs_shape_n = [env.observation_space[i].shape for i in range(env.n)]
    torch.manual_seed(123)                                   # identical initial networks in every copy
    trainers = [MADDPGAgentTrainer('agent_%d' % i, obs_shape_n, env.action_space, i, args, replay_size=B * 64,
                                   seed=seed, env_id0=env.world.env_id0) for i in range(env.n)]
    return env, trainers


def _train_run(B, K, mode, episodes=4, u_noise=None):
    """mode 'ring': the classic loop (act -> step -> add_all into full-record rings); 'store': act -> StepStore.step;
    'graph': the same as one HIP-graph replay per episode.  A weight update after the second episode."""
    from marl_amd.graph import CapturedStoreSteps
    from marl_amd.maddpg import act_all
    from marl_amd.replay_buffer import StepStore, add_all
    env, trainers = _train_setup(B, u_noise=u_noise)
    obs_n = env.reset()
    store = cap = None
    if mode != 'ring':
        store = StepStore(env, 3 * (K + 1))                  # three windows: the fourth episode wraps
        for tr, v in zip(trainers, store.views()):
            tr.replay_buffer = v
    obs_n = env.reset()                                      # (every mode: the reset stream is keyed by the episode count)
    act_all(trainers, obs_n)                                 # packs the actor weights (the action itself is dropped) ...
    for tr in trainers:
        tr._act_calls -= 1                                   # ... and does not count
    if mode == 'graph':
        cap = CapturedStoreSteps(env, trainers, store, K)
        assert store.align(K + 1) == K + 1                   # the reset above left the store at slot 1
    for ep in range(episodes):
        if mode == 'graph':
            obs_n = cap.replay()[0]
        else:
            for k in range(K):
                if mode == 'store':
                    act_all(trainers, obs_n, out=store.action_views())
                    obs_n = store.step()[0]
                else:
                    action_n = act_all(trainers, obs_n, out=env.action_views)
                    new_obs_n, rew_n, done_n, _ = env.step(action_n)
                    add_all([tr.replay_buffer for tr in trainers], obs_n, action_n, rew_n, new_obs_n, done_n)
                    obs_n = new_obs_n
        obs_n = env.reset()
        if ep == 1:
            torch.manual_seed(7)
            idx = torch.randint(0, 2 * K * B, (256,), device='cuda')
            for tr in trainers:
                tr.preupdate()
            for tr in trainers:                              # the same batch in every mode: the next episodes act alike
                tr.update_from_index(trainers, idx)
    torch.cuda.synchronize()
    return env, trainers, store


@pytest.mark.parametrize('K,u_noise', [(4, None), (7, None), (5, 0.3)])
def test_store_loops_hold_what_the_ring_loop_holds(K, u_noise):
    """The reference's loop shape with (a) full-record rings and one append launch per step, (b) a StepStore, (c) the
    StepStore loop as one HIP-graph replay per episode (marl_amd.graph.CapturedStoreSteps; three windows of K + 1 slots,
    the fourth episode reuses the first window's graph): the same actions, states and stored transitions, bit for bit,
    with a learner update in between (the packed actor weights are refilled in place); also with u_noise drawn in the
    step kernels (the recorded launches take the moving part of the noise step counter from device memory)."""
    B, episodes = 512, 4
    ring = _train_run(B, K, 'ring', episodes, u_noise)
    for mode in ('store', 'graph'):
        env, trainers, store = _train_run(B, K, mode, episodes, u_noise)
        assert torch.equal(env.world.pos, ring[0].world.pos) and torch.equal(env.world.vel, ring[0].world.vel)
        assert [tr._act_calls for tr in trainers] == [tr._act_calls for tr in ring[1]]
        L = len(trainers[0].replay_buffer) // B
        assert L == 3 * K - 1                # three windows; the reset after the last episode took the oldest step's slot
        for tr, ref in zip(trainers, ring[1]):
            got = tr.replay_buffer.collect()
            want = ref.replay_buffer.sample_index(torch.arange((episodes * K - L) * B, episodes * K * B, device='cuda'))
            for x, y in zip(got, want):
                assert torch.equal(x, y), mode


def test_captured_store_steps_refuse_a_misaligned_store():
    from marl_amd.graph import CapturedStoreSteps
    from marl_amd.maddpg import act_all
    from marl_amd.replay_buffer import StepStore
    env, trainers = _train_setup(256)
    obs_n = env.reset()
    store = StepStore(env, 10)
    with pytest.raises(RuntimeError):
        CapturedStoreSteps(env, trainers, store, 4)          # no packed weights yet
    act_all(trainers, obs_n, out=store.action_views())
    obs_n = store.step()[0]
    with pytest.raises(ValueError):
        CapturedStoreSteps(env, trainers, store, 5)          # 10 slots, windows of 6
    cap = CapturedStoreSteps(env, trainers, store, 4)
    with pytest.raises(RuntimeError):
        cap.replay()                                         # the store is at slot 1
    assert store.align(5) == 5 and len(store.views()[0]) == 256    # the eager step's transition (slot 0) is still there
    cap.replay()
    assert len(store.views()[0]) == 5 * 256


def test_a_recorded_gather_follows_the_store_as_it_grows():
    """marl_amd.graph.CapturedUpdates records the learners' sample_index calls in a HIP graph, which bakes the kernel
    arguments: the number of stored steps is therefore read from device memory (behind the slot table), and a replay
    after the store has grown -- and wrapped -- gathers what an eager call gathers."""
    from marl_amd import make_env
    from marl_amd.replay_buffer import StepStore
    B = 384
    env = make_env('simple_tag', num_envs=B, dtype=torch.float32, seed=1)
    env.reset()
    store = StepStore(env, 6)
    view = store.views()[1]
    g = torch.Generator(device='cuda').manual_seed(5)

    def steps(k):
        for _ in range(k):
            for v in store.action_views():
                v.copy_(torch.rand(v.shape, generator=g, device='cuda'))
            store.step()
    steps(2)
    idx = torch.randint(0, len(view), (300,), generator=g, device='cuda')
    view.check_index = False                                 # (the range check reads the indices back on the host)
    graph = torch.cuda.CUDAGraph()
    torch.cuda.synchronize()
    with torch.cuda.graph(graph, capture_error_mode='thread_local'):
        rec = view.sample_index(idx)
    for k in (0, 2, 1, 4):                                   # 2, 4, 5 stored steps, then a wrapped ring of 5
        steps(k)
        n = len(view)                                        # (uploads the slot table and the count, like make_index)
        assert n == min(2 + sum((0, 2, 1, 4)[:(0, 2, 1, 4).index(k) + 1]), 5) * B
        idx.copy_(torch.randint(0, n, (300,), generator=g, device='cuda'))
        idx[0] = n - 1
        graph.replay()
        for x, y in zip(rec, view.sample_index(idx.clone())):
            assert torch.equal(x, y)
