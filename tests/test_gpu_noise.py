"""u_noise / c_noise normals drawn INSIDE the step kernels (MPE_FLAG_DEVICE_NOISE, include/mpe_hip.h; reference:
`np.random.randn(*agent.action.u.shape) * agent.u_noise`, multiagent/core.py:147-148, and `... * agent.c_noise`, :229).

The reference draws from numpy's global stream, which device code cannot follow (the reference-generated fixtures G4 /
G14 keep pinning the arithmetic with the draws SUPPLIED: tests/test_gpu_parity.py, tests/test_gpu_comm.py).  What is
pinned here is this build's own draw: oracle.device_normals restates it operation by operation, so that
  * f64: a step that draws its own normals equals, BIT FOR BIT, the same step fed with the oracle's normals, on the
    wide, split and tile kernels, and follows the oracle's trajectory within 1e-9;
  * f32 (hardware log / sin / cos on the same Philox words): within 1e-4 of the same;
  * a draw depends on (seed, global env id, step, agent, component) only: shards of a batch, the device step word of
    HIP-graph replays and the three kernel forms all see the same numbers."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest
import torch

from oracle import mpe_oracle as O

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

if torch.cuda.is_available():
    from marl_amd import _hip
    from tests.gpu_util import HipSim, dev, lib, set_call_flags

SEED = 0x1234567811223344


@pytest.fixture(params=[2, 4, 1], ids=['wide_kernel', 'split_kernel', 'tile_kernel'])
def kernel_form(request):
    set_call_flags(request.param)
    yield request.param
    set_call_flags(0)


def _world(spec, B, rng, spread=0.4):
    """A dense random state (contacts and the speed clamp fire) and soft actions, SoA."""
    pos = rng.uniform(-spread, spread, size=(spec.n_entities, 2, B))
    vel = rng.uniform(-0.5, 0.5, size=(spec.n_agents, 2, B))
    return pos, vel


def _actions(spec, B, rng):
    a = rng.rand(sum(spec.act_dims), B)
    return a.reshape(spec.n_agents, 5, B) if all(d == 5 for d in spec.act_dims) else a


@pytest.mark.parametrize('mk', [lambda: O.simple_tag_spec(u_noise=0.3), lambda: O.simple_spread_spec(u_noise=0.7),
                                lambda: O.simple_tag_spec(5, 3, 4, u_noise=0.2)], ids=['tag', 'spread', 'tag_5_3_4'])
def test_f64_device_u_noise_equals_supplied_oracle_normals_bit_for_bit(mk, kernel_form):
    spec = mk()
    B, env_id0 = 3001, (1 << 33) + 12345           # ragged batch; global ids beyond 32 bits
    rng = np.random.RandomState(3)
    pos0, vel0 = _world(spec, B, rng)
    ids = np.arange(env_id0, env_id0 + B, dtype=np.uint64)
    own, fed = HipSim(spec, B, torch.float64), HipSim(spec, B, torch.float64)
    bo = O.BatchedOracle(spec)
    for s in (own, fed):
        s.set_state(pos0, vel0)
    p, v = pos0, vel0
    seen = []
    for t in range(6):
        step = 0xFFFFFFFE + t if t < 3 else 7 * t          # incl. the wrap of the 32-bit step counter
        step &= 0xFFFFFFFF
        act_np = _actions(spec, B, rng)
        act = dev(act_np, torch.float64)
        nz = O.device_u_noise(spec, SEED, step, ids)
        seen.append(nz)
        own.env_step(act, device_noise=(SEED, step, env_id0))
        fed.env_step(act, noise=dev(nz, torch.float64))
        for a, b in zip(own.host(), fed.host()):
            assert np.array_equal(a, b, equal_nan=True), (t, np.abs(a - b).max())
        p, v, obs, rew, info = bo.env_step(p, v, act_np, nz)
        gp, gv, gobs, grew, ginfo = own.host()
        assert np.abs(gp - p).max() <= 1e-9 and np.abs(gv - v).max() <= 1e-9
        assert np.abs(gobs - np.concatenate(obs, 0)).max() <= 1e-9
    z = np.concatenate([n.ravel() for n in seen])
    assert abs(z.mean()) < 0.02 and abs(z.std() - 1) < 0.02              # they are normals, and they differ per step
    assert not np.array_equal(seen[0], seen[1])


def test_f32_device_u_noise_within_tolerance_of_the_f64_draw(kernel_form):
    spec = O.simple_tag_spec(u_noise=0.3)
    B = 20011
    rng = np.random.RandomState(4)
    pos0, vel0 = _world(spec, B, rng, spread=1.0)
    ids = np.arange(B, dtype=np.uint64)
    act_np = _actions(spec, B, rng)
    own, fed = HipSim(spec, B, torch.float32), HipSim(spec, B, torch.float32)
    for s in (own, fed):
        s.set_state(pos0, vel0)
    nz = O.device_u_noise(spec, SEED, 9, ids)
    own.env_step(dev(act_np, torch.float32), device_noise=(SEED, 9, 0))
    fed.env_step(dev(act_np, torch.float32), noise=dev(nz, torch.float32))
    for a, b in zip(own.host()[:3], fed.host()[:3]):
        np.testing.assert_allclose(a, b, rtol=1e-4, atol=1e-5)
    # the normals themselves, recovered from a world where nothing else acts: vel' = (u + n * u_noise) * dt (no
    # contacts far apart, zero actions, zero velocity, max_speed not reached)
    far = np.zeros((spec.n_entities, 2, B))
    far[:, 0] = 10.0 * np.arange(spec.n_entities)[:, None]
    own.set_state(far, np.zeros((spec.n_agents, 2, B)))
    own.env_step(torch.zeros(spec.n_agents, 5, B, dtype=torch.float32, device='cuda'), device_noise=(SEED, 9, 0))
    got = own.host()[1] / (0.3 * 0.1)
    ok = np.abs(nz).max(1, keepdims=True) < 3.0          # below the speed clamp: |vel| = 0.03 |n| < 1
    assert np.abs(np.where(ok, got - nz, 0.0)).max() < 2e-5


def test_device_noise_is_independent_of_sharding_batch_and_step_word(kernel_form):
    """(seed, global env id, step) decide the draw: two shards with env_id0 offsets reproduce the whole batch, and the
    step may arrive as host word + device word (what a HIP-graph replay moves)."""
    spec = O.simple_tag_spec(u_noise=0.3)
    B, cut = 1500, 700
    rng = np.random.RandomState(5)
    pos0, vel0 = _world(spec, B, rng)
    act_np = _actions(spec, B, rng)
    whole = HipSim(spec, B, torch.float64)
    whole.set_state(pos0, vel0)
    whole.env_step(dev(act_np, torch.float64), device_noise=(SEED, 40, 1000))
    want = whole.host()
    word = torch.tensor([33], dtype=torch.int32, device='cuda')
    for lo, hi in ((0, cut), (cut, B)):
        part = HipSim(spec, hi - lo, torch.float64)
        part.set_state(pos0[..., lo:hi], vel0[..., lo:hi])
        part.env_step(dev(act_np[..., lo:hi], torch.float64), device_noise=(SEED, 7, 1000 + lo, word))
        for a, b in zip(part.host(), want):
            assert np.array_equal(a, b[..., lo:hi])
    # another seed / step / without the flag: different draws / no noise at all
    other = HipSim(spec, B, torch.float64)
    other.set_state(pos0, vel0)
    other.env_step(dev(act_np, torch.float64), device_noise=(SEED + 1, 40, 1000))
    assert not np.array_equal(other.host()[1], want[1])
    quiet, none = HipSim(spec, B, torch.float64), HipSim(O.simple_tag_spec(), B, torch.float64)
    for s in (quiet, none):
        s.set_state(pos0, vel0)
        s.env_step(dev(act_np, torch.float64))
    assert np.array_equal(quiet.host()[1], none.host()[1])               # flag off, no tensor: no noise (as before)
    # a supplied tensor overrides the in-kernel draw
    nz = rng.randn(spec.n_agents, 2, B)
    a, b = HipSim(spec, B, torch.float64), HipSim(spec, B, torch.float64)
    for s in (a, b):
        s.set_state(pos0, vel0)
    a.env_step(dev(act_np, torch.float64), noise=dev(nz, torch.float64), device_noise=(SEED, 40, 1000))
    b.env_step(dev(act_np, torch.float64), noise=dev(nz, torch.float64))
    assert np.array_equal(a.host()[1], b.host()[1])


def test_world_step_and_env_step_draw_the_same_normals(kernel_form):
    spec = O.simple_spread_spec(u_noise=0.5)
    B = 777
    rng = np.random.RandomState(6)
    pos0, vel0 = _world(spec, B, rng)
    act = dev(_actions(spec, B, rng), torch.float64)
    one, two = HipSim(spec, B, torch.float64), HipSim(spec, B, torch.float64)
    for s in (one, two):
        s.set_state(pos0, vel0)
    one.env_step(act, device_noise=(SEED, 3, 0))
    two.world_step(act, device_noise=(SEED, 3, 0))
    two.observe()
    for a, b in zip(one.host(), two.host()):
        assert np.array_equal(a, b)


@pytest.mark.parametrize('mk,cn', [(O.simple_reference_spec, 0.1), (O.simple_world_comm_spec, 0.25)],
                         ids=['simple_reference', 'simple_world_comm'])
@pytest.mark.parametrize('dtype,tol', [(torch.float64, 0.0), (torch.float32, 1e-4)], ids=['f64', 'f32'])
def test_device_c_noise_equals_supplied_oracle_normals(mk, cn, dtype, tol):
    """agent.state.c = action.c + c_noise * randn (core.py:229) with the draw made in the kernel: f64 bit-identical to the
    same step fed with the oracle's normals (state.c, the observations that carry it, everything else), f32 within
    1e-4; an odd dim_c (simple_world_comm has 4, simple_reference 10; the pair stream covers both parities below)."""
    import dataclasses
    base = mk()
    spec = dataclasses.replace(base, c_noise=[None if s else cn for s in base.silent])
    B, env_id0 = 1234, 99
    rng = np.random.RandomState(8)
    pos0, vel0 = _world(spec, B, rng)
    ids = np.arange(env_id0, env_id0 + B, dtype=np.uint64)
    own, fed = HipSim(spec, B, dtype), HipSim(spec, B, dtype)
    goal = rng.randint(0, spec.n_landmarks, size=(max(spec.goal_rows, 1), B))
    for s in (own, fed):
        s.set_state(pos0, vel0, goal)
    for t in range(3):
        act = dev(_actions(spec, B, rng), dtype)
        cnz = O.device_c_noise(spec, SEED, 100 + t, ids)
        own.env_step(act, comm=own.comm, device_noise=(SEED, 100 + t, env_id0))
        fed.env_step(act, comm=fed.comm, cnoise=dev(cnz, dtype))
        pairs = list(zip(own.host(), fed.host())) + [(own.comm[..., :B].double().cpu().numpy(),
                                                      fed.comm[..., :B].double().cpu().numpy())]
        for a, b in pairs:
            if tol == 0.0:
                assert np.array_equal(a, b, equal_nan=True), t
            else:
                np.testing.assert_allclose(a, b, rtol=tol, atol=tol)
        speakers = [i for i in range(spec.n_agents) if not spec.silent[i]]
        c = own.comm[..., :B].double().cpu().numpy()
        assert np.abs(c[speakers]).max() > 0 and all(np.all(c[i] == 0) for i in range(spec.n_agents) if spec.silent[i])


def test_odd_dim_c_uses_the_first_normal_of_the_last_pair():
    import dataclasses
    base = O.speaker_listener_spec()                    # dim_c = 3: components 0, 1 from pair 0, component 2 = z0 of pair 1
    spec = dataclasses.replace(base, c_noise=[0.5, None])
    B = 300
    rng = np.random.RandomState(9)
    pos0, vel0 = _world(spec, B, rng)
    ids = np.arange(B, dtype=np.uint64)
    sim = HipSim(spec, B, torch.float64)
    sim.set_state(pos0, vel0, rng.randint(0, 3, size=(1, B)))
    act_np = _actions(spec, B, rng)
    sim.env_step(dev(act_np, torch.float64), comm=sim.comm, device_noise=(SEED, 1, 0))
    cnz = O.device_c_noise(spec, SEED, 1, ids)
    want = act_np[0:3] + cnz[0] * 0.5
    assert np.array_equal(sim.comm[0, :, :B].cpu().numpy(), want)
    z0, _ = O.device_normals(SEED, 1, ids, O.NOISE_WORD_C | 1)
    assert np.array_equal(cnz[0, 2], z0)


# ---- through the public API ---------------------------------------------------------------------------------------
def _noisy_env(B, dtype, u_noise=0.3, env_id0=0):
    from marl_amd import make_env
    env = make_env('simple_tag', benchmark=True, num_envs=B, dtype=dtype, seed=5)
    for a in env.world.agents:
        a.u_noise = u_noise
    env.world.invalidate_descriptor()
    env.world.env_id0 = env_id0
    return env


def test_env_api_draws_in_kernel_by_default_and_matches_the_oracle():
    """agent.u_noise set on a world from make_env: no torch generator, no noise buffer; the trajectory is the oracle's
    with oracle.device_u_noise(seed, step counter, global env ids)."""
    B = 4096
    env = _noisy_env(B, torch.float64, env_id0=10_000)
    env.reset()
    w = env.world
    spec = O.simple_tag_spec(u_noise=0.3)
    bo = O.BatchedOracle(spec)
    p, v = w.pos.cpu().numpy(), w.vel.cpu().numpy()
    ids = np.arange(10_000, 10_000 + B, dtype=np.uint64)
    rng = np.random.RandomState(1)
    state = torch.cuda.get_rng_state()
    for t in range(5):
        a = rng.rand(4, 5, B)
        assert w.noise_step == t
        obs_n, rew_n, _, _ = env.step(torch.as_tensor(a).cuda().permute(2, 0, 1).contiguous())
        p, v, obs, rew, info = bo.env_step(p, v, a, O.device_u_noise(spec, w.seed, t, ids))
        assert np.abs(w.pos.cpu().numpy() - p).max() <= 1e-9
        assert np.abs(torch.cat(obs_n, 1).cpu().numpy().T - np.concatenate(obs, 0)).max() <= 1e-9
    assert w._noise_buf is None and torch.equal(torch.cuda.get_rng_state(), state)       # torch's generator untouched
    # a supplied tensor still overrides (parity mode), and host_noise = True is the round-2 path
    w.noise = torch.zeros(4, 2, B, dtype=torch.float64, device='cuda')
    quiet = _noisy_env(B, torch.float64, u_noise=None)
    quiet.reset()
    quiet.world.pos.copy_(w.pos)
    quiet.world.vel.copy_(w.vel)
    quiet.world.mark_dirty()
    a = torch.rand(B, 4, 5, dtype=torch.float64, device='cuda')
    env.step(a)
    quiet.step(a)
    assert torch.equal(env.world.pos, quiet.world.pos)


@pytest.mark.parametrize('dtype', [torch.float64, torch.float32], ids=['f64', 'f32'])
def test_graph_replays_of_a_noisy_world_equal_eager_stepping(dtype):
    """CapturedSteps bakes noise_step into the recorded launches; the device word moves it from replay to replay, so
    replays, eager steps and replays of a second graph in any order continue ONE stream of draws."""
    from marl_amd.graph import CapturedSteps
    B = 5000
    eager, graphed = _noisy_env(B, dtype), _noisy_env(B, dtype)
    acts = []
    for k in range(4):
        a = graphed.world.new_action_buffer()
        a.copy_(torch.rand(4, 5, B, dtype=dtype, device='cuda', generator=torch.Generator('cuda').manual_seed(k)))
        acts.append(a)
    eager.reset()
    graphed.reset()
    g3, g2 = CapturedSteps(graphed, acts[:3]), CapturedSteps(graphed, acts[2:4])
    assert graphed.world.noise_step == 0 and torch.equal(graphed.world.pos, eager.world.pos)
    plan = [('g3',), ('e', 3), ('g2',), ('g3',), ('e', 0), ('g2',)]
    for item in plan:
        if item[0] == 'e':
            graphed.step(acts[item[1]])
            eager.step(acts[item[1]].clone())
        else:
            g, seq = (g3, acts[:3]) if item[0] == 'g3' else (g2, acts[2:4])
            g.replay()
            for a in seq:
                eager.step(a.clone())
        assert graphed.world.noise_step == eager.world.noise_step
        assert torch.equal(graphed.world.pos, eager.world.pos) and torch.equal(graphed.world.vel, eager.world.vel), item
        assert torch.equal(graphed.world.obs[:, :B], eager.world.obs[:, :B])


def _json(out):
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    assert len(lines) == 1, out.stdout[-2000:]
    return json.loads(lines[0])


def test_two_ranks_with_u_noise_reproduce_the_unsharded_statistics():
    """SURVEY 8(e) with noise on: two ranks (child processes sharing the GPU, gloo) that draw their u_noise in the kernels,
    keyed by global env id, give the all-reduced episode statistics of the unsharded noisy run."""
    common = ['--steps', '50', '--warmup', '0', '--actions', 'by-env-id', '--no-cpu-baseline', '--no-other-configs',
              '--u-noise', '0.3']
    per_rank = 50_000
    env = {k: v for k, v in os.environ.items() if k not in ('RANK', 'LOCAL_RANK', 'WORLD_SIZE', 'MASTER_ADDR', 'MASTER_PORT')}
    whole = _json(subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--envs', str(2 * per_rank)] + common,
                                 capture_output=True, text=True, timeout=300, cwd=ROOT, env=env))
    two = _json(subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--backend', 'gloo',
                                '--envs', str(per_rank)] + common, capture_output=True, text=True, timeout=400, cwd=ROOT, env=env))
    quiet = _json(subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--envs', str(2 * per_rank)] + common[:-2],
                                 capture_output=True, text=True, timeout=300, cwd=ROOT, env=env))
    a, b, q = (np.array(x['episode_stats_vector']) for x in (whole, two, quiet))
    assert whole['config']['u_noise'] == 0.3 and two['n_gpus'] == 2
    np.testing.assert_allclose(b, a, rtol=1e-12, atol=0)
    n = whole['config']['agents']
    np.testing.assert_array_equal(a[n:], b[n:])
    assert not np.array_equal(a[:n], q[:n])               # the noise really acted


def test_device_noise_rejects_a_misaligned_step_word():
    spec = O.simple_tag_spec(u_noise=0.3)
    sim = HipSim(spec, 64, torch.float64)
    act = torch.zeros(4, 5, 64, dtype=torch.float64, device='cuda')
    b = sim.buffers((SEED, 0, 0), act=act, obs=sim.obs, rew=sim.rew)
    b.noise_step_dev = sim.pos.data_ptr() + 2
    assert lib.mpe_env_step_f64(sim.desc, 64, 64, b, sim.stream) == -4
    b.noise_step_dev = None
    b.flags |= 32
    assert lib.mpe_env_step_f64(sim.desc, 64, 64, b, sim.stream) == -6
    torch.cuda.synchronize()
