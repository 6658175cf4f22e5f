"""Pin the CPU oracle (oracle/mpe_oracle.py) against golden vectors produced by the reference
itself (tests/golden/gen_golden.py).  Bit-exact: the oracle repeats the reference's float64
operations in the reference's order."""
import os

import dataclasses

import numpy as np
import pytest

from oracle import mpe_oracle as O

CASES = [
    ('g1_simple_tag', lambda: O.simple_tag_spec()),
    ('g2_simple_spread', lambda: O.simple_spread_spec()),
    ('g3_scaled_tag', lambda: O.simple_tag_spec(48, 16, 32)),
    ('g3d_scaled_tag_dense', lambda: O.simple_tag_spec(48, 16, 32)),
    ('g4_simple_tag_unoise', lambda: O.simple_tag_spec(u_noise=0.3)),
    ('g5_simple_tag_dense', lambda: O.simple_tag_spec()),
    ('g5_simple_spread_dense', lambda: O.simple_spread_spec()),
    ('g6_simple_tag_oob', lambda: O.simple_tag_spec()),
    ('g7_simple', lambda: O.simple_spec()),
    ('g8_simple_adversary', lambda: O.simple_adversary_spec()),
    ('g8_simple_adversary_dense', lambda: O.simple_adversary_spec()),
    ('g9_simple_push', lambda: O.simple_push_spec()),
    ('g9_simple_push_dense', lambda: O.simple_push_spec()),
]


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, name + '.npz'))


class _FixedNoise:
    """Feeds recorded standard normals to SerialEnv.world_step in the reference's draw order
    (core.py:147: one randn(2) per agent, agents ascending)."""

    def __init__(self, noise):
        self.noise = noise
        self.i = 0

    def __call__(self, n):
        assert n == 2
        row = self.noise[self.i]
        self.i += 1
        return row


@pytest.mark.parametrize('name,mk', CASES)
def test_serial_oracle_bit_exact(golden_dir, name, mk, monkeypatch):
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    n_serial = min(S, 16) if spec.n_entities < 20 else 1      # scaled config: 1 seed is ~2 s
    n_steps = T if spec.n_entities < 20 else min(T, 3)
    for s in range(n_serial):
        env = O.SerialEnv(spec, benchmark=True)
        env.set_state(g['pos0'][s], g['vel0'][s])
        env.goal = max(int(g['goal'][s]), 0)
        assert np.array_equal(np.concatenate([env.observation(i) for i in range(env.n)]), g['obs0'][s])
        for t in range(n_steps):
            if 'noise' in g.files:
                monkeypatch.setattr(np.random, 'randn', _FixedNoise(g['noise'][s, t]))
            obs_n, rew_n, done_n, info_n = env.step(list(g['act'][s, t]))
            pos, vel = env.get_state()
            assert np.array_equal(pos, g['pos'][s, t]), (name, s, t)
            assert np.array_equal(vel, g['vel'][s, t]), (name, s, t)
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]), (name, s, t)
            assert np.array_equal(np.asarray(rew_n, float), g['rew'][s, t]), (name, s, t)
            W = spec.info_width
            info = np.stack([np.concatenate([np.atleast_1d(np.asarray(x, float)), np.zeros(W)])[:W]
                             if not isinstance(x, dict) else np.zeros(0) for x in info_n['n']])
            assert np.array_equal(info, g['info'][s, t]), (name, s, t)
            assert done_n == [False] * env.n


@pytest.mark.parametrize('name,mk', CASES)
def test_batched_oracle_bit_exact(golden_dir, name, mk):
    """All seeds of a case as one batch; free-running trajectory from (pos0, vel0)."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    bo = O.BatchedOracle(spec, np.float64)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))   # [E,2,S]
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    goal = np.maximum(g['goal'], 0)
    obs, _, _ = bo.observe(pos, vel, goal=goal)
    assert np.array_equal(np.concatenate(obs, 0), g['obs0'].T)
    for t in range(T):
        act = np.moveaxis(g['act'][:, t], 0, -1)                 # [A,5,S]
        noise = np.moveaxis(g['noise'][:, t], 0, -1) if 'noise' in g.files else None
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act, noise, goal=goal)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)), (name, t)
        assert np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1)), (name, t)
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
        assert np.array_equal(info, np.moveaxis(g['info'][:, t], 0, -1)), (name, t)


HIER_CASES = ['h_L1F2_3Targets_100observe', 'h_L1F2_3Targets_100observe_dense',
              'h_S1L2F2_3Targets_100observe', 'h_S1L2F2_3Targets_100observe_dense',
              'h_L1F2_3Targets_075observe', 'h_L1F2_3Targets_075observe_dense',
              'h_L1F1_10Targets_wide', 'h_L1F1_10Targets_wide_dense',
              'h_L1F4_3Targets_wide', 'h_L1F4_3Targets_wide_dense',
              'h_S1L2F2_10Targets_wide', 'h_S1L2F2_10Targets_wide_dense',
              'h_S1L2F2_10Targets_wide_rewardFixed', 'h_S1L2F2_10Targets_wide_rewardFixed_dense',
              'h_L1F2_10Targets_wide', 'h_L1F2_10Targets_wide_dense',
              'h_S1L1F1_10Targets_wide', 'h_S1L1F1_10Targets_wide_dense',
              'h_L1F2_3Targets_090observe', 'h_L1F2_3Targets_090observe_dense',
              'h_L1F2_3Targets_095observe', 'h_L1F2_3Targets_095observe_dense',
              'h_L1F2_3Targets_105observe', 'h_L1F2_3Targets_105observe_dense',
              'h_S1L2F2_3Targets_075observe', 'h_S1L2F2_3Targets_075observe_dense',
              'h_S1L2F2_3Targets_090observe', 'h_S1L2F2_3Targets_090observe_dense',
              'h_S1L2F2_3Targets_095observe', 'h_S1L2F2_3Targets_095observe_dense',
              'h_S1L2F2_3Targets_105observe', 'h_S1L2F2_3Targets_105observe_dense',
              # the *_MovingTarget files: a moving prey instead of landmarks, own observation / reward text
              'm_L1F4_MovingTarget', 'm_L1F4_MovingTarget_dense', 'm_L1F4_MovingTarget_crowd',
              'm_S1L2F4_MovingTarget', 'm_S1L2F4_MovingTarget_dense', 'm_S1L2F4_MovingTarget_crowd']


def _split_actions(g, s, t):
    dims = g['act_dims']
    offs = np.concatenate([[0], np.cumsum(dims)])
    return [g['act'][s, t, offs[i]:offs[i + 1]] for i in range(len(dims))]


@pytest.mark.parametrize('name', HIER_CASES)
def test_serial_oracle_hier_bit_exact(golden_dir, name):
    """leader force (core.py:153-194), masked / distance-sorted observations and
    hierarchical rewards of the fork's L1F*/S1L* scenarios."""
    g = _load(golden_dir, name)
    spec = O.hier_spec_from_golden(g)
    assert spec.act_dims == list(g['act_dims'])
    S, T = g['act'].shape[:2]
    for s in range(min(S, 8)):
        env = O.SerialEnv(spec, benchmark=True)
        env.set_state(g['pos0'][s], g['vel0'][s])
        for a, lf in zip(env.agents, g['lforce0'][s]):
            a.leader_force = lf.copy()
        assert np.array_equal(np.concatenate([env.observation(i) for i in range(env.n)]), g['obs0'][s])
        for t in range(T):
            obs_n, rew_n, done_n, info_n = env.step(_split_actions(g, s, t))
            pos, vel = env.get_state()
            assert np.array_equal(pos, g['pos'][s, t]), (name, s, t)
            assert np.array_equal(vel, g['vel'][s, t]), (name, s, t)
            assert np.array_equal(np.stack([a.leader_force for a in env.agents]), g['lforce'][s, t]), (name, s, t)
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]), (name, s, t)
            assert np.array_equal(np.asarray(rew_n, float), g['rew'][s, t]), (name, s, t)
            assert np.array_equal(np.asarray(info_n['n'], float)[:, None], g['info'][s, t]), (name, s, t)


@pytest.mark.parametrize('name', HIER_CASES)
def test_batched_oracle_hier_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    spec = O.hier_spec_from_golden(g)
    S, T = g['act'].shape[:2]
    bo = O.BatchedOracle(spec)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    obs, _, _ = bo.observe(pos, vel, lforce=np.moveaxis(g['lforce0'], 0, -1))
    assert np.array_equal(np.concatenate(obs, 0), g['obs0'].T)
    for t in range(T):
        pos, vel, obs, rew, info, lf = bo.env_step(pos, vel, g['act'][:, t].T, return_lforce=True)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)), (name, t)
        assert np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1)), (name, t)
        assert np.array_equal(lf, np.moveaxis(g['lforce'][:, t], 0, -1)), (name, t)
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
        assert np.array_equal(info, np.moveaxis(g['info'][:, t], 0, -1)), (name, t)


def test_hier_goldens_exercise_the_branches(golden_dir):
    g = _load(golden_dir, 'h_S1L2F2_3Targets_100observe_dense')
    assert (g['rew'] >= 10).any()                                        # landmark contacts
    assert (np.abs(g['lforce'][:, :, :2]) > 0).any()                      # followers are pushed
    lf = g['lforce'][:, :, 2:4]                                           # leaders observe the super-leader's c[0:2]
    assert ((lf >= 0) & (lf <= 1)).all() and (lf > 0).any()
    spec = O.hier_spec_from_golden(g)
    n_t = 2 * min(spec.n_landmarks, 3)
    tail = g['obs'][:, :, spec.obs_dim(0) - n_t:spec.obs_dim(0)]          # follower 0's target slots
    assert (tail != 0).any() and (tail == 0).any()                        # some visible, some masked
    g = _load(golden_dir, 'h_S1L2F2_10Targets_wide_rewardFixed')
    r = g['rew'][:, :, 4]           # is_observed() compares a target with itself: constant +1 per (target, leader)
    assert r.max() == 20.0 and (r <= 20.0).all() and (r == 20.0).mean() > 0.5


def test_moving_target_goldens_exercise_the_branches(golden_dir):
    for name in ('m_L1F4_MovingTarget_crowd', 'm_S1L2F4_MovingTarget_crowd'):
        g = _load(golden_dir, name)
        assert (g['rew'][:, :, 0] >= 89).any()                            # count >= 4: the +50 bonus
        assert (g['rew'][:, :, -1] <= -3).any()                           # the prey pays 1 per touching adversary
    g = _load(golden_dir, 'm_S1L2F4_MovingTarget')
    spec = O.hier_spec_from_golden(g)
    assert spec.kind == O.HIER_MOVING and spec.reward_variant == 1
    assert spec.obs_dims == [34] * 4 + [32] * 4                           # only the followers observe leader_force
    assert (g['obs'][:, :, 4:18] == 0).any() and (g['obs'][:, :, 4:18] != 0).any()
    g = _load(golden_dir, 'm_L1F4_MovingTarget')
    spec = O.hier_spec_from_golden(g)
    assert spec.reward_variant == 0 and spec.obs_dims == [26] * 4 + [24] * 2 and spec.dim_c == 3
    prey = g['obs'][:, :, sum(spec.obs_dims[:5]):]                        # absolute positions of the others
    assert np.array_equal(prey[:, :, 4:6], g['pos'][:, :, 0])


def test_golden_cases_exercise_the_branches(golden_dir):
    """The dense / out-of-bounds fixtures really hit collisions, the clamp and all bound() arms."""
    g = _load(golden_dir, 'g5_simple_tag_dense')
    assert (g['rew'][:, :, 0] >= 10).any() and (g['rew'][:, :, 3] <= -10).any()
    assert (g['info'][:, :, :3] > 0).any()
    speed = np.sqrt((g['vel'] ** 2).sum(-1))
    assert np.isclose(speed[:, :, :3], 1.0, atol=1e-12).any()        # adversary clamp fired
    g = _load(golden_dir, 'g6_simple_tag_oob')
    x = np.abs(g['pos'][:, :, 3])                                   # the good agent
    assert (x < 0.9).any() and ((x >= 0.9) & (x < 1.0)).any()
    assert ((x >= 1.0) & (x < 2.15)).any() and (x > 2.16).any()
    g = _load(golden_dir, 'g5_simple_spread_dense')
    assert (g['info'][:, :, :, 1] > 1).any() and (g['info'][:, :, :, 3] > 0).any()


def test_kat1_one_sided_contact(golden_dir):
    """SURVEY KAT-1: only the lower-indexed entity of a pair is pushed (core.py:203-207)."""
    g = _load(golden_dir, 'kat1_one_sided')
    spec = O.simple_tag_spec()
    env = O.SerialEnv(spec)
    env.set_state(g['pos0'], g['vel0'])
    env.step([np.zeros(5)] * 4)
    pos, vel = env.get_state()
    assert np.array_equal(pos, g['pos']) and np.array_equal(vel, g['vel'])
    np.testing.assert_allclose(vel[0], [-0.5, 0.0], atol=1e-12)
    np.testing.assert_allclose(pos[0], [-0.05, 0.0], atol=1e-12)
    assert np.all(np.abs(vel[1]) < 1e-100)                         # agent 1 receives nothing
    bo = O.BatchedOracle(spec)
    p, v = bo.world_step(g['pos0'][..., None], g['vel0'][..., None], np.zeros((4, 5, 1)))
    assert np.array_equal(p[..., 0], g['pos']) and np.array_equal(v[..., 0], g['vel'])


def test_kat2_spread_self_collision(golden_dir):
    """SURVEY KAT-2: every agent's reward includes -1 for 'colliding' with itself."""
    g = _load(golden_dir, 'kat2_spread_self')
    spec = O.simple_spread_spec()
    env = O.SerialEnv(spec)
    env.set_state(g['pos0'], g['vel0'])
    rew = np.array([env.reward(i) for i in range(3)], float)
    assert np.array_equal(rew, g['rew']) and np.array_equal(rew, [-1.0, -1.0, -1.0])
    info = np.stack([np.asarray(env.info(i), float) for i in range(3)])
    assert np.array_equal(info, g['info'])
    _, r, inf = O.BatchedOracle(spec).observe(g['pos0'][..., None], g['vel0'][..., None])
    assert np.array_equal(r[:, 0], g['rew']) and np.array_equal(inf[..., 0], g['info'])


def test_kat3_spaces():
    """SURVEY KAT-3: observation dims."""
    assert O.simple_tag_spec().obs_dims == [16, 16, 16, 14]
    assert O.simple_spread_spec().obs_dims == [18, 18, 18]
    d = O.simple_tag_spec(48, 16, 32).obs_dims
    assert set(d[:48]) == {226} and set(d[48:]) == {224}


def test_kat4_far_apart_zero_action(golden_dir):
    """SURVEY KAT-4: pos bitwise unchanged, vel tiny but not exactly zero (softplus tail)."""
    g = _load(golden_dir, 'kat4_far_zero')
    bo = O.BatchedOracle(O.simple_tag_spec())
    p, v = bo.world_step(g['pos0'][..., None], g['vel0'][..., None], np.zeros((4, 5, 1)))
    assert np.array_equal(p[..., 0], g['pos']) and np.array_equal(v[..., 0], g['vel'])
    assert np.array_equal(g['pos'], g['pos0'])
    assert 0 < np.abs(g['vel']).max() < 1e-100


def test_serial_reset_uses_reference_draw_order(golden_dir):
    """np.random.seed(s); make_world (one reset) ; env.reset() -> same pos0 as the reference
    (agents ascending then landmarks ascending, 2 uniforms each: simple_tag.py:47-54)."""
    for name, spec in (('g1_simple_tag', O.simple_tag_spec()), ('g2_simple_spread', O.simple_spread_spec()),
                       ('g7_simple', O.simple_spec()), ('g8_simple_adversary', O.simple_adversary_spec()),
                       ('g9_simple_push', O.simple_push_spec())):
        g = _load(golden_dir, name)
        for s in g['seeds'][:8]:
            np.random.seed(int(s))
            env = O.SerialEnv(spec)
            env.reset()             # make_world() -> reset_world (simple_tag.py:35)
            obs = env.reset()       # env.reset()
            pos, vel = env.get_state()
            assert np.array_equal(pos, g['pos0'][s]) and np.array_equal(vel, g['vel0'][s])
            assert np.array_equal(np.concatenate(obs), g['obs0'][s])
            if spec.kind in O.GOAL_KINDS:       # goal = np.random.choice(world.landmarks) drawn first
                assert env.goal == g['goal'][s]


def test_philox_known_answers():
    """Random123 kat_vectors for philox4x32-10."""
    def run(c, k):
        out = O.philox4x32_10([np.array([x], np.uint32) for x in c], [np.array([x], np.uint32) for x in k])
        return [int(x[0]) for x in out]
    assert run([0, 0, 0, 0], [0, 0]) == [0x6627e8d5, 0xe169c58d, 0xbc57ac4c, 0x9b00dbd8]
    assert run([0xffffffff] * 4, [0xffffffff] * 2) == [0x408f276d, 0x41c83b0e, 0xa20bc7c6, 0x6d5451fd]
    assert run([0x243f6a88, 0x85a308d3, 0x13198a2e, 0x03707344], [0xa4093822, 0x299f31d0]) == \
        [0xd16cfe09, 0x94fdcceb, 0x5001e420, 0x24126ea1]


def test_reset_positions_ranges_and_independence():
    spec = O.simple_tag_spec()
    ids = np.arange(4096, dtype=np.uint64)
    p = O.reset_positions(spec, seed=7, episode=3, env_ids=ids)
    assert p.shape == (6, 2, 4096)
    assert np.abs(p[:4]).max() < 1.0 and np.abs(p[4:]).max() < 0.9
    assert abs(p[:4].mean()) < 0.02 and abs(p[:4].std() - 1 / np.sqrt(3)) < 0.02
    # shard-independent: envs [1000,2000) computed alone equal the slice
    q = O.reset_positions(spec, seed=7, episode=3, env_ids=ids[1000:2000])
    assert np.array_equal(q, p[:, :, 1000:2000])
    assert not np.array_equal(O.reset_positions(spec, 7, 4, ids), p)


def test_target_move_oracle_forms_agree(golden_dir):
    """SURVEY 8f row f3, experiments/train_and_eval.py:207-222.  PARITY UNPINNED: no reference fixture
    covers target_move and running the reference function here was refused, so the
    two restatements are only checked against each other: the serial form (reference loop shape) and
    the vectorised form must respawn the same landmarks and report the same count; untouched entities
    keep their positions; respawned ones land in [-1, 1)^2."""
    g = _load(golden_dir, 'h_S1L2F2_10Targets_wide_dense')
    spec = O.hier_spec_from_golden(g)
    assert spec.advfollower == [True, True, False, False, False]
    A = spec.n_agents
    total = 0
    for s in range(g['pos'].shape[0]):
        for t in (0, 5, 24, 100, 105, 124):
            pos = g['pos'][s, t % 100] * (0.12 if t >= 100 else 1.0)   # [E,2]; t >= 100: a crowded copy
            env = O.SerialEnv(spec)
            env.set_state(pos, g['vel'][s, t % 100])
            np.random.seed(s)
            n = O.target_move_serial(env)
            after = env.get_state()[0]
            bpos, cnt = O.target_move(spec, pos[:, :, None], seed=7, counter=t, env_ids=[s])
            moved = np.any(after != pos, axis=1)
            assert cnt[0] == n == moved.sum() and not moved[:A].any()
            assert np.array_equal(np.any(bpos[:, :, 0] != pos, axis=1), moved)
            assert (np.abs(bpos[:, :, 0][moved]) <= 1).all() and (np.abs(after[moved]) <= 1).all()
            total += n
    assert total > 10


def test_scripted_prey_oracle_forms_agree(golden_dir):
    """SURVEY 8f row f3, experiments/simple_train.py:157-205 (good_agent_action / good_agent_victim_action).
    PARITY UNPINNED: the training script holds no fixture for them and its code is not run here (DESIGN.md
    section 4).  The serial form follows the reference text with numpy's own np.linalg.norm; the vectorised
    form must give the same actions bit for bit on the MovingTarget trajectories, and the documented
    properties hold: one slot per axis, capped at 1.3, pointing at the chosen corner / the target."""
    g = _load(golden_dir, 'm_L1F4_MovingTarget')
    spec = O.hier_spec_from_golden(g)
    S, T = g['pos'].shape[:2]
    corners = set()
    for s in range(S):
        for t in range(0, T, 3):
            pos = g['pos'][s, t] * (2.0 if t % 2 else 1.0)             # some states outside the arena: the cap
            env = O.SerialEnv(spec)
            env.set_state(pos, g['vel'][s, t])
            act, edge = O.good_agent_action_serial(env)
            bact, bedge = O.scripted_action(spec, pos[:, :, None], -1, 0)
            assert np.array_equal(act, bact[:, 0]) and np.array_equal(edge, bedge[:, 0])
            corners.add(tuple(edge))
            assert act[0] == 0 and (act >= 0).all() and act.max() <= 1.3
            assert act[1] * act[2] == 0 and act[3] * act[4] == 0
            want = (edge - pos[-1]) / 2
            assert np.allclose(np.minimum(np.abs(want), 1.3), [act[1] + act[2], act[3] + act[4]], rtol=0, atol=0)
            assert (act[1] > 0) == (want[0] > 0) and (act[3] > 0) == (want[1] > 0)
            d = [sum(np.linalg.norm(np.array(e) - p) for p in pos[:-1]) for e in O.WORLD_EDGE]
            assert list(edge) == O.WORLD_EDGE[int(np.argmax(d))]
            for target in (0, 4):
                vact = O.good_agent_victim_action_serial(env, target)
                bv, bp = O.scripted_action(spec, pos[:, :, None], -1, 1, target)
                assert np.array_equal(vact, bv[:, 0]) and np.array_equal(bp[:, 0], pos[target])
                off = (pos[target] - pos[-1]) / 2
                assert np.array_equal(np.minimum(np.abs(off) * 7, 1.3), [vact[1] + vact[2], vact[3] + vact[4]])
    assert len(corners) >= 3


COMM_CASES = [('g10_simple_speaker_listener', O.speaker_listener_spec), ('g11_simple_reference', O.simple_reference_spec),
              ('g12_simple_crypto', O.simple_crypto_spec)]


@pytest.mark.parametrize('name,mk', COMM_CASES)
def test_serial_oracle_comm_scenarios_bit_exact(golden_dir, name, mk):
    """SURVEY 8f row f1, the speaking scenarios: ragged actions (a fixed speaker has no physical
    part, environment.py:174), agent.state.c, per-agent goal_b, incl. the reference's np.random
    order at reset (np.random.choice for the goals first)."""
    g = _load(golden_dir, name)
    spec = mk()
    assert spec.act_dims == list(g['act_dims']) and spec.size == list(g['spec_size'])
    assert spec.movable == list(g['spec_movable']) and spec.silent == list(g['spec_silent'])
    assert not g['spec_shared_reward']
    offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
    S, T = g['act'].shape[:2]
    for s in range(S):
        np.random.seed(int(g['seeds'][s]))
        env = O.SerialEnv(spec, benchmark=spec.kind == O.SIMPLE_CRYPTO)
        env.reset()                                      # make_world() resets once ...
        obs0 = env.reset()                               # ... and env.reset() again
        want_goal = [int(x) for x in g['goal'][s]]
        got_goal = {O.SIMPLE_REFERENCE: env.goals, O.SPEAKER_LISTENER: [env.goals[0], -1],
                    O.SIMPLE_CRYPTO: env.goals[:2]}[spec.kind]
        assert got_goal == want_goal
        assert np.array_equal(np.concatenate(obs0), g['obs0'][s])
        for t in range(T):
            obs_n, rew_n, _, info_n = env.step([g['act'][s, t, offs[i]:offs[i + 1]] for i in range(spec.n_agents)])
            if spec.kind == O.SIMPLE_CRYPTO:
                assert np.array_equal(np.stack(info_n['n']), g['info'][s, t])
            pos, vel = env.get_state()
            assert np.array_equal(pos, g['pos'][s, t]) and np.array_equal(vel, g['vel'][s, t])
            assert np.array_equal(np.stack([a.comm for a in env.agents]), g['comm'][s, t])
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]), (name, s, t)
            assert np.array_equal(np.asarray(rew_n, float), g['rew'][s, t]), (name, s, t)


@pytest.mark.parametrize('name,mk', COMM_CASES)
def test_batched_oracle_comm_scenarios_bit_exact(golden_dir, name, mk):
    g = _load(golden_dir, name)
    spec = mk()
    bo = O.BatchedOracle(spec)
    S, T = g['act'].shape[:2]
    goal = np.maximum(g['goal'].T[:spec.goal_rows], 0)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    obs, _, _ = bo.observe(pos, vel, goal=goal, comm=np.moveaxis(g['comm0'], 0, -1))
    assert np.array_equal(np.concatenate(obs, 0), g['obs0'].T)
    for t in range(T):
        act = g['act'][:, t].T
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act, goal=goal)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)) and np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1))
        assert np.array_equal(bo.comm_state(act), np.moveaxis(g['comm'][:, t], 0, -1))
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
        if spec.kind == O.SIMPLE_CRYPTO:
            assert np.array_equal(info, np.moveaxis(g['info'][:, t], 0, -1))


WORLD_COMM_CASES = ['g13_simple_world_comm', 'g13_simple_world_comm_dense']


@pytest.mark.parametrize('name', WORLD_COMM_CASES)
def test_serial_oracle_world_comm_bit_exact(golden_dir, name):
    """simple_world_comm: tag physics with a speaking leader, forest visibility masks,
    food rewards, integer collision info; reset order incl. the second draw of food and forests."""
    g = _load(golden_dir, name)
    spec = O.simple_world_comm_spec()
    assert spec.act_dims == list(g['act_dims']) == [9, 5, 5, 5, 5, 5] and spec.size == list(g['spec_size'])
    assert spec.collide == list(g['spec_collide']) and spec.silent == list(g['spec_silent'])
    assert (spec.n_food, spec.n_forests) == (int(g['spec_n_food']), int(g['spec_n_forests']))
    assert spec.obs_dims == [34] * 4 + [28] * 2 and not g['spec_shared_reward']
    offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
    S, T = g['act'].shape[:2]
    hidden = 0
    for s in range(min(S, 16)):                          # the batched test below covers every seed
        np.random.seed(int(g['seeds'][s]))
        env = O.SerialEnv(spec, benchmark=True)
        env.reset()
        obs0 = env.reset()
        if name.endswith('dense'):                       # the fixture overwrote the state after reset
            env.set_state(g['pos0'][s], g['vel0'][s])
            obs0 = [env.observation(i) for i in range(env.n)]
        else:
            assert np.array_equal(env.get_state()[0], g['pos0'][s])
        assert np.array_equal(np.concatenate(obs0), g['obs0'][s])
        for t in range(T):
            obs_n, rew_n, _, info_n = env.step([g['act'][s, t, offs[i]:offs[i + 1]] for i in range(spec.n_agents)])
            pos, vel = env.get_state()
            assert np.array_equal(pos, g['pos'][s, t]) and np.array_equal(vel, g['vel'][s, t])
            assert np.array_equal(np.stack([a.comm for a in env.agents]), g['comm'][s, t])
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]), (name, s, t)
            assert np.array_equal(np.asarray(rew_n, float), g['rew'][s, t]), (name, s, t)
            assert np.array_equal(np.asarray(info_n['n'], float), g['info'][s, t, :, 0])
            hidden += int((obs_n[1][14:24] == 0).sum())
    assert hidden > 0                                    # some agents were hidden by a forest


@pytest.mark.parametrize('name', WORLD_COMM_CASES)
def test_batched_oracle_world_comm_bit_exact(golden_dir, name):
    g = _load(golden_dir, name)
    spec = O.simple_world_comm_spec()
    bo = O.BatchedOracle(spec)
    S, T = g['act'].shape[:2]
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    obs, _, _ = bo.observe(pos, vel, comm=np.moveaxis(g['comm0'], 0, -1))
    assert np.array_equal(np.concatenate(obs, 0), g['obs0'].T)
    for t in range(T):
        act = g['act'][:, t].T
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)) and np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1))
        assert np.array_equal(bo.comm_state(act), np.moveaxis(g['comm'][:, t], 0, -1))
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
        assert np.array_equal(info, np.moveaxis(g['info'][:, t], 0, -1))


CNOISE_CASES = [('g14_simple_reference_cnoise', O.simple_reference_spec), ('g14_simple_crypto_cnoise', O.simple_crypto_spec),
                ('g14_simple_world_comm_cnoise', O.simple_world_comm_spec)]


@pytest.mark.parametrize('name,mk', CNOISE_CASES)
def test_oracle_c_noise_bit_exact(golden_dir, name, mk):
    """core.py:224-230 with agent.c_noise set: state.c = action.c + randn(dim_c) * c_noise, drawn per
    speaking agent in agent order after the integration; the noisy state.c is what the other agents
    observe and what simple_crypto scores.  Serial form (global np.random stream, seeded per step like the
    fixture) and vectorised form (noise passed in) against the reference."""
    import dataclasses
    g = _load(golden_dir, name)
    cn = float(g['spec_c_noise'])
    spec = mk()
    spec = dataclasses.replace(spec, c_noise=[cn] * spec.n_agents)
    offs = np.concatenate([[0], np.cumsum(spec.act_dims)])
    S, T = g['act'].shape[:2]
    goal = np.maximum(g['goal'].T[:spec.goal_rows], 0) if 'goal' in g.files else None
    bo = O.BatchedOracle(spec)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    for t in range(T):
        act = g['act'][:, t].T
        cnz = np.moveaxis(g['cnoise'][:, t], 0, -1)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, act, goal=goal, cnoise=cnz)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1))
        assert np.array_equal(bo.comm_state(act, cnz), np.moveaxis(g['comm'][:, t], 0, -1))
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
    assert np.abs(g['comm'][:, :, 0] - g['act'][:, :, offs[0] + spec.comm_row0(0):offs[1]]).max() > 0.05   # noise is in
    for s in range(4):
        env = O.SerialEnv(spec, benchmark=True)
        env.set_state(g['pos0'][s], g['vel0'][s])
        if goal is not None:
            env.goals = [int(x) for x in goal[:, s]] + [0] * 3
        for t in range(T):
            np.random.seed(50_000 + 1000 * int(g['seeds'][s]) + t)
            obs_n, rew_n, _, _ = env.step([g['act'][s, t, offs[i]:offs[i + 1]] for i in range(spec.n_agents)])
            assert np.array_equal(np.stack([a.comm for a in env.agents]), g['comm'][s, t])
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]) and np.array_equal(np.asarray(rew_n, float), g['rew'][s, t])


DECODE_CASES = [('g15_simple_tag_index_actions', O.simple_tag_spec, 'index'),
                ('g15_simple_spread_index_actions', O.simple_spread_spec, 'index'),
                ('g16_simple_tag_force_discrete', O.simple_tag_spec, 'force'),
                ('g16_simple_push_force_discrete', O.simple_push_spec, 'force')]


@pytest.mark.parametrize('name,mk,mode', DECODE_CASES)
def test_oracle_non_default_action_decodes_bit_exact(golden_dir, name, mk, mode):
    """environment.py:176-187: env.discrete_action_input (one integer per agent; 1/2 -> u[0] = -1/+1, 3/4 -> u[1] =
    -1/+1, anything else no force) and env.force_discrete_action (argmax -> one-hot, first maximum).  The fixtures
    hold what was passed to env.step; serial and vectorised restatement against the reference's trajectories."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    goal = np.maximum(g['goal'], 0)
    bo = O.BatchedOracle(spec, discrete_action_input=mode == 'index', force_discrete_action=mode == 'force')
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    for t in range(T):
        pos, vel, obs, rew, info = bo.env_step(pos, vel, np.moveaxis(g['act'][:, t], 0, -1), None, goal=goal)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)), (name, t)
        assert np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1)), (name, t)
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
    for s in range(min(S, 4)):
        env = O.SerialEnv(spec, benchmark=True)
        env.discrete_action_input = mode == 'index'
        env.force_discrete_action = mode == 'force'
        env.set_state(g['pos0'][s], g['vel0'][s])
        env.goal = max(int(g['goal'][s]), 0)
        for t in range(T):
            acts = [int(a[0]) for a in g['act'][s, t]] if mode == 'index' else [a.copy() for a in g['act'][s, t]]
            obs_n, rew_n, _, _ = env.step(acts)
            p, v = env.get_state()
            assert np.array_equal(p, g['pos'][s, t]) and np.array_equal(v, g['vel'][s, t]), (name, s, t)
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]), (name, s, t)
            if mode == 'force':      # in place, like the reference: the caller's arrays are one-hot afterwards
                assert all(sorted(a) == [0, 0, 0, 0, 1] for a in acts)
    if mode == 'index':
        k = g['act'][:, :, :, 0]
        assert set(np.unique(k)) == set(range(8))                       # incl. 0 and the out-of-range 5..7
    else:
        a = g['act']
        assert ((a == a.max(-1, keepdims=True)).sum(-1) > 1).any()      # ties: the first maximum decides


CONTINUOUS_CASES = [('g18_simple_tag_continuous', O.simple_tag_spec, False),
                    ('g18_simple_spread_continuous', O.simple_spread_spec, False),
                    ('g18_simple_tag_continuous_force', O.simple_tag_spec, True)]


@pytest.mark.parametrize('name,mk,forced', CONTINUOUS_CASES)
def test_oracle_continuous_action_space_bit_exact(golden_dir, name, mk, forced):
    """environment.py:191-192 (`env.discrete_action_space = False`, flipped after construction): the action is the
    force command u itself (columns 0, 1 of the fixture's 'act'), times the sensitivity; with force_discrete_action
    the argmax -> one-hot of :184-187 runs on the 2-vector first.  Serial and vectorised restatement against the
    reference's trajectories."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    bo = O.BatchedOracle(spec, discrete_action_space=False, force_discrete_action=forced)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    for t in range(T):
        pos, vel, obs, rew, info = bo.env_step(pos, vel, np.moveaxis(g['act'][:, t], 0, -1), None)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)), (name, t)
        assert np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1)), (name, t)
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew, g['rew'][:, t].T), (name, t)
    for s in range(min(S, 4)):
        env = O.SerialEnv(spec, benchmark=True)
        env.discrete_action_space = False
        env.force_discrete_action = forced
        env.set_state(g['pos0'][s], g['vel0'][s])
        for t in range(T):
            obs_n, rew_n, _, _ = env.step([a[:2].copy() for a in g['act'][s, t]])
            p, v = env.get_state()
            assert np.array_equal(p, g['pos'][s, t]) and np.array_equal(v, g['vel'][s, t]), (name, s, t)
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]), (name, s, t)
    if forced:
        u = g['act'][..., :2]
        assert (u[..., 0] == u[..., 1]).any()                           # ties: the first maximum decides


MOVABLE_CASES = [('g19_simple_tag_movable_landmarks',
                  lambda: dataclasses.replace(O.simple_tag_spec(), movable=[True] * 6, mass=[1, 1, 1, 1, 1, 2.5],
                                              landmark_max_speed=[None, 0.08])),
                 ('g19_simple_spread_movable_landmarks',
                  lambda: dataclasses.replace(O.simple_spread_spec(), movable=[True] * 3 + [True, False, True],
                                              collide=[True] * 6, size=[0.15] * 3 + [0.12] * 3,
                                              landmark_max_speed=[None, None, 0.3]))]


@pytest.mark.parametrize('name,mk', MOVABLE_CASES)
def test_oracle_movable_landmarks_bit_exact(golden_dir, name, mk):
    """core.py:211-222 on movable LANDMARKS (fixtures generated from the reference with Entity.movable / initial_mass /
    max_speed set on landmarks; no scenario file does): both oracle forms reproduce positions, agent and landmark
    velocities, observations and rewards bit for bit; the speed clamp and landmark-landmark pushes fire."""
    g = _load(golden_dir, name)
    spec = mk()
    S, T = g['act'].shape[:2]
    bo = O.BatchedOracle(spec)
    mv = lambda a: np.ascontiguousarray(np.moveaxis(a, 0, -1))
    pos, vel, lvel = mv(g['pos0']), mv(g['vel0']), mv(g['lvel0'])
    for t in range(T):
        pos, vel, obs, rew, info, lvel = bo.env_step(pos, vel, mv(g['act'][:, t]), lvel=lvel)
        assert np.array_equal(pos, mv(g['pos'][:, t])) and np.array_equal(vel, mv(g['vel'][:, t])), (name, t)
        assert np.array_equal(lvel, mv(g['lvel'][:, t])), (name, t)
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T) and np.array_equal(rew, g['rew'][:, t].T)
    for s in range(3):
        env = O.SerialEnv(spec, benchmark=True)
        env.set_state(g['pos0'][s], g['vel0'][s], g['lvel0'][s])
        for t in range(T):
            obs_n, rew_n, _, _ = env.step([a.copy() for a in g['act'][s, t]])
            p, v = env.get_state()
            assert np.array_equal(p, g['pos'][s, t]) and np.array_equal(v, g['vel'][s, t]), (name, s, t)
            assert np.array_equal(np.stack([lm.vel for lm in env.landmarks]), g['lvel'][s, t])
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t])
    lv = g['lvel']
    moved = np.abs(g['pos'][:, -1, spec.n_agents:] - g['pos0'][:, spec.n_agents:]).max(axis=(0, 2))
    assert all((m > 0) == bool(spec.movable[spec.n_agents + i]) for i, m in enumerate(moved))   # only the movable ones moved
    ms = [m for m in spec.landmark_max_speed if m is not None][0]
    l = [i for i, m in enumerate(spec.landmark_max_speed) if m is not None][0]
    sp = np.sqrt((lv[:, :, l] ** 2).sum(-1))
    assert sp.max() <= ms * (1 + 1e-12) and (np.abs(sp - ms) < 1e-12).any()                     # the clamp fired


def test_device_normals_are_standard_normal_and_keyed_by_global_env_id():
    """The in-kernel u_noise / c_noise draw (MPE_FLAG_DEVICE_NOISE; restated in oracle.device_normals): moments of a
    standard normal, agreement with libm's log / sin / cos to ~1e-15, and a draw that depends on (seed, env id, step,
    word) only -- a shard of a batch draws the unsharded batch's numbers."""
    ids = np.arange(400_000, dtype=np.uint64)
    z0, z1 = O.device_normals(11, 5, ids, O.NOISE_WORD_U | 2)
    z = np.concatenate([z0, z1])
    assert np.isfinite(z).all() and abs(z.mean()) < 5e-3 and abs(z.std() - 1) < 5e-3
    assert abs((z ** 3).mean()) < 2e-2 and abs((z ** 4).mean() - 3) < 5e-2 and abs(np.corrcoef(z0, z1)[0, 1]) < 5e-3
    lo = (ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    w = O.philox4x32_10([lo, np.zeros_like(lo), np.full_like(lo, 5), np.full_like(lo, O.NOISE_WORD_U | 2)],
                        [np.full_like(lo, 11), np.zeros_like(lo)])
    u1 = ((w[0] >> np.uint32(5)).astype(float) * 67108864.0 + (w[1] >> np.uint32(6)).astype(float) + 1.0) * 2.0 ** -53
    u2 = ((w[2] >> np.uint32(5)).astype(float) * 67108864.0 + (w[3] >> np.uint32(6)).astype(float)) * 2.0 ** -53
    r = np.sqrt(-2 * np.log(u1))
    assert np.abs(z0 - r * np.cos(2 * np.pi * u2)).max() < 1e-13 and np.abs(z1 - r * np.sin(2 * np.pi * u2)).max() < 1e-13
    a0, a1 = O.device_normals(11, 5, ids[1000:2000], O.NOISE_WORD_U | 2)
    assert np.array_equal(a0, z0[1000:2000]) and np.array_equal(a1, z1[1000:2000])
    b0, _ = O.device_normals(11, 6, ids[:1000], O.NOISE_WORD_U | 2)
    c0, _ = O.device_normals(12, 5, ids[:1000], O.NOISE_WORD_U | 2)
    d0, _ = O.device_normals(11, 5, ids[:1000], O.NOISE_WORD_U | 3)
    assert not np.array_equal(b0, z0[:1000]) and not np.array_equal(c0, z0[:1000]) and not np.array_equal(d0, z0[:1000])
    spec = O.simple_tag_spec(u_noise=0.3)
    n = O.device_u_noise(spec, 11, 5, ids[:64])
    assert n.shape == (4, 2, 64) and np.array_equal(n[2, 0], z0[:64]) and np.array_equal(n[2, 1], z1[:64])


def scripted_callbacks(name):
    """The callbacks of tests/golden/gen_golden.py:SCRIPTED restated on (pos [E, 2, ...], agent index) -> u [2, ...]: capped
    proportional controllers towards an entity."""
    def toward(target, gain, cap):
        return lambda pos, i: np.clip(gain * (pos[target] - pos[i]), -cap, cap)
    if name.startswith('s1_'):
        return {3: toward(4, 2.0, 4.0)}                  # the prey heads for landmark 0 (entity 4)
    return {0: toward(3, 4.0, 3.0), 2: toward(3, 1.5, 3.0)}     # two of the three adversaries chase agent 3


SCRIPTED_CASES = ['s1_simple_tag_scripted_prey', 's1_simple_tag_scripted_prey_dense',
                  's2_simple_tag_scripted_chasers', 's2_simple_tag_scripted_chasers_dense']


@pytest.mark.parametrize('name', SCRIPTED_CASES)
def test_oracle_scripted_agents_bit_exact(golden_dir, name, monkeypatch):
    """core.py:126-127 + environment.py:21-22: agents with an action_callback get their force from it (no sensitivity; the
    u_noise of core.py:147 still applies) and disappear from the env's lists.  Serial and vectorised restatement against
    the reference, incl. the recorded callback outputs."""
    g = _load(golden_dir, name)
    un = float(g['u_noise'])
    spec = O.simple_tag_spec(u_noise=un if un else None)
    cbs = scripted_callbacks(name)
    assert sorted(cbs) == list(g['scripted'])
    policy = [i for i in range(spec.n_agents) if i not in cbs]
    S, T = g['act'].shape[:2]
    rows = np.concatenate([np.arange(sum(spec.obs_dims[:i]), sum(spec.obs_dims[:i + 1])) for i in policy])
    bo = O.BatchedOracle(spec)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    obs, _, _ = bo.observe(pos, vel)
    assert np.array_equal(np.concatenate(obs, 0)[rows], g['obs0'].T)
    for t in range(T):
        u = {i: cb(pos, i) for i, cb in cbs.items()}
        for i in cbs:
            assert np.array_equal(u[i], g['u_scripted'][:, t, i].T), (name, t, i)
        pos, vel, obs, rew, info = bo.env_step(pos, vel, np.moveaxis(g['act'][:, t], 0, -1),
                                               np.moveaxis(g['noise'][:, t], 0, -1) if un else None, scripted_u=u)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)), (name, t)
        assert np.array_equal(vel, np.moveaxis(g['vel'][:, t], 0, -1)), (name, t)
        assert np.array_equal(np.concatenate(obs, 0)[rows], g['obs'][:, t].T), (name, t)
        assert np.array_equal(rew[policy], g['rew'][:, t].T), (name, t)
        assert np.array_equal(info[policy, 0], g['info'][:, t, :, 0].T), (name, t)
    for s in range(4):
        env = O.SerialEnv(spec, benchmark=True)
        env.scripted = {i: (lambda e, k, cb=cb: cb(np.stack([b.pos for b in e.bodies]), k)) for i, cb in cbs.items()}
        env.set_state(g['pos0'][s], g['vel0'][s])
        for t in range(T):
            if un:
                monkeypatch.setattr(np.random, 'randn', _FixedNoise(g['noise'][s, t]))
            obs_n, rew_n, done_n, info_n = env.step([g['act'][s, t, i] for i in policy])
            assert len(obs_n) == len(policy)
            p, v = env.get_state()
            assert np.array_equal(p, g['pos'][s, t]) and np.array_equal(v, g['vel'][s, t]), (name, s, t)
            assert np.array_equal(np.concatenate(obs_n), g['obs'][s, t]) and np.array_equal(np.asarray(rew_n, float), g['rew'][s, t])


def test_oracle_force_discrete_on_multidiscrete_actions_bit_exact(golden_dir):
    """environment.py:161-168,184-187: with a MultiDiscrete action only action[0], the 5 physical entries, goes through the
    argmax -> one-hot; the command entries keep their soft values (and still pick the leader's command by argmax)."""
    g = _load(golden_dir, 'g16_h_S1L2F2_3Targets_100observe_force_discrete')
    spec = O.hier_spec_from_golden(g)
    S, T = g['act'].shape[:2]
    bo = O.BatchedOracle(spec, force_discrete_action=True)
    pos = np.ascontiguousarray(np.moveaxis(g['pos0'], 0, -1))
    vel = np.ascontiguousarray(np.moveaxis(g['vel0'], 0, -1))
    for t in range(T):
        pos, vel, obs, rew, info, lf = bo.env_step(pos, vel, g['act'][:, t].T, return_lforce=True)
        assert np.array_equal(pos, np.moveaxis(g['pos'][:, t], 0, -1)), t
        assert np.array_equal(lf, np.moveaxis(g['lforce'][:, t], 0, -1)), t
        assert np.array_equal(np.concatenate(obs, 0), g['obs'][:, t].T), t
        assert np.array_equal(rew, g['rew'][:, t].T), t
    for s in range(3):
        env = O.SerialEnv(spec, benchmark=True)
        env.force_discrete_action = True
        env.set_state(g['pos0'][s], g['vel0'][s])
        for a, lf in zip(env.agents, g['lforce0'][s]):
            a.leader_force = lf.copy()
        for t in range(T):
            obs_n, rew_n, _, _ = env.step([a.copy() for a in _split_actions(g, s, t)])
            assert np.array_equal(env.get_state()[0], g['pos'][s, t]) and np.array_equal(np.concatenate(obs_n), g['obs'][s, t])
    a = g['act'][:, :, :5]
    assert ((a == a.max(-1, keepdims=True)).sum(-1) > 1).any()          # ties in a follower's physical entries


def test_batched_oracle_equals_serial_on_states_with_coincident_entities(golden_dir):
    """Two colliding entities at the same point get a NaN force (core.py:239-243: delta / 0) and keep NaN positions; from
    then on the scenario gathers run on NaNs: Python's min() over a list keeps a NaN that comes first and skips a later
    one, comparisons are false, sums spread it.  The reference does nothing about it, so neither do we: the vectorised
    oracle (the GPU tests' yardstick) must follow the serial restatement there too, bit for bit, NaN for NaN.  Entities
    on a grid of eighths with sizes on the same grid: coincidences in a third of the envs, contacts exactly at
    dist == dist_min."""
    specs = {'tag': O.simple_tag_spec(), 'spread': O.simple_spread_spec(), 'push': O.simple_push_spec(),
             'wcomm': O.simple_world_comm_spec(), 'adversary': O.simple_adversary_spec()}
    for b in ('h_S1L2F2_10Targets_wide', 'h_L1F2_3Targets_100observe', 'm_S1L2F4_MovingTarget'):
        specs[b] = O.hier_spec_from_golden(_load(golden_dir, b))
    rng = np.random.RandomState(5)
    B, T = 64, 2
    seen_nan = 0
    for name, spec in specs.items():
        A, E, L = spec.n_agents, spec.n_entities, spec.n_landmarks
        n = 2 if E <= 8 else 3
        pos = rng.randint(-n, n + 1, (E, 2, B)) * 0.125
        vel = rng.randint(-6, 7, (A, 2, B)) * 0.125
        spec = dataclasses.replace(spec, size=[float(rng.randint(1, 4)) / 16 for _ in range(E)])
        goal = rng.randint(0, max(L, 1), B)
        acts = [rng.rand(sum(spec.act_dims), B) for _ in range(T)]
        off = np.cumsum([0] + list(spec.act_dims))
        bo = O.BatchedOracle(spec)
        p, v, bres = pos, vel, []
        with np.errstate(all='ignore'):
            for t in range(T):
                res = bo.env_step(p, v, acts[t], None, goal=goal if spec.goal_rows else None)
                p, v = res[0], res[1]
                bres.append(res)
            for b in range(B):
                env = O.SerialEnv(spec, benchmark=True)
                env.set_state(pos[:, :, b], vel[:, :, b])
                env.goal = int(goal[b])
                for t in range(T):
                    obs_n, rew_n, _, info_n = env.step([acts[t][off[i]:off[i + 1], b] for i in range(A)])
                    P, V = env.get_state()
                    bp, bv, bobs, brew, binfo = bres[t][:5]
                    assert np.array_equal(P, bp[:, :, b], equal_nan=True) and np.array_equal(V, bv[:, :, b], equal_nan=True)
                    assert np.array_equal(np.concatenate(obs_n), np.concatenate([o[:, b] for o in bobs]), equal_nan=True), name
                    assert np.array_equal(np.asarray(rew_n, float), brew[:, b], equal_nan=True), (name, b, t)
                    if spec.kind in (O.SIMPLE_TAG, O.SIMPLE_SPREAD, O.WORLD_COMM, O.HIER_PURSUIT, O.HIER_MOVING):
                        flat = np.concatenate([np.ravel(np.asarray(x, float)) for x in info_n['n']])
                        assert np.array_equal(flat, np.asarray(binfo)[:, :, b].ravel(), equal_nan=True), (name, b, t)
                seen_nan += bool(np.isnan(P).any())
    assert seen_nan > 100
