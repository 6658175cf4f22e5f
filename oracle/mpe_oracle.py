"""CPU oracle for the multiagent-particle-envs hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a from-spec numpy restatement of the reference's algorithm for
``MultiAgentEnv.step`` -> ``World.step`` -> scenario observation/reward/info
(reference = /root/reference/multiagent-particle-envs, live ``multiagent/core.py``
and ``multiagent/environment.py``; scenarios from ``build/lib/multiagent/scenarios``).
It is pinned bit-exactly against golden vectors produced by the reference itself
(``tests/golden/gen_golden.py`` -> ``tests/golden/*.npz``; see
``tests/test_oracle_golden.py``).

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of
``bench.py`` may import this module, and only as the checker / reported CPU
baseline.  The product (``marl_amd``) never imports it and has no CPU fallback.

Two restatements live here:

* ``SerialEnv``   one env, one Python object per entity, tiny (2,) float64 arrays,
                  same evaluation order as the reference (this is the "serial numpy
                  loop" timed as the CPU baseline).
* ``BatchedOracle``  the same arithmetic, same per-element operation order,
                  vectorised over a batch axis with SoA arrays ``pos[E,2,B]``,
                  ``vel[A,2,B]`` (usable at 10^5..10^6 envs, float64 or float32).

Plus ``philox4x32_10`` / ``reset_positions``: the restatement of this build's own
device reset RNG (no reference counterpart: the reference draws from the global
``np.random`` stream, simple_tag.py:47-54).
"""
from __future__ import annotations

import dataclasses
from typing import List, Optional, Sequence

import numpy as np

SIMPLE_TAG = 0
SIMPLE_SPREAD = 1
SIMPLE = 2             # build/lib/multiagent/scenarios/simple.py            
SIMPLE_ADVERSARY = 3   # build/lib/multiagent/scenarios/simple_adversary.py
SIMPLE_PUSH = 4        # build/lib/multiagent/scenarios/simple_push.py
HIER_PURSUIT = 5       # multiagent/scenarios/{L1F*,S1L*}.py: the fork's leader/follower pursuit
SPEAKER_LISTENER = 6   # build/lib/multiagent/scenarios/simple_speaker_listener.py  (comm half)
SIMPLE_REFERENCE = 7   # build/lib/multiagent/scenarios/simple_reference.py
SIMPLE_CRYPTO = 8      # build/lib/multiagent/scenarios/simple_crypto.py (nobody moves: messages only)
WORLD_COMM = 9         # build/lib/multiagent/scenarios/simple_world_comm.py (tag physics, food, forests, a speaking leader)
HIER_MOVING = 10       # multiagent/scenarios/{L1F4,S1L2F4}_MovingTarget.py: the command hierarchy chasing a moving prey
COMM_KINDS = (SPEAKER_LISTENER, SIMPLE_REFERENCE)   # per-agent goal landmark (agent.goal_b) + observed comm
# landmark.color tables (simple_speaker_listener.py:48-50 / simple_reference.py:41-43): (base, highlighted)
COMM_COLORS = {SPEAKER_LISTENER: (0.15, 0.65), SIMPLE_REFERENCE: (0.25, 0.75)}
GOAL_KINDS = (SIMPLE_ADVERSARY, SIMPLE_PUSH)   # scenarios with a per-env goal landmark (agent.goal_a)


# --------------------------------------------------------------------------------------
# Scenario constants (what make_world() sets up)
# --------------------------------------------------------------------------------------
@dataclasses.dataclass
class ScenarioSpec:
    kind: int
    n_agents: int
    n_landmarks: int
    adversary: Sequence[bool]            # per agent
    size: Sequence[float]                # per entity (agents then landmarks, core.py:109-111)
    accel: Sequence[Optional[float]]     # per agent; None -> sensitivity 5.0 (environment.py:193-195)
    max_speed: Sequence[Optional[float]]  # per agent; None -> no clamp (core.py:217)
    collide: Sequence[bool]              # per entity
    movable: Sequence[bool]              # per entity
    u_noise: Sequence[Optional[float]]   # per agent (core.py:73,147)
    mass: Sequence[float]                # per entity (core.py:47,51-53 -> 1.0)
    agent_pos_range: float = 1.0         # reset: U(-r, r) (simple_tag.py:48, simple_spread.py:39)
    landmark_pos_range: float = 0.9      # simple_tag.py:53 (0.9), simple_spread.py:43 (1.0)
    dim_c: int = 2
    dt: float = 0.1                      # core.py:100
    damping: float = 0.25                # core.py:102
    contact_force: float = 1e2           # core.py:104
    contact_margin: float = 1e-3         # core.py:105
    # fork additions (per agent; defaults = stock scenarios): core.py:153-194, L1F2_3Targets_100observe.py:36-86
    silent: Optional[Sequence[bool]] = None          # non-silent agents carry dim_c comm action entries
    force_id: Optional[Sequence[int]] = None         # leader gives commands to agents with forced_id == force_id
    forced_id: Optional[Sequence[int]] = None
    advleader: Optional[Sequence[bool]] = None
    advsuperleader: Optional[Sequence[bool]] = None
    obs_r: Optional[Sequence[float]] = None          # observation radius
    reward_variant: int = 0                          # 1 = S1L2F2_10Targets_wide_rewardFixed.py
    advfollower: Optional[Sequence[bool]] = None     # agents whose touch respawns a target (train_and_eval.py:216)
    n_target_obs: int = 3                            # target_pos[0:3]
    # simple_world_comm.py:19-57: world.landmarks = plain landmarks + food + forests, in that order
    n_food: int = 0
    n_forests: int = 0
    leader: Optional[Sequence[bool]] = None          # agent.leader (simple_world_comm.py:27): sees through forests
    c_noise: Optional[Sequence[Optional[float]]] = None   # per agent (core.py:74,229); None -> no comm noise
    # movable landmarks (no scenario file has one; core.py:211-222 integrates EVERY movable entity): max_speed per
    # LANDMARK (Entity.max_speed, core.py:42); movable / mass / collide come from the per-entity lists above
    landmark_max_speed: Optional[Sequence[Optional[float]]] = None

    def __post_init__(self):
        A = self.n_agents
        if self.silent is None:
            self.silent = [True] * A
        if self.force_id is None:
            self.force_id = [-1] * A
        if self.forced_id is None:
            self.forced_id = [-1] * A
        if self.advleader is None:
            self.advleader = [False] * A
        if self.advsuperleader is None:
            self.advsuperleader = [False] * A
        if self.obs_r is None:
            self.obs_r = [0.0] * A
        if self.leader is None:
            self.leader = [False] * A
        if self.c_noise is None:
            self.c_noise = [None] * A
        if self.landmark_max_speed is None:
            self.landmark_max_speed = [None] * self.n_landmarks
        if self.advfollower is None:                 # L1F2_3Targets_100observe.py:40,52,64: the silent adversaries
            self.advfollower = [bool(self.adversary[i]) and not self.advleader[i] and self.kind in (HIER_PURSUIT, HIER_MOVING)
                                for i in range(A)]
        self.check_no_alias()

    def check_no_alias(self):
        """core.py:163 aliases the super-leader's action.c[0:2]; a plain leader with the same force_id would mutate it
        in place (core.py:192).  Not restated (no reference scenario does it): refused, like mpe_desc_check."""
        for a in range(self.n_agents):
            for b in range(self.n_agents):
                if (self.force_id[a] >= 0 and self.force_id[a] == self.force_id[b]
                        and self.advsuperleader[a] and not self.advsuperleader[b]):
                    raise ValueError('a super-leader and a plain leader share force_id %d (core.py:163,192 aliasing)'
                                     % self.force_id[a])

    def act_dim(self, i):
        """environment.py:42-76: Discrete(5) for silent agents, MultiDiscrete([0,4],[0,dim_c-1]) otherwise."""
        return (5 if self.movable[i] else 0) + (0 if self.silent[i] else self.dim_c)   # environment.py:51-63

    def comm_row0(self, i):
        """First comm entry inside agent i's action (after the physical part, if it has one)."""
        return 5 if self.movable[i] else 0

    def observes_leader_force(self, i):
        """HIER_MOVING: does agent i's observation end with its leader_force?  L1F4_MovingTarget.py:252-258
        (variant 0): forced_id >= 0.  S1L2F4_MovingTarget.py:254-260 (variant 1): every agent has the
        attribute (core.py:156), and the ones that are neither advleader nor good return it."""
        if self.reward_variant == 0:
            return self.forced_id[i] >= 0
        return not (self.advleader[i] or not self.adversary[i])

    @property
    def goal_rows(self):
        """Rows of the per-env goal index buffer: agent.goal_a of simple_adversary / simple_push (1),
        agent 0's goal_b of simple_speaker_listener (1), every agent's goal_b of simple_reference (A)."""
        if self.kind in GOAL_KINDS or self.kind == SPEAKER_LISTENER:
            return 1
        if self.kind == SIMPLE_CRYPTO:          # (goal_a landmark, landmark whose colour is the speaker's key)
            return 2
        return self.n_agents if self.kind == SIMPLE_REFERENCE else 0

    @property
    def act_dims(self):
        return [self.act_dim(i) for i in range(self.n_agents)]

    @property
    def has_leader_force(self):
        return any(f >= 0 for f in self.forced_id)

    @property
    def n_entities(self):
        return self.n_agents + self.n_landmarks

    def obs_dim(self, i):
        A, L = self.n_agents, self.n_landmarks
        if self.kind == SIMPLE_TAG:
            n_good_others = sum(1 for j in range(A) if j != i and not self.adversary[j])
            return 4 + 2 * L + 2 * (A - 1) + 2 * n_good_others
        if self.kind == HIER_PURSUIT:           # L1F2_3Targets_100observe.py:250-253
            return 4 + 4 * (A - 1) + (2 if self.forced_id[i] >= 0 else 0) + 2 * min(L, self.n_target_obs)
        if self.kind == HIER_MOVING:            # L1F4_MovingTarget.py:252-258 / S1L2F4_MovingTarget.py:254-260
            return 4 + 4 * (A - 1) + (2 if self.observes_leader_force(i) else 0)
        if self.kind == SPEAKER_LISTENER:       # simple_speaker_listener.py:89-94
            return 3 if not self.movable[i] else 2 + 2 * L + self.dim_c * (A - 1)
        if self.kind == SIMPLE_REFERENCE:       # simple_reference.py:89
            return 2 + 2 * L + 3 + self.dim_c * (A - 1)
        if self.kind == SIMPLE_CRYPTO:          # simple_crypto.py:153-171: adversary / listener / speaker
            return self.dim_c if self.adversary[i] else 2 * self.dim_c
        if self.kind == WORLD_COMM:             # simple_world_comm.py:305-313
            n_good_others = sum(1 for j in range(A) if j != i and not self.adversary[j])
            return 4 + 2 * L + 2 * (A - 1) + 2 * n_good_others + 2 + (self.dim_c if self.adversary[i] else 0)
        if self.kind == SIMPLE:                 # simple.py:50
            return 2 + 2 * L
        if self.kind == SIMPLE_ADVERSARY:       # simple_adversary.py:137-140
            return 2 * L + 2 * (A - 1) + (0 if self.adversary[i] else 2)
        if self.kind == SIMPLE_PUSH:            # simple_push.py:100-104
            return 2 + 2 * L + 2 * (A - 1) + (0 if self.adversary[i] else 2 + 3 + 3 * L)
        return 4 + 2 * L + 2 * (A - 1) + self.dim_c * (A - 1)

    @property
    def obs_dims(self):
        return [self.obs_dim(i) for i in range(self.n_agents)]

    @property
    def info_width(self):
        if self.kind == SIMPLE_ADVERSARY:       # simple_adversary.py:58-67 (tuple of L+1 for good agents)
            return self.n_landmarks + 1
        if self.kind == SIMPLE_CRYPTO:          # (agent.state.c, agent.goal_a.color), simple_crypto.py:78-80
            return 2 * self.dim_c
        if self.kind in (SIMPLE, SIMPLE_PUSH) + COMM_KINDS:  # no (working) benchmark_data in these scenarios
            return 0
        return 1 if self.kind in (SIMPLE_TAG, HIER_PURSUIT, WORLD_COMM, HIER_MOVING) else 4


def simple_tag_spec(n_adv=3, n_good=1, n_landmarks=2, u_noise=None):
    """simple_tag.py:7-36 (per-role constants); scaled variants keep the same roles."""
    A = n_adv + n_good
    adv = [i < n_adv for i in range(A)]
    return ScenarioSpec(
        kind=SIMPLE_TAG, n_agents=A, n_landmarks=n_landmarks, adversary=adv,
        size=[0.075 if a else 0.05 for a in adv] + [0.2] * n_landmarks,
        accel=[3.0 if a else 4.0 for a in adv],
        max_speed=[1.0 if a else 1.3 for a in adv],
        collide=[True] * (A + n_landmarks),
        movable=[True] * A + [False] * n_landmarks,
        u_noise=[u_noise] * A, mass=[1.0] * (A + n_landmarks),
        agent_pos_range=1.0, landmark_pos_range=0.9)


def simple_spread_spec(n_agents=3, n_landmarks=3, u_noise=None):
    """simple_spread.py:7-28."""
    A = n_agents
    return ScenarioSpec(
        kind=SIMPLE_SPREAD, n_agents=A, n_landmarks=n_landmarks, adversary=[False] * A,
        size=[0.15] * A + [0.05] * n_landmarks,
        accel=[None] * A, max_speed=[None] * A,
        collide=[True] * A + [False] * n_landmarks,
        movable=[True] * A + [False] * n_landmarks,
        u_noise=[u_noise] * A, mass=[1.0] * (A + n_landmarks),
        agent_pos_range=1.0, landmark_pos_range=1.0)


def hier_spec_from_golden(g):
    """ScenarioSpec of an L1F*/S1L* scenario from the world parameters the golden generator read back
    from the reference's make_world() (tests/golden/gen_golden.py:rollout_hier)."""
    A = int(g['spec_n_agents'])
    E = len(g['spec_size'])
    moving = 'spec_moving_target' in g.files and bool(g['spec_moving_target'])
    return ScenarioSpec(
        kind=HIER_MOVING if moving else HIER_PURSUIT,
        n_agents=A, n_landmarks=E - A, adversary=[bool(x) for x in g['spec_adversary']],
        size=[float(x) for x in g['spec_size']], accel=[float(x) for x in g['spec_accel']],
        max_speed=[float(x) for x in g['spec_max_speed']], collide=[bool(x) for x in g['spec_collide']],
        movable=[bool(x) for x in g['spec_movable']], u_noise=[None] * A, mass=[1.0] * E,
        agent_pos_range=1.0, landmark_pos_range=0.9 if moving else 1.0,   # L1F4_MovingTarget.py:124
        dim_c=int(g['spec_dim_c']),
        silent=[bool(x) for x in g['spec_silent']], force_id=[int(x) for x in g['spec_force_id']],
        forced_id=[int(x) for x in g['spec_forced_id']], advleader=[bool(x) for x in g['spec_advleader']],
        advsuperleader=[bool(x) for x in g['spec_advsuperleader']], obs_r=[float(x) for x in g['spec_obs_r']],
        reward_variant=int(g['spec_reward_variant']))


def speaker_listener_spec():
    """simple_speaker_listener.py:10-34: agent 0 = speaker (fixed, speaks), agent 1 = listener (silent)."""
    return ScenarioSpec(
        kind=SPEAKER_LISTENER, n_agents=2, n_landmarks=3, adversary=[False, False],
        size=[0.075, 0.075, 0.04, 0.04, 0.04], accel=[None, None], max_speed=[None, None],
        collide=[False] * 5, movable=[False, True, False, False, False], u_noise=[None, None], mass=[1.0] * 5,
        agent_pos_range=1.0, landmark_pos_range=1.0, dim_c=3, silent=[False, True])


def simple_reference_spec():
    """simple_reference.py:6-25: two movable agents that both speak (dim_c = 10)."""
    return ScenarioSpec(
        kind=SIMPLE_REFERENCE, n_agents=2, n_landmarks=3, adversary=[False, False],
        size=[0.05] * 5, accel=[None, None], max_speed=[None, None],
        collide=[False] * 5, movable=[True, True, False, False, False], u_noise=[None, None], mass=[1.0] * 5,
        agent_pos_range=1.0, landmark_pos_range=1.0, dim_c=10, silent=[False, False])


def simple_crypto_spec():
    """simple_crypto.py:21-44: adversary (Eve), listener (Bob), speaker (Alice); nobody moves, all speak."""
    return ScenarioSpec(
        kind=SIMPLE_CRYPTO, n_agents=3, n_landmarks=2, adversary=[True, False, False],
        size=[0.05] * 5, accel=[None] * 3, max_speed=[None] * 3, collide=[False] * 5, movable=[False] * 5,
        u_noise=[None] * 3, mass=[1.0] * 5, agent_pos_range=1.0, landmark_pos_range=1.0, dim_c=4,
        silent=[False] * 3)


def simple_world_comm_spec(n_adv=4, n_good=2, n_plain=1, n_food=2, n_forests=2):
    """simple_world_comm.py:11-61: simple_tag roles and physics; agent 0 is the leader and the only speaker;
    landmarks = 1 colliding obstacle + 2 food + 2 forests (food and forests do not collide)."""
    A, L = n_adv + n_good, n_plain + n_food + n_forests
    adv = [i < n_adv for i in range(A)]
    return ScenarioSpec(
        kind=WORLD_COMM, n_agents=A, n_landmarks=L, adversary=adv,
        size=[0.075 if a else 0.045 for a in adv] + [0.2] * n_plain + [0.03] * n_food + [0.3] * n_forests,
        accel=[3.0 if a else 4.0 for a in adv], max_speed=[1.0 if a else 1.3 for a in adv],
        collide=[True] * A + [True] * n_plain + [False] * (n_food + n_forests),
        movable=[True] * A + [False] * L, u_noise=[None] * A, mass=[1.0] * (A + L),
        agent_pos_range=1.0, landmark_pos_range=0.9, dim_c=4, silent=[i > 0 for i in range(A)],
        n_food=n_food, n_forests=n_forests, leader=[i == 0 for i in range(A)])


def simple_spec():
    """simple.py:6-22 (World defaults: dim_c = 0, size 0.05, nothing collides)."""
    return ScenarioSpec(
        kind=SIMPLE, n_agents=1, n_landmarks=1, adversary=[False], size=[0.05, 0.05],
        accel=[None], max_speed=[None], collide=[False, False], movable=[True, False],
        u_noise=[None], mass=[1.0, 1.0], agent_pos_range=1.0, landmark_pos_range=1.0, dim_c=0)


def simple_adversary_spec(n_agents=3):
    """simple_adversary.py:9-34: 1 adversary first, N-1 landmarks, nothing collides."""
    A, L = n_agents, n_agents - 1
    return ScenarioSpec(
        kind=SIMPLE_ADVERSARY, n_agents=A, n_landmarks=L, adversary=[i < 1 for i in range(A)],
        size=[0.15] * A + [0.08] * L, accel=[None] * A, max_speed=[None] * A,
        collide=[False] * (A + L), movable=[True] * A + [False] * L,
        u_noise=[None] * A, mass=[1.0] * (A + L), agent_pos_range=1.0, landmark_pos_range=1.0)


def simple_push_spec(n_agents=10, n_adv=9, n_landmarks=1):
    """simple_push.py:12-39 (this fork's copy: 10 agents, 9 adversaries first, 1 landmark)."""
    A, L = n_agents, n_landmarks
    return ScenarioSpec(
        kind=SIMPLE_PUSH, n_agents=A, n_landmarks=L, adversary=[i < n_adv for i in range(A)],
        size=[0.05] * (A + L), accel=[None] * A, max_speed=[None] * A,
        collide=[True] * A + [False] * L, movable=[True] * A + [False] * L,
        u_noise=[None] * A, mass=[1.0] * (A + L), agent_pos_range=1.0, landmark_pos_range=1.0)


# --------------------------------------------------------------------------------------
# Serial restatement: one Python object per entity (the CPU baseline shape)
# --------------------------------------------------------------------------------------
class _Body:
    __slots__ = ('is_agent', 'adversary', 'size', 'accel', 'max_speed', 'collide', 'movable',
                 'u_noise', 'mass', 'pos', 'vel', 'comm', 'u', 'c_act', 'silent', 'force_id', 'forced_id',
                 'advleader', 'advsuperleader', 'obs_r', 'leader_force', 'c_noise')


class SerialEnv:
    """One environment instance; mirrors environment.py:92-129 + core.py:124-249."""

    def __init__(self, spec: ScenarioSpec, benchmark=True):
        self.spec = spec
        self.benchmark = benchmark
        self.bodies: List[_Body] = []
        for e in range(spec.n_entities):
            b = _Body()
            b.is_agent = e < spec.n_agents
            b.adversary = bool(spec.adversary[e]) if b.is_agent else False
            b.size = spec.size[e]
            b.accel = spec.accel[e] if b.is_agent else None
            b.max_speed = spec.max_speed[e] if b.is_agent else spec.landmark_max_speed[e - spec.n_agents]
            b.collide = bool(spec.collide[e])
            b.movable = bool(spec.movable[e])
            b.u_noise = spec.u_noise[e] if b.is_agent else None
            b.c_noise = spec.c_noise[e] if b.is_agent else None
            b.mass = spec.mass[e]
            b.pos = np.zeros(2)
            b.vel = np.zeros(2)
            b.comm = np.zeros(spec.dim_c)
            b.u = np.zeros(2)
            b.c_act = np.zeros(spec.dim_c)
            b.silent = bool(spec.silent[e]) if b.is_agent else True
            b.force_id = spec.force_id[e] if b.is_agent else -1
            b.forced_id = spec.forced_id[e] if b.is_agent else -1
            b.advleader = bool(spec.advleader[e]) if b.is_agent else False
            b.advsuperleader = bool(spec.advsuperleader[e]) if b.is_agent else False
            b.obs_r = spec.obs_r[e] if b.is_agent else 0.0
            b.leader_force = np.zeros(2)       # set in make_world, NOT touched by reset_world
            self.bodies.append(b)
        self.agents = self.bodies[:spec.n_agents]
        self.landmarks = self.bodies[spec.n_agents:]
        self.n = spec.n_agents
        self.goal = 0        # index of the goal landmark (agent.goal_a), GOAL_KINDS only
        self.goals = [0] * spec.n_agents   # agent.goal_b landmark index, COMM_KINDS only
        # core.py:83,116-127: agents with an action_callback are scripted: World.step asks the callback for their action
        # (its u is the force as it stands, no sensitivity) and the env's n / action_n / obs_n / reward_n / info_n cover the
        # others only.  {agent index: callback(env, index) -> u (2-vector)}
        self.scripted = {}
        self.discrete_action_input = False   # environment.py:31: the action is an integer
        self.force_discrete_action = False   # environment.py:33: argmax -> one-hot before the decode
        self.discrete_action_space = True    # environment.py:29; False: the action IS the force command (:191-192)

    # -- state helpers ----------------------------------------------------------------
    def set_state(self, pos, vel, lvel=None):
        for e, b in enumerate(self.bodies):
            b.pos = np.array(pos[e], dtype=np.float64)
        for i, a in enumerate(self.agents):
            a.vel = np.array(vel[i], dtype=np.float64)
        if lvel is not None:             # movable landmarks carry a velocity too (core.py:211-222)
            for i, lm in enumerate(self.landmarks):
                lm.vel = np.array(lvel[i], dtype=np.float64)

    def get_state(self):
        return (np.stack([b.pos for b in self.bodies]), np.stack([a.vel for a in self.agents]))

    # -- reset: simple_tag.py:39-54 / simple_spread.py:30-44 (global np.random stream) --
    def reset(self):
        s = self.spec
        if s.kind in GOAL_KINDS:   # goal = np.random.choice(world.landmarks) (simple_adversary.py:45)
            self.goal = int(np.random.choice(len(self.landmarks)))
        if s.kind == SIMPLE_CRYPTO:      # simple_crypto.py:61,63: goal, then the key
            self.goals = [int(np.random.choice(len(self.landmarks))), int(np.random.choice(len(self.landmarks))), 0]
        if s.kind == SPEAKER_LISTENER:   # simple_speaker_listener.py:43
            self.goals[0] = int(np.random.choice(len(self.landmarks)))
        if s.kind == SIMPLE_REFERENCE:   # simple_reference.py:34,36
            self.goals = [int(np.random.choice(len(self.landmarks))) for _ in self.agents]
        for a in self.agents:
            a.pos = np.random.uniform(-s.agent_pos_range, +s.agent_pos_range, 2)
            a.vel = np.zeros(2)
            a.comm = np.zeros(s.dim_c)
        for lm in self.landmarks:
            lm.pos = np.random.uniform(-s.landmark_pos_range, +s.landmark_pos_range, 2)
            lm.vel = np.zeros(2)
        if s.kind == WORLD_COMM:         # simple_world_comm.py:112-117: food, then forests, are drawn a second time
            for lm in self.landmarks[s.n_landmarks - s.n_forests - s.n_food:]:
                lm.pos = np.random.uniform(-s.landmark_pos_range, +s.landmark_pos_range, 2)
        return [self.observation(i) for i in range(self.n)]

    # -- environment.py:157-208 (Discrete(5) soft one-hot path) --------------------------
    def _decode_action(self, agent, action):
        u = np.zeros(2)
        c0 = 0
        if self.discrete_action_input:    # environment.py:176-182; a speaking agent fails the assert at :208
            assert agent.silent, 'environment.py:208: assert len(action) == 0'
            if agent.movable:
                if action == 1: u[0] = -1.0
                if action == 2: u[0] = +1.0
                if action == 3: u[1] = -1.0
                if action == 4: u[1] = +1.0
                u *= 5.0 if agent.accel is None else agent.accel
            agent.u = u
            agent.c_act = np.zeros(self.spec.dim_c)
            return
        if agent.movable and not self.discrete_action_space:
            # environment.py:184-187,191-196 with the Discrete space built at construction: `action = [action]`, the
            # argmax (if forced) runs on that vector, `agent.action.u = action[0]` (the caller's array), `u *= sensitivity`
            assert agent.silent, 'a MultiDiscrete action would make u its first 5 entries (core.py:216 then fails)'
            action = np.asarray(action, dtype=np.float64)
            if self.force_discrete_action:
                d = np.argmax(action)
                action[:] = 0.0
                action[d] = 1.0
            agent.u = action * (5.0 if agent.accel is None else agent.accel)
            agent.c_act = np.zeros(self.spec.dim_c)
            return
        if agent.movable:                 # environment.py:174-197: only a movable agent has the physical part
            if self.force_discrete_action:    # environment.py:184-187: in place, on the caller's array
                d = np.argmax(action[0:5])
                action[0:5] = 0.0
                action[d] = 1.0
            u[0] += action[1] - action[2]
            u[1] += action[3] - action[4]
            u *= 5.0 if agent.accel is None else agent.accel
            c0 = 5
        agent.u = u
        # environment.py:161-168,198-206: a non-silent agent's action carries dim_c comm entries
        agent.c_act = np.zeros(self.spec.dim_c) if agent.silent else np.asarray(action[c0:c0 + self.spec.dim_c])

    # -- core.py:153-194 (fork): leaders push the agents they command ------------------------------------
    def _apply_leader_force(self, force):
        for i, agent in enumerate(self.agents):
            agent.leader_force = np.zeros(2)
            if agent.forced_id < 0:
                continue
            for leader in self.agents:
                if leader.force_id < 0 or agent.forced_id != leader.force_id:
                    continue
                if leader.advsuperleader:          # observed only; NOT added to the physical force
                    # The reference assigns the VIEW leader.action.c[0:2] (core.py:163): a plain leader with the same
                    # force_id would then add its push into the super-leader's own action.c (core.py:192).  No
                    # reference scenario shares a force_id that way; ScenarioSpec refuses it (check_no_alias) and so
                    # does mpe_desc_check (MPE_ERR_UNSUPPORTED), so a copy is equivalent here.
                    agent.leader_force = np.array(leader.c_act[0:2], dtype=np.float64)
                else:
                    cmd = int(np.argmax(np.array(leader.c_act)))
                    if len(leader.c_act) == 3:
                        cmd = [0, 5, 6][cmd]
                    f = np.zeros(2)
                    if cmd == 1:
                        f = np.array([-1, 0])
                    elif cmd == 2:
                        f = np.array([1, 0])
                    elif cmd == 3:
                        f = np.array([0, -1])
                    elif cmd == 4:
                        f = np.array([0, 1])
                    elif cmd == 5:
                        f = leader.pos - agent.pos
                        f = f / np.linalg.norm(f)
                    elif cmd == 6:
                        f = agent.pos - leader.pos
                        f = f / np.linalg.norm(f)
                    f = f * 1
                    force[i] = force[i] + f
                    agent.leader_force = agent.leader_force + f
        return force

    # -- core.py:124-140 ----------------------------------------------------------------
    def world_step(self):
        s = self.spec
        force = [None] * len(self.bodies)
        # core.py:143-149
        for i, a in enumerate(self.agents):
            if a.movable:
                nz = np.random.randn(2) * a.u_noise if a.u_noise else 0.0
                force[i] = a.u + nz
        # core.py:153-194: a no-op for forced_id < 0 (every stock scenario)
        if s.has_leader_force:
            force = self._apply_leader_force(force)
        # core.py:197-208 (one-sided: only the lower-indexed entity of a pair is pushed)
        for ia, ea in enumerate(self.bodies):
            for ib in range(ia + 1, len(self.bodies)):
                eb = self.bodies[ib]
                if not (ea.collide and eb.collide):
                    continue
                d = ea.pos - eb.pos
                dist = np.sqrt(np.sum(np.square(d)))
                dmin = ea.size + eb.size
                k = s.contact_margin
                pen = np.logaddexp(0, -(dist - dmin) / k) * k
                f = s.contact_force * d / dist * pen
                if ea.movable:
                    if force[ia] is None:
                        force[ia] = 0.0
                    force[ia] = f + force[ia]
                if eb.movable and force[ib] is None:
                    force[ib] = 0.0
        # core.py:211-222
        for i, e in enumerate(self.bodies):
            if not e.movable:
                continue
            e.vel = e.vel * (1 - s.damping)
            if force[i] is not None:
                e.vel = e.vel + (force[i] / e.mass) * s.dt
            if e.max_speed is not None:
                speed = np.sqrt(np.square(e.vel[0]) + np.square(e.vel[1]))
                if speed > e.max_speed:
                    e.vel = e.vel / np.sqrt(np.square(e.vel[0]) + np.square(e.vel[1])) * e.max_speed
            e.pos = e.pos + e.vel * s.dt
        # core.py:224-230 (all agents silent)
        for a in self.agents:
            if a.silent:
                a.comm = np.zeros(s.dim_c)
            else:
                noise = np.random.randn(*a.c_act.shape) * a.c_noise if a.c_noise else 0.0
                a.comm = a.c_act + noise

    # -- L1F*/S1L* gathers (e.g. L1F2_3Targets_100observe.py:136-253) ----------------------------------------
    def _hier_observation(self, i):
        me = self.agents[i]
        sq_rad = me.obs_r * me.obs_r
        other_pos, other_vel, seen, zeros = [], [], [], []
        for other in self.agents:
            if other is me:
                continue
            sq = np.sum(np.square(other.pos - me.pos))
            if (sq < sq_rad or (me.force_id >= 0 and me.force_id == other.forced_id)
                    or (me.forced_id >= 0 and me.forced_id == other.force_id)):
                other_pos.append(other.pos - me.pos)
                other_vel.append(other.vel)
            else:
                other_pos.append(np.array([0, 0]))
                other_vel.append(np.array([0, 0]))
        for lm in self.landmarks:
            sq = np.sum(np.square(lm.pos - me.pos))
            if sq < sq_rad:
                seen.append((sq, lm.pos - me.pos))
            else:
                zeros.append(np.array([0, 0]))
        seen.sort(key=lambda x: x[0])
        targets = ([p for _, p in seen] + zeros)[0:self.spec.n_target_obs]
        comm = [me.leader_force] if me.forced_id >= 0 else []
        return np.concatenate([me.vel] + [me.pos] + other_pos + other_vel + comm + targets)

    def _hier_reward(self, i):
        me, s = self.agents[i], self.spec
        r = 0
        if s.reward_variant == 0:               # L1F2_3Targets_100observe.py:167-181
            for lm in self.landmarks:
                if not me.advleader:
                    if self._touching(lm, me):
                        r += 10
                elif not me.advsuperleader:
                    for o in self.agents:
                        if o.forced_id == me.force_id and self._touching(lm, o):
                            r += 10
                else:
                    for o in self.agents:
                        if (not o.advleader) and self._touching(lm, o):
                            r += 10
        else:                                   # S1L2F2_10Targets_wide_rewardFixed.py:127-151
            if me.advsuperleader:
                for lm in self.landmarks:
                    for o in self.agents:       # is_observed compares the target with itself: dist = 0
                        if o.forced_id == me.force_id and 0.0 < o.obs_r:
                            r += 1
            elif me.advleader:
                for lm in self.landmarks:
                    for o in self.agents:
                        if o.forced_id == me.force_id and self._touching(lm, o):
                            r += 10
            else:
                for lm in self.landmarks:
                    if self._touching(lm, me):
                        r += 10
        for p in range(2):
            x = abs(me.pos[p])
            if x < 0.9:
                pen = 0
            elif x < 1.0:
                pen = (x - 0.9) * 10
            else:
                pen = min(np.exp(2 * x - 2), 10)
            r -= pen
        return r

    # -- *_MovingTarget gathers (L1F4_MovingTarget.py:157-258; S1L2F4_MovingTarget.py differs in :227,254,257) --
    def _moving_observation(self, i):
        me, s = self.agents[i], self.spec
        other_pos, other_vel = [], []
        for other in self.agents:
            if other is me:
                continue
            dis = np.sqrt(np.sum(np.square(other.pos - me.pos)))
            if me.adversary:
                linked = ((me.force_id >= 0 and me.force_id == other.forced_id)
                          or (me.forced_id >= 0 and me.forced_id == other.force_id))
                if dis < me.obs_r or (s.reward_variant == 0 and linked):
                    other_pos.append(other.pos - me.pos)
                    other_vel.append(other.vel)
                else:
                    other_pos.append(np.array([0, 0]))
                    other_vel.append(np.array([0, 0]))
            else:                                # the prey sees everybody, in ABSOLUTE coordinates (:249-251)
                other_pos.append(other.pos)
                other_vel.append(other.vel)
        comm = [me.leader_force] if s.observes_leader_force(i) else []
        return np.concatenate([me.vel] + [me.pos] + other_pos + other_vel + comm)

    def _moving_reward(self, i):
        me = self.agents[i]
        rew = 0
        if me.adversary:                         # :186-212: the same contact sum for every adversary
            count = 0
            for a in self.agents:
                if a.adversary:
                    continue
                for other in self.agents:
                    if other is a:
                        continue
                    if self._touching(a, other):
                        count += 1
                        rew += 10
                if count >= 4:
                    rew += 50
        elif me.collide:                         # :162-171
            for a in self.agents:
                if a.adversary and self._touching(a, me):
                    rew -= 1
        for p in range(2):
            x = abs(me.pos[p])
            if x < 0.9:
                pen = 0
            elif x < 1.0:
                pen = (x - 0.9) * 10
            else:
                pen = min(np.exp(2 * x - 2), 10)
            rew -= pen
        return rew

    def _hier_info(self, i):                    # L1F2_3Targets_100observe.py:136-148
        me = self.agents[i]
        return sum(1 for a in self.agents if a.adversary != me.adversary and self._touching(a, me))

    # -- scenario gathers -----------------------------------------------------------------
    @staticmethod
    def _touching(p, q):
        d = p.pos - q.pos
        return bool(np.sqrt(np.sum(np.square(d))) < p.size + q.size)

    def _sq(self, p, q):
        return np.sum(np.square(p.pos - q.pos))

    def _push_colors(self, me):
        """simple_push.py:43-56: landmark i -> 0.1 with [i+1] += 0.8; good agent -> 0.25 with [goal+1] += 0.5."""
        lm_colors = []
        for i in range(len(self.landmarks)):
            c = np.array([0.1, 0.1, 0.1])
            c[i + 1] += 0.8
            lm_colors.append(c)
        mine = np.array([0.25, 0.25, 0.25])
        mine[self.goal + 1] += 0.5
        return mine, lm_colors

    def observation(self, i):
        me = self.agents[i]
        kind = self.spec.kind
        if kind == HIER_PURSUIT:
            return self._hier_observation(i)
        if kind == HIER_MOVING:
            return self._moving_observation(i)
        if kind == SIMPLE_CRYPTO:                                # simple_crypto.py:124-171
            onehot = lambda k: np.eye(self.spec.dim_c)[k]        # landmark.color = one-hot (:55-59)
            goal_color, key = onehot(self.goals[0]), onehot(self.goals[1])
            comm = [self.agents[2].comm] if i != 2 else []       # only the speaker's (agent 2) message is observed
            if i == 2:
                return np.concatenate([goal_color] + [key])
            if not me.adversary:
                return np.concatenate([key] + comm)
            return np.concatenate(comm)
        if kind == WORLD_COMM:                                   # simple_world_comm.py:230-313
            s = self.spec
            forests = self.landmarks[s.n_landmarks - s.n_forests:]
            assert len(forests) == 2
            entity_pos = [lm.pos - me.pos for lm in self.landmarks]
            inf = [self._touching(me, f) for f in forests]
            in_forest = [np.array([1]) if x else np.array([-1]) for x in inf]
            other_pos, other_vel = [], []
            for other in self.agents:
                if other is me:
                    continue
                oth = [self._touching(other, f) for f in forests]
                if (inf[0] and oth[0]) or (inf[1] and oth[1]) or (not inf[0] and not oth[0] and not inf[1] and not oth[1]) \
                        or s.leader[i]:
                    other_pos.append(other.pos - me.pos)
                    if not other.adversary:
                        other_vel.append(other.vel)
                else:
                    other_pos.append([0, 0])
                    if not other.adversary:
                        other_vel.append([0, 0])
            comm = [self.agents[0].comm]
            if me.adversary:                                     # leader and the other adversaries alike (:305-310)
                return np.concatenate([me.vel] + [me.pos] + entity_pos + other_pos + other_vel + in_forest + comm)
            return np.concatenate([me.vel] + [me.pos] + entity_pos + other_pos + in_forest + other_vel)
        if kind in COMM_KINDS:
            base, hi = COMM_COLORS[kind]
            color = np.full(3, base)
            color[self.goals[i]] = hi                            # agent.goal_b.color
            rel_lm = [lm.pos - me.pos for lm in self.landmarks]
            comm = [o.comm for o in self.agents if o is not me]
            if kind == SIMPLE_REFERENCE:                         # simple_reference.py:62-89
                return np.concatenate([me.vel] + rel_lm + [color] + comm)
            if not me.movable:                                   # simple_speaker_listener.py:89-91
                return np.concatenate([color])
            return np.concatenate([me.vel] + rel_lm + comm)      # :93-94
        if kind in (SIMPLE, SIMPLE_ADVERSARY, SIMPLE_PUSH):
            goal = self.landmarks[self.goal]
            rel_lm = [lm.pos - me.pos for lm in self.landmarks]
            rel_ag = [o.pos - me.pos for o in self.agents if o is not me]
            if kind == SIMPLE:                                   # simple.py:45-50
                return np.concatenate([me.vel] + rel_lm)
            if kind == SIMPLE_ADVERSARY:                         # simple_adversary.py:122-140
                if not me.adversary:
                    return np.concatenate([goal.pos - me.pos] + rel_lm + rel_ag)
                return np.concatenate(rel_lm + rel_ag)
            if not me.adversary:                                 # simple_push.py:84-104
                mine, lm_colors = self._push_colors(me)
                return np.concatenate([me.vel] + [goal.pos - me.pos] + [mine] + rel_lm + lm_colors + rel_ag)
            return np.concatenate([me.vel] + rel_lm + rel_ag)
        rel_lm = [lm.pos - me.pos for lm in self.landmarks]
        rel_ag, tail = [], []
        for j, other in enumerate(self.agents):
            if other is me:
                continue
            rel_ag.append(other.pos - me.pos)
            if self.spec.kind == SIMPLE_TAG:
                if not other.adversary:
                    tail.append(other.vel)        # simple_tag.py:145-146
            else:
                tail.append(other.comm)           # simple_spread.py:97,99
        return np.concatenate([me.vel, me.pos] + rel_lm + rel_ag + tail)

    def reward(self, i):
        me = self.agents[i]
        kind = self.spec.kind
        if kind == HIER_PURSUIT:
            return self._hier_reward(i)
        if kind == HIER_MOVING:
            return self._moving_reward(i)
        if kind == SIMPLE_CRYPTO:                                # simple_crypto.py:94-121
            goal_color = np.eye(self.spec.dim_c)[self.goals[0]]
            zero = np.zeros(self.spec.dim_c)
            if me.adversary:
                rew = 0
                if not (me.comm == zero).all():
                    rew -= np.sum(np.square(me.comm - goal_color))
                return rew
            good_rew, adv_rew = 0, 0
            bob, eve = self.agents[1], self.agents[0]
            if not (bob.comm == zero).all():
                good_rew -= np.sum(np.square(bob.comm - goal_color))
            if not (eve.comm == zero).all():
                adv_rew += np.sum(np.square(eve.comm - goal_color))
            return adv_rew + good_rew
        if kind == WORLD_COMM:                                   # simple_world_comm.py:146-204
            s = self.spec
            advs = [a for a in self.agents if a.adversary]
            goods = [a for a in self.agents if not a.adversary]
            food = self.landmarks[s.n_landmarks - s.n_forests - s.n_food:s.n_landmarks - s.n_forests]
            rew = 0
            if me.adversary:
                rew -= 0.1 * min([np.sqrt(np.sum(np.square(a.pos - me.pos))) for a in goods])
                if me.collide:
                    for ag in goods:
                        for adv in advs:
                            if self._touching(ag, adv):
                                rew += 5
                return rew
            if me.collide:
                for a in advs:
                    if self._touching(a, me):
                        rew -= 5
            for p in range(2):
                x = abs(me.pos[p])
                if x < 0.9:
                    pen = 0
                elif x < 1.0:
                    pen = (x - 0.9) * 10
                else:
                    pen = min(np.exp(2 * x - 2), 10)
                rew -= 2 * pen
            for f in food:
                if self._touching(me, f):
                    rew += 2
            rew += 0.05 * min([np.sqrt(np.sum(np.square(f.pos - me.pos))) for f in food])
            return rew
        if kind == SPEAKER_LISTENER:                             # simple_speaker_listener.py:66-70
            return -self._sq(self.agents[1], self.landmarks[self.goals[0]])
        if kind == SIMPLE_REFERENCE:                             # simple_reference.py:56-60 (goal_a = the other agent)
            return -self._sq(self.agents[1 - i], self.landmarks[self.goals[i]])
        if kind == SIMPLE:                                       # simple.py:41-43
            return -self._sq(me, self.landmarks[0])
        if kind in GOAL_KINDS:
            goal = self.landmarks[self.goal]
            goods = [a for a in self.agents if not a.adversary]
            advs = [a for a in self.agents if a.adversary]
            if kind == SIMPLE_ADVERSARY:                         # simple_adversary.py:77-119
                if me.adversary:
                    return -self._sq(me, goal)
                adv_rew = sum([np.sqrt(self._sq(a, goal)) for a in advs])
                pos_rew = -min([np.sqrt(self._sq(a, goal)) for a in goods])
                return pos_rew + adv_rew
            if not me.adversary:                                 # simple_push.py:66-82
                return -np.sqrt(self._sq(me, goal))
            pos_rew = min([np.sqrt(self._sq(a, goal)) for a in goods])
            neg_rew = np.sqrt(np.sum(np.square(goal.pos - me.pos)))
            return pos_rew - neg_rew
        if self.spec.kind == SIMPLE_TAG:
            advs = [a for a in self.agents if a.adversary]
            goods = [a for a in self.agents if not a.adversary]
            r = 0
            if me.adversary:                       # simple_tag.py:115-129
                if me.collide:
                    for g in goods:
                        for a in advs:
                            if self._touching(g, a):
                                r += 10
                return r
            if me.collide:                         # simple_tag.py:89-113
                for a in advs:
                    if self._touching(a, me):
                        r -= 10
            for p in range(2):
                x = abs(me.pos[p])
                if x < 0.9:
                    pen = 0
                elif x < 1.0:
                    pen = (x - 0.9) * 10
                else:
                    pen = min(np.exp(2 * x - 2), 10)
                r -= pen
            return r
        r = 0                                      # simple_spread.py:71-81
        for lm in self.landmarks:
            r -= min(np.sqrt(np.sum(np.square(a.pos - lm.pos))) for a in self.agents)
        if me.collide:
            for a in self.agents:
                if self._touching(a, me):
                    r -= 1
        return r

    def info(self, i):
        me = self.agents[i]
        if self.spec.kind in (HIER_PURSUIT, HIER_MOVING):
            return self._hier_info(i)
        if self.spec.kind == SIMPLE_CRYPTO:                      # simple_crypto.py:78-80
            return np.concatenate([me.comm, np.eye(self.spec.dim_c)[self.goals[0]]])
        if self.spec.kind == SIMPLE_ADVERSARY:                   # simple_adversary.py:58-67
            goal = self.landmarks[self.goal]
            if me.adversary:
                return self._sq(me, goal)
            return tuple([self._sq(me, lm) for lm in self.landmarks] + [self._sq(me, goal)])
        if self.spec.kind in (SIMPLE, SIMPLE_PUSH):
            return {}
        if self.spec.kind in (SIMPLE_TAG, WORLD_COMM):   # simple_tag.py:57-66, simple_world_comm.py:119-127
            if not me.adversary:
                return 0
            return sum(1 for g in self.agents if not g.adversary and self._touching(g, me))
        r, coll, occ, md = 0, 0, 0, 0              # simple_spread.py:46-62
        for lm in self.landmarks:
            m = min(np.sqrt(np.sum(np.square(a.pos - lm.pos))) for a in self.agents)
            md += m
            r -= m
            if m < 0.1:
                occ += 1
        if me.collide:
            for a in self.agents:
                if self._touching(a, me):
                    r -= 1
                    coll += 1
        return (r, coll, md, occ)

    # -- environment.py:92-117 ---------------------------------------------------------------
    def step(self, action_n):
        policy = [i for i in range(self.n) if i not in self.scripted]      # world.policy_agents (core.py:115-116)
        for j, i in enumerate(policy):
            self._decode_action(self.agents[i], action_n[j])
        for i, cb in self.scripted.items():                                  # core.py:126-127
            self.agents[i].u = np.asarray(cb(self, i), dtype=np.float64)
            self.agents[i].c_act = np.zeros(self.spec.dim_c)
        self.world_step()
        obs_n, rew_n, done_n, info_n = [], [], [], {'n': []}
        for i in policy:
            obs_n.append(self.observation(i))
            rew_n.append(self.reward(i))
            done_n.append(False)
            info_n['n'].append(self.info(i) if self.benchmark and self.spec.info_width else {})
        return obs_n, rew_n, done_n, info_n


# --------------------------------------------------------------------------------------
# Batched restatement (SoA, vectorised over envs, same per-element operation order)
# --------------------------------------------------------------------------------------
def _pymin(m, x):
    """One step of Python's min() over a sequence, vectorised: the running minimum is replaced only where `x < m`, so
    a NaN that comes FIRST stays and a later NaN is passed over (np.minimum would spread either; the scenario files call
    the builtin: simple_spread.py:74, simple_adversary.py:96, simple_world_comm.py:163,203).  Such states exist: two
    coincident colliding entities get a NaN force (core.py:239-243) and keep NaN positions to the end of the episode."""
    with np.errstate(invalid='ignore'):
        return np.where(x < m, x, m)


def _norm2_blas(d0, d1):
    """np.linalg.norm of a 2-vector as the reference computes it (core.py:185,188): sqrt(dot(x, x)), and
    this container's BLAS dot rounds as fma(x1, x1, x0*x0) (checked against np.linalg.norm on 20 000
    random vectors).  Emulated with error-free transformations so that it vectorises."""
    d0 = np.asarray(d0, np.float64)
    d1 = np.asarray(d1, np.float64)
    s = d0 * d0
    p = d1 * d1
    c = 134217729.0 * d1                       # Veltkamp split of d1 (2^27 + 1)
    hi = c - (c - d1)
    lo = d1 - hi
    e = ((hi * hi - p) + 2.0 * hi * lo) + lo * lo          # d1*d1 == p + e exactly
    h = p + s                                              # TwoSum(p, s) == h + l exactly
    z = h - p
    l = (p - (h - z)) + (s - z)
    return np.sqrt(h + (l + e))


class BatchedOracle:
    """pos[E,2,B], vel[A,2,B]; act[A,5,B]; noise[A,2,B] standard normals (or None)."""

    def __init__(self, spec: ScenarioSpec, dtype=np.float64, discrete_action_input=False, force_discrete_action=False,
                 discrete_action_space=True):
        self.spec = spec
        self.dt = np.dtype(dtype).type
        # environment.py:29,191-192.  False: rows 0, 1 of an agent's 5 physical rows hold its force command u
        self.discrete_action_space = discrete_action_space
        if not discrete_action_space:
            assert all(sl or not mv for sl, mv in zip(spec.silent, spec.movable)), 'core.py:216 fails for a 5-vector u'
            assert not discrete_action_input
        # environment.py:31,33.  With discrete_action_input the physical rows of an agent hold its integer action
        # (row 0 is read); with force_discrete_action they are replaced by one_hot(argmax) before the decode.
        self.discrete_action_input = discrete_action_input
        self.force_discrete_action = force_discrete_action
        if discrete_action_input:
            assert all(spec.silent), 'environment.py:208: assert len(action) == 0'

    def physical_action(self, a):
        """The 5 physical rows a[5,B] of one movable agent -> the 5 rows the default decode (environment.py:188-190)
        reads: unchanged, or environment.py:176-182 / :184-187 first."""
        T = self.dt
        if not self.discrete_action_space:          # u = action[0] (after the forced argmax, :184-187): (0, ux, 0, uy, 0)
            ux, uy = a[0], a[1]
            if self.force_discrete_action:
                second = np.argmax(np.stack([ux, uy]), axis=0) == 1
                ux, uy = np.where(second, T(0), T(1)), np.where(second, T(1), T(0))
            z = np.zeros_like(a)
            z[1], z[3] = ux, uy
            return z
        if self.discrete_action_input:
            k = a[0]
            z = np.zeros_like(a)
            z[2] = np.where(k == 1, T(1), T(0))     # u[0] = -1  ==  a[1] - a[2] with a[2] = 1
            z[1] = np.where(k == 2, T(1), T(0))
            z[4] = np.where(k == 3, T(1), T(0))
            z[3] = np.where(k == 4, T(1), T(0))
            return z
        if self.force_discrete_action:
            d = np.argmax(a, axis=0)                # first maximum; NaN counts as the maximum, like the reference's call
            return (np.arange(5)[:, None] == d[None, :]).astype(T)
        return a

    def _leader_force(self, pos, act, force):
        """core.py:153-194, vectorised; returns lforce [A,2,B] (agent.leader_force after the step)."""
        s, T = self.spec, self.dt
        A = s.n_agents
        B = pos.shape[-1]
        offs = np.concatenate([[0], np.cumsum(s.act_dims)])
        lf = np.zeros((A, 2, B), T)
        for i in range(A):
            if s.forced_id[i] < 0:
                continue
            for l in range(A):
                if s.force_id[l] < 0 or s.forced_id[i] != s.force_id[l]:
                    continue
                # a silent agent's action.c stays zeros(dim_c) (environment.py:80,198-206)
                c = (np.zeros((s.dim_c, B), T) if s.silent[l]
                     else act[offs[l] + s.comm_row0(l):offs[l] + s.comm_row0(l) + s.dim_c])
                if s.advsuperleader[l]:
                    lf[i] = c[0:2]
                    continue
                cmd = np.argmax(c, axis=0)
                if s.dim_c == 3:
                    cmd = np.array([0, 5, 6])[cmd]
                f = np.zeros((2, B), T)
                f[0] = np.where(cmd == 1, T(-1), np.where(cmd == 2, T(1), T(0)))
                f[1] = np.where(cmd == 3, T(-1), np.where(cmd == 4, T(1), T(0)))
                norm = _norm2_blas if T is np.float64 else (lambda a, b: np.sqrt(a * a + b * b))
                d = pos[l] - pos[i]
                unit = d / norm(d[0], d[1])
                f = np.where(cmd == 5, unit, f)
                d = pos[i] - pos[l]
                unit = d / norm(d[0], d[1])
                f = np.where(cmd == 6, unit, f)
                force[i] = force[i] + f
                lf[i] = lf[i] + f
        return lf

    def world_step(self, pos, vel, act, noise=None, return_lforce=False, scripted_u=None, lvel=None):
        """act: [A,5,B] (all agents silent) or ragged rows [sum(act_dims), B] (environment.py:161-168).
        scripted_u: {agent index: u [2, B]} for agents with an action_callback (core.py:126-127): the force as the
        callback returned it, no sensitivity; their action rows are not looked at.
        lvel: [L,2,B] velocities of the landmarks (worlds with a movable landmark); then returns (pos, vel, lvel)."""
        s, T = self.spec, self.dt
        A, E = s.n_agents, s.n_entities
        pos = np.array(pos, dtype=T)
        vel = np.array(vel, dtype=T)
        if lvel is not None:
            assert not return_lforce
            vel = np.concatenate([vel, np.array(lvel, dtype=T)], 0)      # per entity from here on
        else:
            assert not any(s.movable[A:]), 'a movable landmark needs lvel'
        ms_all = list(s.max_speed) + list(s.landmark_max_speed)
        act = np.asarray(act, dtype=T)
        if act.ndim == 3:
            act = act.reshape(-1, act.shape[-1])
        offs = np.concatenate([[0], np.cumsum(s.act_dims)])
        force = [None] * E
        lforce = None
        with np.errstate(all='ignore'):
            for i in range(A):
                if not s.movable[i]:          # no physical part in its action (environment.py:174)
                    continue
                sens = T(5.0 if s.accel[i] is None else s.accel[i])
                a = self.physical_action(act[offs[i]:offs[i] + 5])
                u = np.stack([(a[1] - a[2]) * sens, (a[3] - a[4]) * sens])
                if scripted_u is not None and i in scripted_u:
                    u = np.asarray(scripted_u[i], dtype=T)
                if s.movable[i]:
                    if s.u_noise[i]:
                        u = u + np.asarray(noise[i], dtype=T) * T(s.u_noise[i])
                    force[i] = u
            if s.has_leader_force:
                lforce = self._leader_force(pos, act, force)
            k = T(s.contact_margin)
            cf = T(s.contact_force)
            for a in range(E):
                if not (s.movable[a] and s.collide[a]):
                    continue
                for b in range(a + 1, E):
                    if not s.collide[b]:
                        continue
                    d = pos[a] - pos[b]
                    dist = np.sqrt(np.square(d[0]) + np.square(d[1]))
                    dmin = T(s.size[a]) + T(s.size[b])
                    pen = np.logaddexp(T(0), -(dist - dmin) / k) * k
                    f = cf * d / dist * pen
                    force[a] = f + (force[a] if force[a] is not None else T(0))
            for i in range(E):
                if not s.movable[i]:
                    continue
                v = vel[i] * T(1 - s.damping)
                if force[i] is not None:
                    v = v + (force[i] / T(s.mass[i])) * T(s.dt)
                if ms_all[i] is not None:
                    ms = T(ms_all[i])
                    speed = np.sqrt(np.square(v[0]) + np.square(v[1]))
                    clamped = v / speed * ms
                    v = np.where(speed > ms, clamped, v)
                vel[i] = v
                pos[i] = pos[i] + v * T(s.dt)
        if return_lforce:
            return pos, vel, (lforce if lforce is not None else np.zeros((A, 2, pos.shape[-1]), T))
        if lvel is not None:
            return pos, vel[:A], vel[A:]
        return pos, vel

    def comm_state(self, act, cnoise=None):
        """agent.state.c after a step (core.py:224-230): the comm entries of the action for agents that
        speak (+ c_noise * cnoise, standard normals [A, dim_c, B]), zeros for silent ones.  [A, dim_c, B]."""
        s, T = self.spec, self.dt
        act = np.asarray(act, dtype=T)
        if act.ndim == 3:
            act = act.reshape(-1, act.shape[-1])
        offs = np.concatenate([[0], np.cumsum(s.act_dims)])
        c = np.zeros((s.n_agents, s.dim_c, act.shape[-1]), T)
        for i in range(s.n_agents):
            if not s.silent[i]:
                c[i] = act[offs[i] + s.comm_row0(i):offs[i] + s.comm_row0(i) + s.dim_c]
                if s.c_noise[i] and cnoise is not None:
                    c[i] = c[i] + np.asarray(cnoise[i], T) * T(s.c_noise[i])
        return c

    def _observe_comm(self, pos, vel, goal, comm):
        """simple_speaker_listener.py:66-94 / simple_reference.py:56-89.  goal: [goal_rows, B] landmark index of
        agent.goal_b; comm: agent.state.c [A, dim_c, B]."""
        s, T = self.spec, self.dt
        A, L = s.n_agents, s.n_landmarks
        B = pos.shape[-1]
        goal = np.asarray(goal, np.int64).reshape(-1, B)
        comm = np.zeros((A, s.dim_c, B), T) if comm is None else np.asarray(comm, T)
        base, hi = COMM_COLORS[s.kind]
        sq = lambda p, q: np.square(p[0] - q[0]) + np.square(p[1] - q[1])
        lm_of = lambda g: pos[A:][g, :, np.arange(B)].T            # [2,B] position of landmark g[b]
        obs, rew = [], np.zeros((A, B), T)
        for i in range(A):
            g = goal[i if s.kind == SIMPLE_REFERENCE else 0]
            color = np.full((3, B), base, T)
            color[g, np.arange(B)] = T(hi)
            rel_lm = [pos[A + l] - pos[i] for l in range(L)]
            other_c = [comm[j] for j in range(A) if j != i]
            if s.kind == SIMPLE_REFERENCE:
                rows = [vel[i]] + rel_lm + [color] + other_c
                rew[i] = -sq(pos[1 - i], lm_of(g))
            else:
                rows = [color] if not s.movable[i] else [vel[i]] + rel_lm + other_c
                rew[i] = -sq(pos[1], lm_of(g))
            obs.append(np.concatenate(rows, axis=0))
        return obs, rew, np.zeros((A, 0, B), T)

    def _observe_crypto(self, goal, comm):
        """simple_crypto.py:78-171.  goal: [2, B] = (goal landmark, key landmark); comm: agent.state.c [3, dim_c, B]."""
        s, T = self.spec, self.dt
        C = s.dim_c
        goal = np.asarray(goal, np.int64).reshape(2, -1)
        B = goal.shape[1]
        comm = np.zeros((3, C, B), T) if comm is None else np.asarray(comm, T)
        onehot = lambda g: (np.arange(C)[:, None] == g[None, :]).astype(T)      # [C, B]
        gcol, key = onehot(goal[0]), onehot(goal[1])
        obs = [comm[2], np.concatenate([key, comm[2]], 0), np.concatenate([gcol, key], 0)]
        sq = lambda c: ((np.square(c[0] - gcol[0]) + np.square(c[1] - gcol[1])) + np.square(c[2] - gcol[2])) \
            + np.square(c[3] - gcol[3])                                         # np.sum of 4 terms: left to right
        spoke = lambda c: ~np.all(c == 0, axis=0)
        with np.errstate(all='ignore'):
            eve = np.where(spoke(comm[0]), sq(comm[0]), T(0))
            bob = np.where(spoke(comm[1]), -sq(comm[1]), T(0))
        rew = np.stack([np.where(spoke(comm[0]), -sq(comm[0]), T(0)), eve + bob, eve + bob]).astype(T)
        info = np.stack([np.concatenate([comm[i], gcol], 0) for i in range(3)])
        return obs, rew, info

    def _observe_moving(self, pos, vel, lforce):
        """*_MovingTarget gathers, vectorised (L1F4_MovingTarget.py:128-258)."""
        s, T = self.spec, self.dt
        A = s.n_agents
        B = pos.shape[-1]
        dist = lambda e, f: self._dist(pos, e, f)
        obs, rew, info = [], np.zeros((A, B), T), np.zeros((A, 1, B), T)
        goods = [a for a in range(A) if not s.adversary[a]]
        with np.errstate(all='ignore'):
            touch = lambda e, f: dist(e, f) < T(s.size[e] + s.size[f])
            adv_r = np.zeros(B, T)
            count = np.zeros(B, np.int64)
            for a in goods:
                for o in range(A):
                    if o == a:
                        continue
                    h = touch(a, o)
                    count = count + h
                    adv_r = adv_r + T(10) * h
                adv_r = adv_r + T(50) * (count >= 4)
            for i in range(A):
                rows = [vel[i], pos[i]]
                vis = []
                for j in range(A):
                    if j == i:
                        continue
                    if s.adversary[i]:
                        linked = ((s.force_id[i] >= 0 and s.force_id[i] == s.forced_id[j])
                                  or (s.forced_id[i] >= 0 and s.forced_id[i] == s.force_id[j]))
                        v = dist(j, i) < T(s.obs_r[i])
                        if s.reward_variant == 0 and linked:
                            v = np.ones(B, bool)
                        rows.append(np.where(v, pos[j] - pos[i], T(0)))
                    else:
                        v = np.ones(B, bool)
                        rows.append(pos[j])
                    vis.append((j, v))
                rows += [np.where(v, vel[j], T(0)) for j, v in vis]
                if s.observes_leader_force(i):
                    rows.append(lforce[i])
                obs.append(np.concatenate(rows, axis=0))
                if s.adversary[i]:
                    r = adv_r.copy()
                else:
                    r = np.zeros(B, T)
                    if s.collide[i]:
                        for a in range(A):
                            if s.adversary[a]:
                                r = r - touch(a, i)
                for p in range(2):
                    x = np.abs(pos[i, p])
                    bnd = np.where(x < T(0.9), T(0),
                                   np.where(x < T(1.0), (x - T(0.9)) * T(10),
                                            np.minimum(np.exp(T(2) * x - T(2)), T(10))))
                    r = r - bnd
                rew[i] = r
                for a in range(A):
                    if s.adversary[a] != s.adversary[i]:
                        info[i, 0] += touch(a, i)
        return obs, rew, info

    def _observe_hier(self, pos, vel, lforce):
        """L1F*/S1L* gathers, vectorised (e.g. L1F2_3Targets_100observe.py:136-253)."""
        s, T = self.spec, self.dt
        A, L = s.n_agents, s.n_landmarks
        B = pos.shape[-1]
        sq = lambda p, q: np.square(p[0] - q[0]) + np.square(p[1] - q[1])
        touch = lambda e, f: np.sqrt(sq(pos[e], pos[f])) < T(s.size[e] + s.size[f])
        obs, rew, info = [], np.zeros((A, B), T), np.zeros((A, 1, B), T)
        with np.errstate(all='ignore'):
            for i in range(A):
                sq_rad = T(s.obs_r[i]) * T(s.obs_r[i])
                rows = [vel[i], pos[i]]
                vis = []
                for j in range(A):
                    if j == i:
                        continue
                    linked = ((s.force_id[i] >= 0 and s.force_id[i] == s.forced_id[j])
                              or (s.forced_id[i] >= 0 and s.forced_id[i] == s.force_id[j]))
                    v = np.ones(B, bool) if linked else (sq(pos[j], pos[i]) < sq_rad)
                    vis.append((j, v))
                    rows.append(np.where(v, pos[j] - pos[i], T(0)))
                rows += [np.where(v, vel[j], T(0)) for j, v in vis]
                if s.forced_id[i] >= 0:
                    rows.append(lforce[i])
                if L:
                    d2 = np.stack([sq(pos[A + l], pos[i]) for l in range(L)])          # [L,B]
                    seen = d2 < sq_rad
                    order = np.argsort(np.where(seen, d2, np.inf), axis=0, kind='stable')
                    rel = np.stack([pos[A + l] - pos[i] for l in range(L)])              # [L,2,B]
                    for k in range(min(L, s.n_target_obs)):
                        idx = order[k]
                        pick = rel[idx, :, np.arange(B)].T                              # [2,B]
                        rows.append(np.where(seen[idx, np.arange(B)], pick, T(0)))
                obs.append(np.concatenate(rows, axis=0))
                # reward
                r = np.zeros(B, T)
                if s.reward_variant == 0:
                    for l in range(L):
                        if not s.advleader[i]:
                            r = r + T(10) * touch(A + l, i)
                        elif not s.advsuperleader[i]:
                            for o in range(A):
                                if s.forced_id[o] == s.force_id[i]:
                                    r = r + T(10) * touch(A + l, o)
                        else:
                            for o in range(A):
                                if not s.advleader[o]:
                                    r = r + T(10) * touch(A + l, o)
                else:
                    for l in range(L):
                        for o in range(A):
                            if s.advsuperleader[i]:
                                if s.forced_id[o] == s.force_id[i] and 0.0 < s.obs_r[o]:
                                    r = r + T(1)
                            elif s.advleader[i]:
                                if s.forced_id[o] == s.force_id[i]:
                                    r = r + T(10) * touch(A + l, o)
                        if not s.advleader[i]:
                            r = r + T(10) * touch(A + l, i)
                for p in range(2):
                    x = np.abs(pos[i, p])
                    bnd = np.where(x < T(0.9), T(0),
                                   np.where(x < T(1.0), (x - T(0.9)) * T(10),
                                            np.minimum(np.exp(T(2) * x - T(2)), T(10))))
                    r = r - bnd
                rew[i] = r
                for a in range(A):
                    if s.adversary[a] != s.adversary[i]:
                        info[i, 0] += touch(a, i)
        return obs, rew, info

    def _dist(self, pos, i, j):
        d = pos[i] - pos[j]
        return np.sqrt(np.square(d[0]) + np.square(d[1]))

    def _observe_f1(self, pos, vel, goal):
        """simple / simple_adversary / simple_push gathers."""
        s, T = self.spec, self.dt
        A, L = s.n_agents, s.n_landmarks
        B = pos.shape[-1]
        goal = np.zeros(B, np.int64) if goal is None else np.asarray(goal, np.int64)
        gp = pos[A:][goal, :, np.arange(B)].T                   # goal landmark position [2,B]
        sq = lambda p, q: np.square(p[0] - q[0]) + np.square(p[1] - q[1])
        advs = [i for i in range(A) if s.adversary[i]]
        goods = [i for i in range(A) if not s.adversary[i]]
        obs, rew = [], np.zeros((A, B), T)
        info = np.zeros((A, s.info_width, B), T)
        for i in range(A):
            rel_lm = [pos[A + l] - pos[i] for l in range(L)]
            rel_ag = [pos[j] - pos[i] for j in range(A) if j != i]
            if s.kind == SIMPLE:
                rows = [vel[i]] + rel_lm
            elif s.kind == SIMPLE_ADVERSARY:
                rows = ([] if s.adversary[i] else [gp - pos[i]]) + rel_lm + rel_ag
            elif s.adversary[i]:
                rows = [vel[i]] + rel_lm + rel_ag
            else:
                mine = np.full((3, B), 0.25, T)
                mine[goal + 1, np.arange(B)] += T(0.5)
                lmc = []
                for l in range(L):
                    c = np.full((3, B), 0.1, T)
                    c[l + 1] += T(0.8)
                    lmc.append(c)
                rows = [vel[i], gp - pos[i], mine] + rel_lm + lmc + rel_ag
            obs.append(np.concatenate(rows, axis=0))
        with np.errstate(all='ignore'):
            if s.kind == SIMPLE:
                rew[0] = -sq(pos[0], pos[A])
            elif s.kind == SIMPLE_ADVERSARY:
                adv_rew = 0
                for a in advs:
                    adv_rew = adv_rew + np.sqrt(sq(pos[a], gp))
                m = np.sqrt(sq(pos[goods[0]], gp))
                for g in goods[1:]:
                    m = _pymin(m, np.sqrt(sq(pos[g], gp)))
                for i in range(A):
                    if s.adversary[i]:
                        rew[i] = -sq(pos[i], gp)
                        info[i, 0] = sq(pos[i], gp)
                    else:
                        rew[i] = -m + adv_rew
                        for l in range(L):
                            info[i, l] = sq(pos[i], pos[A + l])
                        info[i, L] = sq(pos[i], gp)
            else:
                m = np.sqrt(sq(pos[goods[0]], gp))
                for g in goods[1:]:
                    m = _pymin(m, np.sqrt(sq(pos[g], gp)))
                for i in range(A):
                    if s.adversary[i]:
                        rew[i] = m - np.sqrt(sq(gp, pos[i]))
                    else:
                        rew[i] = -np.sqrt(sq(pos[i], gp))
        return obs, rew, info

    def _observe_world_comm(self, pos, vel, comm):
        """simple_world_comm.py:119-313, vectorised over envs in the reference's operation order."""
        s, T = self.spec, self.dt
        A, L = s.n_agents, s.n_landmarks
        B = pos.shape[-1]
        assert s.n_forests == 2
        comm = np.zeros((A, s.dim_c, B), T) if comm is None else np.asarray(comm, T)
        f0 = A + L - s.n_forests
        food = list(range(A + L - s.n_forests - s.n_food, A + L - s.n_forests))
        advs = [i for i in range(A) if s.adversary[i]]
        goods = [i for i in range(A) if not s.adversary[i]]
        with np.errstate(all='ignore'):
            touch = lambda a, b: self._dist(pos, a, b) < T(s.size[a] + s.size[b])
            inf = [[touch(i, f0 + k) for k in range(2)] for i in range(A)]
            obs = []
            for i in range(A):
                rows = [vel[i], pos[i]] + [pos[A + l] - pos[i] for l in range(L)]
                other_vel = []
                for j in range(A):
                    if j == i:
                        continue
                    vis = ((inf[i][0] & inf[j][0]) | (inf[i][1] & inf[j][1])
                           | (~inf[i][0] & ~inf[j][0] & ~inf[i][1] & ~inf[j][1]))
                    if s.leader[i]:
                        vis = np.ones(B, bool)
                    rows.append(np.where(vis, pos[j] - pos[i], T(0)))
                    if not s.adversary[j]:
                        other_vel.append(np.where(vis, vel[j], T(0)))
                in_forest = [np.where(inf[i][k], T(1), T(-1))[None] for k in range(2)]
                if s.adversary[i]:
                    rows += other_vel + in_forest + [comm[0]]
                else:
                    rows += in_forest + other_vel
                obs.append(np.concatenate(rows, axis=0))
            rew = np.zeros((A, B), T)
            info = np.zeros((A, 1, B), T)
            for i in advs:
                m = self._dist(pos, goods[0], i)
                for g in goods[1:]:
                    m = _pymin(m, self._dist(pos, g, i))
                r = T(0) - T(0.1) * m
                if s.collide[i]:
                    for g in goods:
                        for a in advs:
                            r = r + T(5) * touch(g, a)
                rew[i] = r
                for g in goods:
                    info[i, 0] += touch(g, i)
            for i in goods:
                r = np.zeros(B, T)
                if s.collide[i]:
                    for a in advs:
                        r = r - T(5) * touch(a, i)
                for p in range(2):
                    x = np.abs(pos[i, p])
                    bnd = np.where(x < T(0.9), T(0),
                                   np.where(x < T(1.0), (x - T(0.9)) * T(10),
                                            np.minimum(np.exp(T(2) * x - T(2)), T(10))))
                    r = r - T(2) * bnd
                for f in food:
                    r = r + T(2) * touch(i, f)
                m = self._dist(pos, food[0], i)
                for f in food[1:]:
                    m = _pymin(m, self._dist(pos, f, i))
                rew[i] = r + T(0.05) * m
        return obs, rew, info

    def observe(self, pos, vel, want_info=True, goal=None, lforce=None, comm=None):
        """Returns obs (list of [dim_i, B]), rew [A,B], info [A,W,B]."""
        s, T = self.spec, self.dt
        A, L, E = s.n_agents, s.n_landmarks, s.n_entities
        B = pos.shape[-1]
        pos = np.asarray(pos, dtype=T)
        vel = np.asarray(vel, dtype=T)
        if s.kind in (HIER_PURSUIT, HIER_MOVING):
            lforce = np.zeros((A, 2, B), T) if lforce is None else np.asarray(lforce, T)
            obs, rew, info = (self._observe_hier if s.kind == HIER_PURSUIT else self._observe_moving)(pos, vel, lforce)
            return obs, rew, (info if want_info else None)
        if s.kind in (SIMPLE, SIMPLE_ADVERSARY, SIMPLE_PUSH):
            obs, rew, info = self._observe_f1(pos, vel, goal)
            return obs, rew, (info if want_info else None)
        if s.kind in COMM_KINDS:
            obs, rew, info = self._observe_comm(pos, vel, goal, comm)
            return obs, rew, (info if want_info else None)
        if s.kind == SIMPLE_CRYPTO:
            assert s.dim_c == 4
            obs, rew, info = self._observe_crypto(goal, comm)
            return obs, rew, (info if want_info else None)
        if s.kind == WORLD_COMM:
            obs, rew, info = self._observe_world_comm(pos, vel, comm)
            return obs, rew, (info if want_info else None)
        obs = []
        for i in range(A):
            rows = [vel[i], pos[i]]
            rows += [pos[A + l] - pos[i] for l in range(L)]
            rows += [pos[j] - pos[i] for j in range(A) if j != i]
            if s.kind == SIMPLE_TAG:
                rows += [vel[j] for j in range(A) if j != i and not s.adversary[j]]
            else:
                rows += [np.zeros((s.dim_c, B), T) for j in range(A) if j != i]
            obs.append(np.concatenate(rows, axis=0))
        rew = np.zeros((A, B), T)
        info = np.zeros((A, s.info_width, B), T)
        with np.errstate(all='ignore'):
            if s.kind == SIMPLE_TAG:
                advs = [i for i in range(A) if s.adversary[i]]
                goods = [i for i in range(A) if not s.adversary[i]]
                total = np.zeros(B, T)
                for g in goods:
                    n_g = np.zeros(B, T)
                    for a in advs:
                        hit = self._dist(pos, g, a) < T(s.size[g] + s.size[a])
                        n_g = n_g + hit
                        info[a, 0] += hit
                    total = total + n_g
                    r = -T(10) * n_g if s.collide[g] else np.zeros(B, T)
                    for p in range(2):
                        x = np.abs(pos[g, p])
                        bnd = np.where(x < T(0.9), T(0),
                                       np.where(x < T(1.0), (x - T(0.9)) * T(10),
                                                np.minimum(np.exp(T(2) * x - T(2)), T(10))))
                        r = r - bnd
                    rew[g] = r
                for a in advs:
                    rew[a] = T(10) * total if s.collide[a] else 0
            else:
                r0 = np.zeros(B, T)
                md = np.zeros(B, T)
                occ = np.zeros(B, T)
                for l in range(L):
                    m = self._dist(pos, 0, A + l)
                    for a in range(1, A):
                        m = _pymin(m, self._dist(pos, a, A + l))
                    md = md + m
                    r0 = r0 - m
                    occ = occ + (m < T(0.1))
                for i in range(A):
                    r = r0.copy()
                    c = np.zeros(B, T)
                    if s.collide[i]:
                        for a in range(A):
                            hit = self._dist(pos, a, i) < T(s.size[a] + s.size[i])
                            r = r - hit
                            c = c + hit
                    rew[i] = r
                    info[i, 0], info[i, 1], info[i, 2], info[i, 3] = r, c, md, occ
        return obs, rew, (info if want_info else None)

    def env_step(self, pos, vel, act, noise=None, goal=None, return_lforce=False, cnoise=None, scripted_u=None, lvel=None):
        if lvel is not None:             # a world with a movable landmark: returns (..., info, lvel)
            pos, vel, lvel = self.world_step(pos, vel, act, noise, scripted_u=scripted_u, lvel=lvel)
            obs, rew, info = self.observe(pos, vel, goal=goal)
            return pos, vel, obs, rew, info, lvel
        pos, vel, lforce = self.world_step(pos, vel, act, noise, return_lforce=True, scripted_u=scripted_u)
        obs, rew, info = self.observe(pos, vel, goal=goal, lforce=lforce,
                                      comm=self.comm_state(act, cnoise) if self.spec.kind in COMM_KINDS + (SIMPLE_CRYPTO, WORLD_COMM) else None)
        if return_lforce:
            return pos, vel, obs, rew, info, lforce
        return pos, vel, obs, rew, info


# --------------------------------------------------------------------------------------
# target_move (experiments/train_and_eval.py:207-222)          *** parity unpinned ***
# The reference's tests and fixtures hold nothing for this function and running it here was refused
# (see DESIGN.md section 4), so this restatement is checked only against itself: serial form
# (reference loop shape and np.random.rand order) against the vectorised form's decisions.
# --------------------------------------------------------------------------------------
def target_move_serial(env: 'SerialEnv'):
    """train_and_eval.py:212-222 on a SerialEnv: landmarks in order, agents in order, first touching
    follower respawns the landmark at (rand()*2-1, rand()*2-1) and ends the inner loop."""
    count = 0
    for li, land in enumerate(env.landmarks):
        for ai, agent in enumerate(env.agents):
            if not env.spec.advfollower[ai]:
                continue
            if SerialEnv._touching(land, agent):
                count += 1
                land.pos = np.array([np.random.rand() * 2 - 1, np.random.rand() * 2 - 1])
                break
    return count


def target_move(spec: ScenarioSpec, pos, seed: int, counter: int, env_ids, dtype=np.float64):
    """Vectorised form with this build's Philox respawn stream (counter = (env_lo, env_hi, move counter,
    0x80000000 | landmark), key = seed).  Returns (new pos [E,2,B], count [B])."""
    T = np.dtype(dtype).type
    A, L = spec.n_agents, spec.n_landmarks
    pos = np.array(pos, dtype=T)
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    B = env_ids.shape[0]
    count = np.zeros(B, np.int32)
    k0 = np.full(B, seed & 0xFFFFFFFF, np.uint32)
    k1 = np.full(B, (seed >> 32) & 0xFFFFFFFF, np.uint32)
    c0 = (env_ids & np.uint64(0xFFFFFFFF)).astype(np.uint32)
    c1 = (env_ids >> np.uint64(32)).astype(np.uint32)
    with np.errstate(all='ignore'):
        for l in range(L):
            hit = np.zeros(B, bool)
            for a in range(A):
                if not spec.advfollower[a]:
                    continue
                d = pos[A + l] - pos[a]
                hit |= np.sqrt(np.square(d[0]) + np.square(d[1])) < T(spec.size[A + l]) + T(spec.size[a])
            w = philox4x32_10([c0, c1, np.full(B, counter, np.uint32), np.full(B, 0x80000000 | l, np.uint32)], [k0, k1])
            nx = (-1.0 + 2.0 * _u01(w[0], w[1])).astype(T)
            ny = (-1.0 + 2.0 * _u01(w[2], w[3])).astype(T)
            pos[A + l, 0] = np.where(hit, nx, pos[A + l, 0])
            pos[A + l, 1] = np.where(hit, ny, pos[A + l, 1])
            count += hit
    return pos, count


# --------------------------------------------------------------------------------------
# Scripted prey (experiments/simple_train.py:157-205)             *** parity unpinned ***
# good_agent_action / good_agent_victim_action live in the training script, which holds no fixture for
# them and whose code is not run here (same refusal as target_move).  The serial
# form follows the reference text statement by statement; the vectorised form is checked against it.
# --------------------------------------------------------------------------------------
WORLD_EDGE = [[-0.9, 0.9], [0.9, 0.9], [-0.9, -0.9], [0.9, -0.9]]      # simple_train.py:158


def _prey_act(vec, mag, cap):
    """simple_train.py:175-187 / :192-202: half the offset, one of the two opposing slots per axis."""
    vec = [vec[0] / 2, vec[1] / 2]
    act = np.array([0.0 for _ in range(5)])
    if vec[0] < 0:
        act[2] = min(abs(vec[0]) * mag, cap)
    else:
        act[1] = min(vec[0] * mag, cap)
    if vec[1] < 0:
        act[4] = min(abs(vec[1]) * mag, cap)
    else:
        act[3] = min(vec[1] * mag, cap)
    return act


def good_agent_action_serial(env: 'SerialEnv', agent=-1, cap=1.3):
    """simple_train.py:157-188: head for the corner of [-0.9, 0.9]^2 whose summed distance to every other
    agent is largest (first maximum).  Returns (act[5], target_edge[2])."""
    me = env.agents[agent]
    other_pos = [o.pos for o in env.agents if o is not me]
    dis = [0.0 for _ in WORLD_EDGE]
    for i, edge in enumerate(WORLD_EDGE):
        for pos in other_pos:
            dis[i] += np.linalg.norm(edge - pos)
    idx = dis.index(max(dis))
    target_edge = WORLD_EDGE[idx]
    return _prey_act(target_edge - me.pos, 1, cap), np.array(target_edge)


def good_agent_victim_action_serial(env: 'SerialEnv', target, agent=-1, cap=1.3):
    """simple_train.py:190-205: walk towards agent `target` (the curriculum's easy prey), gain 7."""
    me = env.agents[agent]
    return _prey_act(env.agents[target].pos - me.pos, 7, cap)


def scripted_action(spec: ScenarioSpec, pos, agent, mode, target=0, cap=1.3, dtype=np.float64):
    """Vectorised form of the two functions above.  pos [E,2,B]; mode 0 = good_agent_action, 1 =
    good_agent_victim_action(target).  Returns (act [5,B], target point [2,B])."""
    T = np.dtype(dtype).type
    pos = np.asarray(pos, T)
    A = spec.n_agents
    agent = agent % A
    B = pos.shape[-1]
    with np.errstate(all='ignore'):
        if mode == 0:
            best_d, tx, ty = None, None, None
            for ex, ey in WORLD_EDGE:
                d = np.zeros(B, T)
                for o in range(A):
                    if o == agent:
                        continue
                    dx, dy = T(ex) - pos[o, 0], T(ey) - pos[o, 1]
                    d = d + (_norm2_blas(dx, dy) if T is np.float64 else np.sqrt(dx * dx + dy * dy))
                if best_d is None:
                    best_d, tx, ty = d, np.full(B, T(ex)), np.full(B, T(ey))
                else:
                    take = d > best_d
                    best_d = np.where(take, d, best_d)
                    tx, ty = np.where(take, T(ex), tx), np.where(take, T(ey), ty)
            mag = T(1)
        else:
            tx, ty = pos[target, 0], pos[target, 1]
            mag = T(7)
        act = np.zeros((5, B), T)
        for c, t in ((0, tx), (1, ty)):
            v = (t - pos[agent, c]) / T(2)
            neg = v < 0
            m = np.abs(v) * mag
            m = np.where(T(cap) < m, T(cap), m)           # Python min(m, cap)
            act[2 + 2 * c] = np.where(neg, m, T(0))
            act[1 + 2 * c] = np.where(neg, T(0), m)
    return act, np.stack([tx, ty]).astype(T)


# --------------------------------------------------------------------------------------
# Device reset RNG restatement (this build's own spec; no reference counterpart)
# --------------------------------------------------------------------------------------
_M0, _M1 = np.uint64(0xD2511F53), np.uint64(0xCD9E8D57)
_W0, _W1 = np.uint32(0x9E3779B9), np.uint32(0xBB67AE85)
_MASK = np.uint64(0xFFFFFFFF)


def philox4x32_10(ctr, key):
    """Philox4x32-10 (Salmon et al., SC'11).  ctr: 4 arrays of uint32, key: 2 arrays of uint32."""
    c = [np.asarray(x, dtype=np.uint32).copy() for x in ctr]
    k = [np.asarray(x, dtype=np.uint32).copy() for x in key]
    with np.errstate(over='ignore'):
        for r in range(10):
            p0 = _M0 * c[0].astype(np.uint64)
            p1 = _M1 * c[2].astype(np.uint64)
            hi0, lo0 = (p0 >> np.uint64(32)).astype(np.uint32), (p0 & _MASK).astype(np.uint32)
            hi1, lo1 = (p1 >> np.uint64(32)).astype(np.uint32), (p1 & _MASK).astype(np.uint32)
            c = [hi1 ^ c[1] ^ k[0], lo1, hi0 ^ c[3] ^ k[1], lo0]
            k = [(k[0] + _W0).astype(np.uint32), (k[1] + _W1).astype(np.uint32)]
    return c


def _u01(hi, lo):
    """53-bit uniform in [0,1) from two uint32 words."""
    a = (hi >> np.uint32(5)).astype(np.float64)
    b = (lo >> np.uint32(6)).astype(np.float64)
    return (a * 67108864.0 + b) * (1.0 / 9007199254740992.0)


def reset_goals(spec: ScenarioSpec, seed: int, episode, env_ids):
    """Goal landmark index per env: Philox word 0 at entity = 0xFFFFFFFF, mapped to [0, L) by
    (w * L) >> 32  (the reference: np.random.choice(world.landmarks), simple_adversary.py:45)."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    B = env_ids.shape[0]
    episode = np.broadcast_to(np.asarray(episode, dtype=np.uint32), (B,))
    lo = (env_ids & _MASK).astype(np.uint32)
    hi = (env_ids >> np.uint64(32)).astype(np.uint32)
    k0 = np.full(B, seed & 0xFFFFFFFF, np.uint32)
    k1 = np.full(B, (seed >> 32) & 0xFFFFFFFF, np.uint32)
    w = philox4x32_10([lo, hi, episode, np.full(B, 0xFFFFFFFF, np.uint32)], [k0, k1])
    pick = lambda word: ((word.astype(np.uint64) * np.uint64(spec.n_landmarks)) >> np.uint64(32)).astype(np.int32)
    if spec.goal_rows > 1:      # simple_reference: one goal per agent from words 0, 1, ... of the same draw
        return np.stack([pick(w[r]) for r in range(spec.goal_rows)])
    return pick(w[0])


def reset_positions(spec: ScenarioSpec, seed: int, episode, env_ids, dtype=np.float64):
    """pos[E,2,B] for envs ``env_ids`` (global ids) at per-env episode counter ``episode``.

    counter = (env_lo, env_hi, episode, entity), key = (seed_lo, seed_hi);
    x <- words (0,1), y <- words (2,3);  p = -r + (2r) * u  (np.random.uniform's formula).
    """
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    B = env_ids.shape[0]
    episode = np.broadcast_to(np.asarray(episode, dtype=np.uint32), (B,))
    lo = (env_ids & _MASK).astype(np.uint32)
    hi = (env_ids >> np.uint64(32)).astype(np.uint32)
    k0 = np.full(B, seed & 0xFFFFFFFF, np.uint32)
    k1 = np.full(B, (seed >> 32) & 0xFFFFFFFF, np.uint32)
    pos = np.zeros((spec.n_entities, 2, B), dtype)
    for e in range(spec.n_entities):
        r = spec.agent_pos_range if e < spec.n_agents else spec.landmark_pos_range
        w = philox4x32_10([lo, hi, episode, np.full(B, e, np.uint32)], [k0, k1])
        pos[e, 0] = (-r + (2.0 * r) * _u01(w[0], w[1])).astype(dtype)
        pos[e, 1] = (-r + (2.0 * r) * _u01(w[2], w[3])).astype(dtype)
    return pos


# --------------------------------------------------------------------------------------
# In-kernel u_noise / c_noise normals (MPE_FLAG_DEVICE_NOISE; this build's own spec, no reference counterpart: the
# reference draws np.random.randn from the global stream, core.py:147-148,229).  Restated operation by operation from
# marl_amd/csrc/mpe_device.h:normal_pair -- only IEEE +, -, *, /, sqrt on float64, no fused operations -- so the f64
# kernels' normals are reproduced BIT FOR BIT.
# --------------------------------------------------------------------------------------
NOISE_WORD_U, NOISE_WORD_C = 0x40000000, 0x60000000


def normal_pair(w):
    """Box-Muller on four Philox words (uint32 arrays) -> (z0, z1) float64."""
    f8 = np.float64
    u1 = ((w[0] >> np.uint32(5)).astype(f8) * 67108864.0 + (w[1] >> np.uint32(6)).astype(f8) + 1.0) * 2.0 ** -53
    u2 = ((w[2] >> np.uint32(5)).astype(f8) * 67108864.0 + (w[3] >> np.uint32(6)).astype(f8)) * 2.0 ** -53
    m, e = np.frexp(u1)                                   # exact: m in [0.5, 1)
    low = m < 0.70710678118654757
    m = np.where(low, m * 2.0, m)
    e = np.where(low, e - 1, e)
    s = (m - 1.0) / (m + 1.0)
    z = s * s
    P = np.full_like(z, 1.0 / 21.0)
    for c in (19.0, 17.0, 15.0, 13.0, 11.0, 9.0, 7.0, 5.0, 3.0):
        P = P * z + 1.0 / c
    P = P * z + 1.0
    lnu = e.astype(f8) * 0.69314718055994529 + 2.0 * (s * P)
    with np.errstate(invalid='ignore'):
        r = np.sqrt(-2.0 * lnu)
    t = u2 * 8.0
    k = t.astype(np.int64)                                # octant 0..7
    f = t - k.astype(f8)
    g = np.where((k & 1) == 1, 1.0 - f, f)
    a = g * 0.78539816339744828
    a2 = a * a
    S = np.full_like(a, -1.0 / 355687428096000.0)
    for sign, c in ((+1, 1307674368000.0), (-1, 6227020800.0), (+1, 39916800.0), (-1, 362880.0), (+1, 5040.0),
                    (-1, 120.0), (+1, 6.0)):
        S = S * a2 + 1.0 / c if sign > 0 else S * a2 - 1.0 / c
    S = a - a * (a2 * S)
    C = np.full_like(a, -1.0 / 6402373705728000.0)
    for sign, c in ((+1, 20922789888000.0), (-1, 87178291200.0), (+1, 479001600.0), (-1, 3628800.0), (+1, 40320.0),
                    (-1, 720.0), (+1, 24.0)):
        C = C * a2 + 1.0 / c if sign > 0 else C * a2 - 1.0 / c
    C = C * a2 - 0.5
    C = C * a2 + 1.0
    swap = (((k + 1) >> 1) & 1) == 1
    c0, s0 = np.where(swap, S, C), np.where(swap, C, S)
    z0 = r * np.where((((k + 2) >> 2) & 1) == 1, -c0, c0)
    z1 = r * np.where(k >= 4, -s0, s0)
    return z0, z1


def device_normals(seed: int, step: int, env_ids, word: int):
    """(z0, z1) for the global env ids `env_ids` at step counter `step` and stream word `word`:
    Philox counter (env_lo, env_hi, step, word), key = (seed_lo, seed_hi)."""
    env_ids = np.asarray(env_ids, dtype=np.uint64)
    B = env_ids.shape[0]
    lo = (env_ids & _MASK).astype(np.uint32)
    hi = (env_ids >> np.uint64(32)).astype(np.uint32)
    k0 = np.full(B, seed & 0xFFFFFFFF, np.uint32)
    k1 = np.full(B, (seed >> 32) & 0xFFFFFFFF, np.uint32)
    w = philox4x32_10([lo, hi, np.full(B, step & 0xFFFFFFFF, np.uint32), np.full(B, word, np.uint32)], [k0, k1])
    return normal_pair(w)


def device_u_noise(spec: ScenarioSpec, seed: int, step: int, env_ids):
    """[A, 2, B] standard normals the step kernels draw for u_noise (zeros for agents without one): pass as `noise`."""
    out = np.zeros((spec.n_agents, 2, len(env_ids)))
    for a in range(spec.n_agents):
        if spec.u_noise[a]:
            out[a, 0], out[a, 1] = device_normals(seed, step, env_ids, NOISE_WORD_U | a)
    return out


def device_c_noise(spec: ScenarioSpec, seed: int, step: int, env_ids):
    """[A, dim_c, B] standard normals the step kernels draw for c_noise: pass as `cnoise`."""
    out = np.zeros((spec.n_agents, spec.dim_c, len(env_ids)))
    for a in range(spec.n_agents):
        if spec.c_noise[a] and not spec.silent[a]:
            for j in range((spec.dim_c + 1) // 2):
                z0, z1 = device_normals(seed, step, env_ids, NOISE_WORD_C | (a << 8) | j)
                out[a, 2 * j] = z0
                if 2 * j + 1 < spec.dim_c:
                    out[a, 2 * j + 1] = z1
    return out
