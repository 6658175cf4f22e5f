"""The fork's follower / leader / super-leader pursuit family (reference:
multiagent/scenarios/L1F*_*.py and S1L*_*.py, e.g. L1F2_3Targets_100observe.py).

The reference keeps one near-identical file per configuration; they differ only in the constants
listed in PRESETS (agent counts, number of targets, zoom_rate, the leaders' observation radius,
whether sizes scale with zoom_rate, the command ids, and the rewardFixed reward).  Agents are
ordered followers, leaders, super-leader (L1F2_3Targets_100observe.py:31-70); targets are the
landmarks.  ``scenarios.load('<reference file name>')`` resolves to ``Scenario`` bound to that preset.

  * leaders (and the super-leader) are not silent: their action is MultiDiscrete, 5 physical + dim_c
    = 7 command entries (environment.py:61-69); the command pushes the agents whose ``forced_id``
    equals the leader's ``force_id`` (core.py:153-194, inside the step kernel);
  * ``benchmark_data`` returns the number of touching agents of the other team as a ``[B]`` tensor
    (L1F2_3Targets_100observe.py:136-148).

The two *_MovingTarget files (L1F4_MovingTarget.py, S1L2F4_MovingTarget.py) are presets too: no
landmarks, one good agent (the prey, last) that every adversary chases; their own observation and
reward text runs as kind MPE_HIER_MOVING (reward_variant 0 / 1 = which of the two files' text).
"""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


def _preset(n_super, n_leader, n_follower, n_landmarks, zoom, leader_obs, scale_sizes, follower_forced,
            leader_force, leader_forced, super_force=-1, reward_variant=0, n_good=0, dim_c=7, moving=False):
    return dict(n_super=n_super, n_leader=n_leader, n_follower=n_follower, n_landmarks=n_landmarks, zoom=zoom,
                leader_obs=leader_obs, scale_sizes=scale_sizes, follower_forced=follower_forced,
                leader_force=leader_force, leader_forced=leader_forced, super_force=super_force,
                reward_variant=reward_variant, n_good=n_good, dim_c=dim_c, moving=moving)


def _l1(n_follower, n_landmarks, zoom, leader_obs, scale):
    # L1F*: every follower obeys the one leader (forced_id 1 / force_id 1, :45-58)
    return _preset(0, 1, n_follower, n_landmarks, zoom, leader_obs, scale, [1] * n_follower, [1], [-1])


def _s1l2f2(n_landmarks, zoom, scale, reward_variant=0):
    # S1L2F2*: follower 0 -> leader with force_id 2 (agent 2), follower 1 -> force_id 1 (agent 3);
    # both leaders obey the super-leader (forced_id 3 / force_id 3), S1L2F2_10Targets_wide.py:45-71
    return _preset(1, 2, 2, n_landmarks, zoom, 0.4, scale, [2, 1], [2, 1], [3, 3], 3, reward_variant)


PRESETS = {
    'L1F1_10Targets_wide': _l1(1, 10, 0.5, 0.64, True),
    'L1F2_10Targets_wide': _l1(2, 10, 0.5, 0.64, True),
    'L1F2_3Targets_075observe': _l1(2, 3, 0.75, 0.75, False),
    'L1F2_3Targets_090observe': _l1(2, 3, 0.90, 0.75, False),
    'L1F2_3Targets_095observe': _l1(2, 3, 0.95, 0.75, False),
    'L1F2_3Targets_100observe': _l1(2, 3, 1, 0.75, False),
    'L1F2_3Targets_105observe': _l1(2, 3, 1.05, 0.75, False),
    'L1F4_3Targets_wide': _l1(4, 3, 1, 0.64, True),
    # S1L1F1_10Targets_wide.py:45-70: follower <- leader (1), leader <- super-leader (2)
    'S1L1F1_10Targets_wide': _preset(1, 1, 1, 10, 0.5, 0.4, True, [1], [1], [2], 2),
    'S1L2F2_10Targets_wide': _s1l2f2(10, 0.5, True),
    'S1L2F2_10Targets_wide_rewardFixed': _s1l2f2(10, 0.5, True, reward_variant=1),
    'S1L2F2_3Targets_075observe': _s1l2f2(3, 0.75, False),
    'S1L2F2_3Targets_090observe': _s1l2f2(3, 0.90, False),
    'S1L2F2_3Targets_095observe': _s1l2f2(3, 0.95, False),
    'S1L2F2_3Targets_100observe': _s1l2f2(3, 1, False),
    'S1L2F2_3Targets_105observe': _s1l2f2(3, 1.05, False),
    # L1F4_MovingTarget.py:16-79: 3 commands, four followers obey the one leader, one prey
    'L1F4_MovingTarget': _preset(0, 1, 4, 0, 1, 0.4, False, [1] * 4, [1], [-1], n_good=1, dim_c=3, moving=True),
    # S1L2F4_MovingTarget.py:16-81: followers 0,1 <- leader with force_id 2, followers 2,3 <- force_id 1,
    # both leaders <- super-leader (forced_id 0 / force_id 0)
    'S1L2F4_MovingTarget': _preset(1, 2, 4, 0, 1, 0.4, False, [2, 2, 1, 1], [2, 1], [0, 0], 0, reward_variant=1,
                                   n_good=1, dim_c=7, moving=True),
}


class Scenario(HipScenario):
    preset = 'L1F2_3Targets_100observe'

    def __init__(self, observation_radius=1.0, num_envs=1, device=None, dtype=None, seed=0, preset=None, **overrides):
        super(Scenario, self).__init__(num_envs, device, dtype, seed)
        self.obs_r = observation_radius            # stored, never read (L1F2_3Targets_100observe.py:9-10)
        if preset is not None:
            self.preset = preset
        if self.preset not in PRESETS:
            raise KeyError('unknown hierarchy preset %r (known: %s)' % (self.preset, ', '.join(sorted(PRESETS))))
        self.params = dict(PRESETS[self.preset])
        unknown = set(overrides) - set(self.params)
        if unknown:
            raise TypeError('unknown scenario parameter(s): %s' % ', '.join(sorted(unknown)))
        self.params.update(overrides)

    def make_world(self):
        p = self.params
        world = World()
        world.dim_c = p['dim_c']                   # number of commands (:15)
        world.target_sens = 0.0
        nF, nL, nS = p['n_follower'], p['n_leader'], p['n_super']
        self.num_good, self.num_adv, self.num_landmarks = p['n_good'], nF + nL + nS, p['n_landmarks']
        zoom = p['zoom']
        world.agents = [Agent() for _ in range(self.num_adv + self.num_good)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.adversary, agent.good = True, False
            if i >= self.num_adv:                  # prey (L1F4_MovingTarget.py:69-84)
                agent.adversary, agent.good = False, True
                agent.silent = True
                agent.advfollower, agent.advleader, agent.advsuperleader = False, False, False
                agent.obs_r = 10
                agent.force_id, agent.forced_id = -1, -1
                agent.size, agent.accel, agent.max_speed = 0.05, 4.0, 1.3
                continue
            if i < nF:                             # followers (:36-46)
                agent.silent = True
                agent.advfollower, agent.advleader, agent.advsuperleader = True, False, False
                agent.obs_r = 0.2 * zoom
                agent.force_id, agent.forced_id = -1, p['follower_forced'][i]
            elif i < nF + nL:                      # leaders (:48-58)
                agent.silent = False
                agent.advfollower, agent.advleader, agent.advsuperleader = False, True, False
                agent.obs_r = p['leader_obs'] * zoom
                agent.force_id, agent.forced_id = p['leader_force'][i - nF], p['leader_forced'][i - nF]
            else:                                  # super-leader (:60-70)
                agent.silent = False
                agent.advfollower, agent.advleader, agent.advsuperleader = False, True, True
                agent.obs_r = 0.5 * zoom
                agent.force_id, agent.forced_id = p['super_force'], -1
            agent.size = 0.075 * zoom if p['scale_sizes'] else 0.075
            agent.accel = 3.0
            agent.max_speed = 1.0
        world.landmarks = [Landmark() for _ in range(self.num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = True
            landmark.movable = False
            landmark.size = 0.05 * zoom if p['scale_sizes'] else 0.05
            landmark.boundary = False
            landmark.adversary, landmark.good, landmark.silent = False, True, True
            landmark.advfollower = landmark.advleader = landmark.advsuperleader = False
        world.hip_kind = _hip.MPE_HIER_MOVING if p['moving'] else _hip.MPE_HIER_PURSUIT
        world.reward_variant = p['reward_variant']
        world.n_target_obs = 3                     # target_pos[0:3] (:251-253)
        world.agent_pos_range = 1.0                # :127
        world.landmark_pos_range = 0.9 if p['moving'] else 1.0   # :132 (L1F4_MovingTarget.py:124)
        return self._finish_world(world)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def _format_info(self, rows, agent):
        return rows[0]


def bound_module(preset):
    """A module-like object whose ``Scenario`` is bound to ``preset`` (what scenarios.load returns
    for a reference file name such as 'S1L2F2_10Targets_wide.py')."""
    import types
    mod = types.ModuleType('marl_amd.scenarios.' + preset)
    mod.Scenario = type('Scenario', (Scenario,), {'preset': preset, '__module__': mod.__name__})
    mod.__doc__ = 'hierarchy preset %s (reference: multiagent/scenarios/%s.py)' % (preset, preset)
    return mod
