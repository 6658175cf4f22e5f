"""Predator-prey (reference: build/lib/multiagent/scenarios/simple_tag.py).

Adversaries come first in ``world.agents`` (simple_tag.py:21).  ``benchmark_data`` returns the
per-env collision count as a ``[B]`` tensor (simple_tag.py:57-66 returns a Python int per env)."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def __init__(self, num_envs=1, device=None, dtype=None, seed=0,
                 num_good_agents=1, num_adversaries=3, num_landmarks=2):
        super(Scenario, self).__init__(num_envs, device, dtype, seed)
        self.num_good_agents = num_good_agents
        self.num_adversaries = num_adversaries
        self.num_landmarks = num_landmarks

    def make_world(self):
        world = World()
        world.dim_c = 2
        num_agents = self.num_adversaries + self.num_good_agents
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.adversary = i < self.num_adversaries
            agent.size = 0.075 if agent.adversary else 0.05
            agent.accel = 3.0 if agent.adversary else 4.0
            agent.max_speed = 1.0 if agent.adversary else 1.3
        world.landmarks = [Landmark() for _ in range(self.num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = True
            landmark.movable = False
            landmark.size = 0.2
            landmark.boundary = False
        world.hip_kind = _hip.MPE_SIMPLE_TAG
        world.agent_pos_range = 1.0        # simple_tag.py:48
        world.landmark_pos_range = 0.9     # simple_tag.py:53
        return self._finish_world(world)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def _format_info(self, rows, agent):
        return rows[0]
