"""Cooperative navigation (reference: build/lib/multiagent/scenarios/simple_spread.py).

``benchmark_data`` returns the reference's 4-tuple (rew, collisions, min_dists,
occupied_landmarks) (simple_spread.py:46-62), each element a ``[B]`` tensor."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def __init__(self, num_envs=1, device=None, dtype=None, seed=0, num_agents=3, num_landmarks=3):
        super(Scenario, self).__init__(num_envs, device, dtype, seed)
        self.num_agents = num_agents
        self.num_landmarks = num_landmarks

    def make_world(self):
        world = World()
        world.dim_c = 2
        world.agents = [Agent() for _ in range(self.num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.size = 0.15
        world.landmarks = [Landmark() for _ in range(self.num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.hip_kind = _hip.MPE_SIMPLE_SPREAD
        world.agent_pos_range = 1.0        # simple_spread.py:39
        world.landmark_pos_range = 1.0     # simple_spread.py:43
        return self._finish_world(world)

    def _format_info(self, rows):
        return (rows[0], rows[1], rows[2], rows[3])
