"""Keep-away (reference: build/lib/multiagent/scenarios/simple_push.py; this fork's copy has 10
agents -- 9 adversaries first -- and 1 landmark).  The good agent's observation carries its colour
and the landmark colours (simple_push.py:43-56,101).  No benchmark_data in the reference."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def __init__(self, num_envs=1, device=None, dtype=None, seed=0, num_agents=10, num_adversaries=9,
                 num_landmarks=1):
        super(Scenario, self).__init__(num_envs, device, dtype, seed)
        self.num_agents = num_agents
        self.num_adversaries = num_adversaries
        self.num_landmarks = num_landmarks

    def make_world(self):
        world = World()
        world.dim_c = 2
        world.agents = [Agent() for _ in range(self.num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.silent = True
            agent.adversary = i < self.num_adversaries
        world.landmarks = [Landmark() for _ in range(self.num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.hip_kind = _hip.MPE_SIMPLE_PUSH
        world.agent_pos_range = 1.0        # simple_push.py:59
        world.landmark_pos_range = 1.0     # simple_push.py:63
        return self._finish_world(world)

    @property
    def benchmark_data(self):
        raise AttributeError("'Scenario' object has no attribute 'benchmark_data' (simple_push.py defines none)")
