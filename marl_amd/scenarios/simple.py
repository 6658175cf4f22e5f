"""Single agent, single landmark (reference: build/lib/multiagent/scenarios/simple.py).
No benchmark_data in the reference (make_env(..., benchmark=True) fails there too)."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def make_world(self):
        world = World()
        world.agents = [Agent() for _ in range(1)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.silent = True
        world.landmarks = [Landmark() for _ in range(1)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.hip_kind = _hip.MPE_SIMPLE
        world.agent_pos_range = 1.0        # simple.py:34
        world.landmark_pos_range = 1.0     # simple.py:38
        return self._finish_world(world)

    @property
    def benchmark_data(self):
        raise AttributeError("'Scenario' object has no attribute 'benchmark_data' (simple.py defines none)")
