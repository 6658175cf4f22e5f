"""Two agents that tell each other where to go (reference:
build/lib/multiagent/scenarios/simple_reference.py).

Both agents move and speak (``dim_c = 10``): each action is the flat MultiDiscrete vector, 5 physical
+ 10 comm entries (environment.py:61-69,161-168).  Each agent's ``goal_b`` landmark is one int32 index
per env: ``world.goal`` is ``[2, B]``.  The scenario has no ``benchmark_data``."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def make_world(self):
        world = World()
        world.dim_c = 10
        world.agents = [Agent() for _ in range(2)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
        world.landmarks = [Landmark() for _ in range(3)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.hip_kind = _hip.MPE_SIMPLE_REFERENCE
        world.agent_pos_range = 1.0          # simple_reference.py:49
        world.landmark_pos_range = 1.0       # :53
        return self._finish_world(world)

    def _finish_world(self, world):
        world.seed = self.seed
        world.allocate(self.num_envs, self.device, self.dtype)
        world.set_goal_rows(len(world.agents))
        self.reset_world(world)
        return world

    @property
    def benchmark_data(self):
        raise AttributeError("'Scenario' object has no attribute 'benchmark_data' (simple_reference.py defines none)")
