"""Predator-prey with food, forests and a speaking lead predator (reference:
build/lib/multiagent/scenarios/simple_world_comm.py).

Four adversaries chase two good agents (simple_tag roles, sizes and speeds); agent 0 is the ``leader``:
the only speaking agent (``dim_c = 4``, action = 5 physical + 4 comm entries, environment.py:61-69) and
the only one that sees everybody.  ``world.landmarks`` = 1 colliding obstacle + 2 food + 2 forests
(:56-57).  An agent inside a forest is hidden from agents outside it (:258-269); good agents are paid
for touching food (:182-185)."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def __init__(self, observation_radius=1.0, **kwargs):
        super().__init__(**kwargs)
        self.obs_r = observation_radius      # stored, never read (simple_world_comm.py:7-9)

    def make_world(self):
        world = World()
        world.dim_c = 4
        num_good_agents, num_adversaries = 2, 4
        world.agents = [Agent() for _ in range(num_adversaries + num_good_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = True
            agent.leader = i == 0
            agent.silent = i > 0
            agent.adversary = i < num_adversaries
            agent.size = 0.075 if agent.adversary else 0.045
            agent.accel = 3.0 if agent.adversary else 4.0
            agent.max_speed = 1.0 if agent.adversary else 1.3
        world.landmarks = [Landmark() for _ in range(1)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = True
            landmark.movable = False
            landmark.size = 0.2
            landmark.boundary = False
        world.food = [Landmark() for _ in range(2)]
        for i, landmark in enumerate(world.food):
            landmark.name = 'food %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.03
            landmark.boundary = False
        world.forests = [Landmark() for _ in range(2)]
        for i, landmark in enumerate(world.forests):
            landmark.name = 'forest %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.3
            landmark.boundary = False
        world.landmarks += world.food
        world.landmarks += world.forests
        world.hip_kind = _hip.MPE_WORLD_COMM
        world.agent_pos_range = 1.0          # simple_world_comm.py:106
        world.landmark_pos_range = 0.9       # :110-116
        return self._finish_world(world)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def _format_info(self, rows, agent):
        return rows[0]
