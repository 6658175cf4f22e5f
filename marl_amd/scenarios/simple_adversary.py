"""Physical deception (reference: build/lib/multiagent/scenarios/simple_adversary.py): 1 adversary
first, N-1 good agents, N-1 landmarks, one of which is the per-env goal (``world.goal``, drawn at
reset like ``np.random.choice(world.landmarks)``, simple_adversary.py:45-48).

``benchmark_data`` (simple_adversary.py:58-67): the adversary gets its squared distance to the goal
(``[B]``); a good agent gets the tuple of squared distances to every landmark plus the goal."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def __init__(self, num_envs=1, device=None, dtype=None, seed=0, num_agents=3):
        super(Scenario, self).__init__(num_envs, device, dtype, seed)
        self.num_agents = num_agents

    def make_world(self):
        world = World()
        world.dim_c = 2
        num_agents = self.num_agents
        world.num_agents = num_agents
        num_adversaries = 1
        num_landmarks = num_agents - 1
        world.agents = [Agent() for _ in range(num_agents)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.silent = True
            agent.adversary = i < num_adversaries
            agent.size = 0.15
        world.landmarks = [Landmark() for _ in range(num_landmarks)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.08
        world.hip_kind = _hip.MPE_SIMPLE_ADVERSARY
        world.agent_pos_range = 1.0        # simple_adversary.py:51
        world.landmark_pos_range = 1.0     # simple_adversary.py:55
        return self._finish_world(world)

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def _format_info(self, rows, agent):
        if agent.adversary:
            return rows[0]
        return tuple(rows[k] for k in range(rows.shape[0]))
