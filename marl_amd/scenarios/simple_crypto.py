"""One speaker (Alice), one listener (Bob), one eavesdropper (Eve) (reference:
build/lib/multiagent/scenarios/simple_crypto.py).

Nobody moves and everybody speaks (``dim_c = 4``): each action is the 4-entry comm vector
(environment.py:61-76 gives ``Discrete(4)`` spaces).  ``world.goal`` is ``[2, B]`` int32: row 0 the goal
landmark every agent's ``goal_a`` points at, row 1 the landmark whose colour is the speaker's key
(simple_crypto.py:61-66; both colours are one-hots over ``dim_c``, :55-59).  ``benchmark_data`` returns
``(agent.state.c, agent.goal_a.color)`` (:78-80)."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class CryptoAgent(Agent):
    def __init__(self):
        super().__init__()
        self.key = None
        self.speaker = False


class Scenario(HipScenario):
    def make_world(self):
        world = World()
        world.dim_c = 4
        world.agents = [CryptoAgent() for _ in range(3)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.adversary = i < 1
            agent.speaker = i == 2
            agent.movable = False
        world.landmarks = [Landmark() for _ in range(2)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
        world.hip_kind = _hip.MPE_SIMPLE_CRYPTO
        world.agent_pos_range = 1.0          # simple_crypto.py:70
        world.landmark_pos_range = 1.0       # :74
        return self._finish_world(world)

    def _finish_world(self, world):
        world.seed = self.seed
        world.allocate(self.num_envs, self.device, self.dtype)
        world.set_goal_rows(2)
        self.reset_world(world)
        return world

    def good_listeners(self, world):
        return [agent for agent in world.agents if not agent.adversary and not agent.speaker]

    def good_agents(self, world):
        return [agent for agent in world.agents if not agent.adversary]

    def adversaries(self, world):
        return [agent for agent in world.agents if agent.adversary]

    def _format_info(self, rows, agent):
        c = rows.shape[0] // 2
        return (rows[:c].t(), rows[c:].t())
