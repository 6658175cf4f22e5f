"""Speaker / listener (reference: build/lib/multiagent/scenarios/simple_speaker_listener.py).

Agent 0 (speaker) does not move and speaks: its action is the ``dim_c = 3`` comm entries only
(``Discrete(3)``, environment.py:51-63,174); agent 1 (listener) is silent and moves.  The goal landmark
(agent 0's ``goal_b``) is one int32 index per env in ``world.goal``.  ``benchmark_data`` of the
reference raises NameError (:62-64); it is not offered here."""
from marl_amd import _hip
from marl_amd.core import World, Agent, Landmark
from marl_amd.scenario import HipScenario


class Scenario(HipScenario):
    def __init__(self, observation_radius=1.0, num_envs=1, device=None, dtype=None, seed=0):
        super(Scenario, self).__init__(num_envs, device, dtype, seed)
        self.obs_r = observation_radius

    def make_world(self):
        world = World()
        world.dim_c = 3
        world.agents = [Agent() for _ in range(2)]
        for i, agent in enumerate(world.agents):
            agent.name = 'agent %d' % i
            agent.collide = False
            agent.size = 0.075
        world.agents[0].movable = False      # speaker
        world.agents[1].silent = True        # listener
        world.landmarks = [Landmark() for _ in range(3)]
        for i, landmark in enumerate(world.landmarks):
            landmark.name = 'landmark %d' % i
            landmark.collide = False
            landmark.movable = False
            landmark.size = 0.04
        world.hip_kind = _hip.MPE_SPEAKER_LISTENER
        world.agent_pos_range = 1.0          # simple_speaker_listener.py:55
        world.landmark_pos_range = 1.0       # :59
        return self._finish_world(world)

    def benchmark_data(self, agent, world):
        raise NameError("simple_speaker_listener.benchmark_data is broken in the reference (:62-64)")
