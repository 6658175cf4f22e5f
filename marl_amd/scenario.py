"""Scenario plug-in base class (reference: multiagent/scenario.py:4-10)."""


class BaseScenario(object):
    """Same callbacks as the reference; built-in scenarios additionally carry the batch geometry
    (num_envs, device, dtype) they allocate the world with."""

    def __init__(self, num_envs=1, device=None, dtype=None, seed=0):
        import torch
        self.num_envs = int(num_envs)
        self.device = device
        self.dtype = torch.float64 if dtype is None else dtype
        self.seed = int(seed)

    # create elements of the world
    def make_world(self):
        raise NotImplementedError()

    # create initial conditions of the world
    def reset_world(self, world):
        raise NotImplementedError()


class HipScenario(BaseScenario):
    """Common part of the built-in scenarios: every callback of the reference's plug-in API
    (make_env.py:40-43) is served from one HIP gather launch over all agents and envs
    (libmpe_hip.so: mpe_observe_*), cached until the world state changes."""

    def _finish_world(self, world):
        world.seed = self.seed
        world.allocate(self.num_envs, self.device, self.dtype)
        self.reset_world(world)
        return world

    # simple_tag.py:39-54 / simple_spread.py:30-44
    def reset_world(self, world, mask=None):
        from . import _hip
        d = world.descriptor()
        fn = getattr(_hip.lib, 'mpe_reset_' + _hip.suffix(world.dtype))
        # agent.state.c = zeros(dim_c) (simple_tag.py:51): the rows of silent agents hold zeros already (World.allocate,
        # World._clean_silent_comm), so only a world in which somebody speaks has anything to clear
        world._clean_silent_comm()
        b = _hip.Buffers(pos=world.pos, vel=world.vel, comm=world._comm_arg, lvel=world.lvel,
                         goal=world.goal if world.hip_kind in _hip.GOAL_KINDS else None)
        with _hip.on_device(world.device):
            _hip.check(fn(d, world.num_envs, world.ld, world.seed, world.episode, world.env_id0,
                          _hip.ptr(mask), b, _hip.stream_of(world.device)), 'mpe_reset')
        world.episode += 1
        world.mark_dirty()

    def observation(self, agent, world):
        world.gather_obs()
        return world.obs_views[agent._index]

    def reward(self, agent, world):
        world.gather()
        return world.rew[agent._index]

    def benchmark_data(self, agent, world):
        world.gather()
        return self._format_info(world.info[agent._index], agent)

    def _format_info(self, rows, agent):
        raise NotImplementedError()
