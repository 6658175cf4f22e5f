"""Helpers for the GPU parity tests: build mpe_scenario_desc from an oracle ScenarioSpec and call
the C ABI (libmpe_hip.so) directly on torch device tensors."""
import numpy as np
import torch

from marl_amd import _hip
from oracle import mpe_oracle as O

lib = _hip.lib


from tests.gpu_util_host import desc_from_spec  # noqa: E402,F401


def dev(x, dtype):
    return torch.as_tensor(np.ascontiguousarray(x), dtype=dtype).cuda()


class HipSim:
    """State + output buffers for B envs of one scenario; thin wrapper over the C ABI."""

    def __init__(self, spec, B, dtype=torch.float64, ld=None):
        self.spec, self.B, self.dtype = spec, B, dtype
        self.ld = B if ld is None else ld
        self.desc = desc_from_spec(spec)
        self.suf = _hip.suffix(dtype)
        A, E = spec.n_agents, spec.n_entities
        z = dict(dtype=dtype, device='cuda')
        self.pos = torch.zeros(E, 2, self.ld, **z)
        self.vel = torch.zeros(A, 2, self.ld, **z)
        self.obs = torch.full((sum(spec.obs_dims), self.ld), float('nan'), **z)
        self.rew = torch.full((A, self.ld), float('nan'), **z)
        self.info = torch.full((A, spec.info_width, self.ld), float('nan'), **z)
        self.goal = torch.zeros(self.ld, dtype=torch.int32, device='cuda')
        self.stream = torch.cuda.current_stream().cuda_stream

    def set_state(self, pos, vel, goal=None):
        self.pos[:, :, :self.B] = dev(pos, self.dtype)
        self.vel[:, :, :self.B] = dev(vel, self.dtype)
        if goal is not None:
            self.goal[:self.B] = torch.as_tensor(np.maximum(np.asarray(goal), 0), dtype=torch.int32).cuda()

    def _p(self, t):
        return None if t is None else t.data_ptr()

    def env_step(self, act, noise=None, stats=None, want_info=True):
        self._act, self._noise = act, noise   # keep alive until the stream has consumed them
        rc = getattr(lib, 'mpe_env_step_' + self.suf)(
            self.desc, self.B, self.ld, act.data_ptr(), self._p(noise), self.pos.data_ptr(), self.vel.data_ptr(),
            None, self.goal.data_ptr(), self.obs.data_ptr(), self.rew.data_ptr(),
            self.info.data_ptr() if (want_info and self.info.numel()) else None, self._p(stats), self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def world_step(self, act, noise=None, comm=None):
        self._act, self._noise = act, noise
        rc = getattr(lib, 'mpe_world_step_' + self.suf)(
            self.desc, self.B, self.ld, act.data_ptr(), self._p(noise), self.pos.data_ptr(), self.vel.data_ptr(),
            self._p(comm), self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def observe(self, obs=True, rew=True, info=True):
        rc = getattr(lib, 'mpe_observe_' + self.suf)(
            self.desc, self.B, self.ld, self.pos.data_ptr(), self.vel.data_ptr(), self.goal.data_ptr(),
            self.obs.data_ptr() if obs else None, self.rew.data_ptr() if rew else None,
            self.info.data_ptr() if (info and self.info.numel()) else None, self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def reset(self, seed, episode, env_id0=0, mask=None):
        rc = getattr(lib, 'mpe_reset_' + self.suf)(
            self.desc, self.B, self.ld, seed, episode, env_id0, self._p(mask), self.pos.data_ptr(),
            self.vel.data_ptr(), None, self.goal.data_ptr(), self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def host(self):
        """(pos, vel, obs, rew, info) as float64 numpy, first B envs."""
        B = self.B
        return tuple(t[..., :B].double().cpu().numpy() for t in (self.pos, self.vel, self.obs, self.rew, self.info))


def golden_soa(g, key, t=None):
    """[S, ...] (or [S, T, ...] at step t) golden array -> SoA [..., S]."""
    a = g[key] if t is None else g[key][:, t]
    return np.ascontiguousarray(np.moveaxis(a, 0, -1))
