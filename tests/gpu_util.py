"""Helpers for the GPU parity tests: build mpe_scenario_desc from an oracle ScenarioSpec and call
the C ABI (libmpe_hip.so) directly on torch device tensors."""
import numpy as np
import torch

from marl_amd import _hip
from oracle import mpe_oracle as O

lib = _hip.lib


from tests.gpu_util_host import desc_from_spec  # noqa: E402,F401


class _Mode:
    flags = 0               # mpe_buffers.flags on every HipSim call (set by the tests' kernel-selection fixtures)


def set_call_flags(flags):
    """Per-call kernel-selection flags (include/mpe_hip.h MPE_FLAG_*): FORCE_TILE routes the stock worlds through the
    generic LDS-tile kernel; WIDE / SPLIT pick the one-env-per-lane / one-agent-per-wavefront register kernel of
    simple_tag and simple_spread whatever the batch size."""
    _Mode.flags = int(flags)


def set_force_tile(on):
    set_call_flags(_hip.FLAG_FORCE_TILE if on else 0)


def call_flags():
    return _Mode.flags


def stats_rows(desc, B, dtype_bytes):
    return lib.mpe_stats_rows(desc, B, dtype_bytes, call_flags())


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
        self.goal = torch.zeros(max(spec.goal_rows, 1), self.ld, dtype=torch.int32, device='cuda')
        self.lforce = torch.zeros(A, 2, self.ld, **z) if spec.has_leader_force else None
        self.comm = torch.full((A, max(spec.dim_c, 1), self.ld), float('nan'), **z)[:, :spec.dim_c]
        self.lvel = torch.zeros(E - A, 2, self.ld, **z) if any(spec.movable[A:]) else None
        self.stream = torch.cuda.current_stream().cuda_stream

    def set_state(self, pos, vel, goal=None, lforce=None):
        self.pos[:, :, :self.B] = dev(pos, self.dtype)
        self.vel[:, :, :self.B] = dev(vel, self.dtype)
        if goal is not None:
            self.goal[:, :self.B] = torch.as_tensor(np.maximum(np.asarray(goal), 0), dtype=torch.int32).cuda().reshape(-1, self.B)
        if lforce is not None:
            self.lforce[:, :, :self.B] = dev(lforce, self.dtype)

    def _p(self, t):
        return None if t is None else t.data_ptr()

    def buffers(self, device_noise=None, **kw):
        """mpe_buffers with the state members set, plus the given tensors.  device_noise = (seed, step, env_id0[,
        int32 device tensor added to step]): MPE_FLAG_DEVICE_NOISE with those key words."""
        b = _hip.Buffers(flags=call_flags(), pos=self.pos, vel=self.vel, lforce=self.lforce, goal=self.goal,
                         lvel=self.lvel, **kw)
        if device_noise is not None:
            b.flags |= _hip.FLAG_DEVICE_NOISE
            b.noise_seed, b.noise_step, b.env_id0 = device_noise[:3]
            if len(device_noise) > 3 and device_noise[3] is not None:
                b.noise_step_dev = device_noise[3].data_ptr()
        return b

    def env_step(self, act, noise=None, stats=None, want_info=True, comm=None, cnoise=None, device_noise=None):
        self._act, self._noise, self._cnoise = act, noise, cnoise   # keep alive until the stream has consumed them
        b = self.buffers(device_noise, act=act, noise=noise, comm=comm, cnoise=cnoise, obs=self.obs, rew=self.rew,
                         stats=stats, info=self.info if (want_info and self.info.numel()) else None)
        rc = getattr(lib, 'mpe_env_step_' + self.suf)(self.desc, self.B, self.ld, b, self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def world_step(self, act, noise=None, comm=None, cnoise=None, device_noise=None):
        self._act, self._noise, self._cnoise = act, noise, cnoise
        b = self.buffers(device_noise, act=act, noise=noise, comm=comm, cnoise=cnoise)
        rc = getattr(lib, 'mpe_world_step_' + self.suf)(self.desc, self.B, self.ld, b, self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def observe(self, obs=True, rew=True, info=True, comm=None):
        b = self.buffers(comm=comm, obs=self.obs if obs else None, rew=self.rew if rew else None,
                         info=self.info if (info and self.info.numel()) else None)
        rc = getattr(lib, 'mpe_observe_' + self.suf)(self.desc, self.B, self.ld, b, self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def reset(self, seed, episode, env_id0=0, mask=None):
        b = self.buffers()
        rc = getattr(lib, 'mpe_reset_' + self.suf)(self.desc, self.B, self.ld, seed, episode, env_id0,
                                                   self._p(mask), b, self.stream)
        assert rc == 0, lib.mpe_strerror(rc)

    def host(self):
        """(pos, vel, obs, rew, info) as float64 numpy, first B envs."""
        B = self.B
        return tuple(t[..., :B].double().cpu().numpy() for t in (self.pos, self.vel, self.obs, self.rew, self.info))


def golden_soa(g, key, t=None):
    """[S, ...] (or [S, T, ...] at step t) golden array -> SoA [..., S]."""
    a = g[key] if t is None else g[key][:, t]
    return np.ascontiguousarray(np.moveaxis(a, 0, -1))
