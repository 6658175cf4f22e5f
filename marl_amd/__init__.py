"""marl_amd -- MI355X-native batched multiagent-particle-envs hot path.

Drop-in for the reference's ``make_env`` / ``MultiAgentEnv.step()/reset()`` / ``Scenario`` API
(asamakei/marl, multiagent-particle-envs) over batches of env instances; the device work is done
by hand-written HIP kernels in ``libmpe_hip.so`` (C ABI: include/mpe_hip.h).
"""
from .make_env import make_env  # noqa: F401

__all__ = ['make_env']
