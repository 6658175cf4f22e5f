"""CPU-side checks (no GPU needed): the C-ABI library loads and exports every symbol that
include/mpe_hip.h declares, host-only entry points (descriptor geometry, argument validation)
behave, and the package's host logic (spaces, scenario loader, sharding) is right."""
import ctypes
import os
import re

import numpy as np
import pytest

from oracle import mpe_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_functions():
    text = open(os.path.join(ROOT, 'include', 'mpe_hip.h')).read()
    text = re.sub(r'/\*.*?\*/', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mpe_[a-z0-9_]+)\s*\(', text)))


def test_library_exports_every_declared_symbol():
    from marl_amd import _hip
    names = _declared_functions()
    assert len(names) >= 18 and 'mpe_env_step_f32' in names and 'mpe_reset_f64' in names
    raw = ctypes.CDLL(_hip._LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'libmpe_hip.so does not export %s' % n
    assert _hip.lib.mpe_abi_version() == 2
    assert _hip.lib.mpe_desc_sizeof() == ctypes.sizeof(_hip.ScenarioDesc)


def _desc(spec):
    from tests.gpu_util_host import desc_from_spec
    return desc_from_spec(spec)


@pytest.mark.parametrize('mk', [lambda: O.simple_tag_spec(), lambda: O.simple_spread_spec(),
                                lambda: O.simple_tag_spec(48, 16, 32), lambda: O.simple_spread_spec(5, 2),
                                lambda: O.simple_spec(), lambda: O.simple_adversary_spec(),
                                lambda: O.simple_adversary_spec(5), lambda: O.simple_push_spec(),
                                lambda: O.simple_push_spec(4, 2, 2)])
def test_observation_geometry_matches_oracle(mk):
    """KAT-3 through the C ABI's host-only geometry functions."""
    from marl_amd import _hip
    spec = mk()
    d = _desc(spec)
    dims = [_hip.lib.mpe_obs_dim(d, i) for i in range(spec.n_agents)]
    assert dims == spec.obs_dims
    assert _hip.lib.mpe_obs_total(d) == sum(dims)
    assert [_hip.lib.mpe_obs_row0(d, i) for i in range(spec.n_agents)] == list(np.cumsum([0] + dims[:-1]))
    assert _hip.lib.mpe_info_width(d) == spec.info_width
    assert _hip.lib.mpe_stats_width(d) == 2 * spec.n_agents + 2
    assert _hip.lib.mpe_stats_rows(d, 1000, 4) >= 1


def test_descriptor_validation_and_error_codes():
    """Bad arguments are rejected on the host before anything is launched (safe without a GPU)."""
    from marl_amd import _hip
    lib = _hip.lib
    d = _desc(O.simple_tag_spec())
    assert lib.mpe_desc_check(d) == 0
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.n_agents = 0
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.n_landmarks = 125
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.kind = 9
    assert lib.mpe_desc_check(bad) == -5
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.movable[4] = 1
    assert lib.mpe_desc_check(bad) == -5
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.contact_margin = 0.0
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.mass[1] = 0.0
    assert lib.mpe_desc_check(bad) == -2
    assert lib.mpe_obs_dim(d, 4) == -6
    # launch entry points: validation happens first, so these return without touching a device
    assert lib.mpe_env_step_f32(d, 0, 0, 8, None, 8, 8, None, None, None, None, None, None, None) == -3
    assert lib.mpe_env_step_f32(d, 4, 2, 8, None, 8, 8, None, None, None, None, None, None, None) == -3
    assert lib.mpe_env_step_f64(d, 4, 4, None, None, 8, 8, None, None, None, None, None, None, None) == -1
    assert lib.mpe_env_step_f64(d, 4, 4, 8, None, 12, 8, None, None, None, None, None, None, None) == -4
    assert lib.mpe_world_step_f32(d, 4, 4, 8, None, None, 8, None, None) == -1
    assert lib.mpe_observe_f32(d, 4, 4, 8, 8, None, None, None, None, None) == -1
    assert lib.mpe_reset_f32(d, 4, 4, 0, 0, 0, None, None, 8, None, None, None) == -1
    # goal scenarios need the goal pointer; scenarios without benchmark_data ignore `info`
    da = _desc(O.simple_adversary_spec())
    assert lib.mpe_env_step_f32(da, 4, 4, 8, None, 8, 8, None, None, 8, 8, None, None, None) == -1
    assert lib.mpe_observe_f64(da, 4, 4, 8, 8, None, 8, None, None, None) == -1
    ds = _desc(O.simple_spec())
    assert lib.mpe_observe_f64(ds, 4, 4, 8, 8, None, None, None, 8, None) == -1     # only info asked: nothing to do
    bad = _hip.ScenarioDesc.from_buffer_copy(da); bad.adversary[1] = 1; bad.adversary[2] = 1
    assert lib.mpe_desc_check(bad) == -2                                            # no good agent
    dp = _desc(O.simple_push_spec())
    bad = _hip.ScenarioDesc.from_buffer_copy(dp); bad.n_landmarks = 3
    assert lib.mpe_desc_check(bad) == -2                                            # colour index out of range
    assert lib.mpe_stats_reduce(None, 1, 1, None, None) == -1
    assert b'NULL' in lib.mpe_strerror(-1) and b'stride' in lib.mpe_strerror(-3)
    with pytest.raises(_hip.MpeError):
        _hip.check(-2, 'unit test')


def test_spaces_and_scenario_loader():
    from marl_amd import spaces
    import marl_amd.scenarios as scenarios
    assert spaces.Discrete(5).n == 5 and spaces.Discrete(5) == spaces.Discrete(5)
    assert spaces.Box(-np.inf, np.inf, shape=(16,)).shape == (16,)
    tag = scenarios.load('simple_tag.py')
    spread = scenarios.load('simple_spread')
    for name in ('simple', 'simple_adversary', 'simple_push'):
        assert hasattr(scenarios.load(name).Scenario, 'make_world')
    assert hasattr(tag.Scenario, 'make_world') and hasattr(spread.Scenario, 'reset_world')
    for name in ('reward', 'observation', 'benchmark_data'):       # multiagent/scenario.py API + make_env.py:40-43
        assert callable(getattr(tag.Scenario, name))
    with pytest.raises(FileNotFoundError):
        scenarios.load('no_such_scenario.py')


def test_world_requires_a_gpu_and_has_no_cpu_fallback():
    import torch
    from marl_amd.core import World, Agent
    w = World()
    w.agents = [Agent()]
    with pytest.raises(RuntimeError):
        w.allocate(4, device='cpu', dtype=torch.float64)


def test_shard_range_partitions_the_batch():
    from marl_amd.distributed import shard_range
    for total, ws in ((8 << 20, 8), (1000, 3), (7, 8)):
        spans = [shard_range(total, r, ws) for r in range(ws)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
        assert max(c for _, c in spans) - min(c for _, c in spans) <= 1
