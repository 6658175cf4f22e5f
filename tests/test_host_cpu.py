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
    assert len(names) >= 23 and 'mpe_target_move_f64' in names and 'mpe_env_step_f32' in names and 'mpe_reset_f64' in names
    raw = ctypes.CDLL(_hip._LIB_PATH)
    for n in names:
        assert hasattr(raw, n), 'libmpe_hip.so does not export %s' % n
    assert _hip.lib.mpe_abi_version() == _hip.ABI_VERSION == 12
    assert _hip.lib.mpe_desc_sizeof() == ctypes.sizeof(_hip.ScenarioDesc)
    assert _hip.lib.mpe_buffers_sizeof() == ctypes.sizeof(_hip.Buffers)


def _desc(spec):
    from tests.gpu_util_host import desc_from_spec
    return desc_from_spec(spec)


def _hier_spec(name='h_S1L2F2_10Targets_wide'):
    return O.hier_spec_from_golden(np.load(os.path.join(ROOT, 'tests', 'golden', name + '.npz')))


@pytest.mark.parametrize('mk', [lambda: O.simple_tag_spec(), lambda: O.simple_spread_spec(),
                                lambda: O.simple_tag_spec(48, 16, 32), lambda: O.simple_spread_spec(5, 2),
                                lambda: O.simple_spec(), lambda: O.simple_adversary_spec(),
                                lambda: O.simple_adversary_spec(5), lambda: O.simple_push_spec(),
                                lambda: O.simple_push_spec(4, 2, 2), _hier_spec,
                                lambda: _hier_spec('h_L1F1_10Targets_wide'),
                                lambda: _hier_spec('h_L1F4_3Targets_wide')])
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
    assert _hip.lib.mpe_stats_rows(d, 1000, 4, 0) >= 1
    # action rows: 5 physical entries + dim_c comm entries for non-silent agents (environment.py:42-76)
    adims = [_hip.lib.mpe_act_dim(d, i) for i in range(spec.n_agents)]
    assert adims == spec.act_dims
    assert _hip.lib.mpe_act_total(d) == sum(adims)
    assert [_hip.lib.mpe_act_row0(d, i) for i in range(spec.n_agents)] == list(np.cumsum([0] + adims[:-1]))


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
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.kind = 11
    assert lib.mpe_desc_check(bad) == -5
    ok = _hip.ScenarioDesc.from_buffer_copy(d); ok.movable[4] = 1          # a movable landmark: stock silent kinds only
    assert lib.mpe_desc_check(ok) == 0
    bad = _hip.ScenarioDesc.from_buffer_copy(ok); bad.lm_max_speed[4] = float('nan')
    assert lib.mpe_desc_check(bad) == -2
    hd = _desc(_hier_spec()); hd.movable[hd.n_agents] = 1
    assert lib.mpe_desc_check(hd) == -5
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.contact_margin = 0.0
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.mass[1] = 0.0
    assert lib.mpe_desc_check(bad) == -2
    assert lib.mpe_obs_dim(d, 4) == -6
    # launch entry points: validation happens first, so these return without touching a device
    def buf(**kw):
        b = _hip.Buffers()
        for k, v in kw.items():
            setattr(b, k, v)
        return b
    ok = dict(act=8, pos=8, vel=8)
    assert lib.mpe_env_step_f32(d, 0, 0, buf(**ok), None) == -3
    assert lib.mpe_env_step_f32(d, 4, 2, buf(**ok), None) == -3
    assert lib.mpe_env_step_f32(d, 4, 4, None, None) == -1
    assert lib.mpe_env_step_f64(d, 4, 4, buf(pos=8, vel=8), None) == -1
    assert lib.mpe_env_step_f64(d, 4, 4, buf(act=8, pos=12, vel=8), None) == -4
    assert lib.mpe_world_step_f32(d, 4, 4, buf(act=8, vel=8), None) == -1
    assert lib.mpe_observe_f32(d, 4, 4, buf(pos=8, vel=8), None) == -1
    assert lib.mpe_reset_f32(d, 4, 4, 0, 0, 0, None, buf(vel=8), None) == -1
    assert lib.mpe_target_move_f32(d, 4, 4, 0, 0, 0, buf(), None, None) == -1
    assert lib.mpe_target_move_f64(d, 4, 4, 0, 0, 0, buf(pos=12), None, None) == -4
    # goal scenarios need the goal pointer; scenarios without benchmark_data ignore `info`
    da = _desc(O.simple_adversary_spec())
    assert lib.mpe_env_step_f32(da, 4, 4, buf(obs=8, rew=8, **ok), None) == -1
    assert lib.mpe_observe_f64(da, 4, 4, buf(pos=8, vel=8, obs=8), None) == -1
    ds = _desc(O.simple_spec())
    assert lib.mpe_observe_f64(ds, 4, 4, buf(pos=8, vel=8, info=8), None) == -1     # only info asked: nothing to do
    # a command hierarchy needs the leader_force buffer
    dh = _desc(_hier_spec())
    assert lib.mpe_world_step_f64(dh, 4, 4, buf(**ok), None) == -1
    bad = _hip.ScenarioDesc.from_buffer_copy(dh); bad.reward_variant = 2
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(dh); bad.obs_r[0] = -1.0
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(dh); bad.forced_id[0] = -2
    assert lib.mpe_desc_check(bad) == -2
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.silent[0] = 0                  # comm actions: hierarchy only
    assert lib.mpe_desc_check(bad) == -5
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
    md = spaces.MultiDiscrete([[0, 4], [0, 6]])                    # environment.py:69 for a leader
    assert md.shape == 2 and md.n_entries == 12 and md == spaces.MultiDiscrete([[0, 4], [0, 6]])
    assert repr(md) == 'MultiDiscrete2'


GOLDEN_HIER = ['L1F2_3Targets_100observe', 'S1L2F2_3Targets_100observe', 'L1F2_3Targets_075observe',
               'L1F1_10Targets_wide', 'L1F4_3Targets_wide', 'S1L2F2_10Targets_wide',
               'S1L2F2_10Targets_wide_rewardFixed', 'L1F4_MovingTarget', 'S1L2F4_MovingTarget',
               # round 2: the other nine files -- every one of the reference's 18 scenario files now has a fixture
               'L1F2_10Targets_wide', 'S1L1F1_10Targets_wide', 'L1F2_3Targets_090observe', 'L1F2_3Targets_095observe',
               'L1F2_3Targets_105observe', 'S1L2F2_3Targets_075observe', 'S1L2F2_3Targets_090observe',
               'S1L2F2_3Targets_095observe', 'S1L2F2_3Targets_105observe']


@pytest.mark.parametrize('name', GOLDEN_HIER)
def test_hierarchy_presets_match_the_reference_worlds(name):
    """The parametric scenario's preset reproduces the world the reference file builds: compared
    with the parameters the golden generator read back from the reference's make_world()."""
    import marl_amd.scenarios as scenarios
    from marl_amd import _hip
    moving = name.endswith('MovingTarget')
    g = np.load(os.path.join(ROOT, 'tests', 'golden', '%s_%s.npz' % ('m' if moving else 'h', name)))
    spec = O.hier_spec_from_golden(g)
    sc = scenarios.load(name + '.py').Scenario(observation_radius=1.0)
    sc._finish_world = lambda world: world              # no GPU here: skip allocation
    world = sc.make_world()
    d = world.descriptor()
    want = _desc(spec)
    assert bytes(d) == bytes(want)
    assert [a.silent for a in world.agents] == spec.silent
    assert d.kind == (_hip.MPE_HIER_MOVING if moving else _hip.MPE_HIER_PURSUIT)
    assert d.dim_c == (3 if name == 'L1F4_MovingTarget' else 7)
    assert [lib_dim for lib_dim in (_hip.lib.mpe_obs_dim(d, i) for i in range(spec.n_agents))] == spec.obs_dims


def test_hierarchy_presets_cover_every_reference_file():
    from marl_amd.scenarios import hierarchy
    ref = ['L1F1_10Targets_wide', 'L1F2_10Targets_wide', 'L1F4_3Targets_wide', 'S1L1F1_10Targets_wide',
           'S1L2F2_10Targets_wide', 'S1L2F2_10Targets_wide_rewardFixed']
    ref += ['%s_3Targets_%sobserve' % (f, z) for f in ('L1F2', 'S1L2F2') for z in ('075', '090', '095', '100', '105')]
    ref += ['L1F4_MovingTarget', 'S1L2F4_MovingTarget']
    import glob
    live = sorted(os.path.basename(f)[:-3] for f in glob.glob('/root/reference/multiagent-particle-envs/multiagent/scenarios/[LS]*.py'))
    assert not live or live == sorted(ref)                  # the reference tree exists in the build container only
    assert sorted(hierarchy.PRESETS) == sorted(ref) == sorted(GOLDEN_HIER)   # one reference-generated fixture per file
    for name in ref:
        sc = hierarchy.bound_module(name).Scenario()
        sc._finish_world = lambda world: world
        w = sc.make_world()
        assert w.descriptor().n_agents == sum(hierarchy.PRESETS[name][k] for k in ('n_super', 'n_leader', 'n_follower', 'n_good'))
    with pytest.raises(TypeError):
        hierarchy.Scenario(bogus=1)


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


def test_comm_scenarios_geometry_and_validation():
    """simple_speaker_listener / simple_reference: ragged action rows (a fixed speaker has no physical
    entries), goal rows, and the descriptor checks that pin the stock roles."""
    from marl_amd import _hip, spaces
    import marl_amd.scenarios as scenarios
    lib = _hip.lib
    for spec in (O.speaker_listener_spec(), O.simple_reference_spec()):
        d = _desc(spec)
        assert [lib.mpe_act_dim(d, i) for i in range(2)] == spec.act_dims
        assert lib.mpe_act_total(d) == sum(spec.act_dims) and lib.mpe_goal_rows(d) == spec.goal_rows
        assert [lib.mpe_obs_dim(d, i) for i in range(2)] == spec.obs_dims and lib.mpe_info_width(d) == 0
    assert O.speaker_listener_spec().act_dims == [3, 5] and O.simple_reference_spec().act_dims == [15, 15]
    assert lib.mpe_goal_rows(_desc(O.simple_tag_spec())) == 0 and lib.mpe_goal_rows(_desc(O.simple_push_spec())) == 1
    d = _desc(O.speaker_listener_spec())
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.movable[0] = 1
    assert lib.mpe_desc_check(bad) == -2                      # the speaker does not move
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.n_landmarks = 4
    assert lib.mpe_desc_check(bad) == -2                      # three coloured landmarks
    bad = _hip.ScenarioDesc.from_buffer_copy(d); bad.silent[0] = 1
    assert lib.mpe_desc_check(bad) == -5                      # an agent with no action at all
    b = _hip.Buffers()
    b.pos, b.vel, b.obs, b.goal = 8, 8, 8, 8
    assert lib.mpe_observe_f64(d, 4, 4, b, None) == -1        # observing needs agent.state.c (comm)
    for name, kind in (('simple_speaker_listener', _hip.MPE_SPEAKER_LISTENER), ('simple_reference', _hip.MPE_SIMPLE_REFERENCE)):
        sc = scenarios.load(name + '.py').Scenario()
        sc._finish_world = lambda world: world
        w = sc.make_world()
        spec = O.speaker_listener_spec() if kind == _hip.MPE_SPEAKER_LISTENER else O.simple_reference_spec()
        assert bytes(w.descriptor()) == bytes(_desc(spec))


def test_simple_crypto_geometry_and_validation():
    """simple_crypto: nobody moves (dim_c action rows each), two goal rows (goal, key), info = (c, goal colour),
    roles pinned by the descriptor check."""
    from marl_amd import _hip
    import marl_amd.scenarios as scenarios
    lib = _hip.lib
    spec = O.simple_crypto_spec()
    d = _desc(spec)
    assert lib.mpe_desc_check(d) == 0
    assert [lib.mpe_act_dim(d, i) for i in range(3)] == spec.act_dims == [4, 4, 4]
    assert [lib.mpe_obs_dim(d, i) for i in range(3)] == spec.obs_dims == [4, 8, 8]
    assert lib.mpe_goal_rows(d) == 2 and lib.mpe_info_width(d) == spec.info_width == 8
    for field, idx, val in (('movable', 1, 1), ('silent', 0, 1), ('adversary', 1, 1), ('adversary', 0, 0)):
        bad = _hip.ScenarioDesc.from_buffer_copy(d)
        getattr(bad, field)[idx] = val
        assert lib.mpe_desc_check(bad) in (-2, -5), field
    for field, val in (('n_landmarks', 3), ('dim_c', 8), ('dim_c', 1)):
        bad = _hip.ScenarioDesc.from_buffer_copy(d)
        setattr(bad, field, val)
        assert lib.mpe_desc_check(bad) == -2, field
    b = _hip.Buffers()
    b.pos, b.vel, b.obs, b.goal = 8, 8, 8, 8
    assert lib.mpe_observe_f64(d, 4, 4, b, None) == -1        # observing needs agent.state.c (comm)
    sc = scenarios.load('simple_crypto.py').Scenario()
    sc._finish_world = lambda world: world
    w = sc.make_world()
    assert bytes(w.descriptor()) == bytes(d)
    assert [a.speaker for a in w.agents] == [False, False, True]
    assert sc.good_listeners(w) == [w.agents[1]] and sc.adversaries(w) == [w.agents[0]]


def test_simple_world_comm_geometry_and_validation():
    """simple_world_comm: the leader's action has 5 + dim_c rows, observations 34 / 28 wide, info width 1;
    the descriptor check pins two forests, some food and a good agent."""
    from marl_amd import _hip
    import marl_amd.scenarios as scenarios
    lib = _hip.lib
    spec = O.simple_world_comm_spec()
    d = _desc(spec)
    assert lib.mpe_desc_check(d) == 0
    assert [lib.mpe_act_dim(d, i) for i in range(6)] == spec.act_dims == [9, 5, 5, 5, 5, 5]
    assert [lib.mpe_obs_dim(d, i) for i in range(6)] == spec.obs_dims == [34] * 4 + [28] * 2
    assert lib.mpe_goal_rows(d) == 0 and lib.mpe_info_width(d) == 1 and lib.mpe_stats_width(d) == 14
    for field, val in (('n_forests', 1), ('n_food', 0), ('n_food', 4)):
        bad = _hip.ScenarioDesc.from_buffer_copy(d)
        setattr(bad, field, val)
        assert lib.mpe_desc_check(bad) == -2, field
    bad = _hip.ScenarioDesc.from_buffer_copy(d)
    bad.adversary[4] = bad.adversary[5] = 1
    assert lib.mpe_desc_check(bad) == -2                      # nobody to chase
    bad = _hip.ScenarioDesc.from_buffer_copy(_desc(O.simple_tag_spec()))
    bad.n_food = 1
    assert lib.mpe_desc_check(bad) == -2                      # food / forests belong to simple_world_comm only
    b = _hip.Buffers()
    b.pos, b.vel, b.obs = 8, 8, 8
    assert lib.mpe_observe_f64(d, 4, 4, b, None) == -1        # observing needs the leader's state.c (comm)
    sc = scenarios.load('simple_world_comm.py').Scenario()
    sc._finish_world = lambda world: world
    w = sc.make_world()
    assert bytes(w.descriptor()) == bytes(d)
    assert [l.name for l in w.landmarks] == ['landmark 0', 'food 0', 'food 1', 'forest 0', 'forest 1']


def test_scripted_action_argument_checks():
    """mpe_scripted_action_* validates before it launches (safe without a GPU)."""
    from marl_amd import _hip
    lib = _hip.lib
    d = _desc(O.simple_tag_spec())
    ok = (d, 8, 8, 3, 0, 0, 1.3, 8, 8, None, None)
    def call(**kw):
        names = ('desc', 'B', 'ld', 'agent', 'mode', 'target', 'cap', 'pos', 'act', 'point', 'stream')
        args = dict(zip(names, ok))
        args.update(kw)
        return lib.mpe_scripted_action_f64(*[args[n] for n in names])
    assert call(pos=None) == -1 and call(act=None) == -1
    assert call(agent=4) == -6 and call(agent=-1) == -6 and call(mode=2) == -6
    assert call(mode=1, target=9) == -6 and call(cap=float('nan')) == -6
    assert call(B=0) == -3 and call(ld=4) == -3
    assert call(pos=4) == -4 and call(point=12) == -4
    d2 = _desc(O.speaker_listener_spec())
    assert lib.mpe_scripted_action_f64(d2, 8, 8, 0, 0, 0, 1.3, 8, 8, None, None) == -5   # the speaker cannot move



def test_integration_md_stub_matches_the_library():
    """The ctypes stub printed in INTEGRATION.md is what a maintainer would paste: its struct layouts must be
    the library's (descriptor size, every mpe_buffers member)."""
    import ctypes as C
    from marl_amd import _hip
    src = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    ns = {'C': C}
    exec(src[src.index('class ScenarioDesc'):src.index('assert lib.mpe_desc_sizeof')], ns)
    assert C.sizeof(ns['ScenarioDesc']) == _hip.lib.mpe_desc_sizeof()
    assert [n for n, _ in ns['ScenarioDesc']._fields_] == [n for n, _ in _hip.ScenarioDesc._fields_]
    exec(src[src.index('class Buffers'):src.index('lib.mpe_env_step_f32.argtypes')], ns)
    assert [n for n, _ in ns['Buffers']._fields_] == [n for n, _ in _hip.Buffers._fields_]
    assert 'mpe_abi_version() == %d' % _hip.ABI_VERSION in src


def test_shared_force_id_between_super_leader_and_leader_is_refused():
    """core.py:163 makes agent.leader_force an alias of the super-leader's action.c[0:2]; a plain leader with the
    same force_id then mutates it in place (core.py:192).  No reference scenario does that; neither the oracle nor
    the library restates it: both refuse such a world instead of running it with other semantics."""
    import dataclasses
    from marl_amd import _hip
    spec = _hier_spec('h_S1L2F2_3Targets_100observe')
    sup = spec.advsuperleader.index(True)
    plain = [i for i in range(spec.n_agents) if spec.force_id[i] >= 0 and not spec.advsuperleader[i]][0]
    d = _desc(spec)
    assert _hip.lib.mpe_desc_check(d) == 0
    d.force_id[plain] = d.force_id[sup]
    assert _hip.lib.mpe_desc_check(d) == -5
    fid = list(spec.force_id)
    fid[plain] = fid[sup]
    with pytest.raises(ValueError):
        dataclasses.replace(spec, force_id=fid)


def test_residency_plan_of_the_hierarchy_kernels():
    """mpe_residency_plan: everything cached while the rows a step touches fit 250 MiB of the 256 MiB Infinity Cache; beyond,
    the write-only rows go first (comm, then leader_force), then the read-only landmarks, then the velocities."""
    from marl_amd import _hip
    from tests.gpu_util_host import desc_from_spec
    gd = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden')
    ten = desc_from_spec(O.hier_spec_from_golden(np.load(os.path.join(gd, 'h_S1L2F2_10Targets_wide.npz'))))
    mov = desc_from_spec(O.hier_spec_from_golden(np.load(os.path.join(gd, 'm_S1L2F4_MovingTarget.npz'))))
    plan = _hip.lib.mpe_residency_plan
    LM, VEL, LF, COMM = 1, 2, 4, 8
    ld = 1048576 + 128
    assert plan(ten, 65536, 8) == 0 and plan(ten, ld // 2, 4) == 0
    # 5 + 10 entities at 1 Mi f32: 168 MB of state + 42 MB of leader_force fit, the 21 comm rows (88 MB) on top do not
    assert plan(ten, ld // 2, 8) == COMM == plan(ten, ld, 4)
    assert plan(ten, ld, 8) == LM | LF | COMM == plan(ten, 2 * ld, 4)        # f64 1 Mi: 84 + 84 MB cached, 168 MB of targets streamed
    assert plan(ten, 2 * ld, 8) == LM | VEL | LF | COMM                      # velocities: only while the positions still fit
    assert plan(ten, 4 * ld, 8) == LM | LF | COMM
    assert plan(mov, ld // 2, 4) == 0 and plan(mov, ld, 4) == COMM           # 8 agents, no landmarks
    assert plan(mov, ld, 8) == LF | COMM                                     # 268 MB of positions + velocities stay
    assert plan(mov, 2 * ld, 8) == VEL | LF | COMM and plan(mov, 4 * ld, 8) == LF | COMM
    assert plan(ten, 0, 8) < 0 and plan(ten, ld, 2) < 0
