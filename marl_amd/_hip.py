"""ctypes binding of libmpe_hip.so (C ABI: include/mpe_hip.h).

This is the stub a maintainer of the reference would add under its Python API
(multiagent/environment.py:92-129, multiagent/core.py:124-140).  There is NO CPU
fallback: if the shared library is missing, importing this module raises.
"""
import ctypes as C
import os

# torch must be loaded BEFORE libmpe_hip.so: both need libamdhip64.so.7 and the process must end up
# with the HIP runtime that torch ships (streams / device pointers are shared with torch); loading
# this library first would bind the system runtime instead and torch would find no device.
import torch  # noqa: F401

MPE_MAX_AGENTS = 64
MPE_MAX_ENTITIES = 128
MPE_SIMPLE_TAG = 0
MPE_SIMPLE_SPREAD = 1
MPE_SIMPLE = 2
MPE_SIMPLE_ADVERSARY = 3
MPE_SIMPLE_PUSH = 4
MPE_HIER_PURSUIT = 5
MPE_SPEAKER_LISTENER = 6
MPE_SIMPLE_REFERENCE = 7
MPE_SIMPLE_CRYPTO = 8
MPE_WORLD_COMM = 9
MPE_HIER_MOVING = 10
GOAL_KINDS = (MPE_SIMPLE_ADVERSARY, MPE_SIMPLE_PUSH, MPE_SPEAKER_LISTENER, MPE_SIMPLE_REFERENCE, MPE_SIMPLE_CRYPTO)
ABI_VERSION = 12
FLAG_FORCE_TILE = 1
FLAG_WIDE = 2          # simple_tag / simple_spread: one env per lane at any batch size
FLAG_SPLIT = 4         # ... one agent per wavefront (the small-batch form) at any batch size
FLAG_SILENT_COMM_ZERO = 8   # silent agents' comm rows already hold zeros: the step does not rewrite them
FLAG_DEVICE_NOISE = 16      # u_noise / c_noise normals the caller does not supply are drawn inside the step kernels
SPLIT_MAX_ENVS = 32768
HIER_SPLIT_MAX_ENVS = 65536
ACTOR_AUTO, ACTOR_VALU, ACTOR_MFMA, ACTOR_MFMA_F16 = 0, 1, 2, 3

# MARL_AMD_LIB: alternative build of the same library (kernel tuning / ablation builds only)
_LIB_PATH = os.environ.get('MARL_AMD_LIB') or os.path.join(os.path.dirname(os.path.abspath(__file__)),
                                                           'libmpe_hip.so')


class ScenarioDesc(C.Structure):
    """mpe_scenario_desc (include/mpe_hip.h)."""
    _fields_ = [
        ('kind', C.c_int32), ('n_agents', C.c_int32), ('n_landmarks', C.c_int32), ('dim_c', C.c_int32),
        ('dt', C.c_double), ('damping', C.c_double), ('contact_force', C.c_double),
        ('contact_margin', C.c_double),
        ('agent_pos_range', C.c_double), ('landmark_pos_range', C.c_double),
        ('size', C.c_double * MPE_MAX_ENTITIES), ('mass', C.c_double * MPE_MAX_ENTITIES),
        ('accel', C.c_double * MPE_MAX_AGENTS), ('max_speed', C.c_double * MPE_MAX_AGENTS),
        ('u_noise', C.c_double * MPE_MAX_AGENTS),
        ('movable', C.c_uint8 * MPE_MAX_ENTITIES), ('collide', C.c_uint8 * MPE_MAX_ENTITIES),
        ('adversary', C.c_uint8 * MPE_MAX_AGENTS),
        ('silent', C.c_uint8 * MPE_MAX_AGENTS),
        ('force_id', C.c_int8 * MPE_MAX_AGENTS), ('forced_id', C.c_int8 * MPE_MAX_AGENTS),
        ('advleader', C.c_uint8 * MPE_MAX_AGENTS), ('advsuperleader', C.c_uint8 * MPE_MAX_AGENTS),
        ('reward_variant', C.c_int32), ('n_target_obs', C.c_int32),
        ('obs_r', C.c_double * MPE_MAX_AGENTS),
        ('advfollower', C.c_uint8 * MPE_MAX_AGENTS),
        ('n_food', C.c_int32), ('n_forests', C.c_int32),
        ('leader', C.c_uint8 * MPE_MAX_AGENTS),
        ('c_noise', C.c_double * MPE_MAX_AGENTS),
        ('lm_max_speed', C.c_double * MPE_MAX_ENTITIES),
    ]


class Buffers(C.Structure):
    """mpe_buffers (include/mpe_hip.h): device pointers of one batch."""
    _fields_ = [(n, C.c_void_p) for n in ('act', 'noise', 'pos', 'vel', 'comm', 'lforce', 'goal', 'obs', 'rew',
                                          'info', 'stats', 'cnoise')] + [
        ('flags', C.c_uint32), ('noise_step', C.c_uint32), ('noise_seed', C.c_uint64), ('env_id0', C.c_uint64),
        ('noise_step_dev', C.c_void_p), ('lvel', C.c_void_p)]

    def __init__(self, flags=0, **tensors):
        super().__init__()
        self.flags = flags
        self.set(**tensors)

    def set(self, **tensors):
        """Members from tensors (None -> NULL)."""
        for name, t in tensors.items():
            if name not in _BUFFER_NAMES:
                raise AttributeError('mpe_buffers has no member %r' % name)
            setattr(self, name, None if t is None else t.data_ptr())
        return self


_BUFFER_NAMES = frozenset(n for n, t in Buffers._fields_ if t is C.c_void_p)


class ReplaySeg(C.Structure):
    """mpe_replay_seg (include/mpe_hip.h): one source segment of a replay record."""
    _fields_ = [('ptr', C.c_void_p), ('rows', C.c_int32), ('row_stride', C.c_int64), ('env_stride', C.c_int64)]


class ReplayJob(C.Structure):
    """mpe_replay_job (include/mpe_hip.h): one buffer's share of a multi-buffer append."""
    _fields_ = [('capacity', C.c_int64), ('next_idx', C.c_int64), ('width', C.c_int32), ('nseg', C.c_int32),
                ('segs', C.POINTER(ReplaySeg)), ('ring', C.c_void_p)]


class StepStoreDesc(C.Structure):
    """mpe_step_store (include/mpe_hip.h): the rows of a step store, seen by one agent's gather."""
    _fields_ = [('obs', C.c_void_p), ('act', C.c_void_p), ('rew', C.c_void_p), ('done', C.c_void_p), ('live', C.c_void_p),
                ('obs_slot_stride', C.c_int64), ('act_slot_stride', C.c_int64), ('rew_slot_stride', C.c_int64),
                ('ld', C.c_int64), ('n', C.c_int64),
                ('slots', C.c_int32), ('n_live', C.c_int32), ('obs_row0', C.c_int32), ('act_row0', C.c_int32),
                ('rew_row', C.c_int32), ('pad_', C.c_int32)]


class ActorJob(C.Structure):
    """mpe_actor_job (include/mpe_hip.h): one agent's share of a multi-agent actor launch."""
    _fields_ = [('obs', C.c_void_p), ('weights', C.c_void_p), ('u', C.c_void_p), ('act', C.c_void_p),
                ('in_dim', C.c_int32), ('s0', C.c_int32), ('s1', C.c_int32), ('seed', C.c_uint64)]


class MpeError(RuntimeError):
    def __init__(self, code, where):
        self.code = code
        msg = lib.mpe_strerror(code).decode() if lib is not None else '?'
        super().__init__('%s failed: %s (code %d)' % (where, msg, code))


def _load():
    if not os.path.exists(_LIB_PATH):
        raise ImportError(
            'marl_amd: %s not found. Build it with `make -C marl_amd/csrc` (hipcc, gfx950); '
            'there is no CPU fallback.' % _LIB_PATH)
    L = C.CDLL(_LIB_PATH)
    vp, i64, u64, u32, ci = C.c_void_p, C.c_int64, C.c_uint64, C.c_uint32, C.c_int
    dp = C.POINTER(ScenarioDesc)
    L.mpe_abi_version.restype = ci
    L.mpe_abi_version.argtypes = []
    L.mpe_strerror.restype = C.c_char_p
    L.mpe_strerror.argtypes = [ci]
    bp = C.POINTER(Buffers)
    for name in ('mpe_desc_check', 'mpe_obs_total', 'mpe_info_width', 'mpe_stats_width', 'mpe_act_total', 'mpe_goal_rows'):
        f = getattr(L, name)
        f.restype, f.argtypes = ci, [dp]
    for name in ('mpe_obs_dim', 'mpe_obs_row0', 'mpe_act_dim', 'mpe_act_row0'):
        f = getattr(L, name)
        f.restype, f.argtypes = ci, [dp, ci]
    for suf in ('f32', 'f64'):
        for name in ('mpe_world_step_', 'mpe_observe_', 'mpe_env_step_'):   # desc, B, ld, buffers, stream
            f = getattr(L, name + suf)
            f.restype, f.argtypes = ci, [dp, i64, i64, bp, vp]
        f = getattr(L, 'mpe_reset_' + suf)            # desc, B, ld, seed, episode, env_id0, mask, buffers, stream
        f.restype, f.argtypes = ci, [dp, i64, i64, u64, u32, u64, vp, bp, vp]
        f = getattr(L, 'mpe_target_move_' + suf)      # desc, B, ld, seed, counter, env_id0, buffers, count, stream
        f.restype, f.argtypes = ci, [dp, i64, i64, u64, u32, u64, bp, vp, vp]
        f = getattr(L, 'mpe_scripted_action_' + suf)  # desc, B, ld, agent, mode, target, cap, pos, act, point, stream
        f.restype, f.argtypes = ci, [dp, i64, i64, ci, ci, ci, C.c_double, vp, vp, vp, vp]
        f = getattr(L, 'mpe_action_from_index_' + suf)  # desc, B, ld, ld_idx, idx, act, stream
        f.restype, f.argtypes = ci, [dp, i64, i64, i64, vp, vp, vp]
        f = getattr(L, 'mpe_action_from_force_' + suf)  # desc, B, ld, ld_u, u, force_discrete, act, stream
        f.restype, f.argtypes = ci, [dp, i64, i64, i64, vp, ci, vp, vp]
        f = getattr(L, 'mpe_action_force_discrete_' + suf)  # desc, B, ld, act, stream
        f.restype, f.argtypes = ci, [dp, i64, i64, vp, vp]
        f = getattr(L, 'mpe_replay_add_' + suf)       # capacity, next_idx, n, width, nseg, segs, ring, stream
        f.restype, f.argtypes = ci, [i64, i64, i64, ci, ci, C.POINTER(ReplaySeg), vp, vp]
        f = getattr(L, 'mpe_replay_add_multi_' + suf)  # n_jobs, jobs, n, stream
        f.restype, f.argtypes = ci, [ci, C.POINTER(ReplayJob), i64, vp]
        f = getattr(L, 'mpe_replay_gather_' + suf)    # capacity, width, idx, batch, ring, out, stream
        f.restype, f.argtypes = ci, [i64, ci, vp, i64, vp, vp, vp]
        f = getattr(L, 'mpe_replay_gather_compact_' + suf)  # capacity, od, ad, n, idx, batch, ring, ends, own, out, stream
        f.restype, f.argtypes = ci, [i64, ci, ci, i64, vp, i64, vp, vp, vp, vp, vp]
        f = getattr(L, 'mpe_replay_gather_steps_' + suf)  # store, od, ad, idx, batch, out, stream
        f.restype, f.argtypes = ci, [C.POINTER(StepStoreDesc), ci, ci, vp, i64, vp, vp]
    # B, ld_obs, ld_act, in_dim, s0, s1, obs, weights, u, seed, call_counter, env_id0, act, stream
    L.mpe_actor_forward_f32.restype = ci
    L.mpe_actor_forward_f32.argtypes = [i64, i64, i64, ci, ci, ci, vp, vp, vp, u64, u32, u64, vp, vp]
    # n_jobs, jobs, B, ld_obs, ld_act, call_counter, env_id0, kernel (ACTOR_*), stream
    L.mpe_actor_forward_multi_ctr_f32.restype = ci
    L.mpe_actor_forward_multi_ctr_f32.argtypes = [ci, C.POINTER(ActorJob), i64, i64, i64, u32, vp, u64, ci, vp]
    for suf in ('f32', 'f64'):
        f = getattr(L, 'mpe_replay_add_multi_cur_' + suf)  # n_jobs, jobs, cursors (host array of device pointers), n, stream
        f.restype = ci
        f.argtypes = [ci, C.POINTER(ReplayJob), C.POINTER(vp), i64, vp]
    L.mpe_actor_forward_multi_f32.restype = ci
    L.mpe_actor_forward_multi_f32.argtypes = [ci, C.POINTER(ActorJob), i64, i64, i64, u32, u64, ci, vp]
    L.mpe_stats_rows.restype = i64
    L.mpe_stats_rows.argtypes = [dp, i64, ci, u32]
    L.mpe_stats_reduce.restype = ci
    L.mpe_stats_reduce.argtypes = [vp, i64, ci, vp, vp]
    # desc, B, ld, buffers, dtype_bytes, stats_out, stream
    L.mpe_env_step_stats_static.restype = ci
    L.mpe_env_step_stats_static.argtypes = [dp, i64, i64, bp, ci, vp, vp]
    L.mpe_desc_sizeof.restype = i64
    L.mpe_desc_sizeof.argtypes = []
    L.mpe_buffers_sizeof.restype = i64
    L.mpe_buffers_sizeof.argtypes = []
    L.mpe_residency_plan.restype = ci
    L.mpe_residency_plan.argtypes = [C.POINTER(ScenarioDesc), i64, ci]
    if L.mpe_abi_version() != ABI_VERSION:
        raise ImportError('marl_amd: libmpe_hip.so ABI %d != expected %d' % (L.mpe_abi_version(), ABI_VERSION))
    if L.mpe_desc_sizeof() != C.sizeof(ScenarioDesc):
        raise ImportError('marl_amd: mpe_scenario_desc layout mismatch (%d vs %d bytes)'
                          % (L.mpe_desc_sizeof(), C.sizeof(ScenarioDesc)))
    if L.mpe_buffers_sizeof() != C.sizeof(Buffers):
        raise ImportError('marl_amd: mpe_buffers layout mismatch (%d vs %d bytes)' % (L.mpe_buffers_sizeof(), C.sizeof(Buffers)))
    return L


lib = None
lib = _load()


def check(code, where):
    if code != 0:
        raise MpeError(code, where)


def suffix(dtype):
    if dtype == torch.float32:
        return 'f32'
    if dtype == torch.float64:
        return 'f64'
    raise TypeError('marl_amd supports torch.float32 / torch.float64, got %r' % (dtype,))


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else t.data_ptr()


def stream_of(device):
    return torch.cuda.current_stream(device).cuda_stream


class on_device(object):
    """Makes ``device`` the current HIP device for the duration of a library call.  The C ABI launches on
    the current device (include/mpe_hip.h); a world / replay buffer / trainer created on ``cuda:1`` while
    ``cuda:0`` is current would otherwise launch on GPU 0 against GPU 1's memory.  Costs one
    ``torch.cuda.current_device()`` when the device is already current (the normal one-process-per-GPU case)."""
    __slots__ = ('idx', 'prev')

    def __init__(self, device):
        idx = torch.device(device).index
        self.idx = torch.cuda.current_device() if idx is None else idx
        self.prev = None

    def __enter__(self):
        cur = torch.cuda.current_device()
        if cur != self.idx:
            self.prev = cur
            torch.cuda.set_device(self.idx)
        return self

    def __exit__(self, *exc):
        if self.prev is not None:
            torch.cuda.set_device(self.prev)
            self.prev = None
        return False
