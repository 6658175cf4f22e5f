// libmpe_hip.so -- C ABI (include/mpe_hip.h) over the gfx950 kernels.
// Host side: argument validation, descriptor conversion, kernel selection, launch.  Nothing here
// allocates, synchronises or keeps state.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>

#include "mpe_device.h"
#include "mpe_launch.h"
#include "mpe_actor_kernel.h"

using namespace mpe;

// by-value kernarg: 16 KiB launches verified on gfx950 (tools/kernarg_probe.hip)
static_assert(sizeof(KParams<double>) <= 8192, "kernarg segment must stay under 8 KiB");

// ------------------------------------------------------------------------------------------------
// descriptor helpers (host)
// ------------------------------------------------------------------------------------------------
// Does agent i's observation end with its leader_force?  L1F*/S1L* Targets files and L1F4_MovingTarget.py:252:
// forced_id >= 0.  S1L2F4_MovingTarget.py:254-257: hasattr(agent, 'leader_force') holds for every agent
// (core.py:156) and the agents that are neither advleader nor good return it.
static bool observes_lforce(const mpe_scenario_desc *h, int i) {
  if (h->kind == MPE_HIER_MOVING && h->reward_variant == 1) return !h->advleader[i] && h->adversary[i];
  return (h->kind == MPE_HIER_PURSUIT || h->kind == MPE_HIER_MOVING) && h->forced_id[i] >= 0;
}

static int obs_dim_of(const mpe_scenario_desc *h, int i) {
  const int A = h->n_agents, L = h->n_landmarks;
  if (h->kind == MPE_SIMPLE_TAG) {  // simple_tag.py:147
    int good_others = 0;
    for (int j = 0; j < A; ++j)
      if (j != i && !h->adversary[j]) ++good_others;
    return 4 + 2 * L + 2 * (A - 1) + 2 * good_others;
  }
  if (h->kind == MPE_HIER_PURSUIT) {  // L1F2_3Targets_100observe.py:250-253
    const int nt = L < h->n_target_obs ? L : h->n_target_obs;
    return 4 + 4 * (A - 1) + (h->forced_id[i] >= 0 ? 2 : 0) + 2 * nt;
  }
  if (h->kind == MPE_HIER_MOVING)  // L1F4_MovingTarget.py:252-258 / S1L2F4_MovingTarget.py:254-260
    return 4 + 4 * (A - 1) + (observes_lforce(h, i) ? 2 : 0);
  if (h->kind == MPE_SIMPLE_CRYPTO) return h->adversary[i] ? h->dim_c : 2 * h->dim_c;  // simple_crypto.py:153-171
  if (h->kind == MPE_WORLD_COMM) {  // simple_world_comm.py:305-313
    int good_others = 0;
    for (int j = 0; j < A; ++j)
      if (j != i && !h->adversary[j]) ++good_others;
    return 4 + 2 * L + 2 * (A - 1) + 2 * good_others + 2 + (h->adversary[i] ? h->dim_c : 0);
  }
  if (h->kind == MPE_SPEAKER_LISTENER)  // simple_speaker_listener.py:89-94
    return h->movable[i] ? 2 + 2 * L + h->dim_c * (A - 1) : 3;
  if (h->kind == MPE_SIMPLE_REFERENCE) return 2 + 2 * L + 3 + h->dim_c * (A - 1);  // simple_reference.py:89
  if (h->kind == MPE_SIMPLE) return 2 + 2 * L;  // simple.py:50
  if (h->kind == MPE_SIMPLE_ADVERSARY)          // simple_adversary.py:137-140
    return 2 * L + 2 * (A - 1) + (h->adversary[i] ? 0 : 2);
  if (h->kind == MPE_SIMPLE_PUSH)               // simple_push.py:100-104
    return 2 + 2 * L + 2 * (A - 1) + (h->adversary[i] ? 0 : 2 + 3 + 3 * L);
  return 4 + 2 * L + 2 * (A - 1) + h->dim_c * (A - 1);  // simple_spread.py:99
}

static bool kind_has_roles(int kind) {
  return kind == MPE_SIMPLE_TAG || kind == MPE_SIMPLE_ADVERSARY || kind == MPE_SIMPLE_PUSH ||
         kind == MPE_HIER_PURSUIT || kind == MPE_WORLD_COMM || kind == MPE_HIER_MOVING;
}
static bool kind_is_comm(int kind) { return kind == MPE_SPEAKER_LISTENER || kind == MPE_SIMPLE_REFERENCE; }
// kinds whose observations contain another agent's state.c (read from the comm buffer when only observing)
static bool kind_reads_comm(int kind) {
  return kind_is_comm(kind) || kind == MPE_SIMPLE_CRYPTO || kind == MPE_WORLD_COMM;
}
static bool kind_is_hier(int kind) { return kind == MPE_HIER_PURSUIT || kind == MPE_HIER_MOVING; }
static bool kind_is_ragged(int kind) { return kind_is_hier(kind) || kind_reads_comm(kind); }
// environment.py:42-76: 5 physical entries for a movable agent, dim_c comm entries for a speaking one
static int act_dim_of(const mpe_scenario_desc *h, int i) {
  return (h->movable[i] ? 5 : 0) + (h->silent[i] ? 0 : h->dim_c);
}
static int goal_rows_of(const mpe_scenario_desc *h) {
  if (h->kind == MPE_SIMPLE_ADVERSARY || h->kind == MPE_SIMPLE_PUSH || h->kind == MPE_SPEAKER_LISTENER) return 1;
  if (h->kind == MPE_SIMPLE_CRYPTO) return 2;  // goal landmark, landmark of the speaker's key
  return h->kind == MPE_SIMPLE_REFERENCE ? h->n_agents : 0;
}
// rows of agent.state.c a step writes: dim_c per speaking agent (core.py:226-230)
static int speaking_comm_rows(const mpe_scenario_desc *h) {
  int n = 0;
  for (int a = 0; a < h->n_agents; ++a) n += h->silent[a] ? 0 : h->dim_c;
  return n;
}
static bool has_lforce(const mpe_scenario_desc *h) {
  for (int a = 0; a < h->n_agents; ++a)
    if (h->forced_id[a] >= 0) return true;
  return false;
}
static bool kind_has_goal(int kind) { return kind == MPE_SIMPLE_ADVERSARY || kind == MPE_SIMPLE_PUSH; }
static int info_width_of(const mpe_scenario_desc *h) {
  switch (h->kind) {
    case MPE_SIMPLE_TAG: return 1;
    case MPE_HIER_PURSUIT: return 1;  // L1F2_3Targets_100observe.py:136-148
    case MPE_WORLD_COMM: return 1;    // simple_world_comm.py:119-127
    case MPE_HIER_MOVING: return 1;   // L1F4_MovingTarget.py:128-140
    case MPE_SIMPLE_SPREAD: return 4;
    case MPE_SIMPLE_ADVERSARY: return h->n_landmarks + 1;  // simple_adversary.py:58-67
    case MPE_SIMPLE_CRYPTO: return 2 * h->dim_c;           // (state.c, goal colour), simple_crypto.py:78-80
    default: return 0;                                     // simple.py / simple_push.py: no benchmark_data
  }
}

extern "C" int mpe_abi_version(void) { return MPE_ABI_VERSION; }
extern "C" int64_t mpe_desc_sizeof(void) { return (int64_t)sizeof(mpe_scenario_desc); }
extern "C" int64_t mpe_buffers_sizeof(void) { return (int64_t)sizeof(mpe_buffers); }

extern "C" const char *mpe_strerror(int code) {
  switch (code) {
    case MPE_OK: return "ok";
    case MPE_ERR_NULL: return "required pointer is NULL";
    case MPE_ERR_DESC: return "invalid scenario descriptor";
    case MPE_ERR_BATCH: return "bad batch size or row stride (need B >= 1, ld >= B)";
    case MPE_ERR_ALIGN: return "device pointer not aligned to its element type";
    case MPE_ERR_UNSUPPORTED:
      return "unsupported world (movable landmarks / unknown scenario kind / comm actions outside the "
             "command-hierarchy scenarios)";
    case MPE_ERR_ARG: return "bad argument";
    case MPE_ERR_SCRATCH:
      return "batch too large for the built-in scratch array of mpe_env_step_stats_static (its statistics rows x "
             "width must fit MPE_STATIC_SCRATCH_BYTES); use mpe_env_step_* with a caller-owned stats buffer";
    default: return code > 0 ? hipGetErrorString((hipError_t)code) : "unknown error";
  }
}

extern "C" int mpe_desc_check(const mpe_scenario_desc *h) {
  if (!h) return MPE_ERR_NULL;
  if (h->kind < MPE_SIMPLE_TAG || h->kind > MPE_HIER_MOVING) return MPE_ERR_UNSUPPORTED;
  if (h->n_agents < 1 || h->n_agents > MPE_MAX_AGENTS) return MPE_ERR_DESC;
  if (h->n_landmarks < 0 || h->n_agents + h->n_landmarks > MPE_MAX_ENTITIES) return MPE_ERR_DESC;
  if (h->dim_c < 0 || h->dim_c > 16) return MPE_ERR_DESC;
  if (!(h->contact_margin > 0.0) || !std::isfinite(h->dt) || !std::isfinite(h->damping) ||
      !std::isfinite(h->contact_force))
    return MPE_ERR_DESC;
  const int E = h->n_agents + h->n_landmarks;
  for (int e = 0; e < E; ++e) {
    if (!std::isfinite(h->size[e]) || !(h->mass[e] > 0.0)) return MPE_ERR_DESC;
    // a movable landmark (core.py:211-222 integrates every movable entity): the all-silent stock kinds, whose
    // gathers never look at a landmark's velocity
    if (e >= h->n_agents && h->movable[e]) {
      if (h->kind > MPE_SIMPLE_PUSH) return MPE_ERR_UNSUPPORTED;
      if (std::isnan(h->lm_max_speed[e])) return MPE_ERR_DESC;
    }
  }
  for (int a = 0; a < h->n_agents; ++a)
    if (!std::isfinite(h->accel[a]) || std::isnan(h->max_speed[a]) || !std::isfinite(h->u_noise[a]) ||
        !std::isfinite(h->c_noise[a]))
      return MPE_ERR_DESC;
  for (int a = 0; a < h->n_agents; ++a) {
    if (h->force_id[a] < -1 || h->forced_id[a] < -1) return MPE_ERR_DESC;
    // commands exist in the hierarchy scenarios only (no stock scenario sets force_id / forced_id)
    if ((h->force_id[a] >= 0 || h->forced_id[a] >= 0) && !kind_is_hier(h->kind)) return MPE_ERR_UNSUPPORTED;
    if (!(h->obs_r[a] >= 0.0) || !std::isfinite(h->obs_r[a])) return MPE_ERR_DESC;
    // comm entries are consumed by the command hierarchy only; the stock scenarios that observe
    // other.state.c (simple_spread.py:95) are built silent (simple_spread.py:17)
    if (!h->silent[a] && !kind_is_ragged(h->kind)) return MPE_ERR_UNSUPPORTED;
    // an agent that neither moves nor speaks has no action at all (the reference cannot build its
    // action space either, environment.py:76)
    if (!h->movable[a] && h->silent[a]) return MPE_ERR_UNSUPPORTED;
    // a super-leader's command is c[0:2] (core.py:163)
    if (h->advsuperleader[a] && h->force_id[a] >= 0 && !h->silent[a] && h->dim_c < 2) return MPE_ERR_DESC;
  }
  // core.py:163 makes agent.leader_force an ALIAS of the super-leader's action.c[0:2]; a plain leader with the same
  // force_id then adds its push into that array in place (core.py:192), changing the super-leader's own command for
  // every later agent and its state.c.  No scenario file of the reference shares a force_id between a super-leader
  // and a plain leader; a world that does is refused rather than run with different semantics.
  for (int a = 0; a < h->n_agents; ++a)
    for (int b = 0; b < h->n_agents; ++b)
      if (h->force_id[a] >= 0 && h->force_id[a] == h->force_id[b] && h->advsuperleader[a] && !h->advsuperleader[b])
        return MPE_ERR_UNSUPPORTED;
  if (h->kind == MPE_HIER_PURSUIT) {
    if (h->n_target_obs < 0 || h->n_target_obs > MPE_MAX_ENTITIES) return MPE_ERR_DESC;
    if (h->reward_variant != 0 && h->reward_variant != 1) return MPE_ERR_DESC;
  }
  if (h->kind == MPE_HIER_MOVING && h->reward_variant != 0 && h->reward_variant != 1) return MPE_ERR_DESC;
  if (kind_is_comm(h->kind)) {
    // two agents, three coloured landmarks (simple_speaker_listener.py:16,48-50 / simple_reference.py:11,41-43)
    if (h->n_agents != 2 || h->n_landmarks != 3 || h->dim_c < 1) return MPE_ERR_DESC;
    if (h->kind == MPE_SPEAKER_LISTENER &&
        (h->movable[0] || h->silent[0] || !h->movable[1] || !h->silent[1]))  // speaker, then listener
      return MPE_ERR_DESC;
  }
  if (h->kind == MPE_SIMPLE_CRYPTO) {
    // Eve, Bob, Alice by index (simple_crypto.py:33-35,62-63); colours are one-hots over dim_c (:55-59);
    // dim_c <= 7 keeps np.sum's left-to-right order (pairwise summation starts at 8 elements)
    if (h->n_agents != 3 || h->n_landmarks != 2 || h->dim_c < 2 || h->dim_c > 7) return MPE_ERR_DESC;
    if (!h->adversary[0] || h->adversary[1] || h->adversary[2]) return MPE_ERR_DESC;
    for (int a = 0; a < 3; ++a)
      if (h->movable[a] || h->silent[a]) return MPE_ERR_DESC;
  }
  if (h->kind == MPE_WORLD_COMM) {
    // two forests (simple_world_comm.py:240-245), at least one food (min over world.food, :185), and at
    // least one good agent (min over good agents, :196); everybody moves
    if (h->n_forests != 2 || h->n_food < 1 || h->n_food + h->n_forests > h->n_landmarks) return MPE_ERR_DESC;
    int goods = 0;
    for (int a = 0; a < h->n_agents; ++a) {
      goods += h->adversary[a] ? 0 : 1;
      if (!h->movable[a]) return MPE_ERR_DESC;
    }
    if (goods < 1) return MPE_ERR_DESC;
  } else if (h->n_food != 0 || h->n_forests != 0) {
    return MPE_ERR_DESC;
  }
  if (h->kind == MPE_SIMPLE && h->n_landmarks < 1) return MPE_ERR_DESC;
  if (kind_has_goal(h->kind)) {  // a goal landmark and at least one good agent must exist
    int goods = 0;
    for (int a = 0; a < h->n_agents; ++a) goods += h->adversary[a] ? 0 : 1;
    if (h->n_landmarks < 1 || goods < 1) return MPE_ERR_DESC;
    // simple_push colours index a 3-vector with [goal + 1] (simple_push.py:45,56)
    if (h->kind == MPE_SIMPLE_PUSH && h->n_landmarks > 2) return MPE_ERR_DESC;
  }
  return MPE_OK;
}

extern "C" int mpe_obs_dim(const mpe_scenario_desc *h, int agent) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (agent < 0 || agent >= h->n_agents) return MPE_ERR_ARG;
  return obs_dim_of(h, agent);
}
extern "C" int mpe_obs_row0(const mpe_scenario_desc *h, int agent) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (agent < 0 || agent >= h->n_agents) return MPE_ERR_ARG;
  int r = 0;
  for (int i = 0; i < agent; ++i) r += obs_dim_of(h, i);
  return r;
}
extern "C" int mpe_obs_total(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  int r = 0;
  for (int i = 0; i < h->n_agents; ++i) r += obs_dim_of(h, i);
  return r;
}
extern "C" int mpe_act_dim(const mpe_scenario_desc *h, int agent) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (agent < 0 || agent >= h->n_agents) return MPE_ERR_ARG;
  return act_dim_of(h, agent);
}
extern "C" int mpe_act_row0(const mpe_scenario_desc *h, int agent) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (agent < 0 || agent >= h->n_agents) return MPE_ERR_ARG;
  int r = 0;
  for (int i = 0; i < agent; ++i) r += act_dim_of(h, i);
  return r;
}
extern "C" int mpe_act_total(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  int r = 0;
  for (int i = 0; i < h->n_agents; ++i) r += act_dim_of(h, i);
  return r;
}
extern "C" int mpe_goal_rows(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  return goal_rows_of(h);
}
extern "C" int mpe_info_width(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  return info_width_of(h);
}
extern "C" int mpe_stats_width(const mpe_scenario_desc *h) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  return 2 * h->n_agents + 2;
}

template <typename T>
static void to_device_desc(const mpe_scenario_desc *h, DevDesc<T> &d) {
  std::memset(&d, 0, sizeof(d));
  d.kind = h->kind;
  d.A = h->n_agents;
  d.L = h->n_landmarks;
  d.E = d.A + d.L;
  d.dim_c = h->dim_c;
  d.info_w = info_width_of(h);
  d.dt = (T)h->dt;
  d.damp1 = (T)(1 - h->damping);  // core.py:214: (1 - self.damping) evaluated in float64
  d.cforce = (T)h->contact_force;
  d.k = (T)h->contact_margin;
  d.inv_k = (T)(1.0 / h->contact_margin);
  d.has_lforce = has_lforce(h) ? 1 : 0;
  d.reward_variant = h->reward_variant;
  d.n_target_obs = h->n_target_obs;
  d.n_food = h->n_food;
  d.n_forests = h->n_forests;
  int row = 0, arow = 0;
  for (int e = 0; e < d.E; ++e) {
    d.size[e] = (T)h->size[e];
    if (h->movable[e]) d.movable[e >> 5] |= 1u << (e & 31);
    if (h->collide[e]) d.collide[e >> 5] |= 1u << (e & 31);
  }
  for (int e = d.A; e < d.E; ++e) {
    d.mass[e] = (T)h->mass[e];
    d.max_speed[e] = (T)h->lm_max_speed[e];
    if (h->movable[e]) d.lm_movable = 1;
  }
  for (int a = 0; a < d.A; ++a) {
    d.mass[a] = (T)h->mass[a];
    d.accel[a] = (T)h->accel[a];
    d.max_speed[a] = (T)h->max_speed[a];
    d.u_noise[a] = (T)h->u_noise[a];
    if ((kind_has_roles(h->kind) || h->kind == MPE_SIMPLE_CRYPTO) && h->adversary[a]) {
      d.adversary[a >> 5] |= 1u << (a & 31);
      d.adv_list[d.n_adv++] = (uint8_t)a;
    } else {
      d.good_list[d.n_good++] = (uint8_t)a;
    }
    d.obs_row0[a] = row;
    row += obs_dim_of(h, a);
    d.act_row0[a] = arow;
    d.cact_row0[a] = arow + (h->movable[a] ? 5 : 0);
    arow += act_dim_of(h, a);
    if (h->silent[a]) d.silent[a >> 5] |= 1u << (a & 31);
    if (h->advleader[a]) d.advleader[a >> 5] |= 1u << (a & 31);
    if (h->advsuperleader[a]) d.advsuper[a >> 5] |= 1u << (a & 31);
    if (h->leader[a]) d.leader[a >> 5] |= 1u << (a & 31);
    d.force_id[a] = h->force_id[a];
    d.forced_id[a] = h->forced_id[a];
    d.obs_r2[a] = (T)h->obs_r[a] * (T)h->obs_r[a];
    d.obs_r[a] = (T)h->obs_r[a];
    d.c_noise[a] = h->silent[a] ? (T)0 : (T)h->c_noise[a];
    if (observes_lforce(h, a)) d.obs_lf[a >> 5] |= 1u << (a & 31);
  }
  d.obs_total = row;
  int ncol = 0;
  while (ncol < d.E && h->collide[ncol]) ++ncol;
  d.ncol_prefix = ncol;
  for (int e = ncol; e < d.E; ++e)
    if (h->collide[e]) d.ncol_prefix = -1;
}

// ------------------------------------------------------------------------------------------------
// kernel selection
// ------------------------------------------------------------------------------------------------
// Stock shape of a *_MovingTarget world (moving_kernel header), or SMALL_NONE.
static SmallId moving_variant(const mpe_scenario_desc *h) {
  const int A = h->n_agents;
  if (h->n_landmarks != 0) return SMALL_NONE;
  SmallId id;
  if (A == 6 && h->dim_c == 3) id = MOVING_6_4_3;
  else if (A == 8 && h->dim_c == 7) id = MOVING_8_4_7;
  else return SMALL_NONE;
  const int nf = 4;
  for (int a = 0; a < A; ++a) {
    if ((h->silent[a] != 0) != (a < nf || a == A - 1)) return SMALL_NONE;   // followers, leaders, prey
    if ((h->adversary[a] != 0) != (a < A - 1)) return SMALL_NONE;
    if (h->silent[a] && h->force_id[a] >= 0) return SMALL_NONE;             // a silent agent commands nobody
    if (!h->movable[a] || !h->collide[a] || !(h->max_speed[a] >= 0.0) || h->u_noise[a] != 0.0) return SMALL_NONE;
  }
  return id;
}

// Stock shape of simple_world_comm (mpe_wcomm_kernel.h header), or SMALL_NONE.
static SmallId wcomm_variant(const mpe_scenario_desc *h) {
  const int A = 6, L = 5;
  if (h->n_agents != A || h->n_landmarks != L || h->n_food != 2 || h->n_forests != 2 || h->dim_c != 4) return SMALL_NONE;
  for (int a = 0; a < A; ++a) {
    if ((h->adversary[a] != 0) != (a < 4) || (h->silent[a] != 0) != (a > 0)) return SMALL_NONE;
    if (!(h->max_speed[a] >= 0.0) || h->u_noise[a] != 0.0) return SMALL_NONE;
  }
  for (int e = 0; e < A + L; ++e)
    if ((h->collide[e] != 0) != (e < A + 1)) return SMALL_NONE;
  return WCOMM_4_2_1_2_2;
}

// Stock shape of simple_speaker_listener / simple_reference (mpe_comm_kernel.h header), or SMALL_NONE.
static SmallId comm_variant(const mpe_scenario_desc *h) {
  const bool ref = h->kind == MPE_SIMPLE_REFERENCE;
  if (h->dim_c != (ref ? 10 : 3)) return SMALL_NONE;
  for (int e = 0; e < h->n_agents + h->n_landmarks; ++e)
    if (h->collide[e]) return SMALL_NONE;
  for (int a = 0; a < h->n_agents; ++a) {
    if (h->max_speed[a] >= 0.0 || h->u_noise[a] != 0.0) return SMALL_NONE;
    if (ref && (!h->movable[a] || h->silent[a])) return SMALL_NONE;  // speaker_listener roles: mpe_desc_check
  }
  return ref ? COMM_REFERENCE : COMM_SPEAKER_LISTENER;
}

// Stock shape of a command-hierarchy world (mpe_hier_kernel.h header), or SMALL_NONE.
static SmallId hier_variant(const mpe_scenario_desc *h) {
  const int A = h->n_agents, L = h->n_landmarks;
  if (h->dim_c != kHierDimC || h->n_target_obs != kHierTargets) return SMALL_NONE;
  int nf = 0;
  while (nf < A && h->silent[nf]) ++nf;
  for (int a = 0; a < A; ++a) {
    if ((h->silent[a] != 0) != (a < nf)) return SMALL_NONE;       // followers (silent) first
    if (a < nf && h->force_id[a] >= 0) return SMALL_NONE;         // a silent agent commands nobody
    if (!h->movable[a] || !(h->max_speed[a] >= 0.0) || h->u_noise[a] != 0.0) return SMALL_NONE;
  }
  for (int e = 0; e < A + L; ++e)
    if (!h->collide[e]) return SMALL_NONE;
  static const struct { int a, l, nf; SmallId id; } shapes[] = {
      {3, 3, 2, HIER_3_3_2},   {5, 3, 4, HIER_5_3_4},   {2, 10, 1, HIER_2_10_1}, {3, 10, 2, HIER_3_10_2},
      {3, 10, 1, HIER_3_10_1}, {5, 3, 2, HIER_5_3_2},   {5, 10, 2, HIER_5_10_2}};
  for (const auto &s : shapes)
    if (s.a == A && s.l == L && s.nf == nf) return s.id;
  return SMALL_NONE;
}

// The register kernel hard-codes the stock shapes and role flags (mpe_small_kernel.h: which
// entities collide, who has a max_speed, adversaries first); anything else -> tile kernel.
static bool any_movable_landmark(const mpe_scenario_desc *h) {
  for (int e = h->n_agents; e < h->n_agents + h->n_landmarks; ++e)
    if (h->movable[e]) return true;
  return false;
}

static SmallId small_variant(const mpe_scenario_desc *h) {
  const int A = h->n_agents, L = h->n_landmarks;
  if (any_movable_landmark(h)) return SMALL_NONE;  // the register kernels hard-code fixed landmarks
  SmallId id;
  int nadv, ecol;       // adversaries (first nadv agents), leading colliding entities
  bool has_max;
  switch (h->kind) {
    case MPE_SIMPLE_TAG: id = SMALL_TAG_3_1_2; if (A != 4 || L != 2) return SMALL_NONE; nadv = 3; ecol = 6; has_max = true; break;
    case MPE_SIMPLE_SPREAD: id = SMALL_SPREAD_3_3; if (A != 3 || L != 3) return SMALL_NONE; nadv = 0; ecol = 3; has_max = false; break;
    case MPE_SIMPLE: id = SMALL_SIMPLE_1_1; if (A != 1 || L != 1) return SMALL_NONE; nadv = 0; ecol = 0; has_max = false; break;
    case MPE_SIMPLE_ADVERSARY: id = SMALL_ADVERSARY_3_2; if (A != 3 || L != 2) return SMALL_NONE; nadv = 1; ecol = 0; has_max = false; break;
    case MPE_SIMPLE_PUSH: id = SMALL_PUSH_10_1; if (A != 10 || L != 1) return SMALL_NONE; nadv = 9; ecol = 10; has_max = false; break;
    case MPE_HIER_PURSUIT: return hier_variant(h);
    case MPE_SPEAKER_LISTENER:
    case MPE_SIMPLE_REFERENCE: return comm_variant(h);
    case MPE_SIMPLE_CRYPTO: return h->dim_c == 4 ? COMM_CRYPTO : SMALL_NONE;  // roles, shape: mpe_desc_check
    case MPE_WORLD_COMM: return wcomm_variant(h);
    case MPE_HIER_MOVING: return moving_variant(h);
    default: return SMALL_NONE;
  }
  for (int a = 0; a < A; ++a) {
    if (!h->silent[a] || h->forced_id[a] >= 0) return SMALL_NONE;  // fixed [A][5] action rows, no commands
    if (!h->movable[a]) return SMALL_NONE;
    if ((h->max_speed[a] >= 0.0) != has_max) return SMALL_NONE;
    if (h->kind != MPE_SIMPLE_SPREAD && h->kind != MPE_SIMPLE && (h->adversary[a] != 0) != (a < nadv)) return SMALL_NONE;
  }
  for (int e = 0; e < A + L; ++e)
    if ((h->collide[e] != 0) != (e < ecol)) return SMALL_NONE;
  return id;
}

// Envs per workgroup of the register kernel for this launch (= its statistics rows per env), or 0 = use the tile kernel.
// simple_push (10 agents) needs ~164 VGPRs in f32 (no 1024-lane workgroups) and > 256 in f64
// (register kernel would spill: the tile kernel takes it).
static int small_launch_threads(const mpe_scenario_desc *h, int64_t B, int dtype_bytes, uint32_t flags, SmallId *out) {
  SmallId sv = (flags & MPE_FLAG_FORCE_TILE) ? SMALL_NONE : small_variant(h);
  for (int a = 0; a < h->n_agents; ++a)  // comm noise: the generic kernel only (no stock scenario sets it)
    if (h->c_noise[a] != 0.0 && !h->silent[a]) sv = SMALL_NONE;
  if (out) *out = sv;
  if (sv == SMALL_NONE) return 0;
  if (sv == SMALL_PUSH_10_1) return dtype_bytes == 4 ? kSmallThreads : 0;
  if (small_id_splits(sv, B, flags)) return kSplitEnvs;
  if (hier_id_splits(sv, B, flags)) return hier_split_envs(sv);
  if (small_id_is_hier(sv)) return hier_threads(dtype_bytes);
  return kSmallThreads;
}

template <typename T, bool DO_STEP, bool DO_OBS>
static int launch(const mpe_scenario_desc *h, const KParams<T> &p, uint32_t flags, hipStream_t s) {
  SmallId sv;
  const int threads = small_launch_threads(h, p.B, (int)sizeof(T), flags, &sv);
  if (threads != 0) {
    KParams<T> q = p;  // the one-env-per-lane kernels follow the launch's residency plan (mpe_device.h)
    q.resid = residency_plan(h->n_agents, h->n_landmarks, p.lforce != nullptr, p.comm != nullptr ? speaking_comm_rows(h) : 0,
                             p.ld, (int)sizeof(T));
    if (small_id_is_hier(sv)) return launch_hier_id<T, DO_STEP, DO_OBS>(sv, q, hier_id_splits(sv, p.B, flags), s);
    int noise = 0;  // 1: the caller's normals, 2: drawn in the kernel (MPE_FLAG_DEVICE_NOISE)
    if (DO_STEP && (p.noise != nullptr || (flags & MPE_FLAG_DEVICE_NOISE)))
      for (int a = 0; a < h->n_agents; ++a)
        if (h->u_noise[a] != 0.0) noise = p.noise != nullptr ? 1 : 2;
    return launch_stock_id<T, DO_STEP, DO_OBS>(sv, q, noise, small_id_splits(sv, p.B, flags), s);
  }
  return launch_tile_world<T, DO_STEP, DO_OBS>(h, p, s);
}

extern "C" int mpe_residency_plan(const mpe_scenario_desc *h, int64_t ld, int dtype_bytes) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (ld < 1) return MPE_ERR_BATCH;
  if (dtype_bytes != 4 && dtype_bytes != 8) return MPE_ERR_ARG;
  return (int)residency_plan(h->n_agents, h->n_landmarks, has_lforce(h), speaking_comm_rows(h), ld, dtype_bytes);
}

extern "C" int64_t mpe_stats_rows(const mpe_scenario_desc *h, int64_t B, int dtype_bytes, uint32_t flags) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (B < 1) return MPE_ERR_BATCH;
  if (dtype_bytes != 4 && dtype_bytes != 8) return MPE_ERR_ARG;
  if (flags & ~(uint32_t)MPE_FLAG_ALL) return MPE_ERR_ARG;
  const int threads = small_launch_threads(h, B, dtype_bytes, flags, nullptr);
  if (threads != 0) return (B + threads - 1) / threads;
  return tile_grid(B, dtype_bytes == 4 ? tile_envs<float>() : tile_envs<double>());
}

template <typename T>
static bool misaligned(const void *ptr) { return ((uintptr_t)ptr % sizeof(T)) != 0; }

template <typename T>
static int check_common(const mpe_scenario_desc *h, int64_t B, int64_t ld) {
  int rc = mpe_desc_check(h);
  if (rc) return rc;
  if (B < 1 || ld < B) return MPE_ERR_BATCH;
  return MPE_OK;
}

// Pointer checks shared by the three stepping entry points; fills the kernel parameters.
template <typename T>
static int bind_buffers(const mpe_scenario_desc *h, int64_t B, int64_t ld, const mpe_buffers *b, bool do_step,
                        bool do_obs, KParams<T> &p) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!b || !b->pos || !b->vel) return MPE_ERR_NULL;
  if (b->flags & ~(uint32_t)MPE_FLAG_ALL) return MPE_ERR_ARG;
  if (do_step && !b->act) return MPE_ERR_NULL;
  const bool lf = has_lforce(h);
  if (lf && !b->lforce) return MPE_ERR_NULL;
  if (do_obs && goal_rows_of(h) > 0 && !b->goal) return MPE_ERR_NULL;
  if (do_obs && !do_step && kind_reads_comm(h->kind) && !b->comm) return MPE_ERR_NULL;  // observed agent.state.c
  p.pos = (T *)b->pos;
  p.vel = (T *)b->vel;
  p.lforce = lf ? (T *)b->lforce : nullptr;
  if (any_movable_landmark(h)) {  // their velocities: stepped, and nothing observes them in the kinds that allow one
    if (do_step && !b->lvel) return MPE_ERR_NULL;
    if (misaligned<T>(b->lvel)) return MPE_ERR_ALIGN;
    p.lvel = do_step ? (T *)b->lvel : nullptr;
  }
  if (do_step) {
    p.act = (const T *)b->act;
    p.noise = (const T *)b->noise;
    p.cnoise = (const T *)b->cnoise;
    p.comm = (T *)b->comm;
  }
  if (do_obs) {
    if (!do_step) p.comm = (T *)b->comm;
    p.goal = b->goal;
    p.obs = (T *)b->obs;
    p.rew = (T *)b->rew;
    p.info = info_width_of(h) == 0 ? nullptr : (T *)b->info;  // scenario without benchmark_data
    p.stats = do_step ? b->stats : nullptr;
  }
  if (misaligned<T>(p.act) || misaligned<T>(p.noise) || misaligned<T>(p.cnoise) || misaligned<T>(p.pos) || misaligned<T>(p.vel) ||
      misaligned<T>(p.comm) || misaligned<T>(p.lforce) || misaligned<T>(p.obs) || misaligned<T>(p.rew) ||
      misaligned<T>(p.info) || misaligned<double>(p.stats) || misaligned<int32_t>(p.goal))
    return MPE_ERR_ALIGN;
  p.B = B;
  p.ld = ld;
  p.flags = b->flags;
  if (do_step && (b->flags & MPE_FLAG_DEVICE_NOISE)) {
    if (misaligned<uint32_t>(b->noise_step_dev)) return MPE_ERR_ALIGN;
    p.nk0 = (uint32_t)b->noise_seed;
    p.nk1 = (uint32_t)(b->noise_seed >> 32);
    p.nstep = b->noise_step;
    p.nstep_dev = b->noise_step_dev;
    p.env_id0 = b->env_id0;
  }
  to_device_desc<T>(h, p.d);
  return MPE_OK;
}

template <typename T>
static int world_step_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const mpe_buffers *b, void *stream) {
  KParams<T> p{};
  int rc = bind_buffers<T>(h, B, ld, b, true, false, p);
  if (rc) return rc;
  return launch<T, true, false>(h, p, b->flags, (hipStream_t)stream);
}

template <typename T>
static int observe_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const mpe_buffers *b, void *stream) {
  KParams<T> p{};
  int rc = bind_buffers<T>(h, B, ld, b, false, true, p);
  if (rc) return rc;
  if (!p.obs && !p.rew && !p.info) return MPE_ERR_NULL;
  return launch<T, false, true>(h, p, b->flags, (hipStream_t)stream);
}

template <typename T>
static int env_step_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const mpe_buffers *b, void *stream) {
  KParams<T> p{};
  int rc = bind_buffers<T>(h, B, ld, b, true, true, p);
  if (rc) return rc;
  return launch<T, true, true>(h, p, b->flags, (hipStream_t)stream);
}

// ------------------------------------------------------------------------------------------------
// reset (Scenario.reset_world) and statistics reduce
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void reset_kernel(int64_t B, int64_t ld, int E, int A, int dim_c, double ra,
                                                     double rl, uint32_t k0, uint32_t k1, uint32_t episode,
                                                     uint64_t env_id0, const uint8_t *mask, T *pos, T *vel,
                                                     T *comm, int32_t *goal, int goal_rows, T *lvel) {
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  if (mask != nullptr && mask[env] == 0) return;
  const uint64_t gid = env_id0 + (uint64_t)env;
  if (goal != nullptr) {  // goal = np.random.choice(world.landmarks) (simple_adversary.py:45)
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), episode, 0xFFFFFFFFu, k0, k1, w);
    for (int r = 0; r < goal_rows; ++r)  // one word per goal row (goal_rows <= 2 with the stock scenarios)
      goal[(int64_t)r * ld + env] = (int32_t)(((uint64_t)w[r & 3] * (uint64_t)(E - A)) >> 32);
  }
  for (int e = 0; e < E; ++e) {
    uint32_t w[4];
    philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), episode, (uint32_t)e, k0, k1, w);
    const double r = e < A ? ra : rl;
    pos[(int64_t)(2 * e) * ld + env] = (T)uniform_pm(w[0], w[1], r);
    pos[(int64_t)(2 * e + 1) * ld + env] = (T)uniform_pm(w[2], w[3], r);
  }
  for (int r = 0; r < 2 * A; ++r) vel[(int64_t)r * ld + env] = T(0);
  if (lvel != nullptr)  // landmark.state.p_vel = zeros (simple_tag.py:54)
    for (int r = 0; r < 2 * (E - A); ++r) lvel[(int64_t)r * ld + env] = T(0);
  if (comm != nullptr)
    for (int r = 0; r < A * dim_c; ++r) comm[(int64_t)r * ld + env] = T(0);
}

template <typename T>
static int reset_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, uint64_t seed, uint32_t episode,
                      uint64_t env_id0, const uint8_t *mask, const mpe_buffers *b, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!b || !b->pos || !b->vel) return MPE_ERR_NULL;
  T *pos = (T *)b->pos, *vel = (T *)b->vel, *comm = (T *)b->comm;
  int32_t *goal = b->goal;
  if (misaligned<T>(pos) || misaligned<T>(vel) || misaligned<T>(comm) || misaligned<int32_t>(goal) ||
      misaligned<T>(b->lvel))
    return MPE_ERR_ALIGN;
  if (h->n_landmarks < 1 || goal_rows_of(h) < 1) goal = nullptr;
  const dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipLaunchKernelGGL(reset_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld,
                     h->n_agents + h->n_landmarks, h->n_agents, h->dim_c, h->agent_pos_range,
                     h->landmark_pos_range, (uint32_t)seed, (uint32_t)(seed >> 32), episode, env_id0, mask, pos,
                     vel, comm, goal, goal_rows_of(h), (T *)b->lvel);
  return (int)hipGetLastError();
}

// target_move (experiments/train_and_eval.py:207-222): one env per lane; the landmarks are independent
// of each other (a respawn moves only the landmark itself), so the reference's nested loop with
// `break` is "respawn l iff any follower touches it".
template <typename T>
__global__ __launch_bounds__(256) void target_move_kernel(int64_t B, int64_t ld, int A, int L, uint64_t followers,
                                                           DevSizes<T> sz, uint32_t k0, uint32_t k1,
                                                           uint32_t counter, uint64_t env_id0, T *pos, int32_t *count) {
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  const uint64_t gid = env_id0 + (uint64_t)env;
  int n = 0;
  for (int l = 0; l < L; ++l) {
    T *lp = pos + (int64_t)(2 * (A + l)) * ld + env;
    const T lx = lp[0], ly = lp[ld];
    bool hit = false;
    for (int a = 0; a < A; ++a) {
      if (!((followers >> a) & 1)) continue;
      const T dx = lx - pos[(int64_t)(2 * a) * ld + env], dy = ly - pos[(int64_t)(2 * a + 1) * ld + env];
      hit = hit || touching<T>(dx, dy, sz.size[A + l] + sz.size[a]);
    }
    if (hit) {
      uint32_t w[4];
      philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), counter, 0x80000000u | (uint32_t)l, k0, k1, w);
      lp[0] = (T)uniform_pm(w[0], w[1], 1.0);
      lp[ld] = (T)uniform_pm(w[2], w[3], 1.0);
      ++n;
    }
  }
  if (count != nullptr) count[env] = n;
}

template <typename T>
static int target_move_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, uint64_t seed, uint32_t counter,
                            uint64_t env_id0, const mpe_buffers *b, int32_t *count, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!b || !b->pos) return MPE_ERR_NULL;
  if (misaligned<T>(b->pos) || misaligned<int32_t>(count)) return MPE_ERR_ALIGN;
  uint64_t followers = 0;
  DevSizes<T> sz;
  for (int a = 0; a < h->n_agents; ++a)
    if (h->advfollower[a]) followers |= 1ull << a;
  for (int e = 0; e < h->n_agents + h->n_landmarks; ++e) sz.size[e] = (T)h->size[e];
  const dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipLaunchKernelGGL(target_move_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld, h->n_agents, h->n_landmarks,
                     followers, sz, (uint32_t)seed, (uint32_t)(seed >> 32), counter, env_id0, (T *)b->pos, count);
  return (int)hipGetLastError();
}

// good_agent_action / good_agent_victim_action (experiments/simple_train.py:157-205): one env per lane.
template <typename T>
__global__ __launch_bounds__(256) void scripted_action_kernel(int64_t B, int64_t ld, int A, int agent, int mode,
                                                               int target, T cap, const T *pos, T *act, T *point) {
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  const T px = pos[(int64_t)(2 * agent) * ld + env], py = pos[(int64_t)(2 * agent + 1) * ld + env];
  T tx, ty, gain;
  if (mode == 0) {  // :158-168: first corner with the largest summed distance to the other agents
    const T ex[4] = {T(-0.9), T(0.9), T(-0.9), T(0.9)}, ey[4] = {T(0.9), T(0.9), T(-0.9), T(-0.9)};
    T best = T(0);
    tx = ex[0];
    ty = ey[0];
    for (int i = 0; i < 4; ++i) {
      T dsum = T(0);
      for (int o = 0; o < A; ++o) {
        if (o == agent) continue;
        dsum = dsum + t_norm2<T>(ex[i] - pos[(int64_t)(2 * o) * ld + env], ey[i] - pos[(int64_t)(2 * o + 1) * ld + env]);
      }
      const bool take = i == 0 || dsum > best;
      best = take ? dsum : best;
      tx = take ? ex[i] : tx;
      ty = take ? ey[i] : ty;
    }
    gain = T(1);
  } else {  // :191
    tx = pos[(int64_t)(2 * target) * ld + env];
    ty = pos[(int64_t)(2 * target + 1) * ld + env];
    gain = T(7);
  }
  const T vx = (tx - px) / T(2), vy = (ty - py) / T(2);
  T mx = t_abs<T>(vx) * gain, my = t_abs<T>(vy) * gain;
  mx = cap < mx ? cap : mx;  // Python min(m, cap)
  my = cap < my ? cap : my;
  act[env] = T(0);
  act[ld + env] = vx < T(0) ? T(0) : mx;
  act[2 * ld + env] = vx < T(0) ? mx : T(0);
  act[3 * ld + env] = vy < T(0) ? T(0) : my;
  act[4 * ld + env] = vy < T(0) ? my : T(0);
  if (point != nullptr) {
    point[env] = tx;
    point[ld + env] = ty;
  }
}

template <typename T>
static int scripted_action_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, int agent, int mode, int target,
                                double cap, const void *pos, void *act, void *point, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!pos || !act) return MPE_ERR_NULL;
  if (agent < 0 || agent >= h->n_agents || (mode != 0 && mode != 1) || std::isnan(cap)) return MPE_ERR_ARG;
  if (mode == 1 && (target < 0 || target >= h->n_agents)) return MPE_ERR_ARG;
  if (!h->movable[agent]) return MPE_ERR_UNSUPPORTED;  // no physical action entries to write
  if (misaligned<T>(pos) || misaligned<T>(act) || misaligned<T>(point)) return MPE_ERR_ALIGN;
  const dim3 grid((unsigned)((B + 255) / 256)), block(256);
  hipLaunchKernelGGL(scripted_action_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld, h->n_agents, agent, mode,
                     target, (T)cap, (const T *)pos, (T *)act, (T *)point);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// The two non-default decodes of MultiAgentEnv._set_action (environment.py:176-187), as a pass over the SoA action
// rows in front of the step kernels (which always run the default decode, :188-190):
//   discrete_action_input (:176-182): the action is an integer; 1 / 2 -> u[0] = -1 / +1, 3 / 4 -> u[1] = -1 / +1,
//     anything else nothing.  Written as the 5-vector whose default decode gives the same u: note the SIGN -- as a
//     one-hot vector index 1 means u[0] = +1 (a[1] - a[2]), as an integer it means -1 -- so the integer k sets entry
//     {0, 2, 1, 4, 3}[k].
//   force_discrete_action (:184-187): d = np.argmax(a[0:5]) (first maximum, a NaN wins), a[0:5] = one_hot(d), in place;
//     comm entries untouched.
// One launch for every movable agent: blockIdx.y = agent.
// ------------------------------------------------------------------------------------------------
struct ActRows {
  int32_t row0[MPE_MAX_AGENTS];
  uint32_t movable[MPE_MAX_AGENTS / 32];
};

template <typename T>
__global__ __launch_bounds__(256) void action_from_index_kernel(int64_t B, int64_t ld, int64_t ld_idx, ActRows m,
                                                                const int32_t *idx, T *act) {
  const int a = blockIdx.y;
  if (!mpe::bit(m.movable, a)) return;
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  const int k = __builtin_nontemporal_load(idx + (int64_t)a * ld_idx + env);
  const int slot = k == 1 ? 2 : (k == 2 ? 1 : (k == 3 ? 4 : (k == 4 ? 3 : 0)));
  T *o = act + (int64_t)m.row0[a] * ld + env;
#pragma unroll
  for (int j = 0; j < 5; ++j) o[(int64_t)j * ld] = (j == slot && j != 0) ? T(1) : T(0);
}

template <typename T>
__global__ __launch_bounds__(256) void action_force_discrete_kernel(int64_t B, int64_t ld, ActRows m, T *act) {
  const int a = blockIdx.y;
  if (!mpe::bit(m.movable, a)) return;
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  T *o = act + (int64_t)m.row0[a] * ld + env;
  T best = o[0];
  int d = 0;
#pragma unroll
  for (int j = 1; j < 5; ++j) {
    const T v = o[(int64_t)j * ld];
    if (v > best || (v != v && best == best)) {  // np.argmax: first maximum, the first NaN wins
      best = v;
      d = j;
    }
  }
#pragma unroll
  for (int j = 0; j < 5; ++j) o[(int64_t)j * ld] = j == d ? T(1) : T(0);
}

// discrete_action_space == False (environment.py:191-192, "NEVER do here" in the fork's own comments, but a live switch):
// the action of a movable agent IS the force command, `agent.action.u = action[0]` (2 entries), times the sensitivity.
// Written as the 5-vector (0, u_x, 0, u_y, 0), whose default decode `a[1] - a[2]`, `a[3] - a[4]` returns u bit for bit
// (x - 0 is exact; NaN and infinities pass through).  With force_discrete_action as well (:184-187 run first, on the
// 2-vector): u = one_hot(argmax(u)), first maximum, a NaN wins.
template <typename T>
__global__ __launch_bounds__(256) void action_from_force_kernel(int64_t B, int64_t ld, int64_t ld_u, ActRows m,
                                                                int force_discrete, const T *u, T *act) {
  const int a = blockIdx.y;
  if (!mpe::bit(m.movable, a)) return;
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  if (env >= B) return;
  T ux = __builtin_nontemporal_load(u + (int64_t)(2 * a) * ld_u + env);
  T uy = __builtin_nontemporal_load(u + (int64_t)(2 * a + 1) * ld_u + env);
  if (force_discrete) {
    const bool second = uy > ux || (uy != uy && ux == ux);
    ux = second ? T(0) : T(1);
    uy = second ? T(1) : T(0);
  }
  T *o = act + (int64_t)m.row0[a] * ld + env;
  o[0] = T(0);
  o[ld] = ux;
  o[2 * ld] = T(0);
  o[3 * ld] = uy;
  o[4 * ld] = T(0);
}

static void act_rows_of(const mpe_scenario_desc *h, ActRows &m) {
  memset(&m, 0, sizeof(m));
  for (int i = 0; i < h->n_agents; ++i) {
    m.row0[i] = mpe_act_row0(h, i);
    if (h->movable[i]) m.movable[i >> 5] |= 1u << (i & 31);
  }
}

template <typename T>
static int action_from_index_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, int64_t ld_idx, const int32_t *idx,
                                  void *act, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!idx || !act) return MPE_ERR_NULL;
  if (ld_idx < B) return MPE_ERR_BATCH;
  if (misaligned<int32_t>(idx) || misaligned<T>(act)) return MPE_ERR_ALIGN;
  // the reference's integer decode of a speaking agent trips its own `assert len(action) == 0` (environment.py:200-208:
  // the comm branch never consumes its entry), and a MultiDiscrete action cannot be an integer at all
  for (int i = 0; i < h->n_agents; ++i)
    if (!h->silent[i]) return MPE_ERR_UNSUPPORTED;
  ActRows m;
  act_rows_of(h, m);
  const dim3 grid((unsigned)((B + 255) / 256), (unsigned)h->n_agents), block(256);
  hipLaunchKernelGGL(action_from_index_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld, ld_idx, m, idx, (T *)act);
  return (int)hipGetLastError();
}

template <typename T>
static int action_from_force_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, int64_t ld_u, const void *u,
                                  int force_discrete, void *act, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!u || !act) return MPE_ERR_NULL;
  if (ld_u < B) return MPE_ERR_BATCH;
  if (misaligned<T>(u) || misaligned<T>(act)) return MPE_ERR_ALIGN;
  // a movable agent that also speaks has a MultiDiscrete action: environment.py:161-168 hands the decode its first 5
  // entries, `agent.action.u` becomes a 5-vector and World.integrate_state fails to add it to the 2-vector velocity
  // (core.py:216, ValueError): nothing to reproduce
  for (int i = 0; i < h->n_agents; ++i)
    if (h->movable[i] && !h->silent[i]) return MPE_ERR_UNSUPPORTED;
  ActRows m;
  act_rows_of(h, m);
  const dim3 grid((unsigned)((B + 255) / 256), (unsigned)h->n_agents), block(256);
  hipLaunchKernelGGL(action_from_force_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld, ld_u, m,
                     force_discrete ? 1 : 0, (const T *)u, (T *)act);
  return (int)hipGetLastError();
}

template <typename T>
static int action_force_discrete_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, void *act, void *stream) {
  int rc = check_common<T>(h, B, ld);
  if (rc) return rc;
  if (!act) return MPE_ERR_NULL;
  if (misaligned<T>(act)) return MPE_ERR_ALIGN;
  ActRows m;
  act_rows_of(h, m);
  const dim3 grid((unsigned)((B + 255) / 256), (unsigned)h->n_agents), block(256);
  hipLaunchKernelGGL(action_force_discrete_kernel<T>, grid, block, 0, (hipStream_t)stream, B, ld, m, (T *)act);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// replay buffer (maddpg/maddpg/trainer/replay_buffer.py): ring append with an SoA -> record transpose, row gather
// ------------------------------------------------------------------------------------------------
struct ReplaySegs {
  const void *ptr[8];
  int64_t row_stride[8], env_stride[8];
  int32_t row0[9];  // first record column of each segment; row0[nseg] = width
  int32_t nseg;
};

// One workgroup = 64 consecutive transitions; the record columns go through LDS 64 at a time: the loads
// run along the transitions (coalesced for SoA sources), the stores along the record (each wavefront
// writes 64 consecutive columns of one ring row).
template <typename T>
__global__ __launch_bounds__(256) void replay_add_kernel(int64_t capacity, int64_t next_idx, int64_t n, int64_t skip,
                                                          int width, ReplaySegs sg, T *ring) {
  __shared__ T tile[64][65];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e0 = skip + (int64_t)blockIdx.x * 64;  // transitions below `skip` would be overwritten anyway
  for (int c0 = 0; c0 < width; c0 += 64) {
    for (int k = w; k < 64; k += 4) {  // column c0 + k (wave-uniform), transition e0 + lane
      const int c = c0 + k;
      if (c < width && e0 + lane < n) {
        int s = 0;
        while (c >= sg.row0[s + 1]) ++s;
        const T *src = (const T *)sg.ptr[s];
        tile[k][lane] = src[(int64_t)(c - sg.row0[s]) * sg.row_stride[s] + (e0 + lane) * sg.env_stride[s]];
      }
    }
    __syncthreads();
    for (int k = w; k < 64; k += 4) {  // transition e0 + k (wave-uniform), column c0 + lane
      const int64_t e = e0 + k;
      if (e < n && c0 + lane < width) ring[((next_idx + e) % capacity) * width + c0 + lane] = tile[lane][k];
    }
    __syncthreads();
  }
}

// Records of up to kReplayFlatCols columns: the whole tile of TW transitions sits in LDS, and since
// consecutive transitions are consecutive ring rows the tile is one contiguous span of the ring (except at
// the wrap): stored with a flat index, every lane busy whatever the record width.  TW = 256 (each wavefront
// loads its own 64 transitions, so a workgroup reads 1 KiB of every source row, like the env kernels) when
// the tile fits 64 KiB of LDS: that case runs replay_add_multi_kernel below (its tuned form) with one job; this
// kernel is left with TW = 64 (the four wavefronts share the columns) for wider records.
constexpr int kReplayFlatCols = 160;
template <typename T, int TW>
__global__ __launch_bounds__(256) void replay_add_flat_kernel(int64_t capacity, int64_t next_idx, int64_t n, int64_t skip,
                                                               int width, ReplaySegs sg, T *ring) {
  extern __shared__ __align__(16) unsigned char replay_lds[];
  T *tile = reinterpret_cast<T *>(replay_lds);  // [width][TW + 1]
  constexpr int LDT = TW + 1;
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e0 = skip + (int64_t)blockIdx.x * TW;
  const int ne = (int)(n - e0 < TW ? n - e0 : TW);
  // column c (wave-uniform), transition e0 + el; kU independent loads in flight per wavefront
  constexpr int kU = 10;
  constexpr int CSTEP = TW == 64 ? 4 : 1;           // TW == 64: wavefront w takes columns w, w + 4, ...
  const int el = TW == 64 ? lane : 64 * w + lane;   // TW == 256: wavefront w takes transitions 64 w ..
  for (int c0 = (TW == 64 ? w : 0); c0 < width; c0 += CSTEP * kU) {
    T v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int c = c0 + CSTEP * u;
      v[u] = T(0);
      if (c < width && el < ne) {
        int s = 0;
        while (c >= sg.row0[s + 1]) ++s;
        const T *src = (const T *)sg.ptr[s];
        v[u] = ld_stream(src + (int64_t)(c - sg.row0[s]) * sg.row_stride[s] + (e0 + el) * sg.env_stride[s]);
      }
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (c0 + CSTEP * u < width) tile[(c0 + CSTEP * u) * LDT + el] = v[u];
  }
  __syncthreads();
  const int total = ne * width;
  const int64_t r0 = (next_idx + e0) % capacity;  // workgroup-uniform; the tile wraps at most once (ne <= capacity)
  for (int f = threadIdx.x; f < total; f += 256) {
    const int e = f / width, c = f - e * width;
    int64_t r = r0 + e;
    r = r >= capacity ? r - capacity : r;
    st_stream(ring + r * width + c, tile[c * LDT + e]);  // written once, read much later
  }
}

// The same append for several replay buffers in one launch (blockIdx.y = buffer): one per agent and step in the
// reference's loop (experiments/simple_train.py:284-287).  Flat TW = 256 form only; the host falls back to one launch
// per buffer for shapes it does not cover.
constexpr int kReplayMaxJobs = 8;
template <typename T>
struct ReplayJobs {
  ReplaySegs sg[kReplayMaxJobs];
  int64_t capacity[kReplayMaxJobs], next_idx[kReplayMaxJobs], skip[kReplayMaxJobs];
  T *ring[kReplayMaxJobs];
  int32_t width[kReplayMaxJobs];
  // optional device word added to next_idx (mod capacity): the part of the ring cursor that moves between replays of a
  // HIP graph, whose launches bake their arguments (mpe_replay_add_multi_cur_*)
  const int64_t *cursor[kReplayMaxJobs];
};
template <typename T>
__global__ __launch_bounds__(256) void replay_add_multi_kernel(int64_t n, const ReplayJobs<T> jobs) {
  extern __shared__ __align__(16) unsigned char replay_lds[];
  constexpr int TW = 256, LDT = TW + 1;
  T *tile = reinterpret_cast<T *>(replay_lds);
  const int jb = blockIdx.y;
  const ReplaySegs &sg = jobs.sg[jb];
  const int width = jobs.width[jb];
  const int64_t capacity = jobs.capacity[jb];
  const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  const int64_t e0 = jobs.skip[jb] + (int64_t)blockIdx.x * TW;
  if (e0 >= n) return;  // workgroup-uniform: this buffer keeps fewer transitions than the widest job
  const int ne = (int)(n - e0 < TW ? n - e0 : TW);
  // Where each record column comes from, worked out once per workgroup: searching the segment table in front of every
  // load (a chain of dependent scalar loads) paced the loads -- the read side alone took the kernel's whole 92 us.
  __shared__ const T *s_src[kReplayFlatCols];
  __shared__ int64_t s_es[kReplayFlatCols];
  if ((int)threadIdx.x < width) {
    const int c = threadIdx.x;
    int s = 0;
    while (c >= sg.row0[s + 1]) ++s;
    s_src[c] = (const T *)sg.ptr[s] + (int64_t)(c - sg.row0[s]) * sg.row_stride[s] + e0 * sg.env_stride[s];
    s_es[c] = sg.env_stride[s];
  }
  __syncthreads();
  constexpr int kU = 40;  // row loads in flight per lane (1 Mi x 39 columns f32: 10 -> 71.5 us, 20 -> 63.6, 40 -> 61.4)
  const int el = 64 * w + lane;
  const int elc = el < ne ? el : ne - 1;  // lanes / columns beyond the tile re-read its last element: no branch per load,
  for (int c0 = 0; c0 < width; c0 += kU) {  // so a batch is 20 table reads, one wait, 20 loads in flight
    T v[kU];
#pragma unroll
    for (int u = 0; u < kU; ++u) {
      const int c = c0 + u < width ? c0 + u : width - 1;
      // (a pointer read back from LDS has lost its address space: say it is global, or the loads become FLAT ones,
      // which every LDS wait then waits for as well -- one load in flight per wavefront)
      typedef const T __attribute__((address_space(1))) * gptr;
      v[u] = __builtin_nontemporal_load((gptr)(uintptr_t)s_src[c] + elc * s_es[c]);
    }
#pragma unroll
    for (int u = 0; u < kU; ++u)
      if (c0 + u < width) tile[(c0 + u) * LDT + el] = v[u];
  }
  __syncthreads();
  const int total = ne * width;
  T *ring = jobs.ring[jb];
  const int64_t moved = jobs.cursor[jb] != nullptr ? *jobs.cursor[jb] % capacity : 0;  // workgroup-uniform
  const int64_t r0 = (jobs.next_idx[jb] + moved + e0) % capacity;
  // element f = e * width + c of the workgroup's contiguous run of records.  f < 2^16 and width < 2^16, so
  // f / width == umulhi(f, floor(2^32 / width) + 1) exactly: no division per element, and the trips stay independent
  // (carrying (e, c) from trip to trip instead serialises the LDS reads: 99.6 vs 93 us per 1 Mi x 39 append)
  const uint32_t inv = (uint32_t)(0x100000000ull / (uint32_t)width) + 1u;
  // the run is contiguous in the ring unless it wraps: four consecutive elements per lane in one 16-byte store
  // (any 4-byte alignment is legal for global_store_dwordx4), the last total % 4 elements one by one
  const bool wraps = r0 + ne > capacity;
  if (sizeof(T) == 4 && !wraps) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    T *run = ring + r0 * width;
    const int nq = total >> 2;
#pragma unroll 2
    for (int q = threadIdx.x; q < nq; q += 256) {
      int e = (int)__umulhi((uint32_t)(4 * q), inv), c = 4 * q - e * width;
      f4 v;
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        v[k] = (float)tile[c * LDT + e];
        if (++c == width) {
          c = 0;
          ++e;
        }
      }
      __builtin_nontemporal_store(v, reinterpret_cast<f4 *>(run + 4 * q));
    }
    for (int f = 4 * nq + threadIdx.x; f < total; f += 256) {
      const int e = (int)__umulhi((uint32_t)f, inv), c = f - e * width;
      st_stream(run + f, tile[c * LDT + e]);
    }
    return;
  }
#pragma unroll 4
  for (int f = threadIdx.x; f < total; f += 256) {
    const int e = (int)__umulhi((uint32_t)f, inv), c = f - e * width;
    int64_t r = r0 + e;
    r = r >= capacity ? r - capacity : r;
    st_stream(ring + r * width + c, tile[c * LDT + e]);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void replay_gather_kernel(int64_t capacity, int width, const int64_t *idx,
                                                             int64_t total, const T *ring, T *out) {
  // four elements per thread, 256 apart: independent index and row loads in flight
  const int64_t blk = (int64_t)blockIdx.x * 1024;
  const int64_t b0 = blk / width;                 // workgroup-uniform 64-bit division, once
  const uint32_t off0 = (uint32_t)(blk - b0 * width);
  const int64_t g0 = blk + threadIdx.x;
  T v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t g = g0 + 256 * u;
    v[u] = T(0);
    if (g < total) {
      const uint32_t t = off0 + threadIdx.x + 256u * u;  // < 1024 + width: 32-bit division
      const uint32_t db = t / (uint32_t)width;
      const int j = (int)(t - db * (uint32_t)width);
      int64_t i = idx[b0 + db];
      if ((uint64_t)i >= (uint64_t)capacity) {  // out of range: wrap instead of faulting
        i %= capacity;
        i = i < 0 ? i + capacity : i;
      }
      v[u] = ring[i * width + j];
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u)
    if (g0 + 256 * u < total) st_stream(out + g0 + 256 * u, v[u]);
}

template <typename T>
static int replay_add_impl(int64_t capacity, int64_t next_idx, int64_t n, int width, int nseg,
                           const mpe_replay_seg *segs, void *ring, void *stream) {
  if (!segs || !ring) return MPE_ERR_NULL;
  if (capacity < 1 || next_idx < 0 || next_idx >= capacity || n < 0 || width < 1 || nseg < 1 || nseg > 8) return MPE_ERR_ARG;
  ReplaySegs sg{};
  int rows = 0;
  for (int s = 0; s < nseg; ++s) {
    if (!segs[s].ptr) return MPE_ERR_NULL;
    if (segs[s].rows < 1) return MPE_ERR_ARG;
    if (misaligned<T>(segs[s].ptr)) return MPE_ERR_ALIGN;
    sg.ptr[s] = segs[s].ptr;
    sg.row_stride[s] = segs[s].row_stride;
    sg.env_stride[s] = segs[s].env_stride;
    sg.row0[s] = rows;
    rows += segs[s].rows;
  }
  for (int s = nseg; s <= 8; ++s) sg.row0[s] = rows;
  sg.nseg = nseg;
  if (rows != width) return MPE_ERR_ARG;
  if (misaligned<T>(ring)) return MPE_ERR_ALIGN;
  if (n == 0) return MPE_OK;
  const int64_t skip = n > capacity ? n - capacity : 0;
  const dim3 block(256);
  if (width <= kReplayFlatCols) {
    const size_t lds256 = (size_t)width * 257 * sizeof(T), lds64 = (size_t)width * 65 * sizeof(T);
    if (lds256 <= 64 * 1024 && n - skip >= 256) {
      // the one-launch-for-several-buffers kernel with a single job (it is the tuned form of the TW = 256 tile)
      ReplayJobs<T> tab{};
      tab.sg[0] = sg;
      tab.capacity[0] = capacity;
      tab.next_idx[0] = next_idx;
      tab.skip[0] = skip;
      tab.ring[0] = (T *)ring;
      tab.width[0] = width;
      const dim3 grid((unsigned)((n - skip + 255) / 256), 1u);
      hipLaunchKernelGGL(replay_add_multi_kernel<T>, grid, block, lds256, (hipStream_t)stream, n, tab);
    } else {
      const dim3 grid((unsigned)((n - skip + 63) / 64));
      if (lds64 > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute((const void *)replay_add_flat_kernel<T, 64>,
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds64);
        if (e != hipSuccess) return (int)e;
      }
      hipLaunchKernelGGL((replay_add_flat_kernel<T, 64>), grid, block, lds64, (hipStream_t)stream, capacity, next_idx,
                         n, skip, width, sg, (T *)ring);
    }
  } else {
    const dim3 grid((unsigned)((n - skip + 63) / 64));
    hipLaunchKernelGGL(replay_add_kernel<T>, grid, block, 0, (hipStream_t)stream, capacity, next_idx, n, skip, width,
                       sg, (T *)ring);
  }
  return (int)hipGetLastError();
}

// Fills the kernel-side segment table of one append from the C ABI's segments; returns the record width or an error.
template <typename T>
static int replay_bind_segs(int nseg, const mpe_replay_seg *segs, ReplaySegs &sg) {
  if (!segs) return MPE_ERR_NULL;
  if (nseg < 1 || nseg > 8) return MPE_ERR_ARG;
  int rows = 0;
  for (int s = 0; s < nseg; ++s) {
    if (!segs[s].ptr) return MPE_ERR_NULL;
    if (segs[s].rows < 1) return MPE_ERR_ARG;
    if (misaligned<T>(segs[s].ptr)) return MPE_ERR_ALIGN;
    sg.ptr[s] = segs[s].ptr;
    sg.row_stride[s] = segs[s].row_stride;
    sg.env_stride[s] = segs[s].env_stride;
    sg.row0[s] = rows;
    rows += segs[s].rows;
  }
  for (int s = nseg; s <= 8; ++s) sg.row0[s] = rows;
  sg.nseg = nseg;
  return rows;
}

// Does the one-launch append kernel cover this job?  Records of <= kReplayFlatCols columns whose 256-transition tile
// (dynamic LDS) fits 64 KiB TOGETHER WITH the kernel's static column tables (s_src / s_es), and a full tile to store.
constexpr size_t kReplayStaticLds = (size_t)kReplayFlatCols * (sizeof(void *) + sizeof(int64_t));
template <typename T>
static bool replay_one_launch(const mpe_replay_job &j, int64_t n) {
  const int64_t skip = n > j.capacity ? n - j.capacity : 0;
  return j.width <= kReplayFlatCols && (size_t)j.width * 257 * sizeof(T) + kReplayStaticLds <= 64 * 1024 &&
         n - skip >= 256;
}

template <typename T>
static int replay_add_multi_impl(int n_jobs, const mpe_replay_job *jobs, const int64_t *const *cursors, int64_t n,
                                 void *stream) {
  if (!jobs) return MPE_ERR_NULL;
  if (n_jobs < 1 || n < 0) return MPE_ERR_ARG;
  for (int i = 0; i < n_jobs; ++i) {  // validate everything before the first launch
    const mpe_replay_job &j = jobs[i];
    ReplaySegs sg{};
    const int rows = replay_bind_segs<T>(j.nseg, j.segs, sg);
    if (rows < 0) return rows;
    if (!j.ring) return MPE_ERR_NULL;
    if (j.capacity < 1 || j.next_idx < 0 || j.next_idx >= j.capacity || rows != j.width) return MPE_ERR_ARG;
    if (misaligned<T>(j.ring)) return MPE_ERR_ALIGN;
    // the single-buffer kernels take no device cursor: refused here, before anything is launched
    if (n > 0 && cursors && cursors[i] && !replay_one_launch<T>(j, n)) return MPE_ERR_UNSUPPORTED;
  }
  if (n == 0) return MPE_OK;
  int i0 = 0;
  while (i0 < n_jobs) {
    // a run of jobs the one-launch kernel covers (records of <= kReplayFlatCols columns, tile in 64 KiB of LDS, a
    // full 256-transition tile to store); anything else goes through the single-buffer entry point
    ReplayJobs<T> tab{};
    int nj = 0, wmax = 0;
    int64_t keep_max = 0;
    while (i0 + nj < n_jobs && nj < kReplayMaxJobs) {
      const mpe_replay_job &j = jobs[i0 + nj];
      const int64_t skip = n > j.capacity ? n - j.capacity : 0;
      if (!replay_one_launch<T>(j, n)) break;
      replay_bind_segs<T>(j.nseg, j.segs, tab.sg[nj]);
      tab.capacity[nj] = j.capacity;
      tab.next_idx[nj] = j.next_idx;
      tab.skip[nj] = skip;
      tab.ring[nj] = (T *)j.ring;
      tab.width[nj] = j.width;
      tab.cursor[nj] = cursors ? cursors[i0 + nj] : nullptr;
      wmax = j.width > wmax ? j.width : wmax;
      keep_max = n - skip > keep_max ? n - skip : keep_max;
      ++nj;
    }
    if (nj == 0) {
      const mpe_replay_job &j = jobs[i0];
      const int rc = replay_add_impl<T>(j.capacity, j.next_idx, n, j.width, j.nseg, j.segs, j.ring, stream);
      if (rc) return rc;
      ++i0;
      continue;
    }
    const dim3 grid((unsigned)((keep_max + 255) / 256), (unsigned)nj), block(256);
    hipLaunchKernelGGL(replay_add_multi_kernel<T>, grid, block, (size_t)wmax * 257 * sizeof(T), (hipStream_t)stream, n, tab);
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
    i0 += nj;
  }
  return MPE_OK;
}

// Gather from a COMPACT ring (include/mpe_hip.h, mpe_replay_gather_compact): records (obs_t | act | rew | done) of
// width cw = od + ad + 2; obs_tp1 of row i is obs_t of row (i + n) % capacity (the same env one step later) unless the
// row's step slot i / n is flagged: then its own copy ends[i][od].  Output rows keep the full layout
// (obs_t | act | rew | obs_tp1 | done), so callers see what mpe_replay_gather gives them.
template <typename T>
__global__ __launch_bounds__(256) void replay_gather_compact_kernel(int64_t capacity, int od, int ad, int64_t n,
                                                                     const int64_t *idx, int64_t total, const T *ring,
                                                                     const T *ends, const uint8_t *own, T *out) {
  const int width = 2 * od + ad + 2, cw = od + ad + 2;
  const int64_t blk = (int64_t)blockIdx.x * 1024;
  const int64_t b0 = blk / width;
  const uint32_t off0 = (uint32_t)(blk - b0 * width);
  const int64_t g0 = blk + threadIdx.x;
  T v[4];
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t g = g0 + 256 * u;
    v[u] = T(0);
    if (g < total) {
      const uint32_t t = off0 + threadIdx.x + 256u * u;
      const uint32_t db = t / (uint32_t)width;
      const int j = (int)(t - db * (uint32_t)width);
      int64_t i = idx[b0 + db];
      if ((uint64_t)i >= (uint64_t)capacity) {  // out of range: wrap instead of faulting
        i %= capacity;
        i = i < 0 ? i + capacity : i;
      }
      if (j < od + ad + 1) {  // obs_t | act | rew
        v[u] = ring[i * cw + j];
      } else if (j == width - 1) {  // done
        v[u] = ring[i * cw + cw - 1];
      } else {  // obs_tp1, column c
        const int c = j - (od + ad + 1);
        if (own[i / n]) {
          v[u] = ends[i * od + c];
        } else {
          int64_t nx = i + n;
          nx = nx >= capacity ? nx - capacity : nx;
          v[u] = ring[nx * cw + c];
        }
      }
    }
  }
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t g = g0 + 256 * u;
    if (g < total) out[g] = v[u];
  }
}

template <typename T>
static int replay_gather_compact_impl(int64_t capacity, int od, int ad, int64_t n, const int64_t *idx, int64_t batch,
                                      const void *ring, const void *ends, const uint8_t *own, void *out, void *stream) {
  if (!idx || !ring || !ends || !own || !out) return MPE_ERR_NULL;
  if (capacity < 1 || od < 1 || ad < 1 || n < 1 || capacity % n != 0 || batch < 0) return MPE_ERR_ARG;
  if (misaligned<T>(ring) || misaligned<T>(ends) || misaligned<T>(out) || misaligned<int64_t>(idx)) return MPE_ERR_ALIGN;
  if (batch == 0) return MPE_OK;
  const int64_t total = batch * (2 * od + ad + 2);
  const dim3 grid((unsigned)((total + 1023) / 1024)), block(256);
  hipLaunchKernelGGL(replay_gather_compact_kernel<T>, grid, block, 0, (hipStream_t)stream, capacity, od, ad, n, idx,
                     total, (const T *)ring, (const T *)ends, own, (T *)out);
  return (int)hipGetLastError();
}

// Gather from a STEP STORE (include/mpe_hip.h, mpe_replay_gather_steps): the rows the step kernels wrote, kept where
// they were written -- per step slot the observation rows [obs_rows][ld], action rows [act_rows][ld], reward rows and
// (optionally) done rows of all agents.  Transition i = (k = i / n, env = i % n) lives in slot live[k]; its obs_tp1 are
// the observation rows of slot (live[k] + 1) % slots.  Output rows in the full layout of mpe_replay_gather.
template <typename T>
__global__ __launch_bounds__(256) void replay_gather_steps_kernel(mpe_step_store st, int od, int ad, const int64_t *idx,
                                                                  int64_t total, T *out) {
  const int width = 2 * od + ad + 2;
  const T *obs = (const T *)st.obs, *act = (const T *)st.act, *rew = (const T *)st.rew, *done = (const T *)st.done;
  // the count is read from device memory, behind the table: a gather recorded in a HIP graph bakes its arguments, and
  // the store grows between replays
  int n_live = st.live[st.slots];
  n_live = n_live < 1 ? 1 : (n_live >= st.slots ? st.slots - 1 : n_live);
  const int64_t cap = (int64_t)n_live * st.n;
  const int64_t blk = (int64_t)blockIdx.x * 1024;
  const int64_t b0 = blk / width;
  const uint32_t off0 = (uint32_t)(blk - b0 * width);
  const int64_t g0 = blk + threadIdx.x;
#pragma unroll
  for (int u = 0; u < 4; ++u) {
    const int64_t g = g0 + 256 * u;
    if (g >= total) continue;
    const uint32_t t = off0 + threadIdx.x + 256u * u;
    const uint32_t db = t / (uint32_t)width;
    const int j = (int)(t - db * (uint32_t)width);
    int64_t i = idx[b0 + db];
    if ((uint64_t)i >= (uint64_t)cap) {  // out of range: wrap instead of faulting
      i %= cap;
      i = i < 0 ? i + cap : i;
    }
    const int64_t k = i / st.n, e = i - k * st.n;
    int slot = st.live[k];
    slot = (slot < 0 || slot >= st.slots) ? 0 : slot;
    T v;
    if (j < od) {
      v = obs[(int64_t)slot * st.obs_slot_stride + (int64_t)(st.obs_row0 + j) * st.ld + e];
    } else if (j < od + ad) {
      v = act[(int64_t)slot * st.act_slot_stride + (int64_t)(st.act_row0 + j - od) * st.ld + e];
    } else if (j == od + ad) {
      v = rew[(int64_t)slot * st.rew_slot_stride + (int64_t)st.rew_row * st.ld + e];
    } else if (j == width - 1) {
      v = done != nullptr ? done[(int64_t)slot * st.rew_slot_stride + (int64_t)st.rew_row * st.ld + e] : T(0);
    } else {
      const int nx = slot + 1 == st.slots ? 0 : slot + 1;
      v = obs[(int64_t)nx * st.obs_slot_stride + (int64_t)(st.obs_row0 + j - (od + ad + 1)) * st.ld + e];
    }
    out[g] = v;
  }
}

template <typename T>
static int replay_gather_steps_impl(const mpe_step_store *st, int od, int ad, const int64_t *idx, int64_t batch, void *out,
                                    void *stream) {
  if (!st || !idx || !out || !st->obs || !st->act || !st->rew || !st->live) return MPE_ERR_NULL;
  if (st->slots < 2 || st->n_live < 1 || st->n_live >= st->slots || st->n < 1 || st->ld < st->n || od < 1 || ad < 1 ||
      batch < 0 || st->obs_row0 < 0 || st->act_row0 < 0 || st->rew_row < 0 ||
      st->obs_slot_stride < (int64_t)(st->obs_row0 + od) * st->ld || st->act_slot_stride < (int64_t)(st->act_row0 + ad) * st->ld ||
      st->rew_slot_stride < (int64_t)(st->rew_row + 1) * st->ld)
    return MPE_ERR_ARG;
  if (misaligned<T>(st->obs) || misaligned<T>(st->act) || misaligned<T>(st->rew) || misaligned<T>(st->done) ||
      misaligned<T>(out) || misaligned<int64_t>(idx) || misaligned<int32_t>(st->live))
    return MPE_ERR_ALIGN;
  if (batch == 0) return MPE_OK;
  const int64_t total = batch * (2 * od + ad + 2);
  const dim3 grid((unsigned)((total + 1023) / 1024)), block(256);
  hipLaunchKernelGGL(replay_gather_steps_kernel<T>, grid, block, 0, (hipStream_t)stream, *st, od, ad, idx, total, (T *)out);
  return (int)hipGetLastError();
}

template <typename T>
static int replay_gather_impl(int64_t capacity, int width, const int64_t *idx, int64_t batch, const void *ring,
                              void *out, void *stream) {
  if (!idx || !ring || !out) return MPE_ERR_NULL;
  if (capacity < 1 || width < 1 || batch < 0) return MPE_ERR_ARG;
  if (misaligned<T>(ring) || misaligned<T>(out) || misaligned<int64_t>(idx)) return MPE_ERR_ALIGN;
  if (batch == 0) return MPE_OK;
  const int64_t total = batch * width;
  const dim3 grid((unsigned)((total + 1023) / 1024)), block(256);
  hipLaunchKernelGGL(replay_gather_kernel<T>, grid, block, 0, (hipStream_t)stream, capacity, width, idx, total,
                     (const T *)ring, (T *)out);
  return (int)hipGetLastError();
}

static int actor_job_check(const mpe_actor_job *j) {
  if (!j->obs || !j->weights || !j->act) return MPE_ERR_NULL;
  if (j->in_dim < 1 || j->in_dim > 4096 || j->s0 < 1 || j->s1 < 0 || j->s0 + j->s1 > kActorMaxOut) return MPE_ERR_ARG;
  if (misaligned<float>(j->obs) || misaligned<float>(j->weights) || misaligned<float>(j->u) || misaligned<float>(j->act))
    return MPE_ERR_ALIGN;
  return MPE_OK;
}

extern "C" int mpe_actor_forward_multi_ctr_f32(int n_jobs, const mpe_actor_job *jobs, int64_t B, int64_t ld_obs,
                                               int64_t ld_act, uint32_t call_counter, const uint32_t *counter_dev,
                                               uint64_t env_id0, int kernel, void *stream) {
  if (!jobs) return MPE_ERR_NULL;
  if (misaligned<uint32_t>(counter_dev)) return MPE_ERR_ALIGN;
  if (n_jobs < 1 || kernel < MPE_ACTOR_AUTO || kernel > MPE_ACTOR_MFMA_F16) return MPE_ERR_ARG;
  if (B < 1 || ld_obs < B || ld_act < B) return MPE_ERR_BATCH;
  for (int i = 0; i < n_jobs; ++i) {
    const int rc = actor_job_check(&jobs[i]);
    if (rc) return rc;
  }
  for (int i0 = 0; i0 < n_jobs; i0 += kActorMaxJobs) {  // kActorMaxJobs agents per launch
    const int nj = n_jobs - i0 < kActorMaxJobs ? n_jobs - i0 : kActorMaxJobs;
    ActorJobs tab{};
    tab.counter_dev = counter_dev;
    int wide = 0;
    for (int i = 0; i < nj; ++i) {
      const mpe_actor_job &j = jobs[i0 + i];
      tab.j[i] = ActorJob{(const float *)j.obs, (const float *)j.weights, (const float *)j.u, (float *)j.act,
                          j.in_dim, j.s0, j.s1, (uint32_t)j.seed, (uint32_t)(j.seed >> 32)};
      wide = wide || j.s0 + j.s1 > 8;
    }
    if (kernel == MPE_ACTOR_VALU) {  // MPE_ACTOR_AUTO: the matrix-core kernel (1.6-2.5x the vector kernel from 1 Ki to 1 Mi envs)
      const dim3 grid((unsigned)((B + 255) / 256), (unsigned)nj), block(256);
      auto kern = wide ? actor_forward_kernel<16> : actor_forward_kernel<8>;
      hipLaunchKernelGGL(kern, grid, block, 0, (hipStream_t)stream, B, ld_obs, ld_act, tab, call_counter, env_id0);
    } else {
      // matrix-core form: 4 wavefronts per workgroup, each taking `tpw` consecutive tiles of 32 envs (its resident
      // W2 / W3 fragments are fetched once per wavefront: amortise them, but keep >= ~2 workgroups per CU busy)
      const int64_t tiles = (B + 31) / 32;
      int tpw = 8;
      while (tpw > 1 && tiles / (4 * tpw) < 1024 / nj) tpw >>= 1;
      const dim3 grid((unsigned)((tiles + 4 * tpw - 1) / (4 * tpw)), (unsigned)nj), block(256);
      // one instantiation per observation width in k-blocks of 16 rows (all jobs of the launch alike), else the generic one
      int nkb = (tab.j[0].in_dim + 15) / 16;
      for (int i = 1; i < nj; ++i)
        if ((tab.j[i].in_dim + 15) / 16 != nkb) nkb = 0;
      if (nkb > 3 || (int64_t)48 * ld_obs >= ((int64_t)1 << 30)) nkb = 0;  // 32-bit row offsets inside the kernel
      using Kern = void (*)(int64_t, int64_t, int64_t, const ActorJobs, uint32_t, uint64_t, int);
      static const Kern kerns[2][2][4] = {
          {{actor_mfma_kernel<8, 0>, actor_mfma_kernel<8, 1>, actor_mfma_kernel<8, 2>, actor_mfma_kernel<8, 3>},
           {actor_mfma_kernel<16, 0>, actor_mfma_kernel<16, 1>, actor_mfma_kernel<16, 2>, actor_mfma_kernel<16, 3>}},
          {{actor_mfma_kernel<8, 0, true>, actor_mfma_kernel<8, 1, true>, actor_mfma_kernel<8, 2, true>,
            actor_mfma_kernel<8, 3, true>},
           {actor_mfma_kernel<16, 0, true>, actor_mfma_kernel<16, 1, true>, actor_mfma_kernel<16, 2, true>,
            actor_mfma_kernel<16, 3, true>}}};
      hipLaunchKernelGGL(kerns[kernel == MPE_ACTOR_MFMA_F16][wide][nkb], grid, block, 0, (hipStream_t)stream, B, ld_obs,
                         ld_act, tab, call_counter, env_id0, tpw);
    }
    const hipError_t e = hipGetLastError();
    if (e != hipSuccess) return (int)e;
  }
  return MPE_OK;
}

extern "C" int mpe_actor_forward_multi_f32(int n_jobs, const mpe_actor_job *jobs, int64_t B, int64_t ld_obs,
                                           int64_t ld_act, uint32_t call_counter, uint64_t env_id0, int kernel,
                                           void *stream) {
  return mpe_actor_forward_multi_ctr_f32(n_jobs, jobs, B, ld_obs, ld_act, call_counter, nullptr, env_id0, kernel, stream);
}

extern "C" int mpe_actor_forward_f32(int64_t B, int64_t ld_obs, int64_t ld_act, int in_dim, int s0, int s1,
                                     const void *obs, const void *weights, const void *u, uint64_t seed,
                                     uint32_t call_counter, uint64_t env_id0, void *act, void *stream) {
  const mpe_actor_job job = {obs, weights, u, act, in_dim, s0, s1, seed};
  return mpe_actor_forward_multi_f32(1, &job, B, ld_obs, ld_act, call_counter, env_id0, MPE_ACTOR_AUTO, stream);
}

// One workgroup per column; rows are summed in a fixed order (thread t takes rows t, t+256, ...
// sequentially, then a wavefront butterfly, then the 4 wavefront partials in a fixed order), so the
// result is bitwise reproducible for a given (rows, width).
__global__ __launch_bounds__(256) void stats_reduce_kernel(const double *stats, int64_t rows, int width,
                                                           double *out) {
  __shared__ double sp[4];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int c = blockIdx.x;
  double s = 0.0;
  for (int64_t r = threadIdx.x; r < rows; r += 256) s += stats[r * width + c];
  s = group_sum(s, 64);
  if (lane == 0) sp[wave] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[c] = (sp[0] + sp[1]) + (sp[2] + sp[3]);
}

extern "C" int mpe_stats_reduce(const double *stats, int64_t rows, int width, double *out, void *stream) {
  if (!stats || !out) return MPE_ERR_NULL;
  if (rows < 1 || width < 1) return MPE_ERR_ARG;
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)width), dim3(256), 0, (hipStream_t)stream, stats, rows,
                     width, out);
  return (int)hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// mpe_env_step_stats_static: fused env step + statistics reduce with the per-workgroup partial rows in a
// file-scope device array instead of a caller-owned buffer (for callers that cannot allocate: HIP graph
// capture, serving loops).  Calls no hipMalloc / hipFree.
// ------------------------------------------------------------------------------------------------
constexpr size_t kScratchDoubles = MPE_STATIC_SCRATCH_BYTES / sizeof(double);
static __device__ __align__(256) double scratch[kScratchDoubles];
// The array's name is a host-side stand-in, not a device address: each GPU holds its own copy at its own
// address, looked up once per device with hipGetSymbolAddress and kept here.
constexpr int kMaxDevices = 64;
static double *g_scratch_ptr[kMaxDevices];

static int scratch_of_current_device(double **out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return (int)e;
  if (dev < 0 || dev >= kMaxDevices) return MPE_ERR_ARG;
  double *ptr = __atomic_load_n(&g_scratch_ptr[dev], __ATOMIC_ACQUIRE);
  if (ptr == nullptr) {
    e = hipGetSymbolAddress((void **)&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess) return (int)e;
    if (ptr == nullptr) return (int)hipErrorInvalidSymbol;
    __atomic_store_n(&g_scratch_ptr[dev], ptr, __ATOMIC_RELEASE);
  }
  *out = ptr;
  return MPE_OK;
}

template <typename T>
static int env_step_stats_static_impl(const mpe_scenario_desc *h, int64_t B, int64_t ld, const mpe_buffers *b,
                                      double *stats_out, hipStream_t stream) {
  if (!b) return MPE_ERR_NULL;
  mpe_buffers local = *b;  // buf->stats is documented as ignored: whatever it holds must not reach the pointer checks
  local.stats = nullptr;
  b = &local;
  KParams<T> p{};
  int rc = bind_buffers<T>(h, B, ld, b, true, true, p);
  if (rc) return rc;
  const int width = 2 * h->n_agents + 2;
  const int64_t rows = mpe_stats_rows(h, B, (int)sizeof(T), b->flags);
  if (rows < 0) return (int)rows;
  if ((uint64_t)rows * (uint64_t)width > kScratchDoubles) return MPE_ERR_SCRATCH;
  double *ptr = nullptr;
  rc = scratch_of_current_device(&ptr);
  if (rc) return rc;
  // the array keeps the previous call's partial sums and the kernels accumulate (+=): clear what this call uses
  hipError_t e = hipMemsetAsync(ptr, 0, (size_t)rows * width * sizeof(double), stream);
  if (e != hipSuccess) return (int)e;
  p.stats = ptr;
  rc = launch<T, true, true>(h, p, b->flags, stream);
  if (rc) return rc;
  hipLaunchKernelGGL(stats_reduce_kernel, dim3((unsigned)width), dim3(256), 0, stream, ptr, rows, width, stats_out);
  return (int)hipGetLastError();
}

extern "C" int mpe_env_step_stats_static(const mpe_scenario_desc *d, int64_t B, int64_t ld, const mpe_buffers *b,
                                         int dtype_bytes, double *stats_out, void *stream) {
  if (!stats_out) return MPE_ERR_NULL;
  if (((uintptr_t)stats_out % sizeof(double)) != 0) return MPE_ERR_ALIGN;
  if (dtype_bytes == 4) return env_step_stats_static_impl<float>(d, B, ld, b, stats_out, (hipStream_t)stream);
  if (dtype_bytes == 8) return env_step_stats_static_impl<double>(d, B, ld, b, stats_out, (hipStream_t)stream);
  return MPE_ERR_ARG;
}

// ------------------------------------------------------------------------------------------------
// exported typed entry points
// ------------------------------------------------------------------------------------------------
#define MPE_EXPORT_TYPED(SUF, T)                                                                               \
  extern "C" int mpe_world_step_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, const mpe_buffers *b, \
                                      void *stream) {                                                          \
    return world_step_impl<T>(d, B, ld, b, stream);                                                            \
  }                                                                                                            \
  extern "C" int mpe_observe_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, const mpe_buffers *b,    \
                                   void *stream) {                                                             \
    return observe_impl<T>(d, B, ld, b, stream);                                                               \
  }                                                                                                            \
  extern "C" int mpe_env_step_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, const mpe_buffers *b,   \
                                    void *stream) {                                                            \
    return env_step_impl<T>(d, B, ld, b, stream);                                                              \
  }                                                                                                            \
  extern "C" int mpe_target_move_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, uint64_t seed,       \
                                       uint32_t counter, uint64_t env_id0, const mpe_buffers *b,               \
                                       int32_t *count, void *stream) {                                         \
    return target_move_impl<T>(d, B, ld, seed, counter, env_id0, b, count, stream);                            \
  }                                                                                                            \
  extern "C" int mpe_scripted_action_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, int agent,       \
                                           int mode, int target, double cap, const void *pos, void *act_out,   \
                                           void *point_out, void *stream) {                                    \
    return scripted_action_impl<T>(d, B, ld, agent, mode, target, cap, pos, act_out, point_out, stream);       \
  }                                                                                                            \
  extern "C" int mpe_action_from_index_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, int64_t ld_idx, \
                                             const int32_t *idx, void *act, void *stream) {                    \
    return action_from_index_impl<T>(d, B, ld, ld_idx, idx, act, stream);                                      \
  }                                                                                                            \
  extern "C" int mpe_action_from_force_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, int64_t ld_u,    \
                                             const void *u, int force_discrete, void *act, void *stream) {     \
    return action_from_force_impl<T>(d, B, ld, ld_u, u, force_discrete, act, stream);                          \
  }                                                                                                            \
  extern "C" int mpe_action_force_discrete_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, void *act,  \
                                                 void *stream) {                                               \
    return action_force_discrete_impl<T>(d, B, ld, act, stream);                                               \
  }                                                                                                            \
  extern "C" int mpe_replay_add_##SUF(int64_t capacity, int64_t next_idx, int64_t n, int width, int nseg,      \
                                      const mpe_replay_seg *segs, void *ring, void *stream) {                  \
    return replay_add_impl<T>(capacity, next_idx, n, width, nseg, segs, ring, stream);                         \
  }                                                                                                            \
  extern "C" int mpe_replay_add_multi_##SUF(int n_jobs, const mpe_replay_job *jobs, int64_t n, void *stream) { \
    return replay_add_multi_impl<T>(n_jobs, jobs, nullptr, n, stream);                                         \
  }                                                                                                            \
  extern "C" int mpe_replay_add_multi_cur_##SUF(int n_jobs, const mpe_replay_job *jobs,                        \
                                                const int64_t *const *cursors, int64_t n, void *stream) {      \
    if (!cursors) return MPE_ERR_NULL;                                                                         \
    for (int i = 0; i < n_jobs; ++i)                                                                           \
      if (misaligned<int64_t>(cursors[i])) return MPE_ERR_ALIGN;                                               \
    return replay_add_multi_impl<T>(n_jobs, jobs, cursors, n, stream);                                         \
  }                                                                                                            \
  extern "C" int mpe_replay_gather_##SUF(int64_t capacity, int width, const int64_t *idx, int64_t batch,       \
                                         const void *ring, void *out, void *stream) {                          \
    return replay_gather_impl<T>(capacity, width, idx, batch, ring, out, stream);                              \
  }                                                                                                            \
  extern "C" int mpe_replay_gather_compact_##SUF(int64_t capacity, int od, int ad, int64_t n, const int64_t *idx,       \
                                                 int64_t batch, const void *ring, const void *ends,                  \
                                                 const uint8_t *own, void *out, void *stream) {                      \
    return replay_gather_compact_impl<T>(capacity, od, ad, n, idx, batch, ring, ends, own, out, stream);            \
  }                                                                                                            \
  extern "C" int mpe_replay_gather_steps_##SUF(const mpe_step_store *store, int od, int ad, const int64_t *idx,       \
                                               int64_t batch, void *out, void *stream) {                              \
    return replay_gather_steps_impl<T>(store, od, ad, idx, batch, out, stream);                                       \
  }                                                                                                            \
  extern "C" int mpe_reset_##SUF(const mpe_scenario_desc *d, int64_t B, int64_t ld, uint64_t seed,             \
                                 uint32_t episode, uint64_t env_id0, const uint8_t *mask,                      \
                                 const mpe_buffers *b, void *stream) {                                         \
    return reset_impl<T>(d, B, ld, seed, episode, env_id0, mask, b, stream);                                   \
  }

MPE_EXPORT_TYPED(f32, float)
MPE_EXPORT_TYPED(f64, double)
