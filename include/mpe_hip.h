/*
 * mpe_hip.h -- C ABI of libmpe_hip.so: the MI355X (gfx950) batched implementation of the
 * multiagent-particle-envs hot path  MultiAgentEnv.step -> World.step -> scenario gathers.
 *
 * The reference (asamakei/marl, /root/reference/multiagent-particle-envs) is pure Python and has
 * no FFI for this path; its boundary is the duck-typed Python API
 *     make_env(scenario_name, benchmark)                      make_env.py:15-44
 *     MultiAgentEnv.step(action_n) / reset()                  multiagent/environment.py:92-129
 *     Scenario.make_world/reset_world/reward/observation/benchmark_data   multiagent/scenario.py:4-10
 * The functions below are what a ctypes binding under that API calls (marl_amd/_hip.py is that
 * binding; INTEGRATION.md shows the stub a maintainer of the reference would add).
 *
 * Conventions
 *   - Plain C, plain pointers and sizes.  All array arguments (the members of mpe_buffers, `mask`,
 *     the stats vectors) are DEVICE pointers; mpe_scenario_desc and mpe_buffers themselves live in
 *     host memory and are read during the call only.  The library allocates nothing, keeps no state that changes results (the only
 *     process-wide data are a per-device record of which kernels already had their dynamic-LDS limit raised and, for
 *     mpe_env_step_stats_static alone, the per-device address of its built-in scratch array and that array's contents),
 *     never synchronises: every call only enqueues kernels on `stream` (a hipStream_t passed as void*; NULL =
 *     default stream) of the CURRENT device, which must be the device that owns the buffers.  Safe to call
 *     concurrently from several threads / processes on different devices.
 *   - Return value: 0 = OK, < 0 = MPE_ERR_* (bad argument; nothing was launched), > 0 = hipError_t
 *     of the failed launch.  Never throws, never aborts.
 *   - Struct-of-arrays device layout, env index fastest ("row" = one scalar for all envs):
 *         pos  [E][2][ld]    E = n_agents + n_landmarks, agents first (core.py:109-111)
 *         vel  [A][2][ld]    the agents' velocities (a world with a movable landmark also has lvel [L][2][ld])
 *         act  [act_total][ld]  soft one-hot Discrete(5) action per agent (environment.py:188-190),
 *                            [A][5][ld] when every agent is silent; see mpe_buffers
 *         noise[A][2][ld]    standard normals for u_noise (core.py:147); may be NULL (then none, or with
 *                            MPE_FLAG_DEVICE_NOISE drawn in the kernel)
 *         comm [A][dim_c][ld]  agent.state.c, written as zeros (all agents silent, core.py:226-227);
 *                              may be NULL (then not written)
 *         obs  [obs_total][ld]  agent i owns rows [obs_row0(i), obs_row0(i)+obs_dim(i))
 *         rew  [A][ld]
 *         info [A][info_w][ld]  info_w = 1 (simple_tag, simple_world_comm: collisions), 4 (simple_spread:
 *                               rew, collisions, min_dists, occupied_landmarks), L+1
 *                               (simple_adversary: squared distances; the adversary uses column
 *                               0 only) or 0 (simple, simple_push: no benchmark_data); may be NULL
 *         goal [ld] int32       index of the goal landmark (agent.goal_a, simple_adversary.py:45-48
 *                               / simple_push.py:48-50); required for those two kinds, ignored
 *                               (may be NULL) otherwise
 *     `B` = number of envs processed, `ld` = row stride in elements (ld >= B), so a sub-range of a
 *     larger batch can be processed by offsetting the pointers.
 *   - _f32 / _f64 = arithmetic (and storage) type of the path.  f64 follows the reference's
 *     float64 operation order (max abs deviation from numpy <= 1e-9 per step, in practice ~1e-15 over
 *     free-running 25-step episodes); f32 is the throughput mode: <= 1e-4 relative PER STEP FROM IDENTICAL
 *     STATE (hardware rsq / exp / log, ~1 ulp each).  Over a free-running episode f32 trajectories drift apart
 *     like any chaotic contact dynamics in single precision, and the discrete outputs (rewards of +-10, contact
 *     counts, visibility masks, the order of equidistant targets) can differ from the float64 reference on an
 *     env that sits within ~1e-5 of a strict `<` threshold.  "Relative" means relative to the terms summed: an agent in
 *     contact with dozens of entities at once (100-entity worlds packed into a unit square) has a net force far smaller
 *     than its contact terms, and single precision rounds each term to 2^-24 of ITS size -- the same loss numpy float32
 *     shows on those worlds (tools/fuzz_parity.py measures the kernels against it); the BASELINE configurations are
 *     nowhere near that regime.
 */
#ifndef MPE_HIP_H
#define MPE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPE_ABI_VERSION 12
#define MPE_MAX_AGENTS 64
#define MPE_MAX_ENTITIES 128

/* scenario kinds (which observation/reward/info gather runs) */
#define MPE_SIMPLE_TAG 0    /* build/lib/multiagent/scenarios/simple_tag.py    */
#define MPE_SIMPLE_SPREAD 1 /* build/lib/multiagent/scenarios/simple_spread.py */
#define MPE_SIMPLE 2        /* build/lib/multiagent/scenarios/simple.py            */
#define MPE_SIMPLE_ADVERSARY 3 /* build/lib/multiagent/scenarios/simple_adversary.py (per-env goal) */
#define MPE_SIMPLE_PUSH 4   /* build/lib/multiagent/scenarios/simple_push.py      (per-env goal) */
#define MPE_HIER_PURSUIT 5  /* multiagent/scenarios/L1F..py and S1L..py: the fork's follower / leader /
                               super-leader pursuit family (leader force core.py:153-194, masked +
                               distance-sorted observations, hierarchical rewards)               */
#define MPE_SPEAKER_LISTENER 6 /* build/lib/multiagent/scenarios/simple_speaker_listener.py: agent 0 fixed and
                                  speaking, agent 1 moving and silent; per-env goal landmark (agent 0's goal_b) */
#define MPE_SIMPLE_REFERENCE 7 /* build/lib/multiagent/scenarios/simple_reference.py: two moving, speaking agents,
                                  one goal landmark per agent (goal_b)                                          */
#define MPE_SIMPLE_CRYPTO 8    /* build/lib/multiagent/scenarios/simple_crypto.py: adversary, listener, speaker; nobody
                                  moves, everybody speaks; goal rows = (goal landmark, landmark of the speaker's key) */

#define MPE_WORLD_COMM 9       /* build/lib/multiagent/scenarios/simple_world_comm.py: simple_tag roles and physics, a
                                  speaking leader, food (rewards) and two forests (visibility masks)                 */

#define MPE_HIER_MOVING 10     /* multiagent/scenarios/L1F4_MovingTarget.py, S1L2F4_MovingTarget.py: the command hierarchy
                                  chasing a moving prey (no landmarks); reward_variant 0 = the L1F4 file's observation
                                  text (command links are visible, forced agents observe leader_force), 1 = the S1L2F4
                                  file's (radius only; agents that are neither advleader nor good observe it)        */

/* error codes */
#define MPE_OK 0
#define MPE_ERR_NULL (-1)        /* a required pointer is NULL                       */
#define MPE_ERR_DESC (-2)        /* descriptor fails mpe_desc_check                  */
#define MPE_ERR_BATCH (-3)       /* B <= 0 or ld < B                                  */
#define MPE_ERR_ALIGN (-4)       /* a device pointer is not aligned to its element   */
#define MPE_ERR_UNSUPPORTED (-5) /* e.g. movable landmarks outside kinds 0-4         */
#define MPE_ERR_ARG (-6)         /* any other bad scalar argument                    */
#define MPE_ERR_SCRATCH (-7)     /* batch needs more than the built-in scratch array (mpe_env_step_stats_static) */

/* mpe_buffers.flags bits (per call; 0 = default behaviour) */
#define MPE_FLAG_FORCE_TILE 1u   /* run the generic LDS-tile kernel even for a world that has a register-resident
                                    kernel (tests check both kernels against the same golden vectors; tuning)  */
#define MPE_FLAG_WIDE 2u         /* simple_tag / simple_spread / the command-hierarchy worlds (stock shapes): always one env
                                    per lane, never the small-batch form                                               */
#define MPE_FLAG_SPLIT 4u        /* the same worlds: one agent per wavefront (the small-batch form, default for
                                    B <= MPE_SPLIT_MAX_ENVS, for the hierarchy worlds B <= MPE_HIER_SPLIT_MAX_ENVS) at any
                                    batch size.  Both forms produce the same bits; the statistics rows group 64 envs
                                    instead of 256                                                                     */
#define MPE_FLAG_SILENT_COMM_ZERO 8u /* the comm rows of SILENT agents already hold zeros (e.g. the buffer was zeroed when it
                                    was allocated and only the step calls write it): the step leaves them alone instead of
                                    rewriting agent.state.c = zeros(dim_c) (core.py:226-227) every time.  Same state
                                    afterwards, 5-10 % fewer rows written in the hierarchy worlds.  Without the flag
                                    every comm row is written by every step, whatever it held                          */
#define MPE_FLAG_DEVICE_NOISE 16u /* u_noise / c_noise normals (core.py:147-148,229) that the caller does not supply (noise /
                                    cnoise NULL) are drawn inside the step kernels: Philox4x32-10, counter (global env id
                                    lo, hi, step, 0x40000000 | agent) for an agent's two u_noise normals and (.., 0x60000000
                                    | agent << 8 | j) for its c_noise components 2j, 2j + 1, key = noise_seed, Box-Muller;
                                    global env id = env_id0 + index, step = noise_step (+ *noise_step_dev).  A draw depends
                                    on (seed, env, step, agent, component) only: not on B, the sharding or the kernel form.
                                    Without the flag an agent whose normals are not supplied gets no noise (as before).
                                    f64: restated bit for bit by oracle/mpe_oracle.py:device_normals; f32: same Philox
                                    words, hardware log / sin / cos, within ~1e-6 of the f64 normal                  */
#define MPE_FLAG_ALL 31u
#define MPE_SPLIT_MAX_ENVS 32768
#define MPE_HIER_SPLIT_MAX_ENVS 65536

/*
 * World + scenario constants: what Scenario.make_world() sets up (simple_tag.py:7-36,
 * simple_spread.py:7-28) plus the World defaults (core.py:88-106).  Host memory, read during the
 * call only.  Entities are indexed agents first, then landmarks.
 */
typedef struct mpe_scenario_desc {
  int32_t kind;        /* MPE_SIMPLE_TAG / _SPREAD / MPE_SIMPLE / _ADVERSARY / _PUSH */
  int32_t n_agents;    /* A, 1..MPE_MAX_AGENTS */
  int32_t n_landmarks; /* L, A+L <= MPE_MAX_ENTITIES */
  int32_t dim_c;       /* world.dim_c (comm width; 2 in both scenarios) */
  double dt;             /* core.py:100 */
  double damping;        /* core.py:102 */
  double contact_force;  /* core.py:104 */
  double contact_margin; /* core.py:105 */
  double agent_pos_range;    /* reset: agents    ~ U(-r, r)^2  (simple_tag.py:48)            */
  double landmark_pos_range; /* reset: landmarks ~ U(-r, r)^2  (simple_tag.py:53 -> 0.9)     */
  double size[MPE_MAX_ENTITIES];    /* entity.size (core.py:32)                               */
  double mass[MPE_MAX_ENTITIES];    /* entity.mass (core.py:51-53)                            */
  double accel[MPE_MAX_AGENTS];     /* action sensitivity: agent.accel, or 5.0 when None
                                       (environment.py:193-196)                              */
  double max_speed[MPE_MAX_AGENTS]; /* agent.max_speed; < 0 encodes None = no clamp
                                       (core.py:217)                                         */
  double u_noise[MPE_MAX_AGENTS];   /* agent.u_noise; 0 encodes None (core.py:147)            */
  uint8_t movable[MPE_MAX_ENTITIES];
  uint8_t collide[MPE_MAX_ENTITIES];
  uint8_t adversary[MPE_MAX_AGENTS]; /* simple_tag: agent.adversary (simple_tag.py:21)        */
  /* fork additions (L1F2_3Targets_100observe.py:36-86); stock scenarios: silent = 1, ids = -1 */
  uint8_t silent[MPE_MAX_AGENTS];    /* 0: the agent's action carries dim_c comm entries after the
                                        5 physical ones (MultiDiscrete, environment.py:61-69,161-168) */
  int8_t force_id[MPE_MAX_AGENTS];   /* >= 0: commands the agents whose forced_id equals it
                                        (core.py:157-160)                                       */
  int8_t forced_id[MPE_MAX_AGENTS];  /* >= 0: is commanded; -1: none                            */
  uint8_t advleader[MPE_MAX_AGENTS];
  uint8_t advsuperleader[MPE_MAX_AGENTS]; /* its command is the raw c[0:2], observed not applied
                                             (core.py:162-163).  A force_id shared by a super-leader and a
                                             plain leader is refused (MPE_ERR_UNSUPPORTED): the reference
                                             aliases the super-leader's action.c there (core.py:163,192)  */
  int32_t reward_variant;            /* 0: L1F.. / S1L.. default; 1: S1L2F2_10Targets_wide_rewardFixed;
                                        MPE_HIER_MOVING: 0 = L1F4_MovingTarget, 1 = S1L2F4_MovingTarget   */
  int32_t n_target_obs;              /* target_pos[0:n] in the observation (3)                  */
  double obs_r[MPE_MAX_AGENTS];      /* agent.obs_r: observation radius                         */
  uint8_t advfollower[MPE_MAX_AGENTS]; /* agent.advfollower (L1F2_3Targets_100observe.py:40): the
                                          agents whose touch respawns a target (mpe_target_move)   */
  /* simple_world_comm.py:19-57: world.landmarks = plain landmarks, then n_food food, then n_forests
     forests (the observation hard-codes two forests, :240-245); 0 for every other kind             */
  int32_t n_food;
  int32_t n_forests;
  uint8_t leader[MPE_MAX_AGENTS];      /* agent.leader (simple_world_comm.py:27,261): sees everybody  */
  double c_noise[MPE_MAX_AGENTS];      /* agent.c_noise; 0 encodes None (core.py:229)                 */
  /* Movable landmarks (movable[e] != 0 for e >= n_agents; no scenario file of the reference has one, but
     core.py:211-222 integrates every movable entity): entity.max_speed of entity e >= n_agents, < 0 encodes
     None; entries of agents are ignored (theirs is max_speed[]).  Such a world needs mpe_buffers.lvel, runs
     the generic tile kernel, and is accepted for the all-silent stock kinds 0-4 only. */
  double lm_max_speed[MPE_MAX_ENTITIES];
} mpe_scenario_desc;

/*
 * Device buffers of one batch (struct-of-arrays, see the layout list above).  Element type of the
 * void pointers = the _f32 / _f64 suffix of the function called.  Which members a call reads or
 * writes is stated at each function; members it does not use may be NULL.
 *   act    [act_total][ld]   agent i owns rows [act_row0(i), act_row0(i)+act_dim(i)): 5 physical
 *                            entries when the agent is movable (environment.py:51-53,174), then dim_c
 *                            comm entries when it is not silent (:61-63,198-206)
 *   goal   [goal_rows][ld]   landmark indices: agent.goal_a of simple_adversary / simple_push (1 row),
 *                            agent 0's goal_b of simple_speaker_listener (1 row), every agent's goal_b of
 *                            simple_reference (n_agents rows), (goal, key) of simple_crypto (2 rows); see
 *                            mpe_goal_rows
 *   comm   [A][dim_c][ld]    agent.state.c: written by every world step when not NULL; READ by
 *                            mpe_observe of the speaking stock scenarios (kinds 6-9; required there)
 *   lforce [A][2][ld]        agent.leader_force (core.py:156-192): written by every world step,
 *                            read by the HIER_PURSUIT observation; NOT touched by reset (like the
 *                            reference's reset_world).  Required when any forced_id >= 0.
 */
typedef struct mpe_buffers {
  const void *act;
  const void *noise; /* [A][2][ld] standard normals for u_noise, or NULL (see MPE_FLAG_DEVICE_NOISE) */
  void *pos;         /* [E][2][ld] */
  void *vel;         /* [A][2][ld] */
  void *comm;        /* [A][dim_c][ld] agent.state.c, or NULL (then not written) */
  void *lforce;
  int32_t *goal;     /* [ld] goal landmark index (simple_adversary / simple_push) */
  void *obs;         /* [obs_total][ld] */
  void *rew;         /* [A][ld] */
  void *info;        /* [A][info_w][ld] */
  double *stats;     /* [mpe_stats_rows][mpe_stats_width] or NULL */
  const void *cnoise; /* [A][dim_c][ld] standard normals for c_noise (core.py:229): agent.state.c = action.c +
                         c_noise * cnoise for a speaking agent with a c_noise; NULL = no comm noise, or with
                         MPE_FLAG_DEVICE_NOISE drawn in the kernel */
  uint32_t flags;     /* MPE_FLAG_* bits; unknown bits -> MPE_ERR_ARG */
  /* MPE_FLAG_DEVICE_NOISE only (ignored without it) */
  uint32_t noise_step;             /* step counter of this call: the caller bumps it by one per step */
  uint64_t noise_seed;             /* Philox key */
  uint64_t env_id0;                /* global id of env 0 of this call (shards of one batch: their offset) */
  const uint32_t *noise_step_dev;  /* DEVICE word added to noise_step when not NULL: a step recorded in a HIP graph bakes
                                      noise_step, so the part that moves from replay to replay lives in memory */
  void *lvel;        /* [L][2][ld] velocities of the landmarks (entity.state.p_vel of world.landmarks): required by the
                        stepping calls when a landmark is movable (updated in place, like vel), zeroed by mpe_reset when
                        not NULL; ignored otherwise */
} mpe_buffers;

int mpe_abi_version(void);
const char *mpe_strerror(int code);
/* sizeof(mpe_scenario_desc) as compiled into the library (bindings check their struct layout). */
int64_t mpe_desc_sizeof(void);
int64_t mpe_buffers_sizeof(void);

/* Diagnostic (no reference counterpart): which state rows the fused one-env-per-lane register kernels (stock shapes of
 * every scenario) stream past the 256 MiB Infinity Cache at row stride `ld` (elements): bit 0 landmark positions, bit 1
 * velocities (command-hierarchy / moving-target kernels only), bit 2 leader_force, bit 3 the speaking agents' comm rows
 * (assuming the call passes a comm buffer).  While the whole state fits the cache
 * every state load of a step is a hit; well beyond it the read-only landmark rows are read with nontemporal loads so that
 * the rows a step reads AND writes keep their place, and the velocities follow when positions + velocities overflow it
 * while the positions alone still fit.  Cache policy only: results do not depend on it (tests compare the variants bit
 * for bit).  simple_tag f32, 4 Mi envs: 401 -> 326 us per step; S1L2F2_10Targets f64, 1 Mi envs: 450 -> 427.
 * Negative = error code. */
int mpe_residency_plan(const mpe_scenario_desc *desc, int64_t ld, int dtype_bytes);

/* Validates a descriptor (counts, positive masses/margin, what the kind supports). */
int mpe_desc_check(const mpe_scenario_desc *desc);

/* Observation geometry == len(scenario.observation(agent, world)) (environment.py:78);
 * negative = error code. */
int mpe_obs_dim(const mpe_scenario_desc *desc, int agent);
int mpe_obs_row0(const mpe_scenario_desc *desc, int agent);
int mpe_obs_total(const mpe_scenario_desc *desc);
int mpe_info_width(const mpe_scenario_desc *desc);
/* Action geometry: 5 entries for a movable agent + dim_c for a non-silent one (environment.py:42-76). */
int mpe_act_dim(const mpe_scenario_desc *desc, int agent);
int mpe_act_row0(const mpe_scenario_desc *desc, int agent);
int mpe_act_total(const mpe_scenario_desc *desc);
/* Rows of the goal buffer for this scenario (0: the scenario has no goal landmark). */
int mpe_goal_rows(const mpe_scenario_desc *desc);

/*
 * World.step for every env (core.py:124-140): MultiAgentEnv._set_action decode
 * (environment.py:157-208) -> apply_action_force (core.py:143-149) -> apply_leader_force
 * (core.py:153-194) -> apply_environment_force / get_collision_force (core.py:197-208, 233-249;
 * this fork's one-sided response) -> integrate_state (core.py:211-222) -> update_agent_state
 * (core.py:224-230).  Reads act, noise; updates pos, vel in place; writes lforce, comm.
 */
int mpe_world_step_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                       void *stream);
int mpe_world_step_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                       void *stream);

/*
 * Scenario gathers from the current state, for every agent of every env:
 * observation (simple_tag.py:131-147 / simple_spread.py:83-99), reward (simple_tag.py:84-129 /
 * simple_spread.py:71-81), benchmark_data (simple_tag.py:57-66 / simple_spread.py:46-62); for the
 * other kinds simple.py:41-50, simple_adversary.py:58-140, simple_push.py:66-104,
 * L1F2_3Targets_100observe.py:136-253.  Reads pos, vel, goal, lforce; writes whichever of obs /
 * rew / info is not NULL (MultiAgentEnv.reset() needs obs only, environment.py:119-129).
 */
int mpe_observe_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                    void *stream);
int mpe_observe_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                    void *stream);

/*
 * MultiAgentEnv.step (environment.py:92-117) in one launch: mpe_world_step followed by
 * mpe_observe on the post-step state, with the state kept on chip in between.
 * buf->stats (may be NULL): per-workgroup running sums of reward / collision statistics, double
 * [mpe_stats_rows(...)][mpe_stats_width(desc)], accumulated (+=) by this call; reduce with
 * mpe_stats_reduce.  Zero it with hipMemsetAsync before the first step of a window.
 */
int mpe_env_step_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                     void *stream);
int mpe_env_step_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                     void *stream);

/*
 * Scenario.reset_world (simple_tag.py:39-54 / simple_spread.py:30-44): agents ~ U(-ra, ra)^2,
 * landmarks ~ U(-rl, rl)^2, velocities and comm zero.  The reference draws from the global
 * np.random stream; this build draws Philox4x32-10 with counter (env_id_lo, env_id_hi, episode,
 * entity) and key = seed, env_id = env_id0 + index, so results do not depend on how envs are
 * sharded over GPUs.  `mask` (may be NULL): uint8 [B], only envs with mask != 0 are reset.
 * Writes buf->pos, vel, comm (if not NULL) and goal (if not NULL: the goal landmark of each env,
 * drawn uniformly from the same stream at entity word 0xFFFFFFFF; np.random.choice(
 * world.landmarks) in the reference).
 */
int mpe_reset_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, uint64_t seed,
                  uint32_t episode, uint64_t env_id0, const uint8_t *mask, const mpe_buffers *buf,
                  void *stream);
int mpe_reset_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, uint64_t seed,
                  uint32_t episode, uint64_t env_id0, const uint8_t *mask, const mpe_buffers *buf,
                  void *stream);

/*
 * target_move(env) of the fork's training loop (experiments/train_and_eval.py:207-222, called after
 * every env.step, :314,391): every landmark touched by an agent with advfollower set
 * (dist < size_l + size_a, strict) is respawned at U(-1, 1)^2.  The reference draws np.random.rand();
 * this build draws Philox4x32-10 with counter (env_id_lo, env_id_hi, move_counter, 0x80000000 |
 * landmark) and key = seed (same 53-bit uniform as mpe_reset).  Updates buf->pos in place (landmark
 * rows only); `count` (may be NULL): int32 [ld], the function's return value per env (number of
 * respawned landmarks).  Observations already returned by the step are not touched, like in the
 * reference.
 */
int mpe_target_move_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, uint64_t seed,
                        uint32_t move_counter, uint64_t env_id0, const mpe_buffers *buf, int32_t *count,
                        void *stream);
int mpe_target_move_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, uint64_t seed,
                        uint32_t move_counter, uint64_t env_id0, const mpe_buffers *buf, int32_t *count,
                        void *stream);

/*
 * Scripted prey of the fork's training loop (experiments/simple_train.py:157-205; the loop overwrites
 * action_n[-1] with it, :262-273): the 5 physical action entries of agent `agent` for every env.
 *   mode 0  good_agent_action: head for the corner of [-0.9, 0.9]^2 whose summed distance to all other
 *           agents is largest (first maximum; np.linalg.norm per agent, summed in agent order), gain 1
 *   mode 1  good_agent_victim_action: head for agent `target`, gain 7
 * act = min(|offset| / 2 * gain, cap) in the slot of the offset's sign (simple_train.py:175-187), cap =
 * 1.3 in the reference.  Reads pos; writes act_out [5][ld] (e.g. the agent's rows of the action buffer:
 * act + mpe_act_row0(desc, agent) * ld) and, when not NULL, point_out [2][ld] (the corner / target
 * position: the function's second return value).
 */
int mpe_scripted_action_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, int agent, int mode, int target,
                            double cap, const void *pos, void *act_out, void *point_out, void *stream);
int mpe_scripted_action_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, int agent, int mode, int target,
                            double cap, const void *pos, void *act_out, void *point_out, void *stream);

/*
 * The non-default decodes of MultiAgentEnv._set_action (multiagent/environment.py:176-192), applied to the SoA
 * action rows `act` ([mpe_act_total][ld], same layout as mpe_buffers.act) before a step call, which always runs the
 * default decode (:188-190).  One launch for all agents; agents that are not movable have no physical entries and
 * are skipped (:174).
 *   mpe_action_from_index  env.discrete_action_input (:176-182): idx is int32 [n_agents][ld_idx], one integer per agent
 *                          and env; 1 / 2 -> u[0] = -1 / +1, 3 / 4 -> u[1] = -1 / +1, any other value -> no force.  The
 *                          agent's 5 rows are written so that the default decode reproduces that u (as a one-hot INDEX 1
 *                          would mean +1: the rows are not one_hot(idx)).  A world with a speaking agent is refused
 *                          (MPE_ERR_UNSUPPORTED): the reference's integer decode of a comm action fails its own
 *                          `assert len(action) == 0` (:200-208).
 *   mpe_action_force_discrete  env.force_discrete_action, i.e. world.discrete_action (:33,184-187): the physical 5
 *                          entries of every movable agent become one_hot(np.argmax(.)) in place (first maximum; a
 *                          NaN wins); comm entries are left as they are.
 *   mpe_action_from_force  env.discrete_action_space == False (:191-192; marked "NEVER do here" by the fork, but a live
 *                          switch): the action of a movable agent IS the force command, agent.action.u = action[0], a
 *                          2-vector, times the sensitivity.  u is [n_agents][2][ld_u]; the agent's 5 rows become
 *                          (0, u_x, 0, u_y, 0), whose default decode returns u bit for bit.  force_discrete != 0: the
 *                          argmax -> one-hot of :184-187 runs first, on the 2-vector (first maximum, a NaN wins).  A
 *                          world with a movable agent that also speaks is refused (MPE_ERR_UNSUPPORTED): there the
 *                          reference takes the first 5 entries of the MultiDiscrete action as u and fails in
 *                          integrate_state (core.py:216, 2-vector += 5-vector).
 */
int mpe_action_from_force_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, int64_t ld_u, const void *u,
                              int force_discrete, void *act, void *stream);
int mpe_action_from_force_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, int64_t ld_u, const void *u,
                              int force_discrete, void *act, void *stream);
int mpe_action_from_index_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, int64_t ld_idx,
                              const int32_t *idx, void *act, void *stream);
int mpe_action_from_index_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, int64_t ld_idx,
                              const int32_t *idx, void *act, void *stream);
int mpe_action_force_discrete_f32(const mpe_scenario_desc *desc, int64_t B, int64_t ld, void *act, void *stream);
int mpe_action_force_discrete_f64(const mpe_scenario_desc *desc, int64_t B, int64_t ld, void *act, void *stream);

/*
 * Replay buffer of the learner (maddpg/maddpg/trainer/replay_buffer.py; one per agent, maddpg.py:147): a ring
 * of `capacity` transitions kept in HBM, transition-major: ring [capacity][width], one record =
 * (obs_t | action | reward | obs_tp1 | done) concatenated (width = sum of the segment rows).
 *
 * mpe_replay_add: ReplayBuffer.add (:25-32) for n transitions at once, = n add() calls in env order:
 * transition e lands in ring row (next_idx + e) % capacity; when n > capacity only the last `capacity`
 * survive, as in the reference.  The record is gathered from `nseg` (<= 8) source segments, element (row r,
 * transition e) of segment s at ptr[r * row_stride + e * env_stride] (SoA rows of the env kernels: row_stride
 * = ld, env_stride = 1; a contiguous [n][rows] array: row_stride = 1, env_stride = rows).  The caller keeps
 * len and next_idx: next_idx = (next_idx + n) % capacity, len = min(len + n, capacity).
 *
 * mpe_replay_gather: ReplayBuffer.sample_index (:34-44,54-55): out [batch][width] = ring[idx[b]]; idx is a
 * DEVICE array of int64 (indices are taken modulo capacity, so a bad index cannot fault).
 */
typedef struct mpe_replay_seg {
  const void *ptr;
  int32_t rows;
  int64_t row_stride;
  int64_t env_stride;
} mpe_replay_seg;

int mpe_replay_add_f32(int64_t capacity, int64_t next_idx, int64_t n, int width, int nseg,
                       const mpe_replay_seg *segs, void *ring, void *stream);
int mpe_replay_add_f64(int64_t capacity, int64_t next_idx, int64_t n, int width, int nseg,
                       const mpe_replay_seg *segs, void *ring, void *stream);
/* The same append for several buffers (one per agent: maddpg.py:147,154-156) in ONE launch: n transitions go into
 * every job's ring, each from its own segments.  Equal to n_jobs mpe_replay_add calls. */
typedef struct mpe_replay_job {
  int64_t capacity, next_idx;
  int32_t width, nseg;
  const mpe_replay_seg *segs;
  void *ring;
} mpe_replay_job;
int mpe_replay_add_multi_f32(int n_jobs, const mpe_replay_job *jobs, int64_t n, void *stream);
int mpe_replay_add_multi_f64(int n_jobs, const mpe_replay_job *jobs, int64_t n, void *stream);
/* The same with the moving part of every ring cursor in device memory: job i appends at
 * (jobs[i].next_idx + *cursors[i]) % capacity (cursors[i] may be NULL = 0; cursors itself is a host array of n_jobs
 * device pointers, read during the call).  No counterpart in the reference: a launch recorded in a HIP graph bakes its
 * arguments, so a replayed training loop (act -> step -> store, experiments/simple_train.py:257-300) keeps the words that
 * change from replay to replay in memory.  Buffers the one-launch kernel does not cover (records wider than 61 f32 / 30 f64
 * columns, fewer than 256 transitions) are refused with MPE_ERR_UNSUPPORTED when their cursor is not NULL -- before
 * anything is launched, like every error code < 0. */
int mpe_replay_add_multi_cur_f32(int n_jobs, const mpe_replay_job *jobs, const int64_t *const *cursors, int64_t n,
                                 void *stream);
int mpe_replay_add_multi_cur_f64(int n_jobs, const mpe_replay_job *jobs, const int64_t *const *cursors, int64_t n,
                                 void *stream);
int mpe_replay_gather_f32(int64_t capacity, int width, const int64_t *idx, int64_t batch, const void *ring,
                          void *out, void *stream);
int mpe_replay_gather_f64(int64_t capacity, int width, const int64_t *idx, int64_t batch, const void *ring,
                          void *out, void *stream);

/*
 * COMPACT replay records: obs_tp1 of a transition is obs_t of the same env's next transition, so a ring that is
 * appended to one step of n envs at a time can store it once.  The caller keeps
 *     ring [capacity][od + ad + 2]   records (obs_t | act | rew | done), appended with mpe_replay_add* (4 segments)
 *     ends [capacity][od]            obs_tp1 of the steps that have no successor in the ring -- the last step of an
 *                                    episode (the loop's `terminal`, experiments/simple_train.py:283,293-294) or the
 *                                    newest step when a sample is drawn --, appended with mpe_replay_add* (1 segment)
 *                                    at the same position as the step's records
 *     own  [capacity / n] uint8      per step slot: 1 = obs_tp1 is in `ends`, 0 = it is obs_t of row (i + n) % capacity
 * with capacity % n == 0 and every append exactly n transitions.  mpe_replay_gather_compact returns the rows in the FULL
 * layout of mpe_replay_gather, out [batch][2 od + ad + 2] = (obs_t | act | rew | obs_tp1 | done): the same bytes as a
 * full ring fed the same transitions (tests/test_gpu_replay.py), for 40 % fewer bytes per appended step.  No
 * counterpart in the reference (its buffer is a Python list of tuples, replay_buffer.py:25-32).
 */
int mpe_replay_gather_compact_f32(int64_t capacity, int od, int ad, int64_t n, const int64_t *idx, int64_t batch,
                                  const void *ring, const void *ends, const uint8_t *own, void *out, void *stream);
int mpe_replay_gather_compact_f64(int64_t capacity, int od, int ad, int64_t n, const int64_t *idx, int64_t batch,
                                  const void *ring, const void *ends, const uint8_t *own, void *out, void *stream);

/*
 * STEP STORE: replay without an append.  The step kernels already write everything a transition holds -- the actors
 * the action rows, mpe_env_step the reward rows and the NEXT observation rows -- so a caller that hands them a different
 * slot of
 *     obs  [slots][obs_rows][ld]    rew  [slots][rew_rows][ld]
 *     act  [slots][act_rows][ld]    done [slots][rew_rows][ld]  (or NULL: every done is 0, environment.py:108 without a
 *                                                                 done_callback)
 * at every step has its replay ring for free: transition (slot s, env e) is obs[s] | act[s] | rew[s] | obs[(s + 1) %
 * slots] | done[s].  A slot that only holds the last observation of an episode (the reset writes into the next one) has
 * no transition of its own; `live` (device, int32 [slots + 1]) lists the slots that do, oldest first, in its first
 * n_live entries and holds n_live once more in entry [slots] -- the kernel reads the count THERE, so that a gather
 * recorded in a HIP graph follows a store that grows between replays; the struct's n_live is the host's copy, used for
 * the argument checks.  Transition index i addresses env i % n of slot live[i / n] (indices outside [0, n_live n) wrap).  mpe_replay_gather_steps returns the rows of ONE agent
 * (its observation rows obs_row0 .. + od, action rows act_row0 .. + ad, reward row rew_row) in the full layout of
 * mpe_replay_gather, out [batch][2 od + ad + 2]: the same bytes as a full ring fed the same transitions
 * (tests/test_gpu_replay.py).  No counterpart in the reference (replay_buffer.py:25-32 appends Python tuples); host side:
 * marl_amd.replay_buffer.StepStore.
 */
typedef struct mpe_step_store {
  const void *obs, *act, *rew, *done;                  /* done may be NULL */
  const int32_t *live;                                 /* device, [slots + 1]: table, then the count */
  int64_t obs_slot_stride, act_slot_stride, rew_slot_stride; /* elements between consecutive slots */
  int64_t ld, n;                                       /* row stride; envs (transitions) per slot */
  int32_t slots, n_live, obs_row0, act_row0, rew_row, pad_;
} mpe_step_store;
int mpe_replay_gather_steps_f32(const mpe_step_store *store, int od, int ad, const int64_t *idx, int64_t batch, void *out,
                                void *stream);
int mpe_replay_gather_steps_f64(const mpe_step_store *store, int od, int ad, const int64_t *idx, int64_t batch, void *out,
                                void *stream);

/*
 * Actor forward of the learner for every env in one launch (maddpg/maddpg/trainer/maddpg.py:39-45,151-152;
 * model experiments/train_and_eval.py:94-101; sample common/distributions.py:199-201,247-251):
 *     act = softmax_per_sub_space(W3 relu(W2 relu(W1 obs + b1) + b2) + b3 - log(-log(u)))
 * with MPE_ACTOR_HIDDEN = 64 units per hidden layer and one (Discrete: s1 = 0) or two (MultiDiscrete) sub-spaces
 * of s0 / s1 logits, s0 + s1 <= MPE_ACTOR_MAX_OUT.  obs [in_dim][ld_obs] are the SoA observation rows of the
 * agent (mpe_obs_row0), act [s0 + s1][ld_act] SoA action rows (e.g. the agent's rows of the action buffer,
 * mpe_act_row0).  weights: packed f32 device array W1t [in_dim][64], b1 [64], W2t [64][64], b2 [64],
 * W3t [64][16], b3 [16] (t = transposed: input index first; columns >= s0 + s1 zero).  u: [s0 + s1][ld_act]
 * uniforms in (0, 1), or NULL = Philox4x32-10 with counter (env_id_lo, env_id_hi, call_counter, 0xC0000000 |
 * word group) and key = seed.  f32 only (the reference's networks are float32).
 */
#define MPE_ACTOR_HIDDEN 64
#define MPE_ACTOR_MAX_OUT 16
int mpe_actor_forward_f32(int64_t B, int64_t ld_obs, int64_t ld_act, int in_dim, int s0, int s1, const void *obs,
                          const void *weights, const void *u, uint64_t seed, uint32_t call_counter, uint64_t env_id0,
                          void *act, void *stream);

/* Every agent's actor in ONE launch (the reference loops over trainers, experiments/simple_train.py:260): job i reads
 * its own observation rows / weights / uniforms and writes its own action rows; B, the row strides, call_counter and
 * env_id0 are shared; job.seed keys the agent's Philox stream.  Equal to n_jobs mpe_actor_forward_f32 calls. */
typedef struct mpe_actor_job {
  const void *obs, *weights, *u;
  void *act;
  int32_t in_dim, s0, s1;
  uint64_t seed;
} mpe_actor_job;
#define MPE_ACTOR_AUTO 0 /* the library's choice: the matrix-core kernel (1.6-2.5x the vector one, 1 Ki .. 1 Mi envs) */
#define MPE_ACTOR_VALU 1 /* vector kernel: one env per lane, packed f32 FMAs, weights through the scalar cache */
#define MPE_ACTOR_MFMA 2 /* v_mfma_f32_32x32x16_bf16 on exact three-way bf16 splits of the f32 operands, six partial
                            products per product: the error of an f32 dot product; 32 envs per tile on the lane axis */
#define MPE_ACTOR_MFMA_F16 3 /* opt-in: the same on TWO f16 pieces per operand (22 significant bits) and three partial products,
                                v_mfma_f32_32x32x16_f16: ~4e-7 relative instead of f32's 6e-8, operands must lie inside f16's
                                range (|value| < 65 504; pieces below 6.1e-5 resolve 6e-8 absolute); 1.3x the launch rate */
int mpe_actor_forward_multi_f32(int n_jobs, const mpe_actor_job *jobs, int64_t B, int64_t ld_obs, int64_t ld_act,
                                uint32_t call_counter, uint64_t env_id0, int kernel, void *stream);
/* The same with a device word added to call_counter (counter_dev may be NULL = 0): for acting steps recorded in a HIP
 * graph, see mpe_replay_add_multi_cur_*.  No counterpart in the reference. */
int mpe_actor_forward_multi_ctr_f32(int n_jobs, const mpe_actor_job *jobs, int64_t B, int64_t ld_obs, int64_t ld_act,
                                    uint32_t call_counter, const uint32_t *counter_dev, uint64_t env_id0, int kernel,
                                    void *stream);

/*
 * Episode statistics (what the reference's loops sum on the host: rew_n per agent,
 * experiments/simple_train.py:284-286, and info_n per agent, :357).  Width = 2*A + 2 doubles:
 * [0,A) sum of reward per agent, [A,2A) sum of collisions per agent (info column 0 for
 * simple_tag, column 1 for simple_spread, 0 for the other kinds), [2A] sum over envs of colliding
 * (good,adversary) pairs (simple_tag) or occupied landmarks (simple_spread), [2A+1] number of
 * env-steps.
 * mpe_stats_rows = number of per-workgroup rows mpe_env_step_* accumulates into for this (desc,
 * B, dtype, flags): dtype_bytes is 4 or 8, flags = the mpe_buffers.flags of the stepping calls.  mpe_stats_reduce sums the rows in a fixed order (bitwise
 * reproducible) into out[width]; that vector is what ranks all-reduce over RCCL.
 */
int mpe_stats_width(const mpe_scenario_desc *desc);
int64_t mpe_stats_rows(const mpe_scenario_desc *desc, int64_t B, int dtype_bytes, uint32_t flags);
int mpe_stats_reduce(const double *stats, int64_t rows, int width, double *out, void *stream);

/*
 * NO COUNTERPART IN THE REFERENCE.  For callers that cannot allocate device memory (code under HIP graph
 * capture, tight serving loops, harnesses that forbid hipMalloc): mpe_env_step_{f32,f64} (dtype_bytes = 4 / 8)
 * followed by mpe_stats_reduce in one call, with the per-workgroup statistics rows -- the scratch the caller
 * otherwise allocates as buf->stats -- kept in a file-scope device array of MPE_STATIC_SCRATCH_BYTES inside the
 * library (every GPU the library runs on holds its own copy until the process ends).  buf->stats is ignored;
 * stats_out: device double [mpe_stats_width(desc)], OVERWRITTEN with this step's sums (same layout and the
 * same bits as mpe_stats_reduce over a zeroed buf->stats).  Calls no hipMalloc / hipFree / allocating routine.
 * Limit: mpe_stats_rows(desc, B, dtype_bytes, buf->flags) * mpe_stats_width(desc) * 8 <= MPE_STATIC_SCRATCH_BYTES
 * (simple_tag f32: B <= 3 355 392), else MPE_ERR_SCRATCH and nothing is launched.  All calls on one GPU share that
 * GPU's array: calls that overlap on one GPU (two streams, two threads) race; calls on one stream do not.
 * Under HIP graph capture: make the FIRST call on a device outside the capture (it looks the array's address up with
 * hipGetSymbolAddress, once per device; later calls only enqueue a memset and two kernels, all capturable).
 */
#define MPE_STATIC_SCRATCH_BYTES (1 << 20)
int mpe_env_step_stats_static(const mpe_scenario_desc *desc, int64_t B, int64_t ld, const mpe_buffers *buf,
                              int dtype_bytes, double *stats_out, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* MPE_HIP_H */
