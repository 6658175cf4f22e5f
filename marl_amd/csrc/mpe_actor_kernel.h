// Fused actor forward of the MADDPG learner (maddpg/maddpg/trainer/maddpg.py:39-45,151-152 with the model of
// experiments/train_and_eval.py:94-101 and the sample of common/distributions.py:199-201,247-251):
//
//     logits = W3 relu(W2 relu(W1 obs + b1) + b2) + b3        two hidden layers of 64 units
//     action = softmax(logits - log(-log(u))) per sub-space    u ~ U(0, 1): Gumbel-softmax relaxation
//
// for every env in one launch, reading the env kernels' SoA observation rows and writing SoA action rows (the
// rows env_step reads next), so an acting step moves 4 (in + out) bytes per env and nothing else.  One env per
// lane: the 64 hidden activations live in VGPRs, the weights are wave-uniform and come through the scalar
// cache (s_load -> SGPR operand of v_fmac), so the inner loops are pure VALU: 2 (in 64 + 64 64 + 64 out) flop
// per env.  Dense but tiny: at 64 units the whole problem is 11 GFLOP per million envs -- f32 VALU work for
// ~0.1 ms, not worth an MFMA tiling (fp32 MFMA has the vector rate anyway); f32 because the reference's
// networks are.  Packed weights (host side, marl_amd/maddpg.py:pack_actor): W1t [in][64], b1 [64], W2t [64][64],
// b2 [64], W3t [64][16], b3 [16], output columns beyond s0 + s1 zero.
#pragma once
#include "mpe_device.h"

namespace mpe {

constexpr int kActorHidden = 64;
constexpr int kActorMaxOut = 16;

#ifndef MPE_ACTOR_ATTR
#define MPE_ACTOR_ATTR
#endif
// 23 random bits -> a uniform strictly inside (0, 1): (n + 0.5) * 2^-23 is exact in f32 for n < 2^23 (24 bits:
// n = 2^24 - 1 would round (n + 0.5) up to 2^24 and return exactly 1.0 -> -log(-log(1)) = +inf -> NaN action).
// min 2^-24, max 1 - 2^-24.
__device__ __forceinline__ float philox_open01(uint32_t w) { return ((float)(w >> 9) + 0.5f) * (1.0f / 8388608.0f); }

// One agent's actor evaluation: where its observation rows, packed weights, optional uniforms and action rows are.
struct ActorJob {
  const float *obs, *wts, *u;
  float *act;
  int32_t in_dim, s0, s1;
  uint32_t k0, k1;  // Philox key = the agent's seed
};
constexpr int kActorMaxJobs = 16;  // agents per launch (blockIdx.y); more agents = more launches
struct ActorJobs {
  ActorJob j[kActorMaxJobs];
};

template <int O>  // output columns evaluated: 8 or 16 (>= s0 + s1 of every job); W3t / b3 keep their 16-column packing
__global__ __launch_bounds__(256) MPE_ACTOR_ATTR void actor_forward_kernel(int64_t B, int64_t ld_obs, int64_t ld_act,
                                                                           const ActorJobs jobs, uint32_t counter,
                                                                           uint64_t env_id0) {
  // blockIdx.y = agent: every agent of the step in ONE launch (wave-uniform job fields -> scalar loads)
  const ActorJob &job = jobs.j[blockIdx.y];
  const float *__restrict__ obs = job.obs, *__restrict__ wts = job.wts, *__restrict__ u = job.u;
  float *__restrict__ act = job.act;
  const int in_dim = job.in_dim, s0 = job.s0, s1 = job.s1;
  const uint32_t k0 = job.k0, k1 = job.k1;
  constexpr int H = kActorHidden, OP = kActorMaxOut;
  const int64_t env = (int64_t)blockIdx.x * 256 + threadIdx.x;
  const bool active = env < B;
  const int64_t ge = active ? env : B - 1;  // tail lanes shadow the last env, their stores are skipped
  const float *W1 = wts, *b1 = W1 + (int64_t)in_dim * H, *W2 = b1 + H;
  const float *b2 = W2 + H * H, *W3 = b2 + H, *b3 = W3 + H * OP;
  typedef float f2 __attribute__((ext_vector_type(2)));  // packed f32 FMAs (v_pk_fma_f32): two units per instruction
  float h1[H], out[O];
  {
    f2 h1p[H / 2];
#pragma unroll
    for (int q = 0; q < H / 2; ++q) h1p[q] = f2{b1[2 * q], b1[2 * q + 1]};
    for (int i = 0; i < in_dim; ++i) {  // one coalesced row load per input, 32 packed FMAs with scalar weights
      const float x = ld_stream(obs + (int64_t)i * ld_obs + ge);
      const f2 xx = f2{x, x};
      const float *w = W1 + (int64_t)i * H;
#pragma unroll
      for (int q = 0; q < H / 2; ++q) h1p[q] = __builtin_elementwise_fma(f2{w[2 * q], w[2 * q + 1]}, xx, h1p[q]);
    }
#pragma unroll
    for (int q = 0; q < H / 2; ++q) {
      h1[2 * q] = h1p[q].x;
      h1[2 * q + 1] = h1p[q].y;
    }
  }
#pragma unroll
  for (int j = 0; j < H; ++j) h1[j] = h1[j] > 0.0f ? h1[j] : 0.0f;
#pragma unroll
  for (int m = 0; m < O; ++m) out[m] = b3[m];
  // layer 2 in blocks of KB output units (j unrolled: static h1 registers), each block folded into layer 3 at
  // once, so h2 never exists as an array; the block loop stays a real loop, which bounds what the scheduler
  // may hoist (fully unrolled, it pulls all 5 200 scalar weight loads to the top and spills 5 000 SGPRs).
  // Measured, 1 Mi envs, 16 inputs: weights as LDS broadcast reads 424 us (the LDS pipe, shared by the four
  // SIMDs, is the bound), as scalar loads 199 us, scalar loads + packed FMAs 185 us.
  constexpr int KB = 8;
  f2 out2[O / 2];
#pragma unroll
  for (int m = 0; m < O / 2; ++m) out2[m] = f2{out[2 * m], out[2 * m + 1]};
#pragma unroll 1
  for (int kb = 0; kb < H; kb += KB) {
    f2 acc[KB / 2];
#pragma unroll
    for (int q = 0; q < KB / 2; ++q) acc[q] = f2{b2[kb + 2 * q], b2[kb + 2 * q + 1]};
#pragma unroll
    for (int j = 0; j < H; ++j) {
      const f2 hj = f2{h1[j], h1[j]};
#pragma unroll
      for (int q = 0; q < KB / 2; ++q)
        acc[q] = __builtin_elementwise_fma(f2{W2[j * H + kb + 2 * q], W2[j * H + kb + 2 * q + 1]}, hj, acc[q]);
    }
#pragma unroll
    for (int kk = 0; kk < KB; ++kk) {
      const float v = (kk & 1) ? acc[kk / 2].y : acc[kk / 2].x;
      const float a = v > 0.0f ? v : 0.0f;
      const f2 aa = f2{a, a};
#pragma unroll
      for (int m = 0; m < O / 2; ++m)
        out2[m] = __builtin_elementwise_fma(f2{W3[(kb + kk) * OP + 2 * m], W3[(kb + kk) * OP + 2 * m + 1]}, aa, out2[m]);
    }
  }
#pragma unroll
  for (int m = 0; m < O / 2; ++m) {
    out[2 * m] = out2[m].x;
    out[2 * m + 1] = out2[m].y;
  }
  // Gumbel noise: caller-supplied uniforms (tests, replaying a recorded draw) or Philox4x32-10 keyed like the
  // reset stream: counter (env_lo, env_hi, call counter, 0xC0000000 | word group), key = seed
  const int nout = s0 + s1;
  float g[O];
  if (u != nullptr) {
#pragma unroll
    for (int m = 0; m < O; ++m) g[m] = m < nout ? ld_stream(u + (int64_t)m * ld_act + ge) : 0.5f;
  } else {
    const uint64_t gid = env_id0 + (uint64_t)ge;
#pragma unroll
    for (int q = 0; q < O / 4; ++q) {
      uint32_t w[4];
      philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), counter, 0xC0000000u | (uint32_t)q, k0, k1, w);
#pragma unroll
      for (int r = 0; r < 4; ++r) g[4 * q + r] = philox_open01(w[r]);
    }
  }
  float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
  for (int m = 0; m < O; ++m) {
    out[m] = out[m] - __logf(-__logf(g[m]));
    mx0 = (m < s0 && out[m] > mx0) ? out[m] : mx0;
    mx1 = (m >= s0 && m < nout && out[m] > mx1) ? out[m] : mx1;
  }
  float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
  for (int m = 0; m < O; ++m) {
    const float e = __expf(out[m] - (m < s0 ? mx0 : mx1));
    out[m] = e;
    z0 += m < s0 ? e : 0.0f;
    z1 += (m >= s0 && m < nout) ? e : 0.0f;
  }
  const float r0 = 1.0f / z0, r1 = 1.0f / z1;
  if (active) {
#pragma unroll
    for (int m = 0; m < O; ++m)
      if (m < nout) st_stream(act + (int64_t)m * ld_act + env, out[m] * (m < s0 ? r0 : r1));
  }
}

}  // namespace mpe
