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


// ------------------------------------------------------------------------------------------------------------------
// The same actor on the matrix cores: v_mfma_f32_32x32x2_f32 (f32 in, f32 accumulate: bit-for-bit an fmaf chain, so the
// precision of the vector kernel).  MI355X_MICROARCH.md measures f32 MFMA at 155 TFLOP/s sustained against 52 for a
// v_pk_fma_f32 GEMM, so round 1's "MFMA buys nothing" needed a measurement.  Measured (r2, 1 Mi envs, 16 -> 64 -> 64 -> 5,
// tools/actor_bench.py): this kernel 170.8 us (88 TFLOP/s executed = 0.56 of the f32 matrix peak, 14 % of it in the
// padded logits tile), the vector kernel above 158.1 us (73.6 TFLOP/s): a tie, the vector kernel stays the default
// (MPE_ACTOR_AUTO); this one is selectable (MPE_ACTOR_MFMA) and tested the same way.
//
// Transposed formulation, envs on the N (lane) axis:  H1^T = W1^T X^T,  H2^T = W2^T H1^T,  Y^T = W3^T H2^T.
// A tile is 32 envs.  D[row = hidden unit][col = env]: lane (n = l & 31, h = l >> 5) holds column n, register r holds row
// (r & 3) + 8 (r >> 2) + 4 h  -- exactly the B-operand shape of the NEXT layer's k-step r (B[k = h][n]), with the k pair
// (row(r, 0), row(r, 0) + 4): an accumulator register feeds the next MFMA as it is, no LDS, no lane movement; the A
// operand (weights) is fetched to match: lane (m, h) of k-step r supplies W[in = 32 kt + row(r, h)][out = 32 mt + m].
// W2 / W3 fragments stay in registers for all the wavefront's tiles (96 VGPRs), W1 / observations stream per tile.
// Per tile: 2 ceil(in/2) + 64 + 32 MFMAs of 64 cycles; the 16 unused rows of the logits tile are the waste (14 %).
// The relaxed sample's noise comes from the same Philox words as in the vector kernel (word group q = row / 4).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));

__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

template <int O>  // logits rows evaluated in the epilogue: 8 or 16
__global__ __launch_bounds__(256) void actor_mfma_kernel(int64_t B, int64_t ld_obs, int64_t ld_act, const ActorJobs jobs,
                                                         uint32_t counter, uint64_t env_id0, int tiles_per_wave) {
  constexpr int H = kActorHidden, OP = kActorMaxOut;
  static_assert(H == 64 && OP == 16, "two 32-row tiles of hidden units, one (half-used) tile of logits");
  __shared__ __align__(16) float sBias[2 * H + 32];  // b1 | b2 | b3 padded with zeros to 32 rows
  const ActorJob &job = jobs.j[blockIdx.y];
  const float *__restrict__ obs = job.obs, *__restrict__ wts = job.wts, *__restrict__ u = job.u;
  float *__restrict__ act = job.act;
  const int in_dim = job.in_dim, s0 = job.s0, nout = job.s0 + job.s1;
  const float *W1 = wts, *b1 = W1 + (int64_t)in_dim * H, *W2 = b1 + H;
  const float *b2 = W2 + H * H, *W3 = b2 + H, *b3 = W3 + H * OP;
  const int lane = threadIdx.x & 63, n = lane & 31, h = lane >> 5;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  if (threadIdx.x < 2 * H + 32)
    sBias[threadIdx.x] = threadIdx.x < H ? b1[threadIdx.x]
                                          : (threadIdx.x < 2 * H ? b2[threadIdx.x - H]
                                                                 : (threadIdx.x - 2 * H < OP ? b3[threadIdx.x - 2 * H] : 0.0f));
  // resident A fragments: W2 as [mt][kt][r], W3 as [kt][r]
  float a2[2][2][16], a3[2][16];
#pragma unroll
  for (int kt = 0; kt < 2; ++kt)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int k = 32 * kt + mfma_row(r, h);
      a2[0][kt][r] = W2[k * H + n];
      a2[1][kt][r] = W2[k * H + 32 + n];
      a3[kt][r] = n < OP ? W3[k * OP + n] : 0.0f;
    }
  __syncthreads();
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * tiles_per_wave;
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int64_t e0 = (tile0 + t) * 32;
    if (e0 >= B) break;  // wave-uniform
    const int64_t env = e0 + n;
    const bool active = env < B;
    const int64_t ge = active ? env : B - 1;  // tail lanes shadow the last env, their stores are skipped
    // ---- layer 1: acc1[mt] = b1 + W1^T X^T, k-steps of (2 s, 2 s + 1), eight at a time so the loads are batched.
    // (Fetching the next tile's first batch ahead of layers 2 and 3 was tried: 165.4 vs 170.8 us at 16 inputs, but 256
    // VGPRs and 24 % slower at 34 inputs; the two wavefronts per SIMD already cover each other's load waits.)
    f32x16 acc1[2], acc2[2], acc3;
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4v c0 = *reinterpret_cast<const f32x4v *>(sBias + 8 * g + 4 * h);
      const f32x4v c1 = *reinterpret_cast<const f32x4v *>(sBias + 32 + 8 * g + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc1[0][4 * g + q] = c0[q];
        acc1[1][4 * g + q] = c1[q];
      }
    }
    for (int sb = 0; sb < in_dim; sb += 16) {
      float xb[8], wa[8], wb[8];
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        const int k = sb + 2 * s + h;
        const bool ok = k < in_dim;
        const int kk = ok ? k : 0;
        const float x = ld_stream(obs + (int64_t)kk * ld_obs + ge);
        xb[s] = ok ? x : 0.0f;
        wa[s] = W1[kk * H + n];
        wb[s] = W1[kk * H + 32 + n];
      }
#pragma unroll
      for (int s = 0; s < 8; ++s) {
        if (sb + 2 * s < in_dim) {  // wave-uniform: a step with both k beyond the input is skipped
          acc1[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], xb[s], acc1[0], 0, 0, 0);
          acc1[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(wb[s], xb[s], acc1[1], 0, 0, 0);
        }
      }
    }
    // ---- layer 2: relu(acc1) registers are the B operands
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4v c0 = *reinterpret_cast<const f32x4v *>(sBias + H + 8 * g + 4 * h);
      const f32x4v c1 = *reinterpret_cast<const f32x4v *>(sBias + H + 32 + 8 * g + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        acc2[0][4 * g + q] = c0[q];
        acc2[1][4 * g + q] = c1[q];
      }
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const float b = fmaxf(acc1[kt][r], 0.0f);
        acc2[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[0][kt][r], b, acc2[0], 0, 0, 0);
        acc2[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a2[1][kt][r], b, acc2[1], 0, 0, 0);
      }
    // ---- layer 3: one tile of logits (rows >= 16 are padding)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4v c0 = *reinterpret_cast<const f32x4v *>(sBias + 2 * H + 8 * g + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc3[4 * g + q] = c0[q];
    }
#pragma unroll
    for (int kt = 0; kt < 2; ++kt)
#pragma unroll
      for (int r = 0; r < 16; ++r)
        acc3 = __builtin_amdgcn_mfma_f32_32x32x2f32(a3[kt][r], fmaxf(acc2[kt][r], 0.0f), acc3, 0, 0, 0);
    // ---- Gumbel-softmax per sub-space; this lane holds logits rows mfma_row(r, h), r < O / 2, of env n
    constexpr int R = O / 2;
    float lg[R];
    if (u != nullptr) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int m = mfma_row(r, h);
        lg[r] = m < nout ? ld_stream(u + (int64_t)m * ld_act + ge) : 0.5f;
      }
    } else {
      const uint64_t gid = env_id0 + (uint64_t)ge;
#pragma unroll
      for (int q2 = 0; q2 < R / 4; ++q2) {  // rows 8 q2 + 4 h .. + 3 = Philox word group 2 q2 + h
        uint32_t w[4];
        philox4x32_10((uint32_t)gid, (uint32_t)(gid >> 32), counter, 0xC0000000u | (uint32_t)(2 * q2 + h), job.k0, job.k1, w);
#pragma unroll
        for (int q = 0; q < 4; ++q) lg[4 * q2 + q] = philox_open01(w[q]);
      }
    }
    float mx0 = -INFINITY, mx1 = -INFINITY;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = mfma_row(r, h);
      lg[r] = acc3[r] - __logf(-__logf(lg[r]));
      mx0 = (m < s0 && lg[r] > mx0) ? lg[r] : mx0;
      mx1 = (m >= s0 && m < nout && lg[r] > mx1) ? lg[r] : mx1;
    }
    mx0 = fmaxf(mx0, __shfl_xor(mx0, 32, 64));  // the other half of the rows sits in lane l ^ 32
    mx1 = fmaxf(mx1, __shfl_xor(mx1, 32, 64));
    float z0 = 0.0f, z1 = 0.0f;
#pragma unroll
    for (int r = 0; r < R; ++r) {
      const int m = mfma_row(r, h);
      const float e = __expf(lg[r] - (m < s0 ? mx0 : mx1));
      lg[r] = e;
      z0 += m < s0 ? e : 0.0f;
      z1 += (m >= s0 && m < nout) ? e : 0.0f;
    }
    z0 += __shfl_xor(z0, 32, 64);
    z1 += __shfl_xor(z1, 32, 64);
    const float r0 = 1.0f / z0, r1 = 1.0f / z1;
    if (active) {
#pragma unroll
      for (int r = 0; r < R; ++r) {
        const int m = mfma_row(r, h);
        if (m < nout) st_stream(act + (int64_t)m * ld_act + env, lg[r] * (m < s0 ? r0 : r1));
      }
    }
  }
}

}  // namespace mpe
