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
// per env.  f32 because the reference's networks are.  This vector form is MPE_ACTOR_VALU; the default (MPE_ACTOR_AUTO)
// is the matrix-core form further down, which reaches the same accuracy from split bf16 operands.
// Packed weights (host side, marl_amd/maddpg.py:pack_actor): W1t [in][64], b1 [64], W2t [64][64],
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
  // optional device word added to the launch's call counter: a launch recorded in a HIP graph bakes its arguments,
  // so a replayed acting step takes the part of the counter that moves from memory (mpe_actor_forward_multi_ctr_f32)
  const uint32_t *counter_dev;
};

template <int O>  // output columns evaluated: 8 or 16 (>= s0 + s1 of every job); W3t / b3 keep their 16-column packing
__global__ __launch_bounds__(256) MPE_ACTOR_ATTR void actor_forward_kernel(int64_t B, int64_t ld_obs, int64_t ld_act,
                                                                           const ActorJobs jobs, uint32_t counter,
                                                                           uint64_t env_id0) {
  // blockIdx.y = agent: every agent of the step in ONE launch (wave-uniform job fields -> scalar loads)
  const ActorJob &job = jobs.j[blockIdx.y];
  if (jobs.counter_dev != nullptr) counter += *jobs.counter_dev;  // wave-uniform: a scalar load
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
// The same actor on the matrix cores, at f32 accuracy from bf16 MFMAs ("bf16x6").
//
// f32 MFMA has the vector rate (v_mfma_f32_32x32x2_f32: 256 flop / cycle / CU, the same as v_pk_fma_f32): round 2's first
// matrix-core form, built on it, tied with the vector kernel above (170.8 vs 158.1 us per agent at 1 Mi envs).  The bf16
// MFMA is 16x faster per k (v_mfma_f32_32x32x16_bf16: 32 cycles for 16 k), so every f32 operand is split EXACTLY into
// three bf16 pieces, x = hi + mid + lo (the next 8 significant bits each: truncation, so the residuals are exact and
// the range is f32's), and a product is the six partial products whose weight is >= 2^-16 of the leading one:
//     x w = hi hi + hi mid + mid hi + mid mid + hi lo + lo hi        (+ mid lo + lo mid + lo lo < 2^-23 |x w|, dropped)
// Each partial product is exact in the f32 accumulator (8 x 8 bits), so the result carries the error of an f32 dot
// product (tests: <= 1e-5 against PyTorch f32, the vector kernel's bar).  Six MFMAs of 32 cycles replace eight of 64.
// Measured (r2 MI355X, tools/actor_bench.py, 16 -> 64 -> 64 -> 5): 1 Mi envs 91.5 us vs 161.6 us for the vector kernel
// (34 inputs: 127.0 vs 228.6), 262 144 envs 25.7 vs 43.9, 4 096 envs 11.9 vs 20.1 (the vector kernel's floor is its
// 5 000-FMA dependent chain per lane).  What bounds it now is vector ISSUE, not the matrix pipe: per 32-env tile 84 MFMAs
// (2 688 matrix-pipe cycles, 8 issue cycles each) beside ~860 vector instructions -- the operand splits (5.5 per value:
// and, subtract, and, subtract, 1.5 permutes), relu as an integer max, Philox, the softmax; SQ counters: vector issue
// 62 % + MFMA issue 11 % of the SIMD cycles.  Accumulators stay in VGPRs (-mllvm -amdgpu-mfma-vgpr-form in the Makefile:
// no v_accvgpr_read per value, 160 + 64 -> 129 registers = 3 wavefronts per SIMD).
//
// Transposed formulation, envs on the N (lane) axis:  H1^T = W1^T X^T,  H2^T = W2^T H1^T,  Y^T = W3^T H2^T.
// A tile is 32 envs.  D[row = unit][col = env]: lane (n = l & 31, h = l >> 5) holds column n, register r holds row
// (r & 3) + 8 (r >> 2) + 4 h.  A k-block of the next layer is 16 consecutive units = accumulator registers 8 b .. 8 b + 7
// of both lane halves, and the MFMA sums over k in any order as long as A and B agree on it: B slot (h, j) of k-block b
// is register 8 (b & 1) + j of tile b >> 1 as it stands (relu, split, pack: no LDS, no lane movement), and the A operand
// is fetched to match: slot (h, j) of lane (m, h) is W[16 b + (j & 3) + 8 (j >> 2) + 4 h][m].  The pre-split fragments
// of W1 (up to 64 inputs), W2 and W3 sit in LDS (60 KiB; one conflict-free ds_read_b128 per piece): resident in
// registers, W2 alone is 96 VGPRs and leaves no room to fetch the next tile's observations ahead.
// The relaxed sample's noise comes from the same Philox words as in the vector kernel (word group q = row / 4).
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4v __attribute__((ext_vector_type(4)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

__device__ __forceinline__ int mfma_row(int r, int h) { return (r & 3) + 8 * (r >> 2) + 4 * h; }

// Eight f32 values -> three packed-bf16 operands (slot j = element j).  hi / mid / lo are the upper 16 bits of x, of
// x - hi and of x - hi - mid: both subtractions are exact, and so is the sum of the three pieces.
struct Bf16x3 {
  u32x4 hi, mid, lo;
};
__device__ __forceinline__ Bf16x3 split_bf16x3(const float (&x)[8]) {
  Bf16x3 o;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    // (plain subtractions: packed v_pk_add_f32 is an anti-lever beside MFMAs, MI355X_MICROARCH.md cycle constants)
    const uint32_t a0 = __float_as_uint(x[2 * i]), a1 = __float_as_uint(x[2 * i + 1]);
    const float r0 = x[2 * i] - __uint_as_float(a0 & 0xFFFF0000u), r1 = x[2 * i + 1] - __uint_as_float(a1 & 0xFFFF0000u);
    const uint32_t b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
    const float s0 = r0 - __uint_as_float(b0 & 0xFFFF0000u), s1 = r1 - __uint_as_float(b1 & 0xFFFF0000u);
    o.hi[i] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);  // (upper half of a1) : (upper half of a0)
    o.mid[i] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
    o.lo[i] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
  }
  return o;
}
// max(x, 0) in one instruction: as signed integers, negative floats (and -0) are below 0 and positive floats keep
// their order (fmaxf costs a second instruction that quiets signalling NaNs; a NaN stays a NaN here if its sign is +)
__device__ __forceinline__ float relu1(float x) {
  const int i = (int)__float_as_uint(x);
  return __uint_as_float((uint32_t)(i > 0 ? i : 0));
}

__device__ __forceinline__ f32x16 mfma_bf16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a), __builtin_bit_cast(bf16x8, b), c, 0, 0, 0);
}
// c += A B at f32 accuracy: smallest partial products first
__device__ __forceinline__ f32x16 mfma_bf16x6(const Bf16x3 &a, const Bf16x3 &b, f32x16 c) {
  c = mfma_bf16(a.lo, b.hi, c);
  c = mfma_bf16(a.hi, b.lo, c);
  c = mfma_bf16(a.mid, b.mid, c);
  c = mfma_bf16(a.mid, b.hi, c);
  c = mfma_bf16(a.hi, b.mid, c);
  return mfma_bf16(a.hi, b.hi, c);
}

// ---- MPE_ACTOR_MFMA_F16: the same layers from TWO f16 pieces per operand and three partial products ----------------
// x = p1 + p2 with p1 = x truncated to f16 (11 significant bits) and p2 = (x - p1) truncated to f16: the subtraction is
// exact, the pieces carry 22 bits, and x w = p1 q1 + p1 q2 + p2 q1 (+ terms below 2^-22 |x w|, dropped).  Each partial
// product is exact in the f32 accumulator.  2.0 vector instructions per split value instead of 5.5 and 3 MFMAs per
// product instead of 6 -- the loop is bound by vector issue -- at ~4e-7 relative accuracy instead
// of f32's 6e-8, for operands inside f16's range: |value| < 65 504 (beyond it a piece saturates: wrong, finite), and a
// second piece below 2^-14 = 6.1e-5 is a subnormal f16 with 2^-24 = 6e-8 ABSOLUTE resolution.  Opt-in for that reason.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2v __attribute__((ext_vector_type(2)));
__device__ __forceinline__ Bf16x3 split_f16x2(const float (&x)[8]) {
  Bf16x3 o;  // hi = p1, mid = p2, lo unused
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const f16x2v p1 = __builtin_bit_cast(f16x2v, __builtin_amdgcn_cvt_pkrtz(x[2 * i], x[2 * i + 1]));
    // x - p1 as one mixed-precision FMA that reads the f16 half in place (v_fma_mix_f32: p1.half * -1.0 + x, exact): no
    // conversion instruction (written as C the compiler folds it back into v_cvt_f32_f16 + v_sub_f32)
    const uint32_t pw = __builtin_bit_cast(uint32_t, p1);
    float r0, r1;
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel_hi:[1,0,0]" : "=v"(r0) : "v"(pw), "v"(x[2 * i]));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(r1) : "v"(pw), "v"(x[2 * i + 1]));
    o.hi[i] = pw;
    o.mid[i] = __builtin_bit_cast(uint32_t, __builtin_amdgcn_cvt_pkrtz(r0, r1));
    o.lo[i] = 0u;
  }
  return o;
}
__device__ __forceinline__ f32x16 mfma_f16(u32x4 a, u32x4 b, f32x16 c) {
  return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a), __builtin_bit_cast(f16x8, b), c, 0, 0, 0);
}
__device__ __forceinline__ f32x16 mfma_f16x3(const Bf16x3 &a, const Bf16x3 &b, f32x16 c) {
  c = mfma_f16(a.mid, b.hi, c);
  c = mfma_f16(a.hi, b.mid, c);
  return mfma_f16(a.hi, b.hi, c);
}
template <bool F16>
__device__ __forceinline__ Bf16x3 split_pieces(const float (&x)[8]) {
  if constexpr (F16) return split_f16x2(x);
  else return split_bf16x3(x);
}
template <bool F16>
__device__ __forceinline__ f32x16 mfma_pieces(const Bf16x3 &a, const Bf16x3 &b, f32x16 c) {
  if constexpr (F16) return mfma_f16x3(a, b, c);
  else return mfma_bf16x6(a, b, c);
}

// A fragment of a [K][ncols] row-major weight matrix for output column `col` (< 0: a zero column): slot j takes row k_of(j)
template <bool F16, typename KOf>
__device__ __forceinline__ Bf16x3 weight_fragment(const float *__restrict__ W, int K, int ncols, int col, KOf k_of) {
  float w[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int k = k_of(j);
    w[j] = (col >= 0 && k < K) ? W[(int64_t)k * ncols + col] : 0.0f;
  }
  return split_pieces<F16>(w);
}

constexpr int kActorLdsInputs = 64;  // W1 rows kept pre-split in LDS; inputs beyond that are split per tile

// O: logits rows evaluated in the epilogue, 8 or 16.  NKB: k-blocks of 16 observation rows, 1..3 compiled in (the tile
// loop is then one straight block: next tile's rows fetched ahead, LDS reads scheduled across the MFMAs), 0 = any width.
template <int O, int NKB, bool F16 = false>
__global__ __launch_bounds__(256) void actor_mfma_kernel(int64_t B, int64_t ld_obs, int64_t ld_act, const ActorJobs jobs,
                                                         uint32_t counter, uint64_t env_id0, int tiles_per_wave) {
  constexpr int H = kActorHidden, OP = kActorMaxOut;
  static_assert(H == 64 && OP == 16, "two 32-row tiles of hidden units, one (half-used) tile of logits");
  constexpr int NKB1 = NKB > 0 ? NKB : kActorLdsInputs / 16;  // W1 k-blocks held in LDS
  __shared__ __align__(16) float sBias[2 * H + 32];  // b1 | b2 | b3 padded with zeros to 32 rows
  __shared__ u32x4 sW1[2 * NKB1 * 3][64];             // [(mt, kb), piece][lane]
  __shared__ u32x4 sW2[2 * 4 * 3][64];                // [(mt, kb), piece][lane]
  __shared__ u32x4 sW3[4 * 3][64];                    // [kb, piece][lane]
  const ActorJob &job = jobs.j[blockIdx.y];
  if (jobs.counter_dev != nullptr) counter += *jobs.counter_dev;  // wave-uniform: a scalar load
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
  const int nkb1 = NKB > 0 ? NKB : (in_dim + 15) / 16;
  auto k_in = [&](int kb) { return [=](int j) { return 16 * kb + 8 * h + j; }; };                   // layer 1: natural order
  auto k_acc = [&](int kb) { return [=](int j) { return 16 * kb + (j & 3) + 8 * (j >> 2) + 4 * h; }; };  // layers 2, 3
  for (int f = wave; f < 2 * NKB1; f += 4) {  // W1 fragments (mt = f & 1, kb = f >> 1) of the first 64 inputs
    const int kb = f >> 1;
    if (kb < nkb1) {
      const Bf16x3 a = weight_fragment<F16>(W1, in_dim, H, 32 * (f & 1) + n, k_in(kb));
      sW1[3 * f + 0][lane] = a.hi;
      sW1[3 * f + 1][lane] = a.mid;
      sW1[3 * f + 2][lane] = a.lo;
    }
  }
  {  // W3 fragment of k-block `wave`
    const Bf16x3 a = weight_fragment<F16>(W3, H, OP, n < OP ? n : -1, k_acc(wave));
    sW3[3 * wave + 0][lane] = a.hi;
    sW3[3 * wave + 1][lane] = a.mid;
    sW3[3 * wave + 2][lane] = a.lo;
  }
  for (int f = wave; f < 8; f += 4) {  // W2 fragments (mt = f & 1, kb = f >> 1)
    const Bf16x3 a = weight_fragment<F16>(W2, H, H, 32 * (f & 1) + n, k_acc(f >> 1));
    sW2[3 * f + 0][lane] = a.hi;
    sW2[3 * f + 1][lane] = a.mid;
    sW2[3 * f + 2][lane] = a.lo;
  }
  __syncthreads();
  // Observation rows of k-block kb for the tile at env e0: B slot (h, j) is row 16 kb + 8 h + j.  Rows beyond in_dim
  // meet zero weight rows; they read row 0 (a finite observation x 0, where the vector kernel reads nothing).
  // NKB > 0: the row offsets are per-lane constants (32-bit byte offsets: the host checks in_dim ld_obs < 2^30) on a
  // wave-uniform tile base; lanes of the last tile that lie beyond B step back to env B - 1.
  constexpr int PFA = NKB > 0 ? NKB : 1;
  uint32_t roff[PFA][8];
#pragma unroll
  for (int kb = 0; kb < PFA; ++kb)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int k = 16 * kb + 8 * h + j;
      roff[kb][j] = (uint32_t)(((int64_t)(k < in_dim ? k : 0) * ld_obs + n) * 4);
    }
  auto load_obs = [&](int kb, int64_t e0, float (&x)[8]) {
    if (NKB > 0) {
      const char *base = reinterpret_cast<const char *>(obs + e0);  // wave-uniform
      const int64_t over = e0 + n - (B - 1);
      const uint32_t back = over > 0 ? (uint32_t)over * 4u : 0u;
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = ld_stream(reinterpret_cast<const float *>(base + (roff[kb < PFA ? kb : 0][j] - back)));
    } else {
      const int64_t e = e0 + n < B ? e0 + n : B - 1;
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const int k = 16 * kb + 8 * h + j;
        x[j] = ld_stream(obs + (int64_t)(k < in_dim ? k : 0) * ld_obs + e);
      }
    }
  };
  // The first PF k-blocks of the NEXT tile are fetched while this one is computed: with two wavefronts per SIMD and
  // ~3 500 cycles of work per tile, an exposed HBM round trip per tile would cost about as much again.
  constexpr int PF = NKB;
  const int64_t tile0 = ((int64_t)blockIdx.x * 4 + wave) * tiles_per_wave;
  float xc[PFA][8], xn[PFA][8];
  if (tile0 * 32 < B) {
#pragma unroll
    for (int kb = 0; kb < PF; ++kb)
      if (kb < nkb1) load_obs(kb, tile0 * 32, xc[kb]);
  }
  for (int t = 0; t < tiles_per_wave; ++t) {
    const int64_t e0 = (tile0 + t) * 32;
    if (e0 >= B) break;  // wave-uniform
    const int64_t env = e0 + n;
    const bool active = env < B;
    const int64_t ge = active ? env : B - 1;  // tail lanes shadow the last env, their stores are skipped
    if (t + 1 < tiles_per_wave && e0 + 32 < B) {
#pragma unroll
      for (int kb = 0; kb < PF; ++kb)
        if (kb < nkb1) load_obs(kb, e0 + 32, xn[kb]);
    }
    // ---- layer 1: acc1[mt] = b1 + W1^T X^T
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
    auto layer1_block = [&](int kb, const float (&x)[8]) {
      const Bf16x3 b = split_pieces<F16>(x);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        Bf16x3 a;
        if (kb < NKB1) {  // wave-uniform
          const int f = 2 * kb + mt;
          a.hi = sW1[3 * f + 0][lane];
          a.mid = sW1[3 * f + 1][lane];
          if constexpr (!F16) a.lo = sW1[3 * f + 2][lane];
        } else {
          a = weight_fragment<F16>(W1, in_dim, H, 32 * mt + n, k_in(kb));
        }
        acc1[mt] = mfma_pieces<F16>(a, b, acc1[mt]);
      }
    };
#pragma unroll
    for (int kb = 0; kb < PF; ++kb)
      if (kb < nkb1) layer1_block(kb, xc[kb]);
    for (int kb = PF; kb < nkb1; ++kb) {  // wide observations: the rest is fetched here
      float x[8];
      load_obs(kb, e0, x);
      layer1_block(kb, x);
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
    for (int kb = 0; kb < 4; ++kb) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = relu1(acc1[kb >> 1][8 * (kb & 1) + j]);
      const Bf16x3 b = split_pieces<F16>(x);
#pragma unroll
      for (int mt = 0; mt < 2; ++mt) {
        const int f = 2 * kb + mt;
        Bf16x3 a;
        a.hi = sW2[3 * f + 0][lane];
        a.mid = sW2[3 * f + 1][lane];
        if constexpr (!F16) a.lo = sW2[3 * f + 2][lane];
        acc2[mt] = mfma_pieces<F16>(a, b, acc2[mt]);
      }
    }
    // ---- layer 3: one tile of logits (rows >= 16 are padding)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const f32x4v c0 = *reinterpret_cast<const f32x4v *>(sBias + 2 * H + 8 * g + 4 * h);
#pragma unroll
      for (int q = 0; q < 4; ++q) acc3[4 * g + q] = c0[q];
    }
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
      float x[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) x[j] = relu1(acc2[kb >> 1][8 * (kb & 1) + j]);
      const Bf16x3 b = split_pieces<F16>(x);
      Bf16x3 a;
      a.hi = sW3[3 * kb + 0][lane];
      a.mid = sW3[3 * kb + 1][lane];
      if constexpr (!F16) a.lo = sW3[3 * kb + 2][lane];
      acc3 = mfma_pieces<F16>(a, b, acc3);
    }
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
#pragma unroll
    for (int kb = 0; kb < PF; ++kb)
#pragma unroll
      for (int j = 0; j < 8; ++j) xc[kb][j] = xn[kb][j];
  }
}

}  // namespace mpe
