// Shared device helpers for the gfx950 (MI355X, CDNA4) VQA kernels.  Wave = 64 lanes.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <stdlib.h>

// Measurement / ablation switches (tile choices, and the VQA_*_DBG phase switches that give WRONG results on purpose) are
// environment variables only in builds made with -DVQA_ABLATION (tools/build_ablation.py makes its own libvqa_hip_ablation.so).
// The product library ignores the environment: every switch is its shipped default, folded at compile time.
#ifdef VQA_ABLATION
static inline int vqa_env_int(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#else
static inline int vqa_env_int(const char*, int dflt) { return dflt; }
#endif

#define VQA_OK 0
#define VQA_EARG 1000   // argument / shape error (never launches)

typedef uint16_t bf16_t;   // raw bfloat16 bits
typedef __attribute__((ext_vector_type(4))) float f32x4;
typedef __attribute__((ext_vector_type(8))) __bf16 bf16x8;
typedef __attribute__((ext_vector_type(4))) short i16x4;
typedef __attribute__((ext_vector_type(4))) uint32_t u32x4;

__device__ __forceinline__ float bf2f(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }
__device__ __forceinline__ bf16_t f2bf(float f) {   // RNE, NaN stays NaN (v_cvt_pk_bf16_f32)
  __bf16 b = (__bf16)f;
  return __builtin_bit_cast(bf16_t, b);
}
template <typename T> __device__ __forceinline__ float to_f(T v);
template <> __device__ __forceinline__ float to_f<float>(float v) { return v; }
template <> __device__ __forceinline__ float to_f<bf16_t>(bf16_t v) { return bf2f(v); }
template <typename T> __device__ __forceinline__ T from_f(float v);
template <> __device__ __forceinline__ float from_f<float>(float v) { return v; }
template <> __device__ __forceinline__ bf16_t from_f<bf16_t>(float v) { return f2bf(v); }

// 16-byte vector of T (4 floats or 8 bf16) with element access as float.
template <typename T> struct Vec16;
template <> struct Vec16<float> {
  static constexpr int N = 4;
  u32x4 raw;
  __device__ __forceinline__ float get(int i) const { return __uint_as_float(raw[i]); }
  __device__ __forceinline__ void set(int i, float v) { raw[i] = __float_as_uint(v); }
};
template <> struct Vec16<bf16_t> {
  static constexpr int N = 8;
  u32x4 raw;
  __device__ __forceinline__ float get(int i) const {
    uint32_t w = raw[i >> 1];
    return __uint_as_float((i & 1) ? (w & 0xffff0000u) : (w << 16));
  }
  __device__ __forceinline__ void set(int i, float v) {
    uint32_t b = f2bf(v);
    uint32_t w = raw[i >> 1];
    raw[i >> 1] = (i & 1) ? ((w & 0x0000ffffu) | (b << 16)) : ((w & 0xffff0000u) | b);
  }
};
template <typename T> __device__ __forceinline__ Vec16<T> ldg16(const T* p) {
  Vec16<T> v; v.raw = *reinterpret_cast<const u32x4*>(p); return v;
}
template <typename T> __device__ __forceinline__ void stg16(T* p, const Vec16<T>& v) {
  *reinterpret_cast<u32x4*>(p) = v.raw;
}
template <typename T> __device__ __forceinline__ Vec16<T> zero16() {
  Vec16<T> v; v.raw = u32x4{0u, 0u, 0u, 0u}; return v;
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// Counter-based dropout RNG: keep(seed, idx) is a pure function so backward regenerates the mask.
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16; return x;
}
// Fast form for tensors below 2^32 elements (every tensor of this model): the seed-only half of the hash is computed once per
// kernel (drop_key) and each element costs ONE mix32 instead of two; bit-identical to drop_keep for idx < 2^32.
__device__ __forceinline__ uint32_t drop_key(uint64_t seed) { return mix32((uint32_t)seed) ^ (uint32_t)(seed >> 32) * 0x9E3779B9u; }
__device__ __forceinline__ bool drop_keep32(uint32_t key, uint32_t idx, float p) {
  const uint32_t h = mix32(idx ^ key);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}
__device__ __forceinline__ bool drop_keep(uint64_t seed, uint64_t idx, float p) {
  uint32_t h = mix32((uint32_t)idx ^ mix32((uint32_t)(idx >> 32) + (uint32_t)seed) ^ (uint32_t)(seed >> 32) * 0x9E3779B9u);
  return (float)(h >> 8) * (1.0f / 16777216.0f) >= p;
}

// ------------------------------------------------------------------------------------------------------------------
// Order-independent accumulation of fp32 partial sums across workgroups: FIXED POINT in 64-bit integer atomics.  Integer addition
// commutes, so the total is bit-identical whatever order the workgroups arrive in (float atomics are not: the train step is
// bit-reproducible, tests/test_gpu_reproducible.py), and the consumer kernel can read the finished sums in its prologue -- no
// slab, no separate finalize launch between producer and consumer (60 such launches sat on the critical path of a train step).
//
// A partial (one workgroup's fp32 sum v) is split EXACTLY into  v = hi/2^4 + lo,  hi = rint(16 v) an integer, |lo| <= 2^-5, and the
// two halves go to two integer planes: hi as it is, lo scaled by 2^50.  Range and resolution are therefore independent:
//   |v| < 2^41 (2.2e12) per partial, resolution 2^-50 (8.9e-16) -- both BatchNorm statistics (sum y, sum y^2) and the BatchNorm
//   backward sums (sum g, sum g*xhat; under GradScaler these are multiplied by a loss scale that doubles every 2000 clean steps,
//   training/train.py:179-195) fit with one format.
// The TOTAL cannot wrap: with at most VQA_ACC_MAX_PARTS = 2^17 partials per sum (the launchers refuse larger grids) the hi plane
// stays below 2^45 * 2^17 = 2^62 and the lo plane below 2^45 * 2^17 as well, replicas included.  (Round 3 range-checked only the
// partial: ~96 partials just under the old 2^22 backward limit wrapped the int64 total silently -- finite and wrong.)
// A partial outside the range, or NaN / inf, raises the flag word; the consumer then turns the statistics into NaN (loud: under AMP
// the GradScaler sees non-finite gradients, skips the step and backs the scale off, as it would for an fp16 overflow).
// ------------------------------------------------------------------------------------------------------------------
// Replicas: a sum lives in R copies (producer workgroup w adds to copy w % R, the consumer adds the copies as integers) because
// atomics on ONE address are serialised at ~21 ns each on MI355X (tools/atomic_bench.hip: 768 workgroups x 256 addresses 57 us with
// one copy, 4 us with eight); R = 512 / C clamped to 1..8 keeps R*C constant, so a consumer prologue reads the same few KB whatever C.
//   layout (n = R*K*C):  hi plane acc[(r*K + k)*C + c] | flag acc[n] | lo plane acc[n + 1 + (r*K + k)*C + c];
//   vqa_bn_acc_words(K, C) = 2n + 1 rounded up to even, caller-zeroed.
static inline __host__ __device__ int acc_replicas(int C) { const int r = 512 / (C > 0 ? C : 1); return r < 1 ? 1 : (r > 8 ? 8 : r); }
#define VQA_ACC_HI_SHIFT 4
#define VQA_ACC_LO_SHIFT 50
#define VQA_ACC_MAX_PARTS (1 << 17)
#define VQA_ACC_LIMIT_LOG2 (62 - VQA_ACC_HI_SHIFT - 17)       // 41
// idx = (r*K + k)*C + c,  n = R*K*C
__device__ __forceinline__ void acc_add_fixed(unsigned long long* acc, size_t n, size_t idx, float v) {
  constexpr float lim = (float)(1ull << VQA_ACC_LIMIT_LOG2), hs = (float)(1 << VQA_ACC_HI_SHIFT), lscale = (float)(1ull << VQA_ACC_LO_SHIFT);
  if (fabsf(v) < lim) {                                                          // (NaN fails the comparison too)
    const float hi = rintf(v * hs);                                              // integer, |hi| < 2^45: the conversion below is exact
    const float lo = __builtin_fmaf(hi, -1.f / hs, v);                           // exact: a multiple of ulp(v) no larger than 2^-5
    atomicAdd(acc + idx, (unsigned long long)(long long)hi);
    if (lo != 0.f) atomicAdd(acc + n + 1 + idx, (unsigned long long)__float2ll_rn(lo * lscale));
  } else atomicAdd(acc + n, 1ull);
}
// sum k of channel c over the R <= 8 replicas (integer addition: exact, order-free), as a double.  All loads are issued before
// the first add: a serial loop over a runtime R is a chain of dependent L2 latencies (24 of them cost a consumer 12 us).
__device__ __forceinline__ double acc_read_fixed(const unsigned long long* acc, int R, int K, int C, int k, int c) {
  const size_t n = (size_t)R * K * C;
  long long h[8], l[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) {
    const size_t i = ((size_t)r * K + k) * C + c;
    h[r] = r < R ? (long long)acc[i] : 0ll;
    l[r] = r < R ? (long long)acc[n + 1 + i] : 0ll;
  }
  const long long th = ((h[0] + h[1]) + (h[2] + h[3])) + ((h[4] + h[5]) + (h[6] + h[7]));
  const long long tl = ((l[0] + l[1]) + (l[2] + l[3])) + ((l[4] + l[5]) + (l[6] + l[7]));
  return (double)th * (1.0 / (double)(1 << VQA_ACC_HI_SHIFT)) + (double)tl * (1.0 / (double)(1ull << VQA_ACC_LO_SHIFT));
}
__device__ __forceinline__ bool acc_flagged(const unsigned long long* acc, int R, int K, int C) { return acc[(size_t)R * K * C] != 0; }

struct BnAcc {
  const unsigned long long* acc; const float* gamma; const float* beta; float* rm; float* rv; long long* nbt; float* coef;
};
// One evaluation per channel and workgroup (the results are shared through LDS), fp64 only for the cancellation-prone
// var = E[y^2] - mean^2; 1/count arrives precomputed and 1/sqrt is the fp32 hardware instruction -- a first version that divided
// and took square roots in fp64 in every thread cost 3x the streaming work of the kernel.
__device__ __forceinline__ void bn_acc_coef(const BnAcc& f, int C, int c, double inv_count, double unbias, float momentum, float eps, bool writer,
                                            float& sc, float& sh) {
  const int R = acc_replicas(C);
  const double s = acc_read_fixed(f.acc, R, 2, C, 0, c), q = acc_read_fixed(f.acc, R, 2, C, 1, c);
  double mean = s * inv_count;
  double var = q * inv_count - mean * mean;
  if (var < 0.0) var = 0.0;
  if (acc_flagged(f.acc, R, 2, C)) mean = __builtin_nan("");                 // a partial sum left the fixed-point range / was not finite
  const float invstd = rsqrtf((float)var + eps);
  sc = f.gamma[c] * invstd; sh = f.beta[c] - (float)mean * sc;
  if (writer) {
    f.coef[c] = sc; f.coef[C + c] = sh; f.coef[2 * C + c] = (float)mean; f.coef[3 * C + c] = invstd;
    if (f.rm) {
      f.rm[c] = (1.f - momentum) * f.rm[c] + momentum * (float)mean;
      f.rv[c] = (1.f - momentum) * f.rv[c] + momentum * (float)(var * unbias);
      if (c == 0 && f.nbt) *f.nbt += 1;
    }
  }
}

#define VQA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
