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
//   acc[k*C + c]: sum k of channel c, scaled by 2^SHIFT;   acc[K*C]: count of partials that were non-finite or outside the
//   fixed-point range -- the consumer then turns the statistics into NaN (loud), as an fp32 sum would have become inf / NaN.
// A partial is one workgroup's fp32 sum; |partial| < 2^(62-SHIFT) keeps v * 2^SHIFT exact and inside int64.
//   BatchNorm statistics (sum y, sum y^2):          SHIFT 24  (|partial| < 2^38; resolution 6e-8)
//   BatchNorm backward (sum g, sum g*xhat):         SHIFT 40  (|partial| < 2^22; resolution 9e-13: gradients are small numbers)
// ------------------------------------------------------------------------------------------------------------------
// Replicas: a sum lives in R copies (producer workgroup w adds to copy w % R, the consumer adds the copies as integers) because
// atomics on ONE address are serialised at ~21 ns each on MI355X (tools/atomic_bench.hip: 768 workgroups x 256 addresses 57 us with
// one copy, 4 us with eight); R = 512 / C clamped to 1..8 keeps R*C constant, so a consumer prologue reads the same few KB whatever C.
//   layout: acc[(r*K + k)*C + c] for replica r, sum k, channel c;  flag at acc[R*K*C];  vqa_bn_acc_words(K, C) words in all.
static inline __host__ __device__ int acc_replicas(int C) { const int r = 512 / (C > 0 ? C : 1); return r < 1 ? 1 : (r > 8 ? 8 : r); }
#define VQA_ACC_FWD_SHIFT 24
#define VQA_ACC_BWD_SHIFT 40
template <int SHIFT>
__device__ __forceinline__ void acc_add_fixed(unsigned long long* acc, float v, unsigned long long* flag) {
  constexpr float lim = (float)(1ull << (62 - SHIFT)), scale = (float)(1ull << SHIFT);
  if (fabsf(v) < lim) atomicAdd(acc, (unsigned long long)__float2ll_rn(v * scale));      // (NaN fails the comparison too)
  else atomicAdd(flag, 1ull);
}
// sum k of channel c over the R <= 8 replicas (integer addition: exact, order-free), as a double.  All loads are issued before
// the first add: a serial loop over a runtime R is a chain of dependent L2 latencies (24 of them cost a consumer 12 us).
template <int SHIFT>
__device__ __forceinline__ double acc_read_fixed(const unsigned long long* acc, int R, int K, int C, int k, int c) {
  long long v[8];
#pragma unroll
  for (int r = 0; r < 8; ++r) v[r] = r < R ? (long long)acc[((size_t)r * K + k) * C + c] : 0ll;
  const long long t = ((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]));
  return (double)t * (1.0 / (double)(1ull << SHIFT));
}

#define VQA_LAUNCH_CHECK() do { hipError_t e_ = hipGetLastError(); if (e_ != hipSuccess) return (int)e_; } while (0)
