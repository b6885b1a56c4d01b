// Token-side kernels for gfx950: embedding+positional encoding, LayerNorm, multi-head attention
// (self: keys masked with -inf, cross: unmasked), masked mean-pool, sigmoid gate, bias/ReLU/dropout
// backward, cross-entropy, and the fused optimizer tail (global L2 norm, clip, AdamW).
//   embedding*sqrt(d)+pe      models/text_encoder.py:504-510, :112-114
//   LayerNorm                 models/text_encoder.py:390,395,519 ; cross_attention.py:286-287,295 ; fusion.py:326
//   attention                 models/text_encoder.py:237-258 ; models/cross_attention.py:176-198
//   masked mean / gate        models/fusion.py:303-320, :160-166
//   CE / clip / AdamW         training/train.py:120, :204-208, :127-132
// Rows are tokens ([B*L][D], D contiguous).  One 64-lane wave owns one row / one (batch, head).
#include <cstdlib>
#include "common.h"

// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void embed_fwd_kernel(const long long* __restrict__ ids, const float* __restrict__ emb, const float* __restrict__ pe,
                                 T* __restrict__ out, int rows, int L, int D, int V, float scale, float p, uint64_t seed) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)rows * D) return;
  const int d = (int)(i % D), row = (int)(i / D), l = row % L;
  long long id = ids[row];
  float v = (id >= 0 && id < V) ? emb[(size_t)id * D + d] * scale : 0.f;
  v += pe[(size_t)l * D + d];
  if (p > 0.f) v = drop_keep32(drop_key(seed), (uint32_t)(i), p) ? v / (1.f - p) : 0.f;
  out[i] = from_f<T>(v);
}
// ---------------------------------------------------------------------------------------------------------------------------
// Fixed-order cross-workgroup reductions.  A kernel whose workgroups each hold a partial vector (bias / LayerNorm gamma, beta /
// position-embedding gradients, loss terms) stores it to its own row of a caller-provided scratch slab; fold_rows_kernel, launched
// right behind it on the same stream by the same C entry, sums the rows IN INDEX ORDER and performs the single += on the
// destination.  The result does not depend on the order the workgroups ran in (float atomics do): bit-reproducible gradients.
// (A same-kernel "last workgroup finishes" variant was measured first: its agent-scope release/acquire fences write back and
// invalidate the XCD's L2 once per workgroup, +50 us per LayerNorm backward; the kernel boundary gives the same visibility for
// one ~2 us launch.)
// block (64, 16): 64 columns x 16 row groups; a group sums a contiguous chunk of rows in index order with 16 loads in flight,
// the 16 group sums are folded in group order.  Columns [0, n0) go to dst0, [n0, ncols) to dst1.
// ---------------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(1024) void fold_rows_kernel(const float* __restrict__ part, int nrows, size_t stride, int ncols,
                                                         float* dst0, int n0, float* dst1) {
  const int c = blockIdx.x * 64 + threadIdx.x, rg = threadIdx.y;
  const int per = (nrows + 15) / 16, r0 = rg * per, r1 = min(nrows, r0 + per);
  float t = 0.f;
  if (c < ncols) {
    int r = r0;
    for (; r + 16 <= r1; r += 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = part[(size_t)(r + i) * stride + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) t += v[i];
    }
    for (; r < r1; ++r) t += part[(size_t)r * stride + c];
  }
  __shared__ float sh[16][64];
  sh[rg][threadIdx.x] = t;
  __syncthreads();
  if (rg == 0 && c < ncols) {
    float a = 0.f;
#pragma unroll
    for (int g = 0; g < 16; ++g) a += sh[g][threadIdx.x];
    if (c < n0) dst0[c] += a; else dst1[c - n0] += a;
  }
}
static inline void launch_fold(const float* part, int nrows, size_t stride, int ncols, float* dst0, int n0, float* dst1, hipStream_t st) {
  hipLaunchKernelGGL(fold_rows_kernel, dim3((ncols + 63) / 64), dim3(64, 16), 0, st, part, nrows, stride, ncols, dst0, n0, dst1);
}
// Many folds in one launch: the parameter gradients they produce are only read by the optimizer at the end of the step, so the engine
// queues them (defer_fold = 1 in the producing entries) and folds a whole gradient segment at once instead of paying one tiny launch per
// LayerNorm / bias on the data-gradient chain.  Same arithmetic per job as fold_rows_kernel (same row groups, same fold order).
constexpr int FOLD_MAXJOBS = 48;
struct FoldJob { const float* part; float* dst0; float* dst1; int nrows, ncols, n0; unsigned stride; };
struct FoldGroup { FoldJob j[FOLD_MAXJOBS]; int blk0[FOLD_MAXJOBS + 1]; int n; };
__global__ __launch_bounds__(1024) void fold_group_kernel(FoldGroup g) {
  int lo = 0, hi = g.n - 1;                                  // last job whose first block <= blockIdx.x (uniform)
  while (lo < hi) { const int mid = (lo + hi + 1) >> 1; if (g.blk0[mid] <= (int)blockIdx.x) lo = mid; else hi = mid - 1; }
  const FoldJob& job = g.j[lo];
  const float* __restrict__ part = job.part;
  const int nrows = job.nrows, ncols = job.ncols;
  const size_t stride = job.stride;
  const int c = ((int)blockIdx.x - g.blk0[lo]) * 64 + threadIdx.x, rg = threadIdx.y;
  const int per = (nrows + 15) / 16, r0 = rg * per, r1 = min(nrows, r0 + per);
  float t = 0.f;
  if (c < ncols) {
    int r = r0;
    for (; r + 16 <= r1; r += 16) {
      float v[16];
#pragma unroll
      for (int i = 0; i < 16; ++i) v[i] = part[(size_t)(r + i) * stride + c];
#pragma unroll
      for (int i = 0; i < 16; ++i) t += v[i];
    }
    for (; r < r1; ++r) t += part[(size_t)r * stride + c];
  }
  __shared__ float sh[16][64];
  sh[rg][threadIdx.x] = t;
  __syncthreads();
  if (rg == 0 && c < ncols) {
    float a = 0.f;
#pragma unroll
    for (int q = 0; q < 16; ++q) a += sh[q][threadIdx.x];
    if (c < job.n0) job.dst0[c] += a; else job.dst1[c - job.n0] += a;
  }
}

// Gradient of the embedding table WITHOUT atomics: a workgroup owns 16 consecutive table rows, walks the id list in row order
// (1024 ids per pass, hits compacted in order through LDS) and adds the matching rows of dout to LDS accumulators in that order
// -> bit-reproducible, one writer per table element.  (The id list is rows*8 bytes, L2-resident, read V/16 times; the reference's
// nn.Embedding backward is a deterministic CPU scatter-add.)
constexpr int EMB_VB = 16;
template <typename T>
__global__ __launch_bounds__(256) void embed_bwd_kernel(const long long* __restrict__ ids, const T* __restrict__ dout, float* demb,
                                                        int rows, int D, int V, float scale, float p, uint64_t seed) {
  extern __shared__ float eacc[];                             // [EMB_VB][D]
  __shared__ int list[1024];                                  // (row << 4) | (id - v0), in row order
  __shared__ int wcnt[4];
  const long long v0 = (long long)blockIdx.x * EMB_VB;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int i = threadIdx.x; i < EMB_VB * D; i += 256) eacc[i] = 0.f;
  const uint32_t dkey = drop_key(seed);
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  bool any = false;
  for (int base = 0; base < rows; base += 1024) {
    // thread t owns rows base + 4t .. base + 4t + 3 (row order == thread order, then k)
    int hit[4];
    int nh = 0;
    long long idv[4];
    const int rb = base + threadIdx.x * 4;
    if (rb + 3 < rows) {                                      // two 16-byte loads (ids is 8-byte aligned, rb is a multiple of 4)
      const u32x4 lo = *reinterpret_cast<const u32x4*>(ids + rb), hi = *reinterpret_cast<const u32x4*>(ids + rb + 2);
      idv[0] = (long long)(((unsigned long long)lo[1] << 32) | lo[0]); idv[1] = (long long)(((unsigned long long)lo[3] << 32) | lo[2]);
      idv[2] = (long long)(((unsigned long long)hi[1] << 32) | hi[0]); idv[3] = (long long)(((unsigned long long)hi[3] << 32) | hi[2]);
    } else {
#pragma unroll
      for (int k = 0; k < 4; ++k) idv[k] = rb + k < rows ? ids[rb + k] : -1;
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      const long long id = idv[k];
      const bool h = id >= v0 && id < v0 + EMB_VB && id > 0 && id < V;       // padding_idx = 0 receives no gradient
      hit[k] = h ? (int)(id - v0) : -1;
      nh += h;
    }
    // exclusive prefix of nh over the 256 threads: wave scan + wave totals through LDS
    int incl = nh;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const int u = __shfl_up(incl, o, 64); if (lane >= o) incl += u; }
    if (lane == 63) wcnt[wave] = incl;
    __syncthreads();
    int off = incl - nh, total = 0;
#pragma unroll
    for (int w = 0; w < 4; ++w) { if (w < wave) off += wcnt[w]; total += wcnt[w]; }
#pragma unroll
    for (int k = 0; k < 4; ++k)
      if (hit[k] >= 0) list[off++] = ((base + threadIdx.x * 4 + k) << 4) | hit[k];
    __syncthreads();
    for (int i0 = 0; i0 < total; i0 += 8) {                   // 8 matching rows in flight, added in list order
      for (int d = threadIdx.x; d < D; d += 256) {
        float gq[8]; int sl[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
          const int e = list[i0 + u < total ? i0 + u : total - 1], row = e >> 4;
          sl[u] = i0 + u < total ? (e & 15) : -1;
          const size_t o = (size_t)row * D + d;
          float g = to_f<T>(dout[o]) * scale;
          if (p > 0.f) g = drop_keep32(dkey, (uint32_t)o, p) ? g * ks : 0.f;
          gq[u] = g;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
          if (sl[u] >= 0) eacc[sl[u] * D + d] += gq[u];      // column d is only ever touched by this thread
      }
    }
    any |= total > 0;
    __syncthreads();
  }
  if (!any) return;
  for (int i = threadIdx.x; i < EMB_VB * D; i += 256) {
    const long long v = v0 + i / D;
    if (v > 0 && v < V && eacc[i] != 0.f) demb[(size_t)v * D + (i % D)] += eacc[i];
  }
}

// ---------------------------------------------------------------------------------------------
// LayerNorm: y = (x-mean)*rstd*gamma+beta ; optional dropout ; optional + addrow[row % period]
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void layernorm_fwd_kernel(const T* __restrict__ x, const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            T* __restrict__ out, float* __restrict__ stats, int rows, int D, float eps,
                                                            float p, uint64_t seed, const float* __restrict__ addrow, int period) {
  const int lane = threadIdx.x & 63;
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  for (int row = wid; row < rows; row += nw) {
    const T* xr = x + (size_t)row * D;
    float s = 0.f;
    for (int c = lane; c < D; c += 64) s += to_f<T>(xr[c]);
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
    for (int c = lane; c < D; c += 64) { const float t = to_f<T>(xr[c]) - mean; q += t * t; }
    const float rstd = rsqrtf(wave_sum(q) / (float)D + eps);
    if (lane == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    for (int c = lane; c < D; c += 64) {
      float y = (to_f<T>(xr[c]) - mean) * rstd * gamma[c] + beta[c];
      if (p > 0.f) y = drop_keep32(drop_key(seed), (uint32_t)((uint64_t)row * D + c), p) ? y / (1.f - p) : 0.f;
      if (addrow) y += addrow[(size_t)(row % period) * D + c];
      out[(size_t)row * D + c] = from_f<T>(y);
    }
  }
}

// bf16, D = 8 * LPR with LPR = 8 / 16 / 32 / 64 lanes per row: a lane owns one 16-byte segment of its row (one load, one store),
// 64 / LPR rows per wave; mean / variance are sub-wave shuffle reductions.  Same arithmetic as layernorm_fwd_kernel.
template <int LPR>
__global__ __launch_bounds__(256) void layernorm_fwd_bf16v_kernel(const bf16_t* __restrict__ x, const float* __restrict__ gamma,
                                                                  const float* __restrict__ beta, bf16_t* __restrict__ out, float* __restrict__ stats,
                                                                  int rows, float eps, float p, uint64_t seed, const float* __restrict__ addrow, int period) {
  constexpr int D = LPR * 8, RPW = 64 / LPR;
  const int lane = threadIdx.x & 63, sub = lane / LPR, sl = lane % LPR;
  const int wid = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = (gridDim.x * blockDim.x) >> 6;
  const int c0 = sl * 8;
  float gm[8], bt[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { gm[j] = gamma[c0 + j]; bt[j] = beta[c0 + j]; }
  const uint32_t dkey = drop_key(seed);
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  for (int r0 = wid * RPW; r0 < rows; r0 += nw * RPW) {
    const int row = r0 + sub;
    const bool live = row < rows;
    Vec16<bf16_t> v = live ? ldg16(x + (size_t)row * D + c0) : zero16<bf16_t>();
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) s += v.get(j);
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    const float mean = s / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; ++j) { const float t = v.get(j) - mean; q += t * t; }
#pragma unroll
    for (int o = LPR / 2; o > 0; o >>= 1) q += __shfl_xor(q, o, 64);
    const float rstd = rsqrtf(q / (float)D + eps);
    if (!live) continue;
    if (sl == 0 && stats) { stats[row * 2] = mean; stats[row * 2 + 1] = rstd; }
    Vec16<bf16_t> o;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      float y = (v.get(j) - mean) * rstd * gm[j] + bt[j];
      if (p > 0.f) y = drop_keep32(dkey, (uint32_t)((uint64_t)row * D + c0 + j), p) ? y * ks : 0.f;
      if (addrow) y += addrow[(size_t)(row % period) * D + c0 + j];
      o.set(j, y);
    }
    stg16(out + (size_t)row * D + c0, o);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void layernorm_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ gamma,
                                                            const float* __restrict__ stats, const T* __restrict__ addend, T* __restrict__ dx,
                                                            float* dgamma, float* dbeta, int rows, int D, float p, uint64_t seed,
                                                            float* dadd, int period, float* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + wave, nw = gridDim.x * 4;
  float ag[8], ab[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) ag[t] = ab[t] = 0.f;
  for (int row = wid; row < rows; row += nw) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    float gv[8], xh[8];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = lane + 64 * t;
      gv[t] = 0.f; xh[t] = 0.f;
      if (c < D) {
        const size_t o = (size_t)row * D + c;
        float g = to_f<T>(dout[o]);
        if (dadd) atomicAdd(dadd + (size_t)(row % period) * D + c, g);
        if (p > 0.f) g = drop_keep32(drop_key(seed), (uint32_t)(o), p) ? g / (1.f - p) : 0.f;
        const float h = (to_f<T>(x[o]) - mean) * rstd;
        gv[t] = g; xh[t] = h;
        ag[t] += g * h; ab[t] += g;
        const float gy = g * gamma[c];
        m1 += gy; m2 += gy * h;
      }
    }
    m1 = wave_sum(m1) / (float)D; m2 = wave_sum(m2) / (float)D;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      const int c = lane + 64 * t;
      if (c < D) {
        const size_t o = (size_t)row * D + c;
        float r = rstd * (gv[t] * gamma[c] - m1 - xh[t] * m2);
        if (addend) r += to_f<T>(addend[o]);
        dx[o] = from_f<T>(r);
      }
    }
  }
  __shared__ float sh[2][4][512];
#pragma unroll
  for (int t = 0; t < 8; ++t) { sh[0][wave][lane + 64 * t] = ag[t]; sh[1][wave][lane + 64 * t] = ab[t]; }
  __syncthreads();
  if (!part) {                                                // no scratch: float atomics (order-dependent rounding)
    for (int c = threadIdx.x; c < D; c += 256) {
      atomicAdd(dgamma + c, sh[0][0][c] + sh[0][1][c] + sh[0][2][c] + sh[0][3][c]);
      atomicAdd(dbeta + c, sh[1][0][c] + sh[1][1][c] + sh[1][2][c] + sh[1][3][c]);
    }
    return;
  }
  for (int c = threadIdx.x; c < D; c += 256) {
    part[((size_t)blockIdx.x * 2) * D + c] = sh[0][0][c] + sh[0][1][c] + sh[0][2][c] + sh[0][3][c];
    part[((size_t)blockIdx.x * 2 + 1) * D + c] = sh[1][0][c] + sh[1][1][c] + sh[1][2][c] + sh[1][3][c];
  }
}

// out[n] += sum_b x[b][n] for x [B][N] (N % Vec16<T>::N == 0): the gradient of a row-periodic addend (position embedding),
// one 16-byte column group per thread, grid.y slices of the batch, one atomic per column per slice.
template <typename T>
__global__ __launch_bounds__(256) void colsum_rows_kernel(const T* __restrict__ x, float* out, int B, size_t N, float* part) {
  constexpr int VEC = Vec16<T>::N;
  const size_t n0 = ((size_t)blockIdx.x * 256 + threadIdx.x) * VEC;
  const bool live = n0 < N;
  float acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) acc[j] = 0.f;
  if (live) {
#pragma unroll 4
    for (int b = blockIdx.y; b < B; b += gridDim.y) {
      const Vec16<T> t = ldg16(x + (size_t)b * N + n0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) acc[j] += t.get(j);
    }
  }
  if (!part) {                                                // no scratch: float atomics (order-dependent rounding)
    if (live) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) atomicAdd(out + n0 + j, acc[j]);
    }
    return;
  }
  if (live) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) part[(size_t)blockIdx.y * N + n0 + j] = acc[j];
  }
}

// bf16, D % 8 == 0, D <= 512: lane owns channels [8*lane, 8*lane+8) -> one 16-byte load per operand per row, gamma and the
// dgamma/dbeta partial sums stay in registers, 8 waves per workgroup are reduced in LDS before the (few) atomics.
__global__ __launch_bounds__(512) void layernorm_bwd_bf16v_kernel(const bf16_t* __restrict__ dout, const bf16_t* __restrict__ x,
                                                                  const float* __restrict__ gamma, const float* __restrict__ stats,
                                                                  const bf16_t* __restrict__ addend, bf16_t* __restrict__ dx, float* dgamma,
                                                                  float* dbeta, int rows, int D, float p, uint64_t seed, float* dadd, int period,
                                                                  float* part) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * 8 + wave, nw = gridDim.x * 8;
  const int c0 = lane * 8;
  const bool act = c0 < D;
  float gm[8], ag[8], ab[8];
#pragma unroll
  for (int t = 0; t < 8; ++t) { gm[t] = act ? gamma[c0 + t] : 0.f; ag[t] = 0.f; ab[t] = 0.f; }
  for (int row = wid; row < rows; row += nw) {
    const float mean = stats[row * 2], rstd = stats[row * 2 + 1];
    const size_t o = (size_t)row * D + c0;
    Vec16<bf16_t> dv = zero16<bf16_t>(), xv = zero16<bf16_t>(), av = zero16<bf16_t>();
    if (act) {
      dv = ldg16(dout + o); xv = ldg16(x + o);
      if (addend) av = ldg16(addend + o);
    }
    float gv[8], xh[8];
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int t = 0; t < 8; ++t) {
      float g = dv.get(t);
      if (dadd && act) atomicAdd(dadd + (size_t)(row % period) * D + c0 + t, g);
      if (p > 0.f) g = drop_keep32(drop_key(seed), (uint32_t)(o + t), p) ? g / (1.f - p) : 0.f;
      const float h = act ? (xv.get(t) - mean) * rstd : 0.f;
      gv[t] = g; xh[t] = h;
      ag[t] += g * h; ab[t] += g;
      const float gy = g * gm[t];
      m1 += gy; m2 += gy * h;
    }
    m1 = wave_sum(m1) / (float)D; m2 = wave_sum(m2) / (float)D;
    if (act) {
      Vec16<bf16_t> ov;
#pragma unroll
      for (int t = 0; t < 8; ++t) ov.set(t, rstd * (gv[t] * gm[t] - m1 - xh[t] * m2) + av.get(t));
      stg16(dx + o, ov);
    }
  }
  __shared__ float sh[2][8][512];
#pragma unroll
  for (int t = 0; t < 8; ++t) { sh[0][wave][c0 + t] = ag[t]; sh[1][wave][c0 + t] = ab[t]; }
  __syncthreads();
  for (int c = threadIdx.x; c < D; c += 512) {
    float a = 0.f, b = 0.f;
#pragma unroll
    for (int w = 0; w < 8; ++w) { a += sh[0][w][c]; b += sh[1][w][c]; }
    if (part) { part[((size_t)blockIdx.x * 2) * D + c] = a; part[((size_t)blockIdx.x * 2 + 1) * D + c] = b; }
    else { atomicAdd(dgamma + c, a); atomicAdd(dbeta + c, b); }
  }
}

// ---------------------------------------------------------------------------------------------
// Attention, one wave per (batch, head).  q/k/v/ctx are token-major with row strides ld* (elements);
// head h occupies columns [h*hd, (h+1)*hd).  probs [B][H][Lq][Lk] fp32 holds softmax BEFORE dropout.
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void attn_fwd_kernel(const T* __restrict__ q, const T* __restrict__ k, const T* __restrict__ v,
                                                      int ldq, int ldk, int ldv, const float* __restrict__ kmask, float* __restrict__ probs,
                                                      T* __restrict__ ctx, int ldc, int H, int Lq, int Lk, int hd, float scale, float p, uint64_t seed) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ldh = hd + 1, ldp = Lk + 1;
  float* Qs = sm; float* Ks = Qs + Lq * ldh; float* Vs = Ks + Lk * ldh; float* Ps = Vs + Lk * ldh;
  for (int i = tid; i < Lq * hd; i += 256) { const int r = i / hd, d = i - r * hd; Qs[r * ldh + d] = to_f<T>(q[(size_t)(b * Lq + r) * ldq + h * hd + d]); }
  for (int i = tid; i < Lk * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    Ks[r * ldh + d] = to_f<T>(k[(size_t)(b * Lk + r) * ldk + h * hd + d]);
    Vs[r * ldh + d] = to_f<T>(v[(size_t)(b * Lk + r) * ldv + h * hd + d]);
  }
  __syncthreads();
  for (int i = tid; i < Lq * Lk; i += 256) {
    const int r = i / Lk, c = i - r * Lk;
    float s = 0.f;
    for (int d = 0; d < hd; ++d) s += Qs[r * ldh + d] * Ks[c * ldh + d];
    s = s / scale;                                                  // divide by sqrt(hd) before masking
    if (kmask && kmask[b * Lk + c] == 0.f) s = -INFINITY;
    Ps[r * ldp + c] = s;
  }
  __syncthreads();
  float* pg = probs + ((size_t)(b * H + h) * Lq) * Lk;
  for (int r = wave; r < Lq; r += 4) {
    float m = -INFINITY;
    for (int c = lane; c < Lk; c += 64) m = fmaxf(m, Ps[r * ldp + c]);
    m = wave_max(m);
    float sum = 0.f;
    for (int c = lane; c < Lk; c += 64) { const float e = expf(Ps[r * ldp + c] - m); Ps[r * ldp + c] = e; sum += e; }   // all -inf row -> NaN like torch
    sum = wave_sum(sum);
    for (int c = lane; c < Lk; c += 64) {
      float pr = Ps[r * ldp + c] / sum;
      pg[(size_t)r * Lk + c] = pr;
      if (p > 0.f) pr = drop_keep32(drop_key(seed), (uint32_t)(((size_t)(b * H + h) * Lq + r) * Lk + c), p) ? pr / (1.f - p) : 0.f;
      Ps[r * ldp + c] = pr;
    }
  }
  __syncthreads();
  for (int i = tid; i < Lq * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    float a = 0.f;
    for (int c = 0; c < Lk; ++c) a += Ps[r * ldp + c] * Vs[c * ldh + d];
    ctx[(size_t)(b * Lq + r) * ldc + h * hd + d] = from_f<T>(a);
  }
}

template <typename T>
__global__ __launch_bounds__(256) void attn_bwd_kernel(const T* __restrict__ dctx, int ldc, const T* __restrict__ q, const T* __restrict__ k,
                                                      const T* __restrict__ v, int ldq, int ldk, int ldv, const float* __restrict__ probs,
                                                      T* __restrict__ dq, T* __restrict__ dk, T* __restrict__ dv, int lddq, int lddk, int lddv,
                                                      int H, int Lq, int Lk, int hd, float scale, float p, uint64_t seed) {
  extern __shared__ float sm[];
  const int b = blockIdx.x / H, h = blockIdx.x - b * H, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, ldh = hd + 1, ldp = Lk + 1;
  float* Qs = sm; float* Os = Qs + Lq * ldh; float* Ks = Os + Lq * ldh; float* Vs = Ks + Lk * ldh;
  float* Ps = Vs + Lk * ldh; float* Ds = Ps + Lq * ldp;             // Ps: dropped probs, Ds: dS
  for (int i = tid; i < Lq * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    Qs[r * ldh + d] = to_f<T>(q[(size_t)(b * Lq + r) * ldq + h * hd + d]);
    Os[r * ldh + d] = to_f<T>(dctx[(size_t)(b * Lq + r) * ldc + h * hd + d]);
  }
  for (int i = tid; i < Lk * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    Ks[r * ldh + d] = to_f<T>(k[(size_t)(b * Lk + r) * ldk + h * hd + d]);
    Vs[r * ldh + d] = to_f<T>(v[(size_t)(b * Lk + r) * ldv + h * hd + d]);
  }
  __syncthreads();
  const float* pg = probs + ((size_t)(b * H + h) * Lq) * Lk;
  for (int r = wave; r < Lq; r += 4) {
    float t = 0.f;
    for (int c = lane; c < Lk; c += 64) {
      const float pr = pg[(size_t)r * Lk + c];
      float ks = 1.f;
      if (p > 0.f) ks = drop_keep32(drop_key(seed), (uint32_t)(((size_t)(b * H + h) * Lq + r) * Lk + c), p) ? 1.f / (1.f - p) : 0.f;
      float dpd = 0.f;
      for (int d = 0; d < hd; ++d) dpd += Os[r * ldh + d] * Vs[c * ldh + d];
      const float dp = dpd * ks;
      Ps[r * ldp + c] = pr * ks;
      Ds[r * ldp + c] = dp;
      t += dp * pr;
    }
    t = wave_sum(t);
    for (int c = lane; c < Lk; c += 64) Ds[r * ldp + c] = pg[(size_t)r * Lk + c] * (Ds[r * ldp + c] - t) / scale;
  }
  __syncthreads();
  for (int i = tid; i < Lq * hd; i += 256) {
    const int r = i / hd, d = i - r * hd;
    float a = 0.f;
    for (int c = 0; c < Lk; ++c) a += Ds[r * ldp + c] * Ks[c * ldh + d];
    dq[(size_t)(b * Lq + r) * lddq + h * hd + d] = from_f<T>(a);
  }
  for (int i = tid; i < Lk * hd; i += 256) {
    const int c = i / hd, d = i - c * hd;
    float a = 0.f, e = 0.f;
    for (int r = 0; r < Lq; ++r) { a += Ds[r * ldp + c] * Qs[r * ldh + d]; e += Ps[r * ldp + c] * Os[r * ldh + d]; }
    dk[(size_t)(b * Lk + c) * lddk + h * hd + d] = from_f<T>(a);
    dv[(size_t)(b * Lk + c) * lddv + h * hd + d] = from_f<T>(e);
  }
}

// ---------------------------------------------------------------------------------------------
// masked mean over L:  out[b][col0+d] = sum_l x[b][l][d]*m[b][l] / max(sum_l m[b][l], 1)   (mask null -> plain mean)
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void masked_pool_fwd_kernel(const T* __restrict__ x, const float* __restrict__ mask, T* __restrict__ out, int ldo, int col0, int L, int D) {
  const int b = blockIdx.x;
  float cnt = 0.f;
  for (int l = 0; l < L; ++l) cnt += mask ? mask[b * L + l] : 1.f;
  cnt = fmaxf(cnt, 1.f);
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += to_f<T>(x[((size_t)b * L + l) * D + d]) * (mask ? mask[b * L + l] : 1.f);
    out[(size_t)b * ldo + col0 + d] = from_f<T>(s / cnt);
  }
}
// dx[b][l][d] = (addend ? addend : 0) + dpool[b][col0+d]*m[b][l]/cnt
template <typename T>
__global__ void masked_pool_bwd_kernel(const T* __restrict__ dpool, int ldo, int col0, const float* __restrict__ mask, const T* __restrict__ addend,
                                       T* __restrict__ dx, int L, int D) {
  const int b = blockIdx.x;
  float cnt = 0.f;
  for (int l = 0; l < L; ++l) cnt += mask ? mask[b * L + l] : 1.f;
  cnt = fmaxf(cnt, 1.f);
  for (int i = threadIdx.x; i < L * D; i += blockDim.x) {
    const int l = i / D, d = i - l * D;
    const size_t o = ((size_t)b * L + l) * D + d;
    float r = to_f<T>(dpool[(size_t)b * ldo + col0 + d]) * (mask ? mask[b * L + l] : 1.f) / cnt;
    if (addend) r += to_f<T>(addend[o]);
    dx[o] = from_f<T>(r);
  }
}

// Both masked means of the fusion tail in one launch (models/fusion.py:281-296: attended queries and text features pooled with the
// same mask into cat = [att | txt]): blockIdx.y picks the tensor; the arithmetic per element is masked_pool_fwd_kernel's.
template <typename T>
__global__ void masked_pool_pair_fwd_kernel(const T* __restrict__ x0, const T* __restrict__ x1, const float* __restrict__ mask, T* __restrict__ out,
                                            int L, int D) {
  const int b = blockIdx.x, which = blockIdx.y;
  const T* __restrict__ x = which ? x1 : x0;
  float cnt = 0.f;
  for (int l = 0; l < L; ++l) cnt += mask ? mask[b * L + l] : 1.f;
  cnt = fmaxf(cnt, 1.f);
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    float s = 0.f;
    for (int l = 0; l < L; ++l) s += to_f<T>(x[((size_t)b * L + l) * D + d]) * (mask ? mask[b * L + l] : 1.f);
    out[(size_t)b * 2 * D + which * D + d] = from_f<T>(s / cnt);
  }
}
// ... and both backward broadcasts: dx{0,1}[b][l][d] = dcat[b][{0,D}+d] * m[b][l] / cnt
template <typename T>
__global__ void masked_pool_pair_bwd_kernel(const T* __restrict__ dcat, const float* __restrict__ mask, T* __restrict__ dx0, T* __restrict__ dx1,
                                            int L, int D) {
  const int b = blockIdx.x, which = blockIdx.y;
  T* __restrict__ dx = which ? dx1 : dx0;
  float cnt = 0.f;
  for (int l = 0; l < L; ++l) cnt += mask ? mask[b * L + l] : 1.f;
  cnt = fmaxf(cnt, 1.f);
  for (int d = threadIdx.x; d < D; d += blockDim.x) {
    const float g = to_f<T>(dcat[(size_t)b * 2 * D + which * D + d]);
    for (int l = 0; l < L; ++l)
      dx[((size_t)b * L + l) * D + d] = from_f<T>(g * (mask ? mask[b * L + l] : 1.f) / cnt);
  }
}

// gate: g = sigmoid(z); fused = g*att + (1-g)*txt, cat = [att | txt]  (models/fusion.py:160-166)
template <typename T>
__global__ void gate_fwd_kernel(const T* __restrict__ z, const T* __restrict__ cat, T* __restrict__ fused, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  const float g = 1.f / (1.f + expf(-to_f<T>(z[i])));
  const float a = to_f<T>(cat[(size_t)b * 2 * D + d]), t = to_f<T>(cat[(size_t)b * 2 * D + D + d]);
  fused[i] = from_f<T>(g * a + (1.f - g) * t);
}
template <typename T>
__global__ void gate_bwd_kernel(const T* __restrict__ dfused, const T* __restrict__ z, const T* __restrict__ cat, T* __restrict__ dz,
                                T* __restrict__ dcat, int B, int D) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * D) return;
  const int b = i / D, d = i - b * D;
  const float g = 1.f / (1.f + expf(-to_f<T>(z[i])));
  const float a = to_f<T>(cat[(size_t)b * 2 * D + d]), t = to_f<T>(cat[(size_t)b * 2 * D + D + d]);
  const float df = to_f<T>(dfused[i]);
  dz[i] = from_f<T>(df * (a - t) * g * (1.f - g));
  dcat[(size_t)b * 2 * D + d] = from_f<T>(df * g);
  dcat[(size_t)b * 2 * D + D + d] = from_f<T>(df * (1.f - g));
}

// out = a + b (either may alias out)
template <typename T>
__global__ void add_kernel(const T* __restrict__ a, const T* __restrict__ b, T* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = from_f<T>(to_f<T>(a[i]) + to_f<T>(b[i]));
}

// dz = dout * [out>0] * dropout-keep-scale ; dbias[n] += column sums of dz
template <typename T>
__global__ __launch_bounds__(256) void bias_act_bwd_kernel(const T* __restrict__ dout, const T* __restrict__ outact, T* __restrict__ dz,
                                                           float* dbias, int M, int N, int relu_drop, float p, uint64_t seed,
                                                           float* part) {
  // block handles a strip of 64 columns x rows_per_block rows; thread (r = tid/64, c = tid%64)
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), rl = threadIdx.x >> 6;
  const int rows_per = (M + gridDim.y - 1) / gridDim.y;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s = 0.f;
  if (c < N)
    for (int r = r0 + rl; r < r1; r += 4) {
      const size_t o = (size_t)r * N + c;
      float g = to_f<T>(dout[o]);
      if (outact) { if (!(to_f<T>(outact[o]) > 0.f)) g = 0.f; else if (p > 0.f) g /= (1.f - p); }   // relu(+dropout): out>0 encodes both
      else if (p > 0.f) g = drop_keep32(drop_key(seed), (uint32_t)(o), p) ? g / (1.f - p) : 0.f;
      if (dz) dz[o] = from_f<T>(g);
      s += g;
    }
  __shared__ float sh[4][64];
  sh[rl][threadIdx.x & 63] = s;
  __syncthreads();
  (void)relu_drop;
  if (!dbias) return;
  const float tcol = sh[0][threadIdx.x & 63] + sh[1][threadIdx.x & 63] + sh[2][threadIdx.x & 63] + sh[3][threadIdx.x & 63];
  if (!part) { if (rl == 0 && c < N) atomicAdd(dbias + c, tcol); return; }
  if (rl == 0 && c < N) part[(size_t)blockIdx.y * N + c] = tcol;
}

// Vectorised form of bias_act_bwd_kernel for N % 8 == 0 (bf16) / N % 4 == 0 (fp32): a thread owns one 16-byte column group and
// walks the rows of its strip (the scalar kernel above issues 2-byte loads: 18 us per call on tensors a few MB large).
// dbias: per-thread partial sums, folded over the block's row lanes in LDS, one atomic per column per block.
template <typename T>
__global__ __launch_bounds__(256) void bias_act_bwd_vec_kernel(const T* __restrict__ dout, const T* __restrict__ outact, T* __restrict__ dz,
                                                               float* dbias, int M, int N, float p, uint64_t seed, int rows_per,
                                                               float* part) {
  constexpr int VEC = Vec16<T>::N;
  const int cvs = N / VEC;                                   // column groups per row
  const int gpb = cvs < 256 ? cvs : 256;                     // column groups handled by one block (per blockIdx.x)
  const int lanes_r = 256 / gpb;                             // rows in flight per block pass
  const int cg = blockIdx.x * gpb + threadIdx.x % gpb, rl = threadIdx.x / gpb;
  const int r0 = blockIdx.y * rows_per, r1 = min(M, r0 + rows_per);
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  const float ks = p > 0.f ? 1.f / (1.f - p) : 1.f;
  const uint32_t dkey = drop_key(seed);
  if (cg < cvs && rl < lanes_r) {
#pragma unroll 2
    for (int r = r0 + rl; r < r1; r += lanes_r) {
      const size_t o = (size_t)r * N + (size_t)cg * VEC;
      Vec16<T> d = ldg16(dout + o), a, z;
      if (outact) a = ldg16(outact + o);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float g = d.get(j);
        if (outact) { if (!(a.get(j) > 0.f)) g = 0.f; else g *= ks; }          // relu(+dropout): out > 0 encodes both masks
        else if (p > 0.f) g = drop_keep32(dkey, (uint32_t)(o + j), p) ? g * ks : 0.f;
        z.set(j, g);
        s[j] += g;
      }
      if (dz) stg16(dz + o, z);
    }
  }
  if (!dbias) return;
  __shared__ float sh[256 * VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[threadIdx.x * VEC + j] = s[j];
  __syncthreads();
  for (int c = threadIdx.x; c < gpb * VEC; c += 256) {
    const int gq = c / VEC, j = c - gq * VEC;
    float t = 0.f;
    for (int r = 0; r < lanes_r; ++r) t += sh[(r * gpb + gq) * VEC + j];
    const int col = (blockIdx.x * gpb + gq) * VEC + j;
    if (col < N) { if (part) part[(size_t)blockIdx.y * N + col] = t; else atomicAdd(dbias + col, t); }
  }
}

// cross entropy (mean) forward+backward in one pass: wave per row
template <typename T>
__global__ __launch_bounds__(256) void cross_entropy_kernel(const T* __restrict__ logits, const long long* __restrict__ targets, float* loss,
                                                            T* __restrict__ dlogits, float* __restrict__ logits_f32, int B, int N, float gscale,
                                                            int* err, float* part) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row < B) {
    const T* lr = logits + (size_t)row * N;
    float m = -INFINITY;
    for (int c = lane; c < N; c += 64) m = fmaxf(m, to_f<T>(lr[c]));
    m = wave_max(m);
    float s = 0.f;
    for (int c = lane; c < N; c += 64) s += expf(to_f<T>(lr[c]) - m);
    s = wave_sum(s);
    const long long t64 = targets[row];
    // a target outside [0, N) raises in the reference (nn.CrossEntropyLoss, training/train.py:120): never read out of bounds,
    // count the row in *err (the host mirror raises from it) and poison its loss term and gradient row with NaN
    const bool bad = t64 < 0 || t64 >= (long long)N;
    const int t = bad ? 0 : (int)t64;
    const float lse = m + logf(s);
    if (lane == 0) {
      const float term = bad ? __builtin_nanf("") : (lse - to_f<T>(lr[t])) / (float)B;
      if (part) part[row] = term;
      else if (loss) atomicAdd(loss, term);
      if (bad && err) atomicAdd(err, 1);
    }
    for (int c = lane; c < N; c += 64) {
      const float x = to_f<T>(lr[c]);
      if (logits_f32) logits_f32[(size_t)row * N + c] = x;
      if (dlogits) dlogits[(size_t)row * N + c] = from_f<T>(bad ? __builtin_nanf("") : (expf(x - lse) - (c == t ? 1.f : 0.f)) * gscale / (float)B);
    }
  }
}

template <typename TI, typename TO>
__global__ void convert_kernel(const TI* __restrict__ in, TO* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = from_f<TO>(to_f<TI>(in[i]));
}

// ---------------------------------------------------------------------------------------------
// optimizer tail over flat fp32 buffers
// ---------------------------------------------------------------------------------------------
// Deterministic two-stage sum of squares (replicas must compute bit-identical clip factors): per-block partials in a fixed
// slot each, then one block folds them in a fixed order.  out[0] = result, out[1 .. 1+blocks) = partials.
__global__ __launch_bounds__(256) void sumsq_kernel(const float* __restrict__ g, size_t n, float* out) {
  float s = 0.f;
  const size_t nv = n / 4;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
    const float4 v = reinterpret_cast<const float4*>(g)[i];
    s += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w;
  }
  if (blockIdx.x == 0) for (size_t i = nv * 4 + threadIdx.x; i < n; i += blockDim.x) s += g[i] * g[i];
  __shared__ float sh[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[1 + blockIdx.x] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
__global__ __launch_bounds__(256) void sumsq_final_kernel(float* out, int nparts) {
  float s = 0.f;
  for (int i = threadIdx.x; i < nparts; i += 256) s += out[1 + i];
  __shared__ float sh[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) out[0] = (sh[0] + sh[1]) + (sh[2] + sh[3]);
}
// clip_grad_norm_(max_norm) + AdamW (decoupled weight decay), torch semantics (training/train.py:204-208,127-132)
__global__ void adamw_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v, size_t n,
                             float lr, float b1, float b2, float eps, float wd, long long calls, const float* __restrict__ sumsq,
                             float max_norm, float gscale, const int* __restrict__ skip, int* __restrict__ skipped, bf16_t* __restrict__ p_bf16) {
  // p_bf16 != nullptr: the bf16 working copy of the parameters (what the next forward's GEMMs read) is written here as well -- the
  // separate cast launch over the flat buffer (116 MB of traffic, 34 us at the head of every step) is gone.  A skipped launch leaves
  // both the parameters and the copy as they were.
  // skip[0] != 0 (rows with an out-of-range target in THIS step, summed over ranks): the reference raises before
  // optimizer.step() (nn.CrossEntropyLoss, training/train.py:120,182-208) and leaves the model intact -- so does this kernel:
  // parameters and both moments stay untouched; skipped[0] += rows, skipped[1] += 1 (read by the host at its logging interval),
  // skipped[2] += 1 (never reset: the device-side record of how many launches did NOT count as an optimizer step).
  if (skip && *skip != 0) {
    if (skipped && blockIdx.x == 0 && threadIdx.x == 0) { skipped[0] += *skip; skipped[1] += 1; skipped[2] += 1; }
    return;
  }
  // Adam's step number lives on the DEVICE: calls (the host's launch count, this one included) minus the launches skipped so far.
  // skipped[2] is only written by a launch that skips, and such a launch reads it nowhere else: no race.  (Round 3 derived the bias
  // corrections from a host counter that was repaired only at the caller's next check(): every step in between ran one step ahead.)
  const long long t = calls - (skipped ? (long long)skipped[2] : 0ll);
  const float bc1 = (float)(1.0 - pow((double)b1, (double)t)), bc2 = (float)(1.0 - pow((double)b2, (double)t));
  float coef = gscale;
  if (sumsq && max_norm > 0.f) {
    const float norm = sqrtf(*sumsq) * gscale;
    const float c = max_norm / (norm + 1e-6f);
    if (c < 1.f) coef *= c;
  }
  const float step = lr / bc1, rbc2 = 1.f / sqrtf(bc2);
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float gr = g[i] * coef;
    float pi = p[i] * (1.f - lr * wd);
    const float mi = b1 * m[i] + (1.f - b1) * gr;
    const float vi = b2 * v[i] + (1.f - b2) * gr * gr;
    m[i] = mi; v[i] = vi;
    pi -= step * mi / (sqrtf(vi) * rbc2 + eps);
    p[i] = pi;
    if (p_bf16) p_bf16[i] = f2bf(pi);
  }
}

// ---------------------------------------------------------------------------------------------
#define DT(call_f, call_b) do { if (dtype) { call_b; } else { call_f; } } while (0)
static inline unsigned g1(size_t n) { return (unsigned)((n + 255) / 256); }

extern "C" {

int vqa_embed_fwd(int dtype, const long long* ids, const float* emb, const float* pe, void* out, int rows, int L, int D, int V,
                  float scale, float p, unsigned long long seed, hipStream_t st) {
  const size_t n = (size_t)rows * D;
  DT(hipLaunchKernelGGL(embed_fwd_kernel<float>, dim3(g1(n)), dim3(256), 0, st, ids, emb, pe, (float*)out, rows, L, D, V, scale, p, seed),
     hipLaunchKernelGGL(embed_fwd_kernel<bf16_t>, dim3(g1(n)), dim3(256), 0, st, ids, emb, pe, (bf16_t*)out, rows, L, D, V, scale, p, seed));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_embed_bwd(int dtype, const long long* ids, const void* dout, float* demb, int rows, int D, int V, float scale, float p,
                  unsigned long long seed, hipStream_t st) {
  if (D > 2048 || rows <= 0 || rows >= (1 << 27) || V <= 0) return VQA_EARG;
  const size_t shm = (size_t)EMB_VB * D * 4;
  const dim3 grid((V + EMB_VB - 1) / EMB_VB);
  static size_t attr_f = 0, attr_b = 0;
  if (!dtype && shm > 65536 - 4200 && shm > attr_f) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_f = shm; }
  if (dtype && shm > 65536 - 4200 && shm > attr_b) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&embed_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr_b = shm; }
  DT(hipLaunchKernelGGL(embed_bwd_kernel<float>, grid, dim3(256), shm, st, ids, (const float*)dout, demb, rows, D, V, scale, p, seed),
     hipLaunchKernelGGL(embed_bwd_kernel<bf16_t>, grid, dim3(256), shm, st, ids, (const bf16_t*)dout, demb, rows, D, V, scale, p, seed));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_layernorm_fwd(int dtype, const void* x, const float* gamma, const float* beta, void* out, float* stats, int rows, int D, float eps,
                      float p, unsigned long long seed, const float* addrow, int period, hipStream_t st) {
  if (D > 512 || rows <= 0) return VQA_EARG;
  if (dtype && (D == 64 || D == 128 || D == 256 || D == 512)) {
    const int rpw = 512 / D, waves = (rows + rpw - 1) / rpw;
    const int g = (waves + 3) / 4 > 2048 ? 2048 : (waves + 3) / 4;
#define LNV(L) hipLaunchKernelGGL(layernorm_fwd_bf16v_kernel<L>, dim3(g), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)out, stats, rows, eps, p, seed, addrow, period)
    if (D == 64) LNV(8); else if (D == 128) LNV(16); else if (D == 256) LNV(32); else LNV(64);
#undef LNV
    VQA_LAUNCH_CHECK(); return VQA_OK;
  }
  const int grid = (rows + 3) / 4 > 2048 ? 2048 : (rows + 3) / 4;
  DT(hipLaunchKernelGGL(layernorm_fwd_kernel<float>, dim3(grid), dim3(256), 0, st, (const float*)x, gamma, beta, (float*)out, stats, rows, D, eps, p, seed, addrow, period),
     hipLaunchKernelGGL(layernorm_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, st, (const bf16_t*)x, gamma, beta, (bf16_t*)out, stats, rows, D, eps, p, seed, addrow, period));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// launch geometry of vqa_layernorm_bwd (shared with the scratch-size query)
struct LnBwdGeom { int nb; bool colsum; unsigned gx, gy; size_t N; };
static LnBwdGeom ln_bwd_geom(int dtype, int rows, int D, int period) {
  LnBwdGeom g = {0, false, 0, 0, 0};
  if (dtype && D % 8 == 0) {                                 // 8 waves per block, ~2 rows per wave (4+ rows left 10 240-row tensors latency-bound: 13 us for 15 MB)
    const int rpb = vqa_env_int("VQA_LN_BWD_ROWS", 16);
    g.nb = (rows + rpb - 1) / rpb; if (g.nb > 1024) g.nb = 1024;
  }
  else g.nb = (rows + 15) / 16 > 2048 ? 2048 : (rows + 15) / 16;
  if (period > 0 && rows % period == 0 && ((size_t)period * D) % (dtype ? 8 : 4) == 0) {
    const int Bb = rows / period;
    g.colsum = true; g.N = (size_t)period * D;
    g.gx = (unsigned)((g.N / (dtype ? 8 : 4) + 255) / 256); g.gy = Bb < 32 ? Bb : 32;
  }
  return g;
}
// Floats of the `ws` scratch of vqa_layernorm_bwd (per-workgroup partial rows, see fold_rows_kernel).  period = the addrow period
// when dadd is requested, else 0.
long long vqa_layernorm_bwd_ws(int dtype, int rows, int D, int period) {
  if (D > 512 || rows <= 0) return 0;
  const LnBwdGeom g = ln_bwd_geom(dtype, rows, D, period);
  return (long long)g.nb * 2 * D + (g.colsum ? (long long)g.gy * (long long)g.N : 0);
}
// ws: scratch of vqa_layernorm_bwd_ws floats -> dgamma / dbeta / dadd are summed in a fixed order (bit-reproducible);
// nullptr -> float atomics (same values up to rounding order).
// Fold descriptors of vqa_layernorm_bwd for a caller that defers the folds (defer_fold = 1): returns the number of folds (1, or 2 with a
// position-embedding sum) and writes 5 values per fold to out: {offset into ws (floats), rows, row stride, columns, n0}; fold 0 targets
// (dgamma | dbeta) split at column n0 = D, fold 1 targets dadd.
int vqa_layernorm_bwd_folds(int dtype, int rows, int D, int period, long long* out) {
  if (D > 512 || rows <= 0 || !out) return 0;
  const LnBwdGeom g = ln_bwd_geom(dtype, rows, D, period);
  out[0] = 0; out[1] = g.nb; out[2] = 2 * D; out[3] = 2 * D; out[4] = D;
  if (!g.colsum) return 1;
  out[5] = (long long)g.nb * 2 * D; out[6] = g.gy; out[7] = (long long)g.N; out[8] = (long long)g.N; out[9] = (long long)g.N;
  return 2;
}
int vqa_layernorm_bwd(int dtype, const void* dout, const void* x, const float* gamma, const float* stats, const void* addend, void* dx,
                      float* dgamma, float* dbeta, int rows, int D, float p, unsigned long long seed, float* dadd, int period,
                      float* ws, int defer_fold, hipStream_t st) {
  if (D > 512 || rows <= 0 || (defer_fold && !ws)) return VQA_EARG;
  const LnBwdGeom g = ln_bwd_geom(dtype, rows, D, dadd ? period : 0);
  float* part_ln = ws;
  float* part_cs = ws ? ws + (size_t)g.nb * 2 * D : nullptr;
  if (dadd && g.colsum) {
    // position-embedding gradient as its own column-sum pass (one read of dout) instead of rows*D atomics inside the LN kernel
    const int Bb = rows / period;
    dim3 grid(g.gx, g.gy);
    DT(hipLaunchKernelGGL(colsum_rows_kernel<float>, grid, dim3(256), 0, st, (const float*)dout, dadd, Bb, g.N, part_cs),
       hipLaunchKernelGGL(colsum_rows_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)dout, dadd, Bb, g.N, part_cs));
    if (part_cs && !defer_fold) launch_fold(part_cs, (int)g.gy, g.N, (int)g.N, dadd, (int)g.N, nullptr, st);
    dadd = nullptr;
  }
  if (dtype && D % 8 == 0) {                               // vectorised bf16 path, >= 4 rows per wave, at most one workgroup per CU
    hipLaunchKernelGGL(layernorm_bwd_bf16v_kernel, dim3(g.nb), dim3(512), 0, st, (const bf16_t*)dout, (const bf16_t*)x, gamma, stats,
                       (const bf16_t*)addend, (bf16_t*)dx, dgamma, dbeta, rows, D, p, seed, dadd, period, part_ln);
  } else {
    DT(hipLaunchKernelGGL(layernorm_bwd_kernel<float>, dim3(g.nb), dim3(256), 0, st, (const float*)dout, (const float*)x, gamma, stats, (const float*)addend, (float*)dx, dgamma, dbeta, rows, D, p, seed, dadd, period, part_ln),
       hipLaunchKernelGGL(layernorm_bwd_kernel<bf16_t>, dim3(g.nb), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)x, gamma, stats, (const bf16_t*)addend, (bf16_t*)dx, dgamma, dbeta, rows, D, p, seed, dadd, period, part_ln));
  }
  if (part_ln && !defer_fold) launch_fold(part_ln, g.nb, (size_t)2 * D, 2 * D, dgamma, D, dbeta, st);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_attention_fwd(int dtype, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const float* kmask, float* probs,
                      void* ctx, int ldc, int B, int H, int Lq, int Lk, int hd, float p, unsigned long long seed, hipStream_t st) {
  const size_t shm = ((size_t)(Lq + 2 * Lk) * (hd + 1) + (size_t)Lq * (Lk + 1)) * 4;
  if (shm > 160 * 1024) return VQA_EARG;
  const float scale = sqrtf((float)hd);
  if (dtype) { if (shm > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); }
  else { if (shm > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_fwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); }
  DT(hipLaunchKernelGGL(attn_fwd_kernel<float>, dim3(B * H), dim3(256), shm, st, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, kmask, probs, (float*)ctx, ldc, H, Lq, Lk, hd, scale, p, seed),
     hipLaunchKernelGGL(attn_fwd_kernel<bf16_t>, dim3(B * H), dim3(256), shm, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, kmask, probs, (bf16_t*)ctx, ldc, H, Lq, Lk, hd, scale, p, seed));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_attention_bwd(int dtype, const void* dctx, int ldc, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                      const float* probs, void* dq, void* dk, void* dv, int lddq, int lddk, int lddv,
                      int B, int H, int Lq, int Lk, int hd, float p, unsigned long long seed, hipStream_t st) {
  const size_t shm = ((size_t)(2 * Lq + 2 * Lk) * (hd + 1) + (size_t)2 * Lq * (Lk + 1)) * 4;
  if (shm > 160 * 1024) return VQA_EARG;
  const float scale = sqrtf((float)hd);
  if (dtype) { if (shm > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<bf16_t>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); }
  else { if (shm > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&attn_bwd_kernel<float>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); }
  DT(hipLaunchKernelGGL(attn_bwd_kernel<float>, dim3(B * H), dim3(256), shm, st, (const float*)dctx, ldc, (const float*)q, (const float*)k, (const float*)v, ldq, ldk, ldv, probs, (float*)dq, (float*)dk, (float*)dv, lddq, lddk, lddv, H, Lq, Lk, hd, scale, p, seed),
     hipLaunchKernelGGL(attn_bwd_kernel<bf16_t>, dim3(B * H), dim3(256), shm, st, (const bf16_t*)dctx, ldc, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv, probs, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, lddq, lddk, lddv, H, Lq, Lk, hd, scale, p, seed));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_masked_pool_fwd(int dtype, const void* x, const float* mask, void* out, int ldo, int col0, int B, int L, int D, hipStream_t st) {
  DT(hipLaunchKernelGGL(masked_pool_fwd_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)x, mask, (float*)out, ldo, col0, L, D),
     hipLaunchKernelGGL(masked_pool_fwd_kernel<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)x, mask, (bf16_t*)out, ldo, col0, L, D));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_masked_pool_pair_fwd(int dtype, const void* x0, const void* x1, const float* mask, void* out, int B, int L, int D, hipStream_t st) {
  if (!x0 || !x1 || !out || B <= 0 || L <= 0 || D <= 0) return VQA_EARG;
  DT(hipLaunchKernelGGL(masked_pool_pair_fwd_kernel<float>, dim3(B, 2), dim3(256), 0, st, (const float*)x0, (const float*)x1, mask, (float*)out, L, D),
     hipLaunchKernelGGL(masked_pool_pair_fwd_kernel<bf16_t>, dim3(B, 2), dim3(256), 0, st, (const bf16_t*)x0, (const bf16_t*)x1, mask, (bf16_t*)out, L, D));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_masked_pool_pair_bwd(int dtype, const void* dcat, const float* mask, void* dx0, void* dx1, int B, int L, int D, hipStream_t st) {
  if (!dcat || !dx0 || !dx1 || B <= 0 || L <= 0 || D <= 0) return VQA_EARG;
  DT(hipLaunchKernelGGL(masked_pool_pair_bwd_kernel<float>, dim3(B, 2), dim3(256), 0, st, (const float*)dcat, mask, (float*)dx0, (float*)dx1, L, D),
     hipLaunchKernelGGL(masked_pool_pair_bwd_kernel<bf16_t>, dim3(B, 2), dim3(256), 0, st, (const bf16_t*)dcat, mask, (bf16_t*)dx0, (bf16_t*)dx1, L, D));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_masked_pool_bwd(int dtype, const void* dpool, int ldo, int col0, const float* mask, const void* addend, void* dx, int B, int L, int D, hipStream_t st) {
  DT(hipLaunchKernelGGL(masked_pool_bwd_kernel<float>, dim3(B), dim3(256), 0, st, (const float*)dpool, ldo, col0, mask, (const float*)addend, (float*)dx, L, D),
     hipLaunchKernelGGL(masked_pool_bwd_kernel<bf16_t>, dim3(B), dim3(256), 0, st, (const bf16_t*)dpool, ldo, col0, mask, (const bf16_t*)addend, (bf16_t*)dx, L, D));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_gate_fwd(int dtype, const void* z, const void* cat, void* fused, int B, int D, hipStream_t st) {
  DT(hipLaunchKernelGGL(gate_fwd_kernel<float>, dim3(g1((size_t)B * D)), dim3(256), 0, st, (const float*)z, (const float*)cat, (float*)fused, B, D),
     hipLaunchKernelGGL(gate_fwd_kernel<bf16_t>, dim3(g1((size_t)B * D)), dim3(256), 0, st, (const bf16_t*)z, (const bf16_t*)cat, (bf16_t*)fused, B, D));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_gate_bwd(int dtype, const void* dfused, const void* z, const void* cat, void* dz, void* dcat, int B, int D, hipStream_t st) {
  DT(hipLaunchKernelGGL(gate_bwd_kernel<float>, dim3(g1((size_t)B * D)), dim3(256), 0, st, (const float*)dfused, (const float*)z, (const float*)cat, (float*)dz, (float*)dcat, B, D),
     hipLaunchKernelGGL(gate_bwd_kernel<bf16_t>, dim3(g1((size_t)B * D)), dim3(256), 0, st, (const bf16_t*)dfused, (const bf16_t*)z, (const bf16_t*)cat, (bf16_t*)dz, (bf16_t*)dcat, B, D));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_add(int dtype, const void* a, const void* b, void* out, long long n, hipStream_t st) {
  DT(hipLaunchKernelGGL(add_kernel<float>, dim3(g1((size_t)n)), dim3(256), 0, st, (const float*)a, (const float*)b, (float*)out, (size_t)n),
     hipLaunchKernelGGL(add_kernel<bf16_t>, dim3(g1((size_t)n)), dim3(256), 0, st, (const bf16_t*)a, (const bf16_t*)b, (bf16_t*)out, (size_t)n));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// outact != null: relu (and dropout folded into out>0); outact == null && p>0: dropout mask regenerated from (seed, index)
struct BiasBwdGeom { bool vec; int gpb, rows_per; dim3 grid; };
static BiasBwdGeom bias_bwd_geom(int dtype, int M, int N) {
  BiasBwdGeom g;
  const int VEC = dtype ? 8 : 4;
  g.vec = N % VEC == 0 && M > 0 && (256 % (N / VEC < 256 ? N / VEC : 256)) == 0;
  if (g.vec) {
    const int cvs = N / VEC, gpb = cvs < 256 ? cvs : 256, lanes_r = 256 / gpb;
    int gy = (M + 4 * lanes_r - 1) / (4 * lanes_r);          // ~4 row passes per block (16 left a 10 240-row tensor with 80 latency-bound
                                                             // workgroups: 14 us for 5 MB); <= 1024 blocks in y
    if (gy > 1024) gy = 1024;
    if (gy < 1) gy = 1;
    g.gpb = gpb; g.rows_per = (M + gy - 1) / gy;
    g.grid = dim3((cvs + gpb - 1) / gpb, (M + g.rows_per - 1) / g.rows_per);
  } else {
    int gy = (M + 127) / 128; if (gy > 256) gy = 256; if (gy < 1) gy = 1;
    g.gpb = 0; g.rows_per = 0; g.grid = dim3((N + 63) / 64, gy);
  }
  return g;
}
// Floats of the `ws` scratch of vqa_bias_act_bwd (one partial row of N column sums per workgroup row).
long long vqa_bias_act_bwd_ws(int dtype, int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  const BiasBwdGeom g = bias_bwd_geom(dtype, M, N);
  return (long long)g.grid.y * N;
}
// ws: vqa_bias_act_bwd_ws floats -> dbias summed in a fixed order (bit-reproducible); nullptr -> float atomics on dbias.
// rows of the partial slab (= rows of the deferred fold {offset 0, rows, stride N, N columns, n0 = N} into dbias)
int vqa_bias_act_bwd_fold_rows(int dtype, int M, int N) {
  if (M <= 0 || N <= 0) return 0;
  return (int)bias_bwd_geom(dtype, M, N).grid.y;
}
int vqa_bias_act_bwd(int dtype, const void* dout, const void* outact, void* dz, float* dbias, int M, int N, float p, unsigned long long seed,
                     float* ws, int defer_fold, hipStream_t st) {
  if (M <= 0 || N <= 0 || (defer_fold && !ws)) return VQA_EARG;
  const BiasBwdGeom g = bias_bwd_geom(dtype, M, N);
  if (g.vec) {
    DT(hipLaunchKernelGGL(bias_act_bwd_vec_kernel<float>, g.grid, dim3(256), 0, st, (const float*)dout, (const float*)outact, (float*)dz, dbias, M, N, p, seed, g.rows_per, ws),
       hipLaunchKernelGGL(bias_act_bwd_vec_kernel<bf16_t>, g.grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)outact, (bf16_t*)dz, dbias, M, N, p, seed, g.rows_per, ws));
  } else {
    DT(hipLaunchKernelGGL(bias_act_bwd_kernel<float>, g.grid, dim3(256), 0, st, (const float*)dout, (const float*)outact, (float*)dz, dbias, M, N, 0, p, seed, ws),
       hipLaunchKernelGGL(bias_act_bwd_kernel<bf16_t>, g.grid, dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)outact, (bf16_t*)dz, dbias, M, N, 0, p, seed, ws));
  }
  if (ws && dbias && !defer_fold) launch_fold(ws, (int)g.grid.y, (size_t)N, N, dbias, N, nullptr, st);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// dst0_j[c] (c < n0_j) / dst1_j[c - n0_j] += sum over rows r < nrows_j of part_j[r * stride_j + c], rows in index order -- the deferred folds of
// any number of vqa_layernorm_bwd / vqa_bias_act_bwd calls (defer_fold = 1) in ceil(njobs / 48) launches.
int vqa_fold_group(int njobs, const float* const* part, const int* nrows, const long long* stride, const int* ncols, float* const* dst0,
                   const int* n0, float* const* dst1, hipStream_t st) {
  if (njobs <= 0 || !part || !nrows || !stride || !ncols || !dst0 || !n0 || !dst1) return VQA_EARG;
  for (int base = 0; base < njobs; base += FOLD_MAXJOBS) {
    FoldGroup g;
    g.n = njobs - base < FOLD_MAXJOBS ? njobs - base : FOLD_MAXJOBS;
    g.blk0[0] = 0;
    for (int q = 0; q < g.n; ++q) {
      const int j = base + q;
      if (!part[j] || !dst0[j] || nrows[j] <= 0 || ncols[j] <= 0 || stride[j] <= 0 || stride[j] > 0xffffffffll || (n0[j] < ncols[j] && !dst1[j]))
        return VQA_EARG;
      g.j[q] = FoldJob{part[j], dst0[j], dst1[j], nrows[j], ncols[j], n0[j], (unsigned)stride[j]};
      g.blk0[q + 1] = g.blk0[q] + (ncols[j] + 63) / 64;
    }
    hipLaunchKernelGGL(fold_group_kernel, dim3(g.blk0[g.n]), dim3(64, 16), 0, st, g);
  }
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// ws (B floats or nullptr): the per-row loss terms are summed in row order by a second launch (bit-reproducible); nullptr -> one
// float atomic per row.  *loss is accumulated into (+=): zero it first.
int vqa_cross_entropy(int dtype, const void* logits, const long long* targets, float* loss, void* dlogits, float* logits_f32, int B, int N,
                      float gscale, int* err, float* ws, hipStream_t st) {
  if (!logits || !targets || B <= 0 || N <= 0) return VQA_EARG;
  dim3 grid((B + 3) / 4);
  DT(hipLaunchKernelGGL(cross_entropy_kernel<float>, grid, dim3(256), 0, st, (const float*)logits, targets, loss, (float*)dlogits, logits_f32, B, N, gscale, err, ws),
     hipLaunchKernelGGL(cross_entropy_kernel<bf16_t>, grid, dim3(256), 0, st, (const bf16_t*)logits, targets, loss, (bf16_t*)dlogits, logits_f32, B, N, gscale, err, ws));
  if (ws && loss) launch_fold(ws, B, 1, 1, loss, 1, nullptr, st);        // B "rows" of one column
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// dtype_in / dtype_out: 0 = f32, 1 = bf16
int vqa_convert(int dtype_in, int dtype_out, const void* in, void* out, long long n, hipStream_t st) {
  const unsigned g = g1((size_t)n);
  if (!dtype_in && dtype_out) hipLaunchKernelGGL((convert_kernel<float, bf16_t>), dim3(g), dim3(256), 0, st, (const float*)in, (bf16_t*)out, (size_t)n);
  else if (dtype_in && !dtype_out) hipLaunchKernelGGL((convert_kernel<bf16_t, float>), dim3(g), dim3(256), 0, st, (const bf16_t*)in, (float*)out, (size_t)n);
  else if (!dtype_in) hipLaunchKernelGGL((convert_kernel<float, float>), dim3(g), dim3(256), 0, st, (const float*)in, (float*)out, (size_t)n);
  else hipLaunchKernelGGL((convert_kernel<bf16_t, bf16_t>), dim3(g), dim3(256), 0, st, (const bf16_t*)in, (bf16_t*)out, (size_t)n);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_sumsq(const float* g, long long n, float* out, hipStream_t st) {
  size_t blocks = ((size_t)n / 4 + 255) / 256; if (blocks > 2048) blocks = 2048; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(sumsq_kernel, dim3((unsigned)blocks), dim3(256), 0, st, g, (size_t)n, out);
  hipLaunchKernelGGL(sumsq_final_kernel, dim3(1), dim3(256), 0, st, out, (int)blocks);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_adamw(float* p, const float* g, float* m, float* v, long long n, float lr, float b1, float b2, float eps, float wd,
              long long calls, const float* sumsq, float max_norm, float gscale, const int* skip, int* skipped, void* p_bf16, hipStream_t st) {
  if (calls < 1) return VQA_EARG;
  size_t blocks = ((size_t)n + 255) / 256; if (blocks > 4096) blocks = 4096; if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(adamw_kernel, dim3((unsigned)blocks), dim3(256), 0, st, p, g, m, v, (size_t)n, lr, b1, b2, eps, wd, calls, sumsq, max_norm, gscale,
                     skip, skipped, (bf16_t*)p_bf16);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// bf16 MFMA attention forward: one wave per (batch, head), Lq <= 32 queries, Lk <= 64 keys, head dim HD in {32, 64}.
//   S^T (keys x queries) = K . Q^T         v_mfma_f32_32x32x16_bf16, operands loaded straight from global as 16-byte fragments
//   row softmax over keys: the query sits on the lane, its keys in the 32 accumulator registers of the two key tiles
//     (+ one exchange with lane^32); probabilities go to the caller through an LDS staging tile (coalesced rows)
//   ctx = P . V = (P^T)^T . V: the bf16-rounded accumulator registers ARE the A operand of the next MFMA (no lane movement);
//     V is staged in LDS and read transposed with ds_read_b64_tr_b16 in the matching (permuted) k order.
// Same masking semantics as the reference: keys with kmask == 0 get -inf before the softmax (an all-masked row is NaN).
// ---------------------------------------------------------------------------------------------
typedef __attribute__((ext_vector_type(16))) float f32x16;

// NKT = key tiles of 32 (2: Lk <= 64, the 49 image tokens / 20 text tokens; 5: Lk <= 160, the 144 image tokens of the 384x384
// stress shape), WPB = waves (= (batch, head) problems) per workgroup: the wave-private LDS tiles of NKT = 5 fit two per CU.
template <int HD, int NKT, int WPB>
__global__ __launch_bounds__(WPB * 64, 1) void attn_fwd_mfma_kernel(const bf16_t* __restrict__ q, const bf16_t* __restrict__ k, const bf16_t* __restrict__ v,
                                                            int ldq, int ldk, int ldv, const float* __restrict__ kmask, float* __restrict__ probs,
                                                            bf16_t* __restrict__ ctx, int ldc, int BH, int H, int Lq, int Lk, float p, uint64_t seed) {
  constexpr int LDV = HD;                       // V tile row stride (elements): 64 / 128 bytes, 8-byte aligned for the transposed reads
  constexpr int KR = NKT * 32;                  // key rows of the tiles
  constexpr int LDP = KR + 1;                   // probability staging row stride (floats)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * WPB + wave;
  bf16_t* Vs = reinterpret_cast<bf16_t*>(smem) + wave * (KR * LDV);
  float* Ps = reinterpret_cast<float*>(smem + WPB * KR * LDV * 2) + wave * (32 * LDP);
  if (bh >= BH) return;                         // whole wave exits together; no block-level barrier is used below
  const int b = bh / H, h = bh - b * H;
  const int r = lane & 31, hh = lane >> 5;
  const float inv_scale = 1.0f / sqrtf((float)HD);

  // ---- V -> LDS (rows >= Lk zero), 16-byte vectors
  for (int i = lane; i < KR * (HD / 8); i += 64) {
    const int row = i / (HD / 8), cv = i - row * (HD / 8);
    u32x4 val = {0u, 0u, 0u, 0u};
    if (row < Lk) val = *reinterpret_cast<const u32x4*>(v + (size_t)(b * Lk + row) * ldv + h * HD + cv * 8);
    *reinterpret_cast<u32x4*>(&Vs[row * LDV + cv * 8]) = val;
  }
  // ---- S^T = K Q^T
  f32x16 st[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) st[t][e] = 0.f;
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    bf16x8 qf = {};
    if (r < Lq) qf = *reinterpret_cast<const bf16x8*>(q + (size_t)(b * Lq + r) * ldq + h * HD + ks * 16 + 8 * hh);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      bf16x8 kf = {};
      if (32 * t + r < Lk) kf = *reinterpret_cast<const bf16x8*>(k + (size_t)(b * Lk + 32 * t + r) * ldk + h * HD + ks * 16 + 8 * hh);
      st[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf, st[t], 0, 0, 0);
    }
  }
  // ---- scale, mask, softmax over keys (query = lane&31; keys: (e&3) + 8*(e>>2) + 4*hh + 32*t)
  float m = -INFINITY;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      float s = st[t][e] * inv_scale;
      if (key >= Lk || (kmask && kmask[b * Lk + key] == 0.f)) s = -INFINITY;
      st[t][e] = s;
      m = fmaxf(m, s);
    }
  m = fmaxf(m, __shfl_xor(m, 32, 64));
  float sum = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { const float ex = expf(st[t][e] - m); st[t][e] = ex; sum += ex; }   // all -inf row -> NaN like torch
  sum += __shfl_xor(sum, 32, 64);
  const float rs = 1.0f / sum;
  const size_t prow = ((size_t)bh * Lq + r) * Lk;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      float pr = st[t][e] * rs;
      if (sum != sum || m != m) pr = st[t][e] / sum;                       // keep NaN propagation exact (0 * inf etc.)
      Ps[r * LDP + key] = pr;
      if (p > 0.f && r < Lq && key < Lk) pr = drop_keep32(drop_key(seed), (uint32_t)(prow + key), p) ? pr / (1.f - p) : 0.f;
      st[t][e] = pr;
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // wave-private LDS: V tile + probabilities are visible
  for (int i = lane; i < Lq * Lk; i += 64) { const int qi = i / Lk, kj = i - qi * Lk; probs[(size_t)bh * Lq * Lk + i] = Ps[qi * LDP + kj]; }
  // ---- ctx = P V : A operand = bf16(accumulator registers 8s..8s+7), B operand = V rows in the matching permuted key order
  f32x16 o[HD / 32];
#pragma unroll
  for (int c = 0; c < HD / 32; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) o[c][e] = 0.f;
  const int g2 = (lane >> 4) & 1, li = lane & 15, qd = li >> 2, pp = li & 3;
  typedef __attribute__((ext_vector_type(8))) short i16x8;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int s = 0; s < 2; ++s) {
      bf16x8 af;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) af[jj] = (__bf16)st[t][8 * s + jj];
#pragma unroll
      for (int c = 0; c < HD / 32; ++c) {
        const bf16_t* vb = Vs + (32 * t + 16 * s + 4 * hh + qd) * LDV + c * 32 + 16 * g2 + 4 * pp;
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vb));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vb + 8 * LDV));
        i16x8 tt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, __builtin_bit_cast(bf16x8, tt), o[c], 0, 0, 0);
      }
    }
#pragma unroll
  for (int c = 0; c < HD / 32; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (qi < Lq) ctx[(size_t)(b * Lq + qi) * ldc + h * HD + c * 32 + r] = f2bf(o[c][e]);
    }
}

extern "C" int vqa_attention_fwd_mfma(const void* q, const void* k, const void* v, int ldq, int ldk, int ldv, const float* kmask, float* probs,
                                      void* ctx, int ldc, int B, int H, int Lq, int Lk, int hd, float p, unsigned long long seed, hipStream_t st) {
  if (!q || !k || !v || !probs || !ctx || Lq > 32 || Lk > 160 || (hd != 32 && hd != 64) || (ldq % 8) || (ldk % 8) || (ldv % 8)) return VQA_EARG;
  const int BH = B * H;
  auto go = [&](auto kern, int nkt, int wpb) {
    const size_t shm = (size_t)wpb * (nkt * 32) * hd * 2 + (size_t)wpb * 32 * (nkt * 32 + 1) * 4;
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(kern, dim3((BH + wpb - 1) / wpb), dim3(wpb * 64), shm, st, (const bf16_t*)q, (const bf16_t*)k, (const bf16_t*)v, ldq, ldk, ldv,
                       kmask, probs, (bf16_t*)ctx, ldc, BH, H, Lq, Lk, p, seed);
  };
  if (Lk <= 64) { if (hd == 32) go(&attn_fwd_mfma_kernel<32, 2, 4>, 2, 4); else go(&attn_fwd_mfma_kernel<64, 2, 4>, 2, 4); }
  else { if (hd == 32) go(&attn_fwd_mfma_kernel<32, 5, 2>, 5, 2); else go(&attn_fwd_mfma_kernel<64, 5, 2>, 5, 2); }
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

// ---------------------------------------------------------------------------------------------
// bf16 MFMA attention backward: one wave per (batch, head), Lq <= 32, Lk <= 32*NKT (NKT = 2 or 5), head dim HD in {32, 64}.
// dP is formed TWICE from the same global 16-byte fragments (MFMAs are free at this size):
//   orientation 1  dP^T = V . dctx^T   (lane = query, registers = keys)   -> row sums t[q], dS -> dQ = dS . K
//   orientation 2  dP   = dctx . V^T   (lane = key,   registers = queries) -> dS, dropped P  -> dK = dS^T . Q, dV = Pd^T . dctx
// In both, the bf16-rounded accumulator registers are the A operand of the following MFMA (as in the forward kernel) and the
// B operand (K, Q or dctx rows) is read transposed from a wave-private LDS tile in the matching permuted order.
// probs is the softmax BEFORE dropout saved by the forward; the dropout mask is regenerated from (seed, index).
// ---------------------------------------------------------------------------------------------
template <int HD>
__device__ __forceinline__ bf16x8 attn_ldsB(const bf16_t* tile, int row0, int c, int hh, int g2, int qd, int pp) {
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  const bf16_t* vb = tile + (row0 + 4 * hh + qd) * HD + c * 32 + 16 * g2 + 4 * pp;
  i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vb));
  i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(vb + 8 * HD));
  i16x8 tt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  return __builtin_bit_cast(bf16x8, tt);
}

template <int HD, int NKT, int WPB>
__global__ __launch_bounds__(WPB * 64, 1) void attn_bwd_mfma_kernel(const bf16_t* __restrict__ dctx, int ldc, const bf16_t* __restrict__ q,
                                                            const bf16_t* __restrict__ k, const bf16_t* __restrict__ v, int ldq, int ldk, int ldv,
                                                            const float* __restrict__ probs, bf16_t* __restrict__ dq, bf16_t* __restrict__ dk,
                                                            bf16_t* __restrict__ dv, int lddq, int lddk, int lddv, int BH, int H, int Lq, int Lk,
                                                            float p, uint64_t seed) {
  constexpr int KR = NKT * 32;
  constexpr int LDP = KR + 1;
  constexpr int WAVE_BYTES = (KR + 32 + 32) * HD * 2 + 32 * LDP * 4 + 32 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int bh = blockIdx.x * WPB + wave;
  if (bh >= BH) return;                         // whole wave exits together; only wave-private LDS, no block barrier below
  char* base = smem + (size_t)wave * WAVE_BYTES;
  bf16_t* Ks = reinterpret_cast<bf16_t*>(base);            // [KR][HD]
  bf16_t* Qs = Ks + KR * HD;                               // [32][HD]
  bf16_t* Os = Qs + 32 * HD;                               // [32][HD]  (dctx)
  float* Ps = reinterpret_cast<float*>(Os + 32 * HD);      // [32][LDP]
  float* Ts = Ps + 32 * LDP;                               // [32]
  const int b = bh / H, h = bh - b * H;
  const int r = lane & 31, hh = lane >> 5;
  const int g2 = (lane >> 4) & 1, li = lane & 15, qd = li >> 2, pp = li & 3;
  const float inv_scale = 1.0f / sqrtf((float)HD);
  const float keep = p > 0.f ? 1.f / (1.f - p) : 1.f;

  // ---- K, Q, dctx -> LDS (16-byte vectors, padding rows zero); probabilities -> LDS
  for (int i = lane; i < KR * (HD / 8); i += 64) {
    const int row = i / (HD / 8), cv = i - row * (HD / 8);
    u32x4 val = {0u, 0u, 0u, 0u};
    if (row < Lk) val = *reinterpret_cast<const u32x4*>(k + (size_t)(b * Lk + row) * ldk + h * HD + cv * 8);
    *reinterpret_cast<u32x4*>(&Ks[row * HD + cv * 8]) = val;
  }
  for (int i = lane; i < 32 * (HD / 8); i += 64) {
    const int row = i / (HD / 8), cv = i - row * (HD / 8);
    u32x4 a = {0u, 0u, 0u, 0u}, o = {0u, 0u, 0u, 0u};
    if (row < Lq) {
      a = *reinterpret_cast<const u32x4*>(q + (size_t)(b * Lq + row) * ldq + h * HD + cv * 8);
      o = *reinterpret_cast<const u32x4*>(dctx + (size_t)(b * Lq + row) * ldc + h * HD + cv * 8);
    }
    *reinterpret_cast<u32x4*>(&Qs[row * HD + cv * 8]) = a;
    *reinterpret_cast<u32x4*>(&Os[row * HD + cv * 8]) = o;
  }
  for (int i = lane; i < Lq * Lk; i += 64) { const int qi = i / Lk, kj = i - qi * Lk; Ps[qi * LDP + kj] = probs[(size_t)bh * Lq * Lk + i]; }

  // ---- both orientations of dP = dctx V^T from the same global fragments
  f32x16 d1[NKT], d2[NKT];
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) { d1[t][e] = 0.f; d2[t][e] = 0.f; }
#pragma unroll
  for (int ks = 0; ks < HD / 16; ++ks) {
    bf16x8 of = {};
    if (r < Lq) of = *reinterpret_cast<const bf16x8*>(dctx + (size_t)(b * Lq + r) * ldc + h * HD + ks * 16 + 8 * hh);
#pragma unroll
    for (int t = 0; t < NKT; ++t) {
      bf16x8 vf = {};
      if (32 * t + r < Lk) vf = *reinterpret_cast<const bf16x8*>(v + (size_t)(b * Lk + 32 * t + r) * ldv + h * HD + ks * 16 + 8 * hh);
      d1[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, of, d1[t], 0, 0, 0);      // rows = keys, cols = queries
      d2[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(of, vf, d2[t], 0, 0, 0);      // rows = queries, cols = keys
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                     // LDS tiles written above are visible to the whole wave

  // ---- orientation 1: query = r, keys = 32t + (e&3) + 8(e>>2) + 4hh
  const size_t prow = ((size_t)bh * Lq + r) * Lk;
  float tq = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      const bool ok = r < Lq && key < Lk;
      const float pr = ok ? Ps[r * LDP + key] : 0.f;
      float ks = 1.f;
      if (p > 0.f && ok) ks = drop_keep32(drop_key(seed), (uint32_t)(prow + key), p) ? keep : 0.f;
      const float dp = ok ? d1[t][e] * ks : 0.f;
      tq += dp * pr;
      d1[t][e] = dp;
    }
  tq += __shfl_xor(tq, 32, 64);
  if (hh == 0) Ts[r] = tq;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int key = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
      const float pr = (r < Lq && key < Lk) ? Ps[r * LDP + key] : 0.f;
      d1[t][e] = pr * (d1[t][e] - tq) * inv_scale;                     // dS[query r][key]
    }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  // dQ = dS K
  f32x16 oq[HD / 32];
#pragma unroll
  for (int c = 0; c < HD / 32; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) oq[c][e] = 0.f;
#pragma unroll
  for (int t = 0; t < NKT; ++t)
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 af;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) af[jj] = (__bf16)d1[t][8 * s2 + jj];
#pragma unroll
      for (int c = 0; c < HD / 32; ++c)
        oq[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af, attn_ldsB<HD>(Ks, 32 * t + 16 * s2, c, hh, g2, qd, pp), oq[c], 0, 0, 0);
    }
#pragma unroll
  for (int c = 0; c < HD / 32; ++c)
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = (e & 3) + 8 * (e >> 2) + 4 * hh;
      if (qi < Lq) dq[(size_t)(b * Lq + qi) * lddq + h * HD + c * 32 + r] = f2bf(oq[c][e]);
    }

  // ---- orientation 2: key = 32t + r, queries = (e&3) + 8(e>>2) + 4hh ; dK = dS^T Q, dV = Pd^T dctx
#pragma unroll
  for (int t = 0; t < NKT; ++t) {
    const int key = 32 * t + r;
    f32x16 ds2, pd2;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
      const int qi = (e & 3) + 8 * (e >> 2) + 4 * hh;
      const bool ok = qi < Lq && key < Lk;
      const float pr = ok ? Ps[qi * LDP + key] : 0.f;
      float ks = 1.f;
      if (p > 0.f && ok) ks = drop_keep32(drop_key(seed), (uint32_t)(((size_t)bh * Lq + qi) * Lk + key), p) ? keep : 0.f;
      const float dp = ok ? d2[t][e] * ks : 0.f;
      const float tt = ok ? Ts[qi] : 0.f;
      ds2[e] = pr * (dp - tt) * inv_scale;
      pd2[e] = pr * ks;
    }
    f32x16 okk[HD / 32], ovv[HD / 32];
#pragma unroll
    for (int c = 0; c < HD / 32; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) { okk[c][e] = 0.f; ovv[c][e] = 0.f; }
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) {
      bf16x8 as, ap;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) { as[jj] = (__bf16)ds2[8 * s2 + jj]; ap[jj] = (__bf16)pd2[8 * s2 + jj]; }
#pragma unroll
      for (int c = 0; c < HD / 32; ++c) {
        okk[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(as, attn_ldsB<HD>(Qs, 16 * s2, c, hh, g2, qd, pp), okk[c], 0, 0, 0);
        ovv[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap, attn_ldsB<HD>(Os, 16 * s2, c, hh, g2, qd, pp), ovv[c], 0, 0, 0);
      }
    }
#pragma unroll
    for (int c = 0; c < HD / 32; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const int kj = 32 * t + (e & 3) + 8 * (e >> 2) + 4 * hh;
        if (kj < Lk) {
          dk[(size_t)(b * Lk + kj) * lddk + h * HD + c * 32 + r] = f2bf(okk[c][e]);
          dv[(size_t)(b * Lk + kj) * lddv + h * HD + c * 32 + r] = f2bf(ovv[c][e]);
        }
      }
  }
}

extern "C" int vqa_attention_bwd_mfma(const void* dctx, int ldc, const void* q, const void* k, const void* v, int ldq, int ldk, int ldv,
                                      const float* probs, void* dq, void* dk, void* dv, int lddq, int lddk, int lddv,
                                      int B, int H, int Lq, int Lk, int hd, float p, unsigned long long seed, hipStream_t st) {
  if (!dctx || !q || !k || !v || !probs || !dq || !dk || !dv || Lq > 32 || Lk > 160 || (hd != 32 && hd != 64) ||
      (ldq % 8) || (ldk % 8) || (ldv % 8) || (ldc % 8)) return VQA_EARG;
  const int BH = B * H;
  auto go = [&](auto kern, int nkt, int wpb) {
    const size_t shm = (size_t)wpb * ((nkt * 32 + 32 + 32) * hd * 2 + 32 * (nkt * 32 + 1) * 4 + 32 * 4);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm);
    hipLaunchKernelGGL(kern, dim3((BH + wpb - 1) / wpb), dim3(wpb * 64), shm, st, (const bf16_t*)dctx, ldc, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, ldq, ldk, ldv, probs, (bf16_t*)dq, (bf16_t*)dk, (bf16_t*)dv, lddq, lddk, lddv, BH, H, Lq, Lk, p, seed);
  };
  if (Lk <= 64) { if (hd == 32) go(&attn_bwd_mfma_kernel<32, 2, 4>, 2, 4); else go(&attn_bwd_mfma_kernel<64, 2, 4>, 2, 4); }
  else { if (hd == 32) go(&attn_bwd_mfma_kernel<32, 5, 2>, 5, 2); else go(&attn_bwd_mfma_kernel<64, 5, 2>, 5, 2); }
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

// ---------------------------------------------------------------------------------------------
// Device-side accuracy counters (utils/metrics.py:55-94 VQAAccuracy.update without its .cpu()/.item() syncs).
// One wave per sample: rank of the target = #{j : x[j] > x[t]} + #{j < t : x[j] == x[t]} (ties resolve to the lowest index,
// like argmax); top-1 correct <=> rank == 0, top-5 correct <=> rank < 5.  counters = {correct, correct_top5, total} (u64, +=).
// A target outside [0, N) counts as wrong (the reference's comparison can never match it either).
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void accuracy_kernel(const float* __restrict__ logits, const long long* __restrict__ targets,
                                                       unsigned long long* counters, int B, int N) {
  const int lane = threadIdx.x & 63;
  const int row = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  if (row >= B) return;
  const long long t = targets[row];
  const float* x = logits + (size_t)row * N;
  float cnt = 0.f;
  const bool valid = t >= 0 && t < N;
  if (valid) {
    const float xt = x[t];
    for (int j = lane; j < N; j += 64) {
      const float v = x[j];
      cnt += (v > xt || (v == xt && j < (int)t)) ? 1.f : 0.f;
    }
  }
  cnt = wave_sum(cnt);
  if (lane == 0) {
    if (valid && cnt < 0.5f) atomicAdd(counters + 0, 1ull);
    if (valid && cnt < 4.5f) atomicAdd(counters + 1, 1ull);
    atomicAdd(counters + 2, 1ull);
  }
}

extern "C" int vqa_accuracy_update(const float* logits, const long long* targets, unsigned long long* counters, int B, int N, hipStream_t st) {
  if (!logits || !targets || !counters || B <= 0 || N <= 0) return VQA_EARG;
  hipLaunchKernelGGL(accuracy_kernel, dim3((B + 3) / 4), dim3(256), 0, st, logits, targets, counters, B, N);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
