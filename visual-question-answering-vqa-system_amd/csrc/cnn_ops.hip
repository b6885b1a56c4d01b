// HBM-bound CNN-side kernels for gfx950: BatchNorm (train statistics, apply, backward), the stem's fused
// BN+ReLU+MaxPool, squeeze-excitation and spatial attention.  Activations are NHWC with T = float | bf16,
// every per-channel statistic / coefficient is fp32, every global access is a 16-byte vector.
//   BatchNorm2d        models/cnn_backbone.py:151,158,246,351 (nn.BatchNorm2d defaults)
//   ReLU + MaxPool     models/cnn_backbone.py:352-353
//   residual add+ReLU  models/cnn_backbone.py:194-195
//   SEAttention        models/attention_modules.py:109-136
//   SpatialAttention   models/attention_modules.py:223-243
#include "common.h"
#include "stem_route.h"

// ---------------------------------------------------------------------------------------------
// BatchNorm statistics: reduce the igemm partial slab [tiles][2][C] -> mean / invstd / scale / shift
// ---------------------------------------------------------------------------------------------
__global__ void bn_reduce_partials_kernel(const float* __restrict__ part, double* __restrict__ acc, int tiles, int C) {
  // grid (ceil(C/64), G); block 256 = 4 tile-lanes x 64 channels
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), tl = threadIdx.x >> 6;
  __shared__ double sh[2][4][64];
  double s = 0.0, q = 0.0;
  if (c < C) {
    // 8 slab rows in flight per thread: the loop is latency-bound, not bandwidth-bound
    const int step = gridDim.y * 4;
    int t = blockIdx.y * 4 + tl;
    for (; t + 7 * step < tiles; t += 8 * step) {
      float a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = part[((size_t)(t + u * step) * 2) * C + c]; b[u] = part[((size_t)(t + u * step) * 2 + 1) * C + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s += (double)a[u]; q += (double)b[u]; }
    }
    for (; t < tiles; t += step) {
      s += (double)part[((size_t)t * 2) * C + c];
      q += (double)part[((size_t)t * 2 + 1) * C + c];
    }
  }
  sh[0][tl][threadIdx.x & 63] = s; sh[1][tl][threadIdx.x & 63] = q;
  __syncthreads();
  if (tl == 0 && c < C) {
    for (int i = 1; i < 4; ++i) { s += sh[0][i][threadIdx.x]; q += sh[1][i][threadIdx.x]; }
    acc[((size_t)blockIdx.y * 2) * C + c] = s;
    acc[((size_t)blockIdx.y * 2 + 1) * C + c] = q;
  }
}

// coef layout (fp32, 4*C): scale | shift | mean | invstd
__global__ void bn_finalize_kernel(const double* __restrict__ acc, int G, int C, double count, const float* __restrict__ gamma,
                                   const float* __restrict__ beta, float* running_mean, float* running_var, long long* nbt,
                                   float momentum, float eps, float* __restrict__ coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  double s = 0.0, q = 0.0;
  int g = 0;
  for (; g + 8 <= G; g += 8) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) { a[u] = acc[((size_t)(g + u) * 2) * C + c]; b[u] = acc[((size_t)(g + u) * 2 + 1) * C + c]; }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s += a[u]; q += b[u]; }
  }
  for (; g < G; ++g) { s += acc[((size_t)g * 2) * C + c]; q += acc[((size_t)g * 2 + 1) * C + c]; }
  const double mean = s / count;
  double var = q / count - mean * mean;
  if (var < 0.0) var = 0.0;
  const float invstd = (float)(1.0 / sqrt(var + (double)eps));
  const float sc = gamma[c] * invstd;
  coef[c] = sc; coef[C + c] = beta[c] - (float)mean * sc; coef[2 * C + c] = (float)mean; coef[3 * C + c] = invstd;
  if (running_mean) {
    const double unbiased = count > 1.0 ? var * count / (count - 1.0) : var;
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)unbiased;
    if (c == 0 && nbt) *nbt += 1;
  }
}

__global__ void bn_eval_coef_kernel(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* coef) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= C) return;
  const float invstd = 1.0f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * invstd;
  coef[c] = sc; coef[C + c] = beta[c] - rm[c] * sc; coef[2 * C + c] = rm[c]; coef[3 * C + c] = invstd;
}

// out = [relu]( y*scale+shift  [+ res*rscale+rshift | + res] ).  Each thread owns one 16-byte channel vector
// (coefficients stay in registers) and walks rows; a block covers 256/cv consecutive rows = one contiguous span.
// RES: 0 = none, 1 = + res (identity), 2 = + res*rscale+rshift (the 1x1 shortcut's BatchNorm)
template <typename T, int RES>
__global__ __launch_bounds__(256) void bn_apply_kernel(const T* __restrict__ y, const float* __restrict__ coef, const T* __restrict__ res,
                                const float* __restrict__ rcoef, T* __restrict__ out, size_t rows, int C, int relu) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv;
  const int c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  float sc[VEC], sh[VEC], rs[RES == 2 ? VEC : 1], rh[RES == 2 ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = coef[c0 + j]; sh[j] = coef[C + c0 + j];
    if (RES == 2) { rs[j] = rcoef[c0 + j]; rh[j] = rcoef[C + c0 + j]; }
  }
#pragma unroll 2
  for (size_t r = (size_t)blockIdx.x * lanes_r + myr; r < rows; r += (size_t)gridDim.x * lanes_r) {
    const size_t off = r * C + c0;
    Vec16<T> v = ldg16(y + off), rr, o;
    if (RES) rr = ldg16(res + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float x = v.get(j) * sc[j] + sh[j];
      if (RES == 1) x += rr.get(j);
      if (RES == 2) x += rr.get(j) * rs[j] + rh[j];
      o.set(j, (relu && x < 0.f) ? 0.f : x);   // NaN-propagating ReLU like torch
    }
    stg16(out + off, o);
  }
}

// The same map for the LAST residual block of a stage, with the squeeze-excitation global average pool folded in
// (models/cnn_backbone.py:194-197 -> models/attention_modules.py:109-112): a workgroup owns rows [chunk*rpc, (chunk+1)*rpc) of ONE
// sample and also writes the column sums of the values it stores (bf16-rounded, exactly what a pooling pass would read back) to
// part[b][chunk][C]; se_pool_fc_kernel folds the chunks in index order.  Saves the pooling pass's read of the stage output.
template <typename T, int RES>
__global__ __launch_bounds__(256) void bn_apply_pool_kernel(const T* __restrict__ y, const float* __restrict__ coef, const T* __restrict__ res,
                                                            const float* __restrict__ rcoef, T* __restrict__ out, int HW, int C, int relu, int rpc,
                                                            float* __restrict__ part) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv;
  const int c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  float sc[VEC], sh[VEC], rs[RES == 2 ? VEC : 1], rh[RES == 2 ? VEC : 1], acc[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = coef[c0 + j]; sh[j] = coef[C + c0 + j]; acc[j] = 0.f;
    if (RES == 2) { rs[j] = rcoef[c0 + j]; rh[j] = rcoef[C + c0 + j]; }
  }
  const int b = blockIdx.y, r0 = blockIdx.x * rpc, r1 = min(HW, r0 + rpc);
  const size_t base = (size_t)b * HW * C + c0;
#pragma unroll 2
  for (int r = r0 + myr; r < r1; r += lanes_r) {
    const size_t off = base + (size_t)r * C;
    Vec16<T> v = ldg16(y + off), rr, o;
    if (RES) rr = ldg16(res + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float x = v.get(j) * sc[j] + sh[j];
      if (RES == 1) x += rr.get(j);
      if (RES == 2) x += rr.get(j) * rs[j] + rh[j];
      o.set(j, (relu && x < 0.f) ? 0.f : x);
      acc[j] += o.get(j);
    }
    stg16(out + off, o);
  }
  __shared__ float shs[256 * VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) shs[threadIdx.x * VEC + j] = acc[j];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int v = c / VEC, j = c - v * VEC;
    float t = 0.f;
    for (int r = 0; r < lanes_r; ++r) t += shs[(r * cv + v) * VEC + j];
    part[((size_t)b * gridDim.x + blockIdx.x) * C + c] = t;
  }
}

// Train-mode BatchNorm whose batch statistics arrive as fixed-point accumulators (common.h acc_add_fixed; filled by the conv
// kernel's epilogue): every thread finalizes the coefficients of ITS channels in the prologue (fp64, same formulas as
// bn_finalize_kernel), the first workgroup also publishes coef[4][C] (scale | shift | mean | invstd: the backward reads it) and
// updates running_mean / running_var / num_batches_tracked (nn.BatchNorm2d defaults).  No finalize launch in between.
// (BnAcc / bn_acc_coef: common.h -- shared with the stage-1 patch kernels, which finalize the statistics of their INPUT's BatchNorm)
// RES: 0 none, 1 + res, 2 + BatchNorm(res) with its own accumulators.  POOL: grid (chunks, B), SE pooling sums to part (see above).
template <typename T, int RES, bool POOL>
__global__ __launch_bounds__(256) void bn_apply_acc_kernel(const T* __restrict__ y, BnAcc f, const T* __restrict__ res, BnAcc fr,
                                                           T* __restrict__ out, size_t rows, int HW, int C, int relu, int rpc, double inv_count,
                                                           double unbias, float momentum, float eps, float* __restrict__ part) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv;
  const int c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  const bool writer = blockIdx.x == 0 && blockIdx.y == 0;
  extern __shared__ float cf[];                // [2 | 4][C]: scale, shift (, residual scale, shift)
  for (int c = threadIdx.x; c < C; c += 256) {
    bn_acc_coef(f, C, c, inv_count, unbias, momentum, eps, writer, cf[c], cf[C + c]);
    if (RES == 2) bn_acc_coef(fr, C, c, inv_count, unbias, momentum, eps, writer, cf[2 * C + c], cf[3 * C + c]);
  }
  __syncthreads();
  float sc[VEC], sh[VEC], rs[RES == 2 ? VEC : 1], rh[RES == 2 ? VEC : 1], acc[POOL ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = cf[c0 + j]; sh[j] = cf[C + c0 + j];
    if (RES == 2) { rs[j] = cf[2 * C + c0 + j]; rh[j] = cf[3 * C + c0 + j]; }
    if (POOL) acc[j] = 0.f;
  }
  size_t r, r1, rstep, base = c0;
  if (POOL) { const int r0 = blockIdx.x * rpc; r = r0 + myr; r1 = min(HW, r0 + rpc); rstep = lanes_r; base += (size_t)blockIdx.y * HW * C; }
  else { r = (size_t)blockIdx.x * lanes_r + myr; r1 = rows; rstep = (size_t)gridDim.x * lanes_r; }
#pragma unroll 2
  for (; r < r1; r += rstep) {
    const size_t off = base + r * C;
    Vec16<T> v = ldg16(y + off), rr, o;
    if (RES) rr = ldg16(res + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float x = v.get(j) * sc[j] + sh[j];
      if (RES == 1) x += rr.get(j);
      if (RES == 2) x += rr.get(j) * rs[j] + rh[j];
      o.set(j, (relu && x < 0.f) ? 0.f : x);
      if (POOL) acc[j] += o.get(j);
    }
    stg16(out + off, o);
  }
  if constexpr (POOL) {
    __shared__ float shs[256 * VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) shs[threadIdx.x * VEC + j] = acc[j];
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += 256) {
      const int v = c / VEC, j = c - v * VEC;
      float t = 0.f;
      for (int q = 0; q < lanes_r; ++q) t += shs[(q * cv + v) * VEC + j];
      part[((size_t)blockIdx.y * gridDim.x + blockIdx.x) * C + c] = t;
    }
  }
}

// ---------------------------------------------------------------------------------------------
// BatchNorm backward.  g = dout * (out > 0) (relu) or dout.  Partial sums per block -> slab [blocks][3][C]:
//   0: sum g   1: sum g*xhat(y)   2: sum g*xhat(y2) (second BN sharing g: the 1x1 shortcut)
// ---------------------------------------------------------------------------------------------
template <typename T, bool SELF, bool DUAL>
__global__ __launch_bounds__(256) void bn_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ outact, const T* __restrict__ y,
                                                            const float* __restrict__ coef, const T* __restrict__ y2,
                                                            const float* __restrict__ coef2, float* __restrict__ slab, size_t rows, int C,
                                                            unsigned long long* __restrict__ facc) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC;                 // vectors per row
  const int lanes_r = 256 / cv;           // rows processed concurrently by the block (C <= 256*VEC)
  const int myv = threadIdx.x % cv, myr = threadIdx.x / cv, c0 = myv * VEC;
  float sg[VEC], sx[VEC], sx2[DUAL ? VEC : 1], mean[VEC], inv[VEC], ms[SELF ? VEC : 1], mh[SELF ? VEC : 1], mean2[DUAL ? VEC : 1], inv2[DUAL ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sg[j] = sx[j] = 0.f; mean[j] = coef[2 * C + c0 + j]; inv[j] = coef[3 * C + c0 + j];
    if (DUAL) { sx2[j] = 0.f; mean2[j] = coef2[2 * C + c0 + j]; inv2[j] = coef2[3 * C + c0 + j]; }
    if (SELF) { ms[j] = coef[c0 + j]; mh[j] = coef[C + c0 + j]; }
  }
#pragma unroll 2
  for (size_t r = (size_t)blockIdx.x * lanes_r + myr; r < rows; r += (size_t)gridDim.x * lanes_r) {
    const size_t off = r * C + c0;
    Vec16<T> d = ldg16(dout + off), yy = ldg16(y + off), o, y2v;
    if (!SELF && outact) o = ldg16(outact + off);
    if (DUAL) y2v = ldg16(y2 + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float g = d.get(j);
      if (SELF) { if (!(yy.get(j) * ms[j] + mh[j] > 0.f)) g = 0.f; }       // relu(bn(y)) > 0 recomputed from y
      else if (outact && !(o.get(j) > 0.f)) g = 0.f;
      sg[j] += g;
      sx[j] += g * (yy.get(j) - mean[j]) * inv[j];
      if (DUAL) sx2[j] += g * (y2v.get(j) - mean2[j]) * inv2[j];
    }
  }
  extern __shared__ float shm[];          // [3][256][VEC]
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    shm[(0 * 256 + threadIdx.x) * VEC + j] = sg[j];
    shm[(1 * 256 + threadIdx.x) * VEC + j] = sx[j];
    shm[(2 * 256 + threadIdx.x) * VEC + j] = DUAL ? sx2[j] : 0.f;
  }
  __syncthreads();
  for (int o = threadIdx.x; o < 3 * C; o += 256) {
    const int k = o / C, c = o - k * C, v = c / VEC, j = c - v * VEC;
    float s = 0.f;
    for (int r = 0; r < lanes_r; ++r) s += shm[(k * 256 + r * cv + v) * VEC + j];
    if (facc) {                               // fixed point: order-free, no finalize launch; replica blockIdx.x % R
      const int R = acc_replicas(C);
      if (k < 2 || DUAL) acc_add_fixed(facc, (size_t)R * 3 * C, ((size_t)(blockIdx.x % R) * 3 + k) * C + c, s);
    }
    else slab[((size_t)blockIdx.x * 3 + k) * C + c] = s;
  }
}

// reduce slab over blocks; emit dgamma/dbeta (+=) and the apply coefficients  dy = A*g + Bc*y + Cc
// bcoef layout (3*C): A | Bc | Cc
__global__ __launch_bounds__(1024) void bn_bwd_finalize_kernel(const float* __restrict__ slab, int nblk, int C, int which, double count,
                                       const float* __restrict__ gamma, const float* __restrict__ coef, int training,
                                       float* dgamma, float* dbeta, float* __restrict__ bcoef) {
  const int c = blockIdx.x * 64 + (threadIdx.x & 63), tl = threadIdx.x >> 6;
  __shared__ double sh[2][16][64];
  double sg = 0.0, sx = 0.0;
  if (c < C) {
    int b = tl;
    for (; b + 7 * 16 < nblk; b += 8 * 16) {               // 8 slab rows in flight per thread (latency-bound loop)
      float a[8], x[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) { a[u] = slab[((size_t)(b + 16 * u) * 3) * C + c]; x[u] = slab[((size_t)(b + 16 * u) * 3 + which) * C + c]; }
#pragma unroll
      for (int u = 0; u < 8; ++u) { sg += a[u]; sx += x[u]; }
    }
    for (; b < nblk; b += 16) { sg += slab[((size_t)b * 3) * C + c]; sx += slab[((size_t)b * 3 + which) * C + c]; }
  }
  sh[0][tl][threadIdx.x & 63] = sg; sh[1][tl][threadIdx.x & 63] = sx;
  __syncthreads();
  if (tl != 0 || c >= C) return;
  for (int i = 1; i < 16; ++i) { sg += sh[0][i][threadIdx.x]; sx += sh[1][i][threadIdx.x]; }
  if (dgamma) dgamma[c] += (float)sx;
  if (dbeta) dbeta[c] += (float)sg;
  const float mean = coef[2 * C + c], invstd = coef[3 * C + c], gi = gamma[c] * invstd;
  if (training) {
    const float mg = (float)(sg / count), mgx = (float)(sx / count);
    bcoef[c] = gi; bcoef[C + c] = -gi * invstd * mgx; bcoef[2 * C + c] = gi * (mean * invstd * mgx - mg);
  } else { bcoef[c] = gi; bcoef[C + c] = 0.f; bcoef[2 * C + c] = 0.f; }
}

// dy = A*g + B*y + C ; optional second output dy2 = A2*g + B2*y2 + C2 ; optional gout = g (T) for the identity path
// SELF: the ReLU mask is relu(bn(y)) > 0 recomputed from y with mcoef (scale, shift); otherwise outact > 0 (or none).
// DUAL: second BN (1x1 shortcut) sharing g.  Specialised so unused coefficient sets cost no registers.
template <typename T, bool SELF, bool DUAL>
__global__ __launch_bounds__(256) void bn_bwd_apply_kernel(const T* __restrict__ dout, const T* __restrict__ outact, const T* __restrict__ y,
                                    const float* __restrict__ bc, T* __restrict__ dy, const T* __restrict__ y2,
                                    const float* __restrict__ bc2, T* __restrict__ dy2, size_t rows, int C,
                                    const float* __restrict__ mcoef) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv;
  const int c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  float a[VEC], b[VEC], c[VEC], a2[DUAL ? VEC : 1], b2[DUAL ? VEC : 1], c2[DUAL ? VEC : 1], ms[SELF ? VEC : 1], mh[SELF ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    a[j] = bc[c0 + j]; b[j] = bc[C + c0 + j]; c[j] = bc[2 * C + c0 + j];
    if (DUAL) { a2[j] = bc2[c0 + j]; b2[j] = bc2[C + c0 + j]; c2[j] = bc2[2 * C + c0 + j]; }
    if (SELF) { ms[j] = mcoef[c0 + j]; mh[j] = mcoef[C + c0 + j]; }
  }
#pragma unroll 2
  for (size_t r = (size_t)blockIdx.x * lanes_r + myr; r < rows; r += (size_t)gridDim.x * lanes_r) {
    const size_t off = r * C + c0;
    Vec16<T> d = ldg16(dout + off), yy = ldg16(y + off), o, y2v, rr, r2;
    if (!SELF && outact) o = ldg16(outact + off);
    if (DUAL) y2v = ldg16(y2 + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float g = d.get(j);
      if (SELF) { if (!(yy.get(j) * ms[j] + mh[j] > 0.f)) g = 0.f; }
      else if (outact && !(o.get(j) > 0.f)) g = 0.f;
      rr.set(j, a[j] * g + b[j] * yy.get(j) + c[j]);
      if (DUAL) r2.set(j, a2[j] * g + b2[j] * y2v.get(j) + c2[j]);
    }
    stg16(dy + off, rr);
    if (DUAL) stg16(dy2 + off, r2);
  }
}

// The same map with the finalize folded into the prologue: the column sums arrive as fixed-point accumulators facc[3][C] (+ flag),
// every thread derives A | B | C of its channels (formulas of bn_bwd_finalize_kernel, training mode), the first workgroup adds
// d gamma / d beta (and the shortcut BatchNorm's, DUAL) into the gradient buffer.
template <typename T, bool SELF, bool DUAL>
__global__ __launch_bounds__(256) void bn_bwd_apply_acc_kernel(const T* __restrict__ dout, const T* __restrict__ outact, const T* __restrict__ y,
                                    const unsigned long long* __restrict__ facc, const float* __restrict__ gamma, const float* __restrict__ coef,
                                    float* dgamma, float* dbeta, T* __restrict__ dy, const T* __restrict__ y2, const float* __restrict__ gamma2,
                                    const float* __restrict__ coef2, float* dgamma2, float* dbeta2, T* __restrict__ dy2, size_t rows, int C,
                                    double inv_count) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv;
  const int c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  const bool writer = blockIdx.x == 0;
  const int R = acc_replicas(C);
  const bool bad = acc_flagged(facc, R, 3, C);
  extern __shared__ float cf[];                // [3 | 6][C]: A, B, C (, A2, B2, C2): one evaluation per channel and workgroup
  for (int ch = threadIdx.x; ch < C; ch += 256) {
    double sg = acc_read_fixed(facc, R, 3, C, 0, ch), sx = acc_read_fixed(facc, R, 3, C, 1, ch);
    if (bad) sg = sx = __builtin_nan("");
    const float mean = coef[2 * C + ch], invstd = coef[3 * C + ch], gi = gamma[ch] * invstd;
    const float mg = (float)(sg * inv_count), mgx = (float)(sx * inv_count);
    cf[ch] = gi; cf[C + ch] = -gi * invstd * mgx; cf[2 * C + ch] = gi * (mean * invstd * mgx - mg);
    if (writer) { dgamma[ch] += (float)sx; dbeta[ch] += (float)sg; }
    if (DUAL) {
      double sx2 = acc_read_fixed(facc, R, 3, C, 2, ch);
      if (bad) sx2 = __builtin_nan("");
      const float mean2 = coef2[2 * C + ch], inv2 = coef2[3 * C + ch], gi2 = gamma2[ch] * inv2, mgx2 = (float)(sx2 * inv_count);
      cf[3 * C + ch] = gi2; cf[4 * C + ch] = -gi2 * inv2 * mgx2; cf[5 * C + ch] = gi2 * (mean2 * inv2 * mgx2 - mg);
      if (writer) { dgamma2[ch] += (float)sx2; dbeta2[ch] += (float)sg; }
    }
  }
  __syncthreads();
  float a[VEC], b[VEC], c[VEC], a2[DUAL ? VEC : 1], b2[DUAL ? VEC : 1], c2[DUAL ? VEC : 1], ms[SELF ? VEC : 1], mh[SELF ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    const int ch = c0 + j;
    a[j] = cf[ch]; b[j] = cf[C + ch]; c[j] = cf[2 * C + ch];
    if (DUAL) { a2[j] = cf[3 * C + ch]; b2[j] = cf[4 * C + ch]; c2[j] = cf[5 * C + ch]; }
    if (SELF) { ms[j] = coef[ch]; mh[j] = coef[C + ch]; }
  }
#pragma unroll 2
  for (size_t r = (size_t)blockIdx.x * lanes_r + myr; r < rows; r += (size_t)gridDim.x * lanes_r) {
    const size_t off = r * C + c0;
    Vec16<T> d = ldg16(dout + off), yy = ldg16(y + off), o, y2v, rr, r2;
    if (!SELF && outact) o = ldg16(outact + off);
    if (DUAL) y2v = ldg16(y2 + off);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float g = d.get(j);
      if (SELF) { if (!(yy.get(j) * ms[j] + mh[j] > 0.f)) g = 0.f; }
      else if (outact && !(o.get(j) > 0.f)) g = 0.f;
      rr.set(j, a[j] * g + b[j] * yy.get(j) + c[j]);
      if (DUAL) r2.set(j, a2[j] * g + b2[j] * y2v.get(j) + c2[j]);
    }
    stg16(dy + off, rr);
    if (DUAL) stg16(dy2 + off, r2);
  }
}

// ---------------------------------------------------------------------------------------------
// Stem tail: BN + ReLU + MaxPool3x3/2 p1 fused (the 112x112 activation is never written)
// ---------------------------------------------------------------------------------------------
// A workgroup owns a tile of 4 output rows x 7 output pairs (14 columns) x all channel vectors: its 9 x 29 input pixels are
// fetched once per tile through the CU's caches (1.13x the tensor); a row-major walk over the outputs fetched every input row for
// two output rows from HBM (PMC: 1.5x).  A thread owns one channel vector of TWO horizontally adjacent outputs: their 3x3/2
// windows share a column, so 15 loads (BN + ReLU applied once each) serve 2 outputs; the argmax bytes leave as packed stores.
constexpr int SP_ROWS = 4, SP_PAIRS = 7;
template <typename T>
__global__ __launch_bounds__(256) void stem_pool_fwd_kernel(const T* __restrict__ y, const float* __restrict__ coef, T* __restrict__ out,
                                                            uint8_t* __restrict__ idx, int B, int H, int W, int C, int Ho, int Wo, int trows) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, Wp = (Wo + 1) >> 1;
  const int tiles_x = (Wp + SP_PAIRS - 1) / SP_PAIRS, tiles_y = (Ho + trows - 1) / trows;
  const int per_row = SP_PAIRS * cv;                       // threads per output row of the tile (host: per_row * trows <= 256)
  const int ry = threadIdx.x / per_row, rem = threadIdx.x - ry * per_row;
  if (ry >= trows) return;
  const int pr = rem / cv, c0 = (rem - pr * cv) * VEC;
  int t = blockIdx.x;
  const int tx = t % tiles_x; t /= tiles_x; const int ty = t % tiles_y; const int b = t / tiles_y;
  const int oh = ty * trows + ry, owp = tx * SP_PAIRS + pr;
  if (oh >= Ho || owp >= Wp) return;
  float sc[VEC], sh[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sc[j] = coef[c0 + j]; sh[j] = coef[C + c0 + j]; }
  const int ow0 = owp * 2;
  const bool two = ow0 + 1 < Wo;
  float best[2][VEC]; int bi[2][VEC];
#pragma unroll
  for (int o = 0; o < 2; ++o)
#pragma unroll
    for (int j = 0; j < VEC; ++j) { best[o][j] = -INFINITY; bi[o][j] = 0; }
#pragma unroll
  for (int r = 0; r < 3; ++r) {
    const int ih = oh * 2 - 1 + r;
    if (ih < 0 || ih >= H) continue;
    const T* yr = y + (((size_t)b * H + ih) * W) * C + c0;
#pragma unroll
    for (int s5 = 0; s5 < 5; ++s5) {                         // input columns 2*ow0 - 1 .. 2*ow0 + 3
      const int iw = ow0 * 2 - 1 + s5;
      if (iw < 0 || iw >= W) continue;
      const Vec16<T> yy = ldg16(yr + (size_t)iw * C);
      float a[VEC];
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float v = yy.get(j) * sc[j] + sh[j]; a[j] = v < 0.f ? 0.f : v; }
      if (s5 < 3) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (a[j] > best[0][j] || a[j] != a[j]) { best[0][j] = a[j]; bi[0][j] = r * 3 + s5; }     // first max wins, NaN propagates (ATen max_pool2d)
      }
      if (s5 >= 2) {
#pragma unroll
        for (int j = 0; j < VEC; ++j)
          if (a[j] > best[1][j] || a[j] != a[j]) { best[1][j] = a[j]; bi[1][j] = r * 3 + s5 - 2; }
      }
    }
  }
#pragma unroll
  for (int o = 0; o < 2; ++o) {
    if (o == 1 && !two) break;
    const size_t e = ((((size_t)b * Ho + oh) * Wo + ow0 + o) * C + c0);
    Vec16<T> ov;
    uint32_t pk[VEC / 4];
#pragma unroll
    for (int q4 = 0; q4 < VEC / 4; ++q4) pk[q4] = 0u;
#pragma unroll
    for (int j = 0; j < VEC; ++j) { ov.set(j, best[o][j]); pk[j >> 2] |= (uint32_t)bi[o][j] << (8 * (j & 3)); }
    stg16(out + e, ov);
#pragma unroll
    for (int q4 = 0; q4 < VEC / 4; ++q4) reinterpret_cast<uint32_t*>(idx + e)[q4] = pk[q4];
  }
}

template <typename T>
__global__ __launch_bounds__(256) void stem_bwd_reduce_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y,
                                                              const float* __restrict__ coef, float* __restrict__ slab,
                                                              int B, int H, int W, int C, int Ho, int Wo) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv;
  const int myv = threadIdx.x % cv, myr = threadIdx.x / cv, c0 = myv * VEC;
  const size_t rows = (size_t)B * H * W;
  float sg[VEC], sx[VEC], g[VEC], sc[VEC], sh[VEC], mean[VEC], inv[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { sg[j] = sx[j] = 0.f; sc[j] = coef[c0 + j]; sh[j] = coef[C + c0 + j]; mean[j] = coef[2 * C + c0 + j]; inv[j] = coef[3 * C + c0 + j]; }
  if (myr < lanes_r)
    for (size_t r = (size_t)blockIdx.x * lanes_r + myr; r < rows; r += (size_t)gridDim.x * lanes_r) {
      const int w = (int)(r % W); size_t q = r / W; const int h = (int)(q % H); const int b = (int)(q / H);
      Vec16<T> yy = ldg16(y + r * C + c0);
      stem_route<T>(dpool, idx, yy, sc, sh, b, h, w, c0, C, Ho, Wo, g);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { sg[j] += g[j]; sx[j] += g[j] * (yy.get(j) - mean[j]) * inv[j]; }
    }
  extern __shared__ float shm[];
#pragma unroll
  for (int j = 0; j < VEC; ++j) { shm[(0 * 256 + threadIdx.x) * VEC + j] = sg[j]; shm[(1 * 256 + threadIdx.x) * VEC + j] = sx[j]; }
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * C; o += 256) {
    const int k = o / C, c = o - k * C, v = c / VEC, j = c - v * VEC;
    float s = 0.f;
    for (int r = 0; r < lanes_r; ++r) s += shm[(k * 256 + r * cv + v) * VEC + j];
    slab[((size_t)blockIdx.x * 3 + k) * C + c] = s;
    if (k == 0) slab[((size_t)blockIdx.x * 3 + 2) * C + c] = 0.f;
  }
}

template <typename T>
__global__ void stem_bwd_apply_kernel(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const T* __restrict__ y,
                                      const float* __restrict__ coef, const float* __restrict__ bc, T* __restrict__ dy,
                                      int B, int H, int W, int C, int Ho, int Wo) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC;
  const size_t total = (size_t)B * H * W * cv;
  float g[VEC];
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
    const int v = (int)(i % cv); size_t r = i / cv;
    const int w = (int)(r % W); size_t q = r / W; const int h = (int)(q % H); const int b = (int)(q / H);
    const int c0 = v * VEC;
    Vec16<T> yy = ldg16(y + i * VEC), o;
    float sc[VEC], sh[VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sc[j] = coef[c0 + j]; sh[j] = coef[C + c0 + j]; }
    stem_route<T>(dpool, idx, yy, sc, sh, b, h, w, c0, C, Ho, Wo, g);
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, bc[c0 + j] * g[j] + bc[C + c0 + j] * yy.get(j) + bc[2 * C + c0 + j]);
    stg16(dy + i * VEC, o);
  }
}

// ---------------------------------------------------------------------------------------------
// Squeeze-excitation.  One workgroup per sample.
// ---------------------------------------------------------------------------------------------
// se buffers (fp32): pooled[B][C], hidden[B][Cr], scale[B][C]
template <typename T>
__global__ __launch_bounds__(256) void se_pool_fc_kernel(const T* __restrict__ x, const float* __restrict__ w1, const float* __restrict__ w2,
                                                         float* __restrict__ pooled, float* __restrict__ hidden, float* __restrict__ scale,
                                                         int HW, int C, int Cr, const float* __restrict__ part, int chunks) {
  constexpr int VEC = Vec16<T>::N;
  extern __shared__ float sh[];            // [256*VEC] scratch, then pooled[C], hidden[Cr]
  const int b = blockIdx.x, cv = C / VEC, lanes_r = 256 / cv;
  const int myv = threadIdx.x % cv, myr = threadIdx.x / cv;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  if (!part && myr < lanes_r) {
    const T* xb = x + (size_t)b * HW * C + myv * VEC;
    int p = myr;
    for (; p + 3 * lanes_r < HW; p += 4 * lanes_r) {                 // 4 independent 16-byte loads in flight per thread
      Vec16<T> v0 = ldg16(xb + (size_t)p * C), v1 = ldg16(xb + (size_t)(p + lanes_r) * C);
      Vec16<T> v2 = ldg16(xb + (size_t)(p + 2 * lanes_r) * C), v3 = ldg16(xb + (size_t)(p + 3 * lanes_r) * C);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += (v0.get(j) + v1.get(j)) + (v2.get(j) + v3.get(j));
    }
    for (; p < HW; p += lanes_r) {
      Vec16<T> v = ldg16(xb + (size_t)p * C);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += v.get(j);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[threadIdx.x * VEC + j] = s[j];
  __syncthreads();
  float* pl = sh + 256 * VEC; float* hd = pl + C;
  for (int c = threadIdx.x; c < C; c += 256) {
    const int v = c / VEC, j = c - v * VEC;
    float t = 0.f;
    if (part) { for (int k = 0; k < chunks; ++k) t += part[((size_t)b * chunks + k) * C + c]; }    // column sums left by bn_apply_pool_kernel
    else { for (int r = 0; r < lanes_r; ++r) t += sh[(r * cv + v) * VEC + j]; }
    t /= (float)HW;
    pl[c] = t; pooled[(size_t)b * C + c] = t;
  }
  __syncthreads();
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (int jr = wave; jr < Cr; jr += 4) {
    float t = 0.f;
    for (int c = lane; c < C; c += 64) t += w1[(size_t)jr * C + c] * pl[c];
    t = wave_sum(t);
    if (lane == 0) { t = fmaxf(t, 0.f); hd[jr] = t; hidden[(size_t)b * Cr + jr] = t; }
  }
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    float t = 0.f;
    for (int jr = 0; jr < Cr; ++jr) t += w2[(size_t)c * Cr + jr] * hd[jr];
    scale[(size_t)b * C + c] = 1.f / (1.f + expf(-t));
  }
}

// out = x * rowscale[b][c] * pixscale[b][hw]   (either scale may be null)
// A thread owns one channel vector (c0 fixed) and walks pixels with 32-bit indices; b = pix / HW is one multiply-shift
// (mul_hw = ceil(2^40 / HW)).  (The former flat-index form spent more time in 64-bit divisions than in memory traffic.)
template <typename T>
__global__ __launch_bounds__(256) void scale_kernel(const T* __restrict__ x, const float* __restrict__ chscale, const float* __restrict__ pixscale,
                                                    T* __restrict__ out, unsigned npix, int C, unsigned long long mul_hw) {
  constexpr int VEC = Vec16<T>::N;
  const int cv = C / VEC, lanes_r = 256 / cv, c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  for (unsigned pix = blockIdx.x * lanes_r + myr; pix < npix; pix += gridDim.x * lanes_r) {
    const size_t e = (size_t)pix * C + c0;
    Vec16<T> v = ldg16(x + e), o;
    const float ps = pixscale ? pixscale[pix] : 1.f;
    if (chscale) {
      const unsigned b = (unsigned)(((unsigned long long)pix * mul_hw) >> 40);
      const f32x4* cs = reinterpret_cast<const f32x4*>(chscale + (size_t)b * C + c0);
#pragma unroll
      for (int q4 = 0; q4 < VEC / 4; ++q4) {
        const f32x4 c4 = cs[q4];
#pragma unroll
        for (int j = 0; j < 4; ++j) o.set(q4 * 4 + j, v.get(q4 * 4 + j) * ps * c4[j]);
      }
    } else {
#pragma unroll
      for (int j = 0; j < VEC; ++j) o.set(j, v.get(j) * ps);
    }
    stg16(out + e, o);
  }
}

// dhs[jr] = [hidden[jr] > 0] * sum_c z2[c] * w2[c][jr] for one sample, by the whole workgroup (round 4, third session).  w2 is [C][Cr]:
// the first form gave a wave one jr and walked c across its lanes -- 64 different cache lines per load (stride Cr floats), 16 K line
// requests per workgroup at C = 512 / Cr = 32, which is what made the 7 x 7 x 512 stage's SE backward take 80 us for 157 MB.  Here
// consecutive lanes read consecutive jr (thread = (channel group g, jr), G <= 64 groups), the G partials per jr are folded in group
// order through LDS: one writer, fixed order (bit-reproducible).  part: >= NT floats of LDS free at this point; ends with a barrier.
__device__ __forceinline__ void se_fc2_bwd(const float* __restrict__ z2, const float* __restrict__ w2, const float* __restrict__ hidden_b,
                                           float* __restrict__ dhs, float* __restrict__ dh_b, float* __restrict__ part, int C, int Cr, int NT) {
  const int G = min(NT / Cr, 64);
  const int jr = threadIdx.x % Cr, g = threadIdx.x / Cr;
  if (g < G) {
    float t = 0.f;
    for (int c = g; c < C; c += G) t += z2[c] * w2[(size_t)c * Cr + jr];
    part[g * Cr + jr] = t;
  }
  __syncthreads();
  if (threadIdx.x < Cr) {
    float t = 0.f;
    for (int q = 0; q < G; ++q) t += part[q * Cr + threadIdx.x];
    t = hidden_b[threadIdx.x] > 0.f ? t : 0.f;
    dhs[threadIdx.x] = t; dh_b[threadIdx.x] = t;
  }
  __syncthreads();
}

// SE backward, per sample: ds[c] = sum_hw dout*x ; through sigmoid / fc2 / relu / fc1 -> dz2[B][C], dh[B][Cr], dpool[B][C]
template <typename T>
__global__ __launch_bounds__(1024) void se_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ w1,
                                                            const float* __restrict__ w2, const float* __restrict__ hidden,
                                                            const float* __restrict__ scale, float* __restrict__ dz2, float* __restrict__ dh,
                                                            float* __restrict__ dpool, int HW, int C, int Cr) {
  constexpr int VEC = Vec16<T>::N;
  extern __shared__ float sh[];
  const int NT = blockDim.x;                      // 256 ... 1024 threads per sample (vqa_se_bwd picks; both forms use the same count)
  const int b = blockIdx.x, cv = C / VEC, lanes_r = NT / cv;
  const int myv = threadIdx.x % cv, myr = threadIdx.x / cv;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  if (myr < lanes_r) {
    const size_t base = (size_t)b * HW * C + myv * VEC;
    int p = myr;
    for (; p + lanes_r < HW; p += 2 * lanes_r) {                     // 2 pixels = 4 independent 16-byte loads in flight per thread
      const size_t o0 = base + (size_t)p * C, o1 = base + (size_t)(p + lanes_r) * C;
      Vec16<T> d0 = ldg16(dout + o0), v0 = ldg16(x + o0), d1 = ldg16(dout + o1), v1 = ldg16(x + o1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += d0.get(j) * v0.get(j) + d1.get(j) * v1.get(j);
    }
    for (; p < HW; p += lanes_r) {
      const size_t off = base + (size_t)p * C;
      Vec16<T> d = ldg16(dout + off), v = ldg16(x + off);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += d.get(j) * v.get(j);
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[threadIdx.x * VEC + j] = s[j];
  __syncthreads();
  float* z2 = sh + NT * VEC; float* dhs = z2 + C;
  for (int c = threadIdx.x; c < C; c += NT) {
    const int v = c / VEC, j = c - v * VEC;
    float t = 0.f;
    for (int r = 0; r < lanes_r; ++r) t += sh[(r * cv + v) * VEC + j];
    const float sg = scale[(size_t)b * C + c];
    t *= sg * (1.f - sg);
    z2[c] = t; dz2[(size_t)b * C + c] = t;
  }
  __syncthreads();
  se_fc2_bwd(z2, w2, hidden + (size_t)b * Cr, dhs, dh + (size_t)b * Cr, sh, C, Cr, NT);
  for (int c = threadIdx.x; c < C; c += NT) {
    float t = 0.f;
    for (int jr = 0; jr < Cr; ++jr) t += dhs[jr] * w1[(size_t)jr * C + c];
    dpool[(size_t)b * C + c] = t;
  }
}

// dx = dout*scale[b][c] + dpool[b][c]/HW   (same thread / index scheme as scale_kernel)
// xmask != nullptr: dx is additionally multiplied by (xmask > 0).  xmask is the SE input = the post-ReLU output of the stage's
// last residual block, so the block's backward receives its gradient ALREADY masked by its ReLU and never re-reads that
// activation (BatchNorm-backward reduce + apply and the identity-path addend: three reads saved for one here).
// BNRED (bn_y != nullptr): dx is the (masked) gradient entering the last block's bn2, so the BatchNorm-backward column sums
// sum dx | sum dx*xhat(bn_y) over the pixels this workgroup stores go to bn_slab[blockIdx.x][3][C] (the layout
// vqa_bn_bwd_finalize reads) -- the standalone reduce pass over dx and y2 is not run (one read of the stage output saved).
template <typename T, bool BNRED>
__global__ __launch_bounds__(256) void se_bwd_apply_kernel(const T* __restrict__ dout, const float* __restrict__ scale, const float* __restrict__ dpool,
                                                           T* __restrict__ dx, unsigned npix, int HW, int C, unsigned long long mul_hw,
                                                           const T* __restrict__ xmask, const T* __restrict__ bn_y, const float* __restrict__ bn_coef,
                                                           float* __restrict__ bn_slab, unsigned long long* __restrict__ bn_facc) {
  constexpr int VEC = Vec16<T>::N;
  const float inv = 1.f / (float)HW;
  const int cv = C / VEC, lanes_r = 256 / cv, c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  float sg[BNRED ? VEC : 1], sx[BNRED ? VEC : 1], bmean[BNRED ? VEC : 1], binv[BNRED ? VEC : 1];
  if constexpr (BNRED) {
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sg[j] = sx[j] = 0.f; bmean[j] = bn_coef[2 * C + c0 + j]; binv[j] = bn_coef[3 * C + c0 + j]; }
  }
#pragma unroll 2
  for (unsigned pix = blockIdx.x * lanes_r + myr; pix < npix; pix += gridDim.x * lanes_r) {
    const size_t e = (size_t)pix * C + c0;
    const unsigned b = (unsigned)(((unsigned long long)pix * mul_hw) >> 40);
    const f32x4* sp = reinterpret_cast<const f32x4*>(scale + (size_t)b * C + c0);
    const f32x4* dp = reinterpret_cast<const f32x4*>(dpool + (size_t)b * C + c0);
    Vec16<T> d = ldg16(dout + e), o, xm, yy;
    if (xmask) xm = ldg16(xmask + e);
    if constexpr (BNRED) yy = ldg16(bn_y + e);
#pragma unroll
    for (int q4 = 0; q4 < VEC / 4; ++q4) {
      const f32x4 s4 = sp[q4], p4 = dp[q4];
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        float v = fmaf(d.get(q4 * 4 + j), s4[j], p4[j] * inv);     // explicit: both instantiations round identically
        if (xmask && !(xm.get(q4 * 4 + j) > 0.f)) v = 0.f;
        o.set(q4 * 4 + j, v);
      }
    }
    stg16(dx + e, o);
    if constexpr (BNRED) {
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        const float g = o.get(j);                      // the value as stored: what bn_bwd_apply will read back
        sg[j] += g; sx[j] += g * (yy.get(j) - bmean[j]) * binv[j];
      }
    }
  }
  if constexpr (BNRED) {
    __shared__ float shm[2 * 256 * VEC];
#pragma unroll
    for (int j = 0; j < VEC; ++j) { shm[threadIdx.x * VEC + j] = sg[j]; shm[(256 + threadIdx.x) * VEC + j] = sx[j]; }
    __syncthreads();
    for (int o2 = threadIdx.x; o2 < 3 * C; o2 += 256) {
      const int k = o2 / C, c = o2 - k * C, v = c / VEC, j = c - v * VEC;
      float t = 0.f;
      if (k < 2) for (int r = 0; r < lanes_r; ++r) t += shm[(k * 256 + r * cv + v) * VEC + j];
      if (bn_facc) {
        const int R = acc_replicas(C);
        if (k < 2) acc_add_fixed(bn_facc, (size_t)R * 3 * C, ((size_t)(blockIdx.x % R) * 3 + k) * C + c, t);
      }
      else bn_slab[((size_t)blockIdx.x * 3 + k) * C + c] = t;
    }
  }
}

// SE backward in ONE pass structure: the workgroup that reduced sample b (phase 1 of se_bwd_reduce_kernel) applies it right away
// (phase 2 of se_bwd_apply_kernel), so the second read of dout / x of that sample (0.4 MB ... 50 KB each) comes from the caches it
// just filled (L2 / Infinity Cache) instead of from HBM after the whole batch has streamed through twice.  Same arithmetic in the
// same order per element as the two-launch form: dx is bit-identical.  BNRED as in se_bwd_apply_kernel (accumulator mode only).
// NREG > 0 (round 4, third session): a thread visits the SAME (pixel, channel group) elements in both phases, so the first NREG vectors
// of dout it reads in phase 1 stay in its registers (+ one bit per element for x > 0) and phase 2 neither re-reads dout nor x for them:
// 7 -> 5 tensor passes where everything fits (NREG = 4: all of a 7 x 7 x 512 sample, 4 of 7 vectors at 14 x 14 x 256, 4 of 13 / 25 at stages 2 / 1; more kept vectors spill).  Loads are issued unconditionally on
// clamped addresses (a load under a branch makes hipcc drain the queue in front of it), the arithmetic per element and its order are
// the NREG = 0 kernel's: bit-identical dx and sums.
// NLDS > 0: the next NLDS vectors are parked in LDS between the phases (16 bytes per thread and vector behind the scratch region;
// one workgroup per CU is resident anyway at these register counts): 10 of a thread's 13 / 25 vectors at stages 2 / 1.
// TAILM: phase 1 also leaves ONE BYTE of [x > 0] per re-read vector in LDS, so phase 2 re-reads dout only (x was read for its signs).
template <typename T, bool BNRED, int NREG = 0, int NLDS = 0, bool TAILM = false>
__global__ __launch_bounds__(1024) void se_bwd_fused_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ w1,
                                                           const float* __restrict__ w2, const float* __restrict__ hidden,
                                                           const float* __restrict__ scale, float* __restrict__ dz2, float* __restrict__ dh,
                                                           float* __restrict__ dpool, T* __restrict__ dx, int HW, int C, int Cr, int mask_out,
                                                           const T* __restrict__ bn_y, const float* __restrict__ bn_coef,
                                                           unsigned long long* __restrict__ bn_facc) {
  constexpr int VEC = Vec16<T>::N;
  extern __shared__ float sh[];            // [NT*VEC] scratch | z2[C] | dhs[Cr] | dpl[C] | scl[C]
  const int NT = blockDim.x;               // 1024 threads per sample where the shape allows: one workgroup per sample is all the
                                           // parallelism there is (B = 512 samples on 256 CUs), and with 256 threads the two streaming
                                           // passes kept 8 waves per CU in flight: 3.3 TB/s against the 5 TB/s of the BatchNorm passes
  const int b = blockIdx.x, cv = C / VEC, lanes_r = NT / cv;
  const int myv = threadIdx.x % cv, myr = threadIdx.x / cv, c0 = myv * VEC;
  const size_t base = (size_t)b * HW * C + c0;
  float s[VEC];
#pragma unroll
  for (int j = 0; j < VEC; ++j) s[j] = 0.f;
  constexpr int NKEEP = NREG + NLDS;
  Vec16<T> dreg[NREG > 0 ? NREG : 1];
  unsigned mbits[NREG > 0 ? (NREG + 3) / 4 : 1];          // [x > 0] of the vectors kept in registers, 8 bits each
  u32x4* dl = reinterpret_cast<u32x4*>(sh + ((blockDim.x * VEC + 3 * C + Cr + 3) & ~3));      // [NLDS][NT] parked vectors ...
  unsigned char* dm = reinterpret_cast<unsigned char*>(dl + NLDS * blockDim.x);              // ... and their [x > 0] bytes
  unsigned char* dmt = dm + NLDS * blockDim.x;                                               // TAILM: [x > 0] bytes of the re-read vectors
  {
    int p = myr;
    if constexpr (NREG > 0) {
      static_assert(NREG % 2 == 0 && NLDS % 2 == 0 && VEC == 8, "kept vectors: pairs of 8-element vectors");
#pragma unroll
      for (int k = 0; k < (NREG + 3) / 4; ++k) mbits[k] = 0u;
#pragma unroll
      for (int i = 0; i < NREG; i += 2) {
        const int p0 = myr + i * lanes_r, p1 = p0 + lanes_r;
        const size_t o0 = base + (size_t)min(p0, HW - 1) * C, o1 = base + (size_t)min(p1, HW - 1) * C;
        Vec16<T> d0 = ldg16(dout + o0), v0 = ldg16(x + o0), d1 = ldg16(dout + o1), v1 = ldg16(x + o1);
        dreg[i] = d0; dreg[i + 1] = d1;
        unsigned m0 = 0u, m1 = 0u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { m0 |= (v0.get(j) > 0.f ? 1u : 0u) << j; m1 |= (v1.get(j) > 0.f ? 1u : 0u) << j; }
        mbits[i >> 2] |= m0 << ((i & 3) * 8);
        mbits[(i + 1) >> 2] |= m1 << (((i + 1) & 3) * 8);
        if (p1 < HW) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) s[j] += d0.get(j) * v0.get(j) + d1.get(j) * v1.get(j);
        } else if (p0 < HW) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) s[j] += d0.get(j) * v0.get(j);
        }
        // keep the PACKED vectors (4 registers each), not the 8 unpacked floats the products were formed from
        asm volatile("" : "+v"(dreg[i].raw), "+v"(dreg[i + 1].raw));
        if ((i & 2) != 0) __builtin_amdgcn_sched_barrier(0);     // two pairs of loads in flight at a time: hoisting all of them spills
      }
      if constexpr (NLDS > 0) {                                  // a rolled loop: unrolled, the parked vectors crowd the registers again
#pragma unroll 1
        for (int k = 0; k < NLDS; k += 2) {
          const int p0 = myr + (NREG + k) * lanes_r, p1 = p0 + lanes_r;
          const size_t o0 = base + (size_t)min(p0, HW - 1) * C, o1 = base + (size_t)min(p1, HW - 1) * C;
          Vec16<T> d0 = ldg16(dout + o0), v0 = ldg16(x + o0), d1 = ldg16(dout + o1), v1 = ldg16(x + o1);
          unsigned m0 = 0u, m1 = 0u;
#pragma unroll
          for (int j = 0; j < VEC; ++j) { m0 |= (v0.get(j) > 0.f ? 1u : 0u) << j; m1 |= (v1.get(j) > 0.f ? 1u : 0u) << j; }
          dl[k * blockDim.x + threadIdx.x] = d0.raw; dl[(k + 1) * blockDim.x + threadIdx.x] = d1.raw;
          dm[k * blockDim.x + threadIdx.x] = (unsigned char)m0; dm[(k + 1) * blockDim.x + threadIdx.x] = (unsigned char)m1;
          if (p1 < HW) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += d0.get(j) * v0.get(j) + d1.get(j) * v1.get(j);
          } else if (p0 < HW) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) s[j] += d0.get(j) * v0.get(j);
          }
        }
      }
      p = myr + NKEEP * lanes_r;
    }
    int q = 0;
    for (; p + lanes_r < HW; p += 2 * lanes_r) {
      const size_t o0 = base + (size_t)p * C, o1 = base + (size_t)(p + lanes_r) * C;
      Vec16<T> d0 = ldg16(dout + o0), v0 = ldg16(x + o0), d1 = ldg16(dout + o1), v1 = ldg16(x + o1);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += d0.get(j) * v0.get(j) + d1.get(j) * v1.get(j);
      if constexpr (TAILM) {
        unsigned m0 = 0u, m1 = 0u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) { m0 |= (v0.get(j) > 0.f ? 1u : 0u) << j; m1 |= (v1.get(j) > 0.f ? 1u : 0u) << j; }
        dmt[q * blockDim.x + threadIdx.x] = (unsigned char)m0; dmt[(q + 1) * blockDim.x + threadIdx.x] = (unsigned char)m1;
        q += 2;
      }
    }
    for (; p < HW; p += lanes_r) {
      const size_t off = base + (size_t)p * C;
      Vec16<T> d = ldg16(dout + off), v = ldg16(x + off);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s[j] += d.get(j) * v.get(j);
      if constexpr (TAILM) {
        unsigned m0 = 0u;
#pragma unroll
        for (int j = 0; j < VEC; ++j) m0 |= (v.get(j) > 0.f ? 1u : 0u) << j;
        dmt[q * blockDim.x + threadIdx.x] = (unsigned char)m0;
        q += 1;
      }
    }
  }
#pragma unroll
  for (int j = 0; j < VEC; ++j) sh[threadIdx.x * VEC + j] = s[j];
  __syncthreads();
  float* z2 = sh + NT * VEC; float* dhs = z2 + C; float* dpl = dhs + Cr; float* scl = dpl + C;
  for (int c = threadIdx.x; c < C; c += NT) {
    const int v = c / VEC, j = c - v * VEC;
    float t = 0.f;
    for (int r = 0; r < lanes_r; ++r) t += sh[(r * cv + v) * VEC + j];
    const float sg = scale[(size_t)b * C + c];
    scl[c] = sg;
    t *= sg * (1.f - sg);
    z2[c] = t; dz2[(size_t)b * C + c] = t;
  }
  __syncthreads();
  se_fc2_bwd(z2, w2, hidden + (size_t)b * Cr, dhs, dh + (size_t)b * Cr, sh, C, Cr, NT);
  for (int c = threadIdx.x; c < C; c += NT) {
    float t = 0.f;
    for (int jr = 0; jr < Cr; ++jr) t += dhs[jr] * w1[(size_t)jr * C + c];
    dpl[c] = t; dpool[(size_t)b * C + c] = t;
  }
  __syncthreads();
  // ---- phase 2: dx = dout*scale + dpool/HW (* (x > 0)), this sample only
  const float inv = 1.f / (float)HW;
  float sc[VEC], dp[VEC], sg[BNRED ? VEC : 1], sx[BNRED ? VEC : 1], bmean[BNRED ? VEC : 1], binv[BNRED ? VEC : 1];
#pragma unroll
  for (int j = 0; j < VEC; ++j) {
    sc[j] = scl[c0 + j]; dp[j] = dpl[c0 + j] * inv;
    if (BNRED) { sg[j] = sx[j] = 0.f; bmean[j] = bn_coef[2 * C + c0 + j]; binv[j] = bn_coef[3 * C + c0 + j]; }
  }
  if constexpr (NREG > 0) {
#pragma unroll
    for (int i = 0; i < NREG; ++i) {
      const int p = myr + i * lanes_r;
      const size_t e = base + (size_t)min(p, HW - 1) * C;
      Vec16<T> o, yy;
      if constexpr (BNRED) yy = ldg16(bn_y + e);
      const unsigned mb = mbits[i >> 2] >> ((i & 3) * 8);
#pragma unroll
      for (int j = 0; j < VEC; ++j) {
        float v = fmaf(dreg[i].get(j), sc[j], dp[j]);
        if (mask_out && !((mb >> j) & 1u)) v = 0.f;
        o.set(j, v);
      }
      if (p < HW) {
        if constexpr (BNRED) {
#pragma unroll
          for (int j = 0; j < VEC; ++j) { const float g = o.get(j); sg[j] += g; sx[j] += g * (yy.get(j) - bmean[j]) * binv[j]; }
        }
        stg16(dx + e, o);
      }
      if ((i & 1) != 0) __builtin_amdgcn_sched_barrier(0);
    }
    if constexpr (NLDS > 0) {
#pragma unroll 2
      for (int k = 0; k < NLDS; ++k) {
        const int p = myr + (NREG + k) * lanes_r;
        if (p >= HW) break;
        const size_t e = base + (size_t)p * C;
        Vec16<T> o, yy, dk;
        if constexpr (BNRED) yy = ldg16(bn_y + e);
        dk.raw = dl[k * blockDim.x + threadIdx.x];               // (a thread reads back only what it wrote: no barrier needed)
        const unsigned mb = dm[k * blockDim.x + threadIdx.x];
#pragma unroll
        for (int j = 0; j < VEC; ++j) {
          float v = fmaf(dk.get(j), sc[j], dp[j]);
          if (mask_out && !((mb >> j) & 1u)) v = 0.f;
          o.set(j, v);
          if constexpr (BNRED) { const float g = o.get(j); sg[j] += g; sx[j] += g * (yy.get(j) - bmean[j]) * binv[j]; }
        }
        stg16(dx + e, o);
      }
    }
  }
  int qt = 0;
#pragma unroll 2
  for (int p = myr + NKEEP * lanes_r; p < HW; p += lanes_r) {
    const size_t e = base + (size_t)p * C;
    Vec16<T> d = ldg16(dout + e), o, xm, yy;
    unsigned mbt = 0xffu;
    if constexpr (TAILM) { mbt = dmt[qt * blockDim.x + threadIdx.x]; ++qt; }   // (TAILM is only dispatched with mask_out; own bytes: no barrier)
    else if (mask_out) xm = ldg16(x + e);
    if constexpr (BNRED) yy = ldg16(bn_y + e);
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      float v = fmaf(d.get(j), sc[j], dp[j]);
      if (TAILM ? !((mbt >> j) & 1u) : (mask_out && !(xm.get(j) > 0.f))) v = 0.f;
      o.set(j, v);
      if constexpr (BNRED) { const float g = o.get(j); sg[j] += g; sx[j] += g * (yy.get(j) - bmean[j]) * binv[j]; }
    }
    stg16(dx + e, o);
  }
  if constexpr (BNRED) {
    __syncthreads();                         // everyone is done with the phase-1 scratch
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sh[threadIdx.x * VEC + j] = sg[j]; }
    __syncthreads();
    const int R = acc_replicas(C);
    for (int c = threadIdx.x; c < C; c += NT) {
      const int v = c / VEC, j = c - v * VEC;
      float t = 0.f;
      for (int r = 0; r < lanes_r; ++r) t += sh[(r * cv + v) * VEC + j];
      acc_add_fixed(bn_facc, (size_t)R * 3 * C, ((size_t)(blockIdx.x % R) * 3 + 0) * C + c, t);
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < VEC; ++j) { sh[threadIdx.x * VEC + j] = sx[j]; }
    __syncthreads();
    for (int c = threadIdx.x; c < C; c += NT) {
      const int v = c / VEC, j = c - v * VEC;
      float t = 0.f;
      for (int r = 0; r < lanes_r; ++r) t += sh[(r * cv + v) * VEC + j];
      acc_add_fixed(bn_facc, (size_t)R * 3 * C, ((size_t)(blockIdx.x % R) * 3 + 1) * C + c, t);
    }
  }
}

// (Round 4: a SPLIT form -- per-chunk column sums over a (row chunk, sample) grid, then fold + FC backward + apply per workgroup --
// was built to give this pass more parallelism and measured SLOWER in isolation at every stage: 250 / 129 / 77 / 89 us against
// 222 / 110 / 69 / 85 us for this kernel at B = 512 (tools/bench_hbm_kernels.py).  The one-workgroup-per-sample kernel is not short of
// parallelism: it moves 6 tensor passes (r dout x | r dout x y2, w dx) at 5.5 TB/s; the "3.0 TB/s" of the bench line divides the FOUR
// algorithmic passes by its time, i.e. it prices the second read of dout / x as a cache hit, which 512 x 0.8 MB in flight are not.)
// dw2[c][j] += sum_b dz2[b][c]*hidden[b][j] ; dw1[j][c] += sum_b dh[b][j]*pooled[b][c]
// A workgroup owns 8 consecutive (j, c) pairs (c fastest) and splits the batch over 32 thread slices; the slices are folded in LDS
// in slice order, so every weight has ONE writer and a fixed summation order (bit-reproducible, no atomics, no scratch).
// OUTS = 32 (round 4, third session): a wave's load covers two full 128-byte rows of dz2 / pooled instead of eight 32-byte pieces
// (C * Cr = 16 384 outputs at the 512-channel stage: 28 -> ~10 us); same 32 slices, same fold order: the same bits as OUTS = 8.
template <int OUTS>
__global__ __launch_bounds__(OUTS * 32) void se_wgrad_kernel(const float* __restrict__ dz2, const float* __restrict__ hidden, const float* __restrict__ dh,
                                const float* __restrict__ pooled, float* dw1, float* dw2, int B, int C, int Cr) {
  const int pr = threadIdx.x % OUTS, q = threadIdx.x / OUTS;
  const int i = blockIdx.x * OUTS + pr;
  const bool live = i < C * Cr;
  const int j = live ? i / C : 0, c = live ? i - j * C : 0;
  float t2 = 0.f, t1 = 0.f;
  if (live) {
#pragma unroll 4
    for (int b = q; b < B; b += 32) {
      t2 += dz2[(size_t)b * C + c] * hidden[(size_t)b * Cr + j];
      t1 += dh[(size_t)b * Cr + j] * pooled[(size_t)b * C + c];
    }
  }
  __shared__ float sh[2][32][OUTS];
  sh[0][q][pr] = t2; sh[1][q][pr] = t1;
  __syncthreads();
  if (threadIdx.x < 2 * OUTS && blockIdx.x * OUTS + (threadIdx.x % OUTS) < C * Cr) {
    const int which = threadIdx.x / OUTS, pp = threadIdx.x % OUTS;
    const int ii = blockIdx.x * OUTS + pp, jj = ii / C, cc = ii - jj * C;
    float t = 0.f;
#pragma unroll
    for (int s_ = 0; s_ < 32; ++s_) t += sh[which][s_][pp];
    if (which == 0) dw2[(size_t)cc * Cr + jj] += t; else dw1[(size_t)jj * C + cc] += t;
  }
}
static void launch_se_wgrad(const float* dz2, const float* hidden, const float* dh, const float* pooled, float* dw1, float* dw2, int B, int C, int Cr,
                            hipStream_t st) {
  if (C * Cr >= 1024) hipLaunchKernelGGL(se_wgrad_kernel<32>, dim3((C * Cr + 31) / 32), dim3(1024), 0, st, dz2, hidden, dh, pooled, dw1, dw2, B, C, Cr);
  else hipLaunchKernelGGL(se_wgrad_kernel<8>, dim3((C * Cr + 7) / 8), dim3(256), 0, st, dz2, hidden, dh, pooled, dw1, dw2, B, C, Cr);
}

// ---------------------------------------------------------------------------------------------
// Spatial attention.  pooled2[B][H][W][2] fp32 (max, mean), amax[B][H][W] int (argmax channel), amap[B][H][W] fp32
// ---------------------------------------------------------------------------------------------
template <typename T>
__global__ void spatial_pool_kernel(const T* __restrict__ x, float* __restrict__ pooled2, int* __restrict__ amax, size_t npix, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const size_t wid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t p = wid; p < npix; p += nw) {
    float mx = -INFINITY, sm = 0.f; int mi = 0x7fffffff;
    for (int c0 = lane * VEC; c0 < C; c0 += 64 * VEC) {
      Vec16<T> v = ldg16(x + p * C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) { const float f = v.get(j); sm += f; if (f > mx) { mx = f; mi = c0 + j; } }
    }
    sm = wave_sum(sm);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {      // max with lowest-index tie-break (torch.max picks the first maximum)
      const float om = __shfl_xor(mx, o, 64); const int oi = __shfl_xor(mi, o, 64);
      if (om > mx || (om == mx && oi < mi)) { mx = om; mi = oi; }
    }
    if (lane == 0) { pooled2[p * 2] = mx; pooled2[p * 2 + 1] = sm / (float)C; amax[p] = mi; }
  }
}

// amap = sigmoid(conv7x7(pooled2; w[2][7][7]))   (w is the (1,2,7,7) parameter, ch-major)
__global__ void spatial_conv_kernel(const float* __restrict__ pooled2, const float* __restrict__ w, float* __restrict__ amap, int B, int H, int W) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * H * W) return;
  const int ww = (int)(i % W); size_t q = i / W; const int h = (int)(q % H); const int b = (int)(q / H);
  float t = 0.f;
  for (int r = 0; r < 7; ++r) {
    const int ih = h + r - 3; if (ih < 0 || ih >= H) continue;
    for (int s = 0; s < 7; ++s) {
      const int iw = ww + s - 3; if (iw < 0 || iw >= W) continue;
      const float* pp = pooled2 + (((size_t)b * H + ih) * W + iw) * 2;
      t += pp[0] * w[r * 7 + s] + pp[1] * w[49 + r * 7 + s];
    }
  }
  amap[i] = 1.f / (1.f + expf(-t));
}

// dpre[p] = (sum_c dout*x) * a*(1-a)
template <typename T>
__global__ void spatial_bwd_reduce_kernel(const T* __restrict__ dout, const T* __restrict__ x, const float* __restrict__ amap,
                                          float* __restrict__ dpre, size_t npix, int C) {
  constexpr int VEC = Vec16<T>::N;
  const int lane = threadIdx.x & 63;
  const size_t wid = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6, nw = ((size_t)gridDim.x * blockDim.x) >> 6;
  for (size_t p = wid; p < npix; p += nw) {
    float s = 0.f;
    for (int c0 = lane * VEC; c0 < C; c0 += 64 * VEC) {
      Vec16<T> d = ldg16(dout + p * C + c0), v = ldg16(x + p * C + c0);
#pragma unroll
      for (int j = 0; j < VEC; ++j) s += d.get(j) * v.get(j);
    }
    s = wave_sum(s);
    if (lane == 0) { const float a = amap[p]; dpre[p] = s * a * (1.f - a); }
  }
}

// dpool2[b][h][w][ch] = sum_{r,s} dpre[b][h-r+3][w-s+3] * w[ch][r][s]
__global__ void spatial_bwd_conv_kernel(const float* __restrict__ dpre, const float* __restrict__ w, float* __restrict__ dpool2, int B, int H, int W) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * H * W) return;
  const int ww = (int)(i % W); size_t q = i / W; const int h = (int)(q % H); const int b = (int)(q / H);
  float t0 = 0.f, t1 = 0.f;
  for (int r = 0; r < 7; ++r) {
    const int oh = h - r + 3; if (oh < 0 || oh >= H) continue;
    for (int s = 0; s < 7; ++s) {
      const int ow = ww - s + 3; if (ow < 0 || ow >= W) continue;
      const float d = dpre[((size_t)b * H + oh) * W + ow];
      t0 += d * w[r * 7 + s]; t1 += d * w[49 + r * 7 + s];
    }
  }
  dpool2[i * 2] = t0; dpool2[i * 2 + 1] = t1;
}

// dx = dout*amap[p] + dpool2[p][1]/C + (c == amax[p]) * dpool2[p][0]
template <typename T>
__global__ void spatial_bwd_apply_kernel(const T* __restrict__ dout, const float* __restrict__ amap, const float* __restrict__ dpool2,
                                         const int* __restrict__ amax, T* __restrict__ dx, unsigned npix, int C) {
  constexpr int VEC = Vec16<T>::N;
  const float inv = 1.f / (float)C;
  const int cv = C / VEC, lanes_r = 256 / cv, c0 = (threadIdx.x % cv) * VEC, myr = threadIdx.x / cv;
  for (unsigned p = blockIdx.x * lanes_r + myr; p < npix; p += gridDim.x * lanes_r) {
    const size_t e = (size_t)p * C + c0;
    Vec16<T> d = ldg16(dout + e), o;
    const float a = amap[p], dm = dpool2[p * 2], da = dpool2[p * 2 + 1] * inv; const int mi = amax[p];
#pragma unroll
    for (int j = 0; j < VEC; ++j) o.set(j, d.get(j) * a + da + ((c0 + j) == mi ? dm : 0.f));
    stg16(dx + e, o);
  }
}

// part[y][ch][r][s] = sum over slice y of the pixels of dpre[b][h][w] * pooled2[b][h+r-3][w+s-3][ch] ; one block per ((ch,r,s), slice);
// spatial_wgrad_finish_kernel folds the slices in slice order into dw (one writer per weight: bit-reproducible, no atomics).
constexpr int SPATIAL_WG_SLICES = 16;
__global__ __launch_bounds__(256) void spatial_wgrad_kernel(const float* __restrict__ dpre, const float* __restrict__ pooled2, float* __restrict__ part, int B, int H, int W) {
  const int o = blockIdx.x, ch = o / 49, r = (o % 49) / 7, s = o % 7;
  float t = 0.f;
  const size_t n = (size_t)B * H * W;
  for (size_t i = (size_t)blockIdx.y * 256 + threadIdx.x; i < n; i += (size_t)gridDim.y * 256) {
    const int ww = (int)(i % W); size_t q = i / W; const int h = (int)(q % H); const int b = (int)(q / H);
    const int ih = h + r - 3, iw = ww + s - 3;
    if (ih < 0 || ih >= H || iw < 0 || iw >= W) continue;
    t += dpre[i] * pooled2[(((size_t)b * H + ih) * W + iw) * 2 + ch];
  }
  __shared__ float sh[4];
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) part[blockIdx.y * 98 + o] = sh[0] + sh[1] + sh[2] + sh[3];
}
__global__ void spatial_wgrad_finish_kernel(const float* __restrict__ part, float* dw) {
  const int o = threadIdx.x;
  if (o >= 98) return;
  float t = 0.f;
#pragma unroll
  for (int y = 0; y < SPATIAL_WG_SLICES; ++y) t += part[y * 98 + o];
  dw[o] += t;
}

// NHWC(T) <-> NCHW(fp32) boundary conversions (aux['image_features'] is NCHW fp32 at the API boundary)
template <typename T>
__global__ void nhwc_to_nchw_kernel(const T* __restrict__ in, float* __restrict__ out, int B, int HW, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * HW * C) return;
  const int p = (int)(i % HW); size_t q = i / HW; const int c = (int)(q % C); const size_t b = q / C;
  out[i] = to_f<T>(in[(b * HW + p) * C + c]);
}
template <typename T>
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ in, T* __restrict__ out, int B, int HW, int C) {
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)B * HW * C) return;
  const int c = (int)(i % C); size_t q = i / C; const int p = (int)(q % HW); const size_t b = q / HW;
  out[i] = from_f<T>(in[(b * C + c) * HW + p]);
}

// ---------------------------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------------------------
// grid for the pixel-walking elementwise kernels (a thread owns a channel vector): enough workgroups to cover the pixels, capped
static inline int px_grid(size_t npix, int C, int VEC) { const size_t lr = 256 / (C / VEC), g = (npix + lr - 1) / lr; return (int)(g > 16384 ? 16384 : (g ? g : 1)); }
static inline unsigned long long magic40(unsigned d) { return ((1ull << 40) + d - 1) / d; }   // (x * m) >> 40 == x / d for x*d < 2^40
static inline int ew_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g > 16384 ? 16384 : (g ? g : 1)); }
static inline int row_grid(size_t rows, int lanes_r) { size_t g = (rows + lanes_r - 1) / lanes_r; return (int)(g > 8192 ? 8192 : (g ? g : 1)); }
#define DT(call_f, call_b) do { if (dtype) { call_b; } else { call_f; } } while (0)

extern "C" {

// part: igemm slab [tiles][2][C]; scratch: >= 64*2*C doubles; coef out: 4*C floats. running_* / nbt may be null.
int vqa_bn_stats_finalize(const float* part, int tiles, int C, double count, const float* gamma, const float* beta,
                          float* running_mean, float* running_var, long long* nbt, float momentum, float eps,
                          double* scratch, float* coef, hipStream_t st) {
  if (!part || !scratch || !coef || tiles <= 0 || C <= 0) return VQA_EARG;
  int G = (tiles + 63) / 64; if (G > 64) G = 64; if (G < 1) G = 1;
  hipLaunchKernelGGL(bn_reduce_partials_kernel, dim3((C + 63) / 64, G), dim3(256), 0, st, part, scratch, tiles, C);
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((C + 255) / 256), dim3(256), 0, st, scratch, G, C, count, gamma, beta,
                     running_mean, running_var, nbt, momentum, eps, coef);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_bn_eval_coef(int C, const float* gamma, const float* beta, const float* rm, const float* rv, float eps, float* coef, hipStream_t st) {
  hipLaunchKernelGGL(bn_eval_coef_kernel, dim3((C + 255) / 256), dim3(256), 0, st, C, gamma, beta, rm, rv, eps, coef);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_bn_apply(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, void* out, long long numel, int C, int relu, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || numel % C || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  const size_t rows = (size_t)numel / C;
  const int grid = row_grid(rows, 256 / (C / VEC));
#define BN_APPLY(TT, R) hipLaunchKernelGGL((bn_apply_kernel<TT, R>), dim3(grid), dim3(256), 0, st, (const TT*)y, coef, (const TT*)res, rcoef, (TT*)out, rows, C, relu)
  const int mode = !res ? 0 : (rcoef ? 2 : 1);
  if (dtype) { if (mode == 0) BN_APPLY(bf16_t, 0); else if (mode == 1) BN_APPLY(bf16_t, 1); else BN_APPLY(bf16_t, 2); }
  else { if (mode == 0) BN_APPLY(float, 0); else if (mode == 1) BN_APPLY(float, 1); else BN_APPLY(float, 2); }
#undef BN_APPLY
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// rows of one sample a workgroup of vqa_bn_apply_pool owns (14 passes of the workgroup) and the resulting chunks per sample
static inline int pool_rpc(int C, int VEC) { return (256 / (C / VEC)) * 14; }
int vqa_bn_apply_pool_chunks(int dtype, int HW, int C) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || C / VEC > 256 || 256 % (C / VEC) || HW <= 0) return 0;
  return (HW + pool_rpc(C, VEC) - 1) / pool_rpc(C, VEC);
}
// bn_apply of a stage's last block + the SE pooling sums: part [B][vqa_bn_apply_pool_chunks][C] floats (vqa_se_fwd folds them)
int vqa_bn_apply_pool(int dtype, const void* y, const float* coef, const void* res, const float* rcoef, void* out, int B, int HW, int C,
                      int relu, float* part, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  const int chunks = vqa_bn_apply_pool_chunks(dtype, HW, C);
  if (!y || !coef || !out || !part || B <= 0 || chunks <= 0 || B > 65535) return VQA_EARG;
  const int rpc = pool_rpc(C, VEC);
#define BN_APPLYP(TT, R) hipLaunchKernelGGL((bn_apply_pool_kernel<TT, R>), dim3(chunks, B), dim3(256), 0, st, (const TT*)y, coef, (const TT*)res, rcoef, (TT*)out, HW, C, relu, rpc, part)
  const int mode = !res ? 0 : (rcoef ? 2 : 1);
  if (dtype) { if (mode == 0) BN_APPLYP(bf16_t, 0); else if (mode == 1) BN_APPLYP(bf16_t, 1); else BN_APPLYP(bf16_t, 2); }
  else { if (mode == 0) BN_APPLYP(float, 0); else if (mode == 1) BN_APPLYP(float, 1); else BN_APPLYP(float, 2); }
#undef BN_APPLYP
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// Train-mode BatchNorm apply whose statistics are fixed-point accumulators acc[2*C + 1] (filled by the producing conv launched
// with stats_mode = 1): finalize + running-statistics update + apply (+ residual | + BatchNorm(res) from racc, + ReLU) in ONE launch;
// coef_out / rcoef_out [4][C] are published for the backward.  pool_part != NULL: the SE-pooling variant (vqa_bn_apply_pool).
// 64-bit words of a fixed-point accumulator for K sums of C channels (replicas + flag, even): what the caller zeroes and passes
int vqa_bn_acc_words(int K, int C) { const long long w = 2ll * acc_replicas(C) * K * C + 1; return (int)((w + 1) / 2 * 2); }   // hi plane | flag | lo plane (common.h)
int vqa_bn_apply_acc(int dtype, const void* y, const unsigned long long* acc, const float* gamma, const float* beta, float* rm, float* rv,
                     long long* nbt, float* coef_out, const void* res, const unsigned long long* racc, const float* rgamma, const float* rbeta,
                     float* rrm, float* rrv, long long* rnbt, float* rcoef_out, void* out, int B, int HW, int C, int relu, double count,
                     float momentum, float eps, float* pool_part, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (!y || !acc || !gamma || !beta || !coef_out || !out || B <= 0 || HW <= 0 || C % VEC || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  if (racc && (!res || !rgamma || !rbeta || !rcoef_out)) return VQA_EARG;
  if (pool_part && racc) return VQA_EARG;
  BnAcc f = {acc, gamma, beta, rm, rv, nbt, coef_out}, fr = {racc, rgamma, rbeta, rrm, rrv, rnbt, rcoef_out};
  const size_t rows = (size_t)B * HW;
  const int lanes_r = 256 / (C / VEC);
  int g1 = row_grid(rows, lanes_r);
  // every workgroup finalizes all C channels in its prologue (R replicas x 2 planes x 2 sums of 8-byte loads + fp64 per channel): with 2048
  // workgroups the prologues read more bytes than the stage-4 tensor holds.  tools/bn_grid_sweep.py, B = 512: 1024 (plain) / 512 (+ residual)
  // workgroups are 3 .. 18 % faster than 2048 at every stage; 256 loses the latency cover again
  { const int cap = vqa_env_int("VQA_BN_GRID", res ? 512 : 1024); if (g1 > cap) g1 = cap; }
  dim3 grid(g1);
  int rpc = 0;
  const double inv_count = 1.0 / count, unbias = count > 1.0 ? count / (count - 1.0) : 1.0;
  const size_t shm = (size_t)(racc ? 4 : 2) * C * sizeof(float);
  if (pool_part) {
    const int chunks = vqa_bn_apply_pool_chunks(dtype, HW, C);
    if (chunks <= 0 || B > 65535) return VQA_EARG;
    rpc = pool_rpc(C, VEC); grid = dim3(chunks, B);
  }
#define BN_ACC(TT, R, PL) hipLaunchKernelGGL((bn_apply_acc_kernel<TT, R, PL>), grid, dim3(256), shm, st, (const TT*)y, f, (const TT*)res, fr, (TT*)out, rows, HW, C, \
    relu, rpc, inv_count, unbias, momentum, eps, pool_part)
  const int mode = !res ? 0 : (racc ? 2 : 1);
  if (dtype) {
    if (pool_part) { if (mode == 0) BN_ACC(bf16_t, 0, true); else BN_ACC(bf16_t, 1, true); }
    else { if (mode == 0) BN_ACC(bf16_t, 0, false); else if (mode == 1) BN_ACC(bf16_t, 1, false); else BN_ACC(bf16_t, 2, false); }
  } else {
    if (pool_part) { if (mode == 0) BN_ACC(float, 0, true); else BN_ACC(float, 1, true); }
    else { if (mode == 0) BN_ACC(float, 0, false); else if (mode == 1) BN_ACC(float, 1, false); else BN_ACC(float, 2, false); }
  }
#undef BN_ACC
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_bn_bwd_blocks(long long rows) { long long g = (rows + 63) / 64; const int cap = vqa_env_int("VQA_BNR_GRID", rows < 65536 ? 256 : 512); return (int)(g > cap ? cap : (g < 1 ? 1 : g)); }
// slab: [vqa_bn_bwd_blocks(rows)][3][C] floats
// acc_mode = 1: `slab` is a fixed-point accumulator unsigned long long [3*C + 1] (zeroed by the caller) instead of a float slab
int vqa_bn_bwd_reduce(int dtype, const void* dout, const void* outact, const void* y, const float* coef, const void* y2, const float* coef2,
                      float* slab, long long rows, int C, int self_mask, int acc_mode, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  const int nb = vqa_bn_bwd_blocks(rows);
  const size_t shm = (size_t)3 * 256 * VEC * 4;
#define BWD_RED(TT, S, D) hipLaunchKernelGGL((bn_bwd_reduce_kernel<TT, S, D>), dim3(nb), dim3(256), shm, st, (const TT*)dout, (const TT*)outact, \
    (const TT*)y, coef, (const TT*)y2, coef2, acc_mode ? nullptr : slab, (size_t)rows, C, acc_mode ? (unsigned long long*)slab : nullptr)
  if (self_mask && (y2 || outact)) return VQA_EARG;
  if (dtype) { if (self_mask) BWD_RED(bf16_t, true, false); else if (y2) BWD_RED(bf16_t, false, true); else BWD_RED(bf16_t, false, false); }
  else { if (self_mask) BWD_RED(float, true, false); else if (y2) BWD_RED(float, false, true); else BWD_RED(float, false, false); }
#undef BWD_RED
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_bn_bwd_finalize(const float* slab, int nblk, int C, int which, double count, const float* gamma, const float* coef, int training,
                        float* dgamma, float* dbeta, float* bcoef, hipStream_t st) {
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((C + 63) / 64), dim3(1024), 0, st, slab, nblk, C, which, count, gamma, coef, training, dgamma, dbeta, bcoef);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// BatchNorm backward apply with the finalize folded in (training mode): facc[3*C + 1] fixed-point sums from vqa_bn_bwd_reduce /
// vqa_se_bwd in accumulate mode; d gamma / d beta (+=) are written by the first workgroup.  y2 / gamma2 / coef2 / dgamma2 / dbeta2 / dy2:
// the 1x1 shortcut's BatchNorm sharing g (all or none).  self_mask: ReLU mask recomputed from y (coef scale | shift), outact unused.
int vqa_bn_bwd_apply_acc(int dtype, const void* dout, const void* outact, const void* y, const unsigned long long* facc, const float* gamma,
                         const float* coef, float* dgamma, float* dbeta, void* dy, const void* y2, const float* gamma2, const float* coef2,
                         float* dgamma2, float* dbeta2, void* dy2, long long numel, int C, double count, int self_mask, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (!dout || !y || !facc || !gamma || !coef || !dgamma || !dbeta || !dy || C % VEC || numel % C || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  if (y2 && (!gamma2 || !coef2 || !dgamma2 || !dbeta2 || !dy2)) return VQA_EARG;
  if (self_mask && (y2 || outact)) return VQA_EARG;
  const size_t rows = (size_t)numel / C;
  int grid = row_grid(rows, 256 / (C / VEC));
  { const int cap = vqa_env_int("VQA_BNB_GRID", 512); if (grid > cap) grid = cap; }     // (prologue amortisation, as vqa_bn_apply_acc: -4 .. -20 % against 2048)
  const size_t shm = (size_t)(y2 ? 6 : 3) * C * sizeof(float);
  const double inv_count = 1.0 / count;
#define BWD_ACC(TT, S, D) hipLaunchKernelGGL((bn_bwd_apply_acc_kernel<TT, S, D>), dim3(grid), dim3(256), shm, st, (const TT*)dout, (const TT*)outact, \
    (const TT*)y, facc, gamma, coef, dgamma, dbeta, (TT*)dy, (const TT*)y2, gamma2, coef2, dgamma2, dbeta2, (TT*)dy2, rows, C, inv_count)
  if (dtype) { if (self_mask) BWD_ACC(bf16_t, true, false); else if (y2) BWD_ACC(bf16_t, false, true); else BWD_ACC(bf16_t, false, false); }
  else { if (self_mask) BWD_ACC(float, true, false); else if (y2) BWD_ACC(float, false, true); else BWD_ACC(float, false, false); }
#undef BWD_ACC
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_bn_bwd_apply(int dtype, const void* dout, const void* outact, const void* y, const float* bc, void* dy,
                     const void* y2, const float* bc2, void* dy2, long long numel, int C, const float* mask_coef, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || numel % C || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  const size_t rows = (size_t)numel / C;
  const int grid = row_grid(rows, 256 / (C / VEC));
#define BWD_APPLY(TT, S, D) hipLaunchKernelGGL((bn_bwd_apply_kernel<TT, S, D>), dim3(grid), dim3(256), 0, st, (const TT*)dout, (const TT*)outact, \
    (const TT*)y, bc, (TT*)dy, (const TT*)y2, bc2, (TT*)dy2, rows, C, mask_coef)
  if (mask_coef && (y2 || outact)) return VQA_EARG;
  if (dtype) { if (mask_coef) BWD_APPLY(bf16_t, true, false); else if (y2) BWD_APPLY(bf16_t, false, true); else BWD_APPLY(bf16_t, false, false); }
  else { if (mask_coef) BWD_APPLY(float, true, false); else if (y2) BWD_APPLY(float, false, true); else BWD_APPLY(float, false, false); }
#undef BWD_APPLY
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

int vqa_stem_pool_fwd(int dtype, const void* y, const float* coef, void* out, uint8_t* idx, int B, int H, int W, int C, hipStream_t st) {
  const int VEC = dtype ? 8 : 4, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const int per_row = SP_PAIRS * (C / VEC);
  if (C % 8 || per_row > 256) return VQA_EARG;
  const int trows = 256 / per_row < SP_ROWS ? 256 / per_row : SP_ROWS;      // 4 rows (bf16, C = 64), 2 (fp32)
  const int Wp = (Wo + 1) / 2;
  const long long tiles = (long long)B * ((Ho + trows - 1) / trows) * ((Wp + SP_PAIRS - 1) / SP_PAIRS);
  if (tiles <= 0 || tiles > 0x7fffffffll) return VQA_EARG;
  const int threads = (per_row * trows + 63) / 64 * 64;
  DT(hipLaunchKernelGGL(stem_pool_fwd_kernel<float>, dim3((unsigned)tiles), dim3(threads), 0, st, (const float*)y, coef, (float*)out, idx, B, H, W, C, Ho, Wo, trows),
     hipLaunchKernelGGL(stem_pool_fwd_kernel<bf16_t>, dim3((unsigned)tiles), dim3(threads), 0, st, (const bf16_t*)y, coef, (bf16_t*)out, idx, B, H, W, C, Ho, Wo, trows));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_stem_bwd_reduce(int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* coef, float* slab, int B, int H, int W, int C, hipStream_t st) {
  const int VEC = dtype ? 8 : 4, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  if (C % 8 || 256 % (C / VEC)) return VQA_EARG;
  const int nb = vqa_bn_bwd_blocks((long long)B * H * W);
  const size_t shm = (size_t)2 * 256 * VEC * 4;
  DT(hipLaunchKernelGGL(stem_bwd_reduce_kernel<float>, dim3(nb), dim3(256), shm, st, (const float*)dpool, idx, (const float*)y, coef, slab, B, H, W, C, Ho, Wo),
     hipLaunchKernelGGL(stem_bwd_reduce_kernel<bf16_t>, dim3(nb), dim3(256), shm, st, (const bf16_t*)dpool, idx, (const bf16_t*)y, coef, slab, B, H, W, C, Ho, Wo));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_stem_bwd_apply(int dtype, const void* dpool, const uint8_t* idx, const void* y, const float* coef, const float* bc, void* dy, int B, int H, int W, int C, hipStream_t st) {
  const int VEC = dtype ? 8 : 4, Ho = (H + 2 - 3) / 2 + 1, Wo = (W + 2 - 3) / 2 + 1;
  const size_t total = (size_t)B * H * W * (C / VEC);
  DT(hipLaunchKernelGGL(stem_bwd_apply_kernel<float>, dim3(ew_grid(total)), dim3(256), 0, st, (const float*)dpool, idx, (const float*)y, coef, bc, (float*)dy, B, H, W, C, Ho, Wo),
     hipLaunchKernelGGL(stem_bwd_apply_kernel<bf16_t>, dim3(ew_grid(total)), dim3(256), 0, st, (const bf16_t*)dpool, idx, (const bf16_t*)y, coef, bc, (bf16_t*)dy, B, H, W, C, Ho, Wo));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

int vqa_se_fwd(int dtype, const void* x, const float* w1, const float* w2, float* pooled, float* hidden, float* scale, void* out,
               int B, int HW, int C, int Cr, const float* pool_part, int pool_chunks, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  if (pool_part && pool_chunks != vqa_bn_apply_pool_chunks(dtype, HW, C)) return VQA_EARG;
  const size_t shm = ((size_t)256 * VEC + C + Cr) * 4;
  DT(hipLaunchKernelGGL(se_pool_fc_kernel<float>, dim3(B), dim3(256), shm, st, (const float*)x, w1, w2, pooled, hidden, scale, HW, C, Cr, pool_part, pool_chunks),
     hipLaunchKernelGGL(se_pool_fc_kernel<bf16_t>, dim3(B), dim3(256), shm, st, (const bf16_t*)x, w1, w2, pooled, hidden, scale, HW, C, Cr, pool_part, pool_chunks));
  const size_t npix = (size_t)B * HW;
  if (npix >= (1ull << 28)) return VQA_EARG;
  DT(hipLaunchKernelGGL(scale_kernel<float>, dim3(px_grid(npix, C, VEC)), dim3(256), 0, st, (const float*)x, scale, (const float*)nullptr, (float*)out, (unsigned)npix, C, magic40(HW)),
     hipLaunchKernelGGL(scale_kernel<bf16_t>, dim3(px_grid(npix, C, VEC)), dim3(256), 0, st, (const bf16_t*)x, scale, (const float*)nullptr, (bf16_t*)out, (unsigned)npix, C, magic40(HW)));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// rows of the BatchNorm-backward slab vqa_se_bwd fills when bn_slab is given (= workgroups of its apply pass)
int vqa_se_bwd_blocks(int dtype, int B, int HW, int C) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || C / VEC > 256 || 256 % (C / VEC)) return 0;
  const int g = px_grid((size_t)B * HW, C, VEC);
  return g > 2048 ? 2048 : g;          // slab rows vqa_bn_bwd_finalize folds (it is sized for <= ~1k rows: 16384 rows cost it +90 us)
}
// measurement switch (ablation builds only): VQA_SE_NREG=0 -> the re-reading form of the per-sample SE backward
static bool se_bwd_nreg_off() { return vqa_env_int("VQA_SE_NREG", 1) == 0; }
static bool se_bwd_nlds_off() { return vqa_env_int("VQA_SE_NLDS", 1) == 0; }
// scratch floats vqa_se_bwd needs: dz2[B*C] | dh[B*Cr] | dpool[B*C]
long long vqa_se_bwd_scratch(int dtype, int B, int HW, int C, int Cr) { (void)dtype; (void)HW; return (long long)B * (2 * C + Cr); }
// scratch: vqa_se_bwd_scratch floats
// bn_y / bn_coef / bn_slab (all or none): dx is the gradient entering the BatchNorm whose conv output is bn_y (the last block's bn2,
// mask_out = 1): its backward column sums go to bn_slab[vqa_se_bwd_blocks][3][C] and vqa_bn_bwd_reduce is skipped by the caller.
// bn_acc_mode: 1: bn_slab is a fixed-point accumulator (vqa_bn_acc_words(3, C))
int vqa_se_bwd(int dtype, const void* dout, const void* x, const float* w1, const float* w2, const float* pooled, const float* hidden,
               const float* scale, float* scratch, void* dx, float* dw1, float* dw2, int B, int HW, int C, int Cr, int mask_out,
               const void* bn_y, const float* bn_coef, float* bn_slab, int bn_acc_mode, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  if ((bn_slab != nullptr) != (bn_y != nullptr) || (bn_slab != nullptr) != (bn_coef != nullptr)) return VQA_EARG;
  float* dz2 = scratch; float* dh = dz2 + (size_t)B * C; float* dpool = dh + (size_t)B * Cr;
  const int cvh = C / VEC;
  const int nt = (1024 % cvh == 0 && (long long)HW * cvh >= (dtype ? 2048 : 4096)) ? 1024 : 256;      // threads per sample (tiny maps: 256 are plenty; bf16 7 x 7 x 512: 1024, so that a thread keeps all its 4 vectors)
  const size_t shm = ((size_t)nt * VEC + C + Cr) * 4;
  const bool accm = (bn_acc_mode & 1) != 0;
  if (bn_slab && accm && B > VQA_ACC_MAX_PARTS) return VQA_EARG;     // one partial per sample (common.h: the fixed-point total must not wrap)
  if (!bn_slab || accm) {                    // one pass structure: reduce + apply per sample in the same workgroup
    const size_t shm2 = ((size_t)nt * VEC + 3 * C + Cr) * 4;
#define SE_FUSED(TT, R) hipLaunchKernelGGL((se_bwd_fused_kernel<TT, R>), dim3(B), dim3(nt), shm2, st, (const TT*)dout, (const TT*)x, w1, w2, hidden, scale, \
    dz2, dh, dpool, (TT*)dx, HW, C, Cr, mask_out, (const TT*)bn_y, bn_coef, (unsigned long long*)bn_slab)
#define SE_FUSED_K(R_, NR, NL) do { if (tailm) SE_FUSED_KT(R_, NR, NL, true); else SE_FUSED_KT(R_, NR, NL, false); } while (0)
#define SE_FUSED_KT(R_, NR, NL, TM) do { \
      const size_t shl = ((shm2 + 15) & ~(size_t)15) + (size_t)(NL) * nt * 17 + ((TM) ? (size_t)ntail * nt : 0); \
      auto kfn = se_bwd_fused_kernel<bf16_t, R_, NR, NL, TM>; \
      if (shl > 65536) (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shl); \
      hipLaunchKernelGGL(kfn, dim3(B), dim3(nt), shl, st, (const bf16_t*)dout, (const bf16_t*)x, w1, w2, hidden, scale, dz2, dh, dpool, (bf16_t*)dx, \
                         HW, C, Cr, mask_out, (const bf16_t*)bn_y, bn_coef, (unsigned long long*)bn_slab); } while (0)
#define SE_FUSED_D(R_) do { if (nreg == 2) SE_FUSED_K(R_, 2, 0); else if (nlds == 0) SE_FUSED_K(R_, 4, 0); else if (nlds == 2) SE_FUSED_K(R_, 4, 2); \
      else if (nlds == 4) SE_FUSED_K(R_, 4, 4); else SE_FUSED_K(R_, 4, 6); } while (0)
    const int nit = (HW + nt / cvh - 1) / (nt / cvh);          // vectors a thread visits per phase
    const int nreg = !dtype || se_bwd_nreg_off() ? 0 : nit <= 2 ? 2 : 4;       // (8 / 12 kept vectors spill at the 128 registers a 1024-thread workgroup leaves a wave)
    // vectors parked in LDS on top (1024-thread form only: 16 KB each next to the 34 KB of scratch; 6 fill a CU's 160 KB)
    const int nlds = (nreg == 4 && nt == 1024 && !se_bwd_nlds_off()) ? (nit <= 4 ? 0 : nit <= 6 ? 2 : nit <= 8 ? 4 : 6) : 0;
    // one byte of [x > 0] per re-read vector and thread in LDS instead of the second read of x, where it fits beside the rest
    const int ntail = nit - nreg - nlds > 0 ? nit - nreg - nlds : 0;
    const bool tailm = nreg == 4 && mask_out && ntail > 0 && !se_bwd_nlds_off() &&
                       ((shm2 + 15) & ~(size_t)15) + (size_t)nlds * nt * 17 + (size_t)ntail * nt <= 156 * 1024;
    if (dtype && nreg) { if (bn_slab) SE_FUSED_D(true); else SE_FUSED_D(false); }
    else if (dtype) { if (bn_slab) SE_FUSED(bf16_t, true); else SE_FUSED(bf16_t, false); }
    else { if (bn_slab) SE_FUSED(float, true); else SE_FUSED(float, false); }
#undef SE_FUSED_K
#undef SE_FUSED_KT
#undef SE_FUSED_D
#undef SE_FUSED
    launch_se_wgrad(dz2, hidden, dh, pooled, dw1, dw2, B, C, Cr, st);
    VQA_LAUNCH_CHECK(); return VQA_OK;
  }
  DT(hipLaunchKernelGGL(se_bwd_reduce_kernel<float>, dim3(B), dim3(nt), shm, st, (const float*)dout, (const float*)x, w1, w2, hidden, scale, dz2, dh, dpool, HW, C, Cr),
     hipLaunchKernelGGL(se_bwd_reduce_kernel<bf16_t>, dim3(B), dim3(nt), shm, st, (const bf16_t*)dout, (const bf16_t*)x, w1, w2, hidden, scale, dz2, dh, dpool, HW, C, Cr));
  const size_t npix = (size_t)B * HW;
  if (npix >= (1ull << 28)) return VQA_EARG;
  const int ag = bn_slab ? vqa_se_bwd_blocks(dtype, B, HW, C) : px_grid(npix, C, VEC);
#define SE_APPLY(TT, R) hipLaunchKernelGGL((se_bwd_apply_kernel<TT, R>), dim3(ag), dim3(256), 0, st, (const TT*)dout, scale, dpool, (TT*)dx, (unsigned)npix, HW, C, \
    magic40(HW), mask_out ? (const TT*)x : nullptr, (const TT*)bn_y, bn_coef, accm ? nullptr : bn_slab, \
    accm ? (unsigned long long*)bn_slab : nullptr)
  if (dtype) { if (bn_slab) SE_APPLY(bf16_t, true); else SE_APPLY(bf16_t, false); }
  else { if (bn_slab) SE_APPLY(float, true); else SE_APPLY(float, false); }
#undef SE_APPLY
  launch_se_wgrad(dz2, hidden, dh, pooled, dw1, dw2, B, C, Cr, st);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

int vqa_spatial_fwd(int dtype, const void* x, const float* w, float* pooled2, int* amax, float* amap, void* out, int B, int H, int W, int C, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC) return VQA_EARG;
  const size_t npix = (size_t)B * H * W;
  const int pg = (int)((npix + 3) / 4 > 8192 ? 8192 : (npix + 3) / 4);
  DT(hipLaunchKernelGGL(spatial_pool_kernel<float>, dim3(pg), dim3(256), 0, st, (const float*)x, pooled2, amax, npix, C),
     hipLaunchKernelGGL(spatial_pool_kernel<bf16_t>, dim3(pg), dim3(256), 0, st, (const bf16_t*)x, pooled2, amax, npix, C));
  hipLaunchKernelGGL(spatial_conv_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, pooled2, w, amap, B, H, W);
  if (npix >= (1ull << 28) || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  DT(hipLaunchKernelGGL(scale_kernel<float>, dim3(px_grid(npix, C, VEC)), dim3(256), 0, st, (const float*)x, (const float*)nullptr, amap, (float*)out, (unsigned)npix, C, magic40(H * W)),
     hipLaunchKernelGGL(scale_kernel<bf16_t>, dim3(px_grid(npix, C, VEC)), dim3(256), 0, st, (const bf16_t*)x, (const float*)nullptr, amap, (bf16_t*)out, (unsigned)npix, C, magic40(H * W)));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// scratch: dpre[B*H*W] | dpool2[B*H*W*2] floats
// floats of the `scratch` argument of vqa_spatial_bwd: dpre [npix] + dpool2 [npix][2] + the conv-weight partial sums [16][98]
long long vqa_spatial_bwd_scratch(int B, int H, int W) { return 3ll * B * H * W + SPATIAL_WG_SLICES * 98; }
int vqa_spatial_bwd(int dtype, const void* dout, const void* x, const float* w, const float* pooled2, const int* amax, const float* amap,
                    float* scratch, void* dx, float* dw, int B, int H, int W, int C, hipStream_t st) {
  const int VEC = dtype ? 8 : 4;
  if (C % VEC) return VQA_EARG;
  const size_t npix = (size_t)B * H * W;
  float* dpre = scratch; float* dpool2 = dpre + npix; float* wpart = dpool2 + 2 * npix;
  const int pg = (int)((npix + 3) / 4 > 8192 ? 8192 : (npix + 3) / 4);
  DT(hipLaunchKernelGGL(spatial_bwd_reduce_kernel<float>, dim3(pg), dim3(256), 0, st, (const float*)dout, (const float*)x, amap, dpre, npix, C),
     hipLaunchKernelGGL(spatial_bwd_reduce_kernel<bf16_t>, dim3(pg), dim3(256), 0, st, (const bf16_t*)dout, (const bf16_t*)x, amap, dpre, npix, C));
  hipLaunchKernelGGL(spatial_bwd_conv_kernel, dim3((unsigned)((npix + 255) / 256)), dim3(256), 0, st, dpre, w, dpool2, B, H, W);
  if (npix >= (1ull << 28) || C / VEC > 256 || 256 % (C / VEC)) return VQA_EARG;
  DT(hipLaunchKernelGGL(spatial_bwd_apply_kernel<float>, dim3(px_grid(npix, C, VEC)), dim3(256), 0, st, (const float*)dout, amap, dpool2, amax, (float*)dx, (unsigned)npix, C),
     hipLaunchKernelGGL(spatial_bwd_apply_kernel<bf16_t>, dim3(px_grid(npix, C, VEC)), dim3(256), 0, st, (const bf16_t*)dout, amap, dpool2, amax, (bf16_t*)dx, (unsigned)npix, C));
  hipLaunchKernelGGL(spatial_wgrad_kernel, dim3(98, SPATIAL_WG_SLICES), dim3(256), 0, st, dpre, pooled2, wpart, B, H, W);
  hipLaunchKernelGGL(spatial_wgrad_finish_kernel, dim3(1), dim3(128), 0, st, wpart, dw);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

int vqa_nhwc_to_nchw(int dtype, const void* in, float* out, int B, int HW, int C, hipStream_t st) {
  const size_t n = (size_t)B * HW * C;
  DT(hipLaunchKernelGGL(nhwc_to_nchw_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const float*)in, out, B, HW, C),
     hipLaunchKernelGGL(nhwc_to_nchw_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const bf16_t*)in, out, B, HW, C));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_nchw_to_nhwc(int dtype, const float* in, void* out, int B, int HW, int C, hipStream_t st) {
  const size_t n = (size_t)B * HW * C;
  DT(hipLaunchKernelGGL(nchw_to_nhwc_kernel<float>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, (float*)out, B, HW, C),
     hipLaunchKernelGGL(nchw_to_nhwc_kernel<bf16_t>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, in, (bf16_t*)out, B, HW, C));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

}  // extern "C"
