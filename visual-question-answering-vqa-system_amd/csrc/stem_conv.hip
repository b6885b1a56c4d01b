// Stem convolution 7x7 / stride 2 / pad 3, 3 -> 64 channels, bf16 MFMA (reference models/cnn_backbone.py:349-350).
//
// One workgroup = one image, RB = 4 consecutive output rows.  The 13 x (2*Wo+8) x 3 input patch is read ONCE from the
// NCHW fp32 image (coalesced along W), converted to bf16 and kept in LDS; the whole 64 x 192 weight matrix sits in
// LDS too and each wave keeps its 24 B-fragments in registers for all of its M tiles.  The implicit-GEMM K index is
// (c, r, s8) with s padded 7 -> 8, so the 8 contraction values of one MFMA operand are 8 CONSECUTIVE input pixels of
// one (channel, row): 4 aligned ds_read_b32 straight out of the patch, no im2col in memory.
//   out  : NHWC bf16 [B][Ho][Wo][64]   (raw conv output; BN+ReLU+MaxPool is fused in vqa_stem_pool_fwd)
//   stats: per-workgroup column sums / sums of squares [gridDim][2][64] for train-mode BatchNorm
#include <cstdlib>
#include "common.h"
#include <type_traits>
#include "stem_route.h"

namespace {
constexpr int RB = 4;          // output rows per workgroup
constexpr int PR = 2 * RB + 5; // patch rows
constexpr int KP = 192;        // padded K = 24 (c,r) pairs x 8
constexpr int LDW = KP + 8;    // weight row stride in LDS (elements)
constexpr int LDCS = 64 + 8;   // C staging row stride
}

// patch[3][PROWS][PW] (bf16, x = iw + XO) <- img[b][c][ih_base + pr][iw], zero outside the image.  The NCHW fp32 image is read with
// 16-byte loads, CH of them in flight per thread: with one 4-byte load per (row, column), three dependent batches of 23, the fill --
// not the MFMA loop, not HBM -- bounded the one-launch inference stem (380 of its 705 us at 1.2 TB/s; 386 us with this fill,
// tools/bench_stem_eval.py).  The training kernels keep their scalar fills: stem_conv_kernel issues all 39 loads of a thread in ONE
// batch and measured the same either way (674 vs 680 us with the pooling pass, step 12.94 vs 12.89 ms), and inlined into the fused
// weight-gradient kernel this function costs it a workgroup per CU (171 vs 118 registers: 709 -> 940 us).
// W % 4 == 0 (other widths take the scalar loop).  XO is odd, so the four bf16 of an item sit at an odd
// element: one 4-byte store between two 2-byte ones.
template <int PROWS, int XO, int CH>
__device__ __forceinline__ void stem_fill_patch(bf16_t* __restrict__ patch, int PW, __amdgpu_buffer_rsrc_t rsImg, int b, int ih_base,
                                                int H, int W, int tid) {
  static_assert(XO & 1, "odd column offset");
  const int W4 = W >> 2, nitems = 3 * PROWS * W4, nh = PW - W;
  const float inv_w4 = 1.0f / (float)W4, inv_nh = 1.0f / (float)nh;
  for (int i = tid; i < 3 * PROWS * nh; i += 256) {                  // halo columns [0, XO) and [W + XO, PW)
    const int row = (int)(((float)i + 0.5f) * inv_nh), k = i - row * nh;
    patch[row * PW + (k < XO ? k : W + k)] = 0;
  }
  for (int base = tid; base < nitems; base += 256 * CH) {
    f32x4 v[CH]; int dst[CH];
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      const int id = base + 256 * i;
      dst[i] = -1;
      v[i] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (id < nitems) {
        const int row = (int)(((float)id + 0.5f) * inv_w4), j = id - row * W4;
        const int c = row / PROWS, pr = row - c * PROWS, ih = ih_base + pr;
        dst[i] = row * PW + 4 * j + XO;
        if (ih >= 0 && ih < H)
          v[i] = __builtin_bit_cast(f32x4, __builtin_amdgcn_raw_buffer_load_b128(rsImg, (((b * 3 + c) * H + ih) * W + 4 * j) * 4, 0, 0));
      }
    }
#pragma unroll
    for (int i = 0; i < CH; ++i) {
      if (dst[i] < 0) continue;
      bf16_t* d = patch + dst[i];
      d[0] = f2bf(v[i][0]);
      *reinterpret_cast<uint32_t*>(d + 1) = (uint32_t)f2bf(v[i][1]) | ((uint32_t)f2bf(v[i][2]) << 16);
      d[3] = f2bf(v[i][3]);
    }
  }
}

// wstem[n][(c*7+r)*8+s] = w[n][r][s][c] (KRSC fp32 master), zero for s == 7 and pairs 21..23
__global__ void stem_pack_kernel(const float* __restrict__ w, bf16_t* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= 64 * KP) return;
  const int n = i / KP, k = i - n * KP, pair = k >> 3, s = k & 7;
  float v = 0.f;
  if (pair < 21 && s < 7) { const int c = pair / 7, r = pair - c * 7; v = w[((n * 7 + r) * 7 + s) * 3 + c]; }
  out[i] = f2bf(v);
}

__global__ __launch_bounds__(256) void stem_conv_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wst, bf16_t* __restrict__ out,
                                                        float* __restrict__ stats, int H, int W, int Ho, int Wo) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PW = 2 * Wo + 8;                       // patch width (x = iw + 3), multiple of 8
  bf16_t* patch = reinterpret_cast<bf16_t*>(smem);                 // [3][PR][PW]
  bf16_t* Wl = patch + 3 * PR * PW;                                 // [64][LDW]
  bf16_t* Cst = Wl;                                                 // [4 waves][16][LDCS], aliases Wl once the B fragments are in registers
  float* red = reinterpret_cast<float*>(Wl + 64 * LDW);             // [4][64][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int rblocks = Ho / RB;
  const int b = blockIdx.x / rblocks, oh0 = (blockIdx.x - b * rblocks) * RB;

  // ---- stage weights (16-byte vectors) and the input patch (fp32 -> bf16)
  for (int v = tid; v < 64 * (KP / 8); v += 256) {
    const int n = v / (KP / 8), kv = v - n * (KP / 8);
    *reinterpret_cast<u32x4*>(&Wl[n * LDW + kv * 8]) = *reinterpret_cast<const u32x4*>(&wst[n * KP + kv * 8]);
  }
  const int ih_base = 2 * oh0 - 3;
  const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, (int)gridDim.x / rblocks * 3 * H * W * 4, 0x00020000);   // 32-bit offsets
  for (int x = tid; x < PW; x += 256) {                    // lanes walk x (coalesced); all 3*PR row loads are issued
    const int iw = x - 3;                                    // back to back before the first use (latency overlapped)
    const bool cok = iw >= 0 && iw < W;
    float vals[3 * PR];
#pragma unroll
    for (int row = 0; row < 3 * PR; ++row) {
      const int c = row / PR, pr = row - c * PR, ih = ih_base + pr;
      const bool ok = cok && ih >= 0 && ih < H;
      const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);      // always-valid address: unconditional load
      const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsImg, (((b * 3 + c) * H + ihc) * W + iwc) * 4, 0, 0));
      vals[row] = ok ? t : 0.f;
    }
#pragma unroll
    for (int row = 0; row < 3 * PR; ++row) patch[row * PW + x] = f2bf(vals[row]);
  }
  __syncthreads();

  // ---- B fragments: 6 k-steps x 4 column tiles, resident in registers
  bf16x8 bfr[6][4];
#pragma unroll
  for (int kk = 0; kk < 6; ++kk)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bfr[kk][nt] = *reinterpret_cast<const bf16x8*>(&Wl[(nt * 16 + li) * LDW + kk * 32 + g * 8]);
  __syncthreads();                                    // every wave has its B fragments: Wl may now be reused as C staging

  float ssum[4] = {0.f, 0.f, 0.f, 0.f}, ssq[4] = {0.f, 0.f, 0.f, 0.f};
  const int mtiles_row = Wo / 16, mtiles = RB * mtiles_row;
  bf16_t* mycs = Cst + wave * 16 * LDCS;
  for (int mt = wave; mt < mtiles; mt += 4) {
    const int orow = mt / mtiles_row, ow0 = (mt - orow * mtiles_row) * 16;
    f32x4 acc[4];
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kk = 0; kk < 6; ++kk) {
      int pair = kk * 4 + g;
      pair = pair > 20 ? 20 : pair;                     // pairs 21..23 have zero weights; read any valid patch row
      const int c = pair / 7, r = pair - c * 7;
      const uint32_t* ap = reinterpret_cast<const uint32_t*>(&patch[(c * PR + 2 * orow + r) * PW + 2 * (ow0 + li)]);
      u32x4 raw = {ap[0], ap[1], ap[2], ap[3]};
      const bf16x8 af = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[kk][nt], acc[nt], 0, 0, 0);
    }
    // epilogue: BN partial sums from the fp32 accumulators, bf16 tile through LDS, 16-byte row stores
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float v = acc[nt][r];
        ssum[nt] += v; ssq[nt] += v * v;
        mycs[(g * 4 + r) * LDCS + nt * 16 + li] = f2bf(v);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // wave-private staging buffer: writes done, and a compiler fence
    const size_t obase = (((size_t)b * Ho + oh0 + orow) * Wo + ow0) * 64;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
      const int v = lane + 64 * i, px = v >> 3, cv = v & 7;
      *reinterpret_cast<u32x4*>(&out[obase + (size_t)px * 64 + cv * 8]) = *reinterpret_cast<const u32x4*>(&mycs[px * LDCS + cv * 8]);
    }
    asm volatile("" ::: "memory");
  }
  if (stats) {
#pragma unroll
    for (int nt = 0; nt < 4; ++nt) {
      float s = ssum[nt], q = ssq[nt];
      s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) { red[(wave * 64 + nt * 16 + lane) * 2] = s; red[(wave * 64 + nt * 16 + lane) * 2 + 1] = q; }
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += red[(w * 64 + tid) * 2]; q += red[(w * 64 + tid) * 2 + 1]; }
      stats[((size_t)blockIdx.x * 2) * 64 + tid] = s;
      stats[((size_t)blockIdx.x * 2 + 1) * 64 + tid] = q;
    }
  }
}


// ------------------------------------------------------------------------------------------------
// Inference stem: conv7x7/2 + BatchNorm (running statistics: scale / shift per channel) + ReLU + MaxPool3x3/2 p1 in ONE kernel
// (models/cnn_backbone.py:349-354 in eval mode).  The 112 x 112 x 64 conv output -- 822 MB written and read back at B = 512 by the
// two-kernel path -- never exists: a wave computes the conv tile of 16 columns for the THREE conv rows of a pooled row (24 MFMAs
// each, operands as in stem_conv_kernel), applies scale / shift / ReLU to the fp32 accumulators and pools in registers.
//   columns: tile t covers conv columns 14t-1 .. 14t+14, i.e. the windows of pooled columns 7t .. 7t+6.  In the accumulator layout
//            a lane holds 4 consecutive columns p = 4g .. 4g+3 of one channel: pooled column 2g is the max of its p, p+1, p+2 and
//            pooled column 2g+1 needs column 4g+4 from the lane 16 further on (one cross-lane move per channel tile).
//   rows:    pooled row k of the block uses conv rows 2k, 2k+1, 2k+2 of the block's 2*PRB+1; the shared row is carried.
//   padding: conv positions outside the image (column -1, row -1, columns >= Wo of a ragged last tile) are zeroed AFTER the ReLU,
//            which is the identity of a max over non-negative values (MaxPool pads with -inf; every window holds a real position).
// Redundant MFMA work (16 columns per 14, 9 rows per 8): 1.29x of a kernel that is HBM-bound by 4x.  No argmax: inference only.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int PRB = 4;               // pooled rows per workgroup
constexpr int NCR = 2 * PRB + 1;     // conv rows per workgroup
constexpr int PRP = 2 * NCR + 5;     // patch rows (23)
}
__device__ __forceinline__ float pool_max(float a, float b) { return (b > a || b != b) ? b : a; }     // NaN propagates (ATen max_pool2d)

__global__ __launch_bounds__(256, 2) void stem_conv_pool_kernel(const float* __restrict__ img, const bf16_t* __restrict__ wst,
                                                                const float* __restrict__ coef, bf16_t* __restrict__ out,
                                                                int B, int H, int W, int Ho, int Wo, int Hp, int Wp, int ctiles, int PW, int dbg) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  bf16_t* patch = reinterpret_cast<bf16_t*>(smem);                 // [3][PRP][PW], x = iw + 5
  bf16_t* Wl = patch + 3 * PRP * PW;                                // [64][LDW]
  bf16_t* Cst = Wl;                                                 // [4 waves][8][LDCS] once the B fragments are in registers
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int pblocks = Hp / PRB;
  const int b = blockIdx.x / pblocks, ph0 = (blockIdx.x - b * pblocks) * PRB;

  for (int v = tid; v < 64 * (KP / 8); v += 256) {
    const int n = v / (KP / 8), kv = v - n * (KP / 8);
    *reinterpret_cast<u32x4*>(&Wl[n * LDW + kv * 8]) = *reinterpret_cast<const u32x4*>(&wst[n * KP + kv * 8]);
  }
  const int ih_base = 4 * ph0 - 5;                                   // input row of patch row 0: 2 * (2 * ph0 - 1) - 3
  const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, B * 3 * H * W * 4, 0x00020000);
  if (dbg & 8) {}
  else if (!(W & 3)) stem_fill_patch<PRP, 5, 8>(patch, PW, rsImg, b, ih_base, H, W, tid);
  else
  for (int x = tid; x < PW; x += 256) {
    const int iw = x - 5;
    const bool cok = iw >= 0 && iw < W;
    const int iwc = min(max(iw, 0), W - 1);
#pragma unroll
    for (int c = 0; c < 3; ++c) {                                    // one channel's 23 rows in flight at a time
      float vals[PRP];
#pragma unroll
      for (int pr = 0; pr < PRP; ++pr) {
        const int ih = ih_base + pr;
        const int ihc = min(max(ih, 0), H - 1);
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsImg, (((b * 3 + c) * H + ihc) * W + iwc) * 4, 0, 0));
        vals[pr] = (cok && ih >= 0 && ih < H) ? t : 0.f;
      }
#pragma unroll
      for (int pr = 0; pr < PRP; ++pr) patch[(c * PRP + pr) * PW + x] = f2bf(vals[pr]);
    }
  }
  __syncthreads();
  bf16x8 bfr[6][4];
#pragma unroll
  for (int kk = 0; kk < 6; ++kk)
#pragma unroll
    for (int nt = 0; nt < 4; ++nt)
      bfr[kk][nt] = *reinterpret_cast<const bf16x8*>(&Wl[(nt * 16 + li) * LDW + kk * 32 + g * 8]);
  float sc[4], sh[4];
#pragma unroll
  for (int nt = 0; nt < 4; ++nt) { sc[nt] = coef[nt * 16 + li]; sh[nt] = coef[64 + nt * 16 + li]; }
  __syncthreads();                                                   // Wl is dead: C staging may overwrite it
  if (dbg & 4) { if (tid == 0) out[(size_t)blockIdx.x * 64] = f2bf(sc[0] + (float)bfr[0][0][0]); return; }

  bf16_t* mycs = Cst + wave * 8 * LDCS;
  for (int t = wave; t < ctiles; t += 4) {
    const int c0 = 14 * t - 1;                                        // conv column of accumulator row p = 0
    // conv row i of the block (conv row 2*ph0 - 1 + i) -> BN + ReLU -> horizontal 3-max at stride 2: h[nt][0] = pooled column 2g,
    // h[nt][1] = pooled column 2g + 1 (garbage for g == 3: never stored)
    auto row = [&](int i, float (&h)[4][2]) {
      f32x4 acc[4];
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) acc[nt] = f32x4{0.f, 0.f, 0.f, 0.f};
      if (!(dbg & 1))
#pragma unroll
      for (int kk = 0; kk < 6; ++kk) {
        int pair = kk * 4 + g;
        pair = pair > 20 ? 20 : pair;
        const int c = pair / 7, r = pair - c * 7;
        const uint32_t* ap = reinterpret_cast<const uint32_t*>(&patch[(c * PRP + 2 * i + r) * PW + 28 * t + 2 * li]);
        u32x4 raw = {ap[0], ap[1], ap[2], ap[3]};
        const bf16x8 af = __builtin_bit_cast(bf16x8, raw);
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) acc[nt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[kk][nt], acc[nt], 0, 0, 0);
      }
      if (dbg & 2) {
#pragma unroll
        for (int nt = 0; nt < 4; ++nt) { h[nt][0] = acc[nt][0] + acc[nt][1]; h[nt][1] = acc[nt][2] + acc[nt][3]; }
        return;
      }
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) {
        float v[4];
#pragma unroll
        for (int r4 = 0; r4 < 4; ++r4) {
          const int col = c0 + 4 * g + r4;
          const float a = acc[nt][r4] * sc[nt] + sh[nt];
          v[r4] = ((unsigned)col < (unsigned)Wo) ? (a < 0.f ? 0.f : a) : 0.f;
        }
        const float nx = __shfl_down(v[0], 16, 64);                   // column 4g + 4
        h[nt][0] = pool_max(pool_max(v[0], v[1]), v[2]);
        h[nt][1] = pool_max(pool_max(v[2], v[3]), nx);
      }
    };
    float hp[4][2], h1[4][2], h2[4][2];
    if (ph0 > 0) row(0, hp);                                          // conv row 2*ph0 - 1 (row -1 of the image: outside)
    else {
#pragma unroll
      for (int nt = 0; nt < 4; ++nt) { hp[nt][0] = 0.f; hp[nt][1] = 0.f; }
    }
#pragma unroll 1
    for (int k = 0; k < PRB; ++k) {
      row(2 * k + 1, h1);
      row(2 * k + 2, h2);
      // this lane's two pooled columns x four channel tiles -> wave-private staging -> 16-byte row stores
#pragma unroll
      for (int nt = 0; nt < 4; ++nt)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const float o = pool_max(pool_max(hp[nt][q], h1[nt][q]), h2[nt][q]);
          mycs[(2 * g + q) * LDCS + nt * 16 + li] = f2bf(o);
          hp[nt][q] = h2[nt][q];
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const int px = lane >> 3, cv = lane & 7, pw = 7 * t + px;
      if (px < 7 && pw < Wp)
        *reinterpret_cast<u32x4*>(&out[((((size_t)b * Hp + ph0 + k) * Wp + pw) * 64) + cv * 8]) = *reinterpret_cast<const u32x4*>(&mycs[px * LDCS + cv * 8]);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Stem weight gradient: dW[n][(r,s,c)] += sum_pixels dy[pixel][n] * im2col(img)[pixel][(c,r,s8)]
// Persistent workgroups walk (image, 2-output-row) blocks; per output row the im2col slice [pixels][192] is built in LDS
// from the resident patch, dy's row is staged [pixels][64], and both MFMA operands are read with ds_read_b64_tr_b16
// (contraction index = pixel).  Partial dW stays in registers over the whole walk; one atomic flush per workgroup.
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int RBW = 4;             // output rows per block (2 measured 12 % slower, 0.98 vs 0.86 ms at B=512: 9 input rows and one patch-load barrier per 2 output rows)
constexpr int PRW = 2 * RBW + 5;   // patch rows
constexpr int LDA = KP + 8;        // im2col row stride
constexpr int LDD = 64 + 4;        // dy row stride
}

// FUSED = true: dy is never materialised -- it is rebuilt per row from the raw conv output y, the pooled gradient + argmax
// (stem_route) and the BatchNorm backward coefficients (dy = A*g + B*y + C), i.e. vqa_stem_bwd_apply folded into the staging.
template <bool FUSED>
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const bf16_t* __restrict__ dy, float* __restrict__ dw,
                                                         int B, int H, int W, int Ho, int Wo, const bf16_t* __restrict__ dpool,
                                                         const uint8_t* __restrict__ idx, const float* __restrict__ coef,
                                                         const float* __restrict__ bc, int Hp, int Wp, int nsplit, float* __restrict__ ws) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PW = 2 * Wo + 8;
  const int Wh = Wo / nsplit;                                        // pixels per unit: a row is contracted in nsplit pieces so that
  const int MP = (Wh + 31) / 32 * 32;                                // several workgroups fit one CU; padded to the MFMA K step
  bf16_t* patch = reinterpret_cast<bf16_t*>(smem);                   // [3][PRW][PW]
  bf16_t* Acol = patch + 3 * PRW * PW;                               // [MP][LDA]
  bf16_t* Dy = Acol + MP * LDA;                                      // [MP][LDD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pp = li & 3;
  // zero the padded pixel rows once (they are never written again)
  for (int i = tid; i < (MP - Wh) * LDA; i += 256) Acol[Wh * LDA + i] = 0;
  for (int i = tid; i < (MP - Wh) * LDD; i += 256) Dy[Wh * LDD + i] = 0;

  // FUSED: this thread always stages channel vector cv = tid & 7 -> its BN / backward coefficients live in registers
  float f_sc[8], f_sh[8], f_a[8], f_b[8], f_c[8];
  if (FUSED) {
    const int c0f = (tid & 7) * 8;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f_sc[j] = coef[c0f + j]; f_sh[j] = coef[64 + c0f + j];
      f_a[j] = bc[c0f + j]; f_b[j] = bc[64 + c0f + j]; f_c[j] = bc[128 + c0f + j];
    }
  }
  // 32-bit buffer addressing for the gathers of the fused path and the patch fill (the 64-bit pointer arithmetic of the plain loads was
  // 185 v_lshl_add_u64 in this kernel, which is VALU-bound in its staging phase)
  const __amdgpu_buffer_rsrc_t rsP = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(FUSED ? dpool : dy), 0, FUSED ? B * Hp * Wp * 64 * 2 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsI = __builtin_amdgcn_make_buffer_rsrc(const_cast<uint8_t*>(FUSED ? idx : reinterpret_cast<const uint8_t*>(dy)), 0, FUSED ? B * Hp * Wp * 64 : 0, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsImg = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(img), 0, B * 3 * H * W * 4, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsDy = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(dy), 0, B * Ho * Wo * 64 * 2, 0x00020000);
  f32x4 acc[4][3];                                                   // wave owns k2 tiles 3*wave .. 3*wave+2, all 4 n tiles
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 3; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int rblocks = Ho / RBW, nblocks = B * rblocks;
  for (int blk = blockIdx.x; blk < nblocks; blk += gridDim.x) {
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBW;
    __syncthreads();                                                 // previous block's readers are done with the patch
    const int ih_base = 2 * oh0 - 3;
    // (scalar fill on purpose: with stem_fill_patch inlined here the fused kernel needs 171 instead of 118 registers, two workgroups per
    //  CU instead of three, and the stem backward goes 709 -> 940 us)
    for (int x = tid; x < PW; x += 256) {
      const int iw = x - 3;
      const bool cok = iw >= 0 && iw < W;
      float vals[3 * PRW];
#pragma unroll
      for (int row = 0; row < 3 * PRW; ++row) {
        const int c = row / PRW, pr = row - c * PRW, ih = ih_base + pr;
        const bool ok = cok && ih >= 0 && ih < H;
        const int ihc = min(max(ih, 0), H - 1), iwc = min(max(iw, 0), W - 1);
        const float t = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rsImg, (((b * 3 + c) * H + ihc) * W + iwc) * 4, 0, 0));
        vals[row] = ok ? t : 0.f;
      }
#pragma unroll
      for (int row = 0; row < 3 * PRW; ++row) patch[row * PW + x] = f2bf(vals[row]);
    }
    for (int unit = 0; unit < RBW * nsplit; ++unit) {
      const int orow = unit / nsplit, px0 = (unit - orow * nsplit) * Wh;
      __syncthreads();                                               // patch ready / previous unit's MFMA reads done
      // dy row piece -> LDS (16-byte vectors), im2col slice -> LDS
      const int dyr_b = (((b * Ho + oh0 + orow) * Wo + px0) * 64) * 2;     // byte offset of the row piece (32-bit buffer addressing)
      if constexpr (FUSED) {
        // item = (pixel pair (2q, 2q+1) of the piece, channel vector): the pair shares its pooling windows (stem_route_pair_buf)
        auto stage_row = [&](auto odd) {             // the row's parity picks the body: two pooling-window rows, or one
          for (int v = tid; v < (Wh >> 1) * 8; v += 256) {
            const int pq = v >> 3, cv = v & 7;
            Vec16<bf16_t> yy[2];
            yy[0].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsDy, dyr_b + (2 * pq) * 128 + cv * 16, 0, 0));
            yy[1].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsDy, dyr_b + (2 * pq + 1) * 128 + cv * 16, 0, 0));
            float g8[2][8];
            stem_route_pair_buf<decltype(odd)::value>(rsP, rsI, yy, f_sc, f_sh, b, oh0 + orow, (px0 >> 1) + pq, cv * 8, Hp, Wp, g8);
#pragma unroll
            for (int q2 = 0; q2 < 2; ++q2) {
              Vec16<bf16_t> o;
#pragma unroll
              for (int j = 0; j < 8; ++j) o.set(j, f_a[j] * g8[q2][j] + f_b[j] * yy[q2].get(j) + f_c[j]);
              uint32_t* d = reinterpret_cast<uint32_t*>(&Dy[(2 * pq + q2) * LDD + cv * 8]);     // LDD*2 = 136 B rows: 8-byte aligned
              d[0] = o.raw[0]; d[1] = o.raw[1]; d[2] = o.raw[2]; d[3] = o.raw[3];
            }
          }
        };
        if ((oh0 + orow) & 1) stage_row(std::true_type{});
        else stage_row(std::false_type{});
      } else {
        for (int v = tid; v < Wh * 8; v += 256) {
          const int px = v >> 3, cv = v & 7;
          const u32x4 val = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsDy, dyr_b + v * 16, 0, 0));
          uint32_t* d = reinterpret_cast<uint32_t*>(&Dy[px * LDD + cv * 8]);     // LDD*2 = 136 B rows: 8-byte aligned
          d[0] = val[0]; d[1] = val[1]; d[2] = val[2]; d[3] = val[3];
        }
      }
      if (tid < 240) {                                               // 10 pixels x 24 (c,r) pairs per pass
        const int pair = tid % 24, c = pair / 7, r = pair - c * 7;
        const bf16_t* prow = patch + (c * PRW + 2 * orow + r) * PW + 2 * px0;
        for (int px = tid / 24; px < Wh; px += 10) {
          u32x4 val = {0u, 0u, 0u, 0u};
          if (pair < 21) {
            const uint32_t* ap = reinterpret_cast<const uint32_t*>(prow + 2 * px);
            val = u32x4{ap[0], ap[1], ap[2], ap[3] & 0x0000ffffu};   // s8 = 7 is padding
          }
          *reinterpret_cast<u32x4*>(&Acol[px * LDA + pair * 8]) = val;
        }
      }
      __syncthreads();
      typedef __attribute__((ext_vector_type(8))) short i16x8;
      for (int ks = 0; ks < MP / 32; ++ks) {
        const bf16_t* yb = Dy + (ks * 32 + 8 * g + q) * LDD + 4 * pp;
        const bf16_t* xb = Acol + (ks * 32 + 8 * g + q) * LDA + wave * 48 + 4 * pp;
        bf16x8 af[4], bfv[3];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(yb + i * 16));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(yb + 4 * LDD + i * 16));
          i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[i] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(xb + j * 16));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(xb + 4 * LDA + j * 16));
          i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfv[j] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 3; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfv[j], acc[i][j], 0, 0, 0);
      }
    }
  }
  // flush: D[i = n][j = k2], k2 = (c*7+r)*8+s  ->  dw[n][(r*7+s)*3+c]
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    const int k2 = wave * 48 + j * 16 + li, pair = k2 >> 3, s = k2 & 7;
    if (pair >= 21 || s >= 7) continue;
    const int c = pair / 7, r = pair - c * 7;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const size_t o = (size_t)(i * 16 + g * 4 + rr) * 147 + (r * 7 + s) * 3 + c;
        // with a workspace: this workgroup's own [64][147] slab (every element exactly once), summed later in a fixed order
        if (ws) ws[(size_t)blockIdx.x * (64 * 147) + o] = acc[i][j][rr];
        else atomicAdd(dw + o, acc[i][j][rr]);
      }
  }
}

extern "C" int vqa_slab_reduce(const float* ws, float* dw, int nslabs, long long n, hipStream_t st);

// A row is contracted in nsplit pieces (Wo / nsplit pixels each, a multiple of 8).  The kernel is latency-bound in its staging phase
// and lives off the co-resident workgroups: the smallest split whose LDS footprint lets THREE of them share a CU (224 x 224 images:
// 2 pieces, 52 KB; the 384 x 384 stress shape: 6 pieces, 48 KB -- with the 2 pieces it used to get, 82 KB and ONE workgroup per CU,
// the launch took 1.87 ms).
static size_t stem_wgrad_shm(int Wo, int nsplit) {
  const int PW = 2 * Wo + 8, MP = (Wo / nsplit + 31) / 32 * 32;
  return (size_t)(3 * PRW * PW + MP * LDA + MP * LDD) * 2;
}
static int stem_nsplit(int Wo) {
  const int env = vqa_env_int("VQA_STEM_NSPLIT", 0);
  if (env > 0 && Wo % (8 * env) == 0) return env;
  const int floor_ = Wo > 64 ? 2 : 1;
  for (int n = floor_; n <= 16; ++n)
    if (Wo % (8 * n) == 0 && stem_wgrad_shm(Wo, n) <= (size_t)160 * 1024 / 3) return n;
  return floor_;
}
static int stem_wgrad_grid(int B, int Ho, int Wo, size_t* shm_out) {
  const size_t shm = stem_wgrad_shm(Wo, stem_nsplit(Wo));
  if (shm_out) *shm_out = shm;
  if (shm > 160 * 1024) return 0;
  const int nblocks = B * (Ho / RBW);
  const int cap = 256 * (int)((160 * 1024) / shm > 4 ? 4 : (160 * 1024) / shm);
  return nblocks < cap ? nblocks : cap;
}

extern "C" {

// number of workgroups (= rows of the statistics slab) or 0 when the shape is not supported by this kernel
int vqa_stem_conv_blocks(int B, int H, int W) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  if (Ho % RB || Wo % 16 || Wo > 256) return 0;
  return B * (Ho / RB);
}
int vqa_stem_pack(const float* w_krsc, void* wstem /* [64][192] bf16 */, hipStream_t st) {
  if (!w_krsc || !wstem) return VQA_EARG;
  hipLaunchKernelGGL(stem_pack_kernel, dim3((64 * KP + 255) / 256), dim3(256), 0, st, w_krsc, (bf16_t*)wstem);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// bf16 only.  img NCHW fp32 [B][3][H][W]; out NHWC bf16 [B][Ho][Wo][64]; stats [vqa_stem_conv_blocks][2][64] or NULL
int vqa_stem_conv(const float* img, const void* wstem, void* out, float* stats, int B, int H, int W, hipStream_t st) {
  const int nb = vqa_stem_conv_blocks(B, H, W);
  if (!img || !wstem || !out || nb <= 0 || (size_t)B * 3 * H * W * 4 >= 0x7fffffffull) return VQA_EARG;      // 32-bit buffer offsets into the image
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1, PW = 2 * Wo + 8;
  const size_t shm = (size_t)(3 * PR * PW + 64 * LDW) * 2 + 4 * 64 * 2 * 4;
  static size_t attr = 0;
  if (shm > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = shm; }
  hipLaunchKernelGGL(stem_conv_kernel, dim3(nb), dim3(256), shm, st, img, (const bf16_t*)wstem, (bf16_t*)out, stats, H, W, Ho, Wo);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

// Inference stem in one launch (see stem_conv_pool_kernel).  bf16 only.  coef: scale[64] | shift[64] (vqa_bn_eval_coef); out: the
// POOLED activation NHWC bf16 [B][Hp][Wp][64].  vqa_stem_conv_pool_ok: 1 when the shape is supported (even conv output, pooled rows
// a multiple of 4, the workgroup's LDS within 160 KB).
static size_t stem_conv_pool_shm(int H, int W, int* ctiles_out, int* pw_out) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1, Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  if (H < 7 || W < 7 || (Ho & 1) || (Wo & 1) || Hp % PRB) return 0;
  const int ctiles = (Wp + 6) / 7;
  const int PW = (2 * (14 * ctiles + 2) + 8 + 7) / 8 * 8;            // x = 28 t + 2 li + 0..7 <= 28 (ctiles-1) + 37
  if (ctiles_out) *ctiles_out = ctiles;
  if (pw_out) *pw_out = PW;
  const size_t shm = (size_t)(3 * PRP * PW + 64 * LDW) * 2;
  return shm <= 160 * 1024 ? shm : 0;
}
int vqa_stem_conv_pool_ok(int B, int H, int W) {
  return B > 0 && stem_conv_pool_shm(H, W, nullptr, nullptr) > 0 && (size_t)B * 3 * H * W * 4 < 0x7fffffffull;
}
int vqa_stem_conv_pool(const float* img, const void* wstem, const float* coef, void* out, int B, int H, int W, hipStream_t st) {
  int ctiles = 0, PW = 0;
  const size_t shm = stem_conv_pool_shm(H, W, &ctiles, &PW);
  if (!img || !wstem || !coef || !out || !vqa_stem_conv_pool_ok(B, H, W)) return VQA_EARG;
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1, Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  static size_t attr = 0;
  if (shm > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_conv_pool_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = shm; }
  hipLaunchKernelGGL(stem_conv_pool_kernel, dim3(B * (Hp / PRB)), dim3(256), shm, st, img, (const bf16_t*)wstem, coef, (bf16_t*)out,
                     B, H, W, Ho, Wo, Hp, Wp, ctiles, PW, vqa_env_int("VQA_STEMCP_DBG", 0));
  VQA_LAUNCH_CHECK(); return VQA_OK;
}

// workgroups of the stem weight-gradient kernels (0: unsupported shape): their deterministic two-pass accumulation needs a scratch
// of (this many) * 64*147 floats
int vqa_stem_wgrad_blocks(int B, int H, int W) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  if (Ho % RBW || Wo % 16 || Wo > 256) return 0;
  return stem_wgrad_grid(B, Ho, Wo, nullptr);
}

// bf16 only.  dw [64][7][7][3] fp32 (+=).  Same shape support as vqa_stem_conv.  ws: scratch (see vqa_stem_wgrad_blocks) or NULL (atomics)
int vqa_stem_wgrad(const float* img, const void* dy, float* dw, int B, int H, int W, float* ws, long long ws_floats, hipStream_t st) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1;
  if (!img || !dy || !dw || Ho % RBW || Wo % 16 || Wo > 256) return VQA_EARG;
  if ((size_t)B * Ho * Wo * 64 * 2 >= 0x7fffffffull || (size_t)B * 3 * H * W * 4 >= 0x7fffffffull) return VQA_EARG;     // 32-bit buffer offsets
  const int nsplit = stem_nsplit(Wo);
  const int PW = 2 * Wo + 8, MP = (Wo / nsplit + 31) / 32 * 32;
  const size_t shm = (size_t)(3 * PRW * PW + MP * LDA + MP * LDD) * 2;
  if (shm > 160 * 1024) return VQA_EARG;
  static size_t attr = 0;
  if (shm > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = shm; }
  int nblocks = B * (Ho / RBW);
  const int cap = 256 * (int)((160 * 1024) / shm > 4 ? 4 : (160 * 1024) / shm);
  int grid = nblocks < cap ? nblocks : cap;
  float* w = (ws && ws_floats >= (long long)grid * 64 * 147) ? ws : nullptr;
  hipLaunchKernelGGL(stem_wgrad_kernel<false>, dim3(grid), dim3(256), shm, st, img, (const bf16_t*)dy, dw, B, H, W, Ho, Wo,
                     (const bf16_t*)nullptr, (const uint8_t*)nullptr, (const float*)nullptr, (const float*)nullptr, 0, 0, nsplit, w);
  VQA_LAUNCH_CHECK();
  return w ? vqa_slab_reduce(w, dw, grid, 64 * 147, st) : VQA_OK;
}
// Fused stem BatchNorm/ReLU/MaxPool backward + weight gradient: dy = A*g + B*y + C is rebuilt on the fly from the raw conv
// output y [B][Ho][Wo][64], the pooled gradient dpool [B][Hp][Wp][64] + argmax idx, coef (4*64) and bcoef (3*64).
int vqa_stem_wgrad_fused(const float* img, const void* y, const void* dpool, const uint8_t* idx, const float* coef, const float* bcoef,
                         float* dw, int B, int H, int W, float* ws, long long ws_floats, hipStream_t st) {
  const int Ho = (H + 6 - 7) / 2 + 1, Wo = (W + 6 - 7) / 2 + 1, Hp = (Ho + 2 - 3) / 2 + 1, Wp = (Wo + 2 - 3) / 2 + 1;
  if (!img || !y || !dpool || !idx || !coef || !bcoef || !dw || Ho % RBW || Wo % 16 || Wo > 256) return VQA_EARG;
  if ((size_t)B * Ho * Wo * 64 * 2 >= 0x7fffffffull || (size_t)B * 3 * H * W * 4 >= 0x7fffffffull) return VQA_EARG;     // 32-bit buffer offsets
  const int nsplit = stem_nsplit(Wo);
  const int PW = 2 * Wo + 8, MP = (Wo / nsplit + 31) / 32 * 32;
  const size_t shm = (size_t)(3 * PRW * PW + MP * LDA + MP * LDD) * 2;
  if (shm > 160 * 1024) return VQA_EARG;
  static size_t attr = 0;
  if (shm > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&stem_wgrad_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = shm; }
  int nblocks = B * (Ho / RBW);
  const int cap = 256 * (int)((160 * 1024) / shm > 4 ? 4 : (160 * 1024) / shm);
  int grid = nblocks < cap ? nblocks : cap;
  float* w = (ws && ws_floats >= (long long)grid * 64 * 147) ? ws : nullptr;
  hipLaunchKernelGGL(stem_wgrad_kernel<true>, dim3(grid), dim3(256), shm, st, img, (const bf16_t*)y, dw, B, H, W, Ho, Wo,
                     (const bf16_t*)dpool, idx, coef, bcoef, Hp, Wp, nsplit, w);
  VQA_LAUNCH_CHECK();
  return w ? vqa_slab_reduce(w, dw, grid, 64 * 147, st) : VQA_OK;
}

}  // extern "C"
