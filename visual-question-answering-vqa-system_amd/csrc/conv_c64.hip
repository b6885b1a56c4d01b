// 3x3 / stride 1 / pad 1 convolution with 64 input and 64 output channels (stage 1 of the ResNet, bf16):
// forward (models/cnn_backbone.py:182-187 with Cin=Cout=64), its data gradient (same kernel, flipped weights) and its
// weight gradient, written around an LDS-resident INPUT PATCH instead of the generic implicit-GEMM gather:
//   * a workgroup walks (image, 2 output rows) blocks persistently; the 4 x (W+2) x 64 input patch of a block is loaded
//     once (zero padded by the buffer range check), stored XOR-swizzled in LDS and reused by all 9 filter taps;
//   * forward/dgrad: every wave keeps its 36 weight fragments (its 32 output channels x K=576) in registers for the
//     whole launch; A fragments are ds_read_b128 straight out of the patch (implicit im2col in LDS);
//   * wgrad: contraction over pixels, both operands via ds_read_b64_tr_b16, 64x576 partial dW in registers over the
//     whole walk, one atomic flush per workgroup.
// These layers are HBM-bound in bf16 (288 flop/byte); the patch cuts L2/HBM reads ~9x versus a per-tap gather.
#include <cstdlib>
#include <utility>
#include "common.h"

namespace {
constexpr int RBF = 2;            // output rows per block, forward / data gradient
constexpr int RBG = 4;            // output rows per block, weight gradient
constexpr int CH = 64;            // channels (in and out)
constexpr int LDE = 32 + 8;       // epilogue staging row stride (elements)
constexpr int OOBV = (int)0x80000000;

// element offset of (patch row, patch col, channel chunk) in the swizzled patch: 16-byte chunks XOR (col & 7)
__device__ __forceinline__ int patch_off(int prow, int pcol, int chunk, int PWc) {
  return ((prow * PWc + pcol) * 8 + (chunk ^ (pcol & 7))) * 8;
}
}

struct C64Params {
  const bf16_t* x; const bf16_t* w; bf16_t* out; float* stats; const bf16_t* addend; const bf16_t* addmask;
  const bf16_t* outmask;     // 8-wave kernel, EPI variant: out = (conv + addend * (addmask > 0)) * (outmask > 0)  (addmask, outmask optional)
  // 8-wave kernel, BNRED variant: the stored tile is the gradient entering relu(BatchNorm(bn_y)) (conv2's data gradient da1 entering
  // bn1): its BatchNorm-backward column sums  sum g | sum g * xhat,  g = out * [bn_y * scale + shift > 0]  go to bn_facc
  // (vqa_bn_acc_words(3, 64), the layout vqa_bn_bwd_apply_acc reads) and the vqa_bn_bwd_reduce pass over (out, bn_y) is skipped
  const bf16_t* bn_y; const float* bn_coef; unsigned long long* bn_facc;
  int stats_mode;            // 1: stats is a fixed-point accumulator u64 [vqa_bn_acc_words(2, 64)] (common.h acc_add_fixed), not a per-workgroup slab
  int B, H, W; unsigned x_bytes;
  int dbg;                        // VQA_C64P_DBG (measurement only, wrong results): bit 0 no epilogue, bit 1 no MFMA loop, bit 2 no in-loop DMA
  // 8-wave kernel only: the conv runs on relu(BatchNorm(x)) without that tensor ever existing (PreBn below)
  int pre_mode;                   // 0: x as it is; 1: scale | shift from pre_coef[2][64]; 2: finalize pre (fixed-point statistics of x) in the prologue
  const float* pre_coef; BnAcc pre; double pre_inv_count, pre_unbias; float pre_momentum, pre_eps;
};

// ------------------------------------------------------------------------------------------------
// Round 4: training-mode "Conv3x3 + BN + ReLU" (models/cnn_backbone.py:182-187) without the normalised tensor.  The 8-wave patch
// kernels already hold their input patch in LDS for nine taps; when the input is relu(bn1(y1)) they take y1 and apply
// scale / shift / ReLU to the patch IN LDS, once per block, right after it has landed -- a1 is never written or read (stage 1 at
// B = 512: 205 MB each way per residual block, a whole bn_apply pass).  Same arithmetic as bn_apply_acc_kernel (fp32 fma of the
// bf16 value, NaN-propagating ReLU, one rounding to bf16), so conv(patch) is bit-identical to conv(bn_apply(y1)).
// Padding stays ZERO, not relu(shift): halo columns are never touched (zeroed once, DMA never writes them), rows outside the image
// are skipped (the DMA wrote zeros there).  A thread keeps ONE channel chunk (8 channels: 16 coefficient registers, re-read from LDS
// each block while the MFMA ring registers are dead) and walks pixels; SWZ16 selects the weight-gradient kernels' slot swizzle.
// ------------------------------------------------------------------------------------------------
namespace {
template <bool SWZ16, int NROWS>
__device__ __forceinline__ void patch_bn_relu(char* patch, int W, int PWc, int ih0, int H, const float* cf, int tid) {
  const int c8 = tid & 7, pc0 = 1 + (tid >> 3);
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { sc[j] = cf[c8 * 8 + j]; sh[j] = cf[64 + c8 * 8 + j]; }
  // ROWS patch rows per batch: all their 16-byte reads are issued before the first value is used (a one-row-at-a-time loop was a chain
  // of dependent LDS latencies: +33 us on the stage-1 forward conv, as much as the bn_apply pass it replaces)
  constexpr int ROWS = NROWS <= 6 ? NROWS : (NROWS % 5 == 0 ? 5 : 2);
  for (int pc = pc0; pc <= W; pc += 64) {
    const int slot = c8 ^ (SWZ16 ? ((pc & 7) ^ (((pc >> 3) & 1) << 2)) : (pc & 7));
    char* col = patch + (pc * 8 + slot) * 16;
#pragma unroll
    for (int r0 = 0; r0 < NROWS; r0 += ROWS) {
      Vec16<bf16_t> v[ROWS];
#pragma unroll
      for (int i = 0; i < ROWS; ++i) v[i].raw = *reinterpret_cast<const u32x4*>(col + (r0 + i) * PWc * 128);
#pragma unroll
      for (int i = 0; i < ROWS; ++i) {
        if ((unsigned)(ih0 + r0 + i) >= (unsigned)H) continue;       // a padding row (first / last block of an image): stays zero
        // 7 VALU per bf16 pair: 2 unpack, 2 fma, 2 v_maximum3_f32 (IEEE-2019 maximum: NaN-propagating like bn_apply's (x < 0 ? 0 : x);
        // -0 becomes +0, equal as a number and as a conv operand), 1 v_cvt_pk_bf16_f32.  The element-wise get / set form compiled to
        // ~2.5x that (per-element convert + bit-field insert) and made the prologue cost as much as the pass it replaces
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        u32x4 o;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
          const uint32_t w = v[i].raw[k];
          float x0 = __builtin_fmaf(__uint_as_float(w << 16), sc[2 * k], sh[2 * k]);
          float x1 = __builtin_fmaf(__uint_as_float(w & 0xffff0000u), sc[2 * k + 1], sh[2 * k + 1]);
          x0 = __builtin_elementwise_maximum(x0, 0.f); x1 = __builtin_elementwise_maximum(x1, 0.f);
          o[k] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){x0, x1}, bf16x2_t));
        }
        *reinterpret_cast<u32x4*>(col + (r0 + i) * PWc * 128) = o;
      }
    }
  }
}
// scale | shift of the input's BatchNorm into LDS cf[2][64] (call with >= 64 threads, then a barrier).  mode 2: finalized here from
// the fixed-point statistics (workgroup 0 publishes coef[4][64] and updates the running statistics, like bn_apply_acc_kernel)
__device__ __forceinline__ void pre_bn_coef(int mode, const float* pre_coef, const BnAcc& pre, double inv_count, double unbias, float momentum,
                                            float eps, float* cf, int tid) {
  if (tid < 64) {
    if (mode == 2) bn_acc_coef(pre, 64, tid, inv_count, unbias, momentum, eps, blockIdx.x == 0, cf[tid], cf[64 + tid]);
    else { cf[tid] = pre_coef[tid]; cf[64 + tid] = pre_coef[64 + tid]; }
  }
}
}

__global__ __launch_bounds__(256) void conv3x3_c64_kernel(C64Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PWc = p.W + 2;
  const int patch_elems = (RBF + 2) * PWc * CH;
  bf16_t* patch0 = reinterpret_cast<bf16_t*>(smem);
  bf16_t* patch1 = patch0 + patch_elems;
  bf16_t* Est = patch1 + patch_elems;                       // [4 waves][16][LDE]
  float* red = reinterpret_cast<float*>(Est + 4 * 16 * LDE); // [2 m-waves][64][2]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int wn = wave & 1, wm = wave >> 1;                   // wave owns channels [32*wn, 32*wn+32), m tiles wm, wm+2, ...
  const int rblocks = p.H / RBF, nblocks = p.B * rblocks;
  const int mtiles = RBF * p.W / 16;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)p.x_bytes, 0x00020000);

  // ---- weight fragments: w[n][(r,s,c)] (576 per row), resident for the whole launch
  bf16x8 bfr[18][2];
#pragma unroll
  for (int kk = 0; kk < 18; ++kk)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bfr[kk][nt] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(wn * 32 + nt * 16 + li) * 576 + kk * 32 + g * 8);

  // ---- patch staging descriptors, computed once: chunk id -> LDS offset, offset relative to the block's first pixel,
  //      patch row (for the per-block vertical range check).  No integer division inside the block loop.
  constexpr int MAXV = 8;
  const int nchunks = (RBF + 2) * PWc * 8;
  const float inv_pw = 1.0f / (float)PWc, inv_w = 1.0f / (float)p.W;
  int s_lds[MAXV], s_rel[MAXV]; unsigned long long s_prow = 0;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int id = tid + 256 * i;
    s_lds[i] = -1; s_rel[i] = OOBV;
    if (id < nchunks) {
      const int chunk = id & 7, q = id >> 3, prow = (int)(((float)q + 0.5f) * inv_pw), pcol = q - prow * PWc;
      s_lds[i] = patch_off(prow, pcol, chunk, PWc);
      const int iw = pcol - 1;
      if ((unsigned)iw < (unsigned)p.W) s_rel[i] = (((prow - 1) * p.W + iw) * CH + chunk * 8) * 2;
      s_prow |= (unsigned long long)prow << (4 * i);
    }
  }
  u32x4 pre[MAXV];
  auto pload = [&](int blk) {
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBF;
    const int base = ((b * p.H + oh0) * p.W) * CH * 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int ih = oh0 - 1 + (int)((s_prow >> (4 * i)) & 15);
      const int off = ((unsigned)ih < (unsigned)p.H && s_rel[i] != OOBV) ? base + s_rel[i] : OOBV;
      pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
    }
  };
  auto pstore = [&](bf16_t* dst) {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (s_lds[i] >= 0) *reinterpret_cast<u32x4*>(dst + s_lds[i]) = pre[i];
  };

  float ssum[2] = {0.f, 0.f}, ssq[2] = {0.f, 0.f};
  bf16_t* myst = Est + wave * 16 * LDE;
  int blk = blockIdx.x, buf = 0;
  if (blk < nblocks) { pload(blk); pstore(patch0); }
  __syncthreads();
  for (; blk < nblocks; blk += gridDim.x, buf ^= 1) {
    const int nxt = blk + gridDim.x;
    if (nxt < nblocks) pload(nxt);
    const bf16_t* pt = buf ? patch1 : patch0;
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBF;
    for (int mt = wm; mt < mtiles; mt += 2) {
      const int px = mt * 16 + li, orow = (int)(((float)px + 0.5f) * inv_w), ow = px - orow * p.W;
      // per-tile address terms: e[s][half] = offset of (row orow, col ow+s, swizzled chunk half*4+g); taps add r*PWc*64
      int e[3][2];
#pragma unroll
      for (int s = 0; s < 3; ++s) {
        const int col = ow + s, cb = (orow * PWc + col) * 64;
        e[s][0] = cb + ((g ^ (col & 7)) << 3);
        e[s][1] = cb + (((4 + g) ^ (col & 7)) << 3);
      }
      const int rowstep = PWc * 64;
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kk = 0; kk < 18; ++kk) {
        const int tap = kk >> 1, r = tap / 3, s = tap - r * 3;
        const bf16x8 af = *reinterpret_cast<const bf16x8*>(pt + e[s][kk & 1] + r * rowstep);
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[kk][0], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[kk][1], acc[1], 0, 0, 0);
      }
      // epilogue: 16 pixels x 32 channels of this wave -> LDS -> one 16-byte store per lane (+ optional addend)
#pragma unroll
      for (int nt = 0; nt < 2; ++nt)
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) {
          const float v = acc[nt][rr];
          ssum[nt] += v; ssq[nt] += v * v;
          myst[(g * 4 + rr) * LDE + nt * 16 + li] = f2bf(v);
        }
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      {
        const int pxl = lane >> 2, cv = lane & 3;
        const int pxo = mt * 16 + pxl, orow2 = (int)(((float)pxo + 0.5f) * inv_w), ow2 = pxo - orow2 * p.W;
        const size_t off = (((size_t)b * p.H + oh0 + orow2) * p.W + ow2) * CH + wn * 32 + cv * 8;
        Vec16<bf16_t> v; v.raw = *reinterpret_cast<const u32x4*>(&myst[pxl * LDE + cv * 8]);
        if (p.addend) {
          Vec16<bf16_t> av = ldg16(p.addend + off);
          if (p.addmask) {
            Vec16<bf16_t> mv = ldg16(p.addmask + off);
#pragma unroll
            for (int j = 0; j < 8; ++j) v.set(j, v.get(j) + (mv.get(j) > 0.f ? av.get(j) : 0.f));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v.set(j, v.get(j) + av.get(j));
          }
        }
        stg16(p.out + off, v);
      }
      asm volatile("" ::: "memory");
    }
    if (nxt < nblocks) pstore(buf ? patch0 : patch1);
    __syncthreads();
  }
  if (p.stats) {
#pragma unroll
    for (int nt = 0; nt < 2; ++nt) {
      float s = ssum[nt], q = ssq[nt];
      s += __shfl_xor(s, 16, 64); q += __shfl_xor(q, 16, 64);
      s += __shfl_xor(s, 32, 64); q += __shfl_xor(q, 32, 64);
      if (lane < 16) { red[(wm * 64 + wn * 32 + nt * 16 + lane) * 2] = s; red[(wm * 64 + wn * 32 + nt * 16 + lane) * 2 + 1] = q; }
    }
    __syncthreads();
    if (tid < 64) {
      p.stats[((size_t)blockIdx.x * 2) * 64 + tid] = red[tid * 2] + red[(64 + tid) * 2];
      p.stats[((size_t)blockIdx.x * 2 + 1) * 64 + tid] = red[tid * 2 + 1] + red[(64 + tid) * 2 + 1];
    }
  }
}

// ------------------------------------------------------------------------------------------------
// Round 2: persistent patch kernel with LDS-DMA patches, 8 waves, 8 output rows per block (forward and the addend-free data
// gradient of the 64-channel stage).  Why: the generic window-loader tile re-streams the 72 KB weight matrix for every 128-pixel
// tile (76 FLOP per staged byte) and these launches are bound by the per-CU L2->LDS ingest rate; here
//   * the weights never touch LDS: each wave keeps the 36 B fragments of its 32 output channels in registers for the launch;
//   * a block = (image, 8 output rows): its 10 x (W+2) x 64 input patch is brought in ONCE by LDS-DMA (inline asm, hidden from
//     hipcc's vmcnt bookkeeping: it would drain the prefetch in front of the first ds_read), one 1 KB piece = 8 pixels of a row,
//     swizzled on the source side, into the buffer the previous block is not reading -> 186 FLOP per staged byte;
//   * halo columns are zeroed once, halo rows outside the image are out-of-range source offsets;
//   * no vector-memory LOAD is issued by the computing waves besides the DMA (stores only), so nothing ever waits on the
//     prefetch before the block-end barrier -- which is why the variant with an identity addend stays on the generic kernel.
namespace {
// RBP = output rows per block: 8 where two (8+2) x (W+2) x 64 patches fit the 160 KB of LDS (W <= 62: the 56 x 56 maps of the default
// configuration), 4 for wider maps (W <= 126: the 96 x 96 maps of the 384 x 384 stress configuration; 1.5x instead of 1.25x halo reads)
typedef int i32x4_c64 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void dma16_c64(i32x4_c64 rs, unsigned lds_addr, int voff, int soff) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr), "s"(soff) : "memory");
}
}

// EPI (round 4): the data gradient of a residual block's conv1 with the identity-path gradient and the ReLU masks in the epilogue
// (engine._block_bwd; vqa_igemm's epilogue on the bf16 conv value, bit for bit).  A lane's three 16-byte epilogue loads are issued at
// the top of its tile, ~36 MFMAs ahead of their use; the waits the compiler counts for them also cover the (older, hidden) DMA pieces
// of the prefetch, which at one tile's distance have landed.  No statistics in this variant: their 16 registers hold the loads.
// BNRED (round 4, MODE 2): conv2's data gradient also leaves bn1's backward column sums (C64Params).  A lane owns 8 channels of one
// pixel per tile: it loads its 16 bytes of bn_y at the top of the tile, reads its channels' scale | shift | mean from LDS at the
// epilogue (no registers left to keep them: 245 of 256 are taken), masks the bf16 value it stores and adds g, g * (y - mean) to
// the 16 registers the forward variant uses for sum / sum of squares; invstd is applied once, in the final fold.
template <int RBP, int MODE>
__global__ __launch_bounds__(512, 2) void conv3x3_c64p_kernel(C64Params p) {
  constexpr bool EPI = MODE == 1, BNRED = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int PWc = p.W + 2;
  const int patch_elems = (RBP + 2) * PWc * CH;
  bf16_t* patch0 = reinterpret_cast<bf16_t*>(smem);
  bf16_t* patch1 = patch0 + patch_elems;
  float* red = reinterpret_cast<float*>(patch1 + patch_elems);  // [4 m-waves][64][2]
  float* cf = red + 4 * 64 * 2;                                 // [2][64] scale | shift of the input's BatchNorm (pre_mode != 0); BNRED: coef [4][64]
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave & 1, wm = wave >> 1;                    // wave owns channels [32*wn, 32*wn+32), m tiles wm, wm+4, ...
  const int rblocks = p.H / RBP, nblocks = p.B * rblocks;
  const int mtiles = RBP * p.W / 16;
  const unsigned long long xa = (unsigned long long)p.x;
  const i32x4_c64 rsX = {(int)(unsigned)xa, (int)((unsigned)(xa >> 32) & 0xffffu), (int)p.x_bytes, 0x00020000};
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);

  // ---- weight fragments: w[n][(r,s,c)] (576 per row), resident for the whole launch.  The weights are the MFMA's A operand
  //      (D = W X^T: a lane ends up with 4 consecutive rows = output channels of ONE pixel), and the fragment's row i is mapped to
  //      channel 8*(i>>2) + 4*nt + (i&3): the two 16-row tiles together give every lane 8 CONSECUTIVE channels of its pixel, i.e.
  //      one 16-byte global store per lane straight from the accumulators -- no LDS staging in the epilogue.
  bf16x8 bfr[18][2];
#pragma unroll
  for (int kk = 0; kk < 18; ++kk)
#pragma unroll
    for (int nt = 0; nt < 2; ++nt)
      bfr[kk][nt] = *reinterpret_cast<const bf16x8*>(p.w + (size_t)(wn * 32 + 8 * (li >> 2) + 4 * nt + (li & 3)) * 576 + kk * 32 + g * 8);

  // ---- halo columns of both patch buffers: zero, once (the DMA pieces only ever write columns 1 .. W)
  for (int i = tid; i < 2 * (RBP + 2) * 2 * 8; i += 512) {
    const int chunk = i & 7, side = (i >> 3) & 1, prow = (i >> 4) % (RBP + 2), buf = (i >> 4) / (RBP + 2);
    bf16_t* pt = buf ? patch1 : patch0;
    *reinterpret_cast<u32x4*>(pt + ((size_t)(prow * PWc + (side ? PWc - 1 : 0)) * 8 + chunk) * 8) = u32x4{0u, 0u, 0u, 0u};
  }
  // ---- DMA plan: piece id = prow * ppr + j covers pixels 8j .. 8j+7 of image row (oh0 - 1 + prow); wave w issues ids w, w+8, ...
  const int ppr = p.W >> 3, npieces = (RBP + 2) * ppr;
  const int dpx = lane >> 3, dpos = lane & 7;
  const int dvoff = dpx * 128 + ((dpos ^ ((1 + dpx) & 7)) << 4);      // pixel row + swizzled SOURCE chunk (pcol & 7 == (1 + px) & 7)
  constexpr int OOB = (int)0x80000000;
  auto issue_patch = [&](int blk, int buf) {
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBP;
    const unsigned base = lds0 + (unsigned)buf * (unsigned)(patch_elems * 2);
    for (int id = wave; id < npieces; id += 8) {
      const int prow = id / ppr, j = id - prow * ppr, ih = oh0 - 1 + prow;
      const bool ok = (unsigned)ih < (unsigned)p.H;
      dma16_c64(rsX, base + (unsigned)((prow * PWc + 1 + 8 * j) * 128), ok ? dvoff : OOB, ok ? ((b * p.H + ih) * p.W + 8 * j) * 128 : 0);
    }
  };

  const float inv_w = 1.0f / (float)p.W;
  float ssum[8], ssq[8];                                        // this lane's 8 channels (32*wn + 8*g + j), summed over its pixels
#pragma unroll
  for (int j = 0; j < 8; ++j) { ssum[j] = 0.f; ssq[j] = 0.f; }
  const int tiles_pw = (p.dbg & 1) ? 0 : (mtiles - wm + 3) / 4;   // output stores this wave issues per block (uniform)
  int blk = blockIdx.x, buf = 0;
  if (blk < nblocks) issue_patch(blk, 0);
  if (p.pre_mode) pre_bn_coef(p.pre_mode, p.pre_coef, p.pre, p.pre_inv_count, p.pre_unbias, p.pre_momentum, p.pre_eps, cf, tid);   // (under the first DMA)
  if (BNRED && tid < 256) cf[tid] = p.bn_coef[tid];
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; blk < nblocks; blk += gridDim.x, buf ^= 1) {
    const int nxt = blk + gridDim.x;
    if (nxt < nblocks && !(p.dbg & 4)) issue_patch(nxt, buf ^ 1);   // the other buffer: every wave finished reading it at the last barrier
    const unsigned pbase = lds0 + (unsigned)buf * (unsigned)(patch_elems * 2);
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBP;
    if (p.pre_mode) {                                               // normalise + ReLU the landed patch in place (LDS only: no vmcnt wait)
      patch_bn_relu<false, RBP + 2>(smem + (size_t)buf * (size_t)(patch_elems * 2), p.W, PWc, oh0 - 1, p.H, cf, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    // LDS byte addresses of a tile's six (tap column, channel half) fragments in patch row 0
    auto tile_addr = [&](int mt, unsigned (&ea)[3][2], int& orow, int& ow) {
      const int px = mt * 16 + li;
      orow = (int)(((float)px + 0.5f) * inv_w);
      ow = px - orow * p.W;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) {
        const int col = ow + s_, cb = (orow * PWc + col) * 128;
        ea[s_][0] = pbase + (unsigned)(cb + ((g ^ (col & 7)) << 4));
        ea[s_][1] = pbase + (unsigned)(cb + (((4 + g) ^ (col & 7)) << 4));
      }
    };
    const unsigned rowstep = (unsigned)(PWc * 128);
    // The activation fragments go through a ring of PF registers filled by hand-issued ds_read_b128 with counted lgkmcnt waits
    // (LDS returns in order): PF-1 reads stay in flight across the MFMAs and across tile boundaries.  Left to itself the compiler
    // keeps one read ahead, which at two waves per SIMD exposes the LDS latency on every pair of MFMAs.
    constexpr int PF = 6;                                       // divides 18: slot = kk % PF for this tile and the next
    bf16x8 ring[PF];
    unsigned ec[3][2], en[3][2];
    int orow, ow, orow_n, ow_n;
    tile_addr(wm, ec, orow, ow);
#pragma unroll
    for (int kk = 0; kk < PF; ++kk) {
      const int tap = kk >> 1, r = tap / 3, s_ = tap - r * 3;
      asm volatile("ds_read_b128 %0, %1" : "=v"(ring[kk]) : "v"(ec[s_][kk & 1] + (unsigned)r * rowstep));
    }
    // epilogue inputs of a tile (EPI / BNRED): loaded one tile ahead -- for the block's first tile here, for tile t + 1 in the epilogue
    // of tile t BEFORE that tile's output store, so that the (compiler-counted) wait for them never includes a store's acknowledgement
    // (first version: loads at the top of their own tile, i.e. behind the previous tile's store: waves parked on vmcnt for 55-59 % of
    // their life, MFMA pipe 28 % busy against 45 % for the plain variant; profiles/r04_sq_counters.json)
    size_t ooff = (((size_t)b * p.H + oh0 + orow) * p.W + ow) * CH + wn * 32 + 8 * g;
    Vec16<bf16_t> e_a, e_m, e_o;
    auto epi_loads = [&](size_t off) __attribute__((always_inline)) {
      if (EPI) {
        e_a = ldg16(p.addend + off);
        if (p.addmask) e_m = ldg16(p.addmask + off);
        if (p.outmask) e_o = ldg16(p.outmask + off);
      }
      if (BNRED) e_a = ldg16(p.bn_y + off);
    };
    if (MODE && wm < mtiles) epi_loads(ooff);
    for (int mt = wm; mt < mtiles; mt += 4) {
      tile_addr(mt + 4 < mtiles ? mt + 4 : mt, en, orow_n, ow_n);   // the block's last tile re-reads itself (drained at the barrier)
      f32x4 acc[2] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
#pragma unroll
      for (int kk = 0; kk < 18; ++kk) {
        asm volatile("s_waitcnt lgkmcnt(%1)" : "+v"(ring[kk % PF]) : "n"(PF - 1));
        acc[0] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][0], ring[kk % PF], acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfr[kk][1], ring[kk % PF], acc[1], 0, 0, 0);
        const int kn = (kk + PF) % 18, tap = kn >> 1, r = tap / 3, s_ = tap - r * 3;
        const unsigned ad = (kk + PF < 18 ? ec[s_][kn & 1] : en[s_][kn & 1]) + (unsigned)r * rowstep;
        asm volatile("ds_read_b128 %0, %1" : "=v"(ring[kk % PF]) : "v"(ad));
      }
      const int orow_c = orow, ow_c = ow;
      orow = orow_n; ow = ow_n;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_) { ec[s_][0] = en[s_][0]; ec[s_][1] = en[s_][1]; }
      if (p.dbg & 1) { asm volatile("" ::"v"(acc[0]), "v"(acc[1])); continue; }
      // epilogue: acc[nt][rr] = out[pixel li][channel 32*wn + 8*g + 4*nt + rr] -> statistics, packed bf16, ONE 16-byte store
      typedef __attribute__((ext_vector_type(2))) float f32x2_t;
      typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
      u32x4 o;
#pragma unroll
      for (int nt = 0; nt < 2; ++nt) {
#pragma unroll
        for (int rr = 0; rr < 4; ++rr) { if (MODE) break; const float v = acc[nt][rr]; ssum[4 * nt + rr] += v; ssq[4 * nt + rr] += v * v; }
        o[2 * nt] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){acc[nt][0], acc[nt][1]}, bf16x2_t));
        o[2 * nt + 1] = __builtin_bit_cast(uint32_t, __builtin_convertvector((f32x2_t){acc[nt][2], acc[nt][3]}, bf16x2_t));
      }
      if (EPI) {
        Vec16<bf16_t> v; v.raw = o;
        if (p.addmask) {
#pragma unroll
          for (int j = 0; j < 8; ++j) v.set(j, v.get(j) + (e_m.get(j) > 0.f ? e_a.get(j) : 0.f));
        } else {
#pragma unroll
          for (int j = 0; j < 8; ++j) v.set(j, v.get(j) + e_a.get(j));
        }
        if (p.outmask) {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (!(e_o.get(j) > 0.f)) v.set(j, 0.f);
        }
        o = v.raw;
      }
      if (BNRED) {                                                  // per element as bn_bwd_reduce_kernel<SELF>: the STORED bf16 value, the recomputed mask
        const int c0 = wn * 32 + 8 * g;
        Vec16<bf16_t> v; v.raw = o;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
          if (h) __builtin_amdgcn_sched_barrier(0);                 // one half's 12 coefficient registers at a time (the kernel is at the 256-register limit)
          const f32x4 sc = *reinterpret_cast<const f32x4*>(cf + c0 + 4 * h), sh = *reinterpret_cast<const f32x4*>(cf + 64 + c0 + 4 * h),
                      mu = *reinterpret_cast<const f32x4*>(cf + 128 + c0 + 4 * h);
#pragma unroll
          for (int j = 0; j < 4; ++j) {
            const float yj = e_a.get(4 * h + j);
            float gj = v.get(4 * h + j);
            if (!(yj * sc[j] + sh[j] > 0.f)) gj = 0.f;
            ssum[4 * h + j] += gj; ssq[4 * h + j] += gj * (yj - mu[j]);
          }
        }
      }
      (void)orow_c; (void)ow_c;
      const size_t ooff_n = (((size_t)b * p.H + oh0 + orow) * p.W + ow) * CH + wn * 32 + 8 * g;      // (orow / ow are the next tile's by now)
      if (MODE && mt + 4 < mtiles) epi_loads(ooff_n);
      *reinterpret_cast<u32x4*>(p.out + ooff) = o;
      ooff = ooff_n;
    }
    // next patch landed.  vmcnt retires in issue order and this wave issued its DMA pieces BEFORE its output stores: with at
    // least 4 stores behind them, "at most 4 outstanding" already proves the pieces are in LDS -- the stores keep draining
    // under the next block's MFMAs instead of being waited for here.
    if (tiles_pw >= 4) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }
  if (!EPI && (p.stats || BNRED)) {
    // fold the 16 pixels (lanes li) of each channel group, then the 4 m-waves through LDS, in a fixed order
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int o = 1; o < 16; o <<= 1) { ssum[j] += __shfl_xor(ssum[j], o, 64); ssq[j] += __shfl_xor(ssq[j], o, 64); }
    if (li == 0) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        red[(wm * 64 + wn * 32 + 8 * g + j) * 2] = ssum[j];
        red[(wm * 64 + wn * 32 + 8 * g + j) * 2 + 1] = ssq[j];
      }
    }
    __syncthreads();
    if (tid < 64) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < 4; ++w) { s += red[(w * 64 + tid) * 2]; q += red[(w * 64 + tid) * 2 + 1]; }
      if (BNRED) {                                                  // K = 3 layout of the BatchNorm-backward accumulators (row 2: the shortcut's, unused)
        const int R = acc_replicas(64);
        const size_t fr = (size_t)(blockIdx.x % R) * 192;
        acc_add_fixed(p.bn_facc, (size_t)R * 192, fr + tid, s);
        acc_add_fixed(p.bn_facc, (size_t)R * 192, fr + 64 + tid, q * cf[192 + tid]);
      } else if (p.stats_mode) {
        const int R = acc_replicas(64);
        unsigned long long* fa = reinterpret_cast<unsigned long long*>(p.stats);
        const size_t fr = (size_t)(blockIdx.x % R) * 128;
        acc_add_fixed(fa, (size_t)R * 128, fr + tid, s);
        acc_add_fixed(fa, (size_t)R * 128, fr + 64 + tid, q);
      } else {
        p.stats[((size_t)blockIdx.x * 2) * 64 + tid] = s;
        p.stats[((size_t)blockIdx.x * 2 + 1) * 64 + tid] = q;
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient: dw[n][(r,s,c)] += sum_px dy[px][n] * x[px + (r-1, s-1)][c]
// ------------------------------------------------------------------------------------------------
struct C64WgradParams { const bf16_t* x; const bf16_t* dy; float* dw; float* ws; int B, H, W; unsigned x_bytes, dy_bytes; int dbg; /* VQA_C64WP_DBG, measurement only: bit 1 no MFMA loop, bit 2 no in-loop DMA */
                        const float* pre_coef; /* 8-wave kernel: x stands for relu(x * pre_coef[c] + pre_coef[64 + c]) (PreBn above), or NULL */ };

__global__ __launch_bounds__(256) void wgrad3x3_c64_kernel(C64WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  constexpr int LDD = CH + 4;
  const int PWc = p.W + 2;
  const int patch_elems = (RBG + 2) * PWc * CH;
  const int MP = (RBG * p.W + 31) / 32 * 32;                 // pixels per block padded to the MFMA K step
  bf16_t* patch = reinterpret_cast<bf16_t*>(smem);
  bf16_t* Dy = patch + patch_elems;                           // [MP][LDD]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, g = lane >> 4, li = lane & 15;
  const int q = li >> 2, pp = li & 3;
  const int rblocks = p.H / RBG, nblocks = p.B * rblocks, npx = RBG * p.W;
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  for (int i = tid; i < (MP - npx) * LDD; i += 256) Dy[npx * LDD + i] = 0;      // padded pixel rows stay zero

  f32x4 acc[4][9];                                            // [n tile][tap], this wave's c tile = wave
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int t = 0; t < 9; ++t) acc[i][t] = f32x4{0.f, 0.f, 0.f, 0.f};

  constexpr int MAXV = 12, MAXD = 8;
  const int nchunks = (RBG + 2) * PWc * 8, ndy = npx * 8;
  const float inv_pw = 1.0f / (float)PWc, inv_w = 1.0f / (float)p.W;
  int s_lds[MAXV], s_rel[MAXV]; unsigned long long s_prow = 0;
#pragma unroll
  for (int i = 0; i < MAXV; ++i) {
    const int id = tid + 256 * i;
    s_lds[i] = -1; s_rel[i] = OOBV;
    if (id < nchunks) {
      const int chunk = id & 7, qq = id >> 3, prow = (int)(((float)qq + 0.5f) * inv_pw), pcol = qq - prow * PWc;
      s_lds[i] = patch_off(prow, pcol, chunk, PWc);
      const int iw = pcol - 1;
      if ((unsigned)iw < (unsigned)p.W) s_rel[i] = (((prow - 1) * p.W + iw) * CH + chunk * 8) * 2;
      s_prow |= (unsigned long long)prow << (4 * i);
    }
  }
  u32x4 pre[MAXV], prd[MAXD];
  auto gload = [&](int blk) {
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBG;
    const int base = ((b * p.H + oh0) * p.W) * CH * 2;
#pragma unroll
    for (int i = 0; i < MAXV; ++i) {
      const int ih = oh0 - 1 + (int)((s_prow >> (4 * i)) & 15);
      const int off = ((unsigned)ih < (unsigned)p.H && s_rel[i] != OOBV) ? base + s_rel[i] : OOBV;
      pre[i] = __builtin_amdgcn_raw_buffer_load_b128(rsX, off, 0, 0);
    }
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
      const int id = tid + 256 * i;
      prd[i] = __builtin_amdgcn_raw_buffer_load_b128(rsY, id < ndy ? base + id * 16 : OOBV, 0, 0);
    }
  };
  auto sstore = [&]() {
#pragma unroll
    for (int i = 0; i < MAXV; ++i)
      if (s_lds[i] >= 0) *reinterpret_cast<u32x4*>(patch + s_lds[i]) = pre[i];
#pragma unroll
    for (int i = 0; i < MAXD; ++i) {
      const int id = tid + 256 * i;
      if (id < ndy) {
        uint32_t* d = reinterpret_cast<uint32_t*>(&Dy[(id >> 3) * LDD + (id & 7) * 8]);   // 136-byte rows: 8-byte aligned
        d[0] = prd[i][0]; d[1] = prd[i][1]; d[2] = prd[i][2]; d[3] = prd[i][3];
      }
    }
  };

  typedef __attribute__((ext_vector_type(8))) short i16x8;
  int blk = blockIdx.x;
  if (blk < nblocks) gload(blk);
  for (; blk < nblocks; blk += gridDim.x) {
    __syncthreads();                                          // previous block's MFMA reads are done
    sstore();
    __syncthreads();
    if (blk + (int)gridDim.x < nblocks) gload(blk + gridDim.x);
    for (int ks = 0; ks < MP / 32; ++ks) {
      bf16x8 af[4];
      const bf16_t* yb = Dy + (ks * 32 + 8 * g + q) * LDD + 4 * pp;
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(yb + i * 16));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(yb + 4 * LDD + i * 16));
        i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
      // pixel rows this lane addresses for the transposed reads (clamped inside the block for the zero-padded tail)
      int pxa = ks * 32 + 8 * g + q, pxb = pxa + 4;
      pxa = pxa < npx ? pxa : npx - 1; pxb = pxb < npx ? pxb : npx - 1;
      const int ra = (int)(((float)pxa + 0.5f) * inv_w), ca = pxa - ra * p.W, rb = (int)(((float)pxb + 0.5f) * inv_w), cb = pxb - rb * p.W;
      const int chunk = wave * 2 + (pp >> 1), sub = (pp & 1) * 4;             // columns 16*wave + 4*pp .. +3
#pragma unroll
      for (int t = 0; t < 9; ++t) {
        const int r = t / 3, s = t - r * 3;
        const bf16_t* a0 = patch + ((((ra + r) * PWc + ca + s) * 8 + (chunk ^ ((ca + s) & 7))) * 8 + sub);
        const bf16_t* a1 = patch + ((((rb + r) * PWc + cb + s) * 8 + (chunk ^ ((cb + s) & 7))) * 8 + sub);
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)a0);
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)a1);
        i16x8 tt = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8 bfv = __builtin_bit_cast(bf16x8, tt);
#pragma unroll
        for (int i = 0; i < 4; ++i) acc[i][t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfv, acc[i][t], 0, 0, 0);
      }
    }
  }
  // flush: D[i = n][j = c] per tap -> dw[n][(tap*64 + c)].  With a workspace every workgroup stores its partial dW into its own
  // slab (vqa_slab_reduce adds the slabs in a fixed order: bit-reproducible, and 512 x 147 KB of float atomics were a quarter of
  // this kernel's time); without one, fp32 atomics.
  float* slab = p.ws ? p.ws + (size_t)blockIdx.x * 64 * 576 : nullptr;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) {
        const size_t o = (size_t)(i * 16 + g * 4 + rr) * 576 + t * 64 + wave * 16 + li;
        if (slab) slab[o] = acc[i][t][rr];
        else atomicAdd(p.dw + o, acc[i][t][rr]);
      }
}

// ------------------------------------------------------------------------------------------------
// weight gradient, 8-wave LDS-DMA form.  Same contraction as wgrad3x3_c64_kernel; what differs:
//   * both operands of a block (4 output rows: the 6 x (W+2) input patch and the 4 x W tile of dY) arrive by LDS-DMA into a
//     double buffer, issued one block ahead -- no register staging (the 4-wave kernel spends 80 VGPRs and two barriers per block
//     on it, and its LDS stores are not overlapped with the MFMAs);
//   * 8 waves: wave = (c tile of 16 input channels) x (half of the 36 (tap, n tile) accumulators): taps 0-3 + half of tap 4,
//     or the other half of tap 4 + taps 5-8.  18 accumulators (72 VGPRs) per wave, no cross-wave reduction, equal work;
//   * both LDS images are XOR-swizzled on the DMA SOURCE side with f(i) = (i & 7) ^ ((i >> 3 & 1) << 2), i = pixel index (dY) or
//     patch column (x): the transposed reads of a half-wave touch pixels {p..p+3, p+8..p+11}, which the 4-wave kernel's
//     (col & 7) swizzle maps onto the same banks twice.
// ------------------------------------------------------------------------------------------------
namespace {
// RBW = output rows per block: 4 where both operands of two blocks fit the LDS (W <= 62), 2 for wider maps (the 96 x 96 stress shape)
__device__ __forceinline__ int swz16(int i) { return (i & 7) ^ (((i >> 3) & 1) << 2); }

template <class F, int... I>
__device__ __forceinline__ void static_for(F&& f, std::integer_sequence<int, I...>) { (f(std::integral_constant<int, I>{}), ...); }
typedef __attribute__((ext_vector_type(2))) int i32x2_c64;
typedef __attribute__((ext_vector_type(4))) int i32x4v_c64;

template <int KH, int WC, int RBW>                             // WC: compile-time image width (address math folds), 0 = runtime
__device__ __forceinline__ void wgrad_c64p_body(const C64WgradParams& p, char* smem, unsigned lds0) {
  typedef __attribute__((ext_vector_type(8))) short i16x8;
  const int Wd = WC ? WC : p.W;
  const int PWc = Wd + 2;
  const int patch_bytes = (RBW + 2) * PWc * 128, npx = RBW * Wd, dy_bytes = ((npx + 31) / 32 * 32) * 128;
  const int buf_bytes = patch_bytes + dy_bytes;
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cw = wave & 3;
  const int rblocks = p.H / RBW, nblocks = p.B * rblocks, nks = (npx + 31) / 32;
  const unsigned long long xa = (unsigned long long)p.x, ya = (unsigned long long)p.dy;
  const i32x4_c64 rsX = {(int)(unsigned)xa, (int)((unsigned)(xa >> 32) & 0xffffu), (int)p.x_bytes, 0x00020000};
  const i32x4_c64 rsY = {(int)(unsigned)ya, (int)((unsigned)(ya >> 32) & 0xffffu), (int)p.dy_bytes, 0x00020000};

  // ---- halo columns of both patch buffers: zero, once (the DMA pieces only ever write columns 1 .. W)
  for (int i = tid; i < 2 * (RBW + 2) * 2 * 8; i += 512) {
    const int chunk = i & 7, side = (i >> 3) & 1, prow = (i >> 4) % (RBW + 2), buf = (i >> 4) / (RBW + 2);
    *reinterpret_cast<u32x4*>(smem + buf * buf_bytes + ((prow * PWc + (side ? PWc - 1 : 0)) * 8 + chunk) * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  // ---- DMA plan.  x: piece (prow, j) = pixels 8j .. 8j+7 of image row oh0 - 1 + prow -> patch columns 1 + 8j ..;
  //      dY: piece j = pixels 8j .. 8j+7 of the block (rows beyond the block's pixels: out of range -> zeros).
  const int ppr = Wd >> 3, nxp = (RBW + 2) * ppr, nyp = nks * 4, npieces = nxp + nyp;
  const int dpx = lane >> 3, dpos = lane & 7;
  auto issue = [&](int blk, int buf) {
    const int b = blk / rblocks, oh0 = (blk - b * rblocks) * RBW;
    const unsigned base = lds0 + (unsigned)(buf * buf_bytes);
    for (int id = wave; id < npieces; id += 8) {
      if (id < nxp) {
        const int prow = id / ppr, j = id - prow * ppr, ih = oh0 - 1 + prow;
        const bool ok = (unsigned)ih < (unsigned)p.H;
        const int pc = 1 + 8 * j + dpx;
        dma16_c64(rsX, base + (unsigned)((prow * PWc + 1 + 8 * j) * 128), ok ? dpx * 128 + ((dpos ^ swz16(pc)) << 4) : OOBV,
                  ok ? ((b * p.H + ih) * Wd + 8 * j) * 128 : 0);
      } else {
        const int j = id - nxp, px = 8 * j + dpx;
        dma16_c64(rsY, base + (unsigned)(patch_bytes + j * 1024), px < npx ? dpx * 128 + ((dpos ^ swz16(px)) << 4) : OOBV,
                  ((b * p.H + oh0) * Wd + 8 * j) * 128);
      }
    }
  };

  f32x4 acc[18];
#pragma unroll
  for (int a = 0; a < 18; ++a) acc[a] = f32x4{0.f, 0.f, 0.f, 0.f};
  const float inv_w = 1.0f / (float)Wd;
  const int xchunk = cw * 2 + (pp >> 1), sub = (pp & 1) * 4;           // this lane's 4 channels of the wave's c tile

  // ---- WC == 56: absolute LDS byte addresses of every transposed read, held in registers (they follow the buffer toggle).
  //      x: [ks][pixel row a / b][tap column s], tap row r is the instruction's immediate offset (r * 58 * 128);
  //      dY: [a / b][n tile i] -- f(pixel) does not depend on ks here (32 pixels per step), ks is the immediate (ks * 4096).
  constexpr int NKS = WC ? RBW * WC / 32 : 1;
  unsigned xo[NKS][2][3], yo[2][4];
  if constexpr (WC != 0) {
#pragma unroll
    for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
      for (int ab = 0; ab < 2; ++ab) {
        const int px = ks * 32 + 8 * g + q + 4 * ab, rr = px / WC, cc = px - rr * WC;
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_)
          xo[ks][ab][s_] = lds0 + (unsigned)(((rr * (WC + 2) + cc + s_) * 128) + ((xchunk ^ swz16(cc + s_)) << 4) + sub * 2);
      }
#pragma unroll
    for (int ab = 0; ab < 2; ++ab) {
      const int px = 8 * g + q + 4 * ab;
#pragma unroll
      for (int i = 0; i < 4; ++i)
        yo[ab][i] = lds0 + (unsigned)(patch_bytes + px * 128 + (((2 * i + (pp >> 1)) ^ swz16(px)) << 4) + sub * 2);
    }
  }

  float* cf = reinterpret_cast<float*>(smem + 2 * buf_bytes);   // [2][64] scale | shift of x's BatchNorm (pre_coef != NULL)
  int blk = blockIdx.x, buf = 0;
  if (blk < nblocks) issue(blk, 0);
  if (p.pre_coef) pre_bn_coef(1, p.pre_coef, BnAcc{}, 0.0, 0.0, 0.f, 0.f, cf, tid);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; blk < nblocks; blk += gridDim.x, buf ^= 1) {
    const int nxt = blk + gridDim.x;
    if (nxt < nblocks && !(p.dbg & 4)) issue(nxt, buf ^ 1);     // the other buffer: every wave finished reading it at the last barrier
    if (p.pre_coef) {                                           // normalise + ReLU the landed x patch in place (both wave groups take part)
      const int bq = blk / rblocks;
      patch_bn_relu<true, RBW + 2>(smem + buf * buf_bytes, Wd, PWc, (blk - bq * rblocks) * RBW - 1, p.H, cf, tid);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    if constexpr (WC != 0) {
      if (p.dbg & 2) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); continue; }
      // Hand-issued transposed reads, counted lgkmcnt waits (LDS returns in order).  35 steps = 7 k-steps x 5 taps; at the top of
      // step n the wave issues group G_n = the x fragment of step n+2 (2 reads) + (taps 0-3) two of the eight dY reads of the
      // NEXT k-step; step n then needs G_(n-2) complete, i.e. at most |G_(n-1)| + |G_n| reads outstanding.  Left to the compiler
      // the reads sit one tap ahead with lgkmcnt(0) every other tap: the LDS latency is exposed ~5x per k-step.
      constexpr int NST = NKS * 5, ROWB = (WC + 2) * 128;
      i32x2_c64 xl[3], xh[3], al[2][4], ah[2][4];
#define TRRD(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
      // prologue: dY fragments of k-step 0, x fragments of steps 0 and 1
#pragma unroll
      for (int i = 0; i < 4; ++i) { TRRD(al[0][i], yo[0][i], 0); TRRD(ah[0][i], yo[1][i], 0); }
      static_for([&xl, &xh, &xo](auto N_) {
        constexpr int n = decltype(N_)::value, t = KH ? 4 + n : n, r = t / 3, s_ = t - r * 3;
        TRRD(xl[n], xo[0][0][s_], r * ROWB); TRRD(xh[n], xo[0][1][s_], r * ROWB);
      }, std::make_integer_sequence<int, 2>{});
      static_for([&xl, &xh, &al, &ah, &xo, &yo, &acc](auto N_) {
        constexpr int n = decltype(N_)::value, ks = n / 5, tt = n - ks * 5;
        // ---- G_n
        {
          constexpr int m = n + 2 < NST ? n + 2 : n, mk = m / 5, mt = m - mk * 5, t = KH ? 4 + mt : mt, r = t / 3, s_ = t - r * 3;
          TRRD(xl[(n + 2) % 3], xo[mk][0][s_], r * ROWB); TRRD(xh[(n + 2) % 3], xo[mk][1][s_], r * ROWB);      // (last two steps: dummy re-read)
        }
        if constexpr (tt < 4) {
          constexpr int nk = ks + 1 < NKS ? ks + 1 : ks, nb = (ks + 1) & 1;                                      // (last k-step: dummy)
          TRRD(al[nb][tt], yo[0][tt], nk * 4096); TRRD(ah[nb][tt], yo[1][tt], nk * 4096);
        }
        // ---- wait for G_(n-2) (and, at tt == 0, for this k-step's dY fragments, issued before it)
        constexpr int size_n = tt < 4 ? 4 : 2, size_p = n == 0 ? 2 : ((tt + 4) % 5 < 4 ? 4 : 2);
        if constexpr (tt == 0)
          asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(xl[n % 3]), "+v"(xh[n % 3]), "+v"(al[ks & 1][0]), "+v"(al[ks & 1][1]), "+v"(al[ks & 1][2]),
                       "+v"(al[ks & 1][3]), "+v"(ah[ks & 1][0]), "+v"(ah[ks & 1][1]), "+v"(ah[ks & 1][2]), "+v"(ah[ks & 1][3]) : "n"(size_n + size_p));
        else
          asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xl[n % 3]), "+v"(xh[n % 3]) : "n"(size_n + size_p));
        const i32x4v_c64 xv = {xl[n % 3][0], xl[n % 3][1], xh[n % 3][0], xh[n % 3][1]};
        const bf16x8 bfv = __builtin_bit_cast(bf16x8, xv);
        constexpr int t = KH ? 4 + tt : tt;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (t == 4 && (KH ? i < 2 : i >= 2)) continue;
          const int a = KH ? (t == 4 ? i - 2 : 2 + 4 * (t - 5) + i) : (t == 4 ? 16 + i : 4 * t + i);
          const i32x4v_c64 av = {al[ks & 1][i][0], al[ks & 1][i][1], ah[ks & 1][i][0], ah[ks & 1][i][1]};
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), bfv, acc[a], 0, 0, 0);
        }
      }, std::make_integer_sequence<int, NST>{});
#undef TRRD
      // the addresses follow the buffer toggle
      const unsigned d = buf ? (unsigned)(-buf_bytes) : (unsigned)buf_bytes;
#pragma unroll
      for (int ks = 0; ks < NKS; ++ks)
#pragma unroll
        for (int ab = 0; ab < 2; ++ab)
#pragma unroll
          for (int s_ = 0; s_ < 3; ++s_) xo[ks][ab][s_] += d;
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int i = 0; i < 4; ++i) yo[ab][i] += d;
      asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next block's pieces landed (and the dummy reads retired)
      __builtin_amdgcn_s_barrier();
      continue;
    }
    const bf16_t* patch = reinterpret_cast<const bf16_t*>(smem + buf * buf_bytes);
    const bf16_t* Dy = reinterpret_cast<const bf16_t*>(smem + buf * buf_bytes + patch_bytes);
    constexpr int KS_UNROLL = WC ? RBW * WC / 32 : 1;
#pragma unroll KS_UNROLL
    for (int ks = 0; ks < (WC ? RBW * WC / 32 : nks); ++ks) {
      // dY fragments (A operand: 16 output channels x 32 pixels), all four n tiles
      const int pa = ks * 32 + 8 * g + q, pb = pa + 4;
      bf16x8 af[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const int ch = 2 * i + (pp >> 1);
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(Dy + pa * 64 + ((ch ^ swz16(pa)) << 3) + sub));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(Dy + pb * 64 + ((ch ^ swz16(pb)) << 3) + sub));
        i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        af[i] = __builtin_bit_cast(bf16x8, t);
      }
      // patch coordinates of the two pixel rows this lane addresses (clamped inside the block for the zero-padded tail)
      const int pxa = pa < npx ? pa : npx - 1, pxb = pb < npx ? pb : npx - 1;
      const int ra = (int)(((float)pxa + 0.5f) * inv_w), ca = pxa - ra * Wd, rb = (int)(((float)pxb + 0.5f) * inv_w), cb = pxb - rb * Wd;
#pragma unroll
      for (int tt = 0; tt < 5; ++tt) {
        const int t = KH ? 4 + tt : tt, r = t / 3, s = t - r * 3;
        const int cola = ca + s, colb = cb + s;
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(patch + ((ra + r) * PWc + cola) * 64 + ((xchunk ^ swz16(cola)) << 3) + sub));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(patch + ((rb + r) * PWc + colb) * 64 + ((xchunk ^ swz16(colb)) << 3) + sub));
        i16x8 t8 = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        const bf16x8 bfv = __builtin_bit_cast(bf16x8, t8);
        // accumulator index of (tap t, n tile i): KH 0: taps 0-3 -> 4t+i, tap 4 (i < 2) -> 16+i; KH 1: tap 4 (i >= 2) -> i-2, taps 5-8 -> 2+4(t-5)+i
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          if (t == 4 && (KH ? i < 2 : i >= 2)) continue;
          const int a = KH ? (t == 4 ? i - 2 : 2 + 4 * (t - 5) + i) : (t == 4 ? 16 + i : 4 * t + i);
          acc[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfv, acc[a], 0, 0, 0);
        }
      }
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next block's pieces landed
    __builtin_amdgcn_s_barrier();
  }
  // flush: D[i = n][j = c] per tap -> slab[n][(tap*64 + c)] (one slab per workgroup, reduced in slab order by vqa_slab_reduce)
  float* slab = p.ws + (size_t)blockIdx.x * 64 * 576;
#pragma unroll
  for (int tt = 0; tt < 5; ++tt) {
    const int t = KH ? 4 + tt : tt;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (t == 4 && (KH ? i < 2 : i >= 2)) continue;
      const int a = KH ? (t == 4 ? i - 2 : 2 + 4 * (t - 5) + i) : (t == 4 ? 16 + i : 4 * t + i);
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) slab[(size_t)(i * 16 + g * 4 + rr) * 576 + t * 64 + cw * 16 + li] = acc[a][rr];
    }
  }
}
}

template <int RBW>
__global__ __launch_bounds__(512, 2) void wgrad3x3_c64p_kernel(C64WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  if constexpr (RBW == 4) {
    if (p.W == 56) {                                          // the model's stage-1 width: fully unrolled, constant addressing
      if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8)) wgrad_c64p_body<1, 56, 4>(p, smem, lds0);  // waves 4-7
      else wgrad_c64p_body<0, 56, 4>(p, smem, lds0);                                                    // waves 0-3
      return;
    }
  }
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 8)) wgrad_c64p_body<1, 0, RBW>(p, smem, lds0);
  else wgrad_c64p_body<0, 0, RBW>(p, smem, lds0);
}

// ------------------------------------------------------------------------------------------------
// weight gradient of the stage-2 convs (3x3 / 1, 128 -> 128 channels, 28 x 28), same structure as wgrad3x3_c64p_kernel:
// the generic split-K tile kernel runs these at 0.72 PFLOP/s (112 splits of 9 tiles, 66 MB of slabs); here a workgroup keeps a
// 128 (output channels) x 9 taps x 64 (one HALF of the input channels) block of dW in its accumulators for the whole launch
// (8 waves = 4 input-channel tiles x 2 output-channel halves, 36 accumulators = 144 VGPRs per wave) and walks (image, 4 rows) blocks:
// the 6 x 30 x 64 slice of X and the 112 x 128 tile of dY arrive by LDS-DMA into a double buffer one block ahead.  Two workgroup
// populations (input-channel half = blockIdx & 1) of 128 persistent workgroups each.  W = 28 is not a multiple of the 8-pixel DMA
// piece: the patch rows have a pitch of 34 pixels and the last piece of a row fetches pixels 24..31 with the lanes beyond the image
// out of range -> zeros, which also writes the right halo column.  dY rows are 256 bytes (the whole bank window), so its transposed
// reads are made conflict free by XORing the 32-byte slot index with g(p) = (p & 3) | ((p >> 3 & 1) << 2) on the DMA source side.
// Reads are hand-issued two taps ahead with counted lgkmcnt waits, all addresses precomputed (compile-time geometry).
// ------------------------------------------------------------------------------------------------
namespace {
constexpr int C128_W = 28, C128_RB = 4, C128_PW = 34, C128_NPX = C128_RB * C128_W, C128_NKS = 4;
constexpr int C128_PATCH = (C128_RB + 2) * C128_PW * 128, C128_DY = 128 * 256, C128_BUF = C128_PATCH + C128_DY;
__device__ __forceinline__ int gsw8(int p) { return (p & 3) | (((p >> 3) & 1) << 2); }
}
struct C128WgradParams { const bf16_t* x; const bf16_t* dy; float* ws; int B; unsigned bytes; };

__global__ __launch_bounds__(512, 2) void wgrad3x3_c128p_kernel(C128WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), cw = wave & 3, nh = wave >> 2;
  const int half = blockIdx.x & 1, wg = blockIdx.x >> 1, nwg = gridDim.x >> 1;
  const int nblocks = p.B * (C128_W / C128_RB);
  const unsigned long long xa = (unsigned long long)p.x, ya = (unsigned long long)p.dy;
  const i32x4_c64 rsX = {(int)(unsigned)xa, (int)((unsigned)(xa >> 32) & 0xffffu), (int)p.bytes, 0x00020000};
  const i32x4_c64 rsY = {(int)(unsigned)ya, (int)((unsigned)(ya >> 32) & 0xffffu), (int)p.bytes, 0x00020000};

  // ---- once: left halo column of every patch row and the 16 padding pixel rows of the dY tile (never written by the DMA)
  for (int i = tid; i < 2 * (C128_RB + 2) * 8; i += 512) {
    const int chunk = i & 7, prow = (i >> 3) % (C128_RB + 2), buf = (i >> 3) / (C128_RB + 2);
    *reinterpret_cast<u32x4*>(smem + buf * C128_BUF + (prow * C128_PW) * 128 + chunk * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  for (int i = tid; i < 2 * 16 * 16; i += 512) {
    const int buf = i >> 8, o = i & 255;
    *reinterpret_cast<u32x4*>(smem + buf * C128_BUF + C128_PATCH + C128_NPX * 256 + o * 16) = u32x4{0u, 0u, 0u, 0u};
  }
  // ---- DMA plan: 24 x pieces (6 patch rows x 4 pieces of 8 pixels x 128 B) + 28 dY pieces (4 pixels x 256 B)
  auto issue = [&](int blk, int buf) {
    const int b = blk / (C128_W / C128_RB), oh0 = (blk - b * (C128_W / C128_RB)) * C128_RB;
    const unsigned base = lds0 + (unsigned)(buf * C128_BUF);
    for (int id = wave; id < 24 + 28; id += 8) {
      if (id < 24) {
        const int prow = id >> 2, j = id & 3, ih = oh0 - 1 + prow;
        const int dpx = lane >> 3, dpos = lane & 7, px = 8 * j + dpx;
        const bool ok = (unsigned)ih < (unsigned)C128_W && px < C128_W;
        dma16_c64(rsX, base + (unsigned)((prow * C128_PW + 1 + 8 * j) * 128), ok ? dpx * 256 + ((dpos ^ swz16(1 + px)) << 4) : OOBV,
                  (unsigned)ih < (unsigned)C128_W ? ((b * C128_W + ih) * C128_W + 8 * j) * 256 + half * 128 : 0);
      } else {
        const int j = id - 24, pxi = lane >> 4, c16 = lane & 15, px = 4 * j + pxi;
        dma16_c64(rsY, base + (unsigned)(C128_PATCH + j * 1024), pxi * 256 + ((c16 ^ (gsw8(px) << 1)) << 4),
                  ((b * C128_W + oh0) * C128_W + 4 * j) * 256);
      }
    }
  };

  f32x4 acc[9][4];
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i) acc[t][i] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int xchunk = cw * 2 + (pp >> 1), sub = (pp & 1) * 4;

  // ---- absolute LDS byte addresses of every transposed read (they follow the buffer toggle)
  unsigned xo[C128_NKS][2][3], yo[2][4];
#pragma unroll
  for (int ks = 0; ks < C128_NKS; ++ks)
#pragma unroll
    for (int ab = 0; ab < 2; ++ab) {
      int px = ks * 32 + 8 * g + q + 4 * ab;
      px = px < C128_NPX ? px : C128_NPX - 1;                   // padded tail: its dY rows are zero, any valid address will do
      const int rr = px / C128_W, cc = px - rr * C128_W;
#pragma unroll
      for (int s_ = 0; s_ < 3; ++s_)
        xo[ks][ab][s_] = lds0 + (unsigned)(((rr * C128_PW + cc + s_) * 128) + ((xchunk ^ swz16(cc + s_)) << 4) + sub * 2);
    }
#pragma unroll
  for (int ab = 0; ab < 2; ++ab) {
    const int px = 8 * g + q + 4 * ab;
#pragma unroll
    for (int i = 0; i < 4; ++i)
      yo[ab][i] = lds0 + (unsigned)(C128_PATCH + px * 256 + (((8 * nh + 2 * i + (pp >> 1)) ^ (gsw8(px) << 1)) << 4) + sub * 2);
  }

  int blk = wg, buf = 0;
  if (blk < nblocks) issue(blk, 0);
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();
  for (; blk < nblocks; blk += nwg, buf ^= 1) {
    const int nxt = blk + nwg;
    if (nxt < nblocks) issue(nxt, buf ^ 1);                     // the other buffer: every wave finished reading it at the last barrier
    // 36 steps = 4 k-steps x 9 taps.  G_n (issued at the top of step n) = one of the eight dY reads of the NEXT k-step (taps 0-7), then the
    // x fragment of step n+2 (2 reads); LDS returns in order, so step n needs at most |G_(n-1)| + |G_n| reads outstanding.
    constexpr int NST = C128_NKS * 9, ROWB = C128_PW * 128;
    i32x2_c64 xl[3], xh[3], al[2][4], ah[2][4];
#define TRRD(dst, addr, imm) asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(dst) : "v"(addr), "n"(imm))
#pragma unroll
    for (int i = 0; i < 4; ++i) { TRRD(al[0][i], yo[0][i], 0); TRRD(ah[0][i], yo[1][i], 0); }
    static_for([&xl, &xh, &xo](auto N_) {
      constexpr int n = decltype(N_)::value, r = n / 3, s_ = n - r * 3;
      TRRD(xl[n], xo[0][0][s_], r * ROWB); TRRD(xh[n], xo[0][1][s_], r * ROWB);
    }, std::make_integer_sequence<int, 2>{});
    static_for([&xl, &xh, &al, &ah, &xo, &yo, &acc](auto N_) {
      constexpr int n = decltype(N_)::value, ks = n / 9, t = n - ks * 9;
      if constexpr (t < 8) {                                    // dY read t of the next k-step: n tile t >> 1, pixel row a / b = t & 1
        constexpr int nk = ks + 1 < C128_NKS ? ks + 1 : ks, nb = (ks + 1) & 1;       // (last k-step: dummy re-read)
        if constexpr ((t & 1) == 0) TRRD(al[nb][t >> 1], yo[0][t >> 1], nk * 8192); else TRRD(ah[nb][t >> 1], yo[1][t >> 1], nk * 8192);
      }
      {
        constexpr int m = n + 2 < NST ? n + 2 : n, mk = m / 9, mt = m - mk * 9, r = mt / 3, s_ = mt - r * 3;
        TRRD(xl[(n + 2) % 3], xo[mk][0][s_], r * ROWB); TRRD(xh[(n + 2) % 3], xo[mk][1][s_], r * ROWB);          // (last two steps: dummy re-read)
      }
      constexpr int size_n = t < 8 ? 3 : 2, size_p = n == 0 ? 2 : ((t + 8) % 9 < 8 ? 3 : 2);
      if constexpr (t == 0)
        asm volatile("s_waitcnt lgkmcnt(%10)" : "+v"(xl[n % 3]), "+v"(xh[n % 3]), "+v"(al[ks & 1][0]), "+v"(al[ks & 1][1]), "+v"(al[ks & 1][2]),
                     "+v"(al[ks & 1][3]), "+v"(ah[ks & 1][0]), "+v"(ah[ks & 1][1]), "+v"(ah[ks & 1][2]), "+v"(ah[ks & 1][3]) : "n"(size_n + size_p));
      else
        asm volatile("s_waitcnt lgkmcnt(%2)" : "+v"(xl[n % 3]), "+v"(xh[n % 3]) : "n"(size_n + size_p));
      const i32x4v_c64 xv = {xl[n % 3][0], xl[n % 3][1], xh[n % 3][0], xh[n % 3][1]};
      const bf16x8 bfv = __builtin_bit_cast(bf16x8, xv);
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const i32x4v_c64 av = {al[ks & 1][i][0], al[ks & 1][i][1], ah[ks & 1][i][0], ah[ks & 1][i][1]};
        acc[t][i] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, av), bfv, acc[t][i], 0, 0, 0);
      }
    }, std::make_integer_sequence<int, NST>{});
#undef TRRD
    const unsigned d = buf ? (unsigned)(-C128_BUF) : (unsigned)C128_BUF;
#pragma unroll
    for (int ks = 0; ks < C128_NKS; ++ks)
#pragma unroll
      for (int ab = 0; ab < 2; ++ab)
#pragma unroll
        for (int s_ = 0; s_ < 3; ++s_) xo[ks][ab][s_] += d;
#pragma unroll
    for (int ab = 0; ab < 2; ++ab)
#pragma unroll
      for (int i = 0; i < 4; ++i) yo[ab][i] += d;
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");   // next block's pieces landed (and the dummy reads retired)
    __builtin_amdgcn_s_barrier();
  }
  // flush: this workgroup's [128 n][9 taps][64 c of its half] block -> slab blockIdx.x
  float* slab = p.ws + (size_t)blockIdx.x * 128 * 576;
#pragma unroll
  for (int t = 0; t < 9; ++t)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int rr = 0; rr < 4; ++rr) slab[(size_t)(nh * 64 + i * 16 + g * 4 + rr) * 576 + t * 64 + cw * 16 + li] = acc[t][i][rr];
}

// dw[n][tap*128 + half*64 + c] += sum over the slabs of that half (slab index half + 2w, w ascending) of slab[n][tap*64 + c]:
// 64 float4 columns x 16 slab groups per block, fixed fold order (bit-reproducible)
__global__ __launch_bounds__(1024) void wgrad_c128p_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nwg) {
  const int half = blockIdx.y, i4 = blockIdx.x * 64 + threadIdx.x, gq = threadIdx.y;      // i4: float4 index inside [128][576]
  const int per = (nwg + 15) / 16, w0 = gq * per, w1 = min(nwg, w0 + per);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  for (int w = w0; w < w1; ++w) a += reinterpret_cast<const f32x4*>(ws + (size_t)(half + 2 * w) * 128 * 576)[i4];
  __shared__ f32x4 sh[16][64];
  sh[gq][threadIdx.x] = a;
  __syncthreads();
  if (gq == 0) {
    const int e = i4 * 4, n = e / 576, k = e - n * 576, tap = k >> 6, c = k & 63;
    f32x4* dst = reinterpret_cast<f32x4*>(dw + (size_t)n * 1152 + tap * 128 + half * 64 + c);
    f32x4 t = *dst;
#pragma unroll
    for (int s_ = 0; s_ < 16; ++s_) t += sh[s_][threadIdx.x];
    *dst = t;
  }
}

extern "C" int vqa_slab_reduce(const float* ws, float* dw, int nslabs, long long n, hipStream_t st);

extern "C" {

// persistent grid size (= rows of the BN statistics slab) or 0 when the shape is unsupported
int vqa_conv3x3_c64_blocks(int B, int H, int W) {
  if (H % RBG || (RBF * W) % 16 || W > 126 || B * (H / RBG) <= 0) return 0;
  if (((RBF + 2) * (W + 2) * 8 + 255) / 256 > 8 || ((RBG + 2) * (W + 2) * 8 + 255) / 256 > 12 || (RBG * W * 8 + 255) / 256 > 8) return 0;
  const int nb = B * (H / RBG);
  return nb < 512 ? nb : 512;
}
// x NHWC bf16 [B][H][W][64]; w [64][(r,s,c)] bf16 (forward: [Cout][R][S][Cin]; data gradient: flipped+transposed pack);
// out NHWC bf16; stats [blocks][2][64] or NULL; out += addend * (addmask > 0) (identity-path gradient) when given.
int vqa_conv3x3_c64(const void* x, const void* w, void* out, float* stats, const void* addend, const void* addmask,
                    int B, int H, int W, hipStream_t st) {
  const int grid = vqa_conv3x3_c64_blocks(B, H, W);
  if (!x || !w || !out || grid <= 0) return VQA_EARG;
  C64Params p;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.out = (bf16_t*)out; p.stats = stats;
  p.addend = (const bf16_t*)addend; p.addmask = (const bf16_t*)addmask; p.outmask = nullptr; p.bn_y = nullptr; p.bn_coef = nullptr; p.bn_facc = nullptr; p.B = B; p.H = H; p.W = W; p.dbg = 0; p.stats_mode = 0; p.pre_mode = 0; p.pre_coef = nullptr;
  const size_t xb = (size_t)B * H * W * CH * 2;
  if (xb >= 0x7fffffffull) return VQA_EARG;
  p.x_bytes = (unsigned)xb;
  const size_t shm = (size_t)2 * (RBF + 2) * (W + 2) * CH * 2 + 4 * 16 * LDE * 2 + 2 * 64 * 2 * 4;
  static size_t attr = 0;
  if (shm > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = shm; }
  hipLaunchKernelGGL(conv3x3_c64_kernel, dim3(grid), dim3(256), shm, st, p);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
// persistent grid of the 8-wave LDS-DMA patch kernel (= rows of its statistics slab), 0 when the shape is unsupported
static int c64p_rows(int H, int W) {         // output rows per block of the 8-wave patch kernel for this shape, 0: unsupported
  for (int rbp = 8; rbp >= 4; rbp -= 4) {
    const size_t shm = (size_t)2 * (rbp + 2) * (W + 2) * CH * 2 + 4 * 64 * 2 * 4 + 4 * 64 * 4;      // (what c64p_launch asks for)
    if (H % rbp == 0 && (rbp * W) % 16 == 0 && shm <= 160 * 1024) return rbp;
  }
  return 0;
}
int vqa_conv3x3_c64p_blocks(int B, int H, int W) {
  const int rbp = (W % 8 || W > 126 || B <= 0 || H <= 0) ? 0 : c64p_rows(H, W);
  if (!rbp) return 0;
  const int nb = B * (H / rbp);
  return nb < 256 ? nb : 256;
}
// forward / addend-free data gradient of the 64 -> 64 channel 3x3 conv with the 8-wave LDS-DMA patch kernel (no epilogue inputs)
static int c64p_launch(C64Params& p, const void* x, const void* w, void* out, float* stats, int B, int H, int W, int stats_mode, hipStream_t st) {
  const int grid = vqa_conv3x3_c64p_blocks(B, H, W);
  if (!x || !w || !out || grid <= 0) return VQA_EARG;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.out = (bf16_t*)out; p.stats = stats;
  p.B = B; p.H = H; p.W = W; p.stats_mode = stats_mode;
  const int dbg_env = vqa_env_int("VQA_C64P_DBG", 0);
  p.dbg = dbg_env;
  const size_t xb = (size_t)B * H * W * CH * 2;
  if (xb >= 0x7fffffffull) return VQA_EARG;
  p.x_bytes = (unsigned)xb;
  const int rbp = c64p_rows(H, W);
  const size_t shm = (size_t)2 * (rbp + 2) * (W + 2) * CH * 2 + 4 * 64 * 2 * 4 + 4 * 64 * 4;
  static size_t attr8 = 0, attr4 = 0;
  static size_t attr[6] = {0, 0, 0, 0, 0, 0};
#define C64P_GO(RB, EP, SLOT) do { auto kfn = conv3x3_c64p_kernel<RB, EP>; \
    if (shm > attr[SLOT]) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kfn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr[SLOT] = shm; } \
    hipLaunchKernelGGL(kfn, dim3(grid), dim3(512), shm, st, p); } while (0)
  if (p.addend && p.bn_y) return VQA_EARG;
  if (p.addend) { if (rbp == 8) C64P_GO(8, 1, 2); else C64P_GO(4, 1, 3); }
  else if (p.bn_y) { if (rbp == 8) C64P_GO(8, 2, 4); else C64P_GO(4, 2, 5); }
  else { if (rbp == 8) C64P_GO(8, 0, 0); else C64P_GO(4, 0, 1); }
#undef C64P_GO
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_conv3x3_c64p(const void* x, const void* w, void* out, float* stats, int B, int H, int W, int stats_mode, hipStream_t st) {
  C64Params p;
  p.pre_mode = 0; p.pre_coef = nullptr; p.addend = p.addmask = p.outmask = nullptr; p.bn_y = nullptr; p.bn_coef = nullptr; p.bn_facc = nullptr;
  return c64p_launch(p, x, w, out, stats, B, H, W, stats_mode, st);
}
// conv2's data gradient da1 = conv(dy2, flipped pack) that also leaves bn1's BatchNorm-backward column sums (C64Params, BNRED): bn_y = conv1's
// raw output y1, bn_coef = bn1's coef[4][64] (scale | shift | mean | invstd), bn_facc = a zeroed vqa_bn_acc_words(3, 64) accumulator.
int vqa_conv3x3_c64p_bnred(const void* x, const void* w, void* out, const void* bn_y, const float* bn_coef, unsigned long long* bn_facc,
                           int B, int H, int W, hipStream_t st) {
  if (!bn_y || !bn_coef || !bn_facc) return VQA_EARG;
  if ((size_t)vqa_conv3x3_c64p_blocks(B, H, W) > (size_t)VQA_ACC_MAX_PARTS) return VQA_EARG;
  C64Params p;
  p.pre_mode = 0; p.pre_coef = nullptr; p.addend = p.addmask = p.outmask = nullptr;
  p.bn_y = (const bf16_t*)bn_y; p.bn_coef = bn_coef; p.bn_facc = bn_facc;
  return c64p_launch(p, x, w, out, nullptr, B, H, W, 0, st);
}
// The data gradient of a residual block's conv1 (w = the flipped + transposed pack) with the block's identity path in the epilogue:
// out = (conv + addend * (addmask > 0)) * (outmask > 0) on the bf16 conv value, the epilogue of vqa_igemm / vqa_conv8p.  addend required;
// addmask / outmask [B*H*W][64] bf16 or NULL.
int vqa_conv3x3_c64p_epi(const void* x, const void* w, void* out, const void* addend, const void* addmask, const void* outmask,
                         int B, int H, int W, hipStream_t st) {
  if (!addend) return VQA_EARG;
  C64Params p;
  p.pre_mode = 0; p.pre_coef = nullptr;
  p.addend = (const bf16_t*)addend; p.addmask = (const bf16_t*)addmask; p.outmask = (const bf16_t*)outmask;
  p.bn_y = nullptr; p.bn_coef = nullptr; p.bn_facc = nullptr;
  return c64p_launch(p, x, w, out, nullptr, B, H, W, 0, st);
}
// The same conv applied to relu(BatchNorm(y)) in TRAINING mode, the normalised tensor never materialised (models/cnn_backbone.py:182-187
// conv1 -> bn1 -> relu -> conv2): y is conv1's raw output, acc its fixed-point statistics (vqa_bn_acc_words(2, 64), filled by the launch
// that produced y with stats_mode = 1).  Every workgroup finalizes the 64 coefficients in its prologue (under its first patch DMA),
// workgroup 0 publishes coef_out[4][64] (scale | shift | mean | invstd: the backward and vqa_wgrad3x3_c64_bn read it) and updates the
// running statistics -- exactly what vqa_bn_apply_acc does, minus its read of y and write of a.  The patch is transformed in LDS.
int vqa_conv3x3_c64p_bn(const void* y, const unsigned long long* acc, const float* gamma, const float* beta, float* running_mean,
                        float* running_var, long long* num_batches_tracked, float* coef_out, const void* w, void* out, float* stats,
                        int B, int H, int W, int stats_mode, double count, float momentum, float eps, hipStream_t st) {
  if (!acc || !gamma || !beta || !coef_out || count <= 0) return VQA_EARG;
  C64Params p;
  p.pre_mode = 2; p.pre_coef = nullptr; p.addend = p.addmask = p.outmask = nullptr; p.bn_y = nullptr; p.bn_coef = nullptr; p.bn_facc = nullptr;
  p.pre = BnAcc{acc, gamma, beta, running_mean, running_var, num_batches_tracked, coef_out};
  p.pre_inv_count = 1.0 / count; p.pre_unbias = count > 1 ? count / (count - 1) : 1.0; p.pre_momentum = momentum; p.pre_eps = eps;
  return c64p_launch(p, y, w, out, stats, B, H, W, stats_mode, st);
}
// dw [64][576] fp32 (+=).  ws: scratch of >= vqa_conv3x3_c64_blocks(B,H,W) * 64*576 floats for the deterministic two-pass
// accumulation (NULL or too small: fp32 atomics)
// Stage-2 shape only (3x3 / 1 / pad 1, 128 -> 128 channels, 28 x 28 maps, bf16): slabs needed (= workgroups) or 0 when unsupported
int vqa_wgrad3x3_c128_blocks(int B, int H, int W) {
  const int en = vqa_env_int("VQA_C128WP", 1);
  if (!en || H != C128_W || W != C128_W || B <= 0 || (size_t)B * H * W * 128 * 2 >= 0x7fffffffull) return 0;
  const int nblocks = B * (C128_W / C128_RB);
  int per_half = nblocks < 128 ? nblocks : 128;
  return 2 * per_half;
}
// dw [128][9*128] fp32 (+=).  ws: vqa_wgrad3x3_c128_blocks(B, H, W) * 128*576 floats.
int vqa_wgrad3x3_c128(const void* x, const void* dy, float* dw, int B, int H, int W, float* ws, long long ws_floats, hipStream_t st) {
  const int grid = vqa_wgrad3x3_c128_blocks(B, H, W);
  if (!x || !dy || !dw || !ws || grid <= 0 || ws_floats < (long long)grid * 128 * 576) return VQA_EARG;
  C128WgradParams p;
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.ws = ws; p.B = B; p.bytes = (unsigned)((size_t)B * H * W * 128 * 2);
  const size_t shm = (size_t)2 * C128_BUF;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_c128p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = true; }
  hipLaunchKernelGGL(wgrad3x3_c128p_kernel, dim3(grid), dim3(512), shm, st, p);
  hipLaunchKernelGGL(wgrad_c128p_reduce_kernel, dim3(128 * 576 / 4 / 64, 2), dim3(64, 16), 0, st, ws, dw, grid / 2);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

// rows per block of the 8-wave LDS-DMA weight-gradient kernel for this shape (4 where both operands of two blocks fit the LDS, else 2), 0: not eligible
static int c64wp_rows(int H, int W, size_t* shm_out) {
  if (W % 8 || W > 126) return 0;
  for (int r = 4; r >= 2; r -= 2) {
    const size_t shm = (size_t)2 * ((size_t)(r + 2) * (W + 2) * 128 + (size_t)((r * W + 31) / 32 * 32) * 128);
    if (H % r == 0 && shm <= 160 * 1024) { if (shm_out) *shm_out = shm; return r; }
  }
  return 0;
}
// slabs ([64][576] floats each) vqa_wgrad3x3_c64 needs in `ws` for this shape; 0: neither kernel takes it
int vqa_wgrad3x3_c64_blocks(int B, int H, int W) {
  if (B <= 0 || H <= 0 || W <= 0) return 0;
  const int g4 = vqa_conv3x3_c64_blocks(B, H, W);
  const int rbw = c64wp_rows(H, W, nullptr);
  const int nblk = rbw ? B * (H / rbw) : 0, gp = nblk < 256 ? nblk : 256;
  return g4 > gp ? g4 : gp;
}
static int c64_wgrad_launch(const void* x, const float* pre_coef, const void* dy, float* dw, int B, int H, int W, float* ws, long long ws_floats,
                            hipStream_t st) {
  const int grid = vqa_conv3x3_c64_blocks(B, H, W);            // the 4-wave kernel's persistent grid (0: it does not take the shape)
  size_t shm_p = 0;
  const int rbw = c64wp_rows(H, W, &shm_p);
  const int nblk_p = rbw ? B * (H / rbw) : 0, grid_p = nblk_p < 256 ? nblk_p : 256;
  if (!x || !dy || !dw || (grid <= 0 && grid_p <= 0)) return VQA_EARG;
  C64WgradParams p;
  p.x = (const bf16_t*)x; p.dy = (const bf16_t*)dy; p.dw = dw; p.B = B; p.H = H; p.W = W; p.pre_coef = pre_coef;
  const int dbg_env = vqa_env_int("VQA_C64WP_DBG", 0);
  p.dbg = dbg_env;
  const size_t xb = (size_t)B * H * W * CH * 2;
  if (xb >= 0x7fffffffull) return VQA_EARG;
  p.x_bytes = (unsigned)xb; p.dy_bytes = (unsigned)xb;
  // 8-wave LDS-DMA form (needs the slab workspace: it has no atomic flush); VQA_C64WP=0 keeps the 4-wave kernel (measurement)
  const int wp_env = vqa_env_int("VQA_C64WP", 1);
  if (wp_env && ws && grid_p > 0 && ws_floats >= (long long)grid_p * 64 * 576) {
    p.ws = ws;
    shm_p += 2 * 64 * 4;                                        // cf[2][64] behind the two buffers
    static size_t attr4 = 0, attr2 = 0;
    if (rbw == 4) {
      if (shm_p > attr4) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_c64p_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_p); attr4 = shm_p; }
      hipLaunchKernelGGL(wgrad3x3_c64p_kernel<4>, dim3(grid_p), dim3(512), shm_p, st, p);
    } else {
      if (shm_p > attr2) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_c64p_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm_p); attr2 = shm_p; }
      hipLaunchKernelGGL(wgrad3x3_c64p_kernel<2>, dim3(grid_p), dim3(512), shm_p, st, p);
    }
    VQA_LAUNCH_CHECK();
    return vqa_slab_reduce(p.ws, dw, grid_p, 64 * 576, st);
  }
  if (grid <= 0 || pre_coef) return VQA_EARG;                   // (the 4-wave kernel has no BatchNorm prologue)
  p.ws = (ws && ws_floats >= (long long)grid * 64 * 576) ? ws : nullptr;
  const int MP = (RBG * W + 31) / 32 * 32;
  const size_t shm = (size_t)(RBG + 2) * (W + 2) * CH * 2 + (size_t)MP * (CH + 4) * 2;
  static size_t attr = 0;
  if (shm > attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad3x3_c64_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)shm); attr = shm; }
  hipLaunchKernelGGL(wgrad3x3_c64_kernel, dim3(grid), dim3(256), shm, st, p);
  VQA_LAUNCH_CHECK();
  return p.ws ? vqa_slab_reduce(p.ws, dw, grid, 64 * 576, st) : VQA_OK;
}
int vqa_wgrad3x3_c64_bn_ok(int B, int H, int W) {               // 1: vqa_wgrad3x3_c64_bn takes this shape (the 8-wave kernel does)
  return B > 0 && H > 0 && c64wp_rows(H, W, nullptr) > 0 && vqa_env_int("VQA_C64WP", 1) && (size_t)B * H * W * CH * 2 < 0x7fffffffull;
}
int vqa_wgrad3x3_c64(const void* x, const void* dy, float* dw, int B, int H, int W, float* ws, long long ws_floats, hipStream_t st) {
  return c64_wgrad_launch(x, nullptr, dy, dw, B, H, W, ws, ws_floats, st);
}
// dw += dy^T * gather(relu(y * coef[c] + coef[64 + c])): the weight gradient of the conv that vqa_conv3x3_c64p_bn ran on the
// un-materialised activation; coef = the [4][64] table that launch published.  8-wave kernel only (needs the slab workspace).
int vqa_wgrad3x3_c64_bn(const void* y, const float* coef, const void* dy, float* dw, int B, int H, int W, float* ws, long long ws_floats,
                        hipStream_t st) {
  if (!coef) return VQA_EARG;
  return c64_wgrad_launch(y, coef, dy, dw, B, H, W, ws, ws_floats, st);
}

}  // extern "C"
