// Implicit-GEMM convolution / linear kernels for gfx950 (MI355X), MFMA 16x16 tiles, 64-lane waves.
//
//   igemm_kernel  : out[M][N] = gather(A)[M][K] * W[N][K]^T  (+bias, relu, +addend*(mask>0), BN partial stats)
//                   - forward conv over NHWC activations (models/cnn_backbone.py:182-195 conv1/conv2, :243-247 shortcut)
//                   - data-gradient of the same convs (transposed gather)
//                   - every nn.Linear of the token side as a 1x1 "conv" (models/text_encoder.py:219-221,320-324 ...)
//                   - the 7x7/2 stem conv straight from the NCHW fp32 image (models/cnn_backbone.py:349-350)
//   wgrad_kernel  : dW[N][K] += dY[M][N]^T * gather(A)[M][K]   (split over M, fp32 atomics)
//
// T = float  -> v_mfma_f32_16x16x4_f32  (exact fp32 fma chain; parity path)
// T = bf16   -> v_mfma_f32_16x16x32_bf16 (fp32 accumulate; throughput path)
// LDS tiles hold 128 bytes of K per row (+16 B pad): [row][k] for igemm, [k][row] for wgrad
// (wgrad reads its operands with ds_read_b64_tr_b16 so the contraction index ends up lane-contiguous).
#include "common.h"
#include <stdlib.h>

enum { LOADER_NHWC = 0, LOADER_STEM = 1, LOADER_DGRAD2 = 2 };

struct IGemmParams {
  const void* a; const void* w; void* out;
  const float* bias; const void* addend; const void* addmask; float* stats;
  const void* outmask;       // != nullptr: out = (... + addend) * (outmask > 0): the consumer's ReLU mask applied by the producer
  float out_scale;           // with outmask: kept elements are multiplied by this (1 / (1 - p) of a ReLU + dropout whose output IS the mask)
  int stats_mode;            // 0: stats is a float slab [tiles][2][N]; 1: a fixed-point accumulator u64 [vqa_bn_acc_words(2, N)] (common.h acc_add_fixed)
  int M, N, Kp, Kw;          // Kp: reduction length rounded up to BK; Kw: weight row length (elements)
  int B, H, W, C;            // source tensor (NHWC; NCHW fp32 image for the stem loader)
  int Ho, Wo;                // spatial dims of the GEMM rows (M = B*Ho*Wo)
  int R, S, stride, pad, transposed, relu;
  float drop_p; unsigned long long drop_seed;   // dropout applied after bias/relu, before the addend
  // LOADER_DGRAD2 (data gradient of a stride-2 conv, rows grouped by output parity class so only valid taps are issued):
  const void* a2;                               // second source (dY of the 1x1 shortcut), same geometry as a
  unsigned a_bytes, a2_bytes, w_bytes;          // buffer extents for the hardware range check
  int dbg;                                      // measurement switch (VQA_IGEMM_DBG): bit 0 no in-loop DMA, bit 1 no MFMA phase
  unsigned long long mul_howo, mul_wo;          // ceil(2^40 / d) for d = Ho*Wo, Wo (DGRAD2: (Ho/2)*(Wo/2), Wo/2): exact m / d for m*d < 2^40
  int ntaps[4]; int tap_koff[4][5]; int tap_dh[4][5]; int tap_dw[4][5]; int tap_src[4][5];
};

template <typename T> struct GT;
template <> struct GT<float>  { static constexpr int VEC = 4, BK = 32, MK = 4, BKM = 32; };
template <> struct GT<bf16_t> { static constexpr int VEC = 8, BK = 64, MK = 32, BKM = 64; };

struct RowInfo { int pix, ih0, iw0; };

__device__ __forceinline__ int fast_div(int m, unsigned long long mul) { return (int)(((unsigned long long)(unsigned)m * mul) >> 40); }

__device__ __forceinline__ RowInfo decode_row(int m, int M, int HoWo, int Wo, int HW, int stride, int pad, int transposed, bool stem,
                                              unsigned long long mul_howo, unsigned long long mul_wo) {
  RowInfo ri;
  if (m >= M) { ri.pix = -1; ri.ih0 = 0; ri.iw0 = 0; return ri; }
  int b = fast_div(m, mul_howo), rem = m - b * HoWo;
  int oh = fast_div(rem, mul_wo), ow = rem - oh * Wo;
  ri.pix = stem ? b : b * HW;
  if (!transposed) { ri.ih0 = oh * stride - pad; ri.iw0 = ow * stride - pad; }
  else { ri.ih0 = oh + pad; ri.iw0 = ow + pad; }
  return ri;
}

template <typename T>
__device__ __forceinline__ Vec16<T> load_a_nhwc(const T* a, const RowInfo& ri, int r, int s, int c,
                                                int H, int W, int C, int stride, int transposed) {
  bool ok = ri.pix >= 0 && c < C;
  int ih, iw;
  if (!transposed) { ih = ri.ih0 + r; iw = ri.iw0 + s; }
  else {
    int th = ri.ih0 - r, tw = ri.iw0 - s;
    ok = ok && th >= 0 && tw >= 0;
    ih = th / stride; iw = tw / stride;
    ok = ok && (ih * stride == th) && (iw * stride == tw);
  }
  ok = ok && ih >= 0 && ih < H && iw >= 0 && iw < W;
  if (!ok) return zero16<T>();
  return ldg16(a + ((size_t)(ri.pix + ih * W + iw)) * C + c);
}

// stem: k = (r*7 + s)*3 + c over the NCHW fp32 image; ri.pix = batch index
template <typename T>
__device__ __forceinline__ Vec16<T> load_a_stem(const float* img, const RowInfo& ri, int k0, int H, int W, int Kreal) {
  Vec16<T> v = zero16<T>();
  if (ri.pix < 0) return v;
#pragma unroll
  for (int j = 0; j < Vec16<T>::N; ++j) {
    int k = k0 + j;
    if (k < Kreal) {
      int tap = k / 3, c = k - tap * 3;
      int r = tap / 7, s = tap - r * 7;
      int ih = ri.ih0 + r, iw = ri.iw0 + s;
      if (ih >= 0 && ih < H && iw >= 0 && iw < W)
        v.set(j, img[((size_t)(ri.pix * 3 + c) * H + ih) * W + iw]);
    }
  }
  return v;
}

// LDS tiles are unpadded 128-byte rows; the 16-byte chunk c of row r lives at chunk position c ^ (r & 7), which makes both
// the ds_write_b128 staging stores and the ds_read_b128 / ds_read_b32 fragment reads bank-conflict free (the former
// +16 B row padding was 2-way conflicted on every fragment read).
// Two shapes of the same kernel: 4 waves x (64 x 64) with BK = 64 (bf16) and, for 128-wide N tiles, 2 waves x (128 x 64) with
// BK = 32: the larger wave tile reads 25% fewer LDS bytes per MFMA (the 64 x 64 shape keeps the LDS pipe as busy as the MFMA
// pipe), its 64-byte rows use the swizzle c ^ ((r >> 2) & 3), and four 2-wave workgroups share a CU.
template <typename T, int BM, int BN, int BK = GT<T>::BK, int ST = 2, int WIN = 0> struct IGemmCfg {
  static constexpr int LD = BK;
  static constexpr int TILES = ST * (BM + 8 * WIN + BN) * LD * (int)sizeof(T);   // WIN: the A buffers hold a window of BM + 8 pixels
  static constexpr int CST = BM * (BN + GT<T>::VEC) * (int)sizeof(T);
  static constexpr int SMEM = (TILES > CST + 8192 ? TILES : CST + 8192);   // BN-statistics / BN-backward scratch sits right after the C staging area
};

// sum over the 16 lanes of a DPP row (all 16 lanes receive it): quad_perm [1,0,3,2], [2,3,0,1], row_half_mirror, row_mirror
__device__ __forceinline__ float row16_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
  return v;
}

template <typename T, int BM, int BN, int LOADER, int NW = 4, int BK = GT<T>::BK, int OCC = 2, int ST = 2, int WIN = 0>
__global__ __launch_bounds__(NW * 64, OCC) void igemm_kernel(IGemmParams p) {   // OCC waves per SIMD (2 -> <= 256 VGPRs)
  using G = GT<T>;
  constexpr int VEC = G::VEC, LD = BK;
  constexpr int NTHR = NW * 64;
  constexpr int CPR = BK / VEC;                 // 16-byte chunks per LDS row (8 or 4)
  constexpr int RPP = NTHR / CPR;               // tile rows covered by one staging pass of the workgroup
  constexpr int RPI = 64 / CPR;                 // tile rows written by one wave-wide LDS-DMA instruction
  static_assert(RPP % 32 == 0 && (CPR == 8 || CPR == 4) && BN <= NTHR, "staging map");
  constexpr int WN = BN / 64, WM = NW / WN, TM = BM / WM, MT = TM / 16, NT = 4;
  constexpr int AV = BM / RPP, BV = BN / RPP;
  constexpr int SMEM = IGemmCfg<T, BM, BN, BK, ST, WIN>::SMEM;
  static_assert(ST == 2 || ((ST == 3 || ST == 4) && LOADER != LOADER_STEM), "ring depth");
  static_assert(!WIN || (LOADER == LOADER_NHWC && ST == 2 && sizeof(T) == 2 && CPR == 8), "window loader: bf16 NHWC, double buffered");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* As = reinterpret_cast<T*>(smem);
  T* Bs = As + ST * BM * LD;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: LDS-DMA destinations and tile offsets stay in SGPRs
  const int wm = wave / WN, wn = wave % WN;
  const int tiles_n = (p.N + BN - 1) / BN;
  // XCD-aware tile order: hardware deals consecutive workgroup ids round-robin to the 8 XCDs (each with its own L2), so workgroup
  // b of XCD b%8 takes the (b/8)-th tile of that XCD's CONTIGUOUS range: neighbouring tiles (shared halo rows, the other
  // N tiles of the same rows, the same weight tile) meet in one L2.
  int bid = blockIdx.x;
  if (!(p.dbg & 8)) {
    const int nt = gridDim.x, fl = nt >> 3, rem = nt & 7, xcd = bid & 7;
    bid = xcd * fl + (xcd < rem ? xcd : rem) + (bid >> 3);
  }
  int tile_m = bid / tiles_n;
  const int tile_n = bid - tile_m * tiles_n;
  const int n0 = tile_n * BN;
  const int vec = tid % CPR, rbase = tid / CPR;
  const int swz = (CPR == 8) ? (rbase & 7) : ((rbase >> 2) & 3);      // unchanged by the + 32*i of the staging passes
  const T* aT = reinterpret_cast<const T*>(p.a);
  const T* a2T = reinterpret_cast<const T*>(p.a2);
  const float* aImg = reinterpret_cast<const float*>(p.a);
  const T* wT = reinterpret_cast<const T*>(p.w);
  // DGRAD2: rows are (class, b, h/2, w/2); a tile never straddles two parity classes
  const int Hh = p.Ho >> 1, Wh = p.Wo >> 1, class_rows = p.B * Hh * Wh;
  int cls = 0;
  if (LOADER == LOADER_DGRAD2) {
    const int tpc = (class_rows + BM - 1) / BM;
    cls = tile_m / tpc; tile_m -= cls * tpc;
  }
  const int m0 = tile_m * BM;
  const int row_limit = (LOADER == LOADER_DGRAD2) ? class_rows : p.M;

  RowInfo ri[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) {
    if (LOADER == LOADER_DGRAD2) {
      const int r = m0 + rbase + RPP * i;
      if (r >= class_rows) { ri[i].pix = -1; ri[i].ih0 = 0; ri[i].iw0 = 0; }
      else { const int b = fast_div(r, p.mul_howo), rem = r - b * Hh * Wh; ri[i].pix = b * p.H * p.W; ri[i].ih0 = fast_div(rem, p.mul_wo); ri[i].iw0 = rem - ri[i].ih0 * Wh; }
    } else {
      ri[i] = decode_row(m0 + rbase + RPP * i, p.M, p.Ho * p.Wo, p.Wo, p.H * p.W, p.stride, p.pad, p.transposed, LOADER == LOADER_STEM,
                         p.mul_howo, p.mul_wo);
    }
  }

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  if constexpr (WIN) {
    // ---- window loader (stride-1 3x3 pad-1 conv and its data gradient; H == Ho, W == Wo).  Tap (r, s) of output pixel m reads
    //      the flattened input pixel q = m + dr(r)*W + (ds(s) - 1) (dr = r-1 / 1-r, ds = s / 2-s for forward / transposed), so the
    //      three taps of one filter row read ONE window of BM + 2 consecutive pixels: it is staged once per (r, channel chunk) and
    //      sub-step ds reads its fragments at row offset ds.  Positions whose (oh+dr, ow+ds-1) fall outside the image are zeroed
    //      at fragment level with per-lane bit masks (the wrapped pixel q is real data there).  A pieces per wave and K step:
    //      4/3 (+1/3 tail) instead of 4 -- the LDS-DMA issue cost is what bounds this kernel.
    constexpr int AW = BM + 8;
    constexpr int OOB = (int)0x80000000;
    constexpr int ES = (int)sizeof(T);
    T* Aw = reinterpret_cast<T*>(smem);                  // [2][AW][LD]
    T* Bw = Aw + 2 * AW * LD;                            // [2][BN][LD]
    const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)p.a_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
    const int cpb = p.C / BK, nwin = 3 * cpb, nk = 3 * nwin;
    const int csz = p.C * ES, npix = p.B * p.H * p.W;
    const bool trm = p.transposed != 0;
    const int lvec = vec ^ swz;
    unsigned vmask[MT];                                  // bit r*3 + ds: tap row r, window offset ds is inside the image
#pragma unroll
    for (int i = 0; i < MT; ++i) {
      const int m = m0 + wm * TM + i * 16 + (lane & 15);
      unsigned mk = 0;
      if (m < p.M) {
        const int b = fast_div(m, p.mul_howo), rem = m - b * (p.Ho * p.Wo);
        const int oh = fast_div(rem, p.mul_wo), ow = rem - oh * p.Wo;
#pragma unroll
        for (int r = 0; r < 3; ++r) {
          const int dr = trm ? 1 - r : r - 1;
          const bool rok = (unsigned)(oh + dr) < (unsigned)p.H;
#pragma unroll
          for (int ds = 0; ds < 3; ++ds)
            if (rok && (unsigned)(ow + ds - 1) < (unsigned)p.W) mk |= 1u << (r * 3 + ds);
        }
      }
      vmask[i] = mk;
    }
    int boff[BV], aoff[AV], aoff_tail = OOB;
#pragma unroll
    for (int i = 0; i < BV; ++i) {
      const int n = n0 + rbase + RPP * i;
      boff[i] = (n < p.N) ? (n * p.Kw + lvec * VEC) * ES : OOB;
    }
    auto set_row = [&](int r) {                          // vector offsets of this thread's window rows for filter row r
      const int qb = m0 - 1 + (trm ? 1 - r : r - 1) * p.W;
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        const int q = qb + rbase + RPP * i;
        aoff[i] = ((unsigned)q < (unsigned)npix) ? q * csz + lvec * VEC * ES : OOB;
      }
      const int qt = qb + BM + (lane >> 3);              // the 8 tail rows, staged by wave 0 (rbase == lane >> 3 there)
      aoff_tail = ((unsigned)qt < (unsigned)npix) ? qt * csz + lvec * VEC * ES : OOB;
    };
#if defined(__HIP_DEVICE_COMPILE__)
    typedef __attribute__((address_space(3))) void* lds_ptr;
    auto load_a = [&](int w, int cc) {                   // window w -> A buffer w & 1
      char* abase = smem + ((size_t)((w & 1) * AW + wave * RPI) * LD) * ES;
#pragma unroll
      for (int i = 0; i < AV; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(abase + i * RPP * LD * ES), 16, aoff[i], cc * BK * ES, 0, 0);
      if (wave == 0)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(smem + ((size_t)((w & 1) * AW + BM) * LD) * ES), 16, aoff_tail, cc * BK * ES, 0, 0);
    };
    auto load_b = [&](int kt, int r, int cc, int j) {    // K step kt = 3*w + j -> B buffer kt & 1
      const int s_ = trm ? 2 - j : j;
      const int kw = ((r * 3 + s_) * p.C + cc * BK) * ES;
      char* bbase = smem + ((size_t)(2 * AW + (kt & 1) * BN + wave * RPI) * LD) * ES;
#pragma unroll
      for (int i = 0; i < BV; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(bbase + i * RPP * LD * ES), 16, boff[i], kw, 0, 0);
    };
#else
    auto load_a = [&](int, int) {};
    auto load_b = [&](int, int, int, int) {};
#endif
    auto compute_w = [&](int kt, int w, int r, int j) {
      const T* Ab = Aw + ((w & 1) * AW + wm * TM + (lane & 15) + j) * LD;
      const T* Bb = Bw + ((kt & 1) * BN + wn * 64 + (lane & 15)) * LD;
      const int rsa = ((lane & 15) + j) & 7, rsb = lane & 7;
      const unsigned bit = 1u << (r * 3 + j);
#pragma unroll
      for (int kk = 0; kk < BK / G::MK; ++kk) {
        bf16x8 af[MT], bfv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * LD + (((kk * 4 + (lane >> 4)) ^ rsa) * 8));
          if (!(vmask[i] & bit)) af[i] = bf16x8{};
        }
#pragma unroll
        for (int j2 = 0; j2 < NT; ++j2) bfv[j2] = *reinterpret_cast<const bf16x8*>(Bb + j2 * 16 * LD + (((kk * 4 + (lane >> 4)) ^ rsb) * 8));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j2 = 0; j2 < NT; ++j2)
            acc[i][j2] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[j2], af[i], acc[i][j2], 0, 0, 0);   // operands swapped: C^T
      }
    };
    set_row(0); load_a(0, 0); load_b(0, 0, 0, 0);
    __syncthreads();
    int w = 0, r = 0, cc = 0, j = 0;                     // cursor of the CURRENT step: window w = r*cpb + cc, sub-step j
    for (int kt = 0; kt < nk; ++kt) {
      // next step's (w, r, cc, j)
      int nj = j + 1, ncc = cc, nr = r, nw_ = w;
      if (nj == 3) { nj = 0; ++nw_; if (++ncc == cpb) { ncc = 0; ++nr; } }
      if (kt + 1 < nk) load_b(kt + 1, nr, ncc, nj);
      if (j == 0 && w + 1 < nwin) {                      // stage the NEXT window while this one serves its three taps
        int r2 = r, cc2 = cc + 1;
        if (cc2 == cpb) { cc2 = 0; ++r2; set_row(r2); }
        load_a(w + 1, cc2);
      }
      compute_w(kt, w, r, j);
      __syncthreads();
      j = nj; cc = ncc; r = nr; w = nw_;
    }
  } else {
  Vec16<T> ra[AV], rb[BV];
  const int taps = p.R * p.S;
  // K steps per tap; a Linear (1 tap) keeps all its K steps in "tap 0"
  const int cpb = (LOADER == LOADER_DGRAD2 || taps > 1) ? p.C / BK : p.Kp / BK;
  const int nk = (LOADER == LOADER_DGRAD2) ? p.ntaps[cls] * cpb : p.Kp / BK;

  // ---- operand loads: buffer loads with 32-bit byte offsets; out-of-range offsets return zeros in hardware,
  //      so padding / tile tails cost no selects.  Per-row offsets are recomputed only when the filter tap changes.
  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsA = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a), 0, (int)p.a_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsA2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.a2 ? p.a2 : p.a), 0, (int)p.a2_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsW = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.w), 0, (int)p.w_bytes, 0x00020000);
  // LDS-DMA staging (buffer_load ... lds): a wave-instruction writes 64 x 16 B linearly = 8 consecutive 128-byte tile rows,
  // so the XOR swizzle is applied on the SOURCE side: the lane sitting at chunk position `vec` of row r fetches chunk
  // lvec = vec ^ (r & 7).  No staging VGPRs, no ds_write; out-of-range offsets land as zeros.
  constexpr bool DMA = (LOADER != LOADER_STEM);
  const int lvec = DMA ? (vec ^ swz) : vec;
  int aoff[AV], boff[BV];
#pragma unroll
  for (int i = 0; i < BV; ++i) {
    const int n = n0 + rbase + RPP * i;
    boff[i] = (n < p.N) ? (n * p.Kw + lvec * VEC) * (int)sizeof(T) : OOB;
  }
  // Per-row byte offset of the lane's 16-byte chunk for filter tap (r, s): selects only, no divergent control flow.
  // Transposed (data-gradient) gathers support stride 1 and 2 (the host entry rejects others): ih = (oh + pad - r) >> sh.
  const int csz = p.C * (int)sizeof(T);
  const int tsh = (p.stride == 2) ? 1 : 0, tmask = p.stride - 1;
  const bool tr_mode = (LOADER != LOADER_DGRAD2) && p.transposed;
  int pixoff[AV];
#pragma unroll
  for (int i = 0; i < AV; ++i) pixoff[i] = ri[i].pix * csz + lvec * VEC * (int)sizeof(T);
  auto set_tap = [&](int r, int s) {          // r,s: filter tap (NHWC) or (dh,dw) source offsets (DGRAD2)
#pragma unroll
    for (int i = 0; i < AV; ++i) {
      bool ok = ri[i].pix >= 0;
      int ih, iw;
      if (!tr_mode) { ih = ri[i].ih0 + r; iw = ri[i].iw0 + s; }
      else {
        const int th = ri[i].ih0 - r, tw = ri[i].iw0 - s;
        ok = ok && (((th | tw) & tmask) == 0);
        ih = th >> tsh; iw = tw >> tsh;            // arithmetic shift: negative stays negative and fails the range test below
      }
      ok = ok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      aoff[i] = ok ? pixoff[i] + (ih * p.W + iw) * csz : OOB;
    }
  };
  const bool a_tail = (LOADER == LOADER_NHWC) && taps == 1 && (p.C % BK) != 0;
  const bool w_tail = (p.Kw % BK) != 0;
  int cc = 0, tap = 0, tr_ = 0, ts_ = 0;
  if (LOADER == LOADER_DGRAD2) { if (nk > 0) set_tap(p.tap_dh[cls][0], p.tap_dw[cls][0]); }
  else if (LOADER == LOADER_NHWC) set_tap(0, 0);

  auto gload = [&](int kt) {
    if (LOADER == LOADER_STEM) {
      const int k0 = kt * BK;
#pragma unroll
      for (int i = 0; i < AV; ++i) ra[i] = load_a_stem<T>(aImg, ri[i], k0 + vec * VEC, p.H, p.W, 147);
#pragma unroll
      for (int i = 0; i < BV; ++i) {
        const int n = n0 + rbase + RPP * i, k = k0 + vec * VEC;
        rb[i] = (n < p.N && k < p.Kw) ? ldg16(wT + (size_t)n * p.Kw + k) : zero16<T>();
      }
      return;
    }
    // The K position inside the row goes into the instruction's SCALAR offset (not part of the hardware range check, so the
    // out-of-range sentinel in the vector offset still yields zeros); the vector offsets are loop-invariant within a tap.
    const int kbyte = cc * BK * (int)sizeof(T);
    const bool second = (LOADER == LOADER_DGRAD2) && p.tap_src[cls][tap];
    const int kw = (LOADER == LOADER_DGRAD2) ? p.tap_koff[cls][tap] + cc * BK : kt * BK;
#if defined(__HIP_DEVICE_COMPILE__)     // device pass only: the host pass cannot form LDS (address_space 3) pointers
    typedef __attribute__((address_space(3))) void* lds_ptr;
    const int slot = (ST == 2) ? (kt & 1) : (kt % ST);
    char* abase = smem + ((size_t)(slot * BM + wave * RPI) * LD) * sizeof(T);
    char* bbase = smem + ((size_t)(ST * BM + slot * BN + wave * RPI) * LD) * sizeof(T);
    if (a_tail) {                                     // Linear whose K is not a multiple of BK: mask the chunks past the row end
      const bool cok = (cc * BK + lvec * VEC) < p.C;
#pragma unroll
      for (int i = 0; i < AV; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(abase + i * RPP * LD * (int)sizeof(T)), 16, cok ? aoff[i] : OOB, kbyte, 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < AV; ++i) {
        if (second) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA2, (lds_ptr)(abase + i * RPP * LD * (int)sizeof(T)), 16, aoff[i], kbyte, 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsA, (lds_ptr)(abase + i * RPP * LD * (int)sizeof(T)), 16, aoff[i], kbyte, 0, 0);
      }
    }
    if (w_tail) {
      const bool kok = (kw + lvec * VEC) < p.Kw;
#pragma unroll
      for (int i = 0; i < BV; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(bbase + i * RPP * LD * (int)sizeof(T)), 16, kok ? boff[i] : OOB,
                                                 kw * (int)sizeof(T), 0, 0);
    } else {
#pragma unroll
      for (int i = 0; i < BV; ++i)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rsW, (lds_ptr)(bbase + i * RPP * LD * (int)sizeof(T)), 16, boff[i], kw * (int)sizeof(T), 0, 0);
    }
#else
    (void)kbyte; (void)second; (void)kw;
#endif
    // advance the (tap, channel-chunk) cursor for the next call
    if (++cc == cpb) {
      cc = 0; ++tap;
      if (LOADER == LOADER_DGRAD2) { if (tap < p.ntaps[cls]) set_tap(p.tap_dh[cls][tap], p.tap_dw[cls][tap]); }
      else { if (++ts_ == p.S) { ts_ = 0; ++tr_; } if (tr_ < p.R) set_tap(tr_, ts_); }
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < AV; ++i)      // (rbase + 32 i) & 7 == rbase & 7
      *reinterpret_cast<u32x4*>(&As[(buf * BM + rbase + RPP * i) * LD + ((vec ^ swz) * VEC)]) = ra[i].raw;
#pragma unroll
    for (int i = 0; i < BV; ++i)
      *reinterpret_cast<u32x4*>(&Bs[(buf * BN + rbase + RPP * i) * LD + ((vec ^ swz) * VEC)]) = rb[i].raw;
  };
  auto compute = [&](int buf) {
    const T* Ab = As + (buf * BM + wm * TM + (lane & 15)) * LD;
    const T* Bb = Bs + (buf * BN + wn * 64 + (lane & 15)) * LD;
    const int rswz = (CPR == 8) ? (lane & 7) : ((lane >> 2) & 3);     // swizzle term of fragment row (lane & 15) + 16*i
#pragma unroll
    for (int kk = 0; kk < BK / G::MK; ++kk) {
      if constexpr (sizeof(T) == 2) {
        bf16x8 af[MT], bfv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = *reinterpret_cast<const bf16x8*>(Ab + i * 16 * LD + (((kk * 4 + (lane >> 4)) ^ rswz) * 8));
#pragma unroll
        for (int j = 0; j < NT; ++j) bfv[j] = *reinterpret_cast<const bf16x8*>(Bb + j * 16 * LD + (((kk * 4 + (lane >> 4)) ^ rswz) * 8));
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(bfv[j], af[i], acc[i][j], 0, 0, 0);   // operands swapped: C^T
      } else {
        float af[MT], bfv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = Ab[i * 16 * LD + ((kk ^ (lane & 7)) * 4) + (lane >> 4)];
#pragma unroll
        for (int j = 0; j < NT; ++j) bfv[j] = Bb[j * 16 * LD + ((kk ^ (lane & 7)) * 4) + (lane >> 4)];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(bfv[j], af[i], acc[i][j], 0, 0, 0);      // operands swapped: C^T
      }
    }
  };

  // p.dbg (VQA_IGEMM_DBG, measurement only): bit 0 = no DMA inside the K loop, bit 1 = no MFMA / LDS-read phase
  const bool dbg_nodma = p.dbg & 1, dbg_nomma = p.dbg & 2;
  if constexpr (ST >= 3) {
    // ST-slot ring, ST - 1 K steps of DMA in flight: the barrier that ends step kt only waits for tile kt + 1 (counted vmcnt: the
    // AV + BV LDS-DMA instructions of each later tile may still be outstanding), so a tile has ST - 1 compute phases to land.
    // (Round 4 tried it, 3 and 4 slots, on the dense Linears of the token side -- K = 256 ... 1024, 4 ... 16 K steps of ~0.1 us of MFMA
    // work each: 15-50 % SLOWER wherever the launch is not at the ~11 us floor anyway (fewer workgroups per CU); tools/token_gemm_sweep.py.)
#pragma unroll
    for (int t = 0; t < ST - 1; ++t)
      if (t < nk) gload(t);
    auto wait_tiles = [&](int outstanding) __attribute__((always_inline)) {     // wait until at most `outstanding` tiles are in flight, then barrier
      if (outstanding >= 2 && ST >= 4) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(2 * (AV + BV)) : "memory");
      else if (outstanding == 1) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(AV + BV) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    };
    {
      const int issued = nk < ST - 1 ? nk : ST - 1;
      wait_tiles(issued - 1);                                       // tile 0 has landed
    }
    for (int kt = 0; kt < nk; ++kt) {
      const bool more = (kt + ST - 1 < nk) && !dbg_nodma;
      if (more) gload(kt + ST - 1);        // slot (kt - 1) % ST: every wave finished reading it before the last barrier
      if (!dbg_nomma) compute(kt % ST);
      const int issued = dbg_nodma ? (nk < ST - 1 ? nk : ST - 1) : (kt + ST < nk ? kt + ST : nk);      // tiles issued so far
      const int left = issued - (kt + 2);                            // tiles that may stay in flight once tile kt + 1 has landed
      wait_tiles(left < 0 ? 0 : left);
    }
  } else {
    if (nk > 0) { gload(0); if (!DMA) sstore(0); }
    __syncthreads();                     // (with LDS-DMA in flight the barrier's fence waits vmcnt(0): tile 0 has landed)
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk && !dbg_nodma) gload(kt + 1);   // DMA: straight into the other LDS buffer, all waves left it at the last barrier
      if (!dbg_nomma) compute(kt & 1);
      if (!DMA && kt + 1 < nk) sstore((kt + 1) & 1);
      __syncthreads();
    }
  }

  }

  // ---- epilogue.  The MFMA operands were swapped, so a lane holds 4 CONSECUTIVE COLUMNS of one row:
  //        acc[i][j][r] = C[m = wm*TM + i*16 + (lane & 15)][n = wn*64 + j*16 + (lane >> 4)*4 + r]
  //      -> packed conversion and one 8-byte (bf16) / 16-byte (fp32) LDS store per (i, j) instead of four scalar ones.
  const int lm = lane & 15, lq = lane >> 4;
  if (p.bias || p.relu) {
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int nb = n0 + wn * 64 + j * 16 + lq * 4;
      float bv[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) bv[r] = (p.bias && nb + r < p.N) ? p.bias[nb + r] : 0.f;
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float v = acc[i][j][r] + bv[r];
          acc[i][j][r] = (p.relu == 1 && v < 0.f) ? 0.f : v;   // NaN-propagating ReLU like torch
        }
    }
  }
  if (p.drop_p > 0.f) {
    const float ks = 1.f / (1.f - p.drop_p);
    const uint32_t dkey = drop_key(p.drop_seed);          // M*N < 2^32 (checked by the host entry)
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const size_t m = (size_t)(m0 + wm * TM + i * 16 + lm);
          const int n = n0 + wn * 64 + j * 16 + lq * 4 + r;
          acc[i][j][r] = drop_keep32(dkey, (uint32_t)(m * p.N + n), p.drop_p) ? acc[i][j][r] * ks : 0.f;
        }
  }
  // ---- BatchNorm partial statistics (sum, sum of squares per output channel of this M tile) ----
  // fp32: DPP row reductions of the accumulators.  bf16: done on the matrix cores AFTER the C tile is staged (below) -- the
  // VALU form cost 26 % of the stage-1 forward convs (K = 576: nine K steps per tile, so the epilogue weighs heavily).
  float* red = reinterpret_cast<float*>(smem + IGemmCfg<T, BM, BN, BK, ST, WIN>::CST);   // [WM][BN][2], after the C staging area (tiles are dead by now)
  if (p.stats && sizeof(T) == 4) {
#pragma unroll
    for (int j = 0; j < NT; ++j)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        float s = 0.f, q = 0.f;
#pragma unroll
        for (int i = 0; i < MT; ++i) { const float v = acc[i][j][r]; s += v; q += v * v; }
        s = row16_sum(s); q = row16_sum(q);               // the 16 rows of the MFMA tile sit on the 16 lanes of a DPP row
        if (lm == 0) {
          red[(wm * BN + wn * 64 + j * 16 + lq * 4 + r) * 2 + 0] = s;
          red[(wm * BN + wn * 64 + j * 16 + lq * 4 + r) * 2 + 1] = q;
        }
      }
  }
  // ---- stage C through LDS so global stores are full 16-byte row segments ---------------------
  constexpr int LDC = BN + VEC;
  T* Cs = reinterpret_cast<T*>(smem);
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      T* dst = &Cs[(wm * TM + i * 16 + lm) * LDC + wn * 64 + j * 16 + lq * 4];
      if constexpr (sizeof(T) == 2) {
        typedef __attribute__((ext_vector_type(2))) float f32x2_t;
        typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
        typedef __attribute__((ext_vector_type(2))) uint32_t u32x2_t;
        const bf16x2_t lo = __builtin_convertvector((f32x2_t){acc[i][j][0], acc[i][j][1]}, bf16x2_t);   // v_cvt_pk_bf16_f32
        const bf16x2_t hi = __builtin_convertvector((f32x2_t){acc[i][j][2], acc[i][j][3]}, bf16x2_t);
        *reinterpret_cast<u32x2_t*>(dst) = u32x2_t{__builtin_bit_cast(uint32_t, lo), __builtin_bit_cast(uint32_t, hi)};
      } else {
        *reinterpret_cast<f32x4*>(dst) = acc[i][j];
      }
    }
  __syncthreads();
  if constexpr (sizeof(T) == 2) {
    if (p.stats) {
      // Column sums and sums of squares of the staged bf16 tile Y [BM][BN] as two MFMA chains per 16-column block: with the
      // transposed fragment F (lane: column l&15, 8 consecutive rows) as BOTH operands, D = F^T F is the Gram block whose diagonal
      // is sum y^2, and ones^T F gives sum y in every row.  Exact fp32 accumulation of the values as stored (rows past M are zero).
      typedef __attribute__((ext_vector_type(8))) short i16x8;
      constexpr int CB = BN / 16;
      const int sg = lane >> 4, sli = lane & 15, sq = sli >> 2, spp = sli & 3;
      const i16x8 ones_i = {0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80, 0x3F80};
      const bf16x8 ones = __builtin_bit_cast(bf16x8, ones_i);
      for (int cb = wave; cb < CB; cb += NW) {              // wave-uniform: EXEC stays full for the transposed reads
        f32x4 dsum = {0.f, 0.f, 0.f, 0.f}, dsq = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < BM / 32; ++ks) {
          const T* bp = Cs + (ks * 32 + 8 * sg + sq) * LDC + cb * 16 + 4 * spp;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(bp));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(bp + 4 * LDC));
          i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          const bf16x8 yf = __builtin_bit_cast(bf16x8, t);
          dsq = __builtin_amdgcn_mfma_f32_16x16x32_bf16(yf, yf, dsq, 0, 0, 0);
          dsum = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ones, yf, dsum, 0, 0, 0);
        }
        const int n = n0 + cb * 16 + sli;                     // D[i][j]: lane holds column j = l&15, rows 4*(l>>4) + r
        if (n < p.N) {
          const float qv = spp == 0 ? dsq[0] : spp == 1 ? dsq[1] : spp == 2 ? dsq[2] : dsq[3];
          if (p.stats_mode) {                                  // order-free fixed-point sums: the consumer finalizes them itself
            const int R = acc_replicas(p.N);
            unsigned long long* fa = reinterpret_cast<unsigned long long*>(p.stats);
            const size_t fr = (size_t)(tile_m % R) * 2 * p.N;
            if (sg == 0) acc_add_fixed(fa, (size_t)R * 2 * p.N, fr + n, dsum[0]);
            if (sg == sq) acc_add_fixed(fa, (size_t)R * 2 * p.N, fr + p.N + n, qv);
          } else {
            if (sg == 0) p.stats[((size_t)tile_m * 2 + 0) * p.N + n] = dsum[0];
            if (sg == sq) p.stats[((size_t)tile_m * 2 + 1) * p.N + n] = qv;
          }
        }
      }
    }
  } else if (p.stats && tid < BN) {
    const int n = n0 + tid;
    if (n < p.N) {
      float s = 0.f, q = 0.f;
#pragma unroll
      for (int w = 0; w < WM; ++w) { s += red[(w * BN + tid) * 2]; q += red[(w * BN + tid) * 2 + 1]; }
      if (p.stats_mode) {
        const int R = acc_replicas(p.N);
        unsigned long long* fa = reinterpret_cast<unsigned long long*>(p.stats);
        const size_t fr = (size_t)(tile_m % R) * 2 * p.N;
        acc_add_fixed(fa, (size_t)R * 2 * p.N, fr + n, s);
        acc_add_fixed(fa, (size_t)R * 2 * p.N, fr + p.N + n, q);
      } else {
        p.stats[((size_t)tile_m * 2 + 0) * p.N + n] = s;
        p.stats[((size_t)tile_m * 2 + 1) * p.N + n] = q;
      }
    }
  }
  constexpr int VR = BN / VEC, RP = NTHR / VR;
  T* outT = reinterpret_cast<T*>(p.out);
  const T* addT = reinterpret_cast<const T*>(p.addend);
  const T* mskT = reinterpret_cast<const T*>(p.addmask);
  const T* omT = reinterpret_cast<const T*>(p.outmask);
  const bool vec_ok = (p.N % VEC) == 0;
  // The thread's rows are handled in chunks of CHR: ALL global loads of a chunk (addend, masks, BatchNorm operand) are issued
  // before its first store -- a load behind a store to a possibly aliasing pointer waits for it, and eight dependent
  // load -> store round trips per tile were a third of the data-gradient launches' time.
  constexpr int NR = BM / RP, CHR = NR < 4 ? NR : 4;
  static_assert(BM % RP == 0 && NR % CHR == 0, "epilogue row chunks");
  const int n = n0 + (tid % VR) * VEC;
  for (int r0 = 0; r0 < NR; r0 += CHR) {
    size_t offs[CHR]; bool live[CHR];
    Vec16<T> av[CHR], mv[CHR], ov[CHR];
#pragma unroll
    for (int i = 0; i < CHR; ++i) {
      const int row = tid / VR + (r0 + i) * RP;
      int m = m0 + row;
      live[i] = m < row_limit && n < p.N;
      if (LOADER == LOADER_DGRAD2 && live[i]) {
        const int b = fast_div(m, p.mul_howo), rem = m - b * Hh * Wh, hh = fast_div(rem, p.mul_wo), ww = rem - hh * Wh;
        m = (b * p.Ho + 2 * hh + (cls >> 1)) * p.Wo + 2 * ww + (cls & 1);
      }
      offs[i] = (size_t)m * p.N + n;
      if (live[i] && vec_ok) {
        if (addT) av[i] = ldg16(addT + offs[i]);
        if (addT && mskT) mv[i] = ldg16(mskT + offs[i]);
        if (omT) ov[i] = ldg16(omT + offs[i]);
      }
    }
#pragma unroll
    for (int i = 0; i < CHR; ++i) {
      if (!live[i]) continue;
      const int row = tid / VR + (r0 + i) * RP;
      const size_t off = offs[i];
      Vec16<T> v; v.raw = *reinterpret_cast<const u32x4*>(&Cs[row * LDC + (tid % VR) * VEC]);
      if (vec_ok) {
        if (addT) {
          if (mskT) {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, v.get(j) + (mv[i].get(j) > 0.f ? av[i].get(j) : 0.f));
          } else {
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, v.get(j) + av[i].get(j));
          }
        }
        if (p.relu == 2) {                             // ReLU AFTER the addend: relu(conv + bias + residual), the folded eval block
#pragma unroll
          for (int j = 0; j < VEC; ++j) { const float x = v.get(j); v.set(j, x < 0.f ? 0.f : x); }
        }
        if (omT) {                                     // data gradient handed to the previous block already masked by ITS ReLU
#pragma unroll
          for (int j = 0; j < VEC; ++j) if (!(ov[i].get(j) > 0.f)) v.set(j, 0.f);
          if (p.out_scale != 1.f) {                    // (uniform) the bf16 value times the keep scale, as vqa_bias_act_bwd forms it
#pragma unroll
            for (int j = 0; j < VEC; ++j) v.set(j, v.get(j) * p.out_scale);
          }
        }
        stg16(outT + off, v);
      } else {
        for (int j = 0; j < VEC && n + j < p.N; ++j) {
          float x = v.get(j);
          if (addT) { float a = to_f<T>(addT[off + j]); x += (!mskT || to_f<T>(mskT[off + j]) > 0.f) ? a : 0.f; }
          if (p.relu == 2 && x < 0.f) x = 0.f;
          if (omT) x = (to_f<T>(omT[off + j]) > 0.f) ? (p.out_scale != 1.f ? to_f<T>(from_f<T>(x)) * p.out_scale : x) : 0.f;
          outT[off + j] = from_f<T>(x);
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// weight gradient
// ------------------------------------------------------------------------------------------------
struct WgradParams {
  const void* dy; const void* x; float* dw;
  float* ws;                 // != nullptr: split s writes its fp32 tile to ws[s][N][Kw] (no atomics); wgrad_reduce_kernel adds the slabs to dw
  int M, N, Kw;              // dy [M][N]; dw [N][Kw] fp32 (+=)
  int B, H, W, C, Ho, Wo, R, S, stride, pad, chunk, dbg_noatomic;
  unsigned dy_bytes, x_bytes;
  unsigned long long mul_howo, mul_wo;       // ceil(2^40 / d): exact m / d for m*d < 2^40
};

// Offset of element (k, col) in a k-major wgrad tile of WIDTH columns.  bf16: unpadded rows, the 32-byte chunk index is XORed
// with a function of k chosen so that the 8 rows one half-wave touches in a ds_read_b64_tr_b16 ({0..3, 8..11} + 16n) land in
// 8 different 32-byte bank slots (the padded layout was 2-way conflicted).  fp32 keeps the padded layout (plain ds_read_b32).
template <typename T, int WIDTH> __device__ __forceinline__ int kmaj_off(int k, int col) {
  if constexpr (sizeof(T) == 2) {
    // reads (ds_read_b64_tr_b16, 32-lane groups touch rows {0..3, 8..11} + 16n): h must be injective on k bits {0, 1, 3} (WIDTH 128:
    // a row is the whole 256-byte bank window) resp. on bits {1, 3} for a fixed bit 0 (WIDTH 64: bit 0 picks the window half).
    // staging stores (ds_write_b128, 8-lane groups = 64 bytes of row r and of row r + 1, 128-byte bank window): the two rows must
    // land in different 64-byte halves, i.e. k bit 0 has to drive chunk-index bit 1 (it drove bit 0 / nothing: 2-way conflicts,
    // SQ_LDS_BANK_CONFLICT = 20 % of the LDS cycles).
    const int h = (WIDTH == 128) ? (((k & 1) << 1) | ((k >> 1) & 1) | (((k >> 3) & 1) << 2))
                                 : ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) ^ ((k & 1) << 1));
    return k * WIDTH + ((((col >> 4) ^ h) << 4) | (col & 15));
  } else {
    return k * (WIDTH + 4) + col;
  }
}

template <typename T, int BMW, int BNW, int LOADER>
__device__ __forceinline__ void wgrad_body(const WgradParams& p, int bx_in, int by_in, int gx_in, int gy_in) {
  using G = GT<T>;
  constexpr int VEC = G::VEC, BKM = G::BKM;
  constexpr int LDY = (sizeof(T) == 2) ? BMW : BMW + VEC, LDX = (sizeof(T) == 2) ? BNW : BNW + VEC;
  constexpr int TMW = BMW / 2, TNW = BNW / 2, MT = TMW / 16, NT = TNW / 16;
  constexpr int VRY = BMW / VEC, VRX = BNW / VEC;
  constexpr int YV = BKM * VRY / 256, XV = BKM * VRX / 256;
  static_assert(YV >= 1 && XV >= 1, "tile too small");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  T* Ys = reinterpret_cast<T*>(smem);          // [2][BKM][LDY]
  T* Xs = Ys + 2 * BKM * LDY;                  // [2][BKM][LDX]
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);          // wave-uniform tile offsets stay in SGPRs
  const int wm = wave >> 1, wn = wave & 1;
  const int tiles_k = (p.Kw + BNW - 1) / BNW;
  // XCD-aware order (see igemm_kernel): the workgroups of one XCD take a contiguous range of (split, tile) pairs, so the tiles
  // that read the same rows of dY / X (one split-K chunk) share that XCD's L2 instead of filling all eight
  int bx = bx_in, by = by_in;
  if (!(p.dbg_noatomic & 2)) {
    const int gx = gx_in, nt = gx * gy_in, lin = by * gx + bx;
    const int fl = nt >> 3, rem = nt & 7, xcd = lin & 7;
    const int l2 = xcd * fl + (xcd < rem ? xcd : rem) + (lin >> 3);
    by = l2 / gx; bx = l2 - by * gx;
  }
  const int tile_n = bx / tiles_k, tile_k = bx - tile_n * tiles_k;
  const int n0 = tile_n * BMW, k20 = tile_k * BNW;
  const int mbeg = by * p.chunk, mend = min(p.M, mbeg + p.chunk);
  if (mbeg >= mend) return;
  const T* dyT = reinterpret_cast<const T*>(p.dy);
  const T* xT = reinterpret_cast<const T*>(p.x);
  const float* xImg = reinterpret_cast<const float*>(p.x);
  const int taps = p.R * p.S;
  int tap = 0, c0 = k20;
  if (LOADER == LOADER_NHWC && taps > 1) { tap = k20 / p.C; c0 = k20 - tap * p.C; }
  const int tr = tap / p.S, ts = tap - tr * p.S;
  const int HoWo = p.Ho * p.Wo, HW = p.H * p.W;

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  Vec16<T> ry[YV], rx[XV];

  constexpr int OOB = (int)0x80000000;
  const __amdgpu_buffer_rsrc_t rsY = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.dy), 0, (int)p.dy_bytes, 0x00020000);
  const __amdgpu_buffer_rsrc_t rsX = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
  // Staging map: a thread owns ONE pixel row of the step tile for X (so the (b,oh,ow) decode is done once per step)
  // and XV channel vectors of it; Y uses the plain linear map.
  constexpr int XT = 256 / BKM;                       // threads per X row
  static_assert(VRX % XT == 0 && VRX / XT == XV, "X staging map");
  const int xrow = tid / XT, xv0 = tid % XT;
  const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
  int xb, xoh, xow;                                     // pixel of row xrow at the first step of this chunk (exact magic division, once)
  {
    const int m = mbeg + xrow;
    xb = fast_div(m, p.mul_howo);
    const int rem = m - xb * HoWo;
    xoh = fast_div(rem, p.mul_wo); xow = rem - xoh * p.Wo;
  }
  // dY: the lane's offsets inside a step tile never change; the step position goes into the scalar offset of the load
  int yoff[YV];
#pragma unroll
  for (int i = 0; i < YV; ++i) {
    const int idx = tid + 256 * i, row = idx / VRY, v = idx - row * VRY, n = n0 + v * VEC;
    yoff[i] = (n < p.N) ? (row * p.N + n) * (int)sizeof(T) : OOB;
  }
  auto gload = [&](int ms) {
    const int ysoff = ms * p.N * (int)sizeof(T);
    if (ms + BKM <= mend) {
#pragma unroll
      for (int i = 0; i < YV; ++i) ry[i].raw = __builtin_amdgcn_raw_buffer_load_b128(rsY, yoff[i], ysoff, 0);
    } else {                                              // last, partial step of this chunk: rows >= mend contribute zeros
#pragma unroll
      for (int i = 0; i < YV; ++i) {
        const int row = (tid + 256 * i) / VRY;
        ry[i].raw = __builtin_amdgcn_raw_buffer_load_b128(rsY, (ms + row < mend) ? yoff[i] : OOB, ysoff, 0);
      }
    }
    const int m = ms + xrow;
    if (LOADER == LOADER_NHWC) {
      // (xb, xoh, xow) is this thread's pixel for the current step; 24-bit multiplies (all factors < 2^24) instead of the
      // quarter-rate 32-bit ones, and the step-to-step update below needs no wide division
      const int ih = __mul24(xoh, p.stride) - p.pad + tr, iw = __mul24(xow, p.stride) - p.pad + ts;
      const bool ok = m < mend && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
      const int pixel = __mul24(xb, HW) + __mul24(ih, p.W) + iw;
      const int base = ok ? (__mul24(pixel, p.C) + c0) * (int)sizeof(T) : OOB;
      {                                                   // advance to the next step's pixel: m += BKM
        xow += BKM;
        const int q = (int)(((float)xow + 0.5f) * inv_wo);  // exact for these magnitudes: (x + 0.5) / d is never an integer
        xow -= __mul24(q, p.Wo); xoh += q;
        const int q2 = (int)(((float)xoh + 0.5f) * inv_ho);
        xoh -= __mul24(q2, p.Ho); xb += q2;
      }
#pragma unroll
      for (int i = 0; i < XV; ++i) {
        const int c = (xv0 + XT * i) * VEC;
        rx[i].raw = __builtin_amdgcn_raw_buffer_load_b128(rsX, (c0 + c < p.C) ? base + c * (int)sizeof(T) : OOB, 0, 0);
      }
    } else {
      RowInfo ri = decode_row(m, mend, HoWo, p.Wo, HW, p.stride, p.pad, 0, true, p.mul_howo, p.mul_wo);
#pragma unroll
      for (int i = 0; i < XV; ++i) rx[i] = load_a_stem<T>(xImg, ri, k20 + (xv0 + XT * i) * VEC, p.H, p.W, p.Kw);
    }
  };
  auto sstore = [&](int buf) {
#pragma unroll
    for (int i = 0; i < YV; ++i) {
      const int idx = tid + 256 * i, row = idx / VRY, v = idx - row * VRY;
      *reinterpret_cast<u32x4*>(&Ys[buf * BKM * LDY + kmaj_off<T, BMW>(row, v * VEC)]) = ry[i].raw;
    }
#pragma unroll
    for (int i = 0; i < XV; ++i)
      *reinterpret_cast<u32x4*>(&Xs[buf * BKM * LDX + kmaj_off<T, BNW>(xrow, (xv0 + XT * i) * VEC)]) = rx[i].raw;
  };
  auto compute = [&](int buf) {
    const int g = lane >> 4, li = lane & 15;
    if constexpr (sizeof(T) == 2) {
      // ds_read_b64_tr_b16: lane 4q+p of a 16-lane group addresses row q, columns 4p..4p+3 of a 4x16 block and
      // receives column (lane&15) of the four rows -> two reads give the 8 contraction values of one MFMA operand.
      const int q = li >> 2, pp = li & 3;
#pragma unroll
      for (int ks = 0; ks < BKM / 32; ++ks) {
        const int k0 = ks * 32 + 8 * g + q;                                  // rows k0 (lo) and k0 + 4 (hi): same swizzle term
        const T* yb = Ys + buf * BKM * LDY;
        const T* xb = Xs + buf * BKM * LDX;
        typedef __attribute__((ext_vector_type(8))) short i16x8;
        bf16x8 af[MT], bfv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) {
          const int col = wm * TMW + i * 16 + 4 * pp;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(yb + kmaj_off<T, BMW>(k0, col)));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(yb + kmaj_off<T, BMW>(k0 + 4, col)));
          i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          af[i] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int j = 0; j < NT; ++j) {
          const int col = wn * TNW + j * 16 + 4 * pp;
          i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(xb + kmaj_off<T, BNW>(k0, col)));
          i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(xb + kmaj_off<T, BNW>(k0 + 4, col)));
          i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          bfv[j] = __builtin_bit_cast(bf16x8, t);
        }
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfv[j], acc[i][j], 0, 0, 0);
      }
    } else {
#pragma unroll
      for (int kk = 0; kk < BKM / 4; ++kk) {
        float af[MT], bfv[NT];
#pragma unroll
        for (int i = 0; i < MT; ++i) af[i] = Ys[buf * BKM * LDY + kmaj_off<T, BMW>(kk * 4 + g, wm * TMW + i * 16 + li)];
#pragma unroll
        for (int j = 0; j < NT; ++j) bfv[j] = Xs[buf * BKM * LDX + kmaj_off<T, BNW>(kk * 4 + g, wn * TNW + j * 16 + li)];
#pragma unroll
        for (int i = 0; i < MT; ++i)
#pragma unroll
          for (int j = 0; j < NT; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bfv[j], acc[i][j], 0, 0, 0);
      }
    }
  };

  const int nsteps = (mend - mbeg + BKM - 1) / BKM;
  gload(mbeg);
  sstore(0);
  __syncthreads();
  for (int st = 0; st < nsteps; ++st) {
    if (st + 1 < nsteps) gload(mbeg + (st + 1) * BKM);
    compute(st & 1);
    if (st + 1 < nsteps) sstore((st + 1) & 1);
    __syncthreads();
  }
  // flush: stage the fp32 tile through LDS (free now) so every atomic wave-instruction adds 256 CONTIGUOUS bytes of one dW row
  // (the accumulator layout would give 4 x 64-byte segments per instruction)
  float* Ct = reinterpret_cast<float*>(smem);                    // [BMW][BNW + 1]
  constexpr int LDCT = BNW + 1;
  constexpr bool STAGE = (size_t)BMW * LDCT * 4 <= (size_t)2 * BKM * (LDY + LDX) * sizeof(T);
  if (STAGE && (p.ws || !(p.dbg_noatomic & 1))) {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          Ct[(wm * TMW + i * 16 + (lane >> 4) * 4 + r) * LDCT + wn * TNW + j * 16 + (lane & 15)] = acc[i][j][r];
    __syncthreads();
    if (p.ws) {                                                      // deterministic two-pass split: plain coalesced stores into this split's slab
      float* slab = p.ws + (size_t)by * p.N * p.Kw;
      for (int idx = tid; idx < BMW * BNW; idx += 256) {
        const int row = idx / BNW, col = idx - row * BNW;
        const int n = n0 + row, k2 = k20 + col;
        if (n < p.N && k2 < p.Kw) slab[(size_t)n * p.Kw + k2] = Ct[row * LDCT + col];
      }
    } else {
      for (int idx = tid; idx < BMW * BNW; idx += 256) {
        const int row = idx / BNW, col = idx - row * BNW;
        const int n = n0 + row, k2 = k20 + col;
        if (n < p.N && k2 < p.Kw) atomicAdd(p.dw + (size_t)n * p.Kw + k2, Ct[row * LDCT + col]);
      }
    }
  } else {
#pragma unroll
    for (int i = 0; i < MT; ++i)
#pragma unroll
      for (int j = 0; j < NT; ++j)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int n = n0 + wm * TMW + i * 16 + (lane >> 4) * 4 + r;
          const int k2 = k20 + wn * TNW + j * 16 + (lane & 15);
          if (n < p.N && k2 < p.Kw) {
            if (p.ws) p.ws[((size_t)by * p.N + n) * p.Kw + k2] = acc[i][j][r];
            else atomicAdd(p.dw + (size_t)n * p.Kw + k2, acc[i][j][r]);
          }
        }
  }
}

template <typename T, int BMW, int BNW, int LOADER>
__global__ __launch_bounds__(256) void wgrad_kernel(WgradParams p) {
  wgrad_body<T, BMW, BNW, LOADER>(p, blockIdx.x, blockIdx.y, gridDim.x, gridDim.y);
}

// Several independent weight gradients in ONE launch (the token-side Linears: M ~ 10^4 rows, 40-160 workgroups and ~20 us of
// dependent latency each -- launched one by one they leave most CUs idle).  Workgroup b belongs to the job whose block range
// holds b; inside the job it is the (tile, split) workgroup it would have been in its own launch, XCD order included.
constexpr int WG_MAXJOBS = 8;
struct WgradGroup { WgradParams p[WG_MAXJOBS]; int blk0[WG_MAXJOBS + 1]; int gx[WG_MAXJOBS]; int gy[WG_MAXJOBS]; int n; };
template <typename T, int BMW, int BNW, int LOADER>
__global__ __launch_bounds__(256) void wgrad_group_kernel(WgradGroup g) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < WG_MAXJOBS; ++q) if (q < g.n && (int)blockIdx.x >= g.blk0[q]) j = q;
  const int local = (int)blockIdx.x - g.blk0[j], gx = g.gx[j];
  wgrad_body<T, BMW, BNW, LOADER>(g.p[j], local % gx, local / gx, gx, g.gy[j]);
}
struct ReduceGroup { const float* ws[WG_MAXJOBS]; float* dw[WG_MAXJOBS]; int nsplit[WG_MAXJOBS]; unsigned total4[WG_MAXJOBS]; int blk0[WG_MAXJOBS + 1]; int n; };

// ------------------------------------------------------------------------------------------------
// weight gradient, bf16 throughput path: LDS-DMA ring, 8 waves, one workgroup per CU, deterministic two-pass split-K
// ------------------------------------------------------------------------------------------------
// dW[N][Kw] = sum over pixels m of dY[m][n] * X[src(m, tap)][c].  The contraction index is the pixel, so both operands are
// staged k-major ([pixel][column]) and read with ds_read_b64_tr_b16.  Differences from wgrad_kernel above:
//   * tiles of TN x TK = 256x256 / 128x256 / 256x128 output elements, 8 waves with 128x64 or 64x64 wave tiles (0.75 / 1.0 LDS
//     reads per MFMA instead of 1.0 plus a ds_write pass), ONE workgroup per CU (grid ~ 256);
//   * operands go global -> LDS by LDS-DMA (buffer_load ... lds, no staging registers, no ds_write) into a ring of NS stages;
//     a stage is an array of [64 pixels][64 columns] bf16 sub-images (128-byte rows, 8 KB), so that one wave-instruction
//     (64 lanes x 16 B = 8 rows) stays inside one sub-image and its (tap, channel block) sits in the SCALAR offset.  Wave w
//     issues rows 8w .. 8w+7 of EVERY sub-image: a lane owns ONE pixel row per step, i.e. one (b, oh, ow) update per step;
//   * the bank swizzle of the transposed reads is applied on the SOURCE side (the lane at chunk position p of row k fetches
//     chunk p ^ 2h(k)); out-of-image taps, rows past the split and column tails are out-of-range offsets -> zeros;
//   * each split writes its fp32 tile to its own slab of the caller's workspace with plain 16-byte stores (the MFMA operands
//     are swapped so a lane holds 4 consecutive dW columns); wgrad_reduce_kernel then adds the slabs to dW in a FIXED order:
//     no atomics (they capped the old kernel at the chip's ~1.3 TB/s float-atomic rate and made dW differ in the last bits
//     from run to run), dW is bit-reproducible.
template <int TN, int TK, int NS> struct WgradDmaCfg {
  static constexpr int SY = TN / 64, SX = TK / 64, SUB = 64 * 64 * 2;
  static constexpr int STAGE = (SY + SX) * SUB, SMEM = NS * STAGE;
  static constexpr int WK = TK / 64, WN = 8 / WK, TNW = TN / WN, MT = TNW / 16, NT = 4;
};

__device__ __forceinline__ int kmaj64_off(int k, int col) {          // element offset inside a [64][64] bf16 sub-image
  const int h = ((((k >> 1) & 1) | (((k >> 3) & 1) << 1)) ^ ((k & 1) << 1));
  return k * 64 + ((((col >> 4) ^ h) << 4) | (col & 15));
}

// One LDS-DMA piece (64 lanes x 16 B -> 1 KB of LDS at the wave-uniform byte address lds_addr) as inline asm: hipcc treats the
// builtin form as a pending LDS write and puts `s_waitcnt vmcnt(0)` in front of the next ds_read of ANY LDS address, which
// serialises the DMA of the next stage with the fragment reads of the current one.  Hidden from the compiler, the pieces are
// counted by hand (vmcnt: the only vector-memory operations of the main loop are these) and ordered for the readers by
// `s_waitcnt vmcnt(N)` + `s_barrier`.  M0 (LDS base of the DMA) is written in the same statement that uses it.
typedef int i32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ i32x4_t make_rsrc(const void* ptr, unsigned bytes) {
  const unsigned long long a = (unsigned long long)ptr;
  return i32x4_t{(int)(unsigned)a, (int)((unsigned)(a >> 32) & 0xffffu), (int)bytes, 0x00020000};
}
__device__ __forceinline__ void dma16(i32x4_t rs, unsigned lds_addr, int voff, int soff) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);     // wave-uniform by construction; make it provably so ("s" operands)
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr), "s"(soff) : "memory");
}

template <int TN, int TK, int NS>
__global__ __launch_bounds__(512, 2) void wgrad_dma_kernel(WgradParams p) {
  using Cfg = WgradDmaCfg<TN, TK, NS>;
  constexpr int SY = Cfg::SY, SX = Cfg::SX, SUB = Cfg::SUB, STAGE = Cfg::STAGE;
  constexpr int WK = Cfg::WK, TNW = Cfg::TNW, MT = Cfg::MT, NT = Cfg::NT;
  constexpr int OOB = (int)0x80000000;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave / WK, wk = wave % WK;
  const int tiles_k = (p.Kw + TK - 1) / TK;
  // XCD-aware order: the tiles of one split (same pixel range: shared dY / X rows) form a contiguous range on one XCD's L2
  int bx = blockIdx.x, by = blockIdx.y;
  {
    const int gx = gridDim.x, nt = gx * gridDim.y, lin = by * gx + bx;
    const int fl = nt >> 3, rem = nt & 7, xcd = lin & 7;
    const int l2 = xcd * fl + (xcd < rem ? xcd : rem) + (lin >> 3);
    by = l2 / gx; bx = l2 - by * gx;
  }
  const int tile_n = bx / tiles_k, tile_k = bx - tile_n * tiles_k;
  const int n0 = tile_n * TN, k20 = tile_k * TK;
  const int mbeg = by * p.chunk, mend = min(p.M, mbeg + p.chunk);
  const int nsteps = (mend > mbeg) ? (mend - mbeg + 63) >> 6 : 0;

  const i32x4_t rsY = make_rsrc(p.dy, p.dy_bytes), rsX = make_rsrc(p.x, p.x_bytes);
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);     // LDS byte address of the ring

  // ---- staging map: lane -> row krow = 8*wave + lane/8 of every sub-image, 16-byte chunk position lane%8, swizzled source chunk
  const int krow = wave * 8 + (lane >> 3);
  const int hsw = ((((krow >> 1) & 1) | (((krow >> 3) & 1) << 1)) ^ ((krow & 1) << 1));
  const int schunk = (lane & 7) ^ (hsw << 1);                       // source 16-byte chunk of the 128-byte sub-image row
  const int yconst = (krow * p.N) * 2 + schunk * 16;                // + scalar (ms*N + n0 + 64*si)*2
  // per sub-image (tap, channel block) of the X tile -- wave-uniform
  const int taps = p.R * p.S;
  int xtap[SX], xc0[SX];
  bool xlive[SX];
#pragma unroll
  for (int si = 0; si < SX; ++si) {
    const int kk = k20 + 64 * si;
    xlive[si] = kk < p.Kw;
    const int t = (taps > 1) ? kk / p.C : 0;
    xtap[si] = xlive[si] ? t : 0;
    xc0[si] = kk - t * p.C;
  }
  const bool linear = (taps == 1 && p.stride == 1 && p.pad == 0 && p.H == p.Ho && p.W == p.Wo);   // Linear / 1x1 stride-1: src pixel == m
  const int HoWo = p.Ho * p.Wo, HW = p.H * p.W, csz = p.C * 2;
  const float inv_wo = 1.0f / (float)p.Wo, inv_ho = 1.0f / (float)p.Ho;
  int xb = 0, xoh = 0, xow = 0;
  if (!linear && nsteps > 0) {
    const int m = mbeg + krow;
    xb = fast_div(m, p.mul_howo);
    const int rem = m - xb * HoWo;
    xoh = fast_div(rem, p.mul_wo); xow = rem - xoh * p.Wo;
  }

  // ---- staging: prep(st) works out the step's vector / scalar offsets (VALU, once), issue(piece) emits ONE LDS-DMA piece.
  //      All pieces of stage st+NS-1 are issued at the head of step st (spreading them between the MFMA groups of the step
  //      measured 5 % slower: 130 vs 122 us on the stage-3 / stage-4 convs).
  int pv_y = OOB, ps_y = 0, pv_x[SX], ps_x[SX];
  unsigned pbase = lds0;
#pragma unroll
  for (int si = 0; si < SX; ++si) { pv_x[si] = OOB; ps_x[si] = 0; }
  auto prep = [&](int st) {
    const int ms = mbeg + st * 64;
    const bool rowok = (ms + krow) < mend;
    pbase = lds0 + (st % NS) * STAGE + wave * 1024;
    pv_y = rowok ? yconst : OOB;
    ps_y = (ms * p.N + n0) * 2;
    if (linear) {
      const int vox = rowok ? (krow * csz + schunk * 16) : OOB;
#pragma unroll
      for (int si = 0; si < SX; ++si) { pv_x[si] = xlive[si] ? vox : OOB; ps_x[si] = ms * csz + xc0[si] * 2; }
    } else {
      int vox = OOB, last_tap = -1;
#pragma unroll
      for (int si = 0; si < SX; ++si) {
        if (xlive[si] && xtap[si] != last_tap) {                    // wave-uniform: recompute the row offset only when the tap changes
          last_tap = xtap[si];
          const int tr = last_tap / p.S, ts = last_tap - tr * p.S;
          const int ih = __mul24(xoh, p.stride) - p.pad + tr, iw = __mul24(xow, p.stride) - p.pad + ts;
          const bool ok = rowok && (unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W;
          const int pixel = __mul24(xb, HW) + __mul24(ih, p.W) + iw;
          vox = ok ? __mul24(pixel, csz) + schunk * 16 : OOB;       // pixel * C * 2 < 2^31 (host check)
        }
        pv_x[si] = xlive[si] ? vox : OOB; ps_x[si] = xc0[si] * 2;   // (a dead sub-image -- tile columns past Kw -- still issues its piece, out of range: the vmcnt counts stay constant)
      }
      // advance this lane's pixel by 64 rows (exact: (x + 0.5) / d is never an integer)
      xow += 64;
      const int q_ = (int)(((float)xow + 0.5f) * inv_wo);
      xow -= __mul24(q_, p.Wo); xoh += q_;
      const int q2 = (int)(((float)xoh + 0.5f) * inv_ho);
      xoh -= __mul24(q2, p.Ho); xb += q2;
    }
  };
#if defined(__HIP_DEVICE_COMPILE__)
  auto issue = [&](int piece) {          // piece: 0 .. SY-1 = dY sub-images (columns past N belong to dW rows that are never stored), then X
    if (piece < SY) dma16(rsY, pbase + piece * SUB, pv_y, ps_y + piece * 128);
    else dma16(rsX, pbase + piece * SUB, pv_x[piece - SY], ps_x[piece - SY]);
  };
#else
  auto issue = [&](int) {};
#endif
  constexpr int P = SY + SX;                                        // pieces a wave issues per step (always all of them)
  auto stage_load = [&](int st) {
    prep(st);
#pragma unroll
    for (int pc = 0; pc < P; ++pc) issue(pc);
  };

  f32x4 acc[MT][NT];
#pragma unroll
  for (int i = 0; i < MT; ++i)
#pragma unroll
    for (int j = 0; j < NT; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};

  const int g = lane >> 4, li = lane & 15, q = li >> 2, pp = li & 3;
  auto compute = [&](int st) {
    const bf16_t* yb = reinterpret_cast<const bf16_t*>(smem + (st % NS) * STAGE);
    const bf16_t* xb_ = yb + (SY + wk) * (SUB / 2);
    typedef __attribute__((ext_vector_type(8))) short i16x8;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      const int k0 = ks * 32 + 8 * g + q;
      bf16x8 yf[MT], xf[NT];
#pragma unroll
      for (int j = 0; j < NT; ++j) {
        const int col = j * 16 + 4 * pp;
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(xb_ + kmaj64_off(k0, col)));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(xb_ + kmaj64_off(k0 + 4, col)));
        i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        xf[j] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i) {
        const int nl = wn * TNW + i * 16;                           // column of the dY tile
        const bf16_t* sb = yb + (nl >> 6) * (SUB / 2);
        const int col = (nl & 63) + 4 * pp;
        i16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(sb + kmaj64_off(k0, col)));
        i16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((i16x4 __attribute__((address_space(3)))*)(sb + kmaj64_off(k0 + 4, col)));
        i16x8 t = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        yf[i] = __builtin_bit_cast(bf16x8, t);
      }
#pragma unroll
      for (int i = 0; i < MT; ++i)
#pragma unroll
        for (int j = 0; j < NT; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(xf[j], yf[i], acc[i][j], 0, 0, 0);   // swapped: lane holds 4 consecutive dW columns
    }
  };

  // ---- ring: NS-1 steps of DMA in flight; the barrier ending step st publishes stage st+1 (counted vmcnt: younger pieces stay in flight)
#pragma unroll
  for (int s = 0; s < NS - 1; ++s)
    if (s < nsteps) stage_load(s);
  if (NS == 2 || nsteps <= 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(P * (NS - 2)) : "memory");
  __builtin_amdgcn_s_barrier();
  const bool dbg_nodma = p.dbg_noatomic & 4, dbg_nomma = p.dbg_noatomic & 8;     // VQA_WGRAD_DBG (measurement only, wrong results)
  for (int st = 0; st < nsteps; ++st) {
    const bool more = st + NS - 1 < nsteps && !dbg_nodma;
    if (more) stage_load(st + NS - 1);       // its slot was read in step st-1: every wave is past that step's barrier
    if (!dbg_nomma) compute(st);
    if (more && NS > 2) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)" ::"n"(P * (NS - 2)) : "memory");
    else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
  }

  // ---- flush this split's partial tile to its slab ws[split][N][Kw]; a single split (ws == nullptr) adds straight into dW
  //      (one workgroup per tile: no race, still deterministic)
  float* const ws = p.ws;
  float* slab = ws ? ws + (size_t)by * p.N * p.Kw : p.dw;
#pragma unroll
  for (int i = 0; i < MT; ++i) {
    const int n = n0 + wn * TNW + i * 16 + li;
#pragma unroll
    for (int j = 0; j < NT; ++j) {
      const int k2 = k20 + wk * 64 + j * 16 + g * 4;
      if (n < p.N && k2 < p.Kw) {
        f32x4* dst = reinterpret_cast<f32x4*>(slab + (size_t)n * p.Kw + k2);
        *dst = ws ? acc[i][j] : (*dst + acc[i][j]);
      }
    }
  }
}

// dw[i] += sum_s ws[s][i] in a fixed order (bit-reproducible); total % 4 == 0
__global__ __launch_bounds__(256) void wgrad_reduce_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= total4) return;
  const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
  f32x4 a = reinterpret_cast<const f32x4*>(dw)[i];
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {                                 // four independent loads in flight, summed in split order
    const f32x4 v0 = w4[(size_t)s * total4 + i], v1 = w4[(size_t)(s + 1) * total4 + i];
    const f32x4 v2 = w4[(size_t)(s + 2) * total4 + i], v3 = w4[(size_t)(s + 3) * total4 + i];
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; s < nsplit; ++s) a += w4[(size_t)s * total4 + i];
  reinterpret_cast<f32x4*>(dw)[i] = a;
}

// Same sum for many slabs of a SMALL gradient (64x576 from 256 workgroups, 128x1152 from 112 splits): one thread per float4 walks
// every slab in a dependent-latency loop there (64 rounds of 4 loads) while most CUs idle.  Here a block is 64 float4 columns x 16
// slab groups; group g sums its contiguous run of slabs in slab order, the 16 group sums are folded in group order through LDS:
// still one fixed summation order (bit-reproducible), 16x shorter chains, 16x more workgroups.
__global__ __launch_bounds__(1024) void wgrad_reduce2_kernel(const float* __restrict__ ws, float* __restrict__ dw, int nsplit, size_t total4) {
  const size_t i = (size_t)blockIdx.x * 64 + threadIdx.x;
  const int gq = threadIdx.y, per = (nsplit + 15) / 16, s0 = gq * per, s1 = min(nsplit, s0 + per);
  const f32x4* w4 = reinterpret_cast<const f32x4*>(ws);
  f32x4 a = {0.f, 0.f, 0.f, 0.f};
  if (i < total4) {
    int s = s0;
    for (; s + 4 <= s1; s += 4) {
      const f32x4 v0 = w4[(size_t)s * total4 + i], v1 = w4[(size_t)(s + 1) * total4 + i];
      const f32x4 v2 = w4[(size_t)(s + 2) * total4 + i], v3 = w4[(size_t)(s + 3) * total4 + i];
      a += v0; a += v1; a += v2; a += v3;
    }
    for (; s < s1; ++s) a += w4[(size_t)s * total4 + i];
  }
  __shared__ f32x4 sh[16][64];
  sh[gq][threadIdx.x] = a;
  __syncthreads();
  if (gq == 0 && i < total4) {
    f32x4 t = reinterpret_cast<const f32x4*>(dw)[i];
#pragma unroll
    for (int q = 0; q < 16; ++q) t += sh[q][threadIdx.x];
    reinterpret_cast<f32x4*>(dw)[i] = t;
  }
}

// fixed-order slab sums of a group of weight gradients in one launch (see wgrad_group_kernel)
__global__ __launch_bounds__(256) void wgrad_reduce_group_kernel(ReduceGroup g) {
  int j = 0;
#pragma unroll
  for (int q = 1; q < WG_MAXJOBS; ++q) if (q < g.n && (int)blockIdx.x >= g.blk0[q]) j = q;
  const size_t total4 = g.total4[j], i = (size_t)((int)blockIdx.x - g.blk0[j]) * 256 + threadIdx.x;
  if (i >= total4) return;
  const f32x4* w4 = reinterpret_cast<const f32x4*>(g.ws[j]);
  const int nsplit = g.nsplit[j];
  f32x4 a = reinterpret_cast<const f32x4*>(g.dw[j])[i];
  int s = 0;
  for (; s + 4 <= nsplit; s += 4) {
    const f32x4 v0 = w4[(size_t)s * total4 + i], v1 = w4[(size_t)(s + 1) * total4 + i];
    const f32x4 v2 = w4[(size_t)(s + 2) * total4 + i], v3 = w4[(size_t)(s + 3) * total4 + i];
    a += v0; a += v1; a += v2; a += v3;
  }
  for (; s < nsplit; ++s) a += w4[(size_t)s * total4 + i];
  reinterpret_cast<f32x4*>(g.dw[j])[i] = a;
}

// ------------------------------------------------------------------------------------------------
// weight packing: cast (+row pad) and [N][T][C] -> [C][T][N] transpose for the data-gradient GEMM
// ------------------------------------------------------------------------------------------------
template <typename T>
__global__ void pack_rows_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int K, int Kp) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * Kp) return;
  int n = (int)(i / Kp), k = (int)(i - (size_t)n * Kp);
  out[i] = from_f<T>(k < K ? in[(size_t)n * K + k] : 0.f);
}
template <typename T>
__global__ void pack_transpose_kernel(const float* __restrict__ in, T* __restrict__ out, int N, int TT, int C, int ldo, int col0, int flip) {
  // out[c][col0 + t*N + n] = in[n][flip ? TT-1-t : t][c]   (row stride ldo)
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * TT * C) return;
  int n = (int)(i % N); size_t r = i / N; int t = (int)(r % TT); int c = (int)(r / TT);
  const int ts = flip ? TT - 1 - t : t;
  out[(size_t)c * ldo + col0 + t * N + n] = from_f<T>(in[((size_t)n * TT + ts) * C + c]);
}

// All data-gradient operands of one step in ONE launch.  desc[d] = {src_off, dst_off, N, TT, C, ldo, col0, flip, blk0, 0} (int64):
// piece d covers workgroups blk0[d] .. blk0[d+1]-1; out[c][col0 + t*N + n] = in[n][flip ? TT-1-t : t][c] as in pack_transpose_kernel.
template <typename T>
__global__ __launch_bounds__(256) void pack_transpose_batch_kernel(const float* __restrict__ flat, T* __restrict__ outbase,
                                                                   const long long* __restrict__ desc, int nd) {
  // One workgroup = one 32 (n) x 32 (c) tile of one tap: rows are read along c (coalesced), transposed through LDS and written
  // along n (coalesced).  blocks of piece d: TT * ceil(N/32) * ceil(C/32), tap-major.
  int lo = 0, hi = nd - 1;                               // last piece whose blk0 <= blockIdx.x (uniform -> scalar loads)
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[(size_t)mid * 10 + 8] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + (size_t)lo * 10;
  const float* in = flat + d[0];
  T* out = outbase + d[1];
  const int N = (int)d[2], TT = (int)d[3], C = (int)d[4], ldo = (int)d[5], col0 = (int)d[6], flip = (int)d[7];
  const int tn = (N + 31) >> 5, tc = (C + 31) >> 5;
  int rel = blockIdx.x - (int)d[8];
  const int t = rel / (tn * tc); rel -= t * tn * tc;
  const int n0 = (rel / tc) << 5, c0 = (rel % tc) << 5;
  const int ts = flip ? TT - 1 - t : t;
  __shared__ float tile[32][33];
  const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;          // 8 rows per pass
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int n = n0 + r, c = c0 + tx;
    tile[r][tx] = (n < N && c < C) ? in[((size_t)n * TT + ts) * C + c] : 0.f;
  }
  __syncthreads();
#pragma unroll
  for (int r = ty; r < 32; r += 8) {
    const int c = c0 + r, n = n0 + tx;
    if (c < C && n < N) out[(size_t)c * ldo + col0 + t * N + n] = from_f<T>(tile[tx][r]);
  }
}

// Eval-mode Conv+BatchNorm folding for every conv of the network in ONE launch:
//   w'[n][k] = w[n][k] * gamma[n] / sqrt(running_var[n] + eps)   (cast to T),   b'[n] = beta[n] - running_mean[n] * (same scale)
// desc[d] = {w_off, gamma_off, beta_off (floats from flat), running_mean ptr, running_var ptr, N, K, dst_off (elements of wout),
//            bias_off (floats of bout), blk0}; piece d covers workgroups blk0[d] .. blk0[d+1]-1.
template <typename T>
__global__ void fold_bn_batch_kernel(const float* __restrict__ flat, T* __restrict__ wout, float* __restrict__ bout,
                                     const long long* __restrict__ desc, int nd, float eps) {
  int lo = 0, hi = nd - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (desc[(size_t)mid * 10 + 9] <= (long long)blockIdx.x) lo = mid; else hi = mid - 1;
  }
  const long long* d = desc + (size_t)lo * 10;
  const float* w = flat + d[0];
  const float* gamma = flat + d[1];
  const float* beta = flat + d[2];
  const float* rm = reinterpret_cast<const float*>(d[3]);
  const float* rv = reinterpret_cast<const float*>(d[4]);
  const int N = (int)d[5], K = (int)d[6];
  const size_t i = (size_t)(blockIdx.x - (int)d[9]) * blockDim.x + threadIdx.x;
  if (i >= (size_t)N * K) return;
  const int n = (int)(i / K);
  const float sc = gamma[n] / sqrtf(rv[n] + eps);
  wout[d[7] + i] = from_f<T>(w[i] * sc);
  if (i < (size_t)N) {
    const float s2 = gamma[i] / sqrtf(rv[i] + eps);
    bout[d[8] + i] = beta[i] - rm[i] * s2;
  }
}

// ------------------------------------------------------------------------------------------------
// C ABI
// ------------------------------------------------------------------------------------------------
// (the static attr_set flags below are per PROCESS: one process drives one GPU -- the launch model of this library, bench.py and
// torch.distributed; a process that drove several devices would have to set the attribute per device)
template <typename T, int BM, int BN, int LOADER, int NW = 4, int BK = GT<T>::BK, int OCC = 2, int ST = 2, int WIN = 0>
static int launch_igemm(const IGemmParams& p, hipStream_t st) {
  constexpr int SMEM = IGemmCfg<T, BM, BN, BK, ST, WIN>::SMEM;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&igemm_kernel<T, BM, BN, LOADER, NW, BK, OCC, ST, WIN>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  const int tiles = ((p.M + BM - 1) / BM) * ((p.N + BN - 1) / BN);
  hipLaunchKernelGGL((igemm_kernel<T, BM, BN, LOADER, NW, BK, OCC, ST, WIN>), dim3(tiles), dim3(NW * 64), SMEM, st, p);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

static void igemm_tile(int M, int N, int* bm, int* bn) {
  *bm = 128; *bn = N <= 64 ? 64 : 128;
  long tiles = (long)((M + *bm - 1) / *bm) * ((N + *bn - 1) / *bn);
  if (tiles < 384) { *bm = 64; *bn = 64; }
  const int force = vqa_env_int("VQA_IGEMM_TILE", 0);      // measurement only (tools/token_gemm_sweep.py): BM * 1000 + BN
  if (force == 128128 || force == 128064 || force == 64064) { *bm = force / 1000; *bn = force % 1000; }
}

// The ONE place that picks the template instantiation of a launch; vqa_igemm_variant() reports it to the host (parity tests assert
// that the shapes they run reach the kernels the benchmark times).  code = BM*10000 + BN*10 + flavour:
//   0 plain double-buffered LDS-DMA, 1 window loader (stride-1 3x3 pad-1, bf16).
// (Rounds 1-2 also built 8-wave 256x128 / 256x64 tiles, a 3-slot ring, two BK = 32 shapes and an epilogue that reduced the
// BatchNorm-backward sums; all measured slower on MI355X -- DESIGN.md section 3 -- and were deleted in round 3.)
static int igemm_variant(const IGemmParams& p, int loader, bool bf16) {
  int bm, bn; igemm_tile(p.M, p.N, &bm, &bn);
  if (loader == LOADER_STEM) return 128 * 10000 + 64 * 10;
  if (bf16) {
    const int win = vqa_env_int("VQA_IGEMM_WIN", 1);
    if (win && p.R == 3 && p.S == 3 && p.stride == 1 && p.pad == 1 && p.H == p.Ho && p.W == p.Wo && p.C % 64 == 0 && bm == 128 &&
        (long)p.B * p.H * p.W == (long)p.M)
      return 128 * 10000 + bn * 10 + 1;
  }
  return bm * 10000 + bn * 10;
}

template <typename T>
static int igemm_dispatch(const IGemmParams& p, int loader, hipStream_t st) {
  const int v = igemm_variant(p, loader, sizeof(T) == 2);
  if (loader == LOADER_STEM) return launch_igemm<T, 128, 64, LOADER_STEM>(p, st);
  if constexpr (sizeof(T) == 2) {
    switch (v) {
      case 128 * 10000 + 128 * 10 + 1: return launch_igemm<T, 128, 128, LOADER_NHWC, 4, 64, 2, 2, 1>(p, st);
      case 128 * 10000 + 64 * 10 + 1: return launch_igemm<T, 128, 64, LOADER_NHWC, 4, 64, 2, 2, 1>(p, st);
      default: break;
    }
  }
#ifdef VQA_ABLATION     // measurement builds only (tools/token_gemm_sweep.py; profiles/r04_token_gemm_sweep.txt: deeper rings are SLOWER)
  if constexpr (sizeof(T) == 2) {
    // dense Linear launches (1x1 "conv", stride 1): a deeper DMA ring -- these are 4-16 K steps of almost no MFMA work each
    const int ring = vqa_env_int("VQA_IGEMM_ST", 2);
    if (ring >= 3 && p.R == 1 && p.S == 1 && p.stride == 1 && p.pad == 0) {
      if (ring == 3) switch (v) {
        case 128 * 10000 + 128 * 10: return launch_igemm<T, 128, 128, LOADER_NHWC, 4, 64, 2, 3>(p, st);
        case 128 * 10000 + 64 * 10: return launch_igemm<T, 128, 64, LOADER_NHWC, 4, 64, 2, 3>(p, st);
        case 64 * 10000 + 64 * 10: return launch_igemm<T, 64, 64, LOADER_NHWC, 4, 64, 2, 3>(p, st);
        default: break;
      }
      else switch (v) {
        case 128 * 10000 + 128 * 10: return launch_igemm<T, 128, 128, LOADER_NHWC, 4, 64, 2, 4>(p, st);
        case 128 * 10000 + 64 * 10: return launch_igemm<T, 128, 64, LOADER_NHWC, 4, 64, 2, 4>(p, st);
        case 64 * 10000 + 64 * 10: return launch_igemm<T, 64, 64, LOADER_NHWC, 4, 64, 2, 4>(p, st);
        default: break;
      }
    }
  }
#endif
  switch (v) {
    case 128 * 10000 + 128 * 10: return launch_igemm<T, 128, 128, LOADER_NHWC>(p, st);
    case 128 * 10000 + 64 * 10: return launch_igemm<T, 128, 64, LOADER_NHWC>(p, st);
    case 64 * 10000 + 64 * 10: return launch_igemm<T, 64, 64, LOADER_NHWC>(p, st);
    default: return VQA_EARG;
  }
}

template <typename T, int BMW, int BNW, int LOADER>
static int launch_wgrad(const WgradParams& p, int nsplit, hipStream_t st) {
  constexpr int VEC = GT<T>::VEC;
  constexpr int SMEM = 2 * GT<T>::BKM * (BMW + BNW + (sizeof(T) == 2 ? 0 : 2 * VEC)) * (int)sizeof(T);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_kernel<T, BMW, BNW, LOADER>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  const int tiles = ((p.N + BMW - 1) / BMW) * ((p.Kw + BNW - 1) / BNW);
  hipLaunchKernelGGL((wgrad_kernel<T, BMW, BNW, LOADER>), dim3(tiles, nsplit), dim3(256), SMEM, st, p);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

template <typename T>
static int launch_wgrad_group(const WgradGroup& g, const ReduceGroup& r, hipStream_t st) {
  constexpr int VEC = GT<T>::VEC;
  constexpr int SMEM = 2 * GT<T>::BKM * (128 + 128 + (sizeof(T) == 2 ? 0 : 2 * VEC)) * (int)sizeof(T);
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_group_kernel<T, 128, 128, LOADER_NHWC>), hipFuncAttributeMaxDynamicSharedMemorySize, SMEM);
    attr_set = true;
  }
  hipLaunchKernelGGL((wgrad_group_kernel<T, 128, 128, LOADER_NHWC>), dim3(g.blk0[g.n]), dim3(256), SMEM, st, g);
  if (r.blk0[r.n] > 0) hipLaunchKernelGGL(wgrad_reduce_group_kernel, dim3(r.blk0[r.n]), dim3(256), 0, st, r);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}
static int launch_reduce(const WgradParams& p, int nsplit, hipStream_t st) {
  const size_t total4 = (size_t)p.N * p.Kw / 4;
  if (nsplit >= 32 && total4 <= 256 * 1024)        // few columns, many slabs: see wgrad_reduce2_kernel (choice depends on the shape only)
    hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(64, 16), 0, st, p.ws, p.dw, nsplit, total4);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, p.ws, p.dw, nsplit, total4);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

template <int TN, int TK, int NS>
static int launch_wgrad_dma(const WgradParams& p, int nsplit, hipStream_t st) {
  using Cfg = WgradDmaCfg<TN, TK, NS>;
  static bool attr_set = false;
  if (!attr_set) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&wgrad_dma_kernel<TN, TK, NS>), hipFuncAttributeMaxDynamicSharedMemorySize, Cfg::SMEM);
    attr_set = true;
  }
  const int tiles = ((p.N + TN - 1) / TN) * ((p.Kw + TK - 1) / TK);
  hipLaunchKernelGGL((wgrad_dma_kernel<TN, TK, NS>), dim3(tiles, nsplit), dim3(512), Cfg::SMEM, st, p);
  VQA_LAUNCH_CHECK();
  return p.ws ? launch_reduce(p, nsplit, st) : VQA_OK;
}

// Which weight-gradient kernel runs, with what tile and split -- the ONE place that decides (vqa_wgrad_plan reports it).
//   kind 1: wgrad_dma_kernel (bf16, 8 waves, one workgroup per CU) for the large problems: the CNN convs with >= 256 output
//           channels and the 64 -> 128 stage-2 entry conv.  Measured on MI355X at B = 512: stage 3 / 4 convs 178 -> 123 us.
//   kind 0: wgrad_kernel (4 waves, two workgroups per CU, register staging): fp32, the stem loader, the token-side Linears
//           (M ~ 10^4 rows: a 256x256 tile leaves most CUs idle, 30 vs 20 us) and the 128 -> 128 stage-2 convs (160 vs 174 us).
// Either way, with a workspace every split writes its own slab and wgrad_reduce_kernel adds them in a fixed order
// (bit-reproducible dW, no float atomics); without one (ws == NULL) kind 0 falls back to fp32 atomics.
struct WgradPlan { int kind, tn, tk, nsplit, chunk, xcd_order; long long ws_floats; };
static WgradPlan wgrad_plan(int dtype, int loader, int M, int N, int Kw, int B, int H, int W, int C, int R, int S, bool have_ws) {
  WgradPlan pl = {0, 64, 64, 1, 0, 1, 0};
  const bool conv = R * S > 1;
  const int dma_env = vqa_env_int("VQA_WGRAD_DMA", 1);     // measurement (-DVQA_ABLATION builds only): 0 never, 2 whenever eligible
  const long target_env = vqa_env_int("VQA_WGRAD_TARGET", 0);
  const long minrows_env = vqa_env_int("VQA_WGRAD_MINCHUNK", 0);
  const double flops = 2.0 * M * N * Kw;
  bool dma = have_ws && dma_env && dtype && loader == LOADER_NHWC && N >= 128 && (N % 8) == 0 && Kw >= 64 && (Kw % 64) == 0 &&
             (C % 64) == 0 && M >= 256 && (conv || C == Kw);
  if (dma && dma_env != 2) dma = flops >= 3e10 && (N >= 256 || (C % 128) != 0);
  if (dma) {
    // tile: least padded work, the 256x256 tile (128x64 wave tiles) preferred
    static const int cand[3][2] = {{256, 256}, {128, 256}, {256, 128}};
    const int force = vqa_env_int("VQA_WGRAD_TILE", -1);     // measurement: 0 / 1 / 2 forces a candidate
    double best = 1e300;
    for (int c = 0; c < 3; ++c) {
      if (force >= 0 && force != c) continue;
      const int tn = cand[c][0], tk = cand[c][1];
      const double work = (double)((N + tn - 1) / tn * tn) * ((Kw + tk - 1) / tk * tk) * (c == 0 ? 1.0 : 1.12);
      if (work < best) { best = work; pl.tn = tn; pl.tk = tk; }
    }
    pl.kind = 1;
    const long tiles = (long)((N + pl.tn - 1) / pl.tn) * ((Kw + pl.tk - 1) / pl.tk);
    const long target = target_env ? target_env : 256;                // one 8-wave workgroup per CU
    const long minrows = minrows_env ? minrows_env : 512;             // >= 8 steps per split: the flush is a 256 KB store per workgroup
    long nsplit = target / tiles;
    const long maxsplit = (M + minrows - 1) / minrows;
    if (nsplit > maxsplit) nsplit = maxsplit;
    if (nsplit < 1) nsplit = 1;
    int chunk = (int)((M + nsplit - 1) / nsplit);
    chunk = (chunk + 63) / 64 * 64;
    nsplit = (M + chunk - 1) / chunk;
    pl.nsplit = (int)nsplit; pl.chunk = chunk;
    pl.ws_floats = nsplit > 1 ? (long long)nsplit * N * Kw : 0;
    return pl;
  }
  // ---- 4-wave kernel.  tile: 128x128 when both dims allow it and (for multi-tap convs) a tile stays inside one tap;
  //      128 x 64 when only the dY side is wide (64-channel inputs: the 1x1 shortcut of stage 2)
  const bool big = loader == LOADER_NHWC && (N >= 128) && (conv ? (C % 128 == 0) : (Kw >= 128));
  const bool mid = !big && loader == LOADER_NHWC && (N >= 128) && (conv ? (C % 64 == 0) : (Kw >= 64));
  pl.tn = (big || mid) ? 128 : 64; pl.tk = big ? 128 : 64;
  const long tiles = (long)((N + pl.tn - 1) / pl.tn) * ((Kw + pl.tk - 1) / pl.tk);
  // split factor: more workgroups hide latency, but every split adds a tile's worth of flush traffic (measured: ~1000 workgroups
  // is the sweet spot for the 128x128 tile, ~2000 for the 64x64 tile)
  const long target = target_env ? target_env : (big ? 1024 : (mid ? 1536 : 2048));
  // XCD-aware workgroup order shares dY / X reads in one L2: a win (+30-40 %) where the inputs dwarf dW, a loss on stage 3 / 4
  const double in_bytes = ((double)M * N + (double)B * H * W * C) * (dtype ? 2 : 4), dw_bytes = (double)N * Kw * 4;
  pl.xcd_order = in_bytes >= 80.0 * dw_bytes;
  long nsplit = (target + tiles - 1) / tiles;
  // token-side GEMMs (M ~ 10^4) are flush-bound unless a split keeps >= ~1000 rows (16 steps) of work
  // (2048 for the Linears: they are launched 8 per group, ~one round of workgroups per group and half the slab traffic)
  const long minchunk = minrows_env ? minrows_env : ((!conv && M <= 65536) ? 2048 : 1024);
  const long maxsplit = (M + minchunk - 1) / minchunk;
  if (nsplit > maxsplit) nsplit = maxsplit;
  if (nsplit < 1) nsplit = 1;
  int chunk = (int)((M + nsplit - 1) / nsplit);
  chunk = (chunk + 63) / 64 * 64;
  nsplit = (M + chunk - 1) / chunk;
  pl.nsplit = (int)nsplit; pl.chunk = chunk;
  pl.ws_floats = (nsplit > 1 && (N * (long long)Kw) % 4 == 0) ? (long long)nsplit * N * Kw : 0;      // one split: a single += per element, already deterministic
  return pl;
}

extern "C" {

// number of M tiles igemm will use (= rows of the BN partial-statistics slab [tiles][2][N])
int vqa_igemm_mtiles(int M, int N, int loader) {
  int bm, bn; igemm_tile(M, N, &bm, &bn);
  if (loader == LOADER_STEM) bm = 128;
  return (M + bm - 1) / bm;
}

// Template instantiation vqa_igemm picks for this problem: BM*10000 + BN*10 + flavour (0 plain, 1 window loader).  Pure host function.
int vqa_igemm_variant(int dtype, int loader, int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo, int R, int S, int stride, int pad) {
  IGemmParams p;
  p.M = M; p.N = N; p.Kw = Kw; p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  return igemm_variant(p, loader, dtype != 0);
}

static int igemm_entry(int dtype, int loader, const void* a, const void* w, void* out, const float* bias,
              const void* addend, const void* addmask, const void* outmask, float out_scale, float* stats,
              int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo,
              int R, int S, int stride, int pad, int transposed, int relu, float drop_p, unsigned long long drop_seed,
              int stats_mode, hipStream_t st);

int vqa_igemm(int dtype, int loader, const void* a, const void* w, void* out, const float* bias,
              const void* addend, const void* addmask, const void* outmask, float* stats,
              int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo,
              int R, int S, int stride, int pad, int transposed, int relu, float drop_p, unsigned long long drop_seed,
              int stats_mode, hipStream_t st) {
  return igemm_entry(dtype, loader, a, w, out, bias, addend, addmask, outmask, 1.f, stats, M, N, Kw, B, H, W, C, Ho, Wo, R, S, stride, pad,
                     transposed, relu, drop_p, drop_seed, stats_mode, st);
}

// dx[M][Kin] = (dz[M][N] W) * (outact > 0) / (1 - drop_p): the data gradient of a Linear whose INPUT was relu(+dropout(p)) of the previous
// Linear -- the producer applies the consumer's mask (outact > 0 encodes both) and the keep scale 1 / (1 - p) to the bf16 value, bit-equal
// to vqa_bias_act_bwd on the stored gradient (models/text_encoder.py:309-317, models/vqa_model.py:74-82 backward).
int vqa_linear_dgrad_act(int dtype, const void* dz, const void* wt, void* dx, const void* addend, const void* outact, float drop_p,
                         int M, int Kin, int N, hipStream_t st) {
  if (!outact || !(drop_p >= 0.f && drop_p < 1.f)) return VQA_EARG;
  const float out_scale = drop_p > 0.f ? 1.f / (1.f - drop_p) : 1.f;      // the same float expression as bias_act_bwd's keep scale
  return igemm_entry(dtype, LOADER_NHWC, dz, wt, dx, nullptr, addend, nullptr, outact, out_scale, nullptr, M, Kin, N, M, 1, 1, N, 1, 1, 1, 1, 1, 0,
                     0, 0, 0.f, 0ull, 0, st);
}

static int igemm_entry(int dtype, int loader, const void* a, const void* w, void* out, const float* bias,
              const void* addend, const void* addmask, const void* outmask, float out_scale, float* stats,
              int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo,
              int R, int S, int stride, int pad, int transposed, int relu, float drop_p, unsigned long long drop_seed,
              int stats_mode, hipStream_t st) {
  if (M <= 0 || N <= 0 || !a || !w || !out) return VQA_EARG;
  const int VEC = dtype ? 8 : 4, BK = dtype ? 64 : 32;
  if (loader == LOADER_NHWC) {
    if (C % VEC) return VQA_EARG;
    if (R * S > 1 && (C % BK)) return VQA_EARG;
    if (Kw != R * S * C) return VQA_EARG;
  } else {
    if (Kw % BK || N != 64 || C != 3 || R != 7 || S != 7) return VQA_EARG;
  }
  if ((long)M != (long)B * Ho * Wo) return VQA_EARG;
  if (drop_p > 0.f && (unsigned long long)M * (unsigned long long)N >= (1ull << 32)) return VQA_EARG;   // 32-bit dropout counter
  if (stats && stats_mode && (M + 63) / 64 > VQA_ACC_MAX_PARTS) return VQA_EARG;   // one partial per M tile: the fixed-point total must not wrap (common.h)
  IGemmParams p;
  p.a = a; p.w = w; p.out = out; p.bias = bias; p.addend = addend; p.addmask = addmask; p.stats = stats; p.outmask = outmask;
  p.out_scale = out_scale;
  p.stats_mode = stats_mode;
  p.M = M; p.N = N; p.Kw = Kw; p.Kp = (Kw + BK - 1) / BK * BK;
  p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  p.transposed = transposed; p.relu = relu; p.drop_p = drop_p; p.drop_seed = drop_seed; p.a2 = nullptr;
  {
    const size_t es = dtype ? 2 : 4;
    const size_t ab = (size_t)B * H * W * C * es, wb = (size_t)N * Kw * es;
    if (loader == LOADER_NHWC && (ab >= 0x7fffffffull || wb >= 0x7fffffffull)) return VQA_EARG;
    p.a_bytes = (unsigned)(loader == LOADER_NHWC ? ab : 0); p.a2_bytes = p.a_bytes; p.w_bytes = (unsigned)wb;
  }
  if (transposed && stride != 1 && stride != 2) return VQA_EARG;
  {
    const unsigned long long one = 1ull << 40;
    if ((unsigned long long)M * (unsigned long long)(Ho * Wo) >= one) return VQA_EARG;
    p.mul_howo = (one + (unsigned long long)(Ho * Wo) - 1) / (unsigned long long)(Ho * Wo);
    p.mul_wo = (one + (unsigned long long)Wo - 1) / (unsigned long long)Wo;
  }
  for (int c = 0; c < 4; ++c) p.ntaps[c] = 0;
  p.dbg = vqa_env_int("VQA_IGEMM_DBG", 0);       // ablation switches (wrong results): compiled out unless -DVQA_ABLATION
  return dtype ? igemm_dispatch<bf16_t>(p, loader, st) : igemm_dispatch<float>(p, loader, st);
}

// Data gradient of a stride-2 convolution (R x R, pad) [+ the 1x1/2 shortcut's] in one launch, no redundant taps:
//   dx[B][Ho][Wo][N] = sum_taps dy[B][H][W][C] * wt[N][(r,s,C)]  (+ dyd[B][H][W][C] * wt[N][R*R*C + C]);  Ho = 2H, Wo = 2W.
// wt rows are [N][Ktot], Ktot = R*R*C (+ C with the shortcut), built with vqa_pack_transpose(ldo = Ktot, col0).
int vqa_dgrad_s2(int dtype, const void* dy, const void* dyd, const void* wt, void* out, int B, int H, int W, int C,
                 int Ho, int Wo, int N, int R, int pad, hipStream_t st) {
  const int VEC = dtype ? 8 : 4, BK = dtype ? 64 : 32;
  if (!dy || !wt || !out || (Ho & 1) || (Wo & 1) || C % BK || N % VEC || R > 3 || R < 1) return VQA_EARG;
  IGemmParams p;
  p.a = dy; p.a2 = dyd; p.w = wt; p.out = out; p.bias = nullptr; p.addend = nullptr; p.addmask = nullptr; p.stats = nullptr; p.outmask = nullptr; p.out_scale = 1.f;
  p.stats_mode = 0;
  p.N = N; p.Kw = R * R * C + (dyd ? C : 0); p.Kp = p.Kw; p.M = B * Ho * Wo;
  p.B = B; p.H = H; p.W = W; p.C = C; p.Ho = Ho; p.Wo = Wo; p.R = R; p.S = R; p.stride = 2; p.pad = pad;
  p.transposed = 1; p.relu = 0; p.drop_p = 0.f; p.drop_seed = 0;
  p.dbg = 8;     // plain workgroup order: the parity classes cost 1..5 taps, a contiguous tile range per XCD would unbalance the XCDs
  {
    const size_t es = dtype ? 2 : 4;
    const size_t ab = (size_t)B * H * W * C * es, wb = (size_t)N * p.Kw * es;
    if (ab >= 0x7fffffffull || wb >= 0x7fffffffull) return VQA_EARG;
    p.a_bytes = (unsigned)ab; p.a2_bytes = (unsigned)ab; p.w_bytes = (unsigned)wb;
  }
  for (int ph = 0; ph < 2; ++ph)
    for (int pw = 0; pw < 2; ++pw) {
      const int cls = ph * 2 + pw; int nt = 0;
      for (int r = 0; r < R; ++r) {
        if ((ph + pad - r) & 1) continue;
        for (int s = 0; s < R; ++s) {
          if ((pw + pad - s) & 1) continue;
          p.tap_koff[cls][nt] = (r * R + s) * C; p.tap_dh[cls][nt] = (ph + pad - r) / 2; p.tap_dw[cls][nt] = (pw + pad - s) / 2;
          p.tap_src[cls][nt] = 0; ++nt;
        }
      }
      if (dyd && cls == 0) { p.tap_koff[cls][nt] = R * R * C; p.tap_dh[cls][nt] = 0; p.tap_dw[cls][nt] = 0; p.tap_src[cls][nt] = 1; ++nt; }
      p.ntaps[cls] = nt;
      for (int t = nt; t < 5; ++t) { p.tap_koff[cls][t] = 0; p.tap_dh[cls][t] = 0; p.tap_dw[cls][t] = 0; p.tap_src[cls][t] = 0; }
    }
  const int class_rows = B * (Ho / 2) * (Wo / 2);
  {
    const unsigned long long one = 1ull << 40, hw = (unsigned long long)(Ho / 2) * (Wo / 2);
    if ((unsigned long long)class_rows * hw >= one) return VQA_EARG;
    p.mul_howo = (one + hw - 1) / hw;
    p.mul_wo = (one + (unsigned long long)(Wo / 2) - 1) / (unsigned long long)(Wo / 2);
  }
  const int bn = N <= 64 ? 64 : 128;
  const int tiles = 4 * ((class_rows + 127) / 128) * ((N + bn - 1) / bn);
  auto go = [&](auto kern, int smem) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    hipLaunchKernelGGL(kern, dim3(tiles), dim3(256), smem, st, p);
  };
  if (dtype) { if (bn == 64) go(&igemm_kernel<bf16_t, 128, 64, LOADER_DGRAD2>, IGemmCfg<bf16_t, 128, 64>::SMEM);
               else go(&igemm_kernel<bf16_t, 128, 128, LOADER_DGRAD2>, IGemmCfg<bf16_t, 128, 128>::SMEM); }
  else { if (bn == 64) go(&igemm_kernel<float, 128, 64, LOADER_DGRAD2>, IGemmCfg<float, 128, 64>::SMEM);
         else go(&igemm_kernel<float, 128, 128, LOADER_DGRAD2>, IGemmCfg<float, 128, 128>::SMEM); }
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

// plan of vqa_wgrad for this problem when a large enough workspace is handed in (host-only query): *kind 1 = LDS-DMA kernel,
// 0 = 4-wave kernel; tile, split count and the workspace floats the deterministic two-pass split needs (0: single split)
int vqa_wgrad_plan(int dtype, int loader, int M, int N, int Kw, int B, int H, int W, int C, int R, int S, long long* ws_floats, int* kind,
                   int* tile_n, int* tile_k, int* nsplit) {
  const WgradPlan pl = wgrad_plan(dtype, loader, M, N, Kw, B, H, W, C, R, S, true);
  if (ws_floats) *ws_floats = pl.ws_floats;
  if (kind) *kind = pl.kind;
  if (tile_n) *tile_n = pl.tn;
  if (tile_k) *tile_k = pl.tk;
  if (nsplit) *nsplit = pl.nsplit;
  return VQA_OK;
}

int vqa_wgrad(int dtype, int loader, const void* dy, const void* x, float* dw,
              int M, int N, int Kw, int B, int H, int W, int C, int Ho, int Wo,
              int R, int S, int stride, int pad, float* ws, long long ws_floats, hipStream_t st) {
  if (M <= 0 || N <= 0 || !dy || !x || !dw) return VQA_EARG;
  const int VEC = dtype ? 8 : 4;
  if (N % VEC) return VQA_EARG;
  if (loader == LOADER_NHWC) {
    if (C % VEC || Kw != R * S * C) return VQA_EARG;
  } else if (Kw != 147) return VQA_EARG;
  if ((long)M != (long)B * Ho * Wo) return VQA_EARG;
  WgradParams p;
  p.dy = dy; p.x = x; p.dw = dw; p.ws = nullptr; p.M = M; p.N = N; p.Kw = Kw; p.B = B; p.H = H; p.W = W; p.C = C;
  p.Ho = Ho; p.Wo = Wo; p.R = R; p.S = S; p.stride = stride; p.pad = pad;
  {
    const size_t es = dtype ? 2 : 4;
    const size_t yb = (size_t)M * N * es, xb = (size_t)B * H * W * C * es;
    if (yb >= 0x7fffffffull || (loader == LOADER_NHWC && xb >= 0x7fffffffull)) return VQA_EARG;
    if ((double)M * ((double)Ho * Wo) >= 1099511627776.0) return VQA_EARG;
    if (loader == LOADER_NHWC && (long)B * H * W >= (1l << 23)) return VQA_EARG;   // pixel indices go through 24-bit multiplies
    p.dy_bytes = (unsigned)yb; p.x_bytes = (unsigned)(loader == LOADER_NHWC ? xb : 0);
    const unsigned long long one = 1ull << 40;
    p.mul_howo = (one + (unsigned long long)(Ho * Wo) - 1) / (unsigned long long)(Ho * Wo);
    p.mul_wo = (one + (unsigned long long)Wo - 1) / (unsigned long long)Wo;
  }
  // plan with the workspace if it is large enough for that plan, else the workspace-free plan (fp32 atomics)
  WgradPlan pl = wgrad_plan(dtype, loader, M, N, Kw, B, H, W, C, R, S, ws != nullptr);
  if (pl.ws_floats > 0 && (!ws || ws_floats < pl.ws_floats)) pl = wgrad_plan(dtype, loader, M, N, Kw, B, H, W, C, R, S, false), pl.ws_floats = 0;
  p.ws = pl.ws_floats > 0 ? ws : nullptr;
  p.chunk = pl.chunk;
  if (pl.kind == 1) {
    const int dbg_env = vqa_env_int("VQA_WGRAD_DBG", 0);
    p.dbg_noatomic = dbg_env;
    if (pl.tn == 256 && pl.tk == 256) return launch_wgrad_dma<256, 256, 2>(p, pl.nsplit, st);
    if (pl.tn == 128 && pl.tk == 256) return launch_wgrad_dma<128, 256, 2>(p, pl.nsplit, st);
    return launch_wgrad_dma<256, 128, 2>(p, pl.nsplit, st);
  }
  if (loader == LOADER_NHWC && R * S > 1 && (C % pl.tk)) return VQA_EARG;
  const int nostage = vqa_env_int("VQA_WGRAD_NOSTAGE", 0);
  p.dbg_noatomic = nostage;      // A/B switches: bit 0 = atomic flush straight from the accumulators, bit 1 = plain (not XCD-aware) workgroup order
  if (!pl.xcd_order) p.dbg_noatomic |= 2;
  int rc;
  const int ns = pl.nsplit;
  if (loader == LOADER_STEM) rc = dtype ? launch_wgrad<bf16_t, 64, 64, LOADER_STEM>(p, ns, st) : launch_wgrad<float, 64, 64, LOADER_STEM>(p, ns, st);
  else if (pl.tn == 128 && pl.tk == 128) rc = dtype ? launch_wgrad<bf16_t, 128, 128, LOADER_NHWC>(p, ns, st) : launch_wgrad<float, 128, 128, LOADER_NHWC>(p, ns, st);
  else if (pl.tn == 128) rc = dtype ? launch_wgrad<bf16_t, 128, 64, LOADER_NHWC>(p, ns, st) : launch_wgrad<float, 128, 64, LOADER_NHWC>(p, ns, st);
  else rc = dtype ? launch_wgrad<bf16_t, 64, 64, LOADER_NHWC>(p, ns, st) : launch_wgrad<float, 64, 64, LOADER_NHWC>(p, ns, st);
  if (rc != VQA_OK) return rc;
  return p.ws ? launch_reduce(p, ns, st) : VQA_OK;
}

// ---- grouped Linear weight gradients: dw_j[N_j][K_j] += dy_j[M_j][N_j]^T x_j[M_j][K_j], j < njobs <= 8, ONE launch + ONE reduce launch.
// Every job must be one the planner gives the 4-wave 128x128 kernel with a split workspace (the token-side Linears); the value of
// each dw_j is bit-identical to its own vqa_wgrad call (same tiles, same splits, same slab order).
// vqa_wgrad_group_ws: floats of the shared workspace, or -1 when a job does not qualify (the caller then launches one by one).
static bool wgrad_group_plan(int dtype, int M, int N, int Kw, WgradPlan* pl) {
  if (M <= 0 || N <= 0 || Kw <= 0 || (N % (dtype ? 8 : 4)) || (Kw % (dtype ? 8 : 4))) return false;
  *pl = wgrad_plan(dtype, LOADER_NHWC, M, N, Kw, M, 1, 1, Kw, 1, 1, true);
  // (a single split needs no slab: one += per element, already a fixed order)
  return pl->kind == 0 && pl->tn == 128 && pl->tk == 128 && (pl->ws_floats > 0 || pl->nsplit == 1) && (pl->ws_floats % 4) == 0 &&
         (size_t)M * N * (dtype ? 2 : 4) < 0x7fffffffull && (size_t)M * Kw * (dtype ? 2 : 4) < 0x7fffffffull && (long)M < (1l << 23);
}
long long vqa_wgrad_group_ws(int dtype, int njobs, const int* M, const int* N, const int* Kw) {
  if (njobs <= 0 || njobs > WG_MAXJOBS || !M || !N || !Kw) return -1;
  long long tot = 0;
  for (int j = 0; j < njobs; ++j) { WgradPlan pl; if (!wgrad_group_plan(dtype, M[j], N[j], Kw[j], &pl)) return -1; tot += pl.ws_floats; }
  return tot;
}
int vqa_wgrad_group(int dtype, int njobs, const void* const* dy, const void* const* x, float* const* dw, const int* M, const int* N,
                    const int* Kw, float* ws, long long ws_floats, hipStream_t st) {
  if (njobs <= 0 || njobs > WG_MAXJOBS || !dy || !x || !dw || !M || !N || !Kw) return VQA_EARG;
  const int nostage = vqa_env_int("VQA_WGRAD_NOSTAGE", 0);
  WgradGroup g; ReduceGroup r;
  g.n = r.n = njobs; g.blk0[0] = r.blk0[0] = 0;
  long long off = 0;
  for (int j = 0; j < njobs; ++j) {
    WgradPlan pl;
    if (!dy[j] || !x[j] || !dw[j] || !wgrad_group_plan(dtype, M[j], N[j], Kw[j], &pl)) return VQA_EARG;
    if (pl.ws_floats > 0 && (!ws || off + pl.ws_floats > ws_floats)) return VQA_EARG;
    WgradParams& p = g.p[j];
    p.dy = dy[j]; p.x = x[j]; p.dw = dw[j]; p.ws = pl.ws_floats > 0 ? ws + off : nullptr; p.M = M[j]; p.N = N[j]; p.Kw = Kw[j];
    p.B = M[j]; p.H = 1; p.W = 1; p.C = Kw[j]; p.Ho = 1; p.Wo = 1; p.R = 1; p.S = 1; p.stride = 1; p.pad = 0;
    p.chunk = pl.chunk; p.dbg_noatomic = nostage | (pl.xcd_order ? 0 : 2);
    const size_t es = dtype ? 2 : 4;
    p.dy_bytes = (unsigned)((size_t)M[j] * N[j] * es); p.x_bytes = (unsigned)((size_t)M[j] * Kw[j] * es);
    p.mul_howo = 1ull << 40; p.mul_wo = 1ull << 40;
    g.gx[j] = ((N[j] + 127) / 128) * ((Kw[j] + 127) / 128); g.gy[j] = pl.nsplit;
    g.blk0[j + 1] = g.blk0[j] + g.gx[j] * pl.nsplit;
    r.ws[j] = p.ws; r.dw[j] = dw[j]; r.nsplit[j] = pl.nsplit; r.total4[j] = p.ws ? (unsigned)((size_t)N[j] * Kw[j] / 4) : 0u;
    r.blk0[j + 1] = r.blk0[j] + (int)((r.total4[j] + 255) / 256);      // a single-split job has no slab and no reduce blocks
    off += pl.ws_floats;
  }
  return dtype ? launch_wgrad_group<bf16_t>(g, r, st) : launch_wgrad_group<float>(g, r, st);
}

// dw[i] += sum_{s < nslabs} ws[s][i], i < n (n % 4 == 0), slabs added in index order: the second pass of every deterministic
// split weight gradient (vqa_wgrad calls it itself; the stage-1 patch kernel and the stem kernels use it through this entry)
int vqa_slab_reduce(const float* ws, float* dw, int nslabs, long long n, hipStream_t st) {
  if (!ws || !dw || nslabs <= 0 || n <= 0 || (n % 4)) return VQA_EARG;
  const size_t total4 = (size_t)n / 4;
  // few columns, many slabs: split the slab walk over 16 thread groups (fixed fold order); the choice depends on the shape only
  if (nslabs >= 32 && total4 <= 256 * 1024)
    hipLaunchKernelGGL(wgrad_reduce2_kernel, dim3((unsigned)((total4 + 63) / 64)), dim3(64, 16), 0, st, ws, dw, nslabs, total4);
  else
    hipLaunchKernelGGL(wgrad_reduce_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, st, ws, dw, nslabs, total4);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

// out[N][Kp] (T) = cast(in[N][K] fp32), zero padded rows
int vqa_pack_rows(int dtype, const float* in, void* out, int N, int K, int Kp, hipStream_t st) {
  if (!in || !out || N <= 0 || K <= 0 || Kp < K) return VQA_EARG;
  size_t total = (size_t)N * Kp;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype) hipLaunchKernelGGL(pack_rows_kernel<bf16_t>, grid, dim3(256), 0, st, in, (bf16_t*)out, N, K, Kp);
  else hipLaunchKernelGGL(pack_rows_kernel<float>, grid, dim3(256), 0, st, in, (float*)out, N, K, Kp);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}
// out[c][col0 + t*N + n] (T, row stride ldo >= col0 + T*N) = in[N][T][C] fp32
int vqa_pack_transpose(int dtype, const float* in, void* out, int N, int TT, int C, int ldo, int col0, int flip, hipStream_t st) {
  if (!in || !out || N <= 0 || TT <= 0 || C <= 0 || ldo < col0 + TT * N) return VQA_EARG;
  size_t total = (size_t)N * TT * C;
  dim3 grid((unsigned)((total + 255) / 256));
  if (dtype) hipLaunchKernelGGL(pack_transpose_kernel<bf16_t>, grid, dim3(256), 0, st, in, (bf16_t*)out, N, TT, C, ldo, col0, flip);
  else hipLaunchKernelGGL(pack_transpose_kernel<float>, grid, dim3(256), 0, st, in, (float*)out, N, TT, C, ldo, col0, flip);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

// Fold eval-mode BatchNorm into the preceding conv's weights for nd convs at once; desc is a DEVICE table [nd][10] int64
// {w_off, gamma_off, beta_off, running_mean pointer, running_var pointer, N, K, dst_off, bias_off, blk0} (see the kernel).
int vqa_fold_bn_batch(int dtype, const float* flat, void* wout, float* bout, const long long* desc, int nd, int total_blocks, float eps,
                      hipStream_t st) {
  if (!flat || !wout || !bout || !desc || nd <= 0 || total_blocks <= 0) return VQA_EARG;
  if (dtype) hipLaunchKernelGGL(fold_bn_batch_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, flat, (bf16_t*)wout, bout, desc, nd, eps);
  else hipLaunchKernelGGL(fold_bn_batch_kernel<float>, dim3(total_blocks), dim3(256), 0, st, flat, (float*)wout, bout, desc, nd, eps);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

// Batched form of vqa_pack_transpose: nd pieces described by a DEVICE table desc [nd][10] int64
// {src_off (floats from flat), dst_off (elements from out), N, TT, C, ldo, col0, flip, blk0, 0}; blk0 = running sum of
// TT * ceil(N/32) * ceil(C/32) (one workgroup per 32x32 tile of a tap) and total_blocks its final value.  The caller guarantees the pieces stay inside flat / out.
int vqa_pack_transpose_batch(int dtype, const float* flat, void* out, const long long* desc, int nd, int total_blocks, hipStream_t st) {
  if (!flat || !out || !desc || nd <= 0 || total_blocks <= 0) return VQA_EARG;
  if (dtype) hipLaunchKernelGGL(pack_transpose_batch_kernel<bf16_t>, dim3(total_blocks), dim3(256), 0, st, flat, (bf16_t*)out, desc, nd);
  else hipLaunchKernelGGL(pack_transpose_batch_kernel<float>, dim3(total_blocks), dim3(256), 0, st, flat, (float*)out, desc, nd);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

}  // extern "C"
