// 256 x 256 x 64 bf16 GEMM tile with the 8-phase schedule of cdna_hip_programming.md section 5 ("The 256^2 8-phase template"),
// written for this model's convolutions (round 4).  Why a second GEMM core beside igemm_kernel: that kernel is the guide's
// "step-3 structure" (128^2 tile, 4 waves, one or two barriers per K step, 2 workgroups per CU), whose ceiling is ~0.9 PFLOP/s on
// MI355X whatever is tuned inside it -- three rounds of tile / ring / schedule variants in gemm_conv.hip all landed at 0.82-1.04.
// This structure is different in kind:
//   * ONE 8-wave workgroup per CU, 2 (M) x 4 (N) waves, 128 x 64 outputs per wave = 32 accumulator tiles (128 VGPRs);
//   * the two waves that share a SIMD belong to different M halves and run ONE BARRIER APART: while one issues its LDS reads and
//     LDS-DMA pieces, its partner issues 16 MFMAs (one 64 x 32 quadrant of its outputs x K = 64) -- the matrix pipe of every SIMD
//     always has a wave in its MFMA segment;
//   * a K tile (64) is consumed in 4 phases (quadrants 00, 01, 11, 10); operands arrive as four 16 KB half-tiles per K tile
//     (A rows needed first / second, B columns needed first / second), ONE half-tile staged per phase by LDS-DMA into a double
//     buffer (8 x 16 KB = 128 KB), four half-tiles in flight, counted vmcnt waits, raw s_barrier (never vmcnt(0) in the loop);
//   * LDS rows are 128 bytes, XOR-swizzled on the DMA source side: the 16-byte slot index (3 bits) ^= (row >> 1) & 7.  A ds_read_b128
//     is served in four groups of 16 lanes ({0-3, 12-15, 20-27}, ...: MI355X_MICROARCH.md, LDS) over a 256-byte bank window = two rows;
//     with this map the 16 lanes of every group hit 16 different slots of the window: conflict-free fragment reads.  (Measured against the
//     guide's st_16x32 swizzle -- slot bit 1 ^= row bit 2, 4-way conflicted by the same table: 1193 / 1307 vs 1220 / 1291 TFLOP/s at
//     4096^3 / 8192^3, i.e. NEUTRAL: the fragment reads are not what bounds this schedule.  Kept: it costs two address registers.)
// Hazard bookkeeping (phase p of K tile t = global phase 4t + p; group 1 = waves 4-7 runs one barrier behind group 0):
//   staged in phase | half-tile      | its buffer was last read in | first read in
//   P1(t)           | B second (t+1) | P2(t-1)                     | P2(t+1)   (waited for in P1(t+1))
//   P2(t)           | A second (t+1) | P3(t-1)                     | P3(t+1)   (waited for in P2(t+1))
//   P3(t)           | A first  (t+2) | P1(t)                       | P1(t+2)   (waited for in P4(t+1))
//   P4(t)           | B first  (t+2) | P1(t)                       | P1(t+2)   (waited for in P4(t+1))
// i.e. every buffer is restaged >= 2 phases after its last read (the guide's WAR rule incl. the one-barrier stagger) and read
// one phase after the wait + barrier that retires its DMA (RAW rule); each wait is vmcnt(8): the wave's own 2 pieces of the four
// youngest half-tiles stay in flight.
#include <type_traits>
#include "common.h"

namespace {
typedef int i32x4_g8 __attribute__((ext_vector_type(4)));
constexpr int G8_BM = 256, G8_BN = 256, G8_BK = 64;
constexpr int G8_HALF = 128 * 128;                   // bytes of a half-tile: 128 rows x 64 k x 2
constexpr int G8_LDS = 8 * G8_HALF;                  // 2 buffers x {A first, A second, B first, B second}
constexpr int OOB_G8 = (int)0x80000000;

__device__ __forceinline__ void dma16_g8(i32x4_g8 rs, unsigned lds_addr, int voff, int soff) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("s_mov_b32 m0, %2\n\ts_nop 0\n\tbuffer_load_dwordx4 %0, %1, %3 offen lds" ::"v"(voff), "s"(rs), "s"(lds_addr), "s"(soff) : "memory");
}
// the same with M0 already holding the LDS base of the first piece of a group: piece h of the group lands at M0 + OFF (the instruction offset is
// added to BOTH addresses, so the caller's vector offset carries -OFF)
template <int OFF>
__device__ __forceinline__ void dma16_g8_off(i32x4_g8 rs, int voff, int soff) {
  soff = __builtin_amdgcn_readfirstlane(soff);
  asm volatile("buffer_load_dwordx4 %0, %1, %2 offen offset:%3 lds" ::"v"(voff), "s"(rs), "s"(soff), "n"(OFF) : "memory");
}
__device__ __forceinline__ void set_m0_g8(unsigned lds_addr) {
  lds_addr = __builtin_amdgcn_readfirstlane(lds_addr);
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0" ::"s"(lds_addr) : "memory");
}
}  // namespace

struct G8Params {
  const bf16_t* A; const bf16_t* B; bf16_t* C;        // A [M][K], B [N][K] (both K-contiguous), C [M][N]
  int M, N, K; unsigned a_bytes, b_bytes;
  int tiles_n, ntiles;
};

// tile row (0 .. 255) of buffer row r0 (multiple of 8) of half-tile `which` (0 A first, 1 A second, 2 B first, 3 B second)
__device__ __forceinline__ int g8_row_of(int which, int r0) {
  if (which == 0) return r0 < 64 ? r0 : r0 + 64;                    // rows 0-63 | 128-191: the first 64 rows of each M half
  if (which == 1) return r0 < 64 ? r0 + 64 : r0 + 128;              // rows 64-127 | 192-255
  const int q = r0 >> 5, i = r0 & 31;                               // B: 32 of the 64 columns of each of the 4 column groups
  return q * 64 + i + (which == 3 ? 32 : 0);
}

// (Measured and rejected, round 4: issuing the phase's two LDS-DMA pieces in the MIDDLE of its MFMA segment with vmcnt(6) waits --
// 1112 vs 1220 TFLOP/s at 4096^3, 141 vs 128 us on the stage-3 conv: profiles/r04_gemm8p_bench.txt.)
__global__ __launch_bounds__(512, 2) void gemm8p_kernel(G8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 2, wc = w & 3;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  // XCD-aware tile order (bijective form): consecutive workgroup ids go round-robin over the 8 XCDs; give each XCD a contiguous
  // range of tiles (N fastest), so the tiles that share A rows / B columns meet in one L2
  int tile;
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * G8_BM, n0 = tn * G8_BN;
  const unsigned long long aa = (unsigned long long)p.A, ba = (unsigned long long)p.B;
  const i32x4_g8 rsA = {(int)(unsigned)aa, (int)((unsigned)(aa >> 32) & 0xffffu), (int)p.a_bytes, 0x00020000};
  const i32x4_g8 rsB = {(int)(unsigned)ba, (int)((unsigned)(ba >> 32) & 0xffffu), (int)p.b_bytes, 0x00020000};

  // ---- DMA plan: a piece = 8 buffer rows x 128 bytes; wave w stages pieces 2w, 2w+1 of every half-tile.  Lane: row dr of the piece,
  //      LDS slot dc; the SOURCE chunk is dc ^ f(row), f = (row >> 1) & 7 = (dr >> 1) | (piece parity << 2) (the swizzle lives on the source side).
  const int dr = lane >> 3, dc = lane & 7;
  const int voff0 = dr * p.K * 2 + ((dc ^ (dr >> 1)) << 4), voff1 = dr * p.K * 2 + ((dc ^ (dr >> 1) ^ 4) << 4);   // even / odd piece (row bit 3)
  const int nkt = p.K / G8_BK;
  auto stage = [&](int which, int t) {             // half-tile `which` of K tile t into buffer t & 1 (t >= nkt: out of range -> zeros)
    const unsigned base = lds0 + (unsigned)(((t & 1) * 4 + which) * G8_HALF);
    const bool ok = t < nkt;
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      const int pc = 2 * w + h, row = g8_row_of(which, pc * 8);
      const int so = ((which < 2 ? m0 : n0) + row) * p.K * 2 + t * (G8_BK * 2);
      dma16_g8(which < 2 ? rsA : rsB, base + (unsigned)(pc * 1024), ok ? (h ? voff1 : voff0) : OOB_G8, ok ? so : 0);
    }
  };

  // ---- fragment read addresses (bytes, relative to a half-tile buffer): row = (wave's first row) + 16 i + li, slot = (4 kk + g) ^ f
  const int fl = (li >> 1) & 7;
  const unsigned sk[2] = {(unsigned)((g ^ fl) << 4), (unsigned)((g ^ fl ^ 4) << 4)};       // slot of k half kk (the XOR may flip bit 2: not an immediate)
  const unsigned ra = (unsigned)((wr * 64 + li) * 128);      // + i * 2048 + sk[kk]
  const unsigned rb = (unsigned)((wc * 32 + li) * 128);      // + j * 2048 + sk[kk]

  f32x4 acc[8][4];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a0[4][2], a1[4][2], b0[2][2], b1[2][2];

#define G8_RD(dst, which, off) dst = *reinterpret_cast<const bf16x8*>(smem + (d * 4 + (which)) * G8_HALF + (off))
#define G8_BAR() __builtin_amdgcn_s_barrier()
// (k half OUTERMOST: the two MFMAs of an accumulator are 6-8 instructions apart.  With kk innermost hipcc emitted tmp = mfma(.., acc);
//  acc = mfma(.., tmp): every second MFMA waited for the one in front of it)
#define G8_MFMA_R(AF, BFR, MI, NJ, I0, I1)                                                                  \
  _Pragma("unroll") for (int kk = 0; kk < 2; ++kk)                                                          \
    _Pragma("unroll") for (int i = (I0); i < (I1); ++i) _Pragma("unroll") for (int j = 0; j < 2; ++j)      \
          acc[(MI) + i][(NJ) + j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(BFR[j][kk], AF[i][kk], acc[(MI) + i][(NJ) + j], 0, 0, 0)
#define G8_WAIT() asm volatile("s_waitcnt vmcnt(8)" ::: "memory")
// one phase's compute segment
#define G8_COMPUTE(AF, BFR, MI, NJ, NI)                                                                     \
  __builtin_amdgcn_sched_barrier(0);                                                                        \
  __builtin_amdgcn_s_setprio(1);                                                                            \
  G8_MFMA_R(AF, BFR, MI, NJ, 0, NI);                                                                        \
  __builtin_amdgcn_s_setprio(0);                                                                            \
  __builtin_amdgcn_sched_barrier(0)

  // ---- prologue: K tile 0 complete + the first two half-tiles of K tile 1 (the steady-state lead)
  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0); stage(0, 1); stage(2, 1);
  asm volatile("s_waitcnt vmcnt(8)" ::: "memory");                  // A first (0), B first (0) landed (this wave's pieces)
  G8_BAR();
  if (wr == 1) G8_BAR();                                            // group 1 runs one barrier behind group 0 from here on

  for (int t = 0; t < nkt; ++t) {
    const int d = t & 1;
    // ---------------- P1: quadrant (0, 0)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) G8_RD(b0[j][kk], 2, rb + j * 2048 + sk[kk]);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) G8_RD(a0[i][kk], 0, ra + i * 2048 + sk[kk]);
    stage(3, t + 1);
    G8_WAIT();                                                      // B second (t) landed -> read in P2
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_COMPUTE(a0, b0, 0, 0, 4);
    G8_BAR();
    // ---------------- P2: quadrant (0, 1)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) G8_RD(b1[j][kk], 3, rb + j * 2048 + sk[kk]);
    stage(1, t + 1);
    G8_WAIT();                                                      // A second (t) landed -> read in P3
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_COMPUTE(a0, b1, 0, 2, 4);
    G8_BAR();
    // ---------------- P3: quadrant (1, 1)
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) G8_RD(a1[i][kk], 1, ra + i * 2048 + sk[kk]);
    stage(0, t + 2);
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    G8_COMPUTE(a1, b1, 4, 2, 4);
    G8_BAR();
    // ---------------- P4: quadrant (1, 0)
    stage(2, t + 2);
    G8_WAIT();                                                      // A first, B first (t + 1) landed -> read in P1 of the next K tile
    G8_BAR();
    G8_COMPUTE(a1, b0, 4, 0, 4);
    G8_BAR();
  }
  if (wr == 0) G8_BAR();                                            // (group 0 waits for group 1's last barrier)
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the out-of-range tail pieces
#undef G8_RD

  // ---- epilogue: acc[mi][nj][r] = C[m0 + 128 wr + 16 mi + li][n0 + 64 wc + 16 nj + 4 g + r]  (operands swapped: 4 consecutive columns per lane)
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const size_t row = (size_t)(m0 + wr * 128 + mi * 16 + li);
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      const int col = n0 + wc * 64 + nj * 16 + 4 * g;
      u32x2_t o;
      o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[mi][nj][0], acc[mi][nj][1]}, bf16x2_t));
      o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[mi][nj][2], acc[mi][nj][3]}, bf16x2_t));
      *reinterpret_cast<u32x2_t*>(p.C + row * p.N + col) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// gemm4w_kernel (round 4, the OTHER structure: tools/bench_gemm8p.py): the same 256 x 256 x 64 tile on FOUR waves, one per SIMD, each a
// 128 x 128 wave tile = 64 accumulator tiles (256 registers in the accumulator half of the 512-register file).  profiles/r04_conv8p_diag.txt:
// at two waves per SIMD the 8-phase loop is bound by its own phase structure (a fragment-read + wait + barrier chain in front of every
// 16-MFMA segment); here a K tile is ONE barrier and 128 MFMAs per wave:
//   fragments are double-buffered per 32-deep K step (16 ds_read_b128 per step, issued one step ahead, counted lgkmcnt);
//   the barrier sits in the MIDDLE of the second K step's MFMAs: by then every wave has read all of K tile t, so K tile t + 2 is staged
//   into that buffer right behind it (1.5 K tiles = ~3000 cycles of lead for the LDS-DMA) and the first fragments of K tile t + 1 are read
//   under the remaining 32 MFMAs;
//   LDS: two buffers of A [256][128 B] | B [256][128 B] (128 KB), 16-byte slot XORed with (row >> 1) & 7 on the DMA source side.
//   The MFMAs are inline asm with "+a" accumulators: left to itself hipcc spread the 64 accumulator tiles over both halves of the file
//   and moved 188 registers each way between them per K tile.
// ------------------------------------------------------------------------------------------------------------------
template <int NP3>
__global__ __launch_bounds__(256, 1) void gemm4w_kernel(G8Params p) {
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6), wr = w >> 1, wc = w & 1;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  int tile;
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * G8_BM, n0 = tn * G8_BN;
  const unsigned long long aa = (unsigned long long)p.A, ba = (unsigned long long)p.B;
  const i32x4_g8 rsA = {(int)(unsigned)aa, (int)((unsigned)(aa >> 32) & 0xffffu), (int)p.a_bytes, 0x00020000};
  const i32x4_g8 rsB = {(int)(unsigned)ba, (int)((unsigned)(ba >> 32) & 0xffffu), (int)p.b_bytes, 0x00020000};
  const int dr = lane >> 3, dc = lane & 7;
  const int voff0 = dr * p.K * 2 + ((dc ^ (dr >> 1)) << 4), voff1 = dr * p.K * 2 + ((dc ^ (dr >> 1) ^ 4) << 4);   // even / odd piece (row bit 3)
  const int nkt = p.K / G8_BK;
  auto piece = [&](int t, int n) __attribute__((always_inline)) {     // piece n (0..15) of this wave for K tile t -> buffer t & 1 (32 A + 32 B pieces of 8 rows)
    const unsigned base = lds0 + (unsigned)((t & 1) * 65536);
    const bool ok = t < nkt;
    const int pc = w + 4 * n, isB = pc >= 32, q = isB ? pc - 32 : pc;
    const int so = ((isB ? n0 : m0) + q * 8) * p.K * 2 + t * (G8_BK * 2);
    dma16_g8(isB ? rsB : rsA, base + (unsigned)((isB ? 32768 : 0) + q * 1024), ok ? ((q & 1) ? voff1 : voff0) : OOB_G8, ok ? so : 0);
  };
  const int fl = (li >> 1) & 7;
  const unsigned sk[2] = {(unsigned)((g ^ fl) << 4), (unsigned)((g ^ fl ^ 4) << 4)};
  const unsigned ra = (unsigned)((wr * 128 + li) * 128), rb = (unsigned)(32768 + (wc * 128 + li) * 128);

  f32x4 acc[8][8];
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 af[2][8], bf[2][8];
  // one row of 8 MFMAs (accumulator row i, K step kk), then -- between the rows -- a share of the next fragments / the next DMA pieces
#define G4_ROW(kk, i) _Pragma("unroll") for (int j = 0; j < 8; ++j)                                                  \
      asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(acc[i][j]) : "v"(bf[kk][j]), "v"(af[kk][i]))
#define G4_RDA(kk, buf, i) af[kk][i] = *reinterpret_cast<const bf16x8*>(smem + (buf) * 65536 + ra + (i) * 2048 + sk[kk])
#define G4_RDB(kk, buf, i) bf[kk][i] = *reinterpret_cast<const bf16x8*>(smem + (buf) * 65536 + rb + (i) * 2048 + sk[kk])
  // NP3: pieces of K tile t + 1 issued behind the barrier of tile t - 1 (segment 3); the rest in the next segment 1
#pragma unroll
  for (int n = 0; n < 16; ++n) piece(0, n);
#pragma unroll
  for (int n = 0; n < NP3; ++n) piece(1, n);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NP3) : "memory");         // K tile 0 landed (this wave's pieces)
  G8_BAR();
#pragma unroll
  for (int i = 0; i < 8; ++i) { G4_RDA(0, 0, i); G4_RDB(0, 0, i); }
  for (int t = 0; t < nkt; ++t) {
    const int d = t & 1;
    // seg 1: K step 0 (64 MFMAs) + the fragments of K step 1 + the remaining pieces of K tile t + 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      if (i < 4) { G4_RDA(1, d, 2 * i); G4_RDB(1, d, 2 * i); G4_RDA(1, d, 2 * i + 1); G4_RDB(1, d, 2 * i + 1); }
      if (NP3 + 2 * i < 16) piece(t + 1, NP3 + 2 * i);
      if (NP3 + 2 * i + 1 < 16) piece(t + 1, NP3 + 2 * i + 1);
      __builtin_amdgcn_sched_barrier(0);
      G4_ROW(0, i);
    }
    // seg 2: first half of K step 1
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < 4; ++i) { __builtin_amdgcn_sched_barrier(0); G4_ROW(1, i); }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                 // K tile t + 1 landed; every wave is past its last read of K tile t
    G8_BAR();
    // seg 3: second half of K step 1 + the first fragments of K tile t + 1 + the first pieces of K tile t + 2 (into the buffer just freed)
#pragma unroll
    for (int i = 4; i < 8; ++i) {
      __builtin_amdgcn_sched_barrier(0);
      G4_RDA(0, d ^ 1, 2 * (i - 4)); G4_RDB(0, d ^ 1, 2 * (i - 4)); G4_RDA(0, d ^ 1, 2 * (i - 4) + 1); G4_RDB(0, d ^ 1, 2 * (i - 4) + 1);
#pragma unroll
      for (int n = (i - 4) * NP3 / 4; n < (i - 3) * NP3 / 4; ++n) piece(t + 2, n);
      __builtin_amdgcn_sched_barrier(0);
      G4_ROW(1, i);
    }
  }
  asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");       // the out-of-range tail pieces, the dangling fragment reads
#undef G4_ROW
#undef G4_RDA
#undef G4_RDB
  // ---- epilogue: acc[mi][nj][r] = C[m0 + 128 wr + 16 mi + li][n0 + 128 wc + 16 nj + 4 g + r]
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#pragma unroll
  for (int mi = 0; mi < 8; ++mi) {
    const size_t row = (size_t)(m0 + wr * 128 + mi * 16 + li);
#pragma unroll
    for (int nj = 0; nj < 8; ++nj) {
      const int col = n0 + wc * 128 + nj * 16 + 4 * g;
      u32x2_t o;
      o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[mi][nj][0], acc[mi][nj][1]}, bf16x2_t));
      o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[mi][nj][2], acc[mi][nj][3]}, bf16x2_t));
      *reinterpret_cast<u32x2_t*>(p.C + row * p.N + col) = o;
    }
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same schedule as an implicit-GEMM 3x3 / stride 1 / pad 1 convolution (forward, or the data gradient with the packed
// [Cin][(tap, Cout)] operand and mirrored taps), NHWC bf16, for the 14 x 14 and 7 x 7 stages of the model (256 and 512 channels).
// TILE SHAPE = QUANTISATION: at B = 512 those layers have M = 100 352 / 25 088 output pixels; 256-row tiles give 392 / 196 tiles for
// 256 CUs (one 128 KB workgroup per CU): 1.53 / 0.77 rounds, a quarter of the chip idle.  A tile of 196 output pixels (one 14 x 14
// image, four 7 x 7 images) padded to 224 = 2 x 7 MFMA rows gives 512 / 256 tiles: exactly 2 / 1 rounds, for 12.5 % padded MFMAs.
//   wave (wr, wc): rows 112 wr .. +112 (7 row tiles), columns 64 wc .. +64 (4 column tiles): 28 accumulator tiles
//   A first  = tile rows 0-63 | 112-175 (4 row tiles per M half, 128 buffer rows), A second = rows 64-111 | 176-223 (3 row tiles, 96
//   buffer rows; waves 6-7 stage out-of-range dummies so that every wave counts the same vmcnt);  phases 16 / 16 / 12 / 12 MFMAs.
// A rows are gathered: a lane stages the same 4 tile rows for the whole launch, so their (b, oh, ow) are decoded once and a tap costs a
// compare + add per piece; padding taps and rows beyond the tile's valid rows are out-of-range offsets (zeros from the range check).
// Epilogue: the tile is rounded to bf16 and staged in LDS (528-byte rows), written out as full 512-byte row segments, and the BatchNorm
// column sums (of the bf16 values, like igemm_kernel) are folded over the 16 row groups in a fixed order -> fixed-point accumulators.
// ------------------------------------------------------------------------------------------------------------------
struct C8Params {
  const bf16_t* x; const bf16_t* w; bf16_t* out; unsigned long long* stats;
  const bf16_t* addend; const bf16_t* addmask; const bf16_t* outmask;   // epilogue inputs [M][N] or NULL (as vqa_igemm)
  // BatchNorm-backward column sums of the STORED tile, fused (round 4): the tile is the gradient entering relu(BatchNorm(bn_y)) -- the data
  // gradient da1 of conv2 entering bn1 --: g = out * [bn_y * scale + shift > 0],  sum g | sum g * xhat(bn_y)  -> bn_facc (vqa_bn_acc_words(3, N),
  // the layout vqa_bn_bwd_apply_acc reads); the standalone vqa_bn_bwd_reduce pass over (out, bn_y) is then skipped by the caller
  const bf16_t* bn_y; const float* bn_coef; unsigned long long* bn_facc;
  // bn_selfmask = 0: the tile is the gradient entering a BatchNorm whose ReLU mask the epilogue has already applied (outmask = the block
  // output: the gradient handed to the previous residual block, entering ITS bn2): g = out.  bn_y2 / bn_coef2: that block's 1x1-shortcut
  // BatchNorm sharing g -> the third row, sum g * xhat(bn_y2) (what bn_bwd_reduce_kernel<false, DUAL> leaves)
  int bn_selfmask; const bf16_t* bn_y2; const float* bn_coef2;
  int M, N, K, B, H, W, C, transposed, rpt, tiles_n, ntiles, cpk_shift;
  int dbg;   // VQA_C8P_DBG, -DVQA_ABLATION builds only (WRONG results, timing diagnostics): 1 every A piece of a half-tile re-fetches the first one's source, 2 the same for B, 4 no MFMAs, 8 / 16 every A / B piece moves one lane's 16 bytes only
  int Ho, Wo, stride;                 // output map (= H, W for stride 1); forward convs also run with stride 2 (the stage-entry 3x3 / 2 convs)
  unsigned x_bytes, w_bytes;
};

// Wave layout WM (M) x WN (N), WM * WN = 8, every wave 112 rows x 64 columns:
//   2 x 4: tile 224 x 256 (N a multiple of 256: the 256- and 512-channel stages), 196 valid rows;
//   4 x 2: tile 448 x 128 (N = 128: the 128-channel stage, 28 x 28 maps: 392 valid rows = half an image, 1024 tiles = 4 rounds at B = 512).
// SIMD partners are waves w and w + 4: the stagger groups are w < 4 / w >= 4 in both layouts (2 x 4: wm = w >> 2; 4 x 2: wm = w >> 1).
template <int WM, int WN>
struct C8Geo {
  static constexpr int BMP = WM * 112, BN = WN * 64;
  static constexpr int AF_ROWS = WM * 64, AS_ROWS = WM * 48, B_ROWS = WN * 32;
  static constexpr int N_AF = AF_ROWS / 64, N_AS = (AS_ROWS / 8 + 7) / 8, N_B = (B_ROWS / 8 + 7) / 8;      // DMA pieces per wave and half-tile
  // a half-tile's LDS region holds EVERY piece the 8 waves issue for it, the out-of-range dummies included (an out-of-range LDS-DMA still
  // writes its zeros: 2 x 4 layout, A second = 12 real pieces + 4 dummies of waves 6-7)
  static constexpr int AF_B = N_AF * 8192, AS_B = N_AS * 8192, BH_B = N_B * 8192;
  static constexpr int OFF_AF = 0, OFF_AS = AF_B, OFF_BF = AF_B + AS_B, OFF_BS = AF_B + AS_B + BH_B, BUF = AF_B + AS_B + 2 * BH_B;
  static constexpr int INFLIGHT = N_AF + N_AS + 2 * N_B;                                                   // pieces of the four youngest half-tiles
  static constexpr int LDC = BN * 2 + 16, CPR = BN / 8, RG = 512 / CPR;                                     // epilogue staging
  static constexpr int LDS = 2 * BUF > BMP * LDC ? 2 * BUF : BMP * LDC;
  static_assert(LDS <= 160 * 1024 && RG * 2 * BN * 4 <= LDS && 2 * BN <= 512, "LDS budget");
};

#ifdef VQA_ABLATION
// In-kernel stamps (VQA_C8P_DBG & 32, diagnostic build only; cdna_hip_programming.md section 7): s_memtime at three points of every phase -- MFMA
// segment start (behind the mid barrier and the fragment wait), MFMA segment end (all issued), behind the end barrier -- summed per point over the K
// loop in scalar registers; waves of workgroup 0 leave their 12 sums here.  Read the SHARES, not the length (each stamp costs ~40 cycles and drains LDS).
__device__ unsigned g_c8p_stamps[8 * 20];
#define C8_STAMP(IDX)                                                                                              \
  if (p.dbg & 32) {                                                                                                \
    unsigned long long tt_;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    st_acc[IDX] += (unsigned)tt_ - st_prev; st_prev = (unsigned)tt_;                                               \
  }
#define C8_STAMP2(IDX)                                                                                             \
  if (p.dbg & 64) {                                                                                                \
    unsigned long long tt_;                                                                                        \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(tt_)::"memory");                                   \
    __builtin_amdgcn_sched_barrier(0);                                                                             \
    st_acc[IDX] += (unsigned)tt_ - st_prev; st_prev = (unsigned)tt_;                                               \
  }
#else
#define C8_STAMP(IDX)
#define C8_STAMP2(IDX)
#endif
template <int WM, int WN>
__global__ __launch_bounds__(512, 2) void conv8p_kernel(C8Params p) {
  using G = C8Geo<WM, WN>;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, li = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = WM == 2 ? w >> 2 : w >> 1, wn = WM == 2 ? w & 3 : w & 1, grp = w >> 2;
  const unsigned lds0 = (unsigned)(size_t)((__attribute__((address_space(3))) char*)smem);
  int tile;
  {
    const int nwg = (int)gridDim.x, orig = (int)blockIdx.x, xcd = orig & 7, q = nwg >> 3, r = nwg & 7;
    tile = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (orig >> 3);
  }
  const int tm = tile / p.tiles_n, tn = tile - tm * p.tiles_n;
  const int m0 = tm * p.rpt, n0 = tn * G::BN;
  const unsigned long long xa = (unsigned long long)p.x, wa = (unsigned long long)p.w;
  const i32x4_g8 rsX = {(int)(unsigned)xa, (int)((unsigned)(xa >> 32) & 0xffffu), (int)p.x_bytes, 0x00020000};
  const i32x4_g8 rsW = {(int)(unsigned)wa, (int)((unsigned)(wa >> 32) & 0xffffu), (int)p.w_bytes, 0x00020000};
  const int nkt = p.K / G8_BK;
  const int dr = lane >> 3, dc = lane & 7;
  const int slot_e = (dc ^ (dr >> 1)) << 4, slot_o = (dc ^ (dr >> 1) ^ 4) << 4;       // source slot of an even / odd piece (LDS row bit 3)
  const int HW = p.Ho * p.Wo;
#ifndef VQA_ABLATION
  // ONE M0 write per half-tile: the pieces of a wave's group land at M0 + 1024 h through the instruction offset, which is added to the
  // memory address too -- the descriptors start 4 KB below the tensors and the vector offsets carry + 4096 - 1024 h (never negative)
  const unsigned long long xs = xa - 4096ull, ws_ = wa - 4096ull;
  const i32x4_g8 rsXs = {(int)(unsigned)xs, (int)((unsigned)(xs >> 32) & 0xffffu), (int)(p.x_bytes + 4096u), 0x00020000};
  const i32x4_g8 rsWs = {(int)(unsigned)ws_, (int)((unsigned)(ws_ >> 32) & 0xffffu), (int)(p.w_bytes + 4096u), 0x00020000};
#endif

  // ---- this lane's staged A rows: piece pc = w * N + h of A first (h < N_AF) and A second (h < N_AS); decoded once
  constexpr int NA = G::N_AF + G::N_AS;
  int pixb[NA], ohw[NA], voffA[NA];
#pragma unroll
  for (int k = 0; k < NA; ++k) {
    const bool second = k >= G::N_AF;
    const int pck = second ? w * G::N_AS + (k - G::N_AF) : w * G::N_AF + k;                 // piece
    const int r = pck * 8 + dr, slot = (pck & 1) ? slot_o : slot_e;                         // buffer row
    int R;
    if (!second) R = (r / 64) * 112 + (r % 64);                                             // first 64 rows of each M group
    else R = r < G::AS_ROWS ? (r / 48) * 112 + 64 + (r % 48) : G::BMP;                      // the other 48 (pieces beyond: none)
    const int m = m0 + R;
    voffA[k] = OOB_G8;
    if (R < p.rpt && m < p.M) {
      const int b = m / HW, rem = m - b * HW, oh = (rem / p.Wo) * p.stride, ow = (rem - (rem / p.Wo) * p.Wo) * p.stride;   // input pixel of tap (1, 1)
      pixb[k] = ((b * p.H + oh) * p.W + ow) * p.C * 2 + slot;
#ifndef VQA_ABLATION
      pixb[k] += 4096 - 1024 * (second ? k - G::N_AF : k);
#endif
      ohw[k] = (oh << 16) | ow;
    } else { pixb[k] = 0; ohw[k] = (int)0xc0000000; }                // oh = -16384: every tap fails the range test
  }
  const int voffB0 = dr * p.K * 2 + slot_e, voffB1 = dr * p.K * 2 + slot_o;
  // inr_c: compile-time promise that K tile t exists (the steady loop; the last two K tiles and the prologue test t < nkt and stage zeros beyond)
  auto stage_c = [&](auto inr_c, int which, int t) __attribute__((always_inline)) {      // which: 0 A first, 1 A second, 2 B first, 3 B second
    const unsigned base = lds0 + (unsigned)((t & 1) * G::BUF + (which == 0 ? G::OFF_AF : which == 1 ? G::OFF_AS : which == 2 ? G::OFF_BF : G::OFF_BS));
    const bool okt = decltype(inr_c)::value ? true : t < nkt;
    if (which < 2) {                                                 // A: gathered pixels of tap (t >> cpk_shift), channel chunk t & (cpk - 1)
      const int tap = t >> p.cpk_shift, cc = t & ((1 << p.cpk_shift) - 1);
      const int k0 = which ? G::N_AF : 0, nk = which ? G::N_AS : G::N_AF;
      if (cc == 0) {                                                 // a new tap (scalar branch, every C / 64 K tiles): this half's source offsets
        const int r = (tap * 11) >> 5, s_ = tap - 3 * r;
        const int d_r = p.transposed ? 1 - r : r - 1, d_s = p.transposed ? 1 - s_ : s_ - 1;
        const int dpix = (d_r * p.W + d_s) * p.C * 2;
#pragma unroll
        for (int h = 0; h < nk; ++h) {
          const int k = k0 + h, ih = (ohw[k] >> 16) + d_r, iw = (ohw[k] & 0xffff) + d_s;
          voffA[k] = ((unsigned)ih < (unsigned)p.H && (unsigned)iw < (unsigned)p.W) ? pixb[k] + dpix : OOB_G8;
        }
      }
#ifdef VQA_ABLATION
#pragma unroll
      for (int h = 0; h < nk; ++h)
        if (!(p.dbg & 8) || lane == 0)      // 8: every piece moves ONE lane's 16 bytes (same instruction count and waits, 1/64 of the bytes)
        dma16_g8(rsX, base + (unsigned)((w * nk + h) * 1024), okt ? voffA[(p.dbg & 1) ? k0 : k0 + h] : OOB_G8, okt ? cc * 128 : 0);
#endif
#ifndef VQA_ABLATION
      set_m0_g8(base + (unsigned)(w * nk * 1024));
      const int sof = okt ? cc * 128 : 0;                            // (scalar offset: wave-uniform)
      if (nk > 0) dma16_g8_off<0>(rsXs, okt ? voffA[k0] : OOB_G8, sof);
      if (nk > 1) dma16_g8_off<1024>(rsXs, okt ? voffA[k0 + 1] : OOB_G8, sof);
      if (nk > 2) dma16_g8_off<2048>(rsXs, okt ? voffA[k0 + 2] : OOB_G8, sof);
      if (nk > 3) dma16_g8_off<3072>(rsXs, okt ? voffA[k0 + 3] : OOB_G8, sof);
#endif
    } else {
#pragma unroll
      for (int h = 0; h < G::N_B; ++h) {
        const int pc = w * G::N_B + h, r0 = pc * 8;                  // buffer rows: 32 of the 64 columns of each N group
        const bool okp = okt && r0 < G::B_ROWS;
        const int col = (r0 / 32) * 64 + (r0 % 32) + (which == 3 ? 32 : 0);
#ifdef VQA_ABLATION
        if (!(p.dbg & 16) || lane == 0)     // 16: the same for the B pieces
        dma16_g8(rsW, base + (unsigned)(pc * 1024), okp ? ((pc & 1) ? voffB1 : voffB0) : OOB_G8, okp ? ((n0 + ((p.dbg & 2) ? 0 : col)) * p.K + t * G8_BK) * 2 : 0);
#else
        if (h == 0) set_m0_g8(base + (unsigned)(pc * 1024));
        const int vo = okp ? ((pc & 1) ? voffB1 : voffB0) + 4096 - 1024 * h : OOB_G8, so = okp ? ((n0 + col) * p.K + t * G8_BK) * 2 : 0;
        if (h == 0) dma16_g8_off<0>(rsWs, vo, so);
        else if (h == 1) dma16_g8_off<1024>(rsWs, vo, so);
        else dma16_g8_off<2048>(rsWs, vo, so);
#endif
      }
    }
  };
  auto stage = [&](int which, int t) __attribute__((always_inline)) { stage_c(std::false_type{}, which, t); };

  const int fl = (li >> 1) & 7;
  const unsigned sk[2] = {(unsigned)((g ^ fl) << 4), (unsigned)((g ^ fl ^ 4) << 4)};       // slot of k half kk
  const unsigned ra0 = (unsigned)(G::OFF_AF + (wm * 64 + li) * 128);     // + i * 2048 + sk[kk], i < 4
  const unsigned ra1 = (unsigned)(G::OFF_AS + (wm * 48 + li) * 128);     // i < 3
  const unsigned rb0 = (unsigned)(G::OFF_BF + (wn * 32 + li) * 128);
  const unsigned rb1 = (unsigned)(G::OFF_BS + (wn * 32 + li) * 128);

  f32x4 acc[7][4];
#pragma unroll
  for (int i = 0; i < 7; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
  bf16x8 a0[4][2], a1[3][2], b0[2][2], b1[2][2];

#define C8_RD(dst, off) dst = *reinterpret_cast<const bf16x8*>(smem + d * G::BUF + (off))
#define C8_WAIT() asm volatile("s_waitcnt vmcnt(%0)" ::"n"(G::INFLIGHT) : "memory")
#ifdef VQA_ABLATION
#define C8_COMPUTE(AF, BFR, MI, NJ, NI) if (!(p.dbg & 4)) { G8_COMPUTE(AF, BFR, MI, NJ, NI); }
#else
#define C8_COMPUTE(AF, BFR, MI, NJ, NI) G8_COMPUTE(AF, BFR, MI, NJ, NI)
#endif

  stage(0, 0); stage(2, 0); stage(3, 0); stage(1, 0); stage(0, 1); stage(2, 1);
  C8_WAIT();                                                        // A first (0), B first (0) landed (this wave's pieces)
  G8_BAR();
  if (grp == 1) G8_BAR();                                           // group 1 runs one barrier behind group 0 from here on

  // The A-first fragments of K tile t + 1 are read in P4 of tile t (their registers are free after P2's MFMAs): the fragment reads of
  // the four load segments are 4 / 4 / 6 / 8 instead of 12 / 4 / 6 / 0 (round 4: the 12-read P1 segment was the longest interval).
  // Hence A first (t + 1) is waited for in P3 (t) and read one phase later; it is restaged in P3 (t + 1), three phases after that read.
#define C8_RD_AT(dst, dd, off) dst = *reinterpret_cast<const bf16x8*>(smem + (dd) * G::BUF + (off))
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int kk = 0; kk < 2; ++kk) C8_RD_AT(a0[i][kk], 0, ra0 + i * 2048 + sk[kk]);

#ifdef VQA_ABLATION
  unsigned st_prev = 0, st_acc[20] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  if (p.dbg & 224) { unsigned long long t0_; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0_)::"memory"); st_prev = (unsigned)t0_; }
#endif
  auto ktile = [&](auto inr_c, const int t) __attribute__((always_inline)) {
    const int d = t & 1;
    // ---------------- P1: rows 0-63 of the wave x columns 0-31
#ifdef VQA_ABLATION
    unsigned long long q0_ = 0, q1_ = 0, q2_ = 0, q3_ = 0;           // dbg & 128: raw s_memtime inside P1's load segment (no waits of their own; read behind the segment's own lgkmcnt(0))
    if (p.dbg & 128) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(q0_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) C8_RD(b0[j][kk], rb0 + j * 2048 + sk[kk]);
#ifdef VQA_ABLATION
    if (p.dbg & 128) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(q1_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    stage_c(inr_c, 3, t + 1);
#ifdef VQA_ABLATION
    if (p.dbg & 128) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(q2_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    C8_STAMP2(12);
    C8_WAIT();                                                      // B second (t) landed -> read in P2
    C8_STAMP2(16);
#ifdef VQA_ABLATION
    if (p.dbg & 128) { __builtin_amdgcn_sched_barrier(0); asm volatile("s_memtime %0" : "=s"(q3_)::"memory"); __builtin_amdgcn_sched_barrier(0); }
#endif
    G8_BAR();
#ifdef VQA_ABLATION
    if (p.dbg & 128) {
      unsigned long long q4_;
      asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(q4_), "+s"(q0_), "+s"(q1_), "+s"(q2_), "+s"(q3_)::"memory");
      st_acc[0] += (unsigned)(q1_ - q0_); st_acc[1] += (unsigned)(q2_ - q1_); st_acc[2] += (unsigned)(q3_ - q2_); st_acc[3] += (unsigned)(q4_ - q3_);
    }
#endif
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    C8_STAMP(0);
    C8_COMPUTE(a0, b0, 0, 0, 4);
    C8_STAMP(1);
    G8_BAR();
    C8_STAMP(2);
    // ---------------- P2: rows 0-63 x columns 32-63
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) C8_RD(b1[j][kk], rb1 + j * 2048 + sk[kk]);
    stage_c(inr_c, 1, t + 1);
    C8_STAMP2(13);
    C8_WAIT();                                                      // A second (t) landed -> read in P3
    C8_STAMP2(17);
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    C8_STAMP(3);
    C8_COMPUTE(a0, b1, 0, 2, 4);
    C8_STAMP(4);
    G8_BAR();
    C8_STAMP(5);
    // ---------------- P3: rows 64-111 x columns 32-63
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) C8_RD(a1[i][kk], ra1 + i * 2048 + sk[kk]);
    stage_c(inr_c, 0, t + 2);
    C8_STAMP2(14);
    C8_WAIT();                                                      // A first (t + 1) landed -> read in P4
    C8_STAMP2(18);
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    C8_STAMP(6);
    C8_COMPUTE(a1, b1, 4, 2, 3);
    C8_STAMP(7);
    G8_BAR();
    C8_STAMP(8);
    // ---------------- P4: rows 64-111 x columns 0-31; the A-first fragments of the NEXT K tile
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int kk = 0; kk < 2; ++kk) C8_RD_AT(a0[i][kk], d ^ 1, ra0 + i * 2048 + sk[kk]);
    stage_c(inr_c, 2, t + 2);
    C8_STAMP2(15);
    C8_WAIT();                                                      // B first (t + 1) landed -> read in P1 of the next K tile
    C8_STAMP2(19);
    G8_BAR();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    C8_STAMP(9);
    C8_COMPUTE(a1, b0, 4, 0, 3);
    C8_STAMP(10);
    G8_BAR();
    C8_STAMP(11);
  };
  // the steady loop stages K tiles that exist (no range test, no select per piece); the last two K tiles stage the out-of-range tail
  int t_ = 0;
  for (; t_ + 2 < nkt; ++t_) ktile(std::true_type{}, t_);
  for (; t_ < nkt; ++t_) ktile(std::false_type{}, t_);
#undef C8_RD_AT
#ifdef VQA_ABLATION
  if ((p.dbg & 224) && blockIdx.x == 0 && lane == 0) {
#pragma unroll
    for (int i = 0; i < 20; ++i) g_c8p_stamps[w * 20 + i] = st_acc[i];
  }
#endif
  if (grp == 0) G8_BAR();
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // the out-of-range tail pieces have written their zeros
  G8_BAR();
#undef C8_RD
#undef C8_WAIT

  // ---- epilogue inputs (data gradients: identity-path gradient, ReLU masks, the BatchNorm input whose backward sums leave this launch).
  // A thread owns column chunk c16 of rows rg, rg + RG, ... (<= 14 rows).  Its loads are issued NR rows at a time into two register
  // sets: batch 0 BEFORE the tile is staged (in flight across the two barriers of epilogue 1), batch b + 1 before batch b is consumed,
  // and always ahead of the stores of the batch before -- a wait for loads then never includes a store's acknowledgement.  (First
  // version: 2 rows per step, loads of a step issued after the stores of the step before: 6-7 exposed round trips per tile, +17 us per
  // tile and launch, i.e. +70-84 us on the four-tiles-per-CU stage-2 launches; profiles/r04_conv8p_epilogue.txt.)
  const int c16 = tid % G::CPR, rg = tid / G::CPR;
  const int rows = min(p.rpt, p.M - m0);
  const bool bnred = p.bn_y != nullptr;
  const bool fused = p.addend != nullptr || p.outmask != nullptr || p.bn_y != nullptr;
  const bool dual = bnred && p.bn_y2 != nullptr;
  constexpr int NRMAX = 4, TR = G::BMP / G::RG;                    // rows per thread (14)
  struct EpiRow { Vec16<bf16_t> a, m, o, y, y2; };
  EpiRow inA[NRMAX], inB[NRMAX];
  // Straight-line code on purpose: stream presence is a template mask (bit 0 addend, 1 addmask, 2 outmask, 3 bn_y, 4 bn_y2), rows past
  // the tile's last valid row re-read that row, and the batch loop is fully unrolled.  With a load under a branch, or a register set
  // reloaded around a loop back-edge, hipcc guards each load with s_waitcnt vmcnt(0) (write-after-write on a register that MAY still
  // be a pending load's destination): every load then waits for the one before it.
  const bf16_t* const pa = p.addend ? p.addend : p.w;
  const bf16_t* const pm = p.addmask ? p.addmask : p.w;
  const bf16_t* const po = p.outmask ? p.outmask : p.w;
  const bf16_t* const py = p.bn_y ? p.bn_y : p.w;
  const bf16_t* const py2 = p.bn_y2 ? p.bn_y2 : p.w;
  // (mask_c: std::integral_constant<int, MASK | NR << 8>: NR rows per batch -- 4 with one stream, 3 with two, 2 with all five: registers)
  auto issue = [&](auto mask_c, int b, EpiRow (&e)[NRMAX]) __attribute__((always_inline)) {
    constexpr int MASK = decltype(mask_c)::value & 255, NR = decltype(mask_c)::value >> 8;
#pragma unroll
    for (int i = 0; i < NR; ++i) {
      int R = rg + (b * NR + i) * G::RG;
      R = R < rows ? R : rows - 1;
      const size_t off = (size_t)(m0 + R) * p.N + n0 + c16 * 8;
      // (the full set also serves launches WITHOUT some of the streams: those pointers are the weights, read at offset 0)
      if (MASK & 1) e[i].a = ldg16(pa + (p.addend ? off : 0));
      if (MASK & 2) e[i].m = ldg16(pm + (p.addmask ? off : 0));
      if (MASK & 4) e[i].o = ldg16(po + (p.outmask ? off : 0));
      if (MASK & 8) e[i].y = ldg16(py + (p.bn_y ? off : 0));
      if (MASK & 16) e[i].y2 = ldg16(py2 + (p.bn_y2 ? off : 0));
    }
  };
  // the three stream sets the engine uses get their own code; anything else takes the full set with absent streams pointed at the weights
  const int smask = (p.addend ? 1 : 0) | (p.addmask ? 2 : 0) | (p.outmask ? 4 : 0) | (bnred ? 8 : 0) | (dual ? 16 : 0);
  const int emode = !fused ? 0 : (smask == 8 ? 1 : (smask == 5 ? 2 : (smask == 1 ? 3 : 4)));
  const bool ha = smask & 1, hm = smask & 2, ho = smask & 4;
  using EM1 = std::integral_constant<int, 8 | 4 << 8>;  using EM2 = std::integral_constant<int, 5 | 3 << 8>;
  using EM3 = std::integral_constant<int, 1 | 4 << 8>;  using EM4 = std::integral_constant<int, 31 | 2 << 8>;
  if (emode == 1) issue(EM1{}, 0, inA);
  else if (emode == 2) issue(EM2{}, 0, inA);
  else if (emode == 3) issue(EM3{}, 0, inA);
  else if (emode == 4) issue(EM4{}, 0, inA);

  // ---- epilogue 1: bf16 tile -> LDS  (acc[mi][nj][r] = out[row 112 wm + 16 mi + li][column 64 wn + 16 nj + 4 g + r])
  typedef __attribute__((ext_vector_type(2))) float f32x2_t;
  typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_t;
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#pragma unroll
  for (int mi = 0; mi < 7; ++mi)
#pragma unroll
    for (int nj = 0; nj < 4; ++nj) {
      u32x2_t o;
      o[0] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[mi][nj][0], acc[mi][nj][1]}, bf16x2_t));
      o[1] = __builtin_bit_cast(unsigned, __builtin_convertvector((f32x2_t){acc[mi][nj][2], acc[mi][nj][3]}, bf16x2_t));
      *reinterpret_cast<u32x2_t*>(smem + (wm * 112 + mi * 16 + li) * G::LDC + (wn * 64 + nj * 16 + 4 * g) * 2) = o;
    }
  // ---- epilogue 2: full row segments out; column sums of the stored (bf16) values
  float cs[8], cq[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) { cs[j] = 0.f; cq[j] = 0.f; }
  float bsc[8], bsh[8], bmu[8], biv[8], bmu2[8], biv2[8], cr[8];
  if (bnred) {                                                      // (the accumulators are dead: their registers take the coefficients, under the barrier)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = n0 + c16 * 8 + j;
      bsc[j] = p.bn_coef[c]; bsh[j] = p.bn_coef[p.N + c]; bmu[j] = p.bn_coef[2 * p.N + c]; biv[j] = p.bn_coef[3 * p.N + c];
      cr[j] = 0.f; bmu2[j] = dual ? p.bn_coef2[2 * p.N + c] : 0.f; biv2[j] = dual ? p.bn_coef2[3 * p.N + c] : 0.f;
    }
  }
  asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");               // the staged tile is visible after the barrier; NOT __syncthreads():
  G8_BAR();                                                         // its vmcnt(0) would drain the loads issued above
  if (!fused) {
    for (int R = rg; R < rows; R += G::RG) {
      Vec16<bf16_t> v; v.raw = *reinterpret_cast<const u32x4*>(smem + R * G::LDC + c16 * 16);
      *reinterpret_cast<u32x4*>(p.out + (size_t)(m0 + R) * p.N + n0 + c16 * 8) = v.raw;
      if (p.stats) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float f = v.get(j); cs[j] += f; cq[j] += f * f; }
      }
    }
  } else {
    // out = (conv + addend * (addmask > 0)) * (outmask > 0) on the staged bf16 value, like igemm_kernel's epilogue (the identity-path
    // gradient and the ReLU masks of the data gradients, engine._block_bwd)
    auto consume = [&](auto mask_c, int b, EpiRow (&e)[NRMAX]) __attribute__((always_inline)) {
      constexpr int NR = decltype(mask_c)::value >> 8;
#pragma unroll
      for (int i = 0; i < NR; ++i) {
        const int R = rg + (b * NR + i) * G::RG;
        if (R >= rows) continue;
        const size_t off = (size_t)(m0 + R) * p.N + n0 + c16 * 8;
        Vec16<bf16_t> v; v.raw = *reinterpret_cast<const u32x4*>(smem + R * G::LDC + c16 * 16);
        if (ha) {
          if (hm) {
#pragma unroll
            for (int j = 0; j < 8; ++j) v.set(j, v.get(j) + (e[i].m.get(j) > 0.f ? e[i].a.get(j) : 0.f));
          } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v.set(j, v.get(j) + e[i].a.get(j));
          }
        }
        if (ho) {
#pragma unroll
          for (int j = 0; j < 8; ++j) if (!(e[i].o.get(j) > 0.f)) v.set(j, 0.f);
        }
        stg16(p.out + off, v);
        if (bnred) {                                                // (same arithmetic per element as bn_bwd_reduce_kernel: the stored value, the recomputed ReLU mask)
#pragma unroll
          for (int j = 0; j < 8; ++j) {
            const float yj = e[i].y.get(j);
            float gj = v.get(j);
            if (p.bn_selfmask && !(yj * bsc[j] + bsh[j] > 0.f)) gj = 0.f;
            cs[j] += gj; cq[j] += gj * (yj - bmu[j]) * biv[j];
            if (dual) cr[j] += gj * (e[i].y2.get(j) - bmu2[j]) * biv2[j];
          }
        } else if (p.stats) {
#pragma unroll
          for (int j = 0; j < 8; ++j) { const float f = v.get(j); cs[j] += f; cq[j] += f * f; }
        }
      }
    };
    auto run = [&](auto mask_c) __attribute__((always_inline)) {
      constexpr int NR = decltype(mask_c)::value >> 8, NB = (TR + NR - 1) / NR;
#pragma unroll
      for (int b = 0; b < NB; b += 2) {
        if (b + 1 < NB) issue(mask_c, b + 1, inB);
        consume(mask_c, b, inA);
        if (b + 2 < NB) issue(mask_c, b + 2, inA);
        if (b + 1 < NB) consume(mask_c, b + 1, inB);
      }
    };
    if (emode == 1) run(EM1{});
    else if (emode == 2) run(EM2{});
    else if (emode == 3) run(EM3{});
    else run(EM4{});
  }
  if (p.stats || bnred) {
    __syncthreads();                                                // the staged tile has been read: its LDS is reused for the partial sums
    float* part = reinterpret_cast<float*>(smem);                   // [RG row groups][3][BN]
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      part[(rg * 3 + 0) * G::BN + c16 * 8 + j] = cs[j]; part[(rg * 3 + 1) * G::BN + c16 * 8 + j] = cq[j];
      if (dual) part[(rg * 3 + 2) * G::BN + c16 * 8 + j] = cr[j];
    }
    __syncthreads();
    for (int o = tid; o < (dual ? 3 : 2) * G::BN; o += 512) {
      const int k = o / G::BN, c = o - k * G::BN;
      float t = 0.f;
#pragma unroll
      for (int q = 0; q < G::RG; ++q) t += part[(q * 3 + k) * G::BN + c];
      const int Rr = acc_replicas(p.N);
      if (bnred) acc_add_fixed(p.bn_facc, (size_t)Rr * 3 * p.N, ((size_t)(tm % Rr) * 3 + k) * p.N + n0 + c, t);
      else acc_add_fixed(p.stats, (size_t)Rr * 2 * p.N, (size_t)(tm % Rr) * 2 * p.N + (size_t)k * p.N + n0 + c, t);
    }
  }
}

extern "C" {
// C[M][N] = A[M][K] . B[N][K]^T, bf16 in / fp32 accumulate / bf16 out.  M % 256 == 0, N % 256 == 0, K % 64 == 0.
// the same product on the four-wave / one-wave-per-SIMD kernel (gemm4w_kernel)
int vqa_gemm4w(const void* A, const void* B, void* C, int M, int N, int K, hipStream_t st) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || M % G8_BM || N % G8_BN || K % G8_BK) return VQA_EARG;
  const size_t ab = (size_t)M * K * 2, bb = (size_t)N * K * 2;
  if (ab >= 0x7fffffffull || bb >= 0x7fffffffull) return VQA_EARG;
  G8Params p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C; p.M = M; p.N = N; p.K = K;
  p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  p.tiles_n = N / G8_BN; p.ntiles = (M / G8_BM) * p.tiles_n;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<8>), hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
#ifdef VQA_ABLATION
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<12>), hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm4w_kernel<16>), hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS);
#endif
    attr = true;
  }
#ifdef VQA_ABLATION
  const int np3 = vqa_env_int("VQA_G4_NP3", 8);
  if (np3 == 4) { hipLaunchKernelGGL(gemm4w_kernel<4>, dim3(p.ntiles), dim3(256), G8_LDS, st, p); VQA_LAUNCH_CHECK(); return VQA_OK; }
  if (np3 == 12) { hipLaunchKernelGGL(gemm4w_kernel<12>, dim3(p.ntiles), dim3(256), G8_LDS, st, p); VQA_LAUNCH_CHECK(); return VQA_OK; }
  if (np3 == 16) { hipLaunchKernelGGL(gemm4w_kernel<16>, dim3(p.ntiles), dim3(256), G8_LDS, st, p); VQA_LAUNCH_CHECK(); return VQA_OK; }
#endif
  hipLaunchKernelGGL(gemm4w_kernel<8>, dim3(p.ntiles), dim3(256), G8_LDS, st, p);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
int vqa_gemm8p(const void* A, const void* B, void* C, int M, int N, int K, hipStream_t st) {
  if (!A || !B || !C || M <= 0 || N <= 0 || K <= 0 || M % G8_BM || N % G8_BN || K % G8_BK) return VQA_EARG;
  const size_t ab = (size_t)M * K * 2, bb = (size_t)N * K * 2;
  if (ab >= 0x7fffffffull || bb >= 0x7fffffffull) return VQA_EARG;
  G8Params p;
  p.A = (const bf16_t*)A; p.B = (const bf16_t*)B; p.C = (bf16_t*)C; p.M = M; p.N = N; p.K = K;
  p.a_bytes = (unsigned)ab; p.b_bytes = (unsigned)bb;
  p.tiles_n = N / G8_BN; p.ntiles = (M / G8_BM) * p.tiles_n;
  static bool attr = false;
  if (!attr) { (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&gemm8p_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, G8_LDS); attr = true; }
  hipLaunchKernelGGL(gemm8p_kernel, dim3(p.ntiles), dim3(512), G8_LDS, st, p);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
#ifdef VQA_ABLATION
// diagnostic builds only: the stamp sums of the last vqa_conv8p launch with VQA_C8P_DBG & 32 (8 waves x 12 points, cycles summed over the K loop)
int vqa_conv8p_stamps(unsigned* out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(g_c8p_stamps), sizeof(unsigned) * 160); }
#endif
// 1: vqa_conv8p takes the shape (3x3 / stride 1 / pad 1, NHWC bf16, C a power-of-two multiple of 64, N a multiple of 128)
int vqa_conv8p_ok(int B, int H, int W, int C, int N) {      // (H, W: the INPUT map)
  if (B <= 0 || H <= 0 || W <= 0 || H > 16383 || W > 16383 || C < 64 || (C & (C - 1)) || N <= 0 || N % 128) return 0;
  const size_t xb = (size_t)B * H * W * C * 2, wb = (size_t)N * 9 * C * 2, ob = (size_t)B * H * W * N * 2;
  if (xb >= 0x7fffffffull || wb >= 0x7fffffffull || (size_t)B * H * W >= 0x7fffffffull / 2 || ob >= 0xffffffffull * 2) return 0;
  return 1;
}
// out[B*H*W][N] = conv3x3(x [B][H][W][C], w [N][(r, s, c)]) (transposed = 0), or the stride-1 data gradient (transposed = 1: x is dy
// [B][H][W][Cout], w the packed [Cin][(tap, Cout)] operand of vqa_pack_transpose, taps mirrored).  stats: fixed-point BatchNorm
// accumulator (vqa_bn_acc_words(2, N), caller-zeroed) receiving sum y | sum y^2 of the stored values, or NULL.
// addend / addmask / outmask [B*H*W][N] bf16 or NULL: out = (conv + addend * (addmask > 0)) * (outmask > 0), the epilogue of vqa_igemm.
// bn_y / bn_coef / bn_facc (all or none, not with stats): the BatchNorm-backward column sums of the stored tile (C8Params); bn_selfmask 1: the
// ReLU behind that BatchNorm is recomputed from bn_y, 0: the tile is already masked; bn_y2 / bn_coef2: a second BatchNorm sharing the gradient.
// stride: 1, or 2 (forward only: the 3x3 / 2 / pad 1 stage-entry convs; out is [B][(H+1)/2... ][N], i.e. Ho = (H - 1) / 2 + 1).
// Tile: 224 x 256 (2 x 4 waves) when 256 divides N, else 448 x 128 (4 x 2 waves); 7/8 of the rows valid (196 / 392) when that divides
// B*H*W -- the 14 x 14, 7 x 7 and 28 x 28 maps of the model: exactly 2, 1 and 4 rounds of 256 CUs at B = 512.
int vqa_conv8p(const void* x, const void* w, void* out, unsigned long long* stats, const void* addend, const void* addmask, const void* outmask,
               const void* bn_y, const float* bn_coef, unsigned long long* bn_facc, int bn_selfmask, const void* bn_y2, const float* bn_coef2,
               int B, int H, int W, int C, int N, int transposed, int stride, hipStream_t st) {
  if (!x || !w || !out || !vqa_conv8p_ok(B, H, W, C, N) || (stride != 1 && stride != 2) || (stride == 2 && transposed)) return VQA_EARG;
  C8Params p;
  if (addmask && !addend) return VQA_EARG;
  if ((bn_y != nullptr) != (bn_coef != nullptr) || (bn_y != nullptr) != (bn_facc != nullptr) || (bn_y && stats)) return VQA_EARG;
  if ((bn_y2 != nullptr) != (bn_coef2 != nullptr) || (bn_y2 && !bn_y) || (bn_y2 && bn_selfmask)) return VQA_EARG;
  p.bn_y = (const bf16_t*)bn_y; p.bn_coef = bn_coef; p.bn_facc = bn_facc; p.bn_selfmask = bn_selfmask; p.bn_y2 = (const bf16_t*)bn_y2; p.bn_coef2 = bn_coef2;
  p.x = (const bf16_t*)x; p.w = (const bf16_t*)w; p.out = (bf16_t*)out; p.stats = stats;
  p.addend = (const bf16_t*)addend; p.addmask = (const bf16_t*)addmask; p.outmask = (const bf16_t*)outmask;
  p.stride = stride; p.Ho = (H + 2 - 3) / stride + 1; p.Wo = (W + 2 - 3) / stride + 1;
  p.M = B * p.Ho * p.Wo; p.N = N; p.K = 9 * C; p.B = B; p.H = H; p.W = W; p.C = C; p.transposed = transposed; p.dbg = vqa_env_int("VQA_C8P_DBG", 0);
  p.x_bytes = (unsigned)((size_t)B * H * W * C * 2); p.w_bytes = (unsigned)((size_t)N * 9 * C * 2);
  const bool wide = N % 256 == 0;
  const int bmp = wide ? 224 : 448;
  p.rpt = (p.M % (bmp / 8 * 7) == 0) ? bmp / 8 * 7 : bmp;
  p.tiles_n = N / (wide ? 256 : 128);
  const int tiles_m = (p.M + p.rpt - 1) / p.rpt;
  if ((stats || bn_facc) && tiles_m > VQA_ACC_MAX_PARTS) return VQA_EARG;
  p.ntiles = tiles_m * p.tiles_n;
  p.cpk_shift = 0;
  while ((64 << p.cpk_shift) < C) ++p.cpk_shift;
  static bool attr = false;
  if (!attr) {
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8p_kernel<2, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C8Geo<2, 4>::LDS);
    (void)hipFuncSetAttribute(reinterpret_cast<const void*>(&conv8p_kernel<4, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C8Geo<4, 2>::LDS);
    attr = true;
  }
  constexpr int lds_w = C8Geo<2, 4>::LDS, lds_n = C8Geo<4, 2>::LDS;
  if (wide) hipLaunchKernelGGL((conv8p_kernel<2, 4>), dim3(p.ntiles), dim3(512), lds_w, st, p);
  else hipLaunchKernelGGL((conv8p_kernel<4, 2>), dim3(p.ntiles), dim3(512), lds_n, st, p);
  VQA_LAUNCH_CHECK(); return VQA_OK;
}
}
