// Shared by cnn_ops.hip and stem_conv.hip.
#pragma once
#include "common.h"

// gradient wrt the (virtual) 112x112 post-ReLU activation, routed through the pool argmax and the ReLU mask
template <typename T>
__device__ __forceinline__ void stem_route(const T* __restrict__ dpool, const uint8_t* __restrict__ idx, const Vec16<T>& yy,
                                           const float* sc, const float* sh, int b, int h, int w, int c0, int C, int Ho, int Wo, float* g) {   // sc/sh: this thread's 8 (or 4) scale / shift values, register resident
  constexpr int VEC = Vec16<T>::N;
#pragma unroll
  for (int j = 0; j < VEC; ++j) g[j] = 0.f;
  // candidate windows: oh in {(h-1)>>1, (h+1)>>1} (equal for even h), same for w; loads are unconditional (clamped)
  const int oh_a = (h - 1) >> 1, oh_b = (h + 1) >> 1, ow_a = (w - 1) >> 1, ow_b = (w + 1) >> 1;
  Vec16<T> d[4]; uint64_t iw4[4]; bool ok[4]; int code[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oh = (k >> 1) ? oh_b : oh_a, ow = (k & 1) ? ow_b : ow_a;
    const int r = h - (oh * 2 - 1), s = w - (ow * 2 - 1);
    ok[k] = oh >= 0 && oh < Ho && ow >= 0 && ow < Wo && r >= 0 && r <= 2 && s >= 0 && s <= 2 &&
            !((k >> 1) && oh_b == oh_a) && !((k & 1) && ow_b == ow_a);          // do not count a window twice
    code[k] = r * 3 + s;
    const int ohc = min(max(oh, 0), Ho - 1), owc = min(max(ow, 0), Wo - 1);
    const size_t o = (((size_t)b * Ho + ohc) * Wo + owc) * C + c0;
    d[k] = ldg16(dpool + o);
    if constexpr (VEC == 8) iw4[k] = *reinterpret_cast<const uint64_t*>(idx + o);
    else iw4[k] = *reinterpret_cast<const uint32_t*>(idx + o);
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < VEC; ++j) {
      const int id = (int)((iw4[k] >> (8 * j)) & 0xff);
      if (ok[k] && id == code[k]) g[j] += d[k].get(j);
    }
#pragma unroll
  for (int j = 0; j < VEC; ++j)
    if (!(yy.get(j) * sc[j] + sh[j] > 0.f)) g[j] = 0.f;
}

// stem_route with 32-bit buffer offsets (raw buffer loads: no 64-bit address arithmetic per window, the range check replaces nothing --
// the offsets are always clamped in range).  rsP / rsI: buffer resources of dpool (bf16) and idx (bytes).  Same sums, same order.
__device__ __forceinline__ void stem_route_buf(__amdgpu_buffer_rsrc_t rsP, __amdgpu_buffer_rsrc_t rsI, const Vec16<bf16_t>& yy,
                                               const float* sc, const float* sh, int b, int h, int w, int c0, int Ho, int Wo, float* g) {
#pragma unroll
  for (int j = 0; j < 8; ++j) g[j] = 0.f;
  const int oh_a = (h - 1) >> 1, oh_b = (h + 1) >> 1, ow_a = (w - 1) >> 1, ow_b = (w + 1) >> 1;
  Vec16<bf16_t> d[4]; uint32_t ilo[4], ihi[4]; int code[4];
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int oh = (k >> 1) ? oh_b : oh_a, ow = (k & 1) ? ow_b : ow_a;
    const int r = h - (oh * 2 - 1), s = w - (ow * 2 - 1);
    const bool ok = oh >= 0 && oh < Ho && ow >= 0 && ow < Wo && r >= 0 && r <= 2 && s >= 0 && s <= 2 &&
                    !((k >> 1) && oh_b == oh_a) && !((k & 1) && ow_b == ow_a);          // do not count a window twice
    code[k] = ok ? r * 3 + s : 255;                           // 255 never matches an argmax code (0..8)
    const int ohc = min(max(oh, 0), Ho - 1), owc = min(max(ow, 0), Wo - 1);
    const int o = ((b * Ho + ohc) * Wo + owc) * 64 + c0;
    d[k].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, o * 2, 0, 0));
    typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
    const u32x2_t iw = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsI, o, 0, 0));
    ilo[k] = iw[0]; ihi[k] = iw[1];
  }
#pragma unroll
  for (int k = 0; k < 4; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int id = (int)(((j < 4 ? ilo[k] : ihi[k]) >> (8 * (j & 3))) & 0xff);
      if (id == code[k]) g[j] += d[k].get(j);
    }
#pragma unroll
  for (int j = 0; j < 8; ++j)
    if (!(yy.get(j) * sc[j] + sh[j] > 0.f)) g[j] = 0.f;
}

// The same routing for TWO horizontally adjacent pixels (h, 2j) and (h, 2j+1) at once: column 2j lies in pooling-window column j only
// (s = 1), column 2j+1 in window columns j (s = 2) and j+1 (s = 0); window rows: odd h lies in rows (h-1)/2 (r = 2) and (h+1)/2 (r = 0),
// even h only in row h/2 (r = 1).  The row parity is a TEMPLATE parameter: the caller branches once per image row (wave-uniform) into
// a body with two window-row passes or with one, each with all its loads in one batch.  (A run-time `continue` around the needless
// pass of even rows inside ONE body split the batched loads into dependent groups: 0.69 -> 0.95 ms for the fused stem weight
// gradient on MI355X.)  2 or 4 window loads and unpacks for 2 pixels instead of 4 per pixel -- the staging of the fused stem weight
// gradient is VALU-bound.  Sums are formed in the same order as stem_route (window row a before b, column j before j+1), so the
// values are bit-identical.
template <bool ODD>
__device__ __forceinline__ void stem_route_pair_buf(__amdgpu_buffer_rsrc_t rsP, __amdgpu_buffer_rsrc_t rsI, const Vec16<bf16_t> (&yy)[2],
                                                    const float* sc, const float* sh, int b, int h, int j, int c0, int Ho, int Wo,
                                                    float (&g)[2][8]) {
  typedef __attribute__((ext_vector_type(2))) unsigned u32x2_t;
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) g[q][jx] = 0.f;
  constexpr int NP = ODD ? 2 : 1;
#pragma unroll
  for (int kr = 0; kr < NP; ++kr) {
    const int oh = ODD ? ((h - 1) >> 1) + kr : (h >> 1);
    const int r = ODD ? (kr ? 0 : 2) : 1;                     // = h - (2 * oh - 1)
    const bool rok = oh >= 0 && oh < Ho;
    const int ohc = min(max(oh, 0), Ho - 1);
    Vec16<bf16_t> d[2]; u32x2_t iw[2]; bool ok[2];
#pragma unroll
    for (int kc = 0; kc < 2; ++kc) {
      const int ow = j + kc;
      ok[kc] = rok && ow < Wo;
      const int o = ((b * Ho + ohc) * Wo + min(ow, Wo - 1)) * 64 + c0;
      d[kc].raw = __builtin_bit_cast(u32x4, __builtin_amdgcn_raw_buffer_load_b128(rsP, o * 2, 0, 0));
      iw[kc] = __builtin_bit_cast(u32x2_t, __builtin_amdgcn_raw_buffer_load_b64(rsI, o, 0, 0));
    }
    const int c_e = ok[0] ? r * 3 + 1 : 255, c_o0 = ok[0] ? r * 3 + 2 : 255, c_o1 = ok[1] ? r * 3 : 255;
#pragma unroll
    for (int jx = 0; jx < 8; ++jx) {
      const int i0 = (int)((iw[0][jx >> 2] >> (8 * (jx & 3))) & 0xff), i1 = (int)((iw[1][jx >> 2] >> (8 * (jx & 3))) & 0xff);
      const float d0 = d[0].get(jx), d1 = d[1].get(jx);
      if (i0 == c_e) g[0][jx] += d0;
      if (i0 == c_o0) g[1][jx] += d0;
      if (i1 == c_o1) g[1][jx] += d1;
    }
  }
#pragma unroll
  for (int q = 0; q < 2; ++q)
#pragma unroll
    for (int jx = 0; jx < 8; ++jx)
      if (!(yy[q].get(jx) * sc[jx] + sh[jx] > 0.f)) g[q][jx] = 0.f;
}
