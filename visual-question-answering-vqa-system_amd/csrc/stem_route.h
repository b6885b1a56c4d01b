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

