// Input pipeline on the GPU: the two steps either side of the host decode.
//   vqa_image_normalize : uint8 HWC image batch -> float32 NCHW, ToTensor + Normalize (data/preprocess.py:34-35,117-121:
//                         x/255, then (x - mean[c]) / std[c]) with an optional per-sample horizontal flip
//                         (RandomHorizontalFlip of data/preprocess.py:73).  The PIL Resize / ColorJitter in front of it stay on the host.
//   vqa_pack_tokens     : ragged word-index lists -> padded token ids + attention mask with the START / END / truncation /
//                         padding conventions of Tokenizer.encode (utils/tokenizer.py:196-250); the string work (lower-casing,
//                         regex split, dictionary lookup) stays on the host.
// Both are pure byte / index movers: HBM-bound, coalesced 16-byte stores.
#include "common.h"

// One thread = 4 consecutive pixels of one image row: 12 input bytes (three aligned dwords) -> one float4 per channel plane.
// Same operation order as torch (ToTensor: float(u8) / 255; Normalize: (t - mean) / std, IEEE division), so the result is
// bit-identical to the reference transform.
__global__ __launch_bounds__(256) void image_normalize_kernel(const uint8_t* __restrict__ in, float* __restrict__ out,
                                                              const uint8_t* __restrict__ flip, int B, int H, int W,
                                                              float m0, float m1, float m2, float s0, float s1, float s2) {
  const int W4 = W >> 2;
  const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t total = (size_t)B * H * W4;
  if (idx >= total) return;
  const int w4 = (int)(idx % W4);
  const size_t bh = idx / W4;
  const int h = (int)(bh % H), b = (int)(bh / H);
  const uint32_t* src = reinterpret_cast<const uint32_t*>(in + ((size_t)bh * W + (size_t)w4 * 4) * 3);
  const uint32_t d0 = src[0], d1 = src[1], d2 = src[2];
  uint8_t px[12];
#pragma unroll
  for (int i = 0; i < 4; ++i) { px[i] = (d0 >> (8 * i)) & 0xff; px[4 + i] = (d1 >> (8 * i)) & 0xff; px[8 + i] = (d2 >> (8 * i)) & 0xff; }
  const bool fl = flip && flip[b];
  const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
  const size_t plane = (size_t)H * W;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    f32x4 v;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float t = (float)px[(fl ? 3 - i : i) * 3 + c] / 255.0f;
      v[i] = (t - mean[c]) / sd[c];
    }
    const int wo = fl ? (W - 4 - w4 * 4) : w4 * 4;
    *reinterpret_cast<f32x4*>(out + ((size_t)b * 3 + c) * plane + (size_t)h * W + wo) = v;
  }
}

// One thread per output position (b, l).  words [offsets[b], offsets[b+1]) are the question's vocabulary indices (unknown
// words already mapped to UNK by the host lookup).
__global__ __launch_bounds__(256) void pack_tokens_kernel(const int* __restrict__ words, const long long* __restrict__ offsets,
                                                          long long* __restrict__ ids, long long* __restrict__ mask, int B, int L,
                                                          int add_special, int start_idx, int end_idx, int pad_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= B * L) return;
  const int b = i / L, l = i - b * L;
  const long long o0 = offsets[b];
  const int nw = (int)(offsets[b + 1] - o0);
  int n = nw + (add_special ? 2 : 0);                   // tokens before truncation
  const bool trunc = n > L;
  if (trunc) n = L;
  long long id = pad_idx, m = 0;
  if (l < n) {
    m = 1;
    if (add_special) {
      if (l == 0) id = start_idx;
      else if (l == n - 1) id = end_idx;                // the END token, also forced onto the last slot after truncation
      else id = words[o0 + l - 1];
    } else {
      id = words[o0 + l];
    }
  }
  ids[i] = id; mask[i] = m;
}

extern "C" {

int vqa_image_normalize(const uint8_t* in_hwc, float* out_nchw, const uint8_t* flip, int B, int H, int W,
                        float mean0, float mean1, float mean2, float std0, float std1, float std2, hipStream_t st) {
  if (!in_hwc || !out_nchw || B <= 0 || H <= 0 || W <= 0 || (W % 4)) return VQA_EARG;
  if (!(std0 != 0.f && std1 != 0.f && std2 != 0.f)) return VQA_EARG;
  const size_t total = (size_t)B * H * (W / 4);
  hipLaunchKernelGGL(image_normalize_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, in_hwc, out_nchw, flip, B, H, W,
                     mean0, mean1, mean2, std0, std1, std2);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

int vqa_pack_tokens(const int* words, const long long* offsets, long long* ids, long long* mask, int B, int L, int add_special,
                    int start_idx, int end_idx, int pad_idx, hipStream_t st) {
  if (!offsets || !ids || !mask || B <= 0 || L <= 0 || (add_special && L < 2)) return VQA_EARG;
  hipLaunchKernelGGL(pack_tokens_kernel, dim3((unsigned)((B * L + 255) / 256)), dim3(256), 0, st, words, offsets, ids, mask, B, L,
                     add_special, start_idx, end_idx, pad_idx);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

}  // extern "C"
