// Input pipeline on the GPU: the two steps either side of the host decode.
//   vqa_image_normalize : uint8 HWC image batch -> float32 NCHW, ToTensor + Normalize (data/preprocess.py:34-35,117-121:
//                         x/255, then (x - mean[c]) / std[c]) with an optional per-sample horizontal flip
//                         (RandomHorizontalFlip of data/preprocess.py:73).
//   vqa_pack_tokens     : ragged word-index lists -> padded token ids + attention mask with the START / END / truncation /
//                         padding conventions of Tokenizer.encode (utils/tokenizer.py:196-250); the string work (lower-casing,
//                         regex split, dictionary lookup) stays on the host.
//   vqa_image_resize    : transforms.Resize (= PIL.Image.resize BILINEAR, bit-exact) [+ RandomCrop window + flip] fused with
//                         ToTensor + Normalize for a ragged batch of decoded images (see below).
//   vqa_image_color_jitter : transforms.ColorJitter (= PIL ImageEnhance blends + HSV hue shift, bit-exact) fused with ToTensor + Normalize.
// All are pure byte / index movers: HBM-bound.
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

// ---------------------------------------------------------------------------------------------------------------------
// transforms.Resize((RH, RW)) of a PIL image (data/preprocess.py:70,90,118; api/inference.py:140-170) = PIL.Image.resize(BILINEAR):
// Pillow's src/libImaging/Resample.c, restated.  Separable, two passes with a uint8-ROUNDED intermediate:
//   coefficients  precompute_coeffs: scale = in/out, support = max(scale, 1), window [center - support, center + support] clipped to
//                 the axis, triangle weights in DOUBLE precision normalised by their sum, then normalize_coeffs_8bpc: fixed point with
//                 PRECISION_BITS = 32 - 8 - 2 fractional bits, rounded half away from zero;
//   each pass     out = clip8((2^21 + sum_x pixel[xmin + x] * k[x]) >> 22); horizontal first (ImagingResampleHorizontal_8bpc), then
//                 vertical; a pass whose axis keeps its size is skipped.
// Every step is integer arithmetic on bytes except the weights; those are computed with contraction off (no FMA) in the same
// operation order as the C source, so the uint8 result is bit-identical to PIL (tests/golden/resize_pil.npz from PIL itself).
// The vertical pass is fused with the optional RandomCrop window / horizontal flip and with ToTensor + Normalize + the NCHW
// layout of image_normalize_kernel, so a decoded image is read once and the float batch written once.
// ---------------------------------------------------------------------------------------------------------------------
#define RESIZE_PREC 22
#define RESIZE_GROUP 32                  // images per launch (their descriptors ride in the kernel arguments)

struct ResizeImg {                       // one image of a launch group
  long long in_off;                      // byte offset of its [H][W][3] pixels in the packed input
  long long tmp_off, xb_off, xk_off, yb_off, yk_off;    // byte offsets into the workspace
  int H, W, kx, ky, cy, cx;              // source size, taps per output column / row, crop origin inside the resized image
};
struct ResizeGroup { ResizeImg im[RESIZE_GROUP]; };

// one thread = one output column (t < RW) or row (t >= RW) of one image: bounds + fixed-point weights
__global__ __launch_bounds__(256) void resize_coef_kernel(ResizeGroup g, int n, int RH, int RW, uint8_t* __restrict__ ws) {
#pragma clang fp contract(off)
  const ResizeImg im = g.im[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= RW + RH) return;
  const bool horiz = t < RW;
  const int xx = horiz ? t : t - RW;
  const int in = horiz ? im.W : im.H, outn = horiz ? RW : RH, ks = horiz ? im.kx : im.ky;
  int* bounds = reinterpret_cast<int*>(ws + (horiz ? im.xb_off : im.yb_off)) + 2 * xx;
  int* k = reinterpret_cast<int*>(ws + (horiz ? im.xk_off : im.yk_off)) + (size_t)xx * ks;
  const double scale = (double)in / (double)outn;
  const double filterscale = scale < 1.0 ? 1.0 : scale;
  const double support = 1.0 * filterscale;
  const double center = 0.0 + (xx + 0.5) * scale;
  const double ss = 1.0 / filterscale;
  int xmin = (int)(center - support + 0.5);
  if (xmin < 0) xmin = 0;
  int xmax = (int)(center + support + 0.5);
  if (xmax > in) xmax = in;
  xmax -= xmin;
  double ww = 0.0;
  for (int x = 0; x < xmax; ++x) {
    double a = (x + xmin - center + 0.5) * ss;
    if (a < 0.0) a = -a;
    ww += a < 1.0 ? 1.0 - a : 0.0;
  }
  for (int x = 0; x < ks; ++x) {
    int v = 0;
    if (x < xmax) {
      double a = (x + xmin - center + 0.5) * ss;
      if (a < 0.0) a = -a;
      double w = a < 1.0 ? 1.0 - a : 0.0;
      if (ww != 0.0) w = w / ww;
      v = w < 0.0 ? (int)(-0.5 + w * (double)(1 << RESIZE_PREC)) : (int)(0.5 + w * (double)(1 << RESIZE_PREC));
    }
    k[x] = v;
  }
  bounds[0] = xmin; bounds[1] = xmax;
}

__device__ __forceinline__ uint8_t resize_clip8(int acc) {
  const int v = acc >> RESIZE_PREC;
  return (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
}

// horizontal pass: one thread = (source row h, output column cx + j) of one image, the three channels; uint8 intermediate [H][OW][3]
__global__ __launch_bounds__(256) void resize_h_kernel(ResizeGroup g, const uint8_t* __restrict__ in, uint8_t* __restrict__ ws, int OW) {
  const ResizeImg im = g.im[blockIdx.y];
  const long long t = (long long)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (long long)im.H * OW) return;
  const int h = (int)(t / OW), j = (int)(t - (long long)h * OW);
  const int xx = im.cx + j;
  const int* bounds = reinterpret_cast<const int*>(ws + im.xb_off) + 2 * xx;
  const int* k = reinterpret_cast<const int*>(ws + im.xk_off) + (size_t)xx * im.kx;
  const int xmin = bounds[0], xn = bounds[1];
  const uint8_t* row = in + im.in_off + ((size_t)h * im.W + xmin) * 3;
  int a0 = 1 << (RESIZE_PREC - 1), a1 = a0, a2 = a0;
  for (int x = 0; x < xn; ++x) {
    const int kv = k[x];
    a0 += (int)row[3 * x] * kv; a1 += (int)row[3 * x + 1] * kv; a2 += (int)row[3 * x + 2] * kv;
  }
  uint8_t* o = ws + im.tmp_off + ((size_t)h * OW + j) * 3;
  o[0] = resize_clip8(a0); o[1] = resize_clip8(a1); o[2] = resize_clip8(a2);
}

// vertical pass + crop + flip + ToTensor + Normalize: one thread = output pixel (i, j) of one image, the three channels
__global__ __launch_bounds__(256) void resize_v_kernel(ResizeGroup g, const uint8_t* __restrict__ in, const uint8_t* __restrict__ ws, int b0,
                                                       int RH, int RW, int OH, int OW, uint8_t* __restrict__ out_u8, float* __restrict__ out_f,
                                                       const uint8_t* __restrict__ flip, float m0, float m1, float m2, float s0, float s1, float s2) {
  const ResizeImg im = g.im[blockIdx.y];
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= OH * OW) return;
  const int i = t / OW, j = t - i * OW;
  const int b = b0 + blockIdx.y;
  // source of the column: the horizontal intermediate (already cropped: column j), or the input itself when the width is kept
  const bool hpass = im.W != RW;
  const uint8_t* src = hpass ? ws + im.tmp_off + (size_t)j * 3 : in + im.in_off + (size_t)(im.cx + j) * 3;
  const size_t pitch = (size_t)(hpass ? OW : im.W) * 3;
  const int yy = im.cy + i;
  uint8_t px[3];
  if (im.H != RH) {
    const int* bounds = reinterpret_cast<const int*>(ws + im.yb_off) + 2 * yy;
    const int* k = reinterpret_cast<const int*>(ws + im.yk_off) + (size_t)yy * im.ky;
    const int ymin = bounds[0], yn = bounds[1];
    int a0 = 1 << (RESIZE_PREC - 1), a1 = a0, a2 = a0;
    const uint8_t* p = src + (size_t)ymin * pitch;
    for (int y = 0; y < yn; ++y, p += pitch) {
      const int kv = k[y];
      a0 += (int)p[0] * kv; a1 += (int)p[1] * kv; a2 += (int)p[2] * kv;
    }
    px[0] = resize_clip8(a0); px[1] = resize_clip8(a1); px[2] = resize_clip8(a2);
  } else {
    const uint8_t* p = src + (size_t)yy * pitch;
    px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
  }
  const int jo = (flip && flip[b]) ? OW - 1 - j : j;
  if (out_u8) {
    uint8_t* o = out_u8 + (((size_t)b * OH + i) * OW + jo) * 3;
    o[0] = px[0]; o[1] = px[1]; o[2] = px[2];
  }
  if (out_f) {
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
    const size_t plane = (size_t)OH * OW;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tt = (float)px[c] / 255.0f;                    // ToTensor, then Normalize: torch's operation order
      out_f[((size_t)b * 3 + c) * plane + (size_t)i * OW + jo] = (tt - mean[c]) / sd[c];
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// transforms.ColorJitter(brightness, contrast, saturation, hue) on a PIL image (data/preprocess.py:77-82) is Pillow code (third
// party; 12.2.0 pinned by this image): ImageEnhance.Brightness / Contrast / Color = Image.blend(degenerate, image, factor) with the
// degenerate image black / the rounded mean of the L band / the L band, and the hue shift = RGB -> HSV, H += delta (uint8
// wrap-around), HSV -> RGB.  libImaging's Blend.c (float, truncation inside [0, 1], clip outside), Convert.c rgb2l (16.16 fixed
// point), rgb2hsv_row and hsv2rgb (float / double mix of the C source: double literals, float variables, round()) are restated with
// the same operation order and no FMA contraction, so every uint8 is bit-identical to PIL: tests/golden/jitter_pil.npz (real PIL),
// and the oracle restatement the GPU tests compare with is pinned over all 2^24 colours (tests/test_input_cpu.py).
// Contrast needs the mean of the whole image AFTER the adjustments that precede it in the image's permutation: pass 1 re-applies
// those per pixel and adds the L values into one 64-bit integer per image, pass 2 applies everything and normalises.
// ---------------------------------------------------------------------------------------------------------------------
__device__ __forceinline__ int jit_l(const int (&px)[3]) { return (px[0] * 19595 + px[1] * 38470 + px[2] * 7471 + 0x8000) >> 16; }

__device__ __forceinline__ int jit_blend(int a, int b, float alpha) {     // a: degenerate, b: image
#pragma clang fp contract(off)
  if (alpha == 0.0f) return a;
  if (alpha == 1.0f) return b;
  const float prod = alpha * (float)(b - a);
  const float t = (float)a + prod;
  if (alpha >= 0.0f && alpha <= 1.0f) return (int)t & 255;
  return t <= 0.0f ? 0 : (t >= 255.0f ? 255 : (int)t);
}

__device__ __forceinline__ void jit_hue(int (&px)[3], int delta) {
#pragma clang fp contract(off)
  const int r = px[0], g = px[1], b = px[2];
  const int maxc = max(r, max(g, b)), minc = min(r, min(g, b));
  int uh = 0, us = 0;
  const int uv = maxc;
  if (minc != maxc) {
    const float cr = (float)(maxc - minc);
    const float s = __fdiv_rn(cr, (float)maxc);
    const float rc = __fdiv_rn((float)(maxc - r), cr), gc = __fdiv_rn((float)(maxc - g), cr), bc = __fdiv_rn((float)(maxc - b), cr);
    float h;
    if (r == maxc) h = bc - gc;
    else if (g == maxc) h = (float)((2.0 + (double)rc) - (double)bc);
    else h = (float)((4.0 + (double)gc) - (double)rc);
    h = (float)fmod((double)h / 6.0 + 1.0, 1.0);
    uh = (int)((double)h * 255.0); uh = uh < 0 ? 0 : (uh > 255 ? 255 : uh);
    us = (int)((double)s * 255.0); us = us < 0 ? 0 : (us > 255 ? 255 : us);
  }
  uh = (uh + delta) & 255;
  if (us == 0) { px[0] = px[1] = px[2] = uv; return; }
  const double hf = (double)uh * 6.0 / 255.0;
  const double fl = floor(hf);
  const double f = (double)(float)(hf - fl);
  const double fs = (double)(float)((double)us / 255.0);
  const double v = (double)uv;
  const double one_m = 1.0 - f;
  int p = (int)round(v * (1.0 - fs)), q = (int)round(v * (1.0 - fs * f)), t = (int)round(v * (1.0 - fs * one_m));
  p = p < 0 ? 0 : (p > 255 ? 255 : p); q = q < 0 ? 0 : (q > 255 ? 255 : q); t = t < 0 ? 0 : (t > 255 ? 255 : t);
  switch ((int)fl % 6) {
    case 0: px[0] = uv; px[1] = t; px[2] = p; break;
    case 1: px[0] = q; px[1] = uv; px[2] = p; break;
    case 2: px[0] = p; px[1] = uv; px[2] = t; break;
    case 3: px[0] = p; px[1] = q; px[2] = uv; break;
    case 4: px[0] = t; px[1] = p; px[2] = uv; break;
    default: px[0] = uv; px[1] = p; px[2] = q; break;
  }
}

// FINAL = false: adjustments in front of the contrast step, then the L value into sums[b].  FINAL = true: all four, then the outputs.
template <bool FINAL>
__global__ __launch_bounds__(256) void color_jitter_kernel(const uint8_t* __restrict__ in, const uint8_t* __restrict__ order,
                                                           const float* __restrict__ factors, unsigned long long* __restrict__ sums,
                                                           int HW, uint8_t* __restrict__ out_u8, float* __restrict__ out_f,
                                                           float m0, float m1, float m2, float s0, float s1, float s2) {
  const int b = blockIdx.y;
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const bool live = t < HW;
  const float fb = factors[4 * b], fc = factors[4 * b + 1], fsat = factors[4 * b + 2], fh = factors[4 * b + 3];   // NaN: adjustment off
  const uint32_t ord = *reinterpret_cast<const uint32_t*>(order + 4 * b);
  int px[3] = {0, 0, 0};
  if (live) {
    const uint8_t* p = in + ((size_t)b * HW + t) * 3;
    px[0] = p[0]; px[1] = p[1]; px[2] = p[2];
  }
  bool stop = false;
#pragma unroll
  for (int k = 0; k < 4; ++k) {
    const int fn = (ord >> (8 * k)) & 0xff;
    if (stop) continue;
    if (fn == 0 && fb == fb) {
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = jit_blend(0, px[c], fb);
    } else if (fn == 1 && fc == fc) {
      if (!FINAL) { stop = true; continue; }
      const unsigned long long sum = sums[b];
      const int mean = (int)((double)sum / (double)HW + 0.5);
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = jit_blend(mean, px[c], fc);
    } else if (fn == 2 && fsat == fsat) {
      const int L = jit_l(px);
#pragma unroll
      for (int c = 0; c < 3; ++c) px[c] = jit_blend(L, px[c], fsat);
    } else if (fn == 3 && fh == fh) {
      jit_hue(px, (int)fh & 255);
    }
  }
  if (!FINAL) {
    // every image needs its sum only if contrast is on; summing regardless keeps the control flow uniform
    unsigned long long v = live ? (unsigned long long)jit_l(px) : 0ull;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    if ((threadIdx.x & 63) == 0 && fc == fc) atomicAdd(sums + b, v);
    return;
  }
  if (!live) return;
  if (out_u8) {
    uint8_t* o = out_u8 + ((size_t)b * HW + t) * 3;
    o[0] = (uint8_t)px[0]; o[1] = (uint8_t)px[1]; o[2] = (uint8_t)px[2];
  }
  if (out_f) {
    const float mean[3] = {m0, m1, m2}, sd[3] = {s0, s1, s2};
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float tt = (float)px[c] / 255.0f;                      // ToTensor, then Normalize: torch's operation order
      out_f[((size_t)b * 3 + c) * HW + t] = (tt - mean[c]) / sd[c];
    }
  }
}

static inline long long align16(long long v) { return (v + 15) & ~15LL; }
static inline int host_ksize(int in, int out) {
  const double scale = (double)in / (double)out;
  const double support = scale < 1.0 ? 1.0 : scale;
  return (int)ceil(support) * 2 + 1;
}
// workspace layout of image i (shared by the size query and the launch): returns the running end offset
static long long resize_layout(ResizeImg* im, long long off, int H, int W, int RH, int RW, int OW) {
  const int kx = host_ksize(W, RW), ky = host_ksize(H, RH);
  if (im) { im->kx = kx; im->ky = ky; }
  long long o = off;
  if (im) im->xb_off = o; o = align16(o + (long long)RW * 2 * 4);
  if (im) im->xk_off = o; o = align16(o + (long long)RW * kx * 4);
  if (im) im->yb_off = o; o = align16(o + (long long)RH * 2 * 4);
  if (im) im->yk_off = o; o = align16(o + (long long)RH * ky * 4);
  if (im) im->tmp_off = o; o = align16(o + (W != RW ? (long long)H * OW * 3 : 0));
  return o;
}

extern "C" {

long long vqa_image_resize_ws(int n, const int* H, const int* W, int RH, int RW, int OW) {
  if (n <= 0 || !H || !W || RH <= 0 || RW <= 0 || OW <= 0) return -1;
  long long off = 0;
  for (int i = 0; i < n; ++i) {
    if (H[i] <= 0 || W[i] <= 0) return -1;
    off = resize_layout(nullptr, off, H[i], W[i], RH, RW, OW);
  }
  return off;
}

int vqa_image_resize(const uint8_t* in, const long long* in_off, const int* H, const int* W, const int* crop_yx, int n, int RH, int RW,
                     int OH, int OW, uint8_t* out_u8, float* out_nchw, const uint8_t* flip, float mean0, float mean1, float mean2,
                     float std0, float std1, float std2, void* ws, long long ws_bytes, hipStream_t st) {
  if (!in || !in_off || !H || !W || n <= 0 || RH <= 0 || RW <= 0 || OH <= 0 || OW <= 0 || OH > RH || OW > RW) return VQA_EARG;
  if ((!out_u8 && !out_nchw) || !ws) return VQA_EARG;
  if (out_nchw && !(std0 != 0.f && std1 != 0.f && std2 != 0.f)) return VQA_EARG;
  if (vqa_image_resize_ws(n, H, W, RH, RW, OW) > ws_bytes) return VQA_EARG;
  for (int i = 0; i < n; ++i) {
    const int cy = crop_yx ? crop_yx[2 * i] : 0, cx = crop_yx ? crop_yx[2 * i + 1] : 0;
    if (cy < 0 || cx < 0 || cy + OH > RH || cx + OW > RW || in_off[i] < 0) return VQA_EARG;
  }
  long long off = 0;
  for (int b0 = 0; b0 < n; b0 += RESIZE_GROUP) {
    const int cnt = n - b0 < RESIZE_GROUP ? n - b0 : RESIZE_GROUP;
    ResizeGroup g;
    int hmax = 0;
    bool any_h = false;
    for (int i = 0; i < cnt; ++i) {
      ResizeImg& im = g.im[i];
      im.in_off = in_off[b0 + i]; im.H = H[b0 + i]; im.W = W[b0 + i];
      im.cy = crop_yx ? crop_yx[2 * (b0 + i)] : 0; im.cx = crop_yx ? crop_yx[2 * (b0 + i) + 1] : 0;
      off = resize_layout(&im, off, im.H, im.W, RH, RW, OW);
      if (im.W != RW) { any_h = true; if (im.H > hmax) hmax = im.H; }
    }
    for (int i = cnt; i < RESIZE_GROUP; ++i) g.im[i] = g.im[0];
    hipLaunchKernelGGL(resize_coef_kernel, dim3((unsigned)((RW + RH + 255) / 256), (unsigned)cnt), dim3(256), 0, st, g, cnt, RH, RW, (uint8_t*)ws);
    if (any_h) {
      ResizeGroup gh = g;                              // images that keep their width skip the pass (H = 0: no thread passes the bound)
      for (int i = 0; i < cnt; ++i) if (gh.im[i].W == RW) gh.im[i].H = 0;
      hipLaunchKernelGGL(resize_h_kernel, dim3((unsigned)(((long long)hmax * OW + 255) / 256), (unsigned)cnt), dim3(256), 0, st, gh, in,
                         (uint8_t*)ws, OW);
    }
    hipLaunchKernelGGL(resize_v_kernel, dim3((unsigned)((OH * OW + 255) / 256), (unsigned)cnt), dim3(256), 0, st, g, in, (const uint8_t*)ws, b0,
                       RH, RW, OH, OW, out_u8, out_nchw, flip, mean0, mean1, mean2, std0, std1, std2);
  }
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

int vqa_image_color_jitter(const uint8_t* in_hwc, const uint8_t* order, const float* factors, int B, int H, int W, uint8_t* out_u8,
                           float* out_nchw, float mean0, float mean1, float mean2, float std0, float std1, float std2,
                           unsigned long long* sums, hipStream_t st) {
  if (!in_hwc || !order || !factors || !sums || B <= 0 || H <= 0 || W <= 0 || (!out_u8 && !out_nchw)) return VQA_EARG;
  if ((long long)H * W > 0x7fffffffLL / 4 || B > 65535) return VQA_EARG;
  if (out_nchw && !(std0 != 0.f && std1 != 0.f && std2 != 0.f)) return VQA_EARG;
  const int HW = H * W;
  const dim3 grid((unsigned)((HW + 255) / 256), (unsigned)B);
  if (hipMemsetAsync(sums, 0, (size_t)B * sizeof(unsigned long long), st) != hipSuccess) return VQA_EARG;
  hipLaunchKernelGGL(color_jitter_kernel<false>, grid, dim3(256), 0, st, in_hwc, order, factors, sums, HW, (uint8_t*)nullptr, (float*)nullptr,
                     0.f, 0.f, 0.f, 1.f, 1.f, 1.f);
  hipLaunchKernelGGL(color_jitter_kernel<true>, grid, dim3(256), 0, st, in_hwc, order, factors, sums, HW, out_u8, out_nchw,
                     mean0, mean1, mean2, std0, std1, std2);
  VQA_LAUNCH_CHECK();
  return VQA_OK;
}

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
