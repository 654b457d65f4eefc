// Memory-bound helper kernels of the SketchEdit forward (gfx950): final 12->{1,3} convs with fused
// tanh / sigmoid / composites, input packing (NCHW -> NHWC with masking / concat), layout conversion for
// the unit-test entry points, deterministic column reductions (global max / mean pool, L2 norm), and the
// host-side launch profiler.
#include "se_device.h"

#include <mutex>
#include <set>
#include <utility>

namespace se {

// ---- profiler plumbing ----------------------------------------------------------------------------
static thread_local Profiler* g_prof = nullptr;
static thread_local double g_next_flops = 0.0, g_next_bytes = 0.0, g_next_exec = 0.0;
static thread_local const char* g_next_name = nullptr;
static thread_local long g_next_blocks = 0;
void set_profiler(Profiler* p) { g_prof = p; }
void set_launch_grid(long blocks) { g_next_blocks = blocks; }
void set_launch_cost(double flops, double bytes, const char* name, double exec_flops) {
  g_next_flops = flops; g_next_bytes = bytes; g_next_name = name; g_next_exec = exec_flops < 0.0 ? flops : exec_flops;
}

hipError_t ensure_max_lds(const void* func, int bytes) {
  static std::mutex mu;
  static std::set<std::pair<int, const void*>> done;
  int dev = 0;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess) return e;
  std::lock_guard<std::mutex> lk(mu);
  const auto key = std::make_pair(dev, func);
  if (done.count(key)) return hipSuccess;
  e = hipFuncSetAttribute(func, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
  if (e == hipSuccess) done.insert(key);
  return e;
}
const char* prof_label_name(int l) {
  static const char* n[PL_COUNT] = {"gconv_n192", "gconv_n96", "gconv_n48", "gconv_n24", "wino_n192", "wino_n96", "wino_up96", "small_conv", "pack",
                                    "colreduce", "att_prep", "att_score", "att_softmax", "att_boxsum", "att_pv", "layout"};
  return (l >= 0 && l < PL_COUNT) ? n[l] : "?";
}

ProfScope::ProfScope(hipStream_t s, int label) : st(s) {
  Profiler* p = g_prof;
  const double fl = g_next_flops, by = g_next_bytes, ex = g_next_exec;
  const char* nm = g_next_name;
  const long nb = g_next_blocks;
  g_next_blocks = 0;
  g_next_flops = g_next_bytes = g_next_exec = 0.0;
  g_next_name = nullptr;
  if (!p || !p->on) return;
  Profiler::Rec r;
  r.label = label; r.name = nm; r.flops = fl; r.exec_flops = ex; r.bytes = by; r.blocks = nb;
  // events come from a pool created at se_profile_enable(): creating them here cost ~10 us of host time per launch,
  // let the GPU run dry between kernels and inflated the measured durations of short kernels
  if (p->used + 2 > p->pool.size()) return;
  r.a = p->pool[p->used++];
  r.b = p->pool[p->used++];
  (void)hipEventRecord(r.a, st);
  p->recs.push_back(r);
  idx = (int)p->recs.size() - 1;
}
ProfScope::~ProfScope() {
  if (idx >= 0) (void)hipEventRecord(g_prof->recs[idx].b, st);
}


// ------------------------------------------------------------------------------------------------
// final 3x3 conv of each decoder: 12 -> COUT raw, + tanh / sigmoid / composites (VALU, memory bound)
// ------------------------------------------------------------------------------------------------
// BF16: the 12-channel input is stored as bf16 NHWC with a 16-channel pixel stride (channels 12-15 are zero); the
// stage-2 input written by mode 2 is then bf16 NHWC8.  Arithmetic and the NCHW outputs stay fp32.
template <int COUT, bool BF16>
__global__ __launch_bounds__(256) void small_conv_kernel(const SmallConvParams p) {
  const int HW = p.H * p.W;
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)p.B * HW) return;
  const int b = idx / HW, rem = idx - (long)b * HW;
  const int y = rem / p.W, x = rem - y * p.W;
  float acc[COUT];
#pragma unroll
  for (int c = 0; c < COUT; ++c) acc[c] = p.b[c];
#pragma unroll
  for (int ky = 0; ky < 3; ++ky) {
    const int iy = y + ky - 1;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = x + kx - 1;
      if ((unsigned)iy < (unsigned)p.H && (unsigned)ix < (unsigned)p.W) {
        float v[12];
        if (BF16) {
          const uint4 a = *(const uint4*)((const char*)p.x + ((long)(b * p.H + iy) * p.W + ix) * 32);
          const uint2 c = *(const uint2*)((const char*)p.x + ((long)(b * p.H + iy) * p.W + ix) * 32 + 16);
          const unsigned u[6] = {a.x, a.y, a.z, a.w, c.x, c.y};
#pragma unroll
          for (int i = 0; i < 6; ++i) { v[2 * i] = bf16_lo(u[i]); v[2 * i + 1] = bf16_hi(u[i]); }
        } else {
          const f32x4* src = (const f32x4*)(p.x + ((long)(b * p.H + iy) * p.W + ix) * 12);
          const f32x4 v0 = src[0], v1 = src[1], v2 = src[2];
          const float vv[12] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
#pragma unroll
          for (int i = 0; i < 12; ++i) v[i] = vv[i];
        }
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          const float* wc = p.w + (c * 9 + ky * 3 + kx) * 12;
#pragma unroll
          for (int i = 0; i < 12; ++i) acc[c] = fmaf(v[i], wc[i], acc[c]);
        }
      }
    }
  }
  if (p.mode == 4) {   // raw conv output (unit tests of the passthrough rule, utils.py:27)
#pragma unroll
    for (int c = 0; c < COUT; ++c) p.out_nchw[((long)b * COUT + c) * HW + rem] = acc[c];
    return;
  }
  if (p.mode == 0) {
    const float m = sigmoidf_(acc[0]);
    p.out_nchw[p.out_bs ? (long)b * p.out_bs + rem : idx] = m;
    if (p.hard) p.hard[idx] = m > 0.5f ? 1.f : 0.f;
    return;
  }
  if constexpr (COUT == 3) {
    float t[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) t[c] = tanhf(acc[c]);
    if (p.out_nchw) {
#pragma unroll
      for (int c = 0; c < 3; ++c) p.out_nchw[((long)b * 3 + c) * HW + rem] = t[c];
    }
    if (p.mode == 2) {
      const float m = p.mask[idx];
      f32x4 o;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        // xnow = stage1*mask + xin*(1-mask), xin = image*(1-mask)   (editline_g.py:124,179-180)
        const float xin = p.img[((long)b * 3 + c) * HW + rem] * (1.f - m);
        o[c] = p.no_mask_coarse ? t[c] : t[c] * m + xin * (1.f - m);
      }
      o[3] = 0.f;
      if (BF16) *(uint4*)((char*)p.xnow + idx * 16) = make_uint4(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], 0.f), 0u, 0u);
      else *(f32x4*)(p.xnow + idx * 4) = o;
    } else if (p.mode == 3 && (p.composed || p.rgb8 || p.m8)) {
      const float m = p.mask[p.mask_bs ? (long)b * p.mask_bs + rem : idx];
      const long cb = p.comp_bs ? (long)b * p.comp_bs : (long)b * 3 * HW;
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float im = p.img[((long)b * 3 + c) * HW + rem];
        const float v = t[c] * m + im * (1.f - m);                              // editline2_model.py:132
        if (p.composed) p.composed[cb + (long)c * HW + rem] = v;
        // test.py:25-27: (x + 1) / 2 * 255 -> uint8, same fp32 operation order, truncation, no clamp; HWC as test.py:35
        if (p.rgb8) p.rgb8[idx * 3 + c] = (unsigned char)(int)(((v + 1.f) * 0.5f) * 255.f);
      }
      if (p.m8) p.m8[idx] = (unsigned char)(int)(m * 255.f);
    }
  }
}

hipError_t launch_small_conv(const SmallConvParams& p, hipStream_t st) {
  const long n = (long)p.B * p.H * p.W;
  const int grid = (int)((n + 255) / 256);
  ProfScope ps_(st, PL_SMALL_CONV);
  if (p.cout == 1 && !p.bf16)
    hipLaunchKernelGGL((small_conv_kernel<1, false>), dim3(grid), dim3(256), 0, st, p);
  else if (p.cout == 3 && !p.bf16)
    hipLaunchKernelGGL((small_conv_kernel<3, false>), dim3(grid), dim3(256), 0, st, p);
  else if (p.cout == 1)
    hipLaunchKernelGGL((small_conv_kernel<1, true>), dim3(grid), dim3(256), 0, st, p);
  else if (p.cout == 3)
    hipLaunchKernelGGL((small_conv_kernel<3, true>), dim3(grid), dim3(256), 0, st, p);
  else
    return hipErrorInvalidValue;
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// packing / layout
// ------------------------------------------------------------------------------------------------
__global__ void pack_m_kernel(const float* __restrict__ image, const float* __restrict__ sketch,
                              float* __restrict__ dst, int B, int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const int b = idx / HW, rem = idx - (long)b * HW;
  f32x4 o;
  o[0] = image[((long)b * 3 + 0) * HW + rem];
  o[1] = image[((long)b * 3 + 1) * HW + rem];
  o[2] = image[((long)b * 3 + 2) * HW + rem];
  o[3] = sketch[idx];
  *(f32x4*)(dst + idx * 4) = o;   // editline2_g.py:62
}
hipError_t launch_pack_m(const float* image, const float* sketch, float* dst4, int B, int H, int W, hipStream_t st) {
  const long n = (long)B * H * W;
  ProfScope ps_(st, PL_PACK);
  hipLaunchKernelGGL(pack_m_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, image, sketch, dst4, B, H * W);
  return hipGetLastError();
}

__global__ void pack_g_kernel(const float* __restrict__ x, const float* __restrict__ x2, const float* __restrict__ mask,
                              const float* __restrict__ mask2, const float* __restrict__ guide,
                              float* __restrict__ coarse8, float* __restrict__ style8, int B, int HW, int no_mask_cc,
                              int joint) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const int b = idx / HW, rem = idx - (long)b * HW;
  const float m = mask[idx], m2 = mask2[idx], g = guide[idx];
  f32x4 c0, c1, s0, s1;
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const long o = ((long)b * 3 + c) * HW + rem;
    c0[c] = x[o] * (1.f - m);                       // editline_g.py:124
    s0[c] = no_mask_cc ? x2[o] : x2[o] * m2;        // :120-123
  }
  c0[3] = g;                                        // :131
  c1 = (f32x4){m, 0.f, 0.f, 0.f};
  s0[3] = joint ? g * 0.f : g;                      // :132-135
  s1 = (f32x4){m2, 0.f, 0.f, 0.f};
  *(f32x4*)(coarse8 + idx * 8) = c0;
  *(f32x4*)(coarse8 + idx * 8 + 4) = c1;
  if (joint) {            // 4-channel style input: (x2*m2, m2) -- the guide channel is guide * 0 (editline_g.py:132-133)
    s0[3] = m2;
    *(f32x4*)(style8 + idx * 4) = s0;
  } else {
    *(f32x4*)(style8 + idx * 8) = s0;
    *(f32x4*)(style8 + idx * 8 + 4) = s1;
  }
}
hipError_t launch_pack_g(const float* x, const float* x2, const float* mask, const float* mask2, const float* guide,
                         float* coarse8, float* style8, int B, int H, int W, int no_mask_cc, int joint,
                         hipStream_t st) {
  const long n = (long)B * H * W;
  ProfScope ps_(st, PL_PACK);
  hipLaunchKernelGGL(pack_g_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, x2, mask, mask2, guide,
                     coarse8, style8, B, H * W, no_mask_cc, joint);
  return hipGetLastError();
}

// bf16 forms (BASELINE config 5): every network input becomes one 16-byte granule per pixel (NHWC8 bf16)
__global__ void pack_m16_kernel(const float* __restrict__ image, const float* __restrict__ sketch, uint4* __restrict__ dst,
                                int B, int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const int b = idx / HW, rem = idx - (long)b * HW;
  dst[idx] = make_uint4(pack_bf16x2(image[((long)b * 3 + 0) * HW + rem], image[((long)b * 3 + 1) * HW + rem]),
                        pack_bf16x2(image[((long)b * 3 + 2) * HW + rem], sketch[idx]), 0u, 0u);
}
hipError_t launch_pack_m16(const float* image, const float* sketch, float* dst8, int B, int H, int W, hipStream_t st) {
  const long n = (long)B * H * W;
  ProfScope ps_(st, PL_PACK);
  hipLaunchKernelGGL(pack_m16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, image, sketch, (uint4*)dst8, B, H * W);
  return hipGetLastError();
}
__global__ void pack_g16_kernel(const float* __restrict__ x, const float* __restrict__ x2, const float* __restrict__ mask,
                                const float* __restrict__ mask2, const float* __restrict__ guide, uint4* __restrict__ coarse8,
                                uint4* __restrict__ style8, int B, int HW, int no_mask_cc, int joint) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long)B * HW) return;
  const int b = idx / HW, rem = idx - (long)b * HW;
  const float m = mask[idx], m2 = mask2[idx], g = guide[idx];
  float c[3], s[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const long o = ((long)b * 3 + k) * HW + rem;
    c[k] = x[o] * (1.f - m);                        // editline_g.py:124
    s[k] = no_mask_cc ? x2[o] : x2[o] * m2;         // :120-123
  }
  coarse8[idx] = make_uint4(pack_bf16x2(c[0], c[1]), pack_bf16x2(c[2], g), pack_bf16x2(m, 0.f), 0u);      // :131
  // joint_train_inp: (x2*m2, m2) -- the guide channel is guide * 0 (:132-133), wconv1 runs without that weight column
  style8[idx] = joint ? make_uint4(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], m2), 0u, 0u)
                      : make_uint4(pack_bf16x2(s[0], s[1]), pack_bf16x2(s[2], g), pack_bf16x2(m2, 0.f), 0u);
}
hipError_t launch_pack_g16(const float* x, const float* x2, const float* mask, const float* mask2, const float* guide,
                           float* coarse8, float* style8, int B, int H, int W, int no_mask_cc, int joint, hipStream_t st) {
  const long n = (long)B * H * W;
  ProfScope ps_(st, PL_PACK);
  hipLaunchKernelGGL(pack_g16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, x2, mask, mask2, guide,
                     (uint4*)coarse8, (uint4*)style8, B, H * W, no_mask_cc, joint);
  return hipGetLastError();
}
// unit-test layout converters: NCHW fp32 <-> NHWC bf16 with a padded channel stride
__global__ void nchw_to_nhwc16_kernel(const float* __restrict__ src, unsigned short* __restrict__ dst, int B, int C, int Cpad,
                                      int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*HW*Cpad
  if (idx >= (long)B * HW * Cpad) return;
  const int c = idx % Cpad;
  const long pix = idx / Cpad;
  const int b = pix / HW, rem = pix - (long)b * HW;
  dst[idx] = (unsigned short)(pack_bf16x2(c < C ? src[((long)b * C + c) * HW + rem] : 0.f, 0.f) & 0xffffu);
}
hipError_t launch_nchw_to_nhwc16(const float* src, float* dst, int B, int C, int Cpad, int H, int W, hipStream_t st) {
  const long n = (long)B * H * W * Cpad;
  ProfScope ps_(st, PL_LAYOUT);
  hipLaunchKernelGGL(nchw_to_nhwc16_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, (unsigned short*)dst, B, C,
                     Cpad, H * W);
  return hipGetLastError();
}
__global__ void nhwc16_to_nchw_kernel(const unsigned short* __restrict__ src, float* __restrict__ dst, int B, int C, int Cs,
                                      int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*C*HW (NCHW order)
  if (idx >= (long)B * C * HW) return;
  const int rem = idx % HW;
  const long bc = idx / HW;
  const int c = bc % C, b = bc / C;
  dst[idx] = __uint_as_float((unsigned)src[((long)b * HW + rem) * Cs + c] << 16);
}
hipError_t launch_nhwc16_to_nchw(const float* src, float* dst, int B, int C, int Cstride, int H, int W, hipStream_t st) {
  const long n = (long)B * C * H * W;
  ProfScope ps_(st, PL_LAYOUT);
  hipLaunchKernelGGL(nhwc16_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, (const unsigned short*)src, dst,
                     B, C, Cstride, H * W);
  return hipGetLastError();
}

__global__ void nchw_to_nhwc_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int Cpad,
                                    int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*HW*Cpad
  if (idx >= (long)B * HW * Cpad) return;
  const int c = idx % Cpad;
  const long pix = idx / Cpad;
  const int b = pix / HW, rem = pix - (long)b * HW;
  dst[idx] = c < C ? src[((long)b * C + c) * HW + rem] : 0.f;
}
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int B, int C, int Cpad, int H, int W, hipStream_t st) {
  const long n = (long)B * H * W * Cpad;
  ProfScope ps_(st, PL_LAYOUT);
  hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, B, C, Cpad,
                     H * W);
  return hipGetLastError();
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ src, float* __restrict__ dst, int B, int C, int Cs,
                                    int HW) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;   // over B*C*HW (NCHW order)
  if (idx >= (long)B * C * HW) return;
  const int rem = idx % HW;
  const long bc = idx / HW;
  const int c = bc % C, b = bc / C;
  dst[idx] = src[((long)b * HW + rem) * Cs + c];
}
// ---- output quantisation (test.py:25-27): 4 consecutive pixels per thread, 12 + 4 output bytes
__global__ void quantize_u8_kernel(const float* __restrict__ comp, const float* __restrict__ mask,
                                   unsigned char* __restrict__ rgb, unsigned char* __restrict__ m8, int B, int HW) {
  const long q = (long)blockIdx.x * blockDim.x + threadIdx.x;      // group of 4 pixels
  const long nq = (long)B * HW / 4;
  if (q >= nq) return;
  const long pix = q * 4;
  const int b = (int)(pix / HW);
  const long in = pix - (long)b * HW;
  if (rgb) {
    unsigned char o[12];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const float4 v = *(const float4*)(comp + ((long)b * 3 + c) * HW + in);
      const float f[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
      for (int i = 0; i < 4; ++i) o[i * 3 + c] = (unsigned char)(int)(((f[i] + 1.f) * 0.5f) * 255.f);   // (x+1)/2*255, truncated
    }
    unsigned* dst = (unsigned*)(rgb + pix * 3);
    dst[0] = o[0] | (o[1] << 8) | (o[2] << 16) | ((unsigned)o[3] << 24);
    dst[1] = o[4] | (o[5] << 8) | (o[6] << 16) | ((unsigned)o[7] << 24);
    dst[2] = o[8] | (o[9] << 8) | (o[10] << 16) | ((unsigned)o[11] << 24);
  }
  if (m8) {
    const float4 v = *(const float4*)(mask + pix);
    *(unsigned*)(m8 + pix) = (unsigned)(unsigned char)(int)(v.x * 255.f) | ((unsigned)(unsigned char)(int)(v.y * 255.f) << 8) |
                             ((unsigned)(unsigned char)(int)(v.z * 255.f) << 16) | ((unsigned)(unsigned char)(int)(v.w * 255.f) << 24);
  }
}
hipError_t launch_quantize_u8(const float* composed, const float* mask, unsigned char* rgb, unsigned char* m8, int B, int H,
                              int W, hipStream_t st) {
  const long nq = (long)B * H * W / 4;
  ProfScope ps_(st, PL_LAYOUT);
  hipLaunchKernelGGL(quantize_u8_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, composed, mask, rgb, m8, B, H * W);
  return hipGetLastError();
}

hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int Cstride, int H, int W, hipStream_t st) {
  const long n = (long)B * C * H * W;
  ProfScope ps_(st, PL_LAYOUT);
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, B, C, Cstride,
                     H * W);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// column reduce over pixels (global max pool / mean / L2 norm), deterministic two-stage
// ------------------------------------------------------------------------------------------------
template <bool BF16>
__global__ __launch_bounds__(256) void colreduce_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                                int HW, int C, int op) {
  // grid (SPLITS, B).  A thread owns one 16-byte granule of channels (4 fp32 / 8 bf16) and walks the split's pixels
  // `groups` apart, four loads in flight; the groups are then combined through LDS in a fixed order (deterministic).
  constexpr int V = BF16 ? 8 : 4;
  __shared__ float red[256 * V];
  const int b = blockIdx.y, sp = blockIdx.x;
  const int per = (HW + COLREDUCE_SPLITS - 1) / COLREDUCE_SPLITS;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const int CV = C / V;                  // granules per pixel (C % V == 0, CV <= 256: checked by the launcher)
  const int groups = 256 / CV;
  const int cv = threadIdx.x % CV, g = threadIdx.x / CV;
  float a[V];
#pragma unroll
  for (int e = 0; e < V; ++e) a[e] = op == 0 ? -INFINITY : 0.f;
  auto acc = [&](const uint4 u) {
    float v[V];
    if (BF16) {
      v[0] = bf16_lo(u.x); v[1] = bf16_hi(u.x); v[2] = bf16_lo(u.y); v[3] = bf16_hi(u.y);
      if (V == 8) { v[4] = bf16_lo(u.z); v[5] = bf16_hi(u.z); v[6] = bf16_lo(u.w); v[7] = bf16_hi(u.w); }
    } else {
      v[0] = __uint_as_float(u.x); v[1] = __uint_as_float(u.y); v[2] = __uint_as_float(u.z); v[3] = __uint_as_float(u.w);
    }
#pragma unroll
    for (int e = 0; e < V; ++e) a[e] = op == 0 ? fmaxf(a[e], v[e]) : (op == 1 ? a[e] + v[e] : fmaf(v[e], v[e], a[e]));
  };
  if (g < groups) {
    const uint4* row = (const uint4*)x + (long)b * HW * CV + cv;
    int pp = p0 + g;
    for (; pp + 3 * groups < p1; pp += 4 * groups) {
      const uint4 u0 = row[(long)pp * CV], u1 = row[(long)(pp + groups) * CV], u2 = row[(long)(pp + 2 * groups) * CV],
                  u3 = row[(long)(pp + 3 * groups) * CV];
      acc(u0); acc(u1); acc(u2); acc(u3);
    }
    for (; pp < p1; pp += groups) acc(row[(long)pp * CV]);
  }
#pragma unroll
  for (int e = 0; e < V; ++e) red[threadIdx.x * V + e] = a[e];
  __syncthreads();
  for (int c = threadIdx.x; c < C; c += 256) {
    const int cv2 = c / V, e = c - cv2 * V;
    float r = red[cv2 * V + e];
    for (int gg = 1; gg < groups; ++gg) {
      const float v = red[(gg * CV + cv2) * V + e];
      r = op == 0 ? fmaxf(r, v) : r + v;
    }
    partial[((long)b * COLREDUCE_SPLITS + sp) * C + c] = r;
  }
}
__global__ void colreduce_final_kernel(const float* __restrict__ partial, float* __restrict__ out, unsigned short* out16,
                                       int HW, int C, int op, int total) {
  const int idx = blockIdx.x * 256 + threadIdx.x;   // over B*C
  if (idx >= total) return;
  const int b = idx / C, c = idx - b * C;
  float r = op == 0 ? -INFINITY : 0.f;
  for (int sp = 0; sp < COLREDUCE_SPLITS; ++sp) {
    const float v = partial[((long)b * COLREDUCE_SPLITS + sp) * C + c];
    r = op == 0 ? fmaxf(r, v) : r + v;
  }
  if (op == 1) r = r / (float)HW;
  if (op == 2) r = 1.f / sqrtf(r + 1e-8f);
  out[idx] = r;
  if (out16) out16[idx] = (unsigned short)(pack_bf16x2(r, 0.f) & 0xffffu);
}
hipError_t launch_colreduce(const float* x, float* partial, float* out, int B, int HW, int C, int op, hipStream_t st,
                            int x_bf16, float* out_bf16) {
  if (C > 256 || C % (x_bf16 ? 8 : 4)) return hipErrorInvalidValue;
  ProfScope ps_(st, PL_COLREDUCE);
  if (x_bf16) hipLaunchKernelGGL(colreduce_partial_kernel<true>, dim3(COLREDUCE_SPLITS, B), dim3(256), 0, st, x, partial, HW, C, op);
  else hipLaunchKernelGGL(colreduce_partial_kernel<false>, dim3(COLREDUCE_SPLITS, B), dim3(256), 0, st, x, partial, HW, C, op);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, partial, out,
                     (unsigned short*)out_bf16, HW, C, op, B * C);
  return hipGetLastError();
}

}  // namespace se
