// gfx950 (MI355X, CDNA4) kernels for the SketchEdit inference forward pass.
//
// Layout: every internal activation is NHWC fp32 (channels innermost, multiples of 4 floats = one
// 16-byte "granule").  The hot kernel is a gather-GEMM on the fp32 MFMA (v_mfma_f32_16x16x4_f32):
//   A operand (16 rows)  = packed output channels (weights),
//   B operand (16 cols)  = output pixels,
//   K                    = flattened (tap, input channel), consumed in chunks of 32 floats.
// Per chunk a workgroup stages  [pixels][32]  and  [channels][32]  tiles into LDS with direct
// global->LDS DMA (global_load_lds_dwordx4); out-of-image taps are fetched from a zero page, so
// there is no im2col buffer and no padding copy.  LDS rows are 128 B; the 16-byte slot of a
// granule is XOR-swizzled with (row>>1)&7 so that the ds_read_b128 fragment reads of 16 rows at
// one k-slot are bank-conflict free (swizzle applied on the *source* side, LDS image stays
// lane-linear as the DMA requires).
//
// Reference semantics restated here (paths relative to /root/reference):
//   gated conv        models/networks/utils.py:9-33      (ELU/ReLU(x) * sigmoid(y), zero padding)
//   upsample + conv   models/networks/utils.py:35-51     (nearest x2: src = dst >> 1)
//   attention         models/networks/splitcam.py:37-108, 132-153
#include "se_kernels.h"

namespace se {

typedef float f32x4 __attribute__((ext_vector_type(4)));

#define DEVFN __device__ __forceinline__

DEVFN void glds16(const void* g, void* lds_wave_base) {
  // 64 lanes x 16 B: lane i lands at lds_wave_base + 16*i
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g,
                                   (__attribute__((address_space(3))) void*)lds_wave_base, 16, 0, 0);
}

DEVFN float elu1(float x) { return x > 0.f ? x : expm1f(x); }
DEVFN float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ------------------------------------------------------------------------------------------------
// shared MFMA core: one 32-k chunk, wave tile NT x PT tiles of 16x16
// ------------------------------------------------------------------------------------------------
template <int NT, int PT>
DEVFN void mfma_chunk(f32x4 (&acc)[NT][PT], const char* __restrict__ Wt, const char* __restrict__ Xt_wave,
                      int off0, int off1) {
  // Wt: [NT*16 rows][128 B], Xt_wave: this wave's [PT*16 rows][128 B]; off0/off1: per-lane byte offset
  // ((lane&15)*128 + swizzled slot) for k-half 0 / 1.
#pragma unroll
  for (int half = 0; half < 2; ++half) {
    const int off = half ? off1 : off0;
    f32x4 xb[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xb[pt] = *(const f32x4*)(Xt_wave + pt * 2048 + off);
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 wa = *(const f32x4*)(Wt + nt * 2048 + off);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
          acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[r], xb[pt][r], acc[nt][pt], 0, 0, 0);
      }
    }
  }
}

// ------------------------------------------------------------------------------------------------
// gated convolution
// ------------------------------------------------------------------------------------------------
template <int NT, int PT, bool MIXED>
__global__ __launch_bounds__(256) void gconv_kernel(const GConvParams p) {
  constexpr int PIX = PT * 64;
  constexpr int NP = NT * 16;
  constexpr int XBYTES = PIX * 128, WBYTES = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XBYTES;
  int4* rowtab = (int4*)(smem + 2 * XBYTES + 2 * WBYTES);

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile_base = blockIdx.x * PIX;

  const int HoWo = p.Ho * p.Wo;
  for (int r = tid; r < PIX; r += 256) {
    const int pidx = tile_base + r;
    int4 e;
    if (pidx < p.total_pix) {
      const int b = pidx / HoWo, rem = pidx - b * HoWo;
      const int oy = rem / p.Wo, ox = rem - oy * p.Wo;
      e = make_int4(b, oy * p.stride, ox * p.stride, 1);
    } else {
      e = make_int4(0, -(1 << 20), -(1 << 20), 0);
    }
    rowtab[r] = e;
  }
  __syncthreads();

  // staging role of this lane: physical slot ps of rows (rbk*8 + lane>>3); logical slot s
  const int ps = lane & 7;
  const int s_log = ps ^ (4 * (w & 1) + (lane >> 4));
  // fragment-read offsets
  const int swz = (lane >> 1) & 7;
  const int off0 = (lane & 15) * 128 + ((((lane >> 4)) ^ swz) << 4);
  const int off1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ swz) << 4);

  auto stage = [&](int ch, int buf) {
    // ---- X tile
    const int gi = ch * 8 + s_log;
    const int tap = gi / p.CG, cg = gi - tap * p.CG;
    const bool tapok = tap < p.T;
    const int ky = tap / p.KW, kx = tap - ky * p.KW;
    const int dy = ky * p.dil - p.pad, dx = kx * p.dil - p.pad;
    const bool first = cg < p.C0g;
    const float* base = first ? p.src0 : p.src1;
    const int cs = first ? p.C0 : p.C1;
    const int coff = (first ? cg : cg - p.C0g) * 4;
    const bool vec = (!first) && p.src1_vec;
    char* xdst = Xb + buf * XBYTES;
#pragma unroll
    for (int i = 0; i < PT * 2; ++i) {
      const int rbk = i * 4 + w;
      const int4 e = rowtab[rbk * 8 + (lane >> 3)];
      int iy = e.y + dy, ix = e.z + dx;
      const bool ok = tapok && (unsigned)iy < (unsigned)p.Hlim && (unsigned)ix < (unsigned)p.Wlim;
      iy >>= p.ushift;
      ix >>= p.ushift;
      const long pix = vec ? (long)e.x : ((long)e.x * p.Hin + iy) * p.Win + ix;
      const float* g = ok ? base + pix * cs + coff : p.zeros;
      glds16(g, xdst + rbk * 1024);
    }
    // ---- W tile (host-packed LDS image, plain linear copy)
    char* wdst = Wb + buf * WBYTES;
    const float* wsrc = p.wpk + (size_t)ch * NP * 32 + lane * 4;
    for (int rbk = w; rbk < NT * 2; rbk += 4) glds16(wsrc + rbk * 256, wdst + rbk * 1024);
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();
  for (int ch = 0; ch < p.nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < p.nch) stage(ch + 1, buf ^ 1);
    mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    __syncthreads();
  }

  // ---- epilogue: bias, gate, NHWC store (lane: 4 consecutive channels of pixel lane&15)
  const int q = lane >> 4;
  if (!MIXED) {
    constexpr int NF = NT / 2;
#pragma unroll
    for (int nt = 0; nt < NF; ++nt) {
      const int c0 = nt * 16 + q * 4;
      const f32x4 bf = *(const f32x4*)(p.bias + c0);
      const f32x4 bg = *(const f32x4*)(p.bias + NF * 16 + c0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int pidx = tile_base + (w * PT + pt) * 16 + (lane & 15);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float f = acc[nt][pt][r] + bf[r];
          const float g = acc[nt + NF][pt][r] + bg[r];
          const float a = p.act == 0 ? elu1(f) : fmaxf(f, 0.f);
          o[r] = a * sigmoidf_(g);
        }
        if (pidx < p.total_pix) *(f32x4*)(p.dst + (size_t)pidx * p.G + c0) = o;
      }
    }
  } else {
    // tile rows 0-7: features 8nt..8nt+7 (lanes q=0,1), rows 8-15: matching gates (lanes q=2,3)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c0 = nt * 8 + (q & 1) * 4;
      const f32x4 bq = *(const f32x4*)(p.bias + nt * 16 + q * 4);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int pidx = tile_base + (w * PT + pt) * 16 + (lane & 15);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float v = acc[nt][pt][r] + bq[r];
          const float g = __shfl_xor(v, 32);   // lanes 0-31 receive their gate value
          const float a = p.act == 0 ? elu1(v) : fmaxf(v, 0.f);
          o[r] = a * sigmoidf_(g);
        }
        if (q < 2 && c0 < p.G && pidx < p.total_pix) *(f32x4*)(p.dst + (size_t)pidx * p.G + c0) = o;
      }
    }
  }
}

template <int NT, int PT, bool MIXED>
static hipError_t launch_gconv_t(const GConvParams& p, hipStream_t st) {
  constexpr int PIX = PT * 64;
  constexpr int LDS = 2 * PIX * 128 + 2 * NT * 16 * 128 + PIX * 16;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)gconv_kernel<NT, PT, MIXED>,
                                       hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = (p.total_pix + PIX - 1) / PIX;
  hipLaunchKernelGGL((gconv_kernel<NT, PT, MIXED>), dim3(grid), dim3(256), LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_gconv(int cfg, const GConvParams& p, hipStream_t st) {
  switch (cfg) {
    case GC_N192: return launch_gconv_t<12, 4, false>(p, st);
    case GC_N96: return launch_gconv_t<6, 4, false>(p, st);
    case GC_N48: return launch_gconv_t<3, 8, true>(p, st);
    case GC_N24: return launch_gconv_t<2, 8, true>(p, st);
  }
  return hipErrorInvalidValue;
}

// ------------------------------------------------------------------------------------------------
// final 3x3 conv of each decoder: 12 -> COUT raw, + tanh / sigmoid / composites (VALU, memory bound)
// ------------------------------------------------------------------------------------------------
template <int COUT>
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
        const f32x4* src = (const f32x4*)(p.x + ((long)(b * p.H + iy) * p.W + ix) * 12);
        const f32x4 v0 = src[0], v1 = src[1], v2 = src[2];
        const float v[12] = {v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3], v2[0], v2[1], v2[2], v2[3]};
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
    p.out_nchw[idx] = m;
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
      *(f32x4*)(p.xnow + idx * 4) = o;
    } else if (p.mode == 3 && p.composed) {
      const float m = p.mask[idx];
#pragma unroll
      for (int c = 0; c < 3; ++c) {
        const float im = p.img[((long)b * 3 + c) * HW + rem];
        p.composed[((long)b * 3 + c) * HW + rem] = t[c] * m + im * (1.f - m);   // editline2_model.py:132
      }
    }
  }
}

hipError_t launch_small_conv(const SmallConvParams& p, hipStream_t st) {
  const long n = (long)p.B * p.H * p.W;
  const int grid = (int)((n + 255) / 256);
  if (p.cout == 1)
    hipLaunchKernelGGL(small_conv_kernel<1>, dim3(grid), dim3(256), 0, st, p);
  else if (p.cout == 3)
    hipLaunchKernelGGL(small_conv_kernel<3>, dim3(grid), dim3(256), 0, st, p);
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
  *(f32x4*)(style8 + idx * 8) = s0;
  *(f32x4*)(style8 + idx * 8 + 4) = s1;
}
hipError_t launch_pack_g(const float* x, const float* x2, const float* mask, const float* mask2, const float* guide,
                         float* coarse8, float* style8, int B, int H, int W, int no_mask_cc, int joint,
                         hipStream_t st) {
  const long n = (long)B * H * W;
  hipLaunchKernelGGL(pack_g_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, x, x2, mask, mask2, guide,
                     coarse8, style8, B, H * W, no_mask_cc, joint);
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
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int Cstride, int H, int W, hipStream_t st) {
  const long n = (long)B * C * H * W;
  hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, src, dst, B, C, Cstride,
                     H * W);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// column reduce over pixels (global max pool / mean / L2 norm), deterministic two-stage
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void colreduce_partial_kernel(const float* __restrict__ x, float* __restrict__ partial,
                                                                int HW, int C, int op) {
  // grid (SPLITS, B); block 256 threads = (256/C' groups) ... generic: thread t handles channel t%C, pixel lane t/C
  __shared__ float red[256];
  const int b = blockIdx.y, sp = blockIdx.x;
  const int per = (HW + COLREDUCE_SPLITS - 1) / COLREDUCE_SPLITS;
  const int p0 = sp * per, p1 = min(HW, p0 + per);
  const int groups = 256 / C;            // C <= 256
  const int c = threadIdx.x % C, g = threadIdx.x / C;
  float a = op == 0 ? -INFINITY : 0.f;
  if (g < groups) {
    for (int pp = p0 + g; pp < p1; pp += groups) {
      const float v = x[((long)b * HW + pp) * C + c];
      a = op == 0 ? fmaxf(a, v) : (op == 1 ? a + v : fmaf(v, v, a));
    }
  }
  red[threadIdx.x] = a;
  __syncthreads();
  if (threadIdx.x < C) {
    float r = red[threadIdx.x];
    for (int gg = 1; gg < groups; ++gg) {
      const float v = red[gg * C + threadIdx.x];
      r = op == 0 ? fmaxf(r, v) : r + v;
    }
    partial[((long)b * COLREDUCE_SPLITS + sp) * C + threadIdx.x] = r;
  }
}
__global__ void colreduce_final_kernel(const float* __restrict__ partial, float* __restrict__ out, int HW, int C, int op,
                                       int total) {
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
}
hipError_t launch_colreduce(const float* x, float* partial, float* out, int B, int HW, int C, int op, hipStream_t st) {
  if (C > 256) return hipErrorInvalidValue;
  hipLaunchKernelGGL(colreduce_partial_kernel, dim3(COLREDUCE_SPLITS, B), dim3(256), 0, st, x, partial, HW, C, op);
  hipLaunchKernelGGL(colreduce_final_kernel, dim3((B * C + 255) / 256), dim3(256), 0, st, partial, out, HW, C, op,
                     B * C);
  return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// contextual attention
// ------------------------------------------------------------------------------------------------
__global__ void att_prep_kernel(const AttParams p) {
  // xn = x * rn ; valid[b][j] = (mean over the 4x4 patch of (1 - avgpool4(mask))) > th
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long nx = (long)p.B * p.h * p.w * 24;    // granules
  if (idx < nx) {
    const int cg = idx % 24;
    const long pix = idx / 24;
    const int b = pix / (p.h * p.w);
    f32x4 v = *(const f32x4*)(p.x + idx * 4);
    const f32x4 r = *(const f32x4*)(p.rn + b * 96 + cg * 4);
    *(f32x4*)(p.xn + idx * 4) = v * r;
  }
  if (idx < (long)p.B * p.Lp) {
    const int b = idx / p.Lp, j = idx - (long)b * p.Lp;
    float val = 0.f;
    if (j < p.L) {
      const int jy = j / p.ws, jx = j - jy * p.ws;
      const int H = p.h * 4, W = p.w * 4;
      // sum of the hole mask over the 16x16 full-resolution window (exact in fp32: integers <= 256)
      float hole = 0.f;
      for (int yy = 0; yy < 16; ++yy)
        for (int xx = 0; xx < 16; ++xx) hole += p.hard[((long)b * H + jy * 8 + yy) * W + jx * 8 + xx];
      const float mm = 1.f - hole * (1.f / 256.f);       // mean of (1 - avg_pool2d(mask,4,4)) over the patch
      val = mm > p.th ? 1.f : 0.f;
    }
    p.valid[idx] = val;
  }
}

// S[b][i][j] = scale * valid[j] * <K_j, Q_i>   (query-major so the softmax axis is contiguous)
template <int NT, int PT>
__global__ __launch_bounds__(256) void att_score_kernel(const AttParams p) {
  constexpr int PIX = PT * 64, NP = NT * 16;
  constexpr int XBYTES = PIX * 128, WBYTES = NP * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XBYTES;
  int* qtab = (int*)(smem + 2 * XBYTES + 2 * WBYTES);   // pixel offset of patch origin, or -1
  int* ktab = qtab + PIX;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int q0 = blockIdx.x * PIX, k0 = blockIdx.y * NP;
  for (int r = tid; r < PIX + NP; r += 256) {
    const int i = r < PIX ? q0 + r : k0 + (r - PIX);
    int v = -1;
    if (i < p.L) {
      const int py = i / p.ws, px = i - py * p.ws;
      v = (b * p.h + 2 * py) * p.w + 2 * px;
    }
    qtab[r] = v;
  }
  __syncthreads();
  const int ps = lane & 7;
  const int swz = (lane >> 1) & 7;
  const int off0 = (lane & 15) * 128 + ((((lane >> 4)) ^ swz) << 4);
  const int off1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ swz) << 4);

  auto stage = [&](int ch, int buf) {
    char* xdst = Xb + buf * XBYTES;
    char* wdst = Wb + buf * WBYTES;
#pragma unroll
    for (int i = 0; i < PT * 2; ++i) {
      const int rbk = i * 4 + w;
      const int s_log = ps ^ (4 * (rbk & 1) + (lane >> 4));
      const int gi = ch * 8 + s_log;              // granule in [0, 16*24)
      const int tap = gi / 24, cg = gi - tap * 24;
      const int doff = ((tap >> 2) * p.w + (tap & 3)) * 96 + cg * 4;
      const int o = qtab[rbk * 8 + (lane >> 3)];
      const float* g = o >= 0 ? p.x + (long)o * 96 + doff : p.zeros;
      glds16(g, xdst + rbk * 1024);
    }
    for (int rbk = w; rbk < NT * 2; rbk += 4) {
      const int s_log = ps ^ (4 * (rbk & 1) + (lane >> 4));
      const int gi = ch * 8 + s_log;
      const int tap = gi / 24, cg = gi - tap * 24;
      const int doff = ((tap >> 2) * p.w + (tap & 3)) * 96 + cg * 4;
      const int o = ktab[rbk * 8 + (lane >> 3)];
      const float* g = o >= 0 ? p.xn + (long)o * 96 + doff : p.zeros;
      glds16(g, wdst + rbk * 1024);
    }
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NCH = 48;   // 16 taps * 96 ch / 32
  stage(0, 0);
  __syncthreads();
  for (int ch = 0; ch < NCH; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < NCH) stage(ch + 1, buf ^ 1);
    mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    __syncthreads();
  }
  const int q = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int j = k0 + nt * 16 + q * 4;
    if (j >= p.Lp) continue;
    const f32x4 v = *(const f32x4*)(p.valid + (long)b * p.Lp + j);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int i = q0 + (w * PT + pt) * 16 + (lane & 15);
      if (i < p.L) *(f32x4*)(p.S + ((long)b * p.L + i) * p.Lp + j) = acc[nt][pt] * v * p.scale;
    }
  }
}

// softmax over keys j < L of each query row; pad columns [L, Lp) are set to 0.  One wave per row.
__global__ __launch_bounds__(256) void att_softmax_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)p.B * p.L) return;
  float* s = p.S + row * p.Lp;
  float m = -INFINITY;
  for (int j = lane; j < p.L; j += 64) m = fmaxf(m, s[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int j = lane; j < p.L; j += 64) {
    const float e = expf(s[j] - m);
    s[j] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  for (int j = lane; j < p.Lp; j += 64) s[j] = j < p.L ? s[j] * inv : 0.f;
}

// out[b, pos, c] = sum over the <=4 patches covering pos of sum_j P[i][j] * x[b, 2j + (ky,kx), c]
// One workgroup: one parity class (py,px) of output pixels, PT*64 of them, all 96 channels.
template <int PT>
__global__ __launch_bounds__(256) void att_pv_kernel(const AttParams p) {
  constexpr int NT = 6;
  constexpr int PIX = PT * 64;
  constexpr int XBYTES = PIX * 128, VBYTES = 32 * 384;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;                        // P tiles  [PIX][32]
  char* Vb = smem + 2 * XBYTES;           // V tiles  [32 keys][96 ch]
  int2* ptab = (int2*)(smem + 2 * XBYTES + 2 * VBYTES);   // (yy, xx) of the class pixel, or (-1<<20)

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, cls = blockIdx.y, py = cls >> 1, px = cls & 1;
  const int ch_ = p.h >> 1, cw_ = p.w >> 1;      // class image size
  const int t0 = blockIdx.x * PIX;
  for (int r = tid; r < PIX; r += 256) {
    const int i = t0 + r;
    int2 e = make_int2(-(1 << 20), -(1 << 20));
    if (i < ch_ * cw_) e = make_int2(i / cw_, i % cw_);
    ptab[r] = e;
  }
  __syncthreads();
  const int ps = lane & 7;
  const int s_log = ps ^ (4 * (w & 1) + (lane >> 4));
  const int swz = (lane >> 1) & 7;
  const int off0 = (lane & 15) * 128 + ((((lane >> 4)) ^ swz) << 4);
  const int off1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ swz) << 4);
  const int jchunks = p.Lp >> 5;
  const int nch = 4 * jchunks;

  auto stage = [&](int ch, int buf) {
    const int combo = ch / jchunks, jc = ch - combo * jchunks;
    const int a = combo >> 1, bb = combo & 1;
    char* xdst = Xb + buf * XBYTES;
#pragma unroll
    for (int i = 0; i < PT * 2; ++i) {
      const int rbk = i * 4 + w;
      const int2 e = ptab[rbk * 8 + (lane >> 3)];
      const int iy = e.x - a, ix = e.y - bb;
      const bool ok = (unsigned)iy < (unsigned)p.hs && (unsigned)ix < (unsigned)p.ws;
      const float* g = ok ? p.S + ((long)b * p.L + iy * p.ws + ix) * p.Lp + jc * 32 + s_log * 4 : p.zeros;
      glds16(g, xdst + rbk * 1024);
    }
    // V tile: key j -> pixel (2jy + py + 2a, 2jx + px + 2bb), 96 channels = 24 granules; 12 wave-instr
    char* vdst = Vb + buf * VBYTES;
    for (int it = w; it < 12; it += 4) {
      const int gidx = it * 64 + lane;          // granule index in [0, 768)
      const int jr = gidx / 24, cg = gidx - jr * 24;
      const int j = jc * 32 + jr;
      const float* g = p.zeros;
      if (j < p.L) {
        const int jy = j / p.ws, jx = j - jy * p.ws;
        g = p.x + ((long)(b * p.h + 2 * jy + py + 2 * a) * p.w + 2 * jx + px + 2 * bb) * 96 + cg * 4;
      }
      glds16(g, vdst + it * 1024);
    }
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) stage(ch + 1, buf ^ 1);
    const char* Xt = Xb + buf * XBYTES + w * PT * 2048;
    const float* Vt = (const float*)(Vb + buf * VBYTES);
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int off = half ? off1 : off0;
      f32x4 xb[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) xb[pt] = *(const f32x4*)(Xt + pt * 2048 + off);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = half * 16 + (lane >> 4) * 4 + r;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float a = Vt[j * 96 + nt * 16 + (lane & 15)];
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
            acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[pt][r], acc[nt][pt], 0, 0, 0);
        }
      }
    }
    __syncthreads();
  }
  const int q = lane >> 4;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int i = t0 + (w * PT + pt) * 16 + (lane & 15);
    if (i >= ch_ * cw_) continue;
    const int yy = i / cw_, xx = i - yy * cw_;
    float* o = p.out + ((long)(b * p.h + 2 * yy + py) * p.w + 2 * xx + px) * 96;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *(f32x4*)(o + nt * 16 + q * 4) = acc[nt][pt];
  }
}

hipError_t launch_attention(const AttParams& p, hipStream_t st) {
  {
    const long n = (long)p.B * p.h * p.w * 24;
    hipLaunchKernelGGL(att_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  }
  {
    constexpr int NT = 4, PT = 4;
    constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128 + (PT * 64 + NT * 16) * 4;
    static bool set = false;
    if (!set) {
      hipError_t e = hipFuncSetAttribute((const void*)att_score_kernel<NT, PT>,
                                         hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
      if (e != hipSuccess) return e;
      set = true;
    }
    dim3 grid((p.L + PT * 64 - 1) / (PT * 64), (p.L + NT * 16 - 1) / (NT * 16), p.B);
    hipLaunchKernelGGL((att_score_kernel<NT, PT>), grid, dim3(256), LDS, st, p);
  }
  {
    const long rows = (long)p.B * p.L;
    hipLaunchKernelGGL(att_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
  }
  {
    constexpr int PT = 4;
    constexpr int LDS = 2 * PT * 64 * 128 + 2 * 32 * 384 + PT * 64 * 8;
    static bool set = false;
    if (!set) {
      hipError_t e = hipFuncSetAttribute((const void*)att_pv_kernel<PT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                         LDS);
      if (e != hipSuccess) return e;
      set = true;
    }
    const int cpix = (p.h >> 1) * (p.w >> 1);
    dim3 grid((cpix + PT * 64 - 1) / (PT * 64), 4, p.B);
    hipLaunchKernelGGL((att_pv_kernel<PT>), grid, dim3(256), LDS, st, p);
  }
  return hipGetLastError();
}

}  // namespace se
