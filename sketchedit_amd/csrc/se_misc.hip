// Memory-bound helper kernels of the SketchEdit forward (gfx950): final 12->{1,3} convs with fused
// tanh / sigmoid / composites, input packing (NCHW -> NHWC with masking / concat), layout conversion for
// the unit-test entry points, deterministic column reductions (global max / mean pool, L2 norm), and the
// host-side launch profiler.
#include "se_device.h"

#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <set>
#include <utility>

namespace se {

// ---- developer switches (se_kernels.h SE_OPTIONS) ---------------------------------------------------
namespace {
struct OptTable {
  std::atomic<int> v[OPT_COUNT];
  int dflt[OPT_COUNT];
  OptTable() { load(); }
  void load() {
    static const char* names[OPT_COUNT] = {
#define X(name, d) "SE_" #name,
        SE_OPTIONS(X)
#undef X
    };
    static const int builtin[OPT_COUNT] = {
#define X(name, d) d,
        SE_OPTIONS(X)
#undef X
    };
    for (int i = 0; i < OPT_COUNT; ++i) {
      const char* e = i == OPT_TEST_OFFSET_LIMIT ? nullptr : getenv(names[i]);      // (the test aid is not an environment switch)
      dflt[i] = e ? atoi(e) : builtin[i];
      v[i].store(dflt[i], std::memory_order_relaxed);
    }
  }
  static int index_of(const char* name) {
    static const char* names[OPT_COUNT] = {
#define X(name, d) #name,
        SE_OPTIONS(X)
#undef X
    };
    if (!name) return -1;
    if (!strncmp(name, "SE_", 3)) name += 3;
    for (int i = 0; i < OPT_COUNT; ++i)
      if (!strcmp(name, names[i])) return i;
    return -1;
  }
};
OptTable& opt_table() {
  static OptTable t;      // C++11: initialised once, thread-safe -- the only getenv calls of the library's launch path
  return t;
}
}  // namespace
int opt(int o) { return opt_table().v[o].load(std::memory_order_relaxed); }
// bumped by every change of the table: part of the key of a captured forward (SE_FLAG_GRAPH), so that a graph captured under
// one set of kernel forms is never replayed after the forms were switched
static std::atomic<long long> g_opt_epoch{0};
long long opt_epoch() { return g_opt_epoch.load(std::memory_order_relaxed); }
int opt_set(const char* name, int value) {
  const int i = OptTable::index_of(name);
  if (i < 0) return 1;
  opt_table().v[i].store(value, std::memory_order_relaxed);
  g_opt_epoch.fetch_add(1, std::memory_order_relaxed);
  return 0;
}
int opt_get(const char* name, int* value) {
  const int i = OptTable::index_of(name);
  if (i < 0) return 1;
  if (value) *value = opt(i);
  return 0;
}
void opt_reset() {
  OptTable& t = opt_table();
  for (int i = 0; i < OPT_COUNT; ++i) t.v[i].store(t.dflt[i], std::memory_order_relaxed);
  g_opt_epoch.fetch_add(1, std::memory_order_relaxed);
}

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
//
// What bounds it (round 3; the round-2 form was one pixel per lane, 9 pixel loads and 108 scalar FMAs per output channel):
// 324 FMAs per pixel are 69 us of VALU issue at 256x256 B=32 -- the kernel was VALU-bound, with the vector L1 (27 16-byte
// loads per pixel) right behind, and its 3.4x HBM over-fetch came from vertically adjacent row blocks landing on
// different XCDs (round-robin dispatch), so that no L2 ever saw a row twice.  Now:
//   * a lane owns a COLUMN STRIP of SR = 4 output rows and walks the SR + 2 input rows once: 4.5 pixel loads per output
//     instead of 9 (an input row feeds the <= 3 output rows that tap it while it is in registers);
//   * the multiply-adds are packed (v_pk_fma_f32: channel pairs (2i, 2i+1) into a two-lane accumulator, summed once at
//     the end) -- half the VALU issue slots;
//   * blocks are handed to the XCDs in contiguous ranges (xcd_tile), so the two halo rows a strip shares with its
//     vertical neighbours come out of the same L2.
// SR = output rows per lane (a ragged last strip is masked): 4, or 2 for calls of one or two small images, whose 4-row
// strips would occupy only a quarter of the CUs (one 256x256 image: 64 blocks)
template <int COUT, bool BF16, int SR>
__global__ __launch_bounds__(256) void small_conv_kernel(const SmallConvParams p) {
  const int HW = p.H * p.W;
  const int strips = (p.H + SR - 1) / SR;
  const long sidx = (long)xcd_tile(blockIdx.x, gridDim.x) * 256 + threadIdx.x;      // (image, strip, column)
  if (sidx >= (long)p.B * strips * p.W) return;
  const int b = (int)(sidx / ((long)strips * p.W));
  const int rem0 = (int)(sidx - (long)b * strips * p.W);
  const int sy = rem0 / p.W, x = rem0 - sy * p.W, y0 = sy * SR;
  f32x2 acc[SR][COUT];
#pragma unroll
  for (int o = 0; o < SR; ++o)
#pragma unroll
    for (int c = 0; c < COUT; ++c) acc[o][c] = (f32x2){p.b[c], 0.f};
  const f32x2* w2 = (const f32x2*)p.w;        // [COUT][9 taps][6 channel pairs]
  // zero padding through the buffer range check: a tap outside the image gets an out-of-range offset and the hardware
  // returns zeros (one select per pixel load instead of twelve per pixel; no branch).  32-bit offsets from the first
  // image of this launch (launch_small_conv splits batches whose activation exceeds 2 GB).
  constexpr int PXB = BF16 ? 32 : 48;         // bytes per input pixel
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.x, 0, (int)((unsigned)p.B * (unsigned)HW * (unsigned)PXB), 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
  // raw dwords of one input row: 3 pixels x 12 channels (fp32: 3 x 16 B per pixel; bf16: 16 B + 8 B per pixel)
  constexpr int RD = BF16 ? 6 : 12;
  auto load_row = [&](int r, unsigned (&raw)[3][RD]) {
    const int iy = y0 - 1 + r;
    const bool rowin = (unsigned)iy < (unsigned)p.H;
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      const int ix = x + kx - 1;
      const bool in = rowin && (unsigned)ix < (unsigned)p.W;
      const unsigned off = in ? (unsigned)((b * p.H + iy) * p.W + ix) * (unsigned)PXB : 0x80000000u;
      if (BF16) {
        const u32x4 a = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, 0, 0);
        const u32x2 c = __builtin_amdgcn_raw_buffer_load_b64(rs, (int)off, 16, 0);
        raw[kx][0] = a[0]; raw[kx][1] = a[1]; raw[kx][2] = a[2]; raw[kx][3] = a[3]; raw[kx][4] = c[0]; raw[kx][5 % RD] = c[1];
      } else {
#pragma unroll
        for (int q = 0; q < 3; ++q) {
          const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)off, q * 16, 0);
#pragma unroll
          for (int e = 0; e < 4; ++e) raw[kx][(q * 4 + e) % RD] = t[e];
        }
      }
    }
  };
  auto use_row = [&](int r, const unsigned (&raw)[3][RD]) {      // input row y0 - 1 + r feeds output row o with ky = r - o
#pragma unroll
    for (int kx = 0; kx < 3; ++kx) {
      f32x2 v[6];
#pragma unroll
      for (int i = 0; i < 6; ++i)
        v[i] = BF16 ? (f32x2){bf16_lo(raw[kx][i % RD]), bf16_hi(raw[kx][i % RD])}
                    : (f32x2){__uint_as_float(raw[kx][(2 * i) % RD]), __uint_as_float(raw[kx][(2 * i + 1) % RD])};
#pragma unroll
      for (int o = 0; o < SR; ++o) {
        const int ky = r - o;
        if (ky < 0 || ky > 2) continue;        // wave-uniform
#pragma unroll
        for (int c = 0; c < COUT; ++c) {
          const f32x2* wc = w2 + (c * 9 + ky * 3 + kx) * 6;
#pragma unroll
          for (int i = 0; i < 6; ++i) acc[o][c] = v[i] * wc[i] + acc[o][c];
        }
      }
    }
  };
  // two-deep software pipeline over the SR + 2 input rows: the loads of row r + 1 are in flight under the FMAs of row r.
  // The row loop is NOT unrolled (two rows per trip, so the buffers stay compile-time): unrolled, hipcc hoists all 54
  // loads and all 324 weight s_loads to the top and spills SGPRs into VGPR lanes (263 VGPRs, one wave per SIMD); rolled,
  // ky = r - o is a wave-uniform run-time value and the weights of the (<= 3) output rows a row feeds are fetched
  // through the scalar cache as they are needed.
  unsigned rawA[3][RD], rawB[3][RD];
  load_row(0, rawA);
#pragma unroll 1
  for (int r = 0; r < SR + 2; r += 2) {
    load_row(r + 1, rawB);
    use_row(r, rawA);
    if (r + 2 < SR + 2) load_row(r + 2, rawA);
    use_row(r + 1, rawB);
  }
#pragma unroll
  for (int o = 0; o < SR; ++o) {
    const int y = y0 + o;
    if (y >= p.H) break;
    const int rem = y * p.W + x;
    const long idx = (long)b * HW + rem;
    float a1[COUT];
#pragma unroll
    for (int c = 0; c < COUT; ++c) a1[c] = acc[o][c][0] + acc[o][c][1];
    if (p.mode == 4) {   // raw conv output (unit tests of the passthrough rule, utils.py:27)
#pragma unroll
      for (int c = 0; c < COUT; ++c) p.out_nchw[((long)b * COUT + c) * HW + rem] = a1[c];
      continue;
    }
    if (p.mode == 0) {
      const float m = sigmoidf_(a1[0]);
      p.out_nchw[p.out_bs ? (long)b * p.out_bs + rem : idx] = m;
      if (p.hard) p.hard[idx] = m > 0.5f ? 1.f : 0.f;
      continue;
    }
    if constexpr (COUT == 3) {
      float t[3];
#pragma unroll
      for (int c = 0; c < 3; ++c) t[c] = tanh_fast(a1[c]);
      if (p.out_nchw) {
#pragma unroll
        for (int c = 0; c < 3; ++c) p.out_nchw[((long)b * 3 + c) * HW + rem] = t[c];
      }
      if (p.mode == 2) {
        const float m = p.mask[idx];
        f32x4 ov;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          // xnow = stage1*mask + xin*(1-mask), xin = image*(1-mask)   (editline_g.py:124,179-180)
          const float xin = p.img[((long)b * 3 + c) * HW + rem] * (1.f - m);
          ov[c] = p.no_mask_coarse ? t[c] : t[c] * m + xin * (1.f - m);
        }
        ov[3] = 0.f;
        if (BF16) *(uint4*)((char*)p.xnow + idx * 16) = make_uint4(pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], 0.f), 0u, 0u);
        else *(f32x4*)(p.xnow + idx * 4) = ov;
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
}

hipError_t launch_small_conv(const SmallConvParams& p0, hipStream_t st) {
  // the kernel addresses its input through 32-bit byte offsets (buffer loads): batches whose activation reaches 2 GB are
  // run as several launches over sub-batches, every per-image pointer advanced accordingly
  const long HW = (long)p0.H * p0.W;
  const long pxb = p0.bf16 ? 32 : 48;
  if (HW * pxb >= (1l << 31)) return hipErrorInvalidValue;
  const int nbmax = (int)(((1l << 31) - 1) / (HW * pxb));
  ProfScope ps_(st, PL_SMALL_CONV);
  for (int b0 = 0; b0 < p0.B; b0 += nbmax) {
    SmallConvParams p = p0;
    p.B = p0.B - b0 < nbmax ? p0.B - b0 : nbmax;
    p.x = (const float*)((const char*)p0.x + (size_t)b0 * HW * pxb);
    const long cs = p0.mode == 4 ? p0.cout : (p0.mode == 0 ? 1 : 3);
    if (p0.out_nchw) p.out_nchw = p0.out_nchw + (size_t)b0 * (p0.out_bs ? p0.out_bs : cs * HW);
    if (p0.hard) p.hard = p0.hard + (size_t)b0 * HW;
    if (p0.img) p.img = p0.img + (size_t)b0 * 3 * HW;
    if (p0.mask) p.mask = p0.mask + (size_t)b0 * (p0.mask_bs ? p0.mask_bs : HW);
    if (p0.xnow) p.xnow = (float*)((char*)p0.xnow + (size_t)b0 * HW * 16);
    if (p0.composed) p.composed = p0.composed + (size_t)b0 * (p0.comp_bs ? p0.comp_bs : 3 * HW);
    if (p0.rgb8) p.rgb8 = p0.rgb8 + (size_t)b0 * HW * 3;
    if (p0.m8) p.m8 = p0.m8 + (size_t)b0 * HW;
    const long n4 = (long)p.B * ((p.H + 3) / 4) * p.W;
    const bool small = (n4 + 255) / 256 < 256;                       // fewer blocks than CUs: 2-row strips
    const long n = small ? (long)p.B * ((p.H + 1) / 2) * p.W : n4;
    const int grid = (int)((n + 255) / 256);
#define SE_SC_LAUNCH(CO, BF)                                                                                        \
    do {                                                                                                              \
      if (small) hipLaunchKernelGGL((small_conv_kernel<CO, BF, 2>), dim3(grid), dim3(256), 0, st, p);                 \
      else hipLaunchKernelGGL((small_conv_kernel<CO, BF, 4>), dim3(grid), dim3(256), 0, st, p);                       \
    } while (0)
    if (p.cout == 1 && !p.bf16) SE_SC_LAUNCH(1, false);
    else if (p.cout == 3 && !p.bf16) SE_SC_LAUNCH(3, false);
    else if (p.cout == 1) SE_SC_LAUNCH(1, true);
    else if (p.cout == 3) SE_SC_LAUNCH(3, true);
    else return hipErrorInvalidValue;
#undef SE_SC_LAUNCH
  }
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

// ---- input dequantisation (data/testimage_dataset.py:89-111 of the reference: ToTensor + Normalize(0.5, 0.5), sketch > 0):
// 4 consecutive pixels per thread -- 12 + 4 input bytes, four float4 stores.  The 256 possible image values come from a table
// the HOST computed with IEEE fp32 division in the dataset's operation order (se_api.hip dequant_table): bit-identical to the
// float tensors the CPU dataset builds, whatever the device's division does.
__global__ void dequant_u8_kernel(const unsigned char* __restrict__ rgb, const unsigned char* __restrict__ sk8,
                                  const float* __restrict__ lut, float* __restrict__ image, float* __restrict__ sketch, int B, int HW) {
  __shared__ float T[256];
  T[threadIdx.x] = lut[threadIdx.x];
  __syncthreads();
  const long q = (long)blockIdx.x * 256 + threadIdx.x;      // group of 4 pixels
  if (q >= (long)B * HW / 4) return;
  const long pix = q * 4;
  const int b = (int)(pix / HW);
  const long in = pix - (long)b * HW;
  if (image) {
    const unsigned* src = (const unsigned*)(rgb + pix * 3);
    const unsigned w0 = src[0], w1 = src[1], w2 = src[2];
    unsigned char v[12];
#pragma unroll
    for (int i = 0; i < 4; ++i) { v[i] = (w0 >> (8 * i)) & 255u; v[4 + i] = (w1 >> (8 * i)) & 255u; v[8 + i] = (w2 >> (8 * i)) & 255u; }
#pragma unroll
    for (int c = 0; c < 3; ++c)
      *(float4*)(image + ((long)b * 3 + c) * HW + in) = make_float4(T[v[c]], T[v[3 + c]], T[v[6 + c]], T[v[9 + c]]);
  }
  if (sketch) {
    const unsigned w = *(const unsigned*)(sk8 + pix);
    *(float4*)(sketch + pix) = make_float4((w & 0xffu) ? 1.f : 0.f, (w & 0xff00u) ? 1.f : 0.f, (w & 0xff0000u) ? 1.f : 0.f, (w & 0xff000000u) ? 1.f : 0.f);
  }
}
hipError_t launch_dequantize_u8(const unsigned char* rgb, const unsigned char* sk8, const float* lut, float* image, float* sketch,
                                int B, int H, int W, hipStream_t st) {
  const long nq = (long)B * H * W / 4;
  ProfScope ps_(st, PL_LAYOUT);
  hipLaunchKernelGGL(dequant_u8_kernel, dim3((unsigned)((nq + 255) / 256)), dim3(256), 0, st, rgb, sk8, lut, image, sketch, B, H * W);
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
// launch_vecbias (se_kernels.h): grid (image, half of the 192 packed rows), 768 threads = 96 rows x 8 channel groups: a thread
// forms the nine per-tap partial sums of its row over C1 / 8 channels (loads coalesced over the rows, independent
// accumulators), the groups are combined through LDS in a fixed order (deterministic), and the nine border configurations
// are sums of the per-tap values.  ~10 us per step; the round-3 first version (one thread per row, 864 dependent loads)
// took 290.
__global__ __launch_bounds__(768) void vecbias_kernel(const float* __restrict__ wv, const float* __restrict__ vec, float* __restrict__ T, int C1, int round_vec) {
  __shared__ float red[8][9][96];
  const int b = blockIdx.x, half = blockIdx.y;
  const int r = threadIdx.x % 96, g = threadIdx.x / 96, n = half * 96 + r;
  const int cg = C1 / 8;                      // channels per group (C1 % 8 == 0: checked by the launcher)
  const float* v = vec + (size_t)b * C1 + g * cg;
  float part[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) {
    const float* w = wv + ((size_t)t * C1 + g * cg) * 192 + n;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    int c = 0;
    auto vv = [&](int i) { return round_vec ? bf16_lo(pack_bf16x2(v[i], 0.f) & 0xffffu) : v[i]; };
    for (; c + 4 <= cg; c += 4) {
      a0 = fmaf(w[(size_t)(c + 0) * 192], vv(c + 0), a0);
      a1 = fmaf(w[(size_t)(c + 1) * 192], vv(c + 1), a1);
      a2 = fmaf(w[(size_t)(c + 2) * 192], vv(c + 2), a2);
      a3 = fmaf(w[(size_t)(c + 3) * 192], vv(c + 3), a3);
    }
    for (; c < cg; ++c) a0 = fmaf(w[(size_t)c * 192], vv(c), a0);
    part[t] = (a0 + a1) + (a2 + a3);
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) red[g][t][r] = part[t];
  __syncthreads();
  if (g == 0) {
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      float a = red[0][t][r];
#pragma unroll
      for (int k = 1; k < 8; ++k) a += red[k][t][r];
      part[t] = a;
    }
#pragma unroll
    for (int cy = 0; cy < 3; ++cy)
#pragma unroll
      for (int cx = 0; cx < 3; ++cx) {
        float a = 0.f;
#pragma unroll
        for (int ky = 0; ky < 3; ++ky)
#pragma unroll
          for (int kx = 0; kx < 3; ++kx) {
            const bool in = !((cy == 0 && ky == 0) || (cy == 2 && ky == 2) || (cx == 0 && kx == 0) || (cx == 2 && kx == 2));
            if (in) a += part[ky * 3 + kx];
          }
        T[((size_t)b * 9 + cy * 3 + cx) * 192 + n] = a;
      }
  }
}
hipError_t launch_vecbias(const float* wv, const float* vec, float* T, int B, int C1, hipStream_t st, int round_vec) {
  if (C1 % 8) return hipErrorInvalidValue;
  ProfScope ps_(st, PL_COLREDUCE);
  hipLaunchKernelGGL(vecbias_kernel, dim3(B, 2), dim3(768), 0, st, wv, vec, T, C1, round_vec);
  return hipGetLastError();
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
