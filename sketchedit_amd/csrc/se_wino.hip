// Winograd F(2x2,3x3) form of the 3x3 stride-1 gated convolution 96 -> 192 (the dominant layer shape:
// conv5-10, wconv5-10, xconv6-10, pmconv5-10, conv12, ... of /root/reference/models/networks/editline_g.py:48-99
// and editline2_g.py:22-37), any dilation d with h % 2d == 0 and w % 2d == 0.
//
//   Y = A^T [ (G g G^T) .* (B^T x B) ] A      per 2x2 output tile / 4x4 input tile, summed over input channels
//   B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]   G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]   A^T = [1 1 1 0; 0 1 -1 -1]
//
// 16 transform positions x K=96 instead of 9 taps x K=96 per 4 outputs: 2.25x fewer multiply-adds, all transform
// coefficients are 0, +-1, +-1/2 (exact in fp32).  A dilated conv is the same conv on the d x d polyphase
// sub-images, so a tile's 4x4 inputs are (y0 + (i-1)d, x0 + (j-1)d) and its outputs (y0 + a d, x0 + b d).
//
// One workgroup = 8 waves (two per SIMD) = 64 tiles x all 192 packed channels; wave w owns 16 tiles (MFMA columns)
// and one half of the channels (3 feature tiles + their 3 gate tiles, so the gate is a register epilogue).
// Loop over it = (position, 32-k chunk), 48 iterations, double-buffered LDS, one barrier per iteration:
//   * X tile [64 tiles][32 k]: each lane builds ONE granule = (B^T x B) at this position from 4 raw input granules
//     (global loads issued at the start of the previous iteration) and writes it to LDS -- the input transform is
//     fused, no transformed tensor ever exists in HBM;
//   * W tile [192][32 k] of the host-transformed weights G g G^T by LDS-DMA;
//   * 48 MFMAs per wave into the position accumulator; after the last chunk of a position it is folded into the
//     four output accumulators with the A^T coefficients (inverse transform in registers).
// Schedule (from s_memtime stamps: of ~4.1k cycles per iteration only 3.1k are MFMA pipe time; ~1.0-1.6k was spent
// waiting for the pipe to drain before the inverse adds and ~250 issuing the DMA): the fold of position P runs in
// the MFMA shadow of the NEXT iteration (the accumulator is double-buffered), and the DMA / global-load issue for
// iteration it+1 sits between the two MFMA halves, fenced with sched_barrier so hipcc keeps it there.
// Other layouts tried on the GPU and NOT faster (kept out of the tree; all land at ~205-235 us per launch with the
// MFMA pipe ~58 % busy): raw granules resident in registers with a 3-deep W ring (spills at 256 VGPRs), 16-wave
// workgroups with 3 channel tiles per wave and an LDS gate exchange (spills at 128 VGPRs), two 4-wave workgroups
// per CU (344 VGPRs needed), and an "LDS patch" form (raw 64 KiB input patch staged once per chunk, B fragments
// built straight from it, no global load / X tile in the loop, one or two positions per barrier): identical
// time, i.e. neither the gather traffic nor the barrier count limits this kernel -- what all variants share is
// short MFMA bursts (24 per k-half) behind freshly issued LDS fragment reads with both waves of a SIMD in phase
// and no register room to double-buffer the fragments.
#include "se_device.h"

#include <cstdlib>

// Developer aid, compiled only with -DSE_WINO_TRACE (tools/wino_trace.py builds a separate debug library):
// s_memtime stamps of block 0 / wave 0 at the phase boundaries of every iteration.
#ifdef SE_WINO_TRACE
__device__ unsigned long long g_wino_trace[96 * 8];
extern "C" int se_debug_wino_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_trace), sizeof(unsigned long long) * 48 * 8);
}
#define WINO_STAMP(k)                                                   \
  do {                                                                  \
    if (blockIdx.x == 0 && w == 0) {                                    \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
      if (lane == 0) g_wino_trace[it * 8 + (k)] = t_;                   \
    }                                                                   \
  } while (0)
#else
#define WINO_STAMP(k)
#endif

namespace se {

// NCHK = 32-channel chunks per position: 3 for one 96-channel source, 6 for the two-source layers
// (conv11: features + pooled style vector, allconv11: cat([x_hallu, pm]) -- editline_g.py:166-167,211).
template <int NCHK>
__global__ __launch_bounds__(512, 2) void wino_kernel(const WinoParams p) {
  constexpr int TILES = 64;
  constexpr int XB = TILES * 128, WB = 192 * 128;
  constexpr int NIT = 16 * NCHK;       // 16 positions x NCHK chunks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chh = w & 1, tg = w >> 1;          // channel half, tile group
  const int tile_base = (p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x) * TILES;
  const int tpi = p.th * p.tw;                 // tiles per image

  // tile -> (batch, first output pixel).  iy walks the tile grid; y0 = 2d*(iy/d) + iy%d
  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    b = t / tpi;
    const int rem = t - b * tpi;
    const int iy = rem / p.tw, ix = rem - iy * p.tw;
    const int qy = iy / p.d, qx = ix / p.d;
    y0 = 2 * p.d * qy + (iy - qy * p.d);
    x0 = 2 * p.d * qx + (ix - qx * p.d);
  };

  // ---- staging role: granule (row = tile tid>>3, physical slot tid&7)
  const int srow = tid >> 3, ps = tid & 7;
  const int s_log = ps ^ ((srow >> 1) & 7);
  int yo[4], xo[4];     // pixel-row offset (b*h + y)*w resp. x of the 4x4 input tile, or -1 if outside / invalid tile
  int bimg;             // batch index of this lane's tile (address of the per-image vector source)
  {
    const int t = tile_base + srow;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
    bimg = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + (i - 1) * p.d, x = x0 + (i - 1) * p.d;
      yo[i] = (t < p.total_tiles && (unsigned)y < (unsigned)p.h) ? (b * p.h + y) * p.w : -1;
      xo[i] = ((unsigned)x < (unsigned)p.w) ? x : -1;
    }
  }
  const unsigned lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  auto load_x = [&](int it, f32x4 (&r)[4]) {
    const int pos = it / NCHK, chunk = it - pos * NCHK;    // uniform
    const int xi = pos >> 2, nu = pos & 3;
    const int ya = xi == 0 ? yo[0] : yo[1], yb = xi == 3 ? yo[3] : yo[2];
    const int xa = nu == 0 ? xo[0] : xo[1], xb = nu == 3 ? xo[3] : xo[2];
    // always load from a valid (clamped) address; zero padding is applied to the data in write_x
    const int ya_c = max(ya, 0), yb_c = max(yb, 0), xa_c = max(xa, 0), xb_c = max(xb, 0);
    if (NCHK == 6 && chunk >= 3 && p.src1_vec) {
      // spatially constant second source (pooled style vector): one value per image, still zero padded
      const f32x4 v = *(const f32x4*)(p.src1 + ((size_t)bimg * 96 + ((chunk - 3) * 8 + s_log) * 4));
      r[0] = r[1] = r[2] = r[3] = v;
      return;
    }
    const float* base = (NCHK == 6 && chunk >= 3) ? p.src1 : p.src;
    const int coff = (((NCHK == 6 && chunk >= 3) ? chunk - 3 : chunk) * 8 + s_log) * 4;
    r[0] = *(const f32x4*)(base + ((size_t)(unsigned)(ya_c + xa_c) * 96 + coff));
    r[1] = *(const f32x4*)(base + ((size_t)(unsigned)(ya_c + xb_c) * 96 + coff));
    r[2] = *(const f32x4*)(base + ((size_t)(unsigned)(yb_c + xa_c) * 96 + coff));
    r[3] = *(const f32x4*)(base + ((size_t)(unsigned)(yb_c + xb_c) * 96 + coff));
  };
  auto write_x = [&](int it, int buf, const f32x4 (&r)[4]) {
    const int pos = it / NCHK;
    const int xi = pos >> 2, nu = pos & 3;
    // B^T rows: xi=0: +d0 -d2 | 1: +d1 +d2 | 2: -d1 +d2 | 3: +d1 -d3 ; the factor of an outside row / column is 0
    const int ya = xi == 0 ? yo[0] : yo[1], yb = xi == 3 ? yo[3] : yo[2];
    const int xa = nu == 0 ? xo[0] : xo[1], xb = nu == 3 ? xo[3] : xo[2];
    const float sya = ya < 0 ? 0.f : (xi == 2 ? -1.f : 1.f), syb = yb < 0 ? 0.f : ((xi == 0 || xi == 3) ? -1.f : 1.f);
    const float sxa = xa < 0 ? 0.f : (nu == 2 ? -1.f : 1.f), sxb = xb < 0 ? 0.f : ((nu == 0 || nu == 3) ? -1.f : 1.f);
    const f32x4 v = (r[0] * sxa + r[1] * sxb) * sya + (r[2] * sxa + r[3] * sxb) * syb;
    *(f32x4*)(Xb + buf * XB + srow * 128 + ps * 16) = v;
  };
  auto dma_w = [&](int it, int buf) {
    const float* wsrc = p.upk + (size_t)it * 192 * 32 + lane * 4;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int rbk = j * 8 + w;
      glds16(wsrc + rbk * 256, lds_w + buf * WB + rbk * 1024);
    }
  };

  f32x4 af[2][3], ag[2][3];            // position accumulators, two static sets (positions alternate between them)
  f32x4 of[2][2][3], og[2][2][3];      // output accumulators (a, b, tile)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    af[0][j] = ag[0][j] = af[1][j] = ag[1][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) of[a][b][j] = og[a][b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  // fold a finished position accumulator set: Y[a][b] += At[a][xi] * At[b][nu] * M,  At = [1 1 1 0; 0 1 -1 -1]
  auto fold = [&](f32x4 (&pf)[3], f32x4 (&pg)[3], int pos) {
    const int xi = pos >> 2, nu = pos & 3;
    const float ay0 = xi < 3 ? 1.f : 0.f, ay1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);
    const float ax0 = nu < 3 ? 1.f : 0.f, ax1 = nu == 0 ? 0.f : (nu == 1 ? 1.f : -1.f);
    const float c00 = ay0 * ax0, c01 = ay0 * ax1, c10 = ay1 * ax0, c11 = ay1 * ax1;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      of[0][0][j] += pf[j] * c00; of[0][1][j] += pf[j] * c01;
      of[1][0][j] += pf[j] * c10; of[1][1][j] += pf[j] * c11;
      og[0][0][j] += pg[j] * c00; og[0][1][j] += pg[j] * c01;
      og[1][0][j] += pg[j] * c10; og[1][1][j] += pg[j] * c11;
      pf[j] = pg[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
  };

  f32x4 r[4];
  load_x(0, r);
  dma_w(0, 0);
  write_x(0, 0, r);
  dma_wait_all();
  __syncthreads();
  for (int pp = 0; pp < 8; ++pp) {               // position pairs; body unrolled so the accumulator sets are static
#pragma unroll
    for (int k = 0; k < 2 * NCHK; ++k) {
      const int it = pp * 2 * NCHK + k;
      const int set = k / NCHK, chunk = k % NCHK;      // compile-time
      const int pos = pp * 2 + set;
      const int buf = k & 1;                     // == it & 1
      const char* Xt = Xb + buf * XB + tg * 2048;
      const char* Wf = Wb + buf * WB + (3 * chh) * 2048;
      const char* Wg = Wb + buf * WB + (6 + 3 * chh) * 2048;
      f32x4 wf[3], wg[3], xh;
      WINO_STAMP(0);
      // ---- k-half 0: 7 fragment reads, first 6 MFMAs
      xh = *(const f32x4*)(Xt + off0);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wf[j] = *(const f32x4*)(Wf + j * 2048 + off0);
        wg[j] = *(const f32x4*)(Wg + j * 2048 + off0);
      }
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        af[set][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][0], xh[0], af[set][j], 0, 0, 0);
        ag[set][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[j][0], xh[0], ag[set][j], 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
      WINO_STAMP(1);
      // ---- in the shadow of the MFMA pipe: raw granules + W DMA of iteration it+1 (a whole iteration to land),
      //      and the fold of the previous position (its MFMAs finished an iteration ago: no drain wait)
      if (it + 1 < NIT) {
        load_x(it + 1, r);
        dma_w(it + 1, buf ^ 1);
      }
      if (chunk == 0 && it > 0) fold(af[set ^ 1], ag[set ^ 1], pos - 1);
      __builtin_amdgcn_sched_barrier(0);
      WINO_STAMP(2);
      // ---- k-half 1 fragments are fetched BEFORE the remaining 18 MFMAs of k-half 0 (both waves of a SIMD run in
      //      lockstep, so an LDS read latency after the MFMAs would be fully exposed)
      f32x4 wf1[3], wg1[3], xh1;
      xh1 = *(const f32x4*)(Xt + off1);
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wf1[j] = *(const f32x4*)(Wf + j * 2048 + off1);
        wg1[j] = *(const f32x4*)(Wg + j * 2048 + off1);
      }
#pragma unroll
      for (int e = 1; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          af[set][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xh[e], af[set][j], 0, 0, 0);
          ag[set][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg[j][e], xh[e], ag[set][j], 0, 0, 0);
        }
#pragma unroll
      for (int e = 0; e < 4; ++e)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          af[set][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf1[j][e], xh1[e], af[set][j], 0, 0, 0);
          ag[set][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wg1[j][e], xh1[e], ag[set][j], 0, 0, 0);
        }
      WINO_STAMP(3);
      // ---- X tile of it+1 (hipcc waits for the 4 loads; the W DMA has had the same whole iteration to land)
      if (it + 1 < NIT) write_x(it + 1, buf ^ 1, r);
      WINO_STAMP(4);
      dma_wait_all();
      WINO_STAMP(5);
      __syncthreads();
      WINO_STAMP(6);
      __builtin_amdgcn_sched_barrier(0);
    }
  }
  fold(af[1], ag[1], 15);

  // ---- epilogue: lane holds 4 consecutive channels of tile (lane&15): 2x2 output pixels
  const int q = lane >> 4;
  const int t = tile_base + tg * 16 + (lane & 15);
  if (t < p.total_tiles) {
    int b, y0, x0;
    tile_origin(t, b, y0, x0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c0 = (3 * chh + j) * 16 + q * 4;
      const f32x4 bf = *(const f32x4*)(p.bias + c0);
      const f32x4 bg = *(const f32x4*)(p.bias + 96 + c0);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const float f = of[a][bb][j][e] + bf[e];
            const float g = og[a][bb][j][e] + bg[e];
            const float act = p.act == 0 ? elu_fast(f) : fmaxf(f, 0.f);
            o[e] = act * sigmoid_fast(g);
          }
          *(f32x4*)(p.dst + ((size_t)(b * p.h + y0 + a * p.d) * p.w + x0 + bb * p.d) * 96 + c0) = o;
        }
    }
  }
}

template <int NCHK>
static hipError_t launch_wino_t(const WinoParams& p, hipStream_t st) {
  constexpr int LDS = 2 * 64 * 128 + 2 * 192 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wino_kernel<NCHK>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = (p.total_tiles + 63) / 64;
  ProfScope ps_(st, PL_WINO_N192);
  hipLaunchKernelGGL(wino_kernel<NCHK>, dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_wino(const WinoParams& p, hipStream_t st) {
  return p.src1 ? launch_wino_t<6>(p, st) : launch_wino_t<3>(p, st);
}

}  // namespace se
