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
// and one half of the channels (3 feature tiles + their 3 gate tiles).  Loop over (position, 32-k chunk):
//   * X tile [64 tiles][32 k]: each lane builds ONE granule = B^T x B at this position from 4 input granules
//     (global loads, issued one iteration ahead) and writes it to LDS -- the input transform is fused, no
//     transformed tensor ever exists in HBM;
//   * W tile [192][32 k] of the host-transformed weights G g G^T by LDS-DMA;
//   * MFMA into the position accumulator; after the last chunk of a position the accumulator is folded into the
//     four output accumulators with the A^T coefficients (inverse transform in registers);
// epilogue: bias, ELU/ReLU * sigmoid gate, NHWC store.  Zero padding = zero-filled input granules.
#include "se_device.h"

namespace se {

__global__ __launch_bounds__(512) void wino_kernel(const WinoParams p) {
  constexpr int XB = 64 * 128, WB = 192 * 128;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chh = w & 1, tg = w >> 1;          // channel half, tile group
  const int tile_base = blockIdx.x * 64;
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
  {
    const int t = tile_base + srow;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + (i - 1) * p.d, x = x0 + (i - 1) * p.d;
      yo[i] = (t < p.total_tiles && (unsigned)y < (unsigned)p.h) ? (b * p.h + y) * p.w : -1;
      xo[i] = ((unsigned)x < (unsigned)p.w) ? x : -1;
    }
  }
  const unsigned lds_x = lds_addr_of(Xb), lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  auto load_x = [&](int it, f32x4 (&r)[4]) {
    const int pos = it / 3, chunk = it - pos * 3;          // uniform
    const int xi = pos >> 2, nu = pos & 3;
    const int ya = xi == 0 ? yo[0] : yo[1], yb = xi == 3 ? yo[3] : yo[2];
    const int xa = nu == 0 ? xo[0] : xo[1], xb = nu == 3 ? xo[3] : xo[2];
    const int coff = (chunk * 8 + s_log) * 4;
    const float* g00 = (ya >= 0 && xa >= 0) ? p.src + ((size_t)(unsigned)(ya + xa) * 96 + coff) : p.zeros;
    const float* g01 = (ya >= 0 && xb >= 0) ? p.src + ((size_t)(unsigned)(ya + xb) * 96 + coff) : p.zeros;
    const float* g10 = (yb >= 0 && xa >= 0) ? p.src + ((size_t)(unsigned)(yb + xa) * 96 + coff) : p.zeros;
    const float* g11 = (yb >= 0 && xb >= 0) ? p.src + ((size_t)(unsigned)(yb + xb) * 96 + coff) : p.zeros;
    r[0] = *(const f32x4*)g00;
    r[1] = *(const f32x4*)g01;
    r[2] = *(const f32x4*)g10;
    r[3] = *(const f32x4*)g11;
  };
  auto write_x = [&](int it, int buf, const f32x4 (&r)[4]) {
    const int pos = it / 3;
    const int xi = pos >> 2, nu = pos & 3;
    // B^T rows: xi=0: +d0 -d2 | 1: +d1 +d2 | 2: -d1 +d2 | 3: +d1 -d3
    const float sya = xi == 2 ? -1.f : 1.f, syb = (xi == 0 || xi == 3) ? -1.f : 1.f;
    const float sxa = nu == 2 ? -1.f : 1.f, sxb = (nu == 0 || nu == 3) ? -1.f : 1.f;
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

  f32x4 mf[3][1], mg[3][1];            // position accumulators: feature tiles, gate tiles
  f32x4 of[2][2][3], og[2][2][3];      // output accumulators (a, b, tile)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    mf[j][0] = mg[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) of[a][b][j] = og[a][b][j] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  f32x4 r[4];
  load_x(0, r);
  dma_w(0, 0);
  write_x(0, 0, r);
  dma_wait_all();
  __syncthreads();
  constexpr int NIT = 48;              // 16 positions x 3 chunks
  for (int it = 0; it < NIT; ++it) {
    const int buf = it & 1;
    if (it + 1 < NIT) {
      load_x(it + 1, r);
      dma_w(it + 1, buf ^ 1);
    }
    const char* Xt = Xb + buf * XB + tg * 2048;
    mfma_chunk<3, 1>(mf, Wb + buf * WB + (3 * chh) * 2048, Xt, off0, off1);
    mfma_chunk<3, 1>(mg, Wb + buf * WB + (6 + 3 * chh) * 2048, Xt, off0, off1);
    const int pos = it / 3;
    if (it - pos * 3 == 2) {
      // inverse transform: Y[a][b] += At[a][xi] * At[b][nu] * M,  At = [1 1 1 0; 0 1 -1 -1]
      const int xi = pos >> 2, nu = pos & 3;
      const float ay0 = xi < 3 ? 1.f : 0.f, ay1 = xi == 0 ? 0.f : (xi == 1 ? 1.f : -1.f);
      const float ax0 = nu < 3 ? 1.f : 0.f, ax1 = nu == 0 ? 0.f : (nu == 1 ? 1.f : -1.f);
      const float c00 = ay0 * ax0, c01 = ay0 * ax1, c10 = ay1 * ax0, c11 = ay1 * ax1;
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        of[0][0][j] += mf[j][0] * c00; of[0][1][j] += mf[j][0] * c01;
        of[1][0][j] += mf[j][0] * c10; of[1][1][j] += mf[j][0] * c11;
        og[0][0][j] += mg[j][0] * c00; og[0][1][j] += mg[j][0] * c01;
        og[1][0][j] += mg[j][0] * c10; og[1][1][j] += mg[j][0] * c11;
        mf[j][0] = mg[j][0] = (f32x4){0.f, 0.f, 0.f, 0.f};
      }
    }
    if (it + 1 < NIT) write_x(it + 1, buf ^ 1, r);
    dma_wait_all();
    __syncthreads();
  }

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

hipError_t launch_wino(const WinoParams& p, hipStream_t st) {
  constexpr int LDS = 2 * 64 * 128 + 2 * 192 * 128;
  static bool attr_set = false;
  if (!attr_set) {
    hipError_t e = hipFuncSetAttribute((const void*)wino_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, LDS);
    if (e != hipSuccess) return e;
    attr_set = true;
  }
  const int grid = (p.total_tiles + 63) / 64;
  ProfScope ps_(st, PL_WINO_N192);
  hipLaunchKernelGGL(wino_kernel, dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

}  // namespace se
