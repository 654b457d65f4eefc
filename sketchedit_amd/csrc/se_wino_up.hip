// Winograd F(2x2,2x2) form of gen_deconv 96 -> 96 (48 gated): nearest x2 upsample + 3x3 conv
// (/root/reference/models/networks/utils.py:35-51; conv13_upsample_conv, conv_mask_13_upsample_conv,
// allconv13_upsample_conv of editline_g.py:56,96 and editline2_g.py:31,39: the 64x64 -> 128x128 decoder step).
// In the sub-pixel form (se_gconv.hip, pack_layer) each output parity class (py,px) is a 2x2 conv on the source grid
// with pre-summed weights g:  out[2yy+py][2xx+px] = sum_{a,b} g[a][b] x[yy+a-1+py][xx+b-1+px].  Per class and per
// 2x2 tile of class outputs (3x3 source pixels d):
//   Y = A^T [ (G g G^T) .* (B^T d B) ] A,   B^T = [1 -1 0; 0 1 0; 0 -1 1]   G = [1 0; 1 1; 0 1]   A^T = [1 1 0; 0 1 1]
// 9 transform positions x K=96 instead of 4 taps x K=96 per 4 outputs (16/9 fewer multiply-adds than the sub-pixel
// form, 4x fewer than the reference-defined layer); every coefficient is 0 or +-1.  The four class workgroups of a tile
// group run side by side on one XCD (class_tile, se_device.h) so that they share the source tile through its L2.
// Structure, pipeline and epilogue as se_wino48.hip (96 MIXED rows, a wave = 3 row tiles x 32 tiles, 128 tiles per
// workgroup, two staged granules per lane), loop as se_wino.hip (27 iterations = 9 positions x 3 chunks, fold at the
// first chunk of the next position).  Source pixels whose B^T factor is structurally zero are not loaded.
#include "se_device.h"

#include <cstdlib>

namespace se {

template <int TILES>
__global__ __launch_bounds__(TILES * 4, 2) void winoup_kernel(const WinoParams p) {
  constexpr int NTHR = TILES * 4, NWV = TILES / 16;
  constexpr int XB = TILES * 128, WB = 96 * 128;
  constexpr int NIT = 27;              // 9 positions x 3 chunks
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 3 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chh = w & 1, tp = w >> 1;          // row half (3 MIXED tiles = 24 channels), tile pair (32 tiles)
  int tgrp, cls;                               // tile group and output parity class of this workgroup (se_device.h)
  if (!class_tile((int)blockIdx.x, (p.total_tiles + TILES - 1) / TILES, p.xcd, tgrp, cls)) return;
  const int tile_base = tgrp * TILES;
  const int tpi = p.th * p.tw;                 // tiles per image
  const int py = cls >> 1, px = cls & 1;
  const float* upk = p.upk + (size_t)cls * NIT * 96 * 32;

  // tile -> (batch, first class-grid (= source-grid) pixel of its 2x2 outputs)
  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    b = (int)udiv_magic((unsigned)t, p.div_tpi_m, p.div_tpi_l);
    const int rem = t - b * tpi;
    const int iy = (int)udiv_magic((unsigned)rem, p.div_tw_m, p.div_tw_l), ix = rem - iy * p.tw;
    y0 = 2 * iy;
    x0 = 2 * ix;
  };

  // ---- staging role: tile row srow, granule sg (4 channels) of each 16-channel k-half (se_wino48.hip)
  const int sg = tid & 3, srow = (tid >> 6) * 16 + ((tid >> 2) & 1) * 8 + ((tid >> 3) & 7);
  const int swz = (srow >> 1) & 7;
  char* xw0 = Xb + srow * 128 + ((sg ^ swz) << 4);             // k-half 0: logical slot sg
  char* xw1 = Xb + srow * 128 + (((4 + sg) ^ swz) << 4);       // k-half 1: logical slot 4 + sg
  // Source offsets of the 3x3 input tile, kept in LDS (read once per position):
  //   Ysrc[i][tid] = byte offset of source row yy0 - 1 + py + i (+ this lane's granule), or 0x80000000 if outside / invalid tile
  //   Xsrc[i][tid] = byte offset of source column xx0 - 1 + px + i inside the row, or 0x80000000 if outside
  // A gather offset is their saturating sum; the gather goes through a buffer resource (>= 2^31: zeros), the B^T factors
  // that remain are compile-time signs (se_wino.hip, se_wino48.hip).
  int* Ysrc = (int*)(smem + 3 * XB + 4 * WB);
  int* Xsrc = Ysrc + 3 * NTHR;
  {
    const int t = tile_base + srow;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int y = y0 - 1 + py + i, x = x0 - 1 + px + i;
      Ysrc[i * NTHR + tid] = (t < p.total_tiles && (unsigned)y < (unsigned)p.h) ? (int)((unsigned)((b * p.h + y) * p.w) * 384u + (unsigned)sg * 16u) : (int)0x80000000;
      Xsrc[i * NTHR + tid] = ((unsigned)x < (unsigned)p.w) ? x * 384 : (int)0x80000000;
    }
  }
  const unsigned lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  // B^T rows: xi=0: +d0 -d1 | 1: +d1 | 2: -d1 +d2.  Source pixel i of a position = (row a|b, column a|b):
  // i=0 (a,a), 1 (a,b), 2 (b,a), 3 (b,b); the b row (column) does not exist for xi == 1 (nu == 1).
  auto need = [](int pos, int i) { return !((pos / 3 == 1 && i >= 2) || (pos % 3 == 1 && (i & 1))); };
  unsigned o[4];        // byte offsets of the source pixels of the current position (+ this lane's granule); >= 2^31: outside
  auto set_pos = [&](int pos) {             // compile-time argument after unrolling
    const int xi = pos / 3, nu = pos % 3;
    const unsigned ya = (unsigned)Ysrc[(xi == 0 ? 0 : 1) * NTHR + tid], yb = (unsigned)Ysrc[(xi == 2 ? 2 : 1) * NTHR + tid];
    const unsigned xa = (unsigned)Xsrc[(nu == 0 ? 0 : 1) * NTHR + tid], xb = (unsigned)Xsrc[(nu == 2 ? 2 : 1) * NTHR + tid];
    o[0] = __builtin_elementwise_add_sat(ya, xa); o[1] = __builtin_elementwise_add_sat(ya, xb);
    o[2] = __builtin_elementwise_add_sat(yb, xa); o[3] = __builtin_elementwise_add_sat(yb, xb);
  };
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((unsigned)p.B * (unsigned)p.h * (unsigned)p.w * 384u), 0x00020000);
  // one raw granule (source pixel i) of k-half h of iteration `it` (position it/3, channels (it%3)*32 + h*16 ...)
  auto load_x1 = [&](int it, f32x4 (&r)[2][4], int h, int i) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if (need(it / 3, i)) r[h][i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs0, (int)o[i], ((it % 3) * 32 + h * 16) * 4, 0));
  };
  float negone = -1.f;
  asm volatile("" : "+v"(negone));      // opaque -1: a subtraction stays one packed fma (a plain - becomes 4 v_sub)
  // the granules of a position combined with their B^T signs: row a: xi == 2 ? - : +, row b: xi == 0 ? - : +, same in x
  auto signed_sum = [&](const f32x4 (&q)[4], int pos) -> f32x4 {
    const int xi = pos / 3, nu = pos % 3;
    const bool nya = xi == 2, nyb = xi == 0, nxa = nu == 2, nxb = nu == 0;
    const bool ng[4] = {nya != nxa, nya != nxb, nyb != nxa, nyb != nxb};
    f32x4 ps = {0.f, 0.f, 0.f, 0.f}, ns = {0.f, 0.f, 0.f, 0.f};
    bool hp = false, hn = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (!need(pos, i)) continue;
      if (ng[i]) { ns = hn ? ns + q[i] : q[i]; hn = true; }
      else { ps = hp ? ps + q[i] : q[i]; hp = true; }
    }
    return !hn ? ps : (!hp ? ns * negone : ns * negone + ps);
  };
  auto write_x = [&](int it, int buf, const f32x4 (&r)[2][4]) {
    *(f32x4*)(xw0 + buf * XB) = signed_sum(r[0], it / 3);
    *(f32x4*)(xw1 + buf * XB) = signed_sum(r[1], it / 3);
  };
  // W tile: 12 row blocks of 8 rows; wave w stages block w, and block 8 + w if w < 4
  auto dma_w = [&](int it, int buf, int j) {
    const int rbk = j * NWV + w;
    if (rbk < 12) glds16_s(upk + (size_t)it * 96 * 32 + rbk * 256, (unsigned)lane * 16u, lds_w + buf * WB + rbk * 1024);
  };

  f32x4 am[3][2];                      // position accumulators (row tile, tile group)
  f32x4 oy[2][2][3][2];                // output accumulators (a, b, row tile, tile group)
#pragma unroll
  for (int j = 0; j < 3; ++j)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
      am[j][q] = (f32x4){0.f, 0.f, 0.f, 0.f};
      // the output accumulators start at the bias (MIXED packed rows): one add less per value in the epilogue
      const f32x4 b4 = *(const f32x4*)(p.bias + (3 * chh + j) * 16 + (lane >> 4) * 4);
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) oy[a][b][j][q] = b4;
    }
  // fold the finished position: Y[a][b] += At[a][xi] * At[b][nu] * M,  At = [1 1 0; 0 1 1]: plain adds of the terms
  // that exist (pos compile-time)
  auto fold = [&](int pos) {
    const int xi = pos / 3, nu = pos % 3;
    const int ay[2] = {xi < 2 ? 1 : 0, xi > 0 ? 1 : 0};
    const int ax[2] = {nu < 2 ? 1 : 0, nu > 0 ? 1 : 0};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b)
            if (ay[a] * ax[b]) oy[a][b][j][q] += am[j][q];
        // pin the sums: hipcc would otherwise sink every fold to the end of the unrolled kernel
        asm volatile("" : "+v"(oy[0][0][j][q]), "+v"(oy[0][1][j][q]), "+v"(oy[1][0][j][q]), "+v"(oy[1][1][j][q]));
      }
  };

  // ---- pipeline: X tiles in a 3-slot, W tiles in a 4-slot LDS ring, one barrier and one conservative vmcnt(0) per
  // iteration (se_wino.hip / se_wino48.hip)
  auto end_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: X slots 0, 1; W slots 0, 1, 2; granules of iteration 2 in flight (all of position 0)
  f32x4 r[2][4];
  set_pos(0);
#pragma unroll
  for (int i0 = 0; i0 < 3; ++i0) { dma_w(i0, i0, 0); dma_w(i0, i0, 1); if (NWV < 8) dma_w(i0, i0, 2); }     // W DMA first: overlaps the granule round trip
  {
    f32x4 r1[2][4];                    // iterations 0 and 1: both sets of loads in flight before the first transform
#pragma unroll
    for (int i = 0; i < 4; ++i) { load_x1(0, r, 0, i); load_x1(0, r, 1, i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { load_x1(1, r1, 0, i); load_x1(1, r1, 1, i); }
    write_x(0, 0, r);
    write_x(1, 1, r1);
  }
  dma_wait_all();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { load_x1(2, r, 0, i); load_x1(2, r, 1, i); }

  const char* Xw = Xb + tp * 32 * 128;                   // this wave's 32 tile rows
  const char* Ww = Wb + (3 * chh) * 2048;                // this wave's 3 row tiles
  f32x4 wa[3], xa[2];                  // k-half 0 fragments of the current iteration (read one iteration ahead)
  xa[0] = *(const f32x4*)(Xw + off0);
  xa[1] = *(const f32x4*)(Xw + 2048 + off0);
#pragma unroll
  for (int j = 0; j < 3; ++j) wa[j] = *(const f32x4*)(Ww + j * 2048 + off0);

#pragma unroll
  for (int pos = 0; pos < 9; ++pos)    // positions x chunks, fully unrolled: everything below is compile-time
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int it = pos * 3 + c;
    const int b0 = it % 3, b1 = (it + 1) % 3, b2 = (it + 2) % 3;      // X ring
    const int w0 = it % 4, w1 = (it + 1) % 4, w3 = (it + 3) % 4;      // W ring
    const bool more1 = it + 1 < NIT, more2 = it + 2 < NIT, more3 = it + 3 < NIT;
    f32x4 wb[3], xb[2];
    xb[0] = *(const f32x4*)(Xw + b0 * XB + off1);                  // k-half 1 fragments of this iteration
    xb[1] = *(const f32x4*)(Xw + b0 * XB + 2048 + off1);
#pragma unroll
    for (int j = 0; j < 3; ++j) wb[j] = *(const f32x4*)(Ww + w0 * WB + j * 2048 + off1);
    __builtin_amdgcn_sched_barrier(0);
    // one group = 6 MFMAs: k-step e of the 3 x 2 accumulator tiles; `first`: C = 0 (first k-step of a position)
    auto group = [&](const f32x4 (&wf)[3], const f32x4 (&xf)[2], int e, bool first) {
#pragma unroll
      for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int q = 0; q < 2; ++q) {
          const f32x4 cin = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : am[j][q];
          am[j][q] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xf[q][e], cin, 0, 0, 0);
        }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (c == 0 && it > 0) {              // the previous position is complete
      fold(pos - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    group(wa, xa, 0, c == 0);
    group(wa, xa, 1, false);
    group(wa, xa, 2, false);
    group(wa, xa, 3, false);
    if (more2) {
      dma_wait_all();                    // granules and W DMA issued in groups 4-7 of the previous iteration
      write_x(it + 2, b2, r);
    }
    if ((it + 3) % 3 == 0 && more3) set_pos((it + 3) / 3);      // position of the granules fetched next
    __builtin_amdgcn_sched_barrier(0);
    // vector-memory instructions spread over the MFMA groups (a burst from all 8 waves fills the CU's queue and
    // stalls the waves, MFMAs included, in front of it)
    group(wb, xb, 0, false);
    if (more3) { dma_w(it + 3, w3, 0); load_x1(it + 3, r, 0, 0); load_x1(it + 3, r, 0, 1); }
    if (more1) {                                          // k-half 0 fragments of it+1 (published slots)
      xa[0] = *(const f32x4*)(Xw + b1 * XB + off0);
      xa[1] = *(const f32x4*)(Xw + b1 * XB + 2048 + off0);
#pragma unroll
      for (int j = 0; j < 3; ++j) wa[j] = *(const f32x4*)(Ww + w1 * WB + j * 2048 + off0);
    }
    __builtin_amdgcn_sched_barrier(0);
    group(wb, xb, 1, false);
    if (more3) { dma_w(it + 3, w3, 1); load_x1(it + 3, r, 0, 2); load_x1(it + 3, r, 0, 3); }
    __builtin_amdgcn_sched_barrier(0);
    group(wb, xb, 2, false);
    if (more3) { if (NWV < 8) dma_w(it + 3, w3, 2); load_x1(it + 3, r, 1, 0); load_x1(it + 3, r, 1, 1); }
    __builtin_amdgcn_sched_barrier(0);
    group(wb, xb, 3, false);
    if (more3) { load_x1(it + 3, r, 1, 2); load_x1(it + 3, r, 1, 3); }
    end_barrier();
  }
  fold(8);

  // ---- epilogue (se_wino48.hip): v_permlane32_swap gate exchange, two outputs per lane; class (py,px) output
  // pixel of tile output (a, bb) = (2(yy0+a)+py, 2(xx0+bb)+px) of the upsampled grid
  const int q = lane >> 4;
  const int OW = 2 * p.w;
#pragma unroll
  for (int tq = 0; tq < 2; ++tq) {
    const int t = tile_base + tp * 32 + tq * 16 + (lane & 15);
    int b = 0, y0 = 0, x0 = 0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c0 = (3 * chh + j) * 8 + (q & 1) * 4 + (q >> 1) * 2;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const f32x4 v = oy[a][bb][j][tq];      // (bias already inside)
          const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
          const float f0 = __uint_as_float(s02[0]), g0 = __uint_as_float(s02[1]);
          const float f1 = __uint_as_float(s13[0]), g1 = __uint_as_float(s13[1]);
          float2 ov;
          ov.x = act_fast(f0, eluw) * sigmoid_fast(g0);
          ov.y = act_fast(f1, eluw) * sigmoid_fast(g1);
          if (t < p.total_tiles)
            *(float2*)((char*)p.dst + ((unsigned)((b * 2 * p.h + 2 * (y0 + a) + py) * OW + 2 * (x0 + bb) + px) * 192u + (unsigned)c0 * 4u)) = ov;      // 32-bit offset (output bytes < 2^32: twice the guarded input bytes)
        }
    }
  }
}

// 64 tiles / 4 waves / 78 KB per workgroup (default): two workgroups per CU (se_wino48.hip); SE_WINOUP_TILES=128: 8 waves
template <int TILES>
static hipError_t launch_winoup_t(const WinoParams& p, hipStream_t st) {
  constexpr int LDS = 3 * TILES * 128 + 4 * 96 * 128 + 6 * TILES * 4 * 4;     // X ring + W ring 48 KB + source offsets
  {
    hipError_t e = ensure_max_lds((const void*)winoup_kernel<TILES>, LDS);
    if (e != hipSuccess) return e;
  }
  const int grid = class_tile_grid((p.total_tiles + TILES - 1) / TILES);
  set_launch_grid(grid);
  ProfScope ps_(st, PL_WINO_UP96);
  hipLaunchKernelGGL(winoup_kernel<TILES>, dim3(grid), dim3(TILES * 4), LDS, st, p);
  return hipGetLastError();
}
hipError_t launch_winoup(const WinoParams& p, hipStream_t st) {
  const bool big = opt(OPT_WINOUP_TILES) == 128;
  return big ? launch_winoup_t<128>(p, st) : launch_winoup_t<64>(p, st);
}

}  // namespace se
