// Winograd F(2x2,2x2) form of gen_deconv 48 -> 48 (24 gated): nearest x2 upsample + 3x3 conv at the full-resolution
// decoder step (/root/reference/models/networks/utils.py:35-51; conv15_upsample_conv, conv_mask_15_upsample_conv,
// allconv15_upsample_conv of editline_g.py:58,98 and editline2_g.py:33,41: 128x128 -> 256x256 at a 256x256 input).
// Math, transform matrices and class handling as se_wino_up.hip (four sub-pixel parity classes, 9 positions per 2x2 tile
// of class outputs, every coefficient 0 or +-1, class-adjacent dispatch); K pairing as se_wino48.hip: K per position is
// 48 = 1.5 chunks of 32, so two consecutive positions (P0, P1) share three chunks
//       chunk A = [P0 ch 0-15 | P0 ch 16-31]   chunk B = [P0 ch 32-47 | P1 ch 0-15]   chunk C = [P1 ch 16-31 | P1 ch 32-47]
// and the 9 positions are 4 pairs + (8, nothing): 14 iterations, the last k-half carries zero weights.
// 21.7 GFLOP executed per launch at 256x256 batch 32 (22.5 with the padded half) instead of the 38.7 of the sub-pixel
// gather / raw-tile form, 87.0 in reference-defined FLOPs.
// What differs from both: 48 packed rows = 3 MIXED row tiles, so there is no row split across waves -- one workgroup =
// 4 waves = 64 tiles, a wave owns all 3 row tiles x 16 tiles (12 + 48 accumulator registers); every thread still stages
// one tile row (two granules per iteration), i.e. the staging work per MFMA is twice that of se_wino_up.hip.  49.5 KB of
// LDS and 154 VGPRs: three workgroups per CU (the third hides the staging stalls of the other two: a wave of this kernel
// has an MFMA to issue only ~30 % of the time).
#include "se_device.h"

#include <cstdlib>

namespace se {

__global__ __launch_bounds__(256, 3) void winoup48_kernel(const WinoParams p) {
  constexpr int TILES = 64, NTHR = 256, NWV = 4;
  constexpr int XB = TILES * 128, WB = 48 * 128;
  constexpr int NIT = 14;              // 4 position pairs x 3 chunks + chunks A, B of the pair (8, -)
  constexpr int NPOS = 9;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 3 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tp = w;                            // tile group (16 tiles) of this wave
  int tgrp, cls;                               // tile group and output parity class of this workgroup (se_device.h)
  if (!class_tile((int)blockIdx.x, (p.total_tiles + TILES - 1) / TILES, p.xcd, tgrp, cls)) return;
  const int tile_base = tgrp * TILES;
  const int tpi = p.th * p.tw;                 // tiles per image
  const int py = cls >> 1, px = cls & 1;
  const float* upk = p.upk + (size_t)cls * NIT * 48 * 32;

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
  // Source offsets of the 3x3 input tile, kept in LDS (read once per position), one entry per TILE (the four lanes that
  // stage a tile's four granules read the same entry and add their granule offset): 1.5 KB instead of 6 KB, which is what
  // brings the workgroup under a third of the CU's LDS -- three workgroups per CU instead of two (round 3; 154 VGPRs).
  //   Ysrc[i][tile] = byte offset of source row yy0 - 1 + py + i, or -1 if outside / invalid tile
  //   Xsrc[i][tile] = byte offset of source column xx0 - 1 + px + i inside the row, or -1 if outside
  int* Ysrc = (int*)(smem + 3 * XB + 4 * WB);
  int* Xsrc = Ysrc + 3 * TILES;
  if (tid < TILES) {
    const int t = tile_base + tid;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int y = y0 - 1 + py + i, x = x0 - 1 + px + i;
      Ysrc[i * TILES + tid] = (t < p.total_tiles && (unsigned)y < (unsigned)p.h) ? (int)((unsigned)((b * p.h + y) * p.w) * 192u) : (int)0x80000000;
      Xsrc[i * TILES + tid] = ((unsigned)x < (unsigned)p.w) ? x * 192 : (int)0x80000000;
    }
  }
  __syncthreads();
  const unsigned sg16 = (unsigned)sg * 16u;
  const unsigned lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  // B^T rows: xi=0: +d0 -d1 | 1: +d1 | 2: -d1 +d2.  Source pixel i of a position = (row a|b, column a|b):
  // i=0 (a,a), 1 (a,b), 2 (b,a), 3 (b,b); the b row (column) does not exist for xi == 1 (nu == 1).  Position 9 (the
  // missing partner of position 8, zero weights) reads the pixels of position 8.
  auto need = [](int pos, int i) {
    const int pc = pos < NPOS ? pos : NPOS - 1;
    return !((pc / 3 == 1 && i >= 2) || (pc % 3 == 1 && (i & 1)));
  };
  unsigned o[2][4];     // [even / odd position of the pair]: byte offsets of the source pixels (+ this lane's granule); >= 2^31: outside
  auto set_pos = [&](int set, int pos) {    // compile-time arguments after unrolling
    const int pc = pos < NPOS ? pos : NPOS - 1;
    const int xi = pc / 3, nu = pc % 3;
    const unsigned ya = (unsigned)Ysrc[(xi == 0 ? 0 : 1) * TILES + srow] + sg16, yb = (unsigned)Ysrc[(xi == 2 ? 2 : 1) * TILES + srow] + sg16;
    const unsigned xa = (unsigned)Xsrc[(nu == 0 ? 0 : 1) * TILES + srow], xb = (unsigned)Xsrc[(nu == 2 ? 2 : 1) * TILES + srow];
    o[set][0] = __builtin_elementwise_add_sat(ya, xa); o[set][1] = __builtin_elementwise_add_sat(ya, xb);
    o[set][2] = __builtin_elementwise_add_sat(yb, xa); o[set][3] = __builtin_elementwise_add_sat(yb, xb);
  };
  // k-half h of iteration it -> (position set, 16-channel group, position): the chunk table in the header
  auto half_set = [](int it, int h) { return (it % 3) * 2 + h >= 3 ? 1 : 0; };
  auto half_grp = [](int it, int h) { return ((it % 3) * 2 + h) % 3; };
  auto half_pos = [&](int it, int h) { return 2 * (it / 3) + half_set(it, h); };
  // (components pinned to SGPRs: left alone hipcc keeps this resource in VGPRs here and wraps every gather in a
  // v_readfirstlane waterfall loop -- 309 of them)
  const unsigned long long src_a = (unsigned long long)p.src;
  const unsigned src_lo = __builtin_amdgcn_readfirstlane((unsigned)src_a), src_hi = __builtin_amdgcn_readfirstlane((unsigned)(src_a >> 32));
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)(((unsigned long long)src_hi << 32) | src_lo), 0,
      __builtin_amdgcn_readfirstlane((int)((unsigned)p.B * (unsigned)p.h * (unsigned)p.w * 192u)), 0x00020000);
  // one raw granule (source pixel i) of k-half h of iteration `it`: one vector-memory instruction
  auto load_x1 = [&](int it, f32x4 (&r)[2][4], int h, int i) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    if (need(half_pos(it, h), i))
      r[h][i] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs0, (int)o[half_set(it, h)][i], half_grp(it, h) * 64, 0));
  };
  float negone = -1.f;
  asm volatile("" : "+v"(negone));      // opaque -1: a subtraction stays one packed fma (a plain - becomes 4 v_sub)
  auto signed_sum = [&](const f32x4 (&q)[4], int pos) -> f32x4 {
    const int pc = pos < NPOS ? pos : NPOS - 1;
    const int xi = pc / 3, nu = pc % 3;
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
    *(f32x4*)(xw0 + buf * XB) = signed_sum(r[0], half_pos(it, 0));
    *(f32x4*)(xw1 + buf * XB) = signed_sum(r[1], half_pos(it, 1));
  };
  // W tile: 6 row blocks of 8 rows; wave w stages block w, and block 4 + w if w < 2
  auto dma_w = [&](int it, int buf, int j) {
    const int rbk = j * NWV + w;
    if (rbk < 6) glds16_s(upk + (size_t)it * 48 * 32 + rbk * 256, (unsigned)lane * 16u, lds_w + buf * WB + rbk * 1024);
  };

  f32x4 am[3];                         // position accumulators (row tile)
  f32x4 oy[2][2][3];                   // output accumulators (a, b, row tile)
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    am[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 b4 = *(const f32x4*)(p.bias + j * 16 + (lane >> 4) * 4);      // output accumulators start at the bias
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) oy[a][b][j] = b4;
  }
  // fold the finished position: Y[a][b] += At[a][xi] * At[b][nu] * M,  At = [1 1 0; 0 1 1]: plain adds of the terms
  // that exist (pos compile-time)
  auto fold = [&](int pos) {
    const int xi = pos / 3, nu = pos % 3;
    const int ay[2] = {xi < 2 ? 1 : 0, xi > 0 ? 1 : 0};
    const int ax[2] = {nu < 2 ? 1 : 0, nu > 0 ? 1 : 0};
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
          if (ay[a] * ax[b]) oy[a][b][j] += am[j];
      // pin the sums: hipcc would otherwise sink every fold to the end of the unrolled kernel
      asm volatile("" : "+v"(oy[0][0][j]), "+v"(oy[0][1][j]), "+v"(oy[1][0][j]), "+v"(oy[1][1][j]));
    }
  };

  // ---- pipeline: X tiles in a 3-slot, W tiles in a 4-slot LDS ring, one barrier and one conservative vmcnt(0) per
  // iteration (se_wino.hip / se_wino48.hip)
  auto end_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- prologue: X slots 0, 1; W slots 0, 1, 2; granules of iteration 2 in flight
  f32x4 r[2][4];
  set_pos(0, 0);
  set_pos(1, 1);
#pragma unroll
  for (int i0 = 0; i0 < 3; ++i0) { dma_w(i0, i0, 0); dma_w(i0, i0, 1); }     // W DMA first: overlaps the granule round trip
  {
    f32x4 r1[2][4];                    // iterations 0 and 1: both sets of loads in flight before the first transform
#pragma unroll
    for (int i = 0; i < 4; ++i) { load_x1(0, r, 0, i); load_x1(0, r, 1, i); }
#pragma unroll
    for (int i = 0; i < 4; ++i) { load_x1(1, r1, 0, i); load_x1(1, r1, 1, i); }
    write_x(0, 0, r);
    write_x(1, 1, r1);
  }
  set_pos(0, 2);                       // even position of pair 1 (iteration 3 on)
  dma_wait_all();
  __syncthreads();
#pragma unroll
  for (int i = 0; i < 4; ++i) { load_x1(2, r, 0, i); load_x1(2, r, 1, i); }

  const char* Xw = Xb + tp * 16 * 128;                   // this wave's 16 tile rows
  f32x4 wa[3], xa;                     // k-half 0 fragments of the current iteration (read one iteration ahead)
  xa = *(const f32x4*)(Xw + off0);
#pragma unroll
  for (int j = 0; j < 3; ++j) wa[j] = *(const f32x4*)(Wb + j * 2048 + off0);

#pragma unroll
  for (int pp = 0; pp < 5; ++pp)       // position pairs x chunks, fully unrolled: everything below is compile-time
#pragma unroll
  for (int c = 0; c < 3; ++c) {
    const int it = pp * 3 + c;
    if (it >= NIT) continue;
    const int b0 = it % 3, b1 = (it + 1) % 3, b2 = (it + 2) % 3;      // X ring
    const int w0 = it % 4, w1 = (it + 1) % 4, w3 = (it + 3) % 4;      // W ring
    const bool more1 = it + 1 < NIT, more2 = it + 2 < NIT, more3 = it + 3 < NIT;
    f32x4 wb[3], xb;
    xb = *(const f32x4*)(Xw + b0 * XB + off1);                     // k-half 1 fragments of this iteration
#pragma unroll
    for (int j = 0; j < 3; ++j) wb[j] = *(const f32x4*)(Wb + w0 * WB + j * 2048 + off1);
    __builtin_amdgcn_sched_barrier(0);
    // one group = 3 MFMAs: k-step e of the 3 accumulator tiles; `first`: C = 0 (first k-step of a position)
    auto group = [&](const f32x4 (&wf)[3], const f32x4& xf, int e, bool first) {
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        const f32x4 cin = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : am[j];
        am[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[j][e], xf[e], cin, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (c == 0 && it > 0) {              // chunk A: the odd position of the previous pair is complete
      fold(2 * pp - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    group(wa, xa, 0, c == 0);
    group(wa, xa, 1, false);
    group(wa, xa, 2, false);
    group(wa, xa, 3, false);
    if (c == 1) {                        // chunk B: the even position ends with k-half 0, the odd one starts
      fold(2 * pp);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more2) {
      dma_wait_all();                    // granules and W DMA issued in groups 4-7 of the previous iteration
      write_x(it + 2, b2, r);
    }
    // next use of a position set: even set after the last chunk-B write, odd set after the last chunk-C write
    if (c == 2 && 2 * (pp + 2) < NPOS) set_pos(0, 2 * (pp + 2));
    if (c == 0 && 2 * (pp + 1) + 1 <= NPOS) set_pos(1, 2 * (pp + 1) + 1);
    __builtin_amdgcn_sched_barrier(0);
    // vector-memory instructions spread over the MFMA groups (a burst from all waves fills the CU's queue and stalls the
    // waves, MFMAs included, in front of it)
    group(wb, xb, 0, c == 1);
    if (more3) { dma_w(it + 3, w3, 0); load_x1(it + 3, r, 0, 0); load_x1(it + 3, r, 0, 1); }
    if (more1) {                                          // k-half 0 fragments of it+1 (published slots)
      xa = *(const f32x4*)(Xw + b1 * XB + off0);
#pragma unroll
      for (int j = 0; j < 3; ++j) wa[j] = *(const f32x4*)(Wb + w1 * WB + j * 2048 + off0);
    }
    __builtin_amdgcn_sched_barrier(0);
    group(wb, xb, 1, false);
    if (more3) { dma_w(it + 3, w3, 1); load_x1(it + 3, r, 0, 2); load_x1(it + 3, r, 0, 3); }
    __builtin_amdgcn_sched_barrier(0);
    group(wb, xb, 2, false);
    if (more3) { load_x1(it + 3, r, 1, 0); load_x1(it + 3, r, 1, 1); }
    __builtin_amdgcn_sched_barrier(0);
    group(wb, xb, 3, false);
    if (more3) { load_x1(it + 3, r, 1, 2); load_x1(it + 3, r, 1, 3); }
    end_barrier();
  }
  // (position 8 was folded in chunk B of the last pair; the k-half after it carried zero weights)

  // ---- epilogue (se_wino48.hip): v_permlane32_swap gate exchange, two outputs per lane; class (py,px) output
  // pixel of tile output (a, bb) = (2(yy0+a)+py, 2(xx0+bb)+px) of the upsampled grid
  const int q = lane >> 4;
  const int OW = 2 * p.w;
  {
    const int t = tile_base + tp * 16 + (lane & 15);
    int b = 0, y0 = 0, x0 = 0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c0 = j * 8 + (q & 1) * 4 + (q >> 1) * 2;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const f32x4 v = oy[a][bb][j];      // (bias already inside)
          const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
          const float f0 = __uint_as_float(s02[0]), g0 = __uint_as_float(s02[1]);
          const float f1 = __uint_as_float(s13[0]), g1 = __uint_as_float(s13[1]);
          float2 ov;
          ov.x = act_fast(f0, eluw) * sigmoid_fast(g0);
          ov.y = act_fast(f1, eluw) * sigmoid_fast(g1);
          if (t < p.total_tiles)
            *(float2*)((char*)p.dst + ((unsigned)((b * 2 * p.h + 2 * (y0 + a) + py) * OW + 2 * (x0 + bb) + px) * 96u + (unsigned)c0 * 4u)) = ov;      // 32-bit offset: the launch guards the output bytes
        }
    }
  }
}

hipError_t launch_winoup48(const WinoParams& p, hipStream_t st) {
  constexpr int TILES = 64;
  constexpr int LDS = 3 * TILES * 128 + 4 * 48 * 128 + 6 * TILES * 4;     // X ring 24 KB + W ring 24 KB + source offsets 1.5 KB: three per CU
  {
    hipError_t e = ensure_max_lds((const void*)winoup48_kernel, LDS);
    if (e != hipSuccess) return e;
  }
  const int grid = class_tile_grid((p.total_tiles + TILES - 1) / TILES);
  set_launch_grid(grid);
  ProfScope ps_(st, PL_GCONV_N48);
  hipLaunchKernelGGL(winoup48_kernel, dim3(grid), dim3(TILES * 4), LDS, st, p);
  return hipGetLastError();
}

}  // namespace se
