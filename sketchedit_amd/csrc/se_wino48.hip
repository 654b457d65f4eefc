// Winograd F(2x2,3x3) form of the 3x3 stride-1 gated convolution 48 -> 96 (conv3, conv14, xconv3, pmconv3,
// allconv14, wconv3, conv_mask_14 of /root/reference/models/networks/editline_g.py:44-47,87-90 and
// editline2_g.py:20-21: the 128x128 level of every encoder / decoder).  Same math, transform matrices, tile
// geometry (dilation through the polyphase sub-images) and pipeline as se_wino.hip -- read that header first.
// What differs:
//   * K per position is 48, i.e. 1.5 chunks of 32.  Two consecutive positions (P0, P1) share three chunks:
//         chunk A = [P0 ch 0-15 | P0 ch 16-31]   chunk B = [P0 ch 32-47 | P1 ch 0-15]   chunk C = [P1 ch 16-31 | P1 ch 32-47]
//     so no MFMA k-step is spent on padding; in chunk B the two k-halves accumulate into different positions
//     (P0 is folded between them).  24 iterations instead of 48.
//   * 96 packed rows in the MIXED order (8 features + their 8 gates per 16-row tile, gate exchange by a lane
//     swap in the epilogue), so a wave owns 3 row tiles x 32 tiles: 5 LDS fragment reads per 24 MFMAs (7 in
//     se_wino.hip) and the same 24 + 96 accumulator registers.
//   * One workgroup = 8 waves = 128 tiles; a lane stages TWO granules per iteration (same tile and slot in both
//     k-halves), from two offset sets: one for the even, one for the odd position of the pair.
//
// CIN = 24 (round 5: xconv3 / pmconv3 of netG, 24 -> 96 at the 128x128 level -- editline_g.py:63,75 -- two launches of 195 us
// on the direct gather-GEMM at 256x256 B=32).  K per position is 24 = six k-steps of 4: ONE chunk per position, k-half 0 =
// channels 0-15 (four k-steps), k-half 1 = channels 16-23 in TWO k-steps.  A k-step r of the 16x16x4 MFMA takes element r of
// all four 16-byte slots of a k-half, so the eight channels sit as elements 0, 1 of slots 4-7 (slot 4 + q: channels 16 + 2q,
// 17 + 2q -- the layout of se_rtilew.hip): the staging lane q gathers 8 bytes instead of 16 for that half, and k-steps 2, 3
// (elements 2, 3: never written, zero in the W image) are not issued.  16 iterations = 16 positions
// of 36 MFMAs per wave (6 k-steps x 3 row tiles x 2 tile groups), every iteration starts a position (C = 0) and the previous
// one is folded in front of it; the offset sets alternate by position parity.  Pipeline, rings, epilogue: unchanged.
#include "se_device.h"

#include <cstdlib>
#include <type_traits>

// Developer aid (-DSE_WINO_TRACE, tools/wino_trace.py): s_memtime stamps of block 0 / waves 0 and 4, kept in LDS.
#ifdef SE_WINO_TRACE
__device__ unsigned long long g_wino48_trace[96 * 8];
extern "C" int se_debug_wino48_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino48_trace), sizeof(unsigned long long) * 96 * 8);
}
#define W48_TRACE_LDS (3 * 128 * 128 + 4 * 96 * 128 + 8 * 512 * 4)
#define W48_STAMP(k)                                                    \
  do {                                                                  \
    if (blockIdx.x == 0 && (w & 3) == 0) {                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
      if (lane == 0) ((unsigned long long*)(smem + W48_TRACE_LDS))[((w >> 2) * 48 + it) * 8 + (k)] = t_; \
    }                                                                   \
  } while (0)
#define W48_STAMP_AT(slot, k) do { const int it = (slot); W48_STAMP(k); } while (0)
#define W48_TRACE_DUMP()                                                \
  do {                                                                  \
    if (blockIdx.x == 0 && (w & 3) == 0)                                \
      for (int i_ = lane; i_ < 48 * 8; i_ += 64)                        \
        g_wino48_trace[(w >> 2) * 48 * 8 + i_] = ((unsigned long long*)(smem + W48_TRACE_LDS))[(w >> 2) * 48 * 8 + i_]; \
  } while (0)
#else
#define W48_TRACE_LDS 0
#define W48_STAMP(k)
#define W48_STAMP_AT(slot, k)
#define W48_TRACE_DUMP()
#endif

namespace se {

// a compile-time loop index: `#pragma unroll` on a 24-body loop whose body tests it / 3 and it % 3 is only partly honoured
// (measured: 8 bodies in a run-time loop of 3, every "compile-time" condition a branch or a select)
template <int I, int N, typename F>
DEVFN void static_for48(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for48<I + 1, N>(f);
  }
}

template <int TILES, int CIN = 48>
__global__ __launch_bounds__(TILES * 4, 2) void wino48_kernel(const WinoParams p) {
  static_assert(CIN == 48 || CIN == 24, "48 -> 96 (position pairs over three chunks) or 24 -> 96 (one chunk per position)");
  constexpr int NTHR = TILES * 4, NWV = TILES / 16;      // one thread per staged granule; a wave per (row half, 32 tiles)
  constexpr int XB = TILES * 128, WB = 96 * 128;
  constexpr int NIT = CIN == 48 ? 24 : 16;               // 8 position pairs x 3 chunks | 16 positions
  constexpr unsigned PSB = CIN * 4;                      // bytes per source pixel
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 3 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chh = w & 1, tp = w >> 1;          // row half (3 MIXED tiles = 24 channels), tile pair (32 tiles)
  const int tile_base = (p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x) * TILES;
  const int tpi = p.th * p.tw;                 // tiles per image
  W48_STAMP_AT(24, 0);

  // tile -> (batch, first output pixel).  iy walks the tile grid; y0 = 2d*(iy/d) + iy%d
  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    b = (int)udiv_magic((unsigned)t, p.div_tpi_m, p.div_tpi_l);
    const int rem = t - b * tpi;
    const int iy = (int)udiv_magic((unsigned)rem, p.div_tw_m, p.div_tw_l), ix = rem - iy * p.tw;
    const int qy = (int)udiv_magic((unsigned)iy, p.div_d_m, p.div_d_l), qx = (int)udiv_magic((unsigned)ix, p.div_d_m, p.div_d_l);
    y0 = 2 * p.d * qy + (iy - qy * p.d);
    x0 = 2 * p.d * qx + (ix - qx * p.d);
  };

  // ---- staging role: tile row srow, granule s (4 channels) of each 16-channel k-half
  // 8 consecutive lanes = rows R and R+8 (swizzles 4 apart): their 4+4 granules of one k-half fill one 128-byte
  // bank row, so the LDS writes are conflict-free
  const int sg = tid & 3, srow = (tid >> 6) * 16 + ((tid >> 2) & 1) * 8 + ((tid >> 3) & 7);
  const int swz = (srow >> 1) & 7;
  char* xw0 = Xb + srow * 128 + ((sg ^ swz) << 4);             // k-half 0: logical slot sg
  char* xw1 = Xb + srow * 128 + (((4 + sg) ^ swz) << 4);       // k-half 1: logical slot 4 + sg
  // Source offsets of the 4x4 input tile, kept in LDS (read once per position):
  //   Ysrc[i][tid] = byte offset of pixel row y_i (+ this lane's granule), or 0x80000000 if outside / invalid tile
  //   Xsrc[i][tid] = byte offset of column x_i inside the row, or 0x80000000 if outside
  // A gather offset is their SATURATING sum (v_add_u32 clamp): >= 2^31 as soon as either part is outside, and the gather
  // goes through a buffer resource whose range check returns zeros there -- zero padding costs no arithmetic, and the
  // B^T factors that remain are signs, applied as compile-time +/- in the transform (se_wino.hip).
  int* Ysrc = (int*)(smem + 3 * XB + 4 * WB);     // 8 * NTHR ints
  int* Xsrc = Ysrc + 4 * NTHR;
  {
    const int t = tile_base + srow;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + (i - 1) * p.d, x = x0 + (i - 1) * p.d;
      Ysrc[i * NTHR + tid] = (t < p.total_tiles && (unsigned)y < (unsigned)p.h) ? (int)((unsigned)((b * p.h + y) * p.w) * PSB + (unsigned)sg * 16u) : (int)(0x80000000u + (unsigned)sg * 16u);
      Xsrc[i * NTHR + tid] = ((unsigned)x < (unsigned)p.w) ? x * (int)PSB : (int)0x80000000;
    }
  }
  const unsigned lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  unsigned o[2][4];     // [even / odd position of the pair]: byte offsets of the four source pixels (>= 2^31: outside)
  auto set_pos = [&](int set, int pos) {    // compile-time arguments after unrolling
    const int xi = pos >> 2, nu = pos & 3;
    // B^T rows: xi=0: +d0 -d2 | 1: +d1 +d2 | 2: -d1 +d2 | 3: +d1 -d3 ; a pixel outside the image reads as zero
    const unsigned ya = (unsigned)Ysrc[(xi == 0 ? 0 : 1) * NTHR + tid], yb = (unsigned)Ysrc[(xi == 3 ? 3 : 2) * NTHR + tid];
    const unsigned xa = (unsigned)Xsrc[(nu == 0 ? 0 : 1) * NTHR + tid], xb = (unsigned)Xsrc[(nu == 3 ? 3 : 2) * NTHR + tid];
    o[set][0] = __builtin_elementwise_add_sat(ya, xa); o[set][1] = __builtin_elementwise_add_sat(ya, xb);
    o[set][2] = __builtin_elementwise_add_sat(yb, xa); o[set][3] = __builtin_elementwise_add_sat(yb, xb);
  };
  // k-half h of iteration it -> (position set, 16-channel group): see the chunk table in the header
  // (CIN = 24: iteration = position, set = its parity, k-half h = channel group h)
  auto half_set = [](int it, int h) { return CIN == 48 ? ((it % 3) * 2 + h >= 3 ? 1 : 0) : (it & 1); };
  auto half_grp = [](int it, int h) { return CIN == 48 ? ((it % 3) * 2 + h) % 3 : h; };
  // position whose granules k-half h of iteration `it` carries: pair it / 3, even or odd member
  auto half_pos = [&](int it, int h) { return CIN == 48 ? 2 * (it / 3) + half_set(it, h) : it; };
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((unsigned)p.B * (unsigned)p.h * (unsigned)p.w * PSB), 0x00020000);
  // one raw granule (pixel i of the 4 sources) of k-half h of iteration `it`: one vector-memory instruction
  const unsigned adj8 = (unsigned)sg * 8u;      // CIN = 24, k-half 1: this lane's 8 bytes start at 64 + 8 sg, the offsets hold 16 sg
  auto load_x1 = [&](int it, f32x4 (&r)[2][4], int h, int i) {
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if (CIN == 24 && h == 1) {
      // (an "outside" offset stays outside after the subtraction: the row sentinel carries the lane's 16 sg like an in-image
      // row offset does, so o >= 2^31 + 16 sg, or the saturated 2^32 - 1 -- independent of the host's size guard)
      const u32x2 t = __builtin_amdgcn_raw_buffer_load_b64(rs0, (int)(o[half_set(it, h)][i] - adj8), 64, 0);
      r[h][i][0] = __uint_as_float(t[0]);
      r[h][i][1] = __uint_as_float(t[1]);
    } else {
      const u32x4 t = __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)o[half_set(it, h)][i], half_grp(it, h) * 64, 0);
      r[h][i] = __builtin_bit_cast(f32x4, t);
    }
  };
  float negone = -1.f;
  asm volatile("" : "+v"(negone));      // opaque -1: a subtraction stays one packed fma (a plain - becomes 4 v_sub)
  // the four granules of position (xi, nu) combined with their B^T signs (compile-time)
  auto signed_sum = [&](const f32x4 (&q)[4], int pos) -> f32x4 {
    const int xi = pos >> 2, nu = pos & 3;
    const bool nya = xi == 2, nyb = xi == 0 || xi == 3, nxa = nu == 2, nxb = nu == 0 || nu == 3;
    const bool ng[4] = {nya != nxa, nya != nxb, nyb != nxa, nyb != nxb};      // granule i enters with a minus sign
    f32x4 ps = {0.f, 0.f, 0.f, 0.f}, ns = {0.f, 0.f, 0.f, 0.f};
    bool hp = false, hn = false;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ng[i]) { ns = hn ? ns + q[i] : q[i]; hn = true; }
      else { ps = hp ? ps + q[i] : q[i]; hp = true; }
    }
    return !hn ? ps : (!hp ? ns * negone : ns * negone + ps);
  };
  auto write_x = [&](int it, int buf, const f32x4 (&r)[2][4]) {
    *(f32x4*)(xw0 + buf * XB) = signed_sum(r[0], half_pos(it, 0));
    const f32x4 s1 = signed_sum(r[1], half_pos(it, 1));
    if (CIN == 24) *(f32x2*)(xw1 + buf * XB) = (f32x2){s1[0], s1[1]};      // channels 16 + 2 sg, 17 + 2 sg: elements 0, 1 of slot 4 + sg
    else *(f32x4*)(xw1 + buf * XB) = s1;
  };
  // W tile: 12 row blocks of 8 rows; wave w stages blocks w, w + NWV, ... (8 waves: 2 calls, 4 waves: 3 calls per tile)
  auto dma_w = [&](int it, int buf, int j) {
    const int rbk = j * NWV + w;
    if (rbk < 12) glds16_s(p.upk + (size_t)it * 96 * 32 + rbk * 256, (unsigned)lane * 16u, lds_w + buf * WB + rbk * 1024);
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
  // fold the finished position: Y[a][b] += At[a][xi] * At[b][nu] * M,  At = [1 1 1 0; 0 1 -1 -1]; pos compile-time,
  // so only the non-zero terms exist and they are plain adds / subtracts
  float neg1 = -1.f;
  asm volatile("" : "+v"(neg1));      // opaque multiplier: keeps the subtraction a packed fma
  auto fold = [&](int pos) {
    const int xi = pos >> 2, nu = pos & 3;
    const int ay[2] = {xi < 3 ? 1 : 0, xi == 0 ? 0 : (xi == 1 ? 1 : -1)};
    const int ax[2] = {nu < 3 ? 1 : 0, nu == 0 ? 0 : (nu == 1 ? 1 : -1)};
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
      for (int q = 0; q < 2; ++q) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int b = 0; b < 2; ++b) {
            const int c = ay[a] * ax[b];
            if (c > 0) oy[a][b][j][q] += am[j][q];
            else if (c < 0) oy[a][b][j][q] = am[j][q] * neg1 + oy[a][b][j][q];     // v_pk_fma (a plain -= becomes 4 scalar v_sub)
          }
        // pin the sums: hipcc would otherwise sink every fold to the end of the unrolled kernel
        asm volatile("" : "+v"(oy[0][0][j][q]), "+v"(oy[0][1][j][q]), "+v"(oy[1][0][j][q]), "+v"(oy[1][1][j][q]));
      }
  };

  // ---- pipeline (se_wino.hip), with every wait explicit and conservative: the granule loads are ordinary loads that
  // hipcc tracks itself (an inline-asm load whose result register hipcc may copy before the data has landed is a
  // latent race), and one s_waitcnt vmcnt(0) per iteration -- right before the granules of the PREVIOUS iteration
  // are consumed -- also covers the W DMA of the previous iteration.  That DMA therefore targets a 4-slot ring, two
  // barriers ahead of its first reader:
  //   iteration it:  groups 0-3 | vmcnt(0); X(it+2) <- granules loaded in it-1 | groups 4-7: W DMA and granule loads
  //                  of it+3, k-half 0 fragments of it+1 | barrier
  //   X(it+2): slot (it+2)%3, last read in it-1, published by this barrier, first read (fragments) in it+1
  //   W(it+3): slot (it+3)%4, last read in it-1, complete after the vmcnt(0) of it+1, published by the barrier of
  //            it+1, first read (fragments) in it+2
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
  set_pos(0, 2);                       // even position of pair 1 (iteration 3 on) | CIN = 24: position 2 (iteration 2)
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

  W48_STAMP_AT(24, 1);
  // (position pairs x chunks | positions), fully unrolled through a compile-time index: everything below is compile-time
  static_for48<0, NIT>([&](auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value;
    constexpr int pp = CIN == 48 ? it / 3 : it, c = CIN == 48 ? it % 3 : 0;
    constexpr int b0 = it % 3, b1 = (it + 1) % 3, b2 = (it + 2) % 3;      // X ring
    constexpr int w0 = it % 4, w1 = (it + 1) % 4, w3 = (it + 3) % 4;      // W ring
    constexpr bool more1 = it + 1 < NIT, more2 = it + 2 < NIT, more3 = it + 3 < NIT;
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
    W48_STAMP(0);
    if (c == 0 && it > 0) {              // chunk A: the odd position of the previous pair is complete | CIN = 24: the previous position
      fold(CIN == 48 ? 2 * pp - 1 : it - 1);
      __builtin_amdgcn_sched_barrier(0);
    }
    group(wa, xa, 0, c == 0);
    group(wa, xa, 1, false);
    group(wa, xa, 2, false);
    group(wa, xa, 3, false);
    W48_STAMP(1);
    if (CIN == 48 && c == 1) {           // chunk B: the even position ends with k-half 0, the odd one starts
      fold(2 * pp);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (more2) {
      dma_wait_all();                    // granules and W DMA issued in groups 4-7 of the previous iteration
      write_x(it + 2, b2, r);
    }
    // next use of a position set: even set after the last chunk-B write, odd set after the last chunk-C write
    if (CIN == 48) {
      if (c == 2 && 2 * (pp + 2) < 16) set_pos(0, 2 * (pp + 2));
      if (c == 0 && 2 * (pp + 1) + 1 < 16) set_pos(1, 2 * (pp + 1) + 1);
    } else if (it + 3 < NIT) {
      set_pos((it + 3) & 1, it + 3);   // its previous holder (position it + 1) was loaded during iteration it - 2
    }
    __builtin_amdgcn_sched_barrier(0);
    // vector-memory instructions spread over the MFMA groups (a burst from all 8 waves fills the CU's queue and
    // stalls the waves, MFMAs included, in front of it)
    W48_STAMP(2);
    group(wb, xb, 0, CIN == 48 && c == 1);
    if (more3) { dma_w(it + 3, w3, 0); load_x1(it + 3, r, 0, 0); load_x1(it + 3, r, 0, 1); }
    if (more1) {                                          // k-half 0 fragments of it+1 (published slots)
      xa[0] = *(const f32x4*)(Xw + b1 * XB + off0);
      xa[1] = *(const f32x4*)(Xw + b1 * XB + 2048 + off0);
#pragma unroll
      for (int j = 0; j < 3; ++j) wa[j] = *(const f32x4*)(Ww + w1 * WB + j * 2048 + off0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (CIN == 24 && more3) { dma_w(it + 3, w3, 1); load_x1(it + 3, r, 0, 2); load_x1(it + 3, r, 0, 3); load_x1(it + 3, r, 1, 0); }
    group(wb, xb, 1, false);
    if (CIN == 48) {
      if (more3) { dma_w(it + 3, w3, 1); load_x1(it + 3, r, 0, 2); load_x1(it + 3, r, 0, 3); }
      __builtin_amdgcn_sched_barrier(0);
      group(wb, xb, 2, false);
      if (more3) { if (NWV < 8) dma_w(it + 3, w3, 2); load_x1(it + 3, r, 1, 0); load_x1(it + 3, r, 1, 1); }
      __builtin_amdgcn_sched_barrier(0);
      group(wb, xb, 3, false);
      if (more3) { load_x1(it + 3, r, 1, 2); load_x1(it + 3, r, 1, 3); }
    } else if (more3) {                                   // (k-steps 2, 3 of k-half 1 carry no channels: not issued)
      if (NWV < 8) dma_w(it + 3, w3, 2);
      load_x1(it + 3, r, 1, 1); load_x1(it + 3, r, 1, 2); load_x1(it + 3, r, 1, 3);
    }
    W48_STAMP(3);
    end_barrier();
    W48_STAMP(4);
  });
  W48_STAMP_AT(24, 2);
  fold(15);

  // ---- epilogue.  Lane (q = lane>>4, col = lane&15) holds rows 4q..4q+3 of every accumulator tile: features for
  // q < 2 (lanes 0-31), the matching gates in lane + 32.  Two v_permlane32_swap per quad hand each lane two complete
  // (feature, gate) pairs -- lanes 0-31 channels c0, c0+1, lanes 32-63 channels c0+2, c0+3 -- so every lane
  // evaluates two outputs (a ds_bpermute exchange per value cost a quarter of the kernel's time).
  const int q = lane >> 4;
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
          if (t < p.total_tiles)      // (dst: 48 gated channels = 192 bytes per pixel whatever CIN)
            *(float2*)((char*)p.dst + ((unsigned)((b * p.h + y0 + a * p.d) * p.w + x0 + bb * p.d) * 192u + (unsigned)c0 * 4u)) = ov;      // 32-bit offset: the launch guards the bytes
        }
    }
  }
  W48_STAMP_AT(24, 3);
  __syncthreads();
  W48_TRACE_DUMP();
}

// 64 tiles / 4 waves / 80 KB per workgroup (default): two workgroups share a CU, one's prologue, fold and epilogue run
// under the other's MFMAs.  SE_WINO48_TILES=128 selects the 8-wave, one-workgroup-per-CU shape of round 1.
template <int TILES, int CIN = 48>
static hipError_t launch_wino48_t(const WinoParams& p, hipStream_t st) {
  constexpr int LDS = 3 * TILES * 128 + 4 * 96 * 128 + 8 * TILES * 4 * 4 + (W48_TRACE_LDS ? 2 * 48 * 8 * 8 : 0);     // X ring + W ring 48 KB + source offsets
  {
    hipError_t e = ensure_max_lds((const void*)wino48_kernel<TILES, CIN>, LDS);
    if (e != hipSuccess) return e;
  }
  const int grid = (p.total_tiles + TILES - 1) / TILES;
  set_launch_grid(grid);
  ProfScope ps_(st, PL_WINO_N96);
  hipLaunchKernelGGL((wino48_kernel<TILES, CIN>), dim3(grid), dim3(TILES * 4), LDS, st, p);
  return hipGetLastError();
}
hipError_t launch_wino48(const WinoParams& p, hipStream_t st) {
  const bool big = W48_TRACE_LDS || opt(OPT_WINO48_TILES) == 128;
  return big ? launch_wino48_t<128>(p, st) : launch_wino48_t<64>(p, st);
}
// 24 -> 96 (src NHWC 24 channels; upk [16 positions][96 MIXED rows][32 k: channels 0-23, then zeros], pack_wino48 with cin 24)
hipError_t launch_wino48_c24(const WinoParams& p, hipStream_t st) { return launch_wino48_t<64, 24>(p, st); }

}  // namespace se
