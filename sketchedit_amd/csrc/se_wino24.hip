// Hybrid Winograd F(2,3) x F(4,3) form of the 3x3 stride-1 gated convolution 96 -> 192 (the dominant layer shape:
// conv5-10, wconv5-10, xconv6-10, pmconv5-10, conv12, ... of /root/reference/models/networks/editline_g.py:48-99 and
// editline2_g.py:22-37; the layer itself is gen_conv, /root/reference/models/networks/utils.py:21-33), any dilation d
// with h % 2d == 0 and w % 4d == 0.  Round 4.
//
//   Y = Ay^T [ (Gy g Gx^T) .* (By^T x Bx) ] Ax      per 2x4 output tile / 4x6 input tile, summed over input channels
//   rows:    F(2,3)  By^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]           Ay^T = [1 1 1 0; 0 1 -1 -1]
//   columns: F(4,3)  Bx^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]
//                    Ax^T = [1 1 1 1 1 0; 0 1 -1 2 -2 0; 0 1 1 4 4 0; 0 1 -1 8 -8 1]
//                    Gx   = [1/4 0 0; -1/6 -1/6 -1/6; -1/6 1/6 -1/6; 1/24 1/12 1/6; 1/24 -1/12 1/6; 0 0 1]
//
// 24 transform positions x K=96 per 8 outputs: 3 multiply-adds per output and input channel instead of 4 in
// F(2x2,3x3) (se_wino.hip) and 9 in the direct form.  F(4x4,3x3) (2.25) does not fit: 16 output accumulators per (tile,
// channel) -- see DESIGN.md 7b; with 8 the wave tile is 48 packed rows x 16 tiles = 96 accumulator registers, the same
// as the F(2x2,3x3) kernels.  The F(4,3) constants are not dyadic: measured against the oracle the layer differs by
// ~4e-6 at |y| ~ 5 (F(2x2,3x3): ~2e-6); the parity budget is 1e-3 end to end.
//
// One workgroup = 8 waves = 32 tiles (256 outputs) x all 192 packed rows in the MIXED order (tile t = features 8t..8t+7
// and their gates: the gate is a lane swap away in the epilogue).  Wave w: rows of MIXED tiles 3 (w & 3) .. + 2,
// tiles 16 (w >> 2) .. + 15.
// Loop: 72 iterations (position, 32-channel chunk) of 24 MFMAs per wave, in the order
//     stage (xi, h)  ->  chunk c  ->  j            xi = 0..3 row position, h = column-position half, j = 0..2:
//     column position nu = {0, 1, 2}[j] (h = 0) / {5, 3, 4}[j] (h = 1)
// so that the three positions of a sub-stage (xi, h, c) share ONE separable input transform:
//   * a lane owns (tile, 8-byte piece = 2 channels) of a sub-stage: 10 raw loads (2 rows x 5 columns: the F(2,3) row
//     combination is one add, the F(4,3) column transform of the 5 sums costs 6 more packed operations for 3 results:
//     single = 4 c0 - 5 c2 + c4;  pair = (c4 - 4 c2) +- (c3 - 4 c1)  [h = 0]  /  (c3 - c1) +- 2 (c2 - c0)  [h = 1]),
//     11 packed VALU instructions and 3.3 loads per thread and iteration -- se_wino.hip: 8 + 4 per (twice as long)
//     iteration; per MFMA the same, per OUTPUT 25 % fewer MFMAs;
//   * three position accumulators (one per j) are live; after the last chunk the single position is folded alone and the
//     pair as sum / difference (Ax^T columns 1,2 = (1,1,1,1), (1,-1,1,-1); columns 3,4 = (1,2,4,8), (1,-2,4,-8)):
//     76 instead of 108 f32x4 operations per row tile.
// Zero padding, gather offsets (saturating sums of row / column offsets kept in LDS, buffer range check), the W ring
// (LDS-DMA), the fragment prefetch across the barrier and the spreading of vector-memory instructions over the MFMA groups
// are those of se_wino.hip -- read that header first.  W ring: 5 slots; X ring: 2 sub-stage slots of 3 tiles x 4 KB; the
// wait schedule is described at the pipeline below.
#include "se_device.h"

#include <cstdlib>
#include <type_traits>

// Developer aid (-DSE_WINO_TRACE, tools/wino_trace.py): s_memtime stamps of block 0 / waves 0 and 4 (the two waves of SIMD 0),
// kept in LDS; slots 0..71 = iterations, slot 72 = kernel phases (start, loop entry, loop exit, end).
#ifdef SE_WINO_TRACE
__device__ unsigned long long g_wino24_trace[2 * 80 * 8];
extern "C" int se_debug_wino24_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino24_trace), sizeof(unsigned long long) * 2 * 80 * 8);
}
#define W24_TRACE_LDS (2 * 3 * 32 * 128 + 5 * 192 * 128 + 4 * 512 * 4 + 6 * 32 * 4)
#define W24_STAMP_AT(slot, k)                                           \
  do {                                                                  \
    if (NCHK == 3 && blockIdx.x == 0 && (w & 3) == 0) {                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
      if (lane == 0) ((unsigned*)(smem + W24_TRACE_LDS))[((w >> 2) * 80 + (slot)) * 8 + (k)] = (unsigned)t_; \
    }                                                                   \
  } while (0)
#define W24_TRACE_DUMP()                                                \
  do {                                                                  \
    __syncthreads();                                                    \
    if (NCHK == 3 && blockIdx.x == 0 && (w & 3) == 0)                   \
      for (int i_ = lane; i_ < 80 * 8; i_ += 64)                        \
        g_wino24_trace[(w >> 2) * 80 * 8 + i_] = ((unsigned*)(smem + W24_TRACE_LDS))[(w >> 2) * 80 * 8 + i_]; \
  } while (0)
#else
#define W24_TRACE_LDS 0
#define W24_STAMP_AT(slot, k)
#define W24_TRACE_DUMP()
#endif

namespace se {

typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int I, int N, typename F>
DEVFN void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

// NCHK = 32-channel chunks per position: 3 for one 96-channel source, 6 for the two-source layer (allconv11:
// cat([x_hallu, pm]), editline_g.py:211): chunks 3-5 are gathered from the second tensor
// CIN48 (round 4, xconv5 of netG: 48 -> 192): a 48-channel source (192 bytes per pixel), NCHK = 2 chunks per position of
// which the second carries 16 channels: its iterations issue k-half 0 only (12 MFMAs), its k-half 1 columns of the X tile
// hold whatever follows the pixel in memory (finite: the next pixel, or the buffer range check's zeros) and are never read.
template <int NCHK, bool CIN48 = false>
__global__ __launch_bounds__(512, 2) void wino24_kernel(const WinoParams p) {
  constexpr unsigned PSB = CIN48 ? 192u : 384u;      // bytes per source pixel
  constexpr int TILES = 32;
  constexpr int XB = TILES * 128;      // one X tile: 32 tiles x 32 k
  constexpr int XS = 3 * XB;           // one sub-stage: the X tiles of its three positions
  constexpr int WB = 192 * 128;
  constexpr int NSUB = 8 * NCHK, NIT = 3 * NSUB;      // sub-stages (xi, h, chunk); iterations
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XS;
  int* Ysrc = (int*)(smem + 2 * XS + 5 * WB);      // [4 rows][512 threads]
  int* Xsrc = Ysrc + 4 * 512;                      // [6 columns][32 tiles]

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rg = w & 3, tg = w >> 2;           // row group (3 MIXED tiles = 24 gated channels), tile group (16 tiles)
  const int tile_base = (p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x) * TILES;
  const int tpi = p.th * p.tw;                 // tiles per image
  W24_STAMP_AT(72, 0);

  // tile -> (batch, first output pixel).  (iy, ix) walks the tile grid of the d x d polyphase sub-images
  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    b = (int)udiv_magic((unsigned)t, p.div_tpi_m, p.div_tpi_l);
    const int rem = t - b * tpi;
    const int iy = (int)udiv_magic((unsigned)rem, p.div_tw_m, p.div_tw_l), ix = rem - iy * p.tw;
    const int qy = (int)udiv_magic((unsigned)iy, p.div_d_m, p.div_d_l), qx = (int)udiv_magic((unsigned)ix, p.div_d_m, p.div_d_l);
    y0 = 2 * p.d * qy + (iy - qy * p.d);
    x0 = 4 * p.d * qx + (ix - qx * p.d);
  };

  // ---- staging role: tile row srow, 8-byte piece p16 (2 channels) of the 32-channel chunk
  const int srow = tid >> 4, p16 = tid & 15;
  char* xw = Xb + srow * 128 + (((p16 >> 1) ^ ((srow >> 1) & 7)) << 4) + (p16 & 1) * 8;
  {
    const int t = tile_base + srow;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + (i - 1) * p.d;
      const bool yok = t < p.total_tiles && (unsigned)y < (unsigned)p.h;
      Ysrc[i * 512 + tid] = yok ? (int)((unsigned)((b * p.h + y) * p.w) * PSB + (unsigned)p16 * 8u) : (int)0x80000000;
    }
    if (p16 < 6) {
      const int x = x0 + (p16 - 1) * p.d;
      Xsrc[p16 * 32 + srow] = (unsigned)x < (unsigned)p.w ? x * (int)PSB : (int)0x80000000;
    }
  }
  __syncthreads();
  const unsigned lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  // gather offsets of the current load stage (xi, h): 2 rows x 5 columns; >= 2^31: outside the image (reads zeros)
  unsigned o[2][5];
  auto set_offs = [&](int stage) {      // compile-time argument after unrolling
    const int xi = stage >> 1, h = stage & 1;
    // By^T rows: xi=0: +r0 -r2 | 1: +r1 +r2 | 2: -r1 +r2 | 3: +r1 -r3
    const unsigned ya = (unsigned)Ysrc[(xi == 0 ? 0 : 1) * 512 + tid], yb = (unsigned)Ysrc[(xi == 3 ? 3 : 2) * 512 + tid];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const unsigned xc = (unsigned)Xsrc[(c + h) * 32 + srow];
      o[0][c] = __builtin_elementwise_add_sat(ya, xc);
      o[1][c] = __builtin_elementwise_add_sat(yb, xc);
    }
  };
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((unsigned)p.B * (unsigned)p.h * (unsigned)p.w * PSB), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(NCHK == 6 ? p.src1 : p.src), 0, (int)((unsigned)p.B * (unsigned)p.h * (unsigned)p.w * 384u), 0x00020000);
  f32x2 r[10];          // raw pieces of one task: [row][column]
  auto load_x1 = [&](int chunk, int i) {       // piece i = row * 5 + column: one vector-memory instruction
    const u32x2 t = chunk >= 3 ? __builtin_amdgcn_raw_buffer_load_b64(rs1, (int)o[i / 5][i % 5], (chunk - 3) * 128, 0)
                               : __builtin_amdgcn_raw_buffer_load_b64(rs0, (int)o[i / 5][i % 5], chunk * 128, 0);
    r[i] = __builtin_bit_cast(f32x2, t);
  };
  // opaque constants: the transform stays packed fmas (a plain a - b on a vector becomes scalar v_sub per element)
  // (shared with the folds of the inverse transform below)
  float km1 = -1.f, k2 = 2.f, km2 = -2.f, k4 = 4.f, km4 = -4.f, km5 = -5.f, k8 = 8.f, km8 = -8.f;
  asm volatile("" : "+v"(km1), "+v"(k2), "+v"(km2), "+v"(k4), "+v"(km4), "+v"(km5), "+v"(k8), "+v"(km8));
  // transform of one task of stage (xi, h) and its three X tiles (slot = sub-stage % 3)
  auto transform = [&](int stage, int slot) {
    const int xi = stage >> 1, h = stage & 1;
    f32x2 c[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      if (xi == 1) c[j] = r[j] + r[5 + j];
      else if (xi == 2) c[j] = r[j] * km1 + r[5 + j];
      else c[j] = r[5 + j] * km1 + r[j];
    }
    const f32x2 single = c[0] * k4 + (c[2] * km5 + c[4]);
    f32x2 plo, phi;
    if (h == 0) {
      const f32x2 pp = c[2] * km4 + c[4], qq = c[1] * km4 + c[3];
      plo = pp + qq;
      phi = qq * km1 + pp;
    } else {
      const f32x2 rr = c[1] * km1 + c[3], tt = c[0] * km1 + c[2];
      plo = tt * k2 + rr;
      phi = tt * km2 + rr;
    }
    char* dst = xw + slot * XS;
    *(f32x2*)(dst) = single;
    *(f32x2*)(dst + XB) = plo;
    *(f32x2*)(dst + 2 * XB) = phi;
  };
  auto dma_w = [&](int it, int buf, int j) {      // piece j (0..2) of this wave's share of the W tile
    const int rbk = j * 8 + w;
    glds16_s(p.upk + (size_t)it * 192 * 32 + rbk * 256, (unsigned)lane * 16u, lds_w + buf * WB + rbk * 1024);
  };

  f32x4 am[3][3];                      // position accumulators [j][row tile]
  f32x4 oy[2][4][3];                   // output accumulators [a][b][row tile]
#pragma unroll
  for (int t = 0; t < 3; ++t) {        // the output accumulators start at the bias (MIXED packed rows)
    const f32x4 b4 = *(const f32x4*)(p.bias + (3 * rg + t) * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 4; ++b) oy[a][b][t] = b4;
#pragma unroll
    for (int j = 0; j < 3; ++j) am[j][t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
  const float fm1 = km1, f2 = k2, fm2 = km2, f4 = k4, fm4 = km4, f8 = k8, fm8 = km8;
  // Ay^T column xi: a = 0: (1 1 1 0), a = 1: (0 1 -1 -1)
  auto ay_of = [](int a, int xi) { return a == 0 ? (xi < 3 ? 1 : 0) : (xi == 0 ? 0 : (xi == 1 ? 1 : -1)); };
  // the single position of a stage: column position 0 (h = 0) feeds output column 0, 5 (h = 1) feeds column 3, factor 1
  auto fold_single = [&](int stage) {
    const int xi = stage >> 1, h = stage & 1, b = h ? 3 : 0;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int c = ay_of(a, xi);
        if (c > 0) oy[a][b][t] += am[0][t];
        else if (c < 0) oy[a][b][t] = am[0][t] * fm1 + oy[a][b][t];
      }
      asm volatile("" : "+v"(oy[0][b][t]), "+v"(oy[1][b][t]));
    }
  };
  // the pair: columns (1, 2) of Ax^T are (1 1 1 1), (1 -1 1 -1); columns (3, 4) are (1 2 4 8), (1 -2 4 -8)
  auto fold_pair = [&](int stage) {
    const int xi = stage >> 1, h = stage & 1;
#pragma unroll
    for (int t = 0; t < 3; ++t) {
      const f32x4 S = am[1][t] + am[2][t], D = am[2][t] * fm1 + am[1][t];
#pragma unroll
      for (int a = 0; a < 2; ++a) {
        const int c = ay_of(a, xi);
        if (c == 0) continue;
        if (h == 0) {
          if (c > 0) { oy[a][0][t] += S; oy[a][1][t] += D; oy[a][2][t] += S; oy[a][3][t] += D; }
          else {
            oy[a][0][t] = S * fm1 + oy[a][0][t]; oy[a][1][t] = D * fm1 + oy[a][1][t];
            oy[a][2][t] = S * fm1 + oy[a][2][t]; oy[a][3][t] = D * fm1 + oy[a][3][t];
          }
        } else {
          if (c > 0) { oy[a][0][t] += S; oy[a][1][t] = D * f2 + oy[a][1][t]; oy[a][2][t] = S * f4 + oy[a][2][t]; oy[a][3][t] = D * f8 + oy[a][3][t]; }
          else {
            oy[a][0][t] = S * fm1 + oy[a][0][t]; oy[a][1][t] = D * fm2 + oy[a][1][t];
            oy[a][2][t] = S * fm4 + oy[a][2][t]; oy[a][3][t] = D * fm8 + oy[a][3][t];
          }
        }
      }
      // pin the sums here: in unrolled code hipcc would otherwise sink every fold to the end of the kernel
      asm volatile("" : "+v"(oy[0][0][t]), "+v"(oy[0][1][t]), "+v"(oy[0][2][t]), "+v"(oy[0][3][t]),
                        "+v"(oy[1][0][t]), "+v"(oy[1][1][t]), "+v"(oy[1][2][t]), "+v"(oy[1][3][t]));
    }
  };

  auto end_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };

  // ---- pipeline.  One workgroup barrier per iteration; W tiles in a 5-slot ring, X sub-stages in a 2-slot ring.  The
  // vector-memory instructions of a sub-stage s (iterations 3s .. 3s+2; F / S = first / second half of an iteration) are
  // placed so that there is ONE full wait per sub-stage and it never meets an instruction younger than ~one iteration:
  //   3s   : mid: vmcnt(0), transform of task s+1 -> X slot (s+1)%2 | S: raw pieces 0-3 of task s+2, W(3s+4)
  //   3s+1 : F: raw pieces 4-6                                      | S: raw pieces 7-9, W(3s+5)
  //   3s+2 : F: W(3s+6) | mid: vmcnt(K): W(3s+4) has landed (K = the vector-memory instructions issued after it)
  //   W(it+4) -> slot (it+4)%5 = the slot of it-1, free since the barrier of it-1; complete at a mid-iteration wait of
  //   it+2 at the latest, published by that iteration's barrier, first read (k-half 0 fragments) in the second half of it+3
  //   X(s+1): slot last read in sub-stage s-1; written at mid 3s, published by the barrier of 3s, first read in 3s+2
  // (the first version waited vmcnt(0) in EVERY iteration, half an iteration after the youngest instruction: the cycle
  // trace showed both waves of a SIMD idle ~150 of 2200 cycles per iteration in that wait)
  // The raw loads are ordinary loads that hipcc tracks itself (se_wino.hip); the W DMA is hidden from it, so its own wait
  // in front of the transform is a vmcnt(0) as well.
  // What bounds the kernel (round 4, same-box ablations with tools/ab_variants.sh, 33 launches = 4.65 ms): the MFMA stream
  // is 67 % of a launch, but with the MFMAs REMOVED (folds cut, dead-code eliminated) the skeleton -- W DMA, raw gathers,
  // transform, barriers, epilogue -- still takes 3.50 ms: the vector-memory path is the longer pole and the two overlap only
  // partly (a wave in front of a full vector-memory queue issues no MFMAs either).  Removing one ingredient at a time:
  // barriers (2 of 3) -1.6 %, activation arithmetic of the epilogue -1.0 %, two thirds of the fold arithmetic -1.7 %,
  // transform arithmetic + X writes -7 %, all W DMA -4 %, all raw gathers -9 %.  Per OUTPUT the kernel moves what
  // se_wino.hip moves through the vector-memory path (W: 6.75 instead of 4.5 KB -- 32-tile workgroups; raw: 3.75 instead of
  // 6 KB) for 25 % fewer MFMAs, so it sits at the crossover of the two bounds.  Tried and not faster: all ten raw pieces of a
  // task in the 3/4 iteration after the transform (more lead time: +1.7 % -- it is the queue, not the latency), `nt` gathers
  // (+4 %: raw lines ARE re-read from L2), sc0 / sc1 gathers (no change); 16-byte gathers with the five columns of a task
  // split over lanes l / l + 32 (role A: c0, c2, c4; role B: c1, c3; six gathers per wave instead of ten, the halves of the
  // column transform exchanged with v_permlane32_swap: parity green, 256 VGPRs, +390 VALU instructions per wave for the
  // exchange and the per-lane signs: +1.6 %).
  // ---- prologue: W slots 0..3; X sub-stage 0 transformed; raw pieces of sub-stage 1 in flight in r
  set_offs(0);
#pragma unroll
  for (int i0 = 0; i0 < 4; ++i0) {     // the W DMA first: its latency overlaps the raw round trip
    dma_w(i0, i0, 0);
    dma_w(i0, i0, 1);
    dma_w(i0, i0, 2);
  }
#pragma unroll
  for (int i = 0; i < 10; ++i) load_x1(0, i);
  transform(0, 0);
#pragma unroll
  for (int i = 0; i < 10; ++i) load_x1(1, i);
  __builtin_amdgcn_sched_barrier(0);
  asm volatile("s_waitcnt vmcnt(10)" ::: "memory");      // the W tiles (issued before the 10 loads still in flight)
  __syncthreads();

  const char* Xw = Xb + tg * 2048;                       // this wave's 16 tile rows
  const char* Ww = Wb + (3 * rg) * 2048;                 // this wave's 3 row tiles
  f32x4 wa[3], xa;                     // k-half 0 fragments of the current iteration (read one iteration ahead)
  xa = *(const f32x4*)(Xw + off0);
#pragma unroll
  for (int t = 0; t < 3; ++t) wa[t] = *(const f32x4*)(Ww + t * 2048 + off0);

  // (Round 5: a static s_setprio 1 for one half of the waves -- the second-dispatched half, MI355X_MICROARCH.md "two waves per
  // SIMD" item 4, or the first -- measured on the same box: wino_n192 4.774-4.792 / 4.777-4.785 / 4.770-4.785 ms per step
  // without / waves 4-7 / waves 0-3: no effect; the two waves of a SIMD here are symmetric partners, not a compute / load pair.)
  W24_STAMP_AT(72, 1);
  // stages x (chunk, j), fully unrolled through a compile-time index (72 bodies are beyond what `#pragma unroll` accepts):
  // everything below is compile-time
  static_for<0, NIT>([&](auto it_c) __attribute__((always_inline)) {
    constexpr int it = decltype(it_c)::value;
    constexpr int st = it / (3 * NCHK), q = it % (3 * NCHK), s = it / 3, j = q % 3, chunk = q / 3;
    constexpr int w0 = it % 5, w1 = (it + 1) % 5, w4 = (it + 4) % 5;             // W ring
    constexpr bool more1 = it + 1 < NIT, more4 = it + 4 < NIT;
    constexpr int ls = s + 2;          // sub-stage whose raw pieces are fetched during this one
    constexpr bool ld = ls < NSUB;
    constexpr bool khalf1 = !(CIN48 && chunk == NCHK - 1);               // (48 channels: the second chunk has k-half 0 only)
    f32x4 wb[3], xb;
    if (khalf1) {
      xb = *(const f32x4*)(Xw + (s % 2) * XS + j * XB + off1);          // k-half 1 fragments of this iteration
#pragma unroll
      for (int t = 0; t < 3; ++t) wb[t] = *(const f32x4*)(Ww + w0 * WB + t * 2048 + off1);
    }
    __builtin_amdgcn_sched_barrier(0);
    W24_STAMP_AT(it, 0);
    // one group = 3 MFMAs: k-step e of the three row tiles; `first`: C = 0 (first k-step of a position)
    auto group = [&](const f32x4 (&wf)[3], const f32x4& xf, int e, bool first) {
#pragma unroll
      for (int t = 0; t < 3; ++t) {
        const f32x4 cin = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : am[j][t];
        am[j][t] = __builtin_amdgcn_mfma_f32_16x16x4f32(wf[t][e], xf[e], cin, 0, 0, 0);
      }
      __builtin_amdgcn_sched_barrier(0);
    };
    if (q == 3 * NCHK - 2) { fold_single(st); __builtin_amdgcn_sched_barrier(0); }   // am[0] was completed in the iteration before
    if (q == 0 && st > 0) { fold_pair(st - 1); __builtin_amdgcn_sched_barrier(0); }  // am[1], am[2] of the previous stage
    group(wa, xa, 0, chunk == 0);
    if (j == 1 && ld) load_x1(ls % NCHK, 4);
    if (j == 2 && more4) dma_w(it + 4, w4, 0);
    __builtin_amdgcn_sched_barrier(0);
    group(wa, xa, 1, false);
    if (j == 1 && ld) load_x1(ls % NCHK, 5);
    if (j == 2 && more4) dma_w(it + 4, w4, 1);
    __builtin_amdgcn_sched_barrier(0);
    group(wa, xa, 2, false);
    if (j == 1 && ld) load_x1(ls % NCHK, 6);
    if (j == 2 && more4) dma_w(it + 4, w4, 2);
    __builtin_amdgcn_sched_barrier(0);
    group(wa, xa, 3, false);
    W24_STAMP_AT(it, 1);
    if (j == 0) {
      dma_wait_all();                    // everything issued so far: the youngest is the W DMA of the first half of it-1
      if (s + 1 < NSUB) transform((s + 1) / NCHK, (s + 1) % 2);
      if (ld && ls % NCHK == 0) set_offs(ls / NCHK);
    }
    if (j == 2 && it + 2 < NIT) {        // W(it+2) was issued in the second half of it-2, in front of K younger instructions
      constexpr int K = 7 * (ld ? 1 : 0) + 3 * ((it - 1 + 4 < NIT ? 1 : 0) + (more4 ? 1 : 0));
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(K) : "memory");
    }
    __builtin_amdgcn_sched_barrier(0);
    W24_STAMP_AT(it, 2);
    // vector-memory instructions spread over the MFMA groups (se_wino.hip, point 2)
    if (khalf1) group(wb, xb, 0, false);
    if (j == 0 && ld) load_x1(ls % NCHK, 0);
    if (j == 1 && ld) load_x1(ls % NCHK, 7);
    if (j != 2 && more4) dma_w(it + 4, w4, 0);
    if (more1) {                                          // k-half 0 fragments of it+1 (published slots)
      xa = *(const f32x4*)(Xw + (((it + 1) / 3) % 2) * XS + ((it + 1) % 3) * XB + off0);
#pragma unroll
      for (int t = 0; t < 3; ++t) wa[t] = *(const f32x4*)(Ww + w1 * WB + t * 2048 + off0);
    }
    __builtin_amdgcn_sched_barrier(0);
    if (khalf1) group(wb, xb, 1, false);
    if (j == 0 && ld) load_x1(ls % NCHK, 1);
    if (j == 1 && ld) load_x1(ls % NCHK, 8);
    if (j != 2 && more4) dma_w(it + 4, w4, 1);
    __builtin_amdgcn_sched_barrier(0);
    if (khalf1) group(wb, xb, 2, false);
    if (j == 0 && ld) load_x1(ls % NCHK, 2);
    if (j == 1 && ld) load_x1(ls % NCHK, 9);
    if (j != 2 && more4) dma_w(it + 4, w4, 2);
    __builtin_amdgcn_sched_barrier(0);
    if (khalf1) group(wb, xb, 3, false);
    if (j == 0 && ld) load_x1(ls % NCHK, 3);
    __builtin_amdgcn_sched_barrier(0);
    W24_STAMP_AT(it, 3);
    end_barrier();
    W24_STAMP_AT(it, 4);
  });
  W24_STAMP_AT(72, 2);
  fold_pair(7);

  // ---- epilogue (se_wino48.hip).  Lane (q = lane>>4, col = lane&15) holds rows 4q..4q+3 of every accumulator tile:
  // features for q < 2, the matching gates in lane + 32.  Two v_permlane32_swap per quad hand each lane two complete
  // (feature, gate) pairs: lanes 0-31 channels c0, c0+1, lanes 32-63 channels c0+2, c0+3.
  const int q4 = lane >> 4;
  const int t = tile_base + tg * 16 + (lane & 15);
  int b = 0, y0 = 0, x0 = 0;
  tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
  const unsigned dst0 = (unsigned)((b * p.h + y0) * p.w + x0) * 384u;
  const unsigned dinc_y = (unsigned)(p.d * p.w) * 384u, dinc_x = (unsigned)p.d * 384u;
#pragma unroll
  for (int tt = 0; tt < 3; ++tt) {
    const int c0 = (3 * rg + tt) * 8 + (q4 & 1) * 4 + (q4 >> 1) * 2;
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int bb = 0; bb < 4; ++bb) {
        const f32x4 v = oy[a][bb][tt];      // (layer bias already inside)
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
        float f0 = __uint_as_float(s02[0]), g0 = __uint_as_float(s02[1]);
        float f1 = __uint_as_float(s13[0]), g1 = __uint_as_float(s13[1]);
        if (p.vbias) {      // folded vector source (launch_vecbias): a bias that depends on the pixel's border configuration
          const int y = y0 + a * p.d, x = x0 + bb * p.d;
          const int cfg = 3 * (y == 0 ? 0 : (y == p.h - 1 ? 2 : 1)) + (x == 0 ? 0 : (x == p.w - 1 ? 2 : 1));
          const float* tb = (const float*)((const char*)p.vbias + (unsigned)(((b * 9 + cfg) * 192 + c0) * 4));
          f0 += tb[0]; f1 += tb[1];
          g0 += tb[96]; g1 += tb[97];
        }
        float2 ov;
        ov.x = act_fast(f0, eluw) * sigmoid_fast(g0);
        ov.y = act_fast(f1, eluw) * sigmoid_fast(g1);
        if (t < p.total_tiles)
          *(float2*)((char*)p.dst + (dst0 + (unsigned)a * dinc_y + (unsigned)bb * dinc_x + (unsigned)c0 * 4u)) = ov;      // 32-bit offset: the launch guards the bytes
      }
  }
  W24_STAMP_AT(72, 3);
  W24_TRACE_DUMP();
}

template <int NCHK, bool CIN48 = false>
static hipError_t launch_wino24_t(const WinoParams& p, hipStream_t st) {
  constexpr int LDS = 2 * 3 * 32 * 128 + 5 * 192 * 128 + 4 * 512 * 4 + 6 * 32 * 4 + (W24_TRACE_LDS ? 2 * 80 * 8 * 4 : 0);     // X ring 24 KB + W ring 120 KB + source offsets 8.75 KB
  {
    hipError_t e = ensure_max_lds((const void*)wino24_kernel<NCHK, CIN48>, LDS);
    if (e != hipSuccess) return e;
  }
  const int grid = (p.total_tiles + 31) / 32;
  set_launch_grid(grid);
  ProfScope ps_(st, PL_WINO_N192);
  hipLaunchKernelGGL((wino24_kernel<NCHK, CIN48>), dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_wino24(const WinoParams& p, hipStream_t st) {
  return p.src1 ? launch_wino24_t<6>(p, st) : launch_wino24_t<3>(p, st);
}
hipError_t launch_wino24_c48(const WinoParams& p, hipStream_t st) { return launch_wino24_t<2, true>(p, st); }

}  // namespace se
