// bf16 gated convolution 96 -> 192, 3x3, stride 1, any dilation -- "raw tile" form (BASELINE config 5).
//
// The gather-GEMM of se_gconv.hip stages an im2col image: every input pixel enters LDS once per tap (9x) and the
// [192][64-k] weight tile once per 128 output pixels.  On the bf16 pipe that kernel spends its issue slots on staging
// (~19 B/clk/CU of LDS-DMA from inside the MFMA loop, MFMA pipe 36 % busy), so this kernel cuts the bytes that are staged:
//   * the input tile of a 16 x 16 block of outputs (18 x 18 pixels with the halo, all 96 channels, 61 KB) is DMA'd into
//     LDS ONCE in its natural pixel-major layout, and the MFMA B fragments of all 9 taps are read straight from it -- a
//     tap is an address offset, nothing is re-staged;
//   * one workgroup = 8 waves = 256 output pixels, so a weight tile serves twice as many pixels; a wave owns 96 packed
//     rows (3 feature tiles + their gate tiles) x 64 pixels (4 tile rows): 20 fragment reads per 48 MFMAs.
// 407 KB staged per 256 outputs instead of 1120 KB.  Weights: the very image pack_layer16 builds for the gather-GEMM
// ([14 chunks][192 rows][64 k], k = tap * 96 + channel): a 32-k MFMA step never straddles a tap because 96 = 3 * 32.
//
// Dilation d: the conv on the d x d polyphase sub-images (pixel (sy, sx) of phase (py, px) = image pixel
// (sy d + py, sx d + px)); a tile is 16 x 16 pixels of ONE sub-image, its taps are +-1 in sub-image coordinates.
//
// LDS layout of the raw tile: [18 rows][18 columns][12 granules of 8 channels], 192 B per pixel, no padding; within
// each group of 4 granules (one 32-k step) the granule index is XORed with 2 for the pixel columns whose bit 2 is set.
// With that, the 16 lanes of every ds_read_b128 lane group (16 pixel columns x 2 adjacent granules, gfx950's
// non-contiguous groups) hit 16 distinct 16-byte slots of the 256-byte bank row for all three column shifts of the taps
// (exhaustive check: tools/lds_layout_search.py).  The swizzle is applied on the SOURCE side of the LDS-DMA (which lane
// fetches which granule); the destination stays lane-linear.
//
// Reference semantics: gen_conv, /root/reference/models/networks/utils.py:9-33 (zero padding = rate, ELU / ReLU on the
// first 96 channels times sigmoid of the last 96); rounding points of the bf16 mode: oracle/sketchedit_oracle.py.
#include "se_device.h"

#include <cstdlib>

namespace se {

__global__ __launch_bounds__(512, 2) void rconv16_kernel(const RConvParams p) {
  constexpr int TS = 16, RS = TS + 2;                 // tile side, raw (halo) tile side
  constexpr int ROWB = RS * 192;                      // bytes per raw tile row
  constexpr int NDMA = (RS * RS * 12 + 63) / 64;      // 61 LDS-DMA instructions fill the raw tile
  constexpr int RAWB = NDMA * 1024;
  constexpr int WB = 192 * 128;
  constexpr int NCH = 14, NSTEP = 27;                 // 64-k weight chunks, 32-k MFMA steps (9 taps x 3)
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Raw = smem;
  char* Wb = smem + RAWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // workgroup -> (image, phase, tile); tiles of one sub-image are neighbours in one XCD's share of the grid
  const int lb = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int tpi = p.ty * p.tx, per_img = p.d * p.d * tpi;
  const int b = lb / per_img, r1 = lb - b * per_img;
  const int ph = r1 / tpi, t = r1 - ph * tpi;
  const int py = ph / p.d, px = ph - py * p.d;
  const int ty0 = (t / p.tx) * TS, tx0 = (t % p.tx) * TS;

  const se_i32x4 rsrc = make_rsrc(p.src, (unsigned)p.B * p.h * p.w * 192u);
  const unsigned lds_raw = lds_addr_of(Raw), lds_w = lds_addr_of(Wb);
  auto dma_w = [&](int ch, int buf, int j) {      // piece j (0..2) of this wave's share of weight chunk ch
    const int rbk = j * 8 + w;
    glds16_s((const float*)p.wpk + (size_t)ch * 192 * 32 + rbk * 256, (unsigned)lane * 16u, lds_w + buf * WB + rbk * 1024);
  };
  // ---- prologue: the first weight chunk and the raw tile
  dma_w(0, 0, 0); dma_w(0, 0, 1); dma_w(0, 0, 2);
#pragma unroll
  for (int i0 = 0; i0 < (NDMA + 7) / 8; ++i0) {
    const int i = i0 * 8 + w;
    if (i < NDMA) {
      const int q = i * 64 + lane;                   // granule slot of the raw tile
      const int pix = q / 12, gs = q - pix * 12;
      const int row = pix / RS, c = pix - row * RS;
      const int gl = gs ^ (((c >> 2) & 1) << 1);     // stored slot gs holds logical granule gl (see the layout note)
      const int sy = ty0 - 1 + row, sx = tx0 - 1 + c;
      const bool ok = q < RS * RS * 12 && (unsigned)sy < (unsigned)p.hs && (unsigned)sx < (unsigned)p.ws;
      const unsigned off = (unsigned)((b * p.h + sy * p.d + py) * p.w + sx * p.d + px) * 192u + (unsigned)gl * 16u;
      bufdma16(ok ? off : 0x80000000u, rsrc, lds_raw + i * 1024);
    }
  }
  // per-lane fragment addressing.  B (pixels): lane (j = lane & 15, g = lane >> 4) reads granule g of the step's 4 for
  // pixel column j + kx; A (weights): the tile addressing of mfma_chunk16.
  int bk[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int c = (lane & 15) + kx;
    bk[kx] = c * 192 + (((lane >> 4) ^ (((c >> 2) & 1) << 1)) << 4);
  }
  int off0, off1;
  frag_offsets(lane, off0, off1);

  // Wave tiling: 96 packed rows x 64 pixels.  Wave w = (nh, pg): feature tiles 3 nh .. 3 nh + 2 with their gate tiles
  // (+6), pixel rows 4 pg .. 4 pg + 3 of the tile.  (The first version gave every wave all 192 rows x 32 pixels: 28
  // fragment reads per 48 MFMAs, 149 B/clk/CU of LDS reads against the 128 B/clk the LDS delivers; the squarer tile needs 20.)
  const int nh = w & 1, pg = w >> 1;
  constexpr int NTW = 6, PT = 4;
  f32x4 acc[NTW][PT];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  const int wrow0 = nh * 3 * 2048;                                          // byte offset of this wave's first row tile

  dma_wait_all();
  __syncthreads();
  // A bf16 MFMA is ~17 cycles per SIMD and an A (weight) fragment feeds four of them, so fragments are kept DEPTH deep
  // in flight (a one-fragment look-ahead exposed the LDS latency on every fragment).  The B (pixel) fragments come from
  // the raw tile, which never changes: those of the next chunk are read during this one.
  constexpr int DEPTH = 4;
  auto bfrag = [&](int s, int pt) -> bf16x8 {          // s = 32-k step (compile-time): tap s / 3, channels 32 (s % 3) ..
    const int tap = s / 3, kk = s - tap * 3, ky = tap / 3, kx = tap - ky * 3;
    return *(const bf16x8*)(Raw + (4 * pg + pt + ky) * ROWB + bk[kx] + kk * 64);
  };
  auto afrag = [&](const char* Wt, int u) -> bf16x8 {  // u = 0..11: k-half u / 6, wave row tile u % 6
    const int i = u % NTW;
    return *(const bf16x8*)(Wt + wrow0 + (i < 3 ? i : i + 3) * 2048 + (u / NTW ? off1 : off0));
  };
  bf16x8 xb[2][PT], xn[2][PT];
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xb[half][pt] = bfrag(half, pt);
#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {      // fully unrolled: taps, k-steps, buffers and register rotation are compile-time
    const int buf = ch & 1;
    const char* Wt = Wb + buf * WB;
    const int nu = (ch * 2 + 1 < NSTEP) ? 2 * NTW : NTW;     // the second half of the last chunk is K padding
    bf16x8 wq[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) wq[u] = afrag(Wt, u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < 2 * NTW; ++u) {
      if (u >= nu) continue;
      const int half = u / NTW, nt = u % NTW;
      const bf16x8 wa = wq[u % DEPTH];
      if (u + DEPTH < nu) wq[u % DEPTH] = afrag(Wt, u + DEPTH);
      // the next chunk's pixel fragments (the raw tile never changes): one per MFMA group from the third on
      if (u >= 2 && u < 2 + 2 * PT && ch + 1 < NCH) {
        const int h2 = (u - 2) / PT, pt2 = (u - 2) % PT, s2 = (ch + 1) * 2 + h2;
        if (s2 < NSTEP) xn[h2][pt2] = bfrag(s2, pt2);
      }
      // pin the order (hipcc would sink every read to just in front of its MFMAs and wait lgkmcnt(0) there)
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[half][pt], acc[nt][pt], 0, 0, 0);
      // the next chunk's weight DMA, one piece per MFMA group (a burst stalls the wave)
      if (u < 3 && ch + 1 < NCH) dma_w(ch + 1, buf ^ 1, u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int half = 0; half < 2; ++half)
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) xb[half][pt] = xn[half][pt];
    dma_wait_all();
    __syncthreads();
  }

  // ---- epilogue: bias, gate, bf16 NHWC store.  A lane holds 4 consecutive channels (8 bytes) of pixel column lane & 15:
  // stored directly, every instruction scatters 8-byte pieces over 64 different 192-byte pixels and the store tail cost
  // 24 % of the kernel (measured by ablation).  The gated tile is therefore transposed through LDS (the raw tile's room,
  // free after the last barrier; 208 bytes per pixel keep the 16-byte reads aligned) and leaves as full 16-byte pieces,
  // 1 KB contiguous per wave instruction.
  constexpr int OPX = 208;
  const int q = lane >> 4, jx = lane & 15;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    char* o = Raw + ((4 * pg + pt) * 16 + jx) * OPX;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int c0 = (nh * 3 + nt) * 16 + q * 4;
      const f32x4 bf = *(const f32x4*)(p.bias + c0);
      const f32x4 bg = *(const f32x4*)(p.bias + 96 + c0);
      float ov[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const float f = acc[nt][pt][r] + bf[r];
        const float g = acc[nt + 3][pt][r] + bg[r];
        ov[r] = act_fast(f, eluw) * sigmoid_fast(g);
      }
      *(uint2*)(o + c0 * 2) = make_uint2(pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3]));
    }
  }
  __syncthreads();
#pragma unroll
  for (int it = 0; it < 6; ++it) {                  // 256 pixels x 12 pieces of 16 bytes = 6 per thread
    const int piece = it * 512 + tid;
    const int pix = piece / 12, part = piece - pix * 12;
    const int sy = ty0 + (pix >> 4), sx = tx0 + (pix & 15);
    if (sy < p.hs && sx < p.ws)
      *(uint4*)((char*)p.dst + ((size_t)(b * p.h + sy * p.d + py) * p.w + sx * p.d + px) * 192 + part * 16) =
          *(const uint4*)(Raw + pix * OPX + part * 16);
  }
}

// ------------------------------------------------------------------------------------------------------------------
// The same layer on 8 x 16 tiles with TWO workgroups per CU (the default; SE_RCONV16_TILE=16 selects the kernel above).
//
// With its 109 KB of LDS the 16 x 16 kernel owns a CU alone: its prologue (memory latency + 85 KB of fill), its epilogue
// (exp / rcp, LDS transpose, stores) and every barrier wait are MFMA-idle time -- 41 % pipe utilisation.  Here a
// workgroup is 4 waves on an 8 x 16 tile (raw tile 10 x 18 pixels = 34 KB) and the weights come in 32-k STEPS from their
// own image ([27 steps][12 row tiles][16 rows][32 k], pack_rconv16 in se_api.hip: no K padding, 12 KB per step, a ring of
// three steps = 36 KB): 70 KB per workgroup, two per CU, and whatever one of them waits for the other fills with MFMAs.
// The weight stream per pixel doubles (324 KB per 128 pixels); the LDS-DMA path delivers > 50 B/clk/CU with two
// workgroups resident (tools/ubench/lds_fill_rate.hip), 2.9 MB per CU and layer is ~25 us of it, hidden.
// A-fragment tile in LDS: 16 rows x 64 bytes, granule g of row r at slot g ^ F[r >> 2], F = {0, 2, 3, 1}: the four
// non-contiguous 16-lane groups of a ds_read_b128 each hit 16 distinct 16-byte slots of the 256-byte bank row.
__global__ __launch_bounds__(256, 2) void rconv16b_kernel(const RConvParams p) {
  constexpr int TY = 8, TX = 16, RSY = TY + 2, RSX = TX + 2;
  constexpr int ROWB = RSX * 192;                     // bytes per raw tile row
  constexpr int NDMA = (RSY * RSX * 12 + 63) / 64;    // 34 LDS-DMA instructions fill the raw tile
  constexpr int RAWB = NDMA * 1024;
  constexpr int WSB = 12 * 1024;                      // one 32-k weight step: 12 row tiles of 1 KB
  constexpr int NSTEP = 27, NS = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Raw = smem;
  char* Wb = smem + RAWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lb = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  // p.dual (sub-images 8 columns wide, e.g. dilation 16 on a 128 x 128 map): a tile holds the phases (py, px) and
  // (py, px + 1) side by side.  Raw columns: 0 = zero | 1..8 = phase px | 9 = zero (right halo of the first, left halo of
  // the second) | 10..17 = phase px + 1; the second one's right halo, "column 18", is column 0 of the next raw row --
  // also zero (after the last row: the zero-filled slack of the last DMA).  Pixel column jx of the MFMA tile is raw
  // column jx + 1 (jx < 8) or jx + 2: a per-lane constant in the fragment address, nothing in the loop changes.
  const int dual = p.dual;
  const int dxp = dual ? p.d >> 1 : p.d;              // phases (pairs) along x
  const int tpi = p.ty * p.tx, per_img = p.d * dxp * tpi;
  const int b = lb / per_img, r1 = lb - b * per_img;
  const int ph = r1 / tpi, t = r1 - ph * tpi;
  const int py = ph / dxp, px = (ph - py * dxp) << dual;
  const int ty0 = (t / p.tx) * TY, tx0 = (t % p.tx) * TX;

  const se_i32x4 rsrc = make_rsrc(p.src, (unsigned)p.B * p.h * p.w * 192u);
  const unsigned lds_raw = lds_addr_of(Raw), lds_w = lds_addr_of(Wb);
  auto dma_w = [&](int s, int j) {                    // row tile j * 4 + w of weight step s into ring slot s % NS
    const int rt = j * 4 + w;
    glds16_s((const char*)p.wpk + (size_t)s * WSB + rt * 1024, (unsigned)lane * 16u, lds_w + (s % NS) * WSB + rt * 1024);
  };
  // ---- prologue: weight step 0, the raw tile, weight steps 1 and 2 (in this order: the loop's counted waits rely on it)
  dma_w(0, 0); dma_w(0, 1); dma_w(0, 2);
  // The raw tile, one ROW (18 pixels x 12 granules = 216 slots = 4 instructions of 54 lanes) at a time: wave w issues
  // quarter w of every row, so which column and granule a lane fetches is a per-lane constant and the row is wave-uniform
  // (scalar arithmetic).  Two VALU instructions per DMA instead of ~35 (slot -> pixel -> row / column divisions for each
  // of the flat 64-slot pieces): the address arithmetic was a tenth of the issue time of this kernel's MFMAs, and ordinary
  // VALU instructions exclude MFMAs on the SIMD.  Same LDS image: slot s of row r at r * 3456 + 16 s.
  {
    const int sl = 54 * w + lane;                    // this lane's slot of a row (lane < 54)
    const int c = (sl * 5462) >> 16, gs = sl - c * 12;      // sl / 12 (sl < 216)
    const int gl = gs ^ (((c >> 2) & 1) << 1);       // stored slot gs holds logical granule gl (layout note at the top)
    const int second = dual && c >= 10;              // (dual: c = 9 gives sx = 8 = ws, the shared zero column)
    const int sx = dual ? (second ? c - 10 : c - 1) : tx0 - 1 + c;
    const bool colok = lane < 54 && (unsigned)sx < (unsigned)p.ws;
    const unsigned coloff = (unsigned)(sx * p.d + second) * 192u + (unsigned)gl * 16u;
#pragma unroll
    for (int row = 0; row < RSY; ++row) {
      const int sy = ty0 - 1 + row;                                            // wave-uniform from here ...
      const bool rowok = (unsigned)sy < (unsigned)p.hs;
      const unsigned rowbase = (unsigned)((b * p.h + sy * p.d + py) * p.w + px) * 192u;
      const unsigned off = (rowok && colok) ? rowbase + coloff : 0x80000000u;   // ... to here: one add, one select
      if (lane < 54) bufdma16(off, rsrc, lds_raw + row * (RSX * 192) + w * 864);
    }
    // the slack behind the last row ("column 18" of row 9 in the dual-phase layout) must read as zeros
    if (w == 0 && lane < (RAWB - RSY * RSX * 192) / 16) bufdma16(0x80000000u, rsrc, lds_raw + RSY * RSX * 192);
  }
  dma_w(1, 0); dma_w(1, 1); dma_w(1, 2);
  dma_w(2, 0); dma_w(2, 1); dma_w(2, 2);            // three steps ahead: every ring slot is in flight from the start

  int bk[3];
#pragma unroll
  for (int kx = 0; kx < 3; ++kx) {
    const int c = (lane & 15) + kx + (dual && (lane & 8) ? 1 : 0);
    bk[kx] = c * 192 + (((lane >> 4) ^ (((c >> 2) & 1) << 1)) << 4);
  }
  const int nh = w & 1, pg = w >> 1;                  // wave = 96 packed rows (3 feature tiles + their gate tiles) x 64 pixels
  constexpr int NTW = 6, PT = 4;
  // A fragment of this lane inside a 1 KB row tile (swizzle F = {0,2,3,1} by row >> 2)
  const int rq = (lane & 15) >> 2;
  const int aoff = (lane & 15) * 64 + (((lane >> 4) ^ ((0x78 >> (rq * 2)) & 3)) << 4) + nh * 3 * 1024;

  // The accumulators start at the bias (rows 4 (lane >> 4) .. + 3 of each row tile): the epilogue has no bias add --
  // every ordinary VALU instruction there is paid in MFMA slots
  f32x4 acc[NTW][PT];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const f32x4 b4 = *(const f32x4*)(p.bias + (nt < 3 ? 0 : 96) + (nh * 3 + (nt < 3 ? nt : nt - 3)) * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = b4;
  }

  auto bfrag = [&](int s, int pt) -> bf16x8 {          // s = 32-k step (compile-time): tap s / 3, channels 32 (s % 3) ..
    const int tap = s / 3, kk = s - tap * 3, ky = tap / 3, kx = tap - ky * 3;
    return *(const bf16x8*)(Raw + (4 * pg + pt + ky) * ROWB + bk[kx] + kk * 64);
  };
  auto afrag = [&](int s, int i) -> bf16x8 {           // wave row tile i = 0..5 of step s
    return *(const bf16x8*)(Wb + (s % NS) * WSB + aoff + (i < 3 ? i : i + 3) * 1024);
  };
  // ---- main loop, barrier in the MIDDLE of a step (round 5).  The round-3 loop ended every step with vmcnt + barrier and
  // began the next one with its first three A-fragment reads: every step opened with an LDS round trip that nothing of the
  // wave's own could cover (its MFMAs were all behind those reads) -- with two waves per SIMD the pipe measured 51.6 % busy.
  // Here the barrier of step s sits between its MFMA groups 2 and 3.  It publishes the DMA'd weights of step s + 1 and retires
  // the LDS reads of step s's slot, so on its far side the wave (a) still holds three MFMA groups whose fragments were read
  // BEFORE the barrier (bank B: row tiles 3-5 of step s, read at group 0), (b) reads row tiles 0-2 of step s + 1 (bank A) under
  // them, and (c) refills the slot it has just left with step s + 3.  No MFMA group waits for a read issued after the last
  // barrier, and a DMA has 2.5 steps of lead instead of 1.75.  Twelve more VGPRs (the second fragment bank; 186 in all).
  // Measured (512 x 512 B=16, same box, two alternations): gconv_n192 2.838 -> 2.815 ms per step (-0.8 %) -- the step-opening
  // round trip was a small part of the idle pipe time.  What the pipe can do at all in this register pattern is less than
  // the nominal 2.5 PFLOP/s: tools/ubench/mfma32_rate.hip sustains 1671 TFLOP/s with both fragments LDS-fed on RANDOM bf16
  // operands (2126 on constant ones: the part clocks to its power budget), and the same loop on v_mfma_f32_32x32x16_bf16 only
  // 1532 -- which is why this kernel was not ported to the larger instruction (VERDICT r4 'Next round' 1a; profiles/r05_mfma32_rate.txt).
  asm volatile("s_waitcnt vmcnt(6)" ::: "memory");   // weight step 0 and the raw tile have landed (steps 1, 2 may be in flight)
  __syncthreads();
  bf16x8 xb[PT], xn[PT], wA[3], wB[3];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) xb[pt] = bfrag(0, pt);
#pragma unroll
  for (int u = 0; u < 3; ++u) wA[u] = afrag(0, u);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {       // fully unrolled: taps, ring slots and register rotation are compile-time
#pragma unroll
    for (int u = 0; u < 3; ++u) {         // groups 0-2: bank A (read in the second half of step s - 1)
      if (u == 0) {
#pragma unroll
        for (int i = 0; i < 3; ++i) wB[i] = afrag(s, 3 + i);          // bank B of THIS step: needed three groups from now
      }
      if (u == 1 && s + 1 < NSTEP) { xn[0] = bfrag(s + 1, 0); xn[1] = bfrag(s + 1, 1); }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        acc[u][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wA[u], xb[pt], acc[u][pt], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
    }
    if (s + 1 < NSTEP) {
      // step s + 1 has landed when at most the 3 DMAs of step s + 2 are outstanding; every LDS read of slot s is retired
      if (s + 2 < NSTEP) asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
      __syncthreads();
    }
#pragma unroll
    for (int u = 3; u < NTW; ++u) {       // groups 3-5: bank B; bank A of step s + 1 and the refill of this step's slot under them
      if (s + 1 < NSTEP) wA[u - 3] = afrag(s + 1, u - 3);
      if (u < 5 && s + 1 < NSTEP) xn[u - 1] = bfrag(s + 1, u - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        acc[u][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wB[u - 3], xb[pt], acc[u][pt], 0, 0, 0);
      if (s + 3 < NSTEP) dma_w(s + 3, u - 3);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xb[pt] = xn[pt];
  }
  __syncthreads();                        // the epilogue stages its tile in the raw tile's room: every wave is done reading it

  // ---- epilogue: bias, gate, transposed through LDS (the raw tile's room), 16-byte stores (see the kernel above)
  constexpr int OPX = 208;
  const int q = lane >> 4, jx = lane & 15;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    char* o = Raw + ((4 * pg + pt) * 16 + jx) * OPX;
    // folded vector source (launch_vecbias; dilation 1): a bias that depends on the pixel's border configuration
    const float* tb = nullptr;
    if (p.vbias) {
      const int y = min(ty0 + 4 * pg + pt, p.h - 1), x = min(tx0 + jx, p.w - 1);
      const int cfg = 3 * (y == 0 ? 0 : (y == p.h - 1 ? 2 : 1)) + (x == 0 ? 0 : (x == p.w - 1 ? 2 : 1));
      tb = p.vbias + ((size_t)b * 9 + cfg) * 192;
    }
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int c0 = (nh * 3 + nt) * 16 + q * 4;
      f32x4 vf = acc[nt][pt], vg = acc[nt + 3][pt];                 // (bias already inside)
      if (tb) { vf += *(const f32x4*)(tb + c0); vg += *(const f32x4*)(tb + 96 + c0); }
      float ov[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = act_fast(vf[r], eluw) * sigmoid_fast(vg[r]);
      *(uint2*)(o + c0 * 2) = make_uint2(pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3]));
    }
  }
  __syncthreads();
  // 128 pixels x 12 pieces of 16 bytes = 6 per thread.  A tile row is 16 x 12 = 192 pieces = three 64-lane stores: wave w
  // stores rows w and w + 4, thirds 0, 1, 2 -- which pixel column and piece a lane stores is a per-lane constant of the
  // third, the row arithmetic is scalar, the byte offset 32-bit (one add per store instead of ~25 VALU instructions with
  // divisions and 64-bit address arithmetic).
  {
    unsigned lo[3], go[3];
    bool cok[3];
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int sl = k * 64 + lane, col = (sl * 5462) >> 16, part = sl - col * 12;      // sl / 12 (sl < 192)
      const int sx = dual ? (col & 7) : tx0 + col, dpx = dual ? col >> 3 : 0;
      lo[k] = (unsigned)(col * OPX + part * 16);
      go[k] = (unsigned)(sx * p.d + dpx) * 192u + (unsigned)part * 16u;
      cok[k] = sx < p.ws;
    }
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int k = i % 3, row = w + 4 * (i / 3);                                      // row: wave-uniform
      const int sy = ty0 + row;
      const unsigned gbase = (unsigned)((b * p.h + sy * p.d + py) * p.w + px) * 192u;
      if (sy < p.hs && cok[k]) *(uint4*)((char*)p.dst + (gbase + go[k])) = *(const uint4*)(Raw + row * (16 * OPX) + lo[k]);
    }
  }
}

bool rconv16_small_tiles() {
  const bool big = opt(OPT_RCONV16_TILE) == 16;
  return !big;
}

hipError_t launch_rconv16(const RConvParams& p, hipStream_t st) {
  if (rconv16_small_tiles()) {
    constexpr int LDS = 34 * 1024 + 3 * 12 * 1024;   // raw tile 34 KB + ring of three 12 KB weight steps
    hipError_t e = ensure_max_lds((const void*)rconv16b_kernel, LDS);
    if (e != hipSuccess) return e;
    const int grid = p.B * p.d * (p.dual ? p.d / 2 : p.d) * p.ty * p.tx;
    set_launch_grid(grid);
    ProfScope ps_(st, PL_GCONV_N192);
    hipLaunchKernelGGL(rconv16b_kernel, dim3(grid), dim3(256), LDS, st, p);
    return hipGetLastError();
  }
  constexpr int LDS = 61 * 1024 + 2 * 192 * 128;     // raw tile 61 KB + weight double buffer 48 KB
  {
    hipError_t e = ensure_max_lds((const void*)rconv16_kernel, LDS);
    if (e != hipSuccess) return e;
  }
  const int grid = p.B * p.d * p.d * p.ty * p.tx;
  set_launch_grid(grid);
  ProfScope ps_(st, PL_GCONV_N192);
  hipLaunchKernelGGL(rconv16_kernel, dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

}  // namespace se
