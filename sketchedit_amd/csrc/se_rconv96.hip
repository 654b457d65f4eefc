// bf16 gated convolutions with 96 packed rows (48 features + 48 gates), stride 1, rate 1, in "raw tile" form:
//   * 3x3, 48 -> 96 (conv3, conv14, ... at the half-resolution level) and 24 -> 96 (xconv3, pmconv3),
//   * gen_deconv 96 -> 96 (nearest x2 + 3x3) as its four 2x2 sub-pixel classes on the source grid.
// The structure is that of rconv16b_kernel (se_rconv16.hip -- read that header first): the input tile of a 16 x 16 block
// of outputs with its halo is DMA'd into LDS once, pixel-major, and every tap's B fragments are read straight from it;
// the weights stream in 32-k steps ([class][step][6 row tiles][16 rows][32 k], pack_rconv96 in se_api.hip) through a ring
// of three 6 KB slots; 4 waves, each all 96 rows x 64 pixels (4 tile rows), so the gate is a register epilogue;
// 49 - 73 KB of LDS: two workgroups per CU.  The gather-GEMM these layers ran on staged every pixel 9 (4) times and sat
// at 23 % MFMA-busy.
// k order: granule (8 channels) gi = tap * CG + cg; a 32-k step is 4 consecutive granules and may straddle two taps
// (CG = 6, 3): the per-lane raw-tile offset of every step is computed once, before the loop.
// LDS layout of the raw tile: pixel stride P granules.  CG = 12: P = 12 with the XOR-2 swizzle of rconv16; CG = 6: P = 6,
// natural (1.04 average conflict degree over the steps); CG = 3: P = 6, three zero granules per pixel (1.36; natural P = 3
// gives 1.93) -- exhaustive evaluation over the 16-lane groups of ds_read_b128.
// Reference semantics: gen_conv / gen_deconv, /root/reference/models/networks/utils.py:9-51; rounding points of the bf16
// mode: oracle/sketchedit_oracle.py.
#include "se_device.h"

#include <cstdlib>

namespace se {

template <int CG, int P, int KW, bool UP, int STRIDE = 1>
__global__ __launch_bounds__(256, 2) void rconv96_kernel(const RConv96Params p) {
  constexpr int TS = 16, RS = STRIDE * (TS - 1) + KW;   // tile side (outputs), raw (halo) tile side (source pixels)
  // STRIDE 2: a tile row is stored de-interleaved by column parity -- [even columns | odd columns] -- so that the columns
  // 2 j + kx one fragment read touches (all of one parity) are CONSECUTIVE pixels of a plane: pixel-major with a stride of
  // two pixels every ds_read_b128 would be a 4-way bank conflict whatever the padding (2 * P * 4 dwords is a multiple of 8)
  constexpr int HALF = (RS + 1) / 2;                     // pixels per parity plane of a row
  constexpr int RSP = STRIDE == 2 ? 2 * HALF : RS;       // stored pixels per row
  constexpr int PIXB = P * 16, ROWB = RSP * PIXB;
  constexpr int SLOTS = RS * RSP * P;
  constexpr int NDMA = (SLOTS + 63) / 64;
  constexpr int RAWB = NDMA * 1024;
  constexpr int WSB = 6 * 1024;                       // one 32-k weight step: 6 row tiles of 1 KB
  constexpr int T = KW * KW, NG = T * CG, NSTEP = (NG + 3) / 4, NS = 3;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Raw = smem;
  char* Wb = smem + RAWB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.B * p.ty * p.tx;
  int tile, cls = 0;
  if (UP) {
    if (!class_tile((int)blockIdx.x, ntiles, p.xcd, tile, cls)) return;
  } else {
    tile = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  }
  const int b = tile / (p.ty * p.tx), t2 = tile - b * (p.ty * p.tx);
  const int ty0 = (t2 / p.tx) * TS, tx0 = (t2 % p.tx) * TS;
  const int py = cls >> 1, px = cls & 1;
  const int pady = UP ? 1 - py : 1, padx = UP ? 1 - px : 1;

  const se_i32x4 rsrc = make_rsrc(p.src, (unsigned)p.B * p.h * p.w * (unsigned)(CG * 16));
  const unsigned lds_raw = lds_addr_of(Raw), lds_w = lds_addr_of(Wb);
  const char* wsrc = (const char*)p.wpk + (size_t)cls * NSTEP * WSB;
  // weight step s: row tiles w and w + 4 (waves 0, 1 issue two DMAs per step, waves 2, 3 one)
  auto dma_w = [&](int s, int j) {
    const int rt = j * 4 + w;
    if (rt < 6) glds16_s(wsrc + (size_t)s * WSB + rt * 1024, (unsigned)lane * 16u, lds_w + (s % NS) * WSB + rt * 1024);
  };
  auto wait_newest_step = [&]() {                     // everything but the DMAs of the most recently issued weight step
    if (w < 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
  };
  // ---- prologue: weight step 0, the raw tile, weight step 1 (in this order: the counted waits rely on it)
  dma_w(0, 0); dma_w(0, 1);
  // The raw tile row by row (se_rconv16.hip): a row is RSP * P slots = NI instructions of LPI lanes, wave w issues piece
  // w % NI of the rows w / NI, w / NI + 4 / NI, ...: column and granule are per-lane constants, the row arithmetic is scalar,
  // two VALU instructions per DMA instead of ~35.  Same LDS image: slot s of row r at r * ROWB + 16 s.
  {
    constexpr int ROWSLOTS = RSP * P, NI = (ROWSLOTS + 63) / 64, LPI = ROWSLOTS / NI;
    static_assert(NI * LPI == ROWSLOTS && (NI == 1 || NI == 2 || NI == 4), "row pieces");
    const int kp = w % NI, r0 = w / NI;
    const int sl = LPI * kp + lane;                                  // this lane's slot of a row (lane < LPI)
    const int cs = sl / P, gs = sl - cs * P;                         // stored column, stored slot (compile-time divisor)
    const int c = STRIDE == 2 ? 2 * (cs % HALF) + cs / HALF : cs;    // source column of the tile (parity planes)
    const int gl = CG == 12 ? gs ^ (((c >> 2) & 1) << 1) : gs;       // stored slot gs holds logical granule gl
    const int sx = STRIDE * tx0 - padx + c;
    const bool colok = lane < LPI && gs < CG && c < RS && (unsigned)sx < (unsigned)p.w;
    const unsigned coloff = (unsigned)sx * (unsigned)(CG * 16) + (unsigned)gl * 16u;
#pragma unroll
    for (int i = 0; i < (RS * NI + 3) / 4; ++i) {
      const int row = r0 + i * (4 / NI);                             // wave-uniform from here ...
      if (row < RS) {
        const int sy = STRIDE * ty0 - pady + row;
        const bool rowok = (unsigned)sy < (unsigned)p.h;
        const unsigned rowbase = (unsigned)((b * p.h + sy) * p.w) * (unsigned)(CG * 16);
        const unsigned off = (rowok && colok) ? rowbase + coloff : 0x80000000u;      // ... to here: one add, one select
        if (lane < LPI) bufdma16(off, rsrc, lds_raw + row * ROWB + kp * (LPI * 16));   // outside the image / pad granule: zero fill
      }
    }
    // the slack behind the last row is never multiplied by a non-zero weight, but must not hold NaN bit patterns
    if (w == 0 && lane < (RAWB - RS * ROWB) / 16) bufdma16(0x80000000u, rsrc, lds_raw + RS * ROWB);
  }
  if (NSTEP > 1) { dma_w(1, 0); dma_w(1, 1); }

  // per-lane raw-tile offset of each step: lane (j = lane & 15, g4 = lane >> 4) reads granule gi = 4 s + g4
  const int jx = lane & 15, g4 = lane >> 4;
  int boff[NSTEP];
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {
    const int gi = min(s * 4 + g4, NG - 1);          // K padding (zero weights): any valid address
    const int tap = gi / CG, cg = gi - tap * CG;
    const int ky = tap / KW, kx = tap - ky * KW;
    const int c = STRIDE * jx + kx;                                  // source column inside the tile
    const int cs = STRIDE == 2 ? (c & 1) * HALF + (c >> 1) : c;        // its stored position
    const int slot = CG == 12 ? cg ^ (((c >> 2) & 1) << 1) : cg;
    boff[s] = ky * ROWB + cs * PIXB + slot * 16;
  }
  const int pg = w;                                   // wave = all 96 rows x tile rows 4 pg .. 4 pg + 3
  constexpr int NTW = 6, PT = 4;
  const int rq = (lane & 15) >> 2;
  const int aoff = (lane & 15) * 64 + (((lane >> 4) ^ ((0x78 >> (rq * 2)) & 3)) << 4);      // swizzle F = {0,2,3,1}[row >> 2]

  // accumulators start at the bias: no bias add in the epilogue (its ordinary VALU instructions cost MFMA slots)
  f32x4 acc[NTW][PT];
#pragma unroll
  for (int nt = 0; nt < NTW; ++nt) {
    const f32x4 b4 = *(const f32x4*)(p.bias + nt * 16 + (lane >> 4) * 4);      // row tiles 0-2 features, 3-5 gates: bias[48 + ...]
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = b4;
  }

  auto bfrag = [&](int s, int pt) -> bf16x8 { return *(const bf16x8*)(Raw + STRIDE * (4 * pg + pt) * ROWB + boff[s]); };
  auto afrag = [&](int s, int i) -> bf16x8 { return *(const bf16x8*)(Wb + (s % NS) * WSB + aoff + i * 1024); };
  // (The mid-step barrier pipeline of rconv16b_kernel -- se_rconv16.hip, round 5 -- was ported here too and measured on the
  // same box: gconv_n96 1.221 against 1.221 ms per 512 x 512 B=16 step.  Nothing to gain: not kept.)
  if (NSTEP > 1) wait_newest_step();
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  bf16x8 xb[PT], xn[PT];
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) xb[pt] = bfrag(0, pt);
#pragma unroll
  for (int s = 0; s < NSTEP; ++s) {       // fully unrolled: ring slots and register rotation are compile-time
    constexpr int DEPTH = 3;
    bf16x8 wq[DEPTH];
#pragma unroll
    for (int u = 0; u < DEPTH; ++u) wq[u] = afrag(s, u);
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int u = 0; u < NTW; ++u) {
      const bf16x8 wa = wq[u % DEPTH];
      if (u + DEPTH < NTW) wq[u % DEPTH] = afrag(s, u + DEPTH);
      if (u >= 1 && u < 1 + PT && s + 1 < NSTEP) xn[u - 1] = bfrag(s + 1, u - 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt)
        acc[u][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[pt], acc[u][pt], 0, 0, 0);
      // weight step s + 2 goes to the slot step s - 1 has left (everyone passed the barrier that ended it)
      if (u < 2 && s + 2 < NSTEP) dma_w(s + 2, u);
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xb[pt] = xn[pt];
    if (s + 2 < NSTEP) wait_newest_step();
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
  }

  // ---- epilogue: bias, gate, transposed through LDS (the raw tile's room), 16-byte stores: 6 pieces per pixel
  constexpr int OPX = 112;
  const int q = lane >> 4;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    char* o = Raw + ((4 * pg + pt) * 16 + jx) * OPX;
#pragma unroll
    for (int nt = 0; nt < 3; ++nt) {
      const int c0 = nt * 16 + q * 4;
      float ov[4];
#pragma unroll
      for (int r = 0; r < 4; ++r) ov[r] = act_fast(acc[nt][pt][r], eluw) * sigmoid_fast(acc[nt + 3][pt][r]);      // (bias already inside)
      *(uint2*)(o + c0 * 2) = make_uint2(pack_bf16x2(ov[0], ov[1]), pack_bf16x2(ov[2], ov[3]));
    }
  }
  __syncthreads();
  const int OW = UP ? 2 * p.w : (STRIDE == 2 ? p.ow : p.w), OH = UP ? 2 * p.h : (STRIDE == 2 ? p.oh : p.h);
#pragma unroll
  for (int it = 0; it < 6; ++it) {                  // 256 pixels x 6 pieces of 16 bytes = 6 per thread
    const int piece = it * 256 + tid;
    const int pix = piece / 6, part = piece - pix * 6;
    const int sy = ty0 + (pix >> 4), sx = tx0 + (pix & 15);
    if (sy < (STRIDE == 2 ? OH : p.h) && sx < (STRIDE == 2 ? OW : p.w)) {
      const int oy = UP ? 2 * sy + py : sy, ox = UP ? 2 * sx + px : sx;
      *(uint4*)((char*)p.dst + ((size_t)(b * OH + oy) * OW + ox) * 96 + part * 16) = *(const uint4*)(Raw + pix * OPX + part * 16);
    }
  }
}

template <int CG, int P, int KW, bool UP, int STRIDE = 1>
static hipError_t launch_rconv96_t(const RConv96Params& p, hipStream_t st) {
  constexpr int RS = STRIDE * 15 + KW, RSP = STRIDE == 2 ? 2 * ((RS + 1) / 2) : RS;
  constexpr int RAWB = ((RS * RSP * P + 63) / 64) * 1024;
  static_assert(RAWB >= 256 * 112, "the epilogue transposes the gated tile in the raw tile's room");
  constexpr int LDS = RAWB + 3 * 6 * 1024;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  hipError_t e = ensure_max_lds((const void*)rconv96_kernel<CG, P, KW, UP, STRIDE>, LDS);
  if (e != hipSuccess) return e;
  const int tiles = p.B * p.ty * p.tx;
  const int grid = UP ? class_tile_grid(tiles) : tiles;
  set_launch_grid(grid);
  ProfScope ps_(st, PL_GCONV_N96);
  hipLaunchKernelGGL((rconv96_kernel<CG, P, KW, UP, STRIDE>), dim3(grid), dim3(256), LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_rconv96(const RConv96Params& p, hipStream_t st) {
  if (p.up2) return p.CG == 12 ? launch_rconv96_t<12, 12, 2, true>(p, st) : hipErrorInvalidValue;
  if (p.stride == 2) return p.CG == 3 ? launch_rconv96_t<3, 3, 3, false, 2>(p, st) : hipErrorInvalidValue;
  if (p.CG == 6) return launch_rconv96_t<6, 6, 3, false>(p, st);
  if (p.CG == 3) return launch_rconv96_t<3, 6, 3, false>(p, st);
  return hipErrorInvalidValue;
}

}  // namespace se
