// 24 -> 24 gated 3x3 convolution, stride 1 (conv16 / allconv16 / conv_mask_16 of /root/reference/models/networks/
// editline_g.py:90-91,98-99 and editline2_g.py:32-33,40-41: the full-resolution layer in front of every 12 -> 3 / 12 -> 1
// output conv; the layer itself is gen_conv, /root/reference/models/networks/utils.py:21-33) in raw-tile form with a
// ONE-DIMENSIONAL Winograd transform: F(2,3) along x, the direct form along y.  Round 4.
//
//   out[y][2t + o] = sum_ky sum_c  A^T[o][nu] ( U[nu][ky][c] * V[y + ky - 1][t][nu][c] )
//   V[r][t][.] = B^T (x[r][2t - 1 .. 2t + 2])       B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]
//   U[nu][ky]  = G g[ky][.]                         G = [1 0 0; .5 .5 .5; .5 -.5 .5; 0 0 1]     A^T = [1 1 1 0; 0 1 -1 -1]
//
// 12 (position, kernel row) taps per 2 outputs instead of 18: 1.5x fewer multiply-adds, all coefficients 0, +-1, +-1/2.
// Why only along x: these layers run on 32 packed rows (24 real: the 25 % row padding no 16-row MFMA tiling avoids,
// DESIGN.md 7b) with K = 24 per tap, so a two-dimensional transform wants its 16 (F(2x2,3x3)) or 24 transformed weight
// planes resident next to the tiles -- 49 / 74 KB, which leaves no room for two workgroups per CU -- while the 12 taps of
// this form are 36.9 KB.  The transformed tile of a source row serves the three output rows that read it, exactly as a
// raw pixel does in rtile_kernel ("a tap is an address offset"), so the k loop is rtile_kernel's: no staging, no wait,
// no barrier.
//   * PERSISTENT workgroups, one per CU (8 waves, 147 KB of LDS): the weight image is loaded once, a workgroup walks
//     blocks of 16 x 16 outputs; wave w = output rows 2w, 2w+1 x 8 x-tiles = 16 MFMA columns, both row tiles;
//   * every thread gathers the four source pixels of (row, x-tile, granule) tasks through the buffer range check (zero
//     padding), transforms and writes T[row][nu][x-tile][24 channels] -- 864 tasks of 4 gathers, 4 packed adds, 4 LDS
//     writes for 512 threads -- for block i+1 while block i runs: the gathers are issued in front of the k loop and
//     consumed behind the epilogue (two T buffers, one barrier per block).  The first form -- one 8 x 16 block per
//     workgroup, two workgroups per CU, synchronous prologue -- measured 258 us per launch against rtile_kernel's 271:
//     its gather round trip was exposed once per 5120 cycles of MFMAs;
//   * k loop per position nu: K = 3 kernel rows x 24 channels = 18 granules in three 32-k chunks, the third half empty
//     (k-half 0 only, two of its four granules are padding with zero weights): 20 k-steps for 18 of work;
//     160 MFMAs per wave instead of rtile_kernel's 224;
//   * T entries are 96 bytes like rtile_kernel's pixels and a source row is 3072 bytes = 0 mod 256, so the 16 lanes of a
//     fragment read (two rows x eight x-tiles) hit the banks exactly as 16 consecutive pixels do there: conflict-free;
//   * the padding rows of the second row tile (rows 4-7 and 12-15 of [f8-11, -, g8-11, -]) read the LDS rows of the real
//     ones, so the weight image holds 24 rows, not 32.
#include "se_device.h"

#include <type_traits>

// Developer aid (-DSE_WINO_TRACE, tools/wino_trace.py runw2): s_memtime stamps of workgroup 0 / waves 0 and 4 at the phase
// boundaries of the first 16 blocks of rtilew2_kernel.
#ifdef SE_WINO_TRACE
__device__ unsigned long long g_rtilew2_trace[2 * 16 * 8];
extern "C" int se_debug_rtilew2_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_rtilew2_trace), sizeof(unsigned long long) * 2 * 16 * 8);
}
#define RW2_STAMP(k)                                                                      \
  do {                                                                                    \
    if (blockIdx.x == 0 && (w & 3) == 0 && trace_it < 16) {                               \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();                         \
      if (lane == 0) g_rtilew2_trace[((w >> 2) * 16 + trace_it) * 8 + (k)] = t_;          \
    }                                                                                     \
  } while (0)
#else
#define RW2_STAMP(k)
#endif

namespace se {

__global__ __launch_bounds__(512, 2) void rtilew_kernel(const RTileParams p) {
  constexpr int NT = 2, TR = 16, RH = TR + 2, NXT = 8;
  constexpr int ENT = 96;                        // bytes of one T entry: 24 channels
  constexpr int TNU = NXT * ENT;                 // 768: the eight x-tiles of one (row, nu)
  constexpr int TROW = 4 * TNU;                  // 3072: one source row
  constexpr int TB = RH * TROW;                  // 55296
  constexpr int WCH = 24 * 128;                  // one 32-k chunk of the weight image (24 physical rows)
  constexpr int WNU = 3 * WCH;                   // one position: 72 k in three chunks
  constexpr int NTASK = RH * NXT * 6;            // 864 (source row, x-tile, granule) transform tasks per block
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Wres = smem + 2 * TB;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nblk = p.B * p.ty * p.tx;
  auto block_origin = [&](int blk, int& b, int& ty0, int& tx0) {      // (wave-uniform: scalar arithmetic)
    b = (int)udiv_magic((unsigned)blk, p.div_cg_m, p.div_cg_l);        // blk / (ty * tx)
    const int t2 = blk - b * (p.ty * p.tx);
    const int by = (int)udiv_magic((unsigned)t2, p.div_rw_m, p.div_rw_l);   // t2 / tx
    ty0 = by * TR;
    tx0 = (t2 - by * p.tx) * 16;
  };
  // transform tasks of this thread: k = tid, tid + 512 (the second only for tid < 352)
  int trow[2], txt[2], tg[2];
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) {
    const int k = min(tid + rep * 512, NTASK - 1);
    trow[rep] = k / 48;
    const int rem = k - trow[rep] * 48;
    txt[rep] = rem / 6;
    tg[rep] = rem - txt[rep] * 6;
  }
  const bool task1 = tid + 512 < NTASK;
  unsigned lane_src[2];
#pragma unroll
  for (int rep = 0; rep < 2; ++rep) lane_src[rep] = (unsigned)(trow[rep] * p.Win) * 96u + (unsigned)tg[rep] * 16u;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * 96u), 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  // the four source pixels of every task of block `blk`, through the buffer range check (outside the image: zeros)
  auto gather = [&](int blk, f32x4 (&d)[2][4]) {
    int b, ty0, tx0;
    block_origin(blk, b, ty0, tx0);
    const unsigned srow0 = (unsigned)((b * p.Hin + ty0 - 1) * p.Win) * 96u;      // source row ty0 - 1, column 0 (scalar)
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      // row part + column part, either 0x80000000 when outside the image: their SATURATING sum is out of the buffer's range
      // then and the gather returns zeros (se_wino.hip).  Both parts are non-negative when valid (row offset of column 0, column
      // offset inside the row); lane_src[rep] = row * Win * 96 + 16 g and the x-tile's first column are block-invariant.
      const int sy = ty0 - 1 + trow[rep], sx0 = tx0 + 2 * txt[rep] - 1;
      const unsigned yo = ((unsigned)sy < (unsigned)p.Hin && (rep == 0 || task1)) ? srow0 + lane_src[rep] : 0x80000000u;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const unsigned xo = (unsigned)(sx0 + j) < (unsigned)p.Win ? (unsigned)(sx0 + j) * 96u : 0x80000000u;
        d[rep][j] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)__builtin_elementwise_add_sat(yo, xo), 0, 0));
      }
    }
  };
  float negone = -1.f;
  asm volatile("" : "+v"(negone));      // opaque -1: a subtraction stays one packed fma (se_wino.hip)
  auto transform = [&](char* T, const f32x4 (&d)[2][4]) {
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      if (rep == 1 && !task1) break;
      char* at = T + trow[rep] * TROW + txt[rep] * ENT + tg[rep] * 16;
      *(f32x4*)(at) = d[rep][2] * negone + d[rep][0];
      *(f32x4*)(at + TNU) = d[rep][1] + d[rep][2];
      *(f32x4*)(at + 2 * TNU) = d[rep][1] * negone + d[rep][2];
      *(f32x4*)(at + 3 * TNU) = d[rep][3] * negone + d[rep][1];
    }
  };

  const int jx = lane & 15, g4 = lane >> 4;
  const int rl = jx >> 3, xt = jx & 7;
  const int tbase = (2 * w + rl) * TROW + xt * ENT;
  // raw-tile offset of granule gi = c * 4 + g4 of the flattened (kernel row, channel group) axis, c = (chunk, k-half)
  int xoff[5];
#pragma unroll
  for (int c = 0; c < 5; ++c) {
    const int gi = min(c * 4 + g4, 17);          // granules 18, 19: K padding -- zero weights, any valid address
    const int ky = gi / 6, cg = gi - ky * 6;
    xoff[c] = tbase + ky * TROW + cg * 16;
  }
  // A fragments: physical weight row of this lane's packed row in either row tile, with the usual slot swizzle
  int aoff[NT][2];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int prow = nt == 0 ? jx : 16 + ((jx >> 3) << 2) + (jx & 3);
    aoff[nt][0] = prow * 128 + ((g4 ^ ((prow >> 1) & 7)) << 4);
    aoff[nt][1] = prow * 128 + (((4 + g4) ^ ((prow >> 1) & 7)) << 4);
  }
  f32x4 bias4[NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) bias4[nt] = *(const f32x4*)(p.bias + nt * 16 + g4 * 4);
  float neg1 = -1.f;
  asm volatile("" : "+v"(neg1));
  const int q = lane >> 4;
  const int lanec = (q & 1) * 4 + (q >> 1) * 2;                      // this lane's channel pair inside a row tile
  const unsigned lane_dst = (unsigned)(((2 * w + rl) * p.Win + 2 * xt) * p.G + lanec) * 4u;      // byte offset inside a block

  // ---- prologue: the weights once per workgroup (LDS-DMA), the first block's transformed tile
  {
    const unsigned lds_w = lds_addr_of(Wres);
    for (int i = w; i < 4 * WNU / 1024; i += 8) glds16_s(p.wpk + (size_t)i * 256, (unsigned)lane * 16u, lds_w + i * 1024);
  }
  f32x4 d[2][4];
  int blk = blockIdx.x;
  if (blk < nblk) {
    gather(blk, d);
    transform(smem, d);
  }
  dma_wait_all();
  __syncthreads();

  // ---- persistent loop over this workgroup's blocks: the gathers of block i+1 fly under the MFMAs of block i
  for (int it = 0; blk < nblk; blk += gridDim.x, ++it) {
    const int nxt = blk + gridDim.x;
    if (nxt < nblk) gather(nxt, d);
    int b, ty0, tx0;
    block_origin(blk, b, ty0, tx0);
    const char* T = smem + (it & 1) * TB;
    f32x4 oy[2][NT];                             // outputs x0, x1 of this lane's (row, x-tile): start at the bias
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) oy[0][nt] = oy[1][nt] = bias4[nt];
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      f32x4 am[NT];
#pragma unroll
      for (int ch = 0; ch < 3; ++ch) {
        const int nh = ch < 2 ? 2 : 1;           // the third chunk carries k-half 0 only
        f32x4 wq[2][NT], xb[2];
#pragma unroll
        for (int half = 0; half < nh; ++half) {
          xb[half] = *(const f32x4*)(T + xoff[ch * 2 + half] + nu * TNU);
#pragma unroll
          for (int nt = 0; nt < NT; ++nt) wq[half][nt] = *(const f32x4*)(Wres + nu * WNU + ch * WCH + aoff[nt][half]);
        }
#pragma unroll
        for (int half = 0; half < nh; ++half)
#pragma unroll
          for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
              const f32x4 cin = (ch == 0 && half == 0 && r == 0) ? (f32x4){0.f, 0.f, 0.f, 0.f} : am[nt];
              am[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[half][nt][r], xb[half][r], cin, 0, 0, 0);
            }
      }
      // A^T = [1 1 1 0; 0 1 -1 -1]
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        if (nu < 3) oy[0][nt] += am[nt];
        if (nu == 1) oy[1][nt] += am[nt];
        if (nu >= 2) oy[1][nt] = am[nt] * neg1 + oy[1][nt];
      }
    }
    // epilogue (rtile_kernel, MIXED): tile rows 0-7 features (lanes 0-31), rows 8-15 their gates (lane + 32).  Store address =
    // one scalar block base + a block-invariant 32-bit lane offset + compile-time increments (no 64-bit vector arithmetic)
    const int y = ty0 + 2 * w + rl;
    const bool oky = y < p.Hin;
    char* dblk = (char*)p.dst + ((size_t)(b * p.Hin + ty0) * p.Win + tx0) * (size_t)(p.G * 4);
    auto epilogue = [&](auto elu_tag) {
      constexpr bool ELU = decltype(elu_tag)::value;
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const bool okc = oky && nt * 8 + lanec < p.G;
#pragma unroll
        for (int o = 0; o < 2; ++o) {
          const f32x4 v = oy[o][nt];                                    // (bias already inside)
          const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
          const float f0 = __uint_as_float(s02[0]), gg0 = __uint_as_float(s02[1]);
          const float f1 = __uint_as_float(s13[0]), gg1 = __uint_as_float(s13[1]);
          float2 ov;
          ov.x = (ELU ? elu_fast(f0) : fmaxf(f0, 0.f)) * sigmoid_fast(gg0);
          ov.y = (ELU ? elu_fast(f1) : fmaxf(f1, 0.f)) * sigmoid_fast(gg1);
          const int x = tx0 + 2 * xt + o;
          if (okc && x < p.Win)
            *(float2*)(dblk + (lane_dst + (unsigned)(o * p.G * 4 + nt * 32))) = ov;
        }
      }
    };
    if (p.act == 0) epilogue(std::true_type()); else epilogue(std::false_type());
    if (nxt < nblk) transform(smem + ((it + 1) & 1) * TB, d);
    __syncthreads();
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same layers with the full two-dimensional F(2x2,3x3) transform (even heights and widths; the default).  With
// persistent workgroups the 16 transformed weight planes (24 physical rows x 128 B each = 48 KB) are loaded ONCE and the
// rest of the CU's LDS holds the transformed tiles of one 16 x 16 block: T[16 positions][64 tiles][24 channels] = 96 KB --
// what did not fit beside two workgroups per CU fits beside none.  96 MFMAs per wave and block instead of 160 (F(2,3)
// along x) and 224 (direct).
//   * a thread owns (tile, granule): 16 gathers of its 4 x 4 patch (issued under the MFMAs of the block before), the
//     separable transform in registers (32 register operations), 16 LDS writes -- one task for 384 of the 512 threads;
//   * wave w: row tile w & 1, column tile (16 tiles = two tile rows) w >> 1; a position is K = 24 = 6 k-steps: channels
//     0-15 as four 16-byte granules (k-half 0), channels 16-23 as four 8-byte pieces of which k-steps 0, 1 take one
//     element each -- no K padding; two positions run interleaved (two independent accumulators);
//   * T has one buffer: [MFMA phase + epilogue] barrier [transform of the next block] barrier.
// Measured (256 x 256, batch 32; same box): direct rtile_kernel 271 us, F(2,3) along x 207 us, this form 191 us per launch.
// Its MFMAs are no longer what binds: removed one at a time (timing builds), the next block's gathers are 14 % of a launch,
// the transform 13 %, the MFMAs + folds 10 %, the activation arithmetic 6 %; the rest is the serial chain of a block
// (two barriers, 32 LDS writes, fragment reads, stores) that one workgroup per CU cannot overlap with anything.  Also
// measured on the one-dimensional form, without effect: gathers two blocks ahead (two register sets), and two 4-wave
// workgroups per CU with one T buffer each (201 instead of 207 us).  The 8-byte k-half-1 fragment reads put tiles j and j + 8
// on the same banks (41 % of the LDS cycles are conflict cycles); the conflict-free alternative -- 16-byte reads in the pattern
// of k-half 0 with a per-lane element select -- reads twice the bytes and adds 64 selects per block: +15 %, not kept.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512, 2) void rtilew2_kernel(const RTileParams p) {
  constexpr int ENT = 96, NTILE = 64;
  constexpr int TPOS = NTILE * ENT;              // 6144: the 64 tiles of one position
  constexpr int TB = 16 * TPOS;                  // 98304
  constexpr int WPOS = 24 * 128;                 // one position of the weight image
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* T = smem;
  char* Wres = smem + TB;
  char* Ost = smem + TB + 16 * WPOS;             // 12 KB: the block's outputs in their memory layout (16 rows x 16 pixels x 48 B)

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int rt = w & 1, ct = w >> 1;
  const int nblk = p.B * p.ty * p.tx;
  auto block_origin = [&](int blk, int& b, int& ty0, int& tx0) {      // (wave-uniform: scalar arithmetic)
    b = (int)udiv_magic((unsigned)blk, p.div_cg_m, p.div_cg_l);        // blk / (ty * tx)
    const int t2 = blk - b * (p.ty * p.tx);
    const int by = (int)udiv_magic((unsigned)t2, p.div_rw_m, p.div_rw_l);   // t2 / tx
    ty0 = by * 16;
    tx0 = (t2 - by * p.tx) * 16;
  };
  // transform task of this thread: (tile, granule) for tid < 384
  const bool has_task = tid < NTILE * 6;
  const int tt = has_task ? tid / 6 : 0, tgr = has_task ? tid - tt * 6 : 0;
  const int ttyl = tt >> 3, ttxl = tt & 7;
  const unsigned rowbytes = (unsigned)p.Win * 96u;
  const unsigned lane_y0 = (unsigned)(2 * ttyl) * rowbytes + (unsigned)tgr * 16u;       // patch row 0 relative to source row ty0 - 1
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((unsigned)p.B * (unsigned)p.Hin * (unsigned)p.Win * 96u), 0x00020000);
  typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
  auto gather = [&](int blk, f32x4 (&d)[4][4]) {
    int b, ty0, tx0;
    block_origin(blk, b, ty0, tx0);
    const unsigned srow0 = (unsigned)((b * p.Hin + ty0 - 1) * p.Win) * 96u;      // source row ty0 - 1, column 0 (scalar)
    unsigned yo[4], xo[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int sy = ty0 + 2 * ttyl - 1 + i, sx = tx0 + 2 * ttxl - 1 + i;
      yo[i] = (has_task && (unsigned)sy < (unsigned)p.Hin) ? srow0 + lane_y0 + (unsigned)i * rowbytes : 0x80000000u;
      xo[i] = (unsigned)sx < (unsigned)p.Win ? (unsigned)sx * 96u : 0x80000000u;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
        d[i][j] = __builtin_bit_cast(f32x4, (u32x4)__builtin_amdgcn_raw_buffer_load_b128(rs, (int)__builtin_elementwise_add_sat(yo[i], xo[j]), 0, 0));
  };
  float negone = -1.f;
  asm volatile("" : "+v"(negone));      // opaque -1: a subtraction stays one packed fma (se_wino.hip)
  // V = B^T d B, B^T = [1 0 -1 0; 0 1 1 0; 0 -1 1 0; 0 1 0 -1]: columns first (in place), then rows
  auto transform = [&](f32x4 (&d)[4][4]) {
    if (!has_task) return;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const f32x4 c0 = d[i][2] * negone + d[i][0], c1 = d[i][1] + d[i][2], c2 = d[i][1] * negone + d[i][2], c3 = d[i][3] * negone + d[i][1];
      d[i][0] = c0; d[i][1] = c1; d[i][2] = c2; d[i][3] = c3;
    }
    // granules 4, 5 (channels 16-23: the 8-byte k-half-1 pieces) are stored swapped in the tiles with bit 3 set, so that the
    // ds_read_b64 of tiles j and j + 8 -- 8 * 96 bytes = 3 bank rows apart -- fall on different halves of the 32-byte region
    // (round 5: these reads were 2-way bank conflicts, 41 % of the kernel's LDS cycles; conflict-free now, same bytes read)
    char* at = T + tt * ENT + (tgr >= 4 ? tgr ^ ((tt >> 3) & 1) : tgr) * 16;
#pragma unroll
    for (int nu = 0; nu < 4; ++nu) {
      *(f32x4*)(at + (0 * 4 + nu) * TPOS) = d[2][nu] * negone + d[0][nu];
      *(f32x4*)(at + (1 * 4 + nu) * TPOS) = d[1][nu] + d[2][nu];
      *(f32x4*)(at + (2 * 4 + nu) * TPOS) = d[1][nu] * negone + d[2][nu];
      *(f32x4*)(at + (3 * 4 + nu) * TPOS) = d[3][nu] * negone + d[1][nu];
    }
  };

  const int jx = lane & 15, g4 = lane >> 4;
  // B fragments: this lane's tile of the wave's column tile; k-half 0 = granule g4, k-half 1 = the 8-byte piece g4 of channels 16-23
  // (piece g4 of a tile with bit 3 set lives at piece g4 ^ 2: the transform stores granules 4 and 5 swapped there)
  const int xoff0 = (ct * 16 + jx) * ENT + g4 * 16, xoff1 = (ct * 16 + jx) * ENT + 64 + (g4 ^ ((jx >> 3) << 1)) * 8;
  // A fragments: physical weight row of this lane's packed row, with the usual slot swizzle
  const int prow = rt == 0 ? jx : 16 + ((jx >> 3) << 2) + (jx & 3);
  // (k-half 1: channels 16 + 2 g4, 17 + 2 g4 sit in the lower / upper 8 bytes of slot 4 + g4 for even / odd g4 (pack_rtilew):
  // lane groups g4 and g4 ^ 1 of rows 2k, 2k + 2 share a slot after the swizzle -- in different halves, no bank conflict)
  const int aoff0 = prow * 128 + ((g4 ^ ((prow >> 1) & 7)) << 4), aoff1 = prow * 128 + (((4 + g4) ^ ((prow >> 1) & 7)) << 4) + (g4 & 1) * 8;
  const f32x4 bias4 = *(const f32x4*)(p.bias + rt * 16 + g4 * 4);
  float neg1 = -1.f;
  asm volatile("" : "+v"(neg1));
  const int lanec = (g4 & 1) * 4 + (g4 >> 1) * 2;                    // this lane's channel pair inside the row tile
  const int tle = ct * 16 + jx, tyl = tle >> 3, txl = tle & 7;       // the tile whose outputs this lane holds
  const int lane_ost = ((2 * tyl * 16 + 2 * txl) * 12 + rt * 8 + lanec) * 4;      // byte offset inside the staged block
  const unsigned dinc_y = (unsigned)(p.Win * p.G) * 4u;

  // ---- prologue: the weights once per workgroup (LDS-DMA), the first block's transformed tiles
  {
    const unsigned lds_w = lds_addr_of(Wres);
    for (int i = w; i < 16 * WPOS / 1024; i += 8) glds16_s(p.wpk + (size_t)i * 256, (unsigned)lane * 16u, lds_w + i * 1024);
  }
  f32x4 d[4][4];
  int blk = blockIdx.x;
  if (blk < nblk && w < 6) {
    gather(blk, d);
    transform(d);
  }
  dma_wait_all();
  __syncthreads();

  for (int trace_it = 0; blk < nblk; blk += gridDim.x, ++trace_it) {
    const int nxt = blk + gridDim.x;
    RW2_STAMP(0);
    if (nxt < nblk && w < 6) gather(nxt, d);     // flies under the MFMA phase (waves 6, 7 have no transform task)
    RW2_STAMP(1);
    int b, ty0, tx0;
    block_origin(blk, b, ty0, tx0);
    f32x4 oy[2][2];                              // the 2 x 2 outputs of this lane's tile: start at the bias
    oy[0][0] = oy[0][1] = oy[1][0] = oy[1][1] = bias4;
    // ---- MFMA phase: 16 positions, two at a time (independent accumulators), each 4 + 2 k-steps; the fragments of the next
    // pair are read in front of this pair's MFMAs (with one workgroup per CU nothing else hides the LDS latency)
    f32x4 wa[2][2], xa[2][2];
    f32x2 wb[2][2], xb[2][2];
    auto read_pair = [&](int pp, int s_) {
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pos = 2 * pp + u;
        wa[s_][u] = *(const f32x4*)(Wres + pos * WPOS + aoff0);
        wb[s_][u] = *(const f32x2*)(Wres + pos * WPOS + aoff1);
        xa[s_][u] = *(const f32x4*)(T + pos * TPOS + xoff0);
        xb[s_][u] = *(const f32x2*)(T + pos * TPOS + xoff1);
      }
    };
    read_pair(0, 0);
#pragma unroll
    for (int pp = 0; pp < 8; ++pp) {
      const int s_ = pp & 1;
      if (pp + 1 < 8) read_pair(pp + 1, s_ ^ 1);
      __builtin_amdgcn_sched_barrier(0);
      f32x4 am[2];
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          am[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[s_][u][r], xa[s_][u][r], r == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : am[u], 0, 0, 0);
#pragma unroll
      for (int r = 0; r < 2; ++r)
#pragma unroll
        for (int u = 0; u < 2; ++u)
          am[u] = __builtin_amdgcn_mfma_f32_16x16x4f32(wb[s_][u][r], xb[s_][u][r], am[u], 0, 0, 0);
      // Y[a][bb] += At[a][xi] At[bb][nu] M,  At = [1 1 1 0; 0 1 -1 -1]
#pragma unroll
      for (int u = 0; u < 2; ++u) {
        const int pos = 2 * pp + u, xi = pos >> 2, nu = pos & 3;
        const int ay[2] = {xi < 3 ? 1 : 0, xi == 0 ? 0 : (xi == 1 ? 1 : -1)};
        const int ax[2] = {nu < 3 ? 1 : 0, nu == 0 ? 0 : (nu == 1 ? 1 : -1)};
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
          for (int bb = 0; bb < 2; ++bb) {
            const int c = ay[a] * ax[bb];
            if (c > 0) oy[a][bb] += am[u];
            else if (c < 0) oy[a][bb] = am[u] * neg1 + oy[a][bb];
          }
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    RW2_STAMP(2);
    // ---- epilogue (MIXED): rows 0-7 features (lanes 0-31), rows 8-15 their gates (lane + 32)
    char* dblk = (char*)p.dst + ((size_t)(b * p.Hin + ty0) * p.Win + tx0) * (size_t)(p.G * 4);
    const bool okc = rt * 8 + lanec < p.G;
    auto epilogue = [&](auto elu_tag) {
      constexpr bool ELU = decltype(elu_tag)::value;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          const f32x4 v = oy[a][bb];                                    // (bias already inside)
          const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
          const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
          const float f0 = __uint_as_float(s02[0]), gg0 = __uint_as_float(s02[1]);
          const float f1 = __uint_as_float(s13[0]), gg1 = __uint_as_float(s13[1]);
          float2 ov;
          ov.x = (ELU ? elu_fast(f0) : fmaxf(f0, 0.f)) * sigmoid_fast(gg0);
          ov.y = (ELU ? elu_fast(f1) : fmaxf(f1, 0.f)) * sigmoid_fast(gg1);
          if (okc) *(float2*)(Ost + lane_ost + a * (16 * 48) + bb * 48) = ov;
        }
    };
    if (p.act == 0) epilogue(std::true_type()); else epilogue(std::false_type());
    RW2_STAMP(3);
    __syncthreads();                             // every wave has read T; the block's outputs are staged
    RW2_STAMP(4);
    // the staged outputs leave as 16-byte pieces of contiguous 768-byte rows (the epilogue's own 8-byte stores scattered over
    // 16 pixels per instruction were ~3000 of a block's 13500 cycles)
#pragma unroll
    for (int rep = 0; rep < 2; ++rep) {
      const int k = tid + rep * 512;             // piece k: row k / 48, 16-byte piece k % 48 of the row (3 pieces per pixel)
      if (k < 16 * 48) {
        const int row = k / 48, pc = k - row * 48;
        if (ty0 + row < p.Hin && tx0 + pc / 3 < p.Win)
          *(f32x4*)(dblk + (size_t)row * dinc_y + pc * 16) = *(const f32x4*)(Ost + k * 16);
      }
    }
    if (nxt < nblk) transform(d);
    RW2_STAMP(5);
    __syncthreads();
    RW2_STAMP(6);
  }
}

// CUs of the current device (persistent grids = one workgroup per CU); queried once per device, not per launch
static int cu_count() {
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return 256;
  if (cached[dev] == 0) {
    int n = 0;
    cached[dev] = (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) == hipSuccess && n > 0) ? n : 256;
  }
  return cached[dev];
}

hipError_t launch_rtilew(const RTileParams& p, hipStream_t st) {
  constexpr int LDS = 2 * 18 * 3072 + 4 * 3 * 24 * 128;      // two transformed tiles 108 KB + weights 36 KB
  {
    hipError_t e = ensure_max_lds((const void*)rtilew_kernel, LDS);
    if (e != hipSuccess) return e;
  }
  const int nblk = p.B * p.ty * p.tx;
  const int cus = cu_count();
  const int grid = nblk < cus ? nblk : cus;      // persistent: one workgroup per CU (147 KB of LDS), each walks nblk / grid blocks
  set_launch_grid(grid);
  ProfScope ps_(st, PL_GCONV_N24);
  hipLaunchKernelGGL(rtilew_kernel, dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_rtilew2(const RTileParams& p, hipStream_t st) {
  constexpr int LDS = 16 * 64 * 96 + 16 * 24 * 128 + 16 * 16 * 48;      // transformed tiles 96 KB + weights 48 KB + staged outputs 12 KB
  {
    hipError_t e = ensure_max_lds((const void*)rtilew2_kernel, LDS);
    if (e != hipSuccess) return e;
  }
  const int nblk = p.B * p.ty * p.tx;
  const int cus = cu_count();
  const int grid = nblk < cus ? nblk : cus;
  set_launch_grid(grid);
  ProfScope ps_(st, PL_GCONV_N24);
  hipLaunchKernelGGL(rtilew2_kernel, dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

}  // namespace se
