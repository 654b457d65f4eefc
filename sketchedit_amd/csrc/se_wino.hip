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
// Loop over it = (position, 32-k chunk), 48 (96 with two sources) iterations, fully unrolled, LDS rings of 3 X and 4 W
// tiles, one barrier per iteration:
//   * X tile [64 tiles][32 k]: each lane builds ONE granule = (B^T x B) at this position from 4 raw input granules
//     (global loads issued an iteration earlier) and writes it to LDS -- the input transform is fused, no
//     transformed tensor ever exists in HBM;
//   * W tile [192][32 k] of the host-transformed weights G g G^T by LDS-DMA;
//   * 48 MFMAs per wave into the position accumulator; at the start of the next position it is folded into the four
//     output accumulators with the A^T coefficients (inverse transform in registers).
// What the s_memtime traces (tools/wino_trace.py) showed, and what the schedule does about it:
//   1. fp32 MFMA and VALU do not overlap on a SIMD: a wave's VALU instruction does not issue while its SIMD partner
//      streams v_mfma_f32_16x16x4_f32 (s_setprio does not change that), and inside one wave every VALU instruction
//      is a lost MFMA slot.  So the loop carries as few VALU instructions as possible (~25 per wave and iteration,
//      was ~70): gather offsets are computed once per POSITION (4 LDS reads + 4 saturating adds; source offsets live
//      in LDS, not in registers), zero padding is the buffer range check of the gather and the B^T signs are
//      compile-time +/- (round 3), loads and DMA use scalar-base + 32-bit-offset addressing, the first MFMA of a position takes
//      C = 0 instead of zeroed registers, the fold generates only its 12 (of 16) non-zero terms, and there is one
//      accumulator set -- a second one to "hide" the fold buys nothing and costs 24 registers.
//   2. A wave that issues its 7 vector-memory instructions back to back (all 8 waves do so at the same point) stalls
//      ~1000 cycles in front of the full vector-memory queue and cannot issue MFMAs meanwhile: they are spread
//      over the MFMA groups of the second half of the iteration.
//   3. The k-half 0 fragments of the next iteration are read before the barrier (its slot was published one barrier
//      earlier), so the MFMA stream continues straight across the barrier.
//   4. Granules and W DMA are waited for (one vmcnt(0)) half an iteration after they were issued; the W tile goes
//      to a 4-slot ring two barriers ahead of its reader so that this one conservative wait is enough.
// (Also measured: all VALU work right behind the barrier, where both waves of a SIMD could overlap their latency-
// bound VALU phases, with the vector-memory instructions in the first four groups: no gain -- the memory queue
// stalls come back.)
// Per-launch time 200 -> 164 us (B=32, 64x64; 115 TF/s executed = 73 % of the fp32 MFMA peak, 258 TF/s in direct-
// convolution terms).  Tried and NOT faster: a ping-pong split (waves 0-3 / 4-7 one phase apart, two barriers per
// iteration: 237 us), two 4-wave workgroups per CU, 16-wave workgroups with an LDS gate exchange, raw granules in
// registers, an "LDS patch" form without global loads in the loop, two positions per barrier.
#include "se_device.h"

#include <cstdlib>

// Developer aid, compiled only with -DSE_WINO_TRACE (tools/wino_trace.py builds a separate debug library):
// s_memtime stamps of block 0 / waves 0 and 4 (the two waves of SIMD 0) at the phase boundaries of every iteration.
#ifdef SE_WINO_TRACE
__device__ unsigned long long g_wino_trace[96 * 8];
extern "C" int se_debug_wino_trace(unsigned long long* host_out) {
  return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_wino_trace), sizeof(unsigned long long) * 96 * 8);
}
// stamps go to LDS (a global store per stamp would sit in vmcnt and distort the waits being measured)
#define WINO_TRACE_LDS (3 * 64 * 128 + 4 * 192 * 128 + 8 * 512 * 4)
#define WINO_STAMP(k)                                                   \
  do {                                                                  \
    if (blockIdx.x == 0 && (w & 3) == 0) {                              \
      const unsigned long long t_ = __builtin_amdgcn_s_memtime();       \
      if (lane == 0) ((unsigned long long*)(smem + WINO_TRACE_LDS))[((w >> 2) * 48 + it) * 8 + (k)] = t_; \
    }                                                                   \
  } while (0)
#define WINO_TRACE_DUMP()                                               \
  do {                                                                  \
    if (blockIdx.x == 0 && (w & 3) == 0)                                \
      for (int i_ = lane; i_ < 48 * 8; i_ += 64)                        \
        g_wino_trace[(w >> 2) * 48 * 8 + i_] = ((unsigned long long*)(smem + WINO_TRACE_LDS))[(w >> 2) * 48 * 8 + i_]; \
  } while (0)
#else
#define WINO_TRACE_LDS 0
#define WINO_STAMP(k)
#define WINO_TRACE_DUMP()
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
  char* Wb = smem + 3 * XB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int chh = w & 1, tg = w >> 1;          // channel half, tile group
  const int tile_base = (p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x) * TILES;
  const int tpi = p.th * p.tw;                 // tiles per image

  // tile -> (batch, first output pixel).  iy walks the tile grid; y0 = 2d*(iy/d) + iy%d
  auto tile_origin = [&](int t, int& b, int& y0, int& x0) {
    b = (int)udiv_magic((unsigned)t, p.div_tpi_m, p.div_tpi_l);
    const int rem = t - b * tpi;
    const int iy = (int)udiv_magic((unsigned)rem, p.div_tw_m, p.div_tw_l), ix = rem - iy * p.tw;
    const int qy = (int)udiv_magic((unsigned)iy, p.div_d_m, p.div_d_l), qx = (int)udiv_magic((unsigned)ix, p.div_d_m, p.div_d_l);
    y0 = 2 * p.d * qy + (iy - qy * p.d);
    x0 = 2 * p.d * qx + (ix - qx * p.d);
  };

  // ---- staging role: granule (row = tile tid>>3, physical slot tid&7)
  const int srow = tid >> 3, ps = tid & 7;
  const int s_log = ps ^ ((srow >> 1) & 7);
  // Source offsets of the 4x4 input tile, kept in LDS (read once per position; registers are the scarce resource):
  //   Ysrc[i][tid] = byte offset of pixel row y_i (+ this lane's granule), or 0x80000000 if outside / invalid tile
  //   Xsrc[i][tid] = byte offset of column x_i inside the row, or 0x80000000 if outside
  // A gather offset is their SATURATING sum (v_add_u32 clamp): >= 2^31 as soon as either part is outside (valid sums stay
  // below 2^31: the launch guards the tensor's bytes), and the gather goes through a buffer resource whose range check
  // returns zeros there -- zero padding costs no arithmetic at all, and the B^T factors that remain are signs, applied as
  // compile-time +/- in the transform.  set_pos is 4 LDS reads and 4 adds; the transform 8 instructions per granule
  // (round 3: every VALU instruction of the loop is a lost fp32 MFMA slot; the first form had 8 reads, 4 adds, 4
  // multiplies per position and 4 multiplies + 6 packed fmas per granule).
  int* Ysrc = (int*)(smem + 3 * XB + 4 * WB);
  int* Xsrc = Ysrc + 4 * 512;
  const unsigned lane_coff = (unsigned)s_log * 16u;
  int bimg;             // batch index of this lane's tile (address of the per-image vector source)
  {
    const int t = tile_base + srow;
    int b, y0, x0;
    tile_origin(t < p.total_tiles ? t : 0, b, y0, x0);
    bimg = b;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int y = y0 + (i - 1) * p.d, x = x0 + (i - 1) * p.d;
      const bool yok = t < p.total_tiles && (unsigned)y < (unsigned)p.h, xok = (unsigned)x < (unsigned)p.w;
      Ysrc[i * 512 + tid] = yok ? (int)((unsigned)((b * p.h + y) * p.w) * 384u + lane_coff) : (int)0x80000000;
      Xsrc[i * 512 + tid] = xok ? x * 384 : (int)0x80000000;
    }
  }
  const unsigned lds_w = lds_addr_of(Wb);
  int off0, off1;
  frag_offsets(lane, off0, off1);

  // fp32 MFMA and VALU instructions share the SIMD's issue time on gfx950 (a wave's VALU does not issue while its
  // SIMD partner streams fp32 MFMAs, and inside one wave every VALU instruction is a lost MFMA slot), so the loop
  // keeps the VALU count minimal: gather offsets are computed once per POSITION, the gathers and the W DMA use scalar
  // base + 32-bit lane offset addressing, LDS addresses are immediates.
  unsigned o[4];        // byte offsets of the four source pixels of the current position (+ this lane's granule); >= 2^31: outside
  unsigned ov[4];       // the same for the per-image vector source (NCHK == 6, src1_vec): vec_off | the outside bit of o[i]
  const unsigned vec_off = (unsigned)bimg * 384u + lane_coff;     // per-image vector source (NCHK == 6 only)
  auto set_pos = [&](int xi, int nu) {      // xi, nu compile-time in the unrolled loop
    // B^T rows: xi=0: +d0 -d2 | 1: +d1 +d2 | 2: -d1 +d2 | 3: +d1 -d3 ; a pixel outside the image reads as zero
    const int ia = (xi == 0 ? 0 : 1) * 512 + tid, ib = (xi == 3 ? 3 : 2) * 512 + tid;
    const int ja = (nu == 0 ? 0 : 1) * 512 + tid, jb = (nu == 3 ? 3 : 2) * 512 + tid;
    const unsigned ya = (unsigned)Ysrc[ia], yb = (unsigned)Ysrc[ib], xa = (unsigned)Xsrc[ja], xb = (unsigned)Xsrc[jb];
    o[0] = __builtin_elementwise_add_sat(ya, xa); o[1] = __builtin_elementwise_add_sat(ya, xb);
    o[2] = __builtin_elementwise_add_sat(yb, xa); o[3] = __builtin_elementwise_add_sat(yb, xb);
    if (NCHK == 6 && p.src1_vec) {
#pragma unroll
      for (int i = 0; i < 4; ++i) ov[i] = vec_off | (o[i] & 0x80000000u);      // v_and_or_b32
    }
  };
  // the gathers go through buffer resources: an offset with bit 31 set is out of range and delivers zeros
  const __amdgpu_buffer_rsrc_t rs0 = __builtin_amdgcn_make_buffer_rsrc((void*)p.src, 0, (int)((unsigned)p.B * (unsigned)p.h * (unsigned)p.w * 384u), 0x00020000);
  const __amdgpu_buffer_rsrc_t rs1 = __builtin_amdgcn_make_buffer_rsrc((void*)(NCHK == 6 ? p.src1 : p.src), 0,
      (int)(NCHK == 6 && p.src1_vec ? (unsigned)p.B * 384u : (unsigned)p.B * (unsigned)p.h * (unsigned)p.w * 384u), 0x00020000);
  // raw granules of one iteration (chunk = it % NCHK, position already selected by set_pos)
  auto load_x1 = [&](int chunk, f32x4 (&r)[4], int i) {      // granule i (0..3): one vector-memory instruction
    const bool second = NCHK == 6 && chunk >= 3;
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // (second source: a tensor, or the spatially constant pooled style vector -- one value per image, still zero padded)
    const u32x4 t = second ? __builtin_amdgcn_raw_buffer_load_b128(rs1, (int)(p.src1_vec ? ov[i] : o[i]), (chunk - 3) * 128, 0)
                           : __builtin_amdgcn_raw_buffer_load_b128(rs0, (int)o[i], chunk * 128, 0);
    r[i] = __builtin_bit_cast(f32x4, t);
  };
  auto load_x = [&](int chunk, f32x4 (&r)[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) load_x1(chunk, r, i);
  };
  // B^T signs of the four granules of position (xi, nu) (compile-time): sy = (a: xi == 2 ? - : +, b: xi == 0 or 3 ? - : +), same in x
  float negone = -1.f;
  asm volatile("" : "+v"(negone));      // opaque -1: a subtraction stays one packed fma (a plain - becomes 4 v_sub)
  auto write_x = [&](int buf, const f32x4 (&r)[4], int xi, int nu) {
    const bool nya = xi == 2, nyb = xi == 0 || xi == 3, nxa = nu == 2, nxb = nu == 0 || nu == 3;
    const bool n0 = nya != nxa, n1 = nya != nxb, n2 = nyb != nxa, n3 = nyb != nxb;      // granule i enters with a minus sign
    // sum the + terms and the - terms separately, then one subtraction: at most one packed fma besides the packed adds
    f32x4 pos = {0.f, 0.f, 0.f, 0.f}, neg = {0.f, 0.f, 0.f, 0.f};
    bool hp = false, hn = false;
    const bool ng[4] = {n0, n1, n2, n3};
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      if (ng[i]) { neg = hn ? neg + r[i] : r[i]; hn = true; }
      else { pos = hp ? pos + r[i] : r[i]; hp = true; }
    }
    const f32x4 v = !hn ? pos : (!hp ? neg * negone : neg * negone + pos);
    *(f32x4*)(Xb + buf * XB + srow * 128 + ps * 16) = v;
  };
  auto dma_w = [&](int it, int buf, int j) {      // piece j (0..2) of this wave's share of the W tile
    const int rbk = j * 8 + w;
    glds16_s(p.upk + (size_t)it * 192 * 32 + rbk * 256, (unsigned)lane * 16u, lds_w + buf * WB + rbk * 1024);
  };

  f32x4 af[3], ag[3];                  // position accumulators (feature / gate tiles)
  f32x4 of[2][2][3], og[2][2][3];      // output accumulators (a, b, tile)
#pragma unroll
  for (int j = 0; j < 3; ++j) {      // the output accumulators start at the bias: one add less per value in the epilogue
    af[j] = ag[j] = (f32x4){0.f, 0.f, 0.f, 0.f};
    const f32x4 bf0 = *(const f32x4*)(p.bias + (3 * chh + j) * 16 + (lane >> 4) * 4);
    const f32x4 bg0 = *(const f32x4*)(p.bias + 96 + (3 * chh + j) * 16 + (lane >> 4) * 4);
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) { of[a][b][j] = bf0; og[a][b][j] = bg0; }
  }
  // fold the finished position accumulators: Y[a][b] += At[a][xi] * At[b][nu] * M,  At = [1 1 1 0; 0 1 -1 -1].
  // xi and nu are compile-time in the unrolled loop: only the non-zero terms exist, as packed adds / fmas.
  // (fp32 MFMA and VALU do not overlap on a SIMD, so there is nothing to gain from a second accumulator set that
  // would let the fold run "under" the next position's MFMAs; one set saves 24 registers.)
  float neg1 = -1.f;
  asm volatile("" : "+v"(neg1));      // opaque multiplier: keeps a subtraction one packed fma (a plain -= becomes 4 v_sub)
  auto fold = [&](const f32x4 (&pf)[3], const f32x4 (&pg)[3], int xi, int nu, int j) {     // channel tile j
    const int ay[2] = {xi < 3 ? 1 : 0, xi == 0 ? 0 : (xi == 1 ? 1 : -1)};      // At[a][xi]   (xi, nu compile-time:
    const int ax[2] = {nu < 3 ? 1 : 0, nu == 0 ? 0 : (nu == 1 ? 1 : -1)};      // At[b][nu]    9 of 16 terms exist)
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int c = ay[a] * ax[b];
        if (c > 0) { of[a][b][j] += pf[j]; og[a][b][j] += pg[j]; }
        else if (c < 0) { of[a][b][j] = pf[j] * neg1 + of[a][b][j]; og[a][b][j] = pg[j] * neg1 + og[a][b][j]; }
      }
    // pin the sums here: in unrolled code hipcc would otherwise sink every fold to the end of the kernel and keep
    // all 16 position results alive (in scratch) until then
    asm volatile("" : "+v"(of[0][0][j]), "+v"(of[0][1][j]), "+v"(of[1][0][j]), "+v"(of[1][1][j]),
                      "+v"(og[0][0][j]), "+v"(og[0][1][j]), "+v"(og[1][0][j]), "+v"(og[1][1][j]));
  };

  // ---- pipeline.  One workgroup barrier per iteration; X tiles in a 3-slot, W tiles in a 4-slot LDS ring.
  //   iteration it:  groups 0-3 | vmcnt(0); X(it+2) <- granules loaded in it-1 | groups 4-7: W DMA and granule loads
  //                  of it+3, k-half 0 fragments of it+1 | barrier
  //   X(it+2): slot (it+2)%3, last read in it-1, published by this barrier, first read (fragments) in it+1
  //   W(it+3): slot (it+3)%4, last read in it-1, complete after the vmcnt(0) of it+1, published by the barrier of
  //            it+1, first read (fragments) in it+2
  // Every wait is explicit and conservative.  The granule loads are ordinary loads that hipcc tracks itself (an
  // inline-asm load whose result register hipcc may copy before the data has landed is a latent race: it showed up
  // as run-to-run differences at batch 32 in se_wino48.hip), and the single s_waitcnt vmcnt(0) of an iteration sits
  // half an iteration after the youngest vector-memory instruction was issued.  The fragments of it+1 are read
  // before the barrier (their slots were published a barrier ago), so the MFMA stream runs straight across it.
  auto end_barrier = [&]() {
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
  };
  // ---- prologue: X slots 0, 1; W slots 0, 1, 2; granules of iteration 2 in flight
  f32x4 r[4];
  set_pos(0, 0);
#pragma unroll
  for (int i0 = 0; i0 < 3; ++i0) {     // the W DMA first: its latency overlaps the granule round trip
    dma_w(i0, i0, 0);
    dma_w(i0, i0, 1);
    dma_w(i0, i0, 2);
  }
  {
    f32x4 r1[4];                       // chunks 0, 1 of position 0: both loads in flight before the first transform
    load_x(0, r);
    load_x(1, r1);
    write_x(0, r, 0, 0);
    write_x(1, r1, 0, 0);
  }
  dma_wait_all();
  __syncthreads();
  load_x(2, r);                        // granules of iteration 2 (chunk 2 of position 0)
  f32x4 wf[3], wg[3], xh;              // k-half 0 fragments of the current iteration (read one iteration ahead)
  xh = *(const f32x4*)(Xb + tg * 2048 + off0);
#pragma unroll
  for (int j = 0; j < 3; ++j) {
    wf[j] = *(const f32x4*)(Wb + (3 * chh + j) * 2048 + off0);
    wg[j] = *(const f32x4*)(Wb + (6 + 3 * chh + j) * 2048 + off0);
  }
#pragma unroll
  for (int xi = 0; xi < 4; ++xi) {               // rows of the 4x4 position grid; body unrolled: nu, the chunk, the
#pragma unroll                                   // accumulator set and the ring slots are compile-time
    for (int k = 0; k < 4 * NCHK; ++k) {
      const int it = xi * 4 * NCHK + k;
      const int nu = k / NCHK, chunk = k % NCHK;
      const int b0 = k % 3, b1 = (k + 1) % 3, b2 = (k + 2) % 3;     // X ring; == (it + i) % 3: 4*NCHK is a multiple of 3
      const int w0 = k % 4, w1 = (k + 1) % 4, w3 = (k + 3) % 4;     // W ring; == (it + i) % 4: 4*NCHK is a multiple of 4
      const char* Xt = Xb + b0 * XB + tg * 2048;
      const char* Wf = Wb + w0 * WB + (3 * chh) * 2048;
      const char* Wg = Wb + w0 * WB + (6 + 3 * chh) * 2048;
      const bool more1 = it + 1 < NIT, more2 = it + 2 < NIT, more3 = it + 3 < NIT;
      WINO_STAMP(0);
      f32x4 wf1[3], wg1[3], xh1;
      xh1 = *(const f32x4*)(Xt + off1);                      // k-half 1 fragments of this iteration
#pragma unroll
      for (int j = 0; j < 3; ++j) {
        wf1[j] = *(const f32x4*)(Wf + j * 2048 + off1);
        wg1[j] = *(const f32x4*)(Wg + j * 2048 + off1);
      }
      __builtin_amdgcn_sched_barrier(0);
      auto group = [&](const f32x4 (&a)[3], const f32x4 (&b)[3], const f32x4& x, int e, bool first) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          // first k-step of a position: C = 0 (the accumulators still hold the previous position, already folded)
          const f32x4 cf = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : af[j];
          const f32x4 cg = first ? (f32x4){0.f, 0.f, 0.f, 0.f} : ag[j];
          af[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[j][e], x[e], cf, 0, 0, 0);
          ag[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(b[j][e], x[e], cg, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
      };
      if (chunk == 0) {                      // fold the previous position (row 0, nu 0: the zero-initialised set, a no-op)
        const int pxi = nu == 0 ? xi - 1 : xi, pnu = (nu + 3) & 3;
        fold(af, ag, pxi, pnu, 0);
        fold(af, ag, pxi, pnu, 1);
        fold(af, ag, pxi, pnu, 2);
        __builtin_amdgcn_sched_barrier(0);
      }
      group(wf, wg, xh, 0, chunk == 0);
      group(wf, wg, xh, 1, false);
      group(wf, wg, xh, 2, false);
      group(wf, wg, xh, 3, false);
      WINO_STAMP(1);
      if (more2) {
        dma_wait_all();                    // granules and W DMA issued in groups 4-7 of the previous iteration
        write_x(b2, r, ((it + 2) / NCHK) >> 2, ((it + 2) / NCHK) & 3);      // the position of iteration it + 2
      }
      if ((k + 3) % NCHK == 0 && more3)    // position of the granules fetched next
        set_pos(xi + ((k + 3) / NCHK) / 4, ((k + 3) / NCHK) & 3);
      __builtin_amdgcn_sched_barrier(0);
      WINO_STAMP(2);
      // The 7 vector-memory instructions of an iteration (3 W DMA pieces, 4 granule loads) are spread over the MFMA
      // groups: issued back to back by all 8 waves they fill the CU's vector-memory queue, and a wave stuck in
      // front of a full queue cannot issue its MFMAs either.
      group(wf1, wg1, xh1, 0, false);
      if (more3) { dma_w(it + 3, w3, 0); load_x1((k + 3) % NCHK, r, 0); }
      if (more1) {                                          // k-half 0 fragments of it+1 (published slots)
        const char* Xn = Xb + b1 * XB + tg * 2048;
        xh = *(const f32x4*)(Xn + off0);
#pragma unroll
        for (int j = 0; j < 3; ++j) {
          wf[j] = *(const f32x4*)(Wb + w1 * WB + (3 * chh + j) * 2048 + off0);
          wg[j] = *(const f32x4*)(Wb + w1 * WB + (6 + 3 * chh + j) * 2048 + off0);
        }
      }
      __builtin_amdgcn_sched_barrier(0);
      group(wf1, wg1, xh1, 1, false);
      if (more3) { dma_w(it + 3, w3, 1); load_x1((k + 3) % NCHK, r, 1); }
      __builtin_amdgcn_sched_barrier(0);
      group(wf1, wg1, xh1, 2, false);
      if (more3) { dma_w(it + 3, w3, 2); load_x1((k + 3) % NCHK, r, 2); }
      __builtin_amdgcn_sched_barrier(0);
      group(wf1, wg1, xh1, 3, false);
      if (more3) load_x1((k + 3) % NCHK, r, 3);
      __builtin_amdgcn_sched_barrier(0);
      WINO_STAMP(3);
      end_barrier();
      WINO_STAMP(4);
    }
  }
  WINO_TRACE_DUMP();
  fold(af, ag, 3, 3, 0);
  fold(af, ag, 3, 3, 1);
  fold(af, ag, 3, 3, 2);

  // ---- epilogue: lane holds 4 consecutive channels of tile (lane&15): 2x2 output pixels
  const int q = lane >> 4;
  const int t = tile_base + tg * 16 + (lane & 15);
  if (t < p.total_tiles) {
    int b, y0, x0;
    tile_origin(t, b, y0, x0);
    // one 32-bit byte offset per lane (the launch guards the tensor's bytes below 2^31) + wave-uniform increments from a
    // scalar base: the stores need no 64-bit address arithmetic (v_mad_u64_u32 is quarter rate)
    const unsigned dst0 = (unsigned)((b * p.h + y0) * p.w + x0) * 384u + (unsigned)(3 * chh * 16 + q * 4) * 4u;
    const unsigned dinc_y = (unsigned)(p.d * p.w) * 384u, dinc_x = (unsigned)p.d * 384u;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int c0 = (3 * chh + j) * 16 + q * 4;
#pragma unroll
      for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int bb = 0; bb < 2; ++bb) {
          f32x4 vf = of[a][bb][j], vg = og[a][bb][j];      // (layer bias already inside)
          if (p.vbias) {      // folded vector source (launch_vecbias): a bias that depends on the pixel's border configuration
            const int y = y0 + a * p.d, x = x0 + bb * p.d;
            const int cfg = 3 * (y == 0 ? 0 : (y == p.h - 1 ? 2 : 1)) + (x == 0 ? 0 : (x == p.w - 1 ? 2 : 1));
            const float* tb = (const float*)((const char*)p.vbias + (unsigned)(((b * 9 + cfg) * 192 + c0) * 4));
            vf += *(const f32x4*)tb;
            vg += *(const f32x4*)(tb + 96);
          }
          f32x4 o;
#pragma unroll
          for (int e = 0; e < 4; ++e) o[e] = act_fast(vf[e], eluw) * sigmoid_fast(vg[e]);
          *(f32x4*)((char*)p.dst + (dst0 + (unsigned)(j * 64) + (a ? dinc_y : 0u) + (bb ? dinc_x : 0u))) = o;
        }
    }
  }
}

template <int NCHK>
static hipError_t launch_wino_t(const WinoParams& p, hipStream_t st) {
  constexpr int LDS = 3 * 64 * 128 + 4 * 192 * 128 + 8 * 512 * 4 + (WINO_TRACE_LDS ? 2 * 48 * 8 * 8 : 0);     // X ring 24 KB + W ring 96 KB + source offsets 16 KB
  {
    hipError_t e = ensure_max_lds((const void*)wino_kernel<NCHK>, LDS);
    if (e != hipSuccess) return e;
  }
  const int grid = (p.total_tiles + 63) / 64;
  set_launch_grid(grid);
  ProfScope ps_(st, PL_WINO_N192);
  hipLaunchKernelGGL(wino_kernel<NCHK>, dim3(grid), dim3(512), LDS, st, p);
  return hipGetLastError();
}

hipError_t launch_wino(const WinoParams& p, hipStream_t st) {
  return p.src1 ? launch_wino_t<6>(p, st) : launch_wino_t<3>(p, st);
}

}  // namespace se
