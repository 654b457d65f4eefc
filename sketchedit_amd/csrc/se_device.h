// Device-side helpers shared by the gfx950 kernels: LDS-DMA issue, the 32-k MFMA chunk core,
// activation math, and the host-side launch profiler scope.
#pragma once
#include "se_kernels.h"

namespace se {

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
#define DEVFN __device__ __forceinline__

// ---- LDS-DMA ------------------------------------------------------------------------------------
// global_load_lds_dwordx4: lane i of the wave writes 16 B at lds_addr + 16*i (lds_addr wave-uniform,
// in M0).  Issued through inline asm on purpose: hipcc treats a builtin LDS-DMA as a pending LDS
// write and drains vmcnt(0) in front of every later ds_read (measured: it serialised all staging
// loads of a chunk with full memory latency).  Hidden from its bookkeeping, the loads of chunk c+1
// stay in flight under the MFMAs of chunk c; completion is enforced by dma_wait_all() + a barrier
// before the tile is read.
DEVFN void glds16(const void* gsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %2\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, off\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(gsrc), "s"(lds_addr)
      : "memory");
}
// same, scalar base + 32-bit lane offset addressing (no VALU address arithmetic at the call site)
DEVFN void glds16_s(const void* sbase, unsigned voff, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "global_load_lds_dwordx4 %1, %2\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(sbase), "s"(lds_addr)
      : "memory");
}
// LDS-DMA through a buffer resource: lanes whose byte offset is outside [0, num_records) deliver ZEROS to LDS
// (hardware range check), which is how zero padding is staged without a zero page or a pointer select.
typedef int se_i32x4 __attribute__((ext_vector_type(4)));
DEVFN se_i32x4 make_rsrc(const void* base, unsigned bytes) {
  se_i32x4 r;
  r[0] = __builtin_amdgcn_readfirstlane((int)(size_t)base);
  r[1] = __builtin_amdgcn_readfirstlane((int)((size_t)base >> 32));     // stride 0: raw buffer
  r[2] = __builtin_amdgcn_readfirstlane((int)bytes);
  r[3] = 0x00020000;                                                    // 32-bit data format (gfx9 family)
  return r;
}
DEVFN void bufdma16(unsigned voff, se_i32x4 rsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dwordx4 %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}
// one dword per lane: lane i writes 4 B at lds_addr + 4*i (used where a 16-byte piece would straddle a pixel boundary)
DEVFN void bufdma4(unsigned voff, se_i32x4 rsrc, unsigned lds_addr) {
  unsigned keep;
  asm volatile(
      "s_mov_b32 %0, m0\n\t"
      "s_mov_b32 m0, %3\n\t"
      "s_nop 0\n\t"
      "buffer_load_dword %1, %2, 0 offen lds\n\t"
      "s_mov_b32 m0, %0"
      : "=&s"(keep)
      : "v"(voff), "s"(rsrc), "s"(lds_addr)
      : "memory");
}
DEVFN void dma_wait_all() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

DEVFN unsigned lds_addr_of(const void* p) {
  return (unsigned)(size_t)(const __attribute__((address_space(3))) char*)p;
}

DEVFN float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }
// epilogue forms on the hardware transcendentals (v_exp_f32 / v_rcp_f32, ~1-2 ulp): absolute error ~1e-7,
// three orders below the 1e-3 parity budget
DEVFN float fast_exp(float x) { return __expf(x); }
DEVFN float fast_rcp(float x) { return __builtin_amdgcn_rcpf(x); }
// branch-free on purpose: written as `x > 0 ? x : exp(x) - 1`, hipcc guards the exponential with a per-VALUE branch
// (s_cbranch_execnz: "does any lane need it?") -- 48 branches in a 48-value epilogue
DEVFN float elu_fast(float x) {
  const float e = __builtin_amdgcn_exp2f(fminf(x, 0.f) * 1.44269504088896340736f) - 1.f;
  return x > 0.f ? x : e;
}
DEVFN float sigmoid_fast(float x) { return fast_rcp(1.f + fast_exp(-x)); }
// ELU / ReLU of the gated epilogues without a branch or a select on the activation type:
//   act(x) = max(x, w (e^min(x, 0) - 1)),   wave-uniform weight w = 1 (ELU) or 0 (ReLU).
// x >= 0: the second operand is 0 <= x.  x < 0: e^x - 1 > x, so the maximum is the ELU branch (where rounding puts the
// computed e^x - 1 a hair below x -- |x| < 3e-4 -- the result is x, inside the error of the exponential itself).
// Four ordinary VALU instructions (mul, med3, fma, med3) + v_exp_f32; the ordinary ones are what an MFMA kernel pays for
// (they exclude MFMAs on the SIMD, the transcendental does not).  The former max(x,0) + w (e^min(x,0) - 1)
// was five.  (`p.act == 0 ? elu_fast(f) : fmaxf(f, 0.f)` compiled to two scalar branches PER VALUE in the 48-value bf16 epilogues.)
DEVFN float act_fast(float x, float eluw) {
  const float n = __builtin_amdgcn_fmed3f(x * 1.44269504088896340736f, -__builtin_inff(), 0.f);      // min(x log2 e, 0)
  const float neg = fmaf(__builtin_amdgcn_exp2f(n), eluw, -eluw);                                   // w (e^min(x,0) - 1)
  return __builtin_amdgcn_fmed3f(x, neg, __builtin_inff());                                         // max(x, neg)
}
// tanh(x) = 1 - 2 / (1 + e^(2x)): saturates correctly (e^(2x) -> inf gives 1, -> 0 gives -1); absolute error ~2e-7 -- ocml's
// tanhf costs ~40 VALU instructions, a third of small_conv_kernel<3>'s multiply-add work
DEVFN float tanh_fast(float x) { return 1.f - 2.f * fast_rcp(1.f + fast_exp(2.f * x)); }

// ---- division by a launch-invariant divisor (Granlund-Montgomery, round-up form): exact for every 32-bit x.
// Host (se_kernels.h udiv_magic_host): l = ceil(log2 d), m = floor(2^32 * (2^l - d) / d) + 1.  Device: 2 shifts, 1 mul_hi, 2 adds instead of the
// ~30-instruction software division hipcc emits for a runtime divisor.
DEVFN unsigned udiv_magic(unsigned x, unsigned m, int l) {
  const unsigned t = __umulhi(m, x);
  return (t + ((x - t) >> (l < 1 ? l : 1))) >> (l > 1 ? l - 1 : 0);
}

// ---- XCD-aware tile order ---------------------------------------------------------------------------
// The dispatcher places workgroup b on XCD b % 8 (observed, MI355X_MICROARCH.md); each XCD has its own L2.
// Remap so that every XCD walks one CONTIGUOUS eighth of the tile list: vertically adjacent tiles (which
// re-read the same input rows through the 3x3 taps) then share an L2.  Bijective for any grid size; only
// speed depends on the placement assumption, never correctness.
DEVFN int xcd_tile(int bid, int nblocks) {
  const int q = nblocks >> 3, r = nblocks & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Sub-pixel kernels (gen_deconv: 4 output parity classes over the same source tiles).  With the class in blockIdx.y the
// dispatch is class-major and every class streams the whole source from HBM again (measured: 6.6x the algorithmic
// bytes in winoup_kernel).  Here the grid is 1-D, 4 * round_up(ntiles, 8) blocks, and the four class blocks of a tile
// are consecutive entries of ONE XCD's dispatch sequence (block ids b, b+8, b+16, b+24): they start together on four
// CUs that share an L2, so the source tile comes from HBM once.  Returns false for the padding blocks.
DEVFN bool class_tile(int bid, int ntiles, int xcd_remap, int& tile, int& cls) {
  const int xcd = bid & 7, g = bid >> 3;
  cls = g & 3;
  const int b2 = ((g >> 2) << 3) | xcd;
  if (b2 >= ntiles) return false;
  tile = xcd_remap ? xcd_tile(b2, ntiles) : b2;
  return true;
}
static inline int class_tile_grid(int ntiles) { return 4 * ((ntiles + 7) & ~7); }

// ---- MFMA core ------------------------------------------------------------------------------------
// One 32-k chunk for a wave tile of NT (rows: packed channels) x PT (cols: pixels) 16x16 tiles.
// Wt: [NT*16][128 B] tile, Xt: this wave's [PT*16][128 B] tile, both with the 16-B slot of row r
// stored at (slot ^ ((r>>1)&7)).  off0/off1 = per-lane byte offsets for k-half 0/1.
// v_mfma_f32_16x16x4_f32: A[i=lane&15][k=lane>>4], B[k=lane>>4][j=lane&15], D[i=(lane>>4)*4+reg][j=lane&15].
struct NoHook { DEVFN void operator()(int) const {} };
// hook(slot), slot = 0 .. 2*NT-1, is called after the MFMAs of every (k-half, channel tile) step: the caller issues the
// staging DMA of the next chunk there, one piece per step, instead of as one burst in front of the MFMAs.
template <int NT, int PT, typename Hook = NoHook>
DEVFN void mfma_chunk(f32x4 (&acc)[NT][PT], const char* __restrict__ Wt, const char* __restrict__ Xt, int off0,
                      int off1, Hook hook = Hook()) {
  // all B (pixel) fragments of the chunk up front, the A (channel) fragments one tile ahead -- also across the two
  // k-halves, so no LDS latency sits between the halves
  f32x4 xb[2][PT];
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xb[half][pt] = *(const f32x4*)(Xt + pt * 2048 + (half ? off1 : off0));
  f32x4 wn = *(const f32x4*)(Wt + off0);
#pragma unroll
  for (int half = 0; half < 2; ++half) {
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 wa = wn;
      if (nt + 1 < NT) wn = *(const f32x4*)(Wt + (nt + 1) * 2048 + (half ? off1 : off0));
      else if (half == 0) wn = *(const f32x4*)(Wt + off1);
      // k-step outer, pixel tile inner: consecutive MFMAs never share an accumulator (a dependent
      // v_mfma_f32_16x16x4_f32 issues after 40 cycles instead of 32 when the wave has the pipe to itself)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wa[r], xb[half][pt][r], acc[nt][pt], 0, 0, 0);
      }
      hook(half * NT + nt);
    }
  }
}

// ---- bf16 storage (BASELINE config 5: bf16 operands, fp32 accumulate) ------------------------------------------------
// Activations and weights are stored as bf16; a 16-byte granule holds 8 channels, a 128-byte LDS row 64 k-values, and
// the tile / swizzle / fragment-offset geometry is the fp32 one: lane l of v_mfma_f32_16x16x32_bf16 supplies the 8
// consecutive k-values of k-group (l >> 4) for row / column (l & 15), i.e. exactly one 16-byte slot, and the C/D layout
// is the same as for the fp32 instruction.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
// two fp32 -> one dword of two bf16 (round to nearest even), lo in bits 0-15
DEVFN unsigned pack_bf16x2(float lo, float hi) {
  unsigned r;
  asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
  return r;
}
DEVFN float bf16_lo(unsigned u) { return __uint_as_float(u << 16); }
DEVFN float bf16_hi(unsigned u) { return __uint_as_float(u & 0xffff0000u); }

// One 64-k chunk of bf16 operands: same tile addressing as mfma_chunk, one MFMA per (k-half, channel tile, pixel tile).
// A bf16 MFMA is short (~17 cycles per SIMD), so an A fragment feeds only PT * 17 cycles of matrix work: with the fp32
// core's one-fragment look-ahead the LDS latency (~100+ cycles under load) was exposed on every fragment and the pipe
// sat at ~40 cycles per MFMA.  Here a window of DEPTH fragments is kept in flight across the 2 * NT (k-half, tile)
// steps of the chunk.
template <int NT, int PT, typename Hook = NoHook>
DEVFN void mfma_chunk16(f32x4 (&acc)[NT][PT], const char* __restrict__ Wt, const char* __restrict__ Xt, int off0,
                        int off1, Hook hook = Hook()) {
  constexpr int NU = 2 * NT, DEPTH = NU < 6 ? NU : 6;
  bf16x8 xb[2][PT];
#pragma unroll
  for (int half = 0; half < 2; ++half)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) xb[half][pt] = *(const bf16x8*)(Xt + pt * 2048 + (half ? off1 : off0));
  bf16x8 wq[DEPTH];
#pragma unroll
  for (int u = 0; u < DEPTH; ++u) wq[u] = *(const bf16x8*)(Wt + (u % NT) * 2048 + (u / NT ? off1 : off0));
  __builtin_amdgcn_sched_barrier(0);
#pragma unroll
  for (int u = 0; u < NU; ++u) {
    const int half = u / NT, nt = u % NT;
    const bf16x8 wa = wq[u % DEPTH];
    if (u + DEPTH < NU) wq[u % DEPTH] = *(const bf16x8*)(Wt + ((u + DEPTH) % NT) * 2048 + ((u + DEPTH) / NT ? off1 : off0));
    // pin the order: left to itself hipcc sinks every fragment read to just in front of its two MFMAs and waits
    // lgkmcnt(0) there (read - wait - 2 MFMAs, ~40 cycles per MFMA); with the order fixed it emits counted waits
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt)
      acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wa, xb[half][pt], acc[nt][pt], 0, 0, 0);
    hook(u);
    __builtin_amdgcn_sched_barrier(0);
  }
}

// per-lane fragment read offsets (see mfma_chunk)
DEVFN void frag_offsets(int lane, int& off0, int& off1) {
  const int swz = (lane >> 1) & 7;
  off0 = (lane & 15) * 128 + (((lane >> 4) ^ swz) << 4);
  off1 = (lane & 15) * 128 + (((4 + (lane >> 4)) ^ swz) << 4);
}

// ---- host: launch profiler scope -------------------------------------------------------------------
struct ProfScope {
  hipStream_t st;
  int idx = -1;
  ProfScope(hipStream_t s, int label);
  ~ProfScope();
};

}  // namespace se
