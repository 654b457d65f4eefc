// Narrow gated convolutions (24 or 12 gated output channels: the 5x5 first layers, the 48 -> 48 gen_deconv, the 24 -> 24
// layers -- all at full or half image resolution) in "raw tile" form, fp32 and bf16.
//
// On these layers the gather-GEMM of se_gconv.hip is held back by its staging, not by the MFMA pipe: its im2col image
// stages every input pixel once per tap (9x, 25x for the 5x5 layers) -- 16 KB of pixels + 6 KB of weights per 32-k chunk
// and 128 outputs against 1536 cycles of fp32 MFMAs, i.e. 28.6 B/clk for two resident workgroups, issued from inside the
// MFMA loop (which sustained about 19 B/clk; the DMA path alone does 33 - 57, tools/ubench/lds_fill_rate.hip): the pipe
// sat at 65 %.  Here
//   * the input tile of an 8 x 16 (fp32) / 32 x 16 (bf16) block of outputs (with its halo, every channel) is DMA'd into
//     LDS ONCE, pixel-major,
//     and the B fragments of every tap are read straight from it (a tap is an address offset);
//   * the layer's WHOLE packed weight image (25 - 43 KB for these shapes) is resident in LDS, loaded in the prologue:
//     the k loop has no staging, no vmcnt wait and no barrier at all.
// 28 - 66 KB staged per 128 outputs instead of 88 - 170 KB, and two workgroups per CU overlap one's prologue / epilogue
// with the other's MFMAs.  The weight image, the row order (MIXED: 8 features + their 8 gates per 16-row tile), the k
// order, the MFMA sequence and the epilogue are those of gconv_kernel, so the results are bit-identical to it.
//
// gen_deconv (nearest x2 + 3x3) runs in its sub-pixel form: class (py,px) is a 2x2 conv on the source grid whose raw tile
// is 9 x 17 source pixels; the four class workgroups of a tile are neighbours in one XCD's dispatch sequence (class_tile).
// Reference semantics: gen_conv / gen_deconv, /root/reference/models/networks/utils.py:9-51.
#include "se_device.h"

#include <cstdlib>
#include <type_traits>

namespace se {

// D4 (bf16, round 5): dense-K form of the 5x5 first layers whose stored NHWC8 input carries at most FOUR real channels (conv1
// of netM, the 4-channel wconv1, xconv1 / pmconv1).  With 8-channel granules K = 25 taps x 8 = 200 -> 4 chunks of 64, of
// which 75-100 k are real.  Here the raw tile holds 8-byte pixels -- the first four channels, staged one dword per DMA lane --
// so the 16 bytes at pixel x are the four channels of taps kx and kx + 1: a k-granule is a PAIR of horizontally adjacent taps,
// K = 5 rows x 3 pairs x 8 = 120 -> 2 chunks (the sixth tap of a row has zero weights: pack_layer16_d4), half the MFMAs and
// half the A-fragment reads.  A pixel is only 8-byte aligned, and a misaligned ds_read_b128 is replayed at 64 cycles
// (cdna_hip_programming.md, Guideline 17): a B fragment is two 8-byte reads.  Rows, MIXED order, epilogue: unchanged.
template <int NT, int PT, bool BF16, bool D4 = false>
__global__ __launch_bounds__(256, 2) void rtile_kernel(const RTileParams p) {
  static_assert(!D4 || BF16, "the pair-of-taps form exists in bf16 only (fp32: rtile_dense5_kernel)");
  constexpr int ES = BF16 ? 2 : 4;           // bytes per stored element
  constexpr int TR = 4 * PT;                 // a wave = PT rows of 16 output pixels, a workgroup = TR x 16 outputs
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Raw = smem;
  char* Wres = smem + p.raw_bytes;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int ntiles = p.B * p.ty * p.tx;
  int tile, cls = 0;
  if (p.up2) {
    if (!class_tile((int)blockIdx.x, ntiles, p.xcd, tile, cls)) return;
  } else {
    tile = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  }
  const int b = tile / (p.ty * p.tx), t2 = tile - b * (p.ty * p.tx);
  const int ty0 = (t2 / p.tx) * TR, tx0 = (t2 % p.tx) * 16;
  const int py = cls >> 1, px = cls & 1;
  const int pady = p.up2 ? 1 - py : p.pad, padx = p.up2 ? 1 - px : p.pad;
  const int pixb = p.C * ES;                 // bytes per source pixel
  const int cgp = pixb >> 4;                 // 16-byte granules per source pixel

  // ---- prologue: the whole weight image of this layer (class) and the raw tile, by LDS-DMA
  const unsigned lds_raw = lds_addr_of(Raw), lds_w = lds_addr_of(Wres);
  {
    const float* wsrc = p.wpk + (size_t)cls * p.nch * p.NP * 32;
    const int npieces = p.nch * p.NP / 8;    // 1 KB pieces (8 rows of 128 bytes)
    for (int i = w; i < npieces; i += 4) glds16_s(wsrc + (size_t)i * 256, (unsigned)lane * 16u, lds_w + i * 1024);
    const se_i32x4 rsrc = make_rsrc(p.src, (unsigned)p.B * p.Hin * p.Win * (unsigned)pixb);
    // The raw tile in groups of G whole rows (G = 1, or as many rows as fit 64 lanes when a row is short): a group is NI
    // pieces of LPI lanes, wave w issues piece w % NI of the groups w / NI, w / NI + 4 / NI, ...  Which row of the group,
    // column and granule a lane fetches is a per-lane constant (the two divisions below run once, not once per 64 slots),
    // the group's row arithmetic is scalar: ~7 VALU instructions per DMA instead of ~40.  The address arithmetic was up to
    // as much issue time as the layer's MFMAs (48 -> 48 upsampling in bf16), and ordinary VALU instructions exclude MFMAs
    // on the SIMD.  Same LDS image: slot s of row r at (r * RW * cgp + s) * 16.
    const int slots = p.RH * p.RW * cgp;
    const int rowslots = p.RW * cgp;
    const int G = rowslots <= 32 ? 64 / rowslots : 1, GL = G * rowslots, NI = (GL + 63) >> 6;
    if constexpr (D4) {
      // 8-byte pixels: dword q of the tile = (pixel q >> 1, half q & 1) <- bytes 4 (q & 1) .. + 3 of the 16-byte source pixel
      const int ndw = p.RH * p.RW * 2;
      for (int i = w; i * 64 < ndw; i += 4) {
        const int q = i * 64 + lane, pix = q >> 1;
        const int row = (int)udiv_magic((unsigned)pix, p.div_rw_m, p.div_rw_l), c = pix - row * p.RW;
        const int sy = ty0 - pady + row, sx = tx0 - padx + c;
        const bool ok = q < ndw && (unsigned)sy < (unsigned)p.Hin && (unsigned)sx < (unsigned)p.Win;
        const unsigned off = (unsigned)((b * p.Hin + sy) * p.Win + sx) * (unsigned)pixb + (unsigned)(q & 1) * 4u;
        bufdma4(ok ? off : 0x80000000u, rsrc, lds_raw + i * 256);      // outside the image: hardware zero fill
      }
    } else if (NI == 1 || NI == 2 || NI == 4) {
      const int sh = NI == 1 ? 0 : (NI == 2 ? 1 : 2);
      const int LPI = (GL + NI - 1) >> sh, kp = w & (NI - 1), g0 = w >> sh, gstep = 4 >> sh;
      const int sl = kp * LPI + lane;                                  // this lane's slot of a group
      const int pix = (int)udiv_magic((unsigned)sl, p.div_cg_m, p.div_cg_l), gs = sl - pix * cgp;
      const int rig = (int)udiv_magic((unsigned)pix, p.div_rw_m, p.div_rw_l), c = pix - rig * p.RW;      // row in the group, column
      const int sx = tx0 - padx + c;
      const bool mine = lane < LPI && sl < GL;
      const bool colok = mine && (unsigned)sx < (unsigned)p.Win;
      const unsigned lane_off = (unsigned)(rig * p.Win + sx) * (unsigned)pixb + (unsigned)gs * 16u;
      for (int row0 = g0 * G; row0 < p.RH; row0 += gstep * G) {       // wave-uniform
        const int sy0 = ty0 - pady + row0;
        const unsigned base = (unsigned)((b * p.Hin + sy0) * p.Win) * (unsigned)pixb;
        const bool ok = colok && (unsigned)(sy0 + rig) < (unsigned)p.Hin;
        if (mine && row0 + rig < p.RH)                                 // (rows past the tile would land in the weights)
          bufdma16(ok ? base + lane_off : 0x80000000u, rsrc, lds_raw + (row0 * rowslots + kp * LPI) * 16);      // outside the image: hardware zero fill
      }
      if (w == 0 && lane < (p.raw_bytes - slots * 16) / 16) bufdma16(0x80000000u, rsrc, lds_raw + slots * 16);     // the slack: zeros
    } else {
      for (int i = w; i * 64 < slots; i += 4) {
        const int q = i * 64 + lane;
        const int pix = (int)udiv_magic((unsigned)q, p.div_cg_m, p.div_cg_l), gs = q - pix * cgp;
        const int row = (int)udiv_magic((unsigned)pix, p.div_rw_m, p.div_rw_l), c = pix - row * p.RW;
        const int sy = ty0 - pady + row, sx = tx0 - padx + c;
        const bool ok = q < slots && (unsigned)sy < (unsigned)p.Hin && (unsigned)sx < (unsigned)p.Win;
        const unsigned off = (unsigned)((b * p.Hin + sy) * p.Win + sx) * (unsigned)pixb + (unsigned)gs * 16u;
        bufdma16(ok ? off : 0x80000000u, rsrc, lds_raw + i * 1024);      // outside the image: hardware zero fill
      }
    }
  }
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const int jx = lane & 15, g4 = lane >> 4;
  const int rowb = p.RW * (D4 ? 8 : pixb);                                // bytes per raw tile row
  const int xbase = ((PT * w) * p.RW + jx) * (D4 ? 8 : pixb);              // this lane's pixel of the wave's first row

  // accumulators start at the bias: the epilogue has no bias add (its ordinary VALU instructions cost MFMA slots)
  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 bq = *(const f32x4*)(p.bias + nt * 16 + g4 * 4);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = bq;
  }

  dma_wait_all();
  __syncthreads();

  // ---- k loop: no staging, no barrier.  Granule gi of the flattened (tap, channel group) axis -> raw tile offset.
  auto tap_off = [&](int gi) -> int {
    int tap = __umul24(gi, p.magicCG) >> 16;
    const int cg = gi - __umul24(tap, p.CG);
    tap = min(tap, p.T - 1);                                             // K padding: zero weights, any valid address
    const int ky = __umul24(tap, p.magicKW) >> 8, kx = tap - __umul24(ky, p.KW);
    return __mul24(ky, rowb) + __mul24(kx, pixb) + (cg << 4);
  };
  // No register software pipeline: a ping-pong version (fragments of chunk c + 1 read behind the MFMAs of chunk c,
  // pinned with sched_barriers) measured the same in fp32 and 6 % slower in bf16 -- the second resident workgroup
  // already covers the LDS latency, and the loop is MFMA-bound (fp32) or LDS-read-bound (bf16) either way.
  typedef typename std::conditional<BF16, bf16x8, f32x4>::type frag_t;
  for (int ch = 0; ch < p.nch; ++ch) {
    frag_t wq[2 * NT], xb[2][PT];
#pragma unroll
    for (int u = 0; u < 2 * NT; ++u)
      wq[u] = *(const frag_t*)(Wres + ch * (p.NP * 128) + (u % NT) * 2048 + (u / NT ? off1 : off0));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      if constexpr (D4) {
        // granule gi = 3 ky + j: the taps (ky, 2j) and (ky, 2j + 1); gi = 15 is chunk padding (zero weights: any valid address)
        const int gi = min(ch * 8 + half * 4 + g4, 14);
        const int ky = (gi * 11) >> 5, j = gi - 3 * ky;                    // gi / 3 for gi <= 14
        const int xo = xbase + ky * rowb + j * 16;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) {
          const u32x2 lo = *(const u32x2*)(Raw + xo + pt * rowb), hi = *(const u32x2*)(Raw + xo + pt * rowb + 8);
          xb[half][pt] = __builtin_bit_cast(frag_t, (u32x4){lo[0], lo[1], hi[0], hi[1]});
        }
      } else {
        const int xo = xbase + tap_off(ch * 8 + half * 4 + g4);
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xb[half][pt] = *(const frag_t*)(Raw + xo + pt * rowb);
      }
    }
#pragma unroll
    for (int u = 0; u < 2 * NT; ++u) {
      const int half = u / NT, nt = u % NT;
      if constexpr (BF16) {
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(wq[u], xb[half][pt], acc[nt][pt], 0, 0, 0);
      } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
            acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[u][r], xb[half][pt][r], acc[nt][pt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue (gconv_kernel, MIXED): tile rows 0-7 features (lanes 0-31), rows 8-15 their gates (lane + 32); two
  // v_permlane32_swap per quad hand every lane two complete (feature, gate) pairs.
  // In bf16 this epilogue -- 24 (row tile, pixel row) pairs per lane, two gated values each -- is as long as the wave's
  // whole MFMA loop, so it carries nothing that is not arithmetic: the activation switch is hoisted out of the loops
  // (one instantiation per activation), the store address is one 32-bit lane offset plus compile-time / wave-uniform
  // increments from a wave-uniform base, and the row / column / channel bounds are one lane mask and one scalar test.
  const int q = lane >> 4;
  const int lanec = (q & 1) * 4 + (q >> 1) * 2;                      // this lane's channel pair inside a row tile
  const int yw = ty0 + PT * w;                                       // first pixel row of this wave
  const int rowst = (p.up2 ? 2 : 1) * p.OW * p.G * ES;               // bytes between two pixel rows of this wave (uniform)
  char* dbase = (char*)p.dst + ((size_t)(b * p.OH + (p.up2 ? 2 * yw + py : yw)) * p.OW + (p.up2 ? 2 * tx0 + px : tx0)) * p.G * ES;
  const unsigned voff0 = (unsigned)(((p.up2 ? 2 * jx : jx) * p.G + lanec) * ES);
  const bool okx = tx0 + jx < p.Win;
  auto epilogue = [&](auto elu_tag) {
    constexpr bool ELU = decltype(elu_tag)::value;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const bool okc = okx && nt * 8 + lanec < p.G;
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const f32x4 v = acc[nt][pt];                                    // (bias already inside)
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
        const float f0 = __uint_as_float(s02[0]), gg0 = __uint_as_float(s02[1]);
        const float f1 = __uint_as_float(s13[0]), gg1 = __uint_as_float(s13[1]);
        float2 o;
        o.x = (ELU ? elu_fast(f0) : fmaxf(f0, 0.f)) * sigmoid_fast(gg0);
        o.y = (ELU ? elu_fast(f1) : fmaxf(f1, 0.f)) * sigmoid_fast(gg1);
        if (yw + pt < p.Hin && okc) {                                   // (row test wave-uniform)
          char* at = dbase + (size_t)pt * rowst + (voff0 + (unsigned)(nt * 8 * ES));
          if (BF16) *(unsigned*)at = pack_bf16x2(o.x, o.y);
          else *(float2*)at = o;
        }
      }
    }
  };
  if (p.act == 0) epilogue(std::true_type()); else epilogue(std::false_type());
}

// ---------------------------------------------------------------------------------------------------------------------
// Dense-K form of the 5x5 first layers whose inputs carry padding channels (fp32): conv1 of netG reads 5 real channels
// stored as NHWC8 (K = 25 taps x 8 = 200 -> 7 chunks of 32, 125 of them real), xconv1 / pmconv1 read 3 real channels
// stored as NHWC4 (K = 100 -> 4 chunks, 75 real).  These layers are bound by the fp32 MFMA pipe, so the padding is paid
// for in full.  Here the raw tile is staged DENSE -- [row][column][CD real channels], one dword per DMA lane, so a pixel
// outside the image is still exactly a set of out-of-range lanes (hardware zero fill) -- and k = tap * CD + channel
// has no channel padding at all: 125 -> 4 chunks, 75 -> 3 chunks; and exactly ceil(K / 4) MFMA k-steps are issued (the real k
// of the last chunk are packed into its first k-steps): 19 instead of 24 for 3 channels, 25 instead of 32 for the 4-channel
// layers (conv1 of netM, the 4-channel wconv1), which take this form for that reason alone.  A k-granule (4 consecutive k) is then no longer 16
// contiguous bytes of one pixel: its four elements are read with four ds_read_b32 whose tile offsets come from a table
// (built once per workgroup, one entry per k) -- two 16-byte table reads and eight address adds per chunk, against the
// 48 MFMAs of 32 cycles a chunk is.  Same weights, row order, MFMA sequence and epilogue as rtile_kernel; the summation
// order over k differs (channel padding removed), i.e. fp32 rounding only.
// ---------------------------------------------------------------------------------------------------------------------
template <int CD>
__global__ __launch_bounds__(256, 2) void rtile_dense5_kernel(const RTileParams p) {
  constexpr int NT = 3, PT = 2, TR = 8, KW = 5, T = 25;
  constexpr int RH = TR + KW - 1, RW = 16 + KW - 1;       // 12 x 20 source pixels
  constexpr int ROWB = RW * CD * 4;                        // bytes per dense tile row
  constexpr int NDW = RH * RW * CD;                        // dwords of the dense tile
  constexpr int K = T * CD, NCH = (K + 31) / 32;
  constexpr int REM = K - (NCH - 1) * 32;                  // real k of the last chunk
  constexpr int LAST_STEPS = (REM + 3) / 4;                // MFMA k-steps the last chunk needs (dense_kin packs them first)
  constexpr int RAWB = (NDW * 4 + 1023) & ~1023;
  constexpr int TABB = NCH * 32 * 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Raw = smem;
  int* Tab = (int*)(smem + RAWB);
  char* Wres = smem + RAWB + TABB;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = tile / (p.ty * p.tx), t2 = tile - b * (p.ty * p.tx);
  const int ty0 = (t2 / p.tx) * TR, tx0 = (t2 % p.tx) * 16;
  const int pixb = p.C * 4;                                // bytes per stored source pixel (padding channels included)

  const unsigned lds_raw = lds_addr_of(Raw), lds_w = lds_addr_of(Wres);
  {
    const int npieces = NCH * 48 / 8;                      // 1 KB pieces (8 rows of 128 bytes)
    for (int i = w; i < npieces; i += 4) glds16_s(p.wpk + (size_t)i * 256, (unsigned)lane * 16u, lds_w + i * 1024);
    const se_i32x4 rsrc = make_rsrc(p.src, (unsigned)p.B * p.Hin * p.Win * (unsigned)pixb);
    for (int i = w; i * 64 < NDW; i += 4) {
      const int q = i * 64 + lane;
      const int pix = q / CD, ch = q - pix * CD;           // compile-time divisors
      const int row = pix / RW, col = pix - row * RW;
      const int sy = ty0 - 2 + row, sx = tx0 - 2 + col;
      const bool ok = q < NDW && (unsigned)sy < (unsigned)p.Hin && (unsigned)sx < (unsigned)p.Win;
      const unsigned off = (unsigned)((b * p.Hin + sy) * p.Win + sx) * (unsigned)pixb + (unsigned)ch * 4u;
      bufdma4(ok ? off : 0x80000000u, rsrc, lds_raw + i * 256);
    }
    // chunk slot -> byte offset of its k's (tap, channel) inside the dense tile.  The k of a chunk are laid out
    // instruction-major (dense_kin, mirrored by pack_layer_dense): k-step s of the chunk multiplies k = 4 s + lane group.
    // The four lane groups of a ds_read_b32 then read four consecutive dwords of the dense tile per pixel -- with the
    // k-major order they read a stride of four, which with 4 channels per pixel put 64 lanes on 8 banks (50 % conflict
    // cycles, 32 % with 3 / 5 channels) -- and in the LAST chunk only ceil(REM / 4) of the 8 k-steps carry data, the
    // others are not issued at all; unused slots (zero weights, never multiplied) get any valid offset.
    if (tid < NCH * 32) {
      int kf = K - 1;
      {
        const int chk = tid >> 5, kin = tid & 31, half = kin >> 4, g = (kin >> 2) & 3, r = kin & 3;
        const int j = (half * 4 + r) * 4 + g;               // inverse of dense_kin (every chunk: see pack_layer_dense)
        if (chk < NCH - 1 || j < REM) kf = chk * 32 + j;
      }
      const int tap = kf / CD, ch = kf - tap * CD, ky = tap / KW, kx = tap - ky * KW;
      Tab[tid] = ky * ROWB + (kx * CD + ch) * 4;
    }
  }
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const int jx = lane & 15, g4 = lane >> 4;
  const int xbase = ((PT * w) * RW + jx) * CD * 4;          // this lane's pixel of the wave's first row

  f32x4 acc[NT][PT];                                        // start at the bias (rtile_kernel)
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const f32x4 bq = *(const f32x4*)(p.bias + nt * 16 + g4 * 4);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = bq;
  }

  dma_wait_all();
  __syncthreads();

#pragma unroll
  for (int ch = 0; ch < NCH; ++ch) {
    f32x4 wq[2 * NT], xb[2][PT];
#pragma unroll
    for (int u = 0; u < 2 * NT; ++u)
      wq[u] = *(const f32x4*)(Wres + ch * (48 * 128) + (u % NT) * 2048 + (u / NT ? off1 : off0));
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      typedef int i32x4 __attribute__((ext_vector_type(4)));
      const i32x4 to = *(const i32x4*)(Tab + ch * 32 + half * 16 + g4 * 4);
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const char* src = Raw + xbase + to[e];
#pragma unroll
        for (int pt = 0; pt < PT; ++pt) xb[half][pt][e] = *(const float*)(src + pt * ROWB);
      }
    }
#pragma unroll
    for (int u = 0; u < 2 * NT; ++u) {
      const int half = u / NT, nt = u % NT;
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if (ch == NCH - 1 && half * 4 + r >= LAST_STEPS) continue;      // compile-time: k-steps of pure chunk padding
#pragma unroll
        for (int pt = 0; pt < PT; ++pt)
          acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[u][r], xb[half][pt][r], acc[nt][pt], 0, 0, 0);
      }
    }
  }

  // ---- epilogue: rtile_kernel's (MIXED rows, two v_permlane32_swap per quad)
  const int q = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c0 = nt * 8 + (q & 1) * 4 + (q >> 1) * 2;
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int yy = ty0 + PT * w + pt, xx = tx0 + jx;
      const f32x4 v = acc[nt][pt];                          // (bias already inside)
      const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
      const float f0 = __uint_as_float(s02[0]), gg0 = __uint_as_float(s02[1]);
      const float f1 = __uint_as_float(s13[0]), gg1 = __uint_as_float(s13[1]);
      float2 o;
      o.x = act_fast(f0, eluw) * sigmoid_fast(gg0);
      o.y = act_fast(f1, eluw) * sigmoid_fast(gg1);
      if (c0 < p.G && yy < p.Hin && xx < p.Win)
        *(float2*)(p.dst + ((size_t)(b * p.Hin + yy) * p.Win + xx) * p.G + c0) = o;
    }
  }
}

// ---------------------------------------------------------------------------------------------------------------------
// The same 5x5 first layers with a ONE-DIMENSIONAL Winograd transform, F(2,5) along x (round 4; even widths).  These
// kernels are bound by the fp32 MFMA pipe (with every second k-step removed a launch is 34 % shorter), so the multiply-adds
// themselves go: 6 (position, kernel row) products per 2 outputs and kernel row instead of 10.
//   out[y][2t + o] = sum_ky sum_c A^T[o][nu] ( U[nu][ky][c] * V[y + ky - 2][t][nu][c] ),   V = B^T x[r][2t - 2 .. 2t + 3]
//   B^T = [4 0 -5 0 1 0; 0 -4 -4 1 1 0; 0 4 -4 -1 1 0; 0 -2 -1 2 1 0; 0 2 -1 -2 1 0; 0 4 0 -5 0 1]   (that of F(4,3))
//   G   = [1/4 0 0 0 0; -1/6 (1 1 1 1 1); -1/6 (1 -1 1 -1 1); 1/24 (1 2 4 8 16); 1/24 (1 -2 4 -8 16); 0 0 0 0 1]
//   A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 1]
// The dense raw tile is staged as before; a transform pass (LDS -> LDS, <= 2 tasks of 6 reads, 12 fmas, 6 writes per thread)
// builds T[row][nu][x-tile][CD], and the k loop of a position nu is the dense kernel's with K = 5 CD in ONE chunk
// (k = ky * CD + c, instruction-major, exactly ceil(K / 4) k-steps) and the 16 MFMA columns = 2 rows x 8 x-tiles.
// MFMAs per wave and 8 x 16 block: 72 / 90 / 126 for 3 / 4 / 5 channels (dense direct: 114 / 150 / 192).
// Measured (256x256 B=32, same box): gconv_n48 1.85 -> 1.76 ms per step.  A PERSISTENT form (two workgroups per CU, weights
// resident once, the raw tile of block i+1 DMA'd under the MFMAs of block i) measured the same 1.754 ms and was dropped:
// with two non-persistent workgroups per CU the other workgroup's MFMAs already cover this one's prologue.  Nor did a
// WAVE-SPECIALISED persistent form (four MFMA waves: k loop + epilogue; two producer waves: raw-tile DMA two blocks ahead with
// counted waits, transform one block ahead into a second T buffer; parity green): 1.752 against 1.755 ms, same box, three
// alternations.  The kernel is not waiting for anything a second workgroup cannot cover: per block pair and SIMD it issues
// ~4.6k cycles of MFMAs, ~2.6k of other VALU work (transform, the A^T folds: 120 instructions per block, epilogue) that the
// MFMAs exclude, and ~3.7k cycles' worth of LDS traffic (fragment reads, 32 % bank conflicts on the dword gathers) in 10.6k.
// ---------------------------------------------------------------------------------------------------------------------
// Layout of T (round 5; VERDICT r4 item 2: 24-31 % of this kernel's LDS cycles were bank-conflict cycles).  The B operand of a
// k-step is four ds_read_b32 per lane; the 32 lanes of a lane group (two tile rows rl x eight x-tiles xt x two k lane groups)
// read dword (2w + rl + ky) * TROW + nu * TNU + xt * XS + c, and with the dense strides (XS = CD, TROW = 48 CD dwords: a
// multiple of 16) the two tile rows -- and for CD = 4 every pair of lanes -- fell on the same banks: 60 / 80 / 104 LDS cycles
// per wave and position loop where 32 / 40 / 56 are conflict-free.  Padded strides found by exhaustive search over (x-tile
// stride, pad per nu plane, pad per row) with the ds_read_b32 bank model of MI355X_MICROARCH.md (32 banks per 32-lane group):
//   CD = 3: x-tile stride 4 dwords, row pad 2 -> 40 cycles;   CD = 4: row pad 2 -> 40 (conflict-free);
//   CD = 5: x-tile stride 6, nu-plane pad 2, row pad 4 -> 64 -- NOT used: its 3 KB of padding take the workgroup from 53.6 to
//   56.7 KB, i.e. from THREE to two resident workgroups per CU, and the launch from 203 to 241 us (measured, same box); no
//   layout within the 54.6 KB that three workgroups allow removes a conflict (x-tile stride 5 only), so CD = 5 stays dense.
// Measured (256 x 256 B=32, same box, SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE): CD = 4 24.0 -> 6.8 %, CD = 3 30.9 -> 20.2 % of
// the LDS cycles; launch times unchanged (154.5 / 147 us): as DESIGN.md 7b had costed, the conflicts are not what binds these
// kernels (MFMA + excluded VALU + LDS issue together do).  (ds_write_b32 of the transform pass: a 2-way conflict costs a store
// nothing, same guide.)
template <int CD> struct D5WLayout {
  static constexpr int XS = CD == 3 ? 4 : CD;                        // dwords between x-tiles
  static constexpr int NUPAD = 0;                                    // dwords behind the eight x-tiles of one (row, nu)
  static constexpr int ROWPAD = CD == 5 ? 0 : 2;                     // dwords behind the six nu planes of one row
  static constexpr int TNU = (8 * XS + NUPAD) * 4;                   // bytes of one (row, nu) plane
  static constexpr int TROW = 6 * TNU + ROWPAD * 4;                  // bytes of one source row of T
  static constexpr int TBYTES = (12 * TROW + 255) & ~255;
};
template <int CD>
__global__ __launch_bounds__(256, 2) void rtile_dense5w_kernel(const RTileParams p) {
  constexpr int NT = 3, TR = 8, KH = 5;
  constexpr int RH = TR + KH - 1, RW = 20;                 // raw tile: 12 x 20 source pixels (columns tx0 - 2 .. tx0 + 17)
  constexpr int NDW = RH * RW * CD;
  constexpr int K = KH * CD, KS = (K + 3) / 4;             // k per position, MFMA k-steps per position
  constexpr int XS = D5WLayout<CD>::XS;
  constexpr int TNU = D5WLayout<CD>::TNU;                  // bytes of the eight x-tiles of one (row, nu)
  constexpr int TROW = D5WLayout<CD>::TROW;                // bytes of one source row of T
  constexpr int RAWB = (NDW * 4 + 1023) & ~1023;
  constexpr int TBYTES = D5WLayout<CD>::TBYTES;
  static_assert(RH == 12, "D5WLayout::TBYTES assumes 12 source rows");
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Raw = smem;
  char* T = smem + RAWB;
  int* Tab = (int*)(smem + RAWB + TBYTES);                 // [32] byte offset inside T of the (kernel row, channel) of chunk slot kin
  char* Wres = smem + RAWB + TBYTES + 128;

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int tile = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x;
  const int b = tile / (p.ty * p.tx), t2 = tile - b * (p.ty * p.tx);
  const int ty0 = (t2 / p.tx) * TR, tx0 = (t2 % p.tx) * 16;
  const int pixb = p.C * 4;

  const unsigned lds_raw = lds_addr_of(Raw), lds_w = lds_addr_of(Wres);
  {
    for (int i = w; i < 6 * 48 / 8; i += 4) glds16_s(p.wpk + (size_t)i * 256, (unsigned)lane * 16u, lds_w + i * 1024);
    const se_i32x4 rsrc = make_rsrc(p.src, (unsigned)p.B * p.Hin * p.Win * (unsigned)pixb);
    for (int i = w; i * 64 < NDW; i += 4) {
      const int q = i * 64 + lane;
      const int pix = q / CD, ch = q - pix * CD;
      const int row = pix / RW, col = pix - row * RW;
      const int sy = ty0 - 2 + row, sx = tx0 - 2 + col;
      const bool ok = q < NDW && (unsigned)sy < (unsigned)p.Hin && (unsigned)sx < (unsigned)p.Win;
      const unsigned off = (unsigned)((b * p.Hin + sy) * p.Win + sx) * (unsigned)pixb + (unsigned)ch * 4u;
      bufdma4(ok ? off : 0x80000000u, rsrc, lds_raw + i * 256);
    }
    if (tid < 32) {      // chunk slot -> (kernel row, channel): inverse of dense_kin (pack_layer_dense5w); padding slots: any valid offset
      const int half = tid >> 4, g = (tid >> 2) & 3, r = tid & 3;
      int j = (half * 4 + r) * 4 + g;
      if (j >= K) j = K - 1;
      const int ky = j / CD, c = j - ky * CD;
      Tab[tid] = ky * TROW + c * 4;
    }
  }
  dma_wait_all();
  __syncthreads();
  // ---- transform pass: task = (row, x-tile, channel); x-tile t reads raw columns 2t .. 2t + 5 (= source tx0 + 2t - 2 .. + 3)
  for (int k = tid; k < RH * 8 * CD; k += 256) {
    const int row = k / (8 * CD), rem = k - row * (8 * CD), xt = rem / CD, c = rem - xt * CD;
    const float* src = (const float*)(Raw + ((row * RW + 2 * xt) * CD + c) * 4);
    const float d0 = src[0], d1 = src[CD], d2 = src[2 * CD], d3 = src[3 * CD], d4 = src[4 * CD], d5 = src[5 * CD];
    const float pp = fmaf(-4.f, d2, d4), qq = fmaf(-4.f, d1, d3), rr = d4 - d2, tt = d3 - d1;
    float* dst = (float*)(T + row * TROW + xt * XS * 4 + c * 4);
    dst[0] = fmaf(4.f, d0, fmaf(-5.f, d2, d4));
    dst[TNU / 4] = pp + qq;
    dst[2 * TNU / 4] = pp - qq;
    dst[3 * TNU / 4] = fmaf(2.f, tt, rr);
    dst[4 * TNU / 4] = fmaf(-2.f, tt, rr);
    dst[5 * TNU / 4] = fmaf(4.f, d1, fmaf(-5.f, d3, d5));
  }
  __syncthreads();

  int off0, off1;
  frag_offsets(lane, off0, off1);
  const int jx = lane & 15, g4 = lane >> 4;
  const int rl = jx >> 3, xt = jx & 7;
  const int xbase = (2 * w + rl) * TROW + xt * XS * 4;
  // this lane group's four (k-half 0) + four (k-half 1) element offsets
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  const i32x4 to0 = *(const i32x4*)(Tab + g4 * 4), to1 = *(const i32x4*)(Tab + 16 + g4 * 4);
  f32x4 oy[2][NT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) oy[0][nt] = oy[1][nt] = *(const f32x4*)(p.bias + nt * 16 + g4 * 4);
  float neg1 = -1.f, two = 2.f, neg2 = -2.f;
  asm volatile("" : "+v"(neg1), "+v"(two), "+v"(neg2));
#pragma unroll
  for (int nu = 0; nu < 6; ++nu) {
    f32x4 wq[2][NT], xb[2];
#pragma unroll
    for (int half = 0; half < (KS > 4 ? 2 : 1); ++half) {
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) wq[half][nt] = *(const f32x4*)(Wres + nu * (48 * 128) + nt * 2048 + (half ? off1 : off0));
#pragma unroll
      for (int e = 0; e < 4; ++e) xb[half][e] = *(const float*)(T + xbase + nu * TNU + (half ? to1[e] : to0[e]));
    }
    f32x4 am[NT];
#pragma unroll
    for (int s_ = 0; s_ < KS; ++s_)
#pragma unroll
      for (int nt = 0; nt < NT; ++nt) {
        const f32x4 cin = s_ == 0 ? (f32x4){0.f, 0.f, 0.f, 0.f} : am[nt];
        am[nt] = __builtin_amdgcn_mfma_f32_16x16x4f32(wq[s_ / 4][nt][s_ % 4], xb[s_ / 4][s_ % 4], cin, 0, 0, 0);
      }
    // A^T = [1 1 1 1 1 0; 0 1 -1 2 -2 1]
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      if (nu < 5) oy[0][nt] += am[nt];
      if (nu == 1 || nu == 5) oy[1][nt] += am[nt];
      if (nu == 2) oy[1][nt] = am[nt] * neg1 + oy[1][nt];
      if (nu == 3) oy[1][nt] = am[nt] * two + oy[1][nt];
      if (nu == 4) oy[1][nt] = am[nt] * neg2 + oy[1][nt];
    }
  }

  // ---- epilogue (MIXED rows, two v_permlane32_swap per quad): lane = (row 2w + rl, x-tile xt), outputs x = 2 xt, 2 xt + 1
  const int q = lane >> 4;
  const int yy = ty0 + 2 * w + rl;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int c0 = nt * 8 + (q & 1) * 4 + (q >> 1) * 2;
#pragma unroll
    for (int o = 0; o < 2; ++o) {
      const int xx = tx0 + 2 * xt + o;
      const f32x4 v = oy[o][nt];
      const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
      const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
      const float f0 = __uint_as_float(s02[0]), gg0 = __uint_as_float(s02[1]);
      const float f1 = __uint_as_float(s13[0]), gg1 = __uint_as_float(s13[1]);
      float2 ov;
      ov.x = act_fast(f0, eluw) * sigmoid_fast(gg0);
      ov.y = act_fast(f1, eluw) * sigmoid_fast(gg1);
      if (c0 < p.G && yy < p.Hin && xx < p.Win)
        *(float2*)(p.dst + ((size_t)(b * p.Hin + yy) * p.Win + xx) * p.G + c0) = ov;
    }
  }
}

template <int CD>
static hipError_t launch_rtile_dense5w(const RTileParams& p, hipStream_t st) {
  constexpr int NDW = 12 * 20 * CD;
  constexpr int LDS = ((NDW * 4 + 1023) & ~1023) + D5WLayout<CD>::TBYTES + 128 + 6 * 48 * 128;
  static_assert(LDS <= 80 * 1024, "two workgroups per CU");
  hipError_t e = ensure_max_lds((const void*)rtile_dense5w_kernel<CD>, 80 * 1024);
  if (e != hipSuccess) return e;
  const int tiles = p.B * p.ty * p.tx;
  set_launch_grid(tiles);
  ProfScope ps_(st, PL_GCONV_N48);
  hipLaunchKernelGGL((rtile_dense5w_kernel<CD>), dim3(tiles), dim3(256), LDS, st, p);
  return hipGetLastError();
}

template <int CD>
static hipError_t launch_rtile_dense5(const RTileParams& p, hipStream_t st) {
  constexpr int NDW = 12 * 20 * CD, K = 25 * CD, NCH = (K + 31) / 32;
  constexpr int LDS = ((NDW * 4 + 1023) & ~1023) + NCH * 32 * 4 + NCH * 48 * 128;
  hipError_t e = ensure_max_lds((const void*)rtile_dense5_kernel<CD>, 80 * 1024);
  if (e != hipSuccess) return e;
  const int tiles = p.B * p.ty * p.tx;
  set_launch_grid(tiles);
  ProfScope ps_(st, PL_GCONV_N48);
  hipLaunchKernelGGL((rtile_dense5_kernel<CD>), dim3(tiles), dim3(256), LDS, st, p);
  return hipGetLastError();
}

template <int NT, int PT, bool BF16, bool D4 = false>
static hipError_t launch_rtile_t(const RTileParams& p, hipStream_t st, int label) {
  const int lds = p.raw_bytes + p.nch * p.NP * 128;
  {
    hipError_t e = ensure_max_lds((const void*)rtile_kernel<NT, PT, BF16, D4>, 80 * 1024);
    if (e != hipSuccess) return e;
  }
  const int tiles = p.B * p.ty * p.tx;
  const int grid = p.up2 ? class_tile_grid(tiles) : tiles;
  set_launch_grid(grid);
  ProfScope ps_(st, label);
  hipLaunchKernelGGL((rtile_kernel<NT, PT, BF16, D4>), dim3(grid), dim3(256), lds, st, p);
  return hipGetLastError();
}

// Tile rows per workgroup.  fp32: 8 (a wave = 2 rows: 5 fragment reads per 12 MFMAs of 32 cycles).  bf16: 32 (a wave = 8 rows
// = 128 pixels): a bf16 MFMA is 16 cycles for twice the k, and with 2-row waves the LDS delivers the fragments at a
// quarter of the rate the pipe could take them (10 reads per 12 MFMAs; measured 22 % MFMA-busy) -- 11 reads per 24 now.
int rtile_rows(bool bf16) { return bf16 ? 32 : 8; }

hipError_t launch_rtile(const RTileParams& p, hipStream_t st) {
  if (p.dense == 103) return launch_rtile_dense5w<3>(p, st);      // dense + 100: the F(2,5)-along-x form (wpk = image of pack_layer_dense5w)
  if (p.dense == 104) return launch_rtile_dense5w<4>(p, st);
  if (p.dense == 105) return launch_rtile_dense5w<5>(p, st);
  if (p.dense == 3) return launch_rtile_dense5<3>(p, st);
  if (p.dense == 4) return launch_rtile_dense5<4>(p, st);
  if (p.dense == 5) return launch_rtile_dense5<5>(p, st);
  if (p.dense == 204) return (p.bf16 && p.NP == 48) ? launch_rtile_t<3, 8, true, true>(p, st, PL_GCONV_N48) : hipErrorInvalidValue;   // bf16 pair-of-taps form
  if (p.dense) return hipErrorInvalidValue;
  if (p.NP == 48) return p.bf16 ? launch_rtile_t<3, 8, true>(p, st, PL_GCONV_N48) : launch_rtile_t<3, 2, false>(p, st, PL_GCONV_N48);
  if (p.NP == 32) return p.bf16 ? launch_rtile_t<2, 8, true>(p, st, PL_GCONV_N24) : launch_rtile_t<2, 2, false>(p, st, PL_GCONV_N24);
  return hipErrorInvalidValue;
}

}  // namespace se
