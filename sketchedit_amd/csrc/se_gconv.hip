// Gated convolution as a gather-GEMM on the fp32 MFMA (gfx950).
//
//   D[n][p] = sum_k Wp[n][k] * X[p][k]     n: packed output channel, p: output pixel,
//                                           k: flattened (tap, input channel), chunks of 32
//   out[p][c] = act(D[feat c][p] + b) * sigmoid(D[gate c][p] + b)     (models/networks/utils.py:21-33)
//
// A workgroup (4 waves, one per SIMD) owns PT*64 consecutive output pixels (NHWC order) and ALL
// output channels, so the gate is a register-level epilogue.  Per chunk it stages the [pixels][32]
// activation tile (gathered: every 16-byte granule is one tap's 4 channels of one input pixel;
// stride, dilation, nearest-x2 upsampling and zero padding are pure address arithmetic -- padding
// reads come from a zero page) and the [channels][32] weight tile (host-packed LDS image, linear
// copy) into LDS with LDS-DMA, double-buffered, one barrier per chunk.  Reference semantics:
//   gen_conv    models/networks/utils.py:9-33      padding = rate*(k-1)/2, zero padding
//   gen_deconv  models/networks/utils.py:35-51     nearest x2 (src = dst>>1) then 3x3 conv
//   concat      models/networks/editline_g.py:166-167,211 (second source = other tensor / pooled vector)
#include "se_device.h"

#include <cstdlib>

namespace se {

// FAST: single-source layers (everything but the concat fallbacks).  fp32 MFMA and VALU share a SIMD's issue time
// (se_wino.hip), and the generic gather spends ~17 VALU instructions per staged granule on bounds tests, 64-bit
// addresses and the zero-page select -- as much SIMD time as the MFMAs of the narrow (N48/N24) layers.  The fast
// path stages through a buffer resource: per pixel row it keeps one byte offset and one tap-validity bit mask
// (built once per workgroup), so a granule costs 3 VALU (bit extract, add, or) and an out-of-image tap is simply an
// out-of-range offset, for which the hardware delivers zeros.
// SPLIT: the workgroup owns only NT of the layer's row tiles -- group blockIdx.y; for the non-MIXED layouts a group is
// NT/2 feature tiles plus their NT/2 gate tiles, so the gate stays a register epilogue.  Used with PT = 1 for grids
// that would leave most CUs idle (one image): 6x (N=192) / 3x (N=96) more, shorter workgroups.
// BF16: activations and weights stored as bf16 (BASELINE config 5): a granule is 8 channels, a chunk 64 k-values, the
// MFMA v_mfma_f32_16x16x32_bf16 with fp32 accumulators; bias, activation and gate stay fp32, the gated result is
// rounded to bf16 once (round to nearest even) when it is stored.  Same gather, same LDS image, same tile geometry.
// STAGES: depth of the LDS ring.  2 = double buffering, one barrier and one vmcnt(0) per chunk: right when a chunk carries
// microseconds of MFMAs.  > 2 (a power of two): a ring with STAGES - 1 chunks in flight and a COUNTED vmcnt (only the
// oldest chunk is waited for), available for the small-grid shapes (SE_LL_STAGES=4; measured, not the default).
template <int NT, int PT, bool MIXED, int WPS, bool FAST, bool SPLIT, bool BF16, int STAGES>
__global__ __launch_bounds__(256, WPS) void gconv_kernel(const GConvParams p) {
  constexpr int PIX = PT * 64;
  constexpr int NP = NT * 16;
  constexpr int ES = BF16 ? 2 : 4;           // bytes per stored element
  constexpr int XBYTES = PIX * 128, WBYTES = NP * 128;
  constexpr int NX = PT * 2;                 // X staging pieces (8 rows each) per wave per chunk
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + STAGES * XBYTES;
  int2* rowtab = (int2*)smem;   // aliases the X buffers: consumed into registers before the first DMA

  const int tid = threadIdx.x, lane = tid & 63;
  const float eluw = p.act == 0 ? 1.f : 0.f;      // act_fast: ELU weight of the gated epilogue (wave-uniform)
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // sub-pixel upsample form: this workgroup computes output pixels (2yy+py, 2xx+px); taps (a,b) in {0,1}^2 read
  // source (yy + a - 1 + py, xx + b - 1 + px) with weights summed over the 3x3 taps that collapse onto it.  The four
  // class workgroups of a pixel tile sit side by side in one XCD's dispatch sequence (class_tile, se_device.h).
  int tile_idx = p.xcd ? xcd_tile(blockIdx.x, gridDim.x) : (int)blockIdx.x, cls = 0;
  if (p.up2 && !class_tile((int)blockIdx.x, (p.total_pix + PIX - 1) / PIX, p.xcd, tile_idx, cls)) return;
  const int tile_base = tile_idx * PIX;
  const int py = cls >> 1, px = cls & 1;
  const int pady = p.up2 ? 1 - py : p.pad, padx = p.up2 ? 1 - px : p.pad;
  const int npf = SPLIT ? p.np_full : NP;            // rows of one chunk of the packed weight image
  const int grp = SPLIT ? (int)blockIdx.y : 0;
  const float* wbase = p.wpk + (size_t)cls * p.nch * npf * 32;
  // 8-row block `rbk` of this workgroup's W tile -> 8-row block of the layer's packed image
  auto full_rbk = [&](int rbk) -> int {
    if (!SPLIT) return rbk;
    const int lt = rbk >> 1;
    const int ft = MIXED ? grp * NT + lt : (lt < NT / 2 ? grp * (NT / 2) + lt : p.nf_full + grp * (NT / 2) + (lt - NT / 2));
    return ft * 2 + (rbk & 1);
  };

  // ---- row table: (batch, packed y0|x0) of every pixel row of the tile; invalid rows fail the bounds test
  const int HoWo = p.Ho * p.Wo;
  for (int r = tid; r < PIX; r += 256) {
    const int pidx = tile_base + r;
    int2 e = make_int2(0, 0x40000000);
    if (pidx < p.total_pix) {
      const int b = (int)udiv_magic((unsigned)pidx, p.div_hw_m, p.div_hw_l), rem = pidx - b * HoWo;
      const int oy = (int)udiv_magic((unsigned)rem, p.div_w_m, p.div_w_l), ox = rem - oy * p.Wo;
      e = make_int2(b, ((oy * p.stride) << 16) | (ox * p.stride));
    }
    rowtab[r] = e;
  }
  __syncthreads();
  int rb[NX], ryx[NX];
  unsigned pixoff[NX], inv[NX];     // FAST: byte offset of the row's tap-(0,0) pixel before padding; ~tap validity mask
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int2 e = rowtab[(i * 4 + w) * 8 + (lane >> 3)];
    rb[i] = e.x;
    ryx[i] = e.y;
    if (FAST) {
      const int y0 = e.y >> 16, x0 = e.y & 0xffff;       // invalid rows: y0 = 16384, every tap fails the row test
      pixoff[i] = (unsigned)((e.x * p.Hin + y0) * p.Win + x0) * (unsigned)(p.C0 * ES);
      const int KH = p.magicKH;
      unsigned mask = 0;
      if (p.dil == 1) {
        // closed form: taps kx in [padx - x0, Wlim - 1 + padx - x0] (clipped to the kernel) are inside, same for ky;
        // the column mask is replicated to every kernel row by one multiply and cut to the valid rows
        const int clo = max(padx - x0, 0), chi = min(p.Wlim - 1 + padx - x0, p.KW - 1);
        const int rlo = max(pady - y0, 0), rhi = min(p.Hlim - 1 + pady - y0, KH - 1);
        const unsigned colbits = chi >= clo ? (2u << chi) - (1u << clo) : 0u;
        const unsigned rowsel = rhi >= rlo ? (2u << (rhi * p.KW + p.KW - 1)) - (1u << (rlo * p.KW)) : 0u;
        mask = __umul24(colbits, p.rep) & rowsel;
      } else {
        unsigned colbits = 0;
        for (int kx = 0; kx < p.KW; ++kx) colbits |= ((unsigned)(x0 + kx * p.dil - padx) < (unsigned)p.Wlim ? 1u : 0u) << kx;
        for (int ky = 0; ky < KH; ++ky)
          if ((unsigned)(y0 + ky * p.dil - pady) < (unsigned)p.Hlim) mask |= colbits << (ky * p.KW);
      }
      inv[i] = ~mask;                                     // bits >= T stay set: K padding reads zeros
    }
  }
  __syncthreads();

  // staging role of this lane: physical slot ps of rows (piece*8 + lane>>3) <-> logical k-slot s_log
  const int s_log = (lane & 7) ^ (4 * (w & 1) + (lane >> 4));
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const unsigned lds_x = lds_addr_of(Xb), lds_w = lds_addr_of(Wb);

  const se_i32x4 rsrc = make_rsrc(p.src0, (unsigned)p.B * p.Hin * p.Win * p.C0 * (unsigned)ES);
  // fast path staging, split so that the pieces can be issued one per MFMA step (mfma_chunk hook): a burst of
  // vector-memory instructions fills the CU's queue and stalls the wave (and its MFMAs) in front of it
  constexpr int NWP = (NT * 2 + 3) / 4;          // W staging pieces per wave
  constexpr int NPIECE = NX + NWP;
  int f_tap = 0;
  unsigned f_delta = 0;
  auto fast_head = [&](int ch) {
    const int gi = ch * 8 + s_log;
    int tap = __umul24(gi, p.magicCG) >> 16;
    const int cg = gi - __umul24(tap, p.CG);
    const int ky = __umul24(tap, p.magicKW) >> 8, kx = tap - __umul24(ky, p.KW);
    const int dy = __mul24(ky, p.dil) - pady, dx = __mul24(kx, p.dil) - padx;
    f_delta = (unsigned)(__mul24(__mul24(dy, p.Win) + dx, p.C0 * ES) + cg * 16);
    f_tap = min(tap, 31);
  };
  auto fast_piece = [&](int ch, int buf, int q) {      // q compile-time after unrolling
    if (q < NX) {
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)inv[q], f_tap, 1);      // all ones if the tap is outside
      bufdma16((pixoff[q] + f_delta) | m, rsrc, lds_x + buf * XBYTES + (q * 4 + w) * 1024);
    } else if (q < NPIECE) {
      const int rbk = (q - NX) * 4 + w;
      if (rbk < NT * 2)
        glds16_s(wbase + (size_t)ch * npf * 32 + full_rbk(rbk) * 256, (unsigned)lane * 16u, lds_w + buf * WBYTES + rbk * 1024);
    }
  };
  auto stage_fast = [&](int ch, int buf) {
    fast_head(ch);
#pragma unroll
    for (int q = 0; q < NPIECE; ++q) fast_piece(ch, buf, q);
  };
  auto stage_any = [&](int ch, int buf) {
    // which granule of the flattened K axis this lane fetches: (tap, 4-channel group)
    const int gi = ch * 8 + s_log;
    const int tap = (gi * p.magicCG) >> 16, cg = gi - tap * p.CG;
    const int ky = (tap * p.magicKW) >> 8, kx = tap - ky * p.KW;
    const int dy = ky * p.dil - pady, dx = kx * p.dil - padx;
    const bool tapok = tap < p.T;
    const bool first = cg < p.C0g;
    const char* base = (const char*)(first ? p.src0 : p.src1);
    const int cs = (first ? p.C0 : p.C1) * ES;                  // bytes per pixel
    const int coff = (first ? cg : cg - p.C0g) * 16;            // byte offset of the granule inside the pixel
    const bool vec = (!first) && p.src1_vec;
    const unsigned xdst = lds_x + buf * XBYTES;
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      int iy = (ryx[i] >> 16) + dy, ix = (ryx[i] & 0xffff) + dx;
      const bool ok = tapok & ((unsigned)iy < (unsigned)p.Hlim) & ((unsigned)ix < (unsigned)p.Wlim);
      iy >>= p.ushift;
      ix >>= p.ushift;
      const unsigned pix = vec ? (unsigned)rb[i] : (unsigned)((rb[i] * p.Hin + iy) * p.Win + ix);
      const char* g = base + (size_t)pix * (unsigned)cs + (unsigned)coff;
      g = ok ? g : (const char*)p.zeros;
      glds16(g, xdst + (i * 4 + w) * 1024);
    }
    const unsigned wdst = lds_w + buf * WBYTES;
    const float* wsrc = wbase + (size_t)ch * npf * 32 + lane * 4;
#pragma unroll
    for (int j = 0; j < (NT * 2 + 3) / 4; ++j) {
      const int rbk = j * 4 + w;
      if (rbk < NT * 2) glds16(wsrc + full_rbk(rbk) * 256, wdst + rbk * 1024);
    }
  };
  auto stage = [&](int ch, int buf) {
    if (FAST) stage_fast(ch, buf);
    else stage_any(ch, buf);
  };

  // (Zero-initialised accumulators and a bias add in the epilogue, unlike the raw-tile / Winograd kernels: here the
  // accumulators' first use sits behind the issue of the NEXT chunk's DMA, and a bias load that hipcc sinks to that point
  // makes the first MFMA wait for that chunk too -- measured +5.6 % at batch 1, where a workgroup has few chunks.)
  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  if constexpr (STAGES > 2) {
    // ---- ring of STAGES slots, AHEAD = STAGES - 1 chunks issued before the first MFMA.  Iteration ch: wait until chunk ch
    // has landed (every wave has at least PMIN DMA instructions per chunk in flight, so "all but PMIN * (AHEAD - 1)"
    // covers the oldest chunk), barrier (also: every wave is done reading chunk ch - 1), refill that slot with chunk
    // ch + AHEAD under the MFMAs of chunk ch.  The DMA instructions are the only vector-memory operations in the loop.
    constexpr int AHEAD = STAGES - 1;
    constexpr int PMIN = NX + (NT * 2) / 4;
    static_assert((STAGES & (STAGES - 1)) == 0 && PMIN * (AHEAD - 1) <= 63, "ring depth");
    for (int c0 = 0; c0 < AHEAD && c0 < p.nch; ++c0) stage(c0, c0);
    for (int ch = 0; ch < p.nch; ++ch) {
      const int buf = ch & (STAGES - 1);
      if (ch + AHEAD - 1 < p.nch) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PMIN * (AHEAD - 1)) : "memory");
      else dma_wait_all();
      __syncthreads();
      const bool more = ch + AHEAD < p.nch;
      const int nbuf = (ch + AHEAD) & (STAGES - 1);
      if (FAST) {
        if (more) fast_head(ch + AHEAD);
        constexpr int SLOTS = NT, PER = (NPIECE + SLOTS - 1) / SLOTS;
        auto hook = [&](int slot) {
          if (more) {
#pragma unroll
            for (int u = 0; u < PER; ++u) fast_piece(ch + AHEAD, nbuf, slot * PER + u);
          }
        };
        if (BF16) mfma_chunk16<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1, hook);
        else mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1, hook);
      } else {
        if (more) stage(ch + AHEAD, nbuf);
        if (BF16) mfma_chunk16<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
        else mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
      }
    }
  } else {
  stage(0, 0);
  dma_wait_all();
  __syncthreads();
  for (int ch = 0; ch < p.nch; ++ch) {
    const int buf = ch & 1;
    if (FAST) {
      // DMA of the next chunk flies under the MFMAs, one piece per MFMA step
      const bool more = ch + 1 < p.nch;
      if (more) fast_head(ch + 1);
      // ... all within the first k-half of the chunk, so the youngest piece still has half a chunk of MFMAs to land
      constexpr int SLOTS = NT, PER = (NPIECE + SLOTS - 1) / SLOTS;
      auto hook = [&](int slot) {
        if (more) {
#pragma unroll
          for (int u = 0; u < PER; ++u) fast_piece(ch + 1, buf ^ 1, slot * PER + u);
        }
      };
      if (BF16) mfma_chunk16<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1, hook);
      else mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1, hook);
    } else {
      if (ch + 1 < p.nch) stage(ch + 1, buf ^ 1);               // DMA of the next chunk flies under the MFMAs
      if (BF16) mfma_chunk16<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
      else mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    }
    dma_wait_all();
    __syncthreads();
  }
  }

  // ---- epilogue: bias, gate, NHWC store (a lane holds 4 consecutive channels of pixel lane&15)
  const int q = lane >> 4;
  auto out_off = [&](int pidx) -> size_t {
    if (!p.up2) return (size_t)pidx * p.G;
    const int b = (int)udiv_magic((unsigned)pidx, p.div_hw_m, p.div_hw_l), rem = pidx - b * HoWo;
    const int yy = (int)udiv_magic((unsigned)rem, p.div_w_m, p.div_w_l), xx = rem - yy * p.Wo;
    return ((size_t)(b * p.OH + 2 * yy + py) * p.OW + 2 * xx + px) * p.G;
  };
  if (!MIXED) {
    constexpr int NF = NT / 2;       // tiles [0,NF): features, [NF,NT): matching gates
    const int nff = SPLIT ? p.nf_full : NF;
#pragma unroll
    for (int nt = 0; nt < NF; ++nt) {
      const int c0 = (grp * NF + nt) * 16 + q * 4;
      const f32x4 bf = *(const f32x4*)(p.bias + c0);
      const f32x4 bg = *(const f32x4*)(p.bias + nff * 16 + c0);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int pidx = tile_base + (w * PT + pt) * 16 + (lane & 15);
        f32x4 o;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const float f = acc[nt][pt][r] + bf[r];
          const float g = acc[nt + NF][pt][r] + bg[r];
          const float a = act_fast(f, eluw);
          o[r] = a * sigmoid_fast(g);
        }
        if (c0 < p.G && pidx < p.total_pix) {
          if (BF16) *(uint2*)((char*)p.dst + (out_off(pidx) + c0) * 2) = make_uint2(pack_bf16x2(o[0], o[1]), pack_bf16x2(o[2], o[3]));
          else *(f32x4*)(p.dst + out_off(pidx) + c0) = o;
        }
      }
    }
  } else {
    // tile rows 0-7: features 8nt..8nt+7 (lanes q=0,1 = lanes 0-31); rows 8-15: the matching gates (lane + 32).
    // Two v_permlane32_swap per quad give every lane two complete (feature, gate) pairs -- lanes 0-31 channels
    // c0, c0+1, lanes 32-63 channels c0+2, c0+3 -- so each lane evaluates two outputs; the former per-value
    // ds_bpermute exchange cost these narrow layers about as much as a third of their MFMAs.
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const int c0 = (grp * NT + nt) * 8 + (q & 1) * 4 + (q >> 1) * 2;
      const f32x4 bq = *(const f32x4*)(p.bias + (grp * NT + nt) * 16 + q * 4);
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) {
        const int pidx = tile_base + (w * PT + pt) * 16 + (lane & 15);
        const f32x4 v = acc[nt][pt] + bq;
        const auto s02 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[0]), __float_as_uint(v[2]), false, false);
        const auto s13 = __builtin_amdgcn_permlane32_swap(__float_as_uint(v[1]), __float_as_uint(v[3]), false, false);
        const float f0 = __uint_as_float(s02[0]), g0 = __uint_as_float(s02[1]);
        const float f1 = __uint_as_float(s13[0]), g1 = __uint_as_float(s13[1]);
        float2 o;
        o.x = act_fast(f0, eluw) * sigmoid_fast(g0);
        o.y = act_fast(f1, eluw) * sigmoid_fast(g1);
        if (c0 < p.G && pidx < p.total_pix) {
          if (BF16) *(unsigned*)((char*)p.dst + (out_off(pidx) + c0) * 2) = pack_bf16x2(o.x, o.y);
          else *(float2*)(p.dst + out_off(pidx) + c0) = o;
        }
      }
    }
  }
}

template <int NT, int PT, bool MIXED, int WPS, bool FAST, bool SPLIT, bool BF16, int STAGES>
static hipError_t launch_gconv_f(const GConvParams& p, hipStream_t st, int label) {
  constexpr int PIX = PT * 64;
  constexpr int LDS = STAGES * (PIX * 128 + NT * 16 * 128);
  static_assert(PIX * 8 <= 2 * PIX * 128, "row table aliases the X buffers");
  {
    hipError_t e = ensure_max_lds((const void*)gconv_kernel<NT, PT, MIXED, WPS, FAST, SPLIT, BF16, STAGES>, LDS);
    if (e != hipSuccess) return e;
  }
  const int tiles = (p.total_pix + PIX - 1) / PIX;
  const int grid = p.up2 ? class_tile_grid(tiles) : tiles;
  const int groups = SPLIT ? p.np_full / (NT * 16) : 1;
  set_launch_grid((long)grid * groups);
  ProfScope ps_(st, label);
  hipLaunchKernelGGL((gconv_kernel<NT, PT, MIXED, WPS, FAST, SPLIT, BF16, STAGES>), dim3(grid, groups), dim3(256), LDS, st, p);
  return hipGetLastError();
}

// the buffer-resource fast path needs one source, <= 31 taps and a source below 2 GiB (32-bit byte offsets)
static bool fast_eligible(const GConvParams& p) {
  return opt(OPT_GCONV_FAST) != 0 && p.C0g == p.CG && !p.ushift && p.T <= 31 && p.magicKH * p.KW == p.T &&
         (long long)p.B * p.Hin * p.Win * p.C0 * (p.bf16 ? 2 : 4) < (1ll << 31);
}
template <int NT, int PT, bool MIXED, int WPS, bool SPLIT = false, int STAGES = 2>
static hipError_t launch_gconv_t(const GConvParams& p, hipStream_t st, int label) {
  if (p.bf16)
    return fast_eligible(p) ? launch_gconv_f<NT, PT, MIXED, WPS, true, SPLIT, true, STAGES>(p, st, label)
                            : launch_gconv_f<NT, PT, MIXED, WPS, false, SPLIT, true, STAGES>(p, st, label);
  return fast_eligible(p) ? launch_gconv_f<NT, PT, MIXED, WPS, true, SPLIT, false, STAGES>(p, st, label)
                          : launch_gconv_f<NT, PT, MIXED, WPS, false, SPLIT, false, STAGES>(p, st, label);
}

// (A 3-stage LDS ring for the narrow MIXED shapes -- two chunks of DMA in flight, exact vmcnt waits -- was measured
// and is slower: 66 KiB of LDS leave two instead of three workgroups per CU, N48 2.91 -> 3.38 ms.)
// Tile shapes.  variant 0 is the default; SE_GCONV_VARIANT_<cfg>=k (environment) selects another one for
// tuning sweeps.  Shapes with LDS <= 80 KiB and <= 256 registers run two workgroups per CU, so the staging
// code, LDS-read latency and epilogue of one overlap the MFMAs of the other.
static int variant_of(int cfg) {
  static const int o[4] = {OPT_GCONV_VARIANT_N192, OPT_GCONV_VARIANT_N96, OPT_GCONV_VARIANT_N48, OPT_GCONV_VARIANT_N24};
  return opt(o[cfg]);
}

hipError_t launch_gconv(int cfg, const GConvParams& p, hipStream_t st) {
  if (p.small_grid) {
    // low-latency shapes: 64-pixel tiles; the wide layers also split their rows over blockIdx.y (one feature tile + its
    // gate tile per workgroup): 6x / 3x more workgroups, each a sixth / third as long
    // SE_LL_STAGES=4 (developer switch) runs them on the 4-slot ring with a counted vmcnt instead of double buffering.
    // Measured (1 image): 256x256 1.327 -> 1.313 ms, 512x512 3.75 -> 4.08 ms (48 KB of LDS per workgroup cost more
    // residency than the deeper prefetch returns; per chunk the ~0.8 us are barrier + issue overhead, not DMA latency).
    const int stages = opt(OPT_LL_STAGES);
    if (stages == 2) {
      switch (cfg) {
        case GC_N192: return launch_gconv_t<2, 1, false, 4, true>(p, st, PL_GCONV_N192);
        case GC_N96: return launch_gconv_t<2, 1, false, 4, true>(p, st, PL_GCONV_N96);
        case GC_N48: return launch_gconv_t<3, 1, true, 4>(p, st, PL_GCONV_N48);
        case GC_N24: return launch_gconv_t<2, 1, true, 4>(p, st, PL_GCONV_N24);
      }
    }
    switch (cfg) {
      case GC_N192: return launch_gconv_t<2, 1, false, 3, true, 4>(p, st, PL_GCONV_N192);
      case GC_N96: return launch_gconv_t<2, 1, false, 3, true, 4>(p, st, PL_GCONV_N96);
      case GC_N48: return launch_gconv_t<3, 1, true, 3, false, 4>(p, st, PL_GCONV_N48);
      case GC_N24: return launch_gconv_t<2, 1, true, 3, false, 4>(p, st, PL_GCONV_N24);
    }
    return hipErrorInvalidValue;
  }
  const int var = variant_of(cfg);
  switch (cfg) {
    case GC_N192:
      if (var == 1) return launch_gconv_t<12, 4, false, 1>(p, st, PL_GCONV_N192);
      return launch_gconv_t<12, 2, false, 2>(p, st, PL_GCONV_N192);
    case GC_N96:
      if (var == 1) return launch_gconv_t<6, 4, false, 1>(p, st, PL_GCONV_N96);
      if (var == 2) return launch_gconv_t<6, 2, false, 2>(p, st, PL_GCONV_N96);
      return launch_gconv_t<6, 3, false, 2>(p, st, PL_GCONV_N96);
    case GC_N48:
      if (var == 1) return launch_gconv_t<3, 8, true, 1>(p, st, PL_GCONV_N48);
      if (var == 2) return launch_gconv_t<3, 4, true, 2>(p, st, PL_GCONV_N48);
      if (var == 3) return launch_gconv_t<3, 1, true, 4>(p, st, PL_GCONV_N48);
      return launch_gconv_t<3, 2, true, 2>(p, st, PL_GCONV_N48);
    case GC_N24:
      if (var == 1) return launch_gconv_t<2, 8, true, 1>(p, st, PL_GCONV_N24);
      if (var == 2) return launch_gconv_t<2, 4, true, 2>(p, st, PL_GCONV_N24);
      if (var == 3) return launch_gconv_t<2, 1, true, 4>(p, st, PL_GCONV_N24);
      return launch_gconv_t<2, 2, true, 2>(p, st, PL_GCONV_N24);
  }
  return hipErrorInvalidValue;
}

}  // namespace se
