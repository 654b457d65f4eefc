// Contextual attention (patch 4x4, stride 2) on gfx950.
//
// Reference: ReduceContextAttentionP1/P2, /root/reference/models/networks/splitcam.py:37-108,132-153,
// configured at models/networks/editline_g.py:35-42 and called at :203-207 with f = b = x:
//   xn      = x / sqrt(sum_hw x^2 + 1e-8)                      per (batch, channel)        :40
//   S[j,i]  = <patch_j(xn), patch_i(x)>                        keys j, queries i, d = 16*96 :42-44,69
//   S[j,i] *= (mean over patch j of (1 - avgpool4(mask)) > 0.1)  multiplicative zero        :49-53,90,104
//   P       = softmax_j(10 * S)                                                            :105
//   out     = fold(P^T V), V = raw patches of x, overlap-add without normalisation        :138-153
// Three kernels: scores (gather-GEMM, both operands gathered from NHWC x / xn, written query-major
// so the softmax axis is contiguous), row softmax, and P.V fused with the fold as a gather: an output
// pixel of parity class (py,px) sums the <=4 patches covering it, i.e. one GEMM with
// K = 4 (covering patch) x L (keys) -- deterministic, no atomics.
//
// That is the round-1 patch form (att_* kernels, SE_ATT_V1=1).  What runs is the space-to-depth form of round 2 (att2_*,
// DESIGN.md 3.3), in this file in launch order:
//   att2_prep / att2_transpose          keys (fp32: y = x sqrt(rn) for both operands of a bitwise symmetric E), key tables, V^T
//   att2_emean1/2, att2_eoff, _eoff4    bf16 mode: row / column offsets of the doubly centred fp16 E
//   att2_pair                           E GEMM (fp32: tiles on / right of the diagonal only, mirrored; panel order per XCD)
//   att2_stats_lds + att2_ptilde_lds    round 4: statistics and P~ passes on LDS-staged 4 x 4 patches of rows (producer waves)
//   att2_stats + att2_ptilde4 / 1       round 3: the same fused form as row-streaming kernels (widths the LDS form does not take)
//   att2_softmax[_reg] + att2_boxsum[4] three-pass form (P materialised: `similar_out`, SE_ATT_FUSED=0)
//   att2_pv                             P~ . V GEMM per parity class
#include "se_device.h"

#include <cstdlib>

namespace se {

__global__ void att_prep_kernel(const AttParams p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long nx = (long)p.B * p.h * p.w * 24;    // granules
  if (idx < nx) {
    const int cg = idx % 24;
    const long pix = idx / 24;
    const int b = pix / (p.h * p.w);
    const f32x4 v = *(const f32x4*)(p.x + idx * 4);
    const f32x4 r = *(const f32x4*)(p.rn + b * 96 + cg * 4);
    *(f32x4*)(p.xn + idx * 4) = v * r;
  }
  if (idx < (long)p.B * p.Lp) {
    const int b = idx / p.Lp, j = idx - (long)b * p.Lp;
    float val = 0.f;
    if (j < p.L) {
      const int jy = j / p.ws, jx = j - jy * p.ws;
      const int H = p.h * 4, W = p.w * 4;
      // the 4x4 patch of the avg-pooled map covers a 16x16 full-resolution window; for a {0,1} mask the
      // sum is an integer <= 256, so the mean is exact and the > 0.1 test is order independent
      float hole = 0.f;
      for (int yy = 0; yy < 16; ++yy)
        for (int xx = 0; xx < 16; ++xx) hole += p.hard[((long)b * H + jy * 8 + yy) * W + jx * 8 + xx];
      const float mm = 1.f - hole * (1.f / 256.f);
      val = mm > p.th ? 1.f : 0.f;
    }
    p.valid[idx] = val;
  }
}

// S[b][i][j] = scale * valid[j] * <K_j, Q_i>
template <int NT, int PT>
__global__ __launch_bounds__(256) void att_score_kernel(const AttParams p) {
  constexpr int PIX = PT * 64, NP = NT * 16;
  constexpr int XBYTES = PIX * 128, WBYTES = NP * 128;
  constexpr int NX = PT * 2, NW = (NT * 2 + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z;
  const int q0 = blockIdx.x * PIX, k0 = blockIdx.y * NP;
  // byte offset of the patch origin of a query / key row inside THIS image, or an out-of-range offset: both
  // operands are staged through buffer resources (hardware zero fill, one VALU add per granule; se_gconv.hip)
  auto origin = [&](int i) -> unsigned {
    if (i >= p.L) return 0x80000000u;
    const int py = i / p.ws, px = i - py * p.ws;
    return (unsigned)(((2 * py) * p.w + 2 * px) * 384);
  };
  const se_i32x4 rs_q = make_rsrc(p.x + (size_t)b * p.h * p.w * 96, (unsigned)p.h * p.w * 96u * 4u);
  const se_i32x4 rs_k = make_rsrc(p.xn + (size_t)b * p.h * p.w * 96, (unsigned)p.h * p.w * 96u * 4u);
  unsigned qo[NX], ko[NW];
#pragma unroll
  for (int i = 0; i < NX; ++i) qo[i] = origin(q0 + (i * 4 + w) * 8 + (lane >> 3));
#pragma unroll
  for (int j = 0; j < NW; ++j) ko[j] = (j * 4 + w) < NT * 2 ? origin(k0 + (j * 4 + w) * 8 + (lane >> 3)) : 0x80000000u;

  const int s_log = (lane & 7) ^ (4 * (w & 1) + (lane >> 4));
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const unsigned lds_x = lds_addr_of(Xb), lds_w = lds_addr_of(Wb);

  auto stage = [&](int ch, int buf) {
    const int gi = ch * 8 + s_log;                 // granule of the 16 taps x 24 channel-groups
    const int tap = (gi * 2731) >> 16, cg = gi - tap * 24;      // gi / 24 for gi < 4096
    const unsigned doff = (unsigned)((__mul24(tap >> 2, p.w) + (tap & 3)) * 384 + cg * 16);
    const unsigned xdst = lds_x + buf * XBYTES, wdst = lds_w + buf * WBYTES;
#pragma unroll
    for (int i = 0; i < NX; ++i) bufdma16(qo[i] + doff, rs_q, xdst + (i * 4 + w) * 1024);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int rbk = j * 4 + w;
      if (rbk < NT * 2) bufdma16(ko[j] + doff, rs_k, wdst + rbk * 1024);
    }
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  constexpr int NCH = 48;   // 16 taps * 96 ch / 32
  stage(0, 0);
  dma_wait_all();
  __syncthreads();
  for (int ch = 0; ch < NCH; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < NCH) stage(ch + 1, buf ^ 1);
    mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    dma_wait_all();
    __syncthreads();
  }
  const int q = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt) {
    const int j = k0 + nt * 16 + q * 4;
    if (j >= p.Lp) continue;
    const f32x4 v = *(const f32x4*)(p.valid + (long)b * p.Lp + j);
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) {
      const int i = q0 + (w * PT + pt) * 16 + (lane & 15);
      if (i < p.L) *(f32x4*)(p.S + ((long)b * p.L + i) * p.Lp + j) = acc[nt][pt] * v * p.scale;
    }
  }
}

// softmax over keys j < L of each query row; pad columns [L, Lp) are set to 0.  One wave per row.
__global__ __launch_bounds__(256) void att_softmax_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  const long row = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
  if (row >= (long)p.B * p.L) return;
  float* s = p.S + row * p.Lp;
  float m = -INFINITY;
  for (int j = lane; j < p.L; j += 64) m = fmaxf(m, s[j]);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int j = lane; j < p.L; j += 64) {
    const float e = expf(s[j] - m);
    s[j] = e;
    sum += e;
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  for (int j = lane; j < p.Lp; j += 64) s[j] = j < p.L ? s[j] * inv : 0.f;
}

// out[b, pos, c] = sum over the <=4 patches covering pos of sum_j P[i][j] * x[b, 2j + (ky,kx), c]
template <int PT>
__global__ __launch_bounds__(256) void att_pv_kernel(const AttParams p) {
  constexpr int NT = 6;
  constexpr int PIX = PT * 64;
  constexpr int XBYTES = PIX * 128, VBYTES = 32 * 384;
  constexpr int NX = PT * 2;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;                        // P tiles  [PIX][32 keys]
  char* Vb = smem + 2 * XBYTES;           // V tiles  [32 keys][96 ch]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int b = blockIdx.z, cls = blockIdx.y, py = cls >> 1, px = cls & 1;
  const int ch_ = p.h >> 1, cw_ = p.w >> 1;      // class image size
  const int t0 = blockIdx.x * PIX;
  // Staging through buffer resources of THIS image's score matrix and value tensor (32-bit offsets, hardware zero
  // fill for out-of-range lanes; se_gconv.hip fast path): per P row one byte offset and a 4-bit validity mask over
  // the (a, bb) patch combos, per V granule the key's pixel offset -- 3 VALU per staged granule in the loop.
  const int s_log = (lane & 7) ^ (4 * (w & 1) + (lane >> 4));
  const se_i32x4 rs_S = make_rsrc(p.S + (size_t)b * p.L * p.Lp, (unsigned)p.L * p.Lp * 4u);
  const se_i32x4 rs_x = make_rsrc(p.x + (size_t)b * p.h * p.w * 96, (unsigned)p.h * p.w * 96u * 4u);
  unsigned xoff[NX], xinv[NX];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int r = t0 + (i * 4 + w) * 8 + (lane >> 3);
    const bool rowok = r < ch_ * cw_;
    const int ry = rowok ? r / cw_ : 0, rx = rowok ? r - (r / cw_) * cw_ : 0;
    xoff[i] = (unsigned)((ry * p.ws + rx) * p.Lp + s_log * 4) * 4u;
    unsigned m = 0;
#pragma unroll
    for (int combo = 0; combo < 4; ++combo) {
      const int iy = ry - (combo >> 1), ix = rx - (combo & 1);
      if (rowok && (unsigned)iy < (unsigned)p.hs && (unsigned)ix < (unsigned)p.ws) m |= 1u << combo;
    }
    xinv[i] = ~m;
  }
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const unsigned lds_x = lds_addr_of(Xb), lds_v = lds_addr_of(Vb);
  const int jchunks = p.Lp >> 5;
  const int nch = 4 * jchunks;
  // V staging role: 3 pieces per wave; piece it -> granule gidx = it*64 + lane -> key row jr, channel group cg
  int vjr[3];
  unsigned vcoff[3];
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const int gidx = (k * 4 + w) * 64 + lane;
    vjr[k] = gidx / 24;
    // bank swizzle: key rows j and j+4 (read by lane groups q and q+1 of one ds_read_b32) would hit the same 16 banks
    // (row stride 96 floats = 3 bank rows); rows with bit 2 set keep their granules swapped in blocks of 4
    vcoff[k] = (unsigned)((gidx - vjr[k] * 24) ^ (((vjr[k] >> 2) & 1) << 2)) * 16u;
  }
  unsigned ws_m;
  int ws_l;
  {   // x / ws for x < 2^32 (se_device.h udiv_magic), divisor uniform
    int ll = 0;
    while ((1u << ll) < (unsigned)p.ws) ++ll;
    ws_l = ll;
    ws_m = (unsigned)((((unsigned long long)1 << 32) * (((unsigned long long)1 << ll) - (unsigned)p.ws)) / (unsigned)p.ws + 1);
  }

  auto stage = [&](int ch, int buf) {
    const int combo = ch / jchunks, jc = ch - combo * jchunks;      // uniform
    const int a = combo >> 1, bb = combo & 1;
    const unsigned xdst = lds_x + buf * XBYTES, vdst = lds_v + buf * VBYTES;
    const unsigned xdelta = (unsigned)((jc * 32 - (a * p.ws + bb) * p.Lp) * 4);      // uniform
#pragma unroll
    for (int i = 0; i < NX; ++i) {
      const unsigned m = (unsigned)__builtin_amdgcn_sbfe((int)xinv[i], combo, 1);      // all ones: outside -> zeros
      bufdma16((xoff[i] + xdelta) | m, rs_S, xdst + (i * 4 + w) * 1024);
    }
    // V tile: key j -> pixel (2jy + py + 2a, 2jx + px + 2bb), 96 channels = 24 granules
    const unsigned vbase = (unsigned)(((py + 2 * a) * p.w + px + 2 * bb) * 384);        // uniform
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      const int j = jc * 32 + vjr[k];
      const int jy = (int)udiv_magic((unsigned)j, ws_m, ws_l), jx = j - jy * p.ws;
      const unsigned off = (unsigned)(__mul24(jy, p.w) + jx) * 768u + vbase + vcoff[k];
      bufdma16(j < p.L ? off : 0x80000000u, rs_x, vdst + (k * 4 + w) * 1024);
    }
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  dma_wait_all();
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) stage(ch + 1, buf ^ 1);
    const char* Xt = Xb + buf * XBYTES + w * PT * 2048;
    const float* Vt = (const float*)(Vb + buf * VBYTES);
    const int vsw = ((lane >> 4) & 1) << 4;          // (j >> 2) & 1 == (lane >> 4) & 1: see the V staging swizzle
#pragma unroll
    for (int half = 0; half < 2; ++half) {
      const int off = half ? off1 : off0;
      f32x4 xb[PT];
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) xb[pt] = *(const f32x4*)(Xt + pt * 2048 + off);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int j = half * 16 + (lane >> 4) * 4 + r;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
          const float a = Vt[j * 96 + ((nt * 16 + (lane & 15)) ^ vsw)];
#pragma unroll
          for (int pt = 0; pt < PT; ++pt)
            acc[nt][pt] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, xb[pt][r], acc[nt][pt], 0, 0, 0);
        }
      }
    }
    dma_wait_all();
    __syncthreads();
  }
  const int q = lane >> 4;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int i = t0 + (w * PT + pt) * 16 + (lane & 15);
    if (i >= ch_ * cw_) continue;
    const int yy = i / cw_, xx = i - yy * cw_;
    float* o = p.out + ((long)(b * p.h + 2 * yy + py) * p.w + 2 * xx + px) * 96;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) *(f32x4*)(o + nt * 16 + q * 4) = acc[nt][pt];
  }
}

static hipError_t launch_attention_v1(const AttParams& p, hipStream_t st) {
  {
    const long n = (long)p.B * p.h * p.w * 24;
    ProfScope ps_(st, PL_ATT_PREP);
    hipLaunchKernelGGL(att_prep_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
  }
  {
    constexpr int NT = 4, PT = 4;
    constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
    {
    hipError_t e = ensure_max_lds((const void*)att_score_kernel<NT, PT>, LDS);
    if (e != hipSuccess) return e;
  }
    dim3 grid((p.L + PT * 64 - 1) / (PT * 64), (p.L + NT * 16 - 1) / (NT * 16), p.B);
    set_launch_cost(2.0 * p.B * (double)p.L * p.L * 1536.0, 0.0);
    ProfScope ps_(st, PL_ATT_SCORE);
    hipLaunchKernelGGL((att_score_kernel<NT, PT>), grid, dim3(256), LDS, st, p);
  }
  {
    const long rows = (long)p.B * p.L;
    ProfScope ps_(st, PL_ATT_SOFTMAX);
    hipLaunchKernelGGL(att_softmax_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
  }
  {
    constexpr int PT = 4;
    constexpr int LDS = 2 * PT * 64 * 128 + 2 * 32 * 384;
    {
    hipError_t e = ensure_max_lds((const void*)att_pv_kernel<PT>, LDS);
    if (e != hipSuccess) return e;
  }
    const int cpix = (p.h >> 1) * (p.w >> 1);
    dim3 grid((cpix + PT * 64 - 1) / (PT * 64), 4, p.B);
    set_launch_cost(2.0 * p.B * (double)p.L * p.L * 1536.0, 0.0);
    ProfScope ps_(st, PL_ATT_PV);
    hipLaunchKernelGGL((att_pv_kernel<PT>), grid, dim3(256), LDS, st, p);
  }
  return hipGetLastError();
}


// =====================================================================================================================
// Space-to-depth form ("v2", the default).  The 4x4 patches at stride 2 overlap: tap (ky,kx) of patch i is pixel
// 2i + (ky,kx), so a score is a sum of per-pixel dot products that neighbouring (query, key) pairs share,
//   S[i][j] = sum_{d in {0,1}^2} E[i+d][j+d],   E[r][s] = sum_{cls, c} x[2r+cls][c] * xn[2s+cls][c]
// with r, s on the class grid hc x wc = h/2 x w/2 (i, j on its (hc-1) x (wc-1) sub-grid) and cls the 4 pixel parities:
// E is ONE GEMM of the space-to-depth tensors, K = 4*96 = 384 instead of 16*96 = 1536 per score.  The same
// identity on the output side: an output pixel of class cls at grid position r sums P over the <= 4 patches covering it,
//   out[2r+cls] = sum_s P~[r][s] * x[2s+cls],      P~[r][s] = sum_{d} P[r-d][s-d],
// one GEMM with K = R = hc*wc instead of 4*L.  Together 3.75x fewer multiply-adds than the patch form above
// (2*R^2*384*2 vs 2*L^2*1536*2 per image), and both GEMMs run the shared 32-k chunk core with b128 fragments
// (the values are staged transposed, [class][channel][key], so the A operand is k-contiguous).
// Kernels: prep (xn, key validity, transposed values) -> E GEMM -> row softmax (forms S from E on the fly; one wave per
// query) -> P~ box sum (one wave per row) -> P~.V GEMM per class (the four class workgroups of a pixel tile are
// neighbours in one XCD's share of the grid and share the P~ tile through its L2).  Deterministic, no atomics.
// Summation order differs from the reference's conv (fp32 rounding only): measured <= 2e-6 on `similar`.
// =====================================================================================================================

// NT_STORE: E, P and P~ are R x R matrices per image (0.5 - 1 GB per step at 512 x 512), far larger than the L2 and the
// 256 MB Infinity Cache, written once and read a pass later: their stores are non-temporal (`global_store ... nt`).  With
// ordinary stores the write-bound E GEMM of the bf16 mode ran at 2.5 TB/s; streaming, the same kernel takes 296 instead of
// 426 us (512 x 512 B=16) and the pass that reads E gains 5 % (its inputs are no longer evicted by the stores).
// BF16 (all att2 kernels): x, xn, xT, P~ and out are bf16 (8-channel granules, 64-key chunks); E, the softmax and P stay fp32.
template <bool BF16>
__global__ void att2_prep_kernel(const AttParams p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;
  const long nx = (long)p.B * p.h * p.w * (BF16 ? 12 : 24);    // granules
  if (idx < nx) {
    if (BF16) {
      const int cg = idx % 12;
      const long pix = idx / 12;
      const int b = pix / (p.h * p.w);
      const uint4 v = *(const uint4*)((const char*)p.x + idx * 16);
      const f32x4 r0 = *(const f32x4*)(p.rn + b * 96 + cg * 8), r1 = *(const f32x4*)(p.rn + b * 96 + cg * 8 + 4);
      *(uint4*)((char*)p.xn + idx * 16) =
          make_uint4(pack_bf16x2(bf16_lo(v.x) * r0[0], bf16_hi(v.x) * r0[1]), pack_bf16x2(bf16_lo(v.y) * r0[2], bf16_hi(v.y) * r0[3]),
                     pack_bf16x2(bf16_lo(v.z) * r1[0], bf16_hi(v.z) * r1[1]), pack_bf16x2(bf16_lo(v.w) * r1[2], bf16_hi(v.w) * r1[3]));
    } else {
      const int cg = idx % 24;
      const long pix = idx / 24;
      const int b = pix / (p.h * p.w);
      const f32x4 v = *(const f32x4*)(p.x + idx * 4);
      f32x4 r = *(const f32x4*)(p.rn + b * 96 + cg * 4);
      if (p.sym) r = (f32x4){sqrtf(r[0]), sqrtf(r[1]), sqrtf(r[2]), sqrtf(r[3])};      // symmetric operands (att2_pair_kernel)
      *(f32x4*)(p.xn + idx * 4) = v * r;                        // splitcam.py:40
    }
  }
  if (idx < (long)p.B * p.Rp) {
    const int b = idx / p.Rp, s = idx - (long)b * p.Rp;
    float val = -1.f;                                          // not a key position
    const int sy = s / p.wc, sx = s - sy * p.wc;
    if (s < p.R && sy < p.hs && sx < p.ws) {
      const int H = p.h * 4, W = p.w * 4;
      // the 4x4 patch of the avg-pooled map covers a 16x16 full-resolution window; for a {0,1} mask the
      // sum is an integer <= 256, so the mean is exact and the > 0.1 test is order independent (splitcam.py:49-53)
      float hole = 0.f;
      for (int yy = 0; yy < 16; ++yy)
        for (int xx = 0; xx < 16; ++xx) hole += p.hard[((long)b * H + sy * 8 + yy) * W + sx * 8 + xx];
      const float mm = 1.f - hole * (1.f / 256.f);
      val = mm > p.th ? 1.f : 0.f;
    }
    p.validR[idx] = val;
    if (p.kmul) {      // the key test of the fused streaming pass in arithmetic form (att2_ptilde4_kernel)
      p.kmul[idx] = val > 0.f ? p.scale * 1.44269504088896340736f : 0.f;
      p.kadd[idx] = val >= 0.f ? 0.f : -INFINITY;
    }
  }
  if (idx < p.guard) {      // guard bands: in front of image 0 of the key tables (not a key), around E (finite)
    p.validR[-1 - idx] = -1.f;
    if (p.kmul) { p.kmul[-1 - idx] = 0.f; p.kadd[-1 - idx] = -INFINITY; }
    p.E[-1 - idx] = 0.f;
    p.E[(size_t)p.B * p.R * p.Rp + idx] = 0.f;
  }
}

// xT[b][cls][c][s] = x[b][2sy+py][2sx+px][c] (raw values, splitcam.py:138-143), zero for s >= R.  Block = 32 keys of one class.
template <bool BF16>
__global__ __launch_bounds__(256) void att2_transpose_kernel(const AttParams p) {
  __shared__ float t[32][97];
  const int s0 = blockIdx.x * 32, cls = blockIdx.y, b = blockIdx.z, py = cls >> 1, px = cls & 1;
  for (int e = threadIdx.x; e < 32 * 96; e += 256) {
    const int sl = e / 96, c = e - sl * 96, s = s0 + sl;
    float v = 0.f;
    if (s < p.R) {
      const int sy = s / p.wc, sx = s - sy * p.wc;
      const size_t at = ((size_t)(b * p.h + 2 * sy + py) * p.w + 2 * sx + px) * 96 + c;
      v = BF16 ? __uint_as_float((unsigned)((const unsigned short*)p.x)[at] << 16) : p.x[at];
    }
    t[sl][c] = v;
  }
  __syncthreads();
  for (int e = threadIdx.x; e < 96 * 32; e += 256) {
    const int c = e >> 5, sl = e & 31;
    const size_t at = (((size_t)b * 4 + cls) * 96 + c) * p.Rp + s0 + sl;
    if (BF16) ((unsigned short*)p.xT)[at] = (unsigned short)(__float_as_uint(t[sl][c]) >> 16);      // exact: t holds a bf16 value
    else p.xT[at] = t[sl][c];
  }
}

// fp16-E form (bf16 mode): E is stored DOUBLY CENTRED, E'[r][s] = E[r][s] - ea[r] - eb[s] with ea[r] = <X_r, mean XN>,
// eb[s] = <mean X, XN_s> - <mean X, mean XN> (X_r, XN_s: the 384-vectors of a class-grid position; means over the positions of
// the image) -- the additive row and column effects of E, known before the GEMM.  fp16 carries a fixed 2^-12 RELATIVE error:
// gated feature maps share a large mean vector, so raw E is "15 +- 3" and every score is off by ~4e-3 (x 4 terms x scale 10:
// several % in P; an end-to-end fuzz case with the larger-gain weight set moved 10 % further from the oracle than with fp32 E),
// while the centred values are the +- 3.  With zero-mean inputs the offsets vanish and nothing changes.  (Centring on the
// diagonal instead -- E' = minus half the squared patch distance -- was built first: it is exact where patches are alike and
// WORSE than raw fp16 where the deciding keys are unlike the query, e.g. when the like ones are invalid: 9 of 300 op-fuzz
// cases beyond one bf16 ulp.)  The offsets are added back in fp32 by the two LDS-staged readers:
// S[q][k] = sum_d' E'[q+d'][k+d'] + ea4[q] + eb4[k]; the key half is folded into the key test (kadd2 = kadd + eb4 kmul), the
// query half is a scalar per query.
// att2_emean1/2_kernel: the mean query block, deterministic two-stage sum (class-grid row partials, then rows in order); the
// mean KEY block is taken as rn * mean X (xn = x rn up to its bf16 rounding: the offsets need not be the exact means, only
// the same numbers in E' and in the add-back).  (A first one-stage kernel -- 24 blocks per image looping over all rows -- took
// 192 us at 512x512 B=16, more than fp16 E saves.)
template <bool BF16>
__global__ __launch_bounds__(384) void att2_emean1_kernel(const AttParams p) {
  const int ry = blockIdx.x, b = blockIdx.y, e = threadIdx.x, cls = e / 96, c = e - cls * 96;
  const size_t row0 = ((size_t)(b * p.h + 2 * ry + (cls >> 1)) * p.w + (cls & 1)) * 96 + c;
  float acc = 0.f;
  for (int rx = 0; rx < p.wc; ++rx) {
    const size_t at = row0 + (size_t)rx * 192;
    acc += BF16 ? bf16_lo(((const unsigned short*)p.x)[at]) : p.x[at];
  }
  p.epart[((size_t)b * p.hc + ry) * 384 + e] = acc;
}
__global__ __launch_bounds__(384) void att2_emean2_kernel(const AttParams p) {
  const int b = blockIdx.x, e = threadIdx.x, c = e % 96;
  float a4[4] = {0.f, 0.f, 0.f, 0.f};      // four independent chains: the loop is bound by load latency (16 blocks of 384 threads)
  for (int ry = 0; ry < p.hc; ry += 4)
#pragma unroll
    for (int u = 0; u < 4; ++u)
      if (ry + u < p.hc) a4[u] += p.epart[((size_t)b * p.hc + ry + u) * 384 + e];
  const float acc = (a4[0] + a4[1]) + (a4[2] + a4[3]);
  const float m = acc / (float)p.R;
  p.emean[(size_t)b * 768 + e] = m;
  p.emean[(size_t)b * 768 + 384 + e] = m * p.rn[b * 96 + c];
}
// One wave per class-grid position: ea, eb from the stored (bf16) x and xn.
template <bool BF16>
__global__ __launch_bounds__(256) void att2_eoff_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  const long gi = (long)blockIdx.x * 4 + (threadIdx.x >> 6);      // over B * Rp
  if (gi >= (long)p.B * p.Rp) return;
  const int b = (int)(gi / p.Rp), r = (int)(gi - (long)b * p.Rp);
  const float* mx = p.emean + (size_t)b * 768, *mk = mx + 384;
  float a = 0.f, bb = 0.f, g = 0.f;
  if (r < p.R) {
    const int ry = r / p.wc, rx = r - ry * p.wc;
#pragma unroll
    for (int t = 0; t < 6; ++t) {          // 4 pixels x 96 channels = 384 = 6 x 64
      const int e = t * 64 + lane, cls = e / 96, c = e - cls * 96;
      const size_t at = ((size_t)(b * p.h + 2 * ry + (cls >> 1)) * p.w + 2 * rx + (cls & 1)) * 96 + c;
      const float xv = BF16 ? bf16_lo(((const unsigned short*)p.x)[at]) : p.x[at];
      const float kv = BF16 ? bf16_lo(((const unsigned short*)p.xn)[at]) : p.xn[at];
      a = fmaf(xv, mk[e], a);
      bb = fmaf(mx[e], kv, bb);
      g = fmaf(mx[e], mk[e], g);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { a += __shfl_xor(a, o); bb += __shfl_xor(bb, o); g += __shfl_xor(g, o); }
    bb -= g;
  }
  if (lane == 0) { p.ea[gi] = a; p.eb[gi] = bb; }
}
__global__ __launch_bounds__(256) void att2_eoff4_kernel(const AttParams p) {
  const long gi = (long)blockIdx.x * 256 + threadIdx.x;
  if (gi >= (long)p.B * p.Rp) return;
  const int b = (int)(gi / p.Rp), k = (int)(gi - (long)b * p.Rp);
  const float* ea = p.ea + (size_t)b * p.Rp, *eb = p.eb + (size_t)b * p.Rp;
  auto at = [&](const float* t, int i) { return i < p.R ? t[i] : 0.f; };
  p.ea4[gi] = ((at(ea, k) + at(ea, k + 1)) + at(ea, k + p.wc)) + at(ea, k + p.wc + 1);
  const float b4 = ((at(eb, k) + at(eb, k + 1)) + at(eb, k + p.wc)) + at(eb, k + p.wc + 1);
  p.kadd2[gi] = fmaf(b4, p.kmul[gi], p.kadd[gi]);      // -inf stays -inf (kmul = 0 there), an invalid key stays 0
}

// E[b][r][s] = <2x2x96 block of x at class-grid position r, 2x2x96 block of xn at s>; rows of the A tile = keys s,
// MFMA columns = queries r; written query-major so that the softmax axis is contiguous.
template <int NT, int PT, bool BF16>
__global__ __launch_bounds__(256) void att2_pair_kernel(const AttParams p) {
  constexpr int PXB = BF16 ? 192 : 384;        // bytes per pixel (96 channels)
  constexpr int PIX = PT * 64, NP = NT * 16;
  constexpr int XBYTES = PIX * 128, WBYTES = NP * 128;
  constexpr int NX = PT * 2, NW = (NT * 2 + 3) / 4;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;
  char* Wb = smem + 2 * XBYTES;

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // (an image-major 1-D grid with one contiguous range per XCD was measured: no gain in fp32, 0.47 -> 0.67 ms in bf16
  // where the kernel is bound by the 1 GB of E it writes -- the plain 3-D grid spreads those writes over all XCDs)
  int b = blockIdx.z, bx = blockIdx.x, by = blockIdx.y;
  if (p.sym) {
    // 1-D grid over the computed tiles only (bx <= by / 4), every XCD a contiguous range (xcd_tile).  Order: image, PANEL of 16
    // key tiles (by = 16 pn .. + 15), query tile bx, key tile -- the 64 workgroups resident on an XCD are 4 query tiles x the 16
    // key tiles of one panel: 3.1 MB of operands in its 4 MB L2 for 64 tiles (49 KB per tile instead of 491), and the panel's
    // keys stay there for every bx.  (by-major order without the XCD remap: 967 MB fetched per launch at R = 4096 B=8 for
    // 100 MB of operands, and the kernel is then partly bound by that.)  Panel pn holds 64 pn full tiles + the 16 + 12 + 8 + 4
    // of its diagonal blocks; 32 pn^2 + 8 pn tiles before it.
    static_assert(PIX == 4 * NP, "tile enumeration");
    const int lb = xcd_tile(blockIdx.x, gridDim.x);
    b = lb / p.symT;
    const int t = lb - b * p.symT;
    int pn = (int)((sqrtf(64.f + 128.f * (float)t) - 8.f) * (1.f / 64.f));
    while (32 * (pn + 1) * (pn + 1) + 8 * (pn + 1) <= t) ++pn;
    while (32 * pn * pn + 8 * pn > t) --pn;
    const int u = t - (32 * pn * pn + 8 * pn);
    if (u < 64 * pn) {
      bx = u >> 4;
      by = 16 * pn + (u & 15);
    } else {
      const int v = u - 64 * pn;
      const int j = v < 16 ? 0 : (v < 28 ? 1 : (v < 36 ? 2 : 3));
      bx = 4 * pn + j;
      by = 16 * pn + 4 * j + (v - (j == 0 ? 0 : (j == 1 ? 16 : (j == 2 ? 28 : 36))));
    }
    if (by * NP >= p.R) return;                    // (the last panel is enumerated in full)
  }
  const int q0 = bx * PIX, k0 = by * NP;
  // E is symmetric: E[r][s] = sum_c x[r][c] x[s][c] rn[c] (the key normalisation is per CHANNEL, splitcam.py:40).  With p.sym
  // (fp32 mode) BOTH operands are y = x sqrt(rn) (att2_prep_kernel writes that instead of xn = x rn): the products of (r, s)
  // and (s, r) are then the same numbers in the same k order, E[r][s] == E[s][r] BITWISE, and only the tiles with k0 >= q0 are
  // computed: a tile right of its diagonal block (k0 >= q0 + PIX) also stores its transpose, which is exactly the set of tiles
  // left out.  (Mirroring sum x[r] fl(x[s] rn) was built first: equal to rounding only, and WHICH entries are mirrored depends
  // on the tile shape -- the 128 x 32 tiles of a small call against the 256 x 64 of a large one -- so an image's result depended
  // on the batch it came in: test_full_size_properties.)  53 % of the tiles at R = 4096; the grid holds only those (a 3-D grid
  // whose left-of-diagonal workgroups returned at once measured NO gain: 0.786 -> 0.805 ms).
  // byte offset of the 2x2 block origin of a row inside THIS image, or an out-of-range offset (hardware zero fill)
  auto origin = [&](int i) -> unsigned {
    if (i >= p.R) return 0x80000000u;
    const int ry = i / p.wc, rx = i - ry * p.wc;
    return (unsigned)(((2 * ry) * p.w + 2 * rx) * PXB);
  };
  const se_i32x4 rs_q = make_rsrc((const char*)(p.sym ? p.xn : p.x) + (size_t)b * p.h * p.w * PXB, (unsigned)p.h * p.w * (unsigned)PXB);
  const se_i32x4 rs_k = make_rsrc((const char*)p.xn + (size_t)b * p.h * p.w * PXB, (unsigned)p.h * p.w * (unsigned)PXB);
  unsigned qo[NX], ko[NW];
#pragma unroll
  for (int i = 0; i < NX; ++i) qo[i] = origin(q0 + (i * 4 + w) * 8 + (lane >> 3));
#pragma unroll
  for (int j = 0; j < NW; ++j) ko[j] = (j * 4 + w) < NT * 2 ? origin(k0 + (j * 4 + w) * 8 + (lane >> 3)) : 0x80000000u;

  const int s_log = (lane & 7) ^ (4 * (w & 1) + (lane >> 4));
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const unsigned lds_x = lds_addr_of(Xb), lds_w = lds_addr_of(Wb);

  auto stage = [&](int ch, int buf) {
    const int gi = ch * 8 + s_log;                 // granule of the 4 pixels x 24 (bf16: 12) channel groups
    constexpr int GPP = BF16 ? 12 : 24;            // granules per pixel
    const int tap = (gi * (BF16 ? 5462 : 2731)) >> 16, cg = gi - tap * GPP;      // gi / GPP on the range used
    const unsigned doff = (unsigned)((__mul24(tap >> 1, p.w) + (tap & 1)) * PXB + cg * 16);
    const unsigned xdst = lds_x + buf * XBYTES, wdst = lds_w + buf * WBYTES;
#pragma unroll
    for (int i = 0; i < NX; ++i) bufdma16(qo[i] + doff, rs_q, xdst + (i * 4 + w) * 1024);
#pragma unroll
    for (int j = 0; j < NW; ++j) {
      const int rbk = j * 4 + w;
      if (rbk < NT * 2) bufdma16(ko[j] + doff, rs_k, wdst + rbk * 1024);
    }
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  if (BF16 && p.e16) {
    // fp16-E form: the accumulators start at -(ea[query] + eb[key]), so the doubly centred E' comes out of the GEMM itself
    // (20 loads per lane under the first stage; an epilogue that subtracted the offsets row by row cost 90 us at 512x512 B=16)
    const float* ea = p.ea + (size_t)b * p.Rp, *eb = p.eb + (size_t)b * p.Rp;
    float eq[PT];
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) eq[pt] = ea[min(q0 + w * PT * 16 + pt * 16 + (lane & 15), p.Rp - 1)];
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 ek = *(const f32x4*)(eb + min(k0 + nt * 16 + (lane >> 4) * 4, p.Rp - 4));
#pragma unroll
      for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){-(eq[pt] + ek[0]), -(eq[pt] + ek[1]), -(eq[pt] + ek[2]), -(eq[pt] + ek[3])};
    }
  }
  constexpr int NCH = BF16 ? 6 : 12;   // 4 pixels * 96 ch / (64 | 32)
  stage(0, 0);
  dma_wait_all();
  __syncthreads();
  for (int ch = 0; ch < NCH; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < NCH) stage(ch + 1, buf ^ 1);
    if (BF16) mfma_chunk16<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    else mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    dma_wait_all();
    __syncthreads();
  }
  // E tile out through LDS (the staging buffers are free after the last barrier): an accumulator holds 4 consecutive keys
  // of ONE query per lane, so a direct store scatters 64-byte pieces over 16 rows per instruction; transposed, 16 lanes
  // write one query's 256 contiguous bytes (two full cache lines), 4 rows per store instruction.  Each wave owns a
  // private [PT*16 queries][NT*16 keys (+4 pad)] fp32 region.
  constexpr int TS = NT * 16 + 4;                    // padded row stride in floats: breaks the power-of-two bank stride
  static_assert(4 * PT * 16 * TS * 4 <= 2 * XBYTES + 2 * WBYTES, "transpose buffer exceeds the staging buffers");
  float* T = (float*)smem + w * (PT * 16 * TS);
  const int q = lane >> 4;
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) *(f32x4*)(T + (pt * 16 + (lane & 15)) * TS + nt * 16 + q * 4) = acc[nt][pt];
  // (wave-private region: no barrier, the LDS operations of one wave complete in order)
  constexpr int PPR = NT * 4;                        // 16-byte pieces per row
  constexpr int RPI = 64 / PPR;                      // rows per store instruction
  const int pc = lane % PPR, pr = lane / PPR;
  const int j = k0 + pc * 4;
#pragma unroll
  for (int rr = 0; rr < PT * 16; rr += RPI) {
    const int row = rr + pr;
    const f32x4 v = *(const f32x4*)(T + row * TS + pc * 4);
    const int i = q0 + w * PT * 16 + row;
    if (i < p.R && j < p.Rp) {      // streaming store (NT_STORE note at the top of this section)
      if (BF16 && p.e16) {          // E in fp16 (bf16 mode with the LDS-staged fused passes): 8 bytes per lane
        typedef _Float16 h4 __attribute__((ext_vector_type(4)));
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        // (already centred: accumulator start.)  Saturated to the fp16 range: a centred score beyond +-65504 would
        // become inf and the LDS-staged passes would form inf - inf = NaN for the whole row; clamped, such an entry still
        // dominates (or vanishes from) its softmax row exactly as before -- 2^(10 * 65504 * log2 e) is out of range either way.
        auto sat = [](float t) { return __builtin_amdgcn_fmed3f(t, -65504.f, 65504.f); };
        const h4 hv = (h4){(_Float16)sat(v[0]), (_Float16)sat(v[1]), (_Float16)sat(v[2]), (_Float16)sat(v[3])};
        __builtin_nontemporal_store(__builtin_bit_cast(u32x2, hv), (u32x2*)((char*)p.E + (((size_t)b * p.R + i) * p.Rp + j) * 2));
      } else __builtin_nontemporal_store(v, (f32x4*)(p.E + ((size_t)b * p.R + i) * p.Rp + j));
    }
  }
  if (!BF16 && p.sym && k0 >= q0 + PIX) {
    // the transposed tile: row = key k0 + kk, 16 lanes write its PT*16 queries (of this wave) as 16-byte pieces
    constexpr int PPM = PT * 4;                      // pieces per mirrored row
    constexpr int RPM = 64 / PPM;                    // mirrored rows per store instruction
    const int mc = lane % PPM, mr = lane / PPM;
    const int qcol = q0 + w * PT * 16 + mc * 4;
#pragma unroll
    for (int kk0 = 0; kk0 < NT * 16; kk0 += RPM) {
      const int kk = kk0 + mr;
      const float* t = T + (mc * 4) * TS + kk;
      const f32x4 v = (f32x4){t[0], t[TS], t[2 * TS], t[3 * TS]};
      const int srow = k0 + kk;
      if (srow < p.R && qcol < p.Rp) __builtin_nontemporal_store(v, (f32x4*)(p.E + ((size_t)b * p.R + srow) * p.Rp + qcol));
    }
  }
}

// Row order of the two streaming passes.  Row r of E is read by the 4 queries r - d, row i of P by the 4 rows i + d: with
// one wave per row in plain row-major order the second pair of readers comes a whole grid row later, on another XCD, and
// every row is fetched from HBM ~4 times (measured: 2.8 TB/s of traffic for 1 GB of E).  Waves are therefore assigned
// in 8x8 tiles of the grid, and every XCD walks a contiguous range of tiles (xcd_tile): the readers of a row run within
// a few blocks of each other on one L2.  q -> (image, y, x) on an ny x nx grid; false for the padding of ragged tiles.
DEVFN bool tile_order(long q, int B, int ny, int nx, int& b, int& y, int& x) {
  const int ty = (ny + 7) >> 3, tx = (nx + 7) >> 3;
  const long per = (long)ty * tx * 64;
  b = (int)(q / per);
  const int rem = (int)(q - (long)b * per), t = rem >> 6, in = rem & 63;
  y = (t / tx) * 8 + (in >> 3);
  x = (t % tx) * 8 + (in & 7);
  return b < B && y < ny && x < nx;
}
static inline long tile_order_count(int B, int ny, int nx) { return (long)B * ((ny + 7) >> 3) * ((nx + 7) >> 3) * 64; }

// One wave per query i: S[j] = scale * valid[j] * sum_d E[i+d][j+d] (splitcam.py:69,90,104), softmax over the keys
// (:105); written in class-grid indexing, zero at the grid positions that are not keys and in the pad columns.
// P is fp32, or bf16 in bf16 mode (the oracle rounds P before the reconstruction, sketchedit_oracle.py
// attention_reconstruct).  fp32: the row of P is the scratch for the scores; bf16: the scores are recomputed in each
// of the three sweeps (this kernel only runs for rows that do not fit the register form: feature maps above 128 x 128).
template <bool BF16>
__global__ __launch_bounds__(256) void att2_softmax_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  int b, iy, ix;
  if (!tile_order((long)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), p.B, p.hs, p.ws, b, iy, ix)) return;
  const int r0 = iy * p.wc + ix;
  const float* E0 = p.E + ((size_t)b * p.R + r0) * p.Rp;      // (iy, ix)
  const float* E1 = E0 + p.Rp;                                // (iy, ix+1)
  const float* E2 = E0 + (size_t)p.wc * p.Rp;                 // (iy+1, ix)
  const float* E3 = E2 + p.Rp;                                // (iy+1, ix+1)
  const float* vr = p.validR + (size_t)b * p.Rp;
  float* P = p.P + ((size_t)b * p.R + r0) * p.Rp;
  unsigned short* P16 = (unsigned short*)p.P + ((size_t)b * p.R + r0) * p.Rp;
  const int o2 = p.wc, o3 = p.wc + 1;
  auto score = [&](int s, float v) { return (E0[s] + E1[s + 1] + E2[s + o2] + E3[s + o3]) * v * p.scale; };
  float m = -INFINITY;
  for (int s = lane; s < p.R; s += 64) {
    const float v = vr[s];
    if (v >= 0.f) {      // a key: sy < hc-1 and sx < wc-1, so s + wc + 1 < R
      const float sc = score(s, v);
      if (!BF16) P[s] = sc;
      m = fmaxf(m, sc);
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
  for (int s = lane; s < p.R; s += 64) {
    const float v = vr[s];
    if (v >= 0.f) {
      const float e = expf((BF16 ? score(s, v) : P[s]) - m);      // fp32: each lane re-reads only what it wrote itself
      if (!BF16) P[s] = e;
      sum += e;
    }
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  for (int s = lane; s < p.Rp; s += 64) {
    float o = 0.f;
    const float v = s < p.R ? vr[s] : -1.f;
    if (v >= 0.f) o = (BF16 ? expf(score(s, v) - m) : P[s]) * inv;
    if (BF16) P16[s] = (unsigned short)(pack_bf16x2(o, 0.f) & 0xffffu);
    else P[s] = o;
  }
}

// The same with the whole row in registers (R <= 64 * NV): E is read once, P written once.  (A lane owning pairs of
// adjacent columns -- 8-byte loads, half as many instructions -- was measured: 1.6x slower, the shifted operands are
// 4-byte aligned and every such load splits.)
template <int NV, bool BF16>
__global__ __launch_bounds__(256) void att2_softmax_reg_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  int b, iy, ix;
  if (!tile_order((long)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), p.B, p.hs, p.ws, b, iy, ix)) return;
  const int r0 = iy * p.wc + ix;
  const float* E0 = p.E + ((size_t)b * p.R + r0) * p.Rp;
  const float* E1 = E0 + p.Rp;
  const float* E2 = E0 + (size_t)p.wc * p.Rp;
  const float* E3 = E2 + p.Rp;
  const float* vr = p.validR + (size_t)b * p.Rp;
  char* P = (char*)p.P + ((size_t)b * p.R + r0) * p.Rp * (BF16 ? 2 : 4);
  const int o2 = p.wc, o3 = p.wc + 1;
  float v[NV];
  float m = -INFINITY;
  // Every load is unconditional (clamped index, the value is selected afterwards) and the loads of 16 columns are
  // issued as one batch in front of a scheduling barrier: left alone, hipcc emits load / wait / add for each of the 5
  // operands of each column in turn -- ~130 dependent memory round trips per wave -- and the pass is latency-bound.
  const int smax = p.R - 1 - o3;              // the largest key position: s + wc + 1 stays inside the row
  constexpr int BATCH = 16;
#pragma unroll
  for (int k0 = 0; k0 < NV; k0 += BATCH) {
    float a0[BATCH], a1[BATCH], a2[BATCH], a3[BATCH], vv[BATCH];
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int sl = min(lane + 64 * (k0 + j), smax);
      vv[j] = vr[sl];
      a0[j] = E0[sl];
      a1[j] = E1[sl + 1];
      a2[j] = E2[sl + o2];
      a3[j] = E3[sl + o3];
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int j = 0; j < BATCH; ++j) {
      const int s = lane + 64 * (k0 + j);
      const float sc = (a0[j] + a1[j] + a2[j] + a3[j]) * vv[j] * p.scale;
      v[k0 + j] = (s <= smax && vv[j] >= 0.f) ? sc : -INFINITY;      // not a key: exp() below gives 0
      m = fmaxf(m, v[k0 + j]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
  float sum = 0.f;
#pragma unroll
  for (int k = 0; k < NV; ++k) {
    v[k] = expf(v[k] - m);
    sum += v[k];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  const float inv = 1.f / sum;
  if (BF16) {
    // columns lane + 64k: neighbours in a row are neighbour lanes.  One exchange per two k: the even lanes pack and store
    // the pairs of row k, the odd lanes those of row k + 1 (4-byte stores; Rp is a multiple of 64 in bf16 mode)
#pragma unroll
    for (int k = 0; k < NV; k += 2) {
      const float a = v[k] * inv, c = v[k + 1] * inv;
      const float got = __shfl_xor((lane & 1) ? a : c, 1);
      const int s = (lane & 1) ? lane - 1 + 64 * (k + 1) : lane + 64 * k;
      if (s < p.Rp) __builtin_nontemporal_store((lane & 1) ? pack_bf16x2(got, c) : pack_bf16x2(a, got), (unsigned*)(P + s * 2));
    }
  } else {
#pragma unroll
    for (int k = 0; k < NV; ++k) {
      const int s = lane + 64 * k;
      if (s < p.Rp) __builtin_nontemporal_store(v[k] * inv, (float*)(P + s * 4));
    }
  }
}

// One wave per class-grid row r: P~[r][s] = sum over the <= 4 patches (query r-d, key s-d) that pair pixel r with pixel s.
// P is zero at non-key columns, so only s - d < 0 needs a test on the key side.  Overwrites E.  P and P~ are fp32, or
// bf16 in bf16 mode (P~: the sum of the four rounded P values, rounded once, packed into the front half of E).
// This form (any wc): two adjacent columns per lane, element loads.
template <bool BF16>
__global__ __launch_bounds__(256) void att2_boxsum_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  int b, ry, rx;
  if (!tile_order((long)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), p.B, p.hc, p.wc, b, ry, rx)) return;
  const int r = ry * p.wc + rx;
  constexpr int ES = BF16 ? 2 : 4;
  const char* Pb = (const char*)p.P + (size_t)b * p.R * p.Rp * ES;
  char* out = (char*)p.E + ((size_t)b * p.R + r) * p.Rp * ES;
  const char* src[4];
  int off[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int dy = d >> 1, dx = d & 1, qy = ry - dy, qx = rx - dx;
    const bool ok = qy >= 0 && qy < p.hs && qx >= 0 && qx < p.ws;       // wave-uniform
    src[d] = ok ? Pb + (size_t)(qy * p.wc + qx) * p.Rp * ES : nullptr;
    off[d] = dy * p.wc + dx;
  }
  // 4 column pairs per batch: 32 unconditional loads in front of a scheduling barrier (see att2_softmax_reg_kernel); a
  // row that does not exist (src null, wave-uniform) reads row 0 of P and is masked
  const char* rowp[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) rowp[d] = src[d] ? src[d] : Pb;
  for (int s00 = 2 * lane; s00 < p.Rp; s00 += 512) {        // Rp is even
    float v[4][2][4];
#pragma unroll
    for (int g = 0; g < 4; ++g)
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < 4; ++d) {
          const int col = min(max(s00 + g * 128 + u - off[d], 0), p.Rp - 1);
          if (BF16) v[g][u][d] = bf16_lo(((const unsigned short*)rowp[d])[col]);
          else v[g][u][d] = ((const float*)rowp[d])[col];
        }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int s0 = s00 + g * 128;
      if (s0 >= p.Rp) break;
      float a[2] = {0.f, 0.f};
#pragma unroll
      for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int d = 0; d < 4; ++d) a[u] += (src[d] && s0 + u >= off[d] && s0 + u < p.R) ? v[g][u][d] : 0.f;
      if (BF16) *(unsigned*)(out + s0 * 2) = pack_bf16x2(a[0], a[1]);
      else *(float2*)(out + s0 * 4) = make_float2(a[0], a[1]);
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// The same for wc % 4 == 0 (every standard size): four adjacent columns per lane.  The aligned start of the group in
// source row d is s0 - dy * wc (a multiple of 4: one 16-byte / 8-byte load); the dx = 1 rows need the element in front
// of it as well (one more element load): 6 loads per 4 columns instead of 16.
template <bool BF16>
__global__ __launch_bounds__(256) void att2_boxsum4_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  int b, ry, rx;
  if (!tile_order((long)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), p.B, p.hc, p.wc, b, ry, rx)) return;
  const int r = ry * p.wc + rx;
  constexpr int ES = BF16 ? 2 : 4;
  // columns per lane: 4 (fp32, 16-byte loads) / 8 (bf16, 16-byte loads: with 8-byte loads the pass issues twice the
  // vector-memory instructions per byte and runs at 3.0 instead of ~4 TB/s); G groups in flight per lane
  constexpr int W = BF16 ? 8 : 4, G = BF16 ? 2 : 4;
  const char* Pb = (const char*)p.P + (size_t)b * p.R * p.Rp * ES;
  char* out = (char*)p.E + ((size_t)b * p.R + r) * p.Rp * ES;
  bool ok[4];
  const char* rowp[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int dy = d >> 1, dx = d & 1, qy = ry - dy, qx = rx - dx;
    ok[d] = qy >= 0 && qy < p.hs && qx >= 0 && qx < p.ws;               // wave-uniform
    rowp[d] = ok[d] ? Pb + (size_t)(qy * p.wc + qx) * p.Rp * ES : Pb;   // a row that does not exist reads row 0, masked
  }
  for (int s00 = W * lane; s00 < p.Rp; s00 += 64 * W * G) {      // Rp is a multiple of 32 (bf16: of 64)
    float q[G][4][W];
    float e[G][2];
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int s0 = min(s00 + g * 64 * W, p.Rp - W);
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int a0 = max(s0 - (d >> 1) * p.wc, 0);
        if (BF16) {
          const uint4 t = *(const uint4*)(rowp[d] + a0 * 2);
          q[g][d][0] = bf16_lo(t.x); q[g][d][1] = bf16_hi(t.x); q[g][d][2] = bf16_lo(t.y); q[g][d][3] = bf16_hi(t.y);
          q[g][d][4 % W] = bf16_lo(t.z); q[g][d][5 % W] = bf16_hi(t.z); q[g][d][6 % W] = bf16_lo(t.w); q[g][d][7 % W] = bf16_hi(t.w);
          if (d & 1) e[g][d >> 1] = bf16_lo(((const unsigned short*)rowp[d])[max(a0 - 1, 0)]);
        } else {
          const f32x4 t = *(const f32x4*)(rowp[d] + a0 * 4);
          q[g][d][0] = t[0]; q[g][d][1] = t[1]; q[g][d][2] = t[2]; q[g][d][3] = t[3];
          if (d & 1) e[g][d >> 1] = ((const float*)rowp[d])[max(a0 - 1, 0)];
        }
      }
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int g = 0; g < G; ++g) {
      const int s0 = s00 + g * 64 * W;
      if (s0 >= p.Rp) break;
      float a[W];
#pragma unroll
      for (int u = 0; u < W; ++u) a[u] = 0.f;
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int off = (d >> 1) * p.wc + (d & 1);
#pragma unroll
        for (int u = 0; u < W; ++u) {
          const float val = (d & 1) ? (u == 0 ? e[g][d >> 1] : q[g][d][u - 1]) : q[g][d][u];
          a[u] += (ok[d] && s0 + u >= off && s0 + u < p.R) ? val : 0.f;
        }
      }
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      if (BF16) __builtin_nontemporal_store((u32x4){pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]), pack_bf16x2(a[4 % W], a[5 % W]), pack_bf16x2(a[6 % W], a[7 % W])}, (u32x4*)(out + s0 * 2));
      else __builtin_nontemporal_store((f32x4){a[0], a[1], a[2], a[3]}, (f32x4*)(out + s0 * 4));
    }
    __builtin_amdgcn_sched_barrier(0);
  }
}

// =====================================================================================================================
// Fused form of the two streaming passes (round 3, the default in fp32): P is never written.
//   att2_stats_kernel : one wave per query i: row maximum and 1 / row sum of the exponentials   (E read once, 8 bytes out)
//   att2_ptilde_kernel: one wave per class-grid row r: P~[r][s] = sum_d exp(S[r-d][s-d] - m_{r-d}) / l_{r-d}, with
//                       S[r-d][s-d] = scale * valid[s-d] * sum_{d'} E[r-d+d'][s-d+d'] formed on the fly from the 3 x 3
//                       neighbourhood T[dl] = E[r+dl][s+dl], dl in {-1,0,1}^2 (row and column shifted TOGETHER).
// Against the three-pass form (E -> P, P -> P~) this removes one full R x R write and one full read from HBM (P), at
// the price of 4 exps and 15 (mostly L2-hit) loads per 4 outputs.  S, the normalisation and the order of the four-term
// sum are those of att2_softmax_reg_kernel + att2_boxsum4_kernel; the exponential is v_exp_f32 instead of ocml's expf
// (16 of them per 4 outputs: the full-precision routine made the pass VALU-bound), so the two forms agree to ~1e-7
// relative, not bit for bit.  The three-pass form (SE_ATT_FUSED=0) is kept for `similar_out` (it needs P) and A/B runs.
// =====================================================================================================================
// stats[q] = (m2, 1 / l): m2 = max_j t[j], l = sum_j exp2(t[j] - m2), t[j] = fma(sumE[j], kmul[j], kadd[j]) = S[q][j] * log2(e), or
// -inf where j is not a key (the key tables are described at att2_ptilde4_kernel).
template <int NV, bool BF16>
__global__ __launch_bounds__(256) void att2_stats_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  int b, iy, ix;
  if (!tile_order((long)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), p.B, p.hs, p.ws, b, iy, ix)) return;
  const int r0 = iy * p.wc + ix;
  const float* E0 = p.E + ((size_t)b * p.R + r0) * p.Rp;
  const float* E1 = E0 + p.Rp + 1;                             // (iy, ix+1), column shift folded in
  const float* E2 = E0 + (size_t)p.wc * p.Rp + p.wc;           // (iy+1, ix)
  const float* E3 = E2 + p.Rp + 1;                             // (iy+1, ix+1)
  const float* km = p.kmul + (size_t)b * p.Rp;
  const float* ka = p.kadd + (size_t)b * p.Rp;
  const int clast = p.Rp - 1;                 // never a key (last class-grid position or a pad column): kadd = -inf there
  float m = -INFINITY, sum = 0.f;
  if constexpr (NV > 0) {                     // the whole row in registers: E is read once
    float v[NV];
    // Every load is unconditional (clamped index) and the loads of 16 columns are issued as one batch in front of a
    // scheduling barrier: left alone, hipcc emits load / wait / add for each operand of each column in turn and the
    // pass is latency-bound.
    constexpr int BATCH = 16;
#pragma unroll
    for (int k0 = 0; k0 < NV; k0 += BATCH) {
      float a0[BATCH], a1[BATCH], a2[BATCH], a3[BATCH], mu[BATCH], ad[BATCH];
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        const int sl = min(lane + 64 * (k0 + j), clast);
        mu[j] = km[sl];
        ad[j] = ka[sl];
        a0[j] = E0[sl];
        a1[j] = E1[sl];
        a2[j] = E2[sl];
        a3[j] = E3[sl];
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int j = 0; j < BATCH; ++j) {
        v[k0 + j] = fmaf(((a0[j] + a1[j]) + a2[j]) + a3[j], mu[j], ad[j]);
        m = fmaxf(m, v[k0 + j]);
      }
      __builtin_amdgcn_sched_barrier(0);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
#pragma unroll
    for (int k = 0; k < NV; ++k) sum += __builtin_amdgcn_exp2f(v[k] - m);
    // (columns >= Rp were clamped onto the never-a-key column Rp - 1: -inf, exp2 gives 0)
  } else {                                    // rows that do not fit in registers: two sweeps
    auto tval = [&](int s) { return fmaf(((E0[s] + E1[s]) + E2[s]) + E3[s], km[s], ka[s]); };
    for (int s = lane; s < p.Rp; s += 64) m = fmaxf(m, tval(s));
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    for (int s = lane; s < p.Rp; s += 64) sum += __builtin_amdgcn_exp2f(tval(s) - m);
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o);
  if (lane == 0) *(float2*)(p.stats + ((size_t)b * p.R + r0) * 2) = make_float2(m, 1.f / sum);
}

// No per-lane clamps or range tests: E and the key tables are allocated with guard bands (AttParams::guard floats in
// front of and behind E -- zero-filled -- and in front of validR / kmul / kadd), a shifted column that leaves its row
// lands in a neighbouring row or in a guard, and every such (finite) value only feeds a term whose key test fails.
// The key test is arithmetic: kmul[s] = scale * log2(e) for a valid key, 0 for an invalid one (multiplicative zero,
// splitcam.py:90,104) or a position that is not a key; kadd[s] = 0 for a key, -inf otherwise, so that
//   P[r-d][s-d] = exp2(fma(sumE, kmul, kadd - m log2 e)) / l   is exactly 0 where s - d is not a key.
// The exponentials run on v_exp_f32 (~1 ulp; att2_stats_kernel forms the row sums with the same expression, so the
// probabilities of a row still sum to 1 to rounding).
//
// att2_ptilde4_kernel (wc % 4 == 0: every standard size): four adjacent columns per lane.  What bounds this pass is VALU
// issue (16 S values, 16 exponentials and 16 scalings per lane and group), so everything that is not an exponential
// is PACKED two columns per instruction (v_pk_add / v_pk_fma / v_pk_mul_f32), which needs every operand as an aligned
// register pair (column 2k, column 2k+1).  The column-shifted operands E[.][s0 - 1 ..] and E[.][s0 + 1 ..] are therefore
// fetched by their own 16-byte loads at 4-byte-aligned addresses (the texture path splits such a load in two, and has
// the room: 13 loads per group against ~150 VALU issue slots) instead of being assembled from an aligned group and a
// neighbour element (lane shifts or strided element loads were measured: the re-pairing moves cost more than the loads).
template <bool BF16>
__global__ __launch_bounds__(512) void att2_ptilde4_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  const int wv = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  int b, ry, rx;
  // a block = 8 waves = a 2 x 4 patch of class-grid rows (two blocks side by side, four pairs of grid rows = one 8 x 8 tile
  // of tile_order): its 24 source rows of E, 1 KB each per column step, fit the CU's 32 KB vector L1, and the waves are
  // kept in step (barrier per step) so that a row one wave pulls in is a hit for the others -- the pass is bound by the
  // L2 -> L1 traffic of its nine source rows per output row, not by HBM, VALU issue or instruction count (all three were
  // varied without effect).
  const int blk = xcd_tile(blockIdx.x, gridDim.x);
  const long q = (long)(blk >> 3) * 64 + ((blk >> 1) & 3) * 16 + (wv >> 2) * 8 + (blk & 1) * 4 + (wv & 3);
  const bool live = tile_order(q, p.B, p.hc, p.wc, b, ry, rx);
  if (!live) { b = 0; ry = 0; rx = 0; }        // padding wave of a ragged tile: walks row 0 in step with the block, stores nothing
  const int r = ry * p.wc + rx;
  constexpr int ES = BF16 ? 2 : 4;
  const float* Eb = p.E + (size_t)b * p.R * p.Rp;
  const float* km = p.kmul + (size_t)b * p.Rp;
  const float* ka = p.kadd + (size_t)b * p.Rp;
  char* out = (char*)p.P + ((size_t)b * p.R + r) * p.Rp * ES;
  // the four queries r - d that own a patch covering class-grid pixel r (wave-uniform): validity, -m2, 1 / sum
  bool okq[4];
  float mq[4], iq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int dy = d >> 1, dx = d & 1, qy = ry - dy, qx = rx - dx;
    okq[d] = qy >= 0 && qy < p.hs && qx >= 0 && qx < p.ws;
    const float2 st = *(const float2*)(p.stats + ((size_t)b * p.R + (okq[d] ? qy * p.wc + qx : 0)) * 2);
    mq[d] = -st.x; iq[d] = st.y;
  }
  // rows r + dly * wc + dlx of E with their column shift folded into the pointer: rowp[a][c][s] = E[r + dl][s + dl]; a row
  // index outside the matrix is clamped (it only ever feeds a term of a query that does not exist)
  const float* rowp[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int dl = (a - 1) * p.wc + (c - 1);
      rowp[a][c] = Eb + (size_t)min(max(r + dl, 0), p.R - 1) * p.Rp + dl;
    }
  struct Grp {
    f32x4 t[3][3];          // t[a][c][u] = E[r + dl][s0 + u + dl]
    f32x4 m[4], ad[4];      // kmul / kadd of key s0 + u - off_d
  };
  auto load = [&](int s0, Grp& g) {
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) g.t[a][c] = *(const f32x4*)(rowp[a][c] + s0);      // c = 1: 16-byte aligned; c = 0, 2: 4-byte aligned
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      const int off = (d >> 1) * p.wc + (d & 1);
      g.m[d] = *(const f32x4*)(km + s0 - off);
      g.ad[d] = *(const f32x4*)(ka + s0 - off);
    }
  };
  auto compute = [&](int s0, const Grp& g) {
    f32x2 acc[2] = {(f32x2){0.f, 0.f}, (f32x2){0.f, 0.f}};
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (!okq[d]) continue;                   // wave-uniform
      const int dy = d >> 1, dx = d & 1;
      const f32x2 mm = (f32x2){mq[d], mq[d]}, ii = (f32x2){iq[d], iq[d]};
#pragma unroll
      for (int h = 0; h < 2; ++h) {            // column pairs (0,1), (2,3)
        auto pr = [&](const f32x4& v) { return (f32x2){v[2 * h], v[2 * h + 1]}; };
        // the 2 x 2 block of T that forms S[r - d][s - d] (dl = d' - d, d' in {0,1}^2), summed in the order E0 + E1 + E2 + E3 of
        // the softmax kernels
        const f32x2 sum = ((pr(g.t[1 - dy][1 - dx]) + pr(g.t[1 - dy][2 - dx])) + pr(g.t[2 - dy][1 - dx])) + pr(g.t[2 - dy][2 - dx]);
        const f32x2 arg = sum * pr(g.m[d]) + (pr(g.ad[d]) + mm);
        f32x2 pv = (f32x2){__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} * ii;
        if (BF16) {                            // P is a bf16 value in bf16 mode (rounded before the sum)
          const unsigned u = pack_bf16x2(pv[0], pv[1]);
          pv = (f32x2){bf16_lo(u), bf16_hi(u)};
        }
        acc[h] += pv;
      }
    }
    if (s0 >= p.Rp || !live) return;
    typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
    if (BF16) __builtin_nontemporal_store((u32x2){pack_bf16x2(acc[0][0], acc[0][1]), pack_bf16x2(acc[1][0], acc[1][1])}, (u32x2*)(out + s0 * 2));
    else __builtin_nontemporal_store((f32x4){acc[0][0], acc[0][1], acc[1][0], acc[1][1]}, (f32x4*)(out + s0 * 4));
  };
  // one column group per lane and step (a second group in flight was measured: no gain, and it doubles the L1 footprint)
  Grp g;
  for (int s0 = 4 * lane; s0 < p.Rp; s0 += 256) {
    load(s0, g);
    __builtin_amdgcn_sched_barrier(0);
    compute(s0, g);
    __builtin_amdgcn_sched_barrier(0);
    __syncthreads();
  }
}

// any wc: one column per lane, element loads, the same arithmetic unpacked
template <bool BF16>
__global__ __launch_bounds__(256) void att2_ptilde1_kernel(const AttParams p) {
  const int lane = threadIdx.x & 63;
  int b, ry, rx;
  if (!tile_order((long)xcd_tile(blockIdx.x, gridDim.x) * 4 + (threadIdx.x >> 6), p.B, p.hc, p.wc, b, ry, rx)) return;
  const int r = ry * p.wc + rx;
  constexpr int ES = BF16 ? 2 : 4;
  const float* Eb = p.E + (size_t)b * p.R * p.Rp;
  const float* km = p.kmul + (size_t)b * p.Rp;
  const float* ka = p.kadd + (size_t)b * p.Rp;
  char* out = (char*)p.P + ((size_t)b * p.R + r) * p.Rp * ES;
  bool okq[4];
  float mq[4], iq[4];
#pragma unroll
  for (int d = 0; d < 4; ++d) {
    const int dy = d >> 1, dx = d & 1, qy = ry - dy, qx = rx - dx;
    okq[d] = qy >= 0 && qy < p.hs && qx >= 0 && qx < p.ws;
    const float2 st = *(const float2*)(p.stats + ((size_t)b * p.R + (okq[d] ? qy * p.wc + qx : 0)) * 2);
    mq[d] = -st.x; iq[d] = st.y;
  }
  const float* rowp[3][3];
#pragma unroll
  for (int a = 0; a < 3; ++a)
#pragma unroll
    for (int c = 0; c < 3; ++c) {
      const int dl = (a - 1) * p.wc + (c - 1);
      rowp[a][c] = Eb + (size_t)min(max(r + dl, 0), p.R - 1) * p.Rp + dl;
    }
  for (int s0 = lane; s0 < p.Rp; s0 += 64) {
    float t[3][3];
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
      for (int c = 0; c < 3; ++c) t[a][c] = rowp[a][c][s0];
    float acc = 0.f;
#pragma unroll
    for (int d = 0; d < 4; ++d) {
      if (!okq[d]) continue;
      const int dy = d >> 1, dx = d & 1, off = dy * p.wc + dx;
      const float sum = ((t[1 - dy][1 - dx] + t[1 - dy][2 - dx]) + t[2 - dy][1 - dx]) + t[2 - dy][2 - dx];
      float pv = __builtin_amdgcn_exp2f(sum * km[s0 - off] + (ka[s0 - off] + mq[d])) * iq[d];
      if (BF16) pv = bf16_lo(pack_bf16x2(pv, 0.f) & 0xffffu);
      acc += pv;
    }
    if (BF16) *(unsigned short*)(out + s0 * 2) = (unsigned short)(pack_bf16x2(acc, 0.f) & 0xffffu);
    else *(float*)(out + s0 * 4) = acc;
  }
}

// att2_ptilde_lds_kernel (round 4): the same pass with its source rows staged in LDS.  att2_ptilde4_kernel
// forms every S[q][k] four times (once per output it feeds) from nine source rows per output row, and is bound by the
// L2 -> L1 traffic of those rows.  Here a workgroup owns a 4 x 4 PATCH of class-grid positions r = (ry0 + a, rx0 + bx) and
// sweeps the key axis in chunks of C = 128 columns:
//   stage  : the 6 x 6 source rows E[(ry0 - 1 + i') wc + rx0 - 1 + j'][s0 .. s0 + C + wc + 4) and the chunk's slice of the key
//            tables, by LDS-DMA (36 rows for 16 output rows instead of 9 per row: 2.25 + halo E elements per output through
//            the vector-memory path instead of 9)
//   phase B: P[q][k] for the 5 x 5 queries q = r - d that own a patch over one of the 16 positions, ONCE each (25 / 16 = 1.6
//            exponentials per output instead of 4), written to a ring of 256 columns per query (P~[r][s] only needs columns
//            s - wc - 1 .. s, all to the left: the ring keeps the tail of the previous chunk, no halo is recomputed)
//   phase C: P~[r][s] = sum_d P[r - d][s - d] from the ring, 16-byte non-temporal stores
// Ten waves: eight consumers (phases B, C) and TWO PRODUCER waves that issue every DMA.  A first version had all waves issue
// their share of the stage after phase B and wait vmcnt(0) before the next: one stage (28 KB) in flight for one phase C --
// far less than bandwidth x latency; it measured 335 us at 512x512 B=8 (round-3 kernel: 410).  Stores share vmcnt with loads
// on gfx9 and complete out of order with them, so a wave that stores cannot use a counted wait; a producer wave issues
// loads only, keeps NS - 1 stages in flight behind the one being consumed and waits vmcnt((NS - 2) x its instructions per
// stage).  One producer (63 instructions in flight at most: the vmcnt counter) measured 278 us; ablations of that version
// (same box): DMA + barriers alone 174 us = 3.1 TB/s of E -- latency x bytes in flight, not HBM -- phase B +36, phase C +51.
// A phase-B task is (query ROW i, column pair): the 12 source rows (i .. i+1, 0 .. 5) are read once for the five queries
// (i, 0 .. 4) -- 14 instead of 36 bytes of LDS traffic per P value.
// Tried and not faster: phases B and C of consecutive chunks side by side (15 waves: 5 on phase B of chunk k, 8 on phase C of
// chunk k - 1, 2 producers; one barrier per chunk, a 512-column ring, 3 stages = 138 KB): 272 vs 266 us at 512x512 B=8 fp32,
// 556 vs 485 us at 512x512 B=16 bf16 with fp32 E (same box) -- one stage fewer in flight costs more than the overlap gains.
// Expression, summation order and key test are those of att2_ptilde4_kernel (results agree to 1 ulp); a query that does not
// exist carries -inf / 0, so the clamped or wrapped source rows it reads (finite) give exactly 0.
// Needs wc % 4 == 0 (aligned LDS reads at column shift wc) and wc <= 124 (ring of 256 >= C + wc + 4).
// NIP = 1 KB DMA instructions per stage (the NR x ceil((C + wc + 4) / elements per 16 bytes) pieces, flat), a compile-time
// bound -- the counted wait needs an immediate.  EH: E holds fp16 (bf16 mode, launch_attention_v2_t): half the bytes per stage.
//
// The stage machinery shared by att2_ptilde_lds_kernel (NR = 36 source rows on a 6-wide grid) and att2_stats_lds_kernel
// (25 rows, 5 wide): producer wave `pid` of two issues instructions k = 2 m + pid of every stage; one barrier (`BARS` = 1) or
// two per chunk on the consumer side.
template <bool EH, int NIP, int NS, int NR, int GW, int BARS>
DEVFN void lds_stage_producer(const AttParams& p, int b, int r00, int pid, int lane, int nchunks, char* smem) {
  constexpr int C = 128, EB = EH ? 2 : 4, EPP = 16 / EB;
  constexpr int SB = (NIP + 1) * 1024;
  constexpr int NH = NIP / 2;                              // E instructions per producer wave and stage
  static_assert(NIP % 2 == 0 && (NS - 2) * (NH + 1) <= 63, "vmcnt immediate");
  const int NPR = (C + p.wc + 4 + EPP - 1) / EPP;          // 16-byte pieces per source row
  const int NPT = NR * NPR;                                // E pieces of a stage (<= NIP * 64)
  const se_i32x4 rsE = make_rsrc((const char*)p.E + (size_t)b * p.R * p.Rp * EB, (unsigned)p.R * (unsigned)p.Rp * (unsigned)EB);
  unsigned goff[NH];                                       // piece -> byte offset inside the image's E
  {
    // row = pi / NPR without an integer division per piece: pieces advance by 128 per instruction of this wave
    int pi = pid * 64 + lane;
    int row = pi / NPR, cc = pi - row * NPR;
    const int drow = 128 / NPR, dcc = 128 - drow * NPR;
#pragma unroll
    for (int m = 0; m < NH; ++m) {
      const int i = row / GW, j = row - i * GW;
      const int er = min(max(r00 + i * p.wc + j, 0), p.R - 1);
      goff[m] = pi < NPT ? (unsigned)er * (unsigned)p.Rp * (unsigned)EB + (unsigned)cc * 16u : 0x80000000u;      // padding pieces: zeros
      pi += 128; row += drow; cc += dcc;
      if (cc >= NPR) { cc -= NPR; row += 1; }
    }
  }
  const float* ktab = (lane < 32 ? p.kmul : (EH ? p.kadd2 : p.kadd)) + (size_t)b * p.Rp;
  const int kl = 4 * (lane & 31);
  const unsigned lds_e = lds_addr_of(smem);
  auto issue = [&](int n) {
    const unsigned base = lds_e + (unsigned)(n % NS) * SB + pid * 1024;
    const unsigned s0b = (unsigned)(n * C) * (unsigned)EB;
#pragma unroll
    for (int m = 0; m < NH; ++m) bufdma16(__builtin_elementwise_add_sat(goff[m], s0b), rsE, base + m * 2048);
    if (pid == 0) glds16(ktab + min(n * C + kl, p.Rp - 4), lds_e + (unsigned)(n % NS) * SB + NIP * 1024);      // [kmul 128][kadd 128]
  };
#pragma unroll
  for (int n = 0; n < NS - 1; ++n)
    if (n < nchunks) issue(n);
  for (int n = 0; n < nchunks; ++n) {
    const int y = min(NS - 2, nchunks - 1 - n);            // stages younger than n in flight
    if (y >= NS - 2) {
      if (pid == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * (NH + 1)) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"((NS - 2) * NH) : "memory");
    } else if (NS == 4 && y == 1) {
      if (pid == 0) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH + 1) : "memory");
      else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NH) : "memory");
    } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                 // top: stage n is complete; the consumers are done with stage n - 1
#ifndef PT_NODMA
    if (n + NS - 1 < nchunks) issue(n + NS - 1);           // into the buffer of stage n - 1
#endif
    if (BARS == 2) __syncthreads();  // mid
  }
}
// three consecutive elements c, c + 1, c + 2 (c even) of a staged source row
template <bool EH>
DEVFN void lds_row3(const char* row, int c, f32x2& v01, float& v2) {
  if (EH) {
    typedef _Float16 h2 __attribute__((ext_vector_type(2)));
    const h2 a = *(const h2*)(row + c * 2), bb = *(const h2*)(row + c * 2 + 4);
    v01 = (f32x2){(float)a[0], (float)a[1]};
    v2 = (float)bb[0];
  } else {
    v01 = *(const f32x2*)(row + c * 4);
    v2 = *(const float*)(row + c * 4 + 8);
  }
}

template <bool BF16, bool EH, int NIP, int NS>
__global__ __launch_bounds__(640) void att2_ptilde_lds_kernel(const AttParams p, int npy, int npx) {
  constexpr int C = 128, NG = C / 4, RING = 256;
  constexpr int ES = BF16 ? 2 : 4, EB = EH ? 2 : 4, EPP = 16 / EB;
  constexpr int SB = (NIP + 1) * 1024;                     // one stage: E pieces + (kmul, kadd) slice
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int WSB = ((C + p.wc + 4 + EPP - 1) / EPP) * 16;   // bytes per staged source row
  float* Pr = (float*)(smem + NS * SB);                    // [25][RING]
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lb = xcd_tile(blockIdx.x, gridDim.x);
  const int b = lb / (npy * npx), rem = lb - b * (npy * npx);
  const int py = rem / npx, px = rem - py * npx;
  const int ry0 = 4 * py, rx0 = 4 * px;
  const int nchunks = (p.Rp + C - 1) / C;

  if (w >= 8) {      // source row (i', j') = r00 + i' wc + j' (linear, as r + dl in att2_ptilde4_kernel)
    lds_stage_producer<EH, NIP, NS, 36, 6, 2>(p, b, (ry0 - 1) * p.wc + rx0 - 1, w - 8, lane, nchunks, smem);
    return;
  }

  // ---------------- consumer waves (512 threads)
  // ring zeroed (columns < 0 are not keys: P = 0)
  for (int i = tid; i < 25 * RING / 4; i += 512) ((f32x4*)Pr)[i] = (f32x4){0.f, 0.f, 0.f, 0.f};
  // phase-B role: waves 0..4 = query row i, lane = column pair; the five queries (i, 0..4): -m2 (or -inf), 1 / l (or 0)
  float qm[5], qi[5], qc[5];           // (qc: the query half of the fp16-E offset, ea4[q]; 0 with fp32 E)
#pragma unroll
  for (int j = 0; j < 5; ++j) {
    const int qy = ry0 - 1 + w, qx = rx0 - 1 + j;
    const bool ok = w < 5 && qy >= 0 && qy < p.hs && qx >= 0 && qx < p.ws;
    const float2 st = *(const float2*)(p.stats + ((size_t)b * p.R + (ok ? qy * p.wc + qx : 0)) * 2);
    qm[j] = ok ? -st.x : -INFINITY;
    qi[j] = ok ? st.y : 0.f;
    qc[j] = (EH && ok) ? p.ea4[(size_t)b * p.Rp + qy * p.wc + qx] : 0.f;
  }
  for (int n = 0; n < nchunks; ++n) {
    const int s0 = n * C;
    const char* Eb = smem + (n % NS) * SB;
    const float* Kt = (const float*)(smem + (n % NS) * SB + NIP * 1024);
    __syncthreads();                 // top
    // ---- phase B
#ifndef PT_NOB
    if (w < 5) {
      const int c = 2 * lane, s = s0 + c;
      f32x2 r0[6], r1[6];
      float x0[6], x1[6];
#pragma unroll
      for (int j = 0; j < 6; ++j) {
        lds_row3<EH>(Eb + (w * 6 + j) * WSB, c, r0[j], x0[j]);                  // (i, j')      columns c, c + 1 | c + 2
        lds_row3<EH>(Eb + (w * 6 + 6 + j) * WSB, c + p.wc, r1[j], x1[j]);       // (i + 1, j')  columns c + wc, c + wc + 1 | c + wc + 2
      }
      const f32x2 mu = *(const f32x2*)(Kt + c), ad = *(const f32x2*)(Kt + C + c);
#pragma unroll
      for (int j = 0; j < 5; ++j) {
        // S[(i, j)][k] = ((E[(i,j)][k] + E[(i,j+1)][k+1]) + E[(i+1,j)][k+wc]) + E[(i+1,j+1)][k+wc+1]
        f32x2 sum = ((r0[j] + (f32x2){r0[j + 1][1], x0[j + 1]}) + r1[j]) + (f32x2){r1[j + 1][1], x1[j + 1]};
        if (EH) sum += (f32x2){qc[j], qc[j]};
        const f32x2 arg = sum * mu + (ad + (f32x2){qm[j], qm[j]});
        f32x2 pv = (f32x2){__builtin_amdgcn_exp2f(arg[0]), __builtin_amdgcn_exp2f(arg[1])} * (f32x2){qi[j], qi[j]};
        if (BF16) {                                            // P is a bf16 value in bf16 mode (rounded before the sum)
          const unsigned u = pack_bf16x2(pv[0], pv[1]);
          pv = (f32x2){bf16_lo(u), bf16_hi(u)};
        }
        if (s >= p.Rp) pv = (f32x2){0.f, 0.f};
        *(f32x2*)(Pr + (w * 5 + j) * RING + (s & (RING - 1))) = pv;
      }
    }
#endif
    __syncthreads();                 // mid: the ring holds this chunk; the stage buffer is free
    // ---- phase C: task = (output position o of the patch, group g of four columns)
#ifndef PT_NOC
    {
      const int o = tid / NG, g = tid - o * NG;
      const int a = o >> 2, bx = o & 3;
      const int s = s0 + 4 * g;
      f32x4 acc = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int d = 0; d < 4; ++d) {
        const int dy = d >> 1, dx = d & 1;
        const float* row = Pr + ((a + 1 - dy) * 5 + (bx + 1 - dx)) * RING;
        const int cb = (s - dy * p.wc) & (RING - 1);
        const f32x4 v = *(const f32x4*)(row + cb);
        if (dx == 0) acc += v;
        else acc += (f32x4){row[(cb - 1) & (RING - 1)], v[0], v[1], v[2]};
      }
      if (ry0 + a < p.hc && s < p.Rp) {
        const int r = (ry0 + a) * p.wc + rx0 + bx;
        char* out = (char*)p.P + (((size_t)b * p.R + r) * p.Rp + s) * ES;
        typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
        if (BF16) __builtin_nontemporal_store((u32x2){pack_bf16x2(acc[0], acc[1]), pack_bf16x2(acc[2], acc[3])}, (u32x2*)out);
        else __builtin_nontemporal_store(acc, (f32x4*)out);
      }
    }
#endif
  }
}

// att2_stats_lds_kernel: the statistics pass on the same stage machinery.  att2_stats_kernel reads every row of E for each of
// the four queries it feeds (through L2: 4 x the bytes of E); here a workgroup owns a 4 x 4 patch of QUERIES, stages the
// 5 x 5 source rows (25 / 16 = 1.6 x) and keeps an online (maximum, sum) per (query, lane):
//   m' = max(m, t0, t1), l = l exp2(m - m') + exp2(t0 - m') + exp2(t1 - m'),    t = fma(sumE, kmul, kadd)  (-inf: not a key)
// with m starting at a finite -1e30 (exp2(-inf - m') = 0, never inf - inf).  The softmax is invariant to the shift as long as
// l is formed with the same m, so the pair only has to be CONSISTENT; it differs from att2_stats_kernel's (row maximum, one
// sum in column order) in rounding.  Four consumer waves (wave = query row i, lane = column pair) + two producers, one
// barrier per chunk; 3 stages, two workgroups per CU.
template <bool EH, int NIP, int NS>
__global__ __launch_bounds__(384) void att2_stats_lds_kernel(const AttParams p, int npy, int npx) {
  constexpr int C = 128;
  constexpr int EB = EH ? 2 : 4, EPP = 16 / EB;
  constexpr int SB = (NIP + 1) * 1024;
  extern __shared__ __attribute__((aligned(16))) char smem[];
  const int WSB = ((C + p.wc + 4 + EPP - 1) / EPP) * 16;
  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int lb = xcd_tile(blockIdx.x, gridDim.x);
  const int b = lb / (npy * npx), rem = lb - b * (npy * npx);
  const int py = rem / npx, px = rem - py * npx;
  const int qy0 = 4 * py, qx0 = 4 * px;
  const int nchunks = (p.Rp + C - 1) / C;
  if (w >= 4) {
    lds_stage_producer<EH, NIP, NS, 25, 5, 1>(p, b, qy0 * p.wc + qx0, w - 4, lane, nchunks, smem);
    return;
  }
  float m_[4] = {-1e30f, -1e30f, -1e30f, -1e30f}, l_[4] = {0.f, 0.f, 0.f, 0.f};      // online (max, sum) of queries (w, 0..3), this lane's columns
  float qc[4];                         // the query half of the fp16-E offset (0 with fp32 E)
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int qy = qy0 + w, qx = qx0 + j;
    qc[j] = (EH && qy < p.hs && qx < p.ws) ? p.ea4[(size_t)b * p.Rp + qy * p.wc + qx] : 0.f;
  }
  for (int n = 0; n < nchunks; ++n) {
    const char* Eb = smem + (n % NS) * SB;
    const float* Kt = (const float*)(smem + (n % NS) * SB + NIP * 1024);
    __syncthreads();                 // top
    const int c = 2 * lane;
    f32x2 r0[5], r1[5];
    float x0[5], x1[5];
#pragma unroll
    for (int j = 0; j < 5; ++j) {
      lds_row3<EH>(Eb + (w * 5 + j) * WSB, c, r0[j], x0[j]);
      lds_row3<EH>(Eb + (w * 5 + 5 + j) * WSB, c + p.wc, r1[j], x1[j]);
    }
    const f32x2 mu = *(const f32x2*)(Kt + c), ad = *(const f32x2*)(Kt + C + c);
    const bool live = n * C + c < p.Rp;                    // (columns beyond Rp: the table slice was clamped, skip)
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      f32x2 sum = ((r0[j] + (f32x2){r0[j + 1][1], x0[j + 1]}) + r1[j]) + (f32x2){r1[j + 1][1], x1[j + 1]};
      if (EH) sum += (f32x2){qc[j], qc[j]};
      f32x2 t = sum * mu + ad;
      if (!live) t = (f32x2){-INFINITY, -INFINITY};
      const float mn = fmaxf(m_[j], fmaxf(t[0], t[1]));
      l_[j] = l_[j] * __builtin_amdgcn_exp2f(m_[j] - mn) + (__builtin_amdgcn_exp2f(t[0] - mn) + __builtin_amdgcn_exp2f(t[1] - mn));
      m_[j] = mn;
    }
  }
  // lanes -> one (m, l) per query
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    float m = m_[j];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o));
    float l = l_[j] * __builtin_amdgcn_exp2f(m_[j] - m);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) l += __shfl_xor(l, o);
    const int qy = qy0 + w, qx = qx0 + j;
    if (lane == 0 && qy < p.hs && qx < p.ws) *(float2*)(p.stats + ((size_t)b * p.R + qy * p.wc + qx) * 2) = make_float2(m, 1.f / l);
  }
}

// out[b, 2r + cls, c] = sum_s P~[r][s] * xT[cls][c][s]: A tile = NT * 16 channel rows x 32 keys, MFMA columns = class-grid pixels.
// <PT = 4, NT = 6>: 256 pixels x all 96 channels per workgroup (large batches).  <PT = 1, NT = 3>: 64 pixels x one half of
// the channels -- 8x the workgroups for calls whose grid would otherwise leave most of the 256 CUs idle (one 256x256
// image: 16 workgroups -> 128; the k order of every output is the same, so the results are bit-identical).
template <int PT, int NT, bool BF16>
__global__ __launch_bounds__(256) void att2_pv_kernel(const AttParams p) {
  constexpr int PIX = PT * 64, NP = NT * 16, NRH = 6 / NT;      // row halves per (pixel tile, class)
  constexpr int ES = BF16 ? 2 : 4;
  constexpr int XBYTES = PIX * 128, WBYTES = NP * 128;
  constexpr int NX = PT * 2, NW = (NT * 2 + 3) / 4;      // 8-row blocks of the two tiles staged per wave
  extern __shared__ __attribute__((aligned(16))) char smem[];
  char* Xb = smem;                        // P~ tiles [PIX][32 keys]
  char* Wb = smem + 2 * XBYTES;           // V^T tiles [NP ch][32 keys]

  const int tid = threadIdx.x, lane = tid & 63;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  // 1-D grid in (image, pixel tile, row half, class) order, class fastest, every XCD a contiguous range (xcd_tile): the
  // class workgroups of a pixel tile run side by side on one XCD and share the P~ tile through its L2
  const int lb = xcd_tile(blockIdx.x, gridDim.x);
  const int cls = lb & 3, py = cls >> 1, px = cls & 1;
  const int rh = (lb >> 2) % NRH, lt = (lb >> 2) / NRH;
  const int nt_ = (p.R + PIX - 1) / PIX;
  const int b = lt / nt_, t0 = (lt - b * nt_) * PIX;
  const int s_log = (lane & 7) ^ (4 * (w & 1) + (lane >> 4));
  const se_i32x4 rs_P = make_rsrc((const char*)p.Pt + (size_t)b * p.R * p.Rp * ES, (unsigned)p.R * p.Rp * (unsigned)ES);
  const se_i32x4 rs_V = make_rsrc((const char*)p.xT + (((size_t)b * 4 + cls) * 96 + rh * NP) * p.Rp * ES, (unsigned)NP * p.Rp * (unsigned)ES);
  unsigned xo[NX], wo[NW];
#pragma unroll
  for (int i = 0; i < NX; ++i) {
    const int rl = (i * 4 + w) * 8 + (lane >> 3);                // row of the tile staged by this lane
    const int r = t0 + rl;
    xo[i] = (rl < PIX && r < p.R) ? (unsigned)(r * p.Rp) * (unsigned)ES + s_log * 16u : 0x80000000u;
  }
#pragma unroll
  for (int j = 0; j < NW; ++j) {
    const int rl = (j * 4 + w) * 8 + (lane >> 3);
    wo[j] = rl < NP ? (unsigned)(rl * p.Rp) * (unsigned)ES + s_log * 16u : 0x80000000u;
  }
  int off0, off1;
  frag_offsets(lane, off0, off1);
  const unsigned lds_x = lds_addr_of(Xb), lds_w = lds_addr_of(Wb);
  const int nch = BF16 ? p.Rp >> 6 : p.Rp >> 5;

  auto stage = [&](int ch, int buf) {
    const unsigned delta = (unsigned)ch * 128u;
#pragma unroll
    for (int i = 0; i < NX; ++i)
      if ((i * 4 + w) * 8 < PIX) bufdma16(xo[i] + delta, rs_P, lds_x + buf * XBYTES + (i * 4 + w) * 1024);
#pragma unroll
    for (int j = 0; j < NW; ++j)
      if ((j * 4 + w) * 8 < NP) bufdma16(wo[j] + delta, rs_V, lds_w + buf * WBYTES + (j * 4 + w) * 1024);
  };

  f32x4 acc[NT][PT];
#pragma unroll
  for (int nt = 0; nt < NT; ++nt)
#pragma unroll
    for (int pt = 0; pt < PT; ++pt) acc[nt][pt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  stage(0, 0);
  dma_wait_all();
  __syncthreads();
  for (int ch = 0; ch < nch; ++ch) {
    const int buf = ch & 1;
    if (ch + 1 < nch) stage(ch + 1, buf ^ 1);
    if (BF16) mfma_chunk16<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    else mfma_chunk<NT, PT>(acc, Wb + buf * WBYTES, Xb + buf * XBYTES + w * PT * 2048, off0, off1);
    dma_wait_all();
    __syncthreads();
  }
  const int q = lane >> 4;
#pragma unroll
  for (int pt = 0; pt < PT; ++pt) {
    const int i = t0 + (w * PT + pt) * 16 + (lane & 15);
    if (i >= p.R) continue;
    const int yy = i / p.wc, xx = i - yy * p.wc;
    char* o = (char*)p.out + (((long)(b * p.h + 2 * yy + py) * p.w + 2 * xx + px) * 96 + rh * NP) * ES;
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
      const f32x4 a = acc[nt][pt];
      if (BF16) *(uint2*)(o + (nt * 16 + q * 4) * 2) = make_uint2(pack_bf16x2(a[0], a[1]), pack_bf16x2(a[2], a[3]));
      else *(f32x4*)(o + (nt * 16 + q * 4) * 4) = a;
    }
  }
}

// similar (B, L, hs, ws) <- P: channel = key j, pixel = query i (splitcam.py:106-108), for the unit-test entry point
template <bool BF16>
__global__ void att2_similar_kernel(const AttParams p) {
  const long idx = (long)blockIdx.x * 256 + threadIdx.x;      // over B*L*L, query fastest
  if (idx >= (long)p.B * p.L * p.L) return;
  const int i = idx % p.L;
  const long bj = idx / p.L;
  const int j = bj % p.L, b = bj / p.L;
  const int r = (i / p.ws) * p.wc + i % p.ws, s = (j / p.ws) * p.wc + j % p.ws;
  const size_t at = ((size_t)b * p.R + r) * p.Rp + s;
  p.similar[idx] = BF16 ? bf16_lo(((const unsigned short*)p.P)[at]) : p.P[at];
}

// Which form runs (measured on MI355X; same-box A/B runs of round 4, tools/att_ab.sh; statistics + P~ pass):
//   256x256 inputs (R = 1024), 32 images:  three-pass 65 + 71 us | fused, round-3 kernels 44 + 89 us | fused, LDS-staged P~ 44 + 81 us
//   512x512 inputs (R = 4096),  8 images:  three-pass 272 + 269 us | round-3 fused 183 + 410 us | LDS-staged 135 + 263 us
//       (the round-3 streaming kernel was bound by the L2 -> L1 traffic of its nine source rows per output row; the LDS-staged
//       kernel reads and writes HBM at 4.1 TB/s: 1.07 GB in 263 us)
//   bf16 (512x512, 16 images): three-pass 443 + 272 us | LDS-staged, E fp32 245 + 463 us | LDS-staged, E fp16 161 + 386 us
//       (P is bf16 there, so with fp32 E the fused form saves no bytes; with E in fp16 -- written by the E GEMM, converted by
//       the two LDS-staged kernels, the only readers -- the step goes 7.07 -> 6.87 ms)
// Default: fused wherever the LDS-staged kernels apply (wc % 4 == 0, wc <= 124), in fp32 also for R <= 1024 (round-3 kernels).
// SE_ATT_FUSED=0 / 1 forces the three-pass / fused form (in bf16 mode together with SE_ATT_FUSED_BF16=1); SE_ATT_FUSED_BF16=0
// keeps bf16 mode on the three-pass form; SE_ATT_PTILDE_LDS=0 keeps the round-3 streaming kernels inside the fused form;
// SE_ATT_STATS_LDS=0 the round-3 statistics kernel; SE_ATT_E16=0 fp32 E in bf16 mode.  (Entries of the option table,
// se_kernels.h: the tests switch forms inside one process through se_debug_set_option; -1 = not set.)
static bool att_lds_form(int wc) {
  return wc % 4 == 0 && wc <= 124 && opt(OPT_ATT_PTILDE_LDS) != 0;
}
static bool att_fused(bool bf16, int Rp, int wc) {
  const int e1 = opt(OPT_ATT_FUSED), e2 = opt(OPT_ATT_FUSED_BF16);
  const int f32mode = e1 < 0 ? -1 : (e1 != 0 ? 1 : 0);
  if (bf16) {
    if (e2 == 0) return false;
    if (e2 > 0) return f32mode < 0 ? att_lds_form(wc) : f32mode == 1;      // (explicitly enabled: SE_ATT_FUSED decides, as in fp32)
    return f32mode != 0 && att_lds_form(wc);
  }
  return f32mode < 0 ? (Rp <= 1024 || att_lds_form(wc)) : f32mode == 1;
}

template <bool BF16>
static hipError_t launch_attention_v2_t(const AttParams& p0, hipStream_t st) {
  AttParams p = p0;
  const bool fused = att_fused(BF16, p.Rp, p.wc) && !p.similar && p.stats;      // `similar_out` is P itself: three-pass form
  p.Pt = fused ? p.P : p.E;
  // SE_ATT_STATS_LDS=0: att2_stats_kernel instead of the LDS-staged statistics pass.  SE_ATT_E16=0: E stays fp32 in bf16 mode
  // (fp16 E needs both LDS-staged kernels: they are the only readers that convert).
  p.sym = (!BF16 && opt(OPT_ATT_SYM) != 0) ? 1 : 0;      // SE_ATT_SYM=0: every tile of E computed, from x and xn = x rn
  const bool stats_lds = opt(OPT_ATT_STATS_LDS) != 0;
  p.e16 = (BF16 && fused && att_lds_form(p.wc) && stats_lds && opt(OPT_ATT_E16) != 0) ? 1 : 0;
  {
    const long n = (long)p.B * p.h * p.w * (BF16 ? 12 : 24);
    // Preconditions of the fused streaming pass: its column shifts read up to wc + 8 floats in front of / behind
    // E and the key tables without a clamp, so the guard bands must be that wide, and att2_prep_kernel fills them from its
    // threads idx < guard -- the launch below must have at least `guard` threads.  (E's pad columns are written by the E
    // GEMM itself as finite values; an unguarded read is multiplied by kmul = 0, which only a finite value survives.)
    if (p.guard < p.wc + 8 || ((n + 255) / 256) * 256 < (long)p.guard) return hipErrorInvalidValue;
    ProfScope ps_(st, PL_ATT_PREP);
    hipLaunchKernelGGL(att2_prep_kernel<BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    hipLaunchKernelGGL(att2_transpose_kernel<BF16>, dim3(p.Rp / 32, 4, p.B), dim3(256), 0, st, p);
    if (p.e16) {      // offsets of the fp16-E form (att2_emean1_kernel)
      const long nr = (long)p.B * p.Rp;
      hipLaunchKernelGGL(att2_emean1_kernel<BF16>, dim3(p.hc, p.B), dim3(384), 0, st, p);
      hipLaunchKernelGGL(att2_emean2_kernel, dim3(p.B), dim3(384), 0, st, p);
      hipLaunchKernelGGL(att2_eoff_kernel<BF16>, dim3((unsigned)((nr + 3) / 4)), dim3(256), 0, st, p);
      hipLaunchKernelGGL(att2_eoff4_kernel, dim3((unsigned)((nr + 255) / 256)), dim3(256), 0, st, p);
    }
  }
  {
    // executed FLOPs = the tiles really launched x (PT*64 queries x NT*16 keys x K = 384): the symmetric enumeration runs
    // symT = 32 np^2 + 8 np of the 64 np^2 tiles (booking R*R regardless of p.sym printed a rate above the peak)
    const double alg_flops = 2.0 * p.B * (double)p.L * p.L * 1536.0;
    const double alg_bytes = (BF16 ? 2.0 : 4.0) * 2.0 * p.B * (double)p.h * p.w * 96;
    const long big_grid = (long)((p.R + 255) / 256) * ((p.R + 63) / 64) * p.B;
    if (big_grid >= 512) {
      constexpr int NT = 4, PT = 4;
      constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
      hipError_t e = ensure_max_lds((const void*)att2_pair_kernel<NT, PT, BF16>, LDS);
      if (e != hipSuccess) return e;
      // key tiles up to Rp, not R: the fused passes read the pad columns R .. Rp - 1 of every row (times kmul = 0) and need them
      // FINITE; in bf16 mode Rp is a multiple of 64 and a 32-key tile grid over R left columns unwritten (NaN patterns of stale
      // memory: tools/fuzz_attention.py, round 4).  Keys >= R are zero-filled by the buffer range check: E = 0 there.
      dim3 grid((p.R + PT * 64 - 1) / (PT * 64), (p.Rp + NT * 16 - 1) / (NT * 16), p.B);
      if (p.sym) {
        const int np = ((int)grid.y + 15) / 16;      // panels of 16 key tiles
        p.symT = 32 * np * np + 8 * np;
        grid = dim3((unsigned)(p.symT * p.B));
      }
      set_launch_cost(alg_flops, alg_bytes, nullptr, 2.0 * (double)grid.x * grid.y * grid.z * (PT * 64.0) * (NT * 16.0) * 384.0);
      set_launch_grid((long)grid.x * grid.y * grid.z);
      ProfScope ps_(st, PL_ATT_SCORE);
      hipLaunchKernelGGL((att2_pair_kernel<NT, PT, BF16>), grid, dim3(256), LDS, st, p);
    } else {          // one or a few small images: 128 queries x 32 keys per workgroup, 4x the workgroups (same k order: same bits)
      constexpr int NT = 2, PT = 2;
      constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
      hipError_t e = ensure_max_lds((const void*)att2_pair_kernel<NT, PT, BF16>, LDS);
      if (e != hipSuccess) return e;
      // key tiles up to Rp, not R: the fused passes read the pad columns R .. Rp - 1 of every row (times kmul = 0) and need them
      // FINITE; in bf16 mode Rp is a multiple of 64 and a 32-key tile grid over R left columns unwritten (NaN patterns of stale
      // memory: tools/fuzz_attention.py, round 4).  Keys >= R are zero-filled by the buffer range check: E = 0 there.
      dim3 grid((p.R + PT * 64 - 1) / (PT * 64), (p.Rp + NT * 16 - 1) / (NT * 16), p.B);
      if (p.sym) {
        const int np = ((int)grid.y + 15) / 16;      // panels of 16 key tiles
        p.symT = 32 * np * np + 8 * np;
        grid = dim3((unsigned)(p.symT * p.B));
      }
      set_launch_cost(alg_flops, alg_bytes, nullptr, 2.0 * (double)grid.x * grid.y * grid.z * (PT * 64.0) * (NT * 16.0) * 384.0);
      set_launch_grid((long)grid.x * grid.y * grid.z);
      ProfScope ps_(st, PL_ATT_SCORE);
      hipLaunchKernelGGL((att2_pair_kernel<NT, PT, BF16>), grid, dim3(256), LDS, st, p);
    }
  }
  if (fused) {
    const bool lds_form = att_lds_form(p.wc);
    // the two LDS-staged kernels: <E in fp16, DMA instructions per stage> by width (compile-time: the producers' counted waits)
#define SE_LAUNCH_STATS_LDS(EH, NIP)                                                                                   \
  do {                                                                                                                 \
    constexpr int lds = 3 * ((NIP) + 1) * 1024;                                                                        \
    hipError_t e = ensure_max_lds((const void*)att2_stats_lds_kernel<EH, NIP, 3>, lds);                                \
    if (e != hipSuccess) return e;                                                                                     \
    const int npy = (p.hs + 3) / 4, npx = (p.ws + 3) / 4;                                                              \
    hipLaunchKernelGGL((att2_stats_lds_kernel<EH, NIP, 3>), dim3((unsigned)(p.B * npy * npx)), dim3(384), lds, st, p, npy, npx); \
  } while (0)
#define SE_LAUNCH_PTILDE_LDS(EH, NIP, NS)                                                                              \
  do {                                                                                                                 \
    constexpr int lds = (NS) * ((NIP) + 1) * 1024 + 25 * 256 * 4;                                                      \
    hipError_t e = ensure_max_lds((const void*)att2_ptilde_lds_kernel<BF16, EH, NIP, NS>, lds);                        \
    if (e != hipSuccess) return e;                                                                                     \
    const int npy = (p.hc + 3) / 4, npx = p.wc / 4;                                                                    \
    hipLaunchKernelGGL((att2_ptilde_lds_kernel<BF16, EH, NIP, NS>), dim3((unsigned)(p.B * npy * npx)), dim3(640), lds, st, p, npy, npx); \
  } while (0)
    {
      const long rows = tile_order_count(p.B, p.hs, p.ws);
      ProfScope ps_(st, PL_ATT_SOFTMAX);
      const dim3 grid((unsigned)((rows + 3) / 4));
      if (lds_form && stats_lds && (p.e16 || p.Rp > 1024)) {      // (R <= 1024: the register kernel is as fast, 44 vs 46 us)
        if (p.e16) { if (p.wc <= 64) SE_LAUNCH_STATS_LDS(true, 10); else SE_LAUNCH_STATS_LDS(true, 14); }
        else { if (p.wc <= 64) SE_LAUNCH_STATS_LDS(false, 20); else SE_LAUNCH_STATS_LDS(false, 26); }
      } else if (p.Rp <= 64 * 16) hipLaunchKernelGGL((att2_stats_kernel<16, BF16>), grid, dim3(256), 0, st, p);
      else if (p.Rp <= 64 * 64) hipLaunchKernelGGL((att2_stats_kernel<64, BF16>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL((att2_stats_kernel<0, BF16>), grid, dim3(256), 0, st, p);
    }
    {
      const long rows = tile_order_count(p.B, p.hc, p.wc);
      ProfScope ps_(st, PL_ATT_BOXSUM);
      const dim3 grid((unsigned)((rows + 3) / 4));
      if (lds_form) {
        bool done = false;
        if constexpr (BF16) {
          if (p.e16) {
            if (p.wc <= 64) SE_LAUNCH_PTILDE_LDS(true, 16, 3); else SE_LAUNCH_PTILDE_LDS(true, 18, 3);      // 3 stages: 77 / 83 KB, two workgroups per CU
            done = true;
          }
        }
        if (!done) { if (p.wc <= 64) SE_LAUNCH_PTILDE_LDS(false, 28, 4); else SE_LAUNCH_PTILDE_LDS(false, 36, 3); }
      } else if (p.wc % 4 == 0) hipLaunchKernelGGL((att2_ptilde4_kernel<BF16>), dim3((unsigned)((rows + 7) / 8)), dim3(512), 0, st, p);
      else hipLaunchKernelGGL((att2_ptilde1_kernel<BF16>), grid, dim3(256), 0, st, p);
    }
#undef SE_LAUNCH_STATS_LDS
#undef SE_LAUNCH_PTILDE_LDS
  } else {
    {
      const long rows = tile_order_count(p.B, p.hs, p.ws);
      ProfScope ps_(st, PL_ATT_SOFTMAX);
      const dim3 grid((unsigned)((rows + 3) / 4));
      if (p.Rp <= 64 * 16) hipLaunchKernelGGL((att2_softmax_reg_kernel<16, BF16>), grid, dim3(256), 0, st, p);
      else if (p.Rp <= 64 * 64) hipLaunchKernelGGL((att2_softmax_reg_kernel<64, BF16>), grid, dim3(256), 0, st, p);
      else hipLaunchKernelGGL(att2_softmax_kernel<BF16>, grid, dim3(256), 0, st, p);
    }
    if (p.similar) {
      const long n = (long)p.B * p.L * p.L;
      ProfScope ps_(st, PL_LAYOUT);
      hipLaunchKernelGGL(att2_similar_kernel<BF16>, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, st, p);
    }
    {
      const long rows = tile_order_count(p.B, p.hc, p.wc);
      ProfScope ps_(st, PL_ATT_BOXSUM);
      if (p.wc % (BF16 ? 8 : 4) == 0) hipLaunchKernelGGL(att2_boxsum4_kernel<BF16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
      else hipLaunchKernelGGL(att2_boxsum_kernel<BF16>, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p);
    }
  }
  {
    // small calls (one or a few images): 64-pixel tiles x half the channels, 8x the workgroups (att2_pv_kernel)
    const long big_grid = (long)((p.R + 255) / 256) * 4 * p.B;
    set_launch_cost(2.0 * p.B * (double)p.L * p.L * 1536.0, (BF16 ? 2.0 : 4.0) * 2.0 * p.B * (double)p.h * p.w * 96, nullptr,
                    2.0 * p.B * 4.0 * (double)p.R * p.Rp * 96.0);
    // pixel tile of the large-batch form, chosen on the GPU (SE_ATT_PV_PT overrides): 256 pixels (PT = 4) need 88 KB of LDS --
    // ONE workgroup per CU; 128 (fp32) / 192 (bf16) fit two, whose barrier and DMA waits overlap: fp32 784 -> 747 us at
    // 512x512 B=8, bf16 307 -> 247 us at 512x512 B=16 (same box, round 3)
    const int pvpt = opt(OPT_ATT_PV_PT) > 0 ? opt(OPT_ATT_PV_PT) : (BF16 ? 3 : 2);
    if (big_grid >= 512 && pvpt == 4) {
      constexpr int PT = 4, NT = 6;
      constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
      hipError_t e = ensure_max_lds((const void*)att2_pv_kernel<PT, NT, BF16>, LDS);
      if (e != hipSuccess) return e;
      dim3 grid((unsigned)big_grid);
      set_launch_grid((long)grid.x);
      ProfScope ps_(st, PL_ATT_PV);
      hipLaunchKernelGGL((att2_pv_kernel<PT, NT, BF16>), grid, dim3(256), LDS, st, p);
    } else if (big_grid >= 512 && pvpt == 3) {
      constexpr int PT = 3, NT = 6;
      constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
      hipError_t e = ensure_max_lds((const void*)att2_pv_kernel<PT, NT, BF16>, LDS);
      if (e != hipSuccess) return e;
      dim3 grid((unsigned)(((p.R + 191) / 192) * 4 * p.B));
      set_launch_grid((long)grid.x);
      ProfScope ps_(st, PL_ATT_PV);
      hipLaunchKernelGGL((att2_pv_kernel<PT, NT, BF16>), grid, dim3(256), LDS, st, p);
    } else if (big_grid >= 512 && pvpt == 2) {
      constexpr int PT = 2, NT = 6;
      constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
      hipError_t e = ensure_max_lds((const void*)att2_pv_kernel<PT, NT, BF16>, LDS);
      if (e != hipSuccess) return e;
      dim3 grid((unsigned)(((p.R + 127) / 128) * 4 * p.B));
      set_launch_grid((long)grid.x);
      ProfScope ps_(st, PL_ATT_PV);
      hipLaunchKernelGGL((att2_pv_kernel<PT, NT, BF16>), grid, dim3(256), LDS, st, p);
    } else {
      constexpr int PT = 1, NT = 3;
      constexpr int LDS = 2 * PT * 64 * 128 + 2 * NT * 16 * 128;
      hipError_t e = ensure_max_lds((const void*)att2_pv_kernel<PT, NT, BF16>, LDS);
      if (e != hipSuccess) return e;
      dim3 grid((unsigned)(((p.R + 63) / 64) * 2 * 4 * p.B));
      set_launch_grid((long)grid.x);
      ProfScope ps_(st, PL_ATT_PV);
      hipLaunchKernelGGL((att2_pv_kernel<PT, NT, BF16>), grid, dim3(256), LDS, st, p);
    }
  }
  return hipGetLastError();
}
static hipError_t launch_attention_v2(const AttParams& p, hipStream_t st) {
  return p.bf16 ? launch_attention_v2_t<true>(p, st) : launch_attention_v2_t<false>(p, st);
}

// SE_ATT_V1=1 selects the patch form (materialised L x L scores, K = 1536 / 4L) for A/B measurements
hipError_t launch_attention(const AttParams& p, hipStream_t st) {
  return p.E ? launch_attention_v2(p, st) : launch_attention_v1(p, st);
}
bool attention_v2_enabled() {
  return opt(OPT_ATT_V1) == 0;
}

}  // namespace se
