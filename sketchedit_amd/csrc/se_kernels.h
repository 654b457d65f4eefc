// Internal (C++) interface between the host-side forward plan (se_api.hip) and the
// gfx950 kernels (se_gconv.hip, se_wino.hip, se_wino48.hip, se_attention.hip, se_misc.hip).  Not part of the C-ABI.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include <vector>

namespace se {

// ---------------------------------------------------------------------------------------------
// Optional per-launch timing (HIP events on the launch stream), used by bench.py for the roofline
// figures.  Off by default; when off the launch wrappers add nothing.
// ---------------------------------------------------------------------------------------------
enum ProfLabel { PL_GCONV_N192 = 0, PL_GCONV_N96, PL_GCONV_N48, PL_GCONV_N24, PL_WINO_N192, PL_WINO_N96, PL_WINO_UP96, PL_SMALL_CONV, PL_PACK, PL_COLREDUCE,
                 PL_ATT_PREP, PL_ATT_SCORE, PL_ATT_SOFTMAX, PL_ATT_BOXSUM, PL_ATT_PV, PL_LAYOUT, PL_COUNT };
const char* prof_label_name(int l);
struct Profiler {
  struct Rec { int label; const char* name; double flops; double exec_flops; double bytes; long blocks; hipEvent_t a, b; };
  std::vector<Rec> recs;
  std::vector<hipEvent_t> pool;   // pre-created events (se_profile_enable), handed out two per launch
  size_t used = 0;
  bool on = false;
};
void set_profiler(Profiler* p);          // thread-local; set by the API under the ctx lock
// cost/name of the NEXT launch (consumed once).  flops = algorithmic FLOPs of the layer as the reference defines it;
// exec_flops = multiply-add FLOPs the kernel's MFMA pipe actually executes (Winograd / sub-pixel / space-to-depth forms
// execute fewer); < 0: same as flops.  The roofline fraction is computed from exec_flops.
void set_launch_cost(double flops, double bytes, const char* name = nullptr, double exec_flops = -1.0);
// workgroups of the NEXT launch (consumed once): lets the report say how much of the 256-CU machine a layer can occupy
void set_launch_grid(long blocks);
// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (device, kernel)
hipError_t ensure_max_lds(const void* func, int bytes);

// ---------------------------------------------------------------------------------------------
// Developer switches.  ONE process-wide table: every entry is read from the environment exactly once (first use of the
// table) and can afterwards be changed only through the C-ABI test aid se_debug_set_option -- the launch path never calls
// getenv (ADVICE r4 / VERDICT r4 item 7c: getenv per launch is host latency at batch 1 and is not safe against a concurrent
// setenv from another thread).  opt() is a relaxed atomic load.
// ---------------------------------------------------------------------------------------------
#define SE_OPTIONS(X)                                                                                         \
  X(XCD_REMAP, 1)        /* 0: XCD-aware tile order off */                                                     \
  X(RTILE, 1)            /* 0: gather-GEMM instead of the raw-tile kernels */                                  \
  X(RTILE_LL_MIN, 256)   /* low-latency mode: fewest raw tiles for which the raw-tile kernels still run */      \
  X(RTILE_WX, 2)         /* 24 -> 24 layers: 0 direct raw tile, 1 F(2,3) along x, 2 two-dimensional F(2x2,3x3) */ \
  X(RTILE_DENSE, 1)      /* 0: channel-padded K for the 5x5 first layers */                                    \
  X(RTILE_D5W, 1)        /* 0: 5x5 first layers on the direct dense-K kernel */                                \
  X(TEST_OFFSET_LIMIT, 0) /* test aid, se_debug_set_option only: byte range of the 32-bit-offset kernels (0: 2^31) */ \
  X(RCONV16, 1) X(RCONV16_DUAL, 1) X(RCONV16_TILE, 8) X(RCONV96, 1) X(VECBIAS, 1)                              \
  X(WINOGRAD, 1) X(WINOGRAD48, 1) X(WINOGRAD_UP, 1) X(WINOGRAD_UP48, 1)                                         \
  X(WINOGRAD_F43, 1)     /* hybrid F(2,3)xF(4,3): 0 off, 1 everywhere, 2 netG only */ \
  X(WINO48_TILES, 64) X(WINOUP_TILES, 64)                                                                       \
  X(GCONV_FAST, 1) X(GCONV_VARIANT_N192, 0) X(GCONV_VARIANT_N96, 0) X(GCONV_VARIANT_N48, 0) X(GCONV_VARIANT_N24, 0) \
  X(LL_STAGES, 2) X(FORK_DEFAULT, 1) X(LL_WINO_MIN_WG, 64) X(LL_WINO48_MIN_WG, 128)                                                                                               \
  X(ATT_V1, 0) X(ATT_FUSED, -1) X(ATT_FUSED_BF16, -1) X(ATT_PTILDE_LDS, 1) X(ATT_STATS_LDS, 1) X(ATT_E16, 1)      \
  X(ATT_SYM, 1) X(ATT_PV_PT, -1)
enum Opt {
#define X(name, dflt) OPT_##name,
  SE_OPTIONS(X)
#undef X
  OPT_COUNT
};
int opt(int o);                                   // current value
int opt_set(const char* name, int value);         // "SE_<NAME>" or "<NAME>"; returns 0, or 1 for an unknown name
int opt_get(const char* name, int* value);        // the same for reading
void opt_reset();                                 // every entry back to its environment / built-in default
long long opt_epoch();                            // number of opt_set / opt_reset calls so far (key of captured forwards)

// ---------------------------------------------------------------------------------------------
// Gather-GEMM gated convolution (the hot kernel).
//   D[n][p] = sum_k Wp[n][k] * X[p][k]   n: packed output channels, p: output pixels,
//   k: flattened (tap, channel) in 32-float chunks.  X rows are gathered on the fly from
//   one or two NHWC sources (im2col-free); Wp is packed on the host (se_api.hip pack_layer).
// ---------------------------------------------------------------------------------------------
struct GConvParams {
  const float* src0;   // NHWC [B][Hin][Win][C0]  (fp32, or bf16 when `bf16` is set -- the pointer types stay float*)
  const float* src1;   // optional second source (channel concat): NHWC [B][Hin][Win][C1] or vector [B][C1]
  const float* wpk;    // packed weights [nch][NP][128 bytes] (LDS image, pre-swizzled): 32 fp32 or 64 bf16 k-values per row
  const float* bias;   // [NP] in packed-row order (always fp32)
  float* dst;          // NHWC [B][Ho][Wo][G]
  const float* zeros;  // >= 16 bytes of zeros (source of out-of-bounds granules)
  int B, Hin, Win, Ho, Wo;
  int C0, C1;          // channel counts (row strides, floats)
  int C0g, CG;         // granules (4 floats) per tap in src0 / in total
  int T, KW;           // taps, taps per kernel row
  int magicCG, magicKW; // x/CG == (x*magicCG)>>16, t/KW == (t*magicKW)>>8 on the ranges used (host-verified)
  int magicKH;          // kernel rows T / KW
  unsigned rep;         // sum over kernel rows j of 1 << (j*KW): replicates a column-validity mask to every row
  unsigned div_hw_m, div_w_m;   // exact x / (Ho*Wo) and x / Wo for x < 2^32 (se_device.h udiv_magic), with
  int div_hw_l, div_w_l;        // their shift counts
  int stride, dil, pad;
  int ushift;          // 1: source is read through a nearest x2 upsample (coords >> 1)
  int up2;             // 1: sub-pixel form of nearest-x2 + 3x3: one workgroup = one output parity class (py,px), a 2x2 conv
                       //    on the source grid with pre-summed weights; (Ho,Wo) is then the SOURCE grid, (OH,OW) the output
  int OH, OW;
  int Hlim, Wlim;      // validity limits of the pre-shift tap coordinates
  int src1_vec;        // src1 is a per-batch vector (spatially constant, still zero padded)
  int nch;             // number of 32-k chunks
  int G;               // gated (stored) output channels
  int act;             // 0 ELU, 1 ReLU
  int total_pix;       // B*Ho*Wo
  int xcd;             // 1: XCD-aware tile order (se_device.h xcd_tile)
  int np_full;         // packed rows of the whole layer (row count of one chunk of wpk)
  int nf_full;         // feature tiles of the whole layer (non-MIXED layouts: gates start at tile nf_full)
  int bf16;            // 1: sources, weights and dst are bf16 (granule = 8 channels, chunk = 64 k); C0, C1, G count elements
  int small_grid;      // 1: low-latency launch shape (small pixel tiles; the packed rows split over blockIdx.y) for grids
                       //    that would otherwise leave most of the 256 CUs idle (batch 1)
};

// magic numbers of se_device.h udiv_magic for divisor d >= 1
static inline void udiv_magic_host(unsigned d, unsigned* m, int* l) {
  int ll = 0;
  while ((1ull << ll) < d) ++ll;
  *l = ll;
  *m = (unsigned)(((1ull << 32) * ((1ull << ll) - d)) / d + 1);
}

enum GConvCfg { GC_N192 = 0, GC_N96 = 1, GC_N48 = 2, GC_N24 = 3 };
// rows (packed output channels) of each config
static inline int gconv_np(int cfg) { return cfg == GC_N192 ? 192 : cfg == GC_N96 ? 96 : cfg == GC_N48 ? 48 : 32; }
static inline bool gconv_mixed(int cfg) { return cfg >= GC_N48; }

hipError_t launch_gconv(int cfg, const GConvParams& p, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Winograd F(2x2,3x3) form of the 3x3 stride-1 gated conv 96 -> 192 (se_wino.hip)
// ---------------------------------------------------------------------------------------------
struct WinoParams {
  const float* src;    // NHWC [B][h][w][96]
  const float* src1;   // optional second 96-channel source (channel concat): NHWC tensor, or [B][96] vector; else null
  int src1_vec;        // src1 is a per-image vector (spatially constant, zero padded like a tensor)
  const float* upk;    // [16 positions][3 or 6 chunks][192 packed rows][32]: G g G^T, LDS image (pre-swizzled)
  const float* bias;   // [192] packed-row order (features, then gates)
  float* dst;          // NHWC [B][h][w][96]
  int B, h, w, d;      // dilation d; h % 2d == 0 and w % 2d == 0
  int th, tw;          // tile grid h/2 x w/2
  int total_tiles;     // B*th*tw
  int act;             // 0 ELU, 1 ReLU
  int xcd;             // 1: XCD-aware tile order
  unsigned div_tpi_m, div_tw_m, div_d_m;   // udiv_magic numbers for th*tw, tw and d
  int div_tpi_l, div_tw_l, div_d_l;
  const float* vbias;  // optional [B][9][192]: per-image, per-border-configuration bias (the folded vector source, launch_vecbias)
};
hipError_t launch_wino(const WinoParams& p, hipStream_t st);
// A spatially constant second source (the pooled style vector in front of conv11, editline_g.py:166-167) contributes
//   T[b][cfg][n] = sum over the taps inside the image in border configuration cfg, sum_c W[n][C0 + c][tap] * v[b][c]
// to output row n -- a per-image bias that depends only on WHICH taps fall inside the image (zero padding: 3 x 3 configurations
// for a 3x3 kernel at dilation 1): cfg = 3 * (y == 0 ? 0 : y == h-1 ? 2 : 1) + (x == 0 ? 0 : x == w-1 ? 2 : 1).
// wv: [9 taps][C1][192 packed rows] (row fastest), vec: [B][C1], T: [B][9][192].
// round_vec: the vector is rounded to bf16 on the way in (bf16 mode: wv then holds bf16-rounded weights; fp32 accumulate)
hipError_t launch_vecbias(const float* wv, const float* vec, float* T, int B, int C1, hipStream_t st, int round_vec = 0);
// Hybrid F(2,3) x F(4,3) form of the single-source 96 -> 192 layer (se_wino24.hip): 2x4 output tiles, th = h/2, tw = w/4,
// h % 2d == 0 and w % 4d == 0; upk [72 iterations][192 MIXED rows][32] (pack_wino24), bias [192] MIXED order
hipError_t launch_wino24(const WinoParams& p, hipStream_t st);
// the same for a 48-channel source (48 -> 192, xconv5): src NHWC 48, upk [48 iterations][192 MIXED rows][32] (pack_wino24)
hipError_t launch_wino24_c48(const WinoParams& p, hipStream_t st);
// 48 -> 96 form (se_wino48.hip): src / dst NHWC 48 channels, upk [24 iterations][96 MIXED rows][32], bias [96] MIXED order
hipError_t launch_wino48(const WinoParams& p, hipStream_t st);
// 24 -> 96 in the same kernel (round 5; xconv3 / pmconv3): src NHWC 24 channels, upk [16 positions][96 MIXED rows][32 k:
// channels 0-23, then zeros] (pack_wino48_c24), one iteration per position, k-half 1 issues two k-steps
hipError_t launch_wino48_c24(const WinoParams& p, hipStream_t st);
// gen_deconv 96 -> 96 (48 gated), F(2x2,2x2) on the 4 sub-pixel classes (se_wino_up.hip): src NHWC 96 at (h, w), dst NHWC
// 48 at (2h, 2w), upk [4 classes][27 iterations][96 MIXED rows][32], bias [96] MIXED order; h, w even; th = h/2, tw = w/2
hipError_t launch_winoup(const WinoParams& p, hipStream_t st);
// gen_deconv 48 -> 48 (24 gated) in the same form (se_wino_up48.hip): src NHWC 48 at (h, w), dst NHWC 24 at (2h, 2w), upk
// [4 classes][14 iterations][48 MIXED rows][32] in the K pairing of the 48-channel kernels, bias [48] MIXED order
hipError_t launch_winoup48(const WinoParams& p, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// bf16 raw-tile form of the 3x3 stride-1 gated conv 96 -> 192 (se_rconv16.hip)
// ---------------------------------------------------------------------------------------------
struct RConvParams {
  const void* src;     // bf16 NHWC [B][h][w][96]
  const void* wpk;     // bf16 weights: image of pack_layer16 [14 chunks][192 rows][64 k] (16 x 16 tiles) or of
                       // pack_rconv16 [27 steps][12 row tiles][16 rows][32 k] (8 x 16 tiles)
  const float* bias;   // [192] packed-row order (features, then gates)
  void* dst;           // bf16 NHWC [B][h][w][96]
  int B, h, w, d;      // dilation d; h % d == 0 and w % d == 0
  int hs, ws;          // polyphase sub-image size h/d x w/d
  int ty, tx;          // tiles per sub-image (16 x 16, or 8 rows x 16 columns)
  int act;             // 0 ELU, 1 ReLU
  int xcd;             // 1: XCD-aware tile order
  const float* vbias;  // optional [B][9][192] fp32: folded vector source (launch_vecbias; d == 1; 8 x 16 tiles only)
  int dual;            // 8 x 16 tiles only: ws == 8, d even -- two phases (py, px), (py, px + 1) side by side in one tile
};
hipError_t launch_rconv16(const RConvParams& p, hipStream_t st);
bool rconv16_small_tiles();   // 8 x 16 tiles, two workgroups per CU (default) / SE_RCONV16_TILE=16

// ---------------------------------------------------------------------------------------------
// Raw-tile form of the 96-row bf16 gated convs, stride 1 (se_rconv96.hip): 3x3 48 -> 96 / 24 -> 96, gen_deconv 96 -> 96
struct RConv96Params {
  const void* src;     // bf16 NHWC [B][h][w][8 CG]
  const void* wpk;     // bf16 image of pack_rconv96: [class][step][6 row tiles][16 rows][32 k]
  const float* bias;   // [96] packed-row order (48 features, then 48 gates)
  void* dst;           // bf16 NHWC [B][h][w][48], or [B][2h][2w][48] for gen_deconv
  int B, h, w;         // source size
  int CG;              // 8-channel granules per source pixel (3, 6, 12)
  int ty, tx;          // 16 x 16 tiles of the source grid (stride 2: of the OUTPUT grid)
  int stride, oh, ow;  // 1, or 2 with the output size (3x3, 24 -> 96: de-interleaved raw tile, se_rconv96.hip)
  int up2;             // gen_deconv: four 2x2 sub-pixel classes
  int act;             // 0 ELU, 1 ReLU
  int xcd;             // 1: XCD-aware tile order
};
hipError_t launch_rconv96(const RConv96Params& p, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Raw-tile form of the narrow (MIXED-row) stride-1 gated convs, fp32 and bf16 (se_rtile.hip)
// ---------------------------------------------------------------------------------------------
struct RTileParams {
  const float* src;    // NHWC [B][Hin][Win][C] (fp32, or bf16 when `bf16`)
  const float* wpk;    // packed weight image of pack_layer / pack_layer16 ([class][chunk][NP rows][128 bytes])
  const float* bias;   // [NP] packed-row order
  float* dst;          // NHWC [B][OH][OW][G]
  int B, Hin, Win;     // source grid (= output grid; for up2 the output grid is 2x)
  int C;               // elements per source pixel (whole granules)
  int CG;              // granules per tap
  int T, KW;           // taps, taps per kernel row (up2: 4, 2)
  int magicCG, magicKW;
  int pad;             // top / left zero padding (k / 2); up2: 1 - class parity
  int up2, OH, OW;
  int G;               // stored output channels (row stride of dst)
  int nch, NP;         // 32-k (fp32) / 64-k (bf16) chunks, packed rows (48 or 32)
  int RH, RW;          // raw tile: 8 + KH - 1 rows, 16 + KW - 1 columns
  int raw_bytes;       // RH * RW * C * element size, rounded up to 1 KB
  unsigned div_cg_m, div_rw_m;   // udiv_magic numbers for (granules per pixel) and RW
  int div_cg_l, div_rw_l;
  int ty, tx;          // 8 x 16 tiles per image
  int act, bf16, xcd;
  int dense;           // > 0: dense-K form of the 5x5 first layers (fp32): `dense` = real input channels (3 or 5) out of the C
                       // stored per pixel; wpk is then the image of pack_layer_dense (k = tap * dense + channel, no channel padding);
                       // + 100: its F(2,5)-along-x form; 204: the bf16 pair-of-taps form (rtile_kernel<3, 8, true, true>:
                       // 8-byte pixels = the first four stored channels, wpk = image of pack_layer16_d4, RW = 21)
};
hipError_t launch_rtile(const RTileParams& p, hipStream_t st);
// 24 -> 24 3x3 stride 1 with F(2,3) along x (se_rtilew.hip): src NHWC 24, wpk = image of pack_rtilew ([4 positions][3 chunks]
// [24 physical rows][32 k]), bias [32] in the MIXED packed-row order, dst NHWC 12; Win even; ty = ceil(H / 16), tx = ceil(W / 16): blocks of 16 x 16 outputs, walked by persistent workgroups
hipError_t launch_rtilew(const RTileParams& p, hipStream_t st);
// the same layers with the two-dimensional F(2x2,3x3) transform (Hin and Win even): wpk = image of pack_rtilew2
// ([16 positions][24 physical rows][32 floats]: channels 0-15 in slots 0-3, channels 16+2q, 17+2q in slot 4+q)
hipError_t launch_rtilew2(const RTileParams& p, hipStream_t st);
int rtile_rows(bool bf16);    // output rows per workgroup tile (8 fp32, 32 bf16)

// ---------------------------------------------------------------------------------------------
// 3x3 conv 12 -> {1,3} raw output + fused tanh/sigmoid/composite (final layer of each decoder)
// ---------------------------------------------------------------------------------------------
struct SmallConvParams {
  const float* x;      // NHWC [B][H][W][12]
  const float* w;      // [COUT][9][12]
  const float* b;      // [COUT]
  int B, H, W, cout;
  int mode;            // 0: sigmoid -> mask (+hard);  1: tanh -> out;  2: tanh -> coarse + xnow;  3: tanh -> fine + composed;  4: raw
  float* out_nchw;     // mask / maskim / coarse / fine, NCHW (may be null for modes 2,3)
  float* hard;         // mode 0: (B,1,H,W) thresholded mask (may be null)
  const float* img;    // NCHW (B,3,H,W)   modes 2,3
  const float* mask;   // (B,1,H,W)        mode 2: hard mask; mode 3: soft mask
  float* xnow;         // mode 2: NHWC4 next-stage input
  float* composed;     // mode 3: NCHW (B,3,H,W) (may be null)
  int no_mask_coarse;
  // batch strides in floats (0 = dense): the packed output of SE_FLAG_PACKED_OUT is one (B,4,H,W) buffer, composed in
  // planes 0-2 and the soft mask in plane 3 (one all-gather of the packed outputs)
  long out_bs;         // of out_nchw (mode 0: the soft mask)
  long mask_bs;        // of mask (mode 3: the soft mask)
  long comp_bs;        // of composed
  int bf16;            // x is bf16 NHWC with a 16-channel pixel stride; xnow (mode 2) is written as bf16 NHWC8
  // mode 3: the output quantisation of test.py:25-27 fused into this last kernel (either may be null)
  unsigned char* rgb8; // (B,H,W,3) uint8 = trunc((composed + 1) / 2 * 255)
  unsigned char* m8;   // (B,H,W)   uint8 = trunc(soft mask * 255)
};
hipError_t launch_small_conv(const SmallConvParams& p, hipStream_t st);

// ---------------------------------------------------------------------------------------------
// Elementwise / layout / reductions
// ---------------------------------------------------------------------------------------------
// netM input: [image(3), sketch(1)] NCHW -> NHWC4
hipError_t launch_pack_m(const float* image, const float* sketch, float* dst4, int B, int H, int W, hipStream_t st);
// netG inputs: coarse NHWC8 = [x*(1-m) (3), guide, m, 0,0,0]; style NHWC8 = [x2*m2 (3) (or x2), g2, m2, 0,0,0],
// or with joint (guide * 0): style NHWC4 = [x2*m2 (3) (or x2), m2]
hipError_t launch_pack_g(const float* x, const float* x2, const float* mask, const float* mask2, const float* guide,
                         float* coarse8, float* style8, int B, int H, int W, int no_mask_cc, int joint,
                         hipStream_t st);
hipError_t launch_nchw_to_nhwc(const float* src, float* dst, int B, int C, int Cpad, int H, int W, hipStream_t st);
hipError_t launch_dequantize_u8(const unsigned char* rgb, const unsigned char* sk8, const float* lut, float* image, float* sketch,
                                int B, int H, int W, hipStream_t st);
hipError_t launch_nhwc_to_nchw(const float* src, float* dst, int B, int C, int Cstride, int H, int W, hipStream_t st);
// test.py:25-27 output quantisation: composed NCHW (B,3,H,W) -> rgb HWC uint8, mask (B,1,H,W) -> uint8; W % 4 == 0
hipError_t launch_quantize_u8(const float* composed, const float* mask, unsigned char* rgb, unsigned char* m8, int B, int H,
                              int W, hipStream_t st);
// column reduce over pixels: x [B][HW][C] -> out [B][C].  op 0 max, 1 mean, 2 rsqrt(sum(x^2)+1e-8)
// x_bf16: x holds bf16; out_bf16 (optional): a bf16 copy of the result (the style vector as a bf16 conv source)
hipError_t launch_colreduce(const float* x, float* partial, float* out, int B, int HW, int C, int op, hipStream_t st,
                            int x_bf16 = 0, float* out_bf16 = nullptr);
// bf16 forms of the input packing (NHWC8 bf16 for every network input) and of the unit-test layout converters
hipError_t launch_pack_m16(const float* image, const float* sketch, float* dst8, int B, int H, int W, hipStream_t st);
hipError_t launch_pack_g16(const float* x, const float* x2, const float* mask, const float* mask2, const float* guide,
                           float* coarse8, float* style8, int B, int H, int W, int no_mask_cc, int joint, hipStream_t st);
hipError_t launch_nchw_to_nhwc16(const float* src, float* dst, int B, int C, int Cpad, int H, int W, hipStream_t st);
hipError_t launch_nhwc16_to_nchw(const float* src, float* dst, int B, int C, int Cstride, int H, int W, hipStream_t st);
static const int COLREDUCE_SPLITS = 128;

// ---------------------------------------------------------------------------------------------
// Contextual attention (patch 4, stride 2)
// ---------------------------------------------------------------------------------------------
struct AttParams {
  const float* x;      // NHWC [B][h][w][96]   raw features (queries, values)
  const float* rn;     // [B][96]  1/sqrt(sum x^2 + 1e-8)
  float* xn;           // NHWC [B][h][w][96]   workspace: x * rn (keys)
  const float* hard;   // (B,1,4h,4w) full-resolution hole mask
  float* valid;        // [B][Lp]  workspace: key validity {0,1}
  float* S;            // [B][L][Lp] workspace: scores, query-major
  float* out;          // NHWC [B][h][w][96]
  int B, h, w, hs, ws, L, Lp;
  float scale;         // softmax scale (10)
  float th;            // validity threshold (0.1)
  // ---- space-to-depth form (se_attention.hip, "v2"): class grid hc x wc = h/2 x w/2, R = hc*wc rows, Rp = R rounded up to 32
  int hc, wc, R, Rp;
  float* xT;           // [B][4 classes][96][Rp]  workspace: x transposed per parity class (A operand of the P~.V GEMM)
  float* E;            // [B][R][Rp] workspace: pixel-pair dot products E (three-pass form: later overwritten by P~)
  float* P;            // [B][R][Rp] workspace (fp32, or bf16 in bf16 mode): softmax probabilities in class-grid indexing (row = query,
                       // column = key); fused form: P~ is written here directly and P never exists
  float* Pt;           // where att2_pv_kernel reads P~: E (three-pass form) or P (fused form)
  float* stats;        // [B][R][2] fused form: row maximum and 1 / row sum of every query, in class-grid indexing
  float* kmul;         // [B][Rp] fused form: scale * log2(e) for a valid key, 0 for an invalid key or a non-key position
  float* kadd;         // [B][Rp] fused form: 0 for a key, -inf for a position that is not a key
  int guard;           // floats of guard band allocated in front of and behind E and in front of validR (>= wc + 8)
  float* validR;       // [B][Rp]    workspace: key validity in class-grid indexing: 1 / 0, -1 where the position is not a key
  float* similar;      // optional (B, L, hs, ws) NCHW copy of P for the unit-test entry point
  int bf16;            // x, xn, xT, P~ (in the E buffer) and out hold bf16; Rp is then a multiple of 64
  int e16;             // set by the launcher: E holds fp16 (bf16 mode, LDS-staged fused passes), as E[r][s] - ea[r] - eb[s]
  float* emean;        // [B][2][384] fp16-E form: mean over the class-grid positions of the query blocks X_r; rn times that (the key blocks' mean)
  float* epart;        // [B][hc][384] scratch of that mean's two-stage sum
  float* ea;           // [B][Rp] fp16-E form: row offset <X_r, mean XN>            (0 in the pad)
  float* eb;           // [B][Rp] fp16-E form: column offset <mean X, XN_s> - <mean X, mean XN>
  float* ea4;          // [B][Rp] fp16-E form: ea[q] + ea[q+1] + ea[q+wc] + ea[q+wc+1], the offset of S[q][.] owed to the query
  float* kadd2;        // [B][Rp] fp16-E form: kadd + (eb[k] + eb[k+1] + eb[k+wc] + eb[k+wc+1]) * kmul: the key half, folded into the key test
  int sym;             // set by the launcher: att2_pair_kernel computes the tiles on / right of the diagonal and mirrors them (fp32)
  int symT;            //   ... computed tiles per image (1-D grid)
};
hipError_t launch_attention(const AttParams& p, hipStream_t st);    // p.E != null: space-to-depth form, else the patch form
bool attention_v2_enabled();

}  // namespace se
