// Host side of libsketchedit_hip.so: context, weight repacking, workspace arena and the forward
// plans of netM / netG / inference behind the C-ABI declared in include/sketchedit_hip.h.
//
// Layer tables restate the reference constructors
//   DeepFillC2Generator.__init__  /root/reference/models/networks/editline_g.py:44-100
//   MDGenerator.__init__          /root/reference/models/networks/editline2_g.py:18-43
// and the plans restate the forward methods (editline_g.py:119-221, editline2_g.py:59-94,
// editline2_model.py:128-133,338-370).
#include "../../include/sketchedit_hip.h"
#include "se_kernels.h"

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <vector>

using namespace se;

namespace {

enum { ACT_ELU = 0, ACT_RELU = 1, ACT_NONE = 2 };

struct LayerDef {
  const char* name;
  int cin, cout, k, stride, rate, act, up;
};

#define ENC(p)                                                                                    \
  {p "2_downsample", 24, 96, 3, 2, 1, ACT_ELU, 0}, {p "3", 48, 96, 3, 1, 1, ACT_ELU, 0},          \
      {p "4_downsample", 48, 192, 3, 2, 1, ACT_ELU, 0}, {p "5", 96, 192, 3, 1, 1, ACT_ELU, 0},    \
      {p "6", 96, 192, 3, 1, 1, ACT_ELU, 0}, {p "7_atrous", 96, 192, 3, 1, 2, ACT_ELU, 0},        \
      {p "8_atrous", 96, 192, 3, 1, 4, ACT_ELU, 0}, {p "9_atrous", 96, 192, 3, 1, 8, ACT_ELU, 0}, \
      {p "10_atrous", 96, 192, 3, 1, 16, ACT_ELU, 0}
#define DEC(p, c11in, c17out)                                                                        \
  {p "11", c11in, 192, 3, 1, 1, ACT_ELU, 0}, {p "12", 96, 192, 3, 1, 1, ACT_ELU, 0},                 \
      {p "13_upsample_conv", 96, 96, 3, 1, 1, ACT_ELU, 1}, {p "14", 48, 96, 3, 1, 1, ACT_ELU, 0},    \
      {p "15_upsample_conv", 48, 48, 3, 1, 1, ACT_ELU, 1}, {p "16", 24, 24, 3, 1, 1, ACT_ELU, 0},    \
      {p "17", 12, c17out, 3, 1, 1, ACT_NONE, 0}

const LayerDef G_LAYERS[] = {
    {"conv1", 5, 48, 5, 1, 1, ACT_ELU, 0}, ENC("conv"), DEC("conv", 192, 3),
    {"wconv1", 5, 48, 5, 1, 1, ACT_ELU, 0}, ENC("wconv"),
    {"xconv1", 3, 48, 5, 1, 1, ACT_ELU, 0},
    {"xconv2_downsample", 24, 48, 3, 2, 1, ACT_ELU, 0}, {"xconv3", 24, 96, 3, 1, 1, ACT_ELU, 0},
    {"xconv4_downsample", 48, 96, 3, 2, 1, ACT_ELU, 0}, {"xconv5", 48, 192, 3, 1, 1, ACT_ELU, 0},
    {"xconv6", 96, 192, 3, 1, 1, ACT_ELU, 0}, {"xconv7_atrous", 96, 192, 3, 1, 2, ACT_ELU, 0},
    {"xconv8_atrous", 96, 192, 3, 1, 4, ACT_ELU, 0}, {"xconv9_atrous", 96, 192, 3, 1, 8, ACT_ELU, 0},
    {"xconv10_atrous", 96, 192, 3, 1, 16, ACT_ELU, 0},
    {"pmconv1", 3, 48, 5, 1, 1, ACT_ELU, 0},
    {"pmconv2_downsample", 24, 48, 3, 2, 1, ACT_ELU, 0}, {"pmconv3", 24, 96, 3, 1, 1, ACT_ELU, 0},
    {"pmconv4_downsample", 48, 192, 3, 2, 1, ACT_ELU, 0}, {"pmconv5", 96, 192, 3, 1, 1, ACT_ELU, 0},
    {"pmconv6", 96, 192, 3, 1, 1, ACT_RELU, 0}, {"pmconv9", 96, 192, 3, 1, 1, ACT_ELU, 0},
    {"pmconv10", 96, 192, 3, 1, 1, ACT_ELU, 0},
    DEC("allconv", 192, 3),
};
const LayerDef M_LAYERS[] = {
    {"conv1", 4, 48, 5, 1, 1, ACT_ELU, 0}, ENC("conv"), DEC("conv", 96, 3), DEC("conv_mask_", 96, 1),
};
const int NG = sizeof(G_LAYERS) / sizeof(LayerDef), NM = sizeof(M_LAYERS) / sizeof(LayerDef);

struct Layer {
  LayerDef def;
  std::vector<float> w, b;    // host copies in checkpoint layout
  bool have_w = false, have_b = false, packed = false;
  // packed device image
  int cfg = -1, NP = 0, nch = 0, G = 0, CGp = 0, T = 0, C0 = 0, C1 = 0;
  float* d_w = nullptr;
  float* d_b = nullptr;
  float* d_u = nullptr;       // Winograd-transformed weights (eligible layers only)
  float* d_ub = nullptr;      // bias in the row order of the 48 -> 96 Winograd kernel (MIXED tiles)
  // bf16 image (BASELINE config 5): same row order and slot swizzle, 64 bf16 k-values per 128-byte row, 8-channel granules
  float* d_w16 = nullptr;
  int nch16 = 0, CGp16 = 0;
  float* d_w16d = nullptr;    // bf16, 5x5 layers whose stored input has <= 4 real channels: pair-of-taps image (pack_layer16_d4, se_rtile.hip)
  float* d_w16s = nullptr;    // 96 -> 192 3x3 only: the 32-k step image of the 8 x 16 raw-tile kernel (se_rconv16.hip)
  float* d_w96 = nullptr;     // 96-row stride-1 layers: the 32-k step image of se_rconv96.hip
  float* d_u1 = nullptr;      // two-source 96+96 -> 192 layers: Winograd image of the FIRST source's 96 channels alone, and
  float* d_wv = nullptr;      //   the second source's direct weights [9 taps][96][192 packed rows] (vector source folded into a bias)
  float* d_wv16 = nullptr;    //   the same rounded to bf16 (kept as fp32 values) for the bf16 mode
  float* d_u24 = nullptr;     // 96 -> 192 3x3: image of the hybrid F(2,3) x F(4,3) kernel (first 96 input channels), se_wino24.hip
  float* d_ub24 = nullptr;    //   and the bias in its MIXED row order
  float* d_u24b = nullptr;    //   two-source layers: the image over both sources (6 chunks per position)
  float* d_wx = nullptr;      // 24 -> 24 3x3: image of the F(2,3)-along-x raw-tile kernel (se_rtilew.hip)
  float* d_wx2 = nullptr;     //   and of its two-dimensional F(2x2,3x3) form
  float* d_wd = nullptr;      // 5x5 layers with padding channels in their stored input (fp32): dense-K image (se_rtile.hip)
  float* d_wdw = nullptr;     //   and the image of its F(2,5)-along-x form (rtile_dense5w_kernel)
  int dense = 0, nchd = 0;    //   real channels per pixel (3 or 5), 32-k chunks of the dense image
};

// ---- workspace arena: first-fit free list over [0, cap) in bytes, 256-B aligned ------------------
struct Arena {
  struct Blk { size_t off, size; bool used; };
  std::vector<Blk> blks;
  size_t cap = 0, peak = 0;
  bool dry = false;
  char* base = nullptr;
  void reset(char* b, size_t c, bool d, int which = 0) {
    // dry runs only measure: any non-null fake base (the two arenas of a plan get disjoint fake ranges)
    base = d ? (char*)(which ? ((size_t)1 << 62) + 4096 : 4096) : b;
    cap = d ? (size_t)1 << 61 : c; dry = d; peak = 0;
    blks.clear(); blks.push_back({0, cap, false});
  }
  bool owns(const void* p) const { return (size_t)p >= (size_t)base && (size_t)p < (size_t)base + cap; }
  float* alloc(size_t nfloats) {
    size_t need = (nfloats * 4 + 255) & ~(size_t)255;
    for (size_t i = 0; i < blks.size(); ++i) {
      if (!blks[i].used && blks[i].size >= need) {
        size_t off = blks[i].off;
        if (blks[i].size > need) {
          Blk rest{off + need, blks[i].size - need, false};
          blks[i].size = need;
          blks.insert(blks.begin() + i + 1, rest);
        }
        blks[i].used = true;
        if (off + need > peak) peak = off + need;
        return (float*)(base + off);
      }
    }
    return nullptr;
  }
  void release(const float* p) {
    if (!p) return;
    size_t off = (const char*)p - base;
    for (size_t i = 0; i < blks.size(); ++i) {
      if (blks[i].off == off && blks[i].used) {
        blks[i].used = false;
        if (i + 1 < blks.size() && !blks[i + 1].used) { blks[i].size += blks[i + 1].size; blks.erase(blks.begin() + i + 1); }
        if (i > 0 && !blks[i - 1].used) { blks[i - 1].size += blks[i].size; blks.erase(blks.begin() + i); }
        return;
      }
    }
  }
};

}  // namespace

struct se_ctx {
  int device = 0;
  std::mutex mu;
  std::string err;
  std::map<std::string, Layer> G, M;
  Layer wconv1_j4;          // G.wconv1 packed for a 4-channel input: with --joint_train_inp its guide channel is zero
  float* zeros = nullptr;   // zero page for out-of-bounds granules
  float* lut8 = nullptr;    // 256 floats: (v/255 - 0.5)/0.5, the dataset's normalisation of a uint8 value (se_dequantize_u8)
  Arena arena;              // workspace arena of the main branch
  Arena arena2;             // low-latency mode: arena of the concurrent side branch (disjoint region of the workspace)
  hipStream_t st = nullptr; // stream the next launch goes to
  hipStream_t st_main = nullptr;
  hipStream_t st_side = nullptr;        // side-branch stream (created with the ctx, non-blocking)
  hipEvent_t ev_fork = nullptr, ev_join = nullptr;
  bool dry = false;
  bool low_latency = false; // SE_FLAG_LOW_LATENCY of the running call
  int cur_net = SE_NET_G;   // network whose plan is running (plan_netM / plan_netG; per-op entry points: G)
  bool bf16 = false;        // SE_FLAG_BF16 of the running call: bf16 activations / weights, fp32 accumulate
  bool serial = false;             // default mode on one stream (set while the profiler is on: per-kernel durations)
  bool fork2 = true;               // SE_FORK_DEFAULT, snapshot taken where a call enters the library
  bool conservative = false;       // SE_FLAG_CONSERVATIVE of the running call: netM's 96 -> 192 layers on F(2x2,3x3)
  const se_netG_taps* taps = nullptr;      // se_netG_forward_taps: intermediate outputs of the running netG plan
  float* vbias_ws = nullptr;       // [B][9][192] scratch for the folded vector source of the next two-source layer (plan_netG)
  const float* vec32 = nullptr;    // bf16 mode: the fp32 copy of the vector source (the conv source itself is its bf16 rounding)
  unsigned char* rgb8 = nullptr;   // se_inference_u8: uint8 outputs written by the last kernel of the running call
  unsigned char* m8 = nullptr;
  Profiler prof;
  struct Peaks { size_t main, side; };
  std::map<std::vector<long long>, Peaks> peaks;      // dry-run arena peaks per (B, H, W, flags, outputs wanted)
  size_t dry_max_act = 0;                             // dry runs: bytes of the largest activation tensor the plan allocated
  std::map<std::vector<long long>, long long> act_per_img;      // (H, W, bf16) -> bytes per image of the largest activation
  struct GraphEntry {
    std::vector<long long> key;
    hipGraph_t graph = nullptr;
    hipGraphExec_t exec = nullptr;
    hipStream_t stream = nullptr;        // stream of the last launch: synchronised before the exec is destroyed
    unsigned long long last_use = 0;     // LRU stamp
  };
  unsigned long long graph_clock = 0;
  std::vector<GraphEntry> graphs;                     // SE_FLAG_GRAPH: captured forwards, keyed by every argument
};

namespace {

thread_local std::string g_create_err;

int fail(se_ctx* c, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (c) c->err = buf; else g_create_err = buf;
  return 1;
}
#define HIPCHK(c, call)                                                                   \
  do {                                                                                    \
    hipError_t e_ = (call);                                                               \
    if (e_ != hipSuccess) return fail(c, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
  } while (0)

// ---- weight packing -------------------------------------------------------------------------------
// Build the LDS image [nch][NP][32] of a gated conv: row n = packed output channel, k = flattened
// (tap, packed input channel); the 16-B slot s of row n is stored at physical slot s ^ ((n>>1)&7).
// cin_map[pc] = checkpoint input channel of packed channel pc, or -1 for zero padding.
float bf16_round(float f);
bool wino_eligible_layer(const LayerDef& d);
int pack_wino(se_ctx* c, Layer& L);
int pack_wino24(se_ctx* c, Layer& L);
int pack_rtilew(se_ctx* c, Layer& L);
bool wino48_eligible_layer(const LayerDef& d);
int pack_wino48(se_ctx* c, Layer& L);
int pack_wino48_c24(se_ctx* c, Layer& L);
bool winoup_eligible_layer(const LayerDef& d);
int pack_winoup(se_ctx* c, Layer& L);
bool winoup48_eligible_layer(const LayerDef& d);
int pack_winoup48(se_ctx* c, Layer& L);

int xcd_remap_enabled() { return opt(OPT_XCD_REMAP); }     // SE_XCD_REMAP=0 switches the XCD-aware tile order off (A/B measurements)

// Every raw-tile / Winograd / gather kernel addresses its source through a buffer resource with 32-bit BYTE offsets and uses
// byte offset 0x80000000 as the "outside the image -> hardware zero fill" sentinel, so a tensor one launch addresses must stay
// below 2^31 BYTES (the guards used to count elements).  The forwards never reach the limit: they split
// the batch into passes (batch_passes).  SE_TEST_OFFSET_LIMIT (se_debug_set_option only) lowers it so that the tests reach the
// split with a handful of small images.
long long addr_limit() {
  const int v = opt(OPT_TEST_OFFSET_LIMIT);
  return v > 0 ? (long long)v : (1ll << 31);
}

int choose_cfg(int G) {
  if (G <= 16) return GC_N24;
  if (G <= 24) return GC_N48;
  if (G <= 48) return GC_N96;
  if (G <= 96) return GC_N192;
  return -1;
}

int out_channel_of_row(int cfg, int n, int G, int cout) {
  // returns checkpoint output channel for packed row n (features [0,G), gates [G,2G)), or -1
  const int NP = gconv_np(cfg);
  if (!gconv_mixed(cfg)) {
    const int NF = NP / 32;  // feature tiles
    const int nt = n / 16, r = n % 16;
    if (nt < NF) { int f = nt * 16 + r; return f < G ? f : -1; }
    int g = (nt - NF) * 16 + r;
    return g < G ? G + g : -1;
  }
  const int nt = n / 16, r = n % 16;
  if (r < 8) { int f = nt * 8 + r; return f < G ? f : -1; }
  int g = nt * 8 + (r - 8);
  return g < G ? G + g : -1;
}

int pack_layer(se_ctx* c, Layer& L, const std::vector<int>& cin_map) {
  const LayerDef& d = L.def;
  const int G = d.cout / 2;
  const int cfg = choose_cfg(G);
  if (cfg < 0 || (G % 4)) return fail(c, "layer %s: unsupported gated width %d", d.name, G);
  const int NP = gconv_np(cfg);
  const int Cp = (int)cin_map.size();          // packed channels per tap (multiple of 4)
  // gen_deconv (nearest x2 + 3x3) is packed in its sub-pixel form: 4 output parity classes, each a 2x2 conv on
  // the source grid; tap a of class py sums the kernel rows that land on source row yy + a - 1 + py:
  //   py=0: a=0 <- {ky 0}, a=1 <- {ky 1,2};   py=1: a=0 <- {ky 0,1}, a=1 <- {ky 2}      (same for columns)
  const bool up2 = d.up != 0;
  const int KW = up2 ? 2 : d.k;
  const int T = KW * KW;
  const int K = T * Cp;
  const int nch = (K + 31) / 32;
  const int ncls = up2 ? 4 : 1;
  std::vector<float> img((size_t)ncls * nch * NP * 32, 0.f), bias(NP, 0.f);
  auto lo = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2); };
  auto hi = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2); };
  for (int cls = 0; cls < ncls; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    for (int n = 0; n < NP; ++n) {
      const int oc = out_channel_of_row(cfg, n, G, d.cout);
      if (oc < 0) continue;
      bias[n] = L.b[oc];
      for (int kf = 0; kf < K; ++kf) {
        const int tap = kf / Cp, pc = kf % Cp;
        const int ic = cin_map[pc];
        if (ic < 0) continue;
        const int ty = tap / KW, tx = tap % KW;
        float v = 0.f;
        if (up2) {
          for (int ky = lo(py, ty); ky <= hi(py, ty); ++ky)
            for (int kx = lo(px, tx); kx <= hi(px, tx); ++kx) v += L.w[(((size_t)oc * d.cin + ic) * 3 + ky) * 3 + kx];
        } else {
          v = L.w[(((size_t)oc * d.cin + ic) * d.k + ty) * d.k + tx];
        }
        const int ch = kf / 32, kin = kf % 32, s = kin / 4, e = kin % 4;
        const int ps = s ^ ((n >> 1) & 7);
        img[(((size_t)cls * nch + ch) * NP + n) * 32 + ps * 4 + e] = v;
      }
    }
  }
  if (L.d_w) (void)hipFree(L.d_w);
  if (L.d_b) (void)hipFree(L.d_b);
  HIPCHK(c, hipMalloc(&L.d_w, img.size() * 4));
  HIPCHK(c, hipMalloc(&L.d_b, bias.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_w, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(L.d_b, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  L.cfg = cfg; L.NP = NP; L.nch = nch; L.G = G; L.CGp = Cp / 4; L.T = T;
  L.packed = true;
  if (wino_eligible_layer(d) && Cp == d.cin) return pack_wino(c, L);
  if (d.k == 3 && d.stride == 1 && !d.up && d.cin == 48 && d.cout == 192 && d.act != ACT_NONE && Cp == d.cin) return pack_wino24(c, L);      // xconv5
  if (d.k == 3 && d.stride == 1 && !d.up && d.rate == 1 && d.cin == 24 && d.cout == 24 && d.act != ACT_NONE && Cp == d.cin) return pack_rtilew(c, L);   // conv16
  if (wino48_eligible_layer(d) && Cp == d.cin) return pack_wino48(c, L);
  if (d.k == 3 && d.stride == 1 && !d.up && d.cin == 24 && d.cout == 96 && d.act != ACT_NONE && Cp == d.cin) return pack_wino48_c24(c, L);   // xconv3, pmconv3
  if (winoup_eligible_layer(d) && Cp == d.cin) return pack_winoup(c, L);
  if (winoup48_eligible_layer(d) && Cp == d.cin) return pack_winoup48(c, L);
  return 0;
}

// Dense-K image of a 5x5 first layer (se_rtile.hip rtile_dense5_kernel): k = tap * Cd + channel over the Cd channels the
// stored input really carries (cin_map[pc] = checkpoint input channel of stored channel pc), no channel padding; in the
// last chunk the real k are packed instruction-major (k-step (half, r) holds the four k of lane groups 0..3), so the
// kernel issues exactly ceil(K / 4) MFMA k-steps.  Rows in the MIXED order of the N=48 configuration, slot swizzle as
// pack_layer.
int pack_layer_dense(se_ctx* c, Layer& L, const std::vector<int>& cin_map) {
  const LayerDef& d = L.def;
  const int Cd = (int)cin_map.size();
  const int G = d.cout / 2, NP = 48, T = d.k * d.k, K = T * Cd, nch = (K + 31) / 32;
  std::vector<float> img((size_t)nch * NP * 32, 0.f);
  for (int n = 0; n < NP; ++n) {
    const int oc = out_channel_of_row(GC_N48, n, G, d.cout);
    if (oc < 0) continue;
    for (int kf = 0; kf < K; ++kf) {
      const int tap = kf / Cd, ic = cin_map[kf % Cd], ty = tap / d.k, tx = tap % d.k;
      const int ch = kf / 32;
      int kin = kf % 32;
      {      // dense_kin: j-th k of a chunk -> k-step j / 4, lane group j % 4 (instruction-major).  In the last chunk that packs the
             // real k into the first k-steps (the others are not issued); in every chunk it makes the four lane groups of one
             // k-step read four CONSECUTIVE dwords of the dense tile (bank-conflict-free; k-major order: 32-50 % conflict cycles)
        const int j = kin, step = j / 4, g = j % 4;
        kin = (step / 4) * 16 + g * 4 + (step % 4);
      }
      const int s_ = kin / 4, e = kin % 4, ps = s_ ^ ((n >> 1) & 7);
      img[((size_t)ch * NP + n) * 32 + ps * 4 + e] = L.w[(((size_t)oc * d.cin + ic) * d.k + ty) * d.k + tx];
    }
  }
  if (L.d_wd) (void)hipFree(L.d_wd);
  HIPCHK(c, hipMalloc(&L.d_wd, img.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_wd, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  L.dense = Cd; L.nchd = nch;
  // F(2,5)-along-x form (rtile_dense5w_kernel): U[nu][ky][c] = sum_kx Gx[nu][kx] w[ky][kx]; one chunk per position with
  // k = ky * Cd + c in the same instruction-major order
  {
    static const double Gx[6][5] = {{1. / 4, 0., 0., 0., 0.}, {-1. / 6, -1. / 6, -1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6, 1. / 6, -1. / 6},
                                    {1. / 24, 1. / 12, 1. / 6, 1. / 3, 2. / 3}, {1. / 24, -1. / 12, 1. / 6, -1. / 3, 2. / 3}, {0., 0., 0., 0., 1.}};
    std::vector<float> imw((size_t)6 * NP * 32, 0.f);
    for (int n = 0; n < NP; ++n) {
      const int oc = out_channel_of_row(GC_N48, n, G, d.cout);
      if (oc < 0) continue;
      for (int j = 0; j < 5 * Cd; ++j) {
        const int ky = j / Cd, ic = cin_map[j % Cd];
        const float* g = &L.w[(((size_t)oc * d.cin + ic) * 5 + ky) * 5];
        const int step = j / 4, gq = j % 4, kin = (step / 4) * 16 + gq * 4 + (step % 4);
        const int s_ = kin / 4, e = kin % 4, ps = s_ ^ ((n >> 1) & 7);
        for (int nu = 0; nu < 6; ++nu) {
          double u = 0.;
          for (int kx = 0; kx < 5; ++kx) u += Gx[nu][kx] * (double)g[kx];
          imw[((size_t)nu * NP + n) * 32 + ps * 4 + e] = (float)u;
        }
      }
    }
    if (L.d_wdw) (void)hipFree(L.d_wdw);
    HIPCHK(c, hipMalloc(&L.d_wdw, imw.size() * 4));
    HIPCHK(c, hipMemcpy(L.d_wdw, imw.data(), imw.size() * 4, hipMemcpyHostToDevice));
  }
  return 0;
}

// fp32 -> bf16, round to nearest even (the rounding of v_cvt_pk_bf16_f32 and of torch's .to(bfloat16))
unsigned short bf16_bits(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (unsigned short)((u >> 16) | 0x40);   // NaN stays NaN
  u += 0x7fffu + ((u >> 16) & 1u);
  return (unsigned short)(u >> 16);
}
float bf16_round(float f) {
  const unsigned u = (unsigned)bf16_bits(f) << 16;
  float r;
  memcpy(&r, &u, 4);
  return r;
}

// bf16 image of a gated conv: [class][chunk of 64 k][NP rows][64 bf16], k = flattened (tap, packed input channel) with the
// channels of a tap padded to a multiple of 8 (one 16-byte granule = 8 channels); slot s of row n at s ^ ((n>>1)&7).
// The sub-pixel sums of gen_deconv are formed in fp32 and rounded once.
int pack_layer16(se_ctx* c, Layer& L, const std::vector<int>& cin_map) {
  const LayerDef& d = L.def;
  const int G = d.cout / 2;
  const int cfg = choose_cfg(G);
  if (cfg < 0 || (G % 4)) return fail(c, "layer %s: unsupported gated width %d", d.name, G);
  const int NP = gconv_np(cfg);
  const int Cp = (int)cin_map.size();          // packed channels per tap (multiple of 8)
  const bool up2 = d.up != 0;
  const int KW = up2 ? 2 : d.k;
  const int T = KW * KW;
  const int K = T * Cp;
  const int nch = (K + 63) / 64;
  const int ncls = up2 ? 4 : 1;
  std::vector<unsigned short> img((size_t)ncls * nch * NP * 64, 0);
  auto lo = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2); };
  auto hi = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2); };
  for (int cls = 0; cls < ncls; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    for (int n = 0; n < NP; ++n) {
      const int oc = out_channel_of_row(cfg, n, G, d.cout);
      if (oc < 0) continue;
      for (int kf = 0; kf < K; ++kf) {
        const int tap = kf / Cp, pc = kf % Cp;
        const int ic = cin_map[pc];
        if (ic < 0) continue;
        const int ty = tap / KW, tx = tap % KW;
        float v = 0.f;
        if (up2) {
          for (int ky = lo(py, ty); ky <= hi(py, ty); ++ky)
            for (int kx = lo(px, tx); kx <= hi(px, tx); ++kx) v += L.w[(((size_t)oc * d.cin + ic) * 3 + ky) * 3 + kx];
        } else {
          v = L.w[(((size_t)oc * d.cin + ic) * d.k + ty) * d.k + tx];
        }
        const int ch = kf / 64, kin = kf % 64, s_ = kin / 8, e = kin % 8;
        const int ps = s_ ^ ((n >> 1) & 7);
        img[(((size_t)cls * nch + ch) * NP + n) * 64 + ps * 8 + e] = bf16_bits(v);
      }
    }
  }
  if (L.d_w16) (void)hipFree(L.d_w16);
  HIPCHK(c, hipMalloc(&L.d_w16, img.size() * 2));
  HIPCHK(c, hipMemcpy(L.d_w16, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  L.nch16 = nch; L.CGp16 = Cp / 8;
  return 0;
}

// bf16 pair-of-taps image of a 5x5 first layer whose stored NHWC8 input carries at most four real channels (rtile_kernel<3, 8,
// true, true>): granule gi = 3 ky + j (j = 0..2) holds the taps (ky, 2j) and (ky, 2j + 1) x stored channels 0-3, i.e. element
// e = 4 (kx & 1) + c; kx = 5 does not exist (zero).  15 granules -> 2 chunks of 64 k; rows in the MIXED N=48 order, slot swizzle
// as pack_layer16.  cin4[c] = checkpoint input channel of stored channel c, or -1.
int pack_layer16_d4(se_ctx* c, Layer& L, const int (&cin4)[4]) {
  const LayerDef& d = L.def;
  const int G = d.cout / 2, NP = 48, nch = 2;
  std::vector<unsigned short> img((size_t)nch * NP * 64, 0);
  for (int n = 0; n < NP; ++n) {
    const int oc = out_channel_of_row(GC_N48, n, G, d.cout);
    if (oc < 0) continue;
    for (int ky = 0; ky < 5; ++ky)
      for (int kx = 0; kx < 5; ++kx)
        for (int cc = 0; cc < 4; ++cc) {
          const int ic = cin4[cc];
          if (ic < 0) continue;
          const int gi = 3 * ky + kx / 2, e = 4 * (kx & 1) + cc;
          const int ch = gi / 8, s_ = gi % 8, ps = s_ ^ ((n >> 1) & 7);
          img[((size_t)ch * NP + n) * 64 + ps * 8 + e] = bf16_bits(L.w[(((size_t)oc * d.cin + ic) * 5 + ky) * 5 + kx]);
        }
  }
  if (L.d_w16d) (void)hipFree(L.d_w16d);
  HIPCHK(c, hipMalloc(&L.d_w16d, img.size() * 2));
  HIPCHK(c, hipMemcpy(L.d_w16d, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// bf16 image for rconv16b_kernel (96 -> 192, 3x3): [27 steps = tap * 3 + 32-channel group][12 row tiles][16 rows][32 k],
// rows in the N=192 order (features, then gates); granule g (8 k) of row r at slot g ^ F[r >> 2], F = {0, 2, 3, 1}.
int pack_rconv16(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const int F[4] = {0, 2, 3, 1};
  std::vector<unsigned short> img((size_t)27 * 192 * 32, 0);
  for (int n = 0; n < 192; ++n) {
    const int oc = out_channel_of_row(GC_N192, n, 96, 192);
    const int rt = n / 16, r = n % 16;
    for (int s = 0; s < 27; ++s) {
      const int tap = s / 3, kk = s % 3, ty = tap / 3, tx = tap % 3;
      for (int e = 0; e < 32; ++e) {
        const int ic = kk * 32 + e, g = e / 8;
        const float v = L.w[(((size_t)oc * d.cin + ic) * 3 + ty) * 3 + tx];
        img[((size_t)s * 12 + rt) * 512 + r * 32 + ((g ^ F[r >> 2]) * 8) + (e % 8)] = bf16_bits(v);
      }
    }
  }
  if (L.d_w16s) (void)hipFree(L.d_w16s);
  HIPCHK(c, hipMalloc(&L.d_w16s, img.size() * 2));
  HIPCHK(c, hipMemcpy(L.d_w16s, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// bf16 image for rconv96_kernel (96 packed rows: 3x3 24/48 -> 96, gen_deconv 96 -> 96):
// [class][step][6 row tiles][16 rows][32 k], k = granule (8 channels) index tap * CG + cg, four granules per step; rows in
// the N=96 order (features, then gates); granule g of row r at slot g ^ F[r >> 2], F = {0, 2, 3, 1}.
bool rconv96_eligible(const LayerDef& d) {
  if (d.cout != 96 || d.rate != 1 || d.k != 3 || d.act == ACT_NONE) return false;
  if (d.stride == 2) return !d.up && d.cin == 24;                     // stride 2: the 24 -> 96 downsampling layers
  if (d.stride != 1) return false;
  return d.up ? d.cin == 96 : (d.cin == 48 || d.cin == 24);
}
int pack_rconv96(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const int F[4] = {0, 2, 3, 1};
  const bool up2 = d.up != 0;
  const int KW = up2 ? 2 : 3, T = KW * KW, CG = d.cin / 8, NG = T * CG, nstep = (NG + 3) / 4, ncls = up2 ? 4 : 1;
  std::vector<unsigned short> img((size_t)ncls * nstep * 96 * 32, 0);
  auto lo = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2); };
  auto hi = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2); };
  for (int cls = 0; cls < ncls; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    for (int n = 0; n < 96; ++n) {
      const int oc = out_channel_of_row(GC_N96, n, 48, 96);
      const int rt = n / 16, r = n % 16;
      for (int gi = 0; gi < NG; ++gi) {
        const int tap = gi / CG, cg = gi % CG, ty = tap / KW, tx = tap % KW, s_ = gi / 4, g = gi % 4;
        for (int e = 0; e < 8; ++e) {
          const int ic = cg * 8 + e;
          float v = 0.f;
          if (up2) {
            for (int ky = lo(py, ty); ky <= hi(py, ty); ++ky)
              for (int kx = lo(px, tx); kx <= hi(px, tx); ++kx) v += L.w[(((size_t)oc * d.cin + ic) * 3 + ky) * 3 + kx];
          } else {
            v = L.w[(((size_t)oc * d.cin + ic) * 3 + ty) * 3 + tx];
          }
          img[(((size_t)cls * nstep + s_) * 6 + rt) * 512 + r * 32 + ((g ^ F[r >> 2]) * 8) + e] = bf16_bits(v);
        }
      }
    }
  }
  if (L.d_w96) (void)hipFree(L.d_w96);
  HIPCHK(c, hipMalloc(&L.d_w96, img.size() * 2));
  HIPCHK(c, hipMemcpy(L.d_w96, img.data(), img.size() * 2, hipMemcpyHostToDevice));
  return 0;
}

// Winograd F(2x2,3x3) weights: U[pos] = (G g G^T)[xi][nu] per (out, in) pair, packed per position like a 1x1
// conv Cin -> 192 (Cin = 96, or 192 for the two-source layers) in the N=192 row order (features then gates)
// with the same slot swizzle.
bool wino_eligible_layer(const LayerDef& d) {
  return d.k == 3 && d.stride == 1 && !d.up && (d.cin == 96 || d.cin == 192) && d.cout == 192 && d.act != ACT_NONE;
}
int pack_wino(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int NP = 192, nch = d.cin / 32;
  std::vector<float> img((size_t)16 * nch * NP * 32, 0.f);
  for (int n = 0; n < NP; ++n) {
    const int oc = out_channel_of_row(GC_N192, n, 96, 192);
    for (int ic = 0; ic < d.cin; ++ic) {
      const float* g = &L.w[((size_t)oc * d.cin + ic) * 9];
      float t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int kx = 0; kx < 3; ++kx) t[i][kx] = Gm[i][0] * g[kx] + Gm[i][1] * g[3 + kx] + Gm[i][2] * g[6 + kx];
      for (int xi = 0; xi < 4; ++xi)
        for (int nu = 0; nu < 4; ++nu) {
          const float u = t[xi][0] * Gm[nu][0] + t[xi][1] * Gm[nu][1] + t[xi][2] * Gm[nu][2];
          const int pos = xi * 4 + nu, ch = ic / 32, kin = ic % 32, s_ = kin / 4, e = kin % 4;
          const int ps = s_ ^ ((n >> 1) & 7);
          img[(((size_t)pos * nch + ch) * NP + n) * 32 + ps * 4 + e] = u;
        }
    }
  }
  if (L.d_u) (void)hipFree(L.d_u);
  HIPCHK(c, hipMalloc(&L.d_u, img.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_u, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  if (d.cin == 192) {
    // for a spatially constant second source (conv11 of netG: the pooled style vector) the layer runs as the single-source
    // kernel on the first 96 channels plus a per-image bias table (launch_vecbias): the first source's Winograd image
    // ([16 positions][3 chunks][192][32], the first three chunks of every position of `img`) and the second source's
    // DIRECT weights [tap][channel][packed row]
    std::vector<float> img1((size_t)16 * 3 * NP * 32), wv((size_t)9 * 96 * NP);
    for (int pos = 0; pos < 16; ++pos)
      memcpy(&img1[(size_t)pos * 3 * NP * 32], &img[(size_t)pos * nch * NP * 32], (size_t)3 * NP * 32 * 4);
    for (int n = 0; n < NP; ++n) {
      const int oc = out_channel_of_row(GC_N192, n, 96, 192);
      for (int t = 0; t < 9; ++t)
        for (int ch = 0; ch < 96; ++ch) wv[((size_t)t * 96 + ch) * NP + n] = L.w[((size_t)oc * d.cin + 96 + ch) * 9 + t];
    }
    if (L.d_u1) (void)hipFree(L.d_u1);
    if (L.d_wv) (void)hipFree(L.d_wv);
    if (L.d_wv16) (void)hipFree(L.d_wv16);
    HIPCHK(c, hipMalloc(&L.d_u1, img1.size() * 4));
    HIPCHK(c, hipMalloc(&L.d_wv, wv.size() * 4));
    HIPCHK(c, hipMalloc(&L.d_wv16, wv.size() * 4));
    HIPCHK(c, hipMemcpy(L.d_u1, img1.data(), img1.size() * 4, hipMemcpyHostToDevice));
    HIPCHK(c, hipMemcpy(L.d_wv, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
    for (auto& v : wv) v = bf16_round(v);
    HIPCHK(c, hipMemcpy(L.d_wv16, wv.data(), wv.size() * 4, hipMemcpyHostToDevice));
  }
  return pack_wino24(c, L);
}

// 24 -> 24 layers (se_rtilew.hip): U[nu][ky] = G g[ky][.] with the F(2,3) G along x; k = ky * 24 + channel in three 32-k
// chunks per position (72 k, the third chunk half empty), 24 PHYSICAL rows -- tile 0 = features 0-7, gates 0-7; then
// features 8-11, gates 8-11 (the padding rows of the second MIXED tile read these again) --, slot swizzle by physical row.
int pack_rtilew(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int G = d.cout / 2;      // 12
  std::vector<float> img((size_t)4 * 3 * 24 * 32, 0.f);
  for (int prow = 0; prow < 24; ++prow) {
    const int oc = prow < 8 ? prow : prow < 16 ? G + (prow - 8) : prow < 20 ? 8 + (prow - 16) : G + 8 + (prow - 20);
    for (int ic = 0; ic < 24; ++ic)
      for (int ky = 0; ky < 3; ++ky) {
        const float* g = &L.w[(((size_t)oc * d.cin + ic) * 3 + ky) * 3];
        for (int nu = 0; nu < 4; ++nu) {
          const float u = Gm[nu][0] * g[0] + Gm[nu][1] * g[1] + Gm[nu][2] * g[2];
          const int k = ky * 24 + ic, ch = k / 32, kin = k % 32, s_ = kin / 4, e = kin % 4;
          const int ps = s_ ^ ((prow >> 1) & 7);
          img[(((size_t)nu * 3 + ch) * 24 + prow) * 32 + ps * 4 + e] = u;
        }
      }
  }
  if (L.d_wx) (void)hipFree(L.d_wx);
  HIPCHK(c, hipMalloc(&L.d_wx, img.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_wx, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  // two-dimensional form: U = G g G^T per position; a row holds its 24 k as channels 0-15 in slots 0-3 and channels
  // 16 + 2q, 17 + 2q in slot 4 + q, elements 0, 1 for even q and 2, 3 for odd q (k-half 1 issues two k-steps; the halves keep
  // the 8-byte fragment reads of lane groups q, q ^ 1 off each other's banks, se_rtilew.hip)
  std::vector<float> img2((size_t)16 * 24 * 32, 0.f);
  for (int prow = 0; prow < 24; ++prow) {
    const int oc = prow < 8 ? prow : prow < 16 ? G + (prow - 8) : prow < 20 ? 8 + (prow - 16) : G + 8 + (prow - 20);
    for (int ic = 0; ic < 24; ++ic) {
      const float* g = &L.w[((size_t)oc * d.cin + ic) * 9];
      float t[4][3];
      for (int i = 0; i < 4; ++i)
        for (int kx = 0; kx < 3; ++kx) t[i][kx] = Gm[i][0] * g[kx] + Gm[i][1] * g[3 + kx] + Gm[i][2] * g[6 + kx];
      for (int pos = 0; pos < 16; ++pos) {
        const int xi = pos >> 2, nu = pos & 3;
        const float u = t[xi][0] * Gm[nu][0] + t[xi][1] * Gm[nu][1] + t[xi][2] * Gm[nu][2];
        const int s_ = ic < 16 ? ic / 4 : 4 + (ic - 16) / 2, e = ic < 16 ? ic % 4 : (ic - 16) % 2 + 2 * (((ic - 16) / 2) & 1);
        const int ps = s_ ^ ((prow >> 1) & 7);
        img2[((size_t)pos * 24 + prow) * 32 + ps * 4 + e] = u;
      }
    }
  }
  if (L.d_wx2) (void)hipFree(L.d_wx2);
  HIPCHK(c, hipMalloc(&L.d_wx2, img2.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_wx2, img2.data(), img2.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// Hybrid F(2,3) x F(4,3) image of the same layers (se_wino24.hip): U = Gy g Gx^T (4 x 6 positions) of the FIRST 96 input
// channels, 72 iterations in the kernel's order -- stage (xi, h) -> chunk -> j with column position nu = {0,1,2}[j] (h = 0)
// or {5,3,4}[j] (h = 1) --, 192 rows in the MIXED order (tile t = features 8t..8t+7, then their gates), slot swizzle as
// everywhere.  U is formed in double and rounded once (Gx holds 1/6, 1/12, 1/24).
int pack_wino24(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const double Gy[4][3] = {{1., 0., 0.}, {.5, .5, .5}, {.5, -.5, .5}, {0., 0., 1.}};
  static const double Gx[6][3] = {{1. / 4, 0., 0.}, {-1. / 6, -1. / 6, -1. / 6}, {-1. / 6, 1. / 6, -1. / 6},
                                  {1. / 24, 1. / 12, 1. / 6}, {1. / 24, -1. / 12, 1. / 6}, {0., 0., 1.}};
  static const int NU[2][3] = {{0, 1, 2}, {5, 3, 4}};
  const int NP = 192;
  // two images for the two-source layers: the first source alone (vector source folded into a bias: conv11) and both
  // sources (allconv11: 6 chunks per position)
  // (a 48-channel layer -- xconv5 -- has one image of 2 chunks per position, the second half empty)
  const int nchk_first = d.cin == 48 ? 2 : 3, nchk_last = d.cin == 48 ? 2 : d.cin / 32;
  for (int nchk = nchk_first; nchk <= nchk_last; nchk += 3) {
  std::vector<float> img((size_t)24 * nchk * NP * 32, 0.f), bias(NP, 0.f);
  for (int n = 0; n < NP; ++n) {
    const int t = n / 16, r = n % 16;
    const int oc = r < 8 ? t * 8 + r : 96 + t * 8 + (r - 8);
    bias[n] = L.b[oc];
    for (int ic = 0; ic < nchk * 32 && ic < d.cin; ++ic) {
      const float* g = &L.w[((size_t)oc * d.cin + ic) * 9];
      double tt[4][3];
      for (int i = 0; i < 4; ++i)
        for (int kx = 0; kx < 3; ++kx) tt[i][kx] = Gy[i][0] * g[kx] + Gy[i][1] * g[3 + kx] + Gy[i][2] * g[6 + kx];
      for (int xi = 0; xi < 4; ++xi)
        for (int h = 0; h < 2; ++h)
          for (int j = 0; j < 3; ++j) {
            const int nu = NU[h][j];
            const double u = tt[xi][0] * Gx[nu][0] + tt[xi][1] * Gx[nu][1] + tt[xi][2] * Gx[nu][2];
            const int chunk = ic / 32, kin = ic % 32, s_ = kin / 4, e = kin % 4;
            const int it = ((xi * 2 + h) * nchk + chunk) * 3 + j;
            const int ps = s_ ^ ((n >> 1) & 7);
            img[((size_t)it * NP + n) * 32 + ps * 4 + e] = (float)u;
          }
    }
  }
  float*& du = nchk <= 3 ? L.d_u24 : L.d_u24b;
  if (du) (void)hipFree(du);
  HIPCHK(c, hipMalloc(&du, img.size() * 4));
  HIPCHK(c, hipMemcpy(du, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  if (nchk <= 3) {
    if (L.d_ub24) (void)hipFree(L.d_ub24);
    HIPCHK(c, hipMalloc(&L.d_ub24, bias.size() * 4));
    HIPCHK(c, hipMemcpy(L.d_ub24, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  }
  }
  return 0;
}

// 48 -> 96 layers (se_wino48.hip): 24 iterations = 8 position pairs x 3 chunks; chunk c of pair pp holds in its
// k-half h the 16-channel group ((2c+h) % 3) of position 2pp + ((2c+h) >= 3).  Rows in the MIXED order: tile t =
// features 8t..8t+7, then their gates.
bool wino48_eligible_layer(const LayerDef& d) {
  return d.k == 3 && d.stride == 1 && !d.up && d.cin == 48 && d.cout == 96 && d.act != ACT_NONE;
}
int pack_wino48(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int NP = 96;
  std::vector<float> img((size_t)24 * NP * 32, 0.f), bias(NP, 0.f);
  for (int n = 0; n < NP; ++n) {
    const int t = n / 16, r = n % 16;
    const int oc = r < 8 ? t * 8 + r : 48 + t * 8 + (r - 8);
    bias[n] = L.b[oc];
    for (int ic = 0; ic < 48; ++ic) {
      const float* g = &L.w[((size_t)oc * d.cin + ic) * 9];
      float tt[4][3];
      for (int i = 0; i < 4; ++i)
        for (int kx = 0; kx < 3; ++kx) tt[i][kx] = Gm[i][0] * g[kx] + Gm[i][1] * g[3 + kx] + Gm[i][2] * g[6 + kx];
      for (int pos = 0; pos < 16; ++pos) {
        const int xi = pos >> 2, nu = pos & 3;
        const float u = tt[xi][0] * Gm[nu][0] + tt[xi][1] * Gm[nu][1] + tt[xi][2] * Gm[nu][2];
        const int pp = pos >> 1, u6 = (pos & 1) * 3 + ic / 16;        // index of the 16-channel group in the pair
        const int it = pp * 3 + u6 / 2, kin = (u6 % 2) * 16 + ic % 16;
        const int s_ = kin / 4, e = kin % 4;
        const int ps = s_ ^ ((n >> 1) & 7);
        img[((size_t)it * NP + n) * 32 + ps * 4 + e] = u;
      }
    }
  }
  if (L.d_u) (void)hipFree(L.d_u);
  if (L.d_ub) (void)hipFree(L.d_ub);
  HIPCHK(c, hipMalloc(&L.d_u, img.size() * 4));
  HIPCHK(c, hipMalloc(&L.d_ub, bias.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_u, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(L.d_ub, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// 24 -> 96 layers on the same kernel (se_wino48.hip, CIN = 24): one iteration per position, U[pos] as a [96 MIXED rows][32 k]
// tile: channels 0-15 in slots 0-3, channels 16 + 2q, 17 + 2q in elements 0, 1 of slot 4 + q (k-half 1 issues two k-steps:
// a k-step takes one element of every slot), elements 2, 3 zero.
int pack_wino48_c24(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const float Gm[4][3] = {{1.f, 0.f, 0.f}, {.5f, .5f, .5f}, {.5f, -.5f, .5f}, {0.f, 0.f, 1.f}};
  const int NP = 96;
  std::vector<float> img((size_t)16 * NP * 32, 0.f), bias(NP, 0.f);
  for (int n = 0; n < NP; ++n) {
    const int t = n / 16, r = n % 16;
    const int oc = r < 8 ? t * 8 + r : 48 + t * 8 + (r - 8);
    bias[n] = L.b[oc];
    for (int ic = 0; ic < 24; ++ic) {
      const float* g = &L.w[((size_t)oc * d.cin + ic) * 9];
      float tt[4][3];
      for (int i = 0; i < 4; ++i)
        for (int kx = 0; kx < 3; ++kx) tt[i][kx] = Gm[i][0] * g[kx] + Gm[i][1] * g[3 + kx] + Gm[i][2] * g[6 + kx];
      for (int pos = 0; pos < 16; ++pos) {
        const int xi = pos >> 2, nu = pos & 3;
        const float u = tt[xi][0] * Gm[nu][0] + tt[xi][1] * Gm[nu][1] + tt[xi][2] * Gm[nu][2];
        const int s_ = ic < 16 ? ic / 4 : 4 + (ic - 16) / 2, e = ic < 16 ? ic % 4 : (ic - 16) % 2, ps = s_ ^ ((n >> 1) & 7);
        img[((size_t)pos * NP + n) * 32 + ps * 4 + e] = u;
      }
    }
  }
  if (L.d_u) (void)hipFree(L.d_u);
  if (L.d_ub) (void)hipFree(L.d_ub);
  HIPCHK(c, hipMalloc(&L.d_u, img.size() * 4));
  HIPCHK(c, hipMalloc(&L.d_ub, bias.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_u, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(L.d_ub, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// gen_deconv 96 -> 96 (se_wino_up.hip): per output parity class the pre-summed 2x2 weights g (pack_layer) are
// transformed with G = [1 0; 1 1; 0 1]: U = G g G^T (3x3 positions), 27 iterations = 9 positions x 3 chunks of 32
// channels, rows in the MIXED order.
bool winoup_eligible_layer(const LayerDef& d) {
  return d.k == 3 && d.stride == 1 && d.up && d.cin == 96 && d.cout == 96 && d.act != ACT_NONE;
}
int pack_winoup(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const float Gm[3][2] = {{1.f, 0.f}, {1.f, 1.f}, {0.f, 1.f}};
  const int NP = 96;
  std::vector<float> img((size_t)4 * 27 * NP * 32, 0.f), bias(NP, 0.f);
  auto lo = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2); };
  auto hi = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2); };
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    for (int n = 0; n < NP; ++n) {
      const int t = n / 16, r = n % 16;
      const int oc = r < 8 ? t * 8 + r : 48 + t * 8 + (r - 8);
      bias[n] = L.b[oc];
      for (int ic = 0; ic < 96; ++ic) {
        float g[2][2];
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 2; ++b) {
            float v = 0.f;
            for (int ky = lo(py, a); ky <= hi(py, a); ++ky)
              for (int kx = lo(px, b); kx <= hi(px, b); ++kx) v += L.w[(((size_t)oc * d.cin + ic) * 3 + ky) * 3 + kx];
            g[a][b] = v;
          }
        for (int xi = 0; xi < 3; ++xi)
          for (int nu = 0; nu < 3; ++nu) {
            float u = 0.f;
            for (int a = 0; a < 2; ++a)
              for (int b = 0; b < 2; ++b) u += Gm[xi][a] * Gm[nu][b] * g[a][b];
            const int it = (xi * 3 + nu) * 3 + ic / 32, kin = ic % 32;
            const int s_ = kin / 4, e = kin % 4;
            const int ps = s_ ^ ((n >> 1) & 7);
            img[(((size_t)cls * 27 + it) * NP + n) * 32 + ps * 4 + e] = u;
          }
      }
    }
  }
  if (L.d_u) (void)hipFree(L.d_u);
  if (L.d_ub) (void)hipFree(L.d_ub);
  HIPCHK(c, hipMalloc(&L.d_u, img.size() * 4));
  HIPCHK(c, hipMalloc(&L.d_ub, bias.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_u, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(L.d_ub, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

// gen_deconv 48 -> 48 (se_wino_up48.hip): U = G g G^T per class as above; 14 iterations in the pairing of pack_wino48
// (chunk c of pair pp holds in k-half h the 16-channel group ((2c+h) % 3) of position 2pp + ((2c+h) >= 3)); position 8
// has no partner: the second k-half of iteration 13 stays zero.  48 rows in the MIXED order (8 features + their gates).
bool winoup48_eligible_layer(const LayerDef& d) {
  return d.k == 3 && d.stride == 1 && d.up && d.cin == 48 && d.cout == 48 && d.act != ACT_NONE;
}
int pack_winoup48(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  static const float Gm[3][2] = {{1.f, 0.f}, {1.f, 1.f}, {0.f, 1.f}};
  const int NP = 48, NIT = 14;
  std::vector<float> img((size_t)4 * NIT * NP * 32, 0.f), bias(NP, 0.f);
  auto lo = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 1) : (a == 0 ? 0 : 2); };
  auto hi = [](int par, int a) { return par == 0 ? (a == 0 ? 0 : 2) : (a == 0 ? 1 : 2); };
  for (int cls = 0; cls < 4; ++cls) {
    const int py = cls >> 1, px = cls & 1;
    for (int n = 0; n < NP; ++n) {
      const int t = n / 16, r = n % 16;
      const int oc = r < 8 ? t * 8 + r : 24 + t * 8 + (r - 8);
      bias[n] = L.b[oc];
      for (int ic = 0; ic < 48; ++ic) {
        float g[2][2];
        for (int a = 0; a < 2; ++a)
          for (int b = 0; b < 2; ++b) {
            float v = 0.f;
            for (int ky = lo(py, a); ky <= hi(py, a); ++ky)
              for (int kx = lo(px, b); kx <= hi(px, b); ++kx) v += L.w[(((size_t)oc * d.cin + ic) * 3 + ky) * 3 + kx];
            g[a][b] = v;
          }
        for (int pos = 0; pos < 9; ++pos) {
          const int xi = pos / 3, nu = pos % 3;
          float u = 0.f;
          for (int a = 0; a < 2; ++a)
            for (int b = 0; b < 2; ++b) u += Gm[xi][a] * Gm[nu][b] * g[a][b];
          const int pp = pos >> 1, u6 = (pos & 1) * 3 + ic / 16;        // index of the 16-channel group in the pair
          const int it = pp * 3 + u6 / 2, kin = (u6 % 2) * 16 + ic % 16;
          const int s_ = kin / 4, e = kin % 4;
          const int ps = s_ ^ ((n >> 1) & 7);
          img[(((size_t)cls * NIT + it) * NP + n) * 32 + ps * 4 + e] = u;
        }
      }
    }
  }
  if (L.d_u) (void)hipFree(L.d_u);
  if (L.d_ub) (void)hipFree(L.d_ub);
  HIPCHK(c, hipMalloc(&L.d_u, img.size() * 4));
  HIPCHK(c, hipMalloc(&L.d_ub, bias.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_u, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(L.d_ub, bias.data(), bias.size() * 4, hipMemcpyHostToDevice));
  return 0;
}

int pack_small(se_ctx* c, Layer& L) {
  // raw 3x3 conv 12 -> cout: [cout][9][12]
  const LayerDef& d = L.def;
  std::vector<float> img((size_t)d.cout * 9 * 12);
  for (int oc = 0; oc < d.cout; ++oc)
    for (int t = 0; t < 9; ++t)
      for (int ic = 0; ic < 12; ++ic) img[((size_t)oc * 9 + t) * 12 + ic] = L.w[(((size_t)oc * 12 + ic) * 3 + t / 3) * 3 + t % 3];
  if (L.d_w) (void)hipFree(L.d_w);
  if (L.d_b) (void)hipFree(L.d_b);
  HIPCHK(c, hipMalloc(&L.d_w, img.size() * 4));
  HIPCHK(c, hipMalloc(&L.d_b, d.cout * 4));
  HIPCHK(c, hipMemcpy(L.d_w, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  HIPCHK(c, hipMemcpy(L.d_b, L.b.data(), d.cout * 4, hipMemcpyHostToDevice));
  // bf16 mode: the same fp32 kernel with bf16-rounded weights (every conv weight is rounded in that mode)
  for (auto& v : img) v = bf16_round(v);
  if (L.d_w16) (void)hipFree(L.d_w16);
  HIPCHK(c, hipMalloc(&L.d_w16, img.size() * 4));
  HIPCHK(c, hipMemcpy(L.d_w16, img.data(), img.size() * 4, hipMemcpyHostToDevice));
  L.packed = true;
  return 0;
}

std::vector<int> identity_map(int cin) {
  const int Cp = (cin + 3) & ~3;
  std::vector<int> m(Cp, -1);
  for (int i = 0; i < cin; ++i) m[i] = i;
  return m;
}

std::vector<int> identity_map8(int cin) {      // bf16: channels of a tap padded to whole 8-channel granules
  const int Cp = (cin + 7) & ~7;
  std::vector<int> m(Cp, -1);
  for (int i = 0; i < cin; ++i) m[i] = i;
  return m;
}

int pack_net_layer(se_ctx* c, Layer& L) {
  const LayerDef& d = L.def;
  if (d.act == ACT_NONE) return pack_small(c, L);
  {
    // bf16 images: conv16's 12 gated outputs are stored with a 16-channel stride, so conv17 is not a gated layer and
    // everything else reads whole granules
    if (std::string(d.name) == "wconv1" && d.cin == 5) {
      Layer& J = c->wconv1_j4;
      J.def = d; J.w = L.w; J.b = L.b; J.have_w = J.have_b = true;
      const int rc = pack_layer16(c, J, std::vector<int>{0, 1, 2, 4, -1, -1, -1, -1});
      if (rc) return rc;
      if (pack_layer16_d4(c, J, {0, 1, 2, 4})) return 1;
    }
    // 5x5 first layers with at most four real input channels: the pair-of-taps image beside the 8-channel-granule one
    if (d.k == 5 && d.stride == 1 && d.rate == 1 && d.cout == 48 && d.cin >= 3 && d.cin <= 4) {
      const int m4[4] = {0, 1, 2, d.cin == 4 ? 3 : -1};
      if (pack_layer16_d4(c, L, m4)) return 1;
    }
    const int rc = pack_layer16(c, L, identity_map8(d.cin));
    if (rc) return rc;
    // (cin == 192: the image of the FIRST source's 96 channels, for the folded-vector form of conv11)
    if (d.k == 3 && d.stride == 1 && !d.up && (d.cin == 96 || d.cin == 192) && d.cout == 192 && pack_rconv16(c, L)) return 1;
    if (rconv96_eligible(d) && pack_rconv96(c, L)) return 1;
  }
  // 5x5 first layers whose stored input has padding channels (5 of 8, 3 of 4): dense-K image beside the padded one
  if (d.k == 5 && d.stride == 1 && d.rate == 1 && d.cout == 48 && d.cin >= 3 && d.cin <= 5) {
    std::vector<int> id(d.cin);
    for (int i = 0; i < d.cin; ++i) id[i] = i;
    if (pack_layer_dense(c, L, id)) return 1;
  }
  std::vector<int> m;
  if (d.k == 5 && d.cin == 5) { m.assign(8, -1); for (int i = 0; i < 5; ++i) m[i] = i; }   // NHWC8 inputs
  else m = identity_map(d.cin);
  if (std::string(d.name) == "wconv1" && d.cin == 5) {
    // editline_g.py:132-133: with joint_train_inp the style branch sees guide * 0, so its first layer is a 4-channel
    // conv (image*mask, mask) with the guide column of the weights dropped: K = 100 instead of 200 (NHWC8 padding)
    Layer& J = c->wconv1_j4;
    J.def = d; J.w = L.w; J.b = L.b; J.have_w = J.have_b = true;
    const int rc = pack_layer(c, J, std::vector<int>{0, 1, 2, 4});
    if (rc) return rc;
    if (pack_layer_dense(c, J, std::vector<int>{0, 1, 2, 4})) return 1;
  }
  return pack_layer(c, L, m);
}

// ---- narrow layers in raw-tile form (se_rtile.hip): returns 0 and sets *done when the layer was launched there ------
int try_rtile(se_ctx* c, const Layer& L, bool bf, const float* src0, int C0, const float* src1, float* dst, int B, int Hin, int Win,
              int Ho, int Wo, int pad, bool* done) {
  *done = false;
  const LayerDef& d = L.def;
  if (!opt(OPT_RTILE) || src1 || d.stride != 1 || d.rate != 1 || (L.cfg != GC_N48 && L.cfg != GC_N24)) return 0;
  {
    // low-latency mode: only when the tiles still cover the CUs (the small-grid gather-GEMM splits rows over blockIdx.y)
    const int TR0 = rtile_rows(bf);
    const long tiles = (long)B * ((Hin + TR0 - 1) / TR0) * ((Win + 15) / 16) * (d.up ? 4 : 1);
    if (c->low_latency && tiles < opt(OPT_RTILE_LL_MIN)) return 0;
  }
  const int es = bf ? 2 : 4, gran = bf ? 8 : 4;
  const int CG = bf ? L.CGp16 : L.CGp, nch = bf ? L.nch16 : L.nch;
  const float* wimg = bf ? L.d_w16 : L.d_w;
  if (!wimg || C0 != CG * gran) return 0;
  // 24 -> 24 3x3: F(2,3) along x on the raw tile (se_rtilew.hip): 160 instead of 224 MFMAs per wave.  SE_RTILE_WX=0: the direct form
  {
    const int wx_mode = opt(OPT_RTILE_WX);
    if (wx_mode != 0 && !bf && L.d_wx && d.k == 3 && !d.up && C0 == 24 && d.cin == 24 && d.cout == 24 && (Win % 2) == 0 &&
        (long long)B * Hin * Win * 96 < (1ll << 31)) {
      RTileParams p;
      memset(&p, 0, sizeof p);
      p.src = src0; p.wpk = L.d_wx; p.bias = L.d_b; p.dst = dst;
      p.B = B; p.Hin = Hin; p.Win = Win; p.C = C0; p.G = L.G; p.OH = Ho; p.OW = Wo;
      p.ty = (Hin + 15) / 16; p.tx = (Win + 15) / 16;       // blocks of 16 x 16 outputs
      p.act = d.act; p.xcd = xcd_remap_enabled(); p.NP = 32;
      udiv_magic_host((unsigned)(p.ty * p.tx), &p.div_cg_m, &p.div_cg_l);      // (reused fields: blk / (ty * tx), t2 / tx)
      udiv_magic_host((unsigned)p.tx, &p.div_rw_m, &p.div_rw_l);
      const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
      const bool two_d = wx_mode != 1 && L.d_wx2 && (Hin % 2) == 0;       // SE_RTILE_WX=1: the one-dimensional form
      set_launch_cost(alg, 4.0 * ((double)B * Hin * Win * d.cin + (double)B * Ho * Wo * (d.cout / 2)), d.name,
                      (double)B * p.ty * p.tx * 8.0 * (two_d ? 96.0 : 160.0) * 2048.0);       // MFMAs of 16x16x4 per wave and block
      if (two_d) {
        p.wpk = L.d_wx2;
        HIPCHK(c, launch_rtilew2(p, c->st));
        *done = true;
        return 0;
      }
      HIPCHK(c, launch_rtilew(p, c->st));
      *done = true;
      return 0;
    }
  }
  // dense-K form of the 5x5 first layers (fp32): SE_RTILE_DENSE=0 keeps the channel-padded K
  if (opt(OPT_RTILE_DENSE) != 0 && !bf && L.d_wd && L.dense && d.k == 5 && !d.up && L.cfg == GC_N48 && (long long)Hin * Win * C0 * 4 < (1ll << 31)) {
    // The kernel addresses its source with 32-bit byte offsets.  A batch beyond that range is run as sub-launches of the
    // SAME kernel form over image ranges (switching to the channel-padded kernel would change the summation
    // order, i.e. the bit-identity of an image's result across batch compositions within one execution mode).
    const long long per_img = (long long)Hin * Win * C0 * 4;
    const long long lim = addr_limit();       // (SE_TEST_OFFSET_LIMIT, a test aid: a smaller byte range forces the sub-launches)
    const int bmax = (int)std::max<long long>(1, (lim - 1) / per_img);
    for (int b0 = 0; b0 < B; b0 += bmax) {
      const int nb = std::min(bmax, B - b0);
      RTileParams p;
      memset(&p, 0, sizeof p);
      p.src = src0 + (size_t)b0 * Hin * Win * C0; p.wpk = L.d_wd; p.bias = L.d_b; p.dst = dst + (size_t)b0 * Ho * Wo * L.G;
      p.B = nb; p.Hin = Hin; p.Win = Win; p.C = C0; p.G = L.G; p.OH = Ho; p.OW = Wo;
      p.ty = (Hin + 7) / 8; p.tx = (Win + 15) / 16;
      p.act = d.act; p.xcd = xcd_remap_enabled(); p.dense = L.dense; p.nch = L.nchd; p.NP = 48;
      const double alg = 2.0 * (double)nb * Ho * Wo * d.cout * d.cin * 25;
      // F(2,5) along x (rtile_dense5w_kernel; even widths): 6 positions x ceil(5 Cd / 4) k-steps per 2 outputs.  SE_RTILE_D5W=0: direct
      if (opt(OPT_RTILE_D5W) != 0 && L.d_wdw && (Win % 2) == 0) {
        p.wpk = L.d_wdw; p.dense = L.dense + 100;
        set_launch_cost(alg, 4.0 * ((double)nb * Hin * Win * d.cin + (double)nb * Ho * Wo * (d.cout / 2)), d.name,
                        (double)nb * p.ty * p.tx * 4.0 * (6.0 * ((5 * L.dense + 3) / 4) * 3.0) * 2048.0);
        HIPCHK(c, launch_rtile(p, c->st));
        continue;
      }
      set_launch_cost(alg, 4.0 * ((double)nb * Hin * Win * d.cin + (double)nb * Ho * Wo * (d.cout / 2)), d.name,
                      2.0 * (double)nb * Hin * Win * 48.0 * (4.0 * ((25 * L.dense + 3) / 4)));       // ceil(K / 4) k-steps of 4
      HIPCHK(c, launch_rtile(p, c->st));
    }
    *done = true;
    return 0;
  }
  // bf16: pair-of-taps form of the 5x5 first layers whose stored NHWC8 input has at most four real channels (SE_RTILE_DENSE=0:
  // the 8-channel-granule K)
  if (opt(OPT_RTILE_DENSE) != 0 && bf && L.d_w16d && d.k == 5 && !d.up && L.cfg == GC_N48 && C0 == 8 &&
      (long long)B * Hin * Win * 16 < (1ll << 31)) {
    RTileParams p;
    memset(&p, 0, sizeof p);
    p.src = src0; p.wpk = L.d_w16d; p.bias = L.d_b; p.dst = dst;
    p.B = B; p.Hin = Hin; p.Win = Win; p.C = C0; p.CG = 1; p.T = 25; p.KW = 5;
    p.pad = pad; p.OH = Ho; p.OW = Wo; p.G = (L.G + 7) & ~7;
    p.nch = 2; p.NP = 48; p.RH = 32 + 4; p.RW = 21;                     // one column more than the halo: the sixth tap of a pair row
    p.raw_bytes = (p.RH * p.RW * 8 + 1023) & ~1023;
    udiv_magic_host((unsigned)p.RW, &p.div_rw_m, &p.div_rw_l);
    udiv_magic_host(1u, &p.div_cg_m, &p.div_cg_l);
    p.ty = (Hin + 31) / 32; p.tx = (Win + 15) / 16;
    p.act = d.act; p.bf16 = 1; p.xcd = xcd_remap_enabled(); p.dense = 204;
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 25;
    set_launch_cost(alg, 2.0 * ((double)B * Hin * Win * d.cin + (double)B * Ho * Wo * (d.cout / 2)), d.name,
                    2.0 * (double)B * Hin * Win * 48.0 * 128.0);
    HIPCHK(c, launch_rtile(p, c->st));
    *done = true;
    return 0;
  }
  const int KW = d.up ? 2 : d.k, KH = KW;
  RTileParams p;
  memset(&p, 0, sizeof p);
  const int TR = rtile_rows(bf);
  p.RH = TR + KH - 1; p.RW = 16 + KW - 1;
  p.raw_bytes = (p.RH * p.RW * C0 * es + 1023) & ~1023;
  if (p.raw_bytes + nch * L.NP * 128 > 80 * 1024) return 0;                 // two workgroups per CU or not at all
  if ((long long)B * Hin * Win * C0 * es >= (1ll << 31)) return 0;
  p.src = src0; p.wpk = wimg; p.bias = L.d_b; p.dst = dst;
  p.B = B; p.Hin = Hin; p.Win = Win; p.C = C0; p.CG = CG; p.T = L.T; p.KW = KW;
  p.magicCG = (65536 + CG - 1) / CG; p.magicKW = 256 / KW + 1;
  for (int gi = 0; gi < nch * 8 + 8; ++gi)
    if (((gi * p.magicCG) >> 16) != gi / CG) return 0;
  for (int t = 0; t <= L.T + 8; ++t)
    if (((t * p.magicKW) >> 8) != t / KW) return 0;
  p.pad = pad; p.up2 = d.up ? 1 : 0; p.OH = Ho; p.OW = Wo;
  p.G = bf ? (L.G + 7) & ~7 : L.G;
  p.nch = nch; p.NP = L.NP;
  udiv_magic_host((unsigned)(C0 * es / 16), &p.div_cg_m, &p.div_cg_l);
  udiv_magic_host((unsigned)p.RW, &p.div_rw_m, &p.div_rw_l);
  p.ty = (Hin + TR - 1) / TR; p.tx = (Win + 15) / 16;
  p.act = d.act; p.bf16 = bf ? 1 : 0; p.xcd = xcd_remap_enabled();
  {
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * d.k * d.k;
    const double exec = 2.0 * (double)B * Hin * Win * (d.up ? 4.0 : 1.0) * L.NP * (nch * (bf ? 64.0 : 32.0));
    set_launch_cost(alg, (double)es * ((double)B * Hin * Win * d.cin + (double)B * Ho * Wo * (d.cout / 2)), d.name, exec);
  }
  HIPCHK(c, launch_rtile(p, c->st));
  *done = true;
  return 0;
}

// ---- bf16 mode: every gated conv is the direct gather-GEMM on v_mfma_f32_16x16x32_bf16 (se_gconv.hip, BF16) -----------
// (The Winograd transforms would have to run in fp32 on bf16 data and round the transformed tiles again; at 16x the
// MFMA rate the layers are bound by the LDS fill, not by multiply-adds, so there is nothing for them to buy.)
// two polyphase sub-images of 8 columns (and at most 8 rows) share one 8 x 16 raw tile (rconv16b_kernel, p.dual)
static bool rconv16_dual_ok(const Layer& L, const LayerDef& d, int Hin, int Win) {
  if (opt(OPT_RCONV16_DUAL) == 0) return false;
  return rconv16_small_tiles() && L.d_w16s && (d.rate % 2) == 0 && Win / d.rate == 8 && Hin / d.rate <= 8 && Hin / d.rate >= 4;
}

int run_gconv16(se_ctx* c, const Layer& L, const float* src0, int C0, const float* src1, int C1, int src1_vec, float* dst,
                int B, int Hin, int Win, int Ho, int Wo, int pad) {
  const LayerDef& d = L.def;
  if (!L.d_w16) return fail(c, "layer %s: no bf16 weight image", d.name);
  if ((C0 + C1) != L.CGp16 * 8) return fail(c, "layer %s: bf16 source channels %d+%d != packed %d", d.name, C0, C1, L.CGp16 * 8);
  // the dominant shape (96 -> 192, 3x3, stride 1) runs in the raw-tile form when its polyphase sub-images are large
  // enough to fill 16 x 16 tiles reasonably (se_rconv16.hip); SE_RCONV16=0 keeps it on the gather-GEMM
  const bool use_rconv = opt(OPT_RCONV16) != 0;
  // conv11 of netG: the spatially constant second source folded into a bias table (run_gconv / launch_vecbias), the layer on
  // the 8 x 16 raw-tile kernel with the first source alone
  {
    if (opt(OPT_VECBIAS) != 0 && use_rconv && !c->low_latency && src1 && src1_vec && c->vbias_ws && c->vec32 && L.d_w16s && L.d_wv16 &&
        rconv16_small_tiles() && d.k == 3 && d.stride == 1 && d.rate == 1 && !d.up && d.cin == 192 && d.cout == 192 && C0 == 96 && C1 == 96 &&
        Hin >= 12 && Win >= 12 && (long long)B * Hin * Win * 192 < (1ll << 31)) {
      HIPCHK(c, launch_vecbias(L.d_wv16, c->vec32, c->vbias_ws, B, 96, c->st, 1));
      RConvParams rp;
      memset(&rp, 0, sizeof rp);
      rp.src = src0; rp.wpk = L.d_w16s; rp.bias = L.d_b; rp.dst = dst; rp.vbias = c->vbias_ws;
      rp.B = B; rp.h = Hin; rp.w = Win; rp.d = 1; rp.hs = Hin; rp.ws = Win;
      rp.ty = (Hin + 7) / 8; rp.tx = (Win + 15) / 16;
      rp.act = d.act; rp.xcd = xcd_remap_enabled();
      const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
      set_launch_cost(alg, 2.0 * 2.0 * (double)B * Hin * Win * 96, d.name, 2.0 * (double)B * Hin * Win * 192.0 * 864.0);
      HIPCHK(c, launch_rconv16(rp, c->st));
      return 0;
    }
  }
  if (use_rconv && !c->low_latency && d.k == 3 && d.stride == 1 && !d.up && d.cin == 96 && d.cout == 192 && !src1 && C0 == 96 &&
      (Hin % d.rate) == 0 && (Win % d.rate) == 0 && L.nch16 == 14 && (long long)B * Hin * Win * 192 < (1ll << 31) &&
      ((Hin / d.rate >= 12 && Win / d.rate >= 12) || rconv16_dual_ok(L, d, Hin, Win))) {
    RConvParams rp;
    memset(&rp, 0, sizeof rp);
    const bool small = rconv16_small_tiles() && L.d_w16s;     // 8 x 16 tiles, 32-k step image, no K padding
    // sub-images of 8 x 8 (dilation 16 at 128 x 128, 8 at 64 x 64): two phases share an 8 x 16 tile (SE_RCONV16_DUAL=0: gather-GEMM)
    rp.dual = !(Hin / d.rate >= 12 && Win / d.rate >= 12);
    rp.src = src0; rp.wpk = small ? L.d_w16s : L.d_w16; rp.bias = L.d_b; rp.dst = dst;
    rp.B = B; rp.h = Hin; rp.w = Win; rp.d = d.rate; rp.hs = Hin / d.rate; rp.ws = Win / d.rate;
    rp.ty = small ? (rp.hs + 7) / 8 : (rp.hs + 15) / 16; rp.tx = (rp.ws + 15) / 16;
    rp.act = d.act; rp.xcd = xcd_remap_enabled();
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    set_launch_cost(alg, 2.0 * 2.0 * (double)B * Hin * Win * 96, d.name, 2.0 * (double)B * Hin * Win * 192.0 * (small ? 864.0 : 896.0));
    HIPCHK(c, launch_rconv16(rp, c->st));
    return 0;
  }
  // 96-row stride-1 layers (3x3 48/24 -> 96, gen_deconv 96 -> 96): raw-tile form, se_rconv96.hip (SE_RCONV96=0: gather-GEMM)
  const bool use_rconv96 = opt(OPT_RCONV96) != 0;
  if (use_rconv96 && !c->low_latency && L.d_w96 && !src1 && C0 == d.cin && Hin >= 12 && Win >= 12 &&
      (long long)B * Ho * Wo * 96 < (1ll << 31) && (long long)B * Hin * Win * C0 * 2 < (1ll << 31)) {
    RConv96Params rp;
    memset(&rp, 0, sizeof rp);
    rp.src = src0; rp.wpk = L.d_w96; rp.bias = L.d_b; rp.dst = dst;
    rp.B = B; rp.h = Hin; rp.w = Win; rp.CG = C0 / 8;
    rp.stride = d.stride; rp.oh = Ho; rp.ow = Wo;
    rp.ty = ((d.stride == 2 ? Ho : Hin) + 15) / 16; rp.tx = ((d.stride == 2 ? Wo : Win) + 15) / 16;
    rp.up2 = d.up ? 1 : 0; rp.act = d.act; rp.xcd = xcd_remap_enabled();
    const int T = d.up ? 4 : 9, nstep = (T * rp.CG + 3) / 4;
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    set_launch_cost(alg, 2.0 * ((double)B * Hin * Win * d.cin + (double)B * Ho * Wo * 48), d.name,
                    2.0 * (double)B * (d.stride == 2 ? (double)Ho * Wo : (double)Hin * Win) * (d.up ? 4.0 : 1.0) * 96.0 * (nstep * 32.0));
    HIPCHK(c, launch_rconv96(rp, c->st));
    return 0;
  }
  {
    bool done = false;
    if (try_rtile(c, L, true, src0, C0, src1, dst, B, Hin, Win, Ho, Wo, pad, &done)) return 1;
    if (done) return 0;
  }
  GConvParams p;
  memset(&p, 0, sizeof p);
  p.bf16 = 1;
  p.src0 = src0; p.src1 = src1 ? src1 : src0; p.wpk = L.d_w16; p.bias = L.d_b; p.dst = dst; p.zeros = c->zeros;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Ho = d.up ? Hin : Ho; p.Wo = d.up ? Win : Wo;
  p.up2 = d.up ? 1 : 0; p.OH = Ho; p.OW = Wo;
  p.C0 = C0; p.C1 = C1 ? C1 : C0; p.C0g = C0 / 8; p.CG = L.CGp16;
  const int KW = d.up ? 2 : d.k;
  p.T = L.T; p.KW = KW; p.stride = d.stride; p.dil = d.rate; p.pad = pad;
  p.magicCG = (65536 + L.CGp16 - 1) / L.CGp16; p.magicKW = 256 / KW + 1; p.magicKH = L.T / KW;
  for (int gi = 0; gi < L.nch16 * 8 + 8; ++gi)
    if (((gi * p.magicCG) >> 16) != gi / L.CGp16) return fail(c, "layer %s: magic division check failed", d.name);
  for (int t = 0; t <= L.T + 8; ++t)
    if (((t * p.magicKW) >> 8) != t / KW) return fail(c, "layer %s: magic tap division check failed", d.name);
  const int Gs = (L.G + 7) & ~7;                 // stored channel stride of the output
  // 32-bit byte offsets (sentinel 0x80000000): source and destination below 2^31 BYTES
  if ((double)B * Hin * Win * (C0 > C1 ? C0 : C1) * 2.0 >= 2147483648.0 || (double)B * Ho * Wo * Gs * 2.0 >= 2147483648.0)
    return fail(c, "layer %s: a tensor of this launch exceeds 2^31 bytes (32-bit byte offsets) -- split the batch", d.name);
  for (int j = 0; j < p.magicKH; ++j) p.rep |= 1u << (j * KW);
  udiv_magic_host((unsigned)(p.Ho * p.Wo), &p.div_hw_m, &p.div_hw_l);
  udiv_magic_host((unsigned)p.Wo, &p.div_w_m, &p.div_w_l);
  p.Hlim = Hin; p.Wlim = Win;
  p.src1_vec = src1_vec; p.nch = L.nch16; p.G = Gs; p.act = d.act; p.total_pix = B * p.Ho * p.Wo;
  p.xcd = xcd_remap_enabled();
  p.np_full = L.NP; p.nf_full = L.NP / 32; p.small_grid = c->low_latency ? 1 : 0;
  {
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * d.k * d.k;
    const double exec = 2.0 * (double)B * p.Ho * p.Wo * (d.up ? 4.0 : 1.0) * L.NP * (L.nch16 * 64.0);
    set_launch_cost(alg, 2.0 * ((double)B * Hin * Win * d.cin + (double)B * Ho * Wo * (d.cout / 2)), d.name, exec);
  }
  HIPCHK(c, launch_gconv(L.cfg, p, c->st));
  return 0;
}

// ---- launching one gated conv -----------------------------------------------------------------------
int run_gconv(se_ctx* c, const Layer& L, const float* src0, int C0, const float* src1, int C1, int src1_vec, float* dst,
              int B, int Hin, int Win, int* Ho_, int* Wo_) {
  const LayerDef& d = L.def;
  const int pad = d.rate * (d.k - 1) / 2;                               // utils.py:20
  int Ho, Wo;
  if (d.up) { Ho = 2 * Hin; Wo = 2 * Win; }
  else { Ho = (Hin + 2 * pad - d.rate * (d.k - 1) - 1) / d.stride + 1; Wo = (Win + 2 * pad - d.rate * (d.k - 1) - 1) / d.stride + 1; }
  if (Ho_) *Ho_ = Ho;
  if (Wo_) *Wo_ = Wo;
  if (c->dry) return 0;
  if (!L.packed) return fail(c, "layer %s: weights not loaded", d.name);
  if (c->bf16) return run_gconv16(c, L, src0, C0, src1, C1, src1_vec, dst, B, Hin, Win, Ho, Wo, pad);
  if ((C0 + C1) != L.CGp * 4) return fail(c, "layer %s: source channels %d+%d != packed %d", d.name, C0, C1, L.CGp * 4);
  // Low-latency mode (SE_FLAG_LOW_LATENCY, one or two images): the Winograd kernels' 64/128-tile workgroups would
  // occupy 16-32 of the 256 CUs, so every gated conv takes the direct kernel in its small-grid shape instead
  // (launch_gconv: 64-pixel tiles, rows split over blockIdx.y).  2.25x more multiply-adds on ~10x more CUs.
  // ... unless the layer's Winograd grid still covers enough of the chip (round 6; 0: never).  Measured (tools/ll_wino_sweep.sh):
  // the 96 -> 192 kernels pay from 64 workgroups on (a third of the multiply-adds outweighs a quarter of the CUs: one 512x512
  // image 3.22 -> 2.83 ms), the 96- and 48-row kernels only from 128 on (at 64 workgroups they cost one 256x256 image
  // 1.20 -> 1.31 ms) -- SE_LL_WINO_MIN_WG / SE_LL_WINO48_MIN_WG.  The grid is counted PER IMAGE (the call's batch size does not
  // enter): which kernel a layer runs must depend on the image size and the mode only, so that an image's result stays
  // bit-identical across batch sizes, batch positions and ranks within the mode.
  const int ll_min_wg = opt(OPT_LL_WINO_MIN_WG), ll_min_wg48 = opt(OPT_LL_WINO48_MIN_WG);
  auto wino_grid_ok = [&](long wgs, bool n192 = false) {
    const int m = n192 ? ll_min_wg : ll_min_wg48;
    return !c->low_latency || (m > 0 && wgs >= m);
  };
  const bool use_wino = opt(OPT_WINOGRAD) != 0;
  const bool wino_src_ok = (!src1 && C0 == 96 && d.cin == 96) || (src1 && C0 == 96 && C1 == 96 && d.cin == 192);
  // the kernel addresses a source through 32-bit byte offsets (96 floats per pixel)
  const bool wino_addr_ok = (long long)B * Hin * Win * 384 < (1ll << 31);
  if (use_wino && !d.up && L.d_u && wino_src_ok && !wino_addr_ok) {
    static bool said = false;
    if (!said) {
      said = true;
      fprintf(stderr, "sketchedit_hip: layer %s: %d x %dx%d x 96 floats exceed the Winograd kernel's 32-bit byte offsets; "
                      "using the direct kernel (about 2x slower) -- split the batch\n", d.name, B, Hin, Win);
    }
  }
  if (use_wino && !d.up && L.d_u && wino_src_ok && wino_addr_ok && (Hin % (2 * d.rate)) == 0 && (Win % (2 * d.rate)) == 0 &&
      wino_grid_ok(((long)(Hin / 2) * (Win / 2) + 63) / 64, true)) {
    WinoParams wp;
    memset(&wp, 0, sizeof wp);
    wp.src = src0; wp.src1 = src1; wp.src1_vec = src1_vec;
    wp.upk = L.d_u; wp.bias = L.d_b; wp.dst = dst;
    // A spatially constant second source is folded into a per-image, per-border-configuration bias (launch_vecbias) and
    // the layer runs as the SINGLE-source kernel: half the positions' K, 311 -> 170 us at 256x256 B=32 (SE_VECBIAS=0: the
    // two-source kernel reads the vector as a second input)
    const bool vecbias_on = opt(OPT_VECBIAS) != 0;
    bool folded = false;
    if (vecbias_on && src1 && src1_vec && d.rate == 1 && L.d_u1 && L.d_wv && c->vbias_ws && Hin >= 2 && Win >= 2) {
      HIPCHK(c, launch_vecbias(L.d_wv, src1, c->vbias_ws, B, 96, c->st));
      wp.src1 = nullptr; wp.src1_vec = 0; wp.upk = L.d_u1; wp.vbias = c->vbias_ws;
      folded = true;
    }
    // Hybrid F(2,3) x F(4,3) (se_wino24.hip) for the single-source form where the width allows 4-column tiles: 24 instead of
    // 32 positions per 8 outputs.  SE_WINOGRAD_F43=0: F(2x2,3x3) everywhere; 1 (default): the hybrid kernel everywhere; 2: netG
    // only (netM's soft mask feeds the 0.5 threshold, editline2_model.py:346-347).  Measured in round 5 over 72 images / 4.7 M
    // mask pixels of the three weight sets (tools/f43_flips.py, DESIGN.md 3.1b'): hard-mask flips against the oracle 1 (mode 0),
    // 4 (mode 1), 1 (mode 2) -- under one pixel per ten images either way -- for 11.10 / 11.29 ms per step (modes 1 / 2): by the
    // rule VERDICT r4 set (switch only if it costs < 1 %) the default stays 1.  (A mode that kept F(2x2,3x3) for netM's mask
    // decoder alone was measured too: 4 flips, no better than mode 1 -- the error comes from the encoder; removed.)
    const int f43_mode = opt(OPT_WINOGRAD_F43);
    // SE_FLAG_CONSERVATIVE is mode 2 for this call, chosen by the caller through the ABI instead of the process-wide table
    const bool f43_net_ok = c->cur_net == SE_NET_G || (f43_mode == 1 && !c->conservative);
    const bool f43 = (f43_mode == 1 || f43_mode == 2) && f43_net_ok && (wp.src1 ? (!wp.src1_vec && L.d_u24b) : L.d_u24 != nullptr) && L.d_ub24 && (Win % (4 * d.rate)) == 0;
    wp.B = B; wp.h = Hin; wp.w = Win; wp.d = d.rate; wp.th = Hin / 2; wp.tw = f43 ? Win / 4 : Win / 2;
    wp.total_tiles = B * wp.th * wp.tw; wp.act = d.act;
    wp.xcd = xcd_remap_enabled();
    udiv_magic_host((unsigned)(wp.th * wp.tw), &wp.div_tpi_m, &wp.div_tpi_l);
    udiv_magic_host((unsigned)wp.tw, &wp.div_tw_m, &wp.div_tw_l);
    udiv_magic_host((unsigned)wp.d, &wp.div_d_m, &wp.div_d_l);
    if ((double)B * Hin * Win * 384.0 >= 2147483648.0) return fail(c, "layer %s: tensor exceeds 2^31 bytes", d.name);
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    if (f43) {
      wp.upk = wp.src1 ? L.d_u24b : L.d_u24; wp.bias = L.d_ub24;
      // 24 of 72 products per 2x4 outputs; with the vector source folded away only the first source's half of K is executed
      set_launch_cost(alg, 4.0 * 2.0 * (double)B * Hin * Win * 96, d.name, alg * 24.0 / 72.0 * (folded ? 0.5 : 1.0));
      HIPCHK(c, launch_wino24(wp, c->st));
      return 0;
    }
    // F(2x2,3x3): 16 of 36 products; with the vector source folded away only the first source's half of K is executed
    set_launch_cost(alg, 4.0 * 2.0 * (double)B * Hin * Win * 96, d.name, alg * 16.0 / 36.0 * (folded ? 0.5 : 1.0));
    HIPCHK(c, launch_wino(wp, c->st));
    return 0;
  }
  // 48 -> 192 (xconv5 of netG) on the hybrid kernel's 48-channel instantiation: 2 chunks per position, the second half empty
  {
    if (use_wino && opt(OPT_WINOGRAD_F43) != 0 && !d.up && !src1 && C0 == 48 && d.cin == 48 && d.cout == 192 && d.stride == 1 && L.d_u24 && L.d_ub24 &&
        (long long)B * Hin * Win * 384 < (1ll << 31) && (Hin % (2 * d.rate)) == 0 && (Win % (4 * d.rate)) == 0 &&
        wino_grid_ok(((long)(Hin / 2) * (Win / 4) + 31) / 32, true)) {
      WinoParams wp;
      memset(&wp, 0, sizeof wp);
      wp.src = src0; wp.upk = L.d_u24; wp.bias = L.d_ub24; wp.dst = dst;
      wp.B = B; wp.h = Hin; wp.w = Win; wp.d = d.rate; wp.th = Hin / 2; wp.tw = Win / 4;
      wp.total_tiles = B * wp.th * wp.tw; wp.act = d.act;
      wp.xcd = xcd_remap_enabled();
      udiv_magic_host((unsigned)(wp.th * wp.tw), &wp.div_tpi_m, &wp.div_tpi_l);
      udiv_magic_host((unsigned)wp.tw, &wp.div_tw_m, &wp.div_tw_l);
      udiv_magic_host((unsigned)wp.d, &wp.div_d_m, &wp.div_d_l);
      const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
      set_launch_cost(alg, 4.0 * ((double)B * Hin * Win * 48 + (double)B * Hin * Win * 96), d.name, alg * 24.0 / 72.0);
      HIPCHK(c, launch_wino24_c48(wp, c->st));
      return 0;
    }
  }
  const bool use_wino48 = opt(OPT_WINOGRAD48) != 0;
  if (use_wino && use_wino48 && !d.up && L.d_u && L.d_ub && !src1 && C0 == 48 && d.cin == 48 && (long long)B * Hin * Win * 192 < (1ll << 31) &&
      (Hin % (2 * d.rate)) == 0 && (Win % (2 * d.rate)) == 0 && wino_grid_ok(((long)(Hin / 2) * (Win / 2) + 63) / 64)) {
    WinoParams wp;
    memset(&wp, 0, sizeof wp);
    wp.src = src0; wp.upk = L.d_u; wp.bias = L.d_ub; wp.dst = dst;
    wp.B = B; wp.h = Hin; wp.w = Win; wp.d = d.rate; wp.th = Hin / 2; wp.tw = Win / 2;
    wp.total_tiles = B * wp.th * wp.tw; wp.act = d.act;
    wp.xcd = xcd_remap_enabled();
    udiv_magic_host((unsigned)(wp.th * wp.tw), &wp.div_tpi_m, &wp.div_tpi_l);
    udiv_magic_host((unsigned)wp.tw, &wp.div_tw_m, &wp.div_tw_l);
    udiv_magic_host((unsigned)wp.d, &wp.div_d_m, &wp.div_d_l);
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    set_launch_cost(alg, 4.0 * 2.0 * (double)B * Hin * Win * 48, d.name, alg * 16.0 / 36.0);
    HIPCHK(c, launch_wino48(wp, c->st));
    return 0;
  }
  // 24 -> 96 (xconv3 / pmconv3 of netG) on the same kernel, one chunk per position (se_wino48.hip CIN = 24): 16 of 36 products
  if (use_wino && use_wino48 && !d.up && L.d_u && L.d_ub && !src1 && C0 == 24 && d.cin == 24 && d.cout == 96 && d.stride == 1 &&
      (long long)B * Hin * Win * 192 < (1ll << 31) && (Hin % (2 * d.rate)) == 0 && (Win % (2 * d.rate)) == 0 &&
      wino_grid_ok(((long)(Hin / 2) * (Win / 2) + 63) / 64)) {
    WinoParams wp;
    memset(&wp, 0, sizeof wp);
    wp.src = src0; wp.upk = L.d_u; wp.bias = L.d_ub; wp.dst = dst;
    wp.B = B; wp.h = Hin; wp.w = Win; wp.d = d.rate; wp.th = Hin / 2; wp.tw = Win / 2;
    wp.total_tiles = B * wp.th * wp.tw; wp.act = d.act;
    wp.xcd = xcd_remap_enabled();
    udiv_magic_host((unsigned)(wp.th * wp.tw), &wp.div_tpi_m, &wp.div_tpi_l);
    udiv_magic_host((unsigned)wp.tw, &wp.div_tw_m, &wp.div_tw_l);
    udiv_magic_host((unsigned)wp.d, &wp.div_d_m, &wp.div_d_l);
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    set_launch_cost(alg, 4.0 * ((double)B * Hin * Win * 24 + (double)B * Hin * Win * 48), d.name, alg * 16.0 / 36.0);
    HIPCHK(c, launch_wino48_c24(wp, c->st));
    return 0;
  }
  const bool use_winoup = opt(OPT_WINOGRAD_UP) != 0;
  if (use_wino && use_winoup && d.up && L.d_u && L.d_ub && !src1 && C0 == 96 && d.cin == 96 &&
      (long long)B * Hin * Win * 384 < (1ll << 31) && (Hin % 2) == 0 && (Win % 2) == 0 &&
      wino_grid_ok(4 * (((long)(Hin / 2) * (Win / 2) + 63) / 64))) {
    WinoParams wp;
    memset(&wp, 0, sizeof wp);
    wp.src = src0; wp.upk = L.d_u; wp.bias = L.d_ub; wp.dst = dst;
    wp.B = B; wp.h = Hin; wp.w = Win; wp.d = 1; wp.th = Hin / 2; wp.tw = Win / 2;
    wp.total_tiles = B * wp.th * wp.tw; wp.act = d.act;
    wp.xcd = xcd_remap_enabled();
    udiv_magic_host((unsigned)(wp.th * wp.tw), &wp.div_tpi_m, &wp.div_tpi_l);
    udiv_magic_host((unsigned)wp.tw, &wp.div_tw_m, &wp.div_tw_l);
    udiv_magic_host(1u, &wp.div_d_m, &wp.div_d_l);
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    set_launch_cost(alg, 4.0 * ((double)B * Hin * Win * 96 + (double)B * Ho * Wo * 48), d.name,
                    alg * 9.0 / 36.0);                       // F(2x2,2x2) on the sub-pixel classes: 9 of 36 products
    HIPCHK(c, launch_winoup(wp, c->st));
    return 0;
  }
  // gen_deconv 48 -> 48: F(2x2,2x2) on the sub-pixel classes with the K pairing of the 48-channel kernels (se_wino_up48.hip)
  const bool use_winoup48 = opt(OPT_WINOGRAD_UP48) != 0;
  if (use_wino && use_winoup && use_winoup48 && d.up && L.d_u && L.d_ub && !src1 && C0 == 48 && d.cin == 48 && d.cout == 48 &&
      (long long)B * Hin * Win * 192 < (1ll << 31) && (long long)B * Ho * Wo * 96 < (1ll << 31) && (Hin % 2) == 0 && (Win % 2) == 0 &&
      wino_grid_ok(4 * (((long)(Hin / 2) * (Win / 2) + 63) / 64))) {
    WinoParams wp;
    memset(&wp, 0, sizeof wp);
    wp.src = src0; wp.upk = L.d_u; wp.bias = L.d_ub; wp.dst = dst;
    wp.B = B; wp.h = Hin; wp.w = Win; wp.d = 1; wp.th = Hin / 2; wp.tw = Win / 2;
    wp.total_tiles = B * wp.th * wp.tw; wp.act = d.act;
    wp.xcd = xcd_remap_enabled();
    udiv_magic_host((unsigned)(wp.th * wp.tw), &wp.div_tpi_m, &wp.div_tpi_l);
    udiv_magic_host((unsigned)wp.tw, &wp.div_tw_m, &wp.div_tw_l);
    udiv_magic_host(1u, &wp.div_d_m, &wp.div_d_l);
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * 9;
    set_launch_cost(alg, 4.0 * ((double)B * Hin * Win * 48 + (double)B * Ho * Wo * 24), d.name,
                    alg * 9.0 / 36.0 * 28.0 / 27.0);         // 9 of 36 products, 28 k-halves executed for 27 of work
    HIPCHK(c, launch_winoup48(wp, c->st));
    return 0;
  }
  {
    bool done = false;
    if (try_rtile(c, L, false, src0, C0, src1, dst, B, Hin, Win, Ho, Wo, pad, &done)) return 1;
    if (done) return 0;
  }
  GConvParams p;
  memset(&p, 0, sizeof p);
  p.src0 = src0; p.src1 = src1 ? src1 : src0; p.wpk = L.d_w; p.bias = L.d_b; p.dst = dst; p.zeros = c->zeros;
  p.B = B; p.Hin = Hin; p.Win = Win; p.Ho = d.up ? Hin : Ho; p.Wo = d.up ? Win : Wo;   // up2: rows walk the source grid
  p.up2 = d.up ? 1 : 0; p.OH = Ho; p.OW = Wo;
  p.C0 = C0; p.C1 = C1 ? C1 : C0; p.C0g = C0 / 4; p.CG = L.CGp;
  const int KW = d.up ? 2 : d.k;
  p.T = L.T; p.KW = KW; p.stride = d.stride; p.dil = d.rate; p.pad = pad;
  p.magicCG = (65536 + L.CGp - 1) / L.CGp; p.magicKW = 256 / KW + 1; p.magicKH = L.T / KW;
  for (int gi = 0; gi < L.nch * 8 + 8; ++gi)
    if (((gi * p.magicCG) >> 16) != gi / L.CGp) return fail(c, "layer %s: magic division check failed", d.name);
  for (int t = 0; t <= L.T + 8; ++t)
    if (((t * p.magicKW) >> 8) != t / KW) return fail(c, "layer %s: magic tap division check failed", d.name);
  // 32-bit byte offsets (sentinel 0x80000000): source and destination below 2^31 BYTES
  if ((double)B * Hin * Win * (C0 > C1 ? C0 : C1) * 4.0 >= 2147483648.0 || (double)B * Ho * Wo * L.G * 4.0 >= 2147483648.0)
    return fail(c, "layer %s: a tensor of this launch exceeds 2^31 bytes (32-bit byte offsets) -- split the batch", d.name);
  p.ushift = 0;
  p.rep = 0;
  for (int j = 0; j < p.magicKH; ++j) p.rep |= 1u << (j * KW);
  udiv_magic_host((unsigned)(p.Ho * p.Wo), &p.div_hw_m, &p.div_hw_l);
  udiv_magic_host((unsigned)p.Wo, &p.div_w_m, &p.div_w_l);
  p.Hlim = Hin; p.Wlim = Win;
  p.src1_vec = src1_vec; p.nch = L.nch; p.G = L.G; p.act = d.act; p.total_pix = B * p.Ho * p.Wo;
  p.xcd = xcd_remap_enabled();
  p.np_full = L.NP; p.nf_full = L.NP / 32; p.small_grid = c->low_latency ? 1 : 0;
  // algorithmic cost as the reference defines the layer (3x3 on the upsampled grid for gen_deconv)
  {
    const double alg = 2.0 * (double)B * Ho * Wo * d.cout * d.cin * d.k * d.k;
    // executed: the packed K (channels padded to granules, taps x channels padded to 32) and row (NP) sizes the MFMAs
    // really run over; the sub-pixel form of gen_deconv runs 4 classes x 4 taps instead of 9 taps on the upsampled grid
    const double exec = 2.0 * (double)B * p.Ho * p.Wo * (d.up ? 4.0 : 1.0) * L.NP * (L.nch * 32.0);
    set_launch_cost(alg, 4.0 * ((double)B * Hin * Win * d.cin + (double)B * Ho * Wo * (d.cout / 2)), d.name, exec);
  }
  HIPCHK(c, launch_gconv(L.cfg, p, c->st));
  return 0;
}

struct Act {   // an NHWC activation living in the arena
  float* p = nullptr;
  int H = 0, W = 0, C = 0;
};

struct Plan {
  se_ctx* c;
  std::map<std::string, Layer>& net;
  int B;
  int rc = 0;
  Arena* ar;                // arena of the branch being planned
  Plan(se_ctx* c_, std::map<std::string, Layer>& n, int B_) : c(c_), net(n), B(B_), ar(&c_->arena) {}
  float* alloc_raw(size_t nfloats) {
    float* p = ar->alloc(nfloats);
    if (!p && !rc) rc = fail(c, "workspace too small (need more than %zu bytes)", ar->cap);
    return p;
  }
  Act alloc(int H, int W, int C) {
    Act a; a.H = H; a.W = W;
    if (c->bf16) {       // bf16 activations: whole 8-channel granules per pixel (12 -> 16, 4 / 5 -> 8), two per float slot
      a.C = (C + 7) & ~7;
      a.p = alloc_raw(((size_t)B * H * W * a.C + 1) / 2);
    } else {
      a.C = C;
      a.p = alloc_raw((size_t)B * H * W * C);
    }
    if (c->dry) c->dry_max_act = std::max(c->dry_max_act, (size_t)B * H * W * a.C * (c->bf16 ? 2 : 4));
    return a;
  }
  void release(const float* p) {          // to whichever arena the block came from
    if (!p) return;
    if (c->arena2.owns(p)) c->arena2.release(p); else c->arena.release(p);
  }
  void free(Act& a) { release(a.p); a.p = nullptr; }
  // ---- concurrent branches (low-latency mode).  side_begin() .. side_end() plans a branch on the side stream and in the
  // side arena; it starts after everything enqueued on the main stream so far (fork event) and join() makes the main
  // stream wait for it.  A buffer shared by both branches must be released only after join().  Outside low-latency mode
  // the calls do nothing: the branch is planned in line, on the main stream and arena.
  // Default mode (round 6): the same two-stream plan (SE_FORK_DEFAULT=0: one stream) -- the tail of one branch's kernel runs
  // under the head of the other's: +2 % on configs 2, 3 and 5.  While the in-library profiler is on, the default mode is
  // planned on ONE stream (c->serial): an event pair around a launch measures that kernel only when nothing else shares the chip.
  // (c->fork2 = SE_FORK_DEFAULT as it was when the call entered the library: one snapshot per call, so a switch flipped by
  // another thread cannot leave a plan with a fork and no join)
  bool forked() const { return (c->low_latency || (c->fork2 && !c->serial)) && c->st_side; }
  int side_begin() {
    if (!forked()) return 0;
    ar = &c->arena2;
    if (c->dry) return 0;
    HIPCHK(c, hipEventRecord(c->ev_fork, c->st_main));
    HIPCHK(c, hipStreamWaitEvent(c->st_side, c->ev_fork, 0));
    c->st = c->st_side;
    return 0;
  }
  int side_end() {
    if (!forked()) return 0;
    ar = &c->arena;
    if (c->dry) return 0;
    HIPCHK(c, hipEventRecord(c->ev_join, c->st_side));
    c->st = c->st_main;
    return 0;
  }
  int join() {
    if (!forked() || c->dry) return 0;
    HIPCHK(c, hipStreamWaitEvent(c->st_main, c->ev_join, 0));
    return 0;
  }
  // gated conv layer: consumes (and releases, if `rel`) `in`
  Act conv(const char* name, Act& in, bool rel = true, const float* src1 = nullptr, int C1 = 0, int vec = 0) {
    if (rc) return Act();
    return conv_layer(net.at(name), in, rel, src1, C1, vec);
  }
  Act conv_layer(Layer& L, Act& in, bool rel = true, const float* src1 = nullptr, int C1 = 0, int vec = 0) {
    Act out;
    if (rc) return out;
    int Ho = 0, Wo = 0;
    // dry pass to get shape
    bool dry = c->dry; c->dry = true;
    run_gconv(c, L, nullptr, 0, nullptr, 0, 0, nullptr, B, in.H, in.W, &Ho, &Wo);
    c->dry = dry;
    out = alloc(Ho, Wo, L.def.cout / 2);
    if (rc) return out;
    if (!c->dry) rc = run_gconv(c, L, in.p, in.C, src1, C1, vec, out.p, B, in.H, in.W, nullptr, nullptr);
    if (rel) free(in);
    return out;
  }
};

// encoder: conv1 (5x5) .. conv10_atrous; returns conv10 output, optionally keeps conv9's
Act encoder(Plan& P, const std::string& p, Act& in, Act* keep9, Layer* first = nullptr) {
  Act x = first ? P.conv_layer(*first, in) : P.conv((p + "1").c_str(), in);
  x = P.conv((p + "2_downsample").c_str(), x);
  x = P.conv((p + "3").c_str(), x);
  x = P.conv((p + "4_downsample").c_str(), x);
  x = P.conv((p + "5").c_str(), x);
  x = P.conv((p + "6").c_str(), x);
  x = P.conv((p + "7_atrous").c_str(), x);
  x = P.conv((p + "8_atrous").c_str(), x);
  Act x9 = P.conv((p + "9_atrous").c_str(), x);
  Act x10 = P.conv((p + "10_atrous").c_str(), x9, keep9 == nullptr);
  if (keep9) *keep9 = x9;
  return x10;
}

// decoder conv11..conv16 (returns the 12-channel full-resolution activation feeding conv17)
Act decoder(Plan& P, const std::string& p, Act& in, const float* src1 = nullptr, int C1 = 0, int vec = 0) {
  Act x = P.conv((p + "11").c_str(), in, true, src1, C1, vec);
  x = P.conv((p + "12").c_str(), x);
  x = P.conv((p + "13_upsample_conv").c_str(), x);
  x = P.conv((p + "14").c_str(), x);
  x = P.conv((p + "15_upsample_conv").c_str(), x);
  x = P.conv((p + "16").c_str(), x);
  return x;
}

int small(Plan& P, const char* name, Act& in, int mode, float* out_nchw, float* hard, const float* img,
          const float* mask, float* xnow, float* composed, int no_mask_coarse, long packed_bs = 0) {
  if (P.rc) return P.rc;
  se_ctx* c = P.c;
  if (!c->dry) {
    Layer& L = P.net.at(name);
    if (!L.packed) return P.rc = fail(c, "layer %s: weights not loaded", name);
    SmallConvParams sp;
    memset(&sp, 0, sizeof sp);
    sp.x = in.p; sp.w = c->bf16 ? L.d_w16 : L.d_w; sp.b = L.d_b; sp.B = P.B; sp.H = in.H; sp.W = in.W; sp.cout = L.def.cout;
    sp.bf16 = c->bf16 ? 1 : 0;
    sp.mode = mode; sp.out_nchw = out_nchw; sp.hard = hard; sp.img = img; sp.mask = mask; sp.xnow = xnow;
    sp.composed = composed; sp.no_mask_coarse = no_mask_coarse;
    if (mode == 3) { sp.rgb8 = c->rgb8; sp.m8 = c->m8; }
    if (packed_bs) {      // SE_FLAG_PACKED_OUT: soft mask and composite live in one (B,4,H,W) buffer
      if (mode == 0) sp.out_bs = packed_bs;
      if (mode == 3) { sp.mask_bs = packed_bs; sp.comp_bs = packed_bs; }
    }
    set_launch_cost(2.0 * (double)P.B * in.H * in.W * L.def.cout * 108.0,
                    4.0 * (double)P.B * in.H * in.W * (12 + L.def.cout), L.def.name);
    hipError_t e = launch_small_conv(sp, c->st);
    if (e != hipSuccess) return P.rc = fail(c, "small conv %s: %s", name, hipGetErrorString(e));
  }
  P.free(in);
  return 0;
}

// MDGenerator.forward, editline2_g.py:59-94
int plan_netM(se_ctx* c, const float* image, const float* sketch, float* mask_out, float* hard_out, float* maskim_out,
              int B, int H, int W, long packed_bs = 0) {
  Plan P(c, c->M, B);
  c->cur_net = SE_NET_M;
  Act in = P.alloc(H, W, 4);
  if (P.rc) return P.rc;
  if (!c->dry) HIPCHK(c, c->bf16 ? launch_pack_m16(image, sketch, in.p, B, H, W, c->st) : launch_pack_m(image, sketch, in.p, B, H, W, c->st));
  Act x9;
  const bool want_img = maskim_out != nullptr;
  Act x10 = encoder(P, "conv", in, want_img ? &x9 : nullptr);
  if (want_img) {   // image decoder consumes conv9's output (quirk, editline2_g.py:76-77)
    Act d = decoder(P, "conv", x9);
    small(P, "conv17", d, 1, maskim_out, nullptr, nullptr, nullptr, nullptr, nullptr, 0);
  }
  Act d = decoder(P, "conv_mask_", x10);
  small(P, "conv_mask_17", d, 0, mask_out, hard_out, nullptr, nullptr, nullptr, nullptr, 0, packed_bs);
  return P.rc;
}

int run_attention(se_ctx* c, Plan& P, Act& x, const float* mask_full, Act& out, float* similar_nchw);

// DeepFillC2Generator.forward, editline_g.py:119-221.  The two encoders of stage 1 and the two branches of stage 2 are
// independent until their concat: in low-latency mode the second one of each pair runs on the side stream.
int plan_netG(se_ctx* c, const float* x, const float* x2, const float* mask, const float* mask2, const float* guide,
              float* coarse_out, float* fine_out, const float* soft_mask, float* composed_out, int B, int H, int W,
              int flags, long packed_bs = 0) {
  Plan P(c, c->G, B);
  c->cur_net = SE_NET_G;
  const int joint = (flags & SE_FLAG_JOINT_TRAIN_INP) ? 1 : 0;
  // joint_train_inp: the style input is packed without its (zero) guide channel and wconv1 runs as a 4-channel conv
  Act cin = P.alloc(H, W, 8);
  P.ar = P.forked() ? &c->arena2 : &c->arena;       // the style branch's input lives in the style branch's arena
  Act sin = P.alloc(H, W, joint ? 4 : 8);
  P.ar = &c->arena;
  if (P.rc) return P.rc;
  if (!c->dry)
    HIPCHK(c, (c->bf16 ? launch_pack_g16 : launch_pack_g)(x, x2, mask, mask2, guide, cin.p, sin.p, B, H, W,
                                                          (flags & SE_FLAG_NO_MASK_CC) ? 1 : 0, joint, c->st));
  // ---- style branch :149-163 -> (B,96) pooled vector, consumed by conv11 as a second (spatially constant) source
  if (P.side_begin()) return 1;
  Act xs = encoder(P, "wconv", sin, nullptr, joint ? &c->wconv1_j4 : nullptr);
  // (in bf16 mode `part` / `vec32` hold fp32 values in twice the room they need; `vec` is the bf16 vector conv11 reads)
  Act part = P.alloc(1, COLREDUCE_SPLITS, 192), vec32 = P.alloc(1, 1, 192), vec = P.alloc(1, 1, 96);
  if (P.rc) return P.rc;
  if (!c->dry)
    HIPCHK(c, launch_colreduce(xs.p, part.p, c->bf16 ? vec32.p : vec.p, B, xs.H * xs.W, 96, (flags & SE_FLAG_POOL_MAX) ? 0 : 1,
                               c->st, c->bf16 ? 1 : 0, c->bf16 ? vec.p : nullptr));
  if (!c->dry && c->taps && c->taps->style_vec)
    HIPCHK(c, hipMemcpyAsync(c->taps->style_vec, c->bf16 ? vec32.p : vec.p, (size_t)B * 96 * 4, hipMemcpyDeviceToDevice, c->st));
  P.free(xs);
  P.free(part);
  if (!c->bf16) P.free(vec32);        // bf16 mode: the fp32 vector feeds the bias table of conv11 (released after the decoder)
  if (P.side_end()) return 1;
  // ---- coarse branch :138-147
  Act xc = encoder(P, "conv", cin, nullptr);
  if (P.rc) return P.rc;
  if (P.join()) return 1;
  float* vb = P.alloc_raw((size_t)B * 9 * 192);     // bias table of the folded style vector (run_gconv)
  if (P.rc) return P.rc;
  c->vbias_ws = vb;
  c->vec32 = c->bf16 ? vec32.p : nullptr;
  Act d = decoder(P, "conv", xc, vec.p, 96, 1);     // :167-175 (cat is virtual)
  c->vbias_ws = nullptr; c->vec32 = nullptr;
  P.release(vb);
  if (c->bf16) P.free(vec32);
  P.free(vec);
  Act xnow = P.alloc(H, W, 4);
  if (P.rc) return P.rc;
  small(P, "conv17", d, 2, coarse_out, nullptr, x, mask, xnow.p, nullptr, (flags & SE_FLAG_NO_MASK_COARSE) ? 1 : 0);
  if (P.rc) return P.rc;
  // ---- patch-match branch :197-209 (side)
  if (P.side_begin()) return 1;
  Act pm = P.conv("pmconv1", xnow, false);
  pm = P.conv("pmconv2_downsample", pm);
  pm = P.conv("pmconv3", pm);
  pm = P.conv("pmconv4_downsample", pm);
  pm = P.conv("pmconv5", pm);
  pm = P.conv("pmconv6", pm);
  if (P.rc) return P.rc;
  if (!c->dry && c->taps && c->taps->pmconv6)
    HIPCHK(c, (c->bf16 ? launch_nhwc16_to_nchw : launch_nhwc_to_nchw)(pm.p, c->taps->pmconv6, B, 96, 96, pm.H, pm.W, c->st));
  if (flags & SE_FLAG_USE_CAM) {
    Act att = P.alloc(pm.H, pm.W, 96);
    if (P.rc) return P.rc;
    if (run_attention(c, P, pm, mask, att, nullptr)) return P.rc ? P.rc : 1;
    P.free(pm);
    pm = att;
    if (!c->dry && c->taps && c->taps->attn_out)
      HIPCHK(c, (c->bf16 ? launch_nhwc16_to_nchw : launch_nhwc_to_nchw)(pm.p, c->taps->attn_out, B, 96, 96, pm.H, pm.W, c->st));
  }
  pm = P.conv("pmconv9", pm);
  pm = P.conv("pmconv10", pm);
  if (P.rc) return P.rc;
  if (P.side_end()) return 1;
  // ---- hallucination branch :184-194
  Act hcur = P.conv("xconv1", xnow, false);
  hcur = P.conv("xconv2_downsample", hcur);
  hcur = P.conv("xconv3", hcur);
  hcur = P.conv("xconv4_downsample", hcur);
  hcur = P.conv("xconv5", hcur);
  hcur = P.conv("xconv6", hcur);
  hcur = P.conv("xconv7_atrous", hcur);
  hcur = P.conv("xconv8_atrous", hcur);
  hcur = P.conv("xconv9_atrous", hcur);
  Act hallu = P.conv("xconv10_atrous", hcur);
  if (P.rc) return P.rc;
  if (P.join()) return 1;
  P.free(xnow);                                         // read by both branches: released after the join
  Act d2 = decoder(P, "allconv", hallu, pm.p, 96, 0);   // cat([x_hallu, pm]) :211 is virtual
  P.free(pm);
  small(P, "allconv17", d2, 3, fine_out, nullptr, x, soft_mask, nullptr, soft_mask ? composed_out : nullptr, 0, packed_bs);
  c->rgb8 = nullptr; c->m8 = nullptr;
  return P.rc;
}

// Row stride (in elements) of the R x R matrices E, P, P~ and of the transposed values: R rounded up to whole k-chunks
// (32 fp32 / 64 bf16 keys).  (Padding the power-of-two strides of 256x256 / 512x512 inputs by one chunk was measured:
// no gain for the row-streaming passes -- they are not camping on HBM channels.)
int att_row_stride(int R, bool bf) {
  const int chunk = bf ? 64 : 32;
  int Rp = (R + chunk - 1) / chunk * chunk;
  return Rp;
}

int run_attention(se_ctx* c, Plan& P, Act& x, const float* mask_full, Act& out, float* similar_nchw) {
  const int h = x.H, w = x.W, B = P.B;
  const int hs = (h - 4) / 2 + 1, ws = (w - 4) / 2 + 1, L = hs * ws, Lp = (L + 31) & ~31;
  const bool bf = c->bf16;
  const bool v2 = attention_v2_enabled() || bf;       // the bf16 path exists in the space-to-depth form only
  const int hc = h / 2, wc = w / 2, R = hc * wc;
  const int Rp = att_row_stride(R, bf);
  const int guard = (wc + 8 + 63) & ~63;               // floats; att2_ptilde_kernel reads up to wc + 5 columns outside a row
  if (v2 && (double)R * Rp * 4.0 >= 2147483648.0) return P.rc = fail(c, "attention: %dx%d feature map too large", h, w);
  // fp32 scratch is sized in floats whatever the activation type
  float* part = P.alloc_raw((size_t)B * COLREDUCE_SPLITS * 96);
  float* rn = P.alloc_raw((size_t)B * 96);
  float* xn = P.alloc_raw(bf ? ((size_t)B * h * w * 96 + 1) / 2 : (size_t)B * h * w * 96);
  float *valid = nullptr, *S = nullptr, *S2 = nullptr, *xT = nullptr, *stats = nullptr;
  if (v2) {
    valid = P.alloc_raw(7 * ((size_t)B * Rp + guard) + (size_t)B * 768 + (size_t)B * hc * 384);      // validR, kmul, kadd (+ fp16-E form: kadd2, ea4, ea, eb), each behind its guard band; emean, epart
    stats = P.alloc_raw((size_t)B * R * 2);    // fused streaming pass: (row max, 1 / row sum) per query
    xT = P.alloc_raw(bf ? (size_t)B * 4 * 96 * Rp / 2 : (size_t)B * 4 * 96 * Rp);
    S = P.alloc_raw((size_t)B * R * Rp + 2 * guard);       // E (fp32) between two guard bands; three-pass form: then P~
    S2 = P.alloc_raw(bf ? (size_t)B * R * Rp / 2 : (size_t)B * R * Rp);      // P (fp32 / bf16)
  } else {
    valid = P.alloc_raw((size_t)B * Lp);
    S = P.alloc_raw((size_t)B * L * Lp);
  }
  if (P.rc) return P.rc;
  if (!c->dry) {
    HIPCHK(c, launch_colreduce(x.p, part, rn, B, h * w, 96, 2, c->st, bf ? 1 : 0));
    AttParams a;
    memset(&a, 0, sizeof a);
    a.x = x.p; a.rn = rn; a.xn = xn; a.hard = mask_full; a.out = out.p;
    a.B = B; a.h = h; a.w = w; a.hs = hs; a.ws = ws; a.L = L; a.Lp = Lp;
    a.scale = 10.f; a.th = 0.1f;                        // editline_g.py:35-38
    if (v2) {
      a.hc = hc; a.wc = wc; a.R = R; a.Rp = Rp; a.bf16 = bf ? 1 : 0;
      a.validR = valid + guard; a.xT = xT; a.E = S + guard; a.P = S2; a.stats = stats; a.guard = guard; a.similar = similar_nchw;
      a.kmul = valid + ((size_t)B * Rp + guard) + guard; a.kadd = valid + 2 * ((size_t)B * Rp + guard) + guard;
      a.kadd2 = valid + 3 * ((size_t)B * Rp + guard) + guard; a.ea4 = valid + 4 * ((size_t)B * Rp + guard) + guard;
      a.ea = valid + 5 * ((size_t)B * Rp + guard) + guard; a.eb = valid + 6 * ((size_t)B * Rp + guard) + guard;
      a.emean = valid + 7 * ((size_t)B * Rp + guard); a.epart = a.emean + (size_t)B * 768;
      HIPCHK(c, launch_attention(a, c->st));
    } else {
      a.valid = valid; a.S = S;
      HIPCHK(c, launch_attention(a, c->st));
      if (similar_nchw) {
        // (B, Lk, hs, ws) <- S[b][i][j]: "channel" j, pixel i
        HIPCHK(c, launch_nhwc_to_nchw(S, similar_nchw, B, L, Lp, hs, ws, c->st));
      }
    }
  }
  P.release(S2); P.release(S); P.release(xT); P.release(stats); P.release(valid);
  P.release(xn); P.release(rn); P.release(part);
  return 0;
}

int check_dims(se_ctx* c, int B, int H, int W) {
  if (B < 1 || H < 16 || W < 16 || (H % 8) || (W % 8)) return fail(c, "bad shape B=%d H=%d W=%d (H, W must be multiples of 8, >= 16)", B, H, W);
  return 0;
}

// Passes of a forward over a large batch.  The kernels address a tensor with 32-bit byte offsets (addr_limit); the largest
// tensor of either network is the 24-channel full-resolution activation (conv1 / conv15_upsample outputs: 96 bytes per
// pixel in fp32, 48 in bf16 -- max_act_bytes_per_image derives it from the plans), so a forward runs over at most
// (limit - 1) / (H W 96) images at a time and a larger batch is
// run as several passes of the SAME plan over image ranges -- same kernels, same execution mode, so an image's result does
// not depend on the pass it lands in (bit for bit: test_large_batch_is_split_into_passes).  Passes are balanced
// (ceil(B / passes) images each).  Returns the images per pass, or 0 (error set) when ONE image exceeds the range.
// bytes per image of the LARGEST activation tensor either plan allocates at this size -- derived from a dry run of the plans
// themselves (the literal "24 channels at full resolution = 96 / 48 bytes per pixel" would silently go stale with
// a wider layer); cached per (H, W, precision)
long long max_act_bytes_per_image(se_ctx* c, int H, int W, int flags) {
  const std::vector<long long> key = {H, W, (flags & SE_FLAG_BF16) ? 1 : 0};
  auto it = c->act_per_img.find(key);
  if (it != c->act_per_img.end()) return it->second;
  const bool dry0 = c->dry, ll0 = c->low_latency, bf0 = c->bf16;
  c->dry = true; c->low_latency = false; c->bf16 = (flags & SE_FLAG_BF16) != 0;
  c->dry_max_act = 0;
  float dummy;
  c->arena.reset(nullptr, 0, true, 0); c->arena2.reset(nullptr, 0, true, 1);
  plan_netM(c, nullptr, nullptr, nullptr, nullptr, &dummy, 1, H, W, 0);
  c->arena.reset(nullptr, 0, true, 0); c->arena2.reset(nullptr, 0, true, 1);
  plan_netG(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 1, H, W, SE_FLAG_USE_CAM | SE_FLAG_POOL_MAX, 0);
  c->dry = dry0; c->low_latency = ll0; c->bf16 = bf0;
  const long long v = (long long)c->dry_max_act;
  c->act_per_img[key] = v;
  return v;
}

int pass_size(se_ctx* c, int B, int H, int W, int flags) {
  const long long per_img = max_act_bytes_per_image(c, H, W, flags);
  const long long lim = addr_limit();
  if (per_img <= 0) { fail(c, "internal: no activation size for %dx%d", H, W); return 0; }
  if (per_img >= lim) {
    fail(c, "a %dx%d image exceeds the kernels' 32-bit byte offsets (%lld bytes per activation, limit %lld)", H, W, per_img, lim);
    return 0;
  }
  const long long bmax = (lim - 1) / per_img;
  if (B <= bmax) return B;
  const long long passes = (B + bmax - 1) / bmax;
  return (int)((B + passes - 1) / passes);
}

// Arena peaks of a forward, from a dry run of the very plan that will be launched (nothing is enqueued): main-branch
// arena and, in low-latency mode, the side-branch arena.  which: 1 netM, 2 netG, 3 netM then netG (se_inference).
se_ctx::Peaks plan_peaks(se_ctx* c, int which, int B, int H, int W, int flags, bool want_maskim) {
  const std::vector<long long> key = {which, B, H, W,
                                      flags & (SE_FLAG_USE_CAM | SE_FLAG_JOINT_TRAIN_INP | SE_FLAG_LOW_LATENCY | SE_FLAG_BF16),
                                      want_maskim ? 1 : 0, attention_v2_enabled() ? 1 : 0, opt_epoch(), c->serial ? 1 : 0};
  auto it = c->peaks.find(key);
  if (it != c->peaks.end()) return it->second;
  const bool dry0 = c->dry, ll0 = c->low_latency, bf0 = c->bf16;
  c->dry = true;
  c->low_latency = (flags & SE_FLAG_LOW_LATENCY) != 0;
  c->bf16 = (flags & SE_FLAG_BF16) != 0;
  float dummy;      // non-null marker for optional outputs
  se_ctx::Peaks pk{0, 0};
  if (which & 1) {
    c->arena.reset(nullptr, 0, true, 0); c->arena2.reset(nullptr, 0, true, 1);
    plan_netM(c, nullptr, nullptr, nullptr, nullptr, want_maskim ? &dummy : nullptr, B, H, W);
    pk.main = c->arena.peak; pk.side = c->arena2.peak;
  }
  if (which & 2) {
    c->arena.reset(nullptr, 0, true, 0); c->arena2.reset(nullptr, 0, true, 1);
    plan_netG(c, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, B, H, W, flags);
    if (c->arena.peak > pk.main) pk.main = c->arena.peak;
    if (c->arena2.peak > pk.side) pk.side = c->arena2.peak;
  }
  c->dry = dry0; c->low_latency = ll0; c->bf16 = bf0;
  c->peaks[key] = pk;
  return pk;
}

// carve the caller's workspace into the arenas of this call; `tail` bytes at the end stay reserved
int carve(se_ctx* c, const se_ctx::Peaks& pk, void* ws, size_t ws_bytes, size_t tail) {
  if (ws_bytes < pk.main + pk.side + tail)
    return fail(c, "workspace too small: %zu bytes, this call needs %zu", ws_bytes, pk.main + pk.side + tail);
  // the main arena gets whatever the side arena does not need
  c->arena.reset((char*)ws, ws_bytes - tail - pk.side, false);
  c->arena2.reset((char*)ws + (ws_bytes - tail - pk.side), pk.side, false);
  return 0;
}

void begin_call(se_ctx* c, void* stream, int flags) {
  c->st = c->st_main = (hipStream_t)stream;
  c->dry = false;
  c->low_latency = (flags & SE_FLAG_LOW_LATENCY) != 0;
  c->bf16 = (flags & SE_FLAG_BF16) != 0;
  c->conservative = (flags & SE_FLAG_CONSERVATIVE) != 0;
  c->serial = c->prof.on;
  c->taps = nullptr;
  c->rgb8 = nullptr; c->m8 = nullptr;
  c->cur_net = SE_NET_G;       // per-op entry points run as netG layers whatever plan ran last; the plans set their own
  set_profiler(&c->prof);
}

// A captured forward may still be executing on the stream it was last launched on: destroying an in-flight hipGraphExec
// is not safe on every ROCm version, so that stream is drained first (eviction and weight reloads are rare events).
void destroy_graph(se_ctx::GraphEntry& g) {
  if (g.exec) {
    if (g.stream) (void)hipStreamSynchronize(g.stream);
    (void)hipGraphExecDestroy(g.exec);
  }
  if (g.graph) (void)hipGraphDestroy(g.graph);
  g.exec = nullptr; g.graph = nullptr;
}
void drop_graphs(se_ctx* c) {
  for (auto& g : c->graphs) destroy_graph(g);
  c->graphs.clear();
}
// cache full: evict the ONE least recently used entry (hot graphs survive callers that pass fresh output tensors)
void evict_lru_graph(se_ctx* c) {
  size_t lru = 0;
  for (size_t i = 1; i < c->graphs.size(); ++i)
    if (c->graphs[i].last_use < c->graphs[lru].last_use) lru = i;
  destroy_graph(c->graphs[lru]);
  c->graphs.erase(c->graphs.begin() + lru);
}

}  // namespace

// ====================================================================================================
extern "C" {

const char* se_version(void) { return "sketchedit_hip 0.1 (gfx950)"; }

int se_create(int device_id, se_ctx** out) {
  if (!out) return 1;
  *out = nullptr;
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || device_id < 0 || device_id >= n) return fail(nullptr, "no such HIP device %d (count %d)", device_id, n);
  se_ctx* c = new (std::nothrow) se_ctx();
  if (!c) return 1;
  c->device = device_id;
  if (hipSetDevice(device_id) != hipSuccess || hipMalloc(&c->zeros, 4096) != hipSuccess ||
      hipMemset(c->zeros, 0, 4096) != hipSuccess) {
    delete c;
    return fail(nullptr, "device init failed");
  }
  // side-branch stream + fork/join events of the low-latency mode (concurrent branches of netG)
  if (hipStreamCreateWithFlags(&c->st_side, hipStreamNonBlocking) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&c->ev_join, hipEventDisableTiming) != hipSuccess) {
    se_destroy(c);
    return fail(nullptr, "stream / event creation failed");
  }
  {
    // data/testimage_dataset.py:89-111 of the reference (ToTensor: /255, Normalize: -0.5, /0.5), one IEEE fp32 operation at a
    // time on the host (volatile: no reassociation, no reciprocal): the table the device looks up
    float lut[256];
    for (int v = 0; v < 256; ++v) {
      volatile float t = (float)v;
      t = t / 255.0f;
      t = t - 0.5f;
      t = t / 0.5f;
      lut[v] = t;
    }
    if (hipMalloc(&c->lut8, sizeof lut) != hipSuccess || hipMemcpy(c->lut8, lut, sizeof lut, hipMemcpyHostToDevice) != hipSuccess) {
      se_destroy(c);
      return fail(nullptr, "device init failed (dequantisation table)");
    }
  }
  for (int i = 0; i < NG; ++i) c->G[G_LAYERS[i].name].def = G_LAYERS[i];
  for (int i = 0; i < NM; ++i) c->M[M_LAYERS[i].name].def = M_LAYERS[i];
  // (the 4-channel form of wconv1 is planned by the dry runs of se_workspace_bytes before any weight exists: a zero stride in
  // its definition was a division by zero for a ctx that had only netM's weights -- found by tools/f43_flips.py, round 5)
  c->wconv1_j4.def = c->G.at("wconv1").def;
  *out = c;
  return 0;
}

void se_destroy(se_ctx* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  for (auto* net : {&c->G, &c->M})
    for (auto& kv : *net) {
      if (kv.second.d_w) (void)hipFree(kv.second.d_w);
      if (kv.second.d_b) (void)hipFree(kv.second.d_b);
      if (kv.second.d_u) (void)hipFree(kv.second.d_u);
      if (kv.second.d_ub) (void)hipFree(kv.second.d_ub);
      if (kv.second.d_w16) (void)hipFree(kv.second.d_w16);
      if (kv.second.d_w16s) (void)hipFree(kv.second.d_w16s);
      if (kv.second.d_w16d) (void)hipFree(kv.second.d_w16d);
      if (kv.second.d_w96) (void)hipFree(kv.second.d_w96);
      if (kv.second.d_wd) (void)hipFree(kv.second.d_wd);
      if (kv.second.d_wdw) (void)hipFree(kv.second.d_wdw);
      if (kv.second.d_u1) (void)hipFree(kv.second.d_u1);
      if (kv.second.d_wv) (void)hipFree(kv.second.d_wv);
      if (kv.second.d_wv16) (void)hipFree(kv.second.d_wv16);
      if (kv.second.d_wx) (void)hipFree(kv.second.d_wx);
      if (kv.second.d_wx2) (void)hipFree(kv.second.d_wx2);
      if (kv.second.d_u24) (void)hipFree(kv.second.d_u24);
      if (kv.second.d_ub24) (void)hipFree(kv.second.d_ub24);
      if (kv.second.d_u24b) (void)hipFree(kv.second.d_u24b);
    }
  if (c->wconv1_j4.d_w) (void)hipFree(c->wconv1_j4.d_w);
  if (c->wconv1_j4.d_b) (void)hipFree(c->wconv1_j4.d_b);
  if (c->wconv1_j4.d_w16) (void)hipFree(c->wconv1_j4.d_w16);
  if (c->wconv1_j4.d_w16d) (void)hipFree(c->wconv1_j4.d_w16d);
  if (c->wconv1_j4.d_wd) (void)hipFree(c->wconv1_j4.d_wd);
  if (c->wconv1_j4.d_wdw) (void)hipFree(c->wconv1_j4.d_wdw);
  if (c->zeros) (void)hipFree(c->zeros);
  if (c->lut8) (void)hipFree(c->lut8);
  for (auto& e : c->prof.pool) (void)hipEventDestroy(e);
  drop_graphs(c);
  if (c->ev_fork) (void)hipEventDestroy(c->ev_fork);
  if (c->ev_join) (void)hipEventDestroy(c->ev_join);
  if (c->st_side) (void)hipStreamDestroy(c->st_side);
  delete c;
}

// the message of the last failed call on this ctx, copied under the ctx lock into a per-thread buffer (threaded callers)
const char* se_last_error(se_ctx* c) {
  if (!c) return g_create_err.c_str();
  thread_local std::string copy;
  std::lock_guard<std::mutex> lk(c->mu);
  copy = c->err;
  return copy.c_str();
}

int se_load_weights(se_ctx* c, int net_id, const char* name, const float* host, const int* shape, int ndim) {
  if (!c || !name || !host || !shape) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  (void)hipSetDevice(c->device);
  drop_graphs(c);                       // captured forwards hold the old weight images
  std::string key(name);
  if (key.rfind("module.", 0) == 0) key = key.substr(7);          // util/util.py:221-222
  const size_t dot = key.rfind('.');
  if (dot == std::string::npos) return fail(c, "bad key %s", name);
  const std::string lname = key.substr(0, dot), kind = key.substr(dot + 1);
  auto& net = net_id == SE_NET_G ? c->G : c->M;
  auto it = net.find(lname);
  if (it == net.end()) return fail(c, "unexpected key %s for net %d", name, net_id);
  Layer& L = it->second;
  const LayerDef& d = L.def;
  if (kind == "weight") {
    if (ndim != 4 || shape[0] != d.cout || shape[1] != d.cin || shape[2] != d.k || shape[3] != d.k)
      return fail(c, "size mismatch for %s", name);
    L.w.assign(host, host + (size_t)d.cout * d.cin * d.k * d.k);
    L.have_w = true;
  } else if (kind == "bias") {
    if (ndim != 1 || shape[0] != d.cout) return fail(c, "size mismatch for %s", name);
    L.b.assign(host, host + d.cout);
    L.have_b = true;
  } else {
    return fail(c, "unexpected key %s", name);
  }
  if (L.have_w && L.have_b) return pack_net_layer(c, L);
  return 0;
}

int se_weights_ready(se_ctx* c) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(c->mu);
  for (auto* net : {&c->G, &c->M})
    for (auto& kv : *net)
      if (!kv.second.packed) return 0;
  return 1;
}

size_t se_workspace_bytes(se_ctx* c, int B, int H, int W) {
  if (!c) return 0;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 0;
  c->fork2 = opt(OPT_FORK_DEFAULT) != 0;
  // every flag combination that changes the allocation sequence (a first-fit arena does not peak monotonically):
  // attention on/off, 4- or 8-channel style input, in-line or concurrent branches, fp32 or bf16 activations
  size_t peak = 0;
  for (int cam = 0; cam < 2; ++cam)
    for (int joint = 0; joint < 2; ++joint)
      for (int ll = 0; ll < 2; ++ll)
        for (int bf = 0; bf < 2; ++bf) {      // (bf16 activations are smaller, but its attention scratch need not be)
          const int flags = (cam ? SE_FLAG_USE_CAM : 0) | (joint ? SE_FLAG_JOINT_TRAIN_INP : 0) | (ll ? SE_FLAG_LOW_LATENCY : 0) |
                            (bf ? SE_FLAG_BF16 : 0) | SE_FLAG_POOL_MAX;
          // a batch beyond the kernels' 32-bit byte offsets runs as passes (pass_size): the workspace serves one pass at a time
          const int nb = pass_size(c, B, H, W, flags);
          if (!nb) return 0;
          const int last = B % nb;                      // size of a ragged last pass (0: none)
          // with and without netM's image decoder (mode='visualize' vs 'inference'): two allocation sequences
          // ... each on two streams and on one (the profiler's plan of the default mode: c->serial)
          for (int ser = 0; ser < 2; ++ser)
            for (int mi = 0; mi < 2; ++mi)
              for (int bb : {nb, last}) {
                if (!bb) continue;
                c->serial = ser != 0;
                const se_ctx::Peaks pk = plan_peaks(c, 3, bb, H, W, flags, mi != 0);
                if (pk.main + pk.side > peak) peak = pk.main + pk.side;
              }
          c->serial = false;
        }
  // + hard-mask (se_inference) and soft-mask (se_inference_u8) planes + the fp32 image / sketch of se_inference_u8io (4 planes)
  return peak + 6 * (((size_t)B * H * W * 4 + 255) & ~(size_t)255);
}

int se_netM_forward_ex(se_ctx* c, void* stream, const float* image, const float* sketch, float* mask_out,
                       float* maskim_out, void* ws, size_t ws_bytes, int B, int H, int W, int exec_flags) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if (!image || !sketch || !mask_out || !ws) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->serial = c->prof.on;      // profiler on: default mode planned on one stream (Plan::forked)
  c->fork2 = opt(OPT_FORK_DEFAULT) != 0;
  exec_flags &= SE_FLAG_LOW_LATENCY | SE_FLAG_BF16 | SE_FLAG_CONSERVATIVE;
  const int nb = pass_size(c, B, H, W, exec_flags);
  if (!nb) return 1;
  const size_t HW = (size_t)H * W;
  for (int b0 = 0; b0 < B; b0 += nb) {
    const int bb = std::min(nb, B - b0);
    if (carve(c, plan_peaks(c, 1, bb, H, W, exec_flags, maskim_out != nullptr), ws, ws_bytes, 0)) return 1;
    begin_call(c, stream, exec_flags);
    const int rc = plan_netM(c, image + b0 * 3 * HW, sketch + b0 * HW, mask_out + b0 * HW, nullptr,
                             maskim_out ? maskim_out + b0 * 3 * HW : nullptr, bb, H, W);
    if (rc) return rc;
  }
  return 0;
}

int se_netM_forward(se_ctx* c, void* stream, const float* image, const float* sketch, float* mask_out,
                    float* maskim_out, void* ws, size_t ws_bytes, int B, int H, int W) {
  return se_netM_forward_ex(c, stream, image, sketch, mask_out, maskim_out, ws, ws_bytes, B, H, W, 0);
}

int se_netG_forward_taps(se_ctx* c, void* stream, const float* x, const float* x2, const float* mask, const float* mask2,
                         const float* guide, float* coarse_out, float* fine_out, void* ws, size_t ws_bytes, int B, int H,
                         int W, int flags, const se_netG_taps* taps) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if (!x || !x2 || !mask || !mask2 || !guide || !fine_out || !ws) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->serial = c->prof.on;      // profiler on: default mode planned on one stream (Plan::forked)
  c->fork2 = opt(OPT_FORK_DEFAULT) != 0;
  const int nb = pass_size(c, B, H, W, flags);
  if (!nb) return 1;
  if (taps && nb < B) return fail(c, "se_netG_forward_taps: %d images do not fit one pass (%d)", B, nb);
  const size_t HW = (size_t)H * W;
  for (int b0 = 0; b0 < B; b0 += nb) {
    const int bb = std::min(nb, B - b0);
    if (carve(c, plan_peaks(c, 2, bb, H, W, flags, false), ws, ws_bytes, 0)) return 1;
    begin_call(c, stream, flags);
    c->taps = taps;
    const int rc = plan_netG(c, x + b0 * 3 * HW, x2 + b0 * 3 * HW, mask + b0 * HW, mask2 + b0 * HW, guide + b0 * HW,
                             coarse_out ? coarse_out + b0 * 3 * HW : nullptr, fine_out + b0 * 3 * HW, nullptr, nullptr, bb, H, W, flags);
    c->taps = nullptr;
    if (rc) return rc;
  }
  return 0;
}

int se_netG_forward(se_ctx* c, void* stream, const float* x, const float* x2, const float* mask, const float* mask2,
                    const float* guide, float* coarse_out, float* fine_out, void* ws, size_t ws_bytes, int B, int H,
                    int W, int flags) {
  return se_netG_forward_taps(c, stream, x, x2, mask, mask2, guide, coarse_out, fine_out, ws, ws_bytes, B, H, W, flags, nullptr);
}

namespace {
// netM -> threshold -> netG -> composite, enqueued on `stream` (and, in low-latency mode, the ctx's side stream)
int enqueue_inference(se_ctx* c, void* stream, const float* image, const float* sketch, float* composed_out, float* mask_out,
                      float* hard_out, float* maskim_out, float* coarse_out, float* fine_out, void* ws, size_t ws_bytes,
                      int B, int H, int W, int flags) {
  // the hard mask lives at the end of the workspace for the whole call
  const size_t plane = ((size_t)B * H * W * 4 + 255) & ~(size_t)255;
  float* hard_all = hard_out ? hard_out : (float*)((char*)ws + ws_bytes - plane);
  // SE_FLAG_PACKED_OUT: composed_out is one (B,4,H,W) buffer, planes 0-2 the composite, plane 3 the soft mask
  const size_t HW = (size_t)H * W;
  const long packed_bs = (flags & SE_FLAG_PACKED_OUT) ? 4l * H * W : 0;
  if (packed_bs) mask_out = composed_out + 3 * HW;
  const size_t comp_bs = packed_bs ? (size_t)packed_bs : 3 * HW, mask_bs = packed_bs ? (size_t)packed_bs : HW;
  // a batch beyond the kernels' 32-bit byte offsets runs as passes over image ranges (pass_size)
  const int nb = pass_size(c, B, H, W, flags);
  if (!nb) return 1;
  for (int b0 = 0; b0 < B; b0 += nb) {
    const int bb = std::min(nb, B - b0);
    const se_ctx::Peaks pk = plan_peaks(c, 3, bb, H, W, flags, maskim_out != nullptr);
    if (carve(c, pk, ws, ws_bytes, plane)) return 1;
    const float *img = image + b0 * 3 * HW, *sk = sketch + b0 * HW;
    float *hard = hard_all + b0 * HW, *mk = mask_out + b0 * mask_bs;
    begin_call(c, stream, flags);
    int rc = plan_netM(c, img, sk, mk, hard, maskim_out ? maskim_out + b0 * 3 * HW : nullptr, bb, H, W, packed_bs);     // editline2_model.py:339,346-347
    if (rc) return rc;
    if (carve(c, pk, ws, ws_bytes, plane)) return 1;
    // netG(inputs, inputs, mask_inpaint, mask_inpaint, line)  :366-368 ; composite with the soft mask :132
    rc = plan_netG(c, img, img, hard, hard, sk, coarse_out ? coarse_out + b0 * 3 * HW : nullptr, fine_out ? fine_out + b0 * 3 * HW : nullptr,
                   mk, composed_out + b0 * comp_bs, bb, H, W, flags, packed_bs);
    if (rc) return rc;
  }
  return 0;
}
}  // namespace

int se_inference(se_ctx* c, void* stream, const float* image, const float* sketch, float* composed_out,
                 float* mask_out, float* hard_out, float* maskim_out, float* coarse_out, float* fine_out, void* ws,
                 size_t ws_bytes, int B, int H, int W, int flags) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if (!image || !sketch || !composed_out || (!mask_out && !(flags & SE_FLAG_PACKED_OUT)) || !ws) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->serial = c->prof.on;      // profiler on: default mode planned on one stream (Plan::forked)
  c->fork2 = opt(OPT_FORK_DEFAULT) != 0;
  if (!(flags & SE_FLAG_GRAPH) || c->prof.on)
    return enqueue_inference(c, stream, image, sketch, composed_out, mask_out, hard_out, maskim_out, coarse_out, fine_out, ws,
                             ws_bytes, B, H, W, flags);
  // SE_FLAG_GRAPH: the forward for these exact arguments (every pointer is baked into the kernel nodes) is captured
  // into a hipGraph the second time it is seen and replayed from then on: one launch instead of ~85, and the two
  // branches of the low-latency mode become parallel graph nodes.  The first call runs eagerly (it also sets the
  // per-device kernel attributes, which must not happen under capture).
  const std::vector<long long> key = {(long long)(size_t)image, (long long)(size_t)sketch, (long long)(size_t)composed_out,
                                      (long long)(size_t)mask_out, (long long)(size_t)hard_out, (long long)(size_t)maskim_out,
                                      (long long)(size_t)coarse_out, (long long)(size_t)fine_out, (long long)(size_t)ws,
                                      (long long)ws_bytes, B, H, W, flags, opt_epoch()};
  se_ctx::GraphEntry* ge = nullptr;
  for (auto& g : c->graphs)
    if (g.key == key) { ge = &g; break; }
  if (!ge) {
    if (c->graphs.size() >= 16) evict_lru_graph(c);
    c->graphs.emplace_back();
    c->graphs.back().key = key;
    c->graphs.back().last_use = ++c->graph_clock;
    return enqueue_inference(c, stream, image, sketch, composed_out, mask_out, hard_out, maskim_out, coarse_out, fine_out, ws,
                             ws_bytes, B, H, W, flags);
  }
  ge->last_use = ++c->graph_clock;
  if (!ge->exec) {
    if (!stream) return fail(c, "SE_FLAG_GRAPH needs a non-default stream (capture is not permitted on the legacy stream)");
    HIPCHK(c, hipStreamBeginCapture((hipStream_t)stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_inference(c, stream, image, sketch, composed_out, mask_out, hard_out, maskim_out, coarse_out,
                                     fine_out, ws, ws_bytes, B, H, W, flags);
    hipGraph_t graph = nullptr;
    const hipError_t e = hipStreamEndCapture((hipStream_t)stream, &graph);
    if (rc) { if (graph) (void)hipGraphDestroy(graph); return rc; }
    if (e != hipSuccess) return fail(c, "hipStreamEndCapture failed: %s", hipGetErrorString(e));
    hipGraphExec_t exec = nullptr;
    const hipError_t e2 = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    if (e2 != hipSuccess) { (void)hipGraphDestroy(graph); return fail(c, "hipGraphInstantiate failed: %s", hipGetErrorString(e2)); }
    ge->graph = graph; ge->exec = exec;
  }
  ge->stream = (hipStream_t)stream;
  ge->last_use = ++c->graph_clock;
  HIPCHK(c, hipGraphLaunch(ge->exec, (hipStream_t)stream));
  return 0;
}

// The forward with the output quantisation of test.py:25-27 fused into its last kernel: the composite and the soft mask
// leave the device as uint8 only (a quarter of the fp32 bytes, no separate pass); the soft mask needed between netM and
// the final composite lives in the workspace.
namespace {
// `tail_planes` fp32 (B,H,W) planes at the end of the workspace are the caller's; the soft and hard masks sit in front of them
int inference_u8_locked(se_ctx* c, void* stream, const float* image, const float* sketch, unsigned char* rgb_out,
                        unsigned char* mask_u8_out, void* ws, size_t ws_bytes, int B, int H, int W, int flags, int tail_planes) {
  flags &= ~(SE_FLAG_GRAPH | SE_FLAG_PACKED_OUT);
  const size_t plane = ((size_t)B * H * W * 4 + 255) & ~(size_t)255;
  const size_t tail = (2 + (size_t)tail_planes) * plane;
  if (ws_bytes < tail) return fail(c, "workspace too small: %zu bytes", ws_bytes);
  float* hard_all = (float*)((char*)ws + ws_bytes - tail + plane);
  float* soft_all = (float*)((char*)ws + ws_bytes - tail);
  const size_t HW = (size_t)H * W;
  const int nb = pass_size(c, B, H, W, flags);       // passes over image ranges beyond the kernels' 32-bit byte offsets
  if (!nb) return 1;
  for (int b0 = 0; b0 < B; b0 += nb) {
    const int bb = std::min(nb, B - b0);
    const se_ctx::Peaks pk = plan_peaks(c, 3, bb, H, W, flags, false);
    if (carve(c, pk, ws, ws_bytes, tail)) return 1;
    const float *img = image + b0 * 3 * HW, *sk = sketch + b0 * HW;
    float *hard = hard_all + b0 * HW, *soft = soft_all + b0 * HW;
    begin_call(c, stream, flags);
    int rc = plan_netM(c, img, sk, soft, hard, nullptr, bb, H, W);
    if (rc) return rc;
    if (carve(c, pk, ws, ws_bytes, tail)) return 1;
    c->rgb8 = rgb_out + b0 * 3 * HW; c->m8 = mask_u8_out ? mask_u8_out + b0 * HW : nullptr;
    rc = plan_netG(c, img, img, hard, hard, sk, nullptr, nullptr, soft, nullptr, bb, H, W, flags);
    if (rc) return rc;
  }
  return 0;
}
}  // namespace

int se_inference_u8(se_ctx* c, void* stream, const float* image, const float* sketch, unsigned char* rgb_out,
                    unsigned char* mask_u8_out, void* ws, size_t ws_bytes, int B, int H, int W, int flags) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if (!image || !sketch || !rgb_out || !ws) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->serial = c->prof.on;      // profiler on: default mode planned on one stream (Plan::forked)
  c->fork2 = opt(OPT_FORK_DEFAULT) != 0;
  return inference_u8_locked(c, stream, image, sketch, rgb_out, mask_u8_out, ws, ws_bytes, B, H, W, flags, 0);
}

// data/testimage_dataset.py:89-111 on the device (table lookup; see se_create)
int se_dequantize_u8(se_ctx* c, void* stream, const unsigned char* image_u8, const unsigned char* sketch_u8, float* image_out,
                     float* sketch_out, int B, int H, int W) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if ((image_out && !image_u8) || (sketch_out && !sketch_u8)) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  set_profiler(&c->prof);
  HIPCHK(c, launch_dequantize_u8(image_u8, sketch_u8, c->lut8, image_out, sketch_out, B, H, W, (hipStream_t)stream));
  return 0;
}

// uint8 in, uint8 out: the fp32 image (3 planes) and sketch (1 plane) live at the very end of the workspace
int se_inference_u8io(se_ctx* c, void* stream, const unsigned char* image_u8, const unsigned char* sketch_u8,
                      unsigned char* rgb_out, unsigned char* mask_u8_out, void* ws, size_t ws_bytes, int B, int H, int W,
                      int flags) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if (!image_u8 || !sketch_u8 || !rgb_out || !ws) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  c->serial = c->prof.on;      // profiler on: default mode planned on one stream (Plan::forked)
  c->fork2 = opt(OPT_FORK_DEFAULT) != 0;
  const size_t plane = ((size_t)B * H * W * 4 + 255) & ~(size_t)255;
  if (ws_bytes < 6 * plane) return fail(c, "workspace too small: %zu bytes", ws_bytes);
  float* image = (float*)((char*)ws + ws_bytes - 4 * plane);      // (B,3,H,W) contiguous: 3 B H W floats <= 3 planes
  float* sketch = (float*)((char*)ws + ws_bytes - plane);
  set_profiler(&c->prof);
  HIPCHK(c, launch_dequantize_u8(image_u8, sketch_u8, c->lut8, image, sketch, B, H, W, (hipStream_t)stream));
  return inference_u8_locked(c, stream, image, sketch, rgb_out, mask_u8_out, ws, ws_bytes, B, H, W, flags, 4);
}

// test.py:25-27 on the device
int se_quantize_u8(se_ctx* c, void* stream, const float* composed, const float* mask, unsigned char* rgb_out,
                   unsigned char* mask_u8_out, int B, int H, int W) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (check_dims(c, B, H, W)) return 1;
  if ((rgb_out && !composed) || (mask_u8_out && !mask)) return fail(c, "null pointer argument");
  HIPCHK(c, hipSetDevice(c->device));
  set_profiler(&c->prof);
  HIPCHK(c, launch_quantize_u8(composed, mask, rgb_out, mask_u8_out, B, H, W, (hipStream_t)stream));
  return 0;
}

// ---- measurement support (bench.py): per-kernel HIP-event timing ---------------------------------
int se_profile_enable(se_ctx* c, int on) {
  if (!c) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  c->prof.recs.clear();
  c->prof.used = 0;
  if (on && c->prof.pool.empty()) {
    c->prof.pool.resize(2 * 1024);           // enough for ~12 forwards of ~80 launches
    for (auto& e : c->prof.pool) HIPCHK(c, hipEventCreate(&e));
  }
  c->prof.on = on != 0;
  return 0;
}

int se_profile_report(se_ctx* c, char* buf, size_t cap) {
  if (!c || !buf || cap < 64) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  HIPCHK(c, hipDeviceSynchronize());
  double ms[PL_COUNT] = {0}, fl[PL_COUNT] = {0}, by[PL_COUNT] = {0}, ex[PL_COUNT] = {0}, bl[PL_COUNT] = {0};
  long n[PL_COUNT] = {0};
  for (auto& r : c->prof.recs) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) == hipSuccess) { ms[r.label] += t; fl[r.label] += r.flops; ex[r.label] += r.exec_flops; by[r.label] += r.bytes; bl[r.label] += (double)r.blocks; n[r.label]++; }
  }
  std::string s = "{\"kernels\": [";
  bool first = true;
  for (int l = 0; l < PL_COUNT; ++l) {
    if (!n[l]) continue;
    char t[384];
    snprintf(t, sizeof t, "%s{\"kernel\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, \"flops_executed\": %.6e, \"bytes\": %.6e, \"workgroups\": %.1f}",
             first ? "" : ", ", prof_label_name(l), n[l], ms[l], fl[l], ex[l], by[l], bl[l] / (double)n[l]);
    s += t;
    first = false;
  }
  s += "], \"layers\": [";
  // per (kernel, layer name) in first-seen order
  std::vector<std::string> keys;
  std::map<std::string, std::pair<double, double>> agg;   // ms, flops
  std::map<std::string, double> aggx;                       // executed flops
  std::map<std::string, long> cnt;
  for (auto& r : c->prof.recs) {
    if (!r.name) continue;
    float t = 0.f;
    if (hipEventElapsedTime(&t, r.a, r.b) != hipSuccess) continue;
    std::string k = std::string(prof_label_name(r.label)) + ":" + r.name;
    if (!cnt.count(k)) keys.push_back(k);
    agg[k].first += t; agg[k].second += r.flops; aggx[k] += r.exec_flops; cnt[k]++;
  }
  first = true;
  for (auto& k : keys) {
    char t[384];
    snprintf(t, sizeof t, "%s{\"layer\": \"%s\", \"launches\": %ld, \"total_ms\": %.6f, \"flops\": %.6e, \"flops_executed\": %.6e}",
             first ? "" : ", ", k.c_str(), cnt[k], agg[k].first, agg[k].second, aggx[k]);
    s += t;
    first = false;
  }
  s += "]}";
  if (s.size() + 1 > cap) return fail(c, "profile buffer too small");
  memcpy(buf, s.c_str(), s.size() + 1);
  return 0;
}

// ---- developer switches (se_kernels.h SE_OPTIONS): process-wide, read from the environment once ----
int se_debug_set_option(const char* name, int value) { return opt_set(name, value); }
int se_debug_get_option(const char* name, int* value) { return opt_get(name, value); }
void se_debug_reset_options(void) { opt_reset(); }

// ---- unit-test entry points (allocate scratch internally; synchronise the stream before freeing) ----
int se_gated_conv2d_ex(se_ctx* c, void* stream, const float* x, const float* x1, int x1_is_vector, const float* w_host,
                       const float* b_host, float* y, int B, int Cin, int Cin1, int H, int W, int Cout, int k, int stride,
                       int rate, int act, int upsample, int exec_flags) {
  if (!c || !x || !w_host || !b_host || !y) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  HIPCHK(c, hipSetDevice(c->device));
  begin_call(c, stream, exec_flags & (SE_FLAG_LOW_LATENCY | SE_FLAG_BF16));
  const bool bf = c->bf16;
  if (!x1) Cin1 = 0;
  if (x1 && ((Cin % 4) || (Cin1 % 4))) return fail(c, "two-source conv: channel counts must be multiples of 4");
  Layer L;
  static const char* nm = "test";
  const int CinT = Cin + Cin1;
  L.def = LayerDef{nm, CinT, Cout, k, stride, rate, act, upsample};
  L.w.assign(w_host, w_host + (size_t)Cout * CinT * k * k);
  L.b.assign(b_host, b_host + Cout);
  if (bf && x1 && ((Cin % 8) || (Cin1 % 8))) return fail(c, "two-source bf16 conv: channel counts must be multiples of 8");
  const int Cp = bf ? (Cin + 7) & ~7 : (Cin + 3) & ~3;
  float *xin = nullptr, *x1in = nullptr, *yout = nullptr, *vb_test = nullptr;
  int rc = 0;
  HIPCHK(c, hipMalloc(&xin, (size_t)B * H * W * Cp * 4));
  rc = (bf ? launch_nchw_to_nhwc16 : launch_nchw_to_nhwc)(x, xin, B, Cin, Cp, H, W, c->st) != hipSuccess;
  if (!rc && x1) {
    // second source of the virtual concat (editline_g.py:166-167,211): a tensor (B,Cin1,H,W) or a per-image vector (B,Cin1)
    const size_t n1 = x1_is_vector ? (size_t)B * Cin1 : (size_t)B * H * W * Cin1;
    HIPCHK(c, hipMalloc(&x1in, n1 * 4));
    if (x1_is_vector && !bf) rc = hipMemcpyAsync(x1in, x1, n1 * 4, hipMemcpyDeviceToDevice, c->st) != hipSuccess;
    else if (x1_is_vector) rc = launch_nchw_to_nhwc16(x1, x1in, B, Cin1, Cin1, 1, 1, c->st) != hipSuccess;
    else rc = (bf ? launch_nchw_to_nhwc16 : launch_nchw_to_nhwc)(x1, x1in, B, Cin1, Cin1, H, W, c->st) != hipSuccess;
  }
  const bool raw = (act == ACT_NONE) || Cout == 3;    // utils.py:27
  if (!rc && raw) {
    if (k != 3 || CinT != 12 || x1 || (Cout != 1 && Cout != 3) || stride != 1 || rate != 1 || upsample) rc = fail(c, "raw conv: only 3x3 12->{1,3}");
    if (!rc) rc = pack_small(c, L);
    if (!rc) {
      SmallConvParams sp;
      memset(&sp, 0, sizeof sp);
      sp.x = xin; sp.w = bf ? L.d_w16 : L.d_w; sp.b = L.d_b; sp.B = B; sp.H = H; sp.W = W; sp.cout = Cout; sp.mode = 4;
      sp.bf16 = bf ? 1 : 0;
      sp.out_nchw = y;
      rc = launch_small_conv(sp, c->st) != hipSuccess;
    }
  } else if (!rc) {
    if (Cout % 8) rc = fail(c, "gated conv needs Cout %% 8 == 0");
    if (!rc) rc = pack_layer(c, L, identity_map(CinT));
    if (!rc && !bf && k == 5 && stride == 1 && rate == 1 && Cout == 48 && !Cin1 && !upsample && Cin >= 3 && Cin <= 5) {
      std::vector<int> id(Cin);
      for (int i = 0; i < Cin; ++i) id[i] = i;
      rc = pack_layer_dense(c, L, id);
    }
    if (!rc && bf) rc = pack_layer16(c, L, identity_map8(CinT));
    if (!rc && bf && k == 5 && stride == 1 && rate == 1 && Cout == 48 && !Cin1 && !upsample && Cin >= 3 && Cin <= 4) {
      const int m4[4] = {0, 1, 2, Cin == 4 ? 3 : -1};
      rc = pack_layer16_d4(c, L, m4);
    }
    if (!rc && bf && k == 3 && stride == 1 && !upsample && Cout == 192 && ((CinT == 96 && !Cin1) || (CinT == 192 && Cin1 == 96))) rc = pack_rconv16(c, L);
    if (!rc && bf && !Cin1 && rconv96_eligible(L.def)) rc = pack_rconv96(c, L);
    if (!rc) {
      int Ho, Wo;
      c->dry = true; run_gconv(c, L, nullptr, 0, nullptr, 0, 0, nullptr, B, H, W, &Ho, &Wo); c->dry = false;
      const int Gs = bf ? (Cout / 2 + 7) & ~7 : Cout / 2;
      HIPCHK(c, hipMalloc(&yout, (size_t)B * Ho * Wo * Gs * 4));
      if (x1in && x1_is_vector) {               // scratch of the folded vector source (the forwards take it from the workspace)
        HIPCHK(c, hipMalloc(&vb_test, (size_t)B * 9 * 192 * 4));
        c->vbias_ws = vb_test;
        c->vec32 = bf ? x1 : nullptr;           // bf16 mode: the caller's fp32 vector (x1in is its bf16 rounding)
      }
      rc = run_gconv(c, L, xin, Cp, x1in, Cin1, x1_is_vector, yout, B, H, W, nullptr, nullptr);
      if (!rc) rc = (bf ? launch_nhwc16_to_nchw : launch_nhwc_to_nchw)(yout, y, B, Cout / 2, Gs, Ho, Wo, c->st) != hipSuccess;
    }
  }
  (void)hipStreamSynchronize(c->st);
  if (xin) (void)hipFree(xin);
  if (x1in) (void)hipFree(x1in);
  if (yout) (void)hipFree(yout);
  if (L.d_w) (void)hipFree(L.d_w);
  if (L.d_b) (void)hipFree(L.d_b);
  if (L.d_u) (void)hipFree(L.d_u);
  if (L.d_ub) (void)hipFree(L.d_ub);
  if (L.d_w16) (void)hipFree(L.d_w16);
  if (L.d_w16s) (void)hipFree(L.d_w16s);
  if (L.d_w16d) (void)hipFree(L.d_w16d);
  if (L.d_w96) (void)hipFree(L.d_w96);
  if (L.d_wd) (void)hipFree(L.d_wd);
  if (L.d_wdw) (void)hipFree(L.d_wdw);
  if (L.d_u1) (void)hipFree(L.d_u1);
  if (L.d_wv) (void)hipFree(L.d_wv);
  if (L.d_wv16) (void)hipFree(L.d_wv16);
  if (L.d_wx) (void)hipFree(L.d_wx);
  if (L.d_wx2) (void)hipFree(L.d_wx2);
  if (L.d_u24) (void)hipFree(L.d_u24);
  if (L.d_ub24) (void)hipFree(L.d_ub24);
  if (L.d_u24b) (void)hipFree(L.d_u24b);
  if (vb_test) (void)hipFree(vb_test);
  c->vbias_ws = nullptr; c->vec32 = nullptr;
  return rc;
}

int se_gated_conv2d(se_ctx* c, void* stream, const float* x, const float* w_host, const float* b_host, float* y, int B,
                    int Cin, int H, int W, int Cout, int k, int stride, int rate, int act, int upsample) {
  return se_gated_conv2d_ex(c, stream, x, nullptr, 0, w_host, b_host, y, B, Cin, 0, H, W, Cout, k, stride, rate, act,
                            upsample, 0);
}

int se_attention_ex(se_ctx* c, void* stream, const float* x, const float* mask_full, float* out, float* similar_out, int B,
                    int h, int w, int exec_flags) {
  if (!c || !x || !mask_full || !out) return 1;
  std::lock_guard<std::mutex> lk(c->mu);
  if (h < 4 || w < 4 || (h % 2) || (w % 2)) return fail(c, "attention: h, w must be even and >= 4");
  HIPCHK(c, hipSetDevice(c->device));
  begin_call(c, stream, exec_flags & SE_FLAG_BF16);
  const bool bf = c->bf16;
  const int R = (h / 2) * (w / 2), Rp = att_row_stride(R, true) + 64;
  const size_t bytes = ((size_t)B * h * w * 96 * 3 + 2 * (size_t)B * R * Rp + (size_t)B * Rp * (7 + 4 * 96) + 64 * 96 * B + 2 * (size_t)B * R + 9 * (size_t)(w / 2 + 72) + (size_t)B * (768 + (h / 2) * 384)) * 4 + (1 << 16);
  char* ws = nullptr;
  HIPCHK(c, hipMalloc(&ws, bytes));
  c->arena.reset(ws, bytes, false);
  c->arena2.reset(nullptr, 0, false);
  Plan P(c, c->G, B);
  Act xin = P.alloc(h, w, 96), o = P.alloc(h, w, 96);
  int rc = P.rc;
  if (!rc) rc = (bf ? launch_nchw_to_nhwc16 : launch_nchw_to_nhwc)(x, xin.p, B, 96, 96, h, w, c->st) != hipSuccess;
  if (!rc) rc = run_attention(c, P, xin, mask_full, o, similar_out);
  if (!rc) rc = (bf ? launch_nhwc16_to_nchw : launch_nhwc_to_nchw)(o.p, out, B, 96, 96, h, w, c->st) != hipSuccess;
  (void)hipStreamSynchronize(c->st);
  (void)hipFree(ws);
  return rc;
}

int se_attention(se_ctx* c, void* stream, const float* x, const float* mask_full, float* out, float* similar_out, int B,
                 int h, int w) {
  return se_attention_ex(c, stream, x, mask_full, out, similar_out, B, h, w, 0);
}

}  // extern "C"
