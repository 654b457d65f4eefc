/* libsketchedit_hip.so -- C-ABI of the MI355X (gfx950) SketchEdit inference path.
 *
 * The reference (zengxianyu/sketchedit) has no FFI: its de-facto boundary is Python,
 *   models.create_model(opt)(data, mode='inference')     models/editline2_model.py:107-133
 *   netM(x, guide) -> (mask, mask_image)                 models/networks/editline2_g.py:59-94
 *   netG(x, x2, mask, mask2, guide) -> (coarse, fine)    models/networks/editline_g.py:119-221
 * and below it torch.nn.functional.  This header is what a ctypes binding on the reference side
 * binds instead of that arithmetic; sketchedit_amd/_lib.py is that binding.
 *
 * Conventions
 *  - every tensor pointer is a DEVICE pointer owned by the caller (e.g. torch tensor.data_ptr()),
 *    fp32, contiguous NCHW exactly as the reference's tensors; weights are HOST pointers.
 *  - `stream` is a hipStream_t (torch.cuda.current_stream().cuda_stream); calls only enqueue work,
 *    there is no hidden device synchronisation and no allocation inside a forward: the caller
 *    passes a workspace of at least se_workspace_bytes(ctx, B, H, W) bytes.
 *  - return value 0 = ok, non-zero = error (se_last_error(ctx) describes it); no C++ exception
 *    crosses the boundary.  A context serialises its own forwards with an internal mutex, so one
 *    ctx may be shared by threads (demo.py:120 runs Flask threaded) -- one ctx per stream is faster.
 *  - H and W must be multiples of 8 (demo.py:43-45 enforces the same for the reference).
 *  - Any batch size: the kernels address one tensor with 32-bit byte offsets, so a forward over more images than fit
 *    2^31 bytes of its largest activation (96 bytes per pixel in fp32: 341 images at 256x256) runs as several passes of
 *    the same plan over image ranges; an image's result is bit-identical whatever pass (or batch) it is in.  A single
 *    image beyond that range, and a per-op call beyond it, is an error -- never a silently wrong result.
 */
#ifndef SKETCHEDIT_HIP_H
#define SKETCHEDIT_HIP_H
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct se_ctx se_ctx;

enum { SE_NET_G = 0, SE_NET_M = 1 };

/* netG option flags: models/networks/editline_g.py:15-23, options/base_options.py:19 */
enum {
  SE_FLAG_USE_CAM = 1,         /* --use_cam */
  SE_FLAG_POOL_MAX = 2,        /* --pool_type max (unset: avg) */
  SE_FLAG_NO_MASK_CC = 4,      /* --no_mask_cc */
  SE_FLAG_NO_MASK_COARSE = 8,  /* --no_mask_coarse */
  SE_FLAG_JOINT_TRAIN_INP = 16, /* --joint_train_inp */
  /* execution options (no reference counterpart; the results agree with the default mode to fp32 rounding, and an
   * image's result is bit-identical across batch positions / ranks WITHIN one mode):
   * LOW_LATENCY: for one or two images per call (test_celeb.sh:2 --batchSize 1, demo.py:59): small-grid kernel shapes that
   * put every CU to work and the independent branches of netG on two streams (the ctx's side stream is ordered after /
   * before the caller's stream through events -- no host synchronisation).
   * GRAPH (se_inference only): capture the forward for these exact arguments into a hipGraph on its second use and
   * replay it afterwards; the caller keeps every pointer argument (inputs, outputs, workspace) stable. */
  SE_FLAG_LOW_LATENCY = 32,
  SE_FLAG_GRAPH = 64,
  /* PACKED_OUT (se_inference only): composed_out points at ONE (B,4,H,W) buffer -- planes 0-2 the composite, plane 3
   * the soft mask (mask_out is ignored and may be NULL): the unit a batch-sharded caller all-gathers. */
  SE_FLAG_PACKED_OUT = 128,
  /* BF16 (BASELINE config 5): activations and conv weights are stored as bf16 and multiplied on the bf16 matrix
   * pipe (v_mfma_f32_16x16x32_bf16) with fp32 accumulation; bias, activations, gate, softmax and composites stay
   * fp32; the external tensors stay fp32 NCHW.  The comparator is the oracle's bf16 mode (oracle/sketchedit_oracle.py);
   * the 1e-3 fp32 bound of the north star does not apply (stated tolerances: tests/test_gpu_bf16.py). */
  SE_FLAG_BF16 = 256,
  /* CONSERVATIVE (fp32 mode; no reference counterpart): the precision choice for netM, whose soft mask feeds the hard 0.5
   * threshold (editline2_model.py:346-347).  By default the 96->192 3x3 layers of BOTH nets run the hybrid Winograd
   * F(2,3)xF(4,3) form; with this flag netM's run F(2x2,3x3) (dyadic transforms, error like the direct form) and only netG
   * keeps the hybrid.  Measured over 72 images / 4.7 M mask pixels of the three procedural weight sets
   * (profiles/r05_f43_flips.json): hard-mask pixels that differ from the fp32 CPU reference 4 (default) vs 1 (this flag;
   * the direct form also has 1), soft-mask max-abs 2.6e-5 vs 1.6e-5, for +1.7 % per step (11.10 -> 11.29 ms at 256x256
   * batch 32).  A caller with a real checkpoint whose logits cluster at the threshold sets it; honoured by se_inference,
   * se_inference_u8, se_netM_forward_ex.  INTEGRATION.md "Precision choice". */
  SE_FLAG_CONSERVATIVE = 512
};

/* replaces networks.create_network's .cuda() (models/networks/__init__.py:30-38) */
int se_create(int device_id, se_ctx** out);
void se_destroy(se_ctx* ctx);
const char* se_last_error(se_ctx* ctx);
const char* se_version(void);

/* replaces util.load_network / load_state_dict (util/util.py:214-225).  `name` is a state_dict key
 * ("conv1.weight", "conv1.bias", a leading "module." is stripped); `host` points at shape[0..ndim)
 * fp32 values in the checkpoint's own layout (weight OIHW, bias O).  Unknown keys or wrong shapes
 * are errors (strict load).  se_weights_ready() reports 1 when every tensor of both nets is set. */
int se_load_weights(se_ctx* ctx, int net_id, const char* name, const float* host, const int* shape, int ndim);
int se_weights_ready(se_ctx* ctx);

size_t se_workspace_bytes(se_ctx* ctx, int B, int H, int W);

/* MDGenerator.forward (editline2_g.py:59-94).  image (B,3,H,W), sketch (B,1,H,W) ->
 * mask_out (B,1,H,W); maskim_out (B,3,H,W) may be NULL (its decoder is then skipped, as in
 * mode='inference' where the value is unused). */
int se_netM_forward(se_ctx* ctx, void* stream, const float* image, const float* sketch, float* mask_out,
                    float* maskim_out, void* workspace, size_t workspace_bytes, int B, int H, int W);

/* the same with execution options (SE_FLAG_LOW_LATENCY, SE_FLAG_BF16, SE_FLAG_CONSERVATIVE) */
int se_netM_forward_ex(se_ctx* ctx, void* stream, const float* image, const float* sketch, float* mask_out,
                       float* maskim_out, void* workspace, size_t workspace_bytes, int B, int H, int W, int exec_flags);

/* DeepFillC2Generator.forward (editline_g.py:119-221).  x,x2 (B,3,H,W); mask,mask2,guide (B,1,H,W);
 * coarse_out / fine_out (B,3,H,W), coarse_out may be NULL. */
int se_netG_forward(se_ctx* ctx, void* stream, const float* x, const float* x2, const float* mask,
                    const float* mask2, const float* guide, float* coarse_out, float* fine_out, void* workspace,
                    size_t workspace_bytes, int B, int H, int W, int flags);

/* netG with optional intermediate outputs ("taps"; test support -- the reference's counterparts are forward hooks on
 * netG.pmconv6 / netG.cam_2 / netG.conv11, tests/golden/make_golden.py).  Any pointer may be NULL; taps == NULL is
 * se_netG_forward.  The tensors are written by the SAME launches that feed the next layer of the production plan (the
 * attention runs in whatever form the forward would use), so a test sees what the forward computed, in the reference's
 * NCHW fp32 layout.  One pass only: B must fit the 32-bit offset range (341 images at 256x256). */
typedef struct se_netG_taps {
  float* pmconv6;   /* (B,96,H/4,W/4)  output of pmconv6 = input of the attention   editline_g.py:202 */
  float* attn_out;  /* (B,96,H/4,W/4)  output of cam_2                              editline_g.py:203-207 */
  float* style_vec; /* (B,96)          pooled style vector fed to conv11            editline_g.py:159-167 */
} se_netG_taps;
int se_netG_forward_taps(se_ctx* ctx, void* stream, const float* x, const float* x2, const float* mask,
                         const float* mask2, const float* guide, float* coarse_out, float* fine_out, void* workspace,
                         size_t workspace_bytes, int B, int H, int W, int flags, const se_netG_taps* taps);

/* EditLine2Model.forward(mode='inference') (editline2_model.py:128-133 + generate_fake :338-370):
 * netM -> (mask > 0.5) -> netG -> composed = fine*mask + image*(1-mask) with the SOFT mask.
 * Optional outputs (may be NULL): hard_out (B,1,H,W), maskim_out, coarse_out, fine_out (B,3,H,W)
 * -- the extra tensors of mode='visualize' (:134-145). */
int se_inference(se_ctx* ctx, void* stream, const float* image, const float* sketch, float* composed_out,
                 float* mask_out, float* hard_out, float* maskim_out, float* coarse_out, float* fine_out,
                 void* workspace, size_t workspace_bytes, int B, int H, int W, int flags);

/* se_inference with the output quantisation of test.py:25-27 fused into its last kernel: rgb_out (B,H,W,3) uint8 =
 * trunc((composed + 1) / 2 * 255) in the HWC order test.py:35 transposes to, mask_u8_out (B,H,W) uint8 = trunc(mask * 255)
 * (may be NULL); same fp32 operation order as the reference's tensor expressions, no clamp.  No fp32 output tensor is
 * written at all.  flags as se_inference (GRAPH and PACKED_OUT are ignored). */
int se_inference_u8(se_ctx* ctx, void* stream, const float* image, const float* sketch, unsigned char* rgb_out,
                    unsigned char* mask_u8_out, void* workspace, size_t workspace_bytes, int B, int H, int W, int flags);

/* The INPUT side of test.py's loop on the device (data/testimage_dataset.py:89-111 = ToTensor + Normalize(0.5, 0.5) and
 * `sketch > 0`): image_u8 (B,H,W,3) uint8 RGB as the PNG decoder delivers it -> image_out (B,3,H,W) fp32 = (v/255 - 0.5)/0.5;
 * sketch_u8 (B,H,W) uint8 ('L') -> sketch_out (B,1,H,W) fp32 in {0,1}.  The 256 possible image values are tabulated on the
 * host in IEEE fp32 in that operation order and looked up on the device: bit-identical to the tensors the dataset builds on
 * the CPU.  Either pair may be NULL.  A 4x smaller host-to-device copy for the caller. */
int se_dequantize_u8(se_ctx* ctx, void* stream, const unsigned char* image_u8, const unsigned char* sketch_u8, float* image_out,
                     float* sketch_out, int B, int H, int W);

/* se_inference_u8 fed with the uint8 arrays above: uint8 in, uint8 out -- the whole body of test.py:20-37 between the PNG
 * decoder and the PNG encoder as one call; the fp32 inputs live in the workspace (se_workspace_bytes includes them). */
int se_inference_u8io(se_ctx* ctx, void* stream, const unsigned char* image_u8, const unsigned char* sketch_u8,
                      unsigned char* rgb_out, unsigned char* mask_u8_out, void* workspace, size_t workspace_bytes, int B, int H,
                      int W, int flags);

/* Output quantisation of test.py:25-27 on the device: rgb_out (B,H,W,3) uint8 = trunc((composed + 1) / 2 * 255)
 * in the HWC order test.py:35 transposes to, mask_u8_out (B,H,W) uint8 = trunc(mask * 255); same fp32 operation
 * order as the reference's tensor expressions, no clamp (as test.py; demo.py:62 clamps -- a [-1,1] input cannot
 * leave [0,255] either way).  Either output may be NULL.  A 4x smaller device-to-host copy for the caller. */
int se_quantize_u8(se_ctx* ctx, void* stream, const float* composed, const float* mask, unsigned char* rgb_out,
                   unsigned char* mask_u8_out, int B, int H, int W);

/* ---- measurement support (no reference counterpart; used by bench.py) ----------------------------
 * se_profile_enable(ctx, 1): wrap every kernel launch of subsequent forwards in a pair of HIP events
 * recorded on the launch stream; se_profile_report synchronises the device and writes a JSON array
 *   [{"kernel": name, "launches": n, "total_ms": t, "flops": algorithmic FLOPs, "bytes": ...}, ...]
 * aggregated per kernel.  se_profile_enable(ctx, 0) switches it off and drops the records. */
int se_profile_enable(se_ctx* ctx, int on);
int se_profile_report(se_ctx* ctx, char* buf, size_t cap);

/* ---- developer switches (no reference counterpart; used by the tests and the A/B tools) -----------
 * The library's kernel-form switches ("SE_WINOGRAD_F43", "SE_ATT_FUSED", ...) live in ONE process-wide
 * table that is filled from the environment once, at first use; no forward ever calls getenv.  se_debug_set_option
 * changes an entry for every later call of the process (name with or without the "SE_" prefix; returns 0, or 1 for an
 * unknown name), se_debug_get_option reads one, se_debug_reset_options restores the environment / built-in values.
 * "SE_TEST_OFFSET_LIMIT" (settable only here) lowers the byte range the 32-bit-offset kernels may address, so that a
 * test reaches the large-batch passes of the forwards with a few small images. */
int se_debug_set_option(const char* name, int value);
int se_debug_get_option(const char* name, int* value);
void se_debug_reset_options(void);

/* ---- per-op entry points (unit tests; same kernels as the forwards) ----------------------------
 * gen_conv / gen_deconv (models/networks/utils.py:9-51): x (B,Cin,H,W) device, w (Cout,Cin,k,k) and
 * b (Cout) HOST, y device (B, Cout/2 or Cout, Ho, Wo).  act: 0 ELU, 1 ReLU, 2 None (raw conv).
 * Supported: gated Cout%8==0 (any Cin, k in {3,5}); raw only for k=3, Cin=12, Cout in {1,3}. */
int se_gated_conv2d(se_ctx* ctx, void* stream, const float* x, const float* w_host, const float* b_host, float* y,
                    int B, int Cin, int H, int W, int Cout, int k, int stride, int rate, int act, int upsample);
/* The same with the options the forwards use: a second source x1 of the virtual channel concat in front of conv11 /
 * allconv11 (editline_g.py:166-167,211) -- a tensor (B,Cin1,H,W), or with x1_is_vector a spatially constant per-image
 * vector (B,Cin1), still zero padded at the borders; w is (Cout, Cin+Cin1, k, k) -- and exec_flags: SE_FLAG_LOW_LATENCY
 * runs the layer in its small-grid launch shape, SE_FLAG_BF16 on the bf16 path (x, x1 and w are rounded to bf16 on the
 * way in, y is the bf16 result widened to fp32).  x1 may be NULL. */
int se_gated_conv2d_ex(se_ctx* ctx, void* stream, const float* x, const float* x1, int x1_is_vector, const float* w_host,
                       const float* b_host, float* y, int B, int Cin, int Cin1, int H, int W, int Cout, int k, int stride,
                       int rate, int act, int upsample, int exec_flags);
/* cam_1 + cam_2 (models/networks/splitcam.py:57-108,147-174 as configured at editline_g.py:35-42,
 * 203-207): x (B,96,h,w), mask_full (B,1,4h,4w) -> out (B,96,h,w); similar_out (B,L,hs,ws) may be NULL. */
int se_attention(se_ctx* ctx, void* stream, const float* x, const float* mask_full, float* out, float* similar_out,
                 int B, int h, int w);
/* the same with exec_flags (SE_FLAG_BF16: x is rounded to bf16 on the way in, out is the bf16 result widened to fp32) */
int se_attention_ex(se_ctx* ctx, void* stream, const float* x, const float* mask_full, float* out, float* similar_out,
                    int B, int h, int w, int exec_flags);

#ifdef __cplusplus
}
#endif
#endif
