"""ORACLE -- test infrastructure, NOT product code.

CPU restatement of the SketchEdit inference forward pass (mask predictor netM +
two-stage inpainting generator netG + contextual attention), written from the
reference's behaviour with plain torch CPU ops.  Only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg may import this module; nothing under
sketchedit_amd/ does.

Parity pinning: the reference ships no tests / golden vectors for this path
(SURVEY.md section 4), so this restatement is pinned against outputs of the reference
itself, imported in the build container with procedural weights
(tests/golden/make_golden.py -> tests/golden/*.npz, checked by
tests/test_oracle_golden.py).

Every function cites the reference lines it restates (paths relative to
/root/reference).  Weights are passed as a dict name -> tensor with the reference's
checkpoint keys ('<layer>.weight' OIHW, '<layer>.bias').

bf16 mode (BASELINE config 5: "bf16 MFMA path with fp32 accumulate").  The reference has no reduced-precision mode,
so the comparator is DEFINED here and pinned against the reference run with the same roundings injected through
module hooks (tests/golden/make_golden.py, e2e_64_bf16.npz): with `act_dtype=torch.bfloat16`
  * every conv weight is rounded to bf16 (biases stay fp32: they are added in the fp32 epilogue),
  * every tensor a conv reads and every gated-conv output is rounded to bf16 (round-to-nearest-even),
  * inside a layer everything is fp32: products of bf16 values are exact in fp32 and the accumulation, bias, ELU,
    sigmoid, tanh and the composites run in fp32,
  * attention: keys x*rsqrt(.) rounded to bf16, scores / softmax in fp32, probabilities rounded to bf16, P.V
    accumulated in fp32, result rounded to bf16.
The roundings are points, not a bit-exact model of any kernel's summation order: parity against this mode is stated
with a bf16-noise tolerance (tests/test_oracle_golden.py derives it from two equally valid rounding placements).
"""
import torch
import torch.nn.functional as F


def _t(v):
    return v if isinstance(v, torch.Tensor) else torch.from_numpy(v)


def _r(x, dt):
    """Round to the storage dtype of the reduced-precision mode (no-op in fp32 mode)."""
    return x if dt is None else x.to(dt).to(torch.float32)


def gated_conv(x, w, b, stride=1, rate=1, act="elu", dt=None):
    """models/networks/utils.py:9-33 (gen_conv).

    conv2d with zero padding rate*(k-1)/2 and dilation `rate`; when Cout==3 or act is
    None the raw conv is returned, else the channels are split in half and
    act(first half) * sigmoid(second half).  act: 'elu' (nn.ELU, alpha 1), 'relu', None.
    """
    k = w.shape[-1]
    pad = int(rate * (k - 1) / 2)
    y = F.conv2d(_r(x, dt), _r(w, dt), b, stride=stride, padding=pad, dilation=rate)
    cout = w.shape[0]
    if cout == 3 or act is None:
        return y
    f, g = y[:, : cout // 2], y[:, cout // 2:]
    f = F.elu(f) if act == "elu" else torch.relu(f)
    return _r(f * torch.sigmoid(g), dt)


def gated_deconv(x, w, b, dt=None):
    """models/networks/utils.py:35-51 (gen_deconv): nearest x2 upsample then gen_conv(k=3)."""
    x = x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)  # out[i,j] = in[i//2, j//2]
    return gated_conv(x, w, b, 1, 1, "elu", dt)


_DT = [None]      # storage dtype of the forward being evaluated (set by netM_forward / netG_forward)


def _L(W, x, name, stride=1, rate=1, act="elu"):
    return gated_conv(x, _t(W[name + ".weight"]), _t(W[name + ".bias"]), stride, rate, act, _DT[0])


def _D(W, x, name):
    return gated_deconv(x, _t(W[name + ".weight"]), _t(W[name + ".bias"]), _DT[0])


def _encoder(W, x, p, c3=True):
    """The 10-layer gated encoder shared by netM / coarse / style branches
    (editline_g.py:138-147, editline2_g.py:67-76)."""
    x = _L(W, x, p + "1")
    x = _L(W, x, p + "2_downsample", stride=2)
    x = _L(W, x, p + "3")
    x = _L(W, x, p + "4_downsample", stride=2)
    x = _L(W, x, p + "5")
    x = _L(W, x, p + "6")
    x = _L(W, x, p + "7_atrous", rate=2)
    x = _L(W, x, p + "8_atrous", rate=4)
    x9 = _L(W, x, p + "9_atrous", rate=8)
    x10 = _L(W, x9, p + "10_atrous", rate=16)
    return x9, x10


def _decoder(W, x, p, final_act):
    """7-layer decoder: conv11 conv12 up13 conv14 up15 conv16 conv17 (editline_g.py:169-176)."""
    x = _L(W, x, p + "11")
    x = _L(W, x, p + "12")
    x = _D(W, x, p + "13_upsample_conv")
    x = _L(W, x, p + "14")
    x = _D(W, x, p + "15_upsample_conv")
    x = _L(W, x, p + "16")
    x = _L(W, x, p + "17", act=None)
    return torch.tanh(x) if final_act == "tanh" else torch.sigmoid(x)


def netM_forward(W, x, guide, want_image=True, act_dtype=None):
    """models/networks/editline2_g.py:59-94 (MDGenerator.forward) -> (mask, mask_image).

    Quirk kept: the image decoder consumes conv9's output, the mask decoder conv10's
    (editline2_g.py:76-77).
    """
    x, guide = _t(x), _t(guide)
    _DT[0] = act_dtype
    try:
        xin = torch.cat([x, guide], 1)
        x9, x10 = _encoder(W, xin, "conv")
        img = _decoder(W, x9, "conv", "tanh") if want_image else None
        mask = _decoder(W, x10, "conv_mask_", "sigmoid")
    finally:
        _DT[0] = None
    return mask, img


def attention_scores(x, mask_s, scale=10.0, th=0.1, dt=None):
    """models/networks/splitcam.py:37-108 (ReduceContextAttentionP1, patch 4, stride 2, pd 0,
    is_th, norm_type 1) with f = b = x.  Returns softmax scores (B, L_keys, hs, ws)."""
    B, C, h, w = x.shape
    valid = 1.0 - mask_s
    xn = _r(x / torch.sqrt((x * x).sum(3, keepdim=True).sum(2, keepdim=True) + 1e-8), dt)  # :40
    K = F.unfold(xn, 4, stride=2)                     # (B, C*16, L)  :42
    Q = F.unfold(x, 4, stride=2)                      # queries: raw patches (batch_conv2d :69)
    mk = F.unfold(valid, 4, stride=2)                 # (B, 16, L)    :49-53
    mm = mk.view(B, 4, 4, -1).mean(2).mean(1)         # mean over kx then ky -> (B, L)
    S = torch.einsum("bdj,bdi->bji", K, Q)            # S[b, key j, query i]
    S = S * (mm > th).float()[:, :, None]             # :90,104 multiplicative zero
    P = torch.softmax(S * scale, dim=1)               # :105
    hs, ws = (h - 4) // 2 + 1, (w - 4) // 2 + 1
    return P.view(B, -1, hs, ws)


def attention_reconstruct(P, x, dt=None):
    """models/networks/splitcam.py:132-153 (ReduceContextAttentionP2, mk=False, pd 0):
    transposed conv with the raw patches as kernels == P^T V then overlap-add, no
    normalisation by the overlap count."""
    B, C, h, w = x.shape
    V = F.unfold(x, 4, stride=2)                      # (B, C*16, Lk)
    Pm = _r(P, dt).reshape(B, P.shape[1], -1)         # (B, Lk, Lq)
    O = torch.bmm(V, Pm)                              # (B, C*16, Lq)
    return _r(F.fold(O, (h, w), 4, stride=2), dt)


def contextual_attention(x, mask_full, dt=None):
    """editline_g.py:203-207."""
    mask_s = F.avg_pool2d(mask_full, 4, 4)
    P = attention_scores(x, mask_s, dt=dt)
    return attention_reconstruct(P, x, dt), P


def netG_forward(W, x, x2, mask, mask2, guide, use_cam=True, pool_type="max",
                 no_mask_cc=False, no_mask_coarse=False, joint_train_inp=True, taps=None, act_dtype=None):
    _DT[0] = act_dtype
    try:
        return _netG_forward(W, x, x2, mask, mask2, guide, use_cam, pool_type, no_mask_cc, no_mask_coarse,
                             joint_train_inp, taps, act_dtype)
    finally:
        _DT[0] = None


def _netG_forward(W, x, x2, mask, mask2, guide, use_cam, pool_type, no_mask_cc, no_mask_coarse, joint_train_inp, taps, dt):
    """models/networks/editline_g.py:119-221 (DeepFillC2Generator.forward) -> (coarse, fine).

    `taps` (optional dict) receives intermediates for localising a mismatch.
    """
    x, x2, mask, mask2, guide = map(_t, (x, x2, mask, mask2, guide))
    if not no_mask_cc:
        x2 = x2 * mask2                                              # :123
    x = x * (1 - mask)                                               # :124
    xin = x
    x = torch.cat([x, guide, mask], 1)                               # :131
    g2 = guide * 0 if joint_train_inp else guide                     # :132-135
    x2 = torch.cat([x2, g2, mask2], 1)
    _, xc = _encoder(W, x, "conv")                                   # :138-147
    _, xs = _encoder(W, x2, "wconv")                                 # :149-158
    hs, ws = xs.shape[2:]
    if pool_type == "avg":
        sv = xs.mean(3).mean(2)[..., None, None]                     # :161
    elif pool_type == "max":
        sv = xs.amax(dim=(2, 3), keepdim=True)                       # :163 max_pool2d over whole map
    else:
        raise NotImplementedError
    if taps is not None:
        taps["coarse_enc"] = xc
        taps["style_vec"] = sv[:, :, 0, 0]
    xc = torch.cat((xc, sv.expand(-1, -1, hs, ws)), 1)               # :166-167
    stage1 = _decoder(W, xc, "conv", "tanh")                         # :169-177
    x = stage1
    if not no_mask_coarse:
        x = x * mask + xin[:, 0:3] * (1.0 - mask)                    # :179-180
    xnow = x
    # hallucination branch :184-194
    h = _L(W, xnow, "xconv1")
    h = _L(W, h, "xconv2_downsample", stride=2)
    h = _L(W, h, "xconv3")
    h = _L(W, h, "xconv4_downsample", stride=2)
    h = _L(W, h, "xconv5")
    h = _L(W, h, "xconv6")
    h = _L(W, h, "xconv7_atrous", rate=2)
    h = _L(W, h, "xconv8_atrous", rate=4)
    h = _L(W, h, "xconv9_atrous", rate=8)
    h = _L(W, h, "xconv10_atrous", rate=16)
    # patch-match branch :197-209
    p = _L(W, xnow, "pmconv1")
    p = _L(W, p, "pmconv2_downsample", stride=2)
    p = _L(W, p, "pmconv3")
    p = _L(W, p, "pmconv4_downsample", stride=2)
    p = _L(W, p, "pmconv5")
    p = _L(W, p, "pmconv6", act="relu")
    if taps is not None:
        taps["pmconv6"] = p
    if use_cam:
        p, P = contextual_attention(p, mask, dt)
        if taps is not None:
            taps["similar"] = P
            taps["attn_out"] = p
    p = _L(W, p, "pmconv9")
    p = _L(W, p, "pmconv10")
    x = torch.cat([h, p], 1)                                         # :211
    stage2 = _decoder(W, x, "allconv", "tanh")                       # :213-220
    return stage1, stage2


def inference(WM, WG, image, sketch, act_dtype=None, **flags):
    """models/editline2_model.py:128-133 + generate_fake :338-370 (eval mode).

    Returns dict(composed, mask, hard_mask, coarse, fine).  composed uses the SOFT mask
    (:132); netG receives the thresholded mask (:346-347) for both mask arguments (:366).
    """
    image, sketch = _t(image), _t(sketch)
    with torch.no_grad():
        mask, mask_image = netM_forward(WM, image, sketch, act_dtype=act_dtype)
        hard = (mask > 0.5).float()
        coarse, fine = netG_forward(WG, image, image, hard, hard, sketch, act_dtype=act_dtype, **flags)
        composed = fine * mask + image * (1 - mask)
    return dict(composed=composed, mask=mask, mask_image=mask_image, hard_mask=hard, coarse=coarse, fine=fine)
