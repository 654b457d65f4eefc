#!/usr/bin/env python3
"""Generate the golden fixtures in this directory by running the REFERENCE itself.

Runs only in the build container, where /root/reference exists.  The reference is
imported read-only (sys.dont_write_bytecode), with a stub `cv2` module because
models/create_mask.py:1 imports cv2 (train-only, absent here).  Procedural weights
from sketchedit_amd.synth are injected with load_state_dict.  Only the produced
vectors (inputs are regenerated from the seed) are committed -- no reference source.

    python tests/golden/make_golden.py              # everything
    python tests/golden/make_golden.py --round4     # only the fixtures added in round 4 (512x512, samples, weight sets)
    python tests/golden/make_golden.py --round6     # only the fixture added in round 6 (attention key-validity boundary)
"""
import os
import sys
import types
from argparse import Namespace

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "..", ".."))
sys.dont_write_bytecode = True
sys.path.insert(0, "/root/reference")
sys.modules["cv2"] = types.ModuleType("cv2")

import torch  # noqa: E402
import torch.nn.functional as F  # noqa: E402

from sketchedit_amd import synth  # noqa: E402

torch.set_num_threads(8)


def build_reference(gain, seed=0, dist="uniform", **over):
    import models  # the reference's models/__init__.py
    o = dict(gpu_ids=[], isTrain=False, model="editline2", netG="deepfillc2",
             init_type="xavier", init_variance=0.02, use_cam=True, pool_type="max",
             no_mask_cc=False, no_mask_coarse=False, joint_train_inp=True,
             continue_train=False, isSkip=True, which_epoch="latest",
             checkpoints_dir="./checkpoints", name="celeb")
    o.update(over)
    m = models.create_model(Namespace(**o)).eval()
    sdG = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", seed, gain, dist).items()}
    sdM = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", seed, gain, dist).items()}
    m.netG.load_state_dict(sdG)      # strict: key/shape contract check
    m.netM.load_state_dict(sdM)
    return m


def run_case(m, B, H, W, seed):
    img, sk = synth.make_inputs(B, H, W, seed=seed)
    img, sk = torch.from_numpy(img), torch.from_numpy(sk)
    out = {}
    with torch.no_grad():
        composed, soft = m({"image": img, "mask": sk, "gt": img, "edgegt": sk}, mode="inference")
        mask, mask_image = m.netM(img, sk)
        hard = (mask > 0.5).float()
        # intermediates of netG via forward hooks on the reference modules
        taps = {}
        hooks = [
            m.netG.conv10_atrous.register_forward_hook(lambda mod, i, o: taps.__setitem__("coarse_enc", o)),
            m.netG.pmconv6.register_forward_hook(lambda mod, i, o: taps.__setitem__("pmconv6", o)),
            m.netG.cam_1.register_forward_hook(lambda mod, i, o: taps.__setitem__("similar", o)),
            m.netG.cam_2.register_forward_hook(lambda mod, i, o: taps.__setitem__("attn_out", o[0])),
            m.netG.conv11.register_forward_hook(lambda mod, i, o: taps.__setitem__("conv11_in", i[0])),
            m.netM.conv1.register_forward_hook(lambda mod, i, o: taps.__setitem__("m_conv1", o)),
            m.netM.conv10_atrous.register_forward_hook(lambda mod, i, o: taps.__setitem__("m_conv10", o)),
        ]
        coarse, fine = m.netG(img, img, hard, hard, sk)
        m.netM(img, sk)
        for h in hooks:
            h.remove()
    assert torch.equal(soft, mask)
    out.update(composed=composed, mask=mask, mask_image=mask_image, hard_mask=hard, coarse=coarse, fine=fine)
    out["style_vec"] = taps["conv11_in"][:, 96:, 0, 0]
    for k in ("coarse_enc", "pmconv6", "similar", "attn_out", "m_conv1", "m_conv10"):
        if k in taps:
            out[k] = taps[k]
    return {k: v.numpy() for k, v in out.items()}


def _bf16(t):
    return t.to(torch.bfloat16).to(torch.float32)


def run_case_bf16(gain, B, H, W, seed):
    """BASELINE config 5 comparator: the REFERENCE forward with bf16 roundings injected from outside -- conv weights
    rounded to bf16 in the state dict, a forward pre-hook on every gen_conv / gen_deconv that rounds the tensor it
    reads, a forward hook that rounds every gated output (raw 12->3 / 12->1 outputs stay fp32) and the attention
    output.  The reference cannot be made to round INSIDE its attention (keys, probabilities): the oracle's bf16 mode
    does, and tests/test_oracle_golden.py bounds the difference."""
    m = build_reference(gain)
    from models.networks.utils import gen_conv
    for net in (m.netG, m.netM):
        sd = net.state_dict()
        net.load_state_dict({k: (_bf16(v) if k.endswith(".weight") else v) for k, v in sd.items()})
    hooks = []
    for net in (m.netG, m.netM):
        for mod in net.modules():
            if isinstance(mod, gen_conv):
                hooks.append(mod.register_forward_pre_hook(lambda mod, inp: (_bf16(inp[0]),)))
                gated = not (mod.out_channels == 3 or mod.activation is None)
                if gated:
                    hooks.append(mod.register_forward_hook(lambda mod, inp, out: _bf16(out)))
    hooks.append(m.netG.cam_2.register_forward_hook(lambda mod, inp, out: (_bf16(out[0]),) + tuple(out[1:])))
    img, sk = synth.make_inputs(B, H, W, seed=seed)
    img, sk = torch.from_numpy(img), torch.from_numpy(sk)
    with torch.no_grad():
        composed, soft = m({"image": img, "mask": sk, "gt": img, "edgegt": sk}, mode="inference")
        mask, mask_image = m.netM(img, sk)
        hard = (mask > 0.5).float()
        coarse, fine = m.netG(img, img, hard, hard, sk)
    for h in hooks:
        h.remove()
    assert torch.equal(soft, mask)
    return {k: v.numpy().astype(np.float32) for k, v in dict(composed=composed, mask=mask, mask_image=mask_image,
                                                             hard_mask=hard, coarse=coarse, fine=fine).items()}


def summary(a):
    a = a.astype(np.float64)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.min(), a.max()], np.float64)


def op_cases():
    """Per-op known answers from the reference's own op modules (utils.py, splitcam.py)."""
    from models.networks.utils import gen_conv, gen_deconv
    from models.networks.splitcam import ReduceContextAttentionP1, ReduceContextAttentionP2
    import torch.nn as nn
    out = {}
    cases = [  # name, cin, cout, k, stride, rate, act, H, W
        ("c3_s1_d1_elu", 8, 16, 3, 1, 1, "elu", 12, 16),
        ("c3_s2_d1_elu", 8, 16, 3, 2, 1, "elu", 12, 16),
        ("c3_s1_d2_elu", 8, 16, 3, 1, 2, "elu", 12, 16),
        ("c3_s1_d16_elu", 8, 16, 3, 1, 16, "elu", 20, 24),
        ("c3_s1_d1_relu", 8, 16, 3, 1, 1, "relu", 12, 16),
        ("c3_s1_d1_none", 12, 1, 3, 1, 1, None, 12, 16),
        ("c3_s1_d1_rgb", 12, 3, 3, 1, 1, "elu", 12, 16),   # Cout==3 => passthrough even with act
        ("c5_s1_d1_elu", 5, 16, 5, 1, 1, "elu", 12, 16),
    ]
    acts = {"elu": nn.ELU(), "relu": nn.ReLU(), None: None}
    for name, cin, cout, k, s, r, act, H, W in cases:
        w = synth.uniform(7, name + ".w", (cout, cin, k, k), -0.5, 0.5)
        b = synth.uniform(7, name + ".b", (cout,), -0.5, 0.5)
        x = synth.uniform(7, name + ".x", (2, cin, H, W), -1, 1)
        mod = gen_conv(cin, cout, k, s, r, activation=acts[act])
        mod.load_state_dict({"weight": torch.from_numpy(w), "bias": torch.from_numpy(b)})
        with torch.no_grad():
            out["op." + name] = mod(torch.from_numpy(x)).numpy()
    w = synth.uniform(7, "deconv.w", (16, 8, 3, 3), -0.5, 0.5)
    b = synth.uniform(7, "deconv.b", (16,), -0.5, 0.5)
    x = synth.uniform(7, "deconv.x", (2, 8, 6, 8), -1, 1)
    mod = gen_deconv(8, 16)
    mod.load_state_dict({"weight": torch.from_numpy(w), "bias": torch.from_numpy(b)})
    with torch.no_grad():
        out["op.deconv"] = mod(torch.from_numpy(x)).numpy()
    # attention, same ctor args as editline_g.py:35-42
    cam1 = ReduceContextAttentionP1(nn_hard=False, ufstride=2, stride=2, bkg_patch_size=4, pd=0,
                                    is_th=True, th=0.1, norm_type=1)
    cam2 = ReduceContextAttentionP2(ufstride=2, bkg_patch_size=4, stride=2, pd=0, mk=False)
    x = torch.from_numpy(synth.uniform(7, "att.x", (2, 8, 12, 16), -1, 1))
    full = (torch.from_numpy(synth.uniform(7, "att.m", (2, 1, 48, 64), 0, 1)) < 0.6).float()
    full[1, :, :24] = 1.0                        # a big hole: some key patches fully invalid
    full[0, :, 0:16, 0:16] = 1.0
    full[0, :, 0:14, 3:16] = 1.0
    ms = F.avg_pool2d(full, 4, 4)
    with torch.no_grad():
        sim = cam1(x, x, ms)
        rec, _ = cam2(sim, x, ms, {})
    out["op.att.similar"] = sim.numpy()
    out["op.att.out"] = rec.numpy()
    # all keys invalid -> uniform softmax
    ones = torch.ones(2, 1, 12, 16)
    with torch.no_grad():
        sim = cam1(x, x, ones)
        rec, _ = cam2(sim, x, ones, {})
    out["op.att_allinvalid.similar"] = sim.numpy()
    out["op.att_allinvalid.out"] = rec.numpy()
    return out


FACE = "/root/reference/datasets/face_release/%s/602_images_celeb_00033.png"


def face_case(m):
    """BASELINE config 1 (test_celeb.sh): the bundled 256x256 face and its sketch, batch 1, through the reference.
    Inputs as /root/reference/data/testimage_dataset.py:89-111 builds them (ToTensor + Normalize(0.5, 0.5); sketch 'L',
    > 0); outputs also as test.py:25-27 quantises them.  The fixture carries the two input images as uint8 arrays (data,
    not code) so that the GPU box needs nothing from /root/reference."""
    from PIL import Image
    image = Image.open(FACE % "images").convert("RGB")
    w, h = image.size
    sketch = Image.open(FACE % "edges").convert("L").resize((w, h))
    iu8, su8 = np.asarray(image, np.uint8), np.asarray(sketch, np.uint8)
    img = torch.from_numpy(((iu8.astype(np.float32).transpose(2, 0, 1) / 255.0) - 0.5) / 0.5)[None]
    sk = torch.from_numpy((su8.astype(np.float32)[None, None] / 255.0 > 0).astype(np.float32))
    with torch.no_grad():
        composed, soft = m({"image": img, "mask": sk, "gt": img, "edgegt": sk}, mode="inference")
        hard = (soft > 0.5).float()
        coarse, fine = m.netG(img, img, hard, hard, sk)
    out = {"image_u8": iu8, "sketch_u8": su8,
           "composed_u8": ((composed + 1) / 2 * 255).numpy().astype(np.uint8)[0].transpose(1, 2, 0),     # test.py:25-35 (HWC, RGB)
           "mask_u8": (soft * 255).numpy().astype(np.uint8)[0, 0],
           "hard_mask_bits": np.packbits(hard.numpy().astype(np.uint8))}
    for k, v in (("composed", composed), ("mask", soft), ("coarse", coarse), ("fine", fine)):
        out[k + "_sum"] = summary(v.numpy())
        out[k + "_crop"] = v.numpy()[:, :, 96:160, 96:160].astype(np.float32)
    print("face: sketch density %.4f  hole fraction %.3f  mask range [%.3f, %.3f]" % (
        float(sk.mean()), float(hard.mean()), float(soft.min()), float(soft.max())))
    return out


# ------------------------------------------------------------------------------------------------------------------
# Round 4 (VERDICT r3 items 1, 4 of "What's missing"): reference-generated vectors at BASELINE config-3 size, on every
# bundled sample, and for more than one weight set.  Large tensors are carried as four 64x64 crops (two of them touching
# an image border), per-row and per-column sums of every channel (a localised defect ANYWHERE moves one of them) and the
# five global statistics of summary().
# ------------------------------------------------------------------------------------------------------------------
def crop_boxes(H, W):
    """(top, left) of the four 64x64 crops: top-left corner, bottom-right corner, centre, and an off-centre interior one."""
    return [(0, 0), (H - 64, W - 64), ((H - 64) // 2, (W - 64) // 2), (min((H // 4) // 8 * 8, H - 64), min((5 * W // 8) // 8 * 8, W - 64))]


def digest(name, a, out, crops=True):
    """a: (B,C,H,W) float tensor of the reference -> crops, row / column sums, global statistics under keys name_*"""
    a = a.numpy() if hasattr(a, "numpy") else a
    H, W = a.shape[2:]
    out[name + "_sum"] = summary(a)
    out[name + "_rows"] = a.astype(np.float64).sum(3).astype(np.float32)
    out[name + "_cols"] = a.astype(np.float64).sum(2).astype(np.float32)
    if crops:
        out[name + "_crops"] = np.stack([a[:, :, t:t + 64, l:l + 64] for t, l in crop_boxes(H, W)], 0).astype(np.float32)


def similar_digest(sim):
    """similar (B, L, hs, ws) = softmax over the L keys for every query position: per query the largest probability, its
    key index, and a position-weighted checksum sum_k p_k * ((k * 37) % 101) / 101 (a permutation or a shift of keys moves it)"""
    B, L = sim.shape[:2]
    p = sim.reshape(B, L, -1).numpy().astype(np.float64)          # [B][key][query]
    wk = ((np.arange(L) * 37) % 101 / 101.0)[None, :, None]
    return dict(similar_max=p.max(1).astype(np.float32), similar_argmax=p.argmax(1).astype(np.int32),
                similar_chk=(p * wk).sum(1).astype(np.float32), similar_sum=summary(p))


def run_digest(m, img, sk, want_similar=False):
    """one inference of the reference on (img, sk) tensors -> digest dictionary"""
    out = {}
    taps = {}
    with torch.no_grad():
        composed, soft = m({"image": img, "mask": sk, "gt": img, "edgegt": sk}, mode="inference")
        hard = (soft > 0.5).float()
        hk = m.netG.cam_1.register_forward_hook(lambda mod, i, o: taps.__setitem__("similar", o))
        coarse, fine = m.netG(img, img, hard, hard, sk)
        hk.remove()
    for k, v in (("composed", composed), ("mask", soft), ("coarse", coarse), ("fine", fine)):
        digest(k, v, out, crops=k in ("composed", "mask"))      # the two stage outputs: row / column sums and statistics only
    out["hard_mask_bits"] = np.packbits(hard.numpy().astype(np.uint8))
    out["hole_fraction"] = np.array([float(hard.mean())], np.float64)
    out["composed_u8_crops"] = np.stack([((composed + 1) / 2 * 255).numpy().astype(np.uint8)[0].transpose(1, 2, 0)[t:t + 64, l:l + 64]
                                         for t, l in crop_boxes(*composed.shape[2:])], 0)               # test.py:25-35
    if want_similar:
        out.update(similar_digest(taps["similar"]))
    return out, dict(composed=composed, mask=soft, hard_mask=hard, coarse=coarse, fine=fine)


SAMPLES = [("face_release", "822_images_celeb_03375"), ("face_release", "873_images_celeb_22553"),
           ("face_release", "902_images_celeb_16692"), ("general_release", "11"), ("general_release", "556"),
           ("general_release", "830"), ("general_release", "854")]


def sample_case(m, rel, name):
    """one bundled sample as /root/reference/data/testimage_dataset.py:89-111 loads it (the first face is c1_face.npz)"""
    from PIL import Image
    image = Image.open("/root/reference/datasets/%s/images/%s.png" % (rel, name)).convert("RGB")
    w, h = image.size
    sketch = Image.open("/root/reference/datasets/%s/edges/%s.png" % (rel, name)).convert("L").resize((w, h))
    iu8, su8 = np.asarray(image, np.uint8), np.asarray(sketch, np.uint8)
    img = torch.from_numpy(((iu8.astype(np.float32).transpose(2, 0, 1) / 255.0) - 0.5) / 0.5)[None]
    sk = torch.from_numpy((su8.astype(np.float32)[None, None] / 255.0 > 0).astype(np.float32))
    out, _ = run_digest(m, img, sk)
    # the two input images as the dataset holds them (PNG bytes: data, and half the size of the decoded array in an npz)
    out["image_png"] = np.frombuffer(open("/root/reference/datasets/%s/images/%s.png" % (rel, name), "rb").read(), np.uint8)
    out["sketch_png"] = np.frombuffer(open("/root/reference/datasets/%s/edges/%s.png" % (rel, name), "rb").read(), np.uint8)
    print("%s/%s: %dx%d sketch density %.4f hole fraction %.3f" % (rel, name, h, w, float(sk.mean()), float(out["hole_fraction"][0])))
    return out


def round4():
    gain = synth.DEFAULT_GAIN
    m = build_reference(gain)
    # ---- BASELINE config 3 size: 512x512, B=1, seed 1234, attention with L = 3969 keys ---------------------------
    img, sk = synth.make_inputs(1, 512, 512, seed=1234)
    out, _ = run_digest(m, torch.from_numpy(img), torch.from_numpy(sk), want_similar=True)
    out["meta"] = np.array([gain, 0, 1234, 1, 512, 512], np.float64)
    print("512x512 hole fraction %.3f" % float(out["hole_fraction"][0]))
    np.savez_compressed(os.path.join(HERE, "e2e_512.npz"), **out)
    # ---- the 256x256 case of e2e_256.npz again, with four crops and the row / column sums ------------------------
    img, sk = synth.make_inputs(1, 256, 256, seed=1234)
    out, _ = run_digest(m, torch.from_numpy(img), torch.from_numpy(sk), want_similar=True)
    out["meta"] = np.array([gain, 0, 1234, 1, 256, 256], np.float64)
    np.savez_compressed(os.path.join(HERE, "e2e_256_crops.npz"), **out)
    # ---- the other seven bundled samples (three faces, three 512x512 scenes, the 408x512 scene) -------------------
    for rel, name in SAMPLES:
        np.savez_compressed(os.path.join(HERE, "sample_%s.npz" % name.split("_")[0]), **sample_case(m, rel, name))
    # ---- further weight sets: 64x64 B=2 in full, 256x256 B=1 as a digest --------------------------------------------
    for ws in ("w1", "w2"):
        seed, g, dist = synth.WEIGHT_SETS[ws]
        mw = build_reference(g, seed, dist)
        img, sk = synth.make_inputs(2, 64, 64, seed=1234)
        o64, full = run_digest(mw, torch.from_numpy(img), torch.from_numpy(sk))
        keep = {k: v.numpy().astype(np.float32) for k, v in full.items()}
        img, sk = synth.make_inputs(1, 256, 256, seed=1234)
        o256, _ = run_digest(mw, torch.from_numpy(img), torch.from_numpy(sk), want_similar=True)
        keep.update({"d256." + k: v for k, v in o256.items()})
        print("%s: hole fraction %.3f (64x64), %.3f (256x256)" % (ws, float(o64["hole_fraction"][0]), float(o256["hole_fraction"][0])))
        assert 0.1 < float(o64["hole_fraction"][0]) < 0.95 and 0.1 < float(o256["hole_fraction"][0]) < 0.95
        np.savez_compressed(os.path.join(HERE, "weights_%s.npz" % ws), **keep)
    for f in sorted(os.listdir(HERE)):
        if f.endswith(".npz"):
            print("%-24s %8.1f KB" % (f, os.path.getsize(os.path.join(HERE, f)) / 1024))


# ------------------------------------------------------------------------------------------------------------------
# Round 6 (VERDICT r5 "What's missing" 4a): the attention's key-validity boundary.  A key patch is valid when
# the mean over its 4x4 window of (1 - avg_pool2d(mask, 4, 4)) exceeds th = 0.1 (splitcam.py:49-53,90): exact multiples of
# 1/256, so 25/256 = 0.0977 is invalid and 26/256 = 0.1016 valid.  The fixture holds keys with exactly 24, 25, 26 and 27
# non-hole pixels among the 256 of their window, run through the reference's own cam_1 / cam_2 modules at the path's 96
# channels.  The mask is part of the fixture (bits); x is regenerated from the seed.
# ------------------------------------------------------------------------------------------------------------------
def boundary_mask():
    """(2,1,48,64) hole mask in {0,1}.  Image 0: designed -- the window of key (py,px) is the 2x2 block of 8x8-pixel cells
    (py..py+1, px..px+1), every cell holds a prescribed number of non-hole (0) pixels.  Image 1: Bernoulli(0.9) hole per
    pixel, so that a window's count is Binomial(256, 0.1): mean 25.6, keys on both sides of the threshold everywhere."""
    full = np.ones((2, 1, 48, 64), np.float32)
    z = (synth.uniform(11, "att_th.cells", (6, 8), 0, 1) * 14).astype(np.int64)            # 0..13 zeros per cell: window sums 0..52
    z[0:2, 0:4] = 0
    z[0, 0] = 25          # key (0,0): exactly 25 -> invalid
    z[0, 2] = 26          # keys (0,1) and (0,2): 26 + 0 -> valid
    z[4:6, 4:8] = 0
    z[4, 4], z[5, 5] = 12, 12          # key (4,4): 24 -> invalid
    z[4, 7], z[5, 7] = 20, 7           # key (4,6): 27 -> valid
    order = np.argsort(synth.uniform(11, "att_th.order", (64,), 0, 1), kind="stable")       # which pixels of a cell are non-hole
    for cy in range(6):
        for cx in range(8):
            cell = np.ones(64, np.float32)
            cell[order[: z[cy, cx]]] = 0.0
            full[0, 0, 8 * cy:8 * cy + 8, 8 * cx:8 * cx + 8] = cell.reshape(8, 8)
    full[1, 0] = (synth.uniform(11, "att_th.m1", (48, 64), 0, 1) < 0.9).astype(np.float32)
    return full


def window_counts(full):
    """non-hole pixels in the 16x16 window of every key patch, in exact integer arithmetic: (B, 5, 7)"""
    nh = (1 - full[:, 0]).astype(np.int64)
    B, H, W = nh.shape
    return np.stack([[[nh[b, 8 * py:8 * py + 16, 8 * px:8 * px + 16].sum() for px in range((W // 4 - 4) // 2 + 1)]
                      for py in range((H // 4 - 4) // 2 + 1)] for b in range(B)])


def round6():
    from models.networks.splitcam import ReduceContextAttentionP1, ReduceContextAttentionP2
    ctor1 = dict(nn_hard=False, ufstride=2, stride=2, bkg_patch_size=4, pd=0, is_th=True, norm_type=1)      # editline_g.py:35-38
    cam1 = ReduceContextAttentionP1(th=0.1, **ctor1)
    cam2 = ReduceContextAttentionP2(ufstride=2, bkg_patch_size=4, stride=2, pd=0, mk=False)              # editline_g.py:39-42
    full = torch.from_numpy(boundary_mask())
    cnt = window_counts(full.numpy())
    for b in range(2):
        assert {24, 25, 26, 27} <= set(cnt[b].ravel().tolist()), sorted(set(cnt[b].ravel().tolist()))
    assert cnt[0, 0, 0] == 25 and cnt[0, 0, 1] == 26 and cnt[0, 4, 4] == 24 and cnt[0, 4, 6] == 27
    # soft scores: the softmax stays far from one-hot, so the validity of EVERY key shows in every query's row
    x = torch.from_numpy(0.004 * synth.uniform(11, "att_th.x", (2, 96, 12, 16), -1, 1))
    ms = F.avg_pool2d(full, 4, 4)
    with torch.no_grad():
        sim = cam1(x, x, ms)
        rec, _ = cam2(sim, x, ms, {})
        # sensitivity: the same run with the threshold just BELOW 25/256 -- the 25-pixel keys become valid
        sim_lo = ReduceContextAttentionP1(th=0.097, **ctor1)(x, x, ms)
        sim_hi = ReduceContextAttentionP1(th=0.102, **ctor1)(x, x, ms)      # just ABOVE 26/256: the 26-pixel keys become invalid
    d_lo, d_hi = float((sim - sim_lo).abs().max()), float((sim - sim_hi).abs().max())
    print("att_th: P range [%.4f, %.4f]; moving th across 25/256 changes similar by %.3e, across 26/256 by %.3e" % (
        float(sim.min()), float(sim.max()), d_lo, d_hi))
    assert float(sim.max()) < 0.5 and d_lo > 1e-3 and d_hi > 1e-3
    out = {"op.att_th.mask_bits": np.packbits(full.numpy().astype(np.uint8)), "op.att_th.counts": cnt.astype(np.int32),
           "op.att_th.similar": sim.numpy(), "op.att_th.out": rec.numpy(),
           "op.att_th.sensitivity": np.array([d_lo, d_hi], np.float64)}
    np.savez_compressed(os.path.join(HERE, "ops_r6.npz"), **out)
    print("ops_r6.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "ops_r6.npz")) / 1024))


def main():
    if "--round4" in sys.argv:
        return round4()
    if "--round6" in sys.argv:
        return round6()
    gain = synth.DEFAULT_GAIN
    m = build_reference(gain)
    if "--only-face" in sys.argv:
        np.savez_compressed(os.path.join(HERE, "c1_face.npz"), meta=np.array([gain, 0], np.float64), **face_case(m))
        print("c1_face.npz %.1f KB" % (os.path.getsize(os.path.join(HERE, "c1_face.npz")) / 1024))
        return
    # ---- 64x64, B=2: full tensors --------------------------------------------------
    small = run_case(m, 2, 64, 64, seed=1234)
    hf = float(small["hard_mask"].mean())
    print("64x64 hole fraction %.3f  mask range [%.3f, %.3f]  fine range [%.3f, %.3f]" % (
        hf, small["mask"].min(), small["mask"].max(), small["fine"].min(), small["fine"].max()))
    assert 0.1 < hf < 0.9, "degenerate golden: hole never/always open"
    keep = {k: small[k].astype(np.float32) for k in
            ("composed", "mask", "mask_image", "hard_mask", "coarse", "fine", "style_vec", "similar", "attn_out")}
    keep["coarse_enc_sum"] = summary(small["coarse_enc"])
    keep["pmconv6_sum"] = summary(small["pmconv6"])
    keep["m_conv1_sum"] = summary(small["m_conv1"])
    keep["m_conv10_sum"] = summary(small["m_conv10"])
    keep["m_conv10_crop"] = small["m_conv10"][:, :8].astype(np.float32)
    keep["meta"] = np.array([gain, 0, 1234, 2, 64, 64], np.float64)
    np.savez_compressed(os.path.join(HERE, "e2e_64.npz"), **keep)
    # ---- 64x64 flag variants (next-row surface: pool avg / no cam / no_mask_*) ------
    var = {}
    for tag, over in (("avg", dict(pool_type="avg")), ("nocam", dict(use_cam=False)),
                      ("nomaskcc", dict(no_mask_cc=True)), ("nomaskcoarse", dict(no_mask_coarse=True)),
                      ("nojoint", dict(joint_train_inp=False))):
        mv = build_reference(gain, **over)
        r = run_case(mv, 1, 64, 64, seed=1234)
        var[tag + ".coarse"] = r["coarse"].astype(np.float32)
        var[tag + ".fine"] = r["fine"].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, "variants_64.npz"), **var)
    # ---- non-square 40x72 (multiple of 8), B=1 ---------------------------------------
    ns = run_case(m, 1, 40, 72, seed=99)
    print("40x72 hole fraction %.3f" % float(ns["hard_mask"].mean()))
    np.savez_compressed(os.path.join(HERE, "e2e_40x72.npz"),
                        **{k: ns[k].astype(np.float32) for k in ("composed", "mask", "hard_mask", "coarse", "fine")},
                        meta=np.array([gain, 0, 99, 1, 40, 72], np.float64))
    # ---- 256x256, B=1: crops + checksums ------------------------------------------
    big = run_case(m, 1, 256, 256, seed=1234)
    hf = float(big["hard_mask"].mean())
    print("256x256 hole fraction %.3f  mask range [%.3f, %.3f]" % (hf, big["mask"].min(), big["mask"].max()))
    assert 0.1 < hf < 0.9
    kb = {}
    for k in ("composed", "mask", "coarse", "fine", "hard_mask"):
        kb[k + "_sum"] = summary(big[k])
        kb[k + "_crop"] = big[k][:, :, 96:160, 96:160].astype(np.float32)
    kb["hard_mask_bits"] = np.packbits(big["hard_mask"].astype(np.uint8))
    kb["meta"] = np.array([gain, 0, 1234, 1, 256, 256], np.float64)
    np.savez_compressed(os.path.join(HERE, "e2e_256.npz"), **kb)
    # ---- bf16 comparator (BASELINE config 5): reference + rounding hooks, 64x64 B=2 ---------------------------
    b16 = run_case_bf16(gain, 2, 64, 64, seed=1234)
    print("bf16 64x64: |composed - fp32| max %.4f  hard-mask flips vs fp32 %d" % (
        np.abs(b16["composed"] - small["composed"]).max(), int((b16["hard_mask"] != small["hard_mask"]).sum())))
    np.savez_compressed(os.path.join(HERE, "e2e_64_bf16.npz"), meta=np.array([gain, 0, 1234, 2, 64, 64], np.float64), **b16)
    # ---- BASELINE config 1: the bundled face sample --------------------------------
    np.savez_compressed(os.path.join(HERE, "c1_face.npz"), meta=np.array([gain, 0], np.float64), **face_case(m))
    # ---- per-op known answers ------------------------------------------------------
    np.savez_compressed(os.path.join(HERE, "ops.npz"), **op_cases())
    round4()
    round6()


if __name__ == "__main__":
    main()
