"""GPU parity tests: the HIP path (through the C-ABI, sketchedit_amd._lib.Engine) against the golden
vectors captured from the reference and against the oracle on the same seeded inputs.

Tolerance: BASELINE.json north_star -- max-abs <= 1e-3 in fp32 against the reference CPU forward.
Per-op tests use tighter bounds (1e-4) since no depth amplification is involved.
"""
import os

import numpy as np
import pytest
import torch

from sketchedit_amd import synth
from parity_util import composed_for_hard_mask

pytestmark = pytest.mark.gpu

TOL_E2E = 1e-3
TOL_OP = 1e-4
FLAGS = 1 | 2 | 16   # use_cam, pool max, joint_train_inp  (test_celeb.sh)


@pytest.fixture(scope="module")
def eng():
    from sketchedit_amd._lib import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def eng_w(eng):
    eng.load_state_dict("M", synth.make_state_dict("M", 0))
    eng.load_state_dict("G", synth.make_state_dict("G", 0))
    assert eng.weights_ready()
    return eng


def _load(golden_dir, name):
    return dict(np.load(os.path.join(golden_dir, name)))


def _md(a, b):
    a = a.detach().cpu().numpy() if hasattr(a, "detach") else np.asarray(a)
    b = b.detach().cpu().numpy() if hasattr(b, "detach") else np.asarray(b)
    assert a.shape == b.shape, (a.shape, b.shape)
    return float(np.abs(a.astype(np.float64) - b.astype(np.float64)).max())


def _cuda(a):
    return torch.from_numpy(np.ascontiguousarray(a, np.float32)).cuda()


OPS = [("c3_s1_d1_elu", 8, 16, 3, 1, 1, "elu", 12, 16), ("c3_s2_d1_elu", 8, 16, 3, 2, 1, "elu", 12, 16),
       ("c3_s1_d2_elu", 8, 16, 3, 1, 2, "elu", 12, 16), ("c3_s1_d16_elu", 8, 16, 3, 1, 16, "elu", 20, 24),
       ("c3_s1_d1_relu", 8, 16, 3, 1, 1, "relu", 12, 16), ("c3_s1_d1_none", 12, 1, 3, 1, 1, None, 12, 16),
       ("c3_s1_d1_rgb", 12, 3, 3, 1, 1, "elu", 12, 16), ("c5_s1_d1_elu", 5, 16, 5, 1, 1, "elu", 12, 16)]


@pytest.mark.parametrize("case", OPS, ids=[c[0] for c in OPS])
def test_op_gated_conv_golden(eng, golden_dir, case):
    name, cin, cout, k, s, r, act, H, W = case
    g = _load(golden_dir, "ops.npz")
    w = synth.uniform(7, name + ".w", (cout, cin, k, k), -0.5, 0.5)
    b = synth.uniform(7, name + ".b", (cout,), -0.5, 0.5)
    x = synth.uniform(7, name + ".x", (2, cin, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=s, rate=r, act=act)
    assert _md(y, g["op." + name]) < TOL_OP


def test_op_deconv_golden(eng, golden_dir):
    g = _load(golden_dir, "ops.npz")
    w = synth.uniform(7, "deconv.w", (16, 8, 3, 3), -0.5, 0.5)
    b = synth.uniform(7, "deconv.b", (16,), -0.5, 0.5)
    x = synth.uniform(7, "deconv.x", (2, 8, 6, 8), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, upsample=True)
    assert _md(y, g["op.deconv"]) < TOL_OP


# every (Cin, Cout, stride, rate, up) shape class of the network, ragged sizes, vs the oracle
NET_SHAPES = [(96, 192, 1, 1, False, 3), (96, 192, 1, 16, False, 3), (96, 192, 1, 4, False, 3),
              (192, 192, 1, 1, False, 3), (48, 192, 2, 1, False, 3), (48, 192, 1, 1, False, 3),
              (48, 96, 1, 1, False, 3), (24, 96, 2, 1, False, 3), (24, 96, 1, 1, False, 3),
              (96, 96, 1, 1, True, 3), (48, 48, 1, 1, True, 3), (24, 48, 2, 1, False, 3),
              (48, 96, 2, 1, False, 3), (24, 24, 1, 1, False, 3), (4, 48, 1, 1, False, 5),
              (5, 48, 1, 1, False, 5), (3, 48, 1, 1, False, 5)]


@pytest.mark.parametrize("shape", NET_SHAPES, ids=["%d-%d-s%d-d%d-u%d-k%d" % s for s in NET_SHAPES])
def test_op_gated_conv_network_shapes(eng, shape):
    from oracle import sketchedit_oracle as O
    cin, cout, s, r, up, k = shape
    H, W = (10, 14) if up else (22, 18)          # ragged: not a multiple of the pixel tile
    a = 1.5 / np.sqrt(cin * k * k)
    w = synth.uniform(11, "ns.w%s" % (shape,), (cout, cin, k, k), -a, a)
    b = synth.uniform(11, "ns.b%s" % (shape,), (cout,), -0.3, 0.3)
    x = synth.uniform(11, "ns.x%s" % (shape,), (3, cin, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=s, rate=r, upsample=up)
    tw, tb, tx = torch.from_numpy(w), torch.from_numpy(b), torch.from_numpy(x)
    ref = O.gated_deconv(tx, tw, tb) if up else O.gated_conv(tx, tw, tb, s, r, "elu")
    assert _md(y, ref) < TOL_OP


WINO = [(1, 16, 16, "elu"), (2, 16, 24, "elu"), (4, 16, 32, "elu"), (8, 32, 16, "relu"), (16, 64, 32, "elu"),
        (1, 10, 14, "elu")]


@pytest.mark.parametrize("case", WINO, ids=["d%d-%dx%d-%s" % c for c in WINO])
def test_op_winograd_path_vs_oracle(eng, case, seopt):
    """96 -> 192 3x3 stride 1: sizes with h % 2d == w % 2d == 0 take the Winograd F(2x2,3x3) kernel
    (se_wino.hip), the last (10x14 ... but d=1 -> eligible too) covers a ragged tile count."""
    from oracle import sketchedit_oracle as O
    seopt.set("SE_WINOGRAD_F43", "0")      # (the hybrid kernel has its own test below)
    d, H, W, act = case
    a = 1.5 / np.sqrt(96 * 9)
    w = synth.uniform(13, "wino.w%s" % (case,), (192, 96, 3, 3), -a, a)
    b = synth.uniform(13, "wino.b%s" % (case,), (192,), -0.3, 0.3)
    x = synth.uniform(13, "wino.x%s" % (case,), (3, 96, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, act=act)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, d, act)
    assert _md(y, ref) < TOL_OP


WINO24 = [(1, 16, 16, "elu"), (1, 6, 12, "elu"), (1, 10, 20, "relu"), (2, 16, 24, "elu"), (2, 12, 8, "elu"), (4, 16, 32, "elu"),
          (8, 32, 32, "relu"), (16, 64, 64, "elu"), (16, 32, 64, "elu"), (1, 64, 64, "elu"), (3, 12, 24, "elu")]


@pytest.mark.parametrize("case", WINO24, ids=["d%d-%dx%d-%s" % c for c in WINO24])
def test_op_winograd_f43_path_vs_oracle(eng, case, seopt):
    """96 -> 192 3x3 stride 1 with h % 2d == 0 and w % 4d == 0: the hybrid F(2,3) x F(4,3) kernel (se_wino24.hip, 2x4 output
    tiles, non-dyadic transform constants) against the oracle -- dilations 1..16 (and 3: a polyphase grid that is not a
    power of two), tile counts that are not multiples of the 32-tile workgroup (27, 75, 18), both activations -- and against
    the F(2x2,3x3) kernel on the same input (SE_WINOGRAD_F43=0)."""
    from oracle import sketchedit_oracle as O
    d, H, W, act = case
    a = 1.5 / np.sqrt(96 * 9)
    w = synth.uniform(17, "w24.w%s" % (case,), (192, 96, 3, 3), -a, a)
    b = synth.uniform(17, "w24.b%s" % (case,), (192,), -0.3, 0.3)
    x = synth.uniform(17, "w24.x%s" % (case,), (3, 96, H, W), -1, 1)
    seopt.set("SE_WINOGRAD_F43", "1")
    y = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, act=act)
    seopt.set("SE_WINOGRAD_F43", "0")
    y22 = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, act=act)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, d, act)
    assert _md(y, ref) < TOL_OP and _md(y22, ref) < TOL_OP
    assert _md(y, y22) < 2e-5
    # larger activations (|x| up to 8: the range the F(4,3) constants amplify): relative bound
    x8 = x * 8.0
    seopt.set("SE_WINOGRAD_F43", "1")
    y8 = eng.gated_conv2d(_cuda(x8), w, b, stride=1, rate=d, act=act)
    ref8 = O.gated_conv(torch.from_numpy(x8), torch.from_numpy(w), torch.from_numpy(b), 1, d, act)
    assert _md(y8, ref8) < TOL_OP * max(1.0, float(ref8.abs().max()))


@pytest.mark.parametrize("size", [(16, 16), (6, 12), (64, 64), (10, 20)], ids=lambda s: "%dx%d" % s)
def test_op_winograd_f43_48_channel_source_vs_oracle(eng, size, seopt):
    """xconv5 of netG (48 -> 192, 3x3, stride 1): the hybrid kernel's 48-channel instantiation (two chunks per position, the
    second half empty: k-half 0 only) against the oracle and against the direct kernel (SE_WINOGRAD_F43=0)."""
    from oracle import sketchedit_oracle as O
    H, W = size
    a = 1.5 / np.sqrt(48 * 9)
    w = synth.uniform(19, "w24c48.w", (192, 48, 3, 3), -a, a)
    b = synth.uniform(19, "w24c48.b", (192,), -0.3, 0.3)
    x = synth.uniform(19, "w24c48.x%s" % (size,), (3, 48, H, W), -1, 1)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1, "elu")
    seopt.set("SE_WINOGRAD_F43", "1")
    y = eng.gated_conv2d(_cuda(x), w, b)
    seopt.set("SE_WINOGRAD_F43", "0")
    yd = eng.gated_conv2d(_cuda(x), w, b)
    assert _md(y, ref) < TOL_OP and _md(yd, ref) < TOL_OP
    assert _md(y, yd) < 2e-5 and _md(y, yd) > 0.0          # (two different kernels ran)


C24 = [(16, 16, "elu"), (8, 32, "relu"), (13, 22, "elu"), (37, 50, "elu"), (64, 64, "elu"), (3, 2, "elu"), (2, 2, "elu"), (38, 20, "relu"),
       (256, 256, "elu")]


@pytest.mark.parametrize("case", C24, ids=["%dx%d-%s" % c for c in C24])
def test_op_conv24_winograd_along_x_vs_oracle(eng, case, seopt):
    """24 -> 24 3x3 stride 1 (conv16 / allconv16 / conv_mask_16, the full-resolution layer in front of every output conv):
    the persistent raw-tile kernels of se_rtilew.hip -- two-dimensional F(2x2,3x3) (even heights and widths, the default) and
    F(2,3) along x (even widths) -- against the oracle and against the direct raw-tile kernel (SE_RTILE_WX=0) -- heights and widths that are not multiples of the 8 x 16 block, a 2-pixel-wide image (every
    column a border column), both activations, and the network's own 256 x 256."""
    from oracle import sketchedit_oracle as O
    H, W, act = case
    a = 1.5 / np.sqrt(24 * 9)
    w = synth.uniform(23, "c24.w%s" % (case,), (24, 24, 3, 3), -a, a)
    b = synth.uniform(23, "c24.b%s" % (case,), (24,), -0.3, 0.3)
    x = synth.uniform(23, "c24.x%s" % (case,), (2, 24, H, W), -1, 1)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1, act)
    seopt.set("SE_RTILE_WX", "2")              # default: two-dimensional F(2x2,3x3) where the height is even too
    y2 = eng.gated_conv2d(_cuda(x), w, b, act=act)
    seopt.set("SE_RTILE_WX", "1")              # F(2,3) along x only
    y = eng.gated_conv2d(_cuda(x), w, b, act=act)
    seopt.set("SE_RTILE_WX", "0")              # direct raw-tile kernel
    yd = eng.gated_conv2d(_cuda(x), w, b, act=act)
    assert tuple(y.shape) == (2, 12, H, W)
    assert _md(y, ref) < TOL_OP and _md(yd, ref) < TOL_OP and _md(y2, ref) < TOL_OP
    assert 0.0 < _md(y, yd) < 2e-5                      # (different kernels ran)
    if H % 2 == 0:
        assert 0.0 < _md(y2, y) < 2e-5


WINO48 = [(1, 16, 16, "elu"), (1, 64, 64, "elu"), (2, 16, 24, "relu"), (4, 32, 16, "elu"), (1, 10, 14, "elu"),
          (1, 34, 30, "elu")]


@pytest.mark.parametrize("case", WINO48, ids=["d%d-%dx%d-%s" % c for c in WINO48])
def test_op_winograd48_path_vs_oracle(eng, case):
    """48 -> 96 3x3 stride 1 (the 128x128 level of every encoder / decoder): Winograd kernel of se_wino48.hip --
    three 32-k chunks per position pair, MIXED feature/gate rows; ragged tile counts and dilations included."""
    from oracle import sketchedit_oracle as O
    d, H, W, act = case
    a = 1.5 / np.sqrt(48 * 9)
    w = synth.uniform(17, "wino48.w%s" % (case,), (96, 48, 3, 3), -a, a)
    b = synth.uniform(17, "wino48.b%s" % (case,), (96,), -0.3, 0.3)
    x = synth.uniform(17, "wino48.x%s" % (case,), (3, 48, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, act=act)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, d, act)
    assert _md(y, ref) < TOL_OP


@pytest.mark.parametrize("case", WINO48, ids=["d%d-%dx%d-%s" % c for c in WINO48])
def test_op_winograd24_to_96_path_vs_oracle(eng, case, seopt):
    """24 -> 96 3x3 stride 1 (xconv3 / pmconv3 of netG, editline_g.py:63,75) on the 48-channel Winograd kernel's CIN = 24
    form (round 5): one 32-k chunk per position whose k-half 1 carries channels 16-23 in two k-steps.  Ragged tile counts,
    dilations, both activations; against the oracle and against the direct gather-GEMM (SE_WINOGRAD48=0), which must have
    been a different kernel."""
    from oracle import sketchedit_oracle as O
    d, H, W, act = case
    a = 1.5 / np.sqrt(24 * 9)
    w = synth.uniform(18, "wino24c.w%s" % (case,), (96, 24, 3, 3), -a, a)
    b = synth.uniform(18, "wino24c.b%s" % (case,), (96,), -0.3, 0.3)
    x = synth.uniform(18, "wino24c.x%s" % (case,), (3, 24, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, act=act)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, d, act)
    assert _md(y, ref) < TOL_OP
    seopt.set("SE_WINOGRAD48", 0)
    yd = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, act=act)
    assert _md(yd, ref) < TOL_OP
    if H % (2 * d) == 0 and W % (2 * d) == 0:
        assert 0.0 < _md(y, yd) < 2e-5                  # (different kernels ran)


WINOUP = [(8, 8, "elu"), (16, 24, "elu"), (64, 64, "elu"), (10, 14, "relu"), (34, 30, "elu"), (2, 2, "elu")]


@pytest.mark.parametrize("cin", [96, 48])
@pytest.mark.parametrize("case", WINOUP, ids=["%dx%d-%s" % c for c in WINOUP])
def test_op_winograd_upsample_path_vs_oracle(eng, case, cin):
    """gen_deconv 96 -> 96 and 48 -> 48 (nearest x2 + 3x3): F(2x2,2x2) Winograd on the four sub-pixel classes
    (se_wino_up.hip / se_wino_up48.hip, the latter with two positions per three 32-k chunks); even source sizes take it,
    ragged tile counts and the image borders (zero padding on the upsampled grid) included."""
    from oracle import sketchedit_oracle as O
    H, W, act = case
    a = 1.5 / np.sqrt(cin * 9)
    w = synth.uniform(19, "winoup%d.w%s" % (cin, case), (cin, cin, 3, 3), -a, a)
    b = synth.uniform(19, "winoup%d.b%s" % (cin, case), (cin,), -0.3, 0.3)
    x = synth.uniform(19, "winoup%d.x%s" % (cin, case), (3, cin, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, upsample=True, act=act)
    xu = torch.from_numpy(x).repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)      # utils.py:47
    ref = O.gated_conv(xu, torch.from_numpy(w), torch.from_numpy(b), 1, 1, act)
    assert tuple(y.shape) == (3, cin // 2, 2 * H, 2 * W)
    assert _md(y, ref) < TOL_OP


TWO_SRC = [("tensor", 16, 16, 1), ("vector", 16, 24, 1), ("vector", 10, 14, 1), ("tensor", 22, 18, 1)]


@pytest.mark.parametrize("case", TWO_SRC, ids=["%s-%dx%d" % c[:3] for c in TWO_SRC])
@pytest.mark.parametrize("ll", [False, True], ids=["default", "lowlat"])
def test_op_two_source_conv_vs_oracle(eng, case, ll):
    """conv11 / allconv11 (editline_g.py:166-167,211): 3x3 conv over the virtual concat cat([x, x1]), x1 a tensor
    (patch-match branch) or the pooled style vector broadcast over the image (zero padded at the borders like a
    tensor).  Even sizes take wino_kernel<6> (se_wino.hip), the 22x18 case and the low-latency mode the direct
    kernel's two-source gather."""
    from oracle import sketchedit_oracle as O
    kind, H, W, d = case
    a = 1.5 / np.sqrt(192 * 9)
    w = synth.uniform(23, "two.w%s" % (case,), (192, 192, 3, 3), -a, a)
    b = synth.uniform(23, "two.b%s" % (case,), (192,), -0.3, 0.3)
    x = synth.uniform(23, "two.x%s" % (case,), (3, 96, H, W), -1, 1)
    if kind == "tensor":
        x1 = synth.uniform(23, "two.y%s" % (case,), (3, 96, H, W), -1, 1)
        cat = np.concatenate([x, x1], 1)
    else:
        x1 = synth.uniform(23, "two.v%s" % (case,), (3, 96), -1, 1)
        cat = np.concatenate([x, np.broadcast_to(x1[:, :, None, None], (3, 96, H, W))], 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=1, rate=d, x1=_cuda(x1), low_latency=ll)
    ref = O.gated_conv(torch.from_numpy(cat), torch.from_numpy(w), torch.from_numpy(b), 1, d, "elu")
    assert _md(y, ref) < TOL_OP


@pytest.mark.parametrize("size", [(16, 16), (12, 20), (6, 8), (64, 64)], ids=lambda s: "%dx%d" % s)
def test_op_two_source_winograd_forms_vs_oracle(eng, size, seopt):
    """allconv11 (editline_g.py:211, cat([x_hallu, pm])): the two-tensor layer in the hybrid F(2,3) x F(4,3) form
    (wino24_kernel<6>: chunks 3-5 of every position gathered from the second tensor) and in the F(2x2,3x3) form
    (wino_kernel<6>, SE_WINOGRAD_F43=0), both against the oracle on the materialised concat."""
    from oracle import sketchedit_oracle as O
    H, W = size
    a = 1.5 / np.sqrt(192 * 9)
    w = synth.uniform(29, "two24.w", (192, 192, 3, 3), -a, a)
    b = synth.uniform(29, "two24.b", (192,), -0.3, 0.3)
    x = synth.uniform(29, "two24.x%s" % (size,), (3, 96, H, W), -1, 1)
    x1 = synth.uniform(29, "two24.y%s" % (size,), (3, 96, H, W), -1, 1)
    ref = O.gated_conv(torch.from_numpy(np.concatenate([x, x1], 1)), torch.from_numpy(w), torch.from_numpy(b), 1, 1, "elu")
    seopt.set("SE_WINOGRAD_F43", "1")
    y24 = eng.gated_conv2d(_cuda(x), w, b, x1=_cuda(x1))
    seopt.set("SE_WINOGRAD_F43", "0")
    y22 = eng.gated_conv2d(_cuda(x), w, b, x1=_cuda(x1))
    assert _md(y24, ref) < TOL_OP and _md(y22, ref) < TOL_OP
    assert _md(y24, y22) < 2e-5


@pytest.mark.parametrize("shape", NET_SHAPES, ids=["%d-%d-s%d-d%d-u%d-k%d" % s for s in NET_SHAPES])
def test_op_low_latency_shapes_vs_oracle(eng, shape):
    """Every layer shape of the network in its small-grid launch shape (SE_FLAG_LOW_LATENCY: 64-pixel tiles, the wide
    layers' rows split over blockIdx.y), ragged sizes."""
    from oracle import sketchedit_oracle as O
    cin, cout, s, r, up, k = shape
    H, W = (10, 14) if up else (22, 18)
    a = 1.5 / np.sqrt(cin * k * k)
    w = synth.uniform(29, "ll.w%s" % (shape,), (cout, cin, k, k), -a, a)
    b = synth.uniform(29, "ll.b%s" % (shape,), (cout,), -0.3, 0.3)
    x = synth.uniform(29, "ll.x%s" % (shape,), (3, cin, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b, stride=s, rate=r, upsample=up, low_latency=True)
    tw, tb, tx = torch.from_numpy(w), torch.from_numpy(b), torch.from_numpy(x)
    ref = O.gated_deconv(tx, tw, tb) if up else O.gated_conv(tx, tw, tb, s, r, "elu")
    assert _md(y, ref) < TOL_OP


@pytest.mark.parametrize("size", [(16, 16), (12, 20), (2, 8), (64, 64)], ids=lambda s: "%dx%d" % s)
def test_op_vector_source_folded_into_bias(eng, size, seopt):
    """conv11 of netG reads cat([x, pooled style vector]) (editline_g.py:166-167): the spatially constant second source is
    folded into a per-image, per-border-configuration bias table (launch_vecbias: nine configurations of in-bounds taps,
    because the vector is still ZERO PADDED at the image borders) and the layer runs as the single-source Winograd kernel.
    Against the oracle on the materialised concat, and against the two-source kernel (SE_VECBIAS=0); sizes with every
    border configuration, a 2-row image (top and bottom rows only) and the network's own 64x64."""
    from oracle import sketchedit_oracle as O
    H, W = size
    a = 1.5 / np.sqrt(192 * 9)
    w = synth.uniform(53, "vb.w", (192, 192, 3, 3), -a, a)
    b = synth.uniform(53, "vb.b", (192,), -0.3, 0.3)
    x = synth.uniform(53, "vb.x%d" % H, (3, 96, H, W), -1, 1)
    v = synth.uniform(53, "vb.v", (3, 96), -1, 1)
    cat = np.concatenate([x, np.broadcast_to(v[:, :, None, None], (3, 96, H, W))], 1)
    ref = O.gated_conv(torch.from_numpy(cat), torch.from_numpy(w), torch.from_numpy(b), 1, 1, "elu")
    seopt.set("SE_VECBIAS", "1")
    folded = eng.gated_conv2d(_cuda(x), w, b, x1=_cuda(v))
    seopt.set("SE_VECBIAS", "0")
    two = eng.gated_conv2d(_cuda(x), w, b, x1=_cuda(v))
    assert _md(folded, ref) < TOL_OP and _md(two, ref) < TOL_OP
    assert _md(folded, two) < 1e-5


@pytest.mark.parametrize("case", [(5, 22, 18), (3, 22, 18), (4, 22, 18), (5, 8, 16), (3, 40, 33), (4, 16, 48), (5, 9, 70)], ids=lambda c: "c%d-%dx%d" % c)
def test_op_first_layer_dense_k_vs_oracle(eng, case, seopt):
    """The 5x5 first layers whose stored input carries padding channels (5 of NHWC8, 3 of NHWC4) in the dense-K raw-tile form
    (se_rtile.hip rtile_dense5_kernel: k = tap * cin + channel, dword-granular staging): exact, ragged and single-tile
    sizes, image borders on every side; against the oracle and against the channel-padded form (SE_RTILE_DENSE=0)."""
    from oracle import sketchedit_oracle as O
    cin, H, W = case
    a = 1.5 / np.sqrt(cin * 25)
    w = synth.uniform(47, "d5.w%s" % (case,), (48, cin, 5, 5), -a, a)
    b = synth.uniform(47, "d5.b%s" % (case,), (48,), -0.3, 0.3)
    x = synth.uniform(47, "d5.x%s" % (case,), (3, cin, H, W), -1, 1)
    y = eng.gated_conv2d(_cuda(x), w, b)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1, "elu")
    assert _md(y, ref) < TOL_OP
    seopt.set("SE_RTILE_DENSE", "0")
    # the switch takes effect from the next call on (se_debug_set_option): the channel-padded form sums in another order, so
    # the two results agree to rounding but are not the same bits (odd widths: both calls run the direct dense / padded forms)
    y2 = eng.gated_conv2d(_cuda(x), w, b)
    assert _md(y2, ref) < TOL_OP
    assert _md(y, y2) < TOL_OP
    if H * W >= 64:
        assert _md(y, y2) > 0.0


@pytest.mark.parametrize("case", [(5, 22, 18), (3, 22, 18), (4, 22, 18), (5, 8, 16), (3, 40, 34), (4, 16, 48), (5, 9, 70), (3, 5, 2), (4, 256, 256)],
                         ids=lambda c: "c%d-%dx%d" % c)
def test_op_first_layer_winograd_along_x_vs_oracle(eng, case, seopt):
    """The 5x5 first layers with F(2,5) along x (se_rtile.hip rtile_dense5w_kernel; even widths; non-dyadic constants 1/6,
    1/24, 2/3) against the oracle and against the dense direct kernel (SE_RTILE_D5W=0): 3, 4 and 5 real channels, ragged
    blocks, a 2-pixel-wide image, the network's 256 x 256; and at |x| up to 8 with a relative bound."""
    from oracle import sketchedit_oracle as O
    cin, H, W = case
    a = 1.5 / np.sqrt(cin * 25)
    w = synth.uniform(59, "d5w.w%s" % (case,), (48, cin, 5, 5), -a, a)
    b = synth.uniform(59, "d5w.b%s" % (case,), (48,), -0.3, 0.3)
    x = synth.uniform(59, "d5w.x%s" % (case,), (2, cin, H, W), -1, 1)
    ref = O.gated_conv(torch.from_numpy(x), torch.from_numpy(w), torch.from_numpy(b), 1, 1, "elu")
    seopt.set("SE_RTILE_D5W", "1")
    y = eng.gated_conv2d(_cuda(x), w, b)
    seopt.set("SE_RTILE_D5W", "0")
    yd = eng.gated_conv2d(_cuda(x), w, b)
    assert _md(y, ref) < TOL_OP and _md(yd, ref) < TOL_OP
    assert 0.0 < _md(y, yd) < 3e-5                      # (two different kernels ran)
    seopt.set("SE_RTILE_D5W", "1")
    y8 = eng.gated_conv2d(_cuda(x * 8.0), w, b)
    ref8 = O.gated_conv(torch.from_numpy(x * 8.0), torch.from_numpy(w), torch.from_numpy(b), 1, 1, "elu")
    assert _md(y8, ref8) < TOL_OP * max(1.0, float(ref8.abs().max()))


def test_op_first_layer_dense_k_oversized_batch_keeps_the_kernel_form(eng, seopt):
    """a batch whose first-layer input exceeds the kernel's 32-bit byte offsets runs as sub-launches of the SAME
    dense-K kernel (not the channel-padded one, which sums in another order): forced here with a small byte limit
    (SE_TEST_OFFSET_LIMIT), the result is bit-identical to the single launch."""
    a = 1.5 / np.sqrt(5 * 25)
    w = synth.uniform(61, "dk.w", (48, 5, 5, 5), -a, a)
    b = synth.uniform(61, "dk.b", (48,), -0.3, 0.3)
    x = synth.uniform(61, "dk.x", (5, 5, 24, 40), -1, 1)
    y1 = eng.gated_conv2d(_cuda(x), w, b)
    seopt.set("SE_TEST_OFFSET_LIMIT", str(2 * 24 * 40 * 8 * 4 + 1))      # two images (stored NHWC8) per sub-launch: 2 + 2 + 1
    y2 = eng.gated_conv2d(_cuda(x), w, b)
    assert torch.equal(y1, y2)


@pytest.mark.parametrize("shape", [(2, 16, 12), (2, 16, 16), (1, 132, 136)], ids=lambda s: "%dx%dx%d" % s)
def test_op_attention_soft_scores_vs_oracle(eng, shape):
    """Small activations keep the softmax far from one-hot (10 * <q, k> of order 1), so a wrong pairing of pixels in
    the space-to-depth form (se_attention.hip) or a mis-scaled score cannot hide behind a saturated softmax.
    16x12: class grid 8x6 (element-load box sum); 16x16: wc % 4 == 0 (vector box sum); 132x136: 4488 class-grid pixels,
    the softmax form for rows that do not fit in registers."""
    from oracle import sketchedit_oracle as O
    B, h, w = shape
    x = 0.004 * synth.uniform(5, "att96s.x%d" % h, (B, 96, h, w), -1, 1)
    full = (synth.uniform(5, "att96s.m%d" % h, (B, 1, 4 * h, 4 * w), 0, 1) < 0.5).astype(np.float32)
    full[0, :, :, 2 * w:] = 1.0
    out, sim = eng.attention(_cuda(x), _cuda(full), want_similar=True)
    ro, rp = O.contextual_attention(torch.from_numpy(x), torch.from_numpy(full))
    if h < 100:
        assert float(rp.max()) < 0.5 and float(rp.min()) > 1e-3          # far from one-hot
    assert _md(sim, rp) < 2e-6
    assert _md(out, ro) < (1e-5 if h < 100 else 1e-4) * float(ro.abs().max())      # up to 4L products per output


@pytest.mark.parametrize("bf16", [False, True], ids=["f32", "bf16"])
@pytest.mark.parametrize("shape", [(2, 16, 12), (2, 16, 16), (1, 24, 40), (2, 64, 64), (1, 128, 128), (1, 132, 136), (1, 20, 248)],
                         ids=lambda s: "%dx%dx%d" % s)
def test_op_attention_fused_streaming_pass(eng, shape, bf16, seopt):
    """The fused form of the two streaming passes (att2_stats_kernel + att2_ptilde*_kernel: P is never written) against the
    oracle AND against the three-pass form, forced on at every size: 16x12 (wc = 6: element-load kernel), 16x16 / 24x40 /
    64x64 / 128x128 (wc % 4 == 0: the LDS-staged kernel of round 4 -- 64x64 and 128x128 are the 256x256- and 512x512-input
    sizes; 24x40 has wc = 20, 128x128 wc = 64 = the last width of the four-stage variant), 132x136 (wc = 68: three-stage
    variant, hc = 66: ragged last patch row, rows that do not fit in registers: two-sweep statistics), 20x248 (wc = 124, the
    widest the LDS-staged kernel takes).  The round-3 streaming kernel (SE_ATT_PTILDE_LDS=0) must agree to rounding: the two
    kernels share the expression and the summation order; so must the round-3 statistics kernel (SE_ATT_STATS_LDS=0; the
    LDS-staged one keeps an online maximum / sum per lane) and, in bf16 mode, E kept in fp32 (SE_ATT_E16=0; fp16 by default
    there).  Soft (non-saturated) scores, mixed key validity."""
    from oracle import sketchedit_oracle as O
    B, h, w = shape
    x = 0.004 * synth.uniform(5, "att96s.x%d" % h, (B, 96, h, w), -1, 1)
    full = (synth.uniform(5, "att96s.m%d" % h, (B, 1, 4 * h, 4 * w), 0, 1) < 0.5).astype(np.float32)
    full[0, :, :, 2 * w:] = 1.0
    seopt.set("SE_ATT_FUSED", "1")
    seopt.set("SE_ATT_FUSED_BF16", "1")
    fused = eng.attention(_cuda(x), _cuda(full), bf16=bf16)
    seopt.set("SE_ATT_PTILDE_LDS", "0")
    fused_r3 = eng.attention(_cuda(x), _cuda(full), bf16=bf16)
    seopt.unset("SE_ATT_PTILDE_LDS")
    if not bf16:
        assert _md(fused, fused_r3) <= 1e-5 * float(fused.abs().max())      # (P~ differs by 1 ulp here and there: fma contraction)
    seopt.set("SE_ATT_E16", "0")
    fused_e32 = eng.attention(_cuda(x), _cuda(full), bf16=bf16)               # bf16 mode: E in fp32 instead of fp16
    seopt.unset("SE_ATT_E16")
    seopt.set("SE_ATT_STATS_LDS", "0")
    fused_s3 = eng.attention(_cuda(x), _cuda(full), bf16=bf16)                # round-3 statistics kernel (and fp32 E)
    seopt.unset("SE_ATT_STATS_LDS")
    seopt.set("SE_ATT_FUSED", "0")
    three = eng.attention(_cuda(x), _cuda(full), bf16=bf16)
    if bf16:
        ro, _ = O.contextual_attention(torch.from_numpy(x).to(torch.bfloat16).float(), torch.from_numpy(full), torch.bfloat16)
        tol = 2.0 ** -7 * float(ro.abs().max())
        assert _md(fused, ro) < tol and _md(fused, three) < tol
        for other in (fused_r3, fused_e32, fused_s3):
            assert _md(other, ro) < tol and _md(fused, other) < tol
    else:
        ro, _ = O.contextual_attention(torch.from_numpy(x), torch.from_numpy(full))
        tol = (1e-5 if h < 100 else 1e-4) * float(ro.abs().max())
        assert _md(fused, ro) < tol
        assert _md(fused, three) < 1e-5 * float(ro.abs().max())       # v_exp_f32 vs expf: ~1e-7 relative per probability
        assert _md(fused, fused_e32) == 0.0                           # (the switch only exists in bf16 mode)
        assert _md(fused_s3, ro) < tol and _md(fused, fused_s3) < 1e-5 * float(ro.abs().max())


def test_op_attention_fp16_scores_saturate_instead_of_nan(eng, seopt):
    """in bf16 mode the LDS-staged passes read the doubly-centred scores E as fp16.  The procedural weight sets keep
    |E| at a few units; a real checkpoint need not.  With activations scaled so that centred scores pass 65504 the plain
    fp16 conversion gave inf, and the passes formed inf - inf = NaN for whole rows; the store now saturates.  Asserted here:
    the output is finite, and it agrees with the fp32-E form (SE_ATT_E16=0) on most pixels (both are one-hot rows there; a
    saturated row may split a tie among clamped keys differently -- it is finite, no more is claimed)."""
    B, h, w = 1, 16, 16
    x = synth.uniform(23, "att16sat.x", (B, 96, h, w), -1, 1).astype(np.float32)
    full = np.zeros((B, 1, 4 * h, 4 * w), np.float32)
    full[:, :, :, : 2 * w] = 1.0                                         # left half is the hole: its keys are invalid
    small = eng.attention(_cuda(x), _cuda(full), bf16=True)
    assert torch.isfinite(small).all()
    big = eng.attention(_cuda(x * 3e4), _cuda(full), bf16=True)          # E ~ 3e4 * 384 * O(1/16): far beyond 65504
    assert torch.isfinite(big).all()
    seopt.set("SE_ATT_E16", 0)
    big32 = eng.attention(_cuda(x * 3e4), _cuda(full), bf16=True)
    assert torch.isfinite(big32).all()
    same = (big - big32).abs() <= 1e-2 * big32.abs().max()
    assert float(same.float().mean()) > 0.5


@pytest.mark.parametrize("shape", [(2, 16, 16), (2, 24, 40), (1, 72, 64), (1, 132, 136)], ids=lambda s: "%dx%dx%d" % s)
def test_op_attention_symmetric_score_tiles(eng, shape, seopt):
    """E[r][s] = sum_c x[r][c] x[s][c] rn[c] is symmetric, and in fp32 mode att2_pair_kernel computes only the tiles on / right
    of the diagonal and stores their transposes (se_attention.hip).  With soft scores (every key matters: softmax far from
    one-hot) the probabilities must match the oracle and the all-tiles form (SE_ATT_SYM=0) -- a tile missing from the
    enumeration, or a transposed store that lands in the wrong place, changes whole 64 x 256 blocks of E.  16x16: one tile;
    24x40 (R = 240): the 128 x 32 tile shape; 72x64 (R = 1152: 5 x 18 tiles, ragged last query tile); 132x136 (R = 4488)."""
    from oracle import sketchedit_oracle as O
    B, h, w = shape
    x = 0.004 * synth.uniform(5, "att96y.x%d" % h, (B, 96, h, w), -1, 1)
    full = (synth.uniform(5, "att96y.m%d" % h, (B, 1, 4 * h, 4 * w), 0, 1) < 0.5).astype(np.float32)
    out, sim = eng.attention(_cuda(x), _cuda(full), want_similar=True)
    seopt.set("SE_ATT_SYM", "0")
    out0, sim0 = eng.attention(_cuda(x), _cuda(full), want_similar=True)
    ro, rp = O.contextual_attention(torch.from_numpy(x), torch.from_numpy(full))
    if h < 100:
        assert float(rp.max()) < 0.5                                      # far from one-hot
    assert _md(sim, rp) < 2e-6 and _md(sim0, rp) < 2e-6 and _md(sim, sim0) < 2e-6
    tol = (1e-5 if h < 100 else 1e-4) * float(ro.abs().max())
    assert _md(out, ro) < tol and _md(out, out0) < tol


def test_op_attention_vs_oracle(eng):
    from oracle import sketchedit_oracle as O
    x = synth.uniform(5, "att96.x", (2, 96, 12, 16), -1, 1)
    full = (synth.uniform(5, "att96.m", (2, 1, 48, 64), 0, 1) < 0.6).astype(np.float32)
    full[1, :, :24] = 1.0
    full[0, :, 0:16, 0:16] = 1.0
    out, sim = eng.attention(_cuda(x), _cuda(full), want_similar=True)
    ro, rp = O.contextual_attention(torch.from_numpy(x), torch.from_numpy(full))
    ms = torch.nn.functional.avg_pool2d(torch.from_numpy(full), 4, 4)
    valid = torch.nn.functional.unfold(1 - ms, 4, stride=2).mean(1)
    assert (valid <= 0.1).any() and (valid > 0.1).any()
    assert _md(sim, rp) < 1e-5
    assert _md(out, ro) < TOL_OP


def test_op_attention_all_invalid(eng):
    from oracle import sketchedit_oracle as O
    x = synth.uniform(5, "att96b.x", (1, 96, 8, 8), -1, 1)
    ones = np.ones((1, 1, 32, 32), np.float32)
    out, sim = eng.attention(_cuda(x), _cuda(ones), want_similar=True)
    ro, rp = O.contextual_attention(torch.from_numpy(x), torch.from_numpy(ones))
    assert _md(sim, rp) < 1e-6
    assert _md(out, ro) < TOL_OP


@pytest.mark.parametrize("form", ["default", "fused", "three-pass", "all-tiles"])
def test_op_attention_threshold_boundary_golden(eng, golden_dir, form, seopt):
    """The key-validity boundary (key windows with exactly 24 / 25 / 26 / 27 non-hole pixels of 256 around th = 0.1,
    /root/reference/models/networks/splitcam.py:49-53,90) against vectors of the REFERENCE's cam_1 / cam_2 modules
    (tests/golden/make_golden.py --round6, ops_r6.npz).  Soft scores: a key on the wrong side of the threshold moves
    `similar` by 0.26.  Every form of the HIP attention: the production choice, the fused streaming passes forced on, the
    three-pass form, the all-tiles score GEMM."""
    g = _load(golden_dir, "ops_r6.npz")
    x = 0.004 * synth.uniform(11, "att_th.x", (2, 96, 12, 16), -1, 1)
    full = np.unpackbits(g["op.att_th.mask_bits"])[: 2 * 48 * 64].reshape(2, 1, 48, 64).astype(np.float32)
    if form == "fused":
        seopt.set("SE_ATT_FUSED", 1)
    elif form == "three-pass":
        seopt.set("SE_ATT_FUSED", 0)
    elif form == "all-tiles":
        seopt.set("SE_ATT_SYM", 0)
    out = eng.attention(_cuda(x), _cuda(full))                              # (similar_out forces the three-pass form)
    assert _md(out, g["op.att_th.out"]) < 1e-5 * float(np.abs(g["op.att_th.out"]).max()) + 1e-7
    out2, sim = eng.attention(_cuda(x), _cuda(full), want_similar=True)
    assert _md(sim, g["op.att_th.similar"]) < 2e-6
    assert _md(out2, g["op.att_th.out"]) < 1e-5 * float(np.abs(g["op.att_th.out"]).max()) + 1e-7


def test_netG_intermediates_64_golden(eng_w, golden_dir):
    """The reference-held intermediates of e2e_64.npz (forward hooks on netG.conv11 / cam_1 / cam_2 of the reference,
    tests/golden/make_golden.py run_case) against the HIP path's own intermediates (se_netG_forward_taps): the pooled style
    vector, the attention output as the PRODUCTION forward computes it, and `similar` from the HIP attention run on the HIP
    pmconv6 output.  netG is fed the reference's hard mask."""
    from oracle import sketchedit_oracle as O
    g = _load(golden_dir, "e2e_64.npz")
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    hard = _cuda(g["hard_mask"])
    ci, cs = _cuda(img), _cuda(sk)
    r = eng_w.netG_taps(ci, ci, hard, hard, cs, FLAGS)
    assert _md(r["coarse"], g["coarse"]) < TOL_E2E and _md(r["fine"], g["fine"]) < TOL_E2E
    assert _md(r["style_vec"], g["style_vec"]) < 1e-4
    assert _md(r["attn_out"], g["attn_out"]) < 1e-4 * max(1.0, float(np.abs(g["attn_out"]).max()))
    np.testing.assert_allclose(np.array([float(r["pmconv6"].double().sum()), float(r["pmconv6"].double().abs().sum())]),
                               g["pmconv6_sum"][:2], rtol=1e-4)
    out, sim = eng_w.attention(r["pmconv6"], hard, want_similar=True)
    assert _md(sim, g["similar"]) < 5e-4          # 49 keys, saturated softmax on depth-20 activations: 1e-3 is the path's bound
    assert _md(out, g["attn_out"]) < 1e-4 * max(1.0, float(np.abs(g["attn_out"]).max()))
    # the oracle's taps agree with the same vectors (CPU: tests/test_oracle_golden.py) -- and with the HIP ones
    taps = {}
    WG = synth.make_state_dict("G", 0)
    with torch.no_grad():
        O.netG_forward(WG, img, img, g["hard_mask"], g["hard_mask"], sk, taps=taps)
    assert _md(r["pmconv6"], taps["pmconv6"]) < 1e-4 * max(1.0, float(taps["pmconv6"].abs().max()))


@pytest.mark.parametrize("name,H", [("e2e_512.npz", 512), ("e2e_256_crops.npz", 256)], ids=["512", "256"])
def test_similar_digest_golden(eng_w, golden_dir, name, H):
    """the HIP attention at L = 3969 keys (512x512; 961 at 256x256) had only ever met the oracle.  Here its
    `similar` -- se_attention on the pmconv6 output the HIP netG itself produced for the reference's hard mask -- meets the
    REFERENCE's per-query digests (largest probability, its key index where the winner is clear, position-weighted checksum;
    tests/golden/make_golden.py similar_digest, /root/reference/models/networks/splitcam.py:57-108)."""
    from digest_util import check_similar
    g = _load(golden_dir, name)
    img, sk = synth.make_inputs(1, H, H, seed=1234)
    hard_np = np.unpackbits(g["hard_mask_bits"])[: H * H].reshape(1, 1, H, H).astype(np.float32)
    hard, ci, cs = _cuda(hard_np), _cuda(img), _cuda(sk)
    r = eng_w.netG_taps(ci, ci, hard, hard, cs, FLAGS)
    _, sim = eng_w.attention(r["pmconv6"], hard, want_similar=True)
    assert tuple(sim.shape[1:]) == ((H // 8 - 1) ** 2, H // 8 - 1, H // 8 - 1)
    check_similar(sim, g, tol=1e-3)
    # and the production (fused) attention's output against the three-pass one that produced `similar`
    out3 = eng_w.attention(r["pmconv6"], hard)
    assert _md(r["attn_out"], out3) < 1e-4 * max(1.0, float(out3.abs().max()))


@pytest.mark.parametrize("mode", [("default", dict(low_latency=False)), ("lowlat", dict(low_latency=True))], ids=["default", "lowlat"])
def test_conservative_flag_selects_the_f2x2_form_in_netM(eng_w, golden_dir, mode, seopt):
    """SE_FLAG_CONSERVATIVE (include/sketchedit_hip.h): netM's 96 -> 192 layers on F(2x2,3x3), netG unchanged -- bit for bit
    what the process-wide switch SE_WINOGRAD_F43=2 gives, now chosen per call through the ABI; against the reference's
    vectors like every other mode."""
    g = _load(golden_dir, "e2e_256_crops.npz")
    img, sk = synth.make_inputs(1, 256, 256, seed=1234)
    ci, cs = _cuda(img), _cuda(sk)
    base = eng_w.inference(ci, cs, FLAGS, visualize=True, **mode[1])
    eng_w.set_conservative(True)
    try:
        cons = eng_w.inference(ci, cs, FLAGS, visualize=True, **mode[1])
        m_cons, _ = eng_w.netM(ci, cs, want_image=False)
    finally:
        eng_w.set_conservative(False)
    seopt.set("SE_WINOGRAD_F43", 2)
    ref2 = eng_w.inference(ci, cs, FLAGS, visualize=True, **mode[1])
    for k in ("mask", "composed", "coarse", "fine"):
        assert torch.equal(cons[k], ref2[k]), k
    _digest_or_mask_only(cons, g, 256, 256)
    if not mode[1]["low_latency"]:                       # (low-latency mode runs every layer in its direct form: nothing to select)
        assert not torch.equal(cons["mask"], base["mask"])
        assert torch.equal(m_cons, cons["mask"])
    assert _md(cons["mask"], base["mask"]) < 1e-4


def test_netM_64_golden(eng_w, golden_dir):
    g = _load(golden_dir, "e2e_64.npz")
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    mask, mim = eng_w.netM(_cuda(img), _cuda(sk))
    assert _md(mask, g["mask"]) < TOL_E2E
    assert _md(mim, g["mask_image"]) < TOL_E2E


def test_netG_64_golden(eng_w, golden_dir):
    """netG is fed the reference's hard mask so the comparison is independent of threshold flips."""
    g = _load(golden_dir, "e2e_64.npz")
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    hard = _cuda(g["hard_mask"])
    ci, cs = _cuda(img), _cuda(sk)
    coarse, fine = eng_w.netG(ci, ci, hard, hard, cs, FLAGS)
    assert _md(coarse, g["coarse"]) < TOL_E2E
    assert _md(fine, g["fine"]) < TOL_E2E


MODES = [("default", dict(low_latency=False)), ("lowlat", dict(low_latency=True)),
         ("lowlat-graph", dict(low_latency=True, graph=True)), ("graph", dict(low_latency=False, graph=True))]


@pytest.mark.parametrize("mode", MODES, ids=[m[0] for m in MODES])
def test_inference_64_golden(eng_w, golden_dir, mode):
    """Every execution mode (include/sketchedit_hip.h: default kernels, low-latency shapes + concurrent branches,
    hipGraph replay) against the reference's vectors; graph modes are called three times (eager, capture, replay)."""
    g = _load(golden_dir, "e2e_64.npz")
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    for _ in range(3 if mode[1].get("graph") else 1):
        r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True, **mode[1])
    flips = int((r["hard"].cpu().numpy() != g["hard_mask"]).sum())
    assert _md(r["mask"], g["mask"]) < TOL_E2E
    assert flips == 0, "hard-mask flips: %d" % flips
    for k in ("composed", "coarse", "fine"):
        assert _md(r[k], g[k]) < TOL_E2E, k
    assert _md(r["maskim"], g["mask_image"]) < TOL_E2E


def test_inference_nonsquare_golden(eng_w, golden_dir):
    g = _load(golden_dir, "e2e_40x72.npz")
    img, sk = synth.make_inputs(1, 40, 72, seed=99)
    r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True)
    assert int((r["hard"].cpu().numpy() != g["hard_mask"]).sum()) == 0
    for k in ("composed", "mask", "coarse", "fine"):
        assert _md(r[k], g[k]) < TOL_E2E, k


@pytest.mark.parametrize("tag,flags", [("avg", 1 | 16), ("nocam", 2 | 16), ("nomaskcc", 1 | 2 | 4 | 16),
                                        ("nomaskcoarse", 1 | 2 | 8 | 16), ("nojoint", 1 | 2)])
def test_flag_variants_golden(eng_w, golden_dir, tag, flags):
    from oracle import sketchedit_oracle as O
    g = _load(golden_dir, "variants_64.npz")
    img, sk = synth.make_inputs(1, 64, 64, seed=1234)
    WM = synth.make_state_dict("M", 0)
    with torch.no_grad():
        mask, _ = O.netM_forward(WM, img, sk, want_image=False)
    hard = (mask > 0.5).float().cuda()
    ci, cs = _cuda(img), _cuda(sk)
    coarse, fine = eng_w.netG(ci, ci, hard, hard, cs, flags)
    assert _md(coarse, g[tag + ".coarse"]) < TOL_E2E
    assert _md(fine, g[tag + ".fine"]) < TOL_E2E


@pytest.mark.parametrize("mode", MODES[:3], ids=[m[0] for m in MODES[:3]])
def test_inference_256_golden(eng_w, golden_dir, mode):
    g = _load(golden_dir, "e2e_256.npz")
    img, sk = synth.make_inputs(1, 256, 256, seed=1234)
    for _ in range(3 if mode[1].get("graph") else 1):
        r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True, **mode[1])
    hard = r["hard"].cpu().numpy()
    ref_hard = np.unpackbits(g["hard_mask_bits"])[: hard.size].reshape(hard.shape).astype(np.float32)
    flips = int((hard != ref_hard).sum())
    assert flips == 0, "hard-mask flips: %d" % flips
    for k in ("composed", "mask", "coarse", "fine"):
        crop = r[k][:, :, 96:160, 96:160]
        assert _md(crop, g[k + "_crop"]) < TOL_E2E, k
        a = r[k].double().cpu().numpy()
        s = np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.min(), a.max()])
        np.testing.assert_allclose(s, g[k + "_sum"], rtol=2e-3, atol=2e-2)


def _digest_or_mask_only(r, g, H, W, prefix=""):
    """Full digest check (crops 1e-3, the north-star tolerance; a row / column sum adds up to max(H, W) per-pixel
    differences); if a logit within float noise of the 0.5 threshold flipped (<= 2 pixels), netG saw another input: then only
    the soft mask is comparable with the fixture (the composite for the GPU's own hard mask is checked against the oracle in
    test_512_parity_vs_oracle / test_full_size_properties)."""
    from digest_util import check_digest
    hard = r["hard"].cpu().numpy()
    ref_hard = np.unpackbits(g[prefix + "hard_mask_bits"])[: hard.size].reshape(hard.shape)
    flips = int((hard != ref_hard).sum())
    if flips == 0:
        return check_digest(r, g, tol=TOL_E2E, sum_atol=2e-5 * max(H, W), prefix=prefix)
    assert flips <= 2, "hard-mask flips: %d" % flips
    d = _md(r["mask"].sum(3), g[prefix + "mask_rows"])
    assert d < 2e-5 * max(H, W)
    return d


@pytest.mark.parametrize("mode", MODES[:2], ids=[m[0] for m in MODES[:2]])
@pytest.mark.parametrize("name,H", [("e2e_512.npz", 512), ("e2e_256_crops.npz", 256)], ids=["512", "256"])
def test_inference_digest_golden(eng_w, golden_dir, name, H, mode):
    """BASELINE config-3 size and the 256x256 case against the REFERENCE's outputs (tests/golden/make_golden.py --round4):
    four 64x64 crops, two of them on an image border, row / column sums of all four outputs, hard-mask bits."""
    g = _load(golden_dir, name)
    img, sk = synth.make_inputs(1, H, H, seed=1234)
    r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True, **mode[1])
    _digest_or_mask_only(r, g, H, H)


@pytest.mark.parametrize("sid", ["822", "873", "902", "11", "556", "830", "854"])
def test_bundled_samples_golden(eng_w, golden_dir, sid):
    """The other seven bundled samples of the reference (three faces, /root/reference/datasets/face_release/list.txt; three
    512x512 scenes and the 408-wide one, general_release/list.txt, test_places.sh:1-17) against the reference's outputs."""
    from digest_util import sample_inputs
    g = _load(golden_dir, "sample_%s.npz" % sid)
    img, sk = sample_inputs(g)
    r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True)
    _digest_or_mask_only(r, g, img.shape[2], img.shape[3])


@pytest.fixture(scope="module", params=["w1", "w2"])
def eng_ws(request):
    from sketchedit_amd._lib import Engine
    e = Engine(0)
    e.load_state_dict("M", synth.make_weight_set("M", request.param))
    e.load_state_dict("G", synth.make_weight_set("G", request.param))
    yield request.param, e
    e.close()


@pytest.mark.parametrize("mode", MODES[:2], ids=[m[0] for m in MODES[:2]])
def test_further_weight_sets_golden(eng_ws, golden_dir, mode):
    """Two more procedural weight sets (synth.WEIGHT_SETS: seed 1 at gain 3.6 -- outputs up to the tanh's +-1 -- and a
    heavier-tailed Laplace draw) against the reference: 64x64 B=2 in full, 256x256 B=1 as a digest."""
    ws, e = eng_ws
    g = _load(golden_dir, "weights_%s.npz" % ws)
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    r = e.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True, **mode[1])
    assert int((r["hard"].cpu().numpy() != g["hard_mask"]).sum()) == 0
    for k in ("composed", "mask", "coarse", "fine"):
        assert _md(r[k], g[k]) < TOL_E2E, k
    img, sk = synth.make_inputs(1, 256, 256, seed=1234)
    r = e.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True, **mode[1])
    _digest_or_mask_only(r, g, 256, 256, prefix="d256.")


@pytest.mark.parametrize("mode", MODES[:2], ids=[m[0] for m in MODES[:2]])
def test_inference_c1_bundled_face_golden(eng_w, golden_dir, mode):
    """BASELINE config 1: the reference's bundled 256x256 face and sketch (0.34 % dense), batch 1, against the reference's
    own outputs on them (tests/golden/c1_face.npz), in the default and the low-latency execution mode."""
    g = _load(golden_dir, "c1_face.npz")
    img = ((g["image_u8"].astype(np.float32).transpose(2, 0, 1) / 255.0) - 0.5) / 0.5
    sk = (g["sketch_u8"].astype(np.float32)[None, None] / 255.0 > 0).astype(np.float32)
    r = eng_w.inference(_cuda(img[None]), _cuda(sk), FLAGS, visualize=True, **mode[1])
    hard = r["hard"].cpu().numpy()
    ref_hard = np.unpackbits(g["hard_mask_bits"])[: hard.size].reshape(hard.shape).astype(np.float32)
    assert int((hard != ref_hard).sum()) == 0
    for k in ("composed", "mask", "coarse", "fine"):
        assert _md(r[k][:, :, 96:160, 96:160], g[k + "_crop"]) < TOL_E2E, k
        a = r[k].double().cpu().numpy()
        s_ = np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.min(), a.max()])
        np.testing.assert_allclose(s_, g[k + "_sum"], rtol=2e-3, atol=2e-2)


@pytest.mark.parametrize("ll", [False, True], ids=["default", "lowlat"])
def test_batch_shard_invariance(eng_w, ll):
    """Image k gives the same result whichever batch (position) computes it -- the property the
    multi-GPU batch sharding relies on.  Bit-identical within one execution mode."""
    img, sk = synth.make_inputs(4, 64, 64, seed=77)
    full = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, low_latency=ll)
    part = eng_w.inference(_cuda(img[2:3]), _cuda(sk[2:3]), FLAGS, low_latency=ll)
    assert torch.equal(full["composed"][2:3], part["composed"])
    assert torch.equal(full["mask"][2:3], part["mask"])


def test_engine_with_only_netM_weights(golden_dir):
    """A ctx that holds netM's weights only (a caller that wants the mask predictor alone): se_workspace_bytes plans a dry
    run of BOTH networks, and the 4-channel form of netG's wconv1 used to have an all-zero layer definition until netG's
    weights arrived -- a division by zero (SIGFPE) in the planner.  netM must run; netG must refuse with a message."""
    from sketchedit_amd._lib import Engine, SketchEditHipError
    g = _load(golden_dir, "e2e_64.npz")
    e = Engine(0)
    try:
        e.load_state_dict("M", synth.make_state_dict("M", 0))
        assert not e.weights_ready()
        img, sk = synth.make_inputs(2, 64, 64, seed=1234)
        mask, _ = e.netM(_cuda(img), _cuda(sk), want_image=False)
        assert _md(mask, g["mask"]) < TOL_E2E
        with pytest.raises(SketchEditHipError, match="weights not loaded"):
            e.inference(_cuda(img), _cuda(sk), FLAGS)
    finally:
        e.close()


@pytest.mark.parametrize("mode", ["default", "lowlat", "bf16"])
def test_large_batch_is_split_into_passes(eng_w, seopt, mode):
    """VERDICT r4 item 5 / 'Next round' 3.  The kernels address a tensor with 32-bit BYTE offsets (sentinel 0x80000000), so a
    forward may put at most 2^31 bytes of its largest activation (96 bytes per pixel in fp32, 48 in bf16) into one launch;
    a larger batch runs as passes of the same plan over image ranges.  With the byte range lowered to three 64x64 images
    (SE_TEST_OFFSET_LIMIT) a batch of 7 runs as 3 + 3 + 1: every output of every entry point is bit-identical to the
    unsplit call, and a single image beyond the range is an error."""
    from sketchedit_amd._lib import SketchEditHipError
    ll = mode == "lowlat"
    eng_w.set_precision("bf16" if mode == "bf16" else "f32")
    try:
        img, sk = synth.make_inputs(7, 64, 64, seed=123)
        ci, cs = _cuda(img), _cuda(sk)
        whole = eng_w.inference(ci, cs, FLAGS, visualize=True, low_latency=ll)
        whole_u8 = eng_w.inference_u8(ci, cs, FLAGS, low_latency=ll)
        whole_pk = eng_w.inference_packed(ci, cs, FLAGS, torch.empty((7, 4, 64, 64), device="cuda"), low_latency=ll).clone()
        wm, wmi = eng_w.netM(ci, cs)
        wc, wf = eng_w.netG(ci, ci, whole["hard"], whole["hard"], cs, FLAGS)
        per_px = 48 if mode == "bf16" else 96
        seopt.set("SE_TEST_OFFSET_LIMIT", 3 * 64 * 64 * per_px + 1)
        part = eng_w.inference(ci, cs, FLAGS, visualize=True, low_latency=ll)
        for k in ("composed", "mask", "hard", "maskim", "coarse", "fine"):
            assert torch.equal(whole[k], part[k]), k
        pu8 = eng_w.inference_u8(ci, cs, FLAGS, low_latency=ll)
        assert torch.equal(whole_u8[0], pu8[0]) and torch.equal(whole_u8[1], pu8[1])
        ppk = eng_w.inference_packed(ci, cs, FLAGS, torch.empty((7, 4, 64, 64), device="cuda"), low_latency=ll)
        assert torch.equal(whole_pk, ppk)
        pm, pmi = eng_w.netM(ci, cs)
        assert torch.equal(wm, pm) and torch.equal(wmi, pmi)
        pc, pf = eng_w.netG(ci, ci, whole["hard"], whole["hard"], cs, FLAGS)
        assert torch.equal(wc, pc) and torch.equal(wf, pf)
        # and image 5 (second image of the second... third pass: position 5 = pass 1, index 2) equals its single-image result
        one = eng_w.inference(ci[5:6].contiguous(), cs[5:6].contiguous(), FLAGS, low_latency=ll)
        assert torch.equal(one["composed"], part["composed"][5:6]) and torch.equal(one["mask"], part["mask"][5:6])
        # ONE image beyond the byte range: an error, never a wrong answer
        seopt.set("SE_TEST_OFFSET_LIMIT", 64 * 64 * per_px)
        with pytest.raises(SketchEditHipError, match="32-bit byte offsets"):
            eng_w.inference(ci[:1].contiguous(), cs[:1].contiguous(), FLAGS, low_latency=ll)
    finally:
        eng_w.set_precision("f32")


def test_batch_beyond_2_gib_of_activation_256(eng_w):
    """The same at the real limit (no test aid): at 256x256 the 24-channel full-resolution activations of B = 341 images are
    2,145,386,496 bytes -- the largest batch ONE pass may hold (B = 342 crosses 2^31) -- and B = 352 runs as 176 + 176.  In both
    calls the first, the last and the images either side of the pass boundary equal their single-image results bit for bit
    (default execution mode, as a full batch runs it)."""
    img1, sk1 = synth.make_inputs(8, 256, 256, seed=2024)
    for B in (341, 352):
        idx = np.arange(B) % 8
        ci, cs = _cuda(img1[idx]), _cuda(sk1[idx])
        out = eng_w.inference_packed(ci, cs, FLAGS, torch.empty((B, 4, 256, 256), device="cuda"), low_latency=False)
        for k in (0, 175, 176, B - 1):
            j = int(idx[k])
            one = eng_w.inference(_cuda(img1[j:j + 1]), _cuda(sk1[j:j + 1]), FLAGS, low_latency=False)
            assert torch.equal(out[k:k + 1, 0:3], one["composed"]), (B, k)
            assert torch.equal(out[k:k + 1, 3:4], one["mask"]), (B, k)
        del out, ci, cs
        torch.cuda.empty_cache()


def test_batch_beyond_2_gib_of_activation_256_bf16(eng_w):
    """the bf16 mode at ITS real limit -- 48 bytes per pixel of the largest activation, 682 images of 256x256 in
    one pass (2,145,386,496 bytes), 684 as 342 + 342 -- first, last and the images either side of the pass boundary equal
    their single-image results bit for bit."""
    img1, sk1 = synth.make_inputs(6, 256, 256, seed=2025)
    eng_w.set_precision("bf16")
    try:
        for B in (682, 684):
            idx = np.arange(B) % 6
            ci, cs = _cuda(img1[idx]), _cuda(sk1[idx])
            out = eng_w.inference_packed(ci, cs, FLAGS, torch.empty((B, 4, 256, 256), device="cuda"), low_latency=False)
            for k in (0, 341, 342, B - 1):
                j = int(idx[k])
                one = eng_w.inference(_cuda(img1[j:j + 1]), _cuda(sk1[j:j + 1]), FLAGS, low_latency=False)
                assert torch.equal(out[k:k + 1, 0:3], one["composed"]), (B, k)
                assert torch.equal(out[k:k + 1, 3:4], one["mask"]), (B, k)
            del out, ci, cs
            torch.cuda.empty_cache()
    finally:
        eng_w.set_precision("f32")


def test_per_op_call_beyond_2_gib_is_an_error(eng):
    """The per-op entry points do not split: a 24-channel fp32 source of 342 x 256 x 256 pixels (2^31 + 4.2 MB bytes) is
    refused with a message, where the round-4 guards (which counted ELEMENTS) let the kernel run with a wrapped
    num_records."""
    from sketchedit_amd._lib import SketchEditHipError
    a = 1.5 / np.sqrt(24 * 9)
    w = synth.uniform(3, "big.w", (24, 24, 3, 3), -a, a)
    b = synth.uniform(3, "big.b", (24,), -0.3, 0.3)
    x = torch.zeros((342, 24, 256, 256), device="cuda")
    with pytest.raises(SketchEditHipError, match="2\\^31 bytes"):
        eng.gated_conv2d(x, w, b)
    del x
    torch.cuda.empty_cache()


def test_shard_of_a_global_batch_runs_in_the_global_mode(eng_w):
    """20 x 128x128 is a default-mode call, its two 10-image shards would each select the low-latency mode by their own
    size: sharded_inference pins the mode of the GLOBAL batch, so the shard results are the unsharded ones, bit for bit."""
    from sketchedit_amd import shard
    img, sk = synth.make_inputs(20, 128, 128, seed=31)
    ci, cs = _cuda(img), _cuda(sk)
    assert eng_w.is_low_latency(10, 128, 128) and not eng_w.is_low_latency(20, 128, 128)
    full = eng_w.inference(ci, cs, FLAGS)

    def fwd(i, s, low_latency=None):
        return eng_w.inference_packed(i, s, FLAGS, torch.empty((i.shape[0], 4, 128, 128), device="cuda"), low_latency=low_latency)
    mode = shard.global_mode(20, 128, 128)
    assert mode is False
    for lo in (0, 10):
        part = fwd(ci[lo:lo + 10].contiguous(), cs[lo:lo + 10].contiguous(), low_latency=mode)
        assert torch.equal(part[:, 0:3], full["composed"][lo:lo + 10]) and torch.equal(part[:, 3:4], full["mask"][lo:lo + 10])
    comp, mask = shard.sharded_inference(fwd, ci, cs)       # world size 1 here: same code path, one shard
    assert torch.equal(comp, full["composed"]) and torch.equal(mask, full["mask"])


def test_graph_mode_refuses_caller_outputs_and_survives_cache_eviction(eng_w):
    """`out=` cannot be honoured by a graph replay (error, not silence); the library's graph cache evicts ONE least
    recently used entry when full (17+ distinct argument sets), draining its stream first, and the hot entry keeps working."""
    from sketchedit_amd._lib import SketchEditHipError
    img, sk = synth.make_inputs(1, 32, 32, seed=9)
    ci, cs = _cuda(img), _cuda(sk)
    out = {"composed": torch.empty((1, 3, 32, 32), device="cuda"), "mask": torch.empty((1, 1, 32, 32), device="cuda")}
    with pytest.raises(SketchEditHipError):
        eng_w.inference(ci, cs, FLAGS, out=out, graph=True)
    want = eng_w.inference(ci, cs, FLAGS, low_latency=True)
    hot = [eng_w.inference(ci, cs, FLAGS, low_latency=True, graph=True) for _ in range(3)][-1]      # captured on the 2nd call
    assert torch.equal(hot["composed"], want["composed"])
    # 20 more argument sets (other shapes -> other static buffers -> other keys), each seen twice so that it is captured
    for k in range(20):
        i2, s2 = synth.make_inputs(1, 16, 16 + 8 * k, seed=k)
        for _ in range(2):
            eng_w.inference(_cuda(i2), _cuda(s2), FLAGS, low_latency=True, graph=True)
        if k % 5 == 0:                         # the hot entry stays in use: LRU keeps it
            hot = eng_w.inference(ci, cs, FLAGS, low_latency=True, graph=True)
            assert torch.equal(hot["composed"], want["composed"])
    hot = eng_w.inference(ci, cs, FLAGS, low_latency=True, graph=True)
    assert torch.equal(hot["composed"], want["composed"]) and torch.equal(hot["mask"], want["mask"])


def test_graph_replay_follows_new_inputs(eng_w):
    """A captured forward is replayed on whatever the (stable) input buffers hold: three different inputs through the
    same graph give exactly the eager results of the same mode."""
    for seed in (1, 2, 3, 4):
        img, sk = synth.make_inputs(1, 64, 64, seed=seed)
        a = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, low_latency=True, graph=True)
        a = {k: v.clone() for k, v in a.items()}
        b = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, low_latency=True)
        assert torch.equal(a["composed"], b["composed"]) and torch.equal(a["mask"], b["mask"]), seed


def test_full_size_properties(eng_w):
    """BASELINE config 2 size (256x256, B=32): finite outputs, mask in (0,1), composite identity and
    per-image agreement with a B=1 run of the same image (size-independent properties)."""
    img, sk = synth.make_inputs(32, 256, 256, seed=1234)
    ci, cs = _cuda(img), _cuda(sk)
    r = eng_w.inference(ci, cs, FLAGS, visualize=True)
    for k in ("composed", "mask", "fine", "coarse"):
        assert torch.isfinite(r[k]).all(), k
    assert float(r["mask"].min()) >= 0 and float(r["mask"].max()) <= 1
    comp = r["fine"] * r["mask"] + ci * (1 - r["mask"])
    assert float((comp - r["composed"]).abs().max()) < 1e-6
    one = eng_w.inference(ci[5:6].contiguous(), cs[5:6].contiguous(), FLAGS, low_latency=False)
    assert torch.equal(one["composed"], r["composed"][5:6])
    # the low-latency mode runs other kernels for the same layers: same result to fp32 rounding
    fast = eng_w.inference(ci[5:6].contiguous(), cs[5:6].contiguous(), FLAGS, low_latency=True, visualize=True)
    assert float((fast["mask"] - r["mask"][5:6]).abs().max()) < 1e-4
    # both modes against the oracle's composite for the hard mask each of them really used (never skipped)
    from oracle import sketchedit_oracle as O
    WM, WG = synth.make_state_dict("M", 0), synth.make_state_dict("G", 0)
    ref = O.inference(WM, WG, img[5:6], sk[5:6])
    for got, hard in ((fast["composed"], fast["hard"]), (r["composed"][5:6], r["hard"][5:6])):
        want, _ = composed_for_hard_mask(O, WG, img[5:6], sk[5:6], ref["mask"], ref["hard_mask"], ref["composed"], hard)
        assert _md(got, want) < TOL_E2E
    if torch.equal(fast["hard"], r["hard"][5:6]):
        assert float((fast["composed"] - r["composed"][5:6]).abs().max()) < 1e-4


def test_errors(eng_w):
    from sketchedit_amd._lib import SketchEditHipError
    img, sk = synth.make_inputs(1, 60, 64, seed=1)     # H not a multiple of 8
    with pytest.raises(SketchEditHipError):
        eng_w.inference(_cuda(img), _cuda(sk), FLAGS)
    with pytest.raises(SketchEditHipError):
        eng_w.load_state_dict("G", {"nonexistent.weight": np.zeros((1, 1, 3, 3), np.float32)})
    with pytest.raises(SketchEditHipError):
        eng_w.load_state_dict("G", {"conv1.weight": np.zeros((48, 4, 5, 5), np.float32)})


SMALL_SIZES = [(1, 16, 16), (2, 24, 16), (1, 16, 40), (3, 32, 32), (1, 104, 88)]


@pytest.mark.parametrize("case", SMALL_SIZES, ids=["%dx%dx%d" % c for c in SMALL_SIZES])
def test_small_and_odd_sizes_vs_oracle(eng_w, case):
    """Minimum size (16x16: one attention key), non-square multiples of 8 whose quarter-resolution grids are odd
    (Winograd kernels fall back per layer) and tile counts that do not fill a workgroup -- end to end vs the oracle."""
    from oracle import sketchedit_oracle as O
    B, H, W = case
    img, sk = synth.make_inputs(B, H, W, seed=5)
    WM, WG = synth.make_state_dict("M", 0), synth.make_state_dict("G", 0)
    ref = O.inference(WM, WG, img, sk)
    r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True)
    assert _md(r["mask"], ref["mask"]) < TOL_E2E
    hard = ref["hard_mask"].cuda()
    ci, cs = _cuda(img), _cuda(sk)
    coarse, fine = eng_w.netG(ci, ci, hard, hard, cs, FLAGS)
    assert _md(coarse, ref["coarse"]) < TOL_E2E
    assert _md(fine, ref["fine"]) < TOL_E2E
    want, _ = composed_for_hard_mask(O, WG, img, sk, ref["mask"], ref["hard_mask"], ref["composed"], r["hard"])
    assert _md(r["composed"], want) < TOL_E2E


def test_512_parity_vs_oracle(eng_w):
    """BASELINE config 3 resolution (512x512, L = 3969 attention keys): one image against the oracle."""
    from oracle import sketchedit_oracle as O
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    img, sk = synth.make_inputs(1, 512, 512, seed=1234)
    WM, WG = synth.make_state_dict("M", 0), synth.make_state_dict("G", 0)
    ref = O.inference(WM, WG, img, sk)
    r = eng_w.inference(_cuda(img), _cuda(sk), FLAGS, visualize=True)
    assert 0.1 < float(ref["hard_mask"].mean()) < 0.9
    assert _md(r["mask"], ref["mask"]) < TOL_E2E
    flips = int((r["hard"].cpu() != ref["hard_mask"]).sum())
    # netG given the oracle's hard mask (independent of threshold flips)
    hard = ref["hard_mask"].cuda()
    ci, cs = _cuda(img), _cuda(sk)
    coarse, fine = eng_w.netG(ci, ci, hard, hard, cs, FLAGS)
    assert _md(coarse, ref["coarse"]) < TOL_E2E
    assert _md(fine, ref["fine"]) < TOL_E2E
    assert flips <= 2, "hard-mask flips at 512x512: %d" % flips
    # a pixel whose logit sits within float noise of the threshold may flip (one does at this size with the procedural
    # weights): the end-to-end composite is checked against the oracle's netG run on the hard mask the GPU pipeline used
    want, _ = composed_for_hard_mask(O, WG, img, sk, ref["mask"], ref["hard_mask"], ref["composed"], r["hard"])
    assert _md(r["composed"], want) < TOL_E2E


def test_512_batch8_properties(eng_w):
    """BASELINE config 3 size (512x512, B=8): finite, composite identity, shard invariance."""
    img, sk = synth.make_inputs(8, 512, 512, seed=99)
    ci, cs = _cuda(img), _cuda(sk)
    r = eng_w.inference(ci, cs, FLAGS, visualize=True)
    for k in ("composed", "mask", "fine", "coarse"):
        assert torch.isfinite(r[k]).all(), k
    comp = r["fine"] * r["mask"] + ci * (1 - r["mask"])
    assert float((comp - r["composed"]).abs().max()) < 1e-6
    one = eng_w.inference(ci[3:4].contiguous(), cs[3:4].contiguous(), FLAGS, low_latency=False)
    assert torch.equal(one["composed"], r["composed"][3:4])
