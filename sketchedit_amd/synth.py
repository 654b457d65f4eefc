"""Procedural (counter-based) synthetic weights and inputs.

There are no pretrained checkpoints in this environment (the reference downloads
them, /root/reference/download/download_model.sh:1-8), so every parity test,
golden fixture and benchmark uses weights generated here.  The generator is a
pure function of (seed, tensor name, element index) -- a splitmix64 hash -- so
the very same tensors can be produced in the golden-capture script (which
injects them into the imported reference), in the oracle tests, and on the GPU
box, without shipping 30 MB of weights.

Layer shapes follow the reference constructors:
  DeepFillC2Generator  /root/reference/models/networks/editline_g.py:44-100
  MDGenerator          /root/reference/models/networks/editline2_g.py:18-43
"""
import zlib

import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    """Vectorised splitmix64 finaliser on uint64 arrays."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)) & _M64
        z = x
        z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M64
        z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M64
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(seed, stream, n):
    """n float64 values in [0,1): element i depends only on (seed, stream, i)."""
    sid = np.uint64(zlib.crc32(stream.encode()) & 0xFFFFFFFF)
    base = _splitmix64(np.array([np.uint64(seed) * np.uint64(0x100000001B3) ^ sid], dtype=np.uint64))[0]
    with np.errstate(over="ignore"):
        ctr = (np.arange(n, dtype=np.uint64) + base) & _M64
    bits = _splitmix64(ctr)
    return (bits >> np.uint64(11)).astype(np.float64) * (1.0 / 9007199254740992.0)


def uniform(seed, stream, shape, lo, hi):
    n = int(np.prod(shape))
    return (lo + (hi - lo) * uniform01(seed, stream, n)).astype(np.float32).reshape(shape)


# (name, cin, cout, ksize) in constructor order.
def _g_layers():
    c = 48
    L = []
    for p in ("conv", "wconv"):
        L += [(p + "1", 5, c, 5), (p + "2_downsample", c // 2, 2 * c, 3), (p + "3", c, 2 * c, 3),
              (p + "4_downsample", c, 4 * c, 3), (p + "5", 2 * c, 4 * c, 3), (p + "6", 2 * c, 4 * c, 3),
              (p + "7_atrous", 2 * c, 4 * c, 3), (p + "8_atrous", 2 * c, 4 * c, 3),
              (p + "9_atrous", 2 * c, 4 * c, 3), (p + "10_atrous", 2 * c, 4 * c, 3)]
        if p == "conv":
            L += [("conv11", 4 * c, 4 * c, 3), ("conv12", 2 * c, 4 * c, 3),
                  ("conv13_upsample_conv", 2 * c, 2 * c, 3), ("conv14", c, 2 * c, 3),
                  ("conv15_upsample_conv", c, c, 3), ("conv16", c // 2, c // 2, 3), ("conv17", c // 4, 3, 3)]
    L += [("xconv1", 3, c, 5), ("xconv2_downsample", c // 2, c, 3), ("xconv3", c // 2, 2 * c, 3),
          ("xconv4_downsample", c, 2 * c, 3), ("xconv5", c, 4 * c, 3), ("xconv6", 2 * c, 4 * c, 3),
          ("xconv7_atrous", 2 * c, 4 * c, 3), ("xconv8_atrous", 2 * c, 4 * c, 3),
          ("xconv9_atrous", 2 * c, 4 * c, 3), ("xconv10_atrous", 2 * c, 4 * c, 3),
          ("pmconv1", 3, c, 5), ("pmconv2_downsample", c // 2, c, 3), ("pmconv3", c // 2, 2 * c, 3),
          ("pmconv4_downsample", c, 4 * c, 3), ("pmconv5", 2 * c, 4 * c, 3), ("pmconv6", 2 * c, 4 * c, 3),
          ("pmconv9", 2 * c, 4 * c, 3), ("pmconv10", 2 * c, 4 * c, 3),
          ("allconv11", 4 * c, 4 * c, 3), ("allconv12", 2 * c, 4 * c, 3),
          ("allconv13_upsample_conv", 2 * c, 2 * c, 3), ("allconv14", c, 2 * c, 3),
          ("allconv15_upsample_conv", c, c, 3), ("allconv16", c // 2, c // 2, 3), ("allconv17", c // 4, 3, 3)]
    return L


def _m_layers():
    c = 48
    L = [("conv1", 4, c, 5), ("conv2_downsample", c // 2, 2 * c, 3), ("conv3", c, 2 * c, 3),
         ("conv4_downsample", c, 4 * c, 3), ("conv5", 2 * c, 4 * c, 3), ("conv6", 2 * c, 4 * c, 3),
         ("conv7_atrous", 2 * c, 4 * c, 3), ("conv8_atrous", 2 * c, 4 * c, 3),
         ("conv9_atrous", 2 * c, 4 * c, 3), ("conv10_atrous", 2 * c, 4 * c, 3)]
    for p in ("conv", "conv_mask_"):
        L += [(p + "11", 2 * c, 4 * c, 3), (p + "12", 2 * c, 4 * c, 3),
              (p + "13_upsample_conv", 2 * c, 2 * c, 3), (p + "14", c, 2 * c, 3),
              (p + "15_upsample_conv", c, c, 3), (p + "16", c // 2, c // 2, 3),
              (p + "17", c // 4, 3 if p == "conv" else 1, 3)]
    return L


G_LAYERS = _g_layers()
M_LAYERS = _m_layers()

DEFAULT_GAIN = 3.25
# netM's last layer gets a larger gain so the mask logits spread out: with a uniform gain
# the soft mask sits in [0.48, 0.52] and thousands of pixels lie within float noise of the
# 0.5 threshold (editline2_model.py:347), which makes the hard mask a coin toss.
LAYER_GAIN = {"M.conv_mask_17": 24.0}


def laplace(seed, stream, shape, scale):
    """Laplace(0, scale) through the inverse CDF of the same counter stream (heavier tails than the uniform)."""
    n = int(np.prod(shape))
    u = uniform01(seed, stream, n) - 0.5
    return (-scale * np.sign(u) * np.log1p(-2.0 * np.abs(u))).astype(np.float32).reshape(shape)


# Named weight sets (VERDICT r3 "one weight set everywhere"): every end-to-end fixture exists for more than one
# procedural distribution, so that a value-range dependence (Winograd cancellation at large activations, softmax
# saturation, near-threshold masks) cannot hide behind one draw.  (seed, gain, dist): all inside the window in which the
# random network neither dies (no hole) nor saturates (chosen with the reference, tests/golden/make_golden.py --probe).
WEIGHT_SETS = {
    "w0": (0, DEFAULT_GAIN, "uniform"),        # rounds 1-3: every fixture, fuzz case and bench run
    "w1": (1, 3.6, "uniform"),                 # another draw at a larger gain: activations up to the tanh's +-1 (gain 4 saturates)
    "w2": (3, DEFAULT_GAIN, "laplace"),        # heavier tails: the same variance per layer concentrated in fewer, larger weights
}


def make_weight_set(net, name):
    seed, gain, dist = WEIGHT_SETS[name]
    return make_state_dict(net, seed, gain, dist)


def make_state_dict(net, seed=0, gain=DEFAULT_GAIN, dist="uniform"):
    """{'<layer>.weight': (Cout,Cin,k,k) f32, '<layer>.bias': (Cout,) f32} for net in {'G','M'}.

    Same key/shape contract as the reference checkpoints
    (/root/reference/util/util.py:214-225).  weight ~ U(-gain/sqrt(fan_in), +gain/sqrt(fan_in)),
    bias ~ U(-1/sqrt(fan_in), +1/sqrt(fan_in)).  gain 1 would be PyTorch's default
    conv init, which never opens a hole; the default gain makes
    the predicted mask straddle 0.5 so masked / unmasked paths are both exercised.
    """
    layers = G_LAYERS if net == "G" else M_LAYERS
    sd = {}
    for name, cin, cout, k in layers:
        fan_in = cin * k * k
        a = gain * LAYER_GAIN.get(net + "." + name, 1.0) / np.sqrt(fan_in)
        b = 1.0 / np.sqrt(fan_in)
        if dist == "laplace":     # variance of U(-a, a) = a^2 / 3 = 2 scale^2
            sd[name + ".weight"] = laplace(seed, "%s.%s.weight" % (net, name), (cout, cin, k, k), a / np.sqrt(6.0))
        else:
            sd[name + ".weight"] = uniform(seed, "%s.%s.weight" % (net, name), (cout, cin, k, k), -a, a)
        sd[name + ".bias"] = uniform(seed, "%s.%s.bias" % (net, name), (cout,), -b, b)
    return sd


def make_inputs(batch, height, width, seed=1234, density=0.005, first_index=0):
    """image (B,3,H,W) ~ U(-1,1); sketch (B,1,H,W) in {0,1} with the given density.

    Sample k of the batch depends only on (seed, first_index + k) so that a batch
    shard generated on another rank is identical to the same rows of the full batch.
    """
    img = np.empty((batch, 3, height, width), np.float32)
    sk = np.empty((batch, 1, height, width), np.float32)
    for k in range(batch):
        g = first_index + k
        img[k] = uniform(seed, "image.%d" % g, (3, height, width), -1.0, 1.0)
        sk[k] = (uniform01(seed + 1, "sketch.%d" % g, height * width) < density).astype(np.float32).reshape(
            1, height, width)
    return img, sk
