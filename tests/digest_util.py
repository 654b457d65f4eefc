"""Checks against the digest fixtures of tests/golden/make_golden.py (round 4): large reference outputs are carried as four
64x64 crops (two touching an image border), per-row and per-column sums of every channel -- a localised defect anywhere
moves one of them -- the five global statistics, the hard-mask bits, and (attention) per-query digests of `similar`."""
import io

import numpy as np


def crop_boxes(H, W):
    """the (top, left) corners make_golden.py's crop_boxes() uses"""
    return [(0, 0), (H - 64, W - 64), ((H - 64) // 2, (W - 64) // 2), (min((H // 4) // 8 * 8, H - 64), min((5 * W // 8) // 8 * 8, W - 64))]


def _np(a):
    return a.detach().cpu().double().numpy() if hasattr(a, "detach") else np.asarray(a, np.float64)


def summary(a):
    a = _np(a)
    return np.array([a.sum(), np.abs(a).sum(), (a * a).sum(), a.min(), a.max()])


def sample_inputs(g):
    """(image (1,3,H,W) in [-1,1], sketch (1,1,H,W) in {0,1}) as /root/reference/data/testimage_dataset.py:89-111 builds them
    from the two PNG files the fixture carries"""
    from PIL import Image
    image = Image.open(io.BytesIO(g["image_png"].tobytes())).convert("RGB")
    w, h = image.size
    sketch = Image.open(io.BytesIO(g["sketch_png"].tobytes())).convert("L").resize((w, h))
    iu8, su8 = np.asarray(image, np.uint8), np.asarray(sketch, np.uint8)
    img = ((iu8.astype(np.float32).transpose(2, 0, 1) / 255.0) - 0.5) / 0.5
    sk = (su8.astype(np.float32)[None, None] / 255.0 > 0).astype(np.float32)
    return np.ascontiguousarray(img[None]), sk


def check_digest(out, g, tol, sum_atol, prefix="", max_flips=0):
    """out: {'composed','mask','coarse','fine','hard'} tensors / arrays of one forward; g: the fixture.
    tol: max-abs bound on the crops; sum_atol: absolute bound on a row / column sum.  Returns the worst crop difference."""
    hard = _np(out["hard"])
    ref_hard = np.unpackbits(g[prefix + "hard_mask_bits"])[: hard.size].reshape(hard.shape)
    flips = int((hard != ref_hard).sum())
    assert flips <= max_flips, "hard-mask flips: %d" % flips
    worst = 0.0
    for k in ("composed", "mask", "coarse", "fine"):
        a = _np(out[k])
        H, W = a.shape[2:]
        if prefix + k + "_crops" in g:
            for i, (t, l) in enumerate(crop_boxes(H, W)):
                d = float(np.abs(a[:, :, t:t + 64, l:l + 64] - g[prefix + k + "_crops"][i]).max())
                worst = max(worst, d)
                assert d < tol, "%s crop %d: %.3e" % (k, i, d)
        np.testing.assert_allclose(a.sum(3), g[prefix + k + "_rows"], rtol=1e-5, atol=sum_atol, err_msg=k + " row sums")
        np.testing.assert_allclose(a.sum(2), g[prefix + k + "_cols"], rtol=1e-5, atol=sum_atol, err_msg=k + " column sums")
        np.testing.assert_allclose(summary(a)[3:], g[prefix + k + "_sum"][3:], rtol=0, atol=tol, err_msg=k + " min / max")
    return worst


def check_similar(sim, g, tol, prefix=""):
    """sim: (B, L, hs, ws) softmax over the keys (splitcam.py:57-108) against the per-query digests of the fixture"""
    p = _np(sim)
    B, L = p.shape[:2]
    p = p.reshape(B, L, -1)
    wk = ((np.arange(L) * 37) % 101 / 101.0)[None, :, None]
    assert float(np.abs(p.max(1) - g[prefix + "similar_max"]).max()) < tol
    assert float(np.abs((p * wk).sum(1) - g[prefix + "similar_chk"]).max()) < tol
    clear = g[prefix + "similar_max"] > 0.6            # a clear winner: its index must agree
    assert np.array_equal(p.argmax(1)[clear], g[prefix + "similar_argmax"][clear])
    assert clear.mean() > 0.01
