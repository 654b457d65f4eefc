"""Shared by the GPU parity tests: the composite the oracle gives FOR THE HARD MASK THE GPU PIPELINE USED.

`mask_inpaint = (mask > 0.5)` (reference models/editline2_model.py:347) turns fp32 noise on a logit that sits on the
threshold into a different INPUT of netG.  Instead of skipping the composite comparison when a pixel flipped (a guard
that can pass vacuously), every test compares against this: the oracle's own composite when the hard masks agree,
otherwise the oracle's netG re-run on the GPU's hard mask and composited with the oracle's soft mask (:132)."""
import numpy as np
import torch


def _t(a):
    return a.detach().cpu().float() if hasattr(a, "detach") else torch.from_numpy(np.ascontiguousarray(a, np.float32))


def composed_for_hard_mask(O, WG, img, sk, ref_mask, ref_hard, ref_composed, gpu_hard, max_flips=2, **netg_kw):
    """-> (reference composite tensor, flips).  Fails when more than `max_flips` pixels thresholded differently."""
    gpu_hard, ref_hard = _t(gpu_hard), _t(ref_hard)
    flips = int((gpu_hard != ref_hard).sum())
    assert flips <= max_flips, "hard-mask flips: %d" % flips
    if flips == 0:
        return _t(ref_composed), 0
    img, sk, ref_mask = _t(img), _t(sk), _t(ref_mask)
    with torch.no_grad():
        _, fine = O.netG_forward(WG, img, img, gpu_hard, gpu_hard, sk, **netg_kw)
    return fine * ref_mask + img * (1 - ref_mask), flips
