#!/usr/bin/env python3
"""Randomised end-to-end parity sweep (run on the GPU box): random batch / height / width (multiples of 8), both
precisions, default and low-latency execution, against the oracle.  tests/test_gpu_fuzz.py runs 60 cases of it; run more after kernel changes (500 cases:
0 failures, worst 4.0e-6 fp32 / 1.5e-2 bf16 at the end of round 2; round 3: 700 cases, 0 failures, 4.4e-6 / 1.6e-2; final round-3 build: 1000 cases, 0 failures, 4.2e-6 / 1.7e-2).   usage: python tools/fuzz_sizes.py [n_cases] [seed]
Round 4 (final build, fp16 E + LDS-staged attention passes in bf16 mode): 700 cases, seed 33: 1 bf16 failure (w1, 136x48, a
width the three-pass form takes: 1.53 x the triangle against the 1.5 bound) -- the same case fails the same way with fp32 E
(SE_ATT_E16=0), and the distribution of (distance from the bf16 oracle) / (that oracle's distance from the fp32 oracle) is the
same with either: mean 0.89 / 0.93 / 0.99, 90th percentile 1.05 / 1.10 / 1.16 for w0 / w1 / w2.  The bound is statistical: about
one random case in 500 of the larger-gain sets sits a few % above it.  SE_FUZZ_ONLY=k,k re-runs single cases of a sweep.
"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sketchedit_oracle as O  # noqa: E402
from sketchedit_amd import synth  # noqa: E402
from sketchedit_amd._lib import Engine  # noqa: E402

FLAGS = 1 | 2 | 16


def run(n, seed, verbose=True):
    rng = np.random.RandomState(seed)
    torch.set_num_threads(min(32, os.cpu_count() or 1))
    # every case draws one of the procedural weight sets (synth.WEIGHT_SETS: two seeds / gains of the uniform draw and a
    # heavier-tailed Laplace draw): one engine per (weight set, precision)
    names = sorted(synth.WEIGHT_SETS)
    wts = {ws: (synth.make_weight_set("M", ws), synth.make_weight_set("G", ws)) for ws in names}
    engs = {}
    for ws in names:
        for prec in ("f32", "bf16"):
            e = Engine(0)
            e.load_state_dict("M", wts[ws][0])
            e.load_state_dict("G", wts[ws][1])
            if prec == "bf16":
                e.set_precision("bf16")
            engs[ws, prec] = e
    worst = {"f32": 0.0, "bf16": 0.0}
    ratios = {}                      # bf16: distance from the bf16 oracle / the bf16 oracle's own distance from the fp32 oracle
    bad = 0
    t0 = time.time()
    for k in range(n):
        B = int(rng.randint(1, 4))
        H, W = 8 * int(rng.randint(2, 20)), 8 * int(rng.randint(2, 20))
        prec = "bf16" if k % 3 == 2 else "f32"
        ll = bool(rng.randint(0, 2))
        ws = names[int(rng.randint(0, len(names)))]
        WM, WG = wts[ws]
        only = os.environ.get("SE_FUZZ_ONLY")                   # "533,17": re-run single cases of a sweep (same random draws)
        if only and str(k) not in only.split(","):
            if k % 4 == 3:
                [rng.randint(0, 2) for _ in range(5)]           # (the flag draws of the skipped case)
            continue
        img, sk = synth.make_inputs(B, H, W, seed=100 + k)
        # one case in four: a random combination of the option flags (editline_g.py:15-23) instead of test_celeb.sh's
        fl = dict(use_cam=True, pool_type="max", no_mask_cc=False, no_mask_coarse=False, joint_train_inp=True)
        if k % 4 == 3:
            fl = dict(use_cam=bool(rng.randint(0, 2)), pool_type=("max", "avg")[int(rng.randint(0, 2))], no_mask_cc=bool(rng.randint(0, 2)),
                      no_mask_coarse=bool(rng.randint(0, 2)), joint_train_inp=bool(rng.randint(0, 2)))
        bits = (1 if fl["use_cam"] else 0) | (2 if fl["pool_type"] == "max" else 0) | (4 if fl["no_mask_cc"] else 0) | \
            (8 if fl["no_mask_coarse"] else 0) | (16 if fl["joint_train_inp"] else 0)
        ref = O.inference(WM, WG, img, sk, act_dtype=torch.bfloat16 if prec == "bf16" else None, **fl)
        e = engs[ws, prec]
        ci, cs = torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda()
        r = e.inference(ci, cs, bits, visualize=True, low_latency=ll)
        hard = ref["hard_mask"].cuda()
        coarse, fine = e.netG(ci, ci, hard, hard, cs, bits)
        dm = float((r["mask"].cpu() - ref["mask"]).abs().max())
        dc = float((coarse.cpu() - ref["coarse"]).abs().max())
        df = float((fine.cpu() - ref["fine"]).abs().max())
        tol = 1e-3
        if prec == "bf16":
            # the comparator is self-defined (the reference has no reduced precision): two valid placements of the same bf16
            # roundings differ by 3e-2 at the default gain, and by more under the larger-gain weight set -- there the bound is
            # the error triangle of tests/test_gpu_bf16.py: no further from the bf16 oracle than 1.25 x the bf16 oracle's own
            # distance from the fp32 oracle
            r32 = O.inference(WM, WG, img, sk, **fl)
            _, f32g = O.netG_forward(WG, *[torch.from_numpy(a) for a in (img, img)], ref["hard_mask"], ref["hard_mask"], torch.from_numpy(sk), **fl)
            tri = max(float((ref["mask"] - r32["mask"]).abs().max()), float((ref["fine"] - f32g).abs().max()))
            # 1.25 is what tests/test_gpu_bf16.py holds the default weight set to; the larger-gain / heavier-tailed sets amplify
            # the placement of a rounding more (400-case sweep: worst ratio 1.27 on w1), they get 1.5
            tol = max(3e-2, (1.25 if ws == "w0" else 1.5) * tri)
            ratios.setdefault(ws, []).append(max(dm, dc, df) / max(tri, 1e-30))
        ok = dm < tol and dc < tol and df < tol and np.isfinite(dm + dc + df)
        worst[prec] = max(worst[prec], dm, dc, df)
        bad += 0 if ok else 1
        if verbose or not ok:
            print("%2d %-4s %s B=%d %3dx%-3d ll=%d fl=%-2d mask %.2e coarse %.2e fine %.2e tol %.2e %s" % (k, prec, ws, B, H, W, ll, bits, dm, dc, df, tol, "ok" if ok else "FAIL"), flush=True)
    for e in engs.values():
        e.close()
    print("cases %d  failures %d  worst f32 %.2e  worst bf16 %.2e  (%.0f s)" % (n, bad, worst["f32"], worst["bf16"], time.time() - t0))
    for ws_ in sorted(ratios):
        r_ = np.array(ratios[ws_])
        print("  bf16 error triangle, weight set %s: %d cases, ratio mean %.3f  p90 %.3f  max %.3f" % (ws_, len(r_), r_.mean(), np.percentile(r_, 90), r_.max()))
    return bad, worst


def main():
    bad, _ = run(int(sys.argv[1]) if len(sys.argv) > 1 else 24, int(sys.argv[2]) if len(sys.argv) > 2 else 7)
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
