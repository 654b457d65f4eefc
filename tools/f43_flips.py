#!/usr/bin/env python3
"""Hard-mask flips and soft-mask error of netM under the four settings of SE_WINOGRAD_F43 (run on the GPU box).

netM's soft mask feeds `mask > 0.5` (editline2_model.py:346-347), so rounding error in netM's Winograd layers can flip a
pixel of netG's INPUT.  The hybrid F(2,3) x F(4,3) kernel (se_wino24.hip) has non-dyadic constants and about twice the
rounding error of F(2x2,3x3); this tool measures what that costs, per mode
    0  F(2x2,3x3) everywhere          1  hybrid everywhere
    2  hybrid in netG only            (3: hybrid everywhere except netM's mask decoder -- measured in round 5: as many flips
                                       as mode 1, removed from the library)
against the fp32 oracle: flips of the hard mask, max / mean |soft mask error|, and the mean error over the pixels whose
oracle value lies in [0.4, 0.6] (the ones a larger error could flip), over the three procedural weight sets and N seeded
inputs per set at 256x256 (+ N/4 at 512x512).   usage: python tools/f43_flips.py [N=16] > f43_flips.json
"""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import sketchedit_oracle as O  # noqa: E402
from sketchedit_amd import _lib, synth  # noqa: E402
from sketchedit_amd.hostinfo import effective_cpus  # noqa: E402


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    torch.set_num_threads(min(32, effective_cpus()))
    res = {m: {"flips": 0, "pixels": 0, "max_abs": 0.0, "sum_abs": 0.0, "near": 0, "near_sum_abs": 0.0, "near_max_abs": 0.0} for m in (0, 1, 2)}
    per_set = {}
    for ws in sorted(synth.WEIGHT_SETS):
        WM = synth.make_weight_set("M", ws)
        eng = _lib.Engine(0)
        eng.load_state_dict("M", WM)
        WMt = {k: torch.from_numpy(v) for k, v in WM.items()}
        for size, cnt in ((256, n), (512, max(1, n // 4))):
            for b0 in range(0, cnt, 4):
                nb = min(4, cnt - b0)
                img, sk = synth.make_inputs(nb, size, size, seed=9000 + size, first_index=b0)
                ref = O.netM_forward(WMt, torch.from_numpy(img), torch.from_numpy(sk), want_image=False)[0].numpy()
                near = np.abs(ref - 0.5) < 0.1
                for m in res:
                    _lib.set_option("SE_WINOGRAD_F43", m)
                    got = eng.netM(torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda(), want_image=False)[0].cpu().numpy()
                    err = np.abs(got.astype(np.float64) - ref.astype(np.float64))
                    r = res[m]
                    fl = int(((got > 0.5) != (ref > 0.5)).sum())
                    r["flips"] += fl
                    r["pixels"] += ref.size
                    r["max_abs"] = max(r["max_abs"], float(err.max()))
                    r["sum_abs"] += float(err.sum())
                    r["near"] += int(near.sum())
                    r["near_sum_abs"] += float(err[near].sum())
                    r["near_max_abs"] = max(r["near_max_abs"], float(err[near].max()) if near.any() else 0.0)
                    per_set.setdefault("%s/%d" % (ws, size), {}).setdefault(m, [0, 0.0])
                    per_set["%s/%d" % (ws, size)][m][0] += fl
                    per_set["%s/%d" % (ws, size)][m][1] = max(per_set["%s/%d" % (ws, size)][m][1], float(err.max()))
        eng.close()
    _lib.reset_options()
    out = {}
    for m, r in res.items():
        out[str(m)] = {"hard_mask_flips": r["flips"], "pixels": r["pixels"], "max_abs_soft_mask": r["max_abs"],
                       "mean_abs_soft_mask": r["sum_abs"] / max(r["pixels"], 1), "pixels_within_0.1_of_threshold": r["near"],
                       "mean_abs_near_threshold": r["near_sum_abs"] / max(r["near"], 1), "max_abs_near_threshold": r["near_max_abs"]}
    print(json.dumps({"modes": out, "per_weight_set_and_size": {k: {str(m): v for m, v in d.items()} for k, d in per_set.items()},
                      "inputs_per_set": {"256": n, "512": max(1, n // 4)}}, indent=1))


if __name__ == "__main__":
    main()
