#!/usr/bin/env python3
"""Validate a SketchEdit checkpoint pair against the layer tables of this implementation, without a GPU.

    python tools/check_checkpoint.py CHECKPOINTS_DIR/NAME [--epoch latest] [--npz OUT_PREFIX]

Looks for <epoch>_net_G.pth and <epoch>_net_M.pth (the layout of /root/reference/util/util.py:190-225), strips a
DataParallel 'module.' prefix (:221-222) and checks every key and shape against what `se_load_weights` accepts
(strict, like the reference's load_state_dict): 104 tensors / 5,366,430 parameters for netG (DeepFillC2Generator),
48 tensors / 2,112,820 for netM (MDGenerator).  Exit code 0 = both files would load.  `--npz` additionally writes
the validated tensors as <prefix>_G.npz / <prefix>_M.npz (fp32, prefix-free keys).
"""
import argparse
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))


def expected_shapes(net):
    from sketchedit_amd import synth
    shapes = {}
    for name, cin, cout, k in (synth.G_LAYERS if net == "G" else synth.M_LAYERS):
        shapes[name + ".weight"] = (cout, cin, k, k)
        shapes[name + ".bias"] = (cout,)
    return shapes


def check_state_dict(net, sd):
    """-> (clean dict of fp32 numpy arrays, list of problems)."""
    want = expected_shapes(net)
    clean, problems, seen = {}, [], set()
    for k, v in sd.items():
        key = k[len("module."):] if k.startswith("module.") else k
        arr = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if key not in want:
            problems.append("unexpected key %s" % k)
            continue
        seen.add(key)                       # present (possibly malformed): not "missing"
        if tuple(arr.shape) != want[key]:
            problems.append("size mismatch for %s: %s, expected %s" % (k, tuple(arr.shape), want[key]))
        elif not np.isfinite(arr).all():
            problems.append("non-finite values in %s" % k)
        else:
            clean[key] = arr.astype(np.float32)
    for key in want:
        if key not in seen:
            problems.append("missing key %s" % key)
    return clean, problems


def main(argv=None):
    ap = argparse.ArgumentParser(description=__doc__, formatter_class=argparse.RawDescriptionHelpFormatter)
    ap.add_argument("dir")
    ap.add_argument("--epoch", default="latest")
    ap.add_argument("--npz", default=None)
    args = ap.parse_args(argv)
    import torch
    bad = 0
    for net in ("G", "M"):
        path = os.path.join(args.dir, "%s_net_%s.pth" % (args.epoch, net))
        if not os.path.exists(path):
            print("%s: missing file" % path)
            bad += 1
            continue
        sd = torch.load(path, map_location="cpu")
        clean, problems = check_state_dict(net, sd)
        n = sum(int(v.size) for v in clean.values())
        print("%s: %d tensors, %d parameters, %d problem(s)" % (path, len(clean), n, len(problems)))
        for p in problems[:20]:
            print("   " + p)
        bad += len(problems)
        if args.npz and not problems:
            np.savez(args.npz + "_" + net + ".npz", **clean)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
