#!/usr/bin/env python3
"""Developer aid: per-phase cycle trace of the Winograd kernels (block 0 / wave 0, s_memtime stamps).

    python tools/wino_trace.py build      # cross-compiles tools/_build/libse_trace.so with -DSE_WINO_TRACE (no GPU needed)
    python tools/wino_trace.py run        # on the GPU box: runs one 96->192 3x3 layer (B=32, 64x64) and prints the tables
"""
import ctypes
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SO = os.path.join(ROOT, "tools", "_build", "libse_trace.so")


def build():
    sys.path.insert(0, ROOT)
    from sketchedit_amd import _lib
    os.makedirs(os.path.dirname(SO), exist_ok=True)
    srcs = [os.path.join(_lib.CSRC, s) for s in _lib.SOURCES]
    subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-shared", "-DSE_WINO_TRACE",
                           "-o", SO] + srcs)
    print("built", SO)


def table(t, names, nstamp, nit=48):
    """t: [2 waves][48 iterations][8 stamps]; prints both waves' phase durations and their start offset."""
    t0 = t[0, 0, 0]
    print("it   " + "  ".join("%-9s" % n for n in names) + " total | wave 4: start " + "  ".join("%-9s" % n for n in names) + " total")
    for it in range(nit):
        row = "%2d   " % it
        for wv in range(2):
            d = [t[wv, it, k + 1] - t[wv, it, k] for k in range(nstamp - 1)]
            if wv == 1:
                row += " |         %6d " % (t[1, it, 0] - t0)
            row += "  ".join("%-9d" % v for v in d) + " %5d" % (t[wv, it, nstamp - 1] - t[wv, it, 0])
        print(row)
    last = nit - 2
    print("iterations 0..%d: %d cycles, mean per iteration %.0f" % (last, t[0, last, nstamp - 1] - t[0, 0, 0], (t[0, last, nstamp - 1] - t[0, 0, 0]) / (last + 1.0)))


def run():
    os.environ["SKETCHEDIT_HIP_LIB"] = SO
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sketchedit_amd import synth
    from sketchedit_amd._lib import Engine
    a = 1.5 / np.sqrt(96 * 9)
    w = synth.uniform(1, "t.w", (192, 96, 3, 3), -a, a)
    b = synth.uniform(1, "t.b", (192,), -0.1, 0.1)
    x = torch.from_numpy(synth.uniform(1, "t.x", (32, 96, 64, 64), -1, 1)).cuda()
    lib = ctypes.CDLL(SO)
    buf = (ctypes.c_ulonglong * (96 * 8))()
    eng = Engine(0)
    for _ in range(3):
        eng.gated_conv2d(x, w, b)
    torch.cuda.synchronize()
    assert lib.se_debug_wino_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 48, 8).astype(np.int64)
    print("== wino_kernel (se_wino.hip)")
    table(t, ["6mfma", "24mfma", "xw+18mfma", "barrier"], 5)


def run48():
    os.environ["SKETCHEDIT_HIP_LIB"] = SO
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sketchedit_amd import synth
    from sketchedit_amd._lib import Engine
    a = 1.5 / np.sqrt(48 * 9)
    w = synth.uniform(1, "t.w", (96, 48, 3, 3), -a, a)
    b = synth.uniform(1, "t.b", (96,), -0.1, 0.1)
    x = torch.from_numpy(synth.uniform(1, "t.x", (32, 48, 128, 128), -1, 1)).cuda()
    lib = ctypes.CDLL(SO)
    buf = (ctypes.c_ulonglong * (96 * 8))()
    eng = Engine(0)
    for _ in range(3):
        eng.gated_conv2d(x, w, b)
    torch.cuda.synchronize()
    assert lib.se_debug_wino48_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 48, 8).astype(np.int64)
    print("== wino48_kernel (se_wino48.hip)")
    table(t, ["24mfma", "fold/xwrite", "24mfma+ld", "barrier"], 5, nit=24)
    for wv in range(2):
        e = t[wv, 24]
        print("wave %d: prologue %d  loop %d  epilogue %d cycles" % (wv * 4, e[1] - e[0], e[2] - e[1], e[3] - e[2]))


def run24():
    """hybrid F(2,3) x F(4,3) kernel (se_wino24.hip): 72 iterations of 24 MFMAs per wave; slot 72 = kernel phases"""
    os.environ["SKETCHEDIT_HIP_LIB"] = SO
    os.environ["SE_WINOGRAD_F43"] = "1"
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sketchedit_amd import synth
    from sketchedit_amd._lib import Engine
    a = 1.5 / np.sqrt(96 * 9)
    w = synth.uniform(1, "t.w", (192, 96, 3, 3), -a, a)
    b = synth.uniform(1, "t.b", (192,), -0.1, 0.1)
    x = torch.from_numpy(synth.uniform(1, "t.x", (32, 96, 64, 64), -1, 1)).cuda()
    lib = ctypes.CDLL(SO)
    buf = (ctypes.c_ulonglong * (2 * 80 * 8))()
    eng = Engine(0)
    for _ in range(3):
        eng.gated_conv2d(x, w, b)
    torch.cuda.synchronize()
    assert lib.se_debug_wino24_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 80, 8).astype(np.int64)
    print("== wino24_kernel (se_wino24.hip)")
    table(t, ["fold+12mfma", "wait/xform", "12mfma+ld", "barrier"], 5, nit=72)
    for wv in range(2):
        e = t[wv, 72]
        print("wave %d: prologue %d  loop %d  epilogue %d cycles (s_memtime ticks: 100 MHz)" % (wv * 4, e[1] - e[0], e[2] - e[1], e[3] - e[2]))


def runw2():
    """two-dimensional Winograd kernel of the 24 -> 24 layers (se_rtilew.hip rtilew2_kernel): phases of the first 16 blocks"""
    os.environ["SKETCHEDIT_HIP_LIB"] = SO
    sys.path.insert(0, ROOT)
    import numpy as np
    import torch
    from sketchedit_amd import synth
    from sketchedit_amd._lib import Engine
    a = 1.5 / np.sqrt(24 * 9)
    w = synth.uniform(1, "t.w", (24, 24, 3, 3), -a, a)
    b = synth.uniform(1, "t.b", (24,), -0.1, 0.1)
    x = torch.from_numpy(synth.uniform(1, "t.x", (32, 24, 256, 256), -1, 1)).cuda()
    lib = ctypes.CDLL(SO)
    buf = (ctypes.c_ulonglong * (2 * 16 * 8))()
    eng = Engine(0)
    for _ in range(3):
        eng.gated_conv2d(x, w, b)
    torch.cuda.synchronize()
    assert lib.se_debug_rtilew2_trace(buf) == 0
    t = np.frombuffer(buf, dtype=np.uint64).reshape(2, 16, 8).astype(np.int64)
    names = ["gather", "mfma", "epilogue", "barrier1", "transform", "barrier2"]
    print("== rtilew2_kernel: block  " + "  ".join("%-9s" % n for n in names) + " total  | wave 4 ...")
    for it in range(16):
        row = "%2d   " % it
        for wv in range(2):
            dd = [t[wv, it, k + 1] - t[wv, it, k] for k in range(6)]
            row += "  ".join("%-9d" % v for v in dd) + " %6d  | " % (t[wv, it, 6] - t[wv, it, 0])
        print(row)


if __name__ == "__main__":
    {"build": build, "run": run, "run48": run48, "run24": run24, "runw2": runw2}[sys.argv[1]]()
