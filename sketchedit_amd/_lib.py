"""ctypes binding of libsketchedit_hip.so (C-ABI in include/sketchedit_hip.h).

This is the thin layer a maintainer of the reference would add: torch owns the
device tensors, this module passes their raw pointers + the current HIP stream to the library.
There is NO fallback: if the library is missing or fails, an exception is raised.
"""
import ctypes
import os
import subprocess
import threading

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SKETCHEDIT_HIP_LIB points at another build of the same library (developer builds, e.g. tools/wino_trace.py)
LIB_PATH = os.environ.get("SKETCHEDIT_HIP_LIB") or os.path.join(_HERE, "lib", "libsketchedit_hip.so")
CSRC = os.path.join(_HERE, "csrc")
SOURCES = ["se_gconv.hip", "se_rconv16.hip", "se_rconv96.hip", "se_rtile.hip", "se_rtilew.hip", "se_wino.hip", "se_wino24.hip", "se_wino48.hip", "se_wino_up.hip", "se_wino_up48.hip", "se_attention.hip", "se_misc.hip",
           "se_api.hip"]

SE_NET_G, SE_NET_M = 0, 1
FLAG_USE_CAM, FLAG_POOL_MAX, FLAG_NO_MASK_CC, FLAG_NO_MASK_COARSE, FLAG_JOINT_TRAIN_INP = 1, 2, 4, 8, 16
FLAG_LOW_LATENCY, FLAG_GRAPH, FLAG_PACKED_OUT, FLAG_BF16, FLAG_CONSERVATIVE = 32, 64, 128, 256, 512   # execution options (include/sketchedit_hip.h)
# Which calls run in the low-latency mode unless the caller says otherwise: at most three 256x256 images' worth of pixels, or ONE
# image of up to 512x512.  Re-measured on MI355X in round 6 (tools/ll_threshold.sh), low-latency vs default, after the default
# mode learned to overlap netG's branches and the low-latency mode to keep the Winograd kernels where ONE image's grid still
# covers the chip: 256x256 B = 1 / 2 / 3 / 4: 1.20 / 1.75 / 2.45 / 3.05 ms vs 2.33 / 2.43 / 2.54 / 2.66; 384x384 B = 1: 2.02 vs
# 2.59; 512x512 B = 1 / 2: 2.84 / 3.86 vs 2.96 / 3.80
LOW_LATENCY_MAX_PIXELS = 3 * 256 * 256
LOW_LATENCY_MAX_SINGLE_IMAGE = 512 * 512

# every symbol declared in include/sketchedit_hip.h
SYMBOLS = ["se_create", "se_destroy", "se_last_error", "se_version", "se_load_weights", "se_weights_ready",
           "se_workspace_bytes", "se_netM_forward", "se_netM_forward_ex", "se_netG_forward", "se_netG_forward_taps", "se_inference", "se_inference_u8", "se_gated_conv2d",
           "se_gated_conv2d_ex", "se_attention", "se_attention_ex", "se_quantize_u8", "se_dequantize_u8", "se_inference_u8io", "se_profile_enable",
           "se_profile_report", "se_debug_set_option", "se_debug_get_option", "se_debug_reset_options"]


class SketchEditHipError(RuntimeError):
    pass


class NetGTaps(ctypes.Structure):
    """se_netG_taps (include/sketchedit_hip.h): optional intermediate outputs of netG, device pointers or NULL"""
    _fields_ = [("pmconv6", ctypes.c_void_p), ("attn_out", ctypes.c_void_p), ("style_vec", ctypes.c_void_p)]


def build_library(force=False, verbose=False, extra_flags=()):
    """hipcc --offload-arch=gfx950 -> sketchedit_amd/lib/libsketchedit_hip.so (cross-compiles without a GPU).
    One object per source, compiled in parallel; an object is rebuilt when its source or a header is newer OR when its
    command line changed (flags / -D macros: the command is kept beside the object), then one link.  The whole build holds
    an exclusive file lock, so concurrent builders (ranks, test workers) do not write the same objects."""
    import fcntl
    from concurrent.futures import ThreadPoolExecutor
    hdrs = [os.path.join(CSRC, "se_kernels.h"), os.path.join(CSRC, "se_device.h"), os.path.join(_HERE, "..", "include", "sketchedit_hip.h")]
    hdr_t = max(os.path.getmtime(h) for h in hdrs)
    objdir = os.path.join(_HERE, "lib", "obj")
    os.makedirs(objdir, exist_ok=True)
    with open(os.path.join(_HERE, "lib", ".build.lock"), "w") as lock:
        fcntl.flock(lock, fcntl.LOCK_EX)
        jobs, objs = [], []
        for src in SOURCES:
            sp, op = os.path.join(CSRC, src), os.path.join(objdir, src.replace(".hip", ".o"))
            objs.append(op)
            cmd = ["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + list(extra_flags) + ["-c", sp, "-o", op]
            try:
                with open(op + ".cmd") as f:
                    same_cmd = f.read() == " ".join(cmd)
            except OSError:
                same_cmd = False
            if force or not same_cmd or not os.path.exists(op) or os.path.getmtime(op) < max(os.path.getmtime(sp), hdr_t):
                jobs.append(cmd)
        if not jobs and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(o) for o in objs):
            return LIB_PATH

        def run(cmd):
            if verbose:
                print(" ".join(cmd), flush=True)
            subprocess.check_call(cmd)
            if "-c" in cmd:
                with open(cmd[-1] + ".cmd", "w") as f:
                    f.write(" ".join(cmd))
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 1) or 1) as ex:
            list(ex.map(run, jobs))
        run(["hipcc", "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB_PATH + ".tmp"] + objs)
        os.replace(LIB_PATH + ".tmp", LIB_PATH)          # readers never see a half-written library
    return LIB_PATH


_lib = None
_lib_lock = threading.Lock()


def load_library():
    global _lib
    with _lib_lock:
        if _lib is not None:
            return _lib
        if not os.path.exists(LIB_PATH):
            raise SketchEditHipError(
                "%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(there is no CPU or PyTorch fallback for this path)" % LIB_PATH)
        # PyTorch first: it brings its own HIP runtime (torch/lib/libamdhip64.so) and our library must bind to that one.
        # Loaded the other way round the process ends up with two HIP runtimes and hipGetDeviceCount() returns 0.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        c_f = ctypes.c_void_p  # device/host float pointers are passed as raw addresses
        vp, ci, sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t
        lib.se_create.argtypes = [ci, ctypes.POINTER(vp)]
        lib.se_create.restype = ci
        lib.se_destroy.argtypes = [vp]
        lib.se_destroy.restype = None
        lib.se_last_error.argtypes = [vp]
        lib.se_last_error.restype = ctypes.c_char_p
        lib.se_version.argtypes = []
        lib.se_version.restype = ctypes.c_char_p
        lib.se_load_weights.argtypes = [vp, ci, ctypes.c_char_p, c_f, ctypes.POINTER(ci), ci]
        lib.se_load_weights.restype = ci
        lib.se_weights_ready.argtypes = [vp]
        lib.se_weights_ready.restype = ci
        lib.se_workspace_bytes.argtypes = [vp, ci, ci, ci]
        lib.se_workspace_bytes.restype = sz
        lib.se_netM_forward.argtypes = [vp, vp, c_f, c_f, c_f, c_f, vp, sz, ci, ci, ci]
        lib.se_netM_forward.restype = ci
        lib.se_netM_forward_ex.argtypes = [vp, vp, c_f, c_f, c_f, c_f, vp, sz, ci, ci, ci, ci]
        lib.se_netM_forward_ex.restype = ci
        lib.se_netG_forward.argtypes = [vp, vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, vp, sz, ci, ci, ci, ci]
        lib.se_netG_forward.restype = ci
        lib.se_netG_forward_taps.argtypes = [vp, vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, vp, sz, ci, ci, ci, ci, ctypes.POINTER(NetGTaps)]
        lib.se_netG_forward_taps.restype = ci
        lib.se_inference.argtypes = [vp, vp, c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_f, vp, sz, ci, ci, ci, ci]
        lib.se_inference.restype = ci
        lib.se_inference_u8.argtypes = [vp, vp, c_f, c_f, vp, vp, vp, sz, ci, ci, ci, ci]
        lib.se_inference_u8.restype = ci
        lib.se_gated_conv2d.argtypes = [vp, vp, c_f, c_f, c_f, c_f] + [ci] * 10
        lib.se_gated_conv2d.restype = ci
        lib.se_gated_conv2d_ex.argtypes = [vp, vp, c_f, c_f, ci, c_f, c_f, c_f] + [ci] * 12
        lib.se_gated_conv2d_ex.restype = ci
        lib.se_attention.argtypes = [vp, vp, c_f, c_f, c_f, c_f, ci, ci, ci]
        lib.se_attention.restype = ci
        lib.se_attention_ex.argtypes = [vp, vp, c_f, c_f, c_f, c_f, ci, ci, ci, ci]
        lib.se_attention_ex.restype = ci
        lib.se_quantize_u8.argtypes = [vp, vp, c_f, c_f, vp, vp, ci, ci, ci]
        lib.se_quantize_u8.restype = ci
        lib.se_dequantize_u8.argtypes = [vp, vp, vp, vp, c_f, c_f, ci, ci, ci]
        lib.se_dequantize_u8.restype = ci
        lib.se_inference_u8io.argtypes = [vp, vp, vp, vp, vp, vp, vp, sz, ci, ci, ci, ci]
        lib.se_inference_u8io.restype = ci
        lib.se_profile_enable.argtypes = [vp, ci]
        lib.se_profile_enable.restype = ci
        lib.se_profile_report.argtypes = [vp, ctypes.c_char_p, sz]
        lib.se_profile_report.restype = ci
        lib.se_debug_set_option.argtypes = [ctypes.c_char_p, ci]
        lib.se_debug_set_option.restype = ci
        lib.se_debug_get_option.argtypes = [ctypes.c_char_p, ctypes.POINTER(ci)]
        lib.se_debug_get_option.restype = ci
        lib.se_debug_reset_options.argtypes = []
        lib.se_debug_reset_options.restype = None
        _lib = lib
        return lib


_shared = {}
_shared_lock = threading.Lock()


def shared_engine(device=0):
    """A process-wide Engine per device for the per-op entry points (no weights involved)."""
    with _shared_lock:
        if device not in _shared:
            _shared[device] = Engine(device)
        return _shared[device]


def set_option(name, value):
    """Developer switch of the library, process-wide, effective from the next call on.  The library
    reads its switches from the environment ONCE; tests and A/B tools change them afterwards through this call."""
    if load_library().se_debug_set_option(name.encode(), int(value)):
        raise SketchEditHipError("unknown developer switch %r" % name)


def get_option(name):
    v = ctypes.c_int(0)
    if load_library().se_debug_get_option(name.encode(), ctypes.byref(v)):
        raise SketchEditHipError("unknown developer switch %r" % name)
    return v.value


def reset_options():
    load_library().se_debug_reset_options()


def flags_from_opt(opt):
    """netG flag word from a reference-style options namespace (editline_g.py:15-23, base_options.py:19)."""
    f = 0
    if getattr(opt, "use_cam", False):
        f |= FLAG_USE_CAM
    pool = getattr(opt, "pool_type", "avg")
    if pool == "max":
        f |= FLAG_POOL_MAX
    elif pool != "avg":
        raise NotImplementedError(pool)        # editline_g.py:164-165
    if getattr(opt, "no_mask_cc", False):
        f |= FLAG_NO_MASK_CC
    if getattr(opt, "no_mask_coarse", False):
        f |= FLAG_NO_MASK_COARSE
    if getattr(opt, "joint_train_inp", False):
        f |= FLAG_JOINT_TRAIN_INP
    return f


def _ptr(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


def _check_dev(*ts):
    import torch
    for t in ts:
        if t is None:
            continue
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.float32 and t.is_contiguous()):
            raise SketchEditHipError("expected contiguous float32 CUDA(HIP) tensors")


def _check_dev_u8(*ts):
    import torch
    for t in ts:
        if not (isinstance(t, torch.Tensor) and t.is_cuda and t.dtype == torch.uint8 and t.is_contiguous()):
            raise SketchEditHipError("expected contiguous uint8 CUDA(HIP) tensors")


class Engine:
    """One se_ctx on one GPU + its workspace.  Thread-safe (the library serialises forwards per ctx)."""

    def __init__(self, device=0):
        import torch
        if not torch.cuda.is_available():
            raise SketchEditHipError("no HIP device visible: the sketchedit_amd forward only runs on an MI355X")
        self.lib = load_library()
        self.device = int(device)
        h = ctypes.c_void_p()
        if self.lib.se_create(self.device, ctypes.byref(h)) != 0:
            raise SketchEditHipError("se_create: " + self.lib.se_last_error(None).decode())
        self.h = h
        self._ws = None
        self._ws_lock = threading.Lock()
        self._ws_stream = None
        self._graph_stream = None
        self.precision = "f32"     # "bf16": BASELINE config 5 (bf16 storage + MFMA, fp32 accumulate); see set_precision
        self.conservative = False  # SE_FLAG_CONSERVATIVE on every forward: netM's 96 -> 192 layers on F(2x2,3x3) (set_conservative)
        self._static = {}          # graph mode: per-shape input copies and output buffers (stable pointers)
        self._graph_lock = threading.Lock()    # graph mode is single-caller per Engine: one replay at a time

    def close(self):
        if getattr(self, "h", None):
            self.lib.se_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _err(self, what):
        raise SketchEditHipError("%s: %s" % (what, self.lib.se_last_error(self.h).decode()))

    # ---- weights -------------------------------------------------------------------------------
    def load_state_dict(self, net, state_dict):
        """net in {'G','M'}; state_dict: key -> array/tensor in checkpoint layout (strict)."""
        net_id = SE_NET_G if net == "G" else SE_NET_M
        for k, v in state_dict.items():
            a = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
            a = np.ascontiguousarray(a, dtype=np.float32)
            shape = (ctypes.c_int * a.ndim)(*a.shape)
            if self.lib.se_load_weights(self.h, net_id, k.encode(), a.ctypes.data_as(ctypes.c_void_p), shape, a.ndim):
                self._err("se_load_weights(%s)" % k)

    def weights_ready(self):
        return bool(self.lib.se_weights_ready(self.h))

    # ---- workspace -----------------------------------------------------------------------------
    def workspace(self, B, H, W):
        import torch
        need = self.lib.se_workspace_bytes(self.h, B, H, W)
        if need == 0:
            self._err("se_workspace_bytes")
        with self._ws_lock:
            cur = torch.cuda.current_stream(self.device)
            if self._ws is None or self._ws.numel() < need:
                if self._ws is not None:
                    # earlier forwards may still be using the old block on their stream: keep the caching allocator
                    # from handing it out before they finish
                    self._ws.record_stream(self._ws_stream)
                self._ws = torch.empty(need, dtype=torch.uint8, device="cuda:%d" % self.device)
                with self._graph_lock:     # (never taken in the other order: the graph path asks for the workspace first)
                    self._static.clear()   # captured graphs are keyed by the workspace pointer
            elif self._ws_stream is not None and self._ws_stream != cur:
                # one workspace = one stream: forwards on another stream must not overlap the previous ones
                cur.wait_stream(self._ws_stream)
            self._ws_stream = cur
            return self._ws

    def _stream(self):
        import torch
        return ctypes.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)

    # ---- forwards ------------------------------------------------------------------------------
    def netM(self, image, sketch, want_image=True):
        import torch
        _check_dev(image, sketch)
        B, _, H, W = image.shape
        ws = self.workspace(B, H, W)
        mask = torch.empty((B, 1, H, W), dtype=torch.float32, device=image.device)
        mim = torch.empty((B, 3, H, W), dtype=torch.float32, device=image.device) if want_image else None
        if self.lib.se_netM_forward_ex(self.h, self._stream(), _ptr(image), _ptr(sketch), _ptr(mask), _ptr(mim),
                                       _ptr(ws), ws.numel(), B, H, W,
                                       (FLAG_BF16 if self.precision == "bf16" else 0) | (FLAG_CONSERVATIVE if self.conservative else 0)):
            self._err("se_netM_forward_ex")
        return mask, mim

    def netG(self, x, x2, mask, mask2, guide, flags):
        import torch
        _check_dev(x, x2, mask, mask2, guide)
        B, _, H, W = x.shape
        ws = self.workspace(B, H, W)
        coarse = torch.empty((B, 3, H, W), dtype=torch.float32, device=x.device)
        fine = torch.empty_like(coarse)
        flags = (flags & 31) | (FLAG_BF16 if self.precision == "bf16" else 0)
        if self.lib.se_netG_forward(self.h, self._stream(), _ptr(x), _ptr(x2), _ptr(mask), _ptr(mask2), _ptr(guide),
                                    _ptr(coarse), _ptr(fine), _ptr(ws), ws.numel(), B, H, W, flags):
            self._err("se_netG_forward")
        return coarse, fine

    def netG_taps(self, x, x2, mask, mask2, guide, flags):
        """netG with its intermediate outputs (se_netG_forward_taps; the counterparts of the forward hooks
        tests/golden/make_golden.py puts on the reference): -> dict(coarse, fine, pmconv6, attn_out, style_vec).  The
        attention runs in the form the production forward uses."""
        import torch
        _check_dev(x, x2, mask, mask2, guide)
        B, _, H, W = x.shape
        ws = self.workspace(B, H, W)
        mk = lambda *shape: torch.empty(shape, dtype=torch.float32, device=x.device)     # noqa: E731
        r = dict(coarse=mk(B, 3, H, W), fine=mk(B, 3, H, W), pmconv6=mk(B, 96, H // 4, W // 4), style_vec=mk(B, 96))
        if flags & FLAG_USE_CAM:
            r["attn_out"] = mk(B, 96, H // 4, W // 4)
        taps = NetGTaps(r["pmconv6"].data_ptr(), r["attn_out"].data_ptr() if "attn_out" in r else None, r["style_vec"].data_ptr())
        flags = (flags & 31) | (FLAG_BF16 if self.precision == "bf16" else 0)
        if self.lib.se_netG_forward_taps(self.h, self._stream(), _ptr(x), _ptr(x2), _ptr(mask), _ptr(mask2), _ptr(guide),
                                         _ptr(r["coarse"]), _ptr(r["fine"]), _ptr(ws), ws.numel(), B, H, W, flags, ctypes.byref(taps)):
            self._err("se_netG_forward_taps")
        return r

    def set_conservative(self, on=True):
        """SE_FLAG_CONSERVATIVE (include/sketchedit_hip.h): netM -- whose soft mask feeds the hard 0.5 threshold -- keeps the
        F(2x2,3x3) Winograd form; netG keeps the hybrid one.  +1.7 % per step at 256x256 batch 32."""
        self.conservative = bool(on)

    def set_precision(self, precision):
        """'f32' (default; the north star's 1e-3 parity bound applies) or 'bf16' (SE_FLAG_BF16 on every forward)."""
        if precision not in ("f32", "bf16"):
            raise ValueError(precision)
        self.precision = precision

    @staticmethod
    def is_low_latency(B, H, W, low_latency=None):
        """The low-latency mode is chosen by call size unless forced (True / False)."""
        if low_latency is not None:
            return bool(low_latency)
        return B * H * W <= LOW_LATENCY_MAX_PIXELS or (B == 1 and H * W <= LOW_LATENCY_MAX_SINGLE_IMAGE)

    def exec_flags(self, B, H, W, low_latency=None, graph=False):
        """Execution-option bits for a call."""
        return (FLAG_LOW_LATENCY if self.is_low_latency(B, H, W, low_latency) else 0) | (FLAG_GRAPH if graph else 0) | \
            (FLAG_BF16 if self.precision == "bf16" else 0) | (FLAG_CONSERVATIVE if self.conservative else 0)

    def inference(self, image, sketch, flags, visualize=False, out=None, low_latency=None, graph=False):
        """-> dict(composed, mask[, hard, maskim, coarse, fine]).  `out` may hold preallocated composed/mask.

        low_latency: None = by size (small calls), True / False = forced.  graph=True replays the forward from a
        captured hipGraph: that needs stable pointers, so the inputs are copied into buffers this Engine keeps per
        shape and the returned tensors ARE the per-shape output buffers -- consume them before the next call of the
        same shape."""
        import torch
        _check_dev(image, sketch)
        if graph and out is not None:
            raise SketchEditHipError("graph=True replays into buffers the Engine keeps per shape: `out=` cannot be honoured "
                                     "(copy from the returned tensors, or call without graph=True)")
        B, _, H, W = image.shape
        ws = self.workspace(B, H, W)
        dev = image.device
        flags = (flags & 31) | self.exec_flags(B, H, W, low_latency, graph)

        def new_outputs(have=None):
            mk = lambda c: torch.empty((B, c, H, W), dtype=torch.float32, device=dev)     # noqa: E731
            o = dict(composed=have["composed"], mask=have["mask"]) if have else dict(composed=mk(3), mask=mk(1))
            if visualize:
                o.update(hard=mk(1), maskim=mk(3), coarse=mk(3), fine=mk(3))
            return o

        if graph:
            # the per-shape input copies, the launch and the (shared) output buffers form one critical section: two threads
            # replaying the same shape would otherwise overwrite each other's inputs / read each other's outputs.  The
            # returned tensors ARE the static buffers, so a second graph call of the same shape must wait until the first
            # caller has consumed them: graph mode is documented single-caller, the lock only keeps a concurrent call from
            # corrupting a replay in flight.
            with self._graph_lock:
                return self._inference_graph(image, sketch, flags, visualize, ws, B, H, W, new_outputs)
        r = new_outputs(out)
        self._call_inference(image, sketch, r, ws, B, H, W, flags, self._stream())
        return r

    def _call_inference(self, image, sketch, r, ws, B, H, W, flags, stream):
        if self.lib.se_inference(self.h, stream, _ptr(image), _ptr(sketch), _ptr(r["composed"]), _ptr(r["mask"]),
                                 _ptr(r.get("hard")), _ptr(r.get("maskim")), _ptr(r.get("coarse")), _ptr(r.get("fine")),
                                 _ptr(ws), ws.numel(), B, H, W, flags):
            self._err("se_inference")

    def _inference_graph(self, image, sketch, flags, visualize, ws, B, H, W, new_outputs):
        import torch
        key = (B, H, W, bool(visualize))
        st = self._static.get(key)
        if st is None:
            st = self._static[key] = {"image": torch.empty_like(image), "sketch": torch.empty_like(sketch),
                                      "outs": new_outputs()}
        st["image"].copy_(image)
        st["sketch"].copy_(sketch)
        image, sketch = st["image"], st["sketch"]
        r = dict(st["outs"])
        # stream capture is not permitted on the legacy default stream: graph-mode forwards run on a stream of their
        # own, ordered after / before the caller's current stream
        cur = torch.cuda.current_stream(self.device)
        if self._graph_stream is None:
            self._graph_stream = torch.cuda.Stream(device=self.device)
        gs = self._graph_stream
        gs.wait_stream(cur)
        self._call_inference(image, sketch, r, ws, B, H, W, flags, ctypes.c_void_p(gs.cuda_stream))
        cur.wait_stream(gs)
        return r

    def inference_u8(self, image, sketch, flags, low_latency=None):
        """The forward with test.py:25-27's quantisation fused into its last kernel -> (rgb (B,H,W,3) uint8, mask (B,H,W)
        uint8): what test.py writes to disk, without an fp32 output tensor or a separate pass."""
        import torch
        _check_dev(image, sketch)
        B, _, H, W = image.shape
        ws = self.workspace(B, H, W)
        rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device=image.device)
        m8 = torch.empty((B, H, W), dtype=torch.uint8, device=image.device)
        flags = (flags & 31) | self.exec_flags(B, H, W, low_latency, False)
        if self.lib.se_inference_u8(self.h, self._stream(), _ptr(image), _ptr(sketch), _ptr(rgb), _ptr(m8), _ptr(ws),
                                    ws.numel(), B, H, W, flags):
            self._err("se_inference_u8")
        return rgb, m8

    def dequantize_u8(self, image_u8, sketch_u8):
        """data/testimage_dataset.py:89-111 on the device: (B,H,W,3) uint8 RGB, (B,H,W) uint8 'L' -> image (B,3,H,W) fp32 in
        [-1,1] = (v/255 - 0.5)/0.5, sketch (B,1,H,W) fp32 in {0,1} = (v > 0); bit-identical to the CPU dataset's tensors."""
        import torch
        _check_dev_u8(image_u8, sketch_u8)
        B, H, W, _ = image_u8.shape
        image = torch.empty((B, 3, H, W), dtype=torch.float32, device=image_u8.device)
        sketch = torch.empty((B, 1, H, W), dtype=torch.float32, device=image_u8.device)
        if self.lib.se_dequantize_u8(self.h, self._stream(), _ptr(image_u8), _ptr(sketch_u8), _ptr(image), _ptr(sketch), B, H, W):
            self._err("se_dequantize_u8")
        return image, sketch

    def inference_u8io(self, image_u8, sketch_u8, flags, low_latency=None, out=None):
        """uint8 in, uint8 out: test.py:20-37 between the PNG decoder and the PNG encoder as ONE library call
        (se_inference_u8io).  image_u8 (B,H,W,3), sketch_u8 (B,H,W) -> (rgb (B,H,W,3), mask (B,H,W)), all uint8 on the device.
        `out` = (rgb, mask) preallocated."""
        import torch
        _check_dev_u8(image_u8, sketch_u8)
        B, H, W, _ = image_u8.shape
        assert tuple(sketch_u8.shape) == (B, H, W)
        ws = self.workspace(B, H, W)
        rgb, m8 = out if out is not None else (torch.empty((B, H, W, 3), dtype=torch.uint8, device=image_u8.device),
                                               torch.empty((B, H, W), dtype=torch.uint8, device=image_u8.device))
        flags = (flags & 31) | self.exec_flags(B, H, W, low_latency, False)
        if self.lib.se_inference_u8io(self.h, self._stream(), _ptr(image_u8), _ptr(sketch_u8), _ptr(rgb), _ptr(m8), _ptr(ws),
                                      ws.numel(), B, H, W, flags):
            self._err("se_inference_u8io")
        return rgb, m8

    def inference_packed(self, image, sketch, flags, out, low_latency=None):
        """Inference into ONE (B,4,H,W) buffer `out`: planes 0-2 composed, plane 3 the soft mask -- the unit the
        batch-sharded path all-gathers (sketchedit_amd/shard.py)."""
        _check_dev(image, sketch, out)
        B, _, H, W = image.shape
        assert tuple(out.shape) == (B, 4, H, W)
        ws = self.workspace(B, H, W)
        flags = (flags & 31) | self.exec_flags(B, H, W, low_latency, False) | FLAG_PACKED_OUT
        if self.lib.se_inference(self.h, self._stream(), _ptr(image), _ptr(sketch), _ptr(out), None, None, None, None,
                                 None, _ptr(ws), ws.numel(), B, H, W, flags):
            self._err("se_inference")
        return out

    # ---- measurement -----------------------------------------------------------------------------
    def profile(self, on):
        self.lib.se_profile_enable(self.h, int(on))

    def profile_report(self):
        import json
        buf = ctypes.create_string_buffer(1 << 18)
        if self.lib.se_profile_report(self.h, buf, len(buf)):
            self._err("se_profile_report")
        return json.loads(buf.value.decode())

    # ---- per-op entry points (unit tests) --------------------------------------------------------
    def gated_conv2d(self, x, w, b, stride=1, rate=1, act="elu", upsample=False, x1=None, low_latency=False, bf16=False):
        """gen_conv / gen_deconv on x, or on the virtual concat cat([x, x1]) where x1 is a (B,C1,H,W) tensor or a
        (B,C1) per-image vector (broadcast over the image, zero padded at the borders)."""
        import torch
        _check_dev(x, x1)
        w = np.ascontiguousarray(w, np.float32)
        b = np.ascontiguousarray(b, np.float32)
        B, Cin, H, W = x.shape
        Cout, CinT, k, _ = w.shape
        Cin1 = 0 if x1 is None else x1.shape[1]
        assert CinT == Cin + Cin1
        pad = int(rate * (k - 1) / 2)
        if upsample:
            Ho, Wo = 2 * H, 2 * W
        else:
            Ho = (H + 2 * pad - rate * (k - 1) - 1) // stride + 1
            Wo = (W + 2 * pad - rate * (k - 1) - 1) // stride + 1
        raw = act is None or Cout == 3
        y = torch.empty((B, Cout if raw else Cout // 2, Ho, Wo), dtype=torch.float32, device=x.device)
        acode = {"elu": 0, "relu": 1, None: 2}[act]
        if self.lib.se_gated_conv2d_ex(self.h, self._stream(), _ptr(x), _ptr(x1), int(x1 is not None and x1.dim() == 2),
                                       w.ctypes.data_as(ctypes.c_void_p), b.ctypes.data_as(ctypes.c_void_p), _ptr(y),
                                       B, Cin, Cin1, H, W, Cout, k, stride, rate, acode, int(upsample),
                                       (FLAG_LOW_LATENCY if low_latency else 0) | (FLAG_BF16 if bf16 else 0)):
            self._err("se_gated_conv2d_ex")
        return y

    def quantize_u8(self, composed, mask):
        """test.py:25-27 on the device: ((composed+1)/2*255) -> uint8 (B,H,W,3) RGB, (mask*255) -> uint8 (B,H,W)."""
        import torch
        _check_dev(composed, mask)
        B, _, H, W = composed.shape
        rgb = torch.empty((B, H, W, 3), dtype=torch.uint8, device=composed.device)
        m8 = torch.empty((B, H, W), dtype=torch.uint8, device=composed.device)
        if self.lib.se_quantize_u8(self.h, self._stream(), _ptr(composed), _ptr(mask), _ptr(rgb), _ptr(m8), B, H, W):
            self._err("se_quantize_u8")
        return rgb, m8

    def attention(self, x, mask_full, want_similar=False, bf16=False):
        import torch
        _check_dev(x, mask_full)
        B, C, h, w = x.shape
        assert C == 96
        hs, ws = (h - 4) // 2 + 1, (w - 4) // 2 + 1
        out = torch.empty_like(x)
        sim = torch.empty((B, hs * ws, hs, ws), dtype=torch.float32, device=x.device) if want_similar else None
        if self.lib.se_attention_ex(self.h, self._stream(), _ptr(x), _ptr(mask_full), _ptr(out), _ptr(sim), B, h, w,
                                    FLAG_BF16 if bf16 else 0):
            self._err("se_attention_ex")
        return (out, sim) if want_similar else out
