"""test.py's loop as an overlapped pipeline (reference: /root/reference/test.py:20-37 fed by the DataLoader workers of
/root/reference/data/__init__.py:42-48).

The reference runs  decode (worker processes) -> forward -> .cpu() -> cv2.imwrite  with the last three strictly in series on
the main thread.  With the forward at ~2 900 images/s the PNG codec is two orders slower per core, so the stages here overlap:

    decode   --nThreads DataLoader worker processes: PIL decode -> uint8 arrays (the dataset's u8 mode), collated into
             pinned host memory by the loader
    H2D      uint8 batch -> device on a copy stream (a quarter of the fp32 bytes)
    forward  se_inference_u8io on the compute stream: dequantise (table lookup), netM, threshold, netG, composite, quantise
    D2H      the uint8 results -> pinned host buffers on a third stream
    encode   PIL PNG encode + file write on a thread pool (the zlib work releases the GIL)

Streams are ordered with events only; the main thread blocks on an event just before it hands a finished batch to the
encoders (`depth` batches are in flight on the device).  Every PNG is byte-identical to what the serial loop writes: the
same arrays reach the same encoder calls, only their timing differs (tests/test_gpu_model_api.py).
"""
import os
import tempfile
import threading
import time
from collections import deque
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import torch

import atexit
import json
import selectors
import subprocess
import sys
from concurrent.futures import Future

from .png_worker import save_png  # noqa: F401  (also test.py's serial loop)


class _EncoderProcs:
    """`n` codec processes (encoders; the same class runs the decoders of InferencePipeline.run_paths)
    (`python -m sketchedit_amd.png_worker`: numpy + PIL only, no torch, no HIP state -- plain child
    processes, nothing forked from this one).  submit(job) -> Future of the seconds the job took; a job goes to the worker with
    the fewest outstanding jobs; ONE reader thread collects the completion lines from the workers' raw pipe descriptors (no
    buffered reader in between: a buffered readline() can swallow a second line that the selector then never reports)."""

    def __init__(self, n):
        root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
        env = dict(os.environ, PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""), OMP_NUM_THREADS="1")
        self.ws = [subprocess.Popen([sys.executable, "-u", "-m", "sketchedit_amd.png_worker"], stdin=subprocess.PIPE,
                                    stdout=subprocess.PIPE, env=env, cwd=root, bufsize=0) for _ in range(n)]
        self.load = [0] * n
        self.futs = {}
        self.next_id = 0
        self.lock = threading.Lock()
        self.bufs = [bytearray() for _ in range(n)]
        self.sel = selectors.DefaultSelector()
        for k, w in enumerate(self.ws):
            self.sel.register(w.stdout.fileno(), selectors.EVENT_READ, k)
        ready, t_end = set(), time.perf_counter() + 120.0
        while len(ready) < n:                               # "ready <pid>": imports done
            if time.perf_counter() > t_end:
                raise RuntimeError("PNG worker processes failed to start")
            for key, _ in self.sel.select(timeout=1.0):
                for line in self._lines(key.data):
                    if line is None:
                        raise RuntimeError("PNG worker process %d exited at start-up" % key.data)
                    if line.startswith(b"ready"):
                        ready.add(key.data)
        self.alive = True
        self.reader = threading.Thread(target=self._read, name="png-procs", daemon=True)
        self.reader.start()

    def _lines(self, k):
        """complete lines now available from worker k (raw read of its pipe); [None] at end of file"""
        data = os.read(self.ws[k].stdout.fileno(), 65536)
        if not data:
            return [None]
        buf = self.bufs[k]
        buf += data
        out = []
        while True:
            i = buf.find(b"\n")
            if i < 0:
                return out
            out.append(bytes(buf[:i]))
            del buf[:i + 1]

    def _read(self):
        while self.alive:
            for key, _ in self.sel.select(timeout=0.2):
                k = key.data
                for line in self._lines(k):
                    if line is None:
                        self.sel.unregister(key.fileobj)
                        with self.lock:
                            dead = [(i, f) for i, (f, kk) in self.futs.items() if kk == k]
                            for i, _ in dead:
                                del self.futs[i]
                        for _, f in dead:
                            f.set_exception(RuntimeError("PNG worker process %d died" % k))
                        break
                    kind, jid, rest = line.decode().split(" ", 2)
                    with self.lock:
                        f, kk = self.futs.pop(int(jid))
                        self.load[kk] -= 1
                    if kind == "done":
                        f.set_result(float(rest))
                    else:
                        f.set_exception(RuntimeError("PNG worker: " + json.loads(rest)))

    def submit(self, job):
        f = Future()
        with self.lock:
            k = min(range(len(self.ws)), key=self.load.__getitem__)
            jid = self.next_id
            self.next_id += 1
            self.futs[jid] = (f, k)
            self.load[k] += 1
        self.ws[k].stdin.write((json.dumps(dict(job, id=jid)) + "\n").encode())
        return f

    def close(self):
        for w in self.ws:
            try:
                w.stdin.close()
            except OSError:
                pass
        for w in self.ws:
            try:
                w.wait(timeout=60)
            except subprocess.TimeoutExpired:
                w.kill()
        self.alive = False
        self.reader.join(timeout=5)
        self.sel.close()


class _Batch:
    __slots__ = ("paths", "n", "slot_in", "slot_out", "dev", "ev0", "ev_h2d", "ev_fwd0", "ev_fwd", "ev_d2h", "futs", "shape", "rings", "dfuts")


_ring_files = set()      # shared rings of this process still on /dev/shm (removed by close(); by the exit hook if a caller never got there)


def _unlink_leftover_rings():
    for path in list(_ring_files):
        try:
            os.unlink(path)
        except OSError:
            pass


atexit.register(_unlink_leftover_rings)


class _PinnedRing:
    """`n` slots of one uint8 array shape in page-locked host memory, allocated ONCE (a pinned allocation maps memory into the
    GPU's address space -- not something to do per batch).  With `shared` the ring is a file on /dev/shm registered with the
    HIP runtime (hipHostRegister): the device-to-host copy lands where the encoder PROCESSES can map it."""

    def __init__(self, n, shape, shared=False):
        self.n, self.shape, self.path = n, tuple(shape), None
        self.free = deque(range(n))
        nbytes = int(np.prod((n,) + self.shape))
        if shared:
            base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else tempfile.gettempdir()
            try:                                     # a container's /dev/shm may be 64 MB: fall back to the temp directory
                st = os.statvfs(base)
                if st.f_bavail * st.f_frsize < nbytes + (64 << 20):
                    base = tempfile.gettempdir()
            except OSError:
                pass
            fd, self.path = tempfile.mkstemp(prefix="se_ring_%d_" % os.getpid(), dir=base)
            _ring_files.add(self.path)
            try:
                os.posix_fallocate(fd, 0, nbytes)    # reserve the pages now: ENOSPC here instead of a SIGBUS at the first touch
            finally:
                os.close(fd)
            self.mm = np.memmap(self.path, dtype=np.uint8, mode="w+", shape=(n,) + self.shape)
            self.t = torch.from_numpy(self.mm)
            rc = torch.cuda.cudart().cudaHostRegister(self.t.data_ptr(), nbytes, 0)
            self.registered = int(rc) == 0
            if not self.registered:
                raise RuntimeError("hipHostRegister failed (%r)" % (rc,))
        else:
            self.t = torch.empty((n,) + self.shape, dtype=torch.uint8, pin_memory=True)
            self.registered = False

    def close(self):
        if self.registered:
            torch.cuda.cudart().cudaHostUnregister(self.t.data_ptr())
            self.registered = False
        if self.path:
            self.t = self.mm = None
            try:
                os.unlink(self.path)
            except OSError:
                pass
            _ring_files.discard(self.path)
            self.path = None


class InferencePipeline:
    """model: EditLine2Model (eval); out_dir / mask_dir as test.py's --output_dir / --output_mask_dir.
    encode_threads: PNG encoder threads (the zlib work releases the GIL, the bookkeeping around it does not: about 2 000
    images/s is the most one process encodes, measured on a 2 x 64-core host); encode_procs > 0: that many encoder PROCESSES
    fed through a shared page-locked ring instead."""

    def __init__(self, model, out_dir, mask_dir=None, encode_threads=1, depth=2, low_latency=None, max_pending_batches=6,
                 timing=False, verbose=True, encode=True, encode_procs=0, png_writer="pil", decode_procs=0, decode_ahead=4):
        self.model, self.out_dir, self.mask_dir = model, out_dir, mask_dir
        self.depth, self.low_latency, self.verbose, self.encode = max(1, depth), low_latency, verbose, encode
        # never more intra-op threads than the CPUs this process is granted (affinity AND cgroup quota): on a 256-CPU host with a
        # 16-CPU quota, ONE parallel torch CPU op (128 OpenMP threads) gets the whole cgroup throttled for the rest of the
        # scheduler period, launch thread included -- measured in round 6 as a 2 800 -> 1 800 images/s drop from one staging copy
        from .hostinfo import effective_cpus
        torch.set_num_threads(max(1, min(torch.get_num_threads(), effective_cpus() // 2 or 1)))
        self.dev = torch.device("cuda", model.opt.gpu_ids[0])
        self.s_h2d, self.s_fwd, self.s_d2h = (torch.cuda.Stream(device=self.dev) for _ in range(3))
        self.max_pending = max(1, max_pending_batches)
        self.timing = timing
        self.png_writer = png_writer
        self.procs = None
        self.pool = None
        if encode and encode_procs > 0:
            self.procs = _EncoderProcs(int(encode_procs))
        else:
            self.pool = ThreadPoolExecutor(max(1, int(encode_threads)), thread_name_prefix="png")
        # decode_procs > 0: run_paths() decodes with that many worker processes straight into a shared page-locked input ring
        # (no DataLoader: nothing is pickled, collated or staged) `decode_ahead` batches ahead of the device
        self.dprocs = _EncoderProcs(int(decode_procs)) if decode_procs > 0 else None
        self.decode_ahead = max(1, int(decode_ahead))
        self.rings = {}            # (kind, shape) -> _PinnedRing
        self.stats = {"images": 0, "batches": 0, "decode_wait_s": 0.0, "h2d_ms": 0.0, "forward_ms": 0.0, "d2h_ms": 0.0,
                      "encode_cpu_s": 0.0, "sync_wait_s": 0.0, "encode_backpressure_s": 0.0,
                      # main-thread seconds spent ISSUING each stage (host cost of the launches, not the device time)
                      "issue_stage_s": 0.0, "issue_h2d_s": 0.0, "issue_forward_s": 0.0, "issue_d2h_s": 0.0, "submit_encode_s": 0.0}
        self._lock = threading.Lock()

    def _ring(self, kind, shape, n, shared=False):
        key = (kind, tuple(shape))
        r = self.rings.get(key)
        if r is None:
            r = self.rings[key] = _PinnedRing(n, shape, shared)
        return r

    # ---- device side ---------------------------------------------------------------------------------------------
    def _enqueue(self, data_i, pending):
        """a DataLoader batch (u8 mode): stage it into the page-locked input ring, then hand it to the device"""
        b = _Batch()
        b.paths = list(data_i["path"])
        iu8, su8 = data_i["image_u8"], data_i["mask_u8"]
        b.n = iu8.shape[0]
        b.shape = tuple(iu8.shape)
        nin = self.depth + 2
        rin_i, rin_s = self._ring("in_i", iu8.shape, nin), self._ring("in_s", su8.shape, nin)
        t0 = time.perf_counter()
        b.slot_in = rin_i.free.popleft()
        np.copyto(rin_i.t[b.slot_in].numpy(), iu8.numpy())           # staging copy into page-locked memory: one thread, ~1 ms for 8 MB
        np.copyto(rin_s.t[b.slot_in].numpy(), su8.numpy())
        self.stats["issue_stage_s"] += time.perf_counter() - t0
        return self._to_device(b, rin_i, rin_s, pending)

    def _to_device(self, b, rin_i, rin_s, pending):
        """H2D of input slot b.slot_in (first b.n images), forward, D2H into an output slot: three streams, events only"""
        H, W = b.shape[1:3]
        ll = self.low_latency if self.low_latency is not None else self.model.batch_mode(H, W)
        nout = self.depth + self.max_pending + 2
        shared = self.procs is not None
        full = (rin_i.shape[0],) + tuple(b.shape[1:])                # ring slots hold a FULL batch; a ragged last one uses a prefix
        rout = self._ring("out_rgb", full, nout, shared)
        rout_m = self._ring("out_m8", full[:3], nout, shared) if self.mask_dir is not None else None
        t0 = time.perf_counter()
        while not rout.free:                         # every output slot is with the encoders: wait for the oldest batch
            self._release_oldest(pending)
        b.slot_out = rout.free.popleft()
        self.stats["issue_stage_s"] += time.perf_counter() - t0
        hi, hs = rin_i.t[b.slot_in][: b.n], rin_s.t[b.slot_in][: b.n]
        mk = lambda: torch.cuda.Event(enable_timing=self.timing)      # noqa: E731
        b.ev_h2d, b.ev_fwd0, b.ev_fwd, b.ev_d2h = mk(), mk(), mk(), mk()
        b.ev0 = mk() if self.timing else None
        t1 = time.perf_counter()
        with torch.cuda.stream(self.s_h2d):
            if b.ev0 is not None:
                b.ev0.record(self.s_h2d)
            di, ds = hi.to(self.dev, non_blocking=True), hs.to(self.dev, non_blocking=True)
            b.ev_h2d.record(self.s_h2d)
        t2 = time.perf_counter()
        with torch.cuda.stream(self.s_fwd):
            self.s_fwd.wait_event(b.ev_h2d)
            b.ev_fwd0.record(self.s_fwd)
            rgb, m8 = self.model.inference_u8({"image_u8": di, "mask_u8": ds}, low_latency=ll)
            b.ev_fwd.record(self.s_fwd)
        t3 = time.perf_counter()
        with torch.cuda.stream(self.s_d2h):
            self.s_d2h.wait_event(b.ev_fwd)
            rout.t[b.slot_out][: b.n].copy_(rgb, non_blocking=True)
            if rout_m is not None:
                rout_m.t[b.slot_out][: b.n].copy_(m8, non_blocking=True)
            b.ev_d2h.record(self.s_d2h)
        b.dev = (di, ds, rgb, m8)                    # device tensors stay alive until the batch has retired
        b.rings = (rin_i, rout, rout_m)
        t4 = time.perf_counter()
        st = self.stats
        st["issue_h2d_s"] += t2 - t1; st["issue_forward_s"] += t3 - t2; st["issue_d2h_s"] += t4 - t3
        return b

    # ---- host side -----------------------------------------------------------------------------------------------
    def _encode_job(self, arr, marr, path):
        t0 = time.perf_counter()
        save_png(arr, os.path.join(self.out_dir, path), self.png_writer)
        if marr is not None:
            save_png(marr, os.path.join(self.mask_dir, path), self.png_writer)
        with self._lock:
            self.stats["encode_cpu_s"] += time.perf_counter() - t0

    def _release_oldest(self, pending):
        """wait for the encoders of the oldest retired batch and give its output slot back"""
        b = pending.popleft()
        for f in b.futs:
            r = f.result()                                             # re-raises an encoder's exception
            if self.procs is not None:
                self.stats["encode_cpu_s"] += r
        b.rings[1].free.append(b.slot_out)

    def _retire(self, b, pending):
        t0 = time.perf_counter()
        b.ev_d2h.synchronize()
        self.stats["sync_wait_s"] += time.perf_counter() - t0
        if self.timing:
            self.stats["h2d_ms"] += b.ev0.elapsed_time(b.ev_h2d)
            self.stats["forward_ms"] += b.ev_fwd0.elapsed_time(b.ev_fwd)
            self.stats["d2h_ms"] += b.ev_fwd.elapsed_time(b.ev_d2h)
        b.rings[0].free.append(b.slot_in)
        b.dev = None
        rout, rout_m = b.rings[1], b.rings[2]
        b.futs = []
        t0 = time.perf_counter()
        if self.verbose:
            for path in b.paths:
                print("process image... %s" % path)
        if self.encode and self.procs is not None:
            chunk = max(1, -(-b.n // max(1, min(b.n, 8))))            # a few images per job: the pipe carries names only
            for first in range(0, b.n, chunk):
                b.futs.append(self.procs.submit(dict(
                    rgb_ring=rout.path, rgb_shape=(rout.n,) + rout.shape, mask_ring=rout_m.path if rout_m else None,
                    mask_shape=((rout_m.n,) + rout_m.shape) if rout_m else None, slot=b.slot_out, first=first,
                    paths=b.paths[first:first + chunk], out_dir=self.out_dir, mask_dir=self.mask_dir, writer=self.png_writer)))
        elif self.encode:
            rgb = rout.t[b.slot_out][: b.n].numpy()
            m8 = rout_m.t[b.slot_out][: b.n].numpy() if rout_m is not None else None
            for i, path in enumerate(b.paths):
                b.futs.append(self.pool.submit(self._encode_job, rgb[i], None if m8 is None else m8[i], path))
        self.stats["submit_encode_s"] += time.perf_counter() - t0
        pending.append(b)
        t0 = time.perf_counter()
        while len(pending) > self.max_pending:       # back-pressure: the encoders are the slow stage
            self._release_oldest(pending)
        self.stats["encode_backpressure_s"] += time.perf_counter() - t0
        self.stats["images"] += b.n
        self.stats["batches"] += 1

    def run(self, dataloader, how_many=float("inf"), batch_size=1):
        """-> stats.  Stops like test.py:21-22 (`if i * opt.batchSize >= opt.how_many: break`)."""
        inflight, pending = deque(), deque()
        t_start = time.perf_counter()
        it = iter(dataloader)
        i = 0
        while True:
            t0 = time.perf_counter()
            try:
                data_i = next(it)
            except StopIteration:
                break
            self.stats["decode_wait_s"] += time.perf_counter() - t0
            if i * batch_size >= how_many:
                break
            i += 1
            if len(inflight) > self.depth:
                self._retire(inflight.popleft(), pending)
            inflight.append(self._enqueue(data_i, pending))
        while inflight:
            self._retire(inflight.popleft(), pending)
        t0 = time.perf_counter()
        while pending:
            self._release_oldest(pending)
        self.stats["encode_drain_s"] = time.perf_counter() - t0
        self.stats["wall_s"] = time.perf_counter() - t_start
        return self.stats

    def run_paths(self, dataset, how_many=float("inf"), batch_size=1):
        """The same loop fed by this pipeline's own decode processes instead of a DataLoader: `dataset` supplies the path lists
        (data/testimage_dataset.py: image_paths, mask_paths, output_paths; serial order); the workers decode straight into a
        shared page-locked ring (hipHostRegister'ed /dev/shm file), `decode_ahead` batches ahead of the device.  Stops like
        test.py:21-22."""
        from PIL import Image
        assert self.dprocs is not None, "run_paths needs decode_procs > 0"
        n = len(dataset.image_paths)
        starts = [k for i, k in enumerate(range(0, n, batch_size)) if i * batch_size < how_many]
        decoding, inflight, pending = deque(), deque(), deque()
        t_start = time.perf_counter()
        nd = len(self.dprocs.ws)

        def submit_decode(k):
            b = _Batch()
            idx = list(range(k, min(k + batch_size, n)))
            b.paths = [dataset.output_paths[i] for i in idx]
            b.n = len(idx)
            with Image.open(dataset.image_paths[k]) as im:      # header only: the batch's image size
                W, H = im.size
            b.shape = (b.n, H, W, 3)
            nin = self.depth + self.decode_ahead + 2
            rin_i = self._ring("in_i", (batch_size, H, W, 3), nin, True)
            rin_s = self._ring("in_s", (batch_size, H, W), nin, True)
            b.slot_in = rin_i.free.popleft()
            b.rings = (rin_i, rin_s)
            chunk = max(1, -(-b.n // max(1, min(b.n, 2 * nd))))
            b.dfuts = [self.dprocs.submit(dict(kind="decode", img_ring=rin_i.path, img_shape=(rin_i.n,) + rin_i.shape, sk_ring=rin_s.path,
                                               sk_shape=(rin_s.n,) + rin_s.shape, slot=b.slot_in, first=f,
                                               image_paths=[dataset.image_paths[i] for i in idx[f:f + chunk]],
                                               mask_paths=[dataset.mask_paths[i] for i in idx[f:f + chunk]]))
                       for f in range(0, b.n, chunk)]
            return b

        nxt = 0
        while nxt < len(starts) or decoding:
            while nxt < len(starts) and len(decoding) < self.decode_ahead:
                decoding.append(submit_decode(starts[nxt]))
                nxt += 1
            b = decoding.popleft()
            t0 = time.perf_counter()
            for f in b.dfuts:
                self.stats["decode_cpu_s"] = self.stats.get("decode_cpu_s", 0.0) + f.result()      # re-raises a decoder's exception
            self.stats["decode_wait_s"] += time.perf_counter() - t0
            if len(inflight) > self.depth:
                self._retire(inflight.popleft(), pending)
            rin_i, rin_s = b.rings
            inflight.append(self._to_device(b, rin_i, rin_s, pending))
        while inflight:
            self._retire(inflight.popleft(), pending)
        t0 = time.perf_counter()
        while pending:
            self._release_oldest(pending)
        self.stats["encode_drain_s"] = time.perf_counter() - t0
        self.stats["wall_s"] = time.perf_counter() - t_start
        return self.stats

    def __enter__(self):
        return self

    def __exit__(self, *exc):
        self.close()
        return False

    def close(self):
        if self.dprocs is not None:
            self.dprocs.close()
            self.dprocs = None
        if self.pool is not None:
            self.pool.shutdown(wait=True)
        if self.procs is not None:
            self.procs.close()
            self.procs = None
        torch.cuda.synchronize(self.dev)
        for r in self.rings.values():
            r.close()
        self.rings = {}


def _dec_loop(image_paths, mask_paths, idx0, stop_t, q):
    from PIL import Image
    n, i = 0, idx0
    while time.perf_counter() < stop_t:
        im = Image.open(image_paths[i % len(image_paths)]).convert("RGB")
        w, h = im.size
        np.array(im, dtype=np.uint8)
        np.array(Image.open(mask_paths[i % len(mask_paths)]).convert("L").resize((w, h)), dtype=np.uint8)
        n += 1
        i += 1
    q.put(n)


def _enc_loop(arr, writer, out_dir, k, stop_t, q):
    n = 0
    path = os.path.join(out_dir, "cap_%d.png" % k)
    while time.perf_counter() < stop_t:
        save_png(arr, path, writer)
        n += 1
    q.put(n)


def host_codec_capability(image_paths, mask_paths, decode_workers, encode_workers, seconds=4.0, writer="pil", out_dir=None):
    """What the host cores can decode + encode IN PARALLEL, without any GPU work (the yardstick VERDICT r5 asks the end-to-end
    rate to be held against): `decode_workers` processes decode (image, sketch) pairs as the dataset does while
    `encode_workers` processes PNG-encode an image of that size with `writer` and write it to `out_dir` (tmpfs); each leg runs
    for `seconds`.  Call it before the process touches the GPU (it forks).  -> dict(decode_ips, encode_ips, both_ips, ...)."""
    import multiprocessing as mp
    from PIL import Image
    arr = np.array(Image.open(image_paths[0]).convert("RGB"), dtype=np.uint8)
    out_dir = out_dir or os.path.dirname(image_paths[0])
    out = {}
    ctx = mp.get_context("fork")
    for mode in ("decode", "encode", "both"):
        qd, qe = ctx.Queue(), ctx.Queue()
        t0 = time.perf_counter()
        stop_t = t0 + seconds
        dec = [ctx.Process(target=_dec_loop, args=(image_paths, mask_paths, w * 9973, stop_t, qd)) for w in range(decode_workers)] if mode != "encode" else []
        enc = [ctx.Process(target=_enc_loop, args=(arr, writer, out_dir, k, stop_t, qe)) for k in range(encode_workers)] if mode != "decode" else []
        for p in dec + enc:
            p.start()
        nd, ne = sum(qd.get() for _ in dec), sum(qe.get() for _ in enc)
        for p in dec + enc:
            p.join()
        dt = time.perf_counter() - t0
        if mode == "decode":
            out["decode_ips"] = nd / dt
        elif mode == "encode":
            out["encode_ips"] = ne / dt
        else:
            out["both_decode_ips"], out["both_encode_ips"] = nd / dt, ne / dt
            out["both_ips"] = min(nd, ne) / dt          # a pipeline moves at its slower stage
    return out


def encoder_rate_on(array, out_dir, workers, writer="pil", seconds=3.0):
    """images/s at which `workers` encoder PROCESSES (the pipeline's own: python -m sketchedit_amd.png_worker) encode + write
    `array` -- an image the network really produced -- for `seconds`.  Safe after the GPU is in use (nothing is forked)."""
    ring_path = os.path.join(out_dir, "_rate_ring")
    shape = (1, 4) + tuple(array.shape)
    mm = np.memmap(ring_path, dtype=np.uint8, mode="w+", shape=shape)
    mm[0, :] = array
    mm.flush()
    procs = _EncoderProcs(workers)
    done, t0 = 0, time.perf_counter()
    try:
        job = dict(rgb_ring=ring_path, rgb_shape=shape, mask_ring=None, mask_shape=None, slot=0, first=0, out_dir=out_dir, mask_dir=None, writer=writer)
        futs = deque()
        k = 0
        while time.perf_counter() - t0 < seconds:
            while len(futs) < 3 * workers:
                futs.append(procs.submit(dict(job, paths=["_rate_%d_%d.png" % (k % (3 * workers), j) for j in range(4)])))
                k += 1
            futs.popleft().result()
            done += 4
        for f in futs:
            f.result()
            done += 4
        dt = time.perf_counter() - t0
    finally:
        procs.close()
        del mm
        for f in os.listdir(out_dir):
            if f.startswith("_rate_"):
                try:
                    os.unlink(os.path.join(out_dir, f))
                except OSError:
                    pass
    return done / dt
