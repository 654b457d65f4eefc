"""Serving helpers for the inference path: the per-request pre/post-processing of the reference's demo
(/root/reference/demo.py:39-73, `process_image`) and a small dynamic batcher for concurrent callers
(demo.py:120 runs Flask with threaded=True, i.e. concurrent forwards on one model object),
with one worker per GPU when several models are given.

No web framework here -- the Flask UI is out of scope; these are the pieces of it that touch the hot path.
"""
import threading
import time

import numpy as np


def _to_tensors(img, mask):
    """demo.py:40-56: RGB, sizes floored to multiples of 8, image to [-1, 1], mask > 0."""
    import torch
    img = img.convert("RGB")
    w_raw, h_raw = img.size
    h_t, w_t = h_raw // 8 * 8, w_raw // 8 * 8
    if h_t < 16 or w_t < 16:
        raise ValueError("image too small: %dx%d" % (w_raw, h_raw))
    x = np.array(img.resize((w_t, h_t))).transpose((2, 0, 1))
    m = np.array(mask.resize((w_t, h_t)))
    if m.ndim == 3:
        m = m[..., 0]
    m = (torch.from_numpy(m.astype(np.float32)) > 0).float()
    x = (torch.from_numpy(x.astype(np.float32)) / 255 - 0.5) / 0.5
    return x[None], m[None, None], (w_raw, h_raw)


def _to_image(generated, size_raw):
    """demo.py:62-70: clamp, (x+1)/2*255, uint8, HWC, resize back to the request's size."""
    import torch
    from PIL import Image
    g = torch.clamp(generated, -1, 1)
    g = ((g + 1) / 2 * 255).cpu().numpy().astype(np.uint8)
    return Image.fromarray(g[0].transpose((1, 2, 0))).resize(size_raw)


def _accepts_low_latency(model):
    import inspect
    try:
        return "low_latency" in inspect.signature(getattr(model, "forward", model)).parameters
    except (TypeError, ValueError):
        return False


def process_image(model, img, mask, low_latency=None):
    """One request, as demo.py:39-73 handles it: PIL image + PIL sketch/mask in, PIL result out.  `low_latency` pins the
    library's execution mode (EditLine2Model.forward); None = by call size."""
    import torch
    x, m, size_raw = _to_tensors(img, mask)
    with torch.no_grad():
        if low_latency is None:
            generated, _ = model({"image": x, "mask": m}, mode="inference")
        else:
            generated, _ = model({"image": x, "mask": m}, mode="inference", low_latency=low_latency)
    return _to_image(generated, size_raw)


def create_models_for_gpus(opt, gpu_ids=None):
    """One EditLine2Model per GPU of this node (weights replicated, one se_ctx / stream / workspace each): the worker set
    of a multi-GPU BatchingServer.  `gpu_ids` defaults to every visible device."""
    import copy
    import torch
    from . import models
    ids = list(range(torch.cuda.device_count())) if gpu_ids is None else list(gpu_ids)
    out = []
    for i in ids:
        o = copy.copy(opt)
        o.gpu_ids = [i]
        with torch.cuda.device(i):
            out.append(models.create_model(o).eval())
    return out


class BatchingServer:
    """Concurrent `submit(img, mask)` calls are grouped by working size and run as one forward per group
    (up to `max_batch` requests, waiting at most `max_wait_s` for company).  The forward treats the images of a
    batch independently and an image's result is bit-identical across batch positions WITHIN one
    execution mode of the library (include/sketchedit_hip.h: default vs SE_FLAG_LOW_LATENCY run different kernels that
    agree to fp32 rounding -- enough to flip a soft-mask value sitting on the 0.5 threshold).  `mode_policy`:
      "pinned" (default): the mode is a function of the request's working size and `max_batch` only -- low-latency iff
          even a full group of this size (`max_batch` images) is a small call -- so a request's result does NOT depend
          on what it was batched with;
      "by_size": the Engine picks the mode from each group's actual size (lowest latency for a lone request; results
          may differ at fp32 rounding level between a lone and a batched run of the same request).

    `models` = one model per GPU (create_models_for_gpus): every model gets its own worker thread, all workers pull
    groups from the one shared queue -- dynamic batching across the GPUs of the node; an idle GPU
    takes the next group, so the load balances itself.  `model` = the single-GPU form."""

    def __init__(self, model=None, max_batch=32, max_wait_s=0.005, models=None, mode_policy="pinned"):
        if mode_policy not in ("pinned", "by_size"):
            raise ValueError(mode_policy)
        self.mode_policy = mode_policy
        self.models = list(models) if models is not None else [model]
        if not self.models or any(m is None for m in self.models):
            raise ValueError("BatchingServer needs a model (or a list of models, one per GPU)")
        self.model = self.models[0]
        self._has_knob = [_accepts_low_latency(m) for m in self.models]
        self.max_batch, self.max_wait_s = max_batch, max_wait_s
        self._lock = threading.Condition()
        self._queue = []          # (x, m, size_raw, slot, arrival time)
        self._stop = False
        self.batches = []         # sizes of the batches that were run (observability / tests)
        self.batches_by_model = [0] * len(self.models)
        self._collecting = set()  # working sizes some worker is waiting out a deadline for (one collector per size)
        self._workers = [threading.Thread(target=self._run, args=(k,), daemon=True) for k in range(len(self.models))]
        for t in self._workers:
            t.start()

    def submit(self, img, mask):
        x, m, size_raw = _to_tensors(img, mask)
        slot = {"done": threading.Event(), "out": None, "err": None}
        with self._lock:
            if self._stop:
                raise RuntimeError("server is closed")
            self._queue.append((x, m, size_raw, slot, time.monotonic()))
            self._lock.notify_all()
        slot["done"].wait()
        if slot["err"] is not None:
            raise slot["err"]
        return slot["out"]

    def close(self):
        with self._lock:
            self._stop = True
            self._lock.notify_all()
        for t in self._workers:
            t.join()

    def _take_group(self):
        """A worker becomes the collector of the OLDEST queued request whose working size no other worker is collecting, and
        waits for company of that size until the group is full or `max_wait_s` after that oldest request ARRIVED (not after
        collection started).  Several workers collect at once, one per distinct size, so an idle GPU never sits out a
        deadline for size S1 while requests of size S2 are queued; with k sizes queued the waits overlap
        instead of stacking."""
        with self._lock:
            while True:
                head = next((q for q in self._queue if tuple(q[0].shape) not in self._collecting), None)
                if head is None:
                    if self._stop and not self._queue:
                        return None                   # stopped and drained
                    self._lock.wait(0.05 if self._stop else None)
                    continue
                shape = tuple(head[0].shape)
                self._collecting.add(shape)
                try:
                    deadline = head[4] + self.max_wait_s
                    while True:
                        n = sum(1 for q in self._queue if tuple(q[0].shape) == shape)
                        left = deadline - time.monotonic()
                        if n >= self.max_batch or left <= 0 or self._stop:
                            break
                        self._lock.wait(left)
                    group = [q for q in self._queue if tuple(q[0].shape) == shape][:self.max_batch]
                    taken = {id(q) for q in group}    # (list.remove would compare the tensors inside the tuples)
                    self._queue = [q for q in self._queue if id(q) not in taken]
                finally:
                    self._collecting.discard(shape)
                    self._lock.notify_all()           # requests of this size that did not fit may be collected now
                return group

    def _mode(self, shape):
        """Execution mode of a group of requests of working size `shape` (1,3,H,W) -- see the class docstring."""
        if self.mode_policy == "by_size":
            return None
        from ._lib import Engine
        return Engine.is_low_latency(self.max_batch, shape[2], shape[3])

    def _forward(self, k, x, m):
        model = self.models[k]
        if self._has_knob[k]:
            return model({"image": x, "mask": m}, mode="inference", low_latency=self._mode(tuple(x.shape)))
        return model({"image": x, "mask": m}, mode="inference")   # a model object without the execution-mode knob

    def _run(self, k):
        import torch
        while True:
            group = self._take_group()
            if group is None:
                return
            if not group:                 # another worker took the requests this one was waiting with
                continue
            try:
                x = torch.cat([q[0] for q in group], 0)
                m = torch.cat([q[1] for q in group], 0)
                with torch.no_grad():
                    generated, _ = self._forward(k, x, m)
                with self._lock:
                    self.batches.append(len(group))
                    self.batches_by_model[k] += 1
                for i, q in enumerate(group):
                    q[3]["out"] = _to_image(generated[i:i + 1], q[2])
            except Exception as e:      # deliver the failure to every waiting caller
                for q in group:
                    q[3]["err"] = e
            for q in group:
                q[3]["done"].set()
