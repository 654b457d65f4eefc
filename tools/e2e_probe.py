#!/usr/bin/env python3
"""Where does the end-to-end loop lose time?  Variants of the pipeline on one GPU (developer probe, not a test)."""
import os, sys, time, tempfile, shutil
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from argparse import Namespace
from sketchedit_amd import synth, models, data
from sketchedit_amd.pipeline import InferencePipeline

B, S, NB = 32, 256, 100
opt = Namespace(gpu_ids=[0], isTrain=False, model="editline2", netG="deepfillc2", init_type=None, init_variance=0.02, use_cam=True, pool_type="max",
                no_mask_cc=False, no_mask_coarse=False, joint_train_inp=True, isSkip=True, which_epoch="latest", checkpoints_dir="/tmp", name="x",
                batchSize=B, conservative_mask=False)
torch.cuda.set_device(0)
model = models.create_model(opt)
model.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
model.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
model.cuda(); model.eval()
rng = np.random.default_rng(0)
iu8 = torch.from_numpy(rng.integers(0, 256, (B, S, S, 3), dtype=np.uint8)).pin_memory()
su8 = torch.from_numpy(((rng.random((B, S, S)) < 0.005) * 255).astype(np.uint8)).pin_memory()
eng = model.engine()
di, ds = iu8.cuda(), su8.cuda()
for _ in range(3):
    eng.inference_u8io(di, ds, 19, low_latency=False)
torch.cuda.synchronize()
t = time.perf_counter()
for _ in range(NB):
    eng.inference_u8io(di, ds, 19, low_latency=False)
torch.cuda.synchronize()
print("1 forward only (engine): %.0f img/s" % (NB * B / (time.perf_counter() - t)), flush=True)
t = time.perf_counter()
for _ in range(NB):
    model.inference_u8({"image_u8": di, "mask_u8": ds}, low_latency=False)
torch.cuda.synchronize()
print("1b forward only (model wrapper): %.0f img/s" % (NB * B / (time.perf_counter() - t)), flush=True)
t = time.perf_counter()
for _ in range(NB):
    a, b = iu8.cuda(non_blocking=True), su8.cuda(non_blocking=True)
    r, m = eng.inference_u8io(a, b, 19, low_latency=False)
    hr = r.cpu()
torch.cuda.synchronize()
print("1c serial h2d + forward + d2h (one stream, blocking .cpu()): %.0f img/s" % (NB * B / (time.perf_counter() - t)), flush=True)

def fake_loader(n):
    for i in range(n):
        yield {"image_u8": iu8, "mask_u8": su8, "path": ["f%05d.png" % (i * B + j) for j in range(B)]}

out = tempfile.mkdtemp(prefix="se_probe_", dir="/dev/shm")
for depth in (1, 2, 3, 4):
    p = InferencePipeline(model, out, None, encode_threads=1, depth=depth, timing=True, verbose=False, encode=False)
    st = p.run(fake_loader(NB), float("inf"), B); p.close()
    print("2 pipeline depth %d, fake loader, no encode: %.0f img/s  fwd %.0f  h2d %.0f  sync_wait %.2f wall %.2f" % (
        depth, st["images"] / st["wall_s"], st["images"] / (st["forward_ms"] * 1e-3), st["images"] / (st["h2d_ms"] * 1e-3), st["sync_wait_s"], st["wall_s"]), flush=True)
for thr, procs, wr in ((12, 0, "pil"), (0, 12, "pil"), (0, 12, "fast"), (0, 6, "fast")):
    p = InferencePipeline(model, out, None, encode_threads=thr, depth=3, timing=True, verbose=False, encode=True, encode_procs=procs, max_pending_batches=8, png_writer=wr)
    st = p.run(fake_loader(NB), float("inf"), B); p.close()
    print("3 pipeline depth 3, fake loader, %d encoder threads / %d procs, %s writer: %.0f img/s  fwd %.0f  backpressure %.2f submit %.2f issue_fwd %.2f stage %.2f drain %.2f" % (
        thr, procs, wr, st["images"] / st["wall_s"], st["images"] / (st["forward_ms"] * 1e-3), st["encode_backpressure_s"], st["submit_encode_s"], st["issue_forward_s"], st["issue_stage_s"], st["encode_drain_s"]), flush=True)
shutil.rmtree(out, ignore_errors=True)
