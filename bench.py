#!/usr/bin/env python3
"""Benchmark of the SketchEdit inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--batch 32]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

One "step" = one pass of EditLine2Model.forward(mode='inference') (netM -> threshold -> netG ->
composite) over one synthetic batch per GPU: BASELINE.json config 2, 256x256, batch 32, fp32, inputs
resident in HBM.  With N > 1 every rank runs its own batch shard (weak scaling, B per GPU fixed) and the
(B,3,H,W) composites + (B,1,H,W) masks are all-gathered over RCCL inside the timed region (side stream, under the
next step's forward; every step's gather is complete before the closing fence).
Rank 0 prints ONE JSON line (metric images/sec = N*B*K / max-over-ranks time).

Extra objects in the line:
  roofline     -- dominant kernel (the N=192 gated-conv gather-GEMM): algorithmic FLOPs per launch /
                  average launch duration measured with HIP events on the launch stream (in-library
                  profiler, separate un-timed pass), against the 157.3 TFLOP/s fp32 MFMA peak.
  cpu_baseline -- the oracle (CPU restatement of the reference, oracle/sketchedit_oracle.py) timed on the
                  host cores of this box on a bounded sample; reported, not the target.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sketchedit_amd import synth  # noqa: E402
from sketchedit_amd._lib import Engine, FLAG_JOINT_TRAIN_INP, FLAG_POOL_MAX, FLAG_USE_CAM  # noqa: E402

FP32_MFMA_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FLAGS = FLAG_USE_CAM | FLAG_POOL_MAX | FLAG_JOINT_TRAIN_INP   # test_celeb.sh: --use_cam --pool_type max --joint_train_inp

# reference-defined work per image, live in mode='inference'
LIVE_GFLOP_PER_IMAGE = {256: 90.80, 512: 437.27}


def cpu_baseline(size, budget_s=20.0):
    """Time the oracle on the host cores: 256x256 batches of 2 until ~budget_s of CPU work is spent."""
    from oracle import sketchedit_oracle as O
    # oneDNN collapses when oversubscribed on the 2x64-core GPU hosts (measured with tools/cpu_probe.py:
    # 32 threads 10.0 img/s, 64 threads 4.1, 128 threads 1.4, 256 threads 0.03), so cap at 32.
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    WM, WG = synth.make_state_dict("M", 0), synth.make_state_dict("G", 0)
    WM = {k: torch.from_numpy(v) for k, v in WM.items()}
    WG = {k: torch.from_numpy(v) for k, v in WG.items()}
    B = 2
    img, sk = synth.make_inputs(B, size, size, seed=1234)
    img, sk = torch.from_numpy(img), torch.from_numpy(sk)
    O.inference(WM, WG, img, sk)                      # warm-up (oneDNN primitive creation)
    times = []
    t_start = time.perf_counter()
    while len(times) < 3 or (time.perf_counter() - t_start < budget_s and len(times) < 20):
        t0 = time.perf_counter()
        O.inference(WM, WG, img, sk)
        times.append(time.perf_counter() - t0)
    med = float(np.median(times))
    return {"value": B / med, "unit": "images/sec", "cores": cores, "kind": "port",
            "sample": "oracle (torch CPU restatement of the reference) %dx%d batch %d, median of %d runs after 1 warm-up"
                      % (size, size, B, len(times))}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--layers", action="store_true", help="add the per-layer timing table to the JSON line")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=-1, help="force this HIP device for every rank (test aid)")
    ap.add_argument("--check-gather", action="store_true", help="rank 0 verifies the gathered outputs (test aid)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-gather on the compute stream instead of a side stream")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node %d for --gpus %d" % (args.gpus, args.gpus))
    dist = None
    # --device / --backend exist only to exercise the N > 1 code path on a 1-GPU box (both ranks on cuda:0 over
    # gloo); the driver's runs use the defaults: one rank per GPU (LOCAL_RANK) over nccl = RCCL.
    dev_index = local_rank if args.device < 0 else args.device
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(dev_index)
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    else:
        torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    B, S = args.batch, args.size
    eng = Engine(dev_index)
    eng.load_state_dict("M", synth.make_state_dict("M", 0))
    eng.load_state_dict("G", synth.make_state_dict("G", 0))
    img_h, sk_h = synth.make_inputs(B, S, S, seed=1234, first_index=rank * B)   # shard = rows of the global batch
    img = torch.from_numpy(img_h).to(dev)
    sk = torch.from_numpy(sk_h).to(dev)
    # Outputs are double-buffered: for N > 1 the all-gather of step k runs on a side stream under the forward of step
    # k+1 (the forward has no exchange inside; xGMI traffic and MFMA work do not compete).
    def new_out():
        return {"composed": torch.empty((B, 3, S, S), dtype=torch.float32, device=dev),
                "mask": torch.empty((B, 1, S, S), dtype=torch.float32, device=dev)}
    overlap = world > 1 and not args.no_overlap
    outs = [new_out(), new_out()] if overlap else [new_out()]
    out = outs[0]
    gathered_sets = []
    if world > 1:
        for _ in outs:
            gathered_sets.append({"composed": torch.empty((world * B, 3, S, S), dtype=torch.float32, device=dev),
                                  "mask": torch.empty((world * B, 1, S, S), dtype=torch.float32, device=dev)})
    gathered = gathered_sets[0] if gathered_sets else {}
    comm_stream = torch.cuda.Stream(device=dev) if overlap else None
    gather_done = [None, None]
    state = {"i": 0, "last": 0}

    def gather(o, g):
        # the only exchange of the path: all-gather of the outputs
        for k in ("composed", "mask"):
            if args.backend == "nccl":
                dist.all_gather_into_tensor(g[k], o[k])
            else:              # gloo cannot all-gather device tensors: stage through the host (test aid only)
                parts = [torch.empty(o[k].shape, dtype=torch.float32) for _ in range(world)]
                dist.all_gather(parts, o[k].cpu())
                g[k].copy_(torch.cat(parts, 0))

    def step():
        i = state["i"]
        state["i"] = i + 1
        if not overlap:
            eng.inference(img, sk, FLAGS, out=outs[0])
            if world > 1:
                gather(outs[0], gathered_sets[0])
            return
        slot = i & 1
        main = torch.cuda.current_stream(dev)
        if gather_done[slot] is not None:          # the gather that last read this output buffer (step i-2)
            main.wait_event(gather_done[slot])
        eng.inference(img, sk, FLAGS, out=outs[slot])
        ready = torch.cuda.Event()
        ready.record(main)
        with torch.cuda.stream(comm_stream):
            comm_stream.wait_event(ready)
            gather(outs[slot], gathered_sets[slot])
            done = torch.cuda.Event()
            done.record(comm_stream)
        gather_done[slot] = done
        state["last"] = slot

    def fence():
        torch.cuda.synchronize(dev)
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize(dev)

    for _ in range(args.warmup):
        step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if args.check_gather:
            # every rank finds its own rows in the gathered batch; rank 0 recomputes image 0 of rank 1's shard
            out, gathered = outs[state["last"]], gathered_sets[state["last"]]
            ok = torch.equal(gathered["composed"][rank * B:(rank + 1) * B], out["composed"]) and \
                torch.equal(gathered["mask"][rank * B:(rank + 1) * B], out["mask"])
            if rank == 0:
                i1, s1 = synth.make_inputs(1, S, S, seed=1234, first_index=B)
                r1 = eng.inference(torch.from_numpy(i1).to(dev), torch.from_numpy(s1).to(dev), FLAGS)
                ok = ok and torch.equal(r1["composed"], gathered["composed"][B:B + 1])
            f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            if float(f.item()) != 1.0:
                raise SystemExit("gathered outputs do not match the per-rank outputs")

    # ---- per-kernel timing pass (HIP events on the launch stream), not part of the timed region
    roofline, kernels = None, None
    if rank == 0:
        eng.profile(True)
        nprof = 2
        for _ in range(nprof):
            eng.inference(img, sk, FLAGS, out=out)
        full_rep = eng.profile_report()
        rep = full_rep["kernels"]
        eng.profile(False)
        kernels = {r["kernel"]: {"launches_per_step": r["launches"] // nprof,
                                 "ms_per_step": r["total_ms"] / nprof,
                                 "avg_us": 1e3 * r["total_ms"] / r["launches"],
                                 "tflops": (r["flops"] / (r["total_ms"] * 1e-3) / 1e12) if r["flops"] > 0 else None}
                   for r in rep}
        dom = max(rep, key=lambda r: r["total_ms"])
        achieved = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        conv_ms = sum(r["total_ms"] for r in rep if r["kernel"].startswith(("gconv", "wino"))) / nprof
        conv_fl = sum(r["flops"] for r in rep if r["kernel"].startswith(("gconv", "wino"))) / nprof
        # HBM bytes per launch of the dominant kernel come from the separate rocprofv3 --pmc passes
        # (FETCH_SIZE x2 + WRITE_SIZE, tools/pmc_summary.py -> profiles/pmc_traffic.json); null if not collected
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_traffic.json")) as f:
                traffic = json.load(f)["kernels"].get(dom["kernel"])
        except (OSError, ValueError, KeyError):
            pass
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(achieved, 3),
                    "peak": FP32_MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(achieved / FP32_MFMA_PEAK_TFLOPS, 4),
                    "traffic": traffic,
                    # the Winograd kernel executes 16 positions instead of 36 tap-products per 4 outputs: its MFMA
                    # pipe does 4/9 of the algorithmic (reference-defined) FLOPs, so frac can exceed 1
                    "executed_over_algorithmic": round(16.0 / 36.0, 4) if dom["kernel"].startswith("wino") else 1.0,
                    "launches_per_step": dom["launches"] // nprof,
                    "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                    "flops_per_launch": dom["flops"] / dom["launches"],
                    "gated_conv_stack_tflops": round(conv_fl / (conv_ms * 1e-3) / 1e12, 3),
                    "gated_conv_stack_frac": round(conv_fl / (conv_ms * 1e-3) / 1e12 / FP32_MFMA_PEAK_TFLOPS, 4)}

    # ---- parity of this very run against the oracle on image 0 (CPU, rank 0)
    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import sketchedit_oracle as O
        WM = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
        WG = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}
        ref = O.inference(WM, WG, img_h[:1], sk_h[:1])
        r1 = eng.inference(img[:1].contiguous(), sk[:1].contiguous(), FLAGS, visualize=True)
        flips = int((r1["hard"].cpu() != ref["hard_mask"]).sum())
        parity = {"max_abs_composed": float((r1["composed"].cpu() - ref["composed"]).abs().max()),
                  "max_abs_mask": float((r1["mask"].cpu() - ref["mask"]).abs().max()),
                  "hard_mask_flips": flips, "image": 0, "tolerance": 1e-3}

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline(S)

    if rank == 0:
        images = world * B * args.steps
        line = {
            "metric": "images/sec", "value": images / elapsed, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
            "config": {"workload": "SketchEdit inference forward (netM+netG, use_cam, pool max) %dx%d batch %d per GPU, "
                                   "procedural weights" % (S, S, B),
                       "global_batch": world * B, "size": S, "per_gpu_batch": B,
                       "collective": ("all_gather(composed, mask)" + (" on a side stream, under the next step's forward" if overlap else "")) if world > 1 else None},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "kernels": kernels,
            "layers": ({r["layer"]: {"ms": round(r["total_ms"] / nprof, 4), "n": r["launches"] // nprof,
                                     "tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 1)}
                        for r in full_rep["layers"]} if args.layers else None),
            "forward_tflops_live": (LIVE_GFLOP_PER_IMAGE.get(S, 0) * images / elapsed / 1e3) or None,
        }
        print(json.dumps(line))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
