#!/usr/bin/env python3
"""Benchmark of the SketchEdit inference hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--size 256] [--batch 32] [--dtype f32|bf16]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Both forms work for N > 1: started WITHOUT a launcher (no WORLD_SIZE in the environment), `python bench.py --gpus N`
starts its own N rank processes (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR=127.0.0.1 / a free MASTER_PORT), relays
rank 0's ONE JSON line and exits with the first failing rank's code.

One "step" = one pass of EditLine2Model.forward(mode='inference') (netM -> threshold -> netG ->
composite) over one synthetic batch per GPU: BASELINE.json config 2, 256x256, batch 32, fp32, inputs
resident in HBM.  With N > 1 every rank runs its own batch shard (weak scaling, B per GPU fixed) and the packed
(B,4,H,W) outputs (composite + soft mask) are all-gathered over RCCL inside the timed region -- ONE collective per
step, on a side stream under the next step's forward; every step's gather is complete before the closing fence.
Rank 0 prints ONE JSON line (metric images/sec = N*B*K / max-over-ranks time).

Extra objects in the line:
  roofline     -- the dominant kernel label: EXECUTED multiply-add FLOPs per launch (what its MFMA pipe really runs:
                  24/72 of the reference-defined FLOPs for the hybrid Winograd F(2,3)xF(4,3) kernel, 16/36 for
                  F(2x2,3x3)) / average launch duration
                  measured with HIP events on the launch stream (in-library profiler, separate un-timed pass),
                  against the 157.3 TFLOP/s fp32 MFMA peak: `frac` is a hardware fraction, never above 1.  The
                  reference-defined rate is kept beside it (`algorithmic_tflops`).  `traffic` = HBM bytes per launch
                  from rocprofv3 PMC passes (FETCH_SIZE x2 + WRITE_SIZE) of a child run of this very script, made
                  while this run is still alive; null when rocprofv3 is not usable (see `traffic_source`).
  secondary    -- BASELINE.json configs 3 (512x512 batch 8 fp32) and 5 (512x512 batch 16 bf16) timed in the same invocation
                  after the headline (12 steps each between device fences), with their executed-FLOP fractions and the
                  parity of their own last timed step; only at the default invocation (--no-secondary skips them).
  cpu_baseline -- the oracle (CPU restatement of the reference, oracle/sketchedit_oracle.py) timed on the
                  host cores of this box at the sizes SURVEY.md 8d names; reported, not the target.
"""
import argparse
import csv
import faulthandler
import glob
import json
import os
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from sketchedit_amd import synth  # noqa: E402
from sketchedit_amd._lib import Engine, FLAG_JOINT_TRAIN_INP, FLAG_POOL_MAX, FLAG_USE_CAM  # noqa: E402

# /opt/skills/guides/MI355X_MICROARCH.md: "Peak FP32 (matrix)" 157.3 TF; "Peak BF16/FP16 MFMA ~2.5 PF dense"
MFMA_PEAK_TFLOPS = {"f32": 157.3, "bf16": 2500.0}
FLAGS = FLAG_USE_CAM | FLAG_POOL_MAX | FLAG_JOINT_TRAIN_INP   # test_celeb.sh: --use_cam --pool_type max --joint_train_inp

# reference-defined work per image, live in mode='inference'
LIVE_GFLOP_PER_IMAGE = {256: 90.80, 512: 437.27}

from sketchedit_amd.kernel_labels import KERNEL_LABELS, label_of  # noqa: E402  (rocprofv3 kernel names -> profiler labels)


def _lib_opt(name):
    from sketchedit_amd import _lib
    return _lib.get_option(name)


def cpu_baseline(budget_s=25.0):
    """Time the oracle on the host cores at the sizes SURVEY.md 8d names: 256x256 batch 1 and 8, 512x512 batch 1."""
    from oracle import sketchedit_oracle as O
    # oneDNN collapses when oversubscribed on the 2x64-core GPU hosts (measured with tools/cpu_probe.py:
    # 32 threads 10.0 img/s, 64 threads 4.1, 128 threads 1.4, 256 threads 0.03), so the thread count is capped at 32.
    from sketchedit_amd.hostinfo import cgroup_cpu_quota, effective_cpus
    host = os.cpu_count() or 1
    # ... and never more threads than the CPUs the process may really use (affinity AND cgroup quota: the GPU boxes grant 16
    # CPUs' worth of time on a 256-CPU host; 32 threads there are throttled, not faster -- found in round 6)
    cores = min(effective_cpus(), 32)
    torch.set_num_threads(cores)
    WM = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
    WG = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}
    samples = []
    t_all = time.perf_counter()
    for size, B in ((256, 8), (256, 1), (512, 1)):
        img, sk = synth.make_inputs(B, size, size, seed=1234)
        img, sk = torch.from_numpy(img), torch.from_numpy(sk)
        for _ in range(2):                                # two warm-ups: oneDNN primitive creation, allocator
            O.inference(WM, WG, img, sk)
        times = []
        t_start = time.perf_counter()
        while len(times) < 2 or (time.perf_counter() - t_start < budget_s / 3 and len(times) < 5):
            t0 = time.perf_counter()
            O.inference(WM, WG, img, sk)
            times.append(time.perf_counter() - t0)
        samples.append({"size": size, "batch": B, "images_per_sec": B / float(np.median(times)), "runs": len(times)})
    all_cores = cpu_all_cores_sample(host) if host > cores else {"threads": host, "images_per_sec": samples[1]["images_per_sec"], "note": "same as samples[1]: the host has no more than 32 CPUs"}
    return {"value": samples[0]["images_per_sec"], "unit": "images/sec", "cores": cores, "host_cores": host,
            "cgroup_cpu_quota": cgroup_cpu_quota(), "thread_cap": 32, "kind": "port", "all_cores": all_cores,
            "sample": "oracle (torch CPU restatement of the reference): value = 256x256 batch 8, median of %d runs after 2 "
                      "warm-ups; `samples` also holds 256x256 batch 1 and 512x512 batch 1 (%.0f s of CPU work in all)"
                      % (samples[0]["runs"], time.perf_counter() - t_all),
            "samples": samples}


def cpu_all_cores_probe():
    """Child mode of cpu_all_cores_sample: the oracle at 256x256 batch 1 on EVERY host CPU, one line per finished forward."""
    from oracle import sketchedit_oracle as O
    host = os.cpu_count() or 1
    torch.set_num_threads(host)
    WM = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
    WG = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}
    img, sk = synth.make_inputs(1, 256, 256, seed=1234)
    img, sk = torch.from_numpy(img), torch.from_numpy(sk)
    for i in range(4):                                    # run 0 is the warm-up
        t0 = time.perf_counter()
        O.inference(WM, WG, img, sk)
        print(json.dumps({"run": i, "seconds": time.perf_counter() - t0, "threads": torch.get_num_threads()}), flush=True)


def cpu_all_cores_sample(host, budget_s=40.0):
    """SURVEY.md 8d asks for the CPU path on ALL host cores with the count printed; `value` keeps the 32-thread figure because
    oneDNN collapses when oversubscribed on the GPU hosts (tools/cpu_probe.py: 256 threads 0.03 img/s).  This leg puts the
    all-cores number on the line too, bounded: a child process runs 256x256 batch 1 forwards with torch.set_num_threads(host)
    and is stopped after `budget_s` seconds; whatever finished is reported (the first forward is the warm-up)."""
    cmd = [sys.executable, os.path.abspath(__file__), "--cpu-all-cores-probe"]
    runs = []
    try:
        p = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, cwd=ROOT, start_new_session=True)
        try:
            out, _ = p.communicate(timeout=budget_s)
        except subprocess.TimeoutExpired:
            os.killpg(p.pid, 9)
            out, _ = p.communicate()
        for ln in out.decode().splitlines():
            if ln.startswith("{"):
                runs.append(json.loads(ln))
    except (OSError, ValueError) as e:
        return {"threads": host, "images_per_sec": None, "note": "probe failed: %r" % (e,)}
    timed = [r["seconds"] for r in runs if r["run"] > 0]
    if timed:
        return {"threads": host, "images_per_sec": 1.0 / float(np.median(timed)), "runs": len(timed), "size": 256, "batch": 1,
                "note": "oracle with torch.set_num_threads(all %d host CPUs), median after one warm-up, stopped after %.0f s" % (host, budget_s)}
    if runs:
        return {"threads": host, "images_per_sec": 1.0 / runs[0]["seconds"], "runs": 0, "size": 256, "batch": 1,
                "note": "only the warm-up forward finished within %.0f s on all %d host CPUs (oversubscribed oneDNN): its own time" % (budget_s, host)}
    return {"threads": host, "images_per_sec": None, "runs": 0, "size": 256, "batch": 1,
            "note": "no 256x256 forward finished within %.0f s on all %d host CPUs (oversubscribed oneDNN); < %.3f images/sec" % (budget_s, host, 1.0 / budget_s)}


PEAK_HEADROOM = 1.08      # clock headroom over the guide's nominal peak before an executed rate is called impossible


def rate_violations(rep, peak, where):
    """an EXECUTED rate above the MFMA peak is a bookkeeping error (round 5 printed 1.03x / 1.49x for the
    symmetric score GEMM), never a measurement.  Returns the offending labels; main() fails the line when there are any."""
    bad = []
    for r in rep:
        if r["flops_executed"] > 0 and r["total_ms"] > 0:
            tf = r["flops_executed"] / (r["total_ms"] * 1e-3) / 1e12
            if tf > peak * PEAK_HEADROOM:
                bad.append({"where": where, "kernel": r["kernel"], "executed_tflops": round(tf, 2), "peak": peak})
    return bad


TRACE_US = {}      # label -> {"avg_us", "launches"} from the rocprofv3 --kernel-trace child pass (filled by pmc_traffic)


def pmc_traffic(argv_child, timeout_s=240):
    """HBM bytes per launch and kernel label from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE -- they do not fit
    one pass: MI355X_MICROARCH.md 'rocprofv3 PMC slots') over a 1-step child run of this script.  FETCH_SIZE is doubled
    (gfx950 tallies 128-B read requests at 64 B, same guide); both counters are in KiB."""
    exe = shutil.which("rocprofv3")
    if exe is None:
        return None, "rocprofv3 not found"
    if any(k.startswith("ROCPROF") for k in os.environ) or "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None, "already running under a profiler"
    out = {}
    tmp = tempfile.mkdtemp(prefix="se_pmc_", dir="/tmp")
    try:
        # pass 0: --kernel-trace alone -> average launch duration per label WITHOUT the in-library event pairs (which cost the
        # first launch after a different kernel ~50 us at 256x256 batch 32: the events' own cache maintenance)
        d = os.path.join(tmp, "trace")
        # SE_FORK_DEFAULT=0 in every child pass: per-kernel durations / bytes are properties of ONE kernel only when nothing else
        # shares the chip; the timed headline runs the default two-stream plan (config.execution)
        cmd = [exe, "--kernel-trace", "-d", d, "--output-format", "csv", "--", sys.executable, os.path.join(ROOT, "bench.py")] + argv_child
        p = subprocess.run(cmd, cwd="/tmp", env=dict(os.environ, TMPDIR="/tmp", SE_FORK_DEFAULT="0"), stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        files = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
        if p.returncode == 0 and files:
            dur = {}
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    lab = label_of(r["Kernel_Name"])
                    if lab:
                        dur.setdefault(lab, []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
            TRACE_US.clear()
            TRACE_US.update({lab: {"avg_us": round(sum(v) / len(v), 2), "launches": len(v)} for lab, v in dur.items()})
        for counter, scale in (("FETCH_SIZE", 2.0 * 1024.0), ("WRITE_SIZE", 1024.0)):
            d = os.path.join(tmp, counter)
            cmd = [exe, "--kernel-trace", "--pmc", counter, "-d", d, "--output-format", "csv", "--", sys.executable,
                   os.path.join(ROOT, "bench.py")] + argv_child
            env = dict(os.environ, TMPDIR="/tmp", SE_FORK_DEFAULT="0")
            p = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
            files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
            if p.returncode != 0 or not files:
                return None, "rocprofv3 %s pass failed (rc %d): %s" % (counter, p.returncode, p.stderr.decode()[-200:])
            per = {}
            with open(files[0]) as f:
                for r in csv.DictReader(f):
                    if r["Counter_Name"] != counter:
                        continue
                    lab = label_of(r["Kernel_Name"])
                    if lab:
                        per.setdefault(lab, []).append(float(r["Counter_Value"]) * scale)
            for lab, v in per.items():
                out.setdefault(lab, 0.0)
                out[lab] += sum(v) / len(v)
        return {k: round(v) for k, v in out.items()}, "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, child run of this invocation"
    except (subprocess.TimeoutExpired, OSError, KeyError, ValueError) as e:
        return None, "rocprofv3 failed: %r" % (e,)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def parse_cpulist(text):
    """'0-63,128-191' -> [0..63, 128..191] (the format of /sys/devices/system/node/node*/cpulist)"""
    cpus = []
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus += list(range(int(lo), int(hi or lo) + 1))
    return cpus


def cpu_core_key(c):
    """(package, core, cpu) of logical CPU c from sysfs, so that a sorted pool keeps SMT siblings adjacent and a contiguous
    slice owns whole cores; (0, c, c) where sysfs does not say"""
    try:
        base = "/sys/devices/system/cpu/cpu%d/topology/" % c
        with open(base + "physical_package_id") as f:
            pk = int(f.read())
        with open(base + "core_id") as f:
            return (pk, int(f.read()), c)
    except (OSError, ValueError):
        return (0, c, c)


def rank_cpu_slice(local_rank, local_world, allowed, node_cpus=None, node_peers=None, key=None):
    """CPUs rank `local_rank` of `local_world` ranks on this host pins itself to (host threads of a rank: weight packing,
    the launch loop, the oracle legs of rank 0).
      * the GPU's NUMA node is known (node_cpus = that node's CPUs, node_peers = sorted local ranks whose GPU sits on the
        same node): the node's allowed CPUs are divided among those ranks -- host memory and the launch thread stay on the
        socket the GPU hangs off;
      * unknown: a contiguous 1/local_world slice of the allowed CPUs.
    Never empty: falls back to every allowed CPU."""
    allowed = sorted(allowed, key=key)
    pool, idx, cnt = allowed, local_rank, local_world
    if node_cpus and node_peers and local_rank in node_peers:
        on_node = [c for c in allowed if c in set(node_cpus)]
        if len(on_node) >= len(node_peers):
            pool, idx, cnt = on_node, node_peers.index(local_rank), len(node_peers)
    per = len(pool) // max(cnt, 1)
    sl = pool[idx * per:(idx + 1) * per] if per >= 1 else []
    return sorted(sl or allowed)


def gpu_numa_cpus(dev_index):
    """CPUs of the NUMA node GPU `dev_index` is attached to (sysfs, through its PCI address), or None"""
    try:
        bdf = torch.cuda.get_device_properties(dev_index).pci_bus_id if hasattr(torch.cuda.get_device_properties(dev_index), "pci_bus_id") else None
        if bdf is None:
            p = torch.cuda.get_device_properties(dev_index)
            bdf = "%04x:%02x:%02x.0" % (getattr(p, "pci_domain_id", 0), p.pci_bus_id, p.pci_device_id)
        with open("/sys/bus/pci/devices/%s/numa_node" % str(bdf).lower()) as f:
            node = int(f.read().strip())
        if node < 0:
            return None, None
        with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
            return node, parse_cpulist(f.read())
    except (OSError, ValueError, AttributeError, RuntimeError):
        return None, None


def pin_rank(local_rank, local_world, dev_index):
    """Per-rank host placement: CPU affinity next to the rank's GPU and a thread budget to match, so that
    eight ranks' weight packing / oracle legs do not oversubscribe the host (oneDNN collapses when they do, tools/cpu_probe.py).
    HIP_VISIBLE_DEVICES policy: NOT narrowed per rank -- every rank sees every GPU and selects LOCAL_RANK with
    torch.cuda.set_device (RCCL's peer-to-peer transport over xGMI needs the peers visible); a caller-set
    HIP_VISIBLE_DEVICES is honoured, LOCAL_RANK then indexes the visible list.  Returns a description for the JSON line."""
    try:
        allowed = os.sched_getaffinity(0)
    except (AttributeError, OSError):
        return {"affinity": None}
    node, cpus = gpu_numa_cpus(dev_index)
    peers = None
    if cpus is not None and local_world > 1:
        peers = [r for r in range(local_world) if gpu_numa_cpus(r)[0] == node] if torch.cuda.device_count() >= local_world else None
    sl = rank_cpu_slice(local_rank, local_world, allowed, cpus, peers, key=cpu_core_key) if local_world > 1 else sorted(allowed)
    try:
        os.sched_setaffinity(0, sl)
    except OSError:
        return {"affinity": None}
    # ... and never more than this rank's share of the CPUs the cgroup really grants (a 256-CPU host with a 16-CPU quota: eight
    # ranks x 32 threads would only get every rank throttled)
    from sketchedit_amd.hostinfo import effective_cpus
    threads = max(1, min(32, len(sl), effective_cpus() // max(local_world, 1) or 1))
    torch.set_num_threads(threads)
    return {"affinity": "%d CPUs (%d..%d)" % (len(sl), sl[0], sl[-1]), "numa_node": node, "threads": threads,
            "hip_visible_devices": os.environ.get("HIP_VISIBLE_DEVICES")}


def sysfs_gpus():
    """AMD display / processing-accelerator PCI functions in bus order -- the order HIP enumerates them in by default -- with
    their NUMA nodes, from sysfs alone (no HIP context): [(bdf, numa_node or None)]"""
    out = []
    base = "/sys/bus/pci/devices"
    try:
        for bdf in sorted(os.listdir(base)):
            try:
                with open(os.path.join(base, bdf, "vendor")) as f:
                    if f.read().strip() != "0x1002":
                        continue
                with open(os.path.join(base, bdf, "class")) as f:
                    cls = f.read().strip()
                if not (cls.startswith("0x0302") or cls.startswith("0x0380") or cls.startswith("0x1200")):
                    continue
                with open(os.path.join(base, bdf, "numa_node")) as f:
                    node = int(f.read().strip())
                out.append((bdf, node if node >= 0 else None))
            except (OSError, ValueError):
                continue
    except OSError:
        pass
    return out


def dry_run_topology(n, args):
    """`--gpus N --dry-run-topology`: what the N ranks of a real run WOULD use -- GPU, NUMA node, CPU slice, thread budget,
    rendezvous and RCCL environment, bytes of the one collective -- computed from sysfs without touching a GPU, so that the
    first 8-GPU run is diagnosable from its inputs.  One JSON object on stdout."""
    from sketchedit_amd.hostinfo import cgroup_cpu_quota, effective_cpus
    try:
        allowed = sorted(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        allowed = list(range(os.cpu_count() or 1))
    gpus = sysfs_gpus()
    node_cpus = {}
    for _, node in gpus:
        if node is not None and node not in node_cpus:
            try:
                with open("/sys/devices/system/node/node%d/cpulist" % node) as f:
                    node_cpus[node] = parse_cpulist(f.read())
            except (OSError, ValueError):
                node_cpus[node] = None
    ranks = []
    for r in range(n):
        bdf, node = gpus[r] if r < len(gpus) else (None, None)
        cpus = node_cpus.get(node) if node is not None else None
        peers = [q for q in range(n) if q < len(gpus) and gpus[q][1] == node] if cpus else None
        sl = rank_cpu_slice(r, n, allowed, cpus, peers, key=cpu_core_key) if n > 1 else allowed
        ranks.append({"rank": r, "local_rank": r, "hip_device": r, "gpu_pci": bdf, "numa_node": node,
                      "cpu_affinity": "%d CPUs (%d..%d)" % (len(sl), sl[0], sl[-1]), "threads": max(1, min(32, len(sl), effective_cpus() // n or 1)),
                      "shard": "images [%d, %d) of the global batch %d" % (r * args.batch, (r + 1) * args.batch, n * args.batch)})
    env = {k: os.environ.get(k) for k in ("HIP_VISIBLE_DEVICES", "ROCR_VISIBLE_DEVICES", "HSA_ENABLE_IPC_MODE_LEGACY", "NCCL_MAX_NCHANNELS",
                                          "NCCL_DEBUG", "MASTER_ADDR", "MASTER_PORT")}
    return {"dry_run_topology": True, "n_gpus": n, "gpus_in_sysfs": len(gpus), "ranks": ranks,
            "host": {"cpus_visible": os.cpu_count(), "cpus_allowed": len(allowed), "cgroup_cpu_quota": cgroup_cpu_quota(), "cpus_effective": effective_cpus()},
            "launch": "one process per GPU: python -m torch.distributed.run --nnodes=1 --nproc-per-node %d --master-addr 127.0.0.1 ... bench.py --gpus %d "
                      "(or `python bench.py --gpus %d`: self-launch, MASTER_ADDR=127.0.0.1, a free port)" % (n, n, n),
            "backend": args.backend + (" (= RCCL over xGMI)" if args.backend == "nccl" else ""),
            "visible_devices_policy": "not narrowed per rank: every rank sees every GPU and selects LOCAL_RANK (RCCL's P2P transport needs the peers visible); "
                                      "a caller-set HIP_VISIBLE_DEVICES is honoured",
            "environment": dict(env, NCCL_MAX_NCHANNELS_would_be=(str(args.nccl_channels) if args.nccl_channels > 0 else env["NCCL_MAX_NCHANNELS"]),
                                HSA_ENABLE_IPC_MODE_LEGACY_required="0 (dmabuf IPC: RCCL / cross-process device memory fail without it on these hosts)"),
            "collective": {"what": "ONE all_gather_into_tensor of the packed (B,4,H,W) outputs per step" + ("" if args.no_overlap else ", on a side stream under the next forward"),
                           "bytes_per_rank_per_step": args.batch * 4 * args.size * args.size * 4,
                           "bytes_gathered_per_rank_per_step": n * args.batch * 4 * args.size * args.size * 4}}


def self_launch(n):
    """`python bench.py --gpus N` without a launcher: this process becomes the launcher of N rank processes running this
    very command line (one per GPU, the environment torch.distributed.run would give them) and relays rank 0's stdout.
    A rank that fails takes the others down (their group) so that a dead peer cannot leave the rest in a collective."""
    import signal
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    procs = []
    for r in range(n):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=None if r == 0 else subprocess.DEVNULL, start_new_session=True))
    rc = 0
    live = list(procs)
    try:
        while live and rc == 0:
            time.sleep(0.05)
            for p in list(live):
                c = p.poll()
                if c is not None:
                    live.remove(p)
                    rc = rc or c
    finally:
        for p in live:                       # only reached with survivors when a rank failed or on KeyboardInterrupt
            try:
                os.killpg(p.pid, signal.SIGTERM)
            except OSError:
                pass
        for p in live:
            try:
                p.wait(timeout=10)
            except subprocess.TimeoutExpired:
                os.killpg(p.pid, signal.SIGKILL)
    return rc if rc >= 0 else 1


def secondary_config(dev_index, size, batch, dtype, steps=12, warmup=3):
    """One more BASELINE.json configuration, timed inside the SAME invocation after the headline (configs
    3 and 5 were builder-measured only): `steps` forwards between device fences on a fresh engine, the in-library profiler
    pass for the executed-FLOP fractions, and the parity of image 0 of the last timed step against the oracle (fp32), or
    against the oracle's bf16 mode plus the fp32 oracle (bf16: the error triangle of tests/test_gpu_bf16.py)."""
    from oracle import sketchedit_oracle as O
    dev = torch.device("cuda", dev_index)
    eng = Engine(dev_index)
    WMn, WGn = synth.make_state_dict("M", 0), synth.make_state_dict("G", 0)
    eng.load_state_dict("M", WMn)
    eng.load_state_dict("G", WGn)
    if dtype == "bf16":
        eng.set_precision("bf16")
    img_h, sk_h = synth.make_inputs(batch, size, size, seed=1234)
    img, sk = torch.from_numpy(img_h).to(dev), torch.from_numpy(sk_h).to(dev)
    out = torch.empty((batch, 4, size, size), dtype=torch.float32, device=dev)
    for _ in range(warmup):
        eng.inference_packed(img, sk, FLAGS, out)
    torch.cuda.synchronize(dev)
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.inference_packed(img, sk, FLAGS, out)
    torch.cuda.synchronize(dev)
    elapsed = time.perf_counter() - t0
    im0 = out[0:1].clone()
    eng.profile(True)
    for _ in range(2):
        eng.inference_packed(img, sk, FLAGS, out)
    rep = eng.profile_report()["kernels"]
    eng.profile(False)
    peak = MFMA_PEAK_TFLOPS[dtype]
    dom = max(rep, key=lambda r: r["total_ms"])
    mfma = [r for r in rep if r["flops_executed"] > 0]
    mf_ms, mf_ex = sum(r["total_ms"] for r in mfma), sum(r["flops_executed"] for r in mfma)
    att_ms = sum(r["total_ms"] for r in rep if r["kernel"].startswith("att_")) / 2
    WM = {k: torch.from_numpy(v) for k, v in WMn.items()}
    WG = {k: torch.from_numpy(v) for k, v in WGn.items()}
    torch.set_num_threads(min(os.cpu_count() or 1, 32))
    ref = O.inference(WM, WG, img_h[:1], sk_h[:1])
    hard = (im0[:, 3:4] > 0.5).float().cpu()
    if dtype == "f32":
        flips = int((hard != ref["hard_mask"]).sum())
        comp = ref["composed"]
        if flips:      # a logit within float noise of the threshold: compare the composite for the hard mask this run used
            _, fine2 = O.netG_forward(WG, img_h[:1], img_h[:1], hard, hard, sk_h[:1])
            comp = fine2 * ref["mask"] + torch.from_numpy(img_h[:1]) * (1 - ref["mask"])
        parity = {"comparator": "fp32 oracle, image 0 of the last timed step", "tolerance": 1e-3, "hard_mask_flips": flips,
                  "max_abs_mask": float((im0[:, 3:4].cpu() - ref["mask"]).abs().max()),
                  "max_abs_composed": float((im0[:, 0:3].cpu() - comp).abs().max())}
    else:
        refb = O.inference(WM, WG, img_h[:1], sk_h[:1], act_dtype=torch.bfloat16)
        hb = refb["hard_mask"].to(dev)
        _, fine = eng.netG(img[:1].contiguous(), img[:1].contiguous(), hb, hb, sk[:1].contiguous(), FLAGS)
        _, f32g = O.netG_forward(WG, img_h[:1], img_h[:1], refb["hard_mask"], refb["hard_mask"], sk_h[:1])
        parity = {"comparator": "oracle bf16 mode (same roundings, fp32 accumulate), image 0", "tolerance": "3e-2 max-abs; error triangle x1.25 (tests/test_gpu_bf16.py)",
                  "max_abs_mask": float((im0[:, 3:4].cpu() - refb["mask"]).abs().max()),
                  "hard_mask_flip_fraction": float((hard != refb["hard_mask"]).float().mean()),
                  "max_abs_fine_given_oracle_mask": float((fine.cpu() - refb["fine"]).abs().max()),
                  "triangle_mask": float((im0[:, 3:4].cpu() - ref["mask"]).abs().max() / max(float((refb["mask"] - ref["mask"]).abs().max()), 1e-12)),
                  "triangle_fine": float((fine.cpu() - f32g).abs().max() / max(float((refb["fine"] - f32g).abs().max()), 1e-12))}
    eng.close()
    return {"rate_violations": rate_violations(rep, peak, "%dx%d batch %d %s" % (size, size, batch, dtype)),
            "config": "%dx%d batch %d %s, 1 GPU" % (size, size, batch, "fp32" if dtype == "f32" else "bf16 MFMA, fp32 accumulate"),
            "dtype": dtype, "steps": steps, "warmup": warmup, "ms_per_step": round(1e3 * elapsed / steps, 4), "value": round(batch * steps / elapsed, 2),
            "unit": "images/sec",
            # the headline's two conventions: frac = EXECUTED multiply-adds / peak (<= 1), frac_algorithmic = reference-defined
            # FLOPs / peak (above 1 where a Winograd / sub-pixel form executes fewer products)
            "roofline": {"kernel": dom["kernel"], "frac": round(dom["flops_executed"] / (dom["total_ms"] * 1e-3) / 1e12 / peak, 4),
                         "frac_algorithmic": round(dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12 / peak, 4), "peak": peak,
                         "forward_executed_frac": round(mf_ex / (mf_ms * 1e-3) / 1e12 / peak, 4),
                         "forward_algorithmic_frac": round(sum(r["flops"] for r in mfma) / (mf_ms * 1e-3) / 1e12 / peak, 4),
                         "attention_ms": round(att_ms, 4)},
            "parity": parity}


# ---------------------------------------------------------------------------------------------------------------------
# End-to-end leg: test.py's loop -- PNG files in, PNG files out -- through
# sketchedit_amd/pipeline.py, with the stage breakdown and the host's own decode + encode capability beside it.
# ---------------------------------------------------------------------------------------------------------------------
def _write_pair(args):
    d, i, size = args
    from PIL import Image
    rng = np.random.default_rng(1000 + i)
    low = rng.integers(0, 256, (size // 8, size // 8, 3), dtype=np.uint8)        # smooth content + grain: PNG sizes like photographs
    img = np.asarray(Image.fromarray(low).resize((size, size), Image.BICUBIC)).astype(np.int16)
    img = (img + rng.integers(-6, 7, img.shape)).clip(0, 255).astype(np.uint8)
    Image.fromarray(img).save(os.path.join(d, "images", "u%04d.png" % i))
    sk = ((rng.random((size, size)) < 0.005) * 255).astype(np.uint8)              # sketch density of the bundled samples
    Image.fromarray(sk).save(os.path.join(d, "edges", "u%04d.png" % i))


def e2e_leg(args):
    """Child mode `--e2e`: unique synthetic PNG pairs on tmpfs, listed `--e2e-images` times through symlinks.  For each PNG
    writer (pil: this repo's files; fast: the reference's cv2.imwrite settings) the host codec capability is measured FIRST (no
    GPU context in the process yet: it forks), then the pipelined test.py loop runs over the list.  Pools are sized by the CPUs
    the process may really use (affinity AND cgroup quota: the GPU boxes grant 16 CPUs' worth of time on a 256-CPU host)."""
    import multiprocessing as mp
    from argparse import Namespace
    from sketchedit_amd.hostinfo import cgroup_cpu_quota, cgroup_throttle_stats, effective_cpus
    from sketchedit_amd.pipeline import InferencePipeline, encoder_rate_on, host_codec_capability
    ncpu = effective_cpus()
    torch.set_num_threads(max(1, min(torch.get_num_threads(), ncpu // 2 or 1)))
    S, B, N, U = args.size, args.batch, args.e2e_images, min(args.e2e_unique, args.e2e_images)
    base = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else "/tmp"
    d = tempfile.mkdtemp(prefix="se_e2e_", dir=base)
    try:
        for sub in ("images", "edges", "out", "cap"):
            os.makedirs(os.path.join(d, sub))
        t0 = time.perf_counter()
        with mp.get_context("fork").Pool(min(ncpu, 64)) as pool:
            pool.map(_write_pair, [(d, i, S) for i in range(U)], chunksize=8)
        names = []
        for i in range(N):
            n = "f%05d.png" % i
            for sub in ("images", "edges"):
                os.symlink(os.path.join(d, sub, "u%04d.png" % (i % U)), os.path.join(d, sub, n))
            names.append(n)
        with open(os.path.join(d, "list.txt"), "w") as f:
            f.write("\n".join(names) + "\n")
        gen_s = time.perf_counter() - t0
        ip = [os.path.join(d, "images", "u%04d.png" % i) for i in range(U)]
        mp_ = [os.path.join(d, "edges", "u%04d.png" % i) for i in range(U)]
        # pool sizes: `reserve` CPUs stay with the main process (launch thread, staging copies, the loader's and the encoders'
        # helper threads, the HIP runtime's): under a cgroup quota an over-committed pool does not just slow down, it gets EVERY
        # thread of the cgroup stopped until the period ends -- the launch thread too.  The rest is split by the stages' cost
        # per image (decode ~1.6 ms per pair; encode ~7 ms with PIL's defaults, ~1.3 ms with the fast writer, EPYC 9575F)
        plans = {}
        reserve = args.e2e_reserve if args.e2e_reserve >= 0 else max(1, ncpu // 5)
        for writer, dec_share in (("pil", 0.2), ("fast", 0.5)):
            w = args.e2e_workers or max(1, int(round((ncpu - reserve) * dec_share)))
            e = args.e2e_encoders or max(1, ncpu - reserve - w)
            plans[writer] = (w, e, host_codec_capability(ip, mp_, w, e, seconds=args.e2e_cap_seconds, writer=writer, out_dir=os.path.join(d, "cap")))
        # ---- the GPU side starts here
        from sketchedit_amd import data, models
        opt = Namespace(gpu_ids=[0], isTrain=False, model="editline2", netG="deepfillc2", init_type=None, init_variance=0.02,
                        use_cam=True, pool_type="max", no_mask_cc=False, no_mask_coarse=False, joint_train_inp=True, isSkip=True,
                        which_epoch="latest", checkpoints_dir=d, name="bench", batchSize=B, nThreads=1, serial_batches=True,
                        dataset_mode="testimage", image_dirs=os.path.join(d, "images"), mask_dirs=os.path.join(d, "edges"),
                        image_lists=os.path.join(d, "list.txt"), image_postfix=".png", mask_postfix=".png", output_labels=None,
                        output_dir=os.path.join(d, "out"), output_mask_dir=None, u8_io=True, conservative_mask=False)
        torch.cuda.set_device(0)
        model = models.create_model(opt)
        model.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
        model.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
        model.cuda()
        model.eval()
        # warm-up: weights packed, workspace sized, kernels loaded (not part of the timed run)
        wu = torch.zeros((B, S, S, 3), dtype=torch.uint8), torch.zeros((B, S, S), dtype=torch.uint8)
        for _ in range(3):
            model.inference_u8({"image_u8": wu[0], "mask_u8": wu[1]}, low_latency=model.batch_mode(S, S))
        torch.cuda.synchronize()
        legs = {}
        for writer, (w, e, cap) in plans.items():
            opt.nThreads = w
            thr0 = cgroup_throttle_stats()
            loader = data.create_dataloader(opt)
            pipe = InferencePipeline(model, opt.output_dir, None, encode_threads=e, depth=args.e2e_depth, timing=True, verbose=False,
                                     encode_procs=0 if args.e2e_encode_threads else e, max_pending_batches=args.e2e_pending, png_writer=writer,
                                     decode_procs=0 if args.e2e_dataloader else w)
            try:
                st = dict(pipe.run(loader, float("inf"), B) if args.e2e_dataloader else pipe.run_paths(loader.dataset, float("inf"), B))
            finally:
                pipe.close()
            thr1 = cgroup_throttle_stats()
            img = st["images"]
            # the encoders' rate on what the network really wrote (procedural weights paint the hole with texture that
            # compresses worse than the synthetic inputs the host-only capability leg encodes): the fair yardstick for `encode`
            from PIL import Image
            outs = sorted(os.listdir(opt.output_dir))
            sample = np.array(Image.open(os.path.join(opt.output_dir, outs[len(outs) // 2])).convert("RGB"), dtype=np.uint8)
            out_bytes = float(np.mean([os.path.getsize(os.path.join(opt.output_dir, f)) for f in outs[:: max(1, len(outs) // 64)]]))
            os.makedirs(os.path.join(d, "rate"), exist_ok=True)
            enc_out = encoder_rate_on(sample, os.path.join(d, "rate"), e, writer, seconds=args.e2e_cap_seconds)
            stage = {"decode": cap["decode_ips"], "h2d": img / max(st["h2d_ms"] * 1e-3, 1e-9), "forward": img / max(st["forward_ms"] * 1e-3, 1e-9),
                     "d2h": img / max(st["d2h_ms"] * 1e-3, 1e-9), "encode": enc_out}
            e2e = img / st["wall_s"]
            legs[writer] = {
                "e2e_images_per_sec": round(e2e, 1), "images": img, "files_written": len(os.listdir(opt.output_dir)), "wall_s": round(st["wall_s"], 3),
                "decode_workers": w, "encode_workers": e, "encoders_are": "threads" if args.e2e_encode_threads else "processes",
                "decoders_are": "DataLoader workers" if args.e2e_dataloader else "processes writing into the shared page-locked input ring",
                # standalone rate of every stage (images/sec): decode = the host alone at this worker count (no GPU), encode = the
                # encoder processes alone on an image the network produced;
                # h2d / forward / d2h = images / summed HIP-event time of that stage inside the pipelined run
                "stage_images_per_sec": {k: round(v, 1) for k, v in stage.items()},
                "bottleneck": min(stage, key=stage.get),
                "host_codec": dict({k: round(v, 1) for k, v in cap.items()}, encode_ips_on_network_output=round(enc_out, 1),
                                   mean_output_png_bytes=round(out_bytes)),
                # the yardstick: what these CPUs decode + encode in parallel with no GPU in the loop
                "e2e_over_host_codec": round(e2e / max(cap["both_ips"], 1e-9), 3),
                # ... and against the slowest stage's standalone rate (the forward when the codec outruns the GPU)
                "e2e_over_slowest_stage": round(e2e / max(min(stage.values()), 1e-9), 3),
                "main_thread_s": {k: round(st[k], 3) for k in ("decode_wait_s", "sync_wait_s", "encode_backpressure_s", "encode_drain_s", "issue_stage_s",
                                                                "issue_h2d_s", "issue_forward_s", "issue_d2h_s", "submit_encode_s")},
                "encode_cpu_s_summed": round(st["encode_cpu_s"], 2),
                # scheduler periods (100 ms) of the run / periods in which the cgroup's CPU quota stopped every thread / time stopped
                "cgroup_throttling": (None if not (thr0 and thr1) else {"periods": thr1[0] - thr0[0], "throttled_periods": thr1[1] - thr0[1],
                                                                          "throttled_thread_seconds": round((thr1[2] - thr0[2]) * 1e-6, 3)})}
        main_leg = legs["pil"]
        line = {"metric": "e2e_images_per_sec", "value": main_leg["e2e_images_per_sec"], "unit": "images/sec", "size": S, "batch": B,
                "what": "test.py's loop, files to files: PNG pair on tmpfs -> decode (worker processes, straight into a shared page-locked uint8 ring) -> H2D -> "
                        "se_inference_u8io -> D2H into a shared page-locked ring -> PNG encode + write (worker processes); %d list entries "
                        "over %d unique synthetic pairs.  value = the `pil` writer (PIL defaults: the files this repo always wrote); `fast` = the "
                        "reference's cv2.imwrite settings (SUB filter, zlib 1, RLE), same pixels" % (N, U),
                "host": {"cpus_visible": os.cpu_count(), "cgroup_cpu_quota": cgroup_cpu_quota(), "cpus_effective": ncpu,
                         "dataset_seconds": round(gen_s, 2), "tmpfs": base},
                "writers": legs, "bottleneck": main_leg["bottleneck"], "e2e_over_host_codec": main_leg["e2e_over_host_codec"],
                "e2e_over_slowest_stage": main_leg["e2e_over_slowest_stage"],
                "stage_images_per_sec": main_leg["stage_images_per_sec"],
                "e2e_images_per_sec_fast_writer": legs["fast"]["e2e_images_per_sec"]}
        return line
    finally:
        shutil.rmtree(d, ignore_errors=True)


def e2e_child(size, batch, timeout_s=240):
    """the e2e leg as a fresh child process of the default invocation (its host measurements run before it touches the GPU)"""
    cmd = [sys.executable, os.path.abspath(__file__), "--e2e", "--size", str(size), "--batch", str(batch)]
    try:
        p = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout_s)
        lines = [ln for ln in p.stdout.decode().splitlines() if ln.startswith("{")]
        if p.returncode == 0 and lines:
            return json.loads(lines[-1])
        return {"metric": "e2e_images_per_sec", "value": None, "error": "rc %d: %s" % (p.returncode, p.stderr.decode()[-400:])}
    except (subprocess.TimeoutExpired, OSError, ValueError) as e:
        return {"metric": "e2e_images_per_sec", "value": None, "error": repr(e)}


def main():
    faulthandler.enable()
    if "--cpu-all-cores-probe" in sys.argv[1:]:
        return cpu_all_cores_probe()
    # The contract is ONE line on stdout.  RCCL prints a version banner through C stdio (it shows up after the JSON line
    # when the process exits), so everything written to file descriptor 1 during the run is sent to stderr and the JSON
    # line goes to the real stdout at the very end.
    sys.stdout.flush()
    real_stdout = os.dup(1)
    os.dup2(2, 1)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--size", type=int, default=256)
    ap.add_argument("--batch", type=int, default=32, help="images per GPU per step")
    ap.add_argument("--dtype", default="f32", choices=["f32", "bf16"], help="f32: BASELINE config 2/3; bf16: config 5 (bf16 MFMA, fp32 accumulate)")
    ap.add_argument("--low-latency", default="auto", choices=["auto", "on", "off"], help="SE_FLAG_LOW_LATENCY (auto: by call size)")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured hipGraph (SE_FLAG_GRAPH)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true")
    ap.add_argument("--no-traffic", action="store_true", help="skip the rocprofv3 PMC child runs (roofline.traffic = null)")
    ap.add_argument("--layers", action="store_true", help="add the per-layer timing table to the JSON line")
    ap.add_argument("--no-secondary", action="store_true", help="skip BASELINE configs 3 and 5 (timed after the headline at the default invocation)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend for --gpus > 1 (nccl = RCCL)")
    ap.add_argument("--device", type=int, default=-1, help="force this HIP device for every rank (test aid)")
    ap.add_argument("--check-gather", action="store_true", help="rank 0 verifies the gathered outputs (test aid)")
    ap.add_argument("--no-overlap", action="store_true", help="N > 1: all-gather on the compute stream instead of a side stream")
    ap.add_argument("--force-dist", action="store_true", help="initialise the process group and run the gather even with one rank")
    ap.add_argument("--nccl-channels", type=int, default=0, help="N > 1: cap RCCL's channel count (NCCL_MAX_NCHANNELS); 0 = RCCL's own choice")
    ap.add_argument("--dry-run-topology", action="store_true", help="print what the --gpus N ranks WOULD use (GPU, NUMA slice, threads, RCCL / rendezvous environment) and exit; needs no GPU")
    ap.add_argument("--e2e", action="store_true", help="only the end-to-end leg: PNG files -> test.py's pipelined loop -> PNG files (one JSON line)")
    ap.add_argument("--e2e-images", type=int, default=4000, help="list entries of the e2e leg (symlinks over --e2e-unique files)")
    ap.add_argument("--e2e-unique", type=int, default=1000, help="unique synthetic PNG pairs generated on tmpfs")
    ap.add_argument("--e2e-workers", type=int, default=0, help="decode worker processes (0: by host size)")
    ap.add_argument("--e2e-encoders", type=int, default=0, help="PNG encoder threads (0: by host size)")
    ap.add_argument("--e2e-depth", type=int, default=3, help="batches in flight on the device")
    ap.add_argument("--e2e-reserve", type=int, default=-1, help="CPUs left to the main process (-1: a fifth of the effective CPUs)")
    ap.add_argument("--e2e-pending", type=int, default=8, help="batches that may wait for their encoders")
    ap.add_argument("--e2e-dataloader", action="store_true", help="decode in DataLoader workers (--nThreads) instead of the pipeline's own decoder processes")
    ap.add_argument("--e2e-encode-threads", action="store_true", help="encoder THREADS in the main process instead of processes (GIL-bound near 2000 images/s)")
    ap.add_argument("--e2e-cap-seconds", type=float, default=3.0, help="seconds per host-codec capability leg (decode, encode, both)")
    ap.add_argument("--no-e2e", action="store_true", help="default invocation: skip the end-to-end child run")
    args = ap.parse_args()
    if args.dry_run_topology:
        os.write(real_stdout, (json.dumps(dry_run_topology(max(1, args.gpus), args)) + "\n").encode())
        return
    if args.e2e:
        line = e2e_leg(args)
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        return

    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        os.dup2(real_stdout, 1)
        raise SystemExit(self_launch(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; the line reports n_gpus = %d" % (args.gpus, world, world), file=sys.stderr)
    dist = None
    # --device / --backend exist only to exercise the N > 1 code path on a 1-GPU box (both ranks on cuda:0 over
    # gloo); the driver's runs use the defaults: one rank per GPU (LOCAL_RANK) over nccl = RCCL.
    dev_index = local_rank if args.device < 0 else args.device
    use_dist = world > 1 or args.force_dist
    torch.cuda.set_device(dev_index)
    if use_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29513")
        if args.backend == "nccl":
            # The exchange is small (33.5 MB per rank and step at 256x256 batch 32) and runs under the next forward; every
            # RCCL channel is a workgroup on a CU the convolutions want, so a cap MAY pay -- unmeasured (no multi-GPU box
            # in the build loop), hence opt-in: RCCL's own channel choice is the default.
            if args.nccl_channels > 0:
                os.environ["NCCL_MAX_NCHANNELS"] = str(args.nccl_channels)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", dev_index))
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    dev = torch.device("cuda", dev_index)
    placement = pin_rank(local_rank, int(os.environ.get("LOCAL_WORLD_SIZE", world)), dev_index)

    if os.environ.get("SE_BENCH_FAIL_RANK") == str(rank) and world > 1:      # test hook: a rank that dies after the rendezvous
        raise SystemExit(3)
    B, S = args.batch, args.size
    exit_code = 0
    eng = Engine(dev_index)
    eng.load_state_dict("M", synth.make_state_dict("M", 0))
    eng.load_state_dict("G", synth.make_state_dict("G", 0))
    if args.dtype == "bf16":
        eng.set_precision("bf16")
    low_latency = {"auto": None, "on": True, "off": False}[args.low_latency]
    img_h, sk_h = synth.make_inputs(B, S, S, seed=1234, first_index=rank * B)   # shard = rows of the global batch
    img = torch.from_numpy(img_h).to(dev)
    sk = torch.from_numpy(sk_h).to(dev)
    # The packed (B,4,H,W) outputs (composite + soft mask) are double-buffered: with more than one rank the
    # all-gather of step k runs on a side stream under the forward of step k+1 (the forward has no exchange inside).
    overlap = use_dist and not args.no_overlap
    outs = [torch.empty((B, 4, S, S), dtype=torch.float32, device=dev) for _ in range(2 if overlap else 1)]
    gathered_sets = [torch.empty((world * B, 4, S, S), dtype=torch.float32, device=dev) for _ in outs] if use_dist else []
    comm_stream = torch.cuda.Stream(device=dev) if overlap else None
    gather_done = [None, None]
    state = {"i": 0, "last": 0}

    def gather(o, g):
        # the only exchange of the path: ONE all-gather of the packed outputs
        if args.backend == "nccl":
            dist.all_gather_into_tensor(g, o)
        else:              # gloo cannot all-gather device tensors: stage through the host (test aid only)
            parts = [torch.empty(o.shape, dtype=torch.float32) for _ in range(world)]
            dist.all_gather(parts, o.cpu())
            g.copy_(torch.cat(parts, 0))

    def forward(o):
        if args.graph:
            r = state["graph_out"] = eng.inference(img, sk, FLAGS, low_latency=low_latency, graph=True)
            if use_dist:
                o[:, 0:3].copy_(r["composed"])
                o[:, 3:4].copy_(r["mask"])
        else:
            eng.inference_packed(img, sk, FLAGS, o, low_latency=low_latency)

    def step():
        i = state["i"]
        state["i"] = i + 1
        if not overlap:
            forward(outs[0])
            if use_dist:
                gather(outs[0], gathered_sets[0])
            return
        slot = i & 1
        main = torch.cuda.current_stream(dev)
        if gather_done[slot] is not None:          # the gather that last read this output buffer (step i-2)
            main.wait_event(gather_done[slot])
        forward(outs[slot])
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
        if use_dist:
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
    timed_image0 = None                                   # image 0 of the LAST timed step (parity leg)
    if rank == 0:
        g = state.get("graph_out")
        timed_image0 = torch.cat([g["composed"][0:1], g["mask"][0:1]], 1) if g else outs[state["last"]][0:1].clone()
    per_rank_ms, fwd_only_ms = None, None
    if use_dist:
        # diagnostics for the first real multi-GPU run (VERDICT r4 'Next round' 8), outside the timed region: every rank's own
        # ms/step, and the forward alone (no gather) over a few steps -- ms_per_step minus that is the EXPOSED part of the gather
        nfo = max(1, min(args.steps, 10))
        torch.cuda.synchronize(dev)
        tf = time.perf_counter()
        for _ in range(nfo):
            forward(outs[0])
        torch.cuda.synchronize(dev)
        fwd_only = 1e3 * (time.perf_counter() - tf) / nfo
        cdev = dev if args.backend == "nccl" else "cpu"
        mine = torch.tensor([1e3 * elapsed / args.steps, fwd_only], dtype=torch.float64, device=cdev)
        allr = [torch.zeros(2, dtype=torch.float64, device=cdev) for _ in range(world)]
        dist.all_gather(allr, mine)
        per_rank_ms = [round(float(a[0]), 4) for a in allr]
        fwd_only_ms = [round(float(a[1]), 4) for a in allr]
        t = torch.tensor([elapsed], dtype=torch.float64, device=cdev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        if args.check_gather:
            # every rank finds its own rows in the gathered batch; rank 0 recomputes image 0 of the last rank's shard
            out, gathered = outs[state["last"]], gathered_sets[state["last"]]
            ok = torch.equal(gathered[rank * B:(rank + 1) * B], out)
            if rank == 0:
                g0 = (world - 1) * B
                i1, s1 = synth.make_inputs(1, S, S, seed=1234, first_index=g0)
                ll = eng.is_low_latency(B, S, S, low_latency)       # same execution mode as the sharded run
                r1 = eng.inference(torch.from_numpy(i1).to(dev), torch.from_numpy(s1).to(dev), FLAGS, low_latency=ll)
                ok = ok and torch.equal(r1["composed"], gathered[g0:g0 + 1, 0:3]) and torch.equal(r1["mask"], gathered[g0:g0 + 1, 3:4])
            f = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if args.backend == "nccl" else "cpu")
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            if float(f.item()) != 1.0:
                raise SystemExit("gathered outputs do not match the per-rank outputs")

    # ---- per-kernel timing pass (HIP events on the launch stream), not part of the timed region
    peak = MFMA_PEAK_TFLOPS[args.dtype]
    roofline, kernels, full_rep, nprof, violations = None, None, None, 3, []
    if rank == 0:
        eng.profile(True)
        for _ in range(nprof):
            eng.inference_packed(img, sk, FLAGS, outs[0], low_latency=low_latency)
        full_rep = eng.profile_report()
        rep = full_rep["kernels"]
        eng.profile(False)
        kernels = {r["kernel"]: {"launches_per_step": r["launches"] // nprof,
                                 "ms_per_step": round(r["total_ms"] / nprof, 4),
                                 "avg_us": round(1e3 * r["total_ms"] / r["launches"], 2),
                                 "executed_tflops": round(r["flops_executed"] / (r["total_ms"] * 1e-3) / 1e12, 2) if r["flops_executed"] > 0 else None,
                                 "algorithmic_tflops": round(r["flops"] / (r["total_ms"] * 1e-3) / 1e12, 2) if r["flops"] > 0 else None,
                                 # mean workgroups per launch: < 256 leaves CUs of the 256-CU chip without work
                                 "workgroups_per_launch": r.get("workgroups") or None}
                   for r in rep}
        violations = rate_violations(rep, peak, "headline")
        dom = max(rep, key=lambda r: r["total_ms"])
        executed = dom["flops_executed"] / (dom["total_ms"] * 1e-3) / 1e12
        algorithmic = dom["flops"] / (dom["total_ms"] * 1e-3) / 1e12
        mfma = [r for r in rep if r["flops_executed"] > 0]
        mf_ms = sum(r["total_ms"] for r in mfma) / nprof
        mf_ex = sum(r["flops_executed"] for r in mfma) / nprof
        mf_al = sum(r["flops"] for r in mfma) / nprof
        roofline = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(executed, 3), "peak": peak,
                    "unit": "TFLOP/s", "frac": round(executed / peak, 4),
                    # SURVEY.md 8(d)'s definition: reference-defined FLOPs per launch / launch duration / peak -- above 1 when the
                    # kernel is a Winograd form (24 of 72 products executed); `frac` is the share of the MFMA pipe really used
                    "frac_algorithmic": round(algorithmic / peak, 4),
                    "traffic": None, "traffic_source": None,
                    "achieved_is": "executed multiply-add FLOPs per launch / average launch duration (HIP events)",
                    # the timed headline overlaps the independent branches of netG on two streams; a per-launch duration is a
                    # property of one kernel only when nothing else shares the chip, so this pass (and the rocprofv3 child passes)
                    # runs the same kernels on ONE stream
                    "measured_with": "branches serialised (profiler pass: one stream); the timed region runs the two-stream plan",
                    "algorithmic_tflops": round(algorithmic, 3),
                    "executed_over_algorithmic": round(dom["flops_executed"] / dom["flops"], 4),
                    "launches_per_step": dom["launches"] // nprof,
                    "avg_launch_us": round(1e3 * dom["total_ms"] / dom["launches"], 2),
                    "flops_executed_per_launch": dom["flops_executed"] / dom["launches"],
                    "flops_algorithmic_per_launch": dom["flops"] / dom["launches"],
                    # all MFMA kernels of the forward together (gated convs + attention GEMMs)
                    "forward_mfma_ms": round(mf_ms, 3),
                    "forward_executed_tflops": round(mf_ex / (mf_ms * 1e-3) / 1e12, 3),
                    "forward_executed_frac": round(mf_ex / (mf_ms * 1e-3) / 1e12 / peak, 4),
                    "forward_algorithmic_tflops": round(mf_al / (mf_ms * 1e-3) / 1e12, 3),
                    "forward_algorithmic_frac": round(mf_al / (mf_ms * 1e-3) / 1e12 / peak, 4),
                    # time-weighted share of the 256 CUs that the MFMA kernels' grids can occupy (one workgroup per CU
                    # counted as occupied; the batch-1 figure the low-latency mode exists to raise)
                    "cu_occupancy_by_grid": round(sum(r["total_ms"] * min(1.0, r.get("workgroups", 0) / 256.0) for r in mfma) /
                                                  max(sum(r["total_ms"] for r in mfma), 1e-9), 3)}
        if world == 1 and not args.no_traffic and not args.force_dist:
            child = ["--steps", "1", "--warmup", "1", "--size", str(S), "--batch", str(B), "--dtype", args.dtype,
                     "--low-latency", args.low_latency, "--no-cpu-baseline", "--no-parity", "--no-traffic", "--no-secondary"]
            tr, src = pmc_traffic(child)
            roofline["traffic"] = tr.get(dom["kernel"]) if tr else None
            roofline["traffic_source"] = src
            roofline["traffic_all"] = tr
            t_us = TRACE_US.get(dom["kernel"])
            if t_us:     # the same figure from the rocprofv3 --kernel-trace child run (no event pairs around the launches)
                roofline["rocprofv3_avg_launch_us"] = t_us["avg_us"]
                roofline["rocprofv3_frac"] = round(dom["flops_executed"] / dom["launches"] / (t_us["avg_us"] * 1e-6) / 1e12 / peak, 4)
                roofline["rocprofv3_frac_algorithmic"] = round(dom["flops"] / dom["launches"] / (t_us["avg_us"] * 1e-6) / 1e12 / peak, 4)
                roofline["rocprofv3_all"] = dict(TRACE_US)

    # ---- parity of this very run against the oracle on image 0 (CPU, rank 0)
    parity = None
    if rank == 0 and not args.no_parity:
        from oracle import sketchedit_oracle as O
        WM = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
        WG = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}
        torch.set_num_threads(min(os.cpu_count() or 1, 32))
        ref = O.inference(WM, WG, img_h[:1], sk_h[:1])
        # what is compared is image 0 of the packed (B,4,H,W) output the LAST TIMED step wrote (planes 0-2 composite, plane
        # 3 soft mask); the hard mask netG saw is the threshold of that very fp32 plane (editline2_model.py:347)
        r1 = {"composed": timed_image0[:, 0:3], "mask": timed_image0[:, 3:4], "hard": (timed_image0[:, 3:4] > 0.5).float()}
        flips = int((r1["hard"].cpu() != ref["hard_mask"]).sum())
        parity = {"max_abs_composed": float((r1["composed"].cpu() - ref["composed"]).abs().max()),
                  "max_abs_mask": float((r1["mask"].cpu() - ref["mask"]).abs().max()),
                  "hard_mask_flips": flips, "image": 0, "of": "the last timed step's own output (batch %d)" % B, "tolerance": 1e-3 if args.dtype == "f32" else None,
                  "comparator": "fp32 oracle"}
        if flips and args.dtype == "f32":
            # a soft-mask value within float noise of 0.5 thresholded differently (editline2_model.py:347): netG then saw a
            # different INPUT.  The composite is re-checked against the oracle's netG on the hard mask this run used.
            hard = r1["hard"].cpu()
            _, fine2 = O.netG_forward(WG, img_h[:1], img_h[:1], hard, hard, sk_h[:1])
            comp2 = fine2 * ref["mask"] + torch.from_numpy(img_h[:1]) * (1 - ref["mask"])
            parity["max_abs_composed_same_hard_mask"] = float((r1["composed"].cpu() - comp2).abs().max())
        if args.dtype == "bf16":
            # the comparator of the bf16 path is the oracle's bf16 mode (same roundings, fp32 accumulation); netG is
            # compared on the oracle's hard mask so that a threshold flip does not change its input
            refb = O.inference(WM, WG, img_h[:1], sk_h[:1], act_dtype=torch.bfloat16)
            hard = refb["hard_mask"].to(dev)
            _, fine = eng.netG(img[:1].contiguous(), img[:1].contiguous(), hard, hard, sk[:1].contiguous(), FLAGS)
            parity.update({"bf16_oracle": {"max_abs_mask": float((r1["mask"].cpu() - refb["mask"]).abs().max()),
                                           "hard_mask_flip_fraction": float((r1["hard"].cpu() != refb["hard_mask"]).float().mean()),
                                           "max_abs_fine_given_oracle_mask": float((fine.cpu() - refb["fine"]).abs().max()),
                                           "mean_abs_fine_given_oracle_mask": float((fine.cpu() - refb["fine"]).abs().mean()),
                                           "tolerance": "3e-2 max-abs, 3e-3 mean-abs (tests/test_gpu_bf16.py)"}})

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        cpu = cpu_baseline()

    # ---- BASELINE configs 3 (512x512 batch 8 fp32) and 5 (512x512 batch 16 bf16), after the headline's timed region and its
    # own legs; only at the headline invocation (one GPU, 256x256 batch 32 fp32, eager)
    secondary = None
    if (rank == 0 and world == 1 and not args.no_secondary and not args.force_dist and not args.graph and (S, B, args.dtype) == (256, 32, "f32")
            and args.low_latency == "auto"):
        secondary = [secondary_config(dev_index, 512, 8, "f32"), secondary_config(dev_index, 512, 16, "bf16")]
        if not args.no_e2e:
            # the end-to-end rate of test.py's loop at the headline shape, files to files, beside the forward-only rate
            e2e = e2e_child(S, B)
            e2e["forward_only_images_per_sec"] = round(world * B * args.steps / elapsed, 1)
            e2e["e2e_images_per_sec"] = e2e.get("value")
            secondary.append(e2e)

    if rank == 0:
        images = world * B * args.steps
        ll_on = eng.is_low_latency(B, S, S, low_latency)
        line = {
            "metric": "images/sec", "value": images / elapsed, "unit": "images/sec", "n_gpus": world,
            "steps": args.steps, "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": args.dtype, "data": "synthetic",
            "config": {"workload": "SketchEdit inference forward (netM+netG, use_cam, pool max) %dx%d batch %d per GPU, "
                                   "procedural weights" % (S, S, B),
                       "global_batch": world * B, "size": S, "per_gpu_batch": B,
                       "execution": ("low-latency" if ll_on else "default") + ("+graph" if args.graph else "") +
                                    ("" if ll_on else (", two streams" if _lib_opt("SE_FORK_DEFAULT") else ", one stream")),
                       "host_placement_rank0": placement,
                       "collective": ({"what": "one all_gather of the packed (B,4,H,W) outputs per step" +
                                               (" on a side stream, under the next step's forward" if overlap else ""),
                                       "backend": args.backend,
                                       "nccl_version": (".".join(str(v) for v in torch.cuda.nccl.version()) if args.backend == "nccl" else None),
                                       "NCCL_MAX_NCHANNELS": os.environ.get("NCCL_MAX_NCHANNELS"),
                                       "bytes_per_rank_per_step": B * 4 * S * S * 4,
                                       "per_rank_ms_per_step": per_rank_ms,
                                       "per_rank_forward_only_ms": fwd_only_ms,
                                       # what the side-stream overlap did not hide (slowest rank): ms/step with the gather
                                       # minus the forward alone, un-timed leg of min(steps, 10) forwards
                                       "exposed_gather_ms": round(max(per_rank_ms) - max(fwd_only_ms), 4)}
                                      if use_dist else None)},
            "roofline": roofline, "cpu_baseline": cpu, "parity": parity, "secondary": secondary, "kernels": kernels,
            "layers": ({r["layer"]: {"ms": round(r["total_ms"] / nprof, 4), "n": r["launches"] // nprof,
                                     "tflops_executed": round(r["flops_executed"] / (r["total_ms"] * 1e-3) / 1e12, 1)}
                        for r in full_rep["layers"]} if args.layers else None),
            "forward_tflops_live": (LIVE_GFLOP_PER_IMAGE.get(S, 0) * images / elapsed / 1e3) or None,
            # reference-defined FLOPs of the live forward / wall time, over the MFMA peak of the dtype: NOT a utilisation (above 1
            # where Winograd / sub-pixel forms execute fewer multiply-adds than the reference defines; the hardware fraction is
            # roofline.forward_executed_frac); the figure to watch at batch 1, where every layer runs in its direct form
            "forward_algorithmic_tflops_over_peak": (round(LIVE_GFLOP_PER_IMAGE[S] * images / elapsed / 1e3 / peak / world, 4)
                                         if S in LIVE_GFLOP_PER_IMAGE else None),
        }
        for sec in secondary or []:
            violations += sec.pop("rate_violations", []) if "rate_violations" in sec else []
        # every printed executed rate must be physically possible: none above peak x 1.08; otherwise the line says so and
        # the run FAILS (exit code 4) -- a number above the roofline is a bookkeeping bug to fix, not to publish
        line["rate_check"] = {"ok": not violations, "rule": "executed_tflops <= MFMA peak x %.2f for every label" % PEAK_HEADROOM, "violations": violations}
        os.write(real_stdout, (json.dumps(line) + "\n").encode())
        if violations:
            print("bench.py: executed rate above the MFMA peak: %r" % (violations,), file=sys.stderr)
            exit_code = 4
    if use_dist:
        dist.barrier()
        dist.destroy_process_group()
    if exit_code:
        raise SystemExit(exit_code)


if __name__ == "__main__":
    main()
