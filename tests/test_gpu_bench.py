"""bench.py as the driver runs it: the one-line JSON contract, and the N > 1 code path (one rank per process, barrier +
max-over-ranks timing, the packed output all-gather on a side stream) with what a 1-GPU box offers -- two ranks that
share cuda:0 and exchange over gloo (RCCL refuses two ranks on one device; the nccl branch itself is covered by
`bench.py --force-dist`, profiles/r02_c2_rccl_world1.json)."""
import json
import os
import socket
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SMALL = ["--steps", "3", "--warmup", "1", "--batch", "2", "--size", "64", "--no-cpu-baseline", "--no-traffic"]


def _one_json_line(out):
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must hold exactly one line, got %d:\n%s" % (len(lines), out[-2000:])
    return json.loads(lines[0])


@pytest.mark.timeout(900)
def test_bench_prints_one_contract_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + SMALL, cwd=ROOT, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert k in d, k
    assert d["metric"] == "images/sec" and d["n_gpus"] == 1 and d["steps"] == 3 and d["warmup"] == 1
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["dtype"] == "f32" and d["vs_baseline"] is None
    assert abs(d["value"] - 2 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    roof = d["roofline"]
    assert roof["bound"] == "mfma" and 0.0 < roof["frac"] <= 1.0 and roof["peak"] == 157.3
    assert roof["frac_algorithmic"] > 0.0                     # SURVEY 8(d)'s definition beside the executed share (may exceed 1)
    assert d["parity"]["max_abs_composed"] < 1e-3 or d["parity"].get("max_abs_composed_same_hard_mask", 1.0) < 1e-3


@pytest.mark.timeout(900)
def test_bench_two_ranks_share_the_gpu_over_gloo():
    # plain `python bench.py --gpus 2`: no external launcher, bench.py starts its own ranks (the form the driver's N=1
    # command takes at N=2/4/8)
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0",
           "--check-gather", "--no-parity"] + SMALL
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["per_gpu_batch"] == 2
    col = d["config"]["collective"]
    assert col and "all_gather" in col["what"] and col["backend"] == "gloo"
    # diagnostics of the first real multi-GPU run: every rank's own ms/step, the forward alone, the exposed gather time
    assert len(col["per_rank_ms_per_step"]) == 2 and len(col["per_rank_forward_only_ms"]) == 2
    assert abs(max(col["per_rank_ms_per_step"]) - d["ms_per_step"]) < 0.05 * d["ms_per_step"] + 0.5
    assert col["bytes_per_rank_per_step"] == d["config"]["per_gpu_batch"] * 4 * d["config"]["size"] ** 2 * 4
    assert abs(d["value"] - 4 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]      # whole-job images per second


@pytest.mark.timeout(900)
def test_bench_two_ranks_under_torch_distributed_run():
    """The launcher form of the contract: torch.distributed.run starts the ranks, bench.py reads RANK / WORLD_SIZE."""
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "0",
           "--check-gather", "--no-parity"] + SMALL
    r = subprocess.run(cmd, cwd=ROOT, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4


@pytest.mark.timeout(900)
def test_bench_self_launch_propagates_a_rank_failure():
    """A rank that dies must end the whole job with a non-zero code and no JSON line (here: an impossible device index)."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--device", "63",
           "--no-parity"] + SMALL
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode != 0 and not r.stdout.strip()


@pytest.mark.timeout(900)
def test_bench_eight_ranks_share_the_gpu_over_gloo():
    """The world-8 shape of BASELINE config 4 (256x256 batch 256 over 8 GPUs) with what a 1-GPU box offers: eight rank
    processes started by bench.py itself, all on cuda:0, exchanging over gloo, at 64x64 -- rank bookkeeping (every rank
    generates rows rank*B .. of the global batch), the order of the gathered shards (--check-gather: every rank finds its own
    rows at its own offset, rank 0 recomputes image 0 of the LAST rank's shard bit for bit), the max-over-ranks clock, the
    per-rank CPU slices.  RCCL itself has never run with more than one rank here: no scaling curve."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--device", "0",
           "--check-gather", "--no-parity"] + SMALL
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_json_line(r.stdout)
    assert d["n_gpus"] == 8 and d["config"]["global_batch"] == 16 and d["config"]["per_gpu_batch"] == 2
    assert abs(d["value"] - 16 * 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    pl = d["config"]["host_placement_rank0"]
    assert pl and pl.get("threads", 0) >= 1


@pytest.mark.timeout(900)
def test_bench_eight_ranks_propagate_a_late_rank_failure():
    """World 8, the LAST rank fails (SE_BENCH_FAIL_RANK, a test hook read after the process group is up): the job ends with a
    non-zero code and no JSON line instead of seven ranks waiting in a collective."""
    cmd = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--backend", "gloo", "--device", "0", "--no-parity"] + SMALL
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env["SE_BENCH_FAIL_RANK"] = "7"
    r = subprocess.run(cmd, cwd=ROOT, env=env, capture_output=True, text=True, timeout=850)
    assert r.returncode != 0 and not r.stdout.strip()


@pytest.mark.timeout(600)
def test_bench_e2e_leg_files_to_files():
    """`bench.py --e2e` (the child the default invocation runs for `secondary[2]`): PNG files in, PNG files out through the
    pipelined test.py loop, both writers, every file written, stage rates and the host-codec yardstick on the line; also through
    DataLoader workers.  Small: 48 list entries over 12 unique 64x64 pairs."""
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--e2e", "--size", "64", "--batch", "8", "--e2e-images", "48", "--e2e-unique", "12",
            "--e2e-cap-seconds", "0.4", "--e2e-workers", "2", "--e2e-encoders", "2"]
    for extra in ([], ["--e2e-dataloader"], ["--e2e-encode-threads"]):
        r = subprocess.run(base + extra, cwd=ROOT, capture_output=True, text=True, timeout=500)
        assert r.returncode == 0, r.stderr[-3000:]
        d = _one_json_line(r.stdout)
        assert d["metric"] == "e2e_images_per_sec" and d["value"] > 0 and d["host"]["cpus_effective"] >= 1
        assert set(d["writers"]) == {"pil", "fast"}
        for w in d["writers"].values():
            assert w["images"] == 48 and w["files_written"] == 48 and w["e2e_images_per_sec"] > 0
            assert set(w["stage_images_per_sec"]) == {"decode", "h2d", "forward", "d2h", "encode"}
            assert w["bottleneck"] in w["stage_images_per_sec"] and w["host_codec"]["both_ips"] > 0
