"""CPU-side tests: the C-ABI library loads and exports every declared symbol, the host-side mirror of
the reference interface (options, registries, state_dict contract, dataset), loud failure without a GPU,
and the batch-sharding path with world_size 2 over gloo."""
import ctypes
import os
import re
import subprocess
import sys

import numpy as np
import pytest
import torch

from sketchedit_amd import _lib, synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

CELEB_ARGV = ("--batchSize 1 --nThreads 1 --name celeb --joint_train_inp --dataset_mode testimage "
              "--image_dirs {d}/images --mask_dirs {d}/edges --image_lists {d}/list.txt --image_postfix .png "
              "--mask_postfix .png --model editline2 --netG deepfillc2 --pool_type max --use_cam "
              "--which_epoch latest --output_dir {d}/results")


@pytest.fixture(scope="module")
def built_lib():
    _lib.build_library()
    return ctypes.CDLL(_lib.LIB_PATH)


def test_header_symbols_exported(built_lib):
    hdr = open(os.path.join(ROOT, "include", "sketchedit_hip.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    declared = set(re.findall(r"\b(se_[a-z_A-Z0-9]+)\s*\(", hdr))
    assert declared == set(_lib.SYMBOLS), declared ^ set(_lib.SYMBOLS)
    for s in declared:
        assert getattr(built_lib, s) is not None
    built_lib.se_version.restype = ctypes.c_char_p
    assert b"gfx950" in built_lib.se_version()


def test_library_contains_gfx950_code():
    assert b"gfx950" in open(_lib.LIB_PATH, "rb").read()


def test_no_gpu_fails_loudly():
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    with pytest.raises(_lib.SketchEditHipError):
        _lib.Engine(0)


def _opt(tmp_path):
    from sketchedit_amd.options.test_options import TestOptions
    return TestOptions().parse(CELEB_ARGV.format(d=tmp_path).split(), quiet=True)


def test_options_celeb_command_line(tmp_path):
    o = _opt(tmp_path)
    assert (o.model, o.netG, o.pool_type, o.use_cam, o.joint_train_inp) == ("editline2", "deepfillc2", "max", True, True)
    assert o.gpu_ids == [0] and o.isTrain is False and o.which_epoch == "latest" and o.how_many == float("inf")
    assert o.no_mask_cc is False and o.no_mask_coarse is False and o.image_postfix == ".png"
    assert _lib.flags_from_opt(o) == (_lib.FLAG_USE_CAM | _lib.FLAG_POOL_MAX | _lib.FLAG_JOINT_TRAIN_INP)
    o.pool_type = "bogus"
    with pytest.raises(NotImplementedError):
        _lib.flags_from_opt(o)


def test_options_demo_command_line_default_dataset_mode():
    """demo.py:19 parses without --dataset_mode, i.e. with its default 'base' -> data/base_dataset.py BaseDataset."""
    from sketchedit_amd import data
    from sketchedit_amd.data.base_dataset import BaseDataset
    from sketchedit_amd.data.testimage_dataset import TestImageDataset
    from sketchedit_amd.options.test_options import TestOptions
    o = TestOptions().parse("--name celeb --joint_train_inp --model editline2 --netG deepfillc2 --pool_type max "
                            "--use_cam --which_epoch latest".split(), quiet=True)
    assert o.dataset_mode == "base" and o.model == "editline2"
    assert data.find_dataset_using_name("base") is BaseDataset and issubclass(TestImageDataset, BaseDataset)
    assert len(BaseDataset()) == 0


def test_registries_and_state_dict_contract(tmp_path):
    from sketchedit_amd import models
    from sketchedit_amd.models import networks
    o = _opt(tmp_path)
    assert models.find_model_using_name("editline2").__name__ == "EditLine2Model"
    G = networks.find_network_using_name("deepfillc2", "generator")
    M = networks.find_network_using_name("MD", "generator")
    assert (G.__name__, M.__name__) == ("DeepFillC2Generator", "MDGenerator")
    g, m = G(o), M(o)
    # checkpoint contract of the reference: 104 / 48 tensors, 5 366 430 / 2 112 820 parameters
    assert len(g.state_dict()) == 104 and sum(p.numel() for p in g.parameters()) == 5366430
    assert len(m.state_dict()) == 48 and sum(p.numel() for p in m.parameters()) == 2112820
    for net, mod in (("G", g), ("M", m)):
        sd = synth.make_state_dict(net, 0)
        assert set(sd) == set(mod.state_dict())
        for k, v in mod.state_dict().items():
            assert tuple(v.shape) == sd[k].shape, k
        mod.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()})      # strict
    assert g.conv10_atrous.dilation == (16, 16) and g.conv10_atrous.padding == (16, 16)
    assert g.conv2_downsample.stride == (2, 2) and g.pmconv6.act == "relu" and g.conv17.act is None
    assert g.conv13_upsample_conv.upsample and m.conv_mask_17.out_channels == 1
    with pytest.raises(_lib.SketchEditHipError):        # CPU tensors: no fallback
        m(torch.zeros(1, 3, 64, 64), torch.zeros(1, 1, 64, 64))


def test_checkpoint_roundtrip_module_prefix(tmp_path):
    from sketchedit_amd.models.networks.generator import MDGenerator
    from sketchedit_amd.util import util
    o = _opt(tmp_path)
    o.checkpoints_dir = str(tmp_path / "ckpt")
    m = MDGenerator(o)
    sd = {("module." + k): torch.from_numpy(v) for k, v in synth.make_state_dict("M", 3).items()}
    os.makedirs(os.path.join(o.checkpoints_dir, o.name))
    torch.save(sd, util.checkpoint_path("M", "latest", o))
    util.load_network(m, "M", "latest", o)
    assert torch.equal(m.conv1.weight.detach(), sd["module.conv1.weight"])


def test_dataset_contract(tmp_path):
    from PIL import Image
    from sketchedit_amd import data
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    rng = np.random.RandomState(0)
    for i, (w, h) in enumerate([(64, 48), (64, 48)]):
        Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / "images" / ("im%d.png" % i))
        e = (rng.rand(h // 2, w // 2) < 0.05).astype(np.uint8) * 255      # half size: must be resized to the image
        Image.fromarray(e).save(tmp_path / "edges" / ("im%d.png" % i))
    (tmp_path / "list.txt").write_text("im0.png\nim1.png\n")
    o = _opt(tmp_path)
    o.nThreads = 0
    dl = data.create_dataloader(o)
    batch = next(iter(dl))
    assert batch["image"].shape == (1, 3, 48, 64) and batch["mask"].shape == (1, 1, 48, 64)
    assert float(batch["image"].min()) >= -1 and float(batch["image"].max()) <= 1
    assert set(np.unique(batch["mask"].numpy())) <= {0.0, 1.0} and batch["path"][0] == "im0.png"
    assert torch.equal(batch["image"], batch["gt"])


def test_synth_shard_consistency():
    full = synth.make_inputs(4, 16, 16, seed=5)
    part = synth.make_inputs(2, 16, 16, seed=5, first_index=2)
    assert np.array_equal(full[0][2:], part[0]) and np.array_equal(full[1][2:], part[1])


def test_shard_range():
    from sketchedit_amd.shard import shard_range
    for n, w in ((32, 8), (10, 4), (3, 2), (256, 8)):
        spans = [shard_range(n, w, r) for r in range(w)]
        assert spans[0][0] == 0 and spans[-1][1] == n
        assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))


_WORKER = r'''
import os, sys
sys.path.insert(0, sys.argv[1])
import torch, torch.distributed as dist
from sketchedit_amd import synth
from sketchedit_amd.shard import sharded_inference
from oracle import sketchedit_oracle as O
dist.init_process_group("gloo", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=int(sys.argv[3]), world_size=2)
torch.set_num_threads(2)
WM = synth.make_state_dict("M", 0)
img, sk = synth.make_inputs(2, 32, 32, seed=21)
img, sk = torch.from_numpy(img), torch.from_numpy(sk)
def fwd(i, s):   # stand-in for the HIP forward on this rank (CPU box): netM of the oracle
    with torch.no_grad():
        m, mi = O.netM_forward(WM, i, s)
    return mi, m
comp, mask = sharded_inference(fwd, img, sk)
if dist.get_rank() == 0:
    full_c, full_m = fwd(img, sk)
    assert comp.shape == full_c.shape and mask.shape == full_m.shape
    assert float((comp - full_c).abs().max()) < 1e-6 and float((mask - full_m).abs().max()) < 1e-6
    print("SHARD_OK")
dist.barrier()
dist.destroy_process_group()
'''


def test_sharded_inference_gloo_world2(tmp_path):
    script = tmp_path / "worker.py"
    script.write_text(_WORKER)
    port = str(29500 + os.getpid() % 2000)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1")
    procs = [subprocess.Popen([sys.executable, str(script), ROOT, port, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(2)]
    outs = [p.communicate(timeout=240)[0] for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n".join(outs)
    assert "SHARD_OK" in outs[0]


@pytest.mark.timeout(60)
def test_batching_server_groups_by_size_and_preserves_results():
    """sketchedit_amd/serve.py (demo.py:39-73 + threaded callers) with a stand-in model: requests of one working size
    share a forward, results equal one-by-one processing, a failing forward reaches every waiting caller."""
    import threading
    import time
    import torch
    from PIL import Image
    from sketchedit_amd import serve

    class Fake:
        def __call__(self, data, mode):
            assert mode == "inference"
            time.sleep(0.01)
            return data["image"] * 0.5, data["mask"]

    rng = np.random.RandomState(3)
    reqs = []
    for (w, h) in [(70, 67), (70, 67), (70, 67), (96, 64), (70, 67)]:
        reqs.append((Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)),
                     Image.fromarray(((rng.rand(h, w) < 0.01) * 255).astype(np.uint8))))
    single = [serve.process_image(Fake(), i, s) for i, s in reqs]
    assert all(o.size == i.size for o, (i, _) in zip(single, reqs))
    srv = serve.BatchingServer(Fake(), max_batch=8, max_wait_s=0.3)
    outs = [None] * len(reqs)

    def worker(k):
        outs[k] = srv.submit(*reqs[k])
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(len(reqs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    srv.close()
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(single, outs))
    assert sorted(srv.batches) == [1, 4]

    class Broken:
        def __call__(self, data, mode):
            raise RuntimeError("boom")
    srv = serve.BatchingServer(Broken(), max_batch=4, max_wait_s=0.01)
    with pytest.raises(RuntimeError, match="boom"):
        srv.submit(*reqs[0])
    srv.close()
    with pytest.raises(RuntimeError, match="closed"):
        srv.submit(*reqs[0])


@pytest.mark.timeout(60)
def test_batching_server_dispatches_over_several_models():
    """BatchingServer(models=[one per GPU]): a shared queue, one worker per model; every request is answered exactly
    once with its own result and, under load, every model gets work (dynamic batching across the GPUs)."""
    import threading
    import time
    from PIL import Image
    from sketchedit_amd import serve

    class Fake:
        def __init__(self, tag):
            self.tag, self.calls, self.lock = tag, 0, threading.Lock()

        def __call__(self, data, mode):
            with self.lock:
                self.calls += 1
            time.sleep(0.03)
            return data["image"] * 0.5, data["mask"]

    rng = np.random.RandomState(7)
    reqs = [(Image.fromarray(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8)),
             Image.fromarray(((rng.rand(64, 64) < 0.01) * 255).astype(np.uint8))) for _ in range(24)]
    want = [serve.process_image(Fake(0), i, s) for i, s in reqs]
    fakes = [Fake(k) for k in range(4)]
    srv = serve.BatchingServer(models=fakes, max_batch=2, max_wait_s=0.001)
    outs = [None] * len(reqs)

    def worker(k):
        outs[k] = srv.submit(*reqs[k])
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(len(reqs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    srv.close()
    assert all(np.array_equal(np.asarray(a), np.asarray(b)) for a, b in zip(want, outs))
    assert sum(srv.batches) == len(reqs)
    assert all(f.calls > 0 for f in fakes), [f.calls for f in fakes]


def test_batching_server_collects_distinct_sizes_concurrently():
    """with requests of two working sizes queued and two workers, both groups are collected at once (one
    collector per size, deadline counted from the request's ARRIVAL): the waits overlap instead of stacking."""
    import threading
    import time
    from PIL import Image
    from sketchedit_amd import serve

    class Fake:
        def __call__(self, data, mode):
            return data["image"], data["mask"]

    wait = 0.4
    srv = serve.BatchingServer(models=[Fake(), Fake(), Fake()], max_batch=8, max_wait_s=wait)
    sizes = [(64, 64), (64, 128), (128, 64)]
    done = {}

    def worker(k):
        t0 = time.monotonic()
        srv.submit(Image.new("RGB", sizes[k]), Image.new("L", sizes[k]))
        done[k] = time.monotonic() - t0
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(3)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    srv.close()
    assert sorted(srv.batches) == [1, 1, 1]
    # every lone request waits out ITS deadline once (~0.4 s); stacked collection would make the last one wait ~1.2 s
    assert max(done.values()) < 2 * wait, done
    assert min(done.values()) >= 0.9 * wait, done


def test_rank_cpu_slices_for_multi_gpu_launch():
    """bench.py's per-rank host placement: disjoint CPU slices that cover whole cores, on the GPU's NUMA
    node when sysfs names it; never empty."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("se_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.parse_cpulist("0-3,8-9,12\n") == [0, 1, 2, 3, 8, 9, 12]
    allowed = set(range(256))
    sl = [bench.rank_cpu_slice(r, 8, allowed) for r in range(8)]
    assert all(len(x) == 32 for x in sl) and len(set().union(*map(set, sl))) == 256      # disjoint, complete
    # a 2 x 64-core host with SMT: logical CPUs c and c + 128 share a core; GPUs 0-3 on node 0
    key = lambda c: (c // 64 % 2, c % 64, c)         # noqa: E731
    node0 = bench.parse_cpulist("0-63,128-191")
    s2 = bench.rank_cpu_slice(2, 8, allowed, node0, [0, 1, 2, 3], key=key)
    assert len(s2) == 32 and set(s2) <= set(node0) and all((c + 128) % 256 in s2 or (c - 128) in s2 for c in s2)   # whole cores
    s0 = bench.rank_cpu_slice(0, 8, allowed, node0, [0, 1, 2, 3], key=key)
    assert not set(s0) & set(s2)
    assert bench.rank_cpu_slice(0, 1, {3, 4}) == [3, 4]
    assert bench.rank_cpu_slice(5, 8, {0, 1}) == [0, 1]                                   # fewer CPUs than ranks: everything


def test_execution_mode_is_pinned_where_batch_composition_varies():
    """the library's results are bit-identical across batch positions only within one execution mode,
    so the callers whose batch size varies pin the mode: BatchingServer from (max_batch, working size) -- never from the
    group's actual size -- and sharded_inference from the GLOBAL batch, not from the shard."""
    import torch
    from PIL import Image
    from sketchedit_amd import serve, shard

    class Rec:
        def __init__(self):
            self.modes = []

        def forward(self, data, mode, low_latency=None):
            self.modes.append((data["image"].shape[0], low_latency))
            return data["image"], data["mask"]
        __call__ = forward

    img, sk = Image.new("RGB", (256, 256)), Image.new("L", (256, 256))
    for max_batch, want in ((2, True), (32, False)):            # 2 x 256x256 is a small call, 32 x 256x256 is not
        m = Rec()
        srv = serve.BatchingServer(m, max_batch=max_batch, max_wait_s=0.0)
        srv.submit(img, sk)                                      # a lone request: group of 1 whatever max_batch is
        srv.close()
        assert m.modes == [(1, want)], m.modes
    m = Rec()
    srv = serve.BatchingServer(m, max_batch=32, max_wait_s=0.0, mode_policy="by_size")
    srv.submit(img, sk)
    srv.close()
    assert m.modes == [(1, None)]
    with pytest.raises(ValueError):
        serve.BatchingServer(m, mode_policy="fastest")
    # shard: the forward sees the mode of the global batch (8 x 256x256 -> default), not of its shard
    seen = []

    def fwd(i, s, low_latency=None):
        seen.append(low_latency)
        return torch.cat([i, s], 1)
    shard.sharded_inference(fwd, torch.zeros(8, 3, 256, 256), torch.zeros(8, 1, 256, 256))
    shard.sharded_inference(fwd, torch.zeros(2, 3, 256, 256), torch.zeros(2, 1, 256, 256))
    assert seen == [False, True]
    # round 6: up to three 256x256 images' worth of pixels, or ONE image of up to 512x512 (tools/ll_threshold.sh)
    assert shard.global_mode(8, 256, 256) is False and shard.global_mode(1, 512, 512) is True
    assert shard.global_mode(3, 256, 256) is True and shard.global_mode(4, 256, 256) is False and shard.global_mode(2, 512, 512) is False
    # EditLine2Model: a call's own size picks the mode unless pinned (an interactive caller with --batchSize 8 that sends one
    # image keeps the low-latency kernels); test.py pins the mode of a FULL --batchSize batch for its whole file
    # list (batch_mode), so a ragged last batch does not change kernels
    from types import SimpleNamespace
    from sketchedit_amd.models.editline2_model import EditLine2Model
    mf, bm = EditLine2Model._mode_for, EditLine2Model.batch_mode
    m8, m1 = SimpleNamespace(opt=SimpleNamespace(batchSize=8)), SimpleNamespace(opt=SimpleNamespace(batchSize=1))
    assert mf(m8, 1, 256, 256, None) is True and mf(m8, 8, 256, 256, None) is False               # by the call's own size
    assert mf(m1, 32, 256, 256, None) is False                                                    # a caller's own big batch
    assert mf(m8, 4, 256, 256, True) is True and mf(m1, 1, 256, 256, False) is False              # pinned wins
    assert bm(m8, 256, 256) is False and mf(m8, 4, 256, 256, bm(m8, 256, 256)) is False           # test.py: 8, 8, 4 all default mode
    assert bm(m1, 256, 256) is True                                                               # test_celeb.sh


def test_check_checkpoint_missing_key_not_hidden_by_suffix_match(tmp_path):
    """A problem line that merely ENDS with a key name must not hide that the key itself is missing."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_check_ckpt2", os.path.join(root, "tools", "check_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    M = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
    M["wconv1.weight"] = M.pop("conv1.weight")          # 'unexpected key wconv1.weight' ends with 'conv1.weight'
    M["module.xconv1.bias"] = M.pop("conv1.bias")
    _, problems = mod.check_state_dict("M", M)
    text = "\n".join(problems)
    assert "missing key conv1.weight" in text and "missing key conv1.bias" in text
    assert "unexpected key wconv1.weight" in text


def test_check_checkpoint_tool(tmp_path):
    """tools/check_checkpoint.py: a DataParallel-prefixed pair validates (104 / 48 tensors), a wrong shape, a missing
    and an unexpected key are reported (util/util.py:214-225 contract, without a GPU)."""
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_check_ckpt", os.path.join(root, "tools", "check_checkpoint.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    d = tmp_path / "celeb"
    os.makedirs(d)
    G = {"module." + k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}
    M = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
    torch.save(G, d / "latest_net_G.pth")
    torch.save(M, d / "latest_net_M.pth")
    assert mod.main([str(d), "--npz", str(tmp_path / "w")]) == 0
    z = np.load(tmp_path / "w_G.npz")
    assert len(z.files) == 104 and sum(z[k].size for k in z.files) == 5366430
    assert len(np.load(tmp_path / "w_M.npz").files) == 48
    clean, problems = mod.check_state_dict("G", G)
    assert not problems and len(clean) == 104
    bad = dict(M)
    bad["conv3.weight"] = bad["conv3.weight"][:, :, :2]
    del bad["conv5.bias"]
    bad["extra.weight"] = torch.zeros(1)
    _, problems = mod.check_state_dict("M", bad)
    text = "\n".join(problems)
    assert "size mismatch for conv3.weight" in text and "missing key conv5.bias" in text and "unexpected key extra.weight" in text
    torch.save(bad, d / "latest_net_M.pth")
    assert mod.main([str(d)]) == 1


def test_symmetric_score_tile_enumeration():
    """The tile order of the symmetric E GEMM (se_attention.hip, att2_pair_kernel with p.sym): a 1-D index t enumerates, panel
    by panel (16 key tiles), the tiles with query tile bx <= key tile by / 4.  This is the device code's integer arithmetic
    restated: every computed tile exactly once, nothing left of the diagonal, 32 np^2 + 8 np indices for np panels."""
    import math

    def decode(t):
        pn = int((math.sqrt(64.0 + 128.0 * t) - 8.0) / 64.0)
        while 32 * (pn + 1) * (pn + 1) + 8 * (pn + 1) <= t:
            pn += 1
        while 32 * pn * pn + 8 * pn > t:
            pn -= 1
        u = t - (32 * pn * pn + 8 * pn)
        if u < 64 * pn:
            return u >> 4, 16 * pn + (u & 15)
        v = u - 64 * pn
        j = 0 if v < 16 else 1 if v < 28 else 2 if v < 36 else 3
        return 4 * pn + j, 16 * pn + 4 * j + (v - (0, 16, 28, 36)[j])

    for ny in (1, 2, 4, 15, 16, 17, 18, 64, 71, 100, 256):      # key tiles of 64 (or 32) positions: R = 4096 -> 64
        npan = (ny + 15) // 16
        seen = set()
        for t in range(32 * npan * npan + 8 * npan):
            bx, by = decode(t)
            assert 0 <= bx <= by // 4
            if by < ny:                                          # (the last panel is enumerated in full; the kernel returns)
                assert (bx, by) not in seen
                seen.add((bx, by))
        assert seen == {(bx, by) for by in range(ny) for bx in range(by // 4 + 1)}


def test_bench_all_cores_cpu_sample_is_bounded():
    """bench.py's cpu_baseline carries an all-cores sample beside the 32-thread figure (SURVEY.md 8d asked for all host cores;
    oversubscribed oneDNN collapses on the 256-thread GPU hosts, so the leg is a child process stopped after a budget):
    whatever it reports, it reports within the budget, with the thread count stated."""
    import importlib.util
    import time
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_bench_mod", os.path.join(root, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    t0 = time.perf_counter()
    r = bench.cpu_all_cores_sample(os.cpu_count() or 1, budget_s=20.0)
    assert time.perf_counter() - t0 < 40.0
    assert r["threads"] == (os.cpu_count() or 1) and "note" in r
    assert r["images_per_sec"] is None or r["images_per_sec"] > 0.0


def test_developer_switches_are_a_table_read_once():
    """The library's kernel-form switches: one process-wide table filled from the environment ONCE (no getenv on a launch
    path), changed afterwards only through se_debug_set_option.  No device needed: the table lives on the host."""
    import subprocess
    import sys
    # built-in defaults: asserted in a child whose environment carries no SE_* variable (in-process asserts failed
    # whenever an A/B variable was exported)
    scrub = {k: v for k, v in os.environ.items() if not k.startswith("SE_")}
    code0 = ("from sketchedit_amd import _lib; print(_lib.get_option('SE_WINOGRAD_F43'), _lib.get_option('WINOGRAD_F43'), "
             "_lib.get_option('SE_ATT_FUSED'), _lib.get_option('SE_RTILE_WX'))")
    out0 = subprocess.run([sys.executable, "-c", code0], cwd=ROOT, env=scrub, capture_output=True, text=True, timeout=300)
    assert out0.returncode == 0, out0.stderr[-2000:]
    assert out0.stdout.split() == ["1", "1", "-1", "2"]                        # with or without the prefix
    base_f43 = _lib.get_option("SE_WINOGRAD_F43")
    saved_env = os.environ.get("SE_WINOGRAD_F43")
    try:                                                  # every mutation inside try / finally: nothing leaks into later tests
        _lib.set_option("SE_WINOGRAD_F43", 2)
        _lib.set_option("SE_TEST_OFFSET_LIMIT", 12345)
        assert _lib.get_option("SE_WINOGRAD_F43") == 2 and _lib.get_option("SE_TEST_OFFSET_LIMIT") == 12345
        os.environ["SE_WINOGRAD_F43"] = "0"              # the environment is not consulted again
        assert _lib.get_option("SE_WINOGRAD_F43") == 2
        _lib.reset_options()
        assert _lib.get_option("SE_WINOGRAD_F43") == base_f43 and _lib.get_option("SE_TEST_OFFSET_LIMIT") == 0
    finally:
        _lib.reset_options()
        if saved_env is None:
            os.environ.pop("SE_WINOGRAD_F43", None)
        else:
            os.environ["SE_WINOGRAD_F43"] = saved_env
    with pytest.raises(_lib.SketchEditHipError):
        _lib.set_option("SE_NO_SUCH_SWITCH", 1)
    with pytest.raises(_lib.SketchEditHipError):
        _lib.get_option("SE_WINOGRAD_F43_SKIP")           # removed in round 5
    # a fresh process takes its initial values from the environment -- except the test aid, which only the call can set
    code = ("from sketchedit_amd import _lib; print(_lib.get_option('SE_WINOGRAD_F43'), _lib.get_option('SE_ATT_E16'), "
            "_lib.get_option('SE_TEST_OFFSET_LIMIT'))")
    env = dict(os.environ, SE_WINOGRAD_F43="2", SE_ATT_E16="0", SE_TEST_OFFSET_LIMIT="777")
    out = subprocess.run([sys.executable, "-c", code], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.split() == ["2", "0", "0"]


def test_dataset_u8_mode_carries_the_decoders_arrays(tmp_path):
    """u8 mode of the dataset (test.py's pipelined loop): the uint8 arrays the float path normalises -- (v/255 - 0.5)/0.5 and
    sketch > 0 (/root/reference/data/testimage_dataset.py:89-111) applied to them gives the float path's tensors exactly."""
    from PIL import Image
    from sketchedit_amd import data
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    rng = np.random.RandomState(0)
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (48, 64, 3), dtype=np.uint8)).save(tmp_path / "images" / ("im%d.png" % i))
        Image.fromarray((rng.rand(24, 32) < 0.05).astype(np.uint8) * 255).save(tmp_path / "edges" / ("im%d.png" % i))
    (tmp_path / "list.txt").write_text("im0.png\nim1.png\nim2.png\n")
    o = _opt(tmp_path)
    o.nThreads, o.batchSize = 0, 2
    f = next(iter(data.create_dataloader(o)))
    o.u8_io = True
    u = next(iter(data.create_dataloader(o)))
    assert u["image_u8"].dtype == torch.uint8 and tuple(u["image_u8"].shape) == (2, 48, 64, 3) and tuple(u["mask_u8"].shape) == (2, 48, 64)
    assert "image" not in u and u["path"] == f["path"]
    want = (u["image_u8"].numpy().astype(np.float32).transpose(0, 3, 1, 2) / 255.0 - 0.5) / 0.5
    assert np.array_equal(want, f["image"].numpy())
    assert np.array_equal((u["mask_u8"].numpy() > 0).astype(np.float32)[:, None], f["mask"].numpy())


def test_host_codec_capability_and_synthetic_pairs(tmp_path):
    """the host-only yardstick of bench.py --e2e (decode processes + encoder threads, no GPU) on a few tiny pairs"""
    import importlib.util
    from sketchedit_amd.pipeline import host_codec_capability
    spec = importlib.util.spec_from_file_location("se_bench_mod", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    for i in range(3):
        bench._write_pair((str(tmp_path), i, 64))
    ip = [str(tmp_path / "images" / ("u%04d.png" % i)) for i in range(3)]
    mp_ = [str(tmp_path / "edges" / ("u%04d.png" % i)) for i in range(3)]
    os.makedirs(tmp_path / "cap")
    for writer in ("pil", "fast"):
        cap = host_codec_capability(ip, mp_, 2, 2, seconds=0.5, writer=writer, out_dir=str(tmp_path / "cap"))
        assert cap["decode_ips"] > 0 and cap["encode_ips"] > 0 and 0 < cap["both_ips"] <= max(cap["both_decode_ips"], cap["both_encode_ips"])


def test_png_writers_and_encoder_processes(tmp_path):
    """The two PNG writers of the I/O path (sketchedit_amd/png_worker.py): `pil` = Image.save's defaults, byte for byte; `fast`
    = the reference's cv2.imwrite settings (SUB filter, zlib 1, RLE; /root/reference/test.py:37) written directly -- a valid PNG
    with the same pixels.  And the encoder PROCESS protocol of the pipeline: jobs name a slot of a ring file, results land on
    disk, a failing job raises in the caller."""
    import io
    from PIL import Image
    from sketchedit_amd.png_worker import png_bytes_fast, save_png
    from sketchedit_amd.pipeline import _EncoderProcs
    rng = np.random.RandomState(5)
    for shape in ((40, 72, 3), (17, 5, 3), (64, 64), (1, 1, 3)):
        a = rng.randint(0, 256, shape).astype(np.uint8)
        back = np.asarray(Image.open(io.BytesIO(png_bytes_fast(a))))
        assert back.shape == a.shape and np.array_equal(back, a), shape
    a = rng.randint(0, 256, (32, 48, 3)).astype(np.uint8)
    save_png(a, str(tmp_path / "p.png"))
    buf = io.BytesIO()
    Image.fromarray(a).save(buf, format="PNG")
    assert (tmp_path / "p.png").read_bytes() == buf.getvalue()
    # ring of 3 slots x 4 images; two worker processes
    ring = np.memmap(str(tmp_path / "ring"), dtype=np.uint8, mode="w+", shape=(3, 4, 32, 48, 3))
    ring[:] = rng.randint(0, 256, ring.shape).astype(np.uint8)
    mring = np.memmap(str(tmp_path / "mring"), dtype=np.uint8, mode="w+", shape=(3, 4, 32, 48))
    mring[:] = rng.randint(0, 256, mring.shape).astype(np.uint8)
    ring.flush(); mring.flush()
    os.makedirs(tmp_path / "o"); os.makedirs(tmp_path / "m")
    procs = _EncoderProcs(2)
    try:
        job = dict(rgb_ring=str(tmp_path / "ring"), rgb_shape=ring.shape, mask_ring=str(tmp_path / "mring"), mask_shape=mring.shape,
                   slot=1, out_dir=str(tmp_path / "o"), mask_dir=str(tmp_path / "m"))
        futs = [procs.submit(dict(job, first=0, paths=["a.png", "b.png"], writer="pil")),
                procs.submit(dict(job, first=2, paths=["c.png", "d.png"], writer="fast"))]
        assert all(f.result(timeout=60) > 0 for f in futs)
        for i, n in enumerate("abcd"):
            assert np.array_equal(np.asarray(Image.open(tmp_path / "o" / (n + ".png"))), ring[1, i])
            assert np.array_equal(np.asarray(Image.open(tmp_path / "m" / (n + ".png"))), mring[1, i])
        buf = io.BytesIO()
        Image.fromarray(np.asarray(ring[1, 0])).save(buf, format="PNG")
        assert (tmp_path / "o" / "a.png").read_bytes() == buf.getvalue()
        bad = procs.submit(dict(job, first=0, paths=["x.png"], out_dir=str(tmp_path / "no_such_dir"), writer="pil"))
        with pytest.raises(RuntimeError):
            bad.result(timeout=60)
        # decode jobs: pairs straight into slot 2 of the (writable) rings, as the dataset decodes them; a size mismatch is an error
        Image.fromarray(np.asarray(ring[0, 3])).save(tmp_path / "in.png")
        Image.fromarray(np.asarray(mring[0, 3])[::2, ::2].copy()).save(tmp_path / "in_m.png")          # half size: resized to the image
        dj = dict(kind="decode", img_ring=str(tmp_path / "ring"), img_shape=ring.shape, sk_ring=str(tmp_path / "mring"), sk_shape=mring.shape,
                  slot=2, first=1, image_paths=[str(tmp_path / "in.png")], mask_paths=[str(tmp_path / "in_m.png")])
        assert procs.submit(dj).result(timeout=60) > 0
        chk = np.memmap(str(tmp_path / "ring"), dtype=np.uint8, mode="r", shape=ring.shape)
        assert np.array_equal(chk[2, 1], ring[0, 3])
        want = np.asarray(Image.open(tmp_path / "in_m.png").convert("L").resize((48, 32)), dtype=np.uint8)
        assert np.array_equal(np.memmap(str(tmp_path / "mring"), dtype=np.uint8, mode="r", shape=mring.shape)[2, 1], want)
        Image.fromarray(np.zeros((16, 16, 3), np.uint8)).save(tmp_path / "small.png")
        with pytest.raises(RuntimeError):
            procs.submit(dict(dj, image_paths=[str(tmp_path / "small.png")])).result(timeout=60)
    finally:
        procs.close()


def test_worker_pool_reader_loses_no_completion_line(tmp_path):
    """Thousands of tiny jobs on a few workers: completion lines arrive back to back in one pipe read.  A buffered readline()
    behind a selector swallowed the second line of such a read and the caller waited for ever (found in round 6 as a hang of
    the GPU suite); the reader now splits raw pipe reads itself."""
    from sketchedit_amd.pipeline import _EncoderProcs
    ring = np.memmap(str(tmp_path / "ring"), dtype=np.uint8, mode="w+", shape=(1, 4, 8, 8, 3))
    ring[:] = 7
    ring.flush()
    os.makedirs(tmp_path / "o")
    procs = _EncoderProcs(3)
    try:
        job = dict(rgb_ring=str(tmp_path / "ring"), rgb_shape=ring.shape, mask_ring=None, mask_shape=None, slot=0, out_dir=str(tmp_path / "o"),
                   mask_dir=None, writer="fast")
        futs = [procs.submit(dict(job, first=i % 4, paths=["x%d.png" % (i % 50)])) for i in range(2000)]
        assert all(f.result(timeout=120) >= 0 for f in futs)
    finally:
        procs.close()


def test_effective_cpus_honours_a_cgroup_quota(monkeypatch, tmp_path):
    from sketchedit_amd import hostinfo
    n = hostinfo.effective_cpus()
    assert 1 <= n <= (os.cpu_count() or 1)
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: 2.0)
    assert hostinfo.effective_cpus() == min(2, len(os.sched_getaffinity(0)))
    monkeypatch.setattr(hostinfo, "cgroup_cpu_quota", lambda: 0.5)
    assert hostinfo.effective_cpus() == 1


def test_bench_dry_run_topology_needs_no_gpu():
    """`bench.py --gpus 8 --dry-run-topology`: the per-rank placement and environment a real 8-GPU run
    would use, from sysfs alone -- runs here, where there is no GPU: contiguous CPU slices, every rank's shard of the global
    batch, the one collective's byte counts."""
    import json
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--dry-run-topology", "--nccl-channels", "4"],
                         cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads(out.stdout.strip().splitlines()[-1])
    assert d["dry_run_topology"] and d["n_gpus"] == 8 and len(d["ranks"]) == 8
    assert [r["hip_device"] for r in d["ranks"]] == list(range(8))
    assert d["ranks"][7]["shard"] == "images [224, 256) of the global batch 256"
    assert d["collective"]["bytes_per_rank_per_step"] == 32 * 4 * 256 * 256 * 4
    assert d["environment"]["NCCL_MAX_NCHANNELS_would_be"] == "4"
    assert all(r["threads"] >= 1 for r in d["ranks"]) and d["host"]["cpus_effective"] >= 1
