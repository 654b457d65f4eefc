"""GPU tests of the host-side mirror: models.create_model(opt)(data, mode=...) and the per-op modules,
used the way the reference's test.py / demo.py use theirs."""
import os
import shutil

import numpy as np
import pytest
import torch

from sketchedit_amd import synth
from parity_util import composed_for_hard_mask

pytestmark = pytest.mark.gpu

ARGV = ("--batchSize 2 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs x --mask_dirs x "
        "--image_lists x --model editline2 --netG deepfillc2 --pool_type max --use_cam --output_dir {d} --gpu_ids 0")


@pytest.fixture(scope="module")
def model(tmp_path_factory):
    from sketchedit_amd import models
    from sketchedit_amd.options.test_options import TestOptions
    opt = TestOptions().parse(ARGV.format(d=tmp_path_factory.mktemp("out")).split(), quiet=True)
    opt.isSkip = True                      # no checkpoint on disk: weights are injected below
    m = models.create_model(opt)
    m.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
    m.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
    return m.eval()


def _g(golden_dir):
    return dict(np.load(os.path.join(golden_dir, "e2e_64.npz")))


def test_model_inference_matches_reference(model, golden_dir):
    g = _g(golden_dir)
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    data = {"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}      # CPU tensors, as a DataLoader yields
    with torch.no_grad():
        composed, mask = model(data, mode="inference")
    assert composed.is_cuda and composed.shape == (2, 3, 64, 64) and mask.shape == (2, 1, 64, 64)
    assert float((composed.cpu() - torch.from_numpy(g["composed"])).abs().max()) < 1e-3
    assert float((mask.cpu() - torch.from_numpy(g["mask"])).abs().max()) < 1e-3


def test_model_visualize_and_bad_mode(model, golden_dir):
    g = _g(golden_dir)
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    out = model({"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}, mode="visualize")
    assert set(out) == {"mask", "maskim", "coarse", "fine", "composed"}
    assert np.array_equal(out["mask"].cpu().numpy(), g["hard_mask"])
    for k, gk in (("maskim", "mask_image"), ("coarse", "coarse"), ("fine", "fine")):
        assert float((out[k].cpu() - torch.from_numpy(g[gk])).abs().max()) < 1e-3, k
    with pytest.raises(ValueError):
        model({"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}, mode="generator")


def test_networks_called_directly(model, golden_dir):
    g = _g(golden_dir)
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    ci, cs = torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda()
    mask, maskim = model.netM(ci, cs)
    hard = torch.from_numpy(g["hard_mask"]).cuda()
    coarse, fine = model.netG(ci, ci, hard, hard, cs)
    assert float((mask.cpu() - torch.from_numpy(g["mask"])).abs().max()) < 1e-3
    assert float((fine.cpu() - torch.from_numpy(g["fine"])).abs().max()) < 1e-3
    assert float((coarse.cpu() - torch.from_numpy(g["coarse"])).abs().max()) < 1e-3


def test_reload_weights_takes_effect(model):
    img, sk = synth.make_inputs(1, 64, 64, seed=3)
    d = {"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}
    a, _ = model(dict(d), mode="inference")
    model.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 9).items()})
    b, _ = model(dict(d), mode="inference")
    model.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
    c, _ = model(dict(d), mode="inference")
    assert not torch.equal(a, b) and torch.equal(a, c)


def test_op_modules(golden_dir):
    from sketchedit_amd.models.networks.ops import gen_conv, gen_deconv
    g = dict(np.load(os.path.join(golden_dir, "ops.npz")))
    mod = gen_conv(8, 16, 3, 1, 2).cuda()
    mod.load_state_dict({"weight": torch.from_numpy(synth.uniform(7, "c3_s1_d2_elu.w", (16, 8, 3, 3), -0.5, 0.5)),
                         "bias": torch.from_numpy(synth.uniform(7, "c3_s1_d2_elu.b", (16,), -0.5, 0.5))})
    x = torch.from_numpy(synth.uniform(7, "c3_s1_d2_elu.x", (2, 8, 12, 16), -1, 1)).cuda()
    assert float((mod(x).cpu() - torch.from_numpy(g["op.c3_s1_d2_elu"])).abs().max()) < 1e-4
    dec = gen_deconv(8, 16).cuda()
    dec.load_state_dict({"weight": torch.from_numpy(synth.uniform(7, "deconv.w", (16, 8, 3, 3), -0.5, 0.5)),
                         "bias": torch.from_numpy(synth.uniform(7, "deconv.b", (16,), -0.5, 0.5))})
    x = torch.from_numpy(synth.uniform(7, "deconv.x", (2, 8, 6, 8), -1, 1)).cuda()
    assert float((dec(x).cpu() - torch.from_numpy(g["op.deconv"])).abs().max()) < 1e-4


def test_concurrent_callers(model):
    """demo.py runs Flask threaded: several threads call one model object concurrently."""
    import threading
    img, sk = synth.make_inputs(1, 64, 64, seed=8)
    ref, _ = model({"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}, mode="inference")
    ref = ref.cpu()
    errs = []

    def work():
        try:
            for _ in range(3):
                out, _ = model({"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}, mode="inference")
                torch.cuda.synchronize()
                if not torch.equal(out.cpu(), ref):
                    errs.append("mismatch")
        except Exception as e:   # noqa: BLE001
            errs.append(repr(e))

    ts = [threading.Thread(target=work) for _ in range(4)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errs, errs


def test_quantize_u8_matches_test_py_expression(model):
    """se_quantize_u8 == ((x+1)/2*255).astype(uint8) / (mask*255).astype(uint8) of test.py:25-27, bit for bit,
    on real forward outputs and on the edge values -1, 1, 0 and 1-eps."""
    from sketchedit_amd import synth
    from sketchedit_amd._lib import shared_engine
    img, sk = synth.make_inputs(2, 64, 72, seed=77)
    ci, cs = torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda()
    with torch.no_grad():
        comp, mask = model({"image": ci, "mask": cs}, mode="inference")
    comp = comp.clone()
    comp[0, :, 0, :4] = torch.tensor([-1.0, 1.0, 0.0, 0.99999994], device="cuda")
    mask = mask.clone()
    mask[0, 0, 0, :4] = torch.tensor([0.0, 1.0, 0.5, 0.99999994], device="cuda")
    rgb, m8 = shared_engine(0).quantize_u8(comp.contiguous(), mask.contiguous())
    want_rgb = ((comp + 1) / 2 * 255).cpu().numpy().astype(np.uint8).transpose(0, 2, 3, 1)
    want_m = (mask * 255).cpu().numpy().astype(np.uint8)[:, 0]
    assert rgb.dtype == torch.uint8 and tuple(rgb.shape) == (2, 64, 72, 3)
    assert np.array_equal(rgb.cpu().numpy(), want_rgb)
    assert np.array_equal(m8.cpu().numpy(), want_m)


def test_inference_u8_is_the_fused_form_of_quantize_u8(model):
    """se_inference_u8 (quantisation inside the last kernel) == se_inference followed by se_quantize_u8, bit for bit."""
    from sketchedit_amd._lib import shared_engine
    img, sk = synth.make_inputs(2, 64, 72, seed=78)
    d = {"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}
    rgb, m8 = model.inference_u8(dict(d))
    with torch.no_grad():
        comp, mask = model(dict(d), mode="inference")
    rgb2, m82 = shared_engine(0).quantize_u8(comp.contiguous(), mask.contiguous())
    assert rgb.dtype == torch.uint8 and tuple(rgb.shape) == (2, 64, 72, 3) and tuple(m8.shape) == (2, 64, 72)
    assert torch.equal(rgb, rgb2) and torch.equal(m8, m82)


@pytest.mark.timeout(120)
def test_serve_process_image_and_dynamic_batching(model):
    """demo.py:39-73 equivalent: arbitrary request size -> multiples of 8 -> forward -> clamp/uint8 -> resize back;
    concurrent submitters are batched (same results as one-by-one processing: images are independent)."""
    import threading
    from PIL import Image
    from sketchedit_amd import serve
    rng = np.random.RandomState(3)
    reqs = []
    for (w, h) in [(70, 67), (70, 67), (70, 67), (96, 64), (70, 67)]:
        img = Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8))
        sk = Image.fromarray(((rng.rand(h, w) < 0.01) * 255).astype(np.uint8))
        reqs.append((img, sk))
    single = [serve.process_image(model, img, sk) for img, sk in reqs]
    for (img, _), out in zip(reqs, single):
        assert out.size == img.size and out.mode == "RGB"
    # the steps of demo.py written out by hand for request 0
    img, sk = reqs[0]
    x = (torch.from_numpy(np.array(img.resize((64, 64))).transpose(2, 0, 1).astype(np.float32)) / 255 - 0.5) / 0.5
    m = (torch.from_numpy(np.array(sk.resize((64, 64))).astype(np.float32)) > 0).float()
    with torch.no_grad():
        g, _ = model({"image": x[None], "mask": m[None, None]}, mode="inference")
    g = ((torch.clamp(g, -1, 1) + 1) / 2 * 255).cpu().numpy().astype(np.uint8)[0].transpose(1, 2, 0)
    assert np.array_equal(np.asarray(Image.fromarray(g).resize(img.size)), np.asarray(single[0]))
    srv = serve.BatchingServer(model, max_batch=8, max_wait_s=0.2)
    outs = [None] * len(reqs)

    def worker(i):
        outs[i] = srv.submit(*reqs[i])
    ts = [threading.Thread(target=worker, args=(i,)) for i in range(len(reqs))]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    srv.close()
    for a, b in zip(single, outs):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    assert sum(srv.batches) == len(reqs) and max(srv.batches) >= 2     # the four 70x67 requests shared forwards
    with pytest.raises(ValueError):
        serve.process_image(model, Image.new("RGB", (12, 40)), Image.new("L", (12, 40)))


def test_test_py_script_end_to_end(tmp_path, model):
    """The reference's entry point: test.py with a test_celeb.sh-style command line, PNG in -> PNG out."""
    import importlib.util
    from PIL import Image
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    rng = np.random.RandomState(0)
    names = []
    for i in range(3):
        Image.fromarray(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8)).save(tmp_path / "images" / ("f%d.png" % i))
        Image.fromarray(((rng.rand(64, 64) < 0.01) * 255).astype(np.uint8)).save(tmp_path / "edges" / ("f%d.png" % i))
        names.append("f%d.png" % i)
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    argv = ("--batchSize 2 --nThreads 0 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs {d}/images "
            "--mask_dirs {d}/edges --image_lists {d}/list.txt --image_postfix .png --mask_postfix .png --model editline2 "
            "--netG deepfillc2 --pool_type max --use_cam --which_epoch latest --output_dir {d}/results "
            "--output_mask_dir {d}/masks --synthetic_weights").format(d=tmp_path).split()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_test_script", os.path.join(root, "test.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(argv)
    # the same files through the oracle and the reference script's quantisation (test.py:25-27: no clamp, truncation)
    from oracle import sketchedit_oracle as O
    WM, WG = synth.make_state_dict("M", 0), synth.make_state_dict("G", 0)
    for n in names:
        out = np.asarray(Image.open(tmp_path / "results" / n))
        msk = np.asarray(Image.open(tmp_path / "masks" / n))
        assert out.shape == (64, 64, 3) and msk.shape == (64, 64)
        arr = np.asarray(Image.open(tmp_path / "images" / n).convert("RGB"), dtype=np.float32).transpose(2, 0, 1) / 255.0
        x = torch.from_numpy((arr - 0.5) / 0.5)[None]
        e = np.asarray(Image.open(tmp_path / "edges" / n).convert("L"), dtype=np.float32)[None, None] / 255.0
        sk = torch.from_numpy((e > 0).astype(np.float32))
        ref = O.inference(WM, WG, x, sk)
        # the hard mask netG saw in the script's run (same model class, same weights, same input: mode='visualize'); the
        # expected composite is the oracle's for THAT hard mask -- compared always, never skipped
        with torch.no_grad():
            hard = model({"image": x, "mask": sk}, mode="visualize")["mask"]
        comp, _ = composed_for_hard_mask(O, WG, x, sk, ref["mask"], ref["hard_mask"], ref["composed"], hard)
        want = ((comp + 1) / 2 * 255).numpy().astype(np.uint8)[0].transpose(1, 2, 0)
        want_m = (ref["mask"] * 255).numpy().astype(np.uint8)[0, 0]
        # fp32 noise of ~1e-6 can move a value across an integer boundary: at most one grey level, on a few pixels
        dm = np.abs(msk.astype(int) - want_m.astype(int))
        assert dm.max() <= 1 and (dm > 0).mean() < 0.01
        d = np.abs(out.astype(int) - want.astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01


def test_dequantize_u8_is_the_datasets_normalisation_bit_for_bit(model):
    """se_dequantize_u8 (table built on the host, looked up on the device) against the float tensors the dataset builds on the
    CPU (sketchedit_amd/data/testimage_dataset.py = /root/reference/data/testimage_dataset.py:89-111: ToTensor + Normalize(0.5,
    0.5), sketch > 0) for EVERY uint8 value, and se_inference_u8io against se_inference_u8 on those tensors: identical bytes."""
    eng = model.engine()
    rng = np.random.RandomState(3)
    iu8 = rng.randint(0, 256, (2, 64, 64, 3)).astype(np.uint8)
    iu8[0].reshape(-1)[:256] = np.arange(256, dtype=np.uint8)                   # every value occurs
    su8 = (rng.rand(2, 64, 64) < 0.02).astype(np.uint8) * rng.randint(1, 256, (2, 64, 64)).astype(np.uint8)
    img, sk = eng.dequantize_u8(torch.from_numpy(iu8).cuda(), torch.from_numpy(su8).cuda())
    arr = iu8.astype(np.float32).transpose(0, 3, 1, 2) / 255.0
    want = np.ascontiguousarray((arr - 0.5) / 0.5)
    assert want.dtype == np.float32
    assert np.array_equal(img.cpu().numpy(), want)
    assert np.array_equal(sk.cpu().numpy(), (su8.astype(np.float32)[:, None] / 255.0 > 0).astype(np.float32))
    for ll in (False, True):
        a = eng.inference_u8io(torch.from_numpy(iu8).cuda(), torch.from_numpy(su8).cuda(), 1 | 2 | 16, low_latency=ll)
        b = eng.inference_u8(torch.from_numpy(want).cuda(), sk, 1 | 2 | 16, low_latency=ll)
        assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])


def test_pipelined_test_py_writes_the_serial_loops_files(tmp_path):
    """test.py's pipeline (worker processes -> pinned uint8 -> three streams -> encoder threads, sketchedit_amd/pipeline.py)
    against its --serial_io loop (the reference's structure, /root/reference/test.py:20-37): byte-identical PNG files, in
    every directory, for a list that ends in a ragged batch; --how_many stops both at the same file."""
    from PIL import Image
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    rng = np.random.RandomState(1)
    names = ["g%d.png" % i for i in range(7)]
    for n in names:
        Image.fromarray(rng.randint(0, 255, (64, 96, 3), dtype=np.uint8)).save(tmp_path / "images" / n)
        Image.fromarray(((rng.rand(64, 96) < 0.01) * 255).astype(np.uint8)).save(tmp_path / "edges" / n)
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    base = ("--batchSize 3 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs {d}/images "
            "--mask_dirs {d}/edges --image_lists {d}/list.txt --image_postfix .png --mask_postfix .png --model editline2 "
            "--netG deepfillc2 --pool_type max --use_cam --which_epoch latest --synthetic_weights ").format(d=tmp_path)
    _run_test_py((base + "--nThreads 0 --serial_io --output_dir {d}/s --output_mask_dir {d}/sm".format(d=tmp_path)).split())
    _run_test_py((base + "--nThreads 2 --decode_procs 0 --encode_threads 3 --pipeline_depth 2 --output_dir {d}/p --output_mask_dir {d}/pm".format(d=tmp_path)).split())      # DataLoader workers
    _run_test_py((base + "--nThreads 0 --pipeline_depth 1 --output_dir {d}/q".format(d=tmp_path)).split())
    # encoder PROCESSES fed through the shared page-locked ring (hipHostRegister on a /dev/shm file), masks too
    _run_test_py((base + "--nThreads 2 --encode_procs 2 --output_dir {d}/r --output_mask_dir {d}/rm".format(d=tmp_path)).split())
    # the reference's cv2.imwrite settings written directly: other bytes, the same pixels
    _run_test_py((base + "--nThreads 1 --encode_procs 2 --png_writer fast --output_dir {d}/f --output_mask_dir {d}/fm".format(d=tmp_path)).split())
    # the pipeline's own decoder processes (shared page-locked INPUT ring) in front of encoder processes
    _run_test_py((base + "--decode_procs 2 --encode_procs 2 --output_dir {d}/d --output_mask_dir {d}/dm".format(d=tmp_path)).split())
    for n in names:
        ref = (tmp_path / "s" / n).read_bytes()
        assert (tmp_path / "p" / n).read_bytes() == ref and (tmp_path / "q" / n).read_bytes() == ref, n
        assert (tmp_path / "r" / n).read_bytes() == ref and (tmp_path / "d" / n).read_bytes() == ref, n
        assert (tmp_path / "dm" / n).read_bytes() == (tmp_path / "sm" / n).read_bytes(), n
        assert (tmp_path / "pm" / n).read_bytes() == (tmp_path / "sm" / n).read_bytes(), n
        assert (tmp_path / "rm" / n).read_bytes() == (tmp_path / "sm" / n).read_bytes(), n
        assert np.array_equal(np.asarray(Image.open(tmp_path / "f" / n)), np.asarray(Image.open(tmp_path / "s" / n))), n
        assert np.array_equal(np.asarray(Image.open(tmp_path / "fm" / n)), np.asarray(Image.open(tmp_path / "sm" / n))), n
    _run_test_py((base + "--decode_procs 1 --how_many 4 --output_dir {d}/hd".format(d=tmp_path)).split())
    assert sorted(os.listdir(tmp_path / "hd")) == names[:6]
    _run_test_py((base + "--nThreads 1 --how_many 4 --output_dir {d}/h".format(d=tmp_path)).split())
    _run_test_py((base + "--nThreads 0 --serial_io --how_many 4 --output_dir {d}/hs".format(d=tmp_path)).split())
    assert sorted(os.listdir(tmp_path / "h")) == sorted(os.listdir(tmp_path / "hs")) == names[:6]       # test.py:21-22: whole batches


def test_precision_option_reaches_the_bf16_path(tmp_path, model):
    """--precision bf16 (no reference counterpart; BASELINE config 5) through the reference's entry points: the model built from
    the option runs the library's bf16 mode (bit-identical to Engine.set_precision('bf16')), and test.py --precision bf16 writes
    PNGs within bf16 noise of the fp32 run's."""
    from PIL import Image
    from sketchedit_amd import models
    from sketchedit_amd._lib import Engine
    from sketchedit_amd.options.test_options import TestOptions
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    rng = np.random.RandomState(4)
    names = ["b%d.png" % i for i in range(3)]
    for n in names:
        Image.fromarray(rng.randint(0, 255, (64, 64, 3), dtype=np.uint8)).save(tmp_path / "images" / n)
        Image.fromarray(((rng.rand(64, 64) < 0.01) * 255).astype(np.uint8)).save(tmp_path / "edges" / n)
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    base = ("--batchSize 2 --nThreads 0 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs {d}/images "
            "--mask_dirs {d}/edges --image_lists {d}/list.txt --image_postfix .png --mask_postfix .png --model editline2 "
            "--netG deepfillc2 --pool_type max --use_cam --which_epoch latest --synthetic_weights ").format(d=tmp_path)
    _run_test_py((base + "--output_dir {d}/f".format(d=tmp_path)).split())
    _run_test_py((base + "--precision bf16 --output_dir {d}/h".format(d=tmp_path)).split())
    diffs = []
    for n in names:
        a, b = np.asarray(Image.open(tmp_path / "f" / n)).astype(int), np.asarray(Image.open(tmp_path / "h" / n)).astype(int)
        diffs.append(np.abs(a - b))
    d = np.concatenate([x.ravel() for x in diffs])
    assert d.max() > 0 and np.mean(d) < 2.0 and np.mean(d > 16) < 0.02            # another arithmetic, the same picture
    # the option against the engine-level switch, bit for bit
    opt = TestOptions().parse((base + "--precision bf16 --output_dir {d}/x".format(d=tmp_path)).replace("--synthetic_weights ", "").split(), quiet=True)
    opt.isSkip = True
    m = models.create_model(opt)
    m.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
    m.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
    m.eval()
    img, sk = synth.make_inputs(2, 64, 64, seed=7)
    with torch.no_grad():
        comp, mask = m({"image": torch.from_numpy(img), "mask": torch.from_numpy(sk)}, mode="inference")
    e = Engine(0)
    e.load_state_dict("M", synth.make_state_dict("M", 0))
    e.load_state_dict("G", synth.make_state_dict("G", 0))
    e.set_precision("bf16")
    r = e.inference(torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda(), 1 | 2 | 16)
    assert torch.equal(comp, r["composed"]) and torch.equal(mask, r["mask"])
    e.close()


def test_pipeline_mixed_sizes_batch_one(tmp_path):
    """test_places.sh's shape of work: --batchSize 1 over a list whose images have DIFFERENT sizes (512x512 and 408-wide scenes in
    /root/reference/datasets/general_release/list.txt): the pipeline keeps one set of page-locked rings per size; files equal the
    serial loop's byte for byte, through DataLoader workers and through the pipeline's own decoder / encoder processes."""
    from PIL import Image
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    rng = np.random.RandomState(2)
    sizes = [(64, 96), (40, 72), (64, 96), (48, 48), (40, 72)]
    names = ["m%d.png" % i for i in range(len(sizes))]
    for n, (h, w) in zip(names, sizes):
        Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8)).save(tmp_path / "images" / n)
        Image.fromarray(((rng.rand(h, w) < 0.01) * 255).astype(np.uint8)).save(tmp_path / "edges" / n)
    (tmp_path / "list.txt").write_text("\n".join(names) + "\n")
    base = ("--batchSize 1 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs {d}/images "
            "--mask_dirs {d}/edges --image_lists {d}/list.txt --image_postfix .png --mask_postfix .png --model editline2 "
            "--netG deepfillc2 --pool_type max --use_cam --which_epoch latest --synthetic_weights ").format(d=tmp_path)
    _run_test_py((base + "--nThreads 0 --serial_io --output_dir {d}/s".format(d=tmp_path)).split())
    _run_test_py((base + "--nThreads 2 --decode_procs 0 --encode_threads 2 --output_dir {d}/p".format(d=tmp_path)).split())
    _run_test_py((base + "--decode_procs 2 --encode_procs 2 --output_dir {d}/d".format(d=tmp_path)).split())
    for n, (h, w) in zip(names, sizes):
        ref = (tmp_path / "s" / n).read_bytes()
        assert np.asarray(Image.open(tmp_path / "s" / n)).shape == (h, w, 3)
        assert (tmp_path / "p" / n).read_bytes() == ref and (tmp_path / "d" / n).read_bytes() == ref, n
    assert not [f for f in os.listdir("/dev/shm") if f.startswith("se_ring_%d_" % os.getpid())]      # the rings are gone


def test_celeb_sample_through_test_py_matches_the_reference(tmp_path, golden_dir, model):
    """BASELINE config 1: test_celeb.sh's command line (batch 1) on the reference's bundled face + sketch -- written
    back to PNG files from the fixture -- gives the PNGs the reference produces (fixture: tests/golden/make_golden.py)."""
    import importlib.util
    from PIL import Image
    g = dict(np.load(os.path.join(golden_dir, "c1_face.npz")))
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    Image.fromarray(g["image_u8"]).save(tmp_path / "images" / "face.png")
    Image.fromarray(g["sketch_u8"]).save(tmp_path / "edges" / "face.png")
    (tmp_path / "list.txt").write_text("face.png\n")
    argv = ("--batchSize 1 --nThreads 1 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs {d}/images "
            "--mask_dirs {d}/edges --image_lists {d}/list.txt --image_postfix .png --mask_postfix .png --model editline2 "
            "--netG deepfillc2 --pool_type max --use_cam --which_epoch latest --output_dir {d}/results "
            "--output_mask_dir {d}/masks --synthetic_weights").format(d=tmp_path).split()
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_test_script_c1", os.path.join(root, "test.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(argv)
    out = np.asarray(Image.open(tmp_path / "results" / "face.png"))
    msk = np.asarray(Image.open(tmp_path / "masks" / "face.png"))
    dm = np.abs(msk.astype(int) - g["mask_u8"].astype(int))
    assert dm.max() <= 1 and (dm > 0).mean() < 0.01
    hard = np.unpackbits(g["hard_mask_bits"])[:256 * 256].reshape(256, 256)
    assert hard.mean() > 0.1
    # the hard mask of this very input on the GPU (mode='visualize', same weights) must be the REFERENCE's, bit for bit --
    # the fixture is fixed, so this is deterministic -- and then the PNG must be the reference's PNG
    x = torch.from_numpy((g["image_u8"].astype(np.float32).transpose(2, 0, 1) / 255.0 - 0.5) / 0.5)[None]
    sk = torch.from_numpy((g["sketch_u8"].astype(np.float32) / 255.0 > 0).astype(np.float32))[None, None]
    with torch.no_grad():
        vis = model({"image": x, "mask": sk}, mode="visualize")
    assert int((vis["mask"].cpu().numpy()[0, 0] != hard).sum()) == 0, "hard-mask flips on the bundled face"
    d = np.abs(out.astype(int) - g["composed_u8"].astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.01


def _run_test_py(argv):
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_test_script_ckpt", os.path.join(root, "test.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main(argv)


def test_test_celeb_sh_with_checkpoint_files_on_disk(tmp_path, golden_dir):
    """test.py exactly as /root/reference/test_celeb.sh:1-17 runs it -- NO --synthetic_weights: the weights come from
    <checkpoints_dir>/<name>/latest_net_{G,M}.pth through util.load_network (/root/reference/util/util.py:214-225,
    models/editline2_model.py:195-197) -> load_state_dict -> .cuda() -> se_load_weights.  One file is a plain state dict,
    the other carries DataParallel's 'module.' prefix (:221-222).  Inputs: the four bundled faces of
    datasets/face_release/list.txt, written back to PNG files from the fixtures; outputs against the REFERENCE's PNGs."""
    from PIL import Image
    from digest_util import crop_boxes
    ck = tmp_path / "checkpoints" / "celeb"
    os.makedirs(ck)
    torch.save({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}, ck / "latest_net_G.pth")
    torch.save({"module." + k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}, ck / "latest_net_M.pth")
    for sub in ("images", "edges"):
        os.makedirs(tmp_path / sub)
    g0 = dict(np.load(os.path.join(golden_dir, "c1_face.npz")))
    Image.fromarray(g0["image_u8"]).save(tmp_path / "images" / "602.png")
    Image.fromarray(g0["sketch_u8"]).save(tmp_path / "edges" / "602.png")
    others = {sid: dict(np.load(os.path.join(golden_dir, "sample_%s.npz" % sid))) for sid in ("822", "873", "902")}
    for sid, g in others.items():
        (tmp_path / "images" / (sid + ".png")).write_bytes(g["image_png"].tobytes())
        (tmp_path / "edges" / (sid + ".png")).write_bytes(g["sketch_png"].tobytes())
    (tmp_path / "list.txt").write_text("602.png\n822.png\n873.png\n902.png\n")
    base = ("--nThreads 1 --name celeb --joint_train_inp --dataset_mode testimage --image_dirs {d}/images --mask_dirs {d}/edges "
            "--image_lists {d}/list.txt --image_postfix .png --mask_postfix .png --model editline2 --netG deepfillc2 --pool_type max "
            "--use_cam --which_epoch latest --checkpoints_dir {d}/checkpoints").format(d=tmp_path)

    def check(outdir):
        out = np.asarray(Image.open(outdir / "602.png"))
        d = np.abs(out.astype(int) - g0["composed_u8"].astype(int))
        assert d.max() <= 1 and (d > 0).mean() < 0.01
        for sid, g in others.items():
            out = np.asarray(Image.open(outdir / (sid + ".png")))
            assert out.shape == (256, 256, 3)
            for i, (t, l) in enumerate(crop_boxes(256, 256)):
                d = np.abs(out[t:t + 64, l:l + 64].astype(int) - g["composed_u8_crops"][i].astype(int))
                assert d.max() <= 1 and (d > 0).mean() < 0.01, (sid, i)

    _run_test_py((base + " --batchSize 1 --output_dir {d}/results".format(d=tmp_path)).split())       # test_celeb.sh: batch 1
    check(tmp_path / "results")
    # a ragged last batch runs in the execution mode of the FULL batches (test.py pins EditLine2Model.batch_mode): --batchSize 5 puts five
    # 256x256 images above LOW_LATENCY_MAX_PIXELS (default mode); a file list that ends with a batch of two stays in that
    # mode, so a file's PNG is byte-identical wherever the list ends
    shutil.copy(tmp_path / "images" / "602.png", tmp_path / "images" / "602b.png")
    shutil.copy(tmp_path / "edges" / "602.png", tmp_path / "edges" / "602b.png")
    (tmp_path / "list.txt").write_text("602.png\n822.png\n873.png\n902.png\n602b.png\n873.png\n902.png\n")     # 5 + 2
    _run_test_py((base + " --batchSize 5 --output_dir {d}/results5".format(d=tmp_path)).split())
    check(tmp_path / "results5")
    (tmp_path / "list.txt").write_text("902.png\n873.png\n602.png\n822.png\n602b.png\n")                          # one full batch
    _run_test_py((base + " --batchSize 5 --output_dir {d}/results5b".format(d=tmp_path)).split())
    for n in ("602.png", "822.png", "873.png", "902.png"):
        assert np.array_equal(np.asarray(Image.open(tmp_path / "results5" / n)), np.asarray(Image.open(tmp_path / "results5b" / n))), n
    # the validator of the same directory (tools/check_checkpoint.py): both files would load
    import importlib.util
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("se_check_ckpt", os.path.join(root, "tools", "check_checkpoint.py"))
    cc = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(cc)
    assert cc.main([str(ck)]) == 0
    os.remove(ck / "latest_net_M.pth")
    with pytest.raises((FileNotFoundError, OSError, RuntimeError)):
        _run_test_py((base + " --batchSize 1 --output_dir {d}/results4".format(d=tmp_path)).split())


def test_module_prefixed_checkpoint_through_the_c_abi(golden_dir):
    """A DataParallel-style state dict ('module.' prefix, util/util.py:221-222) loaded straight through se_load_weights
    gives the reference's vectors; a truncated dict leaves the engine not ready."""
    from sketchedit_amd._lib import Engine
    g = dict(np.load(os.path.join(golden_dir, "e2e_64.npz")))
    e = Engine(0)
    sdM = {"module." + k: v for k, v in synth.make_state_dict("M", 0).items()}
    sdG = {"module." + k: v for k, v in synth.make_state_dict("G", 0).items()}
    e.load_state_dict("M", sdM)
    assert not e.weights_ready()
    e.load_state_dict("G", sdG)
    assert e.weights_ready()
    img, sk = synth.make_inputs(2, 64, 64, seed=1234)
    r = e.inference(torch.from_numpy(img).cuda(), torch.from_numpy(sk).cuda(), 1 | 2 | 16, visualize=True)
    assert np.array_equal(r["hard"].cpu().numpy(), g["hard_mask"])
    for k in ("composed", "mask", "coarse", "fine"):
        assert float(np.abs(r[k].cpu().numpy() - g[k]).max()) < 1e-3, k
    e.close()


def test_serve_matches_oracle_pipeline(model):
    """process_image (demo.py:39-73) against the same steps written with the ORACLE as the forward."""
    from PIL import Image
    from oracle import sketchedit_oracle as O
    from sketchedit_amd import serve
    rng = np.random.RandomState(11)
    w, h = 83, 70
    img = Image.fromarray(rng.randint(0, 255, (h, w, 3), dtype=np.uint8))
    sk = Image.fromarray(((rng.rand(h, w) < 0.01) * 255).astype(np.uint8))
    got = np.asarray(serve.process_image(model, img, sk))
    x = (torch.from_numpy(np.array(img.resize((80, 64))).transpose(2, 0, 1).astype(np.float32)) / 255 - 0.5) / 0.5
    m = (torch.from_numpy(np.array(sk.resize((80, 64))).astype(np.float32)) > 0).float()
    ref = O.inference(synth.make_state_dict("M", 0), synth.make_state_dict("G", 0), x[None], m[None, None])
    gen = ((torch.clamp(ref["composed"], -1, 1) + 1) / 2 * 255).numpy().astype(np.uint8)[0].transpose(1, 2, 0)
    want = np.asarray(Image.fromarray(gen).resize(img.size))
    with torch.no_grad():
        vis = model({"image": x[None], "mask": m[None, None]}, mode="visualize")
    assert np.array_equal(vis["mask"].cpu().numpy(), ref["hard_mask"].numpy()), "threshold flip on a fixed input"
    d = np.abs(got.astype(int) - want.astype(int))
    assert d.max() <= 1 and (d > 0).mean() < 0.01
    assert got.shape == (h, w, 3)


@pytest.mark.timeout(120)
def test_serve_two_contexts_one_process(model, tmp_path_factory):
    """Multi-GPU dispatch of the serving wrapper, exercised with what a 1-GPU box offers: two models (two se_ctx, two
    workspaces) in one process, one worker thread each, one shared queue.  With a second GPU visible the second model
    lives there (per-device kernel attributes, se_misc.hip ensure_max_lds)."""
    import threading
    from PIL import Image
    from sketchedit_amd import models, serve
    from sketchedit_amd.options.test_options import TestOptions
    dev2 = 1 if torch.cuda.device_count() > 1 else 0
    opt = TestOptions().parse(ARGV.format(d=tmp_path_factory.mktemp("out2")).replace("--gpu_ids 0", "--gpu_ids %d" % dev2).split(),
                              quiet=True)
    opt.isSkip = True
    m2 = models.create_model(opt)
    m2.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
    m2.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
    m2.eval()
    torch.cuda.set_device(0)
    rng = np.random.RandomState(5)
    reqs = [(Image.fromarray(rng.randint(0, 255, (64, 72, 3), dtype=np.uint8)),
             Image.fromarray(((rng.rand(64, 72) < 0.01) * 255).astype(np.uint8))) for _ in range(12)]
    single = [serve.process_image(model, i, s) for i, s in reqs]
    srv = serve.BatchingServer(models=[model, m2], max_batch=2, max_wait_s=0.001)
    outs = [None] * len(reqs)

    def worker(k):
        outs[k] = srv.submit(*reqs[k])
    ts = [threading.Thread(target=worker, args=(k,)) for k in range(len(reqs))]
    [t.start() for t in ts]
    [t.join() for t in ts]
    srv.close()
    for a, b in zip(single, outs):
        assert np.array_equal(np.asarray(a), np.asarray(b))
    assert sum(srv.batches) == len(reqs) and all(n > 0 for n in srv.batches_by_model)
