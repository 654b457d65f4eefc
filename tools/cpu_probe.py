import time, torch, sys, os
sys.path.insert(0, '.')
from oracle import sketchedit_oracle as O
from sketchedit_amd import synth
WM = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()}
WG = {k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()}
img, sk = synth.make_inputs(2, 256, 256)
img, sk = torch.from_numpy(img), torch.from_numpy(sk)
print("cpus", os.cpu_count())
for t in (16, 32, 64, 128):
    torch.set_num_threads(t)
    O.inference(WM, WG, img, sk)
    t0 = time.perf_counter(); O.inference(WM, WG, img, sk); dt = time.perf_counter() - t0
    print("threads", t, "B=2 256x256: %.2f s -> %.2f img/s" % (dt, 2 / dt), flush=True)
