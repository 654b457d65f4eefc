#!/usr/bin/env python3
"""Batch inference script with the reference's command line (/root/reference/test.py:12-37,
test_celeb.sh, test_places.sh):

    python test.py --batchSize 1 --name celeb --joint_train_inp --dataset_mode testimage \
        --image_dirs D/images --mask_dirs D/edges --image_lists D/list.txt --image_postfix .png \
        --mask_postfix .png --model editline2 --netG deepfillc2 --pool_type max --use_cam \
        --which_epoch latest --output_dir out [--output_mask_dir out_masks]

Differences from the reference script: PNGs are written with PIL (no cv2 in this image; the reference's
cv2.imwrite(output[:, :, ::-1]) stores RGB order on disk, as Image.fromarray(output) does), and
`--synthetic_weights` makes the run self-contained when no checkpoint exists.
"""
import os
import sys

import torch
from PIL import Image

from sketchedit_amd import data, models
from sketchedit_amd.options.test_options import TestOptions


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    synthetic = "--synthetic_weights" in argv
    if synthetic:
        argv.remove("--synthetic_weights")
    topt = TestOptions()
    opt = topt.parse(argv)
    if synthetic:
        opt.isSkip = True
    dataloader = data.create_dataloader(opt)
    model = models.create_model(opt)
    if synthetic:
        from sketchedit_amd import synth
        model.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
        model.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
    model.eval()
    for i, data_i in enumerate(dataloader):
        if i * opt.batchSize >= opt.how_many:
            break
        # mode='inference' with (x+1)/2*255 and mask*255 -> uint8 (no clamp, as test.py:26-27) fused into the forward's
        # last kernel, already HWC: nothing but uint8 is written or copied to the host
        # every batch of the list, the ragged last one included, in the execution mode of a FULL --batchSize batch
        H, W = data_i["image"].shape[2:]
        rgb, m8 = model.inference_u8(data_i, low_latency=model.batch_mode(H, W))
        generated, mask = rgb.cpu().numpy(), m8.cpu().numpy()
        for b in range(generated.shape[0]):
            path = data_i["path"][b]
            print("process image... %s" % path)
            Image.fromarray(generated[b]).save(os.path.join(opt.output_dir, path))
            if getattr(opt, "output_mask_dir", None) is not None:
                Image.fromarray(mask[b]).save(os.path.join(opt.output_mask_dir, path))


if __name__ == "__main__":
    main()
