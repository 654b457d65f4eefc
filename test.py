#!/usr/bin/env python3
"""Batch inference script with the reference's command line (/root/reference/test.py:12-37,
test_celeb.sh, test_places.sh):

    python test.py --batchSize 1 --name celeb --joint_train_inp --dataset_mode testimage \
        --image_dirs D/images --mask_dirs D/edges --image_lists D/list.txt --image_postfix .png \
        --mask_postfix .png --model editline2 --netG deepfillc2 --pool_type max --use_cam \
        --which_epoch latest --output_dir out [--output_mask_dir out_masks] [--nThreads 16 --encode_threads 48]

Differences from the reference script: PNGs are written with PIL (no cv2 in this image; the reference's
cv2.imwrite(output[:, :, ::-1]) stores RGB order on disk, as Image.fromarray(output) does),
`--synthetic_weights` makes the run self-contained when no checkpoint exists, and the loop is a pipeline
(sketchedit_amd/pipeline.py): --nThreads worker processes decode into pinned uint8 batches, the host-to-device copy, the
forward (uint8 in, uint8 out: se_inference_u8io) and the device-to-host copy run on three streams, --encode_threads
threads (or --encode_procs processes, through a shared page-locked ring) encode the PNGs; --decode_procs N replaces the DataLoader by
N decoder processes that write straight into a shared page-locked input ring.  `--serial_io` keeps the reference's serial loop; both write byte-identical files.
"""
import os
import sys

import torch

from sketchedit_amd import data, models
from sketchedit_amd.options.test_options import TestOptions
from sketchedit_amd.pipeline import InferencePipeline, save_png


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    synthetic = "--synthetic_weights" in argv
    if synthetic:
        argv.remove("--synthetic_weights")
    topt = TestOptions()
    opt = topt.parse(argv)
    if synthetic:
        opt.isSkip = True
    opt.u8_io = not opt.serial_io          # the dataset hands over the decoder's uint8 arrays (pinned), not float tensors
    dataloader = data.create_dataloader(opt)
    model = models.create_model(opt)
    if synthetic:
        from sketchedit_amd import synth
        model.netG.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("G", 0).items()})
        model.netM.load_state_dict({k: torch.from_numpy(v) for k, v in synth.make_state_dict("M", 0).items()})
    model.eval()
    mask_dir = getattr(opt, "output_mask_dir", None)
    if opt.decode_procs < 0:
        opt.decode_procs = max(0, int(opt.nThreads))      # --nThreads N = N decode workers: the pipeline's own, unless --decode_procs 0
    if not opt.serial_io:
        # every batch of the list, the ragged last one included, runs in the execution mode of a FULL --batchSize batch
        # (InferencePipeline pins model.batch_mode(H, W)): an image's PNG does not depend on where the file list ends
        pipe = InferencePipeline(model, opt.output_dir, mask_dir, encode_threads=opt.encode_threads or max(1, int(opt.nThreads)),
                                 depth=opt.pipeline_depth, encode_procs=opt.encode_procs, png_writer=opt.png_writer,
                                 decode_procs=opt.decode_procs)
        try:
            if opt.decode_procs > 0:          # the pipeline's own decoders: path lists of the same dataset, serial order
                pipe.run_paths(dataloader.dataset, opt.how_many, opt.batchSize)
            else:
                pipe.run(dataloader, opt.how_many, opt.batchSize)
        finally:
            pipe.close()
        return
    for i, data_i in enumerate(dataloader):
        if i * opt.batchSize >= opt.how_many:
            break
        # mode='inference' with (x+1)/2*255 and mask*255 -> uint8 (no clamp, as test.py:26-27) fused into the forward's
        # last kernel, already HWC: nothing but uint8 is written or copied to the host
        H, W = data_i["image"].shape[2:]
        rgb, m8 = model.inference_u8(data_i, low_latency=model.batch_mode(H, W))
        generated, mask = rgb.cpu().numpy(), m8.cpu().numpy()
        for b in range(generated.shape[0]):
            path = data_i["path"][b]
            print("process image... %s" % path)
            save_png(generated[b], os.path.join(opt.output_dir, path), opt.png_writer)
            if mask_dir is not None:
                save_png(mask[b], os.path.join(mask_dir, path), opt.png_writer)


if __name__ == "__main__":
    main()
