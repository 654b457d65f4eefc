"""TestOptions (reference: /root/reference/options/test_options.py:4-21)."""
from .base_options import BaseOptions


class TestOptions(BaseOptions):
    isTrain = False

    def initialize(self, parser):
        super().initialize(parser)
        parser.add_argument("--dataset_mode", type=str, default="base")
        parser.add_argument("--port", type=int, default=9998)
        parser.add_argument("--filelist", type=str, default="./static/images/example.txt")
        parser.add_argument("--results_dir", type=str, default="./results/")
        parser.add_argument("--which_epoch", type=str, default="latest", help="checkpoint epoch to load")
        parser.add_argument("--how_many", type=int, default=float("inf"), help="how many test images to run")
        # no reference counterpart: the I/O pipeline of test.py (sketchedit_amd/pipeline.py)
        parser.add_argument("--encode_threads", type=int, default=0, help="PNG encoder threads (0: as many as --nThreads, at least 1)")
        parser.add_argument("--decode_procs", type=int, default=-1, help="decode with this many worker PROCESSES straight into a shared page-locked ring instead of "
                            "DataLoader workers (-1: as many as --nThreads, i.e. --nThreads keeps its meaning 'decode workers'; 0: the DataLoader); "
                            "images of one batch must have one size")
        parser.add_argument("--encode_procs", type=int, default=0, help="PNG encoder PROCESSES fed through a shared page-locked ring (0: threads); one process tops out near 2000 images/s")
        parser.add_argument("--png_writer", type=str, default="pil", choices=["pil", "fast"],
                            help="pil: PIL's defaults (adaptive filters, zlib 6; the files this repo has always written); fast: the reference's "
                                 "cv2.imwrite defaults (SUB filter, zlib 1, RLE) written directly: 7x cheaper, same pixels")
        parser.add_argument("--pipeline_depth", type=int, default=2, help="batches in flight on the device")
        parser.add_argument("--serial_io", action="store_true", help="the reference's serial loop: forward, .cpu(), write, one batch at a time")
        parser.set_defaults(preprocess_mode="scale_width_and_crop", crop_size=256, load_size=256, display_winsize=256,
                            serial_batches=True, no_flip=True, phase="test")
        return parser
