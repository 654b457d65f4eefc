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
        parser.set_defaults(preprocess_mode="scale_width_and_crop", crop_size=256, load_size=256, display_winsize=256,
                            serial_batches=True, no_flip=True, phase="test")
        return parser
