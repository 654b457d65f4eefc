"""Command-line options of the inference path -- flag names, defaults and the two-pass parse
(base flags -> model flags -> dataset flags) follow /root/reference/options/base_options.py:16-186 so that
test_celeb.sh / test_places.sh style command lines keep working.  Flags that the reference declares but
nothing on this path reads (SPADE heritage: --norm_G, --label_nc, ...) are still accepted."""
import argparse
import sys

import torch

# (flag, kwargs) -- kept as data so the parser is built in one loop
_BASE_FLAGS = [
    ("--name", dict(type=str, default="label2coco", help="experiment name: checkpoints/<name>/")),
    ("--joint_train_inp", dict(action="store_true", help="zero the guide channel of the style branch")),
    ("--gpu_ids", dict(type=str, default="0", help="gpu ids, e.g. 0 or 0,1,2; -1 is rejected at run time (no CPU path)")),
    ("--checkpoints_dir", dict(type=str, default="./checkpoints")),
    ("--model", dict(type=str, default="pix2pix", help="which model: editline2")),
    ("--norm_G", dict(type=str, default="spectralinstance")),
    ("--norm_D", dict(type=str, default="spectralinstance")),
    ("--norm_E", dict(type=str, default="spectralinstance")),
    ("--phase", dict(type=str, default="train")),
    ("--batchSize", dict(type=int, default=1)),
    ("--preprocess_mode", dict(type=str, default="scale_width_and_crop")),
    ("--load_size", dict(type=int, default=1024)),
    ("--crop_size", dict(type=int, default=512)),
    ("--aspect_ratio", dict(type=float, default=1.0)),
    ("--label_nc", dict(type=int, default=182)),
    ("--contain_dontcare_label", dict(action="store_true")),
    ("--output_nc", dict(type=int, default=3)),
    ("--dataroot", dict(type=str, default="./datasets/cityscapes/")),
    ("--serial_batches", dict(action="store_true")),
    ("--no_flip", dict(action="store_true")),
    ("--nThreads", dict(type=int, default=0, help="data loading worker processes")),
    ("--max_dataset_size", dict(type=int, default=sys.maxsize)),
    ("--load_from_opt_file", dict(action="store_true")),
    ("--cache_filelist_write", dict(action="store_true")),
    ("--cache_filelist_read", dict(action="store_true")),
    ("--display_winsize", dict(type=int, default=400)),
    ("--netG", dict(type=str, default="spade", help="generator: deepfillc2")),
    ("--ngf", dict(type=int, default=64)),
    ("--init_type", dict(type=str, default="xavier")),
    ("--init_variance", dict(type=float, default=0.02)),
    ("--z_dim", dict(type=int, default=256)),
    ("--no_instance", dict(action="store_true")),
    ("--nef", dict(type=int, default=16)),
    ("--use_vae", dict(action="store_true")),
]


class BaseOptions:
    isTrain = False

    def __init__(self):
        self.initialized = False
        self.parser = None

    def initialize(self, parser):
        for flag, kw in _BASE_FLAGS:
            parser.add_argument(flag, **kw)
        self.initialized = True
        return parser

    def gather_options(self, argv=None):
        from .. import data, models
        parser = argparse.ArgumentParser(formatter_class=argparse.ArgumentDefaultsHelpFormatter)
        parser = self.initialize(parser)
        opt, _ = parser.parse_known_args(argv)
        # the second pass of the reference re-reads sys.argv inside the setters, so honour argv here too
        saved = sys.argv
        if argv is not None:
            sys.argv = [saved[0] if saved else "prog"] + list(argv)
        try:
            parser = models.get_option_setter(opt.model)(parser, self.isTrain)
            parser = data.get_option_setter(opt.dataset_mode)(parser, self.isTrain)
        finally:
            sys.argv = saved
        self.parser = parser
        return parser.parse_args(argv)

    def print_options(self, opt):
        lines = ["----------------- Options ---------------"]
        for k, v in sorted(vars(opt).items()):
            default = self.parser.get_default(k)
            note = "\t[default: %s]" % str(default) if v != default else ""
            lines.append("{:>25}: {:<30}{}".format(str(k), str(v), note))
        lines.append("----------------- End -------------------")
        print("\n".join(lines))

    def parse(self, argv=None, quiet=False):
        opt = self.gather_options(argv)
        opt.isTrain = self.isTrain
        if not quiet:
            self.print_options(opt)
        opt.semantic_nc = opt.label_nc + (1 if opt.contain_dontcare_label else 0) + (0 if opt.no_instance else 1)
        ids = [int(s) for s in opt.gpu_ids.split(",")]
        opt.gpu_ids = [i for i in ids if i >= 0]
        if opt.gpu_ids and torch.cuda.is_available():
            torch.cuda.set_device(opt.gpu_ids[0])
        assert len(opt.gpu_ids) == 0 or opt.batchSize % len(opt.gpu_ids) == 0, \
            "Batch size %d is wrong. It must be a multiple of # GPUs %d." % (opt.batchSize, len(opt.gpu_ids))
        self.opt = opt
        return opt
