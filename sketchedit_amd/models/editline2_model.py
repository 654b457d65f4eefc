"""EditLine2Model: the wrapper test.py / demo.py call (reference: models/editline2_model.py:49-147,
184-200,223-242,338-370).  Only the inference-side modes exist here ('inference', 'visualize'); the
training modes of the reference ('generator', 'discriminator') depend on code it never released.

    model = create_model(opt); model.eval()
    composed, mask = model({'image': (B,3,H,W) in [-1,1], 'mask': (B,1,H,W) sketch in {0,1}}, mode='inference')

inference = netM -> mask_inpaint = (mask > 0.5) -> netG(inputs, inputs, mask_inpaint, mask_inpaint, line)
-> composed = fake*mask + inputs*(1-mask) with the SOFT mask (:132), all inside one se_inference call.
"""
import torch

from .. import _lib
from ..util import util
from . import networks


class EditLine2Model(torch.nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        networks.modify_commandline_options(parser, is_train)
        # no reference counterpart: SE_FLAG_CONSERVATIVE of the C-ABI (include/sketchedit_hip.h, INTEGRATION.md "Precision
        # choice") -- the mask predictor on the F(2x2,3x3) Winograd form; for checkpoints whose mask logits sit at 0.5
        parser.add_argument("--precision", type=str, default="f32", choices=["f32", "bf16"],
                            help="f32: the reference's arithmetic (1e-3 parity bound); bf16: SE_FLAG_BF16 -- bf16 storage and MFMA, fp32 accumulate "
                                 "(BASELINE config 5; compared against the oracle's bf16 mode, not the 1e-3 bound)")
        parser.add_argument("--conservative_mask", action="store_true",
                            help="netM on the conservative Winograd form (fewer hard-mask flips vs the fp32 CPU reference, +1.7 %% time)")
        return parser

    def __init__(self, opt):
        super().__init__()
        if getattr(opt, "isTrain", False):
            raise NotImplementedError("sketchedit_amd implements the inference path only")
        self.opt = opt
        self.netM, self.netG = self.initialize_networks(opt)
        self._engine = None

    def use_gpu(self):
        return len(self.opt.gpu_ids) > 0

    def initialize_networks(self, opt):
        netG = networks.define_G(opt)
        saved = opt.netG
        opt.netG = "MD"                                   # editline2_model.py:186-188
        netM = networks.define_G(opt)
        opt.netG = saved
        if not hasattr(opt, "isSkip"):                    # :195-197 (isSkip bypasses the checkpoint load)
            netG = util.load_network(netG, "G", opt.which_epoch, opt)
            netM = util.load_network(netM, "M", opt.which_epoch, opt)
            if self.use_gpu():
                netG.cuda()
                netM.cuda()
        return netM, netG

    def engine(self):
        if not self.use_gpu() or not torch.cuda.is_available():
            raise _lib.SketchEditHipError("EditLine2Model: this path needs an MI355X (--gpu_ids 0); the reference's "
                                          "--gpu_ids -1 CPU mode has no counterpart here")
        if self._engine is None:
            self._engine = _lib.Engine(self.opt.gpu_ids[0])
            self._engine.set_conservative(getattr(self.opt, "conservative_mask", False))
            self._engine.set_precision(getattr(self.opt, "precision", "f32") or "f32")
            self.netG.bind_engine(self._engine)
            self.netM.bind_engine(self._engine)
        self.netG.engine()                                # (re)upload weights if they changed
        self.netM.engine()
        return self._engine

    def preprocess_input(self, data):
        """:223-242.  gt / edgegt are filled from image / mask when absent (the reference only does so
        on the GPU branch and raises KeyError on CPU)."""
        dev = torch.device("cuda", self.opt.gpu_ids[0]) if self.use_gpu() else torch.device("cpu")
        data["image"] = data["image"].to(dev)
        data["gt"] = data["gt"].to(dev) if "gt" in data else data["image"]
        data["mask"] = data["mask"].to(dev)
        data["edgegt"] = data["edgegt"].to(dev) if "edgegt" in data else data["mask"]
        return data["image"], data["gt"], data["mask"], data["edgegt"], None

    def _mode_for(self, B, H, W, low_latency):
        """Execution mode of a call of B images: the caller's pin, else by the call's OWN size (an interactive caller that
        configured --batchSize 8 and sends single images keeps the low-latency kernels)."""
        return _lib.Engine.is_low_latency(int(B), H, W, low_latency)

    def batch_mode(self, H, W):
        """The mode a FULL --batchSize batch of H x W images takes -- what a batching loop pins for every batch of its file
        list.  test.py --batchSize 8 over 20 files runs batches of 8, 8, 4: chosen from each call's own size the last one
        would cross LOW_LATENCY_MAX_PIXELS into the other mode (other kernels: fp32-rounding differences, possibly another
        hard-mask pixel) -- an image's PNG must not depend on where the file list ends, so test.py passes
        `low_latency=model.batch_mode(H, W)`."""
        return _lib.Engine.is_low_latency(int(getattr(self.opt, "batchSize", 1) or 1), H, W)

    def inference_u8(self, data, low_latency=None):
        """mode='inference' followed by test.py:25-27 -- `((generated + 1) / 2 * 255).astype(uint8)` in HWC order and
        `(mask * 255).astype(uint8)` -- as ONE library call: the quantisation is fused into the forward's last kernel, so
        only uint8 leaves the device.  -> (rgb (B,H,W,3) uint8, mask (B,H,W) uint8), both on the device."""
        if self.training:
            raise NotImplementedError("call model.eval() first: only the eval branch of generate_fake exists here")
        if "image_u8" in data:
            # uint8 all the way (the dataset's u8 mode, data/testimage_dataset.py): the normalisation of
            # /root/reference/data/testimage_dataset.py:89-111 happens on the device too (table lookup, bit-identical)
            dev = torch.device("cuda", self.opt.gpu_ids[0])
            iu8, su8 = data["image_u8"].to(dev, non_blocking=True), data["mask_u8"].to(dev, non_blocking=True)
            B, H, W, _ = iu8.shape
            with torch.no_grad():
                return self.engine().inference_u8io(iu8.contiguous(), su8.contiguous(), _lib.flags_from_opt(self.opt),
                                                    low_latency=self._mode_for(B, H, W, low_latency))
        inputs, _, line, _, _ = self.preprocess_input(data)
        eng = self.engine()
        with torch.no_grad():
            return eng.inference_u8(inputs.float().contiguous(), line.float().contiguous(), _lib.flags_from_opt(self.opt),
                                    low_latency=self._mode_for(inputs.shape[0], inputs.shape[2], inputs.shape[3], low_latency))

    def forward(self, data, mode, low_latency=None):
        """`low_latency` (no reference counterpart): None = by this call's own size, True / False = pinned.  Results are
        bit-identical across batch compositions only WITHIN one mode (include/sketchedit_hip.h), so callers whose batch size
        varies per request (serve.BatchingServer, shard.sharded_inference, test.py's loop through batch_mode) pin it."""
        inputs, real_image, line, line_full, _ = self.preprocess_input(data)
        if mode not in ("inference", "visualize"):
            raise ValueError("|mode| is invalid")
        if self.training:
            raise NotImplementedError("call model.eval() first: only the eval branch of generate_fake exists here")
        eng = self.engine()
        flags = _lib.flags_from_opt(self.opt)
        with torch.no_grad():
            r = eng.inference(inputs.float().contiguous(), line.float().contiguous(), flags,
                              visualize=(mode == "visualize"),
                              low_latency=self._mode_for(inputs.shape[0], inputs.shape[2], inputs.shape[3], low_latency))
        if mode == "inference":
            return r["composed"], r["mask"]
        return {"mask": r["hard"], "maskim": r["maskim"], "coarse": r["coarse"], "fine": r["fine"],
                "composed": r["composed"]}                # :134-145
