"""The two generators of the path, registered under the reference's names:
    --netG deepfillc2 -> DeepFillC2Generator   (/root/reference/models/networks/editline_g.py:13-221)
    --netG MD         -> MDGenerator           (/root/reference/models/networks/editline2_g.py:13-94)
Same parameter names / shapes (so reference checkpoints load strictly), same forward signatures.
forward() hands the whole network to libsketchedit_hip.so (se_netG_forward / se_netM_forward)."""
import re

from ... import _lib, synth
from .base_network import BaseNetwork
from .ops import ContextualAttention, gen_conv, gen_deconv


def _build_layers(module, table, relu_layers=()):
    for name, cin, cout, k in table:
        if name.endswith("_upsample_conv"):
            layer = gen_deconv(cin, cout)
        else:
            stride = 2 if name.endswith("_downsample") else 1
            m = re.search(r"(\d+)_atrous$", name)
            rate = {7: 2, 8: 4, 9: 8, 10: 16}[int(m.group(1))] if m else 1
            act = None if name.endswith("17") else ("relu" if name in relu_layers else "elu")
            layer = gen_conv(cin, cout, k, stride, rate, activation=act)
        setattr(module, name, layer)


class _HipNetwork(BaseNetwork):
    NET = None

    def __init__(self):
        super().__init__()
        self._engine = None
        self._uploaded_version = None

    # the model wrapper injects ONE engine into both nets so se_inference can run the fused path
    def bind_engine(self, engine):
        self._engine = engine
        self._uploaded_version = None

    def _version(self):
        return tuple(p._version for p in self.parameters()) + tuple(p.data_ptr() for p in self.parameters())

    def engine(self):
        p = next(self.parameters())
        if not p.is_cuda:
            raise _lib.SketchEditHipError("%s: move the network to an MI355X first (.cuda()); there is no CPU "
                                          "implementation of this path" % type(self).__name__)
        if self._engine is None:
            self._engine = _lib.Engine(p.device.index or 0)
        v = self._version()
        if v != self._uploaded_version:          # (re)pack weights after load_state_dict / .cuda()
            self._engine.load_state_dict(self.NET, self.state_dict())
            self._uploaded_version = v
        return self._engine


class DeepFillC2Generator(_HipNetwork):
    NET = "G"

    @staticmethod
    def modify_commandline_options(parser, is_train):
        parser.add_argument("--use_cam", action="store_true", help="use the contextual attention module")
        parser.add_argument("--pool_type", default="avg", help="style-branch global pooling: avg | max")
        parser.add_argument("--no_mask_cc", action="store_true", help="do not mask the style-branch input")
        parser.add_argument("--no_mask_coarse", action="store_true", help="feed stage 2 the raw coarse result")
        return parser

    def __init__(self, opt):
        super().__init__()
        self.opt = opt
        self.cnum = 48
        self.cam = ContextualAttention()
        _build_layers(self, synth.G_LAYERS, relu_layers=("pmconv6",))
        if opt.pool_type not in ("avg", "max"):
            raise NotImplementedError(opt.pool_type)

    def forward(self, x, x2, mask, mask2, guide=None):
        if guide is None:                      # editline_g.py:127-128
            guide = x.new_ones((x.shape[0], 1) + tuple(x.shape[2:]))
        args = [t.float().contiguous() for t in (x, x2, mask, mask2, guide)]
        return self.engine().netG(*args, _lib.flags_from_opt(self.opt))


class MDGenerator(_HipNetwork):
    NET = "M"

    def __init__(self, opt):
        super().__init__()
        _build_layers(self, synth.M_LAYERS)

    def forward(self, x, guide):
        return self.engine().netM(x.float().contiguous(), guide.float().contiguous(), want_image=True)
