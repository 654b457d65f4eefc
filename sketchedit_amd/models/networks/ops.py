"""Operator modules of the path, same names / constructor arguments as the reference's
models/networks/utils.py:9-51 (gen_conv, gen_deconv) and splitcam.py (the two attention halves are
exposed as one ContextualAttention op).  They are parameter holders with nn.Conv2d's state_dict
contract; called on their own they run the per-op C-ABI entry points (se_gated_conv2d / se_attention)
-- the generators never call them layer by layer, they hand the whole forward to the library."""
import torch
import torch.nn as nn

from ... import _lib


def _engine_for(t):
    if not t.is_cuda:
        raise _lib.SketchEditHipError("sketchedit_amd ops run on an MI355X only (got a CPU tensor); "
                                      "there is no CPU fallback")
    return _lib.shared_engine(t.device.index or 0)


class gen_conv(nn.Conv2d):
    """Conv2d(bias, padding=rate*(k-1)/2, dilation=rate) then ELU(x[:half]) * sigmoid(x[half:]);
    raw output when out_channels == 3 or activation is None (utils.py:27)."""

    def __init__(self, cin, cout, ksize, stride=1, rate=1, activation="elu"):
        p = int(rate * (ksize - 1) / 2)
        super().__init__(cin, cout, ksize, stride=stride, padding=p, dilation=rate, groups=1, bias=True)
        if isinstance(activation, nn.ELU):
            activation = "elu"
        elif isinstance(activation, nn.ReLU):
            activation = "relu"
        if activation not in ("elu", "relu", None):
            raise ValueError("activation must be 'elu', 'relu' or None")
        self.act = activation
        self.rate = rate
        self.upsample = False

    def forward(self, x):
        eng = _engine_for(x)
        return eng.gated_conv2d(x.contiguous(), self.weight.detach().cpu().numpy(), self.bias.detach().cpu().numpy(),
                                stride=self.stride[0], rate=self.rate, act=self.act, upsample=self.upsample)


class gen_deconv(gen_conv):
    """nearest x2 upsample followed by gen_conv(k=3) (utils.py:35-51); the upsample is fused into
    the kernel's gather (src = dst >> 1), the 4x tensor is never materialised."""

    def __init__(self, cin, cout):
        super().__init__(cin, cout, 3)
        self.upsample = True


class ContextualAttention(nn.Module):
    """cam_1 + cam_2 as configured at editline_g.py:35-42 (patch 4, stride 2, th 0.1, scale 10)."""

    def forward(self, x, mask_full):
        return _engine_for(x).attention(x.contiguous(), mask_full.contiguous())
