"""Network registry: name -> class lookup exactly like the reference
(/root/reference/models/networks/__init__.py:8-43): opt.netG + 'generator', case-insensitive, looked up in
models/networks/generator.py, must be a BaseNetwork; create_network = ctor, print, .cuda(), init_weights."""
import torch

from ...util import util
from .base_network import BaseNetwork
from .generator import DeepFillC2Generator, MDGenerator  # noqa: F401


def find_network_using_name(target_network_name, filename):
    cls = util.find_class_in_module(target_network_name + filename, __name__ + "." + filename)
    assert issubclass(cls, BaseNetwork), "Class %s should be a subclass of BaseNetwork" % cls
    return cls


def modify_commandline_options(parser, is_train):
    opt, _ = parser.parse_known_args()
    return find_network_using_name(opt.netG, "generator").modify_commandline_options(parser, is_train)


def create_network(cls, opt):
    net = cls(opt)
    net.print_network()
    if len(opt.gpu_ids) > 0:
        assert torch.cuda.is_available()
        net.cuda()
    if opt.init_type is not None:
        net.init_weights(opt.init_type, opt.init_variance)
    return net


def define_G(opt):
    return create_network(find_network_using_name(opt.netG, "generator"), opt)
