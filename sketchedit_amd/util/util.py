"""Checkpoint I/O and class lookup for the inference path.

Mirrors /root/reference/util/util.py:175-187 (find_class_in_module) and :190-225
(save_network / load_network): state_dicts live at <checkpoints_dir>/<name>/<epoch>_net_<label>.pth,
a leading 'module.' (DataParallel) is stripped and the load is strict.
"""
import importlib
import os

import torch


def find_class_in_module(target_cls_name, module):
    wanted = target_cls_name.replace("_", "").lower()
    lib = importlib.import_module(module)
    for name, obj in vars(lib).items():
        if name.lower() == wanted and isinstance(obj, type):
            return obj
    raise ValueError("In %s, there should be a class whose name matches %s in lowercase without underscore(_)"
                     % (module, wanted))


def checkpoint_path(label, epoch, opt):
    return os.path.join(opt.checkpoints_dir, opt.name, "%s_net_%s.pth" % (epoch, label))


def strip_module_prefix(weights):
    return {(k[len("module."):] if k.startswith("module.") else k): v for k, v in weights.items()}


def load_network(net, label, epoch, opt):
    weights = torch.load(checkpoint_path(label, epoch, opt), map_location="cpu")
    net.load_state_dict(strip_module_prefix(weights))      # strict, as util/util.py:224
    return net


def save_network(net, label, epoch, opt):
    path = checkpoint_path(label, epoch, opt)
    os.makedirs(os.path.dirname(path), exist_ok=True)
    torch.save({k: v.detach().cpu() for k, v in net.state_dict().items()}, path)
