"""Model registry, same contract as /root/reference/models/__init__.py:5-39:
--model <name> -> module models/<name>_model.py -> class <Name>Model (case-insensitive, nn.Module)."""
import importlib

import torch


def find_model_using_name(model_name):
    modellib = importlib.import_module(__name__ + "." + model_name + "_model")
    target = model_name.replace("_", "") + "model"
    for name, cls in vars(modellib).items():
        if name.lower() == target.lower() and isinstance(cls, type) and issubclass(cls, torch.nn.Module):
            return cls
    raise ValueError("In %s_model.py, there should be a subclass of torch.nn.Module with class name that matches "
                     "%s in lowercase." % (model_name, target))


def get_option_setter(model_name):
    return find_model_using_name(model_name).modify_commandline_options


def create_model(opt):
    instance = find_model_using_name(opt.model)(opt)
    print("model [%s] was created" % type(instance).__name__)
    return instance
