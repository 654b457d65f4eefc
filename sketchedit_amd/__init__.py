"""sketchedit_amd -- MI355X-native drop-in for the SketchEdit inference path.

Host-side mirror of the reference's interface for that one path:
    sketchedit_amd.models.create_model(opt)(data, mode='inference') -> (composed, mask)
    sketchedit_amd.models.networks.define_G(opt)                     -> DeepFillC2Generator / MDGenerator
    sketchedit_amd.options.test_options.TestOptions().parse()
    sketchedit_amd.data.create_dataloader(opt)
All arithmetic happens in libsketchedit_hip.so (C-ABI: include/sketchedit_hip.h) through
sketchedit_amd._lib; there is no PyTorch or CPU fallback.
"""
__version__ = "0.1.0"
