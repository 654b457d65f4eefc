"""List-file driven (image, sketch) pairs -> tensors: the input contract of the path
(reference: /root/reference/data/testimage_dataset.py:9-111).  PIL + numpy only (no torchvision):
image -> float (3,H,W) in [-1,1] ((x/255 - 0.5)/0.5), sketch -> 'L', resized to the image, (>0) as {0,1}."""
import os

import numpy as np
import torch
from PIL import Image

from .base_dataset import BaseDataset


class TestImageDataset(BaseDataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        for flag, req, default in (("--image_dirs", True, None), ("--mask_dirs", True, None),
                                   ("--image_postfix", False, ".jpg"), ("--mask_postfix", False, ".png"),
                                   ("--image_lists", True, None), ("--output_labels", False, None),
                                   ("--output_dir", True, None), ("--output_mask_dir", False, None)):
            parser.add_argument(flag, type=str, required=req, default=default)
        return parser

    def initialize(self, opt):
        self.opt = opt
        # u8 mode (no reference counterpart; set by test.py's pipelined loop): items carry the decoder's uint8 arrays --
        # 'image_u8' (H,W,3), 'mask_u8' (H,W) -- instead of the float tensors; the normalisation below then runs on the device
        # (se_inference_u8io), bit-identically.  A quarter of the bytes through the worker pipes and the PCIe link.
        self.u8 = bool(getattr(opt, "u8_io", False))
        os.makedirs(opt.output_dir, exist_ok=True)
        if opt.output_mask_dir is not None:
            os.makedirs(opt.output_mask_dir, exist_ok=True)
        self.image_paths, self.mask_paths, self.output_paths = self.get_paths(opt)

    @staticmethod
    def get_paths(opt):
        image_dirs, mask_dirs = opt.image_dirs.split(";"), opt.mask_dirs.split(";")
        labels = opt.output_labels.split(";") if opt.output_labels is not None else None
        images, masks, outputs = [], [], []
        for i, list_file in enumerate(opt.image_lists.split(";")):
            with open(list_file) as f:
                stems = [ln.strip("\n").replace(opt.image_postfix, "") for ln in f.readlines()]
            for s in stems:
                images.append(os.path.join(image_dirs[i], s + opt.image_postfix))
                masks.append(os.path.join(mask_dirs[i], s + opt.mask_postfix))
                outputs.append((labels[i] + "_" if labels is not None else "") + s + opt.image_postfix)
        return images, masks, outputs

    def __len__(self):
        return len(self.image_paths)

    def __getitem__(self, index):
        image = Image.open(self.image_paths[index]).convert("RGB")
        w, h = image.size
        if self.u8:
            sketch = Image.open(self.mask_paths[index]).convert("L").resize((w, h))
            return {"image_u8": torch.from_numpy(np.array(image, dtype=np.uint8)),
                    "mask_u8": torch.from_numpy(np.array(sketch, dtype=np.uint8)), "path": self.output_paths[index]}
        arr = np.asarray(image, dtype=np.float32).transpose(2, 0, 1) / 255.0
        image_tensor = torch.from_numpy((arr - 0.5) / 0.5)
        sketch = Image.open(self.mask_paths[index]).convert("L").resize((w, h))
        mask_tensor = torch.from_numpy((np.asarray(sketch, dtype=np.float32)[None] / 255.0 > 0).astype(np.float32))
        return {"image": image_tensor, "gt": image_tensor, "mask": mask_tensor, "path": self.output_paths[index]}
