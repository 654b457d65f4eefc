"""BaseDataset: the option-setter / initialize contract every `--dataset_mode` resolves to (reference:
/root/reference/data/base_dataset.py:10-19).  `--dataset_mode base` is the default of the option parser and is
what demo.py-style launches (no list files, requests arrive over HTTP) run with; it adds no options and holds no
samples.  The transform helpers further down in the reference's file (get_params / get_transform, torchvision)
belong to the training datasets, which are not part of this path."""
import torch.utils.data


class BaseDataset(torch.utils.data.Dataset):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def initialize(self, opt):
        self.opt = opt

    def __len__(self):
        return 0

    def __getitem__(self, index):
        raise IndexError("BaseDataset holds no samples")
