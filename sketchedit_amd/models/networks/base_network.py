"""BaseNetwork: the common parent the reference's registry insists on
(/root/reference/models/networks/base_network.py:5-57, models/networks/__init__.py:13-14)."""
import torch.nn as nn


class BaseNetwork(nn.Module):
    @staticmethod
    def modify_commandline_options(parser, is_train):
        return parser

    def print_network(self):
        n = sum(p.numel() for p in self.parameters())
        print("Network [%s] was created. Total number of parameters: %.1f million. "
              "To see the architecture, do print(network)." % (type(self).__name__, n / 1e6))

    def init_weights(self, init_type="normal", gain=0.02):
        # The reference's init_func only touches classes whose NAME contains 'Conv' or 'Linear'
        # (base_network.py:31); its layers are called 'gen_conv', so nothing is re-initialised and
        # every layer keeps nn.Conv2d's default init.  Same here: a deliberate no-op.
        return
