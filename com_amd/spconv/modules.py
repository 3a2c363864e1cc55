"""SparseModule / SparseSequential container dispatch (used at
pcdet/models/backbones_3d/spconv_backbone.py:21-25,30,77-81)."""
from collections import OrderedDict

import torch
from torch import nn

from .core import SparseConvTensor


class SparseModule(nn.Module):
    """Marker base class: modules that take and return a SparseConvTensor."""
    pass


def is_spconv_module(module):
    return isinstance(module, SparseModule)


class SparseSequential(SparseModule):
    """Like nn.Sequential; sparse modules get the tensor, plain nn.Modules get ``.features``.
    Children are named "0", "1", ... so state-dict keys match the reference."""

    def __init__(self, *args, **kwargs):
        super().__init__()
        if len(args) == 1 and isinstance(args[0], OrderedDict):
            for key, module in args[0].items():
                self.add_module(key, module)
        else:
            for idx, module in enumerate(args):
                self.add_module(str(idx), module)
        for name, module in kwargs.items():
            if name in self._modules:
                raise ValueError("name exists.")
            self.add_module(name, module)

    def __getitem__(self, idx):
        if not (-len(self) <= idx < len(self)):
            raise IndexError("index {} is out of range".format(idx))
        if idx < 0:
            idx += len(self)
        it = iter(self._modules.values())
        for _ in range(idx):
            next(it)
        return next(it)

    def __len__(self):
        return len(self._modules)

    def add(self, module, name=None):
        if name is None:
            name = str(len(self._modules))
            if name in self._modules:
                raise KeyError("name exists")
        self.add_module(name, module)

    def forward(self, input):
        from . import functional as Fsp
        mods = list(self._modules.values())
        i = 0
        while i < len(mods):
            module = mods[i]
            i += 1
            if is_spconv_module(module):
                assert isinstance(input, SparseConvTensor)
                input = module(input)
            elif isinstance(input, SparseConvTensor):
                if input.indices.shape[0] == 0:
                    continue
                if isinstance(module, nn.BatchNorm1d) and Fsp._fusable(module, input.features):
                    # BatchNorm1d (+ the ReLU that follows it) in one fused pass pair; the modules, their
                    # parameters and state-dict entries are untouched
                    relu = i < len(mods) and type(mods[i]) is nn.ReLU
                    input = input.replace_feature(Fsp.batch_norm_act(module, input.features, None, relu,
                                                                     input.num_rows))
                    if relu:
                        i += 1
                else:
                    input = input.replace_feature(module(input.features))
            else:
                input = module(input)
        return input
