"""RegNet-Y parameter tree with timm's attribute and state-dict names.

The reference builds its backbone with `timm.create_model("regnety_008")`
(train_e2e_spatial.py:305-314) and then touches `.head.fc`, `.stem.conv`, `.s1..s4` and each
block's `.conv1` (model/shift.py:191-205).  timm is a third-party dependency that is not part
of the reference; this module reproduces the structure of its RegNet (stem ConvNormAct, four
RegStages of Bottleneck blocks named b1..bN, SEModule with fc1/fc2 1x1 convs, ClassifierHead)
so that checkpoints written by the reference load by key.  The modules only hold parameters
and buffers: the arithmetic is done by spot_amd.engine on NHWC bf16 tensors.
"""
from collections import OrderedDict

import torch.nn as nn

REGNET_CFGS = {
    # name: (stem width, stage widths, stage depths, group width)
    "regnety_008": (32, (64, 128, 320, 768), (1, 3, 8, 2), 16),
}


def _no_forward(self, *a, **k):
    raise RuntimeError("parameter container: the forward pass runs in spot_amd.engine (HIP kernels)")


class ConvBnAct(nn.Module):
    """timm ConvNormAct: .conv (no bias) + .bn (+ReLU when apply_act)."""

    def __init__(self, cin, cout, k, stride=1, groups=1, apply_act=True):
        super().__init__()
        self.conv = nn.Conv2d(cin, cout, k, stride=stride, padding=k // 2, groups=groups, bias=False)
        self.bn = nn.BatchNorm2d(cout)
        self.apply_act = apply_act

    forward = _no_forward


class SEModule(nn.Module):
    def __init__(self, channels, rd_channels):
        super().__init__()
        self.fc1 = nn.Conv2d(channels, rd_channels, 1, bias=True)
        self.fc2 = nn.Conv2d(rd_channels, channels, 1, bias=True)

    forward = _no_forward


class Bottleneck(nn.Module):
    def __init__(self, cin, cout, stride, group_size, se_ratio=0.25):
        super().__init__()
        self.conv1 = ConvBnAct(cin, cout, 1)
        self.conv2 = ConvBnAct(cout, cout, 3, stride=stride, groups=cout // group_size)
        self.se = SEModule(cout, int(round(cin * se_ratio)))
        self.conv3 = ConvBnAct(cout, cout, 1, apply_act=False)
        nn.init.zeros_(self.conv3.bn.weight)          # timm zero_init_last
        if cin != cout or stride != 1:
            self.downsample = ConvBnAct(cin, cout, 1, stride=stride, apply_act=False)
        else:
            self.downsample = nn.Identity()
        self.cin, self.cout, self.stride = cin, cout, stride

    forward = _no_forward


class ClassifierHead(nn.Module):
    def __init__(self, feat, num_classes=1000):
        super().__init__()
        self.fc = nn.Linear(feat, num_classes)

    forward = _no_forward


class RegNet(nn.Module):
    def __init__(self, name="regnety_008", in_chans=3):
        super().__init__()
        if name not in REGNET_CFGS:
            raise NotImplementedError(name)
        stem_w, widths, depths, gs = REGNET_CFGS[name]
        self.stem = ConvBnAct(in_chans, stem_w, 3, stride=2)
        cin = stem_w
        for si, (w, d) in enumerate(zip(widths, depths)):
            blocks = OrderedDict()
            for bi in range(d):
                blocks[f"b{bi + 1}"] = Bottleneck(cin, w, 2 if bi == 0 else 1, gs)
                cin = w
            setattr(self, f"s{si + 1}", nn.Sequential(blocks))
        self.final_conv = nn.Identity()
        self.head = ClassifierHead(cin)
        self.num_features = cin

    forward = _no_forward


def create_model(name, pretrained=False, in_chans=3):
    """Stands where `timm.create_model` does in the reference.  Pretrained ImageNet weights need
    a network fetch, which this build never does: load a checkpoint with E2EModel.load instead."""
    return RegNet(name, in_chans=in_chans)
