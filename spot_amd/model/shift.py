"""Temporal shift insertion: same names as reference model/shift.py.

GatedShift / _GSM / TemporalShift hold the parameters (`gsm.conv3D`, `gsm.bn`, `net`) under the
reference's state-dict keys; the gate, shift and their backward run as HIP kernels
(csrc/stes_gsm.hip) from spot_amd.engine.
"""
import math

import torch.nn as nn

from ..regnet import ConvBnAct, RegNet, _no_forward


class _GSM(nn.Module):
    """reference model/shift.py:83-95"""

    def __init__(self, fPlane, num_segments=3):
        super().__init__()
        self.conv3D = nn.Conv3d(fPlane, 2, (3, 3, 3), stride=1, padding=(1, 1, 1), groups=2)
        nn.init.constant_(self.conv3D.weight, 0)
        nn.init.constant_(self.conv3D.bias, 0)
        self.fPlane = fPlane
        self.num_segments = num_segments
        self.bn = nn.BatchNorm3d(num_features=fPlane)

    forward = _no_forward


class GatedShift(nn.Module):
    """reference model/shift.py:135-161"""

    def __init__(self, net, n_segment, n_div):
        super().__init__()
        if isinstance(net, ConvBnAct):
            channels = net.conv.in_channels
        elif isinstance(net, nn.Conv2d):
            channels = net.in_channels
        else:
            raise NotImplementedError(type(net))
        self.fold_dim = math.ceil(channels // n_div / 4) * 4
        self.gsm = _GSM(self.fold_dim, n_segment)
        self.net = net
        self.n_segment = n_segment

    forward = _no_forward


class TemporalShift(nn.Module):
    """reference model/shift.py:21-35 (in-place variant: the block's shortcut sees the shifted tensor)"""

    def __init__(self, net, n_segment, n_div, inplace=True):
        super().__init__()
        self.net = net
        self.n_segment = n_segment
        self.fold_div = n_div
        self.inplace = inplace

    forward = _no_forward


def make_temporal_shift(net, clip_len, is_gsm=False):
    """reference model/shift.py:164-226, RegNet branch (:191-205); other backbones are not built here."""
    if not isinstance(net, RegNet):
        raise NotImplementedError("Unsupported architecture")
    for stage in (net.s1, net.s2, net.s3, net.s4):
        for blk in stage.children():
            if is_gsm:
                blk.conv1 = GatedShift(blk.conv1, n_segment=clip_len, n_div=4)
            else:
                blk.conv1 = TemporalShift(blk.conv1, n_segment=clip_len, n_div=8)
