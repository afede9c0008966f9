"""Prediction heads: same names as reference model/modules.py (parameter containers)."""
import torch.nn as nn

from ..regnet import _no_forward


class FCPrediction(nn.Module):
    """reference model/modules.py:7-17"""

    def __init__(self, feat_dim, num_classes):
        super().__init__()
        self._fc_out = nn.Linear(feat_dim, num_classes)

    forward = _no_forward


class GRUPrediction(nn.Module):
    """reference model/modules.py:20-36"""

    def __init__(self, feat_dim, num_classes, hidden_dim, num_layers=1):
        super().__init__()
        self._gru = nn.GRU(feat_dim, hidden_dim, num_layers=num_layers, batch_first=True, bidirectional=True)
        self._fc_out = FCPrediction(2 * hidden_dim, num_classes)
        self._dropout = nn.Dropout()

    forward = _no_forward


class ChannelAttention(nn.Module):
    """reference model/modules.py:86-105 (state-dict keys fc.0.weight [F/r, F], fc.2.weight [F, F/r])"""

    def __init__(self, feat_dim, reduction=8):
        super().__init__()
        self.fc = nn.Sequential(nn.Linear(feat_dim, feat_dim // reduction, bias=False), nn.ReLU(inplace=True),
                                nn.Linear(feat_dim // reduction, feat_dim, bias=False), nn.Sigmoid())

    forward = _no_forward
