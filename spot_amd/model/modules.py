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
