"""minGRU head: same names as reference model/min_gru.py (parameter containers)."""
import torch
import torch.nn as nn

from ..regnet import _no_forward


class MinGRU(nn.Module):
    """reference model/min_gru.py:69-75"""

    def __init__(self, input_size, hidden_size):
        super().__init__()
        self.hidden_size = hidden_size
        self.linear_z = nn.Linear(input_size, hidden_size)
        self.linear_h_tilde = nn.Linear(input_size, hidden_size)

    forward = _no_forward


class MinRNNPredictor(nn.Module):
    """reference model/min_gru.py:4-42 (mingru only: MinLSTM is never constructed by the reference,
    train_e2e_spatial.py:389)"""

    def __init__(self, input_size, hidden_size, output_size, n_layers=1, rnn_type="minGRU", batch_first=True):
        super().__init__()
        if rnn_type.lower() != "mingru":
            raise ValueError("rnn_type must be 'minGRU' or 'minLSTM'")
        self.hidden_size = hidden_size
        self.batch_first = batch_first
        self.n_layers = n_layers
        self.rnn_type = rnn_type.lower()
        self.rnn_layers = nn.ModuleList()
        d = input_size
        for _ in range(n_layers):
            self.rnn_layers.append(MinGRU(d, hidden_size))
            d = hidden_size
        self.fc = nn.Linear(hidden_size, output_size)
        self.h0 = nn.ParameterList(
            [nn.Parameter(torch.zeros(1, hidden_size), requires_grad=False) for _ in range(n_layers)])

    forward = _no_forward
