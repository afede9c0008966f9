"""Same names as reference model/common.py: step(), BaseRGBModel, MLP."""
import abc

import torch
import torch.nn as nn

from ..regnet import _no_forward


class ABCModel:
    @abc.abstractmethod
    def get_optimizer(self, opt_args):
        raise NotImplementedError()

    @abc.abstractmethod
    def epoch(self, loader, **kwargs):
        raise NotImplementedError()

    @abc.abstractmethod
    def predict(self, seq):
        raise NotImplementedError()

    @abc.abstractmethod
    def state_dict(self):
        raise NotImplementedError()

    @abc.abstractmethod
    def load(self, state_dict):
        raise NotImplementedError()


class BaseRGBModel(ABCModel):
    """reference model/common.py:30-58"""

    def get_optimizer(self, opt_args):
        base_lr = opt_args.get("lr", 1e-3)
        pred_loc_lr = base_lr / 5
        param_groups = [
            {"params": param, "lr": pred_loc_lr if "_pred_loc" in name else base_lr}
            for name, param in self._model.named_parameters()
        ]
        # The reference returns a GradScaler for fp16 autocast; this path computes in bf16 with fp32
        # accumulation and needs no loss scaling, but callers save scaler.state_dict(), so one is returned.
        scaler = torch.amp.GradScaler("cuda", enabled=False) if str(self.device).startswith("cuda") else None
        return torch.optim.AdamW(param_groups), scaler

    def _get_params(self):
        return list(self._model.parameters())

    def state_dict(self):
        return self._model.state_dict()

    def load(self, state_dict, strict=True):
        self._model.load_state_dict(state_dict, strict=strict)


def step(optimizer, scaler, loss, lr_scheduler=None, backward_only=False):
    """reference model/common.py:61-75"""
    if scaler is None:
        loss.backward()
    else:
        scaler.scale(loss).backward()
    if not backward_only:
        if scaler is None:
            optimizer.step()
        else:
            scaler.step(optimizer)
            scaler.update()
        if lr_scheduler is not None:
            lr_scheduler.step()
        optimizer.zero_grad()


class MLP(nn.Module):
    """reference model/common.py:78-92"""

    def __init__(self, input_dim, hidden_dim, output_dim, num_layers):
        super().__init__()
        self.num_layers = num_layers
        h = [hidden_dim] * (num_layers - 1)
        self.layers = nn.ModuleList(nn.Linear(n, k) for n, k in zip([input_dim] + h, h + [output_dim]))

    forward = _no_forward
