"""The training step in STOCK torch ops on the CPU -- a second restatement, beside the explicit-algebra oracle.

TEST / MEASUREMENT INFRASTRUCTURE ONLY (same rule as deepsdf_oracle.py): imported by tests/ and by bench.py's
``cpu_baseline`` leg, never by the product.  It exists because the explicit oracle's hand-derived backward is not what a
CPU user of the reference would run: the reference executes its step with F.linear + autograd + nn.Embedding(max_norm) +
torch.optim.Adam (train_deep_sdf.py:385-411,505-545; deep_sdf_decoder.py:76-111).  This module issues exactly that op
sequence from the oracle's layer table, with the dropout masks injected (the hash masks, since torch's Philox stream is not
part of the specification), so that

  * bench.py can time "the reference's CPU path" on the GPU box's host cores (the reference's own files do not travel);
  * tests can cross-check the two restatements against each other and against the reference-generated goldens.
"""
from typing import Dict, List, Optional, Sequence

import torch
import torch.nn.functional as F

from .deepsdf_oracle import Net, dropout_masks


class NativeStep:
    """State = leaf tensors in the reference's named_parameters() order + an nn.Embedding(max_norm) + ONE torch.optim.Adam
    with the two parameter groups (train_deep_sdf.py:400-411)."""

    def __init__(self, net: Net, params: Dict[str, torch.Tensor], latents: torch.Tensor, *, code_bound: Optional[float],
                 lr_decoder: float = 5e-4, lr_latent: float = 1e-3):
        self.net = net
        self.params = {k: v.detach().clone().requires_grad_(True) for k, v in params.items()}
        self.lat = torch.nn.Embedding(latents.shape[0], latents.shape[1], max_norm=code_bound, dtype=latents.dtype)
        with torch.no_grad():
            self.lat.weight.copy_(latents)
        self.opt = torch.optim.Adam([{"params": list(self.params.values()), "lr": lr_decoder},
                                     {"params": self.lat.parameters(), "lr": lr_latent}])
        self.step_count = 0

    def forward(self, x0, training: bool, masks: Optional[Sequence[Optional[torch.Tensor]]] = None):
        net, p = self.net, self.params
        scale = 1.0 / (1.0 - net.dropout_prob) if net.dropout_prob < 1.0 else 0.0
        x = x0
        last = net.n_lin - 1
        for l, ly in enumerate(net.layers):
            if ly.skip_in:
                x = torch.cat([x, x0], 1)
            if ly.weight_norm:       # what parametrizations.weight_norm evaluates on every access (torch._weight_norm, dim 0)
                w = torch._weight_norm(p[f"lin{l}.parametrizations.weight.original1"],
                                       p[f"lin{l}.parametrizations.weight.original0"], 0)
            else:
                w = p[f"lin{l}.weight"]
            x = F.linear(x, w, p[f"lin{l}.bias"])
            if l < last:
                x = F.relu(x)
                if training and ly.dropout and net.dropout_prob > 0.0:
                    x = x * (masks[l].to(x.dtype) * scale)
            elif net.use_tanh:
                x = torch.tanh(x)
        return torch.tanh(x)

    def step(self, indices, xyz, sdf_gt, *, delta: float, code_reg: bool = True, code_reg_lambda: float = 1e-4, epoch: int = 1,
             training: bool = True, seed: int = 0, masks: Optional[List[Optional[torch.Tensor]]] = None) -> float:
        """One optimiser step (no --batch_split).  Returns the loss."""
        n = xyz.shape[0]
        if masks is None and training and self.net.dropout_prob > 0.0:
            masks = dropout_masks(self.net, seed, self.step_count, n)
        self.opt.zero_grad()
        z = self.lat(indices)                                     # in-place max-norm renorm of the looked-up rows, then gather
        pred = torch.clamp(self.forward(torch.cat([z, xyz], 1), training, masks), -delta, delta)
        gt = torch.clamp(sdf_gt.reshape(-1, 1), -delta, delta)
        loss = F.l1_loss(pred, gt, reduction="sum") / n
        if code_reg:
            loss = loss + code_reg_lambda * min(1.0, epoch / 100) * torch.sum(torch.norm(z, dim=1)) / n
        loss.backward()
        self.opt.step()
        self.step_count += 1
        return float(loss.detach())
