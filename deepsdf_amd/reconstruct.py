"""Latent-only reconstruction (BASELINE config 4): optimise per-shape codes against a FROZEN decoder.

The fork deleted upstream's ``reconstruct.py``; only its helper vestiges remain
(deep_sdf/data.py:66-71,113-139, deep_sdf/workspace.py:137-149).  The semantics here are therefore DEFINED from the
pieces of the training step (oracle.latent_step, golden g7): eval-mode decoder, loss = mean|clamp(f(z,x)) - clamp(t)|
+ l2reg * mean(z^2), torch-Adam on z.  PARITY UNPINNED by reference files; pinned against the reference Decoder +
torch.optim.Adam in tests/golden/g7_latent_only.npz.

MI355X shape: many shapes are reconstructed AT ONCE -- the codes form a small latent table [B, L], every step is
one fused forward + dX-chain (no weight gradients: DsdfLossCfg.frozen_decoder) over B x S points, followed by a
fused Adam on the table.
"""
import ctypes as C

import torch

from . import _lib
from .engine import Engine, _ptr, _stream


def reconstruct(engine: Engine, samples_xyz, samples_sdf, *, num_iterations=800, clamp_dist=0.1, lr=5e-3,
                l2reg=1e-4, init_std=0.01, lr_drop_every=None, generator=None, z0=None, callback=None):
    """samples_xyz [B, S, G], samples_sdf [B, S] (device tensors; the SAME S points are used every iteration unless
    `callback(it)` returns new (xyz, sdf)).  Returns (codes [B, L], loss): `loss` is the engine's DEVICE scalar tensor (shape
    [1], the last iteration's sum over the B per-shape mean losses) -- reading it with float() synchronises, so it is left to
    the caller."""
    B, S, G = samples_xyz.shape
    L = engine.spec.latent_size
    dev = engine.device
    z = (torch.randn(B, L, generator=generator) * init_std).to(dev) if z0 is None else z0.to(dev, torch.float32).clone()
    dz, m, v = torch.zeros_like(z), torch.zeros_like(z), torch.zeros_like(z)
    seg_scene = torch.arange(B, dtype=torch.int64, device=dev)
    seg_off = torch.arange(0, B * S + 1, S, dtype=torch.int64, device=dev)
    xyz = samples_xyz.reshape(B * S, G).contiguous()
    sdf = samples_sdf.reshape(B * S).contiguous()
    lr_drop_every = int(num_iterations / 2) if lr_drop_every is None else lr_drop_every
    loss = None
    for it in range(num_iterations):
        if callback is not None:
            nb = callback(it)
            if nb is not None:
                xyz, sdf = nb[0].reshape(B * S, G).contiguous(), nb[1].reshape(B * S).contiguous()
        # every shape has its OWN mean over its S points: normalise by S, then the B losses are independent
        engine.train_forward_backward(z, dz, seg_scene, seg_off, xyz, sdf, n_norm=S, clamp_dist=clamp_dist, reg_coef=0.0,
                                      code_bound=None, training=False, accumulate=False, seg_len=S, frozen_decoder=True)
        if l2reg:
            dz.add_(z, alpha=2.0 * l2reg / L)           # d/dz of l2reg * mean(z^2), per shape
        cur_lr = lr * (0.1 ** (it // lr_drop_every)) if lr_drop_every > 0 else lr
        cfg = _lib.DsdfAdamCfg(it + 1, 0.0, float(cur_lr), 0.9, 0.999, 1e-8, None)
        _lib.check(engine.lib.dsdf_adam_latent_only(_ptr(z), _ptr(dz), _ptr(m), _ptr(v), z.numel(), C.byref(cfg), _stream()))
        loss = engine.loss
    return z, loss
