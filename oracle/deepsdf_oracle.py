"""CPU oracle for the DeepSDF auto-decoder training step.  TEST INFRASTRUCTURE ONLY.

This file is the checker, never the product: only ``tests/``, ``__graft_entry__.smoke()`` and
``bench.py``'s ``cpu_baseline`` leg may import it.  Nothing under ``deepsdf_amd/`` imports it.

It restates, in explicit tensor algebra (no autograd, no nn.Module), the algorithm of the
reference hot path:

  * layer table              /root/reference/deep_sdf/networks/deep_sdf_decoder.py:29-57
  * decoder forward          /root/reference/deep_sdf/networks/deep_sdf_decoder.py:76-111
  * batch prep / clamp       /root/reference/train_deep_sdf.py:483-501
  * max-norm latent lookup   /root/reference/train_deep_sdf.py:385,509 (torch embedding_renorm_)
  * loss + code regulariser  /root/reference/train_deep_sdf.py:517-531
  * backward                 /root/reference/train_deep_sdf.py:533 (autograd of the above, derived by hand)
  * Adam on both groups      /root/reference/train_deep_sdf.py:400-411,545 (torch/optim/adam.py single-tensor math)
  * LR schedules             /root/reference/train_deep_sdf.py:23-56
  * grad clipping            /root/reference/train_deep_sdf.py:541-543 (torch clip_grad_norm_)

The arithmetic of those call sites lives in PyTorch (un-vendored by the reference); the pin is this
container's torch 2.10.0.  PARITY PIN: every function here is checked against the reference itself
(imported by file path in the authoring container) by ``tests/golden/make_golden.py``; the resulting
vectors are committed under ``tests/golden/*.npz`` and re-checked by ``tests/test_oracle_golden.py``.
The reference owns no tests for this path, so the goldens are the pin.

Dropout: the reference draws masks from torch's Philox stream, which no other implementation can
reproduce.  The HIP path uses a stateless counter hash (``dropout_keep``) defined HERE as the
specification; the goldens were produced by injecting these masks into the reference's F.dropout.

All functions take/return torch CPU tensors; ``dtype`` may be float32 (parity) or float64 (truth for
tolerance budgeting).
"""
from __future__ import annotations

import math
from dataclasses import dataclass, field
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch

# --------------------------------------------------------------------------------------------
# layer table (deep_sdf_decoder.py:29-57)
# --------------------------------------------------------------------------------------------


@dataclass
class Layer:
    in_dim: int
    out_dim: int
    weight_norm: bool
    skip_in: bool  # input of this layer is [x || x0]   (deep_sdf_decoder.py:88-89)
    dropout: bool  # dropout applied to this layer's activated output (deep_sdf_decoder.py:105-106)
    xyz_in: bool = False      # input of this layer is [x || xyz]  (xyz_in_all, deep_sdf_decoder.py:90-91)
    layer_norm: bool = False  # nn.LayerNorm(out_dim) between the Linear and the ReLU (deep_sdf_decoder.py:60-65,97-103)
    has_ln_params: bool = False   # a bn{l} module exists (it does for every l in norm_layers, even where forward never calls it)


@dataclass
class Net:
    latent_size: int
    geom_dimension: int
    layers: List[Layer]
    dropout_prob: float
    use_tanh: bool
    forward_bf16: bool = False     # BASELINE config 5 (not a reference option): see decoder_forward
    latent_dropout: bool = False   # F.dropout(latent, 0.2) on layer 0's input only (deep_sdf_decoder.py:79-82)

    @property
    def n_lin(self) -> int:
        return len(self.layers)


def make_net(latent_size, dims, geom_dimension, dropout=None, dropout_prob=0.0, norm_layers=(),
             latent_in=(), weight_norm=False, xyz_in_all=None, use_tanh=False,
             latent_dropout=False, forward_bf16=False) -> Net:
    """Same constructor signature as the reference Decoder (deep_sdf_decoder.py:10-23) (+ forward_bf16, config 5)."""
    d = [latent_size + geom_dimension] + list(dims) + [1]
    n_lin = len(d) - 1
    norm_layers = tuple(norm_layers or ())
    latent_in = tuple(latent_in or ())
    layers = []
    for l in range(n_lin):
        if (l + 1) in latent_in:
            out_dim = d[l + 1] - d[0]
        else:
            out_dim = d[l + 1]
            if xyz_in_all and l != n_lin - 1:          # deep_sdf_decoder.py:45-48 (the last Linear keeps its width 1)
                out_dim -= geom_dimension
        ln_mod = bool((not weight_norm) and l in norm_layers)
        layers.append(Layer(
            in_dim=d[l], out_dim=out_dim,
            weight_norm=bool(weight_norm and l in norm_layers),
            skip_in=l in latent_in,
            dropout=bool(dropout is not None and l in dropout and l < n_lin - 1),
            xyz_in=bool(xyz_in_all and l != 0 and l not in latent_in),
            layer_norm=bool(ln_mod and l < n_lin - 1), has_ln_params=ln_mod,
        ))
    return Net(latent_size, geom_dimension, layers, float(dropout_prob), bool(use_tanh), bool(forward_bf16), bool(latent_dropout))


def param_names(net: Net) -> List[str]:
    """named_parameters() order of the reference module (bias, original0, original1 | weight, bias)."""
    names = []
    for l, ly in enumerate(net.layers):
        if ly.weight_norm:
            names += [f"lin{l}.bias", f"lin{l}.parametrizations.weight.original0",
                      f"lin{l}.parametrizations.weight.original1"]
        else:
            names += [f"lin{l}.weight", f"lin{l}.bias"]
        if ly.has_ln_params:
            names += [f"bn{l}.weight", f"bn{l}.bias"]
    return names


def init_params(net: Net, seed: int, dtype=torch.float32) -> Dict[str, torch.Tensor]:
    """nn.Linear default init (U(+-1/sqrt(in)) for weight and bias), g = row norms of v."""
    gen = torch.Generator().manual_seed(seed)
    p = {}
    for l, ly in enumerate(net.layers):
        bound = 1.0 / math.sqrt(ly.in_dim)
        w = (torch.rand(ly.out_dim, ly.in_dim, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        b = (torch.rand(ly.out_dim, generator=gen, dtype=torch.float64) * 2 - 1) * bound
        if ly.weight_norm:
            p[f"lin{l}.bias"] = b.to(dtype)
            p[f"lin{l}.parametrizations.weight.original0"] = w.norm(dim=1, keepdim=True).to(dtype)
            p[f"lin{l}.parametrizations.weight.original1"] = w.to(dtype)
        else:
            p[f"lin{l}.weight"] = w.to(dtype)
            p[f"lin{l}.bias"] = b.to(dtype)
        if ly.has_ln_params:      # nn.LayerNorm starts at weight 1, bias 0; perturbed here so the tests see the affine part
            p[f"bn{l}.weight"] = (1.0 + 0.2 * (torch.rand(ly.out_dim, generator=gen, dtype=torch.float64) - 0.5)).to(dtype)
            p[f"bn{l}.bias"] = (0.1 * (torch.rand(ly.out_dim, generator=gen, dtype=torch.float64) - 0.5)).to(dtype)
    return p


# --------------------------------------------------------------------------------------------
# dropout mask specification (integer, bit-exact between numpy and the HIP epilogue)
# --------------------------------------------------------------------------------------------

_M32 = np.uint64(0xFFFFFFFF)


def _lowbias32(x):
    """32-bit avalanche mix (xorshift-multiply).  x: numpy uint32 array or python int."""
    x = np.asarray(x, dtype=np.uint64) & _M32
    x ^= x >> np.uint64(16)
    x = (x * np.uint64(0x7FEB352D)) & _M32
    x ^= x >> np.uint64(15)
    x = (x * np.uint64(0x846CA68B)) & _M32
    x ^= x >> np.uint64(16)
    return x


def dropout_layer_key(seed: int, step: int, layer: int) -> int:
    """Per-(seed, step, layer) 32-bit key, computed on the host and passed to the kernels."""
    seed &= (1 << 64) - 1
    step &= (1 << 64) - 1
    k = int(_lowbias32((seed & 0xFFFFFFFF) ^ 0x9E3779B9))
    k = int(_lowbias32(k ^ (seed >> 32)))
    k = int(_lowbias32((k + (step & 0xFFFFFFFF)) & 0xFFFFFFFF))
    k = int(_lowbias32(k ^ (step >> 32)))
    k = int(_lowbias32((k + (layer + 1) * 0x9E3779B1) & 0xFFFFFFFF))
    return k


def dropout_threshold16(p: float) -> int:
    """keep iff 16-bit hash field >= threshold; P(drop) = thr/65536 (|p - thr/65536| <= 2^-17)."""
    return int(min(65535, max(0, int(round(p * 65536.0)))))


def dropout_keep(key: int, n_rows: int, n_cols: int, p: float, row_offset: int = 0) -> np.ndarray:
    """keep[row, col] (bool).  One 32-bit hash serves the two rows 2q and 2q+1 of a column:
         ck   = lowbias32(col * 0x85EBCA77 + key)
         h    = lowbias32(ck ^ ((grow >> 1) * 0x9E3779B1)),   grow = row_offset + row
         bits = (grow & 1) ? h >> 16 : h & 0xFFFF
         keep = bits >= dropout_threshold16(p)
    """
    thr = dropout_threshold16(p)
    cols = np.arange(n_cols, dtype=np.uint64)
    ck = _lowbias32((cols * np.uint64(0x85EBCA77) + np.uint64(key)) & _M32)  # [C]
    grow = np.arange(n_rows, dtype=np.uint64) + np.uint64(row_offset)
    pr = ((grow >> np.uint64(1)) * np.uint64(0x9E3779B1)) & _M32  # [R]
    h = _lowbias32(ck[None, :] ^ pr[:, None])
    bits = np.where((grow & np.uint64(1))[:, None] == 1, h >> np.uint64(16), h & np.uint64(0xFFFF))
    return bits >= np.uint64(thr)


def dropout_masks(net: Net, seed: int, step: int, n_rows: int, row_offset: int = 0) -> List[Optional[torch.Tensor]]:
    """One bool mask [n_rows, out_dim] per layer that has dropout, else None."""
    out = []
    for l, ly in enumerate(net.layers):
        if ly.dropout and net.dropout_prob > 0.0:
            k = dropout_layer_key(seed, step, l)
            out.append(torch.from_numpy(dropout_keep(k, n_rows, ly.out_dim, net.dropout_prob, row_offset)))
        else:
            out.append(None)
    return out


# --------------------------------------------------------------------------------------------
# latent lookup with max_norm (torch embedding_renorm_: functional.py:2446-2450,2555-2566)
# --------------------------------------------------------------------------------------------


def renorm_rows_(table: torch.Tensor, indices: torch.Tensor, max_norm: Optional[float]) -> None:
    """In place, no grad: every looked-up row with ||row||_2 > max_norm is scaled by max_norm/(norm+1e-7)."""
    if max_norm is None:
        return
    for j in torch.unique(indices).tolist():
        nu = table[j].norm(2)
        if nu > max_norm:
            table[j] *= max_norm / (nu + 1e-7)


# --------------------------------------------------------------------------------------------
# weight norm (torch._weight_norm(v, g, 0)), Appendix A.3
# --------------------------------------------------------------------------------------------


def weight_norm(g: torch.Tensor, v: torch.Tensor) -> torch.Tensor:
    return v * (g / v.norm(dim=1, keepdim=True))


def weight_norm_backward(dW, g, v):
    nrm = v.norm(dim=1, keepdim=True)
    dg = (dW * v).sum(dim=1, keepdim=True) / nrm
    dv = (g / nrm) * dW - (g * dg / (nrm * nrm)) * v
    return dg, dv


def effective_weights(net: Net, params) -> List[Tuple[torch.Tensor, torch.Tensor]]:
    out = []
    for l, ly in enumerate(net.layers):
        if ly.weight_norm:
            W = weight_norm(params[f"lin{l}.parametrizations.weight.original0"],
                            params[f"lin{l}.parametrizations.weight.original1"])
        else:
            W = params[f"lin{l}.weight"]
        out.append((W, params[f"lin{l}.bias"]))
    return out


# --------------------------------------------------------------------------------------------
# forward (deep_sdf_decoder.py:76-111), Appendix A.4
# --------------------------------------------------------------------------------------------


@dataclass
class Saved:
    x0: torch.Tensor
    inputs: List[torch.Tensor] = field(default_factory=list)   # in_l  (after skip concat)
    acts: List[torch.Tensor] = field(default_factory=list)     # a_{l+1} (post relu+dropout), hidden layers
    u: Optional[torch.Tensor] = None                           # last linear output [N,1]
    t1: Optional[torch.Tensor] = None                          # tanh(u) if use_tanh
    y: Optional[torch.Tensor] = None                           # network output [N,1]
    min_abs_pre: Optional[torch.Tensor] = None                 # [N] min |hidden pre-activation| (margin checks in tests)
    latent_mask: Optional[torch.Tensor] = None                 # [N, L] keep mask of latent_dropout (training), else None
    xhat: List[Optional[torch.Tensor]] = field(default_factory=list)   # per layer: LayerNorm's normalised pre-activation, or None
    rstd: List[Optional[torch.Tensor]] = field(default_factory=list)


LATENT_DROPOUT_P = 0.2      # nn.Dropout(0.2), deep_sdf_decoder.py:36
LATENT_DROPOUT_LAYER = 15   # dropout_layer_key slot of the latent mask (no Linear layer can have this index: at most 16 layers)
LN_EPS = 1e-5               # nn.LayerNorm default


def latent_dropout_mask(net: Net, seed: int, step: int, n_rows: int, row_offset: int = 0):
    """Keep mask [n_rows, L] of latent_dropout, from the same integer hash as the hidden-layer masks."""
    k = dropout_layer_key(seed, step, LATENT_DROPOUT_LAYER)
    return torch.from_numpy(dropout_keep(k, n_rows, net.latent_size, LATENT_DROPOUT_P, row_offset))


def decoder_forward(net: Net, params, x0: torch.Tensor, training: bool = False,
                    masks: Optional[Sequence[Optional[torch.Tensor]]] = None, track_margin: bool = False,
                    latent_mask: Optional[torch.Tensor] = None):
    """x0: [N, L+G] (latent first, coordinates last).  Returns (y [N,1], Saved)."""
    Wb = effective_weights(net, params)
    sv = Saved(x0=x0)
    L, G = net.latent_size, net.geom_dimension
    xyz = x0[:, L:]
    x = x0
    if net.latent_dropout and training and L > 0:            # deep_sdf_decoder.py:79-82: layer 0 sees the dropped latent,
        if latent_mask is None:                              # the skip layer still concatenates the ORIGINAL input
            raise ValueError("training-mode forward with latent_dropout needs an explicit latent mask")
        sv.latent_mask = latent_mask
        x = torch.cat([x0[:, :L] * latent_mask.to(x0.dtype) * (1.0 / (1.0 - LATENT_DROPOUT_P)), xyz], dim=1)
    n_lin = net.n_lin
    scale = 1.0 / (1.0 - net.dropout_prob) if net.dropout_prob < 1.0 else 0.0
    for l, ly in enumerate(net.layers):
        if ly.skip_in:
            x = torch.cat([x, x0], dim=1)
        elif ly.xyz_in:
            x = torch.cat([x, xyz], dim=1)
        sv.inputs.append(x)
        W, b = Wb[l]
        if net.forward_bf16 and l < n_lin - 1:
            # config 5: the hidden Linear's GEMM takes bf16 inputs (round to nearest even, from fp32) and accumulates wider;
            # the saved (unrounded) input and the fp32 weight are what the backward pass uses
            rb = lambda t: t.float().bfloat16().to(t.dtype)  # noqa: E731
            x = rb(x) @ rb(W).t() + b
        else:
            x = x @ W.t() + b
        xh = rs = None
        if ly.layer_norm:                                    # nn.LayerNorm(out_dim): biased variance, eps 1e-5, affine
            mu = x.mean(dim=1, keepdim=True)
            rs = torch.rsqrt(((x - mu) ** 2).mean(dim=1, keepdim=True) + LN_EPS)
            xh = (x - mu) * rs
            x = xh * params[f"bn{l}.weight"] + params[f"bn{l}.bias"]
        sv.xhat.append(xh)
        sv.rstd.append(rs)
        if l < n_lin - 1:
            if track_margin:
                mn = x.abs().amin(dim=1)
                sv.min_abs_pre = mn if sv.min_abs_pre is None else torch.minimum(sv.min_abs_pre, mn)
            x = torch.clamp_min(x, 0.0)
            if training and ly.dropout and net.dropout_prob > 0.0:
                if masks is None or masks[l] is None:
                    raise ValueError("training-mode forward with dropout needs explicit masks")
                x = x * masks[l].to(x.dtype) * scale
            sv.acts.append(x)
    sv.u = x
    if net.use_tanh:
        sv.t1 = torch.tanh(x)
        x = sv.t1
    sv.y = torch.tanh(x)  # self.th is ALWAYS applied (deep_sdf_decoder.py:108-109)
    return sv.y, sv


# --------------------------------------------------------------------------------------------
# backward, Appendix A.6
# --------------------------------------------------------------------------------------------


def decoder_backward(net: Net, params, sv: Saved, dy: torch.Tensor, training: bool):
    """dy = dLoss/dy [N,1].  Returns (param grads dict, dx0 [N, L+G])."""
    Wb = effective_weights(net, params)
    grads: Dict[str, torch.Tensor] = {}
    n_lin = net.n_lin
    scale = 1.0 / (1.0 - net.dropout_prob) if net.dropout_prob < 1.0 else 0.0
    d = dy * (1.0 - sv.y * sv.y)
    if net.use_tanh:
        d = d * (1.0 - sv.t1 * sv.t1)
    dx0 = torch.zeros_like(sv.x0)
    L = net.latent_size
    dp = d  # gradient w.r.t. the current layer's output BEFORE the ReLU (= after LayerNorm where there is one)
    for l in range(n_lin - 1, -1, -1):
        ly = net.layers[l]
        W, _ = Wb[l]
        inp = sv.inputs[l]
        if ly.layer_norm:                                    # LayerNorm backward: dp is d/dz, z = xhat * gamma + beta
            xh, rs = sv.xhat[l], sv.rstd[l]
            grads[f"bn{l}.weight"] = (dp * xh).sum(dim=0)
            grads[f"bn{l}.bias"] = dp.sum(dim=0)
            dxh = dp * params[f"bn{l}.weight"]
            dp = rs * (dxh - dxh.mean(dim=1, keepdim=True) - xh * (dxh * xh).mean(dim=1, keepdim=True))
        elif ly.has_ln_params:                               # a bn module forward never calls (the last Linear): zero gradient
            grads[f"bn{l}.weight"] = torch.zeros_like(params[f"bn{l}.weight"])
            grads[f"bn{l}.bias"] = torch.zeros_like(params[f"bn{l}.bias"])
        grads[f"lin{l}.bias"] = dp.sum(dim=0)
        dW = dp.t() @ inp
        if ly.weight_norm:
            dg, dv = weight_norm_backward(dW, params[f"lin{l}.parametrizations.weight.original0"],
                                          params[f"lin{l}.parametrizations.weight.original1"])
            grads[f"lin{l}.parametrizations.weight.original0"] = dg
            grads[f"lin{l}.parametrizations.weight.original1"] = dv
        else:
            grads[f"lin{l}.weight"] = dW
        din = dp @ W
        if ly.skip_in:
            k = ly.in_dim - sv.x0.shape[1]
            dx0 = dx0 + din[:, k:]
            din = din[:, :k]
        elif ly.xyz_in:
            k = ly.in_dim - net.geom_dimension
            dx0[:, L:] = dx0[:, L:] + din[:, k:]
            din = din[:, :k]
        if l == 0:
            if sv.latent_mask is not None:                   # layer 0 saw the dropped latent
                din = torch.cat([din[:, :L] * sv.latent_mask.to(din.dtype) * (1.0 / (1.0 - LATENT_DROPOUT_P)), din[:, L:]], dim=1)
            dx0 = dx0 + din
            break
        a = sv.acts[l - 1]           # output of layer l-1 after relu (+dropout)
        prev = net.layers[l - 1]
        s = scale if (training and prev.dropout and net.dropout_prob > 0.0) else 1.0
        # a > 0  <=>  pre-activation > 0 and kept  (mask recovered from the STORED forward value)
        dp = din * (a > 0).to(din.dtype) * s
    return grads, dx0


# --------------------------------------------------------------------------------------------
# loss (train_deep_sdf.py:493,517-531), Appendix A.5
# --------------------------------------------------------------------------------------------


def clamped_l1(y, sdf_gt, delta: float, n_norm: int):
    """Returns (loss scalar, dLoss/dy [N,1]).  sdf_gt is clamped here (train_deep_sdf.py:493)."""
    yh = torch.clamp(y, -delta, delta)
    th = torch.clamp(sdf_gt, -delta, delta)
    diff = yh - th
    loss = diff.abs().sum() / n_norm
    inside = ((y >= -delta) & (y <= delta)).to(y.dtype)  # clamp backward passes at the boundary
    dy = torch.sign(diff) * inside / n_norm
    return loss, dy


def code_regulariser(z, lam: float, epoch: int, n_norm: int):
    """r = lam*min(1,epoch/100)*sum_n ||z_n|| / n_norm ; returns (r, dr/dz [N,L])."""
    c = lam * min(1, epoch / 100) / n_norm
    nrm = z.norm(dim=1, keepdim=True)
    r = c * nrm.sum()
    dz = torch.where(nrm > 0, c * z / nrm, torch.zeros_like(z))
    return r, dz


# --------------------------------------------------------------------------------------------
# Adam (torch/optim/adam.py single-tensor math), Appendix A.7
# --------------------------------------------------------------------------------------------


def adam_update_(p, g, m, v, step: int, lr: float, beta1=0.9, beta2=0.999, eps=1e-8):
    m.lerp_(g, 1.0 - beta1)
    v.mul_(beta2).addcmul_(g, g, value=1.0 - beta2)
    bc1 = 1.0 - beta1 ** step
    bc2 = 1.0 - beta2 ** step
    step_size = lr / bc1
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-step_size)


def clip_grad_norm_(grads: Dict[str, torch.Tensor], max_norm: float) -> torch.Tensor:
    """torch clip_grad_norm_: norm of per-tensor norms, coef = clamp(max/(total+1e-6), max=1)."""
    total = torch.linalg.vector_norm(torch.stack([torch.linalg.vector_norm(g) for g in grads.values()]))
    coef = torch.clamp(max_norm / (total + 1e-6), max=1.0)
    for g in grads.values():
        g.mul_(coef)
    return total


# --------------------------------------------------------------------------------------------
# LR schedules (train_deep_sdf.py:23-93)
# --------------------------------------------------------------------------------------------


def learning_rate(spec: dict, epoch: int) -> float:
    t = spec["Type"]
    if t == "Step":
        return spec["Initial"] * (spec["Factor"] ** (epoch // spec["Interval"]))
    if t == "Warmup":
        if epoch > spec["Length"]:
            return spec["Final"]
        return spec["Initial"] + (spec["Final"] - spec["Initial"]) * epoch / spec["Length"]
    if t == "Constant":
        return spec["Value"]
    raise Exception('no known learning rate schedule of type "{}"'.format(t))


# --------------------------------------------------------------------------------------------
# the whole step (train_deep_sdf.py:481-545)
# --------------------------------------------------------------------------------------------


@dataclass
class TrainState:
    params: Dict[str, torch.Tensor]
    latents: torch.Tensor                      # [S_tot, L]
    m: Dict[str, torch.Tensor]
    v: Dict[str, torch.Tensor]
    m_lat: torch.Tensor
    v_lat: torch.Tensor
    step: int = 0

    @staticmethod
    def create(params, latents):
        return TrainState(
            params=params, latents=latents,
            m={k: torch.zeros_like(t) for k, t in params.items()},
            v={k: torch.zeros_like(t) for k, t in params.items()},
            m_lat=torch.zeros_like(latents), v_lat=torch.zeros_like(latents), step=0)


def step_gradients(net: Net, params, latents, indices, xyz, sdf_gt, *, delta, code_bound,
                   code_reg, code_reg_lambda, epoch, n_norm=None, training=True, masks=None, latent_mask=None):
    """One chunk's forward+backward.  ``indices`` [N] int64 (scene of every point).  Mutates ``latents``
    (renorm) like the reference.  Returns dict(loss, y, grads, dlat [S_tot, L], dx0)."""
    N = xyz.shape[0]
    n_norm = N if n_norm is None else n_norm
    renorm_rows_(latents, indices, code_bound)
    z = latents[indices]
    x0 = torch.cat([z, xyz], dim=1)
    y, sv = decoder_forward(net, params, x0, training=training, masks=masks, latent_mask=latent_mask)
    loss, dy = clamped_l1(y, sdf_gt.reshape(-1, 1), delta, n_norm)
    grads, dx0 = decoder_backward(net, params, sv, dy, training)
    L = net.latent_size
    dz = dx0[:, :L].clone()
    if code_reg:
        r, dzr = code_regulariser(z, code_reg_lambda, epoch, n_norm)
        loss = loss + r
        dz = dz + dzr
    dlat = torch.zeros_like(latents)
    dlat.index_add_(0, indices, dz)
    return dict(loss=loss, y=y, grads=grads, dlat=dlat, dx0=dx0)


def train_step(net: Net, st: TrainState, indices, xyz, sdf_gt, *, delta, code_bound, code_reg=True,
               code_reg_lambda=1e-4, epoch=1, lr_decoder=5e-4, lr_latent=1e-3, batch_split=1,
               grad_clip=None, training=True, seed=0, masks_per_chunk=None):
    """Full optimiser step incl. --batch_split accumulation.  Returns dict(loss, grads, dlat, grad_norm, y [N, 1])."""
    N = xyz.shape[0]
    ys = []
    xs, is_, ts = torch.chunk(xyz, batch_split), torch.chunk(indices, batch_split), torch.chunk(sdf_gt, batch_split)
    tot_g = None
    tot_dlat = torch.zeros_like(st.latents)
    loss = 0.0
    row0 = 0
    for ci in range(len(xs)):
        n = xs[ci].shape[0]
        if masks_per_chunk is not None:
            masks = masks_per_chunk[ci]
        elif training and net.dropout_prob > 0:
            masks = dropout_masks(net, seed, st.step, n, row_offset=row0)
        else:
            masks = None
        lmask = latent_dropout_mask(net, seed, st.step, n, row_offset=row0) if (training and net.latent_dropout) else None
        r = step_gradients(net, st.params, st.latents, is_[ci], xs[ci], ts[ci], delta=delta,
                           code_bound=code_bound, code_reg=code_reg, code_reg_lambda=code_reg_lambda,
                           epoch=epoch, n_norm=N, training=training, masks=masks, latent_mask=lmask)
        loss += float(r["loss"])
        ys.append(r["y"])
        tot_dlat += r["dlat"]
        if tot_g is None:
            tot_g = r["grads"]
        else:
            for k in tot_g:
                tot_g[k] = tot_g[k] + r["grads"][k]
        row0 += n
    gnorm = None
    if grad_clip is not None:
        gnorm = clip_grad_norm_(tot_g, grad_clip)
    st.step += 1
    for k in st.params:
        adam_update_(st.params[k], tot_g[k].reshape(st.params[k].shape), st.m[k], st.v[k], st.step, lr_decoder)
    adam_update_(st.latents, tot_dlat, st.m_lat, st.v_lat, st.step, lr_latent)
    return dict(loss=loss, grads=tot_g, dlat=tot_dlat, grad_norm=gnorm, y=torch.cat(ys))


# --------------------------------------------------------------------------------------------
# frozen-decoder latent optimisation (config 4; upstream reconstruct.py is absent from the fork:
# PARITY UNPINNED by reference files -- semantics defined from the pieces above)
# --------------------------------------------------------------------------------------------


def latent_step(net: Net, params, z, m, v, step, xyz, sdf_gt, *, delta, lr, l2reg=1e-4):
    """One Adam iteration on a single code z [1,L] with a frozen eval-mode decoder:
       loss = mean|clamp(f(z,x)) - clamp(gt)| + l2reg*mean(z^2).  Mutates z, m, v."""
    N = xyz.shape[0]
    x0 = torch.cat([z.expand(N, -1), xyz], dim=1)
    y, sv = decoder_forward(net, params, x0, training=False)
    loss, dy = clamped_l1(y, sdf_gt.reshape(-1, 1), delta, N)
    _, dx0 = decoder_backward(net, params, sv, dy, training=False)
    L = net.latent_size
    dz = dx0[:, :L].sum(dim=0, keepdim=True)
    if l2reg:
        loss = loss + l2reg * (z * z).mean()
        dz = dz + l2reg * 2.0 * z / z.numel()
    adam_update_(z, dz, m, v, step, lr)
    return float(loss), dz


# --------------------------------------------------------------------------------------------
# synthetic data of SURVEY 8(d): sphere SDF scenes
# --------------------------------------------------------------------------------------------
# Per-step subsampling: deep_sdf/data.py:74-110 unpack_sdf_samples -- per scene subsample/2
# positive and negative rows WITHOUT replacement (torch.randperm(len)[:n]), a shortfall of one sign taken from the
# other, positives first.  torch's Philox stream cannot be reproduced on the device, so -- like the dropout hash -- the
# specification is an explicit keyed pseudo-random permutation; the HIP kernel (deepsdf_amd/csrc/sample.hpp) is
# bit-exact with this restatement.
# --------------------------------------------------------------------------------------------


def sample_perm(i, length: int, key: int) -> np.ndarray:
    """perm(i) for i in [0, length): 4-round Feistel network on the smallest even-bit domain >= length (at least 2 bits)
    with cycle walking.  i: integer array; returns uint32 indices in [0, length) -- a bijection of range(length)."""
    bits = 2
    while bits < 32 and (1 << bits) < length:
        bits += 1
    bits += bits & 1
    h = np.uint64(bits >> 1)
    mask = np.uint64((1 << (bits >> 1)) - 1)
    x = np.asarray(i, dtype=np.uint64).copy()
    todo = np.ones(x.shape, dtype=bool)
    while todo.any():
        L, R = x[todo] >> h, x[todo] & mask
        for r in range(4):
            F = _lowbias32((R * np.uint64(0x9E3779B1) + np.uint64(key) + np.uint64(r * 0x85EBCA77)) & _M32) & mask
            L, R = R, L ^ F
        x[todo] = (L << h) | R
        todo = x >= np.uint64(length)
    return x.astype(np.uint32)


def sample_key(key64: int, scene: int, sign: int) -> int:
    """32-bit permutation key of (draw key, scene of the cache, sign: 0 positives / 1 negatives)."""
    lo, hi = key64 & 0xFFFFFFFF, (key64 >> 32) & 0xFFFFFFFF
    inner = int(_lowbias32((hi + ((2 * scene + sign) & 0xFFFFFFFF) * 0x9E3779B1) & 0xFFFFFFFF))
    return int(_lowbias32(lo ^ inner))


def sample_rows(n_pos: int, n_neg: int, subsample: int, key64: int, scene: int) -> Tuple[np.ndarray, np.ndarray]:
    """Row indices (into the scene's positives / negatives) of one draw: deep_sdf/data.py:83-101 with the permutation
    above in place of torch.randperm.  len(pos) + len(neg) == 2 * (subsample // 2)."""
    half = int(subsample / 2)
    cp = cn = half
    if n_pos < half:
        cp, cn = n_pos, 2 * half - n_pos
    elif n_neg < half:
        cn, cp = n_neg, 2 * half - n_neg
    return (sample_perm(np.arange(cp), n_pos, sample_key(key64, scene, 0)),
            sample_perm(np.arange(cn), n_neg, sample_key(key64, scene, 1)))


# --------------------------------------------------------------------------------------------


def sphere_scene(k: int, n_points: int = 50000, unit: bool = False):
    """Returns (pos [*,4], neg [*,4]) float32 like sdf_sampler/sdf_sampler.py:146 writes them."""
    rng = np.random.default_rng(1234 + k)
    if unit:
        c, r = np.zeros(3), 1.0
    else:
        c, r = rng.uniform(-0.3, 0.3, 3), rng.uniform(0.3, 0.6)
    h = n_points // 2
    box = rng.uniform(-1, 1, (h, 3))
    d = rng.normal(size=(n_points - h, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    surf = c + r * d + rng.normal(0, 0.05, (n_points - h, 3))
    pts = np.concatenate([box, surf], 0)
    sdf = np.linalg.norm(pts - c, axis=1) - r
    s = np.concatenate([pts, sdf[:, None]], 1).astype(np.float32)
    return s[s[:, 3] >= 0], s[s[:, 3] < 0]
