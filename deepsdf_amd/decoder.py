"""``Decoder`` -- drop-in for the reference's ``deep_sdf.networks.deep_sdf_decoder.Decoder`` (the plugin seam of
train_deep_sdf.py:275 / deep_sdf/workspace.py:56-58), computing on MI355X through libdsdf_hip.so.

Same constructor, same ``.geom_dimension`` attribute, same ``forward(input[N, L+G]) -> [N, 1]``, same
``state_dict`` keys and shapes (``lin{i}.bias``, ``lin{i}.parametrizations.weight.original0/1``, ``lin{i}.weight``),
so checkpoints move freely between this class and the reference class.  All parameters are views of ONE flat fp32
arena (the layout the HIP kernels and the fused Adam use); gradients land in a second arena.

There is no CPU fallback: ``forward`` on a non-CUDA tensor raises.  ``xyz_in_all``, ``latent_dropout`` and the LayerNorm
variant (norm_layers without weight_norm: modules ``bn{i}``) -- used by no shipped spec -- run on the layer-by-layer kernels.
"""
import warnings

import torch
import torch.nn as nn

from . import _lib
from .engine import Engine
from .net import NetSpec


class _Originals(nn.Module):
    """Holds original0 (g) and original1 (v) like torch's ParametrizationList does."""

    def __init__(self):
        super().__init__()
        self.register_parameter("original0", None)
        self.register_parameter("original1", None)


class _Linear(nn.Module):
    def __init__(self, weight_normed):
        super().__init__()
        if weight_normed:
            self.register_parameter("bias", None)  # first: named_parameters order is bias, original0, original1
            self.parametrizations = nn.ModuleDict({"weight": _Originals()})
        else:
            self.register_parameter("weight", None)
            self.register_parameter("bias", None)


def _check_token(dec, token, what):
    if token != dec._fwd_calls:
        raise RuntimeError(f"deepsdf_amd.Decoder: {what} after a newer forward(): activations live in one "
                           "workspace per module; call it before the next forward")


class _LayerNormParams(nn.Module):
    """Holds bn{l}.weight / bn{l}.bias (the reference's nn.LayerNorm(out_dim), deep_sdf_decoder.py:60-65)."""

    def __init__(self):
        super().__init__()
        self.register_parameter("weight", None)
        self.register_parameter("bias", None)


_warned_second_order = False


class _DecoderBwdFn(torch.autograd.Function):
    """The backward pass as a differentiable function of the incoming gradient: d_input = J^T dy is LINEAR in dy, so its
    own backward is the forward-mode tangent J u (dsdf_module_jvp).  This is what makes the double-backward trick of
    ``torch.autograd.functional.jvp`` work through the decoder (deep_sdf/mesh.py:420).  Parameter gradients are returned
    but not differentiable a second time.

    LIMIT (by design): d_input is differentiable with respect to dy ONLY.  The second-order terms through tanh / the weights --
    what a gradient-penalty or eikonal loss on d sdf / d xyz followed by .backward() would need -- are not produced: the
    reference never asks for them (its only second-order use is the jvp trick above), and autograd gives a Function no way
    to tell "dy's gradient only" from "everything" at backward time, so such a use cannot be refused here either.  Use the
    TorchScript export twin (Decoder.export_torchscript, stock torch ops) for losses of that kind."""

    @staticmethod
    def forward(ctx, dec, dy, token, n, training, need_x):
        _check_token(dec, token, "backward()")
        eng = dec._engine
        d_sdf = dy.reshape(-1).contiguous().to(torch.float32)
        d_in = eng.module_backward(d_sdf, n, training, need_x, accumulate=False)
        grads = tuple(eng.view(eng.grads, p).clone() for p in dec.spec.params)
        ctx.dec, ctx.token, ctx.n, ctx.training = dec, token, n, training
        ctx.mark_non_differentiable(*grads)
        if d_in is None:
            d_in = torch.zeros(0, device=dy.device)             # placeholder output (input needed no gradient)
            ctx.mark_non_differentiable(d_in)
        return (d_in,) + grads

    @staticmethod
    def backward(ctx, g_d_in, *g_params):
        _check_token(ctx.dec, ctx.token, "double backward")
        if g_d_in is None:
            return (None,) * 6
        ju = ctx.dec._engine.module_jvp(g_d_in, ctx.n, ctx.training)    # d <u, J^T dy> / d dy = J u
        return None, ju, None, None, None, None


class _DecoderFn(torch.autograd.Function):
    """Autograd bridge of the module path (dsdf_module_forward / dsdf_module_backward / dsdf_module_jvp)."""

    @staticmethod
    def forward(ctx, dec, x, *params):
        eng = dec._engine_for(x.device)
        training = dec.training
        dec._fwd_calls += 1
        y = eng.module_forward(x, training, seed=dec.dropout_seed, step=dec._fwd_calls)
        ctx.dec, ctx.n, ctx.training, ctx.token = dec, x.shape[0], training, dec._fwd_calls
        ctx.need_x = x.requires_grad
        ctx.n_params = len(params)
        ctx.set_materialize_grads(False)      # forward-mode: parameters without a tangent arrive as None, not as zeros
        return y

    @staticmethod
    def backward(ctx, dy):
        if dy is None:
            return (None,) * (2 + ctx.n_params)
        if torch.is_grad_enabled() and ctx.need_x and not dy.requires_grad and not _warned_second_order:
            # create_graph=True with a constant incoming gradient: the caller is about to differentiate d sdf / d input a second
            # time (gradient penalty, eikonal loss) -- the one thing _DecoderBwdFn does not produce.  (The reference's own
            # second-order use, torch.autograd.functional.jvp in deep_sdf/mesh.py:420, differentiates w.r.t. dy and is served.)
            globals()["_warned_second_order"] = True
            warnings.warn("deepsdf_amd.Decoder: backward(create_graph=True) through the HIP decoder is differentiable with respect to "
                          "the incoming gradient only (the jvp double-backward trick); second-order terms through the input / the "
                          "weights are NOT produced and would silently be missing from a later backward.  Use "
                          "Decoder.export_torchscript() (stock torch ops) for gradient-penalty / eikonal losses.", stacklevel=2)
        out = _DecoderBwdFn.apply(ctx.dec, dy, ctx.token, ctx.n, ctx.training, ctx.need_x)
        return (None, out[0] if ctx.need_x else None) + tuple(out[1:])

    @staticmethod
    def jvp(ctx, _dec, x_t, *param_ts):
        """Forward-mode AD (torch.autograd.forward_ad) with respect to the INPUT: one tangent pass through the same GEMMs."""
        if any(t is not None for t in param_ts):
            raise NotImplementedError("deepsdf_amd.Decoder: forward-mode tangents of the parameters are not supported")
        _check_token(ctx.dec, ctx.token, "jvp")
        if x_t is None:
            return None
        return ctx.dec._engine.module_jvp(x_t, ctx.n, ctx.training)


class Decoder(nn.Module):
    def __init__(self, latent_size, dims, geom_dimension, dropout=None, dropout_prob=0.0, norm_layers=(), latent_in=(),
                 weight_norm=False, xyz_in_all=None, use_tanh=False, latent_dropout=False, forward_bf16=False, gemm_split=None):
        super().__init__()
        self.spec = NetSpec(latent_size, dims, geom_dimension, dropout=dropout, dropout_prob=dropout_prob,
                            norm_layers=norm_layers, latent_in=latent_in, weight_norm=weight_norm,
                            xyz_in_all=xyz_in_all, use_tanh=use_tanh, latent_dropout=latent_dropout, forward_bf16=forward_bf16,
                            gemm_split=gemm_split)
        s = self.spec
        self.num_layers = s.n_layers + 1
        self.geom_dimension = s.geom_dimension
        self.norm_layers, self.latent_in, self.weight_norm = s.norm_layers, s.latent_in, s.weight_norm
        self.dropout, self.dropout_prob, self.use_tanh = s.dropout, s.dropout_prob, s.use_tanh
        self.xyz_in_all, self.latent_dropout = s.xyz_in_all, s.latent_dropout
        self.dropout_seed = int(torch.initial_seed() & 0x7FFFFFFFFFFFFFFF)
        self._fwd_calls = 0
        self._engine = None
        for l in range(s.n_layers):
            setattr(self, f"lin{l}", _Linear(s.wn[l]))
            if s.ln[l]:
                setattr(self, f"bn{l}", _LayerNormParams())
        arena = torch.zeros(s.n_params, dtype=torch.float32)
        self._bind(arena)
        self._init_parameters()

    # ---- arena <-> nn.Parameter plumbing ------------------------------------------------------------------
    def _slot(self, p):
        lin = getattr(self, f"lin{p.layer}")
        if p.kind == "g":
            return lin.parametrizations["weight"], "original0"
        if p.kind == "v":
            return lin.parametrizations["weight"], "original1"
        if p.kind in ("ln_w", "ln_b"):
            return getattr(self, f"bn{p.layer}"), "weight" if p.kind == "ln_w" else "bias"
        return lin, p.kind

    def _bind(self, arena):
        """(Re)create every nn.Parameter as a view of `arena`."""
        object.__setattr__(self, "_arena", arena)
        object.__setattr__(self, "_grad_arena", None)
        for p in self.spec.params:
            mod, attr = self._slot(p)
            old = mod._parameters.get(attr)
            view = arena[p.offset:p.offset + p.numel].view(p.shape)
            if old is None:
                mod._parameters[attr] = nn.Parameter(view, requires_grad=True)
            else:                      # keep Parameter identity (optimizers hold references), like nn.Module._apply
                old.data = view
                old.grad = None
        self._engine = None

    def _init_parameters(self):
        import math
        with torch.no_grad():
            for l in range(self.spec.n_layers):
                o, i = self.spec.out_dim[l], self.spec.in_dim[l]
                w = torch.empty(o, i)
                nn.init.kaiming_uniform_(w, a=math.sqrt(5))          # nn.Linear.reset_parameters
                b = torch.empty(o).uniform_(-1 / math.sqrt(i), 1 / math.sqrt(i))
                for p in self.spec.params:
                    if p.layer == l:
                        src = {"bias": b, "v": w, "weight": w, "g": w.norm(dim=1, keepdim=True), "ln_w": torch.ones(o),
                               "ln_b": torch.zeros(o)}[p.kind]                     # nn.LayerNorm: weight 1, bias 0
                        self._arena[p.offset:p.offset + p.numel].view(p.shape).copy_(src)

    def _apply(self, fn, recurse=True):
        new = fn(self._arena)
        if new.dtype != torch.float32:
            raise TypeError("deepsdf_amd.Decoder is fp32 only")
        if new is not self._arena:     # .cuda()/.to() on a module already in place must not drop the engine
            self._bind(new.contiguous())
        return self

    def _engine_for(self, device):
        if device.type != "cuda":
            raise _lib.DsdfError("deepsdf_amd.Decoder.forward needs CUDA tensors: the HIP path has no CPU fallback")
        if self._arena.device != device:
            raise RuntimeError(f"Decoder parameters are on {self._arena.device}, input on {device}")
        if self._engine is None:
            object.__setattr__(self, "_grad_arena", torch.zeros_like(self._arena))
            self._engine = Engine(self.spec, device, params=self._arena, grads=self._grad_arena)
        return self._engine

    def engine(self):
        """The Engine bound to this module's arenas (fused training path); the module must be on a GPU."""
        return self._engine_for(self._arena.device)

    def load_state_dict(self, state_dict, strict=True, assign=False):
        out = super().load_state_dict(state_dict, strict=strict, assign=False)
        if self._engine is not None:
            self._engine.weights_dirty = True
        return out

    def export_torchscript(self, example_input, path=None):
        """TorchScript module of this decoder for libtorch consumers (create_libtorch_executable.py:4-24), built from a
        stock-torch eval-mode twin on the CPU (deepsdf_amd/export.py) -- export tooling, not a compute path."""
        from .export import export_torchscript
        return export_torchscript(self, example_input, path)

    def jvp(self, input, tangent):
        """(sdf, J . tangent) for input, tangent [N, L+G]: the forward-mode derivative of forward() w.r.t. its input in
        the module's current train/eval mode, one tangent pass after the primal pass (no double backward needed)."""
        if input.shape != tangent.shape or input.dim() != 2 or input.shape[1] != self.spec.in_dim[0]:
            raise ValueError(f"expected input and tangent [N, {self.spec.in_dim[0]}]")
        x = input.detach().to(torch.float32)
        if x.stride(1) != 1:
            x = x.contiguous()
        eng = self._engine_for(x.device)
        eng.weights_dirty = True
        self._fwd_calls += 1
        y = eng.module_forward(x, self.training, seed=self.dropout_seed, step=self._fwd_calls)
        return y, eng.module_jvp(tangent.detach(), x.shape[0], self.training)

    # ---- forward ---------------------------------------------------------------------------------------------
    def forward(self, input):
        if input.dim() != 2 or input.shape[1] != self.spec.in_dim[0]:
            raise ValueError(f"expected input [N, {self.spec.in_dim[0]}], got {tuple(input.shape)}")
        x = input.to(torch.float32)
        if x.stride(1) != 1:
            x = x.contiguous()
        eng = self._engine_for(x.device)
        eng.weights_dirty = True   # parameters may have been changed by any optimizer since the last call
        params = tuple(p for p in self.parameters())
        if torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in params)):
            return _DecoderFn.apply(self, x, *params)
        if self.training and self.spec.dropout_prob > 0 and any(self.spec.drop):
            self._fwd_calls += 1
            return eng.module_forward(x, True, seed=self.dropout_seed, step=self._fwd_calls)
        return eng.decode(x)
