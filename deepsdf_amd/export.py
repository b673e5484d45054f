"""TorchScript export of a trained decoder (the reference's create_libtorch_executable.py:4-24 moves the
decoder to the CPU, ``torch.jit.trace``s it and saves ``cpp_model.pt`` for its libtorch consumers).

The HIP ``Decoder`` has no CPU compute path, so tracing it is impossible by construction.  EXPORT therefore goes through a
stock-torch twin built here from the parameter arena: plain ``nn.Linear`` layers with ``parametrizations.weight_norm``,
the same ``state_dict`` keys and shapes as the reference class, eval-mode forward only.  It is export tooling -- nothing
in the training / inference product path imports this module, and it refuses training mode (dropout is not implemented
here on purpose: this is not a CPU fallback).
"""
import torch
import torch.nn as nn

from .net import NetSpec


class StockDecoder(nn.Module):
    """Eval-mode decoder in stock torch ops, for torch.jit.trace / libtorch (deep_sdf_decoder.py:76-111 semantics:
    skip concat before the skip layer, ReLU on hidden layers, optional tanh on the last layer, then the final tanh)."""

    def __init__(self, spec: NetSpec):
        super().__init__()
        if spec.forward_bf16:
            raise NotImplementedError("export is fp32")
        self.spec = spec
        self.geom_dimension = spec.geom_dimension
        self.n_lin = spec.n_layers
        self.skip = [bool((spec.c_struct().skip_mask >> l) & 1) for l in range(spec.n_layers)]
        self.xyz_in = [bool(spec.xyz_in_all and l != 0 and not self.skip[l]) for l in range(spec.n_layers)]
        self.L = spec.latent_size
        self.use_tanh = bool(spec.use_tanh)
        for l in range(spec.n_layers):
            lin = nn.Linear(spec.in_dim[l], spec.out_dim[l])
            if spec.wn[l]:
                lin = nn.utils.parametrizations.weight_norm(lin)      # keys: parametrizations.weight.original0 / original1
            setattr(self, f"lin{l}", lin)
            if spec.ln[l]:
                setattr(self, f"bn{l}", nn.LayerNorm(spec.out_dim[l]))
        self.ln = [bool(spec.ln[l] and l < spec.n_layers - 1) for l in range(spec.n_layers)]   # applied to hidden layers only
        self.th = nn.Tanh()
        super().train(False)

    def train(self, mode=True):
        if mode:
            raise RuntimeError("deepsdf_amd.export.StockDecoder is an eval-mode export twin, not a CPU training path")
        return super().train(False)

    def forward(self, input):
        x = input
        for l in range(self.n_lin):
            if self.skip[l]:
                x = torch.cat([x, input], 1)
            elif self.xyz_in[l]:                  # xyz_in_all (latent_dropout is the identity in eval mode)
                x = torch.cat([x, input[:, self.L:]], 1)
            x = getattr(self, f"lin{l}")(x)
            if l < self.n_lin - 1:
                if self.ln[l]:
                    x = getattr(self, f"bn{l}")(x)
                x = torch.relu(x)
            elif self.use_tanh:
                x = torch.tanh(x)
        return self.th(x)


def to_stock_torch(decoder):
    """deepsdf_amd.Decoder (on any device) -> StockDecoder on the CPU holding a copy of its parameters."""
    twin = StockDecoder(decoder.spec)
    sd = {k: v.detach().to("cpu", torch.float32).clone() for k, v in decoder.state_dict().items()}
    missing = twin.load_state_dict(sd, strict=True)
    assert not missing.missing_keys and not missing.unexpected_keys
    return twin


def export_torchscript(decoder, example_input, path=None):
    """create_libtorch_executable.py:20-24: trace on an example input [1, L+G], script the trace, optionally save."""
    twin = to_stock_torch(decoder)
    ex = example_input.detach().to("cpu", torch.float32)
    with torch.no_grad():
        traced = torch.jit.trace(twin, ex)
        sm = torch.jit.script(traced)
    if path is not None:
        sm.save(path)
    return sm
