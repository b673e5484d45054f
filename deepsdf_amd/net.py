"""Architecture description shared by the host code and the C ABI.

``NetSpec`` takes the reference Decoder's constructor arguments (deep_sdf/networks/deep_sdf_decoder.py:10-23),
does its layer-size arithmetic (:29-48) and produces (a) the ``DsdfNet`` POD of include/dsdf.h and (b) the
flat parameter-arena layout in the reference module's ``named_parameters()`` order, with the reference's
state-dict key names (``lin{i}.bias``, ``lin{i}.parametrizations.weight.original0/1``, ``lin{i}.weight``).
"""
import os
from dataclasses import dataclass
from typing import List, Tuple

from . import _lib


@dataclass(frozen=True)
class ParamInfo:
    name: str            # reference state-dict key (without any "module." prefix)
    shape: Tuple[int, ...]
    offset: int          # floats from the start of the decoder arena
    layer: int
    kind: str            # "bias" | "g" | "v" | "weight" | "ln_w" | "ln_b"

    @property
    def numel(self):
        n = 1
        for s in self.shape:
            n *= s
        return n


class NetSpec:
    """Validated architecture: every constructor variant of the reference Decoder.  xyz_in_all, latent_dropout and
    LayerNorm (norm_layers without weight_norm) -- used by no shipped spec -- run on the layer-by-layer kernels."""

    def __init__(self, latent_size, dims, geom_dimension, dropout=None, dropout_prob=0.0, norm_layers=(),
                 latent_in=(), weight_norm=False, xyz_in_all=None, use_tanh=False, latent_dropout=False, forward_bf16=False,
                 gemm_split=None):
        """forward_bf16 (not a reference key; BASELINE config 5): hidden-layer forward GEMMs on bf16 MFMA with fp32
        accumulation; backward, master weights and Adam stay fp32.
        gemm_split (not a reference key; opt-in, default from the environment variable DSDF_GEMM_SPLIT=1): the fused kernels'
        hidden-layer GEMMs (forward, backward, dW) on the bf16 matrix pipe with every fp32 operand cut into three bf16 terms -- fp32 accuracy (the
        same parity tolerances), 2.7 x the MFMA rate (include/dsdf.h DsdfNet.gemm_split)."""
        norm_layers = tuple(norm_layers or ())
        latent_in = tuple(latent_in or ())
        self.latent_size = int(latent_size)
        self.geom_dimension = int(geom_dimension)
        self.dims = [int(d) for d in dims]
        self.dropout = None if dropout is None else tuple(int(d) for d in dropout)
        self.dropout_prob = float(dropout_prob)
        self.norm_layers = norm_layers
        self.latent_in = latent_in
        self.weight_norm = bool(weight_norm)
        self.use_tanh = bool(use_tanh)
        self.forward_bf16 = bool(forward_bf16)
        self.gemm_split = ((os.environ.get("DSDF_GEMM_SPLIT") == "1" and os.environ.get("DSDF_NO_FUSED") != "1")
                           if gemm_split is None else bool(gemm_split))
        # variants no shipped spec uses; they run on the layer-by-layer kernels (general mode), never with forward_bf16
        self.xyz_in_all = bool(xyz_in_all)
        self.latent_dropout = bool(latent_dropout)
        layer_norm = (not self.weight_norm) and len(norm_layers) > 0        # deep_sdf_decoder.py:60-65
        if self.xyz_in_all and self.geom_dimension > 4:      # the kernels' d/d(xyz) scratch rows hold 4 floats
            raise NotImplementedError("xyz_in_all needs geom_dimension <= 4")
        if (self.xyz_in_all or self.latent_dropout or layer_norm) and self.forward_bf16:
            raise NotImplementedError("forward_bf16 is not available with xyz_in_all / latent_dropout / LayerNorm")
        if self.forward_bf16 and len(self.dims) + 0 in [int(k) for k in (latent_in or ())]:
            # (layer index len(dims) is the output Linear: the bf16 kernels fold it into the last hidden layer's epilogue as a dot
            # product over the activations only)
            raise NotImplementedError("forward_bf16 is not available when latent_in names the output layer")
        if self.gemm_split and (self.xyz_in_all or self.latent_dropout or layer_norm or max(self.dims) > 512
                                or self.latent_size + self.geom_dimension > 512):
            if gemm_split:      # asked for explicitly
                raise NotImplementedError("gemm_split needs widths <= 512 and none of xyz_in_all / latent_dropout / LayerNorm")
            self.gemm_split = False   # the environment default does not apply to nets the split kernels do not cover
        d = [self.latent_size + self.geom_dimension] + self.dims + [1]
        self.n_layers = len(d) - 1
        if self.n_layers > _lib.MAX_LAYERS:
            raise NotImplementedError(f"more than {_lib.MAX_LAYERS} linear layers")
        self.in_dim, self.out_dim = [], []
        for l in range(self.n_layers):
            self.in_dim.append(d[l])
            if (l + 1) in latent_in:
                self.out_dim.append(d[l + 1] - d[0])
            else:     # deep_sdf_decoder.py:45-48: xyz_in_all narrows every hidden Linear that is not followed by the skip concat
                self.out_dim.append(d[l + 1] - (self.geom_dimension if self.xyz_in_all and l != self.n_layers - 1 else 0))
        self.wn = [bool(weight_norm and l in norm_layers) for l in range(self.n_layers)]
        # a bn{l} = nn.LayerNorm(out_dim) module exists for EVERY l in norm_layers when there is no weight norm -- the last
        # Linear's too, which forward never calls (:97-103 is inside `layer < num_layers - 2`)
        self.ln = [bool(layer_norm and l in norm_layers) for l in range(self.n_layers)]
        if any(self.ln[l] and self.out_dim[l] > 2048 for l in range(self.n_layers)):
            raise NotImplementedError("LayerNorm wider than 2048")
        self.skip = [l in latent_in for l in range(self.n_layers)]
        self.drop = [bool(self.dropout is not None and l in self.dropout and l < self.n_layers - 1)
                     for l in range(self.n_layers)]
        # parameter arena, named_parameters() order
        self.params: List[ParamInfo] = []
        off = 0
        for l in range(self.n_layers):
            o, i = self.out_dim[l], self.in_dim[l]
            if self.wn[l]:
                entries = [(f"lin{l}.bias", (o,), "bias"),
                           (f"lin{l}.parametrizations.weight.original0", (o, 1), "g"),
                           (f"lin{l}.parametrizations.weight.original1", (o, i), "v")]
            else:
                entries = [(f"lin{l}.weight", (o, i), "weight"), (f"lin{l}.bias", (o,), "bias")]
            if self.ln[l]:
                entries += [(f"bn{l}.weight", (o,), "ln_w"), (f"bn{l}.bias", (o,), "ln_b")]
            for name, shape, kind in entries:
                p = ParamInfo(name, shape, off, l, kind)
                self.params.append(p)
                off += p.numel
        self.n_params = off

    def kwargs(self):
        return dict(dims=self.dims, geom_dimension=self.geom_dimension, dropout=self.dropout,
                    dropout_prob=self.dropout_prob, norm_layers=self.norm_layers, latent_in=self.latent_in,
                    weight_norm=self.weight_norm, use_tanh=self.use_tanh, forward_bf16=self.forward_bf16,
                    xyz_in_all=self.xyz_in_all, latent_dropout=self.latent_dropout, gemm_split=self.gemm_split)

    def c_struct(self) -> "_lib.DsdfNet":
        n = _lib.DsdfNet()
        n.n_layers = self.n_layers
        n.latent_size = self.latent_size
        n.geom_dim = self.geom_dimension
        wm = dm = sm = lm = 0
        for l in range(self.n_layers):
            n.in_dim[l] = self.in_dim[l]
            n.out_dim[l] = self.out_dim[l]
            wm |= int(self.wn[l]) << l
            dm |= int(self.drop[l]) << l
            sm |= int(self.skip[l]) << l
            lm |= int(self.ln[l]) << l
        n.weight_norm_mask, n.dropout_mask, n.skip_mask, n.ln_param_mask = wm, dm, sm, lm
        n.dropout_p = self.dropout_prob
        n.fwd_bf16 = int(self.forward_bf16)
        n.use_tanh = int(self.use_tanh)
        n.latent_dropout = int(self.latent_dropout)
        n.xyz_in_all = int(self.xyz_in_all)
        n.gemm_split = int(self.gemm_split)
        return n

    @property
    def w_mac(self):
        """sum over layers of in*out: multiply-accumulates per point per forward pass."""
        return sum(i * o for i, o in zip(self.in_dim, self.out_dim))


# ---- dropout hash keys (host side; spec = oracle/deepsdf_oracle.py dropout_layer_key) -------------------

def _lowbias32(x):
    x &= 0xFFFFFFFF
    x ^= x >> 16
    x = (x * 0x7FEB352D) & 0xFFFFFFFF
    x ^= x >> 15
    x = (x * 0x846CA68B) & 0xFFFFFFFF
    x ^= x >> 16
    return x


def dropout_layer_key(seed: int, step: int, layer: int) -> int:
    seed &= (1 << 64) - 1
    step &= (1 << 64) - 1
    k = _lowbias32((seed & 0xFFFFFFFF) ^ 0x9E3779B9)
    k = _lowbias32(k ^ (seed >> 32))
    k = _lowbias32(k + (step & 0xFFFFFFFF))
    k = _lowbias32(k ^ (step >> 32))
    k = _lowbias32(k + (layer + 1) * 0x9E3779B1)
    return k
