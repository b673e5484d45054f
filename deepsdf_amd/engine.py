"""Device-side state of one decoder replica and the thin calls into libdsdf_hip.so.

Everything the kernels touch lives in flat fp32 arenas on one GPU:
  params / grads / exp_avg / exp_avg_sq  [n_params]   reference named_parameters() order (net.NetSpec.params)
  packed                                  W and W^T of every layer in the padded MFMA layout
  workspace                               activations, dP ping-pong, split-K slabs, partials (grown on demand)
PyTorch only provides the memory and the stream.  There is no CPU path: every method needs a CUDA device.
"""
import ctypes as C
import math

import torch

from . import _lib
from .net import NetSpec, dropout_layer_key


def _ptr(t):
    return C.c_void_p(t.data_ptr()) if t is not None else C.c_void_p(0)


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


class Engine:
    def __init__(self, spec: NetSpec, device="cuda", params=None, grads=None):
        self.spec = spec
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _lib.DsdfError("deepsdf_amd.Engine needs a CUDA/HIP device (no CPU fallback)")
        if self.device.index is None:
            self.device = torch.device("cuda", torch.cuda.current_device())
        self.lib = _lib.lib()
        self.cnet = spec.c_struct()
        lay = _lib.DsdfParamLayout()
        _lib.check(self.lib.dsdf_param_layout(C.byref(self.cnet), C.byref(lay)))
        if lay.total != spec.n_params:
            raise _lib.DsdfError(f"param layout mismatch: C {lay.total} vs host {spec.n_params}")
        for p in spec.params:
            off = {"bias": lay.bias_off, "g": lay.g_off, "v": lay.v_off, "weight": lay.v_off, "ln_w": lay.ln_w_off,
                   "ln_b": lay.ln_b_off}[p.kind][p.layer]
            if off != p.offset:
                raise _lib.DsdfError(f"param layout mismatch at {p.name}: C {off} vs host {p.offset}")
        n = C.c_int64()
        _lib.check(self.lib.dsdf_packed_floats(C.byref(self.cnet), C.byref(n)))
        f32 = dict(dtype=torch.float32, device=self.device)
        # the Decoder nn.Module hands in its own arenas (its nn.Parameters are views of them)
        self.params = params if params is not None else torch.zeros(spec.n_params, **f32)
        self.grads = grads if grads is not None else torch.zeros(spec.n_params, **f32)
        for t in (self.params, self.grads):
            if t.device != self.device or t.dtype != torch.float32 or t.numel() != spec.n_params or not t.is_contiguous():
                raise _lib.DsdfError("parameter/gradient arena must be a contiguous fp32 tensor on the engine's device")
        self.exp_avg = torch.zeros(spec.n_params, **f32)
        self.exp_avg_sq = torch.zeros(spec.n_params, **f32)
        self.packed = torch.zeros(n.value, **f32)
        self.loss = torch.zeros(1, **f32)
        self.clip = torch.zeros(2, **f32)   # [norm, coef]
        self._ws = None
        self._ws_sizes = {}
        self.step = 0
        self.weights_dirty = True

    # ---- parameter access ------------------------------------------------------------------------------
    def view(self, arena, p):
        return arena[p.offset:p.offset + p.numel].view(p.shape)

    def named_views(self, arena=None):
        arena = self.params if arena is None else arena
        return {p.name: self.view(arena, p) for p in self.spec.params}

    def load_params(self, state):
        """state: dict reference-key -> tensor (any device); a leading 'module.' is accepted."""
        for p in self.spec.params:
            t = state.get(p.name, state.get("module." + p.name))
            if t is None:
                raise KeyError(f"missing parameter {p.name}")
            if tuple(t.shape) != p.shape:
                raise ValueError(f"shape mismatch for {p.name}: {tuple(t.shape)} vs {p.shape}")
            self.view(self.params, p).copy_(t.to(self.device, torch.float32))
        self.weights_dirty = True

    def init_like_reference(self, generator=None):
        """nn.Linear default init (kaiming_uniform(a=sqrt(5)) == U(+-1/sqrt(in)) for weight and bias), g = ||v||_row
        (parametrizations.weight_norm initialises original0 to the row norms), deep_sdf_decoder.py:50-57."""
        for l in range(self.spec.n_layers):
            o, i = self.spec.out_dim[l], self.spec.in_dim[l]
            bound = 1.0 / math.sqrt(i)
            w = (torch.rand(o, i, generator=generator) * 2 - 1) * bound
            b = (torch.rand(o, generator=generator) * 2 - 1) * bound
            for p in self.spec.params:
                if p.layer != l:
                    continue
                src = {"bias": b, "v": w, "weight": w, "g": w.norm(dim=1, keepdim=True), "ln_w": torch.ones(o),
                       "ln_b": torch.zeros(o)}[p.kind]
                self.view(self.params, p).copy_(src.to(self.device))
        self.weights_dirty = True

    # ---- workspace ----------------------------------------------------------------------------------------
    def _workspace(self, nbytes):
        if self._ws is None or self._ws.numel() < nbytes:
            self._ws = None
            self._ws = torch.empty(int(nbytes), dtype=torch.uint8, device=self.device)
        return self._ws

    def train_workspace(self, n_points, n_segments, buckets=0):
        key = (int(n_points), int(n_segments), int(buckets) if buckets > 2 else 0)
        size = self._ws_sizes.get(key)
        if size is None:     # (the size query plans the whole step -- every bucket's schedule -- on the host: asked once per shape, not
            b = C.c_size_t()  # once per call; a K-bucket step makes K calls)
            if buckets > 2:      # finer split-K slabs than dsdf_workspace_bytes plans for
                _lib.check(self.lib.dsdf_workspace_bytes_buckets(C.byref(self.cnet), n_points, n_segments, buckets, C.byref(b)))
            else:
                _lib.check(self.lib.dsdf_workspace_bytes(C.byref(self.cnet), n_points, n_segments, C.byref(b)))
            size = self._ws_sizes[key] = b.value
        return self._workspace(size)

    # ---- weights -----------------------------------------------------------------------------------------
    def materialize(self):
        _lib.check(self.lib.dsdf_materialize_weights(C.byref(self.cnet), _ptr(self.params), _ptr(self.packed), _stream()))
        self.weights_dirty = False

    def _fresh_weights(self):
        if self.weights_dirty:
            self.materialize()

    # ---- inference ----------------------------------------------------------------------------------------
    def decode(self, inputs, max_chunk=1 << 18):
        """inputs [n, L+G] fp32 cuda -> sdf [n, 1] (eval-mode Decoder.forward / decode_sdf)."""
        self._fresh_weights()
        x = inputs.to(self.device, torch.float32)
        if x.dim() != 2 or x.shape[1] != self.spec.in_dim[0]:
            raise ValueError(f"expected input [n, {self.spec.in_dim[0]}], got {tuple(x.shape)}")
        if x.stride(1) != 1:
            x = x.contiguous()
        n = x.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        b = C.c_size_t()
        _lib.check(self.lib.dsdf_decode_workspace_bytes(C.byref(self.cnet), min(n, max_chunk), C.byref(b)))
        ws = self._workspace(b.value)
        for s in range(0, n, max_chunk):
            e = min(n, s + max_chunk)
            xs = x[s:e]
            _lib.check(self.lib.dsdf_decode(C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(xs),
                                            xs.stride(0), e - s, C.c_void_p(out.data_ptr() + 4 * s), _ptr(ws),
                                            ws.numel(), _stream()))
        return out.view(n, 1)

    def decode_latent_supported(self):
        """Whether decode_latent takes this net (dsdf_decode_latent_supported: the library's own condition)."""
        rc = self.lib.dsdf_decode_latent_supported(C.byref(self.cnet))
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    def decode_latent(self, latent, xyz, max_chunk=1 << 20):
        """decode_sdf with ONE code for every query point: latent [L] (or [1, L]), xyz [n, G] -> sdf [n, 1].  The [n, L+G]
        input is never materialised (dsdf_decode_latent: the latent's products are hoisted out of the per-point work)."""
        self._fresh_weights()
        z = latent.to(self.device, torch.float32).reshape(-1).contiguous()
        x = xyz.to(self.device, torch.float32).contiguous()
        if z.numel() != self.spec.latent_size or x.dim() != 2 or x.shape[1] != self.spec.geom_dimension:
            raise ValueError(f"expected latent [{self.spec.latent_size}] and xyz [n, {self.spec.geom_dimension}]")
        n = x.shape[0]
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        b = C.c_size_t()
        _lib.check(self.lib.dsdf_decode_workspace_bytes(C.byref(self.cnet), min(max(n, 64), max_chunk), C.byref(b)))
        ws = self._workspace(max(b.value, 16384))
        for s in range(0, n, max_chunk):
            e = min(n, s + max_chunk)
            _lib.check(self.lib.dsdf_decode_latent(C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(z),
                                                   C.c_void_p(x.data_ptr() + 4 * s * x.shape[1]), e - s,
                                                   C.c_void_p(out.data_ptr() + 4 * s), _ptr(ws), ws.numel(), _stream()))
        return out.view(n, 1)

    # ---- module path (autograd) -----------------------------------------------------------------------------
    def module_forward(self, x, training, seed=0, step=0):
        self._fresh_weights()
        n = x.shape[0]
        ws = self.train_workspace(n, 0)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        keys = (C.c_uint32 * _lib.MAX_LAYERS)(*[dropout_layer_key(seed, step, l) for l in range(_lib.MAX_LAYERS)])
        self._module_keys = keys          # module_backward of a latent_dropout net needs the same mask
        _lib.check(self.lib.dsdf_module_forward(C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(x),
                                                x.stride(0), n, int(training), keys, _ptr(out), _ptr(ws), ws.numel(),
                                                _stream()))
        return out.view(n, 1)

    def module_backward(self, d_sdf, n, training, need_input_grad, accumulate):
        ws = self.train_workspace(n, 0)
        d_in = torch.empty(n, self.spec.in_dim[0], dtype=torch.float32, device=self.device) if need_input_grad else None
        _lib.check(self.lib.dsdf_module_backward(C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(d_sdf), n,
                                                 int(training), getattr(self, "_module_keys", None), _ptr(self.grads),
                                                 int(accumulate), _ptr(d_in),
                                                 self.spec.in_dim[0], _ptr(ws), ws.numel(), _stream()))
        return d_in

    def module_jvp(self, tangent, n, training):
        """J . tangent at the point of the last module_forward: tangent [n, L+G] -> [n, 1] (dsdf_module_jvp)."""
        t = tangent.to(self.device, torch.float32)
        if t.dim() != 2 or t.shape != (n, self.spec.in_dim[0]):
            raise ValueError(f"expected tangent [{n}, {self.spec.in_dim[0]}], got {tuple(t.shape)}")
        if t.stride(1) != 1:
            t = t.contiguous()
        ws = self.train_workspace(n, 0)
        out = torch.empty(n, dtype=torch.float32, device=self.device)
        _lib.check(self.lib.dsdf_module_jvp(C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(t), t.stride(0), n,
                                            int(training), getattr(self, "_module_keys", None), _ptr(out), _ptr(ws), ws.numel(),
                                            _stream()))
        return out.view(n, 1)

    # ---- training --------------------------------------------------------------------------------------------
    def train_forward_backward(self, latents, dlat, seg_scene, seg_offset, xyz, sdf_gt, *, n_norm, clamp_dist,
                               reg_coef, code_bound, training=True, seed=0, row_offset=0, accumulate=False,
                               sdf_out=None, step=None, seg_len=0, frozen_decoder=False, loss_out=None, dw_phase=0,
                               dw_buckets=None):
        """One chunk of train_deep_sdf.py:509-533.  Gradients land in self.grads / dlat, loss in self.loss (or in the
        caller's 1-element fp32 device tensor `loss_out`, e.g. a slot of a per-epoch loss buffer: no copy kernel per step)."""
        self._fresh_weights()
        n, R = xyz.shape[0], seg_scene.shape[0]
        if dw_buckets is None:
            dw_buckets = 2 if dw_phase else 0
        ws = self.train_workspace(n, R, dw_buckets)
        b = _lib.DsdfBatch(seg_scene.data_ptr(), seg_offset.data_ptr(), R, xyz.data_ptr(), sdf_gt.data_ptr(), n,
                           int(n_norm), int(row_offset), int(seg_len))
        cfg = _lib.DsdfLossCfg()
        cfg.clamp_dist, cfg.reg_coef = float(clamp_dist), float(reg_coef)
        cfg.code_bound = float(code_bound) if code_bound is not None else -1.0
        cfg.training = int(training)
        cfg.frozen_decoder = int(frozen_decoder)
        cfg.dw_phase, cfg.dw_buckets = int(dw_phase), int(dw_buckets)   # phase p of a K-bucket data-parallel backward (include/dsdf.h)
        st = self.step if step is None else step
        for l in range(_lib.MAX_LAYERS):
            cfg.dropout_key[l] = dropout_layer_key(seed, st, l)
        _lib.check(self.lib.dsdf_train_forward_backward(
            C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(latents), latents.shape[0], C.byref(b),
            C.byref(cfg), _ptr(self.grads), _ptr(dlat), _ptr(self.loss if loss_out is None else loss_out), _ptr(sdf_out),
            int(accumulate), _ptr(ws), ws.numel(), _stream()))

    def dw_phase_supported(self):
        """Whether this net's backward can run in phases (dsdf_dw_phase_supported: the library's own condition)."""
        rc = self.lib.dsdf_dw_phase_supported(C.byref(self.cnet))
        if rc < 0:
            _lib.check(rc)
        return rc == 1

    def grad_buckets(self, k):
        """The k gradient buckets of a phased backward, last layers first (dsdf_grad_buckets): (first_layer [k],
        arena_off [k + 1]); bucket b = arena floats [arena_off[b + 1], arena_off[b])."""
        first, off = (C.c_int32 * k)(), (C.c_int64 * (k + 1))()
        _lib.check(self.lib.dsdf_grad_buckets(C.byref(self.cnet), k, first, off))
        return list(first), list(off)

    def grad_bucket_split(self):
        """(first layer of the late bucket, arena offset separating the two gradient buckets) of a TWO-bucket backward."""
        first, off = self.grad_buckets(2)
        return first[0], off[1]

    def train_step(self, latents, dlat, lat_m, lat_v, seg_scene, seg_offset, xyz, sdf_gt, *, n_norm, clamp_dist, reg_coef,
                   code_bound, lr_decoder, lr_latent, training=True, seed=0, seg_len=0, betas=(0.9, 0.999), eps=1e-8,
                   loss_out=None):
        """Whole optimiser step in ONE library call (single process, no --batch_split, no clipping): forward + backward +
        Adam on both groups + weight re-materialisation, with the decoder's Adam folded into the finalize pass."""
        self._fresh_weights()
        n, R = xyz.shape[0], seg_scene.shape[0]
        ws = self.train_workspace(n, R)
        b = _lib.DsdfBatch(seg_scene.data_ptr(), seg_offset.data_ptr(), R, xyz.data_ptr(), sdf_gt.data_ptr(), n,
                           int(n_norm), 0, int(seg_len))
        cfg = _lib.DsdfLossCfg()
        cfg.clamp_dist, cfg.reg_coef = float(clamp_dist), float(reg_coef)
        cfg.code_bound = float(code_bound) if code_bound is not None else -1.0
        cfg.training, cfg.frozen_decoder = int(training), 0
        for l in range(_lib.MAX_LAYERS):
            cfg.dropout_key[l] = dropout_layer_key(seed, self.step, l)
        ad = _lib.DsdfAdamCfg(self.step + 1, float(lr_decoder), float(lr_latent), betas[0], betas[1], eps, None)
        _lib.check(self.lib.dsdf_train_step(
            C.byref(self.cnet), _ptr(self.packed), _ptr(self.params), _ptr(self.grads), _ptr(self.exp_avg),
            _ptr(self.exp_avg_sq), _ptr(latents), latents.shape[0], _ptr(dlat), _ptr(lat_m), _ptr(lat_v), C.byref(b),
            C.byref(cfg), C.byref(ad), _ptr(self.loss if loss_out is None else loss_out), None, _ptr(ws), ws.numel(), _stream()))
        self.step += 1                # only once the call was accepted: a rejected step must not advance Adam's bias correction
        self.weights_dirty = False

    def grad_norm(self, max_norm):
        """clip_grad_norm_ coefficient into self.clip[1] (device); returns the device tensor [norm, coef]."""
        ws = self._workspace(1 << 16)
        _lib.check(self.lib.dsdf_grad_norm(_ptr(self.grads), self.spec.n_params, float(max_norm), _ptr(self.clip),
                                           C.c_void_p(self.clip.data_ptr() + 4), _ptr(ws), ws.numel(), _stream()))
        return self.clip

    def adam_latents(self, latents, dlat, lat_m, lat_v, lr_latent, *, betas=(0.9, 0.999), eps=1e-8):
        """Adam on the latent table alone, as step `self.step + 1` (the data-parallel step runs it under the decoder
        gradient's all-reduce; the following adam_step(None, ...) advances the shared step counter)."""
        cfg = _lib.DsdfAdamCfg(self.step + 1, 0.0, float(lr_latent), betas[0], betas[1], eps, None)
        _lib.check(self.lib.dsdf_adam_latent_only(_ptr(latents), _ptr(dlat), _ptr(lat_m), _ptr(lat_v), latents.numel(),
                                                  C.byref(cfg), _stream()))

    def adam_step(self, latents, dlat, lat_m, lat_v, lr_decoder, lr_latent, *, clip=False, betas=(0.9, 0.999), eps=1e-8):
        """Adam on the decoder arena (+ the latent table unless `latents` is None) and weight re-materialisation."""
        cfg = _lib.DsdfAdamCfg(self.step + 1, float(lr_decoder), float(lr_latent), betas[0], betas[1], eps,
                               (self.clip.data_ptr() + 4) if clip else None)
        nlat = latents.numel() if latents is not None else 0
        _lib.check(self.lib.dsdf_adam_step(C.byref(self.cnet), _ptr(self.params), _ptr(self.grads), _ptr(self.exp_avg),
                                           _ptr(self.exp_avg_sq), _ptr(latents), _ptr(dlat), _ptr(lat_m), _ptr(lat_v),
                                           nlat, C.byref(cfg), _ptr(self.packed), _stream()))
        self.step += 1
        self.weights_dirty = False


def make_segments(indices):
    """Per-point scene indices [N] (the reference's `indices.unsqueeze(-1).repeat(1, S)` layout, possibly chunked)
    -> (seg_scene [R], seg_offset [R+1]) int64 on the same device."""
    scenes, counts = torch.unique_consecutive(indices, return_counts=True)
    off = torch.zeros(scenes.numel() + 1, dtype=torch.int64, device=indices.device)
    off[1:] = torch.cumsum(counts, 0)
    return scenes.to(torch.int64).contiguous(), off
