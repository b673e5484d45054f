"""Shared helpers for the -m gpu parity tests: drive deepsdf_amd.Engine (C ABI -> HIP kernels) through a
golden case exactly as oracle.train_step is driven in tests/test_oracle_golden.py."""
import torch

from deepsdf_amd.engine import Engine, make_segments
from deepsdf_amd.net import NetSpec


def spec_from_meta(m):
    return NetSpec(m["L"], **m["net_specs"])


class HipTrainer:
    """Minimal trainer around Engine used by the parity tests (mirrors train_deep_sdf.py:495-545)."""

    def __init__(self, spec, params, latents, device="cuda"):
        self.eng = Engine(spec, device)
        self.eng.load_params(params)
        self.lat = latents.to(device, torch.float32).contiguous().clone()
        self.dlat = torch.zeros_like(self.lat)
        self.lat_m = torch.zeros_like(self.lat)
        self.lat_v = torch.zeros_like(self.lat)
        self.dev = device

    def step(self, idx, xyz, gt, *, delta, code_bound, code_reg, lam, epoch, lr, batch_split=1, grad_clip=None,
             seed=0, training=True, want_y=False, do_adam=True, force_ragged=False):
        N = xyz.shape[0]
        idx, xyz, gt = idx.to(self.dev), xyz.to(self.dev).contiguous(), gt.to(self.dev).reshape(-1).contiguous()
        reg = lam * min(1, epoch / 100) if code_reg else 0.0
        ys, row0 = [], 0
        for ci, (ic, xc, gc) in enumerate(zip(torch.chunk(idx, batch_split), torch.chunk(xyz, batch_split),
                                              torch.chunk(gt, batch_split))):
            sc, so = make_segments(ic)
            lens = (so[1:] - so[:-1]).cpu()
            seg_len = int(lens[0]) if bool((lens == lens[0]).all()) and not force_ragged else 0   # 0: general (ragged) path
            y = torch.empty(xc.shape[0], device=self.dev) if want_y else None
            self.eng.train_forward_backward(self.lat, self.dlat, sc, so, xc.contiguous(), gc.contiguous(), n_norm=N,
                                            clamp_dist=delta, reg_coef=reg, code_bound=code_bound, training=training,
                                            seed=seed, row_offset=row0, accumulate=ci > 0, sdf_out=y, seg_len=seg_len)
            row0 += xc.shape[0]
            if want_y:
                ys.append(y)
        out = dict(loss=float(self.eng.loss.item()), grads={k: v.clone().cpu() for k, v in self.eng.named_views(self.eng.grads).items()},
                   dlat=self.dlat.clone().cpu(), y=torch.cat(ys).cpu() if want_y else None)
        if grad_clip is not None:
            out["grad_norm"] = float(self.eng.grad_norm(grad_clip)[0].item())
        if do_adam:
            self.eng.adam_step(self.lat, self.dlat, self.lat_m, self.lat_v, lr[0], lr[1], clip=grad_clip is not None)
        return out

    def params(self):
        return {k: v.clone().cpu() for k, v in self.eng.named_views().items()}

    def adam_m(self):
        return {k: v.clone().cpu() for k, v in self.eng.named_views(self.eng.exp_avg).items()}

    def adam_v(self):
        return {k: v.clone().cpu() for k, v in self.eng.named_views(self.eng.exp_avg_sq).items()}
