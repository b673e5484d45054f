"""Lab: per-parameter gradient error of one config-5 training step against the oracle's bf16-forward step."""
import os, sys, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from oracle import deepsdf_oracle as orc
from tests.golden_io import rel_err
from tests.hip_helpers import HipTrainer, spec_from_meta
from tests.test_gpu_parity import BIG
L, B, S = 256, 64, 256
kw = dict(BIG)
if os.environ.get("LAB_NODROP") == "1": kw["dropout"] = []
net = orc.make_net(L, forward_bf16=True, **kw)
spec = spec_from_meta(dict(L=L, net_specs=dict(kw, forward_bf16=True)))
params = orc.init_params(net, 5)
lat0 = torch.randn(B, L, generator=torch.Generator().manual_seed(6)) / math.sqrt(L)
gen = torch.Generator().manual_seed(1)
idx = torch.arange(B).repeat_interleave(S)
xyz = torch.rand(B * S, 3, generator=gen) * 2 - 1
gt = (xyz.norm(dim=1, keepdim=True) - 0.5) * 0.1
st = orc.TrainState.create({k: v.clone() for k, v in params.items()}, lat0.clone())
ro = orc.train_step(net, st, idx, xyz, gt, delta=0.1, code_bound=1.0, epoch=57, seed=4242)
tr = HipTrainer(spec, params, lat0)
rh = tr.step(idx, xyz, gt, delta=0.1, code_bound=1.0, code_reg=True, lam=1e-4, epoch=57, lr=(5e-4, 1e-3), seed=4242)
print("loss", rh["loss"], ro["loss"])
for k in ro["grads"]:
    print(f"{k:40s} {rel_err(rh['grads'][k], ro['grads'][k]):.3e}")
print("dlat", rel_err(rh["dlat"], ro["dlat"]))
