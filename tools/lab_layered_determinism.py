"""Lab: are the layer-by-layer GEMM kernels (gemm_nt / gemm_tn, launch bounds (256, 2): two waves per SIMD) bit-reproducible, and do they
agree element-wise with the fused path?  (The 8-wave bf16 kernel's first k-loop was not: DESIGN.md 4.2.)"""
import os, sys, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
os.environ["DSDF_NO_FUSED"] = sys.argv[1] if len(sys.argv) > 1 else "1"
import torch
from oracle import deepsdf_oracle as orc
from tests.golden_io import Golden
from tests.hip_helpers import HipTrainer, spec_from_meta
from tests.test_gpu_parity import BIG
from deepsdf_amd.engine import Engine
g = Golden("g8_eval_8x512")
L = g.meta["L"]
params = orc.init_params(orc.make_net(L, **g.meta["net_specs"]), g.meta["seed"])
eng = Engine(spec_from_meta(g.meta)); eng.load_params(params)
gen = torch.Generator().manual_seed(3)
n = 70001
x = torch.cat([(torch.randn(L, generator=gen) / math.sqrt(L)).expand(n, -1), torch.rand(n, 3, generator=gen) * 2 - 1], 1).cuda()
runs = [eng.decode(x).cpu().reshape(-1).clone() for _ in range(6)]
print("decode: rows differing from run 0:", [int((r != runs[0]).sum()) for r in runs[1:]])
yo = orc.decoder_forward(orc.make_net(L, **g.meta["net_specs"]), params, x.cpu(), training=False)[0].reshape(-1)
print("worst row vs oracle: %.2e of the range" % float((runs[0] - yo).abs().max() / yo.abs().max()))
# one training step twice from the same state (forward GEMMs, backward GEMMs nt + tn)
Lb, B, S = 256, 64, 256
net = orc.make_net(Lb, **BIG); spec = spec_from_meta(dict(L=Lb, net_specs=BIG)); p2 = orc.init_params(net, 5)
lat0 = torch.randn(B, Lb, generator=torch.Generator().manual_seed(6)) / math.sqrt(Lb)
idx = torch.arange(B).repeat_interleave(S); xyz = torch.rand(B * S, 3, generator=gen) * 2 - 1; gt = (xyz.norm(dim=1, keepdim=True) - 0.5) * 0.1
outs = []
for k in range(4):
    tr = HipTrainer(spec, p2, lat0)
    outs.append(tr.step(idx, xyz, gt, delta=0.1, code_bound=1.0, code_reg=True, lam=1e-4, epoch=57, lr=(5e-4, 1e-3), seed=4242))
for k in range(1, 4):
    print("step run", k, "gradient tensors differing from run 0:", sum(int(not torch.equal(outs[k]["grads"][q], outs[0]["grads"][q])) for q in outs[0]["grads"]),
          "dlat equal:", torch.equal(outs[k]["dlat"], outs[0]["dlat"]))
