"""Lab: one small launch of the 8-wave bf16 forward (general + segment mode) against the 4-wave kernel."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
NET = dict(dims=[512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
           weight_norm=True, geom_dimension=3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
eng = Engine(NetSpec(256, forward_bf16=True, **NET), "cuda")
eng.init_like_reference(torch.Generator().manual_seed(0))
z = torch.randn(256, device="cuda") / 16
q = torch.rand(n, 3, device="cuda") * 2 - 1
x = torch.cat([z.expand(n, -1), q], 1).contiguous()
print("launch general", flush=True)
y = eng.decode(x); torch.cuda.synchronize(); print("general ok", y.reshape(-1)[:4].tolist(), flush=True)
ys = eng.decode_latent(z, q); torch.cuda.synchronize(); print("segment ok", ys.reshape(-1)[:4].tolist(), flush=True)
os.environ["DSDF_BF16_FWD4"] = "1"
y4 = eng.decode(x); ys4 = eng.decode_latent(z, q); torch.cuda.synchronize()
print("4-wave", y4.reshape(-1)[:4].tolist())
print("max |d| general %.3e segment %.3e" % ((y - y4).abs().max().item(), (ys - ys4).abs().max().item()))
