"""Lab: error of the bf16 forward (decode / decode_latent) against the oracle's bf16 emulation for ragged n."""
import os, sys, math
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R)
import torch
from tests.golden_io import Golden
from tests.hip_helpers import spec_from_meta
from oracle import deepsdf_oracle as orc
from deepsdf_amd.engine import Engine
g = Golden("g8_eval_8x512")
L = g.meta["L"]
params = orc.init_params(orc.make_net(L, **g.meta["net_specs"]), g.meta["seed"])
netb = orc.make_net(L, forward_bf16=True, **g.meta["net_specs"])
engb = Engine(spec_from_meta(dict(L=L, net_specs=dict(g.meta["net_specs"], forward_bf16=True))))
engb.load_params(params)
gen = torch.Generator().manual_seed(3)
z = torch.randn(L, generator=gen) / math.sqrt(L)
for n in (1, 63, 64, 1000, 4096, 70001):
    xyz = torch.rand(n, 3, generator=gen) * 2 - 1
    x = torch.cat([z.expand(n, -1), xyz], 1)
    yob = orc.decoder_forward(netb, params, x, training=False)[0].reshape(-1)
    ylb = engb.decode_latent(z.cuda(), xyz.cuda()).cpu().reshape(-1)
    ydb = engb.decode(x.cuda()).cpu().reshape(-1)
    sc = yob.abs().max().item()
    el, ed = (ylb - yob).abs(), (ydb - yob).abs()
    print(f"n={n}: scale {sc:.3e}  decode_latent max {el.max().item()/sc:.2e} (row {int(el.argmax())})  decode max {ed.max().item()/sc:.2e} (row {int(ed.argmax())})"
          f"  rows > 1e-4: {int((el/sc > 1e-4).sum())} / {int((ed/sc > 1e-4).sum())}", flush=True)
# run-to-run determinism + where the outliers sit inside their workgroup
n = 70001
xyz = torch.rand(n, 3, generator=gen) * 2 - 1
x = torch.cat([z.expand(n, -1), xyz], 1)
yob = orc.decoder_forward(netb, params, x, training=False)[0].reshape(-1)
runs = [engb.decode_latent(z.cuda(), xyz.cuda()).cpu().reshape(-1).clone() for _ in range(4)]
for k in range(1, 4): print("run", k, "differs from run 0 in", int((runs[k] != runs[0]).sum()), "rows")
e = (runs[0] - yob).abs() / yob.abs().max()
bad = torch.nonzero(e > 2e-3).reshape(-1)
print("outliers > 2e-3:", [(int(r), int(r) % 64, f"{e[r].item():.1e}") for r in bad[:40]])
