"""Why is the first bench process on a fresh box slow?  Per-step wall time (with a sync) of the first 120 steps."""
import os, sys, time, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
t_import = time.time()
import torch, bench
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
print("import s:", round(time.time() - t_import, 2))
dev = torch.device("cuda", 0)
eng = Engine(NetSpec(bench.L, **bench.NET), dev)
eng.init_like_reference(torch.Generator().manual_seed(0))
lat = (torch.randn(64, bench.L) / math.sqrt(bench.L)).to(dev)
dlat, m, v = torch.zeros_like(lat), torch.zeros_like(lat), torch.zeros_like(lat)
b = bench.synth_batches(1, 0, 64, dev, 1000)[0]
ts = []
for i in range(120):
    t0 = time.perf_counter()
    eng.train_step(lat, dlat, m, v, b["seg_scene"], b["seg_offset"], b["xyz"], b["gt"], n_norm=16384, clamp_dist=0.1, reg_coef=1e-6,
                   code_bound=1.0, lr_decoder=5e-4, lr_latent=1e-3, training=True, seed=0, seg_len=bench.SAMPLES)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print("ms/step: first 5", [round(x, 2) for x in ts[:5]], "| 5-20 avg %.3f | 20-60 avg %.3f | 60-120 avg %.3f | max after 5: %.2f" % (
    sum(ts[5:20]) / 15, sum(ts[20:60]) / 40, sum(ts[60:]) / 60, max(ts[5:])))
