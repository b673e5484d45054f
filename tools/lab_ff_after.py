"""Lab: which predecessor makes the forward's FIRST epilogue slow?  Dumps the s_memtime stamps (DSDF_LAB library +
DSDF_LAB_DBG) of the LAST forward launched by each variant.  usage: python tools/lab_ff_after.py A|B|C|D"""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
var = sys.argv[1]
dev = torch.device("cuda", 0)
spec = NetSpec(bench.L, **bench.NET)
eng = Engine(spec, dev)
eng.init_like_reference(torch.Generator().manual_seed(0))
lat = (torch.randn(64, bench.L) / math.sqrt(bench.L)).to(dev)
dlat, lat_m, lat_v = torch.zeros_like(lat), torch.zeros_like(lat), torch.zeros_like(lat)
b = bench.synth_batches(1, 0, 64, dev, 1000)[0]
x = torch.randn(16384, 259, device=dev) * 0.1
def tstep():
    eng.train_step(lat, dlat, lat_m, lat_v, b["seg_scene"], b["seg_offset"], b["xyz"], b["gt"], n_norm=16384, clamp_dist=0.1,
                   reg_coef=1e-6, code_bound=1.0, lr_decoder=5e-4, lr_latent=1e-3, training=True, seed=0, seg_len=bench.SAMPLES)
def mfwd():
    eng.module_forward(x, True, seed=1, step=1)
if var == "A":
    for _ in range(4): tstep()
elif var == "B":
    for _ in range(3): tstep()
    mfwd()
elif var == "C":
    for _ in range(4): mfwd()
elif var == "D":
    for _ in range(3): tstep()
    torch.cuda.synchronize(); time.sleep(0.05)
    mfwd()
elif var == "E":   # training forward after an idle gap
    for _ in range(3): tstep()
    torch.cuda.synchronize(); time.sleep(0.05)
    tstep()
if var == "F":   # second module forward after training steps
    for _ in range(3): tstep()
    mfwd(); mfwd()
elif var == "G":   # module forwards only, but inside the (large) training workspace
    eng.train_workspace(16384, 64)
    for _ in range(4): mfwd()
elif var == "H":   # module forward / backward pairs (fused backward + dW + finalize), stamps of the last forward
    dy = torch.full((16384, 1), 1e-4, device=dev)
    for _ in range(3):
        mfwd(); eng.module_backward(dy, 16384, True, False, False)
    mfwd()
elif var == "I":   # module forward, optimiser + weight re-materialisation in between
    for _ in range(3):
        mfwd(); eng.adam_step(lat, dlat, lat_m, lat_v, 5e-4, 1e-3)
    mfwd()
torch.cuda.synchronize()
