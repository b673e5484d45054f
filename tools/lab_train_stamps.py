"""Lab: per-layer s_memtime stamps of the FORWARD inside real training steps (-DDSDF_LAB library + DSDF_LAB_DBG=file: forward and
backward go out as two launches and the last forward's stamps are dumped).  usage: [DSDF_GEMM_SPLIT=1] python tools/lab_train_stamps.py"""
import math, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
from deepsdf_amd.train import FusedTrainStep
dev = torch.device("cuda", 0)
L, B, S = 256, 64, 256
eng = Engine(NetSpec(L, **bench.NET), dev)
eng.init_like_reference(torch.Generator().manual_seed(0))
lat = (torch.randn(B, L) / math.sqrt(L)).to(dev)
fused = FusedTrainStep(eng, lat, clamp_dist=0.1, code_reg=True, code_reg_lambda=1e-4, code_bound=1.0, grad_clip=None, seed=0)
bs = bench.synth_batches(2, 0, B, dev, 1000, B, S)
for i in range(6):
    b = bs[i % 2]
    fused(b["scenes"], S, b["xyz"], b["gt"], 1, 5e-4, 1e-3, batch_split=1, n_norm=B * S)
torch.cuda.synchronize()
