"""Lab: the bf16 forward alone in SEGMENT mode (dsdf_decode_latent: one code, n points, no activation stores, no dropout)."""
import os, sys, math
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
NET = dict(dims=[512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
           weight_norm=True, geom_dimension=3)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
for bf in ((True,) if os.environ.get("LAB_ONLY_BF16") == "1" else (True, False)):
    eng = Engine(NetSpec(256, forward_bf16=bf, **NET), "cuda")
    eng.init_like_reference(torch.Generator().manual_seed(0))
    z = torch.randn(256, device="cuda") / 16
    q = torch.rand(n, 3, device="cuda") * 2 - 1
    for _ in range(5): eng.decode_latent(z, q)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50): eng.decode_latent(z, q)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 50 * 1e3
    print(f"decode_latent {'bf16' if bf else 'fp32'} n={n}: {us:.1f} us  ({2.0 * n * 1835520 / us / 1e6:.1f} TFLOP/s algorithmic, incl. the hoist launch)")
