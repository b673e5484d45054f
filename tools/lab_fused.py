"""Lab: time the fused forward kernel (decode path = no activation stores unless --train) under ablations.
usage: DSDF_LAB_ABLATE=k python tools/lab_fused.py [N]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
NET = dict(dims=[512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
           weight_norm=True, geom_dimension=3)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 16384
mode = sys.argv[2] if len(sys.argv) > 2 else "module_train"
spec = NetSpec(256, forward_bf16=os.environ.get("LAB_BF16") == "1", **NET)   # LAB_BF16=1: BASELINE config 5 forward
eng = Engine(spec, "cuda")
eng.init_like_reference(torch.Generator().manual_seed(0))
x = torch.randn(N, 259, device="cuda") * 0.1
dirty = torch.empty(int(os.environ.get("LAB_DIRTY_MB", "0")) * 262144, device="cuda")
def run():
    if dirty.numel(): dirty.fill_(1.0)     # leave dirty lines in L2 / MALL in front of the forward
    if mode == "decode":
        eng.decode(x)
    else:
        eng.module_forward(x, mode == "module_train", seed=1, step=1)
for _ in range(5): run()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
reps = 20
e0.record()
for _ in range(reps): run()
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / reps
flops = 2.0 * N * (spec.w_mac - 512)
print(f"ablate={os.environ.get('DSDF_LAB_ABLATE','0')} N={N} mode={mode}: {ms*1e3:.1f} us  {flops/ms/1e9:.1f} TFLOP/s (incl. gather+wn if dirty)")
