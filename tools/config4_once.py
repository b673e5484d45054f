"""BASELINE config 4 as worded -- ONE shape x 8000 points per iteration, frozen 8x512 decoder -- for profiling:
`python tools/pmc.py --script tools/config4_once.py` (SQ MFMA counters per kernel) or under rocprofv3 --kernel-trace --stats."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
from deepsdf_amd.reconstruct import reconstruct
dev = torch.device("cuda", 0)
eng = Engine(NetSpec(bench.L, **bench.NET), dev)
eng.init_like_reference(torch.Generator().manual_seed(0))
S = int(sys.argv[1]) if len(sys.argv) > 1 else 8000
xyz = torch.rand(1, S, 3, device=dev) * 2 - 1
sdf = xyz.norm(dim=2) - 0.5
reconstruct(eng, xyz, sdf, num_iterations=60)
torch.cuda.synchronize()
