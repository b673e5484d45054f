"""Non-headline configurations quoted in DESIGN.md: BASELINE config 4 (latent-only reconstruction, 8000 = 125 x 64 pts/iter per
shape: segment mode; batched over shapes) and the locality extreme of config 2 (ONE scene x 16384 points per step)."""
import math, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench
from deepsdf_amd.engine import Engine
from deepsdf_amd.net import NetSpec
from deepsdf_amd.reconstruct import reconstruct
dev = torch.device("cuda", 0)
eng = Engine(NetSpec(bench.L, **bench.NET), dev)
eng.init_like_reference(torch.Generator().manual_seed(0))
# config 4: B shapes x 8000 (= 125 x 64: segment mode) points, frozen decoder, 100 iterations; 8001: the ragged path
for B, S in ((1, 8000), (1, 8001), (16, 8000), (64, 8000)):
    xyz = torch.rand(B, S, 3, device=dev) * 2 - 1
    sdf = xyz.norm(dim=2) - 0.5
    for graph in (False, True) if B == 1 else (False,):
        reconstruct(eng, xyz, sdf, num_iterations=20, graph=graph)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        reconstruct(eng, xyz, sdf, num_iterations=200, graph=graph)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
        print(f"config 4: {B:3d} shapes x {S} pts{' (captured HIP graph, replayed)' if graph else ''}: {dt*1e3:.3f} ms/iteration = "
              f"{B*S/dt/1e6:.2f} M point-samples/s ({B/dt:.0f} shape-iterations/s; {4 * 1835520 * B * S / dt / 1e12:.1f} TFLOP/s = "
              f"{4 * 1835520 * B * S / dt / 157.3e12:.2f} of the fp32-MFMA peak)")
# locality extreme: one scene, 16384 points per step
lat = (torch.randn(1, bench.L) / math.sqrt(bench.L)).to(dev)
dlat, m, v = torch.zeros_like(lat), torch.zeros_like(lat), torch.zeros_like(lat)
xyz = torch.rand(16384, 3, device=dev) * 2 - 1
gt = xyz.norm(dim=1) - 0.5
sc = torch.zeros(1, dtype=torch.int64, device=dev); so = torch.tensor([0, 16384], dtype=torch.int64, device=dev)
def step():
    eng.train_step(lat, dlat, m, v, sc, so, xyz, gt, n_norm=16384, clamp_dist=0.1, reg_coef=1e-6, code_bound=1.0,
                   lr_decoder=5e-4, lr_latent=1e-3, training=True, seed=0, seg_len=16384)
for _ in range(60): step()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(200): step()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 200
print(f"config 2, B=1 x S=16384: {dt*1e3:.3f} ms/step = {16384/dt/1e6:.2f} M point-samples/s")
# the shipped big setting: 64 scenes x 16384 samples = 1,048,576 points per step (256 workgroups per scene)
B, S = 64, 16384
lat = (torch.randn(B, bench.L) / math.sqrt(bench.L)).to(dev)
dlat, m, v = torch.zeros_like(lat), torch.zeros_like(lat), torch.zeros_like(lat)
xyz = torch.rand(B * S, 3, device=dev) * 2 - 1
gt = xyz.norm(dim=1) - 0.5
sc = torch.arange(B, dtype=torch.int64, device=dev); so = torch.arange(0, B * S + 1, S, dtype=torch.int64, device=dev)
def big():
    eng.train_step(lat, dlat, m, v, sc, so, xyz, gt, n_norm=B * S, clamp_dist=0.1, reg_coef=1e-6, code_bound=1.0,
                   lr_decoder=5e-4, lr_latent=1e-3, training=True, seed=0, seg_len=S)
for _ in range(3): big()
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): big()
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f"64 scenes x 16384 samples (1,048,576 pts/step): {dt*1e3:.2f} ms/step = {B*S/dt/1e6:.2f} M point-samples/s; workspace {eng._ws.numel()/2**30:.1f} GiB")
# inference: one code, 1 M query points -- the [n, L+G] input materialised (dsdf_decode) vs hoisted (dsdf_decode_latent)
n = 1 << 20
z = torch.randn(bench.L, device=dev) / math.sqrt(bench.L)
q = torch.rand(n, 3, device=dev) * 2 - 1
def dec_cat():
    return eng.decode(torch.cat([z.expand(n, -1), q], 1))
def dec_lat():
    return eng.decode_latent(z, q)
for name, f in (("decode_sdf as the reference does it (cat + dsdf_decode)", dec_cat), ("dsdf_decode_latent", dec_lat)):
    for _ in range(3): f()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10): f()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f"inference, 1 code x {n} points, {name}: {dt*1e3:.2f} ms = {n/dt/1e6:.1f} M points/s")
# (the multi-GPU call sequence without the collective is measured by `DSDF_FORCE_DP_PATH=1 python bench.py --no-cpu-baseline`)
