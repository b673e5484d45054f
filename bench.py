#!/usr/bin/env python3
"""bench.py -- SDF point-samples/sec per training step (BASELINE.json metric) on MI355X.

One "step" = one full optimiser step of the hot path over one synthetic batch resident in HBM:
latent renorm + gather + concat -> 8x512 decoder forward (weight-norm, skip@4, dropout 0.2) -> clamped-L1 + code
regulariser -> backward -> [N>1: RCCL all-reduce of decoder grads, latent Adam under it] -> Adam on decoder + latent table
-> weight re-materialisation.  The step object is the PRODUCT's (deepsdf_amd.train.FusedTrainStep: what train_deep_sdf.py
runs).  Workload at N=1 = BASELINE.json configs[1]; N>1 = configs[2] (512 scenes sharded, weak scaling).

  python bench.py --gpus 1 --steps 50 --warmup 10
  python bench.py --code-length 2 --scenes-per-batch 10 --samples 16000      # the reference's shipped 8x512 experiment shape (not the headline)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

Beside the contract's line (value / ms_per_step over the barrier-bracketed timed region) rank 0 reports, at N=1:
  step_time_ms    median / p10 / p90 of single steps (device time between per-step events on the compute stream)
  roofline        dominant kernel: algorithmic FLOP / HIP-event time; `traffic` (HBM bytes per launch) and `executed_frac`
                  (MFMA FLOPs actually issued, from SQ counters) measured by rocprofv3 --pmc child passes of THIS command
  cpu_baseline    the step in stock torch ops (oracle/torch_native.py) on this host, timed AFTER the PMC child passes have been joined
                  (nothing else of the bench alive): a thread-count scan incl. k = 1, k = 64 (one socket) and k = all; value = the best
  init_steps      untimed steps in front of the contract's --warmup steps (code-object loading, runtime pools)
  config.one_scene_ms_per_step   the locality extreme B=1 x S=16384
  config.inference_forward       --config bf16 only: the bf16 forward alone (one code x the step's points)
  config.gemm_split              headline config only: ms/step of the same workload with the opt-in gemm_split
"""
import argparse
import ctypes as C
import json
import math
import os
import statistics
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from deepsdf_amd import dist  # noqa: E402  (first: HSA_*/NCCL_* defaults must be in the environment before any GPU call)

import torch  # noqa: E402

INIT_STEPS = 40             # default of --init-steps
PEAK_TFLOPS = 157.3          # fp32 MFMA dense peak, MI355X_MICROARCH.md "Peak FP32 (matrix)"
NET = dict(dims=[512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[4],
           xyz_in_all=False, use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)
HEADLINE = dict(code_length=256, scenes_per_batch=64, samples=256)     # 64 scenes x 256 samples = 16384 pts/step
# --network: the NetworkSpecs the reference SHIPS beside the 8x512 headline net, each with its own CodeLength and the
# shipped batch shape (SamplesPerScene 16000, ScenesPerBatch 10).  NOT the headline: these nets are HBM / latency bound, not MFMA bound.
_SHIPPED = dict(dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), xyz_in_all=False, latent_dropout=False,
                weight_norm=True, geom_dimension=3)
NETWORKS = {
    "8x512": dict(net=NET, defaults=HEADLINE, ref="examples/sofas/specs.json:8-43"),
    "6x128": dict(net=dict(_SHIPPED, dims=[128] * 6, latent_in=[2], use_tanh=False), ref="experiments/round_cross_big_network/specs.json:8-19",
                  defaults=dict(code_length=1, scenes_per_batch=10, samples=16000)),
    "4x64": dict(net=dict(_SHIPPED, dims=[64] * 4, latent_in=[1], use_tanh=True), ref="experiments/corner_spheres_only_small_network/specs.json",
                 defaults=dict(code_length=2, scenes_per_batch=10, samples=16000)),
    "4x32": dict(net=dict(_SHIPPED, dims=[32] * 4, latent_in=[2], use_tanh=False), ref="experiments/double_lattice_3D_small_network/specs.json",
                 defaults=dict(code_length=2, scenes_per_batch=10, samples=16000)),
}
L = HEADLINE["code_length"]      # (tools/ scripts that build the headline net import this)


def synth_batches(n_batches, scene_lo, scene_hi, device, seed, scenes_per_batch, samples):
    """Seeded sphere-SDF batches: every batch = scenes_per_batch scenes x samples points, staged on device."""
    gen = torch.Generator().manual_seed(seed)
    n_scenes = scene_hi - scene_lo
    c = (torch.rand(n_scenes, 3, generator=gen) - 0.5) * 0.6
    r = 0.3 + 0.3 * torch.rand(n_scenes, 1, generator=gen)
    out = []
    for b in range(n_batches):
        scenes = (torch.randperm(n_scenes, generator=gen)[:scenes_per_batch]).sort().values if n_scenes > scenes_per_batch \
            else torch.arange(n_scenes)
        idx = scenes.repeat_interleave(samples)
        half = idx.numel() // 2
        xyz = torch.rand(idx.numel(), 3, generator=gen) * 2 - 1
        d = torch.randn(idx.numel() - half, 3, generator=gen)
        d = d / d.norm(dim=1, keepdim=True)
        sel = torch.randperm(idx.numel(), generator=gen)[:idx.numel() - half]
        xyz[sel] = c[idx[sel]] + r[idx[sel]] * d + 0.05 * torch.randn(sel.numel(), 3, generator=gen)
        gt = (xyz - c[idx]).norm(dim=1, keepdim=True) - r[idx]
        out.append(dict(idx=idx, scenes=scenes.to(torch.int64).to(device), xyz=xyz.to(device).contiguous(),
                        gt=gt.reshape(-1).to(device).contiguous(), xyz_cpu=xyz, gt_cpu=gt))
    return out


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def cpu_baseline(L, B, S, headline, NET=NET):
    """The reference's CPU path restated in stock torch ops (oracle/torch_native.py: F.linear + autograd + nn.Embedding(max_norm)
    + torch.optim.Adam, hash dropout masks injected) timed on this host on a bounded sample of the SAME workload (full 16384-pt
    config-2 optimiser steps), with NO child process alive (the PMC passes have been joined before this is called):
      scan   k in {1, 8, 16, 32, 64 (= one socket of the 2 x 64-core box), all hardware threads}: 1 warm-up + 3 timed steps each, median;
      value  the best count of the scan re-timed: 3 warm-up + 10 timed steps, median   (BASELINE.md section 5 protocol: k = 1 and
             k = all are always reported; torch's CPU GEMMs + autograd do not scale to both sockets, so "all" is not the best).
    The explicit-algebra oracle (hand-derived backward; the parity checker) is timed beside it for reference."""
    from oracle import deepsdf_oracle as orc
    from oracle.torch_native import NativeStep
    if not headline:       # other batch shapes (--scenes-per-batch / --samples / --code-length): the same scenes, at most ~16384 points
        S = max(2, min(S, 16384 // B))      # per CPU step (per-point cost is what is reported), and a shorter scan
    net = orc.make_net(L, **NET)
    params = orc.init_params(net, 0)
    gen = torch.Generator().manual_seed(1)
    lat = torch.randn(B, L, generator=gen) / math.sqrt(L)
    b = synth_batches(1, 0, B, "cpu", 7, B, S)[0]
    n_all = torch.get_num_threads()
    counts = (1, 8, 16, 32, 64, n_all) if headline else (1, 32, n_all)

    def time_native(warm, timed):
        nat = NativeStep(net, params, lat, code_bound=1.0)
        masks = orc.dropout_masks(net, 0, 0, b["idx"].numel())
        ts = []
        for i in range(warm + timed):
            t0 = time.perf_counter()
            nat.step(b["idx"], b["xyz_cpu"], b["gt_cpu"], delta=0.1, epoch=1, masks=masks)
            if i >= warm:
                ts.append(time.perf_counter() - t0)
        return statistics.median(ts)

    pts = B * S
    scan = {}
    try:
        for k in sorted({c for c in counts if c <= n_all}):
            torch.set_num_threads(k)
            scan[k] = time_native(1, 3)
        k_best = min(scan, key=scan.get)
        torch.set_num_threads(k_best)
        t_best = time_native(3, 10)
        st = orc.TrainState.create({k: v.clone() for k, v in params.items()}, lat.clone())
        kw = dict(delta=0.1, code_bound=1.0, epoch=1, masks_per_chunk=[orc.dropout_masks(net, 0, 0, b["idx"].numel())])
        orc.train_step(net, st, b["idx"], b["xyz_cpu"], b["gt_cpu"], **kw)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            orc.train_step(net, st, b["idx"], b["xyz_cpu"], b["gt_cpu"], **kw)
            ts.append(time.perf_counter() - t0)
    finally:
        torch.set_num_threads(n_all)
    at = lambda k: dict(value=pts / scan[k], cores=k, ms_per_step=1e3 * scan[k]) if k in scan else None   # noqa: E731
    return dict(value=pts / t_best, unit="point-samples/s", cores=k_best, kind="port", cpu_model=cpu_model(), host_threads=n_all,
                thread_scan_pts_per_s={str(k): pts / t for k, t in scan.items()},
                sample=f"oracle/torch_native.py (stock torch ops + autograd + torch.optim.Adam = the op sequence the reference runs on "
                       f"a CPU), fp32, {k_best} threads (the best of the scan {sorted(scan)} on a {n_all}-thread host; scan points: "
                       f"median of 3 full steps after 1 warm-up): median of 10 full {pts}-pt optimiser steps ({B} scenes x {S} samples, L={L}"
                       f"{', = config 2' if headline else ''}) after 3 warm-up ({1e3 * t_best:.0f} ms/step); no other process of this bench alive "
                       f"meanwhile",
                k1=at(1), k64=at(64), kall=at(n_all),
                oracle_explicit=dict(value=pts / statistics.median(ts), cores=k_best,
                                     sample="oracle/deepsdf_oracle.py train_step (hand-derived backward, the parity checker), "
                                            "median of 3 full steps"))


def pmc_children(result, bench_extra):
    """rocprofv3 --pmc passes over this very command as child processes (tools/pmc.py); fills `result` in place."""
    try:
        from tools import pmc
        args = pmc.CHILD_ARGS + bench_extra
        result["mfma"] = pmc.mfma(args)
        result["traffic"] = pmc.traffic(args)
    except Exception as e:                                    # no rocprofv3, refused counters, ...: report, never fail the bench
        result["error"] = f"{type(e).__name__}: {e}"


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=300)
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--network", choices=sorted(NETWORKS), default="8x512",
                    help="8x512 = the headline decoder.  NOT the headline otherwise: a NetworkSpecs the reference ships (6x128 latent_in [2]; "
                         "4x64 latent_in [1] use_tanh; 4x32 latent_in [2]), with that experiment's CodeLength and batch shape as the defaults "
                         "of --code-length / --scenes-per-batch / --samples")
    ap.add_argument("--scenes-per-batch", type=int, default=None,
                    help="NOT the headline config: scale the batch (x --samples) to see other batch shapes (headline: 64)")
    ap.add_argument("--samples", type=int, default=None, help="samples per scene (headline: 256)")
    ap.add_argument("--code-length", type=int, default=None,
                    help="CodeLength L (headline: 256).  NOT the headline otherwise: the reference's shipped 8x512 experiments use 2 and 16 "
                         "with --scenes-per-batch 10 --samples 16000 (experiments/double_lattice_3D/specs.json:9-38, simple_geom/specs.json:20)")
    ap.add_argument("--config", choices=["fp32", "bf16", "f32split", "bf16split"], default="fp32",
                    help="fp32 = BASELINE configs[1] (the headline, v_mfma_f32_32x32x2_f32); bf16 = configs[4]: the same workload with the "
                         "hidden-layer forward GEMMs on bf16 inputs / fp32 accumulate (v_mfma_f32_32x32x16_bf16), backward, dW and Adam in "
                         "fp32; f32split = the headline workload with NetworkSpecs gemm_split: the fused kernels' hidden GEMMs on the bf16 "
                         "matrix pipe with every fp32 operand cut into three bf16 terms (6 MFMAs per product, fp32 accumulate: fp32 accuracy, "
                         "the fp32 parity tolerances; dW, Adam and everything else unchanged) -- opt-in, NOT the headline; bf16split = configs[4] with "
                         "gemm_split: the bf16 forward as it is, the backward dX chain in split mode")
    ap.add_argument("--init-steps", type=int, default=INIT_STEPS,
                    help="untimed steps in FRONT of the contract's --warmup steps (reported in the line as init_steps; 0 = only --warmup): "
                         "the first ~100 kernel launches of a process load the code objects and grow the runtime's kernarg / signal pools "
                         "(one-off stalls of 80-90 ms were observed as late as the 4th step), a cost a trainer pays once per process")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-profile", action="store_true", help="skip the HIP-event instrumented pass")
    ap.add_argument("--no-pmc", action="store_true", help="skip the rocprofv3 --pmc child passes (HBM traffic, MFMA counters)")
    ap.add_argument("--no-extras", action="store_true", help="skip per-step percentiles and the one-scene extreme")
    args = ap.parse_args()

    from deepsdf_amd import _lib
    from deepsdf_amd.engine import Engine
    from deepsdf_amd.net import NetSpec
    from deepsdf_amd.train import FusedTrainStep

    nw = NETWORKS[args.network]
    NET = nw["net"]
    B = nw["defaults"]["scenes_per_batch"] if args.scenes_per_batch is None else args.scenes_per_batch
    S = nw["defaults"]["samples"] if args.samples is None else args.samples
    L = nw["defaults"]["code_length"] if args.code_length is None else args.code_length
    headline = args.network == "8x512" and dict(code_length=L, scenes_per_batch=B, samples=S) == HEADLINE
    if args.network != "8x512" and args.config != "fp32":
        raise SystemExit("--network " + args.network + " runs with --config fp32 only")
    # rehearsal knobs (1-GPU box): DSDF_DIST_BACKEND=gloo + DSDF_SINGLE_DEVICE=1 run several ranks on ONE card to exercise
    # the multi-process logic; the driver's real multi-GPU runs use neither (backend nccl = RCCL, one rank per GPU)
    rank, local, world = dist.init()
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {args.gpus}")
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)

    pmc_res = {}
    want_pmc = rank == 0 and world == 1 and not args.no_pmc and "ROCPROFILER" not in " ".join(os.environ.keys()).upper()

    bf16 = args.config in ("bf16", "bf16split")
    split = args.config in ("f32split", "bf16split")
    spec = NetSpec(L, forward_bf16=bf16, gemm_split=split, **NET)
    eng = Engine(spec, dev)
    eng.init_like_reference(torch.Generator().manual_seed(0))      # identical on every rank (replicated decoder)
    total_scenes = B if world == 1 else max(512, B * world)        # configs[1] / configs[2]
    lo, hi = dist.owned_scenes(total_scenes, rank, world)
    gen = torch.Generator().manual_seed(100 + rank)
    lat = (torch.randn(hi - lo, L, generator=gen) / math.sqrt(L)).to(dev)
    fused = FusedTrainStep(eng, lat, clamp_dist=0.1, code_reg=True, code_reg_lambda=1e-4, code_bound=1.0, grad_clip=None, seed=rank)
    batches = synth_batches(8, lo, hi, dev, 1000 + rank, B, S)
    n_local = B * S
    n_global = n_local * world

    def step(i, bs=batches, s=S, n_norm=n_global):
        b = bs[i % len(bs)]
        # epoch 1 of the reference's schedules: lr 5e-4 / 1e-3, regulariser ramp min(1, 1/100)  (examples/sofas/specs.json)
        fused(b["scenes"], s, b["xyz"], b["gt"], 1, 5e-4, 1e-3, batch_split=1, n_norm=n_norm)

    # initialisation, not part of the contract's W warm-up steps: the first ~100 launches of a process load the code objects
    # and grow the runtime's kernarg / signal pools (one-off stalls of 80-90 ms were observed as late as the 4th step).
    # Reported in the line as `init_steps` (untimed, like `warmup`).
    for i in range(args.init_steps):
        step(i)
    torch.cuda.synchronize()
    for i in range(args.warmup):
        step(i)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    torch.cuda.synchronize()
    dist.barrier()
    torch.cuda.synchronize()
    elapsed = dist.max_over_ranks(time.perf_counter() - t0, dev)
    loss_t = eng.loss.detach().clone().reshape(1)
    if world > 1:                      # every rank holds its partial of the globally normalised loss (train_deep_sdf.py:519)
        dist.allreduce_sum_(loss_t)
    loss = float(loss_t.item())
    if not math.isfinite(loss):
        raise SystemExit("non-finite loss in the timed region")

    ms_per_step = 1e3 * elapsed / args.steps
    value = n_global * args.steps / elapsed
    flop_per_pt = 6 * spec.w_mac
    step_tflops = flop_per_pt * (value / world) / 1e12

    # ---- single-step distribution: events on the compute stream around every step (separate pass: outside the timed region) ----
    step_stats = None
    if not args.no_extras:
        k = min(args.steps, 200)
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
        evs[0].record()
        for i in range(k):
            step(args.warmup + i)
            evs[i + 1].record()
        torch.cuda.synchronize()
        ts = sorted(evs[i].elapsed_time(evs[i + 1]) for i in range(k))
        step_stats = dict(median=ts[k // 2], p10=ts[k // 10], p90=ts[(9 * k) // 10], n=k,
                          note="device time between per-step events on the compute stream; includes the event records")

    # ---- instrumented pass: HIP events around every launch of each kernel class, same steps, same stream ----
    roofline = None
    if not args.no_profile:
        lib = _lib.lib()
        lib.dsdf_profile_enable(1)
        for i in range(args.steps):
            step(args.warmup + i)
        torch.cuda.synchronize()
        prof = _lib.DsdfProfile()
        _lib.check(lib.dsdf_profile_read(C.byref(prof)))
        lib.dsdf_profile_enable(0)
        kern = {}
        for c, name in enumerate(_lib.PROF_NAMES):
            if prof.count[c]:
                kern[name] = dict(launches_per_step=prof.count[c] / args.steps, avg_us=1e3 * prof.ms[c] / prof.count[c],
                                  ms_per_step=prof.ms[c] / args.steps,
                                  tflops=prof.flops[c] / (prof.ms[c] * 1e-3) / 1e12 if prof.ms[c] > 0 else None)
        dom = max(kern, key=lambda k: kern[k]["ms_per_step"])
        # The roof of the dominant kernel by the pipe its GEMMs run on: fp32 MFMA 157.3; f32split: bf16 dense peak / 6 (6 MFMAs per fp32
        # product); the merged forward+backward launch of config 5 runs HALF of its algorithmic FLOPs (the forward) on the bf16 pipe at
        # 2500 and the other half (the backward) on fp32 MFMAs (bf16: 157.3) or in split mode (bf16split: 2500 / 6): the roof of such a
        # launch is total FLOPs / (time of each half at its own peak) = the harmonic mean of the two peaks
        p_split = 2500.0 / 6.0
        if dom == "fused_fwd_bwd_kernel" and bf16:
            peak = 2.0 / (1.0 / 2500.0 + 1.0 / (p_split if split else PEAK_TFLOPS))
        else:
            peak = p_split if split else PEAK_TFLOPS
        roofline = dict(bound="mfma", kernel=dom, achieved=kern[dom]["tflops"], peak=peak, unit="TFLOP/s",
                        frac=kern[dom]["tflops"] / peak, traffic=None, executed_frac=None,
                        avg_launch_us=kern[dom]["avg_us"], launches_per_step=kern[dom]["launches_per_step"],
                        step_achieved=step_tflops, step_frac=step_tflops / PEAK_TFLOPS, kernels=kern,
                        peak_note=("harmonic mean of the bf16 dense peak (forward half of the launch) and " + ("bf16 dense peak / 6" if split else "the fp32 MFMA peak")
                                   + " (backward half)" if (bf16 and dom == "fused_fwd_bwd_kernel") else
                                   "bf16 dense peak 2500 TFLOP/s / 6 MFMAs per product (step_frac stays against the fp32 MFMA peak 157.3)"
                                   if split else "fp32 MFMA dense peak"),
                        note="achieved = ALGORITHMIC 2*pts*sum(in*out) of the hidden layers the kernel covers (the reference's "
                             "dense formulation, SURVEY 8d) / HIP-event time around its launches; segment mode executes fewer MFMA "
                             "FLOPs than that (per-scene latent products are hoisted, DESIGN.md 4): executed_frac = MFMA FLOPs "
                             "issued (SQ_INSTS_VALU_MFMA_MOPS_F32 x 512) / the same HIP-event time / peak; "
                             "step_* = 6*W_mac*pts/s over the un-instrumented timed region")

    # ---- the locality extreme of config 2: ONE scene x 16384 samples per step ----
    one_scene = None
    if rank == 0 and world == 1 and headline and not args.no_extras:
        ob = synth_batches(2, 0, 1, dev, 77, 1, 16384)
        olat = (torch.randn(1, L, generator=torch.Generator().manual_seed(5)) / math.sqrt(L)).to(dev)
        ofused = FusedTrainStep(eng, olat, clamp_dist=0.1, code_reg=True, code_reg_lambda=1e-4, code_bound=1.0, grad_clip=None, seed=3)
        ostep = lambda i: ofused(ob[i % 2]["scenes"], 16384, ob[i % 2]["xyz"], ob[i % 2]["gt"], 1, 5e-4, 1e-3, n_norm=16384)  # noqa: E731
        for i in range(20):
            ostep(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(100):
            ostep(i)
        torch.cuda.synchronize()
        one_scene = 1e3 * (time.perf_counter() - t1) / 100

    # ---- headline config only: the same step with NetworkSpecs gemm_split (opt-in), so that the record carries both ----
    split_extra = None
    if rank == 0 and world == 1 and args.config == "fp32" and not args.no_extras and args.network == "8x512":
        n_it = max(10, min(100, (100 * 16384) // n_local))
        seng = Engine(NetSpec(L, gemm_split=True, **NET), dev)
        seng.init_like_reference(torch.Generator().manual_seed(0))
        slat = lat.clone()
        sfused = FusedTrainStep(seng, slat, clamp_dist=0.1, code_reg=True, code_reg_lambda=1e-4, code_bound=1.0, grad_clip=None, seed=3)
        sstep = lambda i: sfused(batches[i % len(batches)]["scenes"], S, batches[i % len(batches)]["xyz"], batches[i % len(batches)]["gt"],  # noqa: E731
                                 1, 5e-4, 1e-3, batch_split=1, n_norm=n_global)
        for i in range(args.init_steps + args.warmup):     # its kernels are new to the process: the same initialisation as the headline's
            sstep(i)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_it):
            sstep(i)
        torch.cuda.synchronize()
        sms = 1e3 * (time.perf_counter() - t1) / n_it
        sev = [torch.cuda.Event(enable_timing=True) for _ in range(n_it + 1)]
        sev[0].record()
        for i in range(n_it):
            sstep(i)
            sev[i + 1].record()
        torch.cuda.synchronize()
        sts = sorted(sev[i].elapsed_time(sev[i + 1]) for i in range(n_it))
        split_extra = dict(ms_per_step=sms, value=n_local / (sms * 1e-3), steps=n_it, step_time_ms_median=sts[n_it // 2],
                           note="NOT the headline: the same workload with NetworkSpecs gemm_split = true (the fused kernels' hidden GEMMs as 6 "
                                "bf16 MFMAs on 3-way split fp32 operands, fp32 accumulate; same parity tolerances) -- `bench.py --config f32split` "
                                "is the full line")
        del seng, sfused

    # ---- config 5 only: its INFERENCE forward alone (one code x the step's points: dsdf_decode_latent = hoist + the 8-wave kernel) ----
    bf16_fwd = None
    if rank == 0 and world == 1 and bf16 and not args.no_extras:
        z = (torch.randn(L, generator=torch.Generator().manual_seed(9)) / math.sqrt(L)).to(dev)
        q = (torch.rand(n_local, 3, generator=torch.Generator().manual_seed(10)) * 2 - 1).to(dev)
        for _ in range(10):
            eng.decode_latent(z, q)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(100):
            eng.decode_latent(z, q)
        e1.record()
        torch.cuda.synchronize()
        us = 1e3 * e0.elapsed_time(e1) / 100
        tf = 2.0 * n_local * spec.w_mac / (us * 1e-6) / 1e12
        bf16_fwd = dict(us_per_call=us, tflops=tf, frac_of_bf16_dense_peak=tf / 2500.0, points=n_local,
                        note="HIP events around 100 calls of Engine.decode_latent (seg_hoist_kernel + fused_forward_bf16x8_kernel + the "
                             "host-side launch path); the kernel alone: profiles/r03_bf16_decode_kernel_stats.csv (round 3; the kernel is unchanged)")

    # ---- PMC child passes (GPU, child processes), THEN the CPU baseline on an otherwise idle host; both after all GPU timing ----
    if want_pmc:
        extra = ["--scenes-per-batch", str(B), "--samples", str(S), "--config", args.config, "--code-length", str(L), "--network", args.network]
        pmc_children(pmc_res, extra)
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline and not bf16:
        torch.cuda.synchronize()
        cpu = cpu_baseline(L, B, S, headline, NET)
    if roofline is not None:
        dom = roofline["kernel"]
        if split:      # the profiling class keeps its name; the launched kernel is the split twin
            dom = dom.replace("_kernel", "_split_kernel")
        if "traffic" in pmc_res and dom in pmc_res["traffic"]:
            roofline["traffic"] = pmc_res["traffic"][dom]["hbm_bytes_per_launch"]
            roofline["traffic_source"] = ("measured in this run: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this command "
                                          "(separate passes, FETCH_SIZE x2 on gfx950), per launch")
            roofline["traffic_per_kernel_MB"] = {k: v["hbm_bytes_per_launch"] / 1e6 for k, v in pmc_res["traffic"].items()}
        else:
            tfile = os.path.join(ROOT, "profiles", "r04_pmc_traffic.json")
            if os.path.exists(tfile) and dom in json.load(open(tfile)):
                roofline["traffic"] = json.load(open(tfile))[dom]["hbm_bytes_per_launch"]
                roofline["traffic_source"] = "NOT measured in this run (" + pmc_res.get("error", "--no-pmc") + "): profiles/r04_pmc_traffic.json"
        if "mfma" in pmc_res and dom in pmc_res["mfma"]:
            m = pmc_res["mfma"][dom]
            t = roofline["avg_launch_us"] * 1e-6
            roofline["executed_frac"] = (m["exec_mfma_flop_bf16"] / t / 2500e12 if split else     # executed bf16 MFMA FLOPs / bf16 dense peak
                                         m["exec_mfma_flop"] / t / (PEAK_TFLOPS * 1e12))
            roofline["executed"] = {k: dict(exec_mfma_gflop=v["exec_mfma_flop"] / 1e9, exec_mfma_bf16_gflop=v.get("exec_mfma_flop_bf16", 0.0) / 1e9, mfma_busy_frac=v.get("mfma_busy_frac"),
                                            clock_ghz_profiled=v.get("clock_ghz"), profiled_us=v.get("duration_us"),
                                            mops_per_mfma_inst=v.get("mops_per_mfma_inst"))
                                    for k, v in pmc_res["mfma"].items() if v.get("exec_mfma_flop", 0) + v.get("exec_mfma_flop_bf16", 0) > 0}
        if "error" in pmc_res:
            roofline["pmc_error"] = pmc_res["error"]

    if rank == 0 and roofline is not None and args.network != "8x512":
        # The shipped small nets are not MFMA bound: algorithmic HBM bytes per point-sample per step in the SURVEY 8(d) convention --
        # the batch stream (xyz + sdf: 16 B) plus, per hidden layer, the activation and the dP of the layer's output written once and
        # read once (4 x 4 B x width; the design materialises both for the weight-gradient pass) -- against the 8 TB/s HBM peak.
        hidden = sum(NET["dims"])
        alg_bytes = 16 + 16 * hidden
        gbs = alg_bytes * value / 1e9
        roofline = dict(bound="hbm", achieved=gbs, peak=8000.0, unit="GB/s", frac=gbs / 8000.0, traffic=None,
                        algorithmic_bytes_per_point=alg_bytes, mfma_step_tflops=step_tflops, mfma_step_frac=step_tflops / PEAK_TFLOPS,
                        kernels=roofline["kernels"], flop_per_point=flop_per_pt,
                        note="WHOLE STEP (all launches), not one kernel: achieved = algorithmic bytes per point x points/s; "
                             "mfma_* = 6 W_mac x points/s against the fp32 MFMA peak, for scale")
    if rank == 0:
        cfg = {"workload": ((f"configs[4]: configs[1] with the hidden-layer FORWARD GEMMs on bf16 inputs / fp32 accumulate; backward, dW, "
                             f"Adam fp32; {n_local} pts/step ({B} scenes x {S} samples)" if bf16 else
                             f"configs[1] with NetworkSpecs gemm_split (opt-in): the fused forward/backward GEMMs as 6 bf16 MFMAs on 3-way "
                             f"split fp32 operands; {n_local} pts/step ({B} scenes x {S} samples)" if split else
                             f"{'configs[1]' if headline else 'NOT the headline (configs[1] at another batch shape / code length)'}: {B} synthetic sphere-SDF scenes, latent_dim={L}, 8x512 decoder + layer-4 skip, "
                             f"weight-norm, dropout 0.2, {n_local} pts/step ({B} scenes x {S} samples), fp32" if args.network == "8x512" else
                             f"NOT the headline: the reference's shipped NetworkSpecs {args.network} ({nw['ref']}: dims {NET['dims']}, latent_in "
                             f"{NET['latent_in']}, use_tanh {NET['use_tanh']}), CodeLength {L}, {n_local} pts/step ({B} scenes x {S} samples), fp32") if world == 1 else
                            f"configs[2]: {total_scenes} scenes sharded over {world} ranks, {n_local} pts/step/rank, RCCL all-reduce of "
                            "decoder grads (asynchronous, latent Adam under it)"),
               "points_per_step_per_gpu": n_local, "headline_config": headline and args.config == "fp32", "parallelism": f"dp{world}", "final_loss": loss}
        if bf16_fwd is not None:
            cfg["inference_forward"] = bf16_fwd
        if split_extra is not None:
            cfg["gemm_split"] = split_extra
        if one_scene is not None:
            cfg["one_scene_ms_per_step"] = one_scene
            cfg["one_scene_value"] = 16384 / (one_scene * 1e-3)
        print(json.dumps({
            "metric": "SDF point-samples/sec per training step (8x512 decoder, 16384 pts)", "value": value,
            "unit": "point-samples/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "init_steps": args.init_steps,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": ("bf16-fwd/f32 (backward dX GEMMs: 3 bf16 terms per operand, 6 bf16 MFMAs per product, fp32 accumulate)" if bf16 and split
                      else "bf16-fwd/f32") if bf16 else ("f32 (hidden GEMMs of the fused kernels: 3 bf16 terms per operand, 6 bf16 MFMAs per product, "
                                                   "fp32 accumulate)" if split else "f32"), "data": "synthetic", "config": cfg, "step_time_ms": step_stats,
            "roofline": roofline, "cpu_baseline": cpu}))
    dist.shutdown()


if __name__ == "__main__":
    main()
