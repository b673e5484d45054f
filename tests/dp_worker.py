"""Child process of tests/test_gpu_dist.py: ONE data-parallel rank driving the HIP training step (not a test module).

Launched as a fresh process (never a fork/exec of a process that has touched the GPU) with the torchrun environment
(RANK / WORLD_SIZE / MASTER_*).  Single-card rehearsal knobs: DSDF_DIST_BACKEND=gloo + DSDF_SINGLE_DEVICE=1 put every
rank on cuda:0 with the gloo transport; on a multi-GPU node the same script runs with backend nccl (= RCCL), one rank
per GPU.  The step is the product path the trainer and bench.py use at world > 1 (deepsdf_amd.train.FusedTrainStep).
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main(work_dir):
    from deepsdf_amd import dist                  # first: HSA_* defaults before any GPU call
    import torch
    from deepsdf_amd.engine import Engine
    from deepsdf_amd.net import NetSpec
    from deepsdf_amd.train import FusedTrainStep

    rank, local, world = dist.init()          # reads the rehearsal knobs itself
    dev = torch.device("cuda", local)
    torch.cuda.set_device(dev)
    case = torch.load(os.path.join(work_dir, "case.pt"), weights_only=True)
    spec = NetSpec(case["L"], **case["net_specs"])
    eng = Engine(spec, dev)
    eng.load_params(case["params"])
    lo, hi = dist.owned_scenes(case["n_scenes"], rank, world)
    lat = case["lat0"][lo:hi].to(dev).contiguous()
    fused = FusedTrainStep(eng, lat, clamp_dist=case["delta"], code_reg=True, code_reg_lambda=case["lam"],
                           code_bound=case["code_bound"], grad_clip=None, seed=case["seed_base"] + rank)
    S = case["S"]
    n_global = case["n_scenes"] * S
    out = dict(lo=lo, hi=hi, steps=[])
    ran_under = []
    for st in case["steps"]:
        rows = slice(lo * S, hi * S)                         # this rank's scenes: a contiguous block of the global batch
        xyz, gt = st["xyz"][rows].to(dev).contiguous(), st["gt"][rows].reshape(-1).to(dev).contiguous()
        scene_rows = torch.arange(hi - lo, dtype=torch.int64, device=dev)      # rows of the LOCAL latent table
        fused(scene_rows, S, xyz, gt, case["epoch"], case["lr"][0], case["lr"][1], batch_split=1, n_norm=n_global,
              under_allreduce=lambda: ran_under.append(1))
        loss = eng.loss.detach().clone()
        dist.allreduce_sum_(loss)                            # per-rank partials of the globally normalised loss
        torch.cuda.synchronize()
        out["steps"].append(dict(loss=loss.cpu(), params=eng.params.cpu().clone(), exp_avg=eng.exp_avg.cpu().clone(),
                                 exp_avg_sq=eng.exp_avg_sq.cpu().clone(), packed=eng.packed.cpu().clone(),
                                 grads=eng.grads.cpu().clone(), lat=lat.cpu().clone(), lat_m=fused.lat_m.cpu().clone(),
                                 lat_v=fused.lat_v.cpu().clone(), step=eng.step))
    out["under_calls"] = len(ran_under)
    out["ar_buckets"] = fused.ar_buckets
    dist.barrier()
    torch.save(out, os.path.join(work_dir, f"rank{rank}.pt"))
    torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main(sys.argv[1])
