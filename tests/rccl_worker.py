"""Child process of tests/test_gpu_dist.py: the data-parallel step through REAL RCCL on a one-GPU box (not a test module).

A one-rank `nccl` process group (DSDF_DIST_FORCE_GROUP=1, deepsdf_amd.dist) makes the product step (FusedTrainStep) take its
world > 1 call sequence -- asynchronous all-reduce on RCCL's stream, latent Adam under it, stream-ordered work.wait(), decoder
Adam -- with the transport the driver's multi-GPU runs use, which the gloo rehearsals cannot show (gloo's wait() blocks the
host).  The same batches are first run WITHOUT a group through the same call sequence (DSDF_FORCE_DP_PATH=1): the two must be
bit-identical, for 1, 2 and 4 gradient buckets.  Then the object collectives and the replica check the trainer uses
(deepsdf_amd/train.py: broadcast_object_list, all_gather_object, replicas_identical) run through the same group.
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

    assert os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY") == "0"        # the import put the pool's default in place
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    case = torch.load(os.path.join(work_dir, "case.pt"), weights_only=True)
    spec = NetSpec(case["L"], **case["net_specs"])
    S, B = case["S"], case["n_scenes"]
    batches = [(st["xyz"].to(dev).contiguous(), st["gt"].reshape(-1).to(dev).contiguous()) for st in case["steps"]]
    scene_rows = torch.arange(B, dtype=torch.int64, device=dev)

    def run(buckets):
        os.environ["DSDF_AR_BUCKETS"] = str(buckets)
        eng = Engine(spec, dev)
        eng.load_params(case["params"])
        lat = case["lat0"].to(dev).contiguous().clone()
        fused = FusedTrainStep(eng, lat, clamp_dist=case["delta"], code_reg=True, code_reg_lambda=case["lam"],
                               code_bound=case["code_bound"], grad_clip=None, seed=case["seed"])
        assert fused.ar_buckets == buckets
        under, losses = [], []
        for xyz, gt in batches:
            fused(scene_rows, S, xyz, gt, case["epoch"], case["lr"][0], case["lr"][1], batch_split=1, n_norm=B * S,
                  under_allreduce=lambda: under.append(1))
            losses.append(eng.loss.detach().clone())
        torch.cuda.synchronize()
        assert len(under) == len(batches) and eng.step == len(batches)
        return dict(params=eng.params.clone(), exp_avg=eng.exp_avg.clone(), exp_avg_sq=eng.exp_avg_sq.clone(),
                    packed=eng.packed.clone(), grads=eng.grads.clone(), lat=lat.clone(), lat_m=fused.lat_m.clone(),
                    lat_v=fused.lat_v.clone(), loss=torch.cat(losses))

    ks = case["buckets"]
    os.environ["DSDF_FORCE_DP_PATH"] = "1"        # the data-parallel call sequence, no process group
    assert not dist.is_multi()
    ref = {k: run(k) for k in ks}
    del os.environ["DSDF_FORCE_DP_PATH"]

    rank, local, world = dist.init()              # DSDF_DIST_FORCE_GROUP=1: a one-rank group, backend nccl (= RCCL)
    assert (rank, local, world) == (0, 0, 1) and dist.is_multi()
    assert torch.distributed.get_backend() == "nccl"
    got = {k: run(k) for k in ks}
    report = {}
    for k in ks:
        for name, t in ref[k].items():
            assert torch.equal(t, got[k][name]), (k, name)
        assert bool(torch.isfinite(got[k]["loss"]).all()) and float(got[k]["params"].abs().sum()) > 0
        report[k] = float(got[k]["loss"][-1])
    # one bucket and K buckets are the same step up to the split-K order of the weight gradients
    for k in ks[1:]:
        d = (got[k]["params"] - got[ks[0]]["params"]).abs().max()
        assert float(d) <= 1e-5, (k, float(d))

    # what the trainer does through the group (deepsdf_amd/train.py main_function)
    t = torch.arange(12.0).reshape(3, 4)
    objs = [t]
    torch.distributed.broadcast_object_list(objs, src=0)
    parts = [None]
    torch.distributed.all_gather_object(parts, t)
    assert torch.equal(objs[0], t) and torch.equal(parts[0], t)
    assert dist.replicas_identical(got[ks[0]]["params"])
    x = torch.ones(5, device=dev)
    dist.allreduce_sum_(x)
    assert torch.equal(x.cpu(), torch.ones(5)) and dist.max_over_ranks(3.5, dev) == 3.5
    dist.barrier()
    torch.save(dict(ok=True, last_loss=report, backend=torch.distributed.get_backend()), os.path.join(work_dir, "rccl.pt"))
    dist.shutdown()


if __name__ == "__main__":
    main(sys.argv[1])
