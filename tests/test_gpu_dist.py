"""-m gpu: the data-parallel HIP step at world_size 2 (replaces nn.DataParallel, train_deep_sdf.py:353).

Two FRESH child processes (tests/dp_worker.py) share the box's single card (DSDF_SINGLE_DEVICE=1) and exchange the
decoder-gradient arena over gloo (DSDF_DIST_BACKEND=gloo; the driver's multi-GPU runs use nccl = RCCL with the same
code).  Each rank owns half of the scenes and runs the product step (FusedTrainStep: HIP forward/backward with the
GLOBAL normaliser -> asynchronous SUM all-reduce, latent Adam under it -> decoder Adam).  Checked:
 (i)   the replicas are bit-identical after every step (parameters, both Adam moments, packed weights, reduced gradient);
 (ii)  they equal the single-process HIP step on the concatenated batch (gradient accumulation over the two shards);
 (iii) they equal the float64 oracle on the concatenated batch; latent rows match the owner rank's rows."""
import math
import os
import socket
import subprocess
import sys

import pytest
import torch

from oracle import deepsdf_oracle as orc
from tests.golden_io import rel_err, worst_elem
from tests.hip_helpers import spec_from_meta
from tests.test_gpu_parity import BIG, GRAD_ELEM_TOL, GRAD_TOL, PARAM_TOL, _safe_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


@pytest.mark.parametrize("buckets", [1, 2, 4], ids=["one_allreduce", "two_buckets", "four_buckets"])
def test_two_rank_hip_step_equals_single_process_and_oracle(tmp_path, buckets):
    """buckets = 2 (DSDF_AR_BUCKETS=2): the decoder gradient travels in two all-reduces, the late layers' under the early layers'
    weight-gradient launch (DsdfLossCfg.dw_phase 1 / 2).  The K-split of the two half-launches is finer than the single launch's,
    so the gradients are not bit-equal to the one-bucket run's -- the replicas still are to each other, and both meet the same
    bounds against the single-process step and the float64 oracle."""
    from deepsdf_amd import dist
    from deepsdf_amd.engine import Engine, make_segments
    world, L, n_scenes, S, seed_base = 2, 256, 8, 256, 100
    N = n_scenes * S
    net = orc.make_net(L, **BIG)
    params = orc.init_params(net, 61)
    lat0 = torch.randn(n_scenes, L, generator=torch.Generator().manual_seed(62)) / math.sqrt(L)
    lat0[5] *= 1.6 / lat0[5].norm()                          # a code above CodeBound on rank 1's shard
    st64 = orc.TrainState.create({k: v.double() for k, v in params.items()}, lat0.double())
    shards = [dist.owned_scenes(n_scenes, r, world) for r in range(world)]

    def masks_for(step):                                     # every rank hashes ITS rows (0-based) with ITS seed
        per = [orc.dropout_masks(net, seed_base + r, step, (hi - lo) * S) for r, (lo, hi) in enumerate(shards)]
        return [None if per[0][l] is None else torch.cat([p[l] for p in per]) for l in range(len(per[0]))]

    steps, oracle_after = [], []
    for step in range(2):
        masks = masks_for(step)
        idx, xyz, gt = _safe_batch(net, st64, n_scenes, S, 900 + step, 0.1, 1.0, None, masks=masks)
        r64 = orc.train_step(net, st64, idx, xyz.double(), gt.double(), delta=0.1, code_bound=1.0, epoch=57,
                             masks_per_chunk=[masks])
        steps.append(dict(xyz=xyz, gt=gt))
        oracle_after.append(dict(loss=r64["loss"], grads={k: v.clone() for k, v in r64["grads"].items()},
                                 params={k: v.clone() for k, v in st64.params.items()}, lat=st64.latents.clone(),
                                 m={k: v.clone() for k, v in st64.m.items()}))
    torch.save(dict(L=L, net_specs=BIG, params=params, lat0=lat0, n_scenes=n_scenes, S=S, delta=0.1, lam=1e-4,
                    code_bound=1.0, epoch=57, lr=[5e-4, 1e-3], seed_base=seed_base, steps=steps),
               os.path.join(str(tmp_path), "case.pt"))

    port = _free_port()
    procs = []
    for r in range(world):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(world), MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), DSDF_DIST_BACKEND="gloo", DSDF_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0",
                   DSDF_AR_BUCKETS=str(buckets))
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dp_worker.py"), str(tmp_path)], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True))
    logs = []
    for p in procs:
        try:
            out, _ = p.communicate(timeout=420)
        except subprocess.TimeoutExpired:
            for q in procs:
                q.kill()
            raise
        logs.append(out)
    for r, p in enumerate(procs):
        assert p.returncode == 0, f"rank {r} failed:\n{logs[r][-3000:]}"
    outs = [torch.load(os.path.join(str(tmp_path), f"rank{r}.pt"), weights_only=True) for r in range(world)]
    assert [(o["lo"], o["hi"]) for o in outs] == shards
    assert all(o["under_calls"] == 2 for o in outs)          # the overlap hook ran once per step
    assert all(o["ar_buckets"] == buckets for o in outs)     # (2: the phased backward was accepted, no silent fall-back)

    # single-process HIP on the concatenated batch: the shards as two accumulated chunks, each with its rank's dropout key
    spec = spec_from_meta(dict(L=L, net_specs=BIG))
    eng = Engine(spec)
    eng.load_params(params)
    lat = lat0.cuda().contiguous().clone()
    dlat, lat_m, lat_v = torch.zeros_like(lat), torch.zeros_like(lat), torch.zeros_like(lat)
    for step in range(2):
        a, b = outs[0]["steps"][step], outs[1]["steps"][step]
        # (i) bit-identical replicas
        for k in ("params", "exp_avg", "exp_avg_sq", "packed", "grads", "loss"):
            assert torch.equal(a[k], b[k]), (step, k)
        assert a["step"] == b["step"] == step + 1
        # (ii) single-process HIP
        xyz, gt = steps[step]["xyz"].cuda(), steps[step]["gt"].reshape(-1).cuda()
        for r, (lo, hi) in enumerate(shards):
            sc, so = make_segments(torch.arange(lo, hi).repeat_interleave(S).cuda())
            eng.train_forward_backward(lat, dlat, sc, so, xyz[lo * S:hi * S].contiguous(), gt[lo * S:hi * S].contiguous(),
                                       n_norm=N, clamp_dist=0.1, reg_coef=1e-4 * 0.57, code_bound=1.0, training=True,
                                       seed=seed_base + r, row_offset=0, accumulate=r > 0, seg_len=S)
        loss1 = float(eng.loss)                                          # accumulate=True adds the chunk's loss to the first one's
        g1 = eng.grads.cpu().clone()
        eng.adam_step(lat, dlat, lat_m, lat_v, 5e-4, 1e-3)
        assert abs(float(a["loss"]) - loss1) <= 1e-6 * abs(loss1), step
        assert rel_err(a["grads"], g1) <= 1e-6, step                     # all-reduce == accumulation over the shards
        assert rel_err(a["params"], eng.params.cpu()) <= 1e-6, step
        assert rel_err(a["exp_avg"], eng.exp_avg.cpu()) <= 1e-6 and rel_err(a["exp_avg_sq"], eng.exp_avg_sq.cpu()) <= 2e-6, step
        lat_dp = torch.cat([o["steps"][step]["lat"] for o in outs])      # owner ranks' rows, no communication
        assert rel_err(lat_dp, lat.cpu()) <= 1e-6, step
        assert rel_err(torch.cat([o["steps"][step]["lat_m"] for o in outs]), lat_m.cpu()) <= 1e-6, step
        # (iii) the float64 oracle on the concatenated batch
        o = oracle_after[step]
        assert abs(float(a["loss"]) - o["loss"]) <= 1e-5 * abs(o["loss"]), step
        P, G, M = (eng.named_views(a[k]) for k in ("params", "grads", "exp_avg"))
        for k in o["params"]:
            assert rel_err(G[k], o["grads"][k]) <= GRAD_TOL, (step, k)
            assert rel_err(M[k], o["m"][k]) <= GRAD_TOL, (step, k)
            assert rel_err(P[k], o["params"][k]) <= PARAM_TOL, (step, k)
            # element-wise (tests/test_gpu_parity.py): worst entry of the reduced gradient against its tensor's max; worst parameter
            # entry against the Adam step (2048 points only: a gradient entry near Adam's eps weighs more than at full size)
            assert worst_elem(G[k], o["grads"][k]) <= GRAD_ELEM_TOL, (step, k)
            assert float((P[k].double() - o["params"][k]).abs().max()) <= (step + 1) * 1e-2 * 5e-4, (step, k)
        assert rel_err(lat_dp, o["lat"]) <= PARAM_TOL, step


def test_dp_step_through_a_one_rank_rccl_group_is_bit_identical(tmp_path):
    """The RCCL branch EXECUTES (round-3 review: `backend="nccl"`, the stream-ordered `work.wait()` and the object collectives had
    only ever run as gloo): tests/rccl_worker.py, a fresh child, runs three steps of the 8 x 512 net through FusedTrainStep's
    data-parallel call sequence twice -- without a group (DSDF_FORCE_DP_PATH=1) and through a one-rank `nccl` process group
    (DSDF_DIST_FORCE_GROUP=1) -- with 1, 2 and 4 gradient buckets; parameters, both Adam moments, packed weights, gradients, the
    latent rows and their moments and every step's loss must be bit-identical, and the trainer's object collectives and replica
    check pass through the same group.  (Replaces nn.DataParallel, train_deep_sdf.py:353.)"""
    L, n_scenes, S = 256, 8, 256
    net = orc.make_net(L, **BIG)
    params = orc.init_params(net, 71)
    lat0 = torch.randn(n_scenes, L, generator=torch.Generator().manual_seed(72)) / math.sqrt(L)
    lat0[3] *= 1.4 / lat0[3].norm()                         # a code above CodeBound: the renorm runs
    steps = []
    for step in range(3):
        g = torch.Generator().manual_seed(730 + step)
        xyz = torch.rand(n_scenes * S, 3, generator=g) * 2 - 1
        gt = (xyz - 0.1).norm(dim=1, keepdim=True) - 0.45 + 0.02 * torch.randn(n_scenes * S, 1, generator=g)
        steps.append(dict(xyz=xyz, gt=gt))
    torch.save(dict(L=L, net_specs=BIG, params=params, lat0=lat0, n_scenes=n_scenes, S=S, delta=0.1, lam=1e-4, code_bound=1.0,
                    epoch=57, lr=[5e-4, 1e-3], seed=41, steps=steps, buckets=[1, 2, 4]), os.path.join(str(tmp_path), "case.pt"))
    env = dict(os.environ, DSDF_DIST_FORCE_GROUP="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DSDF_DIST_BACKEND", "DSDF_SINGLE_DEVICE", "DSDF_FORCE_DP_PATH", "DSDF_AR_BUCKETS",
              "HSA_ENABLE_IPC_MODE_LEGACY"):               # (the last one: deepsdf_amd.dist must put the default in place itself)
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py"), str(tmp_path)], env=env, cwd=ROOT,
                       stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-4000:]
    out = torch.load(os.path.join(str(tmp_path), "rccl.pt"), weights_only=True)
    assert out["ok"] and out["backend"] == "nccl" and sorted(out["last_loss"]) == [1, 2, 4]
    assert all(math.isfinite(v) and v > 0 for v in out["last_loss"].values())


def test_trainer_through_a_one_rank_rccl_group(tmp_path):
    """`train_deep_sdf.py -e DIR` with DSDF_DIST_FORCE_GROUP=1: the drop-in trainer's whole world > 1 leg (arena broadcast,
    broadcast_object_list of the latent table, the phased gradient exchange with DSDF_AR_BUCKETS=4, all_gather_object for the
    checkpoints, the replica check) runs through a one-rank RCCL group, writes the reference's artefacts and resumes."""
    import json
    from tests.test_gpu_module_trainer import _make_experiment
    exp = _make_experiment(str(tmp_path), 4, specs_over={"NumEpochs": 2, "SnapshotFrequency": 2, "AdditionalSnapshots": [],
                                                         "LogFrequency": 1, "SamplesPerScene": 1024})
    script = os.path.join(ROOT, "train_deep_sdf.py")

    def run(extra):
        env = dict(os.environ, DSDF_DIST_FORCE_GROUP="1", DSDF_AR_BUCKETS="4", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
        for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "DSDF_DIST_BACKEND", "DSDF_SINGLE_DEVICE"):
            env.pop(k, None)
        r = subprocess.run([sys.executable, script, "-e", exp] + extra, env=env, cwd=ROOT, stdout=subprocess.PIPE,
                           stderr=subprocess.STDOUT, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-4000:]
        return r.stdout

    out = run([])
    assert out.count("decoder replicas bit-identical on 1 ranks") >= 2, out[-3000:]
    lc = torch.load(os.path.join(exp, "LatentCodes", "latest.pth"), weights_only=True)
    assert lc["epoch"] == 2 and lc["latent_codes"]["weight"].shape == (4, 4)
    logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    assert logs["epoch"] == 2 and len(logs["loss"]) == 4 and all(math.isfinite(v) for v in logs["loss"])
    specs = json.load(open(os.path.join(exp, "specs.json")))
    specs["NumEpochs"], specs["SnapshotFrequency"] = 4, 4
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    out = run(["-c", "latest"])
    assert "starting from epoch 3" in out, out[-3000:]
    logs4 = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    assert logs4["epoch"] == 4 and logs4["loss"][:4] == logs["loss"] and len(logs4["loss"]) == 8


def _torchrun_two_ranks(script_args, timeout=900, **extra_env):
    """`python -m torch.distributed.run --nproc-per-node 2 <script_args>` as a FRESH child of the test process, both ranks on
    the box's one card over gloo (the rehearsal knobs of deepsdf_amd.dist.init).  Returns the merged stdout + stderr."""
    env = dict(os.environ, DSDF_DIST_BACKEND="gloo", DSDF_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0", **extra_env)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port())] + script_args
    r = subprocess.run(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=timeout)
    assert r.returncode == 0, r.stdout[-4000:]
    return r.stdout


def test_bench_runs_at_world_size_two():
    """bench.py's N > 1 leg (BASELINE configs[2]: 512 scenes sharded, 16384 points per rank and step, all-reduce of the decoder
    gradients) launched exactly as the driver launches it, at world size 2: ONE JSON line from rank 0 with the contract's keys."""
    import json
    out = _torchrun_two_ranks([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "3", "--warmup", "1",
                               "--no-cpu-baseline", "--no-pmc"])
    lines = [ln for ln in out.splitlines() if ln.startswith('{"metric"')]
    assert len(lines) == 1, out[-3000:]
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["steps"] == 3 and rec["warmup"] == 1 and rec["scaling"] == "weak"
    assert rec["config"]["workload"].startswith("configs[2]") and rec["config"]["parallelism"] == "dp2"
    assert rec["config"]["points_per_step_per_gpu"] == 16384
    assert math.isfinite(rec["config"]["final_loss"]) and rec["config"]["final_loss"] > 0
    assert rec["value"] > 0 and abs(rec["value"] - 2 * 16384 / (rec["ms_per_step"] * 1e-3)) <= 1e-6 * rec["value"]
    assert rec["cpu_baseline"] is None                     # rank 0 at N = 1 only


def test_trainer_at_world_size_two_checkpoints_and_resumes(tmp_path):
    """The drop-in trainer (`train_deep_sdf.py -e DIR`) at world size 2 on the 4-scene synthetic set (replaces nn.DataParallel,
    train_deep_sdf.py:353): the ScenesPerBatch batch is split over the ranks, every rank owns half of the scenes.  Checked:
    checkpoints hold the FULL latent table and the reference's 2-group optimizer state (train_deep_sdf.py:106-143); the
    trainer's own replica check passed before every checkpoint (decoder parameters and both Adam moments bit-identical on
    the two ranks); `-c latest` resumes at world size 2 -- that run with the two-bucket gradient exchange (DSDF_AR_BUCKETS=2) --, and
    the SAME experiment directory then resumes in ONE process."""
    import json
    from deepsdf_amd import train
    from tests.test_gpu_module_trainer import _make_experiment
    exp = _make_experiment(str(tmp_path), 4, specs_over={"NumEpochs": 2, "SnapshotFrequency": 2, "AdditionalSnapshots": [],
                                                         "LogFrequency": 1, "SamplesPerScene": 1024})
    script = os.path.join(ROOT, "train_deep_sdf.py")

    def check(epoch, files):
        for sub in ("ModelParameters", "OptimizerParameters", "LatentCodes"):
            assert sorted(os.listdir(os.path.join(exp, sub))) == files, sub
        lc = torch.load(os.path.join(exp, "LatentCodes", "latest.pth"), weights_only=True)
        assert lc["epoch"] == epoch and lc["latent_codes"]["weight"].shape == (4, 4)           # all four scenes, both shards
        assert bool((lc["latent_codes"]["weight"].norm(dim=1) > 0).all())
        o = torch.load(os.path.join(exp, "OptimizerParameters", "latest.pth"), weights_only=True)["optimizer_state_dict"]
        assert len(o["param_groups"]) == 2 and o["param_groups"][1]["params"] == [len(o["state"]) - 1]
        last = o["state"][len(o["state"]) - 1]
        assert last["exp_avg"].shape == (4, 4) and bool((last["exp_avg"].abs().sum(dim=1) > 0).all())   # every row was trained
        assert float(last["step"]) == 2.0 * epoch                                                 # 2 steps per epoch on every world size
        mo = torch.load(os.path.join(exp, "ModelParameters", "latest.pth"), weights_only=True)
        assert mo["epoch"] == epoch and all(k.startswith("module.") for k in mo["model_state_dict"])
        logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
        assert logs["epoch"] == epoch and len(logs["loss"]) == 2 * epoch and all(math.isfinite(v) for v in logs["loss"])
        return mo["model_state_dict"], logs

    out = _torchrun_two_ranks([script, "-e", exp])
    assert "training with 2 GPU(s)" in out and out.count("decoder replicas bit-identical on 2 ranks") >= 2, out[-3000:]
    sd2, logs2 = check(2, ["2.pth", "latest.pth"])
    specs = json.load(open(os.path.join(exp, "specs.json")))
    specs["NumEpochs"], specs["SnapshotFrequency"] = 4, 4
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    out = _torchrun_two_ranks([script, "-e", exp, "-c", "latest"], DSDF_AR_BUCKETS="2")   # resume at world size 2, two gradient buckets
    assert "starting from epoch 3" in out and "decoder replicas bit-identical on 2 ranks" in out, out[-3000:]
    sd4, logs4 = check(4, ["2.pth", "4.pth", "latest.pth"])
    assert logs4["loss"][:4] == logs2["loss"]                                     # the first run's log is kept
    assert max(rel_err(sd4[k], sd2[k]) for k in sd2) > 0                          # and training went on
    specs["NumEpochs"], specs["SnapshotFrequency"] = 6, 6
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    torch.manual_seed(5)
    train.main_function(exp, "latest", 1)                                         # the world-2 checkpoint, resumed by ONE process
    sd6, logs6 = check(6, ["2.pth", "4.pth", "6.pth", "latest.pth"])
    assert logs6["loss"][:8] == logs4["loss"] and max(rel_err(sd6[k], sd4[k]) for k in sd4) > 0
    assert sum(logs6["loss"][8:]) / 4 <= 1.5 * sum(logs4["loss"][4:]) / 4         # no restart jump across the world-size change
