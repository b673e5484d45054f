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
from tests.golden_io import rel_err
from tests.hip_helpers import spec_from_meta
from tests.test_gpu_parity import BIG, GRAD_TOL, PARAM_TOL, _safe_batch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_two_rank_hip_step_equals_single_process_and_oracle(tmp_path):
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
                   MASTER_PORT=str(port), DSDF_DIST_BACKEND="gloo", DSDF_SINGLE_DEVICE="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
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
        assert rel_err(lat_dp, o["lat"]) <= PARAM_TOL, step
