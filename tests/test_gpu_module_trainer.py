"""-m gpu: the nn.Module plugin seam (Decoder under the reference's own step recipe: nn.Embedding(max_norm) +
torch.optim.Adam) and the drop-in trainer (specs.json / experiment directory / checkpoints / resume)."""
import json
import math
import os

import numpy as np
import pytest
import torch

from oracle import deepsdf_oracle as orc
from tests.golden_io import Golden, rel_err

pytestmark = pytest.mark.gpu
FWD_TOL, GRAD_TOL, PARAM_TOL = 1e-5, 1e-4, 1e-5


def test_decoder_autograd_vs_oracle():
    from deepsdf_amd.decoder import Decoder
    g = Golden("g3a_dropout_tiny")
    m = g.meta
    net = orc.make_net(m["L"], **m["net_specs"])
    params = g.group("params0")
    dec = Decoder(m["L"], **m["net_specs"]).cuda()
    dec.load_state_dict(params)
    gen = torch.Generator().manual_seed(0)
    N = 100
    x = torch.cat([torch.randn(N, m["L"], generator=gen) * 0.3, torch.rand(N, 3, generator=gen) * 2 - 1], 1)
    dy = torch.randn(N, 1, generator=gen)
    for training in (False, True):
        dec.train(training)
        xg = x.cuda().requires_grad_(True)
        for p in dec.parameters():
            p.grad = None
        y = dec(xg)
        masks = orc.dropout_masks(net, dec.dropout_seed, dec._fwd_calls, N) if training else None
        yo, sv = orc.decoder_forward(net, params, x, training=training, masks=masks)
        assert rel_err(y.detach().cpu(), yo) <= FWD_TOL
        y.backward(dy.cuda())
        go, dx0 = orc.decoder_backward(net, params, sv, dy, training)
        for name, p in dec.named_parameters():
            assert rel_err(p.grad.cpu(), go[name].reshape(p.shape)) <= GRAD_TOL, (training, name)
        assert rel_err(xg.grad.cpu(), dx0) <= GRAD_TOL
    dec.eval()
    with torch.no_grad():
        y2 = dec(x.cuda())
    yo, _ = orc.decoder_forward(net, params, x, training=False)
    assert rel_err(y2.cpu(), yo) <= FWD_TOL


def test_reference_step_recipe_through_the_plugin_seam():
    """The reference's own loop body (train_deep_sdf.py:505-545) with OUR Decoder dropped in: nn.DataParallel wrapper,
    nn.Embedding(max_norm) on the GPU, L1Loss(sum), torch.optim.Adam with two groups -- against the golden."""
    import deep_sdf  # noqa: F401  (shim import path the reference uses)
    arch = __import__("deep_sdf.networks.deep_sdf_decoder", fromlist=["Decoder"])
    g = Golden("g1a_tiny_full")
    m = g.meta
    decoder = torch.nn.DataParallel(arch.Decoder(m["L"], **m["net_specs"]).cuda())
    decoder.module.load_state_dict(g.group("params0"))
    lat = torch.nn.Embedding(m["S_tot"], m["L"], max_norm=m["code_bound"]).cuda()
    lat.weight.data.copy_(g.get("lat0/w"))
    opt = torch.optim.Adam([{"params": decoder.parameters(), "lr": m["lr"][0]}, {"params": lat.parameters(), "lr": m["lr"][1]}])
    loss_l1 = torch.nn.L1Loss(reduction="sum")
    for si in range(m["n_steps"]):
        i = g.group(f"step{si}/in")
        idx, xyz, gt = i["idx"].cuda(), i["xyz"].cuda(), torch.clamp(i["gt"].cuda(), -m["delta"], m["delta"])
        n = xyz.shape[0]
        decoder.train()
        opt.zero_grad()
        batch_vecs = lat(idx)
        pred = torch.clamp(decoder(torch.cat([batch_vecs, xyz], dim=1)), -m["delta"], m["delta"])
        loss = loss_l1(pred, gt) / n + (m["lam"] * min(1, m["epoch"] / 100) * torch.sum(torch.norm(batch_vecs, dim=1))) / n
        loss.backward()
        opt.step()
        assert abs(loss.item() - float(g.get(f"step{si}/out/loss"))) <= 1e-5 * abs(loss.item())
        for k, ref in g.group(f"step{si}/params_after").items():
            assert rel_err(decoder.module.state_dict()[k].cpu(), ref) <= PARAM_TOL, (si, k)
        assert rel_err(lat.weight.detach().cpu(), g.get(f"step{si}/lat_after/w")) <= PARAM_TOL


def _make_experiment(root, n_scenes, specs_over=None):
    data = os.path.join(root, "data")
    d = os.path.join(data, "SdfSamples", "synth", "spheres")
    os.makedirs(d)
    names = []
    for k in range(n_scenes):
        pos, neg = orc.sphere_scene(k, 20000, unit=(n_scenes == 1))
        np.savez(os.path.join(d, f"s{k}.npz"), pos=pos.astype(np.float64), neg=neg)
        names.append(f"s{k}")
    split = os.path.join(root, "split.json")
    json.dump({"synth": {"spheres": names}}, open(split, "w"))
    exp = os.path.join(root, "exp")
    os.makedirs(exp)
    specs = {
        "Description": "synthetic sphere SDFs (BASELINE config 0 shape)", "DataSource": data, "NetworkArch": "deep_sdf_decoder",
        "TrainSplit": split, "TestSplit": split, "ReconstructionSplit": "",
        "NetworkSpecs": {"dims": [128] * 4, "dropout": [0, 1, 2, 3], "dropout_prob": 0.2, "norm_layers": [0, 1, 2, 3],
                         "latent_in": [2], "xyz_in_all": False, "use_tanh": False, "latent_dropout": False,
                         "weight_norm": True, "geom_dimension": 3},
        "CodeLength": 4, "NumEpochs": 6, "SnapshotFrequency": 3, "AdditionalSnapshots": [1],
        "LearningRateSchedule": [{"Type": "Step", "Initial": 0.0005, "Interval": 500, "Factor": 0.5},
                                 {"Type": "Step", "Initial": 0.001, "Interval": 500, "Factor": 0.5}],
        "SamplesPerScene": 4096, "ScenesPerBatch": min(n_scenes, 2), "DataLoaderThreads": 1, "ClampingDistance": 0.1,
        "CodeRegularization": True, "CodeRegularizationLambda": 1e-4, "CodeBound": 1.0, "LogFrequency": 2}
    specs.update(specs_over or {})
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    return exp


def test_trainer_with_a_staged_sample_cache_equals_the_resident_run(tmp_path, monkeypatch):
    """DSDF_SAMPLE_CACHE_GB=0 forces the trainer's sample cache through pinned host memory (StagedSampleCache: datasets beyond the
    HBM budget); the batches are the resident cache's, so three epochs must end in bit-identical codes and decoder parameters."""
    from deepsdf_amd import train
    outs = []
    for staged in (False, True):
        exp = _make_experiment(os.path.join(str(tmp_path), "staged" if staged else "resident"), 4,
                               specs_over={"NumEpochs": 3, "SnapshotFrequency": 3, "AdditionalSnapshots": [], "LogFrequency": 3})
        if staged:
            monkeypatch.setenv("DSDF_SAMPLE_CACHE_GB", "0")
        else:
            monkeypatch.delenv("DSDF_SAMPLE_CACHE_GB", raising=False)
        torch.manual_seed(0)
        train.main_function(exp, None, 1)
        lc = torch.load(os.path.join(exp, "LatentCodes", "latest.pth"), weights_only=True)["latent_codes"]["weight"]
        mo = torch.load(os.path.join(exp, "ModelParameters", "latest.pth"), weights_only=True)["model_state_dict"]
        outs.append((lc, mo))
    monkeypatch.delenv("DSDF_SAMPLE_CACHE_GB", raising=False)
    assert torch.equal(outs[0][0], outs[1][0])
    for k in outs[0][1]:
        assert torch.equal(outs[0][1][k], outs[1][1][k]), k


def test_trainer_end_to_end_and_resume(tmp_path):
    from deepsdf_amd import train, workspace as ws
    from deepsdf_amd.utils import decode_sdf
    exp = _make_experiment(str(tmp_path), 4)
    torch.manual_seed(0)
    train.main_function(exp, None, 1)
    for sub in ("ModelParameters", "OptimizerParameters", "LatentCodes"):
        assert sorted(os.listdir(os.path.join(exp, sub))) == ["1.pth", "3.pth", "6.pth", "latest.pth"], sub
    logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    assert logs["epoch"] == 6 and len(logs["loss"]) == 6 * 2 and len(logs["learning_rate"]) == 6
    assert set(logs["param_magnitude"]) == {n for n, _ in ws.build_decoder(exp, ws.load_experiment_specifications(exp)).named_parameters()}
    assert all(math.isfinite(v) for v in logs["loss"])
    # resume (-c latest) with more epochs and --batch_split 2: epochs continue, logs extend
    specs = json.load(open(os.path.join(exp, "specs.json")))
    specs["NumEpochs"] = 40
    specs["SnapshotFrequency"] = 40
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    train.main_function(exp, "latest", 2)
    logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    assert logs["epoch"] == 40 and len(logs["loss"]) == 40 * 2
    first, last = sum(logs["loss"][:4]) / 4, sum(logs["loss"][-4:]) / 4
    assert last < 0.8 * first, (first, last)                       # it learns
    o = torch.load(os.path.join(exp, "OptimizerParameters", "40.pth"), weights_only=True)["optimizer_state_dict"]
    assert float(o["state"][0]["step"]) == 80.0 and len(o["param_groups"]) == 2 and o["param_groups"][1]["params"] == [len(o["state"]) - 1]
    # the downstream-consumer path: load_trained_model + load_latent_vectors + decode_sdf (deep_sdf/utils.py:54-65)
    decoder = ws.load_trained_model(exp, "40")
    decoder.eval()
    lat = ws.load_latent_vectors(exp, "40").cuda()
    assert lat.shape == (4, 4) and float(lat.norm(dim=1).max()) < 1.0 + 2e-3
    q = torch.rand(1000, 3, device="cuda") * 2 - 1
    with torch.no_grad():
        sdf = decode_sdf(decoder, lat[0:1], q)
    assert sdf.shape == (1000, 1) and float(sdf.abs().max()) <= 1.0
    with pytest.raises(RuntimeError, match="epoch mismatch"):
        bad = torch.load(os.path.join(exp, "LatentCodes", "latest.pth"), weights_only=True)
        bad["epoch"] = 3
        torch.save(bad, os.path.join(exp, "LatentCodes", "latest.pth"))
        train.main_function(exp, "latest", 1)


def test_latent_only_reconstruction_vs_golden():
    """Config 4: frozen decoder, Adam on the code only -- golden g7 = reference Decoder.eval() + torch Adam."""
    from deepsdf_amd.engine import Engine
    from deepsdf_amd.net import NetSpec
    from deepsdf_amd.reconstruct import reconstruct
    g = Golden("g7_latent_only")
    m = g.meta
    eng = Engine(NetSpec(m["L"], **m["net_specs"]))
    eng.load_params(g.group("params0"))
    grads_before = eng.grads.clone()
    its = [g.group(f"it{i}") for i in range(m["iters"])]

    def feed(it):
        return its[it]["xyz"].cuda().unsqueeze(0), its[it]["gt"].cuda().reshape(1, -1)

    z, _ = reconstruct(eng, its[0]["xyz"].cuda().unsqueeze(0), its[0]["gt"].cuda().reshape(1, -1), num_iterations=m["iters"],
                       clamp_dist=m["delta"], lr=m["lr"], l2reg=m["l2reg"], z0=g.get("z0/z"), lr_drop_every=0, callback=feed)
    assert rel_err(z.cpu(), its[-1]["z_after"]) <= PARAM_TOL
    assert torch.equal(eng.grads, grads_before)          # the frozen decoder's gradient arena is never touched


def test_reconstruction_as_a_captured_graph_equals_the_eager_loop(tmp_path):
    """reconstruct(graph=True): one iteration captured into a HIP graph and replayed (Adam scalars from a device schedule, the
    iteration counter on the device) must give the SAME BITS as the eager loop -- with fixed samples and with a fresh device-side
    subsample per iteration (resample: the draw key follows the device counter), whose batches must also be the ones
    DeviceSampleCache.sample() draws with the same generator."""
    from deepsdf_amd.data import DeviceSampleCache
    from deepsdf_amd.engine import Engine
    from deepsdf_amd.net import NetSpec
    from deepsdf_amd.reconstruct import reconstruct
    kw = dict(dims=[128] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.2, norm_layers=[0, 1, 2, 3], latent_in=[2],
              weight_norm=True, geom_dimension=3)
    eng = Engine(NetSpec(16, **kw))
    eng.init_like_reference(torch.Generator().manual_seed(3))
    gen = torch.Generator().manual_seed(4)
    B, S, iters = 3, 128, 12
    xyz = (torch.rand(B, S, 3, generator=gen) * 2 - 1).cuda()
    sdf = (xyz.norm(dim=2) - torch.tensor([0.4, 0.5, 0.6], device="cuda")[:, None])
    z0 = torch.randn(B, 16, generator=gen) * 0.01
    ze, le = reconstruct(eng, xyz, sdf, num_iterations=iters, z0=z0, lr_drop_every=5, graph=False)
    le = le.clone()
    zg, lg = reconstruct(eng, xyz, sdf, num_iterations=iters, z0=z0, lr_drop_every=5, graph=True)
    torch.cuda.synchronize()
    assert torch.equal(ze, zg) and torch.equal(le, lg) and float((ze - z0.cuda()).abs().max()) > 0
    # fresh samples per iteration from a device cache
    scenes = []
    for k in range(B):
        g = torch.Generator().manual_seed(50 + k)
        p = torch.rand(900, 3, generator=g) * 2 - 1
        d = p.norm(dim=1, keepdim=True) - (0.4 + 0.1 * k)
        rows = torch.cat([p, d], 1)
        scenes.append((rows[rows[:, 3] > 0].contiguous(), rows[rows[:, 3] <= 0].contiguous()))
    ids = torch.arange(B)
    runs = {}
    for mode in (False, True):
        cache = DeviceSampleCache(scenes, 3, "cuda")
        gs = torch.Generator(device="cuda")
        gs.manual_seed(7)
        buf_x, buf_s = torch.empty(B, S, 3, device="cuda"), torch.empty(B, S, device="cuda")
        z, _ = reconstruct(eng, buf_x, buf_s, num_iterations=iters, z0=z0, lr_drop_every=5, graph=mode, resample=(cache, ids, gs))
        torch.cuda.synchronize()
        runs[mode] = z.clone()
        assert cache._draws == iters
    assert torch.equal(runs[False], runs[True])
    # ... and the sequence's draws are sample()'s draws: the eager loop fed by sample() through a callback lands on the same codes
    cache = DeviceSampleCache(scenes, 3, "cuda")
    gs = torch.Generator(device="cuda")
    gs.manual_seed(7)

    def feed(it):
        x, s = cache.sample(ids, S, generator=gs)
        return x.view(B, S, 3), s.view(B, S)

    x0, s0 = feed(0)
    cache._draws = 0                                            # (feed(0) above was only for the shapes)
    zc, _ = reconstruct(eng, x0, s0, num_iterations=iters, z0=z0, lr_drop_every=5, callback=feed)
    assert torch.equal(zc, runs[False])
    with pytest.raises(ValueError, match="callback"):
        reconstruct(eng, x0, s0, num_iterations=iters, z0=z0, callback=feed, graph=True)


def test_batched_reconstruction_matches_single():
    """Many shapes at once (one code each, S a multiple of 64 -> segment-sum path) == each shape alone."""
    from deepsdf_amd.engine import Engine
    from deepsdf_amd.net import NetSpec
    from deepsdf_amd.reconstruct import reconstruct
    kw = dict(dims=[128] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.2, norm_layers=[0, 1, 2, 3], latent_in=[2],
              weight_norm=True, geom_dimension=3)
    eng = Engine(NetSpec(16, **kw))
    eng.init_like_reference(torch.Generator().manual_seed(3))
    gen = torch.Generator().manual_seed(4)
    B, S = 3, 128
    xyz = (torch.rand(B, S, 3, generator=gen) * 2 - 1).cuda()
    sdf = (xyz.norm(dim=2) - torch.tensor([0.4, 0.5, 0.6], device="cuda")[:, None])
    z0 = torch.randn(B, 16, generator=gen) * 0.01
    zb, _ = reconstruct(eng, xyz, sdf, num_iterations=20, z0=z0)
    for b in range(B):
        z1, _ = reconstruct(eng, xyz[b:b + 1], sdf[b:b + 1], num_iterations=20, z0=z0[b:b + 1])
        assert rel_err(zb[b:b + 1].cpu(), z1.cpu()) <= 1e-5


def test_reconstruct_cli_writes_codes(tmp_path):
    """reconstruct.py on an experiment trained by the drop-in trainer: codes land where deep_sdf/workspace.py:137-149 says."""
    import subprocess, sys
    from deepsdf_amd import train
    exp = _make_experiment(str(tmp_path), 4)
    torch.manual_seed(1)
    train.main_function(exp, None, 1)
    root = os.path.dirname(exp)
    r = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "reconstruct.py"),
                        "-e", exp, "-c", "latest", "-d", os.path.join(root, "data"), "-s", os.path.join(root, "split.json"),
                        "--iters", "30", "--samples", "1024"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-2000:]
    d = os.path.join(exp, "Reconstructions", "6", "Codes", "synth", "spheres")
    assert sorted(os.listdir(d)) == ["s0.pth", "s1.pth", "s2.pth", "s3.pth"]
    code = torch.load(os.path.join(d, "s0.pth"), weights_only=True)
    assert code.shape == (1, 1, 4) and bool(torch.isfinite(code).all())


def test_resume_from_a_checkpoint_written_by_the_reference_trainer(tmp_path):
    """SURVEY 8f row f4, checkpoint half.  Golden G10 holds the tensors of an experiment the REFERENCE's main_function wrote
    (train_deep_sdf.py:96-143,179-199).  (1) the downstream-consumer loader + HIP eval forward reproduce the reference's
    outputs on those weights; (2) `-c latest` resumes from it and trains on; (3) what this trainer then writes has the
    reference's keys, shapes, dtypes and optimizer-group options."""
    from deepsdf_amd import train, workspace as ws
    from tests.golden_io import write_reference_experiment
    exp, _, g = write_reference_experiment(str(tmp_path))
    m = g.meta
    decoder = ws.load_trained_model(exp, "latest")
    decoder.eval()
    with torch.no_grad():
        y = decoder(g.get("eval/x").cuda())
    assert rel_err(y.cpu().reshape(-1), g.get("eval/y")) <= 1e-5
    lat = ws.load_latent_vectors(exp, "latest")
    assert torch.equal(lat.cpu(), g.get("latent/weight"))
    specs = json.load(open(os.path.join(exp, "specs.json")))
    specs["NumEpochs"], specs["SnapshotFrequency"] = 5, 5
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    torch.manual_seed(3)
    train.main_function(exp, "latest", 1)
    logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    ref_loss = [float(v) for v in g.get("logs/loss")]
    assert logs["epoch"] == 5 and len(logs["loss"]) == 10 and logs["loss"][:6] == ref_loss     # the reference's log is kept
    assert logs["learning_rate"][:3] == [[float(a) for a in r] for r in g.get("logs/learning_rate")]
    assert logs["learning_rate"][3:] == [[5e-4 * 0.25, 1e-3 * 0.25]] * 2
    assert all(math.isfinite(v) for v in logs["loss"])
    assert sum(logs["loss"][6:]) / 4 <= 1.5 * sum(ref_loss[4:]) / 2                           # training continues, no restart jump
    assert list(logs["param_magnitude"].keys()) == m["param_magnitude_keys"] and all(len(v) == 5 for v in logs["param_magnitude"].values())
    mo = torch.load(os.path.join(exp, "ModelParameters", "5.pth"), weights_only=True)
    assert mo["epoch"] == 5 and list(mo["model_state_dict"].keys()) == m["model_keys"]
    moved = 0.0
    for k, v in mo["model_state_dict"].items():
        ref = g.get("model/" + k)
        assert v.dtype == ref.dtype and v.shape == ref.shape, k
        moved = max(moved, rel_err(v, ref))
    assert 0 < moved < 0.1                                                                     # same run, a few Adam steps further
    o = torch.load(os.path.join(exp, "OptimizerParameters", "5.pth"), weights_only=True)["optimizer_state_dict"]
    assert sorted(o["state"].keys()) == m["opt_state_ids"]
    for i in m["opt_state_ids"]:
        for k in ("step", "exp_avg", "exp_avg_sq"):
            ref = g.get(f"opt{i}/{k}")
            assert o["state"][i][k].dtype == ref.dtype and tuple(o["state"][i][k].shape) == tuple(ref.shape), (i, k)
        assert float(o["state"][i]["step"]) == 10.0                                           # 6 reference steps + 4 of ours
    strip = lambda pg: {k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in pg.items() if k != "lr"}  # noqa: E731
    assert [strip(pg) for pg in o["param_groups"]] == [strip(pg) for pg in m["param_groups"]]
    lc = torch.load(os.path.join(exp, "LatentCodes", "5.pth"), weights_only=True)
    assert lc["epoch"] == 5 and list(lc["latent_codes"].keys()) == ["weight"] and lc["latent_codes"]["weight"].shape == (4, 4)


JVP_CASES = {
    # the shipped small-net shape: last layer weight-normed, use_tanh (tanh o tanh), skip right after layer 0
    "tiny_lastnorm_tanh": dict(L=2, N=333, net=dict(dims=[32] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.2, norm_layers=list(range(8)),
                                                    latent_in=[1], weight_norm=True, use_tanh=True, geom_dimension=3)),
    # latent_in names the output layer: its x0 columns' gradient / tangent bypass the ReLU mask
    "skip_into_the_output_layer": dict(L=6, N=300, net=dict(dims=[64, 64, 72], dropout=[0, 1, 2], dropout_prob=0.2, norm_layers=[0, 1, 2, 3],
                                                             latent_in=[3], weight_norm=True, use_tanh=False, geom_dimension=3)),
    # the headline architecture; N off every tile grid
    "8x512": dict(L=256, N=1000, net=dict(dims=[512] * 8, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)),
                                          latent_in=[4], weight_norm=True, use_tanh=False, geom_dimension=3)),
}


@pytest.mark.parametrize("name", sorted(JVP_CASES))
def test_decoder_jvp_vs_oracle(name):
    """forward-mode derivative d sdf / d input . tangent (deep_sdf/mesh.py:420 pushes latent-space tangents
    through the decoder with torch.autograd.functional.jvp = double backward).  Truth: torch.autograd.functional.jvp of the
    ORACLE's forward in float64 on the CPU.  HIP: (a) Decoder.jvp (one tangent pass, dsdf_module_jvp), (b) the reference's
    own call, torch.autograd.functional.jvp through Decoder.forward (double backward: _DecoderBwdFn), (c) forward-mode AD
    (torch.autograd.forward_ad) -- in eval mode and in train mode (the tangent sees the primal pass's dropout decisions)."""
    import torch.autograd.forward_ad as fwAD
    from deepsdf_amd.decoder import Decoder
    c = JVP_CASES[name]
    L, N = c["L"], c["N"]
    net = orc.make_net(L, **c["net"])
    params = orc.init_params(net, 17)
    p64 = {k: v.double() for k, v in params.items()}
    gen = torch.Generator().manual_seed(18)
    x = torch.cat([torch.randn(N, L, generator=gen) / math.sqrt(L), torch.rand(N, 3, generator=gen) * 2 - 1], 1)
    _, sv = orc.decoder_forward(net, p64, x.double(), training=False, track_margin=True)
    x[sv.min_abs_pre < 1e-6] += 0.01                                    # keep every ReLU decision away from fp32 noise
    tangents = {"latent_only": torch.cat([torch.randn(N, L, generator=gen), torch.zeros(N, 3)], 1),   # mesh.py:420
                "full": torch.randn(N, L + 3, generator=gen)}
    dec = Decoder(L, **c["net"]).cuda()
    dec.load_state_dict(params)
    for training in (False, True):
        dec.train(training)
        for tname, v in tangents.items():
            y, jv = dec.jvp(x.cuda(), v.cuda())
            masks = orc.dropout_masks(net, dec.dropout_seed, dec._fwd_calls, N) if training else None
            f64 = lambda inp: orc.decoder_forward(net, p64, inp, training=training, masks=masks)[0]   # noqa: E731
            yo, jo = torch.autograd.functional.jvp(f64, x.double(), v.double())
            assert rel_err(y.cpu(), yo) <= FWD_TOL, (training, tname)
            e = rel_err(jv.cpu(), jo)
            print(f"jvp {name} training={training} {tname}: rel err vs fp64 oracle {e:.2e}")
            assert e <= 2e-5, (training, tname)
    dec.eval()
    v = tangents["latent_only"]
    yo, jo = torch.autograd.functional.jvp(lambda inp: orc.decoder_forward(net, p64, inp, training=False)[0], x.double(), v.double())
    y2, j2 = torch.autograd.functional.jvp(lambda q: dec(q), x.cuda(), v.cuda())        # (b) the reference's call
    assert rel_err(y2.cpu(), yo) <= FWD_TOL and rel_err(j2.cpu(), jo) <= 2e-5
    with fwAD.dual_level():                                                            # (c) forward-mode AD
        out = dec(fwAD.make_dual(x.cuda(), v.cuda()))
        y3, j3 = fwAD.unpack_dual(out)
    assert j3 is not None and rel_err(y3.cpu(), yo) <= FWD_TOL and rel_err(j3.cpu(), jo) <= 2e-5
    # ordinary backward still works after the double-backward machinery (parameter + input gradients)
    xg = x.cuda().requires_grad_(True)
    dec(xg).sum().backward()
    _, svo = orc.decoder_forward(net, p64, x.double(), training=False)
    go, dx0 = orc.decoder_backward(net, p64, svo, torch.ones(N, 1, dtype=torch.float64), False)
    assert rel_err(xg.grad.cpu(), dx0) <= GRAD_TOL
    for pname, p in dec.named_parameters():
        assert rel_err(p.grad.cpu(), go[pname].reshape(p.shape)) <= GRAD_TOL, pname


@pytest.mark.parametrize("seed", range(12))
def test_random_specs_through_the_module_seam_vs_oracle(seed):
    """The seeded random decoders of tests/test_gpu_parity.py::_random_case through the nn.Module seam (dsdf_module_forward /
    dsdf_module_backward, `decode_sdf`): eval forward, train forward, parameter gradients and d/d(input) against the float64 oracle."""
    from deepsdf_amd.decoder import Decoder
    from tests.test_gpu_parity import _random_case
    c = _random_case(100 + seed)
    L, G = c["L"], c["net"]["geom_dimension"]
    N = c["B"] * c["S"] + 7                                             # off every tile grid
    net = orc.make_net(L, **c["net"])
    params = orc.init_params(net, 500 + seed)
    p64 = {k: v.double() for k, v in params.items()}
    gen = torch.Generator().manual_seed(600 + seed)
    x = torch.cat([torch.randn(N, L, generator=gen) / math.sqrt(L), torch.rand(N, G, generator=gen) * 2 - 1], 1)
    _, sv = orc.decoder_forward(net, p64, x.double(), training=False, track_margin=True)
    x[sv.min_abs_pre < 1e-6] += 0.01                                    # keep every ReLU decision away from fp32 noise
    dec = Decoder(L, **c["net"]).cuda()
    dec.load_state_dict(params)
    dec.eval()
    with torch.no_grad():
        ye = dec(x.cuda())
    yo, _ = orc.decoder_forward(net, p64, x.double(), training=False)
    assert rel_err(ye.cpu(), yo) <= FWD_TOL, (seed, c)
    _decode_sdf_agrees_with_the_module(dec, x)
    for training in (False, True):
        dec.train(training)
        dec.zero_grad(set_to_none=True)
        xg = x.cuda().requires_grad_(True)
        y = dec(xg)
        masks = orc.dropout_masks(net, dec.dropout_seed, dec._fwd_calls, N) if training else None
        yo, svo = orc.decoder_forward(net, p64, x.double(), training=training, masks=masks)
        assert rel_err(y.detach().cpu(), yo) <= FWD_TOL, (seed, training, c)
        w = torch.randn(N, 1, generator=gen)
        (y * w.cuda()).sum().backward()
        go, dx0 = orc.decoder_backward(net, p64, svo, w.double(), training)
        assert rel_err(xg.grad.cpu(), dx0) <= GRAD_TOL, (seed, training, c)
        for pname, p in dec.named_parameters():
            assert rel_err(p.grad.cpu(), go[pname].reshape(p.shape)) <= GRAD_TOL, (pname, seed, training, c)


def _decode_sdf_agrees_with_the_module(dec, x):
    """deep_sdf.utils.decode_sdf (deep_sdf/utils.py:54-65) with ONE code for all query points -- the consumers' call, which takes
    dsdf_decode_latent where the library supports the net -- against Decoder.forward on the materialised [z | q] input."""
    import deep_sdf.utils
    L = dec.spec.latent_size
    z, q = x[0, :L].clone(), x[:, L:].clone()
    with torch.no_grad():
        out = deep_sdf.utils.decode_sdf(dec, z.cuda(), q.cuda())
        ref = dec(torch.cat([z.expand(x.shape[0], -1), q], 1).cuda())
    assert out.shape == ref.shape and rel_err(out.cpu(), ref.cpu()) <= 1e-5


def test_second_order_through_the_decoder_warns_once_and_the_jvp_trick_does_not():
    """The HIP decoder's backward is differentiable w.r.t. the incoming gradient only (what deep_sdf/mesh.py:420 needs); a
    create_graph=True backward with a CONSTANT incoming gradient (gradient penalty / eikonal loss) would silently miss its
    second-order terms, so it warns -- once per process -- and names the stock-torch export twin as the way to do it."""
    import warnings
    import deepsdf_amd.decoder as D
    dec = D.Decoder(5, [64, 64, 64], 3, norm_layers=[0, 1, 2], latent_in=[2], weight_norm=True).cuda().eval()
    x = torch.randn(200, 8, device="cuda")
    D._warned_second_order = False
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # the reference's double-backward jvp must stay silent
        torch.autograd.functional.jvp(lambda q: dec(q), x, torch.randn_like(x))
        xg = x.clone().requires_grad_(True)
        dec(xg).sum().backward()                             # so must an ordinary backward
    assert D._warned_second_order is False
    xg = x.clone().requires_grad_(True)
    with pytest.warns(UserWarning, match="second-order"):
        (g,) = torch.autograd.grad(dec(xg).sum(), xg, create_graph=True)
    assert g.shape == x.shape and D._warned_second_order is True
    with warnings.catch_warnings():
        warnings.simplefilter("error")                       # once per process
        torch.autograd.grad(dec(xg).sum(), xg, create_graph=True)


@pytest.mark.parametrize("name", ["g11a_xyz_in_all", "g11b_latent_dropout", "g11c_layer_norm"])
def test_decoder_variants_module_path_vs_oracle(name):
    """xyz_in_all / latent_dropout / LayerNorm through the nn.Module seam (Decoder.forward + autograd) against the oracle: eval and train
    forward, parameter gradients, and d/d(input) -- which for xyz_in_all collects a d/d(xyz) term from EVERY layer and for
    latent_dropout passes the latent part of layer 0's gradient through the forward's mask."""
    from deepsdf_amd.decoder import Decoder
    g = Golden(name)
    m = g.meta
    L = m["L"]
    net = orc.make_net(L, **m["net_specs"])
    params = g.group("params0")
    p64 = {k: v.double() for k, v in params.items()}
    dec = Decoder(L, **m["net_specs"]).cuda()
    dec.load_state_dict(params)
    gen = torch.Generator().manual_seed(5)
    N = 150
    x = torch.cat([torch.randn(N, L, generator=gen) * 0.3, torch.rand(N, 3, generator=gen) * 2 - 1], 1)
    dy = torch.randn(N, 1, generator=gen)
    for training in (False, True):
        dec.train(training)
        xg = x.cuda().requires_grad_(True)
        for p in dec.parameters():
            p.grad = None
        y = dec(xg)
        masks = orc.dropout_masks(net, dec.dropout_seed, dec._fwd_calls, N) if training else None
        lmask = orc.latent_dropout_mask(net, dec.dropout_seed, dec._fwd_calls, N) if (training and net.latent_dropout) else None
        yo, sv = orc.decoder_forward(net, p64, x.double(), training=training, masks=masks, latent_mask=lmask)
        assert rel_err(y.detach().cpu(), yo) <= FWD_TOL, training
        y.backward(dy.cuda())
        go, dx0 = orc.decoder_backward(net, p64, sv, dy.double(), training)
        for pname, p in dec.named_parameters():
            assert rel_err(p.grad.cpu(), go[pname].reshape(p.shape)) <= GRAD_TOL, (training, pname)
        assert rel_err(xg.grad.cpu()[:, :L], dx0[:, :L]) <= GRAD_TOL, training
        assert rel_err(xg.grad.cpu()[:, L:], dx0[:, L:]) <= GRAD_TOL, training
    dec.eval()
    with torch.no_grad():
        y2 = dec(x.cuda())
    assert rel_err(y2.cpu(), orc.decoder_forward(net, p64, x.double(), training=False)[0]) <= FWD_TOL
    # forward-mode tangent (dsdf_module_jvp) through the variant: eval and train (the primal's dropout / latent-dropout decisions)
    v = torch.randn(N, L + 3, generator=gen)
    for training in (False, True):
        dec.train(training)
        yj, jv = dec.jvp(x.cuda(), v.cuda())
        masks = orc.dropout_masks(net, dec.dropout_seed, dec._fwd_calls, N) if training else None
        lmask = orc.latent_dropout_mask(net, dec.dropout_seed, dec._fwd_calls, N) if (training and net.latent_dropout) else None
        f64 = lambda inp: orc.decoder_forward(net, p64, inp, training=training, masks=masks, latent_mask=lmask)[0]   # noqa: E731
        yo, jo = torch.autograd.functional.jvp(f64, x.double(), v.double())
        assert rel_err(yj.cpu(), yo) <= FWD_TOL and rel_err(jv.cpu(), jo) <= 2e-5, training
    dec.eval()
    # export twin (eval) agrees too
    assert rel_err(dec.export_torchscript(x[:1])(x).detach(), y2.cpu()) <= 1e-5


def test_trainer_runs_a_spec_with_every_decoder_variant(tmp_path):
    """The drop-in trainer on a specs.json that switches on xyz_in_all, latent_dropout and LayerNorm at once: trains, writes
    the bn{i} keys into the checkpoint, resumes from it, and the downstream loader + decode_sdf + TorchScript export work."""
    from deepsdf_amd import train, workspace as ws
    from deepsdf_amd.utils import decode_sdf
    ns = {"dims": [64, 64, 64], "dropout": [0, 1, 2], "dropout_prob": 0.2, "norm_layers": [0, 1, 2, 3], "latent_in": [2],
          "xyz_in_all": True, "use_tanh": False, "latent_dropout": True, "weight_norm": False, "geom_dimension": 3}
    exp = _make_experiment(str(tmp_path), 4, specs_over={"NetworkSpecs": ns, "NumEpochs": 30, "SnapshotFrequency": 30,
                                                         "AdditionalSnapshots": [], "LogFrequency": 10})
    torch.manual_seed(0)
    train.main_function(exp, None, 1)
    logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    assert all(math.isfinite(v) for v in logs["loss"]) and sum(logs["loss"][-4:]) < sum(logs["loss"][:4])
    sd = torch.load(os.path.join(exp, "ModelParameters", "30.pth"), weights_only=True)["model_state_dict"]
    assert "module.bn0.weight" in sd and "module.bn3.bias" in sd and sd["module.lin0.weight"].shape == (61, 7)
    specs = json.load(open(os.path.join(exp, "specs.json")))
    specs["NumEpochs"] = 40
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    train.main_function(exp, "latest", 2)
    dec = ws.load_trained_model(exp, "latest")
    dec.eval()
    lat = ws.load_latent_vectors(exp, "latest").cuda()
    q = torch.rand(500, 3, device="cuda") * 2 - 1
    with torch.no_grad():
        y = decode_sdf(dec, lat[1:2], q)
    x = torch.cat([lat[1:2].expand(500, -1), q], 1).cpu()
    assert rel_err(dec.export_torchscript(x[:1])(x).detach(), y.cpu()) <= 1e-5


@pytest.mark.parametrize("name", ["g11a_xyz_in_all", "g11b_latent_dropout", "g11c_layer_norm"])
def test_decode_sdf_on_each_decoder_variant_alone(name):
    """deep_sdf.utils.decode_sdf (deep_sdf/utils.py:54-65: what meshing and reconstruction call) in eval mode on a decoder that
    has exactly ONE of the three layer-by-layer variants -- LayerNorm alone used to be sent to dsdf_decode_latent, which refuses
    every variant.  The library now answers the question itself (dsdf_decode_latent_supported); these nets take the module path."""
    from deepsdf_amd.decoder import Decoder
    from deepsdf_amd.utils import decode_sdf
    g = Golden(name)
    m = g.meta
    L = m["L"]
    net = orc.make_net(L, **m["net_specs"])
    params = g.group("params0")
    dec = Decoder(L, **m["net_specs"]).cuda().eval()
    dec.load_state_dict(params)
    assert not dec.engine().decode_latent_supported()
    gen = torch.Generator().manual_seed(6)
    z = torch.randn(1, L, generator=gen) * 0.3
    q = torch.rand(777, 3, generator=gen) * 2 - 1
    with torch.no_grad():
        y = decode_sdf(dec, z.cuda(), q.cuda())
    x = torch.cat([z.expand(777, -1), q], 1)
    yo = orc.decoder_forward(net, {k: v.double() for k, v in params.items()}, x.double(), training=False)[0]
    assert y.shape == (777, 1) and rel_err(y.cpu(), yo) <= FWD_TOL
    # and a plain weight-normed decoder does take the single-code entry point
    g8 = Golden("g8_eval_6x128")
    plain = Decoder(g8.meta["L"], **g8.meta["net_specs"]).cuda().eval()
    assert plain.engine().decode_latent_supported()
