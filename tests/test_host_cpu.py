"""CPU: host-side mirror of the reference API -- LR schedules, data pipeline, experiment directory, checkpoint
layout (against the layout the reference itself writes, tests/golden/g9_checkpoint_layout.json)."""
import json
import os

import numpy as np
import pytest
import torch

from tests.golden_io import GOLDEN, Golden

SMALL = dict(dims=[16] * 2, dropout=[0, 1], dropout_prob=0.0, norm_layers=[0, 1], latent_in=(), xyz_in_all=False,
             use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)


def test_lr_schedules_match_golden():
    from deepsdf_amd.train import get_learning_rate_schedules
    d = json.load(open(os.path.join(GOLDEN, "g5_lr_schedules.json")))
    sched = get_learning_rate_schedules({"LearningRateSchedule": d["specs"]})
    for s, row in zip(sched, d["values"]):
        for e, v in zip(d["epochs"], row):
            assert s.get_learning_rate(e) == v
    with pytest.raises(Exception, match='no known learning rate schedule of type "Cosine"'):
        get_learning_rate_schedules({"LearningRateSchedule": [{"Type": "Cosine"}]})


def _write_scene(root, name, n_pos, n_neg, nan_rows=0, dtype=np.float64):
    rng = np.random.default_rng(hash(name) % 1000)
    pos = np.concatenate([rng.uniform(-1, 1, (n_pos, 3)), rng.uniform(0.0, 0.5, (n_pos, 1))], 1).astype(dtype)
    neg = np.concatenate([rng.uniform(-1, 1, (n_neg, 3)), -rng.uniform(0.0, 0.5, (n_neg, 1))], 1).astype(dtype)
    if nan_rows:
        pos[:nan_rows, 3] = np.nan
    d = os.path.join(root, "SdfSamples", "ds", "cls")
    os.makedirs(d, exist_ok=True)
    np.savez(os.path.join(d, name + ".npz"), pos=pos, neg=neg)


def test_data_pipeline(tmp_path, caplog):
    from deepsdf_amd import data
    root = str(tmp_path)
    _write_scene(root, "a", 500, 400, nan_rows=7)
    _write_scene(root, "b", 30, 600, dtype=np.float32)          # positive shortfall
    split = {"ds": {"cls": ["a", "b", "missing"]}}
    files = data.get_instance_filenames(root, split)              # missing files only warn (data.py:23-31)
    assert len(files) == 3 and "non-existent file" in caplog.text
    s = data.unpack_sdf_samples(os.path.join(root, "SdfSamples", files[0]), 3, 100)
    assert s.shape == (100, 4) and s.dtype == torch.float32 and not torch.isnan(s).any()
    assert int((s[:, 3] > 0).sum()) == 50
    s = data.unpack_sdf_samples(os.path.join(root, "SdfSamples", files[1]), 3, 100)
    assert int((s[:, 3] > 0).sum()) == 30 and s.shape[0] == 100    # 30 pos + 70 neg
    full = data.unpack_sdf_samples(os.path.join(root, "SdfSamples", files[0]), 3)
    assert full.shape[0] == 500 - 7 + 400
    pos, neg = data.load_scene(os.path.join(root, "SdfSamples", files[0]), 3)
    assert pos.shape == (493, 4) and neg.shape == (400, 4) and pos.dtype == neg.dtype == torch.float32
    assert not hasattr(data, "SDFSamples")                        # no host-side Dataset: the trainer samples on the device
    cache = data.DeviceSampleCache.from_files(root, files[:2], 3, "cpu")   # bookkeeping only: sampling is a HIP kernel
    assert (cache.n_pos, cache.n_neg) == ([493, 30], [400, 600]) and cache.pos_start == [0, 893] and cache.neg_start == [493, 923]
    assert cache.data.shape == (1523, 4)
    g = torch.Generator().manual_seed(5)
    assert cache.draw_key(g) != cache.draw_key(g)                            # a new key per draw, no device sync
    with pytest.raises(RuntimeError, match="no CPU path"):
        cache.sample(torch.tensor([0, 1, 0]), 100, generator=g)


def test_count_rule_matches_the_reference_loader(tmp_path):
    """golden G12 holds what the REFERENCE's unpack_sdf_samples (deep_sdf/data.py:74-110) returned for balanced /
    shortfall / odd / NaN-filtered scenes.  The specification of the device sampler (oracle.sample_rows), the host count
    rule (data._balanced_counts) and the host mirror of the loader must all reproduce those counts and that order."""
    from deepsdf_amd import data
    from oracle import deepsdf_oracle as orc
    from tests.golden_io import g12_scene_file
    g12 = json.load(open(os.path.join(GOLDEN, "g12_sample_counts.json")))["cases"]
    assert {c["id"] for c in g12} >= {"balanced", "positive_shortfall", "negative_shortfall", "odd_subsample", "nan_rows_make_a_shortfall"}
    for c in g12:
        assert c["positives_then_negatives"] and c["without_replacement"] and c["nan_rows_never_drawn"], c["id"]
        vp, vn = c["n_pos"] - c["nan_pos"], c["n_neg"] - c["nan_neg"]          # rows left after remove_nans
        assert c["rows_without_subsample"] == vp + vn
        f = os.path.join(str(tmp_path), c["id"] + ".npz")
        g12_scene_file(f, c)
        torch.manual_seed(3)
        s = data.unpack_sdf_samples(f, 3, c["subsample"])                      # the host mirror truncates like the reference
        assert s.shape[0] == c["rows_returned"] and int((s[:, 3] > 0).sum()) == c["pos_rows"] and s.dtype == torch.float32, c["id"]
        assert bool((s[:c["pos_rows"], 3] > 0).all()) and not torch.isnan(s).any(), c["id"]
        if vp + vn < 2 * (c["subsample"] // 2):                                 # the reference silently returns FEWER rows
            assert c["rows_returned"] == vp + vn                                # (its collate then fails); the device cache raises
            continue
        assert data._balanced_counts(vp, vn, c["subsample"]) == (c["pos_rows"], c["neg_rows"]), c["id"]
        p, q = orc.sample_rows(vp, vn, c["subsample"], 0x5EED, 0)
        assert (len(p), len(q)) == (c["pos_rows"], c["neg_rows"]), c["id"]
        assert len(set(p.tolist())) == len(p) and len(set(q.tolist())) == len(q) and int(p.max()) < vp and int(q.max()) < vn


def test_sampler_oracle_properties():
    """oracle.sample_perm / sample_rows (the specification of deepsdf_amd/csrc/sample.hpp): a bijection for every length,
    balanced counts with shortfall (deep_sdf/data.py:83-91), different draws for different keys, uniform coverage."""
    from oracle import deepsdf_oracle as orc
    for n in (1, 2, 3, 4, 5, 16, 17, 255, 256, 257, 1000, 4099):
        assert sorted(orc.sample_perm(np.arange(n), n, 0xABCDEF01).tolist()) == list(range(n)), n
    p, q = orc.sample_rows(10000, 9000, 257, 0x1234567890ABCDEF, 7)
    assert len(p) == 128 and len(q) == 128 and len(set(p.tolist())) == 128 and p.max() < 10000 and q.max() < 9000
    p, q = orc.sample_rows(30, 9000, 100, 1, 3)
    assert len(p) == 30 and len(q) == 70 and sorted(p.tolist()) == list(range(30))
    p, q = orc.sample_rows(9000, 12, 100, 1, 3)
    assert len(p) == 88 and len(q) == 12
    a = orc.sample_rows(5000, 5000, 64, 10, 0)[0]
    assert not np.array_equal(a, orc.sample_rows(5000, 5000, 64, 11, 0)[0])      # key
    assert not np.array_equal(a, orc.sample_rows(5000, 5000, 64, 10, 1)[0])      # scene
    cnt = np.zeros(500)
    for k in range(1500):
        cnt[orc.sample_perm(np.arange(50), 500, orc.sample_key(k, 2, 1))] += 1
    assert abs(cnt.mean() - 150) < 1e-9 and cnt.std() < 1.25 * np.sqrt(1500 * 0.1 * 0.9) and cnt.min() > 100


def test_workspace_errors_and_dirs(tmp_path):
    from deepsdf_amd import workspace as ws
    with pytest.raises(Exception, match="does not include specifications file"):
        ws.load_experiment_specifications(str(tmp_path))
    with pytest.raises(Exception, match="does not exist"):
        ws.load_model_parameters(str(tmp_path), "latest", None)
    with pytest.raises(Exception, match="does not include a latent code file"):
        ws.load_latent_vectors(str(tmp_path), "latest")
    assert os.path.isdir(ws.get_model_params_dir(str(tmp_path), True))
    assert ws.get_reconstructed_code_filename("e", 5, "d", "c", "i") == os.path.join("e", "Reconstructions", "5", "Codes", "d", "c", "i.pth")
    import deep_sdf.workspace as ws2
    assert ws2.logs_filename == "Logs.pth" and ws2.specifications_filename == "specs.json"


class _FakeEngine:
    """Host stand-in with the arena interface AdamStateBridge uses (no GPU involved)."""

    def __init__(self, spec):
        self.spec = spec
        self.exp_avg = torch.arange(spec.n_params, dtype=torch.float32)
        self.exp_avg_sq = torch.arange(spec.n_params, dtype=torch.float32) * 2
        self.step = 1

    def view(self, arena, p):
        return arena[p.offset:p.offset + p.numel].view(p.shape)


def test_checkpoint_layout_matches_reference(tmp_path):
    from deepsdf_amd import train
    from deepsdf_amd.decoder import Decoder
    ref = json.load(open(os.path.join(GOLDEN, "g9_checkpoint_layout.json")))
    dec = Decoder(3, **SMALL)
    exp = str(tmp_path)
    train.save_model(exp, "latest.pth", dec, 7)
    m = torch.load(os.path.join(exp, "ModelParameters", "latest.pth"), weights_only=True)
    assert m["epoch"] == 7
    assert {k: list(v.shape) for k, v in m["model_state_dict"].items()} == ref["model_state_dict"]
    assert list(m["model_state_dict"].keys()) == list(ref["model_state_dict"].keys())
    lat = torch.randn(5, 3)
    train.save_latent_vectors(exp, "latest.pth", lat, 7)
    c = torch.load(os.path.join(exp, "LatentCodes", "latest.pth"), weights_only=True)
    assert {k: list(v.shape) for k, v in c["latent_codes"].items()} == ref["latent_codes"]
    eng = _FakeEngine(dec.spec)
    lm, lv = torch.ones(5, 3), torch.ones(5, 3) * 3
    bridge = train.AdamStateBridge(dec, eng, torch.nn.Parameter(lat), lm, lv, 5e-4, 1e-3)
    train.save_optimizer(exp, "latest.pth", bridge, 7)
    o = torch.load(os.path.join(exp, "OptimizerParameters", "latest.pth"), weights_only=True)["optimizer_state_dict"]
    groups = [{k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in g.items()} for g in o["param_groups"]]
    assert groups == ref["optimizer_param_groups"]
    assert {str(i): {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in st.items()} for i, st in o["state"].items()} \
        == ref["optimizer_state"]
    assert float(o["state"][0]["step"]) == ref["optimizer_step_value"] and str(o["state"][0]["step"].dtype) == ref["optimizer_step_dtype"]
    # round trip into a fresh bridge restores the arenas and the step counter
    eng2 = _FakeEngine(dec.spec)
    eng2.exp_avg.zero_(); eng2.exp_avg_sq.zero_(); eng2.step = 0
    lm2, lv2 = torch.zeros(5, 3), torch.zeros(5, 3)
    b2 = train.AdamStateBridge(dec, eng2, torch.nn.Parameter(lat.clone()), lm2, lv2, 0.0, 0.0)
    assert train.load_optimizer(exp, "latest.pth", b2) == 7
    assert eng2.step == 1 and torch.equal(eng2.exp_avg, eng.exp_avg) and torch.equal(lv2, lv)
    with pytest.raises(Exception, match="optimizer state dict .* does not exist"):
        train.load_optimizer(exp, "nope.pth", b2)
    full = torch.zeros(4, 3)
    with pytest.raises(Exception, match="num latent codes mismatched"):
        train.load_latent_vectors(exp, "latest.pth", full)


def test_decoder_state_dict_roundtrip_with_reference_keys():
    from deepsdf_amd.decoder import Decoder
    g = Golden("g1a_tiny_full")
    dec = Decoder(g.meta["L"], **g.meta["net_specs"])
    params = g.group("params0")
    dec.load_state_dict(params)
    sd = dec.state_dict()
    assert list(sd.keys()) == list(params.keys())
    for k in params:
        assert torch.equal(sd[k], params[k])
    # DataParallel-prefixed keys (what the reference's checkpoints carry) load through a DataParallel wrapper
    torch.nn.DataParallel(dec).load_state_dict({"module." + k: v * 2 for k, v in params.items()})
    assert torch.equal(dec.state_dict()["lin0.bias"], params["lin0.bias"] * 2)
    assert dec.geom_dimension == 3
    assert sum(p.numel() for p in dec.parameters()) == dec.spec.n_params
    with pytest.raises(Exception, match="no CPU fallback"):
        dec(torch.zeros(2, g.meta["L"] + 3))


def test_clip_logs_and_magnitudes():
    from deepsdf_amd import train
    ll, lr, tl, lm, pm = train.clip_logs(list(range(12)), [[1, 1]] * 4, [0.1] * 4, [0.5] * 4, {"a": [1, 2, 3, 4]}, 2)
    assert ll == list(range(6)) and len(lr) == 2 and pm == {"a": [1, 2]}
    assert train.get_spec_with_default({"a": 1}, "b", 5) == 5


def test_checkpoint_written_by_the_reference_trainer_loads(tmp_path):
    """Golden G10 = the tensors of an experiment directory the REFERENCE's own main_function wrote (3 epochs, CPU).  The
    drop-in loaders (deepsdf_amd.workspace / train: train_deep_sdf.py:116-131,146-176,202-218, workspace.py:38-51) must
    restore exactly that state, and the oracle's eval forward on the trained weights must reproduce the reference's."""
    from deepsdf_amd import train, workspace as ws
    from deepsdf_amd.decoder import Decoder
    from oracle import deepsdf_oracle as orc
    from tests.golden_io import write_reference_experiment, rel_err
    exp, _, g = write_reference_experiment(str(tmp_path))
    m = g.meta
    specs = ws.load_experiment_specifications(exp)
    L = specs["CodeLength"]
    dec = Decoder(L, **specs["NetworkSpecs"])
    assert ws.load_model_parameters(exp, "latest", torch.nn.DataParallel(dec)) == 3
    sd = dec.state_dict()
    assert ["module." + k for k in sd.keys()] == m["model_keys"]
    for k in sd:
        assert torch.equal(sd[k], g.get("model/module." + k)), k
    lat = torch.zeros(4, L)
    assert train.load_latent_vectors(exp, "latest.pth", lat) == 3 and torch.equal(lat, g.get("latent/weight"))
    eng = _FakeEngine(dec.spec)
    eng.exp_avg.zero_(); eng.exp_avg_sq.zero_(); eng.step = 0
    lm, lv = torch.zeros(4, L), torch.zeros(4, L)
    bridge = train.AdamStateBridge(dec, eng, torch.nn.Parameter(lat.clone()), lm, lv, 0.0, 0.0)
    assert train.load_optimizer(exp, "latest.pth", bridge) == 3
    assert eng.step == 6 == int(g.get("opt0/step"))                      # 2 steps per epoch x 3 epochs
    for i, p in enumerate(dec.spec.params):
        assert torch.equal(eng.view(eng.exp_avg, p), g.get(f"opt{i}/exp_avg")), p.name
        assert torch.equal(eng.view(eng.exp_avg_sq, p), g.get(f"opt{i}/exp_avg_sq")), p.name
    n = len(dec.spec.params)
    assert m["param_groups"][1]["params"] == [n] and torch.equal(lm, g.get(f"opt{n}/exp_avg")) and torch.equal(lv, g.get(f"opt{n}/exp_avg_sq"))
    # and the bridge writes the same layout back: keys, shapes, dtypes, the step's dtype/shape, the group options
    out = bridge.state_dict()
    assert [{k: (list(v) if isinstance(v, (tuple, list)) else v) for k, v in pg.items() if k != "lr"} for pg in out["param_groups"]] \
        == [{k: v for k, v in pg.items() if k != "lr"} for pg in m["param_groups"]]
    assert sorted(out["state"].keys()) == m["opt_state_ids"]
    for i in m["opt_state_ids"]:
        for k in ("step", "exp_avg", "exp_avg_sq"):
            ref = g.get(f"opt{i}/{k}")
            assert out["state"][i][k].dtype == ref.dtype and tuple(out["state"][i][k].shape) == tuple(ref.shape), (i, k)
    assert str(out["state"][0]["step"].dtype) == m["step_dtype"] and list(out["state"][0]["step"].shape) == m["step_shape"]
    ll, lr, tl, lmag, pm, ep = train.load_logs(exp)
    assert ep == 3 and len(ll) == m["n_loss"] == 6 and len(lr) == 3 and list(pm.keys()) == m["param_magnitude_keys"]
    for k in pm:                                                          # the logged magnitudes are the norms of these tensors
        assert abs(pm[k][-1] - float(sd[k].norm())) <= 1e-5 * float(sd[k].norm()), k
    assert lr[2] == [5e-4 * 0.5, 1e-3 * 0.5]
    net = orc.make_net(L, **specs["NetworkSpecs"])
    y = orc.decoder_forward(net, {k: v for k, v in sd.items()}, g.get("eval/x"), training=False)[0].reshape(-1)
    assert rel_err(y, g.get("eval/y")) <= 1e-6


def test_torchscript_export_matches_reference_outputs(tmp_path):
    """SURVEY 8f row f4, export half (create_libtorch_executable.py:4-24): the TorchScript module exported from the HIP
    Decoder's parameters reproduces the reference's eval forward on (a) the REAL trained weights the reference ships as
    cpp_model.pt (golden g6, f(0) = -0.1340, last layer weight-normed + tanh o tanh), (b) the 8x512 architecture (g8),
    (c) the net the reference trainer itself trained (g10); it keeps the reference's state-dict keys and survives
    save -> torch.jit.load, and the root script writes <experiment>/cpp_model.pt from a reference-written checkpoint."""
    import subprocess
    import sys
    from deepsdf_amd.decoder import Decoder
    from deepsdf_amd.export import to_stock_torch
    from oracle import deepsdf_oracle as orc
    from tests.golden_io import write_reference_experiment, rel_err
    g = Golden("g6_real_weights")
    dec = Decoder(g.meta["L"], **g.meta["net_specs"])
    dec.load_state_dict(g.group("params"))
    path = os.path.join(str(tmp_path), "cpp_model.pt")
    sm = dec.export_torchscript(torch.zeros(1, g.meta["L"] + 3), path)
    assert abs(float(sm(torch.zeros(1, g.meta["L"] + 3))) - g.meta["f0_survey"]) < 5e-5
    assert list(sm.state_dict().keys()) == list(dec.state_dict().keys())
    loaded = torch.jit.load(path)
    x = torch.rand(37, g.meta["L"] + 3) * 2 - 1                           # the trace is not specialised to the example's batch size
    assert torch.equal(loaded(x), sm(x)) and loaded(x).shape == (37, 1)
    twin = to_stock_torch(dec)
    with pytest.raises(RuntimeError, match="not a CPU training path"):
        twin.train()
    for name in ("g8_eval_8x512", "g8_eval_6x128"):
        e = Golden(name)
        net = orc.make_net(e.meta["L"], **e.meta["net_specs"])
        d2 = Decoder(e.meta["L"], **e.meta["net_specs"])
        d2.load_state_dict(orc.init_params(net, e.meta["seed"]))
        y = d2.export_torchscript(e.get("in/x")[:1])(e.get("in/x"))
        assert rel_err(y.reshape(-1), e.get("out/y")) <= 1e-6, name
    exp, _, g10 = write_reference_experiment(str(tmp_path / "ref"))
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "create_libtorch_executable.py"), "-e", exp, "-c", "latest"],
                       capture_output=True, text=True, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    m = torch.jit.load(os.path.join(exp, "cpp_model.pt"))
    assert rel_err(m(g10.get("eval/x")).reshape(-1), g10.get("eval/y")) <= 1e-6
