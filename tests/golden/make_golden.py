#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ from the REFERENCE itself.

Runs ONLY in the authoring container (needs /root/reference, read-only).  The reference's
``Decoder`` class is imported by file path (it needs nothing but torch) and driven with the exact
torch call sequence of the reference's step body (train_deep_sdf.py:483-545): ``nn.Embedding(max_norm)``
lookup, ``torch.cat``, ``decoder(input)``, ``torch.clamp``, ``L1Loss(sum)/N``, code regulariser,
``backward()``, ``torch.optim.Adam`` with two parameter groups.  Outputs are DATA ONLY (.npz):
inputs + expected outputs.  No reference source, bytecode or TorchScript is written anywhere.

Dropout: the reference's ``F.dropout`` is replaced, for the duration of a forward, by a function that
applies the hash masks of ``oracle.deepsdf_oracle.dropout_keep`` (the HIP kernels' mask spec), so a
training-mode golden is reproducible by any implementation of that integer hash.

Usage:  python tests/golden/make_golden.py          (writes tests/golden/g*.npz)
"""
import importlib.util
import json
import math
import os
import sys
import zipfile

import numpy as np
import torch

sys.dont_write_bytecode = True
HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import deepsdf_oracle as orc  # noqa: E402
from tests.golden_io import sphere_npz  # noqa: E402

REF = "/root/reference"


def ref_decoder_cls():
    spec = importlib.util.spec_from_file_location(
        "ref_deep_sdf_decoder", os.path.join(REF, "deep_sdf/networks/deep_sdf_decoder.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod.Decoder


class MaskInjector:
    """Context manager: F.dropout(x, p, training) -> x * mask/(1-p) with queued masks (call order =
    layer order, deep_sdf_decoder.py:105-106)."""

    def __init__(self, masks):
        self.q = [m for m in masks if m is not None]

    def __enter__(self):
        import torch.nn.functional as F
        self.F = F
        self.orig = F.dropout
        q = self.q

        def fake(x, p=0.5, training=True, inplace=False):
            if not training or p == 0.0:
                return x
            m = q.pop(0)
            assert m.shape == x.shape, (m.shape, x.shape)
            return x * m.to(x.dtype) * (1.0 / (1.0 - p))

        F.dropout = fake
        return self

    def __exit__(self, *a):
        self.F.dropout = self.orig
        assert not self.q, "unused masks"


def build_ref(Decoder, latent_size, net_specs, params):
    dec = Decoder(latent_size, **net_specs)
    sd = dec.state_dict()
    assert list(sd.keys()) == list(params.keys()), (list(sd.keys()), list(params.keys()))
    dec.load_state_dict({k: v.clone() for k, v in params.items()})
    return dec


def ref_step(dec, lat, opt, batches, *, delta, lam, epoch, code_reg, masks_per_chunk, train=True, grad_clip=None):
    """train_deep_sdf.py:483-545 for one optimiser step.  ``batches`` = list of (indices, xyz, sdf_gt) chunks
    (already chunked as torch.chunk would, :495-501)."""
    n_total = sum(b[1].shape[0] for b in batches)
    dec.train(train)
    opt.zero_grad()
    loss_l1 = torch.nn.L1Loss(reduction="sum")
    batch_loss = 0.0
    ys = []
    for ci, (idx, xyz, gt) in enumerate(batches):
        gt = torch.clamp(gt, -delta, delta)
        batch_vecs = lat(idx)
        inp = torch.cat([batch_vecs, xyz], dim=1)
        masks = masks_per_chunk[ci] if masks_per_chunk is not None else []
        with MaskInjector(masks):
            pred = dec(inp)
        ys.append(pred.detach().clone())
        pred = torch.clamp(pred, -delta, delta)
        chunk_loss = loss_l1(pred, gt) / n_total
        if code_reg:
            l2 = torch.sum(torch.norm(batch_vecs, dim=1))
            chunk_loss = chunk_loss + (lam * min(1, epoch / 100) * l2) / n_total
        chunk_loss.backward()
        batch_loss += chunk_loss.item()
    # (a parameter forward never touches -- the bn module of the LAST Linear in the LayerNorm variant -- has grad None)
    grads = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in dec.named_parameters()}
    dlat = lat.weight.grad.detach().clone()
    gn = None
    if grad_clip is not None:
        gn = torch.nn.utils.clip_grad_norm_(dec.parameters(), grad_clip)
        grads_clipped = {k: (p.grad.detach().clone() if p.grad is not None else torch.zeros_like(p)) for k, p in dec.named_parameters()}
    else:
        grads_clipped = None
    opt.step()
    return batch_loss, torch.cat(ys), grads, dlat, gn, grads_clipped


def pack(prefix, d):
    # .clone(): live parameters / Adam buffers keep mutating after this call; snapshot them
    return {f"{prefix}/{k}": (v.detach().clone().numpy() if torch.is_tensor(v) else np.array(v)) for k, v in d.items()}


def synth_batch(gen, scenes, S, G=3):
    """[B scenes] x S points: xyz ~ U(-1,1), sdf = ||x - c|| - r (clamp exercised: many |sdf| < 0.1)."""
    idx, xyz, gt = [], [], []
    for s in scenes:
        c = (torch.rand(G, generator=gen) - 0.5) * 0.6
        r = 0.3 + 0.3 * torch.rand(1, generator=gen)
        half = S // 2
        box = torch.rand(half, G, generator=gen) * 2 - 1
        d = torch.randn(S - half, G, generator=gen)
        d = d / d.norm(dim=1, keepdim=True)
        surf = c + r * d + 0.05 * torch.randn(S - half, G, generator=gen)
        p = torch.cat([box, surf])
        xyz.append(p)
        gt.append((p - c).norm(dim=1, keepdim=True) - r)
        idx.append(torch.full((S,), s, dtype=torch.int64))
    return torch.cat(idx), torch.cat(xyz).float(), torch.cat(gt).float()


def margins_ok(dec_net, params, x0, gt, delta, masks=None, training=False, tol=2e-5, latent_mask=None):
    """Reject batches where a ReLU pre-activation or |y|-delta sits within tol of its threshold."""
    p64 = {k: v.double() for k, v in params.items()}
    _, sv = orc.decoder_forward(dec_net, p64, x0.double(), training=training, masks=masks, track_margin=True,
                                latent_mask=latent_mask)
    # clamp boundary and sign(pred - gt) flips switch a whole point's gradient: keep a hard margin there
    if ((sv.y.abs() - delta).abs() < tol).any():
        return False
    if gt is not None:
        diff = torch.clamp(sv.y, -delta, delta) - torch.clamp(gt.double().reshape(-1, 1), -delta, delta)
        if ((diff != 0) & (diff.abs() < tol)).any():
            return False
    # a ReLU flip only moves one unit of one point (forward is continuous): checked on small nets only
    n_pre = sum(ly.out_dim for ly in dec_net.layers[:-1]) * x0.shape[0]
    if n_pre <= 50000 and sv.min_abs_pre is not None and (sv.min_abs_pre < 1e-6).any():
        return False
    return True


def case_train(Decoder, name, *, L, net_specs, S_tot, scenes_steps, S, seed, delta=0.1, lam=1e-4,
               epoch=37, code_bound=1.0, code_reg=True, lr=(5e-4, 1e-3), batch_split=1, dropout_train=False,
               drop_seed=0, grad_clip=None, store="full", oversize_row=None):
    """Run len(scenes_steps) optimiser steps on the reference; store inputs + expected outputs."""
    net = orc.make_net(L, **net_specs)
    params = orc.init_params(net, seed)
    gen = torch.Generator().manual_seed(seed + 1)
    lat0 = torch.randn(S_tot, L, generator=gen) * (1.0 / math.sqrt(L))
    if oversize_row is not None:
        lat0[oversize_row] *= 3.0 / lat0[oversize_row].norm()   # norm 3 > CodeBound -> renorm must fire
    dec = build_ref(Decoder, L, net_specs, params)
    lat = torch.nn.Embedding(S_tot, L, max_norm=code_bound)
    lat.weight.data.copy_(lat0)
    opt = torch.optim.Adam([{"params": dec.parameters(), "lr": lr[0]}, {"params": lat.parameters(), "lr": lr[1]}])
    out = {"meta": np.frombuffer(json.dumps(dict(
        L=L, net_specs=net_specs, S_tot=S_tot, S=S, seed=seed, delta=delta, lam=lam, epoch=epoch,
        code_bound=code_bound, code_reg=code_reg, lr=list(lr), batch_split=batch_split,
        dropout_train=dropout_train, drop_seed=drop_seed, grad_clip=grad_clip, store=store,
        n_steps=len(scenes_steps))).encode(), dtype=np.uint8)}
    out.update(pack("lat0", {"w": lat0}))
    if store == "full":
        out.update(pack("params0", params))
    for si, scenes in enumerate(scenes_steps):
        for attempt in range(20):
            idx, xyz, gt = synth_batch(gen, scenes, S)
            chunks = list(zip(torch.chunk(idx, batch_split), torch.chunk(xyz, batch_split), torch.chunk(gt, batch_split)))
            masks_pc, row0 = None, 0
            lat_drop = dropout_train and net.latent_dropout
            if dropout_train:
                masks_pc = []
                for (ci, xc, gc) in chunks:
                    ms = orc.dropout_masks(net, drop_seed, si, xc.shape[0], row_offset=row0)
                    if lat_drop:   # the reference's FIRST F.dropout call of a forward is the latent one (deep_sdf_decoder.py:81)
                        ms = [orc.latent_dropout_mask(net, drop_seed, si, xc.shape[0], row_offset=row0)] + ms
                    masks_pc.append(ms)
                    row0 += xc.shape[0]
            # margin check on the concatenated batch with the CURRENT reference params
            cur = {k: v.detach().clone() for k, v in dec.state_dict().items()}
            lat_probe = lat.weight.detach().clone()
            orc.renorm_rows_(lat_probe, idx, code_bound)
            x0 = torch.cat([lat_probe[idx], xyz], 1)
            mm, lm = None, None
            if dropout_train:  # chunk masks continue the row counter, so their concatenation is this
                mm = orc.dropout_masks(net, drop_seed, si, xyz.shape[0], row_offset=0)
                lm = orc.latent_dropout_mask(net, drop_seed, si, xyz.shape[0]) if lat_drop else None
            if margins_ok(net, cur, x0, gt, delta, masks=mm, training=dropout_train, latent_mask=lm):
                break
        else:
            raise RuntimeError("could not draw a batch with safe margins")
        loss, y, grads, dlat, gn, gclip = ref_step(dec, lat, opt, chunks, delta=delta, lam=lam, epoch=epoch,
                                                   code_reg=code_reg, masks_per_chunk=masks_pc,
                                                   train=dropout_train or True, grad_clip=grad_clip)
        pre = f"step{si}"
        out.update(pack(pre + "/in", {"idx": idx, "xyz": xyz, "gt": gt}))
        out.update(pack(pre + "/out", {"loss": np.float64(loss), "y": y.reshape(-1)}))
        out.update(pack(pre + "/dlat", {"w": dlat}))
        out.update(pack(pre + "/lat_after", {"w": lat.weight.detach()}))
        if gn is not None:
            out.update(pack(pre + "/out", {"grad_norm": gn}))
        if store == "full":
            out.update(pack(pre + "/grads", grads))
            out.update(pack(pre + "/params_after", dict(dec.state_dict())))
            st = opt.state_dict()["state"]
            names = list(dict(dec.named_parameters()).keys()) + ["latent"]
            shapes = [p_.shape for p_ in dec.parameters()] + [lat.weight.shape]
            for i, nme in enumerate(names):   # (torch.optim.Adam keeps no state for a parameter that never had a gradient)
                out.update(pack(f"{pre}/adam_m", {nme: st[i]["exp_avg"] if i in st else torch.zeros(shapes[i])}))
                out.update(pack(f"{pre}/adam_v", {nme: st[i]["exp_avg_sq"] if i in st else torch.zeros(shapes[i])}))
        else:  # "slice": small tensors in full, big ones as corner + Frobenius norm
            for k, g in grads.items():
                if g.numel() <= 4096:
                    out.update(pack(pre + "/grads", {k: g}))
                else:
                    out.update(pack(pre + "/grads_corner", {k: g[:8, :8]}))
                    out.update(pack(pre + "/grads_fro", {k: g.double().norm()}))
            for k, p_ in dec.state_dict().items():
                if p_.numel() <= 4096:
                    out.update(pack(pre + "/params_after", {k: p_}))
                else:
                    out.update(pack(pre + "/params_after_corner", {k: p_[:8, :8]}))
                    out.update(pack(pre + "/params_after_fro", {k: p_.double().norm()}))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, {k: v.shape for k, v in list(out.items())[:0]})


def case_forward_eval(Decoder, name, *, L, net_specs, N, seed):
    """Eval-mode forward only (decode_sdf semantics, deep_sdf/utils.py:54-65)."""
    net = orc.make_net(L, **net_specs)
    params = orc.init_params(net, seed)
    gen = torch.Generator().manual_seed(seed + 7)
    x = torch.cat([torch.randn(N, L, generator=gen) / math.sqrt(L), torch.rand(N, 3, generator=gen) * 2 - 1], 1)
    dec = build_ref(Decoder, L, net_specs, params).eval()
    with torch.no_grad():
        y = dec(x)
    out = {"meta": np.frombuffer(json.dumps(dict(L=L, net_specs=net_specs, N=N, seed=seed)).encode(), dtype=np.uint8)}
    out.update(pack("in", {"x": x}))
    out.update(pack("out", {"y": y.reshape(-1)}))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def ref_trainer_module():
    """The reference's train_deep_sdf.py imported as a module (container only).  Its package __init__ star-imports mesh /
    metrics modules whose third-party dependencies are not installed and are not on the hot path: empty stub modules are
    registered for them first; nothing from them is ever called."""
    import types
    for name in ("plyfile", "skimage", "skimage.measure", "splinepy", "trimesh"):
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    if REF not in sys.path:
        sys.path.insert(0, REF)              # `import deep_sdf` inside the reference trainer must find the REFERENCE package
    for k in [k for k in sys.modules if k == "deep_sdf" or k.startswith("deep_sdf.")]:
        del sys.modules[k]
    spec = importlib.util.spec_from_file_location("ref_train_deep_sdf", os.path.join(REF, "train_deep_sdf.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert os.path.realpath(sys.modules["deep_sdf"].__file__).startswith(REF), sys.modules["deep_sdf"].__file__
    return mod


def case_lr(name):
    """G5: learning-rate schedule values produced by the REFERENCE's own schedule classes
    (train_deep_sdf.py:23-93: get_learning_rate_schedules -> Step / Warmup / Constant .get_learning_rate)."""
    ref = ref_trainer_module()
    specs = [
        {"Type": "Step", "Initial": 0.0005, "Interval": 500, "Factor": 0.5},
        {"Type": "Step", "Initial": 0.001, "Interval": 500, "Factor": 0.5},
        {"Type": "Warmup", "Initial": 1e-5, "Final": 1e-3, "Length": 500},
        {"Type": "Constant", "Value": 3e-4},
    ]
    epochs = [0, 1, 499, 500, 501, 1000, 2001]
    schedules = ref.get_learning_rate_schedules({"LearningRateSchedule": specs})
    vals = [[float(sch.get_learning_rate(e)) for e in epochs] for sch in schedules]
    try:
        ref.get_learning_rate_schedules({"LearningRateSchedule": [{"Type": "Cosine"}]})
        unknown = None
    except Exception as e:                       # the error text the drop-in trainer mirrors (:86-91)
        unknown = str(e)
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump({"specs": specs, "epochs": epochs, "values": vals, "unknown_type_error": unknown,
                   "source": "reference train_deep_sdf.get_learning_rate_schedules (imported, container only)"}, f, indent=1)
    print("wrote", name)


def case_reference_run(name):
    """G10: an experiment directory WRITTEN BY THE REFERENCE TRAINER: its own main_function (train_deep_sdf.py:255-581)
    runs 3 epochs on a synthetic 4-scene data set in a temp dir (CPU), then the TENSORS of ModelParameters/latest.pth,
    OptimizerParameters/latest.pth, LatentCodes/latest.pth and Logs.pth are stored as arrays (data only: no pickle, no
    reference code) together with an eval-mode forward of the trained decoder.  Pins f4's checkpoint half:
    tests rebuild the .pth files from these arrays and resume from them with this repo's trainer."""
    import tempfile
    ref = ref_trainer_module()
    import deep_sdf.workspace as rws           # the REFERENCE's workspace module (asserted in ref_trainer_module)
    tmp = tempfile.mkdtemp(prefix="g10_")
    data = os.path.join(tmp, "data")
    os.makedirs(os.path.join(data, "SdfSamples", "synth", "spheres"))
    names = [f"s{k}" for k in range(4)]
    for k, nme in enumerate(names):
        sphere_npz(os.path.join(data, "SdfSamples", "synth", "spheres", nme + ".npz"), k)
    split = os.path.join(tmp, "split.json")
    json.dump({"synth": {"spheres": names}}, open(split, "w"))
    exp = os.path.join(tmp, "exp")
    os.makedirs(exp)
    net_specs = dict(dims=[32] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.2, norm_layers=[0, 1, 2, 3], latent_in=[2],
                     xyz_in_all=False, use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)
    specs = {"Description": "g10 reference run", "DataSource": data, "TrainSplit": split, "TestSplit": split,
             "ReconstructionSplit": split, "NetworkArch": "deep_sdf_decoder", "NetworkSpecs": net_specs, "CodeLength": 4,
             "NumEpochs": 3, "SnapshotFrequency": 3, "AdditionalSnapshots": [],
             "LearningRateSchedule": [{"Type": "Step", "Initial": 0.0005, "Interval": 2, "Factor": 0.5},
                                      {"Type": "Step", "Initial": 0.001, "Interval": 2, "Factor": 0.5}],
             "SamplesPerScene": 512, "ScenesPerBatch": 2, "DataLoaderThreads": 0, "ClampingDistance": 0.1,
             "CodeRegularization": True, "CodeRegularizationLambda": 1e-4, "CodeBound": 1.0, "LogFrequency": 1}
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    torch.manual_seed(1010)
    ref.main_function(exp, None, 1)
    files = {sub: sorted(os.listdir(os.path.join(exp, sub))) for sub in ("ModelParameters", "OptimizerParameters", "LatentCodes")}
    # files our own run of the reference wrote a moment ago: plain dicts of tensors / lists
    m = torch.load(os.path.join(exp, "ModelParameters", "latest.pth"), weights_only=True)
    o = torch.load(os.path.join(exp, "OptimizerParameters", "latest.pth"), weights_only=True)
    lc = torch.load(os.path.join(exp, "LatentCodes", "latest.pth"), weights_only=True)
    lg = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    dec = rws.load_trained_model(exp, "latest").eval()      # the reference's own loader (workspace.py:212-242)
    gen = torch.Generator().manual_seed(5)
    x = torch.cat([lc["latent_codes"]["weight"][torch.arange(64) % 4], torch.rand(64, 3, generator=gen) * 2 - 1], 1)
    with torch.no_grad():
        y = dec(x)
    osd = o["optimizer_state_dict"]
    meta = dict(specs={k: v for k, v in specs.items() if k not in ("DataSource", "TrainSplit", "TestSplit", "ReconstructionSplit")},
                scene_names=names, files=files, epochs=dict(model=m["epoch"], optimizer=o["epoch"], latent=lc["epoch"], logs=lg["epoch"]),
                model_keys=list(m["model_state_dict"].keys()),
                param_groups=[{k: (list(v) if isinstance(v, (list, tuple)) else v) for k, v in g.items()} for g in osd["param_groups"]],
                opt_state_ids=[int(i) for i in osd["state"].keys()],
                step_dtype=str(osd["state"][0]["step"].dtype), step_shape=list(osd["state"][0]["step"].shape),
                param_magnitude_keys=list(lg["param_magnitude"].keys()), n_loss=len(lg["loss"]),
                log_value_types=dict(loss=type(lg["loss"][0]).__name__, timing=type(lg["timing"][0]).__name__,
                                     latent_magnitude=type(lg["latent_magnitude"][0]).__name__,
                                     learning_rate=type(lg["learning_rate"][0]).__name__))
    out = {"meta": np.frombuffer(json.dumps(meta).encode(), dtype=np.uint8)}
    out.update(pack("model", m["model_state_dict"]))
    for i, st in osd["state"].items():
        out.update(pack(f"opt{int(i)}", {k: v for k, v in st.items()}))
    out.update(pack("latent", lc["latent_codes"]))
    out.update(pack("logs", {"loss": torch.tensor(lg["loss"], dtype=torch.float64),
                             "learning_rate": torch.tensor(lg["learning_rate"], dtype=torch.float64),
                             "timing": torch.tensor(lg["timing"], dtype=torch.float64),
                             "latent_magnitude": torch.stack([torch.as_tensor(v) for v in lg["latent_magnitude"]]).double()}))
    out.update(pack("logs_pm", {k: torch.tensor([float(v) for v in vs], dtype=torch.float64) for k, vs in lg["param_magnitude"].items()}))
    out.update(pack("eval", {"x": x, "y": y.reshape(-1)}))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    import shutil
    shutil.rmtree(tmp)
    print("wrote", name, "epochs", meta["epochs"], "files", files)


def case_real_weights(name):
    """G6: real trained weights shipped by the reference
    (experiments/corner_spheres_only_small_network/cpp_model.pt).  torch.load(weights_only=True) REFUSES
    TorchScript archives, and torch.jit.load would deserialise code from the file, so the raw tensor
    storages are read as plain bytes with zipfile (nothing from the file is executed).  Storage order
    follows state_dict order (bias, g, v per layer); the assignment is validated against the per-parameter
    norms the reference logged at the same epoch (Logs.pth 'param_magnitude', safe-loaded)."""
    exp = os.path.join(REF, "experiments/corner_spheres_only_small_network")
    specs = json.load(open(os.path.join(exp, "specs.json")))
    L = specs["CodeLength"]
    ns = dict(specs["NetworkSpecs"])
    net = orc.make_net(L, **ns)
    names = orc.param_names(net)
    shapes = {}
    for l, ly in enumerate(net.layers):
        shapes[f"lin{l}.bias"] = (ly.out_dim,)
        shapes[f"lin{l}.parametrizations.weight.original0"] = (ly.out_dim, 1)
        shapes[f"lin{l}.parametrizations.weight.original1"] = (ly.out_dim, ly.in_dim)
        shapes[f"lin{l}.weight"] = (ly.out_dim, ly.in_dim)
    z = zipfile.ZipFile(os.path.join(exp, "cpp_model.pt"))
    params = {}
    for i, nme in enumerate(names):
        raw = np.frombuffer(z.read(f"cpp_model/data/{i}"), dtype="<f4")
        assert raw.size == int(np.prod(shapes[nme])), (nme, raw.size, shapes[nme])
        params[nme] = torch.from_numpy(raw.reshape(shapes[nme]).copy())
    logs = torch.load(os.path.join(exp, "Logs.pth"), weights_only=True)
    pm = logs["param_magnitude"]
    check = {}
    for nme in names:
        logged = float(pm[nme][-1])
        mine = float(params[nme].norm())
        check[nme] = (logged, mine)
    y0, _ = orc.decoder_forward(net, params, torch.zeros(1, L + ns["geom_dimension"]), training=False)
    out = {"meta": np.frombuffer(json.dumps(dict(L=L, net_specs=ns, f0_survey=-0.1340,
                                                 logged_norms={k: v[0] for k, v in check.items()})).encode(), dtype=np.uint8)}
    out.update(pack("params", params))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name, "f(0)=", float(y0), "norm check (logged, extracted):")
    for k, v in check.items():
        print("   ", k, v)


def case_latent_only(Decoder, name, *, L, net_specs, N, iters, seed, delta=0.1, lr=5e-3, l2reg=1e-4):
    """G7 (config 4 / SURVEY a9): frozen eval-mode reference Decoder + torch.optim.Adam([z])."""
    net = orc.make_net(L, **net_specs)
    params = orc.init_params(net, seed)
    dec = build_ref(Decoder, L, net_specs, params).eval()
    for p in dec.parameters():
        p.requires_grad_(False)
    gen = torch.Generator().manual_seed(seed + 3)
    z = (torch.randn(1, L, generator=gen) * 0.01).requires_grad_(True)
    z0 = z.detach().clone()
    opt = torch.optim.Adam([z], lr=lr)
    out = {"meta": np.frombuffer(json.dumps(dict(L=L, net_specs=net_specs, N=N, iters=iters, seed=seed,
                                                 delta=delta, lr=lr, l2reg=l2reg)).encode(), dtype=np.uint8)}
    out.update(pack("params0", params))
    out.update(pack("z0", {"z": z0}))
    l1 = torch.nn.L1Loss()
    for it in range(iters):
        _, xyz, gt = synth_batch(gen, [0], N)
        opt.zero_grad()
        inp = torch.cat([z.expand(N, -1), xyz], 1)
        pred = torch.clamp(dec(inp), -delta, delta)
        loss = l1(pred, torch.clamp(gt, -delta, delta)) + l2reg * torch.mean(z.pow(2))
        loss.backward()
        opt.step()
        out.update(pack(f"it{it}", {"xyz": xyz, "gt": gt, "loss": np.float64(loss.item()),
                                    "dz": z.grad.detach().clone(), "z_after": z.detach().clone()}))
    np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
    print("wrote", name)


def case_checkpoint_layout(Decoder, name):
    """G9: the KEY LAYOUT of the reference's checkpoints (train_deep_sdf.py:96-143): model_state_dict under
    nn.DataParallel, optimizer_state_dict of Adam with the two param groups after one step, latent_codes."""
    specs = dict(dims=[16] * 2, dropout=[0, 1], dropout_prob=0.0, norm_layers=[0, 1], latent_in=(), xyz_in_all=False,
                 use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)
    dec = torch.nn.DataParallel(Decoder(3, **specs))
    lat = torch.nn.Embedding(5, 3, max_norm=1.0)
    opt = torch.optim.Adam([{"params": dec.parameters(), "lr": 5e-4}, {"params": lat.parameters(), "lr": 1e-3}])
    x = torch.cat([lat(torch.tensor([0, 1, 1])), torch.rand(3, 3)], 1)
    dec(x).sum().backward()
    opt.step()
    osd = opt.state_dict()
    out = {
        "model_state_dict": {k: list(v.shape) for k, v in dec.state_dict().items()},
        "latent_codes": {k: list(v.shape) for k, v in lat.state_dict().items()},
        "optimizer_param_groups": [{k: (v if k != "params" else list(v)) for k, v in g.items()} for g in osd["param_groups"]],
        "optimizer_state": {str(i): {k: (list(v.shape) if torch.is_tensor(v) else v) for k, v in st.items()}
                            for i, st in osd["state"].items()},
        "optimizer_step_value": float(osd["state"][0]["step"]),
        "optimizer_step_dtype": str(osd["state"][0]["step"].dtype),
    }
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump(out, f, indent=1, default=lambda o: list(o) if isinstance(o, tuple) else str(o))
    print("wrote", name)


def case_sample_counts(name):
    """G12: the per-scene subsampling COUNT RULE of the reference's own loader (deep_sdf/data.py:74-110, unpack_sdf_samples:
    half of the subsample from each sign, a shortfall of one sign taken from the other, NaN rows dropped first, torch.randperm
    = without replacement, positives then negatives).  The reference function is CALLED here on synthetic .npz files whose rows
    carry their own identity (x = row index within its sign, sdf = +/-(index + 1)); what it returned is stored as counts.
    Pins f1's count rule: oracle.sample_rows / deepsdf_amd.data._balanced_counts / DeviceSampleCache.sample must reproduce them."""
    import tempfile
    ref_trainer_module()
    rdata = sys.modules["deep_sdf"].data
    assert os.path.realpath(rdata.__file__).startswith(REF)
    cases = [dict(id="balanced", n_pos=500, n_neg=700, nan_pos=0, nan_neg=0, subsample=256, dtype="float32"),
             dict(id="positive_shortfall", n_pos=40, n_neg=900, nan_pos=0, nan_neg=0, subsample=256, dtype="float32"),
             dict(id="negative_shortfall", n_pos=900, n_neg=17, nan_pos=0, nan_neg=0, subsample=256, dtype="float64"),
             dict(id="odd_subsample", n_pos=300, n_neg=310, nan_pos=0, nan_neg=0, subsample=255, dtype="float32"),
             dict(id="exactly_half", n_pos=128, n_neg=128, nan_pos=0, nan_neg=0, subsample=256, dtype="float32"),
             dict(id="nan_rows_make_a_shortfall", n_pos=100, n_neg=400, nan_pos=70, nan_neg=5, subsample=128, dtype="float64"),
             dict(id="both_signs_short", n_pos=40, n_neg=50, nan_pos=0, nan_neg=0, subsample=256, dtype="float32")]
    out = []
    with tempfile.TemporaryDirectory() as d:
        for c in cases:
            def rows(n, n_nan, sign):
                i = np.arange(n, dtype=np.float64)
                r = np.stack([i, np.zeros(n), np.full(n, float(sign)), sign * (i + 1)], 1)
                r[:n_nan, 3] = np.nan                     # the first n_nan rows are NaN samples (remove_nans drops them)
                return r.astype(c["dtype"])
            f = os.path.join(d, c["id"] + ".npz")
            np.savez(f, pos=rows(c["n_pos"], c["nan_pos"], 1), neg=rows(c["n_neg"], c["nan_neg"], -1))
            torch.manual_seed(1)
            s = rdata.unpack_sdf_samples(f, 3, c["subsample"])
            sd = s[:, 3]
            n_p = int((sd > 0).sum())
            pos_first = bool((sd[:n_p] > 0).all() and (sd[n_p:] < 0).all())
            ids_p, ids_n = s[:n_p, 0].long().tolist(), s[n_p:, 0].long().tolist()
            full = rdata.unpack_sdf_samples(f, 3)          # subsample=None: everything, NaN rows removed
            out.append(dict(c, rows_returned=int(s.shape[0]), pos_rows=n_p, neg_rows=int(s.shape[0]) - n_p,
                            positives_then_negatives=pos_first, dtype_returned=str(s.dtype).replace("torch.", ""),
                            without_replacement=len(set(ids_p)) == len(ids_p) and len(set(ids_n)) == len(ids_n),
                            nan_rows_never_drawn=bool(min(ids_p, default=10**9) >= c["nan_pos"] and min(ids_n, default=10**9) >= c["nan_neg"]),
                            rows_without_subsample=int(full.shape[0])))
    with open(os.path.join(HERE, name + ".json"), "w") as f:
        json.dump({"cases": out, "source": "reference deep_sdf.data.unpack_sdf_samples (imported, container only), geom_dimension 3"},
                  f, indent=1)
    print("wrote", name)


def main():
    torch.set_num_threads(4)
    Decoder = ref_decoder_cls()
    wn4 = dict(dims=[64] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.0, norm_layers=[0, 1, 2, 3], latent_in=[2],
               xyz_in_all=False, use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)
    # G1a: tiny-full, 2 steps; step 1 omits scene 2 (dense Adam on an absent row); row 1 starts above CodeBound
    case_train(Decoder, "g1a_tiny_full", L=4, net_specs=wn4, S_tot=3, scenes_steps=[[0, 1, 2], [0, 1]], S=32,
               seed=11, oversize_row=1)
    # G1b: last layer weight-normed too (norm_layers covers it) + use_tanh (tanh o tanh), as shipped small nets
    wn4b = dict(wn4, norm_layers=[0, 1, 2, 3, 4, 5, 6, 7], use_tanh=True, latent_in=[1], dims=[32] * 4)
    case_train(Decoder, "g1b_lastnorm_tanh", L=2, net_specs=wn4b, S_tot=2, scenes_steps=[[0, 1], [1, 0]], S=40, seed=12)
    # G1c: no weight norm at all, no skip, no code bound, no regulariser, grad clipping on
    plain = dict(dims=[48] * 3, dropout=None, dropout_prob=0.0, norm_layers=(), latent_in=(), xyz_in_all=False,
                 use_tanh=False, latent_dropout=False, weight_norm=False, geom_dimension=3)
    case_train(Decoder, "g1c_plain_clip", L=5, net_specs=plain, S_tot=2, scenes_steps=[[0, 1]], S=48, seed=13,
               code_bound=None, code_reg=False, grad_clip=0.05)
    # G2: the 8x512 benchmark architecture, 256-point slice
    big = dict(dims=[512] * 8, dropout=[0, 1, 2, 3, 4, 5, 6, 7], dropout_prob=0.0, norm_layers=[0, 1, 2, 3, 4, 5, 6, 7],
               latent_in=[4], xyz_in_all=False, use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)
    case_train(Decoder, "g2_8x512_slice", L=256, net_specs=big, S_tot=4, scenes_steps=[[0, 1, 2, 3]], S=64, seed=21,
               store="slice", oversize_row=2)
    # G3: dropout-injected training mode (hash masks), small + 8x512
    wn4d = dict(wn4, dropout_prob=0.2)
    case_train(Decoder, "g3a_dropout_tiny", L=4, net_specs=wn4d, S_tot=3, scenes_steps=[[0, 1, 2], [2, 0, 1]], S=32, seed=31,
               dropout_train=True, drop_seed=1234)
    bigd = dict(big, dropout_prob=0.2)
    case_train(Decoder, "g3b_dropout_8x512", L=256, net_specs=bigd, S_tot=4, scenes_steps=[[0, 1, 2, 3]], S=64, seed=32,
               dropout_train=True, drop_seed=99, store="slice")
    # G4: batch_split=2 (chunk boundary inside a scene: 3 scenes x 32 pts -> chunks of 48)
    case_train(Decoder, "g4_batch_split2", L=4, net_specs=wn4d, S_tot=3, scenes_steps=[[0, 1, 2]], S=32, seed=41,
               batch_split=2, dropout_train=True, drop_seed=7)
    # G11: the Decoder variants no shipped spec uses (deep_sdf_decoder.py:79-82, 90-91, 60-65/97-103)
    case_train(Decoder, "g11a_xyz_in_all", L=5, net_specs=dict(wn4d, xyz_in_all=True, dims=[48] * 4), S_tot=3,
               scenes_steps=[[0, 1, 2], [1, 2, 0]], S=32, seed=111, dropout_train=True, drop_seed=21)
    case_train(Decoder, "g11b_latent_dropout", L=6, net_specs=dict(wn4d, latent_dropout=True, dims=[40] * 4), S_tot=3,
               scenes_steps=[[0, 1, 2], [2, 1, 0]], S=32, seed=112, dropout_train=True, drop_seed=22, batch_split=2)
    case_train(Decoder, "g11c_layer_norm", L=4, net_specs=dict(wn4d, weight_norm=False, norm_layers=[0, 1, 2, 3, 4], dims=[56] * 4),
               S_tot=3, scenes_steps=[[0, 1, 2], [0, 2, 1]], S=32, seed=113, dropout_train=True, drop_seed=23)
    case_lr("g5_lr_schedules")
    case_real_weights("g6_real_weights")
    case_latent_only(Decoder, "g7_latent_only", L=8, net_specs=dict(wn4, latent_in=[2]), N=96, iters=5, seed=71)
    case_forward_eval(Decoder, "g8_eval_8x512", L=256, net_specs=bigd, N=128, seed=81)
    case_checkpoint_layout(Decoder, "g9_checkpoint_layout")
    case_forward_eval(Decoder, "g8_eval_6x128", L=1, net_specs=dict(
        dims=[128] * 6, dropout=list(range(8)), dropout_prob=0.2, norm_layers=list(range(8)), latent_in=[2],
        xyz_in_all=False, use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3), N=100, seed=82)
    case_reference_run("g10_reference_run")          # (from here on the reference's deep_sdf package is in sys.modules)
    case_sample_counts("g12_sample_counts")


if __name__ == "__main__":
    if "--layout-only" in sys.argv:
        case_checkpoint_layout(ref_decoder_cls(), "g9_checkpoint_layout")
    elif "--variants-only" in sys.argv:
        torch.set_num_threads(4)
        D = ref_decoder_cls()
        wn4d = dict(dims=[64] * 4, dropout=[0, 1, 2, 3], dropout_prob=0.2, norm_layers=[0, 1, 2, 3], latent_in=[2],
                    xyz_in_all=False, use_tanh=False, latent_dropout=False, weight_norm=True, geom_dimension=3)
        case_train(D, "g11a_xyz_in_all", L=5, net_specs=dict(wn4d, xyz_in_all=True, dims=[48] * 4), S_tot=3,
                   scenes_steps=[[0, 1, 2], [1, 2, 0]], S=32, seed=111, dropout_train=True, drop_seed=21)
        case_train(D, "g11b_latent_dropout", L=6, net_specs=dict(wn4d, latent_dropout=True, dims=[40] * 4), S_tot=3,
                   scenes_steps=[[0, 1, 2], [2, 1, 0]], S=32, seed=112, dropout_train=True, drop_seed=22, batch_split=2)
        case_train(D, "g11c_layer_norm", L=4, net_specs=dict(wn4d, weight_norm=False, norm_layers=[0, 1, 2, 3, 4], dims=[56] * 4),
                   S_tot=3, scenes_steps=[[0, 1, 2], [0, 2, 1]], S=32, seed=113, dropout_train=True, drop_seed=23)
    elif "--reference-trainer-only" in sys.argv:      # the two cases that import the reference's train_deep_sdf.py
        torch.set_num_threads(4)                      # as main(): the run is bit-reproducible for a fixed thread count
        case_lr("g5_lr_schedules")
        case_reference_run("g10_reference_run")
    elif "--sample-counts-only" in sys.argv:
        case_sample_counts("g12_sample_counts")
    else:
        main()
