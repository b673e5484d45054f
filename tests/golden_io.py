"""Loader for the committed golden vectors (tests/golden/*.npz, written by tests/golden/make_golden.py)."""
import json
import os

import numpy as np
import torch

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


class Golden:
    def __init__(self, name):
        self.npz = np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)
        self.meta = json.loads(bytes(self.npz["meta"]).decode())

    def group(self, prefix):
        """dict of torch tensors for keys '<prefix>/<name>' (names may contain dots, never slashes)."""
        pre = prefix + "/"
        out = {}
        for k in self.npz.files:
            if k.startswith(pre) and "/" not in k[len(pre):]:
                out[k[len(pre):]] = torch.from_numpy(np.asarray(self.npz[k]))
        return out

    def get(self, key):
        return torch.from_numpy(np.asarray(self.npz[key]))

    def has(self, key):
        return key in self.npz.files


def rel_err(a, b):
    """norm-wise relative error ||a-b|| / max(||b||, tiny) in float64."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).norm() / max(float(b.norm()), 1e-30))


def worst_elem(a, b):
    """element-wise worst case max|a-b| / max|b| in float64: what a norm-wise bound cannot see (a handful of corrupted rows or
    entries among millions -- the 8-wave bf16 kernel's register-reuse race of round 2 passed every norm-wise test)."""
    a, b = a.double().reshape(-1), b.double().reshape(-1)
    return float((a - b).abs().max() / max(float(b.abs().max()), 1e-30))


def g12_scene_file(path, c):
    """The synthetic .npz of one G12 case (tests/golden/make_golden.py:case_sample_counts): a row carries its own identity,
    x = index within its sign, z = sign, sdf = +/-(index + 1); the first nan_* rows of a sign are NaN samples."""
    def rows(n, n_nan, sign):
        i = np.arange(n, dtype=np.float64)
        r = np.stack([i, np.zeros(n), np.full(n, float(sign)), sign * (i + 1)], 1)
        r[:n_nan, 3] = np.nan
        return r.astype(c["dtype"])
    np.savez(path, pos=rows(c["n_pos"], c["nan_pos"], 1), neg=rows(c["n_neg"], c["nan_neg"], -1))


def sphere_npz(path, k, n=6000):
    """Synthetic scene k (sphere SDF) in the on-disk format of sdf_sampler/sdf_sampler.py:146: pos / neg [*, 4]."""
    g = np.random.default_rng(1234 + k)
    c = g.uniform(-0.3, 0.3, 3) if k else np.zeros(3)
    r = g.uniform(0.3, 0.6) if k else 0.5
    x = np.concatenate([g.uniform(-1, 1, (n // 2, 3)),
                        c + r * (lambda d: d / np.linalg.norm(d, axis=1, keepdims=True))(g.normal(size=(n - n // 2, 3)))
                        + g.normal(0, 0.05, (n - n // 2, 3))]).astype(np.float32)
    sdf = (np.linalg.norm(x - c, axis=1) - r).astype(np.float32)
    rows = np.concatenate([x, sdf[:, None]], 1)
    np.savez(path, pos=rows[sdf >= 0], neg=rows[sdf < 0])


def write_reference_experiment(root, name="g10_reference_run", which=("latest.pth",)):
    """Rebuild, from the arrays of golden G10, the experiment directory the REFERENCE trainer wrote (train_deep_sdf.py:96-143,
    179-199): ModelParameters/ OptimizerParameters/ LatentCodes/ <which> + Logs.pth + specs.json, plus the synthetic data
    set it trained on.  Returns (experiment_dir, data_dir, golden)."""
    g = Golden(name)
    m = g.meta
    data = os.path.join(root, "data")
    os.makedirs(os.path.join(data, "SdfSamples", "synth", "spheres"), exist_ok=True)
    for k, nme in enumerate(m["scene_names"]):
        sphere_npz(os.path.join(data, "SdfSamples", "synth", "spheres", nme + ".npz"), k)
    split = os.path.join(root, "split.json")
    json.dump({"synth": {"spheres": m["scene_names"]}}, open(split, "w"))
    exp = os.path.join(root, "exp")
    os.makedirs(exp, exist_ok=True)
    specs = dict(m["specs"], DataSource=data, TrainSplit=split, TestSplit=split, ReconstructionSplit=split)
    json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
    ep = m["epochs"]
    model = {k: g.get("model/" + k) for k in m["model_keys"]}
    state = {i: {"step": g.get(f"opt{i}/step"), "exp_avg": g.get(f"opt{i}/exp_avg"), "exp_avg_sq": g.get(f"opt{i}/exp_avg_sq")}
             for i in m["opt_state_ids"]}
    osd = {"state": state, "param_groups": [dict(pg, betas=tuple(pg["betas"])) for pg in m["param_groups"]]}
    for sub, payload in (("ModelParameters", {"epoch": ep["model"], "model_state_dict": model}),
                         ("OptimizerParameters", {"epoch": ep["optimizer"], "optimizer_state_dict": osd}),
                         ("LatentCodes", {"epoch": ep["latent"], "latent_codes": {"weight": g.get("latent/weight")}})):
        os.makedirs(os.path.join(exp, sub), exist_ok=True)
        for f in which:
            torch.save(payload, os.path.join(exp, sub, f))
    pm = {k: [float(v) for v in g.get("logs_pm/" + k)] for k in m["param_magnitude_keys"]}
    torch.save({"epoch": ep["logs"], "loss": [float(v) for v in g.get("logs/loss")],
                "learning_rate": [[float(a) for a in row] for row in g.get("logs/learning_rate")],
                "timing": [float(v) for v in g.get("logs/timing")],
                "latent_magnitude": [v.float() for v in g.get("logs/latent_magnitude")], "param_magnitude": pm},
               os.path.join(exp, "Logs.pth"))
    return exp, data, g
