"""End-to-end epoch time of the drop-in trainer on a config-2-shaped synthetic experiment (256 scenes x 20k samples on disk,
ScenesPerBatch 64 x SamplesPerScene 256 = 16384 pts/step, 4 steps per epoch).  Usage: python tools/trainer_bench.py [epochs]
TB_NET=4x64 | 4x32 | 6x128 | 8x512_L16: one of the reference's SHIPPED specs at its batch shape instead (ScenesPerBatch 10 x SamplesPerScene
16000 = 160000 pts/step; TB_SCENES scenes, default 40 -> 4 steps per epoch)."""
import json, os, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch


def sphere_scene(k, n_points):
    """Synthetic sphere-SDF scene in the on-disk sample format (pos / neg arrays [*, 4] fp32)."""
    rng = np.random.default_rng(1234 + k)
    c, r = rng.uniform(-0.3, 0.3, 3), rng.uniform(0.3, 0.6)
    h = n_points // 2
    d = rng.normal(size=(n_points - h, 3))
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    pts = np.concatenate([rng.uniform(-1, 1, (h, 3)), c + r * d + rng.normal(0, 0.05, (n_points - h, 3))], 0)
    s = np.concatenate([pts, (np.linalg.norm(pts - c, axis=1) - r)[:, None]], 1).astype(np.float32)
    return s[s[:, 3] >= 0], s[s[:, 3] < 0]


from deepsdf_amd import train
epochs = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
root = tempfile.mkdtemp()
d = os.path.join(root, "data", "SdfSamples", "synth", "spheres"); os.makedirs(d)
names = []
TB_NET = os.environ.get("TB_NET", "")
NSC = int(os.environ.get("TB_SCENES", "40" if TB_NET else "256"))
for k in range(NSC):
    pos, neg = sphere_scene(k, 20000)
    np.savez(os.path.join(d, f"s{k}.npz"), pos=pos, neg=neg); names.append(f"s{k}")
json.dump({"synth": {"spheres": names}}, open(os.path.join(root, "split.json"), "w"))
exp = os.path.join(root, "exp"); os.makedirs(exp)
specs = {"Description": "config 2 shape", "DataSource": os.path.join(root, "data"), "NetworkArch": "deep_sdf_decoder",
         "TrainSplit": os.path.join(root, "split.json"), "TestSplit": "", "ReconstructionSplit": "",
         "NetworkSpecs": {"dims": [512] * 8, "dropout": list(range(8)), "dropout_prob": 0.2, "norm_layers": list(range(8)),
                          "latent_in": [4], "xyz_in_all": False, "use_tanh": False, "latent_dropout": False, "weight_norm": True,
                          "geom_dimension": 3},
         "CodeLength": 256, "NumEpochs": epochs, "SnapshotFrequency": 10 ** 6, "AdditionalSnapshots": [],
         "LearningRateSchedule": [{"Type": "Step", "Initial": 0.0005, "Interval": 500, "Factor": 0.5},
                                  {"Type": "Step", "Initial": 0.001, "Interval": 500, "Factor": 0.5}],
         "SamplesPerScene": 256, "ScenesPerBatch": 64, "ClampingDistance": 0.1, "CodeRegularization": True,
         "CodeRegularizationLambda": 1e-4, "CodeBound": 1.0, "LogFrequency": 10 ** 6}
SPB, SPS = 64, 256
if TB_NET:
    shipped = {"4x64": (dict(dims=[64] * 4, latent_in=[1], use_tanh=True), 2), "4x32": (dict(dims=[32] * 4, latent_in=[2], use_tanh=False), 2),
               "6x128": (dict(dims=[128] * 6, latent_in=[2], use_tanh=False), 1), "8x512_L16": (dict(dims=[512] * 8, latent_in=[4], use_tanh=False), 16)}[TB_NET]
    specs["NetworkSpecs"].update(shipped[0]); specs["CodeLength"] = shipped[1]
    SPB, SPS = 10, 16000
    specs["ScenesPerBatch"], specs["SamplesPerScene"] = SPB, SPS
json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
import logging; logging.disable(logging.INFO)
specs["NumEpochs"] = 5; json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
train.main_function(exp, None, 1); torch.cuda.synchronize()      # untimed: library load, caches, first-touch of the arenas
specs["NumEpochs"] = epochs; json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
t0 = time.time(); train.main_function(exp, None, 1); torch.cuda.synchronize(); t = time.time() - t0
print(f"trainer{' ' + TB_NET if TB_NET else ''}: {epochs} epochs ({NSC//SPB} steps of {SPB * SPS} pts each) in {t:.2f} s incl. setup")
specs["NumEpochs"] = 3 * epochs; json.dump(specs, open(os.path.join(exp, "specs.json"), "w"))
t0 = time.time(); train.main_function(exp, None, 1); torch.cuda.synchronize(); t3 = time.time() - t0
per = (t3 - t) / (2 * epochs)
spe = NSC // SPB
print(f"marginal: {per*1e3:.3f} ms/epoch = {per/spe*1e3:.3f} ms/step = {SPB*SPS*spe/per/1e6:.2f} M pts/s end-to-end (sampling + step + per-epoch logging)")
