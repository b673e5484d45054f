#!/usr/bin/env python3
"""Latent-only reconstruction CLI (BASELINE config 4): optimise one code per shape against a trained, FROZEN decoder.

The fork deleted upstream's reconstruct.py (README.md:139,185 still advertise it); this script restores the workflow on
top of the artefacts the fork keeps (deep_sdf/workspace.py:122-149): codes are written to
``<experiment>/Reconstructions/<epoch>/Codes/<dataset>/<class>/<instance>.pth`` (a [1, 1, L] tensor, as upstream wrote).
Meshes are NOT produced (marching cubes / FlexiCubes are out of scope).  All shapes of the split are
reconstructed TOGETHER in batches (deepsdf_amd/reconstruct.py), not one after the other.

    python reconstruct.py -e <experiment_dir> -c latest -d <data_dir> -s <split.json> [--iters 800] [--skip]
"""
import argparse
import json
import logging
import os

import torch

import deep_sdf
import deep_sdf.workspace as ws
from deepsdf_amd.data import DeviceSampleCache, get_instance_filenames
from deepsdf_amd.reconstruct import reconstruct

if __name__ == "__main__":
    ap = argparse.ArgumentParser(description="Use a trained DeepSDF decoder to reconstruct shapes given SDF samples.")
    ap.add_argument("--experiment", "-e", dest="experiment_directory", required=True)
    ap.add_argument("--checkpoint", "-c", dest="checkpoint", default="latest")
    ap.add_argument("--data", "-d", dest="data_source", required=True)
    ap.add_argument("--split", "-s", dest="split_filename", required=True)
    ap.add_argument("--iters", dest="iterations", default=800, type=int)
    ap.add_argument("--samples", dest="num_samples", default=8000, type=int, help="SDF samples per shape per iteration")
    ap.add_argument("--shapes_per_batch", default=64, type=int)
    ap.add_argument("--skip", dest="skip", action="store_true", help="skip shapes whose code file already exists")
    deep_sdf.add_common_args(ap)
    args = ap.parse_args()
    deep_sdf.configure_logging(args)
    if not torch.cuda.is_available():
        raise RuntimeError("reconstruct.py (deepsdf_amd) needs an AMD GPU: the HIP path has no CPU fallback")

    specs = ws.load_experiment_specifications(args.experiment_directory)
    decoder = ws.load_trained_model(args.experiment_directory, args.checkpoint)
    decoder.eval()
    eng = decoder.engine()
    saved_epoch = torch.load(os.path.join(args.experiment_directory, ws.model_params_subdir, args.checkpoint + ".pth"),
                             map_location="cpu", weights_only=True)["epoch"]
    with open(args.split_filename) as f:
        split = json.load(f)
    npz = get_instance_filenames(args.data_source, split)
    todo = []
    for f in npz:
        ds, cls, inst = f[:-4].split(os.sep)
        out = ws.get_reconstructed_code_filename(args.experiment_directory, saved_epoch, ds, cls, inst)
        if not (args.skip and os.path.isfile(out)):
            todo.append((f, out))
    S = 64 * max(1, args.num_samples // 64)       # whole 64-row workgroups per shape -> segment-sum path
    clamp = specs["ClampingDistance"]
    gen = torch.Generator(device=eng.device)
    gen.manual_seed(0)
    for b0 in range(0, len(todo), args.shapes_per_batch):
        chunk = todo[b0:b0 + args.shapes_per_batch]
        cache = DeviceSampleCache.from_files(args.data_source, [c[0] for c in chunk], decoder.geom_dimension, eng.device)
        ids = torch.arange(len(chunk))

        # upstream drew a new random subsample every iteration: `resample` does it on the device, inside the captured graph
        shape = torch.empty(len(chunk), S, decoder.geom_dimension, device=eng.device)
        z, loss = reconstruct(eng, shape, torch.empty(len(chunk), S, device=eng.device), num_iterations=args.iterations, clamp_dist=clamp,
                              lr=5e-3, l2reg=1e-4, init_std=0.01, resample=(cache, ids, gen))
        logging.info("batch %d: %d shapes, last loss %.5f", b0 // args.shapes_per_batch, len(chunk), float(loss))
        for (f, out), code in zip(chunk, z.cpu()):
            os.makedirs(os.path.dirname(out), exist_ok=True)
            torch.save(code.view(1, 1, -1), out)
