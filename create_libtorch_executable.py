#!/usr/bin/env python3
"""TorchScript export of a trained experiment for libtorch consumers: the workflow of the reference's
create_libtorch_executable.py:4-24 (same CLI, same output file ``<experiment>/cpp_model.pt``).

The reference traces its Decoder on the CPU; the HIP Decoder has no CPU compute path, so the trace is taken from the
stock-torch eval-mode twin built from its parameters (deepsdf_amd/export.py).

    python create_libtorch_executable.py -e <experiment_dir> -c latest
"""
import argparse
import os

import torch

import deep_sdf.workspace as ws


def main(experiment_directory, checkpoint):
    decoder = ws.load_trained_model(experiment_directory, checkpoint)
    latent = ws.load_latent_vectors(experiment_directory, checkpoint)
    latent = latent.cpu() if torch.is_tensor(latent) else torch.stack([t.reshape(-1) for t in latent]).cpu()
    example_input = torch.cat([latent[0], torch.zeros(decoder.geom_dimension)]).unsqueeze(0)
    out = os.path.join(experiment_directory, "cpp_model.pt")
    sm = decoder.export_torchscript(example_input, out)
    print("Example input: ", example_input)
    print("Example output:", sm(example_input))
    print("wrote", out)
    return out


if __name__ == "__main__":
    parser = argparse.ArgumentParser()
    parser.add_argument("--experiment_directory", "-e", type=str, required=True)
    parser.add_argument("--checkpoint", "-c", type=str, default="latest")
    args = parser.parse_args()
    main(args.experiment_directory, args.checkpoint)
