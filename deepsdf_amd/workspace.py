"""Experiment-directory layout and loaders: the API of the reference's ``deep_sdf/workspace.py`` (constants :8-22,
loaders :25-115, path helpers :118-209, load_trained_model :212-242), so every consumer of an experiment directory
(meshing, analysis, optimisation scripts) keeps working against checkpoints written by this package and vice versa.
Error messages and exception types follow the reference (bare ``Exception`` for missing files)."""
import json
import os

import torch

# directory / file names inside an experiment directory (workspace.py:8-22)
screenshots_subdir = "Screenshots"
model_params_subdir = "ModelParameters"
optimizer_params_subdir = "OptimizerParameters"
latent_codes_subdir = "LatentCodes"
logs_filename = "Logs.pth"
reconstructions_subdir = "Reconstructions"
reconstruction_meshes_subdir = "Meshes"
reconstruction_codes_subdir = "Codes"
specifications_filename = "specs.json"
data_source_map_filename = ".datasources.json"
evaluation_subdir = "Evaluation"
sdf_samples_subdir = "SdfSamples"
surface_samples_subdir = "SurfaceSamples"
normalization_param_subdir = "NormalizationParameters"
training_meshes_subdir = "TrainingMeshes"


def _device():
    return torch.device("cuda") if torch.cuda.is_available() else torch.device("cpu")


def _subdir(experiment_dir, name, create):
    d = os.path.join(experiment_dir, name)
    if create and not os.path.isdir(d):
        os.makedirs(d)
    return d


def load_experiment_specifications(experiment_directory):
    filename = os.path.join(experiment_directory, specifications_filename)
    if not os.path.isfile(filename):
        raise Exception("The experiment directory ({}) does not include specifications file ".format(experiment_directory)
                        + '"specs.json"')
    with open(filename) as f:
        return json.load(f)


def load_model_parameters(experiment_directory, checkpoint, decoder):
    filename = os.path.join(experiment_directory, model_params_subdir, checkpoint + ".pth")
    if not os.path.isfile(filename):
        raise Exception('model state dict "{}" does not exist'.format(filename))
    data = torch.load(filename, map_location=_device(), weights_only=True)
    decoder.load_state_dict(data["model_state_dict"])
    return data["epoch"]


def build_decoder(experiment_directory, experiment_specs):
    arch = __import__("deep_sdf.networks." + experiment_specs["NetworkArch"], fromlist=["Decoder"])
    decoder = arch.Decoder(experiment_specs["CodeLength"], **experiment_specs["NetworkSpecs"])
    return decoder.cuda() if torch.cuda.is_available() else decoder


def load_decoder(experiment_directory, experiment_specs, checkpoint, data_parallel=True):
    decoder = build_decoder(experiment_directory, experiment_specs)
    if data_parallel:
        decoder = torch.nn.DataParallel(decoder)
    epoch = load_model_parameters(experiment_directory, checkpoint, decoder)
    return decoder, epoch


def load_latent_vectors(experiment_directory, checkpoint):
    filename = os.path.join(experiment_directory, latent_codes_subdir, checkpoint + ".pth")
    if not os.path.isfile(filename):
        raise Exception(f"The experiment directory ({experiment_directory}) does not include a latent code file"
                        + f" for checkpoint '{checkpoint}'")
    data = torch.load(filename, map_location="cpu", weights_only=True)
    codes = data["latent_codes"]
    if isinstance(codes, torch.Tensor):          # legacy upstream format: tensor [num, 1, L]
        return [codes[i].to(_device()) for i in range(codes.size()[0])]
    return codes["weight"].detach().clone()


def load_trained_model(experiment_directory: str, checkpoint: str):
    specs = load_experiment_specifications(experiment_directory)
    decoder, _ = load_decoder(experiment_directory, specs, checkpoint)
    return decoder.module.cuda() if torch.cuda.is_available() else decoder.module


def print_model_specifications(experiment_directory: str):
    specs = load_experiment_specifications(experiment_directory)
    print("Model Specifications:")
    for key in specs:
        print(f"  {key}: {specs[key]}")
    print("\n")


def get_data_source_map_filename(data_dir):
    return os.path.join(data_dir, data_source_map_filename)


def get_reconstructed_mesh_filename(experiment_dir, epoch, dataset, class_name, instance_name):
    return os.path.join(experiment_dir, reconstructions_subdir, str(epoch), reconstruction_meshes_subdir, dataset,
                        class_name, instance_name + ".ply")


def get_reconstructed_code_filename(experiment_dir, epoch, dataset, class_name, instance_name):
    return os.path.join(experiment_dir, reconstructions_subdir, str(epoch), reconstruction_codes_subdir, dataset,
                        class_name, instance_name + ".pth")


def get_evaluation_dir(experiment_dir, checkpoint, create_if_nonexistent=False):
    return _subdir(experiment_dir, os.path.join(evaluation_subdir, checkpoint), create_if_nonexistent)


def get_model_params_dir(experiment_dir, create_if_nonexistent=False):
    return _subdir(experiment_dir, model_params_subdir, create_if_nonexistent)


def get_screenshots_dir(experiment_dir, create_if_nonexistent=True):
    return _subdir(experiment_dir, screenshots_subdir, create_if_nonexistent)


def get_optimizer_params_dir(experiment_dir, create_if_nonexistent=False):
    return _subdir(experiment_dir, optimizer_params_subdir, create_if_nonexistent)


def get_latent_codes_dir(experiment_dir, create_if_nonexistent=False):
    return _subdir(experiment_dir, latent_codes_subdir, create_if_nonexistent)


def get_normalization_params_filename(data_dir, dataset_name, class_name, instance_name):
    return os.path.join(data_dir, normalization_param_subdir, dataset_name, class_name, instance_name + ".npz")
