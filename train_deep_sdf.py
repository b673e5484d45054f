#!/usr/bin/env python3
"""Train a DeepSDF autodecoder on MI355X -- drop-in for the reference's train_deep_sdf.py CLI (:584-622):

    python train_deep_sdf.py -e <experiment_dir> [-c latest|<epoch>] [--batch_split K] [--debug|-q] [--log FILE]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 train_deep_sdf.py -e <experiment_dir>
"""
import argparse

import deep_sdf
from deepsdf_amd.train import main_function

if __name__ == "__main__":
    arg_parser = argparse.ArgumentParser(description="Train a DeepSDF autodecoder")
    arg_parser.add_argument("--experiment", "-e", dest="experiment_directory", required=True,
                            help="The experiment directory. This directory should include experiment specifications in "
                                 "'specs.json', and logging will be done in this directory as well.")
    arg_parser.add_argument("--continue", "-c", dest="continue_from",
                            help="A snapshot to continue from. This can be 'latest' to continue from the latest running "
                                 "snapshot, or an integer corresponding to an epochal snapshot.")
    arg_parser.add_argument("--batch_split", dest="batch_split", default=1,
                            help="This splits the batch into separate subbatches which are processed separately, with "
                                 "gradients accumulated across all subbatches.")
    deep_sdf.add_common_args(arg_parser)
    args = arg_parser.parse_args()
    deep_sdf.configure_logging(args)
    main_function(args.experiment_directory, args.continue_from, int(args.batch_split))
