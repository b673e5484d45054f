"""CLI logging flags and frozen-decoder inference helper: API of the reference's deep_sdf/utils.py
(add_common_args :9-30, configure_logging :33-49, decode_sdf :54-65)."""
import logging

import torch


def add_common_args(arg_parser):
    arg_parser.add_argument("--debug", dest="debug", default=False, action="store_true",
                            help="If set, debugging messages will be printed")
    arg_parser.add_argument("--quiet", "-q", dest="quiet", default=False, action="store_true",
                            help="If set, only warnings will be printed")
    arg_parser.add_argument("--log", dest="logfile", default=None,
                            help="If set, the log will be saved using the specified filename.")


def configure_logging(args):
    logger = logging.getLogger("deep_sdf.utils")
    logger.setLevel(logging.DEBUG if args.debug else (logging.WARNING if args.quiet else logging.INFO))
    fmt = logging.Formatter("%(asctime)s DeepSdf - %(levelname)s - %(message)s", datefmt="%H:%M:%S")
    handlers = [logging.StreamHandler()]
    if args.logfile is not None:
        handlers.append(logging.FileHandler(args.logfile))
    for h in handlers:
        h.setFormatter(fmt)
        logger.addHandler(h)


def decode_sdf(decoder, latent_vector, queries):
    """sdf = decoder([latent (broadcast) || queries]); latent_vector None => queries already carry the code."""
    if latent_vector is None:
        inputs = queries
    else:
        inputs = torch.cat([latent_vector.expand(queries.shape[0], -1), queries], 1)
    return decoder(inputs)
