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
    """sdf = decoder([latent (broadcast) || queries]); latent_vector None => queries already carry the code.

    Inference with ONE code for all query points (every caller of the reference's decode_sdf, deep_sdf/utils.py:54-65) never
    builds the [n, L+G] input: ``Engine.decode_latent`` hoists the code's products out of the per-point work.  Anything that
    needs autograd, training-mode dropout, a DataParallel wrapper's module or several codes takes the module path."""
    if latent_vector is None:
        return decoder(queries)
    dec = decoder.module if isinstance(decoder, torch.nn.DataParallel) else decoder
    grad = torch.is_grad_enabled() and (latent_vector.requires_grad or queries.requires_grad
                                        or any(p.requires_grad for p in dec.parameters()))
    spec = getattr(dec, "spec", None)
    if spec is not None and not grad and not dec.training and latent_vector.numel() == spec.latent_size and queries.is_cuda:
        eng = dec._engine_for(queries.device)
        if eng.decode_latent_supported():      # the library's own condition (variants / wide nets take the module path below)
            eng.weights_dirty = True           # parameters may have been changed by any optimizer since the last call
            return eng.decode_latent(latent_vector, queries)
    inputs = torch.cat([latent_vector.expand(queries.shape[0], -1), queries], 1)
    return decoder(inputs)
