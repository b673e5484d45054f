"""deepsdf_amd -- MI355X-native DeepSDF auto-decoder training step (HIP kernels behind a C ABI) with the
reference's specs.json / experiment-directory host API.  See DESIGN.md."""
__version__ = "0.1.0"

from . import dist as _dist  # noqa: F401,E402  -- FIRST: puts the HSA_*/NCCL_* defaults in the environment before any GPU call
