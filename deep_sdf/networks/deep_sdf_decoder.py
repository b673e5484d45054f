"""``"NetworkArch": "deep_sdf_decoder"`` (every shipped specs.json) resolves to the HIP decoder."""
from deepsdf_amd.decoder import Decoder  # noqa: F401
