"""Explicit name for the HIP decoder."""
from deepsdf_amd.decoder import Decoder  # noqa: F401
