"""Import-name shim so that the reference's entry points keep working unchanged:
``__import__("deep_sdf.networks." + specs["NetworkArch"], fromlist=["Decoder"])`` (train_deep_sdf.py:275,
deep_sdf/workspace.py:56-58) and ``import deep_sdf.workspace as ws`` resolve to the MI355X implementation."""
from deepsdf_amd.data import *  # noqa: F401,F403
from deepsdf_amd.utils import *  # noqa: F401,F403
from deepsdf_amd.workspace import *  # noqa: F401,F403
from . import workspace, data, utils  # noqa: F401,E402
