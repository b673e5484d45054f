from deepsdf_amd.workspace import *  # noqa: F401,F403
