"""Top-level ``train_utils`` of the reference repository, served by the MI355X package: put this directory in front of the reference's on
``sys.path`` / PYTHONPATH and ``main.py``'s imports (``main.py:13-15``) resolve to the HIP-backed implementation."""
from scene_graph_commonsense_amd.train_utils import *          # noqa: F401,F403
from scene_graph_commonsense_amd import train_utils as _impl

globals().update({k: v for k, v in vars(_impl).items() if not k.startswith("__")})
