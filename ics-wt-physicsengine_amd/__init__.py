"""MI355X-native multi-zone CSTR physics step (hot path of ICS-WT-PhysicsEngine).

Import with ``importlib.import_module("ics-wt-physicsengine_amd")`` (the
directory name is not a Python identifier) or put this directory's parent on
``sys.path`` and use ``tests/conftest.py``'s ``wtamd`` fixture.
"""
import os as _os

# The step is scheduled as several reactor ranges on their own HIP streams; ROCm maps
# streams onto 4 hardware queues by default, which can put two ranges on one queue.
# Must be set before the HIP runtime initialises (harmless if it already has).
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")

from . import core
from .core import *  # noqa: F401,F403
from .core import __all__ as _core_all

__all__ = ["core"] + list(_core_all)
__version__ = "0.1.0"
