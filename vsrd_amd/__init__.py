"""vsrd_amd: MI355X-native implementation of VSRD's instance-aware volumetric silhouette renderer.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd glue); the renderer itself is
hand-written HIP for gfx950 behind the C ABI of ``include/vsrd_hip.h`` (``vsrd_amd/lib/libvsrd_hip.so``).
The sub-packages mirror the part of the reference's ``vsrd`` package that ``scripts/main.py`` touches on
its hot path: ``rendering``, ``operations``, plus ``fields`` (the closures of ``main.py:433-523`` as
importable objects) and ``models`` (the optimised parameters).
"""
from . import _lib, fields, rendering, operations, models, utils, loss_library

__all__ = ["fields", "rendering", "operations", "models", "install_as_vsrd"]


def install_as_vsrd():
    """Register this package under the name ``vsrd`` so ``import vsrd`` in scripts/main.py resolves here
    (only the sub-modules on the hot path exist)."""
    import sys
    sys.modules.setdefault("vsrd", sys.modules[__name__])
    sys.modules.setdefault("vsrd.rendering", rendering)
    sys.modules.setdefault("vsrd.rendering.sdfs", rendering.sdfs)
    sys.modules.setdefault("vsrd.operations", operations)
    sys.modules.setdefault("vsrd.models", models)
    sys.modules.setdefault("vsrd.utils", utils)
    sys.modules.setdefault("vsrd.losses", loss_library)
