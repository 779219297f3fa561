"""pointcloudlib_amd -- MI355X-native (gfx950) implementation of the point-cloud hot path of
Jittor/PointCloudLib: farthest-point sampling, ball query / k-NN, grouped gather, per-group pointwise
MLP + max, behind the reference's operator signatures (``misc/ops.py``, ``misc/layers.py``) so that the
``networks/cls`` and ``networks/seg`` counterparts read like the originals.

Host code is Python on PyTorch-ROCm (device memory, streams, autograd, torch.distributed/RCCL); all
hot-path arithmetic lives in ``libpcl_hip.so`` (hand-written HIP for gfx950) behind the C ABI declared
in ``include/pcl_hip.h``.  There is no CPU fallback inside this package.
"""
from . import _lib  # noqa: F401
from ._lib import PclError, build  # noqa: F401

__version__ = "0.1.0"
