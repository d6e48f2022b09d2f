"""MI355X-native batched differentiable FK + planning-objective engine.

Drop-in for the hot path of anindex/torch_robotics (FK -> costs -> gradient over
(batch x horizon x DOF) rollouts); see DESIGN.md.  All compute runs in the HIP
library `csrc/libtrk.so` through the C ABI in `include/trk.h`; there is no CPU
compute path in this package.
"""
from .kinmodel import KinModel  # noqa: F401

__version__ = "0.1.0"
