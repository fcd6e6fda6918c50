"""flow_amd -- MI355X-native MLS-MPM step for noahschwad/flow's src/mls-mpm simulator.

The package holds only what the hot path needs: the HIP kernels and their C ABI
(csrc/, include/flowmpm.h), the ctypes binding (_lib), and the host-side mirror of
the reference's simulator class (simulator, structured_array, conf, particles, slab).
"""
from . import _lib
from .conf import Conf, conf
from .simulator import Handle, MlsMpmSimulator, Vector3
from .structured_array import PARTICLE_STRUCT, StructuredArray

__all__ = ["_lib", "Conf", "conf", "Handle", "MlsMpmSimulator", "Vector3", "PARTICLE_STRUCT",
           "StructuredArray"]
