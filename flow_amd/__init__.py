"""flow_amd -- MI355X-native MLS-MPM step for noahschwad/flow's src/mls-mpm simulator.

The package holds only what the hot path needs: the HIP kernels and their C ABI
(csrc/, include/flowmpm.h), the ctypes binding (_lib), and the host-side mirror of
the reference's simulator class (simulator, modes, structured_array, conf, particles, slab)
and of its caller (app).
"""
from . import _lib
from .conf import Conf, conf
from .app import App, run_headless
from .simulator import Handle, MlsMpmSimulator, SlabGroup, Vector3
from .structured_array import PARTICLE_STRUCT, StructuredArray

__all__ = ["_lib", "App", "run_headless", "Conf", "conf", "Handle", "MlsMpmSimulator", "SlabGroup", "Vector3", "PARTICLE_STRUCT",
           "StructuredArray"]
