"""fdapde-core_amd: MI355X-native FEM assemble-and-solve path for fdaPDE-core.

Contents: csrc/ (HIP kernels + the extern "C" shim of include/fdapde_hip.h), capi.py (ctypes plumbing used by
tests/ and bench.py), meshgen.py (seeded synthetic workloads of BASELINE.md), dist.py (element-partitioned
multi-GPU driver).  The directory name is not a Python identifier; load it with `load_package()` from
__graft_entry__.py or tests/conftest.py, which registers it as module `fdapde_core_amd`.
"""
from . import capi  # noqa: F401  (fails loudly if the HIP library has not been built)
