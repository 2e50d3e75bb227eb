"""qpdo_amd -- MI355X-native primal-dual Newton proximal QP engine behind the qpdo.h C API.

The package holds the HIP kernels + C host driver (csrc/), the in-tree build helper, the
Python front end mirroring the reference's MATLAB class (solver.py) and the synthetic
problem generator used by tests and bench.py (problems.py).
"""
__all__ = ["problems"]
