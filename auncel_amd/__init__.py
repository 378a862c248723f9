"""auncel_amd: MI355X (gfx950) implementation of Auncel's IVF-Flat search hot path.

The product is the C-ABI library built from auncel_amd/csrc (include/auncel_amd.h); this
package holds its build recipe, a ctypes binding used by the tests and the benchmark, and the
synthetic dataset generators.  There is no CPU compute path in here."""

import os as _os

# The engine lays its streams out for 8 hardware queues per priority class (auncel_amd/csrc/ivf_engine.hip: HwQueues; ROCm's default is
# 4) and asks for them when its library is loaded -- which is too late if something (torch) has started the HIP runtime before.  Said
# here as well, at package import, so that importing auncel_amd before the first GPU call is enough.
_os.environ.setdefault("GPU_MAX_HW_QUEUES", "8")
