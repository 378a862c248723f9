"""auncel_amd: MI355X (gfx950) implementation of Auncel's IVF-Flat search hot path.

The product is the C-ABI library built from auncel_amd/csrc (include/auncel_amd.h); this
package holds its build recipe, a ctypes binding used by the tests and the benchmark, and the
synthetic dataset generators.  There is no CPU compute path in here."""

