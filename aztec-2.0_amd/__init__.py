"""bn254-mi355x: MI355X-native Pippenger MSM (BN254 G1) and radix-2 NTT family (BN254 Fr).

The product is the C-ABI shared library ``csrc/libbbg.so`` (HIP kernels for gfx950 + ``include/bbg.h``).  This
package is the thin ctypes binding used by tests and bench.py, plus the builder.  The directory name
``aztec-2.0_amd`` is not a Python identifier; ``__graft_entry__.load_package()`` imports it as ``aztec_amd``.

There is deliberately NO CPU fallback here: if the library or a gfx950 device is missing, every entry point raises.
Nothing in this package imports or calls anything under ``oracle/`` (the checker).
"""
from .binding import Bbg, BbgError, LIB_PATH, build_library, load_library  # noqa: F401
from .inputs import fr_reduce_once, splitmix64_limbs, synthetic_scalars, synthetic_scalars_strided  # noqa: F401
