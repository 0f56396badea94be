"""plinking_duck_amd -- MI355X (gfx950) path for PlinkingDuck's .pgen functions.

The product is the C-ABI library ``libpgenhip.so`` (``include/pgenhip.h``) built
from ``plinking_duck_amd/csrc``; this package is only its ctypes binding plus the
host-side mirror of the reference's table functions.  Nothing here computes on
the CPU in place of the HIP path: if the library is missing the import of
:mod:`plinking_duck_amd.lib` raises.
"""

__all__ = ["lib"]
