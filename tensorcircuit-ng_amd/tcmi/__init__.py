"""tcmi -- an MI355X-native executor for the tensorcircuit-ng state-vector / expectation hot path.

Drop-in surface (reference ``tensorcircuit/__init__.py``): ``set_backend`` / ``set_dtype`` /
``set_contractor``, ``Circuit``, ``gates``, ``backend``.  Compute happens in hand-written HIP kernels
behind the C ABI of ``include/tcmi.h``; there is no CPU fallback.
"""

__version__ = "0.1.0"

from . import cons
from .cons import (  # noqa: F401
    set_backend, set_dtype, set_contractor, set_function_backend, set_function_dtype,
    set_function_contractor, runtime_backend, runtime_dtype, runtime_contractor, get_dtype,
)

backend = None
dtypestr = cons.dtypestr
rdtypestr = cons.rdtypestr
idtypestr = cons.idtypestr
npdtype = cons.npdtype
contractor = None

from . import gates  # noqa: E402,F401
from .gates import array_to_tensor, num_to_tensor  # noqa: E402,F401
from . import plan  # noqa: E402,F401
from .circuit import Circuit  # noqa: E402,F401
from .mpscircuit import MPSCircuit  # noqa: E402,F401
from . import linalg  # noqa: E402,F401
from . import quantum, templates  # noqa: E402,F401
from . import interfaces  # noqa: E402,F401
from . import channels  # noqa: E402,F401
from .densitymatrix import DMCircuit, DMCircuit2  # noqa: E402,F401
from . import backends  # noqa: E402,F401
from . import tn, experimental, distributed, simplify  # noqa: E402,F401
from .backends import get_backend  # noqa: E402,F401

set_backend("hip")
set_contractor("greedy")

# ``tc.expectation(*ops, ket=, bra=, ...)`` (reference circuit.py:920-1065).  Bound last: the package attribute takes
# precedence over the internal submodule of the same name, which stays importable as ``tcmi.expectation`` through
# ``from .expectation import ...`` (sys.modules).
from . import expectation as _expectation_module  # noqa: E402,F401  (loaded now: a later first import would rebind the name)
from .circuit import expectation  # noqa: E402,F401
