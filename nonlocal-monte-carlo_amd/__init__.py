"""nlmc_amd -- MI355X-native (gfx950) heat-bath sweep + replica-exchange engine behind the Python API of
usra-riacs/Nonlocal-Monte-Carlo (NMC(J,h).run, NPT(J,h).run, APT_ICM(J,h).run).

The directory name carries a hyphen (repo convention), so load it as `nlmc_amd` via `load()` in
`__graft_entry__.py` / tests/conftest.py, or put this directory on sys.path and import the drop-in modules
`nmc`, `npt`, `apt_ICM` exactly like the reference's examples do.
"""
from . import _abi  # noqa: F401
from ._abi import build_library  # noqa: F401
from .engine import Engine, Instance, device_count  # noqa: F401
from . import hostlogic  # noqa: F401
from . import lbp  # noqa: F401
from .nmc import NMC  # noqa: F401
from .npt import NPT  # noqa: F401
from .apt_ICM import APT_ICM  # noqa: F401
from . import distributed  # noqa: F401
from .apt_preprocessor import APT_preprocessor  # noqa: F401
from . import instances  # noqa: F401
