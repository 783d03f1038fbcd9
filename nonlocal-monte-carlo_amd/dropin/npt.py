"""Drop-in module: put this directory on sys.path and `from npt import NPT` exactly like the reference's examples
(e.g. NMC/examples/general_example.py:5-6 do sys.path.append('../'); from nmc import NMC)."""
from _load import load as _load

NPT = _load().NPT
