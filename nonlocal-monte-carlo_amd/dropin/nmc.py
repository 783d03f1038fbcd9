"""Drop-in module: put this directory on sys.path and `from nmc import NMC` exactly like the reference's examples
(e.g. NMC/examples/general_example.py:5-6 do sys.path.append('../'); from nmc import NMC)."""
from _load import load as _load

NMC = _load().NMC
