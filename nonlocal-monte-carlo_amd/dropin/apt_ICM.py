"""Drop-in module: put this directory on sys.path and `from apt_ICM import APT_ICM` exactly like the reference's examples
(e.g. NMC/examples/general_example.py:5-6 do sys.path.append('../'); from nmc import NMC)."""
from _load import load as _load

APT_ICM = _load().APT_ICM
