"""Drop-in module: put this directory on sys.path and `from apt_preprocessor import APT_preprocessor`
(NPT/examples/general_example.py)."""
from _load import load as _load

APT_preprocessor = _load().APT_preprocessor
