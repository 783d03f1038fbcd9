"""Loads the parent package (its directory name has a hyphen) under the importable alias `nlmc_amd`."""
import importlib.util
import os
import sys


def load():
    if "nlmc_amd" in sys.modules:
        return sys.modules["nlmc_amd"]
    pkg_dir = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("nlmc_amd", os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules["nlmc_amd"] = mod
    spec.loader.exec_module(mod)
    return mod
