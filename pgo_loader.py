"""Imports the package directory `posegraph-ceres_amd/` (hyphenated, so not importable by name) as the
module `posegraph_ceres_amd`, plus its `datasets` submodule."""
import importlib.util
import os
import sys

_ROOT = os.path.dirname(os.path.abspath(__file__))
_PKG_DIR = os.path.join(_ROOT, "posegraph-ceres_amd")
_NAME = "posegraph_ceres_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(_PKG_DIR, "__init__.py"),
                                                  submodule_search_locations=[_PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def datasets():
    load()
    return importlib.import_module(_NAME + ".datasets")
