"""Root-level loader for the product package.

The package directory is named `multimodalmusig.jl_amd` (contains a dot), which Python's import statement
cannot spell; this shim loads it under the module name `multimodalmusig_jl_amd`.
"""
import importlib.util
import os
import sys

_NAME = "multimodalmusig_jl_amd"


def load():
    if _NAME in sys.modules:
        return sys.modules[_NAME]
    pkg_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "multimodalmusig.jl_amd")
    spec = importlib.util.spec_from_file_location(_NAME, os.path.join(pkg_dir, "__init__.py"),
                                                  submodule_search_locations=[pkg_dir])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[_NAME] = mod
    spec.loader.exec_module(mod)
    return mod
