"""Loader for the hyphen-named package directory ``flow-pipeline_amd/``.

``import flow-pipeline_amd`` is not valid Python, so the package is registered
under the importable name ``flow_pipeline_amd``.
"""
import importlib.util
import os
import sys

ROOT = os.path.dirname(os.path.abspath(__file__))
PKG_DIR = os.path.join(ROOT, "flow-pipeline_amd")
NAME = "flow_pipeline_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(
        NAME, os.path.join(PKG_DIR, "__init__.py"), submodule_search_locations=[PKG_DIR])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod


def load_oracle():
    """TEST INFRASTRUCTURE ONLY (tests/, smoke(), bench.py cpu_baseline)."""
    name = "flow_oracle_py"
    if name in sys.modules:
        return sys.modules[name]
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "oracle", "pyoracle.py"))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[name] = mod
    spec.loader.exec_module(mod)
    return mod
