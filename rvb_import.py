"""Loads the package directory `parallel-reverb-raytracer_amd/` (not a valid Python identifier)
under the importable name `parallel_reverb_raytracer_amd`."""
import importlib.util
import os
import sys

PACKAGE_NAME = "parallel_reverb_raytracer_amd"
PACKAGE_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "parallel-reverb-raytracer_amd")


def load():
    if PACKAGE_NAME in sys.modules:
        return sys.modules[PACKAGE_NAME]
    spec = importlib.util.spec_from_file_location(
        PACKAGE_NAME, os.path.join(PACKAGE_DIR, "__init__.py"), submodule_search_locations=[PACKAGE_DIR])
    module = importlib.util.module_from_spec(spec)
    sys.modules[PACKAGE_NAME] = module
    spec.loader.exec_module(module)
    return module
