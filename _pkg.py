"""Import helper: the package directory is named `opencl-raytracer_amd` (not a valid Python
identifier), so it is registered under the module name `opencl_raytracer_amd`."""
import importlib.util
import pathlib
import sys

NAME = "opencl_raytracer_amd"
ROOT = pathlib.Path(__file__).resolve().parent
PKG_DIR = ROOT / "opencl-raytracer_amd"


def load():
    if NAME in sys.modules:
        return sys.modules[NAME]
    spec = importlib.util.spec_from_file_location(NAME, PKG_DIR / "__init__.py",
                                                  submodule_search_locations=[str(PKG_DIR)])
    mod = importlib.util.module_from_spec(spec)
    sys.modules[NAME] = mod
    spec.loader.exec_module(mod)
    return mod
