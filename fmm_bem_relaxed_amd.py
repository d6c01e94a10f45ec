"""Import shim: the package directory is named `fmm-bem-relaxed_amd` (not a valid Python identifier),
so `import fmm_bem_relaxed_amd` loads it from that directory under this name."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "fmm-bem-relaxed_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
