"""Import shim: `import sol_amd` loads the package directory `solver-in-the-loop_amd/`
(its name carries a hyphen, so it cannot be imported by name)."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "solver-in-the-loop_amd")
_spec = importlib.util.spec_from_file_location("sol_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["sol_amd"] = _mod
_spec.loader.exec_module(_mod)
