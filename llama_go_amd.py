"""Import shim: the package directory is `llama.go_amd/` (the name the project mandates), which is not
a valid Python identifier.  `import llama_go_amd` loads that directory as a regular package."""
import importlib.util as _u
import os as _os
import sys as _sys

_dir = _os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "llama.go_amd")
_spec = _u.spec_from_file_location("llama_go_amd", _os.path.join(_dir, "__init__.py"), submodule_search_locations=[_dir])
_mod = _u.module_from_spec(_spec)
_sys.modules["llama_go_amd"] = _mod
_spec.loader.exec_module(_mod)
