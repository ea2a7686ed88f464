"""Import shim: `import lrcn_amd` loads the package kept in `long-term-recurrent-convolutional-nn_amd/`."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "long-term-recurrent-convolutional-nn_amd")
_spec = importlib.util.spec_from_file_location("lrcn_amd", os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules["lrcn_amd"] = _mod
_spec.loader.exec_module(_mod)
