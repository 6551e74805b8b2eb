"""Import shim: `import lvae_amd` loads the package that lives in the directory `ladder-vae-pytorch_amd/`
(a hyphenated directory name is not importable by itself) and registers it as the package `lvae_amd`,
so that `import lvae_amd.models.lvae` etc. work."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'ladder-vae-pytorch_amd')
_spec = importlib.util.spec_from_file_location('lvae_amd', os.path.join(_dir, '__init__.py'),
                                               submodule_search_locations=[_dir])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['lvae_amd'] = _pkg
_spec.loader.exec_module(_pkg)
