"""Import shim: makes the hyphenated package directory `graph-normalizing-flows_amd/` importable as
`gnf_amd` (a Python identifier cannot contain '-').  `import gnf_amd.gnn` etc. resolve inside it."""
import importlib.util
import os
import sys

_dir = os.path.join(os.path.dirname(os.path.abspath(__file__)), "graph-normalizing-flows_amd")
_spec = importlib.util.spec_from_file_location(__name__, os.path.join(_dir, "__init__.py"),
                                               submodule_search_locations=[_dir])
_mod = importlib.util.module_from_spec(_spec)
sys.modules[__name__] = _mod
_spec.loader.exec_module(_mod)
