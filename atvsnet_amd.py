"""Import shim: ``import atvsnet_amd`` loads the package kept in ``a-tvsnet_amd/``.

The product directory carries the reference's name (a hyphen is not a legal
Python identifier), so this module loads it under the importable name
``atvsnet_amd`` and replaces itself in ``sys.modules``; relative imports
inside the package then resolve to ``atvsnet_amd.*`` (one module identity).
"""
import importlib.util
import os
import sys

_root = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'a-tvsnet_amd')
_spec = importlib.util.spec_from_file_location(
    'atvsnet_amd', os.path.join(_root, '__init__.py'), submodule_search_locations=[_root])
_pkg = importlib.util.module_from_spec(_spec)
sys.modules['atvsnet_amd'] = _pkg
_spec.loader.exec_module(_pkg)
