"""Import shim: the package directory is named `linearresponsevariationalbayes.py_amd` (the name
this build was asked to use), which contains a dot and therefore cannot be imported with a plain
`import` statement.  `import lrvb_amd` loads that directory as the package `lrvb_amd`."""
import importlib.util
import os
import sys

_PKG_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'linearresponsevariationalbayes.py_amd')
_spec = importlib.util.spec_from_file_location(
    'lrvb_amd', os.path.join(_PKG_DIR, '__init__.py'), submodule_search_locations=[_PKG_DIR])
_module = importlib.util.module_from_spec(_spec)
sys.modules['lrvb_amd'] = _module
_spec.loader.exec_module(_module)
