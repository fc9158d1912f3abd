"""Import alias: the package directory is `v-express_amd/` (not a valid Python identifier), so this
module makes it importable as `v_express_amd` by pointing `__path__` at that directory and executing
its `__init__.py` in this namespace."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "v-express_amd")]
__package__ = "v_express_amd"
__file__ = _os.path.join(__path__[0], "__init__.py")
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
