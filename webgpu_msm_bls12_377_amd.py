"""Import shim: makes the hyphenated package directory ``webgpu-msm-bls12-377_amd/`` importable
as ``webgpu_msm_bls12_377_amd`` (a module with __path__ is a package to the import system)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "webgpu-msm-bls12-377_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
del _f
