"""Import alias: the package directory is named `curve25519-dalek_amd/` (a hyphen is not a legal
Python identifier), so this one-file module makes `import curve25519_dalek_amd` resolve to it."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "curve25519-dalek_amd")]
__package__ = __name__
if __spec__ is not None:
    __spec__.submodule_search_locations = __path__
with open(_os.path.join(__path__[0], "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
