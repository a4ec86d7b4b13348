"""Import shim: the package directory is `id-grec_amd/` (not a valid Python identifier), so
this module adopts that directory as its package path.  `import idgrec_amd.native` then
resolves to `id-grec_amd/native.py`."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "id-grec_amd")]
__version__ = "0.1.0"
