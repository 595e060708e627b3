"""Import alias: ``streaming_vlm_amd`` -> the package directory ``streaming-vlm_amd/`` next to it
(a hyphen cannot appear in a Python identifier).  No code lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "streaming-vlm_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__, "r", encoding="utf-8") as _f:
    exec(compile(_f.read(), __file__, "exec"))
