"""Import alias for the package directory ``image-retrieval---thesis-2026_amd`` (whose name is
not a valid Python identifier).  ``import mirx`` and ``import mirx.<module>`` resolve to the
files in that directory; nothing lives here."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "image-retrieval---thesis-2026_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _fh:
    exec(compile(_fh.read(), _os.path.join(_real, "__init__.py"), "exec"))
del _fh
