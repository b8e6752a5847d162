"""Importable alias of the package directory `adaptive-speculative-decoding_amd/`.

The directory name required by the project layout contains hyphens, which Python cannot import;
this shim makes `import asd_amd` (and `asd_amd.serving.pipeline`, ...) resolve to the files in that
directory.  It contains no code of its own.
"""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                      "adaptive-speculative-decoding_amd")
__path__ = [_real]
__file__ = _os.path.join(_real, "__init__.py")
with open(__file__) as _f:
    exec(compile(_f.read(), __file__, "exec"))
del _f
