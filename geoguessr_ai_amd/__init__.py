"""Import alias: the package directory is ``geoguessr-ai_amd/`` (not a valid Python identifier), so
``import geoguessr_ai_amd`` resolves its submodules from there."""
import os as _os

_real = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))), "geoguessr-ai_amd")
__path__ = [_real]
with open(_os.path.join(_real, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_real, "__init__.py"), "exec"))
