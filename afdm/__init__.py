"""Import alias: `import afdm` loads the package that lives in `aliasfree-diffusion-models-pytorch_amd/`
(a directory name Python cannot import directly because of the hyphens)."""
import os as _os

_pkg = _os.path.join(_os.path.dirname(_os.path.dirname(_os.path.abspath(__file__))),
                     "aliasfree-diffusion-models-pytorch_amd")
__path__ = [_pkg]
with open(_os.path.join(_pkg, "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(_pkg, "__init__.py"), "exec"))
