"""Import shim: the package directory is named `tfmq-dm_amd/` (not a valid Python identifier);
this module makes it importable as `tfmq_dm_amd` (`import tfmq_dm_amd.quant.quant_model` ...)."""
import os as _os

__path__ = [_os.path.join(_os.path.dirname(_os.path.abspath(__file__)), "tfmq-dm_amd")]
with open(_os.path.join(__path__[0], "__init__.py")) as _f:
    exec(compile(_f.read(), _os.path.join(__path__[0], "__init__.py"), "exec"))
