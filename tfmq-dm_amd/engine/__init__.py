"""Device execution plans: the module tree of a (Quant)Model lowered to C-ABI kernel launches."""
from .ddim_unet import DdimUNetEngine, LayerQ, StopAt, UnitReached  # noqa: F401
from . import ddim_quant  # noqa: F401,E402
from .ldm_unet import LdmUNetEngine  # noqa: F401,E402
