from .bigvgan import BigVGANGenerator  # noqa: F401
from .hifigan import HiFiGANGenerator  # noqa: F401
from .refinegan import RefineGANGenerator  # noqa: F401
from .unify import UnifyGenerator  # noqa: F401
from .vocos import ISTFTHead  # noqa: F401
