from .dit import DiT  # noqa: F401
from .unett import UNetT  # noqa: F401
from .mmdit import MMDiT  # noqa: F401
