"""Same public names as ``f5_tts.model`` for the inference path (reference model/__init__.py:1-8)."""
from .backbones.dit import DiT  # noqa: F401
from .backbones.unett import UNetT  # noqa: F401
from .backbones.mmdit import MMDiT  # noqa: F401
from .cfm import CFM  # noqa: F401
from .duration_predictor import DurationPredictor  # noqa: F401
