"""Graph builders for the hot path (RetinaNet heads + distillation losses)."""
from . import retinanet_heads  # noqa: F401
