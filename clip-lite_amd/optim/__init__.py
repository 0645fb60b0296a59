from .fused_sgd import FusedSGD, Lookahead  # noqa: F401
from . import lr_scheduler  # noqa: F401
