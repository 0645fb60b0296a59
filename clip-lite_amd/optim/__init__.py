from .fused_sgd import FusedAdamW, FusedSGD, Lookahead  # noqa: F401
from . import lr_scheduler  # noqa: F401
