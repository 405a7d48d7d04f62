from .models import make, load, register, models  # noqa: F401
from . import visformer      # noqa: F401  registers 'visformer_micro_80'
from . import deit           # noqa: F401  registers the deit_* factories
from . import meta_baseline  # noqa: F401  registers 'meta-baseline'
from . import classifier     # noqa: F401  registers 'linear-classifier', 'token-label' (distillation phase)
