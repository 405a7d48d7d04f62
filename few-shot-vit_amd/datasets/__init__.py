from .datasets import make, register, datasets  # noqa: F401
from .samplers import CategoriesSampler  # noqa: F401
from . import synthetic  # noqa: F401  registers 'synthetic-episodes'
from . import image_datasets  # noqa: F401  registers 'mini-imagenet', 'tiered-imagenet' (device-resident)
from . import folder_datasets  # noqa: F401  registers 'image-folder', 'cifar-fs'
