from .samplers import CategoriesSampler  # noqa: F401
