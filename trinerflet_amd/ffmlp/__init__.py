from .ffmlp import FFMLP  # noqa: F401
