from .registration import make, register, registry  # noqa: F401
