from .awq import AWQConfig, AWQLinearMethod  # noqa: F401
