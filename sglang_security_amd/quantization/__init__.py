from .awq import AWQConfig, AWQLinearMethod  # noqa: F401
from .gptq import GPTQConfig, GPTQLinearMethod  # noqa: F401
