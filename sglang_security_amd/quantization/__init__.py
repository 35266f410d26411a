from .awq import AWQConfig, AWQLinearMethod  # noqa: F401
from .gptq import GPTQConfig, GPTQLinearMethod  # noqa: F401
from .w8a8_fp8 import W8A8Fp8Config, W8A8Fp8LinearMethod  # noqa: F401
