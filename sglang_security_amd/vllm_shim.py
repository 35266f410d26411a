"""The `vllm` entry points the reference imports on ROCm, on the HIP kernels of libsglk.

On a HIP build of torch the reference does not import `sgl_kernel` at all (`_is_cuda` is false); it takes its
norm, activation, rotary and fp8-quantisation kernels from vllm instead:

    layers/layernorm.py:36-37,73-85      from vllm._custom_ops import fused_add_rms_norm, rms_norm
    layers/activation.py:168-172         from vllm.model_executor.layers.activation import GeluAndMul, SiluAndMul
    layers/rotary_embedding.py:86-89,166-174   from vllm._custom_ops import rotary_embedding
    layers/quantization/fp8_utils.py:10,613-618  ops.scaled_fp8_quant(input_2d, input_scale, use_per_token_if_dynamic=...)
    layers/quantization/fp8_utils.py:490-499,626-635  ops.cutlass_scaled_mm(qinput, weight, out_dtype=, scale_a=, scale_b=, bias=)

    import sglang_security_amd.vllm_shim as shim; shim.install()     # before `import sglang`

registers `vllm`, `vllm._custom_ops` and `vllm.model_executor.layers.activation` in `sys.modules`, so that an
unmodified `python/sglang` on ROCm reaches libsglk through the names it already imports.  vllm itself is not part of
the reference tree; the argument lists below are the reference's call sites, and the arithmetic is pinned by the same
fixtures as the `sgl_kernel` mirror (the reference's own kernel-test formulas).  fp8 is OCP e4m3 (`float8_e4m3fn`),
the format gfx950 computes in.
"""
from __future__ import annotations

import sys
import types
from typing import Dict, Optional, Tuple

import torch

from . import ops as _ops
from ._lib import SglkError


def _rows(t: torch.Tensor) -> torch.Tensor:
    return t.reshape(-1, t.shape[-1])


# ---- vllm._custom_ops --------------------------------------------------------------------------------------
def rms_norm(out: torch.Tensor, input: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    """layernorm.py:84-85: out = rmsnorm(input) * weight."""
    _ops.rmsnorm(_rows(out), _rows(input), weight, epsilon, gemma=False)


def fused_add_rms_norm(input: torch.Tensor, residual: torch.Tensor, weight: torch.Tensor, epsilon: float) -> None:
    """layernorm.py:81-82: residual += input; input = rmsnorm(residual) * weight, both in place."""
    _ops.fused_add_rmsnorm(_rows(input), _rows(residual), weight, epsilon, gemma=False)


def _gated(out: torch.Tensor, x: torch.Tensor, act: int) -> None:
    if x.shape[-1] != 2 * out.shape[-1]:
        raise SglkError(f"gated activation: input width {x.shape[-1]} is not twice the output width {out.shape[-1]}")
    _ops.act_and_mul(_rows(out), _rows(x), act)


def silu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _gated(out, x, 0)


def gelu_tanh_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _gated(out, x, 1)


def gelu_and_mul(out: torch.Tensor, x: torch.Tensor) -> None:
    _gated(out, x, 2)


_fp32_caches: Dict[Tuple, torch.Tensor] = {}


def _fp32_cache(cache: torch.Tensor) -> torch.Tensor:
    """The rotary kernel reads fp32 cos/sin; the reference hands over the cache in the activation dtype on this path
    (rotary_embedding.py:83-84,166).  One widened copy is kept per cache tensor."""
    if cache.dtype == torch.float32:
        return cache
    key = (cache.data_ptr(), tuple(cache.shape), cache.dtype, cache._version, str(cache.device))
    hit = _fp32_caches.get(key)
    if hit is None:
        if len(_fp32_caches) > 16:
            _fp32_caches.clear()
        hit = _fp32_caches[key] = cache.float().contiguous()
    return hit


def rotary_embedding(positions: torch.Tensor, query: torch.Tensor, key: torch.Tensor, head_size: int,
                     cos_sin_cache: torch.Tensor, is_neox: bool) -> None:
    """rotary_embedding.py:166-174: query [..., Hq*head_size] and key [..., Hk*head_size] rotated in place over the
    first cos_sin_cache.shape[1] elements of every head."""
    T = positions.numel()
    q = query.view(T, -1)
    k = key.view(T, -1)
    _ops.rope(q, k, _fp32_cache(cos_sin_cache), positions.reshape(-1), head_size, interleave=not is_neox)


def scaled_fp8_quant(input: torch.Tensor, scale: Optional[torch.Tensor] = None,
                     num_token_padding: Optional[int] = None, scale_ub: Optional[torch.Tensor] = None,
                     use_per_token_if_dynamic: bool = False) -> Tuple[torch.Tensor, torch.Tensor]:
    """fp8_utils.py:525-530,613-618: (fp8 tensor, scale).  scale given: static per-tensor; else dynamic, one scale per
    token when use_per_token_if_dynamic, one for the tensor otherwise."""
    if input.dim() != 2:
        raise SglkError("scaled_fp8_quant: input must be 2-D")
    if scale_ub is not None:
        raise SglkError("scaled_fp8_quant: scale_ub is not implemented")
    M, K = input.shape
    rows = max(M, num_token_padding) if num_token_padding else M
    out = torch.empty(rows, K, dtype=torch.float8_e4m3fn, device=input.device)
    if rows > M:
        out[M:].view(torch.uint8).zero_()
    x = input.contiguous()
    if scale is not None:
        s = scale.to(torch.float32).reshape(1)
        _ops.per_tensor_quant_fp8(x, out[:M], s, True)
        return out, scale
    if use_per_token_if_dynamic:
        s = torch.empty(M, 1, dtype=torch.float32, device=input.device)
        _ops.per_token_quant_fp8(x, out[:M], s)
        return out, s
    s = torch.zeros(1, dtype=torch.float32, device=input.device)
    _ops.per_tensor_quant_fp8(x, out[:M], s, False)
    return out, s


def cutlass_scaled_mm(a: torch.Tensor, b: torch.Tensor, scale_a: torch.Tensor, scale_b: torch.Tensor,
                      out_dtype: torch.dtype, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
    """fp8_utils.py:626-635: a [M, K] fp8 row-major, b [K, N] fp8 column-major, scales per tensor (1 element) or per
    token / per channel."""
    M, N = a.shape[0], b.shape[1]
    sa = scale_a.to(torch.float32).reshape(-1)
    sb = scale_b.to(torch.float32).reshape(-1)
    if sa.numel() == 1:
        sa = sa.expand(M).contiguous()
    if sb.numel() == 1:
        sb = sb.expand(N).contiguous()
    return _ops.fp8_scaled_mm(a, b, sa, sb, out_dtype, bias)


def awq_dequantize(qweight: torch.Tensor, scales: torch.Tensor, zeros: torch.Tensor, split_k_iters: int = 0,
                   thx: int = 0, thy: int = 0) -> torch.Tensor:
    """vllm's spelling of the op quantization/awq.py:210 calls as awq_dequantize(qweight, scales, qzeros)."""
    return _ops.awq_dequantize(qweight, scales, zeros)


# ---- vllm.model_executor.layers.activation -------------------------------------------------------------------
class SiluAndMul(torch.nn.Module):
    """activation.py:168-172 takes this class from vllm when sgl_kernel is not importable: x [..., 2d] -> [..., d]."""

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty(x.shape[:-1] + (x.shape[-1] // 2,), dtype=x.dtype, device=x.device)
        silu_and_mul(out, x)
        return out


class GeluAndMul(torch.nn.Module):
    def __init__(self, approximate: str = "none"):
        super().__init__()
        if approximate not in ("none", "tanh"):
            raise ValueError(f"Unknown approximate mode: {approximate}")
        self.approximate = approximate

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        out = torch.empty(x.shape[:-1] + (x.shape[-1] // 2,), dtype=x.dtype, device=x.device)
        (gelu_and_mul if self.approximate == "none" else gelu_tanh_and_mul)(out, x)
        return out


CUSTOM_OPS = ("rms_norm", "fused_add_rms_norm", "silu_and_mul", "gelu_and_mul", "gelu_tanh_and_mul", "rotary_embedding",
              "scaled_fp8_quant", "cutlass_scaled_mm", "awq_dequantize")


def install(force: bool = False) -> bool:
    """Register the stand-in modules.  A real vllm that is importable wins unless `force`; returns whether the
    stand-ins were installed."""
    if not force and "vllm" not in sys.modules:
        import importlib.util

        try:
            if importlib.util.find_spec("vllm") is not None:
                return False
        except (ImportError, ValueError):
            pass
    elif not force and not getattr(sys.modules["vllm"], "__sglk_shim__", False):
        return False
    me = sys.modules[__name__]
    root = types.ModuleType("vllm")
    root.__sglk_shim__ = True
    root.__path__ = []  # a package: submodule imports consult sys.modules first
    custom = types.ModuleType("vllm._custom_ops")
    for name in CUSTOM_OPS:
        setattr(custom, name, getattr(me, name))
    mex = types.ModuleType("vllm.model_executor")
    mex.__path__ = []
    layers = types.ModuleType("vllm.model_executor.layers")
    layers.__path__ = []
    act = types.ModuleType("vllm.model_executor.layers.activation")
    act.SiluAndMul, act.GeluAndMul = SiluAndMul, GeluAndMul
    root._custom_ops, root.model_executor = custom, mex
    mex.layers, layers.activation = layers, act
    sys.modules.update({"vllm": root, "vllm._custom_ops": custom, "vllm.model_executor": mex,
                        "vllm.model_executor.layers": layers, "vllm.model_executor.layers.activation": act})
    return True
