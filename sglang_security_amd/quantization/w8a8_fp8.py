"""w8a8 fp8 linear method (per-channel weight scales, per-token dynamic activation scales) on the HIP kernels.

Mirror of W8A8Fp8Config / W8A8Fp8LinearMethod (python/sglang/srt/layers/quantization/w8a8_fp8.py:28-182): same
parameter names, shapes and dtypes (weight [N, K] float8_e4m3fn or the model's type, weight_scale [N, 1] float32),
the same `apply` contract (apply_fp8_linear, layers/quantization/fp8_utils.py:454-693: per-token quantisation of the
activations, scaled product, bias, output in the activation type).  A checkpoint that is not fp8-serialised is
quantised per output channel at load time, as the reference does when its scaled product takes per-channel scales
(:118-127).  `process_weights_after_loading` additionally builds the sequential-streaming copy the decode-size product
reads (same byte count; inner dimensions of 4096 and more, where that kernel is the faster one).
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from .. import ops

F8 = torch.float8_e4m3fn


class W8A8Fp8Config:
    """w8a8_fp8.py:28-98 (the fields the linear method reads)."""

    def __init__(self, is_checkpoint_fp8_serialized: bool = False):
        self.is_checkpoint_fp8_serialized = is_checkpoint_fp8_serialized

    @classmethod
    def get_supported_act_dtypes(cls) -> List[torch.dtype]:
        return [torch.float16, torch.bfloat16]

    @classmethod
    def get_name(cls) -> str:
        return "w8a8_fp8"

    @classmethod
    def get_config_filenames(cls) -> List[str]:
        return []

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "W8A8Fp8Config":
        quant_method = config.get("quant_method")
        if quant_method is None:
            raise ValueError("Cannot find any of ['quant_method'] in the model's quantization config.")
        return cls(is_checkpoint_fp8_serialized="compressed-tensors" in quant_method or "w8a8_fp8" in quant_method)

    def get_quant_method(self, layer, prefix: str = ""):
        return W8A8Fp8LinearMethod(self)

    def get_scaled_act_names(self) -> List[str]:
        return []


class W8A8Fp8LinearMethod:
    def __init__(self, quantization_config: W8A8Fp8Config, streaming_max_tokens: int = 256):
        self.quantization_config = quantization_config
        self.streaming_max_tokens = streaming_max_tokens

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        """w8a8_fp8.py:129-168."""
        serialized = self.quantization_config.is_checkpoint_fp8_serialized
        n = sum(output_partition_sizes)
        dev = extra_weight_attrs.get("device", None)
        self.logical_widths = output_partition_sizes
        layer.register_parameter("weight", torch.nn.Parameter(
            torch.empty(n, input_size_per_partition, dtype=F8 if serialized else params_dtype, device=dev),
            requires_grad=False))
        if serialized:
            layer.register_parameter("weight_scale", torch.nn.Parameter(
                torch.empty((n, 1), dtype=torch.float32, device=dev), requires_grad=False))
        else:
            layer.weight_scale = None

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        """w8a8_fp8.py:101-127: the weight ends up as the [K, N] view of an e4m3 [N, K] tensor with one scale per
        output channel; plus the one-time streaming copy."""
        w = layer.weight.data
        if self.quantization_config.is_checkpoint_fp8_serialized:
            qweight = w.contiguous()
            scale = layer.weight_scale.data.detach().float().reshape(-1, 1).contiguous()
        else:
            # per output channel: the reference's per_token_group_quant_fp8(weight, K), i.e. one group per row
            n, k = w.shape
            w2 = w.contiguous()
            qweight = torch.empty(n, k, dtype=F8, device=w.device)
            scale = torch.empty(n, dtype=torch.float32, device=w.device)
            ops.per_token_quant_fp8(w2, qweight, scale)
            scale = scale.reshape(n, 1)
            layer.input_scale = None
        n, k = qweight.shape
        layer.weight = torch.nn.Parameter(qweight.t(), requires_grad=False)
        layer.weight_scale = torch.nn.Parameter(scale, requires_grad=False)
        layer.sglk_shape = (k, n)
        layer.sglk_packed = ops.fp8_repack(qweight) if (n % 16 == 0 and k % 128 == 0 and k >= 4096) else None
        layer.sglk_workspace = None

    def _workspace(self, layer, m: int, device) -> torch.Tensor:
        k, n = layer.sglk_shape
        need = max(ops.fp8_mm_packed_workspace_bytes(m, n, k), ops.fp8_mm_workspace_bytes(m, n, k), 256)
        ws = layer.sglk_workspace
        if ws is None or ws.numel() < need or ws.device != device:
            ws = layer.sglk_workspace = torch.empty(need, dtype=torch.uint8, device=device)
        return ws

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """w8a8_fp8.py:170-182 -> apply_fp8_linear with dynamic per-token activation scales."""
        k, n = layer.sglk_shape
        out_shape = x.shape[:-1] + (n,)
        x2 = x.reshape(-1, k).contiguous()
        m = x2.shape[0]
        if m == 0:
            return x.new_empty(out_shape)
        xq = torch.empty(m, k, dtype=F8, device=x.device)
        xs = torch.empty(m, dtype=torch.float32, device=x.device)
        ops.per_token_quant_fp8(x2, xq, xs)
        ws = self._workspace(layer, m, x.device)
        if layer.sglk_packed is not None and m <= self.streaming_max_tokens:
            out = ops.fp8_scaled_mm_packed(xq, layer.sglk_packed, xs, layer.weight_scale, x.dtype, bias, None, ws)
        else:
            out = ops.fp8_scaled_mm(xq, layer.weight, xs, layer.weight_scale, x.dtype, bias, None, ws)
        return out.reshape(out_shape)
