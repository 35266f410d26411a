"""AWQ int4 linear method on the fused dequant + matrix-core GEMM.

Mirror of AWQConfig / AWQLinearMethod (python/sglang/srt/layers/quantization/awq.py:20-215): same parameter
names, shapes and dtypes (qweight int32 [K, N/8], qzeros int32 [K/G, N/8], scales [K/G, N]), same `apply`
contract.  `process_weights_after_loading` additionally builds the re-laid-out copy the fused kernel streams
(same byte count); the reference-layout qweight is dropped unless `keep_reference_layout` is set.

The reference computes `awq_dequantize` then `torch.matmul` for every batch size.  Here batches of up to
`fused_max_tokens` rows (decode) take the fused kernel; larger ones (prefill) dequantise once with the HIP kernel
and multiply with the library GEMM, as the reference does.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from .. import ops


class AWQConfig:
    """awq.py:20-112 (the fields the linear method reads)."""

    def __init__(self, weight_bits: int = 4, group_size: int = 128, zero_point: bool = True,
                 modules_to_not_convert: Optional[List[str]] = None) -> None:
        if weight_bits != 4:
            raise ValueError(f"Currently, only 4-bit weight quantization is supported for AWQ, but got {weight_bits} bits.")
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.zero_point = zero_point
        self.pack_factor = 32 // weight_bits
        self.modules_to_not_convert = modules_to_not_convert or []

    def get_name(self) -> str:
        return "awq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quant_config.json", "quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "AWQConfig":
        def pick(keys, default=None):
            for k in keys:
                if k in config:
                    return config[k]
            if default is None:
                raise ValueError(f"Cannot find any of {keys} in the model's quantization config.")
            return default

        return cls(pick(["w_bit", "bits"]), pick(["q_group_size", "group_size"]), pick(["zero_point"]),
                   pick(["modules_to_not_convert"], []))

    def get_quant_method(self, layer, prefix: str = ""):
        if any(m in prefix for m in self.modules_to_not_convert):
            return None
        return AWQLinearMethod(self)


class AWQLinearMethod:
    def __init__(self, quant_config: AWQConfig, fused_max_tokens: int = 256, keep_reference_layout: bool = False):
        self.quant_config = quant_config
        self.fused_max_tokens = fused_max_tokens
        self.keep_reference_layout = keep_reference_layout

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        """awq.py:125-190."""
        cfg = self.quant_config
        if input_size_per_partition % cfg.group_size != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        n = sum(output_partition_sizes)
        if n % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        k = input_size_per_partition
        dev = extra_weight_attrs.get("device", None)
        mk = lambda *shape, dtype: torch.nn.Parameter(torch.empty(*shape, dtype=dtype, device=dev), requires_grad=False)
        layer.register_parameter("qweight", mk(k, n // cfg.pack_factor, dtype=torch.int32))
        layer.register_parameter("qzeros", mk(k // cfg.group_size, n // cfg.pack_factor, dtype=torch.int32))
        layer.register_parameter("scales", mk(k // cfg.group_size, n, dtype=params_dtype))

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        """awq.py:192-195, plus the one-time re-layout for the matrix cores."""
        layer.qweight = torch.nn.Parameter(layer.qweight.data, requires_grad=False)
        layer.qzeros = torch.nn.Parameter(layer.qzeros.data.contiguous(), requires_grad=False)
        layer.scales = torch.nn.Parameter(layer.scales.data.contiguous(), requires_grad=False)
        k, n = layer.qweight.shape[0], layer.qweight.shape[1] * self.quant_config.pack_factor
        layer.sglk_fusable = k % 128 == 0 and n % 64 == 0 and self.quant_config.group_size % 128 == 0
        layer.sglk_packed = ops.awq_repack(layer.qweight.data.contiguous()) if layer.sglk_fusable else None
        layer.sglk_shape = (k, n)
        layer.sglk_workspace = None
        if layer.sglk_fusable and not self.keep_reference_layout:
            layer.qweight = None  # the fused path and the prefill path both read the re-laid-out copy

    def _workspace(self, layer, m: int, device) -> torch.Tensor:
        k, n = layer.sglk_shape
        need = max(ops.awq_gemm_workspace_bytes(m, n, k), 256)
        ws = layer.sglk_workspace
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.zeros(need, dtype=torch.uint8, device=device)  # ticket words start at zero
            layer.sglk_workspace = ws
        return ws

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        """awq.py:197-215."""
        k, n = layer.sglk_shape
        out_shape = x.shape[:-1] + (n,)
        x2 = x.reshape(-1, x.shape[-1])
        m = x2.shape[0]
        if layer.sglk_fusable and m <= self.fused_max_tokens and m > 0:
            out = ops.awq_gemm(x2.contiguous(), layer.sglk_packed, layer.scales, layer.qzeros, bias, None,
                               self._workspace(layer, m, x.device))
            return out.reshape(out_shape)
        if layer.qweight is not None:
            w = ops.awq_dequantize(layer.qweight, layer.scales, layer.qzeros)
        else:
            w = ops.awq_dequantize_packed(layer.sglk_packed, layer.scales, layer.qzeros)
        out = torch.matmul(x2, w)
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)
