"""GPTQ int4 linear method on the fused dequant + matrix-core GEMM.

The reference takes GPTQLinearMethod from vllm 0.8.4 (python/sglang/srt/layers/quantization/gptq.py:18-31; vllm is not
part of the reference tree), so this mirror keeps the checkpoint-facing side of that method -- parameter names,
shapes and dtypes: qweight int32 [K/8, N], qzeros int32 [K/G, N/8], scales [K/G, N], g_idx int32 [K] -- and the
arithmetic of the published format: w[k, n] = scales[g, n] * (q[k, n] - (zeros[g, n] + 1)) with g = k // G for v1
checkpoints, zeros stored as they are for v2.  `process_weights_after_loading` re-lays the tensors out once into the
form the AWQ kernels stream, so decode batches take the fused int4 kernel and prefill dequantises + multiplies, as
for AWQ.  Activation reordering (desc_act: g_idx[k] names the group of input row k, in any order) is handled by one
row permutation at load time: the rows are sorted by group (a stable sort, as the exllama shuffle of vllm's
gptq_shuffle does), which makes every group 128 contiguous rows again, and `apply` gathers the activation columns in
the same order before the product.
"""
from __future__ import annotations

from typing import Any, Dict, List, Optional

import torch

from .. import ops


class GPTQConfig:
    """gptq.py:57-170 (the fields the linear method reads)."""

    def __init__(self, weight_bits: int = 4, group_size: int = 128, desc_act: bool = False,
                 lm_head_quantized: bool = False, dynamic: Optional[Dict[str, Dict[str, Any]]] = None,
                 checkpoint_format: str = "gptq") -> None:
        if weight_bits != 4:
            raise ValueError(f"Currently, only 4-bit weight quantization is supported for GPTQ here, but got {weight_bits} bits.")
        self.weight_bits = weight_bits
        self.group_size = group_size
        self.desc_act = desc_act
        self.lm_head_quantized = lm_head_quantized
        self.dynamic = dynamic or {}
        self.pack_factor = 32 // weight_bits
        self.checkpoint_format = checkpoint_format  # "gptq" (v1: zeros stored minus one) or "gptq_v2"

    def get_name(self) -> str:
        return "gptq"

    def get_supported_act_dtypes(self) -> List[torch.dtype]:
        return [torch.half, torch.bfloat16]

    @staticmethod
    def get_config_filenames() -> List[str]:
        return ["quantize_config.json"]

    @classmethod
    def from_config(cls, config: Dict[str, Any]) -> "GPTQConfig":
        return cls(config["bits"], config["group_size"], config.get("desc_act", False), config.get("lm_head", False),
                   config.get("dynamic", {}), config.get("checkpoint_format", "gptq"))

    def get_quant_method(self, layer, prefix: str = ""):
        return GPTQLinearMethod(self)


class GPTQLinearMethod:
    def __init__(self, quant_config: GPTQConfig, fused_max_tokens: int = 256):
        self.quant_config = quant_config
        self.fused_max_tokens = fused_max_tokens

    def create_weights(self, layer: torch.nn.Module, input_size_per_partition: int, output_partition_sizes: List[int],
                       input_size: int, output_size: int, params_dtype: torch.dtype, **extra_weight_attrs):
        cfg = self.quant_config
        k, n = input_size_per_partition, sum(output_partition_sizes)
        group = cfg.group_size if cfg.group_size != -1 else input_size
        if k % group != 0:
            raise ValueError("The input size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        if n % cfg.pack_factor != 0:
            raise ValueError("The output size is not aligned with the quantized weight shape. "
                             "This can be caused by too large tensor parallel size.")
        dev = extra_weight_attrs.get("device", None)
        mk = lambda *shape, dtype: torch.nn.Parameter(torch.empty(*shape, dtype=dtype, device=dev), requires_grad=False)
        layer.register_parameter("qweight", mk(k // cfg.pack_factor, n, dtype=torch.int32))
        layer.register_parameter("g_idx", torch.nn.Parameter(
            (torch.arange(k, dtype=torch.int32, device=dev) // group), requires_grad=False))
        layer.register_parameter("qzeros", mk(k // group, n // cfg.pack_factor, dtype=torch.int32))
        layer.register_parameter("scales", mk(k // group, n, dtype=params_dtype))
        layer.sglk_group = group

    def process_weights_after_loading(self, layer: torch.nn.Module) -> None:
        cfg = self.quant_config
        k, n = layer.qweight.shape[0] * cfg.pack_factor, layer.qweight.shape[1]
        group = getattr(layer, "sglk_group", cfg.group_size if cfg.group_size != -1 else k)
        g_idx = getattr(layer, "g_idx", None)
        layer.sglk_perm = None
        qweight = layer.qweight.data.contiguous()
        if g_idx is not None and g_idx.numel():
            plain = torch.arange(k, dtype=g_idx.dtype, device=g_idx.device) // group
            if not torch.equal(g_idx.data, plain):
                gi = g_idx.data.to(torch.int64)
                if gi.numel() != k or int(gi.min()) < 0 or int(gi.max()) >= k // group or \
                        not torch.all(torch.bincount(gi, minlength=k // group) == group):
                    raise ValueError("GPTQ g_idx must assign exactly group_size input rows to every group")
                perm = torch.argsort(gi, stable=True)
                # unpack the eight rows of every word, reorder whole rows, pack again
                sh = (torch.arange(8, device=qweight.device, dtype=torch.int64) * 4).view(1, 8, 1)
                q = ((qweight.to(torch.int64).unsqueeze(1) >> sh) & 0xF).reshape(k, n).index_select(0, perm)
                w = (q.reshape(k // 8, 8, n) << sh).sum(dim=1)
                qweight = torch.where(w >= 2 ** 31, w - 2 ** 32, w).to(torch.int32).contiguous()
                layer.sglk_perm = perm
        if k % 128 != 0 or n % 64 != 0 or group % 128 != 0:
            raise NotImplementedError(f"GPTQ layer [{k}, {n}] with group size {group}: the int4 kernels need "
                                      "K % 128 == 0, N % 64 == 0 and a group size that is a multiple of 128")
        offset = 0 if cfg.checkpoint_format == "gptq_v2" else 1
        packed, zeros = ops.gptq_repack(qweight, layer.qzeros.data.contiguous(), group, offset)
        layer.sglk_packed, layer.sglk_zeros = packed, zeros
        layer.scales = torch.nn.Parameter(layer.scales.data.contiguous(), requires_grad=False)
        layer.sglk_shape = (k, n)
        layer.sglk_workspace = None
        layer.qweight = None  # every path reads the re-laid-out copy
        layer.qzeros = None

    def _workspace(self, layer, m: int, device) -> torch.Tensor:
        k, n = layer.sglk_shape
        need = max(ops.awq_gemm_workspace_bytes(m, n, k), 256)
        ws = layer.sglk_workspace
        if ws is None or ws.numel() < need or ws.device != device:
            ws = torch.zeros(need, dtype=torch.uint8, device=device)  # ticket words start at zero
            layer.sglk_workspace = ws
        return ws

    def apply(self, layer: torch.nn.Module, x: torch.Tensor, bias: Optional[torch.Tensor] = None) -> torch.Tensor:
        k, n = layer.sglk_shape
        out_shape = x.shape[:-1] + (n,)
        x2 = x.reshape(-1, x.shape[-1])
        if getattr(layer, "sglk_perm", None) is not None:
            x2 = x2.index_select(1, layer.sglk_perm)  # activation columns in the order the rows were sorted into
        m = x2.shape[0]
        if 0 < m <= self.fused_max_tokens:
            out = ops.awq_gemm(x2.contiguous(), layer.sglk_packed, layer.scales, layer.sglk_zeros, bias, None,
                               self._workspace(layer, m, x.device))
            return out.reshape(out_shape)
        out = torch.matmul(x2, ops.awq_dequantize_packed(layer.sglk_packed, layer.scales, layer.sglk_zeros))
        if bias is not None:
            out.add_(bias)
        return out.reshape(out_shape)
