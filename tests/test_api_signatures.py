"""CPU: every Python mirror in this package against the reference's real interfaces.

tests/golden/api_signatures.json holds the reference's function names and parameter lists, extracted with `ast` by
tools/extract_api_signatures.py (nothing from the reference is imported).  A mirror is compatible when a call written
for the reference binds the same way: the reference's parameters come first, under the same names, in the same order,
what the reference makes optional is optional here with the same literal default (what it requires may be optional
here); anything the mirror adds is optional.  What is
deliberately not mirrored is listed here by name, so a reference function this package silently lacks fails the test.
"""
import inspect
import json
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
API = json.load(open(os.path.join(ROOT, "tests", "golden", "api_signatures.json")))


def problems(ref_params, fn, where):
    """List of incompatibilities of `fn` with the reference parameter list."""
    out = []
    ref = [p for p in ref_params if p["name"] != "self"]
    sig = [p for p in inspect.signature(fn).parameters.values() if p.name != "self"]
    ref_pos = [p for p in ref if p["kind"] == "positional"]
    mine_pos = [p for p in sig if p.kind in (p.POSITIONAL_ONLY, p.POSITIONAL_OR_KEYWORD)]
    for i, r in enumerate(ref_pos):
        if i >= len(mine_pos):
            out.append(f"{where}: parameter {i} `{r['name']}` is missing")
            continue
        m = mine_pos[i]
        if m.name != r["name"]:
            out.append(f"{where}: parameter {i} is `{m.name}`, the reference has `{r['name']}`")
            continue
        has = m.default is not inspect.Parameter.empty
        if r["has_default"] and not has:
            out.append(f"{where}: `{r['name']}` is optional in the reference and required here")
        elif has and "default" in r:
            try:
                same = repr(m.default) == r["default"] or m.default == eval(r["default"], {})  # literals only
            except Exception:
                same = True  # a non-literal default expression (a module constant): names match, value not comparable
            if not same:
                out.append(f"{where}: `{r['name']}` defaults to {m.default!r}, the reference to {r['default']}")
    for m in mine_pos[len(ref_pos):]:
        if m.default is inspect.Parameter.empty:
            out.append(f"{where}: extra parameter `{m.name}` must be optional")
    names = {p.name for p in sig}
    var_kw = any(p.kind == p.VAR_KEYWORD for p in sig)
    for r in ref:
        if r["kind"] == "keyword_only" and r["name"] not in names and not var_kw:
            out.append(f"{where}: keyword `{r['name']}` is missing")
    return out


# reference sgl_kernel functions outside the RadixAttention + quantized-matmul decode path (other attention libraries,
# NVIDIA-only formats, constrained decoding, expert-parallel plumbing): not mirrored on purpose
SGL_KERNEL_NOT_MIRRORED = {
    "flash_attn.py": {"is_fa3_supported", "maybe_contiguous", "flash_attn_with_kvcache", "flash_attn_varlen_func"},
    "sparse_flash_attn.py": {"maybe_contiguous", "sparse_attn_func", "sparse_attn_varlen_func",
                             "convert_vertical_slash_indexes", "convert_vertical_slash_indexes_mergehead"},
    "grammar.py": {"apply_token_bitmask_inplace_cuda"},
    "attention.py": {"lightning_attention_decode", "cutlass_mla_decode", "cutlass_mla_get_workspace_size"},
    "gemm.py": {"int8_scaled_mm", "sgl_per_token_group_quant_int8", "cutlass_scaled_fp4_mm", "scaled_fp4_quant",
                "qserve_w4a8_per_chn_gemm", "qserve_w4a8_per_group_gemm"},
    "moe.py": {"fp8_blockwise_scaled_grouped_mm", "prepare_moe_input"},
    "sampling.py": set(),
    "elementwise.py": set(),
    "speculative.py": set(),
    "utils.py": {"get_cuda_stream", "is_hopper_arch"},
    "allreduce.py": set(),
}


def test_sgl_kernel_mirror_matches_the_reference():
    import sglang_security_amd.sgl_kernel as mirror

    bad, missing = [], {}
    for fname, fns in API["sgl_kernel"].items():
        lacking = set()
        for name, rec in fns.items():
            fn = getattr(mirror, name, None)
            if fn is None:
                if not name.startswith("_"):
                    lacking.add(name)
                continue
            bad += problems(rec["params"], fn, f"sgl_kernel.{name} ({fname}:{rec['line']})")
        missing[fname] = lacking
    assert not bad, "\n".join(bad)
    for fname, lacking in missing.items():
        assert lacking == SGL_KERNEL_NOT_MIRRORED[fname], (fname, sorted(lacking ^ SGL_KERNEL_NOT_MIRRORED[fname]))
    # the ROCm all-reduce surface is defined under `if torch.version.hip` in the reference: the extractor must see it
    assert set(API["sgl_kernel"]["allreduce.py"]) == {
        "init_custom_ar", "all_reduce_reg", "all_reduce_unreg", "dispose", "meta_size", "register_buffer",
        "get_graph_buffer_ipc_meta", "register_graph_buffers", "allocate_meta_buffer", "get_meta_buffer_ipc_handle"}


def test_allreduce_torch_ops_carry_the_reference_schemas():
    """torch_extension_rocm.cc:25-55: the ops exist under torch.ops.sgl_kernel with the reference's schemas (the
    schema strings of that file, as data), and the Python-registered ops answer on the Meta key."""
    import torch

    import sglang_security_amd.sgl_kernel.allreduce  # noqa: F401  (loads common_ops.so)

    want = {
        "init_custom_ar": "sgl_kernel::init_custom_ar(Tensor meta, Tensor rank_data, str[] handles, int[] offsets, int rank, bool full_nvlink) -> int",
        "all_reduce_reg": "sgl_kernel::all_reduce_reg(int fa, Tensor inp, Tensor! out) -> ()",
        "all_reduce_unreg": "sgl_kernel::all_reduce_unreg(int fa, Tensor inp, Tensor reg_buffer, Tensor! out) -> ()",
        "register_buffer": "sgl_kernel::register_buffer(int fa, Tensor t, str[] handles, int[] offsets) -> ()",
    }
    def shape(schema):
        return ([(a.name, str(a.type), bool(a.alias_info and a.alias_info.is_write)) for a in schema.arguments],
                [str(r.type) for r in schema.returns])

    for name, schema in want.items():
        assert shape(getattr(torch.ops.sgl_kernel, name).default._schema) == shape(torch._C.parse_schema(schema)), name
    for name in ("dispose", "meta_size", "get_graph_buffer_ipc_meta", "register_graph_buffers", "allocate_meta_buffer",
                 "get_meta_buffer_ipc_handle"):
        assert hasattr(torch.ops.sgl_kernel, name)
    assert torch.ops.sgl_kernel.meta_size.default() > 0
    # fake / meta registrations of the Python-registered ops
    qw = torch.empty(256, 16, dtype=torch.int32, device="meta")
    sc = torch.empty(2, 128, dtype=torch.float16, device="meta")
    assert torch.ops.sgl_kernel.awq_dequantize(qw, sc, qw).shape == (256, 128)
    a = torch.empty(7, 64, dtype=torch.float8_e4m3fn, device="meta")
    b = torch.empty(64, 32, dtype=torch.float8_e4m3fn, device="meta")
    s1 = torch.empty(7, dtype=torch.float32, device="meta")
    out = torch.ops.sgl_kernel.fp8_scaled_mm(a, b, s1, s1, torch.bfloat16, None)
    assert out.shape == (7, 32) and out.dtype == torch.bfloat16
    x = torch.empty(3, 64, dtype=torch.bfloat16, device="meta")
    assert torch.ops.sgl_kernel.rmsnorm(x, x, x[0], 1e-6, False) is None


def _class_cases():
    from sglang_security_amd.attention.backend import SglkAttnBackend
    from sglang_security_amd.distributed.custom_all_reduce import CustomAllreduce
    from sglang_security_amd.distributed.parallel_state import GroupCoordinator
    from sglang_security_amd.mem_cache import memory_pool
    from sglang_security_amd.mem_cache.radix_cache import RadixCache
    from sglang_security_amd.private_service.private_client import PrivateJudgeClient
    from sglang_security_amd.quantization.awq import AWQLinearMethod
    from sglang_security_amd.quantization.w8a8_fp8 import W8A8Fp8LinearMethod

    return {
        # reference class: (mirror, methods that must match, methods deliberately absent)
        "AttentionBackend": (SglkAttnBackend,
                             ["init_forward_metadata", "init_cuda_graph_state", "init_forward_metadata_capture_cuda_graph",
                              "init_forward_metadata_replay_cuda_graph", "get_cuda_graph_seq_len_fill_value", "forward",
                              "forward_decode", "forward_extend"], {"support_triton"}),
        "RadixCache": (RadixCache,
                       ["__init__", "reset", "match_prefix", "insert", "cache_finished_req", "cache_unfinished_req",
                        "pretty_print", "total_size", "evict", "inc_lock_ref", "dec_lock_ref", "evictable_size",
                        "protected_size", "all_values_flatten", "take_events"], set()),
        "CustomAllreduce": (CustomAllreduce,
                            ["__init__", "capture", "should_custom_ar", "all_reduce_reg", "all_reduce_unreg",
                             "custom_all_reduce", "register_buffer", "register_graph_buffers", "close"],
                            # the NVIDIA shared-buffer path (cudaMalloc + cudaIpc through CudaRTLibrary)
                            {"create_shared_buffer", "free_shared_buffer", "all_reduce"}),
        "GroupCoordinator": (GroupCoordinator,
                             ["all_reduce", "reduce_scatter", "all_gather_into_tensor", "all_gather", "barrier", "destroy"],
                             # control plane (object / tensor-dict broadcast, point-to-point, pipeline neighbours)
                             {"__init__", "graph_capture", "gather", "broadcast", "broadcast_object",
                              "broadcast_object_list", "send_object", "recv_object", "broadcast_tensor_dict",
                              "send_tensor_dict", "recv_tensor_dict", "send", "recv", "first_rank", "last_rank",
                              "is_first_rank", "is_last_rank", "next_rank", "prev_rank"}),
        "PrivateJudgeClient": (PrivateJudgeClient, ["__init__", "update_privacy", "close"], set()),
        "ReqToTokenPool": (memory_pool.ReqToTokenPool, ["__init__", "write", "available_size", "alloc", "free", "clear"],
                           set()),
        "TokenToKVPoolAllocator": (memory_pool.TokenToKVPoolAllocator,
                                   ["__init__", "available_size", "get_kvcache", "alloc", "free", "free_group_begin",
                                    "free_group_end", "backup_state", "restore_state", "clear"], set()),
        "MHATokenToKVPool": (memory_pool.MHATokenToKVPool,
                             ["__init__", "get_key_buffer", "get_value_buffer", "get_kv_buffer", "set_kv_buffer",
                              "get_kv_size_bytes"],
                             # host off-loading / disaggregation transfer
                             {"get_contiguous_buf_infos", "get_flat_data", "transfer", "transfer_per_layer"}),
        "AWQLinearMethod": (AWQLinearMethod, ["__init__", "create_weights", "process_weights_after_loading", "apply"],
                            set()),
        "W8A8Fp8LinearMethod": (W8A8Fp8LinearMethod,
                                ["__init__", "create_weights", "process_weights_after_loading", "apply"], set()),
    }


@pytest.mark.parametrize("ref_class", ["AttentionBackend", "RadixCache", "CustomAllreduce", "GroupCoordinator",
                                       "PrivateJudgeClient", "ReqToTokenPool", "TokenToKVPoolAllocator",
                                       "MHATokenToKVPool", "AWQLinearMethod", "W8A8Fp8LinearMethod"])
def test_class_mirror_matches_the_reference(ref_class):
    mirror, must, absent = _class_cases()[ref_class]
    rec = API["classes"][ref_class]
    bad = []
    for name in must:
        assert name in rec["methods"], f"the reference {ref_class} has no method {name}"
        fn = getattr(mirror, name, None)
        if fn is None:
            bad.append(f"{ref_class}.{name} is missing")
            continue
        bad += problems(rec["methods"][name]["params"], fn, f"{ref_class}.{name} ({rec['file']}:{rec['methods'][name]['line']})")
    assert not bad, "\n".join(bad)
    public = {m for m in rec["methods"] if not m.startswith("_") or m == "__init__"}
    assert public - set(must) == absent, sorted((public - set(must)) ^ absent)


def test_communication_ops_match_the_reference():
    from sglang_security_amd.distributed import parallel_state

    fns = API["module_functions"]["python/sglang/srt/distributed/communication_op.py"]
    bad = []
    for name in ("tensor_model_parallel_all_reduce", "tensor_model_parallel_all_gather"):
        bad += problems(fns[name]["params"], getattr(parallel_state, name), name)
    assert not bad, "\n".join(bad)


def test_vllm_call_sites_bind_to_the_shim():
    """Every vllm call the reference makes on ROCm (positional count + keyword names) binds to the stand-in."""
    from sglang_security_amd import vllm_shim

    alias = {"vllm_rotary_embedding": "rotary_embedding"}
    n = 0
    for path, sites in API["vllm_call_sites"].items():
        for c in sites:
            name = c["callee"].split(".")[-1]
            fn = getattr(vllm_shim, alias.get(name, name))
            try:
                inspect.signature(fn).bind(*([None] * c["n_positional"]), **{k: None for k in c["keywords"]})
            except TypeError as e:
                raise AssertionError(f"{path}:{c['line']} {c['callee']}: {e}")
            n += 1
    assert n >= 8


def test_vllm_shim_installs_the_modules_the_reference_imports():
    import importlib

    from sglang_security_amd import vllm_shim

    saved = {k: v for k, v in sys.modules.items() if k == "vllm" or k.startswith("vllm.")}
    try:
        assert vllm_shim.install(force=True)
        ops = importlib.import_module("vllm._custom_ops")
        from vllm._custom_ops import fused_add_rms_norm, rms_norm, rotary_embedding  # noqa: F401  layernorm.py:36-37
        from vllm.model_executor.layers.activation import GeluAndMul, SiluAndMul  # noqa: F401  activation.py:172
        from vllm import _custom_ops as ops2  # fp8_utils.py:10

        assert ops is ops2 and ops.scaled_fp8_quant is vllm_shim.scaled_fp8_quant
        assert vllm_shim.install() is True      # its own stand-in is replaced, a real vllm would not be
    finally:
        for k in [k for k in sys.modules if k == "vllm" or k.startswith("vllm.")]:
            del sys.modules[k]
        sys.modules.update(saved)
