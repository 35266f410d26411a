"""Extract, with `ast` only (nothing from the reference is imported or executed), the Python interfaces this
package mirrors, and write them as data to tests/golden/api_signatures.json:

  * every top-level function of sgl-kernel/python/sgl_kernel/*.py (name -> parameter list),
  * the methods of AttentionBackend, RadixCache / BasePrefixCache / TreeNode, CustomAllreduce, GroupCoordinator,
    PrivateJudgeClient and the KV-pool classes,
  * the argument shapes of the vllm calls the reference makes on ROCm (positional count + keyword names).

Run in the build container, where /root/reference exists:   python tools/extract_api_signatures.py
"""
import ast
import json
import os
import sys

REF = "/root/reference"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "api_signatures.json")


def literal(node):
    if node is None:
        return {"has_default": False}
    try:
        return {"has_default": True, "default": repr(ast.literal_eval(node))}
    except Exception:
        return {"has_default": True, "default": ast.unparse(node)}


def params(fn: ast.FunctionDef):
    a = fn.args
    out = []
    pos = a.posonlyargs + a.args
    defaults = [None] * (len(pos) - len(a.defaults)) + list(a.defaults)
    for arg, d in zip(pos, defaults):
        out.append({"name": arg.arg, "kind": "positional", **literal(d)})
    if a.vararg:
        out.append({"name": a.vararg.arg, "kind": "var_positional", "has_default": False})
    for arg, d in zip(a.kwonlyargs, a.kw_defaults):
        out.append({"name": arg.arg, "kind": "keyword_only", **literal(d)})
    if a.kwarg:
        out.append({"name": a.kwarg.arg, "kind": "var_keyword", "has_default": False})
    return out


def _is_hip_test(test) -> bool:
    """`torch.version.hip is not None` (how sgl_kernel/allreduce.py:5 selects its ROCm definitions)."""
    return "torch.version.hip" in ast.unparse(test)


def module_functions(path):
    """Top-level functions, plus those defined under a module-level `if torch.version.hip ...:` -- the ROCm branch is
    the interface this package mirrors; the `else:` (NVIDIA) branch of such a test is left out."""
    tree = ast.parse(open(path).read())
    out = {}
    for n in tree.body:
        if isinstance(n, ast.FunctionDef):
            out[n.name] = {"line": n.lineno, "params": params(n)}
        elif isinstance(n, ast.If) and _is_hip_test(n.test):
            for m in n.body:
                if isinstance(m, ast.FunctionDef):
                    out[m.name] = {"line": m.lineno, "params": params(m), "branch": "torch.version.hip"}
    return out


def class_methods(path, cls):
    tree = ast.parse(open(path).read())
    for n in ast.walk(tree):
        if isinstance(n, ast.ClassDef) and n.name == cls:
            meths = {}
            for m in n.body:
                if isinstance(m, ast.FunctionDef):
                    deco = [ast.unparse(d) for d in m.decorator_list]
                    meths[m.name] = {"line": m.lineno, "params": params(m), "decorators": deco}
            return {"line": n.lineno, "bases": [ast.unparse(b) for b in n.bases], "methods": meths}
    raise KeyError(f"{cls} not in {path}")


def calls(path, names):
    """Every call of one of `names` (the last attribute or the bare name): positional count and keyword names."""
    tree = ast.parse(open(path).read())
    found = []
    for n in ast.walk(tree):
        if isinstance(n, ast.Call):
            f = n.func
            last = f.attr if isinstance(f, ast.Attribute) else f.id if isinstance(f, ast.Name) else None
            if last in names:
                found.append({"callee": ast.unparse(f), "line": n.lineno, "n_positional": len(n.args),
                              "keywords": [k.arg for k in n.keywords]})
    return sorted(found, key=lambda c: c["line"])


def main():
    if not os.path.isdir(REF):
        sys.exit("the reference tree is not here; the committed tests/golden/api_signatures.json stays as it is")
    srt = os.path.join(REF, "python", "sglang", "srt")
    sk = os.path.join(REF, "sgl-kernel", "python", "sgl_kernel")
    doc = {"source": "ast of the reference tree; no reference code was imported", "sgl_kernel": {}, "classes": {},
           "vllm_call_sites": {}}
    for f in sorted(os.listdir(sk)):
        if f.endswith(".py") and f not in ("__init__.py", "version.py"):
            doc["sgl_kernel"][f] = module_functions(os.path.join(sk, f))
    for rel, classes in {
        "layers/attention/base_attn_backend.py": ["AttentionBackend"],
        "layers/attention/triton_backend.py": ["TritonAttnBackend"],
        "mem_cache/base_prefix_cache.py": ["BasePrefixCache"],
        "mem_cache/radix_cache.py": ["RadixCache"],
        "mem_cache/tree_node.py": ["TreeNode"],
        "mem_cache/memory_pool.py": ["ReqToTokenPool", "TokenToKVPoolAllocator", "MHATokenToKVPool", "MLATokenToKVPool"],
        "distributed/device_communicators/custom_all_reduce.py": ["CustomAllreduce"],
        "distributed/parallel_state.py": ["GroupCoordinator"],
        "managers/private_service/private_client.py": ["PrivateJudgeClient"],
        "layers/quantization/awq.py": ["AWQConfig", "AWQLinearMethod"],
        "layers/quantization/w8a8_fp8.py": ["W8A8Fp8Config", "W8A8Fp8LinearMethod"],
    }.items():
        for c in classes:
            doc["classes"][c] = {"file": "python/sglang/srt/" + rel, **class_methods(os.path.join(srt, rel), c)}
    doc["module_functions"] = {
        "python/sglang/srt/distributed/communication_op.py": module_functions(os.path.join(srt, "distributed", "communication_op.py")),
    }
    for rel, names in {
        "layers/layernorm.py": {"rms_norm", "fused_add_rms_norm"},
        "layers/rotary_embedding.py": {"vllm_rotary_embedding"},
        "layers/quantization/fp8_utils.py": {"scaled_fp8_quant", "cutlass_scaled_mm"},
        "layers/quantization/awq.py": {"awq_dequantize"},
    }.items():
        doc["vllm_call_sites"]["python/sglang/srt/" + rel] = calls(os.path.join(srt, rel), names)
    with open(OUT, "w") as f:
        json.dump(doc, f, indent=1, sort_keys=True)
        f.write("\n")
    n = sum(len(v) for v in doc["sgl_kernel"].values())
    print(f"wrote {OUT}: {n} sgl_kernel functions, {len(doc['classes'])} classes")


if __name__ == "__main__":
    main()
