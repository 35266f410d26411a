"""TEST INFRASTRUCTURE ONLY: load pieces of the reference (read-only, /root/reference) in the
build container, to validate the C oracle and to mint golden vectors (oracle/gen_golden.py).

Nothing here runs on the GPU box (the reference does not travel); tests that use these loaders
skip themselves when /root/reference is absent.
"""
from __future__ import annotations

import importlib.util
import os
import sys
import types

REF = os.environ.get("SGLK_REFERENCE", "/root/reference")
_HERE = os.path.dirname(os.path.abspath(__file__))


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REF, "python", "sglang"))


# ----------------------------------------------------------------------------------------------
# compiled reference CPU kernels (oracle/_ref/common_ops_*.so, built by oracle/Makefile.ref)
# ----------------------------------------------------------------------------------------------
_ref_ops_loaded = None


def _cpu_flags():
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("flags"):
                    return set(line.split(":", 1)[1].split())
    except OSError:
        pass
    return set()


def ref_cpu_ops_path():
    """Pick the reference CPU library this host can execute, or None."""
    d = os.path.join(_HERE, "_ref")
    native, generic = os.path.join(d, "common_ops_native.so"), os.path.join(d, "common_ops_generic.so")
    info = os.path.join(d, "BUILD_INFO.txt")
    flags = _cpu_flags()
    if os.path.exists(native) and os.path.exists(info):
        need = set(open(info).read().split())
        if need and need <= flags:
            return native, "native"
    if os.path.exists(generic) and {"avx2", "fma"} <= flags:
        return generic, "generic"
    return None, None


def load_ref_cpu_ops():
    """torch.ops.sgl_kernel.*_cpu from the reference's own C++ (sgl-kernel/csrc/cpu)."""
    global _ref_ops_loaded
    if _ref_ops_loaded is None:
        import torch

        path, kind = ref_cpu_ops_path()
        if path is None:
            raise RuntimeError("reference CPU library not built or not runnable on this CPU")
        torch.ops.load_library(path)
        _ref_ops_loaded = kind
    import torch

    return torch.ops.sgl_kernel


# ----------------------------------------------------------------------------------------------
# reference Python modules by file path, with the minimum of stubbed package plumbing
# ----------------------------------------------------------------------------------------------
def _stub(name, **attrs):
    m = sys.modules.get(name)
    if m is None:
        m = types.ModuleType(name)
        m.__path__ = []  # behave as a package
        sys.modules[name] = m
    for k, v in attrs.items():
        setattr(m, k, v)
    return m


def _load(dotted, relpath):
    spec = importlib.util.spec_from_file_location(dotted, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    sys.modules[dotted] = mod
    spec.loader.exec_module(mod)
    return mod


def load_triton_ops():
    """Reference Triton attention kernels under the Triton CPU interpreter.

    Returns a namespace with decode_attention_fwd, extend_attention_fwd,
    create_flashinfer_kv_indices_triton, merge_state_triton, get_num_kv_splits_triton.
    """
    os.environ["TRITON_INTERPRET"] = "1"
    import logging

    logging.disable(logging.WARNING)
    for pkg in (
        "sglang",
        "sglang.srt",
        "sglang.srt.layers",
        "sglang.srt.layers.attention",
        "sglang.srt.layers.attention.triton_ops",
    ):
        _stub(pkg)
    _stub("sglang.srt.utils", is_hip=lambda: False, is_cuda=lambda: False)
    base = "python/sglang/srt/layers/attention/"
    ns = types.SimpleNamespace()
    dec = _load("sglang.srt.layers.attention.triton_ops.decode_attention", base + "triton_ops/decode_attention.py")
    _load("sglang.srt.layers.attention.triton_ops.prefill_attention", base + "triton_ops/prefill_attention.py")
    ext = _load("sglang.srt.layers.attention.triton_ops.extend_attention", base + "triton_ops/extend_attention.py")
    utl = _load("sglang.srt.layers.attention.utils", base + "utils.py")
    mrg = _load("sglang.srt.layers.attention.triton_ops.merge_state", base + "triton_ops/merge_state.py")
    ns.decode_attention_fwd = dec.decode_attention_fwd
    ns.extend_attention_fwd = ext.extend_attention_fwd
    ns.create_flashinfer_kv_indices_triton = utl.create_flashinfer_kv_indices_triton
    ns.merge_state_triton = mrg.merge_state_triton
    logging.disable(logging.NOTSET)
    return ns


class ScriptedJudgeClient:
    """Stand-in for PrivateJudgeClient (managers/private_service/private_client.py:70-97): records the
    nodes handed to update_privacy and lets a test decide them synchronously."""

    def __init__(self, decide=None):
        self.nodes = []
        self.decide = decide

    def update_privacy(self, node=None, context="", prompt=""):
        self.nodes.append(node)
        if self.decide is not None:
            private = self.decide(node, prompt)
            if private is not None:
                node.private = bool(private)
                node.need_check_privacy = False
        return None


class FakeAllocator:
    """TokenToKVPoolAllocator stand-in: records frees (memory_pool.py:169-235 interface)."""

    def __init__(self):
        import torch

        self.device = torch.device("cpu")
        self.freed = []

    def free(self, idx):
        import torch

        self.freed.append(torch.as_tensor(idx).clone())


class FakeReqPool:
    """ReqToTokenPool stand-in (memory_pool.py:54-99): the int32 table, write() and free()."""

    def __init__(self, rows=8, width=4096):
        import torch

        self.req_to_token = torch.zeros(rows, width, dtype=torch.int32)
        self.freed = []

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def free(self, idx):
        self.freed.append(int(idx))


class FakeReq:
    """The fields of managers/schedule_batch.py Req that cache_finished_req / cache_unfinished_req touch."""

    def __init__(self, rid, tokens, user, text=""):
        self.rid = rid
        self.origin_input_ids = list(tokens)
        self.output_ids = []
        self.fill_ids = []
        self.prefix_indices = []
        self.last_node = None
        self.req_pool_idx = None
        self.user_id = user
        self.origin_input_text = text


def load_radix_cache():
    """Reference RadixCache / TreeNode with package plumbing stubbed; epoch is settable through
    the returned namespace's set_epoch()."""
    state = {"epoch": 0}

    def get_epoch():
        return state["epoch"]

    _stub("zmq", NULL=None)
    _stub("sglang", get_epoch=get_epoch)
    _stub("sglang.srt")
    _stub("sglang.srt.disaggregation")
    ev = {n: type(n, (), {"__init__": lambda self, **kw: self.__dict__.update(kw)}) for n in
          ("AllBlocksCleared", "BlockRemoved", "BlockStored", "KVCacheEvent")}
    _stub("sglang.srt.disaggregation.kv_events", **ev)
    _stub("sglang.srt.managers")
    _stub("sglang.srt.managers.schedule_batch", global_server_args_dict={})
    _stub("sglang.srt.mem_cache")
    _stub("sglang.srt.mem_cache.memory_pool", ReqToTokenPool=object, TokenToKVPoolAllocator=object)
    _stub("sglang.srt.managers.private_service")
    _stub("sglang.srt.managers.private_service.private_client", PrivateJudgeClient=ScriptedJudgeClient)
    _load("sglang.srt.mem_cache.base_prefix_cache", "python/sglang/srt/mem_cache/base_prefix_cache.py")
    tn = _load("sglang.srt.mem_cache.tree_node", "python/sglang/srt/mem_cache/tree_node.py")
    rc = _load("sglang.srt.mem_cache.radix_cache", "python/sglang/srt/mem_cache/radix_cache.py")
    ns = types.SimpleNamespace(RadixCache=rc.RadixCache, TreeNode=tn.TreeNode, module=rc, tree_node=tn)

    def set_epoch(v):
        state["epoch"] = int(v)

    ns.set_epoch = set_epoch
    ns.get_epoch = get_epoch
    return ns
