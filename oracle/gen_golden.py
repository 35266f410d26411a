"""TEST INFRASTRUCTURE ONLY: mint golden vectors under tests/golden/ from the reference itself.

Runs only in the build container (needs /root/reference and oracle/_ref):

    python oracle/gen_golden.py

Sources of truth used, per fixture:
  * reference Triton kernels run under the Triton CPU interpreter (decode / extend attention,
    kv-index gather, merge_state) -- python/sglang/srt/layers/attention/triton_ops/*.py, utils.py
  * reference CPU C++ kernels compiled from sgl-kernel/csrc/cpu (decode_attention_cpu,
    extend_attention_cpu, rmsnorm_cpu, silu_and_mul_cpu, weight_packed_linear)
  * the torch reference functions held inside the reference's own kernel tests
    (sgl-kernel/tests/test_awq_dequant.py, test_norm.py, test_activation.py,
    test_rotary_embedding.py, test_fp8_gemm.py, test_fp8_blockwise_gemm.py,
    test_per_token_quant_fp8.py, test_merge_state_v2.py), loaded by file path with the
    `sgl_kernel` import stubbed
  * the reference RadixCache (python/sglang/srt/mem_cache/radix_cache.py) driven through
    scripted traces.

Fixtures are data only (inputs + expected outputs) in .npz; half-precision tensors are stored as
raw uint16 words next to a `<name>__dtype` tag.
"""
from __future__ import annotations

import importlib.util
import json
import math
import os
import random
import sys
import types

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import ref_loaders as RL  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
REF = RL.REF


def pack(d):
    out = {}
    for k, v in d.items():
        if isinstance(v, torch.Tensor):
            if v.dtype in (torch.bfloat16, torch.float16):
                out[k] = v.contiguous().view(torch.int16).numpy().view(np.uint16)
                out[k + "__dtype"] = np.array(str(v.dtype).split(".")[1])
            elif v.dtype == torch.float8_e4m3fn:
                out[k] = v.contiguous().view(torch.uint8).numpy()
                out[k + "__dtype"] = np.array("float8_e4m3fn")
            else:
                out[k] = v.contiguous().numpy()
        else:
            out[k] = np.array(v)
    return out


def save(name, d):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **pack(d))
    print("wrote", name)


def load_test_module(relpath, name):
    """Load one of the reference's kernel test files with `sgl_kernel` (and sglang.srt.utils) stubbed,
    to reach the torch reference functions it defines."""
    stub = types.ModuleType("sgl_kernel")

    def _missing(*a, **k):
        raise RuntimeError("sgl_kernel stub")

    stub.__getattr__ = lambda attr: _missing
    sys.modules["sgl_kernel"] = stub
    RL._stub("sglang")
    RL._stub("sglang.srt")
    RL._stub("sglang.srt.utils", is_hip=lambda: False, is_cuda=lambda: False)
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, relpath))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


# ------------------------------------------------------------------------------------------------
def gen_kv_indices(tri):
    torch.manual_seed(42)
    for tag, bs, ctx in (("small", 7, 300), ("ragged", 37, 1100)):
        r2t = torch.randint(0, 100000, (bs + 5, ctx), dtype=torch.int32)
        idx = torch.randperm(bs + 5)[:bs].to(torch.int32)
        lens = torch.randint(1, ctx, (bs,), dtype=torch.int32)
        lens[0] = ctx
        if bs > 3:
            lens[3] = 1
        ptr = torch.zeros(bs + 1, dtype=torch.int32)
        ptr[1:] = torch.cumsum(lens, 0)
        out = torch.empty(int(ptr[-1]), dtype=torch.int32)
        tri.create_flashinfer_kv_indices_triton[(bs,)](r2t, idx, lens, ptr, None, out, r2t.stride(0))
        # with a start offset (sliding-window form, triton_backend.py:93-102)
        start = torch.minimum(torch.randint(0, 50, (bs,), dtype=torch.int32), ctx - lens)
        out2 = torch.empty(int(ptr[-1]), dtype=torch.int32)
        tri.create_flashinfer_kv_indices_triton[(bs,)](r2t, idx, lens, ptr, start, out2, r2t.stride(0))
        save(f"kv_indices_{tag}", dict(req_to_token=r2t, req_pool_indices=idx, lens=lens, kv_indptr=ptr,
                                        kv_start=start, expect=out, expect_start=out2))


def _mk_pool(n, hkv, d, dv, dtype):
    return torch.randn(n, hkv, d).to(dtype), torch.randn(n, hkv, dv).to(dtype)


def gen_decode(tri, ops):
    torch.manual_seed(42)
    cases = [
        # tag, B, Hq, Hkv, D, Dv, seq lens, splits, logit_cap, shared_prefix
        ("gqa128", 3, 32, 8, 128, 128, [200, 77, 33], [4, 2, 1], 0.0, 0),
        ("mha64", 2, 4, 4, 64, 64, [130, 64], [8, 3], 0.0, 0),
        ("mqa_cap", 2, 16, 1, 64, 64, [96, 41], [2, 2], 30.0, 0),
        ("mla", 1, 16, 1, 576, 512, [70], [2], 0.0, 0),
        ("odd", 2, 8, 4, 33, 55, [50, 19], [2, 1], 0.0, 0),
        ("shared", 4, 32, 8, 128, 128, [160, 150, 140, 129], [4, 4, 4, 4], 0.0, 128),
        ("gqa_q05", 2, 14, 2, 64, 64, [128, 5], [4, 1], 0.0, 0),
    ]
    for tag, B, Hq, Hkv, D, Dv, lens, splits, cap, shared in cases:
        n = sum(lens) + 9
        kb, vb = _mk_pool(n, Hkv, D, Dv, torch.bfloat16)
        q = torch.randn(B, Hq, D).to(torch.bfloat16)
        ctx = max(lens) + 4
        r2t = torch.zeros(B, ctx, dtype=torch.int32)
        perm = torch.randperm(n - 1)[: sum(lens)] + 1  # slot 0 is the pool's dummy row
        off = 0
        for b, L in enumerate(lens):
            r2t[b, :L] = perm[off : off + L]
            off += L
        if shared:
            r2t[:, :shared] = r2t[0, :shared]
        seq = torch.tensor(lens, dtype=torch.int64)
        ptr = torch.zeros(B + 1, dtype=torch.int32)
        ptr[1:] = torch.cumsum(seq, 0)
        ind = torch.cat([r2t[b, :L] for b, L in enumerate(lens)])
        ns = torch.tensor(splits, dtype=torch.int32)
        S = 8
        sm = 1.0 / math.sqrt(D)
        # (1) Triton kernel on fp32 copies of the bf16 data: isolates the algorithm from rounding
        o_t = torch.zeros(B, Hq, Dv)
        al, lse = torch.zeros(B, Hq, S, Dv), torch.zeros(B, Hq, S)
        tri.decode_attention_fwd(q.float(), kb.float(), vb.float(), o_t, ptr, ind, al, lse, ns, S, sm, cap)
        d = dict(q=q, k_buffer=kb, v_buffer=vb, req_to_token=r2t, seq_lens=seq, kv_indptr=ptr, kv_indices=ind,
                 num_kv_splits=ns, sm_scale=sm, logit_cap=cap, expect_triton_f32=o_t)
        # (2) compiled reference CPU kernel, bf16 in/out (head_size % 32 alignment is not required by
        # decode.cpp, but logit_cap is supported)
        o_c = torch.zeros(B, Hq, Dv, dtype=torch.bfloat16)
        key, val = torch.randn(B, Hkv, D).to(torch.bfloat16), torch.randn(B, Hkv, Dv).to(torch.bfloat16)
        loc = torch.tensor([int(r2t[b, L - 1]) for b, L in enumerate(lens)], dtype=torch.int64)
        key = kb[loc].clone()
        val = vb[loc].clone()  # rewrite the last token with itself: pools stay as given
        # decode.cpp leaves the partial of an EMPTY split unwritten and still merges it (decode.cpp:789-824
        # walks all splits of attn_logits), so with torch.empty scratch a sequence shorter than its split
        # count gives garbage (seen here: 33 tokens, 8 splits, |err| 0.07-0.17, varying run to run).  The
        # vector is therefore taken with a split count every sequence can fill.
        S_cpu = max(1, min(S, min(lens) // 16))
        logits = torch.empty(B, Hq, S_cpu, Dv + 1)
        ops.decode_attention_cpu(q, kb.clone(), vb.clone(), o_c, key, val, loc, logits, r2t,
                                 torch.arange(B, dtype=torch.int64), seq, sm, cap)
        d["expect_cpu_bf16"] = o_c
        save(f"decode_{tag}", d)


def gen_extend(tri, ops):
    torch.manual_seed(42)
    cases = [
        ("gqa128", 2, 8, 2, 128, 128, [40, 0], [33, 70], 0.0),
        ("mha64_cap", 2, 4, 4, 64, 64, [17, 5], [20, 9], 20.0),
    ]
    for tag, B, Hq, Hkv, D, Dv, pre, ext, cap in cases:
        n = sum(pre) + sum(ext) + 5
        kb, vb = _mk_pool(n, Hkv, D, Dv, torch.bfloat16)
        T = sum(ext)
        q = torch.randn(T, Hq, D).to(torch.bfloat16)
        ke, ve = torch.randn(T, Hkv, D).to(torch.bfloat16), torch.randn(T, Hkv, Dv).to(torch.bfloat16)
        ctx = max(p + e for p, e in zip(pre, ext)) + 2
        r2t = torch.zeros(B, ctx, dtype=torch.int32)
        perm = torch.randperm(n - 1) + 1
        off = 0
        for b in range(B):
            L = pre[b] + ext[b]
            r2t[b, :L] = perm[off : off + L]
            off += L
        qo = torch.zeros(B + 1, dtype=torch.int32)
        qo[1:] = torch.cumsum(torch.tensor(ext), 0)
        kp = torch.zeros(B + 1, dtype=torch.int32)
        kp[1:] = torch.cumsum(torch.tensor(pre), 0)
        ki = torch.cat([r2t[b, : pre[b]] for b in range(B)]) if sum(pre) else torch.zeros(0, dtype=torch.int32)
        sm = 1.0 / math.sqrt(D)
        o_t = torch.zeros(T, Hq, Dv)
        tri.extend_attention_fwd(q.float(), ke.float(), ve.float(), o_t, kb.float(), vb.float(), qo, kp, ki,
                                 None, True, None, max(ext), sm, cap)
        d = dict(q_extend=q, k_extend=ke, v_extend=ve, k_buffer=kb, v_buffer=vb, qo_indptr=qo, kv_indptr=kp,
                 kv_indices=ki, req_to_token=r2t, prefix_lens=torch.tensor(pre), extend_lens=torch.tensor(ext),
                 sm_scale=sm, logit_cap=cap, expect_triton_f32=o_t)
        # reference CPU kernel: the new K/V must already sit in the pool at their slots
        kb2, vb2 = kb.clone(), vb.clone()
        for b in range(B):
            sl = r2t[b, pre[b] : pre[b] + ext[b]].long()
            kb2[sl] = ke[qo[b] : qo[b + 1]]
            vb2[sl] = ve[qo[b] : qo[b + 1]]
        o_c = torch.zeros(T, Hq, Dv, dtype=torch.bfloat16)
        seq = torch.tensor([p + e for p, e in zip(pre, ext)], dtype=torch.int64)
        ops.extend_attention_cpu(q, ke, ve, o_c, kb2, vb2, r2t, torch.arange(B, dtype=torch.int64), seq,
                                 torch.tensor(ext, dtype=torch.int32), qo[:-1].to(torch.int32).contiguous(), max(ext), sm, cap)
        d["expect_cpu_bf16"] = o_c
        save(f"extend_{tag}", d)


def gen_extend_masked(tri):
    """extend_attention_fwd with a custom mask (the draft-tree verification path): bits below or on the diagonal of the
    new tokens chosen at random with the diagonal always set, prefix fully visible (skip_prefix_custom_mask=True) or
    masked at random with its first token kept."""
    torch.manual_seed(7)
    cases = [("tree", 2, 8, 2, 128, 128, [40, 9], [16, 8], True), ("prefix_masked", 2, 4, 4, 64, 64, [17, 5], [12, 6], False)]
    for tag, B, Hq, Hkv, D, Dv, pre, ext, skip in cases:
        n = sum(pre) + sum(ext) + 5
        kb, vb = _mk_pool(n, Hkv, D, Dv, torch.bfloat16)
        T = sum(ext)
        q = torch.randn(T, Hq, D).to(torch.bfloat16)
        ke, ve = torch.randn(T, Hkv, D).to(torch.bfloat16), torch.randn(T, Hkv, Dv).to(torch.bfloat16)
        perm = torch.randperm(n - 1) + 1
        qo = torch.zeros(B + 1, dtype=torch.int32)
        qo[1:] = torch.cumsum(torch.tensor(ext), 0)
        kp = torch.zeros(B + 1, dtype=torch.int32)
        kp[1:] = torch.cumsum(torch.tensor(pre), 0)
        ki = perm[: sum(pre)].to(torch.int32)
        masks, mptr = [], [0]
        for b in range(B):
            m = torch.zeros(ext[b], pre[b] + ext[b], dtype=torch.bool)
            m[:, : pre[b]] = True if skip else (torch.rand(ext[b], pre[b]) < 0.6)
            m[:, 0] = True
            tri_part = torch.tril(torch.rand(ext[b], ext[b]) < 0.5) | torch.eye(ext[b], dtype=torch.bool)
            m[:, pre[b]:] = tri_part
            masks.append(m.flatten())
            mptr.append(mptr[-1] + m.numel())
        mask = torch.cat(masks).to(torch.uint8)
        mask_indptr = torch.tensor(mptr, dtype=torch.int64)
        sm = 1.0 / math.sqrt(D)
        o_t = torch.zeros(T, Hq, Dv)
        tri.extend_attention_fwd(q.float(), ke.float(), ve.float(), o_t, kb.float(), vb.float(), qo, kp, ki,
                                 mask, True, mask_indptr, max(ext), sm, 0.0, skip_prefix_custom_mask=skip)
        save(f"extend_masked_{tag}", dict(
            q_extend=q, k_extend=ke, v_extend=ve, k_buffer=kb, v_buffer=vb, qo_indptr=qo, kv_indptr=kp, kv_indices=ki,
            custom_mask=mask, mask_indptr=mask_indptr, skip_prefix_custom_mask=int(skip), sm_scale=sm,
            expect_triton_f32=o_t))


def gen_extend_window(tri):
    """extend_attention_fwd with sliding_window_size > 0, called as triton_backend.py:759-790 calls it: kv_indices lists
    only the last min(prefix_len, window + 1) prefix tokens of each request (update_sliding_window_buffer, :75-103)."""
    torch.manual_seed(11)
    cases = [("w8", 3, 8, 2, 128, 128, [40, 3, 0], [33, 20, 12], 8), ("w4_h64", 2, 4, 4, 64, 64, [17, 9], [9, 30], 4)]
    for tag, B, Hq, Hkv, D, Dv, pre, ext, W in cases:
        n = sum(pre) + sum(ext) + 5
        kb, vb = _mk_pool(n, Hkv, D, Dv, torch.bfloat16)
        T = sum(ext)
        q = torch.randn(T, Hq, D).to(torch.bfloat16)
        ke, ve = torch.randn(T, Hkv, D).to(torch.bfloat16), torch.randn(T, Hkv, Dv).to(torch.bfloat16)
        perm = torch.randperm(n - 1) + 1
        wpre = [min(p, W + 1) for p in pre]
        qo = torch.zeros(B + 1, dtype=torch.int32)
        qo[1:] = torch.cumsum(torch.tensor(ext), 0)
        kp = torch.zeros(B + 1, dtype=torch.int32)
        kp[1:] = torch.cumsum(torch.tensor(wpre), 0)
        rows, off = [], 0
        for b in range(B):
            full = perm[off: off + pre[b]]
            off += pre[b]
            rows.append(full[pre[b] - wpre[b]:])
        ki = torch.cat(rows).to(torch.int32) if sum(wpre) else torch.zeros(0, dtype=torch.int32)
        sm = 1.0 / math.sqrt(D)
        o_t = torch.zeros(T, Hq, Dv)
        tri.extend_attention_fwd(q.float(), ke.float(), ve.float(), o_t, kb.float(), vb.float(), qo, kp, ki,
                                 None, True, None, max(ext), sm, 0.0, sliding_window_size=W)
        save(f"extend_window_{tag}", dict(
            q_extend=q, k_extend=ke, v_extend=ve, k_buffer=kb, v_buffer=vb, qo_indptr=qo, kv_indptr=kp, kv_indices=ki,
            window=W, sm_scale=sm, expect_triton_f32=o_t))


def gen_merge_state(tri):
    torch.manual_seed(42)
    m = load_test_module("sgl-kernel/tests/test_merge_state_v2.py", "ref_test_merge_state_v2") \
        if os.path.exists(os.path.join(REF, "sgl-kernel/tests/test_merge_state_v2.py")) else None
    for tag, n, h, dd, dtype in (("bf16", 5, 8, 128, torch.bfloat16), ("f16", 3, 4, 64, torch.float16),
                                 ("f32", 2, 2, 256, torch.float32)):
        va, vb = torch.randn(n, h, dd).to(dtype), torch.randn(n, h, dd).to(dtype)
        sa, sb = torch.randn(n, h) * 3, torch.randn(n, h) * 3
        sa[0, 0] = float("inf")  # the inf guard of merge_attn_states.cu:68-69
        sb[1, 1] = float("-inf")
        vo, so = torch.empty_like(va), torch.empty_like(sa)
        tri.merge_state_triton(va, sa, vb, sb, vo, so)
        save(f"merge_state_{tag}", dict(v_a=va, s_a=sa, v_b=vb, s_b=sb, expect_v=vo, expect_s=so))
    del m


def gen_awq():
    torch.manual_seed(42)
    m = load_test_module("sgl-kernel/tests/test_awq_dequant.py", "ref_test_awq")
    for tag, rows, qcols, group, dtype in (("g128_bf16", 256, 16, 128, torch.bfloat16),
                                           ("g128_f16", 256, 24, 128, torch.float16),
                                           ("gall_f16", 128, 16, 128, torch.float16),
                                           ("g64_bf16", 128, 32, 64, torch.bfloat16)):
        qw = torch.randint(0, torch.iinfo(torch.int32).max, (rows, qcols), dtype=torch.int32)
        qw[0, 0] = -1  # all nibbles 0xF incl. the sign bit
        qw[1, 1] = torch.iinfo(torch.int32).min
        sc = torch.rand(rows // group, qcols * 8).to(dtype)
        qz = torch.randint(0, torch.iinfo(torch.int32).max, (rows // group, qcols), dtype=torch.int32)
        qz[0, 0] = -1
        w = m.awq_dequantize_torch(qw, sc, qz, group)
        x = torch.randn(8, rows).to(dtype)
        y = torch.matmul(x.float(), w.float())  # AWQLinearMethod.apply (awq.py:197-215) in fp32
        save(f"awq_{tag}", dict(qweight=qw, scales=sc, qzeros=qz, group_size=group, expect=w.to(dtype), x=x,
                                expect_matmul_f32=y))


def gen_elementwise(ops):
    torch.manual_seed(42)
    mn = load_test_module("sgl-kernel/tests/test_norm.py", "ref_test_norm")
    for tag, rows, hidden, dtype in (("f16", 19, 1024, torch.float16), ("bf16", 5, 4096, torch.bfloat16),
                                     ("odd", 3, 111, torch.float16)):
        x, r = torch.randn(rows, hidden).to(dtype), torch.randn(rows, hidden).to(dtype)
        w = torch.randn(hidden).to(dtype)
        y = mn.llama_rms_norm(x, w, 1e-6)
        y2, r2 = mn.fused_add_rms_norm(x.clone(), r.clone(), w, 1e-6)
        d = dict(x=x, residual=r, weight=w, eps=1e-6, expect_rmsnorm=y, expect_fused_x=y2, expect_fused_residual=r2)
        if dtype == torch.bfloat16:
            d["expect_rmsnorm_cpu"] = ops.rmsnorm_cpu(x, w, 1e-6)
        save(f"norm_{tag}", d)
    for tag, rows, dd, dtype in (("f16", 6, 256, torch.float16), ("bf16", 4, 14336, torch.bfloat16)):
        x = torch.randn(rows, 2 * dd).to(dtype)
        y = x[..., dd:] * torch.nn.functional.silu(x[..., :dd])  # test_activation.py:14
        d = dict(x=x, expect=y)
        if dtype == torch.bfloat16:
            d["expect_cpu"] = ops.silu_and_mul_cpu(x)
        save(f"silu_{tag}", d)
    mr = load_test_module("sgl-kernel/tests/test_rotary_embedding.py", "ref_test_rope")
    for tag, hs, rot, maxpos, neox, nq, nk, toks in (("neox128", 128, 128, 2048, True, 32, 8, 11),
                                                     ("gptj128", 128, 128, 2048, False, 16, 4, 7),
                                                     ("partial", 256, 128, 311, True, 4, 2, 9)):
        rope = mr.RotaryEmbedding(hs, rot, maxpos, 10000, neox, torch.bfloat16)
        pos = torch.randint(0, maxpos, (toks,), dtype=torch.int64)
        q, k = torch.randn(toks, nq * hs).to(torch.bfloat16), torch.randn(toks, nk * hs).to(torch.bfloat16)
        qo, ko = rope.forward_native(pos, q.clone(), k.clone())
        save(f"rope_{tag}", dict(q=q, k=k, positions=pos, cos_sin_cache=rope.cos_sin_cache.float(), head_size=hs,
                                 is_neox=int(neox), expect_q=qo, expect_k=ko))


def gen_fp8():
    torch.manual_seed(42)
    mg = load_test_module("sgl-kernel/tests/test_fp8_gemm.py", "ref_test_fp8_gemm")
    fmax = torch.finfo(torch.float8_e4m3fn).max
    for tag, M, N, K, bias, dt in (("bf16", 8, 48, 512, True, torch.bfloat16), ("f16", 17, 16, 1024, False, torch.float16)):
        a = ((torch.rand(M, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn)
        b = ((torch.rand(N, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn).t()
        sa, sb = torch.randn(M) * 0.001, torch.randn(N) * 0.001
        bi = torch.randn(N).to(dt) if bias else None
        o = mg.torch_scaled_mm(a, b, sa, sb, dt, bi)
        d = dict(a=a, b_nk=b.t().contiguous(), scale_a=sa, scale_b=sb, expect=o)
        if bi is not None:
            d["bias"] = bi
        save(f"fp8_mm_{tag}", d)
    mb = load_test_module("sgl-kernel/tests/test_fp8_blockwise_gemm.py", "ref_test_fp8_block")
    M, N, K = 6, 256, 384
    a = ((torch.rand(M, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn)
    b = ((torch.rand(N, K) - 0.5) * 2 * fmax).clamp(-fmax, fmax).to(torch.float8_e4m3fn).t()
    sa = torch.randn(M, K // 128) * 0.001
    sb = torch.randn(K // 128, N // 128) * 0.001
    o = mb.baseline_scaled_mm(a, b, sa, sb, torch.bfloat16)
    save("fp8_block_mm", dict(a=a, b_nk=b.t().contiguous(), scale_a=sa, scale_b_kn=sb, expect=o))
    mq = load_test_module("sgl-kernel/tests/test_per_token_quant_fp8.py", "ref_test_ptq")
    x = torch.rand(9, 512).to(torch.float16)
    x[0] *= 37.0
    x[1, :] = 0.25
    amax = x.float().abs().amax(dim=1)
    scale = amax / fmax  # per_token_quant_fp8.cu:49
    q = mq.torch_per_token_quant_fp8(x, scale)
    save("fp8_per_token_quant", dict(x=x, expect_scale=scale, expect_q=q))


# ------------------------------------------------------------------------------------------------
def _tree_dump(rc_ns, tree):
    """Canonical structural dump of a RadixCache: pre-order over children sorted by first token."""
    out = []

    def rec(node, depth):
        for ck in sorted(node.children.keys(), key=lambda k: (str(type(k)), k)):
            c = node.children[ck]
            val = c.value
            out.append(dict(depth=depth, key=[int(t) for t in c.key], value=[int(v) for v in val],
                            lock_ref=int(c.lock_ref), private=bool(c.private),
                            need_check=bool(c.need_check_privacy), owner=None if c.owner_id is None else str(c.owner_id),
                            epoch=int(c.epoch), hit_count=int(c.hit_count), hit_pre=int(c.hit_pre),
                            u_cnt_pre=int(c.u_cnt_pre), n_users=len(c.u_cnt_l)))
            rec(c, depth + 1)

    rec(tree.root_node, 0)
    return out


def gen_radix():
    """Scripted traces through the reference RadixCache; each step records the op, its arguments and
    every observable result (matched slots, returned prefix length, sizes, frees, tree dump)."""
    rc = RL.load_radix_cache()
    rng = random.Random(0)
    traces = {}

    def run(name, script, decide=None):
        rc.set_epoch(0)
        alloc = RL.FakeAllocator()
        judge = RL.ScriptedJudgeClient(decide)
        pool = RL.FakeReqPool()
        tree = rc.RadixCache(pool, alloc, page_size=1, private_judge_client=judge)
        steps = []
        handles = {}
        reqs = {}
        next_slot = [100000]  # slots handed to requests (the scripted inserts use small numbers)

        def take(n):
            s0 = next_slot[0]
            next_slot[0] += n
            return torch.arange(s0, s0 + n, dtype=torch.int64)
        for op in script:
            rec = dict(op)
            kind = op["op"]
            nfreed = len(alloc.freed)
            if kind == "epoch":
                rc.set_epoch(op["value"])
            elif kind == "insert":
                val = torch.tensor(op["value"], dtype=torch.int64)
                rec["ret"] = int(tree.insert(list(op["key"]), val, op.get("prompt", ""), user_id=op.get("user")))
            elif kind == "match":
                if "user" in op:
                    v, node = tree.match_prefix(list(op["key"]), user_id=op["user"])
                else:
                    v, node = tree.match_prefix(list(op["key"]))
                rec["ret"] = [int(t) for t in v]
                rec["last_node_key"] = [int(t) for t in (node.key or [])]
                if "handle" in op:
                    handles[op["handle"]] = node
            elif kind == "lock":
                rec["ret"] = int(tree.inc_lock_ref(handles[op["handle"]]))
            elif kind == "unlock":
                rec["ret"] = int(tree.dec_lock_ref(handles[op["handle"]]))
            elif kind == "evict":
                tree.evict(op["n"])
            elif kind in ("req_start", "req_chunk"):
                # what Req.init_next_round_input + ScheduleBatch.prepare_for_extend do around the tree
                # (schedule_batch.py:630-656,1275): match on all but the last token to fill, lock the node, give
                # the rest fresh slots and write the request's row
                if kind == "req_start":
                    r = RL.FakeReq(op["rid"], op["tokens"], op.get("user"))
                    r.req_pool_idx = op["row"]
                    reqs[op["rid"]] = r
                    r.fill_ids = r.origin_input_ids[: op.get("fill", len(r.origin_input_ids))]
                    v, node = tree.match_prefix(rid=r.rid, user_id=r.user_id, key=r.fill_ids[:-1])
                    r.prefix_indices, r.last_node = v, node
                    tree.inc_lock_ref(node)
                    pool.write((r.req_pool_idx, slice(0, len(v))), v.to(torch.int32))
                else:
                    r = reqs[op["rid"]]
                    r.fill_ids = r.origin_input_ids[: op["fill"]]
                have = len(r.prefix_indices)
                new = take(len(r.fill_ids) - have)
                pool.write((r.req_pool_idx, slice(have, len(r.fill_ids))), new.to(torch.int32))
                rec["prefix"] = [int(t) for t in r.prefix_indices]
            elif kind == "req_unfinished":
                r = reqs[op["rid"]]
                tree.cache_unfinished_req(r)
                rec["prefix"] = [int(t) for t in r.prefix_indices]
                rec["last_node_key"] = [int(t) for t in (r.last_node.key or [])]
                rec["row"] = [int(t) for t in pool.req_to_token[r.req_pool_idx, : len(r.fill_ids)]]
            elif kind == "req_finished":
                r = reqs[op["rid"]]
                r.output_ids = list(op["output"])
                n_all = len(r.origin_input_ids) + len(r.output_ids) - 1
                have = len(r.fill_ids)
                if n_all > have:  # decode steps: one slot per generated token
                    pool.write((r.req_pool_idx, slice(have, n_all)), take(n_all - have).to(torch.int32))
                npool = len(pool.freed)
                tree.cache_finished_req(r)
                rec["pool_freed"] = pool.freed[npool:]
            elif kind == "set_public":
                # what PrivateJudgeClient._response_task does (private_client.py:125-140)
                for n in judge.nodes:
                    if n.need_check_privacy and len(n.key) >= op.get("min_len", 0):
                        n.private = False
                        n.need_check_privacy = False
            rec["freed"] = [[int(t) for t in f] for f in alloc.freed[nfreed:]]
            rec["evictable"] = int(tree.evictable_size())
            rec["protected"] = int(tree.protected_size())
            rec["total"] = int(tree.total_size())
            rec["tree"] = _tree_dump(rc, tree)
            steps.append(rec)
        traces[name] = steps

    slot = [1]

    def slots(n):
        s = list(range(slot[0], slot[0] + n))
        slot[0] += n
        return s

    def toks(n):
        return [rng.randrange(1000, 32000) for _ in range(n)]

    # 1. fork defaults: every node private, root owner 0 => user "0" can build and share, others break at root
    a, b = toks(40), toks(25)
    run("user0_basic", [
        dict(op="insert", key=a, value=slots(40), user="0"),
        dict(op="match", key=a[:30] + toks(5), user="0", handle="h1"),
        dict(op="lock", handle="h1"),
        dict(op="insert", key=a[:30] + b, value=slots(55), user="0"),
        dict(op="match", key=a, user="0"),
        dict(op="match", key=a, user="1"),
        dict(op="match", key=a),
        dict(op="evict", n=10),
        dict(op="unlock", handle="h1"),
        dict(op="evict", n=1000),
    ])
    # 2. the root-private quirk (radix_cache.py:510-514,556): a second user's insert overwrites the child
    slot[0] = 1
    c = toks(30)
    run("root_overwrite", [
        dict(op="insert", key=c, value=slots(30), user="alice"),
        dict(op="insert", key=c, value=slots(30), user="alice"),
        dict(op="insert", key=c[:10] + toks(10), value=slots(20), user="bob"),
        dict(op="match", key=c, user="alice"),
        dict(op="match", key=c, user="bob"),
    ])
    # 3. fuzzy key match (radix_cache.py:52-62): up to 11 mismatches are tolerated
    slot[0] = 1
    d = toks(64)
    d_fuzzy = list(d)
    for i in (3, 9, 10, 20, 33):
        d_fuzzy[i] += 1
    d_far = list(d)
    for i in range(5, 30):
        d_far[i] += 1
    run("fuzzy_match", [
        dict(op="insert", key=d, value=slots(64), user="0"),
        dict(op="match", key=d_fuzzy, user="0"),
        dict(op="match", key=d_far, user="0"),
        dict(op="insert", key=d_far, value=slots(64), user="0"),
        dict(op="match", key=d_far, user="0"),
    ])
    # 4. 1024-token node chunking (radix_cache.py:536-541) + public promotion + cross-user sharing
    slot[0] = 1
    e = toks(2500)
    run("chunk_public", [
        dict(op="insert", key=e, value=slots(2500), user="0", prompt="p"),
        dict(op="set_public"),
        dict(op="match", key=e[:2000] + toks(8), user="7", handle="h"),
        dict(op="lock", handle="h"),
        dict(op="match", key=e, user="0"),
        dict(op="evict", n=600),
        dict(op="unlock", handle="h"),
        dict(op="evict", n=5000),
    ])
    # 5. epoch windows: hit counters roll over (radix_cache.py:377-387), eviction ordered by epoch
    slot[0] = 1
    f1, f2, f3 = toks(20), toks(20), toks(20)
    run("epoch_lru", [
        dict(op="insert", key=f1, value=slots(20), user="0"),
        dict(op="epoch", value=1),
        dict(op="insert", key=f2, value=slots(20), user="0"),
        dict(op="epoch", value=2),
        dict(op="insert", key=f3, value=slots(20), user="0"),
        dict(op="match", key=f1, user="0"),
        dict(op="match", key=f1, user="0"),
        dict(op="epoch", value=3),
        dict(op="match", key=f1, user="0"),
        dict(op="match", key=f2, user="0"),
        dict(op="evict", n=20),
        dict(op="evict", n=20),
    ])
    # 6. random workload, single user "0": many inserts/matches/evicts with shared prefixes
    slot[0] = 1
    bases = [toks(rng.randrange(10, 60)) for _ in range(6)]
    script = []
    ep = 0
    for i in range(120):
        r = rng.random()
        base = rng.choice(bases)
        key = base[: rng.randrange(1, len(base) + 1)] + toks(rng.randrange(0, 25))
        if r < 0.45:
            script.append(dict(op="insert", key=key, value=slots(len(key)), user="0"))
        elif r < 0.85:
            script.append(dict(op="match", key=key, user="0"))
        elif r < 0.93:
            script.append(dict(op="evict", n=rng.randrange(1, 80)))
        else:
            ep += 1
            script.append(dict(op="epoch", value=ep))
    run("random_user0", script)
    # 7. hit-rate anomaly release (radix_cache.py:385-411): a burst of hits in one window frees the subtree,
    #    skipping locked nodes
    slot[0] = 1
    g, g2, g3 = toks(30), toks(12), toks(9)
    burst = [dict(op="match", key=g + g2, user="0") for _ in range(12)]
    run("entropy_release", [
        dict(op="insert", key=g + g2, value=slots(42), user="0"),
        dict(op="insert", key=g + g3, value=slots(39), user="0"),
        dict(op="match", key=g + g2, user="0"),
        dict(op="epoch", value=1),
        dict(op="match", key=g + g2, user="0"),
    ] + burst + [
        dict(op="match", key=g + g3, user="0"),
        dict(op="insert", key=g + g2, value=slots(42), user="0"),
        dict(op="evict", n=1000),
    ])
    slot[0] = 1
    run("entropy_release_locked", [
        dict(op="insert", key=g + g2, value=slots(42), user="0"),
        dict(op="insert", key=g + g3, value=slots(39), user="0"),
        dict(op="match", key=g + g3, user="0", handle="h"),
        dict(op="match", key=g + g2, user="0"),
        dict(op="lock", handle="h"),
        dict(op="epoch", value=1),
        dict(op="match", key=g + g2, user="0"),
    ] + burst + [
        dict(op="unlock", handle="h"),
        dict(op="match", key=g + g3, user="0"),
        dict(op="evict", n=1000),
    ])
    # 8. request life cycle (radix_cache.py:160-240), user "0": a chunked prefill caches its first chunk
    #    (cache_unfinished_req: insert, re-match WITHOUT a user id, row rewritten with the tree's slots), two
    #    requests with the same prompt run side by side and the second to finish frees its duplicate slots
    slot[0] = 1
    p1, p2 = toks(48), toks(20)
    run("req_cycle_user0", [
        dict(op="req_start", rid="a", row=1, tokens=p1 + p2 + [7], user="0", fill=40),
        dict(op="req_unfinished", rid="a"),
        dict(op="req_chunk", rid="a", fill=69),
        dict(op="req_start", rid="b", row=2, tokens=p1 + p2 + [7], user="0"),
        dict(op="req_start", rid="c", row=3, tokens=p1 + toks(9), user="0"),
        dict(op="req_unfinished", rid="a"),
        dict(op="req_finished", rid="a", output=[11, 12, 13]),
        dict(op="req_finished", rid="b", output=[11, 12, 14, 15]),
        dict(op="evict", n=10),
        dict(op="req_finished", rid="c", output=[5]),
        dict(op="match", key=p1 + p2 + [7, 11, 12], user="0"),
        dict(op="evict", n=1000),
    ])
    # 9. the same with users other than "0": every insert stops at the private root and overwrites the child
    #    (radix_cache.py:510-514,556); the user-less re-match of cache_unfinished_req still walks the new branch
    slot[0] = 1
    q1, q2 = toks(36), toks(12)
    run("req_cycle_users", [
        dict(op="req_start", rid="a", row=1, tokens=q1 + q2, user="alice", fill=30),
        dict(op="req_unfinished", rid="a"),
        dict(op="req_chunk", rid="a", fill=48),
        dict(op="req_start", rid="b", row=2, tokens=q1 + toks(5), user="bob"),
        dict(op="req_finished", rid="a", output=[3, 4]),
        dict(op="req_start", rid="c", row=3, tokens=q1 + q2 + [3, 4], user="alice"),
        dict(op="req_finished", rid="b", output=[9]),
        dict(op="req_finished", rid="c", output=[8, 8]),
        dict(op="match", key=q1 + q2, user="alice"),
        dict(op="match", key=q1 + q2, user="bob"),
        dict(op="evict", n=1000),
    ])
    os.makedirs(OUT, exist_ok=True)
    with open(os.path.join(OUT, "radix_traces.json"), "w") as f:
        json.dump(traces, f, separators=(",", ":"))
    print("wrote radix_traces", {k: len(v) for k, v in traces.items()})


def gen_num_kv_splits():
    """get_num_kv_splits_triton (triton_backend.py:23-72) under the interpreter."""
    os.environ["TRITON_INTERPRET"] = "1"
    RL.load_triton_ops()
    RL._stub("sglang.srt.layers.attention.base_attn_backend", AttentionBackend=object)
    RL._stub("sglang.srt.layers.dp_attention", get_attention_tp_size=lambda: 1)
    RL._stub("sglang.srt.layers.radix_attention", AttentionType=object)
    RL._stub("sglang.srt.model_executor")
    RL._stub("sglang.srt.model_executor.forward_batch_info", ForwardBatch=object, ForwardMode=object)
    RL._stub("sglang.srt.utils", is_hip=lambda: False, is_cuda=lambda: False,
             get_bool_env_var=lambda *a, **k: False, get_device_core_count=lambda *a: 256)
    tb = RL._load("sglang.srt.layers.attention.triton_backend", "python/sglang/srt/layers/attention/triton_backend.py")
    torch.manual_seed(42)
    cases = []
    for num_seq, hq, hkv, maxs, cores, lo, hi in ((64, 32, 8, 8, 256, 1152, 1153), (64, 32, 8, 16, 256, 1152, 1153),
                                                  (5, 32, 32, 8, 256, 10, 5000), (1, 16, 1, 16, 256, 70000, 70001),
                                                  (37, 14, 2, 8, 80, 1, 300), (3, 128, 1, 16, 304, 4000, 9000)):
        sl = torch.randint(lo, hi, (num_seq,), dtype=torch.int64)
        out = torch.zeros(num_seq, dtype=torch.int32)
        sched = 256 if num_seq < 256 else 1 << (num_seq - 1).bit_length()
        tb.get_num_kv_splits_triton[(1,)](out, sl, num_seq, 1, hq, hkv, maxs, cores, MAX_NUM_SEQ=sched)
        cases.append(dict(seq_lens=sl.tolist(), num_head=hq, num_kv_head=hkv, max_kv_splits=maxs,
                          device_core_count=cores, expect=out.tolist()))
    with open(os.path.join(OUT, "num_kv_splits.json"), "w") as f:
        json.dump(cases, f)
    print("wrote num_kv_splits")


def main():
    assert RL.have_reference(), "reference tree not present"
    which = set(sys.argv[1:])

    def want(n):
        return not which or n in which

    tri = RL.load_triton_ops()
    ops = RL.load_ref_cpu_ops()
    if want("kv"):
        gen_kv_indices(tri)
    if want("decode"):
        gen_decode(tri, ops)
    if want("extend"):
        gen_extend(tri, ops)
    if want("extend_masked"):
        gen_extend_masked(tri)
    if want("extend_window"):
        gen_extend_window(tri)
    if want("merge"):
        gen_merge_state(tri)
    if want("awq"):
        gen_awq()
    if want("elem"):
        gen_elementwise(ops)
    if want("fp8"):
        gen_fp8()
    if want("radix"):
        gen_radix()
    if want("splits"):
        gen_num_kv_splits()


if __name__ == "__main__":
    main()
