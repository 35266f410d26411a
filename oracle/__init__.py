"""TEST INFRASTRUCTURE ONLY: ctypes front-end to the C oracle (oracle/oracle.c).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this package.
The product package never does: it fails loudly when the HIP library is missing instead.

All functions take and return CPU torch tensors; half-precision inputs are widened to fp32
exactly, and outputs are rounded to the activation dtype where the reference rounds.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np
import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_SRC_PATH = os.path.join(_HERE, "oracle.c")


def build(force: bool = False) -> str:
    """Compile oracle.c with gcc (seconds)."""
    if force or not os.path.exists(_LIB_PATH) or os.path.getmtime(_LIB_PATH) < os.path.getmtime(_SRC_PATH):
        subprocess.check_call(
            ["gcc", "-O2", "-fPIC", "-shared", "-fopenmp", _SRC_PATH, "-o", _LIB_PATH, "-lm"]
        )
    return _LIB_PATH


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_LIB_PATH):
            build()
        _lib = C.CDLL(_LIB_PATH)
    return _lib


_DT = {torch.float32: 0, torch.float16: 1, torch.bfloat16: 2}


def _f32(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().cpu().to(torch.float32).numpy())


def _i32(t) -> np.ndarray:
    return np.ascontiguousarray(torch.as_tensor(t).detach().cpu().to(torch.int32).numpy())


def _i64(t) -> np.ndarray:
    return np.ascontiguousarray(torch.as_tensor(t).detach().cpu().to(torch.int64).numpy())


def _p(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _raw16(t: torch.Tensor) -> np.ndarray:
    return np.ascontiguousarray(t.detach().cpu().contiguous().view(torch.int16).numpy()).view(np.uint16)


def create_kv_indices(req_to_token, req_pool_indices, lens, kv_indptr, kv_start=None):
    r2t = _i32(req_to_token)
    idx, ln, ptr = _i64(req_pool_indices), _i64(lens), _i32(kv_indptr)
    st = None if kv_start is None else _i64(kv_start)
    out = np.zeros(int(ptr[len(ln)]), dtype=np.int32)
    lib().oracle_create_kv_indices(_p(r2t), C.c_int64(r2t.shape[1]), _p(idx), _p(ln), _p(ptr), _p(st), _p(out), C.c_int(len(ln)))
    return torch.from_numpy(out)


def get_num_kv_splits(seq_lens, num_group, num_head, num_kv_head, max_kv_splits, device_core_count):
    sl = _i64(seq_lens)
    out = np.zeros(len(sl) * num_group, dtype=np.int32)
    lib().oracle_get_num_kv_splits(_p(out), _p(sl), C.c_int(len(sl)), C.c_int(num_group), C.c_int(num_head),
                                   C.c_int(num_kv_head), C.c_int(max_kv_splits), C.c_int(device_core_count))
    return torch.from_numpy(out)


def decode_attention(q, k_buffer, v_buffer, kv_indptr, kv_indices, num_kv_splits, sm_scale, logit_cap=0.0,
                     return_lse=False):
    """fp32 result [B,Hq,Dv] (not rounded to the activation type)."""
    B, Hq, D = q.shape
    Hkv, Dv = v_buffer.shape[1], v_buffer.shape[2]
    qf, kf, vf = _f32(q), _f32(k_buffer), _f32(v_buffer)
    ptr, ind = _i32(kv_indptr), _i32(kv_indices)
    ns = None if num_kv_splits is None else _i32(num_kv_splits)
    o = np.zeros((B, Hq, Dv), dtype=np.float32)
    lse = np.zeros((B, Hq), dtype=np.float32)
    lib().oracle_decode_attention(_p(qf), _p(kf), _p(vf), _p(o), _p(lse), _p(ptr), _p(ind), _p(ns),
                                  C.c_int(B), C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(Dv),
                                  C.c_float(sm_scale), C.c_float(logit_cap))
    if return_lse:
        return torch.from_numpy(o), torch.from_numpy(lse)
    return torch.from_numpy(o)


def extend_attention(q_ext, k_ext, v_ext, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, sm_scale,
                     logit_cap=0.0, causal=True):
    T, Hq, D = q_ext.shape
    Hkv, Dv = v_ext.shape[1], v_ext.shape[2]
    o = np.zeros((T, Hq, Dv), dtype=np.float32)
    a = [_f32(q_ext), _f32(k_ext), _f32(v_ext)]
    kb, vb = _f32(k_buffer), _f32(v_buffer)
    qo, kp, ki = _i32(qo_indptr), _i32(kv_indptr), _i32(kv_indices)
    lib().oracle_extend_attention(_p(a[0]), _p(a[1]), _p(a[2]), _p(o), _p(kb), _p(vb), _p(qo), _p(kp), _p(ki),
                                  C.c_int(len(qo) - 1), C.c_int(Hq), C.c_int(Hkv), C.c_int(D), C.c_int(Dv),
                                  C.c_float(sm_scale), C.c_float(logit_cap), C.c_int(1 if causal else 0))
    return torch.from_numpy(o)


def merge_state(v_a, s_a, v_b, s_b):
    n, h, d = v_a.shape
    va, vb, sa, sb = _f32(v_a), _f32(v_b), _f32(s_a), _f32(s_b)
    vo = np.zeros_like(va)
    so = np.zeros_like(sa)
    lib().oracle_merge_state(_p(va), _p(sa), _p(vb), _p(sb), _p(vo), _p(so), C.c_int64(n * h), C.c_int(d))
    return torch.from_numpy(vo), torch.from_numpy(so)


def awq_dequantize(qweight, scales, qzeros):
    rows, qcols = qweight.shape
    group = rows // scales.shape[0]
    qw, qz = _i32(qweight), _i32(qzeros)
    sc = _raw16(scales)
    out = np.zeros((rows, qcols * 8), dtype=np.uint16)
    lib().oracle_awq_dequantize(_p(qw), _p(sc), _p(qz), _p(out), C.c_int(rows), C.c_int(qcols), C.c_int(group),
                                C.c_int(_DT[scales.dtype]))
    return torch.from_numpy(out.view(np.int16)).view(scales.dtype)


def half_matmul(x, w, bias=None):
    """x:[M,K] @ w:[K,N] (half/bf16), fp32 accumulate, rounded once to x.dtype."""
    M, K = x.shape
    N = w.shape[1]
    xf = _f32(x)
    wr = _raw16(w.to(x.dtype))
    bf = None if bias is None else _f32(bias)
    out = np.zeros((M, N), dtype=np.float32)
    lib().oracle_half_matmul(_p(xf), _p(wr), _p(bf), _p(out), C.c_int(M), C.c_int(K), C.c_int(N), C.c_int(_DT[x.dtype]))
    return torch.from_numpy(out).to(x.dtype)


def awq_gemm(x, qweight, scales, qzeros, bias=None):
    return half_matmul(x, awq_dequantize(qweight, scales, qzeros), bias)


def rmsnorm(x, weight, eps):
    xf, wf = _f32(x), _f32(weight)
    out = np.zeros_like(xf)
    lib().oracle_rmsnorm(_p(xf), _p(wf), _p(out), C.c_int64(xf.shape[0]), C.c_int(xf.shape[1]), C.c_float(eps),
                         C.c_int(_DT[x.dtype]))
    return torch.from_numpy(out).to(x.dtype)


def fused_add_rmsnorm(x, residual, weight, eps):
    """Returns (normed, new_residual); inputs untouched."""
    xf, rf, wf = _f32(x).copy(), _f32(residual).copy(), _f32(weight)
    lib().oracle_fused_add_rmsnorm(_p(xf), _p(rf), _p(wf), C.c_int64(xf.shape[0]), C.c_int(xf.shape[1]),
                                   C.c_float(eps), C.c_int(_DT[x.dtype]))
    return torch.from_numpy(xf).to(x.dtype), torch.from_numpy(rf).to(x.dtype)


def silu_and_mul(x):
    d = x.shape[-1] // 2
    xf = _f32(x.reshape(-1, 2 * d))
    out = np.zeros((xf.shape[0], d), dtype=np.float32)
    lib().oracle_silu_and_mul(_p(xf), _p(out), C.c_int64(xf.shape[0]), C.c_int(d), C.c_int(_DT[x.dtype]))
    return torch.from_numpy(out).to(x.dtype).reshape(*x.shape[:-1], d)


def rope(x, positions, cos_sin_cache, head_size, is_neox=True):
    """x: [tokens, heads*head_size]; returns rotated copy."""
    tokens = x.shape[0]
    heads = x.shape[1] // head_size
    xf = _f32(x).copy()
    pos, cs = _i64(positions), _f32(cos_sin_cache)
    lib().oracle_rope(_p(xf), _p(pos), _p(cs), C.c_int64(tokens), C.c_int(heads), C.c_int(head_size),
                      C.c_int(cs.shape[1]), C.c_int(0 if is_neox else 1), C.c_int(_DT[x.dtype]))
    return torch.from_numpy(xf).to(x.dtype)


def e4m3_decode(q: torch.Tensor) -> torch.Tensor:
    raw = np.ascontiguousarray(q.detach().cpu().contiguous().view(torch.uint8).numpy())
    out = np.zeros(raw.shape, dtype=np.float32)
    lib().oracle_e4m3_decode(_p(raw), _p(out), C.c_int64(raw.size))
    return torch.from_numpy(out)


def e4m3_encode(x: torch.Tensor) -> torch.Tensor:
    xf = _f32(x)
    out = np.zeros(xf.shape, dtype=np.uint8)
    lib().oracle_e4m3_encode(_p(xf), _p(out), C.c_int64(xf.size))
    return torch.from_numpy(out).view(torch.float8_e4m3fn)


def per_token_quant_fp8(x, fp8_max=448.0):
    xf = _f32(x)
    q = np.zeros(xf.shape, dtype=np.uint8)
    s = np.zeros((xf.shape[0],), dtype=np.float32)
    lib().oracle_per_token_quant_fp8(_p(xf), _p(q), _p(s), C.c_int64(xf.shape[0]), C.c_int(xf.shape[1]), C.c_float(fp8_max))
    return torch.from_numpy(q).view(torch.float8_e4m3fn), torch.from_numpy(s).reshape(-1, 1)


def per_token_group_quant_fp8(x, group_size, eps=1e-10, fp8_min=-448.0, fp8_max=448.0):
    xf = _f32(x)
    q = np.zeros(xf.shape, dtype=np.uint8)
    s = np.zeros((xf.shape[0], xf.shape[1] // group_size), dtype=np.float32)
    lib().oracle_per_token_group_quant_fp8(_p(xf), _p(q), _p(s), C.c_int64(xf.shape[0]), C.c_int(xf.shape[1]),
                                           C.c_int(group_size), C.c_float(eps), C.c_float(fp8_min), C.c_float(fp8_max))
    return torch.from_numpy(q).view(torch.float8_e4m3fn), torch.from_numpy(s)


def fp8_scaled_mm(a, b, scale_a, scale_b, out_dtype, bias=None):
    """a:[M,K] e4m3; b:[K,N] e4m3 column-major (i.e. b.t() contiguous), as the reference takes it."""
    M, K = a.shape
    N = b.shape[1]
    ar = np.ascontiguousarray(a.detach().cpu().contiguous().view(torch.uint8).numpy())
    br = np.ascontiguousarray(b.detach().cpu().t().contiguous().view(torch.uint8).numpy())
    sa, sb = _f32(scale_a.reshape(-1)), _f32(scale_b.reshape(-1))
    bf = None if bias is None else _f32(bias)
    out = np.zeros((M, N), dtype=np.float32)
    lib().oracle_fp8_scaled_mm(_p(ar), _p(br), _p(sa), _p(sb), _p(bf), _p(out), C.c_int(M), C.c_int(N), C.c_int(K),
                               C.c_int(_DT[out_dtype]))
    return torch.from_numpy(out).to(out_dtype)


def fp8_blockwise_mm(a, b_nk, a_scales, b_scales, out_dtype, block_n=128, block_k=128):
    """a:[M,K], a_scales:[M,K/bk]; b_nk:[N,K], b_scales:[N/bn,K/bk]."""
    M, K = a.shape
    N = b_nk.shape[0]
    ar = np.ascontiguousarray(a.detach().cpu().contiguous().view(torch.uint8).numpy())
    br = np.ascontiguousarray(b_nk.detach().cpu().contiguous().view(torch.uint8).numpy())
    sa, sb = _f32(a_scales), _f32(b_scales)
    out = np.zeros((M, N), dtype=np.float32)
    lib().oracle_fp8_blockwise_mm(_p(ar), _p(br), _p(sa), _p(sb), _p(out), C.c_int(M), C.c_int(N), C.c_int(K),
                                  C.c_int(block_n), C.c_int(block_k), C.c_int(_DT[out_dtype]))
    return torch.from_numpy(out).to(out_dtype)


def allreduce_sum(bufs, dtype):
    arrs = [_f32(b) for b in bufs]
    n = arrs[0].size
    ptrs = (C.c_void_p * len(arrs))(*[a.ctypes.data for a in arrs])
    out = np.zeros(arrs[0].shape, dtype=np.float32)
    lib().oracle_allreduce_sum(ptrs, C.c_int(len(arrs)), _p(out), C.c_int64(n), C.c_int(_DT[dtype]))
    return torch.from_numpy(out).to(dtype)


# ---------------------------------------------------------------------------------------------- MoE routing, bmm_fp8
def moe_align_block_size(topk_ids, num_experts, block_size):
    """Returns (sorted_token_ids padded with numel, expert_ids, num_tokens_post_pad) as int32 tensors."""
    ids = _i64(topk_ids).reshape(-1)
    numel = ids.shape[0]
    cap = numel + num_experts * (block_size - 1)
    srt = np.full(cap, numel, dtype=np.int32)
    ex = np.full((cap + block_size - 1) // block_size, -1, dtype=np.int32)
    tot = np.zeros(1, dtype=np.int32)
    lib().oracle_moe_align_block_size(_p(ids), C.c_int64(numel), C.c_int(num_experts), C.c_int(block_size), _p(srt),
                                      _p(ex), _p(tot))
    return torch.from_numpy(srt), torch.from_numpy(ex), int(tot[0])


def topk_softmax(gating_output, topk):
    g = _f32(gating_output)
    T, E = g.shape
    w = np.zeros((T, topk), dtype=np.float32)
    ids = np.zeros((T, topk), dtype=np.int32)
    src = np.zeros((T, topk), dtype=np.int32)
    lib().oracle_topk_softmax(_p(g), _p(w), _p(ids), _p(src), C.c_int(T), C.c_int(E), C.c_int(topk))
    return torch.from_numpy(w), torch.from_numpy(ids), torch.from_numpy(src)


def bmm_fp8(A, B, A_scale, B_scale, dtype):
    """out[b] = (A[b] @ B[b]) * A_scale * B_scale with the fp8 operands widened exactly and fp32 accumulation
    (the arithmetic of bmm_fp8, sgl-kernel/csrc/gemm/bmm_fp8.cu:23-76, up to summation order); rounded to dtype."""
    a = A.detach().cpu().to(torch.float32)
    b = B.detach().cpu().to(torch.float32)
    out = torch.bmm(a, b) * float(A_scale) * float(B_scale)
    return out.to(dtype)


# ---------------------------------------------------------------------------------------------- sampling filters
def top_k_mask(probs, k):
    """Entries >= the k-th largest of their row (sgl-kernel/tests/test_sampling.py:85-88)."""
    p = probs.detach().cpu().double()
    pivot = torch.sort(p, descending=True).values[:, k - 1]
    return p >= pivot.unsqueeze(-1)


def top_p_mask(probs, top_p, eps=0.0):
    """Entries whose ascending cumulative sum reaches 1 - top_p (test_sampling.py:55-60; eps as in :23-27)."""
    p = probs.detach().cpu().double()
    sp, idx = torch.sort(p, descending=False)
    cdf = torch.cumsum(sp, dim=-1)
    mask = torch.zeros_like(p, dtype=torch.bool)
    mask.scatter_(1, idx, cdf >= (1 - top_p) - eps)
    return mask


def min_p_mask(probs, min_p):
    """Entries >= min_p * the row maximum (test_sampling.py:108-116)."""
    p = probs.detach().cpu().float()
    return p >= (min_p * p.max(dim=-1, keepdim=True).values)


def renorm(probs, mask):
    p = probs.detach().cpu().double() * mask
    return (p / p.sum(dim=-1, keepdim=True)).float()


def sample_from_mask(probs, mask, uniform):
    """First kept index (in index order) whose running sum exceeds u * total, in float64."""
    p = probs.detach().cpu().double() * mask
    cum = torch.cumsum(p, dim=-1)
    target = uniform.detach().cpu().double().unsqueeze(-1) * cum[:, -1:]
    return (cum > target).float().argmax(dim=-1).int()


# ---------------------------------------------------------------------------------------------- draft-tree helpers
def build_tree_preprocess(verified_id, score_list, token_list, parents_list, num_verify_tokens):
    """Selection of the draft tokens from the per-step scores (speculative/build_eagle_tree.py:15-40)."""
    scores = torch.cat(score_list, dim=1).flatten(1)
    tokens = torch.cat(token_list, dim=1)
    top = torch.sort(torch.topk(scores, num_verify_tokens - 1, dim=-1).indices).values
    draft = torch.cat((verified_id.unsqueeze(1), torch.gather(tokens, index=top, dim=1)), dim=1).flatten()
    if len(parents_list) > 1:
        parent_list = torch.cat(parents_list[:-1], dim=1)
    else:
        parent_list = torch.empty(parents_list[0].shape[0], 0, dtype=torch.int64)
    return parent_list, top, draft


def build_tree_efficient(parent_list, selected_index, verified_seq_len, topk, depth, n):
    """Serial restatement of build_tree_efficient (sgl-kernel/csrc/speculative/eagle_utils.cu:32-114) on lists.
    Returns tree_mask (flat list of bool, initialised True as the caller does, build_eagle_tree.py:70-77), positions,
    retrive_index, retrive_next_token, retrive_next_sibling (the latter three initialised -1, :78-86).  A token without
    a selected parent is skipped for the links (:79-84) and masks only itself."""
    pl = parent_list.tolist()
    si = selected_index.tolist()
    sl = [int(v) for v in verified_seq_len.tolist()]
    bs = len(si)
    mask = [True] * (sum(sl) * n + n * n * bs)
    positions = [0] * (bs * n)
    ridx = [[-1] * n for _ in range(bs)]
    nxt = [[-1] * n for _ in range(bs)]
    sib = [[-1] * n for _ in range(bs)]
    for b in range(bs):
        base = n * n * b + sum(sl[:b]) * n
        seq = sl[b]

        def parent_slot(i):                       # slot (1-based token) of the parent of token i, 0 = root, None = orphan
            row = si[b][i - 1] // topk
            if row == 0:
                return 0
            want = pl[b][row]
            for j in range(n - 1):
                if si[b][j] == want:
                    return j + 1
            return None

        for t in range(n):
            off = base + (seq + n) * t + seq + 1
            for i in range(n - 1):
                mask[off + i] = False
            level, cur = 0, t
            while cur:
                mask[off + cur - 1] = True
                level += 1
                cur = parent_slot(cur)
                if cur is None:
                    break
            positions[b * n + t] = seq + level
        for i in range(n - 1, 0, -1):
            ridx[b][i] = b * n + i
            p = parent_slot(i)
            if p is None:
                continue
            if nxt[b][p] == -1:
                nxt[b][p] = i
            else:
                sib[b][i] = nxt[b][p]
                nxt[b][p] = i
        ridx[b][0] = b * n
    return mask, positions, ridx, nxt, sib


def verify_tree_greedy(candidates, retrive_index, retrive_next_token, retrive_next_sibling, target_predict, num_spec):
    """VerifyTreeGreedy (eagle_utils.cu:150-191) on lists; predicts / accept_index start at -1 as in the reference's
    test (sgl-kernel/tests/speculative/test_eagle_utils.py:59-65)."""
    cand, ridx = candidates.tolist(), retrive_index.tolist()
    nxt, sib = retrive_next_token.tolist(), retrive_next_sibling.tolist()
    tp = target_predict.flatten().tolist()
    bs, n = len(cand), len(cand[0])
    predicts = [-1] * (bs * n)
    accept_index = [[-1] * num_spec for _ in range(bs)]
    accept_num = [0] * bs
    for b in range(bs):
        last = ridx[b][0]
        accept_index[b][0] = last
        acc, cur = 0, 0
        for _ in range(1, num_spec):
            cur = nxt[b][cur]
            while cur != -1:
                if cand[b][cur] == tp[last]:
                    predicts[last] = tp[last]
                    acc += 1
                    accept_index[b][acc] = ridx[b][cur]
                    last = ridx[b][cur]
                    break
                cur = sib[b][cur]
            if cur == -1:
                break
        accept_num[b] = acc
        predicts[last] = tp[last]
    return predicts, accept_index, accept_num


def tree_speculative_sampling_target_only(candidates, retrive_index, retrive_next_token, retrive_next_sibling, coins,
                                          target_probs, draft_probs, num_spec, threshold_single=1.0, threshold_acc=1.0):
    """TreeSpeculativeSamplingTargetOnly (sgl-kernel/csrc/speculative/speculative_sampling.cuh:38-157), float32 for the
    acceptance test as the kernel, float64 running sum for the final inverse-CDF draw.  draft_probs is updated in place."""
    import numpy as np
    cand, ridx = candidates.tolist(), retrive_index.tolist()
    nxt, sib = retrive_next_token.tolist(), retrive_next_sibling.tolist()
    bs, n = len(cand), len(cand[0])
    d = target_probs.shape[-1]
    q = target_probs.reshape(bs * n, d).numpy()
    p = draft_probs.reshape(bs * n, d).numpy()
    u = coins.numpy().astype(np.float32)
    t_acc = np.float32(max(threshold_acc, 1e-9))
    t_single = np.float32(threshold_single)
    predicts = [-1] * (bs * n)
    accept_index = [[-1] * num_spec for _ in range(bs)]
    accept_num = [0] * bs
    for b in range(bs):
        acc = np.float32(0)
        rowq = b * n
        coin = u[b, 0]
        last = ridx[b][0]
        accept_index[b][0] = last
        k, cur = 0, 0
        for _ in range(1, num_spec):
            cur = nxt[b][cur]
            while cur != -1:
                tok = cand[b][cur]
                pt = q[rowq, tok]
                acc = np.float32(acc + pt)
                if coin <= np.float32(acc / t_acc) or pt >= t_single:
                    acc = np.float32(0)
                    rowq = b * n + cur
                    coin = u[b, cur]
                    predicts[last] = tok
                    k += 1
                    accept_index[b][k] = ridx[b][cur]
                    last = ridx[b][cur]
                    break
                p[rowq, tok] = pt
                cur = sib[b][cur]
            if cur == -1:
                break
        accept_num[b] = k
        rest = q[rowq].astype(np.float64) - (0 if k == num_spec - 1 else p[rowq].astype(np.float64))
        rest = np.maximum(rest, 0)
        cum = np.cumsum(rest)
        hit = np.nonzero((cum > np.float64(coin) * cum[-1]) & (rest > 0))[0]
        predicts[last] = int(hit[0]) if hit.size else d - 1
    return predicts, accept_index, accept_num


# ---------------------------------------------------------------------------------------------- GPTQ int4
def gptq_dequantize(qweight, qzeros, scales, group_size, zero_offset=1, g_idx=None):
    """w[k, n] = (q[k, n] - (z[g(k), n] + zero_offset)) * scales[g(k), n] in fp32, rounded to the scale dtype, with
    g(k) = k // G, or g_idx[k] for an activation-ordered (desc_act) checkpoint.
    Restates the published GPTQ checkpoint format (qweight int32 [K/8, N], nibble e of a word = row 8 * word + e;
    qzeros int32 [K/G, N/8], nibble j = column 8 * word + j; v1 stores zero points minus one).  PARITY UNPINNED: the
    reference delegates this arithmetic to vllm 0.8.4 (layers/quantization/gptq.py:18-31), which is not in the
    reference tree and holds no fixture for it there."""
    qw = qweight.cpu().to(torch.int64) & 0xFFFFFFFF
    qz = qzeros.cpu().to(torch.int64) & 0xFFFFFFFF
    sh = torch.arange(8, dtype=torch.int64) * 4
    q = ((qw.unsqueeze(1) >> sh.view(1, 8, 1)) & 0xF).reshape(-1, qw.shape[1])            # [K, N]
    z = ((qz.unsqueeze(2) >> sh.view(1, 1, 8)) & 0xF).reshape(qz.shape[0], -1) + zero_offset  # [K/G, N]
    g = torch.arange(q.shape[0]) // group_size if g_idx is None else g_idx.cpu().long()
    w = (q - z[g]).float() * scales.cpu().float()[g]
    return w.to(scales.dtype)


def extend_attention_window(q_ext, k_ext, v_ext, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, window,
                            sm_scale, logit_cap=0.0):
    """extend_attention_fwd with SLIDING_WINDOW_SIZE = window (triton_ops/extend_attention.py:179-184): in the stage over
    the listed prefix, row i of a request's extend sees listed token n iff i <= n + window; the new tokens are causal.
    Expressed as a mask for extend_attention_masked below."""
    qo, kp = qo_indptr.tolist(), kv_indptr.tolist()
    masks, mptr = [], [0]
    for b in range(len(qo) - 1):
        ext, pre = qo[b + 1] - qo[b], kp[b + 1] - kp[b]
        i = torch.arange(ext).view(-1, 1)
        allow = torch.cat([i <= torch.arange(pre).view(1, -1) + window, i >= torch.arange(ext).view(1, -1)], dim=1)
        masks.append(allow.flatten())
        mptr.append(mptr[-1] + allow.numel())
    mask = torch.cat(masks).to(torch.uint8)
    return extend_attention_masked(q_ext, k_ext, v_ext, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, mask,
                                   torch.tensor(mptr, dtype=torch.int64), False, sm_scale, logit_cap)


# ---------------------------------------------------------------------------------------------- masked extend
def extend_attention_masked(q_ext, k_ext, v_ext, k_buffer, v_buffer, qo_indptr, kv_indptr, kv_indices, custom_mask,
                            mask_indptr, skip_prefix_custom_mask, sm_scale, logit_cap=0.0):
    """extend_attention_fwd with USE_CUSTOM_MASK (triton_ops/extend_attention.py:93-94,167-178,245-257): per request a
    uint8 [extend_len, prefix_len + extend_len] block; the prefix is fully visible when skip_prefix_custom_mask.
    fp32 torch, one request and head at a time (small cases)."""
    q, ke, ve = q_ext.float().cpu(), k_ext.float().cpu(), v_ext.float().cpu()
    kb, vb = k_buffer.float().cpu(), v_buffer.float().cpu()
    T, Hq, D = q.shape
    Hkv, Dv = ve.shape[1], ve.shape[2]
    o = torch.zeros(T, Hq, Dv)
    qo, kp = qo_indptr.tolist(), kv_indptr.tolist()
    ki, mp = kv_indices.long().cpu(), mask_indptr.tolist()
    m8 = custom_mask.cpu()
    for b in range(len(qo) - 1):
        ext, pre = qo[b + 1] - qo[b], kp[b + 1] - kp[b]
        allow = m8[mp[b]: mp[b] + ext * (pre + ext)].reshape(ext, pre + ext) != 0
        if skip_prefix_custom_mask:
            allow[:, :pre] = True
        loc = ki[kp[b]: kp[b + 1]]
        for h in range(Hq):
            g = h // (Hq // Hkv)
            K = torch.cat([kb[loc, g], ke[qo[b]: qo[b + 1], g]])
            V = torch.cat([vb[loc, g], ve[qo[b]: qo[b + 1], g]])
            s = (q[qo[b]: qo[b + 1], h] @ K.t()) * sm_scale
            if logit_cap > 0:
                s = logit_cap * torch.tanh(s / logit_cap)
            s = s.masked_fill(~allow, float("-inf"))
            o[qo[b]: qo[b + 1], h] = torch.softmax(s, -1) @ V
    return o


# ---------------------------------------------------------------------------------------------- grouped top-k gate
def moe_fused_gate(gating, bias, topk, num_expert_group, topk_group, n_share_experts_fusion=0,
                   routed_scaling_factor=1.0):
    """The grouped, bias-corrected top-k gate with the arithmetic and the tie rules of the reference's kernel
    (sgl-kernel/csrc/moe/moe_fused_gate.cu:102-245): sigmoid, bias add and the sum of a group's two best choice scores
    are rounded to the input type (:104-112,146); the groups kept are the topk_group best, a tie going to the lower
    group (:149-157 removes the higher one first); experts are taken best first, a tie going to the lower id
    (:193-199); weights are the rounded sigmoids over their fp32 sum, the shared expert of row r is
    num_experts + r % n_share at sum / routed_scaling_factor (:225-243).  For fp32 inputs without ties this is
    biased_grouped_topk_impl(renormalize=True) of python/sglang/srt/layers/moe/topk.py:155-215."""
    g = gating.detach().cpu()
    dt = g.dtype
    sig = (1.0 / (1.0 + torch.exp(-g.float()))).to(dt)
    n, E = sig.shape
    choice = (sig.float() + bias.detach().cpu().float().unsqueeze(0)).to(dt).float()
    top2 = choice.view(n, num_expert_group, -1).topk(2, dim=-1)[0]
    gsum = (top2[..., 0] + top2[..., 1]).to(dt).float()
    order = torch.sort(gsum, dim=-1, descending=True, stable=True)[1]
    keep = torch.zeros(n, num_expert_group, dtype=torch.bool)
    keep.scatter_(1, order[:, :topk_group], True)
    mask = keep.unsqueeze(-1).expand(n, num_expert_group, E // num_expert_group).reshape(n, E)
    cand = choice.masked_fill(~mask, float("-inf"))
    k_routed = topk - (1 if n_share_experts_fusion else 0)
    ids = torch.sort(cand, dim=-1, descending=True, stable=True)[1][:, :k_routed]
    w = sig.float().gather(1, ids)
    ssum = w.sum(dim=-1, keepdim=True)
    if n_share_experts_fusion:
        ids = torch.cat([ids, (E + torch.arange(n) % n_share_experts_fusion).unsqueeze(1)], dim=1)
        w = torch.cat([w, ssum / routed_scaling_factor], dim=1)
    return (w / ssum).float(), ids.to(torch.int32)


# ---------------------------------------------------------------------------------------------- fp8 KV cache
# The reference stores an fp8 KV cache as the bytes of `cache_k.to(self.dtype)` (mem_cache/memory_pool.py:389-400,
# after `cache_k.div_(k_scale)` when a scale is given) and its attention kernels read them back with `.to(q.dtype)`
# (triton_ops/decode_attention.py: k / v loads feed tl.dot / tl.sum in the query type).  On ROCm the pool type is
# torch.float8_e5m2fnuz (model_executor/model_runner.py:847-851), on CUDA float8_e5m2 or float8_e4m3fn (:851-854).
# Restated here as integer arithmetic on the fp32 bit pattern: round to nearest even into the format's mantissa,
# denormals included; out of range becomes NaN (e4m3fn, e5m2fnuz) or infinity (e5m2); the fnuz format has no negative
# zero and one NaN (0x80).  tests/test_oracle_golden.py compares both directions with torch's own conversions over every
# byte, every bf16 and every fp16 bit pattern.
KV_FP8_FORMATS = {"e4m3fn": (4, 3, 7), "e5m2": (5, 2, 15), "e5m2fnuz": (5, 2, 16)}  # exponent bits, mantissa bits, bias


def kv_fp8_decode(raw: np.ndarray, fmt: str) -> np.ndarray:
    """uint8 codes -> float32 values."""
    E, M, B = KV_FP8_FORMATS[fmt]
    b = np.asarray(raw, dtype=np.uint8).astype(np.int64)
    sign = np.where(b >> 7, -1.0, 1.0)
    e = (b >> M) & ((1 << E) - 1)
    m = b & ((1 << M) - 1)
    val = np.where(e > 0, (1.0 + m / float(1 << M)) * np.exp2((e - B).astype(np.float64)),
                   m / float(1 << M) * 2.0 ** (1 - B))
    val = sign * val
    if fmt == "e4m3fn":
        val = np.where((b & 0x7F) == 0x7F, np.nan, val)
    elif fmt == "e5m2":
        val = np.where(e == 31, np.where(m == 0, sign * np.inf, np.nan), val)
    else:
        val = np.where(b == 0x80, np.nan, val)
    return val.astype(np.float32)


def kv_fp8_encode(x: np.ndarray, fmt: str) -> np.ndarray:
    """float32 values -> uint8 codes (torch's Tensor.to(float8) rounding and overflow rules)."""
    E, M, B = KV_FP8_FORMATS[fmt]
    bits = np.ascontiguousarray(x, dtype=np.float32).view(np.uint32).astype(np.int64)
    sign = (bits >> 31) & 1
    a = bits & 0x7FFFFFFF
    shift = 23 - M
    # normal results: add the rounding bias (ties to even), re-bias the exponent, keep the top bits
    rounded = a + ((1 << (shift - 1)) - 1) + ((a >> shift) & 1)
    code_n = (rounded >> shift) - ((127 - B) << M)
    # denormal results (below 2^(1-B)): the integer nearest to |x| / 2^(1-B-M), ties to even
    small = a < ((1 - B + 127) << 23)
    mag = np.where(small, a, 0).astype(np.uint32).view(np.float32).astype(np.float64)
    code_d = np.rint(mag / 2.0 ** (1 - B - M)).astype(np.int64)
    code = np.where(small, code_d, code_n)
    if fmt == "e4m3fn":
        code = np.minimum(code, 0x7F)                      # past 464: the NaN code
        code = np.where(a > 0x7F800000, 0x7F, code)
        out = code | (sign << 7)
    elif fmt == "e5m2":
        code = np.minimum(code, 0x7C)                      # infinity
        code = np.where(a > 0x7F800000, 0x7F, code)
        out = code | (sign << 7)
    else:
        over = (code >= 0x80) | (a >= 0x7F800000)
        out = np.where(over, 0x80, np.where(code == 0, 0, code | (sign << 7)))
    return out.astype(np.uint8)


def set_kv_buffer_fp8(cache: torch.Tensor, scale, fmt: str) -> np.ndarray:
    """Bytes the pool receives for rows `cache` (bf16 / fp16 / fp32): memory_pool.py:389-396.  The division by the
    scale is the in-place `div_` of a tensor in the model's type by a Python number, which the device the reference
    runs on evaluates as a product with the fp32 reciprocal, rounded to the tensor's type; then the fp8 rounding."""
    x = cache.detach().cpu()
    if scale is not None:
        inv = np.float32(1.0) / np.float32(scale)
        x = torch.from_numpy(x.float().numpy() * inv).to(x.dtype)
    return kv_fp8_encode(x.float().numpy(), fmt)
