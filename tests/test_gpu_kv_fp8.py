"""GPU parity of the fp8 KV cache: the pool write (bytes equal to torch's Tensor.to(float8), which is what the
reference's MHATokenToKVPool.set_kv_buffer stores, memory_pool.py:389-400) and decode / extend attention over such
pools (every element read back exactly as `.to(q.dtype)`, then the arithmetic of the 16-bit path: same tolerances as
tests/test_gpu_attention.py).  Formats: float8_e5m2fnuz (the reference's pool type on ROCm, model_runner.py:847-851),
float8_e5m2 and float8_e4m3fn (its CUDA types)."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

DEV = "cuda"
FORMATS = {"e4m3fn": torch.float8_e4m3fn, "e5m2": torch.float8_e5m2, "e5m2fnuz": torch.float8_e5m2fnuz}


def _ops():
    from sglang_security_amd import ops

    return ops


def _all_patterns(src):
    """Every 16-bit pattern of a 16-bit float type as [512, 128] rows."""
    return torch.arange(65536, dtype=torch.int32).to(torch.int16).view(src).reshape(512, 128)


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("src", [torch.bfloat16, torch.float16])
def test_set_kv_buffer_fp8_every_pattern(fmt, src):
    """All 65536 bit patterns of the source type (zeros, denormals, ties, out of range, infinities, NaNs) land in the pool
    as the bytes torch's own conversion gives, and as the integer restatement in oracle/ gives."""
    import oracle

    ops = _ops()
    dt = FORMATS[fmt]
    rows = _all_patterns(src)
    T = rows.shape[0]
    k = rows.view(T, 1, 128).to(DEV)
    v = rows.flip(0).contiguous().view(T, 1, 128).to(DEV)
    kb = torch.zeros(T + 5, 1, 128, dtype=torch.uint8, device=DEV).view(dt)
    vb = torch.zeros(T + 5, 1, 128, dtype=torch.uint8, device=DEV).view(dt)
    loc = (torch.randperm(T, generator=torch.Generator().manual_seed(1)) + 3).to(DEV)
    ops.set_kv_buffer(kb, vb, loc, k, v)
    torch.cuda.synchronize()
    got_k, got_v = kb.view(torch.uint8)[loc].cpu().numpy(), vb.view(torch.uint8)[loc].cpu().numpy()
    want_k = rows.to(dt).view(torch.uint8).numpy().reshape(T, 1, 128)
    want_v = rows.flip(0).to(dt).view(torch.uint8).numpy().reshape(T, 1, 128)
    assert np.array_equal(got_k, want_k) and np.array_equal(got_v, want_v)
    assert np.array_equal(got_k.reshape(-1), oracle.kv_fp8_encode(rows.float().numpy().reshape(-1), fmt))
    # the device's own conversion agrees (the reference runs it there)
    assert np.array_equal(got_k, k.to(dt).view(torch.uint8).cpu().numpy())
    # untouched slots stay zero
    assert int(kb.view(torch.uint8)[:3].sum()) == 0 and int(kb.view(torch.uint8)[T + 3:].sum()) == 0


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("src", [torch.bfloat16, torch.float16, torch.float32])
def test_set_kv_buffer_fp8_scaled(fmt, src):
    """With k_scale / v_scale the rows are divided first, in the rows' own type, as `cache_k.div_(k_scale)` does on the
    device (memory_pool.py:391-394); row lengths that are not a multiple of four take the byte-wise path."""
    import oracle

    ops = _ops()
    dt = FORMATS[fmt]
    g = torch.Generator().manual_seed(7)
    T, H, D, Dv = 37, 3, 40, 23
    k = (torch.randn(T, H, D, generator=g) * 3).to(src).to(DEV)
    v = (torch.randn(T, H, Dv, generator=g) * 20).to(src).to(DEV)
    kb = torch.zeros(64, H, D, dtype=torch.uint8, device=DEV).view(dt)
    vb = torch.zeros(64, H, Dv, dtype=torch.uint8, device=DEV).view(dt)
    loc = (torch.randperm(60, generator=g)[:T] + 1).to(torch.int32).to(DEV)
    ks, vs = 0.37, 2.5
    ops.set_kv_buffer(kb, vb, loc, k, v, ks, vs)
    torch.cuda.synchronize()
    idx = loc.long()
    want_k = k.clone().div_(ks).to(dt).view(torch.uint8)
    want_v = v.clone().div_(vs).to(dt).view(torch.uint8)
    assert torch.equal(kb.view(torch.uint8)[idx], want_k) and torch.equal(vb.view(torch.uint8)[idx], want_v)
    assert np.array_equal(kb.view(torch.uint8)[idx].cpu().numpy(), oracle.set_kv_buffer_fp8(k, ks, fmt))
    assert np.array_equal(vb.view(torch.uint8)[idx].cpu().numpy(), oracle.set_kv_buffer_fp8(v, vs, fmt))


def _decode(ops, q, kb, vb, ptr, ind, ns, max_splits, sm, cap=0.0):
    B, Hq, _ = q.shape
    Dv = vb.shape[-1]
    o = torch.zeros(B, Hq, Dv, dtype=q.dtype, device=DEV)
    al = torch.empty(B, Hq, max_splits, Dv, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, max_splits, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q, kb, vb, o, ptr, ind, al, lse, ns, max_splits, sm, cap)
    torch.cuda.synchronize()
    return o


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("qdt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D", [128, 256])
def test_kv8_read_every_code(fmt, qdt, D):
    """One cached token per request whose V rows together hold all 256 codes: softmax over one token is exactly 1, so an
    output row is the decoded V row, which both 16-bit types represent exactly -- compared with torch's decode of the
    same bytes and with the restatement in oracle/ (NaN and infinity codes included).  Head size 128 runs the tile
    kernel (codes widened through its table), 256 the wave-per-row kernel (codes decoded arithmetically)."""
    import oracle

    ops = _ops()
    dt = FORMATS[fmt]
    codes = torch.arange(256, dtype=torch.uint8)
    B = 256 // D
    vb = torch.zeros(B + 2, 1, D, dtype=torch.uint8)
    vb[1: B + 1, 0] = codes.view(B, D)
    kb = torch.zeros(B + 2, 1, D, dtype=torch.uint8)
    q = torch.randn(B, 2, D, generator=torch.Generator().manual_seed(0)).to(qdt).to(DEV)
    ptr = torch.arange(B + 1, dtype=torch.int32, device=DEV)
    ind = torch.arange(1, B + 1, dtype=torch.int32, device=DEV)
    ns = torch.ones(B, dtype=torch.int32, device=DEV)
    o = _decode(ops, q, kb.to(DEV).view(dt), vb.to(DEV).view(dt), ptr, ind, ns, 2, 0.1).float().cpu()
    want = codes.view(dt).float().view(B, 1, D).expand(B, 2, D)
    fin = torch.isfinite(want)
    assert torch.equal(o[fin], want[fin])                       # every finite code, signed zeros aside, exactly
    assert torch.isnan(o[torch.isnan(want)]).all()
    # an infinity code: the tile kernel multiplies the masked rows of its 32-token tile by zero, which turns it into NaN
    # (as a 16-bit infinity in a 16-bit pool does); the wave-per-row kernel keeps it
    if D == 128:
        assert (~torch.isfinite(o[torch.isinf(want)])).all()
    else:
        assert torch.equal(o[torch.isinf(want)], want[torch.isinf(want)])
    mine = torch.from_numpy(oracle.kv_fp8_decode(codes.numpy(), fmt)).view(B, 1, D).expand(B, 2, D)
    assert torch.equal(o[fin], mine[fin])


def _batch(B, Hq, Hkv, D, Dv, prefix, dtype, fmt, seed):
    import oracle

    gen = torch.Generator().manual_seed(seed)
    lens = [prefix + int(torch.randint(1, 150, (1,), generator=gen)) for _ in range(B)]
    n = 1 + prefix + sum(L - prefix for L in lens) + 3
    dt = FORMATS[fmt]
    kb8 = torch.randn(n, Hkv, D, generator=gen).to(dt)
    vb8 = torch.randn(n, Hkv, Dv, generator=gen).to(dt)
    q = torch.randn(B, Hq, D, generator=gen).to(dtype)
    r2t = torch.zeros(B, max(lens) + 4, dtype=torch.int32)
    nxt = 1 + prefix
    for b, L in enumerate(lens):
        r2t[b, :prefix] = torch.arange(1, 1 + prefix)
        r2t[b, prefix:L] = torch.arange(nxt, nxt + L - prefix)
        nxt += L - prefix
    seq = torch.tensor(lens, dtype=torch.int64)
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(seq, 0)
    ind = oracle.create_kv_indices(r2t, torch.arange(B), seq, ptr)
    # the oracle sees the pool through its own decode of the bytes
    kf = torch.from_numpy(oracle.kv_fp8_decode(kb8.view(torch.uint8).numpy(), fmt))
    vf = torch.from_numpy(oracle.kv_fp8_decode(vb8.view(torch.uint8).numpy(), fmt))
    return q, kb8, vb8, kf, vf, ptr, ind, seq


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hkv,D,Dv,prefix,max_splits", [(5, 32, 8, 128, 128, 96, 8), (3, 14, 2, 64, 64, 40, 4),
                                                             (2, 16, 1, 576, 512, 70, 8), (4, 6, 3, 80, 48, 0, 8)])
def test_decode_attention_fp8_pool_vs_oracle(fmt, dtype, B, Hq, Hkv, D, Dv, prefix, max_splits):
    """Shared-prefix batches over fp8 pools (Llama heads, Qwen2's head size 64, the absorbed-MLA row sizes, an odd
    geometry) against the C oracle run on the decoded pool."""
    import oracle

    ops = _ops()
    q, kb8, vb8, kf, vf, ptr, ind, seq = _batch(B, Hq, Hkv, D, Dv, prefix, dtype, fmt, seed=B + Hq)
    ns_d = torch.zeros(B, dtype=torch.int32, device=DEV)
    ops.get_num_kv_splits(ns_d, seq.to(DEV), Hq, Hkv, max_splits, 256)
    sm = 1.0 / math.sqrt(D)
    want = oracle.decode_attention(q, kf, vf, ptr, ind, ns_d.cpu(), sm)
    o = _decode(ops, q.to(DEV), kb8.to(DEV), vb8.to(DEV), ptr.to(DEV), ind.to(DEV), ns_d, max_splits, sm).float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (o - want).abs().max().item() < tol
    assert torch.nn.functional.cosine_similarity(o.flatten(), want.flatten(), dim=0).item() > 0.999


@pytest.mark.parametrize("fmt", ["e5m2fnuz", "e4m3fn"])
def test_shared_prefix_entry_fp8_pool(fmt):
    """The shared-prefix entry point takes an fp8 pool too (per-request kernels over the same lists) and gives the rows
    of the plain entry point, bit for bit."""
    ops = _ops()
    B, Hq, Hkv, D, prefix = 6, 32, 8, 128, 64
    q, kb8, vb8, _, _, ptr, ind, seq = _batch(B, Hq, Hkv, D, D, prefix, torch.bfloat16, fmt, seed=3)
    qd, kd, vd, ptr_d, ind_d = q.to(DEV), kb8.to(DEV), vb8.to(DEV), ptr.to(DEV), ind.to(DEV)
    S = 8
    sm = 1.0 / math.sqrt(D)
    ns = torch.full((B,), S, dtype=torch.int32, device=DEV)
    plain = _decode(ops, qd, kd, vd, ptr_d, ind_d, ns, S, sm)
    ws = ops.decode_shared_workspace(B, Hq, D, S, DEV)
    o = torch.zeros_like(plain)
    gi = torch.tensor([0, B], dtype=torch.int32, device=DEV)
    pl = torch.tensor([prefix], dtype=torch.int32, device=DEV)
    ops.decode_attention_shared_prefix(qd, kd, vd, o, ptr_d, ind_d, gi, pl, ws, S, sm)
    torch.cuda.synchronize()
    assert torch.equal(o, plain)


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hkv,prefix,max_splits", [(24, 32, 8, 200, 8), (5, 28, 4, 33, 4), (70, 8, 8, 64, 2)])
def test_shared_prefix_planned_fp8_pool_vs_oracle(fmt, dtype, B, Hq, Hkv, prefix, max_splits):
    """The one-launch shared-prefix kernel over an fp8 pool (head size 128): groups found on the device from the
    request table, prefix rows streamed once per group and widened through the code table, against the C oracle run on
    the decoded pool and against the per-request kernels over the same lists."""
    import oracle

    ops = _ops()
    D = 128
    q, kb8, vb8, kf, vf, ptr, ind, seq = _batch(B, Hq, Hkv, D, D, prefix, dtype, fmt, seed=B + Hq)
    # the request table _batch built its lists from
    lens = seq.tolist()
    r2t = torch.zeros(B, max(lens) + 4, dtype=torch.int32)
    for b in range(B):
        r2t[b, : lens[b]] = ind[int(ptr[b]): int(ptr[b + 1])]
    qd, kd, vd = q.to(DEV), kb8.to(DEV), vb8.to(DEV)
    ptr_d = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    ind_d = torch.zeros(int(ptr[-1]), dtype=torch.int32, device=DEV)
    ns = torch.zeros(B, dtype=torch.int32, device=DEV)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV)
    ops.decode_metadata_shared(r2t.to(DEV), torch.arange(B, device=DEV), seq.to(DEV), ptr_d, ind_d, ns, Hq, Hkv, max_splits,
                               256, plan)
    torch.cuda.synchronize()
    view = ops.decode_shared_plan_read(plan, B)
    assert view["n_items"] > 0 and view["prefix_lens"][0] == prefix
    sm = 1.0 / math.sqrt(D)
    want = oracle.decode_attention(q, kf, vf, ptr, ind, torch.ones(B, dtype=torch.int32), sm)
    ws = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    outs = []
    for gate in (None, ns, None):          # alone, with the per-request kernels beside it, and again on the workspace
        o = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
        ops.decode_attention_shared_planned(qd, kd, vd, o, ptr_d, ind_d, plan, ws, max_splits, sm, 0.0, None, gate)
        torch.cuda.synchronize()
        assert ops.decode_shared_workspace_status(ws, B, Hq) == 0
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    o = outs[0].float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol
    plain = _decode(ops, qd, kd, vd, ptr_d, ind_d, ns, max_splits, sm).float().cpu()
    assert (o - plain).abs().max().item() < tol


def test_shared_prefix_planned_fp8_pool_nobody_shares():
    """A batch in which nothing is shared: the plan holds no item and every request's whole context is its remainder,
    walked by the same launch (long ones over two tail blocks whose partials meet through the workspace).  Against the
    per-request kernels, twice on one workspace."""
    ops = _ops()
    B, Hq, Hkv, D, S = 7, 32, 8, 128, 8
    q, kb8, vb8, _, _, ptr, ind, seq = _batch(B, Hq, Hkv, D, D, 0, torch.bfloat16, "e4m3fn", seed=9)
    lens = seq.tolist()
    r2t = torch.zeros(B, max(lens) + 4, dtype=torch.int32)
    for b in range(B):
        r2t[b, : lens[b]] = ind[int(ptr[b]): int(ptr[b + 1])]
    qd, kd, vd = q.to(DEV), kb8.to(DEV), vb8.to(DEV)
    ptr_d = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    ind_d = torch.zeros(int(ptr[-1]), dtype=torch.int32, device=DEV)
    ns = torch.zeros(B, dtype=torch.int32, device=DEV)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, S, DEV)
    ops.decode_metadata_shared(r2t.to(DEV), torch.arange(B, device=DEV), seq.to(DEV), ptr_d, ind_d, ns, Hq, Hkv, S, 256, plan)
    assert ops.decode_shared_plan_read(plan, B)["n_items"] == 0
    sm = 1.0 / math.sqrt(D)
    ws = ops.decode_shared_workspace(B, Hq, D, S, DEV)
    plain = _decode(ops, qd, kd, vd, ptr_d, ind_d, ns, S, sm)
    outs = []
    for _ in range(2):
        o = torch.full((B, Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
        ops.decode_attention_shared_planned(qd, kd, vd, o, ptr_d, ind_d, plan, ws, S, sm, 0.0, None, ns)
        torch.cuda.synchronize()
        assert ops.decode_shared_workspace_status(ws, B, Hq) == 0
        outs.append(o)
    assert torch.equal(outs[0], outs[1])
    assert (outs[0].float() - plain.float()).abs().max().item() < 1e-2


@pytest.mark.parametrize("fmt", list(FORMATS))
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("Hq,Hkv,D", [(8, 2, 128), (6, 3, 64)])
def test_extend_attention_fp8_pool_vs_oracle(fmt, dtype, Hq, Hkv, D):
    """Extend over a cached prefix held in an fp8 pool, new tokens in the model's type (the prefix stage reads bytes, the
    causal stage 16-bit rows), ragged, one request without a prefix; against the C oracle on the decoded pool.  Head
    size 128 runs the tile kernel, 64 the wave-per-row kernel."""
    import oracle

    ops = _ops()
    dt = FORMATS[fmt]
    g = torch.Generator().manual_seed(Hq + D)
    pre = [70, 0, 33]
    ext = [17, 40, 5]
    B = len(pre)
    n = sum(pre) + 4
    kb8 = torch.randn(n, Hkv, D, generator=g).to(dt)
    vb8 = torch.randn(n, Hkv, D, generator=g).to(dt)
    T = sum(ext)
    q = torch.randn(T, Hq, D, generator=g).to(dtype)
    ke = torch.randn(T, Hkv, D, generator=g).to(dtype)
    ve = torch.randn(T, Hkv, D, generator=g).to(dtype)
    qo = torch.tensor([0] + list(np.cumsum(ext)), dtype=torch.int32)
    kp = torch.tensor([0] + list(np.cumsum(pre)), dtype=torch.int32)
    ki = (torch.randperm(n - 1, generator=g)[: sum(pre)] + 1).to(torch.int32)
    kf = torch.from_numpy(oracle.kv_fp8_decode(kb8.view(torch.uint8).numpy(), fmt))
    vf = torch.from_numpy(oracle.kv_fp8_decode(vb8.view(torch.uint8).numpy(), fmt))
    sm = 1.0 / math.sqrt(D)
    want = oracle.extend_attention(q, ke, ve, kf, vf, qo, kp, ki, sm)
    o = torch.zeros(T, Hq, D, dtype=dtype, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb8.to(DEV), vb8.to(DEV), qo.to(DEV), kp.to(DEV),
                             ki.to(DEV), None, True, None, max(ext), sm)
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert (o.float().cpu() - want).abs().max().item() < tol


def test_fp8_pool_class():
    """MHATokenToKVPool with the reference's ROCm cache type: byte storage, float8 views out, scaled writes, and decode
    attention over its views equal (to the 16-bit tolerance) to decode over a bf16 pool holding the decoded values."""
    from sglang_security_amd.mem_cache.memory_pool import MHATokenToKVPool

    ops = _ops()
    dt = torch.float8_e5m2fnuz
    pool = MHATokenToKVPool(256, 1, dt, 4, 128, 2, DEV)
    assert pool.store_dtype == torch.uint8 and pool.k_buffer[0].dtype == torch.uint8
    assert pool.get_key_buffer(1).dtype == dt
    g = torch.Generator().manual_seed(11)
    k = torch.randn(50, 4, 128, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(50, 4, 128, generator=g).to(torch.bfloat16).to(DEV)
    loc = torch.arange(1, 51, device=DEV)
    pool.set_kv_buffer(1, loc, k, v, 0.5, None)
    torch.cuda.synchronize()
    assert torch.equal(pool.get_key_buffer(1)[1:51].view(torch.uint8), k.clone().div_(0.5).to(dt).view(torch.uint8))
    assert torch.equal(pool.get_value_buffer(1)[1:51].view(torch.uint8), v.to(dt).view(torch.uint8))
    assert int(pool.k_buffer[0].sum()) == 0
    # decode over the fp8 pool == decode over a bf16 pool holding the decoded values
    q = torch.randn(2, 8, 128, generator=g).to(torch.bfloat16).to(DEV)
    ptr = torch.tensor([0, 20, 50], dtype=torch.int32, device=DEV)
    ind = torch.arange(1, 51, dtype=torch.int32, device=DEV)
    ns = torch.full((2,), 4, dtype=torch.int32, device=DEV)
    kb, vb = pool.get_kv_buffer(1)
    o8 = _decode(ops, q, kb, vb, ptr, ind, ns, 4, 0.088)
    o16 = _decode(ops, q, kb.to(torch.bfloat16), vb.to(torch.bfloat16), ptr, ind, ns, 4, 0.088)
    assert (o8.float() - o16.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("fmt", ["e5m2fnuz", "e4m3fn"])
def test_extend_split_and_paged_prefill_fp8_pool(fmt):
    """One request's short extend over a long fp8 prefix (the split-KV form: prefix slices over blocks + merge) and the
    paged prefill (every token in the pool) against the oracle on the decoded pool."""
    import oracle

    ops = _ops()
    dt = FORMATS[fmt]
    g = torch.Generator().manual_seed(21)
    Hq, Hkv, D, pre, ext = 8, 2, 128, 700, 48
    n = pre + ext + 3
    kb8 = torch.randn(n, Hkv, D, generator=g).to(dt)
    vb8 = torch.randn(n, Hkv, D, generator=g).to(dt)
    q = torch.randn(ext, Hq, D, generator=g).to(torch.bfloat16)
    ke = torch.randn(ext, Hkv, D, generator=g).to(torch.bfloat16)
    ve = torch.randn(ext, Hkv, D, generator=g).to(torch.bfloat16)
    qo = torch.tensor([0, ext], dtype=torch.int32)
    kp = torch.tensor([0, pre], dtype=torch.int32)
    ki = (torch.randperm(n - 1, generator=g)[:pre] + 1).to(torch.int32)
    kf = torch.from_numpy(oracle.kv_fp8_decode(kb8.view(torch.uint8).numpy(), fmt))
    vf = torch.from_numpy(oracle.kv_fp8_decode(vb8.view(torch.uint8).numpy(), fmt))
    sm = 1.0 / math.sqrt(D)
    want = oracle.extend_attention(q, ke, ve, kf, vf, qo, kp, ki, sm)
    assert ops.extend_attention_split_plan(ext, Hq, 1, ext, pre, D, torch.bfloat16) > 1
    o = torch.zeros(ext, Hq, D, dtype=torch.bfloat16, device=DEV)
    ops.extend_attention_fwd(q.to(DEV), ke.to(DEV), ve.to(DEV), o, kb8.to(DEV), vb8.to(DEV), qo.to(DEV), kp.to(DEV),
                             ki.to(DEV), None, True, None, ext, sm)
    torch.cuda.synchronize()
    assert (o.float().cpu() - want).abs().max().item() < 2e-2
    # paged prefill: the last `ext` tokens of the listed pool rows are the new ones (already written to the pool)
    total = pre + ext
    ki_all = (torch.randperm(n - 1, generator=g)[:total] + 1).to(torch.int32)
    kp_all = torch.tensor([0, total], dtype=torch.int32)
    rows = ki_all[pre:].long()
    want_p = oracle.extend_attention(q, kf[rows], vf[rows], kf, vf, qo, kp, ki_all[:pre], sm)
    o2 = torch.zeros_like(o)
    ops.prefill_attention_paged(q.to(DEV), o2, kb8.to(DEV), vb8.to(DEV), qo.to(DEV), kp_all.to(DEV), ki_all.to(DEV), ext,
                                sm)
    torch.cuda.synchronize()
    assert (o2.float().cpu() - want_p).abs().max().item() < 2e-2


def test_fp8_mla_pool_class():
    """MLATokenToKVPool with an fp8 cache type: the latent row [nope | rope] lands as the bytes of Tensor.to(float8), the
    value view is the first kv_lora_rank elements of the key rows, and absorbed-MLA decode over the views agrees with
    decode over a bf16 pool holding the decoded rows."""
    from sglang_security_amd.mem_cache.memory_pool import MLATokenToKVPool

    ops = _ops()
    dt = torch.float8_e5m2fnuz
    pool = MLATokenToKVPool(128, 1, dt, 512, 64, 1, DEV)
    assert pool.kv_buffer[0].dtype == torch.uint8 and pool.get_key_buffer(0).dtype == dt
    g = torch.Generator().manual_seed(2)
    nope = torch.randn(40, 1, 512, generator=g).to(torch.bfloat16).to(DEV)
    rope = torch.randn(40, 1, 64, generator=g).to(torch.bfloat16).to(DEV)
    loc = torch.arange(1, 41, device=DEV)
    pool.set_mla_kv_buffer(0, loc, nope, rope)
    torch.cuda.synchronize()
    kb, vb = pool.get_kv_buffer(0)
    assert kb.shape[-1] == 576 and vb.shape[-1] == 512 and vb.data_ptr() == kb.data_ptr()
    assert torch.equal(kb[1:41].view(torch.uint8), torch.cat([nope, rope], -1).to(dt).view(torch.uint8))
    q = torch.randn(2, 16, 576, generator=g).to(torch.bfloat16).to(DEV)
    ptr = torch.tensor([0, 15, 40], dtype=torch.int32, device=DEV)
    ind = torch.arange(1, 41, dtype=torch.int32, device=DEV)
    ns = torch.full((2,), 2, dtype=torch.int32, device=DEV)
    o8 = _decode(ops, q, kb, vb, ptr, ind, ns, 2, 0.04)
    k16 = kb.to(torch.bfloat16)
    o16 = _decode(ops, q, k16, k16[..., :512], ptr, ind, ns, 2, 0.04)
    assert (o8.float() - o16.float()).abs().max().item() < 2e-2


def test_backend_decode_and_extend_over_fp8_pool():
    """SglkAttnBackend over a pool of the reference's ROCm cache type: forward_extend writes the new tokens' K / V as
    fp8 bytes and attends to the cached prefix, forward_decode (per-request and with the radix groups attached) reads
    them back; against the C oracle on the decoded pool."""
    import types

    import oracle
    from sglang_security_amd import ops as ops_mod
    from sglang_security_amd.attention import SglkAttnBackend
    from test_gpu_backend import _Mode, _layer, _runner

    fmt, dt = "e5m2fnuz", torch.float8_e5m2fnuz
    Hq, Hkv, D, B = 16, 4, 128, 4
    mr, r2t, kv, alloc = _runner(Hq, Hkv, D, dtype=dt)
    g = torch.Generator().manual_seed(4)
    be = SglkAttnBackend(mr)
    layer = _layer(0, Hq, Hkv, D)
    # extend: a shared 64-token prefix is already cached (written through the pool), each request adds its own tail
    prefix = alloc.alloc(64)
    kp = torch.randn(64, Hkv, D, generator=g).to(torch.bfloat16).to(DEV)
    vp = torch.randn(64, Hkv, D, generator=g).to(torch.bfloat16).to(DEV)
    kv.set_kv_buffer(layer, prefix, kp, vp)
    ext = [9, 30, 1, 17]
    locs = []
    for b in range(B):
        e = alloc.alloc(ext[b] + 1)          # one more slot for the decode step below
        r2t.req_to_token[b, :64] = prefix.int()
        r2t.req_to_token[b, 64: 64 + ext[b] + 1] = e.int()
        locs.append(e)
    T = sum(ext)
    q = torch.randn(T, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    k = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    v = torch.randn(T, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    fb = types.SimpleNamespace(forward_mode=_Mode(False), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                               seq_lens=torch.tensor([64 + e for e in ext], device=DEV),
                               extend_prefix_lens=torch.tensor([64] * B, device=DEV),
                               extend_seq_lens=torch.tensor(ext, device=DEV),
                               out_cache_loc=torch.cat([l[:-1] for l in locs]), token_to_kv_pool=kv)
    be.init_forward_metadata(fb)
    o = be.forward(q, k, v, layer, fb)
    torch.cuda.synchronize()
    pool_k = torch.from_numpy(oracle.kv_fp8_decode(kv.k_buffer[0].cpu().numpy(), fmt))
    pool_v = torch.from_numpy(oracle.kv_fp8_decode(kv.v_buffer[0].cpu().numpy(), fmt))
    # the new rows reached the pool as Tensor.to(float8) bytes
    assert torch.equal(kv.k_buffer[0][fb.out_cache_loc].cpu(), k.view(T, Hkv, D).to(dt).view(torch.uint8).cpu())
    qo = torch.tensor([0] + list(np.cumsum(ext)), dtype=torch.int32)
    kpi = torch.tensor([0] + [64 * (i + 1) for i in range(B)], dtype=torch.int32)
    ki = prefix.int().cpu().repeat(B)
    want = oracle.extend_attention(q.cpu().view(T, Hq, D), k.cpu().view(T, Hkv, D), v.cpu().view(T, Hkv, D), pool_k, pool_v,
                                   qo, kpi, ki, layer.scaling)
    assert (o.float().cpu().view(T, Hq, D) - want).abs().max().item() < 2e-2
    # decode: one more token per request
    lens = [64 + e + 1 for e in ext]
    seq_lens = torch.tensor(lens, device=DEV)
    qd = torch.randn(B, Hq * D, generator=g).to(torch.bfloat16).to(DEV)
    kd = torch.randn(B, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    vd = torch.randn(B, Hkv * D, generator=g).to(torch.bfloat16).to(DEV)
    fbd = types.SimpleNamespace(forward_mode=_Mode(True), batch_size=B, req_pool_indices=torch.arange(B, device=DEV),
                                seq_lens=seq_lens, seq_lens_sum=sum(lens),
                                out_cache_loc=torch.stack([l[-1] for l in locs]), token_to_kv_pool=kv)
    be.init_forward_metadata(fbd)
    od = be.forward(qd, kd, vd, layer, fbd)
    torch.cuda.synchronize()
    pool_k = torch.from_numpy(oracle.kv_fp8_decode(kv.k_buffer[0].cpu().numpy(), fmt))
    pool_v = torch.from_numpy(oracle.kv_fp8_decode(kv.v_buffer[0].cpu().numpy(), fmt))
    ptr = torch.zeros(B + 1, dtype=torch.int32)
    ptr[1:] = torch.cumsum(torch.tensor(lens), 0)
    ind = torch.cat([r2t.req_to_token[b, : lens[b]].cpu() for b in range(B)])
    want_d = oracle.decode_attention(qd.cpu().view(B, Hq, D), pool_k, pool_v, ptr, ind, torch.ones(B, dtype=torch.int32),
                                     layer.scaling)
    assert (od.float().cpu().view(B, Hq, D) - want_d).abs().max().item() < 2e-2
    # head size 128: the backend laid a plan out, and its one-launch kernel read the fp8 pool (rows widened through
    # the code table)
    assert be.forward_metadata.shared_plan is not None
    assert ops_mod.decode_shared_plan_read(be.forward_metadata.shared_plan, B)["n_items"] > 0
    # the group-array entry point, handed the groups, routes an fp8 pool per request: the same numbers
    from sglang_security_amd import ops

    md = be.forward_metadata
    gi = torch.tensor([0, B], dtype=torch.int32, device=DEV)
    pl = torch.tensor([64], dtype=torch.int32, device=DEV)
    od2 = torch.empty_like(od)
    ws = ops.decode_shared_workspace(B, Hq, D, 8, DEV)
    ops.decode_attention_shared_prefix(qd.view(B, Hq, D), kv.get_key_buffer(0), kv.get_value_buffer(0), od2.view(B, Hq, D),
                                       md.kv_indptr, md.kv_indices, gi, pl, ws, 8, layer.scaling, 0.0)
    torch.cuda.synchronize()
    assert (od2.float().cpu().view(B, Hq, D) - want_d).abs().max().item() < 2e-2


def test_fused_rotation_kernels_refuse_a_retyped_pool():
    """The kernels that rotate and write the new K / V rows in one launch store 16-bit rows: handed an fp8 pool they must
    refuse (rope + set_kv_buffer is the path that rounds to the pool's format), not write 16-bit rows over bytes."""
    from sglang_security_amd._lib import SglkError

    ops = _ops()
    qkv = torch.zeros(2, (4 + 2 * 2) * 128, dtype=torch.bfloat16, device=DEV)
    cs = torch.zeros(8, 128, device=DEV)
    pos = torch.zeros(2, dtype=torch.int64, device=DEV)
    loc = torch.tensor([1, 2], device=DEV)
    kb = torch.zeros(4, 2, 128, dtype=torch.uint8, device=DEV).view(torch.float8_e5m2fnuz)
    with pytest.raises(SglkError):
        ops.rope_and_store_kv(qkv, cs, pos, kb, kb, loc, 4, 2, 128)
    part = torch.zeros(2 * qkv.numel(), dtype=torch.float32, device=DEV)
    with pytest.raises(SglkError):
        ops.rope_and_store_kv_partial(qkv, part, 2, cs, pos, kb, kb, loc, 4, 2, 128)
    assert int(kb.view(torch.uint8).sum()) == 0
