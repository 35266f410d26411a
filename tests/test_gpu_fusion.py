"""GPU: the fusion forms of the decode layer (split-K partials consumed by the next kernel, block sums produced by
the previous one) against the unfused ops: bit-exact where the arithmetic is the same, oracle tolerance elsewhere."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from sglang_security_amd import ops

    return ops


def _block_sums(t):  # [rows, n] -> [n/128, rows] fp32
    r, n = t.shape
    return t.float().view(r, n // 128, 128).sum(-1).t().contiguous()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("rows,hidden,splits", [(64, 4096, 4), (5, 512, 1), (33, 8192, 7)])
def test_rmsnorm_ex_matches_unfused(dtype, rows, hidden, splits):
    ops = _ops()
    g = torch.Generator().manual_seed(rows + hidden)
    parts = torch.randn(splits, rows, hidden, generator=g).to(DEV)
    res = torch.randn(rows, hidden, generator=g).to(dtype).to(DEV)
    w = (1 + 0.1 * torch.randn(hidden, generator=g)).to(dtype).to(DEV)
    # unfused: sum in split order, round once, then the ordinary op
    acc = torch.zeros(rows, hidden, device=DEV)
    for s in range(splits):
        acc = acc + parts[s]
    x_ref = acc.to(dtype)
    r_ref = res.clone()
    ops.fused_add_rmsnorm(x_ref, r_ref, w, 1e-5)
    x = torch.empty(rows, hidden, dtype=dtype, device=DEV)
    r = res.clone()
    xs = torch.zeros(hidden // 128, rows + 3, device=DEV)
    ops.fused_add_rmsnorm_ex(x, r, w, 1e-5, partials=parts, splits=splits, xsum_out=xs)
    # the residual is a plain sum: bit-exact.  The normalised row depends on the order of the sum of squares (the
    # fused kernel uses wider blocks), so it may differ from the unfused kernel by one rounding step in a few places
    assert torch.equal(r.view(torch.int16), r_ref.view(torch.int16))
    assert (x.view(torch.int16).int() - x_ref.view(torch.int16).int()).abs().max().item() <= 1
    assert (x.view(torch.int16) != x_ref.view(torch.int16)).float().mean().item() < 1e-3
    torch.testing.assert_close(xs[:, :rows], _block_sums(x), rtol=1e-4, atol=2e-2)
    # plain forms: no partials, no residual
    x2 = x_ref.clone()
    y_ref = torch.empty_like(x2)
    ops.rmsnorm(y_ref, x2, w, 1e-5)
    ops.fused_add_rmsnorm_ex(x2, None, w, 1e-5)
    assert (x2.view(torch.int16).int() - y_ref.view(torch.int16).int()).abs().max().item() <= 1


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_act_and_mul_xs(dtype):
    ops = _ops()
    g = torch.Generator().manual_seed(1)
    x = torch.randn(64, 2 * 1024, generator=g).to(dtype).to(DEV)
    a, b = torch.empty(64, 1024, dtype=dtype, device=DEV), torch.empty(64, 1024, dtype=dtype, device=DEV)
    xs = torch.zeros(8, 64, device=DEV)
    ops.act_and_mul(a, x, 0)
    ops.act_and_mul_xs(b, x, 0, xs)
    assert torch.equal(a.view(torch.int16), b.view(torch.int16))
    torch.testing.assert_close(xs, _block_sums(b), rtol=1e-4, atol=2e-2)


def test_rope_and_store_kv_partial():
    ops = _ops()
    g = torch.Generator().manual_seed(2)
    T, Hq, Hkv, D, splits = 9, 8, 2, 128, 3
    row = (Hq + 2 * Hkv) * D
    parts = torch.randn(splits, T, row, generator=g).to(DEV)
    cs = torch.randn(64, D, generator=g).to(DEV)
    pos = torch.randint(0, 64, (T,), generator=g).to(DEV)
    loc = (torch.randperm(30, generator=g)[:T] + 1).to(DEV)
    acc = torch.zeros(T, row, device=DEV)
    for s in range(splits):
        acc = acc + parts[s]
    qkv_ref = acc.to(torch.bfloat16)
    kb1, vb1 = torch.zeros(32, Hkv, D, dtype=torch.bfloat16, device=DEV), torch.zeros(32, Hkv, D, dtype=torch.bfloat16, device=DEV)
    kb2, vb2 = torch.zeros_like(kb1), torch.zeros_like(vb1)
    ops.rope_and_store_kv(qkv_ref, cs, pos, kb1, vb1, loc, Hq, Hkv, D)
    qkv = torch.empty(T, row, dtype=torch.bfloat16, device=DEV)
    ops.rope_and_store_kv_partial(qkv, parts, splits, cs, pos, kb2, vb2, loc, Hq, Hkv, D)
    assert torch.equal(qkv.view(torch.int16), qkv_ref.view(torch.int16))
    assert torch.equal(kb1, kb2) and torch.equal(vb1, vb2)


@pytest.mark.parametrize("M,K,N", [(64, 4096, 6144), (64, 4096, 4096), (17, 1024, 256), (64, 14336, 4096)])
def test_awq_gemm_partial_and_block_sums(M, K, N):
    """Partials summed by the caller, and the GEMM fed with the producer's block sums, against the oracle."""
    import oracle
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_ops import _awq_case

    ops = _ops()
    x, qw, sc, qz = _awq_case(M, K, N, 128, torch.bfloat16, seed=K + N)
    want = oracle.awq_gemm(x, qw, sc, qz).float()
    # per element, the reference's bar for bf16 products (test/srt/cpu/utils.py:6-10: atol = rtol = 1e-2)
    close = lambda got: torch.testing.assert_close(got, want, atol=1e-2, rtol=1e-2)
    xd, packed, scd, qzd = x.to(DEV), ops.awq_repack(qw.to(DEV)), sc.to(DEV), qz.to(DEV)
    xs = _block_sums(xd)
    splits = ops.awq_gemm_partial_splits(M, N, K)
    assert splits >= 1
    for use_xs in (None, xs):
        part = torch.full((splits, M, N), float("nan"), device=DEV)
        assert ops.awq_gemm_partial(xd, packed, scd, qzd, part, use_xs) == splits
        close(part.sum(0).to(torch.bfloat16).float().cpu())
    out = torch.empty(M, N, dtype=torch.bfloat16, device=DEV)
    work = torch.zeros(max(ops.awq_gemm_workspace_bytes(M, N, K), 256), dtype=torch.uint8, device=DEV)
    ops.awq_gemm_xs(xd, packed, scd, qzd, xs, out, work)
    close(out.float().cpu())


@pytest.mark.parametrize("M", [64, 17])
def test_awq_chain_gate_up_down_equals_separate_launches(M):
    """gate_up (+ gated activation) and down in one launch of persistent blocks against the two separate launches, at
    Llama-3-8B's shapes: the activation rows, their block sums and the down projection's split sums must be the same
    bits; twice over one set of counters (every launch leaves them zero)."""
    ops = _ops()
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_ops import _awq_case

    hidden, inter = 4096, 14336
    x, qw, sc, qz = _awq_case(M, hidden, 2 * inter, 128, torch.bfloat16, seed=1)
    _, qw2, sc2, qz2 = _awq_case(M, inter, hidden, 128, torch.bfloat16, seed=2)
    xd = x.to(DEV)
    qi, si, zi = ops.awq_interleave_gate_up(qw.to(DEV), sc.to(DEV), qz.to(DEV))
    gu = (ops.awq_repack(qi), si, zi)
    dn = (ops.awq_repack(qw2.to(DEV)), sc2.to(DEV), qz2.to(DEV))
    xs = _block_sums(xd)
    splits = ops.awq_gemm_partial_splits(M, hidden, inter)
    # separate launches
    act1 = torch.empty(M, inter, dtype=torch.bfloat16, device=DEV)
    xs_act1 = torch.zeros(inter // 128, M, device=DEV)
    part1 = torch.full((splits, M, hidden), float("nan"), device=DEV)
    ops.awq_gemm_silu_mul(xd, *gu, act1, xs, xs_act1)
    ops.awq_gemm_partial(act1, *dn, part1, xs_act1)
    # one launch
    sync = ops.awq_chain_sync(DEV)
    for rep in range(2):
        act2 = torch.full((M, inter), float("nan"), dtype=torch.bfloat16, device=DEV)
        xs_act2 = torch.zeros(inter // 128, M, device=DEV)
        part2 = torch.full((splits, M, hidden), float("nan"), device=DEV)
        ops.awq_chain([ops.chain_silu_mul(xd, *gu, xs, act2, xs_act2), ops.chain_partial(act2, *dn, xs_act2, part2)], sync)
        torch.cuda.synchronize()
        assert ops.awq_chain_status(sync) == 0
        assert int(sync.abs().sum()) == 0
        assert torch.equal(act1.view(torch.int16), act2.view(torch.int16))
        assert torch.equal(xs_act1, xs_act2)
        assert torch.equal(part1, part2)


@pytest.mark.parametrize("B,Hq,Hkv,splits", [(64, 32, 8, 4), (5, 16, 4, 3), (3, 8, 8, 1)])
def test_rope_role_inside_the_attention_launch(B, Hq, Hkv, splits):
    """The rotation + pool write as the first blocks of the attention launch against the two separate launches: query
    rows, new pool rows, attention output and its block sums must be the same bits; two 'layers' of one step (the
    counters the attention blocks wait on grow by launch), then a second step over a fresh plan."""
    ops = _ops()
    g = torch.Generator().manual_seed(B + Hq)
    D, P, max_splits = 128, 96, 8
    tails = [1 + (7 * b) % 40 for b in range(B)]
    n = 1 + P + sum(tails) + 4
    row = (Hq + 2 * Hkv) * D
    r2t = torch.zeros(B, P + 64, dtype=torch.int32)
    nxt = 1 + P
    for b in range(B):
        r2t[b, :P] = torch.arange(1, 1 + P)
        r2t[b, P: P + tails[b]] = torch.arange(nxt, nxt + tails[b])
        nxt += tails[b]
    lens = torch.tensor([P + t for t in tails])
    loc = torch.tensor([int(r2t[b, lens[b] - 1]) for b in range(B)], device=DEV)
    pos = (lens - 1).to(DEV)
    cs = torch.randn(256, D, generator=g).to(DEV)
    ptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    ind = torch.zeros(int(lens.sum()), dtype=torch.int32, device=DEV)
    ns = torch.zeros(B, dtype=torch.int32, device=DEV)
    rpi = torch.arange(B, device=DEV)
    sm = 1.0 / math.sqrt(D)
    pools = [[torch.randn(n, Hkv, D, generator=g).to(torch.bfloat16).to(DEV) for _ in range(2)] for _ in range(2)]
    ref_pools = [[t.clone() for t in pair] for pair in pools]
    ws1 = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    ws2 = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    plan1 = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV)
    plan2 = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV)
    for step in range(2):
        for plan in (plan1, plan2):
            ops.decode_metadata_shared(r2t.to(DEV), rpi, lens.to(DEV), ptr, ind, ns, Hq, Hkv, max_splits, 256, plan)
        for layer in range(2):
            parts = torch.randn(splits, B, row, generator=g).to(DEV)
            kb1, vb1 = ref_pools[layer]
            kb2, vb2 = pools[layer]
            # separate launches
            qkv1 = torch.empty(B, row, dtype=torch.bfloat16, device=DEV)
            ops.rope_and_store_kv_partial(qkv1, parts, splits, cs, pos, kb1, vb1, loc, Hq, Hkv, D)
            o1 = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
            xs1 = torch.zeros(Hq, B, device=DEV)
            ops.decode_attention_shared_planned(qkv1[:, : Hq * D].view(B, Hq, D), kb1, vb1, o1, ptr, ind, plan1, ws1,
                                                max_splits, sm, 0.0, xs1)
            # one launch
            qkv2 = torch.zeros(B, row, dtype=torch.bfloat16, device=DEV)
            o2 = torch.full((B, Hq, D), float("nan"), dtype=torch.bfloat16, device=DEV)
            xs2 = torch.zeros(Hq, B, device=DEV)
            ops.decode_attention_shared_planned_rope(parts, splits, B * row, None, cs, pos, loc,
                                                     qkv2[:, : Hq * D].view(B, Hq, D), kb2, vb2, o2, ptr, ind, plan2,
                                                     layer + 1, ws2, max_splits, sm, 0.0, xs2)
            torch.cuda.synchronize()
            assert ops.decode_shared_workspace_status(ws2, B, Hq) == 0
            assert torch.equal(qkv1[:, : Hq * D].view(torch.int16), qkv2[:, : Hq * D].view(torch.int16)), (step, layer)
            assert torch.equal(kb1, kb2) and torch.equal(vb1, vb2), (step, layer)
            assert torch.equal(o1.view(torch.int16), o2.view(torch.int16)), (step, layer)
            assert torch.equal(xs1, xs2), (step, layer)


def test_attention_block_sums():
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_attention_extend import _group_batch

    ops = _ops()
    q, kb, vb, ptr, ind, gi, pls = _group_batch([(6, 70, 1, 40)], 32, 8, torch.bfloat16, seed=3)
    B, Hq, D = q.shape
    sm = 1.0 / math.sqrt(D)
    o1 = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    o2 = torch.empty_like(o1)
    xs1, xs2 = torch.zeros(Hq, B, device=DEV), torch.zeros(Hq, B + 2, device=DEV)
    al = torch.empty(B, Hq, 8, D, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, 8, dtype=torch.float32, device=DEV)
    ns = torch.full((B,), 4, dtype=torch.int32, device=DEV)
    ops.decode_attention_fwd_xs(q.to(DEV), kb.to(DEV), vb.to(DEV), o1, ptr.to(DEV), ind.to(DEV), al, lse, ns, 8, sm, 0.0, xs1)
    ws = ops.decode_shared_workspace(B, Hq, D, 8, DEV)
    ops.decode_attention_shared_prefix_xs(q.to(DEV), kb.to(DEV), vb.to(DEV), o2, ptr.to(DEV), ind.to(DEV), gi.to(DEV),
                                          pls.to(DEV), ws, 8, sm, 0.0, xs2)
    torch.testing.assert_close(xs1, _block_sums(o1.view(B, Hq * D)), rtol=1e-4, atol=2e-2)
    torch.testing.assert_close(xs2[:, :B], _block_sums(o2.view(B, Hq * D)), rtol=1e-4, atol=2e-2)
    assert (o1.float() - o2.float()).abs().max().item() < 2e-2


@pytest.mark.parametrize("groups,max_splits", [([(6, 70, 1, 40)], 8), ([(3, 200, 1, 9), (2, 0, 1, 33), (4, 64, 64, 65)], 4),
                                               ([(64, 1024, 129, 129)], 8)])
def test_attention_with_rope_folded_in(groups, max_splits):
    """Shared-prefix decode attention fed with the qkv projection's split-K partial sums (rotation and the new pool
    rows inside the kernel) against the separate rope_and_store_kv_partial + attention launches on the same
    inputs: the same pool rows bit for bit, the same outputs and block sums bit for bit."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_attention_extend import _group_batch

    ops = _ops()
    Hq, Hkv, D, splits = 32, 8, 128, 3
    _, kb, vb, ptr, ind, gi, pls = _group_batch(groups, Hq, Hkv, torch.bfloat16, seed=5)
    B = ptr.shape[0] - 1
    g = torch.Generator().manual_seed(11)
    row = (Hq + 2 * Hkv) * D
    parts = (torch.randn(splits, B, row, generator=g) * 0.6).to(DEV)
    cs_ang = torch.rand(512, D // 2, generator=g) * 6.28
    cs = torch.cat([cs_ang.cos(), cs_ang.sin()], dim=1).to(DEV)
    pos = torch.randint(0, 512, (B,), generator=g).to(DEV)
    # every request's new row goes to the slot its list ends with
    loc = ind[(ptr[1:] - 1).long()].to(torch.int64).to(DEV)
    sm = 1.0 / math.sqrt(D)
    kb1, vb1 = kb.to(DEV).clone(), vb.to(DEV).clone()
    kb2, vb2 = kb.to(DEV).clone(), vb.to(DEV).clone()
    meta = [t.to(DEV) for t in (ptr, ind, gi, pls)]
    qkv = torch.empty(B, row, dtype=torch.bfloat16, device=DEV)
    o1 = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    o2 = torch.full_like(o1, float("nan"))
    xs1, xs2 = torch.zeros(Hq, B, device=DEV), torch.zeros(Hq, B, device=DEV)
    ws = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    ops.rope_and_store_kv_partial(qkv, parts, splits, cs, pos, kb1, vb1, loc, Hq, Hkv, D)
    ops.decode_attention_shared_prefix_xs(qkv[:, : Hq * D].view(B, Hq, D), kb1, vb1, o1, *meta, ws, max_splits, sm, 0.0, xs1)
    for _ in range(2):   # twice: the second launch finds the rows already in place and the counters back at zero
        ops.decode_attention_shared_prefix_rope(parts, splits, B * row, None, cs, pos, loc, kb2, vb2, o2, *meta, ws,
                                                max_splits, sm, 0.0, xs2)
        torch.cuda.synchronize()
        assert torch.equal(kb1.view(torch.int16), kb2.view(torch.int16))
        assert torch.equal(vb1.view(torch.int16), vb2.view(torch.int16))
        assert torch.equal(o1.view(torch.int16), o2.view(torch.int16))
        assert torch.equal(xs1, xs2)
    assert ops.decode_shared_workspace_status(ws, B, Hq) == 0


def test_attention_with_rope_folded_in_rejects():
    ops = _ops()
    from sglang_security_amd.ops import SglkError
    B, Hq, Hkv, D = 2, 8, 2, 128
    z = lambda *s, dt=torch.float32: torch.zeros(*s, dtype=dt, device=DEV)
    i32 = lambda *v: torch.tensor(v, dtype=torch.int32, device=DEV)
    ws = ops.decode_shared_workspace(B, Hq, D, 4, DEV)
    kb = z(8, Hkv, D, dt=torch.bfloat16)
    o = z(B, Hq, D, dt=torch.bfloat16)
    args = lambda cs: (z(1, B, (Hq + 2 * Hkv) * D), 1, B * (Hq + 2 * Hkv) * D, None, cs, z(B, dt=torch.int64),
                       z(B, dt=torch.int64), kb, kb.clone(), o, i32(0, 2, 4), i32(1, 2, 1, 3), i32(0, 2), i32(1), ws, 4, 1.0,
                       0.0, None)
    with pytest.raises(SglkError):      # partial rotation
        ops.decode_attention_shared_prefix_rope(*args(z(16, 64)))
    with pytest.raises(SglkError):      # pairs (2p, 2p + 1)
        ops.decode_attention_shared_prefix_rope(*args(z(16, 128)), interleave=True)


def test_fused_decode_step_matches_unfused():
    """Two decoders with identical weights and batch: one runs the layer fusion, one the plain op sequence.  Both are
    fed the plain decoder's tokens after every step (a greedy pick between two logits one rounding apart may
    legitimately differ, and the sequences would part there)."""
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = LlamaConfig.tiny()
    decs = []
    for fused in (False, True):
        dec = LlamaDecoder(cfg, batch=6, pool_slots=1 + 64 + 6 * 24, max_ctx=96, device=DEV, seed=0)
        dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
        dec.enable_shared_prefix(True)
        assert dec.enable_fusion(fused) == fused
        decs.append(dec)
    ref, fus = decs
    for _ in range(3):
        ref.step()
        fus.step()
        torch.cuda.synchronize()
        l0, l1 = ref.logits.float().cpu(), fus.logits.float().cpu()
        assert torch.isfinite(l1).all()
        assert (l0 - l1).abs().max().item() < 0.05 * max(1.0, l0.abs().max().item())
        # same pick wherever the plain decoder's pick is clear of the runner-up by more than the tolerance
        top = l0.topk(2, dim=-1).values
        clear = (top[:, 0] - top[:, 1]) > 0.1
        assert torch.equal(ref.next_ids.cpu()[clear], fus.next_ids.cpu()[clear])
        fus.input_ids.copy_(ref.input_ids)
    k0, k1 = ref.k_pool[1].float().cpu(), fus.k_pool[1].float().cpu()
    assert (k0 - k1).abs().max().item() < 0.05 * max(1.0, k0.abs().max().item())


def test_fused_layer_at_llama3_8b_shape_matches_the_unfused_sequence_per_element():
    """One decoder layer of Llama-3-8B AWQ (hidden 4096, 32 / 8 heads of 128, MLP 14336) at the benchmark's batch (64
    requests, 1024 shared + 128 private tokens): the fused step (block sums from the producers, split sums to the
    consumers, activation in the gate_up epilogue) against the plain op sequence on the same weights and the same
    batch, first with the per-request attention kernel on both sides, then with the shared-prefix kernel driven by
    the device plan on the fused side.  The fused path rounds where the unfused sequence rounds; what differs is the
    order in which the 128-wide block sums of the activations are added (the producer's fp32 sum against the GEMM's
    own), a relative 1e-7 that flips a last bf16 bit here and there.  So: the new K rows are the same bits, and
    V rows and hidden states agree per element within atol = rtol = 1e-2 (one bf16 ulp at the values' size), logits
    within atol 2e-2."""
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = dataclasses_replace_layers(LlamaConfig.llama3_8b("awq"), 1)
    B, P, S = 64, 1024, 128
    decs = {}
    for name, fused, shared in (("plain", False, False), ("fused", True, False), ("fused_shared", True, True)):
        dec = LlamaDecoder(cfg, batch=B, pool_slots=1 + P + B * (S + 8), max_ctx=P + S + 8, device=DEV, seed=0,
                           keep_reference_layout=True)
        dec.setup_shared_prefix_batch(prefix_len=P, suffix_len=S)
        dec.enable_shared_prefix(shared)
        assert dec.enable_fusion(fused) == fused
        decs[name] = dec
    a = decs["plain"]
    for _ in range(2):
        ids = a.input_ids.clone()      # every decoder takes the plain one's tokens, so step 2 compares like with like
        for dec in decs.values():
            dec.input_ids.copy_(ids)
            dec.step()
        torch.cuda.synchronize()
        for name in ("fused", "fused_shared"):
            b = decs[name]
            assert torch.isfinite(b.logits.float()).all()
            assert torch.equal(a.k_pool[0], b.k_pool[0]), name
            torch.testing.assert_close(b.v_pool[0].float(), a.v_pool[0].float(), atol=1e-2, rtol=1e-2)
            # same attention kernel on both sides: one ulp.  The shared-prefix kernel adds the same products in another
            # order (prefix columns per group, then the private ones), which moves attn_out by an ulp in many places
            # and the layer's output with it: the bound there is the attention tests' own, carried through o / MLP
            tol = 1e-2 if name == "fused" else 5e-2
            torch.testing.assert_close(b.hidden.float(), a.hidden.float(), atol=tol, rtol=1e-2)
            # a logit sums 4096 products of those hidden values: a few flipped last bits move one near zero by more
            # than 1e-2 (12 of 8.2 million did, the largest by 0.0146), so the absolute part is doubled here
            torch.testing.assert_close(b.logits.float(), a.logits.float(), atol=2 * tol, rtol=1e-2)
            # greedy tokens: equal except where the two best logits are within the tolerance of each other
            differ = (a.next_ids != b.next_ids).nonzero().flatten().tolist()
            for r in differ:
                row = a.logits[r].float()
                assert (row[a.next_ids[r]] - row[b.next_ids[r]]).abs().item() <= 2 * tol + 2e-2 * row.abs().max().item()


def dataclasses_replace_layers(cfg, n):
    import dataclasses

    return dataclasses.replace(cfg, layers=n)


@pytest.mark.parametrize("quant", ["fp8", "bf16"])
def test_decode_step_other_weight_formats(quant):
    """The decode step on fp8 (w8a8: per-token activation quantisation + fp8_scaled_mm) and bf16 (dense, split-K
    through the workspace) weights: every linear against the oracle on the step's own inputs."""
    import oracle
    from sglang_security_amd import ops
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = LlamaConfig.tiny(quant)
    dec = LlamaDecoder(cfg, batch=6, pool_slots=1 + 64 + 6 * 24, max_ctx=96, device=DEV, seed=0)
    dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
    dec.enable_shared_prefix(True)
    if quant == "fp8":
        assert not dec.enable_fusion(True)        # no layer fusion for the fp8 step
    for _ in range(2):
        dec.step()
    torch.cuda.synchronize()
    assert torch.isfinite(dec.logits.float()).all()
    x = torch.randn(6, cfg.hidden, device=DEV).to(torch.bfloat16)
    w = dec.layers[1]["gate_up"]
    out = torch.empty(6, 2 * cfg.inter, dtype=torch.bfloat16, device=DEV)
    dec._linear("gate_up", w, x, out)
    if quant == "fp8":
        xq = torch.empty(6, cfg.hidden, dtype=torch.float8_e4m3fn, device=DEV)
        xs = torch.empty(6, dtype=torch.float32, device=DEV)
        ops.per_token_quant_fp8(x, xq, xs)
        want = oracle.fp8_scaled_mm(xq.cpu(), w.weight.cpu().t(), xs.cpu(), w.scale.cpu(), torch.bfloat16).float()
    else:
        want = oracle.half_matmul(x.cpu(), w.weight.cpu().t().contiguous()).float()
    got = out.float().cpu()
    assert (got - want).abs().max().item() <= 2e-2 * max(1.0, want.abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,inter", [(64, 1024, 512), (17, 512, 128), (64, 4096, 14336)])
def test_gate_up_with_activation_epilogue(dtype, M, K, inter):
    """gate_up GEMM with silu(gate) * up in its epilogue on the interleaved weight: bit-identical to the planner's
    GEMM on the original weight followed by silu_and_mul (same accumulation order, same rounding points), and the
    row-block sums it leaves match its output."""
    import sys, os
    sys.path.insert(0, os.path.dirname(__file__))
    from test_gpu_ops import _awq_case

    ops = _ops()
    N = 2 * inter
    x, qw, sc, qz = _awq_case(M, K, N, 128, dtype, seed=K + inter)
    xd, qwd, scd, qzd = x.to(DEV), qw.to(DEV), sc.to(DEV), qz.to(DEV)
    import os as _os
    _os.environ["SGLK_AWQ_WIDE"] = "2"  # the comparison kernel must be the wide one (same accumulation order)
    _os.environ["SGLK_AWQ_WIDE_SPS"] = str(K // 512)  # ... with a single split
    try:
        gu = ops.awq_gemm(xd, ops.awq_repack(qwd), scd, qzd)
    finally:
        _os.environ.pop("SGLK_AWQ_WIDE", None)
        _os.environ.pop("SGLK_AWQ_WIDE_SPS", None)
    want = torch.empty(M, inter, dtype=dtype, device=DEV)
    ops.act_and_mul(want, gu, 0)
    qi, si, zi = ops.awq_interleave_gate_up(qwd, scd, qzd)
    out = torch.full((M, inter), float("nan"), dtype=dtype, device=DEV)
    xs_out = torch.zeros(inter // 128, M + 1, device=DEV)
    xs_in = _block_sums(xd)
    ops.awq_gemm_silu_mul(xd, ops.awq_repack(qi), si, zi, out, None, xs_out)
    assert torch.equal(out.view(torch.int16), want.view(torch.int16))
    torch.testing.assert_close(xs_out[:, :M], _block_sums(out), rtol=1e-4, atol=2e-2)
    # with the producer's block sums of x: same result up to the order of the fp32 block sums
    out2 = torch.empty_like(out)
    ops.awq_gemm_silu_mul(xd, ops.awq_repack(qi), si, zi, out2, xs_in, None)
    assert (out2.float() - out.float()).abs().max().item() <= 2e-2 * max(1.0, out.float().abs().max().item())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("M,K,N", [(64, 4096, 6144), (64, 4096, 4096), (64, 14336, 4096), (64, 4096, 28672), (7, 512, 256),
                                   (33, 1024, 48)])
def test_dense_packed_split_k_forms(dtype, M, K, N):
    """The streaming-layout product split along K: slabs summed by the second launch (_ws) and slabs left to the
    consumer (_partial) give the bits of the row-major split-K product (same splits, same order of accumulation),
    and agree with the oracle's 16-bit product."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(M + N)
    x = torch.randn(M, K, generator=g).to(dtype).to(DEV)
    w = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dtype).to(DEV)
    packed = ops.dense_repack(w)
    ws = torch.empty(max(ops.dense_gemm_workspace_bytes(M, N, K), ops.dense_gemm_partial_splits(M, N, K) * M * N * 4, 16)
                     // 4, dtype=torch.float32, device=DEV)
    ref = torch.empty(M, N, dtype=dtype, device=DEV)
    ops.dense_gemm_nt(x, w, None, ref, ws)
    out = torch.full((M, N), float("nan"), dtype=dtype, device=DEV)
    ops.dense_gemm_packed_ws(x, packed, None, out, ws)
    torch.cuda.synchronize()
    assert torch.equal(out, ref)
    want = oracle.half_matmul(x.cpu(), w.cpu().t().contiguous()).float()
    torch.testing.assert_close(out.float().cpu(), want, atol=1e-2, rtol=1e-2)
    splits = ops.dense_gemm_partial_splits(M, N, K)
    assert splits >= 1
    part = torch.full((splits * M * N + 8,), float("nan"), dtype=torch.float32, device=DEV)
    assert ops.dense_gemm_packed_partial(x, packed, part) == splits
    torch.cuda.synchronize()
    assert torch.isnan(part[splits * M * N:]).all()                      # nothing past the slabs
    acc = torch.zeros(M, N, dtype=torch.float32, device=DEV)
    for s_ in range(splits):                                             # split order, fp32, one rounding
        acc += part[s_ * M * N: (s_ + 1) * M * N].view(M, N)
    assert torch.equal(acc.to(dtype), ref)
    if N % 32 == 0:
        # the gate_up form: the activation in the launch that sums the splits == product, then act_and_mul
        for act in (0, 1, 2):
            want_act = torch.empty(M, N // 2, dtype=dtype, device=DEV)
            ops.act_and_mul(want_act, ref, act)
            got_act = torch.full((M, N // 2), float("nan"), dtype=dtype, device=DEV)
            ops.dense_gemm_packed_act(x, packed, got_act, act, ws)
            torch.cuda.synchronize()
            assert torch.equal(got_act, want_act)


def test_dense_fused_decode_step_matches_unfused():
    """16-bit weights: the step over the streaming layout with the split sums folded into their consumers against the
    plain op sequence over the row-major weights.  The arithmetic is the same (same splits, summed in the same order,
    rounded once), so logits and pool rows agree bit for bit."""
    from sglang_security_amd.decode_runner import LlamaConfig, LlamaDecoder

    cfg = LlamaConfig.tiny("bf16")
    decs = []
    for fused in (False, True):
        dec = LlamaDecoder(cfg, batch=6, pool_slots=1 + 64 + 6 * 24, max_ctx=96, device=DEV, seed=0)
        dec.setup_shared_prefix_batch(prefix_len=64, suffix_len=5)
        dec.enable_shared_prefix(True)
        assert dec.enable_fusion(fused) == fused
        assert dec.fused_dense == fused and not dec.fused
        decs.append(dec)
    ref, fus = decs
    for _ in range(3):
        ref.step()
        fus.step()
        torch.cuda.synchronize()
        assert torch.equal(ref.logits, fus.logits)
        assert torch.equal(ref.next_ids, fus.next_ids)
    assert torch.equal(ref.k_pool[1], fus.k_pool[1]) and torch.equal(ref.v_pool[1], fus.v_pool[1])
