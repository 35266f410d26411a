"""GPU parity: extend (prefill with cached prefix) attention and the shared-prefix decode path, through the C ABI.

Extend tolerances: against the fp32 Triton vector 2e-2 absolute on bf16; against the reference CPU kernel's bf16
output atol 3e-2 (test/srt/cpu/test_extend.py:186 uses atol=rtol=1e-2 on normal(0, 1)-sized outputs of a kernel that
differs only in summation order; our reference vector is the fp32 one).  The shared-prefix path must agree with
the ordinary decode kernel and the oracle to the decode tolerance (2e-2 bf16, 4e-3 fp16)."""
import math

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from sglang_security_amd import ops

    return ops


def _run_extend(ops, g, dtype, causal=True):
    d = lambda k: g[k].to(dtype).to(DEV)
    q = d("q_extend")
    o = torch.full((q.shape[0], q.shape[1], g["v_extend"].shape[2]), float("nan"), dtype=dtype, device=DEV)
    max_len = int((g["qo_indptr"][1:] - g["qo_indptr"][:-1]).max())
    ops.extend_attention_fwd(q, d("k_extend"), d("v_extend"), o, d("k_buffer"), d("v_buffer"), g["qo_indptr"].to(DEV),
                             g["kv_indptr"].to(DEV), g["kv_indices"].to(DEV), None, causal, None, max_len,
                             float(g["sm_scale"]), float(g["logit_cap"]))
    torch.cuda.synchronize()
    return o.float().cpu()


@pytest.mark.parametrize("case", ["extend_gqa128", "extend_mha64_cap"])
def test_extend_golden(golden, case):
    ops = _ops()
    g = golden(case)
    o = _run_extend(ops, g, torch.bfloat16)
    assert torch.isfinite(o).all()
    assert (o - g["expect_triton_f32"]).abs().max().item() < 2e-2
    torch.testing.assert_close(o, g["expect_cpu_bf16"].float(), atol=3e-2, rtol=1e-6)
    o32 = _run_extend(ops, g, torch.float32)
    assert (o32 - g["expect_triton_f32"]).abs().max().item() < 2e-5
    o16 = _run_extend(ops, g, torch.float16)
    assert (o16 - g["expect_triton_f32"]).abs().max().item() < 4e-3


def _extend_case(B, Hq, Hkv, D, prefix_hi, ext_hi, dtype, seed, min_ext=1):
    gen = torch.Generator().manual_seed(seed)
    pl = [int(torch.randint(0, prefix_hi + 1, (1,), generator=gen)) for _ in range(B)]
    el = [int(torch.randint(min_ext, ext_hi + 1, (1,), generator=gen)) for _ in range(B)]
    n_pool = sum(pl) + 5
    T = sum(el)
    g = dict(
        q_extend=torch.randn(T, Hq, D, generator=gen).to(dtype), k_extend=torch.randn(T, Hkv, D, generator=gen).to(dtype),
        v_extend=torch.randn(T, Hkv, D, generator=gen).to(dtype), k_buffer=torch.randn(n_pool, Hkv, D, generator=gen).to(dtype),
        v_buffer=torch.randn(n_pool, Hkv, D, generator=gen).to(dtype), sm_scale=1.0 / math.sqrt(D), logit_cap=0.0)
    perm = torch.randperm(n_pool - 1, generator=gen)[: sum(pl)] + 1
    g["kv_indices"] = perm.to(torch.int32)
    g["kv_indptr"] = torch.tensor([0] + list(torch.tensor(pl).cumsum(0)), dtype=torch.int32)
    g["qo_indptr"] = torch.tensor([0] + list(torch.tensor(el).cumsum(0)), dtype=torch.int32)
    return g


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("B,Hq,Hkv,prefix_hi,ext_hi,causal", [
    (3, 8, 2, 70, 150, True),    # several 128-row blocks, ragged, GQA 4
    (2, 4, 4, 0, 33, True),      # no prefix at all
    (4, 16, 1, 40, 20, True),    # MQA, extends shorter than one tile
    (2, 8, 2, 50, 140, False),   # non-causal (encoder-style)
    (1, 32, 8, 300, 1, True),    # single new token per sequence: decode through the extend path
])
def test_extend_vs_oracle(dtype, B, Hq, Hkv, prefix_hi, ext_hi, causal):
    import oracle

    ops = _ops()
    g = _extend_case(B, Hq, Hkv, 128, prefix_hi, ext_hi, dtype, seed=B * 100 + ext_hi)
    want = oracle.extend_attention(g["q_extend"], g["k_extend"], g["v_extend"], g["k_buffer"], g["v_buffer"],
                                   g["qo_indptr"], g["kv_indptr"], g["kv_indices"], g["sm_scale"], 0.0, causal)
    o = _run_extend(ops, g, dtype, causal)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol


def test_extend_generic_odd_head():
    """Head size 80 / value size 48 in fp32 and bf16: the wave-per-row kernel."""
    import oracle

    ops = _ops()
    gen = torch.Generator().manual_seed(5)
    g = _extend_case(2, 6, 3, 80, 30, 40, torch.float32, seed=9)
    T, n_pool = g["v_extend"].shape[0], g["v_buffer"].shape[0]
    g["v_extend"] = torch.randn(T, 3, 48, generator=gen)
    g["v_buffer"] = torch.randn(n_pool, 3, 48, generator=gen)
    want = oracle.extend_attention(g["q_extend"], g["k_extend"], g["v_extend"], g["k_buffer"], g["v_buffer"],
                                   g["qo_indptr"], g["kv_indptr"], g["kv_indices"], g["sm_scale"], 0.0, True)
    o = _run_extend(ops, g, torch.float32)
    assert (o - want).abs().max().item() < 2e-5


def test_extend_full_size_properties():
    """Workload-sized prefill (8 sequences x 128 new tokens over a 1024-token cached prefix, Llama-3-8B heads):
    (1) a sample of rows against the oracle; (2) the last row of each sequence equals decode attention over the
    same tokens; (3) rows of two sequences with identical inputs are identical."""
    import oracle

    ops = _ops()
    B, Hq, Hkv, D, P, E = 8, 32, 8, 128, 1024, 128
    gen = torch.Generator().manual_seed(3)
    dt = torch.bfloat16
    kb = torch.randn(1 + P, Hkv, D, generator=gen).to(dt)
    vb = torch.randn(1 + P, Hkv, D, generator=gen).to(dt)
    g = dict(q_extend=torch.randn(B * E, Hq, D, generator=gen).to(dt), k_extend=torch.randn(B * E, Hkv, D, generator=gen).to(dt),
             v_extend=torch.randn(B * E, Hkv, D, generator=gen).to(dt), k_buffer=kb, v_buffer=vb,
             kv_indices=(torch.arange(P, dtype=torch.int32) + 1).repeat(B),
             kv_indptr=torch.arange(B + 1, dtype=torch.int32) * P, qo_indptr=torch.arange(B + 1, dtype=torch.int32) * E,
             sm_scale=1.0 / math.sqrt(D), logit_cap=0.0)
    for k in ("q_extend", "k_extend", "v_extend"):
        g[k][E:2 * E] = g[k][:E]
    o = _run_extend(ops, g, dt)
    assert torch.isfinite(o).all()
    assert torch.equal(o[:E], o[E:2 * E])
    # sample: sequence 5 alone through the oracle
    s = 5
    sub = dict(g)
    for k in ("q_extend", "k_extend", "v_extend"):
        sub[k] = g[k][s * E:(s + 1) * E]
    want = oracle.extend_attention(sub["q_extend"], sub["k_extend"], sub["v_extend"], kb, vb,
                                   torch.tensor([0, E], dtype=torch.int32), torch.tensor([0, P], dtype=torch.int32),
                                   g["kv_indices"][:P], g["sm_scale"], 0.0, True)
    assert (o[s * E:(s + 1) * E] - want).abs().max().item() < 2e-2


# ------------------------------------------------------------------------------------------ shared-prefix decode
def _group_batch(groups, Hq, Hkv, dtype, seed):
    """groups: list of (n_requests, prefix_len, tail_lo, tail_hi)."""
    gen = torch.Generator().manual_seed(seed)
    D = 128
    rows, lens, gi, pls = [], [], [0], []
    nxt = 1
    for (n, p, lo, hi) in groups:
        pref = list(range(nxt, nxt + p))
        nxt += p
        for _ in range(n):
            t = int(torch.randint(lo, hi + 1, (1,), generator=gen))
            rows.append(pref + list(range(nxt, nxt + t)))
            nxt += t
            lens.append(p + t)
        gi.append(gi[-1] + n)
        pls.append(p)
    B = len(rows)
    kb = torch.randn(nxt + 2, Hkv, D, generator=gen).to(dtype)
    vb = torch.randn(nxt + 2, Hkv, D, generator=gen).to(dtype)
    q = torch.randn(B, Hq, D, generator=gen).to(dtype)
    ind = torch.tensor([s for r in rows for s in r], dtype=torch.int32)
    ptr = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    return q, kb, vb, ptr, ind, torch.tensor(gi, dtype=torch.int32), torch.tensor(pls, dtype=torch.int32)


def _run_shared(ops, q, kb, vb, ptr, ind, gi, pls, max_splits, dtype, sm, cap=0.0):
    B, Hq, D = q.shape
    o = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
    ws = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    args = (q.to(DEV), kb.to(DEV), vb.to(DEV))
    meta = (ptr.to(DEV), ind.to(DEV), gi.to(DEV), pls.to(DEV))
    ops.decode_attention_shared_prefix(*args, o, *meta, ws, max_splits, sm, cap)
    torch.cuda.synchronize()
    o2 = torch.full_like(o, float("nan"))  # the workspace carries nothing from one call to the next
    ops.decode_attention_shared_prefix(*args, o2, *meta, ws, max_splits, sm, cap)
    torch.cuda.synchronize()
    assert torch.equal(o.view(torch.int16), o2.view(torch.int16))
    return o.float().cpu()


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("groups,Hq,Hkv,max_splits", [
    ([(5, 96, 1, 40)], 32, 8, 4),                       # one group, ragged tails
    ([(3, 200, 1, 30), (1, 0, 5, 90), (7, 33, 1, 300)], 32, 8, 8),  # three groups; one without prefix; long tails
    ([(40, 64, 1, 10)], 8, 8, 2),                       # MHA, more columns than one block holds
    ([(2, 1, 1, 1), (2, 700, 600, 600)], 16, 2, 16),    # prefix of one token; tails longer than four splits' worth
    ([(21, 150, 1, 70), (3, 40, 1, 5)], 28, 4, 5),      # 7 heads per kv head: requests straddle column blocks
    ([(3, 70, 1, 40)], 32, 1, 3),                       # 32 heads per kv head: two head tiles
    ([(4, 64, 0, 0), (2, 0, 1, 3)], 8, 2, 2),           # empty remainders beside a group without a prefix
])
def test_shared_prefix_decode_vs_oracle(dtype, groups, Hq, Hkv, max_splits):
    import oracle

    ops = _ops()
    q, kb, vb, ptr, ind, gi, pls = _group_batch(groups, Hq, Hkv, dtype, seed=len(groups) + Hq)
    sm = 1.0 / math.sqrt(128)
    B = q.shape[0]
    want = oracle.decode_attention(q, kb, vb, ptr, ind, torch.ones(B, dtype=torch.int32), sm)
    o = _run_shared(ops, q, kb, vb, ptr, ind, gi, pls, max_splits, dtype, sm)
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol


def _run_planned(ops, q, kb, vb, ptr, ind, plan, max_splits, dtype, sm, cap=0.0, num_kv_splits=None):
    B, Hq, D = q.shape
    o = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
    ws = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    ops.decode_attention_shared_planned(q.to(DEV), kb.to(DEV), vb.to(DEV), o, ptr.to(DEV), ind.to(DEV), plan, ws, max_splits,
                                        sm, cap, None, num_kv_splits)
    torch.cuda.synchronize()
    assert ops.decode_shared_workspace_status(ws, B, Hq) == 0
    return o


@pytest.mark.parametrize("groups,Hq,Hkv,max_splits", [
    ([(5, 96, 1, 40)], 32, 8, 4),
    ([(3, 200, 1, 30), (1, 0, 5, 90), (7, 33, 1, 300)], 32, 8, 8),
    ([(40, 64, 1, 10)], 8, 8, 2),
    ([(2, 1, 1, 1), (2, 700, 600, 600)], 16, 2, 16),
    ([(21, 150, 1, 70), (3, 40, 1, 5)], 28, 4, 5),
    ([(3, 70, 1, 40)], 32, 1, 3),
    ([(4, 64, 0, 0), (2, 0, 1, 3)], 8, 2, 2),
    ([(2, 40, 1, 9)] * 23, 32, 8, 8),                   # many small groups: more prefix items than one round of blocks
])
def test_shared_prefix_planned_from_groups_equals_the_unplanned_kernel(groups, Hq, Hkv, max_splits):
    """The plan laid out from the caller's groups drives the same blocks over the same ranges as the kernel that reads
    the group arrays itself: bit-identical outputs (where no remainder is long enough to span tail blocks)."""
    ops = _ops()
    dtype = torch.bfloat16
    q, kb, vb, ptr, ind, gi, pls = _group_batch(groups, Hq, Hkv, dtype, seed=len(groups) + Hq)
    sm = 1.0 / math.sqrt(128)
    B = q.shape[0]
    want = _run_shared(ops, q, kb, vb, ptr, ind, gi, pls, max_splits, dtype, sm)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV)
    ops.decode_shared_plan_from_groups(ptr.to(DEV), ind.to(DEV), gi.to(DEV), pls.to(DEV), B, Hq, Hkv, max_splits, plan)
    got = _run_planned(ops, q, kb, vb, ptr, ind, plan, max_splits, dtype, sm)
    view = ops.decode_shared_plan_read(plan, B)
    assert view["num_groups"] == len(groups) and view["order"] == list(range(B))
    assert view["group_indptr"] == gi.tolist() and view["prefix_lens"] == pls.tolist()
    if max(hi for (_, _, _, hi) in groups) <= 512:
        assert torch.equal(got.cpu().view(torch.int16), want.to(dtype).view(torch.int16))
    else:
        # remainders of more than 512 tokens span several tail blocks under a plan (more, shorter splits than the
        # kernel that reads the group arrays gives them): the same sums in another order
        assert (got.float().cpu() - want).abs().max().item() < 1e-2


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("seed", [0, 1])
def test_shared_prefix_plan_derived_from_the_request_table(dtype, seed):
    """Groups found on the device from req_to_token rows: members scattered over the batch, ragged lengths, a group
    whose common part is shorter than the shared tree node (lists that diverge early), requests that share nothing, a
    request of one token (a padded graph row).  Against the oracle, and the grouping itself against a host restatement."""
    import oracle
    import random

    ops = _ops()
    rng = random.Random(seed)
    Hq, Hkv, D, max_splits = 32, 8, 128, 8
    specs = [(6, 130), (1, 0), (9, 64), (3, 33), (2, 20), (1, 0), (4, 257)]   # (members, shared slots)
    nxt = 1
    rows = []
    for members, shared in specs:
        base = list(range(nxt, nxt + shared))
        nxt += shared
        for m in range(members):
            tail = rng.randrange(1, 90)
            rows.append(base + list(range(nxt, nxt + tail)))
            nxt += tail
    rows.append([nxt])          # one token only
    nxt += 1
    rng.shuffle(rows)
    B = len(rows)
    lens = [len(r) for r in rows]
    width = max(lens) + 4
    r2t = torch.zeros(B + 3, width, dtype=torch.int32)
    rpi = torch.tensor(rng.sample(range(B + 3), B), dtype=torch.int64)        # rows of the table in any order
    for b, r in enumerate(rows):
        r2t[rpi[b], : len(r)] = torch.tensor(r, dtype=torch.int32)
    gen = torch.Generator().manual_seed(seed)
    kb = torch.randn(nxt + 1, Hkv, D, generator=gen).to(dtype)
    vb = torch.randn(nxt + 1, Hkv, D, generator=gen).to(dtype)
    q = torch.randn(B, Hq, D, generator=gen).to(dtype)
    seq = torch.tensor(lens, dtype=torch.int64)
    ptr = torch.zeros(B + 1, dtype=torch.int32, device=DEV)
    ind = torch.zeros(sum(lens), dtype=torch.int32, device=DEV)
    ns = torch.zeros(B, dtype=torch.int32, device=DEV)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV)
    ops.decode_metadata_shared(r2t.to(DEV), rpi.to(DEV), seq.to(DEV), ptr, ind, ns, Hq, Hkv, max_splits, 256, plan)
    torch.cuda.synchronize()
    # host restatement of the grouping rule: leader = first request with the same first slot (lists of two tokens and
    # more); shared length = the longest slot prefix common to the leader and every member, newest token excluded;
    # fewer than two members or fewer than 32 shared slots: nothing shared
    leader = []
    for b in range(B):
        lead = b
        if lens[b] >= 2:
            for i in range(b):
                if lens[i] >= 2 and rows[i][0] == rows[b][0]:
                    lead = i
                    break
        leader.append(lead)
    leaders = sorted(set(leader))
    want_order, want_indptr, want_plen = [], [0], []
    for L in leaders:
        members = [b for b in range(B) if leader[b] == L]
        plen = lens[L] - 1
        for b in members:
            n = min(lens[L], lens[b]) - 1
            k = 0
            while k < n and rows[L][k] == rows[b][k]:
                k += 1
            plen = min(plen, k)
        if len(members) < 2 or plen < 32:
            plen = 0
        want_order += members
        want_indptr.append(len(want_order))
        want_plen.append(plen)
    view = ops.decode_shared_plan_read(plan, B)
    assert view["num_groups"] == len(leaders)
    assert view["order"] == want_order and view["group_indptr"] == want_indptr and view["prefix_lens"] == want_plen
    assert any(p > 0 for p in want_plen) and view["n_items"] > 0
    assert torch.equal(ind.cpu(), torch.tensor([s for r in rows for s in r], dtype=torch.int32))
    sm = 1.0 / math.sqrt(D)
    want = oracle.decode_attention(q, kb, vb, ptr.cpu(), ind.cpu(), torch.ones(B, dtype=torch.int32), sm)
    o = _run_planned(ops, q, kb, vb, ptr, ind, plan, max_splits, dtype, sm).float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol
    # the same launch sequence with the per-request kernels beside it: the plan has items, so they leave at once
    o_g = _run_planned(ops, q, kb, vb, ptr, ind, plan, max_splits, dtype, sm, num_kv_splits=ns).float().cpu()
    assert torch.equal(o_g, o)


def test_shared_prefix_decode_golden(golden):
    """The committed shared-prefix decode vector (Triton fp32 / reference CPU bf16), one group of 4 requests."""
    ops = _ops()
    g = golden("decode_shared")
    seq = g["seq_lens"]
    r2t = g["req_to_token"]
    B = len(seq)
    common = 0
    while all(int(r2t[b, common]) == int(r2t[0, common]) for b in range(B)) and common < int(seq.min()) - 1:
        common += 1
    assert common > 0
    gi = torch.tensor([0, B], dtype=torch.int32)
    pls = torch.tensor([common], dtype=torch.int32)
    o = _run_shared(ops, g["q"].to(torch.bfloat16), g["k_buffer"].to(torch.bfloat16), g["v_buffer"].to(torch.bfloat16),
                    g["kv_indptr"], g["kv_indices"], gi, pls, 8, torch.bfloat16, float(g["sm_scale"]),
                    float(g["logit_cap"]))
    assert (o - g["expect_triton_f32"]).abs().max().item() < 2e-2
    torch.testing.assert_close(o, g["expect_cpu_bf16"].float(), atol=3e-2, rtol=1e-6)


def test_shared_prefix_full_size_matches_per_request_kernel():
    """Workload size (batch 64, 1024 shared + 128 private): the shared-prefix path against the per-request kernel
    on the same inputs, plus the device-side common-prefix helper."""
    ops = _ops()
    B, Hq, Hkv, D = 64, 32, 8, 128
    q, kb, vb, ptr, ind, gi, pls = _group_batch([(B, 1024, 128, 128)], Hq, Hkv, torch.bfloat16, seed=11)
    sm = 1.0 / math.sqrt(D)
    o = _run_shared(ops, q, kb, vb, ptr, ind, gi, pls, 8, torch.bfloat16, sm)
    o2 = torch.empty(B, Hq, D, dtype=torch.bfloat16, device=DEV)
    al = torch.empty(B, Hq, 8, D, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, 8, dtype=torch.float32, device=DEV)
    ns = torch.full((B,), 8, dtype=torch.int32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o2, ptr.to(DEV), ind.to(DEV), al, lse, ns, 8, sm, 0.0)
    assert (o - o2.float().cpu()).abs().max().item() < 1e-2
    # common prefix helper over a request table built from the same rows
    r2t = torch.zeros(B, 1160, dtype=torch.int32)
    for b in range(B):
        r2t[b, :1152] = ind[int(ptr[b]):int(ptr[b + 1])]
    out = torch.zeros(1, dtype=torch.int32, device=DEV)
    ops.common_prefix_len(r2t.to(DEV), torch.arange(B, device=DEV), torch.full((B,), 1152, device=DEV), out)
    assert int(out.item()) == 1024


@pytest.mark.parametrize("dtype,Hq,Hkv,D,Dv", [(torch.bfloat16, 16, 1, 576, 512), (torch.bfloat16, 14, 2, 64, 64),
                                              (torch.float32, 8, 2, 128, 128), (torch.float16, 4, 4, 96, 96)])
def test_shared_prefix_other_shapes(dtype, Hq, Hkv, D, Dv):
    """Shapes without a tile kernel (absorbed MLA 576 / 512 with the values a view of the key rows, head size 64, fp32)
    go through the shared-prefix entry point too: the library runs the per-request kernels over the same lists, so the
    result is the per-request result (against the oracle, and bit for bit against decode_attention_fwd with the same
    split counts).  Two groups with different prefixes, ragged tails, two calls on one workspace."""
    import oracle

    ops = _ops()
    gen = torch.Generator().manual_seed(Hq * 100 + D)
    groups = [(3, 70), (2, 33)]  # (requests, shared prefix length)
    tails = [5, 40, 1, 17, 64]
    n = 1 + sum(p for _, p in groups) + sum(tails)
    kb = torch.randn(n, Hkv, D, generator=gen).to(dtype)
    mla = D == 576
    vb = kb[..., :Dv] if mla else torch.randn(n, Hkv, Dv, generator=gen).to(dtype)
    B = sum(r for r, _ in groups)
    q = (torch.randn(B, Hq, D, generator=gen) * 0.5).to(dtype)
    rows, nxt, gi, pl, b = [], 1, [0], [], 0
    for r, p in groups:
        shared = list(range(nxt, nxt + p))
        nxt += p
        for _ in range(r):
            rows.append(shared + list(range(nxt, nxt + tails[b])))
            nxt += tails[b]
            b += 1
        gi.append(b)
        pl.append(p)
    lens = [len(r) for r in rows]
    ind = torch.tensor([s for r in rows for s in r], dtype=torch.int32)
    ptr = torch.tensor([0] + list(torch.tensor(lens).cumsum(0)), dtype=torch.int32)
    S = 4
    sm = 1.0 / math.sqrt(192 if mla else D)
    ns = torch.full((B,), S, dtype=torch.int32)
    want = oracle.decode_attention(q, kb, vb.contiguous(), ptr, ind, ns, sm)
    kd = kb.to(DEV)
    vd = kd[..., :Dv] if mla else vb.to(DEV)
    ws = ops.decode_shared_workspace(B, Hq, Dv, S, DEV)
    for _ in range(2):
        o = torch.full((B, Hq, Dv), float("nan"), dtype=dtype, device=DEV)
        ops.decode_attention_shared_prefix(q.to(DEV), kd, vd, o, ptr.to(DEV), ind.to(DEV),
                                           torch.tensor(gi, dtype=torch.int32, device=DEV),
                                           torch.tensor(pl, dtype=torch.int32, device=DEV), ws, S, sm)
        torch.cuda.synchronize()
        tol = 2e-2 if dtype == torch.bfloat16 else 4e-3 if dtype == torch.float16 else 2e-4
        assert (o.float().cpu() - want.float()).abs().max().item() < tol
    o2 = torch.empty_like(o)
    al = torch.empty(B, Hq, S, Dv, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, S, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kd, vd, o2, ptr.to(DEV), ind.to(DEV), al, lse, ns.to(DEV), S, sm, 0.0)
    assert torch.equal(o, o2)
    assert ops.decode_shared_workspace_status(ws, B, Hq) == 0


def _group_batch_d(groups, Hq, Hkv, D, dtype, seed):
    """_group_batch for another head size."""
    q, kb, vb, ptr, ind, gi, pls = _group_batch(groups, Hq, Hkv, dtype, seed)
    gen = torch.Generator().manual_seed(seed + 1000)
    n = kb.shape[0]
    return (torch.randn(q.shape[0], Hq, D, generator=gen).to(dtype), torch.randn(n, Hkv, D, generator=gen).to(dtype),
            torch.randn(n, Hkv, D, generator=gen).to(dtype), ptr, ind, gi, pls)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("groups,Hq,Hkv,max_splits", [
    ([(6, 130, 1, 50), (1, 0, 3, 9), (5, 64, 1, 200)], 14, 2, 8),      # Qwen2-0.5B heads: 7 per kv head, one pair
    ([(40, 96, 1, 20)], 16, 4, 4),                                     # two pairs, more columns than one block
    ([(3, 70, 1, 40), (2, 33, 0, 0)], 64, 2, 3),                       # 64 heads per pair: four head tiles
])
def test_shared_prefix_planned_head_size_64(dtype, groups, Hq, Hkv, max_splits):
    """Head size 64 on the one-launch kernel: pairs of kv heads walked as one 128-wide head, the plan laid out for the
    pairs (ops.plan_kv_heads).  Against the oracle and against the per-request kernels on the same lists."""
    import oracle

    ops = _ops()
    D = 64
    q, kb, vb, ptr, ind, gi, pls = _group_batch_d(groups, Hq, Hkv, D, dtype, seed=Hq + Hkv)
    B = q.shape[0]
    sm = 1.0 / math.sqrt(D)
    want = oracle.decode_attention(q, kb, vb, ptr, ind, torch.ones(B, dtype=torch.int32), sm)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV, head_dim=D)
    ops.decode_shared_plan_from_groups(ptr.to(DEV), ind.to(DEV), gi.to(DEV), pls.to(DEV), B, Hq, Hkv, max_splits, plan,
                                       head_dim=D)
    assert ops.decode_shared_plan_read(plan, B)["n_items"] > 0
    o = _run_planned(ops, q, kb, vb, ptr, ind, plan, max_splits, dtype, sm).float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol
    o2 = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
    al = torch.empty(B, Hq, max_splits, D, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, max_splits, dtype=torch.float32, device=DEV)
    ns = torch.full((B,), max_splits, dtype=torch.int32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kb.to(DEV), vb.to(DEV), o2, ptr.to(DEV), ind.to(DEV), al, lse, ns, max_splits, sm, 0.0)
    assert (o - o2.float().cpu()).abs().max().item() < tol
    # with the per-request kernels beside it (the form the backend launches): the plan has items, they leave at once
    o_g = _run_planned(ops, q, kb, vb, ptr, ind, plan, max_splits, dtype, sm, num_kv_splits=ns).float().cpu()
    assert torch.equal(o_g, o)


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("groups,Hq,max_splits,cap", [
    ([(6, 130, 1, 50), (1, 0, 3, 9), (5, 64, 1, 200)], 16, 8, 0.0),     # a TP = 8 rank's 16 heads
    ([(3, 70, 1, 40), (2, 33, 0, 0)], 128, 4, 0.0),                      # all 128 heads: eight tiles per request
    ([(9, 96, 1, 20)], 8, 4, 30.0),                                      # fewer heads than a tile; logit cap
    ([(2, 40, 1, 9)] * 19, 16, 8, 0.0),                                  # many small groups: items over several rounds
])
def test_shared_prefix_planned_absorbed_mla(dtype, groups, Hq, max_splits, cap):
    """Absorbed MLA (latent rows of 576, values = their first 512 elements) through the plan: a group's shared rows
    are multiplied once per 32 (request, head) columns, whatever request the columns belong to.  Against the oracle,
    against the per-request kernels, and with those launched beside it (the form the backend uses)."""
    import oracle

    ops = _ops()
    D, Dv = 576, 512
    q, kb, _, ptr, ind, gi, pls = _group_batch_d(groups, Hq, 1, D, dtype, seed=Hq + len(groups))
    q = (q.float() * 0.5).to(dtype)
    B = q.shape[0]
    sm = 1.0 / math.sqrt(192)
    want = oracle.decode_attention(q, kb, kb[..., :Dv].contiguous(), ptr, ind, torch.ones(B, dtype=torch.int32), sm, cap)
    kd = kb.to(DEV)
    vd = kd[..., :Dv]
    plan = ops.decode_shared_plan_buffer(B, Hq, 1, max_splits, DEV)
    ops.decode_shared_plan_from_groups(ptr.to(DEV), ind.to(DEV), gi.to(DEV), pls.to(DEV), B, Hq, 1, max_splits, plan)
    assert ops.decode_shared_plan_read(plan, B)["n_items"] > 0
    ws = ops.decode_shared_workspace(B, Hq, Dv, max_splits, DEV)
    ns = torch.full((B,), max_splits, dtype=torch.int32, device=DEV)
    outs = []
    for gate in (None, ns, None):
        o = torch.full((B, Hq, Dv), float("nan"), dtype=dtype, device=DEV)
        ops.decode_attention_shared_planned(q.to(DEV), kd, vd, o, ptr.to(DEV), ind.to(DEV), plan, ws, max_splits, sm, cap,
                                            None, gate)
        torch.cuda.synchronize()
        assert ops.decode_shared_workspace_status(ws, B, Hq) == 0
        outs.append(o)
    assert torch.equal(outs[0], outs[1]) and torch.equal(outs[0], outs[2])
    o = outs[0].float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want.float()).abs().max().item() < tol
    o2 = torch.empty(B, Hq, Dv, dtype=dtype, device=DEV)
    al = torch.empty(B, Hq, max_splits, Dv, dtype=torch.float32, device=DEV)
    lse = torch.empty(B, Hq, max_splits, dtype=torch.float32, device=DEV)
    ops.decode_attention_fwd(q.to(DEV), kd, vd, o2, ptr.to(DEV), ind.to(DEV), al, lse, ns, max_splits, sm, cap)
    assert (o - o2.float().cpu()).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("D,Hq,Hkv,max_splits", [(128, 32, 8, 16), (128, 8, 2, 8), (64, 14, 2, 16), (128, 32, 8, 4)])
def test_shared_prefix_planned_long_remainders_vs_oracle(dtype, D, Hq, Hkv, max_splits):
    """Remainders of more than 512 tokens are spread over several tail blocks (up to max_splits / 4 of them) whose
    partial results the last one to finish merges with the group's prefix partials: requests that share nothing next
    to a group with long private parts, a length that ends exactly on a block's boundary, short requests in the same
    launch.  Against the oracle; a second launch on the same workspace gives the same bits (the tickets are back at
    zero, the merge order does not depend on which block came last)."""
    import oracle

    ops = _ops()
    groups = [(3, 100, 520, 1500), (1, 0, 2048, 2048), (2, 0, 1, 40), (4, 64, 1000, 1100), (1, 0, 513, 513),
              (1, 0, 1024, 1024)]
    q, kb, vb, ptr, ind, gi, pls = _group_batch_d(groups, Hq, Hkv, D, dtype, seed=D + Hq)
    B = q.shape[0]
    sm = 1.0 / math.sqrt(D)
    want = oracle.decode_attention(q, kb, vb, ptr, ind, torch.ones(B, dtype=torch.int32), sm)
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, max_splits, DEV, head_dim=D)
    ops.decode_shared_plan_from_groups(ptr.to(DEV), ind.to(DEV), gi.to(DEV), pls.to(DEV), B, Hq, Hkv, max_splits, plan,
                                       head_dim=D)
    ws = ops.decode_shared_workspace(B, Hq, D, max_splits, DEV)
    dev = (q.to(DEV), kb.to(DEV), vb.to(DEV))
    outs = []
    for _ in range(3):
        o = torch.full((B, Hq, D), float("nan"), dtype=dtype, device=DEV)
        ops.decode_attention_shared_planned(*dev, o, ptr.to(DEV), ind.to(DEV), plan, ws, max_splits, sm, 0.0, None, None)
        torch.cuda.synchronize()
        assert ops.decode_shared_workspace_status(ws, B, Hq) == 0
        outs.append(o)
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16))
    assert torch.equal(outs[0].view(torch.int16), outs[2].view(torch.int16))
    o = outs[0].float().cpu()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o).all()
    assert (o - want).abs().max().item() < tol


def test_shared_prefix_planned_rejects_a_plan_of_another_geometry():
    """A plan laid out for other heads per kv head must not be walked: NaN rows and the workspace's error word, not a
    wrong answer (here: a head-size-64 launch on a plan that counts single kv heads)."""
    ops = _ops()
    dtype, D, Hq, Hkv, S = torch.bfloat16, 64, 16, 4, 4
    q, kb, vb, ptr, ind, gi, pls = _group_batch_d([(6, 70, 1, 20)], Hq, Hkv, D, dtype, seed=5)
    B = q.shape[0]
    plan = ops.decode_shared_plan_buffer(B, Hq, Hkv, S, DEV)          # room and layout for 4 kv heads of 4 heads
    ops.decode_shared_plan_from_groups(ptr.to(DEV), ind.to(DEV), gi.to(DEV), pls.to(DEV), B, Hq, Hkv, S, plan)
    o = torch.zeros(B, Hq, D, dtype=dtype, device=DEV)
    ws = ops.decode_shared_workspace(B, Hq, D, S, DEV)
    ops.decode_attention_shared_planned(q.to(DEV), kb.to(DEV), vb.to(DEV), o, ptr.to(DEV), ind.to(DEV), plan, ws, S,
                                        1.0 / math.sqrt(D))
    torch.cuda.synchronize()
    assert torch.isnan(o.float()).all()
    assert ops.decode_shared_workspace_status(ws, B, Hq) == 1


# ------------------------------------------------------------------------------------------ aiter call surface
@pytest.mark.parametrize("dtype,D", [(torch.bfloat16, 128), (torch.float16, 128), (torch.float32, 64)])
@pytest.mark.parametrize("causal", [True, False])
def test_aiter_shim_prefill_and_decode(dtype, D, causal):
    """The two entry points of the reference's ROCm default backend (aiter_backend.py:353-417): paged prefill with the
    new tokens already in the pool, and ragged paged decode, against the oracle."""
    import oracle
    import sglang_security_amd.aiter_shim as shim

    Hq, Hkv, B = 8, 2, 3
    gen = torch.Generator().manual_seed(17 + D)
    pre, ext = [40, 0, 150], [33, 70, 5]
    n = 1 + sum(pre) + sum(ext)
    kb = torch.randn(n, 1, Hkv, D, generator=gen).to(dtype)
    vb = torch.randn(n, 1, Hkv, D, generator=gen).to(dtype)
    perm = (torch.randperm(n - 1, generator=gen) + 1).to(torch.int32)
    rows, at = [], 0
    for p, e in zip(pre, ext):
        rows.append(perm[at:at + p + e])
        at += p + e
    T = sum(ext)
    q = torch.randn(T, Hq, D, generator=gen).to(dtype)
    qo = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32)
    kvp = torch.tensor([0] + list(torch.tensor([p + e for p, e in zip(pre, ext)]).cumsum(0)), dtype=torch.int32)
    ki = torch.cat(rows)
    o = shim.mha_batch_prefill_func(q.to(DEV), kb.to(DEV), vb.to(DEV), qo.to(DEV), kvp.to(DEV), ki.to(DEV), max(ext),
                                    max(p + e for p, e in zip(pre, ext)), causal=causal, logits_soft_cap=0.0)
    # oracle: the same thing stated as prefix (through indices) + new tokens (contiguous, gathered from the pool)
    k_new = torch.cat([kb[r[p:].long(), 0] for r, p in zip(rows, pre)])
    v_new = torch.cat([vb[r[p:].long(), 0] for r, p in zip(rows, pre)])
    pref_ptr = torch.tensor([0] + list(torch.tensor(pre).cumsum(0)), dtype=torch.int32)
    pref_idx = torch.cat([r[:p] for r, p in zip(rows, pre)])
    want = oracle.extend_attention(q, k_new, v_new, kb[:, 0], vb[:, 0], qo, pref_ptr, pref_idx, D ** -0.5, 0.0, causal)
    tol = {torch.bfloat16: 2e-2, torch.float16: 4e-3, torch.float32: 2e-5}[dtype]
    assert (o.float().cpu() - want).abs().max().item() < tol
    if causal:
        # decode of the last token of every sequence == last prefill row
        qd = torch.stack([q[int(qo[b + 1]) - 1] for b in range(B)])
        od = torch.empty(B, Hq, D, dtype=dtype, device=DEV)
        shim.paged_attention_ragged(od, None, qd.to(DEV), kb.to(DEV), vb.to(DEV), D ** -0.5, kvp.to(DEV), ki.to(DEV),
                                    torch.ones(B, dtype=torch.int32, device=DEV), 1, 8, None, "auto", "NHD", 0.0, None, None,
                                    None, 256)
        last = torch.stack([want[int(qo[b + 1]) - 1] for b in range(B)])
        assert (od.float().cpu() - last).abs().max().item() < tol


@pytest.mark.parametrize("case", ["tree", "prefix_masked"])
def test_extend_masked_golden(golden, case):
    """Custom-mask extend (draft-tree verification) against the reference Triton kernel's output."""
    from sglang_security_amd import ops

    g = golden("extend_masked_" + case)
    dev = "cuda"
    q = g["q_extend"].to(dev)
    o = torch.empty(q.shape[0], q.shape[1], g["v_extend"].shape[2], dtype=q.dtype, device=dev)
    ext = (g["qo_indptr"][1:] - g["qo_indptr"][:-1]).max().item()
    ops.extend_attention_fwd(q, g["k_extend"].to(dev), g["v_extend"].to(dev), o, g["k_buffer"].to(dev), g["v_buffer"].to(dev),
                             g["qo_indptr"].to(dev), g["kv_indptr"].to(dev), g["kv_indices"].to(dev),
                             g["custom_mask"].to(dev), True, g["mask_indptr"].to(dev), ext, g["sm_scale"], 0.0,
                             skip_prefix_custom_mask=bool(g["skip_prefix_custom_mask"]))
    torch.testing.assert_close(o.float().cpu(), g["expect_triton_f32"], rtol=2e-2, atol=2e-2)   # bf16 output


@pytest.mark.parametrize("case", ["w8", "w4_h64"])
@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float32])
def test_extend_window_golden(golden, case, dtype):
    """extend_attention_fwd(sliding_window_size=W) against the reference Triton kernel's output (the rows it gets
    finite: see test_oracle_extend_window_golden) and against the oracle on every row."""
    import oracle

    ops = _ops()
    g = golden("extend_window_" + case)
    W = int(g["window"])
    d = lambda t: t.to(dtype).to(DEV)
    q = d(g["q_extend"])
    o = torch.full((q.shape[0], q.shape[1], g["v_extend"].shape[-1]), float("nan"), dtype=dtype, device=DEV)
    ext = (g["qo_indptr"][1:] - g["qo_indptr"][:-1]).max().item()
    ops.extend_attention_fwd(q, d(g["k_extend"]), d(g["v_extend"]), o, d(g["k_buffer"]), d(g["v_buffer"]),
                             g["qo_indptr"].to(DEV), g["kv_indptr"].to(DEV), g["kv_indices"].to(DEV), None, True, None,
                             ext, g["sm_scale"], 0.0, sliding_window_size=W)
    torch.cuda.synchronize()
    got = o.float().cpu()
    assert torch.isfinite(got).all()
    want = g["expect_triton_f32"]
    ok = ~torch.isnan(want).any(-1).any(-1)
    tol = 2e-2 if dtype == torch.bfloat16 else 2e-4
    assert (got[ok] - want[ok]).abs().max().item() < tol
    ref = oracle.extend_attention_window(g["q_extend"], g["k_extend"], g["v_extend"], g["k_buffer"], g["v_buffer"],
                                         g["qo_indptr"], g["kv_indptr"], g["kv_indices"], W, g["sm_scale"])
    assert (got - ref).abs().max().item() < tol
    # a window no row reaches beyond is no window
    o2 = torch.empty_like(o)
    ops.extend_attention_fwd(q, d(g["k_extend"]), d(g["v_extend"]), o2, d(g["k_buffer"]), d(g["v_buffer"]),
                             g["qo_indptr"].to(DEV), g["kv_indptr"].to(DEV), g["kv_indices"].to(DEV), None, True, None,
                             ext, g["sm_scale"], 0.0, sliding_window_size=10 ** 6)
    o3 = torch.empty_like(o)
    ops.extend_attention_fwd(q, d(g["k_extend"]), d(g["v_extend"]), o3, d(g["k_buffer"]), d(g["v_buffer"]),
                             g["qo_indptr"].to(DEV), g["kv_indptr"].to(DEV), g["kv_indices"].to(DEV), None, True, None,
                             ext, g["sm_scale"], 0.0)
    assert (o2.float() - o3.float()).abs().max().item() < tol


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("pre,ext,splits", [([1024], [128], 8), ([300, 0, 77], [40, 64, 5], 4), ([33], [1], 2),
                                            ([500, 500], [70, 130], 3)])
def test_extend_attention_split_kv(dtype, pre, ext, splits):
    """The cached prefix split over several blocks per (sequence, query block, head) plus the merge launch: against the
    oracle and against the plain kernel (same tolerance: only the order of the softmax merge differs).  Ragged
    prefixes incl. an empty one (its prefix splits contribute nothing) and a prefix shorter than the split count x 32."""
    import oracle

    ops = _ops()
    Hq, Hkv, D = 8, 2, 128
    B = len(pre)
    gen = torch.Generator().manual_seed(sum(pre) + sum(ext))
    n = sum(pre) + 5
    kb = torch.randn(n, Hkv, D, generator=gen).to(dtype)
    vb = torch.randn(n, Hkv, D, generator=gen).to(dtype)
    T = sum(ext)
    q = torch.randn(T, Hq, D, generator=gen).to(dtype)
    ke = torch.randn(T, Hkv, D, generator=gen).to(dtype)
    ve = torch.randn(T, Hkv, D, generator=gen).to(dtype)
    qo = torch.tensor([0] + list(torch.tensor(ext).cumsum(0)), dtype=torch.int32)
    kp = torch.tensor([0] + list(torch.tensor(pre).cumsum(0)), dtype=torch.int32)
    ki = (torch.randperm(n - 1, generator=gen)[: sum(pre)] + 1).to(torch.int32)
    sm = 1.0 / math.sqrt(D)
    want = oracle.extend_attention(q, ke, ve, kb, vb, qo, kp, ki, sm, 0.0, True)
    d = lambda t: t.to(DEV)
    o = torch.full((T, Hq, D), float("nan"), dtype=dtype, device=DEV)
    ops.extend_attention_split(d(q), d(ke), d(ve), o, d(kb), d(vb), d(qo), d(kp), d(ki), max(ext), sm, 0.0, True, splits)
    o1 = torch.empty_like(o)
    ops.extend_attention_split(d(q), d(ke), d(ve), o1, d(kb), d(vb), d(qo), d(kp), d(ki), max(ext), sm, 0.0, True, 1)
    torch.cuda.synchronize()
    tol = 2e-2 if dtype == torch.bfloat16 else 4e-3
    assert torch.isfinite(o.float()).all()
    assert (o.float().cpu() - want).abs().max().item() < tol
    assert (o.float() - o1.float()).abs().max().item() < tol


def test_extend_masked_tree_from_builder():
    """The mask laid out by build_tree_kernel_efficient drives the extend: every draft token attends the prefix and
    its own ancestors only, which the restatement reproduces from the same bytes."""
    import oracle
    from sglang_security_amd import ops
    import sglang_security_amd.sgl_kernel as sk

    dev = "cuda"
    bs, topk, depth, n, Hq, Hkv, D = 3, 4, 3, 8, 8, 2, 128
    gen = torch.Generator().manual_seed(3)
    score = [torch.rand(bs, 1, topk, generator=gen)]
    parents = [torch.arange(-1, topk).unsqueeze(0).repeat(bs, 1)]
    prev = score[0].squeeze(1)
    for step in range(1, depth):
        child = prev.unsqueeze(2) * torch.rand(bs, topk, topk, generator=gen)
        score.append(child)
        top = torch.topk(child.flatten(1), topk, dim=-1)
        parents.append(top.indices + (topk * topk * (step - 1) + topk))
        prev = top.values
    tokens = [torch.randint(0, 100, (bs, s.shape[1] * s.shape[2]), generator=gen) for s in score]
    parent_list, sel, _ = oracle.build_tree_preprocess(torch.zeros(bs, dtype=torch.long), score, tokens, parents, n)
    seq = torch.tensor([20, 7, 33], dtype=torch.int32)
    mask = torch.full((int(seq.sum()) * n + n * n * bs,), True, device=dev)
    pos = torch.empty(bs * n, dtype=torch.long, device=dev)
    ri, nt, ns = (torch.full((bs, n), -1, dtype=torch.long, device=dev) for _ in range(3))
    sk.build_tree_kernel_efficient(parent_list.to(dev), sel.to(dev), seq.to(dev), mask, pos, ri, nt, ns, topk, depth, n)
    total = int(seq.sum())
    kb = torch.randn(total + 1, Hkv, D, generator=gen).to(torch.bfloat16)
    vb = torch.randn(total + 1, Hkv, D, generator=gen).to(torch.bfloat16)
    q = torch.randn(bs * n, Hq, D, generator=gen).to(torch.bfloat16)
    ke = torch.randn(bs * n, Hkv, D, generator=gen).to(torch.bfloat16)
    ve = torch.randn(bs * n, Hkv, D, generator=gen).to(torch.bfloat16)
    qo = (torch.arange(bs + 1) * n).int()
    kp = torch.cat([torch.zeros(1, dtype=torch.int32), torch.cumsum(seq, 0).int()])
    ki = (torch.randperm(total, generator=gen) + 1).int()
    mptr = torch.cat([torch.zeros(1, dtype=torch.long), torch.cumsum((seq.long() + n) * n, 0)])
    o = torch.empty(bs * n, Hq, D, dtype=torch.bfloat16, device=dev)
    ops.extend_attention_fwd(q.to(dev), ke.to(dev), ve.to(dev), o, kb.to(dev), vb.to(dev), qo.to(dev), kp.to(dev),
                             ki.to(dev), mask, True, mptr.to(dev), n, 1.0 / math.sqrt(D))
    want = oracle.extend_attention_masked(q, ke, ve, kb, vb, qo, kp, ki, mask.cpu().to(torch.uint8), mptr, True,
                                          1.0 / math.sqrt(D))
    torch.testing.assert_close(o.float().cpu(), want, rtol=2e-2, atol=2e-2)
    # the root of every tree sees the prefix and itself only: equal to plain causal extend of one token
    m = mask.cpu().view(-1)
    assert m[mptr[0] + 20] and not m[mptr[0] + 21: mptr[0] + 20 + n].any()
