"""GPU parity of the MoE routing helpers and the batched fp8 matmul against the C oracle, on the shapes of the
reference's own kernel tests (sgl-kernel/tests/test_moe_align.py, test_moe_topk_softmax.py, test_bmm_fp8.py)."""
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _ops():
    from sglang_security_amd import ops
    return ops


@pytest.mark.parametrize("tokens,topk,E,block", [(1, 2, 8, 16), (64, 8, 256, 64), (7, 6, 64, 128), (333, 4, 60, 16),
                                                (4096, 8, 256, 128), (1100, 1, 512, 1), (16, 2, 4, 3)])
@pytest.mark.parametrize("idt", [torch.int32, torch.int64])
def test_moe_align_block_size(tokens, topk, E, block, idt):
    """Bit-exact against the C oracle (a restatement of moe_align_kernel.cu:43-100).  Parity unpinned for the order of
    slots inside an expert: the reference leaves it to atomics (its own test compares against a Triton kernel that
    cannot run here); both sides here use order of appearance, and the layout's defining properties are checked."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(tokens * 31 + E)
    ids = torch.stack([torch.randperm(E, generator=g)[:topk] for _ in range(tokens)]).to(idt)
    numel = ids.numel()
    want_sorted, want_experts, want_total = oracle.moe_align_block_size(ids, E, block)
    cap = numel + E * (block - 1)
    srt = torch.full((cap,), numel, dtype=torch.int32, device=DEV)
    ex = torch.full(((cap + block - 1) // block,), -1, dtype=torch.int32, device=DEV)
    tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    cs = torch.zeros(E + 1, dtype=torch.int32, device=DEV)
    ops.moe_align_block_size(ids.to(DEV), E, block, srt, ex, tot, None, cs)
    assert int(tot) == want_total
    assert torch.equal(srt.cpu(), want_sorted)                     # bit-exact, padding included
    assert torch.equal(ex.cpu(), want_experts)
    assert int(cs[E]) == want_total and int(cs[0]) == 0
    # every slot appears once, under its own expert, in order of appearance
    s = srt.cpu()[:want_total]
    real = s[s < numel]
    assert torch.equal(torch.sort(real).values, torch.arange(numel, dtype=torch.int32))
    flat = ids.flatten()
    for blk in range(want_total // block):
        seg = s[blk * block:(blk + 1) * block]
        seg = seg[seg < numel]
        assert (flat[seg.long()] == int(want_experts[blk])).all()


def test_moe_align_rejects_bad_arguments():
    ops = _ops()
    from sglang_security_amd._lib import SglkError
    ids = torch.zeros(4, 2, dtype=torch.int32, device=DEV)
    srt = torch.zeros(4, dtype=torch.int32, device=DEV)
    with pytest.raises(SglkError):
        ops.moe_align_block_size(ids, 8, 16, srt, srt, srt)
    big = torch.zeros(8 + 600 * 15, dtype=torch.int32, device=DEV)
    with pytest.raises(SglkError):
        ops.moe_align_block_size(ids, 600, 16, big, big, big)       # more experts than the kernel's counters hold


@pytest.mark.parametrize("tokens,E,topk", list(itertools.product([1, 16, 128, 1024], [4, 8, 64, 256, 384], [1, 2, 4])) +
                         [(64, 256, 8), (2048, 128, 6)])
def test_topk_softmax(tokens, E, topk):
    """Grid of test_moe_topk_softmax.py:8-17 (+ DeepSeek's 256 experts / top 8 and a non-power-of-two expert count)."""
    import oracle

    ops = _ops()
    g = torch.Generator().manual_seed(tokens + E + topk)
    gating = torch.randn(tokens, E, generator=g)
    w = torch.empty(tokens, topk, dtype=torch.float32, device=DEV)
    ids = torch.empty(tokens, topk, dtype=torch.int32, device=DEV)
    src = torch.empty(tokens, topk, dtype=torch.int32, device=DEV)
    ops.topk_softmax(w, ids, src, gating.to(DEV))
    want_w, want_ids, want_src = oracle.topk_softmax(gating, topk)
    assert torch.equal(ids.cpu(), want_ids)
    assert torch.equal(src.cpu(), want_src)
    torch.testing.assert_close(w.cpu(), want_w, rtol=1e-5, atol=1e-7)
    # the reference's own check: torch softmax + topk, 1e-3
    ref_w, ref_i = torch.topk(torch.softmax(gating, dim=-1), topk, dim=-1)
    assert torch.allclose(ref_w, w.cpu(), atol=1e-3, rtol=1e-3)
    assert torch.equal(ref_i.int(), ids.cpu())


@pytest.mark.parametrize("dtype", [torch.bfloat16, torch.float16])
def test_topk_softmax_half_inputs_and_ties(dtype):
    import oracle

    ops = _ops()
    gating = torch.tensor([[1.0, 3.0, 3.0, -2.0, 3.0, 0.5, 0.25, 0.125]] * 5).to(dtype)
    w = torch.empty(5, 3, dtype=torch.float32, device=DEV)
    ids = torch.empty(5, 3, dtype=torch.int32, device=DEV)
    src = torch.empty(5, 3, dtype=torch.int32, device=DEV)
    ops.topk_softmax(w, ids, src, gating.to(DEV))
    assert ids.cpu().tolist() == [[1, 2, 4]] * 5                  # ties go to the lowest expert id
    want_w, _, _ = oracle.topk_softmax(gating.float(), 3)
    torch.testing.assert_close(w.cpu(), want_w, rtol=1e-5, atol=1e-7)


def _to_float8(x, dtype):
    finfo = torch.finfo(dtype)
    amax = x.abs().max().clamp(min=1e-12)
    scale = finfo.max / amax
    return (x * scale).clamp(min=finfo.min, max=finfo.max).to(dtype), scale.float().reciprocal().reshape(1)


@pytest.mark.parametrize("a_dt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("b_dt", [torch.float8_e4m3fn, torch.float8_e5m2])
@pytest.mark.parametrize("res_dt", [torch.bfloat16, torch.float16])
@pytest.mark.parametrize("b,M,K,N", [(16, 48, 64, 80), (128, 5, 128, 512), (3, 70, 512, 128)])
def test_bmm_fp8(a_dt, b_dt, res_dt, b, M, K, N):
    """test_bmm_fp8.py:21-44 (16 x 48 x 64 x 80, every operand / result type pair) + DeepSeek's absorbed-MLA shapes."""
    import oracle

    ops = _ops()
    from sglang_security_amd._lib import SglkError
    g = torch.Generator().manual_seed(b + M)
    x = torch.randn(b, M, K, generator=g)
    y = torch.randn(b, N, K, generator=g).transpose(-2, -1)      # column-major [b, K, N]
    xq, xs = _to_float8(x, a_dt)
    yq, ys = _to_float8(y, b_dt)
    out = torch.empty(b, M, N, dtype=res_dt, device=DEV)
    if a_dt == torch.float8_e5m2 and b_dt == torch.float8_e5m2:
        with pytest.raises(SglkError):
            ops.bmm_fp8(xq.to(DEV), yq.to(DEV), xs.to(DEV), ys.to(DEV), res_dt, out)
        return
    ops.bmm_fp8(xq.to(DEV), yq.to(DEV), xs.to(DEV), ys.to(DEV), res_dt, out)
    want = oracle.bmm_fp8(xq, yq, xs, ys, res_dt).float()
    got = out.float().cpu()
    scale = want.abs().max().item()
    assert (got - want).abs().max().item() <= 1e-2 * scale         # same operands, fp32 accumulation in another order
    cos = torch.nn.functional.cosine_similarity(torch.bmm(x, y).flatten(), got.flatten(), dim=0)
    assert cos.item() > 0.99                                       # the reference's own criterion


def test_sgl_kernel_surface_moe_and_bmm():
    import sglang_security_amd.sgl_kernel as sk
    from sglang_security_amd.sgl_kernel.torch_ops import register_torch_ops

    register_torch_ops()
    gating = torch.randn(9, 16, device=DEV)
    w = torch.empty(9, 2, device=DEV)
    ids = torch.empty(9, 2, dtype=torch.int32, device=DEV)
    src = torch.empty_like(ids)
    torch.ops.sgl_kernel.topk_softmax(w, ids, src, gating)
    w2, ids2, src2 = torch.empty_like(w), torch.empty_like(ids), torch.empty_like(src)
    sk.topk_softmax(w2, ids2, src2, gating)
    assert torch.equal(ids, ids2) and torch.equal(w, w2)
    srt = torch.full((18 + 16 * 3,), 18, dtype=torch.int32, device=DEV)
    ex = torch.zeros(17, dtype=torch.int32, device=DEV)
    tot = torch.zeros(1, dtype=torch.int32, device=DEV)
    cnt = torch.zeros(17 * 16, dtype=torch.int32, device=DEV)
    cs = torch.zeros(17, dtype=torch.int32, device=DEV)
    torch.ops.sgl_kernel.moe_align_block_size(ids, 16, 4, srt, ex, tot, cnt, cs)
    assert int(tot) % 4 == 0 and int(tot) >= 18


def test_empty_inputs():
    """Nothing to do: every entry point returns without touching its outputs' meaningful part."""
    ops = _ops()
    ids = torch.zeros(0, 2, dtype=torch.int32, device=DEV)
    srt = torch.full((8 * 3,), 0, dtype=torch.int32, device=DEV)
    ex = torch.full((8,), -1, dtype=torch.int32, device=DEV)
    tot = torch.full((1,), 7, dtype=torch.int32, device=DEV)
    ops.moe_align_block_size(ids, 8, 4, srt, ex, tot)
    assert int(tot) == 0 and (ex == -1).all()
    w = torch.empty(0, 2, dtype=torch.float32, device=DEV)
    i = torch.empty(0, 2, dtype=torch.int32, device=DEV)
    ops.topk_softmax(w, i, torch.empty_like(i), torch.empty(0, 16, device=DEV))
    a = torch.empty(0, 4, 32, dtype=torch.float8_e4m3fn, device=DEV)
    b = torch.empty(0, 16, 32, dtype=torch.float8_e4m3fn, device=DEV).transpose(-2, -1)
    one = torch.ones(1, device=DEV)
    assert ops.bmm_fp8(a, b, one, one, torch.bfloat16).shape == (0, 4, 16)
    torch.cuda.synchronize()


@pytest.mark.parametrize("seq_length", [1, 2, 7, 64, 1024, 8192])
@pytest.mark.parametrize("dtype", [torch.float16, torch.float32, torch.bfloat16])
@pytest.mark.parametrize("params", [(128, 4, 2, 4), (256, 8, 4, 8), (512, 16, 8, 16)])
@pytest.mark.parametrize("n_share", [0, 1, 8])
def test_moe_fused_gate(seq_length, dtype, params, n_share):
    """The inputs of the reference's own test (sgl-kernel/tests/test_moe_fused_gate.py:8-45) against the restatement of
    its kernel (arithmetic in the input type, its tie rules): expert ids equal slot for slot, weights to 1e-5, where
    the reference's test (:77-88, against a torch formula that leaves ties open) compares sorted rows at 1e-2."""
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    num_experts, num_expert_group, topk_group, topk = params
    torch.manual_seed(seq_length)
    tensor = torch.rand((seq_length, num_experts)).to(dtype)
    bias = torch.rand(num_experts).to(dtype)
    topk = topk + min(1, n_share)
    out, idx = sk.moe_fused_gate(tensor.cuda(), bias.cuda(), num_expert_group=num_expert_group, topk_group=topk_group,
                                 topk=topk, n_share_experts_fusion=n_share, routed_scaling_factor=2.5)
    out, idx = out.cpu(), idx.cpu()
    ref_out, ref_idx = oracle.moe_fused_gate(tensor, bias, topk, num_expert_group, topk_group, n_share, 2.5)
    assert out.dtype == torch.float32 and idx.dtype == torch.int32
    # a sigmoid that lands on the other side of a rounding boundary on the CPU (expf differs in the last bit) may move
    # a 16-bit choice score by one step and with it a tie: allow a few such rows in the large cases
    same = (ref_idx == idx).all(dim=1)
    assert same.float().mean().item() >= (1.0 if seq_length < 64 or dtype == torch.float32 else 0.995)
    torch.testing.assert_close(out[same], ref_out[same], rtol=1e-5, atol=1e-6)
    if n_share:
        assert torch.equal(idx[:, -1].long(), num_experts + torch.arange(seq_length) % n_share)
    k_routed = topk - min(1, n_share)
    torch.testing.assert_close(out[:, :k_routed].sum(-1), torch.ones(seq_length), rtol=1e-5, atol=1e-5)


def test_moe_fused_gate_matches_torch_formula_fp32():
    """fp32 inputs have no ties: the kernel agrees with the reference's torch formula itself
    (biased_grouped_topk_impl, topk.py:170-211), restated inline."""
    import sglang_security_amd.sgl_kernel as sk

    n, E, ng, tg, k = 333, 256, 8, 4, 8
    g = torch.Generator().manual_seed(0)
    x, b = torch.randn(n, E, generator=g), torch.rand(E, generator=g)
    scores = x.sigmoid()
    choice = scores + b
    gs = choice.view(n, ng, -1).topk(2, dim=-1)[0].sum(-1)
    gmask = torch.zeros_like(gs).scatter_(1, torch.topk(gs, tg, dim=-1)[1], 1)
    smask = gmask.unsqueeze(-1).expand(n, ng, E // ng).reshape(n, -1)
    ids = torch.topk(choice.masked_fill(~smask.bool(), float("-inf")), k, dim=-1)[1]
    w = scores.gather(1, ids)
    w = w / w.sum(-1, keepdim=True)
    out, idx = sk.moe_fused_gate(x.cuda(), b.cuda(), ng, tg, k)
    assert torch.equal(idx.cpu().long().sort()[0], ids.sort()[0])
    torch.testing.assert_close(out.cpu().sort()[0], w.sort()[0], rtol=1e-4, atol=1e-6)


def test_moe_fused_gate_group_limit():
    """Experts outside the topk_group best groups never win, however large their own score."""
    import sglang_security_amd.sgl_kernel as sk

    E, ng, tg, k = 64, 8, 2, 4
    x = torch.full((1, E), -4.0)
    x[0, 0] = 9.0                      # group 0: one huge score, second best tiny -> group sum ~ 1.02
    x[0, 8:10] = 2.0                   # group 1: two good scores -> ~1.76
    x[0, 16:18] = 1.5                  # group 2: ~1.63
    out, idx = sk.moe_fused_gate(x.cuda(), torch.zeros(E).cuda(), ng, tg, k)
    assert sorted(idx[0].tolist()) == [8, 9, 16, 17]
    assert abs(out.sum().item() - 1.0) < 1e-6


def test_moe_fused_gate_empty():
    import sglang_security_amd.sgl_kernel as sk

    out, idx = sk.moe_fused_gate(torch.empty(0, 256, device="cuda"), torch.zeros(256, device="cuda"), 8, 4, 8)
    assert out.shape == (0, 8) and idx.shape == (0, 8)
