"""GPU parity of the sampling filters with the criteria of the reference's own kernel test
(sgl-kernel/tests/test_sampling.py): renormalised distributions against the sort-based formulas (rtol = atol = 1e-3),
samples inside the filter masks, plus the inverse-CDF choice against a float64 restatement."""
import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"


def _probs(batch, vocab, seed=42):
    g = torch.Generator().manual_seed(seed)
    pre = torch.rand(batch, vocab, generator=g)
    return pre / pre.sum(dim=-1, keepdim=True)


@pytest.mark.parametrize("batch", [1, 99])
@pytest.mark.parametrize("vocab", [111, 32000, 128256])
@pytest.mark.parametrize("k", [10, 100, 500])
def test_top_k_renorm_probs(batch, vocab, k):
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    if k > vocab:
        pytest.skip("k should be less than vocab_size")
    p = _probs(batch, vocab)
    want = oracle.renorm(p, oracle.top_k_mask(p, k))
    got = sk.top_k_renorm_prob(p.to(DEV), k).cpu()
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-3)
    assert torch.equal(got > 0, want > 0)                       # the kept set itself is exact (integer selection)
    per_row = torch.full((batch,), k, dtype=torch.int32, device=DEV)
    assert torch.equal(sk.top_k_renorm_probs(p.to(DEV), per_row).cpu(), got)


@pytest.mark.parametrize("batch", [1, 99])
@pytest.mark.parametrize("vocab", [111, 32000, 128256])
@pytest.mark.parametrize("top_p", [0.1, 0.5, 0.9])
def test_top_p_renorm_probs(batch, vocab, top_p):
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    p = _probs(batch, vocab)
    want = oracle.renorm(p, oracle.top_p_mask(p, top_p))
    got = sk.top_p_renorm_prob(p.to(DEV), top_p).cpu()
    torch.testing.assert_close(got, want, rtol=1e-3, atol=1e-3)
    # the cut may move by the few entries whose cumulative sum sits within fp32 rounding of 1 - top_p
    assert ((got > 0) != (want > 0)).sum(dim=-1).max().item() <= 2


@pytest.mark.parametrize("batch", [1, 99])
@pytest.mark.parametrize("vocab", [111, 32000, 128256])
@pytest.mark.parametrize("top_p", [0.1, 0.5])
@pytest.mark.parametrize("order", ["joint", "top_k_first"])
def test_top_k_top_p_sampling_from_probs(batch, vocab, top_p, order):
    import oracle
    import sglang_security_amd.sgl_kernel as sk
    from sglang_security_amd import ops

    k = int(vocab * (0.5 if top_p == 0.1 else 0.1))
    p = _probs(batch, vocab)
    eps = 1e-4
    mk = oracle.top_k_mask(p, k)
    if order == "joint":
        mask = mk & oracle.top_p_mask(p, top_p, eps)
    else:
        mask = mk & oracle.top_p_mask(oracle.renorm(p, mk), top_p, eps)
    pd = p.to(DEV)
    kt = torch.full((batch,), k, dtype=torch.int32, device=DEV)
    pt = torch.full((batch,), top_p, dtype=torch.float32, device=DEV)
    rows = torch.arange(batch)
    gen = torch.Generator(device=DEV).manual_seed(7)
    for _ in range(20):
        s = sk.top_k_top_p_sampling_from_probs(pd, kt, pt, filter_apply_order=order, generator=gen).cpu().long()
        assert torch.all((s >= 0) & (s < vocab))
        assert torch.all(mask[rows, s])
    # the inverse-CDF choice itself, with known uniform numbers, against float64
    u = torch.rand(batch, generator=torch.Generator().manual_seed(3))
    got = ops.sampling_from_probs(pd, u.to(DEV), kt, pt, 0.0, order == "joint").cpu()
    # the kept set of the filters as the library cuts it (checked against the sort-based formulas above; `mask` carries the
    # reference test's eps slack and may hold a few more entries at the low end)
    if order == "joint":
        kept = mk & (sk.top_p_renorm_prob(pd, top_p).cpu() > 0)
    else:
        kept = sk.top_p_renorm_prob(sk.top_k_renorm_prob(pd, k), top_p).cpu() > 0
    want = oracle.sample_from_mask(p, kept, u)
    assert (got != want).float().mean().item() <= 0.05              # fp32 running sums near a boundary may step by one
    assert torch.all(mask[rows, got.long()])
    rank = torch.cumsum(kept.long(), dim=-1)                        # distance counted in kept entries, not in indices
    assert (rank[rows, got.long()] - rank[rows, want.long()]).abs().max().item() <= 1


@pytest.mark.parametrize("batch", [1, 99])
@pytest.mark.parametrize("vocab", [111, 32000, 128256])
@pytest.mark.parametrize("min_p", [0.05, 0.1, 0.2, 0.7, 1.0])
def test_min_p_sampling(batch, vocab, min_p):
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    p = _probs(batch, vocab)
    mask = oracle.min_p_mask(p, min_p)
    pd = p.to(DEV)
    mt = torch.full((batch,), min_p, dtype=torch.float32, device=DEV)
    rows = torch.arange(batch)
    for _ in range(20):
        s = sk.min_p_sampling_from_probs(pd, mt).cpu().long()
        assert torch.all(mask[rows, s])


def test_top_p_sampling_distribution():
    """Frequencies of many draws from a small distribution follow the filtered, renormalised probabilities."""
    import sglang_security_amd.sgl_kernel as sk

    p = torch.tensor([[0.05, 0.4, 0.02, 0.3, 0.03, 0.2]], device=DEV).repeat(4096, 1)
    gen = torch.Generator(device=DEV).manual_seed(11)
    s = sk.top_p_sampling_from_probs(p, 0.9, generator=gen).cpu().long()          # keeps 0.4, 0.3, 0.2 (mass 0.9)
    freq = torch.bincount(s, minlength=6).float() / s.numel()
    want = torch.tensor([0.0, 0.4, 0.0, 0.3, 0.0, 0.2]) / 0.9
    assert torch.all(freq[[0, 2, 4]] == 0)
    assert (freq - want).abs().max().item() < 0.03
