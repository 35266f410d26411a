"""GPU parity of the draft-tree helpers with the golden vectors of the reference's own tests
(python/sglang/srt/speculative/build_eagle_tree.py:115-384, sgl-kernel/tests/speculative/test_eagle_utils.py,
sgl-kernel/tests/speculative/test_speculative_sampling.py) and, on random trees, with the serial restatement."""
import json
import math
import os

import pytest
import torch

pytestmark = pytest.mark.gpu
DEV = "cuda"
GOLDEN = os.path.join(os.path.dirname(__file__), "golden", "eagle_tree.json")


def _t(entry):
    return torch.tensor(entry["data"], dtype=getattr(torch, entry["dtype"]))


def _golden_tree_inputs():
    import oracle

    g = json.load(open(GOLDEN))
    parent_list, top, draft = oracle.build_tree_preprocess(
        _t(g["verified_id"]), [_t(x) for x in g["score_list"]], [_t(x) for x in g["token_list"]],
        [_t(x) for x in g["parents_list"]], g["num_draft_token"])
    return g, parent_list, top, draft, _t(g["seq_lens"]).to(torch.int32)


def _run_build(parent_list, selected, seq_lens, topk, depth, n):
    import sglang_security_amd.sgl_kernel as sk

    bs = selected.shape[0]
    mask = torch.full((int(seq_lens.sum()) * n + n * n * bs,), True, device=DEV)
    positions = torch.empty(bs * n, dtype=torch.long, device=DEV)
    ridx = torch.full((bs, n), -1, dtype=torch.long, device=DEV)
    nxt = torch.full((bs, n), -1, dtype=torch.long, device=DEV)
    sib = torch.full((bs, n), -1, dtype=torch.long, device=DEV)
    sk.build_tree_kernel_efficient(parent_list.to(DEV), selected.to(DEV), seq_lens.to(DEV), mask, positions, ridx, nxt,
                                   sib, topk, depth, n)
    return mask.cpu().tolist(), positions.cpu().tolist(), ridx.cpu().tolist(), nxt.cpu().tolist(), sib.cpu().tolist()


def test_build_tree_golden():
    import oracle

    g, parent_list, top, draft, seq_lens = _golden_tree_inputs()
    exp = g["expected"]
    assert draft.tolist() == exp["draft_tokens"]                              # the preprocessing restatement itself
    mask, positions, ridx, nxt, sib = _run_build(parent_list, top, seq_lens, g["topk"], g["depth"], g["num_draft_token"])
    assert positions == exp["position"]
    assert ridx == exp["retrive_index"]
    assert nxt == exp["retrive_next_token"]
    assert sib == exp["retrive_next_sibling"]
    want = oracle.build_tree_efficient(parent_list, top, seq_lens, g["topk"], g["depth"], g["num_draft_token"])
    assert mask == want[0]                                                    # the reference's test does not hold the mask


def _random_tree(bs, topk, depth, n, seed):
    """Scores shaped like the drafter's ([bs, 1, topk] then [bs, topk, topk] per step) and the parent lists that go with
    them (flat indices of the topk expanded nodes of every step), as eagle_worker builds them."""
    g = torch.Generator().manual_seed(seed)
    score_list, parents_list = [], []
    score = torch.rand(bs, 1, topk, generator=g)
    score_list.append(score)
    parents_list.append(torch.arange(-1, topk).unsqueeze(0).repeat(bs, 1))
    prev = score.squeeze(1)                                                           # [bs, topk] scores of the frontier
    for step in range(1, depth):
        child = prev.unsqueeze(2) * torch.rand(bs, topk, topk, generator=g)            # [bs, topk, topk]
        score_list.append(child)
        top = torch.topk(child.flatten(1), topk, dim=-1)
        parents_list.append(top.indices + (topk * topk * (step - 1) + topk))
        prev = top.values
    token_list = [torch.randint(0, 32000, (bs, s.shape[1] * s.shape[2]), generator=g) for s in score_list]
    return score_list, token_list, parents_list


@pytest.mark.parametrize("bs,topk,depth,n", [(1, 4, 4, 8), (7, 8, 5, 64), (64, 4, 3, 16), (3, 1, 5, 6), (5, 8, 1, 8)])
def test_build_tree_random(bs, topk, depth, n):
    import oracle

    score_list, token_list, parents_list = _random_tree(bs, topk, depth, n, seed=bs * 100 + n)
    verified = torch.randint(0, 32000, (bs,), generator=torch.Generator().manual_seed(1))
    parent_list, top, _ = oracle.build_tree_preprocess(verified, score_list, token_list, parents_list, n)
    seq_lens = torch.randint(1, 300, (bs,), generator=torch.Generator().manual_seed(2), dtype=torch.int32)
    got = _run_build(parent_list.long(), top, seq_lens, topk, depth, n)
    want = oracle.build_tree_efficient(parent_list, top, seq_lens, topk, depth, n)
    for a, b, what in zip(got, want, ("mask", "positions", "retrive_index", "next_token", "next_sibling")):
        assert a == b, what


def test_build_tree_orphan_is_skipped():
    """A selected token whose parent was not selected gets no links (the reference warns and skips it)."""
    import oracle

    topk, depth, n = 2, 3, 4
    parent_list = torch.tensor([[-1, 0, 1, 2, 4]], dtype=torch.int64)        # rows 1, 2 expand nodes 0, 1; rows 3, 4 -> 2, 4
    selected = torch.tensor([[0, 1, 8]], dtype=torch.int64)                  # 8 lives in row 4 whose parent (4) is missing
    seq_lens = torch.tensor([3], dtype=torch.int32)
    got = _run_build(parent_list, selected, seq_lens, topk, depth, n)
    want = oracle.build_tree_efficient(parent_list, selected, seq_lens, topk, depth, n)
    assert got[1:] == tuple(want[1:]) or list(got[1:]) == list(want[1:])
    assert got[0] == want[0]
    assert got[3] == [[1, -1, -1, -1]] and got[4] == [[-1, 2, -1, -1]]


def _golden_walk_inputs():
    candidates = torch.tensor([[0, 1, 2, 3, 4, 5], [7, 8, 9, 10, 11, 12]], dtype=torch.int32)
    retrive_index = torch.tensor([[0, 1, 2, 3, 4, 5], [6, 7, 8, 9, 10, 11]], dtype=torch.int32)
    retrive_next_token = torch.tensor([[1, 2, -1, 4, 5, -1], [4, 2, 3, -1, 5, -1]], dtype=torch.int32)
    retrive_next_sibling = torch.tensor([[-1, 3, -1, -1, -1, -1], [-1, -1, -1, -1, 1, -1]], dtype=torch.int32)
    logits = torch.full((2, 6, 20), 1, dtype=torch.float32)
    logits[0, 0, 3] = 10
    logits[0, 3, 4] = 10
    logits[0, 4, 5] = 10
    logits[1, 0, 11] = 10
    logits[1, 4, 12] = 10
    for i in range(2):
        for j in range(6):
            if torch.max(logits[i, j]) < 10:
                logits[i, j, 18] = 10
    return candidates, retrive_index, retrive_next_token, retrive_next_sibling, logits


def test_verify_tree_greedy_golden():
    """test_eagle_utils.py:7-84."""
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    cand, ridx, nxt, sib, logits = _golden_walk_inputs()
    target_predict = torch.argmax(logits, dim=-1).to(torch.int32)
    predicts = torch.full((12,), -1, dtype=torch.int32, device=DEV)
    accept_index = torch.full((2, 4), -1, dtype=torch.int32, device=DEV)
    accept_num = torch.zeros(2, dtype=torch.int32, device=DEV)
    sk.verify_tree_greedy(predicts=predicts, accept_index=accept_index, accept_token_num=accept_num,
                          candidates=cand.to(DEV), retrive_index=ridx.to(DEV), retrive_next_token=nxt.to(DEV),
                          retrive_next_sibling=sib.to(DEV), target_predict=target_predict.to(DEV))
    assert predicts.tolist() == [3, -1, -1, 4, 5, 18, 11, -1, -1, -1, 12, 18]
    assert accept_index.tolist() == [[0, 3, 4, 5], [6, 10, 11, -1]]
    assert accept_num.tolist() == [3, 2]
    want = oracle.verify_tree_greedy(cand, ridx, nxt, sib, target_predict, 4)
    assert want == (predicts.tolist(), accept_index.tolist(), accept_num.tolist())


@pytest.mark.parametrize("thr_single,thr_acc,exp_predicts,exp_index,exp_num", [
    (1, 1, [3, -1, -1, 4, 5, 18, 11, -1, -1, -1, 12, 18], [[0, 3, 4, 5], [6, 10, 11, -1]], [3, 2]),
    (0, 0, [1, 2, 18, -1, -1, -1, 11, -1, -1, -1, 12, 18], [[0, 1, 2, -1], [6, 10, 11, -1]], [2, 2]),
])
def test_tree_speculative_sampling_golden(thr_single, thr_acc, exp_predicts, exp_index, exp_num):
    """test_speculative_sampling.py:6-123."""
    import sglang_security_amd.sgl_kernel as sk

    cand, ridx, nxt, sib, logits = _golden_walk_inputs()
    target_probs = torch.softmax(logits / 0.01, dim=-1).to(DEV)
    draft_probs = torch.zeros_like(target_probs)
    coins = torch.rand(2, 6, device=DEV)
    predicts = torch.full((12,), -1, dtype=torch.int32, device=DEV)
    accept_index = torch.full((2, len(exp_index[0])), -1, dtype=torch.int32, device=DEV)
    accept_num = torch.zeros(2, dtype=torch.int32, device=DEV)
    sk.tree_speculative_sampling_target_only(
        predicts=predicts, accept_index=accept_index, accept_token_num=accept_num, candidates=cand.to(DEV),
        retrive_index=ridx.to(DEV), retrive_next_token=nxt.to(DEV), retrive_next_sibling=sib.to(DEV),
        uniform_samples=coins, target_probs=target_probs, draft_probs=draft_probs, threshold_single=thr_single,
        threshold_acc=thr_acc, deterministic=True)
    assert predicts.tolist() == exp_predicts
    assert accept_index.tolist() == exp_index
    assert accept_num.tolist() == exp_num


def _tree_for_walk(bs, topk, depth, n, vocab, seed):
    import oracle

    score_list, token_list, parents_list = _random_tree(bs, topk, depth, n, seed)
    token_list = [t % vocab for t in token_list]
    verified = torch.randint(0, vocab, (bs,), generator=torch.Generator().manual_seed(seed + 1))
    parent_list, top, draft = oracle.build_tree_preprocess(verified, score_list, token_list, parents_list, n)
    seq_lens = torch.full((bs,), 4, dtype=torch.int32)
    _, _, ridx, nxt, sib = oracle.build_tree_efficient(parent_list, top, seq_lens, topk, depth, n)
    i32 = lambda x: torch.tensor(x, dtype=torch.int32)
    return draft.reshape(bs, n).to(torch.int32), i32(ridx), i32(nxt), i32(sib)


@pytest.mark.parametrize("bs,topk,depth,n,steps", [(1, 4, 4, 8, 5), (33, 8, 5, 64, 6), (130, 2, 3, 6, 4)])
def test_verify_tree_greedy_random(bs, topk, depth, n, steps):
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    vocab = 50
    cand, ridx, nxt, sib = _tree_for_walk(bs, topk, depth, n, vocab, seed=bs)
    g = torch.Generator().manual_seed(9)
    target = torch.randint(0, vocab, (bs, n), generator=g, dtype=torch.int32)
    # make long accepted paths likely: the target's pick at a node is often one of its children
    nl, cl = nxt.tolist(), cand.tolist()
    for b in range(bs):
        for i in range(n):
            if nl[b][i] != -1 and torch.rand(1, generator=g).item() < 0.7:
                target[b, i] = cl[b][nl[b][i]]
    predicts = torch.full((bs * n,), -1, dtype=torch.int32, device=DEV)
    accept_index = torch.full((bs, steps), -1, dtype=torch.int32, device=DEV)
    accept_num = torch.zeros(bs, dtype=torch.int32, device=DEV)
    sk.verify_tree_greedy(predicts, accept_index, accept_num, cand.to(DEV), ridx.to(DEV), nxt.to(DEV), sib.to(DEV),
                          target.to(DEV))
    want = oracle.verify_tree_greedy(cand, ridx, nxt, sib, target, steps)
    assert (predicts.tolist(), accept_index.tolist(), accept_num.tolist()) == want
    assert max(want[2]) >= 2


@pytest.mark.parametrize("bs,topk,depth,n,steps,vocab", [(1, 4, 4, 8, 5, 50), (17, 8, 5, 32, 6, 1000), (4, 4, 4, 16, 5, 32000)])
@pytest.mark.parametrize("thr_single,thr_acc", [(1.0, 1.0), (0.3, 0.5)])
def test_tree_speculative_sampling_random(bs, topk, depth, n, steps, vocab, thr_single, thr_acc):
    import oracle
    import sglang_security_amd.sgl_kernel as sk

    cand, ridx, nxt, sib = _tree_for_walk(bs, topk, depth, n, vocab, seed=bs + 3)
    g = torch.Generator().manual_seed(4)
    logits = torch.randn(bs, n, vocab, generator=g) * 2
    cl, nl = cand.tolist(), nxt.tolist()
    for b in range(bs):                                                           # give children real acceptance chances
        for i in range(n):
            if nl[b][i] != -1:
                logits[b, i, cl[b][nl[b][i]]] += math.log(vocab) + 2        # about half of the row's mass
    target_probs = torch.softmax(logits, dim=-1)
    draft0 = torch.zeros_like(target_probs)
    coins = torch.rand(bs, n, generator=g)
    predicts = torch.full((bs * n,), -1, dtype=torch.int32, device=DEV)
    accept_index = torch.full((bs, steps), -1, dtype=torch.int32, device=DEV)
    accept_num = torch.zeros(bs, dtype=torch.int32, device=DEV)
    draft_dev = draft0.to(DEV)
    sk.tree_speculative_sampling_target_only(predicts, accept_index, accept_num, cand.to(DEV), ridx.to(DEV), nxt.to(DEV),
                                             sib.to(DEV), coins.to(DEV), target_probs.to(DEV), draft_dev, thr_single,
                                             thr_acc, True)
    draft_cpu = draft0.clone()
    want = oracle.tree_speculative_sampling_target_only(cand, ridx, nxt, sib, coins, target_probs, draft_cpu, steps,
                                                        thr_single, thr_acc)
    assert accept_index.tolist() == want[1]
    assert accept_num.tolist() == want[2]
    assert torch.equal(draft_dev.cpu(), draft_cpu)                                # rejected candidates' probabilities
    got = predicts.tolist()
    # accepted tokens are exact; the drawn token may step to a neighbour when u * total sits within fp32 rounding of a
    # boundary of the running sum
    diff = [i for i, (a, b) in enumerate(zip(got, want[0])) if a != b]
    assert len(diff) <= max(1, bs // 10)
    assert max(want[2]) >= 1


def test_tree_helpers_empty_batch():
    import sglang_security_amd.sgl_kernel as sk

    e32 = lambda *s: torch.empty(*s, dtype=torch.int32, device=DEV)
    sk.verify_tree_greedy(e32(0), e32(0, 4), e32(0), e32(0, 6), e32(0, 6), e32(0, 6), e32(0, 6), e32(0, 6))
    e64 = lambda *s: torch.empty(*s, dtype=torch.int64, device=DEV)
    sk.build_tree_kernel_efficient(e64(0, 13), e64(0, 7), e32(0), torch.empty(0, dtype=torch.bool, device=DEV), e64(0),
                                   e64(0, 8), e64(0, 8), e64(0, 8), 4, 4, 8)


@pytest.mark.parametrize("lens", [[1], [8, 9, 0, 64, 7], [1000, 3, 129]])
def test_segment_packbits(lens):
    """Every segment equals numpy.packbits(bitorder="little") of its bits (the definition flashinfer's SegmentPackBits
    follows; flashinfer itself is not in the reference tree: parity unpinned beyond that definition)."""
    import numpy as np
    import sglang_security_amd.sgl_kernel as sk

    g = torch.Generator().manual_seed(sum(lens))
    x = torch.rand(sum(lens), generator=g) < 0.5
    iptr = torch.tensor([0] + list(np.cumsum(lens)), dtype=torch.int32)
    olen = [(n + 7) // 8 for n in lens]
    optr = torch.tensor([0] + list(np.cumsum(olen)), dtype=torch.int32)
    y = torch.full((sum(olen),), 0xAA, dtype=torch.uint8, device=DEV)
    sk.segment_packbits(x.to(DEV), iptr.to(DEV), optr.to(DEV), y)
    want = np.concatenate([np.packbits(x[iptr[i]: iptr[i + 1]].numpy(), bitorder="little") for i in range(len(lens))]
                          + [np.zeros(0, dtype=np.uint8)])
    assert y.cpu().numpy().tolist() == want.tolist()
