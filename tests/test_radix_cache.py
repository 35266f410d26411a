"""CPU: the native radix tree (libsglk sglk_radix_*, host code) against traces recorded from the reference
RadixCache (python/sglang/srt/mem_cache/radix_cache.py) -- every step's matched slots, returned prefix length,
freed slots, evictable / protected / total sizes and the full tree dump must be identical (integer work:
bit-exact).  The traces cover the fork's quirks: private root overwrite, fuzzy key match, 1024-token chunks,
public promotion by the judge, epoch windows and epoch-ordered eviction."""
import json
import os

import pytest
import torch

from conftest import GOLDEN
from sglang_security_amd.mem_cache.radix_cache import RadixCache

TRACES = json.load(open(os.path.join(GOLDEN, "radix_traces.json")))


class Alloc:
    def __init__(self):
        self.device = torch.device("cpu")
        self.freed = []

    def free(self, idx):
        self.freed.append(torch.as_tensor(idx).clone())


class Judge:
    def __init__(self):
        self.nodes = []

    def update_privacy(self, node=None, context="", prompt=""):
        self.nodes.append(node)


@pytest.mark.parametrize("name", sorted(TRACES.keys()))
def test_trace(name):
    epoch = [0]
    alloc, judge = Alloc(), Judge()
    tree = RadixCache(None, alloc, page_size=1, private_judge_client=judge, get_epoch_fn=lambda: epoch[0])
    handles = {}
    for i, st in enumerate(TRACES[name]):
        kind = st["op"]
        nfreed = len(alloc.freed)
        if kind == "epoch":
            epoch[0] = st["value"]
        elif kind == "insert":
            r = tree.insert(st["key"], torch.tensor(st["value"], dtype=torch.int64), st.get("prompt", ""), user_id=st.get("user"))
            assert r == st["ret"], (name, i, "insert ret")
        elif kind == "match":
            if "user" in st:
                v, node = tree.match_prefix(st["key"], user_id=st["user"])
            else:
                v, node = tree.match_prefix(st["key"])
            assert v.dtype == torch.int64
            assert v.tolist() == st["ret"], (name, i, "matched slots")
            assert (node.key if node.id != 0 else []) == st["last_node_key"], (name, i, "last node")
            if "handle" in st:
                handles[st["handle"]] = node
        elif kind == "lock":
            assert tree.inc_lock_ref(handles[st["handle"]]) == st["ret"]
        elif kind == "unlock":
            assert tree.dec_lock_ref(handles[st["handle"]]) == st["ret"]
        elif kind == "evict":
            tree.evict(st["n"])
        elif kind == "set_public":
            for n in judge.nodes:
                if n.need_check_privacy and len(n.key) >= st.get("min_len", 0):
                    n.private = False
                    n.need_check_privacy = False
        freed = [int(t) for f in alloc.freed[nfreed:] for t in f]
        want_freed = [t for f in st["freed"] for t in f]
        assert freed == want_freed, (name, i, kind, "freed slots")
        assert tree.evictable_size() == st["evictable"], (name, i, kind, "evictable")
        assert tree.protected_size() == st["protected"], (name, i, kind, "protected")
        assert tree.total_size() == st["total"], (name, i, kind, "total")
        assert tree.dump() == st["tree"], (name, i, kind, "tree")


def test_empty_and_disabled():
    tree = RadixCache(None, Alloc(), 1, Judge(), get_epoch_fn=lambda: 0)
    v, n = tree.match_prefix([])
    assert v.numel() == 0 and n.id == 0
    assert tree.insert([], torch.zeros(0, dtype=torch.int64)) == 0
    off = RadixCache(None, Alloc(), 1, Judge(), disable=True, get_epoch_fn=lambda: 0)
    assert off.insert([1, 2, 3], torch.arange(3)) == 0
    assert off.match_prefix([1, 2, 3])[0].numel() == 0


def test_upstream_mode_exact_match_and_sharing():
    """compat_fork=False: exact key match, any user shares, no chunking."""
    tree = RadixCache(None, Alloc(), 1, Judge(), compat_fork=False, get_epoch_fn=lambda: 0)
    a = list(range(100, 1400))
    assert tree.insert(a, torch.arange(1300), user_id="u1") == 0
    assert len(tree.dump()) == 1  # one node, not 1024-token chunks
    b = a[:700] + [7] + a[701:]
    v, _ = tree.match_prefix(b, user_id="u2")
    assert v.tolist() == list(range(700))  # stops at the first mismatch
    assert tree.insert(b, torch.arange(2000, 3300), user_id="u2") == 700
    assert tree.total_size() == 1300 + 600 and tree.evictable_size() == 1900


def test_large_random_consistency():
    """Maximum-size style stress: 2000 operations, sizes stay consistent with a full walk (upstream mode)."""
    import random

    rng = random.Random(1)
    tree = RadixCache(None, Alloc(), 1, Judge(), compat_fork=False, get_epoch_fn=lambda: 0)
    bases = [[rng.randrange(50000) for _ in range(rng.randrange(5, 400))] for _ in range(20)]
    nxt = 1
    for _ in range(2000):
        b = rng.choice(bases)
        key = b[: rng.randrange(1, len(b) + 1)] + [rng.randrange(50000) for _ in range(rng.randrange(0, 30))]
        if rng.random() < 0.5:
            tree.insert(key, torch.arange(nxt, nxt + len(key)))
            nxt += len(key)
        elif rng.random() < 0.9:
            v, node = tree.match_prefix(key)
            assert len(v) <= len(key)
        else:
            tree.evict(rng.randrange(1, 200))
        assert tree.evictable_size() == tree.total_size()


def test_privacy_hook_end_to_end():
    """SafeKV flow on the native tree: nodes start private and unchecked; the judge thread promotes the ones the
    detector calls public; children of a decided-private node inherit without a detector call; another user then
    shares only the public part."""
    from sglang_security_amd.private_service.private_client import PrivateJudgeClient

    seen = []

    def detector(text):
        seen.append(text)
        return "secret" in text

    judge = PrivateJudgeClient(detector, num_workers=2)
    tree = RadixCache(None, Alloc(), 1, judge, get_epoch_fn=lambda: 0)
    pub = list(range(100, 1124))          # exactly one 1024-token chunk
    priv = list(range(5000, 5040))
    tree.insert(pub, torch.arange(1, 1025), "system prompt line\nUser: hi", user_id="0")
    assert judge.drain()
    d = tree.dump()
    assert len(d) == 1 and d[0]["private"] is False and d[0]["need_check"] is False
    tree.insert(pub + priv, torch.arange(1, 1065), "my secret is 42\nUser: continue", user_id="0")
    assert judge.drain()
    d = tree.dump()
    assert [n["private"] for n in d] == [False, True] and [n["need_check"] for n in d] == [False, False]
    # a third level under the decided-private node inherits at once: no detector call
    calls = len(seen)
    tree.insert(pub + priv + [7, 8, 9], torch.arange(1, 1068), "anything\nUser: x", user_id="0")
    assert judge.drain()
    assert len(seen) == calls and tree.dump()[2]["need_check"] is False and tree.dump()[2]["private"] is True
    # another user shares the public chunk only
    v, node = tree.match_prefix(pub + priv, user_id="someone-else")
    assert v.tolist() == list(range(1, 1025))
    v, node = tree.match_prefix(pub + priv, user_id="0")
    assert len(v) == 1064
    # the placeholder rule: lines that look like chat-template residue are public without asking the detector
    assert PrivateJudgeClient.judged_text("a\nAssistant: b\nc") == "hello, you are an simple assistant"
    assert PrivateJudgeClient.judged_text("only one line") == "only one line"
    judge.close()
