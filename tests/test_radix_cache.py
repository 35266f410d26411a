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


class ReqPool:
    """ReqToTokenPool stand-in (memory_pool.py:54-99 of the reference): the int32 table, write() and free()."""

    def __init__(self, rows=8, width=4096):
        self.req_to_token = torch.zeros(rows, width, dtype=torch.int32)
        self.freed = []

    def write(self, indices, values):
        self.req_to_token[indices] = values

    def free(self, idx):
        self.freed.append(int(idx))


class Req:
    """The fields of the reference's Req that cache_finished_req / cache_unfinished_req touch."""

    def __init__(self, rid, tokens, user):
        self.rid, self.origin_input_ids, self.output_ids, self.fill_ids = rid, list(tokens), [], []
        self.prefix_indices, self.last_node, self.req_pool_idx = [], None, None
        self.user_id, self.origin_input_text = user, ""


class Judge:
    def __init__(self):
        self.nodes = []

    def update_privacy(self, node=None, context="", prompt=""):
        self.nodes.append(node)


@pytest.mark.parametrize("name", sorted(TRACES.keys()))
def test_trace(name):
    epoch = [0]
    alloc, judge, pool = Alloc(), Judge(), ReqPool()
    tree = RadixCache(pool, alloc, page_size=1, private_judge_client=judge, compat_fork=True,
                      get_epoch_fn=lambda: epoch[0])
    handles, reqs, next_slot = {}, {}, [100000]

    def take(n):
        s0 = next_slot[0]
        next_slot[0] += n
        return torch.arange(s0, s0 + n, dtype=torch.int64)

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
            assert (node.key if node != tree.root_node else []) == st["last_node_key"], (name, i, "last node")
            if "handle" in st:
                handles[st["handle"]] = node
        elif kind == "lock":
            assert tree.inc_lock_ref(handles[st["handle"]]) == st["ret"]
        elif kind == "unlock":
            assert tree.dec_lock_ref(handles[st["handle"]]) == st["ret"]
        elif kind == "evict":
            tree.evict(st["n"])
        elif kind in ("req_start", "req_chunk"):
            # the scheduler's steps around the tree (schedule_batch.py:630-656,1275 of the reference)
            if kind == "req_start":
                r = Req(st["rid"], st["tokens"], st.get("user"))
                r.req_pool_idx = st["row"]
                reqs[st["rid"]] = r
                r.fill_ids = r.origin_input_ids[: st.get("fill", len(r.origin_input_ids))]
                v, node = tree.match_prefix(rid=r.rid, user_id=r.user_id, key=r.fill_ids[:-1])
                r.prefix_indices, r.last_node = v, node
                tree.inc_lock_ref(node)
                pool.write((r.req_pool_idx, slice(0, len(v))), v.to(torch.int32))
            else:
                r = reqs[st["rid"]]
                r.fill_ids = r.origin_input_ids[: st["fill"]]
            have = len(r.prefix_indices)
            pool.write((r.req_pool_idx, slice(have, len(r.fill_ids))), take(len(r.fill_ids) - have).to(torch.int32))
            assert [int(t) for t in r.prefix_indices] == st["prefix"], (name, i, "matched prefix")
        elif kind == "req_unfinished":
            r = reqs[st["rid"]]
            tree.cache_unfinished_req(r)
            assert [int(t) for t in r.prefix_indices] == st["prefix"], (name, i, "prefix after cache_unfinished_req")
            assert (r.last_node.key if r.last_node != tree.root_node else []) == st["last_node_key"], (name, i)
            assert pool.req_to_token[r.req_pool_idx, : len(r.fill_ids)].tolist() == st["row"], (name, i, "row")
        elif kind == "req_finished":
            r = reqs[st["rid"]]
            r.output_ids = list(st["output"])
            n_all = len(r.origin_input_ids) + len(r.output_ids) - 1
            have = len(r.fill_ids)
            if n_all > have:
                pool.write((r.req_pool_idx, slice(have, n_all)), take(n_all - have).to(torch.int32))
            npool = len(pool.freed)
            tree.cache_finished_req(r)
            assert pool.freed[npool:] == st["pool_freed"], (name, i, "request row freed")
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
    assert v.numel() == 0 and n == tree.root_node
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

    judge = PrivateJudgeClient(detector=detector, num_workers=2)
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


def test_stale_handle_after_reset_names_nothing():
    """A privacy verdict that arrives after reset() must not touch the node that took over the slot: handles carry a
    generation, and a stale one resolves to nothing (no-op setters, no-op lock calls)."""
    judge = Judge()
    tree = RadixCache(None, Alloc(), 1, judge, compat_fork=True, get_epoch_fn=lambda: 0)
    tree.insert(list(range(10, 40)), torch.arange(30), user_id="0")
    old = judge.nodes[-1]
    old_root = tree.root_node
    assert old.private and old.alive
    tree.reset()
    assert not old.alive and not old_root.alive and tree.root_node != old_root
    tree.insert(list(range(50, 90)), torch.arange(40), user_id="0")
    new = judge.nodes[-1]
    assert (new.id & 0xFFFFFFFF) == (old.id & 0xFFFFFFFF) and new.id != old.id  # same slot, next generation
    old.private = False            # the late verdict
    old.need_check_privacy = False
    assert new.private and new.need_check_privacy
    before = (tree.evictable_size(), tree.protected_size())
    assert tree.inc_lock_ref(old) == 0 and tree.dec_lock_ref(old) == 0
    assert (tree.evictable_size(), tree.protected_size()) == before
    with pytest.raises(KeyError):
        _ = old.key


def test_evicted_nodes_release_host_memory():
    """Host memory follows the live tree: after many insert / evict rounds the node count is that of the tree, not
    of everything ever inserted."""
    import random

    rng = random.Random(4)
    for mode in (True, "fixed", False):
        tree = RadixCache(None, Alloc(), 1, Judge(), compat_fork=mode, get_epoch_fn=lambda: 0)
        nxt = 1
        for _ in range(300):
            key = [rng.randrange(50000) for _ in range(rng.randrange(5, 80))]
            tree.insert(key, torch.arange(nxt, nxt + len(key)), user_id="0")
            nxt += len(key)
            if tree.evictable_size() > 400:
                tree.evict(300)
        assert tree.live_nodes() == len(tree.dump()) + 1  # + the root
        assert tree.live_nodes() < 60
        tree.evict(10 ** 9)
        assert tree.live_nodes() == 1 and tree.total_size() == 0


def test_fixed_mode_returns_orphaned_slots():
    """compat_fork="fixed": the child that another user's insert overwrites at the private root gives its slots back
    (at once when unlocked, at the last unlock otherwise) and leaves the size counters; compat_fork=True keeps the
    fork's leak (evictable_size() drifts away from total_size())."""
    a = list(range(100, 130))
    # the fork, bit for bit: the overwritten child's 30 slots stay counted and are never freed
    alloc = Alloc()
    tree = RadixCache(None, alloc, 1, Judge(), compat_fork=True, get_epoch_fn=lambda: 0)
    tree.insert(a, torch.arange(1, 31), user_id="alice")
    tree.insert(a, torch.arange(31, 61), user_id="alice")
    assert tree.evictable_size() == 60 and tree.total_size() == 30 and not alloc.freed
    # leak closed
    alloc = Alloc()
    tree = RadixCache(None, alloc, 1, Judge(), get_epoch_fn=lambda: 0)
    tree.insert(a, torch.arange(1, 31), user_id="alice")
    tree.insert(a, torch.arange(31, 61), user_id="alice")
    assert tree.evictable_size() == 30 and tree.total_size() == 30
    tree.evict(0)  # drains the free queue
    assert sorted(int(t) for f in alloc.freed for t in f) == list(range(1, 31))
    # a locked orphan returns its slots when the request lets go
    alloc = Alloc()
    tree = RadixCache(None, alloc, 1, Judge(), get_epoch_fn=lambda: 0)
    tree.insert(a, torch.arange(1, 31), user_id="alice")
    v, node = tree.match_prefix(a, user_id="alice")
    assert len(v) == 30
    tree.inc_lock_ref(node)
    tree.insert(a, torch.arange(31, 61), user_id="bob")
    assert node.alive and tree.protected_size() == 30 and tree.evictable_size() == 30
    tree.dec_lock_ref(node)
    tree.evict(0)
    assert not node.alive and tree.protected_size() == 0 and tree.evictable_size() == 30 == tree.total_size()
    assert sorted(int(t) for f in alloc.freed for t in f) == list(range(1, 31))


class CountingAlloc(Alloc):
    """An allocator stand-in that hands slots out and notices a slot returned twice."""

    def __init__(self, n):
        super().__init__()
        self.free_list = list(range(1, n + 1))

    def alloc(self, n):
        out, self.free_list = self.free_list[:n], self.free_list[n:]
        assert len(out) == n, "pool exhausted"
        return torch.tensor(out, dtype=torch.int64)

    def free(self, idx):
        super().free(idx)
        self.free_list.extend(int(t) for t in torch.as_tensor(idx).reshape(-1))


@pytest.mark.parametrize("users", [("alice", "alice", "alice"), ("alice", "bob", "alice")])
def test_fixed_mode_request_cycles_share_prefix_slots(users):
    """compat_fork="fixed" through the scheduler's own life cycle (match_prefix -> inc_lock_ref -> cache_finished_req,
    radix_cache.py:160-197 of the reference): a finishing request inserts its whole req_to_token row, prefix slots
    included, and at the private root that insert overwrites the very child it matched.  The overwritten child's slots
    that the new node names again belong to the new node: no slot still in the tree may reach the allocator, and no
    slot may reach it twice."""
    alloc, pool = CountingAlloc(4096), ReqPool(rows=4, width=512)
    tree = RadixCache(pool, alloc, 1, Judge(), get_epoch_fn=lambda: 0)
    base = list(range(1000, 1100))
    for i, user in enumerate(users):
        toks = base + [5000 + 10 * i + j for j in range(20)]
        r = Req(f"r{i}", toks, user)
        r.req_pool_idx = i
        v, node = tree.match_prefix(rid=r.rid, user_id=user, key=toks[:-1])
        r.prefix_indices, r.last_node = v, node
        tree.inc_lock_ref(node)
        pool.write((i, slice(0, len(v))), v.to(torch.int32))
        fresh = alloc.alloc(len(toks) - len(v))
        pool.write((i, slice(len(v), len(toks))), fresh.to(torch.int32))
        r.output_ids = [7]
        tree.cache_finished_req(r)
        tree.evict(0)  # drains the free queue into the allocator
        live = set(tree.all_values_flatten().tolist())
        assert len(alloc.free_list) == len(set(alloc.free_list)), "a slot was returned twice"
        assert not (live & set(alloc.free_list)), "a slot the tree still maps went back to the allocator"
        assert tree.evictable_size() == tree.total_size() and tree.protected_size() == 0
        # conservation: every slot is either in the tree or in the allocator
        assert len(live) + len(alloc.free_list) == 4096
    tree.evict(10 ** 9)
    assert sorted(alloc.free_list) == list(range(1, 4097))


def test_grace_node_survives_later_matches_until_locked():
    """The hit-rate release can cut off the very node match_prefix returns (radix_cache.py:385-411).  The scheduler
    matches many requests before it locks one, so that node must still resolve when inc_lock_ref arrives: the lock walks
    up from it into the live ancestors, as the reference does with its dead TreeNode object."""
    epoch = [0]
    alloc = Alloc()
    tree = RadixCache(None, alloc, 1, Judge(), compat_fork=True, get_epoch_fn=lambda: epoch[0])
    a, b = list(range(10, 50)), list(range(10, 50)) + list(range(200, 230))
    tree.insert(a, torch.arange(1, 41), user_id="0")
    tree.insert(b, torch.arange(1, 71), user_id="0")
    # window 0: one hit on the leaf; window 1: enough hits by one user to trip the release
    tree.match_prefix(b, user_id="0")
    epoch[0] = 1
    tree.match_prefix(b, user_id="0")  # rolls the window: hit_pre = 1, u_cnt_pre = 1
    node = None
    for _ in range(40):
        v, node = tree.match_prefix(b, user_id="0")
        if not any(d["key"] == list(range(200, 230)) for d in tree.dump()):
            break
    else:
        pytest.skip("release did not trigger with this script")
    assert node.alive
    tree.match_prefix(list(range(900, 910)), user_id="0")  # other requests are matched before the lock arrives
    tree.match_prefix(a, user_id="0")
    assert node.alive
    before = tree.protected_size()
    tree.inc_lock_ref(node)
    assert tree.protected_size() > before  # the live ancestors on the returned prefix are now protected
    tree.dec_lock_ref(node)
    assert tree.protected_size() == before


@pytest.mark.gpu
def test_traces_on_the_gpu_box():
    """The trace suite again under the gpu marker: the native tree is host code, but this puts the bit-exact radix
    replay into the same record as the device kernels."""
    for name in sorted(TRACES.keys()):
        test_trace(name)


def test_all_values_flatten_and_pretty_print(capsys):
    tree = RadixCache(None, Alloc(), 1, Judge(), compat_fork=False, get_epoch_fn=lambda: 0)
    tree.insert([1, 2, 3, 4], torch.tensor([10, 11, 12, 13]))
    tree.insert([1, 2, 7, 8, 9], torch.tensor([10, 11, 20, 21, 22]))
    assert sorted(tree.all_values_flatten().tolist()) == [10, 11, 12, 13, 20, 21, 22]
    tree.pretty_print()
    out = capsys.readouterr().out
    assert "#tokens: 7" in out and "[1, 2]" in out and "r=0" in out
