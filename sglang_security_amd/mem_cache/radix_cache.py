"""RadixCache on the native prefix tree (libsglk: sglk_radix_*).

Mirror of the reference's python/sglang/srt/mem_cache/radix_cache.py (class RadixCache, :77-653) and
tree_node.py: same constructor, same BasePrefixCache methods (base_prefix_cache.py:5-53), same return types
(int64 slot tensor on the pool device + a node object), same SafeKV plumbing: new nodes are handed to
`private_judge_client.update_privacy(node=, context=, prompt=)` (radix_cache.py:544-548) and the judge thread
may later set `node.private = False` / `node.need_check_privacy = False` (private_client.py:125-140) on the
node objects it holds.

`compat_fork=True` reproduces the fork's behaviour bit for bit, quirks included -- among them a pool leak: an
insert by any user other than "0" stops at the private root and overwrites the child with the same first token,
whose slots are then never freed.  `compat_fork="fixed"` (the default) keeps the fork's semantics and closes that
leak (the orphaned subtree's slots go back to the allocator, at once or when the last request using them ends);
`False` gives the exact-match upstream semantics.  The epoch comes from `get_epoch` (reference: sglang.get_epoch,
global_epoch.py), injectable for tests.
"""
from __future__ import annotations

import ctypes as C
import threading
import time
import weakref
from typing import Callable, Dict, List, Optional, Tuple

import numpy as np
import torch

from .._lib import lib

TIME_WINDOW_LENGTH = 10  # seconds per epoch (python/sglang/global_epoch.py:10)


class GlobalEpoch:
    """python/sglang/global_epoch.py:12-53: an integer bumped every TIME_WINDOW_LENGTH seconds by a daemon
    thread."""

    def __init__(self, window: float = TIME_WINDOW_LENGTH, autostart: bool = False):
        self._epoch = 0
        self._window = window
        self._stop = threading.Event()
        self._thread = None
        if autostart:
            self.start()

    def start(self):
        if self._thread is None:
            self._thread = threading.Thread(target=self._run, daemon=True)
            self._thread.start()

    def _run(self):
        while not self._stop.wait(self._window):
            self._epoch += 1

    def stop(self):
        self._stop.set()

    def get_epoch(self) -> int:
        return self._epoch


_global_epoch = GlobalEpoch()


def get_epoch() -> int:
    return _global_epoch.get_epoch()


class TreeNode:
    """Handle on a native node; attribute names follow tree_node.py:28-79."""

    __slots__ = ("_tree", "id", "prompt", "__weakref__")

    def __init__(self, tree: "RadixCache", node_id: int):
        self._tree = tree
        self.id = node_id
        self.prompt = ""

    def _fields(self):
        f = (C.c_int64 * 12)()
        if lib().sglk_radix_node_get(self._tree._h, self.id, f) != 0:
            raise KeyError(f"node {self.id} has left the tree")
        return list(f)

    @property
    def alive(self) -> bool:
        """False once the node has been evicted, released or lost to a reset (the handle then names nothing)."""
        f = (C.c_int64 * 12)()
        return lib().sglk_radix_node_get(self._tree._h, self.id, f) == 0

    @property
    def parent(self) -> Optional["TreeNode"]:
        p = self._fields()[0]
        return None if p < 0 else self._tree._node(p)

    @property
    def lock_ref(self) -> int:
        return self._fields()[1]

    @property
    def private(self) -> bool:
        return bool(self._fields()[2])

    @private.setter
    def private(self, v: bool):
        # on a node that has left the tree this changes nothing (the reference mutates a dead object there)
        lib().sglk_radix_node_set_privacy(self._tree._h, self.id, 1 if v else 0, -1)

    @property
    def need_check_privacy(self) -> bool:
        return bool(self._fields()[3])

    @need_check_privacy.setter
    def need_check_privacy(self, v: bool):
        lib().sglk_radix_node_set_privacy(self._tree._h, self.id, -1, 1 if v else 0)

    @property
    def owner_id(self):
        return self._tree._user_name(self._fields()[4])

    @property
    def epoch(self) -> int:
        return self._fields()[5]

    @property
    def hit_count(self) -> int:
        return self._fields()[6]

    @property
    def hit_pre(self) -> int:
        return self._fields()[7]

    @property
    def u_cnt_pre(self) -> int:
        return self._fields()[8]

    @property
    def n_users(self) -> int:
        return self._fields()[9]

    def _key_value(self):
        n = self._fields()[10]
        k = np.empty(n, dtype=np.int32)
        v = np.empty(n, dtype=np.int64)
        lib().sglk_radix_node_key(self._tree._h, self.id, k.ctypes.data, v.ctypes.data, n)
        return k, v

    @property
    def key(self) -> List[int]:
        return self._key_value()[0].tolist()

    @property
    def value(self) -> torch.Tensor:
        return torch.from_numpy(self._key_value()[1])

    def __eq__(self, other):
        return isinstance(other, TreeNode) and other.id == self.id and other._tree is self._tree

    def __hash__(self):
        return hash((id(self._tree), self.id))


class RadixCache:
    def __init__(self, req_to_token_pool, token_to_kv_pool_allocator, page_size: int, private_judge_client,
                 disable: bool = False, enable_kv_cache_events: bool = False, compat_fork="fixed",
                 get_epoch_fn: Optional[Callable[[], int]] = None):
        if page_size != 1:
            raise NotImplementedError("the native tree implements page_size == 1 (the reference default, server_args.py:78)")
        self.req_to_token_pool = req_to_token_pool
        self.token_to_kv_pool_allocator = token_to_kv_pool_allocator
        self.page_size = page_size
        self.disable = disable
        self.enable_kv_cache_events = enable_kv_cache_events
        self.private_judge_client = private_judge_client
        self.device = token_to_kv_pool_allocator.device if token_to_kv_pool_allocator else torch.device("cpu")
        self._get_epoch = get_epoch_fn or get_epoch
        mode = {True: 1, False: 0, "fixed": 2}.get(compat_fork)
        if mode is None:
            raise ValueError('compat_fork is True (the fork, bit for bit), "fixed" (its slot leak closed) or False (upstream)')
        self._h = lib().sglk_radix_create(mode)
        self._users: Dict[str, int] = {"0": 0, "None": -2}
        self._user_names: Dict[int, object] = {0: "0", -2: "None", -1: None}
        self._nodes = weakref.WeakValueDictionary()  # handles live as long as somebody holds them
        self.reset()

    def __del__(self):
        try:
            lib().sglk_radix_destroy(self._h)
        except Exception:
            pass

    # ---- plumbing ------------------------------------------------------------------------------
    @staticmethod
    def _key_array(key) -> np.ndarray:
        """Token ids as a contiguous int32 array.  Lists cost a Python-level walk (about 30 ns per token): callers
        that match the same request more than once can pass a numpy array or a tensor and pay it once."""
        if isinstance(key, np.ndarray):
            return np.ascontiguousarray(key, dtype=np.int32)
        if isinstance(key, torch.Tensor):
            return np.ascontiguousarray(key.detach().cpu().numpy(), dtype=np.int32)
        return np.fromiter(key, dtype=np.int32, count=len(key))

    def _uid(self, user_id) -> int:
        if user_id is None:
            return -1
        s = str(user_id)
        u = self._users.get(s)
        if u is None:
            u = len(self._users)  # 2, 3, ... (0 = "0", negative ids are reserved)
            self._users[s] = u
            self._user_names[u] = s
        return u

    def _user_name(self, uid: int):
        return self._user_names.get(uid, uid)

    def _node(self, node_id: int) -> TreeNode:
        n = self._nodes.get(node_id)
        if n is None:
            n = TreeNode(self, node_id)
            self._nodes[node_id] = n
        return n

    def _sync_epoch(self):
        lib().sglk_radix_set_epoch(self._h, int(self._get_epoch()))

    def _drain_freed(self):
        n = lib().sglk_radix_take_freed(self._h, None, 0)
        if n <= 0:
            return
        buf = np.empty(n, dtype=np.int64)
        lib().sglk_radix_take_freed(self._h, buf.ctypes.data, n)
        if self.token_to_kv_pool_allocator is not None:
            self.token_to_kv_pool_allocator.free(torch.from_numpy(buf).to(self.device))

    # ---- public API (BasePrefixCache) ------------------------------------------------------------
    def reset(self):
        lib().sglk_radix_reset(self._h)
        self._nodes.clear()
        self.root_node = self._node(int(lib().sglk_radix_root(self._h)))

    def match_prefix(self, key: List[int], user_id: Optional[str] = None, **kwargs) -> Tuple[torch.Tensor, TreeNode]:
        if self.disable or len(key) == 0:
            return torch.empty((0,), dtype=torch.int64, device=self.device), self.root_node
        self._sync_epoch()
        k = self._key_array(key)
        out = np.empty(len(k), dtype=np.int64)
        last = C.c_int64(0)
        n = lib().sglk_radix_match_prefix(self._h, k.ctypes.data, len(k), self._uid(user_id), out.ctypes.data, len(k),
                                          C.byref(last))
        self._drain_freed()
        value = torch.from_numpy(out[:n].copy()).to(self.device)
        return value, self._node(last.value)

    def insert(self, key: List, value=None, prompt: str = "", user_id: Optional[str] = None) -> int:
        if self.disable:
            return 0
        if value is None:
            value = [x for x in key]
        self._sync_epoch()
        k = self._key_array(key)
        v = value.detach().cpu().numpy() if isinstance(value, torch.Tensor) else np.asarray(value)
        v = np.ascontiguousarray(v.astype(np.int64))
        cap = len(k) // 1024 + 2
        created = (C.c_int64 * cap)()
        n_new = C.c_int64(0)
        r = lib().sglk_radix_insert(self._h, k.ctypes.data, v.ctypes.data, len(k), self._uid(user_id), created, cap,
                                    C.byref(n_new))
        for i in range(n_new.value):
            node = self._node(created[i])
            node.prompt = prompt
            if self.private_judge_client is not None:
                self.private_judge_client.update_privacy(node=node, context=prompt, prompt=prompt)
        return int(r)

    def cache_finished_req(self, req):
        """radix_cache.py:160-197."""
        if self.disable:
            kv_indices = self.req_to_token_pool.req_to_token[
                req.req_pool_idx, : len(req.origin_input_ids) + len(req.output_ids) - 1]
            self.token_to_kv_pool_allocator.free(kv_indices)
            self.req_to_token_pool.free(req.req_pool_idx)
            return
        token_ids = (req.origin_input_ids + req.output_ids)[:-1]
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, : len(token_ids)]
        new_prefix_len = self.insert(token_ids, kv_indices.clone(), getattr(req, "origin_input_text", ""),
                                     user_id=getattr(req, "user_id", None))
        self.token_to_kv_pool_allocator.free(kv_indices[len(req.prefix_indices): new_prefix_len])
        self.req_to_token_pool.free(req.req_pool_idx)
        self.dec_lock_ref(req.last_node)

    def cache_unfinished_req(self, req):
        """radix_cache.py:199-240 (the re-match is issued without a user id, as in the reference :224)."""
        if self.disable:
            return
        token_ids = req.fill_ids
        kv_indices = self.req_to_token_pool.req_to_token[req.req_pool_idx, : len(token_ids)]
        new_prefix_len = self.insert(token_ids, kv_indices.clone(), getattr(req, "origin_input_text", ""),
                                     user_id=getattr(req, "user_id", None))
        self.token_to_kv_pool_allocator.free(kv_indices[len(req.prefix_indices): new_prefix_len])
        new_indices, new_last_node = self.match_prefix(token_ids)
        self.req_to_token_pool.write(
            (req.req_pool_idx, slice(len(req.prefix_indices), len(new_indices))),
            new_indices[len(req.prefix_indices):])
        self.dec_lock_ref(req.last_node)
        self.inc_lock_ref(new_last_node)
        req.prefix_indices = new_indices
        req.last_node = new_last_node

    def evict(self, num_tokens: int):
        if self.disable:
            return
        lib().sglk_radix_evict(self._h, int(num_tokens))
        self._drain_freed()

    def inc_lock_ref(self, node: TreeNode) -> int:
        if self.disable:
            return 0
        return int(lib().sglk_radix_inc_lock_ref(self._h, node.id))

    def dec_lock_ref(self, node: TreeNode) -> int:
        if self.disable:
            return 0
        return int(lib().sglk_radix_dec_lock_ref(self._h, node.id))

    def evictable_size(self) -> int:
        return int(lib().sglk_radix_evictable_size(self._h))

    def protected_size(self) -> int:
        return int(lib().sglk_radix_protected_size(self._h))

    def total_size(self) -> int:
        return int(lib().sglk_radix_total_size(self._h))

    def live_nodes(self) -> int:
        """Nodes held in host memory: the reachable tree plus orphans still locked by a running request."""
        return int(lib().sglk_radix_live_nodes(self._h))

    def take_events(self):
        return []

    def all_values_flatten(self) -> torch.Tensor:
        """radix_cache.py:320-329: every cached slot, parents before children."""
        vals = [torch.tensor(d["value"], dtype=torch.int64) for d in self.dump()]
        out = torch.cat(vals) if vals else torch.empty(0, dtype=torch.int64)
        return out.to(self.device) if self.device is not None else out

    def pretty_print(self) -> None:
        """radix_cache.py:242-244,566-582: one line per node (key length, first ten tokens, lock count), then the size."""
        root_key = self.root_node.key
        print("", len(root_key), root_key[:10], f"r={self.root_node.lock_ref}")
        for d in self.dump():
            print(" " * (2 * (d["depth"] + 1)), len(d["key"]), d["key"][:10], f"r={d['lock_ref']}")
        print(f"#tokens: {self.total_size()}")

    # ---- introspection used by the tests -----------------------------------------------------------
    def dump(self):
        """Pre-order structural dump, children ordered by first token."""
        n = lib().sglk_radix_dump(self._h, None, 0)
        ids = (C.c_int64 * max(n, 1))()
        lib().sglk_radix_dump(self._h, ids, n)
        out = []
        depth = {self.root_node.id: -1}
        for i in range(n):
            node = self._node(ids[i])
            f = node._fields()
            k, v = node._key_value()
            depth[node.id] = depth[f[0]] + 1
            out.append(dict(depth=depth[node.id], key=k.tolist(), value=v.tolist(), lock_ref=f[1], private=bool(f[2]),
                            need_check=bool(f[3]), owner=None if f[4] == -1 else str(self._user_name(f[4])),
                            epoch=f[5], hit_count=f[6], hit_pre=f[7], u_cnt_pre=f[8], n_users=f[9]))
        return out
