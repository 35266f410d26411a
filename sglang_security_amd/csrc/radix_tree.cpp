// Native prefix tree behind RadixCache (host code, no GPU).
//
// Restates python/sglang/srt/mem_cache/radix_cache.py:77-653 + tree_node.py:28-79 of the reference with
// token ids in contiguous int32 arrays (memcmp-style compare instead of a Python zip loop), node values as
// int64 slot arrays, children in an insertion-ordered map (the reference's dict order decides eviction ties
// and which child `_delete_leaf` falls back to), and CPython's heapq algorithm reproduced so that equal-epoch
// evictions come out in the same order.
//
// compat = 1 reproduces the fork bit for bit:
//   * fuzzy key match tolerating 11 mismatching tokens (radix_cache.py:52-62)
//   * every node private by default, root owner "0": an insert by any other user stops at the root and
//     OVERWRITES the existing child of the same first token (radix_cache.py:510-514,556)
//   * new nodes cut into chunks of 1024 tokens (radix_cache.py:536-541)
//   * per-node hit / user counters rolled over per epoch and the entropy-triggered subtree release
//     (radix_cache.py:377-411)
// compat = 2: the fork's semantics with its slot leak closed: a subtree orphaned by the root overwrite gives its
//   unlocked slots back (locked nodes do when their last lock goes) and leaves the size counters.
// compat = 0: exact-match keys, no ownership gate, no chunking (upstream RadixCache semantics), same epoch LRU.
//
// Node handles are (slot generation << 32) | slot index.  A node that leaves the tree (eviction, hit-rate release,
// reset, an orphaned subtree once unlocked) is destroyed and its slot's generation bumped, so host memory follows
// the live tree and a handle kept past its node (a privacy verdict that arrives late, a request's last_node after
// a reset) resolves to nothing instead of to whichever node took the slot over.
//
// User ids are int64 handles chosen by the host: -1 = None, 0 = the string "0" (== the root's integer owner
// under str()), -2 = the literal string "None" (what str(None) compares equal to in radix_cache.py:513).
#include <algorithm>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <set>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "sglk.h"

namespace {

constexpr int64_t kNone = -1;
constexpr int64_t kNoneString = -2;
constexpr int kThreshold = 10;  // radix_cache.py:49
constexpr int kChunk = 1024;    // radix_cache.py:536

struct Node;

// dict with Python's ordering rules: overwrite keeps position, delete removes, new key appends
struct Children {
  std::vector<std::pair<int32_t, Node*>> items;
  Node* find(int32_t k) const {
    for (auto& it : items)
      if (it.first == k) return it.second;
    return nullptr;
  }
  void set(int32_t k, Node* n) {
    for (auto& it : items)
      if (it.first == k) {
        it.second = n;
        return;
      }
    items.emplace_back(k, n);
  }
  void erase_at(size_t i) { items.erase(items.begin() + (ptrdiff_t)i); }
  bool empty() const { return items.empty(); }
  size_t size() const { return items.size(); }
};

struct Node {
  int64_t id = 0;
  Children children;
  Node* parent = nullptr;
  std::vector<int32_t> key;
  std::vector<int64_t> value;
  int64_t lock_ref = 0;
  int64_t hit_count = 0;
  bool is_private = true;           // tree_node.py:50
  bool need_check_privacy = true;   // tree_node.py:51
  int64_t owner = 0;                // tree_node.py:52 (integer 0)
  std::shared_ptr<std::set<int64_t>> users;  // u_cnt_l; _split_node shares the set object (radix_cache.py:433)
  int64_t hit_pre = 0;
  int64_t u_cnt_pre = 0;
  int64_t epoch = 0;
  bool orphan = false;          // cut off from the root, kept only because a request still locks it
  bool slots_returned = false;  // its slots went back to the pool when it left the tree
  bool value_filtered = false;  // `owned` (not `value`) lists the slots this node may still give back
  std::vector<int64_t> owned;   // value minus the slots the overwriting node took over (compat 2 only)
};

struct Slot {
  std::unique_ptr<Node> node;
  uint32_t gen = 0;
};

struct Tree {
  int compat = 1;
  int64_t epoch = 0;
  std::vector<Slot> slots;
  std::vector<uint32_t> free_idx;
  std::vector<Node*> dead;   // left the tree during the current call: destroyed when the call ends
  std::vector<Node*> grace;  // left the tree but handed to the caller as match_prefix's last node (the hit-rate
                             // release can cut off the very node it returns, radix_cache.py:385-390): kept until
                             // the next call that changes the tree, and beyond it for as long as it is locked
  Node* root = nullptr;
  int64_t evictable = 0, protected_ = 0;
  std::vector<int64_t> freed;
  std::mutex mu;  // the judge thread flips privacy bits while the scheduler thread walks the tree

  Node* make() {
    uint32_t idx;
    if (!free_idx.empty()) {
      idx = free_idx.back();
      free_idx.pop_back();
    } else {
      idx = (uint32_t)slots.size();
      slots.emplace_back();
    }
    slots[idx].node.reset(new Node());
    Node* n = slots[idx].node.get();
    n->id = ((int64_t)slots[idx].gen << 32) | (int64_t)idx;
    n->users = std::make_shared<std::set<int64_t>>();
    n->epoch = epoch;  // tree_node.py:56
    return n;
  }
  void release(Node* n) { dead.push_back(n); }
  void flush_dead() {
    for (Node* n : dead) {
      const uint32_t idx = (uint32_t)(n->id & 0xFFFFFFFF);
      slots[idx].node.reset();
      slots[idx].gen += 1;
      free_idx.push_back(idx);
    }
    dead.clear();
  }
  // start of every call that changes the tree: nodes kept for the previous caller go, unless locked meanwhile
  void settle_grace() {
    for (Node* n : grace) {
      if (n->lock_ref > 0)
        n->orphan = true;
      else
        release(n);
    }
    grace.clear();
  }
  void reset() {
    dead.clear();
    grace.clear();
    free_idx.clear();
    for (size_t i = slots.size(); i-- > 0;) {
      if (slots[i].node) {
        slots[i].node.reset();
        slots[i].gen += 1;
      }
      free_idx.push_back((uint32_t)i);  // slot 0 is handed out first: the root
    }
    root = make();
    root->lock_ref = 1;
    evictable = protected_ = 0;
    freed.clear();
  }
  Node* get(int64_t id) {
    const uint64_t idx = (uint64_t)id & 0xFFFFFFFFull;
    if (id < 0 || idx >= slots.size()) return nullptr;
    Slot& s = slots[(size_t)idx];
    return (s.node && (int64_t)s.gen == (id >> 32)) ? s.node.get() : nullptr;
  }
  size_t live_nodes() const { return slots.size() - free_idx.size(); }

  // A subtree that an overwrite cut off from the root (the fork's private-root quirk).  Unlocked nodes go now,
  // locked ones when their last lock goes; with compat == 2 their slots return to the pool and their sizes leave
  // the counters (compat == 1 keeps both, as the fork does).  A request that matched the old child and then
  // finishes inserts its WHOLE slot row (cache_finished_req, radix_cache.py:160-197: insert() returns 0 at the
  // private root, nothing is freed), so the overwriting node's value names the old child's slots again: those
  // now belong to the new node and must not go back to the pool with the old one.
  void drop_orphan_node(Node* x) {
    if (compat == 2 && !x->slots_returned) {
      free_slots(x->value_filtered ? x->owned : x->value);
      evictable -= (int64_t)x->value.size();
    }
    release(x);
  }
  void orphan_subtree(Node* top, const int64_t* new_val, int64_t new_len) {
    std::unordered_set<int64_t> taken;
    if (compat == 2) taken.insert(new_val, new_val + new_len);
    std::vector<Node*> st{top};
    while (!st.empty()) {
      Node* x = st.back();
      st.pop_back();
      for (auto& it : x->children.items) st.push_back(it.second);
      if (compat == 2 && !x->slots_returned) {
        x->owned.clear();
        for (int64_t s : x->value)
          if (!taken.count(s)) x->owned.push_back(s);
        x->value_filtered = true;
      }
      if (x->lock_ref > 0) {
        x->orphan = true;  // a running request still walks up from here (never down: forget the children)
        x->children.items.clear();
      } else {
        drop_orphan_node(x);
      }
    }
  }

  // _key_match_page_size1 (radix_cache.py:52-62)
  int64_t key_match(const int32_t* a, int64_t na, const int32_t* b, int64_t nb) const {
    const int64_t n = std::min(na, nb);
    if (!compat) {
      int64_t i = 0;
      while (i < n && a[i] == b[i]) ++i;
      return i;
    }
    int64_t i = 0;
    int miss = 0;
    for (; i < n; ++i) {
      if (a[i] != b[i]) {
        if (miss > 10) break;
        ++miss;
      }
    }
    return i;
  }

  void free_slots(const std::vector<int64_t>& v) { freed.insert(freed.end(), v.begin(), v.end()); }

  // _delete_leaf (radix_cache.py:584-589): falls back to the LAST child when the node is not found
  void delete_leaf(Node* n) {
    Node* p = n->parent;
    if (p && !p->children.empty()) {
      size_t idx = p->children.size() - 1;
      for (size_t i = 0; i < p->children.size(); ++i)
        if (p->children.items[i].second == n) {
          idx = i;
          break;
        }
      p->children.erase_at(idx);
    }
    evictable -= (int64_t)n->key.size();
  }

  // _split_node (radix_cache.py:413-447)
  Node* split(Node* child, int64_t split_len) {
    Node* nn = make();
    nn->children.set(child->key[(size_t)split_len], child);
    nn->parent = child->parent;
    nn->lock_ref = child->lock_ref;
    nn->key.assign(child->key.begin(), child->key.begin() + split_len);
    nn->value.assign(child->value.begin(), child->value.begin() + split_len);
    nn->owner = child->owner;
    nn->is_private = child->is_private;  // both branches of radix_cache.py:424-428
    nn->users = child->users;            // shared set object
    nn->hit_pre = child->hit_pre;
    nn->u_cnt_pre = child->u_cnt_pre;
    nn->hit_count = child->hit_count;
    const int32_t first = child->key[0];
    child->parent = nn;
    child->key.erase(child->key.begin(), child->key.begin() + split_len);
    child->value.erase(child->value.begin(), child->value.begin() + split_len);
    nn->parent->children.set(first, nn);
    return nn;
  }

  // _free_with_entropy (radix_cache.py:393-411)
  void free_with_entropy(Node* node) {
    if (node->u_cnt_pre == 0 || node->users->empty()) return;  // the reference would raise ZeroDivisionError
    const double pre = (double)node->hit_pre / (double)node->u_cnt_pre;
    const double cur = (double)node->hit_count / (double)node->users->size();
    if (cur < kThreshold * pre) return;
    std::vector<Node*> rel{node};
    while (!rel.empty()) {
      Node* x = rel.back();
      rel.pop_back();
      if (x->lock_ref > 0) continue;
      for (auto& it : x->children.items) rel.push_back(it.second);
      free_slots(x->value);
      x->slots_returned = true;
      delete_leaf(x);
      release(x);
      for (Node* g : grace)  // a node kept for an earlier caller must not walk up into a destroyed parent
        if (g->parent == x) g->parent = nullptr;
    }
  }

  // _match_prefix_helper (radix_cache.py:333-390)
  int64_t match(const int32_t* key, int64_t len, int64_t user, std::vector<int64_t>& out, Node** last) {
    Node* node = root;
    node->epoch = epoch;
    if (len == 0) {
      *last = root;
      return 0;
    }
    int32_t child_key = key[0];
    while (len > 0) {
      Node* child = node->children.find(child_key);
      if (!child) break;
      if (compat && child->is_private && user != kNone && child->owner != user) break;
      const int64_t pl = key_match(child->key.data(), (int64_t)child->key.size(), key, len);
      if (pl < (int64_t)child->key.size()) {
        Node* nn = split(child, pl);
        out.insert(out.end(), nn->value.begin(), nn->value.end());
        node = nn;
        break;  // the reference leaves the loop before touching the hit counters
      }
      out.insert(out.end(), child->value.begin(), child->value.end());
      node = child;
      key += pl;
      len -= pl;
      if (len) child_key = key[0];
      if (compat) {
        if (child->epoch < epoch) {  // next time window
          child->u_cnt_pre = (int64_t)child->users->size();
          child->users = std::make_shared<std::set<int64_t>>();
          child->users->insert(user);
          child->hit_pre = child->hit_count;
          child->hit_count = 0;
        } else {
          child->users->insert(user);
          child->hit_count += 1;
          if (child->hit_pre != 0 && child->hit_count > child->hit_pre * kThreshold) free_with_entropy(child);
        }
      }
      child->epoch = epoch;
    }
    *last = node;
    auto it = std::find(dead.begin(), dead.end(), node);
    if (it != dead.end()) {  // the release above cut off the node being returned
      dead.erase(it);
      grace.push_back(node);
    }
    return (int64_t)out.size();
  }

  static bool insert_owner_mismatch(int64_t user, int64_t owner) {
    // `user_id != str(node.owner_id)` (radix_cache.py:513): None never equals a string;
    // str(None) == "None"; str(0) == "0"
    if (user == kNone) return true;
    if (owner == kNone) return user != kNoneString;
    return user != owner;
  }

  // _insert_helper (radix_cache.py:501-564)
  int64_t insert(const int32_t* key, const int64_t* val, int64_t len, int64_t user, std::vector<int64_t>& created) {
    Node* node = root;
    node->epoch = epoch;
    if (len == 0) return 0;
    int32_t child_key = key[0];
    int64_t total = 0;
    while (len > 0) {
      Node* child = node->children.find(child_key);
      if (!child) break;
      if (compat && node->is_private && insert_owner_mismatch(user, node->owner)) break;
      node = child;
      node->epoch = epoch;
      const int64_t pl = key_match(node->key.data(), (int64_t)node->key.size(), key, len);
      total += pl;
      key += pl;
      val += pl;
      len -= pl;
      if (pl < (int64_t)node->key.size()) node = split(node, pl);
      if (len) child_key = key[0];
    }
    if (len) {
      const int64_t chunk = compat ? kChunk : len;
      while (len > 0) {
        const int64_t n = std::min(chunk, len);
        Node* nn = make();
        nn->parent = node;
        nn->key.assign(key, key + n);
        nn->value.assign(val, val + n);
        nn->owner = user;
        if (Node* old = node->children.find(child_key)) orphan_subtree(old, val, len);  // the root quirk: overwritten child
        node->children.set(child_key, nn);
        evictable += n;
        created.push_back(nn->id);
        node = nn;
        key += n;
        val += n;
        len -= n;
        if (len) child_key = key[0];
      }
    }
    return total;
  }

  // heapq with `<` = epoch order (tree_node.py:75-78), CPython's sift routines
  static bool lt(Node* a, Node* b) { return a->epoch < b->epoch; }
  static void siftdown(std::vector<Node*>& h, size_t start, size_t pos) {
    Node* item = h[pos];
    while (pos > start) {
      size_t parent = (pos - 1) >> 1;
      if (lt(item, h[parent])) {
        h[pos] = h[parent];
        pos = parent;
        continue;
      }
      break;
    }
    h[pos] = item;
  }
  static void siftup(std::vector<Node*>& h, size_t pos) {
    const size_t end = h.size(), start = pos;
    Node* item = h[pos];
    size_t child = 2 * pos + 1;
    while (child < end) {
      size_t right = child + 1;
      if (right < end && !lt(h[child], h[right])) child = right;
      h[pos] = h[child];
      pos = child;
      child = 2 * pos + 1;
    }
    h[pos] = item;
    siftdown(h, start, pos);
  }

  // evict (radix_cache.py:249-283) with _collect_leaves (:603-617)
  void evict(int64_t num_tokens) {
    std::vector<Node*> leaves, stack{root};
    while (!stack.empty()) {
      Node* cur = stack.back();
      stack.pop_back();
      if (cur->children.empty())
        leaves.push_back(cur);
      else
        for (auto& it : cur->children.items) stack.push_back(it.second);
    }
    for (size_t i = leaves.size() / 2; i-- > 0;) siftup(leaves, i);  // heapify
    int64_t evicted = 0;
    while (evicted < num_tokens && !leaves.empty()) {
      Node* last = leaves.back();
      leaves.pop_back();
      Node* x = last;
      if (!leaves.empty()) {
        x = leaves[0];
        leaves[0] = last;
        siftup(leaves, 0);
      }
      if (x == root) break;
      if (x->lock_ref > 0) continue;
      evicted += (int64_t)x->value.size();
      free_slots(x->value);
      delete_leaf(x);
      release(x);
      if (x->parent->children.empty()) {
        leaves.push_back(x->parent);
        siftdown(leaves, 0, leaves.size() - 1);
      }
    }
  }

  int64_t inc_lock(Node* n) {
    int64_t delta = 0;
    while (n && n != root) {
      if (n->lock_ref == 0) {
        evictable -= (int64_t)n->value.size();
        protected_ += (int64_t)n->value.size();
        delta -= (int64_t)n->value.size();
      }
      n->lock_ref += 1;
      n = n->parent;
    }
    return delta;
  }
  int64_t dec_lock(Node* n) {
    int64_t delta = 0;
    while (n && n != root) {
      if (n->lock_ref == 1) {
        evictable += (int64_t)n->value.size();
        protected_ -= (int64_t)n->value.size();
        delta += (int64_t)n->value.size();
      }
      n->lock_ref -= 1;
      Node* up = n->parent;
      if (n->orphan && n->lock_ref == 0) drop_orphan_node(n);
      n = up;
    }
    return delta;
  }
  int64_t total_size() {
    int64_t t = 0;
    std::vector<Node*> st{root};
    while (!st.empty()) {
      Node* c = st.back();
      st.pop_back();
      t += (int64_t)c->value.size();
      for (auto& it : c->children.items) st.push_back(it.second);
    }
    return t;
  }
  void dump(Node* n, std::vector<int64_t>& ids) {
    std::vector<std::pair<int32_t, Node*>> ch = n->children.items;
    std::sort(ch.begin(), ch.end(), [](auto& a, auto& b) { return a.first < b.first; });
    for (auto& it : ch) {
      ids.push_back(it.second->id);
      dump(it.second, ids);
    }
  }
};

Tree* T(int64_t h) { return reinterpret_cast<Tree*>(h); }

}  // namespace

extern "C" {

int64_t sglk_radix_create(int compat_fork) {
  Tree* t = new Tree();
  t->compat = compat_fork < 0 || compat_fork > 2 ? 1 : compat_fork;
  t->reset();
  return reinterpret_cast<int64_t>(t);
}
void sglk_radix_destroy(int64_t tree) { delete T(tree); }
void sglk_radix_reset(int64_t tree) {
  std::lock_guard<std::mutex> g(T(tree)->mu);
  T(tree)->reset();
}
void sglk_radix_set_epoch(int64_t tree, int64_t epoch) {
  std::lock_guard<std::mutex> g(T(tree)->mu);
  T(tree)->epoch = epoch;
}

int64_t sglk_radix_match_prefix(int64_t tree, const int32_t* key, int64_t key_len, int64_t user_id, int64_t* out_slots,
                                int64_t cap, int64_t* last_node) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  std::vector<int64_t> out;
  Node* last = t->root;
  // Grace nodes are NOT settled here: the scheduler matches many requests before it locks one, and the node a
  // release cut off must still resolve when inc_lock_ref finally arrives (the reference walks the dead node's
  // parent chain and locks the live ancestors, radix_cache.py:285-299).  Insert, evict and reset settle them.
  const int64_t n = t->match(key, key_len, user_id, out, &last);
  if (last_node) *last_node = last->id;
  t->flush_dead();
  const int64_t m = std::min(n, cap);
  if (m > 0 && out_slots) std::memcpy(out_slots, out.data(), sizeof(int64_t) * (size_t)m);
  return n;
}

int64_t sglk_radix_insert(int64_t tree, const int32_t* key, const int64_t* slots, int64_t len, int64_t user_id,
                          int64_t* new_nodes, int64_t new_cap, int64_t* n_new) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  std::vector<int64_t> created;
  t->settle_grace();
  const int64_t r = t->insert(key, slots, len, user_id, created);
  t->flush_dead();
  if (n_new) *n_new = (int64_t)created.size();
  for (int64_t i = 0; i < (int64_t)created.size() && i < new_cap; ++i) new_nodes[i] = created[(size_t)i];
  return r;
}

int64_t sglk_radix_inc_lock_ref(int64_t tree, int64_t node) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  return t->inc_lock(t->get(node));  // a stale handle resolves to nothing: no-op
}
int64_t sglk_radix_dec_lock_ref(int64_t tree, int64_t node) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  const int64_t d = t->dec_lock(t->get(node));
  t->flush_dead();
  return d;
}
void sglk_radix_evict(int64_t tree, int64_t num_tokens) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  t->settle_grace();
  t->evict(num_tokens);
  t->flush_dead();
}
int64_t sglk_radix_take_freed(int64_t tree, int64_t* out, int64_t cap) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  const int64_t n = (int64_t)t->freed.size();
  if (out == nullptr) return n;
  const int64_t m = std::min(n, cap);
  if (m > 0) std::memcpy(out, t->freed.data(), sizeof(int64_t) * (size_t)m);
  t->freed.erase(t->freed.begin(), t->freed.begin() + m);
  return m;
}
int64_t sglk_radix_evictable_size(int64_t tree) { return T(tree)->evictable; }
int64_t sglk_radix_protected_size(int64_t tree) { return T(tree)->protected_; }
int64_t sglk_radix_root(int64_t tree) {
  std::lock_guard<std::mutex> g(T(tree)->mu);
  return T(tree)->root->id;
}
int64_t sglk_radix_live_nodes(int64_t tree) {
  std::lock_guard<std::mutex> g(T(tree)->mu);
  return (int64_t)T(tree)->live_nodes();
}
int64_t sglk_radix_total_size(int64_t tree) {
  std::lock_guard<std::mutex> g(T(tree)->mu);
  return T(tree)->total_size();
}

int sglk_radix_node_set_privacy(int64_t tree, int64_t node, int is_private, int need_check) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  Node* n = t->get(node);
  if (!n) return SGLK_EINVAL;
  if (is_private >= 0) n->is_private = is_private != 0;
  if (need_check >= 0) n->need_check_privacy = need_check != 0;
  return SGLK_OK;
}

// fields: parent id (-1 none), lock_ref, private, need_check_privacy, owner, epoch, hit_count, hit_pre,
//         u_cnt_pre, len(u_cnt_l), len(key), number of children
int sglk_radix_node_get(int64_t tree, int64_t node, int64_t* f) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  Node* n = t->get(node);
  if (!n) return SGLK_EINVAL;
  f[0] = n->parent ? n->parent->id : -1;
  f[1] = n->lock_ref;
  f[2] = n->is_private;
  f[3] = n->need_check_privacy;
  f[4] = n->owner;
  f[5] = n->epoch;
  f[6] = n->hit_count;
  f[7] = n->hit_pre;
  f[8] = n->u_cnt_pre;
  f[9] = (int64_t)n->users->size();
  f[10] = (int64_t)n->key.size();
  f[11] = (int64_t)n->children.size();
  return SGLK_OK;
}

int64_t sglk_radix_node_key(int64_t tree, int64_t node, int32_t* key_out, int64_t* val_out, int64_t cap) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  Node* n = t->get(node);
  if (!n) return -1;
  const int64_t len = (int64_t)n->key.size();
  const int64_t m = std::min(len, cap);
  if (key_out && m > 0) std::memcpy(key_out, n->key.data(), sizeof(int32_t) * (size_t)m);
  if (val_out && m > 0) std::memcpy(val_out, n->value.data(), sizeof(int64_t) * (size_t)std::min<int64_t>(m, (int64_t)n->value.size()));
  return len;
}

int64_t sglk_radix_dump(int64_t tree, int64_t* node_ids, int64_t cap) {
  Tree* t = T(tree);
  std::lock_guard<std::mutex> g(t->mu);
  std::vector<int64_t> ids;
  t->dump(t->root, ids);
  for (int64_t i = 0; i < (int64_t)ids.size() && i < cap; ++i) node_ids[i] = ids[(size_t)i];
  return (int64_t)ids.size();
}
}
