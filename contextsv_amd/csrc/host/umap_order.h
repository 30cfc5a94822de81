// umap_order.h — the iteration order of a libstdc++ std::unordered_map<std::string, T>, without the map.
//
// Why it exists: in the reference's split-read pass the ITERATION ORDER of `unordered_map<std::string, PrimaryAlignment>` is
// observable — it is the insertion order of the interval tree and the seed order of the overlap groups
// (src/sv_caller.cpp:216, :224) — and that map holds EVERY primary alignment of a chromosome (6e5 string-keyed nodes for chr1
// at 30x) before all but the ~1 % with a supplementary record are erased again (:183-202). The order of the survivors depends on
// the whole insertion history (which bucket became non-empty when, every rehash), so it cannot be had from the survivors alone.
// This class replays exactly what libstdc++'s _Hashtable does to its node list — _M_insert_unique_node / _M_insert_bucket_begin /
// _M_rehash_aux(unique keys) of <bits/hashtable.h>, with the library's OWN _Prime_rehash_policy object deciding when and to
// what size to rehash — on 16 bytes per key instead of a heap node with two strings: ~20x faster, and one instance per
// chromosome runs in its own thread. erase() never reorders the remaining nodes, so the survivors' order is this order filtered.
// tests/test_umap_order.py checks it against the real container (random and adversarial key sets, duplicate keys, sizes across
// many rehashes).
#pragma once
#include <cstdint>
#include <string_view>
#include <unordered_map>
#include <vector>

namespace csvhost {

// std::hash<std::string>: the standard guarantees hash<string_view> gives the same value
inline uint64_t std_string_hash(const char *p, size_t n) { return (uint64_t)std::hash<std::string_view>{}(std::string_view(p, n)); }

class UMapOrder {
public:
    UMapOrder() : bkt_(1, kEmpty) {}
    void reserve(size_t n) { hash_.reserve(n); next_.reserve(n); }
    void clear() { hash_.clear(); next_.clear(); bkt_.assign(1, kEmpty); head_ = -1; n_bkt_ = 1; policy_ = std::__detail::_Prime_rehash_policy(); }
    size_t size() const { return hash_.size(); }

    // operator[] / find: the node holding the key with hash h for which same(node) is true, or -1. same() is only asked about
    // nodes whose cached hash equals h (as _M_equals does).
    template <class Same>
    int64_t find(uint64_t h, Same same) const
    {
        const size_t b = (size_t)(h % n_bkt_);
        const int32_t before = bkt_[b];
        if (before == kEmpty) return -1;
        for (int32_t p = before == kBeforeBegin ? head_ : next_[(size_t)before];; ) {
            if (hash_[(size_t)p] == h && same((uint32_t)p)) return p;
            const int32_t nx = next_[(size_t)p];
            if (nx < 0 || (size_t)(hash_[(size_t)nx] % n_bkt_) != b) return -1;
            p = nx;
        }
    }

    // Insert a key known to be absent (call find first); returns its node id (ids count up from 0 in insertion order).
    uint32_t insert_new(uint64_t h)
    {
        const std::pair<bool, std::size_t> grow = policy_._M_need_rehash(n_bkt_, hash_.size(), 1);
        if (grow.first) rehash(grow.second);
        const uint32_t node = (uint32_t)hash_.size();
        hash_.push_back(h);
        next_.push_back(-1);
        const size_t b = (size_t)(h % n_bkt_);
        if (bkt_[b] != kEmpty) {                        // _M_insert_bucket_begin: first in its bucket
            link_after(bkt_[b], (int32_t)node);
        } else {                                        // empty bucket: the node becomes the head of the whole list
            next_[node] = head_;
            head_ = (int32_t)node;
            if (next_[node] >= 0) bkt_[(size_t)(hash_[(size_t)next_[node]] % n_bkt_)] = (int32_t)node;
            bkt_[b] = kBeforeBegin;
        }
        return node;
    }

    // node ids from begin() to end()
    template <class F>
    void for_each(F f) const { for (int32_t p = head_; p >= 0; p = next_[(size_t)p]) f((uint32_t)p); }

    size_t bucket_count() const { return n_bkt_; }

private:
    static constexpr int32_t kEmpty = -2, kBeforeBegin = -1;

    void link_after(int32_t before, int32_t node)
    {
        if (before == kBeforeBegin) { next_[(size_t)node] = head_; head_ = node; }
        else { next_[(size_t)node] = next_[(size_t)before]; next_[(size_t)before] = node; }
    }

    void rehash(size_t n)                                // _M_rehash_aux(n, true_type)
    {
        std::vector<int32_t> nb(n, kEmpty);
        int32_t p = head_;
        head_ = -1;
        size_t bbegin = 0;
        while (p >= 0) {
            const int32_t nx = next_[(size_t)p];
            const size_t b = (size_t)(hash_[(size_t)p] % n);
            if (nb[b] == kEmpty) {
                next_[(size_t)p] = head_;
                head_ = p;
                nb[b] = kBeforeBegin;
                if (next_[(size_t)p] >= 0) nb[bbegin] = p;
                bbegin = b;
            } else {
                link_after(nb[b], p);
            }
            p = nx;
        }
        bkt_.swap(nb);
        n_bkt_ = n;
    }

    std::vector<uint64_t> hash_;       // cached hash code per node
    std::vector<int32_t> next_;        // singly linked list
    std::vector<int32_t> bkt_;         // per bucket: the node BEFORE its first node (kBeforeBegin: the list head), or kEmpty
    int32_t head_ = -1;
    size_t n_bkt_ = 1;
    std::__detail::_Prime_rehash_policy policy_;
};

}  // namespace csvhost
