// FlatMap.hpp — the small unsigned -> unsigned maps of the adapter views.
#pragma once

#include <algorithm>
#include <cstddef>
#include <cstdint>
#include <utility>
#include <vector>

namespace eacham {
namespace hip {

// unsigned -> unsigned map as a sorted vector: the views below are rebuilt for every call, and as std::map they were one heap
// allocation per keypoint-with-a-point and per observer (tens of thousands per TriangulateFrame of a 100-frame sequence).
// Iterates in ascending key order like std::map; the interface is the part of std::map the walk and the glue use.
struct FlatMap {
    typedef std::pair<unsigned, unsigned> value_type;
    typedef std::vector<value_type>::iterator iterator;
    typedef std::vector<value_type>::const_iterator const_iterator;
    std::vector<value_type> v;
    size_t size() const { return v.size(); }
    bool empty() const { return v.empty(); }
    void clear() { v.clear(); }
    iterator begin() { return v.begin(); }
    iterator end() { return v.end(); }
    const_iterator begin() const { return v.begin(); }
    const_iterator end() const { return v.end(); }
    iterator lower(unsigned k) { return std::lower_bound(v.begin(), v.end(), k, [](const value_type& a, unsigned b) { return a.first < b; }); }
    const_iterator lower(unsigned k) const { return std::lower_bound(v.begin(), v.end(), k, [](const value_type& a, unsigned b) { return a.first < b; }); }
    iterator find(unsigned k) { auto it = lower(k); return it != v.end() && it->first == k ? it : v.end(); }
    const_iterator find(unsigned k) const { auto it = lower(k); return it != v.end() && it->first == k ? it : v.end(); }
    size_t count(unsigned k) const { return find(k) != v.end() ? 1 : 0; }
    unsigned& operator[](unsigned k) {
        auto it = lower(k);
        if (it == v.end() || it->first != k) it = v.insert(it, value_type(k, 0u));
        return it->second;
    }
    size_t erase(unsigned k) {
        auto it = find(k);
        if (it == v.end()) return 0;
        v.erase(it);
        return 1;
    }
    void assign_unsorted(std::vector<value_type>&& items) {   // bulk build: keys unique
        v = std::move(items);
        std::sort(v.begin(), v.end());
    }
    bool operator==(const FlatMap& o) const { return v == o.v; }
};

// unsigned -> uint32 table for ONE call's worth of keys (landmark id -> dense index in the glue's graph walks): open addressing
// with a generation stamp per slot, so that a call clears it by bumping the generation — no allocation and no rehash per call once
// it has grown to the size the calls need (std::unordered_map cost one node allocation per landmark and a pointer chase per
// observation: 12 000 of each per local window).
struct IdTable {
    std::vector<unsigned> key;
    std::vector<uint32_t> val, gen;
    uint32_t cur = 0;
    size_t mask = 0;
    void reset(size_t expected_keys) {
        size_t want = 1024;
        while (want < 2 * expected_keys) want <<= 1;
        if (key.size() < want) {
            key.assign(want, 0u), val.assign(want, 0u), gen.assign(want, 0u);
            cur = 0;
        }
        mask = key.size() - 1;
        if (++cur == 0) {  // the stamp wrapped: every slot is stale by definition
            std::fill(gen.begin(), gen.end(), 0u);
            cur = 1;
        }
    }
    // the value slot of k; `fresh` tells whether k was entered by this call (its value is then the caller's to set)
    uint32_t& slot(unsigned k, bool& fresh) {
        size_t h = ((size_t)k * 2654435761u) & mask;
        while (gen[h] == cur && key[h] != k) h = (h + 1) & mask;
        fresh = gen[h] != cur;
        if (fresh) gen[h] = cur, key[h] = k;
        return val[h];
    }
};

}  // namespace hip
}  // namespace eacham
