// Open-addressing hash maps (linear probing, no deletion) from a 128-bit or a 64-bit key to a 32-bit value: the
// value-numbering tables of the graph compiler see one insertion or lookup per node, and std::unordered_map's node
// allocations were half of the compile time of multi-million-node graphs.  Key, value and the occupied flag of a slot
// sit in ONE record (24 / 16 bytes): the tables of a multi-million-node graph are far larger than the caches, and with
// three parallel arrays every probe was three misses (10.5 M nodes: tree-height reduction 27 -> 12 s on the build
// container).
#pragma once
#include <stdint.h>

#include <vector>

namespace cwc {

class FlatMap128 {
public:
    explicit FlatMap128(size_t expected = 16) { rehash(expected * 2 + 16); }
    // returns the stored value for (k0, k1), inserting `value` when the key is new (second = true then)
    uint32_t find_or_insert(uint64_t k0, uint64_t k1, uint32_t value, bool* inserted) {
        if ((n_ + 1) * 10 > cap_ * 7) rehash(cap_ * 2);
        size_t i = slot(k0, k1);
        while (e_[i].used) {
            if (e_[i].k0 == k0 && e_[i].k1 == k1) {
                if (inserted) *inserted = false;
                return e_[i].val;
            }
            i = (i + 1) & (cap_ - 1);
        }
        e_[i] = Entry{k0, k1, value, 1u};
        ++n_;
        if (inserted) *inserted = true;
        return value;
    }
    bool find(uint64_t k0, uint64_t k1, uint32_t* value) const {
        size_t i = slot(k0, k1);
        while (e_[i].used) {
            if (e_[i].k0 == k0 && e_[i].k1 == k1) {
                *value = e_[i].val;
                return true;
            }
            i = (i + 1) & (cap_ - 1);
        }
        return false;
    }
    size_t size() const { return n_; }

private:
    struct Entry {
        uint64_t k0, k1;
        uint32_t val, used;
    };
    size_t slot(uint64_t k0, uint64_t k1) const {
        uint64_t h = (k0 ^ (k1 * 0x9E3779B97F4A7C15ull)) * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 29;
        return (size_t)h & (cap_ - 1);
    }
    void rehash(size_t want) {
        size_t cap = 16;
        while (cap < want) cap <<= 1;
        std::vector<Entry> old;
        old.swap(e_);
        cap_ = cap;
        e_.assign(cap, Entry{0, 0, 0, 0});
        n_ = 0;
        for (const Entry& o : old)
            if (o.used) find_or_insert(o.k0, o.k1, o.val, nullptr);
    }
    std::vector<Entry> e_;
    size_t cap_ = 0, n_ = 0;
};

// The same with a 64-bit key (an operand pair): 16-byte records, never across a cache line.
class FlatMap64 {
public:
    explicit FlatMap64(size_t expected = 16) { rehash(expected * 2 + 16); }
    uint32_t find_or_insert(uint64_t k, uint32_t value, bool* inserted) {
        if ((n_ + 1) * 10 > cap_ * 7) rehash(cap_ * 2);
        size_t i = slot(k);
        while (e_[i].used) {
            if (e_[i].k == k) {
                if (inserted) *inserted = false;
                return e_[i].val;
            }
            i = (i + 1) & (cap_ - 1);
        }
        e_[i] = Entry{k, value, 1u};
        ++n_;
        if (inserted) *inserted = true;
        return value;
    }
    bool find(uint64_t k, uint32_t* value) const {
        size_t i = slot(k);
        while (e_[i].used) {
            if (e_[i].k == k) {
                *value = e_[i].val;
                return true;
            }
            i = (i + 1) & (cap_ - 1);
        }
        return false;
    }
    size_t size() const { return n_; }

private:
    struct Entry {
        uint64_t k;
        uint32_t val, used;
    };
    size_t slot(uint64_t k) const {
        uint64_t h = k * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 29;
        return (size_t)h & (cap_ - 1);
    }
    void rehash(size_t want) {
        size_t cap = 16;
        while (cap < want) cap <<= 1;
        std::vector<Entry> old;
        old.swap(e_);
        cap_ = cap;
        e_.assign(cap, Entry{0, 0, 0});
        n_ = 0;
        for (const Entry& o : old)
            if (o.used) find_or_insert(o.k, o.val, nullptr);
    }
    std::vector<Entry> e_;
    size_t cap_ = 0, n_ = 0;
};

}  // namespace cwc
