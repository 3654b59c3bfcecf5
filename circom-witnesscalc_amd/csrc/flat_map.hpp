// Open-addressing hash map (linear probing, no deletion) from a 128-bit key to a 32-bit value: the value-numbering tables
// of the graph compiler see one insertion or lookup per node, and std::unordered_map's node allocations were half of the
// compile time of multi-million-node graphs.
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
        while (used_[i]) {
            if (keys_[2 * i] == k0 && keys_[2 * i + 1] == k1) {
                if (inserted) *inserted = false;
                return vals_[i];
            }
            i = (i + 1) & (cap_ - 1);
        }
        used_[i] = 1;
        keys_[2 * i] = k0;
        keys_[2 * i + 1] = k1;
        vals_[i] = value;
        ++n_;
        if (inserted) *inserted = true;
        return value;
    }
    bool find(uint64_t k0, uint64_t k1, uint32_t* value) const {
        size_t i = slot(k0, k1);
        while (used_[i]) {
            if (keys_[2 * i] == k0 && keys_[2 * i + 1] == k1) {
                *value = vals_[i];
                return true;
            }
            i = (i + 1) & (cap_ - 1);
        }
        return false;
    }
    size_t size() const { return n_; }

private:
    size_t slot(uint64_t k0, uint64_t k1) const {
        uint64_t h = (k0 ^ (k1 * 0x9E3779B97F4A7C15ull)) * 0xC2B2AE3D27D4EB4Full;
        h ^= h >> 29;
        return (size_t)h & (cap_ - 1);
    }
    void rehash(size_t want) {
        size_t cap = 16;
        while (cap < want) cap <<= 1;
        std::vector<uint64_t> ok;
        std::vector<uint32_t> ov;
        std::vector<uint8_t> ou;
        ok.swap(keys_);
        ov.swap(vals_);
        ou.swap(used_);
        const size_t old_cap = cap_;
        cap_ = cap;
        keys_.assign(2 * cap, 0);
        vals_.assign(cap, 0);
        used_.assign(cap, 0);
        n_ = 0;
        for (size_t i = 0; i < old_cap; ++i)
            if (ou[i]) find_or_insert(ok[2 * i], ok[2 * i + 1], ov[i], nullptr);
    }
    std::vector<uint64_t> keys_;
    std::vector<uint32_t> vals_;
    std::vector<uint8_t> used_;
    size_t cap_ = 0, n_ = 0;
};

}  // namespace cwc
