/**
 *  usearch_amd/csrc/filter.hpp — a caller's predicate as an HBM-resident bitmap, made once and used by any number of batches.
 *
 *  The reference takes `predicate(member) -> bool` as a host callable and runs it inside the traversal, where a candidate is
 *  about to enter `top` (/root/reference/include/usearch/index.hpp:4200-4205, 4236-4240; index_dense.hpp:2071-2084 wraps the
 *  caller's `predicate(key)`), and inside the brute-force scan (index.hpp:4260-4263). A host function cannot run on the device,
 *  so the predicate travels as ONE BIT PER SLOT and the kernels test the bit at exactly those places. `usearch_filtered_search`
 *  (c/usearch.h:391-395) has to evaluate the callback over every member per call to get there; the objects here are how a caller
 *  who can say what the predicate IS — a range of keys, a set of keys, a bitmap of its own — skips that: the bitmap is built by
 *  a kernel over the snapshot's `keys[]` and stays in HBM.
 */
#pragma once
#include <cstddef>
#include <cstdint>
#include <memory>

#include "engine.hpp"

namespace usearch_amd {

class filter_t {
  public:
    filter_t() = default;
    ~filter_t();
    filter_t(const filter_t&) = delete;
    filter_t& operator=(const filter_t&) = delete;

    /// Bit `s & 31` of word `s >> 5` = slot `s` passes; `words` must cover every member. Tombstones never pass.
    static const char* from_bits(snapshot_t& snapshot, const std::uint32_t* bits_host, std::size_t words,
                                 std::unique_ptr<filter_t>& out);
    /// Members whose key lies in [first, last] (both ends included).
    static const char* from_key_range(snapshot_t& snapshot, std::uint64_t first, std::uint64_t last, std::unique_ptr<filter_t>& out);
    /// Members whose key is (`allow`) or is not (`!allow`) among `keys[0 .. count)`.
    static const char* from_keys(snapshot_t& snapshot, const std::uint64_t* keys, std::size_t count, bool allow,
                                 std::unique_ptr<filter_t>& out);

    const std::uint32_t* bits() const { return d_bits_; } ///< device pointer
    std::uint64_t members() const { return members_; }    ///< slots the bitmap covers = the snapshot's size when it was made
    std::uint64_t allowed() const { return allowed_; }    ///< members that pass
    const snapshot_t* owner() const { return owner_; }
    /// nullptr when this filter describes `snapshot` as it is now.
    const char* check(const snapshot_t& snapshot) const;

  private:
    const char* allocate(snapshot_t& snapshot);
    const char* finish(hipStream_t stream);

    const snapshot_t* owner_ = nullptr;
    std::uint32_t* d_bits_ = nullptr;
    unsigned long long* d_allowed_ = nullptr;
    std::uint64_t members_ = 0, allowed_ = 0, mutations_ = 0; ///< … and the snapshot's mutation count when it was made
    int device_ = 0;
};

} // namespace usearch_amd
