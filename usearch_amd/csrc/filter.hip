/**
 *  usearch_amd/csrc/filter.hip — predicates as HBM-resident bitmaps (filter.hpp). One thread per slot, a 64-lane ballot per wave
 *  turns the per-slot verdicts into two words of the bitmap; `keys[]` is read once, coalesced.
 */
#include "filter.hpp"

#include <algorithm>
#include <vector>

#include "host_util.hpp"

namespace usearch_amd {

namespace {

constexpr unsigned filter_threads_k = 256;

/// Writes the verdicts of the wave's 64 slots as two words and adds the wave's count to `allowed`.
__device__ inline void emit_words(bool pass, std::uint64_t slot, std::uint64_t members, std::uint32_t* bits, unsigned long long* allowed) {
    const unsigned long long mask = __ballot(pass);
    if ((threadIdx.x & 63u) == 0 && slot < members) {
        bits[slot >> 5] = (std::uint32_t)mask;
        if (slot + 32 < members)
            bits[(slot >> 5) + 1] = (std::uint32_t)(mask >> 32);
        if (mask)
            atomicAdd(allowed, (unsigned long long)__popcll(mask));
    }
}

__global__ __launch_bounds__(filter_threads_k) void filter_range_kernel(const std::uint64_t* keys, std::uint64_t members, std::uint64_t first,
                                                                        std::uint64_t last, std::uint32_t* bits,
                                                                        unsigned long long* allowed) {
    const std::uint64_t slot = (std::uint64_t)blockIdx.x * filter_threads_k + threadIdx.x;
    bool pass = false;
    if (slot < members) {
        const std::uint64_t key = keys[slot];
        pass = key != free_key_k && key >= first && key <= last;
    }
    emit_words(pass, slot, members, bits, allowed);
}

/// `listed` is sorted ascending; a member passes when its key is (allow) / is not (!allow) in the list.
__global__ __launch_bounds__(filter_threads_k) void filter_keys_kernel(const std::uint64_t* keys, std::uint64_t members,
                                                                       const std::uint64_t* listed, std::uint64_t count, std::uint32_t allow,
                                                                       std::uint32_t* bits, unsigned long long* allowed) {
    const std::uint64_t slot = (std::uint64_t)blockIdx.x * filter_threads_k + threadIdx.x;
    bool pass = false;
    if (slot < members) {
        const std::uint64_t key = keys[slot];
        std::uint64_t low = 0, high = count; // first element not below the key
        while (low < high) {
            const std::uint64_t middle = low + (high - low) / 2;
            if (listed[middle] < key)
                low = middle + 1;
            else
                high = middle;
        }
        const bool found = low < count && listed[low] == key;
        pass = key != free_key_k && (found == (allow != 0));
    }
    emit_words(pass, slot, members, bits, allowed);
}

/// A caller's bitmap: tombstones are cleared, bits beyond the last member too, and the rest is counted.
__global__ __launch_bounds__(filter_threads_k) void filter_bits_kernel(const std::uint64_t* keys, std::uint64_t members,
                                                                       const std::uint32_t* given, std::uint32_t* bits,
                                                                       unsigned long long* allowed) {
    const std::uint64_t slot = (std::uint64_t)blockIdx.x * filter_threads_k + threadIdx.x;
    bool pass = false;
    if (slot < members)
        pass = ((given[slot >> 5] >> (slot & 31)) & 1u) != 0 && keys[slot] != free_key_k;
    emit_words(pass, slot, members, bits, allowed);
}

unsigned blocks_for(std::uint64_t members) { return (unsigned)((members + filter_threads_k - 1) / filter_threads_k); }

} // namespace

filter_t::~filter_t() {
    if (d_bits_ || d_allowed_) {
        (void)hipSetDevice(device_);
        if (d_bits_)
            (void)hipFree(d_bits_);
        if (d_allowed_)
            (void)hipFree(d_allowed_);
    }
}

const char* filter_t::allocate(snapshot_t& snapshot) {
    owner_ = &snapshot;
    device_ = snapshot.device();
    members_ = snapshot.view().size;
    mutations_ = snapshot.mutations();
    if (members_ >= none_slot_k)
        return "Index is too large for 32-bit slots";
    UA_HIP(hipSetDevice(device_));
    const std::size_t words = (std::size_t)((members_ + 31) / 32) + 2; // the kernels write whole pairs of words
    UA_HIP(hipMalloc((void**)&d_bits_, words * 4));
    UA_HIP(hipMalloc((void**)&d_allowed_, 8));
    UA_HIP(hipMemsetAsync(d_bits_, 0, words * 4, snapshot.stream()));
    UA_HIP(hipMemsetAsync(d_allowed_, 0, 8, snapshot.stream()));
    return nullptr;
}

const char* filter_t::finish(hipStream_t stream) {
    UA_HIP(hipGetLastError());
    unsigned long long allowed = 0;
    UA_HIP(hipMemcpyAsync(&allowed, d_allowed_, 8, hipMemcpyDeviceToHost, stream));
    UA_HIP(hipStreamSynchronize(stream));
    allowed_ = allowed;
    return nullptr;
}

const char* filter_t::check(const snapshot_t& snapshot) const {
    if (owner_ != &snapshot)
        return "The filter was made for another index";
    // the count alone would miss a rename, a tombstone or a recycled slot (builder_t::update, set_key): same size, other keys
    if (members_ != snapshot.view().size || mutations_ != snapshot.mutations())
        return "The index changed since the filter was made";
    return nullptr;
}

const char* filter_t::from_bits(snapshot_t& snapshot, const std::uint32_t* bits_host, std::size_t words, std::unique_ptr<filter_t>& out) {
    std::unique_ptr<filter_t> filter(new filter_t());
    if (const char* e = filter->allocate(snapshot))
        return e;
    const std::size_t needed = (std::size_t)((filter->members_ + 31) / 32);
    if (words < needed || (needed && !bits_host))
        return "The bitmap does not cover every member";
    if (filter->members_) {
        std::uint32_t* given = nullptr;
        UA_HIP(hipMalloc((void**)&given, needed * 4));
        hipError_t e = hipMemcpyAsync(given, bits_host, needed * 4, hipMemcpyHostToDevice, snapshot.stream());
        if (e == hipSuccess) {
            hipLaunchKernelGGL(filter_bits_kernel, dim3(blocks_for(filter->members_)), dim3(filter_threads_k), 0, snapshot.stream(),
                               snapshot.view().keys, filter->members_, (const std::uint32_t*)given, filter->d_bits_, filter->d_allowed_);
        }
        const char* error = e == hipSuccess ? filter->finish(snapshot.stream()) : hip_message(e);
        (void)hipFree(given);
        if (error)
            return error;
    }
    out = std::move(filter);
    return nullptr;
}

const char* filter_t::from_key_range(snapshot_t& snapshot, std::uint64_t first, std::uint64_t last, std::unique_ptr<filter_t>& out) {
    std::unique_ptr<filter_t> filter(new filter_t());
    if (const char* e = filter->allocate(snapshot))
        return e;
    if (filter->members_) {
        hipLaunchKernelGGL(filter_range_kernel, dim3(blocks_for(filter->members_)), dim3(filter_threads_k), 0, snapshot.stream(),
                           snapshot.view().keys, filter->members_, first, last, filter->d_bits_, filter->d_allowed_);
        if (const char* e = filter->finish(snapshot.stream()))
            return e;
    }
    out = std::move(filter);
    return nullptr;
}

const char* filter_t::from_keys(snapshot_t& snapshot, const std::uint64_t* keys, std::size_t count, bool allow, std::unique_ptr<filter_t>& out) {
    std::unique_ptr<filter_t> filter(new filter_t());
    if (const char* e = filter->allocate(snapshot))
        return e;
    if (count && !keys)
        return "No keys given";
    if (filter->members_) {
        std::vector<std::uint64_t> sorted(keys, keys + count);
        std::sort(sorted.begin(), sorted.end());
        sorted.erase(std::unique(sorted.begin(), sorted.end()), sorted.end());
        std::uint64_t* listed = nullptr;
        UA_HIP(hipMalloc((void**)&listed, std::max<std::size_t>(sorted.size(), 1) * 8));
        hipError_t e = sorted.empty() ? hipSuccess
                                      : hipMemcpyAsync(listed, sorted.data(), sorted.size() * 8, hipMemcpyHostToDevice, snapshot.stream());
        if (e == hipSuccess) {
            hipLaunchKernelGGL(filter_keys_kernel, dim3(blocks_for(filter->members_)), dim3(filter_threads_k), 0, snapshot.stream(),
                               snapshot.view().keys, filter->members_, (const std::uint64_t*)listed, (std::uint64_t)sorted.size(),
                               allow ? 1u : 0u, filter->d_bits_, filter->d_allowed_);
        }
        const char* error = e == hipSuccess ? filter->finish(snapshot.stream()) : hip_message(e); // waits: `sorted` may go
        (void)hipFree(listed);
        if (error)
            return error;
    }
    out = std::move(filter);
    return nullptr;
}

} // namespace usearch_amd
