// scripts/probes/placement_probe.hip — does WHERE a 15-GB array sits decide how fast random 1536-byte rows can be gathered from it?
// (development tool; scripts/variance_probe.py showed the headline batch at 45.4 or 51.7 ms depending on the allocation alone)
// Allocates the array three ways, a few times each — hipMalloc, hipExtMallocWithFlags(contiguous), virtual-memory API with a
// 1-GB-aligned address — and times the same dependency-free gather on each.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/placement_probe.hip -o scripts/probes/placement_probe.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(call)                                                                                                     \
    do {                                                                                                                \
        hipError_t e_ = (call);                                                                                         \
        if (e_ != hipSuccess) {                                                                                         \
            std::printf("%s -> %s\n", #call, hipGetErrorString(e_));                                                    \
            return 1;                                                                                                   \
        }                                                                                                               \
    } while (0)

__global__ __launch_bounds__(64) void gather_kernel(const uint4* base, std::uint64_t rows, std::uint32_t per_wave, std::uint32_t* out) {
    const std::uint32_t lane = threadIdx.x;
    std::uint64_t state = (blockIdx.x + 1) * 0x9E3779B97F4A7C15ull;
    uint4 acc = {0u, 0u, 0u, 0u};
    for (std::uint32_t i = 0; i < per_wave; i += 2) {
        std::uint64_t picks[2];
        for (int r = 0; r < 2; ++r) {
            state ^= state << 13, state ^= state >> 7, state ^= state << 17;
            picks[r] = (std::uint64_t)(((unsigned __int128)(state & 0xFFFFFFFFFFFFull) * rows) >> 48);
        }
        const uint4* p0 = base + picks[0] * 96;
        const uint4* p1 = base + picks[1] * 96;
        const uint4 a0 = p0[lane], a1 = p1[lane];
        uint4 b0 = {0u, 0u, 0u, 0u}, b1 = {0u, 0u, 0u, 0u};
        if (lane < 32)
            b0 = p0[64 + lane], b1 = p1[64 + lane];
        acc.x ^= a0.x ^ a1.x ^ b0.x ^ b1.x, acc.y ^= a0.y ^ a1.y ^ b0.y ^ b1.y;
        acc.z ^= a0.z ^ a1.z ^ b0.z ^ b1.z, acc.w ^= a0.w ^ a1.w ^ b0.w ^ b1.w;
    }
    out[blockIdx.x * 64 + lane] = acc.x ^ acc.y ^ acc.z ^ acc.w;
}

static int time_gather(const void* array, std::uint64_t rows, const char* label, std::uint32_t* d_out) {
    const std::uint32_t waves = 4096, per_wave = 2000;
    hipEvent_t begin, end;
    CHECK(hipEventCreate(&begin));
    CHECK(hipEventCreate(&end));
    float best = 1e30f, worst = 0.f;
    for (int launch = 0; launch < 6; ++launch) {
        CHECK(hipEventRecord(begin, nullptr));
        hipLaunchKernelGGL(gather_kernel, dim3(waves), dim3(64), 0, nullptr, static_cast<const uint4*>(array), rows, per_wave, d_out);
        CHECK(hipEventRecord(end, nullptr));
        CHECK(hipEventSynchronize(end));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, begin, end));
        if (launch) // the first one warms the translation caches
            best = ms < best ? ms : best, worst = ms > worst ? ms : worst;
    }
    const double bytes = (double)waves * per_wave * 1536.0;
    std::printf("%-44s at %p: %7.3f ms best (%7.3f worst) = %6.2f TB/s\n", label, array, best, worst, bytes / best / 1e9);
    std::fflush(stdout);
    return 0;
}

int main(int argc, char** argv) {
    const std::uint64_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000ull;
    const std::size_t bytes = rows * 1536;
    std::uint32_t* d_out = nullptr;
    CHECK(hipMalloc((void**)&d_out, 4096 * 64 * 4));
    for (int round = 0; round < 2; ++round) {
        std::printf("--- round %d\n", round + 1);
        void* plain = nullptr;
        CHECK(hipMalloc(&plain, bytes));
        if (time_gather(plain, rows, "hipMalloc", d_out))
            return 1;
        void* contiguous = nullptr;
        if (hipExtMallocWithFlags(&contiguous, bytes, hipDeviceMallocContiguous) == hipSuccess) {
            if (time_gather(contiguous, rows, "hipExtMallocWithFlags(contiguous)", d_out))
                return 1;
        } else {
            (void)hipGetLastError();
            std::printf("contiguous allocation refused\n");
        }
        // virtual-memory API: the address is ours to align
        for (std::size_t alignment : {std::size_t(1) << 30, std::size_t(1) << 34}) {
            hipMemAllocationProp prop = {};
            prop.type = hipMemAllocationTypePinned;
            prop.location.type = hipMemLocationTypeDevice;
            prop.location.id = 0;
            std::size_t granularity = 0;
            CHECK(hipMemGetAllocationGranularity(&granularity, &prop, hipMemAllocationGranularityRecommended));
            const std::size_t rounded = (bytes + granularity - 1) / granularity * granularity;
            // the runtime ignores the alignment argument (first version of this probe): reserve `alignment` more and align inside
            void* reserved = nullptr;
            if (hipMemAddressReserve(&reserved, rounded + alignment, 0, nullptr, 0) != hipSuccess) {
                (void)hipGetLastError();
                std::printf("address reservation refused\n");
                continue;
            }
            void* address = reinterpret_cast<void*>((reinterpret_cast<std::uintptr_t>(reserved) + alignment - 1) / alignment * alignment);
            hipMemGenericAllocationHandle_t handle;
            CHECK(hipMemCreate(&handle, rounded, &prop, 0));
            CHECK(hipMemMap(address, rounded, 0, handle, 0));
            hipMemAccessDesc access = {};
            access.location = prop.location;
            access.flags = hipMemAccessFlagsProtReadWrite;
            CHECK(hipMemSetAccess(address, rounded, &access, 1));
            char label[96];
            std::snprintf(label, sizeof(label), "hipMemCreate + map, address aligned to 2^%d", alignment == (std::size_t(1) << 30) ? 30 : 34);
            if (time_gather(address, rows, label, d_out))
                return 1;
            CHECK(hipMemUnmap(address, rounded));
            CHECK(hipMemRelease(handle));
            CHECK(hipMemAddressFree(reserved, rounded + alignment));
        }
        // keep this round's plain array, free the contiguous one: the next round lands elsewhere
        if (contiguous)
            CHECK(hipFree(contiguous));
        if (round == 1)
            CHECK(hipFree(plain));
    }
    // a small array for comparison: translation reach is no issue here
    void* small = nullptr;
    CHECK(hipMalloc(&small, (std::size_t)1000000 * 1536));
    if (time_gather(small, 1000000, "hipMalloc, 1.5 GB", d_out))
        return 1;
    return 0;
}
