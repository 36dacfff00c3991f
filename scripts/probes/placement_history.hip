// scripts/probes/placement_history.hip — is it the HISTORY of the memory that decides how fast a 15-GB array gathers?
// (development tool for the open question of DESIGN.md §3.1 item 3.) Three phases in one process, the same dependency-free gather
// of random 1536-byte rows on every array:
//   A. eight arrays allocated one after the other and all held: untouched memory, eight different physical ranges;
//   B. all freed, eight allocated again: memory that was just given back;
//   C. all freed, the free memory chopped up (many 3-MB blocks allocated, every other one freed), eight arrays again.
// If A is uniformly fast and B / C are not, fragments of recycled memory are the mechanism and the engine should take its arrays
// before anything is freed; if A already shows the spread, it is the physical range itself.
//   hipcc --offload-arch=gfx950 -O3 scripts/probes/placement_history.hip -o scripts/probes/placement_history.bin
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

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

static std::uint32_t* d_out = nullptr;

static int time_gather(const void* array, std::uint64_t rows, double* rate) {
    const std::uint32_t waves = 4096, per_wave = 2000;
    hipEvent_t begin, end;
    CHECK(hipEventCreate(&begin));
    CHECK(hipEventCreate(&end));
    float best = 1e30f;
    for (int launch = 0; launch < 5; ++launch) {
        CHECK(hipEventRecord(begin, nullptr));
        hipLaunchKernelGGL(gather_kernel, dim3(waves), dim3(64), 0, nullptr, static_cast<const uint4*>(array), rows, per_wave, d_out);
        CHECK(hipEventRecord(end, nullptr));
        CHECK(hipEventSynchronize(end));
        float ms = 0.f;
        CHECK(hipEventElapsedTime(&ms, begin, end));
        if (launch)
            best = ms < best ? ms : best;
    }
    *rate = (double)waves * per_wave * 1536.0 / best / 1e9;
    (void)hipEventDestroy(begin);
    (void)hipEventDestroy(end);
    return 0;
}

static int phase(const char* label, std::uint64_t rows, int arrays, std::vector<void*>& held) {
    std::size_t free_bytes = 0, total_bytes = 0;
    CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
    std::printf("--- %s (free %.1f of %.1f GB)\n", label, free_bytes / 1e9, total_bytes / 1e9);
    for (int i = 0; i < arrays; ++i) {
        void* array = nullptr;
        if (hipMalloc(&array, rows * 1536) != hipSuccess) {
            (void)hipGetLastError();
            std::printf("array %d: out of memory\n", i);
            break;
        }
        held.push_back(array);
        double rate = 0;
        if (time_gather(array, rows, &rate))
            return 1;
        std::printf("array %d at %p: %6.2f TB/s\n", i, array, rate);
        std::fflush(stdout);
    }
    return 0;
}

static int release(std::vector<void*>& held) {
    for (void* p : held)
        CHECK(hipFree(p));
    held.clear();
    return 0;
}

int main(int argc, char** argv) {
    const std::uint64_t rows = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 10000000ull;
    const int arrays = argc > 2 ? std::atoi(argv[2]) : 8;
    CHECK(hipMalloc((void**)&d_out, 4096 * 64 * 4));
    std::vector<void*> held;
    if (phase("A: untouched memory, all held", rows, arrays, held) || release(held))
        return 1;
    if (phase("B: the same memory, just given back", rows, arrays, held) || release(held))
        return 1;
    if (argc > 3) // "quick": phases A and B only
        return 0;
    // chop the free memory up: 3-MB blocks over ~60 % of it, every other one freed again
    std::size_t free_bytes = 0, total_bytes = 0;
    CHECK(hipMemGetInfo(&free_bytes, &total_bytes));
    std::vector<void*> crumbs;
    const std::size_t crumb = 3u << 20, wanted = (std::size_t)(free_bytes * 0.6 / crumb);
    for (std::size_t i = 0; i < wanted && i < 60000; ++i) {
        void* p = nullptr;
        if (hipMalloc(&p, crumb) != hipSuccess) {
            (void)hipGetLastError();
            break;
        }
        crumbs.push_back(p);
    }
    for (std::size_t i = 0; i < crumbs.size(); i += 2)
        CHECK(hipFree(crumbs[i]));
    std::printf("chopped: %zu blocks of 3 MB allocated, every other one freed\n", crumbs.size());
    if (phase("C: chopped-up memory", rows, arrays / 2, held) || release(held))
        return 1;
    for (std::size_t i = 1; i < crumbs.size(); i += 2)
        CHECK(hipFree(crumbs[i]));
    if (phase("D: after everything was given back", rows, arrays / 2, held) || release(held))
        return 1;
    return 0;
}
