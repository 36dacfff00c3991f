// scripts/probes/atomic_rate_probe.hip — what the memory side turns over: compare-and-swaps on per-wave slabs of 32 KB (the short-row
// walks' visited sets: 8 192 cells a wave, 24 waves a compute unit), every lane a probe per round as in a hop.
//   independent: a lane's next address does not wait for the last answer (the ceiling of the memory side)
//   dependent:   it does (a hop's round trip: what 24 waves per CU hide)
// hipcc --offload-arch=gfx950 -O3 scripts/probes/atomic_rate_probe.hip -o scripts/probes/_bin/atomic_rate_probe && ./atomic_rate_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

__global__ __launch_bounds__(64) void probe(std::uint32_t* slabs, std::uint32_t cells, std::uint32_t rounds, int dependent, std::uint32_t* sink) {
    std::uint32_t* slab = slabs + (std::uint64_t)blockIdx.x * cells;
    std::uint32_t state = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 1u, folded = 0;
    for (std::uint32_t r = 0; r < rounds; ++r) {
        state = state * 1664525u + 1013904223u;
        const std::uint32_t h = ((state >> 9) ^ (dependent ? folded & 1u : 0u)) & (cells - 1);
        const std::uint32_t old = atomicCAS(slab + h, 0xFFFFFFFFu, state | 1u);
        if (dependent)
            folded = old;
        else
            folded ^= old;
    }
    if (folded == 0x12345678u)
        sink[0] = folded;
}

int main() {
    const std::uint32_t cells = 8192, rounds = 4000;
    for (std::uint32_t waves_per_cu : {8u, 16u, 24u, 32u}) {
        const std::uint32_t waves = 256 * waves_per_cu;
        std::uint32_t *slabs = nullptr, *sink = nullptr;
        hipMalloc(&slabs, (size_t)waves * cells * 4);
        hipMalloc(&sink, 4);
        for (int dependent = 0; dependent < 2; ++dependent) {
            hipMemset(slabs, 0xFF, (size_t)waves * cells * 4);
            hipEvent_t a, b;
            hipEventCreate(&a), hipEventCreate(&b);
            hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, slabs, cells, 200u, dependent, sink); // warm
            hipEventRecord(a);
            hipLaunchKernelGGL(probe, dim3(waves), dim3(64), 0, 0, slabs, cells, rounds, dependent, sink);
            hipEventRecord(b);
            hipEventSynchronize(b);
            float ms = 0;
            hipEventElapsedTime(&ms, a, b);
            const double atomics = (double)waves * 64 * rounds;
            std::printf("%u waves per CU (%u waves, slabs %.0f MB), %s: %.2f ms, %.1f G compare-and-swaps per second, %.2f us per round of a wave\n",
                        waves_per_cu, waves, waves * cells * 4 / 1e6, dependent ? "dependent rounds" : "independent rounds", ms,
                        atomics / ms / 1e6, ms * 1e3 / rounds);
        }
        hipFree(slabs), hipFree(sink);
    }
    return 0;
}
