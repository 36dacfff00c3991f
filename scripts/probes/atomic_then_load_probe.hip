// scripts/probes/atomic_then_load_probe.hip — what a LOAD sees of a cell the memory side has just written by compare-and-swap: the
// precondition of probing the visited set with a plain load first and spending an atomic only to claim an empty cell.
// Every lane: (1) loads its cell (empty: the line is now wherever loads leave lines), (2) compare-and-swaps a value in, (3) loads the
// cell again with each flavour of load. Counts the lanes that still see "empty".
// hipcc --offload-arch=gfx950 -O3 scripts/probes/atomic_then_load_probe.hip -o scripts/probes/_bin/atomic_then_load_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>

__global__ __launch_bounds__(64) void probe(std::uint32_t* cells, std::uint32_t count, unsigned long long* stale, int touch_first) {
    const std::uint32_t i = (blockIdx.x * 64u + threadIdx.x) * 17u % count; // scattered: one cell per line mostly
    std::uint32_t before = 0;
    if (touch_first)
        before = __hip_atomic_load(cells + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT); // a plain load (may stay in L1 / L2)
    const std::uint32_t value = 0x40000000u | i;
    const std::uint32_t old = atomicCAS(cells + i, 0xFFFFFFFFu, value);
    __builtin_amdgcn_s_waitcnt(0);
    const std::uint32_t plain = __hip_atomic_load(cells + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    const std::uint32_t workgroup = __hip_atomic_load(cells + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    const std::uint32_t agent = __hip_atomic_load(cells + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const std::uint32_t system = __hip_atomic_load(cells + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    if (old == 0xFFFFFFFFu) { // this lane's swap went in: every later load of it should see `value`
        if (plain != value) atomicAdd(stale + 0, 1ull);
        if (workgroup != value) atomicAdd(stale + 1, 1ull);
        if (agent != value) atomicAdd(stale + 2, 1ull);
        if (system != value) atomicAdd(stale + 3, 1ull);
        atomicAdd(stale + 4, 1ull);
    }
    if (before == 0x12345u)
        stale[7] = before;
}

int main() {
    const std::uint32_t count = 1u << 24; // 64 MB of cells
    std::uint32_t* cells = nullptr;
    unsigned long long* stale = nullptr;
    hipMalloc(&cells, (size_t)count * 4);
    hipMalloc(&stale, 64);
    for (int touch_first = 0; touch_first < 2; ++touch_first) {
        hipMemset(cells, 0xFF, (size_t)count * 4);
        hipMemset(stale, 0, 64);
        hipLaunchKernelGGL(probe, dim3(8192), dim3(64), 0, 0, cells, count, stale, touch_first);
        unsigned long long host[8] = {0};
        hipMemcpy(host, stale, 64, hipMemcpyDeviceToHost);
        std::printf("%s: of %llu swaps that went in, loads right after still saw the cell empty: plain (wavefront scope) %llu, workgroup scope %llu, "
                    "agent scope %llu, system scope %llu\n",
                    touch_first ? "cell loaded once before the swap" : "cell never loaded before", host[4], host[0], host[1], host[2], host[3]);
    }
    return 0;
}
