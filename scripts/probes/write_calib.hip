// scripts/probes/write_calib.hip — known write patterns for calibrating rocprofv3's WRITE_SIZE / FETCH_SIZE on gfx950
// (the microarchitecture guide calibrates FETCH_SIZE on wide streaming reads only; the search kernels' write traffic is
// slab clears and scattered 4-byte compare-and-swaps). Each kernel's byte count is printed; run under
//     rocprofv3 --pmc WRITE_SIZE -- ./write_calib      and      rocprofv3 --pmc FETCH_SIZE -- ./write_calib
// and compare the per-dispatch counter with the printed bytes.
//     hipcc --offload-arch=gfx950 -O3 -o write_calib write_calib.hip
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                                                       \
    do {                                                                                                               \
        hipError_t e = (x);                                                                                            \
        if (e != hipSuccess) {                                                                                         \
            std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e));                                                \
            std::exit(1);                                                                                              \
        }                                                                                                              \
    } while (0)

/// 16 bytes per lane, fully coalesced: `bytes` written exactly once.
__global__ void calib_stream_write(uint4* out, std::uint64_t cells) {
    const std::uint64_t stride = (std::uint64_t)gridDim.x * blockDim.x;
    for (std::uint64_t i = blockIdx.x * (std::uint64_t)blockDim.x + threadIdx.x; i < cells; i += stride)
        out[i] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
}

/// The search kernel's slab clear: one wave per 32-KB slab, 64 lanes x 16 bytes per store instruction.
__global__ void calib_slab_clear(uint4* slabs, std::uint32_t cells_per_slab) {
    uint4* mine = slabs + (std::uint64_t)blockIdx.x * cells_per_slab;
    for (std::uint32_t i = threadIdx.x; i < cells_per_slab; i += 64)
        mine[i] = uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu};
}

/// Scattered 4-byte compare-and-swaps into a wave-private 32-KB slab: `per_wave` rounds of 32 lanes each, the visited-set
/// insert pattern (every successful swap dirties one 128-byte line of the slab).
__global__ void calib_slab_cas(std::uint32_t* slabs, std::uint32_t cells_per_slab, std::uint32_t rounds) {
    std::uint32_t* mine = slabs + (std::uint64_t)blockIdx.x * cells_per_slab;
    std::uint32_t state = blockIdx.x * 2654435761u + threadIdx.x * 40503u + 1u;
    for (std::uint32_t r = 0; r < rounds; ++r) {
        state = state * 1664525u + 1013904223u;
        if (threadIdx.x < 32) {
            const std::uint32_t cell = (state >> 8) & (cells_per_slab - 1);
            atomicCAS(mine + cell, 0xFFFFFFFFu, state | 1u);
        }
    }
}

/// Scattered 16-byte reads (one per lane) from a big array: the inline-row / short-row read pattern.
__global__ void calib_scatter_read(const uint4* rows, std::uint64_t row_count, std::uint32_t rounds, std::uint32_t* sink) {
    std::uint32_t state = (blockIdx.x * 64u + threadIdx.x) * 2654435761u + 12345u;
    std::uint32_t acc = 0;
    for (std::uint32_t r = 0; r < rounds; ++r) {
        state = state * 1664525u + 1013904223u;
        const std::uint64_t row = ((std::uint64_t)state * row_count) >> 32;
        const uint4 v = rows[row];
        acc += v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u)
        *sink = acc;
}

int main() {
    const std::uint32_t waves = 6144, cells_per_slab = 8192; // the C5 launch: 6144 waves x 32 KB of visited-set slab
    const std::uint64_t slab_bytes = (std::uint64_t)waves * cells_per_slab * 4;
    std::uint32_t* slabs = nullptr;
    CHECK(hipMalloc((void**)&slabs, slab_bytes));
    const std::uint64_t stream_bytes = 2ull << 30;
    uint4* big = nullptr;
    CHECK(hipMalloc((void**)&big, stream_bytes));
    std::uint32_t* sink = nullptr;
    CHECK(hipMalloc((void**)&sink, 4));

    for (int repeat = 0; repeat < 2; ++repeat) {
        hipLaunchKernelGGL(calib_stream_write, dim3(256 * 8), dim3(256), 0, 0, big, stream_bytes / 16);
        CHECK(hipDeviceSynchronize());
        std::printf("calib_stream_write: %llu bytes written (coalesced 16 B per lane)\n", (unsigned long long)stream_bytes);
        hipLaunchKernelGGL(calib_slab_clear, dim3(waves), dim3(64), 0, 0, reinterpret_cast<uint4*>(slabs), cells_per_slab / 4);
        CHECK(hipDeviceSynchronize());
        std::printf("calib_slab_clear: %llu bytes written (%u waves x %u B)\n", (unsigned long long)slab_bytes, waves,
                    cells_per_slab * 4);
        const std::uint32_t rounds = 80; // ~ one query's hops
        hipLaunchKernelGGL(calib_slab_cas, dim3(waves), dim3(64), 0, 0, slabs, cells_per_slab, rounds);
        CHECK(hipDeviceSynchronize());
        std::printf("calib_slab_cas: %llu compare-and-swaps of 4 B (%u waves x %u rounds x 32 lanes); every slab line "
                    "(256 x 128 B per slab) is dirtied: %llu bytes if each line is written back once\n",
                    (unsigned long long)waves * rounds * 32, waves, rounds, (unsigned long long)slab_bytes);
        hipLaunchKernelGGL(calib_scatter_read, dim3(waves), dim3(64), 0, 0, big, stream_bytes / 16, 256u, sink);
        CHECK(hipDeviceSynchronize());
        std::printf("calib_scatter_read: %llu reads of 16 B = %llu bytes asked, %llu bytes in 64-B sectors, %llu in 128-B lines\n",
                    (unsigned long long)waves * 64 * 256, (unsigned long long)waves * 64 * 256 * 16,
                    (unsigned long long)waves * 64 * 256 * 64, (unsigned long long)waves * 64 * 256 * 128);
    }
    return 0;
}
