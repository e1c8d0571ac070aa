// Micro-benchmark: LDS atomic throughput on gfx950 for the accumulate kernel's design decisions.
// hipcc --offload-arch=gfx950 -O3 tools/lds_atomic_bench.hip -o /tmp/lds_atomic_bench && /tmp/lds_atomic_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

template <int MODE>
__global__ __launch_bounds__(256) void k(uint32_t iters, uint32_t mask, unsigned long long *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    unsigned long long *t64 = reinterpret_cast<unsigned long long *>(raw);
    uint32_t *t32 = reinterpret_cast<uint32_t *>(raw);
    double *tf = reinterpret_cast<double *>(raw);
    for (uint32_t i = threadIdx.x; i <= mask; i += 256) t64[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    for (uint32_t it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        const uint32_t idx = (x >> 8) & mask;
        if (MODE == 0) atomicAdd(&t32[idx], 1u);
        if (MODE == 1) atomicAdd(&t64[idx], 0x100000001ull);
        if (MODE == 2) atomicAdd(&tf[idx], 1.0);
        if (MODE == 3) { t32[idx] += 1u; }                       // plain RMW (wrong, for reference)
        if (MODE == 4) { atomicAdd(&t32[idx], 1u); atomicAdd(&t32[idx + mask + 1], 1u); }  // two u32 adds
        if (MODE == 5) atomicAdd(&t64[(x >> 8) & mask & ~63u | (threadIdx.x & 63u)], 1ull);  // conflict-free lanes
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = t64[1];
}

template <int MODE>
void run(const char *name, uint32_t cells, size_t lds, int blocks) {
    unsigned long long *sink;
    hipMalloc(&sink, blocks * 8);
    const uint32_t iters = 4096;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    hipFuncSetAttribute(reinterpret_cast<const void *>(&k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), lds, 0, iters, cells - 1, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
    }
    float ms; hipEventElapsedTime(&ms, a, b);
    const double ops = (double)blocks * 256 * iters * (MODE == 4 ? 2 : 1);
    printf("%-34s cells %6u lds %6zu blocks %5d: %8.3f ms  %8.2f Gatomic/s  (%.2f per clk per CU)\n", name, cells, lds,
           blocks, ms, ops / ms * 1e-6, ops / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(sink);
}

int main() {
    for (int blocks : {256 * 2, 256 * 4, 256 * 8}) {
        run<0>("ds_add_u32 random", 4096, 32768, blocks);
        run<1>("ds_add_u64 random", 4096, 32768, blocks);
        run<2>("ds_add_f64 random", 4096, 32768, blocks);
        run<3>("plain u32 rmw random", 4096, 32768, blocks);
        run<4>("2x ds_add_u32 random", 4096, 65536, blocks);
        run<5>("ds_add_u64 lane-aligned", 4096, 32768, blocks);
    }
    run<1>("ds_add_u64 random, 128KB tile", 16384, 131072, 256);
    run<0>("ds_add_u32 random, 64KB tile", 16384, 65536, 512);
    return 0;
}
