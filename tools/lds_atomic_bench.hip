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

// --json: the peak that bench.py's roofline uses -- random ds_add_u32 (what accumulate_counts issues per update)
// over the geometries the pair kernel could run in; the best of them is the chip's rate for this instruction mix.
template <int CONFLICT_FREE>
__global__ __launch_bounds__(1024) void k32(uint32_t iters, uint32_t mask, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    uint32_t *t32 = reinterpret_cast<uint32_t *>(raw);
    for (uint32_t i = threadIdx.x; i <= mask; i += blockDim.x) t32[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
#pragma unroll 4
    for (uint32_t it = 0; it < iters; ++it) {
        x = x * 1664525u + 1013904223u;
        uint32_t idx = (x >> 8) & mask;
        if (CONFLICT_FREE) idx = (idx & ~31u) | (threadIdx.x & 31u);  // every lane its own bank
        atomicAdd(&t32[idx], (x & 1u) ? 0x10000u : 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = t32[1];
}

// The same instruction with NOTHING else in the loop (the generator above spends a quarter-rate multiply
// and half a dozen other vector instructions per atomic, so its rate might be the vector pipes' and not the LDS
// pipeline's): sixteen random addresses per lane, made before the loop and kept in registers, added to over and over
// -- every wave instruction still has 64 different random addresses, the bank pattern differs between the sixteen.
template <int CONFLICT_FREE>
__global__ __launch_bounds__(1024) void k32r(uint32_t iters, uint32_t mask, uint32_t *sink) {
    extern __shared__ __attribute__((aligned(16))) unsigned char raw[];
    uint32_t *t32 = reinterpret_cast<uint32_t *>(raw);
    for (uint32_t i = threadIdx.x; i <= mask; i += blockDim.x) t32[i] = 0;
    __syncthreads();
    uint32_t x = threadIdx.x * 2654435761u + blockIdx.x * 40503u + 12345u;
    uint32_t *p[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) {
        x = x * 1664525u + 1013904223u;
        uint32_t idx = (x >> 8) & mask;
        if (CONFLICT_FREE) idx = (idx & ~31u) | (threadIdx.x & 31u);
        p[j] = t32 + idx;
    }
    for (uint32_t it = 0; it < iters; it += 16) {
#pragma unroll
        for (int j = 0; j < 16; ++j) atomicAdd(p[j], 1u);
    }
    __syncthreads();
    if (threadIdx.x == 0) sink[blockIdx.x] = t32[1];
}

template <int CONFLICT_FREE, bool REGISTER_ADDRESSES = false>
double run32(int threads, uint32_t cells, int blocks) {
    uint32_t *sink;
    hipMalloc(&sink, blocks * 4);
    const uint32_t iters = 8192;
    const size_t lds = (size_t)cells * 4;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto kern = REGISTER_ADDRESSES ? &k32r<CONFLICT_FREE> : &k32<CONFLICT_FREE>;
    hipFuncSetAttribute(reinterpret_cast<const void *>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    float best = 1e30f;
    for (int rep = 0; rep < 4; ++rep) {
        hipEventRecord(a);
        hipLaunchKernelGGL(kern, dim3(blocks), dim3(threads), lds, 0, iters, cells - 1, sink);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        if (rep) best = ms < best ? ms : best;
    }
    hipFree(sink);
    return (double)blocks * threads * iters / (best * 1e-3);
}

int json_main() {
    hipDeviceProp_t prop;
    hipGetDeviceProperties(&prop, 0);
    struct Cfg { int threads; uint32_t cells; int per_cu; };
    const Cfg cfgs[] = {{256, 8192, 4}, {256, 8192, 8}, {512, 16384, 2}, {512, 16384, 4}, {1024, 16384, 1},
                        {1024, 16384, 2}, {1024, 32768, 1}};
    double peak = 0, peak_cf = 0, peak_reg = 0, peak_reg_cf = 0;
    printf("{\"device\": \"%s\", \"compute_units\": %d, \"clock_mhz\": %d, \"instruction\": \"ds_add_u32, 64 random addresses per "
           "wave instruction\", \"configs\": [", prop.gcnArchName, prop.multiProcessorCount, prop.clockRate / 1000);
    bool first = true;
    for (const Cfg &c : cfgs) {
        const int blocks = prop.multiProcessorCount * c.per_cu * 4;  // four rounds of resident workgroups
        const double r = run32<0>(c.threads, c.cells, blocks), f = run32<1>(c.threads, c.cells, blocks);
        const double rr = run32<0, true>(c.threads, c.cells, blocks), rf = run32<1, true>(c.threads, c.cells, blocks);
        peak = r > peak ? r : peak;
        peak_cf = f > peak_cf ? f : peak_cf;
        peak_reg = rr > peak_reg ? rr : peak_reg;
        peak_reg_cf = rf > peak_reg_cf ? rf : peak_reg_cf;
        printf("%s{\"threads\": %d, \"tile_bytes\": %u, \"workgroups_per_cu\": %d, \"random_gatomic_per_s\": %.1f, "
               "\"bank_conflict_free_gatomic_per_s\": %.1f, \"register_addresses_random_gatomic_per_s\": %.1f, "
               "\"register_addresses_bank_conflict_free_gatomic_per_s\": %.1f}", first ? "" : ", ", c.threads, c.cells * 4,
               c.per_cu, r * 1e-9, f * 1e-9, rr * 1e-9, rf * 1e-9);
        first = false;
    }
    // the roofline's peak: the faster of the two generators at random addresses (what the LDS pipeline gives when the
    // vector pipes ask nothing of the wave)
    printf("], \"peak_generated_addresses_gatomic_per_s\": %.1f, \"peak_register_addresses_gatomic_per_s\": %.1f, "
           "\"peak_register_addresses_bank_conflict_free_gatomic_per_s\": %.1f, "
           "\"peak_random_gatomic_per_s\": %.1f, \"peak_bank_conflict_free_gatomic_per_s\": %.1f}\n", peak * 1e-9,
           peak_reg * 1e-9, peak_reg_cf * 1e-9, (peak_reg > peak ? peak_reg : peak) * 1e-9,
           (peak_reg_cf > peak_cf ? peak_reg_cf : peak_cf) * 1e-9);
    return 0;
}

int main(int argc, char **argv) {
    if (argc > 1 && argv[1][0] == '-' && argv[1][1] == '-' && argv[1][2] == 'j') return json_main();
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
