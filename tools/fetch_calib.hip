// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for the access widths this
// repository's kernels use (MI355X_MICROARCH.md, HBM: "FETCH_SIZE reports exactly 1/2 of the bytes of a wide
// coalesced streaming read ... other access widths are uncalibrated: calibrate on a known byte count in your
// own access pattern"). Streams a 1 GiB buffer (well past the 256 MiB Infinity Cache) once per kernel:
//   read4   4 bytes per lane, coalesced (the entry32 stream of accumulate_tiles)
//   read16  16 bytes per lane
//   write16 16 bytes per lane (the slab flush)
// Run under `rocprofv3 --pmc FETCH_SIZE` and `--pmc WRITE_SIZE`; tools/refresh_profiles_r02.sh divides.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>

__global__ void read4(const uint32_t *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) acc ^= p[i];
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void read16(const uint4 *p, size_t n, uint32_t *sink) {
    uint32_t acc = 0;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = p[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    if (acc == 0x12345678u) *sink = acc;
}
__global__ void write16(uint4 *p, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        p[i] = make_uint4((uint32_t)i, 1u, 2u, 3u);
}

int main() {
    const size_t bytes = 1ull << 30;
    void *buf, *sink;
    if (hipMalloc(&buf, bytes) != hipSuccess || hipMalloc(&sink, 4) != hipSuccess) return 1;
    hipMemset(buf, 1, bytes);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(read4, dim3(4096), dim3(256), 0, 0, (const uint32_t *)buf, bytes / 4, (uint32_t *)sink);
        hipLaunchKernelGGL(read16, dim3(4096), dim3(256), 0, 0, (const uint4 *)buf, bytes / 16, (uint32_t *)sink);
        hipLaunchKernelGGL(write16, dim3(4096), dim3(256), 0, 0, (uint4 *)buf, bytes / 16);
    }
    if (hipDeviceSynchronize() != hipSuccess) return 1;
    std::printf("bytes_per_kernel %zu\n", bytes);
    return 0;
}
