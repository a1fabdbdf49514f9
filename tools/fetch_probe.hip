// Calibration of the FETCH_SIZE counter for the load shapes the image kernels use: every kernel below reads a
// buffer of KNOWN size exactly once and does nothing else (one xor-reduced dword per thread is written so that the
// loads cannot be dropped).  Run each pass under `rocprofv3 --pmc FETCH_SIZE -- tools/fetch_probe` and compare the
// counter (in KB) with the bytes printed here.
//   k_load16_aligned   16 bytes per lane, 16-byte aligned          (the BGRA staging of the level-0 tile kernel)
//   k_load16_align4    16 bytes per lane, 4-byte aligned (base + 4) (the gray staging of the upper levels, the resize)
//   k_load4            4 bytes per lane                             (the narrow-load kernels)
// usage: fetch_probe [MiB, default 512]
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));

__global__ __launch_bounds__(256) void k_load16_aligned(const uint4 *__restrict__ src, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const uint4 v = src[i];
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_load16_align4(const uint8_t *__restrict__ src, size_t n16, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) {
        const u32x4_a4 v = *reinterpret_cast<const u32x4_a4 *>(src + 4 + 16 * i);
        acc ^= v.x ^ v.y ^ v.z ^ v.w;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

__global__ __launch_bounds__(256) void k_load4(const uint32_t *__restrict__ src, size_t n4, uint32_t *out)
{
    uint32_t acc = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
        acc ^= src[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
}

#define CHECK(x)                                                                      \
    do {                                                                              \
        hipError_t e_ = (x);                                                          \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } \
    } while (0)

int main(int argc, char **argv)
{
    const size_t mib = argc > 1 ? (size_t)atoll(argv[1]) : 512;
    const size_t bytes = mib << 20;
    uint8_t *buf = nullptr;
    uint32_t *out = nullptr;
    const int blocks = 256 * 16, threads = 256;
    CHECK(hipMalloc(&buf, bytes + 64));
    CHECK(hipMalloc(&out, sizeof(uint32_t) * blocks * threads));
    CHECK(hipMemset(buf, 0x5a, bytes + 64));
    CHECK(hipDeviceSynchronize());
    // one launch each; the buffer (512 MiB) is larger than the 256 MiB last-level cache, so every launch reads HBM
    hipLaunchKernelGGL(k_load16_aligned, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<const uint4 *>(buf), bytes / 16, out);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_load16_align4, dim3(blocks), dim3(threads), 0, 0, buf, bytes / 16, out);
    CHECK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_load4, dim3(blocks), dim3(threads), 0, 0, reinterpret_cast<const uint32_t *>(buf), bytes / 4, out);
    CHECK(hipDeviceSynchronize());
    printf("bytes read by each kernel: %zu (%zu KB)\n", bytes, bytes / 1024);
    (void)hipFree(buf);
    (void)hipFree(out);
    return 0;
}
