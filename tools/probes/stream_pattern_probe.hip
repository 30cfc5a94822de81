// stream_pattern_probe.hip — does it matter WHERE the concurrently running waves read? Every wave streams T 1 KiB chunks through a four-deep
// LDS-DMA ring (the CIGAR scan's loop with the work taken out); the chunk a wave reads at step c is
//   comb:   g * T + c                    each wave its own contiguous share, shares far apart (the scan's split today)
//   block:  ((c / B) * W + g) * B + c % B    waves take turns in blocks of B chunks: all waves inside a window of W * B chunks
//   weave:  c * W + g                    fully interleaved
// build: hipcc --offload-arch=gfx950 -O3 -o stream_pattern_probe stream_pattern_probe.hip ; run: ./stream_pattern_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>

__device__ __forceinline__ void glds16(const uint32_t *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }

template <int WORK>
__global__ __launch_bounds__(256) void ring_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t T, uint32_t B, uint32_t W)
{
    constexpr int D = 4;
    __shared__ uint32_t ring[4][D][256];
    __shared__ uint32_t pad[2048];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) pad[0] = 0;
    const uint32_t g = blockIdx.x * 4 + wave;
    auto at = [&](uint32_t c) -> const uint32_t * {
        const size_t chunk = B == 0 ? (size_t)g * T + c : ((size_t)(c / B) * W + g) * B + c % B;
        return in + chunk * 256 + lane * 4;
    };
    uint32_t issued = 0, acc = 0;
    for (; issued < D - 1 && issued < T; issued++) glds16(at(issued), __builtin_amdgcn_readfirstlane(lds_addr(&ring[wave][issued % D][0])));
    for (uint32_t c = 0; c < T; c++) {
        if (issued < T) { glds16(at(issued), __builtin_amdgcn_readfirstlane(lds_addr(&ring[wave][issued % D][0]))); issued++; }
        if (c + D - 1 < T) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 v = *reinterpret_cast<const uint4 *>(&ring[wave][c % D][lane * 4]);
        uint32_t x = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int i = 0; i < WORK; i++) x = x * 0x9E3779B1u + (x >> 7);
        acc += x;
    }
    if (acc == 0x12345678u) pad[lane] = acc;
    out[blockIdx.x * 256 + threadIdx.x] = acc + pad[0];
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
    const unsigned grid = 256 * 6 * 2;
    const uint32_t W = grid * 4;
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    for (uint32_t T : {48u, 224u}) {
        const size_t words = (size_t)W * T * 256;
        uint32_t *in, *out;
        CK(hipMalloc(&in, words * 4)); CK(hipMalloc(&out, (size_t)grid * 256 * 4));
        CK(hipMemset(in, 1, words * 4));
        auto run = [&](const char *name, uint32_t B, int work) -> int {
            auto launch = [&] {
                if (work == 0) hipLaunchKernelGGL((ring_kernel<0>), dim3(grid), dim3(256), 0, 0, in, out, T, B, W);
                else hipLaunchKernelGGL((ring_kernel<40>), dim3(grid), dim3(256), 0, 0, in, out, T, B, W);
            };
            for (int i = 0; i < 3; i++) launch();
            CK(hipEventRecord(a));
            for (int i = 0; i < 10; i++) launch();
            CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
            float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 10;
            printf("T=%3u (%5.2f GB) %-10s B=%3u work=%2d  %.4f ms  %.0f GB/s\n", T, words * 4 / 1e9, name, B, work, ms, words * 4 / ms / 1e6);
            return 0;
        };
        for (int work : {0, 40}) {
            run("comb", 0, work);
            run("block", 16, work);
            run("block", 4, work);
            run("weave", 1, work);
        }
        CK(hipFree(in)); CK(hipFree(out));
    }
    return 0;
}
