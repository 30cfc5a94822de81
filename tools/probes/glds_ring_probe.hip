// glds_ring_probe.hip — does a per-wave LDS ring filled by LDS-DMA (global_load_lds_dwordx4, no VGPR destination) stream faster than
// a one-chunk-ahead register prefetch at the CIGAR scan's occupancy? Each wave walks T consecutive 1 KiB chunks, as the scan does.
// build: hipcc --offload-arch=gfx950 -O3 -o glds_ring_probe glds_ring_probe.hip ; run: ./glds_ring_probe
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <vector>

__device__ __forceinline__ void glds16(const uint32_t *gsrc, uint32_t lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ uint32_t lds_addr(const void *p) { return (uint32_t)(uintptr_t)(__attribute__((address_space(3))) const void *)p; }

template <int D, int WORK>
__global__ __launch_bounds__(256) void ring_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t T)
{
    __shared__ uint32_t ring[4][D][256];
    __shared__ uint32_t pad[2048];                      // the scan's signature buffer (8 KiB)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) pad[0] = 0;
    const uint32_t *src = in + ((size_t)blockIdx.x * 4 + wave) * T * 256 + lane * 4;
    uint32_t issued = 0, acc = 0;
    for (; issued < D - 1 && issued < T; issued++) glds16(src + (size_t)issued * 256, __builtin_amdgcn_readfirstlane(lds_addr(&ring[wave][issued % D][0])));
    for (uint32_t c = 0; c < T; c++) {
        if (issued < T) { glds16(src + (size_t)issued * 256, __builtin_amdgcn_readfirstlane(lds_addr(&ring[wave][issued % D][0]))); issued++; }
        if (c + D - 1 < T) {
            if (D == 1) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (D == 2) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            if (D == 3) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            if (D == 4) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            if (D == 6) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            if (D == 8) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        uint4 v = *reinterpret_cast<const uint4 *>(&ring[wave][c % D][lane * 4]);   // the asm waits above carry a "memory" clobber: re-read every time
        uint32_t x = v.x + v.y + v.z + v.w;
#pragma unroll
        for (int i = 0; i < WORK; i++) x = x * 0x9E3779B1u + (x >> 7);      // stand-in for the scan's per-chunk VALU work
        acc += x;
    }
    if (acc == 0x12345678u) pad[lane] = acc;
    out[blockIdx.x * 256 + threadIdx.x] = acc + pad[0];
}

template <int WORK>
__global__ __launch_bounds__(256) void reg_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t T)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t *src = in + ((size_t)blockIdx.x * 4 + wave) * T * 256 + lane * 4;
    uint32_t acc = 0;
    uint4 cur = *reinterpret_cast<const uint4 *>(src);
    for (uint32_t c = 0; c < T; c++) {
        uint4 nxt = cur;
        if (c + 1 < T) nxt = *reinterpret_cast<const uint4 *>(src + (size_t)(c + 1) * 256);
        uint32_t x = cur.x + cur.y + cur.z + cur.w;
#pragma unroll
        for (int i = 0; i < WORK; i++) x = x * 0x9E3779B1u + (x >> 7);
        acc += x;
        cur = nxt;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc;
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)

int main()
{
    const uint32_t T = 48;
    const unsigned grid = 256 * 6 * 2;
    const size_t words = (size_t)grid * 4 * T * 256;
    uint32_t *in, *out;
    CK(hipMalloc(&in, words * 4)); CK(hipMalloc(&out, (size_t)grid * 256 * 4));
    CK(hipMemset(in, 1, words * 4));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    std::vector<uint32_t> ref((size_t)grid * 256), got((size_t)grid * 256);
    auto run = [&](const char *name, auto launch, bool is_ref) -> int {
        for (int i = 0; i < 3; i++) launch();
        CK(hipEventRecord(a));
        for (int i = 0; i < 20; i++) launch();
        CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
        float ms; CK(hipEventElapsedTime(&ms, a, b)); ms /= 20;
        CK(hipMemcpy(is_ref ? ref.data() : got.data(), out, (size_t)grid * 256 * 4, hipMemcpyDeviceToHost));
        bool same = is_ref || got == ref;
        printf("%-28s %.4f ms  %.0f GB/s  %s\n", name, ms, words * 4 / ms / 1e6, same ? "ok" : "MISMATCH");
        return 0;
    };
#define RING(Dv, W) run("ring D=" #Dv " work=" #W, [&] { hipLaunchKernelGGL((ring_kernel<Dv, W>), dim3(grid), dim3(256), 0, 0, in, out, T); }, false)
#define REG(W, R) run("reg prefetch work=" #W, [&] { hipLaunchKernelGGL((reg_kernel<W>), dim3(grid), dim3(256), 0, 0, in, out, T); }, R)
    REG(0, true);  RING(1, 0); RING(2, 0); RING(3, 0); RING(4, 0); RING(6, 0); RING(8, 0);
    REG(40, true); RING(2, 40); RING(4, 40); RING(8, 40);
    REG(100, true); RING(2, 100); RING(4, 100); RING(8, 100);
    return 0;
}
