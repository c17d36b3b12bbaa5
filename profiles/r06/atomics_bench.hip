// atomics_bench.hip -- what random read-modify-write traffic costs on MI355X, by scope, width, return use and working set.
// Decides the data layout of the IXF builder (builder.hip): peeling a 3-uniform hypergraph is ~6 random RMWs per key.
// build: hipcc --offload-arch=gfx950 -O3 profiles/r06/atomics_bench.hip -o /tmp/atomics_bench
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t h)
{
    h ^= h >> 33; h *= 0xff51afd7ed558ccdull; h ^= h >> 33; h *= 0xc4ceb9fe1a85ec53ull; h ^= h >> 33;
    return h;
}

// mode 0: agent-scope non-returning add; 1: agent returning; 2: workgroup-scope non-returning (region of this XCC only);
// 3: workgroup-scope returning (XCC region); 4: plain random load; 5: agent non-returning, XCC region
template <typename T, int MODE>
__global__ __launch_bounds__(256) void k_rmw(T *buf, uint64_t n_words, int iters, uint64_t *sink)
{
    uint32_t xcc;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
    xcc &= 7u;
    const uint64_t gid = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    uint64_t acc = 0;
    const bool regional = MODE == 2 || MODE == 3 || MODE == 5;
    const uint64_t span = regional ? n_words / 8 : n_words;
    T *base = regional ? buf + xcc * span : buf;
    for (int i = 0; i < iters; ++i) {
        const uint64_t r = mix(gid * 0x9E3779B97F4A7C15ull + i);
        const uint64_t a = (uint64_t)(((unsigned __int128)r * span) >> 64);
        if (MODE == 0 || MODE == 5) __hip_atomic_fetch_add(base + a, (T)257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 1) acc += __hip_atomic_fetch_add(base + a, (T)257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else if (MODE == 2) __hip_atomic_fetch_add(base + a, (T)257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else if (MODE == 3) acc += __hip_atomic_fetch_add(base + a, (T)257, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        else acc += __builtin_nontemporal_load(base + a);
    }
    if (acc == 0x1234567) sink[0] = acc;
}

template <typename T, int MODE>
static int run(const char *name, T *buf, uint64_t bytes, uint64_t *sink)
{
    const uint64_t n_words = bytes / sizeof(T);
    const int blocks = 256 * 8, iters = 256;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k_rmw<T, MODE>), dim3(blocks), dim3(256), 0, nullptr, buf, n_words, 16, sink);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    hipLaunchKernelGGL((k_rmw<T, MODE>), dim3(blocks), dim3(256), 0, nullptr, buf, n_words, iters, sink);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    const double ops = (double)blocks * 256 * iters;
    printf("%-34s %2zu B  set %8.1f MB : %8.2f G ops/s  (%.3f ms)\n", name, sizeof(T), bytes / 1048576.0, ops / ms / 1e6, ms);
    fflush(stdout);
    return 0;
}

int main()
{
    uint8_t *buf = nullptr;
    uint64_t *sink = nullptr;
    const uint64_t max_bytes = 8ull << 30;
    CK(hipMalloc((void **)&buf, max_bytes));
    CK(hipMalloc((void **)&sink, 64));
    CK(hipMemset(buf, 0, max_bytes));
    for (uint64_t mb : {1ull, 4ull, 16ull, 32ull, 128ull, 256ull, 1024ull, 8192ull}) {
        const uint64_t b = mb << 20;
        run<uint32_t, 0>("agent add, no return", (uint32_t *)buf, b, sink);
        run<uint32_t, 1>("agent add, returning", (uint32_t *)buf, b, sink);
        run<uint32_t, 5>("agent add, no return, XCC region", (uint32_t *)buf, b, sink);
        run<uint32_t, 2>("workgroup add, no return, XCC reg", (uint32_t *)buf, b, sink);
        run<uint32_t, 3>("workgroup add, returning, XCC reg", (uint32_t *)buf, b, sink);
        run<uint32_t, 4>("nt load", (uint32_t *)buf, b, sink);
        run<uint64_t, 0>("agent add, no return", (uint64_t *)buf, b, sink);
        run<uint64_t, 1>("agent add, returning", (uint64_t *)buf, b, sink);
        run<uint64_t, 2>("workgroup add, no return, XCC reg", (uint64_t *)buf, b, sink);
        run<uint64_t, 3>("workgroup add, returning, XCC reg", (uint64_t *)buf, b, sink);
        run<uint64_t, 4>("nt load", (uint64_t *)buf, b, sink);
        printf("\n");
    }
    return 0;
}
