// membench.hip — which kernel STRUCTURE reaches the in-place read+write ceiling on a 16 GiB fp64-complex state?
// Measurement aid for DESIGN.md §4 (not part of the library).  Every variant does v[i] *= c in place over 2^n amplitudes.
//   hipcc -O3 --offload-arch=gfx950 tools/membench.hip -o gpurun_out/membench && gpurun_out/membench 30
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <initializer_list>
#include <type_traits>
#include <vector>

typedef double amp_t __attribute__((ext_vector_type(2)));
typedef __attribute__((address_space(3))) amp_t lds_amp_t;

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__device__ __forceinline__ amp_t scale(amp_t a, double c) { return amp_t{a.x * c, a.y * c}; }

// one amplitude per thread, huge grid
__global__ __launch_bounds__(256) void k_simple(amp_t *v, double c) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    v[i] = scale(v[i], c);
}

// D loads in flight per thread, then D stores; each workgroup owns contiguous D*T amplitudes per step; grid-stride
template <int D, int T, bool NT>
__global__ __launch_bounds__(T) void k_deep(amp_t *v, uint64_t nchunks, double c) {
    for (uint64_t ch = blockIdx.x; ch < nchunks; ch += gridDim.x) {
        amp_t *p = v + ch * (uint64_t)(D * T) + threadIdx.x;
        amp_t a[D];
#pragma unroll
        for (int k = 0; k < D; k++) a[k] = NT ? __builtin_nontemporal_load(p + k * T) : p[k * T];
#pragma unroll
        for (int k = 0; k < D; k++) {
            if (NT) __builtin_nontemporal_store(scale(a[k], c), p + k * T);
            else p[k * T] = scale(a[k], c);
        }
    }
}

// tile through LDS with registers both ways, next tile prefetched into registers before the current one is stored
template <int D, int T>
__global__ __launch_bounds__(T) void k_tile_reg(amp_t *v, uint64_t ntiles, int tpw, double c) {
    extern __shared__ amp_t lds[];
    const uint64_t first = (uint64_t)blockIdx.x * tpw;
    uint64_t last = first + tpw;
    if (last > ntiles) last = ntiles;
    if (first >= last) return;
    amp_t a[D];
    {
        amp_t *p = v + first * (uint64_t)(D * T) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; k++) a[k] = p[k * T];
    }
    for (uint64_t t = first; t < last; t++) {
#pragma unroll
        for (int k = 0; k < D; k++) lds[threadIdx.x + k * T] = a[k];
        __syncthreads();
        if (t + 1 < last) {
            amp_t *p = v + (t + 1) * (uint64_t)(D * T) + threadIdx.x;
#pragma unroll
            for (int k = 0; k < D; k++) a[k] = p[k * T];
        }
        amp_t *q = v + t * (uint64_t)(D * T) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; k++) q[k * T] = scale(lds[(threadIdx.x + k * T) ^ 1], c); // ^1: a real LDS exchange
        __syncthreads();
    }
}

__device__ __forceinline__ void glds16(const amp_t *g, uint32_t lds_byte_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g,
                                     (__attribute__((address_space(3))) void *)(uintptr_t)lds_byte_wave_base, 16, 0, 0);
}

// tile filled by LDS-DMA (no VGPR staging), single buffer: fill -> wait -> barrier -> read+store -> barrier
template <int D, int T>
__global__ __launch_bounds__(T) void k_tile_glds(amp_t *v, uint64_t ntiles, int tpw, double c) {
    extern __shared__ amp_t lds[];
    const uint64_t first = (uint64_t)blockIdx.x * tpw;
    uint64_t last = first + tpw;
    if (last > ntiles) last = ntiles;
    const uint32_t wave = threadIdx.x >> 6;
    for (uint64_t t = first; t < last; t++) {
        const amp_t *p = v + t * (uint64_t)(D * T) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; k++) glds16(p + k * T, (uint32_t)((wave * 64 + k * T) * 16));
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        amp_t *q = v + t * (uint64_t)(D * T) + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; k++) q[k * T] = scale(lds[(threadIdx.x + k * T) ^ 1], c);
        __syncthreads();
    }
}

// LDS-DMA with two buffers in one workgroup: the fill of tile t+1 is in flight while tile t is read and stored
template <int D, int T>
__global__ __launch_bounds__(T) void k_tile_glds2(amp_t *v, uint64_t ntiles, int tpw, double c) {
    extern __shared__ amp_t lds[];
    const uint64_t first = (uint64_t)blockIdx.x * tpw;
    uint64_t last = first + tpw;
    if (last > ntiles) last = ntiles;
    if (first >= last) return;
    const uint32_t wave = threadIdx.x >> 6;
    constexpr uint32_t E = D * T;
    {
        const amp_t *p = v + first * (uint64_t)E + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; k++) glds16(p + k * T, (uint32_t)((wave * 64 + k * T) * 16));
    }
    uint32_t buf = 0;
    for (uint64_t t = first; t < last; t++, buf ^= 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // fill of tile t landed; stores of tile t-1 retired
        __syncthreads();                                   // ... for every wave; buffer buf^1 has been read out
        if (t + 1 < last) {
            const amp_t *p = v + (t + 1) * (uint64_t)E + threadIdx.x;
#pragma unroll
            for (int k = 0; k < D; k++) glds16(p + k * T, (uint32_t)(((buf ^ 1) * E + wave * 64 + k * T) * 16));
        }
        amp_t *q = v + t * (uint64_t)E + threadIdx.x;
#pragma unroll
        for (int k = 0; k < D; k++) q[k * T] = scale(lds[buf * E + ((threadIdx.x + k * T) ^ 1)], c);
    }
}


// ---- the tile access pattern of k_tile: 512 runs of 8 amplitudes per tile, run positions = index bits HIGH (constant
// masks so the bit deposits fold to a few shifts), consecutive tiles differ in the remaining (outer) bits -------------
template <uint64_t MASK>
__device__ __forceinline__ uint64_t dep(uint64_t x) {
    uint64_t out = 0, m = MASK;
#pragma unroll
    for (int i = 0; i < 64; i++) {
        if (!m) break;
        const uint64_t low = m & (0 - m);
        if (x & 1ULL) out |= low;
        x >>= 1;
        m &= m - 1;
    }
    return out;
}

template <uint64_t HIGH, int NQ>
__device__ __forceinline__ uint64_t tile_addr(uint64_t tile, uint32_t slot) {
    constexpr uint64_t ALL = (1ULL << NQ) - 1ULL;
    constexpr uint64_t OUTER = ALL & ~(HIGH | 7ULL);
    return dep<OUTER>(tile) | dep<HIGH>(slot >> 3) | (slot & 7u);
}

// one amplitude per thread, tile pattern, huge grid
template <uint64_t HIGH, int NQ>
__global__ __launch_bounds__(256) void k_simple_tile(amp_t *v, double c) {
    const uint64_t t = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    const uint64_t a = tile_addr<HIGH, NQ>(t >> 12, (uint32_t)(t & 4095));
    v[a] = scale(v[a], c);
}

// 8 loads then 8 stores per thread, a 512-thread workgroup owns a tile, TPW consecutive tiles per workgroup, no LDS
template <uint64_t HIGH, int NQ>
__global__ __launch_bounds__(512) void k_deep_tile(amp_t *v, uint64_t ntiles, int tpw, double c) {
    for (int j = 0; j < tpw; j++) {
        const uint64_t tile = (uint64_t)blockIdx.x * tpw + j;
        if (tile >= ntiles) return;
        amp_t a[8];
        uint64_t ad[8];
#pragma unroll
        for (int k = 0; k < 8; k++) { ad[k] = tile_addr<HIGH, NQ>(tile, threadIdx.x + k * 512); a[k] = v[ad[k]]; }
#pragma unroll
        for (int k = 0; k < 8; k++) v[ad[k]] = scale(a[k], c);
    }
}

// the same through LDS with the next tile prefetched into registers (the structure of k_tile with its blocks skipped)
template <uint64_t HIGH, int NQ>
__global__ __launch_bounds__(512) void k_lds_tile(amp_t *v, uint64_t ntiles, int tpw, double c) {
    extern __shared__ amp_t lds[];
    const uint64_t first = (uint64_t)blockIdx.x * tpw;
    uint64_t last = first + tpw;
    if (last > ntiles) last = ntiles;
    if (first >= last) return;
    amp_t a[8];
#pragma unroll
    for (int k = 0; k < 8; k++) a[k] = v[tile_addr<HIGH, NQ>(first, threadIdx.x + k * 512)];
    for (uint64_t t = first; t < last; t++) {
#pragma unroll
        for (int k = 0; k < 8; k++) lds[threadIdx.x + k * 512] = a[k];
        __syncthreads();
        if (t + 1 < last) {
#pragma unroll
            for (int k = 0; k < 8; k++) a[k] = v[tile_addr<HIGH, NQ>(t + 1, threadIdx.x + k * 512)];
        }
#pragma unroll
        for (int k = 0; k < 8; k++) v[tile_addr<HIGH, NQ>(t, threadIdx.x + k * 512)] = scale(lds[(threadIdx.x + k * 512) ^ 1], c);
        __syncthreads();
    }
}

constexpr uint64_t bits(std::initializer_list<int> l) { uint64_t m = 0; for (int b : l) m |= 1ULL << b; return m; }

template <typename F>
static double time_ms(F launch, int reps) {
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int r = 0; r < reps; r++) launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, a, b));
    CK(hipGetLastError());
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 30;
    const uint64_t N = 1ULL << n;
    amp_t *v;
    CK(hipMalloc(&v, N * 16));
    CK(hipMemset(v, 0, N * 16));
    const double bytes = 32.0 * (double)N;
    const double c = 1.0000001;
    auto report = [&](const char *name, double ms) { printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / ms * 1e-6); fflush(stdout); };
    const int reps = 10;

    report("simple 1/thread", time_ms([&] { hipLaunchKernelGGL(k_simple, dim3((unsigned)(N / 256)), dim3(256), 0, 0, v, c); }, reps));
    for (unsigned grid : {8192u, 1u << 19}) {
        char nm[96];
        snprintf(nm, sizeof nm, "deep D=8 T=256 grid=%u", grid);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_deep<8, 256, false>), dim3(grid), dim3(256), 0, 0, v, N / 2048, c); }, reps));
        snprintf(nm, sizeof nm, "deep D=4 T=256 grid=%u", grid);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_deep<4, 256, false>), dim3(grid), dim3(256), 0, 0, v, N / 1024, c); }, reps));
        snprintf(nm, sizeof nm, "deep D=8 T=512 grid=%u", grid);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_deep<8, 512, false>), dim3(grid), dim3(512), 0, 0, v, N / 4096, c); }, reps));
        snprintf(nm, sizeof nm, "deep D=8 T=256 nontemporal grid=%u", grid);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_deep<8, 256, true>), dim3(grid), dim3(256), 0, 0, v, N / 2048, c); }, reps));
    }
    const uint64_t ntiles = N / 4096;
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_glds2<4, 1024>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile_glds2<8, 512>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    for (int tpw : {1, 2, 4, 8}) {
        const unsigned grid = (unsigned)((ntiles + tpw - 1) / tpw);
        char nm[96];
        snprintf(nm, sizeof nm, "tile reg-staged 64K T=512 tpw=%d", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_reg<8, 512>), dim3(grid), dim3(512), 65536, 0, v, ntiles, tpw, c); }, reps));
        snprintf(nm, sizeof nm, "tile glds 64K T=512 tpw=%d (2 WG/CU)", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_glds<8, 512>), dim3(grid), dim3(512), 65536, 0, v, ntiles, tpw, c); }, reps));
        snprintf(nm, sizeof nm, "tile glds 64K T=256 tpw=%d (2 WG/CU)", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_glds<16, 256>), dim3(grid), dim3(256), 65536, 0, v, ntiles, tpw, c); }, reps));
        snprintf(nm, sizeof nm, "tile glds 2x64K T=1024 tpw=%d (1 WG/CU)", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_glds2<4, 1024>), dim3(grid), dim3(1024), 131072, 0, v, ntiles, tpw, c); }, reps));
        snprintf(nm, sizeof nm, "tile glds 2x64K T=512 tpw=%d (1 WG/CU)", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_glds2<8, 512>), dim3(grid), dim3(512), 131072, 0, v, ntiles, tpw, c); }, reps));
    }
    // 32 KiB tiles: four single-buffered workgroups per CU, or two double-buffered ones
    const uint64_t nt32 = N / 2048;
    for (int tpw : {1, 2, 8}) {
        const unsigned grid = (unsigned)((nt32 + tpw - 1) / tpw);
        char nm[96];
        snprintf(nm, sizeof nm, "tile glds 32K T=256 tpw=%d (4 WG/CU)", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_glds<8, 256>), dim3(grid), dim3(256), 32768, 0, v, nt32, tpw, c); }, reps));
        snprintf(nm, sizeof nm, "tile glds 2x32K T=512 tpw=%d (2 WG/CU)", tpw);
        report(nm, time_ms([&] { hipLaunchKernelGGL((k_tile_glds2<4, 512>), dim3(grid), dim3(512), 65536, 0, v, nt32, tpw, c); }, reps));
    }

    if (n == 30) {
        auto pattern = [&](const char *name, auto tag) {
            constexpr uint64_t H = decltype(tag)::value;
            char nm[128];
            snprintf(nm, sizeof nm, "%s: simple 1/thread", name);
            report(nm, time_ms([&] { hipLaunchKernelGGL((k_simple_tile<H, 30>), dim3((unsigned)(N / 256)), dim3(256), 0, 0, v, c); }, reps));
            for (int tpw : {1, 8}) {
                const unsigned grid = (unsigned)((ntiles + tpw - 1) / tpw);
                snprintf(nm, sizeof nm, "%s: 8 loads/8 stores, no LDS, tpw=%d", name, tpw);
                report(nm, time_ms([&] { hipLaunchKernelGGL((k_deep_tile<H, 30>), dim3(grid), dim3(512), 0, 0, v, ntiles, tpw, c); }, reps));
                snprintf(nm, sizeof nm, "%s: through LDS + prefetch, tpw=%d", name, tpw);
                report(nm, time_ms([&] { hipLaunchKernelGGL((k_lds_tile<H, 30>), dim3(grid), dim3(512), 65536, 0, v, ntiles, tpw, c); }, reps));
            }
        };
        pattern("high=3..11 (contiguous)", std::integral_constant<uint64_t, bits({3, 4, 5, 6, 7, 8, 9, 10, 11})>{});
        pattern("high=bench pass 2", std::integral_constant<uint64_t, bits({5, 10, 12, 15, 17, 18, 19, 20, 29})>{});
        pattern("high=bench pass 12", std::integral_constant<uint64_t, bits({4, 8, 12, 13, 16, 17, 19, 22, 24})>{});
        pattern("high=21..29 (top)", std::integral_constant<uint64_t, bits({21, 22, 23, 24, 25, 26, 27, 28, 29})>{});
    }
    CK(hipFree(v));
    return 0;
}
