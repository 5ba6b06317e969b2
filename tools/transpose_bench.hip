// transpose_bench.hip — what would a tile pass cost on its memory side if it READ its tile from contiguous memory and only
// its WRITES were scattered (a pass that leaves the state in the layout the NEXT pass wants: "layout-evolving passes",
// DESIGN.md §4 / §9)?  Measurement aid, not part of the library.  Same structure as k_tile's memory side: 512 threads, 2^12
// amplitudes through 64 KiB of LDS, 8 tiles per workgroup, the next tile prefetched into registers, out of place.
//   index in  = low 3 bits | 9 tile bits spread over rd[]  | tile number deposited into the remaining bits
//   index out = low 3 bits | 9 tile bits spread over wr[]  | tile number deposited into the remaining bits
// (a bijection of the index space for any two 9-bit sets).
//   hipcc -O3 --offload-arch=gfx950 tools/transpose_bench.hip -o gpurun_out/transpose_bench && gpurun_out/transpose_bench 30
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef double amp_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

struct Geo { int rd[9], wr[9]; uint64_t rd_outer, wr_outer; };

__device__ __forceinline__ uint64_t deposit(uint64_t x, uint64_t mask) {
    uint64_t out = 0;
    for (uint64_t bit = 1; mask; mask &= mask - 1) {
        const uint64_t lowest = mask & (0 - mask);
        if (x & bit) out |= lowest;
        bit <<= 1;
    }
    return out;
}

__global__ __launch_bounds__(512) void k_move(const amp_t *in, amp_t *out, Geo g, uint64_t ntiles, int tpw) {
    extern __shared__ amp_t lds[];
    const uint32_t tid = threadIdx.x;
    auto spread = [&](uint32_t j, const int *hi) {
        uint64_t o = 0;
        for (int i = 0; i < 9; i++) o |= (uint64_t)((j >> i) & 1u) << hi[i];
        return o;
    };
    // slot e = tid + k*512: run index e >> 3 = (tid >> 3) + k*64
    const uint64_t lane_in = spread(tid >> 3, g.rd) | (tid & 7u), lane_out = spread(tid >> 3, g.wr) | (tid & 7u);
    uint64_t k_in[8], k_out[8];
#pragma unroll
    for (int k = 0; k < 8; k++) { k_in[k] = spread((uint32_t)k * 64u, g.rd); k_out[k] = spread((uint32_t)k * 64u, g.wr); }
    const uint64_t first = (uint64_t)blockIdx.x * tpw;
    if (first >= ntiles) return;
    const uint64_t last = first + tpw < ntiles ? first + tpw : ntiles;
    amp_t pf[8];
    {
        const uint64_t b = deposit(first, g.rd_outer);
#pragma unroll
        for (int k = 0; k < 8; k++) pf[k] = in[(b | k_in[k]) + lane_in];
    }
    for (uint64_t t = first; t < last; t++) {
#pragma unroll
        for (int k = 0; k < 8; k++) lds[tid + k * 512] = pf[k];
        __syncthreads();
        if (t + 1 < last) {
            const uint64_t b = deposit(t + 1, g.rd_outer);
#pragma unroll
            for (int k = 0; k < 8; k++) pf[k] = in[(b | k_in[k]) + lane_in];
        }
        amp_t so[8];
#pragma unroll
        for (int k = 0; k < 8; k++) so[k] = lds[(tid + k * 512) ^ 1]; // something happens in LDS (neighbours swap)
        const uint64_t ob = deposit(t, g.wr_outer);
#pragma unroll
        for (int k = 0; k < 8; k++) out[(ob | k_out[k]) + lane_out] = so[k];
        __syncthreads();
    }
}

static double run(const amp_t *a, amp_t *b, int n, const int *rd, const int *wr, int reps) {
    Geo g;
    uint64_t rm = 7, wm = 7;
    for (int i = 0; i < 9; i++) { g.rd[i] = rd[i]; g.wr[i] = wr[i]; rm |= 1ULL << rd[i]; wm |= 1ULL << wr[i]; }
    const uint64_t nmask = (1ULL << n) - 1;
    g.rd_outer = nmask & ~rm; g.wr_outer = nmask & ~wm;
    const uint64_t ntiles = 1ULL << (n - 12);
    const int tpw = 8;
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    const unsigned grid = (unsigned)((ntiles + tpw - 1) / tpw);
    hipLaunchKernelGGL(k_move, dim3(grid), dim3(512), 65536, 0, a, b, g, ntiles, tpw);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(e0));
    for (int r = 0; r < reps; r++) hipLaunchKernelGGL(k_move, dim3(grid), dim3(512), 65536, 0, a, b, g, ntiles, tpw);
    CK(hipEventRecord(e1));
    CK(hipEventSynchronize(e1));
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    return ms / reps;
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 30;
    const size_t bytes = sizeof(amp_t) << n;
    amp_t *a, *b;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMemset(a, 0, bytes)); CK(hipMemset(b, 0, bytes));
    CK(hipFuncSetAttribute((const void *)k_move, hipFuncAttributeMaxDynamicSharedMemorySize, 65536));
    const int contiguous[9] = {3, 4, 5, 6, 7, 8, 9, 10, 11};
    // the high-bit sets of the planned bench schedule's full sweeps (profiles/r03/phase_split_n30.log) and three more
    const int sets[][9] = {{6, 7, 11, 15, 18, 23, 25, 28, 29}, {4, 5, 8, 11, 14, 16, 19, 22, 24}, {3, 9, 11, 12, 13, 16, 17, 21, 26},
                           {4, 6, 14, 15, 18, 23, 24, 27, 28}, {5, 8, 10, 17, 19, 22, 25, 26, 29}, {3, 8, 9, 11, 16, 20, 21, 27, 28},
                           {5, 7, 12, 13, 14, 15, 18, 24, 29}, {3, 4, 5, 11, 21, 22, 23, 26, 28}, {21, 22, 23, 24, 25, 26, 27, 28, 29},
                           {12, 13, 14, 15, 16, 17, 18, 19, 20}, {3, 4, 5, 6, 7, 8, 27, 28, 29}};
    const int nsets = (int)(sizeof sets / sizeof sets[0]);
    printf("n = %d, %.1f GiB read + %.1f GiB written per launch; ms per launch (TB/s)\n", n, bytes / 1073741824.0, bytes / 1073741824.0);
    const double gb = 2.0 * (double)bytes / 1e9;
    double tot[4] = {0, 0, 0, 0};
    {
        const double ms = run(a, b, n, contiguous, contiguous, 5);
        printf("contiguous -> contiguous              %6.3f (%.2f)\n", ms, gb / ms);
    }
    if (n < 30) { printf("the bit sets below need n >= 30\n"); return 1; }
    for (int s = 0; s < nsets; s++) {
        const int *S = sets[s], *N = sets[(s + 1) % nsets];
        const double both = run(a, b, n, S, S, 5), rd_c = run(a, b, n, contiguous, S, 5), wr_c = run(a, b, n, S, contiguous, 5), cross = run(a, b, n, S, N, 5);
        printf("set %2d: scattered both %6.3f (%.2f)  contiguous read, scattered write %6.3f (%.2f)  scattered read, contiguous write %6.3f (%.2f)  read set s, write set s+1 %6.3f\n",
               s, both, gb / both, rd_c, gb / rd_c, wr_c, gb / wr_c, cross);
        tot[0] += both; tot[1] += rd_c; tot[2] += wr_c; tot[3] += cross;
    }
    printf("mean: scattered both %.3f, contiguous read %.3f, contiguous write %.3f, set s -> set s+1 %.3f ms\n", tot[0] / nsets, tot[1] / nsets, tot[2] / nsets, tot[3] / nsets);
    CK(hipFree(a)); CK(hipFree(b));
    return 0;
}
