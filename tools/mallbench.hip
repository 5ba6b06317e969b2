// mallbench.hip — does the 256 MB Infinity Cache keep what one kernel wrote for the next kernel to read?
// Measurement aid for DESIGN.md §4 (not part of the library).  A 16 GiB fp64-complex state is swept twice (v[i] *= c, in
// place): either sweep A over everything and then sweep B over everything (every byte crosses HBM four times), or
// chunk by chunk — A on chunk g, then B on chunk g — with chunks small enough to stay in the Infinity Cache between the
// two.  If the second form is faster, two tile passes can share one trip to HBM (two-level blocking).
//   hipcc -O3 --offload-arch=gfx950 tools/mallbench.hip -o gpurun_out/mallbench && gpurun_out/mallbench 30
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>

typedef double amp_t __attribute__((ext_vector_type(2)));
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); exit(1); } } while (0)

__global__ __launch_bounds__(256) void k_scale(amp_t *v, double c) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    amp_t a = v[i];
    v[i] = amp_t{a.x * c, a.y * c};
}
// strided second sweep: the same chunk, but walked in a different order (like the next pass's tiles)
__global__ __launch_bounds__(256) void k_scale_perm(amp_t *v, double c, int chunk_bits) {
    const uint64_t i = (uint64_t)blockIdx.x * 256 + threadIdx.x;
    // swap the low 3..11 index bits with the top bits of the chunk (keeps 128-B runs)
    const uint64_t lo = i & 7, mid = (i >> 3) & 511, hi = i >> 12;
    const int hb = chunk_bits - 12;
    const uint64_t j = lo | ((hi & ((1ULL << hb) - 1)) << 3) | (mid << (3 + hb));
    amp_t a = v[j];
    v[j] = amp_t{a.x * c, a.y * c};
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 30;
    const uint64_t N = 1ULL << n;
    amp_t *v;
    CK(hipMalloc(&v, N * sizeof(amp_t)));
    CK(hipMemset(v, 0, N * sizeof(amp_t)));
    hipStream_t s0, s1;
    CK(hipStreamCreate(&s0));
    CK(hipStreamCreate(&s1));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto timeit = [&](const char *name, auto &&body) {
        body();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0, s0));
        body();
        CK(hipEventRecord(e1, s0));
        CK(hipDeviceSynchronize());
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("%-46s %8.3f ms  (%.2f TB/s over the 4 x 16 GiB of the two sweeps)\n", name, ms, 4.0 * N * 16 / ms / 1e9);
        fflush(stdout);
    };
    timeit("two full sweeps", [&]() {
        hipLaunchKernelGGL(k_scale, dim3((unsigned)(N / 256)), dim3(256), 0, s0, v, 1.0);
        hipLaunchKernelGGL(k_scale, dim3((unsigned)(N / 256)), dim3(256), 0, s0, v, 1.0);
    });
    for (int cb : {20, 21, 22, 23, 24}) { // chunk of 2^cb amplitudes = 16 MiB .. 256 MiB
        const uint64_t C = 1ULL << cb;
        char name[96];
        snprintf(name, sizeof name, "chunks of %4llu MiB, A then B, one stream", (unsigned long long)(C * 16 >> 20));
        timeit(name, [&]() {
            for (uint64_t g = 0; g < N / C; g++) {
                hipLaunchKernelGGL(k_scale, dim3((unsigned)(C / 256)), dim3(256), 0, s0, v + g * C, 1.0);
                hipLaunchKernelGGL(k_scale, dim3((unsigned)(C / 256)), dim3(256), 0, s0, v + g * C, 1.0);
            }
        });
        snprintf(name, sizeof name, "chunks of %4llu MiB, B walks it permuted", (unsigned long long)(C * 16 >> 20));
        timeit(name, [&]() {
            for (uint64_t g = 0; g < N / C; g++) {
                hipLaunchKernelGGL(k_scale, dim3((unsigned)(C / 256)), dim3(256), 0, s0, v + g * C, 1.0);
                hipLaunchKernelGGL(k_scale_perm, dim3((unsigned)(C / 256)), dim3(256), 0, s0, v + g * C, 1.0, cb);
            }
        });
    }
    return 0;
}
