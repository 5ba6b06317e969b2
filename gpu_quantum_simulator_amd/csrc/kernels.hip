// kernels.hip — hand-written CDNA4 (gfx950) kernels for the state-vector hot path.
//
// State: 2^n amplitudes, fp64 complex, array-of-structs double2 (16 B) — one global_load_dwordx4 per
// amplitude, 1 KiB per wave-instruction when lanes are consecutive.  Every kernel is HBM-bound by
// design (0.44 flop/B for a dense 2x2, 0.94 flop/B for a dense 4x4; fp64 VALU peak is far above that),
// so the rules that matter are: 16 B per lane, consecutive lanes on consecutive amplitudes, several
// independent loads in flight per lane, no MFMA, no re-reads.
//
// What each kernel stands in for (file:line into the reference):
//   k_init        init_state_vector                  quantum_simulator_naive.cu:64-70, quantum_simulator.c:175-177
//   k_gate1_hi/lo execute_single_qubit_gate          quantum_simulator.c:81-92; kernel_gate naive.cu:72-95
//   k_phase       same, for diag(1, lambda) gates    quantum_simulator.c:190-208 (z s sdg t tdg rz)
//   k_cx          execute_cnot                       quantum_simulator.c:94-106; kernel_cnot naive.cu:97-122
//   k_gate2_hh    kernel_gate_4                      quantum_simulator_4x4.cu:109-146
//   k_tile        kernel_costant (whole op list in one launch, op data in constant/scalar memory)
//                                                    quantum_simulator_preproces_constant.cu:169-178 — rebuilt
//                                                    as a full-grid, LDS-tiled pass instead of one block
// Index arithmetic is 64-bit throughout (the reference's `int th_id` stops at n = 31, naive.cu:74).
#include "qsim_internal.h"

namespace qsim {

// Amplitudes travel as clang's native 2 x double vector: one global_load_dwordx4 / ds_read_b128 each and,
// unlike the HIP_vector_type wrapper, a first-class value (arrays of it stay in registers).
typedef double amp_t __attribute__((ext_vector_type(2)));
static_assert(sizeof(amp_t) == sizeof(double2), "amp_t must alias double2");

constexpr int TPB = 256;              // 4 waves of 64
constexpr uint64_t kMaxGrid = 1u << 22; // beyond this the kernels loop (grid-stride over work tiles)

__device__ __forceinline__ uint64_t insert_zero(uint64_t t, int q) {
    return ((t >> q) << (q + 1)) | (t & ((1ULL << q) - 1ULL));
}

// r = a*u (complex), then r += b*w — written as explicit FMAs so hipcc keeps one v_fma_f64 each.
__device__ __forceinline__ amp_t cmul(amp_t a, double ur, double ui) {
    amp_t r;
    r.x = fma(a.x, ur, -(a.y * ui));
    r.y = fma(a.x, ui, a.y * ur);
    return r;
}
__device__ __forceinline__ amp_t cfma(amp_t a, double ur, double ui, amp_t acc) {
    amp_t r;
    r.x = fma(a.x, ur, fma(-a.y, ui, acc.x));
    r.y = fma(a.x, ui, fma(a.y, ur, acc.y));
    return r;
}
__device__ __forceinline__ amp_t shfl_xor2(amp_t a, int mask) {
    amp_t r;
    r.x = __shfl_xor(a.x, mask, 64);
    r.y = __shfl_xor(a.y, mask, 64);
    return r;
}

// ---------------------------------------------------------------------------------------------------
// |0...0>
__global__ __launch_bounds__(TPB) void k_init(amp_t *__restrict__ v, uint64_t N, double amp0) {
    const uint64_t stride = (uint64_t)gridDim.x * TPB;
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < N; i += stride)
        v[i] = amp_t{i == 0 ? amp0 : 0.0, 0.0};
}

// ---------------------------------------------------------------------------------------------------
// Dense 2x2, target bit q >= 6.  Work item = amplitude pair (i0, i0 | 2^q); consecutive lanes take
// consecutive i0, so each wave-instruction reads/writes one contiguous KiB from each of two streams
// 2^q amplitudes apart.  IPT pairs per thread -> 2*IPT independent 16-B loads in flight per lane.
template <int IPT, bool GUARD>
__global__ __launch_bounds__(TPB) void k_gate1_hi(amp_t *__restrict__ v, uint64_t npairs, int q, M2 U,
                                                  uint64_t ntiles) {
    const uint64_t bit = 1ULL << q;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * (uint64_t)(TPB * IPT) + threadIdx.x;
        uint64_t i0[IPT];
        amp_t a0[IPT], a1[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            i0[k] = insert_zero(t, q);
            if (!GUARD || t < npairs) {
                a0[k] = v[i0[k]];
                a1[k] = v[i0[k] | bit];
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            if (!GUARD || t < npairs) {
                v[i0[k]] = cfma(a1[k], U.re[1], U.im[1], cmul(a0[k], U.re[0], U.im[0]));
                v[i0[k] | bit] = cfma(a1[k], U.re[3], U.im[3], cmul(a0[k], U.re[2], U.im[2]));
            }
        }
    }
}

// Dense 2x2, target bit q < 6: both amplitudes of a pair sit in the same wave's contiguous KiB.  Each
// lane loads its own amplitude (perfectly coalesced), fetches the partner's with a wave shuffle
// (lane ^ 2^q — the butterfly), and computes its own output row.  Same flops per amplitude as the pair
// form, no second pass, no LDS allocation.
template <int IPT, bool GUARD>
__global__ __launch_bounds__(TPB) void k_gate1_lo(amp_t *__restrict__ v, uint64_t N, int q, M2 U, uint64_t ntiles) {
    const bool up = (threadIdx.x >> q) & 1; // bit q of the amplitude index == bit q of the lane id
    const double own_r = up ? U.re[3] : U.re[0], own_i = up ? U.im[3] : U.im[0];
    const double par_r = up ? U.re[2] : U.re[1], par_i = up ? U.im[2] : U.im[1];
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t i0 = tile * (uint64_t)(TPB * IPT) + threadIdx.x;
        amp_t a[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t i = i0 + (uint64_t)k * TPB;
            a[k] = (!GUARD || i < N) ? v[i] : amp_t{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t i = i0 + (uint64_t)k * TPB;
            const amp_t p = shfl_xor2(a[k], 1 << q);
            const amp_t r = cfma(p, par_r, par_i, cmul(a[k], own_r, own_i));
            if (!GUARD || i < N) v[i] = r;
        }
    }
}

// diag(1, lambda): only the bit=1 half is read and written (16*N bytes instead of 32*N).
template <int IPT, bool GUARD>
__global__ __launch_bounds__(TPB) void k_phase(amp_t *__restrict__ v, uint64_t nitems, int q, double lr, double li,
                                               uint64_t ntiles) {
    const uint64_t bit = 1ULL << q;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * (uint64_t)(TPB * IPT) + threadIdx.x;
        uint64_t idx[IPT];
        amp_t a[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            idx[k] = insert_zero(t, q) | bit;
            if (!GUARD || t < nitems) a[k] = v[idx[k]];
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            if (!GUARD || t < nitems) v[idx[k]] = cmul(a[k], lr, li);
        }
    }
}

// diag(d0, d1) over every amplitude (used when d0 != 1, or when q < 2 makes the half form touch every
// 64-B sector anyway).
template <int IPT, bool GUARD>
__global__ __launch_bounds__(TPB) void k_diag1_full(amp_t *__restrict__ v, uint64_t N, int q, double d0r, double d0i,
                                                    double d1r, double d1i, uint64_t ntiles) {
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t i0 = tile * (uint64_t)(TPB * IPT) + threadIdx.x;
        amp_t a[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t i = i0 + (uint64_t)k * TPB;
            if (!GUARD || i < N) a[k] = v[i];
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t i = i0 + (uint64_t)k * TPB;
            const bool up = (i >> q) & 1;
            if (!GUARD || i < N) v[i] = cmul(a[k], up ? d1r : d0r, up ? d1i : d0i);
        }
    }
}

// CX: swap v[i | c] <-> v[i | c | t] over the N/4 indices i with both bits clear.
template <int IPT, bool GUARD>
__global__ __launch_bounds__(TPB) void k_cx(amp_t *__restrict__ v, uint64_t nitems, int lo, int hi, uint64_t cbit,
                                            uint64_t tbit, uint64_t ntiles) {
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * (uint64_t)(TPB * IPT) + threadIdx.x;
        uint64_t ia[IPT];
        amp_t a[IPT], b[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            ia[k] = insert_zero(insert_zero(t, lo), hi) | cbit;
            if (!GUARD || t < nitems) {
                a[k] = v[ia[k]];
                b[k] = v[ia[k] | tbit];
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            if (!GUARD || t < nitems) {
                v[ia[k]] = b[k];
                v[ia[k] | tbit] = a[k];
            }
        }
    }
}

// Dense 4x4 with both target bits >= 6: four coalesced streams, all arithmetic in registers, matrix in
// kernel arguments (scalar registers).  Row/column index = (bit hi, bit lo), row-major
// (quantum_simulator_4x4.cu:119-134).
template <int IPT, bool GUARD>
__global__ __launch_bounds__(TPB) void k_gate2_hh(amp_t *__restrict__ v, uint64_t nitems, int lo, int hi, M4 U,
                                                  uint64_t ntiles) {
    const uint64_t blo = 1ULL << lo, bhi = 1ULL << hi;
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t t0 = tile * (uint64_t)(TPB * IPT) + threadIdx.x;
        uint64_t i00[IPT];
        amp_t x[IPT][4];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            i00[k] = insert_zero(insert_zero(t, lo), hi);
            if (!GUARD || t < nitems) {
                x[k][0] = v[i00[k]];
                x[k][1] = v[i00[k] | blo];
                x[k][2] = v[i00[k] | bhi];
                x[k][3] = v[i00[k] | bhi | blo];
            }
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t t = t0 + (uint64_t)k * TPB;
            if (!GUARD || t < nitems) {
                amp_t y[4];
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    amp_t acc = cmul(x[k][0], U.re[4 * r], U.im[4 * r]);
#pragma unroll
                    for (int c = 1; c < 4; c++) acc = cfma(x[k][c], U.re[4 * r + c], U.im[4 * r + c], acc);
                    y[r] = acc;
                }
                v[i00[k]] = y[0];
                v[i00[k] | blo] = y[1];
                v[i00[k] | bhi] = y[2];
                v[i00[k] | bhi | blo] = y[3];
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// Cache-blocked pass.  A tile is the 2^B amplitudes that agree on every index bit outside the tile set
// T = {0..L-1} U {high[0..H-1]}; it is 2^H contiguous runs of 2^L amplitudes (16*2^L bytes each), so
// global traffic stays in whole-KiB pieces whatever the high qubits are.  One workgroup stages a tile
// in LDS, applies the whole op list to it (every op's qubits lie in T), and writes it back: one read
// and one write of the state for n_ops fused blocks.  Op matrices are read with wave-uniform
// addresses from a const __restrict__ buffer (scalar loads through the constant cache).
//
// Inside the tile an op on local bit b pairs LDS slots (i, i | 2^b); each pair/quad is owned by one
// thread, so only a workgroup barrier between ops is needed.
// Geometry as the kernel sees it: the high tile bits as a mask (no runtime-indexed arrays in device code).
struct TileDev {
    int32_t tile_bits, low_bits, n_high, n;
    int32_t from_zero_ket, pad; // 1: the state is a basis state that has not been written yet: generate it, do not load it
    double amp0;                // its amplitude at index 0 (1 for |0...0>, 0 for a shard that does not hold index 0)
    uint64_t high_mask; // global bit positions of tile-local bits L..B-1
};

// software PDEP: spreads the low bits of x over the set bits of mask, lowest first
__device__ __forceinline__ uint64_t deposit(uint64_t x, uint64_t mask) {
    uint64_t out = 0;
    while (mask) {
        const uint64_t low = mask & (0 - mask);
        if (x & 1ULL) out |= low;
        x >>= 1;
        mask &= mask - 1;
    }
    return out;
}

// Op data is read through the CONSTANT address space with wave-uniform addresses, so hipcc emits scalar
// loads (s_load_dwordx*): the matrix lives in SGPRs / the scalar cache, not in vector registers.
typedef const TileOp __attribute__((address_space(4))) *ConstOps;

// LDS layout swizzle of a tile slot index: slot bits 0..3 (the 16-byte unit inside a 256-byte bank row) are XORed
// with slot bit 4.  An op on tile-local bit h <= 3 makes 16 consecutive lanes touch the slots whose bit h is fixed:
// unswizzled they fall on 8 of the 16 units (2-way bank conflict on every ds_read/ds_write_b128, 29 % of all LDS
// cycles on the bench circuit); with the parity column added any four of the slot bits 0..4 map to independent unit
// bits, so all single-hole patterns are conflict-free.  The map is linear over XOR: sw(a | b) = sw(a) ^ sw(b) for
// disjoint a, b, so per-thread bases are swizzled once and the wave-uniform operand offsets arrive pre-swizzled
// from the host (TileOp::off / rowoff) — no extra instruction per access.
__device__ __forceinline__ uint32_t sw_slot(uint32_t slot) { return slot ^ (((slot >> 4) & 1u) * 15u); }
__device__ __forceinline__ uint32_t sw_byte(uint32_t byte) { return byte ^ (((byte >> 8) & 1u) * 0xF0u); }

// LDS access by raw byte address.  k_tile has no static __shared__, so its dynamic LDS region starts at address 0
// (AMDGPU ABI: dynamic LDS follows the static part) and a tile byte offset IS the LDS address; going through the
// `extern __shared__` symbol instead costs one v_add_u32 (of a link-time zero) per access.
typedef __attribute__((address_space(3))) amp_t lds_amp_t;
__device__ __forceinline__ amp_t lds_load(uint32_t byte) { return *(lds_amp_t *)(uintptr_t)byte; }
__device__ __forceinline__ void lds_store(uint32_t byte, amp_t v) { *(lds_amp_t *)(uintptr_t)byte = v; }

// Index of the k-th work item with a zero inserted at bit b (b wave-uniform): x + (x & ~((1<<b)-1)).
__device__ __forceinline__ uint32_t ins0(uint32_t x, uint32_t himask) { return x + (x & himask); }

// Generic sparse block (TOP_SP): K qubits, T entries per row, all loop bounds static.
// SKIPS = false is the branch-free form for blocks without identity rows: one basic block, so the compiler can hoist
// the next rows' scalar loads and LDS reads above the current row's arithmetic (with the per-row skip branch every
// row is its own latency chain: s_load -> ds_read -> FMA).
template <int B, int THREADS, int K, int T, bool SKIPS>
__device__ __forceinline__ void tile_op_sparse(amp_t *lds, ConstOps ops, int oi, uint32_t tid) {
    constexpr uint32_t E = 1u << B;
    constexpr int R = 1 << K;
    constexpr uint32_t NG = E / R;                                   // groups of R amplitudes in the tile
    constexpr int GPT = NG >= (uint32_t)THREADS ? NG / THREADS : 1;  // groups per thread
    constexpr bool FULL = NG >= (uint32_t)THREADS && NG % THREADS == 0;
    const uint32_t b0 = (uint32_t)ops[oi].b[0], b1 = (uint32_t)ops[oi].b[1], b2 = (uint32_t)ops[oi].b[2];
    const uint32_t skip = (uint32_t)ops[oi].meta;
    (void)lds;
    uint32_t base[GPT]; // LDS BYTE address of the group's slot 0 (swizzled)
#pragma unroll
    for (int g = 0; g < GPT; g++) {
        uint32_t x = ins0(tid + g * THREADS, ~((1u << b0) - 1u));
        if (K >= 2) x = ins0(x, ~((1u << b1) - 1u));
        if (K >= 3) x = ins0(x, ~((1u << b2) - 1u));
        base[g] = sw_byte(x << 4);
    }
    amp_t y[GPT][R];
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (SKIPS && ((skip >> r) & 1u)) continue; // wave-uniform; everything below is straight-line per row
        uint32_t off[T];
        double cr[T], ci[T];
#pragma unroll
        for (int j = 0; j < T; j++) {
            const int e = r * T + j;
            off[j] = ops[oi].off[e];
            cr[j] = ops[oi].re[e];
            ci[j] = ops[oi].im[e];
        }
        amp_t x[GPT][T];
#pragma unroll
        for (int g = 0; g < GPT; g++)
#pragma unroll
            for (int j = 0; j < T; j++)
                if (FULL || tid + g * THREADS < NG) x[g][j] = lds_load(base[g] ^ off[j]); // T reads in flight
#pragma unroll
        for (int g = 0; g < GPT; g++)
            if (FULL || tid + g * THREADS < NG) {
                amp_t acc = cmul(x[g][0], cr[0], ci[0]);
#pragma unroll
                for (int j = 1; j < T; j++) acc = cfma(x[g][j], cr[j], ci[j], acc);
                y[g][r] = acc;
            }
    }
#pragma unroll
    for (int r = 0; r < R; r++) {
        if (SKIPS && ((skip >> r) & 1u)) continue;
        const uint32_t off = ops[oi].rowoff[r];
#pragma unroll
        for (int g = 0; g < GPT; g++)
            if (FULL || tid + g * THREADS < NG) lds_store(base[g] ^ off, y[g][r]);
    }
}

// B (tile size) and THREADS are compile-time so every per-thread loop has a static trip count: all LDS
// reads of an op are issued before its arithmetic, all writes after, and there is no loop bookkeeping.
template <int B, int THREADS>
__global__ __launch_bounds__(THREADS) void k_tile(amp_t *__restrict__ v, TileDev g, const TileOp *__restrict__ ops_g,
                                                  int n_ops, uint64_t ntiles, int tiles_per_wg) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    amp_t *lds = reinterpret_cast<amp_t *>(smem);
    constexpr uint32_t E = 1u << B;
    constexpr int APT = (E + THREADS - 1) / THREADS;                              // amplitudes per thread
    constexpr int PPT = E / 2 >= THREADS ? (E / 2) / THREADS : 1;                  // pairs per thread
    constexpr int QPT = E / 4 >= THREADS ? (E / 4) / THREADS : 1;                  // quads per thread
    constexpr bool FULL = E / 4 >= THREADS && (E / 4) % THREADS == 0;              // no tail guards needed
    const int L = g.low_bits, H = g.n_high;
    uint64_t *hoff = reinterpret_cast<uint64_t *>(smem + ((size_t)16 << B));
    ConstOps ops = (ConstOps)(uintptr_t)ops_g;
    const uint32_t tid = threadIdx.x;
    const uint32_t tid_sw = sw_slot(tid); // slot tid + k*THREADS swizzles to tid_sw + k*THREADS (THREADS is a multiple of 32)
    const uint32_t lowmask = (1u << L) - 1u;
    const uint64_t nmask = g.n >= 64 ? ~0ULL : ((1ULL << g.n) - 1ULL);
    const uint64_t outer_mask = nmask & ~(g.high_mask | (uint64_t)lowmask);

    // lds_load / lds_store address the tile by raw LDS byte address: that is only right while this kernel's dynamic
    // LDS region starts at 0, i.e. while nobody adds a static __shared__ array to it.  Fail loudly otherwise.
    if ((uint32_t)(uintptr_t)(__attribute__((address_space(3))) unsigned char *)smem != 0u) __builtin_trap();

    for (uint32_t j = tid; j < (1u << H); j += THREADS) hoff[j] = deposit(j, g.high_mask);
    __syncthreads();

    // A workgroup walks `tiles_per_wg` consecutive tiles.  While tile j is being processed in LDS, the loads of
    // tile j+1 are already in flight into registers (APT amplitudes per lane), which keeps bytes in flight during
    // the op phase: the kernel is latency-bound (bandwidth follows the number of workgroups currently in their
    // memory phase), and LDS capacity caps the resident tiles at two per CU.
    // vmcnt counts loads and stores together, in issue order, and the compiler merges its counter state at the
    // loop head: the first iteration is peeled so that BOTH ways into the loop carry [prefetch loads][stores of
    // the previous tile] — the wait it inserts before the prefetched registers are used is then vmcnt(#stores),
    // i.e. exact, instead of draining the previous tile's stores as well.  The prefetch is skipped (wave-uniform
    // branch) when there is no next tile; an unconditional re-fetch there showed up as +6 % FETCH_SIZE, and peeling
    // the last iteration as well (four copies of the op code) overflowed the instruction cache: 11.2 vs 7.8 ms/pass.
    const uint64_t first_tile = (uint64_t)blockIdx.x * (uint64_t)tiles_per_wg;
    if (first_tile >= ntiles) return;
    const int cnt = (int)((ntiles - first_tile) < (uint64_t)tiles_per_wg ? (ntiles - first_tile) : (uint64_t)tiles_per_wg);

    // Element k of a lane is tile slot e = tid + k*THREADS.  With THREADS >= 2^L the run index e >> L splits without
    // carry into (tid >> L) + k*(THREADS >> L), and the bit deposit is linear over disjoint bits, so the global byte
    // offset of the element is  lane_off (per lane, once per workgroup)  +  k_off[k] (wave-uniform)  +  tile base.
    // Tiles of a workgroup are consecutive: the next base is a masked increment, not another bit deposit.
    // (the engine guarantees 2^L <= 64 <= THREADS)
    const uint64_t lane_off = (hoff[(tid >> L) & ((1u << H) - 1u)] | (uint64_t)(tid & lowmask)) << 4;
    uint64_t k_off[APT];
#pragma unroll
    for (int k = 0; k < APT; k++) k_off[k] = deposit((uint64_t)(k * THREADS) >> L, g.high_mask) << 4;
    auto elem_ptr = [&](uint64_t tile_base, int k) -> amp_t * {
        return reinterpret_cast<amp_t *>(reinterpret_cast<unsigned char *>(v) + ((tile_base << 4) + k_off[k]) + lane_off);
    };
    auto next_base = [&](uint64_t b) { return ((b | ~outer_mask) + 1ULL) & outer_mask; }; // +1 scattered over the outer bits

    amp_t pf[APT];
    const bool generate = g.from_zero_ket != 0; // wave-uniform
    auto fetch = [&](uint64_t tb) {
        if (generate) { // |0...0>: amplitude 1 at global index 0 (tile base 0, slot 0), nothing to read
#pragma unroll
            for (int k = 0; k < APT; k++) pf[k] = amp_t{(tb == 0 && k == 0 && tid == 0) ? g.amp0 : 0.0, 0.0};
            return;
        }
#pragma unroll
        for (int k = 0; k < APT; k++) {
            const uint32_t e = tid + k * THREADS;
            pf[k] = (FULL || e < E) ? *elem_ptr(tb, k) : amp_t{0.0, 0.0};
        }
    };
    auto process = [&](uint64_t base, bool prefetch_next) {
#pragma unroll
        for (int k = 0; k < APT; k++) {
            const uint32_t e = tid + k * THREADS;
            if (FULL || e < E) lds[tid_sw + k * THREADS] = pf[k];
        }
        __syncthreads();
        if (prefetch_next) fetch(next_base(base));

        for (int oi = 0; oi < n_ops; oi++) {
            const int kind = ops[oi].kind;
            if (kind == TOP_SP) {
                const int nq = ops[oi].nq, terms = ops[oi].terms;
                const bool skips = ops[oi].meta != 0;
                if (nq == 2) {
                    if (terms == 1) { if (skips) tile_op_sparse<B, THREADS, 2, 1, true>(lds, ops, oi, tid); else tile_op_sparse<B, THREADS, 2, 1, false>(lds, ops, oi, tid); }
                    else { if (skips) tile_op_sparse<B, THREADS, 2, 2, true>(lds, ops, oi, tid); else tile_op_sparse<B, THREADS, 2, 2, false>(lds, ops, oi, tid); }
                } else {
                    if (terms == 1) { if (skips) tile_op_sparse<B, THREADS, 3, 1, true>(lds, ops, oi, tid); else tile_op_sparse<B, THREADS, 3, 1, false>(lds, ops, oi, tid); }
                    else if (terms == 2) { if (skips) tile_op_sparse<B, THREADS, 3, 2, true>(lds, ops, oi, tid); else tile_op_sparse<B, THREADS, 3, 2, false>(lds, ops, oi, tid); }
                    else tile_op_sparse<B, THREADS, 3, 4, false>(lds, ops, oi, tid); // a 4-entry row is never an identity row
                }
            } else if (kind == TOP_G2) {
                const uint32_t bl = (uint32_t)ops[oi].b[0], bh = (uint32_t)ops[oi].b[1];
                const uint32_t hm_lo = ~((1u << bl) - 1u), hm_hi = ~((1u << bh) - 1u);
                const uint32_t o1 = sw_slot(1u << bl), o2 = sw_slot(1u << bh), o3 = o1 ^ o2; // wave-uniform
                double ur[16], ui[16];
#pragma unroll
                for (int k = 0; k < 16; k++) { ur[k] = ops[oi].re[k]; ui[k] = ops[oi].im[k]; }
                uint32_t i00[QPT];
                amp_t x[QPT][4];
#pragma unroll
                for (int k = 0; k < QPT; k++) {
                    i00[k] = sw_slot(ins0(ins0(tid + k * THREADS, hm_lo), hm_hi));
                    if (FULL || tid + k * THREADS < E / 4) {
                        x[k][0] = lds[i00[k]]; x[k][1] = lds[i00[k] ^ o1]; x[k][2] = lds[i00[k] ^ o2]; x[k][3] = lds[i00[k] ^ o3];
                    }
                }
#pragma unroll
                for (int k = 0; k < QPT; k++)
                    if (FULL || tid + k * THREADS < E / 4) {
                        amp_t y[4];
#pragma unroll
                        for (int r = 0; r < 4; r++) {
                            amp_t acc = cmul(x[k][0], ur[4 * r], ui[4 * r]);
                            acc = cfma(x[k][1], ur[4 * r + 1], ui[4 * r + 1], acc);
                            acc = cfma(x[k][2], ur[4 * r + 2], ui[4 * r + 2], acc);
                            acc = cfma(x[k][3], ur[4 * r + 3], ui[4 * r + 3], acc);
                            y[r] = acc;
                        }
                        lds[i00[k]] = y[0]; lds[i00[k] ^ o1] = y[1]; lds[i00[k] ^ o2] = y[2]; lds[i00[k] ^ o3] = y[3];
                    }
            } else if (kind == TOP_G1) {
                const uint32_t bh = (uint32_t)ops[oi].b[0];
                const uint32_t hm = ~((1u << bh) - 1u), o1 = sw_slot(1u << bh);
                const double u0r = ops[oi].re[0], u0i = ops[oi].im[0], u1r = ops[oi].re[1], u1i = ops[oi].im[1];
                const double u2r = ops[oi].re[2], u2i = ops[oi].im[2], u3r = ops[oi].re[3], u3i = ops[oi].im[3];
                uint32_t i0[PPT];
                amp_t a0[PPT], a1[PPT];
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    i0[k] = sw_slot(ins0(tid + k * THREADS, hm));
                    if (FULL || tid + k * THREADS < E / 2) { a0[k] = lds[i0[k]]; a1[k] = lds[i0[k] ^ o1]; }
                }
#pragma unroll
                for (int k = 0; k < PPT; k++)
                    if (FULL || tid + k * THREADS < E / 2) {
                        lds[i0[k]] = cfma(a1[k], u1r, u1i, cmul(a0[k], u0r, u0i));
                        lds[i0[k] ^ o1] = cfma(a1[k], u3r, u3i, cmul(a0[k], u2r, u2i));
                    }
            } else { // TOP_DIAG1
                const uint32_t bh = (uint32_t)ops[oi].b[0];
                const uint32_t hm = ~((1u << bh) - 1u), o1 = sw_slot(1u << bh);
                const double d0r = ops[oi].re[0], d0i = ops[oi].im[0], d1r = ops[oi].re[1], d1i = ops[oi].im[1];
                const bool unit0 = ops[oi].meta & 1;
                uint32_t i0[PPT];
                amp_t a0[PPT], a1[PPT];
#pragma unroll
                for (int k = 0; k < PPT; k++) {
                    i0[k] = sw_slot(ins0(tid + k * THREADS, hm));
                    if (FULL || tid + k * THREADS < E / 2) {
                        if (!unit0) a0[k] = lds[i0[k]];
                        a1[k] = lds[i0[k] ^ o1];
                    }
                }
#pragma unroll
                for (int k = 0; k < PPT; k++)
                    if (FULL || tid + k * THREADS < E / 2) {
                        if (!unit0) lds[i0[k]] = cmul(a0[k], d0r, d0i);
                        lds[i0[k] ^ o1] = cmul(a1[k], d1r, d1i);
                    }
            }
            __syncthreads();
        }

        // stage out
#pragma unroll
        for (int k = 0; k < APT; k++) {
            const uint32_t e = tid + k * THREADS;
            if (FULL || e < E) *elem_ptr(base, k) = lds[tid_sw + k * THREADS];
        }
        __syncthreads();
    };

    uint64_t base = deposit(first_tile, outer_mask); // wave-uniform
    fetch(base);
    process(base, cnt > 1);                                                   // peeled first iteration
    for (int j = 1; j < cnt; j++) {                                           // steady state; the last one fetches nothing
        base = next_base(base);
        process(base, j + 1 < cnt);
    }
}

// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(TPB) void k_norm2(const amp_t *__restrict__ v, uint64_t N, double *out) {
    double acc = 0.0;
    const uint64_t stride = (uint64_t)gridDim.x * TPB;
    for (uint64_t i = (uint64_t)blockIdx.x * TPB + threadIdx.x; i < N; i += stride) {
        const amp_t a = v[i];
        acc = fma(a.x, a.x, fma(a.y, a.y, acc));
    }
    for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
    __shared__ double part[TPB / 64];
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double s = 0.0;
        for (int w = 0; w < TPB / 64; w++) s += part[w];
        atomicAdd(out, s);
    }
}

// Probability mass per block of 2^block_bits amplitudes (measurement post-path).  One workgroup per block; the
// reduction order is fixed (lane-strided partial sums, xor-butterfly inside the wave, waves added in order), so the
// result does not depend on scheduling.
__global__ __launch_bounds__(TPB) void k_block_prob(const amp_t *__restrict__ v, uint64_t N, int block_bits,
                                                    double *__restrict__ out, uint64_t nblocks) {
    __shared__ double part[TPB / 64];
    for (uint64_t b = blockIdx.x; b < nblocks; b += gridDim.x) {
        const uint64_t lo = b << block_bits;
        uint64_t hi = lo + (1ULL << block_bits);
        if (hi > N) hi = N;
        double acc = 0.0;
        for (uint64_t i = lo + threadIdx.x; i < hi; i += TPB) {
            const amp_t a = v[i];
            acc += fma(a.x, a.x, a.y * a.y);
        }
        for (int m = 32; m >= 1; m >>= 1) acc += __shfl_xor(acc, m, 64);
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = acc;
        __syncthreads();
        if (threadIdx.x == 0) {
            double t = 0.0;
            for (int w = 0; w < TPB / 64; w++) t += part[w];
            out[b] = t;
        }
        __syncthreads();
    }
}

// Shard re-layout ahead of a global<->local qubit exchange: gathers so that the p selected index bits
// become the top p bits (the destination block id) while the other bits keep their order.  Writes are
// fully coalesced; reads come in runs of 2^bits[0] amplitudes.
// Scatter form: consecutive lanes READ consecutive amplitudes (always fully coalesced); a wave's 64 stores
// fall into 2^(selected bits below 6) contiguous segments, i.e. >= 128 B pieces for up to three selected
// bits wherever they are.  (The gather form would read 16/32-B fragments when bit 0 or 1 is selected and
// fetch those sectors once per destination block.)
// dst = (pext(src, sel) << rest_bits) | pext(src, keep).  PEXT over disjoint bit ranges splits, so the part
// that depends on the work tile is wave-uniform scalar work and the part that depends on the lane is
// computed once per thread, outside the tile loop.
__device__ __forceinline__ uint64_t extract(uint64_t x, uint64_t mask) { // software PEXT
    uint64_t out = 0;
    int k = 0;
    while (mask) {
        const uint64_t low = mask & (0 - mask);
        if (x & low) out |= 1ULL << k;
        k++;
        mask &= mask - 1;
    }
    return out;
}

template <int IPT>
__global__ __launch_bounds__(TPB) void k_pack(const amp_t *__restrict__ in, amp_t *__restrict__ out, uint64_t N, int n,
                                              int p, uint64_t sel_mask, uint64_t ntiles) {
    constexpr int SB = 10; // log2(TPB * IPT): index bits owned by the position inside a work tile
    static_assert(TPB * IPT == (1 << SB), "tile split");
    const int rest_bits = n - p;
    const uint64_t nmask = n >= 64 ? ~0ULL : ((1ULL << n) - 1ULL);
    const uint64_t keep_mask = nmask & ~sel_mask;
    const uint64_t lo = (1ULL << SB) - 1ULL;
    const int pc_keep_lo = __popcll(keep_mask & lo), pc_sel_lo = __popcll(sel_mask & lo);
    uint64_t add[IPT];
#pragma unroll
    for (int k = 0; k < IPT; k++) {
        const uint64_t e = (uint64_t)k * TPB + threadIdx.x;
        add[k] = (extract(e, sel_mask & lo) << rest_bits) | extract(e, keep_mask & lo);
    }
    for (uint64_t tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
        const uint64_t base = ((extract(tile, sel_mask >> SB) << pc_sel_lo) << rest_bits) |
                              (extract(tile, keep_mask >> SB) << pc_keep_lo); // wave-uniform
        const uint64_t s0 = (tile << SB) + threadIdx.x;
        amp_t a[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t sidx = s0 + (uint64_t)k * TPB;
            a[k] = sidx < N ? in[sidx] : amp_t{0.0, 0.0};
        }
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const uint64_t sidx = s0 + (uint64_t)k * TPB;
            if (sidx < N) out[base | add[k]] = a[k];
        }
    }
}

// ===================================================================================================
// launchers
static inline uint64_t ceil_div(uint64_t a, uint64_t b) { return (a + b - 1) / b; }
static inline unsigned grid_for(const LaunchCfg &cfg, uint64_t ntiles) {
    uint64_t g = ntiles;
    if (cfg.grid_cap > 0 && g > (uint64_t)cfg.grid_cap) g = (uint64_t)cfg.grid_cap;
    if (g > kMaxGrid) g = kMaxGrid;
    return (unsigned)(g ? g : 1);
}

hipError_t launch_init(const LaunchCfg &cfg, double2 *v, int n, double amp0) {
    const uint64_t N = 1ULL << n;
    uint64_t blocks = ceil_div(N, TPB);
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(k_init, dim3((unsigned)blocks), dim3(TPB), 0, cfg.stream, (amp_t *)v, N, amp0);
    return hipGetLastError();
}

#define QSIM_DISPATCH_GUARD(KERN, IPT, items, ...)                                                                    \
    do {                                                                                                              \
        const uint64_t nt_ = ceil_div((items), (uint64_t)TPB * (IPT));                                                \
        if ((items) % ((uint64_t)TPB * (IPT)) == 0)                                                                   \
            hipLaunchKernelGGL((KERN<IPT, false>), dim3(grid_for(cfg, nt_)), dim3(TPB), 0, cfg.stream, __VA_ARGS__,    \
                               nt_);                                                                                  \
        else                                                                                                          \
            hipLaunchKernelGGL((KERN<IPT, true>), dim3(grid_for(cfg, nt_)), dim3(TPB), 0, cfg.stream, __VA_ARGS__,     \
                               nt_);                                                                                  \
    } while (0)

hipError_t launch_gate1(const LaunchCfg &cfg, double2 *v, int n, int q, const M2 &U) {
    const uint64_t N = 1ULL << n;
    if (q >= 6) {
        const uint64_t npairs = N >> 1;
        QSIM_DISPATCH_GUARD(k_gate1_hi, 4, npairs, (amp_t *)v, npairs, q, U);
    } else {
        QSIM_DISPATCH_GUARD(k_gate1_lo, 4, N, (amp_t *)v, N, q, U);
    }
    return hipGetLastError();
}

hipError_t launch_phase(const LaunchCfg &cfg, double2 *v, int n, int q, double lr, double li) {
    const uint64_t items = (1ULL << n) >> 1;
    QSIM_DISPATCH_GUARD(k_phase, 4, items, (amp_t *)v, items, q, lr, li);
    return hipGetLastError();
}

hipError_t launch_diag1_full(const LaunchCfg &cfg, double2 *v, int n, int q, double d0r, double d0i, double d1r,
                             double d1i) {
    const uint64_t N = 1ULL << n;
    QSIM_DISPATCH_GUARD(k_diag1_full, 4, N, (amp_t *)v, N, q, d0r, d0i, d1r, d1i);
    return hipGetLastError();
}

hipError_t launch_cx(const LaunchCfg &cfg, double2 *v, int n, int control, int target) {
    if (control == target) return hipSuccess; // quantum_simulator.c:99 — no index qualifies
    const uint64_t items = (1ULL << n) >> 2;
    const int lo = control < target ? control : target, hi = control < target ? target : control;
    const uint64_t cbit = 1ULL << control, tbit = 1ULL << target;
    QSIM_DISPATCH_GUARD(k_cx, 4, items, (amp_t *)v, items, lo, hi, cbit, tbit);
    return hipGetLastError();
}

hipError_t launch_gate2(const LaunchCfg &cfg, double2 *v, int n, int q_hi, int q_lo, const M4 &U) {
    const uint64_t items = (1ULL << n) >> 2;
    QSIM_DISPATCH_GUARD(k_gate2_hh, 2, items, (amp_t *)v, items, q_lo, q_hi, U);
    return hipGetLastError();
}

int tile_lds_bytes(int tile_bits, int n_high) { return (16 << tile_bits) + (8 << n_high); }

template <int B, int THREADS>
static hipError_t launch_tile_t(const LaunchCfg &cfg, double2 *v, const TileGeom &g, const TileOp *d_ops, int n_ops,
                                bool from_zero_ket, double amp0) {
    const uint64_t ntiles = 1ULL << (g.n - g.tile_bits);
    const int lds = tile_lds_bytes(g.tile_bits, g.n_high);
    // the opt-in to more than 64 KiB of dynamic LDS is per device (a cluster drives several from one process)
    static bool attr_set[64] = {false};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
    if (!attr_set[dev]) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(&k_tile<B, THREADS>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (e != hipSuccess) return e;
        attr_set[dev] = true;
    }
    TileDev td;
    td.tile_bits = g.tile_bits; td.low_bits = g.low_bits; td.n_high = g.n_high; td.n = g.n;
    td.from_zero_ket = from_zero_ket ? 1 : 0;
    td.pad = 0;
    td.amp0 = amp0;
    td.high_mask = 0;
    for (int j = 0; j < g.n_high; j++) td.high_mask |= 1ULL << g.high[j];
    // tiles per workgroup: enough to amortise the exposed first load, few enough to keep >= 8 workgroups per CU slot
    int tpw = cfg.grid_cap > 0 ? (int)((ntiles + cfg.grid_cap - 1) / (uint64_t)cfg.grid_cap) : 8;
    while (tpw > 1 && ntiles / (uint64_t)tpw < 4096) tpw >>= 1;
    if (tpw < 1) tpw = 1;
    const uint64_t grid = (ntiles + tpw - 1) / (uint64_t)tpw;
    hipLaunchKernelGGL((k_tile<B, THREADS>), dim3((unsigned)grid), dim3(THREADS), lds, cfg.stream, (amp_t *)v, td, d_ops, n_ops,
                       ntiles, tpw);
    return hipGetLastError();
}

// threads: 0 = the default for the tile size.  Tiles below 2^8 amplitudes (tiny registers) use the 2^8 kernel's
// tail guards with a smaller E, so every size from 1 to 13 bits has an instantiation.
hipError_t launch_tile(const LaunchCfg &cfg, double2 *v, const TileGeom &g, const TileOp *d_ops, int n_ops, int threads,
                       bool from_zero_ket, double amp0) {
    switch (g.tile_bits) {
    case 0: return launch_tile_t<0, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 1: return launch_tile_t<1, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 2: return launch_tile_t<2, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 3: return launch_tile_t<3, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 4: return launch_tile_t<4, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 5: return launch_tile_t<5, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 6: return launch_tile_t<6, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 7: return launch_tile_t<7, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 8: return launch_tile_t<8, 64>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 9: return launch_tile_t<9, 128>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 10: return threads == 512 ? launch_tile_t<10, 512>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0) : launch_tile_t<10, 256>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 11: return threads == 512 ? launch_tile_t<11, 512>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0) : launch_tile_t<11, 256>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 12:
        if (threads == 256) return launch_tile_t<12, 256>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
        if (threads == 1024) return launch_tile_t<12, 1024>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
        return launch_tile_t<12, 512>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    case 13:
        if (threads == 512) return launch_tile_t<13, 512>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
        return launch_tile_t<13, 1024>(cfg, v, g, d_ops, n_ops, from_zero_ket, amp0);
    default: return hipErrorInvalidValue;
    }
}

hipError_t launch_norm2(const LaunchCfg &cfg, const double2 *v, int n, double *d_out) {
    const uint64_t N = 1ULL << n;
    uint64_t blocks = ceil_div(N, (uint64_t)TPB * 8);
    if (blocks > 4096) blocks = 4096;
    if (blocks == 0) blocks = 1;
    hipLaunchKernelGGL(k_norm2, dim3((unsigned)blocks), dim3(TPB), 0, cfg.stream, (const amp_t *)v, N, d_out);
    return hipGetLastError();
}

hipError_t launch_block_prob(const LaunchCfg &cfg, const double2 *v, int n, int block_bits, double *d_out) {
    const uint64_t N = 1ULL << n;
    const uint64_t nblocks = (N + (1ULL << block_bits) - 1) >> block_bits;
    uint64_t grid = nblocks > 65536 ? 65536 : nblocks;
    hipLaunchKernelGGL(k_block_prob, dim3((unsigned)grid), dim3(TPB), 0, cfg.stream, (const amp_t *)v, N, block_bits, d_out,
                       nblocks);
    return hipGetLastError();
}

hipError_t launch_pack(const LaunchCfg &cfg, const double2 *in, double2 *out, int n, const int *bits, int p) {
    uint64_t sel = 0;
    for (int j = 0; j < p; j++) sel |= 1ULL << bits[j];
    const uint64_t N = 1ULL << n;
    const uint64_t nt = ceil_div(N, (uint64_t)TPB * 4);
    unsigned grid = grid_for(cfg, nt);
    if (grid > 8192) grid = 8192; // persistent: the per-thread PEXT above is paid once per 2^10 * (nt / grid) amplitudes
    hipLaunchKernelGGL(k_pack<4>, dim3(grid), dim3(TPB), 0, cfg.stream, (const amp_t *)in, (amp_t *)out, N, n, p, sel, nt);
    return hipGetLastError();
}

} // namespace qsim
