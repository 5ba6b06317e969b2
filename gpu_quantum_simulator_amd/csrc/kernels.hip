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

constexpr int TPB = 256;                // 4 waves of 64
constexpr uint64_t kMaxGrid = 1u << 22; // beyond this the kernels loop (grid-stride over work tiles)

__device__ __forceinline__ uint64_t insert_zero(uint64_t t, int q) {
    return ((t >> q) << (q + 1)) | (t & ((1ULL << q) - 1ULL));
}

// second __launch_bounds__ argument of k_tile (minimum waves per SIMD the register allocation must allow)
#define QSIM_REAL double
#define QSIM_AMP_SHIFT 4
#define QSIM_COEF_PAIRS 0
#define QSIM_TILE_MIN_WAVES(T) 1 /* fp64: B=13 x 512 threads deliberately runs at 210 VGPRs, one workgroup per CU */
namespace f64 {
#include "kernels_impl.inc"
} // namespace f64
#undef QSIM_REAL
#undef QSIM_AMP_SHIFT
#undef QSIM_COEF_PAIRS
#undef QSIM_TILE_MIN_WAVES

#define QSIM_REAL float
#define QSIM_AMP_SHIFT 3
#define QSIM_COEF_PAIRS 1 /* block coefficients arrive as (ur, ui) / (-ui, ur) pairs: v_pk_fma_f32 (kernels_impl.inc coef_t) */
#define QSIM_TILE_MIN_WAVES(T) ((T) >= 512 ? 4 : 1) /* fp32 tiles are half the LDS bytes: always two workgroups per CU */
namespace f32 {
#include "kernels_impl.inc"
} // namespace f32
#undef QSIM_REAL
#undef QSIM_AMP_SHIFT
#undef QSIM_COEF_PAIRS
#undef QSIM_TILE_MIN_WAVES

// ---- precision dispatch (the engine passes the state's precision with every launch) ------------------------------
#define QSIM_DISPATCH(CALL) (f32 ? f32::CALL : f64::CALL)
hipError_t launch_init(const LaunchCfg &cfg, void *v, bool f32, int n, double amp0) { return QSIM_DISPATCH(launch_init(cfg, v, n, amp0)); }
hipError_t launch_zero_outside(const LaunchCfg &cfg, void *v, bool f32, int n, uint64_t zero_mask) { return QSIM_DISPATCH(launch_zero_outside(cfg, v, n, zero_mask)); }
hipError_t launch_gate1(const LaunchCfg &cfg, void *v, bool f32, int n, int q, const M2 &U) { return QSIM_DISPATCH(launch_gate1(cfg, v, n, q, U)); }
hipError_t launch_phase(const LaunchCfg &cfg, void *v, bool f32, int n, int q, double lr, double li) { return QSIM_DISPATCH(launch_phase(cfg, v, n, q, lr, li)); }
hipError_t launch_diag1_full(const LaunchCfg &cfg, void *v, bool f32, int n, int q, double d0r, double d0i, double d1r, double d1i) {
    return QSIM_DISPATCH(launch_diag1_full(cfg, v, n, q, d0r, d0i, d1r, d1i));
}
hipError_t launch_cx(const LaunchCfg &cfg, void *v, bool f32, int n, int control, int target) { return QSIM_DISPATCH(launch_cx(cfg, v, n, control, target)); }
hipError_t launch_gate2(const LaunchCfg &cfg, void *v, bool f32, int n, int q_hi, int q_lo, const M4 &U) { return QSIM_DISPATCH(launch_gate2(cfg, v, n, q_hi, q_lo, U)); }
hipError_t launch_tile(const LaunchCfg &cfg, void *v, void *vout, bool f32, const TileGeom &g, const TileOp *d_ops, int n_ops, int threads, bool from_zero_ket,
                       double amp0, bool nomem, uint64_t zero_mask, const PackMap *pack) {
    return QSIM_DISPATCH(launch_tile(cfg, v, vout, g, d_ops, n_ops, threads, from_zero_ket, amp0, nomem, zero_mask, pack));
}
bool launch_tile_can_pack(bool f32, const TileGeom &g, int threads) { return QSIM_DISPATCH(launch_tile_can_pack(g, threads)); }
hipError_t launch_norm2(const LaunchCfg &cfg, const void *v, bool f32, int n, double *d_out) { return QSIM_DISPATCH(launch_norm2(cfg, v, n, d_out)); }
hipError_t launch_block_prob(const LaunchCfg &cfg, const void *v, bool f32, int n, int block_bits, double *d_out) {
    return QSIM_DISPATCH(launch_block_prob(cfg, v, n, block_bits, d_out));
}
hipError_t launch_pack(const LaunchCfg &cfg, const void *in, void *out, void *const *blocks, bool f32, int n, const int *bits, int p, uint32_t skip_blocks, uint64_t zero_mask) {
    return QSIM_DISPATCH(launch_pack(cfg, in, out, blocks, n, bits, p, skip_blocks, zero_mask));
}
hipError_t launch_block_prob_masked(const LaunchCfg &cfg, const void *v, bool f32, uint64_t hi_mask, uint64_t lo_mask, double *d_out) {
    return QSIM_DISPATCH(launch_block_prob_masked(cfg, v, hi_mask, lo_mask, d_out));
}
hipError_t launch_gather_masked(const LaunchCfg &cfg, const void *v, bool f32, uint64_t base, uint64_t lo_mask, void *d_out) {
    return QSIM_DISPATCH(launch_gather_masked(cfg, v, base, lo_mask, d_out));
}
#undef QSIM_DISPATCH

} // namespace qsim
