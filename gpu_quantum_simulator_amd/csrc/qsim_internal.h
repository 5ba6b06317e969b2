// qsim_internal.h — types shared by the engine (host C++) and the HIP kernels.  Not installed.
#ifndef QSIM_INTERNAL_H
#define QSIM_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/qsim.h"

namespace qsim {

// Row-major complex matrices passed BY VALUE as kernel arguments (they land in SGPRs / the scalar
// cache: the "constant memory" of quantum_simulator_preproces_constant.cu:58-61 without the 64 KiB cap).
struct M2 { double re[4], im[4]; };
struct M4 { double re[16], im[16]; };

// One fused block inside a cache-blocked pass.  Bit positions are TILE-LOCAL (see TileGeom).
enum : int32_t { TOP_G1 = 1, TOP_G2 = 2, TOP_DIAG1 = 3, TOP_DIAG2 = 4 };
struct TileOp {
    int32_t kind;
    int32_t b_hi;     // tile-local bit of the high qubit (G2/DIAG2) or the target (G1/DIAG1)
    int32_t b_lo;     // tile-local bit of the low qubit (G2/DIAG2), unused otherwise
    int32_t pad;
    double re[16];    // row-major; G1 uses [0..3], DIAG1 [0..1], DIAG2 [0..3]
    double im[16];
};
static_assert(sizeof(TileOp) == 272, "TileOp layout is shared with the device");

constexpr int kMaxTileHigh = 8; // high (non-contiguous) qubits per tile
struct TileGeom {
    int32_t tile_bits;          // B: log2 amplitudes per tile
    int32_t low_bits;           // L: tile-local bits [0,L) are global bits [0,L)
    int32_t n_high;             // B - L
    int32_t n;                  // qubits in this state (shard)
    int32_t high[kMaxTileHigh]; // ascending global bit of tile-local bit L+j
};

struct LaunchCfg {
    hipStream_t stream;
    int grid_cap; // 0 = one workgroup per work tile
};

// All launchers are asynchronous on cfg.stream and return the hipError_t of the launch.
hipError_t launch_init(const LaunchCfg &cfg, double2 *v, int n);
hipError_t launch_gate1(const LaunchCfg &cfg, double2 *v, int n, int q, const M2 &U);
hipError_t launch_phase(const LaunchCfg &cfg, double2 *v, int n, int q, double lr, double li);
hipError_t launch_diag1_full(const LaunchCfg &cfg, double2 *v, int n, int q, double d0r, double d0i, double d1r,
                             double d1i);
hipError_t launch_cx(const LaunchCfg &cfg, double2 *v, int n, int control, int target);
hipError_t launch_gate2(const LaunchCfg &cfg, double2 *v, int n, int q_hi, int q_lo, const M4 &U);
hipError_t launch_tile(const LaunchCfg &cfg, double2 *v, const TileGeom &g, const TileOp *d_ops, int n_ops);
hipError_t launch_norm2(const LaunchCfg &cfg, const double2 *v, int n, double *d_out /* zeroed */);
// out[dst] = in[src]: dst = (block << (n-p)) | rest, where block = the p bits of src at positions
// `bits` (ascending) and rest = the remaining n-p bits of src in order.
hipError_t launch_pack(const LaunchCfg &cfg, const double2 *in, double2 *out, int n, const int *bits, int p);

int tile_lds_bytes(int tile_bits);

} // namespace qsim
#endif
