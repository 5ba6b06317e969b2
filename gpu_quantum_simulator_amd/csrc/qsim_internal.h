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

// One fused block inside a cache-blocked pass.  Bit positions are TILE-LOCAL (see TileGeom), ascending:
// b[0] is the block's lowest qubit = bit 0 of a slot code, b[k-1] its highest = the slot code's top bit.
//   TOP_G1     dense 2x2 on b[0]                      re/im[bank][0..3] row-major
//   TOP_DIAG1  diag(d0, d1) on b[0]                   re/im[bank][0..1]; meta[bank] bit 0: d0 == 1 (only the bit=1 half moves)
//   TOP_G2     dense 4x4 on (b[1], b[0])              re/im[bank][0..15] row-major, operands held in registers
//   TOP_SP     sparse 2^k x 2^k block, k = nq in {2..5}, `terms` = T (1, 2 or 4) entries per row.  Per bank the block is a
//              direct sum of small dense matrices in a permuted basis (Scheduler / TileBlock::classes), so its rows are
//              stored CLASS by class: positions cT .. cT+T-1 are T rows that read the same T operand slots.  A class
//              costs T LDS reads for T outputs — ONE read per amplitude whatever T is (the first version read every
//              row's operands separately: T reads per amplitude; the block phase of a pass is LDS-bound, so that
//              was a third of its time):
//                  x[j] = slot off[bank][cT + j]               (one operand list per class)
//                  y[cT+i] = sum_j coef[bank][(cT+i)*T + j] * x[j]     -> written to slot rowoff[bank][cT+i]
//              Offsets are ready LDS byte offsets (wave-uniform, layout swizzle applied).  meta[bank] bit p: position p is
//              an identity row; a class whose T rows all are costs nothing (identity rows are packed together).
//              Most fused clusters of Clifford+T-like circuits are permutations times phases or two independent
//              2x2 blocks, and so are products of neighbouring ones on a few qubits (Scheduler::merge_blocks).
//              k <= 3: one lane owns a whole group of 2^k amplitudes (reads, then writes).  k = 4, 5: the 2^k positions of a
//              group are split over 2 or 4 lanes (8 positions = whole classes each, in different waves), with a workgroup
//              barrier between everybody's reads and the writes.
//   TOP_SCALE  no qubit inside the tile: a factor per tile, applied while the tile is staged in.
// BANKS.  A block may also depend on up to two qubits OUTSIDE the tile, provided it is block-diagonal in them (a CX
// whose control is outside, any diagonal gate): such a qubit is constant over a tile, so it merely selects which
// 2^k x 2^k sub-block the tile gets.  nsel / selbit name those GLOBAL index bits; the bank index is
// bit(selbit[0]) or 2*bit(selbit[0]) + bit(selbit[1]) of the tile's base index, wave-uniform, and only that bank's
// offsets and coefficients are ever loaded.  `ident` bit v: bank v is the identity — the block is skipped on those tiles.
// TOP_SCALE uses the same selection and reads its factor from re/im[bank][0].
enum : int32_t { TOP_G1 = 1, TOP_G2 = 2, TOP_DIAG1 = 3, TOP_SP = 4, TOP_SCALE = 5 };
constexpr int kMaxBanks = 4;
constexpr int kMaxOpQ = 5;                      // TOP_SP blocks span 2..5 tile qubits
constexpr int kMaxOpEntries = 4 << kMaxOpQ;     // 4 entries for each of 32 rows
struct TileOp {
    // 64-byte header; the kernel fetches its first 32 bytes with ONE scalar load (s_load_dwordx8) and decodes them with
    // scalar bit-field extracts, one block ahead of the block it is working on — read field by field, the dispatch was a
    // chain of six dependent scalar-load round trips per block and wave (kind -> selectors -> bank -> identity -> shape ->
    // tile bits) before the first LDS read could be issued.
    uint8_t kind;         // dword 0
    uint8_t nq;           //   qubits of the block inside the tile (0..5)
    uint8_t terms;        //   TOP_SP: T = rows per class = entries per row (1, 2, 4), the same for every bank
    uint8_t nsel;         //   0..2 selecting qubits
    uint8_t selbit[2];    // dword 1: their global index bits, most significant bank bit first
    uint8_t ident;        //   bit v: bank v is the identity
    uint8_t pad0;
    uint8_t b[8];         // dwords 2-3: tile-local bits of the block's qubits, ascending (kMaxOpQ used)
    uint32_t meta[kMaxBanks]; // dwords 4-7
    uint32_t pad1[8];
    uint32_t rowoff[kMaxBanks][1 << kMaxOpQ]; // TOP_SP: LDS BYTE offset of the slot position p writes (class order differs per bank)
    uint32_t off[kMaxBanks][1 << kMaxOpQ];    // TOP_SP: LDS BYTE offset of operand j of class c at [c*T + j] (one list per class)
    double re[kMaxBanks][kMaxOpEntries];      // entry j of position p at [p*T + j]
    double im[kMaxBanks][kMaxOpEntries];
};
static_assert(sizeof(TileOp) == 64 + 4 * 128 + 4 * 128 + 8192 && offsetof(TileOp, meta) == 16 && offsetof(TileOp, rowoff) == 64, "TileOp layout is shared with the device");

constexpr int kMaxTileHigh = 10; // high (non-contiguous) qubits per tile
struct TileGeom {
    int32_t tile_bits;          // B: log2 amplitudes per tile
    int32_t low_bits;           // L: tile-local bits [0,L) are global bits [0,L)
    int32_t n_high;             // B - L
    int32_t n;                  // qubits in this state (shard)
    int32_t high[kMaxTileHigh]; // global bit of tile-local bit L+j; any order (the scheduler emits ascending, the engine may reorder)
    int32_t n_scale;            // leading entries of the pass's op list that are TOP_SCALE factors, not blocks
};

// Output permutation of a tile pass that also does the re-layout of an exchange (the pack): the amplitude read at index x is
// written at index  perm(x) = sum_i ((x & seg[i]) >> i)  |  sum_j (bit sel[j] of x) << to[j]  |  konst  of the OUTPUT buffer.
// sel[0..k) are the k <= 3 index bits that leave (ascending); the bits between them (seg[i]: the ones with i selected bits
// below them) close ranks.  One scratch buffer per shard: to[j] = n - k + j (the block index on top), konst = 0.  All shards
// of a cluster in one allocation: to[j] = the shard-id bit that takes the qubit, konst = the rest of the destination.
struct PackMap {
    uint64_t seg[4];
    uint64_t konst;
    int32_t sel[3], to[3];
    int32_t k; // 0: no re-layout
};

struct LaunchCfg {
    hipStream_t stream;
    int grid_cap; // 0 = one workgroup per work tile
};

// All launchers are asynchronous on cfg.stream and return the hipError_t of the launch.  `f32` selects the amplitude
// precision of the state `v` points to: false = fp64 complex (16 B per amplitude), true = fp32 complex (8 B).
hipError_t launch_init(const LaunchCfg &cfg, void *v, bool f32, int n, double amp0);
hipError_t launch_gate1(const LaunchCfg &cfg, void *v, bool f32, int n, int q, const M2 &U);
hipError_t launch_phase(const LaunchCfg &cfg, void *v, bool f32, int n, int q, double lr, double li);
hipError_t launch_diag1_full(const LaunchCfg &cfg, void *v, bool f32, int n, int q, double d0r, double d0i, double d1r, double d1i);
hipError_t launch_cx(const LaunchCfg &cfg, void *v, bool f32, int n, int control, int target);
hipError_t launch_gate2(const LaunchCfg &cfg, void *v, bool f32, int n, int q_hi, int q_lo, const M4 &U);
// from_zero_ket: the state is a not-yet-written basis state; the pass generates it in LDS instead of loading it
// zero_mask: index bits the state is known to be |0> in — tiles with such a bit in their base index are skipped, slots
// with one inside a tile are staged in as zero (their memory may never have been written)
// vout: where the pass writes the state — v itself (in place) or a second buffer of the same size
// pack (fp64, the default tile shape only — launch_tile_can_pack): vout is the output buffer of the re-layout, written at perm(index)
hipError_t launch_tile(const LaunchCfg &cfg, void *v, void *vout, bool f32, const TileGeom &g, const TileOp *d_ops, int n_ops, int threads,
                       bool from_zero_ket, double amp0, bool nomem = false, uint64_t zero_mask = 0, const PackMap *pack = nullptr);
bool launch_tile_can_pack(bool f32, const TileGeom &g, int threads);
// zeroes the amplitudes whose index has a bit of zero_mask set (the part of a state the tile passes have not written yet)
hipError_t launch_zero_outside(const LaunchCfg &cfg, void *v, bool f32, int n, uint64_t zero_mask);
hipError_t launch_norm2(const LaunchCfg &cfg, const void *v, bool f32, int n, double *d_out /* zeroed */);
// d_out[b] = sum of |a|^2 over amplitudes [b << block_bits, (b+1) << block_bits), fixed summation order
hipError_t launch_block_prob(const LaunchCfg &cfg, const void *v, bool f32, int n, int block_bits, double *d_out);
// d_out[w] = sum of |a|^2 over the amplitudes at deposit(w, hi_mask) | deposit(i, lo_mask), all i; 2^popcount(hi_mask) sums
hipError_t launch_block_prob_masked(const LaunchCfg &cfg, const void *v, bool f32, uint64_t hi_mask, uint64_t lo_mask, double *d_out);
// d_out[i] = v[base | deposit(i, lo_mask)], i < 2^popcount(lo_mask)
hipError_t launch_gather_masked(const LaunchCfg &cfg, const void *v, bool f32, uint64_t base, uint64_t lo_mask, void *d_out);
// out[dst] = in[src]: dst = (block << (n-p)) | rest, where block = the p bits of src at positions
// `bits` (ascending) and rest = the remaining n-p bits of src in order.
// `blocks` != NULL (p <= 3): block b goes to blocks[b] (2^(n-p) amplitudes each) instead of out + b * 2^(n-p).
// skip_blocks bit b: block b is not written (nobody will read it; with `blocks`, blocks[b] may then be NULL)
// zero_mask: index bits the source is zero in by definition (a partially written state): such amplitudes are packed as zeros, not loaded
hipError_t launch_pack(const LaunchCfg &cfg, const void *in, void *out, void *const *blocks, bool f32, int n, const int *bits, int p, uint32_t skip_blocks = 0,
                       uint64_t zero_mask = 0);

} // namespace qsim
#endif
