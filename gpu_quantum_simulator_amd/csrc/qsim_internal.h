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

// One fused block inside a cache-blocked pass = one trip of the tile through LDS.  Bit positions are TILE-LOCAL (see TileGeom).
//
//   TOP_PART   sparse 2^k x 2^k block on k = nq in {3..6} tile qubits, `terms` = T (1, 2 or 4) entries per row.  Per bank the
//              block is a direct sum of small dense matrices in a permuted basis (Scheduler / TileBlock::classes), so its rows
//              are laid out CLASS by class: T rows that read the same T operand slots — ONE LDS read per amplitude whatever T is.
//              The 2^k positions of a group are dealt in PARTS of 8 (8 / T whole classes); a lane owns one (group, part), the
//              parts of a group sit in different waves (part-major), so everything a wave needs of the block — operand offsets,
//              row offsets, coefficients — is ONE wave-uniform record (PartRec) read through scalar loads at immediate offsets
//              from one pointer:
//                  x[j]      = slot rec.off[c*T + j]                           (class c of the part, j < T)
//                  y[c*T+i]  = sum_j rec.coef[(c*T+i)*T + j] * x[j]            -> written to slot rec.rowoff[c*T+i]
//              Offsets are ready LDS byte offsets (layout swizzle applied).  A class whose T rows are all identity rows costs
//              nothing: off[c*T] = kSkipClass and rowoff[c*T .. c*T+T) = kSkipClass.  k > 3 needs a workgroup barrier between everybody's reads and the writes unless
//              every part writes exactly the slots it read (flags bit 1).
//              Blocks on fewer than three qubits are PADDED by the engine with tile bits they act on as the identity (same reads,
//              same multiply-adds, same writes per amplitude; one code path, one dispatch).
//              Most fused clusters of Clifford+T-like circuits are permutations times phases or two independent 2x2 blocks, and
//              so are products of neighbouring ones on a few qubits (Scheduler::merge_blocks).
//   TOP_SCALE  no qubit inside the tile: a factor per tile, applied while the tile is staged in (scale[bank]).
//   TOP_G1 / TOP_DIAG1 / TOP_G2   dense 2x2 / diagonal / dense 4x4 in pair and quad form: only in tiles of fewer than 2^3
//              amplitudes (registers of one or two qubits), which have no three bits to pad to; coefficients row-major in
//              rec[bank][0].coef as (re, im); TOP_DIAG1: rec[bank][0].off[0] = 1 when d0 == 1 (only the bit = 1 half moves).
// BANKS.  A block may also depend on up to two qubits OUTSIDE the tile, provided it is block-diagonal in them (a CX
// whose control is outside, any diagonal gate): such a qubit is constant over a tile, so it merely selects which
// 2^k x 2^k sub-block the tile gets.  nsel / selbit name those GLOBAL index bits; the bank index is
// bit(selbit[0]) or 2*bit(selbit[0]) + bit(selbit[1]) of the tile's base index, wave-uniform, and only that bank's
// records are ever loaded.  `ident` bit v: bank v is the identity — the block is skipped on those tiles.
enum : int32_t { TOP_G1 = 1, TOP_G2 = 2, TOP_DIAG1 = 3, TOP_PART = 4, TOP_SCALE = 5 };
constexpr int kMaxBanks = 4;
constexpr int kMaxOpQ = 6;                      // TOP_PART blocks span 3..6 tile qubits
constexpr int kPartRows = 8;                    // positions per part
constexpr int kMaxParts = 1 << (kMaxOpQ - 3);
constexpr uint32_t kSkipClass = 0xFFFFFFFFu;
constexpr uint32_t kOpFlagSkips = 1u, kOpFlagClosed = 2u;
struct PartRec {
    uint32_t off[kPartRows];      // operand j of class c at [c*T + j]
    uint32_t rowoff[kPartRows];   // position p writes slot rowoff[p]
    double coef[kPartRows * 4 * 2]; // entry j of position p: (re, im) at [(p*T + j)*2]; fp32 states: the float pairs (ur, ui), (-ui, ur) in the same 16 bytes
};
struct TileOp {
    // the first 16 bytes are fetched with ONE scalar load (s_load_dwordx4) a block ahead of their use and decoded with scalar
    // bit-field extracts
    uint8_t kind;         // dword 0
    uint8_t nq;           //   qubits of the block inside the tile (TOP_PART: 3..6 after padding)
    uint8_t terms;        //   TOP_PART: T = rows per class = entries per row (1, 2, 4), the same for every bank
    uint8_t nsel;         //   0..2 selecting qubits
    uint8_t selbit[2];    // dword 1: their global index bits, most significant bank bit first
    uint8_t ident;        //   bit v: bank v is the identity
    uint8_t flags;        //   kOpFlagSkips: some class is skipped; kOpFlagClosed: no barrier between reads and writes
    uint8_t b[8];         // dwords 2-3.  TOP_PART: nibble a of b[0..4] = the free tile-local bit that bit a of a lane's group index walks (15: none; engine.cpp to_tile_op picks bank-conflict-free ones), b[7] = log2 T << 1 | skips << 3 | barrier << 4.  Pair / quad forms: b[0], b[1] = the block's tile-local bits, ascending
    uint32_t pad0[4];
    double scale[kMaxBanks][2]; // TOP_SCALE: the factor per bank as (re, im); fp32 states: two floats in the first 8 bytes
    uint32_t pad1[8];
    PartRec rec[kMaxBanks][kMaxParts];
};
static_assert(sizeof(PartRec) == 64 + 512 && offsetof(TileOp, scale) == 32 && offsetof(TileOp, rec) == 128 &&
              sizeof(TileOp) == 128 + kMaxBanks * kMaxParts * sizeof(PartRec), "TileOp layout is shared with the device");

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
    bool warm_only = false; // launch_tile: load the kernel's code object and set its attributes, launch nothing (the cold path does this while the state is being allocated)
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
