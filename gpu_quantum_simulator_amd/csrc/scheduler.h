// scheduler.h — host-side gate scheduler (no device code): turns the queued gate stream into passes.
//
// Reference rows: a10 fusion algebra (quantum_simulator_4x4.cu:148-250: mm2x2, mm4x4,
// tensorProd, cnotTo4x4), a11 pair state machine (quantum_simulator_4x4.cu:327-501; 2x2-only form
// quantum_simulator_preproces.cu:215-269), a12 op list for a single launch
// (quantum_simulator_preproces_constant.cu:288-369).  The algorithms here are this project's own:
// clusters are folded eagerly (no separate pending 2x2 per paired qubit), identity tests are EXACT
// (the reference's 1e-3 tolerance reorders gates), and level 3 groups clusters into
// cache-blocked passes by a greedy scan over the dependency order.
#ifndef QSIM_SCHEDULER_H
#define QSIM_SCHEDULER_H

#include <complex>
#include <cstdint>
#include <functional>
#include <vector>

#include "qsim_internal.h"

namespace qsim {

using cd = std::complex<double>;

enum OpKind : int { OP_G1 = 1, OP_CX = 2, OP_G2 = 3, OP_G3 = 4 };

struct FusedOp {
    int kind = 0;
    int q_hi = -1;  // OP_G1: target; OP_CX: control; OP_G2/OP_G3: highest qubit
    int q_lo = -1;  // OP_CX: target; OP_G2: low qubit; OP_G3: middle qubit
    int q_lo2 = -1; // OP_G3: lowest qubit
    cd m[64];       // row-major 2x2 / 4x4 / 8x8; index bits = (q_hi, q_lo[, q_lo2]), most significant first
    uint32_t gates = 0;
    // Set when the op is emitted into a tile pass: those of its qubits that are NOT in the pass's tile.  The op is
    // block-diagonal in each of them (selector_mask), so per tile it reduces to the sub-block picked by those bits of
    // the tile's base index: a smaller block on the in-tile qubits, or a plain factor when none is left.
    uint64_t sel_mask = 0;

    int nq() const { return kind == OP_G1 ? 1 : kind == OP_G3 ? 3 : 2; }
    int dim() const { return 1 << nq(); }
    uint64_t qmask() const {
        uint64_t m = 1ULL << q_hi;
        if (kind != OP_G1) m |= 1ULL << q_lo;
        if (kind == OP_G3) m |= 1ULL << q_lo2;
        return m;
    }
    int max_row_nnz() const; // exact-zero structure
    bool is_diag() const;
    bool is_identity() const;
    uint64_t selector_mask() const; // qubits the matrix is block-diagonal in (exact zeros): it never mixes their 0 and 1 halves
    bool is_scalar_in_tile() const { return sel_mask != 0 && sel_mask == qmask(); }
};

struct Pass {
    int kclass = 0;           // QSIM_K_*
    std::vector<FusedOp> ops; // one op unless kclass == QSIM_K_TILE
    TileGeom geom{};          // QSIM_K_TILE only
    double bytes = 0;         // algorithmic bytes this pass must move
    bool diag_full = false;   // QSIM_K_PHASE executed over every amplitude (d0 != 1 or q < 2)
};

struct SchedConfig {
    int n = 0;
    int fuse = 3;
    int tile_bits = 12;
    int tile_low_bits = 3;
    int tile_max_ops = 32;
    int window = 512;  // clusters scanned ahead when grouping a pass
    int merge = 1;     // level 3: merge neighbouring blocks of a pass into sparse <=3-qubit blocks
    int rollout = 8;   // level 3: candidates tried (each by greedily finishing the pass) when a new qubit must be admitted; 0 = off
    int pad_from = 10; // first bit tried when unused tile slots are filled (below tile_low_bits: tile_low_bits)
    int selectors = 0; // level 3: a block may run in a pass whose tile lacks qubits it is block-diagonal in
};

class Scheduler {
  public:
    explicit Scheduler(const SchedConfig &cfg);
    void add_1q(const cd U[4], int q);
    void add_cx(int control, int target);
    void add_2q(const cd U[16], int q_hi, int q_lo);
    // Closes every open cluster and produces the passes for everything added so far, in launch order.  The sink
    // form hands each pass over as soon as it is complete, so the caller can launch it while later passes are
    // still being scheduled.
    using PassSink = std::function<void(Pass &&)>;
    void finish(const PassSink &sink);
    void finish(std::vector<Pass> &out);
    uint64_t gates_seen() const { return gates_; }

    // fusion algebra, exposed for tests
    static void mul2(const cd a[4], const cd b[4], cd out[4]);     // out = a*b
    static void mul4(const cd a[16], const cd b[16], cd out[16]);  // out = a*b
    static void kron(const cd hi[4], const cd lo[4], cd out[16]);  // out = hi (x) lo
    static void cx4(bool control_is_hi, cd out[16]);

  private:
    SchedConfig cfg_;
    uint64_t gates_ = 0;
    std::vector<FusedOp> pool_;   // open clusters
    std::vector<int> open_;       // per qubit: index into pool_, or -1
    std::vector<FusedOp> closed_; // fused ops in a valid execution order

    void close(int idx);
    void fold_2q(const cd U[16], int q_hi, int q_lo, uint32_t gates);
    void build_passes(const PassSink &sink);
    void single_op_pass(const FusedOp &op, const PassSink &sink) const;
    void tile_pass(const std::vector<FusedOp> &ops, uint64_t hset, const PassSink &sink) const;
    void merge_sparse(std::vector<FusedOp> &ops) const;
};

} // namespace qsim
#endif
