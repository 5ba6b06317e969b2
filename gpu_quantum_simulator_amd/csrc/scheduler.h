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

enum OpKind : int { OP_G1 = 1, OP_CX = 2, OP_G2 = 3 };

struct FusedOp {
    int kind = 0;
    int q_hi = -1;  // OP_G1: target; OP_CX: control; OP_G2: high qubit
    int q_lo = -1;  // OP_CX: target; OP_G2: low qubit
    cd m[16];       // row-major 2x2 / 4x4; index bits = (q_hi[, q_lo]), most significant first
    uint32_t gates = 0;

    int nq() const { return kind == OP_G1 ? 1 : 2; }
    int dim() const { return 1 << nq(); }
    uint64_t qmask() const {
        uint64_t m = 1ULL << q_hi;
        if (kind != OP_G1) m |= 1ULL << q_lo;
        return m;
    }
    int max_row_nnz() const; // exact-zero structure
    bool is_diag() const;
    bool is_identity() const;
    uint64_t selector_mask() const; // qubits the matrix is block-diagonal in (exact zeros): it never mixes their 0 and 1 halves
};

// One block of a tile pass = one trip through LDS.  `q` are its qubits inside the tile; `s` are qubits OUTSIDE the tile
// that the block is block-diagonal in: constant over a tile, they only select which bank (sub-block on `q`) applies.
// nq == 0: a tile-uniform factor (entry (0,0) of each bank).  Stored by rows, at most kMaxRowNnz entries each: that is
// what one LDS trip of k_tile can evaluate (a dense 4x4 on two qubits is the fullest case), and it keeps the products
// of merge_blocks cheap on up to kMaxBlockQ qubits.
constexpr int kMaxBlockQ = 6, kMaxRowNnz = 4;
struct TileBlock {
    int nq = 0;
    int q[kMaxBlockQ] = {-1, -1, -1, -1, -1, -1}; // descending: q[0] is the most significant bit of the row/column index
    int ns = 0;
    int s[2] = {-1, -1};                      // descending: s[0] is the most significant bit of the bank index
    struct Row {
        int n = 0;
        uint8_t col[kMaxRowNnz];
        cd val[kMaxRowNnz];
    };
    std::vector<Row> store; // banks() x dim() rows, bank-major; sized by shape()
    uint32_t gates = 0;

    void shape(int nq_, int ns_) { nq = nq_; ns = ns_; store.assign((size_t)1 << (nq_ + ns_), Row()); }
    Row &row(int v, int r) { return store[((size_t)v << nq) + r]; }
    const Row &row(int v, int r) const { return store[((size_t)v << nq) + r]; }

    int dim() const { return 1 << nq; }
    int banks() const { return 1 << ns; }
    uint64_t in_mask() const { uint64_t m = 0; for (int a = 0; a < nq; a++) m |= 1ULL << q[a]; return m; }
    uint64_t sel_mask() const { uint64_t m = 0; for (int a = 0; a < ns; a++) m |= 1ULL << s[a]; return m; }
    cd at(int v, int r, int c) const {
        const Row &rw = row(v, r);
        for (int j = 0; j < rw.n; j++)
            if (rw.col[j] == c) return rw.val[j];
        return cd(0, 0);
    }
    int max_row_nnz() const;    // over all banks
    bool bank_is_identity(int v) const;
    bool is_identity() const;   // every bank
    // the block as ONE matrix on the qubits (s..., q...), most significant first: block-diagonal in the bank index.
    // `out` holds (1 << (ns + nq))^2 entries.
    void full_matrix(cd *out) const;
    // Row classes (what one LDS trip of k_tile evaluates with ONE read per amplitude).  Per bank the nonzero pattern of a
    // block splits into connected components of rows and columns; in every block built from 1- and 2-qubit gates under
    // the kMaxRowNnz limit a component has as many columns as rows, at most kMaxRowNnz of each — the block is a direct
    // sum of small dense matrices in a permuted basis.  classes() packs the components of each bank into classes of
    // exactly T rows sharing exactly T columns (T = 1, 2 or 4, the same for all banks): rows[v] lists the rows class by
    // class, cols[v] the columns of class c at [c*T, (c+1)*T).  Identity rows are packed together so that whole classes
    // can be skipped.  false: some component is larger than kMaxRowNnz or the packing leaves a gap (such a block is
    // never built: merge_blocks asks first).
    bool classes(int &T, std::vector<std::vector<int>> &rows, std::vector<std::vector<int>> &cols) const;
    bool classes_feasible() const; // the same answer as classes() without building the layout (no heap: merge_blocks asks often)
};

struct Pass {
    int kclass = 0;           // QSIM_K_*
    std::vector<FusedOp> ops;      // the one op of a single-op pass; empty for QSIM_K_TILE
    std::vector<TileBlock> blocks; // QSIM_K_TILE: geom.n_scale tile-uniform factors first, then the blocks in order
    TileGeom geom{};               // QSIM_K_TILE only
    uint32_t gates() const {
        uint32_t g = 0;
        for (const FusedOp &op : ops) g += op.gates;
        for (const TileBlock &b : blocks) g += b.gates;
        return g;
    }
    std::vector<uint32_t> src; // SchedConfig::track: the gates folded into this pass, by their order of arrival at the scheduler (0, 1, ...)
    double bytes = 0;         // algorithmic bytes this pass must move when it sweeps the whole register
    double visited = 1.0;     // ... times this: the fraction of the register inside the state's support after the pass (tile passes of a run that starts from a reset, SchedConfig::initial_support)
    bool diag_full = false;   // QSIM_K_PHASE executed over every amplitude (d0 != 1 or q < 2)
};

// What a pass costs, in bytes moved at the rate of a pass that is bound by its memory traffic.  A tile pass is: up to four
// merged blocks hide behind the HBM time of the sweep, every further one adds 6.0 % of it (fp64; fp32 moves half the bytes per
// amplitude under the same blocks: 8.9 % from the fourth on).  Fitted on the tile passes of seven circuits under twelve scheduler
// settings at n = 30 with the kernel of the end of round 4 (tools/pass_model_data.py; profiles/r04/pass_model_n30_last_tree.csv, 1347 passes):
// ms = visited x (7.16 + 0.43 x max(0, blocks - 4)), rms error 0.43 ms (before the lanes walked conflict-free bits: 0.49 per block on the
// same settings; round 4's first fit 0.58, round 3's kernel 0.77); fp32 states (pass_model_n30_f32_last_tree.csv): 3.67 + 0.33 x
// max(0, blocks - 3), rms 0.24 ms (0.57 per block before the fp32 swizzle).  The planning steps (engine: which schedule; shard planner:
// which segmentation) rank by the sum of this — fewer sweeps are worth more than leaner ones, but not at any number of blocks.
inline double pass_time_cost(const Pass &p, bool f32) {
    if (p.kclass != QSIM_K_TILE) return p.bytes;
    const int nb = (int)p.blocks.size() - p.geom.n_scale;
    const int over = f32 ? (nb > 3 ? nb - 3 : 0) : (nb > 4 ? nb - 4 : 0);
    return p.bytes * p.visited * (1.0 + (f32 ? 0.089 : 0.060) * over);
}

struct SchedConfig {
    int n = 0;
    int fuse = 3;
    int tile_bits = 12;
    int tile_low_bits = 3;
    int tile_max_ops = 32;
    int tail_max_ops = 0;  // > tile_max_ops: the cap of a pass that can take ALL remaining clusters (no straggler pass for a handful of gates)
    int window = 512;  // clusters scanned ahead when grouping a pass
    int merge = 1;     // level 3: merge neighbouring blocks of a pass into sparse blocks of up to merge_qubits tile qubits
    int merge_qubits = 6; // round 4: 6 (64 rows = 8 parts of 8, one wave each in a 2^12 tile): 9 % fewer LDS trips than 5 at the same multiply-adds
    int rollout = 8;   // level 3: candidates tried (each by greedily finishing the pass) when a new qubit must be admitted; 0 = off
    int pad_from = 10; // first bit tried when unused tile slots are filled (below tile_low_bits: tile_low_bits)
    int lookahead = 0; // level 3: further passes (built greedily) whose reach is added to a candidate's score
    int local_iters = 0; // level 3: rounds of swap-one-qubit local search on each pass's qubit set
    int objective = 0; // what a pass maximises: 0 blocks, 1 source gates
    int selectors = 1; // level 3: a block may run in a pass whose tile lacks qubits it is block-diagonal in
    // Qubits that may already be 1 somewhere in the state when the first pass runs (all ones: unknown / dense).  After a
    // reset it is 0: a tile pass then only visits 2^(|support after it| - n) of the register (engine: qsim_state::support),
    // so while the support is small the scheduler (a) never takes a qubit outside it into a tile without need and (b)
    // first tries a pass that stays inside it altogether, keeping it when it absorbs more clusters than its share of a
    // full sweep is worth (cheap_margin x tile_max_ops x that share).
    uint64_t initial_support = ~0ULL;
    double cheap_margin = 1.0;
    // 1: a cluster may overtake an earlier pending one when both are block-diagonal in every qubit they share (they commute:
    // controls of CXs, diagonal gates); 0: any shared qubit orders them (round 1).  More freedom is better on average (ten
    // seeded 1000-gate circuits at n = 30: 892 -> 835 ms in total) but the greedy packing does not use it well on every
    // circuit, so the planning step (qsim_tune_circuit) schedules both ways and keeps the cheaper one for that circuit.
    int commute = 1;
    uint64_t seed = 0; // != 0: ties between equally good candidates (which cluster to admit next, which qubit to swap) are broken pseudo-randomly
                       // instead of in index order — another valid schedule of the same circuit per seed, for the planning step to choose among
    int track = 0; // 1: every pass lists the gates it absorbed (Pass::src) — the shard planner asks which gates a segment's last pass holds
};

// The configuration qsim_flush schedules with for a state of n qubits (options as given by the caller; the search settings
// follow the size of the state, see the comment in scheduler.cpp).
SchedConfig engine_sched_config(int n, int fuse, int tile_bits, int tile_low_bits, int tile_max_ops, int pad_from = 10, bool f32 = false,
                                uint64_t initial_support = 0);

// The QSIM_SCHED_* environment variables override the search parameters for experiments (tools/).
// They are not part of the API and change the pass count, never the result — but they DO shape the schedule, so the
// engine folds this record into the identity of a cached plan (a plan built under one setting is never replayed under
// another).  `set` has one bit per variable that is present.
struct SchedEnv {
    uint32_t set = 0;
    int lookahead = 0, rollout = 0, window = 0, local_iters = 0, objective = 0, merge = 0, merge_qubits = 0, cap = 0, seed = 0;
    double cheap_margin = 0;
    bool operator==(const SchedEnv &o) const {
        return set == o.set && lookahead == o.lookahead && rollout == o.rollout && window == o.window && local_iters == o.local_iters &&
               objective == o.objective && merge == o.merge && merge_qubits == o.merge_qubits && cap == o.cap && seed == o.seed && cheap_margin == o.cheap_margin;
    }
};
SchedEnv read_sched_env();                      // the environment as it is now
void apply_sched_env(const SchedEnv &e, SchedConfig &cfg);

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
    std::vector<std::vector<uint32_t>> pool_src_, closed_src_; // SchedConfig::track: the gates behind pool_[i] / closed_[i]
    mutable std::vector<uint32_t> cur_src_;                     // ... and behind the pass being emitted

    void close(int idx);
    void fold_2q(const cd U[16], int q_hi, int q_lo, uint32_t gates, uint32_t gate_index);
    void build_passes(const PassSink &sink);
    void single_op_pass(const FusedOp &op, const PassSink &sink) const;
    // prefer: index bits free tile slots are filled from first (the state's support while it is partial); returns the tile's qubit mask
    uint64_t tile_pass(const std::vector<FusedOp> &ops, uint64_t hset, const PassSink &sink, uint64_t prefer = ~0ULL) const;
    void merge_blocks(std::vector<TileBlock> &blocks) const;
};

} // namespace qsim
#endif
