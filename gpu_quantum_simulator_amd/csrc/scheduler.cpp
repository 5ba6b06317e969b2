// scheduler.cpp — see scheduler.h.  Pure host code; compiled into libqsim.so and exercised on the CPU by
// tests/test_scheduler_cpu.py through qsim_schedule_circuit().
#include "scheduler.h"

#include <algorithm>
#include <cstdlib>
#include <cstring>

namespace qsim {

static const cd kI2[4] = {cd(1, 0), cd(0, 0), cd(0, 0), cd(1, 0)};

static inline bool is_zero(const cd &z) { return z.real() == 0.0 && z.imag() == 0.0; }
static inline bool is_one(const cd &z) { return z.real() == 1.0 && z.imag() == 0.0; }

bool FusedOp::is_diag() const {
    if (kind == OP_CX) return false;
    const int d = dim();
    for (int r = 0; r < d; r++)
        for (int c = 0; c < d; c++)
            if (r != c && !is_zero(m[d * r + c])) return false;
    return true;
}

int FusedOp::max_row_nnz() const {
    if (kind == OP_CX) return 1;
    const int d = dim();
    int best = 0;
    for (int r = 0; r < d; r++) {
        int cnt = 0;
        for (int c = 0; c < d; c++) cnt += !is_zero(m[d * r + c]);
        best = std::max(best, cnt);
    }
    return best;
}

uint64_t FusedOp::selector_mask() const {
    if (kind == OP_CX) return 1ULL << q_hi; // the control
    const int k = nq(), d = dim();
    const int qs[2] = {q_hi, q_lo};
    uint64_t out = 0;
    for (int a = 0; a < k; a++) {
        const int bit = 1 << (k - 1 - a); // position of qs[a] in the row/column index
        bool ok = true;
        for (int r = 0; r < d && ok; r++)
            for (int c = 0; c < d; c++)
                if (((r ^ c) & bit) && !is_zero(m[d * r + c])) { ok = false; break; }
        if (ok) out |= 1ULL << qs[a];
    }
    return out;
}

// EXACT identity only: the reference's isIdentity tolerance of 1e-3 (quantum_simulator_4x4.cu:247-250)
// silently drops and reorders small rotations.
bool FusedOp::is_identity() const {
    if (!is_diag()) return false;
    const int d = dim();
    for (int r = 0; r < d; r++)
        if (!is_one(m[(d + 1) * r])) return false;
    return true;
}

// ---- fusion algebra (own formulation of quantum_simulator_4x4.cu:148-233) -----------------------------
void Scheduler::mul2(const cd a[4], const cd b[4], cd out[4]) {
    cd t[4];
    for (int r = 0; r < 2; r++)
        for (int c = 0; c < 2; c++) t[2 * r + c] = a[2 * r] * b[c] + a[2 * r + 1] * b[2 + c];
    std::copy(t, t + 4, out);
}

void Scheduler::mul4(const cd a[16], const cd b[16], cd out[16]) {
    cd t[16];
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) {
            cd s(0, 0);
            for (int k = 0; k < 4; k++) s += a[4 * r + k] * b[4 * k + c];
            t[4 * r + c] = s;
        }
    std::copy(t, t + 16, out);
}

// (hi (x) lo)[(i1 i2), (j1 j2)] = hi[i1][j1] * lo[i2][j2]
void Scheduler::kron(const cd hi[4], const cd lo[4], cd out[16]) {
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[4 * r + c] = hi[2 * (r >> 1) + (c >> 1)] * lo[2 * (r & 1) + (c & 1)];
}

// CX as a permutation of the basis |hi lo>: control on the high bit exchanges |10> and |11>, control on
// the low bit exchanges |01> and |11>.
void Scheduler::cx4(bool control_is_hi, cd out[16]) {
    const int perm_hi[4] = {0, 1, 3, 2}, perm_lo[4] = {0, 3, 2, 1};
    const int *p = control_is_hi ? perm_hi : perm_lo;
    for (int r = 0; r < 4; r++)
        for (int c = 0; c < 4; c++) out[4 * r + c] = cd(p[r] == c ? 1.0 : 0.0, 0.0);
}

// std::complex multiplication may go through the Annex-G slow path; these matrices are tiny and only
// built on the host, so that is irrelevant.  Zeros stay exact: every term of an off-diagonal entry of a
// product of diagonal matrices has an exact-zero factor.

// The QSIM_SCHED_* variables override the search parameters for experiments (tools/); they are not
// part of the API and change the pass count, never the result (scheduler.h SchedEnv).
SchedEnv read_sched_env() {
    SchedEnv e;
    int bit = 0;
    auto geti = [&](const char *name, int &dst) { if (const char *v = getenv(name)) { dst = atoi(v); e.set |= 1u << bit; } bit++; };
    geti("QSIM_SCHED_LOOKAHEAD", e.lookahead);
    geti("QSIM_SCHED_ROLLOUT", e.rollout);
    geti("QSIM_SCHED_WINDOW", e.window);
    geti("QSIM_SCHED_LOCAL", e.local_iters);
    geti("QSIM_SCHED_OBJ", e.objective);
    geti("QSIM_SCHED_MERGE", e.merge);
    geti("QSIM_SCHED_MERGEQ", e.merge_qubits);
    geti("QSIM_SCHED_CAP", e.cap);
    if (const char *v = getenv("QSIM_SCHED_CHEAP")) { e.cheap_margin = atof(v); e.set |= 1u << bit; }
    bit++;
    if (getenv("QSIM_SCHED_NOCOMMUTE")) e.set |= 1u << bit;
    bit++;
    geti("QSIM_SCHED_SEED", e.seed);
    return e;
}

void apply_sched_env(const SchedEnv &e, SchedConfig &cfg) {
    int bit = 0;
    auto on = [&]() { return (e.set >> bit++) & 1u; };
    if (on()) cfg.lookahead = e.lookahead;
    if (on()) cfg.rollout = e.rollout;
    if (on()) cfg.window = e.window;
    if (on()) cfg.local_iters = e.local_iters;
    if (on()) cfg.objective = e.objective;
    if (on()) cfg.merge = e.merge;
    if (on()) cfg.merge_qubits = e.merge_qubits;
    if (on()) { cfg.tile_max_ops = e.cap; cfg.tail_max_ops = 0; }
    if (on()) cfg.cheap_margin = e.cheap_margin;
    if (on()) cfg.commute = 0;
    if (on()) cfg.seed = (uint64_t)e.seed;
}

Scheduler::Scheduler(const SchedConfig &cfg) : cfg_(cfg), open_(cfg.n > 0 ? cfg.n : 0, -1) { apply_sched_env(read_sched_env(), cfg_); }

SchedConfig engine_sched_config(int n, int fuse, int tile_bits, int tile_low_bits, int tile_max_ops, int pad_from, bool f32, uint64_t initial_support) {
    SchedConfig c;
    c.pad_from = pad_from;
    c.initial_support = initial_support; // 0: the run starts from a reset (what the planning entry points assume)
    c.n = n; c.fuse = fuse; c.tile_bits = tile_bits; c.tile_low_bits = tile_low_bits; c.tile_max_ops = tile_max_ops;
    // The pass-set local search (SchedConfig::local_iters with one pass of lookahead) cost ~1.5 ms of host time per pass when the
    // thresholds below were measured (~0.7 ms since the scans of build_passes became incremental).  Passes are launched as they are produced, so the search is free once a pass runs longer than that on the
    // GPU: from 4 GiB of state (n = 28 fp64: 1.9 ms per pass).  Since the row-class form of the sparse blocks (one LDS
    // read per amplitude) most passes are bound by their memory time again, so one pass less is ~7 ms less at n = 30
    // (round 1: the fuller passes were LDS-bound and the total did not move).  With the search on, a pass is capped at
    // 24 clusters (5-6 merged blocks): ~1.6 + 0.8 ms per block then stays under the pass's ~6.8 ms of memory time.
    // Twelve seeded 1000-gate circuits at n = 30: 204 passes without the search, 188 with it (191 / 193 with two /
    // three passes of lookahead, which also cost more host time, so one it is).  n = 30 bench circuit: 16 passes /
    // 119.7 ms without, 15 / 115.5 ms with; n = 28: 31.1 -> 29.8 ms; n = 32: 471 -> 451 ms; n = 26 would LOSE (8.5 -> 9.9 ms,
    // the host becomes the bottleneck), hence the threshold.  QSIM_SCHED_LOCAL / QSIM_SCHED_LOOKAHEAD override.
    const int size_class = n - (f32 ? 1 : 0); // log2 of the state size in 16-byte units
    if (fuse >= 3 && size_class >= 28) {
        c.local_iters = 3;
        c.lookahead = 1;
        // Round 3 re-measured the cap with the cheaper block phase of round 2 (tools/cap_sweep.py, five seeded circuits at n = 30, no
        // planning step): 24 clusters 404 ms in total, 28 373 ms, 32 385 ms, 40 374 ms — and no single value is best for every
        // circuit (per-circuit minima add up to 363 ms), so 28 is the default and the planning step tries 24 / 32 / 40 as well.
        if (tile_max_ops == 32) { c.tile_max_ops = 28; c.tail_max_ops = 32; } // 32 = the option's default, i.e. not chosen by the caller
    }
    return c;
}

void Scheduler::close(int idx) {
    if (idx < 0) return;
    FusedOp &op = pool_[idx];
    open_[op.q_hi] = -1;
    if (op.kind == OP_G2) open_[op.q_lo] = -1;
    if (!op.is_identity()) {
        closed_.push_back(op);
        if (cfg_.track) closed_src_.push_back(std::move(pool_src_[(size_t)idx]));
    }
    op.kind = 0;
}

void Scheduler::add_1q(const cd U[4], int q) {
    const uint32_t g = (uint32_t)gates_++;
    if (cfg_.fuse == 0) {
        FusedOp op;
        op.kind = OP_G1; op.q_hi = q; op.gates = 1;
        std::copy(U, U + 4, op.m);
        closed_.push_back(op);
        if (cfg_.track) closed_src_.push_back({g});
        return;
    }
    const int idx = open_[q];
    if (idx < 0) {
        FusedOp op;
        op.kind = OP_G1; op.q_hi = q; op.gates = 1;
        std::copy(U, U + 4, op.m);
        pool_.push_back(op);
        if (cfg_.track) pool_src_.push_back({g});
        open_[q] = (int)pool_.size() - 1;
        return;
    }
    // Level 3 keeps a pair cluster block-diagonal in a qubit for as long as it can: such a cluster can run in passes
    // whose tile does not contain that qubit.  A gate that would mix the qubit's halves starts a new cluster instead
    // (inside one pass the two are merged again by merge_blocks).
    if (cfg_.fuse >= 3 && cfg_.selectors && pool_[idx].kind == OP_G2 && (!is_zero(U[1]) || !is_zero(U[2])) &&
        (pool_[idx].selector_mask() >> q & 1ULL)) {
        close(idx);
        FusedOp op;
        op.kind = OP_G1; op.q_hi = q; op.gates = 1;
        std::copy(U, U + 4, op.m);
        pool_.push_back(op);
        if (cfg_.track) pool_src_.push_back({g});
        open_[q] = (int)pool_.size() - 1;
        return;
    }
    FusedOp &c = pool_[idx];
    c.gates++;
    if (cfg_.track) pool_src_[(size_t)idx].push_back(g);
    if (c.kind == OP_G1) {
        mul2(U, c.m, c.m); // later gate multiplies from the left
    } else {
        cd e[16];
        if (q == c.q_hi) kron(U, kI2, e);
        else kron(kI2, U, e);
        mul4(e, c.m, c.m);
    }
}

void Scheduler::add_cx(int control, int target) {
    const uint32_t g = (uint32_t)gates_++;
    if (control == target) return; // quantum_simulator.c:99: a silent no-op
    if (cfg_.fuse <= 1) {
        if (cfg_.fuse == 1) { close(open_[control]); close(open_[target]); }
        FusedOp op;
        op.kind = OP_CX; op.q_hi = control; op.q_lo = target; op.gates = 1;
        closed_.push_back(op);
        if (cfg_.track) closed_src_.push_back({g});
        return;
    }
    cd m[16];
    cx4(control > target, m);
    fold_2q(m, std::max(control, target), std::min(control, target), 1, g);
}

void Scheduler::add_2q(const cd U[16], int q_hi, int q_lo) {
    const uint32_t g = (uint32_t)gates_++;
    if (cfg_.fuse <= 1) {
        if (cfg_.fuse == 1) { close(open_[q_hi]); close(open_[q_lo]); }
        FusedOp op;
        op.kind = OP_G2; op.q_hi = q_hi; op.q_lo = q_lo; op.gates = 1;
        std::copy(U, U + 16, op.m);
        closed_.push_back(op);
        if (cfg_.track) closed_src_.push_back({g});
        return;
    }
    fold_2q(U, q_hi, q_lo, 1, g);
}

void Scheduler::fold_2q(const cd U[16], int q_hi, int q_lo, uint32_t gates, uint32_t g) {
    int ia = open_[q_hi], ib = open_[q_lo];
    if (ia >= 0 && ia == ib) { // the pair is already one cluster: keep folding
        FusedOp &c = pool_[ia];
        mul4(U, c.m, c.m);
        c.gates += gates;
        if (cfg_.track) pool_src_[(size_t)ia].push_back(g);
        return;
    }
    // a cluster shared with a third qubit has to run first
    if (ia >= 0 && pool_[ia].kind == OP_G2) { close(ia); ia = -1; }
    if (ib >= 0 && pool_[ib].kind == OP_G2) { close(ib); ib = -1; }
    if (cfg_.fuse >= 3 && cfg_.selectors) {
        // same idea when the pair cluster is created: a pending 1-qubit product that is not diagonal would destroy
        // the block-diagonal structure U has in that qubit (e.g. the control of a CX) — let it run on its own
        FusedOp probe;
        probe.kind = OP_G2; probe.q_hi = q_hi; probe.q_lo = q_lo;
        std::copy(U, U + 16, probe.m);
        const uint64_t sel = probe.selector_mask();
        if (ia >= 0 && (sel >> q_hi & 1ULL) && !pool_[ia].is_diag()) { close(ia); ia = -1; }
        if (ib >= 0 && (sel >> q_lo & 1ULL) && !pool_[ib].is_diag()) { close(ib); ib = -1; }
    }
    FusedOp op;
    op.kind = OP_G2; op.q_hi = q_hi; op.q_lo = q_lo; op.gates = gates;
    const cd *a = kI2, *b = kI2;
    if (ia >= 0) { a = pool_[ia].m; op.gates += pool_[ia].gates; }
    if (ib >= 0) { b = pool_[ib].m; op.gates += pool_[ib].gates; }
    cd k[16];
    kron(a, b, k);
    mul4(U, k, op.m);
    if (ia >= 0) pool_[ia].kind = 0;
    if (ib >= 0) pool_[ib].kind = 0;
    pool_.push_back(op);
    if (cfg_.track) {
        std::vector<uint32_t> src;
        if (ia >= 0) src = std::move(pool_src_[(size_t)ia]);
        if (ib >= 0) src.insert(src.end(), pool_src_[(size_t)ib].begin(), pool_src_[(size_t)ib].end());
        src.push_back(g);
        pool_src_.push_back(std::move(src));
    }
    open_[q_hi] = open_[q_lo] = (int)pool_.size() - 1;
}

void Scheduler::finish(const PassSink &sink) {
    for (int q = 0; q < cfg_.n; q++) close(open_[q]);
    pool_.clear();
    pool_src_.clear();
    build_passes(sink);
    closed_.clear();
    closed_src_.clear();
}

void Scheduler::finish(std::vector<Pass> &out) {
    finish([&out](Pass &&p) { out.push_back(std::move(p)); });
}

// ---- tile blocks ----------------------------------------------------------------------------------------
int TileBlock::max_row_nnz() const {
    int best = 0;
    for (int v = 0; v < banks(); v++)
        for (int r = 0; r < dim(); r++) best = std::max(best, row(v, r).n);
    return best;
}

bool TileBlock::bank_is_identity(int v) const {
    for (int r = 0; r < dim(); r++) {
        const Row &rw = row(v, r);
        if (rw.n != 1 || rw.col[0] != r || !is_one(rw.val[0])) return false;
    }
    return true;
}

bool TileBlock::is_identity() const {
    for (int v = 0; v < banks(); v++)
        if (!bank_is_identity(v)) return false;
    return true;
}

void TileBlock::full_matrix(cd *out) const {
    const int d = dim(), D = d << ns;
    std::fill(out, out + (size_t)D * D, cd(0, 0));
    for (int v = 0; v < banks(); v++)
        for (int r = 0; r < d; r++)
            for (int j = 0; j < row(v, r).n; j++) out[(size_t)(v * d + r) * D + (v * d + row(v, r).col[j])] = row(v, r).val[j];
}

bool TileBlock::classes(int &T, std::vector<std::vector<int>> &rows, std::vector<std::vector<int>> &cols) const {
    const int D = dim(), NB = banks();
    struct Comp { std::vector<int> r, c; bool ident; };
    std::vector<std::vector<Comp>> all((size_t)NB);
    int maxsz = 1;
    for (int v = 0; v < NB; v++) {
        std::vector<int> parent((size_t)2 * D); // rows 0..D-1, columns D..2D-1
        for (int i = 0; i < 2 * D; i++) parent[i] = i;
        auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (int r = 0; r < D; r++)
            for (int j = 0; j < row(v, r).n; j++) {
                const int a = find(r), b = find(D + row(v, r).col[j]);
                if (a != b) parent[a] = b;
            }
        std::vector<int> root_of((size_t)2 * D, -1);
        std::vector<Comp> &cs = all[v];
        for (int r = 0; r < D; r++) {
            const int f = find(r);
            if (root_of[f] < 0) { root_of[f] = (int)cs.size(); cs.push_back(Comp{{}, {}, true}); }
            Comp &cp = cs[root_of[f]];
            cp.r.push_back(r);
            const Row &rw = row(v, r);
            if (!(rw.n == 1 && rw.col[0] == r && is_one(rw.val[0]))) cp.ident = false;
        }
        for (int c = 0; c < D; c++) {
            const int f = find(D + c);
            if (root_of[f] < 0) return false; // a column nobody reads: not a unitary block
            cs[root_of[f]].c.push_back(c);
        }
        for (const Comp &cp : cs) {
            if (cp.r.size() != cp.c.size() || (int)cp.r.size() > kMaxRowNnz) return false;
            maxsz = std::max(maxsz, (int)cp.r.size());
        }
    }
    T = maxsz <= 1 ? 1 : maxsz <= 2 ? 2 : 4;
    if (D % T) return false;
    rows.assign((size_t)NB, {});
    cols.assign((size_t)NB, {});
    for (int v = 0; v < NB; v++) {
        std::vector<Comp> &cs = all[v];
        // first-fit decreasing, identity components last so that they share classes with one another
        std::stable_sort(cs.begin(), cs.end(), [](const Comp &a, const Comp &b) {
            if (a.ident != b.ident) return !a.ident;
            return a.r.size() > b.r.size();
        });
        std::vector<std::vector<int>> br, bc; // bins
        for (const Comp &cp : cs) {
            size_t k = 0;
            while (k < br.size() && br[k].size() + cp.r.size() > (size_t)T) k++;
            if (k == br.size()) { br.emplace_back(); bc.emplace_back(); }
            br[k].insert(br[k].end(), cp.r.begin(), cp.r.end());
            bc[k].insert(bc[k].end(), cp.c.begin(), cp.c.end());
        }
        for (size_t k = 0; k < br.size(); k++) {
            if (br[k].size() != (size_t)T) return false; // a gap: cannot be laid out as whole classes
            rows[v].insert(rows[v].end(), br[k].begin(), br[k].end());
            cols[v].insert(cols[v].end(), bc[k].begin(), bc[k].end());
        }
    }
    return true;
}

bool TileBlock::classes_feasible() const {
    const int D = dim(), NB = banks();
    if (D > 64) return false;
    int maxsz = 1;
    int cnt[kMaxBanks][kMaxRowNnz + 1]; // components by size, per bank
    for (int v = 0; v < NB; v++) {
        unsigned char parent[128], nrow[128], ncol[128];
        for (int i = 0; i < 2 * D; i++) { parent[i] = (unsigned char)i; nrow[i] = ncol[i] = 0; }
        auto find = [&](int x) { while (parent[x] != x) { parent[x] = parent[parent[x]]; x = parent[x]; } return x; };
        for (int r = 0; r < D; r++)
            for (int j = 0; j < row(v, r).n; j++) {
                const int a = find(r), b = find(D + row(v, r).col[j]);
                if (a != b) parent[a] = (unsigned char)b;
            }
        for (int r = 0; r < D; r++) nrow[find(r)]++;
        for (int c = 0; c < D; c++) ncol[find(D + c)]++;
        for (int s = 0; s <= kMaxRowNnz; s++) cnt[v][s] = 0;
        for (int i = 0; i < 2 * D; i++) {
            if (parent[i] != i || (nrow[i] == 0 && ncol[i] == 0)) continue;
            if (nrow[i] != ncol[i] || nrow[i] > kMaxRowNnz) return false;
            cnt[v][nrow[i]]++;
            maxsz = std::max(maxsz, (int)nrow[i]);
        }
    }
    const int T = maxsz <= 1 ? 1 : maxsz <= 2 ? 2 : 4;
    if (D % T) return false;
    if (T < 4) return true; // sizes 1 and 2 always fill bins of 2 (D is even)
    for (int v = 0; v < NB; v++) { // bins of 4: every 3 takes a 1, an odd 2 takes two 1s, the rest fills up by itself
        const int ones = cnt[v][1] - cnt[v][3];
        if (ones < 0 || ones < 2 * (cnt[v][2] & 1)) return false;
    }
    return true;
}

// Splits a fused op (1 or 2 qubits at level 3) by the tile: qubits in `inside` stay matrix indices, the others become
// bank selectors.  The op must be block-diagonal in every qubit left outside.
static TileBlock to_block(const FusedOp &op, uint64_t inside) {
    TileBlock t;
    t.gates = op.gates;
    const int k = op.nq(), D = op.dim();
    const int qs[2] = {op.q_hi, op.q_lo};
    int in_pos[2], sel_pos[2]; // bit positions (in the op's row index) of the inside / outside qubits, most significant first
    for (int a = 0; a < k; a++) {
        const int pos = k - 1 - a;
        if (inside >> qs[a] & 1ULL) { in_pos[t.nq] = pos; t.q[t.nq++] = qs[a]; }
        else { sel_pos[t.ns] = pos; t.s[t.ns++] = qs[a]; }
    }
    const int d = 1 << t.nq;
    t.shape(t.nq, t.ns);
    auto compose = [&](int v, int r) { // op row index from bank index v and inside row index r
        int idx = 0;
        for (int a = 0; a < t.ns; a++) idx |= ((v >> (t.ns - 1 - a)) & 1) << sel_pos[a];
        for (int a = 0; a < t.nq; a++) idx |= ((r >> (t.nq - 1 - a)) & 1) << in_pos[a];
        return idx;
    };
    for (int v = 0; v < (1 << t.ns); v++)
        for (int r = 0; r < d; r++) {
            TileBlock::Row &row = t.row(v, r);
            for (int c = 0; c < d; c++) {
                const cd z = op.m[D * compose(v, r) + compose(v, c)];
                if (!is_zero(z)) { row.col[row.n] = (uint8_t)c; row.val[row.n++] = z; } // d <= 4 = kMaxRowNnz
            }
        }
    return t;
}

// ---- pass construction --------------------------------------------------------------------------------
void Scheduler::single_op_pass(const FusedOp &op, const PassSink &sink) const {
    const double S = 16.0 * (double)(1ULL << cfg_.n); // bytes of state
    Pass p;
    p.src = cur_src_;
    p.ops.push_back(op);
    if (op.kind == OP_G1) {
        if (op.is_identity()) return;
        if (op.is_diag()) {
            p.kclass = QSIM_K_PHASE;
            const bool unit0 = is_one(op.m[0]);
            p.diag_full = !unit0 || op.q_hi < 2; // below 64-B runs every sector is touched anyway
            p.bytes = unit0 ? S : 2 * S;
        } else {
            p.kclass = op.q_hi >= 6 ? QSIM_K_GATE1 : QSIM_K_GATE1_LO;
            p.bytes = 2 * S;
        }
        sink(std::move(p));
    } else if (op.kind == OP_CX) {
        p.kclass = QSIM_K_CX;
        p.bytes = S;
        sink(std::move(p));
    } else {
        if (op.is_identity()) return;
        // a pair cluster that is still exactly one CX (nothing else was folded into it) moves half the state, not all
        // of it: the swap kernel instead of the dense 4x4 (or a tile pass)
        for (int hi_ctl = 0; hi_ctl < 2; hi_ctl++) {
            cd ref[16];
            cx4(hi_ctl != 0, ref);
            bool same = true;
            for (int k = 0; k < 16 && same; k++) same = op.m[k] == ref[k];
            if (!same) continue;
            FusedOp cx;
            cx.kind = OP_CX;
            cx.q_hi = hi_ctl ? op.q_hi : op.q_lo; // control
            cx.q_lo = hi_ctl ? op.q_lo : op.q_hi; // target
            cx.gates = op.gates;
            p.ops[0] = cx;
            p.kclass = QSIM_K_CX;
            p.bytes = S;
            sink(std::move(p));
            return;
        }
        if (op.kind == OP_G2 && op.q_lo >= 6) {
            p.kclass = QSIM_K_GATE2;
            p.bytes = 2 * S;
            sink(std::move(p));
        } else {
            tile_pass(p.ops, op.qmask(), sink);
        }
    }
}

uint64_t Scheduler::tile_pass(const std::vector<FusedOp> &ops, uint64_t hset, const PassSink &sink, uint64_t prefer) const {
    const int B = std::min(cfg_.tile_bits, cfg_.n);
    const int L = std::min(cfg_.tile_low_bits, B);
    const uint64_t lowmask = (1ULL << L) - 1ULL;
    Pass p;
    p.src = cur_src_;
    p.kclass = QSIM_K_TILE;
    p.bytes = 32.0 * (double)(1ULL << cfg_.n);
    // `hset`: the high qubits the blocks NEED in the tile.  Qubits a block is merely block-diagonal in may stay outside
    // (they select a sub-block per tile); while slots are free they are taken in anyway, most used first, because a
    // block that lies entirely inside the tile can be merged with its neighbours.
    uint64_t high = hset & ~lowmask;
    {
        int uses[64] = {0};
        for (const FusedOp &op : ops)
            for (uint64_t rest = op.qmask() & ~lowmask & ~high; rest; rest &= rest - 1) uses[__builtin_ctzll(rest)]++;
        while (__builtin_popcountll(high) < B - L) {
            int best = -1;
            for (int b = L; b < cfg_.n; b++)
                if (uses[b] > 0 && (prefer >> b & 1ULL) && (best < 0 || uses[b] > uses[best])) best = b;
            if (best < 0) break;
            high |= 1ULL << best;
            uses[best] = 0;
        }
    }
    // unused slots are filled with free bits starting at pad_from, wrapping around to the low end.  Measured at
    // n = 30 (tools/pad_sweep.py): with three or four genuinely high qubits in the tile, padding with the lowest
    // bits (longest contiguous runs) costs up to 8.6 ms per pass against 6.8 ms when bits 10.. are used; 10 had the
    // best worst case over the geometries tried (<= 6.9 ms)
    const int start = cfg_.pad_from >= L && cfg_.pad_from < cfg_.n ? cfg_.pad_from : L;
    for (int round = 0; round < 2; round++) { // bits of `prefer` first: padding must not grow a partial state's support
        const uint64_t ok = round == 0 ? prefer : ~0ULL;
        for (int b = start; b < cfg_.n && __builtin_popcountll(high) < B - L; b++) if (ok >> b & 1ULL) high |= 1ULL << b;
        for (int b = L; b < start && __builtin_popcountll(high) < B - L; b++) if (ok >> b & 1ULL) high |= 1ULL << b;
    }
    p.geom.tile_bits = L + __builtin_popcountll(high);
    p.geom.low_bits = L;
    p.geom.n = cfg_.n;
    p.geom.n_high = 0;
    for (int b = L; b < cfg_.n; b++)
        if (high >> b & 1ULL) p.geom.high[p.geom.n_high++] = b;

    // Every block is split into the qubits it has inside the tile and the ones outside (it is block-diagonal in those:
    // build_passes only leaves such qubits out).  Blocks with nothing inside are tile-uniform factors: they commute
    // with everything else in the pass and go to the front, one entry per qubit set.
    const uint64_t inside = lowmask | high;
    std::vector<TileBlock> scalars, blocks;
    for (const FusedOp &op : ops) {
        TileBlock tb = to_block(op, inside);
        if (tb.nq > 0) { blocks.push_back(tb); continue; }
        bool folded = false;
        for (TileBlock &sc : scalars)
            if (sc.ns == tb.ns && sc.s[0] == tb.s[0] && sc.s[1] == tb.s[1]) {
                for (int v = 0; v < tb.banks(); v++) { // 1x1 banks: the factors multiply
                    const cd z = tb.at(v, 0, 0) * sc.at(v, 0, 0);
                    sc.row(v, 0).n = 1; sc.row(v, 0).col[0] = 0; sc.row(v, 0).val[0] = z;
                }
                sc.gates += tb.gates;
                folded = true;
                break;
            }
        if (!folded) scalars.push_back(tb);
    }
    if (cfg_.merge && blocks.size() > 1) merge_blocks(blocks);
    else blocks.erase(std::remove_if(blocks.begin(), blocks.end(), [](const TileBlock &t) { return t.is_identity(); }), blocks.end());
    scalars.erase(std::remove_if(scalars.begin(), scalars.end(), [](const TileBlock &t) { return t.is_identity(); }), scalars.end());
    p.geom.n_scale = (int)scalars.size();
    p.blocks = std::move(scalars);
    p.blocks.insert(p.blocks.end(), blocks.begin(), blocks.end());
    const uint64_t tmask = lowmask | high;
    if (p.blocks.empty()) return 0; // every block was the identity: nothing is launched, nothing changes
    if (prefer != ~0ULL) { // the state's support is known: the pass visits the tiles inside support | tile
        const uint64_t all = cfg_.n >= 64 ? ~0ULL : ((1ULL << cfg_.n) - 1ULL);
        p.visited = 1.0 / (double)(1ULL << (cfg_.n - __builtin_popcountll((prefer | tmask) & all)));
    }
    sink(std::move(p));
    return tmask;
}

void Scheduler::build_passes(const PassSink &sink) {
    auto note = [&](const std::vector<long> &idx) { // the gates behind the clusters of the pass about to be emitted
        cur_src_.clear();
        if (!cfg_.track) return;
        for (long i : idx) cur_src_.insert(cur_src_.end(), closed_src_[(size_t)i].begin(), closed_src_[(size_t)i].end());
    };
    if (cfg_.fuse <= 2) {
        for (size_t i = 0; i < closed_.size(); i++) { note({(long)i}); single_op_pass(closed_[i], sink); }
        return;
    }
    const int B = std::min(cfg_.tile_bits, cfg_.n);
    const int L = std::min(cfg_.tile_low_bits, B);
    const int kmax = std::min(B - L, (int)kMaxTileHigh);
    const uint64_t lowmask = (1ULL << L) - 1ULL;
    const uint64_t all = cfg_.n >= 64 ? ~0ULL : ((1ULL << cfg_.n) - 1ULL);
    const size_t m = closed_.size();
    std::vector<uint64_t> qm(m), must(m); // all qubits of a block (ordering); the ones that have to be tile qubits
    for (size_t i = 0; i < m; i++) {
        qm[i] = closed_[i].qmask();
        must[i] = cfg_.selectors ? (qm[i] & ~closed_[i].selector_mask()) : qm[i];
    }
    std::vector<char> done(m, 0), trial;
    uint64_t rng = cfg_.seed * 0x9E3779B97F4A7C15ULL + 0xD1B54A32D192ED03ULL; // xorshift64*: SchedConfig::seed
    auto coin = [&]() {
        if (!cfg_.seed) return false;
        rng ^= rng >> 12; rng ^= rng << 25; rng ^= rng >> 27;
        return ((rng * 0x2545F4914F6CDD1DULL) >> 63) != 0;
    };
    int cap = cfg_.tile_max_ops; // clusters per pass; lifted to tail_max_ops when that lets a pass finish the circuit
    size_t first = 0;
    std::vector<FusedOp> group;
    struct Cand { long idx; int need; };

    // Order: a cluster may run before an earlier pending one unless one of them MIXES a qubit they share — two clusters that
    // are both block-diagonal in every shared qubit (controls of CXs, diagonal gates) commute.  (The first version chained
    // everything on a shared qubit; with ~a quarter of the gates being CXs the controls were most of the chain.)
    const bool commute = cfg_.selectors && cfg_.commute;
    auto conflicts = [&](size_t i, uint64_t bm, uint64_t bs) {
        if (!commute) return (qm[i] & (bm | bs)) != 0;
        return (must[i] & (bm | bs)) != 0 || (qm[i] & ~must[i] & bm) != 0;
    };
    // Runnable blocks under the qubits chosen so far, cheapest (fewest new high-qubit slots) first.  A block is
    // runnable when no earlier pending block shares a qubit with it, directly or through a chain of pending blocks
    // (those qubits are "blocked"), so emitting blocks in the order they are picked respects every dependency.
    auto scan = [&](const std::vector<char> &dn, size_t from, size_t to, uint64_t hset, std::vector<Cand> *cands) -> Cand {
        Cand best{-1, 1 << 30};
        uint64_t bm = 0, bs = 0; // qubits mixed by / merely selecting in the pending clusters passed over so far
        const int used = __builtin_popcountll(hset);
        for (size_t i = from; i < to; i++) {
            if (dn[i]) continue;
            if (!conflicts(i, bm, bs)) {
                const int need = __builtin_popcountll(must[i] & ~lowmask & ~hset);
                if (used + need <= kmax) {
                    if (cands) cands->push_back({(long)i, need});
                    if (need < best.need) {
                        best = {(long)i, need};
                        if (need == 0) return best; // free: take it right away
                    }
                }
            }
            bm |= must[i];
            bs |= qm[i] & ~must[i];
            if (bm == all) break;
        }
        return best;
    };
    // A pass filled greedily from (dn, hset): up to `limit` times the cluster scan() would return, marked done in dn and its
    // qubits added to hset.  Same picks as calling scan() once per pick, without starting over each time: a pick that needs
    // no new qubit (scan's early return) changes nothing for the clusters in front of it — the scan simply goes on behind
    // it with the blocked qubits and the cheapest candidate seen so far; only a pick that admits a qubit (it comes after a
    // whole scan) changes what the others need, and the next scan starts over.  (The local search fills ~4000 trial passes
    // per schedule this way; with the pruned local search below 34 -> 15.6 ms of host time for the 1000 gates of the bench circuit.)
    auto fill = [&](std::vector<char> &dn, size_t from, size_t to, uint64_t &hset, int limit) {
        int cnt = 0;
        size_t i = from;
        uint64_t bm = 0, bs = 0;
        Cand best{-1, 1 << 30};
        int used = __builtin_popcountll(hset);
        while (cnt < limit) {
            long free_pick = -1;
            for (; i < to; i++) {
                if (dn[i]) continue;
                if (!conflicts(i, bm, bs)) {
                    const int need = __builtin_popcountll(must[i] & ~lowmask & ~hset);
                    if (need == 0) { free_pick = (long)i; break; } // used + 0 <= kmax always holds
                    if (used + need <= kmax && need < best.need) best = {(long)i, need};
                }
                bm |= must[i];
                bs |= qm[i] & ~must[i];
                if (bm == all) { i = to; break; }
            }
            if (free_pick >= 0) {
                dn[(size_t)free_pick] = 1;
                cnt++;
                i = (size_t)free_pick + 1; // not added to the blocked qubits: it ran
                continue;
            }
            if (best.idx < 0) break;
            dn[(size_t)best.idx] = 1;
            hset |= must[(size_t)best.idx] & ~lowmask;
            cnt++;
            i = from; bm = bs = 0; best = {-1, 1 << 30};
            used = __builtin_popcountll(hset);
        }
        return cnt;
    };
    // how many blocks a pass reaches when it is finished greedily from (dn, hset)
    auto rollout = [&](std::vector<char> &dn, size_t from, size_t to, uint64_t hset, int have) {
        int cnt = fill(dn, from, to, hset, cap - have);
        // look further: what the following passes reach when each is simply built greedily
        for (int extra = 0; extra < cfg_.lookahead; extra++) {
            uint64_t h2 = 0;
            cnt += fill(dn, from, to, h2, cap);
        }
        return cnt;
    };

    // Blocks a pass with high-qubit set S executes, in index order (a valid execution order: a block runs only if
    // every earlier pending block on its qubits ran), and their score.
    std::vector<long> picks, best_picks;
    bool endgame = false; // few blocks left: search the last pass sets so that no straggler pass remains
    // one_short (optional): the qubits b for which some cluster the walk reaches could run if b alone were added to S.  For any
    // other b, S | b executes exactly what S executes (the first cluster to be treated differently would have to be one of those).
    auto eval = [&](const std::vector<char> &dn, size_t from, size_t to, uint64_t S, std::vector<long> *out, uint64_t *one_short = nullptr) {
        uint64_t bm = 0, bs = 0;
        int score = 0, cnt = 0;
        if (out) out->clear();
        for (size_t i = from; i < to && cnt < cap; i++) {
            if (dn[i]) continue;
            const bool free_to_run = !conflicts(i, bm, bs);
            const uint64_t missing = must[i] & ~lowmask & ~S;
            if (free_to_run && !missing) {
                score += cfg_.objective ? (int)closed_[i].gates : 1;
                cnt++;
                if (out) out->push_back((long)i);
            } else {
                if (one_short && free_to_run && !(missing & (missing - 1))) *one_short |= missing;
                bm |= must[i];
                bs |= qm[i] & ~must[i];
                if (bm == all) break;
            }
        }
        return score;
    };
    // greedy passes needed to finish everything pending in [from, to) (at most `limit` are tried)
    auto passes_to_finish = [&](std::vector<char> &dn, size_t from, size_t to, int limit) {
        int n_pass = 0;
        for (; n_pass < limit; n_pass++) {
            bool pending = false;
            for (size_t i = from; i < to && !pending; i++) pending = !dn[i];
            if (!pending) break;
            uint64_t h2 = 0;
            if (fill(dn, from, to, h2, cap) == 0) return limit; // cannot happen (a pass always takes something); keeps the loop finite
        }
        return n_pass;
    };
    auto eval_ahead = [&](size_t from, size_t to, uint64_t S, std::vector<long> *out, uint64_t *one_short = nullptr) {
        int score = eval(done, from, to, S, out ? out : &picks, one_short);
        if (cfg_.lookahead > 0 || endgame) {
            trial = done;
            for (long i : (out ? *out : picks)) trial[(size_t)i] = 1;
            if (endgame) {
                // near the end what counts is how many MORE sweeps over the state the circuit needs (a straggler pass for
                // a handful of gates costs as much as a full one): fewer first, then more clusters in this pass
                const int more = passes_to_finish(trial, from, to, 8);
                score += (8 - more) * 100000;
            } else {
                score += rollout(trial, from, to, 0, 0); // the next pass, built greedily (plus cfg_.lookahead more inside)
            }
        }
        return score;
    };

    std::vector<Cand> cands;
    // The state's support as the engine will track it (SchedConfig::initial_support): tile passes add their tile's qubits,
    // anything else makes the engine write the zeros out (dense from then on).
    uint64_t support = cfg_.initial_support & all;
    auto note_tile = [&](uint64_t tmask) { if (tmask) support |= tmask; };
    std::vector<long> picks0;
    while (first < m) {
        if (done[first]) { first++; continue; }
        group.clear();
        uint64_t hset = 0;
        if (support != all && support != 0 && cfg_.cheap_margin > 0) {
            // A pass that stays inside the support visits 2^(|support| - n) of the register: greedy, cheapest cluster first
            const size_t end0 = std::min(m, first + (size_t)cfg_.window);
            std::vector<char> work0 = done;
            uint64_t h0 = 0;
            picks0.clear();
            while ((int)picks0.size() < cfg_.tile_max_ops) {
                long pick = -1;
                int bestneed = 1 << 30;
                uint64_t bm = 0, bs = 0;
                const int used = __builtin_popcountll(h0);
                for (size_t i = first; i < end0; i++) {
                    if (work0[i]) continue;
                    if (!conflicts(i, bm, bs) && !(must[i] & ~lowmask & ~support)) {
                        const int need = __builtin_popcountll(must[i] & ~lowmask & ~h0);
                        if (used + need <= kmax && need < bestneed) { bestneed = need; pick = (long)i; if (need == 0) break; }
                    }
                    bm |= must[i];
                    bs |= qm[i] & ~must[i];
                    if (bm == all) break;
                }
                if (pick < 0) break;
                work0[(size_t)pick] = 1;
                h0 |= must[(size_t)pick] & ~lowmask;
                picks0.push_back(pick);
            }
            const double share = 1.0 / (double)(1ULL << (cfg_.n - __builtin_popcountll(support)));
            if (picks0.size() >= 2 && (double)picks0.size() >= cfg_.cheap_margin * (double)cfg_.tile_max_ops * share) {
                std::sort(picks0.begin(), picks0.end());
                for (long i : picks0) { group.push_back(closed_[(size_t)i]); done[(size_t)i] = 1; }
                note(picks0);
                note_tile(tile_pass(group, h0, sink, support));
                continue;
            }
        }
        {   // the cap balances a pass's block phase against its memory time; a pass that would leave only a few clusters
            // for one more sweep over the state (6.6 ms at n = 30 for, on the bench circuit, ONE gate) takes them instead
            size_t left = 0;
            for (size_t i = first; i < m && left <= (size_t)std::max(cfg_.tail_max_ops, cfg_.tile_max_ops); i++) left += !done[i];
            cap = (cfg_.tail_max_ops > cfg_.tile_max_ops && left <= (size_t)cfg_.tail_max_ops) ? cfg_.tail_max_ops : cfg_.tile_max_ops;
        }
        const size_t end = std::min(m, first + (size_t)cfg_.window);
        // 1. greedy construction on a copy: cheapest new qubit first, ties broken by a rollout
        std::vector<char> work = done;
        int have = 0;
        while (have < cap) {
            cands.clear();
            Cand pick = scan(work, first, end, hset, cfg_.rollout > 1 ? &cands : nullptr);
            if (pick.idx < 0) break;
            if (pick.need > 0 && cfg_.rollout > 1 && cands.size() > 1) {
                // a new qubit has to be admitted: try the cheapest few candidates and keep the one after which a
                // greedy completion of this pass absorbs the most blocks
                std::stable_sort(cands.begin(), cands.end(), [](const Cand &a, const Cand &b) { return a.need < b.need; });
                int best_score = -1;
                const size_t tries = std::min(cands.size(), (size_t)cfg_.rollout);
                for (size_t t = 0; t < tries; t++) {
                    trial = work;
                    trial[(size_t)cands[t].idx] = 1;
                    const int score = rollout(trial, first, end, hset | (must[(size_t)cands[t].idx] & ~lowmask), have + 1);
                    if (score > best_score || (score == best_score && coin())) { best_score = score; pick = cands[t]; }
                }
            }
            hset |= must[(size_t)pick.idx] & ~lowmask;
            work[(size_t)pick.idx] = 1;
            have++;
        }
        // 2. local search over the qubit set: swap one chosen high qubit for one left out while the pass (and, with
        //    lookahead, the greedy passes after it) executes more.  The host has milliseconds per pass to spend here:
        //    the GPU is busy with the previous pass for ~7 ms at n = 30.
        {
            size_t left = 0;
            for (size_t i = first; i < m && left <= 2 * (size_t)cfg_.tile_max_ops; i++) left += !done[i];
            endgame = left <= 2 * (size_t)cfg_.tile_max_ops;
        }
        const int iters = endgame ? std::max(cfg_.local_iters, 3) : cfg_.local_iters;
        if (iters > 0 && __builtin_popcountll(hset) >= 2) {
            int best = eval_ahead(first, end, hset, &best_picks);
            for (int it = 0; it < iters; it++) {
                uint64_t bestS = hset;
                for (uint64_t in = hset; in; in &= in - 1) {
                    const uint64_t qi = in & (0 - in);
                    // the set without qi, once: most qubits b put in qi's place change nothing about what the pass executes
                    // (no reachable cluster is short of exactly b), and those all score what the smaller set scores
                    uint64_t one_short = 0;
                    const int v_without = eval_ahead(first, end, hset & ~qi, nullptr, &one_short);
                    for (int b = L; b < cfg_.n; b++) {
                        if (hset >> b & 1ULL) continue;
                        const uint64_t S2 = (hset & ~qi) | (1ULL << b);
                        const int v = (one_short >> b & 1ULL) ? eval_ahead(first, end, S2, nullptr) : v_without;
                        if (v > best || (v == best && bestS != hset && coin())) { best = v; bestS = S2; }
                    }
                }
                if (bestS == hset) break;
                hset = bestS;
            }
        }
        eval(done, first, end, hset, &best_picks);
        if (best_picks.empty()) { // the set cannot be worse than the greedy one; keep the scheduler total anyway
            best_picks.push_back((long)first);
            hset = must[first] & ~lowmask;
        }
        for (long i : best_picks) {
            group.push_back(closed_[(size_t)i]);
            done[(size_t)i] = 1;
        }
        note(group.empty() ? std::vector<long>{(long)first} : best_picks);
        if (group.empty()) { // cannot happen while kmax >= 2; keep the scheduler total anyway
            single_op_pass(closed_[first], sink);
            done[first] = 1;
            support = all;
        } else if (group.size() == 1) {
            single_op_pass(group[0], sink); // may be a tile pass of its own or a single-gate kernel: count it as dense
            support = all;
        } else {
            note_tile(tile_pass(group, hset, sink, support == all ? ~0ULL : support));
        }
    }
}

// ---- sparse merging inside a pass -------------------------------------------------------------------------
// Most fused clusters are permutations-times-phases or two independent 2x2 blocks (exact zeros), so the product
// of neighbours on a few tile qubits usually still has <= 4 entries per row.  Such a product costs ONE trip through
// LDS in k_tile instead of one per factor, which is what bounds a pass once it carries more than ~6 blocks.
// Order: a block may hop over earlier blocks it shares no TILE qubit with (they commute: outside the tile every block
// of the pass is block-diagonal); everything it shares a tile qubit with and cannot join blocks those qubits for the
// rest of the scan.  Selecting qubits are merged too: the product has one bank per value of the union (at most two).
namespace {

// Row `r` of bank `bv` of block `b`, embedded into the space of the tile qubits qs[0..k) (descending): the block acts
// on its own qubits and as the identity on the rest, so the row keeps its entries with the spectator bits copied.
struct Embedding {
    int pos[kMaxBlockQ]; // bit position (inside the k-bit index) of each of the block's qubits
    int mask = 0;
    Embedding(const TileBlock &b, const int *qs, int k) {
        for (int a = 0; a < b.nq; a++) {
            pos[a] = 0;
            for (int j = 0; j < k; j++)
                if (qs[j] == b.q[a]) pos[a] = k - 1 - j;
            mask |= 1 << pos[a];
        }
    }
    int sub(const TileBlock &b, int idx) const { // idx restricted to the block's qubits, most significant first
        int r = 0;
        for (int a = 0; a < b.nq; a++) r = (r << 1) | ((idx >> pos[a]) & 1);
        return r;
    }
    int spread(const TileBlock &b, int sub_idx) const {
        int r = 0;
        for (int a = 0; a < b.nq; a++) r |= ((sub_idx >> (b.nq - 1 - a)) & 1) << pos[a];
        return r;
    }
};

// the block's own bank index under the joint selector value v over ss[0..nss)
int own_bank(const TileBlock &b, const int *ss, int nss, int v) {
    int bv = 0;
    for (int a = 0; a < b.ns; a++)
        for (int j = 0; j < nss; j++)
            if (ss[j] == b.s[a]) bv |= ((v >> (nss - 1 - j)) & 1) << (b.ns - 1 - a);
    return bv;
}

} // namespace

void Scheduler::merge_blocks(std::vector<TileBlock> &blocks) const {
    // Tiles of fewer than 2^11 amplitudes keep to 3 qubits: a block on k > 3 qubits deals the 2^(k-3) parts of a group to
    // different waves, so a part must be at least one wave: 2^(B-k) >= 64 groups (k_tile's tile_op_part).  2^11 tiles take
    // 5 qubits, 2^12 and larger 6.
    const int B = std::min(cfg_.tile_bits, cfg_.n);
    const int kMaxQ = B >= 11 ? std::max(3, std::min(std::min(cfg_.merge_qubits, kMaxBlockQ), B - 6)) : 3;
    constexpr int kMaxSel = 2;
    std::vector<int> rem(blocks.size()), next; // indices into `blocks`: the blocks themselves are moved, never copied
    for (size_t i = 0; i < blocks.size(); i++) rem[i] = (int)i;
    std::vector<TileBlock> out;
    TileBlock m;
    while (!rem.empty()) {
        TileBlock cur = std::move(blocks[(size_t)rem[0]]);
        uint64_t blocked = 0;
        next.clear();
        for (size_t i = 1; i < rem.size(); i++) {
            const TileBlock &op = blocks[(size_t)rem[i]];
            const uint64_t qm = op.in_mask();
            if (qm & blocked) { blocked |= qm; next.push_back(rem[i]); continue; }
            const uint64_t un = cur.in_mask() | qm, us = cur.sel_mask() | op.sel_mask();
            bool merged = false;
            if (__builtin_popcountll(un) <= kMaxQ && __builtin_popcountll(us) <= kMaxSel) {
                int qs[kMaxBlockQ], k = 0, ss[2], nss = 0;
                for (int b = 63; b >= 0; b--) {
                    if (un >> b & 1ULL) qs[k++] = b;
                    if (us >> b & 1ULL) ss[nss++] = b;
                }
                const int D = 1 << k;
                m = TileBlock();
                m.shape(k, nss);
                for (int a = 0; a < k; a++) m.q[a] = qs[a];
                for (int a = 0; a < nss; a++) m.s[a] = ss[a];
                m.gates = cur.gates + op.gates;
                const Embedding ea(cur, qs, k), eb(op, qs, k);
                bool fits = true;
                for (int v = 0; v < (1 << nss) && fits; v++) {
                    const int va = own_bank(cur, ss, nss, v), vb = own_bank(op, ss, nss, v);
                    for (int r = 0; r < D && fits; r++) { // row r of (op after cur) = sum_t op[r][t] * cur[t][.]
                        int cols[kMaxRowNnz * kMaxRowNnz], n = 0;
                        cd vals[kMaxRowNnz * kMaxRowNnz];
                        const TileBlock::Row &rb = op.row(vb, eb.sub(op, r));
                        for (int jb = 0; jb < rb.n; jb++) {
                            const int t = (r & ~eb.mask) | eb.spread(op, rb.col[jb]);
                            const TileBlock::Row &ra = cur.row(va, ea.sub(cur, t));
                            for (int ja = 0; ja < ra.n; ja++) {
                                const int c = (t & ~ea.mask) | ea.spread(cur, ra.col[ja]);
                                const cd z = rb.val[jb] * ra.val[ja];
                                int e = 0;
                                while (e < n && cols[e] != c) e++;
                                if (e == n) { cols[n] = c; vals[n++] = z; }
                                else vals[e] += z;
                            }
                        }
                        TileBlock::Row &row = m.row(v, r);
                        row.n = 0;
                        for (int e = 0; e < n; e++) {
                            if (is_zero(vals[e])) continue; // exact cancellation
                            // one LDS trip evaluates at most kMaxRowNnz entries per row — except on two qubits, where
                            // that is all four columns anyway
                            if (row.n == kMaxRowNnz) { fits = false; break; }
                            row.col[row.n] = (uint8_t)cols[e];
                            row.val[row.n++] = vals[e];
                        }
                    }
                }
                if (fits && k >= 2) fits = m.classes_feasible(); // one LDS trip evaluates whole row classes (TileBlock::classes)
                if (fits) {
                    for (int v = 0; v < (1 << nss); v++) // keep every row's entries in ascending column order
                        for (int r = 0; r < D; r++) {
                            TileBlock::Row &row = m.row(v, r);
                            for (int x = 1; x < row.n; x++)
                                for (int y = x; y > 0 && row.col[y - 1] > row.col[y]; y--) {
                                    std::swap(row.col[y - 1], row.col[y]);
                                    std::swap(row.val[y - 1], row.val[y]);
                                }
                        }
                    std::swap(cur, m);
                    merged = true;
                }
            }
            if (!merged) { blocked |= qm; next.push_back(rem[i]); }
        }
        if (!cur.is_identity()) out.push_back(std::move(cur));
        rem.swap(next);
    }
    blocks.swap(out);
}

} // namespace qsim
