// dist.cpp — sharded state vector inside ONE process: P = 2^p shards, each a qsim_state on some device (the same
// device may appear several times: "virtual shards", used to validate the sharded path where fewer than P GPUs
// exist).  New design — the reference is single-device.
//
// Physical index bits 0..m-1 (m = n - p) are local to a shard, bits m..n-1 are the shard id.  A host-side
// logical->physical qubit map decides what needs data movement:
//   * gates on local qubits run through the single-GPU engine on every shard;
//   * a diagonal gate on a global qubit is a per-shard scalar, a CX with global control and local target is an X on
//     the shards whose control bit is 1 — no communication;
//   * anything else on a global qubit waits; when nothing more can run, ONE exchange swaps k global qubits with k
//     local ones: k_pack lays every shard out as 2^k contiguous blocks, then block b of shard r goes to group
//     member b (device-to-device copies here; the one-process-per-GPU driver in distributed.py does the same with
//     RCCL send/recv).  New globals = furthest next non-diagonal use (Belady); the first placement is free because
//     |0...0> is symmetric under qubit permutations.
// The planner below is the C++ twin of distributed.ShardPlan (tests compare the two step by step).
#include <algorithm>
#include <complex>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include <rccl/rccl.h>

#include "circuit.h"
#include "qsim_internal.h"
#include "scheduler.h"

using cd = std::complex<double>;

namespace {

struct LGate { // logical gate
    int kind;  // QSIM_GATE_U1 / QSIM_GATE_CX
    int q0, q1;
    cd m[4];
    long idx = 0; // position in the circuit
    bool diag() const { return m[1] == cd(0, 0) && m[2] == cd(0, 0); }
};

struct LocalOp { // per-shard op in local physical coordinates
    int kind;    // 1 = u1, 2 = cx, 3 = scale
    int a, b;
    cd m[4];
};

struct Step {
    bool exchange = false;
    std::vector<int> J, Lsel;                    // exchange: shard-id bits and local positions, ascending, paired
    std::vector<std::vector<LocalOp>> per_shard; // local: ops for every shard
    // exchange: where the state can be non-zero just before it, as PHYSICAL bit sets (local positions / shard-id bits).  A run
    // starts from |0...0> (quantum_simulator.c:175-177) and a qubit stays |0> until a gate mixes it (a non-diagonal 1-qubit
    // gate, or a CX onto it whose control may be 1), so every amplitude with a 1 at a qubit outside this set is exactly
    // zero.  The same on every rank (it follows from the gate list alone), which is what lets an exchange leave out the
    // blocks of shards that hold nothing and lets the receivers keep visiting only the part of the shard that can be non-zero.
    uint64_t mixed_local = 0, mixed_rank = 0;
};

struct Plan {
    int n = 0, p = 0, m = 0;
    std::vector<Step> steps;
    std::vector<int> final_pos;
    int exchanges = 0;
    int tail_gates = 0; // gate statements handed on across an exchange (small_tail)
    double local_sweeps = 0; // predicted time of the local steps on their busiest shard, in sweeps of the shard (pass_time_cost)
};

constexpr long kInf = 1L << 60;

// qubits this gate needs in LOCAL positions
void needs_local(const LGate &g, int out[2], int &cnt) {
    cnt = 0;
    if (g.kind == QSIM_GATE_CX) {
        if (g.q0 != g.q1) out[cnt++] = g.q1;
    } else if (!g.diag()) {
        out[cnt++] = g.q0;
    }
}

// local_only: candidates are the qubits that are local now, so an exchange swaps ALL p global qubits (k = p).  On a
// fully connected node a k-qubit swap sends 2^k - 1 blocks of 2^-k of the shard over as many links at once, so its
// time FALLS with k; whether the extra qubits it evicts come back too soon is what plan_cost decides.
std::vector<int> choose_globals(const std::vector<LGate> &gates, const std::vector<int> &pos, int n, int p, int m, bool local_only = false) {
    if (m < p) local_only = false; // fewer local qubits than global ones: nothing to choose from
    std::vector<long> nxt(n, kInf);
    int found = 0;
    for (size_t i = 0; i < gates.size() && found < n; i++) {
        int q[2], c;
        needs_local(gates[i], q, c);
        for (int k = 0; k < c; k++)
            if (nxt[q[k]] == kInf) { nxt[q[k]] = (long)i; found++; }
    }
    std::vector<int> order;
    for (int q = 0; q < n; q++)
        if (!local_only || pos[q] < m) order.push_back(q);
    // far next use first; then already-global (nothing to move); then a high position — same key as the Python twin
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (nxt[a] != nxt[b]) return nxt[a] > nxt[b];
        const bool ga = pos[a] >= m, gb = pos[b] >= m;
        if (ga != gb) return ga;
        return pos[a] > pos[b];
    });
    order.resize(p);
    return order;
}

void peers_of(int rank, const std::vector<int> &J, int &mine, std::vector<int> &members) {
    const int k = (int)J.size();
    mine = 0;
    int base = rank;
    for (int i = 0; i < k; i++) { mine |= ((rank >> J[i]) & 1) << i; base &= ~(1 << J[i]); }
    members.resize((size_t)1 << k);
    for (int b = 0; b < (1 << k); b++) {
        int r = base;
        for (int i = 0; i < k; i++) r |= ((b >> i) & 1) << J[i];
        members[b] = r;
    }
}

// Who sends what in one exchange, for one rank, when only part of the register can be non-zero (Step::mixed_*).  A shard whose
// id has a 1 at a shard-id bit outside mixed_rank holds nothing; after the exchange the J bits of its id carry the qubits that
// sat at the local positions Lsel, so a shard is empty afterwards when one of THOSE is outside mixed_local.  Nothing travels
// from or to an empty shard, and a receiver's new contents can only be non-zero where the local index stays inside
// `new_support`: the surviving mixed local positions, moved down over the ones that left, plus the top k positions (the
// sender's member index) for the shard-id bits that were mixed.
struct Roles {
    int mine = 0;
    std::vector<int> members;
    bool empty_before = false, empty_after = false;
    uint32_t send = 0, recv = 0; // bit b: block b goes to / comes from members[b] (never bit `mine`)
    bool keep_own = false;       // block `mine` stays here and holds data
    uint32_t unread = 0;         // blocks of this rank's packed layout nobody looks at
    uint64_t new_support = 0;
};
constexpr int kMaxRoleBits = 5; // Roles::send / recv / unread and k_pack's skip mask have one bit per block: groups of at most 32 shards
Roles roles_of(int rank, int m, const Step &st) {
    Roles r;
    const int k = (int)st.J.size(); // callers bound k by kMaxRoleBits (exchange, rank_exchange, the planner's P <= 32)
    peers_of(rank, st.J, r.mine, r.members);
    uint32_t jmask = 0, jin = 0, lin = 0;
    uint64_t lsel = 0;
    for (int i = 0; i < k; i++) {
        jmask |= 1u << st.J[i];
        lsel |= 1ULL << st.Lsel[i];
        if (st.mixed_rank >> st.J[i] & 1ULL) jin |= 1u << i;
        if (st.mixed_local >> st.Lsel[i] & 1ULL) lin |= 1u << i;
    }
    const bool base_ok = (((uint64_t)rank & ~(uint64_t)jmask) & ~st.mixed_rank) == 0;
    auto before = [&](int b) { return !base_ok || ((uint32_t)b & ~jin) != 0; }; // member b holds nothing before / after
    auto after = [&](int b) { return !base_ok || ((uint32_t)b & ~lin) != 0; };
    r.empty_before = before(r.mine);
    r.empty_after = after(r.mine);
    for (int b = 0; b < (1 << k); b++) {
        if (after(b)) r.unread |= 1u << b;
        if (b == r.mine) continue;
        if (!r.empty_before && !after(b)) r.send |= 1u << b;
        if (!r.empty_after && !before(b)) r.recv |= 1u << b;
    }
    r.keep_own = !r.empty_before && !r.empty_after;
    for (int b = 0; b < m; b++) {
        if (!(st.mixed_local >> b & 1ULL) || (lsel >> b & 1ULL)) continue;
        r.new_support |= 1ULL << (b - __builtin_popcountll(lsel & ((1ULL << b) - 1ULL)));
    }
    for (int i = 0; i < k; i++)
        if (jin >> i & 1u) r.new_support |= 1ULL << (m - k + i);
    return r;
}
// What the shard's engine is told once the blocks are in place.
int settle(qsim_state *s, const Roles &r) {
    return r.empty_after ? qsim_reset_shard(s, 0) : qsim_set_support(s, r.new_support);
}

// The gates of `run` (indices) that the engine's scheduler would put into the LAST pass of the segment, when that pass is
// a small one: the segment's gates are scheduled here exactly as a shard's engine will schedule them (same Scheduler, same
// settings; the shard-dependent ops in their busiest form: a CX controlled by a shard-id bit as an X), with every pass
// listing the gates it absorbed.  A segment ends where the next gate needs a qubit that is not local, not where a pass
// is full, so its last pass often carries a handful of gates and still costs a whole sweep over the shard — on every
// segment.  Those gates can just as well wait for the exchange: they come last in a valid order, and what they touch stays
// local (the planner evicts by furthest next use, and theirs is now the nearest).
std::vector<size_t> small_tail(const std::vector<LGate> &run, const std::vector<int> &pos, int m, bool from_reset, int max_gates) {
    std::vector<size_t> out;
    if (max_gates <= 0 || m < 12 || run.size() < 2) return out;
    qsim::SchedConfig cfg = qsim::engine_sched_config(m, 3, 12, 3, 32, 10, false, from_reset ? 0 : ~0ULL);
    cfg.track = 1;
    qsim::Scheduler sched(cfg);
    std::vector<size_t> which; // scheduler gate number -> index into run
    static const cd X[4] = {cd(0, 0), cd(1, 0), cd(1, 0), cd(0, 0)};
    for (size_t i = 0; i < run.size(); i++) {
        const LGate &g = run[i];
        if (g.kind == QSIM_GATE_CX) {
            if (g.q0 == g.q1) continue;
            if (pos[g.q0] < m) sched.add_cx(pos[g.q0], pos[g.q1]);
            else sched.add_1q(X, pos[g.q1]);
        } else if (pos[g.q0] < m) {
            sched.add_1q(g.m, pos[g.q0]);
        } else {
            continue; // a factor per shard
        }
        which.push_back(i);
    }
    std::vector<qsim::Pass> passes;
    sched.finish(passes);
    if (passes.size() < 2 || passes.back().kclass != QSIM_K_TILE) return out;
    const qsim::Pass &last = passes.back();
    if ((int)last.src.size() > max_gates) return out;
    for (uint32_t gi : last.src) out.push_back(which[gi]);
    std::sort(out.begin(), out.end());
    return out;
}

// Where a shard's state can be non-zero after a local step, as its engine will know it: the step's ops are scheduled exactly
// as qsim_flush will schedule them, and whatever lies outside (support before | the tile qubits of the passes) has not been
// touched since it was zero.  ANY valid schedule of the same ops gives a valid bound (the final state does not depend on the
// schedule), so it does not matter whether the engine later takes this very schedule or another variant of it; what matters
// is that every rank computes the same mask, which it does: same plan, same code.  Tighter than counting the qubits some
// non-diagonal gate has touched (x q; cx q,t; x q leaves q where it was, and the fused cluster shows it).
// cost_sweeps (optional): += what the step is predicted to take, in sweeps of the shard (pass_time_cost).
uint64_t scheduled_support(const std::vector<LocalOp> &ops, int m, uint64_t support, double *cost_sweeps = nullptr) {
    const uint64_t all = m >= 64 ? ~0ULL : ((1ULL << m) - 1ULL);
    if (ops.empty() || (!cost_sweeps && (support & all) == all)) return support & all;
    qsim::Scheduler sched(qsim::engine_sched_config(m, 3, 12, 3, 32, 10, false, (support & all) == all ? ~0ULL : (support & all)));
    for (const LocalOp &o : ops) {
        if (o.kind == 2) sched.add_cx(o.a, o.b);
        else if (o.kind == 1) sched.add_1q(o.m, o.a);
        else { const cd d[4] = {o.m[0], cd(0, 0), cd(0, 0), o.m[0]}; sched.add_1q(d, 0); }
    }
    uint64_t sup = support & all;
    sched.finish([&](qsim::Pass &&ps) {
        if (cost_sweeps) *cost_sweeps += qsim::pass_time_cost(ps, false) / (32.0 * (double)(1ULL << m));
        if (ps.kclass != QSIM_K_TILE) { sup = all; return; } // a single-gate kernel: the engine writes the zeros out first
        sup |= (1ULL << ps.geom.low_bits) - 1ULL;
        for (int j = 0; j < ps.geom.n_high; j++) sup |= 1ULL << ps.geom.high[j];
    });
    return sup & all;
}

// QSIM_SHARD_TAIL: the largest last pass (in gate statements) that is handed on to the next segment; 0 = never (the plain
// "run everything that can run" planner, which tests/py_shard_plan.py restates).  An experiment override like QSIM_SCHED_*.
int tail_limit() {
    if (const char *v = getenv("QSIM_SHARD_TAIL")) return atoi(v);
    return 24;
}

// want_cost: also price the local steps (Plan::local_sweeps) — every holding shard's every segment is then scheduled even when
// its support is already everything; without it only the segments whose support can still grow are (a few at the start of a run).
bool build_plan_policy(int n, int p, const std::vector<LGate> &gates, Plan &plan, bool full_swap, int tail, bool want_cost) {
    const int P = 1 << p, m = n - p;
    plan.n = n; plan.p = p; plan.m = m;
    std::vector<int> pos(n);
    for (int q = 0; q < n; q++) pos[q] = q;
    std::vector<LGate> remaining(gates);
    bool first = true;
    uint64_t mixed = 0; // logical qubits some gate may have moved away from |0> (a first, gate-level bound for Step::mixed_local)
    // ... and the bound the shards' engines will have themselves (scheduled_support), per shard; holds[r] = 0: shard r holds nothing
    std::vector<uint64_t> sup((size_t)P, 0);
    std::vector<char> holds((size_t)P, 0);
    holds[0] = 1;
    while (!remaining.empty()) {
        if (p && first) { // free initial placement
            std::vector<int> ng = choose_globals(remaining, pos, n, p, m);
            std::vector<int> outgoing, incoming;
            for (int q : ng) if (pos[q] < m) outgoing.push_back(q);
            for (int q = 0; q < n; q++)
                if (pos[q] >= m && std::find(ng.begin(), ng.end(), q) == ng.end()) incoming.push_back(q);
            for (size_t i = 0; i < outgoing.size() && i < incoming.size(); i++) std::swap(pos[outgoing[i]], pos[incoming[i]]);
        }
        first = false;
        // split into runnable / deferred
        std::vector<LGate> run, deferred;
        uint64_t blocked = 0;
        for (const LGate &g : remaining) {
            uint64_t qs = 1ULL << g.q0;
            if (g.kind == QSIM_GATE_CX) qs |= 1ULL << g.q1;
            if (qs & blocked) { blocked |= qs; deferred.push_back(g); continue; }
            int q[2], c;
            needs_local(g, q, c);
            bool ok = true;
            for (int k = 0; k < c; k++) ok = ok && pos[q[k]] < m;
            if (ok) run.push_back(g);
            else { blocked |= qs; deferred.push_back(g); }
        }
        if (p && !deferred.empty()) { // an exchange follows: a small last pass waits for it
            // What the scheduler put into the last pass is only a PROPOSAL: it saw one shard's version of the segment (a CX controlled
            // by a shard-id bit as an X; on the shards where that bit is 0 there is no gate at all, and products that cancel on one
            // shard do not on another), so its order proves nothing for the others.  A gate may wait for the exchange if it
            // commutes, by what it IS — not by what some product of matrices happens to be —, with every gate of the segment
            // that comes after it in the circuit and stays: no shared qubit, or only qubits in which both are block-diagonal
            // (a diagonal gate, the control of a CX).  Gates that fail stay, which can make others fail: iterate.
            std::vector<size_t> tail_gates = small_tail(run, pos, m, plan.steps.empty(), tail);
            {
                auto diag_mask = [](const LGate &g) -> uint64_t { // qubits the gate is block-diagonal in
                    if (g.kind == QSIM_GATE_CX) return g.q0 == g.q1 ? 0 : 1ULL << g.q0;
                    return g.diag() ? 1ULL << g.q0 : 0;
                };
                auto qubits = [](const LGate &g) -> uint64_t { return (1ULL << g.q0) | (g.kind == QSIM_GATE_CX ? 1ULL << g.q1 : 0); };
                std::vector<char> moving(run.size(), 0);
                for (size_t t : tail_gates) moving[t] = 1;
                for (bool changed = true; changed;) {
                    changed = false;
                    uint64_t later_mix = 0, later_any = 0; // over the staying gates behind the current position: qubits they mix / touch
                    for (size_t i = run.size(); i-- > 0;) {
                        const LGate &g = run[i];
                        const uint64_t q = qubits(g), d = diag_mask(g);
                        if (moving[i]) {
                            // shared qubits must be diagonal on both sides: none of g's qubits may be mixed later, none of g's mixed qubits touched later
                            if ((q & later_mix) || ((q & ~d) & later_any)) { moving[i] = 0; changed = true; }
                        }
                        if (!moving[i]) { later_mix |= q & ~d; later_any |= q; }
                    }
                }
                tail_gates.clear();
                for (size_t i = 0; i < run.size(); i++)
                    if (moving[i]) tail_gates.push_back(i);
            }
            if (!tail_gates.empty()) {
                std::vector<LGate> keep, moved;
                size_t t = 0;
                for (size_t i = 0; i < run.size(); i++) {
                    if (t < tail_gates.size() && tail_gates[t] == i) { moved.push_back(run[i]); t++; }
                    else keep.push_back(run[i]);
                }
                moved.insert(moved.end(), deferred.begin(), deferred.end()); // in front of what was deferred already: nothing there precedes them on a shared qubit
                deferred.swap(moved);
                run.swap(keep);
                plan.tail_gates += (int)tail_gates.size();
            }
        }
        for (const LGate &g : run) { // in program order
            if (g.kind == QSIM_GATE_CX) { if (g.q0 != g.q1 && (mixed >> g.q0 & 1ULL)) mixed |= 1ULL << g.q1; }
            else if (!g.diag()) mixed |= 1ULL << g.q0;
        }
        if (!run.empty()) {
            Step st;
            st.per_shard.resize(P);
            for (int r = 0; r < P; r++) {
                std::vector<LocalOp> &ops = st.per_shard[r];
                for (const LGate &g : run) {
                    LocalOp o{};
                    if (g.kind == QSIM_GATE_CX) {
                        if (g.q0 == g.q1) continue;
                        if (pos[g.q0] < m) { o.kind = 2; o.a = pos[g.q0]; o.b = pos[g.q1]; ops.push_back(o); }
                        else if ((r >> (pos[g.q0] - m)) & 1) {
                            o.kind = 1; o.a = pos[g.q1];
                            o.m[0] = 0; o.m[1] = 1; o.m[2] = 1; o.m[3] = 0;
                            ops.push_back(o);
                        }
                    } else if (pos[g.q0] < m) {
                        o.kind = 1; o.a = pos[g.q0];
                        std::copy(g.m, g.m + 4, o.m);
                        ops.push_back(o);
                    } else {
                        const int b = (r >> (pos[g.q0] - m)) & 1;
                        const cd z = g.m[b ? 3 : 0];
                        if (z != cd(1, 0)) { o.kind = 3; o.m[0] = z; ops.push_back(o); }
                    }
                }
            }
            if (p) {
                double worst = 0; // the step takes as long as its busiest shard: the first and the last shard stand for all
                for (int r = 0; r < P; r++) {
                    if (!holds[(size_t)r]) continue;
                    double cost = 0;
                    const bool rep = want_cost && (r == 0 || r == P - 1);
                    sup[(size_t)r] = scheduled_support(st.per_shard[(size_t)r], m, sup[(size_t)r], rep ? &cost : nullptr);
                    worst = std::max(worst, cost);
                }
                plan.local_sweeps += worst;
            }
            plan.steps.push_back(std::move(st));
        }
        if (!deferred.empty()) {
            std::vector<int> ng = choose_globals(deferred, pos, n, p, m, full_swap);
            std::vector<int> outgoing, incoming;
            for (int q : ng) if (pos[q] < m) outgoing.push_back(q);
            for (int q = 0; q < n; q++)
                if (pos[q] >= m && std::find(ng.begin(), ng.end(), q) == ng.end()) incoming.push_back(q);
            std::sort(outgoing.begin(), outgoing.end(), [&](int a, int b) { return pos[a] < pos[b]; });
            std::sort(incoming.begin(), incoming.end(), [&](int a, int b) { return pos[a] < pos[b]; });
            const int k = (int)outgoing.size();
            if (k == 0 || k != (int)incoming.size()) return false; // no progress possible
            Step st;
            st.exchange = true;
            for (int q : outgoing) st.Lsel.push_back(pos[q]);
            for (int q : incoming) st.J.push_back(pos[q] - m);
            uint64_t gate_level = 0, engine_level = 0;
            for (int q = 0; q < n; q++)
                if ((mixed >> q & 1ULL) && pos[q] < m) gate_level |= 1ULL << pos[q];
            for (int r = 0; r < P; r++)
                if (holds[(size_t)r]) { engine_level |= sup[(size_t)r]; st.mixed_rank |= (uint64_t)r; }
            st.mixed_local = gate_level & engine_level; // both are bounds on where the state can be non-zero
            for (int r = 0; r < P; r++) { // what every shard holds afterwards
                const Roles ro = roles_of(r, m, st);
                holds[(size_t)r] = !ro.empty_after;
                sup[(size_t)r] = ro.empty_after ? 0 : ro.new_support;
            }
            std::vector<int> np(pos);
            for (int q = 0; q < n; q++)
                if (pos[q] < m && std::find(st.Lsel.begin(), st.Lsel.end(), pos[q]) == st.Lsel.end()) {
                    int below = 0;
                    for (int s : st.Lsel) below += s < pos[q];
                    np[q] = pos[q] - below;
                }
            for (int i = 0; i < k; i++) np[incoming[i]] = m - k + i;
            for (int i = 0; i < k; i++) np[outgoing[i]] = m + st.J[i];
            pos = np;
            plan.steps.push_back(std::move(st));
            plan.exchanges++;
        }
        remaining.swap(deferred);
    }
    plan.final_pos = pos;
    return true;
}

// Exchange cost of a plan in integer units (so that the C++ planner and its Python twin decide identically): one
// exchange of k qubits = a pack pass over the shard (2 S bytes of HBM traffic) + 2^-k of the shard over each of 2^k - 1
// links in parallel.  With S / link = kLinkUnits and 2 S / HBM = kPackUnits (defaults: 50 GB/s per link direction, 5 TB/s
// pack kernel, i.e. 200 : 1 per byte; qsim_shard_plan_predict takes the real figures) the cost is additive.
constexpr long kLinkUnits = 25600, kPackUnits = 256;
long plan_cost(const Plan &plan) {
    long c = 0;
    for (const Step &st : plan.steps)
        if (st.exchange) c += kPackUnits + (kLinkUnits >> st.J.size());
    return c;
}

// Two placement policies are planned in full and the cheaper plan (by plan_cost) is kept; ties keep the first.
// One sweep of a shard in plan_cost's units: 2 S bytes at the tile kernel's ~4.5 TB/s against S over a 50 GB/s link = kLinkUnits.
constexpr double kSweepUnits = 570.0;

// The placement policies (keep far-next-use globals / swap all log2 P of them) and, unless QSIM_SHARD_TAIL pins it, a few
// limits for the hand-over of a segment's small last pass are planned in full; the plan with the least predicted time — the
// exchanges over the links (plan_cost) plus the local steps on their busiest shard (Plan::local_sweeps: the segments scheduled
// with the engine's own scheduler and priced by pass_time_cost) — is kept; ties keep the first.  With QSIM_SHARD_TAIL set only
// the two policies are compared, by their exchanges alone: the planner proper, which tests/py_shard_plan.py restates.
bool build_plan(int n, int p, const std::vector<LGate> &gates, Plan &plan) {
    const bool pinned = getenv("QSIM_SHARD_TAIL") != nullptr;
    // the search schedules every segment of every candidate plan: seconds for the circuits it is meant for (thousands of
    // gates on registers that need many GPUs), minutes for a 400 000-gate file like the reference's own benchmark circuits
    // (OverallTest.csv) — those get the default plan
    const bool search = !pinned && p > 0 && gates.size() <= 20000;
    std::vector<int> tails{tail_limit()};
    if (search) for (int t : {0, 12, 40}) tails.push_back(t);
    bool have = false;
    double best = 0;
    for (int tail : tails)
        for (int full = 0; full < (p > 1 ? 2 : 1); full++) {
            Plan cand;
            if (!build_plan_policy(n, p, gates, cand, full != 0, tail, search)) { if (!have && tail == tails[0] && full == 0) return false; continue; }
            const double cost = (double)plan_cost(cand) + (search ? kSweepUnits * cand.local_sweeps : 0.0);
            if (!have || cost < best) { best = cost; plan = std::move(cand); have = true; }
        }
    return have;
}

void gates_of(const qsim_circuit *c, std::vector<LGate> &out) {
    out.reserve((size_t)c->count);
    for (long i = 0; i < c->count; i++) {
        const qsim_gate_rec &g = c->gates[i];
        LGate lg{};
        lg.kind = g.kind; lg.q0 = g.q0; lg.q1 = g.q1; lg.idx = i;
        if (g.kind == QSIM_GATE_U1)
            for (int k = 0; k < 4; k++) lg.m[k] = cd(c->mats2[8 * (long)g.mat + 2 * k], c->mats2[8 * (long)g.mat + 2 * k + 1]);
        out.push_back(lg);
    }
}

} // namespace

struct qsim_cluster {
    int n = 0, p = 0, m = 0, P = 0;
    std::vector<int> devices;
    std::vector<qsim_state *> shard;
    std::vector<double2 *> scratch;
    std::vector<int> pos; // logical -> physical after the last run
    uint64_t exchanges = 0;
    double exchange_bytes = 0;       // per shard, summed over exchanges, if every block travelled (the dense figure)
    double exchange_bytes_moved = 0; // what the shards together really sent (blocks of and for empty shards stay home)
    // How blocks travel: every shard on its own device -> RCCL (one ncclGroup of sends + recvs per exchange, on the shard
    // streams); every shard on the SAME device (virtual shards) -> the pack kernel writes its blocks straight into the
    // members' spare buffers and the buffers change roles; anything else -> pack + device-to-device copies.
    std::vector<ncclComm_t> comms;
    bool same_device = false;
    // same_device: the shards' state buffers are slices of ONE allocation and their scratch buffers slices of another, so
    // "block b of member j's new contents" is an index of the scratch pool and the re-layout of an exchange is a permutation
    // of index bits across the whole pool (qsim_flush_pack) — the last tile pass before an exchange writes straight there.
    char *pool[2] = {nullptr, nullptr};
    int state_pool = 0; // which pool the states are in (exchanges flip it)
    // the plan of the last circuit: a loop that runs one circuit again and again plans it once (compared gate by gate, not hashed)
    std::vector<LGate> planned_gates;
    Plan planned;
    int planned_tail = -1;
    uint64_t fused_packs = 0, separate_packs = 0;
    std::vector<hipEvent_t> packed; // per shard: its pack of the current exchange has finished
    // A plan is only right from |0...0>: its first qubit placement is free BECAUSE that state is permutation-symmetric, and its
    // exchanges leave out what is zero from there (Step::mixed_*).  Set by qsim_cluster_reset, cleared when a circuit starts.
    bool fresh = false;
};

static thread_local std::string g_derr;
static int cfail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_derr = buf;
    return code;
}
extern "C" const char *qsim_cluster_error(void) { return g_derr.c_str(); }

extern "C" void qsim_cluster_destroy(qsim_cluster *c) {
    if (!c) return;
    for (size_t r = 0; r < c->shard.size(); r++) (void)qsim_sync(c->shard[r]);
    for (ncclComm_t comm : c->comms) (void)ncclCommDestroy(comm);
    for (size_t r = 0; r < c->shard.size(); r++) {
        (void)hipSetDevice(c->devices[r]);
        if (r < c->packed.size() && c->packed[r]) (void)hipEventDestroy(c->packed[r]);
        if (c->scratch[r] && !c->pool[0]) (void)hipFree(c->scratch[r]);
        qsim_destroy(c->shard[r]);
    }
    for (char *pl : c->pool)
        if (pl) (void)hipFree(pl);
    delete c;
}

extern "C" int qsim_cluster_create(qsim_cluster **out, int num_q, int num_shards, const int *devices) {
    if (!out) return cfail(QSIM_ERR_ARG, "out is NULL");
    *out = nullptr;
    int p = 0;
    while ((1 << p) < num_shards) p++;
    if (num_shards < 1 || (1 << p) != num_shards) return cfail(QSIM_ERR_ARG, "shard count %d is not a power of two", num_shards);
    if (num_q - p < 2 && p > 0) return cfail(QSIM_ERR_ARG, "%d qubits cannot be split over %d shards", num_q, num_shards);
    const int ndev = qsim_device_count();
    if (ndev <= 0) return cfail(QSIM_ERR_DEVICE, "no HIP device available (libqsim has no CPU fallback)");
    qsim_cluster *c = new qsim_cluster();
    c->n = num_q; c->p = p; c->m = num_q - p; c->P = num_shards;
    c->pos.resize(num_q);
    for (int q = 0; q < num_q; q++) c->pos[q] = q;
    bool one_device = p > 0;
    for (int r = 0; r < num_shards; r++) {
        const int dev = devices ? devices[r] : (r % ndev);
        if (dev < 0 || dev >= ndev) { qsim_cluster_destroy(c); return cfail(QSIM_ERR_ARG, "device %d out of range", dev); }
        c->devices.push_back(dev);
        one_device = one_device && dev == c->devices[0];
    }
    if (one_device) { // virtual shards: two pools, see qsim_cluster::pool
        (void)hipSetDevice(c->devices[0]);
        for (int i = 0; i < 2; i++)
            if (hipMalloc((void **)&c->pool[i], (size_t)16 << num_q) != hipSuccess) {
                (void)hipGetLastError();
                qsim_cluster_destroy(c);
                return cfail(QSIM_ERR_ALLOC, "Malloc error");
            }
    }
    for (int r = 0; r < num_shards; r++) {
        const int dev = c->devices[r];
        qsim_state *s = nullptr;
        int rc = c->pool[0] ? qsim_create_external(&s, c->m, dev, c->pool[0] + ((size_t)r * 16 << c->m)) : qsim_create(&s, c->m, dev);
        c->shard.push_back(s);
        c->scratch.push_back(c->pool[1] ? (double2 *)(c->pool[1] + ((size_t)r * 16 << c->m)) : nullptr);
        if (rc == QSIM_OK && p > 0 && !c->pool[0]) {
            (void)hipSetDevice(dev);
            if (hipMalloc((void **)&c->scratch[r], (size_t)16 << c->m) != hipSuccess) rc = QSIM_ERR_ALLOC;
        }
        if (rc != QSIM_OK) {
            const std::string msg = rc == QSIM_ERR_ALLOC ? "Malloc error" : qsim_last_error();
            qsim_cluster_destroy(c);
            return cfail(rc, "shard %d: %s", r, msg.c_str());
        }
    }
    // let every device reach its peers directly where the platform allows it
    for (int a = 0; a < num_shards; a++)
        for (int b = 0; b < num_shards; b++)
            if (c->devices[a] != c->devices[b]) {
                int can = 0;
                (void)hipSetDevice(c->devices[a]);
                if (hipDeviceCanAccessPeer(&can, c->devices[a], c->devices[b]) == hipSuccess && can)
                    (void)hipDeviceEnablePeerAccess(c->devices[b], 0); // "already enabled" is fine
            }
    (void)hipGetLastError();
    if (p > 0) {
        bool same = true, distinct = true;
        for (int a = 0; a < num_shards; a++)
            for (int b = a + 1; b < num_shards; b++) {
                if (c->devices[a] == c->devices[b]) distinct = false;
                else same = false;
            }
        c->same_device = same;
        c->packed.assign((size_t)num_shards, nullptr);
        for (int r = 0; r < num_shards; r++) {
            (void)hipSetDevice(c->devices[r]);
            if (hipEventCreateWithFlags(&c->packed[r], hipEventDisableTiming) != hipSuccess) {
                qsim_cluster_destroy(c);
                return cfail(QSIM_ERR_DEVICE, "event creation failed");
            }
        }
        if (distinct) { // one RCCL communicator per device, all in this process
            c->comms.assign((size_t)num_shards, nullptr);
            const ncclResult_t nr = ncclCommInitAll(c->comms.data(), num_shards, c->devices.data());
            if (nr != ncclSuccess) {
                c->comms.clear();
                qsim_cluster_destroy(c);
                return cfail(QSIM_ERR_DEVICE, "ncclCommInitAll failed: %s", ncclGetErrorString(nr));
            }
            // A shard's scratch is touched by its own stream only in this mode (pack, then the sends): idle during local
            // steps, so it doubles as the second buffer of the shard's out-of-place tile passes (QSIM_OPT_PINGPONG).
            for (int r = 0; r < num_shards; r++) (void)qsim_set_spare_buffer(c->shard[r], c->scratch[r]);
        }
    }
    *out = c;
    return QSIM_OK;
}

extern "C" int qsim_cluster_num_shards(const qsim_cluster *c) { return c ? c->P : -1; }
extern "C" qsim_state *qsim_cluster_shard(qsim_cluster *c, int r) { return (c && r >= 0 && r < c->P) ? c->shard[r] : nullptr; }

extern "C" int qsim_cluster_set_option(qsim_cluster *c, int option, long value) {
    if (!c) return cfail(QSIM_ERR_ARG, "NULL cluster");
    for (qsim_state *s : c->shard) {
        const int rc = qsim_set_option(s, option, value);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

// Resets every shard to its part of |0...0> and the map to the identity.
extern "C" int qsim_cluster_reset(qsim_cluster *c) {
    if (!c) return cfail(QSIM_ERR_ARG, "NULL cluster");
    for (int r = 0; r < c->P; r++) {
        const int rc = qsim_reset_shard(c->shard[r], r == 0);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    for (int q = 0; q < c->n; q++) c->pos[q] = q;
    c->fresh = true;
    return QSIM_OK;
}

// flush: launch every shard's passes now.  The local step in front of an exchange leaves them queued: the exchange flushes
// them itself, so that the last tile pass can do the exchange's re-layout (qsim_flush_pack).
static int apply_local(qsim_cluster *c, const Step &st, bool flush) {
    for (int r = 0; r < c->P; r++) {
        qsim_state *s = c->shard[r];
        for (const LocalOp &o : st.per_shard[r]) {
            int rc;
            if (o.kind == 2) rc = qsim_apply_cx(s, o.a, o.b);
            else if (o.kind == 1) {
                const double U[8] = {o.m[0].real(), o.m[0].imag(), o.m[1].real(), o.m[1].imag(),
                                     o.m[2].real(), o.m[2].imag(), o.m[3].real(), o.m[3].imag()};
                rc = qsim_apply_1q(s, U, o.a);
            } else rc = qsim_scale(s, o.m[0].real(), o.m[0].imag());
            if (rc) return cfail(rc, "%s", qsim_last_error());
        }
        const int rc = flush ? qsim_flush(s) : QSIM_OK; // every shard's passes are in flight before the next one is scheduled
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

// Same device for every shard: shard r's pack writes block j of its new layout straight into the spare buffer of group
// member j (at block position mine(r)), then every shard takes its spare buffer — now complete — as its state.  One
// kernel per shard and no copy stage; ordering is by events between the shard streams, the host never waits.
static int exchange_direct(qsim_cluster *c, const Step &st) {
    const int k = (int)st.J.size();
    const size_t blk_bytes = ((size_t)16 << c->m) >> k;
    std::vector<Roles> roles;
    for (int r = 0; r < c->P; r++) roles.push_back(roles_of(r, c->m, st));
    char *out_pool = c->pool[1 - c->state_pool];
    int to[3] = {0, 0, 0};
    uint32_t jmask = 0;
    for (int j = 0; j < k; j++) { to[j] = c->m + st.J[j]; jmask |= 1u << st.J[j]; }
    for (int r = 0; r < c->P; r++) {
        const Roles &ro = roles[(size_t)r];
        if (!ro.empty_before) {
            // destination of this shard's amplitudes inside the scratch pool: shard id = its own with the J bits replaced by the
            // amplitude's Lsel bits (to[]), block `mine` of that shard, the other local bits closed up below
            const uint64_t konst = ((uint64_t)((uint32_t)r & ~jmask) << c->m) | ((uint64_t)ro.mine << (c->m - k));
            int fused = 0;
            const int rc = qsim_flush_pack(c->shard[r], st.Lsel.data(), k, to, konst, out_pool, st.mixed_local, ro.unread, nullptr, &fused);
            if (rc) return cfail(rc, "%s", qsim_last_error());
            (fused ? c->fused_packs : c->separate_packs)++;
            c->exchange_bytes_moved += (double)blk_bytes * __builtin_popcount(ro.send);
        } else {
            const int rc = qsim_flush(c->shard[r]); // an empty shard: its queue is dropped, nothing to pack
            if (rc) return cfail(rc, "%s", qsim_last_error());
        }
        if (hipEventRecord(c->packed[r], (hipStream_t)qsim_stream(c->shard[r])) != hipSuccess) return cfail(QSIM_ERR_DEVICE, "event record failed");
    }
    // every stream waits for every pack: the members' packs filled this shard's new buffer, and nobody may write into a
    // buffer (next exchange) that a straggler still reads
    for (int r = 0; r < c->P; r++)
        for (int o = 0; o < c->P; o++)
            if (o != r && hipStreamWaitEvent((hipStream_t)qsim_stream(c->shard[r]), c->packed[o], 0) != hipSuccess)
                return cfail(QSIM_ERR_DEVICE, "stream wait failed");
    for (int r = 0; r < c->P; r++) {
        void *buf = c->scratch[r];
        int rc = qsim_swap_buffer(c->shard[r], &buf);
        if (rc == QSIM_OK) rc = settle(c->shard[r], roles[(size_t)r]);
        if (rc) return cfail(rc, "%s", qsim_last_error());
        c->scratch[r] = (double2 *)buf;
    }
    c->state_pool = 1 - c->state_pool;
    return QSIM_OK;
}

// One device per shard: RCCL.  Every shard packs into its own scratch; then ONE group holds, for every shard, the
// 2^k - 1 sends of its scratch blocks and the 2^k - 1 receives into its state buffer, each pair of shards on its own
// xGMI link (k = 1: the pairwise half-shard exchange; k = log2 P: an all-to-all over all P - 1 links at once).
// Everything is stream-ordered on the shard streams (pack -> send/recv -> the next pass); the host does not wait.
static int exchange_rccl(qsim_cluster *c, const Step &st) {
    const int k = (int)st.J.size();
    const size_t blk_bytes = ((size_t)16 << c->m) >> k;
    std::vector<Roles> roles;
    for (int r = 0; r < c->P; r++) roles.push_back(roles_of(r, c->m, st));
    for (int r = 0; r < c->P; r++) {
        int fused = 0;
        const int rc = roles[(size_t)r].empty_before ? qsim_flush(c->shard[r])
                                                     : qsim_flush_pack(c->shard[r], st.Lsel.data(), k, nullptr, 0, c->scratch[r], st.mixed_local, roles[(size_t)r].unread, nullptr, &fused);
        if (rc) return cfail(rc, "%s", qsim_last_error());
        if (!roles[(size_t)r].empty_before) (fused ? c->fused_packs : c->separate_packs)++;
    }
    std::vector<char *> state((size_t)c->P);
    for (int r = 0; r < c->P; r++) state[(size_t)r] = (char *)qsim_state_buffer(c->shard[r]);
    ncclResult_t nr = ncclGroupStart();
    for (int r = 0; r < c->P && nr == ncclSuccess; r++) {
        const Roles &ro = roles[(size_t)r];
        const char *scr = (const char *)c->scratch[r];
        hipStream_t stream = (hipStream_t)qsim_stream(c->shard[r]);
        (void)hipSetDevice(c->devices[r]);
        for (int b = 0; b < (1 << k) && nr == ncclSuccess; b++) {
            if (ro.send >> b & 1u) nr = ncclSend(scr + (size_t)b * blk_bytes, blk_bytes / 8, ncclDouble, ro.members[b], c->comms[r], stream);
            if (nr == ncclSuccess && (ro.recv >> b & 1u))
                nr = ncclRecv(state[(size_t)r] + (size_t)b * blk_bytes, blk_bytes / 8, ncclDouble, ro.members[b], c->comms[r], stream);
        }
        c->exchange_bytes_moved += (double)blk_bytes * __builtin_popcount(ro.send);
    }
    const ncclResult_t ne = ncclGroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return cfail(QSIM_ERR_DEVICE, "RCCL exchange failed: %s", ncclGetErrorString(nr));
    for (int r = 0; r < c->P; r++) { // the block a shard keeps
        const Roles &ro = roles[(size_t)r];
        (void)hipSetDevice(c->devices[r]);
        if (ro.keep_own && hipMemcpyAsync(state[(size_t)r] + (size_t)ro.mine * blk_bytes, (const char *)c->scratch[r] + (size_t)ro.mine * blk_bytes,
                                          blk_bytes, hipMemcpyDeviceToDevice, (hipStream_t)qsim_stream(c->shard[r])) != hipSuccess)
            return cfail(QSIM_ERR_DEVICE, "exchange copy failed");
        const int rc = settle(c->shard[r], ro);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

// Mixed placements (some shards share a device, some do not): pack, then device-to-device copies of the blocks.
static int exchange_copies(qsim_cluster *c, const Step &st) {
    const int k = (int)st.J.size();
    const size_t blk_bytes = ((size_t)16 << c->m) >> k;
    for (int r = 0; r < c->P; r++) {
        const int rc = qsim_pack_bits(c->shard[r], st.Lsel.data(), k, c->scratch[r]);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    for (int r = 0; r < c->P; r++) {
        const int rc = qsim_sync(c->shard[r]);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    // state block b of shard r  <-  scratch block mine(r) of group member b
    for (int r = 0; r < c->P; r++) {
        int mine;
        std::vector<int> members;
        peers_of(r, st.J, mine, members);
        char *dst = (char *)qsim_device_ptr(c->shard[r]);
        hipStream_t stream = (hipStream_t)qsim_stream(c->shard[r]);
        (void)hipSetDevice(c->devices[r]);
        for (int b = 0; b < (1 << k); b++) {
            const int peer = members[b];
            const char *src = (const char *)c->scratch[peer] + (size_t)mine * blk_bytes;
            hipError_t e;
            if (c->devices[peer] == c->devices[r])
                e = hipMemcpyAsync(dst + (size_t)b * blk_bytes, src, blk_bytes, hipMemcpyDeviceToDevice, stream);
            else
                e = hipMemcpyPeerAsync(dst + (size_t)b * blk_bytes, c->devices[r], src, c->devices[peer], blk_bytes, stream);
            if (e != hipSuccess) return cfail(QSIM_ERR_DEVICE, "exchange copy failed: %s", hipGetErrorString(e));
        }
    }
    for (int r = 0; r < c->P; r++) {
        const int rc = qsim_sync(c->shard[r]);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

static int exchange(qsim_cluster *c, const Step &st) {
    const int k = (int)st.J.size();
    int rc;
    if (k > kMaxRoleBits && !c->comms.empty()) return cfail(QSIM_ERR_ARG, "exchange of %d qubits: groups of more than %d shards are not supported on RCCL", k, 1 << kMaxRoleBits);
    if (!c->comms.empty()) rc = exchange_rccl(c, st);
    else if (c->same_device && k <= 3) rc = exchange_direct(c, st);
    else rc = exchange_copies(c, st);
    if (rc) return rc;
    c->exchanges++;
    c->exchange_bytes += (double)(((size_t)16 << c->m) >> k) * ((1 << k) - 1);
    if (c->comms.empty() && !(c->same_device && k <= 3)) c->exchange_bytes_moved += (double)(((size_t)16 << c->m) >> k) * ((1 << k) - 1) * c->P;
    return QSIM_OK;
}

extern "C" const char *qsim_cluster_exchange_mode(const qsim_cluster *c) {
    if (!c || c->p == 0) return "none";
    return !c->comms.empty() ? "rccl" : c->same_device ? "direct" : "copies";
}

// Plans and runs ONE circuit from |0...0> (compute_state_vector semantics, quantum_simulator.c:115-254: one circuit per
// state): qsim_cluster_reset must come first.  The plan depends on it twice — the first qubit placement moves no data because
// |0...0> is permutation-symmetric, and the exchanges neither send nor read what is still zero (Step::mixed_*, roles_of) — so
// a second circuit on top of the first one's result, or on a state the caller wrote through qsim_cluster_shard(), is refused
// instead of silently dropping amplitudes.
static int cluster_plan_for(qsim_cluster *c, const qsim_circuit *circ);
static int plan_shard_steps(const Plan &plan, int shard, qsim_state *s, int max_candidates, double budget_ms, qsim_tune_report *total);

extern "C" int qsim_cluster_run_circuit(qsim_cluster *c, const qsim_circuit *circ) {
    if (!c || !circ) return cfail(QSIM_ERR_ARG, "NULL argument");
    if (circ->num_q != c->n) return cfail(QSIM_ERR_ARG, "circuit has %d qubits, cluster has %d", circ->num_q, c->n);
    if (!c->fresh) return cfail(QSIM_ERR_ARG, "cluster does not hold |0...0>: qsim_cluster_run_circuit runs one circuit per reset (call qsim_cluster_reset first)");
    for (int q = 0; q < c->n; q++)
        if (c->pos[q] != q) return cfail(QSIM_ERR_ARG, "cluster already holds a permuted state: reset it first");
    c->fresh = false;
    {
        const int rc = cluster_plan_for(c, circ);
        if (rc) return rc;
    }
    const Plan &plan = c->planned;
    for (size_t i = 0; i < plan.steps.size(); i++) {
        const Step &st = plan.steps[i];
        const bool before_exchange = i + 1 < plan.steps.size() && plan.steps[i + 1].exchange;
        const int rc = st.exchange ? exchange(c, st) : apply_local(c, st, !before_exchange);
        if (rc) return rc;
    }
    c->pos = plan.final_pos;
    return QSIM_OK;
}

// One shard's ops of a local step as a circuit on its m local qubits (a per-shard factor as diag(z, z) on qubit 0: qsim_scale).
static int step_circuit(const Step &st, int shard, int m, qsim_circuit **out) {
    qsim_circuit *c = nullptr;
    int rc = qsim_circuit_create(m, &c);
    for (const LocalOp &o : st.per_shard[(size_t)shard]) {
        if (rc) break;
        if (o.kind == 2) rc = qsim_circuit_append_cx(c, o.a, o.b);
        else {
            const cd z = o.m[0];
            const double U[8] = {o.m[0].real(), o.m[0].imag(), o.kind == 1 ? o.m[1].real() : 0.0, o.kind == 1 ? o.m[1].imag() : 0.0,
                                 o.kind == 1 ? o.m[2].real() : 0.0, o.kind == 1 ? o.m[2].imag() : 0.0,
                                 o.kind == 1 ? o.m[3].real() : z.real(), o.kind == 1 ? o.m[3].imag() : z.imag()};
            rc = qsim_circuit_append_1q(c, U, o.kind == 1 ? o.a : 0);
        }
    }
    if (rc) { qsim_circuit_free(c); return rc; }
    *out = c;
    return QSIM_OK;
}

// Planning of one shard's local steps: for each, the support the shard will have there (0 at the start on the shard that holds
// index 0; after an exchange what roles_of says; a shard that holds nothing is skipped) and the schedule choice / geometry
// tuning for exactly that situation.
static int plan_shard_steps(const Plan &plan, int shard, qsim_state *s, int max_candidates, double budget_ms, qsim_tune_report *total) {
    const uint64_t all = ~0ULL;
    uint64_t support = 0;
    bool empty = shard != 0;
    int locals = 0;
    for (const Step &st : plan.steps) locals += !st.exchange;
    for (const Step &st : plan.steps) {
        if (st.exchange) {
            const Roles ro = roles_of(shard, plan.m, st);
            empty = ro.empty_after;
            support = ro.new_support;
            continue;
        }
        if (empty) continue;
        qsim_circuit *c = nullptr;
        int rc = step_circuit(st, shard, plan.m, &c);
        qsim_tune_report r{};
        if (rc == QSIM_OK) {
            if (max_candidates > 1) rc = qsim_tune_circuit_support(s, c, max_candidates, budget_ms > 0 ? budget_ms / locals : 0.0, &r, support);
            else rc = qsim_choose_schedule_for(s, c, support);
        }
        qsim_circuit_free(c);
        if (rc) return cfail(rc, "%s", qsim_last_error());
        if (total) {
            total->tile_passes += r.tile_passes; total->already_known += r.already_known; total->passes_tuned += r.passes_tuned;
            total->passes_reordered += r.passes_reordered; total->candidates_timed += r.candidates_timed;
            total->ms_ascending += r.ms_ascending; total->ms_best += r.ms_best; total->seconds += r.seconds;
        }
        support = all; // a local step leaves the shard dense (its passes cover every qubit, or nearly: the engine knows better, the key then simply misses)
    }
    return QSIM_OK;
}

static int cluster_plan_for(qsim_cluster *c, const qsim_circuit *circ) {
    if (circ->num_q != c->n) return cfail(QSIM_ERR_ARG, "circuit has %d qubits, cluster has %d", circ->num_q, c->n);
    for (long i = 0; i < circ->count; i++)
        if (circ->gates[i].kind == QSIM_GATE_U2) return cfail(QSIM_ERR_ARG, "generic 2-qubit gates are not supported on clusters");
    std::vector<LGate> gates;
    gates_of(circ, gates);
    bool same = c->planned_tail == tail_limit() && gates.size() == c->planned_gates.size();
    for (size_t i = 0; same && i < gates.size(); i++) {
        const LGate &a = gates[i], &b = c->planned_gates[i];
        same = a.kind == b.kind && a.q0 == b.q0 && a.q1 == b.q1 && (a.kind != QSIM_GATE_U1 || memcmp(a.m, b.m, sizeof a.m) == 0);
    }
    if (!same) {
        c->planned = Plan();
        c->planned_tail = -1;
        if (!build_plan(c->n, c->p, gates, c->planned)) return cfail(QSIM_ERR_ARG, "planner made no progress");
        c->planned_gates.swap(gates);
        c->planned_tail = tail_limit();
    }
    return QSIM_OK;
}

// Schedule choice (and, with max_candidates > 1, timing) for every shard's every local step, step by step for all shards, so
// that each exchange can be told what its senders' engines will actually have written by then: a schedule chosen here may
// leave other qubits untouched than the default one the planner assumed, and the last tile pass in front of an exchange does
// the re-layout itself only when what it writes covers what the receivers look at (Step::mixed_local).  Both masks bound
// the same state, so their intersection does too; holders and supports after the exchange follow from it.
extern "C" int qsim_cluster_plan(qsim_cluster *c, const qsim_circuit *circ, int max_candidates, double budget_ms) {
    if (!c || !circ) return cfail(QSIM_ERR_ARG, "NULL argument");
    int rc = cluster_plan_for(c, circ);
    if (rc) return rc;
    Plan &plan = c->planned;
    int locals = 0;
    for (const Step &st : plan.steps) locals += !st.exchange;
    const double budget_each = budget_ms > 0 && locals > 0 ? budget_ms / c->P / locals : 0.0;
    std::vector<uint64_t> sup((size_t)c->P, 0);
    std::vector<char> holds((size_t)c->P, 0);
    holds[0] = 1;
    for (Step &st : plan.steps) {
        if (st.exchange) {
            uint64_t written = 0, ranks = 0;
            for (int r = 0; r < c->P; r++)
                if (holds[(size_t)r]) { written |= sup[(size_t)r]; ranks |= (uint64_t)r; }
            st.mixed_local &= written;
            st.mixed_rank &= ranks;
            for (int r = 0; r < c->P; r++) {
                const Roles ro = roles_of(r, plan.m, st);
                holds[(size_t)r] = !ro.empty_after;
                sup[(size_t)r] = ro.empty_after ? 0 : ro.new_support;
            }
            continue;
        }
        for (int r = 0; r < c->P; r++) {
            if (!holds[(size_t)r]) continue;
            qsim_circuit *sc = nullptr;
            rc = step_circuit(st, r, plan.m, &sc);
            if (rc == QSIM_OK) {
                qsim_tune_report rep{};
                if (max_candidates > 1) rc = qsim_tune_circuit_support(c->shard[r], sc, max_candidates, budget_each, &rep, sup[(size_t)r]);
                else rc = qsim_choose_schedule_for(c->shard[r], sc, sup[(size_t)r]);
            }
            if (rc == QSIM_OK) rc = qsim_support_after(c->shard[r], sc, sup[(size_t)r], &sup[(size_t)r]);
            qsim_circuit_free(sc);
            if (rc) return cfail(rc, "%s", qsim_last_error());
        }
    }
    return qsim_cluster_reset(c);
}

extern "C" int qsim_cluster_sync(qsim_cluster *c) {
    if (!c) return cfail(QSIM_ERR_ARG, "NULL cluster");
    for (qsim_state *s : c->shard) {
        const int rc = qsim_sync(s);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

static uint64_t physical_index(const qsim_cluster *c, uint64_t logical) {
    uint64_t out = 0;
    for (int q = 0; q < c->n; q++) out |= ((logical >> q) & 1ULL) << c->pos[q];
    return out;
}

// Amplitudes by LOGICAL basis index (gathered one by one: meant for samples and small registers).
extern "C" int qsim_cluster_read(qsim_cluster *c, uint64_t first, uint64_t count, double *out) {
    if (!c || !out) return cfail(QSIM_ERR_ARG, "NULL argument");
    const uint64_t N = 1ULL << c->n;
    if (first > N || count > N - first) return cfail(QSIM_ERR_ARG, "read range outside the state");
    const uint64_t mmask = (1ULL << c->m) - 1ULL;
    if (c->p == 0 || count > 4096) { // bulk: pull whole shards once and permute on the host
        std::vector<std::vector<double>> host(c->P);
        for (int r = 0; r < c->P; r++) {
            host[r].resize((size_t)2 << c->m);
            const int rc = qsim_read(c->shard[r], 0, 1ULL << c->m, host[r].data());
            if (rc) return cfail(rc, "%s", qsim_last_error());
        }
        for (uint64_t i = 0; i < count; i++) {
            const uint64_t ph = physical_index(c, first + i);
            out[2 * i] = host[ph >> c->m][2 * (ph & mmask)];
            out[2 * i + 1] = host[ph >> c->m][2 * (ph & mmask) + 1];
        }
        return QSIM_OK;
    }
    for (uint64_t i = 0; i < count; i++) {
        const uint64_t ph = physical_index(c, first + i);
        const int rc = qsim_read(c->shard[ph >> c->m], ph & mmask, 1, out + 2 * i);
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

// measurement() of quantum_simulator.c:270-283 on a sharded state, in LOGICAL index order (the order the reference's
// cumulative distribution runs in, whatever the qubit map of the last run left behind).  A logical block = the 2^12
// amplitudes that agree on logical bits >= 12.  On shard r it occupies the local positions holding logical bits < 12
// (lo_mask), at the base given by the local positions holding logical bits >= 12 (hi_mask); logical bits sitting on
// rank-id positions are fixed by r.  So every shard sums |a|^2 per block ON ITS DEVICE (qsim_block_prob_masked), the
// host adds the "P partial sums" of SURVEY 8f row 1 in shard order, and a draw fetches only its own block
// (qsim_gather_masked from the shards that hold a part of it).  Nothing else crosses PCIe: 2^(n-12) doubles per shard
// plus 64 KiB per distinct block drawn.
extern "C" int qsim_cluster_sample(qsim_cluster *c, const double *randoms, long shots, uint64_t *out) {
    if (!c || (shots > 0 && (!randoms || !out))) return cfail(QSIM_ERR_ARG, "NULL argument");
    constexpr int kBlockBits = 12;
    const int bb = c->n < kBlockBits ? c->n : kBlockBits;
    const uint64_t N = 1ULL << c->n, nblocks = N >> bb, bsize = 1ULL << bb;
    std::vector<int> inv(c->n); // physical bit -> logical qubit
    for (int q = 0; q < c->n; q++) inv[c->pos[q]] = q;
    uint64_t hi_mask = 0, lo_mask = 0; // local positions by the kind of logical bit they hold
    for (int b = 0; b < c->m; b++) (inv[b] >= bb ? hi_mask : lo_mask) |= 1ULL << b;
    std::vector<int> hi_pos, lo_pos; // ascending local positions = the order deposit() fills them in
    for (int b = 0; b < c->m; b++) (inv[b] >= bb ? hi_pos : lo_pos).push_back(b);
    // block id / in-block index contributed by the rank-id bits of shard r
    auto rank_part = [&](int r, uint64_t &blk_bits, uint64_t &in_bits) {
        blk_bits = in_bits = 0;
        for (int g = 0; g < c->p; g++) {
            const int L = inv[c->m + g];
            if ((r >> g) & 1) (L >= bb ? blk_bits : in_bits) |= 1ULL << (L >= bb ? L - bb : L);
        }
    };
    std::vector<double> prefix(nblocks, 0.0), part((size_t)1 << hi_pos.size());
    for (int r = 0; r < c->P; r++) {
        const int rc = qsim_block_prob_masked(c->shard[r], hi_mask, lo_mask, part.data());
        if (rc) return cfail(rc, "%s", qsim_last_error());
        uint64_t rb, ri;
        rank_part(r, rb, ri);
        for (uint64_t w = 0; w < part.size(); w++) {
            uint64_t blk = rb;
            for (size_t j = 0; j < hi_pos.size(); j++)
                if ((w >> j) & 1ULL) blk |= 1ULL << (inv[hi_pos[j]] - bb);
            prefix[blk] += part[w];
        }
    }
    double acc = 0.0;
    for (uint64_t b = 0; b < nblocks; b++) { acc += prefix[b]; prefix[b] = acc; } // cumulative at the END of block b
    std::vector<double> blk(2 * bsize), piece((size_t)2 << lo_pos.size());
    uint64_t cached = ~0ULL;
    auto fetch_block = [&](uint64_t b) -> int {
        for (int r = 0; r < c->P; r++) {
            uint64_t rb, ri;
            rank_part(r, rb, ri);
            uint64_t gmask = 0; // block-id bits decided by rank-id positions
            for (int g = 0; g < c->p; g++)
                if (inv[c->m + g] >= bb) gmask |= 1ULL << (inv[c->m + g] - bb);
            if ((b & gmask) != rb) continue; // this shard holds no part of block b
            uint64_t base = 0;
            for (size_t j = 0; j < hi_pos.size(); j++)
                if ((b >> (inv[hi_pos[j]] - bb)) & 1ULL) base |= 1ULL << hi_pos[j];
            const int rc = qsim_gather_masked(c->shard[r], base, lo_mask, piece.data());
            if (rc) return cfail(rc, "%s", qsim_last_error());
            for (uint64_t i = 0; i < ((uint64_t)1 << lo_pos.size()); i++) {
                uint64_t in = ri;
                for (size_t j = 0; j < lo_pos.size(); j++)
                    if ((i >> j) & 1ULL) in |= 1ULL << inv[lo_pos[j]];
                blk[2 * in] = piece[2 * i];
                blk[2 * in + 1] = piece[2 * i + 1];
            }
        }
        return QSIM_OK;
    };
    for (long k = 0; k < shots; k++) {
        const double rnd = randoms[k];
        uint64_t lo = 0, hi = nblocks;
        while (lo < hi) { // first block whose end value is non-zero and >= r (quantum_simulator.c:279)
            const uint64_t mid = (lo + hi) >> 1;
            if (prefix[mid] == 0.0 || prefix[mid] < rnd) lo = mid + 1;
            else hi = mid;
        }
        uint64_t idx = N - 1;
        bool found = false;
        for (uint64_t b = lo; b < nblocks && !found; b++) {
            if (b != cached) {
                const int rc = fetch_block(b);
                if (rc) return rc;
                cached = b;
            }
            double cum = b ? prefix[b - 1] : 0.0;
            for (uint64_t i = 0; i < bsize; i++) {
                cum += blk[2 * i] * blk[2 * i] + blk[2 * i + 1] * blk[2 * i + 1];
                if (!(cum == 0.0 || cum < rnd)) { idx = b * bsize + i; found = true; break; }
            }
        }
        out[k] = idx;
    }
    return QSIM_OK;
}

extern "C" int qsim_cluster_norm2(qsim_cluster *c, double *out) {
    if (!c || !out) return cfail(QSIM_ERR_ARG, "NULL argument");
    double tot = 0;
    for (qsim_state *s : c->shard) {
        double v = 0;
        const int rc = qsim_norm2(s, &v);
        if (rc) return cfail(rc, "%s", qsim_last_error());
        tot += v;
    }
    *out = tot;
    return QSIM_OK;
}

extern "C" int qsim_cluster_exchange_stats(const qsim_cluster *c, uint64_t *exchanges, double *bytes_per_shard) {
    if (!c) return QSIM_ERR_ARG;
    if (exchanges) *exchanges = c->exchanges;
    if (bytes_per_shard) *bytes_per_shard = c->exchange_bytes;
    return QSIM_OK;
}
extern "C" int qsim_cluster_exchange_bytes_moved(const qsim_cluster *c, double *bytes_all_shards) {
    if (!c || !bytes_all_shards) return QSIM_ERR_ARG;
    *bytes_all_shards = c->exchange_bytes_moved;
    return QSIM_OK;
}
extern "C" int qsim_cluster_pack_counts(const qsim_cluster *c, uint64_t *fused, uint64_t *separate) {
    if (!c) return QSIM_ERR_ARG;
    if (fused) *fused = c->fused_packs;
    if (separate) *separate = c->separate_packs;
    return QSIM_OK;
}

// ---- the plan as an object (host only): what distributed.py's one-process-per-GPU driver executes -------------
struct qsim_shard_plan {
    Plan plan;
    int P = 0;
};

extern "C" int qsim_shard_plan_create(qsim_shard_plan **out, const qsim_circuit *circ, int num_shards) {
    if (!out || !circ) return cfail(QSIM_ERR_ARG, "NULL argument");
    *out = nullptr;
    int p = 0;
    while ((1 << p) < num_shards) p++;
    if (num_shards < 1 || (1 << p) != num_shards) return cfail(QSIM_ERR_ARG, "shard count %d is not a power of two", num_shards);
    if (p > 0 && circ->num_q - p < 2) return cfail(QSIM_ERR_ARG, "%d qubits cannot be split over %d shards", circ->num_q, num_shards);
    for (long i = 0; i < circ->count; i++)
        if (circ->gates[i].kind == QSIM_GATE_U2) return cfail(QSIM_ERR_ARG, "generic 2-qubit gates are not supported on shards");
    std::vector<LGate> gates;
    gates_of(circ, gates);
    qsim_shard_plan *sp = new qsim_shard_plan();
    sp->P = num_shards;
    if (!build_plan(circ->num_q, p, gates, sp->plan)) {
        delete sp;
        return cfail(QSIM_ERR_ARG, "planner made no progress");
    }
    *out = sp;
    return QSIM_OK;
}

extern "C" void qsim_shard_plan_free(qsim_shard_plan *p) { delete p; }
extern "C" int qsim_shard_plan_num_steps(const qsim_shard_plan *p) { return p ? (int)p->plan.steps.size() : -1; }

extern "C" int qsim_shard_plan_step(const qsim_shard_plan *p, int step, int *kind, int *k, int *shard_bits, int *local_bits) {
    if (!p || step < 0 || step >= (int)p->plan.steps.size()) return QSIM_ERR_ARG;
    const Step &st = p->plan.steps[step];
    if (kind) *kind = st.exchange ? 1 : 0;
    if (k) *k = (int)st.J.size();
    for (size_t i = 0; i < st.J.size(); i++) {
        if (shard_bits) shard_bits[i] = st.J[i];
        if (local_bits) local_bits[i] = st.Lsel[i];
    }
    return QSIM_OK;
}

extern "C" int qsim_shard_plan_step_support(const qsim_shard_plan *p, int step, uint64_t *mixed_local, uint64_t *mixed_rank) {
    if (!p || step < 0 || step >= (int)p->plan.steps.size() || !p->plan.steps[(size_t)step].exchange) return QSIM_ERR_ARG;
    if (mixed_local) *mixed_local = p->plan.steps[(size_t)step].mixed_local;
    if (mixed_rank) *mixed_rank = p->plan.steps[(size_t)step].mixed_rank;
    return QSIM_OK;
}

extern "C" int qsim_shard_plan_exchange_roles(const qsim_shard_plan *p, int step, int shard, qsim_exchange_roles *out) {
    if (!p || !out || step < 0 || step >= (int)p->plan.steps.size() || shard < 0 || shard >= p->P || !p->plan.steps[(size_t)step].exchange) return QSIM_ERR_ARG;
    const Roles r = roles_of(shard, p->plan.m, p->plan.steps[(size_t)step]);
    out->mine = r.mine; out->empty_before = r.empty_before; out->empty_after = r.empty_after; out->keep_own = r.keep_own;
    out->send = r.send; out->recv = r.recv; out->unread = r.unread; out->new_support = r.new_support;
    return QSIM_OK;
}

extern "C" int qsim_shard_plan_final_pos(const qsim_shard_plan *p, int *pos) {
    if (!p || !pos) return QSIM_ERR_ARG;
    for (size_t q = 0; q < p->plan.final_pos.size(); q++) pos[q] = p->plan.final_pos[q];
    return QSIM_OK;
}

extern "C" int qsim_shard_plan_local_ops(const qsim_shard_plan *p, int step, int shard, qsim_local_op_cb cb, void *user) {
    if (!p || !cb || step < 0 || step >= (int)p->plan.steps.size() || shard < 0 || shard >= p->P) return QSIM_ERR_ARG;
    const Step &st = p->plan.steps[step];
    if (st.exchange) return QSIM_ERR_ARG;
    for (const LocalOp &o : st.per_shard[shard]) {
        const double m[8] = {o.m[0].real(), o.m[0].imag(), o.m[1].real(), o.m[1].imag(),
                             o.m[2].real(), o.m[2].imag(), o.m[3].real(), o.m[3].imag()};
        cb(user, o.kind, o.a, o.b, m);
    }
    return QSIM_OK;
}

extern "C" int qsim_shard_plan_apply_local(const qsim_shard_plan *p, int step, int shard, qsim_state *s) {
    if (!p || !s || step < 0 || step >= (int)p->plan.steps.size() || shard < 0 || shard >= p->P) return cfail(QSIM_ERR_ARG, "bad argument");
    const Step &st = p->plan.steps[step];
    if (st.exchange) return cfail(QSIM_ERR_ARG, "step %d is an exchange", step);
    for (const LocalOp &o : st.per_shard[shard]) {
        int rc;
        if (o.kind == 2) rc = qsim_apply_cx(s, o.a, o.b);
        else if (o.kind == 1) {
            const double U[8] = {o.m[0].real(), o.m[0].imag(), o.m[1].real(), o.m[1].imag(),
                                 o.m[2].real(), o.m[2].imag(), o.m[3].real(), o.m[3].imag()};
            rc = qsim_apply_1q(s, U, o.a);
        } else rc = qsim_scale(s, o.m[0].real(), o.m[0].imag());
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

// Predicted exchange cost of the plan on one fully connected xGMI node: bytes each rank sends, and the time of the
// exchanges alone (pack pass + the largest per-link transfer; a k-qubit swap puts 2^-k of the shard on each of 2^k - 1
// links, both directions at once).  link_gbps is per link and direction, pack_gbps the pack kernel's HBM rate.
extern "C" int qsim_shard_plan_predict(const qsim_shard_plan *p, double link_gbps, double pack_gbps, double *bytes_per_rank, double *seconds) {
    if (!p || link_gbps <= 0 || pack_gbps <= 0) return QSIM_ERR_ARG;
    const double S = 16.0 * (double)(1ULL << p->plan.m);
    double bytes = 0, secs = 0;
    for (const Step &st : p->plan.steps) {
        if (!st.exchange) continue;
        const int k = (int)st.J.size();
        const double blk = S / (double)(1 << k);
        bytes += blk * ((1 << k) - 1);
        secs += 2.0 * S / (pack_gbps * 1e9) + blk / (link_gbps * 1e9);
    }
    if (bytes_per_rank) *bytes_per_rank = bytes;
    if (seconds) *seconds = secs;
    return QSIM_OK;
}

// ---- one process per GPU: this rank's end of the exchanges, on RCCL -------------------------------------------------
// The launcher's own channel (torch.distributed, MPI, a file) only carries the 128-byte RCCL id from rank 0 to the
// others; every byte of state travels through ncclSend / ncclRecv issued here, on the shard's own stream, behind the
// pack kernel and in front of the next pass — no host synchronisation inside an exchange.
struct qsim_rank_comm {
    ncclComm_t comm = nullptr;
    qsim_state *shard = nullptr;
    int world = 0, rank = 0, device = 0;
    void *scratch = nullptr;
    bool owns_scratch = false;
    uint64_t exchanges = 0, fused_packs = 0, separate_packs = 0;
    double bytes_sent = 0, ms = 0;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> timing; // start/stop of exchanges not yet resolved
};

static_assert(sizeof(ncclUniqueId) == QSIM_RCCL_ID_BYTES, "QSIM_RCCL_ID_BYTES must equal sizeof(ncclUniqueId)");

extern "C" int qsim_rccl_unique_id(void *id) {
    if (!id) return cfail(QSIM_ERR_ARG, "NULL argument");
    ncclUniqueId uid;
    const ncclResult_t nr = ncclGetUniqueId(&uid);
    if (nr != ncclSuccess) return cfail(QSIM_ERR_DEVICE, "ncclGetUniqueId failed: %s", ncclGetErrorString(nr));
    memcpy(id, &uid, sizeof uid);
    return QSIM_OK;
}

extern "C" void qsim_rank_comm_destroy(qsim_rank_comm *c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    if (c->shard) (void)qsim_sync(c->shard);
    for (auto &pr : c->timing) { (void)hipEventDestroy(pr.first); (void)hipEventDestroy(pr.second); }
    if (c->comm) (void)ncclCommDestroy(c->comm);
    if (c->owns_scratch && c->scratch) (void)hipFree(c->scratch);
    delete c;
}

extern "C" int qsim_rank_comm_create(qsim_rank_comm **out, qsim_state *shard, int device, int world, int rank, const void *id, void *scratch) {
    if (!out || !shard || !id) return cfail(QSIM_ERR_ARG, "NULL argument");
    *out = nullptr;
    if (world < 1 || (world & (world - 1)) || rank < 0 || rank >= world) return cfail(QSIM_ERR_ARG, "bad world size / rank (%d, %d)", world, rank);
    if (qsim_precision_bits(shard) != 64) return cfail(QSIM_ERR_ARG, "sharded states are fp64");
    qsim_rank_comm *c = new qsim_rank_comm();
    c->shard = shard; c->world = world; c->rank = rank; c->device = device;
    if (hipSetDevice(device) != hipSuccess) { delete c; return cfail(QSIM_ERR_DEVICE, "hipSetDevice(%d) failed", device); }
    if (scratch) c->scratch = scratch;
    else {
        if (hipMalloc(&c->scratch, (size_t)16 << qsim_num_qubits(shard)) != hipSuccess) { delete c; return cfail(QSIM_ERR_ALLOC, "Malloc error"); }
        c->owns_scratch = true;
    }
    (void)qsim_set_spare_buffer(shard, c->scratch); // idle between exchanges: the second buffer of the shard's out-of-place passes
    ncclUniqueId uid;
    memcpy(&uid, id, sizeof uid);
    const ncclResult_t nr = ncclCommInitRank(&c->comm, world, uid, rank);
    if (nr != ncclSuccess) {
        c->comm = nullptr;
        qsim_rank_comm_destroy(c);
        return cfail(QSIM_ERR_DEVICE, "ncclCommInitRank failed: %s", ncclGetErrorString(nr));
    }
    *out = c;
    return QSIM_OK;
}

// Swaps k rank-id bits (st.J, ascending) with k local bits (st.Lsel, ascending) of this rank's shard; st.mixed_* say where the
// register can be non-zero (all ones: anywhere), see roles_of.
static int rank_exchange(qsim_rank_comm *c, const Step &st) {
    const int k = (int)st.J.size();
    const int m = qsim_num_qubits(c->shard);
    if (k < 1 || k > m || k > kMaxRoleBits || (1 << k) > c->world) return cfail(QSIM_ERR_ARG, "exchange of %d qubits unsupported here (at most %d: groups of %d ranks)", k, kMaxRoleBits, 1 << kMaxRoleBits);
    for (int j : st.J)
        if (j < 0 || (1 << j) >= c->world) return cfail(QSIM_ERR_ARG, "rank bit %d outside the world", j);
    const Roles ro = roles_of(c->rank, m, st);
    // The plan says this rank holds nothing here; that is only true on a run from |0...0> (qsim_reset_shard, then the plan's
    // steps in order).  A shard that was written since would silently lose its amplitudes: refuse.
    if (ro.empty_before && !qsim_holds_nothing(c->shard))
        return cfail(QSIM_ERR_ARG, "the plan's exchange assumes a run from |0...0> (this rank should hold nothing here and does): reset the shards, then run the plan's steps in order");
    const size_t blk_bytes = ((size_t)16 << m) >> k;
    if (hipSetDevice(c->device) != hipSuccess) return cfail(QSIM_ERR_DEVICE, "hipSetDevice failed");
    hipStream_t stream = (hipStream_t)qsim_stream(c->shard);
    // exchanges are timed (HIP events on the shard's stream) only while the shard is in profile mode, and never more
    // than a bounded number of them stay unresolved: a long-running program that never asks for the statistics must
    // not collect events
    const bool timed = qsim_get_option(c->shard, QSIM_OPT_PROFILE) != 0 && c->timing.size() < 4096;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    if (timed && (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess)) return cfail(QSIM_ERR_DEVICE, "event creation failed");
    // Everything queued so far belongs in front of the exchange, and the last tile pass of it writes the state straight into
    // the packed layout where it can (qsim_flush_pack): the exchange then costs no sweep of its own.  (The timing below
    // therefore starts behind that pass: what it measures is the transfer, plus the pack kernel when one had to run.)
    int rc, fused = 0;
    if (ro.empty_before) rc = qsim_flush(c->shard);
    else rc = qsim_flush_pack(c->shard, st.Lsel.data(), k, nullptr, 0, c->scratch, st.mixed_local, ro.unread, nullptr, &fused);
    if (rc) return cfail(rc, "%s", qsim_last_error());
    if (!ro.empty_before) (fused ? c->fused_packs : c->separate_packs)++;
    if (timed) (void)hipEventRecord(e0, stream);
    char *state = (char *)qsim_state_buffer(c->shard);
    const char *scr = (const char *)c->scratch;
    if (ro.send | ro.recv) {
        ncclResult_t nr = ncclGroupStart();
        for (int b = 0; b < (1 << k) && nr == ncclSuccess; b++) {
            if (ro.send >> b & 1u) nr = ncclSend(scr + (size_t)b * blk_bytes, blk_bytes / 8, ncclDouble, ro.members[b], c->comm, stream);
            if (nr == ncclSuccess && (ro.recv >> b & 1u)) nr = ncclRecv(state + (size_t)b * blk_bytes, blk_bytes / 8, ncclDouble, ro.members[b], c->comm, stream);
        }
        const ncclResult_t ne = ncclGroupEnd();
        if (nr == ncclSuccess) nr = ne;
        if (nr != ncclSuccess) return cfail(QSIM_ERR_DEVICE, "RCCL exchange failed: %s", ncclGetErrorString(nr));
    }
    if (ro.keep_own && hipMemcpyAsync(state + (size_t)ro.mine * blk_bytes, scr + (size_t)ro.mine * blk_bytes, blk_bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess)
        return cfail(QSIM_ERR_DEVICE, "exchange copy failed");
    rc = settle(c->shard, ro);
    if (rc) return cfail(rc, "%s", qsim_last_error());
    if (timed) {
        (void)hipEventRecord(e1, stream);
        c->timing.emplace_back(e0, e1);
    }
    c->exchanges++;
    c->bytes_sent += (double)blk_bytes * __builtin_popcount(ro.send);
    return QSIM_OK;
}

extern "C" int qsim_rank_comm_exchange(qsim_rank_comm *c, const int *shard_bits, const int *local_bits, int k) {
    if (!c || !shard_bits || !local_bits) return cfail(QSIM_ERR_ARG, "NULL argument");
    if (k < 1 || k > kMaxRoleBits) return cfail(QSIM_ERR_ARG, "exchange of %d qubits unsupported here (at most %d)", k, kMaxRoleBits);
    Step st;
    st.exchange = true;
    st.J.assign(shard_bits, shard_bits + k);
    st.Lsel.assign(local_bits, local_bits + k);
    st.mixed_local = st.mixed_rank = ~0ULL; // nothing is known about the contents: every block travels
    return rank_exchange(c, st);
}

// The exchange of one step of a plan, with what the plan knows about the state at that point: a run starts from |0...0>,
// so early exchanges involve shards that hold nothing and blocks that are zero throughout (Step::mixed_*); those neither
// travel nor get written, and the shard goes on visiting only the part of itself that can be non-zero.
extern "C" int qsim_rank_comm_exchange_step(qsim_rank_comm *c, const qsim_shard_plan *p, int step) {
    if (!c || !p || step < 0 || step >= (int)p->plan.steps.size()) return cfail(QSIM_ERR_ARG, "bad argument");
    const Step &st = p->plan.steps[(size_t)step];
    if (!st.exchange) return cfail(QSIM_ERR_ARG, "step %d is not an exchange", step);
    if (p->P != c->world) return cfail(QSIM_ERR_ARG, "plan for %d shards, communicator of %d ranks", p->P, c->world);
    return rank_exchange(c, st);
}

// Exchanges so far, bytes this rank sent, and the seconds its stream spent in them (pack + send/recv, HIP events on the
// shard's stream; waits for the stream).  reset != 0 clears the counters afterwards.
extern "C" int qsim_rank_comm_stats(qsim_rank_comm *c, uint64_t *exchanges, double *bytes_sent, double *seconds, int reset) {
    if (!c) return cfail(QSIM_ERR_ARG, "NULL argument");
    if (!c->timing.empty()) {
        const int rc = qsim_sync(c->shard);
        if (rc) return cfail(rc, "%s", qsim_last_error());
        for (auto &pr : c->timing) {
            float ms = 0.f;
            if (hipEventElapsedTime(&ms, pr.first, pr.second) == hipSuccess) c->ms += ms;
            (void)hipEventDestroy(pr.first);
            (void)hipEventDestroy(pr.second);
        }
        c->timing.clear();
    }
    if (exchanges) *exchanges = c->exchanges;
    if (bytes_sent) *bytes_sent = c->bytes_sent;
    if (seconds) *seconds = c->ms * 1e-3;
    if (reset) { c->exchanges = 0; c->bytes_sent = 0; c->ms = 0; }
    return QSIM_OK;
}

extern "C" int qsim_rank_comm_pack_counts(const qsim_rank_comm *c, uint64_t *fused, uint64_t *separate) {
    if (!c) return QSIM_ERR_ARG;
    if (fused) *fused = c->fused_packs;
    if (separate) *separate = c->separate_packs;
    return QSIM_OK;
}

// Diagnostic: the first `count` doubles of the shard travel through ncclSend -> ncclRecv to this same rank (one group, on
// the shard's stream) into the scratch buffer and are compared on the host.  It is the only way to drive the RCCL call
// path of qsim_rank_comm_exchange where a single GPU is present (a 1-rank communicator has nobody to exchange with).
extern "C" int qsim_rank_comm_loopback(qsim_rank_comm *c, uint64_t count) {
    if (!c) return cfail(QSIM_ERR_ARG, "NULL argument");
    const uint64_t cap = (uint64_t)2 << qsim_num_qubits(c->shard);
    if (count < 1 || count > cap) return cfail(QSIM_ERR_ARG, "loopback count outside the shard");
    if (hipSetDevice(c->device) != hipSuccess) return cfail(QSIM_ERR_DEVICE, "hipSetDevice failed");
    hipStream_t stream = (hipStream_t)qsim_stream(c->shard);
    const void *state = qsim_device_ptr(c->shard);
    if (!state) return cfail(QSIM_ERR_DEVICE, "%s", qsim_last_error());
    ncclResult_t nr = ncclGroupStart();
    if (nr == ncclSuccess) nr = ncclSend(state, count, ncclDouble, c->rank, c->comm, stream);
    if (nr == ncclSuccess) nr = ncclRecv(c->scratch, count, ncclDouble, c->rank, c->comm, stream);
    const ncclResult_t ne = ncclGroupEnd();
    if (nr == ncclSuccess) nr = ne;
    if (nr != ncclSuccess) return cfail(QSIM_ERR_DEVICE, "RCCL loopback failed: %s", ncclGetErrorString(nr));
    if (hipStreamSynchronize(stream) != hipSuccess) return cfail(QSIM_ERR_DEVICE, "stream sync failed");
    std::vector<double> a(count), b(count);
    if (hipMemcpy(a.data(), state, count * 8, hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(b.data(), c->scratch, count * 8, hipMemcpyDeviceToHost) != hipSuccess)
        return cfail(QSIM_ERR_DEVICE, "copy back failed");
    if (memcmp(a.data(), b.data(), count * 8) != 0) return cfail(QSIM_ERR_DEVICE, "RCCL loopback: received data differs");
    return QSIM_OK;
}

// Geometry planning for a shard's part of a plan: every local step's ops for `shard` as a circuit through
// qsim_tune_circuit (include/qsim.h, "measured pass geometry").  Leaves `s` reset; budget_ms bounds the total.
extern "C" int qsim_shard_plan_tune(const qsim_shard_plan *p, int shard, qsim_state *s, int max_candidates, double budget_ms,
                                    qsim_tune_report *report) {
    if (!p || !s || shard < 0 || shard >= p->P) return cfail(QSIM_ERR_ARG, "bad argument");
    qsim_tune_report total{};
    const int rc = plan_shard_steps(p->plan, shard, s, max_candidates < 2 ? 2 : max_candidates, budget_ms, &total);
    if (rc) return rc;
    if (report) *report = total;
    return QSIM_OK;
}
