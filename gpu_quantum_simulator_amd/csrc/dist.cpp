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
#include <cstring>
#include <string>
#include <vector>

#include "circuit.h"
#include "qsim_internal.h"

using cd = std::complex<double>;

namespace {

struct LGate { // logical gate
    int kind;  // QSIM_GATE_U1 / QSIM_GATE_CX
    int q0, q1;
    cd m[4];
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
};

struct Plan {
    int n = 0, p = 0, m = 0;
    std::vector<Step> steps;
    std::vector<int> final_pos;
    int exchanges = 0;
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

std::vector<int> choose_globals(const std::vector<LGate> &gates, const std::vector<int> &pos, int n, int p, int m) {
    std::vector<long> nxt(n, kInf);
    int found = 0;
    for (size_t i = 0; i < gates.size() && found < n; i++) {
        int q[2], c;
        needs_local(gates[i], q, c);
        for (int k = 0; k < c; k++)
            if (nxt[q[k]] == kInf) { nxt[q[k]] = (long)i; found++; }
    }
    std::vector<int> order(n);
    for (int q = 0; q < n; q++) order[q] = q;
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

bool build_plan(int n, int p, const std::vector<LGate> &gates, Plan &plan) {
    const int P = 1 << p, m = n - p;
    plan.n = n; plan.p = p; plan.m = m;
    std::vector<int> pos(n);
    for (int q = 0; q < n; q++) pos[q] = q;
    std::vector<LGate> remaining(gates);
    bool first = true;
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
            plan.steps.push_back(std::move(st));
        }
        if (!deferred.empty()) {
            std::vector<int> ng = choose_globals(deferred, pos, n, p, m);
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

void gates_of(const qsim_circuit *c, std::vector<LGate> &out) {
    out.reserve((size_t)c->count);
    for (long i = 0; i < c->count; i++) {
        const qsim_gate_rec &g = c->gates[i];
        LGate lg{};
        lg.kind = g.kind; lg.q0 = g.q0; lg.q1 = g.q1;
        if (g.kind == QSIM_GATE_U1)
            for (int k = 0; k < 4; k++) lg.m[k] = cd(c->mats2[8 * (long)g.mat + 2 * k], c->mats2[8 * (long)g.mat + 2 * k + 1]);
        out.push_back(lg);
    }
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

} // namespace

struct qsim_cluster {
    int n = 0, p = 0, m = 0, P = 0;
    std::vector<int> devices;
    std::vector<qsim_state *> shard;
    std::vector<double2 *> scratch;
    std::vector<int> pos; // logical -> physical after the last run
    uint64_t exchanges = 0;
    double exchange_bytes = 0; // per shard, summed over exchanges
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
    for (size_t r = 0; r < c->shard.size(); r++) {
        if (c->scratch[r]) { (void)hipSetDevice(c->devices[r]); (void)hipFree(c->scratch[r]); }
        qsim_destroy(c->shard[r]);
    }
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
    for (int r = 0; r < num_shards; r++) {
        const int dev = devices ? devices[r] : (r % ndev);
        if (dev < 0 || dev >= ndev) { qsim_cluster_destroy(c); return cfail(QSIM_ERR_ARG, "device %d out of range", dev); }
        c->devices.push_back(dev);
        qsim_state *s = nullptr;
        int rc = qsim_create(&s, c->m, dev);
        c->shard.push_back(s);
        c->scratch.push_back(nullptr);
        if (rc == QSIM_OK && p > 0) {
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
    return QSIM_OK;
}

static int apply_local(qsim_cluster *c, const Step &st) {
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
        const int rc = qsim_flush(s); // every shard's passes are in flight before the next one is scheduled
        if (rc) return cfail(rc, "%s", qsim_last_error());
    }
    return QSIM_OK;
}

static int exchange(qsim_cluster *c, const Step &st) {
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
    c->exchanges++;
    c->exchange_bytes += (double)blk_bytes * ((1 << k) - 1);
    return QSIM_OK;
}

// Plans and runs the circuit from the CURRENT state with the map reset to what the planner assumes, i.e. call
// qsim_cluster_reset first (compute_state_vector semantics: one circuit per state).
extern "C" int qsim_cluster_run_circuit(qsim_cluster *c, const qsim_circuit *circ) {
    if (!c || !circ) return cfail(QSIM_ERR_ARG, "NULL argument");
    if (circ->num_q != c->n) return cfail(QSIM_ERR_ARG, "circuit has %d qubits, cluster has %d", circ->num_q, c->n);
    for (int q = 0; q < c->n; q++)
        if (c->pos[q] != q) return cfail(QSIM_ERR_ARG, "cluster already holds a permuted state: reset it first");
    for (long i = 0; i < circ->count; i++)
        if (circ->gates[i].kind == QSIM_GATE_U2) return cfail(QSIM_ERR_ARG, "generic 2-qubit gates are not supported on clusters");
    std::vector<LGate> gates;
    gates_of(circ, gates);
    Plan plan;
    if (!build_plan(c->n, c->p, gates, plan)) return cfail(QSIM_ERR_ARG, "planner made no progress");
    for (const Step &st : plan.steps) {
        const int rc = st.exchange ? exchange(c, st) : apply_local(c, st);
        if (rc) return rc;
    }
    c->pos = plan.final_pos;
    return QSIM_OK;
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
// cumulative distribution runs in, whatever the qubit map of the last run left behind).  Each shard is streamed to the
// host once in 64 MiB pieces and contributes to the sums of the 2^12-amplitude logical blocks it holds a part of — the
// "P partial sums" of SURVEY 8f row 1, added in shard order; a draw then fetches only its own block.
extern "C" int qsim_cluster_sample(qsim_cluster *c, const double *randoms, long shots, uint64_t *out) {
    if (!c || (shots > 0 && (!randoms || !out))) return cfail(QSIM_ERR_ARG, "NULL argument");
    constexpr int kBlockBits = 12;
    const int bb = c->n < kBlockBits ? c->n : kBlockBits;
    const uint64_t N = 1ULL << c->n, nblocks = N >> bb, bsize = 1ULL << bb, M = 1ULL << c->m;
    std::vector<int> inv(c->n); // physical bit -> logical qubit
    for (int q = 0; q < c->n; q++) inv[c->pos[q]] = q;
    // logical index of a physical index, by halves of the local bits (a bit permutation is linear over OR)
    const int lo_bits = c->m < 11 ? c->m : 11;
    std::vector<uint64_t> t_lo(1ULL << lo_bits);
    for (uint64_t j = 0; j < t_lo.size(); j++) {
        uint64_t l = 0;
        for (int b = 0; b < lo_bits; b++) l |= ((j >> b) & 1ULL) << inv[b];
        t_lo[j] = l;
    }
    auto logical_hi = [&](uint64_t ph_hi) { // ph_hi = physical index >> lo_bits
        uint64_t l = 0;
        for (int b = lo_bits; b < c->n; b++) l |= ((ph_hi >> (b - lo_bits)) & 1ULL) << inv[b];
        return l;
    };
    std::vector<double> prefix(nblocks, 0.0);
    const uint64_t piece = M < (1ULL << 22) ? M : (1ULL << 22);
    std::vector<double> buf(2 * piece);
    for (int r = 0; r < c->P; r++)
        for (uint64_t at = 0; at < M; at += piece) {
            const int rc = qsim_read(c->shard[r], at, piece, buf.data());
            if (rc) return cfail(rc, "%s", qsim_last_error());
            for (uint64_t j = 0; j < piece; j++) {
                const uint64_t ph = ((uint64_t)r << c->m) | (at + j);
                const uint64_t l = logical_hi(ph >> lo_bits) | t_lo[ph & (t_lo.size() - 1)];
                prefix[l >> bb] += buf[2 * j] * buf[2 * j] + buf[2 * j + 1] * buf[2 * j + 1];
            }
        }
    double acc = 0.0;
    for (uint64_t b = 0; b < nblocks; b++) { acc += prefix[b]; prefix[b] = acc; } // cumulative at the END of block b
    std::vector<double> blk(2 * bsize);
    uint64_t cached = ~0ULL;
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
                for (uint64_t i = 0; i < bsize; i++) {
                    const uint64_t ph = physical_index(c, b * bsize + i);
                    const int rc = qsim_read(c->shard[ph >> c->m], ph & (M - 1), 1, blk.data() + 2 * i);
                    if (rc) return cfail(rc, "%s", qsim_last_error());
                }
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
