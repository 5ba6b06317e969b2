// engine.cpp — the C ABI of include/qsim.h: device state, gate queue, scheduler driver, launches, stats.
// Host C++ compiled by hipcc for the HIP runtime API; every kernel lives in kernels.hip.
//
// There is no CPU execution path in this file or anywhere in libqsim.so: if the HIP runtime reports no
// usable device, qsim_create() fails with QSIM_ERR_DEVICE.
#include <algorithm>
#include <array>
#include <atomic>
#include <chrono>
#include <map>
#include <mutex>
#include <thread>
#include <tuple>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "circuit.h"
#include "qsim_internal.h"
#include "scheduler.h"

using namespace qsim;

// ---- errors --------------------------------------------------------------------------------------------
static thread_local std::string g_err;

static int fail(int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(expr)                                                                                       \
    do {                                                                                                    \
        hipError_t e_ = (expr);                                                                             \
        if (e_ != hipSuccess)                                                                               \
            return fail(e_ == hipErrorOutOfMemory ? QSIM_ERR_ALLOC : QSIM_ERR_DEVICE, "%s failed: %s", #expr, \
                        hipGetErrorString(e_));                                                             \
    } while (0)

extern "C" const char *qsim_last_error(void) { return g_err.c_str(); }

extern "C" int qsim_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

extern "C" int qsim_device_init(int device) {
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(QSIM_ERR_DEVICE, "no HIP device available (libqsim has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(QSIM_ERR_ARG, "device %d out of range (%d present)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    HIP_TRY(hipFree(nullptr)); // forces context creation
    return QSIM_OK;
}

// ---- state ---------------------------------------------------------------------------------------------
struct QueuedGate {
    int kind, q0, q1;
    cd m[16];
};

struct ProfEvent {
    hipEvent_t start, stop;
    int kclass;
    int n_ops;
    uint64_t high_mask;
    uint64_t order_code; // tile passes: the high tile bits in tile-local order, 5 bits each, lowest first
    double visited;      // tile passes: fraction of the register's tiles the pass works on (the state's support)
    std::vector<uint8_t> forms; // tile passes: one byte per block (qsim_launch_log_blocks)
};
struct LaunchRec { int kclass, n_ops; uint64_t high_mask; double ms; uint64_t order_code; double visited; std::vector<uint8_t> forms; };

// Everything a schedule depends on: the options that shape it, the state's support, the QSIM_SCHED_* overrides and the
// gates themselves.  A cached plan is only replayed for a queue whose identity EQUALS the one it was built from, field by
// field and gate by gate; the 64-bit key merely finds the candidates (FNV-1a is not collision resistant, and "results
// identical to the reference" must not rest on a hash).
struct PlanIdentity {
    int opts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    uint64_t support = 0;
    SchedEnv env;
    std::vector<QueuedGate> gates;
};
static bool same_gates(const QueuedGate *a, const QueuedGate *b, size_t count) {
    for (size_t i = 0; i < count; i++) {
        if (a[i].kind != b[i].kind || a[i].q0 != b[i].q0 || a[i].q1 != b[i].q1) return false;
        if (a[i].kind != QSIM_GATE_CX && memcmp(a[i].m, b[i].m, (a[i].kind == QSIM_GATE_U1 ? 4 : 16) * sizeof(cd)) != 0) return false;
    }
    return true;
}

struct CachedPlan {
    uint64_t key = 0, wisdom_epoch = 0, last_use = 0;
    PlanIdentity id;
    std::vector<Pass> passes;
    std::vector<TileGeom> geoms;   // per pass; meaningful for tile passes: the geometry in the order it was launched with
    std::vector<size_t> op_first;  // per pass: index of its first TileOp in d_ops
    TileOp *d_ops = nullptr;
};

struct qsim_state {
    int n = 0, device = 0;
    hipStream_t stream = nullptr;
    void *amps = nullptr; // 2^n amplitudes: (re, im) pairs of double (16 B) or, with f32, of float (8 B)
    bool f32 = false;
    bool owns = false;
    // Second buffer of the same size for out-of-place tile passes (QSIM_OPT_PINGPONG; k_tile comment): a pass reads
    // `amps` and writes `spare`, then the two swap.  Within one qsim_flush an even number of passes run that way, so the
    // state is back in the buffer it started from when the flush returns (qsim_device_ptr stays what it was, an external
    // buffer holds the result).  Allocated on first use for states that own their buffer, or lent by the caller
    // (qsim_set_spare_buffer: a sharded run lends its exchange scratch, which is idle between exchanges).
    void *spare = nullptr;
    bool owns_spare = false, spare_failed = false;
    int pingpong = 1; // 0 never, 1 when the state is large enough to gain (kPingPongMinBytes), 2 whenever a second buffer can be had
    size_t amp_bytes() const { return f32 ? 8 : 16; }
    // options
    int fuse = 3, profile = 0, tile_bits = 12, tile_low_bits = 3, tile_max_ops = 32, grid_cap = 0, tile_threads = 0, tile_pad_from = 10, debug_skip_ops = 0, debug_skip_mem = 0, debug_tile_order = 0;
    uint64_t tile_passes = 0; // launched so far (seeds the probe permutations of QSIM_OPT_DEBUG_TILE_ORDER)
    long max_pending = 1L << 16;
    // queue
    std::vector<QueuedGate> queue;
    double zero_ket_amp = 1.0;     // amplitude at index 0 of the pending basis state (0: a shard that does not hold index 0)
    bool zero_ket_pending = false; // |0...0> requested but not written yet (folded into the first tile pass if possible)
    // After a reset the state is zero wherever an index bit outside `support` is set, and stays so until a pass mixes that
    // qubit in: the first tile pass writes ONE tile (every other tile is zero), the second 2^(tile qubits new to it)
    // tiles, and so on until the support is the whole register — typically the third pass of a random circuit.  While
    // `partial` is set, memory outside the support has never been written (it is zero by definition): tile passes visit
    // only tiles inside it and stage the rest of a tile in as zeros (launch_tile zero_mask); anything else that looks at
    // the buffer (other kernels, reads, exchanges) first gets the zeros written (materialize_zero_ket).
    bool partial = false;
    uint64_t support = 0; // qubits some tile pass has had inside its tile since the reset
    int sparse_start = 1; // QSIM_OPT_SPARSE_START
    // op ring for tile passes
    TileOp *d_ops = nullptr, *h_ops = nullptr;
    size_t ops_cap = 0, ops_used = 0;
    double *d_scalar = nullptr;
    // stats
    qsim_stats stats{};
    std::vector<ProfEvent> events;      // recorded, not yet resolved
    std::vector<LaunchRec> launch_log;  // per-launch times since the last qsim_reset_stats (profile mode)
    std::vector<hipEvent_t> event_pool; // reusable
    // Plans of recently flushed gate queues (QSIM_OPT_PLAN_CACHE): the passes as scheduled, the tile passes' bit orders and
    // their TileOps resident on the device.  A queue that hashes to a cached plan is replayed launch by launch — no
    // scheduling, no block preparation, no H2D copy — which is what a loop that re-runs one circuit shape (a benchmark's
    // steps, a variational algorithm's iterations) pays for at n <= 26, where a pass is shorter than its planning.
    std::vector<struct CachedPlan> plans;
    uint64_t plan_clock = 0;
    int plan_cache = 1;
    int tune_schedules = 4; // qsim_tune_circuit: how many of the model's best schedules are run (QSIM_TUNE_SCHEDULES)
    long debug_plan_key = 0; // QSIM_OPT_DEBUG_PLAN_KEY: != 0 = every queue gets this key (forced collisions, for the tests of the identity check)
    uint64_t plan_hits = 0, plan_key_collisions = 0; // replays; key matches whose identity differed
    // qsim_create_async: the amplitude buffer is being allocated by this thread (hipMalloc of 16 GiB takes 0.04-0.25 s) while the
    // caller parses, sets options and chooses a schedule; whatever needs the buffer joins it first (await_buffer).
    std::thread alloc_thread;
    std::atomic<bool> alloc_done{true};
    hipError_t alloc_err = hipSuccess;
};

// Joins the allocation of an asynchronously created state; QSIM_ERR_ALLOC ("Malloc error", quantum_simulator.c:170) if it failed.
static int await_buffer(qsim_state *s) {
    if (s->alloc_thread.joinable()) s->alloc_thread.join();
    if (s->alloc_err != hipSuccess) {
        return fail(s->alloc_err == hipErrorOutOfMemory ? QSIM_ERR_ALLOC : QSIM_ERR_DEVICE, "Malloc error: %s (state needs %zu bytes)",
                    hipGetErrorString(s->alloc_err), s->amp_bytes() << s->n);
    }
    return QSIM_OK;
}

static constexpr size_t kOpsCap = 512;  // a pass holds <= tile_max_ops blocks; the ring wraps with a stream sync

static int make_state(qsim_state **out, int num_q, int device, void *ext, bool f32 = false, bool async = false) {
    if (!out) return fail(QSIM_ERR_ARG, "qsim_create: out is NULL");
    *out = nullptr;
    if (num_q < 0 || num_q > 40) return fail(QSIM_ERR_ARG, "qsim_create: %d qubits unsupported", num_q);
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
        return fail(QSIM_ERR_DEVICE, "no HIP device available (libqsim has no CPU fallback)");
    if (device < 0 || device >= ndev) return fail(QSIM_ERR_ARG, "device %d out of range (%d present)", device, ndev);
    HIP_TRY(hipSetDevice(device));
    qsim_state *s = new qsim_state();
    s->n = num_q;
    s->device = device;
    s->f32 = f32;
    if (f32) { s->tile_bits = 13; s->tile_low_bits = 4; } // same 64 KiB of LDS per tile and the same 128-B runs as the fp64 default (64-B runs: 13 208 vs 14 130 gate-applies/s at n = 30)
    const size_t bytes = s->amp_bytes() << num_q;
    hipError_t e = hipStreamCreateWithFlags(&s->stream, hipStreamNonBlocking);
    if (e == hipSuccess) {
        if (ext) s->amps = ext;
        else if (async) {
            s->owns = true;
            s->alloc_done.store(false);
            try {
                s->alloc_thread = std::thread([s, device, bytes]() {
                    hipError_t ae = hipSetDevice(device);
                    if (ae == hipSuccess) ae = hipMalloc(&s->amps, bytes);
                    if (ae != hipSuccess) { s->amps = nullptr; (void)hipGetLastError(); }
                    s->alloc_err = ae;
                    s->alloc_done.store(true);
                });
            } catch (...) { // no thread to be had: allocate here, like qsim_create
                s->alloc_done.store(true);
                e = hipMalloc(&s->amps, bytes);
                s->owns = (e == hipSuccess);
            }
        } else { e = hipMalloc(&s->amps, bytes); s->owns = (e == hipSuccess); }
    }
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_ops, kOpsCap * sizeof(TileOp));
    if (e == hipSuccess) e = hipHostMalloc((void **)&s->h_ops, kOpsCap * sizeof(TileOp), hipHostMallocDefault);
    if (e == hipSuccess) e = hipMalloc((void **)&s->d_scalar, 64);
    if (e != hipSuccess) {
        const int code = fail(e == hipErrorOutOfMemory ? QSIM_ERR_ALLOC : QSIM_ERR_DEVICE,
                              "Malloc error: %s (state needs %zu bytes)", hipGetErrorString(e), bytes);
        qsim_destroy(s);
        return code;
    }
    s->ops_cap = kOpsCap;
    *out = s;
    return qsim_reset(s);
}

extern "C" int qsim_create(qsim_state **out, int num_q, int device) { return make_state(out, num_q, device, nullptr); }
extern "C" int qsim_create_f32(qsim_state **out, int num_q, int device) { return make_state(out, num_q, device, nullptr, true); }
// The same with the amplitude buffer allocated on a helper thread: returns at once; gates may be queued, options set and a
// schedule chosen (qsim_choose_schedule_while_allocating) meanwhile, and the first call that needs the buffer waits for it.  An
// allocation failure surfaces there as QSIM_ERR_ALLOC.
extern "C" int qsim_create_async(qsim_state **out, int num_q, int device, int precision_bits) {
    if (precision_bits != 32 && precision_bits != 64) return fail(QSIM_ERR_ARG, "precision must be 32 or 64");
    return make_state(out, num_q, device, nullptr, precision_bits == 32, true);
}
extern "C" int qsim_precision_bits(const qsim_state *s) { return s ? (s->f32 ? 32 : 64) : -1; }
extern "C" int qsim_create_external(qsim_state **out, int num_q, int device, void *device_amps) {
    if (!device_amps) return fail(QSIM_ERR_ARG, "qsim_create_external: device_amps is NULL");
    return make_state(out, num_q, device, device_amps);
}

extern "C" void qsim_destroy(qsim_state *s) {
    if (!s) return;
    if (s->alloc_thread.joinable()) s->alloc_thread.join();
    (void)hipSetDevice(s->device);
    if (s->stream) (void)hipStreamSynchronize(s->stream);
    for (auto &pe : s->events) { (void)hipEventDestroy(pe.start); (void)hipEventDestroy(pe.stop); }
    for (auto ev : s->event_pool) (void)hipEventDestroy(ev);
    for (CachedPlan &pl : s->plans)
        if (pl.d_ops) (void)hipFree(pl.d_ops);
    if (s->owns && s->amps) (void)hipFree(s->amps);
    if (s->owns_spare && s->spare) (void)hipFree(s->spare);
    if (s->d_ops) (void)hipFree(s->d_ops);
    if (s->h_ops) (void)hipHostFree(s->h_ops);
    if (s->d_scalar) (void)hipFree(s->d_scalar);
    if (s->stream) (void)hipStreamDestroy(s->stream);
    delete s;
}

extern "C" int qsim_num_qubits(const qsim_state *s) { return s ? s->n : -1; }
static int materialize_zero_ket(qsim_state *s);
// The buffer as every queued gate left it: pending gates are launched and a lazily held |0...0> is written first (the
// work is ON the state's stream, not finished: order later accesses after qsim_stream() or call qsim_sync).
extern "C" void *qsim_device_ptr(qsim_state *s) {
    if (!s) return nullptr;
    if (qsim_flush(s) != QSIM_OK || materialize_zero_ket(s) != QSIM_OK) return nullptr;
    return s->amps;
}
extern "C" void *qsim_stream(qsim_state *s) { return s ? (void *)s->stream : nullptr; }
// The buffer itself, nothing launched and nothing written first: for a caller that is about to overwrite (part of) it.
extern "C" void *qsim_state_buffer(qsim_state *s) { return s && await_buffer(s) == QSIM_OK ? s->amps : nullptr; }

extern "C" int qsim_set_option(qsim_state *s, int option, long value) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    if (!s->queue.empty()) {
        const int rc = qsim_flush(s); // options apply to gates queued after the call
        if (rc) return rc;
    }
    switch (option) {
    case QSIM_OPT_FUSE:
        if (value < 0 || value > 3) return fail(QSIM_ERR_ARG, "fuse level %ld not in 0..3", value);
        s->fuse = (int)value;
        break;
    case QSIM_OPT_PROFILE: s->profile = value < 0 ? 0 : value > 2 ? 2 : (int)value; break;
    case QSIM_OPT_TILE_BITS:
        if (value < 8 || value > (s->f32 ? 14 : 13)) return fail(QSIM_ERR_ARG, "tile_bits %ld not in 8..%d", value, s->f32 ? 14 : 13);
        s->tile_bits = (int)value;
        break;
    case QSIM_OPT_TILE_LOW_BITS:
        if (value < 2 || value > 6) return fail(QSIM_ERR_ARG, "tile_low_bits %ld not in 2..6", value);
        s->tile_low_bits = (int)value;
        break;
    case QSIM_OPT_MAX_PENDING:
        if (value < 1) return fail(QSIM_ERR_ARG, "max_pending must be positive");
        s->max_pending = value;
        break;
    case QSIM_OPT_TILE_MAX_OPS:
        if (value < 1 || value > (long)kOpsCap) return fail(QSIM_ERR_ARG, "tile_max_ops %ld not in 1..%zu", value, kOpsCap);
        s->tile_max_ops = (int)value;
        break;
    case QSIM_OPT_GRID_CAP:
        if (value < 0) return fail(QSIM_ERR_ARG, "grid_cap must be >= 0");
        s->grid_cap = (int)value;
        break;
    case QSIM_OPT_TILE_PAD_FROM:
        s->tile_pad_from = (int)value;
        break;
    case QSIM_OPT_DEBUG_SKIP_OPS:
        s->debug_skip_ops = value != 0;
        break;
    case QSIM_OPT_DEBUG_SKIP_MEM:
        s->debug_skip_mem = value != 0;
        break;
    case QSIM_OPT_DEBUG_TILE_ORDER:
        s->debug_tile_order = (int)value;
        break;
    case QSIM_OPT_PLAN_CACHE:
        s->plan_cache = value != 0;
        break;
    case QSIM_OPT_SPARSE_START:
        s->sparse_start = value != 0;
        break;
    case QSIM_OPT_DEBUG_PLAN_KEY:
        s->debug_plan_key = value;
        break;
    case QSIM_OPT_PINGPONG:
        if (value < 0 || value > 2) return fail(QSIM_ERR_ARG, "pingpong must be 0 (never), 1 (auto) or 2 (always)");
        s->pingpong = (int)value;
        s->spare_failed = false;
        break;

    case QSIM_OPT_TILE_THREADS:
        if (value != 0 && value != 256 && value != 512 && value != 1024)
            return fail(QSIM_ERR_ARG, "tile_threads must be 0 (auto), 256, 512 or 1024");
        s->tile_threads = (int)value;
        break;
    default: return fail(QSIM_ERR_ARG, "unknown option %d", option);
    }
    return QSIM_OK;
}

extern "C" long qsim_get_option(const qsim_state *s, int option) {
    if (!s) return -1;
    switch (option) {
    case QSIM_OPT_FUSE: return s->fuse;
    case QSIM_OPT_PROFILE: return s->profile;
    case QSIM_OPT_TILE_BITS: return s->tile_bits;
    case QSIM_OPT_TILE_LOW_BITS: return s->tile_low_bits;
    case QSIM_OPT_MAX_PENDING: return s->max_pending;
    case QSIM_OPT_TILE_MAX_OPS: return s->tile_max_ops;
    case QSIM_OPT_GRID_CAP: return s->grid_cap;
    case QSIM_OPT_TILE_THREADS: return s->tile_threads;
    case QSIM_OPT_TILE_PAD_FROM: return s->tile_pad_from;
    case QSIM_OPT_DEBUG_SKIP_OPS: return s->debug_skip_ops;
    case QSIM_OPT_DEBUG_SKIP_MEM: return s->debug_skip_mem;
    case QSIM_OPT_DEBUG_TILE_ORDER: return s->debug_tile_order;
    case QSIM_OPT_PLAN_CACHE: return s->plan_cache;
    case QSIM_OPT_PINGPONG: return s->pingpong;
    case QSIM_OPT_SPARSE_START: return s->sparse_start;
    case QSIM_OPT_DEBUG_PLAN_KEY: return s->debug_plan_key;
    default: return -1;
    }
}

// ---- profiling events ----------------------------------------------------------------------------------
static hipEvent_t take_event(qsim_state *s) {
    if (!s->event_pool.empty()) {
        hipEvent_t e = s->event_pool.back();
        s->event_pool.pop_back();
        return e;
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
}

static int resolve_events(qsim_state *s) {
    if (s->events.empty()) return QSIM_OK;
    HIP_TRY(hipStreamSynchronize(s->stream));
    for (auto &pe : s->events) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, pe.start, pe.stop) == hipSuccess) s->stats.k_ms[pe.kclass] += ms;
        if (s->launch_log.size() < (1u << 20)) s->launch_log.push_back({pe.kclass, pe.n_ops, pe.high_mask, (double)ms, pe.order_code, pe.visited, std::move(pe.forms)});
        s->event_pool.push_back(pe.start);
        s->event_pool.push_back(pe.stop);
    }
    s->events.clear();
    return QSIM_OK;
}

struct LaunchScope { // records a start/stop pair around one launch when profiling is on
    qsim_state *s;
    ProfEvent pe{};
    bool on;
    LaunchScope(qsim_state *st, int kclass, int n_ops = 1, uint64_t high_mask = 0, uint64_t order_code = 0, double visited = 1.0) : s(st), on(st->profile != 0) {
        if (on) {
            pe.visited = visited;
            pe.kclass = kclass;
            pe.n_ops = n_ops;
            pe.high_mask = high_mask;
            pe.order_code = order_code;
            pe.start = take_event(s);
            pe.stop = take_event(s);
            (void)hipEventRecord(pe.start, s->stream);
        }
    }
    ~LaunchScope() {
        if (on) {
            (void)hipEventRecord(pe.stop, s->stream);
            s->events.push_back(pe);
        }
    }
};

static void account(qsim_state *s, int kclass, double bytes) {
    s->stats.launches++;
    s->stats.algorithmic_bytes += bytes;
    s->stats.k_launches[kclass]++;
    s->stats.k_bytes[kclass] += bytes;
}

// Writes the pending |0...0> with the init kernel (when the next operation cannot generate it itself), or the zeros of a
// state that has only been written inside its support so far.
static int materialize_zero_ket(qsim_state *s) {
    { const int rc = await_buffer(s); if (rc) return rc; }
    if (!s->zero_ket_pending && !s->partial) return QSIM_OK;
    HIP_TRY(hipSetDevice(s->device)); // a cluster drives several devices from one thread
    LaunchCfg cfg{s->stream, s->grid_cap};
    const double state_bytes = (double)s->amp_bytes() * (double)(1ULL << s->n);
    if (s->zero_ket_pending) {
        s->zero_ket_pending = false;
        {
            LaunchScope scope(s, QSIM_K_INIT);
            HIP_TRY(launch_init(cfg, s->amps, s->f32, s->n, s->zero_ket_amp));
        }
        account(s, QSIM_K_INIT, state_bytes);
    } else {
        const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
        {
            LaunchScope scope(s, QSIM_K_INIT);
            HIP_TRY(launch_zero_outside(cfg, s->amps, s->f32, s->n, nmask & ~s->support));
        }
        account(s, QSIM_K_INIT, state_bytes * (1.0 - 1.0 / (double)(1ULL << __builtin_popcountll(nmask & ~s->support))));
    }
    s->partial = false;
    return QSIM_OK;
}

extern "C" int qsim_reset_shard(qsim_state *s, int holds_index0);
extern "C" int qsim_reset(qsim_state *s) { return qsim_reset_shard(s, 1); }

extern "C" int qsim_reset_shard(qsim_state *s, int holds_index0) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    HIP_TRY(hipSetDevice(s->device));
    s->queue.clear();
    s->zero_ket_amp = holds_index0 ? 1.0 : 0.0;
    // |0...0> is not written here: if the first pass after the reset is a tile pass it generates the state in LDS
    // (one write of the state instead of write + read + write); anything else materialises it first.
    s->zero_ket_pending = true;
    s->partial = false;
    s->support = 0;
    return QSIM_OK;
}

// The caller filled the buffer itself (the receiving end of an exchange) and knows where the new contents can be non-zero:
// every amplitude whose index has a bit outside `support` is zero BY DEFINITION from now on (its memory need not have been
// written), exactly the situation after the first tile passes of a run (qsim_state::support).  Tile passes then visit only
// that part; anything else that looks at the buffer gets the zeros written first.
extern "C" int qsim_set_support(qsim_state *s, uint64_t support) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    const int rc = qsim_flush(s);
    if (rc) return rc;
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    s->zero_ket_pending = false;
    s->support = support & nmask;
    s->partial = s->support != nmask;
    if (!s->sparse_start && s->partial) return materialize_zero_ket(s); // the option is off: keep the state dense
    return QSIM_OK;
}

// What the buffer holds right now, without touching it: *support = index bits that may be 1 in a written, possibly non-zero
// amplitude (all ones for a dense state); *kind = 0 written (inside the support), 1 a pending basis state amp0 * |0...0> that no
// kernel has written yet (amp0 = 0: the all-zero vector of a shard that holds nothing).  Queued gates are launched first.
extern "C" int qsim_get_support(qsim_state *s, uint64_t *support, int *kind, double *amp0) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    const int rc = qsim_flush(s);
    if (rc) return rc;
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    if (support) *support = s->zero_ket_pending ? 0 : s->partial ? (s->support & nmask) : nmask;
    if (kind) *kind = s->zero_ket_pending ? 1 : 0;
    if (amp0) *amp0 = s->zero_ket_pending ? s->zero_ket_amp : 0.0;
    return QSIM_OK;
}

// 1 when the state is the all-zero vector of a shard that holds nothing (qsim_reset_shard(s, 0), nothing written since): gates
// queued on it change nothing, so the queue does not matter and nothing is flushed.
extern "C" int qsim_holds_nothing(const qsim_state *s) { return s && s->zero_ket_pending && s->zero_ket_amp == 0.0 ? 1 : 0; }

// ---- gate queue ----------------------------------------------------------------------------------------
static int enqueue(qsim_state *s, const QueuedGate &g) {
    s->queue.push_back(g);
    s->stats.gates++;
    if ((long)s->queue.size() >= s->max_pending) return qsim_flush(s);
    return QSIM_OK;
}

extern "C" int qsim_apply_1q(qsim_state *s, const double *U, int target) {
    if (!s || !U) return fail(QSIM_ERR_ARG, "NULL argument");
    if (target < 0 || target >= s->n) return fail(QSIM_ERR_ARG, "qubit %d out of range (n = %d)", target, s->n);
    QueuedGate g;
    g.kind = QSIM_GATE_U1; g.q0 = target; g.q1 = -1;
    for (int k = 0; k < 4; k++) g.m[k] = cd(U[2 * k], U[2 * k + 1]);
    return enqueue(s, g);
}

extern "C" int qsim_apply_cx(qsim_state *s, int control, int target) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    if (control < 0 || control >= s->n || target < 0 || target >= s->n)
        return fail(QSIM_ERR_ARG, "cx operands (%d, %d) out of range (n = %d)", control, target, s->n);
    QueuedGate g;
    g.kind = QSIM_GATE_CX; g.q0 = control; g.q1 = target;
    return enqueue(s, g);
}

extern "C" int qsim_apply_2q(qsim_state *s, const double *U, int q_hi, int q_lo) {
    if (!s || !U) return fail(QSIM_ERR_ARG, "NULL argument");
    if (q_lo < 0 || q_hi >= s->n || q_lo >= q_hi)
        return fail(QSIM_ERR_ARG, "2q operands need 0 <= q_lo < q_hi < n (got %d, %d, n = %d)", q_hi, q_lo, s->n);
    QueuedGate g;
    g.kind = QSIM_GATE_U2; g.q0 = q_hi; g.q1 = q_lo;
    for (int k = 0; k < 16; k++) g.m[k] = cd(U[2 * k], U[2 * k + 1]);
    return enqueue(s, g);
}

// ---- scheduling + launch -------------------------------------------------------------------------------
static SchedConfig sched_config(int n, int fuse, int tile_bits, int tile_low_bits, int tile_max_ops, int pad_from = 10, bool f32 = false,
                                uint64_t initial_support = 0) {
    return engine_sched_config(n, fuse, tile_bits, tile_low_bits, tile_max_ops, pad_from, f32, initial_support);
}

static inline void to_m2(const FusedOp &op, M2 &u) {
    for (int k = 0; k < 4; k++) { u.re[k] = op.m[k].real(); u.im[k] = op.m[k].imag(); }
}
static inline void to_m4(const FusedOp &op, M4 &u) {
    for (int k = 0; k < 16; k++) { u.re[k] = op.m[k].real(); u.im[k] = op.m[k].imag(); }
}

// tile-local bit of global qubit q under geometry g
static inline int local_bit(const TileGeom &g, int q) {
    if (q < g.low_bits) return q;
    for (int j = 0; j < g.n_high; j++)
        if (g.high[j] == q) return g.low_bits + j;
    return -1;
}

// TileBlock -> device TileOp.  Returns false when a qubit is on the wrong side of the tile or the block cannot be
// expressed (Scheduler::merge_blocks never produces such a block).  f32: the state holds fp32 amplitudes — 8-byte LDS slots,
// coefficients rounded once, here, and stored as the float pairs the fp32 kernels consume (kernels_impl.inc coef_t).
static bool to_tile_op(const TileGeom &g, const TileBlock &blk, TileOp &t, bool f32 = false) {
    memset(&t, 0, sizeof t);
    const int amp_shift = f32 ? 3 : 4;
    const int k = blk.nq, NB = blk.banks();
    if (k > kMaxOpQ || blk.ns > 2) return false;
    t.nsel = blk.ns;
    for (int a = 0; a < blk.ns; a++) {
        if (local_bit(g, blk.s[a]) >= 0 || blk.s[a] < 0 || blk.s[a] >= g.n) return false; // selectors lie outside the tile
        t.selbit[a] = blk.s[a];
    }
    // qbit[a]: tile-local bit of the block's a-th qubit in ascending GLOBAL order = bit a of a row / column index (the pair / quad
    // forms of tiny tiles get the same bits sorted ascending in t.b[]; the two orders agree while TileGeom::high is ascending and
    // differ once the engine reorders the tile bits).
    int qbit[kMaxOpQ] = {0, 0, 0, 0, 0, 0};
    uint32_t used = 0;
    for (int a = 0; a < k; a++) {
        const int lb = local_bit(g, blk.q[k - 1 - a]);
        if (lb < 0) return false;
        qbit[a] = lb;
        used |= 1u << lb;
    }
    auto is1 = [&](const cd &z) { return z.real() == 1.0 && z.imag() == 0.0; };
    // a coefficient in the form the kernels read it: fp64 (re, im); fp32 the pairs (ur, ui), (-ui, ur) in the same 16 bytes
    auto put = [&](double *slot, const cd &z) {
        if (!f32) { slot[0] = z.real(); slot[1] = z.imag(); return; }
        const float r = (float)z.real(), i = (float)z.imag();
        const float four[4] = {r, i, -i, r};
        memcpy(slot, four, sizeof four);
    };
    for (int v = 0; v < NB; v++)
        if (blk.bank_is_identity(v)) t.ident |= 1 << v;
    if (k == 0) { // tile-uniform factor
        if (blk.ns == 0) return false;
        t.kind = TOP_SCALE;
        for (int v = 0; v < NB; v++) {
            const cd z = blk.at(v, 0, 0);
            if (f32) { const float two[2] = {(float)z.real(), (float)z.imag()}; memcpy(t.scale[v], two, sizeof two); }
            else { t.scale[v][0] = z.real(); t.scale[v][1] = z.imag(); }
        }
        return true;
    }
    const int maxnnz = blk.max_row_nnz();
    if (maxnnz > 4) return false;
    if (g.tile_bits < 3) { // a register of one or two qubits: no three tile bits to pad a block to, the pair / quad forms stay
        for (int a = 0; a < k; a++) t.b[a] = (uint8_t)qbit[a];
        std::sort(t.b, t.b + k);
        t.nq = k;
        if (k == 1) {
            bool diag = true;
            for (int v = 0; v < NB; v++) diag = diag && blk.at(v, 0, 1) == cd(0, 0) && blk.at(v, 1, 0) == cd(0, 0);
            t.kind = diag ? TOP_DIAG1 : TOP_G1;
            for (int v = 0; v < NB; v++) {
                if (diag) {
                    put(&t.rec[v][0].coef[0], blk.at(v, 0, 0));
                    put(&t.rec[v][0].coef[2], blk.at(v, 1, 1));
                    t.rec[v][0].off[0] = is1(blk.at(v, 0, 0)) ? 1 : 0;
                } else {
                    for (int e = 0; e < 4; e++) put(&t.rec[v][0].coef[2 * e], blk.at(v, e >> 1, e & 1));
                }
            }
            return true;
        }
        if (k != 2) return false;
        t.kind = TOP_G2; // the kernel's index bit 0 is t.b[0], bit 1 is t.b[1]: swap the qubits' roles if the tile order did
        const bool swapped = qbit[0] > qbit[1];
        auto sw = [&](int i) { return swapped ? ((i & 1) << 1) | (i >> 1) : i; };
        for (int v = 0; v < NB; v++)
            for (int e = 0; e < 16; e++) put(&t.rec[v][0].coef[2 * e], blk.at(v, sw(e >> 2), sw(e & 3)));
        return true;
    }
    // rows laid out class by class (TileBlock::classes): T rows that read the same T operand slots
    int T = 1;
    std::vector<std::vector<int>> crows, ccols;
    if (k == 1) { // classes() speaks about blocks on two and more qubits; a 2x2 is one class of two rows, or two of one
        bool diag = true;
        for (int v = 0; v < NB; v++) diag = diag && blk.at(v, 0, 1) == cd(0, 0) && blk.at(v, 1, 0) == cd(0, 0);
        T = diag ? 1 : 2;
        crows.assign((size_t)NB, {0, 1});
        ccols.assign((size_t)NB, {0, 1});
    } else if (!blk.classes(T, crows, ccols)) return false;
    // Fewer than three qubits: pad with tile bits the block does not touch (it acts on them as the identity).  Same LDS
    // reads, multiply-adds and writes per amplitude as the pair / quad forms had, through the one code path of the part form.
    int K = k;
    for (int lb = 0; K < 3 && lb < g.tile_bits; lb++)
        if (!(used >> lb & 1u)) { qbit[K++] = lb; used |= 1u << lb; }
    if (K < 3) return false;
    const int pad = K - k, D = 1 << k, DK = 1 << K;
    t.kind = TOP_PART;
    t.nq = K;
    t.terms = T;
    {
        // Which free tile-local bit each bit of a lane's group index walks (kernels_impl.inc part_geometry; nibble a of b[0..4], 15
        // for the item bits that select the part).  Any assignment enumerates the groups; this one is chosen so that the lanes that
        // share an LDS cycle fall on different banks (MI355X_MICROARCH.md, LDS): a ds_read_b128 serves the 16 lanes of a 32-lane half
        // with lane bits l2 ^ l3 ^ l4 = const in one cycle when they hit 16 different 16-byte units of a 256-byte row, i.e. when the
        // (swizzled) unit images of the bits walked by l0, l1, l2 ^ l3, l2 ^ l4 are independent; a ds_write_b128 the 8 lanes of l0..l2
        // when theirs are independent modulo 8 units.  The layout swizzle makes that true for holes-free low bits; a block's qubits
        // punch holes, and the ascending assignment then collides for many hole patterns (24 % of the LDS-active cycles of the bench
        // schedule were bank conflicts).  fp32 states (8-byte slots) have other lane groups and their own conditions, below.
        int freeb[16], nf = 0;
        for (int lb = 0; lb < g.tile_bits; lb++)
            if (!(used >> lb & 1u)) freeb[nf++] = lb;
        auto image = [&](int b) -> uint32_t { // unit bits of the swizzled slot 1 << b (= sw_slot of kernels_impl.inc)
            if (f32) { // 8-byte slots: unit = slot bits 0..4
                if (b < 5) return 1u << b;
                if (b < 10) return (1u << (b - 5)) | (1u << ((b - 4) % 5));
                return (7u << (b - 10)) & 31u;
            }
            if (b < 4) return 1u << b;
            const int j = (b - 4) % 5;
            return j == 0 ? 15u : 1u << (j - 1);
        };
        auto rank_of = [](std::initializer_list<uint32_t> vs) { // rank of a few vectors of GF(2)^5
            uint32_t v[5] = {0, 0, 0, 0, 0};
            int n = 0, r = 0;
            for (uint32_t x : vs) v[n++] = x;
            for (int bit = 0; bit < 5; bit++) {
                int piv = -1;
                for (int i = r; i < n; i++)
                    if (v[i] >> bit & 1u) { piv = i; break; }
                if (piv < 0) continue;
                std::swap(v[r], v[piv]);
                for (int i = 0; i < n; i++)
                    if (i != r && (v[i] >> bit & 1u)) v[i] ^= v[r];
                r++;
            }
            return r;
        };
        auto rank4 = [&](uint32_t a, uint32_t b, uint32_t c, uint32_t d) { return rank_of({a, b, c, d}) == 4; };
        int order[16];
        for (int i = 0; i < nf; i++) order[i] = freeb[i];
        if (f32 && nf >= 5) {
            // fp32: a ds_read_b64 serves the 32 lanes of a half in one cycle when they hit 32 different 8-byte units of a 256-byte row
            // (the images of l0 .. l4 independent in GF(2)^5), a ds_write_b64 16 contiguous lanes out of a 128-byte row (l0 .. l3
            // independent modulo 16 units)
            int best[5] = {-1, -1, -1, -1, -1}, best_score = -1;
            for (int a0 = 0; a0 < nf && best_score < 2; a0++)
                for (int a1 = 0; a1 < nf && best_score < 2; a1++) {
                    if (a1 == a0) continue;
                    for (int a2 = 0; a2 < nf && best_score < 2; a2++) {
                        if (a2 == a0 || a2 == a1) continue;
                        for (int a3 = 0; a3 < nf && best_score < 2; a3++) {
                            if (a3 == a0 || a3 == a1 || a3 == a2) continue;
                            const uint32_t v0 = image(freeb[a0]), v1 = image(freeb[a1]), v2 = image(freeb[a2]), v3 = image(freeb[a3]);
                            if (rank_of({v0, v1, v2, v3}) < 4) continue;
                            const bool writes_ok = rank_of({v0 & 15u, v1 & 15u, v2 & 15u, v3 & 15u}) == 4;
                            for (int a4 = 0; a4 < nf && best_score < 2; a4++) {
                                if (a4 == a0 || a4 == a1 || a4 == a2 || a4 == a3) continue;
                                if (rank_of({v0, v1, v2, v3, image(freeb[a4])}) < 5) continue;
                                const int score = writes_ok ? 2 : 1;
                                if (score > best_score) { best_score = score; best[0] = a0; best[1] = a1; best[2] = a2; best[3] = a3; best[4] = a4; }
                            }
                        }
                    }
                }
            if (best_score > 0) {
                bool taken[16] = {false};
                int n_o = 0;
                for (int i = 0; i < 5; i++) { order[n_o++] = freeb[best[i]]; taken[best[i]] = true; }
                for (int i = 0; i < nf; i++)
                    if (!taken[i]) order[n_o++] = freeb[i];
            }
        }
        if (!f32 && nf >= 5) {
            int best[5] = {-1, -1, -1, -1, -1}, best_score = -1;
            for (int a0 = 0; a0 < nf && best_score < 2; a0++)
                for (int a1 = 0; a1 < nf && best_score < 2; a1++) {
                    if (a1 == a0) continue;
                    for (int a2 = 0; a2 < nf && best_score < 2; a2++) {
                        if (a2 == a0 || a2 == a1) continue;
                        const uint32_t v0 = image(freeb[a0]), v1 = image(freeb[a1]), v2 = image(freeb[a2]);
                        const bool writes_ok = rank4(v0 & 7u, v1 & 7u, v2 & 7u, 8u); // independent modulo 8 units
                        for (int a3 = 0; a3 < nf && best_score < 2; a3++) {
                            if (a3 == a0 || a3 == a1 || a3 == a2) continue;
                            for (int a4 = 0; a4 < nf && best_score < 2; a4++) {
                                if (a4 == a0 || a4 == a1 || a4 == a2 || a4 == a3) continue;
                                if (!rank4(v0, v1, v2 ^ image(freeb[a3]), v2 ^ image(freeb[a4]))) continue;
                                const int score = writes_ok ? 2 : 1;
                                if (score > best_score) { best_score = score; best[0] = a0; best[1] = a1; best[2] = a2; best[3] = a3; best[4] = a4; }
                            }
                        }
                    }
                }
            if (best_score > 0) { // the five lowest lane bits as chosen, the rest of the free bits ascending behind them
                bool taken[16] = {false};
                int n_o = 0;
                for (int i = 0; i < 5; i++) { order[n_o++] = freeb[best[i]]; taken[best[i]] = true; }
                for (int i = 0; i < nf; i++)
                    if (!taken[i]) order[n_o++] = freeb[i];
            }
        }
        uint64_t nib = 0;
        for (int a = 0; a < 10; a++) nib |= (uint64_t)(a < nf ? order[a] : 15) << (4 * a);
        for (int a = 0; a < 5; a++) t.b[a] = (uint8_t)(nib >> (8 * a));
        t.b[5] = t.b[6] = 0;
    }
    // LDS BYTE offset of a slot code (bit a of the code sits at tile-local bit qbit[a]), already passed through the
    // kernel's layout swizzle (kernels_impl.inc sw_slot: unit bits 0..3 ^= a linear image of the higher slot bits;
    // linear, so it commutes with the XOR the kernel combines it with)
    auto slot_off = [&](int code) {
        uint32_t o = 0;
        for (int a = 0; a < K; a++) o |= (uint32_t)((code >> a) & 1) << qbit[a];
        if (f32) { // = sw_fold of kernels_impl.inc for 8-byte slots: the unit is slot bits 0..4
            const uint32_t hi = o >> 5, a = hi & 31u, b = (hi >> 5) & 7u;
            const uint32_t fa = a ^ (((a << 1) | (a >> 4)) & 31u);
            const uint32_t fb = ((0u - (b & 1u)) & 7u) ^ ((0u - ((b >> 1) & 1u)) & 14u) ^ ((0u - ((b >> 2) & 1u)) & 28u);
            return (o ^ fa ^ fb) << amp_shift;
        }
        const uint32_t hi = o >> 4, f = (hi ^ (hi >> 5) ^ (hi >> 10)) & 31u; // = sw_fold of kernels_impl.inc for 16-byte slots
        return (o ^ (((f >> 1) & 15u) ^ ((0u - (f & 1u)) & 15u))) << amp_shift;
    };
    bool closed = true, skips = false;
    for (int v = 0; v < NB; v++)
        for (int p = 0; p < DK; p++) { // position p = copy (p / D) of the block over the padding bits, row crows[v][p % D] of it
            const int hi = (p / D) << k, r0 = crows[v][(size_t)(p % D)], r = hi | r0, c0 = ((p % D) / T) * T;
            PartRec &rec = t.rec[v][p / kPartRows];
            const int pp = p % kPartRows, cc = (pp / T) * T;
            rec.rowoff[pp] = slot_off(r);
            for (int j = 0; j < T; j++) {
                const int col = ccols[v][(size_t)(c0 + j)];
                rec.off[cc + j] = slot_off(hi | col); // the same list from every row of the class
                put(&rec.coef[(size_t)(pp * T + j) * 2], blk.at(v, r0, col)); // exact zero where the row does not use the column
            }
        }
    (void)pad;
    for (int v = 0; v < NB; v++)
        for (int part = 0; part < DK / kPartRows; part++) {
            PartRec &rec = t.rec[v][part];
            uint64_t reads = 0, writes = 0; // slot codes are < 64: compare the part's operand slots with the slots it writes
            for (int pp = 0; pp < kPartRows; pp++) {
                const int p = part * kPartRows + pp, hi = (p / D) << k;
                writes |= 1ULL << (hi | crows[v][(size_t)(p % D)]);
                reads |= 1ULL << (hi | ccols[v][(size_t)(p % D)]);
            }
            if (reads != writes) closed = false;
            for (int c = 0; c < kPartRows / T; c++) { // a class of identity rows only: nothing to do
                bool ident = true;
                for (int i = 0; i < T && ident; i++) {
                    const int p = part * kPartRows + c * T + i, r0 = crows[v][(size_t)(p % D)];
                    const TileBlock::Row &row = blk.row(v, r0);
                    ident = row.n == 1 && row.col[0] == r0 && is1(row.val[0]);
                }
                if (ident) { // the kernel asks off[] before its reads and rowoff[] before its writes
                    rec.off[c * T] = kSkipClass;
                    for (int i = 0; i < T; i++) rec.rowoff[c * T + i] = kSkipClass;
                    skips = true;
                }
            }
        }
    if (skips) t.flags |= kOpFlagSkips;
    if (closed) t.flags |= kOpFlagClosed;
    // what the kernel branches on, in the bit positions it uses (kernels_impl.inc PartPlan::info): log2 T, skips, barrier between reads and writes
    t.b[7] = (uint8_t)(((T == 4 ? 2 : T == 2 ? 1 : 0) << 1) | (skips ? 8 : 0) | ((K > 3 && !closed) ? 16 : 0));
    return true;
}

// Which role each high tile bit plays.  Tile-local bit L+j is global bit high[j], in ANY order (the blocks address LDS
// by tile-local bit and are translated through local_bit(), so the order is invisible to them); with 2^L <= 8 amplitudes
// per run and 512 threads, high[0..2] are walked by the lanes of a wave (the 8 runs one load instruction touches),
// high[3..5] by the waves of the workgroup, high[6..8] by the 8 registers of a lane.  The memory-only time of a pass
// depends on this order as much as on the set itself (n = 30, tools/geom_probe4.py: one set 6.56 ... 9.05 ms over 48
// random orders, ascending 7.67; another 8.40 ... 14.05, ascending 14.06) and no simple rule predicts it (a boosted-tree
// model on 3000 samples explains a third of the variance), so it is MEASURED: qsim_tune_circuit times candidate
// orders for every pass of a circuit's schedule and keeps the best in a process-wide table keyed by (register size,
// precision, tile shape, bit set) — planning in the sense of FFTW's wisdom, outside any timed region.  Untuned passes
// walk their bits in ascending order (what the scheduler emits).  QSIM_OPT_DEBUG_TILE_ORDER = k > 0 shuffles every
// pass's order with a generator seeded by k and the pass count instead (probes, and the parity tests of the reordering).
struct GeomKey {
    int n, f32, tile_bits, low_bits;
    uint64_t high_mask;
    bool operator<(const GeomKey &o) const {
        return std::tie(n, f32, tile_bits, low_bits, high_mask) < std::tie(o.n, o.f32, o.tile_bits, o.low_bits, o.high_mask);
    }
};
struct GeomOrder { int8_t high[kMaxTileHigh]; float ms, ms_ascending; };
static std::mutex g_wisdom_mu;
static std::map<GeomKey, GeomOrder> g_wisdom;
static std::atomic<uint64_t> g_wisdom_epoch{1}; // bumped whenever the table changes: cached plans carry the orders they were built with

static GeomKey geom_key(const qsim_state *s, const TileGeom &g) {
    GeomKey k{g.n, s->f32 ? 1 : 0, g.tile_bits, g.low_bits, 0};
    for (int j = 0; j < g.n_high; j++) k.high_mask |= 1ULL << g.high[j];
    return k;
}

static void shuffle_high(TileGeom &g, uint64_t seed) {
    uint64_t x = seed | 1ULL;
    for (int i = g.n_high - 1; i > 0; i--) { // Fisher-Yates with xorshift64*
        x ^= x >> 12; x ^= x << 25; x ^= x >> 27;
        const int j = (int)(((x * 0x2545F4914F6CDD1DULL) >> 33) % (uint64_t)(i + 1));
        std::swap(g.high[i], g.high[j]);
    }
}

static void order_tile_bits(qsim_state *s, TileGeom &g) {
    s->tile_passes++;
    if (g.n_high < 2) return;
    if (s->debug_tile_order > 0) {
        shuffle_high(g, 0x9E3779B97F4A7C15ULL * (uint64_t)(s->debug_tile_order + 1) + 0xD1B54A32D192ED03ULL * s->tile_passes);
        return;
    }
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    auto it = g_wisdom.find(geom_key(s, g));
    if (it != g_wisdom.end())
        for (int j = 0; j < g.n_high; j++) g.high[j] = it->second.high[j];
}

// The second buffer for out-of-place tile passes, or NULL when the passes of this state run in place.
// measured (bench circuits, same box, alternating runs): n = 24 -12 %, n = 25..28 +-0.2 %, n = 29 +1.4 %, n = 30 +1.6 %, n = 31 +1.5 %
constexpr size_t kPingPongMinBytes = (size_t)8 << 30;
static void *spare_buffer(qsim_state *s) {
    const size_t bytes = s->amp_bytes() << s->n;
    if (s->pingpong == 0 || (s->pingpong == 1 && bytes < kPingPongMinBytes)) return nullptr;
    if (s->spare) return s->spare;
    if (!s->owns || s->spare_failed) return nullptr; // an external buffer is only ever paired with a lent one
    size_t free_b = 0, total_b = 0;
    if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < bytes + total_b / 16 ||
        hipMalloc(&s->spare, bytes) != hipSuccess) {
        (void)hipGetLastError();
        s->spare = nullptr;
        s->spare_failed = true; // not enough memory for two copies: stay in place, do not ask again
        return nullptr;
    }
    s->owns_spare = true;
    return s->spare;
}

// The re-layout of an exchange done by the stores of the last tile pass in front of it (qsim_flush_pack; kernels_impl.inc PACK).
struct PackJob {
    PackMap map{};
    int bits[3] = {0, 0, 0};
    void *out = nullptr;    // where the caller wants the packed state (NULL: whichever of the state's two buffers it is not in)
    uint32_t skip = 0;      // blocks nobody will read (only the separate pack kernel leaves them out)
    uint64_t needed = ~0ULL; // source index bits that may be 1 where the receivers expect data
    void *packed_at = nullptr; // set when a tile pass did the re-layout: the buffer that now holds the packed state
};

// Launches a tile pass whose TileOps are already on the device (no statistics, no profiling events).  oop: write the
// state to the spare buffer and make that the state (the caller checked spare_buffer()).  job: the pass writes the state,
// re-laid-out, to job->out (or the buffer the state is not in) and records where.
static int launch_tile_prepared(qsim_state *s, const TileGeom &geom, const TileOp *d, int need, bool from_zero_ket, bool oop = false, uint64_t zero_mask = 0,
                                PackJob *job = nullptr) {
    LaunchCfg cfg{s->stream, s->grid_cap};
    const int threads = s->tile_threads; // 0: default for the tile size
    void *out = oop ? s->spare : s->amps;
    hipError_t e;
    if (job) {
        void *dst = job->out ? job->out : (s->spare && s->spare != s->amps ? s->spare : nullptr);
        if (!dst || dst == s->amps) return fail(QSIM_ERR_ARG, "internal: no buffer for the re-layout");
        e = launch_tile(cfg, s->amps, dst, s->f32, geom, d, need, threads, from_zero_ket, s->zero_ket_amp, false, zero_mask, &job->map);
        if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
        job->packed_at = dst;
        return QSIM_OK;
    }
    if (s->debug_skip_ops) {
        TileGeom bare = geom;
        bare.n_scale = 0;
        e = launch_tile(cfg, s->amps, out, s->f32, bare, d, 0, threads, from_zero_ket, s->zero_ket_amp, false, zero_mask);
    } else {
        e = launch_tile(cfg, s->amps, out, s->f32, geom, d, need, threads, from_zero_ket, s->zero_ket_amp, s->debug_skip_mem != 0, zero_mask);
    }
    if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    if (oop) std::swap(s->amps, s->spare);
    return QSIM_OK;
}

// Prepares the blocks of a tile pass for the given bit order in the pinned ring, uploads and launches them; `capture`
// (optional) receives a copy of the prepared TileOps for the plan cache.
static int launch_tile_pass(qsim_state *s, const Pass &p, const TileGeom &geom, bool from_zero_ket, std::vector<TileOp> *capture = nullptr, bool oop = false, uint64_t zero_mask = 0,
                            PackJob *job = nullptr) {
    const size_t need = p.blocks.size();
    if (need > s->ops_cap) return fail(QSIM_ERR_ARG, "tile pass with %zu ops exceeds the op buffer", need);
    if (s->ops_used + need > s->ops_cap) { // ring is full: wait until earlier passes have read their ops
        HIP_TRY(hipStreamSynchronize(s->stream));
        s->ops_used = 0;
    }
    TileOp *h = s->h_ops + s->ops_used;
    for (size_t k = 0; k < need; k++)
        if (!to_tile_op(geom, p.blocks[k], h[k], s->f32)) return fail(QSIM_ERR_ARG, "internal: block does not fit its tile pass");
    if (capture) capture->insert(capture->end(), h, h + need);
    TileOp *d = s->d_ops + s->ops_used;
    HIP_TRY(hipMemcpyAsync(d, h, need * sizeof(TileOp), hipMemcpyHostToDevice, s->stream));
    s->ops_used += need;
    return launch_tile_prepared(s, geom, d, (int)need, from_zero_ket, oop, zero_mask, job);
}

// cached_geom / cached_ops: replay of a cached plan (the tile pass's order and device-resident TileOps);
// capture / geom_out: the first run of a plan records them.
static int launch_pass(qsim_state *s, const Pass &p, const TileGeom *cached_geom = nullptr, const TileOp *cached_ops = nullptr,
                       std::vector<TileOp> *capture = nullptr, TileGeom *geom_out = nullptr, bool oop = false, PackJob *job = nullptr) {
    const bool from_zero_ket = s->zero_ket_pending && p.kclass == QSIM_K_TILE;
    if ((s->zero_ket_pending || s->partial) && p.kclass != QSIM_K_TILE) { // only tile passes work on a partially written state
        const int rc = materialize_zero_ket(s);
        if (rc) return rc;
    }
    s->zero_ket_pending = false;
    LaunchCfg cfg{s->stream, s->grid_cap};
    const FusedOp &op = p.ops[0];
    hipError_t e = hipSuccess;
    double visited = 1.0; // fraction of the tiles a tile pass works on
    switch (p.kclass) {
    case QSIM_K_GATE1:
    case QSIM_K_GATE1_LO: {
        M2 u;
        to_m2(op, u);
        LaunchScope scope(s, p.kclass);
        e = launch_gate1(cfg, s->amps, s->f32, s->n, op.q_hi, u);
        break;
    }
    case QSIM_K_PHASE: {
        LaunchScope scope(s, p.kclass);
        if (p.diag_full)
            e = launch_diag1_full(cfg, s->amps, s->f32, s->n, op.q_hi, op.m[0].real(), op.m[0].imag(), op.m[3].real(),
                                  op.m[3].imag());
        else
            e = launch_phase(cfg, s->amps, s->f32, s->n, op.q_hi, op.m[3].real(), op.m[3].imag());
        break;
    }
    case QSIM_K_CX: {
        LaunchScope scope(s, p.kclass);
        e = launch_cx(cfg, s->amps, s->f32, s->n, op.q_hi, op.q_lo);
        break;
    }
    case QSIM_K_GATE2: {
        M4 u;
        to_m4(op, u);
        LaunchScope scope(s, p.kclass);
        e = launch_gate2(cfg, s->amps, s->f32, s->n, op.q_hi, op.q_lo, u);
        break;
    }
    case QSIM_K_TILE: {
        TileGeom geom = cached_geom ? *cached_geom : p.geom;
        if (!cached_geom) order_tile_bits(s, geom);
        if (geom_out) *geom_out = geom;
        uint64_t hm = 0, oc = 0;
        for (int j = 0; j < geom.n_high; j++) { hm |= 1ULL << geom.high[j]; oc |= (uint64_t)geom.high[j] << (5 * j); }
        // the part of the register this pass has to visit (qsim_state::support)
        const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
        const uint64_t tmask = hm | ((1ULL << geom.low_bits) - 1ULL);
        uint64_t zero_mask = 0;
        if (s->sparse_start && from_zero_ket) zero_mask = nmask;
        else if (s->partial) zero_mask = nmask & ~s->support;
        visited = 1.0 / (double)(1ULL << __builtin_popcountll(zero_mask & ~tmask));
        LaunchScope scope(s, p.kclass, (int)p.blocks.size(), hm, oc, visited);
        if (scope.on && s->profile >= 2) // what the blocks look like, for the pass-time model's data (tools/pass_model_data.py); host work per launch: only on request
            for (size_t k = (size_t)geom.n_scale; k < p.blocks.size(); k++) {
                const TileBlock &b = p.blocks[k];
                int T = 0;
                std::vector<std::vector<int>> rows, cols;
                if (!b.classes(T, rows, cols)) T = 4;
                int ident = 0; // rows that are identity in every bank
                for (int r = 0; r < b.dim(); r++) {
                    bool id = true;
                    for (int v = 0; v < b.banks() && id; v++) { const auto &rw = b.row(v, r); id = rw.n == 1 && rw.col[0] == r && rw.val[0] == cd(1, 0); }
                    ident += id;
                }
                scope.pe.forms.push_back((uint8_t)((T == 4 ? 2 : T == 2 ? 1 : 0) | (b.nq << 2) | (2 * ident >= b.dim() ? 32 : 0) | (b.ns << 6)));
            }
        const int rc = cached_ops ? launch_tile_prepared(s, geom, cached_ops, (int)p.blocks.size(), from_zero_ket, oop, zero_mask, job)
                                  : launch_tile_pass(s, p, geom, from_zero_ket, capture, oop, zero_mask, job);
        if (rc) return rc;
        if (zero_mask) {
            s->support = (from_zero_ket ? 0 : s->support) | tmask;
            s->partial = (s->support & nmask) != nmask;
        }
        break;
    }
    default: return fail(QSIM_ERR_ARG, "internal: unknown kernel class %d", p.kclass);
    }
    if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "kernel launch failed: %s", hipGetErrorString(e));
    const double scale = s->f32 ? 0.5 : 1.0; // the scheduler prices passes for 16-byte amplitudes
    account(s, p.kclass, scale * visited * (from_zero_ket ? p.bytes / 2 : p.bytes)); // a generating pass only writes
    return QSIM_OK;
}

// Qubits that may be 1 somewhere in the state when the next pass runs (SchedConfig::initial_support).
static uint64_t current_support(const qsim_state *s) {
    if (!s->sparse_start) return ~0ULL;
    if (s->zero_ket_pending) return 0;
    return s->partial ? s->support : ~0ULL;
}

// The identity of a schedule (PlanIdentity) without the gates, and its 64-bit name: FNV-1a over the options that shape a
// plan, the state's support, the QSIM_SCHED_* overrides and every gate.  The key only FINDS cached plans and scheduler
// hints; a plan is replayed only after plan_matches() has compared the identity itself.
static PlanIdentity plan_identity(const qsim_state *s, size_t count, uint64_t support) {
    PlanIdentity id;
    const int opts[8] = {s->n, s->f32 ? 1 : 0, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, (int)count};
    memcpy(id.opts, opts, sizeof opts);
    id.support = support; // the schedule depends on where the state is known to be zero
    id.env = read_sched_env();
    return id;
}
static uint64_t gates_key(const qsim_state *s, const PlanIdentity &id, const QueuedGate *gates, size_t count) {
    if (s->debug_plan_key) return (uint64_t)s->debug_plan_key;
    uint64_t h = 0xcbf29ce484222325ULL;
    auto mix = [&](const void *p, size_t n) {
        const unsigned char *b = (const unsigned char *)p;
        for (size_t i = 0; i < n; i++) { h ^= b[i]; h *= 0x100000001b3ULL; }
    };
    mix(id.opts, sizeof id.opts);
    mix(&id.support, sizeof id.support);
    const int env_i[9] = {(int)id.env.set, id.env.lookahead, id.env.rollout, id.env.window, id.env.local_iters, id.env.objective, id.env.merge, id.env.merge_qubits, id.env.cap};
    mix(env_i, sizeof env_i);
    mix(&id.env.cheap_margin, sizeof id.env.cheap_margin);
    for (size_t i = 0; i < count; i++) {
        const QueuedGate &g = gates[i];
        const int hd[3] = {g.kind, g.q0, g.q1};
        mix(hd, sizeof hd);
        if (g.kind != QSIM_GATE_CX) mix(g.m, (g.kind == QSIM_GATE_U1 ? 4 : 16) * sizeof(cd));
    }
    return h;
}
static bool plan_matches(const PlanIdentity &have, const PlanIdentity &want, const QueuedGate *gates, size_t count) {
    return memcmp(have.opts, want.opts, sizeof have.opts) == 0 && have.support == want.support && have.env == want.env &&
           have.gates.size() == count && same_gates(have.gates.data(), gates, count);
}

// Scheduler variant per circuit, decided by the planning step (qsim_tune_circuit): key -> SchedConfig::commute.  Circuits
// that were never planned use the default.
// The table is found by key alone: a colliding circuit would be scheduled with another circuit's variant — a valid schedule
// either way (every variant is; the results never depend on it).  Bounded: beyond kMaxSchedHints circuits it starts afresh.
struct SchedHint { int commute; double cheap_margin; int lookahead; int cap; /* clusters per pass; 0: the configuration's own */ uint64_t seed; /* SchedConfig::seed */ };
static std::mutex g_hints_mu;
static std::map<uint64_t, SchedHint> g_sched_hints;
constexpr size_t kMaxSchedHints = 4096;
static void apply_sched_hint(uint64_t key, SchedConfig &cfg) {
    std::lock_guard<std::mutex> lock(g_hints_mu);
    auto it = g_sched_hints.find(key);
    if (it == g_sched_hints.end()) return;
    cfg.commute = it->second.commute;
    cfg.cheap_margin = it->second.cheap_margin;
    cfg.lookahead = it->second.lookahead;
    if (it->second.cap > 0) { cfg.tile_max_ops = it->second.cap; cfg.tail_max_ops = std::max(cfg.tail_max_ops, it->second.cap); }
    cfg.seed = it->second.seed;
}
static bool have_sched_hints() {
    std::lock_guard<std::mutex> lock(g_hints_mu);
    return !g_sched_hints.empty();
}

// A tile pass can take the re-layout on board when the kernel has that variant for its shape (fp64, 2^12-amplitude tiles, 512
// threads) and when what it writes covers what the receivers will look at: a pass over a partially written state only
// visits the tiles inside (support | its own tile bits), so source indices outside that never reach the output — fine as
// long as job->needed (where the receivers expect data) lies inside it; else the pack kernel does the job (it writes zeros).
// support_before: where the state can be non-zero when the pass starts (current_support() once every earlier pass has been launched).
static bool pass_can_pack(const qsim_state *s, const Pass &p, const TileGeom &geom, const PackJob *job, uint64_t support_before) {
    static const bool trace = getenv("QSIM_TRACE_PACK") != nullptr; // says on stderr why a re-layout got its own sweep
    if (p.kclass != QSIM_K_TILE || s->debug_skip_ops || s->debug_skip_mem || !launch_tile_can_pack(s->f32, geom, s->tile_threads)) {
        if (trace) fprintf(stderr, "qsim: re-layout not fused: last pass is %s\n", p.kclass != QSIM_K_TILE ? "no tile pass" : "a tile pass without the packing variant");
        return false;
    }
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    uint64_t after = (1ULL << geom.low_bits) - 1ULL;
    for (int j = 0; j < geom.n_high; j++) after |= 1ULL << geom.high[j];
    after |= support_before;
    if (trace && ((job->needed & nmask) & ~after) != 0)
        fprintf(stderr, "qsim: re-layout not fused: the pass writes support %llx, the receivers look at %llx\n", (unsigned long long)(after & nmask), (unsigned long long)(job->needed & nmask));
    return ((job->needed & nmask) & ~after) == 0;
}
static uint64_t current_support(const qsim_state *s);

// job != NULL: if the LAST pass of the queue is a tile pass that can do it, that pass writes the state re-laid-out (job->packed_at
// says where) and the state's own buffers are left holding stale data; otherwise everything runs as usual and job->packed_at
// stays NULL (the caller then runs the pack kernel).
static int flush_impl(qsim_state *s, PackJob *job) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    { const int rc = await_buffer(s); if (rc) return rc; } // (everything that looks at the buffer flushes first: the one place to wait for qsim_create_async)
    if (s->queue.empty()) return QSIM_OK;
    if (s->zero_ket_pending && s->zero_ket_amp == 0.0) { // the all-zero vector (a shard that holds nothing yet): every gate maps it to itself
        s->queue.clear();
        return QSIM_OK;
    }
    if (s->tile_bits - s->tile_low_bits < 2 || s->tile_bits - s->tile_low_bits > kMaxTileHigh)
        return fail(QSIM_ERR_ARG, "tile_bits - tile_low_bits must be in 2..%d", kMaxTileHigh);
    HIP_TRY(hipSetDevice(s->device));
    constexpr size_t kMaxPlans = 8, kMaxCachedOps = 4096;
    const bool cacheable = s->plan_cache && s->fuse >= 3 && s->debug_tile_order == 0 && s->queue.size() >= 8;
    const bool hinted = s->fuse >= 3 && have_sched_hints();
    PlanIdentity ident;
    uint64_t key = 0;
    if (cacheable || hinted) {
        ident = plan_identity(s, s->queue.size(), current_support(s));
        key = gates_key(s, ident, s->queue.data(), s->queue.size());
    }
    const uint64_t epoch = g_wisdom_epoch.load();
    if (cacheable) {
        for (CachedPlan &pl : s->plans) {
            if (pl.key != key || pl.wisdom_epoch != epoch) continue;
            if (!plan_matches(pl.id, ident, s->queue.data(), s->queue.size())) { s->plan_key_collisions++; continue; } // same name, other circuit
            s->plan_hits++;
            pl.last_use = ++s->plan_clock;
            s->queue.clear();
            // out-of-place tile passes come in pairs (the state ends where it started): with an odd count the last one stays in place
            size_t tiles = 0, seen = 0;
            for (const Pass &p : pl.passes) tiles += p.kclass == QSIM_K_TILE;
            bool fuse_pack = false;
            if (job && !pl.passes.empty()) {
                uint64_t sup = current_support(s); // ... as the last pass will find it
                for (size_t i = 0; i + 1 < pl.passes.size(); i++) {
                    if (pl.passes[i].kclass != QSIM_K_TILE) { sup = ~0ULL; break; }
                    sup |= (1ULL << pl.geoms[i].low_bits) - 1ULL;
                    for (int j = 0; j < pl.geoms[i].n_high; j++) sup |= 1ULL << pl.geoms[i].high[j];
                }
                fuse_pack = pass_can_pack(s, pl.passes.back(), pl.geoms.back(), job, sup);
            }
            if (fuse_pack) tiles--; // the last tile pass writes the re-layout: the ones before it bring the state home
            const bool pp = tiles >= 2 && spare_buffer(s) != nullptr;
            for (size_t i = 0; i < pl.passes.size(); i++) {
                const Pass &p = pl.passes[i];
                int rc;
                if (fuse_pack && i + 1 == pl.passes.size()) {
                    rc = launch_pass(s, p, &pl.geoms[i], pl.d_ops + pl.op_first[i], nullptr, nullptr, false, job);
                } else if (p.kclass == QSIM_K_TILE) {
                    seen++;
                    rc = launch_pass(s, p, &pl.geoms[i], pl.d_ops + pl.op_first[i], nullptr, nullptr, pp && !(seen == tiles && (tiles & 1)));
                } else {
                    rc = launch_pass(s, p);
                }
                if (rc) return rc;
            }
            return QSIM_OK;
        }
    }
    SchedConfig scfg = sched_config(s->n, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, s->f32, current_support(s));
    if (hinted) apply_sched_hint(key, scfg);
    Scheduler sched(scfg);
    for (const QueuedGate &g : s->queue) {
        if (g.kind == QSIM_GATE_U1) sched.add_1q(g.m, g.q0);
        else if (g.kind == QSIM_GATE_CX) sched.add_cx(g.q0, g.q1);
        else sched.add_2q(g.m, g.q0, g.q1);
    }
    int rc = QSIM_OK;
    CachedPlan fresh;
    if (cacheable) { fresh.id = std::move(ident); fresh.id.gates = std::move(s->queue); } // the plan remembers what it was built from
    s->queue.clear();
    std::vector<TileOp> host_ops;
    auto launch = [&](Pass &&p, bool oop, PackJob *pj = nullptr) {
        if (rc != QSIM_OK) return;
        if (!cacheable) { rc = launch_pass(s, p, nullptr, nullptr, nullptr, nullptr, oop, pj); return; }
        TileGeom g = p.geom;
        fresh.op_first.push_back(host_ops.size());
        rc = launch_pass(s, p, nullptr, nullptr, &host_ops, &g, oop, pj);
        fresh.geoms.push_back(g);
        fresh.passes.push_back(std::move(p));
    };
    // Passes are launched as soon as they are scheduled — the GPU works while later passes are planned — except that
    // with two buffers (or a re-layout to do) the most recent tile pass and whatever followed it are held back until the
    // next tile pass arrives: only then is it known not to be the last one.  The last one brings the state back to its own
    // buffer — or, when it is the very last pass and a re-layout is asked for, writes the state re-laid-out (job); a tile
    // pass that may turn out to be that one reads the state from its own buffer, so the one before it has to lead home,
    // which is why, with a job, TWO tile passes are held back.
    void *const home = s->amps;
    int pp = -1; // second buffer available?  asked when the first tile pass arrives (a queue without tile passes allocates nothing)
    std::vector<Pass> held, held2; // a tile pass, then the non-tile passes scheduled after it; held2: the tile pass before that one (job only)
    auto run_group = [&](std::vector<Pass> &grp, bool oop, PackJob *pj) {
        for (size_t i = 0; i < grp.size(); i++) launch(std::move(grp[i]), i == 0 && oop, i == 0 ? pj : nullptr);
        grp.clear();
    };
    auto release = [&](bool last) {
        if (!job) { // the last tile pass: out of place only if that leads home
            run_group(held, pp > 0 && (!last || s->amps != home), nullptr);
            return;
        }
        if (!last) { // a newer tile pass exists: held2 is neither last nor second to last
            run_group(held2, pp > 0, nullptr);
            held2.swap(held);
            return;
        }
        // the end of the queue: held2 (if any) is the second-to-last tile pass, held the last one
        bool fuse_last = false;
        if (held.size() == 1) { // ... and the very last pass
            uint64_t sup = current_support(s);
            if (!held2.empty()) { // not launched yet
                if (held2.size() > 1) sup = ~0ULL;
                sup |= (1ULL << held2[0].geom.low_bits) - 1ULL;
                for (int j = 0; j < held2[0].geom.n_high; j++) sup |= 1ULL << held2[0].geom.high[j];
            }
            fuse_last = pass_can_pack(s, held[0], held[0].geom, job, sup);
        }
        if (fuse_last) {
            run_group(held2, pp > 0 && s->amps != home, nullptr); // must lead home: the packing pass reads the state's own buffer
            if (s->amps != home && rc == QSIM_OK) { // no second-to-last pass to bring it home (cannot happen: out-of-place passes before came in pairs)
                rc = fail(QSIM_ERR_ARG, "internal: state not in its own buffer before the re-layout");
                return;
            }
            run_group(held, false, job);
        } else {
            run_group(held2, pp > 0, nullptr);
            run_group(held, pp > 0 && s->amps != home, nullptr);
        }
    };
    sched.finish([&](Pass &&p) {
        if (rc != QSIM_OK) return;
        if (p.kclass == QSIM_K_TILE && pp < 0) pp = spare_buffer(s) != nullptr ? 1 : 0;
        if (pp <= 0 && !job) { launch(std::move(p), false); return; }
        if (p.kclass == QSIM_K_TILE) release(false);
        if (p.kclass == QSIM_K_TILE || !held.empty()) held.push_back(std::move(p));
        else launch(std::move(p), false);
    });
    release(true);
    if (s->amps != home) std::swap(s->amps, s->spare); // only after a failed launch: the buffers keep their roles
    if (rc != QSIM_OK || !cacheable || host_ops.size() > kMaxCachedOps) return rc;
    // keep the plan: its TileOps move to a device buffer of their own (one copy, ordered behind the launches above)
    if (!host_ops.empty()) {
        if (hipMalloc((void **)&fresh.d_ops, host_ops.size() * sizeof(TileOp)) != hipSuccess) { (void)hipGetLastError(); return QSIM_OK; }
        if (hipMemcpy(fresh.d_ops, host_ops.data(), host_ops.size() * sizeof(TileOp), hipMemcpyHostToDevice) != hipSuccess) {
            (void)hipFree(fresh.d_ops);
            (void)hipGetLastError();
            return QSIM_OK;
        }
    }
    fresh.key = key;
    fresh.wisdom_epoch = epoch;
    fresh.last_use = ++s->plan_clock;
    size_t slot = s->plans.size();
    for (size_t i = 0; i < s->plans.size(); i++) // a stale plan of the same queue (the geometry table changed) is replaced
        if (s->plans[i].key == key && plan_matches(s->plans[i].id, fresh.id, fresh.id.gates.data(), fresh.id.gates.size())) slot = i;
    if (slot == s->plans.size() && s->plans.size() >= kMaxPlans) {
        slot = 0;
        for (size_t i = 1; i < s->plans.size(); i++)
            if (s->plans[i].last_use < s->plans[slot].last_use) slot = i;
    }
    if (slot < s->plans.size()) {
        if (s->plans[slot].d_ops) {
            HIP_TRY(hipStreamSynchronize(s->stream)); // a replay of the evicted plan may still be reading its ops
            (void)hipFree(s->plans[slot].d_ops);
        }
        s->plans[slot] = std::move(fresh);
    } else {
        s->plans.push_back(std::move(fresh));
    }
    return QSIM_OK;
}

extern "C" int qsim_flush(qsim_state *s) { return flush_impl(s, nullptr); }

extern "C" int qsim_plan_cache_stats(const qsim_state *s, uint64_t *plans, uint64_t *replays, uint64_t *key_collisions) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    if (plans) *plans = s->plans.size();
    if (replays) *replays = s->plan_hits;
    if (key_collisions) *key_collisions = s->plan_key_collisions;
    return QSIM_OK;
}

extern "C" int qsim_sync(qsim_state *s) {
    int rc = qsim_flush(s);
    if (rc == QSIM_OK) rc = materialize_zero_ket(s); // nothing consumed the pending |0...0>: write it now
    if (rc) return rc;
    HIP_TRY(hipStreamSynchronize(s->stream));
    return QSIM_OK;
}

// ---- amplitudes ----------------------------------------------------------------------------------------
// `out` holds m doubles' worth of room and m floats in its second half: widen them front to back (element i is read
// from byte 4m + 4i before byte 8i is written, and no later element starts below 8i + 8).
static void widen_in_place(double *out, uint64_t m) {
    const char *src = reinterpret_cast<const char *>(out) + 4 * m;
    for (uint64_t i = 0; i < m; i++) {
        float f;
        memcpy(&f, src + 4 * i, 4);
        out[i] = (double)f;
    }
}

extern "C" int qsim_read(qsim_state *s, uint64_t first, uint64_t count, double *out) {
    if (!s || !out) return fail(QSIM_ERR_ARG, "NULL argument");
    const uint64_t N = 1ULL << s->n;
    if (first > N || count > N - first) return fail(QSIM_ERR_ARG, "read range outside the state");
    const int rc = qsim_sync(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    if (!count) return QSIM_OK;
    if (!s->f32) {
        HIP_TRY(hipMemcpy(out, (const char *)s->amps + first * 16, count * 16, hipMemcpyDeviceToHost));
        return QSIM_OK;
    }
    // fp32 state: the API stays double; copy into the second half of the output and widen in place, front to back
    char *tmp = reinterpret_cast<char *>(out) + 8 * count;
    HIP_TRY(hipMemcpy(tmp, (const char *)s->amps + first * 8, count * 8, hipMemcpyDeviceToHost));
    widen_in_place(out, 2 * count);
    return QSIM_OK;
}

extern "C" int qsim_write(qsim_state *s, uint64_t first, uint64_t count, const double *in) {
    if (!s || !in) return fail(QSIM_ERR_ARG, "NULL argument");
    const uint64_t N = 1ULL << s->n;
    if (first > N || count > N - first) return fail(QSIM_ERR_ARG, "write range outside the state");
    const int rc = qsim_sync(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    if (!count) return QSIM_OK;
    if (!s->f32) {
        HIP_TRY(hipMemcpy((char *)s->amps + first * 16, in, count * 16, hipMemcpyHostToDevice));
        return QSIM_OK;
    }
    std::vector<float> tmp(2 * count);
    for (uint64_t i = 0; i < 2 * count; i++) tmp[i] = (float)in[i];
    HIP_TRY(hipMemcpy((char *)s->amps + first * 8, tmp.data(), count * 8, hipMemcpyHostToDevice));
    return QSIM_OK;
}

extern "C" int qsim_norm2(qsim_state *s, double *out) {
    if (!s || !out) return fail(QSIM_ERR_ARG, "NULL argument");
    int rc = qsim_flush(s);
    if (rc == QSIM_OK) rc = materialize_zero_ket(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    HIP_TRY(hipMemsetAsync(s->d_scalar, 0, 8, s->stream));
    LaunchCfg cfg{s->stream, s->grid_cap};
    HIP_TRY(launch_norm2(cfg, s->amps, s->f32, s->n, s->d_scalar));
    HIP_TRY(hipMemcpyAsync(out, s->d_scalar, 8, hipMemcpyDeviceToHost, s->stream));
    HIP_TRY(hipStreamSynchronize(s->stream));
    return QSIM_OK;
}

// ---- measurement post-path ----------------------------------------------------------------------------------
extern "C" double qsim_draw_randn(void) { // measurement, quantum_simulator.c:271-276
    double randn = 0.0, coeff = 1.0 / RAND_MAX;
    for (int i = 0; i < 10; i++) {
        randn += rand() * coeff;
        coeff *= 1.0 / RAND_MAX;
    }
    return randn;
}

extern "C" void qsim_putb(long long n, int len, char *buf) { // putb, quantum_simulator.c:285-293
    if (!buf || len < 0) return;
    for (int k = 0; k < len; k++) buf[k] = ((n >> (len - 1 - k)) & 1) ? '1' : '0';
    buf[len] = 0;
}

extern "C" int qsim_sample(qsim_state *s, const double *randoms, long shots, uint64_t *out) {
    if (!s || (shots > 0 && (!randoms || !out))) return fail(QSIM_ERR_ARG, "NULL argument");
    int rc = qsim_flush(s);
    if (rc == QSIM_OK) rc = materialize_zero_ket(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    constexpr int kBlockBits = 12;
    const uint64_t N = 1ULL << s->n;
    const int bb = s->n < kBlockBits ? s->n : kBlockBits;
    const uint64_t nblocks = N >> bb, bsize = 1ULL << bb;
    double *d_part = nullptr;
    HIP_TRY(hipMalloc((void **)&d_part, nblocks * sizeof(double)));
    LaunchCfg cfg{s->stream, s->grid_cap};
    hipError_t e = launch_block_prob(cfg, s->amps, s->f32, s->n, bb, d_part);
    std::vector<double> prefix(nblocks);
    if (e == hipSuccess) e = hipMemcpyAsync(prefix.data(), d_part, nblocks * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_part);
    if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "qsim_sample: %s", hipGetErrorString(e));
    double acc = 0.0;
    for (uint64_t b = 0; b < nblocks; b++) { acc += prefix[b]; prefix[b] = acc; } // cumulative at the END of block b

    std::vector<double> blk(2 * bsize);
    uint64_t cached = ~0ULL;
    for (long k = 0; k < shots; k++) {
        const double r = randoms[k];
        // first block whose end value is non-zero and >= r (quantum_simulator.c:279: skip while == 0 or < r)
        uint64_t lo = 0, hi = nblocks;
        while (lo < hi) {
            const uint64_t mid = (lo + hi) >> 1;
            if (prefix[mid] == 0.0 || prefix[mid] < r) lo = mid + 1;
            else hi = mid;
        }
        uint64_t idx = N - 1;
        bool found = false;
        for (uint64_t b = lo; b < nblocks && !found; b++) { // normally one block; rounding can push it to the next
            if (b != cached) {
                if (!s->f32) {
                    HIP_TRY(hipMemcpy(blk.data(), (const char *)s->amps + b * bsize * 16, bsize * 16, hipMemcpyDeviceToHost));
                } else {
                    char *tmp = reinterpret_cast<char *>(blk.data()) + 8 * bsize;
                    HIP_TRY(hipMemcpy(tmp, (const char *)s->amps + b * bsize * 8, bsize * 8, hipMemcpyDeviceToHost));
                    widen_in_place(blk.data(), 2 * bsize);
                }
                cached = b;
            }
            double c = b ? prefix[b - 1] : 0.0;
            for (uint64_t i = 0; i < bsize; i++) {
                c += blk[2 * i] * blk[2 * i] + blk[2 * i + 1] * blk[2 * i + 1];
                if (!(c == 0.0 || c < r)) { idx = b * bsize + i; found = true; break; }
            }
        }
        out[k] = idx;
    }
    return QSIM_OK;
}

// Block sums and block contents for index sets that are bit-deposits rather than ranges (what a permuted qubit map of a
// sharded state needs for the measurement post-path; see k_block_prob_masked).
extern "C" int qsim_block_prob_masked(qsim_state *s, uint64_t hi_mask, uint64_t lo_mask, double *out) {
    if (!s || !out) return fail(QSIM_ERR_ARG, "NULL argument");
    const uint64_t all = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    if ((hi_mask & lo_mask) || ((hi_mask | lo_mask) & ~all)) return fail(QSIM_ERR_ARG, "masks must be disjoint and inside the state");
    int rc = qsim_flush(s);
    if (rc == QSIM_OK) rc = materialize_zero_ket(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    const uint64_t nblocks = 1ULL << __builtin_popcountll(hi_mask);
    double *d_part = nullptr;
    HIP_TRY(hipMalloc((void **)&d_part, nblocks * sizeof(double)));
    LaunchCfg cfg{s->stream, s->grid_cap};
    hipError_t e = launch_block_prob_masked(cfg, s->amps, s->f32, hi_mask, lo_mask, d_part);
    if (e == hipSuccess) e = hipMemcpyAsync(out, d_part, nblocks * sizeof(double), hipMemcpyDeviceToHost, s->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_part);
    if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "qsim_block_prob_masked: %s", hipGetErrorString(e));
    return QSIM_OK;
}

extern "C" int qsim_gather_masked(qsim_state *s, uint64_t base, uint64_t lo_mask, double *out) {
    if (!s || !out) return fail(QSIM_ERR_ARG, "NULL argument");
    const uint64_t all = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    if ((base & lo_mask) || ((base | lo_mask) & ~all)) return fail(QSIM_ERR_ARG, "base and mask must be disjoint and inside the state");
    const uint64_t count = 1ULL << __builtin_popcountll(lo_mask);
    if (count > (1ULL << 24)) return fail(QSIM_ERR_ARG, "gather of %llu amplitudes is not a block", (unsigned long long)count);
    int rc = qsim_flush(s);
    if (rc == QSIM_OK) rc = materialize_zero_ket(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    void *d_buf = nullptr;
    HIP_TRY(hipMalloc(&d_buf, count * s->amp_bytes()));
    LaunchCfg cfg{s->stream, s->grid_cap};
    hipError_t e = launch_gather_masked(cfg, s->amps, s->f32, base, lo_mask, d_buf);
    if (e == hipSuccess) {
        if (!s->f32) e = hipMemcpyAsync(out, d_buf, count * 16, hipMemcpyDeviceToHost, s->stream);
        else e = hipMemcpyAsync(reinterpret_cast<char *>(out) + 8 * count, d_buf, count * 8, hipMemcpyDeviceToHost, s->stream);
    }
    if (e == hipSuccess) e = hipStreamSynchronize(s->stream);
    (void)hipFree(d_buf);
    if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "qsim_gather_masked: %s", hipGetErrorString(e));
    if (s->f32) widen_in_place(out, 2 * count);
    return QSIM_OK;
}

// keep_partial: a state that has only been written inside its support is packed as it is — amplitudes outside the support
// are packed as zeros without being loaded (k_pack zero_mask) — instead of having the zeros written out first.
static int pack_common(qsim_state *s, const int *bits, int nbits, void *dst, void *const *blocks, bool keep_partial, uint32_t skip_blocks) {
    if (!s || !bits || (!dst && !blocks)) return fail(QSIM_ERR_ARG, "NULL argument");
    if (nbits < 1 || nbits > (blocks ? 3 : 8) || nbits > s->n) return fail(QSIM_ERR_ARG, "pack: %d bits unsupported", nbits);
    for (int j = 0; j < nbits; j++)
        if (bits[j] < 0 || bits[j] >= s->n || (j && bits[j] <= bits[j - 1]))
            return fail(QSIM_ERR_ARG, "pack: bit positions must be ascending and inside the shard");
    const size_t blk = (s->amp_bytes() << s->n) >> nbits;
    for (int b = 0; b < (blocks ? 1 << nbits : 0); b++) {
        if ((skip_blocks >> b) & 1u) continue;
        if (!blocks[b]) return fail(QSIM_ERR_ARG, "pack: destination block %d is NULL", b);
        const char *p = (const char *)blocks[b], *a = (const char *)s->amps;
        if (p < a + (s->amp_bytes() << s->n) && a < p + blk) return fail(QSIM_ERR_ARG, "pack: a destination block overlaps the state");
    }
    if (dst == s->amps) return fail(QSIM_ERR_ARG, "pack: dst must not alias the state");
    int rc = qsim_flush(s);
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    const bool as_is = keep_partial && !s->zero_ket_pending && s->partial;
    if (rc == QSIM_OK && !as_is) rc = materialize_zero_ket(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    LaunchCfg cfg{s->stream, s->grid_cap};
    hipError_t e;
    {
        LaunchScope scope(s, QSIM_K_PACK);
        e = launch_pack(cfg, s->amps, dst, blocks, s->f32, s->n, bits, nbits, skip_blocks, as_is ? nmask & ~s->support : 0);
    }
    if (e != hipSuccess) return fail(QSIM_ERR_DEVICE, "pack launch failed: %s", hipGetErrorString(e));
    account(s, QSIM_K_PACK, 2.0 * (double)s->amp_bytes() * (double)(1ULL << s->n));
    return QSIM_OK;
}

extern "C" int qsim_pack_bits(qsim_state *s, const int *bits, int nbits, void *dst) { return pack_common(s, bits, nbits, dst, nullptr, false, 0); }
extern "C" int qsim_pack_bits_to(qsim_state *s, const int *bits, int nbits, void *const *dst_blocks) {
    return pack_common(s, bits, nbits, nullptr, dst_blocks, false, 0);
}
extern "C" int qsim_pack_bits_sparse(qsim_state *s, const int *bits, int nbits, void *dst, void *const *dst_blocks, uint32_t skip_blocks) {
    return pack_common(s, bits, nbits, dst_blocks ? nullptr : dst, dst_blocks, true, skip_blocks);
}

// qsim_flush + the re-layout of qsim_pack_bits_sparse in one call, so that the LAST tile pass of the queue can do the
// re-layout with its own stores (PackJob): no separate sweep over the state.  The output is one buffer in which source bit
// bits[j] lands on index bit to_bits[j] (NULL: n - nbits + j, the block index on top of a shard-sized buffer), the other
// bits close ranks below, and konst is ORed in (a cluster that keeps all its shards' buffers in one allocation addresses
// "block b of member j" that way).  Afterwards the state's own buffer holds stale data: the caller hands it its new contents
// (an exchange's receives, qsim_swap_buffer) and says what they are (qsim_set_support / qsim_reset_shard).
extern "C" int qsim_flush_pack(qsim_state *s, const int *bits, int nbits, const int *to_bits, uint64_t konst, void *out, uint64_t needed, uint32_t skip_blocks,
                               void **packed_at, int *fused) {
    if (!s || !bits) return fail(QSIM_ERR_ARG, "NULL argument");
    if (nbits < 1 || nbits > 8 || nbits > s->n) return fail(QSIM_ERR_ARG, "flush_pack: %d bits unsupported", nbits);
    for (int j = 0; j < nbits; j++)
        if (bits[j] < 0 || bits[j] >= s->n || (j && bits[j] <= bits[j - 1])) return fail(QSIM_ERR_ARG, "flush_pack: bit positions must be ascending and inside the shard");
    if (nbits > 3 || s->f32) {
        // What a tile pass cannot re-lay-out (PackMap carries three selected bits, fp64): exchanges of 4..8 qubits — groups of 16
        // and more shards — and fp32 states take the plain route, flush then the pack kernel, with the same sparse roles (blocks
        // nobody reads are left out while the mask has a bit for each: k <= 5).  Only the one-buffer layout exists there.
        if (to_bits || konst) return fail(QSIM_ERR_ARG, "flush_pack: %d bits%s only into one buffer (no to_bits / konst)", nbits, s->f32 ? " of an fp32 state" : "");
        void *dst = out ? out : s->spare;
        if (!dst) return fail(QSIM_ERR_ARG, "flush_pack: no output buffer (lend one with qsim_set_spare_buffer)");
        const int rc = pack_common(s, bits, nbits, dst, nullptr, true, nbits <= 5 ? skip_blocks : 0);
        if (rc) return rc;
        if (packed_at) *packed_at = dst;
        if (fused) *fused = 0;
        return QSIM_OK;
    }
    PackJob job;
    job.out = out;
    job.skip = skip_blocks;
    job.needed = needed;
    job.map.k = nbits;
    job.map.konst = konst;
    for (int j = 0; j < nbits; j++) {
        job.bits[j] = job.map.sel[j] = bits[j];
        job.map.to[j] = to_bits ? to_bits[j] : s->n - nbits + j;
        if (job.map.to[j] < s->n - nbits || job.map.to[j] > 62) return fail(QSIM_ERR_ARG, "flush_pack: destination bit %d collides with the bits that stay", job.map.to[j]);
    }
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    for (int i = 0; i <= nbits; i++) { // keep bits with i selected bits below them
        const uint64_t lo = i == 0 ? 0 : ((2ULL << bits[i - 1]) - 1ULL), hi = i == nbits ? nmask : ((1ULL << bits[i]) - 1ULL);
        job.map.seg[i] = hi & ~lo & nmask;
    }
    for (int i = nbits + 1; i < 4; i++) job.map.seg[i] = 0;
    if (!out && !s->spare) return fail(QSIM_ERR_ARG, "flush_pack: no output buffer (lend one with qsim_set_spare_buffer)");
    int rc = flush_impl(s, &job);
    if (rc) return rc;
    if (job.packed_at) {
        if (packed_at) *packed_at = job.packed_at;
        if (fused) *fused = 1;
        return QSIM_OK;
    }
    char *dst = (char *)(out ? out : s->spare);
    void *blocks[8];
    for (int b = 0; b < (1 << nbits); b++) {
        uint64_t at = konst;
        for (int j = 0; j < nbits; j++) at |= (uint64_t)((b >> j) & 1) << job.map.to[j];
        blocks[b] = dst + 16 * at;
    }
    if (getenv("QSIM_TRACE_PACK")) fprintf(stderr, "qsim: re-layout by the pack kernel (n = %d, support %llx)\n", s->n, (unsigned long long)current_support(s));
    rc = pack_common(s, bits, nbits, nullptr, blocks, true, skip_blocks);
    if (rc) return rc;
    if (packed_at) *packed_at = dst;
    if (fused) *fused = 0;
    return QSIM_OK;
}

// Hands the state a different amplitude buffer and returns the old one: the second half of an exchange whose pack kernels
// wrote every shard's NEW contents into the group members' spare buffers.  Both buffers hold 2^n amplitudes on the
// state's device; whoever holds a buffer when it is destroyed frees it, so ownership simply travels with the pointers.
extern "C" int qsim_swap_buffer(qsim_state *s, void **buffer) {
    if (!s || !buffer || !*buffer) return fail(QSIM_ERR_ARG, "NULL argument");
    const int rc = qsim_flush(s);
    if (rc) return rc;
    void *old = s->amps;
    s->amps = *buffer;
    *buffer = old;
    s->zero_ket_pending = false; // the new buffer's contents ARE the state: taken as written everywhere unless the caller says
    s->partial = false;          // otherwise (qsim_set_support, qsim_reset_shard)
    if (s->spare == s->amps && !s->owns_spare) s->spare = old; // a lent spare that just became the state: the old state takes its place
    return QSIM_OK;
}

// Lends the state a second buffer of 2^n amplitudes on its device for out-of-place tile passes (QSIM_OPT_PINGPONG); the
// caller keeps ownership and may use the buffer itself whenever no gates are pending (after qsim_flush / qsim_sync the
// state is in its own buffer and the lent one holds garbage).  NULL takes it back.
extern "C" int qsim_set_spare_buffer(qsim_state *s, void *buffer) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    const int rc = qsim_flush(s);
    if (rc) return rc;
    if (buffer == s->amps) return fail(QSIM_ERR_ARG, "set_spare_buffer: that is the state's own buffer");
    if (s->owns_spare && s->spare) {
        HIP_TRY(hipSetDevice(s->device));
        HIP_TRY(hipStreamSynchronize(s->stream));
        (void)hipFree(s->spare);
    }
    s->spare = buffer;
    s->owns_spare = false;
    s->spare_failed = false;
    return QSIM_OK;
}

extern "C" int qsim_scale(qsim_state *s, double re, double im) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    if (s->n < 1) return fail(QSIM_ERR_ARG, "scale needs at least one local qubit");
    const double U[8] = {re, im, 0, 0, 0, 0, re, im}; // diag(z, z) on local qubit 0: folds into the next fused block
    return qsim_apply_1q(s, U, 0);
}

extern "C" int qsim_get_stats(qsim_state *s, qsim_stats *out) {
    if (!s || !out) return fail(QSIM_ERR_ARG, "NULL argument");
    const int rc = resolve_events(s);
    if (rc) return rc;
    *out = s->stats;
    return QSIM_OK;
}

extern "C" int qsim_reset_stats(qsim_state *s) {
    if (!s) return fail(QSIM_ERR_ARG, "NULL state");
    const int rc = resolve_events(s);
    if (rc) return rc;
    memset(&s->stats, 0, sizeof s->stats);
    s->launch_log.clear();
    return QSIM_OK;
}

extern "C" int qsim_launch_log_order(qsim_state *s, long index, int *order, int *count) {
    if (!s || !order || !count) return QSIM_ERR_ARG;
    if (resolve_events(s)) return QSIM_ERR_DEVICE;
    if (index < 0 || index >= (long)s->launch_log.size()) return QSIM_ERR_ARG;
    const LaunchRec &r = s->launch_log[index];
    const int n = __builtin_popcountll(r.high_mask);
    for (int j = 0; j < n; j++) order[j] = (int)((r.order_code >> (5 * j)) & 31u);
    *count = r.kclass == QSIM_K_TILE ? n : 0;
    return QSIM_OK;
}

extern "C" int qsim_launch_log_visited(qsim_state *s, long index, double *visited) {
    if (!s || !visited) return QSIM_ERR_ARG;
    if (resolve_events(s)) return QSIM_ERR_DEVICE;
    if (index < 0 || index >= (long)s->launch_log.size()) return QSIM_ERR_ARG;
    *visited = s->launch_log[index].visited;
    return QSIM_OK;
}

extern "C" int qsim_launch_log_blocks(qsim_state *s, long index, uint8_t *codes, int cap, int *count) {
    if (!s || !count) return QSIM_ERR_ARG;
    if (resolve_events(s)) return QSIM_ERR_DEVICE;
    if (index < 0 || index >= (long)s->launch_log.size()) return QSIM_ERR_ARG;
    const std::vector<uint8_t> &f = s->launch_log[index].forms;
    *count = (int)f.size();
    for (int j = 0; codes && j < cap && j < (int)f.size(); j++) codes[j] = f[j];
    return QSIM_OK;
}

extern "C" long qsim_launch_log(qsim_state *s, long index, int *kclass, int *n_ops, uint64_t *high_mask, double *ms) {
    if (!s) return -1;
    if (resolve_events(s)) return -1;
    const long count = (long)s->launch_log.size();
    if (index >= 0 && index < count) {
        const LaunchRec &r = s->launch_log[index];
        if (kclass) *kclass = r.kclass;
        if (n_ops) *n_ops = r.n_ops;
        if (high_mask) *high_mask = r.high_mask;
        if (ms) *ms = r.ms;
    }
    return count;
}

// ---- circuits ------------------------------------------------------------------------------------------
extern "C" int qsim_run_circuit(qsim_state *s, const qsim_circuit *c, long first, long count) {
    if (!s || !c) return fail(QSIM_ERR_ARG, "NULL argument");
    if (c->num_q != s->n) return fail(QSIM_ERR_ARG, "circuit has %d qubits, state has %d", c->num_q, s->n);
    if (first < 0 || first > c->count) return fail(QSIM_ERR_ARG, "first gate %ld outside the circuit", first);
    long end = count < 0 ? c->count : first + count;
    if (end > c->count) end = c->count;
    for (long i = first; i < end; i++) {
        const qsim_gate_rec &g = c->gates[i];
        int rc;
        if (g.kind == QSIM_GATE_U1) rc = qsim_apply_1q(s, c->mats2 + 8 * (long)g.mat, g.q0);
        else if (g.kind == QSIM_GATE_CX) rc = qsim_apply_cx(s, g.q0, g.q1);
        else rc = qsim_apply_2q(s, c->mats4 + 32 * (long)g.mat, g.q0, g.q1);
        if (rc) return rc;
    }
    return QSIM_OK;
}

static void feed(Scheduler &sched, const qsim_circuit *c) {
    for (long i = 0; i < c->count; i++) {
        const qsim_gate_rec &g = c->gates[i];
        if (g.kind == QSIM_GATE_U1) {
            cd m[4];
            for (int k = 0; k < 4; k++) m[k] = cd(c->mats2[8 * (long)g.mat + 2 * k], c->mats2[8 * (long)g.mat + 2 * k + 1]);
            sched.add_1q(m, g.q0);
        } else if (g.kind == QSIM_GATE_CX) {
            sched.add_cx(g.q0, g.q1);
        } else {
            cd m[16];
            for (int k = 0; k < 16; k++) m[k] = cd(c->mats4[32 * (long)g.mat + 2 * k], c->mats4[32 * (long)g.mat + 2 * k + 1]);
            sched.add_2q(m, g.q0, g.q1);
        }
    }
}

static double pass_cost(const Pass &p, bool f32) { return pass_time_cost(p, f32); } // scheduler.h

// Schedules the circuit under a few dozen scheduler settings, remembers the one whose passes are predicted to take the least
// time (pass_cost) under the key qsim_flush will compute for the same gates on a state with this support, and hands its
// passes back.
struct RankedVariant { SchedHint hint; double cost; bool is_default; };
static SchedConfig with_hint(SchedConfig v, const SchedHint &h) {
    v.commute = h.commute; v.cheap_margin = h.cheap_margin; v.lookahead = h.lookahead; v.seed = h.seed;
    if (h.cap > 0) { v.tile_max_ops = h.cap; v.tail_max_ops = std::max(v.tail_max_ops, h.cap); }
    return v;
}
// circuits whose schedule was chosen by MEASUREMENT (qsim_tune_circuit): the choice stands until the table is cleared — timing
// the same candidates again could flip between near-equal schedules and invalidate the geometries measured for the winner
static std::map<uint64_t, RankedVariant> g_sched_measured; // guarded by g_hints_mu
static void set_sched_hint(uint64_t key, const SchedHint &now, bool is_default, const SchedConfig &scfg) {
    std::lock_guard<std::mutex> lock(g_hints_mu);
    const auto it = g_sched_hints.find(key);
    const SchedHint dflt{scfg.commute, scfg.cheap_margin, scfg.lookahead, 0, 0};
    const SchedHint before = it == g_sched_hints.end() ? dflt : it->second;
    if (is_default) g_sched_hints.erase(key);
    else {
        if (g_sched_hints.size() >= kMaxSchedHints && it == g_sched_hints.end()) { // full: both tables start over together — a measured
            g_sched_hints.clear();                                                // entry without its hint would pin a schedule nobody runs
            const auto mine = g_sched_measured.find(key);
            const bool keep = mine != g_sched_measured.end();
            const RankedVariant kept = keep ? mine->second : RankedVariant{};
            g_sched_measured.clear();
            if (keep) g_sched_measured[key] = kept;
        }
        g_sched_hints[key] = now;
    }
    if (before.commute != now.commute || before.cheap_margin != now.cheap_margin || before.lookahead != now.lookahead || before.cap != now.cap || before.seed != now.seed)
        g_wisdom_epoch++; // cached plans of this circuit were scheduled another way
}

// The circuit as the gate queue qsim_run_circuit would leave in a state (what the plan and schedule-hint keys are computed from).
static std::vector<QueuedGate> queue_of(const qsim_circuit *c) {
    std::vector<QueuedGate> q((size_t)c->count);
    for (long i = 0; i < c->count; i++) {
        const qsim_gate_rec &g = c->gates[i];
        QueuedGate &o = q[(size_t)i];
        o.kind = g.kind; o.q0 = g.q0; o.q1 = g.kind == QSIM_GATE_U1 ? -1 : g.q1;
        const double *U = g.kind == QSIM_GATE_U1 ? c->mats2 + 8 * (long)g.mat : g.kind == QSIM_GATE_CX ? nullptr : c->mats4 + 32 * (long)g.mat;
        for (int k = 0; U && k < (g.kind == QSIM_GATE_U1 ? 4 : 16); k++) o.m[k] = cd(U[2 * k], U[2 * k + 1]);
    }
    return q;
}

static void choose_schedule(qsim_state *s, const qsim_circuit *c, const SchedConfig &scfg, std::vector<Pass> *out,
                            std::vector<RankedVariant> *ranked = nullptr, uint64_t *key_out = nullptr, const std::atomic<bool> *stop = nullptr) {
    std::vector<Pass> passes;
    if (s->fuse < 3) {
        Scheduler sv(scfg);
        feed(sv, c);
        sv.finish(passes);
        if (out) *out = std::move(passes);
        return;
    }
    const std::vector<QueuedGate> q = queue_of(c);
    const uint64_t key = gates_key(s, plan_identity(s, q.size(), scfg.initial_support), q.data(), q.size());
    if (key_out) *key_out = key;
    {
        // A circuit whose schedule was chosen by measurement keeps it: no candidate is scheduled again (80 schedules cost about a
        // second at n = 30), and the hint is put back in case the hint table was emptied in between (kMaxSchedHints) — without
        // it the circuit would silently run its default schedule with the geometries measured for another one.
        bool measured = false;
        RankedVariant kept{};
        {
            std::lock_guard<std::mutex> lock(g_hints_mu);
            auto it = g_sched_measured.find(key);
            if (it != g_sched_measured.end()) { measured = true; kept = it->second; }
        }
        if (measured) {
            set_sched_hint(key, kept.hint, kept.is_default, scfg);
            Scheduler sv(with_hint(scfg, kept.hint));
            feed(sv, c);
            sv.finish(passes);
            if (ranked) ranked->clear(); // nothing left to try
            if (out) *out = std::move(passes);
            return;
        }
    }
    // the variants: how many clusters a pass may take (where the engine sets a cap of its own: states of 4 GiB and more),
    // clusters may / may not overtake (commute), how eagerly passes inside the support are kept (cheap_margin), one more
    // pass of lookahead where the local search is on; the default comes first and wins ties
    std::vector<SchedHint> variants;
    std::vector<int> caps{0};
    if (scfg.tail_max_ops > scfg.tile_max_ops) // the engine's own cap is in force (engine_sched_config), not a caller's
        for (int cap : {24, 28, 32, 40})
            if (cap != scfg.tile_max_ops) caps.push_back(cap);
    // (commuting clusters first: over 16 seeded 1000-gate circuits at n = 30 a schedule without them never came within 15 % of the best
    // of these candidates, so a search that is cut short — `stop` — spends its time on the half that wins)
    for (int com = 1; com >= 0; com--)
        for (int cap : caps)
            for (double mar : {scfg.cheap_margin, 2.0 * scfg.cheap_margin})
                for (int la = scfg.lookahead; la <= scfg.lookahead + (scfg.lookahead >= 1 ? 1 : 0); la++) variants.push_back({com, mar, la, cap, 0});
    // Every candidate is an independent run of the scheduler on the same gates: they are evaluated on up to 16 host threads
    // (80 schedules at n = 30: 0.9-1.3 s on one thread) and REDUCED in candidate order with the same rule as before — the default
    // first, a later one only when it is at least 0.5 % cheaper — so the choice does not depend on the thread count.  `stop`
    // (the cold path of the C host: "plan while the state is being allocated, no longer") ends the search early: candidates not
    // evaluated by then simply do not take part; the default always does.
    std::vector<double> costs; // per candidate; < 0: not evaluated
    auto evaluate = [&](size_t first, size_t last) {
        costs.resize(last, -1.0);
        std::atomic<size_t> next{first};
        auto worker = [&]() {
            for (;;) {
                const size_t vi = next.fetch_add(1);
                if (vi >= last) return;
                if (vi != 0 && stop && stop->load()) return;
                Scheduler sv(with_hint(scfg, variants[vi]));
                feed(sv, c);
                double cost = 0;
                sv.finish([&](Pass &&p) { cost += pass_cost(p, s->f32); });
                costs[vi] = cost;
            }
        };
        const unsigned hw = std::thread::hardware_concurrency();
        const size_t nthreads = std::min<size_t>({(size_t)16, (size_t)(hw ? hw : 1), last - first}); // a GPU's share of its host's cores
        std::vector<std::thread> pool;
        for (size_t t = 1; t < nthreads; t++) pool.emplace_back(worker);
        worker();
        for (std::thread &t : pool) t.join();
    };
    double best_cost = 0;
    size_t best = 0;
    auto reduce = [&](size_t first, size_t last) {
        for (size_t vi = first; vi < last; vi++) {
            if (costs[vi] < 0) continue;
            if (ranked) ranked->push_back({variants[vi], costs[vi], vi == 0});
            if (vi == 0 || costs[vi] < best_cost * 0.995) { best_cost = costs[vi]; best = vi; }
        }
    };
    evaluate(0, variants.size());
    reduce(0, variants.size());
    // ... and, for the three settings that came out best, the same setting with its ties broken differently (SchedConfig::seed):
    // the greedy packing is sensitive to which of several equally good clusters or qubits it takes first — over 40 seeds the
    // swept bytes of one setting spread by 10 % and more (bench circuit 9.57 -> 8.63 sweeps, another 9.13 -> 8.06)
    if ((variants.size() > 1 || scfg.local_iters > 0) && !(stop && stop->load())) {
        std::vector<size_t> order;
        for (size_t i = 0; i < variants.size(); i++)
            if (costs[i] >= 0) order.push_back(i);
        std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return costs[a] < costs[b]; });
        const size_t base_count = std::min<size_t>(3, order.size()), first_seeded = variants.size();
        constexpr int kSeeds = 16;
        for (size_t b = 0; b < base_count; b++)
            for (int sd = 1; sd <= kSeeds; sd++) {
                SchedHint h = variants[order[b]];
                h.seed = (uint64_t)sd;
                variants.push_back(h);
            }
        evaluate(first_seeded, variants.size());
        reduce(first_seeded, variants.size());
    }
    set_sched_hint(key, variants[best], best == 0, scfg);
    if (out) { // the passes of the choice (one more run of the scheduler: the candidates kept their costs only)
        Scheduler sv(with_hint(scfg, variants[best]));
        feed(sv, c);
        sv.finish(passes);
        *out = std::move(passes);
    }
}

// The schedule choice alone (no timing): for a run from a reset and for a run on a dense state.
extern "C" int qsim_choose_schedule(qsim_state *s, const qsim_circuit *c) {
    if (!s || !c) return fail(QSIM_ERR_ARG, "NULL argument");
    if (c->num_q != s->n) return fail(QSIM_ERR_ARG, "circuit has %d qubits, state has %d", c->num_q, s->n);
    const int rc = qsim_flush(s);
    if (rc) return rc;
    for (int dense = 0; dense < 2; dense++) {
        if (!dense && !s->sparse_start) continue;
        const SchedConfig scfg = sched_config(s->n, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, s->f32, dense ? ~0ULL : 0);
        choose_schedule(s, c, scfg, nullptr);
    }
    return QSIM_OK;
}

// The cold path of the C host (bin/qsim: one circuit, one run, quantum_simulator.c:143-248): the schedule choice for a run from a
// reset, for as long as the state's buffer is still being allocated (qsim_create_async) and no longer — the candidates evaluated
// by then compete, the default always does.  With a buffer that is already there it returns at once with the default schedule.
extern "C" int qsim_choose_schedule_while_allocating(qsim_state *s, const qsim_circuit *c) {
    if (!s || !c) return fail(QSIM_ERR_ARG, "NULL argument");
    if (c->num_q != s->n) return fail(QSIM_ERR_ARG, "circuit has %d qubits, state has %d", c->num_q, s->n);
    if (s->alloc_done.load() || s->fuse < 3) return QSIM_OK;
    const SchedConfig scfg = sched_config(s->n, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, s->f32, s->sparse_start ? 0 : ~0ULL);
    { // the tile kernel's code object is loaded at its first use: here, beside the allocation, instead of in front of the first pass
        HIP_TRY(hipSetDevice(s->device));
        TileGeom g{};
        g.tile_bits = std::min(scfg.tile_bits, s->n);
        g.low_bits = std::min(scfg.tile_low_bits, g.tile_bits);
        g.n_high = g.tile_bits - g.low_bits;
        g.n = s->n;
        LaunchCfg cfg{s->stream, s->grid_cap, true};
        (void)launch_tile(cfg, nullptr, nullptr, s->f32, g, nullptr, 0, s->tile_threads, false, 1.0);
        (void)hipGetLastError();
    }
    choose_schedule(s, c, scfg, nullptr, nullptr, nullptr, &s->alloc_done);
    return QSIM_OK;
}

// The same choice for a run that finds the state with exactly this support (what qsim_flush will key its lookup with: all ones
// for a dense state, 0 fresh from a reset, the mask of qsim_set_support after a sparse exchange).
extern "C" int qsim_choose_schedule_for(qsim_state *s, const qsim_circuit *c, uint64_t support) {
    if (!s || !c) return fail(QSIM_ERR_ARG, "NULL argument");
    if (c->num_q != s->n) return fail(QSIM_ERR_ARG, "circuit has %d qubits, state has %d", c->num_q, s->n);
    const int rc = qsim_flush(s);
    if (rc) return rc;
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    if (!s->sparse_start || (support & nmask) == nmask) support = ~0ULL;
    else support &= nmask;
    const SchedConfig scfg = sched_config(s->n, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, s->f32, support);
    choose_schedule(s, c, scfg, nullptr);
    return QSIM_OK;
}

// Where a state that has this support can be non-zero after the circuit, as THIS engine will know it then: the circuit is
// scheduled the way qsim_flush will schedule the same gates (same options, same remembered schedule choice) and every tile pass
// adds its tile qubits; a single-gate kernel makes the state dense.  A cluster's planner derives from it what an exchange's
// receivers look at, so that the sender's last tile pass — which writes exactly this — can do the re-layout itself.
extern "C" int qsim_support_after(qsim_state *s, const qsim_circuit *c, uint64_t support, uint64_t *after) {
    if (!s || !c || !after) return fail(QSIM_ERR_ARG, "NULL argument");
    if (c->num_q != s->n) return fail(QSIM_ERR_ARG, "circuit has %d qubits, state has %d", c->num_q, s->n);
    const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
    if (!s->sparse_start || (support & nmask) == nmask) { *after = nmask; return QSIM_OK; }
    support &= nmask;
    if (c->count == 0) { *after = support; return QSIM_OK; }
    SchedConfig scfg = sched_config(s->n, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, s->f32, support);
    if (s->fuse >= 3 && have_sched_hints()) {
        const std::vector<QueuedGate> q = queue_of(c);
        apply_sched_hint(gates_key(s, plan_identity(s, q.size(), support), q.data(), q.size()), scfg);
    }
    Scheduler sv(scfg);
    feed(sv, c);
    uint64_t sup = support;
    sv.finish([&](Pass &&p) {
        if (p.kclass != QSIM_K_TILE) { sup = nmask; return; } // the engine writes the zeros out first (materialize_zero_ket)
        sup |= (1ULL << p.geom.low_bits) - 1ULL;
        for (int j = 0; j < p.geom.n_high; j++) sup |= 1ULL << p.geom.high[j];
    });
    *after = sup & nmask;
    return QSIM_OK;
}

// ---- measured pass geometry -------------------------------------------------------------------------------------------
// Plans the circuit exactly as qsim_run_circuit + qsim_flush would and, for every tile pass whose geometry is not in the
// table yet, times the pass (its real blocks, on whatever the state buffer holds) under candidate orders of its high
// tile bits: ascending first, then pseudo-random permutations seeded by the bit set, until max_candidates have been
// tried or the pass's share of budget_ms is spent (at least four).  The fastest order goes into the process-wide table
// order_tile_bits() consults.  The state's contents are clobbered, so it is left reset to |0...0>.  Results never
// depend on the order; only the pass times do.
extern "C" int qsim_tune_circuit(qsim_state *s, const qsim_circuit *c, int max_candidates, double budget_ms, qsim_tune_report *rep) {
    return qsim_tune_circuit_from(s, c, max_candidates, budget_ms, rep, 0);
}

extern "C" int qsim_tune_circuit_support(qsim_state *s, const qsim_circuit *c, int max_candidates, double budget_ms, qsim_tune_report *rep, uint64_t support);
extern "C" int qsim_tune_circuit_from(qsim_state *s, const qsim_circuit *c, int max_candidates, double budget_ms, qsim_tune_report *rep,
                                      int dense_start) {
    return qsim_tune_circuit_support(s, c, max_candidates, budget_ms, rep, dense_start ? ~0ULL : 0);
}

extern "C" int qsim_tune_circuit_support(qsim_state *s, const qsim_circuit *c, int max_candidates, double budget_ms, qsim_tune_report *rep, uint64_t support) {
    if (!s || !c) return fail(QSIM_ERR_ARG, "NULL argument");
    if (c->num_q != s->n) return fail(QSIM_ERR_ARG, "circuit has %d qubits, state has %d", c->num_q, s->n);
    if (max_candidates < 1) max_candidates = 1;
    int rc = qsim_sync(s);
    if (rc) return rc;
    HIP_TRY(hipSetDevice(s->device));
    {
        const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
        if (!s->sparse_start || (support & nmask) == nmask) support = ~0ULL;
        else support &= nmask;
    }
    SchedConfig scfg = sched_config(s->n, s->fuse, s->tile_bits, s->tile_low_bits, s->tile_max_ops, s->tile_pad_from, s->f32,
                                    support); // 0: the run that follows starts from the reset this call ends with
    // Which way to schedule THIS circuit is decided first; its passes are the ones measured below.  The pass-time model ranks
    // the scheduler settings (choose_schedule); with timing allowed (max_candidates > 1) the four schedules it likes best
    // are then RUN once each on the state — the model is good to ~0.4 ms per pass, i.e. it cannot tell schedules apart
    // that differ by less than ~2 % — and the fastest is kept.
    std::vector<Pass> passes;
    std::vector<RankedVariant> ranked;
    uint64_t sched_key = 0;
    choose_schedule(s, c, scfg, &passes, &ranked, &sched_key);
    qsim_tune_report r{};
    if (max_candidates > 1 && s->fuse >= 3 && ranked.size() > 1) {
        std::stable_sort(ranked.begin(), ranked.end(), [](const RankedVariant &a, const RankedVariant &b) { return a.cost < b.cost; });
        std::vector<RankedVariant> tries;
        if (const char *v = getenv("QSIM_TUNE_SCHEDULES")) s->tune_schedules = std::max(1, std::min(80, atoi(v)));
        for (const RankedVariant &v : ranked) { // distinct predicted costs = (almost surely) distinct schedules
            bool dup = false;
            for (const RankedVariant &t : tries) dup = dup || t.cost == v.cost;
            if (!dup) tries.push_back(v);
            if (tries.size() == (size_t)s->tune_schedules) break;
        }
        hipEvent_t t0 = nullptr, t1 = nullptr;
        HIP_TRY(hipEventCreate(&t0));
        HIP_TRY(hipEventCreate(&t1));
        const int saved_profile = s->profile;
        s->profile = 0;
        const uint64_t nmask = s->n >= 64 ? ~0ULL : ((1ULL << s->n) - 1ULL);
        float best_ms = 0.f;
        size_t best_i = 0;
        for (size_t i = 0; i < tries.size() && rc == QSIM_OK; i++) {
            set_sched_hint(sched_key, tries[i].hint, tries[i].is_default, scfg);
            float ms = 0.f;
            for (int rep2 = 0; rep2 < 2 && rc == QSIM_OK; rep2++) { // the second run replays the cached plan: no host work in the way
                if (support == 0) rc = qsim_reset(s);
                else if (support != ~0ULL) rc = qsim_set_support(s, support & nmask);
                else { s->zero_ket_pending = false; s->partial = false; } // a dense state: whatever the buffer holds
                if (rc) break;
                (void)hipEventRecord(t0, s->stream);
                rc = qsim_run_circuit(s, c, 0, -1);
                if (rc == QSIM_OK) rc = qsim_flush(s);
                (void)hipEventRecord(t1, s->stream);
                if (rc == QSIM_OK && hipEventSynchronize(t1) != hipSuccess) rc = fail(QSIM_ERR_DEVICE, "planning: event sync failed");
                if (rc == QSIM_OK && hipEventElapsedTime(&ms, t0, t1) != hipSuccess) rc = fail(QSIM_ERR_DEVICE, "planning: event time failed");
            }
            if (i == 0 || ms < best_ms) { best_ms = ms; best_i = i; }
        }
        s->profile = saved_profile;
        s->stats.gates -= std::min<uint64_t>(s->stats.gates, (uint64_t)c->count * 2 * tries.size()); // planning runs are not gate statements of the caller
        (void)hipEventDestroy(t0);
        (void)hipEventDestroy(t1);
        if (rc) return rc;
        set_sched_hint(sched_key, tries[best_i].hint, tries[best_i].is_default, scfg);
        {
            std::lock_guard<std::mutex> lock(g_hints_mu);
            if (g_sched_measured.size() >= kMaxSchedHints) g_sched_measured.clear();
            g_sched_measured[sched_key] = tries[best_i];
        }
        { // the passes of the schedule that won: the ones whose tile-bit orders are measured below
            Scheduler sv(with_hint(scfg, tries[best_i].hint));
            feed(sv, c);
            passes.clear();
            sv.finish(passes);
        }
        rc = qsim_sync(s);
        if (rc) return rc;
    }
    std::vector<const Pass *> todo;
    for (const Pass &p : passes) {
        if (p.kclass != QSIM_K_TILE) continue;
        r.tile_passes++;
        if (p.geom.n_high < 2) continue;
        std::lock_guard<std::mutex> lock(g_wisdom_mu);
        if (g_wisdom.count(geom_key(s, p.geom))) { r.already_known++; continue; }
        bool dup = false;
        for (const Pass *q : todo) dup = dup || (geom_key(s, q->geom).high_mask == geom_key(s, p.geom).high_mask);
        if (!dup) todo.push_back(&p);
    }
    hipEvent_t e0 = nullptr, e1 = nullptr;
    HIP_TRY(hipEventCreate(&e0));
    HIP_TRY(hipEventCreate(&e1));
    const auto t_begin = std::chrono::steady_clock::now();
    auto elapsed_ms = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_begin).count(); };
    const int saved_skip_mem = s->debug_skip_mem;
    s->debug_skip_mem = 0;
    void *const home = s->amps;
    const bool tune_oop = !todo.empty() && spare_buffer(s) != nullptr;
    auto timed = [&](const Pass &p, const TileGeom &g, float &ms) -> int {
        (void)hipEventRecord(e0, s->stream);
        const int rc2 = launch_tile_pass(s, p, g, false, nullptr, tune_oop); // timed the way most passes of a run go
        if (rc2) return rc2;
        (void)hipEventRecord(e1, s->stream);
        if (hipEventSynchronize(e1) != hipSuccess) return fail(QSIM_ERR_DEVICE, "tuning: event sync failed");
        if (hipEventElapsedTime(&ms, e0, e1) != hipSuccess) return fail(QSIM_ERR_DEVICE, "tuning: event time failed");
        return QSIM_OK;
    };
    rc = QSIM_OK;
    for (size_t i = 0; i < todo.size() && rc == QSIM_OK; i++) {
        const Pass &p = *todo[i];
        const double share_end = budget_ms > 0 ? budget_ms * (double)(i + 1) / (double)todo.size() : 1e300;
        TileGeom asc = p.geom;
        std::sort(asc.high, asc.high + asc.n_high);
        float ms = 0.f;
        rc = timed(p, asc, ms); // warm: first touch of the op buffer and of this geometry's code path
        if (rc == QSIM_OK) rc = timed(p, asc, ms);
        if (rc) break;
        GeomOrder best{};
        for (int j = 0; j < asc.n_high; j++) best.high[j] = (int8_t)asc.high[j];
        best.ms = best.ms_ascending = ms;
        const GeomKey key = geom_key(s, asc);
        for (int cand = 1; cand < max_candidates; cand++) {
            if (cand >= 4 && elapsed_ms() > share_end) break;
            TileGeom g = asc;
            shuffle_high(g, key.high_mask * 0x9E3779B97F4A7C15ULL + (uint64_t)cand * 0xD1B54A32D192ED03ULL);
            rc = timed(p, g, ms);
            if (rc) break;
            r.candidates_timed++;
            if (ms < best.ms) {
                best.ms = ms;
                for (int j = 0; j < g.n_high; j++) best.high[j] = (int8_t)g.high[j];
            }
        }
        if (rc) break;
        if (best.ms < best.ms_ascending * 0.985f) { // keep ascending unless the gain is beyond the timing noise
            r.passes_reordered++;
        } else {
            for (int j = 0; j < asc.n_high; j++) best.high[j] = (int8_t)asc.high[j];
            best.ms = best.ms_ascending;
        }
        r.ms_ascending += best.ms_ascending;
        r.ms_best += best.ms;
        r.passes_tuned++;
        std::lock_guard<std::mutex> lock(g_wisdom_mu);
        g_wisdom[key] = best;
        g_wisdom_epoch++;
    }
    s->debug_skip_mem = saved_skip_mem;
    (void)hipStreamSynchronize(s->stream);
    if (s->amps != home) std::swap(s->amps, s->spare); // contents are scratch here (reset below); the buffers keep their roles
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    r.seconds = elapsed_ms() * 1e-3;
    if (rep) *rep = r;
    const int rc_reset = qsim_reset(s);
    return rc ? rc : rc_reset;
}

extern "C" long qsim_tune_table_size(void) {
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    return (long)g_wisdom.size();
}

// The table as text, one geometry per line: n f32 tile_bits low_bits high_mask(hex) ms ms_ascending order...  A table
// measured once (per machine) can be loaded by later processes: the C host does so when QSIM_WISDOM names a file.
extern "C" int qsim_tune_table_save(const char *path) {
    if (!path) return fail(QSIM_ERR_ARG, "NULL path");
    FILE *f = fopen(path, "w");
    if (!f) return fail(QSIM_ERR_OPEN, "cannot write %s", path);
    {   // measured schedule choices: "sched <key> <commute> <cheap_margin> <lookahead> <cap> <is_default>"
        std::lock_guard<std::mutex> lock(g_hints_mu);
        for (const auto &kv : g_sched_measured)
            fprintf(f, "sched %llx %d %.17g %d %d %d %llu\n", (unsigned long long)kv.first, kv.second.hint.commute, kv.second.hint.cheap_margin,
                    kv.second.hint.lookahead, kv.second.hint.cap, kv.second.is_default ? 1 : 0, (unsigned long long)kv.second.hint.seed);
    }
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    for (const auto &kv : g_wisdom) {
        const int nh = __builtin_popcountll(kv.first.high_mask);
        fprintf(f, "%d %d %d %d %llx %.4f %.4f", kv.first.n, kv.first.f32, kv.first.tile_bits, kv.first.low_bits,
                (unsigned long long)kv.first.high_mask, kv.second.ms, kv.second.ms_ascending);
        for (int j = 0; j < nh; j++) fprintf(f, " %d", (int)kv.second.high[j]);
        fprintf(f, "\n");
    }
    fclose(f);
    return QSIM_OK;
}

extern "C" long qsim_tune_table_load(const char *path) {
    if (!path) return -1;
    FILE *f = fopen(path, "r");
    if (!f) return -1;
    long loaded = 0;
    char line[512];
    while (fgets(line, sizeof line, f)) {
        if (strncmp(line, "sched ", 6) == 0) {
            unsigned long long key = 0, seed = 0;
            RankedVariant rv{};
            int isd = 0;
            if (sscanf(line + 6, "%llx %d %lf %d %d %d %llu", &key, &rv.hint.commute, &rv.hint.cheap_margin, &rv.hint.lookahead, &rv.hint.cap, &isd, &seed) >= 6 &&
                rv.hint.cheap_margin > 0 && rv.hint.lookahead >= 0 && rv.hint.lookahead <= 8 && rv.hint.cap >= 0 && rv.hint.cap <= 512) {
                rv.is_default = isd != 0;
                rv.hint.seed = seed;
                std::lock_guard<std::mutex> lock(g_hints_mu);
                g_sched_measured[key] = rv;
                if (rv.is_default) g_sched_hints.erase(key); else g_sched_hints[key] = rv.hint;
                g_wisdom_epoch++;
                loaded++;
            }
            continue;
        }
        GeomKey k{};
        GeomOrder o{};
        unsigned long long hm = 0;
        int used = 0;
        if (sscanf(line, "%d %d %d %d %llx %f %f%n", &k.n, &k.f32, &k.tile_bits, &k.low_bits, &hm, &o.ms, &o.ms_ascending, &used) < 7) continue;
        k.high_mask = hm;
        const int nh = __builtin_popcountll(hm);
        if (nh < 2 || nh > kMaxTileHigh) continue;
        const char *p = line + used;
        uint64_t seen = 0;
        bool ok = true;
        for (int j = 0; j < nh && ok; j++) {
            int v = -1, adv = 0;
            if (sscanf(p, "%d%n", &v, &adv) < 1 || v < 0 || v > 62 || !((hm >> v) & 1ULL) || ((seen >> v) & 1ULL)) ok = false;
            else { o.high[j] = (int8_t)v; seen |= 1ULL << v; p += adv; }
        }
        if (!ok) continue; // not a permutation of the set: ignore the line
        std::lock_guard<std::mutex> lock(g_wisdom_mu);
        g_wisdom[k] = o;
        g_wisdom_epoch++;
        loaded++;
    }
    fclose(f);
    return loaded;
}

extern "C" void qsim_tune_table_clear(void) {
    {
        std::lock_guard<std::mutex> lock(g_hints_mu);
        g_sched_hints.clear();
        g_sched_measured.clear();
    }
    std::lock_guard<std::mutex> lock(g_wisdom_mu);
    g_wisdom.clear();
    g_wisdom_epoch++;
}

extern "C" int qsim_plan_circuit_from(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, uint64_t initial_support, qsim_stats *out);
extern "C" int qsim_plan_circuit(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, qsim_stats *out) {
    return qsim_plan_circuit_from(c, fuse, tile_bits, tile_low_bits, 0, out);
}

extern "C" int qsim_plan_circuit_from(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, uint64_t initial_support, qsim_stats *out) {
    if (!c || !out) return fail(QSIM_ERR_ARG, "NULL argument");
    if (fuse < 0 || fuse > 3) return fail(QSIM_ERR_ARG, "fuse level %d not in 0..3", fuse);
    Scheduler sched(sched_config(c->num_q, fuse, tile_bits, tile_low_bits, 32, 10, false, initial_support));
    feed(sched, c);
    std::vector<Pass> passes;
    sched.finish(passes);
    memset(out, 0, sizeof *out);
    out->gates = sched.gates_seen();
    for (const Pass &p : passes) { // a run from a reset: the first tile passes visit part of the register (Pass::visited)
        out->launches++;
        out->algorithmic_bytes += p.bytes * p.visited;
        out->k_launches[p.kclass]++;
        out->k_bytes[p.kclass] += p.bytes * p.visited;
    }
    return QSIM_OK;
}

// The same schedule pass by pass: which qubits each tile pass holds in its tile, how much of the register it visits and what the
// planning steps price it at.  What a host-side model needs to decide which passes could run chunk by chunk beside an exchange
// (bench.py exchange_model: a pass can only be pipelined over index bits that are NOT in its tile).
extern "C" int qsim_plan_passes(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, uint64_t initial_support, qsim_pass_info *out, int cap,
                                int *count) {
    if (!c || !count || (cap > 0 && !out)) return fail(QSIM_ERR_ARG, "NULL argument");
    if (fuse < 0 || fuse > 3) return fail(QSIM_ERR_ARG, "fuse level %d not in 0..3", fuse);
    Scheduler sched(sched_config(c->num_q, fuse, tile_bits, tile_low_bits, 32, 10, false, initial_support));
    feed(sched, c);
    std::vector<Pass> passes;
    sched.finish(passes);
    *count = (int)passes.size();
    for (size_t i = 0; i < passes.size() && (int)i < cap; i++) {
        const Pass &p = passes[i];
        qsim_pass_info &o = out[i];
        o.kernel_class = p.kclass;
        o.blocks = p.kclass == QSIM_K_TILE ? (int)p.blocks.size() - p.geom.n_scale : 1;
        o.tile_mask = 0;
        if (p.kclass == QSIM_K_TILE) {
            o.tile_mask = (1ULL << p.geom.low_bits) - 1ULL;
            for (int j = 0; j < p.geom.n_high; j++) o.tile_mask |= 1ULL << p.geom.high[j];
        } else {
            o.tile_mask = c->num_q >= 64 ? ~0ULL : ((1ULL << c->num_q) - 1ULL); // a single-gate kernel: treat every bit as touched
        }
        o.visited = p.visited;
        o.bytes = p.bytes * p.visited;
        o.cost_bytes = pass_cost(p, false);
    }
    return QSIM_OK;
}

extern "C" int qsim_schedule_circuit(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, int tile_max_ops,
                                     qsim_sched_cb cb, void *user) {
    if (!c || !cb) return fail(QSIM_ERR_ARG, "NULL argument");
    if (fuse < 0 || fuse > 3) return fail(QSIM_ERR_ARG, "fuse level %d not in 0..3", fuse);
    Scheduler sched(sched_config(c->num_q, fuse, tile_bits, tile_low_bits, tile_max_ops));
    feed(sched, c);
    std::vector<Pass> passes;
    sched.finish(passes);
    int pi = 0;
    std::vector<double> big((size_t)2 * 256 * 256);
    std::vector<cd> full((size_t)256 * 256);
    for (const Pass &p : passes) {
        for (const FusedOp &op : p.ops) {
            double U[128];
            const int d = op.kind == OP_CX ? 0 : op.dim();
            for (int k = 0; k < d * d; k++) { U[2 * k] = op.m[k].real(); U[2 * k + 1] = op.m[k].imag(); }
            const int kind = op.kind == OP_G1 ? QSIM_GATE_U1 : op.kind == OP_CX ? QSIM_GATE_CX : QSIM_GATE_U2;
            const int qs[2] = {op.q_hi, op.q_lo};
            cb(user, pi, p.kclass, kind, qs, op.kind == OP_CX ? 2 : op.nq(), d ? U : nullptr, (int)op.gates);
        }
        for (const TileBlock &blk : p.blocks) { // reported as ONE matrix on (selecting qubits..., tile qubits...)
            TileOp t;
            if (!to_tile_op(p.geom, blk, t)) return fail(QSIM_ERR_ARG, "internal: block does not fit its tile pass");
            const int nq = blk.ns + blk.nq, D = 1 << nq;
            blk.full_matrix(full.data());
            for (int k = 0; k < D * D; k++) { big[2 * k] = full[k].real(); big[2 * k + 1] = full[k].imag(); }
            int qs[kMaxBlockQ + 2], j = 0;
            for (int a = 0; a < blk.ns; a++) qs[j++] = blk.s[a];
            for (int a = 0; a < blk.nq; a++) qs[j++] = blk.q[a];
            static const int kinds[9] = {0, QSIM_GATE_U1, QSIM_GATE_U2, QSIM_GATE_U3, QSIM_GATE_U4, QSIM_GATE_U5, QSIM_GATE_U6, QSIM_GATE_U7, QSIM_GATE_U8};
            cb(user, pi, p.kclass, kinds[nq], qs, nq, big.data(), (int)blk.gates);
        }
        pi++;
    }
    return QSIM_OK;
}
