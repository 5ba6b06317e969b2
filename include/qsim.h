/*
 * qsim.h — C ABI of libqsim.so, the MI355X-native state-vector engine.
 *
 * Plain C: opaque handles, pointers and sizes only.  Every entry point names the reference interface
 * it stands in for (file:line into RiccardoFiorentini/GPU_quantum_simulator).  The reference has no
 * plugin/FFI layer: its drop-in surface is the CLI of quantum_simulator.c plus the three C functions
 * declared at quantum_simulator.c:25-27, which live in qsim_legacy.h.  This header is the handle API
 * those are built on — the state vector stays in HBM between gates.
 *
 * Conventions (quantum_simulator.c:83): qubit k is bit k of the amplitude index, q[0] = LSB.
 * Amplitudes are fp64 complex, array-of-structs (re, im), 16 bytes each, exactly the memory layout of
 * C99 `double _Complex` (quantum_simulator.c:35,168).  Matrices are row-major, (re, im) interleaved,
 * and mean the standard U·v (NOT the transposed product of quantum_simulator.c:88-89; the legacy
 * wrapper transposes for you).
 *
 * All functions return QSIM_OK (0) or a QSIM_ERR_* code; qsim_last_error() gives the message.
 * The library never falls back to a CPU path: without a usable HIP device qsim_create fails.
 */
#ifndef QSIM_H
#define QSIM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct qsim_state qsim_state;     /* one state vector (or one shard of it) resident on one GPU */
typedef struct qsim_circuit qsim_circuit; /* parsed gate list, host memory */

enum {
    QSIM_OK = 0,
    QSIM_ERR_ARG = 1,    /* bad argument (NULL handle, qubit out of range, ...) */
    QSIM_ERR_ALLOC = 2,  /* host or device allocation failed  ("Malloc error", quantum_simulator.c:170) */
    QSIM_ERR_DEVICE = 3, /* HIP runtime error / no device */
    QSIM_ERR_OPEN = 4,   /* cannot open circuit file          (quantum_simulator.c:128-131) */
    QSIM_ERR_PARSE = 5   /* unknown token / malformed circuit (quantum_simulator.c:212-223) */
};

/* Gate kinds in a qsim_circuit. */
enum { QSIM_GATE_U1 = 1, QSIM_GATE_CX = 2, QSIM_GATE_U2 = 3, QSIM_GATE_U3 = 4, QSIM_GATE_U4 = 5, QSIM_GATE_U5 = 6, QSIM_GATE_U6 = 7, QSIM_GATE_U7 = 8, QSIM_GATE_U8 = 9 /* U3..U8: scheduler output only */ };

/* Options for qsim_set_option. */
enum {
    /* How adjacent gates are merged before launch:
     *   0  one launch per gate                      (quantum_simulator_naive.cu:163-189)
     *   1  per-qubit 2x2 accumulation, flush at CX  (quantum_simulator_preproces.cu:215-269)
     *   2  pair clusters folded into one 4x4        (quantum_simulator_4x4.cu:327-501)
     *   3  level 2 + cache-blocked passes: several clusters applied to an LDS-resident tile per
     *      launch, op list read from device memory  (idea of quantum_simulator_preproces_constant.cu:169-178,
     *      done with a full grid)                   [default]                                        */
    QSIM_OPT_FUSE = 1,
    QSIM_OPT_PROFILE = 2,      /* 1: bracket every launch with HIP events on the engine's stream; 2: also record each tile pass's block forms (qsim_launch_log_blocks) */
    QSIM_OPT_TILE_BITS = 3,    /* log2 amplitudes per LDS tile for level 3 (8..13, default 12 = 64 KiB) */
    QSIM_OPT_TILE_LOW_BITS = 4,/* contiguous low index bits always inside a tile (2..6, default 3 -> 128-B runs, 9 free high-qubit slots) */
    QSIM_OPT_MAX_PENDING = 5,  /* queued gates that force a flush (default 1<<16) */
    QSIM_OPT_TILE_MAX_OPS = 6, /* upper bound on fused blocks per tile pass (default 32) */
    QSIM_OPT_GRID_CAP = 7,     /* 0: one workgroup per work tile; >0: at most that many workgroups (grid-stride loop) */
    QSIM_OPT_TILE_PAD_FROM = 9,/* first index bit used to fill unused high slots of a tile (default 10) */
    QSIM_OPT_TILE_THREADS = 8, /* threads per tile workgroup: 0 auto (256 below 2^12 amplitudes, else 512), 256, 512, 1024 */
    QSIM_OPT_DEBUG_SKIP_OPS = 10,/* measurement aid, default 0: 1 = tile passes move their tiles HBM -> LDS -> HBM but apply
                                  * no blocks (amplitudes are then WRONG); splits a pass's memory time from its compute time */
    QSIM_OPT_PLAN_CACHE = 13,  /* default 1: the plans of the last few flushed gate queues are kept (passes, bit orders, blocks on the
                                * device); a queue with the same gates is replayed without scheduling or uploads.  0 = plan anew */
    QSIM_OPT_PINGPONG = 14,    /* tile passes out of place: each reads the state from one buffer and writes it to a second one, and the
                                * two swap (an even number of times per flush: the state is back in its own buffer afterwards).  The
                                * same bytes move ~5 % faster than in place at n = 30.  0 never, 1 (default) for states of >= 8 GiB
                                * when the second buffer fits (allocated on first use, or lent with qsim_set_spare_buffer), 2 always */
    QSIM_OPT_SPARSE_START = 15,/* default 1: after a reset the state is zero wherever a qubit no pass has mixed yet is 1, and the tile
                                * passes only visit the rest: the first pass of a circuit writes one tile, the second a few hundred,
                                * the full sweeps start when every qubit has been inside a tile (memory outside that support is
                                * written as zeros the moment anything else looks at the buffer).  0 = every pass sweeps the register */
    QSIM_OPT_DEBUG_PLAN_KEY = 16,/* test aid, default 0: k != 0 = every flushed queue gets the plan-cache key k, i.e. all circuits collide;
                                  * results must not change — a cached plan is only replayed after its gate list, options and
                                  * support have been compared with the queue's (qsim_plan_cache_stats counts the collisions) */
    QSIM_OPT_DEBUG_TILE_ORDER = 12,/* measurement aid, default 0: k > 0 = every tile pass walks its high tile bits in a pseudo-random
                                  * order seeded by k (results are unchanged: the order only decides which bits lanes, waves and
                                  * registers walk) */
    QSIM_OPT_DEBUG_SKIP_MEM = 11 /* measurement aid, default 0: 1 = tile passes apply their blocks to zero-filled tiles and
                                  * neither load nor store the state (amplitudes are then WRONG): the block phase alone */
};

/* Kernel classes reported by qsim_get_stats. */
enum {
    QSIM_K_INIT = 0,
    QSIM_K_GATE1 = 1,  /* dense 2x2, target bit >= 6: two coalesced streams            */
    QSIM_K_GATE1_LO = 2,/* dense 2x2, target bit < 6: in-wave shuffle butterfly         */
    QSIM_K_PHASE = 3,  /* diag(1, lambda): touches the bit=1 half only                 */
    QSIM_K_CX = 4,     /* swap on the control=1 half                                   */
    QSIM_K_GATE2 = 5,  /* dense 4x4 (any bit positions)                                */
    QSIM_K_TILE = 6,   /* cache-blocked multi-op pass                                  */
    QSIM_K_PACK = 7,   /* shard re-layout before a global<->local qubit exchange       */
    QSIM_K_COUNT = 8
};

typedef struct {
    uint64_t gates;                     /* gate statements accepted (pre-fusion)                     */
    uint64_t launches;                  /* kernels launched                                          */
    double algorithmic_bytes;           /* sum over launches of the bytes the pass must move         */
    uint64_t k_launches[QSIM_K_COUNT];
    double k_bytes[QSIM_K_COUNT];       /* algorithmic bytes per class                               */
    double k_ms[QSIM_K_COUNT];          /* HIP-event time per class (QSIM_OPT_PROFILE=1), else 0     */
} qsim_stats;

/* ---- device / lifecycle -------------------------------------------------------------------------- */
int qsim_device_count(void);
/* Creates the HIP context of `device` (first-touch cost of the runtime, a few hundred ms per process).  The C host
 * calls it before starting its clock: it is process start-up, not part of the gate path the reference times. */
int qsim_device_init(int device);
const char *qsim_last_error(void);

/* Allocates 2^num_q amplitudes on `device` and sets |0...0>.  Replaces the malloc + init loop of the
 * `qubit` statement (quantum_simulator.c:168-177) and init_state_vector (quantum_simulator_naive.cu:64-70). */
int qsim_create(qsim_state **out, int num_q, int device);
/* Same, with the amplitudes held as fp32 complex (8 bytes each): the precision of the reference's CUDA variants
 * (`cuFloatComplex`, quantum_simulator_naive.cu:38,64-70).  Half the HBM bytes per pass and one more qubit per GPU;
 * every entry point keeps its double-typed interface (matrices are rounded once when a pass is built, qsim_read /
 * qsim_write convert), sums (norm, sampling) still accumulate in fp64.  Not the parity configuration: results agree
 * with quantum_simulator.c to fp32 rounding (~1e-6 per amplitude), not to 1e-10.  Clusters are fp64 only. */
int qsim_create_f32(qsim_state **out, int num_q, int device);
/* Same as qsim_create / qsim_create_f32 (precision_bits 64 / 32) with the amplitude buffer allocated on a helper thread: the call
 * returns at once — hipMalloc of a 16 GiB state takes 0.04-0.25 s, the largest single item of a cold run
 * (quantum_simulator.c:168-177 has malloc in the same place) — so that parsing, options, gate queueing and the schedule choice
 * run beside it.  The first call that needs the buffer waits for it; an allocation failure surfaces there as QSIM_ERR_ALLOC. */
int qsim_create_async(qsim_state **out, int num_q, int device, int precision_bits);
int qsim_precision_bits(const qsim_state *s); /* 64 or 32; -1 for NULL */
/* Same, on caller-owned device memory of 16<<num_q bytes (e.g. a torch tensor's storage). */
int qsim_create_external(qsim_state **out, int num_q, int device, void *device_amps);
void qsim_destroy(qsim_state *s);
int qsim_reset(qsim_state *s); /* back to |0...0>; drops queued gates.  Lazy: written by the first pass that can
                                 * generate it in LDS, or by the init kernel when anything else comes first */
/* Same for one shard of a larger register: holds_index0 = 0 gives the all-zero vector (the shard does not contain
 * basis index 0). */
int qsim_reset_shard(qsim_state *s, int holds_index0);
int qsim_num_qubits(const qsim_state *s);
int qsim_set_option(qsim_state *s, int option, long value);
long qsim_get_option(const qsim_state *s, int option);

/* ---- gates: queued in program order, merged and launched at flush ---------------------------------- */
/* execute_single_qubit_gate (quantum_simulator.c:81-92), kernel_gate (quantum_simulator_naive.cu:72-95).
 * U: 4 complex, row-major, standard U·v. */
int qsim_apply_1q(qsim_state *s, const double *U, int target);
/* execute_cnot (quantum_simulator.c:94-106), kernel_cnot (quantum_simulator_naive.cu:97-122). */
int qsim_apply_cx(qsim_state *s, int control, int target);
/* kernel_gate_4 (quantum_simulator_4x4.cu:109-146): U is 16 complex, row/col index = (bit q_hi, bit q_lo). */
int qsim_apply_2q(qsim_state *s, const double *U, int q_hi, int q_lo);
int qsim_flush(qsim_state *s); /* schedule + launch everything queued; returns without waiting */
/* QSIM_OPT_PLAN_CACHE bookkeeping: plans held, flushes served by replaying one, and 64-bit key matches that were REJECTED because
 * the cached plan's gate list / options / support differed from the queue's (the key only finds candidates). */
int qsim_plan_cache_stats(const qsim_state *s, uint64_t *plans, uint64_t *replays, uint64_t *key_collisions);
int qsim_sync(qsim_state *s);  /* flush, then wait for the stream */

/* ---- amplitudes in / out (the reference's final cudaMemcpy D2H, quantum_simulator_naive.cu:193-194) -- */
int qsim_read(qsim_state *s, uint64_t first, uint64_t count, double *out_re_im);
int qsim_write(qsim_state *s, uint64_t first, uint64_t count, const double *in_re_im);
int qsim_norm2(qsim_state *s, double *out); /* sum |a|^2 computed on the device */
void *qsim_device_ptr(qsim_state *s);        /* amplitude array in HBM: launches pending gates and writes a lazily held
                                              * |0...0> first (work is queued on qsim_stream(), not waited for); NULL on error */
void *qsim_stream(qsim_state *s);            /* the hipStream_t every launch goes to */

/* ---- measurement post-path (dead code in the reference's main, quantum_simulator.c:67-73) ---- */
/* compute_state_cumulative_distribution + the search of measurement (quantum_simulator.c:256-283) without ever
 * materialising the 2^n-entry cumulative array: |a|^2 is summed per block of 4096 amplitudes on the device, the
 * block sums are prefix-summed on the host, and only the block a draw falls into is scanned.  For every random
 * number in randoms[0..shots) the result is the first basis index whose cumulative probability is non-zero and
 * not below it, or 2^n - 1 (exactly the reference's loop; summation order differs from its strictly sequential
 * one, so a draw within ~1e-15 of a boundary may land on the neighbouring index). */
int qsim_sample(qsim_state *s, const double *randoms, long shots, uint64_t *out_indices);
/* The random number measurement() draws (:271-276): ten rand() values, each scaled by a further 1/RAND_MAX. */
double qsim_draw_randn(void);
/* putb (:285-293): `len` binary digits of n, most significant first, NUL-terminated into buf (len + 1 bytes). */
void qsim_putb(long long n, int len, char *buf);

/* ---- measured pass geometry (planning, in the sense of FFTW's wisdom) -------------------------------------------------
 * A cache-blocked pass walks nine "high" index bits besides the contiguous low ones; which of them the lanes of a wave,
 * the waves of a workgroup and the registers of a lane walk changes nothing in the result and up to 2x in the pass's
 * HBM time.  qsim_tune_circuit plans `circuit` as qsim_run_circuit would and times every pass of
 * the schedule under up to max_candidates orders of its bits (ascending first, then seeded permutations; budget_ms
 * bounds the total, 0 = unbounded); the fastest order per (register size, precision, tile shape, bit set) is kept in
 * a process-wide table that every later run with that geometry uses.  The state's contents are clobbered: it is left
 * reset to |0...0>.  Not part of any timed region — call it once per circuit shape, like building a plan. */
typedef struct {
    int tile_passes;      /* tile passes in the circuit's schedule */
    int already_known;    /* ... whose geometry was in the table already */
    int passes_tuned;     /* ... measured now */
    int passes_reordered; /* ... for which an order beat ascending by more than the timing noise */
    int candidates_timed;
    double ms_ascending;  /* sum over the measured passes: time in ascending order */
    double ms_best;       /* ... and in the order kept */
    double seconds;       /* wall time spent measuring */
} qsim_tune_report;
int qsim_tune_circuit(qsim_state *s, const qsim_circuit *circuit, int max_candidates, double budget_ms, qsim_tune_report *report);
/* Only the first half of that planning step, without any timing: schedules the circuit under a handful of scheduler
 * settings (clusters that commute may or may not overtake each other, ...), keeps the one whose passes move the fewest
 * bytes for this circuit — for a run from a reset and for a run on a dense state — and returns.  qsim_tune_circuit does
 * this too. */
int qsim_choose_schedule(qsim_state *s, const qsim_circuit *circuit);
/* The same choice for a run from a reset, for as long as the buffer of a state made by qsim_create_async is still being allocated
 * and no longer: the candidates scheduled by then compete (on up to 16 host threads), the default always does.  Returns at
 * once when the buffer is there already.  What bin/qsim does between the parse and the first launch. */
int qsim_choose_schedule_while_allocating(qsim_state *s, const qsim_circuit *circuit);
/* The same for a circuit that will run on a state that is NOT fresh from a reset (dense_start != 0): the schedule of such a
 * run differs in its first passes (QSIM_OPT_SPARSE_START), e.g. a shard's local gates after its first exchange. */
int qsim_tune_circuit_from(qsim_state *s, const qsim_circuit *circuit, int max_candidates, double budget_ms, qsim_tune_report *report,
                           int dense_start);
/* Both for a run that finds the state with exactly the support `support` (index bits that may be 1: 0 fresh from a reset, all
 * ones dense, the mask given to qsim_set_support after a sparse exchange): a shard's local steps between exchanges. */
int qsim_choose_schedule_for(qsim_state *s, const qsim_circuit *circuit, uint64_t support);
/* Index bits that may be 1 somewhere in the state after `circuit` ran on a state with support `support`, as the engine itself
 * will track it (the circuit is scheduled as qsim_flush will schedule it, remembered schedule choice included; nothing runs). */
int qsim_support_after(qsim_state *s, const qsim_circuit *circuit, uint64_t support, uint64_t *after);
int qsim_tune_circuit_support(qsim_state *s, const qsim_circuit *circuit, int max_candidates, double budget_ms, qsim_tune_report *report, uint64_t support);
long qsim_tune_table_size(void);
void qsim_tune_table_clear(void);
int qsim_tune_table_save(const char *path);  /* text, one geometry per line */
long qsim_tune_table_load(const char *path); /* entries loaded (malformed lines are skipped), -1 if the file cannot be read */

/* ---- sharded states (new: the reference is single-device) --------------------------------
 * A qsim_state may hold one contiguous shard of a larger register: the caller (one process per GPU)
 * keeps the top log2(P) index bits as the rank id and owns the logical->physical qubit map.
 * qsim_pack_bits re-lays the shard out ahead of a global<->local qubit exchange:
 *     dst[(block << (n-k)) | rest] = amps[src],  block = the k bits of src at positions bits[0..k)
 *                                                (ascending), rest = the other n-k bits in order,
 * so block b is the contiguous piece rank-group member b must receive (RCCL send/recv or all-to-all of
 * 16<<(n-k) byte blocks over xGMI).  dst is caller-owned device memory of 16<<n bytes. */
int qsim_pack_bits(qsim_state *s, const int *bits, int nbits, void *dst_device);
/* The same re-layout with one destination per block (nbits <= 3): block b is written to dst_blocks[b] (2^(n-k)
 * amplitudes each), e.g. straight into the spare buffer of the group member that will own it — another shard's buffer
 * on the same device or a peer-mapped one — so that pack and transfer are one kernel.  qsim_swap_buffer then makes the
 * spare buffer (now holding the shard's new contents) the state and hands back the old one. */
int qsim_pack_bits_to(qsim_state *s, const int *bits, int nbits, void *const *dst_blocks);
int qsim_swap_buffer(qsim_state *s, void **buffer_device);
/* The receiving side of an exchange knows where its new contents can be non-zero (a run starts from |0...0>, and a qubit
 * stays |0> until a gate mixes it): qsim_set_support declares every amplitude whose index has a bit outside `support` zero BY
 * DEFINITION — that memory need not have been written — which is the state the first tile passes of a run leave behind
 * (QSIM_OPT_SPARSE_START): later tile passes visit only that part, anything else gets the zeros written out first.
 * qsim_get_support reports the current situation without touching the buffer: *support = index bits that may be 1 (all
 * ones: dense), *kind = 1 while the state is a basis state amp0 * |0...0> that no kernel has written yet (amp0 = 0: the
 * all-zero vector of a shard that holds nothing — gates applied to it are dropped), else 0.
 * qsim_pack_bits_sparse is qsim_pack_bits / qsim_pack_bits_to (dst_blocks != NULL) without the detour of writing the zeros out
 * first: amplitudes outside the state's support are packed as zeros straight away (never loaded), and the blocks in skip_blocks
 * (bit b: block b; its destination may be NULL) are not written at all — nobody will look at them.
 * qsim_state_buffer: the amplitude buffer as is — nothing launched, nothing written (for a caller about to overwrite it). */
int qsim_set_support(qsim_state *s, uint64_t support);
int qsim_get_support(qsim_state *s, uint64_t *support, int *kind, double *amp0);
/* 1 when the state is the all-zero vector of a shard that holds nothing (qsim_reset_shard(s, 0), nothing written since); does
 * not flush. */
int qsim_holds_nothing(const qsim_state *s);
int qsim_pack_bits_sparse(qsim_state *s, const int *bits, int nbits, void *dst_device, void *const *dst_blocks, uint32_t skip_blocks);
void *qsim_state_buffer(qsim_state *s);
/* qsim_flush and the re-layout of qsim_pack_bits_sparse in ONE call: when the last pass of the queue is a tile pass of the
 * default shape, its stores write the state re-laid-out (*fused = 1) and the exchange costs no sweep of its own; otherwise the
 * passes run as usual and the pack kernel follows (*fused = 0).  The output is one buffer (`out`; NULL: the state's spare
 * buffer) in which source index bit bits[j] lands on bit to_bits[j] (NULL: num_q - nbits + j, i.e. the block index on top of
 * a shard-sized buffer), the other bits close ranks below, and konst is ORed into the index (a cluster that keeps every
 * shard's buffer in one allocation addresses "block b of member j" that way).  needed: source index bits that may be 1 where
 * the receivers expect data (a fusing pass over a partially written state only writes inside its support; if that does not
 * cover `needed` the pack kernel, which writes the zeros, is used).  nbits = 4..8 (groups of 16 and more shards) and fp32
 * states always take the pack kernel (*fused = 0) and only know the one-buffer layout (to_bits NULL, konst 0).  Afterwards the state's own buffer holds stale data:
 * hand it its new contents (receives, qsim_swap_buffer) and say what they are (qsim_set_support / qsim_reset_shard). */
int qsim_flush_pack(qsim_state *s, const int *bits, int nbits, const int *to_bits, uint64_t konst, void *out_device, uint64_t needed,
                    uint32_t skip_blocks, void **packed_at, int *fused);
/* Lends the state a second device buffer of 2^n amplitudes for out-of-place tile passes (QSIM_OPT_PINGPONG) — e.g. the
 * exchange scratch of a sharded run, idle between exchanges.  The caller keeps ownership; between qsim_flush / qsim_sync
 * and the next gate the buffer is the caller's to use (its contents are garbage).  NULL takes it back. */
int qsim_set_spare_buffer(qsim_state *s, void *buffer_device);
/* Block sums / block contents for blocks that are bit-deposits instead of ranges: block w = the amplitudes at
 * deposit(w, hi_mask) | deposit(i, lo_mask), i = 0 .. 2^popcount(lo_mask) - 1 (deposit spreads the low bits of its first
 * argument over the set bits of the mask, lowest first; the masks are disjoint).  After exchanges a LOGICAL block of the
 * measurement post-path (quantum_simulator.c:256-283) is such a set of a shard's local positions: the sums are formed
 * on the device and only 2^popcount(hi_mask) doubles, or one block, reach the host.
 *   qsim_block_prob_masked: out[w] = sum_i |a|^2             qsim_gather_masked: out[i] = a[base | deposit(i, lo_mask)] */
int qsim_block_prob_masked(qsim_state *s, uint64_t hi_mask, uint64_t lo_mask, double *out_host);
int qsim_gather_masked(qsim_state *s, uint64_t base, uint64_t lo_mask, double *out_host_re_im);
/* Multiplies every amplitude of the shard by (re, im): a diagonal gate on a global qubit is a per-rank scalar. */
int qsim_scale(qsim_state *s, double re, double im);

/* ---- clusters: P = 2^p shards driven by ONE process (the C host's multi-GPU path) --------------------------
 * Each shard is a qsim_state on devices[r] (NULL: round-robin over the visible devices; the same device may
 * repeat, which gives "virtual shards" for validation on fewer GPUs).  qsim_cluster_run_circuit plans the
 * circuit (logical->physical qubit map, communication-free handling of diagonal gates / controls on global
 * qubits, Belady choice of the qubits that become global) and executes it: local steps through the ordinary
 * engine, exchanges as qsim_pack_bits + device-to-device block copies.  The one-process-per-GPU driver
 * (gpu_quantum_simulator_amd/distributed.py) runs the same plan with RCCL send/recv. */
typedef struct qsim_cluster qsim_cluster;
int qsim_cluster_create(qsim_cluster **out, int num_q, int num_shards, const int *devices);
void qsim_cluster_destroy(qsim_cluster *c);
int qsim_cluster_num_shards(const qsim_cluster *c);
qsim_state *qsim_cluster_shard(qsim_cluster *c, int shard);
int qsim_cluster_set_option(qsim_cluster *c, int option, long value);
int qsim_cluster_reset(qsim_cluster *c); /* |0...0>, identity qubit map */
/* ONE circuit per reset (compute_state_vector semantics): the plan's free first qubit placement and its sparse exchanges are
 * only right from |0...0>, so a call that does not follow qsim_cluster_reset fails with QSIM_ERR_ARG instead of dropping data. */
int qsim_cluster_run_circuit(qsim_cluster *c, const qsim_circuit *circuit);
int qsim_cluster_sync(qsim_cluster *c);
/* Planning for a circuit the cluster will run (repeatedly): the shard plan is built and kept, and every shard's local steps go
 * through the schedule choice of qsim_choose_schedule — and, with max_candidates > 1, through the measured tile-bit orders of
 * qsim_tune_circuit (budget_ms for all shards together, 0 = unbounded) — each for the support the shard will have at that
 * point of the run.  The steps are planned in order for all shards, and every exchange of the kept plan is told what its
 * senders will have written under the schedules chosen (qsim_support_after), so that their last tile pass can do the
 * re-layout.  Outside any timed region; results never depend on it.  Leaves the cluster reset.
 * QSIM_TRACE_PACK=1 in the environment: one line on stderr for every re-layout that needed a sweep of its own, with the reason. */
int qsim_cluster_plan(qsim_cluster *c, const qsim_circuit *circuit, int max_candidates, double budget_ms);
int qsim_cluster_read(qsim_cluster *c, uint64_t logical_first, uint64_t count, double *out_re_im);
int qsim_cluster_norm2(qsim_cluster *c, double *out);
/* qsim_sample for a sharded state: basis indices in LOGICAL order for random numbers in [0,1] (measurement(),
 * quantum_simulator.c:270-283), whatever qubit map the exchanges left behind.  Every shard forms the |a|^2 sums of the
 * logical 2^12-amplitude blocks it holds a part of on its own device (qsim_block_prob_masked); the host adds the P
 * partial sums per block in shard order and fetches only the blocks the draws land in (qsim_gather_masked). */
int qsim_cluster_sample(qsim_cluster *c, const double *randoms, long shots, uint64_t *out_indices);
int qsim_cluster_exchange_stats(const qsim_cluster *c, uint64_t *exchanges, double *bytes_per_shard); /* bytes: if every block travelled */
/* What all shards together really sent: blocks of shards that hold nothing yet, and blocks for shards that will hold nothing
 * afterwards, stay home (a run starts from |0...0>; qsim_shard_plan_step_support). */
int qsim_cluster_exchange_bytes_moved(const qsim_cluster *c, double *bytes_all_shards);
/* Re-layouts so far, per shard and exchange: done by the last tile pass in front of the exchange / by the pack kernel. */
int qsim_cluster_pack_counts(const qsim_cluster *c, uint64_t *fused, uint64_t *separate);
/* How this cluster moves blocks: "rccl" (every shard on its own device: one ncclGroup of ncclSend/ncclRecv per exchange,
 * stream-ordered), "direct" (all shards on one device: the pack kernel writes into the members' buffers, which then
 * change roles), "copies" (mixed placements: pack + device-to-device copies) or "none" (one shard). */
const char *qsim_cluster_exchange_mode(const qsim_cluster *c);
const char *qsim_cluster_error(void);
/* The plan as an object (host only).  This is what the one-process-per-GPU driver executes: every rank builds the same
 * plan, applies its own local steps with qsim_shard_plan_apply_local on its shard state and performs the exchanges
 * (qsim_pack_bits + send/recv of the blocks) itself.  step kinds: 0 local, 1 exchange (k shard-id bits and k local bit
 * positions, ascending and paired).  local ops: 1 = 2x2 on local qubit a (m = 8 doubles), 2 = cx a -> b,
 * 3 = multiply the shard by the scalar m[0..1]. */
typedef struct qsim_shard_plan qsim_shard_plan;
typedef void (*qsim_local_op_cb)(void *user, int kind, int a, int b, const double *m);
int qsim_shard_plan_create(qsim_shard_plan **out, const qsim_circuit *circuit, int num_shards);
void qsim_shard_plan_free(qsim_shard_plan *p);
int qsim_shard_plan_num_steps(const qsim_shard_plan *p);
int qsim_shard_plan_step(const qsim_shard_plan *p, int step, int *kind, int *k, int *shard_bits, int *local_bits);
int qsim_shard_plan_final_pos(const qsim_shard_plan *p, int *pos /* num_q entries: logical -> physical */);
/* For an exchange step: where the register can be non-zero just before it, as sets of local positions / shard-id bits (a qubit
 * stays |0> until a non-diagonal gate, or a CX whose control may be 1, acts on it).  The same on every rank. */
int qsim_shard_plan_step_support(const qsim_shard_plan *p, int step, uint64_t *mixed_local, uint64_t *mixed_rank);
/* What that means for one shard in the exchange of `step` (host only; what qsim_cluster / qsim_rank_comm act on).  The shard is
 * member `mine` of its group; block b of its packed layout goes to member b and block b of its new contents comes from member b.
 * send / recv: bit b set = that block really travels (never bit `mine`); keep_own: block `mine` stays and holds data; unread:
 * blocks of the packed layout nobody looks at; empty_before / empty_after: the shard holds nothing (all zero) before / after;
 * new_support: local index bits that may be 1 in its new contents (qsim_set_support). */
typedef struct {
    int mine, empty_before, empty_after, keep_own;
    uint32_t send, recv, unread;
    uint64_t new_support;
} qsim_exchange_roles;
int qsim_shard_plan_exchange_roles(const qsim_shard_plan *p, int step, int shard, qsim_exchange_roles *out);
int qsim_shard_plan_local_ops(const qsim_shard_plan *p, int step, int shard, qsim_local_op_cb cb, void *user);
int qsim_shard_plan_apply_local(const qsim_shard_plan *p, int step, int shard, qsim_state *s);
/* Cost model of the plan's exchanges on one fully connected xGMI node: bytes each rank sends, and seconds =
 * sum over exchanges of (pack pass: 2 * shard bytes / pack_gbps) + (largest per-link transfer: 2^-k of the shard /
 * link_gbps; a k-qubit swap uses 2^k - 1 links in both directions at once).  The planner itself keeps the cheaper of
 * two placements (keep far-next-use globals vs swap all log2 P of them) under the same model with default rates. */
/* Geometry planning (qsim_tune_circuit) for one shard's local steps of the plan; leaves the shard state reset. */
int qsim_shard_plan_tune(const qsim_shard_plan *p, int shard, qsim_state *s, int max_candidates, double budget_ms, qsim_tune_report *report);
int qsim_shard_plan_predict(const qsim_shard_plan *p, double link_gbps, double pack_gbps, double *bytes_per_rank, double *seconds);

/* ---- one process per GPU: the exchanges on RCCL (pairwise ncclSend/ncclRecv of half-shards for one global
 * qubit, one group of 2^k - 1 sends + receives for k of them) ----------------------------------------------------------
 * Rank 0 calls qsim_rccl_unique_id and hands the QSIM_RCCL_ID_BYTES bytes to every rank over the launcher's own channel;
 * each rank then joins with qsim_rank_comm_create (scratch: a caller-owned device buffer as large as the shard, or NULL
 * to let the library allocate one).  qsim_rank_comm_exchange = pack kernel + ONE ncclGroup on the shard's stream, no
 * host synchronisation; qsim_rank_comm_stats reports the HIP-event time its stream spent in exchanges (measured while
 * the shard is in profile mode, QSIM_OPT_PROFILE). */
#define QSIM_RCCL_ID_BYTES 128
typedef struct qsim_rank_comm qsim_rank_comm;
int qsim_rccl_unique_id(void *id_bytes);
int qsim_rank_comm_create(qsim_rank_comm **out, qsim_state *shard, int device, int world, int rank, const void *id_bytes, void *scratch_device);
void qsim_rank_comm_destroy(qsim_rank_comm *c);
int qsim_rank_comm_exchange(qsim_rank_comm *c, const int *shard_bits, const int *local_bits, int k);
/* The exchange of step `step` of a plan, using what the plan knows about the state there (qsim_shard_plan_step_support): ranks
 * that hold nothing do not pack or send, blocks that are zero throughout are not received, and the shard's engine is told
 * where its new contents can be non-zero (qsim_set_support) or that it holds nothing (qsim_reset_shard). */
/* A plan's steps are only right in order from qsim_reset_shard (|0...0>): a rank the plan takes to hold nothing and that does hold
 * something is refused (QSIM_ERR_ARG).  Exchanges swap at most 5 qubits (groups of 32 ranks). */
int qsim_rank_comm_exchange_step(qsim_rank_comm *c, const qsim_shard_plan *p, int step);
int qsim_rank_comm_stats(qsim_rank_comm *c, uint64_t *exchanges, double *bytes_sent, double *seconds, int reset);
int qsim_rank_comm_pack_counts(const qsim_rank_comm *c, uint64_t *fused, uint64_t *separate); /* as qsim_cluster_pack_counts */
/* Diagnostic: `count` doubles of the shard through ncclSend -> ncclRecv to this same rank, compared on the host (drives
 * the RCCL call path where only one GPU is present). */
int qsim_rank_comm_loopback(qsim_rank_comm *c, uint64_t count);

int qsim_get_stats(qsim_state *s, qsim_stats *out); /* waits for outstanding profile events */
int qsim_reset_stats(qsim_state *s);
/* Per-launch record (QSIM_OPT_PROFILE=1) since the last qsim_reset_stats: returns the number of records and,
 * for 0 <= index < count, fills the kernel class, the fused blocks in that launch, the tile's high-qubit
 * mask and the HIP-event time. */
long qsim_launch_log(qsim_state *s, long index, int *kernel_class, int *n_ops, uint64_t *high_mask, double *ms);
/* For a tile pass of the launch log: its high tile bits in tile-local order (order[j] = global bit of tile-local bit L+j;
 * the first three are walked by the lanes of a wave, the next by its waves, the last by a lane's registers); count = 0
 * for other kernels.  order needs room for 10 entries. */
int qsim_launch_log_order(qsim_state *s, long index, int *order, int *count);
/* ... and the fraction of the register's tiles it worked on (1 for a full sweep and for other kernels; less while the state's
 * support is partial, QSIM_OPT_SPARSE_START / qsim_set_support). */
int qsim_launch_log_visited(qsim_state *s, long index, double *visited);
/* ... and what its blocks looked like, one byte per block in order (tile passes; *count = 0 otherwise; at most `cap` are written):
 * bits 0-1 log2 of the entries per row the block is evaluated with (1, 2 or 4), bits 2-4 its qubits inside the tile, bit 5 set when
 * at least half of its rows are identity rows, bits 6-7 its selector qubits outside the tile.  Recorded under QSIM_OPT_PROFILE = 2
 * only (host work per launch).  Data for the pass-time model. */
int qsim_launch_log_blocks(qsim_state *s, long index, uint8_t *codes, int cap, int *count);

/* ---- circuits: the tokenizer of compute_state_vector (quantum_simulator.c:115-254) ---------------- */
/* Parses the OPENQASM-3 subset of quantum_simulator.c (two header statements, `qubit[n] q;` or
 * `qubit q[n];`, gates cx x sx z s sdg t tdg rz(<number>) h, operands q[k] or $k).  A file whose first
 * token is a number is read in the CUDA variants' `<num_qubit> <num_gates>` form
 * (quantum_simulator_naive.cu:239-240). */
int qsim_circuit_parse_file(const char *path, qsim_circuit **out);
int qsim_circuit_parse_text(const char *text, size_t len, qsim_circuit **out);
int qsim_circuit_create(int num_q, qsim_circuit **out);
void qsim_circuit_free(qsim_circuit *c);
int qsim_circuit_num_qubits(const qsim_circuit *c);
long qsim_circuit_num_gates(const qsim_circuit *c);
int qsim_circuit_append_1q(qsim_circuit *c, const double *U, int target);
int qsim_circuit_append_cx(qsim_circuit *c, int control, int target);
int qsim_circuit_append_2q(qsim_circuit *c, const double *U, int q_hi, int q_lo); /* q_hi > q_lo */
/* kind: QSIM_GATE_*; q0 = target / control / q_hi, q1 = -1 / target / q_lo; U receives 8 (U1) or 32 (U2) doubles */
int qsim_circuit_gate(const qsim_circuit *c, long index, int *kind, int *q0, int *q1, double *U);
/* Text of the parse error (the reference prints "Unknown token: %s", quantum_simulator.c:213). */
const char *qsim_circuit_error(void);
/* Queues gates [first, first+count) of the circuit on the state (count < 0: to the end). */
int qsim_run_circuit(qsim_state *s, const qsim_circuit *c, long first, long count);
/* The gate table (quantum_simulator.c:184-211): name -> 2x2, standard orientation. Returns QSIM_GATE_U1,
 * QSIM_GATE_CX for "cx", or 0 for an unknown token. */
int qsim_gate_matrix(const char *token, double *U);

/* ---- scheduling only (host code, no device): what a flush WOULD launch ----------------------------- */
/* Runs the fusion scheduler on a circuit and reports launches and algorithmic bytes per kernel class
 * for a state of num_q qubits at the given fuse level.  Used by tests and by the planner. */
int qsim_plan_circuit(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, qsim_stats *out);
/* The same for a run that does not start from a reset: initial_support = index bits that may be 1 in the state the circuit
 * finds (all ones: a dense state, e.g. a shard after its second exchange; 0: fresh from a reset, what qsim_plan_circuit assumes). */
int qsim_plan_circuit_from(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, uint64_t initial_support, qsim_stats *out);
/* The same schedule pass by pass (at most `cap` entries are written, *count receives the number of passes): the kernel class,
 * the blocks of a tile pass, the index bits inside its tile (all bits for a single-gate kernel), the fraction of the register it
 * visits, its algorithmic bytes and the bytes-equivalent the planning steps price it at (pass-time model).  Host-side models
 * use it to tell which passes could be pipelined beside an exchange (bench.py exchange_model). */
typedef struct {
    int32_t kernel_class, blocks;
    uint64_t tile_mask;
    double visited, bytes, cost_bytes;
} qsim_pass_info;
int qsim_plan_passes(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, uint64_t initial_support, qsim_pass_info *out, int cap, int *count);
/* Same scheduler, op by op: calls `cb` for every fused block in launch order with the pass it belongs to, the
 * kernel class of that pass, the block kind (QSIM_GATE_U1 / _CX / _U2 .. _U8 by qubit count), its qubits (most
 * significant first; CX: control, target) and its matrix (2^nq x 2^nq complex, row-major; NULL for CX).  A block of a
 * tile pass may span up to 8 qubits: at most 6 inside the tile plus, listed first, at most 2 outside it in which the
 * matrix is block-diagonal (they select the sub-block a tile gets).  Lets a CPU test replay the schedule with numpy
 * and compare it with the unfused circuit. */
typedef void (*qsim_sched_cb)(void *user, int pass, int kernel_class, int kind, const int *qubits, int nq,
                              const double *U, int gates_folded);
int qsim_schedule_circuit(const qsim_circuit *c, int fuse, int tile_bits, int tile_low_bits, int tile_max_ops,
                          qsim_sched_cb cb, void *user);

#ifdef __cplusplus
}
#endif
#endif /* QSIM_H */
