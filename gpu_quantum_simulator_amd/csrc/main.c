/*
 * main.c — `qsim <circuit_file> [number_of_measurement]`: the C host, drop-in for the CLI of
 * quantum_simulator.c:32-79 (and of the CUDA variants, which take the file only, naive.cu:135-139).
 *
 * stdout is exactly what the reference prints: one "%lf\n" line with the elapsed seconds (parse + state
 * allocation and init + gates + device sync; HIP context creation, the result copy and any dump are outside), or the reference's error texts with exit code 1.  Everything else is opt-in through the
 * environment and goes to files / stderr so scripts in the style of tester.bash keep working:
 *   QSIM_DUMP=<path>       raw little-endian doubles (re, im) of all 2^n amplitudes
 *   QSIM_DUMP_TEXT=<path>  "<index> <re> <im>" with %.17g, one amplitude per line
 *   QSIM_STATS=1           one JSON line on stderr: gates, launches, algorithmic bytes, GB/s
 *   QSIM_MEASURE=1         after the time line, the <number_of_measurement> lines the reference has commented out
 *                          (quantum_simulator.c:67-73): "MEASUREMENT: <bits> (<index>)", drawn like :270-283
 *   QSIM_SHARDS=P          split the register over P = 2^p shards driven by this process: devices round-robin over the
 *                          visible GPUs (all on one GPU = virtual shards); QSIM_DUMP then writes LOGICAL order
 *   QSIM_PRECISION=32      hold the amplitudes as fp32 complex like the CUDA variants (naive.cu:38); default 64
 *   QSIM_WISDOM=<path>     measured pass geometries (qsim_tune_circuit, include/qsim.h): loaded before the run if the file exists
 *   QSIM_TUNE=1            measure them for this circuit first (seconds of planning, outside the printed time) and, with
 *                          QSIM_WISDOM, save the table afterwards
 *   QSIM_DEVICE, QSIM_FUSE, QSIM_TILE_BITS, QSIM_TILE_LOW_BITS, QSIM_TILE_MAX_OPS, QSIM_GRID_CAP, QSIM_PROFILE, QSIM_PINGPONG
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>
#include <time.h>

#include "../../include/qsim.h"

int qsim_apply_env_options(qsim_state *s);
void qsim_print_format_help(const char *first_line);
int qsim_dump_raw(qsim_state *s, const char *path);

static double wall_seconds(void) {
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + (double)tv.tv_usec * 1e-6;
}

static int dump_text(qsim_state *s, const char *path) {
    const int n = qsim_num_qubits(s);
    const uint64_t N = 1ULL << n, chunk = N < (1ULL << 20) ? N : (1ULL << 20);
    double *buf = (double *)malloc((size_t)chunk * 16);
    FILE *f = fopen(path, "w");
    int rc = (buf && f) ? QSIM_OK : QSIM_ERR_ALLOC;
    for (uint64_t at = 0; rc == QSIM_OK && at < N; at += chunk) {
        rc = qsim_read(s, at, chunk, buf);
        for (uint64_t i = 0; rc == QSIM_OK && i < chunk; i++)
            fprintf(f, "%llu %.17g %.17g\n", (unsigned long long)(at + i), buf[2 * i], buf[2 * i + 1]);
    }
    if (f) fclose(f);
    free(buf);
    return rc;
}

/* The multi-GPU path of the C host: same stdout contract, state sharded over QSIM_SHARDS shards. */
static int run_sharded(qsim_circuit *c, int shards, double t_start, long shots) {
    qsim_cluster *cl = NULL;
    const char *v;
    if ((v = getenv("QSIM_PRECISION")) && atoi(v) == 32) { /* shards are fp64 only: say so instead of running fp64 silently */
        printf("ERROR: QSIM_PRECISION=32 is not supported together with QSIM_SHARDS (sharded states are fp64)\n");
        printf("ERROR while parsing quantum circuit\n");
        exit(1);
    }
    int ndev = qsim_device_count(), first_dev = (v = getenv("QSIM_DEVICE")) && *v ? atoi(v) : 0;
    int *devices = (int *)malloc((size_t)shards * sizeof(int)); /* round-robin over the visible GPUs, starting at QSIM_DEVICE */
    for (int r = 0; devices && r < shards; r++) devices[r] = ndev > 0 ? (first_dev + r) % ndev : 0;
    int rc = devices ? qsim_cluster_create(&cl, qsim_circuit_num_qubits(c), shards, devices) : QSIM_ERR_ALLOC;
    free(devices);
    static const struct { const char *env; int opt; } fwd[] = {
        {"QSIM_FUSE", QSIM_OPT_FUSE}, {"QSIM_TILE_BITS", QSIM_OPT_TILE_BITS}, {"QSIM_TILE_LOW_BITS", QSIM_OPT_TILE_LOW_BITS},
        {"QSIM_TILE_MAX_OPS", QSIM_OPT_TILE_MAX_OPS}, {"QSIM_GRID_CAP", QSIM_OPT_GRID_CAP}, {"QSIM_PROFILE", QSIM_OPT_PROFILE},
        {"QSIM_TILE_THREADS", QSIM_OPT_TILE_THREADS}, {"QSIM_PINGPONG", QSIM_OPT_PINGPONG}};
    for (size_t i = 0; rc == QSIM_OK && i < sizeof fwd / sizeof fwd[0]; i++)
        if ((v = getenv(fwd[i].env)) && *v) rc = qsim_cluster_set_option(cl, fwd[i].opt, atol(v));
    if (rc == QSIM_OK) rc = qsim_cluster_reset(cl);
    if (rc == QSIM_OK) rc = qsim_cluster_run_circuit(cl, c);
    if (rc == QSIM_OK) rc = qsim_cluster_sync(cl);
    if (rc != QSIM_OK) {
        if (rc == QSIM_ERR_ALLOC) printf("Malloc error\n");
        else printf("ERROR: %s\n", qsim_cluster_error());
        printf("ERROR while parsing quantum circuit\n");
        exit(1);
    }
    const double t_exe = wall_seconds() - t_start;
    printf("%lf\n", t_exe);
    fflush(stdout);
    if ((v = getenv("QSIM_MEASURE")) && *v && atoi(v) && shots > 0) { /* quantum_simulator.c:67-73, on the sharded state */
        const int nq = qsim_circuit_num_qubits(c);
        srand((unsigned)time(NULL));
        for (int i = 0; i < 10; i++) rand();
        char *bits = (char *)malloc((size_t)nq + 1);
        double *r = (double *)malloc((size_t)shots * sizeof(double));
        uint64_t *idx = (uint64_t *)malloc((size_t)shots * sizeof(uint64_t));
        for (long m = 0; r && m < shots; m++) r[m] = qsim_draw_randn();
        if (bits && r && idx && qsim_cluster_sample(cl, r, shots, idx) == QSIM_OK) {
            for (long m = 0; m < shots; m++) {
                qsim_putb((long long)idx[m], nq, bits);
                printf("MEASUREMENT: %s (%llu)\n", bits, (unsigned long long)idx[m]);
            }
        } else {
            fprintf(stderr, "qsim: %s\n", qsim_cluster_error());
        }
        free(bits); free(r); free(idx);
    }
    if ((v = getenv("QSIM_DUMP")) && *v) {
        const int n = qsim_circuit_num_qubits(c);
        const uint64_t N = 1ULL << n, chunk = N < (1ULL << 20) ? N : (1ULL << 20);
        double *buf = (double *)malloc((size_t)chunk * 16);
        FILE *f = fopen(v, "wb");
        for (uint64_t at = 0; buf && f && at < N; at += chunk)
            if (qsim_cluster_read(cl, at, chunk, buf) != QSIM_OK || fwrite(buf, 16, (size_t)chunk, f) != (size_t)chunk) {
                fprintf(stderr, "qsim: dump failed: %s\n", qsim_cluster_error());
                break;
            }
        if (f) fclose(f);
        free(buf);
    }
    if ((v = getenv("QSIM_STATS")) && *v && atoi(v)) {
        uint64_t ex = 0;
        double bytes = 0;
        qsim_cluster_exchange_stats(cl, &ex, &bytes);
        fprintf(stderr, "{\"qubits\": %d, \"shards\": %d, \"gates\": %ld, \"seconds\": %.6f, \"exchanges\": %llu, "
                        "\"exchange_bytes_per_shard\": %.0f}\n",
                qsim_circuit_num_qubits(c), shards, qsim_circuit_num_gates(c), t_exe, (unsigned long long)ex, bytes);
    }
    qsim_circuit_free(c);
    qsim_cluster_destroy(cl);
    return 0;
}

int main(int argc, char *argv[]) {
    if (argc < 2) { /* quantum_simulator.c:39-43 */
        printf("QUANTUM CIRCUIT SIMULATOR\n");
        printf("Usage: %s <circuit_file_name> <number_of_measurement>\n", argv[0]);
        exit(1);
    }
    const char *v;
    qsim_circuit *c = NULL;
    qsim_state *s = NULL;

    /* HIP context creation (a few hundred ms, once per process) is start-up, not gate time: outside the clock.  A
     * failure here is reported by qsim_create below, in the reference's own words. */
    (void)qsim_device_init((v = getenv("QSIM_DEVICE")) && *v ? atoi(v) : 0);
    const double t_start = wall_seconds();
    int rc = qsim_circuit_parse_file(argv[1], &c);
    if (rc == QSIM_ERR_OPEN) {
        printf("ERROR: cannot open circuit file\n");
        exit(1);
    }
    if (rc != QSIM_OK) {
        qsim_print_format_help(qsim_circuit_error());
        printf("ERROR while parsing quantum circuit\n"); /* :55-58 */
        exit(1);
    }
    const int shards = (v = getenv("QSIM_SHARDS")) && *v ? atoi(v) : 1;
    if (shards > 1) return run_sharded(c, shards, t_start, argc > 2 ? atol(argv[2]) : 0);
    const int device = (v = getenv("QSIM_DEVICE")) && *v ? atoi(v) : 0;
    const int f32 = (v = getenv("QSIM_PRECISION")) && atoi(v) == 32;
    const double t_parsed = wall_seconds();
    /* The state's buffer is allocated on a helper thread (hipMalloc of 16 GiB: 0.04-0.25 s, the largest single item of a cold
     * run); meanwhile the options are applied and the schedule is chosen among as many candidates as fit into that wait
     * (qsim_choose_schedule_while_allocating: the cold run no longer takes the default schedule because choosing cost a second). */
    rc = qsim_create_async(&s, qsim_circuit_num_qubits(c), device, f32 ? 32 : 64);
    /* One circuit, one run: a second 2^n buffer for out-of-place passes would cost more to allocate (50 ms .. 1 s for 16 GiB,
     * measured) than the ~1.6 % it saves on the passes; QSIM_PINGPONG overrides. */
    if (rc == QSIM_OK) rc = qsim_set_option(s, QSIM_OPT_PINGPONG, 0);
    if (rc == QSIM_OK) rc = qsim_apply_env_options(s);
    if (rc == QSIM_OK && !((v = getenv("QSIM_TUNE")) && *v && atoi(v))) rc = qsim_choose_schedule_while_allocating(s, c);
    if (rc == QSIM_OK) rc = qsim_flush(s); /* nothing is queued yet: this only waits for the buffer ("Malloc error" comes from here) */
    const double t_created = wall_seconds();
    double t_plan = 0.0; /* planning is start-up like context creation: not part of the printed time */
    if (rc == QSIM_OK) {
        const double t0 = wall_seconds();
        const char *w = getenv("QSIM_WISDOM");
        if (w && *w) (void)qsim_tune_table_load(w);
        if ((v = getenv("QSIM_TUNE")) && *v && atoi(v)) {
            rc = qsim_tune_circuit(s, c, 32, 6000.0, NULL);
            if (rc == QSIM_OK && w && *w) (void)qsim_tune_table_save(w);
        }
        t_plan = wall_seconds() - t0;
    }
    if (rc == QSIM_OK) rc = qsim_run_circuit(s, c, 0, -1);
    if (rc == QSIM_OK) rc = qsim_flush(s);
    const double t_launched = wall_seconds();
    if (rc == QSIM_OK) rc = qsim_sync(s);
    if (rc != QSIM_OK) {
        if (rc == QSIM_ERR_ALLOC) printf("Malloc error\n"); /* :170 */
        else printf("ERROR: %s\n", qsim_last_error());
        printf("ERROR while parsing quantum circuit\n");
        exit(1);
    }
    const double t_exe = wall_seconds() - t_start - t_plan;
    printf("%lf\n", t_exe); /* :248 */
    fflush(stdout);

    if ((v = getenv("QSIM_MEASURE")) && *v && atoi(v) && argc > 2) {
        const long shots = atol(argv[2]); /* num_m, quantum_simulator.c:50 */
        const int nq = qsim_num_qubits(s);
        srand((unsigned)time(NULL)); /* RAND PRE-HEAT, :45-47 */
        for (int i = 0; i < 10; i++) rand();
        /* all draws first (same rand() order as the reference's loop, :67-73), then ONE pass over the state */
        char *bits = (char *)malloc((size_t)nq + 1);
        double *r = (double *)malloc((size_t)(shots > 0 ? shots : 1) * sizeof(double));
        uint64_t *idx = (uint64_t *)malloc((size_t)(shots > 0 ? shots : 1) * sizeof(uint64_t));
        for (long m = 0; r && m < shots; m++) r[m] = qsim_draw_randn();
        if (bits && r && idx && shots > 0 && qsim_sample(s, r, shots, idx) == QSIM_OK) {
            for (long m = 0; m < shots; m++) {
                qsim_putb((long long)idx[m], nq, bits);
                printf("MEASUREMENT: %s (%llu)\n", bits, (unsigned long long)idx[m]);
            }
        } else if (shots > 0) {
            fprintf(stderr, "qsim: %s\n", qsim_last_error());
        }
        free(bits); free(r); free(idx);
    }
    if ((v = getenv("QSIM_DUMP")) && *v && qsim_dump_raw(s, v) != QSIM_OK) fprintf(stderr, "qsim: dump failed: %s\n", qsim_last_error());
    if ((v = getenv("QSIM_DUMP_TEXT")) && *v && dump_text(s, v) != QSIM_OK) fprintf(stderr, "qsim: dump failed: %s\n", qsim_last_error());
    if ((v = getenv("QSIM_STATS")) && *v && atoi(v)) {
        qsim_stats st;
        if (qsim_get_stats(s, &st) == QSIM_OK) {
            double kms = 0;
            for (int k = 1; k < QSIM_K_COUNT; k++) kms += st.k_ms[k];
            fprintf(stderr,
                    "{\"qubits\": %d, \"gates\": %llu, \"launches\": %llu, \"algorithmic_bytes\": %.0f, "
                    "\"seconds\": %.6f, \"gate_applies_per_s\": %.3f, \"kernel_ms\": %.3f, "
                    "\"parse_s\": %.6f, \"allocate_s\": %.6f, \"schedule_and_launch_s\": %.6f, \"wait_s\": %.6f}\n",
                    qsim_num_qubits(s), (unsigned long long)st.gates, (unsigned long long)st.launches,
                    st.algorithmic_bytes, t_exe, t_exe > 0 ? (double)st.gates / t_exe : 0.0, kms,
                    t_parsed - t_start, t_created - t_parsed, t_launched - t_created - t_plan, t_start + t_plan + t_exe - t_launched);
        }
    }
    qsim_circuit_free(c);
    qsim_destroy(s);
    return 0;
}
