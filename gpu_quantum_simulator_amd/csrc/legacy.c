/*
 * legacy.c — the reference's three C entry points (quantum_simulator.c:25-27) on top of the handle API.
 * Same names, same argument meaning, same printed lines and exit behaviour; the arithmetic runs on the GPU.
 */
#include <complex.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include "../../include/qsim.h"
#include "../../include/qsim_legacy.h"

static double wall_seconds(void) { /* get_time, quantum_simulator.c:109-113 */
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + (double)tv.tv_usec * 1e-6;
}

static int env_int(const char *name, int dflt) {
    const char *v = getenv(name);
    return (v && *v) ? atoi(v) : dflt;
}

/* Applies QSIM_FUSE / QSIM_TILE_BITS / QSIM_TILE_LOW_BITS / QSIM_GRID_CAP from the environment. */
int qsim_apply_env_options(qsim_state *s) {
    static const struct { const char *env; int opt; } map[] = {
        {"QSIM_FUSE", QSIM_OPT_FUSE}, {"QSIM_TILE_BITS", QSIM_OPT_TILE_BITS},
        {"QSIM_TILE_LOW_BITS", QSIM_OPT_TILE_LOW_BITS}, {"QSIM_TILE_MAX_OPS", QSIM_OPT_TILE_MAX_OPS},
        {"QSIM_GRID_CAP", QSIM_OPT_GRID_CAP}, {"QSIM_PROFILE", QSIM_OPT_PROFILE},
        {"QSIM_TILE_THREADS", QSIM_OPT_TILE_THREADS}, {"QSIM_PINGPONG", QSIM_OPT_PINGPONG}};
    for (size_t i = 0; i < sizeof map / sizeof map[0]; i++) {
        const char *v = getenv(map[i].env);
        if (v && *v) {
            const int rc = qsim_set_option(s, map[i].opt, atol(v));
            if (rc) return rc;
        }
    }
    return QSIM_OK;
}

/* The usage block the reference prints for an unknown token (quantum_simulator.c:213-219). */
void qsim_print_format_help(const char *first_line) {
    printf("%s\n", first_line);
    printf("Input format: \n\n");
    printf("OPENQASM 3.0;\n");
    printf("include \"stdgates.inc\";\n");
    printf("qubit[<num_qubit>] q; or qubit q[<num_qubit>]; \\\\single quantum register \n");
    printf("<quantum_circuit>\n\n");
    printf("Supported operations: cx, x, sx, z, s, sdg, t, tdg, rz, h\n");
}

int qsim_dump_raw(qsim_state *s, const char *path) {
    const int n = qsim_num_qubits(s);
    const uint64_t N = 1ULL << n, chunk = N < (1ULL << 24) ? N : (1ULL << 24);
    double *buf = (double *)malloc((size_t)chunk * 16);
    FILE *f = fopen(path, "wb");
    int rc = (buf && f) ? QSIM_OK : QSIM_ERR_ALLOC;
    for (uint64_t at = 0; rc == QSIM_OK && at < N; at += chunk) {
        rc = qsim_read(s, at, chunk, buf);
        if (rc == QSIM_OK && fwrite(buf, 16, (size_t)chunk, f) != (size_t)chunk) rc = QSIM_ERR_OPEN;
    }
    if (f) fclose(f);
    free(buf);
    return rc;
}

double _Complex *compute_state_vector(char *filename, int *num_q) {
    qsim_circuit *c = NULL;
    qsim_state *s = NULL;
    (void)qsim_device_init(env_int("QSIM_DEVICE", 0)); /* context creation is process start-up, outside the clock */
    const double t_start = wall_seconds();
    int rc = qsim_circuit_parse_file(filename, &c);
    if (rc == QSIM_ERR_OPEN) {
        printf("ERROR: cannot open circuit file\n"); /* quantum_simulator.c:128-131 */
        exit(1);
    }
    if (rc != QSIM_OK) {
        qsim_print_format_help(qsim_circuit_error());
        return NULL;
    }
    const int n = qsim_circuit_num_qubits(c);
    rc = qsim_create(&s, n, env_int("QSIM_DEVICE", 0));
    if (rc == QSIM_OK) rc = qsim_set_option(s, QSIM_OPT_PINGPONG, 0); /* one circuit, one run: a second buffer costs more to allocate than it saves (main.c) */
    if (rc == QSIM_OK) rc = qsim_apply_env_options(s);
    if (rc == QSIM_OK) rc = qsim_run_circuit(s, c, 0, -1);
    if (rc == QSIM_OK) rc = qsim_sync(s);
    qsim_circuit_free(c);
    if (rc != QSIM_OK) {
        printf(rc == QSIM_ERR_ALLOC ? "Malloc error\n" : "ERROR: %s\n", qsim_last_error());
        qsim_destroy(s);
        return NULL;
    }
    printf("%lf\n", wall_seconds() - t_start); /* quantum_simulator.c:244-248 */

    double _Complex *v = (double _Complex *)malloc(sizeof(double _Complex) << n);
    if (!v) {
        printf("Malloc error\n");
        qsim_destroy(s);
        return NULL;
    }
    rc = qsim_read(s, 0, 1ULL << n, (double *)v);
    const char *dump = getenv("QSIM_DUMP");
    if (rc == QSIM_OK && dump && *dump) {
        FILE *f = fopen(dump, "wb");
        if (f) { fwrite(v, 16, (size_t)1 << n, f); fclose(f); }
    }
    qsim_destroy(s);
    if (rc != QSIM_OK) { free(v); return NULL; }
    if (num_q) *num_q = n;
    return v;
}

/* One gate on a host vector: upload, apply, download.  Correct, and as slow as that sounds — programs
 * that apply many gates should hold a qsim_state instead. */
static int on_device(double _Complex *v, int num_q, int is_cx, const double *U, int a, int b) {
    qsim_state *s = NULL;
    int rc = qsim_create(&s, num_q, env_int("QSIM_DEVICE", 0));
    if (rc == QSIM_OK) rc = qsim_write(s, 0, 1ULL << num_q, (const double *)v);
    if (rc == QSIM_OK) rc = is_cx ? qsim_apply_cx(s, a, b) : qsim_apply_1q(s, U, a);
    if (rc == QSIM_OK) rc = qsim_read(s, 0, 1ULL << num_q, (double *)v);
    qsim_destroy(s);
    if (rc != QSIM_OK) fprintf(stderr, "qsim: %s\n", qsim_last_error());
    return rc;
}

void execute_single_qubit_gate(double _Complex *v, int num_q, double _Complex U[4], int target) {
    /* the reference multiplies by the transpose (quantum_simulator.c:88-89): hand the engine U^T */
    const double Ut[8] = {creal(U[0]), cimag(U[0]), creal(U[2]), cimag(U[2]),
                          creal(U[1]), cimag(U[1]), creal(U[3]), cimag(U[3])};
    (void)on_device(v, num_q, 0, Ut, target, -1);
}

void execute_cnot(double _Complex *v, int num_q, int control, int target) {
    (void)on_device(v, num_q, 1, NULL, control, target);
}
