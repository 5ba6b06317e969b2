/*
 * oracle/cpu_ref.c — CPU restatement of the reference's hot path.  TEST INFRASTRUCTURE ONLY.
 *
 * This file is the parity checker for the HIP path.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load it; the product library (libqsim.so) never links, loads or
 * calls anything in oracle/.  It is single-threaded on purpose: the reference is single-threaded.
 *
 * What it restates (all citations are into /root/reference/quantum_simulator.c):
 *   oracle_apply_1q      <- execute_single_qubit_gate  :81-92   (full 2^n scan, `i < i^mask` predicate,
 *                                                                 applies U^T: v0' = v0*U0 + v1*U2)
 *   oracle_apply_cx      <- execute_cnot               :94-106  (full scan, control bit set, swap)
 *   oracle_gate_matrix   <- gate table                 :184-211 (PI = 2*asin(1), cexp for phases)
 *   oracle_run_qasm      <- compute_state_vector       :115-254 (two-statement header skip, char-level
 *                                                                 tokenizer, `qubit` allocation + |0..0>)
 *   oracle_cumulative    <- compute_state_cumulative_distribution :256-268
 *   oracle_measure / oracle_draw_randn <- measurement  :270-283
 *   oracle_putb          <- putb                       :285-293
 *
 * Parity pinning: tests/test_oracle_golden.py checks this file bit-for-bit against tests/golden/*.npy,
 * which were produced by the real reference compiled from /root/reference (oracle/Makefile target
 * `_ref`, script tests/golden/make_golden.py).  When oracle/_ref exists the same test also compares
 * the two live.
 *
 * Arithmetic note: the reference multiplies C99 `double complex` values; gcc lowers that to
 * (ac-bd, ad+bc) in plain doubles (plus a NaN-recovery branch that never triggers on finite data) and
 * adds component-wise.  This file spells the same operations on re/im pairs, compiled with
 * -ffp-contract=off, so results are bit-identical to the reference on finite inputs.
 *
 * Deliberate deviations (reference behaviour is undefined there): a qubit index outside [0,n), a gate
 * before the `qubit` statement, or n outside [0,40] is reported as an error instead of a wild access.
 */
#include <ctype.h>
#include <math.h>
#include <complex.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#define ORACLE_TOKEN_MAX 63 /* GATE_MAX_LEN, quantum_simulator.c:10 */

typedef struct { double re, im; } oc64;

static inline oc64 oc_mul(oc64 a, oc64 b) {
    oc64 r;
    r.re = a.re * b.re - a.im * b.im;
    r.im = a.re * b.im + a.im * b.re;
    return r;
}
static inline oc64 oc_add(oc64 a, oc64 b) {
    oc64 r; r.re = a.re + b.re; r.im = a.im + b.im; return r;
}

static double now_seconds(void) { /* get_time, :109-113 */
    struct timeval tv;
    gettimeofday(&tv, NULL);
    return (double)tv.tv_sec + (double)tv.tv_usec * 1e-6;
}

/* ---- hot loops ------------------------------------------------------------------------------- */

/* execute_single_qubit_gate (:81-92).  U is 4 complex numbers (8 doubles, re/im interleaved).
 * NOTE the transpose: the reference forms v0' = v0*U[0] + v1*U[2], v1' = v0*U[1] + v1*U[3]. */
void oracle_apply_1q(double *state, int num_q, const double *U, int target) {
    oc64 *v = (oc64 *)state;
    const oc64 *u = (const oc64 *)U;
    const uint64_t dim = 1ULL << num_q, bit = 1ULL << target;
    for (uint64_t i = 0; i < dim; i++) {
        const uint64_t j = i ^ bit;
        if (i < j) {
            const oc64 a = v[i], b = v[j];
            v[i] = oc_add(oc_mul(a, u[0]), oc_mul(b, u[2]));
            v[j] = oc_add(oc_mul(a, u[1]), oc_mul(b, u[3]));
        }
    }
}

/* execute_cnot (:94-106) */
void oracle_apply_cx(double *state, int num_q, int control, int target) {
    oc64 *v = (oc64 *)state;
    const uint64_t dim = 1ULL << num_q, cbit = 1ULL << control, tbit = 1ULL << target;
    for (uint64_t i = 0; i < dim; i++) {
        const uint64_t j = i ^ tbit;
        if (i < j && (i & cbit)) {
            const oc64 a = v[i];
            v[i] = v[j];
            v[j] = a;
        }
    }
}

/* ---- gate table (:184-211) ------------------------------------------------------------------- */

enum { OG_UNKNOWN = 0, OG_QUBIT, OG_CX, OG_U1 };

/* Classifies a token; for single-qubit gates fills U (8 doubles).  Matching rules follow the
 * reference: exact strcmp for every name except rz, which is recognised by its first two characters
 * and takes its angle from the text after "rz(" (:205-206). */
int oracle_gate_matrix(const char *tok, double *U) {
    const double pi = 2 * asin(1); /* :9 */
    double complex m[4];
    int diag_phase = 0;
    double complex ph = 0;
    if (!strcmp(tok, "qubit")) return OG_QUBIT;
    if (!strcmp(tok, "cx")) return OG_CX;
    if (!strcmp(tok, "x")) {
        m[0] = 0.0; m[1] = 1.0; m[2] = 1.0; m[3] = 0.0;
    } else if (!strcmp(tok, "sx")) {
        m[0] = (1.0 + I) / 2.0; m[1] = (1.0 - I) / 2.0;
        m[2] = (1.0 - I) / 2.0; m[3] = (1.0 + I) / 2.0;
    } else if (!strcmp(tok, "z")) {
        m[0] = 1.0; m[1] = 0.0; m[2] = 0.0; m[3] = -1.0;
    } else if (!strcmp(tok, "s")) {
        diag_phase = 1; ph = cexp(I * pi / 2.0);
    } else if (!strcmp(tok, "sdg")) {
        diag_phase = 1; ph = cexp(-I * pi / 2.0);
    } else if (!strcmp(tok, "t")) {
        diag_phase = 1; ph = cexp(I * pi / 4.0);
    } else if (!strcmp(tok, "tdg")) {
        diag_phase = 1; ph = cexp(-I * pi / 4.0);
    } else if (tok[0] == 'r' && tok[1] == 'z') {
        double theta = 0.0; /* the reference leaves this uninitialised when sscanf fails */
        if (strlen(tok) > 3) sscanf(tok + 3, "%lf", &theta);
        diag_phase = 1; ph = cexp(I * theta);
    } else if (!strcmp(tok, "h")) {
        m[0] = 1.0 / sqrt(2.0); m[1] = 1.0 / sqrt(2.0);
        m[2] = 1.0 / sqrt(2.0); m[3] = -1.0 / sqrt(2.0);
    } else {
        return OG_UNKNOWN;
    }
    if (diag_phase) { m[0] = 1.0; m[1] = 0.0; m[2] = 0.0; m[3] = ph; }
    for (int k = 0; k < 4; k++) { U[2 * k] = creal(m[k]); U[2 * k + 1] = cimag(m[k]); }
    return OG_U1;
}

/* ---- tokenizer (:133-242) ---------------------------------------------------------------------
 * The reference reads one character at a time with fscanf("%c"), which leaves the variable
 * untouched at end of file and raises feof.  `rd_next` reproduces exactly that on a memory buffer. */
typedef struct { const char *p; size_t len, pos; int eof; char c; } reader;

static void rd_next(reader *r) {
    if (r->pos < r->len) r->c = r->p[r->pos++];
    else r->eof = 1;
}
/* fscanf("%d"): skip white space, optional sign, digits; on failure the target keeps its value. */
static void rd_int(reader *r, int *out) {
    size_t q = r->pos;
    while (q < r->len && isspace((unsigned char)r->p[q])) q++;
    size_t s = q;
    if (q < r->len && (r->p[q] == '+' || r->p[q] == '-')) q++;
    size_t d = q;
    long val = 0;
    while (q < r->len && isdigit((unsigned char)r->p[q])) { val = val * 10 + (r->p[q] - '0'); q++; }
    if (q == d) { if (q >= r->len) r->eof = 1; return; }
    if (r->p[s] == '-') val = -val;
    *out = (int)val;
    r->pos = q;
}
/* the separator class used between statements (:136,:147; `]` is added after a gate, :241) */
static int is_sep(char c, int also_bracket) {
    unsigned char u = (unsigned char)c;
    return isblank(u) || c == 10 || c == ',' || c == ';' || !isgraph(u) || (also_bracket && c == ']');
}
static void skip_to_operand(reader *r) { /* :163-164, :225-226 */
    while (r->c != '$' && r->c != '[' && !r->eof) rd_next(r);
}

static char *slurp(const char *path, size_t *len) {
    FILE *f = fopen(path, "rb");
    if (!f) return NULL;
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)sz + 1);
    if (!buf) { fclose(f); return NULL; }
    size_t got = fread(buf, 1, (size_t)sz, f);
    fclose(f);
    buf[got] = 0;
    *len = got;
    return buf;
}

/* compute_state_vector (:115-254).
 *   max_gates < 0 : run the whole file; otherwise stop after that many gate statements (bench.py's
 *                   bounded cpu_baseline sample).
 *   *seconds      : elapsed time measured like the reference (after the header, through the last gate).
 *   *gates_done   : number of gate statements applied.
 * Returns a malloc'd array of 2^n (re,im) pairs, or NULL with *err set:
 *   1 cannot open, 2 allocation failure, 3 unknown token, 4 operand out of range / gate before qubit. */
double *oracle_run_qasm(const char *path, int *num_q, double *seconds, long max_gates, long *gates_done,
                        int *err) {
    size_t len = 0;
    char *text = slurp(path, &len);
    int e = 0;
    long done = 0;
    oc64 *v = NULL;
    int n = 0;
    if (err) *err = 0;
    if (!text) { if (err) *err = 1; return NULL; }

    reader r = { text, len, 0, 0, 0 };
    /* header: two statements, each up to and including ';', each followed by separators (:133-141) */
    for (int h = 0; h < 2; h++) {
        do rd_next(&r); while (r.c != ';' && !r.eof);
        do rd_next(&r); while (is_sep(r.c, 0) && !r.eof);
    }
    const double t0 = now_seconds(); /* :143 */

    while (!r.eof) {
        char tok[ORACLE_TOKEN_MAX + 1];
        int tl = 0;
        while (is_sep(r.c, 0) && !r.eof) rd_next(&r);
        tok[tl++] = r.c; tok[tl] = 0;
        rd_next(&r);
        while (isgraph((unsigned char)r.c) && r.c != '[' && tl < ORACLE_TOKEN_MAX) {
            tok[tl++] = r.c; tok[tl] = 0;
            rd_next(&r);
        }

        double U[8];
        const int kind = oracle_gate_matrix(tok, U);
        if (kind == OG_QUBIT) { /* :162-181 */
            skip_to_operand(&r);
            rd_int(&r, &n);
            if (n < 0 || n > 40) { e = 4; break; }
            free(v);
            v = (oc64 *)malloc(sizeof(oc64) << n);
            if (!v) { e = 2; break; }
            for (uint64_t i = 1; i < (1ULL << n); i++) { v[i].re = 0; v[i].im = 0; }
            v[0].re = 1.0; v[0].im = 0.0;
            while (r.c != '\n' && !r.eof) rd_next(&r);
            continue;
        }
        if (kind == OG_UNKNOWN) { e = 3; break; } /* :212-223 */

        int qa = -1, qb = -1;
        skip_to_operand(&r);
        rd_int(&r, &qa);
        if (kind == OG_CX) { /* :229-233 */
            rd_next(&r);
            skip_to_operand(&r);
            rd_int(&r, &qb);
        }
        if (!v || qa < 0 || qa >= n || (kind == OG_CX && (qb < 0 || qb >= n))) { e = 4; break; }
        if (kind == OG_CX) oracle_apply_cx((double *)v, n, qa, qb);
        else oracle_apply_1q((double *)v, n, U, qa);
        done++;
        if (max_gates >= 0 && done >= max_gates) break;

        rd_next(&r); /* :240-242 */
        while (is_sep(r.c, 1) && !r.eof) rd_next(&r);
    }
    const double t1 = now_seconds(); /* :244 */
    free(text);
    if (seconds) *seconds = t1 - t0;
    if (gates_done) *gates_done = done;
    if (num_q) *num_q = n;
    if (e) { free(v); if (err) *err = e; return NULL; }
    return (double *)v;
}

void oracle_free(double *p) { free(p); }

/* ---- measurement post-path (:256-293; the call site in main, :67-73, is commented out in the reference) ---- */

/* compute_state_cumulative_distribution (:256-268): cumul[i] = sum_{k<=i} cabs(v[k])*cabs(v[k]), accumulated in
 * index order.  (The reference allocates twice the bytes it needs, B10; harmless, not reproduced.) */
double *oracle_cumulative(const double *state, int num_q) {
    const uint64_t dim = 1ULL << num_q;
    double *res = (double *)malloc(sizeof(double) * dim);
    if (!res) return NULL;
    double acc = 0.0;
    for (uint64_t i = 0; i < dim; i++) {
        const double a = cabs(state[2 * i] + I * state[2 * i + 1]);
        acc += a * a;
        res[i] = acc;
    }
    return res;
}

/* The search of measurement (:278-282) for a given random number: the first index whose cumulative value is
 * non-zero and not below randn, or 2^n - 1.  (The reference draws randn itself from rand(), :271-276.) */
long long oracle_measure(const double *cumul, int num_q, double randn) {
    const long long last = (long long)((1ULL << num_q) - 1);
    long long idx = 0;
    while ((cumul[idx] == 0.0 || cumul[idx] < randn) && idx < last) idx++;
    return idx;
}

/* measurement's random number (:271-276): ten rand() draws, each scaled by a further 1/RAND_MAX. */
double oracle_draw_randn(void) {
    double randn = 0.0, coeff = 1.0 / RAND_MAX;
    for (int i = 0; i < 10; i++) {
        randn += rand() * coeff;
        coeff *= 1.0 / RAND_MAX;
    }
    return randn;
}

/* putb (:285-293): n as `len` binary digits, most significant first, into buf (len + 1 bytes). */
void oracle_putb(long long n, int len, char *buf) {
    for (int k = 0; k < len; k++) buf[k] = ((n >> (len - 1 - k)) & 1) ? '1' : '0';
    buf[len] = 0;
}

/* Fresh |0...0> of n qubits (:168-177), for kernel-level tests that do not go through a file. */
double *oracle_alloc_zero_state(int num_q) {
    oc64 *v = (oc64 *)calloc((size_t)1 << num_q, sizeof(oc64));
    if (v) v[0].re = 1.0;
    return (double *)v;
}

#ifdef ORACLE_MAIN
/* CLI twin of main (:32-79): `cpu_ref <file> <shots>` prints the elapsed seconds.  With a third
 * argument it also writes the amplitudes as raw little-endian doubles to that path. */
int main(int argc, char **argv) {
    if (argc < 3) {
        printf("QUANTUM CIRCUIT SIMULATOR\n");
        printf("Usage: %s <circuit_file_name> <number_of_measurement>\n", argv[0]);
        return 1;
    }
    int n = 0, err = 0;
    double secs = 0;
    long done = 0;
    double *v = oracle_run_qasm(argv[1], &n, &secs, -1, &done, &err);
    if (err == 1) { printf("ERROR: cannot open circuit file\n"); return 1; }
    if (!v) { printf("ERROR while parsing quantum circuit\n"); return 1; }
    printf("%lf\n", secs);
    if (argc > 3) {
        FILE *o = fopen(argv[3], "wb");
        if (o) { fwrite(v, 16, (size_t)1 << n, o); fclose(o); }
    }
    free(v);
    return 0;
}
#endif
