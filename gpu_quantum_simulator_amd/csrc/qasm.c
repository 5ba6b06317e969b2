/*
 * qasm.c — circuit front-end of the C host: the OPENQASM-3 subset that compute_state_vector
 * (quantum_simulator.c:115-254) tokenizes, turned into a gate list for the engine.
 *
 * Grammar accepted:
 *   - two header statements, each ended by ';' (content ignored)                    :133-141
 *   - `qubit[n] q;` or `qubit q[n];`  -> register size, rest of the line ignored    :162-181
 *   - gate token = graph characters up to blank or '['; names cx x sx z s sdg t tdg h matched exactly,
 *     rz by its first two characters with the angle read after "rz("              :150-159, :182-211
 *   - operands: scan to '[' or '$', read an integer; cx reads two                  :225-233
 *   - separators between statements: blank, newline, ',', ';', ']', any non-graph  :136, :147, :241
 *   - unknown token -> error QSIM_ERR_PARSE with the reference's message           :212-223
 * Unlike the reference (which executes while it reads), the file is read whole and parsed from memory;
 * the observable result is the same because a parse error discards the state there too (:221-222).
 *
 * Also accepted: the CUDA variants' `<num_qubit> <num_gates>` form, detected by a leading digit
 * (quantum_simulator_naive.cu:239-240, gate lines :258-397).
 *
 * Two corners of that loop are reproduced on purpose (tests/test_scheduler_cpu.py pins both against the oracle):
 *   - a second `qubit` statement allocates a fresh |0...0> of the new size (:162-181), so every gate read before it
 *     has no effect: the gate list is emptied;
 *   - a file whose LAST statement is `qubit ...` followed by at least one more character (the usual final newline)
 *     fails with "Unknown token: <that character>": the skip to the end of the line (:179-180) stops at '\n' without
 *     raising end-of-file, so the statement loop runs once more and takes the blank as a gate name (:147-151).
 *     Without any character after the line the same file is accepted and yields |0...0>.
 *
 * Deliberate deviations — where the reference has undefined behaviour this front-end reports an error instead:
 * operand outside [0, n), gate before the `qubit` statement, `rz` without a readable angle, and a file without any
 * `qubit` statement (the reference prints its time line and returns the NULL vector, :125,:248-252).
 */
#include <ctype.h>
#include <math.h>
#include <complex.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "circuit.h"

#define TOKEN_MAX 63 /* GATE_MAX_LEN, quantum_simulator.c:10 */

static _Thread_local char g_circ_err[256];

void qsim_set_circuit_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_circ_err, sizeof g_circ_err, fmt, ap);
    va_end(ap);
}
const char *qsim_circuit_error(void) { return g_circ_err; }

/* ---- gate table (quantum_simulator.c:184-211), standard orientation ------------------------------ */
int qsim_gate_matrix(const char *tok, double *U) {
    const double pi = 2 * asin(1); /* quantum_simulator.c:9 */
    double complex m00 = 1.0, m01 = 0.0, m10 = 0.0, m11 = 1.0;
    if (!tok) return 0;
    if (!strcmp(tok, "cx")) return QSIM_GATE_CX;
    if (!strcmp(tok, "x")) { m00 = 0.0; m01 = 1.0; m10 = 1.0; m11 = 0.0; }
    else if (!strcmp(tok, "sx")) {
        m00 = (1.0 + I) / 2.0; m01 = (1.0 - I) / 2.0;
        m10 = (1.0 - I) / 2.0; m11 = (1.0 + I) / 2.0;
    }
    else if (!strcmp(tok, "z")) m11 = -1.0;
    else if (!strcmp(tok, "s")) m11 = cexp(I * pi / 2.0);
    else if (!strcmp(tok, "sdg")) m11 = cexp(-I * pi / 2.0);
    else if (!strcmp(tok, "t")) m11 = cexp(I * pi / 4.0);
    else if (!strcmp(tok, "tdg")) m11 = cexp(-I * pi / 4.0);
    else if (tok[0] == 'r' && tok[1] == 'z') {
        double theta;
        if (strlen(tok) < 4 || sscanf(tok + 3, "%lf", &theta) != 1) return 0;
        m11 = cexp(I * theta);
    }
    else if (!strcmp(tok, "h")) {
        m00 = 1.0 / sqrt(2.0); m01 = 1.0 / sqrt(2.0);
        m10 = 1.0 / sqrt(2.0); m11 = -1.0 / sqrt(2.0);
    }
    else return 0;
    /* every matrix above is symmetric, so the reference's transposed product gives the same state (S7) */
    U[0] = creal(m00); U[1] = cimag(m00); U[2] = creal(m01); U[3] = cimag(m01);
    U[4] = creal(m10); U[5] = cimag(m10); U[6] = creal(m11); U[7] = cimag(m11);
    return QSIM_GATE_U1;
}

/* ---- circuit container ----------------------------------------------------------------------------- */
int qsim_circuit_create(int num_q, qsim_circuit **out) {
    if (!out || num_q < 0 || num_q > 40) { qsim_set_circuit_error("bad qubit count %d", num_q); return QSIM_ERR_ARG; }
    qsim_circuit *c = (qsim_circuit *)calloc(1, sizeof *c);
    if (!c) return QSIM_ERR_ALLOC;
    c->num_q = num_q;
    *out = c;
    return QSIM_OK;
}

void qsim_circuit_free(qsim_circuit *c) {
    if (!c) return;
    free(c->gates);
    free(c->mats2);
    free(c->mats4);
    free(c);
}

static int grow(void **p, long *cap, long need, size_t elem) {
    if (need <= *cap) return 1;
    long nc = *cap ? *cap * 2 : 256;
    while (nc < need) nc *= 2;
    void *np = realloc(*p, (size_t)nc * elem);
    if (!np) return 0;
    *p = np;
    *cap = nc;
    return 1;
}

static int push_gate(qsim_circuit *c, int kind, int q0, int q1, int mat) {
    if (!grow((void **)&c->gates, &c->cap, c->count + 1, sizeof(qsim_gate_rec))) return QSIM_ERR_ALLOC;
    qsim_gate_rec *g = &c->gates[c->count++];
    g->kind = kind; g->q0 = q0; g->q1 = q1; g->mat = mat;
    return QSIM_OK;
}

int qsim_circuit_append_1q(qsim_circuit *c, const double *U, int target) {
    if (!c || !U || target < 0 || target >= c->num_q) { qsim_set_circuit_error("qubit %d out of range", target); return QSIM_ERR_ARG; }
    /* consecutive identical matrices (a run of `h`, say) share one slot */
    long idx = c->n2 - 1;
    if (idx < 0 || memcmp(c->mats2 + 8 * idx, U, 8 * sizeof(double)) != 0) {
        if (!grow((void **)&c->mats2, &c->cap2, c->n2 + 1, 8 * sizeof(double))) return QSIM_ERR_ALLOC;
        idx = c->n2++;
        memcpy(c->mats2 + 8 * idx, U, 8 * sizeof(double));
    }
    return push_gate(c, QSIM_GATE_U1, target, -1, (int)idx);
}

int qsim_circuit_append_cx(qsim_circuit *c, int control, int target) {
    if (!c || control < 0 || control >= c->num_q || target < 0 || target >= c->num_q) {
        qsim_set_circuit_error("cx operand out of range (%d, %d)", control, target);
        return QSIM_ERR_ARG;
    }
    return push_gate(c, QSIM_GATE_CX, control, target, -1);
}

int qsim_circuit_append_2q(qsim_circuit *c, const double *U, int q_hi, int q_lo) {
    if (!c || !U || q_lo < 0 || q_hi >= c->num_q || q_lo >= q_hi) { qsim_set_circuit_error("bad 2q operands (%d, %d)", q_hi, q_lo); return QSIM_ERR_ARG; }
    if (!grow((void **)&c->mats4, &c->cap4, c->n4 + 1, 32 * sizeof(double))) return QSIM_ERR_ALLOC;
    long idx = c->n4++;
    memcpy(c->mats4 + 32 * idx, U, 32 * sizeof(double));
    return push_gate(c, QSIM_GATE_U2, q_hi, q_lo, (int)idx);
}

int qsim_circuit_num_qubits(const qsim_circuit *c) { return c ? c->num_q : -1; }
long qsim_circuit_num_gates(const qsim_circuit *c) { return c ? c->count : -1; }

int qsim_circuit_gate(const qsim_circuit *c, long i, int *kind, int *q0, int *q1, double *U) {
    if (!c || i < 0 || i >= c->count) return QSIM_ERR_ARG;
    const qsim_gate_rec *g = &c->gates[i];
    if (kind) *kind = g->kind;
    if (q0) *q0 = g->q0;
    if (q1) *q1 = g->q1;
    if (U && g->kind == QSIM_GATE_U1) memcpy(U, c->mats2 + 8 * (long)g->mat, 8 * sizeof(double));
    if (U && g->kind == QSIM_GATE_U2) memcpy(U, c->mats4 + 32 * (long)g->mat, 32 * sizeof(double));
    return QSIM_OK;
}

/* ---- tokenizer ------------------------------------------------------------------------------------ */
typedef struct { const char *p; size_t len, pos; int eof; char c; } cursor;

/* fscanf("%c") semantics: at end of input the character is left as it was and eof is raised */
static void advance(cursor *s) {
    if (s->pos < s->len) s->c = s->p[s->pos++];
    else s->eof = 1;
}
static int separator(char c, int closing_bracket_too) {
    const unsigned char u = (unsigned char)c;
    if (closing_bracket_too && c == ']') return 1;
    return isblank(u) || c == '\n' || c == ',' || c == ';' || !isgraph(u);
}
/* fscanf("%d") semantics; returns 1 when an integer was read */
static int read_int(cursor *s, int *out) {
    size_t q = s->pos;
    while (q < s->len && isspace((unsigned char)s->p[q])) q++;
    int neg = 0;
    if (q < s->len && (s->p[q] == '+' || s->p[q] == '-')) { neg = s->p[q] == '-'; q++; }
    if (q >= s->len || !isdigit((unsigned char)s->p[q])) return 0;
    long v = 0;
    while (q < s->len && isdigit((unsigned char)s->p[q])) { if (v < 100000000L) v = v * 10 + (s->p[q] - '0'); q++; }
    *out = (int)(neg ? -v : v);
    s->pos = q;
    return 1;
}
static void seek_operand(cursor *s) {
    while (s->c != '$' && s->c != '[' && !s->eof) advance(s);
}

/* statement loop shared by both file forms; max_gates < 0 = until end of input */
static int parse_statements(cursor *s, qsim_circuit *c, int *have_reg, long max_gates) {
    int have_register = *have_reg;
    while (!s->eof && (max_gates < 0 || c->count < max_gates)) {
        char tok[TOKEN_MAX + 1];
        int tl = 0;
        while (separator(s->c, 0) && !s->eof) advance(s);
        if (s->eof && separator(s->c, 0)) {
            /* Only blanks were left.  The reference takes the last one as a gate name (:147-151); in the OPENQASM form
             * that can only happen right after a `qubit` line (after a gate the trailing blanks are consumed by
             * :240-242, which raises end-of-file before the loop test). */
            if (max_gates < 0) { qsim_set_circuit_error("Unknown token: %c", s->c); return QSIM_ERR_PARSE; }
            break;
        }
        tok[tl++] = s->c; tok[tl] = 0;
        advance(s);
        while (isgraph((unsigned char)s->c) && s->c != '[' && tl < TOKEN_MAX) {
            tok[tl++] = s->c; tok[tl] = 0;
            advance(s);
        }
        if (!strcmp(tok, "qubit")) { /* :162-181 */
            int n = -1;
            seek_operand(s);
            if (!read_int(s, &n) || n < 0 || n > 40) { qsim_set_circuit_error("bad register size in qubit statement"); return QSIM_ERR_PARSE; }
            c->num_q = n;
            have_register = *have_reg = 1;
            c->count = 0; c->n2 = 0; c->n4 = 0; /* a fresh |0...0>: gates read so far are void (:168-177) */
            while (s->c != '\n' && !s->eof) advance(s);
            continue;
        }
        double U[8];
        const int kind = qsim_gate_matrix(tok, U);
        if (!kind) { qsim_set_circuit_error("Unknown token: %s", tok); return QSIM_ERR_PARSE; }
        if (!have_register) { qsim_set_circuit_error("gate '%s' before the qubit statement", tok); return QSIM_ERR_PARSE; }
        int a = -1, b = -1;
        seek_operand(s);
        if (!read_int(s, &a)) { qsim_set_circuit_error("missing operand after '%s'", tok); return QSIM_ERR_PARSE; }
        if (kind == QSIM_GATE_CX) { /* :229-233 */
            advance(s);
            seek_operand(s);
            if (!read_int(s, &b)) { qsim_set_circuit_error("missing second operand of cx"); return QSIM_ERR_PARSE; }
        }
        if (a < 0 || a >= c->num_q || (kind == QSIM_GATE_CX && (b < 0 || b >= c->num_q))) {
            qsim_set_circuit_error("operand out of range in '%s' (register has %d qubits)", tok, c->num_q);
            return QSIM_ERR_PARSE;
        }
        const int rc = kind == QSIM_GATE_CX ? qsim_circuit_append_cx(c, a, b) : qsim_circuit_append_1q(c, U, a);
        if (rc) return rc;
        advance(s); /* :240-242 */
        while (separator(s->c, 1) && !s->eof) advance(s);
    }
    return QSIM_OK;
}

static int parse_openqasm(cursor *s, qsim_circuit *c) {
    for (int h = 0; h < 2; h++) { /* header: :133-141 */
        do advance(s); while (s->c != ';' && !s->eof);
        do advance(s); while (separator(s->c, 0) && !s->eof);
    }
    int have_register = 0;
    const int rc = parse_statements(s, c, &have_register, -1);
    if (rc) return rc;
    if (!have_register) { qsim_set_circuit_error("no qubit statement in the circuit file"); return QSIM_ERR_PARSE; }
    return QSIM_OK;
}

/* `<num_qubit> <num_gates>` then the same gate statements (`h q[0];`, `cx q[0], q[1];`, `$k` operands),
 * at most num_gates of them (quantum_simulator_naive.cu:239-240, loop :258-397). */
static int parse_counted(cursor *s, qsim_circuit *c) {
    int n = -1, ng = -1;
    s->pos = 0;
    if (!read_int(s, &n) || !read_int(s, &ng) || n < 0 || n > 40 || ng < 0) {
        qsim_set_circuit_error("bad '<num_qubit> <num_gates>' header");
        return QSIM_ERR_PARSE;
    }
    c->num_q = n;
    advance(s);
    while (separator(s->c, 0) && !s->eof) advance(s);
    if (s->eof || ng == 0) return QSIM_OK;
    int have_register = 1;
    return parse_statements(s, c, &have_register, ng);
}

int qsim_circuit_parse_text(const char *text, size_t len, qsim_circuit **out) {
    if (!text || !out) return QSIM_ERR_ARG;
    qsim_circuit *c = NULL;
    int rc = qsim_circuit_create(0, &c);
    if (rc) return rc;
    cursor s = { text, len, 0, 0, 0 };
    size_t k = 0;
    while (k < len && isspace((unsigned char)text[k])) k++;
    rc = (k < len && isdigit((unsigned char)text[k])) ? parse_counted(&s, c) : parse_openqasm(&s, c);
    if (rc) { qsim_circuit_free(c); return rc; }
    *out = c;
    return QSIM_OK;
}

int qsim_circuit_parse_file(const char *path, qsim_circuit **out) {
    if (!path || !out) return QSIM_ERR_ARG;
    FILE *f = fopen(path, "rb");
    if (!f) { qsim_set_circuit_error("ERROR: cannot open circuit file"); return QSIM_ERR_OPEN; }
    fseek(f, 0, SEEK_END);
    long sz = ftell(f);
    fseek(f, 0, SEEK_SET);
    char *buf = (char *)malloc((size_t)(sz > 0 ? sz : 0) + 1);
    if (!buf) { fclose(f); return QSIM_ERR_ALLOC; }
    size_t got = fread(buf, 1, (size_t)(sz > 0 ? sz : 0), f);
    fclose(f);
    int rc = qsim_circuit_parse_text(buf, got, out);
    free(buf);
    return rc;
}
