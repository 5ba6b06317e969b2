/* circuit.h — host-side gate list shared by the C tokenizer (qasm.c) and the C++ engine. */
#ifndef QSIM_CIRCUIT_H
#define QSIM_CIRCUIT_H

#include "../../include/qsim.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct {
    int kind;   /* QSIM_GATE_U1 / QSIM_GATE_CX / QSIM_GATE_U2 */
    int q0, q1; /* U1: target,-1   CX: control,target   U2: q_hi,q_lo */
    int mat;    /* U1: index into mats2 (8 doubles each); U2: index into mats4 (32 doubles each) */
} qsim_gate_rec;

struct qsim_circuit {
    int num_q;
    long count, cap;
    qsim_gate_rec *gates;
    long n2, cap2;
    double *mats2; /* 8 doubles per 2x2, row-major (re, im), standard U.v orientation */
    long n4, cap4;
    double *mats4; /* 32 doubles per 4x4 */
};

void qsim_set_circuit_error(const char *fmt, ...);

#ifdef __cplusplus
}
#endif
#endif
