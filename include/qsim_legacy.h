/*
 * qsim_legacy.h — the three C functions quantum_simulator.c declares at :25-27, with the same names,
 * argument meaning and error behaviour, executed on the GPU through libqsim.so.  A program that was
 * written against quantum_simulator.c links against libqsim.so instead of compiling that file.
 *
 * C only (uses C99 `double _Complex`, the reference's `complex`).
 */
#ifndef QSIM_LEGACY_H
#define QSIM_LEGACY_H

#ifdef __cplusplus
#error "qsim_legacy.h mirrors a C99 interface (double _Complex); include qsim.h from C++"
#endif

/* quantum_simulator.c:115-254.  Parses `filename`, simulates it on GPU 0 (env QSIM_DEVICE), prints the
 * elapsed seconds as "%lf\n" on stdout, stores the qubit count in *num_q and returns a malloc'd array of
 * 2^n amplitudes the caller frees.  Unknown token: prints the reference's usage block and returns NULL.
 * Unreadable file: prints "ERROR: cannot open circuit file" and exit(1), as the reference does.
 * Env QSIM_DUMP=<path> additionally writes the amplitudes as raw little-endian doubles (re, im). */
double _Complex *compute_state_vector(char *filename, int *num_q);

/* quantum_simulator.c:81-92.  In place on caller-owned host memory.  Like the reference it applies the
 * TRANSPOSE of U: v[i] = v[i]*U[0] + v[i^m]*U[2], v[i^m] = v[i]*U[1] + v[i^m]*U[3]. */
void execute_single_qubit_gate(double _Complex *v, int num_q, double _Complex U[4], int target);

/* quantum_simulator.c:94-106.  In place on caller-owned host memory. */
void execute_cnot(double _Complex *v, int num_q, int control, int target);

#endif
