/*
 * reordering.h -- source-compatible names of the reference's host pre-step.
 *
 * The reference compiles its .c files as C++ (Makefile:6,21-22), so a caller such as
 * solver_test.c:370-383 links against C++-mangled symbols.  libehyb.so exports the
 * same four names with the same signatures (reference reordering.h:6-10); they are
 * thin wrappers over the C-ABI of ehyb.h, which is what non-C++ callers bind.
 */
#ifndef REORDERING_H
#define REORDERING_H
#include "spmv.h"

#ifdef __cplusplus
/* P*A*P^T for a symmetric-pattern matrix (reordering.c:231-378).  Like the reference, both calls replace
 * m->partBoundary by a fresh malloc() (dimension + 1 ints here, dimension there: reordering.c:44,234) -- the
 * caller's pointer, NULL included, is never written through -- and exit(1) on failure. */
void matrixReorder(matrixCOO* localMatrixCOO);
/* Same after symmetrising the pattern (reordering.c:41-228). */
void matrixReorder_unsym(matrixCOO* localMatrixCOO);
/* v_rodr[rodr_list[i]] = v_in[i] (reordering.c:380-384). */
void vectorReorder(const int dimension, const double* v_in, double* v_rodr, const int* rodr_list);
/* v[i] = v_rodr[rodr_list[i]] (reordering.c:386-391). */
void vectorRecover(const int dimension, const double* v_rodr, double* v, const int* rodr_list);
#endif

#endif /* REORDERING_H */
