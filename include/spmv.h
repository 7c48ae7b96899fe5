/*
 * spmv.h -- drop-in boundary header of the MI355X-native Explicit-Caching HYB SpMV.
 *
 * This header is the *contract* a caller of the reference project compiles against
 * (reference spmv.h:7-33 data types, spmv.h:75-78 entry point).  Only the pieces a
 * caller touches are kept, byte-for-byte layout compatible:
 *
 *   cb_s        run switches                        (reference spmv.h:7-15, init: 65-73)
 *   matrixCOO   permuted, row-grouped COO/CSR view  (reference spmv.h:17-33)
 *   spmvGPuEHYB the one C-linkage symbol            (reference spmv.h:75-78, def spmv.cu:61-64)
 *
 * The reference's matrixEHYB (spmv.h:35-63) is an implementation detail of its CUDA
 * kernels (32-row slabs, int16 widths, global work-queue heads); it is deliberately
 * NOT part of this boundary.  The MI355X layout lives behind the opaque ehyb_plan
 * of ehyb.h.
 *
 * Usable from C and C++ (the reference header is C++-only: default member
 * initialisers and an unguarded extern "C").
 */
#ifndef SPMV_H
#define SPMV_H

#include <stdint.h>
#ifndef __cplusplus
#include <stdbool.h>
#endif

/* Run switches.  Same members, same order as reference spmv.h:7-15.  The harness
 * refuses to run unless RODR, CACHE and BLOCK are all set (solver_test.c:322-325). */
typedef struct _cb {
    bool PRECOND;
    bool GPU;
    bool RODR;
    bool CACHE;
    bool BLOCK;
    bool FACT;
    bool SORT;
} cb_s;

/*
 * Matrix handed across the boundary.  Field order and widths follow reference
 * spmv.h:17-33 exactly: 4 x int32, uint16, int16, then nine pointers.
 *
 *  totalNum        number of stored entries of the expanded matrix (explicit zeros count)
 *  dimension       rows == columns
 *  maxCol          longest row (entries)
 *  nParts          partitions; partBoundary has nParts+1 entries
 *  vectorCacheSize x-window length (rows) the caller sized its partitions for.  16 bit
 *                  in the reference (overflows past ~2.6 M rows);
 *                  the callee here treats it as a hint and re-derives its own window
 *                  from partBoundary, so large matrices stay representable.
 *  kernelPerPart   workgroups co-operating on one partition in the reference's
 *                  "_small" kernel (kernel.cu:197-284).  Hint only.
 *  rowIdx[n+1]     CSR row pointer into I/J/V (entries of a row are contiguous)
 *  numInRow[n]     entries per row
 *  numInRow2[n]    entries per row whose column falls inside the row's partition window
 *  I,J,V           row, column, value (0-based, permuted numbering)
 *  diag            diagonal (symmetric reader only; unused by the callee)
 *  partBoundary    first permuted row of each partition.  matrixReorder / matrixReorder_unsym ALLOCATE it
 *                  (malloc, dimension + 1 ints) over whatever the field held, as reordering.c:44,234 do, and
 *                  may write back a larger nParts than they were handed (sized for 256 CUs);
 *                  ehyb_matrix_reorder writes into the caller's array (see ehyb_config.part_boundary_cap)
 *  reorderList     reorderList[old] = new
 */
typedef struct _matrixCOO {
    int      totalNum;
    int      dimension;
    int      maxCol;
    int      nParts;
    uint16_t vectorCacheSize;
    int16_t  kernelPerPart;
    int*     rowIdx;
    int*     numInRow;
    int*     numInRow2;
    int*     I;
    int*     J;
    double*  V;
    double*  diag;
    int*     partBoundary;
    int*     reorderList;
} matrixCOO;

/* Same defaults as reference spmv.h:65-73 (SORT is left untouched there too). */
static inline void init_cb(cb_s* s)
{
    s->PRECOND = false;
    s->GPU     = false;
    s->RODR    = true;
    s->BLOCK   = true;
    s->CACHE   = true;
    s->FACT    = true;
}

#ifdef __cplusplus
extern "C" {
#endif

/*
 * vectorOut = A_perm * vectorIn in fp64, executed 10 (warm-up) + MAXIter times on the
 * GPU with a constant x; the result of the last multiply is returned.
 * Replaces reference spmv.cu:61-133.  Pre-conditions (established by the caller,
 * solver_test.c:350-376): localMatrix is permuted and grouped by row, partBoundary is
 * filled, vectorIn is already permuted.  Post: caller un-permutes (solver_test.c:383).
 *
 * Differences from the reference, all deliberate:
 *   - the residual ("ER") part is recomputed on every multiply;
 *   - the timed window contains kernels only (no H2D/D2H);
 *   - *realIter is set to MAXIter (the reference never assigns it);
 *   - HIP errors abort with a message instead of being ignored.
 * Prints "sizeER is %d" and "iter is %d, time is %f ms, GPU Gflops is %f" like
 * spmv.cu:82,121 (GFLOP/s = 2*totalNum*iter/time).
 */
void spmvGPuEHYB(matrixCOO* localMatrix,
                 const double* vectorIn, double* vectorOut,
                 const int MAXIter, int* realIter);

/* Additive variant with a status code (0 = ok, non-zero = ehyb_status of ehyb.h; text via ehyb_last_error).
 * PREFER THIS ONE in a solver: spmvGPuEHYB keeps the reference's void signature and therefore can only
 * exit() the caller's process when something fails (as the reference's converter does, convert.c:122-125). */
int spmvGPuEHYB_status(matrixCOO* localMatrix,
                       const double* vectorIn, double* vectorOut,
                       const int MAXIter, int* realIter);

#ifdef __cplusplus
}
#endif

#endif /* SPMV_H */
