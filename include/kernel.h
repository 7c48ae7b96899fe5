/*
 * kernel.h -- what the reference's driver (solver_test.c:1) needs from the reference's
 * kernel.h, and nothing else, so that solver_test.c compiles UNCHANGED against this
 * include directory and links against libehyb.so.
 *
 * The reference header (kernel.h:4-28) pulls in libc, the CUDA/cuBLAS/cuSPARSE headers
 * and the sizing constants of its 82-SM target; the driver uses
 *   - libc/libm through it (malloc, printf, ceil, fabs, fmin, srand/rand, getopt),
 *   - smSize, smSize2, threadELL, maxSharedMem in its sizing heuristic
 *     (solver_test.c:53-77, 158-182).
 * The CUDA includes, gpuErrchk and the kernel prototypes (kernel.h:14-18, 30-59) belong
 * to the reference's device half, which this library replaces: they are not declared here.
 *
 * The constants keep the reference's values on purpose.  They only shape the nParts /
 * vectorCacheSize HINT the driver writes into matrixCOO; matrixReorder[_unsym] of this
 * library re-derive the partition count for 256 CUs x 160 KiB of LDS (ehyb_sizing) and
 * write the values they used back into the struct, and spmvGPuEHYB sizes its windows
 * from partBoundary.  (With MI355X numbers in the reference's formula the 16-bit
 * vectorCacheSize of solver_test.c:55,160 would overflow.)
 */
#ifndef KERNEL_H
#define KERNEL_H

#include <stdlib.h>
#include <stdbool.h>
#include <stdio.h>
#include <stdint.h>
#include <string.h>
#include <math.h>
#include <sys/time.h>
#include <time.h>
#include <unistd.h>

#include "spmv.h"

/* reference kernel.h:20-23 */
#define warpSize 32
#define smSize 82
#define smSize2 80
#define maxSharedMem 93*1024

/* reference kernel.h:25-28 */
enum {
    threadELL = 1024,
    threadLongVec = 512,
    warpPerBlock = 1024 / 32,
    elementSize = 8 /* double precision */
};

#endif /* KERNEL_H */
