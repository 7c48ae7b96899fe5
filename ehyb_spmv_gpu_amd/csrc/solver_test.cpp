// solver_test -- command-line harness with the reference driver's contract
// (reference solver_test.c:267-408):  -m <name> reads ./read/<name>.mtx, -i <iters> is the
// number of timed multiplies; symmetric vs general follows the Matrix Market banner.
// Flow: read -> x (glibc rule) -> CPU reference y -> reorder -> P*x -> spmvGPuEHYB ->
// un-permute -> compare.
//
// The CPU product computed here is the harness's comparison baseline, exactly as in the
// reference (solver_test.c:102, 247, 254); it is never substituted for the GPU result.
// Additions: -g <generator spec> for synthetic input (no .mtx files exist offline), a strict
// per-row tolerance next to the reference's 1 % check, and a non-zero exit status on failure
// (the reference always returns 0).  y is calloc'ed (the reference mallocs it un-zeroed,
// solver_test.c:38,138) and the out-of-bounds debug print of rows 30000.. (385-388) is gone.
#include <getopt.h>
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <string>
#include <vector>

#include "ehyb.h"
#include "reordering.h"
#include "spmv.h"

// Restates compare() of solver_test.c:7-29: flags |y - yResult| > threshold*min(|y|,|yResult|),
// prints at most 100 offenders, then the two sums.
static int compare(const double* yResult, const double* y, double threshold, int dimension)
{
    double diff = 0, ampldiff = 0;
    int shown = 0, offenders = 0;
    for (int i = 0; i < dimension; ++i) {
        double d = fabs(y[i] - yResult[i]);
        double ampl = fmin(fabs(y[i]), fabs(yResult[i]));
        if (d > ampl * threshold) {
            ++offenders;
            if (shown < 100) {
                printf("large difference at %d  : realy %f vs yResult %f\n", i, y[i], yResult[i]);
                ++shown;
            }
        }
        diff += d;
        if (ampl > 0) ampldiff += d / ampl;
    }
    printf("diff is %e, ampldiff is %e\n", diff, ampldiff);
    return offenders;
}

static std::vector<long long> split_numbers(const char* s)
{
    std::vector<long long> out;
    while (s && *s) {
        out.push_back(atoll(s));
        s = strchr(s, ':');
        if (s) ++s;
    }
    return out;
}

static void usage()
{
    printf("usage: solver_test -i <iters> (-m <name> | -g <spec>) [-w 1|2] [-l lds_doubles] [-T threads] [-c plan.cache] [-S 0|1] [-X 0|1] [-I items_per_cu] [-v]\n"
           "  -X 0|1    work items in blockIdx order / one contiguous run of items per XCD (default)\n"
           "  -S 0|1    symmetric pair storage off / on (default: on for symmetric matrices of >= 45056 rows --\n"
           "            each in-partition pair a_ij == a_ji is stored once)\n"
           "  -c file   plan cache: reuse the permutation + layout saved by an earlier run on the same matrix,\n"
           "            or write it (the reference repeats mt-metis + COO2EHYB on every run)\n"
           "  -m name   ./read/name.mtx (Matrix Market, general or symmetric)\n"
           "  -g spec   banded:n:band:block | fem3d:n:dof:nx:ny:ppm:scramble | rmat:scale:edges |\n"
           "            stencil2d:nx:ny:points:extra | kkt3d:nx | mesh3d:n:dof:knn:grade_permille\n"
           "  -w mode   1 = reference window (contiguous), 2 = halo window (default)\n");
}

int main(int argc, char* argv[])
{
    int MAXIter = 0;
    char fileName[512];
    fileName[0] = '\0';
    std::string gen;
    cb_s cb;
    init_cb(&cb);
    ehyb_config cfg;
    memset(&cfg, 0, sizeof cfg);  // zero = default; resolved once the matrix (and its symmetry) is known
    int sym_opt = -1;             // -S: symmetric pair storage for symmetric matrices (default on)

    int oc;
    std::string cache;
    while ((oc = getopt(argc, argv, "m:i:r:t:f:p:g:w:l:T:c:S:X:I:vh")) != -1) {
        switch (oc) {
            case 'c': cache = optarg; break;
            case 'S': sym_opt = atoi(optarg); break;
            case 'm':
                snprintf(fileName, sizeof fileName, "./read/%s.mtx", optarg);
                printf("filename is %s\n", fileName);
                break;
            case 'i': MAXIter = atoi(optarg); break;
            case 't': break;  // parsed and ignored (solver_test.c:291-294)
            case 'r': break;
            case 'p':
                if (atoi(optarg) == 1) cb.PRECOND = true;
                break;
            case 'f':
                if (atoi(optarg) == 1) cb.FACT = false;
                break;
            case 'g': gen = optarg; break;
            case 'w': cfg.window_mode = atoi(optarg); break;
            case 'l': cfg.lds_doubles = atoi(optarg); cfg.part_rows = 0; break;
            case 'T': cfg.threads = atoi(optarg); break;
            case 'X': cfg.xcd_map = atoi(optarg) ? 1 : 2; break;
            case 'I': cfg.items_per_cu = atoi(optarg); break;
            case 'v': cfg.verbose = 1; break;
            case 'h': usage(); return 0;
            default: printf("option/arguments error!\n"); usage(); return 2;
        }
    }
    if ((fileName[0] == '\0' && gen.empty()) || MAXIter == 0) {
        printf("file name or max iteration number missing\n");
        return 2;
    }
    if (!cb.RODR || !cb.CACHE || !cb.BLOCK) {
        printf("this program only test RODR, BLOCK, and CACHE enabled case\n");
        return 2;
    }
    // --------------------------------- read / generate the matrix
    matrixCOO A;
    int symmetric = 0;
    int rc;
    if (!gen.empty()) {
        size_t colon = gen.find(':');
        std::string kind = gen.substr(0, colon);
        std::vector<long long> a = split_numbers(colon == std::string::npos ? "" : gen.c_str() + colon + 1);
        auto arg = [&](size_t k, long long dflt) { return k < a.size() ? a[k] : dflt; };
        if (kind == "banded") {
            rc = ehyb_gen_banded((int)arg(0, 1 << 16), (int)arg(1, 32), (int)arg(2, 1024), &cfg, &A);
        } else if (kind == "fem3d") {
            rc = ehyb_gen_fem3d((int)arg(0, 30000), (int)arg(1, 3), (int)arg(2, 22), (int)arg(3, 22), (int)arg(4, 13500),
                                (int)arg(5, 1), 1, &cfg, &A);
            symmetric = 1;
        } else if (kind == "rmat") {
            rc = ehyb_gen_rmat((int)arg(0, 14), arg(1, 1 << 17), 1, &cfg, &A);
        } else if (kind == "stencil2d") {
            rc = ehyb_gen_stencil2d((int)arg(0, 150), (int)arg(1, 150), (int)arg(2, 5), (int)arg(3, 3000), 1, &cfg, &A);
            symmetric = 1;
        } else if (kind == "mesh3d") {
            rc = ehyb_gen_mesh3d((int)arg(0, 30000), (int)arg(1, 3), (int)arg(2, 14), (int)arg(3, 1500), 1, &cfg, &A);
            symmetric = 1;
        } else if (kind == "kkt3d") {
            rc = ehyb_gen_kkt3d((int)arg(0, 20), &cfg, &A);
            symmetric = 1;
        } else {
            printf("unknown generator '%s'\n", kind.c_str());
            return 2;
        }
        printf("generated %s: %d rows, %d entries\n", gen.c_str(), rc == EHYB_OK ? A.dimension : 0, rc == EHYB_OK ? A.totalNum : 0);
    } else {
        rc = ehyb_mm_read(fileName, &cfg, &A, &symmetric);
        if (rc == EHYB_OK) printf(symmetric ? "read symmetric matrix\n" : "read unsymmetric matrix\n");
    }
    if (rc != EHYB_OK) {
        printf("%s\n", ehyb_last_error());
        return 1;
    }
    const int n = A.dimension;
    // A symmetric matrix (MM banner, as solver_test.c:348-354 branches on it) of at least
    // EHYB_SYM_MIN_ROWS rows gets symmetric pair storage; -S 1 asks for it on smaller ones too, -S 0
    // turns it off.  The partition sizing depends on the choice, so it is (re)done here.
    if (symmetric && (sym_opt > 0 || (sym_opt < 0 && n >= EHYB_SYM_MIN_ROWS)) && cfg.window_mode != EHYB_WINDOW_REFERENCE)
        cfg.sym_pairs = 1;
    ehyb_config_resolve(&cfg, &cfg);
    {
        int np = 1, cache_rows = 0, kpp = 1;
        if (ehyb_sizing(n, &cfg, &np, &cache_rows, &kpp) != EHYB_OK) {
            printf("%s\n", ehyb_last_error());
            return 1;
        }
        A.nParts = np;
        A.vectorCacheSize = (uint16_t)(cache_rows > 65535 ? 65535 : cache_rows);
        A.kernelPerPart = (int16_t)kpp;
    }
    if (cfg.sym_pairs == 1) printf("symmetric pair storage on\n");
    printf("parts is %d with cachSize %d\n", A.nParts, (int)A.vectorCacheSize);  // solver_test.c:78,183
    printf("maxCol is %d\n", A.maxCol);

    // --------------------------------- x and the CPU reference product
    double* xCompare = (double*)malloc(sizeof(double) * n);
    double* y = (double*)calloc(n, sizeof(double));
    double* yAbs = (double*)calloc(n, sizeof(double));
    ehyb_x_glibc(n, xCompare);  // solver_test.c:89-92, 228-231
    struct timeval t0, t1;
    gettimeofday(&t0, NULL);
    for (int k = 0; k < A.totalNum; ++k) y[A.I[k]] += A.V[k] * xCompare[A.J[k]];  // solver_test.c:102 / 247,254
    gettimeofday(&t1, NULL);
    double cpu_ms = (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) * 1e-3;
    printf("CPU reference product: %f ms, %f Gflops (1 thread)\n", cpu_ms, 2e-6 * A.totalNum / cpu_ms);
    for (int k = 0; k < A.totalNum; ++k) yAbs[A.I[k]] += fabs(A.V[k] * xCompare[A.J[k]]);

    double* yResult = (double*)calloc(n, sizeof(double));
    double* xReorder = (double*)calloc(n, sizeof(double));
    double* yReorder = (double*)calloc(n, sizeof(double));

    // --------------------------------- plan cache (-c): permutation + finished layout from an
    // earlier run of the same matrix; the partitioner and the conversion are skipped on a hit
    ehyb_plan* plan = nullptr;
    std::vector<int> cachedList;
    uint64_t key = 0;
    if (!cache.empty()) {
        key = ehyb_matrix_key(&A);  // of the matrix as read, before it is permuted
        cachedList.resize(n);
        if (ehyb_plan_load(cache.c_str(), key, &plan, cachedList.data()) == EHYB_OK) {
            printf("plan cache hit: %s (reorder and conversion skipped)\n", cache.c_str());
        } else {
            printf("plan cache miss: %s\n", ehyb_last_error());
            plan = nullptr;
        }
    }

    // --------------------------------- reorder (solver_test.c:369-376)
    if (!plan) {
        gettimeofday(&t0, NULL);
        cfg.part_boundary_cap = n + 1;  // the library's readers and generators allocate dimension + 1 boundaries
        rc = ehyb_matrix_reorder(&A, symmetric, &cfg);
        gettimeofday(&t1, NULL);
        if (rc != EHYB_OK) {
            printf("reorder failed: %s\n", ehyb_last_error());
            return 1;
        }
        printf("reorder time is %f ms\n", (t1.tv_sec - t0.tv_sec) * 1e3 + (t1.tv_usec - t0.tv_usec) * 1e-3);
    }
    const int* reorderList = plan ? cachedList.data() : A.reorderList;
    vectorReorder(n, xCompare, xReorder, reorderList);

    // --------------------------------- the hot path (solver_test.c:382-383)
    int realIter = 0;
    if (cache.empty()) {
        rc = spmvGPuEHYB_cfg(&A, xReorder, yReorder, MAXIter, &realIter, &cfg, nullptr);  // spmvGPuEHYB with this run's knobs
        if (rc != EHYB_OK) {
            printf("spmvGPuEHYB failed (%d): %s\n", rc, ehyb_last_error());
            return 1;
        }
    } else {
        // the same sequence through the plan API (what spmvGPuEHYB does inside), so that the
        // layout can come from / go to the cache file
        if (!plan) {
            rc = ehyb_plan_create_host(&A, 0, n, &cfg, &plan);
            if (rc == EHYB_OK && (rc = ehyb_plan_save(plan, A.reorderList, key, cache.c_str())) == EHYB_OK)
                printf("plan cache written: %s\n", cache.c_str());
        }
        ehyb_stats st;
        void *dx = nullptr, *dy = nullptr;
        double ms = 0;
        if (rc == EHYB_OK) rc = ehyb_plan_upload(plan);
        if (rc == EHYB_OK) rc = ehyb_plan_stats(plan, &st);
        if (rc == EHYB_OK) printf("sizeER is %lld\n", (long long)st.size_er);  // spmv.cu:82
        if (rc == EHYB_OK) rc = ehyb_dev_alloc(sizeof(double) * n, &dx);
        if (rc == EHYB_OK) rc = ehyb_dev_alloc(sizeof(double) * n, &dy);
        if (rc == EHYB_OK) rc = ehyb_h2d(dx, xReorder, sizeof(double) * n);
        if (rc == EHYB_OK) rc = ehyb_spmv_bench(plan, (const double*)dx, (double*)dy, nullptr, 10, MAXIter, &ms, nullptr, nullptr);
        if (rc == EHYB_OK) rc = ehyb_d2h(yReorder, dy, sizeof(double) * n);
        if (rc != EHYB_OK) {
            printf("plan path failed (%d): %s\n", rc, ehyb_last_error());
            return 1;
        }
        printf("iter is %d, time is %f ms, GPU Gflops is %f\n ", MAXIter, ms, 2.0 * st.nnz * MAXIter / (ms * 1e6));  // spmv.cu:121
        realIter = MAXIter;
        ehyb_dev_free(dx);
        ehyb_dev_free(dy);
        ehyb_plan_destroy(plan);
    }
    vectorRecover(n, yReorder, yResult, reorderList);

    // --------------------------------- compare (solver_test.c:389) + strict tolerance
    int loose = compare(yResult, y, 0.01, n);
    double worst = 0;
    int strict = 0;
    for (int i = 0; i < n; ++i) {
        double d = fabs(y[i] - yResult[i]);
        double scale = yAbs[i] > 0 ? yAbs[i] : 1e-300;
        if (d / scale > worst) worst = d / scale;
        if (d > 1e-12 * yAbs[i]) ++strict;
    }
    printf("strict check: max |dy|/sum|a*x| = %.3e, rows over 1e-12: %d\n", worst, strict);

    ehyb_matrix_free(&A);
    free(yResult), free(xReorder), free(yReorder), free(y), free(yAbs), free(xCompare);
    // The verdict is the stated tolerance.  compare() above is the reference's own report
    // (solver_test.c:389 prints it and carries on): its 1 % test is relative to |y_i| alone, so a row
    // whose terms cancel to rounding level is flagged under any change of summation order.
    if (loose && !strict)
        printf("compare(): %d rows flagged, all within 1e-12 of sum|a*x| (cancellation rows)\n", loose);
    if (strict) {
        printf("FAILED\n");
        return 1;
    }
    printf("PASSED\n");
    return 0;
}
