// Test infrastructure (NOT part of the product): exercises libehyb.so's optional mt-metis backend
// (cfg.partitioner = EHYB_PART_MTMETIS, csrc/reorder.cpp) with the partitioner the reference links:
// /root/reference/libmtmetis.a (mt-metis 0.7.2, vendored there as a binary) is linked STATICALLY into
// this executable, exactly as the reference's Makefile links it into spmvAlg1.out (Makefile:11,19), and
// exported with -rdynamic so that libehyb.so finds MTMETIS_PartGraphKway in the process image -- the
// integration INTEGRATION.md describes.  (The archive is not position independent, so it cannot be
// wrapped into a shared object.)
//
// usage: mtmetis_driver <generator> <args...> -- out_prefix
//   generators: fem3d n dof nx ny ppm scramble seed | stencil2d nx ny points extra seed | rmat scale edges seed | kkt3d nx
// For the built-in partitioner and for mt-metis (same call as reordering.c:270-293: unit weights,
// ubvec 1.001, 1 thread) it prints one JSON line {partitioner, parts, edge_cut, ell_share, padding,
// halo_cols, seconds} and saves the mt-metis plan + permutation to <out_prefix>.plan for the oracle walk.
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/time.h>

#include <string>
#include <vector>

#include "ehyb.h"

static double now()
{
    struct timeval t;
    gettimeofday(&t, NULL);
    return t.tv_sec + 1e-6 * t.tv_usec;
}

static int generate(int argc, char** argv, const ehyb_config* cfg, matrixCOO* m, int* symmetric)
{
    const std::string g = argv[1];
    auto a = [&](int i) { return atoll(argv[2 + i]); };
    *symmetric = 1;
    if (g == "fem3d" && argc >= 9) return ehyb_gen_fem3d((int)a(0), (int)a(1), (int)a(2), (int)a(3), (int)a(4), (int)a(5), (uint64_t)a(6), cfg, m);
    if (g == "stencil2d" && argc >= 7) return ehyb_gen_stencil2d((int)a(0), (int)a(1), (int)a(2), (int)a(3), (uint64_t)a(4), cfg, m);
    if (g == "kkt3d" && argc >= 3) return ehyb_gen_kkt3d((int)a(0), cfg, m);
    if (g == "rmat" && argc >= 5) {
        *symmetric = 0;
        return ehyb_gen_rmat((int)a(0), a(1), (uint64_t)a(2), cfg, m);
    }
    fprintf(stderr, "unknown generator / too few arguments\n");
    return EHYB_ERR_ARG;
}

int main(int argc, char** argv)
{
    int sep = -1;
    for (int i = 1; i < argc; ++i)
        if (!strcmp(argv[i], "--")) sep = i;
    if (sep < 2 || sep + 1 >= argc) {
        fprintf(stderr, "usage: %s <generator> <args...> -- <out_prefix> [lds_doubles]\n", argv[0]);
        return 2;
    }
    const std::string prefix = argv[sep + 1];
    const int lds = sep + 2 < argc ? atoi(argv[sep + 2]) : 0;
    for (int pass = 0; pass < 2; ++pass) {
        ehyb_config cfg;
        memset(&cfg, 0, sizeof cfg);
        cfg.lds_doubles = lds;
        cfg.partitioner = pass == 0 ? EHYB_PART_MULTILEVEL : EHYB_PART_MTMETIS;
        cfg.cap_split = 2;  // compare the partitioners themselves: no capacity-driven bisection afterwards
        matrixCOO m;
        int symmetric = 1;
        int rc = generate(sep, argv, &cfg, &m, &symmetric);
        if (rc != EHYB_OK) {
            fprintf(stderr, "generate: %s\n", ehyb_last_error());
            return 1;
        }
        const uint64_t key = ehyb_matrix_key(&m);
        cfg.part_boundary_cap = m.dimension + 1;
        const double t0 = now();
        rc = ehyb_matrix_reorder(&m, symmetric, &cfg);
        const double t1 = now();
        if (rc != EHYB_OK) {
            fprintf(stderr, "reorder (%s): %s\n", pass ? "mt-metis" : "built-in", ehyb_last_error());
            return 1;
        }
        // edge cut of the partition, counted on the permuted matrix (off-diagonal entries whose ends lie in
        // different partitions; both directions of a symmetric pattern are stored, so halve)
        std::vector<int> part_of(m.dimension);
        for (int p = 0; p < m.nParts; ++p)
            for (int r = m.partBoundary[p]; r < m.partBoundary[p + 1]; ++r) part_of[r] = p;
        long long cut = 0;
        int max_rows = 0;
        for (int p = 0; p < m.nParts; ++p) max_rows = m.partBoundary[p + 1] - m.partBoundary[p] > max_rows ? m.partBoundary[p + 1] - m.partBoundary[p] : max_rows;
        for (int k = 0; k < m.totalNum; ++k) cut += part_of[m.I[k]] != part_of[m.J[k]];
        ehyb_plan* plan = NULL;
        rc = ehyb_plan_create_host(&m, 0, m.dimension, &cfg, &plan);
        if (rc != EHYB_OK) {
            fprintf(stderr, "plan: %s\n", ehyb_last_error());
            return 1;
        }
        ehyb_stats st;
        ehyb_plan_stats(plan, &st);
        printf("{\"partitioner\": \"%s\", \"rows\": %d, \"nnz\": %lld, \"parts\": %d, \"max_part_rows\": %d, \"cut_entries\": %lld, "
               "\"ell_share\": %.6f, \"padding_share\": %.6f, \"halo_cols\": %lld, \"residual_entries\": %lld, \"seconds\": %.3f}\n",
               pass ? "mt-metis 0.7.2 (reference's libmtmetis.a)" : "built-in multilevel", m.dimension, (long long)st.nnz, m.nParts, max_rows,
               symmetric ? cut / 2 : cut, (double)st.nnz_ell / (double)st.nnz, (double)st.ell_padding / (double)(st.size_block_ell ? st.size_block_ell : 1),
               (long long)st.halo_cols, (long long)st.nnz_er, t1 - t0);
        if (pass == 1) {
            rc = ehyb_plan_save(plan, m.reorderList, key, (prefix + ".plan").c_str());
            if (rc != EHYB_OK) {
                fprintf(stderr, "save: %s\n", ehyb_last_error());
                return 1;
            }
        }
        ehyb_plan_destroy(plan);
        ehyb_matrix_free(&m);
    }
    return 0;
}
