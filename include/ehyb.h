/*
 * ehyb.h -- C-ABI of the MI355X-native Explicit-Caching HYB (EHYB) fp64 SpMV.
 *
 * Plain C: pointers, sizes and an opaque plan handle; no C++ or torch types cross
 * this boundary.  One shared library (libehyb.so) exports everything declared here
 * and in spmv.h / reordering.h.  Each entry point names the reference interface it
 * stands in for (paths are into the reference repository).
 *
 *   reference symbol / code                         replaced by
 *   ---------------------------------------------   --------------------------------------
 *   spmvGPuEHYB                 spmv.cu:61-133      spmvGPuEHYB (spmv.h) = plan_create +
 *                                                   10 warm-ups + MAXIter x ehyb_spmv
 *   COO2EHYB + helpers          convert.c:61-369    ehyb_plan_create_host (layout builder)
 *   cudaMallocTransDataEHYB     spmv.cu:6-60        ehyb_plan_upload
 *   matrixVectorEHYB[_small]    kernel.cu:490-552   ehyb_spmv / ehyb_spmv_phase
 *     kernelCachedBlockedELL*   kernel.cu:110-284     -> ehyb_ell_kernel   (HIP, gfx950)
 *     ER loop + vecReorderER    kernel.cu:169-194,69-77 -> ehyb_er_kernel  (HIP, gfx950)
 *     longRowKernel             kernel.cu:43-67       -> long segments of ehyb_er_kernel
 *   matrixReorder[_unsym]       reordering.c:41-378 ehyb_matrix_reorder (+ C++ names in
 *                                                   reordering.h)
 *   vectorReorder/vectorRecover reordering.c:380-391 ehyb_vector_reorder / _recover
 *   sizing heuristic            solver_test.c:53-77,158-182  ehyb_sizing
 *   matrixRead_sym/_unsym       solver_test.c:31-265 ehyb_mm_read (+ ehyb_x_glibc)
 *   MTMETIS_PartGraphKway       reordering.c:126-139,280-293  ehyb_partition_graph
 *                                                   (built-in multilevel k-way; an
 *                                                   mt-metis build can be plugged in,
 *                                                   see INTEGRATION.md)
 */
#ifndef EHYB_H
#define EHYB_H

#include <stdint.h>
#include <stddef.h>
#include "spmv.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ------------------------------------------------------------------ status */
typedef enum ehyb_status {
    EHYB_OK            = 0,
    EHYB_ERR_ARG       = 1,  /* null pointer, negative size, inconsistent matrixCOO   */
    EHYB_ERR_ALLOC     = 2,  /* host allocation failed                                */
    EHYB_ERR_HIP       = 3,  /* a HIP runtime call failed (message via ehyb_last_error) */
    EHYB_ERR_NO_DEVICE = 4,  /* no gfx950-capable device visible                      */
    EHYB_ERR_IO        = 5,  /* file missing / unreadable                             */
    EHYB_ERR_FORMAT    = 6,  /* Matrix Market content not supported or malformed      */
    EHYB_ERR_INTERNAL  = 7,  /* layout self-check failed (the reference exit()s here:
                                convert.c:122-125,226-263,287-303)                    */
    EHYB_ERR_STATE     = 8   /* plan not uploaded / wrong call order                  */
} ehyb_status;

/* Thread-local text of the last failure ("" if none). Never NULL. */
const char* ehyb_last_error(void);
/* "ehyb-mi355x <version> gfx950" */
const char* ehyb_version(void);

/* ------------------------------------------------------------------ config */
enum { EHYB_WINDOW_DEFAULT = 0, EHYB_WINDOW_REFERENCE = 1, EHYB_WINDOW_HALO = 2 };
enum { EHYB_PART_AUTO = 0, EHYB_PART_CONTIGUOUS = 1, EHYB_PART_MULTILEVEL = 2, EHYB_PART_MTMETIS = 3,
       EHYB_PART_DEGREE = 4 /* rows in order of falling degree (row + column entries), cut into equal blocks: what
                               EHYB_PART_AUTO falls back to on graphs a k-way partitioner finds nothing to cut in
                               (power-law: R-MAT) -- the hub columns then share x panels and the hub rows share row
                               blocks of the panel-form residual (er_panel.cpp) */ };

#define EHYB_LDS_MAX_DOUBLES 20480   /* 160 KiB of LDS per workgroup on gfx950 */
#define EHYB_SLAB_ROWS       64      /* one row per lane of a wave64           */

/* Which shape for which size (round 2, NOTEBOOK.md 2: fem3d with 3 unknowns per node, ~78 entries per
 * row, microseconds per SpMV: direct / window with every entry stored / symmetric pair storage):
 *    24,576 rows   5.6 /  8.8 /  9.4        65,535 rows  14.8 / 16.5 / 13.1
 *    32,766 rows   7.2 /  9.3 /  9.7        81,918 rows  17.4 / 17.4 / 13.8
 *    40,959 rows   9.5 / 10.6 /  9.8        98,304 rows  20.6 / 19.7 / 15.0
 *    49,152 rows  11.6 / 14.3 / 11.1       131,070 rows  25.9 / 21.9 / 15.7
 * Symmetric pair storage (one workgroup per partition) needs enough partitions to fill the 256 CUs: it
 * pays from about 45,000 rows; callers that pick the storage from the matrix's symmetry (matrixReorder,
 * spmvGPuEHYB, solver_test, bench.py) use EHYB_SYM_MIN_ROWS.  Below, and for unsymmetric matrices up to
 * EHYB_DIRECT_MAX_ROWS, the direct shape (cfg.direct: no window, one small launch) is the fastest. */
#define EHYB_SYM_MIN_ROWS 45056
#define EHYB_DIRECT_MAX_ROWS 81920

/* OpenMP threads the host builder uses when cfg.host_threads is 0: what OpenMP would take, capped by
 * the CPUs the process may really use (affinity mask, cgroup CPU quota). */
int ehyb_host_threads(void);

/*
 * Tuning knobs.  They take the place of the reference's compile-time constants
 * (kernel.h:20-28: warpSize 32, smSize 82, maxSharedMem 93 KiB, threadELL 1024,
 * threadLongVec 512).  Zero in any field means "use the default".
 */
typedef struct ehyb_config {
    int32_t lds_doubles;   /* x-window capacity per workgroup (own rows + halo), <= 20480 */
    int32_t part_rows;     /* upper bound of rows per partition, <= lds_doubles           */
    int32_t threads;       /* ELL workgroup size: multiple of 64, <= 1024                 */
    int32_t window_mode;   /* EHYB_WINDOW_REFERENCE: window = [start, start+cache) exactly
                              as convert.c:247; EHYB_WINDOW_HALO: own rows + the most
                              referenced outside columns gathered into LDS                */
    int32_t items_per_cu;  /* ELL work items per CU (load-balance granularity)            */
    int32_t partitioner;   /* EHYB_PART_*                                                 */
    int32_t er_seg_len;    /* residual rows longer than this are split into segments      */
    int32_t host_threads;  /* OpenMP threads of the host builder (0 = runtime default)    */
    int32_t verbose;       /* 1: print the reference's format metrics (toER, waste, ...)  */
    int32_t seed;          /* partitioner tie-breaking seed (deterministic per seed)      */
    int32_t n_top;         /* top-level row blocks (one per GPU); 0/1 = single GPU        */
    int32_t er_threads;    /* residual workgroup size                                     */
    int32_t ell_variant;   /* how ELL slabs reach the waves of a workgroup: 0/1 = LDS counter (default),
                              3 = static round-robin (A/B arm)                                   */
    int32_t col_sharing;   /* 0/1 = rows with the column list of the row above share its indices, 2 = off */
    int32_t fuse_er;       /* residual placement: 0 = automatic (inside the ELL launch iff it holds < 0.2 %
                              of the entries), 1 = always inside (single GPU only), 2 = own launch     */
    int32_t cap_split;     /* 0/1 = bisect partitions whose halo overflows the window (reorder step), 2 = off */
    int32_t hub_rule;      /* 0/1 = rows that would pad their slab by > 25 % go to the residual whole
                              (the reference's long-row intent, convert.c:92-101), 2 = off          */
    int32_t sym_pairs;     /* symmetric pair storage (halo window): an in-partition pair a_ij == a_ji is stored
                              once; the owning lane adds a_ij*x_j to its own row and a_ij*x_i to row j's
                              accumulator in LDS (ds_add_f64), so one value read serves two entries.  One
                              workgroup per partition; ehyb_sizing makes nParts a multiple of 256 and the
                              window the whole 160 KiB (own rows twice -- x image and accumulators -- plus halo).  Set it BEFORE reading/generating/reordering the matrix (the
                              partition sizes depend on it).  Results do not depend on the matrix being
                              symmetric (entries without an equal partner stay as they are), but the order
                              of the LDS adds varies from run to run (last-bit differences).
                              1 = on -- what solver_test and bench.py choose for symmetric inputs of at
                              least EHYB_SYM_MIN_ROWS rows; 0/2 = off */
    int32_t part_boundary_cap; /* ints the caller's matrixCOO.partBoundary can hold.  ehyb_matrix_reorder may end
                              with MORE partitions than m->nParts asked for (partitions whose halo overflows the
                              window are bisected, cap_split; two-level partitions, n_top > 1) and writes
                              nParts+1 boundaries.  0 = unknown: exactly the m->nParts+1 entries of the reference
                              contract (spmv.h:31) -- then the partition count is never raised (no capacity
                              split; n_top > 1 fails with EHYB_ERR_ARG if it would need more).  Matrices
                              allocated by this library (ehyb_mm_read, ehyb_gen_*, ehyb_matrix_from_csr) hold
                              dimension+1 entries; the reference harness callocs `dimension`
                              (solver_test.c:42,146), which is what matrixReorder[_unsym] assume.          */
    int32_t er_mode;       /* form of a residual too large to ride inside the ELL launch: 1 = CSR segments
                              (ehyb_er_kernel: x gathered from global memory), 2 = panel form (two streaming passes,
                              x panels and y blocks in LDS: er_panel.cpp), 0 = automatic: the panel form from 2^21
                              residual entries up when the residual shows no locality (more than one distinct
                              128-byte line of x per two consecutive entries), else CSR segments            */
    int32_t er_panel_cols; /* panel form: columns per x panel staged in LDS (<= 16384 = 128 KiB, the default)          */
    int32_t er_block_rows; /* panel form: most rows of a y block accumulated in LDS (<= 16384, default 2048)          */
    int32_t direct;        /* small matrices: 0 = automatic (plans of at most EHYB_DIRECT_MAX_ROWS rows, single GPU,
                              plain storage, window sizing left at its defaults), 1 = on, 2 = off.  On: no LDS window at all -- every row is multiplied by
                              the row-segment kernel straight from global x (which sits in L2 at this size), one
                              launch of 256-thread workgroups, y assigned, not accumulated.  A 160 KiB window per
                              1024-thread workgroup is the wrong shape when the whole matrix is a few MB: the
                              reference has a small-matrix branch for the same reason (kernel.cu:197-284,
                              solver_test.c:56-69).                                                           */
    int32_t ell_prune;     /* with the residual in panel form: 0/1 = a partition whose LDS window costs more bytes and
                              L2 requests than the panel form would for its entries (padding, halo gathers: power-law
                              matrices) goes to the residual whole, 2 = never                                 */
    int32_t value_map;     /* 1 = the plan remembers, for every slot of its value streams, which entry of the matrix it
                              was filled from (4 B per stored value on the host, and on the device from the first
                              ehyb_plan_set_values on): the NUMERIC phase of the build can then be repeated on the
                              GPU for new values on the same pattern.  0 = off                                 */
    /* ---- tuning knobs of the A/B tools (tools/); every one of them used to be an environment variable of
       the library.  The library reads NO tuning variable from the environment any more (the one variable left,
       EHYB_MTMETIS_LIB, names a shared object to load the optional mt-metis backend from). */
    int32_t prune_pct;     /* ell_prune: a window is given up when it costs more than this share (per cent) of what
                              the panel form would cost for its entries; 0 = 110                                */
    int32_t er_units1;     /* panel form: work items (workgroups) pass 1 aims at (0 = one per 49,152
                              residual entries, between 512 and 4096)                                           */
    int32_t er_units2;     /* panel form: row blocks pass 2 aims at (0 = 2048)                                 */
    int32_t graph_compress;/* the k-way partitioner works on the compressed graph where rows come in groups with one column
                              list (the unknowns of a node): 0 = with symmetric pair storage only (plain storage runs 4 %
                              slower on such partitions, reorder.cpp), 1 = always, 2 = never, 3 = always, followed by one
                              refinement on the rows themselves (the groups may then be cut)                      */
    int32_t balance;       /* symmetric pair storage, what the partitions are balanced on: 0 = rows unless the row lengths
                              vary by more than 30 % (sigma/mean), 1 = entries, 2 = rows                         */
    int32_t req_margin;    /* entry-balanced partitions of a graded mesh: partitions asked for = nParts - margin; 0 =
                              max(2, nParts/64), -1 = no margin                                                   */
    int32_t sym_slack_permille; /* ehyb_sizing, symmetric pair storage: rows a partition may hold above the mean, in
                              1/1000 (0 = 30)                                                                     */
    int32_t xcd_map;       /* 0/1 = workgroup b takes the work item that gives every XCD one contiguous run of items
                              (plain storage; the panel form's pass 1: the units of one x panel on one XCD), 2 = items
                              in blockIdx order                                                                   */
    int32_t graphs;        /* 0/1 = the timed loop of ehyb_spmv_bench and the iterations of ehyb_cg/ehyb_pcg are replayed
                              from hipGraphs, 2 = plain launches (A/B, debugging)                                */
    int32_t er_sums;       /* panel form, pass 1: how the products of one row inside a 64-entry chunk are added up: 0/1 =
                              segmented DPP scan in registers, 2 = ds_add_f64 into per-wave LDS words (round 2's way) */
    int32_t er_panel_threads; /* panel form, pass 1: workgroup size, 512 or 1024 (0 = automatic)                 */
    int32_t er_queue;      /* panel form, pass 1: 2 = one workgroup per item; 1 = one resident round of workgroups takes the items from
                              per-XCD queues, an XCD whose own eighth is used up helping the one with the most left (the 8 XCDs do not
                              stream equally fast), and a workgroup that takes the next item of the SAME panel does not stage it again;
                              0 = automatic: the queues from six items per resident workgroup up (R-MAT 2^24, 2,700 items: 574 -> 544 us;
                              2^22, 770 items: 132 -> 138 us, so not there)                                                    */
    int32_t symbolic;      /* ehyb_plan_create / ehyb_plan_create_segs (the calls that build AND upload): where the panel form
                              of a residual is built when it is certain to be used (R-MAT: every partition given up, or
                              er_mode = 2).  0/2 = on the DEVICE, from the entries in row order (radix sort by panel, row,
                              column; scans for the partial sums and their slots): the arrays are the host builder's, entry
                              for entry, where the rows arrive in column order; 1 = on the host.  ehyb_plan_create_host never
                              leaves anything to the device                                                        */
    int32_t cg_fused_dot;  /* ehyb_cg / ehyb_pcg: 0/1 = p . (A p) is left by the multiply itself where the plan multiplies in one
                              window launch (one vector kernel fewer per iteration), 2 = always the separate dot kernel (A/B) */
    int32_t ell_alternate; /* successive multiplies of a plan walk the slabs of every partition in alternating directions, so that a
                              launch starts with what the one before it left in the 256 MB Infinity Cache (a solver multiplies with
                              the same matrix again and again): 0 = where the plan's stream is larger than that cache and at most
                              8 GB (a smaller one stays resident anyway, of a far larger one the cache holds too little to pay for
                              the walk from the short slabs up), 1 = always, 2 = never (always first to last).  With more work items
                              than resident workgroups the items are taken from the far end as well.  Pass 1 of the panel
                              residual alternates the same way                                                        */
    int32_t row_split;     /* panel form, multi-GPU: a row block of pass 2 never straddles this row (0 = none).  The rows from it on
                              are the FOREIGN rows of a rank -- partial sums it computes for other ranks from its own x entries
                              (dist.py: exchange "cover") -- and ehyb_spmv_part can close them early (EHYB_PART_LAST_FOREIGN), so
                              that they travel while the rank's own rows are still being multiplied                       */
    int32_t col_map;       /* host builder, how a partition finds the window place of an outside column: 0/1 = in a look-up array over the
                              columns, one per host thread (matrices of up to 4 M columns: 16 MiB per thread), 2 = in the sorted list
                              of the window's outside columns by binary search (A/B; the way for larger inputs).  The layouts are
                              the same array for array                                                                  */
    int32_t er_nt;         /* panel form, pass 2: the partial sums and their row words are read with the non-temporal hint (past the caches):
                              0 = where they are more than half the 256 MB Infinity Cache (10 B per partial sum), 1 = always, 2 = never (A/B) */
    int32_t ell_nt;        /* the window kernel reads its value stream (read once per multiply) with the non-temporal hint, past the caches:
                              0/3 = every slab but the END of an alternating walk (the share of the stream the Infinity Cache can hold
                              is read with plain loads, to be found there by the next launch; every slab where the walk does not
                              alternate; none where the whole stream fits that cache), 1 = every slab, 2 = never (rounds 1-3)      */
    int32_t reserved[21];  /* zero; keeps sizeof(ehyb_config) = 260 bytes when knobs are added                  */
} ehyb_config;

void ehyb_config_default(ehyb_config* cfg);
/* *out = *in with every "0 = default" field replaced by the value the library will use given the
 * other fields (window and partition sizes depend on window_mode and sym_pairs).  in may be NULL
 * (all defaults) and may alias out. */
void ehyb_config_resolve(const ehyb_config* in, ehyb_config* out);

/* ------------------------------------------------- host pre-step (a-7, a-9) */

/*
 * Partition/window sizing for an n-row matrix: the MI355X re-derivation of
 * solver_test.c:53-77 / 158-182 (82 SMs x 93 KiB -> 256 CUs x 160 KiB, wave64).
 * Outputs may be NULL.  vectorCacheSize is the window the partitions are sized for,
 * kernelPerPart the number of work items one partition is cut into.
 */
int ehyb_sizing(int dimension, const ehyb_config* cfg,
                int* nParts, int* vectorCacheSize, int* kernelPerPart);

/*
 * k-way partition of an undirected graph in CSR form (xadj has n+1 entries; adjncy may
 * contain self loops, they are ignored).  Stands in for MTMETIS_PartGraphKway as called
 * at reordering.c:126-139 / 280-293 (unit vertex weights, ubvec 1.001): every part gets
 * at most max_part_rows vertices (<= 0: ceil(1.001*n/nparts)).  part[v] in [0,nparts).
 * If vwgt is non-NULL parts are balanced on it instead (used for the per-GPU blocks).
 */
int ehyb_partition_graph(int n, const int64_t* xadj, const int* adjncy, const int* vwgt,
                         int nparts, int max_part_rows, const ehyb_config* cfg,
                         int* part, int64_t* edgecut);

/*
 * In-place symmetric permutation P*A*P^T of a row-grouped COO matrix, as
 * matrixReorder (symmetric_pattern != 0, reordering.c:231-378) or matrixReorder_unsym
 * (== 0: the pattern is symmetrised first, reordering.c:41-228):
 *   - k-way partition into m->nParts parts (m->nParts and m->vectorCacheSize must be set,
 *     e.g. by ehyb_sizing);
 *   - new numbering: partition-contiguous, rows of a partition sorted by their number of
 *     in-partition entries, descending (reordering.c:312-334) -- ties keep the old order,
 *     so the permutation is deterministic here;
 *   - I/J/V are replaced (old arrays freed with free()), rowIdx/numInRow/numInRow2/
 *     partBoundary/reorderList are filled (reordering.c:335-362).
 * partBoundary and reorderList must be caller-allocated with >= nParts+1 and
 * >= dimension entries (the reference harness callocs `dimension` ints for both).
 */
int ehyb_matrix_reorder(matrixCOO* m, int symmetric_pattern, const ehyb_config* cfg);
/* The same with cfg.n_top > 1 row blocks (one per GPU): block_first[b] = first partition of block b,
 * block_first[n_top] = nParts (n_top+1 ints, caller-allocated; NULL: not wanted). */
int ehyb_matrix_reorder_blocks(matrixCOO* m, int symmetric_pattern, const ehyb_config* cfg, int* block_first);

/* v_rodr[list[i]] = v_in[i]  (reordering.c:380-384) */
void ehyb_vector_reorder(int dimension, const double* v_in, double* v_rodr, const int* list);
/* v[i] = v_rodr[list[i]]     (reordering.c:386-391) */
void ehyb_vector_recover(int dimension, const double* v_rodr, double* v, const int* list);

/* Cuts a reordered matrix into n_top runs of whole partitions with (nearly) equal entry counts:
 * part_of_block[b] = first partition of run b, n_top+1 entries.  A pure function of m (for matrices
 * whose two-level block boundaries were not kept; ehyb_matrix_reorder_blocks returns the real ones). */
int ehyb_top_boundary(const matrixCOO* m, const ehyb_config* cfg, int n_top, int* part_of_block);

/* ------------------------------------------------------------------- plan */
typedef struct ehyb_plan ehyb_plan;

/*
 * Build the EHYB layout on the host from a permuted matrix (COO2EHYB, convert.c:316-369).
 * No GPU needed.  Rows [row_begin,row_end) only (whole matrix: 0, dimension); the range
 * must start and end on partition boundaries.  The plan keeps no pointer into m.
 */
int ehyb_plan_create_host(const matrixCOO* m, int row_begin, int row_end,
                          const ehyb_config* cfg, ehyb_plan** plan);
/*
 * The same for a multi-GPU rank (no reference counterpart): the columns come in n_col_segs SEGMENTS,
 * col_seg_first[0] = 0 < ... <= col_seg_first[n_col_segs] = dimension, every start but the first even -- the rank's
 * own columns first, then the ghost columns in the order their x entries ARRIVE (one segment per exchange step).
 * A panel of the panel-form residual never straddles a boundary, so ehyb_spmv_part can multiply segment by segment
 * while later segments are still on the wire.  n_col_segs = 0: one segment (ehyb_plan_create_host).
 */
int ehyb_plan_create_host_segs(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg,
                               int n_col_segs, const int* col_seg_first, ehyb_plan** plan);
/* Allocate device arrays and copy the layout (cudaMallocTransDataEHYB, spmv.cu:6-60). */
int ehyb_plan_upload(ehyb_plan* plan);
/* Build + upload in one call, for the rows [0, dimension).  Unlike ehyb_plan_create_host followed by ehyb_plan_upload
 * this call may build on the DEVICE what it can (cfg.symbolic: the panel form of a residual without locality -- the
 * part of the symbolic phase that costs a power-law matrix seconds on the host).  Such a plan keeps the
 * pb_* streams on the device; ehyb_plan_host_array and ehyb_plan_save fetch them when asked. */
int ehyb_plan_create(const matrixCOO* m, const ehyb_config* cfg, ehyb_plan** plan);
/* The same for a row range and column segments (ehyb_plan_create_host_segs' arguments). */
int ehyb_plan_create_segs(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg, int n_col_segs, const int* col_seg_first,
                          ehyb_plan** plan);
void ehyb_plan_destroy(ehyb_plan* plan);

/*
 * On-disk cache of the pre-step: the reference repeats mt-metis and COO2EHYB on
 * every run (solver_test.c:369-382, spmv.cu:74).  ehyb_plan_save writes the host layout of a plan
 * together with the permutation that produced its matrix (reorder_list: n_cols ints, may be NULL)
 * and a key of the caller's choice -- ehyb_matrix_key(m) of the UNPERMUTED matrix is the intended
 * one.  ehyb_plan_load rebuilds a host plan (then ehyb_plan_upload); it fails with
 * EHYB_ERR_FORMAT when the file is damaged, from another version, or expect_key (non-zero) differs.
 * Same-machine cache, native byte order.
 */
uint64_t ehyb_matrix_key(const matrixCOO* m);
int ehyb_plan_save(const ehyb_plan* plan, const int* reorder_list, uint64_t matrix_key, const char* path);
int ehyb_plan_load(const char* path, uint64_t expect_key, ehyb_plan** plan, int* reorder_list);

/* Format metrics; the first five are the ones the reference prints
 * (convert.c:140,310; spmv.cu:82). */
typedef struct ehyb_stats {
    int64_t nnz;            /* stored entries in the plan's rows                      */
    int64_t nnz_ell;        /* entries multiplied from the LDS window ("kernel calculation") */
    int64_t nnz_er;         /* entries in the residual ("toER")                       */
    int64_t ell_padding;    /* zero fill in the ELL slabs ("wasteElement")            */
    int64_t size_block_ell; /* stored ELL elements incl. padding ("sizeBlockELL")     */
    int64_t size_er;        /* stored residual elements ("sizeER")                    */
    int64_t rows_er;        /* rows with a residual part ("numOfRowER")               */
    int64_t er_segments;    /* residual segments (long rows are split)                */
    int64_t n_rows;         /* rows covered by the plan                               */
    int64_t n_cols;         /* dimension                                              */
    int64_t n_parts;
    int64_t n_slabs;
    int64_t n_items;        /* ELL workgroups per multiply                            */
    int64_t halo_cols;      /* gathered window entries over all partitions            */
    int64_t window_loads;   /* doubles staged into LDS per multiply (all items)       */
    int64_t bytes_format;   /* bytes the kernels must move per multiply in this format */
    int64_t bytes_alg;      /* 12*nnz + 4*(rows+1) + 8*cols + 8*rows      */
    int64_t max_row;        /* longest row                                            */
    int64_t lds_bytes;      /* dynamic LDS per ELL workgroup                          */
    int64_t col_words;      /* stored 4-byte column words (2 x 16 bit) after sharing  */
    int64_t er_inline;      /* stored inline-residual elements incl. padding; > 0: the residual rides in
                               the ELL launch (ehyb_spmv is one launch), see EHYB_ARR_SLAB_META */
    int64_t sym_pairs;      /* stored entries that stand for a symmetric pair (cfg.sym_pairs): nnz_ell counts
                               both entries of such a pair, size_block_ell one                          */
    int64_t bytes_format_ell; /* the part of bytes_format the ELL launch moves (an inline residual included);
                               the rest belongs to the residual launch                                   */
    int64_t er_partials;    /* > 0: the residual is stored in panel form; partial sums written by its first pass
                               and read by its second, per multiply                                    */
} ehyb_stats;
int ehyb_plan_stats(const ehyb_plan* plan, ehyb_stats* out);

/* Read-only views of the host layout, for invariant tests (i-iv). */
enum {
    EHYB_ARR_PART_BOUNDARY = 0, /* int32  [n_parts+1]  first row of each partition            */
    EHYB_ARR_WIN_LEN       = 1, /* int32  [n_parts]    contiguous window length               */
    EHYB_ARR_HALO_PTR      = 2, /* int32  [n_parts+1]  into HALO_COLS                         */
    EHYB_ARR_HALO_COLS     = 3, /* int32  [halo_cols]  global column of each gathered entry   */
    EHYB_ARR_SLAB_PAIR_PTR = 4, /* uint32 [n_slabs+1]  prefix of slab widths / 2              */
    EHYB_ARR_SLAB_ROW      = 5, /* int32  [n_slabs]    first row of the slab                  */
    EHYB_ARR_SLAB_PART     = 6, /* int32  [n_slabs]    partition of the slab                  */
    EHYB_ARR_ELL_VAL       = 7, /* double [size_block_ell]  [pair][lane][2]                   */
    EHYB_ARR_ELL_COL       = 8, /* uint32 [col_words] two 16-bit window-local columns per word:
                                   word (pair k, group g) of slab s at SLAB_COL_PTR[s] + k*G_s + g  */
    EHYB_ARR_ITEMS         = 9, /* int32  [n_items*8]  one workgroup's work: {seg_begin, seg_end, slab_begin,
                                   slab_end, er_begin, er_end_len>=128, er_end_len>16, er_end} -- a run
                                   of slabs of equal cost (cut into SEGS at partition boundaries) and
                                   the residual segments of its rows, longest first                 */
    EHYB_ARR_ER_SEG_PTR    = 10,/* int64  [er_segments+1]                                     */
    EHYB_ARR_ER_SEG_ROW    = 11,/* int32  [er_segments] bit31 set: row has several segments   */
    EHYB_ARR_ER_COL        = 12,/* int32  [size_er]    global column                          */
    EHYB_ARR_ER_VAL        = 13,/* double [size_er]                                           */
    EHYB_ARR_ER_BINS       = 14,/* int32  [8]  {0, -, -, er_segments, ...}                        */
    EHYB_ARR_SLAB_COL_PTR  = 15,/* uint32 [n_slabs+1]  prefix of pairs * groups               */
    EHYB_ARR_LANE_GROUP    = 16,/* uint8  [n_slabs*64] column-list group of every lane        */
    EHYB_ARR_SLAB_META     = 17,/* uint32 [n_slabs*4]  {pair_ptr, col_ptr, row, pairs<<16 | er_pairs<<8 | G-1}:
                                   what the kernel reads.  er_pairs > 0 only in the inline-residual form
                                   (stats.er_inline): behind the slab's `pairs` ELL pairs the value stream
                                   holds er_pairs more pairs [pair][lane][2], and behind its pairs*G column
                                   words the column stream holds [er pair][2][lane] GLOBAL 32-bit columns */
    EHYB_ARR_SEGS          = 18,/* int32  [n_segs*8]   {partition, slab_begin, slab_end, halo_count, first row,
                                   end row, win_len, halo_begin}: one LDS window staging each       */
    EHYB_ARR_PERM          = 19,/* int32  [n_cols]     reorderList stored with a plan that came from
                                   ehyb_plan_load (empty for plans built in this process)          */
    EHYB_ARR_SLAB_LROW     = 20,/* uint16 [n_slabs*64] symmetric pair storage only: the row every lane works on, as
                                   its place in the partition's LDS image (row - even(partition start));
                                   0xFFFF = no row.  The rows of a partition sit in its slabs longest first. */
    /* panel form of the residual (stats.er_partials > 0; the CSR arrays 10-13 describe the same entries) */
    EHYB_ARR_PB_VAL        = 21,/* double [entries, every panel padded to a multiple of 64]  pass-1 order: by column
                                   panel, inside a panel by (row, column); padding = 0.0                   */
    EHYB_ARR_PB_COL        = 22,/* uint16 same length: column - first column of the panel                  */
    EHYB_ARR_PB_DST        = 23,/* uint32 same length: partial sum the entry belongs to (entries of one row that are
                                   neighbours inside a 64-entry chunk share one); 0xFFFFFFFF = padding     */
    EHYB_ARR_PB_UNITS1     = 24,/* int32  [4*u1] pass-1 units {first column, columns, first entry, end entry}: a stretch of the
                                   entries of one panel (multiples of 64), staged once                     */
    EHYB_ARR_PB_ROW        = 25,/* uint16 [er_partials] row of the partial - first row of its row block; partials are
                                   numbered by (row block, panel, row)                                     */
    EHYB_ARR_PB_UNITS2     = 26,/* int32  [4*u2] pass-2 work units {first partial, end partial, first row, rows}    */
    /* slot maps of the value streams (cfg.value_map; empty otherwise): index into the V array of the matrix the plan
       was built from, -1 = padding.  What ehyb_plan_set_values gathers through on the device. */
    EHYB_ARR_ELL_SRC       = 27,/* int32  same length as ELL_VAL                                                      */
    EHYB_ARR_ER_SRC        = 28,/* int32  same length as ER_VAL                                                       */
    EHYB_ARR_PB_SRC        = 29,/* int32  same length as PB_VAL                                                       */
    EHYB_ARR_ELL_SRC2      = 30,/* int32  same length as ELL_VAL, symmetric pair storage only: the mirror entry a_ji
                                   this slot also stands for (-1: the slot stands for one entry)                      */
    /* panel form: what pass 1 streams in place of PB_COL + PB_DST (derived from them; PB_DST stays the definition).
       Inside a 64-entry chunk the slots are runs, so 4 bytes per entry become two flag bits:                          */
    EHYB_ARR_PB_COLF       = 31,/* uint16 like PB_COL: bits 0-13 column, bit 15 = first entry of a piece (its partial),
                                   bit 14 = that piece's slot does not follow the previous piece's: a "jump" (the
                                   first entry of a chunk always is one)                                              */
    EHYB_ARR_PB_CHUNK      = 32,/* uint32 [chunks+1] index of the chunk's first jump in PB_JUMP; last = number of jumps */
    EHYB_ARR_PB_JUMP       = 33,/* uint32 per jump: its slot minus the pieces before it in its chunk (mod 2^32), so that
                                   slot(entry) = PB_JUMP[chunk's first + jumps up to the entry - 1] + pieces before the
                                   entry's; the padding piece of a panel's last chunk yields 0xFFFFFFFF             */
    EHYB_ARR_COL_SEG_FIRST = 34,/* int32  [segments+1] column segments of ehyb_plan_create_host_segs (empty: one segment)  */
    EHYB_ARR_PB_SEG_ITEM   = 35,/* int32  [segments+1] panel form: first pass-1 item of every column segment              */
    EHYB_ARR_PB_ITEMS1     = 36 /* int32  [2*items] {first unit, end unit}: the work of one pass-1 workgroup -- consecutive
                                   units of (nearly) equal total cost (entries streamed + panels staged)                */
};
/* Read-only view of one array of the plan's host layout.  A plan whose panel form was built on the device (ehyb_plan_create,
 * cfg.symbolic) downloads the PB_* streams on the first call that asks for one; it has no CSR form of that residual
 * (ER_SEG_*, ER_COL, ER_VAL empty, er_segments = 0). */
int ehyb_plan_host_array(const ehyb_plan* plan, int which, const void** ptr, int64_t* count);

/* ------------------------------------------------------------ multiply (GPU) */

/*
 * y = A*x on `stream` (a hipStream_t passed as void*; NULL = the null stream).
 * x and y are DEVICE pointers to full-length vectors in the permuted numbering
 * (x: n_cols doubles; y: rows [row_begin,row_end) of it are written).  Asynchronous.
 * One launch (ehyb_ell_kernel) when the residual is empty or tiny enough to ride inside
 * it (stats.er_inline > 0), else two: ehyb_ell_kernel then ehyb_er_kernel.
 * Replaces matrixVectorEHYB / matrixVectorEHYB_small (kernel.cu:490-552).
 * A plan multiplies ONE vector at a time: with the residual in panel form (stats.er_partials > 0) the
 * partial sums of a multiply live in a buffer of the plan, so two multiplies of the same plan must not
 * overlap on different streams (the reference is not re-entrant either: global device counters).
 */
int ehyb_spmv(ehyb_plan* plan, const double* x_dev, double* y_dev, void* stream);
/*
 * THE WALK.  Successive multiplies of a plan whose streams do not fit the 256 MB Infinity Cache walk them in ALTERNATING
 * directions (cfg.ell_alternate), so that a launch starts with what the one before it left in that cache: a loop of multiplies
 * -- a solver's -- runs about 10 % faster than the same launches all first to last (audikw_1-like: 76 against 83 us).  A single
 * multiply after other work gets the cold-cache time either way.  The direction is per-plan state, flipped atomically by every
 * launch: several host threads may multiply with ONE plan on their own streams and vectors at the same time where the plan
 * multiplies in ELL launches only (no panel-form residual: see above); with plain storage the result does not depend on the
 * direction bit for bit, with symmetric pair storage it differs by the rounding of the LDS adds' order, as between any two launches.
 * ehyb_spmv_walk states the direction per call: EHYB_WALK_AUTO = ehyb_spmv, _FIRST_TO_LAST / _LAST_TO_FIRST explicit (honoured
 * whatever cfg.ell_alternate says) -- what a caller that captures multiplies into its own hipGraph uses, because a captured launch
 * keeps the direction it was captured with: capture an even number of multiplies with alternating directions, or use
 * ehyb_spmv_graph_create, which does exactly that.
 */
enum { EHYB_WALK_AUTO = -1, EHYB_WALK_FIRST_TO_LAST = 0, EHYB_WALK_LAST_TO_FIRST = 1 };
int ehyb_spmv_walk(ehyb_plan* plan, const double* x_dev, double* y_dev, void* stream, int walk);
/*
 * `multiplies` back-to-back multiplies y = A x captured into a hipGraph with the directions alternating explicitly inside the run;
 * for an odd count (1: the solver that launches one multiply per iteration) TWO executables are captured, beginning first-to-last
 * and last-to-first, and ehyb_graph_launch replays them in turn -- the alternation survives the capture.  x and y are bound at
 * capture.  One launch at a time per graph object.
 */
typedef struct ehyb_graph ehyb_graph;
int ehyb_spmv_graph_create(ehyb_plan* plan, const double* x_dev, double* y_dev, int multiplies, ehyb_graph** graph);
int ehyb_graph_launch(ehyb_graph* graph, void* stream);
void ehyb_graph_destroy(ehyb_graph* graph);

/* phase 1: ELL part only (needs only window columns);  phase 2: residual only
 * (y += ...);  phase 0: both.  Lets a multi-GPU caller overlap the x exchange.
 * Phase 2 must follow phase 1 on the same stream before y is read: where partitions were given up to a
 * panel-form residual (their windows did not pay), phase 1 leaves their rows alone and phase 2 ASSIGNS them. */
int ehyb_spmv_phase(ehyb_plan* plan, const double* x_dev, double* y_dev, void* stream, int phase);

/*
 * Tuning on the device the plan lives on (optional; no reference counterpart).  The 8 XCDs of an MI355X do not stream at
 * the same rate -- on every box measured two of them finish a launch 8-12 % behind the fastest at equal bytes, and which
 * ones is a property of the box -- while the ELL launch of a plan with one resident round of workgroups lasts as long as
 * its LAST workgroup.  This call times a few stamped launches of the ELL kernel (per-workgroup clocks, XCC id), ranks the
 * XCDs by the rate they streamed at, and lets workgroup b take item item_map[b] with the heaviest items on the fastest XCD
 * and so on down; it keeps the new map only if the stamped launch got shorter.  *span_before_us / *span_after_us (may be
 * NULL): launch span before and with the map the plan ends with.  Synchronous (null stream); y is overwritten with A*x.
 * No-op for plans whose ELL launch needs more than one round of workgroups, for the direct shape and for plans without
 * an ELL launch.  spmvGPuEHYB does it during its warm-up; plan-API callers decide for themselves -- before they capture
 * multiplies of the plan into a hipGraph (a captured launch keeps the map it was captured with; replaced maps stay
 * allocated until the plan is destroyed, so such a graph stays valid).
 */
int ehyb_plan_tune(ehyb_plan* plan, const double* x_dev, double* y_dev, int reps, double* span_before_us, double* span_after_us);

/*
 * The multiply in PARTS, for a caller that receives x segment by segment (multi-GPU, ehyb_plan_create_host_segs):
 *   flags & EHYB_PART_FIRST   the ELL launch (window columns: all inside column segment 0) runs first;
 *   column segments [seg_begin, seg_end): pass 1 of the panel-form residual over the panels of those segments
 *                             (needs x of those columns only);
 *   flags & EHYB_PART_LAST    pass 2 of the panel form (every pass 1 must have been enqueued on `stream` before) --
 *                             or, for a plan whose residual is in CSR form, the residual launch (all of x).
 * ehyb_spmv == one call with every segment and both flags.  Parts of one multiply go to ONE stream, in order.
 */
enum { EHYB_PART_FIRST = 1, EHYB_PART_LAST = 2,
       EHYB_PART_LAST_FOREIGN = 4 /* panel form with cfg.row_split: pass 2 over the row blocks from row_split on only (they have entries in
                                     column segment 0 alone, so they are complete once segment 0's pass 1 is enqueued); with such a
                                     plan EHYB_PART_LAST closes the rows IN FRONT of row_split only: ehyb_spmv == one call with every
                                     segment and all three flags */ };
int ehyb_spmv_part(ehyb_plan* plan, const double* x_dev, double* y_dev, void* stream, int seg_begin, int seg_end, int flags);
/* Column segments of the plan (1 unless made by ehyb_plan_create_host_segs). */
int ehyb_plan_col_segs(const ehyb_plan* plan, int* n_col_segs);
/* dst[i] = src[idx[i]], i < n, on `stream`: packs the x entries the other ranks asked for (the send list of a halo
 * exchange) -- device pointers. */
int ehyb_gather(const double* src_dev, const int32_t* idx_dev, double* dst_dev, int64_t n, void* stream);
/* y[idx[i]] += src[i], i < n, on `stream` (fp64 atomics: an index may occur more than once): the partial sums other ranks
 * computed for this rank's rows, added in (exchange "cover") -- device pointers. */
int ehyb_scatter_add(double* y_dev, const int32_t* idx_dev, const double* src_dev, int64_t n, void* stream);
/*
 * Stream plumbing of one exchange step, so that the host issues ONE call per part instead of an event record, a stream
 * wait and a launch each (the host side of a step costs as much as the device side at 8 GPUs: DESIGN.md 5):
 *   ehyb_step_pack   on compute_stream: ehyb_gather into send_buf; then comm_stream waits for it (the collective that
 *                    the caller enqueues on comm_stream next reads send_buf);
 *   ehyb_step_part   wait_comm != 0: compute_stream first waits for everything enqueued on comm_stream so far (the
 *                    collective that delivers this part's columns); then ehyb_spmv_part on compute_stream.
 */
int ehyb_step_pack(const double* x_dev, const int32_t* idx_dev, double* send_buf_dev, int64_t n, void* compute_stream, void* comm_stream);
int ehyb_step_part(ehyb_plan* plan, const double* x_dev, double* y_dev, void* compute_stream, void* comm_stream, int wait_comm,
                   int seg_begin, int seg_end, int flags);
/*
 * The whole exchange step as ONE call, for a caller that issues its collectives from C (RCCL: ncclGroupStart / ncclSend /
 * ncclRecv / ncclGroupEnd) -- no reference counterpart.  The plan was made by ehyb_plan_create_host_segs with 1 + n_chunks
 * column segments (own columns, then one per chunk).  Sequence: ehyb_step_pack; own columns (EHYB_PART_FIRST); then for
 * k = 0 .. n_chunks-1: exchange(k, comm_stream, user) -- the caller ENQUEUES on comm_stream the collective that fills chunk k's
 * ghost columns of x_dev (0 = ok; anything else aborts the step with EHYB_ERR_STATE) -- and the panels of chunk k on
 * compute_stream behind it, the last one with EHYB_PART_LAST.  Everything asynchronous.
 */
typedef int (*ehyb_exchange_fn)(int chunk, void* comm_stream, void* user);
int ehyb_halo_step(ehyb_plan* plan, const double* x_dev, double* y_dev, const int32_t* send_idx_dev, double* send_buf_dev, int64_t n_send,
                   int n_chunks, ehyb_exchange_fn exchange, void* user, void* compute_stream, void* comm_stream);

/*
 * ---- RCCL-native exchange (multi-GPU, one process per GPU; no reference counterpart: kernel.h:12 is a commented-out mpi.h).
 * librccl is opened at run time (dlopen; inside a process that has loaded torch the soname resolves to torch's copy), so a
 * single-GPU caller needs no RCCL.  A communicator is made from a ncclUniqueId that rank 0 creates and the caller hands to
 * every rank (bench.py / dist.py: a torch.distributed broadcast of the 128 bytes); the device current at
 * ehyb_comm_create is the rank's GPU.  The communicator owns a high-priority stream its exchanges run on.
 */
#define EHYB_COMM_ID_BYTES 128
typedef struct ehyb_comm ehyb_comm;
int ehyb_rccl_version(int* version, char* where, int where_len);   /* ncclGetVersion + the file name dlopen found; EHYB_ERR_STATE: no librccl */
int ehyb_comm_unique_id(void* id_out /* EHYB_COMM_ID_BYTES */);     /* ncclGetUniqueId */
int ehyb_comm_create(const void* id, int rank, int world, ehyb_comm** comm);   /* ncclCommInitRank: collective over all ranks */
void ehyb_comm_destroy(ehyb_comm* comm);
int ehyb_comm_info(const ehyb_comm* comm, int* rank, int* world, void** stream);
/* sum of `count` doubles over the ranks, in place (the dot products of a distributed CG); stream NULL = the communicator's */
int ehyb_comm_allreduce_sum(ehyb_comm* comm, double* buf_dev, int64_t count, void* stream);
int ehyb_comm_allgather(ehyb_comm* comm, const double* send_dev, double* recv_dev, int64_t count_per_rank, void* stream);
/*
 * One rank's halo exchange + multiply as ONE host call per step (ehyb_halo_spmv), for a plan made by ehyb_plan_create_segs
 * with 1 + n_chunks column segments: [own columns | ghost columns of chunk 0 | chunk 1 | ...], inside a chunk the entries
 * of peer 0, peer 1, ... in rank order.
 *   send_idx_host   n_send own columns (plan order) the peers asked for, chunk by chunk, inside a chunk peer by peer;
 *   send_counts / recv_counts   [n_chunks * world]: doubles sent to / received from peer p in chunk k (entry k*world + p;
 *                   what rank a sends to b in chunk k must be what b receives from a -- the caller agrees on that when it
 *                   builds the lists, dist.py: RankLocalMatrix).  A rank may send to itself (world 1: the loop-back test).
 * ehyb_halo_spmv(h, x, y, stream): pack (one gather) on `stream`; on the communicator's stream, behind the pack, chunk
 * after chunk as a group of ncclSend / ncclRecv pairs straight into the ghost columns of x; on `stream` the own-column
 * part at once, then chunk k's panels as soon as chunk k has landed, the closing pass last -- the panels of chunk k multiply
 * while chunk k+1 is on the wire.  Plans that multiply in one launch exchange first, then multiply.  Asynchronous.
 * The plan, x and y must outlive the calls; one step at a time per halo object.
 */
typedef struct ehyb_halo ehyb_halo;
int ehyb_halo_create(ehyb_comm* comm, ehyb_plan* plan, int n_chunks, const int32_t* send_idx_host, int64_t n_send,
                     const int64_t* send_counts, const int64_t* recv_counts, ehyb_halo** halo);
void ehyb_halo_destroy(ehyb_halo* halo);
int ehyb_halo_spmv(ehyb_halo* halo, double* x_dev, double* y_dev, void* compute_stream);
/*
 * Exchange "cover" (dist.py: RankLocalMatrix(exchange="cover")): per pair of ranks only the hub columns of the block
 * A[r, s] travel as x entries; the rest of the block is handed to the columns' owner s at set-up, who multiplies it with its own x
 * and ships ONE PARTIAL SUM PER ROW -- a vertex cover of the block's bipartite graph instead of all its columns (R-MAT 2^24: 2.2-2.7 x
 * fewer doubles on the wire at 2 / 4 / 8 ranks, tools/dist_volume_model.py).  The plan of such a rank was built with cfg.row_split
 * = its number of own rows; the rows behind are its FOREIGN rows (ehyb_matrix_append_rows), grouped by destination rank:
 * ysend_counts[p] of them belong to rank p, yrecv_counts[p] partial sums arrive from rank p, partial i (arrival order, peer 0's first)
 * is added to row yrecv_idx_host[i].  ehyb_halo_spmv then: pack; the x chunks on the wire at once; segment 0 (own columns: own AND
 * foreign rows); the foreign rows closed (EHYB_PART_LAST_FOREIGN) and sent off; the chunks' panels as they land; the own rows closed;
 * the received partial sums added (ehyb_scatter_add).  y_dev holds own rows followed by the foreign rows.  Panel-form plans only.
 */
int ehyb_halo_set_partials(ehyb_halo* halo, int row_split, const int64_t* ysend_counts, const int64_t* yrecv_counts,
                           const int32_t* yrecv_idx_host, int64_t n_yrecv);
/* north_star's "all-gatherv of x" as one call per step: x = [own segment padded to seg_len | segment of rank 0 | ... | of rank
 * world-1]; ncclAllGather on the communicator's stream while the ELL phase runs, then the residual phase (dist.py: GatherSpmv) */
int ehyb_gather_spmv(ehyb_comm* comm, ehyb_plan* plan, double* x_dev, double* y_dev, int64_t seg_len, void* compute_stream);

/*
 * Timed loop on device-resident vectors: `warmup` untimed multiplies, then `iters`
 * multiplies bracketed by HIP events on `stream` (no copies inside, residual recomputed
 * every time: SURVEY 8d "Timing protocol").  ms_total = whole loop.  If ms_ell / ms_er
 * are non-NULL a second pass of min(iters, 200) multiplies brackets every launch with its
 * own event pair and returns the AVERAGE duration of the ELL launch and of the residual
 * launch (0-length interval when the residual is inline or empty).
 */
int ehyb_spmv_bench(ehyb_plan* plan, const double* x_dev, double* y_dev, void* stream,
                    int warmup, int iters, double* ms_total, double* ms_ell, double* ms_er);

/*
 * spmvGPuEHYB (spmv.h) with an explicit configuration and a status code; *ms_total (may be NULL)
 * receives the time of the MAXIter timed multiplies.  cfg == NULL is what spmvGPuEHYB itself does:
 * every knob at its default and the storage chosen from the matrix it is handed -- symmetric pair
 * storage iff the matrix has at least EHYB_SYM_MIN_ROWS rows, its partitions leave room for the y
 * accumulators (they do when matrixReorder made them) and a sample of its entries has bitwise equal
 * mirror images; no environment variable takes part.
 */
int spmvGPuEHYB_cfg(matrixCOO* localMatrix, const double* vectorIn, double* vectorOut, const int MAXIter,
                    int* realIter, const ehyb_config* cfg, double* ms_total);

/*
 * Numeric phase of the build on the GPU: new VALUES on the pattern the plan was built from.
 * The reference fills its ELL / ER value arrays on one host thread (convert.c:316-369, after the scatter of V
 * through the permutation in reordering.c:348-362) and repeats partition, conversion and upload for every
 * matrix.  A plan built with cfg.value_map = 1 keeps the slot maps of its value streams; this call gathers
 * `values` through them straight into the device arrays the kernels read (ehyb_fill_kernel: ELL stream,
 * residual segments or panel stream), padding = 0.0 -- no partitioner, no layout pass, no re-upload of the
 * index arrays.
 *   values       count doubles = the V array of a matrix with the SAME row-grouped pattern: in the order of the
 *                reordered matrix the plan was built from (entry_order NULL), or in the caller's order BEFORE
 *                ehyb_matrix_reorder with entry_order[k] = place of reordered entry k in `values`
 *                (ehyb_entry_order computes it).
 *   on_device    0: values / entry_order are host arrays (uploaded for the call);  1: both are device pointers.
 * With symmetric pair storage a slot stands for a_ij AND a_ji: the new values must be equal there (a_ij == a_ji, the test the builder paired them with); they
 * are checked on the device first and the plan is left untouched (EHYB_ERR_ARG) if any pair differs.
 * After a successful call the HOST copy of the value arrays (ehyb_plan_host_array, ehyb_plan_save) is stale:
 * ehyb_plan_save refuses such a plan.  Synchronous with respect to `stream` when on_device = 0.
 */
int ehyb_plan_set_values(ehyb_plan* plan, const double* values, int64_t count, const int32_t* entry_order,
                         int on_device, void* stream);
/* entry_order[k_new] = k_old: where entry k_new of the matrix AFTER ehyb_matrix_reorder sat in the caller's
 * row-grouped arrays before it (rows move as wholes and keep their stored order).  row_idx_before = the rowIdx of
 * the matrix before the call (dimension+1 ints), reorder_list as the call left it. */
int ehyb_entry_order(int dimension, const int* row_idx_before, const int* reorder_list, int32_t* entry_order);

/* Convenience: host vectors in, host vector out (H2D, `iters` multiplies, D2H). */
int ehyb_spmv_host(ehyb_plan* plan, const double* x_host, double* y_host, int iters);

/* Device plumbing for callers without their own allocator (tests, the CLI). */
int ehyb_device_count(int* count);
int ehyb_device_set(int device);
int ehyb_device_name(char* buf, int len);
int ehyb_dev_alloc(size_t bytes, void** ptr);
int ehyb_dev_free(void* ptr);
int ehyb_h2d(void* dst_dev, const void* src_host, size_t bytes);
int ehyb_d2h(void* dst_host, const void* src_dev, size_t bytes);
int ehyb_dev_sync(void);
/* a non-blocking stream of the current device (hipStream_t as void*), for callers without a HIP binding of their own */
int ehyb_stream_create(void** stream);
int ehyb_stream_destroy(void* stream);
int ehyb_stream_sync(void* stream);
/* free and total device memory in bytes (hipMemGetInfo): what a harness checks a plan's life cycle against */
int ehyb_dev_mem_info(size_t* free_bytes, size_t* total_bytes);
/* Streaming-read ceiling of this device: sums `bytes` of doubles `iters` times, returns GB/s. */
int ehyb_measure_read_bw(size_t bytes, int iters, double* gbps);

/* ------------------------------------------------ iterative caller */

/*
 * Unpreconditioned conjugate gradients for A x = b on the plan's (symmetric positive definite)
 * matrix, entirely on the device: x_dev starts as the initial guess and ends as the solution,
 * both in the permuted numbering.  This is the solver the reference was cut down from (its
 * leftovers: kernelMyxpy y = x + gamma*y, kernel.cu:288-296; initialize_all, kernel.cu:20-31;
 * the PRECOND/FACT switches of cb_s) and the caller for which "x changes every multiply" matters.
 * One ehyb_spmv plus three fused vector kernels per iteration; dot products stay on the device
 * (per-workgroup partial sums added in a fixed order, no atomics), two iterations are replayed
 * from one hipGraph, and the host looks at the residual every `check_every` iterations (<= 0: 10;
 * odd values are rounded up to even).
 * Stops when ||r|| <= rtol * ||b|| or after max_iter iterations.  Outputs may be NULL.
 * Needs a plan over all rows (single GPU).  stream NULL: a private stream, synchronised with the
 * default stream on entry and finished on return.
 */
int ehyb_cg(ehyb_plan* plan, const double* b_dev, double* x_dev, int max_iter, double rtol,
            int check_every, void* stream, int* iters_done, double* rel_residual);
/* The same with the diagonal (Jacobi) preconditioner -- the PRECOND switch of the reference's cb_s
 * (spmv.h:7-15): inv_diag_dev[i] = 1 / a_ii in the permuted numbering (matrixCOO.diag, permuted
 * like x); NULL = ehyb_cg.  z = M^-1 r is recomputed on the fly, no extra vector is stored. */
int ehyb_pcg(ehyb_plan* plan, const double* inv_diag_dev, const double* b_dev, double* x_dev, int max_iter,
             double rtol, int check_every, void* stream, int* iters_done, double* rel_residual);

/*
 * The vector kernels of ehyb_pcg as building blocks for a caller that owns the loop -- the multi-GPU CG of
 * ehyb_spmv_gpu_amd/dist.py (HaloCG), where q = A p goes through the halo exchange.  s is the partial-sum
 * array: `slots` slots of `slot_doubles` doubles (ehyb_cg_layout).  Every rank launches the same grid, so an
 * all-reduce (sum) of a slot makes every rank's partials the element-wise global ones and the kernel that
 * needs the scalar adds them up as in the single-GPU solve.  cur (0/1) says which of the two r.z slots
 * holds the current r.z (number c lives in slot rz0 + 2 c, r.r in the slot between them, so what an
 * iteration writes is one contiguous range): it alternates from iteration to iteration.  All
 * asynchronous on `stream`.
 */
int ehyb_cg_layout(int* slots, int* slot_doubles, int* slot_bb, int* slot_pq, int* slot_rr, int* slot_rz0);
/* r = b - q, p = M^-1 r; partials of r.z (slot rz0), r.r, b.b */
int ehyb_cg_init_step(int n, const double* b_dev, const double* q_dev, const double* inv_diag_dev, double* r_dev, double* p_dev,
                      double* s_dev, void* stream);
/* partials of p.q */
int ehyb_cg_dot_step(int n, const double* p_dev, const double* q_dev, double* s_dev, void* stream);
/* alpha = r.z[cur] / p.q;  x += alpha p;  r -= alpha q;  partials of the new r.z (number cur ^ 1) and r.r */
int ehyb_cg_update_step(int n, const double* p_dev, const double* q_dev, const double* inv_diag_dev, double* x_dev, double* r_dev,
                        double* s_dev, int cur, void* stream);
/* beta = r.z[cur ^ 1] / r.z[cur];  p = M^-1 r + beta p   (the reference's kernelMyxpy, kernel.cu:288-296) */
int ehyb_cg_direction_step(int n, const double* r_dev, const double* inv_diag_dev, double* p_dev, const double* s_dev, int cur,
                           void* stream);

/* -------------------------------------------- harness pieces (solver_test.c) */

/*
 * Read ./path as Matrix Market coordinate real/integer/pattern, general or symmetric,
 * into a row-grouped COO exactly as matrixRead_unsym (solver_test.c:31-126: file order
 * kept) or matrixRead_sym (solver_test.c:127-265: lower triangle mirrored, totalNum =
 * 2*stored - dimension) build it, including rowIdx/numInRow/maxCol.  nParts,
 * vectorCacheSize and kernelPerPart are filled by ehyb_sizing.  pattern entries get 1.0.
 * All arrays are malloc()ed; release with ehyb_matrix_free.
 */
int ehyb_mm_read(const char* path, const ehyb_config* cfg, matrixCOO* out, int* is_symmetric);
int ehyb_mm_write(const char* path, const matrixCOO* m, int symmetric_lower_only);
/* Copy a CSR matrix (rowptr[n+1], cols, vals; 0-based) into a freshly malloc()ed row-grouped
 * matrixCOO, entry order preserved; sizing fields filled by ehyb_sizing. */
int ehyb_matrix_from_csr(int n, const int64_t* rowptr, const int* cols, const double* vals,
                         const ehyb_config* cfg, matrixCOO* out);
void ehyb_matrix_free(matrixCOO* m);
/*
 * Multi-GPU, rank-local build (no reference counterpart -- the reference is single-GPU).
 * m is a rank's square diagonal block after ehyb_matrix_reorder.  It grows n_ghost columns, one
 * per x entry the rank receives from other ranks, and the nnz_g coupling entries (gi[k] = row in
 * m's current numbering, gj[k] in [0, n_ghost), gv[k]); the n_ghost new rows stay empty and the
 * partition data is kept.  A plan over rows [0, old dimension) with cfg.n_top > 1 then multiplies
 * the ghost columns in phase 2 from x = [local x | receive buffer].
 */
int ehyb_matrix_append_ghosts(matrixCOO* m, int n_ghost, int64_t nnz_g,
                              const int* gi, const int* gj, const double* gv);
/*
 * The other direction (exchange "cover"): rows [row0, row0 + n_rows) of m -- empty so far, at or behind the last partition
 * boundary, inside the dimension ehyb_matrix_append_ghosts gave the matrix -- receive nnz entries (ri[k] in [0, n_rows) ascending,
 * cj[k] any column of m, v[k]): the FOREIGN rows of a rank, whose entries another rank handed over because shipping one partial sum
 * per row is cheaper than shipping the x entries of their columns.  They form new partitions of at most rows_per_part rows behind
 * the existing ones (nParts and partBoundary grow; rows_per_part <= 0: m->vectorCacheSize).  A plan over rows [0, row0 + n_rows) with
 * cfg.row_split = row0 then multiplies them with everything else.
 */
int ehyb_matrix_append_rows(matrixCOO* m, int row0, int n_rows, int64_t nnz, const int* ri, const int* cj, const double* v, int rows_per_part);

/* x[i]: srand(i); (rand()%200-100)/1000.0  -- solver_test.c:89-92, 228-231 (glibc rand). */
void ehyb_x_glibc(int n, double* x);

/*
 * Synthetic matrices (no .mtx files exist offline).  All deterministic,
 * integer-hash values v = ((h mod 200) - 100)/1000 with 0 replaced by 0.001.
 * Each fills a row-grouped, column-sorted matrixCOO like ehyb_mm_read does.
 */
/* config 3: block-circulant band, `band` entries per row inside blocks of `block` rows */
int ehyb_gen_banded(int n, int band, int block, const ehyb_config* cfg, matrixCOO* out);
/* audikw_1-like: nodes of a 3-D grid (truncated to n/dof nodes), dof unknowns per node,
 * 27-point node coupling plus hashed second-shell couplings with probability extra_ppm/1e6,
 * symmetric values; scramble != 0 applies a random relabelling of the nodes.            */
int ehyb_gen_fem3d(int n, int dof, int nx, int ny, int extra_ppm, int scramble, uint64_t seed,
                   const ehyb_config* cfg, matrixCOO* out);
/* audikw_1-like with a graded mesh: a smooth density field over the grid decides how many couplings a node
 * keeps (first shell: probability near_min + (1 - near_min) * g; second shell: far_max * g^3; g in [0,1] at the
 * sparser end of the coupling; both in parts per million), so row lengths spread like those of an
 * unstructured mesh -- (250000, 900000) with 3 unknowns per node: about 21 to 345 entries per row. */
int ehyb_gen_fem3d_graded(int n, int dof, int nx, int ny, int near_min_ppm, int far_max_ppm, int scramble, uint64_t seed,
                          const ehyb_config* cfg, matrixCOO* out);
/* The rows of block `block` of n_blocks such grids stacked along z (weak scaling: one block per
 * GPU, each rank generates only its own rows): dimension n*n_blocks, rows outside
 * [block*n, (block+1)*n) empty, every block labelled (scrambled) on its own, couplings reach two
 * grid layers into the neighbouring blocks.  (block, n_blocks) = (0, 1) is ehyb_gen_fem3d. */
int ehyb_gen_fem3d_block(int n, int dof, int nx, int ny, int extra_ppm, int scramble, uint64_t seed,
                         int block, int n_blocks, const ehyb_config* cfg, matrixCOO* out);
/* R-MAT (a,b,c,d)=(.57,.19,.19,.05), 2^scale rows, `edges` samples, duplicates merged */
int ehyb_gen_rmat(int scale, int64_t edges, uint64_t seed, const ehyb_config* cfg, matrixCOO* out);
/* The rows of row block `block` of n_blocks of that same matrix, for a process that owns one block (strong
 * scaling, one process per GPU): contiguous row ranges of about equal cost (edge samples + 2 per row); cuts
 * (n_blocks+1 ints, out) are the same on every process.  Dimension 2^scale, rows outside the block empty. */
int ehyb_gen_rmat_block(int scale, int64_t edges, uint64_t seed, int block, int n_blocks, int* cuts, const ehyb_config* cfg,
                        matrixCOO* out);
/* The same with the cost the cuts balance named: 0 = ehyb_gen_rmat_block's (a row costs its samples + 2); 1 = the cost under the "cover"
 * exchange (ehyb_halo_set_partials), where an entry is multiplied by the owner of its row if its column has the higher degree of the two
 * (the column's x entry travels) and by the owner of its column otherwise (a partial sum travels back): hub rows then cost their owner
 * little and the blocks of hub rows get more rows (work max / mean at 8 ranks 1.24 -> 1.05). */
int ehyb_gen_rmat_block_cost(int scale, int64_t edges, uint64_t seed, int block, int n_blocks, int cost_model, int* cuts, const ehyb_config* cfg,
                             matrixCOO* out);
/* The rows [row0, row1) of that same matrix, for a process whose row range was decided elsewhere (bench.py re-cuts the ranks' rows once
 * the cost of the "cover" exchange is known).  Dimension 2^scale, rows outside the range empty. */
int ehyb_gen_rmat_rows(int scale, int64_t edges, uint64_t seed, int row0, int row1, const ehyb_config* cfg, matrixCOO* out);
/* 2-D 5/9-point stencil plus `extra` random symmetric couplings (small test inputs) */
int ehyb_gen_stencil2d(int nx, int ny, int points, int extra, uint64_t seed,
                       const ehyb_config* cfg, matrixCOO* out);
/* unstructured-mesh stand-in: ceil(n/dof) random points in the unit cube (x, y = u^(grade_permille/1000): denser towards
 * one corner; 0 = uniform), every node coupled to its knn nearest neighbours, made symmetric, dof unknowns per node;
 * random labels, row lengths between (knn+1)*dof and ~2*knn*dof -- a check that thresholds fitted on lattices hold */
int ehyb_gen_mesh3d(int n, int dof, int knn, int grade_permille, uint64_t seed, const ehyb_config* cfg, matrixCOO* out);
/* nlpkkt-like 3-D KKT system: [H A^T; A 0] on an nx^3 grid */
int ehyb_gen_kkt3d(int nx, const ehyb_config* cfg, matrixCOO* out);

#ifdef __cplusplus
}
#endif
#endif /* EHYB_H */
