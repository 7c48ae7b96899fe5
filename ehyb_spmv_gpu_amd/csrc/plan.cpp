// Host half of the plan object: layout construction and read-only views (no HIP calls).
#include "ehyb_internal.h"

#include <new>

using namespace ehyb;

extern "C" {

int ehyb_plan_create_host(const matrixCOO* m, int row_begin, int row_end, const ehyb_config* cfg,
                          ehyb_plan** plan)
{
    clear_error();
    if (!plan) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host: null output");
    *plan = nullptr;
    if (!m) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_create_host: null matrix");
    ehyb_plan* P = new (std::nothrow) ehyb_plan();
    if (!P) EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_plan_create_host: out of memory");
    P->cfg = resolve_config(cfg);
    int rc;
    try {
        rc = build_layout(m, row_begin, row_end, P->cfg, &P->host);
    } catch (const std::bad_alloc&) {
        set_error("ehyb_plan_create_host: out of memory while building the layout");
        rc = EHYB_ERR_ALLOC;
    }
    if (rc != EHYB_OK) {
        delete P;
        return rc;
    }
    *plan = P;
    return EHYB_OK;
}

int ehyb_plan_stats(const ehyb_plan* plan, ehyb_stats* out)
{
    if (!plan || !out) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_stats: null argument");
    *out = plan->host.stats;
    return EHYB_OK;
}

int ehyb_plan_host_array(const ehyb_plan* plan, int which, const void** ptr, int64_t* count)
{
    if (!plan || !ptr || !count) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_host_array: null argument");
    const HostLayout& H = plan->host;
#define VIEW(v)                      \
    *ptr = (const void*)(v).data(); \
    *count = (int64_t)(v).size();   \
    return EHYB_OK
    switch (which) {
        case EHYB_ARR_PART_BOUNDARY: VIEW(H.part_boundary);
        case EHYB_ARR_WIN_LEN: VIEW(H.win_len);
        case EHYB_ARR_HALO_PTR: VIEW(H.halo_ptr);
        case EHYB_ARR_HALO_COLS: VIEW(H.halo_cols);
        case EHYB_ARR_SLAB_PAIR_PTR: VIEW(H.slab_pair_ptr);
        case EHYB_ARR_SLAB_ROW: VIEW(H.slab_row);
        case EHYB_ARR_SLAB_PART: VIEW(H.slab_part);
        case EHYB_ARR_ELL_VAL: VIEW(H.ell_val);
        case EHYB_ARR_ELL_COL: VIEW(H.ell_col);
        case EHYB_ARR_ITEMS: VIEW(H.items);
        case EHYB_ARR_ER_SEG_PTR: VIEW(H.er_seg_ptr);
        case EHYB_ARR_ER_SEG_ROW: VIEW(H.er_seg_row);
        case EHYB_ARR_ER_COL: VIEW(H.er_col);
        case EHYB_ARR_ER_VAL: VIEW(H.er_val);
        case EHYB_ARR_SLAB_COL_PTR: VIEW(H.slab_col_ptr);
        case EHYB_ARR_LANE_GROUP: VIEW(H.lane_group);
        case EHYB_ARR_SLAB_META: VIEW(H.slab_meta);
        case EHYB_ARR_SEGS: VIEW(H.segs);
        case EHYB_ARR_PERM: VIEW(plan->perm);
        case EHYB_ARR_SLAB_LROW: VIEW(H.slab_lrow);
        case EHYB_ARR_ER_BINS:
            *ptr = (const void*)H.er_bins;
            *count = 8;
            return EHYB_OK;
        default: break;
    }
#undef VIEW
    EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_host_array: unknown array id %d", which);
}

}  // extern "C"
