// On-disk cache of a plan: the permutation and the finished EHYB layout in one file, so that the
// partitioner and the layout builder run once per matrix.  The reference redoes
// mt-metis + COO2EHYB on every run (solver_test.c:369-382, spmv.cu:74); on the bench matrix that
// is seconds of host work in front of a 0.14 ms multiply.
//
// File: "EHYBPLN4", key, resolved Config, layout scalars, stats, then every array as
// {u64 count, bytes}, then "EHYBEND4".  Native byte order, same-machine cache -- not an
// interchange format.  A file whose magic, version, key or sizes do not fit is rejected.
#include "ehyb_internal.h"

#include <cstdio>
#include <memory>
#include <new>
#include <type_traits>

using namespace ehyb;

namespace {

const char kMagic[8] = {'E', 'H', 'Y', 'B', 'P', 'L', 'N', '9'};
const char kEnd[8] = {'E', 'H', 'Y', 'B', 'E', 'N', 'D', '4'};

struct FileCloser {
    void operator()(FILE* f) const
    {
        if (f) fclose(f);
    }
};
using File = std::unique_ptr<FILE, FileCloser>;

template <class T>
bool put(FILE* f, const T& v)
{
    static_assert(std::is_trivially_copyable<T>::value, "raw write");
    return fwrite(&v, sizeof(T), 1, f) == 1;
}
template <class T>
bool get(FILE* f, T& v)
{
    return fread(&v, sizeof(T), 1, f) == 1;
}
template <class T, class A>
bool put_vec(FILE* f, const std::vector<T, A>& v)
{
    const uint64_t n = v.size();
    return put(f, n) && (n == 0 || fwrite(v.data(), sizeof(T), n, f) == n);
}
template <class T, class A>
bool get_vec(FILE* f, std::vector<T, A>& v, uint64_t limit)
{
    uint64_t n = 0;
    if (!get(f, n) || n > limit) return false;
    v.resize(n);
    return n == 0 || fread(v.data(), sizeof(T), n, f) == n;
}

// Every array of the layout, in file order.
template <class F>
bool each_array(HostLayout& H, F&& io)
{
    return io(H.part_boundary) && io(H.win_len) && io(H.halo_ptr) && io(H.halo_cols) && io(H.slab_pair_ptr) &&
           io(H.slab_row) && io(H.slab_part) && io(H.ell_val) && io(H.ell_col) && io(H.slab_col_ptr) && io(H.lane_group) &&
           io(H.slab_meta) && io(H.items) && io(H.segs) && io(H.er_seg_ptr) && io(H.er_seg_row) && io(H.er_col) &&
           io(H.er_val) && io(H.er_blocks) && io(H.slab_lrow) && io(H.pb_val) && io(H.pb_col) && io(H.pb_dst) &&
           io(H.pb_units1) && io(H.pb_row) && io(H.pb_units2) && io(H.col_seg_first) && io(H.pb_seg_item) && io(H.pb_items1);
}

struct Scalars {
    int32_t n_cols, row_begin, row_end, n_parts, lds_doubles, inline_er;
    int32_t er_bins[8];
    int32_t sym, yacc_doubles;
    int32_t er_panel, pb_panel_cols, pb_rows_max, direct;
    int64_t pb_partials, pb_bytes;
    int32_t pb_assign, reserved;
};

// The panel residual's arrays must fit together as the two kernels index them.
bool panel_consistent(const HostLayout& H)
{
    if (!H.er_panel)
        return H.pb_val.empty() && H.pb_col.empty() && H.pb_dst.empty() && H.pb_units1.empty() && H.pb_row.empty() && H.pb_units2.empty();
    const size_t ne = H.pb_val.size();
    if (H.pb_col.size() != ne || H.pb_dst.size() != ne || ne % 64 != 0 || H.pb_units1.size() % 4 || H.pb_units2.size() % 4) return false;
    if (H.pb_partials < 0 || H.pb_row.size() != (size_t)H.pb_partials) return false;
    if (H.pb_panel_cols < 64 || H.pb_panel_cols > 16384 || H.pb_rows_max < 1 || H.pb_rows_max > 16384) return false;
    if (H.pb_items1.size() % 2) return false;
    for (size_t i = 0; i < H.pb_items1.size(); i += 2)  // the items are consecutive runs of the unit list, all of it
        if (H.pb_items1[i] != (i ? H.pb_items1[i - 1] : 0) || H.pb_items1[i + 1] <= H.pb_items1[i] || (size_t)H.pb_items1[i + 1] > H.pb_units1.size() / 4) return false;
    if (!H.pb_items1.empty() && (size_t)H.pb_items1.back() != H.pb_units1.size() / 4) return false;
    if (H.pb_items1.empty() != H.pb_units1.empty()) return false;
    if (!H.pb_seg_item.empty() && (H.pb_seg_item.size() != H.col_seg_first.size() + (H.col_seg_first.empty() ? 2 : 0) || H.pb_seg_item.front() != 0 || (size_t)H.pb_seg_item.back() != H.pb_items1.size() / 2)) return false;
    for (size_t u = 0; u < H.pb_units1.size(); u += 4) {
        const int32_t* q = &H.pb_units1[u];
        if (q[0] < 0 || q[1] < 1 || q[1] > H.pb_panel_cols || (int64_t)q[0] + q[1] > H.n_cols || q[2] < 0 || q[3] < q[2] || (size_t)q[3] > ne ||
            q[2] % 64 || q[3] % 64)
            return false;
    }
    for (size_t u = 0; u < H.pb_units2.size(); u += 4) {
        const int32_t* q = &H.pb_units2[u];
        const int32_t rows = q[3] < 0 ? -q[3] : q[3];  // negative: the block assigns y (pb_assign)
        if (q[0] < 0 || q[1] < q[0] || q[1] > H.pb_partials || q[2] < H.row_begin || rows < 1 || rows > H.pb_rows_max || q[2] + rows > H.row_end) return false;
        if (q[3] < 0 && !H.pb_assign) return false;
    }
    for (uint32_t d : H.pb_dst)
        if (d != 0xFFFFFFFFu && d >= (uint64_t)H.pb_partials) return false;
    for (uint16_t c : H.pb_col)
        if (c >= H.pb_panel_cols) return false;  // bits 14 and 15 of the streamed column word are flags
    return true;
}

}  // namespace

extern "C" {

// Order-sensitive 64-bit digest of the matrix a plan was built from (dimension, entry count,
// coordinates and values as stored).  0 is never returned: it means "do not check" to ehyb_plan_load.
uint64_t ehyb_matrix_key(const matrixCOO* m)
{
    if (!m) return 1;
    uint64_t h = mix64(0x45485942ull ^ (uint64_t)(uint32_t)m->dimension) ^ mix64((uint64_t)(uint32_t)m->totalNum + 0x9E37ull);
    const int64_t nnz = m->totalNum;
    if (m->I && m->J && m->V) {
        // four independent lanes, so the loop is not one long multiply chain
        uint64_t a[4] = {h, h ^ 0x1111, h ^ 0x2222, h ^ 0x3333};
        for (int64_t k = 0; k < nnz; ++k) {
            uint64_t bits;
            memcpy(&bits, &m->V[k], 8);
            const uint64_t w = ((uint64_t)(uint32_t)m->I[k] << 32 | (uint32_t)m->J[k]) ^ (bits * 0x9E3779B97F4A7C15ull);
            uint64_t& s = a[k & 3];
            s = (s ^ w) * 0xBF58476D1CE4E5B9ull;
            s ^= s >> 29;
        }
        h = mix64(a[0]) ^ mix64(a[1] + 1) ^ mix64(a[2] + 2) ^ mix64(a[3] + 3);
    }
    return h ? h : 1;
}

int ehyb_plan_save(const ehyb_plan* plan, const int* reorder_list, uint64_t matrix_key, const char* path)
{
    clear_error();
    if (!plan || !path) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_save: null argument");
    if (plan->host_values_stale)
        EHYB_FAIL(EHYB_ERR_STATE, "ehyb_plan_save: the plan's values were replaced on the device (ehyb_plan_set_values); its host copy is stale");
    if (plan->host.pb_host_missing) {  // a panel form built on the device: its streams come back first
        const int rc = materialize_panel_host(const_cast<ehyb_plan*>(plan));
        if (rc != EHYB_OK) return rc;
    }
    HostLayout& H = const_cast<HostLayout&>(plan->host);  // each_array takes non-const; nothing is modified
    File f(fopen(path, "wb"));
    if (!f) EHYB_FAIL(EHYB_ERR_IO, "ehyb_plan_save: cannot create %s", path);
    Scalars s{H.n_cols, H.row_begin, H.row_end, H.n_parts, H.lds_doubles, H.inline_er ? 1 : 0, {0}, H.sym ? 1 : 0, H.yacc_doubles,
              H.er_panel ? 1 : 0, H.pb_panel_cols, H.pb_rows_max, H.direct ? 1 : 0, H.pb_partials, H.pb_bytes, H.pb_assign ? 1 : 0, 0};
    memcpy(s.er_bins, H.er_bins, sizeof s.er_bins);
    std::vector<int32_t> perm;
    if (reorder_list) perm.assign(reorder_list, reorder_list + H.n_cols);
    bool ok = fwrite(kMagic, 8, 1, f.get()) == 1 && put(f.get(), matrix_key) && put(f.get(), plan->cfg) && put(f.get(), s) &&
              put(f.get(), H.stats) && put_vec(f.get(), perm) &&
              each_array(H, [&](auto& v) { return put_vec(f.get(), v); }) && fwrite(kEnd, 8, 1, f.get()) == 1;
    FILE* raw = f.release();
    ok = (fclose(raw) == 0) && ok;
    if (!ok) {
        remove(path);
        EHYB_FAIL(EHYB_ERR_IO, "ehyb_plan_save: write to %s failed", path);
    }
    return EHYB_OK;
}

int ehyb_plan_load(const char* path, uint64_t expect_key, ehyb_plan** plan, int* reorder_list)
{
    clear_error();
    if (!path || !plan) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_plan_load: null argument");
    *plan = nullptr;
    File f(fopen(path, "rb"));
    if (!f) EHYB_FAIL(EHYB_ERR_IO, "ehyb_plan_load: cannot open %s", path);
    char magic[8];
    uint64_t key = 0;
    if (fread(magic, 8, 1, f.get()) != 1 || memcmp(magic, kMagic, 8) != 0 || !get(f.get(), key))
        EHYB_FAIL(EHYB_ERR_FORMAT, "ehyb_plan_load: %s is not a plan file of this version", path);
    if (expect_key != 0 && key != expect_key)
        EHYB_FAIL(EHYB_ERR_FORMAT, "ehyb_plan_load: %s was built from another matrix (key %016llx, expected %016llx)", path,
                  (unsigned long long)key, (unsigned long long)expect_key);
    std::unique_ptr<ehyb_plan> P(new (std::nothrow) ehyb_plan());
    if (!P) EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_plan_load: out of memory");
    Scalars s;
    HostLayout& H = P->host;
    std::vector<int32_t> perm;
    const uint64_t limit = 1ull << 36;  // sanity bound for any array count
    try {
        bool ok = get(f.get(), P->cfg) && get(f.get(), s) && get(f.get(), H.stats) && get_vec(f.get(), perm, limit) &&
                  each_array(H, [&](auto& v) { return get_vec(f.get(), v, limit); });
        char end[8];
        ok = ok && fread(end, 8, 1, f.get()) == 1 && memcmp(end, kEnd, 8) == 0;
        if (!ok) EHYB_FAIL(EHYB_ERR_FORMAT, "ehyb_plan_load: %s is truncated or damaged", path);
    } catch (const std::bad_alloc&) {
        EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_plan_load: out of memory reading %s", path);
    }
    H.n_cols = s.n_cols, H.row_begin = s.row_begin, H.row_end = s.row_end, H.n_parts = s.n_parts;
    H.lds_doubles = s.lds_doubles, H.inline_er = s.inline_er != 0;
    H.sym = s.sym != 0, H.yacc_doubles = s.yacc_doubles;
    H.er_panel = s.er_panel != 0, H.pb_panel_cols = s.pb_panel_cols, H.pb_rows_max = s.pb_rows_max, H.direct = s.direct != 0;
    H.pb_partials = s.pb_partials, H.pb_bytes = s.pb_bytes;
    H.pb_padded = (int64_t)H.pb_val.size();
    H.pb_assign = s.pb_assign != 0 && H.er_panel;
    memcpy(H.er_bins, s.er_bins, sizeof s.er_bins);
    // the sizes the kernels rely on must fit together -- a damaged file must not reach the GPU
    const size_t nslab = H.slab_row.size(), nseg = H.er_seg_row.size();
    const bool consistent =
        H.n_cols > 0 && H.row_begin >= 0 && H.row_end <= H.n_cols && H.row_begin < H.row_end && H.n_parts >= 1 &&
        H.part_boundary.size() == (size_t)H.n_parts + 1 && H.win_len.size() == (size_t)H.n_parts &&
        H.halo_ptr.size() == (size_t)H.n_parts + 1 && H.halo_cols.size() == (size_t)H.halo_ptr.back() &&
        H.slab_pair_ptr.size() == nslab + 1 && H.slab_col_ptr.size() == nslab + 1 && H.slab_part.size() == nslab &&
        H.slab_meta.size() == nslab * 4 && H.lane_group.size() == nslab * kSlabRows &&
        H.ell_val.size() == (size_t)H.slab_pair_ptr.back() * 2 * kSlabRows && H.ell_col.size() == (size_t)H.slab_col_ptr.back() &&
        H.items.size() % 8 == 0 && H.segs.size() % 8 == 0 && H.er_seg_ptr.size() == nseg + 1 &&
        H.er_col.size() == (size_t)H.er_seg_ptr.back() && H.er_val.size() == H.er_col.size() && H.er_blocks.size() % 4 == 0 && (H.sym ? H.slab_lrow.size() == nslab * kSlabRows : H.slab_lrow.empty()) &&
        H.lds_doubles > 0 && H.lds_doubles <= EHYB_LDS_MAX_DOUBLES && (perm.empty() || perm.size() == (size_t)H.n_cols) &&
        panel_consistent(H);
    if (!consistent) EHYB_FAIL(EHYB_ERR_FORMAT, "ehyb_plan_load: %s holds inconsistent array sizes", path);
    if (H.er_panel) encode_panel_slots(&H);  // what pass 1 streams is derived from pb_col + pb_dst, not stored
    P->cfg.value_map = 0;  // the slot maps are not part of the file: a loaded plan cannot be refilled
    if (reorder_list) {
        if (perm.empty()) EHYB_FAIL(EHYB_ERR_FORMAT, "ehyb_plan_load: %s holds no permutation", path);
        std::copy(perm.begin(), perm.end(), reorder_list);
    }
    P->perm.swap(perm);  // also readable through ehyb_plan_host_array(EHYB_ARR_PERM)
    *plan = P.release();
    return EHYB_OK;
}

}  // extern "C"
