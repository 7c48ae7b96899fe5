// RCCL-native exchange step of the multi-GPU multiply (the reference has nothing here: its kernel.h:12 is a
// commented-out mpi.h).  One process per GPU; a communicator made from a ncclUniqueId the caller distributes (bench.py /
// dist.py broadcast it through torch.distributed, a C caller through whatever it has); every exchange is a group of
// ncclSend / ncclRecv pairs on the communicator's own stream -- every pair of GPUs of an MI355X node has its own xGMI link,
// so the direct all-to-all is link-optimal -- and the WHOLE step (pack, own columns, chunk k on the wire while chunk k-1's
// panels multiply, closing pass) is ONE host call, ehyb_halo_spmv: the host side of a step issued call by call from Python
// cost as much as the device side.
//
// librccl is opened at run time (dlopen), never linked: a single-GPU caller needs no RCCL at all, and inside a Python
// process that has imported torch the soname resolves to the copy torch already loaded -- one RCCL per process.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <dlfcn.h>

#include <mutex>
#include <vector>

#include "ehyb_internal.h"

using namespace ehyb;

#define HIP_TRY(expr)                                                                 \
    do {                                                                              \
        hipError_t e_ = (expr);                                                       \
        if (e_ != hipSuccess) EHYB_FAIL(EHYB_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

namespace {

struct Rccl {
    void* handle = nullptr;
    decltype(&ncclGetVersion) GetVersion = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    decltype(&ncclSend) Send = nullptr;
    decltype(&ncclRecv) Recv = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    std::string where;
};

Rccl g_rccl;
std::once_flag g_rccl_once;
std::string g_rccl_error;

void load_rccl()
{
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1", "/opt/rocm/lib/librccl.so"};
    void* h = nullptr;
    for (const char* n : names) {
        h = dlopen(n, RTLD_NOW | RTLD_LOCAL);
        if (h) {
            g_rccl.where = n;
            break;
        }
    }
    if (!h) {
        g_rccl_error = std::string("librccl not found (dlopen: ") + (dlerror() ? dlerror() : "?") + ")";
        return;
    }
#define SYM(field, name)                                                       \
    g_rccl.field = (decltype(g_rccl.field))dlsym(h, name);                     \
    if (!g_rccl.field) {                                                       \
        g_rccl_error = std::string("librccl has no symbol ") + name;           \
        return;                                                                \
    }
    SYM(GetVersion, "ncclGetVersion")
    SYM(GetUniqueId, "ncclGetUniqueId")
    SYM(CommInitRank, "ncclCommInitRank")
    SYM(CommDestroy, "ncclCommDestroy")
    SYM(GetErrorString, "ncclGetErrorString")
    SYM(GroupStart, "ncclGroupStart")
    SYM(GroupEnd, "ncclGroupEnd")
    SYM(Send, "ncclSend")
    SYM(Recv, "ncclRecv")
    SYM(AllGather, "ncclAllGather")
    SYM(AllReduce, "ncclAllReduce")
#undef SYM
    // ONE HIP runtime for RCCL and this library: streams and events made here are handed to RCCL.  A Python process that loads
    // libehyb.so first and torch afterwards ends up with two (torch's wheel brings its own libamdhip64.so under another file
    // name, its RCCL binds to that one): refuse, and say how to avoid it.
    void* theirs = dlsym(h, "hipStreamCreate");
    Dl_info a{}, b{};
    if (theirs && dladdr(theirs, &a) && dladdr((void*)&hipStreamCreate, &b) && a.dli_fbase != b.dli_fbase) {
        g_rccl_error = std::string("librccl (") + g_rccl.where + ") is bound to another HIP runtime (" + (a.dli_fname ? a.dli_fname : "?") + ") than libehyb.so (" +
                       (b.dli_fname ? b.dli_fname : "?") + "): in a process that uses torch, import torch BEFORE libehyb.so is loaded";
        return;
    }
    g_rccl.handle = h;
}

int need_rccl()
{
    std::call_once(g_rccl_once, load_rccl);
    if (!g_rccl.handle) EHYB_FAIL(EHYB_ERR_STATE, "RCCL: %s", g_rccl_error.c_str());
    return EHYB_OK;
}

#define NCCL_TRY(expr)                                                                                   \
    do {                                                                                                 \
        ncclResult_t r_ = (expr);                                                                        \
        if (r_ != ncclSuccess) EHYB_FAIL(EHYB_ERR_HIP, "RCCL: %s: %s", #expr, g_rccl.GetErrorString(r_)); \
    } while (0)

}  // namespace

struct ehyb_comm {
    ncclComm_t comm = nullptr;
    int rank = 0, world = 1, device = 0;
    hipStream_t stream = nullptr;  // the exchanges run here, beside the caller's compute stream
    hipEvent_t ev_gather[2] = {nullptr, nullptr};  // ehyb_gather_spmv: x ready / segments gathered
};

// One rank's halo exchange + multiply, everything the step needs resident: the send list, the packed send buffer, who
// gets / sends how much in which chunk, and the events of the step's waits (recorded again every step: a wait refers to
// the record that preceded it).
struct ehyb_halo {
    ehyb_comm* comm = nullptr;
    ehyb_plan* plan = nullptr;
    int n_chunks = 0;
    bool parts = false;                 // the plan multiplies column segment by column segment (panel form / CSR residual)
    int64_t n_send = 0;
    int32_t* d_send_idx = nullptr;
    double* d_send_buf = nullptr;
    std::vector<int64_t> send_cnt, recv_cnt;   // [chunk * world + peer]
    std::vector<int64_t> send_off;             // [chunk * world + peer] into d_send_buf
    std::vector<int64_t> recv_col;             // [chunk * world + peer] column of x the peer's entries of that chunk land at
    std::vector<hipEvent_t> ev;                // [0] packed; [1 + k] chunk k delivered; [n_chunks + 1] step done (comm stream may be reused)
    int64_t steps = 0;
    // exchange "cover" (ehyb_halo_set_partials): the plan's rows from row_split on are partial sums for OTHER ranks' rows, computed
    // here from this rank's own x entries; they leave as soon as they are closed, the ones computed elsewhere for this rank's rows
    // arrive in d_ybuf and are added into y at the end of the step
    bool cover = false;
    int row_split = 0;
    std::vector<int64_t> ysend_cnt, yrecv_cnt;  // [peer]
    int64_t n_yrecv = 0;
    double* d_ybuf = nullptr;
    int32_t* d_yidx = nullptr;
    hipEvent_t ev_foreign = nullptr, ev_partials = nullptr;
};

extern "C" {

int ehyb_rccl_version(int* version, char* where, int where_len)
{
    clear_error();
    if (int rc = need_rccl()) return rc;
    int v = 0;
    NCCL_TRY(g_rccl.GetVersion(&v));
    if (version) *version = v;
    if (where && where_len > 0) snprintf(where, (size_t)where_len, "%s", g_rccl.where.c_str());
    return EHYB_OK;
}

int ehyb_comm_unique_id(void* id_out)
{
    clear_error();
    if (!id_out) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_comm_unique_id: null");
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    NCCL_TRY(g_rccl.GetUniqueId(&id));
    static_assert(sizeof(id) == EHYB_COMM_ID_BYTES, "ncclUniqueId size");
    memcpy(id_out, &id, sizeof(id));
    return EHYB_OK;
}

int ehyb_comm_create(const void* id_in, int rank, int world, ehyb_comm** out)
{
    clear_error();
    if (!id_in || !out || world < 1 || rank < 0 || rank >= world) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_comm_create: bad arguments (rank %d of %d)", rank, world);
    *out = nullptr;
    if (int rc = need_rccl()) return rc;
    ncclUniqueId id;
    memcpy(&id, id_in, sizeof(id));
    ehyb_comm* c = new ehyb_comm;
    c->rank = rank;
    c->world = world;
    if (hipGetDevice(&c->device) != hipSuccess) {
        delete c;
        EHYB_FAIL(EHYB_ERR_NO_DEVICE, "ehyb_comm_create: no current device");
    }
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
    if (r != ncclSuccess) {
        delete c;
        EHYB_FAIL(EHYB_ERR_HIP, "RCCL: ncclCommInitRank(rank %d of %d): %s", rank, world, g_rccl.GetErrorString(r));
    }
    int lo = 0, hi = 0;
    (void)hipDeviceGetStreamPriorityRange(&lo, &hi);   // hi = the numerically lowest = highest priority
    if (hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, hi) != hipSuccess) {
        (void)g_rccl.CommDestroy(c->comm);
        delete c;
        EHYB_FAIL(EHYB_ERR_HIP, "ehyb_comm_create: hipStreamCreateWithPriority failed");
    }
    for (auto& e : c->ev_gather)
        if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess) {
            ehyb_comm_destroy(c);
            EHYB_FAIL(EHYB_ERR_HIP, "ehyb_comm_create: hipEventCreate failed");
        }
    *out = c;
    return EHYB_OK;
}

void ehyb_comm_destroy(ehyb_comm* c)
{
    if (!c) return;
    if (c->stream) {
        (void)hipStreamSynchronize(c->stream);
        (void)hipStreamDestroy(c->stream);
    }
    for (auto e : c->ev_gather)
        if (e) (void)hipEventDestroy(e);
    if (c->comm && g_rccl.handle) (void)g_rccl.CommDestroy(c->comm);
    delete c;
}

int ehyb_comm_info(const ehyb_comm* c, int* rank, int* world, void** stream)
{
    if (!c) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_comm_info: null");
    if (rank) *rank = c->rank;
    if (world) *world = c->world;
    if (stream) *stream = (void*)c->stream;
    return EHYB_OK;
}

// Thin collectives on device buffers of doubles, for the callers of the step (dot products of a distributed CG; the
// all-gather arm): asynchronous on `stream` (NULL = the communicator's own).
int ehyb_comm_allreduce_sum(ehyb_comm* c, double* buf_dev, int64_t count, void* stream)
{
    if (!c || !buf_dev || count < 0) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_comm_allreduce_sum: bad arguments");
    if (count == 0) return EHYB_OK;
    NCCL_TRY(g_rccl.AllReduce(buf_dev, buf_dev, (size_t)count, ncclFloat64, ncclSum, c->comm, stream ? (hipStream_t)stream : c->stream));
    return EHYB_OK;
}

int ehyb_comm_allgather(ehyb_comm* c, const double* send_dev, double* recv_dev, int64_t count_per_rank, void* stream)
{
    if (!c || !send_dev || !recv_dev || count_per_rank < 0) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_comm_allgather: bad arguments");
    if (count_per_rank == 0) return EHYB_OK;
    NCCL_TRY(g_rccl.AllGather(send_dev, recv_dev, (size_t)count_per_rank, ncclFloat64, c->comm, stream ? (hipStream_t)stream : c->stream));
    return EHYB_OK;
}

// north_star's "all-gatherv of x": x = [own segment, padded to seg_len | segment of rank 0 | ... | segment of rank world-1]
// (dist.py: GatherSpmv).  ONE call per multiply: all-gather on the communicator's stream while the ELL phase (own columns
// only) runs on compute_stream, then the residual phase.
int ehyb_gather_spmv(ehyb_comm* c, ehyb_plan* plan, double* x_dev, double* y_dev, int64_t seg_len, void* compute_stream)
{
    if (!c || !plan || !x_dev || !y_dev || seg_len < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gather_spmv: bad arguments");
    hipStream_t cs = (hipStream_t)compute_stream;
    hipEvent_t* ev = c->ev_gather;
    HIP_TRY(hipEventRecord(ev[0], cs));              // x of this step is ready (and last step's reads of the gathered part are done)
    HIP_TRY(hipStreamWaitEvent(c->stream, ev[0], 0));
    NCCL_TRY(g_rccl.AllGather(x_dev, x_dev + seg_len, (size_t)seg_len, ncclFloat64, c->comm, c->stream));
    HIP_TRY(hipEventRecord(ev[1], c->stream));
    ehyb_stats st;
    (void)ehyb_plan_stats(plan, &st);
    if (st.er_inline > 0) {   // one launch: nothing to overlap
        HIP_TRY(hipStreamWaitEvent(cs, ev[1], 0));
        return ehyb_spmv(plan, x_dev, y_dev, compute_stream);
    }
    int rc = ehyb_spmv_phase(plan, x_dev, y_dev, compute_stream, 1);
    if (rc != EHYB_OK) return rc;
    HIP_TRY(hipStreamWaitEvent(cs, ev[1], 0));
    return ehyb_spmv_phase(plan, x_dev, y_dev, compute_stream, 2);
}

int ehyb_halo_create(ehyb_comm* c, ehyb_plan* plan, int n_chunks, const int32_t* send_idx_host, int64_t n_send,
                     const int64_t* send_counts, const int64_t* recv_counts, ehyb_halo** out)
{
    clear_error();
    if (!c || !plan || !out || n_chunks < 1 || n_send < 0 || !send_counts || !recv_counts || (n_send > 0 && !send_idx_host))
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_create: bad arguments");
    *out = nullptr;
    const void* seg_p = nullptr;
    int64_t seg_n = 0;
    if (int rc = ehyb_plan_host_array(plan, EHYB_ARR_COL_SEG_FIRST, &seg_p, &seg_n)) return rc;
    if (seg_n != n_chunks + 2) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_create: the plan has %lld column segments, %d chunks need %d (ehyb_plan_create_segs)",
                                         (long long)(seg_n > 0 ? seg_n - 1 : 1), n_chunks, n_chunks + 1);
    const int32_t* seg = (const int32_t*)seg_p;
    ehyb_stats st;
    if (int rc = ehyb_plan_stats(plan, &st)) return rc;
    ehyb_halo* h = new ehyb_halo;
    h->comm = c;
    h->plan = plan;
    h->n_chunks = n_chunks;
    h->parts = !(st.er_inline > 0) && !plan->host.direct;
    h->n_send = n_send;
    const int W = c->world;
    h->send_cnt.assign(send_counts, send_counts + (size_t)n_chunks * W);
    h->recv_cnt.assign(recv_counts, recv_counts + (size_t)n_chunks * W);
    h->send_off.resize((size_t)n_chunks * W);
    h->recv_col.resize((size_t)n_chunks * W);
    int64_t so = 0;
    for (int k = 0; k < n_chunks; ++k) {
        int64_t col = seg[1 + k];
        for (int p = 0; p < W; ++p) {
            const size_t i = (size_t)k * W + p;
            if (h->send_cnt[i] < 0 || h->recv_cnt[i] < 0) {
                delete h;
                EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_create: negative count");
            }
            h->send_off[i] = so;
            so += h->send_cnt[i];
            h->recv_col[i] = col;
            col += h->recv_cnt[i];
        }
        if (col > seg[2 + k]) {
            delete h;
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_create: chunk %d receives %lld doubles, its column segment holds %d", k, (long long)(col - seg[1 + k]),
                      seg[2 + k] - seg[1 + k]);
        }
    }
    if (so != n_send) {
        delete h;
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_create: the send counts add up to %lld, the send list holds %lld", (long long)so, (long long)n_send);
    }
    for (int64_t i = 0; i < n_send; ++i)
        if (send_idx_host[i] < 0 || send_idx_host[i] >= seg[1]) {
            delete h;
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_create: send list entry %lld = %d is not one of the rank's own %d columns", (long long)i, send_idx_host[i], seg[1]);
        }
    bool ok = hipMalloc((void**)&h->d_send_idx, (size_t)std::max<int64_t>(n_send, 1) * 4) == hipSuccess &&
              hipMalloc((void**)&h->d_send_buf, (size_t)std::max<int64_t>(n_send, 1) * 8) == hipSuccess;
    if (ok && n_send > 0) ok = hipMemcpy(h->d_send_idx, send_idx_host, (size_t)n_send * 4, hipMemcpyHostToDevice) == hipSuccess;
    h->ev.assign((size_t)n_chunks + 2, nullptr);
    for (auto& e : h->ev)
        if (ok) ok = hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
    if (!ok) {
        set_error("ehyb_halo_create: %s", hipGetErrorString(hipGetLastError()));
        ehyb_halo_destroy(h);
        return EHYB_ERR_HIP;
    }
    *out = h;
    return EHYB_OK;
}

void ehyb_halo_destroy(ehyb_halo* h)
{
    if (!h) return;
    if (h->comm && h->comm->stream) (void)hipStreamSynchronize(h->comm->stream);
    for (auto e : h->ev)
        if (e) (void)hipEventDestroy(e);
    if (h->ev_foreign) (void)hipEventDestroy(h->ev_foreign);
    if (h->ev_partials) (void)hipEventDestroy(h->ev_partials);
    if (h->d_ybuf) (void)hipFree(h->d_ybuf);
    if (h->d_yidx) (void)hipFree(h->d_yidx);
    if (h->d_send_idx) (void)hipFree(h->d_send_idx);
    if (h->d_send_buf) (void)hipFree(h->d_send_buf);
    delete h;
}

// the grouped send/recv pairs of chunk k, on the communicator's stream
static int exchange_chunk(ehyb_halo* h, double* x, int k)
{
    const ehyb_comm* c = h->comm;
    const int W = c->world;
    NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    for (int p = 0; p < W && r == ncclSuccess; ++p) {
        const size_t i = (size_t)k * W + p;
        if (h->send_cnt[i] > 0) r = g_rccl.Send(h->d_send_buf + h->send_off[i], (size_t)h->send_cnt[i], ncclFloat64, p, c->comm, c->stream);
        if (r == ncclSuccess && h->recv_cnt[i] > 0) r = g_rccl.Recv(x + h->recv_col[i], (size_t)h->recv_cnt[i], ncclFloat64, p, c->comm, c->stream);
    }
    ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess || e != ncclSuccess)
        EHYB_FAIL(EHYB_ERR_HIP, "RCCL: exchange of chunk %d: %s", k, g_rccl.GetErrorString(r != ncclSuccess ? r : e));
    return EHYB_OK;
}

// y = A [x own | ghosts] with the ghosts fetched on the way.  x_dev: the rank's own x entries in plan order, the ghost
// columns behind them are WRITTEN by the exchange.  Asynchronous: everything is enqueued on compute_stream and the
// communicator's stream; the result is complete when compute_stream has drained.
// the partial sums this rank computed for the others (its plan's rows from row_split on, grouped by destination) against the ones the
// others computed for it, on the communicator's stream
static int exchange_partials(ehyb_halo* h, double* y)
{
    const ehyb_comm* c = h->comm;
    NCCL_TRY(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    int64_t so = h->row_split, ro = 0;
    for (int p = 0; p < c->world && r == ncclSuccess; ++p) {
        if (h->ysend_cnt[(size_t)p] > 0) r = g_rccl.Send(y + so, (size_t)h->ysend_cnt[(size_t)p], ncclFloat64, p, c->comm, c->stream);
        if (r == ncclSuccess && h->yrecv_cnt[(size_t)p] > 0) r = g_rccl.Recv(h->d_ybuf + ro, (size_t)h->yrecv_cnt[(size_t)p], ncclFloat64, p, c->comm, c->stream);
        so += h->ysend_cnt[(size_t)p];
        ro += h->yrecv_cnt[(size_t)p];
    }
    ncclResult_t e = g_rccl.GroupEnd();
    if (r != ncclSuccess || e != ncclSuccess) EHYB_FAIL(EHYB_ERR_HIP, "RCCL: exchange of the partial sums: %s", g_rccl.GetErrorString(r != ncclSuccess ? r : e));
    return EHYB_OK;
}

// Exchange "cover": per pair of ranks the hub columns of the block travel as x entries (as in the plain halo step), the rest of the
// block was handed to the column's owner, who multiplies it with its OWN x and ships one partial sum per row.  One step:
//   compute   pack | segment 0 (own columns: own rows AND foreign rows) | close the foreign rows | chunk 0's panels | ... | close own rows | add partials
//   comm           x chunk 0, x chunk 1, ...                          (after the foreign close:) partial sums out / in
// The x entries leave at once and are needed only after segment 0 has been multiplied; the partial sums leave as soon as they exist
// and are needed only at the very end: both directions hide behind the multiply.
static int halo_step_cover(ehyb_halo* h, double* x_dev, double* y_dev, void* compute_stream)
{
    ehyb_comm* c = h->comm;
    hipStream_t cs = (hipStream_t)compute_stream;
    const int K = h->n_chunks;
    int rc = ehyb_gather(x_dev, h->d_send_idx, h->d_send_buf, h->n_send, compute_stream);
    if (rc != EHYB_OK) return rc;
    HIP_TRY(hipEventRecord(h->ev[0], cs));
    HIP_TRY(hipStreamWaitEvent(c->stream, h->ev[0], 0));
    for (int k = 0; k < K; ++k) {
        if ((rc = exchange_chunk(h, x_dev, k)) != EHYB_OK) return rc;
        HIP_TRY(hipEventRecord(h->ev[1 + k], c->stream));
    }
    rc = ehyb_spmv_part(h->plan, x_dev, y_dev, compute_stream, 0, 1, EHYB_PART_FIRST | EHYB_PART_LAST_FOREIGN);
    if (rc != EHYB_OK) return rc;
    HIP_TRY(hipEventRecord(h->ev_foreign, cs));
    HIP_TRY(hipStreamWaitEvent(c->stream, h->ev_foreign, 0));
    if ((rc = exchange_partials(h, y_dev)) != EHYB_OK) return rc;
    HIP_TRY(hipEventRecord(h->ev_partials, c->stream));
    for (int k = 0; k < K && rc == EHYB_OK; ++k) {
        HIP_TRY(hipStreamWaitEvent(cs, h->ev[1 + k], 0));
        rc = ehyb_spmv_part(h->plan, x_dev, y_dev, compute_stream, 1 + k, 2 + k, k == K - 1 ? EHYB_PART_LAST : 0);
    }
    if (rc != EHYB_OK) return rc;
    HIP_TRY(hipStreamWaitEvent(cs, h->ev_partials, 0));
    return ehyb_scatter_add(y_dev, h->d_yidx, h->d_ybuf, h->n_yrecv, compute_stream);
}

static int halo_step_eager(ehyb_halo* h, double* x_dev, double* y_dev, void* compute_stream)
{
    if (h->cover) return halo_step_cover(h, x_dev, y_dev, compute_stream);
    ehyb_comm* c = h->comm;
    hipStream_t cs = (hipStream_t)compute_stream;
    const int K = h->n_chunks;
    // pack: the x entries the peers asked for, one gather for all chunks; the communicator's stream waits for it -- and, through
    // the same record, for the previous step's multiply, which still read the ghost columns this step overwrites
    int rc = ehyb_gather(x_dev, h->d_send_idx, h->d_send_buf, h->n_send, compute_stream);
    if (rc != EHYB_OK) return rc;
    HIP_TRY(hipEventRecord(h->ev[0], cs));
    HIP_TRY(hipStreamWaitEvent(c->stream, h->ev[0], 0));
    if (!h->parts) {
        for (int k = 0; k < K; ++k)
            if ((rc = exchange_chunk(h, x_dev, k)) != EHYB_OK) return rc;
        HIP_TRY(hipEventRecord(h->ev[K], c->stream));
        HIP_TRY(hipStreamWaitEvent(cs, h->ev[K], 0));
        rc = ehyb_spmv(h->plan, x_dev, y_dev, compute_stream);
    } else {
        // all chunks go on the wire in order, back to back; compute picks them up one by one
        for (int k = 0; k < K; ++k) {
            if ((rc = exchange_chunk(h, x_dev, k)) != EHYB_OK) return rc;
            HIP_TRY(hipEventRecord(h->ev[1 + k], c->stream));
        }
        rc = ehyb_spmv_part(h->plan, x_dev, y_dev, compute_stream, 0, 1, EHYB_PART_FIRST);   // own columns, while chunk 0 travels
        for (int k = 0; k < K && rc == EHYB_OK; ++k) {
            HIP_TRY(hipStreamWaitEvent(cs, h->ev[1 + k], 0));
            rc = ehyb_spmv_part(h->plan, x_dev, y_dev, compute_stream, 1 + k, 2 + k, k == K - 1 ? EHYB_PART_LAST : 0);
        }
    }
    return rc;
}

int ehyb_halo_spmv(ehyb_halo* h, double* x_dev, double* y_dev, void* compute_stream)
{
    if (!h || !x_dev || !y_dev) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_spmv: null argument");
    ++h->steps;
    // (Replaying the whole step from a hipGraph was built and taken out again in round 4: capturing RCCL's grouped send / recv
    // on a non-default stream crashes inside this RCCL 2.26.6 / HIP 7.0 -- a segmentation fault under hipStreamBeginCapture,
    // gpurun_out r04_g -- and the eager call costs the host 30-50 us against ~115 us of device time per step.)
    return halo_step_eager(h, x_dev, y_dev, compute_stream);
}

// Turns the halo object into the "cover" exchange: the plan was built with cfg.row_split = row_split and its rows from there on are
// this rank's FOREIGN rows, grouped by destination rank -- ysend_counts[p] of them are rank p's; yrecv_counts[p] partial sums arrive
// from rank p, and partial i (in arrival order: peer 0's first) belongs to row yrecv_idx_host[i] of this rank (plan order).
// y_dev of ehyb_halo_spmv then holds the own rows followed by the foreign rows.
int ehyb_halo_set_partials(ehyb_halo* h, int row_split, const int64_t* ysend_counts, const int64_t* yrecv_counts, const int32_t* yrecv_idx_host, int64_t n_yrecv)
{
    clear_error();
    if (!h || row_split <= 0 || !ysend_counts || !yrecv_counts || n_yrecv < 0 || (n_yrecv > 0 && !yrecv_idx_host))
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_set_partials: bad arguments");
    if (h->cover) EHYB_FAIL(EHYB_ERR_STATE, "ehyb_halo_set_partials: already set");
    ehyb_stats st;
    if (int rc = ehyb_plan_stats(h->plan, &st)) return rc;
    if (h->plan->cfg.row_split != row_split) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_set_partials: the plan was built with cfg.row_split = %d, not %d", h->plan->cfg.row_split, row_split);
    if (!(st.er_partials > 0) || st.nnz_ell != 0 || !h->parts)
        EHYB_FAIL(EHYB_ERR_STATE, "ehyb_halo_set_partials: the cover exchange needs a plan that multiplies in panel form alone (every window given up)");
    const int W = h->comm->world;
    int64_t ns = 0, nr = 0;
    for (int p = 0; p < W; ++p) {
        if (ysend_counts[p] < 0 || yrecv_counts[p] < 0) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_set_partials: negative count");
        ns += ysend_counts[p], nr += yrecv_counts[p];
    }
    if (nr != n_yrecv) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_set_partials: the receive counts add up to %lld, the index list holds %lld", (long long)nr, (long long)n_yrecv);
    if (row_split + ns > st.n_rows) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_set_partials: %lld foreign rows behind row %d, the plan has %lld rows", (long long)ns, row_split, (long long)st.n_rows);
    for (int64_t i = 0; i < n_yrecv; ++i)
        if (yrecv_idx_host[i] < 0 || yrecv_idx_host[i] >= row_split) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_halo_set_partials: partial %lld goes to row %d, not one of the rank's own %d", (long long)i, yrecv_idx_host[i], row_split);
    HIP_TRY(hipMalloc((void**)&h->d_ybuf, (size_t)std::max<int64_t>(n_yrecv, 1) * 8));
    HIP_TRY(hipMalloc((void**)&h->d_yidx, (size_t)std::max<int64_t>(n_yrecv, 1) * 4));
    if (n_yrecv > 0) HIP_TRY(hipMemcpy(h->d_yidx, yrecv_idx_host, (size_t)n_yrecv * 4, hipMemcpyHostToDevice));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_foreign, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&h->ev_partials, hipEventDisableTiming));
    h->ysend_cnt.assign(ysend_counts, ysend_counts + W);
    h->yrecv_cnt.assign(yrecv_counts, yrecv_counts + W);
    h->n_yrecv = n_yrecv;
    h->row_split = row_split;
    h->cover = true;
    return EHYB_OK;
}

}  // extern "C"
