// Harness-side matrix sources: Matrix Market reader/writer and deterministic synthetic
// generators.  The reader restates what matrixRead_unsym / matrixRead_sym build
// (reference solver_test.c:31-126, 127-265) without the x/y side effects; the banner and
// size-line grammar is the subset of mmio.c:96-217 the reference uses.
#include "ehyb_internal.h"

#include <fcntl.h>
#include <omp.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>
#include <zlib.h>

#include <algorithm>
#include <cctype>
#include <cmath>
#include <memory>
#include <new>
#include <numeric>

namespace ehyb {
namespace {

int alloc_matrix(int n, int64_t nnz, matrixCOO* m)
{
    memset(m, 0, sizeof *m);
    if (nnz > 0x7FFFFFFFll) EHYB_FAIL(EHYB_ERR_ARG, "matrix with %lld entries does not fit matrixCOO's int counts", (long long)nnz);
    m->dimension = n;
    m->totalNum = (int)nnz;
    size_t e = (size_t)std::max<int64_t>(nnz, 1);
    m->rowIdx = (int*)calloc((size_t)n + 1, sizeof(int));
    m->numInRow = (int*)calloc((size_t)n + 1, sizeof(int));
    m->numInRow2 = (int*)calloc((size_t)n + 1, sizeof(int));
    m->partBoundary = (int*)calloc((size_t)n + 1, sizeof(int));
    m->reorderList = (int*)calloc((size_t)n + 1, sizeof(int));
    m->diag = (double*)calloc((size_t)n + 1, sizeof(double));
    m->I = (int*)malloc(e * sizeof(int));
    m->J = (int*)malloc(e * sizeof(int));
    m->V = (double*)malloc(e * sizeof(double));
    if (!m->rowIdx || !m->numInRow || !m->numInRow2 || !m->partBoundary || !m->reorderList || !m->diag ||
        !m->I || !m->J || !m->V) {
        ehyb_matrix_free(m);
        EHYB_FAIL(EHYB_ERR_ALLOC, "out of memory for a %d-row, %lld-entry matrix", n, (long long)nnz);
    }
    prefault(m->I, e * sizeof(int)), prefault(m->J, e * sizeof(int)), prefault(m->V, e * sizeof(double));
    return EHYB_OK;
}

// rowIdx from numInRow, maxCol, sizing (solver_test.c:105-124 / 208-226 and 53-77 / 158-182)
int finish_matrix(matrixCOO* m, const ehyb_config* cfg)
{
    const int n = m->dimension;
    int maxc = 0;
    m->rowIdx[0] = 0;
    for (int i = 0; i < n; ++i) {
        m->rowIdx[i + 1] = m->rowIdx[i] + m->numInRow[i];
        maxc = std::max(maxc, m->numInRow[i]);
    }
    m->maxCol = maxc;
    int np = 1, cache = 0, kpp = 1;
    int rc = ehyb_sizing(n, cfg, &np, &cache, &kpp);
    if (rc != EHYB_OK) return rc;
    m->nParts = np;
    m->vectorCacheSize = (uint16_t)std::min(cache, 65535);
    m->kernelPerPart = (int16_t)kpp;
    m->partBoundary[0] = 0;
    return EHYB_OK;
}

// ---- number parsing of the Matrix Market body.  glibc's strtod costs 300-500 ns on a 17-digit value; tens of millions of
// lines are the common case (audikw_1: 39 M).  parse_double below is EXACT -- the value fscanf("%lg") would give
// (solver_test.c:97,197) -- or it declines: up to 19 significant digits m and a decimal exponent |e| <= 27 are both exact in
// the x87 extended format (64-bit mantissa; 5^27 < 2^63), so m * 10^e or m / 10^e is ONE correctly rounded extended operation,
// and rounding that to double is the correctly rounded double unless the extended result sits exactly on a midpoint between two
// doubles (low 11 bits = 0x400: one case in 2048) -- then, and for anything else out of the fast path's range (more digits, huge
// exponents, inf / nan, hex floats), strtod decides.
const long double kPow10L[28] = {1e0L,  1e1L,  1e2L,  1e3L,  1e4L,  1e5L,  1e6L,  1e7L,  1e8L,  1e9L,  1e10L, 1e11L, 1e12L, 1e13L,
                                 1e14L, 1e15L, 1e16L, 1e17L, 1e18L, 1e19L, 1e20L, 1e21L, 1e22L, 1e23L, 1e24L, 1e25L, 1e26L, 1e27L};
const double kPow10D[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                            1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

inline const char* skip_blanks(const char* p)
{
    while (*p == ' ' || *p == '\t') ++p;
    return p;
}

// decimal integer at p (after blanks); -> end of the digits, or nullptr if there is no digit / it overflows
inline const char* parse_index(const char* p, long* out)
{
    p = skip_blanks(p);
    bool neg = false;
    if (*p == '+' || *p == '-') neg = *p++ == '-';
    if ((unsigned)(*p - '0') > 9u) return nullptr;
    unsigned long v = 0;
    int nd = 0;
    for (; (unsigned)(*p - '0') <= 9u; ++p, ++nd) v = v * 10 + (unsigned)(*p - '0');
    if (nd > 18) return nullptr;
    *out = neg ? -(long)v : (long)v;
    return p;
}

// -> end of the number, value in *out; nullptr: no number here
inline const char* parse_double(const char* p, double* out)
{
    const char* const start = skip_blanks(p);
    p = start;
    bool neg = false;
    if (*p == '+' || *p == '-') neg = *p++ == '-';
    uint64_t m = 0;
    int nd = 0, dropped = 0, e10 = 0;
    bool any = false, fast = true;
    for (; (unsigned)(*p - '0') <= 9u; ++p) {
        any = true;
        if (nd < 19) {
            m = m * 10 + (unsigned)(*p - '0');
            nd += (m != 0);
        } else {
            fast = false;   // more significant digits than the mantissa holds
            ++dropped;
        }
    }
    if (*p == '.') {
        ++p;
        for (; (unsigned)(*p - '0') <= 9u; ++p) {
            any = true;
            if (nd < 19) {
                m = m * 10 + (unsigned)(*p - '0');
                nd += (m != 0);
                --e10;
            } else {
                fast = fast && *p == '0';   // trailing zeros beyond 19 digits change nothing
            }
        }
    }
    if (!any) fast = false;
    if (any && (*p == 'e' || *p == 'E' || *p == 'd' || *p == 'D')) {
        const char* q = p + 1;
        bool eneg = false;
        if (*q == '+' || *q == '-') eneg = *q++ == '-';
        if ((unsigned)(*q - '0') <= 9u) {
            int ex = 0;
            for (; (unsigned)(*q - '0') <= 9u; ++q) ex = ex < 100000 ? ex * 10 + (*q - '0') : ex;
            e10 += eneg ? -ex : ex;
            if (*p == 'd' || *p == 'D') fast = false;   // Fortran exponent letter: not what strtod reads -- let it decide
            p = q;
        }
    }
    (void)dropped;
    if (*p == 'x' || *p == 'X') fast = false;   // a hexadecimal float ("0x1p3"): strtod reads those
    if (fast) {
        if (m == 0) {
            *out = neg ? -0.0 : 0.0;
            return p;
        }
        if (m < (1ull << 53) && e10 >= -22 && e10 <= 22) {   // both factors exact doubles: one rounding
            const double d = e10 < 0 ? (double)m / kPow10D[-e10] : (double)m * kPow10D[e10];
            *out = neg ? -d : d;
            return p;
        }
        if (e10 >= -27 && e10 <= 27 && sizeof(long double) >= 10) {
            const long double r = e10 < 0 ? (long double)m / kPow10L[-e10] : (long double)m * kPow10L[e10];
            uint64_t mant;
            memcpy(&mant, &r, 8);   // x87 extended: the 64-bit mantissa comes first
            if ((mant & 0x7FFu) != 0x400u) {
                const double d = (double)r;
                *out = neg ? -d : d;
                return p;
            }
        }
    }
    char* e = nullptr;
    const double d = strtod(start, &e);
    if (e == start) return nullptr;
    *out = d;
    return e;
}

inline double hash_value(uint64_t a, uint64_t b)
{
    // ((31*i + 17*j) mod 200 - 100)/1000, an exact 0 replaced by 0.001
    int64_t h = (int64_t)((31 * a + 17 * b) % 200) - 100;
    return h == 0 ? 0.001 : (double)h / 1000.0;
}
inline double hash_value_mixed(uint64_t a, uint64_t b, uint64_t seed)
{
    int64_t h = (int64_t)(mix64(a * 0x9E3779B97F4A7C15ull ^ mix64(b + seed)) % 200) - 100;
    return h == 0 ? 0.001 : (double)h / 1000.0;
}

}  // namespace
}  // namespace ehyb

using namespace ehyb;

extern "C" {

// ------------------------------------------------------------------ Matrix Market
int ehyb_mm_read(const char* path, const ehyb_config* cfg, matrixCOO* out, int* is_symmetric)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!path || !out) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_mm_read: null argument");
    // The whole file in memory, plain or gzip (zlib reads both; "<path>.gz" is tried when <path>
    // does not exist, so `-m audikw_1` also finds ./read/audikw_1.mtx.gz).  The reference parses
    // with fscanf one entry at a time (solver_test.c:96-103,196-206); here the body is cut into
    // line-aligned pieces parsed in parallel -- tens of millions of lines are the common case.
    const bool verbose = cfg && cfg->verbose;
    double t_mark = wall_seconds();
    auto mark = [&](const char* what) {
        const double t = wall_seconds();
        if (verbose) fprintf(stderr, "[ehyb_mm_read] %-28s %.3f s\n", what, t - t_mark);
        t_mark = t;
    };
    // gzip: inflated into a buffer that grows by realloc (zlib's inflate is one serial stream: ~0.25 GB/s of text, the floor for
    // .gz input); plain: see below
    struct Text {
        const char* data = nullptr;
        size_t size = 0;      // bytes of text; data[size] is readable and '\0' unless `tail_copy` says otherwise
        void* map = nullptr;
        size_t map_len = 0;
        char* heap = nullptr;
        ~Text()
        {
            if (map) munmap(map, map_len);
            free(heap);
        }
    } text;
    {
        std::string real = path;
        int fd = open(real.c_str(), O_RDONLY);
        if (fd < 0) {
            real += ".gz";
            fd = open(real.c_str(), O_RDONLY);
        }
        if (fd < 0) EHYB_FAIL(EHYB_ERR_IO, "file read error: %s", path);
        unsigned char magic[2] = {0, 0};
        const ssize_t got_magic = pread(fd, magic, 2, 0);
        struct stat sb;
        const bool gz = got_magic == 2 && magic[0] == 0x1f && magic[1] == 0x8b;
        if (!gz && fstat(fd, &sb) == 0 && S_ISREG(sb.st_mode) && sb.st_size > 0) {
            // plain file: every thread reads its slice straight into place (pread), so the copy out of the page cache and the
            // first touch of the buffer's pages are both parallel.  (Parsing from a read-only mapping of the file was measured
            // too: 0.7 s instead of 0.1 s for 310 MB -- every page of the mapping faults in the parse, on few threads at a time.)
            const size_t size = (size_t)sb.st_size;
            text.heap = (char*)malloc(size + 1);
            if (!text.heap) {
                close(fd);
                EHYB_FAIL(EHYB_ERR_ALLOC, "out of memory reading %s", path);
            }
            prefault(text.heap, size);
            const size_t slice = size_t(8) << 20;
            const int64_t n_slices = (int64_t)((size + slice - 1) / slice);
            bool read_ok = true;
#pragma omp parallel for schedule(dynamic, 1)
            for (int64_t i = 0; i < n_slices; ++i) {
                size_t off = (size_t)i * slice;
                const size_t stop = std::min(size, off + slice);
                while (off < stop) {
                    const ssize_t got = pread(fd, text.heap + off, stop - off, (off_t)off);
                    if (got <= 0) {
#pragma omp atomic write
                        read_ok = false;
                        break;
                    }
                    off += (size_t)got;
                }
            }
            if (!read_ok) {
                close(fd);
                EHYB_FAIL(EHYB_ERR_IO, "file read error: %s", path);
            }
            text.heap[size] = '\0';
            text.data = text.heap, text.size = size;
        }
        close(fd);
        if (!text.data) {
            gzFile g = gzopen(real.c_str(), "rb");   // zlib reads plain files too
            if (!g) EHYB_FAIL(EHYB_ERR_IO, "file read error: %s", path);
            gzbuffer(g, 1u << 20);
            size_t used = 0, cap = size_t(1) << 26;
            text.heap = (char*)malloc(cap);
            for (;;) {
                if (!text.heap) {
                    gzclose(g);
                    EHYB_FAIL(EHYB_ERR_ALLOC, "out of memory reading %s", path);
                }
                if (cap - used < (size_t(1) << 22)) {
                    cap *= 2;
                    char* bigger = (char*)realloc(text.heap, cap);
                    if (!bigger) {
                        gzclose(g);
                        EHYB_FAIL(EHYB_ERR_ALLOC, "out of memory reading %s", path);
                    }
                    text.heap = bigger;
                }
                const int got = gzread(g, text.heap + used, (unsigned)std::min<size_t>(cap - used - 1, size_t(1) << 30));
                if (got < 0) {
                    gzclose(g);
                    EHYB_FAIL(EHYB_ERR_IO, "file read error: %s (damaged gzip stream?)", path);
                }
                if (got == 0) break;
                used += (size_t)got;
            }
            gzclose(g);
            text.heap[used] = '\0';  // number parsing never runs off the end
            text.data = text.heap, text.size = used;
        }
    }
    mark("file into memory");
    const char* const end = text.data + text.size;
    const char* cur = text.data;
    auto next_line = [&](const char* p) {
        const char* q = (const char*)memchr(p, '\n', (size_t)(end - p));
        return q ? q + 1 : end;
    };
    char line[1100];
    auto copy_line = [&](const char* p) {
        const char* q = next_line(p);
        const size_t len = std::min<size_t>((size_t)(q - p), sizeof line - 1);
        memcpy(line, p, len);
        line[len] = '\0';
        return q;
    };
    if (cur >= end) EHYB_FAIL(EHYB_ERR_FORMAT, "Could not process Matrix Market banner.");
    cur = copy_line(cur);
    char banner[64], object[64], format[64], field[64], symm[64];
    if (sscanf(line, "%63s %63s %63s %63s %63s", banner, object, format, field, symm) != 5)
        EHYB_FAIL(EHYB_ERR_FORMAT, "Could not process Matrix Market banner.");
    auto lower = [](char* s) {
        for (; *s; ++s) *s = (char)tolower((unsigned char)*s);
    };
    lower(object), lower(format), lower(field), lower(symm);
    if (strcmp(banner, "%%MatrixMarket") != 0 || strcmp(object, "matrix") != 0)
        EHYB_FAIL(EHYB_ERR_FORMAT, "Could not process Matrix Market banner.");
    if (strcmp(format, "coordinate") != 0)
        EHYB_FAIL(EHYB_ERR_FORMAT, "only coordinate (sparse) Matrix Market files are supported, got '%s'", format);
    const bool pattern = strcmp(field, "pattern") == 0;
    if (!pattern && strcmp(field, "real") != 0 && strcmp(field, "integer") != 0 && strcmp(field, "double") != 0)
        // solver_test.c:339-345 rejects complex
        EHYB_FAIL(EHYB_ERR_FORMAT, "Sorry, this application does not support Market Market type: [%s %s %s %s]", object, format, field, symm);
    const bool sym = strcmp(symm, "symmetric") == 0;
    const bool skew = strcmp(symm, "skew-symmetric") == 0;
    if (!sym && !skew && strcmp(symm, "general") != 0) EHYB_FAIL(EHYB_ERR_FORMAT, "unsupported Matrix Market symmetry '%s'", symm);
    // size line: first non-comment, non-blank line (mmio.c:189-217)
    long M = 0, N = 0, stored = 0;
    for (;;) {
        if (cur >= end) EHYB_FAIL(EHYB_ERR_FORMAT, "premature end of file before the size line");
        cur = copy_line(cur);
        if (line[0] == '%') continue;
        if (sscanf(line, "%ld %ld %ld", &M, &N, &stored) == 3) break;
    }
    if (M <= 0 || M != N || stored < 0 || M > 0x7FFFFFF0l)
        EHYB_FAIL(EHYB_ERR_FORMAT, "size line %ld x %ld with %ld entries: a square matrix is required", M, N, stored);
    const int n = (int)M;
    // (uninitialised: a value-initialised vector would touch 16 bytes per entry on one thread before the parse starts)
    std::unique_ptr<int[]> fi_own(new (std::nothrow) int[(size_t)stored + 1]), fj_own(new (std::nothrow) int[(size_t)stored + 1]);
    std::unique_ptr<double[]> fv_own(new (std::nothrow) double[(size_t)stored + 1]);
    if (!fi_own || !fj_own || !fv_own) EHYB_FAIL(EHYB_ERR_ALLOC, "out of memory for %ld entries of %s", stored, path);
    prefault(fi_own.get(), sizeof(int) * (size_t)stored), prefault(fj_own.get(), sizeof(int) * (size_t)stored), prefault(fv_own.get(), sizeof(double) * (size_t)stored);
    int* const fi = fi_own.get();
    int* const fj = fj_own.get();
    double* const fv = fv_own.get();
    {
        // pieces of the body that start at line starts; entry lines per piece; then parse
        const int pieces = (int)std::max<int64_t>(1, std::min<int64_t>(omp_get_max_threads(), (end - cur) / (1 << 16)));
        std::vector<const char*> start(pieces + 1);
        for (int t = 0; t <= pieces; ++t) {
            const char* p = cur + (int64_t)(end - cur) * t / pieces;
            start[t] = (t == 0) ? cur : (t == pieces ? end : next_line(p - 1));  // p-1: p itself may be a line start
        }
        auto is_entry = [](const char* p, const char* q) {  // a line with something other than blanks, not a comment
            for (; p < q; ++p)
                if (*p != ' ' && *p != '\t' && *p != '\r' && *p != '\n') return *p != '%';
            return false;
        };
        std::vector<int64_t> first(pieces + 1, 0);
        // lines per piece.  First try: every line of the body is an entry (no comment or blank line behind the size line --
        // the usual file), so counting newlines is enough; a piece that meets anything else while parsing says so and the
        // exact count (every line looked at twice) is taken instead.
        for (int attempt = 0; attempt < 2; ++attempt) {
            const bool exact = attempt == 1;
#pragma omp parallel for schedule(static, 1)
            for (int t = 0; t < pieces; ++t) {
                int64_t c = 0;
                if (exact) {
                    for (const char* p = start[t]; p < start[t + 1];) {
                        const char* q = next_line(p);
                        c += is_entry(p, q);
                        p = q;
                    }
                } else {
                    const char* p = start[t];
                    while (p < start[t + 1]) {
                        const char* q = (const char*)memchr(p, '\n', (size_t)(start[t + 1] - p));
                        if (!q) {
                            c += is_entry(p, start[t + 1]);   // a last line without a newline
                            break;
                        }
                        ++c;
                        p = q + 1;
                    }
                }
                first[t + 1] = c;
            }
            first[0] = 0;
            for (int t = 0; t < pieces; ++t) first[t + 1] += first[t];
            if (first[pieces] < stored)
                EHYB_FAIL(EHYB_ERR_FORMAT, "bad entry %lld of %ld in %s", (long long)first[pieces] + 1, stored, path);
            int64_t bad = -1;
            bool other_lines = false;
#pragma omp parallel for schedule(static, 1)
            for (int t = 0; t < pieces; ++t) {
                int64_t k = first[t];
                for (const char* p = start[t]; p < start[t + 1] && k < stored;) {
                    const char* q = next_line(p);
                    if (is_entry(p, q)) {
                        long a = 0, b = 0;
                        double v = 1.0;
                        const char* e1 = parse_index(p, &a);
                        const char* e2 = e1 ? parse_index(e1, &b) : nullptr;
                        const char* e3 = e2;
                        if (e2 && !pattern) e3 = parse_double(e2, &v);
                        if (!e1 || !e2 || !e3 || e3 > q || a < 1 || b < 1 || a > n || b > n) {
#pragma omp critical
                            if (bad < 0 || k < bad) bad = k;
                        } else {
                            fi[k] = (int)a - 1;  // 1-based -> 0-based (solver_test.c:98-99, 198-199)
                            fj[k] = (int)b - 1;
                            fv[k] = v;
                        }
                        ++k;
                    } else if (!exact) {
#pragma omp atomic write
                        other_lines = true;
                    }
                    p = q;
                }
            }
            if (other_lines && !exact) continue;   // comments or blank lines inside the body: count them properly and parse again
            if (bad >= 0) EHYB_FAIL(EHYB_ERR_FORMAT, "bad entry %lld of %ld in %s", (long long)bad + 1, stored, path);
            break;
        }
    }
    if (text.map) munmap(text.map, text.map_len), text.map = nullptr;
    free(text.heap), text.heap = nullptr;
    mark("parse");

    const bool mirror = sym || skew;
    int64_t total = stored;
    if (mirror) {
        int64_t off = 0;
#pragma omp parallel for schedule(static) reduction(+ : off)
        for (long k = 0; k < stored; ++k) off += fi[k] != fj[k];
        total += off;
    }
    int rc = alloc_matrix(n, total, out);
    if (rc != EHYB_OK) return rc;
    // Row-grouped placement in file order, the mirrored entry right after its original (solver_test.c:235-255); for general
    // files this groups rows stably, which leaves the per-row accumulation order of solver_test.c:102 unchanged.  In parallel by
    // ROW RANGE: every thread walks the whole entry list in file order and places the entries whose (target) row is its own --
    // the order inside a row is the serial one, the writes of a thread stay inside its rows.
    const int nt = std::max(1, omp_get_max_threads());
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const int r0 = (int)((int64_t)n * t / T), r1 = (int)((int64_t)n * (t + 1) / T);
        for (long k = 0; k < stored; ++k) {
            const int a = fi[k], b = fj[k];
            if (a >= r0 && a < r1) out->numInRow[a]++;
            if (mirror && a != b && b >= r0 && b < r1) out->numInRow[b]++;
        }
    }
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) {
        ehyb_matrix_free(out);
        return rc;
    }
    std::vector<int> fill((size_t)n, 0);
#pragma omp parallel num_threads(nt)
    {
        const int t = omp_get_thread_num(), T = omp_get_num_threads();
        const int r0 = (int)((int64_t)n * t / T), r1 = (int)((int64_t)n * (t + 1) / T);
        for (long k = 0; k < stored; ++k) {
            const int a = fi[k], b = fj[k];
            if (a >= r0 && a < r1) {
                const int64_t at = (int64_t)out->rowIdx[a] + fill[a]++;
                out->I[at] = a;
                out->J[at] = b;
                out->V[at] = fv[k];
                if (a == b) out->diag[a] = fv[k];
            }
            if (mirror && a != b && b >= r0 && b < r1) {
                const int64_t at2 = (int64_t)out->rowIdx[b] + fill[b]++;
                out->I[at2] = b;
                out->J[at2] = a;
                out->V[at2] = skew ? -fv[k] : fv[k];
            }
        }
    }
    mark("row-grouped placement");
    if (is_symmetric) *is_symmetric = mirror ? 1 : 0;
    return EHYB_OK;
}

int ehyb_mm_write(const char* path, const matrixCOO* m, int symmetric_lower_only)
{
    if (!path || !m) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_mm_write: null argument");
    FILE* f = fopen(path, "w");
    if (!f) EHYB_FAIL(EHYB_ERR_IO, "cannot write %s", path);
    int64_t cnt = 0;
#pragma omp parallel for schedule(static) reduction(+ : cnt)
    for (int k = 0; k < m->totalNum; ++k) cnt += !symmetric_lower_only || m->I[k] >= m->J[k];
    fprintf(f, "%%%%MatrixMarket matrix coordinate real %s\n", symmetric_lower_only ? "symmetric" : "general");
    fprintf(f, "%d %d %lld\n", m->dimension, m->dimension, (long long)cnt);
    // blocks of entries formatted by all threads (17 significant digits: the value reads back bit for bit), written in order
    const int T = std::max(1, omp_get_max_threads());
    constexpr int64_t kBlock = 1 << 18;
    std::vector<std::string> buf((size_t)T);
    bool ok = true;
    for (int64_t base = 0; base < m->totalNum && ok; base += kBlock * T) {
#pragma omp parallel for schedule(static, 1) num_threads(T)
        for (int t = 0; t < T; ++t) {
            std::string& b = buf[(size_t)t];
            b.clear();
            const int64_t k0 = std::min<int64_t>(m->totalNum, base + kBlock * t), k1 = std::min<int64_t>(m->totalNum, k0 + kBlock);
            char line[96];
            for (int64_t k = k0; k < k1; ++k)
                if (!symmetric_lower_only || m->I[k] >= m->J[k]) b.append(line, (size_t)snprintf(line, sizeof line, "%d %d %.17g\n", m->I[k] + 1, m->J[k] + 1, m->V[k]));
        }
        for (int t = 0; t < T && ok; ++t) ok = buf[(size_t)t].empty() || fwrite(buf[(size_t)t].data(), 1, buf[(size_t)t].size(), f) == buf[(size_t)t].size();
    }
    if (fclose(f) != 0 || !ok) EHYB_FAIL(EHYB_ERR_IO, "cannot write %s (disk full?)", path);
    return EHYB_OK;
}

int ehyb_matrix_from_csr(int n, const int64_t* rowptr, const int* cols, const double* vals,
                         const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || n <= 0 || !rowptr || rowptr[0] != 0 || rowptr[n] < 0 || (rowptr[n] > 0 && (!cols || !vals)))
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_from_csr: bad arguments");
    int rc = alloc_matrix(n, rowptr[n], out);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) {
        if (rowptr[i + 1] < rowptr[i]) {
            ehyb_matrix_free(out);
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_from_csr: rowptr not monotone at %d", i);
        }
        out->numInRow[i] = (int)(rowptr[i + 1] - rowptr[i]);
    }
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i)
        for (int64_t k = rowptr[i]; k < rowptr[i + 1]; ++k) {
            if ((unsigned)cols[k] >= (unsigned)n) {
                ehyb_matrix_free(out);
                EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_from_csr: column %d outside [0,%d)", cols[k], n);
            }
            out->I[k] = i;
            out->J[k] = cols[k];
            out->V[k] = vals[k];
            if (cols[k] == i) out->diag[i] = vals[k];
        }
    return EHYB_OK;
}

// ------------------------------------------------------------------ generators
int ehyb_gen_banded(int n, int band, int block, const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || n <= 0 || band <= 0 || block <= 0 || band > block || n % block != 0)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_banded: need n %% block == 0 and band <= block");
    int rc = alloc_matrix(n, (int64_t)n * band, out);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) out->numInRow[i] = band;
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < n; ++i) {
        int base = i / block * block, r = i % block;
        int64_t at = (int64_t)i * band;
        for (int d = 0; d < band; ++d) {
            int j = base + ((r + d - band / 2) % block + block) % block;
            out->I[at + d] = i;
            out->J[at + d] = j;
            out->V[at + d] = hash_value((uint64_t)i, (uint64_t)j);
            if (i == j) out->diag[i] = out->V[at + d];
        }
    }
    return EHYB_OK;
}

static int gen_fem3d_impl(int n, int dof, int nx, int ny, int extra_ppm, int scramble, uint64_t seed, int block, int n_blocks,
                          int near_min_ppm, const ehyb_config* cfg, matrixCOO* out);

int ehyb_gen_fem3d(int n, int dof, int nx, int ny, int extra_ppm, int scramble, uint64_t seed,
                   const ehyb_config* cfg, matrixCOO* out)
{
    return gen_fem3d_impl(n, dof, nx, ny, extra_ppm, scramble, seed, 0, 1, -1, cfg, out);
}

// Graded mesh: the same grid, but how many couplings a node keeps follows a smooth density field
// g in [0,1] over the grid (fine and coarse regions of an unstructured mesh): a first-shell coupling
// is kept with probability near_min + (1 - near_min) * g, a second-shell coupling with probability
// far_max * g^3, g taken at the sparser end of the coupling (the decision is the same from both
// ends, so the matrix stays symmetric).  With near_min 0.25 and far_max 0.9 rows run from about 7 to
// 115 node couplings -- 21 to 345 entries at 3 unknowns per node, audikw_1's range.
int ehyb_gen_fem3d_graded(int n, int dof, int nx, int ny, int near_min_ppm, int far_max_ppm, int scramble, uint64_t seed,
                          const ehyb_config* cfg, matrixCOO* out)
{
    if (near_min_ppm < 0 || near_min_ppm > 1000000 || far_max_ppm < 0 || far_max_ppm > 1000000) {
        set_error("ehyb_gen_fem3d_graded: probabilities are parts per million");
        return EHYB_ERR_ARG;
    }
    return gen_fem3d_impl(n, dof, nx, ny, far_max_ppm, scramble, seed, 0, 1, near_min_ppm, cfg, out);
}

// Rows of block `block` of n_blocks fem3d grids stacked along z: a matrix of dimension
// n * n_blocks whose rows outside [block*n, (block+1)*n) are empty.  Nodes are labelled
// block by block (each block scrambled on its own), so a block's rows reference its own
// columns plus those of the two layers next to it in the neighbouring blocks.  Every process of
// a weak-scaling run generates its own rows only; (0, 1) is ehyb_gen_fem3d itself.
int ehyb_gen_fem3d_block(int n, int dof, int nx, int ny, int extra_ppm, int scramble, uint64_t seed,
                         int block, int n_blocks, const ehyb_config* cfg, matrixCOO* out)
{
    return gen_fem3d_impl(n, dof, nx, ny, extra_ppm, scramble, seed, block, n_blocks, -1, cfg, out);
}

static int gen_fem3d_impl(int n, int dof, int nx, int ny, int extra_ppm, int scramble, uint64_t seed, int block, int n_blocks,
                          int near_min_ppm, const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    if (!out || n <= 0 || dof <= 0 || nx <= 0 || ny <= 0 || n % dof != 0 || extra_ppm < 0)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_fem3d: need n %% dof == 0 and positive grid sizes");
    if (n_blocks < 1 || block < 0 || block >= n_blocks || (int64_t)n * n_blocks > 0x7FFFFFFFll)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_fem3d_block: block %d of %d, %d rows each", block, n_blocks, n);
    OmpScope omp_scope(cfg);
    const int N = n / dof;
    const int64_t layer = (int64_t)nx * ny;
    const int nz = (int)((N + layer - 1) / layer);
    const int nzf = (int)(N / layer);  // full layers
    if (n_blocks > 1 && nzf < 3) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_fem3d_block: a block needs at least three full grid layers");
    // node labels of this block and of its two neighbours
    std::vector<int> perms[3];
    for (int o = -1; o <= 1; ++o) {
        const int b = block + o;
        if (b < 0 || b >= n_blocks) continue;
        std::vector<int>& pm = perms[o + 1];
        pm.resize(N);
        std::iota(pm.begin(), pm.end(), 0);
        if (scramble) {
            uint64_t s = (seed + 0x9E3779B97F4A7C15ull * (uint64_t)b) ^ 0xABCDEF12345ull;
            for (int i = N - 1; i > 0; --i) std::swap(pm[i], pm[(int)(splitmix64(s) % (uint64_t)(i + 1))]);
        }
    }
    const std::vector<int>& perm = perms[1];
    const uint64_t thr = (uint64_t)extra_ppm;
    const bool graded = near_min_ppm >= 0;
    if (graded && n_blocks != 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_fem3d_graded: single block only");
    // density field of the graded mesh, per grid position: a smooth product of three waves
    auto density = [&](int x, int y, int z) {
        const double a = std::sin(6.2831853 * 1.5 * (x + 0.5) / nx + 0.7), b = std::sin(6.2831853 * 1.0 * (y + 0.5) / ny + 1.3),
                     c = std::sin(6.2831853 * 1.25 * (z + 0.5) / nz + 0.4);
        return 0.5 + 0.5 * a * b * c;
    };
    const int64_t grid_block = n_blocks > 1 ? (int64_t)nz * layer : 0;  // grid ids per block (last layer may be partial)
    // neighbours of grid node g of this block, including g itself, as global node labels
    auto neighbours = [&](int g, int* nb) {
        int ix = (int)(g % nx), iy = (int)((g / nx) % ny), iz = (int)(g / layer);
        int c = 0;
        for (int dz = -2; dz <= 2; ++dz)
            for (int dy = -2; dy <= 2; ++dy)
                for (int dx = -2; dx <= 2; ++dx)
                    // Block b+1 sits on the last FULL layer of block b (height b*nzf), so that the
                    // interface is a whole grid layer even when a block ends in a partial one; a
                    // position may then hold a node of each block, both are neighbours.
                    for (int o = (block > 0 ? -1 : 0); o <= (block + 1 < n_blocks ? 1 : 0); ++o) {
                        const int x = ix + dx, y = iy + dy, z = iz + dz - o * nzf;
                        if (x < 0 || y < 0 || z < 0 || x >= nx || y >= ny || z >= nz) continue;
                        int64_t h = (int64_t)z * layer + (int64_t)y * nx + x;
                        if (h >= N) continue;
                        bool near = dx >= -1 && dx <= 1 && dy >= -1 && dy <= 1 && dz >= -1 && dz <= 1;
                        if (graded) {
                            const bool face = std::abs(dx) + std::abs(dy) + std::abs(dz) == 1;  // always kept: no row below 7 node couplings
                            if (h != g && !face) {
                                const double gd = std::min(density(ix, iy, iz), density(x, y, z));
                                const double p = near ? near_min_ppm * 1e-6 + (1.0 - near_min_ppm * 1e-6) * gd : extra_ppm * 1e-6 * gd * gd * gd;
                                uint64_t lo = std::min<int64_t>(g, h), hi = std::max<int64_t>(g, h);
                                if ((double)(mix64(lo * 0x100000001B3ull + hi + seed) % 1000000ull) >= p * 1e6) continue;
                            }
                        } else if (!near) {
                            if (thr == 0) continue;
                            const int64_t gg = block * grid_block + g, hh = (block + o) * grid_block + h;
                            uint64_t lo = std::min<int64_t>(gg, hh), hi = std::max<int64_t>(gg, hh);
                            if (mix64(lo * 0x100000001B3ull + hi + seed) % 1000000ull >= thr) continue;
                        }
                        nb[c++] = (block + o) * N + perms[o + 1][(size_t)h];
                    }
        return c;
    };
    // pass 1: row counts
    std::vector<int> cnt(N);
#pragma omp parallel for schedule(static, 1024)
    for (int g = 0; g < N; ++g) {
        int nb[256];  // 125 positions, at the block interface up to two nodes each
        cnt[perm[g]] = neighbours(g, nb);
    }
    int64_t pairs = 0;
    for (int a = 0; a < N; ++a) pairs += cnt[a];
    int rc = alloc_matrix(n * n_blocks, pairs * dof * dof, out);
    if (rc != EHYB_OK) return rc;
    const int row0 = block * n;
    for (int a = 0; a < N; ++a)
        for (int d = 0; d < dof; ++d) out->numInRow[row0 + a * dof + d] = cnt[a] * dof;
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
    // pass 2: fill, columns ascending within a row
#pragma omp parallel for schedule(static, 1024)
    for (int g = 0; g < N; ++g) {
        int nb[256];  // 125 positions, at the block interface up to two nodes each
        int c = neighbours(g, nb);
        std::sort(nb, nb + c);
        const int a = perm[g];
        for (int d = 0; d < dof; ++d) {
            const int i = row0 + a * dof + d;
            int64_t at = out->rowIdx[i];
            for (int k = 0; k < c; ++k)
                for (int e = 0; e < dof; ++e) {
                    const int j = nb[k] * dof + e;
                    out->I[at] = i;
                    out->J[at] = j;
                    double v = hash_value_mixed((uint64_t)std::min(i, j), (uint64_t)std::max(i, j), seed);
                    out->V[at] = v;
                    if (i == j) out->diag[i] = v;
                    ++at;
                }
        }
    }
    return EHYB_OK;
}

// One R-MAT edge sample: every sample has its own generator state, so the result does not depend on the
// thread count (nor on which process draws it).  (a,b,c,d) = (0.57,0.19,0.19,0.05).
static inline void rmat_sample(int scale, uint64_t seed, int64_t e, int* pi, int* pj)
{
    uint64_t s = mix64(seed * 0x9E3779B97F4A7C15ull + (uint64_t)e);
    int i = 0, j = 0;
    for (int l = 0; l < scale; ++l) {
        uint32_t r = (uint32_t)(splitmix64(s) >> 40) % 100;  // percent
        int bi, bj;
        if (r < 57) bi = 0, bj = 0;
        else if (r < 76) bi = 0, bj = 1;
        else if (r < 95) bi = 1, bj = 0;
        else bi = 1, bj = 1;
        i = (i << 1) | bi;
        j = (j << 1) | bj;
    }
    *pi = i;
    *pj = j;
}

// Rows of block `block` of the SAME matrix ehyb_gen_rmat(scale, edges, seed) makes, for a process that
// owns one of n_blocks row blocks (strong scaling, one process per GPU): the blocks are contiguous row
// ranges holding about equally many edge samples (cuts[0..n_blocks], the same on every process: they
// come from a histogram pass over all samples), and only the samples of the own block are kept, sorted
// and merged.  Dimension 2^scale, rows outside the block empty.
// rows [r0, r1) of the R-MAT: block < 0 -- the caller names the rows (ehyb_gen_rmat_rows); else block `block` of n_blocks blocks of equal cost,
// cuts filled (ehyb_gen_rmat_block)
// cost_model (blocks only): 0 = a row costs its samples + 2; 1 = the cost under the "cover" exchange of the multi-GPU step, where an entry is
// multiplied by the owner of its ROW if its column is the hub of the two (column degree >= row degree: the column's x entry travels) and by the
// owner of its COLUMN otherwise (a partial sum travels back): index k costs the samples it ends up multiplying + 2
static int gen_rmat_rows_impl(int scale, int64_t edges, uint64_t seed, int block, int n_blocks, int* cuts, int r0_in, int r1_in, const ehyb_config* cfg,
                              matrixCOO* out, int cost_model = 0)
{
    const int n = 1 << scale;
    // pass 1: samples per row (every process, nothing stored)
    std::vector<int64_t> hist((size_t)n + 1, 0);
    std::vector<int64_t> work;   // cost_model 1: prefix of the samples every index multiplies
    {
        const int nt = omp_get_max_threads();
        const bool cover = cost_model == 1 && block >= 0;
        std::vector<std::vector<int32_t>> part((size_t)nt), cpart((size_t)(cover ? nt : 0));
#pragma omp parallel
        {
            std::vector<int32_t>& h = part[omp_get_thread_num()];
            h.assign((size_t)n, 0);
            int32_t* hc = nullptr;
            if (cover) {
                cpart[omp_get_thread_num()].assign((size_t)n, 0);
                hc = cpart[omp_get_thread_num()].data();
            }
#pragma omp for schedule(static, 65536)
            for (int64_t e = 0; e < edges; ++e) {
                int i, j;
                rmat_sample(scale, seed, e, &i, &j);
                ++h[i];
                if (hc) ++hc[j];
            }
        }
        for (const auto& h : part)
            if (!h.empty())
                for (int i = 0; i < n; ++i) hist[i + 1] += h[i];
        if (cover) {
            std::vector<int32_t> rdeg((size_t)n), cdeg((size_t)n, 0);
            for (int i = 0; i < n; ++i) rdeg[(size_t)i] = (int32_t)hist[(size_t)i + 1];
            for (const auto& h : cpart)
                if (!h.empty())
                    for (int i = 0; i < n; ++i) cdeg[(size_t)i] += h[(size_t)i];
            // second sweep over the samples: who multiplies each (the per-thread row histograms are reused as counters)
#pragma omp parallel
            {
                std::vector<int32_t>& w = part[omp_get_thread_num()];
                std::fill(w.begin(), w.end(), 0);
#pragma omp for schedule(static, 65536)
                for (int64_t e = 0; e < edges; ++e) {
                    int i, j;
                    rmat_sample(scale, seed, e, &i, &j);
                    ++w[(size_t)(cdeg[(size_t)j] >= rdeg[(size_t)i] ? i : j)];
                }
            }
            work.assign((size_t)n + 1, 0);
            for (const auto& w : part)
                if (!w.empty())
                    for (int i = 0; i < n; ++i) work[(size_t)i + 1] += w[(size_t)i];
            for (int i = 0; i < n; ++i) work[(size_t)i + 1] += work[(size_t)i];
        }
    }
    for (int i = 0; i < n; ++i) hist[i + 1] += hist[i];
    // Blocks of equal COST, not of equal samples: besides its entries a row costs its x and y entries and -- in the panel
    // form the ranks of a power-law matrix multiply with -- about one partial sum of its own, i.e. per-row bytes.  Measured
    // on the plans of R-MAT 2^24 at 8 ranks (tools/dist_stats.py): format bytes = 18 B per entry + 34 B per row, so that
    // equal samples gave the rank with the 7.3 M low-degree rows 552 MB to move and the rank with the 70 k hub rows 295 MB.
    // A row counts for two samples more.
    // (cost_model 1, the "cover" exchange: model rows [0.13 .. 5.7 M] -> work max / mean 1.24 with the row cost, 1.05 with this one at 8 ranks;
    // 1.23 -> 1.08 at 4, 1.14 -> 1.01 at 2: R-MAT 2^22, the cover really chosen per block afterwards)
    auto cost_before = [&](int i) { return (work.empty() ? hist[(size_t)i] : work[(size_t)i]) + 2 * (int64_t)i; };
    if (block >= 0) cuts[0] = 0;
    for (int b = 1; block >= 0 && b < n_blocks; ++b) {
        const int64_t goal = cost_before(n) * b / n_blocks;
        int lo = 0, hi = n;  // first row index whose cost_before reaches the goal
        while (lo < hi) {
            const int mid = lo + (hi - lo) / 2;
            if (cost_before(mid) < goal)
                lo = mid + 1;
            else
                hi = mid;
        }
        int c = std::min(std::max(lo, cuts[b - 1] + 1), n - (n_blocks - b));
        cuts[b] = c;
    }
    if (block >= 0) cuts[n_blocks] = n;
    const int r0 = block >= 0 ? cuts[block] : r0_in, r1 = block >= 0 ? cuts[block + 1] : r1_in;
    // pass 2: the own block's samples, grouped by row
    const int64_t mine = hist[r1] - hist[r0];
    std::vector<int> cols((size_t)mine);
    {
        std::vector<int64_t> fill(hist.begin() + r0, hist.begin() + r1);
        for (auto& f : fill) f -= hist[r0];
        // samples of one row come from different threads: a serial scatter keeps it simple and ordered
        // (the sort below makes the order irrelevant anyway); the sampling itself is the parallel part
        const int64_t chunk = 1 << 22;
        std::vector<int> bi((size_t)chunk), bj((size_t)chunk);
        for (int64_t e0 = 0; e0 < edges; e0 += chunk) {
            const int64_t m2 = std::min(chunk, edges - e0);
#pragma omp parallel for schedule(static, 65536)
            for (int64_t q = 0; q < m2; ++q) rmat_sample(scale, seed, e0 + q, &bi[(size_t)q], &bj[(size_t)q]);
            for (int64_t q = 0; q < m2; ++q)
                if (bi[(size_t)q] >= r0 && bi[(size_t)q] < r1) cols[(size_t)fill[bi[(size_t)q] - r0]++] = bj[(size_t)q];
        }
    }
    std::vector<int> ucnt((size_t)(r1 - r0));
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = r0; i < r1; ++i) {
        auto b = cols.begin() + (hist[i] - hist[r0]), e = cols.begin() + (hist[i + 1] - hist[r0]);
        std::sort(b, e);
        ucnt[(size_t)(i - r0)] = (int)(std::unique(b, e) - b);
    }
    int64_t nnz = 0;
    for (int c : ucnt) nnz += c;
    int rc = alloc_matrix(n, nnz, out);
    if (rc != EHYB_OK) return rc;
    for (int i = r0; i < r1; ++i) out->numInRow[i] = ucnt[(size_t)(i - r0)];
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = r0; i < r1; ++i) {
        int64_t at = out->rowIdx[i];
        const int64_t src = hist[i] - hist[r0];
        for (int k = 0; k < ucnt[(size_t)(i - r0)]; ++k) {
            const int j = cols[(size_t)(src + k)];
            out->I[at + k] = i;
            out->J[at + k] = j;
            out->V[at + k] = hash_value_mixed((uint64_t)i, (uint64_t)j, seed);
            if (i == j) out->diag[i] = out->V[at + k];
        }
    }
    return EHYB_OK;
}

int ehyb_gen_rmat_block(int scale, int64_t edges, uint64_t seed, int block, int n_blocks, int* cuts, const ehyb_config* cfg,
                        matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || !cuts || scale < 1 || scale > 30 || edges < 1 || n_blocks < 1 || block < 0 || block >= n_blocks)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_rmat_block: bad arguments");
    if (n_blocks > (1 << scale)) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_rmat_block: more blocks than rows");
    return gen_rmat_rows_impl(scale, edges, seed, block, n_blocks, cuts, 0, 0, cfg, out);
}

int ehyb_gen_rmat_block_cost(int scale, int64_t edges, uint64_t seed, int block, int n_blocks, int cost_model, int* cuts, const ehyb_config* cfg,
                             matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || !cuts || scale < 1 || scale > 30 || edges < 1 || n_blocks < 1 || block < 0 || block >= n_blocks || cost_model < 0 || cost_model > 1)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_rmat_block_cost: bad arguments");
    if (n_blocks > (1 << scale)) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_rmat_block_cost: more blocks than rows");
    return gen_rmat_rows_impl(scale, edges, seed, block, n_blocks, cuts, 0, 0, cfg, out, cost_model);
}

int ehyb_gen_rmat_rows(int scale, int64_t edges, uint64_t seed, int row0, int row1, const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || scale < 1 || scale > 30 || edges < 1 || row0 < 0 || row1 > (1 << scale) || row0 >= row1)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_rmat_rows: bad arguments");
    return gen_rmat_rows_impl(scale, edges, seed, -1, 1, nullptr, row0, row1, cfg, out);
}

int ehyb_gen_rmat(int scale, int64_t edges, uint64_t seed, const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || scale < 1 || scale > 30 || edges < 1) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_rmat: bad arguments");
    const int n = 1 << scale;
    std::vector<int> ei((size_t)edges), ej((size_t)edges);
    // (a,b,c,d) = (0.57,0.19,0.19,0.05); every edge has its own generator state, so the
    // result does not depend on the thread count
#pragma omp parallel for schedule(static, 65536)
    for (int64_t e = 0; e < edges; ++e) rmat_sample(scale, seed, e, &ei[e], &ej[e]);
    // group by row, sort + unique the columns of each row
    std::vector<int64_t> rp((size_t)n + 1, 0);
    for (int64_t e = 0; e < edges; ++e) rp[ei[e] + 1]++;
    for (int i = 0; i < n; ++i) rp[i + 1] += rp[i];
    std::vector<int> cols((size_t)edges);
    {
        std::vector<int64_t> fill(rp.begin(), rp.end() - 1);
        for (int64_t e = 0; e < edges; ++e) cols[fill[ei[e]]++] = ej[e];
    }
    std::vector<int>().swap(ei);
    std::vector<int>().swap(ej);
    std::vector<int> ucnt(n);
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = 0; i < n; ++i) {
        auto b = cols.begin() + rp[i], e = cols.begin() + rp[i + 1];
        std::sort(b, e);
        ucnt[i] = (int)(std::unique(b, e) - b);
    }
    int64_t nnz = 0;
    for (int i = 0; i < n; ++i) nnz += ucnt[i];
    int rc = alloc_matrix(n, nnz, out);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) out->numInRow[i] = ucnt[i];
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
#pragma omp parallel for schedule(dynamic, 4096)
    for (int i = 0; i < n; ++i) {
        int64_t at = out->rowIdx[i];
        for (int k = 0; k < ucnt[i]; ++k) {
            int j = cols[rp[i] + k];
            out->I[at + k] = i;
            out->J[at + k] = j;
            out->V[at + k] = hash_value_mixed((uint64_t)i, (uint64_t)j, seed);
            if (i == j) out->diag[i] = out->V[at + k];
        }
    }
    return EHYB_OK;
}

int ehyb_gen_stencil2d(int nx, int ny, int points, int extra, uint64_t seed, const ehyb_config* cfg,
                       matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || nx <= 0 || ny <= 0 || (points != 5 && points != 9) || extra < 0)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_stencil2d: bad arguments");
    const int n = nx * ny;
    std::vector<std::vector<int>> adj(n);
    for (int y = 0; y < ny; ++y)
        for (int x = 0; x < nx; ++x) {
            int i = y * nx + x;
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    if (points == 5 && dx != 0 && dy != 0) continue;
                    int xx = x + dx, yy = y + dy;
                    if (xx < 0 || yy < 0 || xx >= nx || yy >= ny) continue;
                    adj[i].push_back(yy * nx + xx);
                }
        }
    uint64_t s = seed + 77;
    for (int k = 0; k < extra; ++k) {
        int a = (int)(splitmix64(s) % (uint64_t)n), b = (int)(splitmix64(s) % (uint64_t)n);
        if (a == b) continue;
        adj[a].push_back(b);
        adj[b].push_back(a);
    }
    int64_t nnz = 0;
    for (int i = 0; i < n; ++i) {
        std::sort(adj[i].begin(), adj[i].end());
        adj[i].erase(std::unique(adj[i].begin(), adj[i].end()), adj[i].end());
        nnz += (int64_t)adj[i].size();
    }
    int rc = alloc_matrix(n, nnz, out);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) out->numInRow[i] = (int)adj[i].size();
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) {
        int64_t at = out->rowIdx[i];
        for (int j : adj[i]) {
            out->I[at] = i;
            out->J[at] = j;
            out->V[at] = hash_value_mixed((uint64_t)std::min(i, j), (uint64_t)std::max(i, j), seed);
            if (i == j) out->diag[i] = out->V[at];
            ++at;
        }
    }
    return EHYB_OK;
}

// Unstructured-mesh stand-in (nothing here is a lattice): `nodes` random points in the unit cube, denser towards one
// corner (coordinates u^grade), every node coupled to its `knn` nearest neighbours -- found through a cell grid --, the
// coupling made symmetric (node degrees knn .. ~2 knn), `dof` unknowns per node with dense dof x dof blocks and
// symmetric hashed values.  Labels are the points' random order: no locality in the numbering, row lengths vary.
int ehyb_gen_mesh3d(int n, int dof, int knn, int grade_permille, uint64_t seed, const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || n <= 0 || dof < 1 || dof > 8 || knn < 1 || knn > 64 || grade_permille < 0)
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_mesh3d: bad arguments");
    const int nodes = (n + dof - 1) / dof;
    const double grade = grade_permille > 0 ? grade_permille / 1000.0 : 1.0;
    std::vector<float> px(nodes), py(nodes), pz(nodes);
    {
        uint64_t s = seed * 0x9E3779B97F4A7C15ull + 12345;
        for (int i = 0; i < nodes; ++i) {
            const double u = (double)(splitmix64(s) >> 11) / 9007199254740992.0, v = (double)(splitmix64(s) >> 11) / 9007199254740992.0,
                         w = (double)(splitmix64(s) >> 11) / 9007199254740992.0;
            px[i] = (float)std::pow(u, grade);
            py[i] = (float)std::pow(v, grade);
            pz[i] = (float)w;
        }
    }
    // cell grid with about 4 points per cell on average
    const int g = std::max(1, (int)std::cbrt(nodes / 4.0));
    auto cell_of = [&](int i) {
        const int cx = std::min(g - 1, (int)(px[i] * g)), cy = std::min(g - 1, (int)(py[i] * g)), cz = std::min(g - 1, (int)(pz[i] * g));
        return (cz * g + cy) * g + cx;
    };
    std::vector<int> cell_ptr((size_t)g * g * g + 1, 0), cell_nodes(nodes);
    for (int i = 0; i < nodes; ++i) ++cell_ptr[(size_t)cell_of(i) + 1];
    for (size_t c = 0; c < (size_t)g * g * g; ++c) cell_ptr[c + 1] += cell_ptr[c];
    {
        std::vector<int> fill(cell_ptr.begin(), cell_ptr.end() - 1);
        for (int i = 0; i < nodes; ++i) cell_nodes[fill[cell_of(i)]++] = i;
    }
    // knn nearest of every node: rings of cells until the knn-th best distance is inside the ring searched
    std::vector<int> nbr((size_t)nodes * knn, -1);
#pragma omp parallel
    {
        std::vector<std::pair<float, int>> cand;
#pragma omp for schedule(dynamic, 256)
        for (int i = 0; i < nodes; ++i) {
            const int cx = std::min(g - 1, (int)(px[i] * g)), cy = std::min(g - 1, (int)(py[i] * g)), cz = std::min(g - 1, (int)(pz[i] * g));
            cand.clear();
            for (int ring = 0; ring <= g; ++ring) {
                for (int z = cz - ring; z <= cz + ring; ++z)
                    for (int y = cy - ring; y <= cy + ring; ++y)
                        for (int x = cx - ring; x <= cx + ring; ++x) {
                            if (std::max(std::abs(z - cz), std::max(std::abs(y - cy), std::abs(x - cx))) != ring) continue;  // this ring's shell only
                            if (x < 0 || y < 0 || z < 0 || x >= g || y >= g || z >= g) continue;
                            const int c = (z * g + y) * g + x;
                            for (int q = cell_ptr[c]; q < cell_ptr[c + 1]; ++q) {
                                const int j = cell_nodes[q];
                                if (j == i) continue;
                                const float dx = px[j] - px[i], dy = py[j] - py[i], dz = pz[j] - pz[i];
                                cand.push_back({dx * dx + dy * dy + dz * dz, j});
                            }
                        }
                if ((int)cand.size() >= knn) {
                    std::nth_element(cand.begin(), cand.begin() + (knn - 1), cand.end());
                    const float reach = (float)ring / g;  // everything nearer than this has been seen
                    if (cand[knn - 1].first <= reach * reach || ring == g) break;
                }
            }
            const int k = std::min(knn, (int)cand.size());
            std::partial_sort(cand.begin(), cand.begin() + k, cand.end());
            for (int q = 0; q < k; ++q) nbr[(size_t)i * knn + q] = cand[q].second;
        }
    }
    // symmetric node adjacency, self included
    std::vector<int> deg(nodes, 1);
    for (int i = 0; i < nodes; ++i)
        for (int q = 0; q < knn; ++q) {
            const int j = nbr[(size_t)i * knn + q];
            if (j >= 0) ++deg[i], ++deg[j];
        }
    std::vector<int64_t> aptr((size_t)nodes + 1, 0);
    for (int i = 0; i < nodes; ++i) aptr[i + 1] = aptr[i] + deg[i];
    std::vector<int> adj((size_t)aptr[nodes]);
    {
        std::vector<int64_t> fill(aptr.begin(), aptr.end() - 1);
        for (int i = 0; i < nodes; ++i) {
            adj[fill[i]++] = i;
            for (int q = 0; q < knn; ++q) {
                const int j = nbr[(size_t)i * knn + q];
                if (j >= 0) adj[fill[i]++] = j, adj[fill[j]++] = i;
            }
        }
    }
    std::vector<int> cnt(nodes, 0);
#pragma omp parallel for schedule(static, 1024)
    for (int i = 0; i < nodes; ++i) {
        std::sort(adj.begin() + aptr[i], adj.begin() + aptr[i + 1]);
        cnt[i] = (int)(std::unique(adj.begin() + aptr[i], adj.begin() + aptr[i + 1]) - (adj.begin() + aptr[i]));
    }
    // rows: unknown d of node a = row a*dof + d (rows beyond n dropped, columns beyond n too)
    int64_t nnz = 0;
    std::vector<int> rowlen(n, 0);
    for (int a = 0; a < nodes; ++a)
        for (int d = 0; d < dof && a * dof + d < n; ++d) {
            int len = 0;
            for (int q = 0; q < cnt[a]; ++q) {
                const int b = adj[aptr[a] + q];
                len += std::min(dof, n - b * dof) > 0 ? std::min(dof, n - b * dof) : 0;
            }
            rowlen[a * dof + d] = len;
            nnz += len;
        }
    int rc = alloc_matrix(n, nnz, out);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) out->numInRow[i] = rowlen[i];
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
#pragma omp parallel for schedule(static, 256)
    for (int a = 0; a < nodes; ++a)
        for (int d = 0; d < dof && a * dof + d < n; ++d) {
            const int i = a * dof + d;
            int64_t at = out->rowIdx[i];
            for (int q = 0; q < cnt[a]; ++q) {
                const int b = adj[aptr[a] + q];
                for (int e = 0; e < dof && b * dof + e < n; ++e) {
                    const int j = b * dof + e;
                    out->I[at] = i;
                    out->J[at] = j;
                    out->V[at] = hash_value_mixed((uint64_t)std::min(i, j), (uint64_t)std::max(i, j), seed);
                    if (i == j) out->diag[i] = out->V[at];
                    ++at;
                }
            }
        }
    return EHYB_OK;
}

// KKT-like saddle point system [H A^T; A 0] on an nx^3 grid: H = 7-point stencil, A couples a
// constraint to the 19 primal unknowns within one face/edge step; the zero block keeps an
// explicit zero diagonal, as nlpkkt200 stores it.
int ehyb_gen_kkt3d(int nx, const ehyb_config* cfg, matrixCOO* out)
{
    clear_error();
    OmpScope omp_scope(cfg);
    if (!out || nx < 2 || (int64_t)nx * nx * nx * 2 > 0x7FFFFFF0ll) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_gen_kkt3d: bad size");
    const int n1 = nx * nx * nx, n = 2 * n1;
    auto id = [&](int x, int y, int z) { return (z * nx + y) * nx + x; };
    auto inside = [&](int x, int y, int z) { return x >= 0 && y >= 0 && z >= 0 && x < nx && y < nx && z < nx; };
    auto row_cols = [&](int i, int* c) {
        int k = 0;
        const bool primal = i < n1;
        const int g = primal ? i : i - n1;
        const int x = g % nx, y = (g / nx) % nx, z = g / (nx * nx);
        for (int dz = -1; dz <= 1; ++dz)
            for (int dy = -1; dy <= 1; ++dy)
                for (int dx = -1; dx <= 1; ++dx) {
                    int man = abs(dx) + abs(dy) + abs(dz);
                    if (man == 3 || !inside(x + dx, y + dy, z + dz)) continue;
                    int h = id(x + dx, y + dy, z + dz);
                    if (primal) {
                        if (man <= 1) c[k++] = h;  // H
                        c[k++] = n1 + h;           // A^T
                    } else {
                        c[k++] = h;  // A
                    }
                }
        if (!primal) c[k++] = i;  // explicit zero diagonal
        std::sort(c, c + k);
        return k;
    };
    std::vector<int> cnt(n);
#pragma omp parallel for schedule(static, 4096)
    for (int i = 0; i < n; ++i) {
        int c[64];
        cnt[i] = row_cols(i, c);
    }
    int64_t nnz = 0;
    for (int i = 0; i < n; ++i) nnz += cnt[i];
    int rc = alloc_matrix(n, nnz, out);
    if (rc != EHYB_OK) return rc;
    for (int i = 0; i < n; ++i) out->numInRow[i] = cnt[i];
    rc = finish_matrix(out, cfg);
    if (rc != EHYB_OK) return rc;
#pragma omp parallel for schedule(static, 4096)
    for (int i = 0; i < n; ++i) {
        int c[64];
        int k = row_cols(i, c);
        int64_t at = out->rowIdx[i];
        for (int q = 0; q < k; ++q) {
            int j = c[q];
            out->I[at + q] = i;
            out->J[at + q] = j;
            double v = (i >= n1 && j == i) ? 0.0 : hash_value_mixed((uint64_t)std::min(i, j), (uint64_t)std::max(i, j), 7);
            out->V[at + q] = v;
            if (i == j) out->diag[i] = v;
        }
    }
    return EHYB_OK;
}

// A rank's square diagonal block (already reordered: partBoundary covers [0, n_loc)) grows n_ghost
// columns for the x entries it receives from other ranks: dimension n_loc + n_ghost, the new rows
// empty, the coupling entries (gi = row in the block's current numbering, gj = ghost slot) merged
// into their rows behind the local columns.  Partition data stays as it is; a plan over rows
// [0, n_loc) with cfg.n_top > 1 sends every ghost column to the residual (phase 2), whose x is
// [local x | receive buffer] in one allocation.
int ehyb_matrix_append_ghosts(matrixCOO* m, int n_ghost, int64_t nnz_g, const int* gi, const int* gj, const double* gv)
{
    clear_error();
    OmpScope omp_scope(0);
    if (!m || !m->rowIdx || n_ghost < 0 || nnz_g < 0 || (nnz_g > 0 && (!gi || !gj || !gv)))
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_ghosts: bad arguments");
    const int n0 = m->dimension;
    const int64_t n1 = (int64_t)n0 + n_ghost, nnz1 = (int64_t)m->totalNum + nnz_g;
    if (n1 > 0x7FFFFFFFll || nnz1 > 0x7FFFFFFFll) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_ghosts: result does not fit int counts");
    std::vector<int> add(n0, 0);
    for (int64_t k = 0; k < nnz_g; ++k) {
        if ((unsigned)gi[k] >= (unsigned)n0 || (unsigned)gj[k] >= (unsigned)n_ghost)
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_ghosts: entry %lld (%d,%d) out of range", (long long)k, gi[k], gj[k]);
        ++add[gi[k]];
    }
    const size_t e = (size_t)std::max<int64_t>(nnz1, 1);
    int* I = (int*)malloc(e * sizeof(int));
    int* J = (int*)malloc(e * sizeof(int));
    double* V = (double*)malloc(e * sizeof(double));
    int* rowIdx = (int*)calloc((size_t)n1 + 1, sizeof(int));
    auto grow = [&](auto*& p) {
        using T = std::remove_reference_t<decltype(*p)>;
        T* q = (T*)realloc(p, ((size_t)n1 + 1) * sizeof(T));
        if (!q) return false;
        memset(q + n0 + 1, 0, ((size_t)n1 - n0) * sizeof(T));  // the arrays hold n0 + 1 entries so far
        p = q;
        return true;
    };
    if (!I || !J || !V || !rowIdx || !grow(m->numInRow) || !grow(m->numInRow2) || !grow(m->partBoundary) ||
        !grow(m->reorderList) || !grow(m->diag)) {
        free(I), free(J), free(V), free(rowIdx);
        EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_matrix_append_ghosts: out of memory");
    }
    std::vector<int> fill(n0);
    for (int r = 0; r < n0; ++r) {
        const int len = m->rowIdx[r + 1] - m->rowIdx[r];
        rowIdx[r + 1] = rowIdx[r] + len + add[r];
        std::copy(m->I + m->rowIdx[r], m->I + m->rowIdx[r + 1], I + rowIdx[r]);
        std::copy(m->J + m->rowIdx[r], m->J + m->rowIdx[r + 1], J + rowIdx[r]);
        std::copy(m->V + m->rowIdx[r], m->V + m->rowIdx[r + 1], V + rowIdx[r]);
        fill[r] = rowIdx[r] + len;
        m->numInRow[r] = len + add[r];
        m->maxCol = std::max(m->maxCol, len + add[r]);
    }
    for (int64_t r = n0; r < n1; ++r) {
        rowIdx[r + 1] = rowIdx[n0];
        m->reorderList[r] = (int)r;
    }
    for (int64_t k = 0; k < nnz_g; ++k) {
        const int at = fill[gi[k]]++;
        I[at] = gi[k];
        J[at] = n0 + gj[k];
        V[at] = gv[k];
    }
    // ghost columns ascending within a row
    std::vector<std::pair<int, double>> tmp;
    for (int r = 0; r < n0; ++r) {
        if (add[r] < 2) continue;
        const int b = rowIdx[r + 1] - add[r];
        tmp.resize(add[r]);
        for (int k = 0; k < add[r]; ++k) tmp[k] = {J[b + k], V[b + k]};
        std::sort(tmp.begin(), tmp.end(), [](const std::pair<int, double>& a, const std::pair<int, double>& c) { return a.first < c.first; });
        for (int k = 0; k < add[r]; ++k) J[b + k] = tmp[k].first, V[b + k] = tmp[k].second;
    }
    free(m->I), free(m->J), free(m->V), free(m->rowIdx);
    m->I = I, m->J = J, m->V = V, m->rowIdx = rowIdx;
    m->dimension = (int)n1;
    m->totalNum = (int)nnz1;
    return EHYB_OK;
}

int ehyb_matrix_append_rows(matrixCOO* m, int row0, int n_rows, int64_t nnz, const int* ri, const int* cj, const double* v, int rows_per_part)
{
    clear_error();
    if (!m || !m->rowIdx || !m->partBoundary || row0 < 0 || n_rows < 0 || nnz < 0 || (nnz > 0 && (!ri || !cj || !v)))
        EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_rows: bad arguments");
    const int n = m->dimension;
    if ((int64_t)row0 + n_rows > n) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_rows: rows [%d, %d) outside the dimension %d", row0, row0 + n_rows, n);
    if (m->partBoundary[m->nParts] > row0) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_rows: row %d lies inside the partitions (they end at %d)", row0, m->partBoundary[m->nParts]);
    if (m->rowIdx[n] != m->rowIdx[row0]) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_rows: rows from %d on are not empty", row0);
    const int64_t nnz0 = m->totalNum, nnz1 = nnz0 + nnz;
    if (nnz1 > 0x7FFFFFFFll) EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_rows: result does not fit int counts");
    for (int64_t k = 0; k < nnz; ++k)
        if ((unsigned)ri[k] >= (unsigned)n_rows || (unsigned)cj[k] >= (unsigned)n || (k > 0 && ri[k] < ri[k - 1]))
            EHYB_FAIL(EHYB_ERR_ARG, "ehyb_matrix_append_rows: entry %lld (%d,%d) out of range or rows not ascending", (long long)k, ri[k], cj[k]);
    if (n_rows == 0) return EHYB_OK;
    const size_t e = (size_t)std::max<int64_t>(nnz1, 1);
    int* I = (int*)realloc(m->I, e * sizeof(int));
    if (I) m->I = I;
    int* J = (int*)realloc(m->J, e * sizeof(int));
    if (J) m->J = J;
    double* V = (double*)realloc(m->V, e * sizeof(double));
    if (V) m->V = V;
    if (!I || !J || !V) EHYB_FAIL(EHYB_ERR_ALLOC, "ehyb_matrix_append_rows: out of memory");
    std::vector<int> cnt((size_t)n_rows, 0);
    for (int64_t k = 0; k < nnz; ++k) {
        m->I[nnz0 + k] = row0 + ri[k];
        m->J[nnz0 + k] = cj[k];
        m->V[nnz0 + k] = v[k];
        ++cnt[(size_t)ri[k]];
    }
    for (int r = 0; r < n_rows; ++r) {
        m->rowIdx[row0 + r + 1] = m->rowIdx[row0 + r] + cnt[(size_t)r];
        m->numInRow[row0 + r] = cnt[(size_t)r];
        m->numInRow2[row0 + r] = 0;   // no entry of a foreign row lies in its own partition's window
        m->maxCol = std::max(m->maxCol, cnt[(size_t)r]);
    }
    for (int r = row0 + n_rows; r < n; ++r) m->rowIdx[r + 1] = m->rowIdx[row0 + n_rows];
    m->totalNum = (int)nnz1;
    // new partitions behind the existing ones; a gap between the last boundary and row0 (padding rows) becomes a partition of its own
    const int per = rows_per_part > 0 ? rows_per_part : std::max<int>(kSlabRows, m->vectorCacheSize);
    int np = m->nParts;
    if (m->partBoundary[np] < row0) m->partBoundary[++np] = row0;
    for (int r = row0; r < row0 + n_rows; r += per) m->partBoundary[++np] = std::min(row0 + n_rows, r + per);
    m->nParts = np;
    return EHYB_OK;
}

}  // extern "C"
