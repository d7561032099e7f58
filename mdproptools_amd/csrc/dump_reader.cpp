// dump_reader.cpp — native LAMMPS text-dump reader (host side of the path).
//
// Replaces, for the hot path's inputs, what the reference gets from the un-vendored
// pymatgen.io.lammps.outputs.parse_lammps_dumps + pandas.read_csv (call sites
// structural/rdf_cn.py:176, dynamical/diffusion.py:172, dynamical/conductivity.py:87): it turns the
// text of a frame straight into the SoA float64 planes the kernels take, optionally ordered by atom id
// (rdf_cn.py:192 `sort_values("id")`), without building a DataFrame.
//
// Number parsing: decimal digits are accumulated into a 64-bit integer with a power-of-ten exponent;
// when the digit string has <= 19 digits, the integer is < 2^53 and |exponent| <= 22 the result is
// double(m) * 10^e or double(m) / 10^e — ONE correctly rounded IEEE operation on two exact doubles
// (Clinger's fast path), otherwise glibc strtod (also correctly rounded). Every field is therefore the
// correctly rounded double of its text. pandas' default reader agrees with that for fields of up to
// 15 significant digits (everything LAMMPS writes by default); for longer mantissas pandas itself can be
// 1 ulp off the correctly rounded value, this reader is not (tests/test_dump_reader_cpu.py).
#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/mdhip.h"

namespace {

struct FrameIndex {
    int64_t timestep = 0;
    int64_t natoms = 0;
    double bounds[6] = {0, 0, 0, 0, 0, 0};  // xlo xhi ylo yhi zlo zhi (as written)
    double tilt[3] = {0, 0, 0};
    int triclinic = 0;
    std::string columns;       // space separated
    int n_cols = 0;
    size_t body_begin = 0;     // offset of the first atom line
    size_t body_end = 0;       // offset one past the last atom line
};

}  // namespace

struct mdhip_dump {
    int fd = -1;
    const char *data = nullptr;
    size_t size = 0;
    bool mapped = true;  // data is an mmap of fd (false: a caller-owned buffer, nothing to unmap)
    std::vector<FrameIndex> frames;
    std::string err;
};


namespace {

thread_local std::string g_open_error;

const double kPow10[23] = {1e0,  1e1,  1e2,  1e3,  1e4,  1e5,  1e6,  1e7,  1e8,  1e9,  1e10, 1e11,
                           1e12, 1e13, 1e14, 1e15, 1e16, 1e17, 1e18, 1e19, 1e20, 1e21, 1e22};

inline const char *skip_ws(const char *p, const char *end)
{
    while (p < end && (*p == ' ' || *p == '\t' || *p == '\r')) ++p;
    return p;
}

// Parses one number starting at p (no leading whitespace); returns the position after it.
inline const char *parse_double(const char *p, const char *end, double *out)
{
    const char *start = p;
    bool neg = false;
    if (p < end && (*p == '-' || *p == '+')) {
        neg = *p == '-';
        ++p;
    }
    uint64_t m = 0;
    int digits = 0, frac = 0;  // significant digits held in m, digits after the decimal point held in m
    bool any = false, dropped = false;
    while (p < end && *p >= '0' && *p <= '9') {
        any = true;
        if (digits < 19) {
            m = m * 10 + (uint64_t)(*p - '0');
            if (m) ++digits;
        } else {
            dropped = true;
        }
        ++p;
    }
    if (p < end && *p == '.') {
        ++p;
        while (p < end && *p >= '0' && *p <= '9') {
            any = true;
            if (digits < 19) {
                m = m * 10 + (uint64_t)(*p - '0');
                if (m) ++digits;
                ++frac;
            } else {
                dropped = true;
            }
            ++p;
        }
    }
    const bool slow = dropped || !any;
    int e10 = 0;
    if (p < end && (*p == 'e' || *p == 'E')) {
        const char *q = p + 1;
        bool eneg = false;
        if (q < end && (*q == '-' || *q == '+')) {
            eneg = *q == '-';
            ++q;
        }
        if (q < end && *q >= '0' && *q <= '9') {
            int ev = 0;
            while (q < end && *q >= '0' && *q <= '9') {
                if (ev < 100000) ev = ev * 10 + (*q - '0');
                ++q;
            }
            e10 = eneg ? -ev : ev;
            p = q;
        }
    }
    if (!slow) {
        const int e = e10 - frac;
        if (m < (1ULL << 53) && e >= -22 && e <= 22) {
            double v = (double)m;
            v = e < 0 ? v / kPow10[-e] : v * kPow10[e];
            *out = neg ? -v : v;
            return p;
        }
    }
    // general case (long mantissa, huge exponent, nan/inf, garbage): glibc strtod on a bounded copy
    char buf[96];
    size_t len = (size_t)(p - start);
    if (!any) {  // not a plain number: take the whole token
        const char *q = start;
        while (q < end && *q != ' ' && *q != '\t' && *q != '\n' && *q != '\r') ++q;
        len = (size_t)(q - start);
        p = q;
    }
    if (len >= sizeof buf) len = sizeof buf - 1;
    memcpy(buf, start, len);
    buf[len] = 0;
    *out = strtod(buf, nullptr);
    return p;
}

inline const char *line_end(const char *p, const char *end)
{
    const char *nl = (const char *)memchr(p, '\n', (size_t)(end - p));
    return nl ? nl : end;
}

inline bool starts_with(const char *p, const char *end, const char *lit)
{
    const size_t n = strlen(lit);
    return (size_t)(end - p) >= n && memcmp(p, lit, n) == 0;
}

int index_frames(mdhip_dump *d)
{
    const char *p = d->data, *end = d->data + d->size;
    while (p < end) {
        const char *le = line_end(p, end);
        if (!starts_with(p, le, "ITEM: TIMESTEP")) {
            p = le < end ? le + 1 : end;
            continue;
        }
        FrameIndex fr;
        auto next_line = [&](const char *&b, const char *&e) -> bool {
            if (le >= end) return false;
            b = le + 1;
            e = line_end(b, end);
            le = e;
            return true;
        };
        const char *b, *e;
        double v;
        if (!next_line(b, e)) break;
        parse_double(skip_ws(b, e), e, &v);
        if (!(std::fabs(v) < 9.2e18)) {  // (also NaN: the cast below would be undefined)
            d->err = "dump: TIMESTEP is not an integer";
            return MDHIP_EINVAL;
        }
        fr.timestep = (int64_t)v;
        if (!next_line(b, e) || !starts_with(b, e, "ITEM: NUMBER OF ATOMS")) {
            d->err = "dump: expected ITEM: NUMBER OF ATOMS";
            return MDHIP_EINVAL;
        }
        if (!next_line(b, e)) break;
        parse_double(skip_ws(b, e), e, &v);
        if (!(v >= 0.0 && v <= (double)d->size)) {  // negative, NaN, or more atoms than the file has bytes
            d->err = "dump: bad NUMBER OF ATOMS";
            return MDHIP_EINVAL;
        }
        fr.natoms = (int64_t)v;
        if (!next_line(b, e) || !starts_with(b, e, "ITEM: BOX BOUNDS")) {
            d->err = "dump: expected ITEM: BOX BOUNDS";
            return MDHIP_EINVAL;
        }
        fr.triclinic = std::string(b, e).find(" xy ") != std::string::npos ? 1 : 0;
        for (int ax = 0; ax < 3; ++ax) {
            if (!next_line(b, e)) break;
            const char *q = skip_ws(b, e);
            q = parse_double(q, e, &fr.bounds[2 * ax]);
            q = parse_double(skip_ws(q, e), e, &fr.bounds[2 * ax + 1]);
            if (fr.triclinic) parse_double(skip_ws(q, e), e, &fr.tilt[ax]);
        }
        if (!next_line(b, e) || !starts_with(b, e, "ITEM: ATOMS")) {
            d->err = "dump: expected ITEM: ATOMS";
            return MDHIP_EINVAL;
        }
        {
            const char *q = b + strlen("ITEM: ATOMS");
            std::string cols;
            int n = 0;
            while (q < e) {
                q = skip_ws(q, e);
                const char *t = q;
                while (q < e && *q != ' ' && *q != '\t' && *q != '\r') ++q;
                if (q > t) {
                    if (n) cols += ' ';
                    cols.append(t, q);
                    ++n;
                }
            }
            fr.columns = cols;
            fr.n_cols = n;
        }
        fr.body_begin = le < end ? (size_t)(le + 1 - d->data) : d->size;
        // the body is exactly natoms lines
        const char *q = d->data + fr.body_begin;
        for (int64_t k = 0; k < fr.natoms && q < end; ++k) {
            const char *l2 = line_end(q, end);
            q = l2 < end ? l2 + 1 : end;
        }
        fr.body_end = (size_t)(q - d->data);
        d->frames.push_back(fr);
        p = q;
    }
    return MDHIP_OK;
}

// parse the lines [l0, l1) of a frame body into rows (sel columns), row-major scratch [line][n_sel (+1 key)]
// Returns the number of malformed rows: fewer than n_cols tokens, or a token that is not a number (pandas would give
// NaN / object columns and pymatgen's callers would fail later; here the read fails loudly instead of filling zeros).
int64_t parse_lines(const char *p, const char *end, int64_t n_lines, int n_cols, int n_sel, const int *col_idx,
                    int key_col, double *vals, double *keys)
{
    int64_t bad = 0;
    std::vector<int> slot(n_cols, -1);
    for (int s = 0; s < n_sel; ++s) slot[col_idx[s]] = s;  // a column selected twice keeps the last slot
    for (int64_t k = 0; k < n_lines && p < end; ++k) {
        const char *le = line_end(p, end);
        const char *q = p;
        for (int c = 0; c < n_cols; ++c) {
            q = skip_ws(q, le);
            if (q >= le) {
                ++bad;
                break;
            }
            if (slot[c] < 0 && c != key_col) {
                // a column nobody asked for may hold text (`element`, pandas reads it as an object column and the
                // reference never touches it): skip the token, numeric or not
                while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;
                continue;
            }
            const char ch = *q;
            if (!((ch >= '0' && ch <= '9') || ch == '-' || ch == '+' || ch == '.' || ch == 'n' || ch == 'N' || ch == 'i' ||
                  ch == 'I'))
                ++bad;
            double v;
            q = parse_double(q, le, &v);
            if (slot[c] >= 0) vals[k * n_sel + slot[c]] = v;
            if (c == key_col) keys[k] = v;
        }
        // duplicated selections
        for (int s = 0; s < n_sel; ++s)
            if (slot[col_idx[s]] != s) vals[k * n_sel + s] = vals[k * n_sel + slot[col_idx[s]]];
        p = le < end ? le + 1 : end;
    }
    return bad;
}


// One pass over a frame body with DIRECT placement: every line's selected tokens are parsed into a small local row and
// stored at their destination row at once — the row of the line itself (key_col < 0) or id - 1 when the key column holds
// a permutation of 1..n (atom ids: what sort_values("id") of the reference amounts to, rdf_cn.py:192) — so that no
// scratch table, no second (scatter) pass and no line-count pass is needed. `seen` [n] must be zeroed by the caller.
// Returns 0 ok, 1 malformed row (fewer values than columns, a non-numeric token in a wanted column, fewer lines than
// atoms), 2 the keys are not a permutation of 1..n (the caller takes the general route: stable sort by key).
int parse_frame_direct(const char *p, const char *end, int64_t n, int n_cols, int n_sel, const int *col_idx, int key_col,
                       double *const *outs, unsigned char *seen)
{
    constexpr int MAXC = 64;
    if (n_cols > MAXC) return 2;
    int want[MAXC];  // 0: skip the token, 1: parse it
    for (int c = 0; c < n_cols; ++c) want[c] = c == key_col ? 1 : 0;
    int last = key_col;
    for (int s = 0; s < n_sel; ++s) {
        want[col_idx[s]] = 1;
        last = col_idx[s] > last ? col_idx[s] : last;
    }
    double tok[MAXC];
    for (int64_t k = 0; k < n; ++k) {
        if (p >= end) return 1;
        const char *le = line_end(p, end);
        const char *q = p;
        for (int c = 0; c < n_cols; ++c) {
            q = skip_ws(q, le);
            if (q >= le) return 1;
            if (!want[c]) {
                if (c > last) break;  // nothing wanted behind this column: the rest of the line is only counted below
                while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;
                continue;
            }
            const char ch = *q;
            if (!((ch >= '0' && ch <= '9') || ch == '-' || ch == '+' || ch == '.' || ch == 'n' || ch == 'N' || ch == 'i' ||
                  ch == 'I'))
                return 1;
            q = parse_double(q, le, &tok[c]);
        }
        if (last + 1 < n_cols) {
            // the columns behind the last wanted one must still be there (a short row is malformed)
            int c = last + 1;
            // (q stands behind token `last`, or at the start of token last + 1 when that one was reached above)
            for (;;) {
                q = skip_ws(q, le);
                if (q >= le) break;
                while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;
                ++c;
            }
            if (c < n_cols) return 1;
        }
        int64_t r = k;
        if (key_col >= 0) {
            const double v = tok[key_col];
            const int64_t id = (v >= 1.0 && v <= (double)n) ? (int64_t)v : 0;  // (range first: the cast of NaN is undefined)
            if (id < 1 || (double)id != v || seen[(size_t)id - 1]) return 2;
            seen[(size_t)id - 1] = 1;
            r = id - 1;
        }
        for (int s = 0; s < n_sel; ++s) outs[s][r] = tok[col_idx[s]];
        p = le < end ? le + 1 : end;
    }
    return 0;
}


// ---- LAMMPS log files (thermo tables) ------------------------------------------------------------
// Role of pymatgen's parse_lammps_log (not in the reference tree; call sites dynamical/viscosity.py:211,
// utilities/log.py:21): a run's thermo table sits between the line that starts with "Memory usage per
// processor =" / "Per MPI rank memory allocation" and the line that starts with "Loop time of"; its first
// line holds the column names. Lines that start with "WARNING" and blank lines are not rows.
struct LogRun {
    const char *body = nullptr;  // first line after the header
    const char *end = nullptr;   // start of the "Loop time of" line
    std::vector<std::string> names;
    int64_t n_rows = 0;
    bool regular = true;  // every row has exactly one numeric token per column
};

inline bool log_row_line(const char *p, const char *le)
{
    const char *q = skip_ws(p, le);
    return q < le && !starts_with(p, le, "WARNING");
}

// counts the rows of [p, end) and checks their shape
void log_scan(const char *p, const char *end, int n_cols, int64_t *rows, bool *regular)
{
    int64_t n = 0;
    bool ok = true;
    while (p < end) {
        const char *le = line_end(p, end);
        if (log_row_line(p, le)) {
            ++n;
            int tok = 0;
            const char *q = p;
            for (;;) {
                q = skip_ws(q, le);
                if (q >= le) break;
                const char c = *q;
                if (!((c >= '0' && c <= '9') || c == '-' || c == '+' || c == '.')) ok = false;  // text inside the table
                while (q < le && *q != ' ' && *q != '\t' && *q != '\r') ++q;
                ++tok;
            }
            if (tok != n_cols) ok = false;
        }
        p = le < end ? le + 1 : end;
    }
    *rows = n;
    *regular = ok;
}

// rows of [p, end) into column planes out[c * stride + row0 + k]; all_int[c] cleared when a token of column c
// is not a plain integer
void log_parse(const char *p, const char *end, int n_cols, int64_t row0, int64_t stride, double *out,
               unsigned char *all_int)
{
    int64_t k = row0;
    while (p < end) {
        const char *le = line_end(p, end);
        if (log_row_line(p, le)) {
            const char *q = p;
            for (int c = 0; c < n_cols; ++c) {
                q = skip_ws(q, le);
                if (q >= le) break;
                const char *t0 = q;
                double v;
                q = parse_double(q, le, &v);
                out[(size_t)c * stride + k] = v;
                const char *d = t0;
                if (d < q && (*d == '-' || *d == '+')) ++d;
                bool is_int = d < q;
                for (; d < q; ++d)
                    if (*d < '0' || *d > '9') is_int = false;
                if (!is_int) all_int[c] = 0;
            }
            ++k;
        }
        p = le < end ? le + 1 : end;
    }
}

// nothing may unwind across the C boundary
template <typename F>
static int guarded(F f)
{
    try {
        return f();
    } catch (const std::bad_alloc &) {
        return MDHIP_ENOMEM;
    } catch (...) {
        return MDHIP_EINVAL;
    }
}

}  // namespace

extern "C" {

// `into` != nullptr: the file is READ into that (reusable, thread-owned) buffer instead of being mapped. A worker that
// walks through hundreds of small files pays mmap + page faults + munmap per file otherwise, all of them serialised on
// the process's address-space lock: with 8 threads the batch reader ran SLOWER than with one; a pread from the page
// cache into a warm buffer touches no page table.
static int dump_open_impl(const char *path, mdhip_dump **out, std::vector<char> *into = nullptr)
{
    if (!path || !out) return MDHIP_EINVAL;
    *out = nullptr;
    mdhip_dump *d = new mdhip_dump();
    d->fd = open(path, O_RDONLY);
    if (d->fd < 0) {
        g_open_error = std::string("cannot open ") + path + ": " + strerror(errno);
        delete d;
        return MDHIP_EINVAL;
    }
    struct stat st;
    fstat(d->fd, &st);
    d->size = (size_t)st.st_size;
    if (into) {
        d->mapped = false;
        if (into->size() < d->size) into->resize(d->size + d->size / 4 + 4096);
        size_t got = 0;
        while (got < d->size) {
            const ssize_t r = pread(d->fd, into->data() + got, d->size - got, (off_t)got);
            if (r < 0 && errno == EINTR) continue;
            if (r <= 0) break;
            got += (size_t)r;
        }
        if (got != d->size) {
            g_open_error = std::string("short read of ") + path;
            close(d->fd);
            delete d;
            return MDHIP_EINVAL;
        }
        d->data = into->data();
    } else if (d->size) {
        void *m = mmap(nullptr, d->size, PROT_READ, MAP_PRIVATE, d->fd, 0);
        if (m == MAP_FAILED) {
            g_open_error = std::string("mmap failed for ") + path;
            close(d->fd);
            delete d;
            return MDHIP_ENOMEM;
        }
        d->data = (const char *)m;
        madvise(m, d->size, MADV_SEQUENTIAL);
    }
    int rc = index_frames(d);
    if (rc) {
        g_open_error = d->err;
        mdhip_dump_close(d);
        return rc;
    }
    *out = d;
    return MDHIP_OK;
}

void mdhip_dump_close(mdhip_dump *d)
{
    if (!d) return;
    if (d->data && d->mapped) munmap((void *)d->data, d->size);
    if (d->fd >= 0) close(d->fd);
    delete d;
}

const char *mdhip_dump_error(mdhip_dump *d) { return d ? d->err.c_str() : g_open_error.c_str(); }

int64_t mdhip_dump_n_frames(mdhip_dump *d) { return d ? (int64_t)d->frames.size() : -1; }

int mdhip_dump_frame_info(mdhip_dump *d, int64_t f, int64_t *timestep, int64_t *natoms, double *bounds6,
                          double *tilt3, int *triclinic, int *n_cols, char *columns, int columns_len)
{
    if (!d || f < 0 || f >= (int64_t)d->frames.size()) return MDHIP_EINVAL;
    const FrameIndex &fr = d->frames[f];
    if (timestep) *timestep = fr.timestep;
    if (natoms) *natoms = fr.natoms;
    if (bounds6) memcpy(bounds6, fr.bounds, sizeof fr.bounds);
    if (tilt3) memcpy(tilt3, fr.tilt, sizeof fr.tilt);
    if (triclinic) *triclinic = fr.triclinic;
    if (n_cols) *n_cols = fr.n_cols;
    if (columns && columns_len > 0) {
        strncpy(columns, fr.columns.c_str(), (size_t)columns_len - 1);
        columns[columns_len - 1] = 0;
    }
    return MDHIP_OK;
}

// outs[s] = destination plane [natoms] of selected column s (mdhip_dump_read: consecutive planes of one buffer;
// mdhip_dump_read_cols: anywhere, e.g. x, y, z straight into a page-locked staging slot)
static int dump_read_impl(mdhip_dump *d, int64_t f, int n_sel, const int32_t *col_idx, int sort_col,
                          double *const *outs, int n_threads)
{
    if (!d || f < 0 || f >= (int64_t)d->frames.size() || n_sel < 0 || (n_sel && (!col_idx || !outs)))
        return MDHIP_EINVAL;
    for (int s = 0; s < n_sel; ++s)
        if (!outs[s]) return MDHIP_EINVAL;
    const FrameIndex &fr = d->frames[f];
    for (int s = 0; s < n_sel; ++s)
        if (col_idx[s] < 0 || col_idx[s] >= fr.n_cols) {
            d->err = "mdhip_dump_read: column index out of range";
            return MDHIP_EINVAL;
        }
    if (sort_col >= fr.n_cols) {
        d->err = "mdhip_dump_read: sort column out of range";
        return MDHIP_EINVAL;
    }
    const int64_t n = fr.natoms;
    if (n == 0 || n_sel == 0) return MDHIP_OK;
    const char *body = d->data + fr.body_begin, *body_end = d->data + fr.body_end;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 4096) n_threads = 1;
    if (n_threads == 1) {
        // one thread: a single pass with direct placement (no scratch table, no thread start-up)
        std::vector<unsigned char> seen(sort_col >= 0 ? (size_t)n : 0, 0);
        std::vector<int> ci(col_idx, col_idx + n_sel);
        const int st = parse_frame_direct(body, body_end, n, fr.n_cols, n_sel, ci.data(), sort_col, outs, seen.data());
        if (st == 0) return MDHIP_OK;
        if (st == 1) {
            // which of the two messages: fewer lines than atoms, or a bad row
            int64_t lines = 0;
            for (const char *q = body; q < body_end;) {
                const char *le = line_end(q, body_end);
                ++lines;
                q = le < body_end ? le + 1 : body_end;
            }
            d->err = lines < n ? "mdhip_dump_read: frame body has fewer lines than NUMBER OF ATOMS"
                               : "mdhip_dump_read: a row has fewer values than columns or a value that is not a number";
            return MDHIP_EINVAL;
        }
        // st == 2: keys that are not a permutation of 1..n -> the general route below (stable sort by key)
    }
    // line offsets of the thread chunks (chunk boundaries by bytes, aligned to line starts, then counted)
    std::vector<const char *> cstart(n_threads + 1);
    std::vector<int64_t> cline(n_threads + 1, 0);
    cstart[0] = body;
    for (int t = 1; t < n_threads; ++t) {
        const char *q = body + (size_t)((body_end - body) * (double)t / n_threads);
        if (q < cstart[t - 1]) q = cstart[t - 1];
        const char *le = line_end(q, body_end);
        cstart[t] = le < body_end ? le + 1 : body_end;
    }
    cstart[n_threads] = body_end;
    std::vector<int64_t> nl(n_threads, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                int64_t c = 0;
                for (const char *q = cstart[t]; q < cstart[t + 1];) {
                    const char *le = line_end(q, cstart[t + 1]);
                    ++c;
                    q = le < cstart[t + 1] ? le + 1 : cstart[t + 1];
                }
                nl[t] = c;
            });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < n_threads; ++t) cline[t + 1] = cline[t] + nl[t];
    if (cline[n_threads] < n) {
        d->err = "mdhip_dump_read: frame body has fewer lines than NUMBER OF ATOMS";
        return MDHIP_EINVAL;
    }
    std::vector<double> vals((size_t)n * n_sel), keys(sort_col >= 0 ? (size_t)n : 0);
    std::vector<int64_t> bad(n_threads, 0);
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                const int64_t l0 = cline[t], l1 = std::min<int64_t>(cline[t + 1], n);
                if (l1 > l0)
                    bad[t] = parse_lines(cstart[t], cstart[t + 1], l1 - l0, fr.n_cols, n_sel, col_idx, sort_col,
                                         vals.data() + (size_t)l0 * n_sel, sort_col >= 0 ? keys.data() + l0 : nullptr);
            });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < n_threads; ++t)
        if (bad[t]) {
            d->err = "mdhip_dump_read: a row has fewer values than columns or a value that is not a number";
            return MDHIP_EINVAL;
        }
    // destination row of every line: ascending key (stable), fast path when keys are a permutation of 1..n
    std::vector<int64_t> dest(n);
    if (sort_col < 0) {
        for (int64_t k = 0; k < n; ++k) dest[k] = k;
    } else {
        bool perm = true;
        std::vector<char> seen((size_t)n, 0);
        for (int64_t k = 0; k < n && perm; ++k) {
            const double v = keys[k];
            const int64_t id = (v >= 1.0 && v <= (double)n) ? (int64_t)v : 0;  // (range first: the cast of NaN is undefined)
            if (id < 1 || (double)id != v || seen[(size_t)id - 1])
                perm = false;
            else {
                seen[(size_t)id - 1] = 1;
                dest[k] = id - 1;
            }
        }
        if (!perm) {
            std::vector<int64_t> order(n);
            for (int64_t k = 0; k < n; ++k) order[k] = k;
            std::stable_sort(order.begin(), order.end(), [&](int64_t a, int64_t b) { return keys[a] < keys[b]; });
            for (int64_t r = 0; r < n; ++r) dest[order[r]] = r;
        }
    }
    // scatter into SoA planes out[s][row]
    for (int64_t k = 0; k < n; ++k) {
        const int64_t r = dest[k];
        for (int s = 0; s < n_sel; ++s) outs[s][r] = vals[(size_t)k * n_sel + s];
    }
    return MDHIP_OK;
}


// ---- log reader C-ABI ---------------------------------------------------------------------------

struct mdhip_log {
    int fd = -1;
    const char *data = nullptr;
    size_t size = 0;
    std::vector<LogRun> runs;
    std::string err;
};

static int log_open_impl(const char *path, mdhip_log **out)
{
    if (!path || !out) return MDHIP_EINVAL;
    *out = nullptr;
    mdhip_log *l = new mdhip_log();
    l->fd = open(path, O_RDONLY);
    if (l->fd < 0) {
        g_open_error = std::string("cannot open ") + path + ": " + strerror(errno);
        delete l;
        return MDHIP_EINVAL;
    }
    struct stat st;
    fstat(l->fd, &st);
    l->size = (size_t)st.st_size;
    if (l->size) {
        void *m = mmap(nullptr, l->size, PROT_READ, MAP_PRIVATE, l->fd, 0);
        if (m == MAP_FAILED) {
            g_open_error = std::string("mmap failed for ") + path;
            close(l->fd);
            delete l;
            return MDHIP_ENOMEM;
        }
        l->data = (const char *)m;
        madvise(m, l->size, MADV_SEQUENTIAL);
    }
    const char *p = l->data, *end = l->data + l->size;
    const char *begin = nullptr;
    while (p && p < end) {
        const char *le = line_end(p, end);
        const char *next = le < end ? le + 1 : end;
        if (starts_with(p, le, "Memory usage per processor =") || starts_with(p, le, "Per MPI rank memory allocation")) {
            begin = next;
        } else if (begin && starts_with(p, le, "Loop time of")) {
            LogRun r;
            // header = first row-like line of the block
            const char *q = begin;
            while (q < p) {
                const char *qe = line_end(q, p);
                if (log_row_line(q, qe)) {
                    const char *t = q;
                    for (;;) {
                        t = skip_ws(t, qe);
                        if (t >= qe) break;
                        const char *t0 = t;
                        while (t < qe && *t != ' ' && *t != '\t' && *t != '\r') ++t;
                        r.names.emplace_back(t0, (size_t)(t - t0));
                    }
                    q = qe < p ? qe + 1 : p;
                    break;
                }
                q = qe < p ? qe + 1 : p;
            }
            if (!r.names.empty()) {
                r.body = q;
                r.end = p;
                log_scan(r.body, r.end, (int)r.names.size(), &r.n_rows, &r.regular);
                l->runs.push_back(std::move(r));
            }
            begin = nullptr;
        }
        p = next;
    }
    *out = l;
    return MDHIP_OK;
}

void mdhip_log_close(mdhip_log *l)
{
    if (!l) return;
    if (l->data) munmap((void *)l->data, l->size);
    if (l->fd >= 0) close(l->fd);
    delete l;
}

const char *mdhip_log_error(mdhip_log *l) { return l ? l->err.c_str() : g_open_error.c_str(); }

int64_t mdhip_log_n_runs(mdhip_log *l) { return l ? (int64_t)l->runs.size() : -1; }

int mdhip_log_run_info(mdhip_log *l, int64_t run, int64_t *n_rows, int *n_cols, int *regular, char *names,
                       int names_len)
{
    if (!l || run < 0 || run >= (int64_t)l->runs.size()) return MDHIP_EINVAL;
    const LogRun &r = l->runs[(size_t)run];
    if (n_rows) *n_rows = r.n_rows;
    if (n_cols) *n_cols = (int)r.names.size();
    if (regular) *regular = r.regular ? 1 : 0;
    if (names && names_len > 0) {
        std::string joined;
        for (size_t k = 0; k < r.names.size(); ++k) joined += (k ? " " : "") + r.names[k];
        if ((int)joined.size() >= names_len) {
            l->err = "column names do not fit the buffer";
            return MDHIP_ELIMIT;
        }
        memcpy(names, joined.c_str(), joined.size() + 1);
    }
    return MDHIP_OK;
}

static int log_read_impl(mdhip_log *l, int64_t run, double *out, int32_t *is_int, int n_threads)
{
    if (!l || run < 0 || run >= (int64_t)l->runs.size() || !out) return MDHIP_EINVAL;
    const LogRun &r = l->runs[(size_t)run];
    if (!r.regular) {
        l->err = "thermo table has rows that are not one number per column";
        return MDHIP_EINVAL;
    }
    const int n_cols = (int)r.names.size();
    const int64_t n = r.n_rows;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 64) n_threads = 64;
    if (n < 8192) n_threads = 1;
    // byte ranges cut at line starts, rows counted per range, then parsed in place
    std::vector<const char *> cut(n_threads + 1);
    cut[0] = r.body;
    for (int t = 1; t < n_threads; ++t) {
        const char *q = r.body + (size_t)((r.end - r.body) * (double)t / n_threads);
        const char *le = line_end(q, r.end);
        cut[t] = le < r.end ? le + 1 : r.end;
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    cut[n_threads] = r.end;
    std::vector<int64_t> rows(n_threads, 0), row0(n_threads + 1, 0);
    std::vector<std::vector<unsigned char>> ints(n_threads, std::vector<unsigned char>((size_t)n_cols, 1));
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] {
                bool reg;
                log_scan(cut[t], cut[t + 1], n_cols, &rows[t], &reg);
            });
        for (auto &x : th) x.join();
    }
    for (int t = 0; t < n_threads; ++t) row0[t + 1] = row0[t] + rows[t];
    if (row0[n_threads] != n) {
        l->err = "row count changed between index and read";
        return MDHIP_EINVAL;
    }
    {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t)
            th.emplace_back([&, t] { log_parse(cut[t], cut[t + 1], n_cols, row0[t], n, out, ints[t].data()); });
        for (auto &x : th) x.join();
    }
    if (is_int)
        for (int c = 0; c < n_cols; ++c) {
            int32_t a = 1;
            for (int t = 0; t < n_threads; ++t) a = a && ints[t][(size_t)c];
            is_int[c] = n > 0 ? a : 0;
        }
    return MDHIP_OK;
}

// ---- exception barriers: nothing may unwind across the C boundary (a std::bad_alloc from a vector sized by a
// corrupt header would otherwise call std::terminate in the caller's process) ---------------------------------------
int mdhip_dump_open(const char *path, mdhip_dump **out)
{
    return guarded([&] { return dump_open_impl(path, out); });
}

int mdhip_dump_read(mdhip_dump *d, int64_t f, int n_sel, const int32_t *col_idx, int sort_col, double *out, int n_threads)
{
    return guarded([&] {
        if (!d || f < 0 || f >= (int64_t)d->frames.size() || n_sel < 0 || (n_sel && !out)) return (int)MDHIP_EINVAL;
        std::vector<double *> outs((size_t)n_sel);
        for (int s = 0; s < n_sel; ++s) outs[s] = out + (size_t)s * (size_t)d->frames[f].natoms;
        return dump_read_impl(d, f, n_sel, col_idx, sort_col, outs.data(), n_threads);
    });
}

int mdhip_dump_read_cols(mdhip_dump *d, int64_t f, int n_sel, const int32_t *col_idx, int sort_col, double *const *outs,
                         int n_threads)
{
    return guarded([&] { return dump_read_impl(d, f, n_sel, col_idx, sort_col, outs, n_threads); });
}

// ---- many single-frame files in one call --------------------------------------------------------------
static int split_names(const std::string &cols, std::vector<std::string> &out)
{
    out.clear();
    size_t i = 0;
    while (i < cols.size()) {
        size_t j = cols.find(' ', i);
        if (j == std::string::npos) j = cols.size();
        if (j > i) out.emplace_back(cols.substr(i, j - i));
        i = j + 1;
    }
    return (int)out.size();
}

static int dump_read_files_impl(const char *const *paths, int n_files, int n_sel, const char *const *col_names,
                                const char *sort_name, int64_t n_atoms, double *const *dst, const int64_t *dst_stride,
                                int64_t *timesteps, double *bounds6, double *tilt3, int32_t *triclinic, int n_threads,
                                char *err, int err_len, int cmp_sel, const double *cmp_ref, int32_t *cmp_equal)
{
    if (err && err_len > 0) err[0] = 0;
    if (n_files < 0 || n_sel < 0 || n_atoms < 0 || (n_files && !paths) || (n_sel && (!col_names || !dst || !dst_stride)))
        return MDHIP_EINVAL;
    if (n_files == 0) return MDHIP_OK;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > n_files) n_threads = n_files;
    if (n_threads > 64) n_threads = 64;
    std::atomic<int> next{0};
    std::atomic<int> status{MDHIP_OK};  // first failure wins
    std::string first_error;
    std::mutex err_mu;
    auto fail = [&](int code, const std::string &msg) {
        int expect = MDHIP_OK;
        if (status.compare_exchange_strong(expect, code)) {
            std::lock_guard<std::mutex> g(err_mu);
            first_error = msg;
        }
    };
    auto work = [&] {
        std::vector<unsigned char> seen;
        std::vector<std::string> names;
        std::vector<int> ci((size_t)n_sel);
        std::vector<double *> outs((size_t)n_sel);
        std::vector<char> text;  // this worker's file buffer, reused from file to file
        for (;;) {
            const int k = next.fetch_add(1);
            if (k >= n_files || status.load() != MDHIP_OK) return;
            mdhip_dump *d = nullptr;
            const int rc = dump_open_impl(paths[k], &d, &text);
            if (rc) {
                fail(rc, g_open_error);
                return;
            }
            struct Closer {
                mdhip_dump *d;
                ~Closer() { mdhip_dump_close(d); }
            } closer{d};
            if (d->frames.size() != 1 || d->frames[0].natoms != n_atoms) {
                fail(1, std::string(paths[k]) + ": not one frame of the expected atom count");
                return;
            }
            const FrameIndex &fr = d->frames[0];
            split_names(fr.columns, names);
            int key = -1;
            bool ok = true;
            for (int s = 0; s < n_sel && ok; ++s) {
                ci[s] = -1;
                for (int c = 0; c < (int)names.size(); ++c)
                    if (names[c] == col_names[s]) ci[s] = c;
                ok = ci[s] >= 0;
                outs[s] = dst[s] + (size_t)k * (size_t)dst_stride[s];
            }
            if (ok && sort_name) {
                for (int c = 0; c < (int)names.size(); ++c)
                    if (names[c] == sort_name) key = c;
                ok = key >= 0;
            }
            if (!ok) {
                fail(1, std::string(paths[k]) + ": a requested column is missing");
                return;
            }
            if (timesteps) timesteps[k] = fr.timestep;
            if (bounds6) memcpy(bounds6 + (size_t)k * 6, fr.bounds, 48);
            if (tilt3) memcpy(tilt3 + (size_t)k * 3, fr.tilt, 24);
            if (triclinic) triclinic[k] = fr.triclinic;
            if (n_atoms == 0 || n_sel == 0) continue;
            seen.assign(key >= 0 ? (size_t)n_atoms : 0, 0);
            const char *body = d->data + fr.body_begin, *body_end = d->data + fr.body_end;
            int st = parse_frame_direct(body, body_end, n_atoms, fr.n_cols, n_sel, ci.data(), key, outs.data(), seen.data());
            if (st == 2) {  // ids that are not a permutation of 1..n: the general route (stable sort by key)
                std::vector<int32_t> ci32(ci.begin(), ci.end());
                st = dump_read_impl(d, 0, n_sel, ci32.data(), key, outs.data(), 1) == MDHIP_OK ? 0 : 1;
            }
            if (st) {
                fail(MDHIP_EINVAL, std::string(paths[k]) + ": " + (d->err.empty() ? "a row has fewer values than columns or a value that is not a number" : d->err));
                return;
            }
            // (the plane is still in this core's cache: the comparison the caller would otherwise make in numpy, one
            // file at a time on one thread, costs nothing here)
            if (cmp_equal && cmp_ref && cmp_sel >= 0 && cmp_sel < n_sel)
                cmp_equal[k] = memcmp(outs[cmp_sel], cmp_ref, (size_t)n_atoms * 8) == 0 ? 1 : 0;
        }
    };
    if (n_threads == 1) {
        work();
    } else {
        std::vector<std::thread> th;
        for (int t = 0; t < n_threads; ++t) th.emplace_back(work);
        for (auto &x : th) x.join();
    }
    const int rc = status.load();
    if (rc != MDHIP_OK && err && err_len > 0) {
        strncpy(err, first_error.c_str(), (size_t)err_len - 1);
        err[err_len - 1] = 0;
    }
    if (rc == MDHIP_OK && cmp_equal && !cmp_ref && cmp_sel >= 0 && cmp_sel < n_sel) {
        // no reference given: every file against the first one of this call
        const double *first = dst[cmp_sel];
        cmp_equal[0] = 1;
        std::atomic<int> nk{1};
        auto cmp = [&] {
            for (;;) {
                const int k = nk.fetch_add(1);
                if (k >= n_files) return;
                cmp_equal[k] = memcmp(dst[cmp_sel] + (size_t)k * (size_t)dst_stride[cmp_sel], first, (size_t)n_atoms * 8) == 0;
            }
        };
        if (n_threads == 1 || n_files < 4) {
            cmp();
        } else {
            std::vector<std::thread> th;
            for (int t = 0; t < n_threads; ++t) th.emplace_back(cmp);
            for (auto &x : th) x.join();
        }
    }
    return rc;
}

int mdhip_dump_read_files(const char *const *paths, int n_files, int n_sel, const char *const *col_names,
                          const char *sort_name, int64_t n_atoms, double *const *dst, const int64_t *dst_stride,
                          int64_t *timesteps, double *bounds6, double *tilt3, int32_t *triclinic, int n_threads,
                          char *err, int err_len, int cmp_sel, const double *cmp_ref, int32_t *cmp_equal)
{
    return guarded([&] {
        return dump_read_files_impl(paths, n_files, n_sel, col_names, sort_name, n_atoms, dst, dst_stride, timesteps,
                                    bounds6, tilt3, triclinic, n_threads, err, err_len, cmp_sel, cmp_ref, cmp_equal);
    });
}

int mdhip_log_open(const char *path, mdhip_log **out)
{
    return guarded([&] { return log_open_impl(path, out); });
}

int mdhip_log_read(mdhip_log *l, int64_t run, double *out, int32_t *is_int, int n_threads)
{
    return guarded([&] { return log_read_impl(l, run, out, is_int, n_threads); });
}

}  // extern "C"
