// dig_tsv.hip -- the result-file writer of the drop-in (host code only; no kernel).
//
// DigDriver.py writes a result frame with `df.to_csv(path, header=True, index=True, sep="\t")` (DigDriver.py:115-118,
// DigPretrain.py likewise): pandas formats every float with Python's repr -- the shortest decimal string that reads back
// to the same double, positional for 1e-4 <= |x| < 1e16, d.ddde-XX otherwise, "12.0" for whole numbers -- one Python call
// per cell: 2.0 s per 120 091-row frame of 24 columns, 75 s for the 37 cohorts of BASELINE configs[2] (tools/e2e_bench.py:
// 20x everything else in the pipeline together).  dig_write_tsv_host produces the same bytes from the column arrays:
// std::to_chars gives the shortest round-trip digits, the layout rules of float_repr_style 'short' are applied here; rows
// are formatted in chunks by a few threads and written in order.
#include <fcntl.h>
#include <unistd.h>

#include <charconv>
#include <cmath>
#include <cstring>
#include <memory>
#include <string>
#include <thread>
#include <vector>

#include "dig_common.hpp"

namespace {

// repr(float) of CPython (Python/pystrtod.c format_float_short, mode 'r'): NaN -> na_rep (pandas writes '' by default)
inline char* put_float(char* p, double v)
{
    if (std::isnan(v)) return p;
    if (std::isinf(v)) {
        if (v < 0) *p++ = '-';
        std::memcpy(p, "inf", 3);
        return p + 3;
    }
    if (v == 0.0) {
        if (std::signbit(v)) *p++ = '-';
        std::memcpy(p, "0.0", 3);
        return p + 3;
    }
    char buf[40];
    const auto r = std::to_chars(buf, buf + sizeof(buf), v, std::chars_format::scientific);      // [-]d[.ddd]e[+-]XX, shortest digits
    const char* s = buf;
    if (*s == '-') *p++ = *s++;
    char digits[24];
    int nd = 0;
    digits[nd++] = *s++;
    if (*s == '.') {
        ++s;
        while (*s != 'e') digits[nd++] = *s++;
    }
    ++s;                                                   // 'e'
    const bool eneg = *s == '-';
    ++s;
    int e10 = 0;
    while (s < r.ptr) e10 = e10 * 10 + (*s++ - '0');
    if (eneg) e10 = -e10;
    const int decpt = e10 + 1;                             // position of the decimal point relative to the digit string
    if (decpt > 16 || decpt < -3) {                        // exponent form: d[.ddd]e[+-]XX, at least two exponent digits
        *p++ = digits[0];
        if (nd > 1) {
            *p++ = '.';
            std::memcpy(p, digits + 1, (size_t)(nd - 1));
            p += nd - 1;
        }
        *p++ = 'e';
        int e = decpt - 1;
        *p++ = e < 0 ? '-' : '+';
        if (e < 0) e = -e;
        char eb[8];
        int ne = 0;
        do {
            eb[ne++] = (char)('0' + e % 10);
            e /= 10;
        } while (e);
        if (ne < 2) eb[ne++] = '0';
        while (ne) *p++ = eb[--ne];
        return p;
    }
    if (decpt <= 0) {                                      // 0.000ddd
        *p++ = '0';
        *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        std::memcpy(p, digits, (size_t)nd);
        return p + nd;
    }
    if (decpt >= nd) {                                     // ddd000.0
        std::memcpy(p, digits, (size_t)nd);
        p += nd;
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.';
        *p++ = '0';
        return p;
    }
    std::memcpy(p, digits, (size_t)decpt);                 // dd.ddd
    p += decpt;
    *p++ = '.';
    std::memcpy(p, digits + decpt, (size_t)(nd - decpt));
    return p + (nd - decpt);
}

inline char* put_int(char* p, int64_t v)
{
    const auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}

}  // namespace

extern "C" {

/* One tab-separated text file: `header` (a complete first line, without the newline), then n_rows rows
 *   label <TAB> col_0 <TAB> ... <TAB> col_{n_cols-1}
 * labels: the row labels as one UTF-8 blob, label r = bytes label_off[r] .. label_off[r + 1] - 1.
 * col_kind[j]: 0 = float64 (written as Python's repr; NaN as the empty field, +-inf as inf / -inf: what DataFrame.to_csv
 * writes), 1 = int64, 2 = bool as uint8 (True / False).  Returns DIG_OK, or DIG_EINVAL with dig_last_error(). */
int dig_write_tsv_host(const char* path, const char* header, const char* labels, const int64_t* label_off, int64_t n_rows,
                       int n_cols, const void* const* col_ptr, const int* col_kind, int n_threads)
{
    DIG_REQUIRE(path && header && n_rows >= 0 && n_cols >= 0, "path, header, non-negative sizes");
    DIG_REQUIRE(n_rows == 0 || (labels && label_off), "row labels");
    DIG_REQUIRE(n_cols == 0 || (col_ptr && col_kind), "columns");
    for (int j = 0; j < n_cols; ++j) DIG_REQUIRE(col_ptr[j] && col_kind[j] >= 0 && col_kind[j] <= 2, "column pointers and kinds (0 f64, 1 i64, 2 bool)");
    const int fd = ::open(path, O_WRONLY | O_CREAT | O_TRUNC, 0666);
    if (fd < 0) return ::dig::set_error(DIG_EINVAL, "dig_write_tsv_host: cannot open %s for writing", path);
    // One thread: chunks of 4 096 rows through one scratch buffer, written as soon as each is full.  Several threads: thread t
    // formats ONE contiguous range of rows into ONE buffer of its own (never cleared: a value-initialised buffer of the
    // worst-case size was 77 MB of zeroing and page faults per file), the byte counts give the ranges' offsets, then every
    // thread writes its range where it belongs.  (Round 4 gave every 4 096-row chunk a buffer of its own: with several files
    // written side by side the hundreds of mappings and unmappings contended for the process's address space -- 8 files x 8
    // threads took longer than 37 x 1; one mapping per thread does not.)
    const int64_t chunk = 4096;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if ((int64_t)n_threads * chunk > n_rows) n_threads = (int)std::max<int64_t>(1, n_rows / chunk);
    const bool seekable = ::lseek(fd, 0, SEEK_CUR) != (off_t)-1;      // (a pipe or a terminal: one thread, plain writes in order)
    if (!seekable) n_threads = 1;
    auto put = [&](const char* src, int64_t n, int64_t at) {
        while (n > 0) {
            const ssize_t w = seekable ? ::pwrite(fd, src, (size_t)n, (off_t)at) : ::write(fd, src, (size_t)n);
            if (w <= 0) return false;
            src += w;
            n -= w;
            at += w;
        }
        return true;
    };
    const int64_t head = (int64_t)std::strlen(header);
    auto cap = [&](int64_t r0, int64_t r1) { return (size_t)((label_off[r1] - label_off[r0]) + (r1 - r0) * (1 + (int64_t)n_cols * 26) + 64); };
    auto format_rows = [&](int64_t r0, int64_t r1, char* p) -> char* {
        for (int64_t r = r0; r < r1; ++r) {
            const int64_t ln = label_off[r + 1] - label_off[r];
            std::memcpy(p, labels + label_off[r], (size_t)ln);
            p += ln;
            for (int j = 0; j < n_cols; ++j) {
                *p++ = '\t';
                if (col_kind[j] == 0) p = put_float(p, static_cast<const double*>(col_ptr[j])[r]);
                else if (col_kind[j] == 1) p = put_int(p, static_cast<const int64_t*>(col_ptr[j])[r]);
                else {
                    const bool b = static_cast<const uint8_t*>(col_ptr[j])[r] != 0;
                    std::memcpy(p, b ? "True" : "False", b ? 4 : 5);
                    p += b ? 4 : 5;
                }
            }
            *p++ = '\n';
        }
        return p;
    };
    bool ok = put(header, head, 0) && put("\n", 1, head);
    if (n_threads == 1) {
        int64_t max_cap = 0;
        for (int64_t r0 = 0; r0 < n_rows; r0 += chunk) max_cap = std::max<int64_t>(max_cap, (int64_t)cap(r0, std::min(r0 + chunk, n_rows)));
        std::unique_ptr<char[]> scratch(n_rows ? new char[(size_t)max_cap] : nullptr);
        int64_t at = head + 1;
        for (int64_t r0 = 0; r0 < n_rows && ok; r0 += chunk) {
            const int64_t r1 = std::min(r0 + chunk, n_rows);
            const int64_t nb = (int64_t)(format_rows(r0, r1, scratch.get()) - scratch.get());
            ok = put(scratch.get(), nb, at);
            at += nb;
        }
    } else {
        std::vector<std::unique_ptr<char[]>> text((size_t)n_threads);
        std::vector<int64_t> bytes((size_t)n_threads, 0), offset((size_t)n_threads + 1, head + 1);
        std::vector<char> good((size_t)n_threads, 1);
        auto rows_of = [&](int t) { return std::make_pair(n_rows * t / n_threads, n_rows * (t + 1) / n_threads); };
        auto format = [&](int t) {
            const auto [r0, r1] = rows_of(t);
            text[(size_t)t].reset(new char[cap(r0, r1)]);
            bytes[(size_t)t] = (int64_t)(format_rows(r0, r1, text[(size_t)t].get()) - text[(size_t)t].get());
        };
        auto write_out = [&](int t) {
            if (!put(text[(size_t)t].get(), bytes[(size_t)t], offset[(size_t)t])) good[(size_t)t] = 0;
            text[(size_t)t].reset();
        };
        auto run = [&](auto&& fn) {
            std::vector<std::thread> pool;
            for (int t = 1; t < n_threads; ++t) pool.emplace_back(fn, t);
            fn(0);
            for (auto& th : pool) th.join();
        };
        run(format);
        for (int t = 0; t < n_threads; ++t) offset[(size_t)t + 1] = offset[(size_t)t] + bytes[(size_t)t];
        run(write_out);
        for (char g : good) ok = ok && g;
    }
    ok = (::close(fd) == 0) && ok;
    if (!ok) return ::dig::set_error(DIG_EINVAL, "dig_write_tsv_host: short write to %s", path);
    return DIG_OK;
}

}  // extern "C"

// =======================================================================================
// Annotated mutation files (ABI 7; host code only).
//
// The reference reads a cohort's mutations with pandas (mutation_tools.read_mutation_file, mutation_tools.py:45-104: tab-
// separated, no header, CHROM START END REF ALT SAMPLE GENE ANNOT and any further columns) and hands the frame on; the
// many-cohort driver (driver_model/cohort_batch.py) needs them as integer arrays for the interval join on the device.
// pyarrow parses a 300 000-row file in 40 ms, but the dictionary / numpy steps behind it hold the interpreter lock: 37 files
// side by side on 256 cores took 0.83 s (tools/parse_probe.py: 0.1 s of work per file, 11x inflated when run together).
// dig_mutation_file_parse_host does the whole of tabulate_gpu.encode_mutation_file for one file without the interpreter:
//   * rows whose CHROM, with one leading "chr" removed, is not one of "1" .. "22" are dropped;
//   * REF, ALT, SAMPLE, GENE, ANNOT become ids in order of first appearance (over all rows, as a dictionary encoding does);
//     the samples are then renumbered in order of first appearance among the KEPT rows (pandas.factorize on the kept frame);
//   * uid = dense rank of (CHROM, START, END, REF id, ALT id) among the kept rows, in sorted order;
//   * indel = (ANNOT == "INDEL").
// The arrays are bit-identical to the Python path's (tests/test_host_tools.py).  Content the parser does not cover -- a double
// quote anywhere (pyarrow would treat it as quoting), a START / END that is not a plain integer, rows with different numbers
// of fields -- is reported as *n_rows = -1 with no handle: the caller falls back to the Python path, which raises or parses
// as before.
// =======================================================================================
#include <algorithm>
#include <cstdio>
#include <string_view>
#include <unordered_map>

namespace {

struct MutFile {
    std::vector<int64_t> chrom, start, end, uid, sample, indel, gene;
    std::string sample_names;      // '\n'-joined, in id order
    int64_t n_samples = 0;
};

struct Interner {
    std::unordered_map<std::string_view, int64_t> ids;
    std::vector<std::string_view> labels;
    std::string_view last{};
    int64_t last_id = -1;
    int64_t get(std::string_view s)
    {
        if (last_id >= 0 && s == last) return last_id;
        auto it = ids.find(s);
        int64_t id;
        if (it == ids.end()) {
            id = (int64_t)labels.size();
            ids.emplace(s, id);
            labels.push_back(s);
        } else
            id = it->second;
        last = s;
        last_id = id;
        return id;
    }
};

inline bool parse_i64(std::string_view s, int64_t& out)
{
    if (s.empty()) return false;
    size_t i = 0;
    bool neg = false;
    if (s[0] == '-') {
        neg = true;
        i = 1;
        if (s.size() == 1) return false;
    }
    if (s.size() - i > 18) return false;
    int64_t v = 0;
    for (; i < s.size(); ++i) {
        const unsigned d = (unsigned)(s[i] - '0');
        if (d > 9u) return false;
        v = v * 10 + (int64_t)d;
    }
    out = neg ? -v : v;
    return true;
}

inline int64_t autosome(std::string_view s)      // "1" .. "22" -> 1 .. 22, else -1 (one leading "chr" allowed)
{
    if (s.size() >= 3 && s[0] == 'c' && s[1] == 'h' && s[2] == 'r') s.remove_prefix(3);
    if (s.size() == 1 && s[0] >= '1' && s[0] <= '9') return s[0] - '0';
    if (s.size() == 2 && s[0] >= '1' && s[0] <= '2' && s[1] >= '0' && s[1] <= '9') {
        const int v = 10 * (s[0] - '0') + (s[1] - '0');
        return v <= 22 ? v : -1;
    }
    return -1;
}

}  // namespace

extern "C" {

int dig_mutation_file_parse_host(const char* path, void** handle, int64_t* n_rows, int64_t* n_samples, int64_t* names_bytes)
{
    DIG_REQUIRE(path && handle && n_rows && n_samples && names_bytes, "non-null arguments");
    *handle = nullptr;
    *n_rows = -1;
    *n_samples = 0;
    *names_bytes = 0;
    FILE* f = fopen(path, "rb");
    if (!f) return dig::set_error(DIG_EINVAL, "dig_mutation_file_parse_host: cannot open %s", path);
    std::string buf;
    {
        fseek(f, 0, SEEK_END);
        const long sz = ftell(f);
        fseek(f, 0, SEEK_SET);
        buf.resize(sz > 0 ? (size_t)sz : 0);
        const size_t got = buf.empty() ? 0 : fread(&buf[0], 1, buf.size(), f);
        fclose(f);
        if (got != buf.size()) return dig::set_error(DIG_EINVAL, "dig_mutation_file_parse_host: short read of %s", path);
    }
    if (memchr(buf.data(), '"', buf.size())) return DIG_OK;          // quoting: not covered (caller falls back)
    Interner ref, alt, samp, gene, annot;
    std::vector<int64_t> ch, st, en, r_id, a_id, s_id, g_id, n_id;
    const size_t guess = buf.size() / 40 + 16;
    for (auto* v : {&ch, &st, &en, &r_id, &a_id, &s_id, &g_id, &n_id}) v->reserve(guess);
    const char* p = buf.data();
    const char* const stop = p + buf.size();
    int n_fields = -1;
    while (p < stop) {
        const char* nl = (const char*)memchr(p, '\n', (size_t)(stop - p));
        const char* le = nl ? nl : stop;
        const char* next = nl ? nl + 1 : stop;
        if (le > p && le[-1] == '\r') --le;
        if (le == p) {                                               // empty line: skipped, as the reader does
            p = next;
            continue;
        }
        std::string_view fld[8];
        int nf = 0;
        const char* q = p;
        for (;;) {
            const char* tab = (const char*)memchr(q, '\t', (size_t)(le - q));
            const char* fe = tab ? tab : le;
            if (nf < 8) fld[nf] = std::string_view(q, (size_t)(fe - q));
            ++nf;
            if (!tab) break;
            q = tab + 1;
        }
        if (n_fields < 0) n_fields = nf;
        if (nf != n_fields || nf < 8) return DIG_OK;                 // ragged or short rows: the Python path reports them
        int64_t s0, e0;
        if (!parse_i64(fld[1], s0) || !parse_i64(fld[2], e0)) return DIG_OK;
        ch.push_back(autosome(fld[0]));
        st.push_back(s0);
        en.push_back(e0);
        r_id.push_back(ref.get(fld[3]));
        a_id.push_back(alt.get(fld[4]));
        s_id.push_back(samp.get(fld[5]));
        g_id.push_back(gene.get(fld[6]));
        n_id.push_back(annot.get(fld[7]));
        p = next;
    }
    if (n_fields < 0) return DIG_OK;                                 // no row at all: the Python path says what it says about such a file
    // the reference picks the schema by the column count (mutation_tools.py:56-78): GENE and ANNOT sit in columns 6 and 7 of
    // files with 8, 10 or 11 columns only (a 9-column file has ANNOT in column 6 and no GENE): anything else is the Python path's
    if (n_fields != 8 && n_fields != 10 && n_fields != 11) return DIG_OK;
    // pandas.read_csv turns these spellings of "missing" into NaN (and the reference's groupby then drops such samples): a file
    // with one of them in a label column goes the Python path, which reads it with pandas
    static const char* const kNa[] = {"", "#N/A", "#N/A N/A", "#NA", "-1.#IND", "-1.#QNAN", "-NaN", "-nan", "1.#IND", "1.#QNAN", "<NA>",
                                      "N/A", "NA", "NULL", "NaN", "None", "n/a", "nan", "null"};
    for (const Interner* in : {&ref, &alt, &samp, &gene, &annot})
        for (const std::string_view lab : in->labels)
            for (const char* na : kNa)
                if (lab == std::string_view(na)) return DIG_OK;
    int64_t indel_id = -1;
    {
        auto it = annot.ids.find(std::string_view("INDEL"));
        if (it != annot.ids.end()) indel_id = it->second;
    }
    auto* m = new MutFile();
    const size_t n_all = ch.size();
    size_t n_keep = 0;
    for (size_t i = 0; i < n_all; ++i) n_keep += ch[i] > 0;
    for (auto* v : {&m->chrom, &m->start, &m->end, &m->uid, &m->sample, &m->indel, &m->gene}) v->resize(n_keep);
    std::vector<int64_t> remap(samp.labels.size(), -1), kr(n_keep), ka(n_keep);
    std::vector<std::string_view> order;
    size_t k = 0;
    for (size_t i = 0; i < n_all; ++i) {
        if (ch[i] <= 0) continue;
        m->chrom[k] = ch[i];
        m->start[k] = st[i];
        m->end[k] = en[i];
        int64_t& rm = remap[(size_t)s_id[i]];
        if (rm < 0) {
            rm = (int64_t)order.size();
            order.push_back(samp.labels[(size_t)s_id[i]]);
        }
        m->sample[k] = rm;
        m->indel[k] = n_id[i] == indel_id ? 1 : 0;
        m->gene[k] = g_id[i];
        kr[k] = r_id[i];
        ka[k] = a_id[i];
        ++k;
    }
    // uid: dense rank of (chrom, start, end, ref id, alt id) in sorted order
    std::vector<uint32_t> idx(n_keep);
    for (size_t i = 0; i < n_keep; ++i) idx[i] = (uint32_t)i;
    auto less = [&](uint32_t a, uint32_t b) {
        if (m->chrom[a] != m->chrom[b]) return m->chrom[a] < m->chrom[b];
        if (m->start[a] != m->start[b]) return m->start[a] < m->start[b];
        if (m->end[a] != m->end[b]) return m->end[a] < m->end[b];
        if (kr[a] != kr[b]) return kr[a] < kr[b];
        return ka[a] < ka[b];
    };
    if (n_keep >= ((size_t)1 << 32)) {
        delete m;
        return DIG_OK;
    }
    bool packs = true;                                               // coordinates and ids that fit two 64-bit keys: a plain sort
    for (size_t i = 0; i < n_keep && packs; ++i)
        packs = m->start[i] >= 0 && m->start[i] < ((int64_t)1 << 40) && m->end[i] >= m->start[i] &&
                m->end[i] - m->start[i] < ((int64_t)1 << 22) && kr[i] < (1 << 20) && ka[i] < (1 << 20);
    if (packs) {
        struct Key {
            uint64_t a, b;
            uint32_t i;
        };
        std::vector<Key> keys(n_keep);
        for (size_t i = 0; i < n_keep; ++i)
            keys[i] = Key{((uint64_t)m->chrom[i] << 40) | (uint64_t)m->start[i],
                          ((uint64_t)(m->end[i] - m->start[i]) << 40) | ((uint64_t)kr[i] << 20) | (uint64_t)ka[i], (uint32_t)i};
        std::sort(keys.begin(), keys.end(), [](const Key& x, const Key& y) { return x.a != y.a ? x.a < y.a : x.b < y.b; });
        int64_t rank = -1;
        for (size_t i = 0; i < n_keep; ++i) {
            if (i == 0 || keys[i].a != keys[i - 1].a || keys[i].b != keys[i - 1].b) ++rank;
            m->uid[keys[i].i] = rank;
        }
    } else {
        std::sort(idx.begin(), idx.end(), less);
        int64_t rank = -1;
        for (size_t i = 0; i < n_keep; ++i) {
            if (i == 0 || less(idx[i - 1], idx[i])) ++rank;
            m->uid[idx[i]] = rank;
        }
    }
    for (size_t i = 0; i < order.size(); ++i) {
        if (i) m->sample_names.push_back('\n');
        m->sample_names.append(order[i].data(), order[i].size());
    }
    m->n_samples = (int64_t)order.size();
    *handle = m;
    *n_rows = (int64_t)n_keep;
    *n_samples = m->n_samples;
    *names_bytes = (int64_t)m->sample_names.size();
    return DIG_OK;
}

int dig_mutation_file_fetch_host(void* handle, int64_t* chrom, int64_t* start, int64_t* end, int64_t* uid, int64_t* sample,
                                 int64_t* indel, int64_t* gene, char* sample_names)
{
    DIG_REQUIRE(handle, "a handle of dig_mutation_file_parse_host");
    const MutFile* m = static_cast<const MutFile*>(handle);
    const size_t nb = m->chrom.size() * sizeof(int64_t);
    if (nb) {
        DIG_REQUIRE(chrom && start && end && uid && sample && indel && gene, "non-null arrays of n_rows int64");
        memcpy(chrom, m->chrom.data(), nb);
        memcpy(start, m->start.data(), nb);
        memcpy(end, m->end.data(), nb);
        memcpy(uid, m->uid.data(), nb);
        memcpy(sample, m->sample.data(), nb);
        memcpy(indel, m->indel.data(), nb);
        memcpy(gene, m->gene.data(), nb);
    }
    if (!m->sample_names.empty()) {
        DIG_REQUIRE(sample_names, "a buffer of names_bytes bytes");
        memcpy(sample_names, m->sample_names.data(), m->sample_names.size());
    }
    return DIG_OK;
}

int dig_mutation_file_flags_host(void* handle, int64_t* first_row, int64_t* first_indel)
{
    // The two de-duplications the reference runs one after the other on a cohort's rows (read_mutation_file(drop_duplicates=True,
    // unique_indels=True), mutation_tools.py:106-117), as per-row flags in file order:
    //   first_row[i]   = 1: no earlier row has the same mutation (uid) AND sample       (drop_duplicate_mutations keeps it)
    //   first_indel[i] = 1: row i is kept, is an INDEL, and no earlier kept INDEL row has the same mutation AND gene label
    //                       (get_unique_indels keeps it)
    // A caller that counts kept rows (the genome-mode scale factors, transfer_tools.py:129-159) needs no sort on the device.
    DIG_REQUIRE(handle && first_row && first_indel, "a handle of dig_mutation_file_parse_host and two arrays of n_rows int64");
    const MutFile* m = static_cast<const MutFile*>(handle);
    const size_t n = m->chrom.size();
    struct Key {
        uint64_t k;
        uint32_t i;
    };
    std::vector<Key> keys(n);
    for (size_t i = 0; i < n; ++i) keys[i] = Key{((uint64_t)m->uid[i] << 32) | (uint64_t)(uint32_t)m->sample[i], (uint32_t)i};
    auto by_key_then_row = [](const Key& x, const Key& y) { return x.k != y.k ? x.k < y.k : x.i < y.i; };
    std::sort(keys.begin(), keys.end(), by_key_then_row);
    for (size_t j = 0; j < n; ++j) first_row[keys[j].i] = (j == 0 || keys[j].k != keys[j - 1].k) ? 1 : 0;
    size_t ni = 0;
    for (size_t i = 0; i < n; ++i) {
        first_indel[i] = 0;
        if (first_row[i] && m->indel[i]) keys[ni++] = Key{((uint64_t)m->uid[i] << 32) | (uint64_t)(uint32_t)m->gene[i], (uint32_t)i};
    }
    std::sort(keys.begin(), keys.begin() + (ptrdiff_t)ni, by_key_then_row);
    for (size_t j = 0; j < ni; ++j) first_indel[keys[j].i] = (j == 0 || keys[j].k != keys[j - 1].k) ? 1 : 0;
    return DIG_OK;
}

int dig_mutation_file_free_host(void* handle)
{
    delete static_cast<MutFile*>(handle);
    return DIG_OK;
}

}  // extern "C"
