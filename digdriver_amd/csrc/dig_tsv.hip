// dig_tsv.hip -- the result-file writer of the drop-in (host code only; no kernel).
//
// DigDriver.py writes a result frame with `df.to_csv(path, header=True, index=True, sep="\t")` (DigDriver.py:115-118,
// DigPretrain.py likewise): pandas formats every float with Python's repr -- the shortest decimal string that reads back
// to the same double, positional for 1e-4 <= |x| < 1e16, d.ddde-XX otherwise, "12.0" for whole numbers -- one Python call
// per cell: 2.0 s per 120 091-row frame of 24 columns, 75 s for the 37 cohorts of BASELINE configs[2] (tools/e2e_bench.py:
// 20x everything else in the pipeline together).  dig_write_tsv_host produces the same bytes from the column arrays:
// std::to_chars gives the shortest round-trip digits, the layout rules of float_repr_style 'short' are applied here; rows
// are formatted in chunks by a few threads and written in order.
#include <charconv>
#include <cmath>
#include <cstring>
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
    FILE* f = std::fopen(path, "wb");
    if (!f) return ::dig::set_error(DIG_EINVAL, "dig_write_tsv_host: cannot open %s for writing", path);
    std::fputs(header, f);
    std::fputc('\n', f);
    const int64_t chunk = 8192;
    const int64_t n_chunks = (n_rows + chunk - 1) / chunk;
    if (n_threads < 1) n_threads = 1;
    if (n_threads > 16) n_threads = 16;
    if ((int64_t)n_threads > n_chunks) n_threads = (int)(n_chunks > 0 ? n_chunks : 1);
    std::vector<std::string> text((size_t)n_chunks);
    auto work = [&](int t) {
        for (int64_t c = t; c < n_chunks; c += n_threads) {
            const int64_t r0 = c * chunk, r1 = r0 + chunk < n_rows ? r0 + chunk : n_rows;
            std::string& out = text[(size_t)c];
            out.resize((size_t)((label_off[r1] - label_off[r0]) + (r1 - r0) * (1 + (int64_t)n_cols * 26)));
            char* p = &out[0];
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
            out.resize((size_t)(p - &out[0]));
        }
    };
    std::vector<std::thread> pool;
    for (int t = 1; t < n_threads; ++t) pool.emplace_back(work, t);
    work(0);
    for (auto& th : pool) th.join();
    bool ok = true;
    for (const auto& s : text) ok = ok && std::fwrite(s.data(), 1, s.size(), f) == s.size();
    ok = (std::fclose(f) == 0) && ok;
    if (!ok) return ::dig::set_error(DIG_EINVAL, "dig_write_tsv_host: short write to %s", path);
    return DIG_OK;
}

}  // extern "C"
