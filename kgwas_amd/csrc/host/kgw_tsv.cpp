// kgw_tsv.cpp -- host-side (no GPU) writer of the result tables KGWAS.train() ends with (kgwas/kgwas.py:205-212:
// lr_uni_to_save.to_csv(path, index=False, sep='\t')).  pandas spends 3-7 s on the ~0.54 M-row table (generic per-cell
// formatting): a quarter of a 10-epoch run on MI355X.  This writer produces the SAME BYTES as pandas' default formatting --
// floats as Python's repr (shortest round-trip digits; exponent form iff the decimal point falls at or before 1e-5 or beyond
// 1e16, two-digit exponent; ".0" on integral values), float32 columns with float32's own shortest digits, NaN as the empty
// field, integers and booleans as Python prints them, strings raw -- from the columns' own buffers, one pass, one write per
// 4 MB.  The caller (kgwas_amd/utils.py::write_tsv) falls back to pandas for anything else (a string that would need quoting,
// another dtype): the output never differs, only the time.  Plain C ABI, no allocation the caller sees, status return.
#include <charconv>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

namespace {

enum { COL_F64 = 0, COL_F32 = 1, COL_I64 = 2, COL_BOOL = 3, COL_STR = 4 };

// Python's float_repr_style 'short' applied to the shortest digits std::to_chars gives
template <typename F>
inline char* fmt_float(char* p, F x) {
    if (x != x) return p;                                   // NaN: empty field (pandas na_rep='')
    if (x == (F)(1.0 / 0.0)) { memcpy(p, "inf", 3); return p + 3; }
    if (x == -(F)(1.0 / 0.0)) { memcpy(p, "-inf", 4); return p + 4; }
    char b[48];
    auto r = std::to_chars(b, b + sizeof(b), x, std::chars_format::scientific);       // [-]d[.ddd]e[+-]XX
    char* s = b;
    if (*s == '-') { *p++ = '-'; ++s; }
    char dig[32];
    int nd = 0;
    char* e = s;
    while (e < r.ptr && *e != 'e') { if (*e != '.') dig[nd++] = *e; ++e; }
    int ex = 0;
    { const char* q = e + 1; bool neg = (*q == '-'); if (*q == '+' || *q == '-') ++q; while (q < r.ptr) ex = ex * 10 + (*q++ - '0'); if (neg) ex = -ex; }
    const int decpt = ex + 1;                              // value = 0.DIGITS x 10^decpt
    // (float32: numpy decides by the VALUE -- float32(1e-4) = 9.9999997e-05 < 1e-4 prints "1e-04" although its shortest digits
    //  "1" sit at the 1e-4 position; for float64 the two rules agree)
    const bool small32 = sizeof(F) == 4 && decpt == -3 && (double)(x < 0 ? -x : x) < 1e-4;
    if (decpt <= -4 || decpt > 16 || small32) {                        // exponent form: d[.ddd]e[+-]XX, at least two exponent digits
        *p++ = dig[0];
        if (nd > 1) { *p++ = '.'; memcpy(p, dig + 1, nd - 1); p += nd - 1; }
        *p++ = 'e';
        int x10 = decpt - 1;
        *p++ = x10 < 0 ? '-' : '+';
        if (x10 < 0) x10 = -x10;
        char t[8]; int nt = 0;
        do { t[nt++] = (char)('0' + x10 % 10); x10 /= 10; } while (x10);
        if (nt < 2) t[nt++] = '0';
        while (nt) *p++ = t[--nt];
        return p;
    }
    if (decpt <= 0) {                                       // 0.000ddd
        *p++ = '0'; *p++ = '.';
        for (int i = 0; i < -decpt; ++i) *p++ = '0';
        memcpy(p, dig, nd); return p + nd;
    }
    if (decpt >= nd) {                                      // ddd000.0
        memcpy(p, dig, nd); p += nd;
        for (int i = nd; i < decpt; ++i) *p++ = '0';
        *p++ = '.'; *p++ = '0';
        return p;
    }
    memcpy(p, dig, decpt); p += decpt;                      // dd.ddd
    *p++ = '.';
    memcpy(p, dig + decpt, nd - decpt);
    return p + (nd - decpt);
}

inline char* fmt_i64(char* p, int64_t v) {
    auto r = std::to_chars(p, p + 24, v);
    return r.ptr;
}

}  // namespace

// Writes `header` (already formatted, without the newline) and n_rows rows of n_cols fields separated by `sep`.
// col_kind[c]: 0 float64, 1 float32, 2 int64, 3 bool (uint8), 4 string; col_ptr[c]: the column's contiguous buffer -- for strings
// the concatenated UTF-8 bytes, with str_off[c] pointing at n_rows + 1 int64 offsets into them.  Returns 0, or -1 (arguments),
// -2 (a string field that pandas would quote: the caller falls back), -3 (I/O).
extern "C" int kgw_write_tsv(const char* path, const char* header, int64_t n_rows, int32_t n_cols, const int32_t* col_kind,
                             const void* const* col_ptr, const int64_t* const* str_off, char sep) {
    if (!path || !header || n_rows < 0 || n_cols < 1 || !col_kind || !col_ptr) return -1;
    for (int c = 0; c < n_cols; ++c) {
        if (col_kind[c] < 0 || col_kind[c] > COL_STR || !col_ptr[c]) return -1;
        if (col_kind[c] == COL_STR) {
            if (!str_off || !str_off[c]) return -1;
            const char* s = (const char*)col_ptr[c];
            const int64_t n = str_off[c][n_rows];
            for (int64_t i = 0; i < n; ++i) {
                const char ch = s[i];
                if (ch == sep || ch == '"' || ch == '\n' || ch == '\r') return -2;
            }
            for (int64_t i = 0; i < n_rows; ++i)             // (pandas writes an EMPTY string field as "": quoted)
                if (str_off[c][i + 1] == str_off[c][i]) return -2;
        }
    }
    FILE* f = fopen(path, "wb");
    if (!f) return -3;
    std::vector<char> buf((size_t)4 << 20);
    char* p = buf.data();
    char* const hi = buf.data() + buf.size() - 4096;
    const size_t hl = strlen(header);
    if (fwrite(header, 1, hl, f) != hl || fputc('\n', f) == EOF) { fclose(f); return -3; }
    for (int64_t i = 0; i < n_rows; ++i) {
        for (int c = 0; c < n_cols; ++c) {
            if (c) *p++ = sep;
            switch (col_kind[c]) {
                case COL_F64: p = fmt_float<double>(p, ((const double*)col_ptr[c])[i]); break;
                case COL_F32: p = fmt_float<float>(p, ((const float*)col_ptr[c])[i]); break;
                case COL_I64: p = fmt_i64(p, ((const int64_t*)col_ptr[c])[i]); break;
                case COL_BOOL:
                    if (((const uint8_t*)col_ptr[c])[i]) { memcpy(p, "True", 4); p += 4; } else { memcpy(p, "False", 5); p += 5; }
                    break;
                default: {
                    const int64_t a = str_off[c][i], b = str_off[c][i + 1];
                    if (b - a > 2048) {                       // (a field longer than the slack: flush around it)
                        if (fwrite(buf.data(), 1, p - buf.data(), f) != (size_t)(p - buf.data())) { fclose(f); return -3; }
                        p = buf.data();
                        if (fwrite((const char*)col_ptr[c] + a, 1, b - a, f) != (size_t)(b - a)) { fclose(f); return -3; }
                    } else {
                        memcpy(p, (const char*)col_ptr[c] + a, b - a); p += b - a;
                    }
                }
            }
            if (p > hi) {
                if (fwrite(buf.data(), 1, p - buf.data(), f) != (size_t)(p - buf.data())) { fclose(f); return -3; }
                p = buf.data();
            }
        }
        *p++ = '\n';
    }
    if (fwrite(buf.data(), 1, p - buf.data(), f) != (size_t)(p - buf.data())) { fclose(f); return -3; }
    return fclose(f) == 0 ? 0 : -3;
}
