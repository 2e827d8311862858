// Newline-delimited JSON -> Arrow columns on gfx950 (SURVEY.md section 8(f) rank 3, and the decode step of row a12):
// the reference keeps every epoch's events as serde_json lines (flock/src/datasource/nexmark/generator.rs:79-93) and turns
// them into RecordBatches with arrow's json::Reader at the granule (`event_bytes_to_batch`, flock/src/transmute.rs:255-266,
// called from nexmark.rs:180-205) -- by the survey's own account the reference's real bottleneck.  HBM-bound byte work,
// no MFMA.
//
//   lines   : count -> scan -> emit over 16 KiB tiles of the text: every raw 0x0A ends a line (JSON strings cannot hold one);
//             the count pass keeps a 64-bit newline mask per 64 bytes, which is all the emit pass reads
//   parse   : 256 lines per workgroup; their bytes are one contiguous range, staged in LDS with 16-byte loads; one lane
//             walks one line -- first against the exact shape serde_json writes (`"name":value` in schema order, no white
//             space, no escapes: straight-line checks), and only if that fails through the general parser: keys matched against the schema's field names, integers parsed exactly (sign, overflow),
//             strings delimited with escape awareness, unknown keys skipped with a depth-counting value skipper.
//             Int32 / Int64 (Timestamp) columns are written directly (row = line: coalesced); for a Utf8 field the lane
//             records where the value's bytes lie in the INPUT (start, end) and whether it holds escapes
//   strings : a field without escapes is exactly a `take` of byte ranges of the input, so it reuses the Utf8 gather
//             (lengths -> tile scan -> LDS-staged emit, gather.hip) with (start, end) pairs as the "offsets"; a field
//             with escapes goes through an unescaping copy (one lane per value)
// Numbers with a fraction or an exponent, blank lines, keys with escapes and rows that lack a schema field are reported
// as errors with their line number (the generator's lines never have them): FLOCKGPU_ERR_UNSUPPORTED / _INVALID.
#include <algorithm>
#include <cstring>

#include "gather.hpp"

using namespace flockgpu;

namespace {

constexpr int kMaxFields = 16;
constexpr int kMaxName = 32;
constexpr int kNlBytes = 64;                    // bytes per lane in the newline passes
constexpr int kNlTile = kBlock * kNlBytes;      // 16 KiB
constexpr int kParseLines = kBlock;             // lines per workgroup
constexpr int kStageBytes = 48 * 1024;          // most LDS a workgroup stages its lines in

enum : int32_t { kInt32 = 0, kInt64 = 1, kUtf8 = 2 };
enum : uint32_t { kErrSyntax = 1, kErrNumber = 2, kErrMissing = 3, kErrBlank = 4, kErrKeyEscape = 5, kErrRange = 6 };

struct JsonSpec {
    int32_t n;
    int32_t type[kMaxFields];
    int32_t name_len[kMaxFields];
    char name[kMaxFields][kMaxName];
    // the bytes that precede field f's value in a line as serde_json writes it: `{"name":` for the first field, `,"name":`
    // after it, as little-endian words (parse_line_words compares four bytes at a time)
    int32_t pre_len[kMaxFields];
    uint32_t pre[kMaxFields][(kMaxName + 3 + 7) / 8 * 2];
};

struct JsonOut {
    void *values[kMaxFields];    // int32 / int64 columns
    int32_t *pairs[kMaxFields];  // Utf8: (start, end) of the raw value in the input, 2 per row
    int32_t *ulen[kMaxFields];   // Utf8: length after unescaping
};

// ---- line index --------------------------------------------------------------------------------------------------
__device__ __forceinline__ uint64_t newline_mask(const uint8_t *__restrict__ bytes, int64_t n, int64_t p0) {
    uint64_t m = 0;
    if (p0 + kNlBytes <= n) {
#pragma unroll
        for (int q = 0; q < kNlBytes / 16; ++q) {
            const uint4 v = *reinterpret_cast<const uint4 *>(bytes + p0 + q * 16);
            const uint32_t w[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int b = 0; b < 4; ++b) m |= (uint64_t)(((w[i] >> (8 * b)) & 0xFFu) == 0x0Au) << (q * 16 + i * 4 + b);
        }
    } else {
        for (int i = 0; i < kNlBytes; ++i)
            if (p0 + i < n && bytes[p0 + i] == 0x0A) m |= uint64_t(1) << i;
    }
    return m;
}

// The lane's 64-bit newline mask is kept (1/8 of the text): the emit pass reads the masks, not the text again.
__global__ __launch_bounds__(kBlock) void json_newline_count_kernel(const uint8_t *__restrict__ bytes, int64_t n,
                                                                   uint64_t *__restrict__ masks, uint32_t *__restrict__ counts) {
    const int64_t p0 = (int64_t)blockIdx.x * kNlTile + (int64_t)threadIdx.x * kNlBytes;
    const uint64_t m = newline_mask(bytes, n, p0);
    masks[(size_t)blockIdx.x * kBlock + threadIdx.x] = m;
    const uint32_t c = (uint32_t)__popcll((unsigned long long)m);
    const uint32_t incl = wave_incl_scan_u32(c);
    if (lane_id() == 63) counts[(size_t)blockIdx.x * kWavesPerBlock + (threadIdx.x >> 6)] = incl;
}

// line_start[k + 1] = position after the k-th newline
__global__ __launch_bounds__(kBlock) void json_newline_emit_kernel(const uint64_t *__restrict__ masks,
                                                                  const uint32_t *__restrict__ counts,
                                                                  const uint64_t *__restrict__ tile_base,
                                                                  int32_t *__restrict__ line_start) {
    const int64_t p0 = (int64_t)blockIdx.x * kNlTile + (int64_t)threadIdx.x * kNlBytes;
    uint64_t m = masks[(size_t)blockIdx.x * kBlock + threadIdx.x];
    const uint32_t c = (uint32_t)__popcll((unsigned long long)m);
    const uint4 wc = *reinterpret_cast<const uint4 *>(counts + (size_t)blockIdx.x * kWavesPerBlock);
    const int wave = threadIdx.x >> 6;
    uint64_t k = tile_base[blockIdx.x] + (wave > 0 ? wc.x : 0u) + (wave > 1 ? wc.y : 0u) + (wave > 2 ? wc.z : 0u) +
                 wave_incl_scan_u32(c) - c;
    for (; m; m &= m - 1) line_start[++k] = (int32_t)(p0 + (__ffsll((unsigned long long)m) - 1) + 1);
    if (blockIdx.x == 0 && threadIdx.x == 0) line_start[0] = 0;
}

// ---- one line ------------------------------------------------------------------------------------------------------
template <bool kLds>
struct Text {
    const uint8_t *g;       // the input
    const uint8_t *s;       // LDS copy of [s_base, ...)
    int32_t s_base;
    __device__ __forceinline__ uint32_t at(int32_t p) const { return kLds ? s[p - s_base] : g[p]; }
};

__device__ __forceinline__ bool is_ws(uint32_t c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
__device__ __forceinline__ int hex_of(uint32_t c) {
    return c >= '0' && c <= '9' ? (int)c - '0' : c >= 'a' && c <= 'f' ? (int)c - 'a' + 10 : c >= 'A' && c <= 'F' ? (int)c - 'A' + 10 : -1;
}

// p at the opening quote.  Returns the position after the closing quote, or -1; [*b, *e) = raw bytes between the quotes,
// *ulen = bytes after unescaping, *esc = the value holds a backslash.
template <bool kLds>
__device__ int32_t scan_string(const Text<kLds> &t, int32_t p, int32_t end, int32_t *b, int32_t *e, int32_t *ulen, bool *esc) {
    ++p;
    *b = p;
    int32_t len = 0;
    bool any = false;
    while (p < end) {
        const uint32_t c = t.at(p);
        if (c == '"') {
            *e = p;
            *ulen = len;
            *esc = any;
            return p + 1;
        }
        if (c < 0x20) return -1;  // raw control characters are not allowed inside a JSON string
        if (c == '\\') {
            any = true;
            if (p + 1 >= end) return -1;
            const uint32_t d = t.at(p + 1);
            if (d == 'u') {
                if (p + 6 > end) return -1;
                int cp = 0;
                for (int i = 0; i < 4; ++i) {
                    const int h = hex_of(t.at(p + 2 + i));
                    if (h < 0) return -1;
                    cp = cp * 16 + h;
                }
                p += 6;
                if (cp >= 0xD800 && cp < 0xDC00) {  // high surrogate: must be followed by \uDC00..DFFF
                    if (p + 6 > end || t.at(p) != '\\' || t.at(p + 1) != 'u') return -1;
                    int lo = 0;
                    for (int i = 0; i < 4; ++i) {
                        const int h = hex_of(t.at(p + 2 + i));
                        if (h < 0) return -1;
                        lo = lo * 16 + h;
                    }
                    if (lo < 0xDC00 || lo > 0xDFFF) return -1;
                    p += 6;
                    len += 4;
                } else if (cp >= 0xDC00 && cp <= 0xDFFF) {
                    return -1;
                } else {
                    len += cp < 0x80 ? 1 : cp < 0x800 ? 2 : 3;
                }
                continue;
            }
            if (!(d == '"' || d == '\\' || d == '/' || d == 'b' || d == 'f' || d == 'n' || d == 'r' || d == 't')) return -1;
            p += 2;
            ++len;
            continue;
        }
        ++p;
        ++len;
    }
    return -1;
}

// Skips any JSON value starting at p (after white space); returns the position after it or -1.
template <bool kLds>
__device__ int32_t skip_value(const Text<kLds> &t, int32_t p, int32_t end) {
    if (p >= end) return -1;
    uint32_t c = t.at(p);
    if (c == '"') {
        int32_t b, e, u;
        bool esc;
        return scan_string(t, p, end, &b, &e, &u, &esc);
    }
    if (c == '{' || c == '[') {
        int depth = 0;
        while (p < end) {
            c = t.at(p);
            if (c == '"') {
                int32_t b, e, u;
                bool esc;
                p = scan_string(t, p, end, &b, &e, &u, &esc);
                if (p < 0) return -1;
                continue;
            }
            if (c == '{' || c == '[') ++depth;
            if (c == '}' || c == ']') {
                if (--depth == 0) return p + 1;
            }
            ++p;
        }
        return -1;
    }
    // number / true / false / null: up to the next delimiter
    const int32_t p0 = p;
    while (p < end) {
        c = t.at(p);
        if (c == ',' || c == '}' || c == ']' || is_ws(c)) break;
        ++p;
    }
    return p > p0 ? p : -1;
}

template <bool kLds>
__device__ uint32_t parse_line(const Text<kLds> &t, int32_t p, int32_t end, const JsonSpec &spec, int64_t row, const JsonOut &out) {
    while (end > p && is_ws(t.at(end - 1))) --end;
    while (p < end && is_ws(t.at(p))) ++p;
    if (p >= end) return kErrBlank;
    if (t.at(p) != '{') return kErrSyntax;
    ++p;
    uint32_t seen = 0;
    int next_f = 0;
    while (p < end && is_ws(t.at(p))) ++p;
    const bool empty_object = p < end && t.at(p) == '}';
    if (empty_object) ++p;
    while (!empty_object) {  // one member per trip; leaves through the closing brace
        while (p < end && is_ws(t.at(p))) ++p;
        if (p >= end) return kErrSyntax;
        if (t.at(p) != '"') return kErrSyntax;
        int32_t kb, ke, ku;
        bool kesc;
        p = scan_string(t, p, end, &kb, &ke, &ku, &kesc);
        if (p < 0) return kErrSyntax;
        if (kesc) return kErrKeyEscape;
        int f = -1;
        for (int k = 0; k < spec.n; ++k) {  // members usually arrive in schema order: start at the field after the last match
            int i = next_f + k;
            i = i >= spec.n ? i - spec.n : i;
            if (spec.name_len[i] != ke - kb) continue;
            bool same = true;
            for (int j = 0; j < ke - kb; ++j) same = same && t.at(kb + j) == (uint32_t)(uint8_t)spec.name[i][j];
            if (same) {
                f = i;
                next_f = i + 1 == spec.n ? 0 : i + 1;
                break;
            }
        }
        while (p < end && is_ws(t.at(p))) ++p;
        if (p >= end || t.at(p) != ':') return kErrSyntax;
        ++p;
        while (p < end && is_ws(t.at(p))) ++p;
        if (p >= end) return kErrSyntax;
        if (f < 0) {
            p = skip_value(t, p, end);
            if (p < 0) return kErrSyntax;
        } else if (spec.type[f] == kUtf8) {
            if (t.at(p) != '"') return kErrSyntax;
            int32_t b, e, u;
            bool esc;
            p = scan_string(t, p, end, &b, &e, &u, &esc);
            if (p < 0) return kErrSyntax;
            out.pairs[f][2 * row] = b;
            out.pairs[f][2 * row + 1] = e;
            out.ulen[f][row] = esc ? -u - 1 : u;  // negative: needs unescaping (length = -(v + 1))
            seen |= 1u << f;
        } else {
            bool neg = false;
            if (t.at(p) == '-') {
                neg = true;
                ++p;
            }
            if (p >= end || t.at(p) < '0' || t.at(p) > '9') return kErrSyntax;  // not a number at all (a string, true, null ...)
            if (t.at(p) == '0' && p + 1 < end && t.at(p + 1) >= '0' && t.at(p + 1) <= '9') return kErrSyntax;  // JSON has no leading zeros
            uint64_t v = 0;
            int digits = 0;
            while (p < end) {
                const uint32_t c = t.at(p);
                if (c < '0' || c > '9') break;
                if (++digits > 19) return kErrRange;
                v = v * 10 + (c - '0');
                ++p;
            }
            if (p < end) {
                const uint32_t c = t.at(p);
                if (!(c == ',' || c == '}' || is_ws(c))) return kErrNumber;  // fraction / exponent: not an integer literal
            }
            if (v > (uint64_t)0x7fffffffffffffffull + (neg ? 1u : 0u)) return kErrRange;
            const int64_t sv = neg ? (int64_t)(0 - v) : (int64_t)v;
            if (spec.type[f] == kInt32) {
                if (sv < -2147483648ll || sv > 2147483647ll) return kErrRange;
                reinterpret_cast<int32_t *>(out.values[f])[row] = (int32_t)sv;
            } else {
                reinterpret_cast<int64_t *>(out.values[f])[row] = sv;
            }
            seen |= 1u << f;
        }
        while (p < end && is_ws(t.at(p))) ++p;
        if (p >= end) return kErrSyntax;
        if (t.at(p) == ',') {  // a member must follow: the next trip insists on a key
            ++p;
            continue;
        }
        if (t.at(p) == '}') {
            ++p;
            break;
        }
        return kErrSyntax;
    }
    if (p != end) return kErrSyntax;  // trailing bytes after the object
    if (seen != (spec.n >= 32 ? ~0u : (1u << spec.n) - 1u)) return kErrMissing;
    return 0;
}

// The shape serde_json writes (generator.rs:81-91): {"name":value,...} with the members in schema order, no white space,
// and -- for the generator's strings -- no escapes.  Straight-line checks against the literal `"name":` of every field;
// anything else (other order, white space, escapes, unknown members, long or out-of-range numbers) returns false and
// the line goes through parse_line, which also produces the error codes.  Columns written here before a `false` are
// simply rewritten by it.
template <bool kLds>
__device__ bool parse_line_fast(const Text<kLds> &t, int32_t p, int32_t end, const JsonSpec &spec, int64_t row, const JsonOut &out) {
    if (p >= end || t.at(p) != '{') return false;
    ++p;
    for (int f = 0; f < spec.n; ++f) {
        const int32_t nl = spec.name_len[f];
        if (p + nl + 4 > end || t.at(p) != '"' || t.at(p + nl + 1) != '"' || t.at(p + nl + 2) != ':') return false;
        bool same = true;
        for (int j = 0; j < nl; ++j) same = same && t.at(p + 1 + j) == (uint32_t)(uint8_t)spec.name[f][j];
        if (!same) return false;
        p += nl + 3;
        if (spec.type[f] == kUtf8) {
            if (t.at(p) != '"') return false;
            const int32_t b = ++p;
            uint32_t c = 0;
            while (p < end && (c = t.at(p)) != '"' && c != '\\' && c >= 0x20) ++p;
            if (p >= end || c != '"') return false;
            out.pairs[f][2 * row] = b;
            out.pairs[f][2 * row + 1] = p;
            out.ulen[f][row] = p - b;
            ++p;
        } else {
            const bool neg = t.at(p) == '-';
            p += neg;
            uint64_t v = 0;
            int digits = 0;
            uint32_t c;
            while (p < end && (c = t.at(p) - '0') <= 9u) {
                v = v * 10 + c;
                ++digits;
                ++p;
            }
            if (digits == 0 || digits > 18 || (digits > 1 && t.at(p - digits) == '0')) return false;
            const int64_t sv = neg ? -(int64_t)v : (int64_t)v;
            if (spec.type[f] == kInt32) {
                if (sv < -2147483648ll || sv > 2147483647ll) return false;
                reinterpret_cast<int32_t *>(out.values[f])[row] = (int32_t)sv;
            } else {
                reinterpret_cast<int64_t *>(out.values[f])[row] = sv;
            }
        }
        if (p >= end || t.at(p) != (uint32_t)(f + 1 == spec.n ? '}' : ',')) return false;
        ++p;
    }
    return p == end;
}

// ---- the same shape, four and eight bytes at a time (text staged in LDS) ------------------------------------------------
// One lane still walks one line, but through unaligned 4- and 8-byte LDS reads and SWAR arithmetic instead of a byte per
// trip: ~16 instructions per byte of text made the byte-wise walk instruction-bound at a third of the HBM rate.
// The staged range carries 16 bytes of slack, so a read that starts inside a line may run past its end; whatever lies
// there is cut off by `end` before it is used.
// Four / eight bytes of the staged text from ANY byte offset, out of dword-aligned reads and a funnel shift: an LDS read
// that is not dword-aligned is ~6x as expensive on gfx950 (64 lanes at a 74-byte stride: ~68 cycles of the CU's LDS pipe per
// wave read at a byte offset whatever the width, 10-12 for an aligned b32 / a 4-byte-aligned b64, ~17 for the three dwords
// read here -- tools/micro/lds_unaligned.hip).  `s` is 16-byte aligned.
__device__ __forceinline__ uint32_t lds_u32(const uint8_t *s, int32_t o) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(s + (o & ~3));
    return __funnelshift_r(w[0], w[1], (uint32_t)(o & 3) * 8);
}
__device__ __forceinline__ uint64_t lds_u64(const uint8_t *s, int32_t o) {
    const uint32_t *w = reinterpret_cast<const uint32_t *>(s + (o & ~3));
    const uint32_t sh = (uint32_t)(o & 3) * 8, w0 = w[0], w1 = w[1], w2 = w[2];
    return (uint64_t)__funnelshift_r(w0, w1, sh) | ((uint64_t)__funnelshift_r(w1, w2, sh) << 32);
}
// Bytes of x that are NOT an ASCII digit, as 0x80 flags (exact for every byte: no carry crosses a byte).
__device__ __forceinline__ uint64_t non_digit_flags(uint64_t x) {
    const uint64_t t = x ^ 0x3030303030303030ull;
    return (((t & 0x7F7F7F7F7F7F7F7Full) + 0x7676767676767676ull) | t) & 0x8080808080808080ull;
}
// Value of four digits, most significant in the lowest byte (w = text ^ '0000').
__device__ __forceinline__ uint32_t four_digits(uint32_t w) { return ((((w * 2561u) >> 8) & 0x00FF00FFu) * 6553601u) >> 16; }
// Value of the first nd (1..8) digits of the eight text bytes x.
__device__ __forceinline__ uint32_t leading_digits(uint64_t x, int nd) {
    const uint64_t t = (x ^ 0x3030303030303030ull) << (8 * (8 - nd));  // right-aligned: the vacated low bytes read as 0 digits
    return four_digits((uint32_t)t) * 10000u + four_digits((uint32_t)(t >> 32));
}

// (The LDS pipe is what this walk is bound by -- SQ_LDS_IDX_ACTIVE at 82 % of the kernel's cycles with the first version's
// ~25 byte-offset reads per bid line: hence few reads, ~14 per line, and every one of them dword-aligned.)
// kUnroll >= spec.n: the trip over the fields is unrolled at compile time, so that everything read from `spec` (the kernel's
// argument block) sits at a constant offset and arrives in a few wide scalar loads up front -- indexed by a loop counter it
// was ~8 dependent scalar loads per field, each waited for with the LDS counter.
template <int kUnroll>
__device__ __forceinline__ bool parse_line_words(const uint8_t *s, int32_t s_base, int32_t p, int32_t end, const JsonSpec &spec,
                                                 int64_t row, const JsonOut &out) {
    p -= s_base;
    end -= s_base;
#pragma unroll
    for (int f = 0; f < kUnroll; ++f) {
        if (f >= spec.n) break;
        const int32_t pl = spec.pre_len[f];
        if (p + pl + 1 > end) return false;  // the prefix and at least one byte of value
        uint64_t diff = 0;
#pragma unroll
        for (int j = 0; j < (kMaxName + 3 + 7) / 8; ++j) {
            if (8 * j >= pl) break;
            const int32_t left = pl - 8 * j;
            const uint64_t want = (uint64_t)spec.pre[f][2 * j] | ((uint64_t)spec.pre[f][2 * j + 1] << 32);
            if (left > 4) diff |= (lds_u64(s, p + 8 * j) ^ want) & (left >= 8 ? ~0ull : (1ull << (8 * left)) - 1ull);
            else diff |= ((uint64_t)lds_u32(s, p + 8 * j) ^ want) & ((1ull << (8 * left)) - 1ull);
        }
        if (diff) return false;
        p += pl;
        if (spec.type[f] == kUtf8) {
            const int32_t b = p + 1;
            uint64_t x = lds_u64(s, p);
            if ((x & 0xFFu) != '"') return false;
            x |= 0xFFu;  // the opening quote is not the closing one
            uint32_t c = 0;
            for (;;) {  // the first quote, backslash or control character (the lowest flag of each test is exact)
                const uint64_t q = x ^ 0x2222222222222222ull, bs = x ^ 0x5C5C5C5C5C5C5C5Cull;
                const uint64_t m = (((q - 0x0101010101010101ull) & ~q) | ((bs - 0x0101010101010101ull) & ~bs) |
                                    ((x - 0x2020202020202020ull) & ~x)) & 0x8080808080808080ull;
                if (m) {
                    const int at = (__ffsll((unsigned long long)m) - 1) >> 3;
                    p += at;
                    c = (uint32_t)(x >> (8 * at)) & 0xFFu;
                    break;
                }
                p += 8;
                if (p >= end) return false;
                x = lds_u64(s, p);
            }
            if (p >= end || c != '"') return false;
            out.pairs[f][2 * row] = b + s_base;
            out.pairs[f][2 * row + 1] = p + s_base;
            out.ulen[f][row] = p - b;
            ++p;
        } else {
            uint64_t x0 = lds_u64(s, p);
            const bool neg = (x0 & 0xFFu) == '-';
            if (neg) x0 = lds_u64(s, ++p);
            // how many digits (up to 18), eight at a time; then the leading 1..8 digits and whole groups of eight, the
            // groups cut out of the words already read
            uint64_t x1 = 0, x2 = 0;
            const uint64_t m0 = non_digit_flags(x0);
            int total = m0 ? (__ffsll((unsigned long long)m0) - 1) >> 3 : 8;
            if (total == 8 && p + 8 < end) {  // (never a read beyond the slack)
                x1 = lds_u64(s, p + 8);
                const uint64_t m1 = non_digit_flags(x1);
                total += m1 ? (__ffsll((unsigned long long)m1) - 1) >> 3 : 8;
                if (total == 16 && p + 16 < end) {
                    x2 = lds_u64(s, p + 16);
                    const uint64_t m2 = non_digit_flags(x2);
                    total += m2 ? (__ffsll((unsigned long long)m2) - 1) >> 3 : 8;
                }
            }
            total = min(total, end - p);
            if (total <= 0 || total > 18 || (total > 1 && (x0 & 0xFFu) == '0')) return false;  // (leading zero: the general parser's error)
            const int lead = total - ((total - 1) & ~7);
            uint64_t v = leading_digits(x0, lead);
            if (total > 8) {
                const int sh = 8 * lead;  // 8 .. 64
                v = v * 100000000ull + leading_digits(lead == 8 ? x1 : (x0 >> sh) | (x1 << (64 - sh)), 8);
                if (total > 16) v = v * 100000000ull + leading_digits(lead == 8 ? x2 : (x1 >> sh) | (x2 << (64 - sh)), 8);
            }
            p += total;
            const int64_t sv = neg ? -(int64_t)v : (int64_t)v;
            if (spec.type[f] == kInt32) {
                if (sv < -2147483648ll || sv > 2147483647ll) return false;
                reinterpret_cast<int32_t *>(out.values[f])[row] = (int32_t)sv;
            } else {
                reinterpret_cast<int64_t *>(out.values[f])[row] = sv;
            }
        }
    }
    return p + 1 == end && s[p] == '}';
}

// ---- any flat object, four and eight bytes at a time -------------------------------------------------------------------
// What other writers produce (Python's json.dumps puts a space after ':' and ','; members come in any order): white space where JSON
// allows it, the schema's members in ANY order, nothing else -- no unknown members, no escapes, no repeated keys (those lines go on to
// parse_line, which also words the errors).  Same machinery as parse_line_words: dword-aligned LDS reads, SWAR searches, integers eight
// digits per step; the key of a member is compared against every schema name as 64-bit words (the names come out of the kernel
// arguments, wave-uniform).  The byte-wise general parser spends ~60 instructions per character on such lines.
__device__ __forceinline__ uint64_t bytes_equal_flags(uint64_t x, uint32_t c) {   // 0x80 in every byte of x that equals c (exact)
    const uint64_t v = x ^ (0x0101010101010101ull * c);
    return ~(((v & 0x7F7F7F7F7F7F7F7Full) + 0x7F7F7F7F7F7F7F7Full) | v) & 0x8080808080808080ull;
}
// position of the first byte at or after p that is not JSON white space (end if there is none)
__device__ __forceinline__ int32_t skip_ws_words(const uint8_t *s, int32_t p, int32_t end) {
    while (p < end) {
        const uint64_t x = lds_u64(s, p);
        const uint64_t ws = bytes_equal_flags(x, ' ') | bytes_equal_flags(x, '\t') | bytes_equal_flags(x, '\r') | bytes_equal_flags(x, '\n');
        const uint64_t other = ~ws & 0x8080808080808080ull;
        if (other) return min(end, p + ((__ffsll((unsigned long long)other) - 1) >> 3));
        p += 8;
    }
    return end;
}

// The first byte at or after p that is not white space (`at`, `c`; at = end, c = 0 when there is none) and the first such byte behind it
// (`after`) -- both out of ONE 8-byte read in the usual case (`": 12`, `, "k`): every LDS read counts in this walker.
struct Tok {
    int32_t at, after;
    uint32_t c;
};
__device__ __forceinline__ Tok next_tok(const uint8_t *s, int32_t p, int32_t end) {
    while (p < end) {
        const uint64_t x = lds_u64(s, p);
        const uint64_t ws = bytes_equal_flags(x, ' ') | bytes_equal_flags(x, '\t') | bytes_equal_flags(x, '\r') | bytes_equal_flags(x, '\n');
        const uint64_t other = ~ws & 0x8080808080808080ull;
        if (other) {
            const int i = (__ffsll((unsigned long long)other) - 1) >> 3;
            if (p + i >= end) break;
            Tok t;
            t.at = p + i;
            t.c = (uint32_t)(x >> (8 * i)) & 0xFFu;
            const uint64_t rest = i < 7 ? other >> (8 * (i + 1)) : 0ull;
            t.after = rest ? min(end, t.at + 1 + ((__ffsll((unsigned long long)rest) - 1) >> 3)) : skip_ws_words(s, p + 8, end);
            return t;
        }
        p += 8;
    }
    return Tok{end, end, 0u};
}

__device__ bool parse_line_flex(const uint8_t *s, int32_t s_base, int32_t p, int32_t end, const JsonSpec &spec, const JsonOut &out, int64_t row) {
    p -= s_base;
    end -= s_base;
    while (end > p && is_ws(s[end - 1])) --end;
    p = skip_ws_words(s, p, end);
    if (p >= end || s[p] != '{') return false;
    p = skip_ws_words(s, p + 1, end);
    uint32_t seen = 0;
    for (int member = 0; member < spec.n; ++member) {   // one schema field per member, each once
        // ---- "key"
        if (p >= end || s[p] != '"') return false;
        const int32_t kb = p + 1;
        int32_t ke = kb;
        for (;;) {
            if (ke >= end) return false;
            const uint64_t x = lds_u64(s, ke);
            const uint64_t stop = bytes_equal_flags(x, '"') | bytes_equal_flags(x, '\\') | (((x - 0x2020202020202020ull) & ~x) & 0x8080808080808080ull);
            if (stop) {
                const int at = (__ffsll((unsigned long long)stop) - 1) >> 3;
                ke += at;
                if (((x >> (8 * at)) & 0xFFu) != '"') return false;   // an escape or a control character in a key
                break;
            }
            ke += 8;
        }
        if (ke >= end) return false;
        const int32_t klen = ke - kb;
        if (klen <= 0 || klen >= kMaxName) return false;
        uint64_t kw[kMaxName / 8];
#pragma unroll
        for (int j = 0; j < kMaxName / 8; ++j) {
            const int32_t left = klen - 8 * j;
            kw[j] = left <= 0 ? 0ull : lds_u64(s, kb + 8 * j) & (left >= 8 ? ~0ull : (1ull << (8 * left)) - 1ull);
        }
        int f = -1;
        for (int i = 0; i < spec.n; ++i) {   // (wave-uniform trip: the names come out of the kernel arguments through scalar loads)
            const uint32_t *nm = reinterpret_cast<const uint32_t *>(spec.name[i]);
            bool same = spec.name_len[i] == klen;
#pragma unroll
            for (int j = 0; j < kMaxName / 8; ++j) same = same && kw[j] == ((uint64_t)nm[2 * j] | ((uint64_t)nm[2 * j + 1] << 32));
            if (same) f = i;
        }
        if (f < 0 || ((seen >> f) & 1u)) return false;   // unknown member / repeated key: the general parser's business
        seen |= 1u << f;
        // ---- : value
        const Tok colon = next_tok(s, ke + 1, end);
        if (colon.c != ':') return false;
        p = colon.after;
        if (p >= end) return false;
        int32_t type = 0;
        for (int i = 0; i < spec.n; ++i)
            if (i == f) type = spec.type[i];
        if (type == kUtf8) {
            const int32_t b = p + 1;
            uint64_t x = lds_u64(s, p);
            if ((x & 0xFFu) != '"') return false;
            x |= 0xFFu;
            uint32_t c = 0;
            for (;;) {
                const uint64_t stop = bytes_equal_flags(x, '"') | bytes_equal_flags(x, '\\') | (((x - 0x2020202020202020ull) & ~x) & 0x8080808080808080ull);
                if (stop) {
                    const int at = (__ffsll((unsigned long long)stop) - 1) >> 3;
                    p += at;
                    c = (uint32_t)(x >> (8 * at)) & 0xFFu;
                    break;
                }
                p += 8;
                if (p >= end) return false;
                x = lds_u64(s, p);
            }
            if (p >= end || c != '"') return false;
            int32_t *pairs = nullptr, *ulen = nullptr;
            for (int i = 0; i < spec.n; ++i)
                if (i == f) pairs = out.pairs[i], ulen = out.ulen[i];
            pairs[2 * row] = b + s_base;
            pairs[2 * row + 1] = p + s_base;
            ulen[row] = p - b;
            ++p;
        } else {
            uint64_t x0 = lds_u64(s, p);
            const bool neg = (x0 & 0xFFu) == '-';
            if (neg) x0 = lds_u64(s, ++p);
            uint64_t x1 = 0, x2 = 0;
            const uint64_t m0 = non_digit_flags(x0);
            int total = m0 ? (__ffsll((unsigned long long)m0) - 1) >> 3 : 8;
            if (total == 8 && p + 8 < end) {
                x1 = lds_u64(s, p + 8);
                const uint64_t m1 = non_digit_flags(x1);
                total += m1 ? (__ffsll((unsigned long long)m1) - 1) >> 3 : 8;
                if (total == 16 && p + 16 < end) {
                    x2 = lds_u64(s, p + 16);
                    const uint64_t m2 = non_digit_flags(x2);
                    total += m2 ? (__ffsll((unsigned long long)m2) - 1) >> 3 : 8;
                }
            }
            total = min(total, end - p);
            if (total <= 0 || total > 18 || (total > 1 && (x0 & 0xFFu) == '0')) return false;
            const int lead = total - ((total - 1) & ~7);
            uint64_t v = leading_digits(x0, lead);
            if (total > 8) {
                const int sh = 8 * lead;
                v = v * 100000000ull + leading_digits(lead == 8 ? x1 : (x0 >> sh) | (x1 << (64 - sh)), 8);
                if (total > 16) v = v * 100000000ull + leading_digits(lead == 8 ? x2 : (x1 >> sh) | (x2 << (64 - sh)), 8);
            }
            p += total;
            if (p < end) {   // what follows an integer literal: a delimiter (a fraction or an exponent is the general parser's error)
                const uint32_t c = s[p];
                if (!(c == ',' || c == '}' || is_ws(c))) return false;
            }
            const int64_t sv = neg ? -(int64_t)v : (int64_t)v;
            void *col = nullptr;
            for (int i = 0; i < spec.n; ++i)
                if (i == f) col = out.values[i];
            if (type == kInt32) {
                if (sv < -2147483648ll || sv > 2147483647ll) return false;
                reinterpret_cast<int32_t *>(col)[row] = (int32_t)sv;
            } else {
                reinterpret_cast<int64_t *>(col)[row] = sv;
            }
        }
        // ---- , or }
        const Tok sep = next_tok(s, p, end);
        if (sep.c == '}') {
            p = sep.at + 1;
            break;
        }
        if (sep.c != ',') return false;
        p = sep.after;
    }
    return p == end && seen == (spec.n >= 32 ? ~0u : (1u << spec.n) - 1u);
}

// err[0] = first bad line + 1 (0: none) as atomicMin over (line + 1) stored inverted, err[1] = its code
//
// Two kernels share one frame (256 lines per workgroup, their bytes staged in LDS):
//   kRetry = false : the compact walk only (parse_line_words).  A workgroup that meets a line it cannot take -- or whose lines do not
//                    fit the stage -- raises its flag in block_retry[] and *any_retry; nothing else.  This is the kernel every call runs,
//                    and it carries neither the flat-object walker nor the general parser (with them inlined it needed 412 bytes of
//                    scratch per lane for spilled scalar registers and ran 3x slower on lines it takes itself).
//   kRetry = true  : only the flagged workgroups (all of them when block_retry is null): the flat-object walker, else the general
//                    parser, which also words the errors.  Launched when *any_retry came back non-zero.
template <int kUnroll, bool kRetry>
__global__ __launch_bounds__(kBlock) void json_parse_kernel(const uint8_t *__restrict__ bytes, int64_t n_bytes,
                                                            const int32_t *__restrict__ line_start, int64_t n_lines, JsonSpec spec,
                                                            JsonOut out, int32_t stage_bytes, uint32_t *block_retry, uint32_t *any_retry,
                                                            unsigned long long *err) {
    extern __shared__ __attribute__((aligned(16))) uint8_t s_stage[];  // stage_bytes of dynamic LDS
    __shared__ JsonSpec s_spec;  // (kRetry) lanes index the field names with their own (i, j): from LDS, not from the kernel arguments
    if (kRetry) {
        if (block_retry && !block_retry[blockIdx.x]) return;
        for (int i = threadIdx.x; i < (int)(sizeof(JsonSpec) / 4); i += kBlock)
            reinterpret_cast<uint32_t *>(&s_spec)[i] = reinterpret_cast<const uint32_t *>(&spec)[i];
    }
    const int64_t l0 = (int64_t)blockIdx.x * kParseLines;
    const int64_t l1 = min(l0 + kParseLines, n_lines);
    const int32_t b0 = line_start[l0], b1 = min((int64_t)line_start[l1], n_bytes);
    const int32_t a0 = b0 & ~15;
    const bool staged = b1 - a0 + 16 <= stage_bytes;  // block-uniform; 16 bytes of slack for reads that overrun a line
    // this lane's line: asked for before the staging so that the loads overlap it
    const int64_t line = l0 + threadIdx.x;
    const int64_t line_c = min(line, n_lines - 1);
    const int32_t p = line_start[line_c];
    const int32_t e = min((int64_t)line_start[line_c + 1] - 1, n_bytes);  // without the newline
    if (staged) {
        for (int32_t o = a0 + (int32_t)threadIdx.x * 16; o < b1; o += kBlock * 16) {
            if ((int64_t)o + 16 <= n_bytes) {
                *reinterpret_cast<uint4 *>(s_stage + (o - a0)) = *reinterpret_cast<const uint4 *>(bytes + o);
            } else {
                for (int32_t i = o; i < b1; ++i) s_stage[i - a0] = bytes[i];
            }
        }
    }
    __syncthreads();
    if (!kRetry) {
        const bool bad = line < n_lines && !(staged && parse_line_words<kUnroll>(s_stage, a0, p, e, spec, line, out));
        const bool any = __syncthreads_or(bad);
        if (threadIdx.x == 0) {
            block_retry[blockIdx.x] = any ? 1u : 0u;
            if (any) atomicOr(any_retry, 1u);
        }
        return;
    }
    if (line >= n_lines) return;
    uint32_t rc;
    if (staged) {
        const Text<true> t{bytes, s_stage, a0};
        rc = parse_line_flex(s_stage, a0, p, e, spec, out, line) ? 0u : parse_line(t, p, e, s_spec, line, out);   // (compact lines are flat objects, too)
    } else {
        const Text<false> t{bytes, nullptr, 0};
        rc = parse_line_fast(t, p, e, s_spec, line, out) ? 0u : parse_line(t, p, e, s_spec, line, out);
    }
    if (rc) atomicMin(err, ((unsigned long long)line << 8) | rc);
}

// Utf8 field after the parse: len[i] = |ulen[i]| (unescaped length), any = some value needs unescaping
__global__ __launch_bounds__(kBlock) void json_string_lengths_kernel(const int32_t *__restrict__ ulen, int64_t n, int32_t *__restrict__ len,
                                                                     uint32_t *any) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    bool esc = false;
    if (i < n) {
        const int32_t u = ulen[i];
        esc = u < 0;
        len[i + 1] = esc ? -(u + 1) : u;
    }
    if (i == 0) len[0] = 0;
    if (__ballot(esc) && lane_id() == 0) atomicOr(any, 1u);
}

__device__ __forceinline__ int utf8_put(uint8_t *dst, int cp) {
    if (cp < 0x80) {
        dst[0] = (uint8_t)cp;
        return 1;
    }
    if (cp < 0x800) {
        dst[0] = (uint8_t)(0xC0 | (cp >> 6));
        dst[1] = (uint8_t)(0x80 | (cp & 0x3F));
        return 2;
    }
    if (cp < 0x10000) {
        dst[0] = (uint8_t)(0xE0 | (cp >> 12));
        dst[1] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F));
        dst[2] = (uint8_t)(0x80 | (cp & 0x3F));
        return 3;
    }
    dst[0] = (uint8_t)(0xF0 | (cp >> 18));
    dst[1] = (uint8_t)(0x80 | ((cp >> 12) & 0x3F));
    dst[2] = (uint8_t)(0x80 | ((cp >> 6) & 0x3F));
    dst[3] = (uint8_t)(0x80 | (cp & 0x3F));
    return 4;
}

// Byte a single-character escape stands for (the character after the backslash; `"`, `\` and `/` stand for themselves).
__device__ __attribute__((noinline)) uint32_t escape_value(uint32_t d) {
    uint32_t v = d;
    if (d == 'b') v = 8;
    if (d == 'f') v = 12;
    if (d == 'n') v = 10;
    if (d == 'r') v = 13;
    if (d == 't') v = 9;
    return v;
}

// The unescaping copy of a field that holds escapes: one lane per value (offsets already scanned).  (One store site and
// one advance per trip: the first version, with a store in each branch, faulted on "\b\f"-like values -- two different
// single-character escapes in a row -- although the same source ran clean on the host under ASan.)
__global__ __launch_bounds__(kBlock) void json_unescape_kernel(const uint8_t *bytes, const int32_t *pairs, const int32_t *off, int64_t n,
                                                               uint8_t *dst) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    int32_t p = pairs[2 * i];
    const int32_t e = pairs[2 * i + 1];
    int64_t o = off[i];
    while (p < e) {
        uint32_t c = bytes[p];
        int32_t adv = 1;
        if (c == '\\') {
            const uint32_t d = bytes[p + 1];
            adv = 2;
            if (d == 'u') {
                int cp = 0;
                for (int k = 0; k < 4; ++k) cp = cp * 16 + hex_of(bytes[p + 2 + k]);
                adv = 6;
                if (cp >= 0xD800 && cp < 0xDC00) {
                    int lo = 0;
                    for (int k = 0; k < 4; ++k) lo = lo * 16 + hex_of(bytes[p + 8 + k]);
                    adv = 12;
                    cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
                }
                if (cp >= 0x80) {  // multi-byte: everything but the last byte here, the last one at the common store
                    uint8_t tmp[4];
                    const int len = utf8_put(tmp, cp);
                    for (int k = 0; k + 1 < len; ++k) dst[o++] = tmp[k];
                    c = tmp[len - 1];
                } else {
                    c = (uint32_t)cp;
                }
            } else {
                c = escape_value(d);
            }
        }
        dst[o++] = (uint8_t)c;
        p += adv;
    }
}

__global__ __launch_bounds__(kBlock) void json_even_rows_kernel(int32_t *__restrict__ rows, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (i < n) rows[i] = (int32_t)(2 * i);
}

}  // namespace

extern "C" {

int flockgpu_json_lines_decode(flockgpu_ctx *ctx, const uint8_t *json, int64_t n_bytes, const flockgpu_json_field *fields,
                               int32_t n_fields, flockgpu_json_column *out, int64_t *rows) {
    if (!ctx) return FLOCKGPU_ERR_INVALID;
    if (!fields || !out || !rows || n_bytes < 0 || n_fields < 1 || (n_bytes > 0 && !json))
        return fail(ctx, FLOCKGPU_ERR_INVALID, "json: null argument");
    if (n_fields > kMaxFields) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "json: more than %d fields", kMaxFields);
    if (n_bytes >= (int64_t(1) << 31) - 16)
        return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "json: at most 2^31 bytes of text per call (Arrow Utf8 offsets are int32)");
    if (reinterpret_cast<uintptr_t>(json) & 15) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "json: the text must be 16-byte aligned");
    JsonSpec spec{};
    spec.n = n_fields;
    for (int f = 0; f < n_fields; ++f) {
        if (!fields[f].name || fields[f].type < kInt32 || fields[f].type > kUtf8) return fail(ctx, FLOCKGPU_ERR_INVALID, "json: bad field %d", f);
        const size_t len = std::strlen(fields[f].name);
        if (len == 0 || len >= (size_t)kMaxName) return fail(ctx, FLOCKGPU_ERR_UNSUPPORTED, "json: field name longer than %d bytes", kMaxName - 1);
        spec.type[f] = fields[f].type;
        spec.name_len[f] = (int32_t)len;
        std::memcpy(spec.name[f], fields[f].name, len);
        char pre[sizeof spec.pre[0]] = {};
        pre[0] = f == 0 ? '{' : ',';
        pre[1] = '"';
        std::memcpy(pre + 2, fields[f].name, len);
        pre[len + 2] = '"';
        pre[len + 3] = ':';
        spec.pre_len[f] = (int32_t)len + 4;
        std::memcpy(spec.pre[f], pre, sizeof spec.pre[f]);
    }
    FG_HIP(ctx, hipSetDevice(ctx->device));
    for (int f = 0; f < n_fields; ++f) out[f] = flockgpu_json_column{};
    *rows = 0;

    // ---- line index
    const int64_t tiles = div_up(std::max<int64_t>(n_bytes, 1), kNlTile);
    uint32_t *counts = nullptr;
    uint64_t *tile_base = nullptr, *h_total = nullptr;
    uint8_t *h_last = nullptr;
    FG_TRY(arena_get_t(ctx, "json.nl_counts", (size_t)tiles * kWavesPerBlock + 4, &counts));
    FG_TRY(arena_get_t(ctx, "json.nl_base", (size_t)tiles + 1, &tile_base));
    uint64_t *masks = nullptr;
    FG_TRY(arena_get_t(ctx, "json.nl_masks", (size_t)tiles * kBlock, &masks));
    FG_TRY(pinned_get_t(ctx, "json.nl_total", 2, &h_total));
    h_last = reinterpret_cast<uint8_t *>(h_total + 1);
    if (n_bytes == 0) return FLOCKGPU_OK;
    {
        LaunchScope ls(ctx, "json_newline_count_kernel");
        hipLaunchKernelGGL(json_newline_count_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, json, n_bytes, masks, counts);
    }
    FG_TRY(check_launch(ctx, "json_newline_count_kernel"));
    FG_TRY(launch_tile_scan(ctx, counts, (int32_t)tiles, tile_base, nullptr, 0, nullptr));
    FG_HIP(ctx, hipMemcpyAsync(h_total, tile_base + tiles, sizeof(uint64_t), hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipMemcpyAsync(h_last, json + n_bytes - 1, 1, hipMemcpyDeviceToHost, ctx->stream));
    FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
    const int64_t n_newlines = (int64_t)*h_total;
    const bool open_tail = *h_last != 0x0A;  // the last line has no newline
    const int64_t n_lines = n_newlines + (open_tail ? 1 : 0);
    int32_t *line_start = nullptr;
    FG_TRY(arena_get_t(ctx, "json.line_start", (size_t)n_lines + 2, &line_start));
    {
        LaunchScope ls(ctx, "json_newline_emit_kernel");
        hipLaunchKernelGGL(json_newline_emit_kernel, dim3((unsigned)tiles), dim3(kBlock), 0, ctx->stream, masks, counts, tile_base,
                           line_start);
    }
    FG_TRY(check_launch(ctx, "json_newline_emit_kernel"));
    if (open_tail) {  // a virtual newline at n_bytes closes the last line
        const int32_t v = (int32_t)(n_bytes + 1);
        int32_t *h_v = nullptr;
        FG_TRY(pinned_get_t(ctx, "json.tail", 1, &h_v));
        *h_v = v;
        FG_HIP(ctx, hipMemcpyAsync(line_start + n_lines, h_v, sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    }

    // ---- parse
    JsonOut jo{};
    for (int f = 0; f < n_fields; ++f) {
        char name[48];
        if (spec.type[f] == kUtf8) {
            snprintf(name, sizeof name, "json.pairs.%d", f);
            FG_TRY(arena_get_t(ctx, name, (size_t)2 * n_lines + 4, &jo.pairs[f]));
            snprintf(name, sizeof name, "json.ulen.%d", f);
            FG_TRY(arena_get_t(ctx, name, (size_t)n_lines + 4, &jo.ulen[f]));
        } else {
            snprintf(name, sizeof name, "json.values.%d", f);
            void *p = nullptr;
            FG_TRY(arena_get(ctx, name, ((size_t)n_lines + 4) * (spec.type[f] == kInt32 ? 4 : 8), &p));
            jo.values[f] = p;
        }
    }
    unsigned long long *d_err = nullptr, *h_err = nullptr;
    uint32_t *d_any = nullptr, *h_any = nullptr;
    FG_TRY(arena_get_t(ctx, "json.err", 1, &d_err));
    FG_TRY(pinned_get_t(ctx, "json.err", 1, &h_err));
    FG_TRY(arena_get_t(ctx, "json.any_escape", kMaxFields, &d_any));
    FG_TRY(pinned_get_t(ctx, "json.any_escape", kMaxFields, &h_any));
    FG_HIP(ctx, hipMemsetAsync(d_err, 0xFF, sizeof(unsigned long long), ctx->stream));
    uint32_t *d_retry = nullptr, *d_any_retry = nullptr, *h_any_retry = nullptr;
    const int64_t n_blocks = div_up(std::max<int64_t>(n_lines, 1), kParseLines);
    FG_TRY(arena_get_t(ctx, "json.block_retry", (size_t)n_blocks + 4, &d_retry));
    FG_TRY(arena_get_t(ctx, "json.any_retry", 4, &d_any_retry));
    FG_TRY(pinned_get_t(ctx, "json.any_retry", 4, &h_any_retry));
    int32_t *soff[kMaxFields] = {};
    for (int f = 0; f < n_fields; ++f) {
        if (spec.type[f] != kUtf8) continue;
        char name[48];
        snprintf(name, sizeof name, "json.str_off.%d", f);
        FG_TRY(arena_get_t(ctx, name, (size_t)n_lines + 4, &soff[f]));
    }
    // LDS per workgroup follows the text: 1.25 x the average bytes of 256 lines (+ alignment slack), so short lines
    // (bids: 74 B -> 24 KB, six workgroups per CU) do not pay for the longest relation; a workgroup whose lines do
    // not fit reads them from global memory instead.
    const int64_t want = (n_bytes / std::max<int64_t>(n_lines, 1) + 1) * kParseLines * 5 / 4 + 64;
    const int32_t stage_bytes = (int32_t)std::min<int64_t>(kStageBytes, (want + 1023) & ~int64_t(1023));
    // hint = {route, calls on that route}: when most workgroups of a call ended in the second kernel (a writer other than serde_json)
    // the next calls go there directly; every sixteenth of them tries the compact walk first again
    std::vector<int64_t> &hint = ctx->host_i64["json.retry_hint"];
    if (hint.size() < 2) hint.assign(2, 0);
    const bool straight = hint[0] == 1 && (++hint[1] % 16) != 0;
    *h_any_retry = 0;
    // one round = a parse kernel, the string lengths behind it, ONE host wait for the errors, the escape flags and the retry flag
    auto round = [&](bool second) -> int {
        if (n_lines > 0) {
            if (!second) FG_HIP(ctx, hipMemsetAsync(d_any_retry, 0, sizeof(uint32_t), ctx->stream));
            FG_HIP(ctx, hipMemsetAsync(d_any, 0, sizeof(uint32_t) * kMaxFields, ctx->stream));
            auto kernel = second ? (n_fields <= 4 ? json_parse_kernel<4, true> : n_fields <= 8 ? json_parse_kernel<8, true> : json_parse_kernel<kMaxFields, true>)
                                 : (n_fields <= 4 ? json_parse_kernel<4, false> : n_fields <= 8 ? json_parse_kernel<8, false> : json_parse_kernel<kMaxFields, false>);
            {
                LaunchScope ls(ctx, second ? "json_parse_retry_kernel" : "json_parse_kernel");
                hipLaunchKernelGGL(kernel, dim3((unsigned)n_blocks), dim3(kBlock), (size_t)stage_bytes, ctx->stream, json, n_bytes, line_start, n_lines, spec, jo,
                                   stage_bytes, second && straight ? (uint32_t *)nullptr : d_retry, d_any_retry, d_err);
            }
            FG_TRY(check_launch(ctx, second ? "json_parse_retry_kernel" : "json_parse_kernel"));
            for (int f = 0; f < n_fields; ++f) {
                if (spec.type[f] != kUtf8) continue;
                hipLaunchKernelGGL(json_string_lengths_kernel, dim3((unsigned)div_up(n_lines, kBlock)), dim3(kBlock), 0, ctx->stream, jo.ulen[f], n_lines, soff[f],
                                   d_any + f);
                FG_TRY(check_launch(ctx, "json_string_lengths_kernel"));
            }
            if (!second) FG_HIP(ctx, hipMemcpyAsync(h_any_retry, d_any_retry, sizeof(uint32_t), hipMemcpyDeviceToHost, ctx->stream));
        }
        FG_HIP(ctx, hipMemcpyAsync(h_err, d_err, sizeof(unsigned long long), hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipMemcpyAsync(h_any, d_any, sizeof(uint32_t) * kMaxFields, hipMemcpyDeviceToHost, ctx->stream));
        FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
        return FLOCKGPU_OK;
    };
    if (straight) {
        FG_TRY(round(true));
    } else {
        FG_TRY(round(false));
        if (*h_any_retry) {   // some workgroup met a line the compact walk does not take
            FG_TRY(round(true));
            std::vector<uint32_t> flags((size_t)n_blocks);   // how many needed it: the next calls' route
            FG_HIP(ctx, hipMemcpy(flags.data(), d_retry, sizeof(uint32_t) * (size_t)n_blocks, hipMemcpyDeviceToHost));
            int64_t flagged = 0;
            for (uint32_t f : flags) flagged += f != 0;
            hint[0] = flagged * 2 > n_blocks ? 1 : 0;
        } else {
            hint[0] = 0;
        }
    }
    if (*h_err != ~0ull) {
        const long long line = (long long)(*h_err >> 8);
        const uint32_t code = (uint32_t)(*h_err & 0xFF);
        static const char *what[] = {"", "malformed JSON", "number with a fraction / exponent (integer literal expected)",
                                     "a schema field is missing", "blank line", "escape sequence in a key", "integer out of range"};
        return fail(ctx, code == kErrNumber || code == kErrBlank || code == kErrKeyEscape ? FLOCKGPU_ERR_UNSUPPORTED : FLOCKGPU_ERR_INVALID,
                    "json: line %lld: %s", line + 1, what[code < 7 ? code : 1]);
    }
    int32_t *even = nullptr;
    bool have_even = false;
    for (int f = 0; f < n_fields; ++f) {
        if (spec.type[f] != kUtf8) {
            out[f].values = jo.values[f];
            continue;
        }
        char name[48];
        if (!h_any[f]) {  // a plain take of byte ranges of the input
            if (!have_even) {
                FG_TRY(arena_get_t(ctx, "json.even_rows", (size_t)n_lines + 4, &even));
                if (n_lines > 0) {
                    hipLaunchKernelGGL(json_even_rows_kernel, dim3((unsigned)div_up(n_lines, kBlock)), dim3(kBlock), 0, ctx->stream, even, n_lines);
                    FG_TRY(check_launch(ctx, "json_even_rows_kernel"));
                }
                have_even = true;
            }
            snprintf(name, sizeof name, "json.utf8.%d", f);
            flockgpu_utf8 src{jo.pairs[f], json};
            FG_TRY(gather_utf8(ctx, name, src, even, n_lines, &out[f].utf8, &out[f].utf8_bytes));
        } else {
            FG_TRY(inclusive_scan_i32(ctx, "json.str_scan", soff[f], n_lines + 1));
            int32_t *h_tot = nullptr;
            FG_TRY(pinned_get_t(ctx, "json.str_total", 1, &h_tot));
            FG_HIP(ctx, hipMemcpyAsync(h_tot, soff[f] + n_lines, sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            FG_HIP(ctx, hipStreamSynchronize(ctx->stream));
            uint8_t *dst = nullptr;
            snprintf(name, sizeof name, "json.str_bytes.%d", f);
            FG_TRY(arena_get_t(ctx, name, (size_t)*h_tot + 16, &dst));
            if (n_lines > 0) {
                LaunchScope ls(ctx, "json_unescape_kernel");
                hipLaunchKernelGGL(json_unescape_kernel, dim3((unsigned)div_up(n_lines, kBlock)), dim3(kBlock), 0, ctx->stream, json, jo.pairs[f],
                                   soff[f], n_lines, dst);
            }
            FG_TRY(check_launch(ctx, "json_unescape_kernel"));
            out[f].utf8.offsets = soff[f];
            out[f].utf8.data = dst;
            out[f].utf8_bytes = *h_tot;
        }
    }
    *rows = n_lines;
    return FLOCKGPU_OK;
}

}  // extern "C"
