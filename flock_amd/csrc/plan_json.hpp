// Minimal JSON reader for the serde_json text of the reference's physical plans (flock/src/runtime/context.rs:477-480,
// flock/src/distributed_plan/stage.rs:271).  Host-side only.
#pragma once
#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <utility>
#include <vector>

namespace flockgpu {

struct JValue;
using JPtr = std::shared_ptr<JValue>;
struct JValue {
    enum Kind { Null, Bool, Num, Str, Arr, Obj } kind = Null;
    bool b = false;
    double num = 0;
    int64_t inum = 0;
    bool is_int = false;
    bool is_big_unsigned = false;   // inum holds the bit pattern of an unsigned value above INT64_MAX
    std::string str;
    std::vector<JPtr> arr;
    std::vector<std::pair<std::string, JPtr>> obj;
    const JValue *get(const char *key) const {
        for (auto &kv : obj)
            if (kv.first == key) return kv.second.get();
        return nullptr;
    }
    std::string s(const char *key) const {
        const JValue *v = get(key);
        return v && v->kind == Str ? v->str : std::string();
    }
};

struct JParser {
    const char *p, *end;
    std::string err;
    void ws() { while (p < end && (*p == ' ' || *p == '\n' || *p == '\t' || *p == '\r')) ++p; }
    bool fail(const char *m) { if (err.empty()) err = m; return false; }
    bool parse_string(std::string &out) {
        if (p >= end || *p != '"') return fail("expected string");
        ++p;
        while (p < end && *p != '"') {
            if (*p == '\\') {
                if (++p >= end) return fail("bad escape");
                switch (*p) {
                    case 'n': out += '\n'; break;
                    case 't': out += '\t'; break;
                    case 'r': out += '\r'; break;
                    case 'b': out += '\b'; break;
                    case 'f': out += '\f'; break;
                    case 'u': {
                        if (end - p < 5) return fail("bad \\u escape");
                        unsigned v = (unsigned)strtoul(std::string(p + 1, p + 5).c_str(), nullptr, 16);
                        if (v < 0x80) out += (char)v;
                        else if (v < 0x800) { out += (char)(0xC0 | (v >> 6)); out += (char)(0x80 | (v & 0x3F)); }
                        else { out += (char)(0xE0 | (v >> 12)); out += (char)(0x80 | ((v >> 6) & 0x3F)); out += (char)(0x80 | (v & 0x3F)); }
                        p += 4;
                        break;
                    }
                    default: out += *p;
                }
                ++p;
            } else {
                out += *p++;
            }
        }
        if (p >= end) return fail("unterminated string");
        ++p;
        return true;
    }
    bool parse(JPtr &out, int depth = 0) {
        if (depth > 200) return fail("plan nested too deep");
        ws();
        if (p >= end) return fail("unexpected end");
        out = std::make_shared<JValue>();
        if (*p == '{') {
            out->kind = JValue::Obj;
            ++p; ws();
            if (p < end && *p == '}') { ++p; return true; }
            for (;;) {
                ws();
                std::string key;
                if (!parse_string(key)) return false;
                ws();
                if (p >= end || *p != ':') return fail("expected ':'");
                ++p;
                JPtr v;
                if (!parse(v, depth + 1)) return false;
                out->obj.emplace_back(std::move(key), v);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == '}') { ++p; return true; }
                return fail("expected ',' or '}'");
            }
        }
        if (*p == '[') {
            out->kind = JValue::Arr;
            ++p; ws();
            if (p < end && *p == ']') { ++p; return true; }
            for (;;) {
                JPtr v;
                if (!parse(v, depth + 1)) return false;
                out->arr.push_back(v);
                ws();
                if (p < end && *p == ',') { ++p; continue; }
                if (p < end && *p == ']') { ++p; return true; }
                return fail("expected ',' or ']'");
            }
        }
        if (*p == '"') { out->kind = JValue::Str; return parse_string(out->str); }
        if (!strncmp(p, "true", std::min<size_t>(4, end - p)) && end - p >= 4) { out->kind = JValue::Bool; out->b = true; p += 4; return true; }
        if (!strncmp(p, "false", std::min<size_t>(5, end - p)) && end - p >= 5) { out->kind = JValue::Bool; p += 5; return true; }
        if (!strncmp(p, "null", std::min<size_t>(4, end - p)) && end - p >= 4) { p += 4; return true; }
        const char *q = p;
        bool is_int = true;
        if (q < end && (*q == '-' || *q == '+')) ++q;
        while (q < end && ((*q >= '0' && *q <= '9') || *q == '.' || *q == 'e' || *q == 'E' || *q == '-' || *q == '+')) {
            if (*q == '.' || *q == 'e' || *q == 'E') is_int = false;
            ++q;
        }
        if (q == p) return fail("unexpected character");
        std::string tok(p, q);
        out->kind = JValue::Num;
        out->num = strtod(tok.c_str(), nullptr);
        out->is_int = is_int;
        if (is_int) {
            out->inum = strtoll(tok.c_str(), nullptr, 10);
            if (tok[0] != '-' && out->inum == INT64_MAX && tok != "9223372036854775807") {   // a UInt64 beyond 2^63: its bit pattern
                out->inum = (int64_t)strtoull(tok.c_str(), nullptr, 10);
                out->is_big_unsigned = true;
            }
        }
        p = q;
        return true;
    }
};


}  // namespace flockgpu
