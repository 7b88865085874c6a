// json.cpp -- recursive-descent JSON parser for model files (see ic2/json.h).
#include <cctype>
#include <charconv>
#include <cmath>
#include <cstdlib>

#include "ic2/json.h"

#include <locale.h>

namespace snn {
namespace json {
namespace {
struct P {
    const std::string& s;
    size_t i = 0;
    int depth = 0;
    static constexpr int kMaxDepth = 64;
    std::string err;
    explicit P(const std::string& t) : s(t) {}
    void ws() {
        while (i < s.size() && isspace(static_cast<unsigned char>(s[i]))) ++i;
    }
    bool fail(const char* m) {
        if (err.empty()) err = std::string(m) + " at offset " + std::to_string(i);
        return false;
    }
    bool str(std::string& out) {
        if (s[i] != '"') return fail("expected string");
        ++i;
        out.clear();
        while (i < s.size() && s[i] != '"') {
            if (s[i] == '\\' && i + 1 < s.size()) {
                char c = s[i + 1];
                i += 2;
                switch (c) {
                case 'n': out += '\n'; break;
                case 't': out += '\t'; break;
                case 'r': out += '\r'; break;
                case 'b': out += '\b'; break;
                case 'f': out += '\f'; break;
                case 'u':
                    if (i + 4 <= s.size()) {
                        out += static_cast<char>(strtol(s.substr(i, 4).c_str(), nullptr, 16) & 0x7F);
                        i += 4;
                    }
                    break;
                default: out += c;
                }
            } else {
                out += s[i++];
            }
        }
        if (i >= s.size()) return fail("unterminated string");
        ++i;
        return true;
    }
    bool value(Value& v) {
        ws();
        if (i >= s.size()) return fail("unexpected end");
        const char c = s[i];
        struct DepthGuard { // nesting is bounded: a hostile or corrupt file must not overflow the stack
            int& d;
            explicit DepthGuard(int& dd) : d(dd) { ++d; }
            ~DepthGuard() { --d; }
        } guard(depth);
        if (depth > kMaxDepth) return fail("nesting deeper than 64 levels");
        if (c == '{') {
            ++i;
            v.type = Value::ObjectT;
            v.obj = std::make_shared<Object>();
            ws();
            if (i < s.size() && s[i] == '}') {
                ++i;
                return true;
            }
            while (true) {
                ws();
                std::string k;
                if (i >= s.size() || !str(k)) return fail("expected key");
                ws();
                if (i >= s.size() || s[i] != ':') return fail("expected ':'");
                ++i;
                Value child;
                if (!value(child)) return false;
                (*v.obj)[k] = std::move(child);
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == '}') {
                    ++i;
                    return true;
                }
                return fail("expected ',' or '}'");
            }
        }
        if (c == '[') {
            ++i;
            v.type = Value::ArrayT;
            v.arr = std::make_shared<Array>();
            ws();
            if (i < s.size() && s[i] == ']') {
                ++i;
                return true;
            }
            while (true) {
                Value child;
                if (!value(child)) return false;
                v.arr->push_back(std::move(child));
                ws();
                if (i < s.size() && s[i] == ',') {
                    ++i;
                    continue;
                }
                if (i < s.size() && s[i] == ']') {
                    ++i;
                    return true;
                }
                return fail("expected ',' or ']'");
            }
        }
        if (c == '"') {
            v.type = Value::String;
            return str(v.str);
        }
        if (!s.compare(i, 4, "true")) {
            v.type = Value::Bool;
            v.b = true;
            i += 4;
            return true;
        }
        if (!s.compare(i, 5, "false")) {
            v.type = Value::Bool;
            v.b = false;
            i += 5;
            return true;
        }
        if (!s.compare(i, 4, "null")) {
            v.type = Value::Null;
            i += 4;
            return true;
        }
        // JSON number grammar, validated here: -? (0 | [1-9][0-9]*) (. [0-9]+)? ([eE] [+-]? [0-9]+)?  -- "nan", "inf", hex floats and a
        // locale's decimal comma are not numbers (picojson, the reference's parser, rejects them too); the value itself is converted by
        // std::from_chars, which does not look at the process locale (strtod does)
        size_t j = i;
        auto digits = [&]() {
            const size_t j0 = j;
            while (j < s.size() && s[j] >= '0' && s[j] <= '9') ++j;
            return j > j0;
        };
        if (j < s.size() && s[j] == '-') ++j;
        if (j < s.size() && s[j] == '0') ++j;
        else if (!digits()) return fail("unexpected character");
        if (j < s.size() && s[j] == '.') {
            ++j;
            if (!digits()) return fail("digits expected after the decimal point");
        }
        if (j < s.size() && (s[j] == 'e' || s[j] == 'E')) {
            ++j;
            if (j < s.size() && (s[j] == '+' || s[j] == '-')) ++j;
            if (!digits()) return fail("digits expected in the exponent");
        }
        double d = 0.0;
        const auto res = std::from_chars(s.data() + i, s.data() + j, d);
        if (res.ec == std::errc::result_out_of_range) {
            // from_chars reports overflow AND underflow this way and leaves d untouched: let strtod (whose span was validated above: JSON's number
            // grammar) tell them apart -- +-HUGE_VAL on overflow, +-0 / a denormal on underflow
            // ... in the "C" locale explicitly (strtod_l): under a process locale whose radix character is ',' plain strtod would stop at the '.'
            static const locale_t cLocale = newlocale(LC_ALL_MASK, "C", static_cast<locale_t>(0));
            const std::string span(s.data() + i, j - i);
            d = cLocale ? strtod_l(span.c_str(), nullptr, cLocale) : strtod(span.c_str(), nullptr);
        }
        else if (res.ec != std::errc() || res.ptr != s.data() + j) return fail("malformed number");
        v.type = Value::Number;
        v.num = d;
        i = j;
        return true;
    }
};
} // namespace

std::string parse(Value& out, const std::string& text) {
    P p(text);
    if (!p.value(out)) return p.err;
    p.ws();
    if (p.i != text.size()) return "trailing characters at offset " + std::to_string(p.i);
    return "";
}
} // namespace json
} // namespace snn
