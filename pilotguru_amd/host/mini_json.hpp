// mini_json.hpp -- just enough JSON to read the recorder files fit_motion takes
// (rotations / accelerations / locations: an object holding one array of flat objects of numbers)
// and to keep integers apart from reals the way nlohmann::json does (time_usec is a long).
#pragma once
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

namespace pgorb {

struct JsonValue {
    enum Kind { Null, Bool, Int, Real, String, Array, Object } kind = Null;
    bool b = false;
    int64_t i = 0;
    double d = 0;
    std::string s;
    std::vector<JsonValue> a;
    std::map<std::string, JsonValue> o;

    bool has(const std::string& k) const { return kind == Object && o.count(k); }
    const JsonValue& at(const std::string& k) const { static const JsonValue none; auto it = o.find(k); return it == o.end() ? none : it->second; }
    bool is_number() const { return kind == Int || kind == Real; }
    double as_double() const { return kind == Int ? (double)i : d; }          // nlohmann: integer -> double conversion
    int64_t as_int() const { return kind == Int ? i : (int64_t)d; }            // nlohmann: double -> long truncates
};

class JsonParser {
public:
    explicit JsonParser(const std::string& text) : t_(text) {}
    bool parse(JsonValue& out) { ws(); if (!value(out)) return false; ws(); return p_ == t_.size(); }

private:
    const std::string& t_;
    size_t p_ = 0;
    void ws() { while (p_ < t_.size() && (t_[p_] == ' ' || t_[p_] == '\n' || t_[p_] == '\t' || t_[p_] == '\r')) p_++; }
    bool lit(const char* w) { const size_t n = strlen(w); if (t_.compare(p_, n, w) == 0) { p_ += n; return true; } return false; }
    bool string(std::string& out)
    {
        if (t_[p_] != '"') return false;
        for (p_++; p_ < t_.size() && t_[p_] != '"'; p_++) {
            if (t_[p_] == '\\' && p_ + 1 < t_.size()) {
                const char c = t_[++p_];
                out += c == 'n' ? '\n' : c == 't' ? '\t' : c == 'r' ? '\r' : c == 'b' ? '\b' : c == 'f' ? '\f' : c;   // \uXXXX not needed here
            } else out += t_[p_];
        }
        if (p_ >= t_.size()) return false;
        p_++;
        return true;
    }
    bool number(JsonValue& v)
    {
        const size_t b = p_;
        bool real = false;
        if (p_ < t_.size() && t_[p_] == '-') p_++;
        while (p_ < t_.size() && (isdigit((unsigned char)t_[p_]) || t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E' || t_[p_] == '+' || t_[p_] == '-')) {
            if (t_[p_] == '.' || t_[p_] == 'e' || t_[p_] == 'E') real = true;
            p_++;
        }
        if (p_ == b) return false;
        const std::string n = t_.substr(b, p_ - b);
        if (real) { v.kind = JsonValue::Real; v.d = strtod(n.c_str(), nullptr); }
        else { v.kind = JsonValue::Int; v.i = strtoll(n.c_str(), nullptr, 10); }
        return true;
    }
    bool value(JsonValue& v)
    {
        if (p_ >= t_.size()) return false;
        const char c = t_[p_];
        if (c == '{') {
            v.kind = JsonValue::Object; p_++; ws();
            if (p_ < t_.size() && t_[p_] == '}') { p_++; return true; }
            for (;;) {
                std::string k; ws();
                if (!string(k)) return false;
                ws(); if (p_ >= t_.size() || t_[p_++] != ':') return false;
                ws(); if (!value(v.o[k])) return false;
                ws(); if (p_ >= t_.size()) return false;
                if (t_[p_] == ',') { p_++; continue; }
                if (t_[p_] == '}') { p_++; return true; }
                return false;
            }
        }
        if (c == '[') {
            v.kind = JsonValue::Array; p_++; ws();
            if (p_ < t_.size() && t_[p_] == ']') { p_++; return true; }
            for (;;) {
                v.a.emplace_back(); ws();
                if (!value(v.a.back())) return false;
                ws(); if (p_ >= t_.size()) return false;
                if (t_[p_] == ',') { p_++; continue; }
                if (t_[p_] == ']') { p_++; return true; }
                return false;
            }
        }
        if (c == '"') { v.kind = JsonValue::String; return string(v.s); }
        if (lit("true")) { v.kind = JsonValue::Bool; v.b = true; return true; }
        if (lit("false")) { v.kind = JsonValue::Bool; v.b = false; return true; }
        if (lit("null")) { v.kind = JsonValue::Null; return true; }
        return number(v);
    }
};

inline bool read_json_file(const std::string& path, JsonValue& out)
{
    std::ifstream f(path);
    if (!f.good()) return false;
    std::stringstream ss; ss << f.rdbuf();
    const std::string text = ss.str();
    return JsonParser(text).parse(out);
}

}  // namespace pgorb
