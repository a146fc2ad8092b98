// Minimal JSON reader for scene files: objects, arrays, strings, numbers, true/false/null.
// Numbers remember whether they were written as floating point ('.', 'e' or 'E' present), because the
// reference stringifies ints and floats differently (reference src/core/Scene/SceneParser.cpp:44-69).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace rth {

struct JsonValue {
    enum Kind { Null, Bool, Int, Float, String, Array, Object } kind = Null;
    bool b = false;
    long long i = 0;
    double f = 0;
    std::string s;
    std::vector<JsonValue> arr;
    std::vector<std::pair<std::string, JsonValue>> obj;   // file order

    const JsonValue *find(const std::string &k) const {
        for (auto &kv : obj) if (kv.first == k) return &kv.second;
        return nullptr;
    }
};

class JsonParser {
public:
    explicit JsonParser(const std::string &text) : t(text) {}
    JsonValue parse() {
        JsonValue v = value();
        ws();
        if (p != t.size()) fail("trailing characters");
        return v;
    }

private:
    const std::string &t;
    size_t p = 0;
    [[noreturn]] void fail(const char *m) { throw std::runtime_error(std::string("json: ") + m + " at offset " + std::to_string(p)); }
    void ws() { while (p < t.size() && (t[p] == ' ' || t[p] == '\t' || t[p] == '\n' || t[p] == '\r')) ++p; }
    JsonValue value() {
        ws();
        if (p >= t.size()) fail("unexpected end");
        char c = t[p];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { JsonValue v; v.kind = JsonValue::String; v.s = string(); return v; }
        if (t.compare(p, 4, "true") == 0) { p += 4; JsonValue v; v.kind = JsonValue::Bool; v.b = true; return v; }
        if (t.compare(p, 5, "false") == 0) { p += 5; JsonValue v; v.kind = JsonValue::Bool; v.b = false; return v; }
        if (t.compare(p, 4, "null") == 0) { p += 4; return JsonValue(); }
        return number();
    }
    std::string string() {
        ++p;
        std::string out;
        while (p < t.size() && t[p] != '"') {
            if (t[p] == '\\' && p + 1 < t.size()) {
                char e = t[p + 1];
                out += (e == 'n') ? '\n' : (e == 't') ? '\t' : e;
                p += 2;
            } else {
                out += t[p++];
            }
        }
        if (p >= t.size()) fail("unterminated string");
        ++p;
        return out;
    }
    JsonValue number() {
        size_t s = p;
        bool isf = false;
        if (p < t.size() && (t[p] == '-' || t[p] == '+')) ++p;
        while (p < t.size() && (isdigit((unsigned char)t[p]) || t[p] == '.' || t[p] == 'e' || t[p] == 'E' || t[p] == '-' || t[p] == '+')) {
            if (t[p] == '.' || t[p] == 'e' || t[p] == 'E') isf = true;
            ++p;
        }
        if (s == p) fail("bad token");
        std::string tok = t.substr(s, p - s);
        JsonValue v;
        if (isf) { v.kind = JsonValue::Float; v.f = strtod(tok.c_str(), nullptr); }
        else { v.kind = JsonValue::Int; v.i = strtoll(tok.c_str(), nullptr, 10); v.f = (double)v.i; }
        return v;
    }
    JsonValue array() {
        ++p;
        JsonValue v; v.kind = JsonValue::Array;
        ws();
        if (p < t.size() && t[p] == ']') { ++p; return v; }
        for (;;) {
            v.arr.push_back(value());
            ws();
            if (p < t.size() && t[p] == ',') { ++p; continue; }
            if (p < t.size() && t[p] == ']') { ++p; break; }
            fail("expected , or ]");
        }
        return v;
    }
    JsonValue object() {
        ++p;
        JsonValue v; v.kind = JsonValue::Object;
        ws();
        if (p < t.size() && t[p] == '}') { ++p; return v; }
        for (;;) {
            ws();
            if (p >= t.size() || t[p] != '"') fail("expected key");
            std::string k = string();
            ws();
            if (p >= t.size() || t[p] != ':') fail("expected :");
            ++p;
            v.obj.emplace_back(k, value());
            ws();
            if (p < t.size() && t[p] == ',') { ++p; continue; }
            if (p < t.size() && t[p] == '}') { ++p; break; }
            fail("expected , or }");
        }
        return v;
    }
};

}  // namespace rth
