// rvb_json.h — a small JSON reader (objects, arrays, numbers, strings, booleans, null) for the
// material and config files of the ray tracer.  Host-side plumbing, not part of the hot path.
#pragma once

#include <cstdlib>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

namespace rvbjson {

struct Value {
    enum Kind { Null, Bool, Number, String, Array, Object } kind = Null;
    bool boolean = false;
    double number = 0.0;
    std::string string;
    std::vector<Value> array;
    std::vector<std::pair<std::string, Value>> object;     // file order kept

    bool isObject() const { return kind == Object; }
    bool isArray() const { return kind == Array; }
    bool isNumber() const { return kind == Number; }
    bool isBool() const { return kind == Bool; }
    bool isString() const { return kind == String; }
    const Value * find(const std::string & key) const
    {
        for (const auto & kv : object)
            if (kv.first == key)
                return &kv.second;
        return nullptr;
    }
};

class Parser {
public:
    explicit Parser(const std::string & text) : s(text), i(0) {}
    Value parse()
    {
        Value v = value();
        ws();
        if (i != s.size())
            fail("trailing characters");
        return v;
    }

private:
    const std::string & s;
    size_t i;

    [[noreturn]] void fail(const char * what) const
    {
        throw std::runtime_error(std::string("JSON parse error at offset ") + std::to_string(i) + ": " + what);
    }
    void ws() { while (i < s.size() && (s[i] == ' ' || s[i] == '\t' || s[i] == '\n' || s[i] == '\r')) ++i; }
    bool eat(char c) { ws(); if (i < s.size() && s[i] == c) { ++i; return true; } return false; }

    Value value()
    {
        ws();
        if (i >= s.size()) fail("unexpected end");
        const char c = s[i];
        if (c == '{') return object();
        if (c == '[') return array();
        if (c == '"') { Value v; v.kind = Value::String; v.string = str(); return v; }
        if (s.compare(i, 4, "true") == 0) { i += 4; Value v; v.kind = Value::Bool; v.boolean = true; return v; }
        if (s.compare(i, 5, "false") == 0) { i += 5; Value v; v.kind = Value::Bool; return v; }
        if (s.compare(i, 4, "null") == 0) { i += 4; return Value(); }
        return number();
    }
    Value number()
    {
        const char * begin = s.c_str() + i;
        char * end = nullptr;
        const double d = std::strtod(begin, &end);
        if (end == begin) fail("invalid value");
        i += (size_t) (end - begin);
        Value v;
        v.kind = Value::Number;
        v.number = d;
        return v;
    }
    std::string str()
    {
        std::string out;
        ++i;                                        // opening quote
        while (i < s.size() && s[i] != '"') {
            char c = s[i++];
            if (c == '\\') {
                if (i >= s.size()) fail("bad escape");
                const char e = s[i++];
                switch (e) {
                case 'n': c = '\n'; break;
                case 't': c = '\t'; break;
                case 'r': c = '\r'; break;
                case 'b': c = '\b'; break;
                case 'f': c = '\f'; break;
                case 'u': {                         // kept as the raw escape: names here are ASCII
                    out += "\\u";
                    continue;
                }
                default: c = e;
                }
            }
            out += c;
        }
        if (i >= s.size()) fail("unterminated string");
        ++i;
        return out;
    }
    Value array()
    {
        Value v;
        v.kind = Value::Array;
        ++i;
        if (eat(']')) return v;
        for (;;) {
            v.array.push_back(value());
            if (eat(']')) return v;
            if (!eat(',')) fail("expected ',' or ']'");
        }
    }
    Value object()
    {
        Value v;
        v.kind = Value::Object;
        ++i;
        if (eat('}')) return v;
        for (;;) {
            ws();
            if (i >= s.size() || s[i] != '"') fail("expected a member name");
            std::string key = str();
            if (!eat(':')) fail("expected ':'");
            v.object.emplace_back(key, value());
            if (eat('}')) return v;
            if (!eat(',')) fail("expected ',' or '}'");
        }
    }
};

inline Value parse(const std::string & text) { return Parser(text).parse(); }

}  // namespace rvbjson
