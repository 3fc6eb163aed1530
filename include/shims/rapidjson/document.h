// rapidjson/document.h — a minimal DOM with the part of the rapidjson API that the ray tracer's
// callers use (reference cmd/main.cpp:161-175, rayverb/config.h, rayverb/rayverb.cpp:304-326).
// rapidjson itself is not available offline; this is an independent, header-only stand-in, NOT a
// copy: Parse, HasParseError/GetParseError, Is*/Get*, HasMember, operator[], Size, Begin/End,
// MemberBegin/MemberEnd.
#pragma once

#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace rapidjson {

enum ParseErrorCode {
    kParseErrorNone = 0,
    kParseErrorDocumentEmpty,
    kParseErrorDocumentRootNotSingular,
    kParseErrorValueInvalid,
    kParseErrorObjectMissName,
    kParseErrorObjectMissColon,
    kParseErrorObjectMissCommaOrCurlyBracket,
    kParseErrorArrayMissCommaOrSquareBracket,
    kParseErrorStringMissQuotationMark,
    kParseErrorTermination
};

class Value;
struct Member;

class Value {
public:
    enum Kind { kNull, kFalse, kTrue, kObject, kArray, kString, kNumber };
    typedef const Value * ConstValueIterator;
    typedef const Member * ConstMemberIterator;

    Value() : kind_(kNull), number_(0), integer_(false) {}

    bool IsNull() const { return kind_ == kNull; }
    bool IsBool() const { return kind_ == kTrue || kind_ == kFalse; }
    bool IsObject() const { return kind_ == kObject; }
    bool IsArray() const { return kind_ == kArray; }
    bool IsString() const { return kind_ == kString; }
    bool IsNumber() const { return kind_ == kNumber; }
    bool IsInt() const { return kind_ == kNumber && integer_ && number_ >= -2147483648.0 && number_ <= 2147483647.0; }
    bool IsDouble() const { return kind_ == kNumber && !integer_; }

    bool GetBool() const { return kind_ == kTrue; }
    int GetInt() const { return (int) number_; }
    double GetDouble() const { return number_; }
    const char * GetString() const { return string_.c_str(); }

    unsigned Size() const { return (unsigned) array_.size(); }
    const Value & operator[](int i) const { return array_[(size_t) i]; }
    const Value & operator[](unsigned i) const { return array_[i]; }
    ConstValueIterator Begin() const { return array_.data(); }
    ConstValueIterator End() const { return array_.data() + array_.size(); }

    bool HasMember(const char * name) const { return FindMemberValue(name) != nullptr; }
    const Value & operator[](const char * name) const
    {
        const Value * v = FindMemberValue(name);
        static const Value null_value;
        return v ? *v : null_value;
    }
    inline ConstMemberIterator MemberBegin() const;
    inline ConstMemberIterator MemberEnd() const;

protected:
    inline const Value * FindMemberValue(const char * name) const;
    friend class Document;
    Kind kind_;
    double number_;
    bool integer_;
    std::string string_;
    std::vector<Value> array_;
    std::vector<Member> members_;
};

struct Member {
    Value name;
    Value value;
};

inline Value::ConstMemberIterator Value::MemberBegin() const { return members_.data(); }
inline Value::ConstMemberIterator Value::MemberEnd() const { return members_.data() + members_.size(); }
inline const Value * Value::FindMemberValue(const char * name) const
{
    for (const Member & m : members_)
        if (m.name.string_ == name)
            return &m.value;
    return nullptr;
}

class Document : public Value {
public:
    Document() : error_(kParseErrorNone), offset_(0) {}

    Document & Parse(const char * text)
    {
        text_ = text;
        pos_ = 0;
        error_ = kParseErrorNone;
        *static_cast<Value *>(this) = Value();
        skip();
        if (pos_ >= std::strlen(text_)) return fail(kParseErrorDocumentEmpty);
        Value v;
        if (!parse(v)) return *this;
        skip();
        if (text_[pos_] != 0) return fail(kParseErrorDocumentRootNotSingular);
        *static_cast<Value *>(this) = v;
        return *this;
    }
    bool HasParseError() const { return error_ != kParseErrorNone; }
    ParseErrorCode GetParseError() const { return error_; }
    size_t GetErrorOffset() const { return offset_; }

private:
    const char * text_ = "";
    size_t pos_ = 0;
    ParseErrorCode error_;
    size_t offset_;

    Document & fail(ParseErrorCode e) { error_ = e; offset_ = pos_; *static_cast<Value *>(this) = Value(); return *this; }
    bool bad(ParseErrorCode e) { fail(e); return false; }
    void skip() { while (text_[pos_] == ' ' || text_[pos_] == '\t' || text_[pos_] == '\n' || text_[pos_] == '\r') ++pos_; }

    bool parse(Value & out)
    {
        skip();
        const char c = text_[pos_];
        if (c == '{') return parseObject(out);
        if (c == '[') return parseArray(out);
        if (c == '"') { out.kind_ = kString; return parseString(out.string_); }
        if (!std::strncmp(text_ + pos_, "true", 4)) { pos_ += 4; out.kind_ = kTrue; return true; }
        if (!std::strncmp(text_ + pos_, "false", 5)) { pos_ += 5; out.kind_ = kFalse; return true; }
        if (!std::strncmp(text_ + pos_, "null", 4)) { pos_ += 4; out.kind_ = kNull; return true; }
        if (c == 0) return bad(kParseErrorTermination);
        char * end = nullptr;
        const double d = std::strtod(text_ + pos_, &end);
        if (end == text_ + pos_) return bad(kParseErrorValueInvalid);
        bool integer = true;
        for (const char * p = text_ + pos_; p != end; ++p)
            if (*p == '.' || *p == 'e' || *p == 'E') integer = false;
        pos_ = (size_t) (end - text_);
        out.kind_ = kNumber;
        out.number_ = d;
        out.integer_ = integer;
        return true;
    }
    bool parseString(std::string & out)
    {
        ++pos_;
        out.clear();
        while (text_[pos_] && text_[pos_] != '"') {
            char c = text_[pos_++];
            if (c == '\\' && text_[pos_]) {
                const char e = text_[pos_++];
                c = e == 'n' ? '\n' : e == 't' ? '\t' : e == 'r' ? '\r' : e == 'b' ? '\b' : e == 'f' ? '\f' : e;
            }
            out += c;
        }
        if (text_[pos_] != '"') return bad(kParseErrorStringMissQuotationMark);
        ++pos_;
        return true;
    }
    bool parseArray(Value & out)
    {
        out.kind_ = kArray;
        ++pos_;
        skip();
        if (text_[pos_] == ']') { ++pos_; return true; }
        for (;;) {
            Value v;
            if (!parse(v)) return false;
            out.array_.push_back(v);
            skip();
            if (text_[pos_] == ',') { ++pos_; continue; }
            if (text_[pos_] == ']') { ++pos_; return true; }
            return bad(kParseErrorArrayMissCommaOrSquareBracket);
        }
    }
    bool parseObject(Value & out)
    {
        out.kind_ = kObject;
        ++pos_;
        skip();
        if (text_[pos_] == '}') { ++pos_; return true; }
        for (;;) {
            skip();
            if (text_[pos_] != '"') return bad(kParseErrorObjectMissName);
            Member m;
            m.name.kind_ = kString;
            if (!parseString(m.name.string_)) return false;
            skip();
            if (text_[pos_] != ':') return bad(kParseErrorObjectMissColon);
            ++pos_;
            if (!parse(m.value)) return false;
            out.members_.push_back(m);
            skip();
            if (text_[pos_] == ',') { ++pos_; continue; }
            if (text_[pos_] == '}') { ++pos_; return true; }
            return bad(kParseErrorObjectMissCommaOrCurlyBracket);
        }
    }
};

}  // namespace rapidjson
