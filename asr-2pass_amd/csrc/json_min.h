// Minimal JSON reader for the weight manifest (objects, arrays, numbers, strings, true/false/null).
// No dependency on the reference's vendored nlohmann/json (out of scope).
#pragma once
#include <cstdlib>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

namespace pfhip {

struct JValue {
  enum Kind { NUL, BOOL, NUM, STR, ARR, OBJ } kind = NUL;
  double num = 0;
  bool b = false;
  std::string str;
  std::vector<JValue> arr;
  std::map<std::string, JValue> obj;

  const JValue* get(const std::string& k) const {
    auto it = obj.find(k);
    return it == obj.end() ? nullptr : &it->second;
  }
  double number(const std::string& k, double dflt) const {
    const JValue* v = get(k);
    return (v && v->kind == NUM) ? v->num : dflt;
  }
};

class JParser {
 public:
  explicit JParser(const char* s) : p_(s) {}
  JValue parse() {
    JValue v = value();
    ws();
    if (*p_) throw std::runtime_error("json: trailing characters");
    return v;
  }

 private:
  const char* p_;
  void ws() { while (*p_ == ' ' || *p_ == '\n' || *p_ == '\t' || *p_ == '\r') ++p_; }
  JValue value() {
    ws();
    JValue v;
    if (*p_ == '{') {
      v.kind = JValue::OBJ;
      ++p_; ws();
      if (*p_ == '}') { ++p_; return v; }
      for (;;) {
        ws();
        std::string k = string();
        ws();
        if (*p_ != ':') throw std::runtime_error("json: expected ':'");
        ++p_;
        v.obj.emplace(std::move(k), value());
        ws();
        if (*p_ == ',') { ++p_; continue; }
        if (*p_ == '}') { ++p_; break; }
        throw std::runtime_error("json: expected ',' or '}'");
      }
    } else if (*p_ == '[') {
      v.kind = JValue::ARR;
      ++p_; ws();
      if (*p_ == ']') { ++p_; return v; }
      for (;;) {
        v.arr.push_back(value());
        ws();
        if (*p_ == ',') { ++p_; continue; }
        if (*p_ == ']') { ++p_; break; }
        throw std::runtime_error("json: expected ',' or ']'");
      }
    } else if (*p_ == '"') {
      v.kind = JValue::STR;
      v.str = string();
    } else if (*p_ == 't' || *p_ == 'f' || *p_ == 'n') {
      if (!strncmp_(p_, "true")) { v.kind = JValue::BOOL; v.b = true; p_ += 4; }
      else if (!strncmp_(p_, "false")) { v.kind = JValue::BOOL; v.b = false; p_ += 5; }
      else if (!strncmp_(p_, "null")) { p_ += 4; }
      else throw std::runtime_error("json: bad literal");
    } else {
      char* end = nullptr;
      v.kind = JValue::NUM;
      v.num = std::strtod(p_, &end);
      if (end == p_) throw std::runtime_error("json: bad number");
      p_ = end;
    }
    return v;
  }
  static int strncmp_(const char* a, const char* lit) {
    while (*lit) { if (*a++ != *lit++) return 1; }
    return 0;
  }
  std::string string() {
    if (*p_ != '"') throw std::runtime_error("json: expected string");
    ++p_;
    std::string s;
    while (*p_ && *p_ != '"') {
      if (*p_ == '\\') {
        ++p_;
        switch (*p_) {
          case 'n': s += '\n'; break;
          case 't': s += '\t'; break;
          case 'r': s += '\r'; break;
          case 'b': s += '\b'; break;
          case 'f': s += '\f'; break;
          case 'u': p_ += 4; s += '?'; break;   // manifest keys are ASCII
          default: s += *p_;
        }
        ++p_;
      } else {
        s += *p_++;
      }
    }
    if (*p_ != '"') throw std::runtime_error("json: unterminated string");
    ++p_;
    return s;
  }
};

}  // namespace pfhip
