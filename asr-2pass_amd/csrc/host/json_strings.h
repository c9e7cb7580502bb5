// Minimal reader for JSON arrays of strings (tokens.json, punc_list): escapes incl. \uXXXX and surrogate pairs -> UTF-8.
#pragma once
#include <string>
#include <vector>

namespace pfhip_host {

inline void AppendUtf8(std::string& s, unsigned cp) {
  if (cp < 0x80) s += (char)cp;
  else if (cp < 0x800) { s += (char)(0xC0 | (cp >> 6)); s += (char)(0x80 | (cp & 0x3F)); }
  else if (cp < 0x10000) { s += (char)(0xE0 | (cp >> 12)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
  else { s += (char)(0xF0 | (cp >> 18)); s += (char)(0x80 | ((cp >> 12) & 0x3F)); s += (char)(0x80 | ((cp >> 6) & 0x3F)); s += (char)(0x80 | (cp & 0x3F)); }
}

inline bool Hex4(const std::string& s, size_t i, unsigned& v) {
  if (i + 4 > s.size()) return false;
  v = 0;
  for (int k = 0; k < 4; ++k) {
    const char c = s[i + k];
    v <<= 4;
    if (c >= '0' && c <= '9') v |= (unsigned)(c - '0');
    else if (c >= 'a' && c <= 'f') v |= (unsigned)(c - 'a' + 10);
    else if (c >= 'A' && c <= 'F') v |= (unsigned)(c - 'A' + 10);
    else return false;
  }
  return true;
}

// s[i] == '"': reads the string, leaves i after the closing quote
inline bool ReadJsonString(const std::string& s, size_t& i, std::string& out) {
  out.clear();
  if (i >= s.size() || s[i] != '"') return false;
  for (++i; i < s.size(); ++i) {
    const char c = s[i];
    if (c == '"') { ++i; return true; }
    if (c != '\\') { out += c; continue; }
    if (++i >= s.size()) return false;
    switch (s[i]) {
      case 'n': out += '\n'; break;
      case 't': out += '\t'; break;
      case 'r': out += '\r'; break;
      case 'b': out += '\b'; break;
      case 'f': out += '\f'; break;
      case 'u': {
        unsigned cp = 0, lo = 0;
        if (!Hex4(s, i + 1, cp)) return false;
        i += 4;
        if (cp >= 0xD800 && cp < 0xDC00 && i + 6 < s.size() && s[i + 1] == '\\' && s[i + 2] == 'u' && Hex4(s, i + 3, lo) &&
            lo >= 0xDC00 && lo < 0xE000) {
          cp = 0x10000 + ((cp - 0xD800) << 10) + (lo - 0xDC00);
          i += 6;
        }
        AppendUtf8(out, cp);
        break;
      }
      default: out += s[i];                    // \" \\ \/
    }
  }
  return false;
}

// the array starting at the first '[' at or after `from`
inline bool ReadJsonStringArray(const std::string& s, size_t from, std::vector<std::string>& out) {
  out.clear();
  size_t i = s.find('[', from);
  if (i == std::string::npos) return false;
  for (++i; i < s.size();) {
    const char c = s[i];
    if (c == ']') return true;
    if (c == '"') {
      std::string t;
      if (!ReadJsonString(s, i, t)) return false;
      out.push_back(t);
    } else {
      ++i;                                      // whitespace, commas
    }
  }
  return false;
}

}  // namespace pfhip_host
