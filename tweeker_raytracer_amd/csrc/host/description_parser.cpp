#include "description_parser.h"

#include <cctype>
#include <fstream>
#include <sstream>

namespace twk {

static inline bool isBlank(char c)     { return c == ' ' || c == '\t'; }
static inline bool isDelimiter(char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\n'; }
static inline bool isValueChar(char c) { return (c >= '0' && c <= '9') || c == '+' || c == '-' || c == '.' || c == 'e' || c == 'E'; }

bool DescriptionParser::loadFile(const std::string& filename)
{
  std::ifstream in(filename, std::ios::binary);
  if (!in) return false;
  std::stringstream ss;
  ss << in.rdbuf();
  if (in.fail()) return false;
  loadString(ss.str());
  return true;
}

TokenType DescriptionParser::nextToken(std::string& token)
{
  token.clear();
  const size_t n = m_text.size();
  for (;;)
  {
    while (m_pos < n && isBlank(m_text[m_pos])) ++m_pos;
    if (m_pos >= n) return TOKEN_EOF;

    const char c = m_text[m_pos];
    if (c == '#')
    {
      // comment: skip to and past the next linefeed
      while (m_pos < n && m_text[m_pos] != '\n') ++m_pos;
      if (m_pos >= n) return TOKEN_EOF;
      ++m_pos;
      ++m_line;
    }
    else if (c == '\r') { ++m_pos; }
    else if (c == '\n') { ++m_pos; ++m_line; }
    else
    {
      const size_t first = m_pos;
      while (m_pos < n && !isDelimiter(m_text[m_pos])) ++m_pos;
      token.assign(m_text, first, m_pos - first);
      if (std::isdigit(static_cast<unsigned char>(c)) || c == '-' || c == '+' || c == '.')
      {
        bool allValue = true;
        for (char t : token) if (!isValueChar(t)) { allValue = false; break; }
        if (allValue) return TOKEN_VAL;
      }
      return TOKEN_ID;
    }
  }
}

TokenType DescriptionParser::restOfLine(std::string& token)
{
  token.clear();
  const size_t n = m_text.size();
  while (m_pos < n && isBlank(m_text[m_pos])) ++m_pos;
  if (m_pos >= n) return TOKEN_EOF;

  const char c = m_text[m_pos];
  if (c == '\r') { ++m_pos; return TOKEN_EOL; }
  if (c == '\n') { ++m_pos; ++m_line; return TOKEN_EOL; }

  const size_t first = m_pos;
  while (m_pos < n && m_text[m_pos] != '\r' && m_text[m_pos] != '\n') ++m_pos;
  size_t last = m_pos;
  while (first < last && isDelimiter(m_text[last - 1])) --last;
  if (first == last) return TOKEN_EOL;
  token.assign(m_text, first, last - first);
  return TOKEN_ID;
}

} // namespace twk
