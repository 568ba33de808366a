// Tokeniser of rtigo3's system/scene description files.
// Behaviour follows reference apps/rtigo3/src/Parser.cpp:72-148 (getNextToken) and :152-226 (getNextLine):
// tokens are separated by space/tab/CR/LF, '#' starts a comment to the end of the line, a token that
// starts with a digit, '+', '-' or '.' and consists only of "+-0123456789.eE" is a value, anything else
// an identifier. getNextLine returns the rest of the line (paths with blanks), trailing blanks pruned.
#pragma once
#include <string>

namespace twk {

enum TokenType
{
  TOKEN_UNKNOWN = 0, // error
  TOKEN_ID      = 1, // keyword, identifier, filename
  TOKEN_VAL     = 2, // number
  TOKEN_EOL     = 3,
  TOKEN_EOF     = 4
};

class DescriptionParser
{
public:
  bool loadFile(const std::string& filename);
  void loadString(const std::string& text) { m_text = text; m_pos = 0; m_line = 1; }

  TokenType nextToken(std::string& token);
  TokenType restOfLine(std::string& token);
  unsigned int line() const { return m_line; }

private:
  std::string  m_text;
  size_t       m_pos  = 0;
  unsigned int m_line = 1;
};

} // namespace twk
