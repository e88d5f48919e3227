// Host-side TSV row parser of the input pipeline (no device code in this file).
//
// Replaces, per field, the reference's csv.reader + int() / int(v, 16) + fmod arithmetic (nasrec/torchrec/utils.py:175-193,
// nasrec/torchrec/criteo.py:45-58, nasrec/utils/data_pipes.py:137-175), which tops out at ~26 k rows/s per Python process
// while the engine consumes ~400 k rows/s.  Exactness contract: a line is parsed here only if every field is in the
// plain subset on which Python's parsers and this code agree by construction — integers: optional sign + decimal
// digits (anything else that int() rejects, including the empty string, is 0 exactly like safe_cast; characters that
// int() treats specially — whitespace, '_', or a '"' anywhere in the line, which the csv module would interpret — send
// the line back to the caller); categorical ids: empty (= missing, -1) or 1..15 hexadecimal digits.  The caller
// (nasrec_amd/utils/data_pipes.py) runs the handful of returned lines through the Python path, so results are identical
// to the reference for every input.
#include <stdint.h>
#include <string.h>

#include "common.h"

namespace {

inline bool int_special_char(unsigned char c) {
  // int() accepts these in places (whitespace, digit-group underscores, non-ASCII digits); the csv module interprets '"'
  return c == ' ' || c == '_' || c == '\r' || c == '\f' || c == '\v' || c == '"' || c >= 0x80;
}

}  // namespace

extern "C" int64_t nasrec_tsv_parse(const char* buf, int64_t len, int32_t Fd, int32_t Fs, const int64_t* table_rows, int64_t max_rows,
                                    int64_t* label, int64_t* dense, int64_t* cat, int64_t* consumed, int32_t* status) {
  const int ncol = 1 + Fd + Fs;
  const char* const bend = buf + len;
  const char* p = buf;
  int64_t rows = 0;
  *status = NASREC_TSV_OK;
  while (rows < max_rows && p < bend) {
    const char* const line = p;
    // the line must be complete: find its end first (one memchr per line; the fields are then parsed in a single pass)
    const char* nl = (const char*)memchr(line, '\n', (size_t)(bend - line));
    if (nl == nullptr) break;  // incomplete last line: the caller carries it over
    const char* end = nl;
    if (end > line && end[-1] == '\r') --end;
    int64_t* drow = dense + rows * Fd;
    int64_t* crow = cat + rows * Fs;
    bool special = false;
    int col = 0;
    const char* q = line;
    for (;;) {  // one field per iteration; q at its first character
      if (col <= Fd) {  // label or integer column: int() semantics with 0 for anything it rejects
        int64_t v = 0;
        bool neg = false, junk = false;
        const char* f0 = q;
        if (q < end && (*q == '+' || *q == '-')) {
          neg = (*q == '-');
          ++q;
        }
        const char* d0 = q;
        while (q < end && *q != '\t') {
          const unsigned d = (unsigned)(*q - '0');
          if (d <= 9u) {
            v = v * 10 + (int64_t)d;
          } else {
            junk = true;
            if (int_special_char((unsigned char)*q)) special = true;
          }
          ++q;
        }
        if (q - d0 > 18) special = true;          // may not fit int64: Python decides
        if (junk || q == d0 || q == f0) v = 0;    // rejected by int() (empty, sign only, letters ...) -> safe_cast default
        else if (neg) v = -v;
        if (col < ncol) {
          if (col == 0) label[rows] = v;
          else drow[col - 1] = v;
        }
      } else {  // categorical column: empty = missing (-1), else hexadecimal digits
        int64_t v = 0;
        const char* f0 = q;
        while (q < end && *q != '\t') {
          const char c = *q;
          unsigned d;
          if (c >= '0' && c <= '9') d = (unsigned)(c - '0');
          else if (c >= 'a' && c <= 'f') d = (unsigned)(c - 'a' + 10);
          else if (c >= 'A' && c <= 'F') d = (unsigned)(c - 'A' + 10);
          else {
            special = true;  // int(v, 16) raises or applies prefix/underscore rules: the Python path reproduces either
            d = 0;
          }
          v = (v << 4) | (int64_t)d;
          ++q;
        }
        if (q - f0 > 15) special = true;
        if (col < ncol) {
          const int j = col - 1 - Fd;
          const int64_t mod = table_rows[j] - 1;
          if (mod <= 0) {
            special = true;  // torch.fmod by zero raises: leave it to the Python path
          } else {
            // torch.fmod: the sign follows the dividend; a missing id (-1) maps to -(1 % mod) + 1
            crow[j] = (q == f0) ? (1 - (1 % mod)) : (v % mod + 1);
          }
        }
      }
      ++col;
      if (q >= end) break;
      ++q;  // skip the tab
      if (col > ncol) {  // too many columns: no need to parse further
        col = ncol + 1;
        break;
      }
    }
    if (special) {
      *status = NASREC_TSV_NEEDS_PYTHON;
      break;
    }
    if (col != ncol) {
      *status = NASREC_TSV_BAD_COLUMNS;
      break;
    }
    ++rows;
    p = nl + 1;
  }
  *consumed = p - buf;
  return rows;
}
