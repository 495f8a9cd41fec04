// "%.8f" without printf: the writers of gamma.txt / theta.txt / beta.txt (save_gamma, save_beta:
// src/snpsamplinge.cc:546-576, :761-798) print 2 N K + L K doubles with fprintf("%.8f\t") -- 7.35 s per report at
// N = 1M, K = 8 with glibc, as long as the training of a whole report period on the GPU.
//
// fmt_fixed8 produces the SAME BYTES as snprintf(buf, n, "%.8f", v) in the default rounding mode, exactly, not
// approximately: a double is m * 2^e with a 53-bit integer m, so v * 10^8 = (m * 10^8) * 2^e is an integer of at most 80
// bits shifted by e -- formed in unsigned __int128 without any rounding, then rounded half-to-even on its exact
// remainder, which is what glibc's exact decimal conversion does (a tie exists only for dyadic values such as 1/512 =
// 0.001953125 -> "0.00195312").  |v| >= 2^63, infinities and NaNs take snprintf itself.
// tests/test_host_format_cpu.py compares 10^7 random and edge-case doubles with snprintf byte for byte.
#pragma once
#include <cstdint>
#include <cstdio>
#include <cstring>

namespace tsfmt {

// two digits at a time
static const char kDigits2[201] =
    "00010203040506070809101112131415161718192021222324252627282930313233343536373839404142434445464748495051525354555657585960616263646566676869"
    "707172737475767778798081828384858687888990919293949596979899";

// writes "%.8f" of v at p (no terminating NUL), returns the end; at most 32 bytes for |v| < 2^63, what snprintf
// needs otherwise (kMaxLen covers every double)
constexpr int kMaxLen = 336;
inline char *fmt_fixed8(char *p, double v) {
  uint64_t bits;
  memcpy(&bits, &v, 8);
  const bool neg = (bits >> 63) != 0;
  const int bexp = (int)((bits >> 52) & 0x7ff);
  uint64_t m = bits & ((1ull << 52) - 1);
  if (bexp == 0x7ff || bexp >= 1023 + 63) {  // inf, nan, |v| >= 2^63: glibc does it
    return p + snprintf(p, kMaxLen, "%.8f", v);
  }
  int e;  // v = m * 2^e
  if (bexp == 0) {
    e = -1074;
  } else {
    m |= 1ull << 52;
    e = bexp - 1075;
  }
  unsigned __int128 q;  // round-half-even(|v| * 10^8)
  const unsigned __int128 P = (unsigned __int128)m * 100000000ull;  // < 2^80
  if (e >= 0) {
    q = P << e;  // (e <= 10 here: bexp < 1023 + 63)
  } else {
    const int s = -e;
    if (s >= 128) {
      q = 0;  // P < 2^80 <= half of 2^s: rounds to zero (a non-zero remainder below one half, or exactly zero)
    } else {
      q = P >> s;
      const unsigned __int128 rem = P & ((((unsigned __int128)1) << s) - 1), half = ((unsigned __int128)1) << (s - 1);
      if (rem > half || (rem == half && (q & 1))) q += 1;
    }
  }
  const uint64_t ip = (uint64_t)(q / 100000000ull);  // < 2^63
  uint32_t fp = (uint32_t)(q % 100000000ull);
  if (neg) *p++ = '-';
  // integer part
  char tmp[24];
  int nd = 0;
  uint64_t x = ip;
  if (x == 0) {
    tmp[nd++] = '0';
  } else {
    while (x >= 100) {
      const unsigned r = (unsigned)(x % 100);
      x /= 100;
      tmp[nd++] = kDigits2[2 * r + 1];
      tmp[nd++] = kDigits2[2 * r];
    }
    if (x >= 10) {
      tmp[nd++] = kDigits2[2 * x + 1];
      tmp[nd++] = kDigits2[2 * x];
    } else {
      tmp[nd++] = (char)('0' + x);
    }
  }
  while (nd) *p++ = tmp[--nd];
  *p++ = '.';
  // eight fraction digits
  const unsigned a = fp / 1000000u;
  fp -= a * 1000000u;
  const unsigned b = fp / 10000u;
  fp -= b * 10000u;
  const unsigned c = fp / 100u, d = fp - c * 100u;
  memcpy(p, kDigits2 + 2 * a, 2);
  memcpy(p + 2, kDigits2 + 2 * b, 2);
  memcpy(p + 4, kDigits2 + 2 * c, 2);
  memcpy(p + 6, kDigits2 + 2 * d, 2);
  return p + 8;
}

// "%d" of a non-negative int (the location column of beta.txt)
inline char *fmt_uint(char *p, uint32_t x) {
  char tmp[12];
  int nd = 0;
  do {
    tmp[nd++] = (char)('0' + x % 10u);
    x /= 10u;
  } while (x);
  while (nd) *p++ = tmp[--nd];
  return p;
}

}  // namespace tsfmt
