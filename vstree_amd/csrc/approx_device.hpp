// Device-side dynamic programming for approximate complete matches
// (vmatch -complete -e K / -h K): bit-parallel columns after Myers, several
// 64-bit words per pattern, one work-item per text region or start position.
//
// Reference code these columns restate (values, not instruction sequences):
//   verifyedistlongmatch / verifyedistshortmatch  Vmengine/splitesaapm.c:44-196
//   long/medium/shortpatternlongestmatch          Vmengine/longestmatch.c:18-153
//   verifyhammingmatch                            Vmengine/splitesaapm.c:198-250
#ifndef VSA_APPROX_DEVICE_HPP
#define VSA_APPROX_DEVICE_HPP

#include "esa_device.hpp"

#define VSA_APM_MAXWORDS 8 // patterns up to 512 symbols (round 3; 4 words before)

// Column state of one pattern: the vertical +1 / -1 vectors of Myers'
// algorithm and the Eq masks of a 4-letter alphabet.  W words.
template <int W>
struct ApmColumn
{
  uint64_t peq[4][W]; // peq[c] bit i: pattern symbol i equals c
  uint64_t peqw[W];   // the same for the wildcard (254) of the TEXT
  uint64_t pv[W], mv[W];
  uint32_t m, score;
  uint64_t topbit; // bit of row m in the last word

  // reversed != 0: row i belongs to pattern symbol m-1-i (the verification
  // scans the text from right to left).  rawequal: bytes are compared as
  // they are, so that a wildcard in the pattern matches a wildcard in the
  // text (splitesaapm.c:72, patterns longer than 32); otherwise a wildcard
  // matches nothing (the Eq masks of kurtz-basic/getEqs.gen).
  __device__ __forceinline__ void init(const uint8_t *pattern, uint32_t len,
                                       bool reversed, bool rawequal)
  {
    m = len;
#pragma unroll
    for (int w = 0; w < W; w++)
    {
      peq[0][w] = peq[1][w] = peq[2][w] = peq[3][w] = peqw[w] = 0;
    }
    for (uint32_t i = 0; i < len; i++)
    {
      const uint8_t c = pattern[reversed ? len - 1 - i : i];
      const uint64_t bit = 1ull << (i & 63);
#pragma unroll
      for (int w = 0; w < W; w++)
      {
        if ((int) (i >> 6) == w)
        {
          peq[0][w] |= (c == 0) ? bit : 0;
          peq[1][w] |= (c == 1) ? bit : 0;
          peq[2][w] |= (c == 2) ? bit : 0;
          peq[3][w] |= (c == 3) ? bit : 0;
          peqw[w] |= (rawequal && c == VSA_WILDCARD) ? bit : 0;
        }
      }
    }
    topbit = 1ull << ((len - 1) & 63);
    reset();
  }

  // column 0: D[i] = i
  __device__ __forceinline__ void reset()
  {
#pragma unroll
    for (int w = 0; w < W; w++)
    {
      pv[w] = ~0ull;
      mv[w] = 0;
    }
    score = m;
  }

  // next column for text symbol c (not a separator).  HIN = the change of
  // row 0 from the previous column: 0 for "an occurrence may start
  // anywhere" (verification), +1 for a global alignment (longest match).
  template <int HIN>
  __device__ __forceinline__ void step(uint8_t c)
  {
    int hin = HIN;
    const int last = (int) ((m - 1) >> 6);
#pragma unroll
    for (int w = 0; w < W; w++)
    {
      if (w > last)
      {
        break;
      }
      uint64_t eq = (c < 4) ? ((c & 2) ? ((c & 1) ? peq[3][w] : peq[2][w])
                                       : ((c & 1) ? peq[1][w] : peq[0][w]))
                            : ((c == VSA_WILDCARD) ? peqw[w] : 0ull);
      const uint64_t p = pv[w], n = mv[w];
      const uint64_t xv = eq | n;
      if (hin < 0)
      {
        eq |= 1ull;
      }
      const uint64_t xh = (((eq & p) + p) ^ p) | eq;
      uint64_t ph = n | ~(xh | p);
      uint64_t mh = p & xh;
      const uint64_t top = (w == last) ? topbit : (1ull << 63);
      const int hout = (ph & top) ? 1 : ((mh & top) ? -1 : 0);
      ph <<= 1;
      mh <<= 1;
      if (hin < 0)
      {
        mh |= 1ull;
      } else if (hin > 0)
      {
        ph |= 1ull;
      }
      pv[w] = mh | ~(xv | ph);
      mv[w] = ph & xv;
      hin = hout;
    }
    score = (uint32_t) ((int) score + hin);
  }
};

#endif
