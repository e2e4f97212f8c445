/*
  TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see vsoracle.h).

  CPU restatement of the reference's approximate complete matching on the
  index: vmatch -complete -e K | -h K -q Q IDX (BASELINE.json configs[4]).

    findapproxcompletematchesindex   Vmengine/approxcompl.c:138-199
    splitesaapm / realsplitesaapm    Vmengine/splitesaapm.c:369-558
    getoptsplit                      Vmengine/splitesaapm.c:317-353
    storeapmposition (regions)       Vmengine/splitesaapm.c:268-302
    insertnewregion (merging)        kurtz/regionsmerger.c:318-357
    verifyedistlongmatch/shortmatch  Vmengine/splitesaapm.c:44-196
    verifyhammingmatch               Vmengine/splitesaapm.c:198-250
    edistprocessstartpos             Vmengine/approxcompl.c:14-66
    long/medium/shortpatternlongestmatch  Vmengine/longestmatch.c:18-153

  What the reference computes, per query P of length m with threshold k:
  the pattern is cut into splitsize pieces of length splitlen = m/splitsize
  (getoptsplit); each piece is searched in the index with threshold
  k/splitsize; every hit s of the piece at pattern offset o contributes the
  text region that an occurrence of P containing it can cover; the regions
  are kept in a red-black tree that merges overlapping and adjacent ones;
  the merged regions are visited in ascending order and each is scanned
  from its right end to its left end by a dynamic programming column
  (reversed pattern): every text position where an occurrence with at most k
  errors STARTS is reported, i.e. in descending order inside a region.  For
  edit distance the reported length is that of the best-distance, then
  longest prefix of the text behind the start position (longestmatch.c).

  Pieces with threshold 0 (k < splitsize: the pigeonhole case, every
  read-length configuration such as m=150,k=2 or m=100,k=2) are searched
  exactly; pieces with a threshold of their own (k >= m/10) and patterns that
  are not cut at all (splitsize 1: short patterns) go through esaapm /
  esahamming (esaapm.c:296-383, esahamming.c:86-164), restated from what they
  report (hamminghits, edithits).  Dynamic programming is done cell by cell
  here (the reference uses cut-off columns or bit vectors, which compute the
  same values).

  A match is (length, dbstart, queryseq, distance) with the distance in the
  querystart field (number of mismatches for Hamming; the reference stores
  the negative of it in Match.distance, approxcompl.c:78).
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "vsoracle.h"

#define DPWORDSIZE4 32u /* include/dpbitvec48.h:66 */
#define DPWORDSIZE8 64u

void orc_push_match(orc_matches *out, uint64_t length, uint64_t dbstart,
                    uint64_t queryseq, uint64_t querystart);

static uint64_t sufat(const orc_index *ix, uint64_t i)
{
  return ix->isize == 4 ? ((const uint32_t *) ix->suf)[i]
                        : ((const uint64_t *) ix->suf)[i];
}

/* Vmengine/splitesaapm.c:317-353 */
uint64_t orc_getoptsplit(int doedist, uint64_t spliterrorbound,
                         uint64_t numofchars, uint64_t textlen,
                         uint64_t patternlength, uint64_t threshold)
{
  uint64_t optsplit;

  if (threshold * spliterrorbound >= patternlength)
  {
    optsplit = threshold;
  } else
  {
    double logtextlen = log((double) textlen),
           lognumofchars = log((double) numofchars);
    if (doedist)
    {
      optsplit = (uint64_t) ((patternlength + threshold) /
                             (logtextlen / lognumofchars));
    } else
    {
      optsplit = (uint64_t) (patternlength / (logtextlen / lognumofchars));
    }
    if (optsplit > threshold + 1)
    {
      optsplit = threshold + 1;
    }
  }
  while (patternlength > DPWORDSIZE4 * optsplit)
  {
    optsplit++;
  }
  return optsplit;
}

/* piece against the suffix starting at text position s: <0 piece smaller,
   0 piece is a prefix, >0 piece larger.  Special symbols and the end of the
   text are larger than every regular symbol (kurtz/bese.c:27-49). */
static int cmppiece(const orc_index *ix, const uint8_t *p, uint64_t plen,
                    uint64_t s)
{
  uint64_t k;

  for (k = 0; k < plen; k++)
  {
    uint8_t t;
    if (s + k >= ix->n)
    {
      return -1;
    }
    t = ix->tis[s + k];
    if (ORC_ISSPECIAL(t))
    {
      return -1;
    }
    if (p[k] != t)
    {
      return p[k] < t ? -1 : 1;
    }
  }
  return 0;
}

/* suffix array interval of the exact occurrences of a piece without special
   symbols: what esaapm / esahamming report with threshold 0
   (esaapm.c:293-383, esahamming.c:82-164) */
static void pieceinterval(const orc_index *ix, const uint8_t *p,
                          uint64_t plen, uint64_t *first, uint64_t *behind)
{
  uint64_t lo = 0, hi = ix->n; /* suffixes 0..n-1 (suf[n] = n is the end) */

  while (lo < hi)
  {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (cmppiece(ix, p, plen, sufat(ix, mid)) > 0)
    {
      lo = mid + 1;
    } else
    {
      hi = mid;
    }
  }
  *first = lo;
  hi = ix->n;
  while (lo < hi)
  {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (cmppiece(ix, p, plen, sufat(ix, mid)) >= 0)
    {
      lo = mid + 1;
    } else
    {
      hi = mid;
    }
  }
  *behind = lo;
}

typedef struct
{
  uint64_t lo, hi;
} Region;

static int cmpregion(const void *a, const void *b)
{
  const Region *p = (const Region *) a, *q = (const Region *) b;
  return (p->lo > q->lo) - (p->lo < q->lo);
}

typedef struct
{
  Region *r;
  uint64_t n, cap;
} Regions;

static void pushregion(Regions *rs, uint64_t lo, uint64_t hi)
{
  if (rs->n == rs->cap)
  {
    rs->cap = rs->cap ? 2 * rs->cap : 64;
    rs->r = (Region *) realloc(rs->r, rs->cap * sizeof(Region));
    if (rs->r == NULL)
    {
      fprintf(stderr, "oracle: out of memory\n");
      exit(EXIT_FAILURE);
    }
  }
  rs->r[rs->n].lo = lo;
  rs->r[rs->n].hi = hi;
  rs->n++;
}

/* the state of the tree after all insertions: overlapping and adjacent
   regions are one (regionsmerger.c:196-213,228-232: uint1 + 1 >= uint0) */
static void mergeregions(Regions *rs)
{
  uint64_t i, k = 0;

  if (rs->n == 0)
  {
    return;
  }
  qsort(rs->r, rs->n, sizeof(Region), cmpregion);
  for (i = 1; i < rs->n; i++)
  {
    if (rs->r[k].hi + 1 >= rs->r[i].lo)
    {
      if (rs->r[i].hi > rs->r[k].hi)
      {
        rs->r[k].hi = rs->r[i].hi;
      }
    } else
    {
      rs->r[++k] = rs->r[i];
    }
  }
  rs->n = k + 1;
}

/* longestmatch.c:18-71 (and the bit-vector twins :83-153, which compute the
   same column): global alignment of all of u against prefixes of v; the
   smallest distance wins, among equals the longest prefix.  A wildcard in
   u matches nothing. */
static void longestmatch(const uint8_t *u, uint64_t ulen, const uint8_t *v,
                         uint64_t vlen, uint64_t *col, uint64_t *bestlen,
                         uint64_t *bestdist)
{
  uint64_t i, j;

  for (i = 0; i <= ulen; i++)
  {
    col[i] = i;
  }
  *bestlen = 0;
  *bestdist = ulen;
  for (j = 0; j < vlen; j++)
  {
    uint64_t nw = col[0];
    const uint8_t c = v[j];
    if (c == ORC_SEPARATOR)
    {
      return;
    }
    col[0]++;
    for (i = 1; i <= ulen; i++)
    {
      const uint64_t we = col[i];
      uint64_t val = (u[i - 1] != c || u[i - 1] == ORC_WILDCARD) ? nw + 1
                                                                  : nw;
      if (col[i - 1] + 1 < val)
      {
        val = col[i - 1] + 1;
      }
      if (we + 1 < val)
      {
        val = we + 1;
      }
      col[i] = val;
      nw = we;
    }
    if (*bestdist >= col[ulen])
    {
      *bestlen = j + 1;
      *bestdist = col[ulen];
    }
  }
}

/* splitesaapm.c:44-196: the region is read from right to left against the
   reversed pattern; the first row is 0 (an occurrence may start anywhere), a
   separator resets the column.  rawequal: patterns longer than 32 compare
   bytes (:72, wildcards match each other), shorter ones use the Eq masks of
   getEqsrev4 in which a wildcard of the pattern matches nothing. */
static void verifyedist(const orc_index *ix, const uint8_t *p, uint64_t m,
                        uint64_t k, const Region *reg, uint64_t q,
                        uint64_t *col, uint64_t *ecol, orc_matches *out)
{
  const int rawequal = m > DPWORDSIZE4;
  uint64_t width = reg->hi - reg->lo + 1, regionmaxlength = m + k, i;
  int64_t t;

  if (regionmaxlength > width)
  {
    regionmaxlength = width;
  }
  for (i = 0; i <= m; i++)
  {
    col[i] = i;
  }
  for (t = (int64_t) reg->hi; t >= (int64_t) reg->lo; t--)
  {
    const uint8_t c = ix->tis[t];
    if (c == ORC_SEPARATOR)
    {
      for (i = 0; i <= m; i++)
      {
        col[i] = i;
      }
      continue;
    }
    {
      uint64_t nw = 0; /* col[0] stays 0 */
      for (i = 1; i <= m; i++)
      {
        const uint8_t pc = p[m - i];
        const uint64_t we = col[i];
        const int eq = rawequal ? (pc == c) : (pc == c && pc != ORC_WILDCARD);
        uint64_t val = eq ? nw : nw + 1;
        const uint64_t up = (i == 1 ? 0 : col[i - 1]) + 1;
        if (up < val)
        {
          val = up;
        }
        if (we + 1 < val)
        {
          val = we + 1;
        }
        col[i] = val;
        nw = we;
      }
    }
    if (col[m] <= k)
    {
      /* edistprocessstartpos, approxcompl.c:14-66; the text behind the end
         reads as a separator */
      uint64_t vlen = regionmaxlength, len, dist;
      if ((uint64_t) t + vlen > ix->n)
      {
        vlen = ix->n - (uint64_t) t;
      }
      longestmatch(p, m, ix->tis + t, vlen, ecol, &len, &dist);
      orc_push_match(out, len, (uint64_t) t, q, dist);
    }
  }
}

/* splitesaapm.c:198-250 */
static void verifyhamming(const orc_index *ix, const uint8_t *p, uint64_t m,
                          uint64_t k, const Region *reg, uint64_t q,
                          orc_matches *out)
{
  int64_t t;

  for (t = (int64_t) reg->hi - (int64_t) m + 1; t >= (int64_t) reg->lo; t--)
  {
    if (ix->tis[t] == ORC_SEPARATOR)
    {
      t -= (int64_t) m; /* as the reference: m + 1 positions are passed */
    } else
    {
      uint64_t i, mm = 0;
      int skip = 0;
      for (i = 0; i < m; i++)
      {
        const uint8_t c = ix->tis[(uint64_t) t + i];
        if (c == ORC_SEPARATOR)
        {
          skip = 1;
          break;
        }
        if (c != p[i])
        {
          mm++;
          if (mm > k)
          {
            break;
          }
        }
      }
      if (!skip && mm <= k)
      {
        orc_push_match(out, m, (uint64_t) t, q, mm);
      }
    }
  }
}

/*
  esahamming (Vmengine/esahamming.c:86-164), restated from what it reports:
  it walks the suffixes 0 .. n-1 in suffix array order with a stack of
  mismatch counts per depth and the skip table; a suffix is reported iff its
  first plen symbols exist, hold no separator and differ from the pattern in
  at most `threshold` places (bytes are compared: a wildcard of the pattern
  equals a wildcard of the text, :66), with that number of mismatches.
  Suffixes that share more symbols with their predecessor than were looked
  at (lcp > dvalue+1) are skipped and inherit its verdict -- the same verdict
  a comparison of their own gives, since the symbols looked at are shared.
*/
typedef void (*Hitfunction)(void *info, uint64_t pos, uint64_t value);

static void hamminghits(const orc_index *ix, const uint8_t *p, uint64_t plen,
                        uint64_t threshold, Hitfunction report, void *info)
{
  uint64_t i;

  for (i = 0; i < ix->n; i++)
  {
    const uint64_t s = sufat(ix, i);
    uint64_t d, mm = 0;
    int ok = ix->n - s >= plen;

    for (d = 0; ok && d < plen; d++)
    {
      const uint8_t c = ix->tis[s + d];
      if (c == ORC_SEPARATOR || (c != p[d] && ++mm > threshold))
      {
        ok = 0;
      }
    }
    if (ok)
    {
      report(info, s, mm);
    }
  }
}

/*
  esaapm (Vmengine/esaapm.c:296-383): suffixes in suffix array order; for each
  the distance column of the pattern against the first d symbols of the
  suffix is advanced (nextEDcolumn :143-243, Eq masks of getEqs4 in which a
  wildcard matches nothing) for d = 1 .. maxlength = min(plen + threshold,
  symbols left in the text) until column d-1 has its last entry <= threshold
  (success, reported with `maxlength`), a separator is met, or no entry of the
  column is <= threshold any more.  dvalue = the depth the walk stopped at; a
  suffix whose lcp byte with its predecessor exceeds dvalue is skipped
  together with all that follow it while their lcp bytes exceed dvalue (the
  skip table jumps over them): they inherit the verdict AND are all reported
  with the maxlength of the FIRST skipped suffix (SETMAXLENGTH runs at the
  top of the loop, the macro APMSUCCESS reads that variable, :341-378) --
  restated as it stands.
*/
static void edithits(const orc_index *ix, const uint8_t *p, uint64_t plen,
                     uint64_t threshold, Hitfunction report, void *info)
{
  uint64_t i, dvalue = 0, maxlength = 0, col[2 * DPWORDSIZE4 + 2];
  int success = 0, skipping = 0;

  for (i = 0; i < ix->n; i++)
  {
    const uint64_t s = sufat(ix, i);
    const uint64_t vlen = ix->n - s;
    const int evaluate = i == 0 || dvalue >= (uint64_t) ix->lcp[i];

    if (evaluate || !skipping)
    {
      maxlength = plen + threshold;
      if (maxlength > vlen)
      {
        maxlength = vlen;
      }
    }
    skipping = !evaluate;
    if (evaluate)
    {
      uint64_t d, k;

      for (k = 0; k <= plen; k++)
      {
        col[k] = k;
      }
      success = 0;
      dvalue = maxlength;
      for (d = 1; d <= maxlength; d++)
      {
        uint64_t nw, alive = 0;
        const uint8_t c = ix->tis[s + d - 1];

        if (col[plen] <= threshold)
        {
          dvalue = d - 1;
          break;
        }
        if (c == ORC_SEPARATOR)
        {
          dvalue = d - 1;
          break;
        }
        nw = col[0];
        col[0] = d;
        for (k = 1; k <= plen; k++)
        {
          const uint64_t we = col[k];
          uint64_t val = (p[k - 1] == c && c != ORC_WILDCARD) ? nw : nw + 1;
          if (col[k - 1] + 1 < val)
          {
            val = col[k - 1] + 1;
          }
          if (we + 1 < val)
          {
            val = we + 1;
          }
          col[k] = val;
          nw = we;
        }
        for (k = 0; k <= plen; k++)
        {
          alive |= col[k] <= threshold;
        }
        if (!alive)
        {
          /* column d is dead: the walk stands at d-1, whose last entry is
             above the threshold (it was tested at the top of the round) */
          dvalue = d - 1;
          col[plen] = threshold + 1;
          break;
        }
      }
      success = col[plen] <= threshold;
    }
    if (success)
    {
      report(info, s, maxlength);
    }
  }
}

typedef struct
{
  const orc_index *ix;
  const uint8_t *p;
  uint64_t m, q, *ecol;
  orc_matches *out;
} Directinfo;

/* edistprocessstartpos, approxcompl.c:14-66 */
static void directedit(void *info, uint64_t pos, uint64_t maxlength)
{
  Directinfo *di = (Directinfo *) info;
  uint64_t len, dist;

  longestmatch(di->p, di->m, di->ix->tis + pos, maxlength, di->ecol, &len,
               &dist);
  orc_push_match(di->out, len, pos, di->q, dist);
}

/* hammingprocessstartpos, approxcompl.c:68-83 */
static void directhamming(void *info, uint64_t pos, uint64_t mm)
{
  Directinfo *di = (Directinfo *) info;

  orc_push_match(di->out, di->m, pos, di->q, mm);
}

typedef struct
{
  Regions *rs;
  uint64_t n, start, end;
} Regioninfo;

/* storeapmposition, splitesaapm.c:268-302 */
static void regionhit(void *info, uint64_t s, uint64_t unused)
{
  Regioninfo *ri = (Regioninfo *) info;
  const uint64_t lo = (ri->start > s) ? 0 : s - ri->start;
  const uint64_t hi = (ri->end + s - 1 < ri->n) ? ri->end + s - 1 : ri->n - 1;

  (void) unused;
  pushregion(ri->rs, lo, hi);
}

int orc_findcompletematches(const orc_index *idx, const uint8_t *qbuf,
                            const uint64_t *qstart, const uint64_t *qlen,
                            uint64_t nq, orc_matches *out, char *err);

/*
  vmatch -complete -e K / -h K (percent != 0: -e Kp / -h Kp, the threshold
  is m*K/100, initcompl.c:52-56).  Returns 0, -1 for the reference's errors
  (message in err), -4 if the configuration is not covered here.
*/
int orc_findapproxcompletematches(const orc_index *idx, const uint8_t *qbuf,
                                  const uint64_t *qstart,
                                  const uint64_t *qlen, uint64_t nq,
                                  int doedist, uint64_t distvalue,
                                  int percent, orc_matches *out, char *err)
{
  uint64_t q, maxm = 0, *col, *ecol;
  Regions rs;
  int rc = 0;

  memset(&rs, 0, sizeof rs);
  for (q = 0; q < nq; q++)
  {
    if (qlen[q] > maxm)
    {
      maxm = qlen[q];
    }
  }
  col = (uint64_t *) malloc((maxm + 2) * sizeof(uint64_t));
  ecol = (uint64_t *) malloc((maxm + 2) * sizeof(uint64_t));
  for (q = 0; q < nq && rc == 0; q++)
  {
    const uint8_t *p = qbuf + qstart[q];
    const uint64_t m = qlen[q];
    const uint64_t k = percent ? (m * distvalue) / 100 : distvalue;
    uint64_t splitsize, splitlen, poffset, i;

    if (k == 0)
    {
      /* approxcompl.c:167-176: the exact search, with its own error for
         patterns shorter than prefixlength */
      orc_matches one;
      const uint64_t zero = 0;
      uint64_t j;

      orc_matches_init(&one);
      if (orc_findcompletematches(idx, p, &zero, &m, 1, &one, err) != 0)
      {
        rc = -1;
      }
      for (j = 0; j < one.n; j++)
      {
        orc_push_match(out, one.m[j].length, one.m[j].dbstart, q, 0);
      }
      orc_matches_free(&one);
      continue;
    }
    if (k >= m)
    {
      /* splitesaapm.c:496-501 */
      snprintf(err, 256, "threshold=%lu>=%lu=patternlen not allowed",
               (unsigned long) k, (unsigned long) m);
      rc = -1;
      break;
    }
    splitsize = orc_getoptsplit(doedist, 10, idx->numofchars, idx->n, m, k);
    if (splitsize <= 1)
    {
      /* splitesaapm.c:523-543: the whole pattern goes through esaapm /
         esahamming, every reported suffix straight to the output function
         (edistprocessstartpos / hammingprocessstartpos,
         approxcompl.c:14-83): suffix array order */
      Directinfo di;
      di.ix = idx;
      di.p = p;
      di.m = m;
      di.q = q;
      di.ecol = ecol;
      di.out = out;
      if (doedist)
      {
        edithits(idx, p, m, k, directedit, &di);
      } else
      {
        hamminghits(idx, p, m, k, directhamming, &di);
      }
      continue;
    }
    splitlen = m / splitsize;
    rs.n = 0;
    if (k / splitsize != 0)
    {
      /* realsplitesaapm with a piece threshold: every piece through esaapm /
         esahamming, every reported suffix into the region tree
         (splitesaapm.c:400-425) */
      Regioninfo ri;
      ri.rs = &rs;
      ri.n = idx->n;
      for (poffset = 0; poffset < m - splitlen + 1; poffset += splitlen)
      {
        ri.start = doedist ? k + poffset : poffset;
        ri.end = doedist ? m + k - poffset : m - poffset;
        if (doedist)
        {
          edithits(idx, p + poffset, splitlen, k / splitsize, regionhit, &ri);
        } else
        {
          hamminghits(idx, p + poffset, splitlen, k / splitsize, regionhit,
                      &ri);
        }
      }
    } else
    /* realsplitesaapm, splitesaapm.c:378-432 */
    for (poffset = 0; poffset < m - splitlen + 1; poffset += splitlen)
    {
      uint64_t first, behind, start, end, j;
      int special = 0;

      if (doedist)
      {
        start = k + poffset;
        end = m + k - poffset;
      } else
      {
        start = poffset;
        end = m - poffset;
      }
      for (j = 0; j < splitlen; j++)
      {
        if (ORC_ISSPECIAL(p[poffset + j]))
        {
          special = 1; /* getEqs4: a wildcard matches nothing */
        }
      }
      if (special && doedist)
      {
        continue;
      }
      if (special)
      {
        /* esahamming compares bytes (esahamming.c:66): a wildcard of the
           piece matches a wildcard of the text.  Such suffixes are not
           contiguous in the suffix array; scan the text. */
        uint64_t s;
        for (s = 0; s + splitlen <= idx->n; s++)
        {
          if (memcmp(idx->tis + s, p + poffset, splitlen) == 0)
          {
            const uint64_t lo = (start > s) ? 0 : s - start;
            const uint64_t hi = (end + s - 1 < idx->n) ? end + s - 1
                                                       : idx->n - 1;
            pushregion(&rs, lo, hi);
          }
        }
        continue;
      }
      pieceinterval(idx, p + poffset, splitlen, &first, &behind);
      for (j = first; j < behind; j++)
      {
        /* storeapmposition, splitesaapm.c:268-302 */
        const uint64_t s = sufat(idx, j);
        const uint64_t lo = (start > s) ? 0 : s - start;
        const uint64_t hi = (end + s - 1 < idx->n) ? end + s - 1 : idx->n - 1;
        pushregion(&rs, lo, hi);
      }
    }
    mergeregions(&rs);
    for (i = 0; i < rs.n; i++)
    {
      if (doedist)
      {
        verifyedist(idx, p, m, k, rs.r + i, q, col, ecol, out);
      } else
      {
        verifyhamming(idx, p, m, k, rs.r + i, q, out);
      }
    }
  }
  free(col);
  free(ecol);
  free(rs.r);
  return rc;
}
