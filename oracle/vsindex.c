/*
  TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle/vsoracle.h).

  Naive CPU construction of the tables mkvtree -dna -pl -allout writes, used
  to check the GPU index builder and to give the CPU-only tests an index when
  the reference binary is not around.  It states the DEFINITIONS of the
  tables, not the reference's algorithms (bucket sort + multikey quicksort,
  Mkvtree/ppsort.c:83, Mkvtree/bese.c:710):

    suf  suffixes of tis[0..n] (position n = end sentinel) in lexicographic
         order of the mapped symbols, where every special symbol (>= 254) and
         the sentinel is a symbol of its own, larger than all regular ones,
         specials ordered by text position (Mkvtree/bese.c:27-49,602)
    lcp  lcp[i] = number of leading regular symbols suf[i-1] and suf[i]
         share, capped at 255; larger values go to llv as (i, value)
         (Mkvtree/bese.c:533-566); lcp[0] = 0
    bck  per code c of prefixlength regular symbols the pair (left, mid):
         suf[left..mid) are the suffixes starting with that q-gram; suffixes
         cut short by a special symbol follow in [mid, next left)
         (Mkvtree/mkvprocess.c:251-327)
    bwt  tis[suf[i]-1], 253 where suf[i] = 0 (kurtz/bwtcode.c:293-311)
    sti1 sti1[suf[i]] = min(255, distance of i from the start of its run of
         lcp >= prefixlength) (Mkvtree/mkvprocess.c:583-612)

  Pinned against the reference's own files by tests/test_oracle_golden.py.
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
#include "vsoracle.h"

static const uint8_t *g_tis;
static uint64_t g_n;

static int suffix_compare(const void *pa, const void *pb)
{
  uint64_t i = *(const uint64_t *) pa, j = *(const uint64_t *) pb, k;

  if (i == j)
  {
    return 0;
  }
  for (k = 0;; k++)
  {
    int ispecial = (i + k >= g_n) || ORC_ISSPECIAL(g_tis[i + k]);
    int jspecial = (j + k >= g_n) || ORC_ISSPECIAL(g_tis[j + k]);

    if (ispecial || jspecial)
    {
      if (ispecial && jspecial)
      {
        return (i < j) ? -1 : 1; /* unique symbols, ordered by position */
      }
      return ispecial ? 1 : -1;
    }
    if (g_tis[i + k] != g_tis[j + k])
    {
      return (g_tis[i + k] < g_tis[j + k]) ? -1 : 1;
    }
  }
}

/* kurtz/detpfxlen.c:31-61 with sizeofbckentry = 2*sizeof(Uint) = 16
   (include/virtualdef.h:104) as in the reference's 64-bit build */
uint32_t orc_recommendedprefixlength(uint32_t numofchars, uint64_t totallength)
{
  double value = (double) totallength / 16.0;
  uint32_t pl;

  if (value <= (double) numofchars)
  {
    return 1;
  }
  pl = (uint32_t) floor(log(value) / log((double) numofchars));
  return pl == 0 ? 1 : pl;
}

/*
  All output arrays are caller-allocated: suf[n+1] (64 bit), lcp[n+1],
  llv[2*llvcap], bck[2*numofchars^pl], bwt[n+1], sti1[n+1]; any of bck, bwt,
  sti1 may be NULL.  Returns the number of llv pairs, or -1 if llvcap is too
  small.
*/
int64_t orc_build_tables(const uint8_t *tis, uint64_t n, uint32_t numofchars,
                         uint32_t pl, uint64_t *suf, uint8_t *lcp,
                         uint64_t *llv, uint64_t llvcap, uint64_t *bck,
                         uint8_t *bwt, uint8_t *sti1)
{
  uint64_t i, nllv = 0, numofcodes = 1, *rank, h;
  uint32_t k;

  for (i = 0; i <= n; i++)
  {
    suf[i] = i;
  }
  g_tis = tis;
  g_n = n;
  qsort(suf, (size_t) (n + 1), sizeof(uint64_t), suffix_compare);

  /* lcp by the inverse permutation (Kasai et al.), regular symbols only */
  rank = (uint64_t *) malloc((n + 1) * sizeof(uint64_t));
  if (rank == NULL)
  {
    return -2;
  }
  for (i = 0; i <= n; i++)
  {
    rank[suf[i]] = i;
  }
  lcp[0] = 0;
  h = 0;
  for (i = 0; i <= n; i++)
  {
    uint64_t r = rank[i], j;

    if (r == 0)
    {
      h = 0;
      continue;
    }
    j = suf[r - 1];
    while (i + h < n && j + h < n && !ORC_ISSPECIAL(tis[i + h]) &&
           tis[i + h] == tis[j + h])
    {
      h++;
    }
    if (h < 255)
    {
      lcp[r] = (uint8_t) h;
    } else
    {
      lcp[r] = 255;
    }
    rank[i] = h; /* reuse: value for position i, written out below in order */
    if (h > 0)
    {
      h--;
    }
  }
  /* exceptions in increasing index order */
  for (i = 1; i <= n; i++)
  {
    if (lcp[i] == 255)
    {
      if (nllv >= llvcap)
      {
        free(rank);
        return -1;
      }
      llv[2 * nllv] = i;
      llv[2 * nllv + 1] = rank[suf[i]];
      nllv++;
    }
  }
  free(rank);

  if (bwt != NULL)
  {
    for (i = 0; i <= n; i++)
    {
      bwt[i] = suf[i] > 0 ? tis[suf[i] - 1] : (uint8_t) ORC_UNDEFBWT;
    }
  }
  if (sti1 != NULL)
  {
    uint8_t cur = 0;

    sti1[suf[0]] = 0;
    for (i = 1; i <= n; i++)
    {
      if (lcp[i] < (uint8_t) pl)
      {
        cur = 0;
      } else if (cur < 255)
      {
        cur++;
      }
      sti1[suf[i]] = cur;
    }
  }
  if (bck != NULL)
  {
    uint64_t *full, *padded, c, acc;

    for (k = 0; k < pl; k++)
    {
      numofcodes *= numofchars;
    }
    full = (uint64_t *) calloc(numofcodes, sizeof(uint64_t));
    padded = (uint64_t *) calloc(numofcodes + 1, sizeof(uint64_t));
    if (full == NULL || padded == NULL)
    {
      return -2;
    }
    /* a suffix cut short by a special symbol sorts behind every suffix
       that continues with a regular symbol, i.e. where the code padded
       with the largest symbol sits */
    for (i = 0; i < n; i++)
    {
      int cut = 0;

      c = 0;
      for (k = 0; k < pl; k++)
      {
        if (!cut && (i + k >= n || ORC_ISSPECIAL(tis[i + k])))
        {
          cut = 1;
        }
        c = c * numofchars + (cut ? numofchars - 1 : tis[i + k]);
      }
      if (cut)
      {
        padded[c]++;
      } else
      {
        full[c]++;
      }
    }
    acc = 0;
    for (c = 0; c < numofcodes; c++)
    {
      bck[2 * c] = acc;
      bck[2 * c + 1] = acc + full[c];
      acc += full[c] + padded[c];
    }
    free(full);
    free(padded);
  }
  return (int64_t) nllv;
}
