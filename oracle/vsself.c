/*
  TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see vsoracle.h).

  CPU restatement of the reference's supermaximal repeats,
  vmatch -supermax -l L IDX:
    findsupermax / selectsupermaxialrepeats / verifysupermaximality
                                         Vmengine/fsuper.c:60-165
    the depth first traversal            include/vdfstrav.c:247-420
    processexactselfmatch (ACCEPTMATCH)  Vmengine/fself.c:21-38,95-125

  The traversal completes a node [left, right] of depth d with "alwaysontop"
  exactly when all its children are leaves: lcp[left] < d, lcp[left+1 ..
  right] == d, lcp[right+1] < d (vdfstrav.c:330-397 with the edge macros of
  fsuper.c:11-19).  Such a node is reported if d >= L and the characters to
  the left of its suffixes are pairwise different, special symbols not
  counted, the suffix at text position 0 counted once (fsuper.c:60-103); then
  every pair of its suffixes is a match (fsuper.c:105-127), smaller start
  first (ACCEPTMATCH).  Nodes come in suffix array order.
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vsoracle.h"

void orc_push_match(orc_matches *out, uint64_t length, uint64_t dbstart,
                    uint64_t queryseq, uint64_t querystart);

static uint64_t sufat(const orc_index *ix, uint64_t i)
{
  return ix->isize == 4 ? ((const uint32_t *) ix->suf)[i]
                        : ((const uint64_t *) ix->suf)[i];
}

/* lcp value with the exceptions >= 255 from llv (virtualdef.h:121-136) */
static uint64_t lcpat(const orc_index *ix, uint64_t i)
{
  uint64_t lo = 0, hi = ix->nllv;

  if (ix->lcp[i] < 255)
  {
    return ix->lcp[i];
  }
  while (lo < hi)
  {
    const uint64_t mid = lo + (hi - lo) / 2;
    const uint64_t key = ix->isize == 4
                             ? ((const uint32_t *) ix->llv)[2 * mid]
                             : ((const uint64_t *) ix->llv)[2 * mid];
    if (key < i)
    {
      lo = mid + 1;
    } else
    {
      hi = mid;
    }
  }
  return ix->isize == 4 ? ((const uint32_t *) ix->llv)[2 * lo + 1]
                        : ((const uint64_t *) ix->llv)[2 * lo + 1];
}

int orc_findsupermax(const orc_index *ix, uint64_t searchlength,
                     orc_matches *out, char *err)
{
  uint64_t c;

  if (ix->bwt == NULL)
  {
    snprintf(err, 256, "table bwt is not loaded");
    return -1;
  }
  if (ix->hasqueries)
  {
    /* fself.c:193-198 */
    snprintf(err, 256, "supermaximal repeat search does not allow query "
             "files in index");
    return -1;
  }
  if (ix->n < 2)
  {
    snprintf(err, 256, "repeat search requires a sequence of length >= 2");
    return -1;
  }
  /* leaves 0 .. n; lcp[i] belongs to leaves i-1 and i */
  c = 0;
  while (c + 1 <= ix->n)
  {
    const uint64_t d = lcpat(ix, c + 1);
    uint64_t r, q, s, t;
    int marktab[256], marksep = 0, ok = 1;

    if (d < searchlength || d == 0 || (c > 0 && lcpat(ix, c) >= d))
    {
      c++;
      continue;
    }
    r = c + 1;
    while (r + 1 <= ix->n && lcpat(ix, r + 1) == d)
    {
      r++;
    }
    if (r + 1 <= ix->n && lcpat(ix, r + 1) > d)
    {
      c = r; /* a deeper node follows: not all children are leaves */
      continue;
    }
    memset(marktab, 0, sizeof marktab);
    for (q = c; q <= r && ok; q++)
    {
      if (sufat(ix, q) == 0) /* q == longest */
      {
        if (marksep)
        {
          ok = 0;
        }
        marksep = 1;
      } else
      {
        const uint8_t cc = ix->bwt[q];
        if (!ORC_ISSPECIAL(cc))
        {
          if (marktab[cc])
          {
            ok = 0;
          }
          marktab[cc] = 1;
        }
      }
    }
    if (ok)
    {
      for (s = c; s < r; s++)
      {
        for (t = s + 1; t <= r; t++)
        {
          const uint64_t i = sufat(ix, s), j = sufat(ix, t);
          orc_push_match(out, d, i < j ? i : j, i < j ? j : i, 0);
        }
      }
    }
    c = r;
  }
  return 0;
}
