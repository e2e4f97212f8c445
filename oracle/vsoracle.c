/*
  TEST INFRASTRUCTURE -- NOT PRODUCT CODE (see oracle/vsoracle.h).

  Plain C, single threaded, no GPU.  Build:
    gcc -O2 -shared -fPIC oracle/vsoracle.c -o oracle/liboracle.so
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "vsoracle.h"

static orc_counters cnt;

void orc_counters_get(orc_counters *c)
{
  *c = cnt;
}

void orc_counters_reset(void)
{
  memset(&cnt, 0, sizeof cnt);
}

void orc_matches_init(orc_matches *out)
{
  out->m = NULL;
  out->n = out->cap = 0;
}

void orc_matches_free(orc_matches *out)
{
  free(out->m);
  orc_matches_init(out);
}

static void push_match(orc_matches *out, uint64_t length, uint64_t dbstart,
                       uint64_t queryseq, uint64_t querystart)
{
  orc_match *m;

  if (out->n == out->cap)
  {
    out->cap = out->cap ? 2 * out->cap : 1024;
    out->m = (orc_match *) realloc(out->m, out->cap * sizeof(orc_match));
    if (out->m == NULL)
    {
      fprintf(stderr, "oracle: out of memory\n");
      exit(EXIT_FAILURE);
    }
  }
  m = out->m + out->n++;
  m->length = length;
  m->dbstart = dbstart;
  m->queryseq = queryseq;
  m->querystart = querystart;
  cnt.emitted++;
}

void orc_push_match(orc_matches *out, uint64_t length, uint64_t dbstart,
                    uint64_t queryseq, uint64_t querystart)
{
  push_match(out, length, dbstart, queryseq, querystart);
}

/* kurtz/cleanMUMcand.c:25-45: increasing dbstart, then decreasing length.
   Like the reference's comparator this one never reports equality. */
static int compare_mumcand(const void *pv, const void *qv)
{
  const orc_match *p = (const orc_match *) pv, *q = (const orc_match *) qv;

  if (p->dbstart == q->dbstart)
  {
    return (p->length < q->length) ? 1 : -1;
  }
  return (p->dbstart > q->dbstart) ? 1 : -1;
}

/* kurtz/cleanMUMcand.c:55-118: keep the candidates that are unique in the
   whole query set; survivors leave in dbstart order */
int orc_mumuniqueinquery(orc_match *cand, uint64_t ncand, orc_matches *out)
{
  return orc_mumuniqueinquery_carry(cand, ncand, 0, out);
}

/* the same loop entered with a given value of the running variable dbright:
   what the reference's loop does from the middle of a sorted list on */
int orc_mumuniqueinquery_carry(orc_match *cand, uint64_t ncand,
                               uint64_t carry, orc_matches *out)
{
  uint64_t i, dbright = carry, currentright;
  int ignorecurrent, ignoreprevious = 0;

  if (ncand == 0)
  {
    return 0;
  }
  qsort(cand, (size_t) ncand, sizeof(orc_match), compare_mumcand);
  for (i = 0; i < ncand; i++)
  {
    ignorecurrent = 0;
    currentright = cand[i].dbstart + cand[i].length - 1;
    if (dbright > currentright)
    {
      ignorecurrent = 1;
    } else if (dbright == currentright)
    {
      ignorecurrent = 1;
      /* the reference reads the element before the first one when i == 0;
         that value can only matter together with i > 0 below */
      if (!ignoreprevious && i > 0 && cand[i - 1].dbstart == cand[i].dbstart)
      {
        ignoreprevious = 1;
      }
    } else
    {
      dbright = currentright;
    }
    if (i > 0 && !ignoreprevious)
    {
      push_match(out, cand[i - 1].length, cand[i - 1].dbstart,
                 cand[i - 1].queryseq, cand[i - 1].querystart);
    }
    ignoreprevious = ignorecurrent;
  }
  if (!ignoreprevious)
  {
    push_match(out, cand[ncand - 1].length, cand[ncand - 1].dbstart,
               cand[ncand - 1].queryseq, cand[ncand - 1].querystart);
  }
  return 0;
}

/* Vmengine/exactcompl.c:277-325 */
static int findcompletematches_online(const orc_index *ix,
                                          const uint8_t *qbuf,
                                          const uint64_t *qstart,
                                          const uint64_t *qlen, uint64_t nq,
                                          orc_matches *out)
{
  uint64_t q, n = ix->n;
  const uint8_t *text = ix->tis;

  for (q = 0; q < nq; q++)
  {
    const uint8_t *pattern = qbuf + qstart[q];
    uint64_t plen = qlen[q], rmostocc[256], i, ppos, s;

    if (plen == 0 || plen > n)
    {
      continue;
    }
    for (i = 0; i < 256; i++)
    {
      rmostocc[i] = plen;
    }
    for (ppos = 0; ppos + 1 < plen; ppos++)
    {
      if (!ORC_ISSPECIAL(pattern[ppos]))
      {
        rmostocc[pattern[ppos]] = plen - ppos - 1;
      }
    }
    for (s = 0; s + plen <= n; s += rmostocc[text[s + plen - 1]])
    {
      for (i = plen - 1; !ORC_ISSPECIAL(text[s + i]) &&
                         pattern[i] == text[s + i]; i--)
      {
        if (i == 0)
        {
          push_match(out, plen, s, q, 0);
          break;
        }
      }
    }
  }
  return 0;
}

#define IDX uint32_t
#define FN(x) x##_32
#include "vsoracle_body.inc"
#undef IDX
#undef FN

#define IDX uint64_t
#define FN(x) x##_64
#include "vsoracle_body.inc"
#undef IDX
#undef FN

static int checkisize(const orc_index *idx, char *err)
{
  if (idx->isize != 4 && idx->isize != 8)
  {
    sprintf(err, "integersize=%u is not 32 or 64 bit", idx->isize * 8);
    return -1;
  }
  return 0;
}

int orc_findcompletematches(const orc_index *idx, const uint8_t *qbuf,
                            const uint64_t *qstart, const uint64_t *qlen,
                            uint64_t nq, orc_matches *out, char *err)
{
  if (checkisize(idx, err) != 0)
  {
    return -1;
  }
  return idx->isize == 4
           ? findcompletematches_32(idx, qbuf, qstart, qlen, nq, out, err)
           : findcompletematches_64(idx, qbuf, qstart, qlen, nq, out, err);
}

int orc_findcompletematches_online(const orc_index *idx, const uint8_t *qbuf,
                                   const uint64_t *qstart,
                                   const uint64_t *qlen, uint64_t nq,
                                   orc_matches *out, char *err)
{
  (void) err;
  return findcompletematches_online(idx, qbuf, qstart, qlen, nq, out);
}

int orc_findquerymatches(const orc_index *idx, const uint8_t *qbuf,
                         const uint64_t *qstart, const uint64_t *qlen,
                         uint64_t nq, int domum, int domumcand,
                         uint64_t searchlength, int speedup, orc_matches *out,
                         char *err)
{
  if (checkisize(idx, err) != 0)
  {
    return -1;
  }
  return idx->isize == 4
           ? findquerymatches_32(idx, qbuf, qstart, qlen, nq, domum,
                                 domumcand, searchlength, speedup, out, err)
           : findquerymatches_64(idx, qbuf, qstart, qlen, nq, domum,
                                 domumcand, searchlength, speedup, out, err);
}

int orc_findmaximaluniquematches(const orc_index *idx, uint64_t searchlength,
                                 orc_matches *out, char *err)
{
  if (checkisize(idx, err) != 0)
  {
    return -1;
  }
  return idx->isize == 4
           ? findmaximaluniquematches_32(idx, searchlength, out, err)
           : findmaximaluniquematches_64(idx, searchlength, out, err);
}
