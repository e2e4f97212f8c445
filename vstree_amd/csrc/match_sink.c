/*
  The host match sink: from the engine's match records to the lines vmatch
  prints (host side, plain C).

  Restates, for the matches of this path and the default output options,
    processfinal        Vmatch/procfinal.c:515-637
      fetchpositions    Vmatch/procfinal.c:72-176   (sequence number and
                        relative position: getseqinfo / findboundaries,
                        kurtz-basic/multiseq-adv.c:277-340; the flip of
                        relpos2 for palindromic matches :152-158)
      convertthematch   Vmatch/procfinal.c:408-497
      assignEvalue      Vmatch/procfinal.c:195-257 with the table of
                        kurtz/evalues.c:307-420 (inithammingEvalues,
                        incprecomputehammingEvalues, incgetEvalue), built by
                        the same sequence of double operations
      matchokay         Vmatch/mokay.c:7-27 (least length)
    vmatchnormaloutmatch Vmatch/echomatch.c:878-987 with echomatchpart1/2
                        :104-228, the column widths of Vmatch/assigndig.c:6-46,
                        score and identity of include/match.h:114-138
  Line: len1 seq1 pos1 D|P len2 seq2 pos2 dist evalue score identity.

  tests/test_sink.py formats the golden match lists and demands the md5 of the
  lines vmatch itself printed (tests/golden/manifest.json: md5_lines).
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <stdint.h>
#include <math.h>
#include <pthread.h>
#include <unistd.h>
#include "vstree_amd.h"

char *vsa_errbuf(void);
#define ERRSIZE 1024

#define SMALLESTEVALUE 1.0e-300 /* include/evaluedef.h:26 */
#define MAXEXPONENTOF2 100      /* kurtz/evalues.c:51 */

/* kurtz/evalues.c:59-83 */
static const double averagequot[] = {
    0.0,      3.97e+00, 1.28e+01, 3.26e+01, 7.60e+01, 1.71e+02, 3.77e+02,
    8.22e+02, 1.78e+03, 3.91e+03, 8.50e+03, 1.76e+04, 3.78e+04, 7.98e+04,
    1.66e+05, 3.58e+05, 7.44e+05, 1.52e+06, 3.20e+06, 6.40e+06, 1.31e+07};

typedef struct
{
  double probmatch, first;
  int64_t *linestart; /* nextline entries are valid */
  uint64_t nextline, alloclines;
  double *table;
  uint64_t nexttab, alloctab;
} Evalues;

/* strings that depend on (distance, length) only, per formatting thread:
   "%.2e" of the E-value with its leading blanks, and "%.2f" of the identity
   (sprintf of a double costs more than the rest of the line) */
#define CACHE_MAXD 32
#define CACHE_MAXLEN 1024
typedef struct
{
  char str[20];
  uint8_t len;
} Cstr;

typedef struct
{
  double multiplier; /* the E-value entries belong to this multiplier */
  Cstr *evalue;      /* [2*CACHE_MAXD+1][CACHE_MAXLEN+1], len 0 = empty */
  Cstr *identity;    /* [CACHE_MAXD+1][CACHE_MAXLEN+1] */
} Fcache;

#define LINEMAX 192 /* upper bound of one output line */

struct vsa_sink
{
  vsa_sinkparams p;
  uint64_t *dbmarkpos, *qstart, *qlength;
  uint64_t dblen; /* DATABASELENGTH, include/multidef.h:91-92 */
  int wlength, wpos1, wseq1, wpos2, wseq2;
  Evalues ev;
  uint64_t idnumber;
  char *line;
  Fcache cache; /* of the calling thread */
};

/* incprecomputehammingEvalues, kurtz/evalues.c:316-368 */
static int evalues_extend(Evalues *h, int64_t kmax)
{
  int64_t k, l;

  if ((uint64_t) kmax + 3 > h->alloclines)
  {
    h->alloclines = (uint64_t) kmax + 3 + 256;
    h->linestart =
        (int64_t *) realloc(h->linestart, h->alloclines * sizeof(int64_t));
    if (h->linestart == NULL)
    {
      return -1;
    }
  }
  for (k = (int64_t) h->nextline; k <= kmax; k++)
  {
    double prob;
    h->linestart[k] = (int64_t) h->nexttab - (k + 1);
    prob = h->first;
    h->first *= (((double) (k + 2) / (k + 1)) * (1.0 - h->probmatch));
    for (l = k + 1; prob > SMALLESTEVALUE; l++)
    {
      if (h->nexttab == h->alloctab)
      {
        h->alloctab = h->alloctab ? 2 * h->alloctab : 4096;
        h->table = (double *) realloc(h->table, h->alloctab * sizeof(double));
        if (h->table == NULL)
        {
          return -1;
        }
      }
      h->table[h->nexttab++] = prob;
      prob *= ((l + 1) * h->probmatch / (l + 1 - k));
    }
  }
  h->linestart[kmax + 1] = (int64_t) h->nexttab - (kmax + 1 + 1);
  h->nextline = (uint64_t) (kmax + 1);
  return 0;
}

/* inclookupEvalue, kurtz/evalues.c:370-386 */
static double evalues_lookup(Evalues *h, int64_t distance, int64_t length)
{
  int64_t i;

  if (distance + 1 > (int64_t) h->nextline)
  {
    if (evalues_extend(h, distance) != 0)
    {
      return 0.0;
    }
  }
  i = h->linestart[distance] + length;
  if (i < h->linestart[distance + 1] + distance + 2)
  {
    return h->table[i];
  }
  return 0.0;
}

/* incgetEvalue, kurtz/evalues.c:388-426 */
static double evalues_get(Evalues *h, double multiplier, int64_t distance,
                          uint64_t length)
{
  if (distance <= 0)
  {
    return multiplier * evalues_lookup(h, -distance, (int64_t) length);
  }
  if (distance > 20)
  {
    if (distance - 20 > MAXEXPONENTOF2)
    {
      return 0.0;
    }
    return multiplier * (1.31e+07 * pow(2.0, (double) (distance - 20))) *
           evalues_lookup(h, distance, (int64_t) length);
  }
  return multiplier * averagequot[distance] *
         evalues_lookup(h, distance, (int64_t) length);
}

static void cache_free(Fcache *c);

static int digitsof(uint64_t v)
{
  /* 1 + (Uint) log10((double) v), Vmatch/assigndig.c:29-45 */
  return 1 + (int) log10((double) v);
}

int vsa_sink_open(const vsa_sinkparams *params, vsa_sink **sink)
{
  vsa_sink *s;
  uint64_t nsep;

  if (params == NULL || sink == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_sink_open: NULL argument");
    return -1;
  }
  *sink = NULL;
  if (params->numofsequences == 0 || params->totallength == 0 ||
      params->numofchars < 2 ||
      (params->numofsequences > 1 && params->markpos == NULL) ||
      (params->kind != VSA_SINK_SELF && params->numofqueries > 0 &&
       (params->querystart == NULL || params->querylength == NULL)))
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_sink_open: incomplete parameters");
    return -1;
  }
  s = (vsa_sink *) calloc(1, sizeof *s);
  if (s == NULL)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -101;
  }
  s->p = *params;
  nsep = params->numofsequences - 1;
  s->dbmarkpos = (uint64_t *) malloc((nsep + 1) * sizeof(uint64_t));
  s->qstart = (uint64_t *) malloc((params->numofqueries + 1) * 8);
  s->qlength = (uint64_t *) malloc((params->numofqueries + 1) * 8);
  s->line = (char *) malloc(512);
  if (s->dbmarkpos == NULL || s->qstart == NULL || s->qlength == NULL ||
      s->line == NULL)
  {
    vsa_sink_close(s);
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -101;
  }
  if (nsep > 0)
  {
    memcpy(s->dbmarkpos, params->markpos, nsep * 8);
  }
  if (params->kind != VSA_SINK_SELF && params->numofqueries > 0)
  {
    memcpy(s->qstart, params->querystart, params->numofqueries * 8);
    memcpy(s->qlength, params->querylength, params->numofqueries * 8);
  }
  s->p.markpos = s->dbmarkpos;
  s->p.querystart = s->qstart;
  s->p.querylength = s->qlength;
  /* assignvirtualdigits / assignquerydigits, Vmatch/assigndig.c:6-46 */
  s->dblen = params->totallength - params->totalquerylength - 1;
  s->wlength = s->dblen < 1000 ? 2
               : s->dblen < 10000 ? 3
               : s->dblen < 100000 ? 4 : 5;
  s->wpos1 = digitsof(s->dblen);
  s->wseq1 = digitsof(params->numofsequences - params->numofquerysequences);
  s->wpos2 = s->wpos1;
  s->wseq2 = s->wseq1;
  if (params->kind != VSA_SINK_SELF)
  {
    s->wpos2 = digitsof(params->querytotallength);
    s->wseq2 = digitsof(params->numofqueries);
  }
  /* inithammingEvalues(&evalues, 1.0 / (mapsize - 1)),
     Vmatch/procmatch.c:545, kurtz/evalues.c:307-314 */
  s->ev.probmatch = 1.0 / (double) params->numofchars;
  s->ev.first = s->ev.probmatch *
                ((1.0 - s->ev.probmatch) * (1.0 - s->ev.probmatch));
  *sink = s;
  return 0;
}

void vsa_sink_close(vsa_sink *s)
{
  if (s != NULL)
  {
    free(s->dbmarkpos);
    free(s->qstart);
    free(s->qlength);
    free(s->line);
    cache_free(&s->cache);
    free(s->ev.linestart);
    free(s->ev.table);
    free(s);
  }
}

/* getseqinfo on the index (kurtz-basic/multiseq-adv.c:277-284): number of
   separators in front of pos, start and length of that sequence */
static void seqinfo(const vsa_sink *s, uint64_t pos, uint64_t *seqnum,
                    uint64_t *seqstart, uint64_t *seqlength)
{
  uint64_t lo = 0, hi = s->p.numofsequences - 1, end;

  while (lo < hi) /* first separator position > pos */
  {
    const uint64_t mid = lo + (hi - lo) / 2;
    if (s->dbmarkpos[mid] > pos)
    {
      hi = mid;
    } else
    {
      lo = mid + 1;
    }
  }
  *seqnum = lo;
  *seqstart = (lo == 0) ? 0 : s->dbmarkpos[lo - 1] + 1;
  end = (lo == s->p.numofsequences - 1) ? s->p.totallength : s->dbmarkpos[lo];
  *seqlength = end - *seqstart;
}

/* right-aligned unsigned number, like "%*lu" */
static char *putunsigned(char *o, uint64_t v, int width)
{
  char tmp[24];
  int n = 0, i;

  do
  {
    tmp[n++] = (char) ('0' + v % 10);
    v /= 10;
  } while (v != 0);
  for (i = n; i < width; i++)
  {
    *o++ = ' ';
  }
  while (n > 0)
  {
    *o++ = tmp[--n];
  }
  return o;
}

/* like "%*ld" */
static char *putsigned(char *o, int64_t v, int width)
{
  char tmp[24];
  int n = 0, i;
  uint64_t u = v < 0 ? (uint64_t) (-v) : (uint64_t) v;

  do
  {
    tmp[n++] = (char) ('0' + u % 10);
    u /= 10;
  } while (u != 0);
  if (v < 0)
  {
    tmp[n++] = '-';
  }
  for (i = n; i < width; i++)
  {
    *o++ = ' ';
  }
  while (n > 0)
  {
    *o++ = tmp[--n];
  }
  return o;
}

static int cache_init(Fcache *c)
{
  c->multiplier = -1.0;
  c->evalue = (Cstr *) calloc((size_t) (2 * CACHE_MAXD + 1) *
                                  (CACHE_MAXLEN + 1), sizeof(Cstr));
  c->identity = (Cstr *) calloc((size_t) (CACHE_MAXD + 1) *
                                    (CACHE_MAXLEN + 1), sizeof(Cstr));
  return (c->evalue == NULL || c->identity == NULL) ? -1 : 0;
}

static void cache_free(Fcache *c)
{
  free(c->evalue);
  free(c->identity);
  c->evalue = c->identity = NULL;
}

/* vmatchnormaloutmatch, echomatch.c:955-962 */
static int evaluestring(char *o, double evalue)
{
  int n = 0;

  if (evalue >= 1.0e-99 || evalue == 0.0)
  {
    o[n++] = ' ';
  }
  n += sprintf(o + n, "   %.2e", evalue);
  return n;
}

/* EVALIDENTITY, include/match.h:122-135; echomatch.c:968-979 */
static int identitystring(char *o, int64_t ad, uint64_t longer)
{
  const double identity = 100.0 * (1.0 - (double) ad / longer);
  int n = 0;

  if (identity < 100.0)
  {
    o[n++] = ' ';
  }
  n += sprintf(o + n, "   %.2f", identity);
  return n;
}

/* one match -> one line with its newline at o; returns the end of the line,
   o itself if matchokay rejects the match, NULL on error.  The E-value table
   must already reach the distance (vsa_sink_* see to that): no writes to
   shared state from here. */
static char *formatmatch(const vsa_sink *s, Fcache *cache,
                         const vsa_match *m, char *o)
{
  uint64_t seqnum1, start1, len1seq, length1, length2, seqnum2, relpos2,
      position1, position2, lenmatch;
  int64_t distance, ad, score;
  double multiplier = 0.0;
  const int isquery = s->p.kind != VSA_SINK_SELF;
  const int iscomplete = s->p.kind == VSA_SINK_COMPLETE ||
                         s->p.kind == VSA_SINK_APPROX_EDIST ||
                         s->p.kind == VSA_SINK_APPROX_HAMMING;

  position1 = m->dbstart;
  length1 = m->length;
  seqinfo(s, position1, &seqnum1, &start1, &len1seq);
  if (isquery)
  {
    const uint64_t q = m->queryseq;
    uint64_t seqstart2, seqlength2;
    if (q >= s->p.numofqueries)
    {
      return NULL;
    }
    seqstart2 = s->qstart[q];
    seqlength2 = s->qlength[q];
    seqnum2 = q;
    if (s->p.kind == VSA_SINK_QUERY)
    {
      length2 = m->length;
      relpos2 = m->querystart;
      distance = 0;
    } else
    {
      /* initcompletematchstruct, Vmengine/initcompl.c:7-21 */
      length2 = seqlength2;
      relpos2 = 0;
      distance = (s->p.kind == VSA_SINK_APPROX_EDIST)
                     ? (int64_t) m->querystart
                     : (s->p.kind == VSA_SINK_APPROX_HAMMING)
                           ? -(int64_t) m->querystart
                           : 0;
    }
    if (s->p.palindromic)
    {
      relpos2 = seqlength2 - (relpos2 + length2); /* procfinal.c:152-158 */
      if (s->p.selfpalindromic &&
          (seqnum1 > seqnum2 ||
           (seqnum1 == seqnum2 && position1 - start1 > relpos2)))
      {
        return o; /* the mirror image is the one reported, :159-167 */
      }
    }
    position2 = seqstart2 + relpos2;
  } else
  {
    /* self match: (length, start1, start2), Vmengine/fself.c:95-125 */
    uint64_t start2, len2seq;
    length2 = m->length;
    distance = 0;
    seqinfo(s, m->queryseq, &seqnum2, &start2, &len2seq);
    relpos2 = m->queryseq - start2;
    position2 = m->queryseq;
    if (s->p.totalquerylength > 0) /* convertthematch, procfinal.c:466-474 */
    {
      seqnum2 -= s->p.numofsequences - s->p.numofquerysequences;
      position2 -= s->dblen + 1;
    }
  }
  /* matchokay, Vmatch/mokay.c:17-27 */
  if (length1 < s->p.leastlength || length2 < s->p.leastlength)
  {
    return o;
  }
  ad = distance < 0 ? -distance : distance;
  /* vmatchnormaloutmatch, Vmatch/echomatch.c:878-987 */
  o = putunsigned(o, length1, s->wlength);
  if (s->p.showmode & VSA_SHOW_ABSOLUTE)
  {
    *o++ = ' ';
    o = putunsigned(o, position1, s->wpos1);
  } else
  {
    memcpy(o, "    ", 4);
    o = putunsigned(o + 4, seqnum1, s->wseq1);
    *o++ = ' ';
    o = putunsigned(o, position1 - start1, s->wpos1);
  }
  memcpy(o, s->p.palindromic ? "   P " : "   D ", 5);
  o = putunsigned(o + 5, length2, s->wlength);
  if (s->p.showmode & VSA_SHOW_ABSOLUTE)
  {
    *o++ = ' ';
    o = putunsigned(o, position2, s->wpos2);
  } else
  {
    memcpy(o, "    ", 4);
    o = putunsigned(o + 4, seqnum2, s->wseq2);
    *o++ = ' ';
    o = putunsigned(o, relpos2, s->wpos2);
  }
  if (!(s->p.showmode & VSA_SHOW_NODIST))
  {
    *o++ = ' ';
    o = putsigned(o, distance, 3);
  }
  if (!(s->p.showmode & VSA_SHOW_NOEVALUE))
  {
    /* assignEvalue, Vmatch/procfinal.c:195-257 */
    Cstr *slot = NULL;
    if (isquery)
    {
      multiplier = iscomplete ? (double) s->p.totallength
                   : (s->p.palindromic && s->p.selfpalindromic)
                       ? 0.5 * (double) s->p.totallength *
                             (double) s->p.querytotallength
                       : (double) s->p.totallength *
                             (double) s->qlength[m->queryseq];
    } else if (s->p.totalquerylength > 0)
    {
      multiplier = (double) s->dblen * (double) s->p.totalquerylength;
    } else
    {
      multiplier = 0.5 * (double) s->p.totallength *
                   (double) s->p.totallength;
    }
    lenmatch = (iscomplete || distance == 0)
                   ? length2
                   : (length1 > length2 ? length1 : length2);
    if (ad <= CACHE_MAXD && lenmatch <= CACHE_MAXLEN)
    {
      if (multiplier != cache->multiplier)
      {
        memset(cache->evalue, 0, (size_t) (2 * CACHE_MAXD + 1) *
                                     (CACHE_MAXLEN + 1) * sizeof(Cstr));
        cache->multiplier = multiplier;
      }
      slot = cache->evalue + (size_t) (distance + CACHE_MAXD) *
                                 (CACHE_MAXLEN + 1) + lenmatch;
    }
    if (slot != NULL && slot->len != 0)
    {
      memcpy(o, slot->str, slot->len);
      o += slot->len;
    } else
    {
      /* const cast: reads only, the table reaches this distance */
      const double evalue = evalues_get((Evalues *) &s->ev, multiplier,
                                        distance, lenmatch);
      const int n = evaluestring(o, evalue);
      if (slot != NULL && n < (int) sizeof slot->str)
      {
        memcpy(slot->str, o, (size_t) n);
        slot->len = (uint8_t) n;
      }
      o += n;
    }
  }
  if (!(s->p.showmode & VSA_SHOW_NOSCORE))
  {
    /* EVALDISTANCE2SCORE, include/match.h:114-116 */
    score = (distance >= 0)
                ? (int64_t) (length1 + length2) - 3 * distance
                : -((int64_t) (length1 + length2) + 3 * distance);
    *o++ = ' ';
    o = putsigned(o, score, s->wlength + 1);
  }
  if (!(s->p.showmode & VSA_SHOW_NOIDENTITY))
  {
    const uint64_t longer = length1 > length2 ? length1 : length2;
    if (ad == 0)
    {
      memcpy(o, "   100.00", 9);
      o += 9;
    } else if (ad <= CACHE_MAXD && longer <= CACHE_MAXLEN)
    {
      Cstr *slot = cache->identity + (size_t) ad * (CACHE_MAXLEN + 1) + longer;
      if (slot->len == 0)
      {
        slot->len = (uint8_t) identitystring(slot->str, ad, longer);
      }
      memcpy(o, slot->str, slot->len);
      o += slot->len;
    } else
    {
      o += identitystring(o, ad, longer);
    }
  }
  *o++ = '\n';
  return o;
}

/* the E-value table up to the largest distance of the batch, so that the
   formatting threads only read it */
static int prepare(vsa_sink *s, const vsa_match *matches, uint64_t n)
{
  uint64_t i, maxd = 0;

  if (s->cache.evalue == NULL && cache_init(&s->cache) != 0)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -101;
  }
  if (s->p.kind == VSA_SINK_APPROX_EDIST ||
      s->p.kind == VSA_SINK_APPROX_HAMMING)
  {
    for (i = 0; i < n; i++)
    {
      if (matches[i].querystart > maxd)
      {
        maxd = matches[i].querystart;
      }
    }
  }
  if (maxd + 1 > s->ev.nextline &&
      evalues_extend(&s->ev, (int64_t) maxd) != 0)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -101;
  }
  return 0;
}

typedef struct
{
  const vsa_sink *sink;
  const vsa_match *matches;
  uint64_t n;
  char *buffer; /* n * LINEMAX bytes */
  uint64_t used;
  int failed;
} Chunk;

static void *formatchunk(void *arg)
{
  Chunk *c = (Chunk *) arg;
  Fcache cache;
  char *o = c->buffer;
  uint64_t i;

  c->failed = cache_init(&cache);
  for (i = 0; c->failed == 0 && i < c->n; i++)
  {
    char *e = formatmatch(c->sink, &cache, c->matches + i, o);
    if (e == NULL)
    {
      c->failed = 1;
      break;
    }
    o = e;
  }
  c->used = (uint64_t) (o - c->buffer);
  cache_free(&cache);
  return NULL;
}

#define ROUNDMATCHES (1u << 20) /* matches per round of the worker threads */

/* formats matches[0..n) in order, handing every finished piece of text to
   emit(); threads: 0 = one per online processor (at most 32) */
static int formatall(vsa_sink *s, const vsa_match *matches, uint64_t n,
                     int threads, int (*emit)(void *, const char *, uint64_t),
                     void *emitinfo)
{
  uint64_t done = 0;
  int t, nthreads = threads;
  Chunk *chunks;
  pthread_t *tid;
  char *arena;

  if (prepare(s, matches, n) != 0)
  {
    return -101;
  }
  if (nthreads <= 0)
  {
    const long cpus = sysconf(_SC_NPROCESSORS_ONLN);
    nthreads = cpus < 1 ? 1 : (cpus > 32 ? 32 : (int) cpus);
  }
  if (n < 4096)
  {
    nthreads = 1;
  }
  chunks = (Chunk *) calloc((size_t) nthreads, sizeof(Chunk));
  tid = (pthread_t *) calloc((size_t) nthreads, sizeof(pthread_t));
  arena = (char *) malloc((size_t) (n < ROUNDMATCHES ? n : ROUNDMATCHES) *
                              LINEMAX + LINEMAX);
  if (chunks == NULL || tid == NULL || arena == NULL)
  {
    free(chunks);
    free(tid);
    free(arena);
    snprintf(vsa_errbuf(), ERRSIZE, "out of memory");
    return -101;
  }
  while (done < n)
  {
    const uint64_t round = (n - done < ROUNDMATCHES) ? n - done : ROUNDMATCHES;
    const uint64_t per = (round + (uint64_t) nthreads - 1) / (uint64_t) nthreads;
    int used = 0, rc = 0;
    for (t = 0; t < nthreads; t++)
    {
      const uint64_t first = (uint64_t) t * per;
      if (first >= round)
      {
        break;
      }
      chunks[t].sink = s;
      chunks[t].matches = matches + done + first;
      chunks[t].n = (first + per <= round) ? per : round - first;
      chunks[t].buffer = arena + first * LINEMAX;
      chunks[t].used = 0;
      chunks[t].failed = 0;
      used++;
    }
    if (used == 1)
    {
      (void) formatchunk(chunks);
    } else
    {
      for (t = 0; t < used; t++)
      {
        if (pthread_create(tid + t, NULL, formatchunk, chunks + t) != 0)
        {
          (void) formatchunk(chunks + t); /* no thread: do it here */
          tid[t] = 0;
        }
      }
      for (t = 0; t < used; t++)
      {
        if (tid[t] != 0)
        {
          (void) pthread_join(tid[t], NULL);
        }
      }
    }
    for (t = 0; t < used && rc == 0; t++)
    {
      if (chunks[t].failed)
      {
        snprintf(vsa_errbuf(), ERRSIZE, "match refers to a query outside "
                 "the %lu of the sink, or out of memory",
                 (unsigned long) s->p.numofqueries);
        rc = -2;
      } else
      {
        rc = emit(emitinfo, chunks[t].buffer, chunks[t].used);
      }
    }
    if (rc != 0)
    {
      free(chunks);
      free(tid);
      free(arena);
      return rc;
    }
    done += round;
  }
  free(chunks);
  free(tid);
  free(arena);
  return 0;
}

typedef struct
{
  char *buffer;
  uint64_t used, capacity;
} Membuf;

static int emitmemory(void *info, const char *text, uint64_t len)
{
  Membuf *mb = (Membuf *) info;

  if (mb->used + len > mb->capacity)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_sink_format: buffer of %lu bytes "
             "is too small", (unsigned long) mb->capacity);
    return -3;
  }
  memcpy(mb->buffer + mb->used, text, (size_t) len);
  mb->used += len;
  return 0;
}

static int emitfile(void *info, const char *text, uint64_t len)
{
  if (len > 0 && fwrite(text, 1, (size_t) len, (FILE *) info) != (size_t) len)
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_sink_write: write failed");
    return -4;
  }
  return 0;
}

int64_t vsa_sink_format(vsa_sink *s, const vsa_match *matches, uint64_t n,
                        char *buffer, uint64_t capacity)
{
  Membuf mb;
  int rc;

  if (s == NULL || (n > 0 && matches == NULL) ||
      (capacity > 0 && buffer == NULL))
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_sink_format: NULL argument");
    return -1;
  }
  mb.buffer = buffer;
  mb.used = 0;
  mb.capacity = capacity;
  rc = formatall(s, matches, n, s->p.threads, emitmemory, &mb);
  return rc != 0 ? (int64_t) rc : (int64_t) mb.used;
}

int vsa_sink_write(vsa_sink *s, const vsa_match *matches, uint64_t n,
                   void *file)
{
  if (s == NULL || file == NULL || (n > 0 && matches == NULL))
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_sink_write: NULL argument");
    return -1;
  }
  return formatall(s, matches, n, s->p.threads, emitfile, file);
}
