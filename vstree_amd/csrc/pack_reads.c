/* Reads at two bits per symbol, made on the host (include/vstree_amd.h:
   vsa_pack_reads, vsa_pack_reads_mt).  The reads of a sequencer arrive as one
   mapped symbol per byte (the reference's Multiseq, kurtz-basic/multiseq.c:
   129-166); a row of W 64-bit words per read is what crosses PCIe and lies in
   HBM.  Eight symbols per step where the CPU has PEXT (BMI2: byte-swap so
   that the first symbol ends in the top bits, then gather the two low bits of
   every byte), one symbol per step otherwise; reads with a special symbol are
   rare and are put on the side list in a second, sequential pass so that
   their numbers do not depend on the number of threads. */
#include <pthread.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#if defined(__x86_64__)
#include <immintrin.h>
#endif
#include "vstree_amd.h"

char *vsa_errbuf(void);
#define ERRSIZE 1024

typedef struct
{
  const uint8_t *symbols;
  uint64_t first, last; /* reads [first, last) */
  uint32_t m, W;
  uint64_t stride;
  uint64_t *rows;
  uint64_t *flagged; /* reads of the range with a special symbol, ascending */
  uint64_t nflagged, room;
  int failed, usepext;
} Piece;

static int noteflagged(Piece *p, uint64_t read)
{
  if (p->nflagged == p->room)
  {
    const uint64_t room = p->room == 0 ? 64 : 2 * p->room;
    uint64_t *f = (uint64_t *) realloc(p->flagged, room * sizeof *f);
    if (f == NULL)
    {
      return -1;
    }
    p->flagged = f;
    p->room = room;
  }
  p->flagged[p->nflagged++] = read;
  return 0;
}

static void packscalar(Piece *p)
{
  const uint32_t m = p->m, W = p->W;
  uint64_t i;

  for (i = p->first; i < p->last; i++)
  {
    const uint8_t *r = p->symbols + i * p->stride;
    uint64_t *row = p->rows + i * W;
    uint8_t bad = 0;
    uint32_t w, j;
    for (w = 0; w < W; w++)
    {
      uint64_t acc = 0;
      const uint32_t lo = 32 * w, hi = lo + 32 < m ? lo + 32 : m;
      for (j = lo; j < hi; j++)
      {
        const uint8_t c = r[j];
        bad |= c;
        acc |= (uint64_t) (c & 3u) << (62 - 2 * (j - lo));
      }
      row[w] = acc;
    }
    if (bad > 3 && noteflagged(p, i) != 0)
    {
      p->failed = 1;
      return;
    }
  }
}

#if defined(__x86_64__)
__attribute__((target("bmi2"))) static void packpext(Piece *p)
{
  const uint32_t m = p->m, W = p->W;
  const uint64_t low2 = 0x0303030303030303ull;
  uint64_t i;

  for (i = p->first; i < p->last; i++)
  {
    const uint8_t *r = p->symbols + i * p->stride;
    uint64_t *row = p->rows + i * W;
    uint64_t bad = 0;
    uint32_t w;
    for (w = 0; w < W; w++)
    {
      uint64_t acc = 0;
      const uint32_t lo = 32 * w, hi = lo + 32 < m ? lo + 32 : m;
      uint32_t j = lo, shift = 48;
      for (; j + 8 <= hi; j += 8, shift -= 16)
      {
        uint64_t x;
        memcpy(&x, r + j, 8);
        bad |= x;
        acc |= _pext_u64(__builtin_bswap64(x), low2) << shift;
      }
      if (j < hi)
      {
        /* (the last symbols of the read: nothing behind them is touched) */
        uint64_t x = 0;
        memcpy(&x, r + j, hi - j);
        bad |= x;
        acc |= _pext_u64(__builtin_bswap64(x), low2) << shift;
      }
      row[w] = acc;
    }
    if ((bad & ~low2) != 0 && noteflagged(p, i) != 0)
    {
      p->failed = 1;
      return;
    }
  }
}
#endif

static void *packpiece(void *arg)
{
  Piece *p = (Piece *) arg;
#if defined(__x86_64__)
  if (p->usepext)
  {
    packpext(p);
    return NULL;
  }
#endif
  packscalar(p);
  return NULL;
}

int vsa_pack_reads_mt(const uint8_t *symbols, uint64_t numofqueries,
                      uint32_t querylength, uint64_t stride, uint64_t *rows,
                      uint8_t *special, uint64_t specialcapacity,
                      uint64_t *numofspecial, uint32_t threads)
{
  const uint32_t m = querylength, W = (2 * m + 8 + 63) / 64;
  Piece *pieces;
  pthread_t *tids;
  uint32_t t, started = 0;
  uint64_t ns;
  int rc = 0, usepext = 0;

  if ((numofqueries > 0 && (symbols == NULL || rows == NULL)) ||
      numofspecial == NULL || querylength == 0 ||
      (specialcapacity > 0 && special == NULL))
  {
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_pack_reads: bad argument");
    return -1;
  }
  if (threads == 0)
  {
    threads = 1;
  }
  if (threads > 256)
  {
    threads = 256;
  }
  /* (a thread per 64 k reads at least: below that its start costs more) */
  while (threads > 1 && numofqueries / threads < 65536)
  {
    threads--;
  }
#if defined(__x86_64__)
  usepext = __builtin_cpu_supports("bmi2") && getenv("VSA_PACK_SCALAR") == NULL;
#endif
  pieces = (Piece *) calloc(threads, sizeof *pieces);
  tids = (pthread_t *) calloc(threads, sizeof *tids);
  if (pieces == NULL || tids == NULL)
  {
    free(pieces);
    free(tids);
    snprintf(vsa_errbuf(), ERRSIZE, "vsa_pack_reads: out of memory");
    return -100;
  }
  for (t = 0; t < threads; t++)
  {
    Piece *p = pieces + t;
    p->symbols = symbols;
    p->first = numofqueries / threads * t +
               (t < numofqueries % threads ? t : numofqueries % threads);
    p->last = p->first + numofqueries / threads +
              (t < numofqueries % threads ? 1 : 0);
    p->m = m;
    p->W = W;
    p->stride = stride;
    p->rows = rows;
    p->usepext = usepext;
  }
  for (t = 1; t < threads; t++)
  {
    if (pthread_create(tids + t, NULL, packpiece, pieces + t) != 0)
    {
      break;
    }
    started = t;
  }
  packpiece(pieces); /* the caller's thread takes the first piece ... */
  for (t = started + 1; t < threads; t++)
  {
    packpiece(pieces + t); /* ... and those that got no thread */
  }
  for (t = 1; t <= started; t++)
  {
    pthread_join(tids[t], NULL);
  }
  /* the side list, in the order of the reads */
  ns = *numofspecial;
  for (t = 0; t < threads && rc == 0; t++)
  {
    uint64_t k;
    if (pieces[t].failed)
    {
      snprintf(vsa_errbuf(), ERRSIZE, "vsa_pack_reads: out of memory");
      rc = -100;
    }
    for (k = 0; k < pieces[t].nflagged && rc == 0; k++)
    {
      const uint64_t i = pieces[t].flagged[k];
      uint64_t *row = rows + i * W;
      uint32_t w;
      if (ns >= specialcapacity)
      {
        /* a symbol that is no base (a wildcard; in a Multiseq also a
           separator would be): the read travels as bytes */
        snprintf(vsa_errbuf(), ERRSIZE,
                 "vsa_pack_reads: more than %lu reads with a special symbol",
                 (unsigned long) specialcapacity);
        rc = -2;
        break;
      }
      memcpy(special + ns * m, symbols + i * stride, m);
      for (w = 0; w < W; w++)
      {
        row[w] = 0;
      }
      row[0] = ns++ << 8; /* (bits 8 ..: the flag byte is word W - 1's, which
                             is this word when W = 1) */
      row[W - 1] |= 1u;
    }
  }
  for (t = 0; t < threads; t++)
  {
    free(pieces[t].flagged);
  }
  free(pieces);
  free(tids);
  *numofspecial = ns;
  return rc;
}

int vsa_pack_reads(const uint8_t *symbols, uint64_t numofqueries,
                   uint32_t querylength, uint64_t stride, uint64_t *rows,
                   uint8_t *special, uint64_t specialcapacity,
                   uint64_t *numofspecial)
{
  return vsa_pack_reads_mt(symbols, numofqueries, querylength, stride, rows,
                           special, specialcapacity, numofspecial, 1);
}
