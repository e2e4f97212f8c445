/*
  TEST INFRASTRUCTURE -- NOT PRODUCT CODE.

  oracle/ is a plain-C, single-threaded CPU restatement of the reference's
  Vmengine query path (vmatch -complete / -l / -mum / -mum cand against a query
  file, and -mum on an index that contains its queries).  It exists so that
  the HIP kernels can be checked bit for bit; only tests/,
  __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it.

  Parity status: PINNED.  tests/test_oracle_golden.py checks this restatement
  against (a) the reference's own known-answer file
  src/Vmatch/Testdir/LargePat.res, (b) outputs of the real reference programs
  built by oracle/Makefile.ref (fixtures under tests/golden/, generator
  scripts/make_golden.py) and (c) live against oracle/_ref/vmatch_ref when
  that binary is present.

  Every function cites the reference file:line it follows (paths relative to
  /root/reference/src).
*/
#ifndef VSORACLE_H
#define VSORACLE_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ORC_SEPARATOR 255u /* include/chardef.h:19 */
#define ORC_WILDCARD  254u /* include/chardef.h:25 */
#define ORC_UNDEFBWT  253u /* include/chardef.h:31,50 */
#define ORC_ISSPECIAL(C) ((C) >= (uint8_t) ORC_WILDCARD) /* chardef.h:37 */

/*
  The tables of one enhanced suffix array as mkvtree writes them
  (include/virtualdef.h:186-219, writers src/Mkvtree/mkvprocess.c:99-816).
  isize is the byte width of the entries of suf, bck and llv (4 or 8); llv
  holds nllv pairs (index, value), sorted by index.
*/
typedef struct
{
  uint64_t n;            /* multiseq.totallength */
  uint32_t prefixlength;
  uint32_t numofchars;   /* alpha.mapsize - 1; 4 for DNA */
  uint32_t isize;
  uint64_t nllv;
  const uint8_t *tis;    /* n */
  const void *suf;       /* n + 1 */
  const uint8_t *lcp;    /* n + 1 */
  const void *llv;       /* 2 * nllv */
  const void *bck;       /* 2 * numofchars^prefixlength */
  const uint8_t *bwt;    /* n + 1 or NULL */
  const uint8_t *sti1;   /* n + 1 or NULL (needed by algorithm 2 only) */
  /* only for orc_selfmum: position of the separator between the database
     and the indexed queries (getqueryseppos), valid if hasqueries != 0 */
  uint64_t querysepposition;
  int hasqueries;
} orc_index;

/* layout of the reference's MUMcandidate (include/mumcand.h:17-23) */
typedef struct
{
  uint64_t length, dbstart, queryseq, querystart;
} orc_match;

typedef struct
{
  orc_match *m;
  uint64_t n, cap;
} orc_matches;

/* counters in the spirit of the reference's -DCOUNT probes
   (kurtz/maxpref.c:17-23): they feed the algorithmic-bytes figure */
typedef struct
{
  uint64_t charcomp;   /* symbol comparisons in COMPARE */
  uint64_t sufprobes;  /* suffixes examined by findmaxprefixlen */
  uint64_t lcpreads;   /* lcptab entries read by the left/right scans */
  uint64_t bckreads;   /* bcktab pairs read */
  uint64_t searches;   /* calls of findmaxprefixlen */
  uint64_t emitted;    /* matches emitted */
} orc_counters;

void orc_matches_init(orc_matches *out);
void orc_matches_free(orc_matches *out);
void orc_counters_get(orc_counters *c);
void orc_counters_reset(void);

/*
  All query functions take the queries as nq (start, length) pairs into one
  symbol buffer qbuf of alphabet-mapped symbols; byte qbuf[start-1] is only
  read for start > 0.  Return 0 on success, a negative code on error with a
  message in err (at least 256 bytes), mirroring the reference's Sint/ERRORn
  convention (include/errordef.h:45-82).
*/

/* vmatch -complete -q Q IDX: Vmengine/fcomplete.c:263-321 ->
   Vmengine/exactcompl.c:168-239.  Emits per query, in suffix array order. */
int orc_findcompletematches(const orc_index *idx, const uint8_t *qbuf,
                            const uint64_t *qstart, const uint64_t *qlen,
                            uint64_t nq, orc_matches *out, char *err);

/* vmatch -online -complete: Boyer-Moore-Horspool scan of the text,
   Vmengine/exactcompl.c:277-325.  Independent second checker. */
int orc_findcompletematches_online(const orc_index *idx, const uint8_t *qbuf,
                                   const uint64_t *qstart,
                                   const uint64_t *qlen, uint64_t nq,
                                   orc_matches *out, char *err);

/* vmatch [-mum [cand]] -l L -q Q IDX: Vmengine/fquery.c:1009-1058.
   speedup 0 = kurtz/matchsub.c:165-235, 2 = kurtz/matchsub.c:353-537. */
int orc_findquerymatches(const orc_index *idx, const uint8_t *qbuf,
                         const uint64_t *qstart, const uint64_t *qlen,
                         uint64_t nq, int domum, int domumcand,
                         uint64_t searchlength, int speedup, orc_matches *out,
                         char *err);

/* vmatch -mum -l L IDX (queries inside the index):
   Vmengine/fmumself.c:10-66.  length = mum length, dbstart = start1,
   queryseq = start2 (absolute position), querystart = 0. */
int orc_findmaximaluniquematches(const orc_index *idx, uint64_t searchlength,
                                 orc_matches *out, char *err);

/* kurtz/cleanMUMcand.c:55-118 on its own (cand is sorted in place) */
int orc_mumuniqueinquery(orc_match *cand, uint64_t ncand, orc_matches *out);
int orc_mumuniqueinquery_carry(orc_match *cand, uint64_t ncand,
                               uint64_t carry, orc_matches *out);

/* vmatch -complete -e K | -h K -q Q IDX (oracle/vsapprox.c):
   Vmengine/approxcompl.c:138-199 -> Vmengine/splitesaapm.c:369-558.
   A match carries the distance in querystart.  percent != 0: threshold =
   m * distvalue / 100.  -4: configuration not covered by the restatement. */
int orc_findapproxcompletematches(const orc_index *idx, const uint8_t *qbuf,
                                  const uint64_t *qstart,
                                  const uint64_t *qlen, uint64_t nq,
                                  int doedist, uint64_t distvalue,
                                  int percent, orc_matches *out, char *err);
/* vmatch -supermax -l L IDX (oracle/vsself.c): Vmengine/fsuper.c:60-165.
   length, dbstart = smaller start, queryseq = larger start, querystart 0. */
int orc_findsupermax(const orc_index *idx, uint64_t searchlength,
                     orc_matches *out, char *err);
/* vmatch -l L IDX, maximal repeats (oracle/vsself.c):
   Vmengine/vmatfind.c:330-541 in the reference's order. */
int orc_findmaximalrepeats(const orc_index *idx, uint64_t searchlength,
                           orc_matches *out, char *err);
/* vmatch -tandem -l L IDX, right branching tandem repeats (oracle/vsself.c):
   Vmengine/ftandem.c:24-304 in the reference's order.  length, dbstart =
   start of the repeat, queryseq = dbstart + length, querystart 0. */
int orc_findtandems(const orc_index *idx, uint64_t searchlength,
                    orc_matches *out, char *err);
uint64_t orc_getoptsplit(int doedist, uint64_t spliterrorbound,
                         uint64_t numofchars, uint64_t textlen,
                         uint64_t patternlength, uint64_t threshold);

#ifdef __cplusplus
}
#endif
#endif
