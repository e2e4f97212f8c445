/*
  cpridxps_amd.so -- the GPU engine behind the reference's OWN plugin hook for
  complete matches (boundary B2 of SURVEY.md 8b):

      vmatch -complete cpridxps_amd.so -q QUERIES INDEX

  works with an UNMODIFIED vmatch binary.  The hook is the Vpluginbundle of
  src/include/vplugin-interface.h:15-54 with the data block of
  src/include/cpridx-data.h:16-30; vmatch opens a shared object whose name
  starts with "cpridxps" (src/Vmatch/parsevm.c:1138-1179), asks it for
  vplugingetinterface and then calls
      vplugininit       src/Vmatch/vmatch.c:98-107
      vpluginadddemand  src/Vmatch/procmatch.c:192-201
      vpluginsearch     once per query, src/Vmengine/fcomplete.c:122-138
      vpluginwrap       src/Vmatch/procmatch.c:699-706
  (vpluginparse is never called for this bundle).

  The per-query call pattern would cost one kernel launch per query.  The
  first vpluginsearch of a pass therefore reaches ALL queries through
  ((Matchstate *) voidMatchstate)->queryinfo->multiseq, runs the whole batch
  in one vsa_findcompletematches, and every call (including the first) is
  answered from that result: the matches of query seqnum2, in suffix array
  order, each reported through data->processfinal exactly like
  processfinalexactmatchinterval does (src/Vmengine/exactcompl.c:142-166).
  A second pass over the reverse complements (vmatch -d -p: the same
  Multiseq, pattern pointers into rcsequence) is recognised by the pattern
  pointer and batched again.

  Compiled against the reference's headers; needs nothing from the vmatch
  executable at link time (vmatch is not linked with -rdynamic), so the
  Match is filled in here like initcompletematchstruct
  (src/Vmengine/initcompl.c:7-21) and messagespace() is looked up at run
  time.
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <dlfcn.h>
#include "types.h"
#include "errordef.h"
#include "virtualdef.h"
#include "multidef.h"
#include "match.h"
#include "select.h"
#include "matchstate.h"
#include "mparms.h"
#include "vplugin-interface.h"
#include "cpridx-data.h"
#include "vstree_amd.h"

typedef struct
{
  vsa_index *index;
  Virtualtree *indexowner;
  /* the batch answered at the moment */
  Multiseq *batchowner;
  int batchrc;            /* direction: 0 forward, 1 reverse complement */
  vsa_match *matches;     /* query order, suffix array order */
  uint64_t nmatches;
  uint64_t *first;        /* first[q] .. first[q+1]: matches of query q */
  uint64_t nq;
  uint64_t failedquery;   /* first query shorter than prefixlength, or nq */
  char failmessage[1024];
} Pluginstate;

static Pluginstate state;

static Sint pluginerror(const char *msg)
{
  /* the reference's ERRORn writes to messagespace() (include/errordef.h);
     reachable only if the executable exports it */
  char *(*ms)(void) = (char *(*)(void)) dlsym(RTLD_DEFAULT, "messagespace");

  if (ms != NULL)
  {
    snprintf(ms(), 1024, "%s", msg);
  } else
  {
    fprintf(stderr, "cpridxps_amd: %s\n", msg);
  }
  return (Sint) -1;
}

static void dropbatch(void)
{
  free(state.matches);
  free(state.first);
  state.matches = NULL;
  state.first = NULL;
  state.nmatches = 0;
  state.batchowner = NULL;
}

static Sint init(/*@unused@*/ void *data)
{
  memset(&state, 0, sizeof state);
  return 0;
}

static Sint adddemand(void *data)
{
  Cpridxpatsearchdata *d = (Cpridxpatsearchdata *) data;

  /* what the exact path reads, Vmatch/mapdemand.c:100-210 */
  d->includedemand = TISTAB | SUFTAB | LCPTAB | BCKTAB;
  d->excludedemand = 0;
  return 0;
}

static Sint parse(/*@unused@*/ void *data)
{
  return 0;
}

static Sint getindex(Virtualtree *virtualtree)
{
  vsa_tables t;

  if (state.index != NULL && state.indexowner == virtualtree)
  {
    return 0;
  }
  if (state.index != NULL)
  {
    vsa_index_close(state.index);
    state.index = NULL;
  }
  if (virtualtree->suftab == NULL || virtualtree->lcptab == NULL ||
      virtualtree->bcktab == NULL || virtualtree->multiseq.sequence == NULL)
  {
    return pluginerror("cpridxps_amd needs the tables tis, suf, lcp and bck");
  }
  memset(&t, 0, sizeof t);
  t.totallength = virtualtree->multiseq.totallength;
  t.prefixlength = (uint32_t) virtualtree->prefixlength;
  t.numofchars = (uint32_t) (virtualtree->alpha.mapsize - 1);
  t.integersize = (uint32_t) (8 * sizeof(Uint));
  t.largelcpvalues = virtualtree->largelcpvalues.nextfreePairUint;
  t.tis = virtualtree->multiseq.sequence;
  t.suf = virtualtree->suftab;
  t.lcp = virtualtree->lcptab;
  t.llv = virtualtree->largelcpvalues.spacePairUint;
  t.bck = virtualtree->bcktab;
  if (vsa_index_from_tables(&t, 0, &state.index) != 0)
  {
    return pluginerror(vsa_messagespace());
  }
  state.indexowner = virtualtree;
  return 0;
}

/* one launch for all queries of this pass */
static Sint runbatch(Multiseq *multiseq, int rc)
{
  Uint i, nq = multiseq->numofsequences;
  uint64_t *start, *length, k;
  vsa_queries *queries = NULL;
  vsa_result *result = NULL;
  int ret;

  dropbatch();
  start = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (nq + 1));
  length = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (nq + 1));
  state.first = (uint64_t *) calloc((size_t) nq + 2, sizeof(uint64_t));
  if (start == NULL || length == NULL || state.first == NULL)
  {
    free(start);
    free(length);
    return pluginerror("out of memory");
  }
  /* sequence boundaries of a Multiseq, kurtz-basic/multiseq.c:129-166 */
  for (i = 0; i < nq; i++)
  {
    Uint s = (i == 0) ? 0 : multiseq->markpos.spaceUint[i - 1] + 1;
    Uint e = (i == nq - 1) ? multiseq->totallength
                           : multiseq->markpos.spaceUint[i];
    start[i] = s;
    length[i] = e - s;
  }
  ret = vsa_queries_from_host(rc ? multiseq->rcsequence : multiseq->sequence,
                              multiseq->totallength, start, length, nq, 0,
                              &queries);
  if (ret != 0)
  {
    free(start);
    free(length);
    return pluginerror(vsa_messagespace());
  }
  state.nq = nq;
  state.failedquery = nq;
  ret = vsa_findcompletematches(state.index, queries, &result);
  if (ret != 0 && result == NULL)
  {
    vsa_queries_free(queries);
    free(start);
    free(length);
    return pluginerror(vsa_messagespace());
  }
  if (ret != 0)
  {
    /* a query shorter than prefixlength: the reference stops AT that query
       (exactcompl.c:179-185), the queries before it are answered */
    snprintf(state.failmessage, sizeof state.failmessage, "%s",
             vsa_messagespace());
    for (i = 0; i < nq; i++)
    {
      if (length[i] < (uint64_t) state.indexowner->prefixlength)
      {
        state.failedquery = i;
        break;
      }
    }
  }
  free(start);
  free(length);
  vsa_queries_free(queries);
  state.nmatches = vsa_result_count(result);
  state.matches =
      (vsa_match *) malloc(sizeof(vsa_match) * (size_t) (state.nmatches + 1));
  if (state.matches == NULL ||
      vsa_result_fetch(result, state.matches, state.nmatches) < 0)
  {
    vsa_result_free(result);
    return pluginerror(state.matches == NULL ? "out of memory"
                                             : vsa_messagespace());
  }
  vsa_result_free(result);
  for (k = 0; k < state.nmatches; k++)
  {
    state.first[state.matches[k].queryseq + 1]++;
  }
  for (i = 0; i < nq; i++)
  {
    state.first[i + 1] += state.first[i];
  }
  state.batchowner = multiseq;
  state.batchrc = rc;
  return 0;
}

static Sint search(void *data)
{
  Cpridxpatsearchdata *d = (Cpridxpatsearchdata *) data;
  Matchstate *matchstate = (Matchstate *) d->voidMatchstate;
  Multiseq *multiseq;
  Match match;
  uint64_t k;
  int rc;

  if (matchstate == NULL || matchstate->queryinfo == NULL ||
      matchstate->queryinfo->multiseq == NULL)
  {
    return pluginerror("cpridxps_amd: no query set behind voidMatchstate");
  }
  multiseq = matchstate->queryinfo->multiseq;
  /* forward or reverse-complement pass? (readmulti.c:93-125) */
  rc = !(d->pattern >= multiseq->sequence &&
         d->pattern <= multiseq->sequence + multiseq->totallength);
  if (rc && (multiseq->rcsequence == NULL ||
             d->pattern < multiseq->rcsequence ||
             d->pattern > multiseq->rcsequence + multiseq->totallength))
  {
    return pluginerror("cpridxps_amd: pattern is not part of the query set");
  }
  if (getindex(d->virtualtree) != 0)
  {
    return (Sint) -1;
  }
  if (state.batchowner != multiseq || state.batchrc != rc)
  {
    if (runbatch(multiseq, rc) != 0)
    {
      return (Sint) -1;
    }
  }
  if ((uint64_t) d->seqnum2 >= state.nq)
  {
    return pluginerror("cpridxps_amd: query number out of range");
  }
  if ((uint64_t) d->seqnum2 >= state.failedquery)
  {
    return pluginerror(state.failmessage);
  }
  /* initcompletematchstruct + processfinalexactmatchinterval */
  match.position2 = 0;
  match.relpos2 = 0;
  match.seqnum2 = d->seqnum2;
  match.length2 = d->plen;
  match.flag = FLAGQUERY | FLAGCOMPLETEMATCH;
  if (CHECKSHOWPALINDROMIC(matchstate))
  {
    match.flag |= FLAGPALINDROMIC;
  }
  match.length1 = d->plen;
  match.distance = 0;
  for (k = state.first[d->seqnum2]; k < state.first[d->seqnum2 + 1]; k++)
  {
    match.position1 = (Uint) state.matches[k].dbstart;
    if (d->processfinal(d->voidMatchstate, &match) != 0)
    {
      return (Sint) -1;
    }
  }
  return 0;
}

static Sint wrap(/*@unused@*/ void *data)
{
  dropbatch();
  if (state.index != NULL)
  {
    vsa_index_close(state.index);
    state.index = NULL;
    state.indexowner = NULL;
  }
  return 0;
}

char vplugingetinterface(Uchar ptrsize, Uchar ifacesize,
                         Vplugininterface *iface)
{
  /* VPLUGINCHECKSIZES, vplugin-interface.h:17-29, without ERROR2 (which
     needs messagespace from the executable) */
  if ((size_t) ptrsize != sizeof(void *) ||
      (size_t) ifacesize != sizeof(Vplugininterface))
  {
    (void) pluginerror("cpridxps_amd: pointer or interface size mismatch");
    return (char) -1;
  }
  iface->vplugininit = init;
  iface->vpluginadddemand = adddemand;
  iface->vpluginparse = parse;
  iface->vpluginsearch = search;
  iface->vpluginwrap = wrap;
  return 0;
}
