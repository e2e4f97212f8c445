/*
  Vmengine-side binding of the GPU engine: what a Vmatch maintainer adds.

  It is compiled against the reference's OWN headers (src/include,
  src/Vmengine) and linked into vmatch together with libvstree_amd.so using

      -Wl,--wrap=findcompletematches -Wl,--wrap=findquerymatches
      -Wl,--wrap=findmaximaluniquematches -Wl,--wrap=findsupermax
      -Wl,--wrap=vmatmaxout4 -Wl,--wrap=vmatmaxout12 -Wl,--wrap=vmatmaxout21
      -Wl,--wrap=vmatmaxout31 -Wl,--wrap=findtandems

  (the nine wraps of integration/Makefile) so that every call site in
  src/Vmatch/runquery.c:97-115,149-169 and src/Vmengine/fself.c:203-260
  lands here unchanged.  Exact matching on the
  index goes to the GPU; every other mode (-online, -e/-h/-xdrop, plugin
  index, protein-vs-DNA, ...) is handed to the reference's own function
  (__real_...), so this is a drop-in for the one path and nothing else.

  Matches come back from the GPU in reference order and are reported through
  the reference's own sinks -- processexactquerymatch
  (src/Vmengine/procexqu.c:17) or a Match filled like
  initcompletematchstruct (src/Vmengine/initcompl.c:7) -- on the calling
  thread, so processfinal, SelectBundle plugins, E-values and output
  formatting stay exactly what they were.

  The uploaded index is kept between calls as long as the same Virtualtree
  is passed (vmatch calls the engine once or twice per run: forward and
  reverse-complement queries).
*/
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <time.h>
#include "types.h"
#include "errordef.h"
#include "virtualdef.h"
#include "multidef.h"
#include "match.h"
#include "select.h"
#include "xdropdef.h"
#include "matchstate.h"
#include "mparms.h"
#include "vplugin-interface.h"
#include "cpridx-data.h"
#include "vstree_amd.h"
#include "vstree_amd_multi.h"

/* reference functions this binding keeps using */
Sint initMatchstate(Matchstate *, Virtualtree *, void *, Matchparam *,
                    Bestflag, Uint, Uint, SelectBundle *, Uint, void *,
                    Currentdirection, BOOL, Processfinalfunction, Evalues *,
                    BOOL);
Sint processexactquerymatch(void *info, Uint l, Uint i, Uint queryseq,
                            Uint querystart);
void initcompletematchstruct(Match *match, Uint seqnum2, Uint plen,
                             BOOL ispalindromic);
Uint getqueryseppos(Multiseq *multiseq);

Sint __real_findcompletematches(Virtualtree *, char *, Queryinfo *, BOOL,
                                BOOL, Matchparam *, Bestflag, Uint, Uint,
                                SelectBundle *, void *, Currentdirection,
                                Processfinalfunction, Vpluginbundle *,
                                Cpridxpatsearchdata *, Evalues *, BOOL);
Sint __real_findquerymatches(Virtualtree *, Uint, Queryinfo *, BOOL, BOOL,
                             BOOL, Matchparam *, Bestflag, Uint, Uint,
                             SelectBundle *, void *, Currentdirection, BOOL,
                             Processfinalfunction, Evalues *, BOOL);
Sint __real_findmaximaluniquematches(Virtualtree *, Uint, Uint, void *,
                                     void *, Outputfunction);

static vsa_index *gpuindex = NULL;
/* the replicas of VMATCH_GPUS / VMATCH_GPU_DEVICES (see multidevices) */
static vsa_multi *gpumulti = NULL;
static Virtualtree *gpumultiowner = NULL;
static Virtualtree *gpuindexowner = NULL;

static int usegpu(void)
{
  const char *e = getenv("VMATCH_GPU");
  return e == NULL || strcmp(e, "0") != 0;
}

/* VMATCH_GPU_TRACE=1: one line on stderr per engine call that ran on the
   GPU (tests/test_gpu_dropin.py uses it to make sure that a passing
   comparison is not the CPU fallback in disguise) */
static void trace(const char *what)
{
  const char *e = getenv("VMATCH_GPU_TRACE");
  if (e != NULL && strcmp(e, "0") != 0)
  {
    fprintf(stderr, "vstree_amd: %s on the GPU\n", what);
  }
}

static double nowseconds(void)
{
  struct timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
}

static void tracetime(const char *what, double since)
{
  const char *e = getenv("VMATCH_GPU_TRACE");
  if (e != NULL && strcmp(e, "0") != 0)
  {
    fprintf(stderr, "vstree_amd: %s took %.3f s\n", what,
            nowseconds() - since);
  }
}

static int gpufail(void)
{
  ERROR1("%s", vsa_messagespace());
  return -1;
}

static int getgpuindex(Virtualtree *virtualtree, int needbwt,
                       vsa_index **index)
{
  vsa_tables t;

  if (gpuindex != NULL && gpuindexowner == virtualtree)
  {
    *index = gpuindex;
    return 0;
  }
  if (gpumulti != NULL && gpumultiowner == virtualtree &&
      (!needbwt || virtualtree->bwttab == NULL))
  {
    /* the replicas are there (an exact call ran on all GPUs): replica 0 serves
       the single-GPU paths, the index is not uploaded a second time */
    *index = vsa_multi_index(gpumulti, 0);
    return 0;
  }
  if (gpuindex != NULL)
  {
    vsa_index_close(gpuindex);
    gpuindex = NULL;
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
  t.bwt = needbwt ? virtualtree->bwttab : NULL;
  if (HASINDEXEDQUERIES(&virtualtree->multiseq))
  {
    t.hasindexedqueries = 1;
    t.querysepposition = getqueryseppos(&virtualtree->multiseq);
  }
  {
    const double t0 = nowseconds();
    if (vsa_index_from_tables(&t, 0, &gpuindex) != 0)
    {
      return gpufail();
    }
    tracetime("index upload (tables to HBM + derived search tables)", t0);
  }
  gpuindexowner = virtualtree;
  *index = gpuindex;
  return 0;
}

/* the query Multiseq as (start, length) pairs, kurtz-basic/multiseq.c:129 */
static int getquerybounds(Multiseq *multiseq, uint64_t **start,
                          uint64_t **length)
{
  Uint i, nq = multiseq->numofsequences;

  *start = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (nq + 1));
  *length = (uint64_t *) malloc(sizeof(uint64_t) * (size_t) (nq + 1));
  if (*start == NULL || *length == NULL)
  {
    ERROR0("out of memory");
    return -1;
  }
  for (i = 0; i < nq; i++)
  {
    Uint s = (i == 0) ? 0 : multiseq->markpos.spaceUint[i - 1] + 1;
    Uint e = (i == nq - 1) ? multiseq->totallength
                           : multiseq->markpos.spaceUint[i];
    (*start)[i] = s;
    (*length)[i] = e - s;
  }
  return 0;
}

static int getgpuqueries(Multiseq *multiseq, BOOL rcmode,
                         vsa_queries **queries)
{
  uint64_t *start, *length;
  int rc;

  if (getquerybounds(multiseq, &start, &length) != 0)
  {
    return -1;
  }
  rc = vsa_queries_from_host(rcmode ? multiseq->rcsequence
                                    : multiseq->sequence,
                             multiseq->totallength, start, length,
                             multiseq->numofsequences, 0, queries);
  free(start);
  free(length);
  return rc != 0 ? gpufail() : 0;
}

/* ---- all GPUs of the node (include/vstree_amd_multi.h) -------------------
   VMATCH_GPUS=N        replicas on devices 0 .. N-1
   VMATCH_GPU_DEVICES=a,b,c   replicas on exactly these devices (a device may
                        be named twice: how a one-GPU box rehearses it)
   Exact -complete, -l, -mum cand and -mum then run on all replicas: the
   queries are cut into blocks, the match lists come back in the reference's
   order and go through the same sinks as in the single-GPU case. */
#define MAXGPUS 64

static uint32_t multidevices(int *devices)
{
  const char *list = getenv("VMATCH_GPU_DEVICES"), *n = getenv("VMATCH_GPUS");
  uint32_t count = 0;

  const int have = vsa_device_count();

  if (list != NULL && *list != '\0')
  {
    while (*list != '\0')
    {
      char *end = NULL;
      const long id = strtol(list, &end, 10);
      if (end == list || id < 0 || id >= have || count >= MAXGPUS)
      {
        /* no digits here (a token such as "a", ";"), a device that does not
           exist, or more replicas than the table holds: say so and stop --
           never guess a device */
        fprintf(stderr, "%s: VMATCH_GPU_DEVICES=\"%s\": expected up to %d "
                "device numbers below %d, separated by commas\n",
                "vmatch", getenv("VMATCH_GPU_DEVICES"), MAXGPUS, have);
        exit(EXIT_FAILURE);
      }
      devices[count++] = (int) id;
      list = end;
      while (*list == ',' || *list == ' ')
      {
        list++;
      }
    }
  } else if (n != NULL && atoi(n) > 1)
  {
    if (atoi(n) > have || atoi(n) > MAXGPUS)
    {
      fprintf(stderr, "%s: VMATCH_GPUS=%s, but this node shows %d GPU(s)\n",
              "vmatch", n, have);
      exit(EXIT_FAILURE);
    }
    for (count = 0; count < (uint32_t) atoi(n); count++)
    {
      devices[count] = (int) count;
    }
  }
  return count > 1 ? count : 0;
}

static int getgpumulti(Virtualtree *virtualtree, const int *devices,
                       uint32_t ndevices, vsa_multi **multi)
{
  vsa_tables t;

  if (gpumulti != NULL && gpumultiowner == virtualtree)
  {
    *multi = gpumulti;
    return 0;
  }
  if (gpumulti != NULL)
  {
    vsa_multi_close(gpumulti);
    gpumulti = NULL;
  }
  if (gpuindex != NULL)
  {
    /* (replica 0 takes its place: one copy of the index per device) */
    vsa_index_close(gpuindex);
    gpuindex = NULL;
    gpuindexowner = NULL;
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
  if (vsa_multi_from_tables(&t, devices, ndevices, &gpumulti) != 0)
  {
    return gpufail();
  }
  gpumultiowner = virtualtree;
  *multi = gpumulti;
  return 0;
}

/* Matchparam.maxdist.distinterpretation (qualint.h) -> the `percent` argument
   of vsa_findapproxcompletematches: 0 K, 1 Kp, 2 Kb */
static int thresholdkind(const Matchparam *matchparam)
{
  return matchparam->maxdist.distinterpretation == Qualpercentaway
             ? 1
             : (matchparam->maxdist.distinterpretation == Qualbestof ? 2 : 0);
}

/* one exact engine call on all replicas; the matches go to `sink` */
static int runonallgpus(Virtualtree *virtualtree, Multiseq *queries,
                        BOOL rcmode, int mode, Uint searchlength,
                        Uint queryspeedup, const int *devices,
                        uint32_t ndevices, vsa_processmatch sink, void *info)
{
  vsa_multi *multi;
  uint64_t *start, *length;
  uint32_t r;
  int rc;

  if (getgpumulti(virtualtree, devices, ndevices, &multi) != 0 ||
      getquerybounds(queries, &start, &length) != 0)
  {
    return -2;
  }
  for (r = 0; r < vsa_multi_ndevices(multi); r++)
  {
    (void) vsa_index_set_queryspeedup(vsa_multi_index(multi, r),
                                      (uint32_t) queryspeedup);
  }
  rc = vsa_multi_findmatches_cb(multi, mode, searchlength,
                                rcmode ? queries->rcsequence
                                       : queries->sequence,
                                queries->totallength, start, length,
                                queries->numofsequences, sink, info);
  free(start);
  free(length);
  return rc;
}

/* ---- vmatch -complete -q ------------------------------------------------ */

/* length of query sequence number seqnum (kurtz-basic/multiseq.c:129-166) */
static Uint querylength(Matchstate *matchstate, Uint seqnum)
{
  Multiseq *multiseq = matchstate->queryinfo->multiseq;
  Uint s = (seqnum == 0) ? 0 : multiseq->markpos.spaceUint[seqnum - 1] + 1;
  Uint e = (seqnum == multiseq->numofsequences - 1)
               ? multiseq->totallength
               : multiseq->markpos.spaceUint[seqnum];
  return e - s;
}

static int completesink(void *info, const vsa_match *m)
{
  Matchstate *matchstate = (Matchstate *) info;
  Match match;

  initcompletematchstruct(&match, (Uint) m->queryseq, (Uint) m->length,
                          CHECKSHOWPALINDROMIC(matchstate) ? True : False);
  match.length1 = (Uint) m->length;
  match.distance = 0;
  match.position1 = (Uint) m->dbstart;
  return matchstate->processfinal(matchstate, &match) != 0 ? 1 : 0;
}

/* edistprocessstartpos / hammingprocessstartpos, approxcompl.c:14-83 */
typedef struct
{
  Matchstate *matchstate;
  int hamming;
} Approxsink;

static int approxsink(void *info, const vsa_match *m)
{
  Approxsink *sink = (Approxsink *) info;
  Match match;

  initcompletematchstruct(&match, (Uint) m->queryseq,
                          /* plen */ 0,
                          CHECKSHOWPALINDROMIC(sink->matchstate) ? True
                                                                 : False);
  match.length2 = querylength(sink->matchstate, (Uint) m->queryseq);
  match.position1 = (Uint) m->dbstart;
  match.length1 = (Uint) m->length;
  match.distance = sink->hamming ? -(Sint) m->querystart
                                 : (Sint) m->querystart;
  return sink->matchstate->processfinal(sink->matchstate, &match) != 0 ? 1
                                                                       : 0;
}

Sint __wrap_findcompletematches(Virtualtree *virtualtree,
                                char *indexormatchfile, Queryinfo *queryinfo,
                                BOOL rcmode, BOOL online,
                                Matchparam *matchparam, Bestflag bestflag,
                                Uint shownoevalue, Uint showselfpalindromic,
                                SelectBundle *selectbundle,
                                void *procmultiseq,
                                Currentdirection currentdirection,
                                Processfinalfunction processfinal,
                                Vpluginbundle *cpridxpatsearchbundle,
                                Cpridxpatsearchdata *cpridxpatsearchdata,
                                Evalues *evalues, BOOL domatchbuffering)
{
  Matchstate matchstate;
  vsa_index *index;
  vsa_queries *queries;
  int rc;

  const int approx = !MPARMEXACTMATCH(&matchparam->maxdist);

  /* approximate matching: -e K / -h K, the percent forms and "best of"
     (qualint.h:7-13, initcompl.c:52-77) go to the GPU */
  if (!usegpu() || online ||
      cpridxpatsearchbundle->handle != NULL || virtualtree->suftab == NULL ||
      virtualtree->bcktab == NULL || virtualtree->lcptab == NULL)
  {
    return __real_findcompletematches(
        virtualtree, indexormatchfile, queryinfo, rcmode, online, matchparam,
        bestflag, shownoevalue, showselfpalindromic, selectbundle,
        procmultiseq, currentdirection, processfinal, cpridxpatsearchbundle,
        cpridxpatsearchdata, evalues, domatchbuffering);
  }
  if (initMatchstate(&matchstate, virtualtree, (void *) queryinfo, matchparam,
                     bestflag, shownoevalue, showselfpalindromic,
                     selectbundle, 0, procmultiseq, currentdirection, False,
                     processfinal, evalues, domatchbuffering) != 0)
  {
    return (Sint) -1;
  }
  cpridxpatsearchdata->voidMatchstate = NULL;
  {
    int devices[MAXGPUS];
    const uint32_t ndevices = multidevices(devices);
    if (ndevices > 0 && approx)
    {
      /* the reads are independent: every replica answers a block of them
         (vsa_multi_findapproxcompletematches) */
      vsa_multi *multi;
      uint64_t *start, *length;
      Multiseq *qseq = queryinfo->multiseq;
      Approxsink sink;

      sink.matchstate = &matchstate;
      sink.hamming = MPARMHAMMINGMATCH(&matchparam->maxdist) ? 1 : 0;
      if (getgpumulti(virtualtree, devices, ndevices, &multi) != 0 ||
          getquerybounds(qseq, &start, &length) != 0)
      {
        return (Sint) -2;
      }
      rc = vsa_multi_findapproxcompletematches_cb(
          multi, MPARMEDISTMATCH(&matchparam->maxdist) ? 1 : 0,
          (uint64_t) matchparam->maxdist.distvalue,
          thresholdkind(matchparam),
          rcmode ? qseq->rcsequence : qseq->sequence, qseq->totallength,
          start, length, qseq->numofsequences, approxsink, &sink);
      free(start);
      free(length);
      trace("approximate complete matches, all replicas");
      if (rc == VSA_NOT_COVERED)
      {
        /* nothing has been reported yet: the reference's own function takes
           the whole batch */
        return __real_findcompletematches(
            virtualtree, indexormatchfile, queryinfo, rcmode, online,
            matchparam, bestflag, shownoevalue, showselfpalindromic,
            selectbundle, procmultiseq, currentdirection, processfinal,
            cpridxpatsearchbundle, cpridxpatsearchdata, evalues,
            domatchbuffering);
      }
      if (rc != 0)
      {
        if (rc != -1)
        {
          (void) gpufail();
        }
        return (Sint) -2;
      }
      return 0;
    }
    if (ndevices > 0)
    {
      rc = runonallgpus(virtualtree, queryinfo->multiseq, rcmode,
                        VSA_MULTI_COMPLETE, 0, 2, devices, ndevices,
                        completesink, &matchstate);
      trace("complete matches, all replicas");
      if (rc != 0)
      {
        if (rc != -1)
        {
          (void) gpufail();
        }
        return (Sint) -2;
      }
      return 0;
    }
  }
  if (getgpuindex(virtualtree, 0, &index) != 0 ||
      getgpuqueries(queryinfo->multiseq, rcmode, &queries) != 0)
  {
    return (Sint) -2;
  }
  if (approx)
  {
    Approxsink sink;
    sink.matchstate = &matchstate;
    sink.hamming = MPARMHAMMINGMATCH(&matchparam->maxdist) ? 1 : 0;
    rc = vsa_findapproxcompletematches_cb(
        index, queries, MPARMEDISTMATCH(&matchparam->maxdist) ? 1 : 0,
        (uint64_t) matchparam->maxdist.distvalue,
        thresholdkind(matchparam),
        approxsink, &sink);
    if (rc == VSA_NOT_COVERED)
    {
      /* nothing has been reported yet: the reference's own function takes
         the whole batch */
      vsa_queries_free(queries);
      return __real_findcompletematches(
          virtualtree, indexormatchfile, queryinfo, rcmode, online,
          matchparam, bestflag, shownoevalue, showselfpalindromic,
          selectbundle, procmultiseq, currentdirection, processfinal,
          cpridxpatsearchbundle, cpridxpatsearchdata, evalues,
          domatchbuffering);
    }
  } else
  {
    rc = vsa_findcompletematches_cb(index, queries, completesink,
                                    &matchstate);
  }
  trace(approx ? "approximate complete matches" : "complete matches");
  vsa_queries_free(queries);
  if (rc != 0)
  {
    if (rc != -1) /* -1: the sink stopped the run, message is the sink's */
    {
      (void) gpufail();
    }
    return (Sint) -2;
  }
  return 0;
}

/* ---- vmatch [-mum [cand]] -l L -q --------------------------------------- */

static int querysink(void *info, const vsa_match *m)
{
  return processexactquerymatch(info, (Uint) m->length, (Uint) m->dbstart,
                                (Uint) m->queryseq, (Uint) m->querystart)
                 != 0 ? 1 : 0;
}

Sint __wrap_findquerymatches(Virtualtree *virtualtree,
                             Uint onlinequerynumoffset, Queryinfo *queryinfo,
                             BOOL domaximaluniquematch,
                             BOOL domaximaluniquematchcandidates, BOOL rcmode,
                             Matchparam *matchparam, Bestflag bestflag,
                             Uint shownoevalue, Uint showselfpalindromic,
                             SelectBundle *selectbundle, void *procmultiseq,
                             Currentdirection currentdirection,
                             BOOL revmposorder,
                             Processfinalfunction processfinal,
                             Evalues *evalues, BOOL domatchbuffering)
{
  Matchstate matchstate;
  vsa_index *index;
  vsa_queries *queries;
  int rc;

  /* -qspeedup 0 and 2 (the documented levels, Vmatch/parsevm.c:781-784) are
     reproduced in their own order; the undocumented 3..5 stay on the CPU */
  if (!usegpu() || !MPARMEXACTMATCH(&matchparam->maxdist) ||
      matchparam->xdropbelowscore != UNDEFXDROPBELOWSCORE ||
      (matchparam->queryspeedup != 0 && matchparam->queryspeedup != 2) ||
      virtualtree->suftab == NULL || virtualtree->bcktab == NULL ||
      virtualtree->lcptab == NULL)
  {
    return __real_findquerymatches(
        virtualtree, onlinequerynumoffset, queryinfo, domaximaluniquematch,
        domaximaluniquematchcandidates, rcmode, matchparam, bestflag,
        shownoevalue, showselfpalindromic, selectbundle, procmultiseq,
        currentdirection, revmposorder, processfinal, evalues,
        domatchbuffering);
  }
  if (initMatchstate(&matchstate, virtualtree, (void *) queryinfo, matchparam,
                     bestflag, shownoevalue, showselfpalindromic,
                     selectbundle, onlinequerynumoffset, procmultiseq,
                     currentdirection, revmposorder, processfinal, evalues,
                     domatchbuffering) != 0)
  {
    return (Sint) -1;
  }
  {
    int devices[MAXGPUS];
    const uint32_t ndevices = multidevices(devices);
    if (ndevices > 0)
    {
      rc = runonallgpus(virtualtree, queryinfo->multiseq, rcmode,
                        !domaximaluniquematch
                            ? VSA_MULTI_MEM
                            : (domaximaluniquematchcandidates
                                   ? VSA_MULTI_MUMCAND
                                   : VSA_MULTI_MUM),
                        matchparam->seedlength, matchparam->queryspeedup,
                        devices, ndevices, querysink, &matchstate);
      trace("query matches, all replicas");
      if (rc != 0)
      {
        if (rc != -1)
        {
          (void) gpufail();
        }
        return (Sint) -1;
      }
      return 0;
    }
  }
  if (getgpuindex(virtualtree, 0, &index) != 0 ||
      getgpuqueries(queryinfo->multiseq, rcmode, &queries) != 0)
  {
    return (Sint) -1;
  }
  (void) vsa_index_set_queryspeedup(index,
                                    (uint32_t) matchparam->queryspeedup);
  {
    const double t0 = nowseconds();
    rc = vsa_findquerymatches_cb(index, queries,
                                 domaximaluniquematch ? 1 : 0,
                                 domaximaluniquematchcandidates ? 1 : 0,
                                 matchparam->seedlength, querysink,
                                 &matchstate);
    tracetime("engine call + delivery of the matches to processfinal", t0);
  }
  trace("query matches");
  vsa_queries_free(queries);
  if (rc != 0)
  {
    if (rc != -1)
    {
      (void) gpufail();
    }
    return (Sint) -1;
  }
  return 0;
}

/* ---- vmatch -mum -l L on an index that contains its queries ------------- */

typedef struct
{
  void *outinfo;
  Outputfunction output;
} Selfsink;

static int selfsink(void *info, const vsa_match *m)
{
  Selfsink *s = (Selfsink *) info;

  return s->output(s->outinfo, (Uint) m->length, (Uint) m->dbstart,
                   (Uint) m->queryseq) != 0 ? 1 : 0;
}

Sint __wrap_findmaximaluniquematches(Virtualtree *virtualtree,
                                     Uint numberofprocessors,
                                     Uint searchlength, void *repeatgapspec,
                                     void *outinfo, Outputfunction output)
{
  vsa_index *index;
  Selfsink s;
  int rc;

  if (!usegpu() || virtualtree->suftab == NULL ||
      virtualtree->lcptab == NULL || virtualtree->bwttab == NULL)
  {
    return __real_findmaximaluniquematches(virtualtree, numberofprocessors,
                                           searchlength, repeatgapspec,
                                           outinfo, output);
  }
  if (gpuindexowner == virtualtree && gpuindex != NULL)
  {
    /* make sure the cached copy carries bwt */
    vsa_index_close(gpuindex);
    gpuindex = NULL;
  }
  if (getgpuindex(virtualtree, 1, &index) != 0)
  {
    return (Sint) -3;
  }
  s.outinfo = outinfo;
  s.output = output;
  rc = vsa_findmaximaluniquematches_cb(index, searchlength, selfsink, &s);
  trace("maximal unique matches of the index");
  if (rc != 0)
  {
    if (rc != -1)
    {
      (void) gpufail();
    }
    return (Sint) -3;
  }
  return 0;
}

/* ---- vmatch -l L IDX (maximal repeats) and vmatch -supermax -l L IDX ------ */

/* the traversals findselfmatches dispatches to (Vmengine/fself.c:215-222):
   findsupermax (fsuper.c:142) and, behind vmatmaxoutgeneric
   (Vmengine/vmatgen.c), the instance of vmatfind.c for the alphabet size */
Sint __real_findsupermax(Virtualtree *, Uint, Uint, void *, void *,
                         Outputfunction);
Sint __real_vmatmaxout4(Virtualtree *, Uint, Uint, void *, void *,
                        Outputfunction);
Sint __real_vmatmaxout12(Virtualtree *, Uint, Uint, void *, void *,
                         Outputfunction);
Sint __real_vmatmaxout21(Virtualtree *, Uint, Uint, void *, void *,
                         Outputfunction);
Sint __real_vmatmaxout31(Virtualtree *, Uint, Uint, void *, void *,
                         Outputfunction);

typedef Sint (*Selftraversal)(Virtualtree *, Uint, Uint, void *, void *,
                              Outputfunction);

static Sint gpuselfmatches(int supermax, Selftraversal real,
                           Virtualtree *virtualtree, Uint numberofprocessors,
                           Uint searchlength, void *repeatgapspec,
                           void *outinfo, Outputfunction output)
{
  vsa_index *index;
  Selfsink s;
  int rc;

  if (!usegpu() || virtualtree->suftab == NULL ||
      virtualtree->lcptab == NULL || virtualtree->bwttab == NULL ||
      (supermax && HASINDEXEDQUERIES(&virtualtree->multiseq)))
  {
    return real(virtualtree, numberofprocessors, searchlength, repeatgapspec,
                outinfo, output);
  }
  if (gpuindexowner == virtualtree && gpuindex != NULL)
  {
    vsa_index_close(gpuindex); /* make sure the cached copy carries bwt */
    gpuindex = NULL;
  }
  if (getgpuindex(virtualtree, 1, &index) != 0)
  {
    return (Sint) -3;
  }
  s.outinfo = outinfo;
  s.output = output;
  rc = supermax
           ? vsa_findsupermaximalrepeats_cb(index, searchlength, selfsink, &s)
           : vsa_findmaximalrepeats_cb(index, searchlength, selfsink, &s);
  if (rc == VSA_NOT_COVERED)
  {
    return real(virtualtree, numberofprocessors, searchlength, repeatgapspec,
                outinfo, output);
  }
  trace(supermax ? "supermaximal repeats" : "maximal repeats");
  if (rc != 0)
  {
    if (rc != -1)
    {
      (void) gpufail();
    }
    return (Sint) -3;
  }
  return 0;
}

Sint __wrap_findsupermax(Virtualtree *virtualtree, Uint numberofprocessors,
                         Uint searchlength, void *repeatgapspec,
                         void *outinfo, Outputfunction output)
{
  return gpuselfmatches(1, __real_findsupermax, virtualtree,
                        numberofprocessors, searchlength, repeatgapspec,
                        outinfo, output);
}

#define WRAPMAXOUT(I)                                                         \
  Sint __wrap_vmatmaxout##I(Virtualtree *virtualtree,                         \
                            Uint numberofprocessors, Uint searchlength,       \
                            void *repeatgapspec, void *outinfo,               \
                            Outputfunction output)                            \
  {                                                                           \
    return gpuselfmatches(0, __real_vmatmaxout##I, virtualtree,               \
                          numberofprocessors, searchlength, repeatgapspec,    \
                          outinfo, output);                                   \
  }
WRAPMAXOUT(4)
WRAPMAXOUT(12)
WRAPMAXOUT(21)
WRAPMAXOUT(31)

/* ---- vmatch -tandem -l L IDX ---------------------------------------------- */

/* findtandems (Vmengine/ftandem.c:261-304) builds a Matchstate and hands
   every repeat to processfinal (OUTTANDEM :34-42): the same here */
Sint __real_findtandems(Virtualtree *, Matchparam *, Bestflag, Uint, Uint,
                        SelectBundle *, void *, Processfinalfunction,
                        Evalues *, BOOL);

static int tandemsink(void *info, const vsa_match *m)
{
  Matchstate *matchstate = (Matchstate *) info;
  Match match;

  match.distance = 0;
  match.flag = 0;
  match.seqnum2 = UNDEFSEQNUM2(&matchstate->virtualtree->multiseq);
  match.length1 = match.length2 = (Uint) m->length;
  match.position1 = (Uint) m->dbstart;
  match.position2 = (Uint) m->queryseq;
  return matchstate->processfinal(matchstate, &match) != 0 ? 1 : 0;
}

Sint __wrap_findtandems(Virtualtree *virtualtree, Matchparam *matchparam,
                        Bestflag bestflag, Uint shownoevalue,
                        Uint showselfpalindromic, SelectBundle *selectbundle,
                        void *procmultiseq, Processfinalfunction processfinal,
                        Evalues *evalues, BOOL domatchbuffering)
{
  Matchstate matchstate;
  vsa_index *index;
  int rc;

  if (!usegpu() || virtualtree->suftab == NULL ||
      virtualtree->lcptab == NULL ||
      virtualtree->multiseq.sequence == NULL ||
      HASINDEXEDQUERIES(&virtualtree->multiseq))
  {
    return __real_findtandems(virtualtree, matchparam, bestflag, shownoevalue,
                              showselfpalindromic, selectbundle, procmultiseq,
                              processfinal, evalues, domatchbuffering);
  }
  if (initMatchstate(&matchstate, virtualtree, NULL, matchparam, bestflag,
                     shownoevalue, showselfpalindromic, selectbundle, 0,
                     procmultiseq, DirectionForward, False, processfinal,
                     evalues, domatchbuffering) != 0)
  {
    return (Sint) -1;
  }
  if (getgpuindex(virtualtree, 0, &index) != 0)
  {
    return (Sint) -1;
  }
  rc = vsa_findtandems_cb(index, matchparam->userdefinedleastlength,
                          tandemsink, &matchstate);
  trace("tandem repeats");
  if (rc != 0)
  {
    if (rc != -1)
    {
      (void) gpufail();
    }
    return (Sint) -2;
  }
  return 0;
}
