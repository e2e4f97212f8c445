/*
  The reference's delivery model on top of the batched GPU engine (host side,
  plain C): Vmengine reports every match by calling a function
  (Processfinalfunction, include/match.h:232; reached through
  processexactquerymatch, Vmengine/procexqu.c:17-64, or
  processfinalexactmatchinterval, Vmengine/exactcompl.c:142-166) from the
  one calling thread, and stops as soon as that function returns non-zero.
  Here the whole batch is matched on the GPU first; the match list comes back
  in reference order and is then replayed to the callback on the calling
  thread.
*/
#include <stdlib.h>
#include <stdio.h>
#include "vstree_amd.h"

char *vsa_errbuf(void);

static int replay(vsa_result *result, int searchrc,
                  vsa_processmatch processmatch, void *info)
{
  uint64_t i, n;
  vsa_match *m = NULL;
  int rc = 0;

  if (result == NULL)
  {
    return searchrc;
  }
  n = vsa_result_count(result);
  if (n > 0)
  {
    m = (vsa_match *) malloc((size_t) n * sizeof(vsa_match));
    if (m == NULL)
    {
      snprintf(vsa_errbuf(), 1024, "out of memory for %lu matches",
               (unsigned long) n);
      vsa_result_free(result);
      return -101;
    }
    rc = vsa_result_fetch(result, m, n);
  }
  vsa_result_free(result);
  for (i = 0; rc == 0 && i < n; i++)
  {
    if (processmatch(info, m + i) != 0)
    {
      rc = -1; /* exactcompl.c:160-163, procexqu.c:61-64 */
    }
  }
  free(m);
  /* an engine error (e.g. a query shorter than prefixlength) surfaces after
     the matches found before it, as in the reference */
  return rc != 0 ? rc : searchrc;
}

int vsa_findcompletematches_cb(const vsa_index *index,
                               const vsa_queries *queries,
                               vsa_processmatch processmatch, void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findcompletematches(index, queries, &result);

  return replay(result, rc, processmatch, info);
}

int vsa_findapproxcompletematches_cb(const vsa_index *index,
                                     const vsa_queries *queries, int doedist,
                                     uint64_t distvalue, int percent,
                                     vsa_processmatch processmatch,
                                     void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findapproxcompletematches(index, queries, doedist, distvalue,
                                         percent, &result);

  return replay(result, rc, processmatch, info);
}

int vsa_findquerymatches_cb(const vsa_index *index,
                            const vsa_queries *queries,
                            int domaximaluniquematch,
                            int domaximaluniquematchcandidates,
                            uint64_t searchlength,
                            vsa_processmatch processmatch, void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findquerymatches(index, queries, domaximaluniquematch,
                                domaximaluniquematchcandidates, searchlength,
                                &result);

  return replay(result, rc, processmatch, info);
}

int vsa_findmaximaluniquematches_cb(const vsa_index *index,
                                    uint64_t searchlength,
                                    vsa_processmatch processmatch, void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findmaximaluniquematches(index, searchlength, &result);

  return replay(result, rc, processmatch, info);
}

int vsa_findsupermaximalrepeats_cb(const vsa_index *index,
                                   uint64_t searchlength,
                                   vsa_processmatch processmatch, void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findsupermaximalrepeats(index, searchlength, &result);

  return replay(result, rc, processmatch, info);
}

int vsa_findmaximalrepeats_cb(const vsa_index *index, uint64_t searchlength,
                              vsa_processmatch processmatch, void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findmaximalrepeats(index, searchlength, &result);

  return replay(result, rc, processmatch, info);
}

int vsa_findtandems_cb(const vsa_index *index, uint64_t searchlength,
                       vsa_processmatch processmatch, void *info)
{
  vsa_result *result = NULL;
  int rc = vsa_findtandems(index, searchlength, &result);

  return replay(result, rc, processmatch, info);
}
