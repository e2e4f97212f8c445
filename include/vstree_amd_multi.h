/*
  vstree_amd_multi.h -- the query path on all GPUs of one node, C ABI.

  One process, one host thread per GPU.  The index is replicated in every
  GPU's HBM (uploaded once per device in parallel from the host tables, or
  copied device to device from a replica that exists already); the queries of
  a call are cut into contiguous blocks, one per GPU, and keep their global
  numbers (Vmengine/fquery.c:1010 `onlinequerynumoffset`).  SURVEY.md 8e:

    -complete, -l (MEM), -mum cand   every query is independent
        (Vmengine/fcomplete.c:313-319, Vmengine/fquery.c:468-475): no exchange;
        the lists of the GPUs concatenated in block order are the reference's
        list.
    -mum                             the candidates of ALL queries pass one
        filter (kurtz/cleanMUMcand.c:55-118): each GPU groups its candidates
        by the range of the index their dbstart falls into, range r goes to GPU
        r (peer copies over xGMI), GPU r filters its range given the largest
        right end of the ranges before it; the lists concatenated in GPU order
        are the reference's list (ascending dbstart).
    match counters                   one RCCL all-reduce (sum) over the GPUs.

  What a caller of the reference's engine binds instead of
  findcompletematches / findquerymatches (Vmengine/vmengineexport.h:4-81) when
  it wants all GPUs: integration/vmengine_shim.c does so for VMATCH_GPUS > 1.
  Library: vstree_amd/libvstree_amd_multi.so (needs libvstree_amd.so, librccl).
*/
#ifndef VSTREE_AMD_MULTI_H
#define VSTREE_AMD_MULTI_H

#include "vstree_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct vsa_multi vsa_multi;

/* the host tables uploaded to the first device of the list, the derived
   search tables made there, and every other replica copied from it device to
   device (as vsa_multi_replicate does): all replicas of a set have the same
   tables of the same depth (vsa_index_info.deepprefix), whatever memory each
   device had free.  A device may be named twice (two replicas on one GPU: how
   the tests run on a one-GPU box) */
int vsa_multi_from_tables(const vsa_tables *tables, const int *devices,
                          uint32_t ndevices, vsa_multi **multi);

/* replicas of an index that lives on a device already (built there by
   vsa_index_build / opened by vsa_index_open): vsa_index_clone to every other
   entry of the list.  `first` becomes replica 0 and belongs to the set from
   now on (vsa_multi_close closes it); devices[0] must be its device. */
int vsa_multi_replicate(vsa_index *first, const int *devices,
                        uint32_t ndevices, vsa_multi **multi);

uint32_t vsa_multi_ndevices(const vsa_multi *multi);
vsa_index *vsa_multi_index(vsa_multi *multi, uint32_t replica);
void vsa_multi_close(vsa_multi *multi);

/* 1 if the counters of the last call were summed by RCCL (distinct devices,
   communicators initialised), 0 if on the host (replicas sharing a device) */
int vsa_multi_uses_rccl(const vsa_multi *multi);

#define VSA_MULTI_COMPLETE 0 /* vmatch -complete            findcompletematches */
#define VSA_MULTI_MEM      1 /* vmatch -l L                 findquerymatches    */
#define VSA_MULTI_MUMCAND  2 /* vmatch -mum cand -l L       findquerymatches    */
#define VSA_MULTI_MUM      3 /* vmatch -mum -l L            findquerymatches    */

/*
  One engine call over all replicas.  Queries as for vsa_queries_from_host
  (symbols / start / length of a Multiseq, host memory; any order in the
  buffer; a query that does not lie inside the nsymbols given is refused
  with -2 before anything is uploaded).  The matches come
  back in host memory (free with vsa_multi_free_matches), in the reference's
  order; total = the counters of the whole job (count, sumlength, searches,
  candidates).  Errors as in the single-GPU calls, incl. the reference's
  "patternlength=... must be >= ...=prefixlen" with the matches of the
  queries before the offending one.
*/
int vsa_multi_findmatches(vsa_multi *multi, int mode, uint64_t searchlength,
                          const uint8_t *symbols, uint64_t nsymbols,
                          const uint64_t *start, const uint64_t *length,
                          uint64_t nq, vsa_match **matches, uint64_t *count,
                          vsa_stats *total);
void vsa_multi_free_matches(vsa_match *matches);

/*
  The same engine call on queries that already lie in HBM, with the lists left
  there: blocks[r] = the block of replica r, a vsa_queries on ITS device whose
  vsa_queries_set_offset is the number of its first query in the job (blocks
  in ascending query order, like the split of vsa_multi_findmatches);
  results[r] (ndevices entries, the caller frees them) = the list of replica
  r in its HBM.  -complete, -l, -mum cand: the lists in replica order are the
  reference's list (Vmengine/fcomplete.c:313-319, Vmengine/fquery.c:468-475).
  -mum: results[r] holds the MUMs whose dbstart lies in range r of the index,
  ascending; in replica order: the reference's list (kurtz/cleanMUMcand.c:55-118
  over the candidates of all replicas -- rows of 16 bytes, range r of every
  replica pulled to GPU r by peer copies, filtered there with the largest right
  end of the ranges below it).  Nothing crosses PCIe but the split sizes of
  that exchange (2 * ndevices words per replica) and the four counters, which
  take the one ncclAllReduce when every replica has a GPU of its own.  total:
  the counters of the whole job.  On an error the lists in front of the
  failing replica stay (NULL behind it; a -mum job that failed has none).
  -mum here takes queries below 65 535 symbols and query numbers below 2^47
  (-2 otherwise: vsa_multi_findmatches handles such jobs through records).
*/
int vsa_multi_findmatches_device(vsa_multi *multi, int mode,
                                 uint64_t searchlength,
                                 vsa_queries *const *blocks,
                                 vsa_result **results, vsa_stats *total);

/*
  Host memory to host memory at the rate of the GPUs: one packed pipeline
  (vsa_pipeline_open_packed: page-locked slots the caller packs its reads into
  with vsa_pack_reads, three batches in flight per GPU, upload / search /
  download overlapped) per replica.  The batches of a job are dealt out to
  the replicas in turn and are delivered in the order they were submitted --
  query order, the reference's order for -complete, -l and -mum cand
  (Vmengine/fcomplete.c:313-319, Vmengine/fquery.c:468-475); queryseq counts
  over the whole job.  Calls mirror vsa_pipeline_* (one calling thread):

    vsa_multi_pipeline_open(multi, VSA_MULTI_MUM, 20, 100, 4000000, 65536, &p);
    while (more reads) {
      while (vsa_multi_pipeline_hostrows(p, &rows, &special) == 1)
        vsa_multi_pipeline_next(p, &matches, &n);     -- take a finished batch
      ns = 0; vsa_pack_reads(reads, nq, 100, stride, rows, special, 65536, &ns);
      vsa_multi_pipeline_submit(p, nq, ns);
    }
    while (vsa_multi_pipeline_next(p, &matches, &n) != 1) ...;
    vsa_multi_pipeline_finish(p, lists, counts, &total);      -- -mum only

  -mum: the batches deliver nothing; the candidates stay in the HBM of the
  replica that found them, and vsa_multi_pipeline_finish runs the filter of
  kurtz/cleanMUMcand.c:55-118 over all of them -- grouped by range of the
  index, range r moved to GPU r by peer copies, filtered there -- and leaves
  list r (ascending dbstart; the lists in replica order are the reference's
  list) in page-locked host memory that stays valid until the next finish or
  close: lists[r], counts[r] for r < vsa_multi_ndevices.  The counters of the
  job take the one ncclAllReduce.
*/
typedef struct vsa_multi_pipeline vsa_multi_pipeline;
int vsa_multi_pipeline_open(vsa_multi *multi, int mode, uint64_t searchlength,
                            uint32_t querylength, uint64_t maxqueries,
                            uint64_t maxspecial,
                            vsa_multi_pipeline **pipeline);
int vsa_multi_pipeline_hostrows(vsa_multi_pipeline *pipeline, uint64_t **rows,
                                uint8_t **special);
int vsa_multi_pipeline_submit(vsa_multi_pipeline *pipeline,
                              uint64_t numofqueries, uint64_t numofspecial);
int vsa_multi_pipeline_next(vsa_multi_pipeline *pipeline,
                            const vsa_match **matches, uint64_t *count);
int vsa_multi_pipeline_finish(vsa_multi_pipeline *pipeline,
                              const vsa_match **lists, uint64_t *counts,
                              vsa_stats *total);
void vsa_multi_pipeline_close(vsa_multi_pipeline *pipeline);

/* the same with the reference's delivery model: callbacks on the calling
   thread, in reference order, stop on a non-zero return */
int vsa_multi_findmatches_cb(vsa_multi *multi, int mode,
                             uint64_t searchlength, const uint8_t *symbols,
                             uint64_t nsymbols, const uint64_t *start,
                             const uint64_t *length, uint64_t nq,
                             vsa_processmatch processmatch, void *info);

/*
  findcompletematches for approximate matching (vmatch -complete -e K | -h K,
  vsa_findapproxcompletematches; Vmengine/approxcompl.c:138-199) over all
  replicas: the reads are independent, every replica answers its block, the
  lists are concatenated in block order = the reference's order; the distance
  of a match travels in its querystart field.  VSA_NOT_COVERED (from any
  replica) = nothing was delivered for the whole job.  The reference's
  "threshold=...>=...=patternlen not allowed" ends the job at that read, with
  the matches of the reads before it.
*/
int vsa_multi_findapproxcompletematches(
    vsa_multi *multi, int doedist, uint64_t distvalue, int percent,
    const uint8_t *symbols, uint64_t nsymbols, const uint64_t *start,
    const uint64_t *length, uint64_t nq, vsa_match **matches, uint64_t *count,
    vsa_stats *total);
int vsa_multi_findapproxcompletematches_cb(
    vsa_multi *multi, int doedist, uint64_t distvalue, int percent,
    const uint8_t *symbols, uint64_t nsymbols, const uint64_t *start,
    const uint64_t *length, uint64_t nq, vsa_processmatch processmatch,
    void *info);

#ifdef __cplusplus
}
#endif

#endif
