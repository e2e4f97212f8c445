/*
  vstree_amd.h -- C ABI of the MI355X-native Vmengine query path.

  One shared library (vstree_amd/libvstree_amd.so, HIP for gfx950 inside)
  replaces the reference's CPU implementation of

    vmatch -complete -q Q IDX          exact complete matches
    vmatch -l L -q Q IDX               maximal exact matches (MEM)
    vmatch -mum [cand] -l L -q Q IDX   maximal unique matches / candidates
    vmatch -mum -l L IDX               MUMs when the queries are in the index

  on the enhanced suffix array mkvtree writes.  Everything here is plain C:
  pointers, sizes, opaque handles; no HIP or torch types.  Each entry point
  names the reference interface it stands in for (paths relative to
  /root/reference/src); INTEGRATION.md shows the few lines of C a Vmatch
  maintainer adds so that findcompletematches / findquerymatches /
  findmaximaluniquematches of Vmengine/vmengineexport.h:4-81 call these.

  Conventions (the reference's, include/errordef.h:45-82): functions return
  0 on success and a negative code on error; the message is then available
  from vsa_messagespace().  All calls on one vsa_index must come from one
  thread at a time, like the reference's engine.
*/
#ifndef VSTREE_AMD_H
#define VSTREE_AMD_H

#include <stdint.h>
#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VSA_SEPARATOR 255u /* include/chardef.h:19 */
#define VSA_WILDCARD  254u /* include/chardef.h:25 */
#define VSA_UNDEFBWT  253u /* include/chardef.h:31,50 */
#define VSA_NO_SUBST  0xFFFFFFFFu

/* ---- errors: messagespace(), include/errordef.h:13 -------------------- */

const char *vsa_messagespace(void);

/* ---- the index: Virtualtree, include/virtualdef.h:186-219 ------------- */

/*
  Raw tables of one index exactly as mkvtree -allout writes them and
  mapvirtualtreeifyoucan (kurtz-basic/readvirt.c:776-907) maps them: host
  pointers, host endianness, no headers.  integersize is the bit width of
  the entries of suf, bck and llv (32 or 64, the `integersize=` line of the
  .prj file, Mkvtree/mkvprocess.c:403-504).  bwt may be NULL unless
  vsa_findmaximaluniquematches is used.
*/
typedef struct
{
  uint64_t totallength;     /* multiseq.totallength                     */
  uint32_t prefixlength;    /* Virtualtree.prefixlength                 */
  uint32_t numofchars;      /* alpha.mapsize - 1 (4 for DNA)            */
  uint32_t integersize;     /* 32 or 64                                 */
  uint64_t largelcpvalues;  /* pairs in llv                             */
  const uint8_t *tis;       /* [totallength]   alphabet-mapped text     */
  const void *suf;          /* [totallength+1] suffix array             */
  const uint8_t *lcp;       /* [totallength+1] min(lcp, 255)            */
  const void *llv;          /* [2*largelcpvalues] (index, value) pairs  */
  const void *bck;          /* [2*numofchars^prefixlength] (left, mid)  */
  const uint8_t *bwt;       /* [totallength+1] or NULL                  */
  /* indexes built with mkvtree -db G -q Q: position of the separator
     between database and queries (getqueryseppos,
     kurtz-basic/multiseq-adv.c:1005); ignored if hasindexedqueries == 0 */
  uint64_t querysepposition;
  int hasindexedqueries;
} vsa_tables;

typedef struct vsa_index vsa_index; /* ESA resident in one GPU's HBM */

/* uploads the tables to HIP device `device`; the host tables are not
   referenced after return */
int vsa_index_from_tables(const vsa_tables *tables, int device,
                          vsa_index **index);

/* reads indexname.prj and maps indexname.{tis,suf,lcp,llv,bck,bwt} like
   mapvirtualtreeifyoucan(…, TISTAB|SUFTAB|LCPTAB|BCKTAB[|BWTTAB])
   (kurtz-basic/readvirt.c:776, demand: Vmatch/mapdemand.c:174-191) */
int vsa_index_open(const char *indexname, int device, vsa_index **index);

void vsa_index_close(vsa_index *index);

/* A replica of an index on another HIP device of the node (or on the same
   one): every table, the derived search tables included, is copied device to
   device -- over xGMI between two GPUs -- and nothing is built again.  The
   multi-GPU entry points (include/vstree_amd_multi.h) replicate with it. */
int vsa_index_clone(const vsa_index *index, int device, vsa_index **clone);

typedef struct
{
  uint64_t totallength, numofcodes, largelcpvalues, device_bytes;
  uint32_t prefixlength, numofchars, device_integersize;
  int device, hasindexedqueries, hasbwt;
  /* symbols of the derived deep bucket table (esa8/slot16), 0 if
     this index has none and is searched probe for probe like the reference */
  uint32_t deepprefix;
} vsa_index_info;

int vsa_index_getinfo(const vsa_index *index, vsa_index_info *info);

/*
  Builds the tables on the GPU from an alphabet-mapped text that is already
  on the host (symbols 0..numofchars-1, VSA_WILDCARD, VSA_SEPARATOR), with
  the semantics of mkvtree -pl <prefixlength> -tis -suf -lcp -bck -bwt
  (Mkvtree/mkvprocess.c:875-1089; suffix order Mkvtree/bese.c:27-49,602).
  prefixlength 0 selects vm_recommendedprefixlength (kurtz/detpfxlen.c:52).
*/
int vsa_index_build(const uint8_t *tis, uint64_t totallength,
                    uint32_t numofchars, uint32_t prefixlength, int device,
                    vsa_index **index);

/*
  mkvtree on the GPU: reads multiple-FASTA files like
  `mkvtree -db dbfiles.. [-q queryfiles..] -indexname NAME -dna -pl [n] -allout`
  (Mkvtree/mkvtree.c:689, input Mkvtree/mkvinput.c:173, DNA symbol map) and
  writes NAME.{prj,al1,tis,ois,des,sds,ssp,suf,lcp,llv,bck,bwt,sti1[,skp]} byte
  for byte as the reference does (Mkvtree/mkvprocess.c:99-816); the reference's
  vmatch reads them.  prefixlength 0 = the reference's recommendation;
  integersize 64 matches the reference's LP64 build, 32 halves suf/bck/llv.
  Plain (uncompressed) FASTA only.
*/
int vsa_mkvtree(const char *const *dbfiles, uint32_t numofdbfiles,
                const char *const *queryfiles, uint32_t numofqueryfiles,
                const char *indexname, uint32_t prefixlength,
                uint32_t integersize, int withskp, int device);

/* same, for a text that already lives in device memory (bench.py) */
int vsa_index_build_device(const void *device_tis, uint64_t totallength,
                           uint32_t numofchars, uint32_t prefixlength,
                           int device, vsa_index **index);

/* table sti1 of mkvtree (Mkvtree/mkvprocess.c:583-612), computed on the GPU
   from suf and lcp into host memory sti1[totallength+1]; the GPU search does
   not use it, the reference's default algorithm (kurtz/matchsub.c:353) does */
int vsa_index_make_sti1(const vsa_index *index, uint8_t *sti1);

/* declares the text of a built index as "database, separator at
   querysepposition, queries" -- what mkvtree -db G -q Q records in the .prj
   file -- so that vsa_findmaximaluniquematches can run on it */
int vsa_index_set_queryseparator(vsa_index *index, uint64_t querysepposition);

/* Matchparam.queryspeedup (Vmengine/mparms.h:53, `vmatch -qspeedup`): which
   of the reference's algorithms the MEM lists (`-l L -q`) follow in their
   order inside one query offset -- 0 = matchquerysubstring0
   (kurtz/matchsub.c:165), 2 = matchquerysubstring2 (:353), the reference's
   default and the default here.  The set of matches is the same; complete
   matches, MUMs and MUM candidates do not depend on it.  Other values are an
   error ("illegal speedup value", Vmengine/fquery.c:433). */
int vsa_index_set_queryspeedup(vsa_index *index, uint32_t queryspeedup);

/* copies the device tables back to host buffers sized by the caller from
   vsa_index_getinfo (entries of suf/bck/llv have device_integersize bits);
   NULL pointers are skipped */
int vsa_index_download(const vsa_index *index, uint8_t *tis, void *suf,
                       uint8_t *lcp, void *llv, void *bck, uint8_t *bwt);

/* ---- queries: Queryinfo.multiseq, Vmengine/mparms.h:73-84 ------------- */

typedef struct vsa_queries vsa_queries;

/*
  nq query sequences given as (start, length) pairs into one buffer of
  alphabet-mapped symbols -- the layout of a reference Multiseq: sequences
  separated by VSA_SEPARATOR, start[i] = markpos[i-1]+1
  (include/multidef.h:113-133, kurtz-basic/multiseq.c:129-166).
*/
int vsa_queries_from_host(const uint8_t *symbols, uint64_t nsymbols,
                          const uint64_t *start, const uint64_t *length,
                          uint64_t nq, int device, vsa_queries **queries);

/* nq queries of equal length m stored back to back in device memory
   (device_symbols[i*m .. i*m+m)); the buffer is copied */
int vsa_queries_from_device(const void *device_symbols, uint64_t nq,
                            uint32_t m, int device, vsa_queries **queries);

/*
  Reads at two bits per symbol: what crosses PCIe and lies in HBM for a batch
  of short reads of ONE length m is a quarter of the Multiseq's bytes.
    row of read i = words [i * W, (i + 1) * W) of `rows`, W =
      vsa_packed_words(m) = ceil((2 m + 8) / 64) 64-bit words; symbol j in
      bits 63 - 2 (j mod 32), 62 - 2 (j mod 32) of word j / 32 -- the first
      symbol in the top bits, codes a 0, c 1, g 2, t 3 as the DNA symbol map
      assigns them (kurtz-basic/alphabet.c:369, `mkvtree -dna`); every other
      bit 0, except the lowest byte of the last word, the row's flag:
        0  all m symbols are bases
        1  the read holds a special symbol (a wildcard, kurtz/maxpref.c:30-41:
           it matches nothing, not even itself): the row's symbol bits are
           ignored, word 0 = k << 8 (the flag byte stays free where W = 1)
           names entry k of `special`, the read's m mapped symbols as bytes
           at special + k * m.
  vsa_pack_reads makes rows (and the side list) from numofqueries reads of m
  mapped symbols, read i at symbols + i * stride (stride = m: back to back;
  m + 1: a Multiseq with its separators); *numofspecial counts the entries of
  `special` in use, before and after (several calls -- several threads over
  disjoint pieces with side lists of their own, or one reader in turn -- fill
  one batch); -2 if specialcapacity does not suffice.  Host code, no GPU:
  eight symbols per step on CPUs with BMI2 (PEXT), about 60 M reads of 100
  symbols per second and thread.  vsa_pack_reads_mt does the same on `threads`
  host threads over disjoint pieces of the batch (the side list is numbered
  afterwards, in the order of the reads: the rows do not depend on the number
  of threads).
  A packed batch is a batch like any other to every engine call.  -complete,
  -mum and -mum cand on an index with deep tables read the rows directly
  (reads of up to 252 symbols: whole in registers up to 124, through 16-byte
  windows of the row beyond); the other modes and longer reads make the bytes
  on the device first (once per batch: 0.3 ms per 10 M reads of 100 symbols).
*/
uint32_t vsa_packed_words(uint32_t querylength);
int vsa_pack_reads(const uint8_t *symbols, uint64_t numofqueries,
                   uint32_t querylength, uint64_t stride, uint64_t *rows,
                   uint8_t *special, uint64_t specialcapacity,
                   uint64_t *numofspecial);
int vsa_pack_reads_mt(const uint8_t *symbols, uint64_t numofqueries,
                      uint32_t querylength, uint64_t stride, uint64_t *rows,
                      uint8_t *special, uint64_t specialcapacity,
                      uint64_t *numofspecial, uint32_t threads);
int vsa_queries_from_host_packed(const uint64_t *rows, uint64_t numofqueries,
                                 uint32_t querylength, const uint8_t *special,
                                 uint64_t numofspecial, int device,
                                 vsa_queries **queries);

/* vmatch -p: the batch with every sequence replaced by its reverse
   complement (symbol 3 - c, wildcards stay), what copymultiseqRC
   (kurtz-basic/readmulti.c:93-125) stores in Multiseq.rcsequence and the
   engine receives with rcmode = True (Vmatch/runquery.c:179-279).  A symbol
   above 3 that is not a wildcard is the reference's error "reverse
   complement of %lu undefined" (readmulti.c:45-49).  Match positions inside
   the query refer to the reverse complement; the flip back to the forward
   strand is the sink's (Vmatch/procfinal.c:152-168). */
int vsa_queries_reverse_complement(const vsa_queries *queries,
                                   vsa_queries **rcqueries);

void vsa_queries_free(vsa_queries *queries);

/* queryseq of every match = index in the batch + offset: a rank that holds
   queries [offset, offset+nq) of a larger job reports global numbers
   (the reference's onlinequerynumoffset, Vmengine/fquery.c:1010,
   Vmengine/initmstate.c:7-45) */
int vsa_queries_set_offset(vsa_queries *queries, uint64_t offset);

typedef struct
{
  uint64_t numofqueries, numofsymbols;
  uint64_t minlength, maxlength; /* of a query; 0, 0 for an empty batch      */
  uint64_t offset;               /* vsa_queries_set_offset                   */
  int device;
} vsa_queries_info;

int vsa_queries_getinfo(const vsa_queries *queries, vsa_queries_info *info);

/* ---- matches ---------------------------------------------------------- */

/* field for field the reference's MUMcandidate (include/mumcand.h:17-23);
   what processexactquerymatch(info, l, i, queryseq, querystart)
   (Vmengine/procexqu.c:17-64) receives for one match */
typedef struct
{
  uint64_t length;     /* length1 = length2                           */
  uint64_t dbstart;    /* position1: absolute position in the index   */
  uint64_t queryseq;   /* seqnum2                                     */
  uint64_t querystart; /* relpos2                                     */
} vsa_match;

typedef struct vsa_result vsa_result; /* match list resident in HBM */

typedef struct
{
  uint64_t count;          /* matches                                     */
  uint64_t sumlength;      /* sum of match lengths ("bp matched")         */
  uint64_t searches;       /* bucket lookups + binary searches performed  */
  uint64_t candidates;     /* MUM candidates before the query-side filter */
  double search_kernel_ms; /* HIP-event time of the dominant search kernel */
  double total_device_ms;  /* HIP-event time of the whole call            */
  double anchor_ms;        /* -mum: anchor pass + work list (0 if unused)  */
  uint64_t kernel_searches; /* of those: by the dominant search kernel     */
  double first_kernel_ms;  /* -mum: HIP-event time of the first pass kernel
                              (offset 0 of every query, whole query);
                              -complete -e/-h: of the banded alignment     */
} vsa_stats;

uint64_t vsa_result_count(const vsa_result *result);
int vsa_result_getstats(const vsa_result *result, vsa_stats *stats);
/* copies min(count, capacity) matches to the host, in reference order */
int vsa_result_fetch(const vsa_result *result, vsa_match *matches,
                     uint64_t capacity);
const void *vsa_result_device_matches(const vsa_result *result);
/* device-to-device copy of min(count, capacity) matches into caller memory
   (e.g. a torch tensor that then goes through an RCCL collective) */
int vsa_result_copy_device(const vsa_result *result, void *device_matches,
                           uint64_t capacity);
void vsa_result_free(vsa_result *result);

/* ---- the engine: Vmengine/vmengineexport.h:4-81 ----------------------- */

/*
  findcompletematches (Vmengine/fcomplete.c:263-321) for exact matching on
  the index (decidefcm -> findexactcompletematchesindex,
  Vmengine/exactcompl.c:168-239).  Order: query order, within a query suffix
  array order.  A query shorter than prefixlength is the reference's hard
  error "patternlength=%lu must be >= %lu=prefixlen" (exactcompl.c:179-185):
  the matches of the queries before it are delivered, the return code is
  negative.  In every match length = query length and querystart = 0.
*/
int vsa_findcompletematches(const vsa_index *index,
                            const vsa_queries *queries, vsa_result **result);

/*
  The MUM candidates of a batch (vmatch -mum cand -l L), in reference order or
  -- ordered = 0 -- as the search kernel left them, for callers that hand
  them to a filter which sorts them anyway: the multi-GPU form of vmatch -mum
  (kurtz/cleanMUMcand.c:55-118 range-partitioned over the ranks, DESIGN.md
  section 6).
*/
int vsa_findmumcandidates(const vsa_index *index, const vsa_queries *queries,
                          uint64_t searchlength, int ordered,
                          vsa_result **result);

/*
  The same candidates as PAIRS of 8-byte words, never as 32-byte records:
    key   = dbstart << lengthbits | (2^lengthbits - 1 - length)
    value = queryseq << 16 | querystart
  -- what the filter sorts by, and what it needs to write a record once a
  candidate has survived.  lengthbits: the same on every rank of a job, at
  least the bits of the longest query anywhere (at most 16); 0 = the bits of
  this batch's longest query.  The result is for vsa_result_partition, which
  then writes rows of the two words (half the bytes of the exchange);
  vsa_result_fetch / vsa_result_copy_device deliver it as records,
  vsa_result_device_matches is NULL for it.
  Limits of the pair form: queries shorter than 65 535 symbols and global
  query numbers below 2^48; a batch beyond them is answered with -2 and a
  message, and the caller takes the record form -- vsa_findmumcandidates,
  vsa_result_partition on records, vsa_mumuniqueinquery_range -- as
  vsa_multi_findmatches (multi_gpu.cpp) does by itself.
*/
int vsa_findmumcandidates_packed(const vsa_index *index,
                                 const vsa_queries *queries,
                                 uint64_t searchlength, uint32_t lengthbits,
                                 vsa_result **result);
/* the lengthbits of a packed result, 0 for a result of records */
uint32_t vsa_result_packbits(const vsa_result *result);

/*
  The records of a result grouped by the range of the index their dbstart
  falls into -- part p = floor(dbstart * nparts / (totallength + 1)), equal
  dbstarts in the same part -- written to device_matches (room for
  vsa_result_count records) part by part; counts[p] (host, nparts entries) =
  records of part p; maxright[p] (host, may be NULL) = the largest right end
  dbstart + length - 1 among them, 0 if there is none.  The send buffer and
  the split sizes of the all-to-all that brings every candidate to the rank
  filtering its range, and what the ranks need to agree on the carry of
  vsa_mumuniqueinquery_range without looking at the records again.
*/
int vsa_result_partition(const vsa_result *result, uint32_t nparts,
                         uint64_t totallength, void *device_matches,
                         uint64_t *counts, uint64_t *maxright);
/* ... with the part `ownpart` (0 .. nparts-1) written behind all others, the
   others in ascending order in front of it: a rank's own candidates stay
   where they are, the rows in front of them are the send buffer of an
   all-to-all whose split for the rank itself is 0.  counts and maxright are
   indexed by part as above.  ownpart < 0: vsa_result_partition. */
int vsa_result_partition_own(const vsa_result *result, uint32_t nparts,
                             int ownpart, uint64_t totallength,
                             void *device_matches, uint64_t *counts,
                             uint64_t *maxright);
/* ... with the 2 * nparts numbers -- counts[0 .. nparts-1], then
   maxright[0 .. nparts-1] -- left in DEVICE memory (device_meta), where the
   all-gather of the ranks reads them: the call does not wait for the GPU.
   STREAM CONTRACT: the work is queued on the device's default (NULL) stream
   and nothing else orders it: whoever reads device_matches / device_meta does
   so on that stream (a copy or collective queued there), or synchronises
   with it first (hipStreamSynchronize(NULL) / an event recorded there).  A
   non-blocking stream of the caller's is NOT ordered behind it.  The same
   holds for the rows vsa_mumuniqueinquery_range_packed[2] reads: they must
   be complete on the default stream's terms when the call is made. */
int vsa_result_partition_device(const vsa_result *result, uint32_t nparts,
                                int ownpart, uint64_t totallength,
                                void *device_matches, uint64_t *device_meta);
/* vsa_findmumcandidates_packed and vsa_result_partition_device (totallength
   = that of the index) in one call, for a caller that keeps its row buffer
   from batch to batch: 0 = the rows (vsa_result_count(*result) of them) lie
   grouped in device_rows; 1 = there are more than `capacity` rows: nothing
   was grouped, *result holds the candidates (make room, then
   vsa_result_partition_device); < 0 as vsa_findmumcandidates_packed. */
int vsa_findmumcandidates_grouped(const vsa_index *index,
                                  const vsa_queries *queries,
                                  uint64_t searchlength, uint32_t lengthbits,
                                  uint32_t nparts, int ownpart,
                                  void *device_rows, uint64_t capacity,
                                  uint64_t *device_meta, vsa_result **result);

/*
  findcompletematches for approximate matching on the index, vmatch
  -complete -e K | -h K -q Q IDX: decidefcm -> findedistcompletematchesindex
  / findhammingcompletematchesindex (Vmengine/fcomplete.c:140-261) ->
  findapproxcompletematchesindex (Vmengine/approxcompl.c:138-199) ->
  splitesaapm (Vmengine/splitesaapm.c:458-558).
    doedist   1: edit distance (-e), 0: Hamming distance (-h)
    distvalue K;  percent 1: the threshold is m*K/100 (-e Kp, -h Kp,
              Vmengine/initcompl.c:52-56); percent 2: "best of" (-e Kb, -h Kb,
              initcompl.c:59-77) -- read by read (Vmengine/fcomplete.c:251-252
              restores K in front of every read) the smallest threshold
              t <= m*K/100 at which the read has a match at all
              (Vmengine/approxcompl.c:80-122), then the read's matches at
              threshold t; a read without one reports nothing.  Two passes
              here: the smallest distance of every read from a run at the
              percent thresholds, then every read at exactly its own
  A match is (length, dbstart, queryseq, distance): length = Match.length1
  (for -e the best-distance, then longest prefix behind dbstart,
  Vmengine/longestmatch.c; for -h the query length), the distance travels in
  the querystart field (the number of mismatches for -h, where the reference
  stores its negative in Match.distance, approxcompl.c:78).  Order: query
  order; within a query the merged candidate regions in ascending order,
  within a region DESCENDING dbstart, exactly as the reference's right-to-left
  verification reports them.
  Errors: a threshold >= query length is the reference's
  "threshold=%lu>=%lu=patternlen not allowed" (splitesaapm.c:496-501): the
  matches of the queries before it are delivered, the return code is -2.
  Patterns that are not cut (splitsize 1: short patterns -- reported in
  suffix array order, splitesaapm.c:523-543) and pieces with a threshold of
  their own (K >= m/10), i.e. what the reference hands to esaapm /
  esahamming, and Hamming distance with a wildcard in a read take a general
  path (a depth-first walk of the lcp-interval tree per piece,
  approx_tree.inc) instead of the pigeonhole path; a batch with ONE such
  query takes it as a whole.
  Reads whose threshold is 0 (-e Kp / -h Kp on short reads) are exact
  searches (approxcompl.c:167-176), also inside a batch whose other reads
  have thresholds > 0: the batch is cut into the two kinds and the lists are
  merged back into query order; the first read the reference would stop at
  (shorter than prefixlength with threshold 0, or threshold >= length) ends
  the run with its message, -2, and the matches of the reads before it.
  Text positions are kept in the width of the index tables: texts of 2^32
  symbols and more are searched like the others.
  VSA_NOT_COVERED (-4), no result: the configuration is one this engine does
  not implement (alphabets other than 4 symbols; a read of more than 512
  symbols; a batch of 2^31 reads and more, or more reads than
  bits(reads) + bits(text length) <= 64 allows) -- the caller keeps using its
  CPU function for such batches (integration/vmengine_shim.c does).
*/
#define VSA_NOT_COVERED (-4)
int vsa_findapproxcompletematches(const vsa_index *index,
                                  const vsa_queries *queries, int doedist,
                                  uint64_t distvalue, int percent,
                                  vsa_result **result);

/*
  findquerymatches (Vmengine/fquery.c:1009-1058) for exact matches:
    domaximaluniquematch = 0                      vmatch -l L        (MEM)
    domaximaluniquematch = 1, ...candidates = 1   vmatch -mum cand -l L
    domaximaluniquematch = 1, ...candidates = 0   vmatch -mum -l L
  searchlength is Matchparam.seedlength (Vmatch/matchlenparm.c:17-22); a
  value below prefixlength is the reference's error (fquery.c:440-446).
  Order: MEM and candidates by query, query offset, then witness / left /
  right like kurtz/matchsub.c with Vmengine/fquery.c:139-270 (the witness is
  that of the reference's default algorithm 2, or of algorithm 0 after
  vsa_index_set_queryspeedup(index, 0)); MUMs by ascending dbstart
  (kurtz/cleanMUMcand.c:55-118).
*/
int vsa_findquerymatches(const vsa_index *index, const vsa_queries *queries,
                         int domaximaluniquematch,
                         int domaximaluniquematchcandidates,
                         uint64_t searchlength, vsa_result **result);

/*
  findselfmatches with vmatmaxoutgeneric (Vmengine/fself.c:203-300,
  Vmengine/vmatfind.c:487-541), vmatch -l L IDX: maximal repeats of the
  index -- all pairs of positions whose common prefix has length >= L, cannot
  be extended to the right (its length is the reported one) and is left
  maximal (different left characters, or a special symbol / the start of the
  text on one side).  A match is (length, start1, start2, 0) with start1 <
  start2.  Order: the reference's -- attachments of children to their fathers
  in the order of its bottom-up traversal, the pairs of one attachment in the
  order of the nested loops of processbranch (vmatfind.c:433-469).  With
  queries inside the index only pairs of a database and a query position are
  reported (ACCEPTMATCH, fself.c:29-37).  Needs the bwt table.
  VSA_NOT_COVERED for alphabets of more than 32 symbols.
*/
int vsa_findmaximalrepeats(const vsa_index *index, uint64_t searchlength,
                           vsa_result **result);

/*
  findsupermax (Vmengine/fsuper.c:142-165), vmatch -supermax -l L IDX:
  supermaximal repeats of the index.  A match is (length, start1, start2, 0)
  with start1 < start2, laid out like the self-index MUMs (dbstart = start1,
  queryseq = start2 as an absolute position).  Order: the nodes of the
  lcp-interval tree in suffix array order, the pairs of a node by first,
  then second suffix -- the reference's order.  An index that holds queries
  is the reference's error "supermaximal repeat search does not allow query
  files in index" (Vmengine/fself.c:193-198).  Needs the bwt table.
*/
int vsa_findsupermaximalrepeats(const vsa_index *index, uint64_t searchlength,
                                vsa_result **result);

/*
  findtandems (Vmengine/ftandem.c:261-304), vmatch -tandem -l L IDX: right
  branching tandem repeats -- every position v where a string of length
  d >= L that names an lcp-interval occurs twice in a row and the repeat
  cannot be shifted right by one symbol.  A match is (d, v, v + d, 0), laid
  out like the other self matches.  Order: the lcp-intervals as the
  reference's bottom-up traversal completes them, the repeats of one interval
  from the reference's witness leftwards, then rightwards.  An index that
  holds queries is the reference's error "tandem repeat search does not allow
  query files in index" (ftandem.c:271-275).  Needs tis, suf, lcp (llv).
*/
int vsa_findtandems(const vsa_index *index, uint64_t searchlength,
                    vsa_result **result);

/*
  findmaximaluniquematches (Vmengine/fmumself.c:10-66): MUMs between the
  database and the query part of one index.  Reported like the reference's
  Outputfunction(outinfo, len, start1, start2): length, dbstart = start1,
  queryseq = start2 (absolute), querystart = 0; suffix array order.
  VSA_NOT_COVERED for alphabets of more than 128 symbols (mkvtree -smap with
  such a map: the reference's engine keeps those).
*/
int vsa_findmaximaluniquematches(const vsa_index *index,
                                 uint64_t searchlength, vsa_result **result);

/*
  The same scan over a part of the suffix array: the values first <= i < last
  of the reference's loop variable (fmumself.c:33 runs i = 2 .. totallength-1;
  the bounds are clamped to that).  This is the multi-GPU form:
  every rank holds the whole index, rank r of N scans
  [2 + r*(n-2)/N, 2 + (r+1)*(n-2)/N) -- the entries i-2, i-1, i around the ends
  of its range ("halo") come from its own copy of lcptab/bwttab -- and the
  lists of the ranks, concatenated in rank order, are the list of the whole
  scan; only the match counters are reduced.
*/
int vsa_findmaximaluniquematches_range(const vsa_index *index,
                                       uint64_t searchlength, uint64_t first,
                                       uint64_t last, vsa_result **result);

/*
  mumuniqueinquery (kurtz/cleanMUMcand.c:55-118) on its own: MUM candidates
  resident in device memory (any order; e.g. gathered from several GPUs) ->
  MUMs in ascending dbstart order.  The candidate buffer is reordered.
*/
int vsa_mumuniqueinquery(void *device_candidates, uint64_t ncandidates,
                         int device, vsa_result **result);

/*
  The same filter on ONE dbstart range of a job whose candidates are
  partitioned by dbstart over several GPUs: carry_dbright = the largest
  right end (dbstart + length - 1) among all candidates with a smaller
  dbstart, i.e. the value the reference's running variable `dbright`
  (kurtz/cleanMUMcand.c:63,90) has when its loop reaches this range.  Equal
  dbstarts must not be split between ranges.
*/
int vsa_mumuniqueinquery_range(void *device_candidates, uint64_t ncandidates,
                               int device, uint64_t carry_dbright,
                               vsa_result **result);
/* ... on rows of (key, value) pairs as vsa_result_partition wrote them for a
   packed result; totallength = that of the index.  The MUMs are records. */
int vsa_mumuniqueinquery_range_packed(const void *device_rows, uint64_t nrows,
                                      uint32_t lengthbits,
                                      uint64_t totallength, int device,
                                      uint64_t carry_dbright,
                                      vsa_result **result);
/* ... on two lists of such rows taken as one (a rank's own rows, which need
   not travel through the exchange, and the rows it received) */
int vsa_mumuniqueinquery_range_packed2(const void *device_rows, uint64_t nrows,
                                       const void *more_rows, uint64_t nmore,
                                       uint32_t lengthbits,
                                       uint64_t totallength, int device,
                                       uint64_t carry_dbright,
                                       vsa_result **result);

/*
  The same three entry points with the reference's delivery model: every
  match is handed to a callback on the calling thread, in reference order;
  a non-zero return stops the run and is propagated
  (Processfinalfunction, include/match.h:232; procexqu.c:61).
*/
typedef int (*vsa_processmatch)(void *info, const vsa_match *match);

int vsa_findcompletematches_cb(const vsa_index *index,
                               const vsa_queries *queries,
                               vsa_processmatch processmatch, void *info);
int vsa_findapproxcompletematches_cb(const vsa_index *index,
                                     const vsa_queries *queries, int doedist,
                                     uint64_t distvalue, int percent,
                                     vsa_processmatch processmatch,
                                     void *info);
int vsa_findquerymatches_cb(const vsa_index *index,
                            const vsa_queries *queries,
                            int domaximaluniquematch,
                            int domaximaluniquematchcandidates,
                            uint64_t searchlength,
                            vsa_processmatch processmatch, void *info);
int vsa_findmaximaluniquematches_cb(const vsa_index *index,
                                    uint64_t searchlength,
                                    vsa_processmatch processmatch,
                                    void *info);
int vsa_findsupermaximalrepeats_cb(const vsa_index *index,
                                   uint64_t searchlength,
                                   vsa_processmatch processmatch, void *info);
int vsa_findmaximalrepeats_cb(const vsa_index *index, uint64_t searchlength,
                              vsa_processmatch processmatch, void *info);
int vsa_findtandems_cb(const vsa_index *index, uint64_t searchlength,
                       vsa_processmatch processmatch, void *info);

/* ---- end to end: queries in host memory -> matches in host memory ------
   Three batches in flight: while one is searched, the next is uploaded and
   the matches of the one before are downloaded (page-locked buffers the caller
   fills and reads in place, one HIP stream per direction).  Reads of one
   length, packed back to back.  mode: 0 -complete, 1 -l L (MEM), 2 -mum cand,
   3 -mum (candidates of all batches stay on the device; the filter of
   kurtz/cleanMUMcand.c:55-118 runs once, in vsa_pipeline_finish).  queryseq
   of a match counts over all batches (Vmengine/fquery.c:1010
   onlinequerynumoffset).

     vsa_pipeline_open(index, 3, 20, 100, 10000000, &p);
     while (more reads) {
       uint8_t *buf;
       while ((buf = vsa_pipeline_hostbuffer(p)) == NULL)
         vsa_pipeline_next(p, &m, &n);          -- take a finished batch
       n_reads = fill(buf);  vsa_pipeline_submit(p, n_reads);
     }
     while (vsa_pipeline_next(p, &m, &n) != 1) ...;
     vsa_pipeline_finish(p, &mums, &nmums, &stats);                        */
typedef struct vsa_pipeline vsa_pipeline;
int vsa_pipeline_open(const vsa_index *index, int mode, uint64_t searchlength,
                      uint32_t querylength, uint64_t maxqueries,
                      vsa_pipeline **pipeline);
uint8_t *vsa_pipeline_hostbuffer(vsa_pipeline *pipeline);
int vsa_pipeline_submit(vsa_pipeline *pipeline, uint64_t numofqueries);
/* The same pipeline for reads at two bits per symbol (vsa_pack_reads): a
   quarter of the bytes cross PCIe and lie in HBM.  The caller packs into the
   slot's page-locked room -- *rows: maxqueries rows of vsa_packed_words(m)
   words, *special: maxspecial reads as bytes -- and submits the numbers it
   used.  vsa_pipeline_hostrows: 0 = a slot, 1 = all three batches are in
   flight (take results first).  Everything else as above. */
int vsa_pipeline_open_packed(const vsa_index *index, int mode,
                             uint64_t searchlength, uint32_t querylength,
                             uint64_t maxqueries, uint64_t maxspecial,
                             vsa_pipeline **pipeline);
int vsa_pipeline_hostrows(vsa_pipeline *pipeline, uint64_t **rows,
                          uint8_t **special);
int vsa_pipeline_submit_packed(vsa_pipeline *pipeline, uint64_t numofqueries,
                               uint64_t numofspecial);
/* 0: the oldest batch not yet delivered (host memory, valid until the next
   call that needs its slot); 1: nothing outstanding; < 0: that batch failed
   (message in vsa_messagespace(), matches up to the error delivered) */
int vsa_pipeline_next(vsa_pipeline *pipeline, const vsa_match **matches,
                      uint64_t *count);
int vsa_pipeline_finish(vsa_pipeline *pipeline, const vsa_match **matches,
                        uint64_t *count, vsa_stats *stats);
/* the same list at 16 bytes per MUM -- half the bytes on the host link, which
   is what a -mum job of short reads waits for at its end: reads of up to
   65 535 symbols, texts below 2^40 symbols */
typedef struct
{
  uint64_t dbstart_length;      /* dbstart << 24 | length */
  uint64_t queryseq_querystart; /* queryseq << 16 | querystart */
} vsa_match16;
#define VSA_MATCH16_LENGTH(x) ((x).dbstart_length & 0xFFFFFFull)
#define VSA_MATCH16_DBSTART(x) ((x).dbstart_length >> 24)
#define VSA_MATCH16_QUERYSEQ(x) ((x).queryseq_querystart >> 16)
#define VSA_MATCH16_QUERYSTART(x) ((x).queryseq_querystart & 0xFFFFull)
int vsa_pipeline_finish16(vsa_pipeline *pipeline, const vsa_match16 **matches,
                          uint64_t *count, vsa_stats *stats);
void vsa_pipeline_close(vsa_pipeline *pipeline);

/* For a caller that runs one pipeline per GPU and deals the batches of a job
   out to them (include/vstree_amd_multi.h, vsa_multi_pipeline_*):
   vsa_pipeline_set_offset: the number the first query of the NEXT submitted
   batch gets.  vsa_pipeline_take_candidates (-mum pipelines, all batches
   taken): the candidate rows of the job where they lie in device memory -- 16
   bytes each, sort key dbstart << lengthbits | (2^lengthbits - 1 - length) and
   value queryseq << 16 | querystart, valid until the next batch is submitted
   -- instead of vsa_pipeline_finish; ends the job.
   vsa_rows_partition_device: such rows grouped by the range of the index
   their dbstart falls into, like vsa_result_partition_device (same stream
   contract: the device's default stream). */
int vsa_pipeline_set_offset(vsa_pipeline *pipeline, uint64_t firstquery);
int vsa_pipeline_take_candidates(vsa_pipeline *pipeline,
                                 const void **device_rows, uint64_t *nrows,
                                 uint32_t *lengthbits);
int vsa_rows_partition_device(const void *device_rows, uint64_t nrows,
                              uint32_t lengthbits, uint32_t nparts,
                              int ownpart, uint64_t totallength, int device,
                              void *device_out, uint64_t *device_meta);

/* ---- synthetic inputs (bench.py, tests): SURVEY.md section 8d ---------- */

uint64_t vsa_splitmix64_at(uint64_t seed, uint64_t idx);
void vsa_synth_genome(uint64_t seed, uint64_t n, uint8_t *codes);
void vsa_synth_query_plan(uint64_t seed, uint64_t n, uint64_t nq, uint32_t m,
                          uint64_t *pos, uint32_t *substidx, uint8_t *step);
void vsa_synth_queries(uint64_t seed, const uint8_t *genome, uint64_t n,
                       uint64_t nq, uint32_t m, uint8_t *queries,
                       uint64_t *srcpos);
/* device-side generators writing into caller-provided device memory */
int vsa_synth_genome_device(uint64_t seed, uint64_t n, void *device_codes,
                            int device);
int vsa_synth_queries_device(const void *device_genome, uint64_t n,
                             const uint64_t *pos, const uint32_t *substidx,
                             const uint8_t *step, uint64_t nq, uint32_t m,
                             void *device_queries, int device);

/* plain device memory for callers without a HIP runtime of their own */
int vsa_device_malloc(uint64_t bytes, int device, void **ptr);
int vsa_device_free(void *ptr, int device);
int vsa_device_upload(void *device_dst, const void *host_src, uint64_t bytes,
                      int device);
int vsa_device_download(void *host_dst, const void *device_src, uint64_t bytes,
                        int device);
int vsa_device_count(void);
int vsa_device_synchronize(int device);
/* temporaries and freed result lists are recycled inside the library; this
   hands the cached device memory back to HIP */
int vsa_device_trim(int device);
/* what HIP reports as free / total memory of the device (the library's own
   cache of freed blocks counts as used until vsa_device_trim): how much room
   an index has, and what a failed job must leave unchanged */
int vsa_device_meminfo(int device, uint64_t *freebytes, uint64_t *totalbytes);
/* measured device-to-device streaming read rate in GB/s (roofline
   denominator cross-check in bench.py) */
int vsa_measure_stream_read(uint64_t bytes, int device, double *gbps);
/* independent random 8-byte reads over a table of `bytes` bytes, `inflight`
   (1, 4 or 8) of them issued per work-item before any is used: 10^9 reads
   per second.  The ceiling for the search kernels, whose traffic is one
   64-byte sector per read that misses the caches. */
int vsa_measure_random_read(uint64_t bytes, int inflight, int device,
                            double *greads);
/* the same reads over a table of a live index, where it lies in device
   memory: 0 slot16 (16-byte reads), 1 esa8, 2 tis2, 3 suf */
int vsa_measure_table_read(const vsa_index *ix, int table, int inflight,
                           double *greads);

/* ---- host match sink: from match records to vmatch's output lines ------ */

/*
  processfinal (Vmatch/procfinal.c:515-637: fetchpositions, convertthematch,
  assignEvalue, matchokay) and the default output line of
  vmatchnormaloutmatch (Vmatch/echomatch.c:878-987)
      len1 seq1 pos1 D|P len2 seq2 pos2 dist evalue score identity
  for the matches of this path, byte for byte what vmatch prints behind its
  "# args=" line (tests/test_sink.py: md5 of the lines of every golden run).
  Host side, no GPU involved; large batches are formatted by several
  threads, the text comes out in the order of the records.
  totalquerylength of an index with queries = totallength - (position of
  the separator in front of the first query sequence) - 1.
*/
#define VSA_SINK_COMPLETE       0 /* vsa_findcompletematches              */
#define VSA_SINK_QUERY          1 /* vsa_findquerymatches (-l, -mum ...)  */
#define VSA_SINK_SELF           2 /* vsa_findmaximaluniquematches         */
#define VSA_SINK_APPROX_EDIST   3 /* vsa_findapproxcompletematches, -e    */
#define VSA_SINK_APPROX_HAMMING 4 /* vsa_findapproxcompletematches, -h    */

/* Vmatch option -> bit of showmode (include/outinfo.h SHOW...) */
#define VSA_SHOW_ABSOLUTE   1u /* -absolute   */
#define VSA_SHOW_NODIST     2u /* -nodist     */
#define VSA_SHOW_NOEVALUE   4u /* -noevalue   */
#define VSA_SHOW_NOSCORE    8u /* -noscore    */
#define VSA_SHOW_NOIDENTITY 16u /* -noidentity */

typedef struct
{
  int kind;              /* VSA_SINK_...                                  */
  int palindromic;       /* matches of the reverse-complement pass (-p)   */
  int selfpalindromic;   /* ... of the index against itself (vmatch -p IDX,
                            Vmatch/runself.c:127-178): the query set is the
                            index; of the two mirror images of a match only
                            the one with the smaller left position is kept
                            (procfinal.c:159-167)                         */
  uint32_t showmode;     /* VSA_SHOW_... bits                             */
  uint32_t numofchars;   /* alpha.mapsize - 1: E-value match probability  */
  int threads;           /* formatting threads; 0 = one per processor     */
  uint64_t leastlength;  /* Matchparam.userdefinedleastlength (-l), or 0  */
  /* the index (Multiseq of the Virtualtree): */
  uint64_t totallength, numofsequences;
  const uint64_t *markpos;      /* numofsequences - 1 separator positions
                                   (IDX.ssp)                              */
  uint64_t numofquerysequences; /* sequences of queries INSIDE the index  */
  uint64_t totalquerylength;    /* their total length (IDX.prj), else 0   */
  /* the query set (not for VSA_SINK_SELF): sequence i occupies
     [querystart[i], querystart[i] + querylength[i]) of a Multiseq of
     querytotallength symbols */
  uint64_t numofqueries, querytotallength;
  const uint64_t *querystart, *querylength;
} vsa_sinkparams;

typedef struct vsa_sink vsa_sink;

int vsa_sink_open(const vsa_sinkparams *params, vsa_sink **sink);
void vsa_sink_close(vsa_sink *sink);
/* lines (each ending in a newline) into buffer; returns the bytes written or
   a negative code */
int64_t vsa_sink_format(vsa_sink *sink, const vsa_match *matches, uint64_t n,
                        char *buffer, uint64_t capacity);
/* the same to a FILE * */
int vsa_sink_write(vsa_sink *sink, const vsa_match *matches, uint64_t n,
                   void *file);

#ifdef __cplusplus
}
#endif
#endif
