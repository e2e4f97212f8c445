// Internal declarations shared by the HIP translation units of
// libvstree_amd.so.  The public C ABI is include/vstree_amd.h.
#pragma once

#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "vstree_amd.h"

// ---- error plumbing: one message buffer like the reference's messagespace()
// (include/errordef.h:13), thread local because the library may be used from
// one thread per GPU.
extern "C" char *vsa_errbuf();
#define VSA_ERRBUF_SIZE 1024
#define VSA_ERROR(...)                                                        \
  do                                                                          \
  {                                                                           \
    snprintf(vsa_errbuf(), VSA_ERRBUF_SIZE, __VA_ARGS__);                     \
  } while (0)

#define VSA_HIP(call)                                                         \
  do                                                                          \
  {                                                                           \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
    {                                                                         \
      VSA_ERROR("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,            \
                hipGetErrorString(e_));                                       \
      return -100;                                                            \
    }                                                                         \
  } while (0)

// Bytes in front of / behind the text on the device.  In front: tis[-1] may
// be loaded as part of an 8-byte word.  Behind: every suffix comparison stops
// at the first special symbol, and position totallength.. holds 0xFF, which
// reproduces "sptr >= sentinel => retcode = -1" (kurtz/maxpref.c:57-61)
// without a bounds test; 8-byte loads may run 7 bytes past it.
#define VSA_TIS_FRONTPAD 64
#define VSA_TIS_BACKPAD 64
// queries are compared 8 bytes at a time, too
#define VSA_QUERY_BACKPAD 64

// Device view of one index; IDX = uint32_t while totallength+1 fits, else
// uint64_t.  Passed to kernels by value.
template <typename IDX>
struct DevIndex
{
  const uint8_t *tis; // [-FRONTPAD, n + BACKPAD)
  const IDX *suf;     // [n+1]
  const uint8_t *lcp; // [n+1]
  const IDX *llv;     // [2*nllv]
  const IDX *bck;     // [2*numofcodes]
  const uint8_t *bwt; // [n+1] or nullptr
  // Deep locate (esa_device.hpp), DNA alphabets with 32-bit suf only:
  // esa8 [n+1] per suffix {suf:32 | lcp byte:8 | key:20 | left:2 | flag:1 |
  // leftspecial:1}, key = the VSA_KEYSYMS symbols behind the first D ones, 2
  // bits each, first symbol most significant, flag = a special symbol in that
  // window; left = the symbol in front of the suffix, leftspecial = that one
  // is a wildcard / separator or the suffix starts the text;
  // bck2 [2*4^D] (left, mid) pairs like bck, for D >= pl symbols.
  const uint64_t *esa8;
  const uint32_t *bck2;
  // slot16 [2*4^D] u64: per deep prefix (left | mid << 32, esa8[left]): the
  // bucket bounds AND its first entry in one 16-byte load; replaces bck2
  // when present (a bucket of one suffix -- the usual non-empty bucket --
  // then needs no second access)
  const uint64_t *slot16;
  uint32_t slotwords; // 2: bounds + first entry; 4: bounds + three entries
  // Long comparisons on a quarter of the bytes (DNA): tis2 = the text with
  // 2 bits per symbol, four symbols per byte, the first in the top bits
  // (special symbols stand as 0); spec64 = one bit per block of 64 text
  // positions, set if the block holds a special symbol or reaches beyond the
  // text, so that a comparison that touched such a block is repeated on the
  // bytes; firstspecial = the first position that is special (n if none)
  const uint8_t *tis2;
  const uint8_t *spec64;
  uint64_t firstspecial;
  uint64_t n, nllv, numofcodes;
  uint32_t pl, numofchars, D;
  uint32_t tune; // experiment switches (VSA_TUNE), see esa_search.hip
  // Matchparam.queryspeedup (Vmengine/mparms.h:53): 0 or 2, decides the
  // witness the MEM enumeration starts from
  uint32_t qspeedup;
};

#define VSA_KEYSYMS 10u
#define VSA_KEYSHIFT 40u
#define VSA_KEYMASK 0xFFFFFu
#define VSA_LEFTSHIFT 60u            // two bits: the symbol in front of the suffix
#define VSA_KEYFLAG (1ull << 62)
#define VSA_LEFTSPECIAL (1ull << 63) // ... is not a regular one / does not exist

struct vsa_index
{
  int device;
  hipStream_t stream;
  uint32_t isize; // bytes per suf/bck/llv entry on the device: 4 or 8
  uint64_t n, nllv, numofcodes;
  uint32_t pl, numofchars;
  uint8_t *tis_alloc; // allocation; text starts at tis_alloc + FRONTPAD
  void *suf, *llv, *bck;
  uint8_t *lcp, *bwt;
  uint64_t *esa8; // deep-locate tables, see DevIndex (may be nullptr)
  uint32_t *bck2;
  uint64_t *slot16;
  uint32_t slotwords;
  uint8_t *tis2, *spec64; // see DevIndex (may be nullptr)
  uint64_t firstspecial;
  uint32_t D, tune;
  uint32_t qspeedup; // vsa_index_set_queryspeedup: 0 or 2 (default)
  uint64_t querysepposition;
  int hasindexedqueries;
  uint64_t device_bytes;
  // 1: the two largest suffixes share >= 255 symbols, the one situation in
  // which the reference's uniqueness test for lcp >= 255 (fquery.c:352) can
  // call a repeated match unique; 0: cannot happen; -1: not looked at yet
  mutable int lcpquirk;
  // inverse suffix array, isa32[suf[i]] = i: made by the first call that wants
  // it (tandem repeats, selfmatch_search.inc) and kept; nullptr before
  mutable uint32_t *isa32;
  // one bit per text position: its suffix has a suffix-array neighbour with
  // lcp >= repleast (mem_workplan.inc); made by the first MEM batch that
  // asks for this least length and kept; nullptr before
  mutable uint32_t *repbits = nullptr;
  mutable uint32_t repleast = 0;

  template <typename IDX>
  DevIndex<IDX> view() const
  {
    DevIndex<IDX> v;
    v.tis = tis_alloc + VSA_TIS_FRONTPAD;
    v.suf = (const IDX *) suf;
    v.lcp = lcp;
    v.llv = (const IDX *) llv;
    v.bck = (const IDX *) bck;
    v.bwt = bwt;
    v.esa8 = esa8;
    v.bck2 = bck2;
    v.slot16 = slot16;
    v.slotwords = slotwords;
    v.tis2 = tis2;
    v.spec64 = spec64;
    v.firstspecial = firstspecial;
    v.D = D;
    v.tune = tune;
    v.qspeedup = qspeedup;
    v.n = n;
    v.nllv = nllv;
    v.numofcodes = numofcodes;
    v.pl = pl;
    v.numofchars = numofchars;
    return v;
  }
};

struct vsa_queries
{
  int device;
  uint64_t nq, nsymbols;
  // Reads at two bits per symbol (vsa_queries_from_host_packed, the packed
  // pipeline): rows != nullptr, roww 64-bit words per read, the reads with a
  // special symbol as bytes in `side` (nside of them, uniform length each).
  // `symbols` is then made on the device by the first call that needs bytes
  // (vsa_queries_bytes: MEM, approximate matching, indexes without the deep
  // tables) and kept.
  uint64_t *rows = nullptr;
  uint32_t roww = 0;
  uint8_t *side = nullptr;
  uint64_t nside = 0;
  bool ownsrows = true;
  mutable bool bytesvalid = false; // `symbols` holds this batch's bytes
  uint64_t bytescapacity = 0;      // pipeline slots: symbols of the largest
                                   // batch (0: this batch's)
  mutable uint8_t *symbols; // device, nsymbols + VSA_QUERY_BACKPAD
  uint64_t *start;  // device [nq]
  uint64_t *length; // device [nq]
  // host copies of the lengths' summary, for validation without a sync
  uint64_t minlength, maxlength;
  uint64_t seqoffset; // added to queryseq of every match
  std::vector<uint64_t> hlength; // host copy (needed for ragged batches)
  bool uniform;                  // all lengths equal
  bool dense;                    // uniform and start[i] = i * length
};

struct vsa_result
{
  int device;
  uint64_t count;
  vsa_match *matches; // device
  vsa_stats stats;
  // packbits != 0: MUM candidates as pairs instead of records (multi-GPU
  // -mum, vsa_findmumcandidates_packed): `matches` holds count sort keys
  // dbstart << packbits | (2^packbits - 1 - length), packvals the values
  // queryseq << 16 | querystart that go with them
  uint32_t packbits;
  uint64_t *packvals;
};

// pairs of a packed result as records on the device (esa_search.hip)
int vsa_unpack_result(const vsa_result *r, uint64_t count, vsa_match *device);

// device-side view of a query batch
struct DevQueries
{
  // packed batches: rows of roww words, see vsa_queries (symbols may be null)
  const uint64_t *rows;
  const uint8_t *side;
  uint64_t nside; // reads in the side list (indexes are clamped to it)
  uint32_t roww;
  const uint8_t *symbols;
  const uint64_t *start;
  const uint64_t *length;
  uint64_t nq;
  uint32_t uniformlen; // != 0: every query has this length
  uint32_t dense;      // != 0: and query i starts at i * uniformlen
  uint64_t seqoffset;
};

static inline DevQueries devqueries(const vsa_queries *q)
{
  DevQueries d;
  d.rows = q->rows;
  d.side = q->side;
  d.nside = q->nside;
  d.roww = q->roww;
  d.symbols = q->symbols;
  d.start = q->start;
  d.length = q->length;
  d.nq = q->nq;
  d.seqoffset = q->seqoffset;
  d.dense = (q->uniform && q->dense) ? 1u : 0u;
  d.uniformlen = (q->uniform && q->maxlength < 0xFFFFFFFFull)
                     ? (uint32_t) q->maxlength
                     : 0;
  return d;
}

int vsa_set_device(int device);

// words of a row of a packed batch of reads of m symbols: two bits per
// symbol and a flag byte (include/vstree_amd.h, vsa_packed_words)
static inline uint32_t vsa_rowwords(uint32_t m)
{
  return (2 * m + 8 + 63) / 64;
}
// the symbols of a packed batch as bytes on the device (made once, kept):
// for the kernels that read bytes (api.hip)
int vsa_queries_bytes(const vsa_queries *q, hipStream_t stream);

// recycled device memory for temporaries and result lists (devmem.hip)
int vsa_dev_alloc(void **ptr, size_t bytes);
void vsa_dev_free(void *ptr);
void vsa_dev_trim();
// the stream the calling thread's pipeline runs on: blocks are handed out and
// taken back in the order of that stream (see devmem.hip)
void vsa_dev_set_stream(hipStream_t stream);
void vsa_dev_forget_stream(hipStream_t stream);
// hipMalloc for long-lived tables; trims the cache and retries when HIP runs
// out of memory
hipError_t vsa_hip_malloc(void **ptr, size_t bytes);

// One grid dimension holds fewer than 2^32 work-items (the dispatch packet
// counts them in 32 bits): a launch over more blocks than VSA_GRID_X folds them
// into (x, y), and every kernel takes its block number from vsa_bid().  Blocks
// beyond the last item find nothing to do -- each kernel checks its bounds.
#define VSA_GRID_X (1u << 22)
inline dim3 vsa_grid(uint64_t blocks)
{
  if (blocks <= VSA_GRID_X)
  {
    return dim3((unsigned int) (blocks > 0 ? blocks : 1));
  }
  return dim3(VSA_GRID_X, (unsigned int) ((blocks + VSA_GRID_X - 1) / VSA_GRID_X));
}
// blocks such a launch really has (for tables with one entry per block)
inline uint64_t vsa_grid_blocks(uint64_t blocks)
{
  const dim3 g = vsa_grid(blocks);
  return (uint64_t) g.x * g.y;
}
#ifdef __HIPCC__
__device__ __forceinline__ uint64_t vsa_bid()
{
  return (uint64_t) blockIdx.y * gridDim.x + blockIdx.x;
}
__device__ __forceinline__ uint64_t vsa_nblocks()
{
  return (uint64_t) gridDim.x * gridDim.y;
}
#endif

// builds the deep-locate tables bck2/esa8 from tis/suf/lcp (esa_search.hip);
// a no-op for alphabets beyond 4 symbols, 64-bit tables or VSA_NO_ESA8=1
int vsa_index_make_esa8(vsa_index *ix);

// (left, mid) bucket boundaries for the first pl symbols, from the sorted
// suffixes (index_build.hip): out[2*numofchars^pl] on the device
int vsa_build_bucket_table(const uint8_t *tis, uint64_t n, const uint32_t *sa,
                           uint32_t pl, uint32_t numofchars, uint32_t *out,
                           hipStream_t stream);
int vsa_build_bucket_table(const uint8_t *tis, uint64_t n, const uint64_t *sa,
                           uint32_t pl, uint32_t numofchars, uint64_t *out,
                           hipStream_t stream);

// device tables of the given shape, contents undefined (api.hip)
// mayforcewide: VSA_FORCE_WIDE=1 may make the tables 64 bits wide (uploads of
// host tables; the GPU builder writes 32-bit tables and says false)
int vsa_index_alloc(uint64_t n, uint32_t pl, uint32_t numofchars,
                    uint64_t nllv, bool withbwt, int device, vsa_index **out,
                    bool mayforcewide = true);

