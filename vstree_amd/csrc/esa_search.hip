// Search kernels of the Vmengine query path on MI355X and their host-side
// pipelines.  Kernel inventory (DESIGN.md has the byte budgets):
//
//   K1  k_complete_search   one work-item per query: bucket -> lcp-aware
//                           binary search -> lcptab widening; writes the
//                           suffix-array interval [left, left+count)
//       k_complete_expand   one work-item per match (load-balanced over the
//                           scanned counts): suf[left+k] -> vsa_match
//   K2  k_query_search      one work-item per query suffix: bucket -> binary
//                           search -> MEM enumeration or MUM-candidate test;
//                           wavefront-aggregated append, then a stable radix
//                           sort by work-item number restores reference order
//   K4  k_mum_*             candidates sorted by (dbstart asc, length desc),
//                           prefix-max scan, flag, compact
//                           (kurtz/cleanMUMcand.c:55-118)
//   K3  k_selfmum_peaks     streaming scan over lcptab for indexes that hold
//       k_selfmum_emit      their queries (Vmengine/fmumself.c:10-66)
//
// rocPRIM supplies radix sort / scan / select / reduce only.
#include <cstring>
#include <algorithm>
#include "esa_device.hpp"
#include <rocprim/rocprim.hpp>

#define VSA_BLOCK 256
#define VSA_CURSOR_STRIDE 8   // uint64 words: one cursor per 64-byte line
#define VSA_CURSOR_SHARDS 2048 // power of two

// ---------------------------------------------------------------------------
// K1: exact complete matches (Vmengine/exactcompl.c:168-216)
// ---------------------------------------------------------------------------

// findsufboundaries, Vmengine/exactcompl.c:64-140: widen from the witness to
// all suffixes sharing >= least symbols, inside the bucket [vleft, vright]
template <typename IDX, bool KEYED>
__device__ __forceinline__ void
vsa_findsufboundaries(const DevIndex<IDX> &ix, uint32_t maxlcp,
                      uint64_t witness, uint32_t least, uint64_t vleft,
                      uint64_t vright, uint64_t &l, uint64_t &r)
{
  uint64_t i;

  if (maxlcp < 255)
  {
    for (i = witness;
         i != vleft && vsa_lcpbyte<IDX, KEYED>(ix, i) >= (least & 0xFFu); i--)
    {
    }
    l = i;
    for (i = witness + 1;
         i <= vright && vsa_lcpbyte<IDX, KEYED>(ix, i) >= (least & 0xFFu); i++)
    {
    }
    r = i - 1;
  } else
  {
    for (i = witness; i != vleft && vsa_evallcp(ix, i) >= least; i--)
    {
    }
    l = i;
    for (i = witness + 1; i <= vright && vsa_evallcp(ix, i) >= least; i++)
    {
    }
    r = i - 1;
  }
}

template <typename IDX, bool DEEP>
__global__ void __launch_bounds__(VSA_BLOCK)
k_complete_search(const DevIndex<IDX> ix, const DevQueries qs,
                  uint64_t qlimit, uint64_t *__restrict__ outleft,
                  uint64_t *__restrict__ outcount)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  const bool active = q < qlimit;
  const uint8_t *pattern = qs.symbols;
  uint32_t plen = 0, maxlcp = 0;
  uint64_t witness = 0, l = 0, count = 0;
  bool have = false;

  if (active)
  {
    if (qs.dense)
    {
      pattern = qs.symbols + q * qs.uniformlen;
      plen = qs.uniformlen;
    } else
    {
      pattern = qs.symbols + qs.start[q];
      plen = (uint32_t) qs.length[q];
    }
  }
  if constexpr (DEEP)
  {
    DeepHit hit;
    const int st = vsa_locate_deep(ix, active, pattern, plen, maxlcp, witness,
                                   hit);
    if (st == VSA_LOC_SLOW)
    {
      have = vsa_locate_reference(ix, pattern, plen, maxlcp, witness);
    } else
    {
      have = st == VSA_LOC_FOUND;
    }
  } else
  {
    if (active)
    {
      have = vsa_locate_reference(ix, pattern, plen, maxlcp, witness);
    }
  }
  if (have && maxlcp >= plen)
  {
    // the widening stops where lcp < plen, which every bucket boundary
    // satisfies (lcp < prefixlength <= plen): no need for the bucket here
    uint64_t r;
    vsa_findsufboundaries<IDX, DEEP>(ix, maxlcp, witness, plen, 0, ix.n, l,
                                     r);
    count = r - l + 1;
  }
  if (active)
  {
    outleft[q] = l;
    outcount[q] = count;
  }
}

// processfinalexactmatchinterval, Vmengine/exactcompl.c:142-166, for all
// queries at once: match t belongs to the query whose scanned count range
// contains t
template <typename IDX>
__global__ void __launch_bounds__(VSA_BLOCK)
k_complete_expand(const DevIndex<IDX> ix, const DevQueries qs, uint64_t nq,
                  const uint64_t *__restrict__ left,
                  const uint64_t *__restrict__ offsets, uint64_t total,
                  vsa_match *__restrict__ out)
{
  const uint64_t t = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;

  if (t >= total)
  {
    return;
  }
  // largest q with offsets[q] <= t (offsets has nq+1 entries, last = total)
  uint64_t lo = 0, hi = nq;
  while (hi - lo > 1)
  {
    const uint64_t mid = (lo + hi) >> 1;
    if (offsets[mid] <= t)
    {
      lo = mid;
    } else
    {
      hi = mid;
    }
  }
  vsa_match m;
  m.length = qs.length[lo];
  m.dbstart = (uint64_t) ix.suf[left[lo] + (t - offsets[lo])];
  m.queryseq = lo + qs.seqoffset;
  m.querystart = 0;
  out[t] = m;
}

// ---------------------------------------------------------------------------
// K2: matches of all query suffixes (kurtz/matchsub.c:165-235 driving
// Vmengine/fquery.c:139-270 / :297-386)
// ---------------------------------------------------------------------------

// macro PROCESSSUFFIX, Vmengine/fquery.c:54-81: a match is reported iff it
// cannot be extended to the left
template <typename IDX>
__device__ __forceinline__ bool vsa_leftmaximal(const DevIndex<IDX> &ix,
                                                uint64_t sufstart,
                                                uint8_t leftchar)
{
  return sufstart == 0 || VSA_ISSPECIAL(leftchar) ||
         leftchar != ix.tis[sufstart - 1];
}

// leftrightsubmatch, Vmengine/fquery.c:139-270 with the bounds algorithm 2
// passes (left = 0, right = totallength-1, kurtz/matchsub.c:504-515).
// WRITE = false counts the reports, WRITE = true stores them at out[0..).
template <typename IDX, bool KEYED, bool WRITE>
__device__ __forceinline__ uint32_t
vsa_mem_walk(const DevIndex<IDX> &ix, uint32_t maxlcp, uint64_t witness,
             uint8_t leftchar, uint32_t searchlength, uint64_t qseq,
             uint64_t qoff, vsa_match *out, uint64_t *outkey, uint64_t key)
{
  const uint64_t right = ix.n - 1;
  uint32_t c = 0, minprefix = maxlcp;
  uint64_t idx, lcpval;

#define VSA_REPORT(I, LEN)                                                    \
  {                                                                           \
    const uint64_t ss_ = vsa_sufstart<IDX, KEYED>(ix, I);                     \
    if (vsa_leftmaximal(ix, ss_, leftchar))                                   \
    {                                                                         \
      if (WRITE)                                                              \
      {                                                                       \
        vsa_match m_;                                                         \
        m_.length = (LEN);                                                    \
        m_.dbstart = ss_;                                                     \
        m_.queryseq = qseq;                                                   \
        m_.querystart = qoff;                                                 \
        out[c] = m_;                                                          \
        outkey[c] = key;                                                      \
      }                                                                       \
      c++;                                                                    \
    }                                                                         \
  }

  for (idx = witness;; idx--)
  {
    VSA_REPORT(idx, minprefix);
    if (idx == 0)
    {
      break;
    }
    lcpval = (maxlcp < 255) ? (uint64_t) vsa_lcpbyte<IDX, KEYED>(ix, idx)
                            : vsa_evallcp(ix, idx);
    if (lcpval < searchlength)
    {
      break;
    }
    if (minprefix > lcpval)
    {
      minprefix = (uint32_t) lcpval;
    }
  }
  minprefix = maxlcp;
  for (idx = witness + 1; idx <= right; idx++)
  {
    lcpval = (maxlcp < 255) ? (uint64_t) vsa_lcpbyte<IDX, KEYED>(ix, idx)
                            : vsa_evallcp(ix, idx);
    if (lcpval < searchlength)
    {
      break;
    }
    if (minprefix > lcpval)
    {
      minprefix = (uint32_t) lcpval;
    }
    VSA_REPORT(idx, minprefix);
  }
#undef VSA_REPORT
  return c;
}

// leftrightmaximaluniquematch, Vmengine/fquery.c:297-386, bounds as above.
// The reference's branch for maxlcp >= 255 looks at the right neighbour only
// if witness + 1 < right; kept as it stands.
template <typename IDX, bool KEYED>
__device__ __forceinline__ bool
vsa_mum_candidate(const DevIndex<IDX> &ix, uint32_t maxlcp, uint64_t witness)
{
  const uint64_t right = ix.n - 1;

  if (maxlcp < 255)
  {
    return (witness == 0 ||
            vsa_lcpbyte<IDX, KEYED>(ix, witness) < (maxlcp & 0xFFu)) &&
           (witness + 1 > right ||
            vsa_lcpbyte<IDX, KEYED>(ix, witness + 1) < (maxlcp & 0xFFu));
  }
  bool okay = (witness == 0) || vsa_evallcp(ix, witness) < maxlcp;
  if (okay && witness + 1 < right)
  {
    okay = vsa_evallcp(ix, witness + 1) < maxlcp;
  }
  return okay;
}

// Work-item t of the batch = (query q, offset off).  Queries of one length:
// arithmetic; ragged batches: binary search in the scanned per-query counts.
__device__ __forceinline__ void
vsa_decode_workitem(const DevQueries &qs, const uint64_t *__restrict__ base,
                    uint32_t perquery, uint64_t t, uint64_t &q, uint32_t &off)
{
  if (base == nullptr)
  {
    q = t / perquery;
    off = (uint32_t) (t - q * perquery);
  } else
  {
    uint64_t lo = 0, hi = qs.nq;
    while (hi - lo > 1)
    {
      const uint64_t mid = (lo + hi) >> 1;
      if (base[mid] <= t)
      {
        lo = mid;
      } else
      {
        hi = mid;
      }
    }
    q = lo;
    off = (uint32_t) (t - base[lo]);
  }
}

// [l, r] = all suffixes that share maxlcp symbols with the query, given one
// of them: neighbours in the suffix array whose lcp is >= maxlcp
template <typename IDX, bool KEYED>
__device__ __forceinline__ void
vsa_maxlcp_interval(const DevIndex<IDX> &ix, uint32_t maxlcp, uint64_t w,
                    uint64_t &l, uint64_t &r)
{
  for (l = w; l > 0; l--)
  {
    const uint64_t v = (maxlcp < 255)
                           ? (uint64_t) vsa_lcpbyte<IDX, KEYED>(ix, l)
                           : vsa_evallcp(ix, l);
    if (v < maxlcp)
    {
      break;
    }
  }
  for (r = w; r < ix.n; r++)
  {
    const uint64_t v = (maxlcp < 255)
                           ? (uint64_t) vsa_lcpbyte<IDX, KEYED>(ix, r + 1)
                           : vsa_evallcp(ix, r + 1);
    if (v < maxlcp)
    {
      break;
    }
  }
}

template <typename IDX, bool MUM, bool DEEP, int BLK>
__global__ void __launch_bounds__(BLK)
k_query_search(const DevIndex<IDX> ix, const DevQueries qs,
               const uint64_t *__restrict__ base, uint32_t perquery,
               const uint32_t *__restrict__ wlq,
               const uint32_t *__restrict__ wloff, uint64_t nitems,
               uint32_t searchlength,
               vsa_match *__restrict__ out, uint64_t *__restrict__ outkey,
               uint64_t shardcap, uint32_t shardmask,
               unsigned long long *__restrict__ cursors)
{
  uint64_t t = (uint64_t) blockIdx.x * BLK + threadIdx.x;
  const bool active = t < nitems;
  uint32_t c = 0, maxlcp = 0, off = 0, remaining = 0;
  uint64_t witness = 0, q = 0;
  uint8_t leftchar = (uint8_t) VSA_SEPARATOR;
  const uint8_t *qptr = qs.symbols;
  bool have = false, refwitness = true;

  if (active)
  {
    if (wlq != nullptr)
    {
      // explicit work list (k_mum_anchor / k_expand_worklist); the sort key
      // stays the number the work-item has in the full (query, offset) grid
      q = wlq[t];
      off = wloff[t];
      t = q * perquery + off;
    } else
    {
      vsa_decode_workitem(qs, base, perquery, t, q, off);
    }
    if (qs.dense)
    {
      qptr = qs.symbols + q * qs.uniformlen + off;
      remaining = qs.uniformlen - off;
    } else
    {
      qptr = qs.symbols + qs.start[q] + off;
      remaining = (uint32_t) qs.length[q] - off;
    }
    if (off > 0)
    {
      leftchar = qptr[-1];
    }
  }
  DeepHit hit;
  bool fast = false; // hit holds what the MUM test needs
  if constexpr (DEEP)
  {
    const int st = vsa_locate_deep(ix, active, qptr, remaining, maxlcp,
                                   witness, hit);
    if (st == VSA_LOC_SLOW)
    {
      have = vsa_locate_reference(ix, qptr, remaining, maxlcp, witness);
    } else
    {
      have = st == VSA_LOC_FOUND;
      refwitness = false;
      fast = have && maxlcp < 255;
    }
  } else
  {
    if (active)
    {
      have = vsa_locate_reference(ix, qptr, remaining, maxlcp, witness);
    }
  }
  const bool found = have && maxlcp >= searchlength;
  if (found)
  {
    if (MUM)
    {
      if (fast)
      {
        // leftrightmaximaluniquematch (fquery.c:297-386) and PROCESSSUFFIX
        // (fquery.c:54-81) on values already in registers
        const uint32_t lcpw = (uint32_t) (hit.ew >> 32) & 0xFFu;
        const uint64_t ss = hit.ew & 0xFFFFFFFFull;
        const bool unique = (witness == 0 || lcpw < maxlcp) &&
                            (witness + 1 > ix.n - 1 || hit.lcpnext < maxlcp);
        c = (unique && (ss == 0 || VSA_ISSPECIAL(leftchar) ||
                        leftchar != hit.leftsym))
                ? 1u
                : 0u;
      } else
      {
        c = (vsa_mum_candidate<IDX, DEEP>(ix, maxlcp, witness) &&
             vsa_leftmaximal(ix, vsa_sufstart<IDX, DEEP>(ix, witness),
                             leftchar))
                ? 1u
                : 0u;
      }
    } else
    {
      if (!refwitness)
      {
        // the enumeration starts at the reference's witness
        uint64_t l, r, vleft, vright;
        vsa_maxlcp_interval<IDX, DEEP>(ix, maxlcp, witness, l, r);
        if (l != r && vsa_bucket(ix, qptr, vleft, vright))
        {
          witness = vsa_reference_witness(vleft, vright, l, r);
        }
      }
      c = vsa_mem_walk<IDX, DEEP, false>(ix, maxlcp, witness, leftchar,
                                         searchlength, q, off, nullptr,
                                         nullptr, t);
    }
  }
  // all 64 lanes arrive here.  Output space comes from one of many cursors
  // (one 64-byte line each, picked by workgroup number): a single cursor
  // word takes ~11 ns per returning atomic, which for 10^7 wavefronts is
  // longer than the whole search.  The regions are compacted afterwards.
  const uint32_t shard = blockIdx.x & shardmask;
  const uint64_t inshard =
      vsa_wave_reserve(cursors + (uint64_t) shard * VSA_CURSOR_STRIDE, c);
  if (c > 0 && inshard + c <= shardcap)
  {
    const uint64_t mybase = (uint64_t) shard * shardcap + inshard;
    if (MUM)
    {
      vsa_match m;
      m.length = maxlcp;
      m.dbstart = fast ? (hit.ew & 0xFFFFFFFFull)
                       : vsa_sufstart<IDX, DEEP>(ix, witness);
      m.queryseq = q + qs.seqoffset;
      m.querystart = off;
      out[mybase] = m;
      outkey[mybase] = t;
    } else
    {
      vsa_mem_walk<IDX, DEEP, true>(ix, maxlcp, witness, leftchar,
                                    searchlength, q + qs.seqoffset, off,
                                    out + mybase, outkey + mybase, t);
    }
  }
}

// gathers the filled part of every cursor region into one dense list
__global__ void __launch_bounds__(VSA_BLOCK)
k_compact_shards(const vsa_match *__restrict__ in,
                 const uint64_t *__restrict__ inkey, uint64_t shardcap,
                 const unsigned long long *__restrict__ cursors,
                 const uint64_t *__restrict__ offsets,
                 vsa_match *__restrict__ out, uint64_t *__restrict__ outkey)
{
  const uint32_t shard = blockIdx.x;
  const uint64_t count = cursors[(uint64_t) shard * VSA_CURSOR_STRIDE],
                 src = (uint64_t) shard * shardcap, dst = offsets[shard];
  for (uint64_t i = threadIdx.x; i < count; i += VSA_BLOCK)
  {
    out[dst + i] = in[src + i];
    outkey[dst + i] = inkey[src + i];
  }
}

// ---------------------------------------------------------------------------
// MUM work reduction.  If the query suffix at offset j matches the text up to
// the END of the query, no offset j' > j of that query can produce a MUM
// candidate: its longest match is the rest of the query as well, and if that
// match is unique in the index it is the one at (position + j' - j), whose
// left neighbour is the matched query symbol, so the test of
// leftrightmaximaluniquematch / PROCESSSUFFIX (Vmengine/fquery.c:54-81,
// 297-386) rejects it; if it is not unique it is rejected as well.  (The
// reference's own algorithm 2 exploits the same suffix-link structure,
// kurtz/matchsub.c:400-491.)  So: locate the LAST offset of every query; if
// it matches completely, walk that occurrence backwards through the text as
// far as it agrees with the query -- down to offset m -- and only offsets
// 0..m need a search.  Queries that match end to end cost 2 searches instead
// of (length - l + 1).
// ---------------------------------------------------------------------------

template <typename IDX, bool DEEP>
__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_anchor(const DevIndex<IDX> ix, const DevQueries qs, uint32_t perquery,
             uint32_t searchlength, uint32_t *__restrict__ count)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  const bool active = q < qs.nq && perquery > 0;
  const uint32_t lastoff = perquery > 0 ? perquery - 1 : 0;
  const uint8_t *qstart = qs.symbols, *qptr = qs.symbols;
  uint32_t maxlcp = 0;
  uint64_t witness = 0, sufstart = 0;
  bool have = false;

  if (active)
  {
    qstart = qs.dense ? qs.symbols + q * qs.uniformlen
                      : qs.symbols + qs.start[q];
    qptr = qstart + lastoff;
  }
  if constexpr (DEEP)
  {
    DeepHit hit;
    const int st = vsa_locate_deep(ix, active, qptr, searchlength, maxlcp,
                                   witness, hit);
    if (st == VSA_LOC_SLOW)
    {
      have = vsa_locate_reference(ix, qptr, searchlength, maxlcp, witness);
      sufstart = have ? (uint64_t) ix.suf[witness] : 0;
    } else
    {
      have = st == VSA_LOC_FOUND;
      sufstart = hit.ew & 0xFFFFFFFFull;
    }
  } else
  {
    if (active)
    {
      have = vsa_locate_reference(ix, qptr, searchlength, maxlcp, witness);
      sufstart = have ? (uint64_t) ix.suf[witness] : 0;
    }
  }
  if (!active)
  {
    return;
  }
  uint32_t need = perquery;
  if (have && maxlcp >= searchlength)
  {
    // walk backwards: query[lastoff-1-k] against text[sufstart-1-k]; the
    // text has 0xFF in front of position 0, specials never match
    uint32_t k = 0;
    while (k < lastoff)
    {
      const uint32_t room = lastoff - k; // query symbols still available
      if (room >= 32)
      {
        // 32 symbols per round trip, most recent first
        uint64_t m[4];
        {
          // bytes [-32, 0) in front of the current position, two 16-byte
          // loads per side; word 0 = the most recent eight symbols
          const vsa_u128 qn = vsa_load16(qptr - k - 16),
                         qf = vsa_load16(qptr - k - 32),
                         tn = vsa_load16(ix.tis + sufstart - k - 16),
                         tf = vsa_load16(ix.tis + sufstart - k - 32);
          const uint64_t a[4] = {qn.hi, qn.lo, qf.hi, qf.lo},
                         b[4] = {tn.hi, tn.lo, tf.hi, tf.lo};
#pragma unroll
          for (int i = 0; i < 4; i++)
          {
            m[i] = (a[i] ^ b[i]) | vsa_specialmask(a[i]) |
                   vsa_specialmask(b[i]);
          }
        }
        if ((m[0] | m[1] | m[2] | m[3]) == 0)
        {
          k += 32;
          continue;
        }
        const int i = m[0] ? 0 : (m[1] ? 1 : (m[2] ? 2 : 3));
        const uint64_t mm = m[0] ? m[0] : (m[1] ? m[1] : (m[2] ? m[2] : m[3]));
        k += 8 * i + ((uint32_t) __builtin_clzll(mm) >> 3);
        break;
      } else if (room >= 8)
      {
        const uint64_t a = vsa_load8(qptr - k - 8),
                       b = vsa_load8(ix.tis + sufstart - k - 8);
        const uint64_t m = (a ^ b) | vsa_specialmask(a) | vsa_specialmask(b);
        if (m != 0)
        {
          k += (uint32_t) __builtin_clzll(m) >> 3; // matching bytes from top
          break;
        }
        k += 8;
      } else
      {
        const uint8_t a = qptr[-(int64_t) k - 1],
                      b = ix.tis[(int64_t) sufstart - (int64_t) k - 1];
        if (a != b || VSA_ISSPECIAL(a))
        {
          break;
        }
        k++;
      }
    }
    need = lastoff - k + 1; // offsets 0 .. lastoff-k
  }
  count[q] = need;
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_expand_worklist(const uint32_t *__restrict__ count,
                  const uint64_t *__restrict__ wbase, uint64_t nq,
                  uint32_t *__restrict__ wlq, uint32_t *__restrict__ wloff)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (q >= nq)
  {
    return;
  }
  const uint32_t c = count[q];
  const uint64_t b = wbase[q];
  for (uint32_t j = 0; j < c; j++)
  {
    wlq[b + j] = (uint32_t) q;
    wloff[b + j] = j;
  }
}

// ---- MUM work plan -------------------------------------------------------
//
// e(j) = j + (length of the longest match of the query suffix at offset j)
// never decreases with j, and an offset whose e equals that of the offset in
// front of it cannot be a MUM candidate: its longest match is the one of its
// predecessor moved by one symbol, so it is either not unique or not left
// maximal (leftrightmaximaluniquematch / PROCESSSUFFIX, fquery.c:54-81,
// 297-386; the reference's default algorithm 2 rides the same suffix links,
// matchsub.c:400-491).  So after the suffix at offset j has been located,
// with E = e(j): if the suffix at jp = E + 1 - l (the last one whose first l
// symbols end at E) also stops at E, every offset in (j, jp] stops at E and
// is dead; the offsets (jp, E] have to be searched, and E + 1 starts afresh.
// A read with one substitution at x costs 1 + l searches instead of x + 2.
//
// A plan is at most VSA_PLAN_RANGES ranges of offsets per query, 16 bits
// each for first offset and length.

#define VSA_PLAN_RANGES 4
#define VSA_PLAN_ROUNDS 6
#define VSA_PLAN_UNKNOWN 0xFFFFFFFFu

struct PlanRanges
{
  uint32_t r[VSA_PLAN_RANGES]; // first | length << 16
};

__device__ __forceinline__ void plan_add(PlanRanges &pr, uint32_t &nr,
                                         uint32_t first, uint32_t last)
{
  // [first, last], behind everything added so far
  if (first > last)
  {
    return;
  }
  if (nr > 0)
  {
    const uint32_t pf = pr.r[nr - 1] & 0xFFFFu, pl = pr.r[nr - 1] >> 16;
    if (pf + pl == first || nr == VSA_PLAN_RANGES)
    {
      // contiguous, or no slot left: extend the last range up to here
      // (searching a dead offset is harmless, it reports nothing)
      pr.r[nr - 1] = pf | ((last - pf + 1) << 16);
      return;
    }
  }
  pr.r[nr++] = first | ((last - first + 1) << 16);
}

// default plan of every query: offsets 0 .. need-1
__global__ void __launch_bounds__(VSA_BLOCK)
k_plan_default(const uint32_t *__restrict__ count, uint64_t nq,
               PlanRanges *__restrict__ plan)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (q < nq)
  {
    PlanRanges pr;
    pr.r[0] = count[q] << 16;
#pragma unroll
    for (int i = 1; i < VSA_PLAN_RANGES; i++)
    {
      pr.r[i] = 0;
    }
    plan[q] = pr;
  }
}

// one work-item per listed query (those with many offsets to search)
template <typename IDX, bool DEEP>
__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_plan(const DevIndex<IDX> ix, const DevQueries qs,
           const uint32_t *__restrict__ list, uint64_t nlist,
           uint32_t searchlength, const uint32_t *__restrict__ firste,
           uint32_t *__restrict__ count, PlanRanges *__restrict__ plan)
{
  const uint64_t t = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  bool busy = t < nlist;
  uint64_t q = 0;
  uint32_t need = 0, j = 0, nr = 0;
  // offset 0 has been located by k_mum_first: its e is known, and it is
  // not searched again
  uint32_t e0 = VSA_PLAN_UNKNOWN;
  const uint32_t qlen = qs.uniformlen;
  const uint8_t *qstart = qs.symbols;
  PlanRanges pr;

#pragma unroll
  for (int i = 0; i < VSA_PLAN_RANGES; i++)
  {
    pr.r[i] = 0;
  }
  if (busy)
  {
    q = list[t];
    need = count[q];
    qstart = qs.dense ? qs.symbols + q * qs.uniformlen
                      : qs.symbols + qs.start[q];
    if (firste != nullptr)
    {
      e0 = firste[q];
    }
  }
  for (int round = 0; round < VSA_PLAN_ROUNDS && __any(busy); round++)
  {
    // A: the suffix at offset j
    uint32_t maxlcp = 0, maxlcp2 = 0;
    uint64_t witness = 0;
    bool have = false, have2 = false;
    const bool known = round == 0 && e0 != VSA_PLAN_UNKNOWN;
    if (busy && !known && need - j <= 2)
    {
      plan_add(pr, nr, j, need - 1); // nothing to gain any more
      busy = false;
    }
    if constexpr (DEEP)
    {
      DeepHit hit;
      const int st = vsa_locate_deep(ix, busy && !known, qstart + j, qlen - j,
                                     maxlcp, witness, hit);
      if (st == VSA_LOC_SLOW)
      {
        have = vsa_locate_reference(ix, qstart + j, qlen - j, maxlcp, witness);
      } else
      {
        have = st == VSA_LOC_FOUND;
      }
    } else
    {
      if (busy && !known)
      {
        have = vsa_locate_reference(ix, qstart + j, qlen - j, maxlcp, witness);
      }
    }
    if (known)
    {
      have = true;
      maxlcp = e0;
    }
    uint32_t E = 0, jp = 0;
    bool probe = false;
    if (busy)
    {
      if (!known)
      {
        plan_add(pr, nr, j, j);
      }
      if (!have)
      {
        // fewer than prefixlength symbols match: no exact E; the offsets
        // up to there are searched, the next round starts behind them
        const uint32_t last = (j + ix.pl - 1 < need - 1) ? j + ix.pl - 1
                                                         : need - 1;
        plan_add(pr, nr, j + 1, last);
        j = last + 1;
        busy = j < need;
      } else
      {
        E = j + maxlcp;
        if (E >= qlen)
        {
          busy = false; // matches to the end: every later offset is dead
        } else
        {
          probe = E + 1 >= searchlength + j + 3 &&
                  E + 1 - searchlength < need;
          jp = probe ? E + 1 - searchlength : 0;
        }
      }
    }
    // B: the probe at jp
    const bool doprobe = busy && have && probe;
    if constexpr (DEEP)
    {
      DeepHit hit;
      const int st = vsa_locate_deep(ix, doprobe, qstart + jp, qlen - jp,
                                     maxlcp2, witness, hit);
      if (st == VSA_LOC_SLOW)
      {
        have2 = vsa_locate_reference(ix, qstart + jp, qlen - jp, maxlcp2,
                                     witness);
      } else
      {
        have2 = st == VSA_LOC_FOUND;
      }
    } else
    {
      if (doprobe)
      {
        have2 = vsa_locate_reference(ix, qstart + jp, qlen - jp, maxlcp2,
                                     witness);
      }
    }
    if (busy && have)
    {
      const bool dead = doprobe && have2 && jp + maxlcp2 == E;
      const uint32_t first = dead ? jp + 1 : j + 1;
      const uint32_t last = (E < need - 1) ? E : need - 1;
      plan_add(pr, nr, first, last);
      j = E + 1;
      busy = j < need;
    }
  }
  if (t < nlist)
  {
    if (busy)
    {
      plan_add(pr, nr, j, need - 1); // out of rounds: search the rest
    }
    uint32_t total = 0;
#pragma unroll
    for (int i = 0; i < VSA_PLAN_RANGES; i++)
    {
      total += pr.r[i] >> 16;
    }
    plan[q] = pr;
    count[q] = total;
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_expand_plan(const PlanRanges *__restrict__ plan,
              const uint64_t *__restrict__ wbase, uint64_t nq,
              uint32_t *__restrict__ wlq, uint32_t *__restrict__ wloff)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (q >= nq)
  {
    return;
  }
  const PlanRanges pr = plan[q];
  uint64_t b = wbase[q];
#pragma unroll
  for (int i = 0; i < VSA_PLAN_RANGES; i++)
  {
    const uint32_t first = pr.r[i] & 0xFFFFu, len = pr.r[i] >> 16;
    for (uint32_t j = 0; j < len; j++)
    {
      wlq[b] = (uint32_t) q;
      wloff[b] = first + j;
      b++;
    }
  }
}


// The first pass of a MUM batch: offset 0 of every query, located with the
// whole query.  A query that matches completely is finished here: e(0) is
// the end of the query, so no later offset can be a candidate (see above),
// and offset 0 is one iff its match is unique -- its left neighbour is the
// start of the query, which PROCESSSUFFIX treats as left maximal
// (fquery.c:54-81).  That is one search for an exact read instead of two
// (last offset + offset 0) and a backward walk.  For the others e(0) goes to
// the work plan.
template <typename IDX, bool DEEP>
__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_first(const DevIndex<IDX> ix, const DevQueries qs, uint32_t perquery,
            uint32_t searchlength, uint32_t *__restrict__ count,
            uint32_t *__restrict__ firste, uint32_t *__restrict__ fmlen,
            uint64_t *__restrict__ fmdb)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  const bool active = q < qs.nq && perquery > 0;
  const uint32_t qlen = qs.uniformlen;
  const uint8_t *qptr = qs.symbols;
  uint32_t maxlcp = 0;
  uint64_t witness = 0;
  bool have = false, fast = false;
  DeepHit hit;

  if (active)
  {
    qptr = qs.dense ? qs.symbols + q * qs.uniformlen
                    : qs.symbols + qs.start[q];
  }
  if constexpr (DEEP)
  {
    const int st = vsa_locate_deep(ix, active, qptr, qlen, maxlcp, witness,
                                   hit);
    if (st == VSA_LOC_SLOW)
    {
      have = vsa_locate_reference(ix, qptr, qlen, maxlcp, witness);
    } else
    {
      have = st == VSA_LOC_FOUND;
      fast = have && maxlcp < 255;
    }
  } else
  {
    if (active)
    {
      have = vsa_locate_reference(ix, qptr, qlen, maxlcp, witness);
    }
  }
  if (!active)
  {
    return;
  }
  uint32_t len = 0;
  uint64_t db = 0;
  if (have && maxlcp >= searchlength)
  {
    bool unique;
    if (fast)
    {
      const uint32_t lcpw = (uint32_t) (hit.ew >> 32) & 0xFFu;
      unique = (witness == 0 || lcpw < maxlcp) &&
               (witness + 1 > ix.n - 1 || hit.lcpnext < maxlcp);
      db = hit.ew & 0xFFFFFFFFull;
    } else
    {
      unique = vsa_mum_candidate<IDX, DEEP>(ix, maxlcp, witness);
      db = vsa_sufstart<IDX, DEEP>(ix, witness);
    }
    len = unique ? maxlcp : 0;
  }
  fmlen[q] = len;
  fmdb[q] = db;
  firste[q] = have ? maxlcp : VSA_PLAN_UNKNOWN;
  count[q] = (have && maxlcp >= qlen) ? 0 : perquery;
}

// candidates of the first pass -> behind the matches of the search kernel,
// with the sort key of work-item (query, offset 0)
__global__ void __launch_bounds__(VSA_BLOCK)
k_append_first(const uint32_t *__restrict__ fmlen,
               const uint64_t *__restrict__ fmdb,
               const uint32_t *__restrict__ slot, uint64_t nq,
               uint32_t perquery, uint64_t seqoffset, uint64_t base,
               vsa_match *__restrict__ out, uint64_t *__restrict__ outkey)
{
  const uint64_t q = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (q < nq && fmlen[q] != 0)
  {
    vsa_match m;
    m.length = fmlen[q];
    m.dbstart = fmdb[q];
    m.queryseq = q + seqoffset;
    m.querystart = 0;
    out[base + slot[q]] = m;
    outkey[base + slot[q]] = q * perquery;
  }
}

struct NonZeroToU32
{
  __device__ uint32_t operator()(uint32_t v) const
  {
    return v != 0 ? 1u : 0u;
  }
};

struct PlanWanted
{
  const uint32_t *count;
  uint32_t threshold;
  __device__ bool operator()(uint32_t q) const
  {
    return count[q] > threshold;
  }
};

// ---------------------------------------------------------------------------
// K4: MUM candidates -> MUMs (kurtz/cleanMUMcand.c:55-118)
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_keys(const vsa_match *__restrict__ cand, uint64_t n,
           uint64_t *__restrict__ keylen, uint64_t *__restrict__ keydb)
{
  const uint64_t i = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    keylen[i] = ~cand[i].length; // decreasing length
    keydb[i] = cand[i].dbstart;
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_rightends(const vsa_match *__restrict__ cand, uint64_t n,
                uint64_t *__restrict__ rightend)
{
  const uint64_t i = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    rightend[i] = cand[i].dbstart + cand[i].length - 1;
  }
}

// dbright[i] = max(0, rightend[0..i)) is what the reference's running
// variable holds when it looks at candidate i.  Candidate i survives iff it
// is not covered (dbright < rightend) and its successor does not end at the
// same position with the same start.
__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_flags(const vsa_match *__restrict__ cand,
            const uint64_t *__restrict__ rightend,
            const uint64_t *__restrict__ dbright, uint64_t n,
            uint8_t *__restrict__ keep)
{
  const uint64_t i = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (i >= n)
  {
    return;
  }
  bool k = dbright[i] < rightend[i];
  if (k && i + 1 < n)
  {
    // dbright[i+1] = rightend[i] here
    if (rightend[i + 1] == rightend[i] &&
        cand[i + 1].dbstart == cand[i].dbstart)
    {
      k = false;
    }
  }
  keep[i] = k ? 1 : 0;
}

// ---------------------------------------------------------------------------
// K3: MUMs on an index that contains its queries (Vmengine/fmumself.c:10-66)
// ---------------------------------------------------------------------------

// K3a  k_selfmum_peaks: the streaming pass.  A workgroup takes 16 KiB of
//      lcptab (256 work-items x 4 x one 128-bit load, consecutive lanes on
//      consecutive 16-byte pieces), every work-item tests its 64 positions on
//      the lcp bytes alone -- "second >= l, first < second, third < second"
//      (fmumself.c:36-37) -- and the positions that pass (or that need the
//      exception table because a byte is 255) and whose two suffixes differ
//      in the symbol to their left (bwt, streamed alongside) leave through
//      ONE cursor reservation per workgroup.  2n bytes in, 4 bytes out per
//      surviving peak.
// K3b  k_selfmum_emit: one work-item per peak (sorted by position): exact lcp
//      values, the two suffix starts, the db/query sides, left maximality on
//      bwt; writes the match and a keep flag, compacted in order afterwards.


// Four suffix array positions per 32-bit operation.  All helpers deliver
// their verdict in bit 7 of each byte and leave the other bits undefined.
#define VSA_B7 0x80808080u
#define VSA_L7 0x7F7F7F7Fu

// (m & a) | (~m & b), one v_bfi_b32
__device__ __forceinline__ uint32_t vsa_bfi(uint32_t m, uint32_t a, uint32_t b)
{
  return (m & a) | (~m & b);
}

// bytewise unsigned x < y; ylow = y & VSA_L7.  Where the top bits differ y's
// top bit decides, elsewhere the carry of ylow + (127 - xlow) into bit 7.
__device__ __forceinline__ uint32_t vsa_bytes_lt(uint32_t x, uint32_t y,
                                                 uint32_t ylow)
{
  return vsa_bfi(x ^ y, y, ylow + (~x & VSA_L7));
}

// The peak test of fmumself.c:36-37,50-52 on four suffix array positions:
// s = lcp bytes of the four centres, f / t = their left / right neighbours,
// a / b = bwt of the two suffixes of each centre.  Exact on bytes below 255;
// a centre of 255 always passes and is resolved through llv by k_selfmum_emit.
// ltmin = vsa_bytes_lt(s, l, ...) comes from the caller, which has it already.
__device__ __forceinline__ uint32_t
vsa_peakbits4(uint32_t f, uint32_t s, uint32_t t, uint32_t a, uint32_t b,
              uint32_t ltmin)
{
  const uint32_t slow = s & VSA_L7;
  const uint32_t is255 = (slow + 0x01010101u) & s;
  const uint32_t peak = vsa_bytes_lt(f, s, slow) & vsa_bytes_lt(t, s, slow);
  const uint32_t x = a ^ b;
  const uint32_t differ = ((x & VSA_L7) + VSA_L7) | x;    // bwt bytes differ
  const uint32_t special = ((a & VSA_L7) + 0x02020202u) & a; // a >= 254
  return (is255 | peak) & ~ltmin & (differ | special);
}

// 16 aligned bytes of a table that is read once (NT: nontemporal hint)
template <bool NT>
__device__ __forceinline__ uint4 vsa_stream16(const uint8_t *ptr)
{
  typedef uint32_t v4u __attribute__((ext_vector_type(4)));
  const v4u *q = reinterpret_cast<const v4u *>(ptr);
  const v4u r = NT ? __builtin_nontemporal_load(q) : *q;
  return make_uint4(r.x, r.y, r.z, r.w);
}

// One wavefront = one tile of (64 * PIECES * 16) suffix array positions at a time;
// wavefronts walk the tiles grid-stride and never synchronise with each other
// (no LDS, no barrier).  The loads of the NEXT tile are issued before the
// current one is examined, so that the arithmetic (about 11 operations per
// position when every word has an lcp byte >= l, i.e. two near-identical
// genomes) overlaps the memory stream instead of alternating with it.  A word
// of four positions runs the full test only if one of its lcp bytes is >= l.
template <int PIECES>
struct PeakTile
{
  uint4 v[PIECES], u[PIECES];
  uint32_t lcpbefore, bwtbefore, lcpafter; // halo words, valid in every lane
};

template <int PIECES, bool NT>
__device__ __forceinline__ void
vsa_peak_load(PeakTile<PIECES> &t, const uint8_t *__restrict__ lcp,
              const uint8_t *__restrict__ bwt, uint64_t n, uint64_t base,
              uint32_t lane)
{
  // lcp and bwt have n+1 entries and at least 32 bytes of slack behind them;
  // reads beyond that are clamped away
#pragma unroll
  for (int p = 0; p < PIECES; p++)
  {
    const uint64_t off = base + ((uint64_t) p * 64 + lane) * 16;
    const bool inside = off <= n;
    t.v[p] = inside ? vsa_stream16<NT>(lcp + off) : make_uint4(0, 0, 0, 0);
    t.u[p] = inside ? vsa_stream16<NT>(bwt + off) : make_uint4(0, 0, 0, 0);
  }
  t.lcpbefore =
      (base >= 4) ? *reinterpret_cast<const uint32_t *>(lcp + base - 4) : 0u;
  t.bwtbefore =
      (base >= 4) ? *reinterpret_cast<const uint32_t *>(bwt + base - 4) : 0u;
  t.lcpafter = (base + (64 * PIECES * 16) <= n)
                   ? *reinterpret_cast<const uint32_t *>(lcp + base +
                                                         (64 * PIECES * 16))
                   : 0u;
}

template <int PIECES, bool NT>
__global__ void __launch_bounds__(VSA_BLOCK)
k_selfmum_peaks(const uint8_t *__restrict__ lcp,
                const uint8_t *__restrict__ bwt, uint64_t n,
                uint32_t slmin, uint32_t *__restrict__ outpos,
                uint64_t shardcap, uint32_t shardmask,
                unsigned long long *__restrict__ cursors, uint64_t ntiles)
{
  const uint32_t lane = threadIdx.x & 63;
  const uint64_t nwaves = (uint64_t) gridDim.x * (VSA_BLOCK / 64);
  const uint32_t minv = slmin * 0x01010101u;
  uint64_t tile = (uint64_t) blockIdx.x * (VSA_BLOCK / 64) + (threadIdx.x >> 6);
  const uint32_t shard = (uint32_t) tile & shardmask;
  PeakTile<PIECES> cur;

  if (tile >= ntiles)
  {
    return;
  }
  vsa_peak_load<PIECES, NT>(cur, lcp, bwt, n, tile * (64 * PIECES * 16), lane);
  while (true)
  {
    const uint64_t next = tile + nwaves, base = tile * (64 * PIECES * 16);
    PeakTile<PIECES> nxt;
    if (next < ntiles)
    {
      vsa_peak_load<PIECES, NT>(nxt, lcp, bwt, n, next * (64 * PIECES * 16), lane);
    }
    uint32_t hits[PIECES];
    uint32_t c = 0;
#pragma unroll
    for (int p = 0; p < PIECES; p++)
    {
      const uint64_t off = base + ((uint64_t) p * 64 + lane) * 16;
      // the words around this piece live in the neighbouring lanes
      uint32_t wb = __shfl_up(cur.v[p].w, 1, 64),
               bb = __shfl_up(cur.u[p].w, 1, 64),
               wa = __shfl_down(cur.v[p].x, 1, 64);
      if (lane == 0)
      {
        wb = (p > 0) ? (uint32_t) __builtin_amdgcn_readlane(
                           (int) cur.v[p > 0 ? p - 1 : 0].w, 63)
                     : cur.lcpbefore;
        bb = (p > 0) ? (uint32_t) __builtin_amdgcn_readlane(
                           (int) cur.u[p > 0 ? p - 1 : 0].w, 63)
                     : cur.bwtbefore;
      }
      if (lane == 63)
      {
        wa = (p + 1 < PIECES)
                 ? (uint32_t) __builtin_amdgcn_readlane(
                       (int) cur.v[p + 1 < PIECES ? p + 1 : p].x, 0)
                 : cur.lcpafter;
      }
      const uint32_t w[6] = {wb, cur.v[p].x, cur.v[p].y, cur.v[p].z,
                             cur.v[p].w, wa};
      const uint32_t b[5] = {bb, cur.u[p].x, cur.u[p].y, cur.u[p].z,
                             cur.u[p].w};
      // verdict of centre off + 4k + i lands in bit 8i + 7 - k of h
      uint32_t h = 0;
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        const uint32_t ltmin = vsa_bytes_lt(w[k + 1], minv, minv & VSA_L7);
        if ((ltmin & VSA_B7) != VSA_B7) // some lcp byte >= l
        {
          const uint32_t x =
              vsa_peakbits4(__builtin_amdgcn_alignbit(w[k + 1], w[k], 24),
                            w[k + 1],
                            __builtin_amdgcn_alignbit(w[k + 2], w[k + 1], 8),
                            b[k + 1],
                            __builtin_amdgcn_alignbit(b[k + 1], b[k], 24),
                            ltmin);
          h |= (x >> k) & (VSA_B7 >> k);
        }
      }
      // centres j with 1 <= j <= n-2 (the reference's i = j+1 runs 2 .. n-1)
      if (off == 0)
      {
        h &= ~0x80u;
      }
      if (off + 17 > n)
      {
        uint32_t valid = 0;
        for (uint32_t q = 0; q < 16 && off + q + 2 <= n; q++)
        {
          valid |= 1u << (8 * (q & 3u) + 7 - (q >> 2));
        }
        h &= valid;
      }
      hits[p] = h;
      c += (uint32_t) __builtin_popcount(h);
    }
    // wavefront-wide exclusive scan of c, one reservation per wavefront
    uint32_t incl = c;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1)
    {
      const uint32_t o = __shfl_up(incl, d, 64);
      if (lane >= (uint32_t) d)
      {
        incl += o;
      }
    }
    const uint32_t total = __shfl(incl, 63, 64);
    if (total > 0)
    {
      unsigned long long wbase = 0;
      if (lane == 0)
      {
        wbase = atomicAdd(cursors + (uint64_t) shard * VSA_CURSOR_STRIDE,
                          (unsigned long long) total);
      }
      wbase = __shfl(wbase, 0, 64);
      if (c > 0 && wbase + total <= shardcap)
      {
        uint32_t *dst = outpos + (uint64_t) shard * shardcap;
        uint64_t slot = wbase + incl - c;
#pragma unroll
        for (int p = 0; p < PIECES; p++)
        {
          // position reported = j + 1, as the reference counts it
          const uint64_t first = base + ((uint64_t) p * 64 + lane) * 16 + 1;
          uint32_t h = hits[p];
          while (h != 0)
          {
            const uint32_t bit = (uint32_t) __builtin_ctz(h);
            h &= h - 1;
            dst[slot++] = (uint32_t) (first + 4 * (7 - (bit & 7u)) + (bit >> 3));
          }
        }
      }
    }
    if (next >= ntiles)
    {
      break;
    }
    cur = nxt;
    tile = next;
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_gather_u32_shards(const uint32_t *__restrict__ in, uint64_t shardcap,
                    const unsigned long long *__restrict__ cursors,
                    const uint64_t *__restrict__ offsets,
                    uint32_t *__restrict__ out)
{
  const uint32_t shard = blockIdx.x;
  const uint64_t count = cursors[(uint64_t) shard * VSA_CURSOR_STRIDE],
                 src = (uint64_t) shard * shardcap, dst = offsets[shard];
  for (uint64_t i = threadIdx.x; i < count; i += VSA_BLOCK)
  {
    out[dst + i] = in[src + i];
  }
}

template <typename IDX>
__global__ void __launch_bounds__(VSA_BLOCK)
k_selfmum_emit(const DevIndex<IDX> ix, const uint32_t *__restrict__ peaks,
               uint64_t npeaks, uint64_t searchlength,
               uint64_t querysepposition, vsa_match *__restrict__ out,
               uint8_t *__restrict__ keep)
{
  const uint64_t t = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (t >= npeaks)
  {
    return;
  }
  const uint64_t i = peaks[t];
  uint64_t first = ix.lcp[i - 2], second = ix.lcp[i - 1], third = ix.lcp[i];
  // SEQUENTIALEVALLCPVALUE (virtualdef.h:121-136) by index instead of in
  // sequence
  if (first == 255)
  {
    first = vsa_largelcp(ix, i - 2);
  }
  if (second == 255)
  {
    second = vsa_largelcp(ix, i - 1);
  }
  if (third == 255)
  {
    third = vsa_largelcp(ix, i);
  }
  bool ok = second >= searchlength && first < second && third < second;
  vsa_match m;
  m.length = second;
  m.dbstart = m.queryseq = m.querystart = 0;
  if (ok)
  {
    uint64_t s1 = (uint64_t) ix.suf[i - 2], s2 = (uint64_t) ix.suf[i - 1];
    if (s1 > s2)
    {
      const uint64_t tmp = s1;
      s1 = s2;
      s2 = tmp;
    }
    // left maximality (fmumself.c:50-52) was decided exactly by the peak
    // pass: the suffix at text position 0 carries bwt 253, which differs
    // from every symbol, so that "start1 == 0" needs no case of its own
    ok = s1 < querysepposition && s2 > querysepposition;
    m.dbstart = s1;
    m.queryseq = s2;
  }
  out[t] = m;
  keep[t] = ok ? 1 : 0;
}

struct KeepToU32
{
  __device__ uint32_t operator()(uint8_t k) const
  {
    return k;
  }
};

// order-preserving compaction of 32-byte records: slot[] = exclusive scan of
// keep[] (rocprim::select moves records of this size at a fraction of the
// memory rate: 3.6 ms for 15.6 M records, this pair of passes 0.2 ms)
__global__ void __launch_bounds__(VSA_BLOCK)
k_scatter_kept(const vsa_match *__restrict__ in,
               const uint8_t *__restrict__ keep,
               const uint32_t *__restrict__ slot, uint64_t count,
               vsa_match *__restrict__ out, uint64_t *__restrict__ nkept)
{
  const uint64_t t = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (t >= count)
  {
    return;
  }
  const uint32_t k = keep[t], s = slot[t];
  if (k != 0)
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(in + t);
    uint4 *dst = reinterpret_cast<uint4 *>(out + s);
    const uint4 lo = src[0], hi = src[1];
    dst[0] = lo;
    dst[1] = hi;
  }
  if (t == count - 1)
  {
    *nkept = (uint64_t) s + k;
  }
}

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

namespace
{

struct DevBuf
{
  void *p = nullptr;
  ~DevBuf()
  {
    vsa_dev_free(p);
  }
  int alloc(size_t bytes)
  {
    vsa_dev_free(p);
    p = nullptr;
    return vsa_dev_alloc(&p, bytes > 0 ? bytes : 16);
  }
  template <typename T>
  T *as()
  {
    return (T *) p;
  }
  void *release()
  {
    void *r = p;
    p = nullptr;
    return r;
  }
};

struct Timer
{
  hipEvent_t a = nullptr, b = nullptr;
  hipStream_t s;
  bool started = false, stopped = false;
  explicit Timer(hipStream_t stream) : s(stream)
  {
    (void) hipEventCreate(&a);
    (void) hipEventCreate(&b);
  }
  ~Timer()
  {
    (void) hipEventDestroy(a);
    (void) hipEventDestroy(b);
  }
  void start()
  {
    started = hipEventRecord(a, s) == hipSuccess;
  }
  void stop()
  {
    stopped = hipEventRecord(b, s) == hipSuccess;
  }
  double ms() // after the stream has been synchronised
  {
    // a timer that never ran must not leave an error behind: the runtime
    // keeps the last error, and the next library call would report it
    float f = 0;
    if (!started || !stopped ||
        hipEventElapsedTime(&f, a, b) != hipSuccess)
    {
      (void) hipGetLastError();
      return 0.0;
    }
    return (double) f;
  }
};

inline unsigned int gridfor(uint64_t items)
{
  return (unsigned int) ((items + VSA_BLOCK - 1) / VSA_BLOCK);
}

// out[] = the records of in[] with keep != 0, in order; *nkept (device) = count
int compact_matches(const vsa_match *in, const uint8_t *keep,
                           uint64_t count, vsa_match *out, uint64_t *nkept,
                           hipStream_t stream)
{
  DevBuf slots, temp;
  size_t tb = 0;
  auto keepit = rocprim::make_transform_iterator(keep, KeepToU32());

  if (slots.alloc(count * 4))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(nullptr, tb, keepit, slots.as<uint32_t>(),
                                  (uint32_t) 0, (size_t) count,
                                  rocprim::plus<uint32_t>(), stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(temp.p, tb, keepit, slots.as<uint32_t>(),
                                  (uint32_t) 0, (size_t) count,
                                  rocprim::plus<uint32_t>(), stream));
  k_scatter_kept<<<gridfor(count), VSA_BLOCK, 0, stream>>>(
      in, keep, slots.as<uint32_t>(), count, out, nkept);
  VSA_HIP(hipGetLastError());
  return 0;
}

struct MatchLength
{
  __device__ uint64_t operator()(const vsa_match &m) const
  {
    return m.length;
  }
};

int sumlengths(const vsa_match *matches, uint64_t n, hipStream_t stream,
               uint64_t *result)
{
  *result = 0;
  if (n == 0)
  {
    return 0;
  }
  DevBuf out, temp;
  size_t tb = 0;
  auto in = rocprim::make_transform_iterator(matches, MatchLength());
  if (out.alloc(sizeof(uint64_t)) != 0)
  {
    return -100;
  }
  VSA_HIP(rocprim::reduce(nullptr, tb, in, out.as<uint64_t>(), (uint64_t) 0,
                          (size_t) n, rocprim::plus<uint64_t>(), stream));
  if (temp.alloc(tb) != 0)
  {
    return -100;
  }
  VSA_HIP(rocprim::reduce(temp.p, tb, in, out.as<uint64_t>(), (uint64_t) 0,
                          (size_t) n, rocprim::plus<uint64_t>(), stream));
  VSA_HIP(hipMemcpyAsync(result, out.p, sizeof(uint64_t),
                         hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  return 0;
}

unsigned int bitsfor(uint64_t maxvalue)
{
  unsigned int b = 1;
  while (b < 64 && (maxvalue >> b) != 0)
  {
    b++;
  }
  return b;
}

// stable sort of (key, match) pairs by key bits [0, endbit); results land in
// keys_out / matches_out
int sortbykey(uint64_t *keys_in, uint64_t *keys_out, vsa_match *in,
              vsa_match *out, uint64_t n, unsigned int endbit,
              hipStream_t stream)
{
  DevBuf temp;
  size_t tb = 0;
  VSA_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out, in, out,
                                    (size_t) n, 0u, endbit, stream));
  if (temp.alloc(tb) != 0)
  {
    return -100;
  }
  VSA_HIP(rocprim::radix_sort_pairs(temp.p, tb, keys_in, keys_out, in, out,
                                    (size_t) n, 0u, endbit, stream));
  return 0;
}

vsa_result *newresult(int device)
{
  vsa_result *r = new vsa_result;
  r->device = device;
  r->count = 0;
  r->matches = nullptr;
  memset(&r->stats, 0, sizeof r->stats);
  return r;
}

// ---- K1 pipeline ----

template <typename IDX>
int run_complete(const vsa_index *index, const vsa_queries *queries,
                 uint64_t qlimit, vsa_result *res)
{
  hipStream_t stream = index->stream;
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  const DevQueries qs = devqueries(queries);
  DevBuf left, count, offsets, temp, matches;
  uint64_t total = 0;

  res->stats.searches = qlimit;
  if (qlimit == 0)
  {
    return 0;
  }
  if (left.alloc(qlimit * 8) || count.alloc((qlimit + 1) * 8) ||
      offsets.alloc((qlimit + 1) * 8))
  {
    return -100;
  }
  tall.start();
  VSA_HIP(hipMemsetAsync(count.as<uint64_t>() + qlimit, 0, 8, stream));
  tsearch.start();
  if constexpr (sizeof(IDX) == 4)
  {
    if (ix.esa8 != nullptr)
    {
      k_complete_search<IDX, true><<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
          ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    } else
    {
      k_complete_search<IDX, false>
          <<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
              ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    }
  } else
  {
    k_complete_search<IDX, false><<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
        ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
  }
  tsearch.stop();
  VSA_HIP(hipGetLastError());
  size_t tb = 0;
  VSA_HIP(rocprim::exclusive_scan(nullptr, tb, count.as<uint64_t>(),
                                  offsets.as<uint64_t>(), (uint64_t) 0,
                                  (size_t) (qlimit + 1),
                                  rocprim::plus<uint64_t>(), stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(temp.p, tb, count.as<uint64_t>(),
                                  offsets.as<uint64_t>(), (uint64_t) 0,
                                  (size_t) (qlimit + 1),
                                  rocprim::plus<uint64_t>(), stream));
  VSA_HIP(hipMemcpyAsync(&total, offsets.as<uint64_t>() + qlimit, 8,
                         hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  if (total > 0)
  {
    if (matches.alloc(total * sizeof(vsa_match)))
    {
      return -100;
    }
    k_complete_expand<IDX><<<gridfor(total), VSA_BLOCK, 0, stream>>>(
        ix, qs, qlimit, left.as<uint64_t>(), offsets.as<uint64_t>(), total,
        matches.as<vsa_match>());
    VSA_HIP(hipGetLastError());
  }
  tall.stop();
  VSA_HIP(hipStreamSynchronize(stream));
  res->count = total;
  res->matches = (vsa_match *) matches.release();
  res->stats.count = total;
  res->stats.search_kernel_ms = tsearch.ms();
  res->stats.total_device_ms = tall.ms();
  // every complete match has the length of its query
  return sumlengths(res->matches, total, stream, &res->stats.sumlength);
}

// ---- K2 (+K4) pipeline ----

// MUM candidates, any order -> MUMs in dbstart order
// (max dbstart, max length) of a candidate list
struct MaxPair
{
  uint64_t db, len;
};

struct MaxPairOf
{
  __device__ MaxPair operator()(const vsa_match &m) const
  {
    MaxPair p;
    p.db = m.dbstart;
    p.len = m.length;
    return p;
  }
};

struct MaxPairOp
{
  __device__ MaxPair operator()(const MaxPair &a, const MaxPair &b) const
  {
    MaxPair p;
    p.db = a.db > b.db ? a.db : b.db;
    p.len = a.len > b.len ? a.len : b.len;
    return p;
  }
};

// one key for "dbstart ascending, then length descending"
__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_compositekeys(const vsa_match *__restrict__ cand, uint64_t n,
                    unsigned int lenbits, uint64_t *__restrict__ key,
                    uint32_t *__restrict__ idx)
{
  const uint64_t i = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t lenmask = (1ull << lenbits) - 1;
    key[i] = (cand[i].dbstart << lenbits) | (lenmask - cand[i].length);
    idx[i] = (uint32_t) i;
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_gather(const vsa_match *__restrict__ cand,
             const uint32_t *__restrict__ idx, uint64_t n,
             vsa_match *__restrict__ sorted, uint64_t *__restrict__ rightend)
{
  const uint64_t i = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const vsa_match m = cand[idx[i]];
    sorted[i] = m;
    rightend[i] = m.dbstart + m.length - 1;
  }
}

// carry = the reference's running `dbright` when it reaches the first of
// these candidates: 0 for a whole job, the largest right end of all
// candidates with a smaller dbstart when the list is one dbstart range of a
// job that is filtered in pieces (multi-GPU).  *maxright (optional) receives
// the largest right end in this list.
int mumuniqueinquery(DevBuf &cand, uint64_t ncand, hipStream_t stream,
                     DevBuf &mums, uint64_t *nmums, uint64_t carry = 0,
                     uint64_t *maxright = nullptr)
{
  *nmums = 0;
  if (maxright != nullptr)
  {
    *maxright = 0;
  }
  if (ncand == 0)
  {
    return 0;
  }
  DevBuf ends, dbright, keep, temp, dcount, sorted;
  if (ends.alloc(ncand * 8) || dbright.alloc(ncand * 8) ||
      keep.alloc(ncand) || dcount.alloc(sizeof(MaxPair)) ||
      sorted.alloc(ncand * sizeof(vsa_match)))
  {
    return -100;
  }
  // how many bits do dbstart and length need?
  MaxPair mx;
  {
    size_t tb = 0;
    auto in = rocprim::make_transform_iterator(cand.as<vsa_match>(),
                                               MaxPairOf());
    MaxPair init;
    init.db = init.len = 0;
    VSA_HIP(rocprim::reduce(nullptr, tb, in, dcount.as<MaxPair>(), init,
                            (size_t) ncand, MaxPairOp(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(temp.p, tb, in, dcount.as<MaxPair>(), init,
                            (size_t) ncand, MaxPairOp(), stream));
    VSA_HIP(hipMemcpyAsync(&mx, dcount.p, sizeof mx, hipMemcpyDeviceToHost,
                           stream));
    VSA_HIP(hipStreamSynchronize(stream));
  }
  const unsigned int lenbits = bitsfor(mx.len), dbbits = bitsfor(mx.db);
  if (lenbits + dbbits <= 64 && ncand < 0xFFFFFFFFull)
  {
    // one radix sort of (composite key, index) over just the bits in use,
    // then one gather
    DevBuf k1, k2, i1, i2;
    if (k1.alloc(ncand * 8) || k2.alloc(ncand * 8) || i1.alloc(ncand * 4) ||
        i2.alloc(ncand * 4))
    {
      return -100;
    }
    k_mum_compositekeys<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), ncand, lenbits, k1.as<uint64_t>(),
        i1.as<uint32_t>());
    VSA_HIP(hipGetLastError());
    size_t tb = 0;
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, k1.as<uint64_t>(), k2.as<uint64_t>(), i1.as<uint32_t>(),
        i2.as<uint32_t>(), (size_t) ncand, 0u, lenbits + dbbits, stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, k1.as<uint64_t>(), k2.as<uint64_t>(), i1.as<uint32_t>(),
        i2.as<uint32_t>(), (size_t) ncand, 0u, lenbits + dbbits, stream));
    k_mum_gather<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), i2.as<uint32_t>(), ncand,
        sorted.as<vsa_match>(), ends.as<uint64_t>());
    VSA_HIP(hipGetLastError());
  } else
  {
    // wide values: least significant key first (length descending), then a
    // stable sort by dbstart
    DevBuf k1, k2, kout;
    if (k1.alloc(ncand * 8) || k2.alloc(ncand * 8) || kout.alloc(ncand * 8))
    {
      return -100;
    }
    k_mum_keys<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), ncand, k1.as<uint64_t>(), k2.as<uint64_t>());
    VSA_HIP(hipGetLastError());
    if (sortbykey(k1.as<uint64_t>(), kout.as<uint64_t>(),
                  cand.as<vsa_match>(), sorted.as<vsa_match>(), ncand, 64,
                  stream))
    {
      return -100;
    }
    k_mum_keys<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        sorted.as<vsa_match>(), ncand, k1.as<uint64_t>(), k2.as<uint64_t>());
    VSA_HIP(hipGetLastError());
    if (sortbykey(k2.as<uint64_t>(), kout.as<uint64_t>(),
                  sorted.as<vsa_match>(), cand.as<vsa_match>(), ncand, 64,
                  stream))
    {
      return -100;
    }
    VSA_HIP(hipMemcpyAsync(sorted.p, cand.p, ncand * sizeof(vsa_match),
                           hipMemcpyDeviceToDevice, stream));
    k_mum_rightends<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        sorted.as<vsa_match>(), ncand, ends.as<uint64_t>());
    VSA_HIP(hipGetLastError());
  }
  size_t tb = 0;
  VSA_HIP(rocprim::exclusive_scan(nullptr, tb, ends.as<uint64_t>(),
                                  dbright.as<uint64_t>(), carry,
                                  (size_t) ncand, rocprim::maximum<uint64_t>(),
                                  stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(temp.p, tb, ends.as<uint64_t>(),
                                  dbright.as<uint64_t>(), carry,
                                  (size_t) ncand, rocprim::maximum<uint64_t>(),
                                  stream));
  if (maxright != nullptr)
  {
    // sorted by dbstart, so the running maximum behind the last element
    uint64_t lastend = 0, lastmax = 0;
    VSA_HIP(hipMemcpyAsync(&lastend, ends.as<uint64_t>() + ncand - 1, 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipMemcpyAsync(&lastmax, dbright.as<uint64_t>() + ncand - 1, 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    *maxright = std::max(lastend, lastmax);
  }
  k_mum_flags<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
      sorted.as<vsa_match>(), ends.as<uint64_t>(), dbright.as<uint64_t>(),
      ncand, keep.as<uint8_t>());
  VSA_HIP(hipGetLastError());
  if (mums.alloc(ncand * sizeof(vsa_match)))
  {
    return -100;
  }
  if (compact_matches(sorted.as<vsa_match>(), keep.as<uint8_t>(), ncand,
                      mums.as<vsa_match>(), dcount.as<uint64_t>(), stream))
  {
    return -100;
  }
  VSA_HIP(hipMemcpyAsync(nmums, dcount.p, 8, hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  return 0;
}

template <typename IDX>
int run_query(const vsa_index *index, const vsa_queries *queries, bool domum,
              bool domumcand, uint32_t searchlength, vsa_result *res)
{
  hipStream_t stream = index->stream;
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  const DevQueries qs = devqueries(queries);
  DevBuf base, cursor, out, keys;
  uint64_t nitems = 0;
  uint32_t perquery = 0;
  const uint64_t *dbase = nullptr;

  // work-items: one per query suffix with remaining >= searchlength
  // (kurtz/matchsub.c:187-196: shorter queries are skipped silently)
  if (qs.uniformlen != 0)
  {
    perquery = (qs.uniformlen >= searchlength)
                   ? qs.uniformlen - searchlength + 1
                   : 0;
    nitems = (uint64_t) perquery * queries->nq;
  } else
  {
    std::vector<uint64_t> hb(queries->nq + 1);
    for (uint64_t q = 0; q < queries->nq; q++)
    {
      hb[q] = nitems;
      const uint64_t len = queries->hlength[q];
      nitems += (len >= searchlength) ? len - searchlength + 1 : 0;
    }
    hb[queries->nq] = nitems;
    if (base.alloc(hb.size() * 8))
    {
      return -100;
    }
    VSA_HIP(hipMemcpyAsync(base.p, hb.data(), hb.size() * 8,
                           hipMemcpyHostToDevice, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    dbase = base.as<uint64_t>();
  }
  res->stats.searches = nitems;
  if (nitems == 0)
  {
    return 0;
  }
  const uint32_t nshards = VSA_CURSOR_SHARDS;
  const int qblock = (int) ((index->tune >> 8) & 0xFFF); // experiment switch
  bool deepok = false;
  if constexpr (sizeof(IDX) == 4)
  {
    deepok = ix.esa8 != nullptr && searchlength >= ix.D;
  }
  tall.start();
  // MUM modes over batches of equal-length queries: anchor pass + work list
  DevBuf wcount, wbase, wlq, wloff, wtemp, wplan, wlist, wnlist, wfirste,
      wfmlen, wfmdb, wfslot;
  uint64_t plansearches = 0, nfirst = 0;
  bool firstpass = false;
  const uint32_t *dwlq = nullptr, *dwloff = nullptr;
  uint64_t nwork = nitems;
  double anchorms = 0;
  if (domum && qs.uniformlen != 0 && perquery > 1 &&
      queries->nq < 0xFFFFFFFFull && (index->tune & 2u) == 0)
  {
    const uint64_t nq = queries->nq;
    Timer tanchor(stream);
    if (wcount.alloc((nq + 1) * 4) || wbase.alloc((nq + 1) * 8))
    {
      return -100;
    }
    tanchor.start();
    VSA_HIP(hipMemsetAsync(wcount.as<uint32_t>() + nq, 0, 4, stream));
    // work plan (see k_mum_plan); the quirk of the reference's uniqueness
    // test for lcp >= 255 (fquery.c:352) keeps longer queries out
    const bool planned = qs.uniformlen < 255 && perquery < 0xFFFFu &&
                         (index->tune & 4u) == 0;
    // planned batches start with offset 0 (k_mum_first), the others with the
    // anchor pass from the last offset
    firstpass = planned && (index->tune & 8u) == 0;
    if (firstpass)
    {
      if (wfirste.alloc(nq * 4) || wfmlen.alloc(nq * 4) ||
          wfmdb.alloc(nq * 8))
      {
        return -100;
      }
      if (deepok)
      {
        if constexpr (sizeof(IDX) == 4)
        {
          k_mum_first<IDX, true><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
              ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
              wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
              wfmdb.as<uint64_t>());
        }
      } else
      {
        k_mum_first<IDX, false><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
            ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
            wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
            wfmdb.as<uint64_t>());
      }
    } else if (deepok)
    {
      if constexpr (sizeof(IDX) == 4)
      {
        k_mum_anchor<IDX, true><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
            ix, qs, perquery, searchlength, wcount.as<uint32_t>());
      }
    } else
    {
      k_mum_anchor<IDX, false><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          ix, qs, perquery, searchlength, wcount.as<uint32_t>());
    }
    VSA_HIP(hipGetLastError());
    size_t tb = 0;
    if (planned)
    {
      uint64_t nlist = 0;
      // after the anchor pass only queries with many offsets left are worth
      // a plan; after the first pass every unfinished query gets one
      PlanWanted wanted{wcount.as<uint32_t>(),
                        firstpass ? 0u : searchlength + 6};
      if (wplan.alloc(nq * sizeof(PlanRanges)) || wlist.alloc(nq * 4) ||
          wnlist.alloc(8))
      {
        return -100;
      }
      k_plan_default<<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          wcount.as<uint32_t>(), nq, wplan.as<PlanRanges>());
      VSA_HIP(hipGetLastError());
      VSA_HIP(rocprim::select(nullptr, tb,
                              rocprim::counting_iterator<uint32_t>(0),
                              wlist.as<uint32_t>(), wnlist.as<uint64_t>(),
                              (size_t) nq, wanted, stream));
      if (wtemp.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::select(wtemp.p, tb,
                              rocprim::counting_iterator<uint32_t>(0),
                              wlist.as<uint32_t>(), wnlist.as<uint64_t>(),
                              (size_t) nq, wanted, stream));
      VSA_HIP(hipMemcpyAsync(&nlist, wnlist.p, 8, hipMemcpyDeviceToHost,
                             stream));
      VSA_HIP(hipStreamSynchronize(stream));
      if (nlist > 0)
      {
        if (deepok)
        {
          if constexpr (sizeof(IDX) == 4)
          {
            k_mum_plan<IDX, true><<<gridfor(nlist), VSA_BLOCK, 0, stream>>>(
                ix, qs, wlist.as<uint32_t>(), nlist, searchlength,
                firstpass ? wfirste.as<uint32_t>() : nullptr,
                wcount.as<uint32_t>(), wplan.as<PlanRanges>());
          }
        } else
        {
          k_mum_plan<IDX, false><<<gridfor(nlist), VSA_BLOCK, 0, stream>>>(
              ix, qs, wlist.as<uint32_t>(), nlist, searchlength,
              firstpass ? wfirste.as<uint32_t>() : nullptr,
              wcount.as<uint32_t>(), wplan.as<PlanRanges>());
        }
        VSA_HIP(hipGetLastError());
      }
      plansearches = 2 * nlist;
    }
    tb = 0;
    auto widen = rocprim::make_transform_iterator(
        wcount.as<uint32_t>(),
        [] __device__(uint32_t v) { return (uint64_t) v; });
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, widen, wbase.as<uint64_t>(),
                                    (uint64_t) 0, (size_t) (nq + 1),
                                    rocprim::plus<uint64_t>(), stream));
    if (wtemp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(wtemp.p, tb, widen, wbase.as<uint64_t>(),
                                    (uint64_t) 0, (size_t) (nq + 1),
                                    rocprim::plus<uint64_t>(), stream));
    VSA_HIP(hipMemcpyAsync(&nwork, wbase.as<uint64_t>() + nq, 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    if (wlq.alloc(nwork * 4) || wloff.alloc(nwork * 4))
    {
      return -100;
    }
    if (planned)
    {
      k_expand_plan<<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          wplan.as<PlanRanges>(), wbase.as<uint64_t>(), nq,
          wlq.as<uint32_t>(), wloff.as<uint32_t>());
    } else
    {
      k_expand_worklist<<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          wcount.as<uint32_t>(), wbase.as<uint64_t>(), nq,
          wlq.as<uint32_t>(), wloff.as<uint32_t>());
    }
    VSA_HIP(hipGetLastError());
    tanchor.stop();
    VSA_HIP(hipStreamSynchronize(stream));
    anchorms = tanchor.ms();
    dwlq = wlq.as<uint32_t>();
    dwloff = wloff.as<uint32_t>();
    res->stats.searches = nwork + nq + plansearches;
  }
  std::vector<uint64_t> hcur(nshards * VSA_CURSOR_STRIDE), hoff(nshards);
  DevBuf doff, rawout, rawkeys;
  if (cursor.alloc(hcur.size() * 8) || doff.alloc(nshards * 8))
  {
    return -100;
  }
  // first guess: MUM modes report at most one match per work-item but
  // typically about one per query; MEM is unbounded.  The kernel counts what
  // it needs and never writes past a region's capacity; on overflow of any
  // region run again with regions of the size that was asked for.
  uint64_t shardcap =
      std::max<uint64_t>((queries->nq * 2 / nshards) * 5 / 4 + 64, 256);
  uint64_t needed = 0, maxshard = 0;
  double searchms = 0;
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (rawout.alloc(nshards * shardcap * sizeof(vsa_match)) ||
        rawkeys.alloc(nshards * shardcap * 8))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(cursor.p, 0, hcur.size() * 8, stream));
    tsearch.start();
#define VSA_LAUNCH_QUERY_B(MUMFLAG, KEYFLAG, BLK)                              \
  k_query_search<IDX, MUMFLAG, KEYFLAG, BLK>                                  \
      <<<(unsigned int) ((nwork + BLK - 1) / BLK), BLK, 0, stream>>>(         \
          ix, qs, dbase, perquery, dwlq, dwloff, nwork, searchlength,         \
          rawout.as<vsa_match>(), rawkeys.as<uint64_t>(), shardcap,           \
          nshards - 1, cursor.as<unsigned long long>())
#define VSA_LAUNCH_QUERY(MUMFLAG, KEYFLAG)                                     \
  do                                                                          \
  {                                                                           \
    if (qblock == 64)                                                         \
    {                                                                         \
      VSA_LAUNCH_QUERY_B(MUMFLAG, KEYFLAG, 64);                               \
    } else if (qblock == 128)                                                 \
    {                                                                         \
      VSA_LAUNCH_QUERY_B(MUMFLAG, KEYFLAG, 128);                              \
    } else if (qblock == 512)                                                 \
    {                                                                         \
      VSA_LAUNCH_QUERY_B(MUMFLAG, KEYFLAG, 512);                              \
    } else                                                                    \
    {                                                                         \
      VSA_LAUNCH_QUERY_B(MUMFLAG, KEYFLAG, 256);                              \
    }                                                                         \
  } while (0)
    // deep locate needs the deep prefix to fit into every search
    bool deep = nwork == 0; // nothing left to search: no launch at all
    if constexpr (sizeof(IDX) == 4)
    {
      deep = deep || deepok;
      if (deep && nwork > 0)
      {
        if (domum)
        {
          VSA_LAUNCH_QUERY(true, true);
        } else
        {
          VSA_LAUNCH_QUERY(false, true);
        }
      }
    }
    if (!deep)
    {
      if (domum)
      {
        VSA_LAUNCH_QUERY(true, false);
      } else
      {
        VSA_LAUNCH_QUERY(false, false);
      }
    }
#undef VSA_LAUNCH_QUERY
#undef VSA_LAUNCH_QUERY_B
    tsearch.stop();
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipMemcpyAsync(hcur.data(), cursor.p, hcur.size() * 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    searchms += tsearch.ms();
    needed = maxshard = 0;
    for (uint32_t sh = 0; sh < nshards; sh++)
    {
      const uint64_t cnt = hcur[(uint64_t) sh * VSA_CURSOR_STRIDE];
      hoff[sh] = needed;
      needed += cnt;
      maxshard = std::max(maxshard, cnt);
    }
    if (maxshard <= shardcap)
    {
      break;
    }
    shardcap = maxshard;
  }
  if (maxshard > shardcap)
  {
    VSA_ERROR("match buffer overflow: %llu > %llu",
              (unsigned long long) maxshard, (unsigned long long) shardcap);
    return -5;
  }
  if (firstpass)
  {
    // slots of the first-pass candidates behind the kernel's matches
    const uint64_t nq = queries->nq;
    size_t tb = 0;
    uint32_t lastslot = 0, lastlen = 0;
    auto flag = rocprim::make_transform_iterator(wfmlen.as<uint32_t>(),
                                                 NonZeroToU32());
    if (wfslot.alloc(nq * 4))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, flag, wfslot.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) nq,
                                    rocprim::plus<uint32_t>(), stream));
    if (wtemp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(wtemp.p, tb, flag, wfslot.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) nq,
                                    rocprim::plus<uint32_t>(), stream));
    VSA_HIP(hipMemcpyAsync(&lastslot, wfslot.as<uint32_t>() + nq - 1, 4,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipMemcpyAsync(&lastlen, wfmlen.as<uint32_t>() + nq - 1, 4,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    nfirst = (uint64_t) lastslot + (lastlen != 0 ? 1 : 0);
  }
  if (needed + nfirst > 0)
  {
    if (out.alloc((needed + nfirst) * sizeof(vsa_match)) ||
        keys.alloc((needed + nfirst) * 8))
    {
      return -100;
    }
    if (needed > 0)
    {
      VSA_HIP(hipMemcpyAsync(doff.p, hoff.data(), nshards * 8,
                             hipMemcpyHostToDevice, stream));
      k_compact_shards<<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<vsa_match>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<vsa_match>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    }
    if (nfirst > 0)
    {
      k_append_first<<<gridfor(queries->nq), VSA_BLOCK, 0, stream>>>(
          wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>(),
          wfslot.as<uint32_t>(), queries->nq, perquery, qs.seqoffset, needed,
          out.as<vsa_match>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    }
    needed += nfirst;
  }
  res->stats.candidates = domum ? needed : 0;
  if (domum && !domumcand)
  {
    DevBuf mums;
    uint64_t nm = 0;
    if (mumuniqueinquery(out, needed, stream, mums, &nm))
    {
      return -100;
    }
    res->count = nm;
    res->matches = (vsa_match *) mums.release();
  } else if (needed > 0)
  {
    // reference order = work-item order; appends of one work-item are
    // contiguous and in order, the radix sort is stable
    DevBuf sk, sm;
    if (sk.alloc(needed * 8) || sm.alloc(needed * sizeof(vsa_match)))
    {
      return -100;
    }
    if (sortbykey(keys.as<uint64_t>(), sk.as<uint64_t>(),
                  out.as<vsa_match>(), sm.as<vsa_match>(), needed,
                  bitsfor(nitems), stream))
    {
      return -100;
    }
    res->count = needed;
    res->matches = (vsa_match *) sm.release();
  }
  tall.stop();
  VSA_HIP(hipStreamSynchronize(stream));
  res->stats.count = res->count;
  res->stats.search_kernel_ms = searchms;
  res->stats.anchor_ms = anchorms;
  res->stats.kernel_searches = nwork;
  res->stats.total_device_ms = tall.ms();
  return sumlengths(res->matches, res->count, stream, &res->stats.sumlength);
}

// ---- K3 pipeline ----

// VSA_PEAKVARIANT (experiments): bit 0 = nontemporal loads, bit 1 = tiles of
// 2 instead of 4 pieces.  Measured at 3 Gbp (dense case, 1024 workgroups):
// 1.13 ms plain, 1.09 ms nontemporal, 1.40 / 1.35 ms with 2 pieces.
static int vsa_peakvariant(void)
{
  const char *e = getenv("VSA_PEAKVARIANT");
  return (e != NULL) ? atoi(e) & 3 : 1;
}

// workgroups of the streaming pass: 4 per CU = all the wavefronts that fit
// with 100 VGPRs each, every one walking its tiles grid-stride
// (VSA_PEAKBLOCKS overrides, for experiments)
static uint64_t vsa_peakblocks(void)
{
  const char *e = getenv("VSA_PEAKBLOCKS");
  const long v = (e != NULL) ? atol(e) : 0;
  return v > 0 ? (uint64_t) v : 1024;
}

template <typename IDX>
int run_selfmum(const vsa_index *index, uint64_t searchlength,
                vsa_result *res)
{
  hipStream_t stream = index->stream;
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  const uint64_t n = index->n;
  const uint32_t nshards = VSA_CURSOR_SHARDS;
  const int variant = vsa_peakvariant();
  const uint64_t pieces = (variant & 2) ? 2 : 4, tilesize = 64 * pieces * 16;
  const uint64_t ntiles = (n + 1 + tilesize - 1) / tilesize,
                 wavesperblock = VSA_BLOCK / 64;
  const uint64_t nblocks =
      std::min<uint64_t>((ntiles + wavesperblock - 1) / wavesperblock,
                         (uint64_t) vsa_peakblocks());
  const uint32_t slmin = (uint32_t) (searchlength < 255 ? searchlength : 255);
  std::vector<uint64_t> hcur(nshards * VSA_CURSOR_STRIDE), hoff(nshards);
  DevBuf cursor, doff, rawpos, peaks, sorted, temp, cand, keep, dcount, mums;
  uint64_t shardcap = std::max<uint64_t>(n / 64 / nshards + 1024, 4096),
           needed = 0, maxshard = 0;
  double searchms = 0;

  res->stats.searches = n;
  if (n + 1 >= 0xFFFFFFFFull)
  {
    VSA_ERROR("self-index MUM scan: texts beyond 2^32 are not supported");
    return -3;
  }
  if (cursor.alloc(hcur.size() * 8) || doff.alloc(nshards * 8) ||
      dcount.alloc(8))
  {
    return -100;
  }
  tall.start();
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (rawpos.alloc(nshards * shardcap * 4))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(cursor.p, 0, hcur.size() * 8, stream));
    tsearch.start();
#define VSA_PEAKS(PIECES, NT)                                                 \
  k_selfmum_peaks<PIECES, NT><<<(unsigned int) nblocks, VSA_BLOCK, 0,         \
                                stream>>>(                                    \
      ix.lcp, ix.bwt, n, slmin, rawpos.as<uint32_t>(), shardcap, nshards - 1, \
      cursor.as<unsigned long long>(), ntiles)
    switch (variant)
    {
      case 1: VSA_PEAKS(4, true); break;
      case 2: VSA_PEAKS(2, false); break;
      case 3: VSA_PEAKS(2, true); break;
      default: VSA_PEAKS(4, false); break;
    }
#undef VSA_PEAKS
    tsearch.stop();
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipMemcpyAsync(hcur.data(), cursor.p, hcur.size() * 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    searchms = tsearch.ms(); // the streaming pass (of the last attempt)
    needed = maxshard = 0;
    for (uint32_t sh = 0; sh < nshards; sh++)
    {
      const uint64_t cnt = hcur[(uint64_t) sh * VSA_CURSOR_STRIDE];
      hoff[sh] = needed;
      needed += cnt;
      maxshard = std::max(maxshard, cnt);
    }
    if (maxshard <= shardcap)
    {
      break;
    }
    shardcap = maxshard;
  }
  if (maxshard > shardcap)
  {
    VSA_ERROR("peak buffer overflow");
    return -5;
  }
  uint64_t nm = 0;
  if (needed > 0)
  {
    if (peaks.alloc(needed * 4) || sorted.alloc(needed * 4) ||
        cand.alloc(needed * sizeof(vsa_match)) || keep.alloc(needed) ||
        mums.alloc(needed * sizeof(vsa_match)))
    {
      return -100;
    }
    VSA_HIP(hipMemcpyAsync(doff.p, hoff.data(), nshards * 8,
                           hipMemcpyHostToDevice, stream));
    k_gather_u32_shards<<<nshards, VSA_BLOCK, 0, stream>>>(
        rawpos.as<uint32_t>(), shardcap, cursor.as<unsigned long long>(),
        doff.as<uint64_t>(), peaks.as<uint32_t>());
    VSA_HIP(hipGetLastError());
    // the reference reports in suffix-array order
    size_t tb = 0;
    VSA_HIP(rocprim::radix_sort_keys(nullptr, tb, peaks.as<uint32_t>(),
                                     sorted.as<uint32_t>(), (size_t) needed,
                                     0u, bitsfor(n), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_keys(temp.p, tb, peaks.as<uint32_t>(),
                                     sorted.as<uint32_t>(), (size_t) needed,
                                     0u, bitsfor(n), stream));
    k_selfmum_emit<IDX><<<gridfor(needed), VSA_BLOCK, 0, stream>>>(
        ix, sorted.as<uint32_t>(), needed, searchlength,
        index->querysepposition, cand.as<vsa_match>(), keep.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    if (compact_matches(cand.as<vsa_match>(), keep.as<uint8_t>(), needed,
                        mums.as<vsa_match>(), dcount.as<uint64_t>(), stream))
    {
      return -100;
    }
    VSA_HIP(hipMemcpyAsync(&nm, dcount.p, 8, hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    res->count = nm;
    res->matches = (vsa_match *) mums.release();
  }
  tall.stop();
  VSA_HIP(hipStreamSynchronize(stream));
  res->stats.count = res->count;
  res->stats.candidates = needed;
  res->stats.search_kernel_ms = searchms;
  res->stats.total_device_ms = tall.ms();
  return sumlengths(res->matches, res->count, stream, &res->stats.sumlength);
}

#include "approx_search.inc"
#include "selfmatch_search.inc"

} // namespace

// ---------------------------------------------------------------------------
// the keyed search array (see DevIndex::esa8)
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(VSA_BLOCK)
k_make_esa8(const uint8_t *__restrict__ tis, const uint32_t *__restrict__ suf,
            const uint8_t *__restrict__ lcp, uint64_t count, uint32_t D,
            uint64_t *__restrict__ esa8)
{
  const uint64_t j = (uint64_t) blockIdx.x * VSA_BLOCK + threadIdx.x;
  if (j >= count)
  {
    return;
  }
  const uint32_t s = suf[j];
  const uint8_t *t = tis + (uint64_t) s + D; // padded with 0xFF behind n
  uint64_t key = 0, flag = 0;
#pragma unroll
  for (uint32_t k = 0; k < VSA_KEYSYMS; k++)
  {
    const uint8_t a = t[k];
    if (VSA_ISSPECIAL(a))
    {
      flag = VSA_KEYFLAG;
    }
    key = (key << 2) | (a & 3);
  }
  esa8[j] = (uint64_t) s | ((uint64_t) lcp[j] << 32) |
            (key << VSA_KEYSHIFT) | flag;
}

int vsa_index_make_esa8(vsa_index *ix)
{
  const char *off = getenv("VSA_NO_ESA8");
  if (ix->esa8 != nullptr)
  {
    (void) hipFree(ix->esa8);
    ix->esa8 = nullptr;
  }
  if (ix->bck2 != nullptr)
  {
    (void) hipFree(ix->bck2);
    ix->bck2 = nullptr;
  }
  if (ix->numofchars != 4 || ix->isize != 4 || ix->bck == nullptr ||
      (off != nullptr && strcmp(off, "1") == 0))
  {
    return 0;
  }
  // deep prefix: about one suffix per bucket, never shorter than the
  // reference's prefixlength; the table takes 8 * 4^D bytes (at most 32n)
  // D = ceil(log4(n)), at most 16: about one suffix per bucket
  uint32_t D = 1;
  while (D < 16 && (1ull << (2 * D)) < ix->n)
  {
    D++;
  }
  if (D < ix->pl)
  {
    D = ix->pl;
  }
  const char *fd = getenv("VSA_DEEP_PREFIX");
  if (fd != nullptr && atoi(fd) >= (int) ix->pl && atoi(fd) <= 16)
  {
    D = (uint32_t) atoi(fd);
  }
  if (D > 16)
  {
    return 0;
  }
  static_assert(VSA_TIS_BACKPAD >= 16 + VSA_KEYSYMS + 8, "text pad too small");
  ix->D = D;
  const char *tune = getenv("VSA_TUNE");
  ix->tune = tune != nullptr ? (uint32_t) atoi(tune) : 0;
  const uint64_t count = ix->n + 1, ncodes = 1ull << (2 * D);
  VSA_HIP(hipMalloc((void **) &ix->bck2, 2 * ncodes * 4 + 16));
  VSA_HIP(hipMalloc((void **) &ix->esa8, count * 8 + 64));
  ix->device_bytes += count * 8 + 2 * ncodes * 4;
  if (vsa_build_bucket_table(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                             (const uint32_t *) ix->suf, D, 4, ix->bck2,
                             ix->stream))
  {
    return -100;
  }
  k_make_esa8<<<gridfor(count), VSA_BLOCK, 0, ix->stream>>>(
      ix->tis_alloc + VSA_TIS_FRONTPAD, (const uint32_t *) ix->suf, ix->lcp,
      count, D, ix->esa8);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipStreamSynchronize(ix->stream));
  return 0;
}

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" int vsa_findcompletematches(const vsa_index *index,
                                       const vsa_queries *queries,
                                       vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findcompletematches: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (queries->device != index->device)
  {
    VSA_ERROR("queries live on device %d, index on device %d",
              queries->device, index->device);
    return -1;
  }
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (queries->maxlength > 0xFFFFFFF0ull)
  {
    VSA_ERROR("query length beyond 32 bits is not supported");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  // Vmengine/exactcompl.c:179-185: the first query shorter than
  // prefixlength stops the run; queries before it are still matched
  uint64_t qlimit = queries->nq;
  bool shortquery = false;
  if (queries->minlength < index->pl)
  {
    for (uint64_t q = 0; q < queries->nq; q++)
    {
      if (queries->hlength[q] < index->pl)
      {
        qlimit = q;
        shortquery = true;
        break;
      }
    }
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_complete<uint32_t>(index, queries, qlimit, res)
                     : run_complete<uint64_t>(index, queries, qlimit, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  if (shortquery)
  {
    VSA_ERROR("patternlength=%lu must be >= %lu=prefixlen",
              (unsigned long) queries->hlength[qlimit],
              (unsigned long) index->pl);
    return -2;
  }
  return 0;
}

extern "C" int vsa_findapproxcompletematches(const vsa_index *index,
                                             const vsa_queries *queries,
                                             int doedist, uint64_t distvalue,
                                             int percent,
                                             vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findapproxcompletematches: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (index->numofchars != 4)
  {
    VSA_ERROR("approximate search on alphabets of %lu symbols is not covered "
              "by the GPU engine", (unsigned long) index->numofchars);
    return VSA_NOT_COVERED;
  }
  ApmPlan plan;
  int rc = apm_plan(index, queries, doedist != 0, distvalue, percent != 0,
                    plan);
  if (rc != 0)
  {
    return rc;
  }
  if (plan.allexact)
  {
    // approxcompl.c:167-176: threshold 0 is the exact search
    return vsa_findcompletematches(index, queries, result);
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  rc = (index->isize == 4)
           ? run_approx<uint32_t>(index, queries, doedist != 0, plan, res)
           : run_approx<uint64_t>(index, queries, doedist != 0, plan, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  if (plan.qlimit < queries->nq)
  {
    // splitesaapm.c:496-501; the queries before it have been answered
    VSA_ERROR("threshold=%lu>=%lu=patternlen not allowed",
              (unsigned long) plan.failk, (unsigned long) plan.failm);
    return -2;
  }
  return 0;
}

extern "C" int vsa_findquerymatches(const vsa_index *index,
                                    const vsa_queries *queries,
                                    int domaximaluniquematch,
                                    int domaximaluniquematchcandidates,
                                    uint64_t searchlength,
                                    vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findquerymatches: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (queries->device != index->device)
  {
    VSA_ERROR("queries live on device %d, index on device %d",
              queries->device, index->device);
    return -1;
  }
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  // Vmengine/fquery.c:440-446
  if (searchlength < index->pl)
  {
    VSA_ERROR("searchlength=%lu must be >= %lu=prefixlen",
              (unsigned long) searchlength, (unsigned long) index->pl);
    return -2;
  }
  if (searchlength > 0xFFFFFFFFull || queries->maxlength > 0xFFFFFFF0ull)
  {
    VSA_ERROR("query or search length beyond 32 bits is not supported");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const bool mum = domaximaluniquematch != 0,
             cand = domaximaluniquematchcandidates != 0;
  const int rc =
      (index->isize == 4)
          ? run_query<uint32_t>(index, queries, mum, cand,
                                (uint32_t) searchlength, res)
          : run_query<uint64_t>(index, queries, mum, cand,
                                (uint32_t) searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

static int mumfilter_entry(void *device_candidates, uint64_t ncandidates,
                           int device, uint64_t carry, vsa_result **result)
{
  if (result == nullptr || (device_candidates == nullptr && ncandidates > 0))
  {
    VSA_ERROR("vsa_mumuniqueinquery: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(device);
  hipStream_t stream = nullptr; // default stream: no index handle here
  Timer tall(stream);
  DevBuf cand, mums;
  cand.p = device_candidates; // borrowed, released below
  uint64_t nm = 0;
  tall.start();
  const int rc = mumuniqueinquery(cand, ncandidates, stream, mums, &nm, carry);
  tall.stop();
  cand.release();
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  (void) hipStreamSynchronize(stream);
  res->count = nm;
  res->matches = (vsa_match *) mums.release();
  res->stats.count = nm;
  res->stats.candidates = ncandidates;
  res->stats.total_device_ms = tall.ms();
  if (sumlengths(res->matches, nm, stream, &res->stats.sumlength) != 0)
  {
    vsa_result_free(res);
    return -100;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_mumuniqueinquery(void *device_candidates,
                                    uint64_t ncandidates, int device,
                                    vsa_result **result)
{
  return mumfilter_entry(device_candidates, ncandidates, device, 0, result);
}

extern "C" int vsa_mumuniqueinquery_range(void *device_candidates,
                                          uint64_t ncandidates, int device,
                                          uint64_t carry_dbright,
                                          vsa_result **result)
{
  return mumfilter_entry(device_candidates, ncandidates, device,
                         carry_dbright, result);
}

extern "C" int vsa_findmaximalrepeats(const vsa_index *index,
                                      uint64_t searchlength,
                                      vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmaximalrepeats: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->n < 2)
  {
    // Vmengine/fself.c:246-250
    VSA_ERROR("repeat search requires a sequence of length >= 2");
    return -2;
  }
  if (index->bwt == nullptr)
  {
    VSA_ERROR("table bwt is not loaded");
    return -3;
  }
  if (index->numofchars > VSA_REP_MAXC)
  {
    VSA_ERROR("maximal repeats on alphabets of %lu symbols are not covered "
              "by the GPU engine", (unsigned long) index->numofchars);
    return VSA_NOT_COVERED;
  }
  if (searchlength == 0)
  {
    VSA_ERROR("maximal repeats need a length of at least 1");
    return -2;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_repeats<uint32_t>(index, searchlength, res)
                     : run_repeats<uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findsupermaximalrepeats(const vsa_index *index,
                                           uint64_t searchlength,
                                           vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findsupermaximalrepeats: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->hasindexedqueries)
  {
    // Vmengine/fself.c:193-198
    VSA_ERROR("supermaximal repeat search does not allow query files in "
              "index");
    return -2;
  }
  if (index->n < 2)
  {
    // Vmengine/fself.c:246-250
    VSA_ERROR("repeat search requires a sequence of length >= 2");
    return -2;
  }
  if (index->bwt == nullptr)
  {
    VSA_ERROR("table bwt is not loaded");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_supermax<uint32_t>(index, searchlength, res)
                     : run_supermax<uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findmaximaluniquematches(const vsa_index *index,
                                            uint64_t searchlength,
                                            vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmaximaluniquematches: NULL argument");
    return -1;
  }
  *result = nullptr;
  // Vmengine/fmumself.c:21-31
  if (!index->hasindexedqueries)
  {
    VSA_ERROR("maximal unique matches search requires at least one query "
              "file");
    return -1;
  }
  if (index->n < 2)
  {
    VSA_ERROR("search for maximal unique matches requires at least a table "
              "of length 2");
    return -2;
  }
  if (index->bwt == nullptr)
  {
    VSA_ERROR("table bwt is not loaded");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_selfmum<uint32_t>(index, searchlength, res)
                     : run_selfmum<uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}
