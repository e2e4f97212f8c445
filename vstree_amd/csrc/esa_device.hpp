// Device-side primitives of the enhanced-suffix-array search, CDNA4 (gfx950).
//
// One work-item owns one search (one query for -complete, one query suffix
// for -l / -mum): the path is a chain of dependent random reads
// (bck -> suf -> tis -> ... -> lcp), so throughput comes from the number of
// independent chains in flight, 64 per wavefront and thousands of wavefronts
// per launch, not from lanes cooperating on one chain.  Text and query are
// compared eight symbols per 64-bit load.
//
// Semantics follow the reference exactly (paths relative to
// /root/reference/src); each function names its counterpart.
#pragma once

#include "vsa_internal.hpp"

#define VSA_ISSPECIAL(c) ((c) >= (uint8_t) VSA_WILDCARD) // chardef.h:37

__device__ __forceinline__ uint64_t vsa_load8(const uint8_t *p)
{
  uint64_t v;
  __builtin_memcpy(&v, p, 8); // one global_load_dwordx2, any alignment
  return v;
}

// 0x80 in every byte of v that is a special symbol (>= 254)
__device__ __forceinline__ uint64_t vsa_specialmask(uint64_t v)
{
  return ((v & 0x7F7F7F7F7F7F7F7Full) + 0x0202020202020202ull) & v &
         0x8080808080808080ull;
}

// Macro pair COMPARE / CHECKRETURN, kurtz/maxpref.c:30-65: continue the
// comparison of query[lcplen..] with the suffix at sufstart; returns the sign
// of the reference's retcode (0 query exhausted, <0 query smaller or suffix
// hit a special symbol / the end of the text, >0 query larger) and leaves the
// matched length in lcplen.  Positions >= n of the device text hold 0xFF.
template <typename IDX>
__device__ __forceinline__ int vsa_compare(const DevIndex<IDX> &ix,
                                           uint64_t sufstart,
                                           const uint8_t *query,
                                           uint32_t querylen,
                                           uint32_t &lcplen)
{
  const uint8_t *t = ix.tis + sufstart;
  uint32_t l = lcplen;

  for (;;)
  {
    if (l >= querylen)
    {
      lcplen = querylen;
      return 0;
    }
    const uint64_t a = vsa_load8(query + l), b = vsa_load8(t + l);
    const uint64_t m = (a ^ b) | vsa_specialmask(a) | vsa_specialmask(b);
    if (m == 0)
    {
      l += 8;
      continue;
    }
    const uint32_t j = (uint32_t) __builtin_ctzll(m) >> 3;
    l += j;
    if (l >= querylen)
    {
      lcplen = querylen;
      return 0;
    }
    lcplen = l;
    const int qa = (int) ((a >> (8 * j)) & 0xFF),
              tb = (int) ((b >> (8 * j)) & 0xFF);
    return (qa == tb) ? -1 : qa - tb; // special == special: "query smaller"
  }
}

// ---- keyed probes ----------------------------------------------------------
//
// With the search array esa8 a probe of the in-bucket binary search is ONE
// 8-byte load: suffix start, lcp byte and the VSA_KEYSYMS symbols that follow
// the bucket prefix.  The comparison the reference makes on the text
// (COMPARE) is answered from those symbols whenever it ends inside them --
// in random-like sequence nearly always, except for the suffix that really
// matches -- and continues on the text otherwise.  Result and probe order
// are unchanged; only where the bytes come from differs.

struct QueryKey
{
  uint32_t key;   // query symbols [pl, pl+VSA_KEYSYMS) packed like esa8's key
  uint32_t valid; // how many of them are regular symbols inside the query
};

template <typename IDX>
__device__ __forceinline__ QueryKey
vsa_querykey(const DevIndex<IDX> &ix, const uint8_t *query, uint32_t querylen)
{
  QueryKey qk;
  qk.key = 0;
  qk.valid = 0;
  bool open = true;
#pragma unroll
  for (uint32_t k = 0; k < VSA_KEYSYMS; k++)
  {
    const uint32_t p = ix.pl + k;
    uint32_t c = 0;
    if (open && p < querylen)
    {
      const uint8_t a = query[p];
      if (VSA_ISSPECIAL(a))
      {
        open = false;
      } else
      {
        c = a & 3;
        qk.valid = k + 1;
      }
    } else
    {
      open = false;
    }
    qk.key = (qk.key << 2) | c;
  }
  return qk;
}

// probe of suffix-array entry i: keyed when esa8 is there, text otherwise
template <typename IDX, bool KEYED>
__device__ __forceinline__ int
vsa_probe(const DevIndex<IDX> &ix, uint64_t i, const uint8_t *query,
          uint32_t querylen, const QueryKey &qk, uint32_t &lcplen)
{
  if (!KEYED)
  {
    return vsa_compare(ix, (uint64_t) ix.suf[i], query, querylen, lcplen);
  }
  const uint64_t e = ix.esa8[i];
  if ((e & VSA_KEYFLAG) == 0 && lcplen >= ix.pl)
  {
    const uint32_t d = lcplen - ix.pl;
    const uint32_t limit = qk.valid; // <= VSA_KEYSYMS
    if (d < limit)
    {
      const uint32_t tk = (uint32_t) (e >> VSA_KEYSHIFT) & VSA_KEYMASK;
      // symbol p sits in bits [21-2p, 20-2p]; drop the positions < d
      const uint32_t x = (tk ^ qk.key) & ((1u << (22 - 2 * d)) - 1u);
      if (x != 0)
      {
        const uint32_t p = ((uint32_t) __builtin_clz(x) - 10u) >> 1;
        if (p < limit)
        {
          lcplen = ix.pl + p;
          const uint32_t sh = 20 - 2 * p;
          return (int) ((qk.key >> sh) & 3) - (int) ((tk >> sh) & 3);
        }
      }
      lcplen = ix.pl + limit; // equal as far as the key (or the query) goes
    }
  }
  return vsa_compare(ix, e & 0xFFFFFFFFull, query, querylen, lcplen);
}

template <typename IDX, bool KEYED>
__device__ __forceinline__ uint64_t vsa_sufstart(const DevIndex<IDX> &ix,
                                                 uint64_t i)
{
  return KEYED ? (ix.esa8[i] & 0xFFFFFFFFull) : (uint64_t) ix.suf[i];
}

template <typename IDX, bool KEYED>
__device__ __forceinline__ uint32_t vsa_lcpbyte(const DevIndex<IDX> &ix,
                                                uint64_t i)
{
  return KEYED ? (uint32_t) (ix.esa8[i] >> 32) & 0xFFu : (uint32_t) ix.lcp[i];
}

// kurtz/findmaxpref.gen:1-96 (instantiated kurtz/maxpref.c:74-87): lcp-aware
// binary search over suf[vleft..vright]; all suffixes there share `offset`
// symbols with the query.  The probe sequence is the reference's, so the
// witness is the reference's witness.
template <typename IDX, bool KEYED>
__device__ __forceinline__ void
vsa_findmaxprefixlen(const DevIndex<IDX> &ix, uint64_t vleft, uint64_t vright,
                     uint32_t offset, const uint8_t *query, uint32_t querylen,
                     uint32_t &maxlcp, uint64_t &witness)
{
  QueryKey qk;
  if (KEYED)
  {
    qk = vsa_querykey(ix, query, querylen);
  } else
  {
    qk.key = qk.valid = 0;
  }
  uint32_t lcplen = offset, lpref, rpref;
  int ret = vsa_probe<IDX, KEYED>(ix, vleft, query, querylen, qk, lcplen);

  maxlcp = lcplen;
  witness = vleft;
  if (ret <= 0)
  {
    return;
  }
  lpref = lcplen;
  lcplen = offset;
  ret = vsa_probe<IDX, KEYED>(ix, vright, query, querylen, qk, lcplen);
  rpref = lcplen;
  if (lpref < rpref)
  {
    maxlcp = rpref;
    witness = vright;
    lcplen = lpref;
  } else
  {
    maxlcp = lpref;
    witness = vleft;
  }
  if (ret >= 0 || maxlcp >= querylen)
  {
    return;
  }
  uint64_t left = vleft, right = vright;
  while (right > left + 1)
  {
    const uint64_t mid = (left + right) >> 1;
    ret = vsa_probe<IDX, KEYED>(ix, mid, query, querylen, qk, lcplen);
    if (maxlcp < lcplen)
    {
      maxlcp = lcplen;
      witness = mid;
    }
    if (ret < 0)
    {
      rpref = lcplen;
      if (lpref < rpref)
      {
        lcplen = lpref;
      }
      right = mid;
    } else if (ret > 0)
    {
      lpref = lcplen;
      if (rpref < lpref)
      {
        lcplen = rpref;
      }
      left = mid;
    } else
    {
      break;
    }
  }
}

// getexception, kurtz-basic/accvirt.c:69-150: value of the lcp entry i whose
// byte is 255.  The table is sorted by index; the reference's cache and its
// habit of walking to neighbouring exceptions only save time.
template <typename IDX>
__device__ __forceinline__ uint64_t vsa_largelcp(const DevIndex<IDX> &ix,
                                                 uint64_t i)
{
  uint64_t lo = 0, hi = ix.nllv;

  while (lo < hi)
  {
    const uint64_t mid = lo + ((hi - lo) >> 1);
    const uint64_t k = (uint64_t) ix.llv[2 * mid];
    if (i < k)
    {
      hi = mid;
    } else if (i > k)
    {
      lo = mid + 1;
    } else
    {
      return (uint64_t) ix.llv[2 * mid + 1];
    }
  }
  return 255; // not reachable on a consistent index
}

// macro EVALLCP, include/virtualdef.h:292-299
template <typename IDX>
__device__ __forceinline__ uint64_t vsa_evallcp(const DevIndex<IDX> &ix,
                                                uint64_t i)
{
  const uint64_t v = ix.lcp[i];
  return (v < 255) ? v : vsa_largelcp(ix, i);
}

// include/qgram2code.c:7-37
template <typename IDX>
__device__ __forceinline__ bool vsa_qgram2code(const DevIndex<IDX> &ix,
                                               const uint8_t *qgram,
                                               uint64_t &code)
{
  uint64_t c = 0;
  bool ok = true;

  if (ix.numofchars == 4)
  {
    for (uint32_t i = 0; i < ix.pl; i++)
    {
      const uint8_t a = qgram[i];
      ok = ok && !VSA_ISSPECIAL(a);
      c = (c << 2) | (a & 3);
    }
  } else
  {
    for (uint32_t i = 0; i < ix.pl; i++)
    {
      const uint8_t a = qgram[i];
      ok = ok && !VSA_ISSPECIAL(a);
      c = c * ix.numofchars + a;
    }
  }
  code = c;
  return ok;
}

// bucket of the first prefixlength symbols: Vmengine/exactcompl.c:186-194 /
// kurtz/matchsub.c:199-205.  false: q-gram has a special symbol or the
// bucket holds no suffix.
template <typename IDX>
__device__ __forceinline__ bool vsa_bucket(const DevIndex<IDX> &ix,
                                           const uint8_t *qgram,
                                           uint64_t &vleft, uint64_t &vright)
{
  uint64_t code;

  if (!vsa_qgram2code(ix, qgram, code))
  {
    return false;
  }
  // (left, mid) sit next to each other: one 8/16-byte load
  vleft = (uint64_t) ix.bck[2 * code];
  vright = (uint64_t) ix.bck[2 * code + 1];
  if (vright <= vleft)
  {
    return false;
  }
  vright--;
  return true;
}

// 64-lane helpers ----------------------------------------------------------

__device__ __forceinline__ uint32_t vsa_lane()
{
  return __builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
}

__device__ __forceinline__ uint64_t vsa_shfl64(uint64_t v, int src)
{
  const uint32_t lo = __shfl((uint32_t) v, src, 64),
                 hi = __shfl((uint32_t) (v >> 32), src, 64);
  return ((uint64_t) hi << 32) | lo;
}

// Every lane of the wavefront brings a count c; one atomic per wavefront
// reserves the sum in *cursor and each lane gets the start of its own range.
// Must be reached by all 64 lanes.
__device__ __forceinline__ uint64_t
vsa_wave_reserve(unsigned long long *cursor, uint32_t c)
{
  const uint32_t lane = vsa_lane();
  uint32_t incl = c;

#pragma unroll
  for (int d = 1; d < 64; d <<= 1)
  {
    const uint32_t v = __shfl_up(incl, d, 64);
    if (lane >= (uint32_t) d)
    {
      incl += v;
    }
  }
  const uint32_t total = __shfl(incl, 63, 64);
  uint64_t base = 0;
  if (lane == 63 && total > 0)
  {
    base = atomicAdd(cursor, (unsigned long long) total);
  }
  base = vsa_shfl64(base, 63);
  return base + incl - c;
}
