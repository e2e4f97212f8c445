// Device-side primitives of the enhanced-suffix-array search, CDNA4 (gfx950).
//
// One work-item owns one search (one query for -complete, one query suffix
// for -l / -mum): the path is a chain of dependent random reads, so
// throughput comes from the number of independent chains in flight (64 per
// wavefront, 32 wavefronts per CU) and from how FEW dependent steps each
// wavefront needs -- a wavefront advances at the pace of its slowest lane,
// and every step costs one loaded memory latency (~1000 cycles measured).
//
// Two ways to locate a query suffix Q in the index:
//
//   reference walk   bucket of the first prefixlength symbols (bck) + the
//                    lcp-aware binary search of kurtz/findmaxpref.gen on
//                    suf/tis.  Probe for probe the reference's algorithm.
//   deep locate      an internal, deeper bucket table (bck2, D symbols, about
//                    one to three suffixes per bucket) + the keyed array esa8
//                    {suf:32, lcp byte:8, key:20 = the 10 symbols behind the
//                    first D, the symbol in FRONT of the suffix:2, flag:1,
//                    left-is-special:1}: a short, wavefront-uniform search on
//                    keys, then ONE text comparison for the lanes whose key
//                    ties.  ~10 dependent steps instead of ~140, ~3 random
//                    64-byte sectors instead of ~15.
//
// Both deliver the same two numbers the reference computes -- the maximal
// matched length and a suffix attaining it -- and everything reported is a
// function of those.  The one place where the
// reference's probe ORDER shows (which of several equally good suffixes is
// the "witness" the MEM enumeration starts from) is reproduced by replaying
// the probe sequence arithmetically (vsa_reference_witness).
//
// Semantics follow the reference (paths relative to /root/reference/src).
#pragma once

#include "vsa_internal.hpp"
#include <type_traits>

#define VSA_ISSPECIAL(c) ((c) >= (uint8_t) VSA_WILDCARD) // chardef.h:37

__device__ __forceinline__ uint64_t vsa_load8(const uint8_t *p)
{
  uint64_t v;
  __builtin_memcpy(&v, p, 8); // one global_load_dwordx2, any alignment
  return v;
}

// sixteen bytes, any alignment: one global_load_dwordx4.  A fully divergent
// wavefront load costs the address unit about one cycle per lane and
// instruction, so the search uses few, wide loads.
struct vsa_u128
{
  uint64_t lo, hi;
};

__device__ __forceinline__ vsa_u128 vsa_load16(const void *p)
{
  vsa_u128 v;
  __builtin_memcpy(&v, p, 16);
  return v;
}

// ---- reads that live in HBM at two bits per symbol (DevQueries::rows) ------
// The fast paths take such a read as a PackedQuery (registers); the paths that
// compare byte for byte -- the reference walk, a comparison that met a special
// symbol of the text -- see it through a QSrc, a "pointer to symbols" that
// expands eight of them from the row when asked.  A read with a special symbol
// of its own has its bytes in the batch's side buffer (`bytes` != nullptr) and
// is read from there.  Positions behind the read hold 0xFF, like the padding
// behind a byte batch.
struct QSrc
{
  const uint64_t *row;  // the read's words, symbol 0 in the top bits of row[0]
  const uint8_t *bytes; // or its symbols as bytes (reads with a wildcard)
  int32_t pos;          // the symbol this "pointer" stands at
  int32_t m;            // symbols of the read

  __device__ __forceinline__ QSrc operator+(uint32_t k) const
  {
    QSrc q = *this;
    q.pos += (int32_t) k;
    return q;
  }
};

// symbols [at, at + 32) of a row of `words` words, zeros behind them
__device__ __forceinline__ uint64_t vsa_row_window(const uint64_t *row,
                                                   uint32_t at)
{
  // (rows lie back to back and the batch has a word of slack behind it: the
  // word behind the one that holds `at` can always be read)
  const uint32_t i = at >> 5, sh = 2u * (at & 31u);
  const uint64_t a = row[i], b = row[i + 1];
  return sh != 0 ? (a << sh) | (b >> (64 - sh)) : a;
}

// eight symbols from q.pos on, one byte each, the first in the lowest byte
__device__ __forceinline__ uint64_t vsa_load8(const QSrc &q)
{
  const int32_t rem = q.m - q.pos;
  if (rem <= 0)
  {
    return ~0ull;
  }
  uint64_t r;
  if (q.bytes != nullptr)
  {
    __builtin_memcpy(&r, q.bytes + q.pos, 8);
  } else
  {
    const uint32_t v = (uint32_t) (vsa_row_window(q.row, (uint32_t) q.pos) >> 48);
    r = 0;
#pragma unroll
    for (int k = 0; k < 8; k++)
    {
      r |= (uint64_t) ((v >> (14 - 2 * k)) & 3u) << (8 * k);
    }
  }
  if (rem < 8)
  {
    r |= ~0ull << (8 * rem);
  }
  return r;
}

__device__ __forceinline__ vsa_u128 vsa_load16(const QSrc &q)
{
  vsa_u128 v;
  v.lo = vsa_load8(q);
  v.hi = vsa_load8(q + 8u);
  return v;
}

__device__ __forceinline__ uint8_t vsa_qsym(const QSrc &q, uint32_t i)
{
  return (uint8_t) (vsa_load8(q + i) & 0xFFu);
}
__device__ __forceinline__ uint8_t vsa_qsym(const uint8_t *q, uint32_t i)
{
  return q[i];
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
template <typename IDX, typename QP = const uint8_t *>
__device__ __forceinline__ int vsa_compare(const DevIndex<IDX> &ix,
                                           uint64_t sufstart, QP query,
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

// The same comparison, 32 symbols per round trip: used where all lanes of a
// wavefront extend a long match at the same time, so that the number of
// dependent memory steps, not the number of loads, sets the pace.
// AHEAD: 32-symbol pieces fetched per round trip.  1 suits searches that
// usually stop within the first piece; the first pass of a MUM batch, where
// most reads match over their whole length, asks for 3 so that a 100 bp read
// is compared in one round trip instead of three.
template <typename IDX, int AHEAD = 1, typename QP = const uint8_t *>
__device__ __forceinline__ int vsa_compare32(const DevIndex<IDX> &ix,
                                             uint64_t sufstart, QP query,
                                             uint32_t querylen,
                                             uint32_t &lcplen)
{
  uint32_t l = lcplen;

  for (;;)
  {
    if (l >= querylen)
    {
      lcplen = querylen;
      return 0;
    }
    uint64_t a[AHEAD][4], b[AHEAD][4];
#pragma unroll
    for (int r = 0; r < AHEAD; r++)
    {
      // pieces behind the end of the query are not fetched; pieces behind
      // the end of the text are never looked at (the padding behind the text
      // makes an earlier piece mismatch): keep their loads inside it
      const uint32_t off = l + 32u * (uint32_t) r;
      if (r == 0 || off < querylen)
      {
        uint64_t tpos = sufstart + off;
        if (r > 0 && tpos > ix.n + 32)
        {
          tpos = ix.n + 32;
        }
        const vsa_u128 qa = vsa_load16(query + off),
                       qb = vsa_load16(query + off + 16),
                       ta = vsa_load16(ix.tis + tpos),
                       tb = vsa_load16(ix.tis + tpos + 16);
        a[r][0] = qa.lo;
        a[r][1] = qa.hi;
        a[r][2] = qb.lo;
        a[r][3] = qb.hi;
        b[r][0] = ta.lo;
        b[r][1] = ta.hi;
        b[r][2] = tb.lo;
        b[r][3] = tb.hi;
      } else
      {
#pragma unroll
        for (int k = 0; k < 4; k++)
        {
          a[r][k] = b[r][k] = 0;
        }
      }
    }
#pragma unroll
    for (int r = 0; r < AHEAD; r++)
    {
      uint64_t m[4];
#pragma unroll
      for (int k = 0; k < 4; k++)
      {
        m[k] = (a[r][k] ^ b[r][k]) | vsa_specialmask(a[r][k]) |
               vsa_specialmask(b[r][k]);
      }
      if ((m[0] | m[1] | m[2] | m[3]) == 0)
      {
        l += 32;
        if (l >= querylen)
        {
          lcplen = querylen;
          return 0;
        }
        continue;
      }
      int k = m[0] ? 0 : (m[1] ? 1 : (m[2] ? 2 : 3));
      const uint64_t mm = m[0] ? m[0] : (m[1] ? m[1] : (m[2] ? m[2] : m[3]));
      const uint64_t aa = m[0] ? a[r][0]
                               : (m[1] ? a[r][1] : (m[2] ? a[r][2] : a[r][3]));
      const uint64_t bb = m[0] ? b[r][0]
                               : (m[1] ? b[r][1] : (m[2] ? b[r][2] : b[r][3]));
      const uint32_t j = (uint32_t) __builtin_ctzll(mm) >> 3;
      l += 8 * k + j;
      if (l >= querylen)
      {
        lcplen = querylen;
        return 0;
      }
      lcplen = l;
      const int qa = (int) ((aa >> (8 * j)) & 0xFF),
                tb = (int) ((bb >> (8 * j)) & 0xFF);
      return (qa == tb) ? -1 : qa - tb;
    }
  }
}

// four symbols (the bytes of h, first symbol in the lowest) -> 8 bits in the
// TOP byte of the result, first symbol most significant; the bytes below are
// scrap.  One multiplication lines the four 2-bit fields up: byte k moves by
// 30 - 10k bits, and none of the other partial products reaches bit 24.
__device__ __forceinline__ uint32_t vsa_pack4top(uint32_t h)
{
  return (h & 0x03030303u) * 0x40100401u;
}

// 16 symbols (bytes 0..3 each, two words) -> 32 bits, first symbol in the top two
__device__ __forceinline__ uint32_t vsa_pack16(uint64_t a, uint64_t b)
{
  return (vsa_pack4top((uint32_t) a) & 0xFF000000u) |
         ((vsa_pack4top((uint32_t) (a >> 32)) >> 8) & 0x00FF0000u) |
         ((vsa_pack4top((uint32_t) b) >> 16) & 0xFF00u) |
         (vsa_pack4top((uint32_t) (b >> 32)) >> 24);
}

// Continues the comparison of query[lcplen..querylen) with the suffix at
// sufstart on the 2-bit text (DevIndex::tis2): 60 symbols per 16-byte load of
// the text instead of 16.  Returns false if the answer needs the bytes -- a
// block of the text it touched holds a special symbol or the end of the text
// (the caller repeats the comparison with vsa_compare32) -- and otherwise
// leaves in lcplen what COMPARE / CHECKRETURN (kurtz/maxpref.c:30-65) would:
// the position of the first mismatch, of the first special symbol of the
// query, or querylen.  CHUNKS: pieces of 60 symbols fetched per round trip.
template <int CHUNKS, typename IDX, typename QP = const uint8_t *>
__device__ __forceinline__ bool
vsa_extend_packed(const DevIndex<IDX> &ix, uint64_t sufstart, QP query,
                  uint32_t querylen, uint32_t &lcplen)
{
  const uint32_t from = lcplen;
  uint32_t l = lcplen, qeff = querylen;
  bool open = true; // no mismatch seen yet

  // the packed text knows no end: a comparison that reaches position n stops
  // there, and the block that holds n sends it to the bytes
  if ((uint64_t) qeff > ix.n - sufstart)
  {
    qeff = (uint32_t) (ix.n - sufstart);
  }

  while (open && l < qeff)
  {
    uint64_t A[CHUNKS][2], Q[CHUNKS][2];
    uint32_t bad[CHUNKS];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
    {
      const uint32_t lc = l + 60u * (uint32_t) c;
      bad[c] = 64;
      A[c][0] = A[c][1] = Q[c][0] = Q[c][1] = 0;
      if (c == 0 || lc < qeff)
      {
        const uint64_t pos = sufstart + lc;
        const uint32_t sh = 2u * (uint32_t) (pos & 3u);
        const vsa_u128 t = vsa_load16(ix.tis2 + (pos >> 2));
        const uint64_t t0 = __builtin_bswap64(t.lo),
                       t1 = __builtin_bswap64(t.hi);
        A[c][0] = sh != 0 ? (t0 << sh) | (t1 >> (64 - sh)) : t0;
        A[c][1] = t1 << sh;
        // 16-byte pieces of the query behind its end are not fetched
        const vsa_u128 zero = {0, 0};
        const vsa_u128 q0 = vsa_load16(query + lc),
                       q1 = lc + 16 < qeff ? vsa_load16(query + lc + 16) : zero,
                       q2 = lc + 32 < qeff ? vsa_load16(query + lc + 32) : zero,
                       q3 = lc + 48 < qeff ? vsa_load16(query + lc + 48) : zero;
        Q[c][0] = ((uint64_t) vsa_pack16(q0.lo, q0.hi) << 32) |
                  vsa_pack16(q1.lo, q1.hi);
        Q[c][1] = ((uint64_t) vsa_pack16(q2.lo, q2.hi) << 32) |
                  vsa_pack16(q3.lo, q3.hi);
        // first byte of the query piece that is no DNA symbol
        const uint64_t notdna = 0xFCFCFCFCFCFCFCFCull;
        const uint64_t w[8] = {q0.lo, q0.hi, q1.lo, q1.hi,
                               q2.lo, q2.hi, q3.lo, q3.hi};
#pragma unroll
        for (int k = 7; k >= 0; k--)
        {
          const uint64_t s = w[k] & notdna;
          if (s != 0)
          {
            bad[c] = 8u * (uint32_t) k + ((uint32_t) __builtin_ctzll(s) >> 3);
          }
        }
      }
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
    {
      if (open && l < qeff)
      {
        if (bad[c] < 60 && l + bad[c] < qeff)
        {
          qeff = l + bad[c]; // the comparison ends at a special query symbol
        }
        const uint64_t x0 = A[c][0] ^ Q[c][0],
                       x1 = (A[c][1] ^ Q[c][1]) & ~0xFFull; // 28 symbols
        uint32_t same = 60;
        if (x0 != 0)
        {
          same = (uint32_t) __builtin_clzll(x0) >> 1;
        } else if (x1 != 0)
        {
          same = 32 + ((uint32_t) __builtin_clzll(x1) >> 1);
        }
        l += same;
        open = same == 60;
      }
    }
  }
  if (l > qeff)
  {
    l = qeff;
  }
  // did the comparison touch a block with a special symbol (packed as 0)?
  if (sufstart + l + 64 >= ix.firstspecial)
  {
    uint64_t b = (sufstart + from) >> 6;
    const uint64_t blast = (sufstart + l) >> 6;
    while (b <= blast)
    {
      uint32_t bits;
      __builtin_memcpy(&bits, ix.spec64 + (b >> 3), 4);
      bits >>= (uint32_t) (b & 7u);
      const uint64_t have = 25, want = blast - b + 1;
      const uint32_t mask = want >= have ? (1u << have) - 1u
                                         : (1u << want) - 1u;
      if ((bits & mask) != 0)
      {
        return false;
      }
      b += have;
    }
  }
  lcplen = l;
  return true;
}

// A whole query of up to 128 symbols at two bits per symbol in registers:
// symbol 0 in the top two bits of w[0].  valid = the number of leading DNA
// symbols (a wildcard ends it; <= the query's length).  Kernels whose
// work-item is a query fetch the queries of a workgroup with coalesced loads
// (vsa_pq_stage) instead of seven divergent 16-byte loads per lane, and every
// later look at the query -- deep prefix, key, the comparison behind a key
// tie -- is register arithmetic.
struct PackedQuery
{
  uint64_t w[4];
  uint32_t valid;
};

// symbols [at, at + 32) of pq, zeros behind symbol 127
__device__ __forceinline__ uint64_t vsa_pq_window(const PackedQuery &pq,
                                                  uint32_t at)
{
  const uint32_t i = at >> 5, sh = 2u * (at & 31u);
  const uint64_t a = i == 0 ? pq.w[0]
                            : (i == 1 ? pq.w[1]
                                      : (i == 2 ? pq.w[2]
                                                : (i == 3 ? pq.w[3] : 0))),
                 b = i == 0 ? pq.w[1]
                            : (i == 1 ? pq.w[2] : (i == 2 ? pq.w[3] : 0));
  return sh != 0 ? (a << sh) | (b >> (64 - sh)) : a;
}

// The queries q0 .. q0 + BLK - 1 of a dense batch (every query m symbols, m a
// multiple of 4, m <= 128; BLK * m a multiple of 16) through LDS: the
// workgroup copies its BLK * m bytes with coalesced 16-byte loads, then every
// lane packs its own query from LDS (row stride m/4 words: odd multiples of
// four symbols are conflict free, m = 100 is).  lds: BLK * m bytes.  Must be
// reached by all threads of the workgroup.
template <int BLK>
__device__ __forceinline__ void vsa_pq_stage(const DevQueries &qs, uint64_t q0,
                                             uint32_t m, uint32_t *lds,
                                             PackedQuery &pq)
{
  const uint64_t nhere = q0 < qs.nq ? (qs.nq - q0 < (uint64_t) BLK
                                           ? qs.nq - q0
                                           : (uint64_t) BLK)
                                    : 0;
  const uint32_t bytes = (uint32_t) nhere * m;
  const uint8_t *src = qs.symbols + q0 * m;
  for (uint32_t i = threadIdx.x; 16 * i < bytes; i += BLK)
  {
    // (the last piece may read up to 15 bytes of the next workgroup's
    // queries or of the padding behind the batch)
    const vsa_u128 v = vsa_load16(src + 16 * (uint64_t) i);
    uint4 *dst = reinterpret_cast<uint4 *>(lds) + i;
    *dst = make_uint4((uint32_t) v.lo, (uint32_t) (v.lo >> 32),
                      (uint32_t) v.hi, (uint32_t) (v.hi >> 32));
  }
  __syncthreads();
  pq.w[0] = pq.w[1] = pq.w[2] = pq.w[3] = 0;
  pq.valid = 0;
  if (threadIdx.x < nhere)
  {
    const uint32_t *row = lds + threadIdx.x * (m >> 2);
    uint32_t firstbad = m;
#pragma unroll
    for (uint32_t j = 0; j < 32; j++)
    {
      if (4 * j < m)
      {
        const uint32_t x = row[j];
        const uint32_t bad = x & 0xFCFCFCFCu;
        if (bad != 0 && firstbad == m)
        {
          firstbad = 4 * j + ((uint32_t) __builtin_ctz(bad) >> 3);
        }
        pq.w[j >> 3] |= (uint64_t) (vsa_pack4top(x) >> 24)
                        << (56 - 8 * (j & 7));
      }
    }
    pq.valid = firstbad;
  }
}

// Read q of a packed batch (DevQueries::rows, reads of m <= 124 symbols: at
// most four words): its symbols into pq -- two 16-byte loads, neighbouring
// lanes on neighbouring rows --, and the QSrc the byte paths look through.
// A read with a special symbol (flag byte 1) has its bytes in the side buffer
// at the index its word 0 names; it is packed from there, up to the symbol.
__device__ __forceinline__ void vsa_pq_from_row(const DevQueries &qs,
                                                uint64_t q, uint32_t m,
                                                PackedQuery &pq, QSrc &src)
{
  const uint32_t W = qs.roww;
  const uint64_t *row = qs.rows + q * W;
  // (32 bytes whatever W is: the batch has that much slack behind it)
  const vsa_u128 a = vsa_load16(row), b = vsa_load16(row + 2);
  const uint64_t last = W == 1 ? a.lo : (W == 2 ? a.hi : (W == 3 ? b.lo : b.hi));
  pq.w[0] = a.lo;
  pq.w[1] = W > 1 ? a.hi : 0;
  pq.w[2] = W > 2 ? b.lo : 0;
  pq.w[3] = W > 3 ? b.hi : 0;
  // the flag byte is no symbol
  if (W == 1)
  {
    pq.w[0] &= ~0xFFull;
  } else if (W == 2)
  {
    pq.w[1] &= ~0xFFull;
  } else if (W == 3)
  {
    pq.w[2] &= ~0xFFull;
  } else
  {
    pq.w[3] &= ~0xFFull;
  }
  pq.valid = m;
  src.row = row;
  src.bytes = nullptr;
  src.pos = 0;
  src.m = (int32_t) m;
  if ((last & 0xFFu) != 0 && qs.nside > 0)
  {
    // (the index is clamped to the list: a row that lies reads another read,
    // never another buffer)
    const uint64_t k = (a.lo >> 8) < qs.nside ? (a.lo >> 8) : qs.nside - 1;
    const uint8_t *sym = qs.side + k * (uint64_t) m;
    uint32_t firstbad = m;
    src.bytes = sym;
#pragma unroll
    for (int wi = 0; wi < 4; wi++)
    {
      uint64_t acc = 0;
      for (uint32_t k = 0; k < 32; k++)
      {
        const uint32_t j = 32u * (uint32_t) wi + k;
        if (j < m)
        {
          const uint8_t c = sym[j];
          if (c > 3 && firstbad == m)
          {
            firstbad = j;
          }
          acc |= (uint64_t) (c & 3u) << (62 - 2 * k);
        }
      }
      pq.w[wi] = acc;
    }
    pq.valid = firstbad;
  }
}

// A read of a packed batch for the kernels whose work-item looks at a window
// of it (work plan, search kernel): nothing but the row's address travels in
// registers, every window is one 16-byte load of the row (a cache hit for all
// but the first of a read's work-items).  Only for reads WITHOUT a special
// symbol: `valid` = the read's length.  (A PackedQuery's `valid` is the
// position of its FIRST special symbol -- right for a search that starts at
// symbol 0, wrong for the suffixes behind the symbol; reads with one take the
// byte path through their QSrc instead.)
struct RowQuery
{
  const uint64_t *row;
  uint32_t valid;
};

__device__ __forceinline__ uint64_t vsa_pq_window(const RowQuery &rq,
                                                  uint32_t at)
{
  const uint32_t i = at >> 5, sh = 2u * (at & 31u);
  const vsa_u128 w = vsa_load16(rq.row + i);
  return sh != 0 ? (w.lo << sh) | (w.hi >> (64 - sh)) : w.lo;
}

// the row of read q, its flag (a read with a special symbol), and the QSrc
// of the read: its bytes in the side list if flagged, the row otherwise
__device__ __forceinline__ bool vsa_row_of(const DevQueries &qs, uint64_t q,
                                           uint32_t m, RowQuery &rq,
                                           QSrc &src)
{
  const uint32_t W = qs.roww;
  rq.row = qs.rows + q * W;
  rq.valid = m;
  src.row = rq.row;
  src.bytes = nullptr;
  src.pos = 0;
  src.m = (int32_t) m;
  const bool flagged = (rq.row[W - 1] & 0xFFu) != 0 && qs.nside > 0;
  if (flagged)
  {
    const uint64_t k0 = rq.row[0] >> 8,
                   k = k0 < qs.nside ? k0 : qs.nside - 1;
    src.bytes = qs.side + k * (uint64_t) m;
  }
  return flagged;
}

// symbol i of a packed query (i < 128)
__device__ __forceinline__ uint8_t vsa_pq_symbol(const PackedQuery &pq,
                                                 uint32_t i)
{
  return (uint8_t) (vsa_pq_window(pq, i) >> 62);
}

// vsa_extend_packed for a query that is in registers already
template <int CHUNKS, typename IDX, typename QT>
__device__ __forceinline__ bool
vsa_extend_packed_pq(const DevIndex<IDX> &ix, uint64_t sufstart,
                     const QT &pq, uint32_t pqoff,
                     uint32_t querylen, uint32_t &lcplen)
{
  // the query suffix at offset pqoff of the packed query, querylen symbols
  const uint32_t from = lcplen,
                 left = pq.valid > pqoff ? pq.valid - pqoff : 0;
  uint32_t l = lcplen, qeff = querylen < left ? querylen : left;
  bool open = true;

  if ((uint64_t) qeff > ix.n - sufstart)
  {
    qeff = (uint32_t) (ix.n - sufstart);
  }
  while (open && l < qeff)
  {
    uint64_t A[CHUNKS][2];
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
    {
      const uint32_t lc = l + 60u * (uint32_t) c;
      A[c][0] = A[c][1] = 0;
      if (c == 0 || lc < qeff)
      {
        const uint64_t pos = sufstart + lc;
        const uint32_t sh = 2u * (uint32_t) (pos & 3u);
        const vsa_u128 t = vsa_load16(ix.tis2 + (pos >> 2));
        const uint64_t t0 = __builtin_bswap64(t.lo),
                       t1 = __builtin_bswap64(t.hi);
        A[c][0] = sh != 0 ? (t0 << sh) | (t1 >> (64 - sh)) : t0;
        A[c][1] = t1 << sh;
      }
    }
#pragma unroll
    for (int c = 0; c < CHUNKS; c++)
    {
      if (open && l < qeff)
      {
        const uint64_t x0 = A[c][0] ^ vsa_pq_window(pq, pqoff + l),
                       x1 = (A[c][1] ^ vsa_pq_window(pq, pqoff + l + 32)) &
                            ~0xFFull;
        uint32_t same = 60;
        if (x0 != 0)
        {
          same = (uint32_t) __builtin_clzll(x0) >> 1;
        } else if (x1 != 0)
        {
          same = 32 + ((uint32_t) __builtin_clzll(x1) >> 1);
        }
        l += same;
        open = same == 60;
      }
    }
  }
  if (l > qeff)
  {
    l = qeff;
  }
  if (sufstart + l + 64 >= ix.firstspecial)
  {
    uint64_t b = (sufstart + from) >> 6;
    const uint64_t blast = (sufstart + l) >> 6;
    while (b <= blast)
    {
      uint32_t bits;
      __builtin_memcpy(&bits, ix.spec64 + (b >> 3), 4);
      bits >>= (uint32_t) (b & 7u);
      const uint64_t have = 25, want = blast - b + 1;
      const uint32_t mask = want >= have ? (1u << have) - 1u
                                         : (1u << want) - 1u;
      if ((bits & mask) != 0)
      {
        return false;
      }
      b += have;
    }
  }
  lcplen = l;
  return true;
}

// table accessors: esa8 carries suf (32-bit tables; of a wider suf only the
// low half, which nobody reads) and the lcp byte next to each other
template <typename IDX, bool KEYED>
__device__ __forceinline__ uint64_t vsa_sufstart(const DevIndex<IDX> &ix,
                                                 uint64_t i)
{
  return (KEYED && sizeof(IDX) == 4) ? (ix.esa8[i] & 0xFFFFFFFFull)
                                     : (uint64_t) ix.suf[i];
}

// start of the suffix with index w whose esa8 entry is at hand
template <typename IDX>
__device__ __forceinline__ uint64_t vsa_entrystart(const DevIndex<IDX> &ix,
                                                   uint64_t entry, uint64_t w)
{
  if constexpr (sizeof(IDX) == 4)
  {
    return entry & 0xFFFFFFFFull;
  } else
  {
    return (uint64_t) ix.suf[w];
  }
}

// word 0 of a deep bucket's slot (and the entries of bck2): 32-bit tables
// left | mid << 32; wide tables left (40 bits) | number of suffixes << 40 --
// an index whose deep buckets do not fit 24 bits gets no deep tables
#define VSA_WIDE_LEFTBITS 40
template <typename IDX>
__device__ __forceinline__ void vsa_slotbounds(uint64_t b, uint64_t &left,
                                               uint32_t &cnt)
{
  if constexpr (sizeof(IDX) == 4)
  {
    const uint32_t dl = (uint32_t) b, dm = (uint32_t) (b >> 32);
    left = dl;
    cnt = (dm > dl) ? dm - dl : 0;
  } else
  {
    left = b & ((1ull << VSA_WIDE_LEFTBITS) - 1);
    cnt = (uint32_t) (b >> VSA_WIDE_LEFTBITS);
  }
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
template <typename IDX, typename QP = const uint8_t *>
__device__ __forceinline__ void
vsa_findmaxprefixlen(const DevIndex<IDX> &ix, uint64_t vleft, uint64_t vright,
                     uint32_t offset, QP query, uint32_t querylen,
                     uint32_t &maxlcp, uint64_t &witness)
{
  uint32_t lcplen = offset, lpref, rpref;
  int ret = vsa_compare(ix, (uint64_t) ix.suf[vleft], query, querylen, lcplen);

  maxlcp = lcplen;
  witness = vleft;
  if (ret <= 0)
  {
    return;
  }
  lpref = lcplen;
  lcplen = offset;
  ret = vsa_compare(ix, (uint64_t) ix.suf[vright], query, querylen, lcplen);
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
    ret = vsa_compare(ix, (uint64_t) ix.suf[mid], query, querylen, lcplen);
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

// The witness findmaxpref.gen ends with, given the bucket [vleft, vright] it
// searches and the interval [l, r] of the suffixes that attain the maximal
// matched length: the first probe of its fixed sequence (vleft, vright, then
// bisection towards the query's position) that falls into [l, r].  Every
// suffix left of l is smaller than the query and every suffix right of r is
// larger, which is all the bisection looks at; only a probe inside [l, r]
// raises the running maximum to its final value (findmaxpref.gen:57-63).
__device__ __forceinline__ uint64_t
vsa_reference_witness(uint64_t vleft, uint64_t vright, uint64_t l, uint64_t r)
{
  if (vleft >= l) // vleft <= r always: the bucket contains [l, r]
  {
    return vleft;
  }
  if (vright <= r)
  {
    return vright;
  }
  uint64_t left = vleft, right = vright;
  while (right > left + 1)
  {
    const uint64_t mid = (left + right) >> 1;
    if (mid < l)
    {
      left = mid;
    } else if (mid > r)
    {
      right = mid;
    } else
    {
      return mid;
    }
  }
  return l; // not reached: [l, r] lies strictly between left and right
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
template <typename IDX, typename QP = const uint8_t *>
__device__ __forceinline__ bool vsa_qgram2code(const DevIndex<IDX> &ix,
                                               QP qgram, uint64_t &code)
{
  uint64_t c = 0;
  bool ok = true;

  if (ix.numofchars == 4)
  {
    for (uint32_t i = 0; i < ix.pl; i++)
    {
      const uint8_t a = vsa_qsym(qgram, i);
      ok = ok && !VSA_ISSPECIAL(a);
      c = (c << 2) | (a & 3);
    }
  } else
  {
    for (uint32_t i = 0; i < ix.pl; i++)
    {
      const uint8_t a = vsa_qsym(qgram, i);
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
template <typename IDX, typename QP = const uint8_t *>
__device__ __forceinline__ bool vsa_bucket(const DevIndex<IDX> &ix, QP qgram,
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

// the reference's way to a (maxlcp, witness) pair
template <typename IDX, typename QP = const uint8_t *>
__device__ __forceinline__ bool
vsa_locate_reference(const DevIndex<IDX> &ix, QP query, uint32_t querylen,
                     uint32_t &maxlcp, uint64_t &witness)
{
  uint64_t vleft, vright;

  if (!vsa_bucket(ix, query, vleft, vright))
  {
    return false;
  }
  vsa_findmaxprefixlen(ix, vleft, vright, ix.pl, query, querylen, maxlcp,
                       witness);
  return true;
}

// ---- deep locate (DNA) ------------------------------------------------------

// eight symbols (one byte each, 0..3) -> 16 bits, first symbol most
// significant, so that integers compare like the strings they pack
__device__ __forceinline__ uint64_t vsa_pack8(uint64_t w)
{
  uint64_t x = __builtin_bswap64(w) & 0x0303030303030303ull;
  x = (x | (x >> 6)) & 0x000F000F000F000Full;
  x = (x | (x >> 12)) & 0x000000FF000000FFull;
  x = (x | (x >> 24)) & 0xFFFFull;
  return x;
}

// number of leading key symbols (2 bits each, `nsyms` of them in the low
// bits of a and b) that agree
__device__ __forceinline__ uint32_t vsa_keylcp(uint32_t a, uint32_t b,
                                               uint32_t nsyms)
{
  const uint32_t x = (a ^ b) & ((1u << (2 * nsyms)) - 1u);
  if (x == 0)
  {
    return nsyms;
  }
  // highest differing bit h -> symbol (2*nsyms-1-h)/2 counted from the front
  const uint32_t h = 31u - (uint32_t) __builtin_clz(x);
  return (2 * nsyms - 1 - h) >> 1;
}

enum
{
  VSA_LOC_NONE = 0, // nothing in the index shares D symbols with the query
  VSA_LOC_FOUND = 1,
  VSA_LOC_SLOW = 2  // take the reference walk (special symbols, ties, ...)
};

// Deep locate.  Must be called by all lanes of the wavefront (inactive lanes
// pass active = false): the key search runs a wavefront-uniform number of
// rounds.  On VSA_LOC_FOUND: maxlcp = the maximal matched length over the
// whole index, w = a suffix-array index attaining it.
__device__ __forceinline__ uint64_t vsa_ld_entry(const uint64_t *p)
{
  return *p;
}

// what the deep locate knows about the located suffix w without further
// memory traffic: its esa8 entry (start, lcp byte), the lcp byte of w+1 and
// the text symbol in front of it
struct DeepHit
{
  uint64_t ew;      // esa8[w]
  uint32_t lcpnext; // lcp byte of entry w+1 (0 if w = n)
  uint8_t leftsym;  // tis[suf[w]-1] (separator if suf[w] = 0)
  bool notleftmax;  // see vsa_locate_deep, qleft
};

// What the first round trip of the deep locate brings: the bounds of the deep
// bucket of the query's first D symbols with the bucket's first entry, and
// the query's key symbols.
struct DeepFront
{
  uint64_t left;             // first suffix of the bucket
  uint64_t first;            // esa8[left] from the slot (0: empty bucket)
  uint32_t cnt, qkey, limit; // suffixes in the bucket; key symbols, how
                             // many of them the query has
  int state;                 // VSA_LOC_NONE / FOUND (= go on) / SLOW
};

template <bool PQ = false, typename IDX = uint32_t, typename QT = PackedQuery,
          typename QP = const uint8_t *>
__device__ __forceinline__ void
vsa_deep_front(const DevIndex<IDX> &ix, bool active, QP query,
               uint32_t querylen, DeepFront &f,
               const QT *pq = nullptr, uint32_t pqoff = 0)
{
  const uint32_t D = ix.D;
  f.state = VSA_LOC_NONE;
  f.left = 0;
  f.cnt = f.qkey = f.limit = 0;
  f.first = 0;
  if (active)
  {
    uint32_t valid = 32; // leading regular symbols inside the query
    uint64_t S = 0;      // its first 32 symbols, two bits each
    if (PQ)
    {
      const uint32_t left = pq->valid > pqoff ? pq->valid - pqoff : 0;
      valid = left < 32 ? left : 32;
      S = vsa_pq_window(*pq, pqoff);
    } else
    {
      // 32 query symbols as four 8-byte words (the buffer is padded)
      const vsa_u128 qlo = vsa_load16(query), qhi = vsa_load16(query + 16u);
      const uint64_t w0 = qlo.lo, w1 = qlo.hi, w2 = qhi.lo, w3 = qhi.hi;
      // this path exists for the DNA alphabet only (symbols 0..3): every
      // other byte -- wildcard, separator -- ends the regular prefix
      const uint64_t notdna = 0xFCFCFCFCFCFCFCFCull;
      const uint64_t s0 = w0 & notdna, s1 = w1 & notdna, s2 = w2 & notdna,
                     s3 = w3 & notdna;
      if ((s0 | s1 | s2 | s3) != 0)
      {
        if (s0 != 0)
        {
          valid = (uint32_t) __builtin_ctzll(s0) >> 3;
        } else if (s1 != 0)
        {
          valid = 8 + ((uint32_t) __builtin_ctzll(s1) >> 3);
        } else if (s2 != 0)
        {
          valid = 16 + ((uint32_t) __builtin_ctzll(s2) >> 3);
        } else
        {
          valid = 24 + ((uint32_t) __builtin_ctzll(s3) >> 3);
        }
      }
      S = ((uint64_t) vsa_pack16(w0, w1) << 32) | vsa_pack16(w2, w3);
    }
    if (valid > querylen)
    {
      valid = querylen;
    }
    if (valid < ix.pl)
    {
      f.state = VSA_LOC_NONE; // qgram2code fails or query shorter than pl
    } else if (valid < D)
    {
      f.state = VSA_LOC_SLOW;
    } else
    {
      const uint64_t code = S >> (64 - 2 * D);
      f.qkey = (uint32_t) (S >> (64 - 2 * D - 2 * VSA_KEYSYMS)) & VSA_KEYMASK;
      f.limit = valid - D;
      if (f.limit > VSA_KEYSYMS)
      {
        f.limit = VSA_KEYSYMS;
      }
      // (left, mid) of the deep bucket with the bucket's first entry: one
      // 16-byte load (deep tables always come with the fused slot table)
      const vsa_u128 sl = vsa_load16(ix.slot16 + 2 * code);
      f.first = sl.hi;
      vsa_slotbounds<IDX>(sl.lo, f.left, f.cnt);
      f.state = (f.cnt > 0) ? VSA_LOC_FOUND : VSA_LOC_NONE;
    }
  }
}

// The one suffix that ties with the query on all D + limit symbols the keys
// cover, at sstart: how far does the match go?  maxlcp: in, the symbols known
// to match; out, the matched length.  On the 2-bit text where there is one
// (two pieces of 60 symbols per round trip when the caller expects long
// matches); lanes whose comparison met a special symbol of the text repeat it
// on the bytes.  PQ: the query is *pq (symbols from pqoff on), `query` its
// bytes for the fallback.
template <int AHEAD, bool PQ, typename IDX, typename QT,
          typename QP = const uint8_t *>
__device__ __forceinline__ void
vsa_extend_tie(const DevIndex<IDX> &ix, uint64_t sstart, QP query,
               uint32_t querylen, uint32_t &maxlcp, const QT *pq,
               uint32_t pqoff)
{
  uint32_t lcplen = maxlcp;
  bool done = false;
  if (ix.tis2 != nullptr)
  {
    if constexpr (PQ)
    {
      done = vsa_extend_packed_pq<(AHEAD > 1 ? 2 : 1)>(ix, sstart, *pq, pqoff,
                                                       querylen, lcplen);
    } else
    {
      done = vsa_extend_packed<(AHEAD > 1 ? 2 : 1)>(ix, sstart, query,
                                                    querylen, lcplen);
    }
  }
  if (!done)
  {
    lcplen = maxlcp;
    (void) vsa_compare32<IDX, AHEAD>(ix, sstart, query, querylen, lcplen);
  }
  maxlcp = lcplen;
}

// the deep locate from a front that is at hand (all lanes, see below)
template <int AHEAD = 1, bool PQ = false, typename IDX = uint32_t,
          typename QT = PackedQuery, typename QP = const uint8_t *>
__device__ __forceinline__ int
vsa_locate_deep_from(const DevIndex<IDX> &ix, const DeepFront &f, QP query,
                     uint32_t querylen, uint32_t &maxlcp,
                     uint64_t &w, DeepHit &hit, uint32_t needleft = 0xFFFFFFFFu,
                     uint32_t qleft = 0x100u, const QT *pq = nullptr,
                     uint32_t pqoff = 0)
{
  // PQ: the whole query is in *pq (vsa_pq_stage / vsa_pq_load) and the
  // suffix searched starts at its offset pqoff; `query` points at the bytes
  // of that suffix for the lanes that fall back to them
  // qleft < 0x100 (MEM enumeration, needleft = the least length): the query
  // symbol in front of this suffix.  Only members of the deep bucket share D
  // or more symbols with the query; in a bucket of up to four their keys tell
  // which of them can reach the least length.  If every one of those has
  // qleft in front (the symbol is in the entry), no match of this work-item
  // is left maximal and nothing will be reported whatever the lengths
  // (leftrightsubmatch, fquery.c:139-270 -> PROCESSSUFFIX :54-81):
  // hit.notleftmax is set and the comparison on the text -- the expensive
  // part for reads that match end to end at every offset -- is skipped.
  // maxlcp is then only a lower bound.
  // AHEAD: see vsa_compare32.  needleft: hit.leftsym is wanted for matches
  // of at least this length (the MUM test).
  const uint32_t D = ix.D;
  int state = f.state;
  // first suffix of the deep bucket: a register pair only for wide tables
  const typename std::conditional<sizeof(IDX) == 4, uint32_t, uint64_t>::type
      dl = (typename std::conditional<sizeof(IDX) == 4, uint32_t,
                                      uint64_t>::type) f.left;
  const uint32_t cnt = f.cnt, qkey = f.qkey, limit = f.limit;
  const uint64_t first = f.first;

  const bool searching = state == VSA_LOC_FOUND;
  // only the first `limit` key symbols of the query exist
  const uint32_t qk = qkey >> (2 * (VSA_KEYSYMS - limit));

  // Buckets of up to four suffixes (nearly all of them: the deep prefix is
  // chosen so that a bucket holds about one) are fetched whole, together
  // with the entry behind them, in one round trip: five independent loads.
  const bool small = searching && cnt <= 4;
  const uint32_t ksh = 2 * (VSA_KEYSYMS - limit);
  uint64_t e[5] = {0, 0, 0, 0, 0};
  if (small && cnt == 1)
  {
    // the whole bucket came with the bounds.  The entry behind it belongs to
    // another bucket and shares fewer than D symbols with it, so its lcp byte
    // (0 here) is below every match length either way: nothing else is needed.
    e[0] = first;
  } else if (small)
  {
    // the first entry came with the bounds: entries 1, 2 in one load, and 3, 4
    // in a second one for the buckets that have them (a divergent load costs
    // its CU's address unit about a cycle per lane whatever its width, and a
    // bucket of two -- the usual case here -- needs one instead of three)
    // (esa8 has eight entries of slack behind index n)
    const uint64_t *p = ix.esa8 + (uint64_t) dl + 1;
    const vsa_u128 e12 = vsa_load16(p);
    e[0] = first;
    e[1] = e12.lo;
    e[2] = e12.hi;
    if (cnt > 2)
    {
      const vsa_u128 e34 = vsa_load16(p + 2);
      e[3] = e34.lo;
      e[4] = e34.hi;
    }
  }
  // lower bound on keys: lo = number of bucket entries whose key is smaller
  // than the query's
  uint32_t lo = 0;
  bool flagged = false;
  // qleft: is there a member of the bucket that shares enough symbols with
  // the query to be reported, and do all such members have qleft in front?
  bool anylong = false, allnotleft = true;
  const uint32_t enough = needleft < D + limit ? needleft : D + limit;
  if (small)
  {
#pragma unroll
    for (uint32_t i = 0; i < 4; i++)
    {
      if (i < cnt)
      {
        flagged = flagged || (e[i] & VSA_KEYFLAG) != 0;
        const uint32_t tk =
            ((uint32_t) (e[i] >> VSA_KEYSHIFT) & VSA_KEYMASK) >> ksh;
        lo += (tk < qk) ? 1u : 0u;
        if (qleft < 0x100u)
        {
          const bool islong = D + vsa_keylcp(tk, qk, limit) >= enough;
          const bool sameleft =
              (e[i] & VSA_LEFTSPECIAL) == 0 &&
              (uint32_t) ((e[i] >> VSA_LEFTSHIFT) & 3u) == qleft;
          anylong = anylong || islong;
          allnotleft = allnotleft && (!islong || sameleft);
        }
      }
    }
  }
  // larger buckets: binary search; trip count = that of the largest such
  // bucket in the wavefront, so the lanes stay converged
  uint32_t hi = (searching && !small) ? cnt : 0;
  uint32_t maxcnt = hi;
  if (__ballot(hi != 0) != 0) // rare: skip the six shuffle steps otherwise
  {
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1)
    {
      const uint32_t o = __shfl_xor(maxcnt, d, 64);
      maxcnt = o > maxcnt ? o : maxcnt;
    }
  }
  for (uint32_t span = maxcnt; span > 0; span >>= 1)
  {
    if (lo < hi)
    {
      const uint32_t mid = (lo + hi) >> 1;
      const uint64_t em = vsa_ld_entry(ix.esa8 + (uint64_t) dl + mid);
      flagged = flagged || (em & VSA_KEYFLAG) != 0;
      const uint32_t tk =
          ((uint32_t) (em >> VSA_KEYSHIFT) & VSA_KEYMASK) >> ksh;
      if (tk < qk)
      {
        lo = mid + 1;
      } else
      {
        hi = mid;
      }
    }
  }
  // neighbours of the insertion point: pred = lo-1, succ = lo, and succ+1
  // for the size of a tie.  They are taken by suffix-array index whether or
  // not they lie in the bucket: their lcp bytes serve the uniqueness test of
  // the located suffix.
  uint64_t epred = 0, esucc = 0, enext = 0;
  bool haspred = false, hassucc = false, hasnext = false;
  if (searching)
  {
    haspred = lo > 0;
    hassucc = lo < cnt;
    hasnext = lo + 1 < cnt;
    if (small)
    {
      // lo <= cnt <= 4: everything is in e[0..4]
      epred = lo == 1 ? e[0] : (lo == 2 ? e[1] : (lo == 3 ? e[2] : e[3]));
      esucc = lo == 0 ? e[0]
                      : (lo == 1 ? e[1]
                                 : (lo == 2 ? e[2] : (lo == 3 ? e[3] : e[4])));
      enext = lo == 0 ? e[1] : (lo == 1 ? e[2] : (lo == 2 ? e[3] : e[4]));
    } else
    {
      const uint64_t base = (uint64_t) dl + lo;
      if (base > 0)
      {
        epred = vsa_ld_entry(ix.esa8 + base - 1);
      }
      if (base <= ix.n)
      {
        esucc = vsa_ld_entry(ix.esa8 + base);
      }
      if (base + 1 <= ix.n)
      {
        enext = vsa_ld_entry(ix.esa8 + base + 1);
      }
    }
    flagged = flagged || (haspred && (epred & VSA_KEYFLAG) != 0) ||
              (hassucc && (esucc & VSA_KEYFLAG) != 0);
  }
  bool extend = false; // key tie: the text decides
  if (searching)
  {
    if (flagged)
    {
      state = VSA_LOC_SLOW;
    } else
    {
      const uint32_t kp = ((uint32_t) (epred >> VSA_KEYSHIFT) & VSA_KEYMASK)
                          >> ksh,
                     ks = ((uint32_t) (esucc >> VSA_KEYSHIFT) & VSA_KEYMASK)
                          >> ksh;
      const uint32_t lp = haspred ? vsa_keylcp(kp, qk, limit) : 0,
                     ls = hassucc ? vsa_keylcp(ks, qk, limit) : 0;
      if (hassucc && ls == limit)
      {
        // succ ties with the query on all key symbols.  More than one tying
        // suffix (lcp byte of the next entry >= D + limit) is left to the
        // reference walk.
        const uint32_t nextlcp = (uint32_t) (enext >> 32) & 0xFFu;
        if (hasnext && nextlcp >= D + limit)
        {
          state = VSA_LOC_SLOW;
        } else
        {
          extend = true;
          w = (uint64_t) dl + lo;
          maxlcp = D + limit;
        }
      } else if (ls > lp || !haspred)
      {
        w = (uint64_t) dl + lo;
        maxlcp = D + ls;
      } else
      {
        w = (uint64_t) dl + lo - 1;
        maxlcp = D + lp;
      }
      if (state == VSA_LOC_FOUND)
      {
        const bool atsucc = w == (uint64_t) dl + lo;
        hit.ew = atsucc ? esucc : epred;
        hit.lcpnext = (uint32_t) ((atsucc ? enext : esucc) >> 32) & 0xFFu;
      }
    }
  }
  // the symbol in front of the located suffix (left maximality) travels
  // with the first text words: one round trip less.  Front pad = separator.
  hit.notleftmax = false;
  if (state == VSA_LOC_FOUND && (extend || maxlcp >= needleft))
  {
    // the symbol in front of the located suffix travels in its entry: the
    // left maximality tests cost no access to the text
    hit.leftsym = (hit.ew & VSA_LEFTSPECIAL) != 0
                      ? (uint8_t) VSA_SEPARATOR
                      : (uint8_t) ((hit.ew >> VSA_LEFTSHIFT) & 3u);
    if (qleft < 0x100u && small && anylong && allnotleft)
    {
      hit.notleftmax = true;
      extend = false;
    }
  }
  // one text comparison for the lanes with a tie, all at the same time
  if (extend)
  {
    vsa_extend_tie<AHEAD, PQ>(ix, vsa_entrystart(ix, esucc, (uint64_t) dl + lo),
                              query, querylen, maxlcp, pq, pqoff);
  }
  return state;
}

template <int AHEAD = 1, bool PQ = false, typename IDX = uint32_t,
          typename QT = PackedQuery, typename QP = const uint8_t *>
__device__ __forceinline__ int
vsa_locate_deep(const DevIndex<IDX> &ix, bool active, QP query,
                uint32_t querylen, uint32_t &maxlcp,
                uint64_t &w, DeepHit &hit, uint32_t needleft = 0xFFFFFFFFu,
                uint32_t qleft = 0x100u, const QT *pq = nullptr,
                uint32_t pqoff = 0)
{
  DeepFront f;
  vsa_deep_front<PQ>(ix, active, query, querylen, f, pq, pqoff);
  return vsa_locate_deep_from<AHEAD, PQ>(ix, f, query, querylen, maxlcp, w,
                                         hit, needleft, qleft, pq, pqoff);
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

// The same for counts of 0 or 1 (the MUM modes: a work-item reports at most
// one candidate): a ballot and two population counts instead of six shuffle
// steps.  Must be reached by all 64 lanes.
__device__ __forceinline__ uint64_t
vsa_wave_reserve01(unsigned long long *cursor, bool one)
{
  const uint64_t mask = __ballot(one);
  const uint32_t lane = vsa_lane();
  const uint32_t total = (uint32_t) __popcll(mask),
                 before = (uint32_t) __popcll(mask & ((1ull << lane) - 1));
  uint64_t base = 0;
  if (lane == 0 && total > 0)
  {
    base = atomicAdd(cursor, (unsigned long long) total);
  }
  base = vsa_shfl64(base, 0);
  return base + before;
}

// inclusive prefix sum over the 64 lanes in vector operations only: shifts
// inside the rows of 16 lanes, then the last lane of a row broadcast to the
// rows behind it (lanes without a source add 0: bound_ctrl / the old value)
__device__ __forceinline__ uint32_t vsa_wave_inclusive_sum(uint32_t v)
{
  int x = (int) v;
  x += __builtin_amdgcn_update_dpp(0, x, 0x111, 0xF, 0xF, true); // row_shr:1
  x += __builtin_amdgcn_update_dpp(0, x, 0x112, 0xF, 0xF, true); // row_shr:2
  x += __builtin_amdgcn_update_dpp(0, x, 0x114, 0xF, 0xF, true); // row_shr:4
  x += __builtin_amdgcn_update_dpp(0, x, 0x118, 0xF, 0xF, true); // row_shr:8
  x += __builtin_amdgcn_update_dpp(0, x, 0x142, 0xA, 0xF, false); // row_bcast:15
  x += __builtin_amdgcn_update_dpp(0, x, 0x143, 0xC, 0xF, false); // row_bcast:31
  return (uint32_t) x;
}
