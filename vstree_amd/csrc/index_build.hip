// GPU construction of the enhanced suffix array mkvtree writes
// (tis -> suf, lcp + llv, bck, bwt), byte-identical to the reference's tables
// (definitions: Mkvtree/bese.c:27-49,533-566,602, Mkvtree/mkvprocess.c:251-400,
// kurtz/bwtcode.c:293-311).  The reference sorts with a bucket sort plus
// Bentley-Sedgewick multikey quicksort on one core (Mkvtree/ppsort.c:83,
// Mkvtree/bese.c:710); this builder is designed for the GPU instead:
//
//   B1  k_pack_keys      key(i) = first H symbols of suffix i, 3 (or more)
//                        bits each, specials and everything behind them
//                        masked: one streaming pass over the text
//   B2  rocPRIM radix sort of (key, i): stable, so suffixes that run into a
//                        special symbol inside H symbols are already in
//                        their final order (specials are unique symbols
//                        ordered by text position)
//   B3  prefix doubling (Larsson-Sadakane) only on the groups that are still
//                        tied: rank of suffix i+h as second key, h = H, 2H, ...
//   B4  k_lcp_chunks     Kasai's lcp(i+1) >= lcp(i)-1 inside chunks of
//                        consecutive text positions, one work-item per chunk,
//                        8 symbols per comparison; values >= 255 go to llv
//   B5  k_bck_*          bucket boundaries (left, mid) from the sorted order
//   B6  k_bwt            bwt[j] = tis[suf[j]-1]
//
// Every kernel is a template over the width of the tables (IDX = uint32_t for
// totallength + 1 < 2^32, uint64_t above: include/types.h:41-61 of the
// reference lifts the limit the same way, a 64-bit Uint).
#include <cstring>
#include <algorithm>
#include <cmath>
#include "esa_device.hpp"
#include <rocprim/rocprim.hpp>

#define VB_BLOCK 256
#define VB_LCP_CHUNK 32

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
  void free()
  {
    vsa_dev_free(p);
    p = nullptr;
  }
  template <typename T>
  T *as()
  {
    return (T *) p;
  }
};

inline dim3 gridfor(uint64_t items)
{
  return vsa_grid((items + VB_BLOCK - 1) / VB_BLOCK);
}

} // namespace

// ---- B1: keys -------------------------------------------------------------

// symbol -> code with `bits` bits: regular c -> c, special -> numofchars.
// The key holds H = 63/bits (at most) symbols, first symbol most significant;
// from the first special on everything is zero, and bit 63 is set iff the key
// contains a special: such keys are unique up to text position.
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_pack_keys(const uint8_t *__restrict__ tis, uint64_t n, uint32_t numofchars,
            uint32_t bits, uint32_t H, uint64_t *__restrict__ keys,
            IDX *__restrict__ sa)
{
  const uint64_t i = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (i > n)
  {
    return;
  }
  uint64_t key = 0;
  bool special = false;
  // tis is padded with 0xFF from position n on
  for (uint32_t k = 0; k < H; k++)
  {
    uint64_t c = 0;
    if (!special)
    {
      const uint8_t a = tis[i + k];
      if (VSA_ISSPECIAL(a))
      {
        special = true;
        c = numofchars;
      } else
      {
        c = a;
      }
    }
    key = (key << bits) | c;
  }
  keys[i] = key | (special ? (1ull << 63) : 0ull);
  sa[i] = (IDX) i;
}

// Caution on bit 63: it must not take part in the ORDER (a key with a special
// at offset j sorts by its symbols, the special code numofchars being the
// largest symbol); it is only a marker.  The sort runs on bits [0, 63).

// ---- B3: groups -----------------------------------------------------------

// head[j] = j where a new group starts, else 0 (then max-scanned)
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_initial_heads(const uint64_t *__restrict__ keys, uint64_t count,
                IDX *__restrict__ head)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j >= count)
  {
    return;
  }
  const uint64_t k = keys[j];
  const bool start = (j == 0) || (k >> 63) != 0 || keys[j - 1] != k;
  head[j] = start ? (IDX) j : (IDX) 0;
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_scatter_rank(const IDX *__restrict__ sa, const IDX *__restrict__ head,
               uint64_t count, IDX *__restrict__ isa)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j < count)
  {
    isa[sa[j]] = head[j];
  }
}

// flag[j] = 1 iff j sits in a group of more than one suffix
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_flag_unresolved(const IDX *__restrict__ head, uint64_t count,
                  uint8_t *__restrict__ flag)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j >= count)
  {
    return;
  }
  const bool tied = head[j] != (IDX) j ||
                    (j + 1 < count && head[j + 1] == (IDX) j);
  flag[j] = tied ? 1 : 0;
}

// for the tied positions pos[r]: composite key (group head, rank of the
// suffix h symbols further on) and the suffix itself.  The key takes
// 2 * VB_RANKBITS bits: 64 for 32-bit tables, 80 (in a 128-bit word) for wide
// ones.
template <typename IDX>
struct DoublingKey
{
  typedef uint64_t type;
  static constexpr unsigned int rankbits = 32;
};
template <>
struct DoublingKey<uint64_t>
{
  typedef rocprim::uint128_t type;
  static constexpr unsigned int rankbits = 40;
};

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_doubling_keys(const IDX *__restrict__ pos, uint64_t m,
                const IDX *__restrict__ sa, const IDX *__restrict__ head,
                const IDX *__restrict__ isa, uint64_t h, uint64_t n,
                typename DoublingKey<IDX>::type *__restrict__ ckey,
                IDX *__restrict__ csuf)
{
  const uint64_t r = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (r >= m)
  {
    return;
  }
  typedef typename DoublingKey<IDX>::type CK;
  const IDX j = pos[r];
  const IDX s = sa[j];
  // tied suffixes share h regular symbols, so s + h <= n
  const uint64_t next = (uint64_t) s + h;
  const IDX second = isa[next <= n ? next : n];
  ckey[r] = ((CK) head[j] << DoublingKey<IDX>::rankbits) | (CK) second;
  csuf[r] = s;
}

// after sorting the tied suffixes by composite key: new group starts
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_doubling_heads(const typename DoublingKey<IDX>::type *__restrict__ ckey,
                 const IDX *__restrict__ pos, uint64_t m,
                 IDX *__restrict__ newhead)
{
  const uint64_t r = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (r >= m)
  {
    return;
  }
  const bool start = (r == 0) || ckey[r - 1] != ckey[r];
  newhead[r] = start ? pos[r] : (IDX) 0;
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_doubling_writeback(const IDX *__restrict__ pos, const IDX *__restrict__ csuf,
                     const IDX *__restrict__ newhead, uint64_t m,
                     IDX *__restrict__ sa, IDX *__restrict__ head,
                     IDX *__restrict__ isa)
{
  const uint64_t r = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (r >= m)
  {
    return;
  }
  const IDX j = pos[r], s = csuf[r], hd = newhead[r];
  sa[j] = s;
  head[j] = hd;
  isa[s] = hd;
}

// among the previously tied positions: which are still tied
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_doubling_flags(const IDX *__restrict__ pos, const IDX *__restrict__ newhead,
                 uint64_t m, uint8_t *__restrict__ flag)
{
  const uint64_t r = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (r >= m)
  {
    return;
  }
  const bool tied = newhead[r] != pos[r] ||
                    (r + 1 < m && newhead[r + 1] == pos[r]);
  flag[r] = tied ? 1 : 0;
}

// ---- B4: lcp --------------------------------------------------------------

// number of leading regular symbols two suffixes share, starting the
// comparison at offset h (8 symbols per step; the text is padded with 0xFF)
__device__ __forceinline__ uint64_t vb_extend(const uint8_t *__restrict__ tis,
                                              uint64_t a, uint64_t b,
                                              uint64_t h)
{
  for (;;)
  {
    const uint64_t x = vsa_load8(tis + a + h), y = vsa_load8(tis + b + h);
    const uint64_t m = (x ^ y) | vsa_specialmask(x);
    if (m != 0)
    {
      return h + ((uint32_t) __builtin_ctzll(m) >> 3);
    }
    h += 8;
  }
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_lcp_chunks(const uint8_t *__restrict__ tis, uint64_t n,
             const IDX *__restrict__ sa, const IDX *__restrict__ isa,
             uint8_t *__restrict__ lcp, IDX *__restrict__ llvidx,
             IDX *__restrict__ llvval, uint64_t llvcap,
             unsigned long long *__restrict__ llvcount)
{
  const uint64_t c = vsa_bid() * VB_BLOCK + threadIdx.x;
  const uint64_t i0 = c * VB_LCP_CHUNK;
  if (i0 > n)
  {
    return;
  }
  const uint64_t i1 = (i0 + VB_LCP_CHUNK <= n + 1) ? i0 + VB_LCP_CHUNK : n + 1;
  uint64_t h = 0;
  for (uint64_t i = i0; i < i1; i++)
  {
    const IDX r = isa[i];
    if (r == 0)
    {
      lcp[0] = 0;
      h = 0;
      continue;
    }
    const IDX j = sa[r - 1];
    h = vb_extend(tis, i, j, h);
    if (h < 255)
    {
      lcp[r] = (uint8_t) h;
    } else
    {
      lcp[r] = 255;
      const unsigned long long slot = atomicAdd(llvcount, 1ull);
      if (slot < llvcap)
      {
        llvidx[slot] = r;
        llvval[slot] = (IDX) h;
      }
    }
    if (h > 0)
    {
      h--;
    }
  }
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_llv_pairs(const IDX *__restrict__ idx, const IDX *__restrict__ val,
            uint64_t m, IDX *__restrict__ llv)
{
  const uint64_t r = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (r < m)
  {
    llv[2 * r] = idx[r];
    llv[2 * r + 1] = val[r];
  }
}

// ---- B5: bck --------------------------------------------------------------

// code of the first pl symbols, a suffix cut short by a special symbol padded
// with the largest regular symbol: that is where it sorts
__device__ __forceinline__ uint64_t vb_padcode(const uint8_t *__restrict__ tis,
                                               uint64_t s, uint32_t pl,
                                               uint32_t numofchars, bool &cut)
{
  uint64_t c = 0;
  cut = false;
  for (uint32_t k = 0; k < pl; k++)
  {
    uint64_t a = numofchars - 1;
    if (!cut)
    {
      const uint8_t t = tis[s + k];
      if (VSA_ISSPECIAL(t))
      {
        cut = true;
      } else
      {
        a = t;
      }
    }
    c = c * numofchars + a;
  }
  return c;
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_bck_init(IDX *__restrict__ left, IDX *__restrict__ mid, uint64_t numofcodes)
{
  // grid-stride: 4^16 codes exceed the 2^32 work-items of one launch
  for (uint64_t c = vsa_bid() * VB_BLOCK + threadIdx.x;
       c < numofcodes; c += vsa_nblocks() * VB_BLOCK)
  {
    left[c] = ~(IDX) 0;
    mid[c] = ~(IDX) 0;
  }
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_bck_boundaries(const uint8_t *__restrict__ tis, uint64_t n,
                 const IDX *__restrict__ sa, uint32_t pl, uint32_t numofchars,
                 IDX *__restrict__ left, IDX *__restrict__ mid)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j > n)
  {
    return;
  }
  bool cut, prevcut = false;
  const uint64_t code = vb_padcode(tis, sa[j], pl, numofchars, cut);
  // the code of suffix j - 1 is what the lane below has just computed (a
  // second look at the text for it -- a random read per suffix -- was half of
  // this kernel's time: 0.7 s of a drop-in run's start at 3 Gbp); the first
  // lane of a wavefront computes it itself
  const uint32_t lane = threadIdx.x & 63u;
  uint64_t prevcode = ((uint64_t) (uint32_t) __shfl_up((int) (code >> 32), 1, 64)
                       << 32) |
                      (uint32_t) __shfl_up((int) (uint32_t) code, 1, 64);
  prevcut = __shfl_up(cut ? 1 : 0, 1, 64) != 0;
  if (lane == 0 && j > 0)
  {
    prevcode = vb_padcode(tis, sa[j - 1], pl, numofchars, prevcut);
  }
  if (j == 0 || prevcode != code)
  {
    left[code] = (IDX) j;
  }
  if (cut && (j == 0 || prevcode != code || !prevcut))
  {
    mid[code] = (IDX) j;
  }
}

// left[] has been min-scanned from the right: empty buckets start where the
// next occupied one starts.  A bucket without cut suffixes ends there, too.
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_bck_finish(const IDX *__restrict__ left, const IDX *__restrict__ mid,
             uint64_t numofcodes, IDX *__restrict__ bck)
{
  for (uint64_t c = vsa_bid() * VB_BLOCK + threadIdx.x;
       c < numofcodes; c += vsa_nblocks() * VB_BLOCK)
  {
    const IDX l = left[c];
    IDX m = mid[c];
    if (m == ~(IDX) 0)
    {
      // the last code always holds the end sentinel as a cut suffix, so c+1
      // exists whenever mid is unset
      m = left[c + 1];
    }
    bck[2 * c] = l;
    bck[2 * c + 1] = m;
  }
}

// ---- B6: bwt --------------------------------------------------------------

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_bwt(const uint8_t *__restrict__ tis, const IDX *__restrict__ sa, uint64_t n,
      uint8_t *__restrict__ bwt)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j <= n)
  {
    const IDX s = sa[j];
    bwt[j] = (s > 0) ? tis[s - 1] : (uint8_t) VSA_UNDEFBWT;
  }
}

// ---------------------------------------------------------------------------

namespace
{

// kurtz/detpfxlen.c:31-61 with sizeofbckentry = 16 (include/virtualdef.h:104,
// 64-bit Uint as in the reference's build)
uint32_t recommendedprefixlength(uint32_t numofchars, uint64_t totallength)
{
  const double value = (double) totallength / 16.0;
  if (value <= (double) numofchars)
  {
    return 1;
  }
  const uint32_t pl =
      (uint32_t) floor(log(value) / log((double) numofchars));
  return pl == 0 ? 1 : pl;
}

struct MaxOp
{
  template <typename T>
  __device__ T operator()(T a, T b) const
  {
    return a > b ? a : b;
  }
};

struct MinOp
{
  template <typename T>
  __device__ T operator()(T a, T b) const
  {
    return a < b ? a : b;
  }
};

template <typename IDX>
int maxscan_inplace(IDX *data, uint64_t count, hipStream_t stream)
{
  DevBuf temp;
  size_t tb = 0;
  VSA_HIP(rocprim::inclusive_scan(nullptr, tb, data, data, (size_t) count,
                                  MaxOp(), stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::inclusive_scan(temp.p, tb, data, data, (size_t) count,
                                  MaxOp(), stream));
  return 0;
}

} // namespace

namespace
{

template <typename IDX>
int build_bucket_table(const uint8_t *tis, uint64_t n, const IDX *sa,
                       uint32_t pl, uint32_t numofchars, IDX *out,
                       hipStream_t stream)
{
  DevBuf left, mid, temp;
  vsa_dev_set_stream(stream);
  uint64_t nc = 1;
  for (uint32_t k = 0; k < pl; k++)
  {
    nc *= numofchars;
  }
  const uint64_t count = n + 1;
  if (left.alloc((nc + 1) * sizeof(IDX)) || mid.alloc(nc * sizeof(IDX)))
  {
    return -100;
  }
  const unsigned int codegrid =
      (unsigned int) std::min<uint64_t>((nc + VB_BLOCK - 1) / VB_BLOCK,
                                        1u << 22);
  k_bck_init<IDX><<<codegrid, VB_BLOCK, 0, stream>>>(
      left.as<IDX>(), mid.as<IDX>(), nc);
  VSA_HIP(hipGetLastError());
  k_bck_boundaries<IDX><<<gridfor(count), VB_BLOCK, 0, stream>>>(
      tis, n, sa, pl, numofchars, left.as<IDX>(), mid.as<IDX>());
  VSA_HIP(hipGetLastError());
  // suffix-min over the codes = inclusive min-scan on the reversed array
  size_t tb = 0;
  auto rin = rocprim::make_reverse_iterator(left.as<IDX>() + nc);
  VSA_HIP(rocprim::inclusive_scan(nullptr, tb, rin, rin, (size_t) nc, MinOp(),
                                  stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::inclusive_scan(temp.p, tb, rin, rin, (size_t) nc, MinOp(),
                                  stream));
  k_bck_finish<IDX><<<codegrid, VB_BLOCK, 0, stream>>>(
      left.as<IDX>(), mid.as<IDX>(), nc, out);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipStreamSynchronize(stream));
  return 0;
}

} // namespace

int vsa_build_bucket_table(const uint8_t *tis, uint64_t n, const uint32_t *sa,
                           uint32_t pl, uint32_t numofchars, uint32_t *out,
                           hipStream_t stream)
{
  return build_bucket_table<uint32_t>(tis, n, sa, pl, numofchars, out, stream);
}

int vsa_build_bucket_table(const uint8_t *tis, uint64_t n, const uint64_t *sa,
                           uint32_t pl, uint32_t numofchars, uint64_t *out,
                           hipStream_t stream)
{
  return build_bucket_table<uint64_t>(tis, n, sa, pl, numofchars, out, stream);
}

namespace
{

template <typename IDX>
int build_tables(vsa_index *ix)
{
  typedef typename DoublingKey<IDX>::type CK;
  hipStream_t stream = ix->stream;
  vsa_dev_set_stream(stream);
  const uint64_t n = ix->n, count = n + 1;
  const uint8_t *tis = ix->tis_alloc + VSA_TIS_FRONTPAD;
  IDX *sa = (IDX *) ix->suf;
  uint32_t bits = 1;
  while ((1u << bits) < ix->numofchars + 1)
  {
    bits++;
  }
  const uint32_t H = 63 / bits;
  DevBuf isa, head;

  // B1 + B2
  {
    DevBuf keys, keys2, sa2, temp;
    if (keys.alloc(count * 8) || keys2.alloc(count * 8) ||
        sa2.alloc(count * sizeof(IDX)))
    {
      return -100;
    }
    k_pack_keys<IDX><<<gridfor(count), VB_BLOCK, 0, stream>>>(
        tis, n, ix->numofchars, bits, H, keys.as<uint64_t>(),
        sa2.as<IDX>());
    VSA_HIP(hipGetLastError());
    size_t tb = 0;
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, keys.as<uint64_t>(), keys2.as<uint64_t>(),
        sa2.as<IDX>(), sa, (size_t) count, 0u, H * bits, stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, keys.as<uint64_t>(), keys2.as<uint64_t>(),
        sa2.as<IDX>(), sa, (size_t) count, 0u, H * bits, stream));
    temp.free();
    keys.free();
    sa2.free();
    if (isa.alloc(count * sizeof(IDX)) || head.alloc(count * sizeof(IDX)))
    {
      return -100;
    }
    k_initial_heads<IDX><<<gridfor(count), VB_BLOCK, 0, stream>>>(
        keys2.as<uint64_t>(), count, head.as<IDX>());
    VSA_HIP(hipGetLastError());
  }
  if (maxscan_inplace(head.as<IDX>(), count, stream))
  {
    return -100;
  }
  k_scatter_rank<IDX><<<gridfor(count), VB_BLOCK, 0, stream>>>(
      sa, head.as<IDX>(), count, isa.as<IDX>());
  VSA_HIP(hipGetLastError());

  // B3: positions still tied, refined until none is left
  {
    DevBuf flag, pos, dcount, temp, iota;
    uint64_t m = 0;
    if (flag.alloc(count) || dcount.alloc(8))
    {
      return -100;
    }
    k_flag_unresolved<IDX><<<gridfor(count), VB_BLOCK, 0, stream>>>(
        head.as<IDX>(), count, flag.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    if (pos.alloc(count * sizeof(IDX)))
    {
      return -100;
    }
    {
      size_t tb = 0;
      auto counting = rocprim::counting_iterator<IDX>(0);
      VSA_HIP(rocprim::select(nullptr, tb, counting, flag.as<uint8_t>(),
                              pos.as<IDX>(), dcount.as<uint64_t>(),
                              (size_t) count, stream));
      if (temp.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::select(temp.p, tb, counting, flag.as<uint8_t>(),
                              pos.as<IDX>(), dcount.as<uint64_t>(),
                              (size_t) count, stream));
      VSA_HIP(hipMemcpyAsync(&m, dcount.p, 8, hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipStreamSynchronize(stream));
    }
    flag.free();
    uint64_t h = H;
    const char *trace = getenv("VSA_BUILD_TRACE");
    while (m > 0)
    {
      if (trace != nullptr)
      {
        fprintf(stderr, "vsa build: %lu suffixes tied after %lu symbols\n",
                (unsigned long) m, (unsigned long) h);
      }
      DevBuf ckey, ckey2, csuf, csuf2, newhead, rflag, pos2, t2;
      if (ckey.alloc(m * sizeof(CK)) || ckey2.alloc(m * sizeof(CK)) ||
          csuf.alloc(m * sizeof(IDX)) || csuf2.alloc(m * sizeof(IDX)) ||
          newhead.alloc(m * sizeof(IDX)) || rflag.alloc(m) ||
          pos2.alloc(m * sizeof(IDX)))
      {
        return -100;
      }
      k_doubling_keys<IDX><<<gridfor(m), VB_BLOCK, 0, stream>>>(
          pos.as<IDX>(), m, sa, head.as<IDX>(), isa.as<IDX>(),
          h, n, ckey.as<CK>(), csuf.as<IDX>());
      VSA_HIP(hipGetLastError());
      size_t tb = 0;
      VSA_HIP(rocprim::radix_sort_pairs(
          nullptr, tb, ckey.as<CK>(), ckey2.as<CK>(),
          csuf.as<IDX>(), csuf2.as<IDX>(), (size_t) m, 0u,
          2 * DoublingKey<IDX>::rankbits, stream));
      if (t2.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::radix_sort_pairs(
          t2.p, tb, ckey.as<CK>(), ckey2.as<CK>(),
          csuf.as<IDX>(), csuf2.as<IDX>(), (size_t) m, 0u,
          2 * DoublingKey<IDX>::rankbits, stream));
      k_doubling_heads<IDX><<<gridfor(m), VB_BLOCK, 0, stream>>>(
          ckey2.as<CK>(), pos.as<IDX>(), m,
          newhead.as<IDX>());
      VSA_HIP(hipGetLastError());
      if (maxscan_inplace(newhead.as<IDX>(), m, stream))
      {
        return -100;
      }
      k_doubling_writeback<IDX><<<gridfor(m), VB_BLOCK, 0, stream>>>(
          pos.as<IDX>(), csuf2.as<IDX>(), newhead.as<IDX>(), m,
          sa, head.as<IDX>(), isa.as<IDX>());
      VSA_HIP(hipGetLastError());
      k_doubling_flags<IDX><<<gridfor(m), VB_BLOCK, 0, stream>>>(
          pos.as<IDX>(), newhead.as<IDX>(), m,
          rflag.as<uint8_t>());
      VSA_HIP(hipGetLastError());
      uint64_t m2 = 0;
      tb = 0;
      VSA_HIP(rocprim::select(nullptr, tb, pos.as<IDX>(),
                              rflag.as<uint8_t>(), pos2.as<IDX>(),
                              dcount.as<uint64_t>(), (size_t) m, stream));
      if (t2.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::select(t2.p, tb, pos.as<IDX>(),
                              rflag.as<uint8_t>(), pos2.as<IDX>(),
                              dcount.as<uint64_t>(), (size_t) m, stream));
      VSA_HIP(hipMemcpyAsync(&m2, dcount.p, 8, hipMemcpyDeviceToHost,
                             stream));
      VSA_HIP(hipStreamSynchronize(stream));
      if (m2 > 0)
      {
        VSA_HIP(hipMemcpyAsync(pos.p, pos2.p, m2 * sizeof(IDX),
                               hipMemcpyDeviceToDevice, stream));
        VSA_HIP(hipStreamSynchronize(stream));
      }
      m = m2;
      h *= 2;
      if (h > 2 * count + 2 * H && m > 0)
      {
        VSA_ERROR("suffix sorting did not converge");
        return -6;
      }
    }
  }
  head.free();

  // B4: lcp + exceptions
  {
    DevBuf llvidx, llvval, cnt, sidx, sval, temp;
    uint64_t llvcap = 1 << 20, needed = 0;
    if (cnt.alloc(8))
    {
      return -100;
    }
    for (int attempt = 0; attempt < 2; attempt++)
    {
      if (llvidx.alloc(llvcap * sizeof(IDX)) ||
          llvval.alloc(llvcap * sizeof(IDX)))
      {
        return -100;
      }
      VSA_HIP(hipMemsetAsync(cnt.p, 0, 8, stream));
      const uint64_t chunks = (count + VB_LCP_CHUNK - 1) / VB_LCP_CHUNK;
      k_lcp_chunks<IDX><<<gridfor(chunks), VB_BLOCK, 0, stream>>>(
          tis, n, sa, isa.as<IDX>(), ix->lcp, llvidx.as<IDX>(),
          llvval.as<IDX>(), llvcap,
          cnt.as<unsigned long long>());
      VSA_HIP(hipGetLastError());
      VSA_HIP(hipMemcpyAsync(&needed, cnt.p, 8, hipMemcpyDeviceToHost,
                             stream));
      VSA_HIP(hipStreamSynchronize(stream));
      if (needed <= llvcap)
      {
        break;
      }
      llvcap = needed;
    }
    isa.free();
    ix->nllv = needed;
    (void) hipFree(ix->llv);
    ix->llv = nullptr;
    VSA_HIP(vsa_hip_malloc(&ix->llv, 2 * needed * sizeof(IDX) + 16));
    ix->device_bytes += 2 * needed * sizeof(IDX);
    if (needed > 0)
    {
      // exceptions sorted by index (Mkvtree/bese.c:557-566 emits them so)
      if (sidx.alloc(needed * sizeof(IDX)) || sval.alloc(needed * sizeof(IDX)))
      {
        return -100;
      }
      size_t tb = 0;
      VSA_HIP(rocprim::radix_sort_pairs(
          nullptr, tb, llvidx.as<IDX>(), sidx.as<IDX>(),
          llvval.as<IDX>(), sval.as<IDX>(), (size_t) needed, 0u,
          (unsigned int) (8 * sizeof(IDX)), stream));
      if (temp.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::radix_sort_pairs(
          temp.p, tb, llvidx.as<IDX>(), sidx.as<IDX>(),
          llvval.as<IDX>(), sval.as<IDX>(), (size_t) needed, 0u,
          (unsigned int) (8 * sizeof(IDX)), stream));
      k_llv_pairs<IDX><<<gridfor(needed), VB_BLOCK, 0, stream>>>(
          sidx.as<IDX>(), sval.as<IDX>(), needed,
          (IDX *) ix->llv);
      VSA_HIP(hipGetLastError());
      VSA_HIP(hipStreamSynchronize(stream));
    }
  }

  // B5: bck
  if (build_bucket_table<IDX>(tis, n, sa, ix->pl, ix->numofchars,
                              (IDX *) ix->bck, stream))
  {
    return -100;
  }

  // B6: bwt
  if (ix->bwt != nullptr)
  {
    k_bwt<IDX><<<gridfor(count), VB_BLOCK, 0, stream>>>(tis, sa, n, ix->bwt);
    VSA_HIP(hipGetLastError());
  }
  VSA_HIP(hipStreamSynchronize(stream));
  return 0;
}

int build_common(const void *src, bool srcondevice, uint64_t totallength,
                 uint32_t numofchars, uint32_t prefixlength, int device,
                 vsa_index **index)
{
  if (index == nullptr || (src == nullptr && totallength > 0))
  {
    VSA_ERROR("vsa_index_build: NULL argument");
    return -1;
  }
  *index = nullptr;
  if (numofchars == 0 || numofchars > 253)
  {
    VSA_ERROR("numofchars=%u is not a usable alphabet size", numofchars);
    return -2;
  }
  if (totallength + 1 >= (1ull << DoublingKey<uint64_t>::rankbits))
  {
    VSA_ERROR("totallength=%lu: the GPU index builder handles texts below "
              "2^%u symbols", (unsigned long) totallength,
              DoublingKey<uint64_t>::rankbits);
    return -3;
  }
  if (prefixlength == 0)
  {
    prefixlength = recommendedprefixlength(numofchars, totallength);
  }
  if (pow((double) numofchars, (double) prefixlength) > 4.0e9)
  {
    VSA_ERROR("prefixlength=%u is too large for alphabet size %u",
              prefixlength, numofchars);
    return -4;
  }
  vsa_index *ix = nullptr;
  // (VSA_FORCE_WIDE=1 builds 64-bit tables for a short text, too)
  int rc = vsa_index_alloc(totallength, prefixlength, numofchars, 0, true,
                           device, &ix, true);
  if (rc != 0)
  {
    vsa_index_close(ix);
    return rc;
  }
  if (totallength > 0)
  {
    if (hipMemcpy(ix->tis_alloc + VSA_TIS_FRONTPAD, src, totallength,
                  srcondevice ? hipMemcpyDeviceToDevice
                              : hipMemcpyHostToDevice) != hipSuccess)
    {
      VSA_ERROR("copy of the text failed");
      vsa_index_close(ix);
      return -100;
    }
  }
  rc = ix->isize == 4 ? build_tables<uint32_t>(ix)
                      : build_tables<uint64_t>(ix);
  if (rc == 0)
  {
    rc = vsa_index_make_esa8(ix);
  }
  if (rc != 0)
  {
    vsa_index_close(ix);
    return rc;
  }
  *index = ix;
  return 0;
}

} // namespace

// ---- sti1 (Mkvtree/mkvprocess.c:583-612) ------------------------------------

// runstart[j] = j where lcp[j] < prefixlength (a new bucket run starts), else
// 0; after a max-scan sti1[suf[j]] = min(255, j - runstart[j])
template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_sti1_runstarts(const uint8_t *__restrict__ lcp, uint64_t count, uint32_t pl,
                 IDX *__restrict__ runstart)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j < count)
  {
    runstart[j] = (j == 0 || lcp[j] < (uint8_t) pl) ? (IDX) j : (IDX) 0;
  }
}

template <typename IDX>
__global__ void __launch_bounds__(VB_BLOCK)
k_sti1_scatter(const IDX *__restrict__ sa, const IDX *__restrict__ runstart,
               uint64_t count, uint8_t *__restrict__ sti1)
{
  const uint64_t j = vsa_bid() * VB_BLOCK + threadIdx.x;
  if (j < count)
  {
    const uint64_t d = j - runstart[j];
    sti1[sa[j]] = (uint8_t) (d < 255 ? d : 255);
  }
}

namespace
{

template <typename IDX>
int make_sti1(const vsa_index *ix, uint8_t *sti1)
{
  const uint64_t count = ix->n + 1;
  DevBuf runstart, out;
  vsa_dev_set_stream(ix->stream);
  if (runstart.alloc(count * sizeof(IDX)) || out.alloc(count))
  {
    return -100;
  }
  k_sti1_runstarts<IDX><<<gridfor(count), VB_BLOCK, 0, ix->stream>>>(
      ix->lcp, count, ix->pl, runstart.as<IDX>());
  VSA_HIP(hipGetLastError());
  if (maxscan_inplace(runstart.as<IDX>(), count, ix->stream))
  {
    return -100;
  }
  k_sti1_scatter<IDX><<<gridfor(count), VB_BLOCK, 0, ix->stream>>>(
      (const IDX *) ix->suf, runstart.as<IDX>(), count, out.as<uint8_t>());
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipStreamSynchronize(ix->stream));
  VSA_HIP(hipMemcpy(sti1, out.p, count, hipMemcpyDeviceToHost));
  return 0;
}

} // namespace

extern "C" int vsa_index_make_sti1(const vsa_index *ix, uint8_t *sti1)
{
  if (ix == nullptr || sti1 == nullptr)
  {
    VSA_ERROR("vsa_index_make_sti1: NULL argument");
    return -1;
  }
  if (vsa_set_device(ix->device) != 0)
  {
    return -100;
  }
  return ix->isize == 4 ? make_sti1<uint32_t>(ix, sti1)
                        : make_sti1<uint64_t>(ix, sti1);
}

extern "C" int vsa_index_build(const uint8_t *tis, uint64_t totallength,
                               uint32_t numofchars, uint32_t prefixlength,
                               int device, vsa_index **index)
{
  return build_common(tis, false, totallength, numofchars, prefixlength,
                      device, index);
}

extern "C" int vsa_index_build_device(const void *device_tis,
                                      uint64_t totallength,
                                      uint32_t numofchars,
                                      uint32_t prefixlength, int device,
                                      vsa_index **index)
{
  return build_common(device_tis, true, totallength, numofchars,
                      prefixlength, device, index);
}
