// The search engine on MI355X: kernels (in the .inc files below, one
// translation unit) and their host-side pipelines.  DESIGN.md has the byte
// budgets and the measurements.
//
//   search_complete.inc  K1  k_complete_search   one work-item per query:
//                            locate -> lcptab widening -> suffix-array
//                            interval [left, left+count)
//                            k_complete_expand   one workgroup per 256 queries
//   search_query.inc     K2  k_query_search      one work-item per (query,
//                            offset): locate -> MEM enumeration or
//                            MUM-candidate test; wavefront-aggregated append
//                            into sharded regions, compaction, stable radix
//                            sort by work-item number = reference order
//   mum_workplan.inc     K2a k_mum_first / k_mum_plan / k_expand_plan (and
//                            the older k_mum_anchor): which offsets of a
//                            query can be MUM candidates at all
//   mum_filter.inc       K4  candidates as (sort key, value) pairs sorted by
//                            dbstart, prefix-max scan of the right ends,
//                            flags from the keys (runs of equal dbstarts
//                            looked at as runs), survivors written as records
//                            in order (kurtz/cleanMUMcand.c:55-118)
//   selfmum_scan.inc     K3  k_selfmum_peaks / k_selfmum_emit: streaming scan
//                            over lcptab + bwttab for indexes that hold
//                            their queries (Vmengine/fmumself.c:10-66)
//   approx_search.inc    A   approximate complete matches (-complete -e/-h)
//   selfmatch_search.inc R,S,T maximal / supermaximal / tandem repeats of the
//                          index
//
// rocPRIM supplies radix sort / scan / reduce only.
//
// Switches (environment, read when an index is created):
//   VSA_TUNE=2        no MUM work reduction: every offset of every read is
//                     searched by the list form of the search kernel (the
//                     cross-check of first pass + work plan)
//   VSA_NO_ESA8=1     no deep tables: the reference walk, probe for probe
//   VSA_DEEP_PREFIX=D their depth (default ceil(log4 n), at most 16)
//   VSA_FORCE_WIDE=1  64-bit device tables whatever the size of the text
// What was measured and dropped (32-byte slots, the two-phase and the deferring
// search kernel, the anchor pass, the list form of planned batches, the filter
// on rocPRIM scans, ...) is described in DESIGN.md section 4 with its numbers
// under profiles/; the code left with round 4.
#include <cstring>
#include <algorithm>
#include "esa_device.hpp"
#include <rocprim/rocprim.hpp>

#define VSA_BLOCK 256
#define VSA_CURSOR_STRIDE 8   // uint64 words: one cursor per 64-byte line
#define VSA_CURSOR_SHARDS 2048 // power of two

#include "search_complete.inc"
#include "search_query.inc"
#include "mum_workplan.inc"
#include "mem_workplan.inc"
#include "mum_filter.inc"
#include "selfmum_scan.inc"

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

// Small results the host needs before it can go on (counts, maxima) come
// back through a page of pinned memory: a device-to-host copy into pageable
// memory is staged by the runtime and costs 30-150 us each, several times
// per batch.  The page lives as long as the thread (never freed: the runtime
// may be gone when thread-local destructors run).
struct Fetch
{
  const void *src;
  size_t bytes; // <= 8
};

inline int fetchwords(hipStream_t stream, const Fetch *items, int count,
                      uint64_t *out)
{
  static thread_local uint64_t *page = nullptr;
  if (page == nullptr)
  {
    void *v = nullptr;
    VSA_HIP(hipHostMalloc(&v, 4096, hipHostMallocDefault));
    page = (uint64_t *) v;
  }
  if (count > 512)
  {
    return -100;
  }
  for (int i = 0; i < count;)
  {
    // words that sit next to each other on the device travel as one copy
    // (a copy of 8 bytes takes the GPU 5 us: five of them behind the search
    // kernel were 25 us of a 4 ms step)
    int j = i + 1;
    size_t bytes = items[i].bytes;
    page[i] = 0;
    while (j < count && items[j].bytes == 8 && items[j - 1].bytes == 8 &&
           (const char *) items[j].src == (const char *) items[j - 1].src + 8)
    {
      page[j] = 0;
      bytes += 8;
      j++;
    }
    VSA_HIP(hipMemcpyAsync(page + i, items[i].src, bytes,
                           hipMemcpyDeviceToHost, stream));
    i = j;
  }
  VSA_HIP(hipStreamSynchronize(stream));
  for (int i = 0; i < count; i++)
  {
    out[i] = page[i];
  }
  return 0;
}

// offsets[sh] = sum of the fill counts of the cursor regions before sh;
// summary = {total, largest count, *extra_a, *extra_b}: one workgroup
__global__ void __launch_bounds__(1024)
k_shard_summary(const unsigned long long *__restrict__ cursors,
                uint32_t nshards, uint64_t *__restrict__ offsets,
                const uint32_t *__restrict__ extra_a,
                const uint32_t *__restrict__ extra_b,
                uint64_t *__restrict__ summary)
{
  __shared__ uint64_t sums[1024], maxs[1024];
  const uint32_t per = (nshards + 1023) / 1024, t = threadIdx.x;
  uint64_t mine = 0, mx = 0;
  for (uint32_t k = 0; k < per; k++)
  {
    const uint32_t sh = t * per + k;
    if (sh < nshards)
    {
      const uint64_t c = cursors[(uint64_t) sh * VSA_CURSOR_STRIDE];
      mine += c;
      mx = c > mx ? c : mx;
    }
  }
  sums[t] = mine;
  maxs[t] = mx;
  __syncthreads();
  for (uint32_t d = 1; d < 1024; d <<= 1)
  {
    const uint64_t a = t >= d ? sums[t - d] : 0,
                   b = t >= d ? maxs[t - d] : 0;
    __syncthreads();
    sums[t] += a;
    maxs[t] = b > maxs[t] ? b : maxs[t];
    __syncthreads();
  }
  uint64_t run = sums[t] - mine;
  for (uint32_t k = 0; k < per; k++)
  {
    const uint32_t sh = t * per + k;
    if (sh < nshards)
    {
      offsets[sh] = run;
      run += cursors[(uint64_t) sh * VSA_CURSOR_STRIDE];
    }
  }
  if (t == 1023)
  {
    summary[0] = sums[t];
    summary[1] = maxs[t];
    summary[2] = extra_a != nullptr ? *extra_a : 0;
    summary[3] = extra_b != nullptr ? *extra_b : 0;
  }
}

inline uint64_t blocksfor(uint64_t items)
{
  return (items + VSA_BLOCK - 1) / VSA_BLOCK;
}

inline dim3 gridfor(uint64_t items)
{
  return vsa_grid(blocksfor(items));
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

__global__ void __launch_bounds__(VSA_BLOCK)
k_iota_u32(uint32_t *__restrict__ out, uint64_t n)
{
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (t < n)
  {
    out[t] = (uint32_t) t;
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_gather_matches(const vsa_match *__restrict__ in,
                 const uint32_t *__restrict__ order, uint64_t n,
                 vsa_match *__restrict__ out)
{
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (t < n)
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(in + order[t]);
    uint4 *dst = reinterpret_cast<uint4 *>(out + t);
    const uint4 lo = src[0], hi = src[1];
    dst[0] = lo;
    dst[1] = hi;
  }
}

// stable sort of (key, match) pairs by key bits [0, endbit); results land in
// keys_out / matches_out.  The 32-byte records do not travel through the
// radix passes: (key, index) pairs do, and one gather follows.
int sortbykey(uint64_t *keys_in, uint64_t *keys_out, vsa_match *in,
              vsa_match *out, uint64_t n, unsigned int endbit,
              hipStream_t stream)
{
  DevBuf temp;
  size_t tb = 0;
  if (n >= 0xFFFFFFFFull)
  {
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
  DevBuf order, order2;
  if (order.alloc(n * 4 + 4) || order2.alloc(n * 4 + 4))
  {
    return -100;
  }
  k_iota_u32<<<gridfor(n), VSA_BLOCK, 0, stream>>>(order.as<uint32_t>(), n);
  VSA_HIP(hipGetLastError());
  VSA_HIP(rocprim::radix_sort_pairs(nullptr, tb, keys_in, keys_out,
                                    order.as<uint32_t>(),
                                    order2.as<uint32_t>(), (size_t) n, 0u,
                                    endbit, stream));
  if (temp.alloc(tb) != 0)
  {
    return -100;
  }
  VSA_HIP(rocprim::radix_sort_pairs(temp.p, tb, keys_in, keys_out,
                                    order.as<uint32_t>(),
                                    order2.as<uint32_t>(), (size_t) n, 0u,
                                    endbit, stream));
  k_gather_matches<<<gridfor(n), VSA_BLOCK, 0, stream>>>(
      in, order2.as<uint32_t>(), n, out);
  VSA_HIP(hipGetLastError());
  return 0;
}

vsa_result *newresult(int device)
{
  vsa_result *r = new vsa_result;
  r->device = device;
  r->count = 0;
  r->matches = nullptr;
  r->packbits = 0;
  r->packvals = nullptr;
  memset(&r->stats, 0, sizeof r->stats);
  return r;
}

// ---- K1 pipeline ----

template <typename IDX>
int run_complete(const vsa_index *index, const vsa_queries *queries,
                 uint64_t qlimit, vsa_result *res)
{
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  // packed batches: read from their rows by the deep kernel; an index
  // without deep tables (or reads beyond four words) takes their bytes
  const bool rows = queries->rows != nullptr && ix.esa8 != nullptr &&
                    queries->roww <= 4 && queries->maxlength >= ix.D;
  if (queries->rows != nullptr && !rows &&
      vsa_queries_bytes(queries, stream) != 0)
  {
    return -100;
  }
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
  {
    const bool staged = ix.esa8 != nullptr && qs.dense != 0 &&
                        qs.symbols != nullptr && qs.uniformlen <= 128 &&
                        (qs.uniformlen & 3u) == 0 && qs.uniformlen >= ix.D;
    if (rows)
    {
      k_complete_search<IDX, true, true, true>
          <<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
              ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    } else if (staged)
    {
      k_complete_search<IDX, true, true>
          <<<gridfor(qlimit), VSA_BLOCK, (size_t) VSA_BLOCK * qs.uniformlen,
             stream>>>(ix, qs, qlimit, left.as<uint64_t>(),
                       count.as<uint64_t>());
    } else if (ix.esa8 != nullptr)
    {
      k_complete_search<IDX, true><<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
          ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    } else
    {
      k_complete_search<IDX, false>
          <<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
              ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    }
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
    k_complete_expand<IDX><<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
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
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t lenmask = (1ull << lenbits) - 1;
    key[i] = (cand[i].dbstart << lenbits) | (lenmask - cand[i].length);
    idx[i] = (uint32_t) i;
  }
}

// carry = the reference's running `dbright` when it reaches the first of
// these candidates: 0 for a whole job, the largest right end of all
// candidates with a smaller dbstart when the list is one dbstart range of a
// job that is filtered in pieces (multi-GPU).  *maxright (optional) receives
// the largest right end in this list.
int mumuniqueinquery(DevBuf &cand, uint64_t ncand, hipStream_t stream,
                     DevBuf &mums, uint64_t *nmums, uint64_t carry = 0,
                     uint64_t *maxright = nullptr, uint64_t dbbound = 0,
                     uint64_t lenbound = 0, uint64_t *sumlength = nullptr)
{
  // *sumlength (if asked for) = sum of the lengths of the MUMs, or ~0 if this
  // call did not compute it
  // dbbound / lenbound: upper bounds of dbstart and length if the caller
  // knows them (index length, longest query), else 0: they are looked up
  *nmums = 0;
  if (maxright != nullptr)
  {
    *maxright = 0;
  }
  if (sumlength != nullptr)
  {
    *sumlength = (ncand == 0) ? 0 : ~0ull;
  }
  if (ncand == 0)
  {
    return 0;
  }
  DevBuf ends, dbright, keep, temp, dcount, sorted, k1, k2, i1, i2;
  bool onkeys = false; // the filter runs on the sorted composite keys
  if (ends.alloc(ncand * 8) || dbright.alloc(ncand * 8) ||
      keep.alloc(ncand) || dcount.alloc(sizeof(MaxPair) + 16))
  {
    return -100;
  }
  // how many bits do dbstart and length need?
  MaxPair mx;
  mx.db = dbbound;
  mx.len = lenbound;
  if (dbbound == 0 || lenbound == 0)
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
    // one radix sort of (composite key, index) over just the bits in use;
    // the keys carry all the filter looks at
    onkeys = true;
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
    k_mum_keyends<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), ncand, lenbits, ends.as<uint64_t>());
    VSA_HIP(hipGetLastError());
  } else
  {
    // wide values: least significant key first (length descending), then a
    // stable sort by dbstart
    DevBuf kout;
    if (k1.alloc(ncand * 8) || k2.alloc(ncand * 8) || kout.alloc(ncand * 8) ||
        sorted.alloc(ncand * sizeof(vsa_match)))
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
  if (mums.alloc(ncand * sizeof(vsa_match)))
  {
    return -100;
  }
  if (onkeys)
  {
    DevBuf slots;
    uint64_t hsum = 0;
    if (slots.alloc(ncand * 4))
    {
      return -100;
    }
    k_mum_keyflags<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), dbright.as<uint64_t>(), ncand, lenbits,
        keep.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    auto keepit =
        rocprim::make_transform_iterator(keep.as<uint8_t>(), KeepToU32());
    tb = 0;
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(temp.p, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    DevBuf blocksum;
    const size_t nblocks = blocksfor(ncand);
    if (blocksum.alloc(vsa_grid_blocks(nblocks) * 8))
    {
      return -100;
    }
    k_mum_writekept<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), i2.as<uint32_t>(), keep.as<uint8_t>(),
        slots.as<uint32_t>(), ncand, mums.as<vsa_match>(),
        dcount.as<uint64_t>(), blocksum.as<unsigned long long>());
    VSA_HIP(hipGetLastError());
    tb = 0;
    VSA_HIP(rocprim::reduce(nullptr, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull,
                            nblocks, rocprim::plus<unsigned long long>(),
                            stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(temp.p, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull,
                            nblocks, rocprim::plus<unsigned long long>(),
                            stream));
    {
      const Fetch f[2] = {{dcount.p, 8}, {dcount.as<uint64_t>() + 1, 8}};
      uint64_t got[2];
      if (fetchwords(stream, f, 2, got))
      {
        return -100;
      }
      *nmums = got[0];
      hsum = got[1];
    }
    if (sumlength != nullptr)
    {
      *sumlength = hsum;
    }
    return 0;
  }
  k_mum_flags<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
      sorted.as<vsa_match>(), ends.as<uint64_t>(), dbright.as<uint64_t>(),
      ncand, keep.as<uint8_t>());
  VSA_HIP(hipGetLastError());
  if (compact_matches(sorted.as<vsa_match>(), keep.as<uint8_t>(), ncand,
                      mums.as<vsa_match>(), dcount.as<uint64_t>(), stream))
  {
    return -100;
  }
  VSA_HIP(hipMemcpyAsync(nmums, dcount.p, 8, hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  return 0;
}

// The filter of mumuniqueinquery on a packed candidate list: keys =
// dbstart << lenbits | (2^lenbits - 1 - length), values = queryseq << 16 |
// querystart (k_query_search with packbits).  One sort of the pairs, the
// filter on the keys, and the surviving pairs become the records: the
// candidates never exist as 32-byte records.
// keys_in / vals_in: anything rocPRIM can read (pointers, or iterators over
// rows of pairs as they come out of the exchange: no copy into two arrays)
template <typename VAL = uint64_t, typename KeyIn = const uint64_t *,
          typename ValIn = const VAL *>
int mumfilter_packed(KeyIn keys_in, ValIn vals_in, uint64_t ncand,
                     unsigned int lenbits, unsigned int dbbits,
                     hipStream_t stream, DevBuf &mums, uint64_t *nmums,
                     uint64_t *sumlength, uint64_t carry = 0,
                     unsigned int valbits = 0, uint64_t seqoffset = 0)
{
  // VAL, valbits, seqoffset: see k_mum_writepacked
  // carry: as for mumuniqueinquery
  *nmums = 0;
  *sumlength = 0;
  if (ncand == 0)
  {
    return 0;
  }
  DevBuf k2, v2, dbright, keep, slots, temp, dcount, blocksum;
  const size_t nblocks = blocksfor(ncand);
  // the passes behind the sort by tiles (mum_filter.inc; VSA_FILTER_TILES=0:
  // the rocPRIM scans of round 2)
  const char *tilesenv = getenv("VSA_FILTER_TILES");
  const bool bytiles = !(tilesenv != nullptr && strcmp(tilesenv, "0") == 0);
  const uint64_t ntiles = (ncand + VSA_FT_TILE - 1) / VSA_FT_TILE;
  DevBuf tmax, tcarry, tcount, toff, tsum, tsumscan;
  if (bytiles &&
      (tmax.alloc((ntiles + 1) * 8) || tcarry.alloc((ntiles + 1) * 8) ||
       tcount.alloc((ntiles + 1) * 8) || toff.alloc((ntiles + 1) * 8) ||
       tsum.alloc((ntiles + 1) * 8) || tsumscan.alloc((ntiles + 1) * 8)))
  {
    return -100;
  }
  if (k2.alloc(ncand * 8) || v2.alloc(ncand * sizeof(VAL)) ||
      (!bytiles && dbright.alloc(ncand * 8)) ||
      keep.alloc(ntiles * VSA_FT_TILE) ||
      (!bytiles && slots.alloc(ncand * 4)) || dcount.alloc(24) ||
      blocksum.alloc(vsa_grid_blocks(nblocks) * 8) ||
      mums.alloc(ncand * sizeof(vsa_match)))
  {
    return -100;
  }
  // Sorted by dbstart alone where the runs of equal dbstarts are short (see
  // k_mum_keyflags_runs; decided afterwards from a flag that comes back with
  // the counts), by (dbstart, length down) otherwise.
  size_t tb = 0;
  uint64_t got[3] = {0, 0, 0};
  for (int pass = 0; pass < 2; pass++)
  {
    const bool byruns = pass == 0;
    const unsigned int firstbit = byruns ? lenbits : 0u;
    VSA_HIP(hipMemsetAsync(dcount.p, 0, 24, stream));
    tb = 0;
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, keys_in, k2.as<uint64_t>(), vals_in,
        v2.as<VAL>(), (size_t) ncand, firstbit, lenbits + dbbits, stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, keys_in, k2.as<uint64_t>(), vals_in,
        v2.as<VAL>(), (size_t) ncand, firstbit, lenbits + dbbits, stream));
    if (byruns && bytiles)
    {
      const dim3 tg = vsa_grid(ntiles);
      k_mumf_tilemax<<<tg, VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), ncand, lenbits, tmax.as<uint64_t>());
      k_mumf_scan<1><<<1, VSA_BLOCK, 0, stream>>>(
          tmax.as<uint64_t>(), ntiles, carry, tcarry.as<uint64_t>());
      k_mumf_flags<<<tg, VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), ncand, lenbits, tcarry.as<uint64_t>(),
          keep.as<uint8_t>(), tcount.as<uint64_t>(), tsum.as<uint64_t>(),
          dcount.as<unsigned int>() + 4);
      // (offsets of the tiles and, in a second workgroup, the sum of the
      // lengths)
      k_mumf_scan<0><<<2, VSA_BLOCK, 0, stream>>>(
          tcount.as<uint64_t>(), ntiles, 0, toff.as<uint64_t>(),
          tsum.as<uint64_t>(), tsumscan.as<uint64_t>());
      k_mumf_write<VAL><<<tg, VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), v2.as<VAL>(), keep.as<uint8_t>(), ncand,
          toff.as<uint64_t>(), lenbits, valbits, seqoffset,
          mums.as<vsa_match>());
      VSA_HIP(hipGetLastError());
      // number of MUMs, sum of their lengths, "a run was too long"
      const Fetch f[3] = {{toff.as<uint64_t>() + ntiles, 8},
                          {tsumscan.as<uint64_t>() + ntiles, 8},
                          {dcount.as<uint64_t>() + 2, 8}};
      if (fetchwords(stream, f, 3, got))
      {
        return -100;
      }
      if (got[2] == 0)
      {
        break;
      }
      continue;
    }
    if (dbright.p == nullptr &&
        (dbright.alloc(ncand * 8) || slots.alloc(ncand * 4)))
    {
      return -100;
    }
    // running maximum of the right ends, which are a function of the keys
    auto ends = rocprim::make_transform_iterator(k2.as<uint64_t>(),
                                                 KeyToRightEnd{lenbits});
    tb = 0;
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, ends, dbright.as<uint64_t>(),
                                    carry, (size_t) ncand,
                                    rocprim::maximum<uint64_t>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(temp.p, tb, ends, dbright.as<uint64_t>(),
                                    carry, (size_t) ncand,
                                    rocprim::maximum<uint64_t>(), stream));
    if (byruns)
    {
      k_mum_keyflags_runs<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), dbright.as<uint64_t>(), ncand, lenbits,
          keep.as<uint8_t>(), dcount.as<unsigned int>() + 4);
    } else
    {
      k_mum_keyflags<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), dbright.as<uint64_t>(), ncand, lenbits,
          keep.as<uint8_t>());
    }
    VSA_HIP(hipGetLastError());
    auto keepit =
        rocprim::make_transform_iterator(keep.as<uint8_t>(), KeepToU32());
    tb = 0;
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(temp.p, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    k_mum_writepacked<VAL><<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), v2.as<VAL>(), keep.as<uint8_t>(),
        slots.as<uint32_t>(), ncand, lenbits, valbits, seqoffset,
        mums.as<vsa_match>(), dcount.as<uint64_t>(),
        blocksum.as<unsigned long long>());
    VSA_HIP(hipGetLastError());
    tb = 0;
    VSA_HIP(rocprim::reduce(nullptr, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull, nblocks,
                            rocprim::plus<unsigned long long>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(temp.p, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull, nblocks,
                            rocprim::plus<unsigned long long>(), stream));
    const Fetch f[3] = {{dcount.p, 8}, {dcount.as<uint64_t>() + 1, 8},
                        {dcount.as<uint64_t>() + 2, 8}};
    if (fetchwords(stream, f, 3, got))
    {
      return -100;
    }
    if (!byruns || got[2] == 0)
    {
      break;
    }
  }
  *nmums = got[0];
  *sumlength = got[1];
  return 0;
}

template <typename IDX>
int run_query(const vsa_index *index, const vsa_queries *queries, bool domum,
              bool domumcand, uint32_t searchlength, vsa_result *res,
              bool ordered = true, uint32_t forcebits = 0)
{
  // forcebits != 0 (with domumcand, !ordered): the candidates stay pairs with
  // this many length bits (vsa_findmumcandidates_packed)
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  DevQueries qs = devqueries(queries);
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
    // base[q] = number of work-items in front of query q, from the lengths
    // on the device (a host loop and an upload of 8 bytes per query cost
    // more than the search for a batch of millions of reads)
    const uint64_t nqr = queries->nq;
    DevBuf btemp;
    size_t tb = 0;
    const uint64_t least = searchlength;
    auto items = rocprim::make_transform_iterator(
        rocprim::counting_iterator<uint64_t>(0),
        [len = qs.length, nqr, least] __device__(uint64_t q) -> uint64_t {
          if (q >= nqr)
          {
            return 0; // the entry behind the last query: the total
          }
          const uint64_t l = len[q];
          return l >= least ? l - least + 1 : 0;
        });
    if (base.alloc((nqr + 1) * 8))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, items, base.as<uint64_t>(),
                                    (uint64_t) 0, (size_t) (nqr + 1),
                                    rocprim::plus<uint64_t>(), stream));
    if (btemp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(btemp.p, tb, items, base.as<uint64_t>(),
                                    (uint64_t) 0, (size_t) (nqr + 1),
                                    rocprim::plus<uint64_t>(), stream));
    {
      const Fetch f = {base.as<uint64_t>() + nqr, 8};
      if (fetchwords(stream, &f, 1, &nitems))
      {
        return -100;
      }
    }
    dbase = base.as<uint64_t>();
  }
  res->stats.searches = nitems;
  if (nitems == 0)
  {
    return 0;
  }
  const uint32_t nshards = VSA_CURSOR_SHARDS;
  const bool deepok = ix.esa8 != nullptr && searchlength >= ix.D;
  tall.start();
  // MUM modes: first pass + work plan (mum_workplan.inc)
  DevBuf wcount, wtemp, wplan, wlist, wfirste, wfmlen, wfmdb, wboffset;
  uint64_t nfirstpass = 0; // candidates of the first pass (k_append_first)
  uint64_t plansearches = 0, nfirst = 0, mumsum = ~0ull;
  // -mum with the filter: candidates as (sort key, value) pairs, see
  // mumfilter_packed
  const bool keeppairs = domum && domumcand && !ordered && forcebits != 0;
  const unsigned int lenbits =
                         keeppairs ? forcebits : bitsfor(queries->maxlength),
                     dbbits = bitsfor(index->n);
  const bool packed = domum && (!domumcand || keeppairs) &&
                      lenbits + dbbits <= 64 &&
                      lenbits >= bitsfor(queries->maxlength) &&
                      queries->maxlength < 0xFFFFu &&
                      ((queries->nq + qs.seqoffset) >> 48) == 0;
  if (keeppairs && !packed)
  {
    VSA_ERROR("packed candidates: %u length bits do not fit this batch "
              "(longest query %lu, index %lu)", forcebits,
              (unsigned long) queries->maxlength, (unsigned long) index->n);
    return -2;
  }
  const uint32_t packbits = packed ? lenbits : 0;
  // 4-byte values where query number and offset fit (not for pairs that
  // travel to other ranks: those carry the global query number)
  const uint32_t valbits =
      (packed && !keeppairs && ((queries->nq << lenbits) >> 32) == 0)
          ? lenbits
          : 0;
  const size_t recsize = valbits != 0 ? 4 : (packed ? 8 : sizeof(vsa_match));
  bool fromplan = false, planemit = false;
  DevBuf pcursor, pdoff, psummary, prawout, prawkeys; // see PlanEmit
  uint64_t pcap = 0, nplan = 0;
  uint64_t nwork = nitems;
  Timer tfirst(stream); // the first pass kernel (k_mum_first) alone
  // ragged batches take the same route with per-query geometry
  const uint64_t maxoffsets =
      (queries->maxlength >= searchlength)
          ? queries->maxlength - searchlength + 1
          : 0;
  if (domum && queries->maxlength >= 255 && index->lcpquirk < 0)
  {
    uint8_t b = 0;
    if (index->n >= 2)
    {
      VSA_HIP(hipMemcpyAsync(&b, index->lcp + index->n - 1, 1,
                             hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipStreamSynchronize(stream));
    }
    index->lcpquirk = (b == 255) ? 1 : 0;
  }
  // The work reduction (see k_mum_first, k_mum_plan) rests on "a match that
  // is not unique is no candidate"; the reference's test for lcp >= 255
  // (fquery.c:352) breaks that rule in one situation, which one byte of
  // lcptab rules out (see vsa_index::lcpquirk).  A plan holds 16-bit offsets.
  // VSA_TUNE=2: no work reduction (every offset is searched by the list form
  // of the search kernel -- the cross-check of everything below).
  const bool reduce = domum && maxoffsets > 1 && maxoffsets < 0xFFFFu &&
                      queries->nq < 0xFFFFFFFFull && (index->tune & 2u) == 0 &&
                      !(queries->maxlength >= 255 && index->lcpquirk != 0);
  // packed batches (reads at two bits per symbol): first pass, plan and
  // search kernel read the rows; everything else takes the bytes, which are
  // made on the device once per batch
  const bool rows = queries->rows != nullptr && reduce && deepok &&
                    queries->roww <= 4;
  const char *dbgrows = getenv("VSA_DEBUG_ROWS");
  const int rowbits = dbgrows != nullptr ? atoi(dbgrows) : 7;
  if (queries->rows != nullptr && (!rows || rowbits != 7))
  {
    if (vsa_queries_bytes(queries, stream) != 0)
    {
      return -100;
    }
    qs = devqueries(queries);
  }
  // MEM (-l L): first pass, then the plan of mem_workplan.inc -- aligned
  // stretches answered from one bit per text position, the rest searched
  // (a packed batch has its bytes by now: MEM reads bytes)
  const bool memplan = !domum && deepok && searchlength <= 255 &&
                       maxoffsets > 1 &&
                       maxoffsets < 0xFFFFu && queries->nq < 0xFFFFFFFFull &&
                       (index->tune & 2u) == 0;
  if (memplan)
  {
    const uint64_t nq = queries->nq;
    // one bit per text position: does its suffix have a neighbour in the
    // suffix array with lcp >= L?  Made once per (index, L), kept.
    if (index->repbits == nullptr || index->repleast != searchlength)
    {
      if (index->repbits == nullptr)
      {
        VSA_HIP(vsa_hip_malloc((void **) &index->repbits,
                               ((index->n + 1) / 32 + 4) * 4));
      }
      VSA_HIP(hipMemsetAsync(index->repbits, 0, ((index->n + 1) / 32 + 4) * 4,
                             stream));
      k_repeat_bits<IDX><<<vsa_grid(blocksfor((index->n + 16) / 16)),
                           VSA_BLOCK, 0, stream>>>(
          ix.lcp, ix.suf, index->n, searchlength, index->repbits);
      VSA_HIP(hipGetLastError());
      index->repleast = searchlength;
    }
    if (wcount.alloc((nq + 1) * 4) || wfirste.alloc(nq * 4) ||
        wfmlen.alloc(nq * 4) || wfmdb.alloc(nq * 8) ||
        wplan.alloc(nq * sizeof(PlanRanges)))
    {
      return -100;
    }
    tfirst.start();
    {
      const bool staged = qs.dense != 0 && qs.uniformlen <= 128 &&
                          (qs.uniformlen & 3u) == 0;
      if (staged)
      {
        k_mum_first<IDX, true, true>
            <<<gridfor(nq), VSA_BLOCK, (size_t) VSA_BLOCK * qs.uniformlen,
               stream>>>(ix, qs, perquery, searchlength,
                         wcount.as<uint32_t>(), wfirste.as<uint32_t>(),
                         wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>());
      } else
      {
        k_mum_first<IDX, true><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
            ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
            wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
            wfmdb.as<uint64_t>());
      }
    }
    tfirst.stop();
    VSA_HIP(hipGetLastError());
    {
      // room for every answer of the workgroups that share a region (a read
      // answers at most one offset per round)
      const uint64_t nb = blocksfor(nq),
                     pershard = (nb + nshards - 1) / nshards;
      pcap = pershard * VSA_BLOCK * VSA_PLAN_ROUNDS;
      if (pcursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
          pdoff.alloc(nshards * 8) || psummary.alloc(4 * 8) ||
          prawout.alloc(nshards * pcap * recsize) ||
          prawkeys.alloc(nshards * pcap * 8))
      {
        return -100;
      }
      VSA_HIP(hipMemsetAsync(pcursor.p, 0,
                             (size_t) nshards * VSA_CURSOR_STRIDE * 8,
                             stream));
      PlanEmit em;
      em.base = dbase;
      em.perquery = perquery;
      em.out = prawout.as<vsa_match>();
      em.outkey = prawkeys.as<uint64_t>();
      em.shardcap = pcap;
      em.shardmask = nshards - 1;
      em.cursors = pcursor.as<unsigned long long>();
      em.packbits = 0;
      em.valbits = 0;
      k_mem_plan<IDX><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          ix, qs, perquery, searchlength, wfirste.as<uint32_t>(),
          wfmdb.as<uint64_t>(), index->repbits, wcount.as<uint32_t>(),
          wplan.as<PlanRanges>(), em);
      VSA_HIP(hipGetLastError());
      k_shard_summary<<<1, 1024, 0, stream>>>(
          pcursor.as<unsigned long long>(), nshards, pdoff.as<uint64_t>(),
          nullptr, nullptr, psummary.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    }
    planemit = true;
    fromplan = true;
    plansearches = nq; // (an upper bound of the plan's own locates per round)
  }
  if (reduce)
  {
    const uint64_t nq = queries->nq;
    if (wcount.alloc((nq + 1) * 4) || wfirste.alloc(nq * 4) ||
        wfmlen.alloc(nq * 4) || wfmdb.alloc(nq * 8) ||
        wplan.alloc(nq * sizeof(PlanRanges)) || wlist.alloc(nq * 4))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(wcount.as<uint32_t>() + nq, 0, 4, stream));
    tfirst.start();
    if (rows && (rowbits & 1))
    {
      k_mum_first<IDX, true, true, true>
          <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
              ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
              wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
              wfmdb.as<uint64_t>());
    } else if (deepok)
    {
      // reads of one length m (a multiple of 4, <= 128), back to back:
      // staged through LDS and packed
      const bool staged = qs.dense != 0 && qs.uniformlen <= 128 &&
                          (qs.uniformlen & 3u) == 0;
      if (staged)
      {
        k_mum_first<IDX, true, true>
            <<<gridfor(nq), VSA_BLOCK, (size_t) VSA_BLOCK * qs.uniformlen,
               stream>>>(ix, qs, perquery, searchlength,
                         wcount.as<uint32_t>(), wfirste.as<uint32_t>(),
                         wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>());
      } else
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
    tfirst.stop();
    VSA_HIP(hipGetLastError());
    // the reads the first pass has not finished, as a list (counts per
    // workgroup, a scan over the workgroups, an ordered fill); with it come
    // the places of the first pass's candidates
    uint64_t nlist = 0;
    size_t tb = 0;
    {
      // (queries < 2^32: the condition of this branch)
      const uint64_t nb = blocksfor(nq), nbr = vsa_grid_blocks(nb);
      DevBuf bcount;
      if (bcount.alloc((nbr + 1) * 8) || wboffset.alloc((nbr + 1) * 8))
      {
        return -100;
      }
      // (both halves of a count stay below 2^32: nq does)
      VSA_HIP(hipMemsetAsync(bcount.as<uint64_t>() + nb, 0, 8, stream));
      k_wanted_count<<<vsa_grid(nb), VSA_BLOCK, 0, stream>>>(
          wcount.as<uint32_t>(), nq, 0u, wfmlen.as<uint32_t>(),
          bcount.as<uint64_t>());
      VSA_HIP(hipGetLastError());
      VSA_HIP(rocprim::exclusive_scan(nullptr, tb, bcount.as<uint64_t>(),
                                      wboffset.as<uint64_t>(), (uint64_t) 0,
                                      (size_t) (nb + 1),
                                      rocprim::plus<uint64_t>(), stream));
      if (wtemp.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::exclusive_scan(wtemp.p, tb, bcount.as<uint64_t>(),
                                      wboffset.as<uint64_t>(), (uint64_t) 0,
                                      (size_t) (nb + 1),
                                      rocprim::plus<uint64_t>(), stream));
      k_wanted_fill<<<vsa_grid(nb), VSA_BLOCK, 0, stream>>>(
          wcount.as<uint32_t>(), nq, 0u, wboffset.as<uint64_t>(),
          wlist.as<uint32_t>());
      VSA_HIP(hipGetLastError());
      const Fetch f = {wboffset.as<uint64_t>() + nb, 8};
      uint64_t both = 0;
      if (fetchwords(stream, &f, 1, &both))
      {
        return -100;
      }
      nlist = both & 0xFFFFFFFFull;
      nfirstpass = both >> 32;
    }
    if (nlist > 0)
    {
      // on the deep tables the plan answers the offsets it locates itself
      // (PlanEmit)
      planemit = deepok;
      if (planemit)
      {
        // room for every search A of the workgroups that share a region
        const uint64_t nb = blocksfor(nlist),
                       pershard = (nb + nshards - 1) / nshards;
        pcap = pershard * VSA_BLOCK * (VSA_PLAN_ROUNDS - 1);
        if (pcursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
            pdoff.alloc(nshards * 8) || psummary.alloc(4 * 8) ||
            prawout.alloc(nshards * pcap * recsize) ||
            prawkeys.alloc(nshards * pcap * 8))
        {
          return -100;
        }
        VSA_HIP(hipMemsetAsync(pcursor.p, 0,
                               (size_t) nshards * VSA_CURSOR_STRIDE * 8,
                               stream));
        PlanEmit em;
        em.base = dbase;
        em.perquery = perquery;
        em.out = prawout.as<vsa_match>();
        em.outkey = prawkeys.as<uint64_t>();
        em.shardcap = pcap;
        em.shardmask = nshards - 1;
        em.cursors = pcursor.as<unsigned long long>();
        em.packbits = packbits;
        em.valbits = valbits;
        if (rows && (rowbits & 2))
        {
          k_mum_plan<IDX, true, true, true>
              <<<gridfor(nlist), VSA_BLOCK, 0, stream>>>(
                  ix, qs, wlist.as<uint32_t>(), nlist, searchlength,
                  wfirste.as<uint32_t>(), wcount.as<uint32_t>(),
                  wplan.as<PlanRanges>(), em);
        } else
        {
          k_mum_plan<IDX, true, true>
              <<<gridfor(nlist), VSA_BLOCK, 0, stream>>>(
                  ix, qs, wlist.as<uint32_t>(), nlist, searchlength,
                  wfirste.as<uint32_t>(), wcount.as<uint32_t>(),
                  wplan.as<PlanRanges>(), em);
        }
        k_shard_summary<<<1, 1024, 0, stream>>>(
            pcursor.as<unsigned long long>(), nshards, pdoff.as<uint64_t>(),
            nullptr, nullptr, psummary.as<uint64_t>());
      } else
      {
        k_mum_plan<IDX, false><<<gridfor(nlist), VSA_BLOCK, 0, stream>>>(
            ix, qs, wlist.as<uint32_t>(), nlist, searchlength,
            wfirste.as<uint32_t>(), wcount.as<uint32_t>(),
            wplan.as<PlanRanges>());
      }
      VSA_HIP(hipGetLastError());
    }
    plansearches = 2 * nlist;
    if (const char *pf = getenv("VSA_DEBUG_PLANFILE"))
    {
      // the plans of the first 65 536 queries as they stand when the search
      // kernel starts -- per query its count and VSA_PLAN_RANGES ranges
      // (first | length << 16), 32-bit words -- for bench.py, which prices
      // the kernel on exactly the searches it runs
      const uint64_t k = std::min<uint64_t>(nq, 65536);
      std::vector<uint32_t> hc(k), hp(k * VSA_PLAN_RANGES);
      VSA_HIP(hipMemcpyAsync(hc.data(), wcount.p, k * 4,
                             hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipMemcpyAsync(hp.data(), wplan.p, k * sizeof(PlanRanges),
                             hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipStreamSynchronize(stream));
      if (FILE *f = fopen(pf, "wb"))
      {
        for (uint64_t q = 0; q < k; q++)
        {
          (void) fwrite(&hc[q], 4, 1, f);
          (void) fwrite(&hp[q * VSA_PLAN_RANGES], 4, VSA_PLAN_RANGES, f);
        }
        fclose(f);
      }
    }
    fromplan = true;
  }
  DevBuf doff, rawout, rawkeys, summary, blocksum, rtemp;
  const uint64_t nplanblocks = (queries->nq + 255) / 256;
  uint64_t plannedwork = 0;
  size_t rbytes = 0;
  if (fromplan)
  {
    // the work-items of the planned search: summed per workgroup by the
    // kernel, reduced behind it into the fifth word of the shard summary
    if (blocksum.alloc((vsa_grid_blocks(nplanblocks) + 1) * 8))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(nullptr, rbytes,
                            blocksum.as<unsigned long long>(),
                            (unsigned long long *) nullptr, 0ull,
                            (size_t) nplanblocks,
                            rocprim::plus<unsigned long long>(), stream));
    if (rtemp.alloc(rbytes))
    {
      return -100;
    }
  }
  if (cursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
      doff.alloc(nshards * 8) || summary.alloc(5 * 8))
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
    if (rawout.alloc(nshards * shardcap * recsize) ||
        rawkeys.alloc(nshards * shardcap * 8))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(cursor.p, 0,
                           (size_t) nshards * VSA_CURSOR_STRIDE * 8, stream));
    tsearch.start();
#define VSA_LAUNCH_QUERY(MUMFLAG, KEYFLAG)                                     \
  k_query_search<IDX, MUMFLAG, KEYFLAG, 256>                                  \
      <<<vsa_grid((nwork + 255) / 256), 256, 0, stream>>>(                    \
          ix, qs, dbase, perquery, nwork, searchlength,                       \
          rawout.as<vsa_match>(), rawkeys.as<uint64_t>(), shardcap,           \
          nshards - 1, cursor.as<unsigned long long>(), packbits, valbits)
    if (fromplan && memplan)
    {
      k_query_search_planned<IDX, 256, true, false, false>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (fromplan && rows && (rowbits & 4))
    {
      k_query_search_planned<IDX, 256, true, true>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (fromplan && deepok)
    {
      k_query_search_planned<IDX, 256, true>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (fromplan)
    {
      k_query_search_planned<IDX, 256, false>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (nwork > 0)
    {
      // every (query, offset) pair: MEM, and MUM batches without a plan
      // (deep locate needs the deep prefix to fit into every search)
      if (domum && deepok)
      {
        VSA_LAUNCH_QUERY(true, true);
      } else if (domum)
      {
        VSA_LAUNCH_QUERY(true, false);
      } else if (deepok)
      {
        VSA_LAUNCH_QUERY(false, true);
      } else
      {
        VSA_LAUNCH_QUERY(false, false);
      }
    }
#undef VSA_LAUNCH_QUERY
    tsearch.stop();
    VSA_HIP(hipGetLastError());
    if (fromplan)
    {
      VSA_HIP(rocprim::reduce(rtemp.p, rbytes,
                              blocksum.as<unsigned long long>(),
                              summary.as<unsigned long long>() + 4, 0ull,
                              (size_t) nplanblocks,
                              rocprim::plus<unsigned long long>(), stream));
    }
    // where each region goes in the dense list, how much there is
    k_shard_summary<<<1, 1024, 0, stream>>>(
        cursor.as<unsigned long long>(), nshards, doff.as<uint64_t>(),
        nullptr, nullptr,
        summary.as<uint64_t>());
    VSA_HIP(hipGetLastError());
    {
      const Fetch f[6] = {{summary.as<uint64_t>(), 8},
                          {summary.as<uint64_t>() + 1, 8},
                          {summary.as<uint64_t>() + 2, 8},
                          {summary.as<uint64_t>() + 3, 8},
                          {summary.as<uint64_t>() + 4, 8},
                          {planemit ? psummary.p : summary.p, 8}};
      uint64_t got[6];
      if (fetchwords(stream, f, 6, got))
      {
        return -100;
      }
      nplan = planemit ? got[5] : 0;
      needed = got[0];
      maxshard = got[1];
      nfirst = fromplan ? nfirstpass : 0;
      plannedwork = fromplan ? got[4] : 0;
    }
    searchms += tsearch.ms();
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
  if (needed + nplan + nfirst > 0)
  {
    if (out.alloc((needed + nplan + nfirst) * recsize) ||
        keys.alloc((needed + nplan + nfirst) * 8))
    {
      return -100;
    }
    if (nplan > 0 && valbits != 0)
    {
      k_compact_shards<uint32_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          prawout.as<uint32_t>(), prawkeys.as<uint64_t>(), pcap,
          pcursor.as<unsigned long long>(), pdoff.as<uint64_t>(),
          out.as<uint32_t>() + needed, keys.as<uint64_t>() + needed);
      VSA_HIP(hipGetLastError());
    } else if (nplan > 0 && packed)
    {
      k_compact_shards<uint64_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          prawout.as<uint64_t>(), prawkeys.as<uint64_t>(), pcap,
          pcursor.as<unsigned long long>(), pdoff.as<uint64_t>(),
          out.as<uint64_t>() + needed, keys.as<uint64_t>() + needed);
      VSA_HIP(hipGetLastError());
    } else if (nplan > 0)
    {
      k_compact_shards<vsa_match><<<nshards, VSA_BLOCK, 0, stream>>>(
          prawout.as<vsa_match>(), prawkeys.as<uint64_t>(), pcap,
          pcursor.as<unsigned long long>(), pdoff.as<uint64_t>(),
          out.as<vsa_match>() + needed, keys.as<uint64_t>() + needed);
      VSA_HIP(hipGetLastError());
    }
    if (needed > 0 && valbits != 0)
    {
      k_compact_shards<uint32_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<uint32_t>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<uint32_t>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    } else if (needed > 0 && packed)
    {
      k_compact_shards<uint64_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<uint64_t>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<uint64_t>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    } else if (needed > 0)
    {
      k_compact_shards<vsa_match><<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<vsa_match>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<vsa_match>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    }
    if (nfirst > 0)
    {
      k_append_first<<<gridfor(queries->nq), VSA_BLOCK, 0, stream>>>(
          wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>(),
          wboffset.as<uint64_t>(), queries->nq, perquery, dbase, qs.seqoffset,
          needed + nplan,
          out.as<vsa_match>(), keys.as<uint64_t>(), packbits, valbits);
      VSA_HIP(hipGetLastError());
    }
    needed += nplan + nfirst;
  }
  res->stats.candidates = domum ? needed : 0;
  if (keeppairs)
  {
    res->count = needed;
    res->packbits = lenbits;
    if (needed > 0)
    {
      res->matches = (vsa_match *) keys.release();
      res->packvals = (uint64_t *) out.release();
    }
  } else if (domum && !domumcand)
  {
    DevBuf mums;
    uint64_t nm = 0;
    if (valbits != 0)
    {
      if (mumfilter_packed<uint32_t>(keys.as<const uint64_t>(),
                                     out.as<const uint32_t>(), needed, lenbits, dbbits,
                                     stream, mums, &nm, &mumsum, 0, valbits,
                                     qs.seqoffset))
      {
        return -100;
      }
    } else if (packed)
    {
      if (mumfilter_packed(keys.as<const uint64_t>(),
                           out.as<const uint64_t>(), needed, lenbits, dbbits, stream, mums,
                           &nm, &mumsum))
      {
        return -100;
      }
    } else if (mumuniqueinquery(out, needed, stream, mums, &nm, 0, nullptr,
                                index->n, queries->maxlength, &mumsum))
    {
      return -100;
    }
    res->count = nm;
    res->matches = (vsa_match *) mums.release();
  } else if (needed > 0 && !ordered && domum)
  {
    // candidates for a filter that sorts them anyway (multi-GPU -mum): as
    // they lie
    res->count = needed;
    res->matches = (vsa_match *) out.release();
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
  res->stats.anchor_ms = 0; // (the anchor pass of round 1 is gone)
  res->stats.first_kernel_ms = tfirst.ms();
  if (fromplan)
  {
    nwork = plannedwork;
    res->stats.searches = nwork + queries->nq + plansearches;
  }
  res->stats.kernel_searches = nwork;
  res->stats.total_device_ms = tall.ms();
  if (mumsum != ~0ull)
  {
    res->stats.sumlength = mumsum; // the filter summed the lengths already
    return 0;
  }
  if (keeppairs)
  {
    res->stats.sumlength = 0; // of candidates: nobody asks
    return 0;
  }
  return sumlengths(res->matches, res->count, stream, &res->stats.sumlength);
}

// ---- K3 pipeline ----

// workgroups of the streaming pass: 4 per CU = all the wavefronts that fit
// (86 registers, four per SIMD), every one walking its tiles grid-stride
#define VSA_PEAK_BLOCKS 1024

// [first, last) = the values of the reference's loop variable i
// (fmumself.c:33: i = 2 .. n-1) this call covers
template <typename IDX>
int run_selfmum(const vsa_index *index, uint64_t searchlength,
                uint64_t first, uint64_t last, vsa_result *res)
{
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  const uint64_t n = index->n;
  const uint32_t nshards = VSA_CURSOR_SHARDS;
  const uint64_t pieces = 4, tilesize = 64 * pieces * 16;
  // centres j = i - 1
  const uint64_t jlo = std::max<uint64_t>(first, 2) - 1,
                 jhi = std::max<uint64_t>(std::min<uint64_t>(last, n), 2) - 1;
  const uint64_t tile0 = jlo / tilesize,
                 ntiles = jhi > jlo ? (jhi + tilesize - 1) / tilesize : tile0,
                 wavesperblock = VSA_BLOCK / 64;
  const uint64_t nblocks = std::max<uint64_t>(
      1, std::min<uint64_t>((ntiles - tile0 + wavesperblock - 1) /
                                wavesperblock,
                            (uint64_t) VSA_PEAK_BLOCKS));
  const uint32_t slmin = (uint32_t) (searchlength < 255 ? searchlength : 255);
  DevBuf cursor, doff, rawpos, peaks, sorted, temp, cand, keep, dcount, mums;
  uint64_t shardcap =
               std::max<uint64_t>((jhi - std::min(jlo, jhi)) / 64 / nshards +
                                      1024,
                                  4096),
           needed = 0, maxshard = 0;
  double searchms = 0;

  res->stats.searches = jhi > jlo ? jhi - jlo : 0;
  DevBuf summary;
  if (cursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
      doff.alloc(nshards * 8) || dcount.alloc(8) || summary.alloc(4 * 8))
  {
    return -100;
  }
  tall.start();
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (rawpos.alloc(nshards * shardcap * sizeof(IDX)))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(cursor.p, 0,
                           (size_t) nshards * VSA_CURSOR_STRIDE * 8, stream));
    tsearch.start();
    k_selfmum_peaks<true, IDX><<<(unsigned int) nblocks, VSA_BLOCK, 0,
                                 stream>>>(
        ix.lcp, ix.bwt, n, slmin, rawpos.as<IDX>(), shardcap, nshards - 1,
        cursor.as<unsigned long long>(), tile0, ntiles, jlo, jhi);
    tsearch.stop();
    VSA_HIP(hipGetLastError());
    k_shard_summary<<<1, 1024, 0, stream>>>(
        cursor.as<unsigned long long>(), nshards, doff.as<uint64_t>(),
        nullptr, nullptr, summary.as<uint64_t>());
    VSA_HIP(hipGetLastError());
    {
      const Fetch f[2] = {{summary.as<uint64_t>(), 8},
                          {summary.as<uint64_t>() + 1, 8}};
      uint64_t got[2];
      if (fetchwords(stream, f, 2, got))
      {
        return -100;
      }
      needed = got[0];
      maxshard = got[1];
    }
    searchms = tsearch.ms(); // the streaming pass (of the last attempt)
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
    if (peaks.alloc(needed * sizeof(IDX)) ||
        sorted.alloc(needed * sizeof(IDX)) ||
        cand.alloc(needed * sizeof(vsa_match)) || keep.alloc(needed) ||
        mums.alloc(needed * sizeof(vsa_match)))
    {
      return -100;
    }
    k_gather_shards<IDX><<<nshards, VSA_BLOCK, 0, stream>>>(
        rawpos.as<IDX>(), shardcap, cursor.as<unsigned long long>(),
        doff.as<uint64_t>(), peaks.as<IDX>());
    VSA_HIP(hipGetLastError());
    // the reference reports in suffix-array order
    size_t tb = 0;
    VSA_HIP(rocprim::radix_sort_keys(nullptr, tb, peaks.as<IDX>(),
                                     sorted.as<IDX>(), (size_t) needed,
                                     0u, bitsfor(n), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_keys(temp.p, tb, peaks.as<IDX>(),
                                     sorted.as<IDX>(), (size_t) needed,
                                     0u, bitsfor(n), stream));
    k_selfmum_emit<IDX><<<gridfor(needed), VSA_BLOCK, 0, stream>>>(
        ix, sorted.as<IDX>(), needed, searchlength,
        index->querysepposition, cand.as<vsa_match>(), keep.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    if (compact_matches(cand.as<vsa_match>(), keep.as<uint8_t>(), needed,
                        mums.as<vsa_match>(), dcount.as<uint64_t>(), stream))
    {
      return -100;
    }
    {
      const Fetch f = {dcount.p, 8};
      if (fetchwords(stream, &f, 1, &nm))
      {
        return -100;
      }
    }
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
#include "approx_tree.inc"
#include "selfmatch_search.inc"

} // namespace

// ---------------------------------------------------------------------------
// the keyed search array (see DevIndex::esa8)
// ---------------------------------------------------------------------------

template <typename IDX>
__global__ void __launch_bounds__(VSA_BLOCK)
k_make_esa8(const uint8_t *__restrict__ tis, const IDX *__restrict__ suf,
            const uint8_t *__restrict__ lcp, uint64_t count, uint32_t D,
            uint64_t *__restrict__ esa8)
{
  const uint64_t j = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (j >= count)
  {
    return;
  }
  const uint64_t s = suf[j];
  const uint8_t *t = tis + s + D; // padded with 0xFF behind n
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
  // front pad of the text = separator: suffix 0 has nothing in front
  const uint8_t l = tis[(int64_t) s - 1];
  const uint64_t left = VSA_ISSPECIAL(l) ? VSA_LEFTSPECIAL
                                         : ((uint64_t) (l & 3) << VSA_LEFTSHIFT);
  // (of a wide suf only the low half: vsa_entrystart reads suf itself then)
  esa8[j] = (s & 0xFFFFFFFFull) | ((uint64_t) lcp[j] << 32) |
            (key << VSA_KEYSHIFT) | flag | left;
}

// tis2 / spec64 / firstspecial (see DevIndex): one work-item packs a block of
// 64 text positions into 16 bytes; the wavefront's ballot is 8 bytes of the
// block bitmap.  Positions >= n count as special.
__global__ void __launch_bounds__(VSA_BLOCK)
k_pack_text(const uint8_t *__restrict__ tis, uint64_t n, uint64_t nblocks,
            uint8_t *__restrict__ tis2, uint8_t *__restrict__ spec64,
            unsigned long long *__restrict__ firstspecial)
{
  const uint64_t b = vsa_bid() * VSA_BLOCK + threadIdx.x;
  bool special = false;
  if (b < nblocks)
  {
    const uint8_t *p = tis + 64 * b; // 0xFF behind position n
    uint64_t out[2] = {0, 0}, firstbad = ~0ull;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const vsa_u128 v = vsa_load16(p + 16 * k);
      const uint64_t notdna = 0xFCFCFCFCFCFCFCFCull;
      const uint64_t s0 = v.lo & notdna, s1 = v.hi & notdna;
      if (firstbad == ~0ull && (s0 | s1) != 0)
      {
        firstbad = 64 * b + 16 * k +
                   (s0 != 0 ? ((uint64_t) __builtin_ctzll(s0) >> 3)
                            : 8 + ((uint64_t) __builtin_ctzll(s1) >> 3));
      }
      const uint64_t packed = vsa_pack16(v.lo, v.hi); // 16 symbols, 32 bits
      out[k >> 1] |= packed << (32 * (1 - (k & 1)));
    }
    // first symbol in the top bits of the first byte
    out[0] = __builtin_bswap64(out[0]);
    out[1] = __builtin_bswap64(out[1]);
    reinterpret_cast<uint64_t *>(tis2)[2 * b] = out[0];
    reinterpret_cast<uint64_t *>(tis2)[2 * b + 1] = out[1];
    special = firstbad != ~0ull || 64 * b + 64 > n;
    if (firstbad != ~0ull)
    {
      atomicMin(firstspecial, (unsigned long long) firstbad);
    }
  }
  const uint64_t mask = __ballot(special);
  if ((threadIdx.x & 63) == 0 && b < nblocks)
  {
    reinterpret_cast<uint64_t *>(spec64)[b >> 6] = mask;
  }
}

// slot[code] = (bck2 pair, the first W-1 entries of the bucket), W = 2 or 4
// words; entries the bucket does not have are 0 (they stand for the entry
// behind the bucket, whose lcp byte is below D anyway)
// Wide tables: word 0 in the form of vsa_slotbounds (left | count << 40);
// *toobig is set when a bucket's count does not fit.
template <int W, typename IDX>
__global__ void __launch_bounds__(VSA_BLOCK)
k_make_slots(const IDX *__restrict__ bck2, const uint64_t *__restrict__ esa8,
             uint64_t ncodes, uint64_t *__restrict__ slot,
             unsigned int *__restrict__ toobig)
{
  for (uint64_t c = vsa_bid() * VSA_BLOCK + threadIdx.x;
       c < ncodes; c += vsa_nblocks() * VSA_BLOCK)
  {
    const IDX left = bck2[2 * c], mid = bck2[2 * c + 1];
    if constexpr (sizeof(IDX) == 4)
    {
      slot[W * c] = (uint64_t) left | ((uint64_t) mid << 32);
    } else
    {
      const uint64_t cnt = mid > left ? mid - left : 0;
      if (cnt >> (64 - VSA_WIDE_LEFTBITS) != 0)
      {
        *toobig = 1;
      }
      slot[W * c] = left | (cnt << VSA_WIDE_LEFTBITS);
    }
#pragma unroll
    for (int k = 0; k + 1 < W; k++)
    {
      slot[W * c + 1 + k] = (mid > left + k) ? esa8[left + k] : 0;
    }
  }
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
  if (ix->slot16 != nullptr)
  {
    (void) hipFree(ix->slot16);
    ix->slot16 = nullptr;
  }
  if (ix->tis2 != nullptr)
  {
    (void) hipFree(ix->tis2);
    (void) hipFree(ix->spec64);
    ix->tis2 = ix->spec64 = nullptr;
  }
  const bool wide = ix->isize != 4;
  if (ix->numofchars != 4 || ix->bck == nullptr ||
      (wide && ((ix->n + 1) >> VSA_WIDE_LEFTBITS) != 0) ||
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
  } else
  {
    // one symbol less where the device has not room for the slot table, the
    // bucket bounds it is made from, the keyed array and a tenth of the
    // device for the searches themselves: half the index for a fifth more
    // time per batch (profiles/r03/footprint_deep_prefix.txt)
    size_t freeb = 0, totalb = 0;
    (void) hipStreamSynchronize(ix->stream);
    vsa_dev_trim(); // (what the builder's temporaries held counts as free)
    while (D > ix->pl && D > 12 &&
           hipMemGetInfo(&freeb, &totalb) == hipSuccess)
    {
      const uint64_t codes = 1ull << (2 * D),
                     need = 16 * codes + 2 * codes * ix->isize +
                            8 * (ix->n + 1) + ix->n / 4 + totalb / 10;
      if (need <= freeb)
      {
        break;
      }
      D--;
    }
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
  // The slot table is the one every search starts in, at a random place: it is
  // allocated FIRST, with the temporaries of the builder handed back to the
  // driver.  Placed last, between what the builder had left, the 68.7 GB of a
  // 3 Gbp index were mapped in small pages and a random read of it cost a read
  // of the page table on top (profiles/r03/table_read_probe.txt).
  VSA_HIP(hipStreamSynchronize(ix->stream));
  vsa_dev_trim();
  if (vsa_hip_malloc((void **) &ix->slot16, 2 * ncodes * 8 + 32) != hipSuccess)
  {
    // no room for it (VSA_DEEP_PREFIX asked for more than fits): this index
    // is searched the reference's way
    (void) hipGetLastError();
    ix->slot16 = nullptr;
    ix->D = 0;
    return 0;
  }
  VSA_HIP(vsa_hip_malloc((void **) &ix->bck2, 2 * ncodes * ix->isize + 16));
  VSA_HIP(vsa_hip_malloc((void **) &ix->esa8, count * 8 + 64));
  ix->device_bytes += count * 8 + 2 * ncodes * ix->isize;
  if (wide ? vsa_build_bucket_table(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                                    (const uint64_t *) ix->suf, D, 4,
                                    (uint64_t *) ix->bck2, ix->stream)
           : vsa_build_bucket_table(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                                    (const uint32_t *) ix->suf, D, 4,
                                    ix->bck2, ix->stream))
  {
    return -100;
  }
  if (wide)
  {
    k_make_esa8<uint64_t><<<gridfor(count), VSA_BLOCK, 0, ix->stream>>>(
        ix->tis_alloc + VSA_TIS_FRONTPAD, (const uint64_t *) ix->suf, ix->lcp,
        count, D, ix->esa8);
  } else
  {
    k_make_esa8<uint32_t><<<gridfor(count), VSA_BLOCK, 0, ix->stream>>>(
        ix->tis_alloc + VSA_TIS_FRONTPAD, (const uint32_t *) ix->suf, ix->lcp,
        count, D, ix->esa8);
  }
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipStreamSynchronize(ix->stream));
  // the 2-bit text for long comparisons
  {
    // blocks 0 .. n >> 6: a comparison ends at position n at the latest (the
    // last block reads into the 0xFF padding behind the text, not beyond it)
    static_assert(VSA_TIS_BACKPAD >= 64, "text pad too small for k_pack_text");
    const uint64_t nblocks = (ix->n >> 6) + 1,
                   nwaves = (nblocks + 63) / 64;
    unsigned long long *dfirst = nullptr, hfirst = ix->n;
    VSA_HIP(vsa_hip_malloc((void **) &ix->tis2, nblocks * 16 + 64));
    VSA_HIP(vsa_hip_malloc((void **) &ix->spec64, nwaves * 8 + 64));
    VSA_HIP(vsa_hip_malloc((void **) &dfirst, 8));
    VSA_HIP(hipMemsetAsync(ix->tis2 + nblocks * 16, 0, 64, ix->stream));
    VSA_HIP(hipMemsetAsync(ix->spec64 + nwaves * 8, 0xFF, 64, ix->stream));
    VSA_HIP(hipMemcpyAsync(dfirst, &hfirst, 8, hipMemcpyHostToDevice,
                           ix->stream));
    k_pack_text<<<vsa_grid(nwaves * 64 / VSA_BLOCK + 1), VSA_BLOCK, 0,
                  ix->stream>>>(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                                nblocks, ix->tis2, ix->spec64, dfirst);
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipMemcpyAsync(&hfirst, dfirst, 8, hipMemcpyDeviceToHost,
                           ix->stream));
    VSA_HIP(hipStreamSynchronize(ix->stream));
    (void) hipFree(dfirst);
    ix->firstspecial = hfirst < ix->n ? hfirst : ix->n;
    ix->device_bytes += nblocks * 16 + nwaves * 8;
  }
  // the fused table takes the place of bck2: 16 bytes per deep prefix (bounds
  // + the first entry: 69 % of the non-empty buckets of a random text are
  // answered by one access; 68.7 GB at 3 Gbp).  (32-byte slots with three
  // entries were measured in round 2, profiles/r02/slot32_ab.txt, and read
  // again in round 4, profiles/r04/README.md: fewer HBM lines, but a second
  // load instruction per lane.)
  unsigned int *dtoobig = nullptr, htoobig = 0;
  VSA_HIP(vsa_hip_malloc((void **) &dtoobig, 4));
  VSA_HIP(hipMemsetAsync(dtoobig, 0, 4, ix->stream));
  {
    const unsigned int grid = (unsigned int) std::min<uint64_t>(
        (ncodes + VSA_BLOCK - 1) / VSA_BLOCK, 1u << 20);
    if (wide)
    {
      k_make_slots<2, uint64_t><<<grid, VSA_BLOCK, 0, ix->stream>>>(
          (const uint64_t *) ix->bck2, ix->esa8, ncodes, ix->slot16, dtoobig);
    } else
    {
      k_make_slots<2, uint32_t><<<grid, VSA_BLOCK, 0, ix->stream>>>(
          ix->bck2, ix->esa8, ncodes, ix->slot16, dtoobig);
    }
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipMemcpyAsync(&htoobig, dtoobig, 4, hipMemcpyDeviceToHost,
                           ix->stream));
    VSA_HIP(hipStreamSynchronize(ix->stream));
    (void) hipFree(ix->bck2);
    ix->bck2 = nullptr;
    ix->slotwords = 2;
    ix->device_bytes += 2 * ncodes * 8 - 2 * ncodes * ix->isize;
  }
  (void) hipFree(dtoobig);
  if (wide && htoobig != 0)
  {
    // a deep bucket with 2^24 suffixes or more: this index is searched the
    // reference's way
    const uint64_t nblocks = (ix->n >> 6) + 1, nwaves = (nblocks + 63) / 64;
    ix->device_bytes -= count * 8;
    ix->device_bytes -= (uint64_t) ix->slotwords * ncodes * 8;
    if (ix->tis2 != nullptr)
    {
      ix->device_bytes -= nblocks * 16 + nwaves * 8;
    }
    (void) hipFree(ix->esa8);
    (void) hipFree(ix->bck2);
    (void) hipFree(ix->slot16);
    (void) hipFree(ix->tis2);
    (void) hipFree(ix->spec64);
    ix->esa8 = ix->slot16 = nullptr;
    ix->bck2 = nullptr;
    ix->tis2 = ix->spec64 = nullptr;
    ix->D = 0;
  }
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

extern "C" int vsa_findmumcandidates(const vsa_index *index,
                                     const vsa_queries *queries,
                                     uint64_t searchlength, int ordered,
                                     vsa_result **result)
{
  if (ordered)
  {
    return vsa_findquerymatches(index, queries, 1, 1, searchlength, result);
  }
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmumcandidates: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (searchlength < index->pl || searchlength > 0xFFFFFFF0ull)
  {
    // Vmengine/fquery.c:440-446
    VSA_ERROR("searchlength=%lu must be >= %lu=prefixlen",
              (unsigned long) searchlength, (unsigned long) index->pl);
    return -2;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc =
      (index->isize == 4)
          ? run_query<uint32_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false)
          : run_query<uint64_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findmumcandidates_packed(const vsa_index *index,
                                            const vsa_queries *queries,
                                            uint64_t searchlength,
                                            uint32_t lengthbits,
                                            vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmumcandidates_packed: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (searchlength < index->pl || searchlength > 0xFFFFFFF0ull)
  {
    // Vmengine/fquery.c:440-446
    VSA_ERROR("searchlength=%lu must be >= %lu=prefixlen",
              (unsigned long) searchlength, (unsigned long) index->pl);
    return -2;
  }
  if (lengthbits == 0)
  {
    lengthbits = bitsfor(queries->maxlength);
  }
  if (lengthbits > 16)
  {
    VSA_ERROR("packed candidates: at most 16 length bits");
    return -2;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc =
      (index->isize == 4)
          ? run_query<uint32_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false,
                                lengthbits)
          : run_query<uint64_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false,
                                lengthbits);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

// records of a result by the range of the index their dbstart falls into:
// part p = floor(dbstart * nparts / (totallength + 1)).  A counting sort in
// two passes over the records (the order inside a part is free): per
// workgroup and part a count (and the largest right end), one exclusive scan
// over the counts laid out part-major = the place of every (part, workgroup)
// in the output, then every workgroup puts its records there.  No global
// atomics: a cursor word per part would take one returning atomic per
// wavefront, which is slower than the whole rest (measured).
#define VSA_PART_MAX 256

// PACKBITS view of the input: records (m) or pairs (key[], val[])
struct PartInput
{
  const vsa_match *m;
  const uint64_t *key, *val;
  uint32_t stride; // 1: keys and values in arrays of their own; 2: in rows
  uint32_t packbits;
  // the part written behind all others (a rank's own: it does not travel),
  // its place among the parts and the number of parts; own = nparts: none
  uint32_t own, nparts;
  // part -> its place in the output
  __device__ __forceinline__ uint32_t place(uint32_t p) const
  {
    return p < own ? p : (p == own ? nparts - 1 : p - 1);
  }
};

__device__ __forceinline__ void part_read(const PartInput &in, uint64_t t,
                                          uint64_t &dbstart, uint64_t &length)
{
  if (in.packbits != 0)
  {
    const uint64_t k = in.key[t * in.stride], mask = (1ull << in.packbits) - 1;
    dbstart = k >> in.packbits;
    length = mask - (k & mask);
  } else
  {
    dbstart = in.m[t].dbstart;
    length = in.m[t].length;
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_partition_count(const PartInput in, uint64_t n,
                  uint32_t nparts, uint64_t totallength, uint64_t nblocks,
                  uint32_t *__restrict__ blockhist,
                  unsigned long long *__restrict__ blocktop)
{
  __shared__ unsigned int hist[VSA_PART_MAX];
  __shared__ unsigned long long top[VSA_PART_MAX];
  if (vsa_bid() >= nblocks) // surplus block of a folded grid
  {
    return;
  }
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
  for (uint32_t p = threadIdx.x; p < nparts; p += VSA_BLOCK)
  {
    hist[p] = 0;
    top[p] = 0;
  }
  __syncthreads();
  if (t < n)
  {
    uint64_t dbstart, length;
    part_read(in, t, dbstart, length);
    const uint32_t p =
        in.place((uint32_t) ((dbstart * nparts) / (totallength + 1)));
    atomicAdd(&hist[p], 1u);
    // right end of the match in the index (cleanMUMcand.c: dbright)
    atomicMax(&top[p], (unsigned long long) (dbstart + length - 1));
  }
  __syncthreads();
  for (uint32_t p = threadIdx.x; p < nparts; p += VSA_BLOCK)
  {
    blockhist[(uint64_t) p * nblocks + vsa_bid()] = hist[p];
    blocktop[(uint64_t) p * nblocks + vsa_bid()] = top[p];
  }
}

// per part: where it starts in the output and its largest right end
// (1024 lanes: one workgroup per part walks all the blocks' maxima)
__global__ void __launch_bounds__(1024)
k_partition_summary(const uint64_t *__restrict__ offsets,
                    const unsigned long long *__restrict__ blocktop,
                    uint32_t nparts, uint64_t nblocks,
                    uint64_t *__restrict__ partstart,
                    unsigned long long *__restrict__ parttop)
{
  __shared__ unsigned long long red[1024];
  const uint32_t p = vsa_bid();
  unsigned long long best = 0;
  for (uint64_t b = threadIdx.x; b < nblocks; b += 1024)
  {
    const unsigned long long v = blocktop[(uint64_t) p * nblocks + b];
    best = v > best ? v : best;
  }
  red[threadIdx.x] = best;
  __syncthreads();
  for (int d = 512; d > 0; d >>= 1)
  {
    if ((int) threadIdx.x < d && red[threadIdx.x + d] > red[threadIdx.x])
    {
      red[threadIdx.x] = red[threadIdx.x + d];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0)
  {
    parttop[p] = red[0];
    partstart[p] = offsets[(uint64_t) p * nblocks];
    if (p + 1 == nparts)
    {
      partstart[nparts] = offsets[(uint64_t) nparts * nblocks];
    }
  }
}

__global__ void __launch_bounds__(VSA_BLOCK)
k_partition_place(const PartInput in, uint64_t n,
                  uint32_t nparts, uint64_t totallength, uint64_t nblocks,
                  const uint64_t *__restrict__ offsets,
                  void *__restrict__ out)
{
  __shared__ unsigned int taken[VSA_PART_MAX];
  if (vsa_bid() >= nblocks) // surplus block of a folded grid
  {
    return;
  }
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
  for (uint32_t p = threadIdx.x; p < nparts; p += VSA_BLOCK)
  {
    taken[p] = 0;
  }
  __syncthreads();
  if (t < n && in.packbits != 0)
  {
    // rows of two words: key, value
    const uint64_t k = in.key[t * in.stride], v = in.val[t * in.stride];
    const uint32_t p = in.place(
        (uint32_t) (((k >> in.packbits) * nparts) / (totallength + 1)));
    const uint64_t slot = offsets[(uint64_t) p * nblocks + vsa_bid()] +
                          atomicAdd(&taken[p], 1u);
    uint4 row;
    row.x = (uint32_t) k;
    row.y = (uint32_t) (k >> 32);
    row.z = (uint32_t) v;
    row.w = (uint32_t) (v >> 32);
    reinterpret_cast<uint4 *>(out)[slot] = row;
  } else if (t < n)
  {
    const uint4 *src = reinterpret_cast<const uint4 *>(in.m + t);
    const uint4 lo = src[0], hi = src[1];
    const uint64_t dbstart = ((uint64_t) lo.w << 32) | lo.z;
    const uint32_t p =
        in.place((uint32_t) ((dbstart * nparts) / (totallength + 1)));
    const uint64_t slot = offsets[(uint64_t) p * nblocks + vsa_bid()] +
                          atomicAdd(&taken[p], 1u);
    uint4 *dst = reinterpret_cast<uint4 *>(reinterpret_cast<vsa_match *>(out) +
                                           slot);
    dst[0] = lo;
    dst[1] = hi;
  }
}

// ---- up to 8 parts (the ranks of one node): tiles of 2 048 records, eight
// per lane, everything counted in registers.  The kernels above spend their
// time in LDS atomics that all 256 lanes aim at the same few words (one part:
// one word) and in a 64-bit division per record; here a record's part is a sum
// of comparisons with the seven range boundaries, counts and largest right
// ends are kept per lane and part and reduced once per tile across the
// wavefront (DPP), and the tile leaves through LDS grouped by part, so that
// the rows of a part are written as one contiguous run.
#define VSA_PT_ITEMS 8
#define VSA_PT_TILE (VSA_BLOCK * VSA_PT_ITEMS)
#define VSA_PT_SMALL 8

#define VSA_DPP64(old, v, ctrl, rowmask)                                      \
  (((uint64_t) (uint32_t) __builtin_amdgcn_update_dpp(                        \
        (int) ((old) >> 32), (int) ((v) >> 32), ctrl, rowmask, 0xF, false)    \
    << 32) |                                                                  \
   (uint32_t) __builtin_amdgcn_update_dpp((int) (old), (int) (v), ctrl,       \
                                          rowmask, 0xF, false))

// lane 63 receives the maximum of all 64 lanes (an inclusive scan with max;
// lanes without a source keep their own value)
__device__ __forceinline__ uint64_t vsa_wave_inclusive_max64(uint64_t x)
{
  uint64_t y;
  y = VSA_DPP64(x, x, 0x111, 0xF); x = y > x ? y : x; // row_shr:1
  y = VSA_DPP64(x, x, 0x112, 0xF); x = y > x ? y : x; // row_shr:2
  y = VSA_DPP64(x, x, 0x114, 0xF); x = y > x ? y : x; // row_shr:4
  y = VSA_DPP64(x, x, 0x118, 0xF); x = y > x ? y : x; // row_shr:8
  y = VSA_DPP64(x, x, 0x142, 0xA); x = y > x ? y : x; // row_bcast:15
  y = VSA_DPP64(x, x, 0x143, 0xC); x = y > x ? y : x; // row_bcast:31
  return x;
}

// first position of part j: ceil(j (T + 1) / nparts); nothing lies in the
// parts from nparts on
__device__ __forceinline__ void part_bounds(uint64_t *bound, uint32_t nparts,
                                            uint64_t totallength)
{
  if (threadIdx.x <= VSA_PT_SMALL)
  {
    const uint64_t j = threadIdx.x;
    bound[j] = (j < nparts) ? (j * (totallength + 1) + nparts - 1) / nparts
                            : ~0ull;
  }
}

struct PartBounds
{
  uint64_t b[VSA_PT_SMALL - 1];
  __device__ __forceinline__ void load(const uint64_t *bound)
  {
#pragma unroll
    for (int j = 0; j < VSA_PT_SMALL - 1; j++)
    {
      b[j] = bound[j + 1];
    }
  }
  __device__ __forceinline__ uint32_t part(uint64_t dbstart) const
  {
    uint32_t p = 0;
#pragma unroll
    for (int j = 0; j < VSA_PT_SMALL - 1; j++)
    {
      p += dbstart >= b[j] ? 1u : 0u;
    }
    return p;
  }
};

__global__ void __launch_bounds__(VSA_BLOCK)
k_partition_count_small(const PartInput in, uint64_t n, uint32_t nparts,
                        uint64_t totallength, uint64_t ntiles,
                        uint32_t *__restrict__ blockhist,
                        unsigned long long *__restrict__ blocktop)
{
  __shared__ uint64_t bound[VSA_PT_SMALL + 1];
  __shared__ uint32_t wcount[VSA_BLOCK / 64][VSA_PT_SMALL];
  __shared__ uint64_t wtop[VSA_BLOCK / 64][VSA_PT_SMALL];
  const uint64_t tile = vsa_bid();
  if (tile >= ntiles) // surplus block of a folded grid
  {
    return;
  }
  part_bounds(bound, nparts, totallength);
  __syncthreads();
  PartBounds pb;
  pb.load(bound);
  uint32_t cnt[VSA_PT_SMALL];
  uint64_t top[VSA_PT_SMALL];
#pragma unroll
  for (int j = 0; j < VSA_PT_SMALL; j++)
  {
    cnt[j] = 0;
    top[j] = 0;
  }
#pragma unroll
  for (int i = 0; i < VSA_PT_ITEMS; i++)
  {
    const uint64_t t = tile * VSA_PT_TILE + (uint64_t) i * VSA_BLOCK +
                       threadIdx.x;
    if (t < n)
    {
      uint64_t dbstart, length;
      part_read(in, t, dbstart, length);
      const uint32_t p = in.place(pb.part(dbstart));
      // right end of the match in the index (cleanMUMcand.c: dbright)
      const uint64_t right = dbstart + length - 1;
#pragma unroll
      for (int j = 0; j < VSA_PT_SMALL; j++)
      {
        const bool hit = p == (uint32_t) j;
        cnt[j] += hit ? 1u : 0u;
        top[j] = (hit && right > top[j]) ? right : top[j];
      }
    }
  }
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
  for (int j = 0; j < VSA_PT_SMALL; j++)
  {
    const uint32_t c = vsa_wave_inclusive_sum(cnt[j]);
    const uint64_t m = vsa_wave_inclusive_max64(top[j]);
    if (lane == 63)
    {
      wcount[w][j] = c;
      wtop[w][j] = m;
    }
  }
  __syncthreads();
  if (threadIdx.x < nparts)
  {
    uint32_t c = 0;
    uint64_t m = 0;
    for (uint32_t k = 0; k < VSA_BLOCK / 64; k++)
    {
      c += wcount[k][threadIdx.x];
      m = wtop[k][threadIdx.x] > m ? wtop[k][threadIdx.x] : m;
    }
    blockhist[(uint64_t) threadIdx.x * ntiles + tile] = c;
    blocktop[(uint64_t) threadIdx.x * ntiles + tile] = m;
  }
}

// pairs (key, value) only: a tile of records would not fit the 64 KB of LDS
// a workgroup may declare
__global__ void __launch_bounds__(VSA_BLOCK)
k_partition_place_small(const PartInput in, uint64_t n, uint32_t nparts,
                        uint64_t totallength, uint64_t ntiles,
                        const uint64_t *__restrict__ offsets,
                        uint4 *__restrict__ out)
{
  __shared__ uint64_t bound[VSA_PT_SMALL + 1];
  __shared__ uint32_t wcount[VSA_BLOCK / 64][VSA_PT_SMALL];
  __shared__ uint32_t localbase[VSA_PT_SMALL + 1];
  __shared__ uint64_t globalbase[VSA_PT_SMALL];
  __shared__ uint4 stage[VSA_PT_TILE];
  const uint64_t tile = vsa_bid();
  if (tile >= ntiles) // surplus block of a folded grid
  {
    return;
  }
  part_bounds(bound, nparts, totallength);
  if (threadIdx.x < VSA_PT_SMALL)
  {
    globalbase[threadIdx.x] =
        threadIdx.x < nparts
            ? offsets[(uint64_t) threadIdx.x * ntiles + tile]
            : 0;
  }
  __syncthreads();
  PartBounds pb;
  pb.load(bound);
  uint4 row[VSA_PT_ITEMS];
  uint32_t parts = 0; // 4 bits per item: its part, 15 = no item
  uint32_t cnt[VSA_PT_SMALL];
#pragma unroll
  for (int j = 0; j < VSA_PT_SMALL; j++)
  {
    cnt[j] = 0;
  }
#pragma unroll
  for (int i = 0; i < VSA_PT_ITEMS; i++)
  {
    const uint64_t t = tile * VSA_PT_TILE + (uint64_t) i * VSA_BLOCK +
                       threadIdx.x;
    uint32_t p = 15;
    row[i] = make_uint4(0, 0, 0, 0);
    if (t < n)
    {
      const uint64_t k = in.key[t * in.stride], v = in.val[t * in.stride];
      row[i] = make_uint4((uint32_t) k, (uint32_t) (k >> 32), (uint32_t) v,
                          (uint32_t) (v >> 32));
      p = in.place(pb.part(k >> in.packbits));
    }
    parts |= p << (4 * i);
#pragma unroll
    for (int j = 0; j < VSA_PT_SMALL; j++)
    {
      cnt[j] += p == (uint32_t) j ? 1u : 0u;
    }
  }
  // where this lane's rows of part j start inside the tile's run of part j
  const uint32_t lane = threadIdx.x & 63, w = threadIdx.x >> 6;
  uint32_t before[VSA_PT_SMALL];
#pragma unroll
  for (int j = 0; j < VSA_PT_SMALL; j++)
  {
    const uint32_t incl = vsa_wave_inclusive_sum(cnt[j]);
    before[j] = incl - cnt[j];
    if (lane == 63)
    {
      wcount[w][j] = incl;
    }
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    uint32_t run = 0;
    for (int j = 0; j < VSA_PT_SMALL; j++)
    {
      localbase[j] = run;
      for (uint32_t k = 0; k < VSA_BLOCK / 64; k++)
      {
        run += wcount[k][j];
      }
    }
    localbase[VSA_PT_SMALL] = run;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < VSA_PT_SMALL; j++)
  {
    uint32_t lower = localbase[j];
    for (uint32_t k = 0; k < w; k++)
    {
      lower += wcount[k][j];
    }
    before[j] += lower;
  }
#pragma unroll
  for (int i = 0; i < VSA_PT_ITEMS; i++)
  {
    const uint32_t p = (parts >> (4 * i)) & 15;
    uint32_t at = 0;
#pragma unroll
    for (int j = 0; j < VSA_PT_SMALL; j++)
    {
      const bool hit = p == (uint32_t) j;
      at = hit ? before[j] : at;
      before[j] += hit ? 1u : 0u;
    }
    if (p != 15)
    {
      stage[at] = row[i];
    }
  }
  __syncthreads();
  const uint32_t total = localbase[VSA_PT_SMALL];
  for (uint32_t r = threadIdx.x; r < total; r += VSA_BLOCK)
  {
    uint32_t p = 0;
#pragma unroll
    for (int j = 1; j < VSA_PT_SMALL; j++)
    {
      p += r >= localbase[j] ? 1u : 0u;
    }
    out[globalbase[p] + (r - localbase[p])] = stage[r];
  }
}

// counts[p], maxright[p] by part from the summary by place in the output
__global__ void __launch_bounds__(VSA_PART_MAX)
k_partition_meta(const uint64_t *__restrict__ summary, uint32_t nparts,
                 uint32_t own, uint64_t *__restrict__ meta)
{
  const uint32_t p = threadIdx.x;
  if (p < nparts)
  {
    const uint32_t at = p < own ? p : (p == own ? nparts - 1 : p - 1);
    meta[p] = summary[at + 1] - summary[at];
    meta[nparts + p] = summary[VSA_PART_MAX + 1 + at];
  }
}

namespace
{

int partition_impl(const vsa_result *result, uint32_t nparts, int ownpart,
                   uint64_t totallength, void *device_matches,
                   uint64_t *counts, uint64_t *maxright,
                   uint64_t *device_meta);

} // namespace

extern "C" int vsa_result_partition_own(const vsa_result *result,
                                        uint32_t nparts, int ownpart,
                                        uint64_t totallength,
                                        void *device_matches, uint64_t *counts,
                                        uint64_t *maxright)
{
  if (counts == nullptr)
  {
    VSA_ERROR("vsa_result_partition: bad argument (counts)");
    return -1;
  }
  return partition_impl(result, nparts, ownpart, totallength, device_matches,
                        counts, maxright, nullptr);
}

extern "C" int vsa_result_partition_device(const vsa_result *result,
                                           uint32_t nparts, int ownpart,
                                           uint64_t totallength,
                                           void *device_matches,
                                           uint64_t *device_meta)
{
  if (device_meta == nullptr)
  {
    VSA_ERROR("vsa_result_partition_device: bad argument (device_meta)");
    return -1;
  }
  return partition_impl(result, nparts, ownpart, totallength, device_matches,
                        nullptr, nullptr, device_meta);
}

namespace
{

int partition_impl(const vsa_result *result, uint32_t nparts, int ownpart,
                   uint64_t totallength, void *device_matches,
                   uint64_t *counts, uint64_t *maxright,
                   uint64_t *device_meta)
{
  if (result == nullptr || nparts == 0 ||
      nparts > VSA_PART_MAX || ownpart >= (int) nparts ||
      (result->count > 0 && device_matches == nullptr))
  {
    VSA_ERROR("vsa_result_partition: bad argument (1..256 parts, own part "
              "among them or < 0)");
    return -1;
  }
  for (uint32_t p = 0; p < nparts && counts != nullptr; p++)
  {
    counts[p] = 0;
    if (maxright != nullptr)
    {
      maxright[p] = 0;
    }
  }
  const uint64_t n = result->count;
  if (vsa_set_device(result->device) != 0)
  {
    return -100;
  }
  hipStream_t stream = nullptr;
  vsa_dev_set_stream(stream);
  if (n == 0)
  {
    if (device_meta != nullptr)
    {
      VSA_HIP(hipMemsetAsync(device_meta, 0, 2 * (size_t) nparts * 8,
                             stream));
    }
    return 0;
  }
  // tiles of eight records per lane for pairs that go to up to 8 parts (the
  // general kernels take records, and more parts)
  const bool small = nparts <= VSA_PT_SMALL &&
                     result->packbits != 0;
  const uint64_t nblocks =
                     small ? (n + VSA_PT_TILE - 1) / VSA_PT_TILE
                           : blocksfor(n),
                 cells = (uint64_t) nparts * nblocks;
  DevBuf hist, top, offsets, summary, temp;
  uint64_t host[2 * VSA_PART_MAX + 1];
  size_t tb = 0;
  if (hist.alloc((cells + 1) * 4) || top.alloc(cells * 8) ||
      offsets.alloc((cells + 1) * 8) ||
      summary.alloc((2 * VSA_PART_MAX + 1) * 8))
  {
    return -100;
  }
  PartInput in;
  in.m = result->matches;
  in.key = reinterpret_cast<const uint64_t *>(result->matches);
  in.val = result->packvals;
  in.stride = 1;
  if (result->packbits != 0 && result->packvals == nullptr)
  {
    // rows of (key, value) pairs (vsa_rows_partition_device)
    in.val = in.key + 1;
    in.stride = 2;
  }
  in.packbits = result->packbits;
  in.nparts = nparts;
  in.own = ownpart < 0 ? nparts : (uint32_t) ownpart;
  VSA_HIP(hipMemsetAsync(hist.as<uint32_t>() + cells, 0, 4, stream));
  if (small)
  {
    k_partition_count_small<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        in, n, nparts, totallength, nblocks, hist.as<uint32_t>(),
        top.as<unsigned long long>());
  } else
  {
    k_partition_count<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        in, n, nparts, totallength, nblocks, hist.as<uint32_t>(),
        top.as<unsigned long long>());
  }
  VSA_HIP(hipGetLastError());
  auto widen = rocprim::make_transform_iterator(hist.as<uint32_t>(),
                                                U32ToU64());
  VSA_HIP(rocprim::exclusive_scan(nullptr, tb, widen, offsets.as<uint64_t>(),
                                  (uint64_t) 0, (size_t) (cells + 1),
                                  rocprim::plus<uint64_t>(), stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(temp.p, tb, widen, offsets.as<uint64_t>(),
                                  (uint64_t) 0, (size_t) (cells + 1),
                                  rocprim::plus<uint64_t>(), stream));
  k_partition_summary<<<nparts, 1024, 0, stream>>>(
      offsets.as<uint64_t>(), top.as<unsigned long long>(), nparts, nblocks,
      summary.as<uint64_t>(),
      summary.as<unsigned long long>() + VSA_PART_MAX + 1);
  VSA_HIP(hipGetLastError());
  if (small)
  {
    k_partition_place_small<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        in, n, nparts, totallength, nblocks, offsets.as<uint64_t>(),
        reinterpret_cast<uint4 *>(device_matches));
  } else
  {
    k_partition_place<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        in, n, nparts, totallength, nblocks, offsets.as<uint64_t>(),
        device_matches);
  }
  VSA_HIP(hipGetLastError());
  if (device_meta != nullptr)
  {
    // the numbers stay on the device (the input of the ranks' all-gather):
    // nothing here waits for the GPU
    k_partition_meta<<<1, VSA_PART_MAX, 0, stream>>>(
        summary.as<uint64_t>(), nparts, in.own, device_meta);
    VSA_HIP(hipGetLastError());
    return 0;
  }
  VSA_HIP(hipMemcpyAsync(host, summary.p, (2 * VSA_PART_MAX + 1) * 8,
                         hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  for (uint32_t p = 0; p < nparts; p++)
  {
    // (the device counted by place in the output)
    const uint32_t at = p < in.own ? p : (p == in.own ? nparts - 1 : p - 1);
    counts[p] = host[at + 1] - host[at];
    if (maxright != nullptr)
    {
      maxright[p] = host[VSA_PART_MAX + 1 + at];
    }
  }
  return 0;
}

} // namespace

extern "C" int vsa_rows_partition_device(const void *device_rows,
                                         uint64_t nrows, uint32_t lengthbits,
                                         uint32_t nparts, int ownpart,
                                         uint64_t totallength, int device,
                                         void *device_out,
                                         uint64_t *device_meta)
{
  if ((nrows > 0 && device_rows == nullptr) || device_meta == nullptr ||
      lengthbits == 0 || lengthbits > 16)
  {
    VSA_ERROR("vsa_rows_partition_device: bad argument");
    return -1;
  }
  // the rows seen as a packed result whose values lie next to their keys
  vsa_result view;
  view.device = device;
  view.count = nrows;
  view.matches =
      reinterpret_cast<vsa_match *>(const_cast<void *>(device_rows));
  memset(&view.stats, 0, sizeof view.stats);
  view.packbits = lengthbits;
  view.packvals = nullptr;
  return partition_impl(&view, nparts, ownpart, totallength, device_out,
                        nullptr, nullptr, device_meta);
}

extern "C" int vsa_result_partition(const vsa_result *result, uint32_t nparts,
                                    uint64_t totallength,
                                    void *device_matches, uint64_t *counts,
                                    uint64_t *maxright)
{
  return vsa_result_partition_own(result, nparts, -1, totallength,
                                  device_matches, counts, maxright);
}

extern "C" int vsa_findmumcandidates_grouped(const vsa_index *index,
                                             const vsa_queries *queries,
                                             uint64_t searchlength,
                                             uint32_t lengthbits,
                                             uint32_t nparts, int ownpart,
                                             void *device_rows,
                                             uint64_t capacity,
                                             uint64_t *device_meta,
                                             vsa_result **result)
{
  const int rc = vsa_findmumcandidates_packed(index, queries, searchlength,
                                              lengthbits, result);
  if (rc != 0)
  {
    return rc;
  }
  if ((*result)->count > capacity)
  {
    return 1; // the caller makes room and groups the result itself
  }
  // no return to the caller between the search and the grouping: the GPU
  // waits for one kernel launch, not for an interpreter
  const int prc = vsa_result_partition_device(*result, nparts, ownpart,
                                              index->n, device_rows,
                                              device_meta);
  if (prc != 0)
  {
    vsa_result_free(*result);
    *result = nullptr;
  }
  return prc;
}

// ---- batches whose thresholds (-e Kp / -h Kp) are 0 for the short reads and
// > 0 for the long ones: the reference sends the former through the exact
// search and the latter through splitesaapm, read by read
// (Vmengine/approxcompl.c:167-191).  Here the batch is cut into the two kinds,
// each kind runs as a batch of its own over the same symbols, and the two
// lists are merged back into query order.

__global__ void __launch_bounds__(VSA_BLOCK)
k_subquery_gather(const uint64_t *__restrict__ start,
                  const uint64_t *__restrict__ length,
                  const uint64_t *__restrict__ which, uint64_t n,
                  uint64_t *__restrict__ substart,
                  uint64_t *__restrict__ sublength)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t q = which[i];
    substart[i] = start[q];
    sublength[i] = length[q];
  }
}

// rows [0, nfirst) come from the sub-batch `whicha`, the others from
// `whichb`; their queryseq becomes the number in the whole batch
__global__ void __launch_bounds__(VSA_BLOCK)
k_subquery_renumber(vsa_match *__restrict__ rows, uint64_t nfirst,
                    uint64_t n, const uint64_t *__restrict__ whicha,
                    const uint64_t *__restrict__ whichb, uint64_t seqoffset,
                    uint32_t *__restrict__ keys, uint32_t *__restrict__ index)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t sub = rows[i].queryseq;
    const uint64_t q = (i < nfirst) ? whicha[sub] : whichb[sub];
    rows[i].queryseq = q + seqoffset;
    keys[i] = (uint32_t) q;
    index[i] = (uint32_t) i;
  }
}

namespace
{

struct SubQueries
{
  vsa_queries q;
  DevBuf start, length, which;
};

// the queries `which` (ascending numbers) of a batch as a batch that shares
// the symbols
int make_subqueries(const vsa_index *index, const vsa_queries *queries,
                    const std::vector<uint64_t> &which, SubQueries &sub)
{
  const uint64_t n = which.size();

  sub.q.device = queries->device;
  sub.q.nq = n;
  sub.q.nsymbols = queries->nsymbols;
  sub.q.symbols = queries->symbols;
  sub.q.seqoffset = 0;
  sub.q.dense = false;
  sub.q.hlength.resize(n);
  sub.q.minlength = n ? ~0ull : 0;
  sub.q.maxlength = 0;
  for (uint64_t i = 0; i < n; i++)
  {
    const uint64_t m = queries->uniform ? queries->maxlength
                                        : queries->hlength[which[i]];
    sub.q.hlength[i] = m;
    sub.q.minlength = std::min(sub.q.minlength, m);
    sub.q.maxlength = std::max(sub.q.maxlength, m);
  }
  sub.q.uniform = n != 0 && sub.q.minlength == sub.q.maxlength;
  vsa_dev_set_stream(index->stream);
  if (sub.start.alloc(n * 8) || sub.length.alloc(n * 8) ||
      sub.which.alloc(n * 8))
  {
    return -100;
  }
  sub.q.start = sub.start.as<uint64_t>();
  sub.q.length = sub.length.as<uint64_t>();
  if (n > 0)
  {
    VSA_HIP(hipMemcpyAsync(sub.which.p, which.data(), n * 8,
                           hipMemcpyHostToDevice, index->stream));
    k_subquery_gather<<<gridfor(n), VSA_BLOCK, 0, index->stream>>>(
        queries->start, queries->length, sub.which.as<uint64_t>(), n,
        sub.q.start, sub.q.length);
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipStreamSynchronize(index->stream));
  }
  return 0;
}

int approx_batch(const vsa_index *index, const vsa_queries *queries,
                 int doedist, uint64_t distvalue, int percent,
                 vsa_result **result);

// explicitk (the second pass of a "best of" job): the threshold of every read
// instead of distvalue percent of its length; VSA_NO_THRESHOLD = the read is
// left out
#define VSA_NO_THRESHOLD 0xFFFFFFFFu
int approx_mixed(const vsa_index *index, const vsa_queries *queries,
                 int doedist, uint64_t distvalue, vsa_result **result,
                 const std::vector<uint32_t> *explicitk = nullptr)
{
  const uint64_t nq = queries->nq;
  std::vector<uint64_t> exact, approx;
  std::vector<uint32_t> approxk;
  uint64_t qlimit = nq, failk = 0, failm = 0;
  bool failshort = false;

  if (nq >= 0xFFFFFFFFull)
  {
    VSA_ERROR("a batch of %lu reads that mixes thresholds 0 and > 0 is not "
              "covered by the GPU engine", (unsigned long) nq);
    return VSA_NOT_COVERED;
  }
  for (uint64_t q = 0; q < nq; q++)
  {
    const uint64_t m = queries->uniform ? queries->maxlength
                                        : queries->hlength[q],
                   k = explicitk != nullptr ? (*explicitk)[q]
                                            : (m * distvalue) / 100;
    if (explicitk != nullptr && k == VSA_NO_THRESHOLD)
    {
      continue;
    }
    if (k == 0)
    {
      if (m < index->pl)
      {
        // exactcompl.c:179-185
        qlimit = q;
        failshort = true;
        failm = m;
        break;
      }
      exact.push_back(q);
    } else
    {
      if (k >= m)
      {
        // splitesaapm.c:496-501
        qlimit = q;
        failk = k;
        failm = m;
        break;
      }
      approx.push_back(q);
      approxk.push_back((uint32_t) k);
    }
  }
  SubQueries sa, sb;
  vsa_result *ra = nullptr, *rb = nullptr;
  int rc = 0;
  if (!exact.empty())
  {
    rc = make_subqueries(index, queries, exact, sa);
    if (rc == 0)
    {
      rc = vsa_findcompletematches(index, &sa.q, &ra);
    }
  }
  if (rc == 0 && !approx.empty())
  {
    rc = make_subqueries(index, queries, approx, sb);
    if (rc == 0)
    {
      apm_explicitk = explicitk != nullptr ? approxk.data() : nullptr;
      rc = approx_batch(index, &sb.q, doedist, distvalue, 1, &rb);
      apm_explicitk = nullptr;
    }
  }
  if (rc != 0)
  {
    vsa_result_free(ra);
    vsa_result_free(rb);
    return rc;
  }
  if (vsa_set_device(index->device) != 0)
  {
    vsa_result_free(ra);
    vsa_result_free(rb);
    return -100;
  }
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  vsa_result *res = newresult(index->device);
  const uint64_t na = ra ? ra->count : 0, nb = rb ? rb->count : 0,
                 n = na + nb;
  // (a macro that returns would leak the three lists)
  auto merge = [&]() -> int {
    DevBuf all, merged, keys, keys2, order, order2, temp;
    size_t tb = 0;
    if (n == 0)
    {
      return 0;
    }
    if (n >= 0xFFFFFFFFull)
    {
      VSA_ERROR("%lu matches of a batch that mixes thresholds 0 and > 0 are "
                "not covered by the GPU engine", (unsigned long) n);
      return VSA_NOT_COVERED;
    }
    if (all.alloc(n * sizeof(vsa_match)) ||
        merged.alloc(n * sizeof(vsa_match)) || keys.alloc(n * 4) ||
        keys2.alloc(n * 4) || order.alloc(n * 4) || order2.alloc(n * 4))
    {
      return -100;
    }
    if (na > 0)
    {
      VSA_HIP(hipMemcpyAsync(all.p, ra->matches, na * sizeof(vsa_match),
                             hipMemcpyDeviceToDevice, stream));
    }
    if (nb > 0)
    {
      VSA_HIP(hipMemcpyAsync(all.as<vsa_match>() + na, rb->matches,
                             nb * sizeof(vsa_match), hipMemcpyDeviceToDevice,
                             stream));
    }
    k_subquery_renumber<<<gridfor(n), VSA_BLOCK, 0, stream>>>(
        all.as<vsa_match>(), na, n, sa.which.as<uint64_t>(),
        sb.which.as<uint64_t>(), queries->seqoffset, keys.as<uint32_t>(),
        order.as<uint32_t>());
    VSA_HIP(hipGetLastError());
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, keys.as<uint32_t>(), keys2.as<uint32_t>(),
        order.as<uint32_t>(), order2.as<uint32_t>(), (size_t) n, 0u,
        bitsfor(nq), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, keys.as<uint32_t>(), keys2.as<uint32_t>(),
        order.as<uint32_t>(), order2.as<uint32_t>(), (size_t) n, 0u,
        bitsfor(nq), stream));
    k_gather_matches<<<gridfor(n), VSA_BLOCK, 0, stream>>>(
        all.as<vsa_match>(), order2.as<uint32_t>(), n,
        merged.as<vsa_match>());
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipStreamSynchronize(stream));
    res->matches = (vsa_match *) merged.release();
    return 0;
  };
  rc = merge();
  res->count = res->stats.count = n;
  for (const vsa_result *r : {(const vsa_result *) ra,
                              (const vsa_result *) rb})
  {
    if (r != nullptr)
    {
      res->stats.sumlength += r->stats.sumlength;
      res->stats.searches += r->stats.searches;
      res->stats.kernel_searches += r->stats.kernel_searches;
      res->stats.search_kernel_ms += r->stats.search_kernel_ms;
      res->stats.total_device_ms += r->stats.total_device_ms;
      res->stats.first_kernel_ms += r->stats.first_kernel_ms;
    }
  }
  vsa_result_free(ra);
  vsa_result_free(rb);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  if (qlimit < nq)
  {
    // the reads before the failing one have been answered
    if (failshort)
    {
      VSA_ERROR("patternlength=%lu must be >= %lu=prefixlen",
                (unsigned long) failm, (unsigned long) index->pl);
    } else
    {
      VSA_ERROR("threshold=%lu>=%lu=patternlen not allowed",
                (unsigned long) failk, (unsigned long) failm);
    }
    return -2;
  }
  return 0;
}

} // namespace

// best[q - seqoffset] = the smallest distance among the matches of read q
// (the distance of a match travels in its querystart field)
__global__ void __launch_bounds__(VSA_BLOCK)
k_best_distance(const vsa_match *__restrict__ matches, uint64_t n,
                uint64_t seqoffset, uint32_t *__restrict__ best)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    atomicMin(best + (matches[i].queryseq - seqoffset),
              (uint32_t) matches[i].querystart);
  }
}

namespace
{

// vmatch -complete -e Kb | -h Kb, "best of" (Vmengine/initcompl.c:59-77): read
// by read -- decidefcm restores the job's K in front of every read,
// Vmengine/fcomplete.c:251-252 -- the reference looks for the smallest
// threshold t <= m K / 100 at which the read has a match at all (a binary
// search over existence checks, Vmengine/approxcompl.c:80-122) and then
// reports the read's matches at threshold t; a read without a match within
// m K / 100 reports nothing.  Here: one pass at the percent thresholds gives
// every read's smallest distance, a second pass runs every read at exactly
// that threshold (the regions, and with them the order of the matches, are
// those of the threshold: Vmengine/splitesaapm.c:458-558).
int approx_bestof(const vsa_index *index, const vsa_queries *queries,
                  int doedist, uint64_t distvalue, vsa_result **result)
{
  const uint64_t nq = queries->nq;
  vsa_result *first = nullptr;
  *result = nullptr;
  if (nq >= 0xFFFFFFFFull)
  {
    VSA_ERROR("a best-of batch of %lu reads is not covered by the GPU engine",
              (unsigned long) nq);
    return VSA_NOT_COVERED;
  }
  int rc = vsa_findapproxcompletematches(index, queries, doedist, distvalue,
                                         1, &first);
  if (rc != 0)
  {
    vsa_result_free(first);
    return rc;
  }
  std::vector<uint32_t> best(nq, VSA_NO_THRESHOLD);
  {
    hipStream_t stream = index->stream;
    vsa_dev_set_stream(stream);
    DevBuf dbest;
    if (vsa_set_device(index->device) != 0 || dbest.alloc((nq + 1) * 4))
    {
      vsa_result_free(first);
      return -100;
    }
    auto run = [&]() -> int {
      VSA_HIP(hipMemsetAsync(dbest.p, 0xFF, (nq + 1) * 4, stream));
      if (first->count > 0)
      {
        k_best_distance<<<gridfor(first->count), VSA_BLOCK, 0, stream>>>(
            first->matches, first->count, queries->seqoffset,
            dbest.as<uint32_t>());
        VSA_HIP(hipGetLastError());
      }
      if (nq > 0)
      {
        VSA_HIP(hipMemcpyAsync(best.data(), dbest.p, nq * 4,
                               hipMemcpyDeviceToHost, stream));
      }
      VSA_HIP(hipStreamSynchronize(stream));
      return 0;
    };
    rc = run();
  }
  const vsa_stats s1 = first->stats;
  vsa_result_free(first);
  if (rc != 0)
  {
    return rc;
  }
  // an exact match found by the first pass has distance 0 whichever way it
  // was found (the percent threshold of a short read is 0: exact search,
  // whose matches carry querystart 0 as well)
  rc = approx_mixed(index, queries, doedist, distvalue, result, &best);
  if (*result != nullptr)
  {
    (*result)->stats.searches += s1.searches;
    (*result)->stats.total_device_ms += s1.total_device_ms;
  }
  return rc;
}

} // namespace

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
  if (queries->device != index->device)
  {
    VSA_ERROR("queries live on device %d, index on device %d",
              queries->device, index->device);
    return -1;
  }
  if (queries->rows != nullptr)
  {
    // a packed batch: the approximate kernels read bytes
    if (vsa_set_device(index->device) != 0 ||
        vsa_queries_bytes(queries, index->stream) != 0)
    {
      return -100;
    }
  }
  if (percent == 2)
  {
    if (index->numofchars != 4)
    {
      VSA_ERROR("approximate search on alphabets of %lu symbols is not "
                "covered by the GPU engine",
                (unsigned long) index->numofchars);
      return VSA_NOT_COVERED;
    }
    return approx_bestof(index, queries, doedist, distvalue, result);
  }
  if (percent != 0 && index->bck != nullptr && index->numofchars == 4 &&
      (queries->minlength * distvalue) / 100 == 0 &&
      (queries->maxlength * distvalue) / 100 != 0)
  {
    return approx_mixed(index, queries, doedist, distvalue, result);
  }
  return approx_batch(index, queries, doedist, distvalue, percent, result);
}

namespace
{

int approx_batch(const vsa_index *index, const vsa_queries *queries,
                 int doedist, uint64_t distvalue, int percent,
                 vsa_result **result)
{
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
  if (rc != 0 && rc != VSA_NOT_COVERED)
  {
    return rc;
  }
  if (rc == 0 && plan.allexact)
  {
    // approxcompl.c:167-176: threshold 0 is the exact search
    return vsa_findcompletematches(index, queries, result);
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  if (rc == 0)
  {
    rc = (index->isize == 4)
             ? run_approx<uint32_t>(index, queries, doedist != 0, plan, res)
             : run_approx<uint64_t>(index, queries, doedist != 0, plan, res);
  }
  if (rc == VSA_NOT_COVERED)
  {
    // pieces with a threshold of their own, patterns that are not cut,
    // Hamming distance with wildcards in a read: the reference's esaapm /
    // esahamming configurations (approx_tree.inc)
    TreePlan tplan;
    vsa_result_free(res);
    res = nullptr;
    rc = apm_treeplan(index, queries, doedist != 0, distvalue, percent != 0,
                      tplan);
    if (rc != 0)
    {
      return rc;
    }
    res = newresult(index->device);
    rc = (index->isize == 4)
             ? run_approx_tree<uint32_t>(index, queries, doedist != 0, tplan,
                                         res)
             : run_approx_tree<uint64_t>(index, queries, doedist != 0, tplan,
                                         res);
    plan.qlimit = tplan.qlimit;
    plan.failk = tplan.failk;
    plan.failm = tplan.failm;
  }
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

} // namespace

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
  vsa_dev_set_stream(stream);
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

// word 0 / word 1 of row i of (key, value) pairs
// word 0 (key) or 1 (value) of row i of two lists of rows laid end to end
struct RowWord
{
  const uint64_t *rows, *more;
  uint64_t nrows; // in `rows`
  uint32_t word;
  __device__ uint64_t operator()(size_t i) const
  {
    return i < nrows ? rows[2 * i + word] : more[2 * (i - nrows) + word];
  }
};

// pairs -> records, in place order (vsa_result_fetch of a packed result)
__global__ void __launch_bounds__(VSA_BLOCK)
k_unpack_pairs(const uint64_t *__restrict__ key,
               const uint64_t *__restrict__ val, uint64_t n,
               uint32_t packbits, vsa_match *__restrict__ out)
{
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (t < n)
  {
    const uint64_t k = key[t], v = val[t], mask = (1ull << packbits) - 1;
    vsa_match m;
    m.length = mask - (k & mask);
    m.dbstart = k >> packbits;
    m.queryseq = v >> 16;
    m.querystart = v & 0xFFFFu;
    out[t] = m;
  }
}

int vsa_unpack_result(const vsa_result *r, uint64_t count, vsa_match *device)
{
  k_unpack_pairs<<<gridfor(count), VSA_BLOCK>>>(
      reinterpret_cast<const uint64_t *>(r->matches), r->packvals, count,
      r->packbits, device);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int vsa_mumuniqueinquery_range_packed2(const void *device_rows,
                                                  uint64_t nrows,
                                                  const void *more_rows,
                                                  uint64_t nmore,
                                                  uint32_t lengthbits,
                                                  uint64_t totallength,
                                                  int device,
                                                  uint64_t carry_dbright,
                                                  vsa_result **result)
{
  if (result == nullptr || (device_rows == nullptr && nrows > 0) ||
      (more_rows == nullptr && nmore > 0) || lengthbits == 0 ||
      lengthbits > 16 || lengthbits + bitsfor(totallength) > 64)
  {
    VSA_ERROR("vsa_mumuniqueinquery_range_packed: bad argument");
    return -1;
  }
  *result = nullptr;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(device);
  hipStream_t stream = nullptr; // default stream: no index handle here
  vsa_dev_set_stream(stream);
  Timer tall(stream);
  DevBuf mums;
  uint64_t nm = 0, sum = 0;
  const uint64_t total = nrows + nmore;
  tall.start();
  int rc = 0;
  if (total > 0)
  {
    // the sort reads the rows as they lie (the first pass of the radix sort
    // takes iterators): no split into two arrays, no copy of the two lists
    // into one
    const uint64_t *rows = reinterpret_cast<const uint64_t *>(device_rows),
                   *more = reinterpret_cast<const uint64_t *>(more_rows);
    auto rowkeys = rocprim::make_transform_iterator(
        rocprim::counting_iterator<size_t>(0), RowWord{rows, more, nrows, 0});
    auto rowvals = rocprim::make_transform_iterator(
        rocprim::counting_iterator<size_t>(0), RowWord{rows, more, nrows, 1});
    rc = mumfilter_packed<uint64_t>(rowkeys, rowvals, total, lengthbits,
                                    bitsfor(totallength), stream, mums, &nm,
                                    &sum, carry_dbright);
  }
  tall.stop();
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  (void) hipStreamSynchronize(stream);
  res->count = nm;
  res->matches = (vsa_match *) mums.release();
  res->stats.count = nm;
  res->stats.candidates = total;
  res->stats.sumlength = sum;
  res->stats.total_device_ms = tall.ms();
  *result = res;
  return 0;
}

extern "C" int vsa_mumuniqueinquery_range_packed(const void *device_rows,
                                                 uint64_t nrows,
                                                 uint32_t lengthbits,
                                                 uint64_t totallength,
                                                 int device,
                                                 uint64_t carry_dbright,
                                                 vsa_result **result)
{
  return vsa_mumuniqueinquery_range_packed2(device_rows, nrows, nullptr, 0,
                                            lengthbits, totallength, device,
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
                     ? run_repeats<uint32_t, uint32_t>(index, searchlength, res)
                     : run_repeats<uint64_t, uint64_t>(index, searchlength, res);
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
                     ? run_supermax<uint32_t, uint32_t>(index, searchlength, res)
                     : run_supermax<uint64_t, uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findtandems(const vsa_index *index, uint64_t searchlength,
                               vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findtandems: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->hasindexedqueries)
  {
    // Vmengine/ftandem.c:271-275
    VSA_ERROR("tandem repeat search does not allow query files in index");
    return -2;
  }
  if (searchlength == 0)
  {
    VSA_ERROR("tandem repeat search needs a length >= 1");
    return -2;
  }
  if (index->tis_alloc == nullptr || index->suf == nullptr ||
      index->lcp == nullptr)
  {
    VSA_ERROR("tables tis, suf and lcp are required");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_tandems<uint32_t, uint32_t>(index, searchlength, res)
                     : run_tandems<uint64_t, uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findmaximaluniquematches_range(const vsa_index *index,
                                                  uint64_t searchlength,
                                                  uint64_t first,
                                                  uint64_t last,
                                                  vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmaximaluniquematches: NULL argument");
    return -1;
  }
  if (first > last)
  {
    VSA_ERROR("vsa_findmaximaluniquematches_range: first > last");
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
                     ? run_selfmum<uint32_t>(index, searchlength, first, last,
                                             res)
                     : run_selfmum<uint64_t>(index, searchlength, first, last,
                                             res);
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
  return vsa_findmaximaluniquematches_range(index, searchlength, 2, ~0ull,
                                            result);
}
