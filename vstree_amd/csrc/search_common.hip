// What the translation units of the search engine share and what instantiates
// rocPRIM on their behalf (see search_host.hpp).
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

// Small results the host needs before it can go on (counts, maxima) come
// back through a page of pinned memory: a device-to-host copy into pageable
// memory is staged by the runtime and costs 30-150 us each, several times
// per batch.  The page lives as long as the thread (never freed: the runtime
// may be gone when thread-local destructors run).
namespace
{

// up to VSA_FETCH_MAX words from anywhere in device memory into the calling
// thread's page of pinned host memory, which the device writes directly: one
// tiny kernel instead of one copy per group of adjacent words (a copy of 8
// bytes occupies the queue for 5 us; a step asks for three groups twice)
#define VSA_FETCH_MAX 8
struct FetchList
{
  const void *src[VSA_FETCH_MAX];
  uint32_t bytes[VSA_FETCH_MAX];
  int count;
};

__global__ void k_fetch_words(const FetchList list, uint64_t *__restrict__ page)
{
  const int i = threadIdx.x;
  if (i < list.count)
  {
    uint64_t v = 0;
    const uint8_t *p = (const uint8_t *) list.src[i];
    if (list.bytes[i] == 8)
    {
      v = *(const uint64_t *) p;
    } else
    {
      for (uint32_t b = 0; b < list.bytes[i]; b++)
      {
        v |= (uint64_t) p[b] << (8 * b);
      }
    }
    page[i] = v;
  }
}

} // namespace

int fetchwords(hipStream_t stream, const Fetch *items, int count,
               uint64_t *out)
{
  static thread_local uint64_t *page = nullptr, *devpage = nullptr;
  if (page == nullptr)
  {
    void *v = nullptr, *d = nullptr;
    // (portable and mapped: the thread may work on another device next time)
    VSA_HIP(hipHostMalloc(&v, 4096,
                          hipHostMallocPortable | hipHostMallocMapped));
    page = (uint64_t *) v;
    if (hipHostGetDevicePointer(&d, v, 0) == hipSuccess)
    {
      devpage = (uint64_t *) d;
    } else
    {
      (void) hipGetLastError();
    }
  }
  if (count > 512)
  {
    return -100;
  }
  bool aligned = true;
  for (int i = 0; i < count; i++)
  {
    aligned = aligned && items[i].bytes <= 8 &&
              (items[i].bytes != 8 || ((uintptr_t) items[i].src & 7u) == 0);
  }
  if (devpage != nullptr && count > 1 && count <= VSA_FETCH_MAX && aligned)
  {
    FetchList list;
    list.count = count;
    for (int i = 0; i < count; i++)
    {
      list.src[i] = items[i].src;
      list.bytes[i] = (uint32_t) items[i].bytes;
    }
    k_fetch_words<<<1, 64, 0, stream>>>(list, devpage);
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipStreamSynchronize(stream));
    for (int i = 0; i < count; i++)
    {
      out[i] = page[i];
    }
    return 0;
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

namespace
{

// order-preserving compaction of 32-byte records: slot[] = exclusive scan of
// keep[] (rocprim::select moves records of this size at a fraction of the
// memory rate: 3.6 ms for 15.6 M records, this pair of passes 0.2 ms)
__global__ void __launch_bounds__(VSA_BLOCK)
k_scatter_kept(const vsa_match *__restrict__ in,
               const uint8_t *__restrict__ keep,
               const uint32_t *__restrict__ slot, uint64_t count,
               vsa_match *__restrict__ out, uint64_t *__restrict__ nkept)
{
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
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

struct MatchLength
{
  __device__ uint64_t operator()(const vsa_match &m) const
  {
    return m.length;
  }
};

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

} // namespace

hipError_t shard_summary(const unsigned long long *cursors, uint32_t nshards,
                         uint64_t *offsets, uint64_t *summary,
                         hipStream_t stream)
{
  k_shard_summary<<<1, 1024, 0, stream>>>(cursors, nshards, offsets, nullptr,
                                          nullptr, summary);
  return hipGetLastError();
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

hipError_t gather_matches(const vsa_match *in, const uint32_t *order,
                          uint64_t n, vsa_match *out, hipStream_t stream)
{
  k_gather_matches<<<gridfor(n), VSA_BLOCK, 0, stream>>>(in, order, n, out);
  return hipGetLastError();
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
  VSA_HIP(gather_matches(in, order2.as<uint32_t>(), n, out, stream));
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
