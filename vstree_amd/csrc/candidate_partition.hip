// MUM candidates grouped by the range of the index their dbstart falls into:
// what every replica does before the exchange of the N > 1 form
// (kurtz/cleanMUMcand.c:55-118 runs per range afterwards).
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

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
