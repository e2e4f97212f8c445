// Host-side helpers shared by the translation units of the search engine
// (esa_search.hip, selfmum_search.hip, approx_entry.hip, selfmatch_entry.hip,
// candidate_partition.hip, index_derive.hip).  Small things are inline here;
// what instantiates rocPRIM is defined once, in search_common.hip.
#ifndef VSA_SEARCH_HOST_HPP
#define VSA_SEARCH_HOST_HPP
#include <cstring>
#include <algorithm>
#include "esa_device.hpp"

#define VSA_BLOCK 256
#define VSA_CURSOR_STRIDE 8   // uint64 words: one cursor per 64-byte line
#define VSA_CURSOR_SHARDS 2048 // power of two
#define VSA_HIDDEN __attribute__((visibility("hidden")))

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

// (search_common.hip)
VSA_HIDDEN int fetchwords(hipStream_t stream, const Fetch *items, int count,
                          uint64_t *out);

inline uint64_t blocksfor(uint64_t items)
{
  return (items + VSA_BLOCK - 1) / VSA_BLOCK;
}

inline dim3 gridfor(uint64_t items)
{
  return vsa_grid(blocksfor(items));
}

struct KeepToU32
{
  __device__ uint32_t operator()(uint8_t k) const
  {
    return k;
  }
};

// out[] = the records of in[] with keep != 0, in order; *nkept (device) = count
VSA_HIDDEN int compact_matches(const vsa_match *in, const uint8_t *keep,
                               uint64_t count, vsa_match *out, uint64_t *nkept,
                               hipStream_t stream);

VSA_HIDDEN int sumlengths(const vsa_match *matches, uint64_t n,
                          hipStream_t stream, uint64_t *result);

inline unsigned int bitsfor(uint64_t maxvalue)
{
  unsigned int b = 1;
  while (b < 64 && (maxvalue >> b) != 0)
  {
    b++;
  }
  return b;
}

// out[t] = in[order[t]]
VSA_HIDDEN hipError_t gather_matches(const vsa_match *in,
                                     const uint32_t *order, uint64_t n,
                                     vsa_match *out, hipStream_t stream);

// stable sort of (key, match) pairs by key bits [0, endbit); results land in
// keys_out / matches_out.  The 32-byte records do not travel through the
// radix passes: (key, index) pairs do, and one gather follows.
VSA_HIDDEN int sortbykey(uint64_t *keys_in, uint64_t *keys_out, vsa_match *in,
                         vsa_match *out, uint64_t n, unsigned int endbit,
                         hipStream_t stream);

VSA_HIDDEN vsa_result *newresult(int device);

// offsets[sh] = sum of the fill counts of the cursor regions before sh (one
// cursor per VSA_CURSOR_STRIDE words); summary = {total, largest count, 0, 0}
VSA_HIDDEN hipError_t shard_summary(const unsigned long long *cursors,
                                    uint32_t nshards, uint64_t *offsets,
                                    uint64_t *summary, hipStream_t stream);

struct U32ToU64
{
  __device__ uint64_t operator()(uint32_t v) const
  {
    return v;
  }
};

#endif
