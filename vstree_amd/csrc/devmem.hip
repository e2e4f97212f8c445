// Device-memory recycling for the temporaries and result lists of the query
// path.  hipMalloc / hipFree cost 50-100 us each and hipFree synchronises the
// device; one -mum step used ~30 of them, i.e. milliseconds of a 20 ms step.
// Blocks are binned by size class and handed out again; big blocks (index
// construction) and anything beyond the cache budget go straight back to HIP.
#include "vsa_internal.hpp"
#include <map>
#include <mutex>
#include <unordered_map>

namespace
{

// 288 GB of HBM, 114 GB of it the 3 Gbp index: keeping the temporaries of
// the largest batches costs nothing; giving them back costs a device
// synchronisation per block (vsa_dev_alloc trims the cache and retries when
// an allocation fails)
const size_t kMaxCachedBlock = 8ull << 30;  // do not keep blocks above 8 GiB
const size_t kMaxCachedTotal = 48ull << 30; // per process

// Reuse is stream ordered.  A block remembers the stream its user works on
// (vsa_dev_set_stream, set by every pipeline entry for the calling thread).
// Handing a freed block to a user on the same stream needs nothing (the new
// work is queued behind the old); handing it to another stream records an
// event on the old stream at that moment -- everything queued there before the
// block was freed lies in front of it -- and makes the new stream wait for it.
// Freeing a block therefore costs no HIP call (a -mum step frees ~45 blocks;
// an event per free was 100 us of host time between two steps), and a DevBuf
// destructor may run while kernels that touch the block are still queued --
// the early exits of the pipelines (VSA_HIP returns) do exactly that.
struct Block
{
  size_t cls;
  int device;
  hipStream_t stream;
};

struct Cached
{
  void *ptr;
  hipStream_t stream;
  bool pending; // false: nothing queued can touch it (stream synchronised)
};

std::mutex g_lock;
std::unordered_map<void *, Block> g_live;                   // handed out
std::map<std::pair<int, size_t>, std::vector<Cached>> g_free; // cached
// spare events, per device: an event belongs to the device it was made on
// (one thread per GPU in vsa_multi_*: an event of device A recorded on a stream
// of device B fails, and the fallback is a device-wide wait)
std::map<int, std::vector<hipEvent_t>> g_events;
size_t g_cached = 0;
thread_local hipStream_t t_stream = nullptr;

size_t sizeclass(size_t bytes)
{
  if (bytes < 256)
  {
    return 256;
  }
  if (bytes <= (1u << 20))
  {
    size_t c = 256;
    while (c < bytes)
    {
      c <<= 1;
    }
    return c;
  }
  // above 1 MiB: steps of 1/8 of the next lower power of two
  size_t p = 1u << 20;
  while ((p << 1) <= bytes)
  {
    p <<= 1;
  }
  const size_t step = p >> 3;
  return (bytes + step - 1) / step * step;
}

} // namespace

void vsa_dev_set_stream(hipStream_t stream)
{
  t_stream = stream;
}

// the stream is about to be destroyed (vsa_index_close): finish its work and
// detach every block from it
void vsa_dev_forget_stream(hipStream_t stream)
{
  if (stream == nullptr)
  {
    return;
  }
  (void) hipStreamSynchronize(stream);
  if (t_stream == stream)
  {
    t_stream = nullptr;
  }
  std::lock_guard<std::mutex> g(g_lock);
  for (auto &kv : g_live)
  {
    if (kv.second.stream == stream)
    {
      kv.second.stream = nullptr;
    }
  }
  for (auto &kv : g_free)
  {
    for (Cached &c : kv.second)
    {
      if (c.stream == stream)
      {
        c.stream = nullptr;
        c.pending = false; // synchronised above
      }
    }
  }
}

int vsa_dev_alloc(void **ptr, size_t bytes)
{
  int device = 0;
  VSA_HIP(hipGetDevice(&device));
  const size_t cls = sizeclass(bytes);
  {
    std::unique_lock<std::mutex> g(g_lock);
    auto it = g_free.find(std::make_pair(device, cls));
    if (it != g_free.end() && !it->second.empty())
    {
      const Cached c = it->second.back();
      it->second.pop_back();
      g_cached -= cls;
      g_live[c.ptr] = Block{cls, device, t_stream};
      g.unlock();
      *ptr = c.ptr;
      if (c.pending && c.stream != t_stream)
      {
        // what is queued on the block's old stream may still use it
        hipEvent_t ev = nullptr;
        {
          std::lock_guard<std::mutex> g2(g_lock);
          std::vector<hipEvent_t> &spare = g_events[device];
          if (!spare.empty())
          {
            ev = spare.back();
            spare.pop_back();
          }
        }
        if (ev == nullptr &&
            hipEventCreateWithFlags(&ev, hipEventDisableTiming) != hipSuccess)
        {
          ev = nullptr;
        }
        if (ev == nullptr || hipEventRecord(ev, c.stream) != hipSuccess ||
            hipStreamWaitEvent(t_stream, ev, 0) != hipSuccess)
        {
          // no event to be had, or the stream is gone: wait for the device
          (void) hipGetLastError();
          (void) hipDeviceSynchronize();
        }
        if (ev != nullptr)
        {
          std::lock_guard<std::mutex> g2(g_lock);
          g_events[device].push_back(ev); // the wait holds its own reference
        }
      }
      return 0;
    }
  }
  hipError_t e = hipMalloc(ptr, cls);
  if (e != hipSuccess)
  {
    // give cached memory back and try once more
    (void) hipGetLastError();
    vsa_dev_trim();
    e = hipMalloc(ptr, cls);
  }
  if (e != hipSuccess)
  {
    VSA_ERROR("hipMalloc of %lu bytes failed: %s", (unsigned long) cls,
              hipGetErrorString(e));
    *ptr = nullptr;
    return -100;
  }
  std::lock_guard<std::mutex> g(g_lock);
  g_live[*ptr] = Block{cls, device, t_stream};
  return 0;
}

// Long-lived tables (index, query batches): plain hipMalloc, but memory the
// cache holds is given back before an allocation is declared impossible.
hipError_t vsa_hip_malloc(void **ptr, size_t bytes)
{
  hipError_t e = hipMalloc(ptr, bytes);
  if (e == hipErrorOutOfMemory)
  {
    (void) hipGetLastError();
    vsa_dev_trim();
    e = hipMalloc(ptr, bytes);
  }
  return e;
}

void vsa_dev_free(void *ptr)
{
  if (ptr == nullptr)
  {
    return;
  }
  {
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_live.find(ptr);
    if (it != g_live.end())
    {
      const Block b = it->second;
      g_live.erase(it);
      if (b.cls <= kMaxCachedBlock && g_cached + b.cls <= kMaxCachedTotal)
      {
        g_free[std::make_pair(b.device, b.cls)].push_back(
            Cached{ptr, b.stream, true});
        g_cached += b.cls;
        return;
      }
    }
    // not ours (allocated with plain hipMalloc), too large or the cache is
    // full: back to HIP -- outside the lock, hipFree waits for the device and
    // the replica threads of the other GPUs must not wait with it
  }
  (void) hipFree(ptr);
}

void vsa_dev_trim()
{
  std::vector<Cached> all;
  {
    std::lock_guard<std::mutex> g(g_lock);
    for (auto &kv : g_free)
    {
      for (const Cached &c : kv.second)
      {
        all.push_back(c);
      }
      kv.second.clear();
    }
    g_cached = 0;
  }
  for (const Cached &c : all)
  {
    (void) hipFree(c.ptr); // synchronises: pending work is over afterwards
  }
}

extern "C" int vsa_device_trim(int device)
{
  (void) device;
  vsa_dev_trim();
  return 0;
}

extern "C" int vsa_device_meminfo(int device, uint64_t *freebytes,
                                  uint64_t *totalbytes)
{
  size_t f = 0, t = 0;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  VSA_HIP(hipMemGetInfo(&f, &t));
  if (freebytes != nullptr)
  {
    *freebytes = f;
  }
  if (totalbytes != nullptr)
  {
    *totalbytes = t;
  }
  return 0;
}
