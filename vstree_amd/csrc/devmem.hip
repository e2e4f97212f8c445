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

struct Block
{
  size_t cls;
  int device;
};

std::mutex g_lock;
std::unordered_map<void *, Block> g_live;                   // handed out
std::map<std::pair<int, size_t>, std::vector<void *>> g_free; // cached
size_t g_cached = 0;

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

int vsa_dev_alloc(void **ptr, size_t bytes)
{
  int device = 0;
  VSA_HIP(hipGetDevice(&device));
  const size_t cls = sizeclass(bytes);
  {
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_free.find(std::make_pair(device, cls));
    if (it != g_free.end() && !it->second.empty())
    {
      *ptr = it->second.back();
      it->second.pop_back();
      g_cached -= cls;
      g_live[*ptr] = Block{cls, device};
      return 0;
    }
  }
  hipError_t e = hipMalloc(ptr, cls);
  if (e != hipSuccess)
  {
    // give cached memory back and try once more
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
  g_live[*ptr] = Block{cls, device};
  return 0;
}

void vsa_dev_free(void *ptr)
{
  if (ptr == nullptr)
  {
    return;
  }
  Block b;
  {
    std::lock_guard<std::mutex> g(g_lock);
    auto it = g_live.find(ptr);
    if (it == g_live.end())
    {
      // not ours (allocated with plain hipMalloc)
      (void) hipFree(ptr);
      return;
    }
    b = it->second;
    g_live.erase(it);
    if (b.cls <= kMaxCachedBlock && g_cached + b.cls <= kMaxCachedTotal)
    {
      g_free[std::make_pair(b.device, b.cls)].push_back(ptr);
      g_cached += b.cls;
      return;
    }
  }
  (void) hipFree(ptr);
}

void vsa_dev_trim()
{
  std::vector<void *> all;
  {
    std::lock_guard<std::mutex> g(g_lock);
    for (auto &kv : g_free)
    {
      for (void *p : kv.second)
      {
        all.push_back(p);
      }
      kv.second.clear();
    }
    g_cached = 0;
  }
  for (void *p : all)
  {
    (void) hipFree(p);
  }
}

extern "C" int vsa_device_trim(int device)
{
  (void) device;
  vsa_dev_trim();
  return 0;
}
