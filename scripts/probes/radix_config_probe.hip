// Probe: rocprim::radix_sort_pairs of 11.6 M (uint64 key, uint64 value) pairs
// over 39 key bits with the default onesweep configuration (8 bits per pass,
// 5 passes) against wider digits (fewer passes).
//   hipcc --offload-arch=gfx950 -O3 -o radix_probe radix_config_probe.hip
#include <cstring>
#include <hip/hip_runtime.h>
#include <rocprim/rocprim.hpp>
#include <cstdio>
#include <cstdint>
#include <vector>

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_fill(uint64_t *k, uint64_t *v, uint64_t n)
{
  const uint64_t i = (uint64_t) blockIdx.x * 256 + threadIdx.x;
  if (i < n)
  {
    uint64_t z = i * 0x9E3779B97F4A7C15ull + 12345;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    z ^= z >> 31;
    k[i] = z & ((1ull << 39) - 1);
    v[i] = i;
  }
}

template <class Config>
int run(const char *name, uint64_t *k1, uint64_t *k2, uint64_t *v1, uint64_t *v2, size_t n, std::vector<uint64_t> &ref)
{
  size_t tb = 0;
  void *temp = nullptr;
  CK(rocprim::radix_sort_pairs<Config>(nullptr, tb, k1, k2, v1, v2, n, 0u, 39u, 0));
  CK(hipMalloc(&temp, tb));
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  for (int w = 0; w < 2; w++)
    CK(rocprim::radix_sort_pairs<Config>(temp, tb, k1, k2, v1, v2, n, 0u, 39u, 0));
  CK(hipEventRecord(a, 0));
  for (int w = 0; w < 10; w++)
    CK(rocprim::radix_sort_pairs<Config>(temp, tb, k1, k2, v1, v2, n, 0u, 39u, 0));
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  std::vector<uint64_t> h(n);
  CK(hipMemcpy(h.data(), k2, n * 8, hipMemcpyDeviceToHost));
  bool ok = true;
  if (ref.empty())
    ref = h;
  else
    ok = ref == h;
  for (size_t i = 1; i < n && ok; i++)
    ok = h[i - 1] <= h[i];
  printf("%-28s %.1f us per sort, temp %zu bytes, %s\n", name, ms * 100.0, tb, ok ? "sorted, same as default" : "WRONG");
  CK(hipFree(temp));
  return 0;
}

int main()
{
  const size_t n = 11583115;
  uint64_t *k1, *k2, *v1, *v2;
  CK(hipMalloc(&k1, n * 8));
  CK(hipMalloc(&k2, n * 8));
  CK(hipMalloc(&v1, n * 8));
  CK(hipMalloc(&v2, n * 8));
  k_fill<<<(n + 255) / 256, 256>>>(k1, v1, n);
  CK(hipDeviceSynchronize());
  std::vector<uint64_t> ref;
  using rocprim::default_config;
  using rocprim::kernel_config;
  constexpr auto algo = rocprim::block_radix_rank_algorithm::match;
  if (run<default_config>("default (8 bits)", k1, k2, v1, v2, n, ref)) return 1;
#define TRY(BITS, BS, IPT)                                                              \
  if (run<rocprim::radix_sort_config<default_config, default_config,                    \
          rocprim::radix_sort_onesweep_config<kernel_config<BS, IPT>, kernel_config<BS, IPT>, BITS, algo>>>( \
          #BITS " bits, " #BS " x " #IPT, k1, k2, v1, v2, n, ref)) return 1;
  TRY(8, 1024, 8)
  TRY(8, 512, 12)
  TRY(10, 1024, 8)
  TRY(10, 512, 12)
  TRY(10, 1024, 4)
  return 0;
}
