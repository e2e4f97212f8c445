// Device-side generators of the synthetic inputs (same numbers as synth.c):
// the splitmix64 state is a counter, so element i is computed directly.
#include "vsa_internal.hpp"

__device__ __forceinline__ uint64_t sm64_mix(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

// 16 bases per work-item, one 128-bit store
__global__ void __launch_bounds__(256)
k_synth_genome(uint64_t seed, uint64_t n, uint8_t *__restrict__ codes)
{
  const uint64_t i0 = (vsa_bid() * 256 + threadIdx.x) * 16;
  if (i0 >= n)
  {
    return;
  }
  uint8_t b[16];
#pragma unroll
  for (int k = 0; k < 16; k++)
  {
    b[k] = (uint8_t) (sm64_mix(seed + (i0 + k + 1) * 0x9E3779B97F4A7C15ull)
                      >> 62);
  }
  if (i0 + 16 <= n)
  {
    uint4 v;
    __builtin_memcpy(&v, b, 16);
    *reinterpret_cast<uint4 *>(codes + i0) = v;
  } else
  {
    for (uint64_t k = 0; i0 + k < n; k++)
    {
      codes[i0 + k] = b[k];
    }
  }
}

__global__ void __launch_bounds__(256)
k_synth_queries(const uint8_t *__restrict__ genome,
                const uint64_t *__restrict__ pos,
                const uint32_t *__restrict__ substidx,
                const uint8_t *__restrict__ step, uint64_t nq, uint32_t m,
                uint8_t *__restrict__ queries)
{
  const uint64_t t = vsa_bid() * 256 + threadIdx.x;
  if (t >= nq * (uint64_t) m)
  {
    return;
  }
  const uint64_t q = t / m;
  const uint32_t j = (uint32_t) (t - q * m);
  uint8_t c = genome[pos[q] + j];
  if (substidx[q] == j)
  {
    c = (uint8_t) ((c + step[q]) & 3);
  }
  queries[t] = c;
}

extern "C" int vsa_synth_genome_device(uint64_t seed, uint64_t n,
                                       void *device_codes, int device)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  if (n == 0)
  {
    return 0;
  }
  const uint64_t items = (n + 15) / 16;
  k_synth_genome<<<vsa_grid((items + 255) / 256), 256>>>(
      seed, n, (uint8_t *) device_codes);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int vsa_synth_queries_device(const void *device_genome, uint64_t n,
                                        const uint64_t *pos,
                                        const uint32_t *substidx,
                                        const uint8_t *step, uint64_t nq,
                                        uint32_t m, void *device_queries,
                                        int device)
{
  (void) n;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  if (nq == 0)
  {
    return 0;
  }
  uint64_t *dpos = nullptr;
  uint32_t *dsub = nullptr;
  uint8_t *dstep = nullptr;
  VSA_HIP(hipMalloc((void **) &dpos, nq * 8));
  VSA_HIP(hipMalloc((void **) &dsub, nq * 4));
  VSA_HIP(hipMalloc((void **) &dstep, nq));
  VSA_HIP(hipMemcpy(dpos, pos, nq * 8, hipMemcpyHostToDevice));
  VSA_HIP(hipMemcpy(dsub, substidx, nq * 4, hipMemcpyHostToDevice));
  VSA_HIP(hipMemcpy(dstep, step, nq, hipMemcpyHostToDevice));
  const uint64_t items = nq * (uint64_t) m;
  k_synth_queries<<<vsa_grid((items + 255) / 256), 256>>>(
      (const uint8_t *) device_genome, dpos, dsub, dstep, nq, m,
      (uint8_t *) device_queries);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipDeviceSynchronize());
  (void) hipFree(dpos);
  (void) hipFree(dsub);
  (void) hipFree(dstep);
  return 0;
}
