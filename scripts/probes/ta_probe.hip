// What bounds a wavefront of divergent loads on MI355X?  Random reads of a
// table far beyond every cache, one lane = one address, in several shapes:
//   width    4 / 8 / 16 bytes per lane, and 2 x 16 bytes from one 32-byte slot
//   inflight independent loads issued before the first is used
//   active   only every k-th lane takes part (partially filled wavefronts)
//   chain    the next address depends on the loaded value
// Prints G lane-loads/s and G wave-instructions/s.
//   hipcc --offload-arch=gfx950 -O3 -o _bin/ta_probe ta_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CK(x)                                                                 \
  do                                                                          \
  {                                                                           \
    hipError_t e_ = (x);                                                      \
    if (e_ != hipSuccess)                                                     \
    {                                                                         \
      fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_));                 \
      exit(1);                                                                \
    }                                                                         \
  } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  uint64_t z = x;
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

template <typename T> __device__ __forceinline__ uint64_t fold(T v);
template <> __device__ __forceinline__ uint64_t fold(uint32_t v) { return v; }
template <> __device__ __forceinline__ uint64_t fold(uint64_t v) { return v; }
template <> __device__ __forceinline__ uint64_t fold(uint4 v)
{
  return (uint64_t) v.x + v.y + v.z + v.w;
}

// T = element type (4, 8, 16 bytes); PAIR: a second load of the neighbouring
// element (same 2*sizeof(T) slot); STRIDE: every STRIDE-th lane is active
template <typename T, int INFLIGHT, bool PAIR, int STRIDE, bool CHAIN>
__global__ void __launch_bounds__(256)
k_rr(const T *__restrict__ buf, uint64_t nelem, uint32_t per,
     unsigned long long *sink)
{
  const uint64_t t = (uint64_t) blockIdx.x * 256 + threadIdx.x;
  if ((threadIdx.x % STRIDE) != 0)
  {
    return;
  }
  uint64_t x = t * (uint64_t) per * 0x9E3779B97F4A7C15ull + 1 /* disjoint sequences: see api.hip */, acc = 0;
  for (uint32_t i = 0; i < per; i += INFLIGHT)
  {
    T v[INFLIGHT], w[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++)
    {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t a = mix(CHAIN ? x + acc : x) % nelem;
      if (PAIR)
      {
        a &= ~1ull;
      }
      v[k] = buf[a];
      if (PAIR)
      {
        w[k] = buf[a + 1];
      }
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++)
    {
      acc += fold(v[k]);
      if (PAIR)
      {
        acc += fold(w[k]);
      }
    }
  }
  if (acc == 0x1234567ull)
  {
    atomicAdd(sink, 1ull);
  }
}

template <typename T, int INFLIGHT, bool PAIR, int STRIDE, bool CHAIN>
static void run(const char *name, void *buf, uint64_t bytes,
                unsigned long long *sink, int wavespersimd)
{
  const uint64_t nelem = bytes / sizeof(T);
  const uint32_t per = 64;
  // wavespersimd resident waves: blocks of 256 = 4 waves = 1 per SIMD
  const unsigned int blocks = 256u * 8u * 4u; // 4 rounds of a full chip
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  (void) wavespersimd;
  k_rr<T, INFLIGHT, PAIR, STRIDE, CHAIN><<<blocks, 256>>>(
      (const T *) buf, nelem, per, sink);
  CK(hipEventRecord(a, 0));
  const int reps = 3;
  for (int r = 0; r < reps; r++)
  {
    k_rr<T, INFLIGHT, PAIR, STRIDE, CHAIN><<<blocks, 256>>>(
        (const T *) buf, nelem, per, sink);
  }
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  const double lanes = (double) blocks * 256 / STRIDE * per * reps,
               instr = (double) blocks * 4 * per * reps * (PAIR ? 2 : 1);
  printf("%-44s %7.1f G lane-loads/s %7.2f G wave-instr/s  %6.2f TB/s of "
         "64-B sectors\n",
         name, lanes / (ms * 1e-3) / 1e9, instr / (ms * 1e-3) / 1e9,
         lanes * 64 / (ms * 1e-3) / 1e12);
  fflush(stdout);
}

int main(int argc, char **argv)
{
  const uint64_t bytes = (uint64_t) (argc > 1 ? atof(argv[1]) : 64e9);
  void *buf;
  unsigned long long *sink;
  CK(hipMalloc(&buf, bytes + 64));
  CK(hipMalloc((void **) &sink, 8));
  CK(hipMemset(buf, 1, bytes));
  CK(hipMemset(sink, 0, 8));
  printf("table %.0f GB\n", bytes / 1e9);
#define R(T, I, P, S, C) run<T, I, P, S, C>(#T " inflight " #I " pair " #P " stride " #S " chain " #C, buf, bytes, sink, 8)
  R(uint32_t, 1, false, 1, false);
  R(uint64_t, 1, false, 1, false);
  R(uint4, 1, false, 1, false);
  R(uint4, 1, true, 1, false);
  R(uint32_t, 4, false, 1, false);
  R(uint64_t, 4, false, 1, false);
  R(uint4, 4, false, 1, false);
  R(uint4, 4, true, 1, false);
  R(uint4, 2, false, 1, false);
  R(uint4, 1, false, 2, false);
  R(uint4, 1, false, 4, false);
  R(uint4, 1, false, 16, false);
  R(uint4, 4, false, 4, false);
  R(uint4, 4, false, 16, false);
  R(uint64_t, 1, false, 1, true);
  R(uint4, 1, false, 1, true);
  R(uint4, 1, true, 1, true);
  return 0;
}
