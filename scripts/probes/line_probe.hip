// What is the unit HBM charges a random read by on MI355X -- the 64-byte
// sector the L2 counts, or the 128-byte line a streaming read asks for?
// ("What's weak" 1: 3.1 TB/s of random 64-byte sectors is half of
// the 6.2 TB/s streaming rate; 32-byte slots cut sectors by 20 % for 0 % time.)
//
// Every lane draws a random record of `align` bytes out of a table far beyond
// every cache and issues NL 16-byte loads at phase, phase + delta, ... inside
// or behind it.  Rows of the output: records/s, 64-B sectors/s, 128-B lines/s.
//
//   line_probe GB [mode]     mode: all | sizes | one:<NL>:<align>:<phase>:<delta>
//   hipcc --offload-arch=gfx950 -O3 -o _bin/line_probe line_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <set>

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

// NL loads of 16 bytes per record; the record index is random per lane and
// step, disjoint sequences per lane (a probe whose lanes shared their
// sequences measured the L2, profiles/r03/README.md)
template <int NL>
__global__ void __launch_bounds__(256)
k_line(const uint8_t *__restrict__ buf, uint64_t nrec, uint32_t align,
       uint32_t phase, uint32_t delta, uint32_t per,
       unsigned long long *sink)
{
  const uint64_t t = (uint64_t) blockIdx.x * 256 + threadIdx.x;
  uint64_t x = t * (uint64_t) per * 0x9E3779B97F4A7C15ull + 1, acc = 0;
  for (uint32_t i = 0; i < per; i++)
  {
    x += 0x9E3779B97F4A7C15ull;
    const uint64_t a = (mix(x) % nrec) * align + phase;
    uint4 v[NL];
#pragma unroll
    for (int k = 0; k < NL; k++)
    {
      v[k] = *(const uint4 *) (buf + a + (uint64_t) k * delta);
    }
#pragma unroll
    for (int k = 0; k < NL; k++)
    {
      acc += (uint64_t) v[k].x + v[k].y + v[k].z + v[k].w;
    }
  }
  if (acc == 0x1234567ull)
  {
    atomicAdd(sink, 1ull);
  }
}

// a record read by a GROUP of G neighbouring lanes, 16 bytes each (G * 16
// contiguous bytes): what a cooperative fetch of a 64/128/256-byte record costs
template <int G>
__global__ void __launch_bounds__(256)
k_group(const uint8_t *__restrict__ buf, uint64_t nrec, uint32_t align,
        uint32_t per, unsigned long long *sink)
{
  const uint64_t t = ((uint64_t) blockIdx.x * 256 + threadIdx.x) / G;
  const uint32_t sub = threadIdx.x % G;
  uint64_t x = t * (uint64_t) per * 0x9E3779B97F4A7C15ull + 1, acc = 0;
  for (uint32_t i = 0; i < per; i++)
  {
    x += 0x9E3779B97F4A7C15ull;
    const uint64_t a = (mix(x) % nrec) * align + sub * 16;
    const uint4 v = *(const uint4 *) (buf + a);
    acc += (uint64_t) v.x + v.y + v.z + v.w;
  }
  if (acc == 0x1234567ull)
  {
    atomicAdd(sink, 1ull);
  }
}

static void *g_buf;
static unsigned long long *g_sink;

static double timeit(void (*launch)(void *), void *arg)
{
  hipEvent_t a, b;
  CK(hipEventCreate(&a));
  CK(hipEventCreate(&b));
  launch(arg);
  CK(hipEventRecord(a, 0));
  const int reps = 3;
  for (int r = 0; r < reps; r++)
  {
    launch(arg);
  }
  CK(hipEventRecord(b, 0));
  CK(hipEventSynchronize(b));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, a, b));
  CK(hipEventDestroy(a));
  CK(hipEventDestroy(b));
  return ms * 1e-3 / reps;
}

struct Shape
{
  int nl;
  uint64_t bytes;
  uint32_t align, phase, delta;
};
static const unsigned int kBlocks = 256u * 8u * 4u;
static const uint32_t kPer = 64;

static void launch_line(void *p)
{
  const Shape *s = (const Shape *) p;
  // the last record keeps its loads inside the table
  const uint64_t span = s->phase + (uint64_t) (s->nl - 1) * s->delta + 16;
  const uint64_t nrec = (s->bytes - span) / s->align;
  const uint8_t *b = (const uint8_t *) g_buf;
#define L(N)                                                                  \
  case N:                                                                     \
    k_line<N><<<kBlocks, 256>>>(b, nrec, s->align, s->phase, s->delta, kPer,  \
                                g_sink);                                      \
    break
  switch (s->nl)
  {
    L(1);
    L(2);
    L(3);
    L(4);
    L(8);
    default: fprintf(stderr, "NL %d not instantiated\n", s->nl); exit(1);
  }
#undef L
}

static void row(const char *what, Shape s)
{
  const double sec = timeit(launch_line, &s);
  const double recs = (double) kBlocks * 256 * kPer;
  // distinct 64-byte sectors and 128-byte lines one record touches
  std::set<uint64_t> sect, line;
  for (int k = 0; k < s.nl; k++)
  {
    const uint64_t a = s.phase + (uint64_t) k * s.delta;
    sect.insert(a / 64);
    sect.insert((a + 15) / 64);
    line.insert(a / 128);
    line.insert((a + 15) / 128);
  }
  printf("%-58s NL %d align %7u phase %3u delta %8u : %6.1f G rec/s %6.1f G "
         "sect/s %6.1f G lines/s %6.2f TB/s(sect)\n",
         what, s.nl, s.align, s.phase, s.delta, recs / sec / 1e9,
         recs * sect.size() / sec / 1e9, recs * line.size() / sec / 1e9,
         recs * sect.size() * 64 / sec / 1e12);
  fflush(stdout);
}

struct GShape
{
  int g;
  uint64_t bytes;
  uint32_t align;
};
static void launch_group(void *p)
{
  const GShape *s = (const GShape *) p;
  const uint64_t nrec = s->bytes / s->align - 1;
  const uint8_t *b = (const uint8_t *) g_buf;
  switch (s->g)
  {
    case 2: k_group<2><<<kBlocks, 256>>>(b, nrec, s->align, kPer, g_sink); break;
    case 4: k_group<4><<<kBlocks, 256>>>(b, nrec, s->align, kPer, g_sink); break;
    case 8: k_group<8><<<kBlocks, 256>>>(b, nrec, s->align, kPer, g_sink); break;
    case 16: k_group<16><<<kBlocks, 256>>>(b, nrec, s->align, kPer, g_sink); break;
    default: exit(1);
  }
}
static void grow(const char *what, GShape s)
{
  const double sec = timeit(launch_group, &s);
  const double recs = (double) kBlocks * 256 * kPer / s.g;
  const double bytes = recs * s.g * 16;
  printf("%-58s group of %2d lanes, record %4u B      : %6.1f G rec/s %6.1f G "
         "sect/s %6.1f G lines/s %6.2f TB/s\n",
         what, s.g, s.align, recs / sec / 1e9, bytes / 64 / sec / 1e9,
         recs * ((s.g * 16 + 127) / 128) / sec / 1e9, bytes / sec / 1e12);
  fflush(stdout);
}

int main(int argc, char **argv)
{
  const double gb = argc > 1 ? atof(argv[1]) : 64.0;
  const char *mode = argc > 2 ? argv[2] : "all";
  const uint64_t bytes = (uint64_t) (gb * 1e9) & ~4095ull;
  CK(hipMalloc(&g_buf, bytes + 4096));
  CK(hipMalloc((void **) &g_sink, 8));
  CK(hipMemset(g_buf, 1, bytes + 4096));
  CK(hipMemset(g_sink, 0, 8));
  printf("table %.2f GB, %u workgroups x 256 lanes x %u records\n", bytes / 1e9,
         kBlocks, kPer);
  if (strncmp(mode, "one:", 4) == 0)
  {
    Shape s;
    s.bytes = bytes;
    if (sscanf(mode + 4, "%d:%u:%u:%u", &s.nl, &s.align, &s.phase,
               &s.delta) != 4)
    {
      fprintf(stderr, "one:<NL>:<align>:<phase>:<delta>\n");
      return 1;
    }
    row("one shape", s);
    return 0;
  }
  if (strcmp(mode, "sizes") == 0)
  {
    // rate against the size of the table (same allocation, a prefix of it)
    const double sizes[] = {0.064, 0.128, 0.256, 0.512, 1, 2, 4, 8, 16, 32, 64,
                            128, 200};
    for (double g : sizes)
    {
      const uint64_t b = (uint64_t) (g * 1e9) & ~4095ull;
      if (b > bytes)
      {
        break;
      }
      char name[96];
      snprintf(name, sizeof name, "prefix of %.3f GB: one 16-byte load", g);
      row(name, Shape{1, b, 64, 0, 0});
      snprintf(name, sizeof name, "prefix of %.3f GB: two sectors, one line", g);
      row(name, Shape{2, b, 128, 0, 64});
    }
    return 0;
  }
  row("one 16-byte load per record (the r3 probe)", Shape{1, bytes, 64, 0, 0});
  row("two loads, same 64-byte sector", Shape{2, bytes, 64, 0, 16});
  row("two sectors of ONE aligned 128-byte line", Shape{2, bytes, 128, 0, 64});
  row("two adjacent sectors of TWO lines (64..191)", Shape{2, bytes, 128, 64, 64});
  row("two sectors 128 B apart (aligned 256-B block)", Shape{2, bytes, 256, 0, 128});
  row("two sectors 128 B apart (lines 1 and 2 of a 256 block)", Shape{2, bytes, 256, 128, 128});
  row("two sectors 256 B apart", Shape{2, bytes, 512, 0, 256});
  row("two sectors 512 B apart", Shape{2, bytes, 1024, 0, 512});
  row("two sectors 1 KB apart", Shape{2, bytes, 2048, 0, 1024});
  row("two sectors 2 KB apart", Shape{2, bytes, 4096, 0, 2048});
  row("two sectors 4 KB apart", Shape{2, bytes, 8192, 0, 4096});
  row("two sectors 64 KB apart", Shape{2, bytes, 131072, 0, 65536});
  row("two sectors 2 MB apart", Shape{2, bytes, 4194304, 0, 2097152});
  row("two sectors 1 GB apart", Shape{2, bytes, 64, 0, 1u << 30});
  row("three sectors of a 256-B block (0, 64, 128)", Shape{3, bytes, 256, 0, 64});
  row("four sectors = two lines = one 256-B block", Shape{4, bytes, 256, 0, 64});
  row("four loads inside one 128-B line (32 B apart)", Shape{4, bytes, 128, 0, 32});
  row("a whole 128-B line by one lane (8 loads)", Shape{8, bytes, 128, 0, 16});
  row("eight sectors = one aligned 512-B block", Shape{8, bytes, 512, 0, 64});
  grow("64-B record by 4 lanes", GShape{4, bytes, 64});
  grow("128-B record by 8 lanes", GShape{8, bytes, 128});
  grow("256-B record by 16 lanes", GShape{16, bytes, 256});
  grow("128-B record at 64-B alignment by 8 lanes (straddles)", GShape{8, bytes, 192});
  return 0;
}
