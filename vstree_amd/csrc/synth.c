/*
  Deterministic synthetic DNA for the parity tests and for bench.py.

  Generator: splitmix64; genome seed 42, query seed 4242 (SURVEY.md section 8d,
  BASELINE.md section 3).  Genome base i = next() >> 62.  Query i: p = next()
  % (n-m+1), copy G[p..p+m); if next() % 4 == 0 then k = next() % m and base
  k becomes (old + 1 + next() % 3) % 4.  All outputs are alphabet-mapped
  symbols (a,c,g,t = 0..3, the coding mkvtree -dna uses, see
  /root/reference/src/kurtz-basic/alphabet.c:369).
*/
#include <stdint.h>
#include <stddef.h>
#include "vstree_amd.h"

#define SM64_GAMMA 0x9E3779B97F4A7C15ULL

static inline uint64_t sm64_mix(uint64_t z)
{
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
  return z ^ (z >> 31);
}

/* value returned by the (idx+1)-th call of next() after seeding with seed:
   the state is a plain counter, so any element is addressable directly */
uint64_t vsa_splitmix64_at(uint64_t seed, uint64_t idx)
{
  return sm64_mix(seed + (idx + 1) * SM64_GAMMA);
}

void vsa_synth_genome(uint64_t seed, uint64_t n, uint8_t *codes)
{
  uint64_t i, x = seed;

  for (i = 0; i < n; i++)
  {
    x += SM64_GAMMA;
    codes[i] = (uint8_t) (sm64_mix(x) >> 62);
  }
}

/*
  Draws the plan of nq queries of length m over a genome of length n without
  touching the genome: start position, index of the substituted base (or
  VSA_NO_SUBST) and the substitution step 1..3.  The device-side generator
  (synth_kernels.hip) and vsa_synth_queries below materialise the same plan.
*/
void vsa_synth_query_plan(uint64_t seed, uint64_t n, uint64_t nq, uint32_t m,
                          uint64_t *pos, uint32_t *substidx, uint8_t *step)
{
  uint64_t i, x = seed;

  for (i = 0; i < nq; i++)
  {
    x += SM64_GAMMA;
    pos[i] = sm64_mix(x) % (n - m + 1);
    x += SM64_GAMMA;
    if (sm64_mix(x) % 4 == 0)
    {
      x += SM64_GAMMA;
      substidx[i] = (uint32_t) (sm64_mix(x) % m);
      x += SM64_GAMMA;
      step[i] = (uint8_t) (1 + sm64_mix(x) % 3);
    } else
    {
      substidx[i] = VSA_NO_SUBST;
      step[i] = 0;
    }
  }
}

/* queries[i*m .. i*m+m) = query i; srcpos may be NULL */
void vsa_synth_queries(uint64_t seed, const uint8_t *genome, uint64_t n,
                       uint64_t nq, uint32_t m, uint8_t *queries,
                       uint64_t *srcpos)
{
  uint64_t i, x = seed;
  uint32_t j;

  for (i = 0; i < nq; i++)
  {
    uint64_t p;
    uint8_t *q = queries + i * (uint64_t) m;

    x += SM64_GAMMA;
    p = sm64_mix(x) % (n - m + 1);
    for (j = 0; j < m; j++)
    {
      q[j] = genome[p + j];
    }
    x += SM64_GAMMA;
    if (sm64_mix(x) % 4 == 0)
    {
      uint64_t k;

      x += SM64_GAMMA;
      k = sm64_mix(x) % m;
      x += SM64_GAMMA;
      q[k] = (uint8_t) ((q[k] + 1 + sm64_mix(x) % 3) % 4);
    }
    if (srcpos != NULL)
    {
      srcpos[i] = p;
    }
  }
}
