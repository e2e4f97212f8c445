// The derived search tables of an index (esa8, slot16, tis2 / spec64: DESIGN.md
// section 3), made on the device from the reference's tables.
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

// ---------------------------------------------------------------------------
// the keyed search array (see DevIndex::esa8)
// ---------------------------------------------------------------------------

template <typename IDX>
__global__ void __launch_bounds__(VSA_BLOCK)
k_make_esa8(const uint8_t *__restrict__ tis, const IDX *__restrict__ suf,
            const uint8_t *__restrict__ lcp, uint64_t count, uint32_t D,
            uint64_t *__restrict__ esa8)
{
  const uint64_t j = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (j >= count)
  {
    return;
  }
  const uint64_t s = suf[j];
  const uint8_t *t = tis + s + D; // padded with 0xFF behind n
  uint64_t key = 0, flag = 0;
#pragma unroll
  for (uint32_t k = 0; k < VSA_KEYSYMS; k++)
  {
    const uint8_t a = t[k];
    if (VSA_ISSPECIAL(a))
    {
      flag = VSA_KEYFLAG;
    }
    key = (key << 2) | (a & 3);
  }
  // front pad of the text = separator: suffix 0 has nothing in front
  const uint8_t l = tis[(int64_t) s - 1];
  const uint64_t left = VSA_ISSPECIAL(l) ? VSA_LEFTSPECIAL
                                         : ((uint64_t) (l & 3) << VSA_LEFTSHIFT);
  // (of a wide suf only the low half: vsa_entrystart reads suf itself then)
  esa8[j] = (s & 0xFFFFFFFFull) | ((uint64_t) lcp[j] << 32) |
            (key << VSA_KEYSHIFT) | flag | left;
}

// tis2 / spec64 / firstspecial (see DevIndex): one work-item packs a block of
// 64 text positions into 16 bytes; the wavefront's ballot is 8 bytes of the
// block bitmap.  Positions >= n count as special.
__global__ void __launch_bounds__(VSA_BLOCK)
k_pack_text(const uint8_t *__restrict__ tis, uint64_t n, uint64_t nblocks,
            uint8_t *__restrict__ tis2, uint8_t *__restrict__ spec64,
            unsigned long long *__restrict__ firstspecial)
{
  const uint64_t b = vsa_bid() * VSA_BLOCK + threadIdx.x;
  bool special = false;
  if (b < nblocks)
  {
    const uint8_t *p = tis + 64 * b; // 0xFF behind position n
    uint64_t out[2] = {0, 0}, firstbad = ~0ull;
#pragma unroll
    for (int k = 0; k < 4; k++)
    {
      const vsa_u128 v = vsa_load16(p + 16 * k);
      const uint64_t notdna = 0xFCFCFCFCFCFCFCFCull;
      const uint64_t s0 = v.lo & notdna, s1 = v.hi & notdna;
      if (firstbad == ~0ull && (s0 | s1) != 0)
      {
        firstbad = 64 * b + 16 * k +
                   (s0 != 0 ? ((uint64_t) __builtin_ctzll(s0) >> 3)
                            : 8 + ((uint64_t) __builtin_ctzll(s1) >> 3));
      }
      const uint64_t packed = vsa_pack16(v.lo, v.hi); // 16 symbols, 32 bits
      out[k >> 1] |= packed << (32 * (1 - (k & 1)));
    }
    // first symbol in the top bits of the first byte
    out[0] = __builtin_bswap64(out[0]);
    out[1] = __builtin_bswap64(out[1]);
    reinterpret_cast<uint64_t *>(tis2)[2 * b] = out[0];
    reinterpret_cast<uint64_t *>(tis2)[2 * b + 1] = out[1];
    special = firstbad != ~0ull || 64 * b + 64 > n;
    if (firstbad != ~0ull)
    {
      atomicMin(firstspecial, (unsigned long long) firstbad);
    }
  }
  const uint64_t mask = __ballot(special);
  if ((threadIdx.x & 63) == 0 && b < nblocks)
  {
    reinterpret_cast<uint64_t *>(spec64)[b >> 6] = mask;
  }
}

// slot[code] = (bck2 pair, the first W-1 entries of the bucket), W = 2 or 4
// words; entries the bucket does not have are 0 (they stand for the entry
// behind the bucket, whose lcp byte is below D anyway)
// Wide tables: word 0 in the form of vsa_slotbounds (left | count << 40);
// *toobig is set when a bucket's count does not fit.
template <int W, typename IDX>
__global__ void __launch_bounds__(VSA_BLOCK)
k_make_slots(const IDX *__restrict__ bck2, const uint64_t *__restrict__ esa8,
             uint64_t ncodes, uint64_t *__restrict__ slot,
             unsigned int *__restrict__ toobig)
{
  for (uint64_t c = vsa_bid() * VSA_BLOCK + threadIdx.x;
       c < ncodes; c += vsa_nblocks() * VSA_BLOCK)
  {
    const IDX left = bck2[2 * c], mid = bck2[2 * c + 1];
    if constexpr (sizeof(IDX) == 4)
    {
      slot[W * c] = (uint64_t) left | ((uint64_t) mid << 32);
    } else
    {
      const uint64_t cnt = mid > left ? mid - left : 0;
      if (cnt >> (64 - VSA_WIDE_LEFTBITS) != 0)
      {
        *toobig = 1;
      }
      slot[W * c] = left | (cnt << VSA_WIDE_LEFTBITS);
    }
#pragma unroll
    for (int k = 0; k + 1 < W; k++)
    {
      slot[W * c + 1 + k] = (mid > left + k) ? esa8[left + k] : 0;
    }
  }
}

int vsa_index_make_esa8(vsa_index *ix)
{
  const char *off = getenv("VSA_NO_ESA8");
  if (ix->esa8 != nullptr)
  {
    (void) hipFree(ix->esa8);
    ix->esa8 = nullptr;
  }
  if (ix->bck2 != nullptr)
  {
    (void) hipFree(ix->bck2);
    ix->bck2 = nullptr;
  }
  if (ix->slot16 != nullptr)
  {
    (void) hipFree(ix->slot16);
    ix->slot16 = nullptr;
  }
  if (ix->tis2 != nullptr)
  {
    (void) hipFree(ix->tis2);
    (void) hipFree(ix->spec64);
    ix->tis2 = ix->spec64 = nullptr;
  }
  const bool wide = ix->isize != 4;
  if (ix->numofchars != 4 || ix->bck == nullptr ||
      (wide && ((ix->n + 1) >> VSA_WIDE_LEFTBITS) != 0) ||
      (off != nullptr && strcmp(off, "1") == 0))
  {
    return 0;
  }
  // deep prefix: about one suffix per bucket, never shorter than the
  // reference's prefixlength; the table takes 8 * 4^D bytes (at most 32n)
  // D = ceil(log4(n)), at most 16: about one suffix per bucket
  uint32_t D = 1;
  while (D < 16 && (1ull << (2 * D)) < ix->n)
  {
    D++;
  }
  if (D < ix->pl)
  {
    D = ix->pl;
  }
  const char *fd = getenv("VSA_DEEP_PREFIX");
  if (fd != nullptr && atoi(fd) >= (int) ix->pl && atoi(fd) <= 16)
  {
    D = (uint32_t) atoi(fd);
  } else
  {
    // one symbol less where the device has not room for the slot table, the
    // bucket bounds it is made from, the keyed array and a tenth of the
    // device for the searches themselves: half the index for a fifth more
    // time per batch (profiles/r03/footprint_deep_prefix.txt)
    size_t freeb = 0, totalb = 0;
    const uint32_t wanted = D;
    (void) hipStreamSynchronize(ix->stream);
    vsa_dev_trim(); // (what the builder's temporaries held counts as free)
    while (D > ix->pl && D > 12 &&
           hipMemGetInfo(&freeb, &totalb) == hipSuccess)
    {
      const uint64_t codes = 1ull << (2 * D),
                     need = 16 * codes + 2 * codes * ix->isize +
                            8 * (ix->n + 1) + ix->n / 4 + totalb / 10;
      if (need <= freeb)
      {
        break;
      }
      D--;
    }
    // (the choice depends on what else lives on the device: said aloud under
    // VSA_TRACE, readable as vsa_index_info.deepprefix, and taken once per
    // replica set -- the other replicas are copies, multi_gpu.cpp)
    const char *tr = getenv("VSA_TRACE");
    if (tr != nullptr && strcmp(tr, "0") != 0)
    {
      fprintf(stderr, "vstree_amd: derived tables: deep prefix %u (wanted %u; "
              "%.1f GB free of %.1f on the device)\n", D, wanted,
              (double) freeb / 1e9, (double) totalb / 1e9);
    }
  }
  if (D > 16)
  {
    return 0;
  }
  static_assert(VSA_TIS_BACKPAD >= 16 + VSA_KEYSYMS + 8, "text pad too small");
  ix->D = D;
  const char *tune = getenv("VSA_TUNE");
  ix->tune = tune != nullptr ? (uint32_t) atoi(tune) : 0;
  const uint64_t count = ix->n + 1, ncodes = 1ull << (2 * D);
  // The slot table is the one every search starts in, at a random place: it is
  // allocated FIRST, with the temporaries of the builder handed back to the
  // driver.  Placed last, between what the builder had left, the 68.7 GB of a
  // 3 Gbp index were mapped in small pages and a random read of it cost a read
  // of the page table on top (profiles/r03/table_read_probe.txt).
  VSA_HIP(hipStreamSynchronize(ix->stream));
  vsa_dev_trim();
  if (vsa_hip_malloc((void **) &ix->slot16, 2 * ncodes * 8 + 32) != hipSuccess)
  {
    // no room for it (VSA_DEEP_PREFIX asked for more than fits): this index
    // is searched the reference's way
    (void) hipGetLastError();
    ix->slot16 = nullptr;
    ix->D = 0;
    return 0;
  }
  VSA_HIP(vsa_hip_malloc((void **) &ix->bck2, 2 * ncodes * ix->isize + 16));
  VSA_HIP(vsa_hip_malloc((void **) &ix->esa8, count * 8 + 64));
  ix->device_bytes += count * 8 + 2 * ncodes * ix->isize;
  if (wide ? vsa_build_bucket_table(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                                    (const uint64_t *) ix->suf, D, 4,
                                    (uint64_t *) ix->bck2, ix->stream)
           : vsa_build_bucket_table(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                                    (const uint32_t *) ix->suf, D, 4,
                                    ix->bck2, ix->stream))
  {
    return -100;
  }
  if (wide)
  {
    k_make_esa8<uint64_t><<<gridfor(count), VSA_BLOCK, 0, ix->stream>>>(
        ix->tis_alloc + VSA_TIS_FRONTPAD, (const uint64_t *) ix->suf, ix->lcp,
        count, D, ix->esa8);
  } else
  {
    k_make_esa8<uint32_t><<<gridfor(count), VSA_BLOCK, 0, ix->stream>>>(
        ix->tis_alloc + VSA_TIS_FRONTPAD, (const uint32_t *) ix->suf, ix->lcp,
        count, D, ix->esa8);
  }
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipStreamSynchronize(ix->stream));
  // the 2-bit text for long comparisons
  {
    // blocks 0 .. n >> 6: a comparison ends at position n at the latest (the
    // last block reads into the 0xFF padding behind the text, not beyond it)
    static_assert(VSA_TIS_BACKPAD >= 64, "text pad too small for k_pack_text");
    const uint64_t nblocks = (ix->n >> 6) + 1,
                   nwaves = (nblocks + 63) / 64;
    unsigned long long *dfirst = nullptr, hfirst = ix->n;
    VSA_HIP(vsa_hip_malloc((void **) &ix->tis2, nblocks * 16 + 64));
    VSA_HIP(vsa_hip_malloc((void **) &ix->spec64, nwaves * 8 + 64));
    VSA_HIP(vsa_hip_malloc((void **) &dfirst, 8));
    VSA_HIP(hipMemsetAsync(ix->tis2 + nblocks * 16, 0, 64, ix->stream));
    VSA_HIP(hipMemsetAsync(ix->spec64 + nwaves * 8, 0xFF, 64, ix->stream));
    VSA_HIP(hipMemcpyAsync(dfirst, &hfirst, 8, hipMemcpyHostToDevice,
                           ix->stream));
    k_pack_text<<<vsa_grid(nwaves * 64 / VSA_BLOCK + 1), VSA_BLOCK, 0,
                  ix->stream>>>(ix->tis_alloc + VSA_TIS_FRONTPAD, ix->n,
                                nblocks, ix->tis2, ix->spec64, dfirst);
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipMemcpyAsync(&hfirst, dfirst, 8, hipMemcpyDeviceToHost,
                           ix->stream));
    VSA_HIP(hipStreamSynchronize(ix->stream));
    (void) hipFree(dfirst);
    ix->firstspecial = hfirst < ix->n ? hfirst : ix->n;
    ix->device_bytes += nblocks * 16 + nwaves * 8;
  }
  // the fused table takes the place of bck2: 16 bytes per deep prefix (bounds
  // + the first entry: 69 % of the non-empty buckets of a random text are
  // answered by one access; 68.7 GB at 3 Gbp).  (32-byte slots with three
  // entries were measured in round 2, profiles/r02/slot32_ab.txt, and read
  // again in round 4, profiles/r04/README.md: fewer HBM lines, but a second
  // load instruction per lane.)
  unsigned int *dtoobig = nullptr, htoobig = 0;
  VSA_HIP(vsa_hip_malloc((void **) &dtoobig, 4));
  VSA_HIP(hipMemsetAsync(dtoobig, 0, 4, ix->stream));
  {
    const unsigned int grid = (unsigned int) std::min<uint64_t>(
        (ncodes + VSA_BLOCK - 1) / VSA_BLOCK, 1u << 20);
    if (wide)
    {
      k_make_slots<2, uint64_t><<<grid, VSA_BLOCK, 0, ix->stream>>>(
          (const uint64_t *) ix->bck2, ix->esa8, ncodes, ix->slot16, dtoobig);
    } else
    {
      k_make_slots<2, uint32_t><<<grid, VSA_BLOCK, 0, ix->stream>>>(
          ix->bck2, ix->esa8, ncodes, ix->slot16, dtoobig);
    }
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipMemcpyAsync(&htoobig, dtoobig, 4, hipMemcpyDeviceToHost,
                           ix->stream));
    VSA_HIP(hipStreamSynchronize(ix->stream));
    (void) hipFree(ix->bck2);
    ix->bck2 = nullptr;
    ix->slotwords = 2;
    ix->device_bytes += 2 * ncodes * 8 - 2 * ncodes * ix->isize;
  }
  (void) hipFree(dtoobig);
  if (wide && htoobig != 0)
  {
    // a deep bucket with 2^24 suffixes or more: this index is searched the
    // reference's way
    const uint64_t nblocks = (ix->n >> 6) + 1, nwaves = (nblocks + 63) / 64;
    ix->device_bytes -= count * 8;
    ix->device_bytes -= (uint64_t) ix->slotwords * ncodes * 8;
    if (ix->tis2 != nullptr)
    {
      ix->device_bytes -= nblocks * 16 + nwaves * 8;
    }
    (void) hipFree(ix->esa8);
    (void) hipFree(ix->bck2);
    (void) hipFree(ix->slot16);
    (void) hipFree(ix->tis2);
    (void) hipFree(ix->spec64);
    ix->esa8 = ix->slot16 = nullptr;
    ix->bck2 = nullptr;
    ix->tis2 = ix->spec64 = nullptr;
    ix->D = 0;
  }
  return 0;
}
