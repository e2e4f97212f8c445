// vmatch -mum -l L IDX on an index that holds its queries: the streaming scan
// over lcptab + bwttab (selfmum_scan.inc, K3) and the pass over its survivors
// (Vmengine/fmumself.c:10-66) -- host pipeline and C ABI.
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

namespace
{

#include "selfmum_scan.inc"

// ---- K3 pipeline ----

// workgroups of the streaming pass: 4 per CU = all the wavefronts that fit
// (86 registers, four per SIMD), every one walking its tiles grid-stride
#define VSA_PEAK_BLOCKS 1024

// [first, last) = the values of the reference's loop variable i
// (fmumself.c:33: i = 2 .. n-1) this call covers
template <typename IDX>
int run_selfmum(const vsa_index *index, uint64_t searchlength,
                uint64_t first, uint64_t last, vsa_result *res)
{
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  const uint64_t n = index->n;
  const uint32_t nshards = VSA_CURSOR_SHARDS;
  const uint64_t pieces = 4, tilesize = 64 * pieces * 16;
  // centres j = i - 1
  const uint64_t jlo = std::max<uint64_t>(first, 2) - 1,
                 jhi = std::max<uint64_t>(std::min<uint64_t>(last, n), 2) - 1;
  const uint64_t tile0 = jlo / tilesize,
                 ntiles = jhi > jlo ? (jhi + tilesize - 1) / tilesize : tile0,
                 wavesperblock = VSA_BLOCK / 64;
  const uint64_t nblocks = std::max<uint64_t>(
      1, std::min<uint64_t>((ntiles - tile0 + wavesperblock - 1) /
                                wavesperblock,
                            (uint64_t) VSA_PEAK_BLOCKS));
  const uint32_t slmin = (uint32_t) (searchlength < 255 ? searchlength : 255);
  DevBuf cursor, doff, rawpos, peaks, sorted, temp, cand, keep, dcount, mums;
  uint64_t shardcap =
               std::max<uint64_t>((jhi - std::min(jlo, jhi)) / 64 / nshards +
                                      1024,
                                  4096),
           needed = 0, maxshard = 0;
  double searchms = 0;

  res->stats.searches = jhi > jlo ? jhi - jlo : 0;
  DevBuf summary;
  if (cursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
      doff.alloc(nshards * 8) || dcount.alloc(8) || summary.alloc(4 * 8))
  {
    return -100;
  }
  tall.start();
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (rawpos.alloc(nshards * shardcap * sizeof(IDX)))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(cursor.p, 0,
                           (size_t) nshards * VSA_CURSOR_STRIDE * 8, stream));
    tsearch.start();
    k_selfmum_peaks<true, IDX><<<(unsigned int) nblocks, VSA_BLOCK, 0,
                                 stream>>>(
        ix.lcp, ix.bwt, n, slmin, rawpos.as<IDX>(), shardcap, nshards - 1,
        cursor.as<unsigned long long>(), tile0, ntiles, jlo, jhi);
    tsearch.stop();
    VSA_HIP(hipGetLastError());
    VSA_HIP(shard_summary(cursor.as<unsigned long long>(), nshards,
                          doff.as<uint64_t>(), summary.as<uint64_t>(),
                          stream));
    {
      const Fetch f[2] = {{summary.as<uint64_t>(), 8},
                          {summary.as<uint64_t>() + 1, 8}};
      uint64_t got[2];
      if (fetchwords(stream, f, 2, got))
      {
        return -100;
      }
      needed = got[0];
      maxshard = got[1];
    }
    searchms = tsearch.ms(); // the streaming pass (of the last attempt)
    if (maxshard <= shardcap)
    {
      break;
    }
    shardcap = maxshard;
  }
  if (maxshard > shardcap)
  {
    VSA_ERROR("peak buffer overflow");
    return -5;
  }
  uint64_t nm = 0;
  if (needed > 0)
  {
    if (peaks.alloc(needed * sizeof(IDX)) ||
        sorted.alloc(needed * sizeof(IDX)) ||
        cand.alloc(needed * sizeof(vsa_match)) || keep.alloc(needed) ||
        mums.alloc(needed * sizeof(vsa_match)))
    {
      return -100;
    }
    k_gather_shards<IDX><<<nshards, VSA_BLOCK, 0, stream>>>(
        rawpos.as<IDX>(), shardcap, cursor.as<unsigned long long>(),
        doff.as<uint64_t>(), peaks.as<IDX>());
    VSA_HIP(hipGetLastError());
    // the reference reports in suffix-array order
    size_t tb = 0;
    VSA_HIP(rocprim::radix_sort_keys(nullptr, tb, peaks.as<IDX>(),
                                     sorted.as<IDX>(), (size_t) needed,
                                     0u, bitsfor(n), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_keys(temp.p, tb, peaks.as<IDX>(),
                                     sorted.as<IDX>(), (size_t) needed,
                                     0u, bitsfor(n), stream));
    k_selfmum_emit<IDX><<<gridfor(needed), VSA_BLOCK, 0, stream>>>(
        ix, sorted.as<IDX>(), needed, searchlength,
        index->querysepposition, cand.as<vsa_match>(), keep.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    if (compact_matches(cand.as<vsa_match>(), keep.as<uint8_t>(), needed,
                        mums.as<vsa_match>(), dcount.as<uint64_t>(), stream))
    {
      return -100;
    }
    {
      const Fetch f = {dcount.p, 8};
      if (fetchwords(stream, &f, 1, &nm))
      {
        return -100;
      }
    }
    VSA_HIP(hipStreamSynchronize(stream));
    res->count = nm;
    res->matches = (vsa_match *) mums.release();
  }
  tall.stop();
  VSA_HIP(hipStreamSynchronize(stream));
  res->stats.count = res->count;
  res->stats.candidates = needed;
  res->stats.search_kernel_ms = searchms;
  res->stats.total_device_ms = tall.ms();
  return sumlengths(res->matches, res->count, stream, &res->stats.sumlength);
}

} // namespace

extern "C" int vsa_findmaximaluniquematches_range(const vsa_index *index,
                                                  uint64_t searchlength,
                                                  uint64_t first,
                                                  uint64_t last,
                                                  vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmaximaluniquematches: NULL argument");
    return -1;
  }
  if (first > last)
  {
    VSA_ERROR("vsa_findmaximaluniquematches_range: first > last");
    return -1;
  }
  *result = nullptr;
  // Vmengine/fmumself.c:21-31
  if (!index->hasindexedqueries)
  {
    VSA_ERROR("maximal unique matches search requires at least one query "
              "file");
    return -1;
  }
  if (index->n < 2)
  {
    VSA_ERROR("search for maximal unique matches requires at least a table "
              "of length 2");
    return -2;
  }
  if (index->bwt == nullptr)
  {
    VSA_ERROR("table bwt is not loaded");
    return -3;
  }
  if (index->numofchars > 128)
  {
    // the streaming pass takes "is a special symbol" from bit 7 of the bwt
    // byte (vsa_peakbits4): true for 253, 254, 255 and for no symbol code
    // below 128 -- an alphabet with more symbols stays with the reference
    VSA_ERROR("maximal unique matches of an index over %u symbols: not "
              "covered (at most 128)", index->numofchars);
    return VSA_NOT_COVERED;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_selfmum<uint32_t>(index, searchlength, first, last,
                                             res)
                     : run_selfmum<uint64_t>(index, searchlength, first, last,
                                             res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findmaximaluniquematches(const vsa_index *index,
                                            uint64_t searchlength,
                                            vsa_result **result)
{
  return vsa_findmaximaluniquematches_range(index, searchlength, 2, ~0ull,
                                            result);
}
