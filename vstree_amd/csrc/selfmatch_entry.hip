// Repeats of the index itself: vmatch -l L IDX (maximal repeats,
// Vmengine/fself.c:203), -supermax (fsuper.c:142), -tandem (ftandem.c:261);
// kernels and pipelines in selfmatch_search.inc, the C ABI here.
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

namespace
{

#include "selfmatch_search.inc"

} // namespace

extern "C" int vsa_findmaximalrepeats(const vsa_index *index,
                                      uint64_t searchlength,
                                      vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmaximalrepeats: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->n < 2)
  {
    // Vmengine/fself.c:246-250
    VSA_ERROR("repeat search requires a sequence of length >= 2");
    return -2;
  }
  if (index->bwt == nullptr)
  {
    VSA_ERROR("table bwt is not loaded");
    return -3;
  }
  if (index->numofchars > VSA_REP_MAXC)
  {
    VSA_ERROR("maximal repeats on alphabets of %lu symbols are not covered "
              "by the GPU engine", (unsigned long) index->numofchars);
    return VSA_NOT_COVERED;
  }
  if (searchlength == 0)
  {
    VSA_ERROR("maximal repeats need a length of at least 1");
    return -2;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_repeats<uint32_t, uint32_t>(index, searchlength, res)
                     : run_repeats<uint64_t, uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findsupermaximalrepeats(const vsa_index *index,
                                           uint64_t searchlength,
                                           vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findsupermaximalrepeats: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->hasindexedqueries)
  {
    // Vmengine/fself.c:193-198
    VSA_ERROR("supermaximal repeat search does not allow query files in "
              "index");
    return -2;
  }
  if (index->n < 2)
  {
    // Vmengine/fself.c:246-250
    VSA_ERROR("repeat search requires a sequence of length >= 2");
    return -2;
  }
  if (index->bwt == nullptr)
  {
    VSA_ERROR("table bwt is not loaded");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_supermax<uint32_t, uint32_t>(index, searchlength, res)
                     : run_supermax<uint64_t, uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findtandems(const vsa_index *index, uint64_t searchlength,
                               vsa_result **result)
{
  if (index == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findtandems: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->hasindexedqueries)
  {
    // Vmengine/ftandem.c:271-275
    VSA_ERROR("tandem repeat search does not allow query files in index");
    return -2;
  }
  if (searchlength == 0)
  {
    VSA_ERROR("tandem repeat search needs a length >= 1");
    return -2;
  }
  if (index->tis_alloc == nullptr || index->suf == nullptr ||
      index->lcp == nullptr)
  {
    VSA_ERROR("tables tis, suf and lcp are required");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_tandems<uint32_t, uint32_t>(index, searchlength, res)
                     : run_tandems<uint64_t, uint64_t>(index, searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}
