// GPU construction of the index tables (placeholder until the builder lands)
#include "vsa_internal.hpp"

extern "C" int vsa_index_build(const uint8_t *, uint64_t, uint32_t, uint32_t,
                               int, vsa_index **)
{
  VSA_ERROR("vsa_index_build: not available in this build");
  return -1;
}

extern "C" int vsa_index_build_device(const void *, uint64_t, uint32_t,
                                      uint32_t, int, vsa_index **)
{
  VSA_ERROR("vsa_index_build_device: not available in this build");
  return -1;
}
