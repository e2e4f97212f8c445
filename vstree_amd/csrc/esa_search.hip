// Query matching on MI355X (vmatch -complete | -l L | -mum [cand] -q Q IDX):
// kernels (in the .inc files below) and their host-side pipelines.  DESIGN.md has the byte
// budgets and the measurements.
//
//   search_complete.inc  K1  k_complete_search   one work-item per query:
//                            locate -> lcptab widening -> suffix-array
//                            interval [left, left+count)
//                            k_complete_expand   one workgroup per 256 queries
//   search_query.inc     K2  k_query_search      one work-item per (query,
//                            offset): locate -> MEM enumeration or
//                            MUM-candidate test; wavefront-aggregated append
//                            into sharded regions, compaction, stable radix
//                            sort by work-item number = reference order
//   mum_workplan.inc     K2a k_mum_first / k_mum_plan: which offsets of a read
//                            can be MUM candidates at all; K2 in its planned
//                            form, k_query_search_planned
//   mem_workplan.inc     K2m k_repeat_bits / k_mem_plan: which offsets of a
//                            read a MEM search has to look at
//   mum_filter.inc       K4  candidates as (sort key, value) pairs sorted by
//                            dbstart, prefix-max scan of the right ends,
//                            flags from the keys (runs of equal dbstarts
//                            looked at as runs), survivors written as records
//                            in order (kurtz/cleanMUMcand.c:55-118)
// The other families have translation units of their own since round 4:
//   selfmum_search.hip       K3, the scan over an index that holds its queries
//   approx_entry.hip         -complete -e/-h (approx_search.inc, approx_tree.inc)
//   selfmatch_entry.hip      maximal / supermaximal / tandem repeats
//   candidate_partition.hip  grouping of MUM candidates for the N > 1 form
//   index_derive.hip         the derived tables (esa8, slot16, tis2)
//   search_common.hip        what they share (search_host.hpp)
//
// rocPRIM supplies radix sort / scan / reduce only.
//
// Switches (environment, read when an index is created):
//   VSA_TUNE=2        no MUM work reduction: every offset of every read is
//                     searched by the list form of the search kernel (the
//                     cross-check of first pass + work plan)
//   VSA_NO_ESA8=1     no deep tables: the reference walk, probe for probe
//   VSA_DEEP_PREFIX=D their depth (default ceil(log4 n), at most 16)
//   VSA_FORCE_WIDE=1  64-bit device tables whatever the size of the text
// What was measured and dropped (32-byte slots, the two-phase and the deferring
// search kernel, the anchor pass, the list form of planned batches, the filter
// on rocPRIM scans, ...) is described in DESIGN.md section 4 with its numbers
// under profiles/; the code left with round 4.
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

#include "search_complete.inc"
#include "search_query.inc"
#include "mum_workplan.inc"
#include "mem_workplan.inc"
#include "mum_filter.inc"

// ---------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------

namespace
{


// ---- K1 pipeline ----

template <typename IDX>
int run_complete(const vsa_index *index, const vsa_queries *queries,
                 uint64_t qlimit, vsa_result *res)
{
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  // packed batches: read from their rows by the deep kernel (whole in
  // registers up to four words, through windows up to eight); an index
  // without deep tables, or longer reads, takes their bytes
  const bool rows = queries->rows != nullptr && ix.esa8 != nullptr &&
                    queries->roww <= 8 && queries->maxlength >= ix.D;
  if (queries->rows != nullptr && !rows &&
      vsa_queries_bytes(queries, stream) != 0)
  {
    return -100;
  }
  const DevQueries qs = devqueries(queries);
  DevBuf left, count, offsets, temp, matches;
  uint64_t total = 0;

  res->stats.searches = qlimit;
  if (qlimit == 0)
  {
    return 0;
  }
  if (left.alloc(qlimit * 8) || count.alloc((qlimit + 1) * 8) ||
      offsets.alloc((qlimit + 1) * 8))
  {
    return -100;
  }
  tall.start();
  VSA_HIP(hipMemsetAsync(count.as<uint64_t>() + qlimit, 0, 8, stream));
  tsearch.start();
  {
    const bool staged = ix.esa8 != nullptr && qs.dense != 0 &&
                        qs.symbols != nullptr && qs.uniformlen <= 128 &&
                        (qs.uniformlen & 3u) == 0 && qs.uniformlen >= ix.D;
    if (rows && queries->roww > 4)
    {
      k_complete_search<IDX, true, true, true, true>
          <<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
              ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    } else if (rows)
    {
      k_complete_search<IDX, true, true, true>
          <<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
              ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    } else if (staged)
    {
      k_complete_search<IDX, true, true>
          <<<gridfor(qlimit), VSA_BLOCK, (size_t) VSA_BLOCK * qs.uniformlen,
             stream>>>(ix, qs, qlimit, left.as<uint64_t>(),
                       count.as<uint64_t>());
    } else if (ix.esa8 != nullptr)
    {
      k_complete_search<IDX, true><<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
          ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    } else
    {
      k_complete_search<IDX, false>
          <<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
              ix, qs, qlimit, left.as<uint64_t>(), count.as<uint64_t>());
    }
  }
  tsearch.stop();
  VSA_HIP(hipGetLastError());
  size_t tb = 0;
  VSA_HIP(rocprim::exclusive_scan(nullptr, tb, count.as<uint64_t>(),
                                  offsets.as<uint64_t>(), (uint64_t) 0,
                                  (size_t) (qlimit + 1),
                                  rocprim::plus<uint64_t>(), stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(temp.p, tb, count.as<uint64_t>(),
                                  offsets.as<uint64_t>(), (uint64_t) 0,
                                  (size_t) (qlimit + 1),
                                  rocprim::plus<uint64_t>(), stream));
  VSA_HIP(hipMemcpyAsync(&total, offsets.as<uint64_t>() + qlimit, 8,
                         hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  if (total > 0)
  {
    if (matches.alloc(total * sizeof(vsa_match)))
    {
      return -100;
    }
    k_complete_expand<IDX><<<gridfor(qlimit), VSA_BLOCK, 0, stream>>>(
        ix, qs, qlimit, left.as<uint64_t>(), offsets.as<uint64_t>(), total,
        matches.as<vsa_match>());
    VSA_HIP(hipGetLastError());
  }
  tall.stop();
  VSA_HIP(hipStreamSynchronize(stream));
  res->count = total;
  res->matches = (vsa_match *) matches.release();
  res->stats.count = total;
  res->stats.search_kernel_ms = tsearch.ms();
  res->stats.total_device_ms = tall.ms();
  // every complete match has the length of its query
  return sumlengths(res->matches, total, stream, &res->stats.sumlength);
}

// ---- K2 (+K4) pipeline ----

// MUM candidates, any order -> MUMs in dbstart order
// (max dbstart, max length) of a candidate list
struct MaxPair
{
  uint64_t db, len;
};

struct MaxPairOf
{
  __device__ MaxPair operator()(const vsa_match &m) const
  {
    MaxPair p;
    p.db = m.dbstart;
    p.len = m.length;
    return p;
  }
};

struct MaxPairOp
{
  __device__ MaxPair operator()(const MaxPair &a, const MaxPair &b) const
  {
    MaxPair p;
    p.db = a.db > b.db ? a.db : b.db;
    p.len = a.len > b.len ? a.len : b.len;
    return p;
  }
};

// one key for "dbstart ascending, then length descending"
__global__ void __launch_bounds__(VSA_BLOCK)
k_mum_compositekeys(const vsa_match *__restrict__ cand, uint64_t n,
                    unsigned int lenbits, uint64_t *__restrict__ key,
                    uint32_t *__restrict__ idx)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t lenmask = (1ull << lenbits) - 1;
    key[i] = (cand[i].dbstart << lenbits) | (lenmask - cand[i].length);
    idx[i] = (uint32_t) i;
  }
}

// carry = the reference's running `dbright` when it reaches the first of
// these candidates: 0 for a whole job, the largest right end of all
// candidates with a smaller dbstart when the list is one dbstart range of a
// job that is filtered in pieces (multi-GPU).  *maxright (optional) receives
// the largest right end in this list.
int mumuniqueinquery(DevBuf &cand, uint64_t ncand, hipStream_t stream,
                     DevBuf &mums, uint64_t *nmums, uint64_t carry = 0,
                     uint64_t *maxright = nullptr, uint64_t dbbound = 0,
                     uint64_t lenbound = 0, uint64_t *sumlength = nullptr)
{
  // *sumlength (if asked for) = sum of the lengths of the MUMs, or ~0 if this
  // call did not compute it
  // dbbound / lenbound: upper bounds of dbstart and length if the caller
  // knows them (index length, longest query), else 0: they are looked up
  *nmums = 0;
  if (maxright != nullptr)
  {
    *maxright = 0;
  }
  if (sumlength != nullptr)
  {
    *sumlength = (ncand == 0) ? 0 : ~0ull;
  }
  if (ncand == 0)
  {
    return 0;
  }
  DevBuf ends, dbright, keep, temp, dcount, sorted, k1, k2, i1, i2;
  bool onkeys = false; // the filter runs on the sorted composite keys
  if (ends.alloc(ncand * 8) || dbright.alloc(ncand * 8) ||
      keep.alloc(ncand) || dcount.alloc(sizeof(MaxPair) + 16))
  {
    return -100;
  }
  // how many bits do dbstart and length need?
  MaxPair mx;
  mx.db = dbbound;
  mx.len = lenbound;
  if (dbbound == 0 || lenbound == 0)
  {
    size_t tb = 0;
    auto in = rocprim::make_transform_iterator(cand.as<vsa_match>(),
                                               MaxPairOf());
    MaxPair init;
    init.db = init.len = 0;
    VSA_HIP(rocprim::reduce(nullptr, tb, in, dcount.as<MaxPair>(), init,
                            (size_t) ncand, MaxPairOp(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(temp.p, tb, in, dcount.as<MaxPair>(), init,
                            (size_t) ncand, MaxPairOp(), stream));
    VSA_HIP(hipMemcpyAsync(&mx, dcount.p, sizeof mx, hipMemcpyDeviceToHost,
                           stream));
    VSA_HIP(hipStreamSynchronize(stream));
  }
  const unsigned int lenbits = bitsfor(mx.len), dbbits = bitsfor(mx.db);
  if (lenbits + dbbits <= 64 && ncand < 0xFFFFFFFFull)
  {
    // one radix sort of (composite key, index) over just the bits in use;
    // the keys carry all the filter looks at
    onkeys = true;
    if (k1.alloc(ncand * 8) || k2.alloc(ncand * 8) || i1.alloc(ncand * 4) ||
        i2.alloc(ncand * 4))
    {
      return -100;
    }
    k_mum_compositekeys<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), ncand, lenbits, k1.as<uint64_t>(),
        i1.as<uint32_t>());
    VSA_HIP(hipGetLastError());
    size_t tb = 0;
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, k1.as<uint64_t>(), k2.as<uint64_t>(), i1.as<uint32_t>(),
        i2.as<uint32_t>(), (size_t) ncand, 0u, lenbits + dbbits, stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, k1.as<uint64_t>(), k2.as<uint64_t>(), i1.as<uint32_t>(),
        i2.as<uint32_t>(), (size_t) ncand, 0u, lenbits + dbbits, stream));
    k_mum_keyends<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), ncand, lenbits, ends.as<uint64_t>());
    VSA_HIP(hipGetLastError());
  } else
  {
    // wide values: least significant key first (length descending), then a
    // stable sort by dbstart
    DevBuf kout;
    if (k1.alloc(ncand * 8) || k2.alloc(ncand * 8) || kout.alloc(ncand * 8) ||
        sorted.alloc(ncand * sizeof(vsa_match)))
    {
      return -100;
    }
    k_mum_keys<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), ncand, k1.as<uint64_t>(), k2.as<uint64_t>());
    VSA_HIP(hipGetLastError());
    if (sortbykey(k1.as<uint64_t>(), kout.as<uint64_t>(),
                  cand.as<vsa_match>(), sorted.as<vsa_match>(), ncand, 64,
                  stream))
    {
      return -100;
    }
    k_mum_keys<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        sorted.as<vsa_match>(), ncand, k1.as<uint64_t>(), k2.as<uint64_t>());
    VSA_HIP(hipGetLastError());
    if (sortbykey(k2.as<uint64_t>(), kout.as<uint64_t>(),
                  sorted.as<vsa_match>(), cand.as<vsa_match>(), ncand, 64,
                  stream))
    {
      return -100;
    }
    VSA_HIP(hipMemcpyAsync(sorted.p, cand.p, ncand * sizeof(vsa_match),
                           hipMemcpyDeviceToDevice, stream));
    k_mum_rightends<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        sorted.as<vsa_match>(), ncand, ends.as<uint64_t>());
    VSA_HIP(hipGetLastError());
  }
  size_t tb = 0;
  VSA_HIP(rocprim::exclusive_scan(nullptr, tb, ends.as<uint64_t>(),
                                  dbright.as<uint64_t>(), carry,
                                  (size_t) ncand, rocprim::maximum<uint64_t>(),
                                  stream));
  if (temp.alloc(tb))
  {
    return -100;
  }
  VSA_HIP(rocprim::exclusive_scan(temp.p, tb, ends.as<uint64_t>(),
                                  dbright.as<uint64_t>(), carry,
                                  (size_t) ncand, rocprim::maximum<uint64_t>(),
                                  stream));
  if (maxright != nullptr)
  {
    // sorted by dbstart, so the running maximum behind the last element
    uint64_t lastend = 0, lastmax = 0;
    VSA_HIP(hipMemcpyAsync(&lastend, ends.as<uint64_t>() + ncand - 1, 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipMemcpyAsync(&lastmax, dbright.as<uint64_t>() + ncand - 1, 8,
                           hipMemcpyDeviceToHost, stream));
    VSA_HIP(hipStreamSynchronize(stream));
    *maxright = std::max(lastend, lastmax);
  }
  if (mums.alloc(ncand * sizeof(vsa_match)))
  {
    return -100;
  }
  if (onkeys)
  {
    DevBuf slots;
    uint64_t hsum = 0;
    if (slots.alloc(ncand * 4))
    {
      return -100;
    }
    k_mum_keyflags<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), dbright.as<uint64_t>(), ncand, lenbits,
        keep.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    auto keepit =
        rocprim::make_transform_iterator(keep.as<uint8_t>(), KeepToU32());
    tb = 0;
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(temp.p, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    DevBuf blocksum;
    const size_t nblocks = blocksfor(ncand);
    if (blocksum.alloc(vsa_grid_blocks(nblocks) * 8))
    {
      return -100;
    }
    k_mum_writekept<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        cand.as<vsa_match>(), i2.as<uint32_t>(), keep.as<uint8_t>(),
        slots.as<uint32_t>(), ncand, mums.as<vsa_match>(),
        dcount.as<uint64_t>(), blocksum.as<unsigned long long>());
    VSA_HIP(hipGetLastError());
    tb = 0;
    VSA_HIP(rocprim::reduce(nullptr, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull,
                            nblocks, rocprim::plus<unsigned long long>(),
                            stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(temp.p, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull,
                            nblocks, rocprim::plus<unsigned long long>(),
                            stream));
    {
      const Fetch f[2] = {{dcount.p, 8}, {dcount.as<uint64_t>() + 1, 8}};
      uint64_t got[2];
      if (fetchwords(stream, f, 2, got))
      {
        return -100;
      }
      *nmums = got[0];
      hsum = got[1];
    }
    if (sumlength != nullptr)
    {
      *sumlength = hsum;
    }
    return 0;
  }
  k_mum_flags<<<gridfor(ncand), VSA_BLOCK, 0, stream>>>(
      sorted.as<vsa_match>(), ends.as<uint64_t>(), dbright.as<uint64_t>(),
      ncand, keep.as<uint8_t>());
  VSA_HIP(hipGetLastError());
  if (compact_matches(sorted.as<vsa_match>(), keep.as<uint8_t>(), ncand,
                      mums.as<vsa_match>(), dcount.as<uint64_t>(), stream))
  {
    return -100;
  }
  VSA_HIP(hipMemcpyAsync(nmums, dcount.p, 8, hipMemcpyDeviceToHost, stream));
  VSA_HIP(hipStreamSynchronize(stream));
  return 0;
}

// The filter of mumuniqueinquery on a packed candidate list: keys =
// dbstart << lenbits | (2^lenbits - 1 - length), values = queryseq << 16 |
// querystart (k_query_search with packbits).  One sort of the pairs, the
// filter on the keys, and the surviving pairs become the records: the
// candidates never exist as 32-byte records.
// keys_in / vals_in: anything rocPRIM can read (pointers, or iterators over
// rows of pairs as they come out of the exchange: no copy into two arrays)
template <typename VAL = uint64_t, typename KeyIn = const uint64_t *,
          typename ValIn = const VAL *>
int mumfilter_packed(KeyIn keys_in, ValIn vals_in, uint64_t ncand,
                     unsigned int lenbits, unsigned int dbbits,
                     hipStream_t stream, DevBuf &mums, uint64_t *nmums,
                     uint64_t *sumlength, uint64_t carry = 0,
                     unsigned int valbits = 0, uint64_t seqoffset = 0)
{
  // VAL, valbits, seqoffset: see k_mum_writepacked
  // carry: as for mumuniqueinquery
  *nmums = 0;
  *sumlength = 0;
  if (ncand == 0)
  {
    return 0;
  }
  DevBuf k2, v2, dbright, keep, slots, temp, dcount, blocksum;
  const size_t nblocks = blocksfor(ncand);
  // the passes behind the sort run by tiles (mum_filter.inc); the rocPRIM
  // scans of round 2 are what the rare second pass falls back on
  const uint64_t ntiles = (ncand + VSA_FT_TILE - 1) / VSA_FT_TILE;
  DevBuf tmax, tcarry, tcount, toff, tsum, tsumscan;
  if (tmax.alloc((ntiles + 1) * 8) || tcarry.alloc((ntiles + 1) * 8) ||
      tcount.alloc((ntiles + 1) * 8) || toff.alloc((ntiles + 1) * 8) ||
      tsum.alloc((ntiles + 1) * 8) || tsumscan.alloc((ntiles + 1) * 8))
  {
    return -100;
  }
  if (k2.alloc(ncand * 8) || v2.alloc(ncand * sizeof(VAL)) ||
      keep.alloc(ntiles * VSA_FT_TILE) || dcount.alloc(24) ||
      blocksum.alloc(vsa_grid_blocks(nblocks) * 8) ||
      mums.alloc(ncand * sizeof(vsa_match)))
  {
    return -100;
  }
  // Sorted by dbstart alone where the runs of equal dbstarts are short (see
  // k_mumf_flags; decided afterwards from a flag that comes back with the
  // counts), by (dbstart, length down) otherwise.
  size_t tb = 0;
  uint64_t got[3] = {0, 0, 0};
  for (int pass = 0; pass < 2; pass++)
  {
    const bool byruns = pass == 0;
    const unsigned int firstbit = byruns ? lenbits : 0u;
    VSA_HIP(hipMemsetAsync(dcount.p, 0, 24, stream));
    tb = 0;
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, keys_in, k2.as<uint64_t>(), vals_in,
        v2.as<VAL>(), (size_t) ncand, firstbit, lenbits + dbbits, stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, keys_in, k2.as<uint64_t>(), vals_in,
        v2.as<VAL>(), (size_t) ncand, firstbit, lenbits + dbbits, stream));
    if (byruns)
    {
      const dim3 tg = vsa_grid(ntiles);
      k_mumf_tilemax<<<tg, VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), ncand, lenbits, tmax.as<uint64_t>());
      k_mumf_scan<1><<<1, VSA_BLOCK, 0, stream>>>(
          tmax.as<uint64_t>(), ntiles, carry, tcarry.as<uint64_t>());
      k_mumf_flags<<<tg, VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), ncand, lenbits, tcarry.as<uint64_t>(),
          keep.as<uint8_t>(), tcount.as<uint64_t>(), tsum.as<uint64_t>(),
          dcount.as<unsigned int>() + 4);
      // (offsets of the tiles and, in a second workgroup, the sum of the
      // lengths)
      k_mumf_scan<0><<<2, VSA_BLOCK, 0, stream>>>(
          tcount.as<uint64_t>(), ntiles, 0, toff.as<uint64_t>(),
          tsum.as<uint64_t>(), tsumscan.as<uint64_t>());
      k_mumf_write<VAL><<<tg, VSA_BLOCK, 0, stream>>>(
          k2.as<uint64_t>(), v2.as<VAL>(), keep.as<uint8_t>(), ncand,
          toff.as<uint64_t>(), lenbits, valbits, seqoffset,
          mums.as<vsa_match>());
      VSA_HIP(hipGetLastError());
      // number of MUMs, sum of their lengths, "a run was too long"
      const Fetch f[3] = {{toff.as<uint64_t>() + ntiles, 8},
                          {tsumscan.as<uint64_t>() + ntiles, 8},
                          {dcount.as<uint64_t>() + 2, 8}};
      if (fetchwords(stream, f, 3, got))
      {
        return -100;
      }
      if (got[2] == 0)
      {
        break;
      }
      continue;
    }
    // a run of more than 64 equal dbstarts: sorted on all bits by now
    if (dbright.alloc(ncand * 8) || slots.alloc(ncand * 4))
    {
      return -100;
    }
    // running maximum of the right ends, which are a function of the keys
    auto ends = rocprim::make_transform_iterator(k2.as<uint64_t>(),
                                                 KeyToRightEnd{lenbits});
    tb = 0;
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, ends, dbright.as<uint64_t>(),
                                    carry, (size_t) ncand,
                                    rocprim::maximum<uint64_t>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(temp.p, tb, ends, dbright.as<uint64_t>(),
                                    carry, (size_t) ncand,
                                    rocprim::maximum<uint64_t>(), stream));
    k_mum_keyflags<<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), dbright.as<uint64_t>(), ncand, lenbits,
        keep.as<uint8_t>());
    VSA_HIP(hipGetLastError());
    auto keepit =
        rocprim::make_transform_iterator(keep.as<uint8_t>(), KeepToU32());
    tb = 0;
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(temp.p, tb, keepit, slots.as<uint32_t>(),
                                    (uint32_t) 0, (size_t) ncand,
                                    rocprim::plus<uint32_t>(), stream));
    k_mum_writepacked<VAL><<<vsa_grid(nblocks), VSA_BLOCK, 0, stream>>>(
        k2.as<uint64_t>(), v2.as<VAL>(), keep.as<uint8_t>(),
        slots.as<uint32_t>(), ncand, lenbits, valbits, seqoffset,
        mums.as<vsa_match>(), dcount.as<uint64_t>(),
        blocksum.as<unsigned long long>());
    VSA_HIP(hipGetLastError());
    tb = 0;
    VSA_HIP(rocprim::reduce(nullptr, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull, nblocks,
                            rocprim::plus<unsigned long long>(), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(temp.p, tb, blocksum.as<unsigned long long>(),
                            dcount.as<unsigned long long>() + 1, 0ull, nblocks,
                            rocprim::plus<unsigned long long>(), stream));
    const Fetch f[3] = {{dcount.p, 8}, {dcount.as<uint64_t>() + 1, 8},
                        {dcount.as<uint64_t>() + 2, 8}};
    if (fetchwords(stream, f, 3, got))
    {
      return -100;
    }
    break; // (the second pass: sorted on all bits, nothing left to decide)
  }
  *nmums = got[0];
  *sumlength = got[1];
  return 0;
}

template <typename IDX>
int run_query(const vsa_index *index, const vsa_queries *queries, bool domum,
              bool domumcand, uint32_t searchlength, vsa_result *res,
              bool ordered = true, uint32_t forcebits = 0)
{
  // forcebits != 0 (with domumcand, !ordered): the candidates stay pairs with
  // this many length bits (vsa_findmumcandidates_packed)
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  Timer tall(stream), tsearch(stream);
  const DevIndex<IDX> ix = index->view<IDX>();
  DevQueries qs = devqueries(queries);
  DevBuf base, cursor, out, keys;
  uint64_t nitems = 0;
  uint32_t perquery = 0;
  const uint64_t *dbase = nullptr;

  // work-items: one per query suffix with remaining >= searchlength
  // (kurtz/matchsub.c:187-196: shorter queries are skipped silently)
  if (qs.uniformlen != 0)
  {
    perquery = (qs.uniformlen >= searchlength)
                   ? qs.uniformlen - searchlength + 1
                   : 0;
    nitems = (uint64_t) perquery * queries->nq;
  } else
  {
    // base[q] = number of work-items in front of query q, from the lengths
    // on the device (a host loop and an upload of 8 bytes per query cost
    // more than the search for a batch of millions of reads)
    const uint64_t nqr = queries->nq;
    DevBuf btemp;
    size_t tb = 0;
    const uint64_t least = searchlength;
    auto items = rocprim::make_transform_iterator(
        rocprim::counting_iterator<uint64_t>(0),
        [len = qs.length, nqr, least] __device__(uint64_t q) -> uint64_t {
          if (q >= nqr)
          {
            return 0; // the entry behind the last query: the total
          }
          const uint64_t l = len[q];
          return l >= least ? l - least + 1 : 0;
        });
    if (base.alloc((nqr + 1) * 8))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(nullptr, tb, items, base.as<uint64_t>(),
                                    (uint64_t) 0, (size_t) (nqr + 1),
                                    rocprim::plus<uint64_t>(), stream));
    if (btemp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::exclusive_scan(btemp.p, tb, items, base.as<uint64_t>(),
                                    (uint64_t) 0, (size_t) (nqr + 1),
                                    rocprim::plus<uint64_t>(), stream));
    {
      const Fetch f = {base.as<uint64_t>() + nqr, 8};
      if (fetchwords(stream, &f, 1, &nitems))
      {
        return -100;
      }
    }
    dbase = base.as<uint64_t>();
  }
  res->stats.searches = nitems;
  if (nitems == 0)
  {
    return 0;
  }
  const uint32_t nshards = VSA_CURSOR_SHARDS;
  const bool deepok = ix.esa8 != nullptr && searchlength >= ix.D;
  tall.start();
  // MUM modes: first pass + work plan (mum_workplan.inc)
  DevBuf wcount, wtemp, wplan, wlist, wfirste, wfmlen, wfmdb, wboffset;
  uint64_t nfirstpass = 0; // candidates of the first pass (k_append_first)
  // reads the first pass left to the plan (low half) | first-pass candidates
  // (high half), on the device
  const uint64_t *nlistword = nullptr;
  uint64_t plansearches = 0, nfirst = 0, mumsum = ~0ull;
  // -mum with the filter: candidates as (sort key, value) pairs, see
  // mumfilter_packed
  const bool keeppairs = domum && domumcand && !ordered && forcebits != 0;
  const unsigned int lenbits =
                         keeppairs ? forcebits : bitsfor(queries->maxlength),
                     dbbits = bitsfor(index->n);
  const bool packed = domum && (!domumcand || keeppairs) &&
                      lenbits + dbbits <= 64 &&
                      lenbits >= bitsfor(queries->maxlength) &&
                      queries->maxlength < 0xFFFFu &&
                      ((queries->nq + qs.seqoffset) >> 48) == 0;
  if (keeppairs && !packed)
  {
    VSA_ERROR("packed candidates: %u length bits do not fit this batch "
              "(longest query %lu, index %lu)", forcebits,
              (unsigned long) queries->maxlength, (unsigned long) index->n);
    return -2;
  }
  const uint32_t packbits = packed ? lenbits : 0;
  // 4-byte values where query number and offset fit (not for pairs that
  // travel to other ranks: those carry the global query number)
  const uint32_t valbits =
      (packed && !keeppairs && ((queries->nq << lenbits) >> 32) == 0)
          ? lenbits
          : 0;
  const size_t recsize = valbits != 0 ? 4 : (packed ? 8 : sizeof(vsa_match));
  bool fromplan = false, planemit = false;
  DevBuf pcursor, pdoff, psummary, prawout, prawkeys; // see PlanEmit
  uint64_t pcap = 0, nplan = 0;
  uint64_t nwork = nitems;
  Timer tfirst(stream); // the first pass kernel (k_mum_first) alone
  // ragged batches take the same route with per-query geometry
  const uint64_t maxoffsets =
      (queries->maxlength >= searchlength)
          ? queries->maxlength - searchlength + 1
          : 0;
  if (domum && queries->maxlength >= 255 && index->lcpquirk < 0)
  {
    uint8_t b = 0;
    if (index->n >= 2)
    {
      VSA_HIP(hipMemcpyAsync(&b, index->lcp + index->n - 1, 1,
                             hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipStreamSynchronize(stream));
    }
    index->lcpquirk = (b == 255) ? 1 : 0;
  }
  // The work reduction (see k_mum_first, k_mum_plan) rests on "a match that
  // is not unique is no candidate"; the reference's test for lcp >= 255
  // (fquery.c:352) breaks that rule in one situation, which one byte of
  // lcptab rules out (see vsa_index::lcpquirk).  A plan holds 16-bit offsets.
  // VSA_TUNE=2: no work reduction (every offset is searched by the list form
  // of the search kernel -- the cross-check of everything below).
  const bool reduce = domum && maxoffsets > 1 && maxoffsets < 0xFFFFu &&
                      queries->nq < 0xFFFFFFFFull && (index->tune & 2u) == 0 &&
                      !(queries->maxlength >= 255 && index->lcpquirk != 0);
  // packed batches (reads at two bits per symbol): first pass, plan and
  // search kernel read the rows; everything else takes the bytes, which are
  // made on the device once per batch
  // (rows of up to four words -- reads of up to 124 symbols -- come into
  // registers whole in the first pass; longer ones, 150 bp, are looked at
  // through windows of their rows there as well)
  const bool rows = queries->rows != nullptr && reduce && deepok &&
                    queries->roww <= 8;
  if (queries->rows != nullptr && !rows)
  {
    if (vsa_queries_bytes(queries, stream) != 0)
    {
      return -100;
    }
    qs = devqueries(queries);
  }
  // MEM (-l L): first pass, then the plan of mem_workplan.inc -- aligned
  // stretches answered from one bit per text position, the rest searched
  // (a packed batch has its bytes by now: MEM reads bytes)
  const bool memplan = !domum && deepok && searchlength <= 255 &&
                       maxoffsets > 1 &&
                       maxoffsets < 0xFFFFu && queries->nq < 0xFFFFFFFFull &&
                       (index->tune & 2u) == 0;
  if (memplan)
  {
    const uint64_t nq = queries->nq;
    // one bit per text position: does its suffix have a neighbour in the
    // suffix array with lcp >= L?  Made once per (index, L), kept.
    if (index->repbits == nullptr || index->repleast != searchlength)
    {
      if (index->repbits == nullptr)
      {
        VSA_HIP(vsa_hip_malloc((void **) &index->repbits,
                               ((index->n + 1) / 32 + 4) * 4));
      }
      VSA_HIP(hipMemsetAsync(index->repbits, 0, ((index->n + 1) / 32 + 4) * 4,
                             stream));
      k_repeat_bits<IDX><<<vsa_grid(blocksfor((index->n + 16) / 16)),
                           VSA_BLOCK, 0, stream>>>(
          ix.lcp, ix.suf, index->n, searchlength, index->repbits);
      VSA_HIP(hipGetLastError());
      index->repleast = searchlength;
    }
    if (wcount.alloc((nq + 1) * 4) || wfirste.alloc(nq * 4) ||
        wfmlen.alloc(nq * 4) || wfmdb.alloc(nq * 8) ||
        wplan.alloc(nq * sizeof(PlanRanges)))
    {
      return -100;
    }
    tfirst.start();
    {
      const bool staged = qs.dense != 0 && qs.uniformlen <= 128 &&
                          (qs.uniformlen & 3u) == 0;
      if (queries->rows != nullptr && queries->roww <= 4)
      {
        // (a packed batch: offset 0 of every read from its row, as -mum does)
        k_mum_first<IDX, true, true, true>
            <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
                ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
                wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
                wfmdb.as<uint64_t>());
      } else if (queries->rows != nullptr && queries->roww <= 8)
      {
        k_mum_first<IDX, true, true, true, true>
            <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
                ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
                wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
                wfmdb.as<uint64_t>());
      } else if (staged)
      {
        k_mum_first<IDX, true, true>
            <<<gridfor(nq), VSA_BLOCK, (size_t) VSA_BLOCK * qs.uniformlen,
               stream>>>(ix, qs, perquery, searchlength,
                         wcount.as<uint32_t>(), wfirste.as<uint32_t>(),
                         wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>());
      } else
      {
        k_mum_first<IDX, true><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
            ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
            wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
            wfmdb.as<uint64_t>());
      }
    }
    tfirst.stop();
    VSA_HIP(hipGetLastError());
    {
      // room for every answer of the workgroups that share a region (a read
      // answers at most one offset per round)
      const uint64_t nb = blocksfor(nq),
                     pershard = (nb + nshards - 1) / nshards;
      pcap = pershard * VSA_BLOCK * VSA_PLAN_ROUNDS;
      if (pcursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
          pdoff.alloc(nshards * 8) || psummary.alloc(4 * 8) ||
          prawout.alloc(nshards * pcap * recsize) ||
          prawkeys.alloc(nshards * pcap * 8))
      {
        return -100;
      }
      VSA_HIP(hipMemsetAsync(pcursor.p, 0,
                             (size_t) nshards * VSA_CURSOR_STRIDE * 8,
                             stream));
      PlanEmit em;
      em.base = dbase;
      em.perquery = perquery;
      em.out = prawout.as<vsa_match>();
      em.outkey = prawkeys.as<uint64_t>();
      em.shardcap = pcap;
      em.shardmask = nshards - 1;
      em.cursors = pcursor.as<unsigned long long>();
      em.packbits = 0;
      em.valbits = 0;
      k_mem_plan<IDX><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          ix, qs, perquery, searchlength, wfirste.as<uint32_t>(),
          wfmdb.as<uint64_t>(), index->repbits, wcount.as<uint32_t>(),
          wplan.as<PlanRanges>(), em);
      VSA_HIP(hipGetLastError());
      VSA_HIP(shard_summary(pcursor.as<unsigned long long>(), nshards,
                            pdoff.as<uint64_t>(), psummary.as<uint64_t>(),
                            stream));
    }
    planemit = true;
    fromplan = true;
    plansearches = nq; // (an upper bound of the plan's own locates per round)
  }
  if (reduce)
  {
    const uint64_t nq = queries->nq;
    if (wcount.alloc((nq + 1) * 4) || wfirste.alloc(nq * 4) ||
        wfmlen.alloc(nq * 4) || wfmdb.alloc(nq * 8) ||
        wplan.alloc(nq * sizeof(PlanRanges)) || wlist.alloc(nq * 4))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(wcount.as<uint32_t>() + nq, 0, 4, stream));
    // per workgroup of the first pass: reads it leaves to the plan | reads
    // whose offset 0 is a candidate (counted by the first pass itself)
    // (queries < 2^32: the condition of this branch)
    const uint64_t nb = blocksfor(nq), nbr = vsa_grid_blocks(nb);
    DevBuf bcount;
    if (bcount.alloc((nbr + 1) * 8) || wboffset.alloc((nbr + 1) * 8))
    {
      return -100;
    }
    // (both halves of a count stay below 2^32: nq does)
    VSA_HIP(hipMemsetAsync(bcount.as<uint64_t>() + nb, 0, 8, stream));
    tfirst.start();
    if (rows && queries->roww > 4)
    {
      k_mum_first<IDX, true, true, true, true>
          <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
              ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
              wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
              wfmdb.as<uint64_t>(), bcount.as<uint64_t>());
    } else if (rows)
    {
      k_mum_first<IDX, true, true, true>
          <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
              ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
              wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
              wfmdb.as<uint64_t>(), bcount.as<uint64_t>());
    } else if (deepok)
    {
      // reads of one length m (a multiple of 4, <= 128), back to back:
      // staged through LDS and packed
      const bool staged = qs.dense != 0 && qs.uniformlen <= 128 &&
                          (qs.uniformlen & 3u) == 0;
      if (staged)
      {
        k_mum_first<IDX, true, true>
            <<<gridfor(nq), VSA_BLOCK, (size_t) VSA_BLOCK * qs.uniformlen,
               stream>>>(ix, qs, perquery, searchlength,
                         wcount.as<uint32_t>(), wfirste.as<uint32_t>(),
                         wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>(),
                         bcount.as<uint64_t>());
      } else
      {
        k_mum_first<IDX, true><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
            ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
            wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
            wfmdb.as<uint64_t>(), bcount.as<uint64_t>());
      }
    } else
    {
      k_mum_first<IDX, false><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
          ix, qs, perquery, searchlength, wcount.as<uint32_t>(),
          wfirste.as<uint32_t>(), wfmlen.as<uint32_t>(),
          wfmdb.as<uint64_t>(), bcount.as<uint64_t>());
    }
    tfirst.stop();
    VSA_HIP(hipGetLastError());
    // the reads the first pass has not finished, as a list (its counts per
    // workgroup, a scan over the workgroups, an ordered fill); with it come
    // the places of the first pass's candidates
    size_t tb = 0;
    {
      VSA_HIP(rocprim::exclusive_scan(nullptr, tb, bcount.as<uint64_t>(),
                                      wboffset.as<uint64_t>(), (uint64_t) 0,
                                      (size_t) (nb + 1),
                                      rocprim::plus<uint64_t>(), stream));
      if (wtemp.alloc(tb))
      {
        return -100;
      }
      VSA_HIP(rocprim::exclusive_scan(wtemp.p, tb, bcount.as<uint64_t>(),
                                      wboffset.as<uint64_t>(), (uint64_t) 0,
                                      (size_t) (nb + 1),
                                      rocprim::plus<uint64_t>(), stream));
      k_wanted_fill<<<vsa_grid(nb), VSA_BLOCK, 0, stream>>>(
          wcount.as<uint32_t>(), nq, 0u, wboffset.as<uint64_t>(),
          wlist.as<uint32_t>());
      VSA_HIP(hipGetLastError());
      // (the length of the list and the number of first-pass candidates,
      // wboffset[nb], come to the host with the counts behind the search
      // kernel: no wait here -- the plan kernel is launched over all reads
      // and reads the length on the device)
      nlistword = wboffset.as<uint64_t>() + nb;
    }
    {
      // on the deep tables the plan answers the offsets it locates itself
      // (PlanEmit)
      planemit = deepok;
      if (planemit)
      {
        // room for every search A of the workgroups that share a region
        // (sized for a list of all reads: its length is not known here)
        const uint64_t nb = blocksfor(nq),
                       pershard = (nb + nshards - 1) / nshards;
        pcap = pershard * VSA_BLOCK * (VSA_PLAN_ROUNDS - 1);
        if (pcursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
            pdoff.alloc(nshards * 8) || psummary.alloc(4 * 8) ||
            prawout.alloc(nshards * pcap * recsize) ||
            prawkeys.alloc(nshards * pcap * 8))
        {
          return -100;
        }
        VSA_HIP(hipMemsetAsync(pcursor.p, 0,
                               (size_t) nshards * VSA_CURSOR_STRIDE * 8,
                               stream));
        PlanEmit em;
        em.base = dbase;
        em.perquery = perquery;
        em.out = prawout.as<vsa_match>();
        em.outkey = prawkeys.as<uint64_t>();
        em.shardcap = pcap;
        em.shardmask = nshards - 1;
        em.cursors = pcursor.as<unsigned long long>();
        em.packbits = packbits;
        em.valbits = valbits;
        if (rows)
        {
          k_mum_plan<IDX, true, true, true>
              <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
                  ix, qs, wlist.as<uint32_t>(), nlistword, searchlength,
                  wfirste.as<uint32_t>(), wcount.as<uint32_t>(),
                  wplan.as<PlanRanges>(), em);
        } else
        {
          k_mum_plan<IDX, true, true>
              <<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
                  ix, qs, wlist.as<uint32_t>(), nlistword, searchlength,
                  wfirste.as<uint32_t>(), wcount.as<uint32_t>(),
                  wplan.as<PlanRanges>(), em);
        }
        VSA_HIP(shard_summary(pcursor.as<unsigned long long>(), nshards,
                              pdoff.as<uint64_t>(), psummary.as<uint64_t>(),
                              stream));
      } else
      {
        k_mum_plan<IDX, false><<<gridfor(nq), VSA_BLOCK, 0, stream>>>(
            ix, qs, wlist.as<uint32_t>(), nlistword, searchlength,
            wfirste.as<uint32_t>(), wcount.as<uint32_t>(),
            wplan.as<PlanRanges>());
      }
      VSA_HIP(hipGetLastError());
    }
    if (const char *pf = getenv("VSA_DEBUG_PLANFILE"))
    {
      // the plans of the first 65 536 queries as they stand when the search
      // kernel starts -- per query its count and VSA_PLAN_RANGES ranges
      // (first | length << 16), 32-bit words -- for bench.py, which prices
      // the kernel on exactly the searches it runs
      const uint64_t k = std::min<uint64_t>(nq, 65536);
      std::vector<uint32_t> hc(k), hp(k * VSA_PLAN_RANGES);
      VSA_HIP(hipMemcpyAsync(hc.data(), wcount.p, k * 4,
                             hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipMemcpyAsync(hp.data(), wplan.p, k * sizeof(PlanRanges),
                             hipMemcpyDeviceToHost, stream));
      VSA_HIP(hipStreamSynchronize(stream));
      if (FILE *f = fopen(pf, "wb"))
      {
        for (uint64_t q = 0; q < k; q++)
        {
          (void) fwrite(&hc[q], 4, 1, f);
          (void) fwrite(&hp[q * VSA_PLAN_RANGES], 4, VSA_PLAN_RANGES, f);
        }
        fclose(f);
      }
    }
    fromplan = true;
  }
  DevBuf doff, rawout, rawkeys, summary, blocksum, rtemp;
  const uint64_t nplanblocks = (queries->nq + 255) / 256;
  uint64_t plannedwork = 0;
  size_t rbytes = 0;
  if (fromplan)
  {
    // the work-items of the planned search: summed per workgroup by the
    // kernel, reduced behind it into the fifth word of the shard summary
    if (blocksum.alloc((vsa_grid_blocks(nplanblocks) + 1) * 8))
    {
      return -100;
    }
    VSA_HIP(rocprim::reduce(nullptr, rbytes,
                            blocksum.as<unsigned long long>(),
                            (unsigned long long *) nullptr, 0ull,
                            (size_t) nplanblocks,
                            rocprim::plus<unsigned long long>(), stream));
    if (rtemp.alloc(rbytes))
    {
      return -100;
    }
  }
  if (cursor.alloc((size_t) nshards * VSA_CURSOR_STRIDE * 8) ||
      doff.alloc(nshards * 8) || summary.alloc(5 * 8))
  {
    return -100;
  }
  // first guess: MUM modes report at most one match per work-item but
  // typically about one per query; MEM is unbounded.  The kernel counts what
  // it needs and never writes past a region's capacity; on overflow of any
  // region run again with regions of the size that was asked for.
  uint64_t shardcap =
      std::max<uint64_t>((queries->nq * 2 / nshards) * 5 / 4 + 64, 256);
  uint64_t needed = 0, maxshard = 0;
  double searchms = 0;
  for (int attempt = 0; attempt < 2; attempt++)
  {
    if (rawout.alloc(nshards * shardcap * recsize) ||
        rawkeys.alloc(nshards * shardcap * 8))
    {
      return -100;
    }
    VSA_HIP(hipMemsetAsync(cursor.p, 0,
                           (size_t) nshards * VSA_CURSOR_STRIDE * 8, stream));
    tsearch.start();
#define VSA_LAUNCH_QUERY(MUMFLAG, KEYFLAG)                                     \
  k_query_search<IDX, MUMFLAG, KEYFLAG, 256>                                  \
      <<<vsa_grid((nwork + 255) / 256), 256, 0, stream>>>(                    \
          ix, qs, dbase, perquery, nwork, searchlength,                       \
          rawout.as<vsa_match>(), rawkeys.as<uint64_t>(), shardcap,           \
          nshards - 1, cursor.as<unsigned long long>(), packbits, valbits)
    if (fromplan && memplan)
    {
      k_query_search_planned<IDX, 256, true, false, false>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (fromplan && rows)
    {
      k_query_search_planned<IDX, 256, true, true>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (fromplan && deepok)
    {
      k_query_search_planned<IDX, 256, true>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (fromplan)
    {
      k_query_search_planned<IDX, 256, false>
          <<<vsa_grid(nplanblocks), 256, 0, stream>>>(
              ix, qs, dbase, perquery, wplan.as<PlanRanges>(),
              wcount.as<uint32_t>(), searchlength, rawout.as<vsa_match>(),
              rawkeys.as<uint64_t>(), shardcap, nshards - 1,
              cursor.as<unsigned long long>(), packbits, valbits,
              blocksum.as<unsigned long long>());
    } else if (nwork > 0)
    {
      // every (query, offset) pair: MEM, and MUM batches without a plan
      // (deep locate needs the deep prefix to fit into every search)
      if (domum && deepok)
      {
        VSA_LAUNCH_QUERY(true, true);
      } else if (domum)
      {
        VSA_LAUNCH_QUERY(true, false);
      } else if (deepok)
      {
        VSA_LAUNCH_QUERY(false, true);
      } else
      {
        VSA_LAUNCH_QUERY(false, false);
      }
    }
#undef VSA_LAUNCH_QUERY
    tsearch.stop();
    VSA_HIP(hipGetLastError());
    if (fromplan)
    {
      VSA_HIP(rocprim::reduce(rtemp.p, rbytes,
                              blocksum.as<unsigned long long>(),
                              summary.as<unsigned long long>() + 4, 0ull,
                              (size_t) nplanblocks,
                              rocprim::plus<unsigned long long>(), stream));
    }
    // where each region goes in the dense list, how much there is
    VSA_HIP(shard_summary(cursor.as<unsigned long long>(), nshards,
                          doff.as<uint64_t>(), summary.as<uint64_t>(),
                          stream));
    {
      const Fetch f[7] = {{summary.as<uint64_t>(), 8},
                          {summary.as<uint64_t>() + 1, 8},
                          {summary.as<uint64_t>() + 2, 8},
                          {summary.as<uint64_t>() + 3, 8},
                          {summary.as<uint64_t>() + 4, 8},
                          {planemit ? psummary.p : summary.p, 8},
                          {nlistword != nullptr ? (const void *) nlistword
                                                : summary.p, 8}};
      uint64_t got[7];
      if (fetchwords(stream, f, 7, got))
      {
        return -100;
      }
      if (nlistword != nullptr)
      {
        nfirstpass = got[6] >> 32;
        plansearches = 2 * (got[6] & 0xFFFFFFFFull);
      }
      nplan = planemit ? got[5] : 0;
      needed = got[0];
      maxshard = got[1];
      nfirst = fromplan ? nfirstpass : 0;
      plannedwork = fromplan ? got[4] : 0;
    }
    searchms += tsearch.ms();
    if (maxshard <= shardcap)
    {
      break;
    }
    shardcap = maxshard;
  }
  if (maxshard > shardcap)
  {
    VSA_ERROR("match buffer overflow: %llu > %llu",
              (unsigned long long) maxshard, (unsigned long long) shardcap);
    return -5;
  }
  if (needed + nplan + nfirst > 0)
  {
    if (out.alloc((needed + nplan + nfirst) * recsize) ||
        keys.alloc((needed + nplan + nfirst) * 8))
    {
      return -100;
    }
    if (nplan > 0 && valbits != 0)
    {
      k_compact_shards<uint32_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          prawout.as<uint32_t>(), prawkeys.as<uint64_t>(), pcap,
          pcursor.as<unsigned long long>(), pdoff.as<uint64_t>(),
          out.as<uint32_t>() + needed, keys.as<uint64_t>() + needed);
      VSA_HIP(hipGetLastError());
    } else if (nplan > 0 && packed)
    {
      k_compact_shards<uint64_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          prawout.as<uint64_t>(), prawkeys.as<uint64_t>(), pcap,
          pcursor.as<unsigned long long>(), pdoff.as<uint64_t>(),
          out.as<uint64_t>() + needed, keys.as<uint64_t>() + needed);
      VSA_HIP(hipGetLastError());
    } else if (nplan > 0)
    {
      k_compact_shards<vsa_match><<<nshards, VSA_BLOCK, 0, stream>>>(
          prawout.as<vsa_match>(), prawkeys.as<uint64_t>(), pcap,
          pcursor.as<unsigned long long>(), pdoff.as<uint64_t>(),
          out.as<vsa_match>() + needed, keys.as<uint64_t>() + needed);
      VSA_HIP(hipGetLastError());
    }
    if (needed > 0 && valbits != 0)
    {
      k_compact_shards<uint32_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<uint32_t>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<uint32_t>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    } else if (needed > 0 && packed)
    {
      k_compact_shards<uint64_t><<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<uint64_t>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<uint64_t>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    } else if (needed > 0)
    {
      k_compact_shards<vsa_match><<<nshards, VSA_BLOCK, 0, stream>>>(
          rawout.as<vsa_match>(), rawkeys.as<uint64_t>(), shardcap,
          cursor.as<unsigned long long>(), doff.as<uint64_t>(),
          out.as<vsa_match>(), keys.as<uint64_t>());
      VSA_HIP(hipGetLastError());
    }
    if (nfirst > 0)
    {
      k_append_first<<<gridfor(queries->nq), VSA_BLOCK, 0, stream>>>(
          wfmlen.as<uint32_t>(), wfmdb.as<uint64_t>(),
          wboffset.as<uint64_t>(), queries->nq, perquery, dbase, qs.seqoffset,
          needed + nplan,
          out.as<vsa_match>(), keys.as<uint64_t>(), packbits, valbits);
      VSA_HIP(hipGetLastError());
    }
    needed += nplan + nfirst;
  }
  res->stats.candidates = domum ? needed : 0;
  if (keeppairs)
  {
    res->count = needed;
    res->packbits = lenbits;
    if (needed > 0)
    {
      res->matches = (vsa_match *) keys.release();
      res->packvals = (uint64_t *) out.release();
    }
  } else if (domum && !domumcand)
  {
    DevBuf mums;
    uint64_t nm = 0;
    if (valbits != 0)
    {
      if (mumfilter_packed<uint32_t>(keys.as<const uint64_t>(),
                                     out.as<const uint32_t>(), needed, lenbits, dbbits,
                                     stream, mums, &nm, &mumsum, 0, valbits,
                                     qs.seqoffset))
      {
        return -100;
      }
    } else if (packed)
    {
      if (mumfilter_packed(keys.as<const uint64_t>(),
                           out.as<const uint64_t>(), needed, lenbits, dbbits, stream, mums,
                           &nm, &mumsum))
      {
        return -100;
      }
    } else if (mumuniqueinquery(out, needed, stream, mums, &nm, 0, nullptr,
                                index->n, queries->maxlength, &mumsum))
    {
      return -100;
    }
    res->count = nm;
    res->matches = (vsa_match *) mums.release();
  } else if (needed > 0 && !ordered && domum)
  {
    // candidates for a filter that sorts them anyway (multi-GPU -mum): as
    // they lie
    res->count = needed;
    res->matches = (vsa_match *) out.release();
  } else if (needed > 0)
  {
    // reference order = work-item order; appends of one work-item are
    // contiguous and in order, the radix sort is stable
    DevBuf sk, sm;
    if (sk.alloc(needed * 8) || sm.alloc(needed * sizeof(vsa_match)))
    {
      return -100;
    }
    if (sortbykey(keys.as<uint64_t>(), sk.as<uint64_t>(),
                  out.as<vsa_match>(), sm.as<vsa_match>(), needed,
                  bitsfor(nitems), stream))
    {
      return -100;
    }
    res->count = needed;
    res->matches = (vsa_match *) sm.release();
  }
  tall.stop();
  VSA_HIP(hipStreamSynchronize(stream));
  res->stats.count = res->count;
  res->stats.search_kernel_ms = searchms;
  res->stats.anchor_ms = 0; // (the anchor pass of round 1 is gone)
  res->stats.first_kernel_ms = tfirst.ms();
  if (fromplan)
  {
    nwork = plannedwork;
    res->stats.searches = nwork + queries->nq + plansearches;
  }
  res->stats.kernel_searches = nwork;
  res->stats.total_device_ms = tall.ms();
  if (mumsum != ~0ull)
  {
    res->stats.sumlength = mumsum; // the filter summed the lengths already
    return 0;
  }
  if (keeppairs)
  {
    res->stats.sumlength = 0; // of candidates: nobody asks
    return 0;
  }
  return sumlengths(res->matches, res->count, stream, &res->stats.sumlength);
}

} // namespace

// ---------------------------------------------------------------------------
// C ABI
// ---------------------------------------------------------------------------

extern "C" int vsa_findcompletematches(const vsa_index *index,
                                       const vsa_queries *queries,
                                       vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findcompletematches: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (queries->device != index->device)
  {
    VSA_ERROR("queries live on device %d, index on device %d",
              queries->device, index->device);
    return -1;
  }
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (queries->maxlength > 0xFFFFFFF0ull)
  {
    VSA_ERROR("query length beyond 32 bits is not supported");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  // Vmengine/exactcompl.c:179-185: the first query shorter than
  // prefixlength stops the run; queries before it are still matched
  uint64_t qlimit = queries->nq;
  bool shortquery = false;
  if (queries->minlength < index->pl)
  {
    for (uint64_t q = 0; q < queries->nq; q++)
    {
      if (queries->hlength[q] < index->pl)
      {
        qlimit = q;
        shortquery = true;
        break;
      }
    }
  }
  vsa_result *res = newresult(index->device);
  const int rc = (index->isize == 4)
                     ? run_complete<uint32_t>(index, queries, qlimit, res)
                     : run_complete<uint64_t>(index, queries, qlimit, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  if (shortquery)
  {
    VSA_ERROR("patternlength=%lu must be >= %lu=prefixlen",
              (unsigned long) queries->hlength[qlimit],
              (unsigned long) index->pl);
    return -2;
  }
  return 0;
}

extern "C" int vsa_findmumcandidates(const vsa_index *index,
                                     const vsa_queries *queries,
                                     uint64_t searchlength, int ordered,
                                     vsa_result **result)
{
  if (ordered)
  {
    return vsa_findquerymatches(index, queries, 1, 1, searchlength, result);
  }
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmumcandidates: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (searchlength < index->pl || searchlength > 0xFFFFFFF0ull)
  {
    // Vmengine/fquery.c:440-446
    VSA_ERROR("searchlength=%lu must be >= %lu=prefixlen",
              (unsigned long) searchlength, (unsigned long) index->pl);
    return -2;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc =
      (index->isize == 4)
          ? run_query<uint32_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false)
          : run_query<uint64_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findmumcandidates_packed(const vsa_index *index,
                                            const vsa_queries *queries,
                                            uint64_t searchlength,
                                            uint32_t lengthbits,
                                            vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findmumcandidates_packed: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (searchlength < index->pl || searchlength > 0xFFFFFFF0ull)
  {
    // Vmengine/fquery.c:440-446
    VSA_ERROR("searchlength=%lu must be >= %lu=prefixlen",
              (unsigned long) searchlength, (unsigned long) index->pl);
    return -2;
  }
  if (lengthbits == 0)
  {
    lengthbits = bitsfor(queries->maxlength);
  }
  if (lengthbits > 16)
  {
    VSA_ERROR("packed candidates: at most 16 length bits");
    return -2;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const int rc =
      (index->isize == 4)
          ? run_query<uint32_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false,
                                lengthbits)
          : run_query<uint64_t>(index, queries, true, true,
                                (uint32_t) searchlength, res, false,
                                lengthbits);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_findquerymatches(const vsa_index *index,
                                    const vsa_queries *queries,
                                    int domaximaluniquematch,
                                    int domaximaluniquematchcandidates,
                                    uint64_t searchlength,
                                    vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findquerymatches: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (queries->device != index->device)
  {
    VSA_ERROR("queries live on device %d, index on device %d",
              queries->device, index->device);
    return -1;
  }
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  // Vmengine/fquery.c:440-446
  if (searchlength < index->pl)
  {
    VSA_ERROR("searchlength=%lu must be >= %lu=prefixlen",
              (unsigned long) searchlength, (unsigned long) index->pl);
    return -2;
  }
  if (searchlength > 0xFFFFFFFFull || queries->maxlength > 0xFFFFFFF0ull)
  {
    VSA_ERROR("query or search length beyond 32 bits is not supported");
    return -3;
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  const bool mum = domaximaluniquematch != 0,
             cand = domaximaluniquematchcandidates != 0;
  const int rc =
      (index->isize == 4)
          ? run_query<uint32_t>(index, queries, mum, cand,
                                (uint32_t) searchlength, res)
          : run_query<uint64_t>(index, queries, mum, cand,
                                (uint32_t) searchlength, res);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  return 0;
}

static int mumfilter_entry(void *device_candidates, uint64_t ncandidates,
                           int device, uint64_t carry, vsa_result **result)
{
  if (result == nullptr || (device_candidates == nullptr && ncandidates > 0))
  {
    VSA_ERROR("vsa_mumuniqueinquery: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(device);
  hipStream_t stream = nullptr; // default stream: no index handle here
  vsa_dev_set_stream(stream);
  Timer tall(stream);
  DevBuf cand, mums;
  cand.p = device_candidates; // borrowed, released below
  uint64_t nm = 0;
  tall.start();
  const int rc = mumuniqueinquery(cand, ncandidates, stream, mums, &nm, carry);
  tall.stop();
  cand.release();
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  (void) hipStreamSynchronize(stream);
  res->count = nm;
  res->matches = (vsa_match *) mums.release();
  res->stats.count = nm;
  res->stats.candidates = ncandidates;
  res->stats.total_device_ms = tall.ms();
  if (sumlengths(res->matches, nm, stream, &res->stats.sumlength) != 0)
  {
    vsa_result_free(res);
    return -100;
  }
  *result = res;
  return 0;
}

extern "C" int vsa_mumuniqueinquery(void *device_candidates,
                                    uint64_t ncandidates, int device,
                                    vsa_result **result)
{
  return mumfilter_entry(device_candidates, ncandidates, device, 0, result);
}

extern "C" int vsa_mumuniqueinquery_range(void *device_candidates,
                                          uint64_t ncandidates, int device,
                                          uint64_t carry_dbright,
                                          vsa_result **result)
{
  return mumfilter_entry(device_candidates, ncandidates, device,
                         carry_dbright, result);
}

// word 0 / word 1 of row i of (key, value) pairs
// word 0 (key) or 1 (value) of row i of two lists of rows laid end to end
struct RowWord
{
  const uint64_t *rows, *more;
  uint64_t nrows; // in `rows`
  uint32_t word;
  __device__ uint64_t operator()(size_t i) const
  {
    return i < nrows ? rows[2 * i + word] : more[2 * (i - nrows) + word];
  }
};

// pairs -> records, in place order (vsa_result_fetch of a packed result)
__global__ void __launch_bounds__(VSA_BLOCK)
k_unpack_pairs(const uint64_t *__restrict__ key,
               const uint64_t *__restrict__ val, uint64_t n,
               uint32_t packbits, vsa_match *__restrict__ out)
{
  const uint64_t t = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (t < n)
  {
    const uint64_t k = key[t], v = val[t], mask = (1ull << packbits) - 1;
    vsa_match m;
    m.length = mask - (k & mask);
    m.dbstart = k >> packbits;
    m.queryseq = v >> 16;
    m.querystart = v & 0xFFFFu;
    out[t] = m;
  }
}

int vsa_unpack_result(const vsa_result *r, uint64_t count, vsa_match *device)
{
  k_unpack_pairs<<<gridfor(count), VSA_BLOCK>>>(
      reinterpret_cast<const uint64_t *>(r->matches), r->packvals, count,
      r->packbits, device);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int vsa_mumuniqueinquery_range_packed2(const void *device_rows,
                                                  uint64_t nrows,
                                                  const void *more_rows,
                                                  uint64_t nmore,
                                                  uint32_t lengthbits,
                                                  uint64_t totallength,
                                                  int device,
                                                  uint64_t carry_dbright,
                                                  vsa_result **result)
{
  if (result == nullptr || (device_rows == nullptr && nrows > 0) ||
      (more_rows == nullptr && nmore > 0) || lengthbits == 0 ||
      lengthbits > 16 || lengthbits + bitsfor(totallength) > 64)
  {
    VSA_ERROR("vsa_mumuniqueinquery_range_packed: bad argument");
    return -1;
  }
  *result = nullptr;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(device);
  hipStream_t stream = nullptr; // default stream: no index handle here
  vsa_dev_set_stream(stream);
  Timer tall(stream);
  DevBuf mums;
  uint64_t nm = 0, sum = 0;
  const uint64_t total = nrows + nmore;
  tall.start();
  int rc = 0;
  if (total > 0)
  {
    // the sort reads the rows as they lie (the first pass of the radix sort
    // takes iterators): no split into two arrays, no copy of the two lists
    // into one
    const uint64_t *rows = reinterpret_cast<const uint64_t *>(device_rows),
                   *more = reinterpret_cast<const uint64_t *>(more_rows);
    auto rowkeys = rocprim::make_transform_iterator(
        rocprim::counting_iterator<size_t>(0), RowWord{rows, more, nrows, 0});
    auto rowvals = rocprim::make_transform_iterator(
        rocprim::counting_iterator<size_t>(0), RowWord{rows, more, nrows, 1});
    rc = mumfilter_packed<uint64_t>(rowkeys, rowvals, total, lengthbits,
                                    bitsfor(totallength), stream, mums, &nm,
                                    &sum, carry_dbright);
  }
  tall.stop();
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  (void) hipStreamSynchronize(stream);
  res->count = nm;
  res->matches = (vsa_match *) mums.release();
  res->stats.count = nm;
  res->stats.candidates = total;
  res->stats.sumlength = sum;
  res->stats.total_device_ms = tall.ms();
  *result = res;
  return 0;
}

extern "C" int vsa_mumuniqueinquery_range_packed(const void *device_rows,
                                                 uint64_t nrows,
                                                 uint32_t lengthbits,
                                                 uint64_t totallength,
                                                 int device,
                                                 uint64_t carry_dbright,
                                                 vsa_result **result)
{
  return vsa_mumuniqueinquery_range_packed2(device_rows, nrows, nullptr, 0,
                                            lengthbits, totallength, device,
                                            carry_dbright, result);
}
