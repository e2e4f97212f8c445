// vmatch -complete -e K | -h K (and Kp, Kb): host side and C ABI of approximate
// complete matching (Vmengine/fcomplete.c:140-261, approxcompl.c:138,
// splitesaapm.c:458); kernels in approx_search.inc (pigeonhole path) and
// approx_tree.inc (lcp-interval tree path).
#include "search_host.hpp"
#include <rocprim/rocprim.hpp>

namespace
{

#include "search_complete.inc"
#include "approx_search.inc"
#include "approx_tree.inc"

} // namespace

// ---- batches whose thresholds (-e Kp / -h Kp) are 0 for the short reads and
// > 0 for the long ones: the reference sends the former through the exact
// search and the latter through splitesaapm, read by read
// (Vmengine/approxcompl.c:167-191).  Here the batch is cut into the two kinds,
// each kind runs as a batch of its own over the same symbols, and the two
// lists are merged back into query order.

__global__ void __launch_bounds__(VSA_BLOCK)
k_subquery_gather(const uint64_t *__restrict__ start,
                  const uint64_t *__restrict__ length,
                  const uint64_t *__restrict__ which, uint64_t n,
                  uint64_t *__restrict__ substart,
                  uint64_t *__restrict__ sublength)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t q = which[i];
    substart[i] = start[q];
    sublength[i] = length[q];
  }
}

// rows [0, nfirst) come from the sub-batch `whicha`, the others from
// `whichb`; their queryseq becomes the number in the whole batch
__global__ void __launch_bounds__(VSA_BLOCK)
k_subquery_renumber(vsa_match *__restrict__ rows, uint64_t nfirst,
                    uint64_t n, const uint64_t *__restrict__ whicha,
                    const uint64_t *__restrict__ whichb, uint64_t seqoffset,
                    uint32_t *__restrict__ keys, uint32_t *__restrict__ index)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    const uint64_t sub = rows[i].queryseq;
    const uint64_t q = (i < nfirst) ? whicha[sub] : whichb[sub];
    rows[i].queryseq = q + seqoffset;
    keys[i] = (uint32_t) q;
    index[i] = (uint32_t) i;
  }
}

namespace
{

struct SubQueries
{
  vsa_queries q;
  DevBuf start, length, which;
};

// the queries `which` (ascending numbers) of a batch as a batch that shares
// the symbols
int make_subqueries(const vsa_index *index, const vsa_queries *queries,
                    const std::vector<uint64_t> &which, SubQueries &sub)
{
  const uint64_t n = which.size();

  sub.q.device = queries->device;
  sub.q.nq = n;
  sub.q.nsymbols = queries->nsymbols;
  sub.q.symbols = queries->symbols;
  sub.q.seqoffset = 0;
  sub.q.dense = false;
  sub.q.hlength.resize(n);
  sub.q.minlength = n ? ~0ull : 0;
  sub.q.maxlength = 0;
  for (uint64_t i = 0; i < n; i++)
  {
    const uint64_t m = queries->uniform ? queries->maxlength
                                        : queries->hlength[which[i]];
    sub.q.hlength[i] = m;
    sub.q.minlength = std::min(sub.q.minlength, m);
    sub.q.maxlength = std::max(sub.q.maxlength, m);
  }
  sub.q.uniform = n != 0 && sub.q.minlength == sub.q.maxlength;
  vsa_dev_set_stream(index->stream);
  if (sub.start.alloc(n * 8) || sub.length.alloc(n * 8) ||
      sub.which.alloc(n * 8))
  {
    return -100;
  }
  sub.q.start = sub.start.as<uint64_t>();
  sub.q.length = sub.length.as<uint64_t>();
  if (n > 0)
  {
    VSA_HIP(hipMemcpyAsync(sub.which.p, which.data(), n * 8,
                           hipMemcpyHostToDevice, index->stream));
    k_subquery_gather<<<gridfor(n), VSA_BLOCK, 0, index->stream>>>(
        queries->start, queries->length, sub.which.as<uint64_t>(), n,
        sub.q.start, sub.q.length);
    VSA_HIP(hipGetLastError());
    VSA_HIP(hipStreamSynchronize(index->stream));
  }
  return 0;
}

int approx_batch(const vsa_index *index, const vsa_queries *queries,
                 int doedist, uint64_t distvalue, int percent,
                 vsa_result **result);

// explicitk (the second pass of a "best of" job): the threshold of every read
// instead of distvalue percent of its length; VSA_NO_THRESHOLD = the read is
// left out
#define VSA_NO_THRESHOLD 0xFFFFFFFFu
int approx_mixed(const vsa_index *index, const vsa_queries *queries,
                 int doedist, uint64_t distvalue, vsa_result **result,
                 const std::vector<uint32_t> *explicitk = nullptr)
{
  const uint64_t nq = queries->nq;
  std::vector<uint64_t> exact, approx;
  std::vector<uint32_t> approxk;
  uint64_t qlimit = nq, failk = 0, failm = 0;
  bool failshort = false;

  if (nq >= 0xFFFFFFFFull)
  {
    VSA_ERROR("a batch of %lu reads that mixes thresholds 0 and > 0 is not "
              "covered by the GPU engine", (unsigned long) nq);
    return VSA_NOT_COVERED;
  }
  for (uint64_t q = 0; q < nq; q++)
  {
    const uint64_t m = queries->uniform ? queries->maxlength
                                        : queries->hlength[q],
                   k = explicitk != nullptr ? (*explicitk)[q]
                                            : (m * distvalue) / 100;
    if (explicitk != nullptr && k == VSA_NO_THRESHOLD)
    {
      continue;
    }
    if (k == 0)
    {
      if (m < index->pl)
      {
        // exactcompl.c:179-185
        qlimit = q;
        failshort = true;
        failm = m;
        break;
      }
      exact.push_back(q);
    } else
    {
      if (k >= m)
      {
        // splitesaapm.c:496-501
        qlimit = q;
        failk = k;
        failm = m;
        break;
      }
      approx.push_back(q);
      approxk.push_back((uint32_t) k);
    }
  }
  SubQueries sa, sb;
  vsa_result *ra = nullptr, *rb = nullptr;
  int rc = 0;
  if (!exact.empty())
  {
    rc = make_subqueries(index, queries, exact, sa);
    if (rc == 0)
    {
      rc = vsa_findcompletematches(index, &sa.q, &ra);
    }
  }
  if (rc == 0 && !approx.empty())
  {
    rc = make_subqueries(index, queries, approx, sb);
    if (rc == 0)
    {
      apm_explicitk = explicitk != nullptr ? approxk.data() : nullptr;
      rc = approx_batch(index, &sb.q, doedist, distvalue, 1, &rb);
      apm_explicitk = nullptr;
    }
  }
  if (rc != 0)
  {
    vsa_result_free(ra);
    vsa_result_free(rb);
    return rc;
  }
  if (vsa_set_device(index->device) != 0)
  {
    vsa_result_free(ra);
    vsa_result_free(rb);
    return -100;
  }
  hipStream_t stream = index->stream;
  vsa_dev_set_stream(stream);
  vsa_result *res = newresult(index->device);
  const uint64_t na = ra ? ra->count : 0, nb = rb ? rb->count : 0,
                 n = na + nb;
  // (a macro that returns would leak the three lists)
  auto merge = [&]() -> int {
    DevBuf all, merged, keys, keys2, order, order2, temp;
    size_t tb = 0;
    if (n == 0)
    {
      return 0;
    }
    if (n >= 0xFFFFFFFFull)
    {
      VSA_ERROR("%lu matches of a batch that mixes thresholds 0 and > 0 are "
                "not covered by the GPU engine", (unsigned long) n);
      return VSA_NOT_COVERED;
    }
    if (all.alloc(n * sizeof(vsa_match)) ||
        merged.alloc(n * sizeof(vsa_match)) || keys.alloc(n * 4) ||
        keys2.alloc(n * 4) || order.alloc(n * 4) || order2.alloc(n * 4))
    {
      return -100;
    }
    if (na > 0)
    {
      VSA_HIP(hipMemcpyAsync(all.p, ra->matches, na * sizeof(vsa_match),
                             hipMemcpyDeviceToDevice, stream));
    }
    if (nb > 0)
    {
      VSA_HIP(hipMemcpyAsync(all.as<vsa_match>() + na, rb->matches,
                             nb * sizeof(vsa_match), hipMemcpyDeviceToDevice,
                             stream));
    }
    k_subquery_renumber<<<gridfor(n), VSA_BLOCK, 0, stream>>>(
        all.as<vsa_match>(), na, n, sa.which.as<uint64_t>(),
        sb.which.as<uint64_t>(), queries->seqoffset, keys.as<uint32_t>(),
        order.as<uint32_t>());
    VSA_HIP(hipGetLastError());
    VSA_HIP(rocprim::radix_sort_pairs(
        nullptr, tb, keys.as<uint32_t>(), keys2.as<uint32_t>(),
        order.as<uint32_t>(), order2.as<uint32_t>(), (size_t) n, 0u,
        bitsfor(nq), stream));
    if (temp.alloc(tb))
    {
      return -100;
    }
    VSA_HIP(rocprim::radix_sort_pairs(
        temp.p, tb, keys.as<uint32_t>(), keys2.as<uint32_t>(),
        order.as<uint32_t>(), order2.as<uint32_t>(), (size_t) n, 0u,
        bitsfor(nq), stream));
    VSA_HIP(gather_matches(all.as<vsa_match>(), order2.as<uint32_t>(), n,
                           merged.as<vsa_match>(), stream));
    VSA_HIP(hipStreamSynchronize(stream));
    res->matches = (vsa_match *) merged.release();
    return 0;
  };
  rc = merge();
  res->count = res->stats.count = n;
  for (const vsa_result *r : {(const vsa_result *) ra,
                              (const vsa_result *) rb})
  {
    if (r != nullptr)
    {
      res->stats.sumlength += r->stats.sumlength;
      res->stats.searches += r->stats.searches;
      res->stats.kernel_searches += r->stats.kernel_searches;
      res->stats.search_kernel_ms += r->stats.search_kernel_ms;
      res->stats.total_device_ms += r->stats.total_device_ms;
      res->stats.first_kernel_ms += r->stats.first_kernel_ms;
    }
  }
  vsa_result_free(ra);
  vsa_result_free(rb);
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  if (qlimit < nq)
  {
    // the reads before the failing one have been answered
    if (failshort)
    {
      VSA_ERROR("patternlength=%lu must be >= %lu=prefixlen",
                (unsigned long) failm, (unsigned long) index->pl);
    } else
    {
      VSA_ERROR("threshold=%lu>=%lu=patternlen not allowed",
                (unsigned long) failk, (unsigned long) failm);
    }
    return -2;
  }
  return 0;
}

} // namespace

// best[q - seqoffset] = the smallest distance among the matches of read q
// (the distance of a match travels in its querystart field)
__global__ void __launch_bounds__(VSA_BLOCK)
k_best_distance(const vsa_match *__restrict__ matches, uint64_t n,
                uint64_t seqoffset, uint32_t *__restrict__ best)
{
  const uint64_t i = vsa_bid() * VSA_BLOCK + threadIdx.x;
  if (i < n)
  {
    atomicMin(best + (matches[i].queryseq - seqoffset),
              (uint32_t) matches[i].querystart);
  }
}

namespace
{

// vmatch -complete -e Kb | -h Kb, "best of" (Vmengine/initcompl.c:59-77): read
// by read -- decidefcm restores the job's K in front of every read,
// Vmengine/fcomplete.c:251-252 -- the reference looks for the smallest
// threshold t <= m K / 100 at which the read has a match at all (a binary
// search over existence checks, Vmengine/approxcompl.c:80-122) and then
// reports the read's matches at threshold t; a read without a match within
// m K / 100 reports nothing.  Here: one pass at the percent thresholds gives
// every read's smallest distance, a second pass runs every read at exactly
// that threshold (the regions, and with them the order of the matches, are
// those of the threshold: Vmengine/splitesaapm.c:458-558).
int approx_bestof(const vsa_index *index, const vsa_queries *queries,
                  int doedist, uint64_t distvalue, vsa_result **result)
{
  const uint64_t nq = queries->nq;
  vsa_result *first = nullptr;
  *result = nullptr;
  if (nq >= 0xFFFFFFFFull)
  {
    VSA_ERROR("a best-of batch of %lu reads is not covered by the GPU engine",
              (unsigned long) nq);
    return VSA_NOT_COVERED;
  }
  int rc = vsa_findapproxcompletematches(index, queries, doedist, distvalue,
                                         1, &first);
  if (rc != 0)
  {
    vsa_result_free(first);
    return rc;
  }
  std::vector<uint32_t> best(nq, VSA_NO_THRESHOLD);
  {
    hipStream_t stream = index->stream;
    vsa_dev_set_stream(stream);
    DevBuf dbest;
    if (vsa_set_device(index->device) != 0 || dbest.alloc((nq + 1) * 4))
    {
      vsa_result_free(first);
      return -100;
    }
    auto run = [&]() -> int {
      VSA_HIP(hipMemsetAsync(dbest.p, 0xFF, (nq + 1) * 4, stream));
      if (first->count > 0)
      {
        k_best_distance<<<gridfor(first->count), VSA_BLOCK, 0, stream>>>(
            first->matches, first->count, queries->seqoffset,
            dbest.as<uint32_t>());
        VSA_HIP(hipGetLastError());
      }
      if (nq > 0)
      {
        VSA_HIP(hipMemcpyAsync(best.data(), dbest.p, nq * 4,
                               hipMemcpyDeviceToHost, stream));
      }
      VSA_HIP(hipStreamSynchronize(stream));
      return 0;
    };
    rc = run();
  }
  const vsa_stats s1 = first->stats;
  vsa_result_free(first);
  if (rc != 0)
  {
    return rc;
  }
  // an exact match found by the first pass has distance 0 whichever way it
  // was found (the percent threshold of a short read is 0: exact search,
  // whose matches carry querystart 0 as well)
  rc = approx_mixed(index, queries, doedist, distvalue, result, &best);
  if (*result != nullptr)
  {
    (*result)->stats.searches += s1.searches;
    (*result)->stats.total_device_ms += s1.total_device_ms;
  }
  return rc;
}

} // namespace

extern "C" int vsa_findapproxcompletematches(const vsa_index *index,
                                             const vsa_queries *queries,
                                             int doedist, uint64_t distvalue,
                                             int percent,
                                             vsa_result **result)
{
  if (index == nullptr || queries == nullptr || result == nullptr)
  {
    VSA_ERROR("vsa_findapproxcompletematches: NULL argument");
    return -1;
  }
  *result = nullptr;
  if (queries->device != index->device)
  {
    VSA_ERROR("queries live on device %d, index on device %d",
              queries->device, index->device);
    return -1;
  }
  if (queries->rows != nullptr)
  {
    // a packed batch: the approximate kernels read bytes
    if (vsa_set_device(index->device) != 0 ||
        vsa_queries_bytes(queries, index->stream) != 0)
    {
      return -100;
    }
  }
  if (percent == 2)
  {
    if (index->numofchars != 4)
    {
      VSA_ERROR("approximate search on alphabets of %lu symbols is not "
                "covered by the GPU engine",
                (unsigned long) index->numofchars);
      return VSA_NOT_COVERED;
    }
    return approx_bestof(index, queries, doedist, distvalue, result);
  }
  if (percent != 0 && index->bck != nullptr && index->numofchars == 4 &&
      (queries->minlength * distvalue) / 100 == 0 &&
      (queries->maxlength * distvalue) / 100 != 0)
  {
    return approx_mixed(index, queries, doedist, distvalue, result);
  }
  return approx_batch(index, queries, doedist, distvalue, percent, result);
}

namespace
{

int approx_batch(const vsa_index *index, const vsa_queries *queries,
                 int doedist, uint64_t distvalue, int percent,
                 vsa_result **result)
{
  *result = nullptr;
  if (index->bck == nullptr)
  {
    VSA_ERROR("table bck is not loaded");
    return -3;
  }
  if (index->numofchars != 4)
  {
    VSA_ERROR("approximate search on alphabets of %lu symbols is not covered "
              "by the GPU engine", (unsigned long) index->numofchars);
    return VSA_NOT_COVERED;
  }
  ApmPlan plan;
  int rc = apm_plan(index, queries, doedist != 0, distvalue, percent != 0,
                    plan);
  if (rc != 0 && rc != VSA_NOT_COVERED)
  {
    return rc;
  }
  if (rc == 0 && plan.allexact)
  {
    // approxcompl.c:167-176: threshold 0 is the exact search
    return vsa_findcompletematches(index, queries, result);
  }
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_result *res = newresult(index->device);
  if (rc == 0)
  {
    rc = (index->isize == 4)
             ? run_approx<uint32_t>(index, queries, doedist != 0, plan, res)
             : run_approx<uint64_t>(index, queries, doedist != 0, plan, res);
  }
  if (rc == VSA_NOT_COVERED)
  {
    // pieces with a threshold of their own, patterns that are not cut,
    // Hamming distance with wildcards in a read: the reference's esaapm /
    // esahamming configurations (approx_tree.inc)
    TreePlan tplan;
    vsa_result_free(res);
    res = nullptr;
    rc = apm_treeplan(index, queries, doedist != 0, distvalue, percent != 0,
                      tplan);
    if (rc != 0)
    {
      return rc;
    }
    res = newresult(index->device);
    rc = (index->isize == 4)
             ? run_approx_tree<uint32_t>(index, queries, doedist != 0, tplan,
                                         res)
             : run_approx_tree<uint64_t>(index, queries, doedist != 0, tplan,
                                         res);
    plan.qlimit = tplan.qlimit;
    plan.failk = tplan.failk;
    plan.failm = tplan.failm;
  }
  if (rc != 0)
  {
    vsa_result_free(res);
    return rc;
  }
  *result = res;
  if (plan.qlimit < queries->nq)
  {
    // splitesaapm.c:496-501; the queries before it have been answered
    VSA_ERROR("threshold=%lu>=%lu=patternlen not allowed",
              (unsigned long) plan.failk, (unsigned long) plan.failm);
    return -2;
  }
  return 0;
}

} // namespace
