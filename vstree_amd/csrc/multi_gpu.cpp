// The query path on all GPUs of one node: include/vstree_amd_multi.h.
//
// Host code only (no kernels): one std::thread per replica drives the
// single-GPU entry points of libvstree_amd.so on its device; the exchange of
// `vmatch -mum` moves candidate rows between GPUs with hipMemcpyPeerAsync (a
// direct xGMI hop between two GPUs of a node); the match counters are summed
// with one ncclAllReduce (RCCL) when every replica has a GPU of its own.
// SURVEY.md 8e; reference semantics: Vmengine/fcomplete.c:313-319,
// Vmengine/fquery.c:468-475, kurtz/cleanMUMcand.c:55-118.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "vstree_amd_multi.h"

extern "C" char *vsa_errbuf(); // the calling thread's message buffer

namespace
{

const size_t kErrSize = 1024;

void seterror(const std::string &s)
{
  snprintf(vsa_errbuf(), kErrSize, "%s", s.c_str());
}

// contiguous block of `rank`: blocks differ by at most one query
void shard(uint64_t total, uint32_t rank, uint32_t world, uint64_t &first,
           uint64_t &count)
{
  const uint64_t base = total / world, extra = total % world;
  first = rank * base + std::min<uint64_t>(rank, extra);
  count = base + (rank < extra ? 1 : 0);
}

struct RankOut
{
  int rc = 0;
  std::string message;
  std::vector<vsa_match> matches;
  vsa_stats stats;
  // -mum: this rank's candidates grouped by receiving rank (device memory)
  void *sendbuf = nullptr;
  std::vector<uint64_t> counts, maxright;
  uint64_t ncand = 0;
};

} // namespace

struct vsa_multi
{
  std::vector<vsa_index *> ix;
  std::vector<int> dev;
  std::vector<ncclComm_t> comms; // empty: counters are summed on the host
  std::vector<hipStream_t> streams;
  std::vector<unsigned long long *> counters; // device, 4 words per replica
  int usedrccl = 0;
};

namespace
{

bool distinct(const std::vector<int> &d)
{
  std::vector<int> s(d);
  std::sort(s.begin(), s.end());
  return std::adjacent_find(s.begin(), s.end()) == s.end();
}

// RCCL communicators for the counter reduction: one per replica, all in this
// process.  Replicas that share a device (tests on a one-GPU box) cannot form
// a communicator; their counters are summed on the host.
void initcomms(vsa_multi *m)
{
  const char *off = getenv("VSA_MULTI_RCCL");
  if (!distinct(m->dev) || (off != nullptr && strcmp(off, "0") == 0))
  {
    return;
  }
  std::vector<ncclComm_t> comms(m->dev.size());
  if (ncclCommInitAll(comms.data(), (int) m->dev.size(), m->dev.data()) !=
      ncclSuccess)
  {
    (void) hipGetLastError();
    return;
  }
  m->comms = comms;
  m->streams.resize(m->dev.size());
  m->counters.resize(m->dev.size());
  for (size_t r = 0; r < m->dev.size(); r++)
  {
    (void) hipSetDevice(m->dev[r]);
    (void) hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking);
    (void) hipMalloc((void **) &m->counters[r], 4 * sizeof(unsigned long long));
  }
}

// sums 4 counters per replica over all replicas: RCCL when there are
// communicators, the host otherwise.  Every replica ends with the totals.
int reducecounters(vsa_multi *m, std::vector<RankOut> &out, vsa_stats *total)
{
  const size_t world = m->dev.size();
  unsigned long long sum[4] = {0, 0, 0, 0};
  m->usedrccl = 0;
  if (!m->comms.empty())
  {
    bool ok = true;
    for (size_t r = 0; r < world && ok; r++)
    {
      const unsigned long long mine[4] = {
          out[r].stats.count, out[r].stats.sumlength, out[r].stats.searches,
          out[r].stats.candidates};
      ok = hipSetDevice(m->dev[r]) == hipSuccess &&
           hipMemcpyAsync(m->counters[r], mine, sizeof mine,
                          hipMemcpyHostToDevice, m->streams[r]) == hipSuccess &&
           hipStreamSynchronize(m->streams[r]) == hipSuccess;
    }
    if (ok)
    {
      ok = ncclGroupStart() == ncclSuccess;
      for (size_t r = 0; r < world && ok; r++)
      {
        ok = ncclAllReduce(m->counters[r], m->counters[r], 4, ncclUint64,
                           ncclSum, m->comms[r], m->streams[r]) == ncclSuccess;
      }
      ok = ncclGroupEnd() == ncclSuccess && ok;
    }
    for (size_t r = 0; r < world && ok; r++)
    {
      ok = hipSetDevice(m->dev[r]) == hipSuccess &&
           hipStreamSynchronize(m->streams[r]) == hipSuccess;
    }
    if (ok)
    {
      ok = hipSetDevice(m->dev[0]) == hipSuccess &&
           hipMemcpy(sum, m->counters[0], sizeof sum,
                     hipMemcpyDeviceToHost) == hipSuccess;
    }
    if (!ok)
    {
      seterror("vsa_multi: RCCL all-reduce of the match counters failed");
      return -100;
    }
    m->usedrccl = 1;
  } else
  {
    for (size_t r = 0; r < world; r++)
    {
      sum[0] += out[r].stats.count;
      sum[1] += out[r].stats.sumlength;
      sum[2] += out[r].stats.searches;
      sum[3] += out[r].stats.candidates;
    }
  }
  if (total != nullptr)
  {
    memset(total, 0, sizeof *total);
    total->count = sum[0];
    total->sumlength = sum[1];
    total->searches = sum[2];
    total->candidates = sum[3];
    for (size_t r = 0; r < world; r++)
    {
      total->search_kernel_ms =
          std::max(total->search_kernel_ms, out[r].stats.search_kernel_ms);
      total->total_device_ms =
          std::max(total->total_device_ms, out[r].stats.total_device_ms);
      total->kernel_searches += out[r].stats.kernel_searches;
    }
  }
  return 0;
}

void takeerror(RankOut &o, int rc)
{
  o.rc = rc;
  o.message = vsa_messagespace();
}

// fetches the list of a result into host memory
int takematches(vsa_result *res, RankOut &o)
{
  const uint64_t c = vsa_result_count(res);
  o.matches.resize(c);
  if (c > 0 && vsa_result_fetch(res, o.matches.data(), c) != 0)
  {
    return -100;
  }
  return 0;
}

struct Job
{
  vsa_multi *m;
  int mode;
  uint64_t searchlength;
  const uint8_t *symbols;
  const uint64_t *start, *length;
  uint64_t nq, totallength;
  uint32_t lengthbits; // -mum: 0 = records instead of pairs
  // VSA_MULTI_APPROX (vsa_multi_findapproxcompletematches)
  int doedist, percent;
  uint64_t distvalue;
};

// -complete -e K | -h K: a mode of this file only (the public modes end at
// VSA_MULTI_MUM)
#define VSA_MULTI_APPROX (VSA_MULTI_MUM + 1)

// phase 1 of replica r: upload its block of queries, search
void searchblock(const Job &job, uint32_t r, RankOut &o)
{
  vsa_multi *m = job.m;
  const uint32_t world = (uint32_t) m->dev.size();
  uint64_t first, count;
  shard(job.nq, r, world, first, count);
  memset(&o.stats, 0, sizeof o.stats);
  if (hipSetDevice(m->dev[r]) != hipSuccess)
  {
    o.rc = -100;
    o.message = "hipSetDevice failed";
    return;
  }
  // the block's symbols: from the start of its first query to the end of its
  // last one; starts relative to that
  // (the queries of a Multiseq may lie in any order in the buffer:
  // vsa_multi_findmatches has checked every one against nsymbols)
  std::vector<uint64_t> st(count + 1), ln(count + 1);
  uint64_t lo = 0, hi = 0;
  if (count > 0)
  {
    lo = ~0ull;
    for (uint64_t i = 0; i < count; i++)
    {
      lo = std::min(lo, job.start[first + i]);
      hi = std::max(hi, job.start[first + i] + job.length[first + i]);
    }
  }
  for (uint64_t i = 0; i < count; i++)
  {
    st[i] = job.start[first + i] - lo;
    ln[i] = job.length[first + i];
  }
  vsa_queries *q = nullptr;
  vsa_result *res = nullptr;
  int rc = vsa_queries_from_host(job.symbols + lo, hi - lo, st.data(),
                                 ln.data(), count, m->dev[r], &q);
  if (rc == 0)
  {
    rc = vsa_queries_set_offset(q, first);
  }
  if (rc != 0)
  {
    takeerror(o, rc);
    vsa_queries_free(q);
    return;
  }
  switch (job.mode)
  {
    case VSA_MULTI_COMPLETE:
      rc = vsa_findcompletematches(m->ix[r], q, &res);
      break;
    case VSA_MULTI_MEM:
      rc = vsa_findquerymatches(m->ix[r], q, 0, 0, job.searchlength, &res);
      break;
    case VSA_MULTI_MUMCAND:
      rc = vsa_findquerymatches(m->ix[r], q, 1, 1, job.searchlength, &res);
      break;
    case VSA_MULTI_APPROX:
      rc = vsa_findapproxcompletematches(m->ix[r], q, job.doedist,
                                         job.distvalue, job.percent, &res);
      break;
    default:
      rc = job.lengthbits != 0
               ? vsa_findmumcandidates_packed(m->ix[r], q, job.searchlength,
                                              job.lengthbits, &res)
               : vsa_findmumcandidates(m->ix[r], q, job.searchlength, 0, &res);
      break;
  }
  if (rc != 0)
  {
    takeerror(o, rc); // -complete: the matches found so far still count
  }
  if (res != nullptr)
  {
    (void) vsa_result_getstats(res, &o.stats);
    if (job.mode != VSA_MULTI_MUM)
    {
      if (takematches(res, o) != 0 && o.rc == 0)
      {
        takeerror(o, -100);
      }
    } else if (rc == 0)
    {
      // candidates grouped by the replica that filters their dbstart range
      o.ncand = vsa_result_count(res);
      o.counts.assign(world, 0);
      o.maxright.assign(world, 0);
      const uint64_t rowbytes = job.lengthbits != 0 ? 16 : sizeof(vsa_match);
      if (vsa_device_malloc(std::max<uint64_t>(o.ncand, 1) * rowbytes,
                            m->dev[r], &o.sendbuf) != 0 ||
          vsa_result_partition(res, world, job.totallength, o.sendbuf,
                               o.counts.data(), o.maxright.data()) != 0)
      {
        takeerror(o, -100);
      }
    }
    vsa_result_free(res);
  }
  vsa_queries_free(q);
}

// phase 2 of replica r (-mum): pull range r from every replica, filter it
void filterrange(const Job &job, uint32_t r, std::vector<RankOut> &all,
                 RankOut &o)
{
  vsa_multi *m = job.m;
  const uint32_t world = (uint32_t) m->dev.size();
  const uint64_t rowbytes = job.lengthbits != 0 ? 16 : sizeof(vsa_match);
  uint64_t rows = 0, carry = 0;
  for (uint32_t s = 0; s < world; s++)
  {
    rows += all[s].counts[r];
    for (uint32_t p = 0; p < r; p++)
    {
      carry = std::max(carry, all[s].maxright[p]);
    }
  }
  void *recv = nullptr;
  if (hipSetDevice(m->dev[r]) != hipSuccess ||
      vsa_device_malloc(std::max<uint64_t>(rows, 1) * rowbytes, m->dev[r],
                        &recv) != 0)
  {
    takeerror(o, -100);
    return;
  }
  uint64_t at = 0;
  bool ok = true;
  for (uint32_t s = 0; s < world && ok; s++)
  {
    uint64_t before = 0; // rows of replica s for the ranges below r
    for (uint32_t p = 0; p < r; p++)
    {
      before += all[s].counts[p];
    }
    const uint64_t c = all[s].counts[r];
    if (c > 0)
    {
      ok = hipMemcpyPeerAsync((char *) recv + at * rowbytes, m->dev[r],
                              (const char *) all[s].sendbuf +
                                  before * rowbytes,
                              m->dev[s], c * rowbytes, nullptr) == hipSuccess;
      at += c;
    }
  }
  ok = ok && hipDeviceSynchronize() == hipSuccess;
  vsa_result *res = nullptr;
  int rc = ok ? 0 : -100;
  if (rc == 0)
  {
    rc = job.lengthbits != 0
             ? vsa_mumuniqueinquery_range_packed(recv, rows, job.lengthbits,
                                                 job.totallength, m->dev[r],
                                                 carry, &res)
             : vsa_mumuniqueinquery_range(recv, rows, m->dev[r], carry, &res);
  }
  if (rc != 0)
  {
    if (!ok)
    {
      o.rc = -100;
      o.message = "vsa_multi: peer copy of MUM candidates failed";
    } else
    {
      takeerror(o, rc);
    }
  } else
  {
    vsa_stats fs;
    (void) vsa_result_getstats(res, &fs);
    o.stats.count = fs.count;
    o.stats.sumlength = fs.sumlength;
    if (takematches(res, o) != 0)
    {
      takeerror(o, -100);
    }
  }
  vsa_result_free(res);
  (void) vsa_device_free(recv, m->dev[r]);
}

template <typename F> void onallreplicas(uint32_t world, F f)
{
  std::vector<std::thread> threads;
  for (uint32_t r = 1; r < world; r++)
  {
    threads.emplace_back(f, r);
  }
  f(0u); // the calling thread drives replica 0
  for (std::thread &t : threads)
  {
    t.join();
  }
}

} // namespace

extern "C" int vsa_multi_from_tables(const vsa_tables *tables,
                                     const int *devices, uint32_t ndevices,
                                     vsa_multi **multi)
{
  if (tables == nullptr || devices == nullptr || ndevices == 0 ||
      multi == nullptr)
  {
    seterror("vsa_multi_from_tables: bad argument");
    return -1;
  }
  *multi = nullptr;
  vsa_multi *m = new vsa_multi;
  m->dev.assign(devices, devices + ndevices);
  m->ix.assign(ndevices, nullptr);
  std::vector<int> rcs(ndevices, 0);
  std::vector<std::string> msgs(ndevices);
  onallreplicas(ndevices, [&](uint32_t r) {
    rcs[r] = vsa_index_from_tables(tables, m->dev[r], &m->ix[r]);
    if (rcs[r] != 0)
    {
      msgs[r] = vsa_messagespace();
    }
  });
  for (uint32_t r = 0; r < ndevices; r++)
  {
    if (rcs[r] != 0)
    {
      seterror(msgs[r]);
      const int rc = rcs[r];
      vsa_multi_close(m);
      return rc;
    }
  }
  initcomms(m);
  *multi = m;
  return 0;
}

extern "C" int vsa_multi_replicate(vsa_index *first, const int *devices,
                                   uint32_t ndevices, vsa_multi **multi)
{
  vsa_index_info info;
  if (first == nullptr || devices == nullptr || ndevices == 0 ||
      multi == nullptr || vsa_index_getinfo(first, &info) != 0 ||
      info.device != devices[0])
  {
    seterror("vsa_multi_replicate: bad argument (devices[0] must be the "
             "device of the index)");
    return -1;
  }
  *multi = nullptr;
  vsa_multi *m = new vsa_multi;
  m->dev.assign(devices, devices + ndevices);
  m->ix.assign(ndevices, nullptr);
  m->ix[0] = first;
  std::vector<int> rcs(ndevices, 0);
  std::vector<std::string> msgs(ndevices);
  // every other replica pulls its copy from replica 0 at the same time: the
  // GPUs of a node are connected pairwise, each copy has a link of its own
  onallreplicas(ndevices, [&](uint32_t r) {
    if (r > 0)
    {
      rcs[r] = vsa_index_clone(first, m->dev[r], &m->ix[r]);
      if (rcs[r] != 0)
      {
        msgs[r] = vsa_messagespace();
      }
    }
  });
  for (uint32_t r = 0; r < ndevices; r++)
  {
    if (rcs[r] != 0)
    {
      seterror(msgs[r]);
      const int rc = rcs[r];
      m->ix[0] = nullptr; // the caller keeps its index on failure
      vsa_multi_close(m);
      return rc;
    }
  }
  initcomms(m);
  *multi = m;
  return 0;
}

extern "C" uint32_t vsa_multi_ndevices(const vsa_multi *m)
{
  return m == nullptr ? 0 : (uint32_t) m->dev.size();
}

extern "C" vsa_index *vsa_multi_index(vsa_multi *m, uint32_t replica)
{
  return (m == nullptr || replica >= m->ix.size()) ? nullptr : m->ix[replica];
}

extern "C" int vsa_multi_uses_rccl(const vsa_multi *m)
{
  return m == nullptr ? 0 : m->usedrccl;
}

extern "C" void vsa_multi_close(vsa_multi *m)
{
  if (m == nullptr)
  {
    return;
  }
  for (size_t r = 0; r < m->comms.size(); r++)
  {
    (void) hipSetDevice(m->dev[r]);
    (void) ncclCommDestroy(m->comms[r]);
    (void) hipStreamDestroy(m->streams[r]);
    (void) hipFree(m->counters[r]);
  }
  for (vsa_index *ix : m->ix)
  {
    vsa_index_close(ix);
  }
  delete m;
}

extern "C" void vsa_multi_free_matches(vsa_match *matches)
{
  free(matches);
}

namespace
{

int multi_findmatches(vsa_multi *m, int mode, uint64_t searchlength,
                      int doedist, uint64_t distvalue, int percent,
                      const uint8_t *symbols, uint64_t nsymbols,
                      const uint64_t *start, const uint64_t *length,
                      uint64_t nq, vsa_match **matches, uint64_t *count,
                      vsa_stats *total)
{
  if (m == nullptr || matches == nullptr || count == nullptr || mode < 0 ||
      mode > VSA_MULTI_APPROX || (nq > 0 && (start == nullptr ||
                                             length == nullptr)) ||
      (nsymbols > 0 && symbols == nullptr))
  {
    seterror("vsa_multi_findmatches: bad argument");
    return -1;
  }
  *matches = nullptr;
  *count = 0;
  // every query inside the caller's buffer (any order, overlaps allowed, as
  // for vsa_queries_from_host): the blocks are uploaded from start/length
  // alone, and nothing behind nsymbols is the library's to read
  for (uint64_t i = 0; i < nq; i++)
  {
    if (start[i] > nsymbols || length[i] > nsymbols - start[i])
    {
      char msg[160];
      snprintf(msg, sizeof msg,
               "vsa_multi_findmatches: query %llu (start %llu, length %llu) "
               "lies outside the %llu symbols given",
               (unsigned long long) i, (unsigned long long) start[i],
               (unsigned long long) length[i], (unsigned long long) nsymbols);
      seterror(msg);
      return -2;
    }
  }
  const uint32_t world = (uint32_t) m->dev.size();
  vsa_index_info info;
  if (vsa_index_getinfo(m->ix[0], &info) != 0)
  {
    return -1;
  }
  Job job;
  job.m = m;
  job.mode = mode;
  job.searchlength = searchlength;
  job.symbols = symbols;
  job.start = start;
  job.length = length;
  job.nq = nq;
  job.totallength = info.totallength;
  job.lengthbits = 0;
  job.doedist = doedist;
  job.distvalue = distvalue;
  job.percent = percent;
  if (mode == VSA_MULTI_MUM)
  {
    // the pairs of all replicas are laid out alike: the length bits of the
    // longest query of the job; queries too long for pairs travel as records
    uint64_t longest = 1;
    for (uint64_t i = 0; i < nq; i++)
    {
      longest = std::max(longest, length[i]);
    }
    uint32_t bits = 0;
    while ((longest >> bits) != 0)
    {
      bits++;
    }
    job.lengthbits = (longest < 0xFFFFu && (nq >> 47) == 0) ? bits : 0;
  }
  std::vector<RankOut> out(world), filtered(world);
  onallreplicas(world, [&](uint32_t r) { searchblock(job, r, out[r]); });
  int rc = 0;
  uint32_t failed = world;
  for (uint32_t r = 0; r < world; r++)
  {
    // a configuration the engine does not take: of the whole job, nothing is
    // delivered (the caller hands all of it to the reference's own function)
    if (out[r].rc == VSA_NOT_COVERED)
    {
      seterror(out[r].message);
      return VSA_NOT_COVERED;
    }
  }
  for (uint32_t r = 0; r < world; r++)
  {
    if (out[r].rc != 0)
    {
      rc = out[r].rc;
      failed = r;
      seterror(out[r].message);
      break;
    }
  }
  std::vector<RankOut> *lists = &out;
  if (mode == VSA_MULTI_MUM && rc == 0)
  {
    onallreplicas(world, [&](uint32_t r) {
      filtered[r].stats = out[r].stats;
      filterrange(job, r, out, filtered[r]);
    });
    for (uint32_t r = 0; r < world; r++)
    {
      // candidates are a job-wide figure of phase 1, MUMs of phase 2
      filtered[r].stats.candidates = out[r].ncand;
      if (filtered[r].rc != 0 && rc == 0)
      {
        rc = filtered[r].rc;
        failed = r;
        seterror(filtered[r].message);
      }
    }
    lists = &filtered;
  }
  if (mode == VSA_MULTI_MUM)
  {
    for (uint32_t r = 0; r < world; r++)
    {
      if (out[r].sendbuf != nullptr)
      {
        (void) vsa_device_free(out[r].sendbuf, m->dev[r]);
      }
    }
  }
  // the reference stops at the first error: lists of the replicas before the
  // failing one, then what that one had delivered
  uint64_t totalcount = 0;
  const uint32_t upto = (rc != 0 && mode != VSA_MULTI_MUM)
                            ? failed + 1
                            : (rc != 0 ? 0 : world);
  for (uint32_t r = 0; r < upto; r++)
  {
    totalcount += (*lists)[r].matches.size();
  }
  vsa_match *all =
      (vsa_match *) malloc(std::max<uint64_t>(totalcount, 1) * sizeof(vsa_match));
  if (all == nullptr)
  {
    seterror("vsa_multi_findmatches: out of host memory");
    return -100;
  }
  uint64_t at = 0;
  for (uint32_t r = 0; r < upto; r++)
  {
    const std::vector<vsa_match> &v = (*lists)[r].matches;
    if (!v.empty())
    {
      memcpy(all + at, v.data(), v.size() * sizeof(vsa_match));
      at += v.size();
    }
  }
  *matches = all;
  *count = totalcount;
  if (rc == 0)
  {
    rc = reducecounters(m, *lists, total);
  }
  return rc;
}

} // namespace

extern "C" int vsa_multi_findmatches(vsa_multi *m, int mode,
                                     uint64_t searchlength,
                                     const uint8_t *symbols, uint64_t nsymbols,
                                     const uint64_t *start,
                                     const uint64_t *length, uint64_t nq,
                                     vsa_match **matches, uint64_t *count,
                                     vsa_stats *total)
{
  if (mode > VSA_MULTI_MUM)
  {
    seterror("vsa_multi_findmatches: bad argument");
    return -1;
  }
  return multi_findmatches(m, mode, searchlength, 0, 0, 0, symbols, nsymbols,
                           start, length, nq, matches, count, total);
}

extern "C" int vsa_multi_findapproxcompletematches(
    vsa_multi *m, int doedist, uint64_t distvalue, int percent,
    const uint8_t *symbols, uint64_t nsymbols, const uint64_t *start,
    const uint64_t *length, uint64_t nq, vsa_match **matches, uint64_t *count,
    vsa_stats *total)
{
  return multi_findmatches(m, VSA_MULTI_APPROX, 0, doedist, distvalue,
                           percent, symbols, nsymbols, start, length, nq,
                           matches, count, total);
}

extern "C" int vsa_multi_findapproxcompletematches_cb(
    vsa_multi *m, int doedist, uint64_t distvalue, int percent,
    const uint8_t *symbols, uint64_t nsymbols, const uint64_t *start,
    const uint64_t *length, uint64_t nq, vsa_processmatch processmatch,
    void *info)
{
  vsa_match *matches = nullptr;
  uint64_t count = 0;
  if (processmatch == nullptr)
  {
    seterror("vsa_multi_findapproxcompletematches_cb: NULL callback");
    return -1;
  }
  int rc = vsa_multi_findapproxcompletematches(
      m, doedist, distvalue, percent, symbols, nsymbols, start, length, nq,
      &matches, &count, nullptr);
  for (uint64_t i = 0; i < count; i++)
  {
    if (processmatch(info, matches + i) != 0)
    {
      rc = -1; // stopped by the callback, like the single-GPU entries
      break;
    }
  }
  vsa_multi_free_matches(matches);
  return rc;
}

extern "C" int vsa_multi_findmatches_cb(vsa_multi *m, int mode,
                                        uint64_t searchlength,
                                        const uint8_t *symbols,
                                        uint64_t nsymbols,
                                        const uint64_t *start,
                                        const uint64_t *length, uint64_t nq,
                                        vsa_processmatch processmatch,
                                        void *info)
{
  vsa_match *matches = nullptr;
  uint64_t count = 0;
  if (processmatch == nullptr)
  {
    seterror("vsa_multi_findmatches_cb: NULL callback");
    return -1;
  }
  int rc = vsa_multi_findmatches(m, mode, searchlength, symbols, nsymbols,
                                 start, length, nq, &matches, &count, nullptr);
  for (uint64_t i = 0; i < count; i++)
  {
    if (processmatch(info, matches + i) != 0)
    {
      rc = -1; // stopped by the callback, like the single-GPU entries
      break;
    }
  }
  vsa_multi_free_matches(matches);
  return rc;
}
