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
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <mutex>
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
  // -mum: this rank's candidates grouped by receiving rank (device memory;
  // released with the object, whichever way the job ends)
  void *sendbuf = nullptr;
  int sendbufdevice = 0;
  std::vector<uint64_t> counts, maxright;
  uint64_t ncand = 0;
  RankOut() = default;
  RankOut(const RankOut &) = delete;
  RankOut &operator=(const RankOut &) = delete;
  ~RankOut()
  {
    if (sendbuf != nullptr)
    {
      (void) vsa_device_free(sendbuf, sendbufdevice);
    }
  }
};

// The host threads of a replica set: replica r >= 1 has a thread of its own
// for as long as the set lives, the caller's thread drives replica 0.  A job
// is handed to all of them at once and joined (two condition variables; a
// std::thread per replica and call cost more than the exchange they drive).
struct Crew
{
  std::mutex lock;
  std::condition_variable wake, done;
  std::function<void(uint32_t)> job;
  uint64_t generation = 0;
  uint32_t pending = 0;
  bool stop = false;
  std::vector<std::thread> threads;

  void start(uint32_t world)
  {
    for (uint32_t r = 1; r < world; r++)
    {
      threads.emplace_back([this, r] { loop(r); });
    }
  }
  void loop(uint32_t r)
  {
    uint64_t seen = 0;
    for (;;)
    {
      std::function<void(uint32_t)> f;
      {
        std::unique_lock<std::mutex> g(lock);
        wake.wait(g, [&] { return stop || generation != seen; });
        if (stop)
        {
          return;
        }
        seen = generation;
        f = job;
      }
      f(r);
      {
        std::lock_guard<std::mutex> g(lock);
        pending--;
      }
      done.notify_all();
    }
  }
  template <typename F> void run(F f)
  {
    if (threads.empty())
    {
      f(0u);
      return;
    }
    {
      std::lock_guard<std::mutex> g(lock);
      job = f;
      generation++;
      pending = (uint32_t) threads.size();
    }
    wake.notify_all();
    f(0u);
    std::unique_lock<std::mutex> g(lock);
    done.wait(g, [&] { return pending == 0; });
  }
  void finish()
  {
    {
      std::lock_guard<std::mutex> g(lock);
      stop = true;
    }
    wake.notify_all();
    for (std::thread &t : threads)
    {
      t.join();
    }
    threads.clear();
  }
};

// what a replica keeps from call to call for the device-resident -mum form:
// its candidate rows grouped by receiving replica, the rows it receives, the
// split sizes / right ends (2 * world words) on the device and page-locked
struct Exchange
{
  void *rows = nullptr, *recv = nullptr;
  uint64_t rowcap = 0, recvcap = 0;
  uint64_t *devmeta = nullptr, *hostmeta = nullptr;
  // one stream per replica the rows are pulled from: every pair of GPUs has
  // an xGMI link of its own, and copies queued on one stream would use them
  // one after the other
  std::vector<hipStream_t> from;
};

} // namespace

struct vsa_multi
{
  std::vector<vsa_index *> ix;
  std::vector<int> dev;
  std::vector<ncclComm_t> comms; // empty: counters are summed on the host
  std::vector<hipStream_t> streams;
  std::vector<unsigned long long *> counters; // device, 4 words per replica
  std::vector<unsigned long long *> hcounters; // page-locked, 8 per replica
  int usedrccl = 0;
  Crew crew;
  std::vector<Exchange> xch;
};

namespace
{

bool distinct(const std::vector<int> &d)
{
  std::vector<int> s(d);
  std::sort(s.begin(), s.end());
  return std::adjacent_find(s.begin(), s.end()) == s.end();
}

// Peer copies (the index to its replicas, candidate rows between replicas)
// go straight over xGMI where the devices can map each other's memory; where
// they cannot, or the switch says no, the runtime stages them.  Failures are
// not errors here.
void allowpeers(const std::vector<int> &dev)
{
  const char *off = getenv("VSA_MULTI_PEER_ACCESS");
  if (off != nullptr && strcmp(off, "0") == 0)
  {
    return;
  }
  for (size_t r = 0; r < dev.size(); r++)
  {
    if (hipSetDevice(dev[r]) != hipSuccess)
    {
      (void) hipGetLastError();
      continue;
    }
    for (size_t s = 0; s < dev.size(); s++)
    {
      int can = 0;
      if (dev[s] == dev[r] ||
          hipDeviceCanAccessPeer(&can, dev[r], dev[s]) != hipSuccess || !can)
      {
        (void) hipGetLastError();
        continue;
      }
      (void) hipDeviceEnablePeerAccess(dev[s], 0); // (again: already enabled)
      (void) hipGetLastError();
    }
  }
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
  m->hcounters.assign(m->dev.size(), nullptr);
  for (size_t r = 0; r < m->dev.size(); r++)
  {
    (void) hipSetDevice(m->dev[r]);
    (void) hipStreamCreateWithFlags(&m->streams[r], hipStreamNonBlocking);
    (void) hipMalloc((void **) &m->counters[r], 4 * sizeof(unsigned long long));
    (void) hipHostMalloc((void **) &m->hcounters[r],
                         8 * sizeof(unsigned long long), hipHostMallocDefault);
  }
}

// sums 4 counters per replica over all replicas: RCCL when there are
// communicators, the host otherwise.  Every replica ends with the totals.
int reducecounters(vsa_multi *m, const std::vector<vsa_stats> &st,
                   vsa_stats *total)
{
  const size_t world = m->dev.size();
  unsigned long long sum[4] = {0, 0, 0, 0};
  m->usedrccl = 0;
  if (!m->comms.empty())
  {
    // in: page-locked words -> device (async); the all-reduce on the same
    // stream; out: device -> page-locked words of replica 0; ONE wait per
    // replica for all of it
    bool ok = true;
    for (size_t r = 0; r < world && ok; r++)
    {
      unsigned long long *h = m->hcounters[r];
      ok = h != nullptr;
      if (ok)
      {
        h[0] = st[r].count;
        h[1] = st[r].sumlength;
        h[2] = st[r].searches;
        h[3] = st[r].candidates;
        ok = hipSetDevice(m->dev[r]) == hipSuccess &&
             hipMemcpyAsync(m->counters[r], h, 4 * sizeof *h,
                            hipMemcpyHostToDevice, m->streams[r]) ==
                 hipSuccess;
      }
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
    if (ok)
    {
      ok = hipSetDevice(m->dev[0]) == hipSuccess &&
           hipMemcpyAsync(m->hcounters[0] + 4, m->counters[0],
                          4 * sizeof(unsigned long long),
                          hipMemcpyDeviceToHost, m->streams[0]) == hipSuccess;
    }
    for (size_t r = 0; r < world && ok; r++)
    {
      ok = hipSetDevice(m->dev[r]) == hipSuccess &&
           hipStreamSynchronize(m->streams[r]) == hipSuccess;
    }
    if (ok)
    {
      memcpy(sum, m->hcounters[0] + 4, sizeof sum);
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
      sum[0] += st[r].count;
      sum[1] += st[r].sumlength;
      sum[2] += st[r].searches;
      sum[3] += st[r].candidates;
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
          std::max(total->search_kernel_ms, st[r].search_kernel_ms);
      total->total_device_ms =
          std::max(total->total_device_ms, st[r].total_device_ms);
      total->kernel_searches += st[r].kernel_searches;
      total->first_kernel_ms =
          std::max(total->first_kernel_ms, st[r].first_kernel_ms);
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
      o.sendbufdevice = m->dev[r];
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

// before the set has its crew (construction): a thread per replica
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

// the set is complete: communicators, host threads, exchange state
void commission(vsa_multi *m)
{
  initcomms(m);
  m->xch.resize(m->dev.size());
  m->crew.start((uint32_t) m->dev.size());
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
  // The tables go to the first device once; the other replicas are copies of
  // it, device to device (vsa_multi_replicate).  An upload per device would
  // let every device choose the depth of its derived tables from the memory
  // it happens to have free (index_derive.hip): replicas of one set must not
  // differ in the kernels they run.
  vsa_index *first = nullptr;
  int rc = vsa_index_from_tables(tables, devices[0], &first);
  if (rc != 0)
  {
    return rc; // (the message is in this thread's buffer already)
  }
  rc = vsa_multi_replicate(first, devices, ndevices, multi);
  if (rc != 0)
  {
    vsa_index_close(first);
  }
  return rc;
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
  allowpeers(m->dev);
  (void) hipSetDevice(m->dev[0]);
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
  commission(m);
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
  m->crew.finish();
  for (size_t r = 0; r < m->xch.size(); r++)
  {
    Exchange &x = m->xch[r];
    (void) hipSetDevice(m->dev[r]);
    if (x.rows != nullptr)
    {
      (void) vsa_device_free(x.rows, m->dev[r]);
    }
    if (x.recv != nullptr)
    {
      (void) vsa_device_free(x.recv, m->dev[r]);
    }
    if (x.devmeta != nullptr)
    {
      (void) vsa_device_free(x.devmeta, m->dev[r]);
    }
    if (x.hostmeta != nullptr)
    {
      (void) hipHostFree(x.hostmeta);
    }
    for (hipStream_t st : x.from)
    {
      if (st != nullptr)
      {
        (void) hipStreamDestroy(st);
      }
    }
  }
  for (size_t r = 0; r < m->comms.size(); r++)
  {
    (void) hipSetDevice(m->dev[r]);
    (void) ncclCommDestroy(m->comms[r]);
    (void) hipStreamDestroy(m->streams[r]);
    (void) hipFree(m->counters[r]);
    (void) hipHostFree(m->hcounters[r]);
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
  m->crew.run([&](uint32_t r) { searchblock(job, r, out[r]); });
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
    m->crew.run([&](uint32_t r) {
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
  // (the send buffers of a -mum job go with `out`, whichever way this ends)
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
    std::vector<vsa_stats> st(world);
    for (uint32_t r = 0; r < world; r++)
    {
      st[r] = (*lists)[r].stats;
    }
    rc = reducecounters(m, st, total);
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

// ---- the device-resident form ------------------------------------------------
// Every replica's block of queries lies in its HBM already and the lists stay
// there: nothing of a job crosses PCIe but 2 * world numbers per replica
// (-mum: the split sizes of the exchange) and the four counters.

namespace
{

struct DeviceJob
{
  vsa_multi *m;
  int mode;
  uint64_t searchlength, totallength;
  uint32_t lengthbits;
  vsa_queries *const *blocks;
  vsa_result **results;
};

struct DeviceOut
{
  int rc = 0;
  std::string message;
  vsa_stats stats;
  uint64_t ncand = 0;
};

int fail(DeviceOut &o, int rc, const char *what)
{
  o.rc = rc;
  o.message = what != nullptr ? what : vsa_messagespace();
  return rc;
}

// room for `rows` rows of 16 bytes in *buf (kept from call to call)
int growrows(void **buf, uint64_t *cap, uint64_t rows, int device)
{
  if (rows <= *cap && *buf != nullptr)
  {
    return 0;
  }
  if (*buf != nullptr)
  {
    (void) vsa_device_free(*buf, device);
    *buf = nullptr;
    *cap = 0;
  }
  const uint64_t want = rows + rows / 8 + 4096;
  if (vsa_device_malloc(want * 16, device, buf) != 0)
  {
    return -100;
  }
  *cap = want;
  return 0;
}

// phase 1 of replica r: the search on its block; -mum: the candidates grouped
// by the replica that filters their range of the index (the rows for r itself
// behind all others: they do not travel), split sizes and right ends fetched
void devicesearch(const DeviceJob &job, uint32_t r, DeviceOut &o)
{
  vsa_multi *m = job.m;
  const uint32_t world = (uint32_t) m->dev.size();
  memset(&o.stats, 0, sizeof o.stats);
  job.results[r] = nullptr;
  if (hipSetDevice(m->dev[r]) != hipSuccess)
  {
    (void) fail(o, -100, "hipSetDevice failed");
    return;
  }
  vsa_result *res = nullptr;
  int rc = 0;
  switch (job.mode)
  {
    case VSA_MULTI_COMPLETE:
      rc = vsa_findcompletematches(m->ix[r], job.blocks[r], &res);
      break;
    case VSA_MULTI_MEM:
      rc = vsa_findquerymatches(m->ix[r], job.blocks[r], 0, 0,
                                job.searchlength, &res);
      break;
    case VSA_MULTI_MUMCAND:
      rc = vsa_findquerymatches(m->ix[r], job.blocks[r], 1, 1,
                                job.searchlength, &res);
      break;
    default:
    {
      Exchange &x = m->xch[r];
      if (x.devmeta == nullptr &&
          (vsa_device_malloc(2 * (uint64_t) world * 8, m->dev[r],
                             (void **) &x.devmeta) != 0 ||
           hipHostMalloc((void **) &x.hostmeta, 2 * (size_t) world * 8,
                         hipHostMallocDefault) != hipSuccess))
      {
        (void) fail(o, -100, "vsa_multi: no memory for the exchange state");
        return;
      }
      if (growrows(&x.rows, &x.rowcap, 1, m->dev[r]) != 0)
      {
        (void) fail(o, -100, nullptr);
        return;
      }
      rc = vsa_findmumcandidates_grouped(m->ix[r], job.blocks[r],
                                         job.searchlength, job.lengthbits,
                                         world, (int) r, x.rows, x.rowcap,
                                         x.devmeta, &res);
      if (rc == 1)
      {
        // more candidates than the row buffer holds: make room, group
        rc = growrows(&x.rows, &x.rowcap, vsa_result_count(res), m->dev[r]);
        if (rc == 0)
        {
          rc = vsa_result_partition_device(res, world, (int) r,
                                           job.totallength, x.rows, x.devmeta);
        }
      }
      if (rc == 0)
      {
        // (the grouping is queued on the device's default stream: the copy
        // behind it on the same stream sees its numbers, and the rows are in
        // place when it has arrived)
        if (hipMemcpyAsync(x.hostmeta, x.devmeta, 2 * (size_t) world * 8,
                           hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
            hipStreamSynchronize(nullptr) != hipSuccess)
        {
          rc = fail(o, -100, "vsa_multi: split sizes did not arrive");
        }
      }
      if (res != nullptr)
      {
        (void) vsa_result_getstats(res, &o.stats);
        o.ncand = vsa_result_count(res);
        vsa_result_free(res); // the rows hold what the filter needs
        res = nullptr;
      }
      if (rc != 0 && o.rc == 0)
      {
        (void) fail(o, rc, nullptr);
      }
      return;
    }
  }
  if (rc != 0)
  {
    (void) fail(o, rc, nullptr); // -complete: the matches so far still count
  }
  if (res != nullptr)
  {
    (void) vsa_result_getstats(res, &o.stats);
  }
  job.results[r] = res;
}

// phase 2 of replica r (-mum): range r of every other replica comes over by
// peer copies (one xGMI hop each); the filter reads r's own rows where they
// lie and the received ones as one list
void devicefilter(const DeviceJob &job, uint32_t r,
                  const std::vector<DeviceOut> &found, DeviceOut &o)
{
  vsa_multi *m = job.m;
  const uint32_t world = (uint32_t) m->dev.size();
  Exchange &x = m->xch[r];
  if (hipSetDevice(m->dev[r]) != hipSuccess)
  {
    (void) fail(o, -100, "hipSetDevice failed");
    return;
  }
  uint64_t nrecv = 0, carry = 0;
  for (uint32_t s = 0; s < world; s++)
  {
    const uint64_t *meta = m->xch[s].hostmeta;
    if (s != r)
    {
      nrecv += meta[r];
    }
    for (uint32_t p = 0; p < r; p++)
    {
      carry = std::max(carry, meta[world + p]);
    }
  }
  if (growrows(&x.recv, &x.recvcap, nrecv, m->dev[r]) != 0)
  {
    (void) fail(o, -100, nullptr);
    return;
  }
  uint64_t at = 0;
  bool ok = true;
  if (x.from.size() != world)
  {
    x.from.assign(world, nullptr);
  }
  for (uint32_t s = 0; s < world && ok; s++)
  {
    if (s == r)
    {
      continue;
    }
    const uint64_t *meta = m->xch[s].hostmeta;
    uint64_t before = 0; // rows of replica s in front of its part r
    for (uint32_t p = 0; p < r; p++)
    {
      before += (p == s) ? 0 : meta[p];
    }
    const uint64_t c = meta[r];
    if (c > 0)
    {
      if (x.from[s] == nullptr)
      {
        ok = hipStreamCreateWithFlags(&x.from[s], hipStreamNonBlocking) ==
             hipSuccess;
      }
      ok = ok &&
           hipMemcpyPeerAsync((char *) x.recv + at * 16, m->dev[r],
                              (const char *) m->xch[s].rows + before * 16,
                              m->dev[s], c * 16, x.from[s]) == hipSuccess;
      at += c;
    }
  }
  for (uint32_t s = 0; s < world; s++)
  {
    if (x.from[s] != nullptr)
    {
      ok = hipStreamSynchronize(x.from[s]) == hipSuccess && ok;
    }
  }
  if (!ok)
  {
    (void) fail(o, -100, "vsa_multi: peer copy of MUM candidates failed");
    return;
  }
  const uint64_t own = x.hostmeta[r];
  vsa_result *res = nullptr;
  const int rc = vsa_mumuniqueinquery_range_packed2(
      (const char *) x.rows + (found[r].ncand - own) * 16, own, x.recv, nrecv,
      job.lengthbits, job.totallength, m->dev[r], carry, &res);
  if (rc != 0)
  {
    (void) fail(o, rc, nullptr);
    return;
  }
  vsa_stats fs;
  (void) vsa_result_getstats(res, &fs);
  o.stats = found[r].stats;
  o.stats.count = fs.count;
  o.stats.sumlength = fs.sumlength;
  o.stats.candidates = found[r].ncand;
  o.stats.total_device_ms += fs.total_device_ms;
  job.results[r] = res;
}

} // namespace

extern "C" int vsa_multi_findmatches_device(vsa_multi *m, int mode,
                                            uint64_t searchlength,
                                            vsa_queries *const *blocks,
                                            vsa_result **results,
                                            vsa_stats *total)
{
  if (m == nullptr || blocks == nullptr || results == nullptr || mode < 0 ||
      mode > VSA_MULTI_MUM)
  {
    seterror("vsa_multi_findmatches_device: bad argument");
    return -1;
  }
  const uint32_t world = (uint32_t) m->dev.size();
  vsa_index_info info;
  if (vsa_index_getinfo(m->ix[0], &info) != 0)
  {
    return -1;
  }
  uint64_t longest = 1, lastquery = 0;
  for (uint32_t r = 0; r < world; r++)
  {
    vsa_queries_info qi;
    results[r] = nullptr;
    if (blocks[r] == nullptr || vsa_queries_getinfo(blocks[r], &qi) != 0 ||
        qi.device != m->dev[r])
    {
      char msg[160];
      snprintf(msg, sizeof msg,
               "vsa_multi_findmatches_device: block %u is not a batch of "
               "queries on device %d", r, m->dev[r]);
      seterror(msg);
      return -1;
    }
    longest = std::max(longest, qi.maxlength);
    lastquery = std::max(lastquery, qi.offset + qi.numofqueries);
  }
  DeviceJob job;
  job.m = m;
  job.mode = mode;
  job.searchlength = searchlength;
  job.totallength = info.totallength;
  job.blocks = blocks;
  job.results = results;
  job.lengthbits = 0;
  if (mode == VSA_MULTI_MUM)
  {
    // the pairs of all replicas are laid out alike: the length bits of the
    // longest query of the job
    uint32_t bits = 0;
    while ((longest >> bits) != 0)
    {
      bits++;
    }
    if (longest >= 0xFFFFu || (lastquery >> 47) != 0)
    {
      seterror("vsa_multi_findmatches_device: -mum takes queries of fewer "
               "than 65 535 symbols and query numbers below 2^47 (the pair "
               "form of the candidates); vsa_multi_findmatches has no such "
               "limit");
      return -2;
    }
    job.lengthbits = bits;
  }
  std::vector<DeviceOut> found(world), filtered(world);
  m->crew.run([&](uint32_t r) { devicesearch(job, r, found[r]); });
  int rc = 0;
  uint32_t failed = world;
  for (uint32_t r = 0; r < world && rc == 0; r++)
  {
    if (found[r].rc != 0)
    {
      rc = found[r].rc;
      failed = r;
      seterror(found[r].message);
    }
  }
  std::vector<DeviceOut> *outs = &found;
  if (mode == VSA_MULTI_MUM && rc == 0)
  {
    m->crew.run(
        [&](uint32_t r) { devicefilter(job, r, found, filtered[r]); });
    for (uint32_t r = 0; r < world && rc == 0; r++)
    {
      if (filtered[r].rc != 0)
      {
        rc = filtered[r].rc;
        failed = r;
        seterror(filtered[r].message);
      }
    }
    outs = &filtered;
  }
  if (rc != 0)
  {
    // the reference stops at the first error: the lists of the replicas in
    // front of the failing one and what that one had found stay (-complete,
    // -l, -mum cand); a -mum job that failed has no list
    for (uint32_t r = 0; r < world; r++)
    {
      if (results[r] != nullptr && (mode == VSA_MULTI_MUM || r > failed))
      {
        vsa_result_free(results[r]);
        results[r] = nullptr;
      }
    }
    return rc;
  }
  std::vector<vsa_stats> st(world);
  for (uint32_t r = 0; r < world; r++)
  {
    st[r] = (*outs)[r].stats;
  }
  return reducecounters(m, st, total);
}

// ---- host memory to host memory over all replicas ----------------------------
// One packed pipeline (vsa_pipeline_open_packed: page-locked slots, three
// batches in flight, upload / search / download overlapped) per replica; the
// batches of a job are dealt out to the replicas in turn and come back in
// the order they were submitted, which is query order -- the reference's
// order for -complete, -l and -mum cand.  -mum: the candidates of a replica's
// batches stay in its HBM; vsa_multi_pipeline_finish groups them by range,
// moves range r to GPU r (peer copies) and filters there, like the
// device-resident form.

struct vsa_multi_pipeline
{
  vsa_multi *m = nullptr;
  int mode = 0;
  uint32_t qlen = 0;
  uint64_t submitted = 0; // queries of the job so far
  uint64_t batches = 0;   // batches of the job so far: the next one goes to
                          // replica batches % ndevices
  std::vector<vsa_pipeline *> pipes;
  std::vector<uint32_t> order; // replicas of the batches not yet delivered
  size_t delivered = 0;
  // -mum: the lists of the replicas in page-locked memory, kept between jobs
  std::vector<vsa_match *> hostlist;
  std::vector<uint64_t> hostcap;
};

extern "C" void vsa_multi_pipeline_close(vsa_multi_pipeline *p)
{
  if (p == nullptr)
  {
    return;
  }
  for (size_t r = 0; r < p->pipes.size(); r++)
  {
    vsa_pipeline_close(p->pipes[r]);
    if (r < p->hostlist.size() && p->hostlist[r] != nullptr)
    {
      (void) hipHostFree(p->hostlist[r]);
    }
  }
  delete p;
}

extern "C" int vsa_multi_pipeline_open(vsa_multi *m, int mode,
                                       uint64_t searchlength,
                                       uint32_t querylength,
                                       uint64_t maxqueries,
                                       uint64_t maxspecial,
                                       vsa_multi_pipeline **pipeline)
{
  if (m == nullptr || pipeline == nullptr || mode < 0 ||
      mode > VSA_MULTI_MUM)
  {
    seterror("vsa_multi_pipeline_open: bad argument");
    return -1;
  }
  *pipeline = nullptr;
  vsa_multi_pipeline *p = new vsa_multi_pipeline;
  const uint32_t world = (uint32_t) m->dev.size();
  p->m = m;
  p->mode = mode;
  p->qlen = querylength;
  p->pipes.assign(world, nullptr);
  p->hostlist.assign(world, nullptr);
  p->hostcap.assign(world, 0);
  for (uint32_t r = 0; r < world; r++)
  {
    const int rc = vsa_pipeline_open_packed(m->ix[r], mode, searchlength,
                                            querylength, maxqueries,
                                            maxspecial, &p->pipes[r]);
    if (rc != 0)
    {
      vsa_multi_pipeline_close(p);
      return rc;
    }
  }
  *pipeline = p;
  return 0;
}

extern "C" int vsa_multi_pipeline_hostrows(vsa_multi_pipeline *p,
                                           uint64_t **rows, uint8_t **special)
{
  if (p == nullptr)
  {
    seterror("vsa_multi_pipeline_hostrows: NULL argument");
    return -1;
  }
  return vsa_pipeline_hostrows(p->pipes[p->batches % p->pipes.size()], rows,
                               special);
}

extern "C" int vsa_multi_pipeline_submit(vsa_multi_pipeline *p,
                                         uint64_t numofqueries,
                                         uint64_t numofspecial)
{
  if (p == nullptr)
  {
    seterror("vsa_multi_pipeline_submit: NULL argument");
    return -1;
  }
  const uint32_t r = (uint32_t) (p->batches % p->pipes.size());
  int rc = vsa_pipeline_set_offset(p->pipes[r], p->submitted);
  if (rc == 0)
  {
    rc = vsa_pipeline_submit_packed(p->pipes[r], numofqueries, numofspecial);
  }
  if (rc == 0)
  {
    p->order.push_back(r);
    p->batches++;
    p->submitted += numofqueries;
  }
  return rc;
}

extern "C" int vsa_multi_pipeline_next(vsa_multi_pipeline *p,
                                       const vsa_match **matches,
                                       uint64_t *count)
{
  if (p == nullptr || matches == nullptr || count == nullptr)
  {
    seterror("vsa_multi_pipeline_next: NULL argument");
    return -1;
  }
  *matches = nullptr;
  *count = 0;
  if (p->delivered == p->order.size())
  {
    return 1;
  }
  const uint32_t r = p->order[p->delivered++];
  if (p->delivered == p->order.size())
  {
    p->order.clear();
    p->delivered = 0;
  }
  return vsa_pipeline_next(p->pipes[r], matches, count);
}

namespace
{

// range r: peer copies of its rows from every other replica, then the filter
// over r's own rows and the received ones (devicefilter, on the rows of a
// pipeline job instead of a search call)
void pipelinefilter(vsa_multi *m, uint32_t r, uint32_t lengthbits,
                    uint64_t totallength, const std::vector<uint64_t> &nrows,
                    DeviceOut &o, vsa_result **result)
{
  vsa_queries *noblocks[1] = {nullptr};
  DeviceJob job;
  job.m = m;
  job.mode = VSA_MULTI_MUM;
  job.searchlength = 0;
  job.totallength = totallength;
  job.lengthbits = lengthbits;
  job.blocks = noblocks;
  std::vector<vsa_result *> results(m->dev.size(), nullptr);
  job.results = results.data();
  std::vector<DeviceOut> found(m->dev.size());
  for (size_t s = 0; s < found.size(); s++)
  {
    memset(&found[s].stats, 0, sizeof found[s].stats);
    found[s].ncand = nrows[s];
  }
  devicefilter(job, r, found, o);
  *result = results[r];
}

} // namespace

extern "C" int vsa_multi_pipeline_finish(vsa_multi_pipeline *p,
                                         const vsa_match **lists,
                                         uint64_t *counts, vsa_stats *total)
{
  if (p == nullptr || lists == nullptr || counts == nullptr ||
      p->mode != VSA_MULTI_MUM)
  {
    seterror("vsa_multi_pipeline_finish: bad argument (a -mum pipeline?)");
    return -1;
  }
  vsa_multi *m = p->m;
  const uint32_t world = (uint32_t) m->dev.size();
  vsa_index_info info;
  if (vsa_index_getinfo(m->ix[0], &info) != 0)
  {
    return -1;
  }
  for (uint32_t r = 0; r < world; r++)
  {
    lists[r] = nullptr;
    counts[r] = 0;
  }
  p->submitted = 0;
  p->batches = 0;
  // phase 1: every replica groups the candidates of its batches by range
  std::vector<DeviceOut> grouped(world), filtered(world);
  std::vector<uint64_t> nrows(world, 0);
  std::vector<uint32_t> bits(world, 0);
  m->crew.run([&](uint32_t r) {
    DeviceOut &o = grouped[r];
    Exchange &x = m->xch[r];
    const void *rows = nullptr;
    memset(&o.stats, 0, sizeof o.stats);
    if (hipSetDevice(m->dev[r]) != hipSuccess)
    {
      (void) fail(o, -100, "hipSetDevice failed");
      return;
    }
    int rc = vsa_pipeline_take_candidates(p->pipes[r], &rows, &nrows[r],
                                          &bits[r]);
    if (rc != 0)
    {
      (void) fail(o, rc, nullptr);
      return;
    }
    if (x.devmeta == nullptr &&
        (vsa_device_malloc(2 * (uint64_t) world * 8, m->dev[r],
                           (void **) &x.devmeta) != 0 ||
         hipHostMalloc((void **) &x.hostmeta, 2 * (size_t) world * 8,
                       hipHostMallocDefault) != hipSuccess))
    {
      (void) fail(o, -100, "vsa_multi: no memory for the exchange state");
      return;
    }
    if (growrows(&x.rows, &x.rowcap, nrows[r], m->dev[r]) != 0)
    {
      (void) fail(o, -100, nullptr);
      return;
    }
    rc = vsa_rows_partition_device(rows, nrows[r], bits[r], world, (int) r,
                                   info.totallength, m->dev[r], x.rows,
                                   x.devmeta);
    if (rc != 0)
    {
      (void) fail(o, rc, nullptr);
      return;
    }
    if (hipMemcpyAsync(x.hostmeta, x.devmeta, 2 * (size_t) world * 8,
                       hipMemcpyDeviceToHost, nullptr) != hipSuccess ||
        hipStreamSynchronize(nullptr) != hipSuccess)
    {
      (void) fail(o, -100, "vsa_multi: split sizes did not arrive");
    }
  });
  for (uint32_t r = 0; r < world; r++)
  {
    if (grouped[r].rc != 0)
    {
      seterror(grouped[r].message);
      return grouped[r].rc;
    }
  }
  // phase 2: exchange, filter, lists to page-locked host memory
  std::vector<vsa_stats> st(world);
  m->crew.run([&](uint32_t r) {
    DeviceOut &o = filtered[r];
    vsa_result *res = nullptr;
    pipelinefilter(m, r, bits[r], info.totallength, nrows, o, &res);
    if (o.rc != 0 || res == nullptr)
    {
      if (o.rc == 0)
      {
        (void) fail(o, -100, "vsa_multi: range filter gave no list");
      }
      return;
    }
    const uint64_t c = vsa_result_count(res);
    if (c > p->hostcap[r])
    {
      if (p->hostlist[r] != nullptr)
      {
        (void) hipHostFree(p->hostlist[r]);
        p->hostlist[r] = nullptr;
      }
      p->hostcap[r] = c + c / 8 + 1024;
      if (hipHostMalloc((void **) &p->hostlist[r],
                        p->hostcap[r] * sizeof(vsa_match),
                        hipHostMallocDefault) != hipSuccess)
      {
        p->hostcap[r] = 0;
        (void) fail(o, -100, "vsa_multi: no page-locked memory for the list");
      }
    }
    if (o.rc == 0 && c > 0 && vsa_result_fetch(res, p->hostlist[r], c) != 0)
    {
      (void) fail(o, -100, nullptr);
    }
    if (o.rc == 0)
    {
      lists[r] = p->hostlist[r];
      counts[r] = c;
    }
    vsa_result_free(res);
  });
  for (uint32_t r = 0; r < world; r++)
  {
    if (filtered[r].rc != 0)
    {
      seterror(filtered[r].message);
      return filtered[r].rc;
    }
    st[r] = filtered[r].stats;
    st[r].candidates = nrows[r];
  }
  return reducecounters(m, st, total);
}
