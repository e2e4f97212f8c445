// End-to-end pipeline: queries in host memory -> matches in host memory.
//
// The engine calls of this library take queries that are resident in HBM and
// leave their matches there; around a 4.5 ms search of 10 M reads a caller who
// uploads 1 GB of reads from pageable memory and fetches 0.36 GB of matches
// with synchronous copies spends 70-110 ms.  The pipeline keeps three batches
// in flight instead: while batch i is searched (the index's stream), batch
// i+1 is on its way up and the matches of batch i-1 on their way down (one
// HIP stream each, i.e. the two DMA engines), all through page-locked host
// buffers the caller fills and reads directly.  One worker thread drives the
// synchronous engine calls; the caller's thread never waits for the GPU except
// in vsa_pipeline_next.
//
// Reads of one length, packed back to back -- the layout of the reference's
// query Multiseq for short reads without the separators (kurtz-basic/
// multiseq.c:129-166) and what a FASTA/FASTQ reader for reads produces.
// Reference semantics per batch: findcompletematches (Vmengine/fcomplete.c:263)
// / findquerymatches (Vmengine/fquery.c:1009) with onlinequerynumoffset = the
// number of queries submitted before; `-mum` keeps the candidates of all
// batches on the device and runs mumuniqueinquery (kurtz/cleanMUMcand.c:55)
// once, in vsa_pipeline_finish, over all of them -- the filter is global.
#include "vsa_internal.hpp"

#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

namespace
{

const int kSlots = 3;

__global__ void k_pipeline_starts(uint64_t *start, uint64_t *length,
                                  uint64_t nq, uint64_t m)
{
  const uint64_t i = vsa_bid() * blockDim.x + threadIdx.x;
  if (i < nq)
  {
    start[i] = i * m;
    length[i] = m;
  }
}

enum SlotState
{
  SLOT_FREE,      // the caller may fill its host buffer
  SLOT_SUBMITTED, // upload enqueued, waiting for the worker
  SLOT_DONE       // matches are (being) copied to the host buffer
};

struct Slot
{
  SlotState state = SLOT_FREE;
  // packed pipelines: rows (maxqueries * W words) and the side list
  // (maxspecial * m bytes) instead of hostq, page-locked
  uint64_t *hostrows = nullptr;
  uint8_t *hostside = nullptr;
  uint64_t nside = 0;
  uint8_t *hostq = nullptr;   // page-locked, maxqueries * m
  vsa_match *hostm = nullptr; // page-locked, grown on demand
  uint64_t hostmcap = 0;
  vsa_queries *q = nullptr;   // device buffers of the slot
  vsa_result *res = nullptr;  // kept until its copy to the host is over
  hipEvent_t uploaded = nullptr, downloaded = nullptr;
  uint64_t nq = 0, first = 0, count = 0;
  int rc = 0;
  std::string message;
};

} // namespace

struct vsa_pipeline
{
  const vsa_index *index = nullptr;
  int mode = 0;
  uint64_t searchlength = 0, maxqueries = 0, submitted = 0;
  uint32_t m = 0, lengthbits = 0;
  bool packed = false; // reads at two bits per symbol (vsa_pipeline_open_packed)
  uint32_t roww = 0;
  uint64_t maxspecial = 0;
  Slot slot[kSlots];
  hipStream_t up = nullptr, down = nullptr;
  std::mutex lock;
  std::condition_variable wake;
  std::deque<int> todo, inorder; // for the worker / for vsa_pipeline_next
  int filling = -1, lastdelivered = -1;
  bool stop = false;
  std::thread worker;
  // -mum: candidate rows (key, value) of all batches, one device buffer
  uint64_t *rows = nullptr;
  uint64_t nrows = 0, rowcap = 0, candidates = 0;
  vsa_match *hostmums = nullptr; // page-locked, kept between jobs
  uint64_t hostmumscap = 0;
  int failed = 0;
  std::string failure;
};

namespace
{

int growhost(Slot &s, uint64_t need)
{
  if (need <= s.hostmcap)
  {
    return 0;
  }
  if (s.hostm != nullptr)
  {
    (void) hipHostFree(s.hostm);
    s.hostm = nullptr;
  }
  const uint64_t cap = need + need / 4 + 1024;
  if (hipHostMalloc((void **) &s.hostm, cap * sizeof(vsa_match),
                    hipHostMallocDefault) != hipSuccess)
  {
    (void) hipGetLastError();
    s.hostmcap = 0;
    return -100;
  }
  s.hostmcap = cap;
  return 0;
}

// candidate rows of one batch behind those of the batches before
int keeprows(vsa_pipeline *p, vsa_result *res)
{
  const uint64_t c = vsa_result_count(res);
  if (p->nrows + c > p->rowcap)
  {
    const uint64_t cap = (p->nrows + c) * 2 + (1u << 20);
    uint64_t *bigger = nullptr;
    if (vsa_hip_malloc((void **) &bigger, cap * 16) != hipSuccess)
    {
      return -100;
    }
    if (p->nrows > 0 &&
        hipMemcpy(bigger, p->rows, p->nrows * 16, hipMemcpyDeviceToDevice) !=
            hipSuccess)
    {
      (void) hipFree(bigger);
      return -100;
    }
    (void) hipFree(p->rows);
    p->rows = bigger;
    p->rowcap = cap;
  }
  uint64_t counts[1] = {0}, top[1] = {0};
  if (c > 0 &&
      vsa_result_partition(res, 1, p->index->n, p->rows + 2 * p->nrows, counts,
                           top) != 0)
  {
    return -100;
  }
  p->nrows += c;
  p->candidates += c;
  return 0;
}

void work(vsa_pipeline *p)
{
  (void) hipSetDevice(p->index->device);
  for (;;)
  {
    int k;
    {
      std::unique_lock<std::mutex> g(p->lock);
      p->wake.wait(g, [&] { return p->stop || !p->todo.empty(); });
      if (p->todo.empty())
      {
        return;
      }
      k = p->todo.front();
      p->todo.pop_front();
    }
    Slot &s = p->slot[k];
    int rc = hipEventSynchronize(s.uploaded) == hipSuccess ? 0 : -100;
    vsa_result *res = nullptr;
    if (rc == 0)
    {
      s.q->nq = s.nq;
      s.q->nsymbols = s.nq * (uint64_t) p->m;
      s.q->seqoffset = s.first;
      s.q->nside = s.nside;
      s.q->bytesvalid = false; // (packed: the slot holds another batch now)
      switch (p->mode)
      {
        case 0: rc = vsa_findcompletematches(p->index, s.q, &res); break;
        case 1:
          rc = vsa_findquerymatches(p->index, s.q, 0, 0, p->searchlength,
                                    &res);
          break;
        case 2:
          rc = vsa_findquerymatches(p->index, s.q, 1, 1, p->searchlength,
                                    &res);
          break;
        default:
          rc = vsa_findmumcandidates_packed(p->index, s.q, p->searchlength,
                                            p->lengthbits, &res);
          break;
      }
    }
    s.count = 0;
    if (rc != 0)
    {
      s.message = vsa_messagespace();
    }
    if (res != nullptr && p->mode == 3)
    {
      if (rc == 0 && keeprows(p, res) != 0)
      {
        rc = -100;
        s.message = "vsa_pipeline: out of device memory for MUM candidates";
      }
      vsa_result_free(res);
      res = nullptr;
    } else if (res != nullptr)
    {
      // the list found so far counts even after an engine error (a query
      // shorter than prefixlength, exactcompl.c:179-185)
      const uint64_t c = vsa_result_count(res);
      const void *dm = vsa_result_device_matches(res);
      if (c > 0 && (growhost(s, c) != 0 || dm == nullptr ||
                    hipMemcpyAsync(s.hostm, dm, c * sizeof(vsa_match),
                                   hipMemcpyDeviceToHost, p->down) !=
                        hipSuccess))
      {
        if (rc == 0)
        {
          rc = -100;
          s.message = "vsa_pipeline: copy of the matches to the host failed";
        }
      } else
      {
        s.count = c;
      }
    }
    (void) hipEventRecord(s.downloaded, p->down);
    {
      std::lock_guard<std::mutex> g(p->lock);
      s.res = res; // freed when the slot is handed back
      s.rc = rc;
      s.state = SLOT_DONE;
      if (rc != 0 && p->mode == 3 && p->failed == 0)
      {
        // -mum: the candidates of this batch are missing from the job, and
        // the filter over the rest would call matches unique that are not
        p->failed = rc;
        p->failure = s.message;
      }
    }
    p->wake.notify_all();
  }
}

void releaseslot(vsa_pipeline *p, int k)
{
  Slot &s = p->slot[k];
  if (s.res != nullptr)
  {
    vsa_result_free(s.res);
    s.res = nullptr;
  }
  s.state = SLOT_FREE;
}

} // namespace

namespace
{
int pipeline_open(const vsa_index *index, int mode, uint64_t searchlength,
                  uint32_t querylength, uint64_t maxqueries, bool packed,
                  uint64_t maxspecial, vsa_pipeline **pipeline);
}

extern "C" int vsa_pipeline_open(const vsa_index *index, int mode,
                                 uint64_t searchlength, uint32_t querylength,
                                 uint64_t maxqueries, vsa_pipeline **pipeline)
{
  return pipeline_open(index, mode, searchlength, querylength, maxqueries,
                       false, 0, pipeline);
}

extern "C" int vsa_pipeline_open_packed(const vsa_index *index, int mode,
                                        uint64_t searchlength,
                                        uint32_t querylength,
                                        uint64_t maxqueries,
                                        uint64_t maxspecial,
                                        vsa_pipeline **pipeline)
{
  return pipeline_open(index, mode, searchlength, querylength, maxqueries,
                       true, maxspecial, pipeline);
}

namespace
{
int pipeline_open(const vsa_index *index, int mode, uint64_t searchlength,
                  uint32_t querylength, uint64_t maxqueries, bool packed,
                  uint64_t maxspecial, vsa_pipeline **pipeline)
{
  if (index == nullptr || pipeline == nullptr || mode < 0 || mode > 3 ||
      querylength == 0 || maxqueries == 0)
  {
    VSA_ERROR("vsa_pipeline_open: bad argument");
    return -1;
  }
  *pipeline = nullptr;
  if (vsa_set_device(index->device) != 0)
  {
    return -100;
  }
  vsa_pipeline *p = new vsa_pipeline;
  p->index = index;
  p->mode = mode;
  p->searchlength = searchlength;
  p->m = querylength;
  p->maxqueries = maxqueries;
  p->packed = packed;
  p->roww = vsa_rowwords(querylength);
  p->maxspecial = maxspecial;
  while ((querylength >> p->lengthbits) != 0)
  {
    p->lengthbits++;
  }
  if (mode == 3 && (querylength >= 0xFFFFu))
  {
    VSA_ERROR("vsa_pipeline_open: reads of %u symbols are too long for the "
              "-mum pipeline", querylength);
    delete p;
    return -2;
  }
  const uint64_t nsym = maxqueries * (uint64_t) querylength;
  bool ok = hipStreamCreateWithFlags(&p->up, hipStreamNonBlocking) ==
                hipSuccess &&
            hipStreamCreateWithFlags(&p->down, hipStreamNonBlocking) ==
                hipSuccess;
  for (int k = 0; k < kSlots && ok; k++)
  {
    Slot &s = p->slot[k];
    vsa_queries *q = new vsa_queries;
    q->device = index->device;
    q->nq = 0;
    q->nsymbols = 0;
    q->seqoffset = 0;
    q->symbols = nullptr;
    q->start = q->length = nullptr;
    q->minlength = q->maxlength = querylength;
    q->uniform = q->dense = true;
    q->hlength.assign(1, querylength); // uniform batches never look at it
    s.q = q;
    if (packed)
    {
      // rows and side list in place of the symbols; the bytes (MEM mode) are
      // made on the device when a batch needs them, into room kept here
      const uint64_t rowbytes = maxqueries * p->roww * 8 + 64,
                     sidebytes = maxspecial * (uint64_t) querylength + 64;
      q->roww = p->roww;
      q->bytescapacity = nsym;
      ok = hipHostMalloc((void **) &s.hostrows, rowbytes,
                         hipHostMallocDefault) == hipSuccess &&
           hipHostMalloc((void **) &s.hostside, sidebytes,
                         hipHostMallocDefault) == hipSuccess &&
           vsa_hip_malloc((void **) &q->rows, rowbytes) == hipSuccess &&
           vsa_hip_malloc((void **) &q->side, sidebytes) == hipSuccess &&
           hipMemsetAsync(q->rows, 0, rowbytes, p->up) == hipSuccess &&
           hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming) ==
               hipSuccess &&
           hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming) ==
               hipSuccess &&
           growhost(s, maxqueries + maxqueries / 2) == 0;
      continue;
    }
    ok = hipHostMalloc((void **) &s.hostq, nsym + VSA_QUERY_BACKPAD,
                       hipHostMallocDefault) == hipSuccess &&
         vsa_hip_malloc((void **) &q->symbols, nsym + VSA_QUERY_BACKPAD) ==
             hipSuccess &&
         vsa_hip_malloc((void **) &q->start, (maxqueries + 1) * 8) ==
             hipSuccess &&
         vsa_hip_malloc((void **) &q->length, (maxqueries + 1) * 8) ==
             hipSuccess &&
         hipEventCreateWithFlags(&s.uploaded, hipEventDisableTiming) ==
             hipSuccess &&
         hipEventCreateWithFlags(&s.downloaded, hipEventDisableTiming) ==
             hipSuccess;
    if (ok)
    {
      k_pipeline_starts<<<(unsigned int) ((maxqueries + 255) / 256), 256, 0,
                          p->up>>>(q->start, q->length, maxqueries,
                                   querylength);
      ok = hipGetLastError() == hipSuccess &&
           growhost(s, maxqueries + maxqueries / 2) == 0;
    }
  }
  ok = ok && hipStreamSynchronize(p->up) == hipSuccess;
  if (!ok)
  {
    VSA_ERROR("vsa_pipeline_open: out of (page-locked) memory for %lu "
              "queries of %u symbols per batch", (unsigned long) maxqueries,
              querylength);
    (void) hipGetLastError();
    *pipeline = p;
    vsa_pipeline_close(p);
    *pipeline = nullptr;
    return -100;
  }
  p->worker = std::thread(work, p);
  *pipeline = p;
  return 0;
}
} // namespace

// The host buffer of the next batch: maxqueries * querylength bytes, symbols
// of read i at [i * querylength, (i+1) * querylength).  NULL if all three
// batches are in flight: take results with vsa_pipeline_next first.  The
// matches handed out by the last vsa_pipeline_next stay valid unless their
// slot is the only one left, in which case it is taken.
extern "C" uint8_t *vsa_pipeline_hostbuffer(vsa_pipeline *p)
{
  if (p == nullptr)
  {
    return nullptr;
  }
  std::lock_guard<std::mutex> g(p->lock);
  if (p->filling < 0)
  {
    for (int k = 0; k < kSlots && p->filling < 0; k++)
    {
      if (p->slot[k].state == SLOT_FREE)
      {
        p->filling = k;
      }
    }
    if (p->filling < 0 && p->lastdelivered >= 0)
    {
      releaseslot(p, p->lastdelivered);
      p->filling = p->lastdelivered;
      p->lastdelivered = -1;
    }
  }
  return p->filling >= 0 ? p->slot[p->filling].hostq : nullptr;
}

// The same for a packed pipeline: *rows = room for maxqueries rows of
// vsa_packed_words(querylength) words, *special = room for maxspecial reads as
// bytes (vsa_pack_reads fills both).  0: a slot is handed out; 1: all three
// batches are in flight (take results with vsa_pipeline_next first).
extern "C" int vsa_pipeline_hostrows(vsa_pipeline *p, uint64_t **rows,
                                     uint8_t **special)
{
  if (p == nullptr || !p->packed || rows == nullptr || special == nullptr)
  {
    VSA_ERROR("vsa_pipeline_hostrows: bad argument (a packed pipeline?)");
    return -1;
  }
  *rows = nullptr;
  *special = nullptr;
  (void) vsa_pipeline_hostbuffer(p); // picks the slot (hostq is null here)
  std::lock_guard<std::mutex> g(p->lock);
  if (p->filling < 0)
  {
    return 1;
  }
  *rows = p->slot[p->filling].hostrows;
  *special = p->slot[p->filling].hostside;
  return 0;
}

namespace
{
int pipeline_submit(vsa_pipeline *p, uint64_t nq, uint64_t nspecial);
}

extern "C" int vsa_pipeline_submit(vsa_pipeline *p, uint64_t nq)
{
  if (p == nullptr || p->packed)
  {
    VSA_ERROR("vsa_pipeline_submit: bad argument (a packed pipeline takes "
              "vsa_pipeline_submit_packed)");
    return -1;
  }
  return pipeline_submit(p, nq, 0);
}

extern "C" int vsa_pipeline_submit_packed(vsa_pipeline *p, uint64_t nq,
                                          uint64_t nspecial)
{
  if (p == nullptr || !p->packed || nspecial > p->maxspecial)
  {
    VSA_ERROR("vsa_pipeline_submit_packed: bad argument");
    return -1;
  }
  return pipeline_submit(p, nq, nspecial);
}

namespace
{
int pipeline_submit(vsa_pipeline *p, uint64_t nq, uint64_t nspecial)
{
  if (p == nullptr || nq > p->maxqueries)
  {
    VSA_ERROR("vsa_pipeline_submit: bad argument");
    return -1;
  }
  int k;
  {
    std::lock_guard<std::mutex> g(p->lock);
    if (p->filling < 0)
    {
      VSA_ERROR("vsa_pipeline_submit: no buffer handed out "
                "(vsa_pipeline_hostbuffer first)");
      return -1;
    }
    k = p->filling;
    p->filling = -1;
  }
  Slot &s = p->slot[k];
  s.nq = nq;
  s.nside = nspecial;
  s.first = p->submitted;
  const uint64_t nsym = nq * (uint64_t) p->m;
  // (a batch that could not be queued leaves the pipeline as it was: the
  // buffer stays handed out, the numbering of the queries does not move)
  hipError_t e = vsa_set_device(p->index->device) != 0 ? hipErrorInvalidDevice
                                                       : hipSuccess;
  if (p->packed)
  {
    // (a flagged row names a read of the side list; the kernels clamp the
    // index to the list, so a row that lies cannot make them read elsewhere)
    const uint32_t W = p->roww;
    memset(s.hostrows + nq * W, 0, 32);
    if (e == hipSuccess)
    {
      e = hipMemcpyAsync(s.q->rows, s.hostrows, nq * W * 8 + 32,
                         hipMemcpyHostToDevice, p->up);
    }
    if (e == hipSuccess && nspecial > 0)
    {
      e = hipMemcpyAsync(s.q->side, s.hostside, nspecial * (uint64_t) p->m,
                         hipMemcpyHostToDevice, p->up);
    }
  } else
  {
    // what lies behind the last read must stop every comparison
    memset(s.hostq + nsym, 0xFF, VSA_QUERY_BACKPAD);
    if (e == hipSuccess)
    {
      e = hipMemcpyAsync(s.q->symbols, s.hostq, nsym + VSA_QUERY_BACKPAD,
                         hipMemcpyHostToDevice, p->up);
    }
  }
  if (e == hipSuccess)
  {
    e = hipEventRecord(s.uploaded, p->up);
  }
  if (e != hipSuccess)
  {
    VSA_ERROR("vsa_pipeline_submit: upload failed: %s", hipGetErrorString(e));
    std::lock_guard<std::mutex> g(p->lock);
    p->filling = k;
    return -100;
  }
  p->submitted += nq;
  {
    std::lock_guard<std::mutex> g(p->lock);
    s.state = SLOT_SUBMITTED;
    p->todo.push_back(k);
    p->inorder.push_back(k);
  }
  p->wake.notify_all();
  return 0;
}
} // namespace

// The matches of the oldest batch not delivered yet (host memory, valid until
// the next call of vsa_pipeline_next / _hostbuffer that needs the slot).
// Returns 1 if nothing is outstanding, 0 on success, the engine's negative
// code if that batch failed (its matches up to the error are delivered).
extern "C" int vsa_pipeline_next(vsa_pipeline *p, const vsa_match **matches,
                                 uint64_t *count)
{
  if (p == nullptr || matches == nullptr || count == nullptr)
  {
    VSA_ERROR("vsa_pipeline_next: NULL argument");
    return -1;
  }
  *matches = nullptr;
  *count = 0;
  int k;
  {
    std::unique_lock<std::mutex> g(p->lock);
    if (p->lastdelivered >= 0)
    {
      releaseslot(p, p->lastdelivered);
      p->lastdelivered = -1;
    }
    if (p->inorder.empty())
    {
      return 1;
    }
    k = p->inorder.front();
    p->wake.wait(g, [&] { return p->slot[k].state == SLOT_DONE; });
    p->inorder.pop_front();
    p->lastdelivered = k;
  }
  Slot &s = p->slot[k];
  if (hipEventSynchronize(s.downloaded) != hipSuccess)
  {
    VSA_ERROR("vsa_pipeline_next: copy of the matches failed");
    return -100;
  }
  *matches = s.hostm;
  *count = s.count;
  if (s.rc != 0)
  {
    VSA_ERROR("%s", s.message.c_str());
  }
  return s.rc;
}

namespace
{

// records -> 16 bytes each (vsa_match16, include/vstree_amd.h)
__global__ void k_compact_records(const vsa_match *__restrict__ in, uint64_t n,
                                  vsa_match16 *__restrict__ out)
{
  const uint64_t i = vsa_bid() * blockDim.x + threadIdx.x;
  if (i < n)
  {
    const vsa_match m = in[i];
    vsa_match16 c;
    c.dbstart_length = (m.dbstart << 24) | (m.length & 0xFFFFFFu);
    c.queryseq_querystart = (m.queryseq << 16) | (m.querystart & 0xFFFFu);
    out[i] = c;
  }
}

// -mum: the filter over the candidates of ALL batches submitted so far (the
// batches must have been taken with vsa_pipeline_next); MUMs in host memory,
// ascending dbstart, valid until the pipeline is closed or finished again.
// compact: 16 bytes per MUM instead of 32 over the host link.
int pipeline_finish(vsa_pipeline *p, bool compact, const void **matches,
                    uint64_t *count, vsa_stats *stats)
{
  if (p == nullptr || matches == nullptr || count == nullptr || p->mode != 3)
  {
    VSA_ERROR("vsa_pipeline_finish: bad argument (a -mum pipeline?)");
    return -1;
  }
  *matches = nullptr;
  *count = 0;
  {
    std::lock_guard<std::mutex> g(p->lock);
    if (!p->inorder.empty())
    {
      VSA_ERROR("vsa_pipeline_finish: batches are still outstanding");
      return -1;
    }
  }
  {
    std::lock_guard<std::mutex> g(p->lock);
    if (p->failed != 0)
    {
      // a batch of this job failed: no list; the next job starts afresh
      const int frc = p->failed;
      VSA_ERROR("vsa_pipeline_finish: a batch of the job failed: %s",
                p->failure.c_str());
      p->failed = 0;
      p->failure.clear();
      p->nrows = 0;
      p->candidates = 0;
      return frc;
    }
  }
  if (vsa_set_device(p->index->device) != 0)
  {
    return -100;
  }
  if (compact && (p->m > 0xFFFFu || (p->index->n >> 40) != 0))
  {
    VSA_ERROR("vsa_pipeline_finish16: reads of %u symbols on a text of %lu: "
              "a match does not fit 16 bytes", p->m,
              (unsigned long) p->index->n);
    return -1;
  }
  vsa_result *res = nullptr;
  const int rc = vsa_mumuniqueinquery_range_packed(
      p->rows, p->nrows, p->lengthbits, p->index->n, p->index->device, 0,
      &res);
  if (rc != 0)
  {
    p->nrows = 0; // (whatever the outcome, the next job starts afresh)
    p->candidates = 0;
    return rc;
  }
  const uint64_t c = vsa_result_count(res);
  int out = 0;
  if (c * (compact ? 16 : 32) > p->hostmumscap * 32)
  {
    // page-locking costs about a third of a second per GB: the buffer is
    // kept for the next job
    if (p->hostmums != nullptr)
    {
      (void) hipHostFree(p->hostmums);
      p->hostmums = nullptr;
    }
    p->hostmumscap = (c + c / 8 + 1024) / (compact ? 2 : 1) + 1;
    if (hipHostMalloc((void **) &p->hostmums,
                      p->hostmumscap * sizeof(vsa_match),
                      hipHostMallocDefault) != hipSuccess)
    {
      (void) hipGetLastError();
      p->hostmumscap = 0;
      VSA_ERROR("vsa_pipeline_finish: no page-locked memory for %lu MUMs",
                (unsigned long) c);
      out = -100;
    }
  }
  if (out == 0 && c > 0 && !compact &&
      vsa_result_fetch(res, p->hostmums, c) != 0)
  {
    out = -100;
  }
  if (out == 0 && c > 0 && compact)
  {
    void *small = nullptr;
    if (vsa_dev_alloc(&small, c * sizeof(vsa_match16)) != 0)
    {
      out = -100;
    } else
    {
      k_compact_records<<<vsa_grid((c + 255) / 256), 256, 0, p->down>>>(
          res->matches, c, (vsa_match16 *) small);
      if (hipGetLastError() != hipSuccess ||
          hipMemcpyAsync(p->hostmums, small, c * sizeof(vsa_match16),
                         hipMemcpyDeviceToHost, p->down) != hipSuccess ||
          hipStreamSynchronize(p->down) != hipSuccess)
      {
        (void) hipGetLastError();
        VSA_ERROR("vsa_pipeline_finish16: copy of %lu MUMs to the host failed",
                  (unsigned long) c);
        out = -100;
      }
      vsa_dev_free(small);
    }
  }
  if (stats != nullptr)
  {
    (void) vsa_result_getstats(res, stats);
    stats->candidates = p->candidates;
  }
  vsa_result_free(res);
  if (out == 0)
  {
    *matches = p->hostmums;
    *count = c;
  }
  p->nrows = 0; // the next job starts afresh
  p->candidates = 0;
  return out;
}

} // namespace

extern "C" int vsa_pipeline_finish(vsa_pipeline *p, const vsa_match **matches,
                                   uint64_t *count, vsa_stats *stats)
{
  return pipeline_finish(p, false, (const void **) matches, count, stats);
}

extern "C" int vsa_pipeline_finish16(vsa_pipeline *p,
                                     const vsa_match16 **matches,
                                     uint64_t *count, vsa_stats *stats)
{
  return pipeline_finish(p, true, (const void **) matches, count, stats);
}

// ---- for a caller that runs one pipeline per GPU (multi_gpu.cpp) ----------

// the number the first query of the NEXT submitted batch gets (a job whose
// batches are dealt out to several pipelines numbers its queries itself)
extern "C" int vsa_pipeline_set_offset(vsa_pipeline *p, uint64_t firstquery)
{
  if (p == nullptr)
  {
    VSA_ERROR("vsa_pipeline_set_offset: NULL argument");
    return -1;
  }
  p->submitted = firstquery;
  return 0;
}

// -mum pipelines: the candidate rows of all batches since the last job, where
// they lie in device memory (16 bytes each: sort key, value; valid until the
// next batch is submitted), instead of vsa_pipeline_finish -- the caller
// filters them together with the rows of other GPUs.  Ends the job.
extern "C" int vsa_pipeline_take_candidates(vsa_pipeline *p,
                                            const void **device_rows,
                                            uint64_t *nrows,
                                            uint32_t *lengthbits)
{
  if (p == nullptr || device_rows == nullptr || nrows == nullptr ||
      lengthbits == nullptr || p->mode != 3)
  {
    VSA_ERROR("vsa_pipeline_take_candidates: bad argument (a -mum "
              "pipeline?)");
    return -1;
  }
  *device_rows = nullptr;
  *nrows = 0;
  *lengthbits = p->lengthbits;
  std::lock_guard<std::mutex> g(p->lock);
  if (!p->inorder.empty())
  {
    VSA_ERROR("vsa_pipeline_take_candidates: batches are still outstanding");
    return -1;
  }
  if (p->failed != 0)
  {
    const int frc = p->failed;
    VSA_ERROR("a batch of the job failed: %s", p->failure.c_str());
    p->failed = 0;
    p->failure.clear();
    p->nrows = 0;
    p->candidates = 0;
    return frc;
  }
  *device_rows = p->rows;
  *nrows = p->nrows;
  p->nrows = 0; // the next job starts afresh
  p->candidates = 0;
  return 0;
}

extern "C" void vsa_pipeline_close(vsa_pipeline *p)
{
  if (p == nullptr)
  {
    return;
  }
  {
    std::lock_guard<std::mutex> g(p->lock);
    p->stop = true;
  }
  p->wake.notify_all();
  if (p->worker.joinable())
  {
    p->worker.join();
  }
  (void) hipSetDevice(p->index->device);
  (void) hipDeviceSynchronize();
  for (int k = 0; k < kSlots; k++)
  {
    Slot &s = p->slot[k];
    if (s.res != nullptr)
    {
      vsa_result_free(s.res);
    }
    (void) hipHostFree(s.hostq);
    (void) hipHostFree(s.hostrows);
    (void) hipHostFree(s.hostside);
    (void) hipHostFree(s.hostm);
    vsa_queries_free(s.q);
    if (s.uploaded != nullptr)
    {
      (void) hipEventDestroy(s.uploaded);
      (void) hipEventDestroy(s.downloaded);
    }
  }
  (void) hipHostFree(p->hostmums);
  (void) hipFree(p->rows);
  if (p->up != nullptr)
  {
    (void) hipStreamDestroy(p->up);
  }
  if (p->down != nullptr)
  {
    (void) hipStreamDestroy(p->down);
  }
  delete p;
}
