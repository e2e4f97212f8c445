// Handles of the C ABI (include/vstree_amd.h): index upload, query batches,
// result lists, device utilities.
#include "vsa_internal.hpp"
#include <atomic>
#include <thread>
#include <time.h>
#include <algorithm>

static thread_local char g_errbuf[VSA_ERRBUF_SIZE] = "";

extern "C" char *vsa_errbuf()
{
  return g_errbuf;
}

extern "C" const char *vsa_messagespace(void)
{
  return g_errbuf;
}

int vsa_set_device(int device)
{
  VSA_HIP(hipSetDevice(device));
  return 0;
}

extern "C" int vsa_device_count(void)
{
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess)
  {
    return 0;
  }
  return n;
}

extern "C" int vsa_device_synchronize(int device)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  VSA_HIP(hipDeviceSynchronize());
  return 0;
}

extern "C" int vsa_device_malloc(uint64_t bytes, int device, void **ptr)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  VSA_HIP(vsa_hip_malloc(ptr, bytes > 0 ? bytes : 16));
  return 0;
}

extern "C" int vsa_device_upload(void *device_dst, const void *host_src,
                                 uint64_t bytes, int device)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  if (bytes > 0)
  {
    VSA_HIP(hipMemcpy(device_dst, host_src, bytes, hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int vsa_device_download(void *host_dst, const void *device_src,
                                   uint64_t bytes, int device)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  if (bytes > 0)
  {
    VSA_HIP(hipMemcpy(host_dst, device_src, bytes, hipMemcpyDeviceToHost));
  }
  return 0;
}

extern "C" int vsa_device_free(void *ptr, int device)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  VSA_HIP(hipFree(ptr));
  return 0;
}

// ---------------------------------------------------------------------------
// index
// ---------------------------------------------------------------------------

namespace
{

// Host table -> device array, through page-locked staging buffers filled by
// several host threads at once.  The tables of an index that vmatch has mapped
// are pageable memory: one hipMemcpy of 12 GB from there is staged by the
// runtime in ONE thread (3 GB/s here), and the 64-bit suftab of a reference
// index (include/types.h:41-61) had to be narrowed to 32 bits in one more
// single-threaded loop first -- 1.9 of the 2.8 s a drop-in run spent before
// its first search (profiles/r02).  Each worker takes chunks in turn,
// converts or copies its chunk into its own pinned buffer and sends it on its
// own stream; the copies of the others run meanwhile.
//   hostbits / devbytes: width of an entry on either side (8 -> 1 byte for
//   tis, lcp, bwt; 32 / 64 -> 4 / 8 for suf, bck, llv)
#define VSA_UPLOAD_CHUNKBYTES (8u << 20)
#define VSA_UPLOAD_MAXTHREADS 12u

// the page-locked buffers of the workers, made once per index upload (page
// locking costs about a third of a second per GB)
struct UploadPool
{
  uint8_t *pin[VSA_UPLOAD_MAXTHREADS][2];
  unsigned int nthreads;
  UploadPool() : nthreads(0)
  {
    memset(pin, 0, sizeof pin);
  }
  int init()
  {
    unsigned int want = std::thread::hardware_concurrency();
    if (const char *e = getenv("VSA_UPLOAD_THREADS"))
    {
      want = (unsigned int) atoi(e);
    }
    want = std::max(1u, std::min(want, VSA_UPLOAD_MAXTHREADS));
    for (unsigned int t = 0; t < want; t++)
    {
      for (int k = 0; k < 2; k++)
      {
        if (hipHostMalloc((void **) &pin[t][k], VSA_UPLOAD_CHUNKBYTES,
                          hipHostMallocDefault) != hipSuccess)
        {
          (void) hipGetLastError();
          pin[t][k] = nullptr;
          return nthreads > 0 ? 0 : -100; // (fewer workers will do)
        }
      }
      nthreads = t + 1;
    }
    return 0;
  }
  ~UploadPool()
  {
    for (unsigned int t = 0; t < VSA_UPLOAD_MAXTHREADS; t++)
    {
      for (int k = 0; k < 2; k++)
      {
        if (pin[t][k] != nullptr)
        {
          (void) hipHostFree(pin[t][k]);
        }
      }
    }
  }
};

int upload_table(UploadPool &pool, const void *host, uint32_t hostbytes,
                 uint64_t count, uint32_t devbytes, void *dev, int device)
{
  if (count == 0)
  {
    return 0;
  }
  // entries per chunk: the wider side fills the buffer
  const uint64_t chunk =
      VSA_UPLOAD_CHUNKBYTES / std::max(hostbytes, devbytes);
  const uint64_t nchunks = (count + chunk - 1) / chunk;
  const unsigned int nthreads =
      (unsigned int) std::min<uint64_t>(pool.nthreads, nchunks);
  std::atomic<uint64_t> next(0);
  std::atomic<int> failed(0);
  auto work = [&](unsigned int me) {
    if (hipSetDevice(device) != hipSuccess)
    {
      failed = 1;
      return;
    }
    hipStream_t st = nullptr;
    uint8_t *pin[2] = {pool.pin[me][0], pool.pin[me][1]};
    hipEvent_t sent[2] = {nullptr, nullptr};
    bool ok = hipStreamCreateWithFlags(&st, hipStreamNonBlocking) == hipSuccess;
    for (int k = 0; ok && k < 2; k++)
    {
      ok = hipEventCreateWithFlags(&sent[k], hipEventDisableTiming) ==
           hipSuccess;
    }
    int slot = 0;
    bool used[2] = {false, false};
    while (ok && failed == 0)
    {
      const uint64_t c = next.fetch_add(1);
      if (c >= nchunks)
      {
        break;
      }
      const uint64_t first = c * chunk, m = std::min(chunk, count - first);
      if (used[slot] && hipEventSynchronize(sent[slot]) != hipSuccess)
      {
        ok = false;
        break;
      }
      if (hostbytes == devbytes)
      {
        memcpy(pin[slot], (const uint8_t *) host + first * hostbytes,
               m * hostbytes);
      } else if (hostbytes == 8) // 64 -> 32
      {
        const uint64_t *src = (const uint64_t *) host + first;
        uint32_t *dst = (uint32_t *) pin[slot];
        for (uint64_t i = 0; i < m; i++)
        {
          dst[i] = (uint32_t) src[i];
        }
      } else // 32 -> 64
      {
        const uint32_t *src = (const uint32_t *) host + first;
        uint64_t *dst = (uint64_t *) pin[slot];
        for (uint64_t i = 0; i < m; i++)
        {
          dst[i] = src[i];
        }
      }
      ok = hipMemcpyAsync((uint8_t *) dev + first * devbytes, pin[slot],
                          m * devbytes, hipMemcpyHostToDevice, st) ==
               hipSuccess &&
           hipEventRecord(sent[slot], st) == hipSuccess;
      used[slot] = true;
      slot ^= 1;
    }
    if (st != nullptr && hipStreamSynchronize(st) != hipSuccess)
    {
      ok = false;
    }
    for (int k = 0; k < 2; k++)
    {
      if (sent[k] != nullptr)
      {
        (void) hipEventDestroy(sent[k]);
      }
    }
    if (st != nullptr)
    {
      (void) hipStreamDestroy(st);
    }
    if (!ok)
    {
      (void) hipGetLastError();
      failed = 1;
    }
  };
  std::vector<std::thread> threads;
  for (unsigned int t = 1; t < nthreads; t++)
  {
    threads.emplace_back(work, t);
  }
  work(0);
  for (std::thread &t : threads)
  {
    t.join();
  }
  if (failed != 0)
  {
    VSA_ERROR("upload of an index table failed");
    return -100;
  }
  return 0;
}


// VSA_TRACE=1: where the time of an index upload goes (stderr)
struct UploadTrace
{
  double t0;
  bool on;
  UploadTrace()
  {
    const char *e = getenv("VSA_TRACE");
    on = e != nullptr && strcmp(e, "0") != 0;
    t0 = now();
  }
  static double now()
  {
    struct timespec ts;
    clock_gettime(CLOCK_MONOTONIC, &ts);
    return (double) ts.tv_sec + 1e-9 * (double) ts.tv_nsec;
  }
  void step(const char *what)
  {
    if (on)
    {
      const double t = now();
      fprintf(stderr, "vstree_amd: upload: %-28s %.3f s\n", what, t - t0);
      t0 = t;
    }
  }
};

uint64_t powu64(uint64_t b, uint32_t e)
{
  uint64_t r = 1;
  while (e-- > 0)
  {
    r *= b;
  }
  return r;
}

// Uploaded tables come from files anybody may have written: entries that
// point outside the tables would turn into out-of-bounds reads in the search
// kernels (a GPU fault, not an error return).  One pass over suf, bck and llv:
// bit 0 of *bad: a suf entry beyond n; bit 1: bck not non-decreasing or beyond
// n + 1; bit 2: an llv index beyond n or not increasing.
template <typename IDX>
__global__ void __launch_bounds__(256)
k_validate_tables(const IDX *__restrict__ suf, uint64_t n,
                  const IDX *__restrict__ bck, uint64_t nbck,
                  const IDX *__restrict__ llv, uint64_t nllv,
                  unsigned int *__restrict__ bad)
{
  const uint64_t stride = vsa_nblocks() * 256;
  unsigned int mine = 0;
  for (uint64_t i = vsa_bid() * 256 + threadIdx.x; i <= n; i += stride)
  {
    mine |= ((uint64_t) suf[i] > n) ? 1u : 0u;
  }
  if (bck != nullptr)
  {
    for (uint64_t i = vsa_bid() * 256 + threadIdx.x; i < nbck; i += stride)
    {
      const uint64_t v = bck[i];
      mine |= (v > n + 1 || (i > 0 && (uint64_t) bck[i - 1] > v)) ? 2u : 0u;
    }
  }
  for (uint64_t i = vsa_bid() * 256 + threadIdx.x; i < nllv; i += stride)
  {
    const uint64_t v = llv[2 * i];
    mine |= (v > n || (i > 0 && (uint64_t) llv[2 * i - 2] >= v)) ? 4u : 0u;
  }
  if (mine != 0)
  {
    atomicOr(bad, mine);
  }
}

template <typename IDX>
int validate_tables(const vsa_index *ix, unsigned int *verdict)
{
  unsigned int *dbad = nullptr;
  VSA_HIP(vsa_hip_malloc((void **) &dbad, 4));
  VSA_HIP(hipMemsetAsync(dbad, 0, 4, ix->stream));
  k_validate_tables<IDX><<<4096, 256, 0, ix->stream>>>(
      (const IDX *) ix->suf, ix->n, (const IDX *) ix->bck,
      2 * ix->numofcodes, (const IDX *) ix->llv, ix->nllv, dbad);
  VSA_HIP(hipGetLastError());
  VSA_HIP(hipMemcpyAsync(verdict, dbad, 4, hipMemcpyDeviceToHost,
                         ix->stream));
  VSA_HIP(hipStreamSynchronize(ix->stream));
  (void) hipFree(dbad);
  return 0;
}

} // namespace

// allocates the device tables of an index of the given shape; contents are
// filled by the caller (upload or the GPU builder)
int vsa_index_alloc(uint64_t n, uint32_t pl, uint32_t numofchars,
                    uint64_t nllv, bool withbwt, int device,
                    vsa_index **out, bool mayforcewide)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_index *ix = new vsa_index;
  memset((void *) ix, 0, sizeof *ix);
  ix->device = device;
  ix->n = n;
  ix->pl = pl;
  ix->numofchars = numofchars;
  ix->nllv = nllv;
  ix->numofcodes = powu64(numofchars, pl);
  ix->isize = (n + 1 <= 0xFFFFFFFFull) ? 4 : 8;
  ix->lcpquirk = -1;
  ix->qspeedup = 2; // the reference's default, Vmengine/mparms.h:53
  {
    // VSA_FORCE_WIDE=1: 64-bit device tables whatever the length (tests of
    // the wide instantiations on small inputs)
    // (not for the GPU builder, which writes 32-bit tables)
    const char *wide = getenv("VSA_FORCE_WIDE");
    if (mayforcewide && wide != nullptr && strcmp(wide, "1") == 0)
    {
      ix->isize = 8;
    }
  }
  *out = ix;
  VSA_HIP(hipStreamCreateWithFlags(&ix->stream, hipStreamNonBlocking));
  const uint64_t tisbytes = VSA_TIS_FRONTPAD + n + VSA_TIS_BACKPAD,
                 sufbytes = (n + 1) * ix->isize, lcpbytes = n + 1 + 32,
                 llvbytes = 2 * nllv * ix->isize + 16,
                 bckbytes = 2 * ix->numofcodes * ix->isize;
  VSA_HIP(vsa_hip_malloc((void **) &ix->tis_alloc, tisbytes));
  VSA_HIP(vsa_hip_malloc(&ix->suf, sufbytes));
  VSA_HIP(vsa_hip_malloc((void **) &ix->lcp, lcpbytes));
  VSA_HIP(vsa_hip_malloc(&ix->llv, llvbytes));
  VSA_HIP(vsa_hip_malloc(&ix->bck, bckbytes));
  ix->device_bytes = tisbytes + sufbytes + lcpbytes + llvbytes + bckbytes;
  if (withbwt)
  {
    VSA_HIP(vsa_hip_malloc((void **) &ix->bwt, n + 1 + 32));
    ix->device_bytes += n + 1 + 32;
  }
  // pads: separator symbols around the text, zeros behind lcp
  VSA_HIP(hipMemsetAsync(ix->tis_alloc, 0xFF, VSA_TIS_FRONTPAD, ix->stream));
  VSA_HIP(hipMemsetAsync(ix->tis_alloc + VSA_TIS_FRONTPAD + n, 0xFF,
                         VSA_TIS_BACKPAD, ix->stream));
  VSA_HIP(hipMemsetAsync(ix->lcp + n + 1, 0, 32, ix->stream));
  VSA_HIP(hipStreamSynchronize(ix->stream));
  return 0;
}

// A replica of an index in the memory of another device (or of the same one):
// every table, the derived search tables included, is copied device to device
// -- over xGMI between two GPUs of a node -- and nothing is built again.
extern "C" int vsa_index_clone(const vsa_index *src, int device,
                               vsa_index **clone)
{
  if (src == nullptr || clone == nullptr)
  {
    VSA_ERROR("vsa_index_clone: NULL argument");
    return -1;
  }
  *clone = nullptr;
  vsa_index *ix = nullptr;
  // same shapes (isize of the source, whatever VSA_FORCE_WIDE says now)
  int rc = vsa_index_alloc(src->n, src->pl, src->numofchars, src->nllv,
                           src->bwt != nullptr, device, &ix, false);
  auto fail = [&](int code) {
    vsa_index_close(ix);
    return code;
  };
  if (rc != 0)
  {
    return fail(rc);
  }
  if (ix->isize != src->isize)
  {
    // vsa_index_alloc chose 4-byte entries, the source holds 8-byte ones
    (void) hipFree(ix->suf);
    (void) hipFree(ix->llv);
    (void) hipFree(ix->bck);
    ix->suf = ix->llv = ix->bck = nullptr;
    ix->isize = src->isize;
    if (vsa_hip_malloc(&ix->suf, (src->n + 1) * ix->isize) != hipSuccess ||
        vsa_hip_malloc(&ix->llv, 2 * src->nllv * ix->isize + 16) != hipSuccess ||
        vsa_hip_malloc(&ix->bck, 2 * ix->numofcodes * ix->isize) != hipSuccess)
    {
      VSA_ERROR("vsa_index_clone: out of device memory");
      return fail(-100);
    }
  }
  const uint64_t n = src->n;
  struct Piece
  {
    void **dst;
    const void *from;
    uint64_t bytes;
    bool allocate;
  };
  const uint64_t ncodes = src->D > 0 ? 1ull << (2 * src->D) : 0,
                 nblocks = (n >> 6) + 1, nwaves = (nblocks + 63) / 64;
  Piece pieces[] = {
      {(void **) &ix->tis_alloc, src->tis_alloc,
       VSA_TIS_FRONTPAD + n + VSA_TIS_BACKPAD, false},
      {&ix->suf, src->suf, (n + 1) * src->isize, false},
      {(void **) &ix->lcp, src->lcp, n + 1 + 32, false},
      {&ix->llv, src->llv, 2 * src->nllv * src->isize, false},
      {&ix->bck, src->bck, 2 * src->numofcodes * src->isize, false},
      {(void **) &ix->bwt, src->bwt, n + 1 + 32, false},
      {(void **) &ix->esa8, src->esa8, (n + 1) * 8 + 64, true},
      {(void **) &ix->bck2, src->bck2, 2 * ncodes * 4 + 16, true},
      {(void **) &ix->slot16, src->slot16,
       (uint64_t) src->slotwords * ncodes * 8 + 32, true},
      {(void **) &ix->tis2, src->tis2, nblocks * 16 + 64, true},
      {(void **) &ix->spec64, src->spec64, nwaves * 8 + 64, true}};
  for (const Piece &p : pieces)
  {
    if (p.from == nullptr || p.bytes == 0)
    {
      continue;
    }
    if (p.allocate)
    {
      if (vsa_hip_malloc(p.dst, p.bytes) != hipSuccess)
      {
        VSA_ERROR("vsa_index_clone: out of device memory");
        return fail(-100);
      }
      ix->device_bytes += p.bytes;
    }
    const hipError_t e =
        hipMemcpyPeerAsync(*p.dst, device, p.from, src->device, p.bytes,
                           ix->stream);
    if (e != hipSuccess)
    {
      VSA_ERROR("vsa_index_clone: copy from device %d to device %d: %s",
                src->device, device, hipGetErrorString(e));
      return fail(-100);
    }
  }
  if (hipStreamSynchronize(ix->stream) != hipSuccess)
  {
    VSA_ERROR("vsa_index_clone: copy failed");
    return fail(-100);
  }
  ix->device_bytes = src->device_bytes;
  ix->D = src->D;
  ix->tune = src->tune;
  ix->qspeedup = src->qspeedup;
  ix->slotwords = src->slotwords;
  ix->firstspecial = src->firstspecial;
  ix->querysepposition = src->querysepposition;
  ix->hasindexedqueries = src->hasindexedqueries;
  ix->lcpquirk = src->lcpquirk;
  *clone = ix;
  return 0;
}

extern "C" void vsa_index_close(vsa_index *ix)
{
  if (ix == nullptr)
  {
    return;
  }
  (void) hipSetDevice(ix->device);
  (void) hipFree(ix->tis_alloc);
  (void) hipFree(ix->suf);
  (void) hipFree(ix->lcp);
  (void) hipFree(ix->llv);
  (void) hipFree(ix->bck);
  (void) hipFree(ix->bwt);
  (void) hipFree(ix->esa8);
  (void) hipFree(ix->bck2);
  (void) hipFree(ix->slot16);
  (void) hipFree(ix->tis2);
  (void) hipFree(ix->spec64);
  (void) hipFree(ix->isa32);
  (void) hipFree(ix->repbits);
  if (ix->stream != nullptr)
  {
    vsa_dev_forget_stream(ix->stream);
    (void) hipStreamDestroy(ix->stream);
  }
  delete ix;
}

extern "C" int vsa_index_from_tables(const vsa_tables *t, int device,
                                     vsa_index **index)
{
  if (t == nullptr || index == nullptr)
  {
    VSA_ERROR("vsa_index_from_tables: NULL argument");
    return -1;
  }
  *index = nullptr;
  // kurtz-basic/multiseq-adv.c:1856-1898 rejects an index whose integer
  // size does not fit the program; here both widths are accepted
  if (t->integersize != 32 && t->integersize != 64)
  {
    VSA_ERROR("integersize=%u: only 32 and 64 bit indexes are supported",
              t->integersize);
    return -2;
  }
  // tis may be missing together with bck: vmatch maps neither for the scans
  // over the index itself (-supermax, repeats: Vmatch/mapdemand.c:100-210);
  // every search that reads the text needs bck as well and checks for it
  if ((t->tis == nullptr && t->bck != nullptr) || t->suf == nullptr ||
      t->lcp == nullptr || (t->largelcpvalues > 0 && t->llv == nullptr))
  {
    VSA_ERROR("tables tis, suf, lcp (and llv) are required");
    return -3;
  }
  // bck may be missing: vmatch does not map it for the self-index MUM scan
  // (Vmatch/mapdemand.c:100-210); the query entry points then refuse to run
  if (t->numofchars == 0 || t->numofchars > 253 ||
      (t->bck != nullptr && t->prefixlength == 0))
  {
    VSA_ERROR("prefixlength=%u numofchars=%u: not a usable bucket table",
              t->prefixlength, t->numofchars);
    return -4;
  }
  vsa_index *ix = nullptr;
  int rc = vsa_index_alloc(t->totallength,
                           t->bck != nullptr ? t->prefixlength : 0,
                           t->numofchars, t->largelcpvalues,
                           t->bwt != nullptr, device, &ix);
  if (rc != 0)
  {
    vsa_index_close(ix);
    return rc;
  }
  ix->querysepposition = t->querysepposition;
  ix->hasindexedqueries = t->hasindexedqueries;
  const uint64_t n = ix->n;
  hipStream_t s = ix->stream;
  auto fail = [&](int code) {
    vsa_index_close(ix);
    return code;
  };
  UploadTrace tr;
  UploadPool pool;
  // (the workers' streams are their own: what is queued on the index's stream
  // -- the padding around the text -- is waited for first)
  if (pool.init() != 0 || hipStreamSynchronize(s) != hipSuccess ||
      (n > 0 && t->tis != nullptr &&
       upload_table(pool, t->tis, 1, n, 1, ix->tis_alloc + VSA_TIS_FRONTPAD,
                    device)) ||
      upload_table(pool, t->lcp, 1, n + 1, 1, ix->lcp, device) ||
      (t->bwt != nullptr &&
       upload_table(pool, t->bwt, 1, n + 1, 1, ix->bwt, device)))
  {
    VSA_ERROR("upload of the text tables failed");
    return fail(-100);
  }
  tr.step("tis, lcp, bwt");
  if (t->bck == nullptr)
  {
    (void) hipFree(ix->bck);
    ix->bck = nullptr;
  }
  const uint32_t hb = t->integersize / 8;
  if (upload_table(pool, t->suf, hb, n + 1, ix->isize, ix->suf, device) ||
      (t->bck != nullptr &&
       upload_table(pool, t->bck, hb, 2 * ix->numofcodes, ix->isize, ix->bck,
                    device)) ||
      upload_table(pool, t->llv, hb, 2 * ix->nllv, ix->isize, ix->llv, device))
  {
    return fail(-100);
  }
  tr.step("suf, bck, llv");
  {
    unsigned int bad = 0;
    if (ix->isize == 4 ? validate_tables<uint32_t>(ix, &bad)
                       : validate_tables<uint64_t>(ix, &bad))
    {
      return fail(-100);
    }
    if (bad != 0)
    {
      VSA_ERROR("inconsistent index tables:%s%s%s",
                (bad & 1u) ? " a suf entry beyond the text;" : "",
                (bad & 2u) ? " bck not non-decreasing or beyond the table;"
                           : "",
                (bad & 4u) ? " llv indices not increasing or beyond the table;"
                           : "");
      return fail(-2);
    }
  }
  tr.step("validation");
  if (t->bck != nullptr && vsa_index_make_esa8(ix) != 0)
  {
    return fail(-100);
  }
  tr.step("derived search tables");
  *index = ix;
  return 0;
}

extern "C" int vsa_index_getinfo(const vsa_index *ix, vsa_index_info *info)
{
  if (ix == nullptr || info == nullptr)
  {
    VSA_ERROR("vsa_index_getinfo: NULL argument");
    return -1;
  }
  info->totallength = ix->n;
  info->numofcodes = ix->numofcodes;
  info->largelcpvalues = ix->nllv;
  info->device_bytes = ix->device_bytes;
  info->prefixlength = ix->pl;
  info->numofchars = ix->numofchars;
  info->device_integersize = ix->isize * 8;
  info->device = ix->device;
  info->hasindexedqueries = ix->hasindexedqueries;
  info->hasbwt = ix->bwt != nullptr;
  info->deepprefix = ix->esa8 != nullptr ? ix->D : 0;
  return 0;
}

extern "C" int vsa_index_set_queryseparator(vsa_index *ix,
                                            uint64_t querysepposition)
{
  if (ix == nullptr || querysepposition >= ix->n)
  {
    VSA_ERROR("vsa_index_set_queryseparator: position outside the text");
    return -1;
  }
  uint8_t sym = 0;
  if (vsa_set_device(ix->device) != 0)
  {
    return -100;
  }
  VSA_HIP(hipMemcpy(&sym, ix->tis_alloc + VSA_TIS_FRONTPAD + querysepposition,
                    1, hipMemcpyDeviceToHost));
  if (sym != VSA_SEPARATOR)
  {
    VSA_ERROR("position %lu has no separator",
              (unsigned long) querysepposition);
    return -2;
  }
  ix->querysepposition = querysepposition;
  ix->hasindexedqueries = 1;
  return 0;
}

extern "C" int vsa_index_set_queryspeedup(vsa_index *ix, uint32_t queryspeedup)
{
  if (ix == nullptr || (queryspeedup != 0 && queryspeedup != 2))
  {
    // the message of Vmengine/fquery.c:433-436 for values it does not know;
    // 1 is refused by the reference's parser, 3..5 are undocumented variants
    VSA_ERROR("illegal speedup value %lu", (unsigned long) queryspeedup);
    return -1;
  }
  ix->qspeedup = queryspeedup;
  return 0;
}

extern "C" int vsa_index_download(const vsa_index *ix, uint8_t *tis,
                                  void *suf, uint8_t *lcp, void *llv,
                                  void *bck, uint8_t *bwt)
{
  if (ix == nullptr)
  {
    VSA_ERROR("vsa_index_download: NULL argument");
    return -1;
  }
  if (vsa_set_device(ix->device) != 0)
  {
    return -100;
  }
  const uint64_t n = ix->n;
  if (tis != nullptr && n > 0)
  {
    VSA_HIP(hipMemcpy(tis, ix->tis_alloc + VSA_TIS_FRONTPAD, n,
                      hipMemcpyDeviceToHost));
  }
  if (suf != nullptr)
  {
    VSA_HIP(hipMemcpy(suf, ix->suf, (n + 1) * ix->isize,
                      hipMemcpyDeviceToHost));
  }
  if (lcp != nullptr)
  {
    VSA_HIP(hipMemcpy(lcp, ix->lcp, n + 1, hipMemcpyDeviceToHost));
  }
  if (llv != nullptr && ix->nllv > 0)
  {
    VSA_HIP(hipMemcpy(llv, ix->llv, 2 * ix->nllv * ix->isize,
                      hipMemcpyDeviceToHost));
  }
  if (bck != nullptr)
  {
    VSA_HIP(hipMemcpy(bck, ix->bck, 2 * ix->numofcodes * ix->isize,
                      hipMemcpyDeviceToHost));
  }
  if (bwt != nullptr && ix->bwt != nullptr)
  {
    VSA_HIP(hipMemcpy(bwt, ix->bwt, n + 1, hipMemcpyDeviceToHost));
  }
  return 0;
}

// ---------------------------------------------------------------------------
// queries
// ---------------------------------------------------------------------------

extern "C" void vsa_queries_free(vsa_queries *q)
{
  if (q == nullptr)
  {
    return;
  }
  (void) hipSetDevice(q->device);
  (void) hipFree(q->symbols);
  (void) hipFree(q->start);
  (void) hipFree(q->length);
  if (q->ownsrows)
  {
    (void) hipFree(q->rows);
    (void) hipFree(q->side);
  }
  delete q;
}

namespace
{

void summarise_lengths(vsa_queries *q)
{
  q->minlength = ~0ull;
  q->maxlength = 0;
  for (uint64_t v : q->hlength)
  {
    q->minlength = std::min(q->minlength, v);
    q->maxlength = std::max(q->maxlength, v);
  }
  if (q->hlength.empty())
  {
    q->minlength = 0;
  }
  q->uniform = !q->hlength.empty() && q->minlength == q->maxlength;
}

} // namespace

// a HIP call inside a constructor of a query batch: on failure the half-built
// object is released and the out-parameter cleared before the error returns
#define VSA_HIPQ(obj, out, call)                                              \
  do                                                                          \
  {                                                                           \
    hipError_t e_ = (call);                                                   \
    if (e_ != hipSuccess)                                                     \
    {                                                                         \
      VSA_ERROR("%s:%d: %s failed: %s", __FILE__, __LINE__, #call,            \
                hipGetErrorString(e_));                                       \
      vsa_queries_free(obj);                                                  \
      *(out) = nullptr;                                                       \
      return -100;                                                            \
    }                                                                         \
  } while (0)

extern "C" int vsa_queries_from_host(const uint8_t *symbols,
                                     uint64_t nsymbols, const uint64_t *start,
                                     const uint64_t *length, uint64_t nq,
                                     int device, vsa_queries **queries)
{
  if (queries == nullptr || (nq > 0 && (start == nullptr ||
                                        length == nullptr)) ||
      (nsymbols > 0 && symbols == nullptr))
  {
    VSA_ERROR("vsa_queries_from_host: NULL argument");
    return -1;
  }
  *queries = nullptr;
  for (uint64_t i = 0; i < nq; i++)
  {
    if (start[i] > nsymbols || length[i] > nsymbols - start[i])
    {
      VSA_ERROR("query %lu [%lu, +%lu) lies outside the symbol buffer of "
                "%lu symbols",
                (unsigned long) i, (unsigned long) start[i],
                (unsigned long) length[i], (unsigned long) nsymbols);
      return -2;
    }
  }
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_queries *q = new vsa_queries;
  q->device = device;
  q->nq = nq;
  q->nsymbols = nsymbols;
  q->seqoffset = 0;
  q->symbols = nullptr;
  q->start = q->length = nullptr;
  q->hlength.assign(length, length + nq);
  summarise_lengths(q);
  q->dense = q->uniform;
  for (uint64_t i = 0; q->dense && i < nq; i++)
  {
    q->dense = start[i] == i * length[0];
  }
  *queries = q;
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->symbols, nsymbols + VSA_QUERY_BACKPAD));
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->start, (nq + 1) * 8));
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->length, (nq + 1) * 8));
  if (nsymbols > 0)
  {
    VSA_HIPQ(q, queries, hipMemcpy(q->symbols, symbols, nsymbols, hipMemcpyHostToDevice));
  }
  VSA_HIPQ(q, queries, hipMemset(q->symbols + nsymbols, 0xFF, VSA_QUERY_BACKPAD));
  if (nq > 0)
  {
    VSA_HIPQ(q, queries, hipMemcpy(q->start, start, nq * 8, hipMemcpyHostToDevice));
    VSA_HIPQ(q, queries, hipMemcpy(q->length, length, nq * 8, hipMemcpyHostToDevice));
  }
  return 0;
}

__global__ void k_uniform_starts(uint64_t *start, uint64_t *length,
                                 uint64_t nq, uint64_t m)
{
  const uint64_t i = vsa_bid() * blockDim.x + threadIdx.x;
  if (i < nq)
  {
    start[i] = i * m;
    length[i] = m;
  }
}

extern "C" int vsa_queries_from_device(const void *device_symbols,
                                       uint64_t nq, uint32_t m, int device,
                                       vsa_queries **queries)
{
  if (queries == nullptr || (nq > 0 && device_symbols == nullptr))
  {
    VSA_ERROR("vsa_queries_from_device: NULL argument");
    return -1;
  }
  *queries = nullptr;
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_queries *q = new vsa_queries;
  q->device = device;
  q->nq = nq;
  q->nsymbols = nq * (uint64_t) m;
  q->seqoffset = 0;
  q->symbols = nullptr;
  q->start = q->length = nullptr;
  q->hlength.assign(nq, (uint64_t) m);
  summarise_lengths(q);
  q->dense = q->uniform;
  *queries = q;
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->symbols,
                    q->nsymbols + VSA_QUERY_BACKPAD));
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->start, (nq + 1) * 8));
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->length, (nq + 1) * 8));
  if (q->nsymbols > 0)
  {
    VSA_HIPQ(q, queries, hipMemcpy(q->symbols, device_symbols, q->nsymbols,
                      hipMemcpyDeviceToDevice));
  }
  VSA_HIPQ(q, queries, hipMemset(q->symbols + q->nsymbols, 0xFF, VSA_QUERY_BACKPAD));
  if (nq > 0)
  {
    k_uniform_starts<<<vsa_grid((nq + 255) / 256), 256>>>(
        q->start, q->length, nq, m);
    VSA_HIPQ(q, queries, hipGetLastError());
    VSA_HIPQ(q, queries, hipDeviceSynchronize());
  }
  return 0;
}

// copymultiseqRC, kurtz-basic/readmulti.c:93-125: every sequence reversed
// and complemented on its own, in place of the forward sequence
__global__ void __launch_bounds__(256)
k_reverse_complement(const uint8_t *__restrict__ in,
                     const uint64_t *__restrict__ start,
                     const uint64_t *__restrict__ length, uint64_t nq,
                     uint8_t *__restrict__ out, uint32_t *__restrict__ bad)
{
  // one wavefront per sequence, lanes stride over its symbols
  const uint64_t q = (vsa_bid() * 256 + threadIdx.x) >> 6;
  const uint32_t lane = threadIdx.x & 63;
  if (q >= nq)
  {
    return;
  }
  const uint64_t s = start[q], len = length[q];
  for (uint64_t i = lane; i < len; i += 64)
  {
    const uint8_t c = in[s + len - 1 - i];
    uint8_t r;
    if (c == VSA_WILDCARD)
    {
      r = (uint8_t) VSA_WILDCARD;
    } else if (c > 3)
    {
      r = c; // readmulti.c:45-49: "reverse complement of %lu undefined"
      atomicMax(bad, (uint32_t) c + 1);
    } else
    {
      r = (uint8_t) (3 - c);
    }
    out[s + i] = r;
  }
}

extern "C" int vsa_queries_reverse_complement(const vsa_queries *q,
                                              vsa_queries **rcqueries)
{
  if (q == nullptr || rcqueries == nullptr)
  {
    VSA_ERROR("vsa_queries_reverse_complement: NULL argument");
    return -1;
  }
  *rcqueries = nullptr;
  if (vsa_set_device(q->device) != 0)
  {
    return -100;
  }
  // (a packed batch: its bytes first; the reverse complement is a byte batch)
  if (q->rows != nullptr)
  {
    if (vsa_queries_bytes(q, nullptr) != 0)
    {
      return -100;
    }
    VSA_HIP(hipStreamSynchronize(nullptr));
  }
  vsa_queries *r = new vsa_queries;
  r->device = q->device;
  r->nq = q->nq;
  r->nsymbols = q->nsymbols;
  r->seqoffset = q->seqoffset;
  r->symbols = nullptr;
  r->start = r->length = nullptr;
  r->hlength = q->hlength;
  r->minlength = q->minlength;
  r->maxlength = q->maxlength;
  r->uniform = q->uniform;
  r->dense = q->dense;
  *rcqueries = r;
  uint32_t *dbad = nullptr, hbad = 0;
  VSA_HIPQ(r, rcqueries, vsa_hip_malloc((void **) &r->symbols, r->nsymbols + VSA_QUERY_BACKPAD));
  VSA_HIPQ(r, rcqueries, vsa_hip_malloc((void **) &r->start, (r->nq + 1) * 8));
  VSA_HIPQ(r, rcqueries, vsa_hip_malloc((void **) &r->length, (r->nq + 1) * 8));
  VSA_HIPQ(r, rcqueries, vsa_hip_malloc((void **) &dbad, 4));
  VSA_HIPQ(r, rcqueries, hipMemset(dbad, 0, 4));
  // separators between the sequences and the padding stay what they are
  VSA_HIPQ(r, rcqueries, hipMemcpy(r->symbols, q->symbols,
                    r->nsymbols + VSA_QUERY_BACKPAD,
                    hipMemcpyDeviceToDevice));
  VSA_HIPQ(r, rcqueries, hipMemcpy(r->start, q->start, (r->nq + 1) * 8,
                    hipMemcpyDeviceToDevice));
  VSA_HIPQ(r, rcqueries, hipMemcpy(r->length, q->length, (r->nq + 1) * 8,
                    hipMemcpyDeviceToDevice));
  if (r->nq > 0)
  {
    k_reverse_complement<<<vsa_grid((r->nq * 64 + 255) / 256), 256>>>(
        q->symbols, q->start, q->length, r->nq, r->symbols, dbad);
    VSA_HIPQ(r, rcqueries, hipGetLastError());
  }
  VSA_HIPQ(r, rcqueries, hipMemcpy(&hbad, dbad, 4, hipMemcpyDeviceToHost));
  (void) hipFree(dbad);
  if (hbad != 0)
  {
    VSA_ERROR("reverse complement of %lu undefined",
              (unsigned long) (hbad - 1));
    vsa_queries_free(r);
    *rcqueries = nullptr;
    return -2;
  }
  return 0;
}

extern "C" int vsa_queries_set_offset(vsa_queries *q, uint64_t offset)
{
  if (q == nullptr)
  {
    VSA_ERROR("vsa_queries_set_offset: NULL argument");
    return -1;
  }
  q->seqoffset = offset;
  return 0;
}

// ---- reads at two bits per symbol -------------------------------------------

extern "C" uint32_t vsa_packed_words(uint32_t querylength)
{
  return vsa_rowwords(querylength);
}

// (vsa_pack_reads, vsa_pack_reads_mt: pack_reads.c -- host code, no GPU)

// One work-item per four symbols of a read (one byte of its row): 32-bit
// stores where the reads' length is a multiple of four, which puts the 256
// work-items of a workgroup on 1 KB of consecutive output.  (The first form --
// a wavefront per read, a byte per lane and store -- took 2.0 ms for 10 M
// reads of 100 symbols; this one streams.)
__global__ void __launch_bounds__(256)
k_unpack_rows(const uint64_t *__restrict__ rows, uint32_t W,
              const uint8_t *__restrict__ side, uint64_t nside, uint64_t nq,
              uint32_t m, uint32_t quads, uint8_t *__restrict__ out)
{
  const uint64_t g = vsa_bid() * 256 + threadIdx.x;
  const uint64_t q = g / quads;
  const uint32_t j0 = (uint32_t) (g - q * quads) * 4;
  if (q >= nq)
  {
    return;
  }
  const uint64_t *row = rows + q * W;
  const bool flagged = (row[W - 1] & 0xFFu) != 0 && nside > 0;
  uint8_t c[4];
  if (flagged)
  {
    const uint64_t k0 = row[0] >> 8, k = k0 < nside ? k0 : nside - 1;
    const uint8_t *sym = side + k * (uint64_t) m;
#pragma unroll
    for (int t = 0; t < 4; t++)
    {
      c[t] = j0 + t < m ? sym[j0 + t] : 0;
    }
  } else
  {
    // symbols j0 .. j0 + 3 are one byte of word j0 / 32, the first on top
    const uint32_t b = (uint32_t) (row[j0 >> 5] >> (56 - 2 * (j0 & 31u))) & 0xFFu;
    c[0] = (uint8_t) (b >> 6);
    c[1] = (uint8_t) ((b >> 4) & 3u);
    c[2] = (uint8_t) ((b >> 2) & 3u);
    c[3] = (uint8_t) (b & 3u);
  }
  uint8_t *dst = out + q * m + j0;
  if ((m & 3u) == 0)
  {
    *reinterpret_cast<uint32_t *>(dst) = (uint32_t) c[0] | ((uint32_t) c[1] << 8) |
                                         ((uint32_t) c[2] << 16) |
                                         ((uint32_t) c[3] << 24);
  } else
  {
#pragma unroll
    for (int t = 0; t < 4; t++)
    {
      if (j0 + t < m)
      {
        dst[t] = c[t];
      }
    }
  }
}

// bytes (and the start / length arrays) of a packed batch, on the device, for
// the kernels that read bytes; made once per batch
int vsa_queries_bytes(const vsa_queries *cq, hipStream_t stream)
{
  vsa_queries *q = const_cast<vsa_queries *>(cq);
  if (q->rows == nullptr || (q->symbols != nullptr && q->bytesvalid))
  {
    return 0;
  }
  const uint32_t m = (uint32_t) q->maxlength;
  if (q->symbols == nullptr)
  {
    // (a pipeline slot: room for its largest batch, kept from batch to batch)
    const uint64_t room = std::max(q->nsymbols, q->bytescapacity);
    VSA_HIP(vsa_hip_malloc((void **) &q->symbols, room + VSA_QUERY_BACKPAD));
  }
  VSA_HIP(hipMemsetAsync(q->symbols + q->nsymbols, 0xFF, VSA_QUERY_BACKPAD,
                         stream));
  if (q->start == nullptr)
  {
    const uint64_t most = std::max<uint64_t>(q->nq, q->bytescapacity / m);
    VSA_HIP(vsa_hip_malloc((void **) &q->start, (most + 1) * 8));
    VSA_HIP(vsa_hip_malloc((void **) &q->length, (most + 1) * 8));
  }
  if (q->nq > 0)
  {
    const uint32_t quads = (m + 3) / 4;
    k_unpack_rows<<<vsa_grid((q->nq * quads + 255) / 256), 256, 0, stream>>>(
        q->rows, q->roww, q->side, q->nside, q->nq, m, quads, q->symbols);
    VSA_HIP(hipGetLastError());
    k_uniform_starts<<<vsa_grid((q->nq + 255) / 256), 256, 0, stream>>>(
        q->start, q->length, q->nq, m);
    VSA_HIP(hipGetLastError());
  }
  q->bytesvalid = true;
  return 0;
}

extern "C" int vsa_queries_from_host_packed(const uint64_t *rows,
                                            uint64_t numofqueries,
                                            uint32_t querylength,
                                            const uint8_t *special,
                                            uint64_t numofspecial, int device,
                                            vsa_queries **queries)
{
  if (queries == nullptr || querylength == 0 ||
      (numofqueries > 0 && rows == nullptr) ||
      (numofspecial > 0 && special == nullptr))
  {
    VSA_ERROR("vsa_queries_from_host_packed: bad argument");
    return -1;
  }
  *queries = nullptr;
  const uint32_t m = querylength, W = vsa_rowwords(m);
  // a flagged row must name a read of the side list: checked here, on the
  // host, because the kernels follow the index without asking
  for (uint64_t i = 0; i < numofqueries; i++)
  {
    const uint64_t *row = rows + i * W;
    if ((row[W - 1] & 0xFFu) != 0 && (row[0] >> 8) >= numofspecial)
    {
      VSA_ERROR("vsa_queries_from_host_packed: read %lu names entry %lu of a "
                "side list of %lu reads", (unsigned long) i,
                (unsigned long) (row[0] >> 8), (unsigned long) numofspecial);
      return -2;
    }
  }
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  vsa_queries *q = new vsa_queries;
  q->device = device;
  q->nq = numofqueries;
  q->nsymbols = numofqueries * (uint64_t) m;
  q->seqoffset = 0;
  q->symbols = nullptr;
  q->start = q->length = nullptr;
  q->hlength.assign(1, (uint64_t) m); // uniform batches never look at it
  q->minlength = q->maxlength = numofqueries > 0 ? m : 0;
  q->uniform = q->dense = numofqueries > 0;
  q->roww = W;
  q->nside = numofspecial;
  *queries = q;
  // (32 bytes of slack: the kernels read a row as two 16-byte loads)
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->rows,
                                      numofqueries * W * 8 + 64));
  VSA_HIPQ(q, queries, vsa_hip_malloc((void **) &q->side,
                                      numofspecial * (uint64_t) m + 64));
  if (numofqueries > 0)
  {
    VSA_HIPQ(q, queries, hipMemcpy(q->rows, rows, numofqueries * W * 8,
                                   hipMemcpyHostToDevice));
  }
  VSA_HIPQ(q, queries, hipMemset(q->rows + numofqueries * W, 0, 64));
  if (numofspecial > 0)
  {
    VSA_HIPQ(q, queries, hipMemcpy(q->side, special,
                                   numofspecial * (uint64_t) m,
                                   hipMemcpyHostToDevice));
  }
  return 0;
}

extern "C" int vsa_queries_getinfo(const vsa_queries *q,
                                   vsa_queries_info *info)
{
  if (q == nullptr || info == nullptr)
  {
    VSA_ERROR("vsa_queries_getinfo: NULL argument");
    return -1;
  }
  info->numofqueries = q->nq;
  info->numofsymbols = q->nsymbols;
  info->minlength = q->minlength;
  info->maxlength = q->maxlength;
  info->offset = q->seqoffset;
  info->device = q->device;
  return 0;
}

// ---------------------------------------------------------------------------
// results
// ---------------------------------------------------------------------------

extern "C" uint64_t vsa_result_count(const vsa_result *r)
{
  return r == nullptr ? 0 : r->count;
}

extern "C" int vsa_result_getstats(const vsa_result *r, vsa_stats *stats)
{
  if (r == nullptr || stats == nullptr)
  {
    VSA_ERROR("vsa_result_getstats: NULL argument");
    return -1;
  }
  *stats = r->stats;
  return 0;
}

extern "C" int vsa_result_fetch(const vsa_result *r, vsa_match *matches,
                                uint64_t capacity)
{
  if (r == nullptr || (matches == nullptr && capacity > 0))
  {
    VSA_ERROR("vsa_result_fetch: NULL argument");
    return -1;
  }
  const uint64_t m = std::min(capacity, r->count);
  if (m == 0)
  {
    return 0;
  }
  if (vsa_set_device(r->device) != 0)
  {
    return -100;
  }
  if (r->packbits != 0)
  {
    // pairs: as records through a temporary
    void *tmp = nullptr;
    vsa_dev_set_stream(nullptr); // vsa_unpack_result: default stream
    if (vsa_dev_alloc(&tmp, m * sizeof(vsa_match)) != 0)
    {
      return -100;
    }
    int rc = vsa_unpack_result(r, m, (vsa_match *) tmp);
    if (rc == 0 && hipMemcpy(matches, tmp, m * sizeof(vsa_match),
                             hipMemcpyDeviceToHost) != hipSuccess)
    {
      rc = -100;
    }
    vsa_dev_free(tmp);
    return rc;
  }
  VSA_HIP(hipMemcpy(matches, r->matches, m * sizeof(vsa_match),
                    hipMemcpyDeviceToHost));
  return 0;
}

extern "C" int vsa_result_copy_device(const vsa_result *r,
                                      void *device_matches,
                                      uint64_t capacity)
{
  if (r == nullptr || (device_matches == nullptr && capacity > 0))
  {
    VSA_ERROR("vsa_result_copy_device: NULL argument");
    return -1;
  }
  const uint64_t m = std::min(capacity, r->count);
  if (m == 0)
  {
    return 0;
  }
  if (vsa_set_device(r->device) != 0)
  {
    return -100;
  }
  if (r->packbits != 0)
  {
    return vsa_unpack_result(r, m, (vsa_match *) device_matches);
  }
  VSA_HIP(hipMemcpy(device_matches, r->matches, m * sizeof(vsa_match),
                    hipMemcpyDeviceToDevice));
  return 0;
}

extern "C" const void *vsa_result_device_matches(const vsa_result *r)
{
  // a packed result has no records on the device
  return (r == nullptr || r->packbits != 0) ? nullptr : r->matches;
}

extern "C" uint32_t vsa_result_packbits(const vsa_result *r)
{
  return r == nullptr ? 0 : r->packbits;
}

extern "C" void vsa_result_free(vsa_result *r)
{
  if (r == nullptr)
  {
    return;
  }
  (void) hipSetDevice(r->device);
  vsa_dev_free(r->matches);
  vsa_dev_free(r->packvals);
  delete r;
}

// ---------------------------------------------------------------------------
// streaming-read probe: the measured denominator next to the 8 TB/s spec
// ---------------------------------------------------------------------------

__global__ void __launch_bounds__(256)
k_stream_read(const uint4 *__restrict__ p, uint64_t n16,
              unsigned long long *sink)
{
  uint64_t i = vsa_bid() * blockDim.x + threadIdx.x;
  const uint64_t stride = vsa_nblocks() * blockDim.x;
  uint32_t acc = 0;
  for (; i < n16; i += stride)
  {
    const uint4 v = p[i];
    acc ^= v.x ^ v.y ^ v.z ^ v.w;
  }
  if (acc == 0x12345678u) // practically never: keeps the loads alive
  {
    atomicAdd(sink, 1ull);
  }
}

// independent random 8-byte reads: INFLIGHT loads per work-item are issued
// before any of them is used, the addresses come from a counter-based
// generator (no dependence between them) -- the rate the memory system
// sustains for 64-byte sectors that miss every cache
template <int INFLIGHT>
__global__ void __launch_bounds__(256)
k_random_read(const uint64_t *__restrict__ buf, uint64_t nwords,
              uint64_t perthread, unsigned long long *__restrict__ sink)
{
  const uint64_t t = vsa_bid() * 256 + threadIdx.x;
  // (x = (t * perthread + i + 1) * golden: every (thread, step) has a value
  // of its own.  Until round 3 the sequence was (t + i) * golden -- thread t's
  // step i read what thread t + 1 had read one step earlier, 92 % of the
  // "random" reads hit L2, and the 76 G reads/s this probe reported for two
  // rounds was not the rate of HBM.)
  uint64_t x = t * perthread * 0x9E3779B97F4A7C15ull + 1, acc = 0;
  for (uint64_t i = 0; i < perthread; i += INFLIGHT)
  {
    uint64_t v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++)
    {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      v[k] = buf[z % nwords];
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++)
    {
      acc += v[k];
    }
  }
  if (acc == 0x1234567ull)
  {
    atomicAdd(sink, 1ull);
  }
}

extern "C" int vsa_measure_random_read(uint64_t bytes, int inflight,
                                       int device, double *greads)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  void *buf = nullptr;
  unsigned long long *sink = nullptr;
  const uint64_t nwords = bytes / 8, threads = 256ull * 256 * 32,
                 perthread = 64;
  VSA_HIP(vsa_hip_malloc(&buf, nwords * 8 + 16));
  VSA_HIP(vsa_hip_malloc((void **) &sink, 8));
  VSA_HIP(hipMemset(buf, 1, nwords * 8));
  VSA_HIP(hipMemset(sink, 0, 8));
  hipEvent_t a, b;
  VSA_HIP(hipEventCreate(&a));
  VSA_HIP(hipEventCreate(&b));
  const int reps = 3;
  for (int r = 0; r <= reps; r++)
  {
    if (r == 1)
    {
      VSA_HIP(hipEventRecord(a, 0));
    }
    if (inflight >= 8)
    {
      k_random_read<8><<<(unsigned int) (threads / 256), 256>>>(
          (const uint64_t *) buf, nwords, perthread, sink);
    } else if (inflight >= 4)
    {
      k_random_read<4><<<(unsigned int) (threads / 256), 256>>>(
          (const uint64_t *) buf, nwords, perthread, sink);
    } else
    {
      k_random_read<1><<<(unsigned int) (threads / 256), 256>>>(
          (const uint64_t *) buf, nwords, perthread, sink);
    }
  }
  VSA_HIP(hipEventRecord(b, 0));
  VSA_HIP(hipEventSynchronize(b));
  float ms = 0;
  VSA_HIP(hipEventElapsedTime(&ms, a, b));
  *greads = (double) threads * perthread * reps / ((double) ms * 1e-3) / 1e9;
  (void) hipEventDestroy(a);
  (void) hipEventDestroy(b);
  (void) hipFree(buf);
  (void) hipFree(sink);
  return 0;
}

// the same probe on a table of a live index, where the driver happened to
// place it: 0 slot16 (16-byte reads of whole slots), 1 esa8 (8-byte), 2 tis2,
// 3 suf
template <typename T, int INFLIGHT>
__global__ void __launch_bounds__(256)
k_table_read(const T *__restrict__ buf, uint64_t nitems, uint64_t perthread,
             unsigned long long *__restrict__ sink)
{
  const uint64_t t = vsa_bid() * 256 + threadIdx.x;
  uint64_t x = t * perthread * 0x9E3779B97F4A7C15ull + 1, acc = 0;
  for (uint64_t i = 0; i < perthread; i += INFLIGHT)
  {
    T v[INFLIGHT];
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++)
    {
      x += 0x9E3779B97F4A7C15ull;
      uint64_t z = x;
      z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
      z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
      z ^= z >> 31;
      v[k] = buf[z % nitems];
    }
#pragma unroll
    for (int k = 0; k < INFLIGHT; k++)
    {
      if constexpr (sizeof(T) == 16)
      {
        acc += (uint64_t) v[k].x + v[k].y + v[k].z + v[k].w;
      } else
      {
        acc += (uint64_t) v[k];
      }
    }
  }
  if (acc == 0x1234567ull)
  {
    atomicAdd(sink, 1ull);
  }
}

extern "C" int vsa_measure_table_read(const vsa_index *ix, int table,
                                      int inflight, double *greads)
{
  if (ix == nullptr || greads == nullptr || vsa_set_device(ix->device) != 0)
  {
    return -100;
  }
  const void *buf = nullptr;
  uint64_t nitems = 0;
  int width = 8;
  if (table == 0 && ix->slot16 != nullptr)
  {
    buf = ix->slot16;
    nitems = (1ull << (2 * ix->D)) * (ix->slotwords / 2);
    width = 16;
  } else if (table == 1 && ix->esa8 != nullptr)
  {
    buf = ix->esa8;
    nitems = ix->n + 1;
  } else if (table == 2 && ix->tis2 != nullptr)
  {
    buf = ix->tis2;
    nitems = (ix->n / 4) / 8;
  } else if (table == 3)
  {
    buf = ix->suf;
    nitems = (ix->n + 1) * ix->isize / 8;
  } else if (table == 4 && ix->slot16 != nullptr) // slot16, 8-byte reads
  {
    buf = ix->slot16;
    nitems = (1ull << (2 * ix->D)) * ix->slotwords;
  } else if (table == 5 && ix->esa8 != nullptr) // esa8, 16-byte reads
  {
    buf = ix->esa8;
    nitems = (ix->n + 1) / 2;
    width = 16;
  } else
  {
    VSA_ERROR("vsa_measure_table_read: the index has no table %d", table);
    return -2;
  }
  unsigned long long *sink = nullptr;
  VSA_HIP(vsa_hip_malloc((void **) &sink, 8));
  VSA_HIP(hipMemset(sink, 0, 8));
  const uint64_t threads = 256ull * 256 * 32, perthread = 64;
  hipEvent_t a, b;
  VSA_HIP(hipEventCreate(&a));
  VSA_HIP(hipEventCreate(&b));
  const int reps = 3;
  for (int r = 0; r <= reps; r++)
  {
    if (r == 1)
    {
      VSA_HIP(hipEventRecord(a, 0));
    }
    const unsigned int grid = (unsigned int) (threads / 256);
    if (width == 16 && inflight >= 4)
    {
      k_table_read<uint4, 4><<<grid, 256>>>((const uint4 *) buf, nitems,
                                             perthread, sink);
    } else if (width == 16)
    {
      k_table_read<uint4, 1><<<grid, 256>>>((const uint4 *) buf, nitems,
                                             perthread, sink);
    } else if (inflight >= 4)
    {
      k_table_read<uint64_t, 4><<<grid, 256>>>((const uint64_t *) buf, nitems,
                                                perthread, sink);
    } else
    {
      k_table_read<uint64_t, 1><<<grid, 256>>>((const uint64_t *) buf, nitems,
                                                perthread, sink);
    }
  }
  VSA_HIP(hipEventRecord(b, 0));
  VSA_HIP(hipEventSynchronize(b));
  float ms = 0;
  VSA_HIP(hipEventElapsedTime(&ms, a, b));
  *greads = (double) threads * perthread * reps / ((double) ms * 1e-3) / 1e9;
  (void) hipEventDestroy(a);
  (void) hipEventDestroy(b);
  (void) hipFree(sink);
  return 0;
}

extern "C" int vsa_measure_stream_read(uint64_t bytes, int device,
                                       double *gbps)
{
  if (vsa_set_device(device) != 0)
  {
    return -100;
  }
  void *buf = nullptr;
  unsigned long long *sink = nullptr;
  const uint64_t n16 = bytes / 16;
  VSA_HIP(vsa_hip_malloc(&buf, n16 * 16 + 16));
  VSA_HIP(vsa_hip_malloc((void **) &sink, 8));
  VSA_HIP(hipMemset(buf, 1, n16 * 16));
  VSA_HIP(hipMemset(sink, 0, 8));
  hipEvent_t a, b;
  VSA_HIP(hipEventCreate(&a));
  VSA_HIP(hipEventCreate(&b));
  const int reps = 5;
  k_stream_read<<<256 * 8, 256>>>((const uint4 *) buf, n16, sink); // warm up
  VSA_HIP(hipEventRecord(a, 0));
  for (int r = 0; r < reps; r++)
  {
    k_stream_read<<<256 * 8, 256>>>((const uint4 *) buf, n16, sink);
  }
  VSA_HIP(hipEventRecord(b, 0));
  VSA_HIP(hipEventSynchronize(b));
  float ms = 0;
  VSA_HIP(hipEventElapsedTime(&ms, a, b));
  *gbps = (double) (n16 * 16) * reps / ((double) ms * 1e-3) / 1e9;
  (void) hipEventDestroy(a);
  (void) hipEventDestroy(b);
  (void) hipFree(buf);
  (void) hipFree(sink);
  return 0;
}
