// plan.hip -- host orchestration and the C ABI of libfregrid_hip.so (see include/fregrid_hip.h).
//
// A plan owns, in HBM, everything one destination tile needs: per-cell records of the
// source tiles and of the destination tile, the exchange cells in canonical order and the
// destination-row CSR layout for the sweep.  Device memory comes from a small caching pool
// so that repeated plan creation (one per destination tile / per benchmark step) does not
// pay hipMalloc/hipFree.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <map>
#include <mutex>
#include <string>
#include <type_traits>
#include <vector>
#include "fregrid_hip.h"
#include "xgrid_device.h"

// ----------------------------------------------------------------------------- errors
static thread_local std::string g_err;
static int fail(int code, const char *fmt, ...)
{
  char buf[512];
  va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
  g_err = buf;
  return code;
}
#define HIPCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) \
  return fail(FG_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

extern "C" const char *fg_last_error(void) { return g_err.c_str(); }
void fg_set_last_error(const char *msg) { g_err = msg ? msg : ""; }      // for the other translation units (sweep.hip)

extern "C" int fg_device_count(void)
{
  int n = 0;
  hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) { fail(FG_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e)); return FG_ERR_HIP; }
  return n;
}

// ----------------------------------------------------------------------------- memory pool
namespace {
struct Pool {
  std::mutex mu;
  std::map<std::pair<int, size_t>, std::vector<void *>> free_;   // (device, class bytes) -> blocks
  std::map<void *, std::pair<int, size_t>> live_;
  static size_t klass(size_t n)
  {
    size_t c = 256;
    while (c < n) c += (c < (1u << 20) ? c : (c >> 2));           // x2 up to 1 MiB, then x1.25
    return c;
  }
  // FREGRID_HIP_POISON=1 (tests): every block handed out is filled with 0x7f bytes first, so that a kernel reading something it
  // or its predecessors never wrote meets wild indices / NaN-like doubles instead of a lucky zero or a stale valid value
  static void *poison(void *p, size_t c)
  {
    static const bool on = getenv("FREGRID_HIP_POISON") && atoi(getenv("FREGRID_HIP_POISON")) != 0;
    if (on && p) { (void)hipMemset(p, 0x7f, c); (void)hipDeviceSynchronize(); }
    return p;
  }
  void *get(int dev, size_t n) { return poison(get_raw(dev, n), klass(n ? n : 1)); }
  void *get_raw(int dev, size_t n)
  {
    if (n == 0) n = 1;
    size_t c = klass(n);
    {
      std::lock_guard<std::mutex> lk(mu);
      auto it = free_.find({dev, c});
      if (it != free_.end() && !it->second.empty()) {
        void *p = it->second.back(); it->second.pop_back();
        live_[p] = {dev, c};
        return p;
      }
    }
    void *p = nullptr;
    if (hipMalloc(&p, c) != hipSuccess) {
      release_all();
      if (hipMalloc(&p, c) != hipSuccess) return nullptr;
    }
    std::lock_guard<std::mutex> lk(mu);
    live_[p] = {dev, c};
    return p;
  }
  void put(void *p)
  {
    if (!p) return;
    std::lock_guard<std::mutex> lk(mu);
    auto it = live_.find(p);
    if (it == live_.end()) return;
    free_[it->second].push_back(p);
    live_.erase(it);
  }
  void release_all()
  {
    std::lock_guard<std::mutex> lk(mu);
    for (auto &kv : free_) for (void *p : kv.second) (void)hipFree(p);
    free_.clear();
  }
};
Pool g_pool;
}  // namespace

extern "C" void fg_pool_release(void) { g_pool.release_all(); }

// Device memory for C callers that do not include the HIP headers (the fregrid replacement objects, integration/): blocks come
// from the plans' caching pool; copies are synchronous.
extern "C" void *fg_dev_alloc(size_t bytes, int device)
{
  if (hipSetDevice(device) != hipSuccess) { fail(FG_ERR_HIP, "fg_dev_alloc: no such HIP device %d", device); return nullptr; }
  void *p = g_pool.get(device, bytes);
  if (!p) fail(FG_ERR_HIP, "fg_dev_alloc: out of device memory (%zu bytes)", bytes);
  return p;
}
extern "C" void fg_dev_free(void *p) { g_pool.put(p); }
extern "C" int fg_dev_upload(void *dst_dev, const void *src_host, size_t bytes)
{
  if (bytes && (!dst_dev || !src_host)) return fail(FG_ERR_ARG, "fg_dev_upload: null pointer");
  if (bytes) HIPCHK(hipMemcpy(dst_dev, src_host, bytes, hipMemcpyHostToDevice));
  return 0;
}
extern "C" int fg_dev_download(void *dst_host, const void *src_dev, size_t bytes)
{
  if (bytes && (!dst_host || !src_dev)) return fail(FG_ERR_ARG, "fg_dev_download: null pointer");
  if (bytes) HIPCHK(hipMemcpy(dst_host, src_dev, bytes, hipMemcpyDeviceToHost));
  return 0;
}
// dst_dev[i] = src_dev[idx_dev[i]] / dst_dev[idx_dev[i]] = src_dev[i], i < n: moving a short list of source cells in and out of the
// [3][ncells] sum arrays (integration/conserve_interp_hip.c hands the shared cells' running sums from rank to rank with them)
__global__ __launch_bounds__(256) void k_gather_f64(long n, double *dst, const double *src, const int *idx, int scatter)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (scatter) dst[idx[i]] = src[i]; else dst[i] = src[idx[i]];
}
static int dev_gather(double *dst_dev, const double *src_dev, const int *idx_dev, long n, int scatter)
{
  if (n < 0 || (n > 0 && (!dst_dev || !src_dev || !idx_dev))) return fail(FG_ERR_ARG, "fg_dev_gather_f64 / fg_dev_scatter_f64: bad argument");
  if (n == 0) return 0;
  k_gather_f64<<<(unsigned)((n + 255) / 256), 256>>>(n, dst_dev, src_dev, idx_dev, scatter);
  HIPCHK(hipGetLastError());
  HIPCHK(hipDeviceSynchronize());
  return 0;
}
extern "C" int fg_dev_gather_f64(double *dst_dev, const double *src_dev, const int *idx_dev, long n) { return dev_gather(dst_dev, src_dev, idx_dev, n, 0); }
extern "C" int fg_dev_scatter_f64(double *dst_dev, const double *src_dev, const int *idx_dev, long n) { return dev_gather(dst_dev, src_dev, idx_dev, n, 1); }

// ----------------------------------------------------------------------------- host <-> device transfers of large arrays
// The host-pointer entry points (B1: create_xgrid_*, B2: fg_plan_create / fg_plan_get_xgrid) move corner arrays up and 166 MB of
// exchange cells (C384 -> 0.25 deg) down, to and from the caller's PAGEABLE memory (the reference's malloc'ed Interp_config
// arrays).  A plain hipMemcpy does that at ~14 GB/s -- one thread staging through a bounce buffer and taking the page faults of
// freshly allocated destination arrays -- which was 12 of the 14 ms of the whole call.  XferPool splits a batch of copies into
// 2 MB chunks taken by a few persistent worker threads; each has a stream and two page-locked slots, so that the DMA of one chunk
// runs beside the host memcpy (and page faults) of another and the workers add up to the link's rate.
#include <atomic>
#include <condition_variable>
#include <thread>
namespace {
struct XferJob { void *host; void *dev; size_t bytes; bool d2h; };
class XferPool {
  const size_t CHUNK = [] { const char *e = getenv("FREGRID_HIP_XFER_CHUNK_MB"); const int m = e ? atoi(e) : 0; return (size_t)(m > 0 && m <= 64 ? m : 2) << 20; }();   // 2 MB x 8 workers measured best (scripts/pcie_time.py)
  std::mutex batch_mu;                        // one batch at a time
  std::mutex mu; std::condition_variable cv_work, cv_done;
  std::vector<std::thread> threads;
  const std::vector<XferJob> *jobs = nullptr;
  std::vector<std::pair<int, size_t>> chunks; // (job, offset)
  std::atomic<size_t> next{0};
  std::atomic<int> failed{0};
  int device = 0, active = 0;
  unsigned long long gen = 0;
  struct Slot { void *pin = nullptr; hipEvent_t ev = nullptr; int pending = 0; void *host = nullptr; size_t bytes = 0; };   // pending: 1 h2d, 2 d2h
  struct Worker { int dev = -1; hipStream_t st = nullptr; Slot slot[2]; };

  bool retire(Slot &sl)
  {
    if (!sl.pending) return true;
    const bool ok = hipEventSynchronize(sl.ev) == hipSuccess;
    if (ok && sl.pending == 2) memcpy(sl.host, sl.pin, sl.bytes);
    sl.pending = 0;
    return ok;
  }
  bool prepare(Worker &w)
  {
    if (w.dev == device && w.st) return true;
    if (hipSetDevice(device) != hipSuccess) return false;
    if (w.st) {                                            // another device than last time: start over
      for (Slot &sl : w.slot) { if (sl.pin) (void)hipHostFree(sl.pin); if (sl.ev) (void)hipEventDestroy(sl.ev); sl = Slot{}; }
      (void)hipStreamDestroy(w.st); w.st = nullptr;
    }
    if (hipStreamCreateWithFlags(&w.st, hipStreamNonBlocking) != hipSuccess) return false;
    for (Slot &sl : w.slot)
      if (hipHostMalloc(&sl.pin, CHUNK, hipHostMallocDefault) != hipSuccess || hipEventCreateWithFlags(&sl.ev, hipEventDisableTiming) != hipSuccess) return false;
    w.dev = device;
    return true;
  }
  void work(Worker &w)
  {
    bool ok = prepare(w);
    int s = 0;
    for (;;) {
      const size_t c = next.fetch_add(1);
      if (c >= chunks.size() || !ok) break;
      const XferJob &j = (*jobs)[chunks[c].first];
      const size_t off = chunks[c].second, n = std::min(CHUNK, j.bytes - off);
      Slot &sl = w.slot[s]; s ^= 1;
      ok = retire(sl);
      if (!ok) break;
      if (j.d2h) {
        ok = hipMemcpyAsync(sl.pin, (const char *)j.dev + off, n, hipMemcpyDeviceToHost, w.st) == hipSuccess && hipEventRecord(sl.ev, w.st) == hipSuccess;
        sl.pending = 2; sl.host = (char *)j.host + off; sl.bytes = n;
      } else {
        memcpy(sl.pin, (const char *)j.host + off, n);
        ok = hipMemcpyAsync((char *)j.dev + off, sl.pin, n, hipMemcpyHostToDevice, w.st) == hipSuccess && hipEventRecord(sl.ev, w.st) == hipSuccess;
        sl.pending = 1;
      }
    }
    for (Slot &sl : w.slot) ok = retire(sl) && ok;
    if (!ok) { (void)hipGetLastError(); failed.store(1); }
  }
  void loop()
  {
    Worker w;
    unsigned long long seen = 0;
    for (;;) {
      {
        std::unique_lock<std::mutex> lk(mu);
        cv_work.wait(lk, [&] { return gen != seen; });
        seen = gen;
      }
      work(w);
      std::lock_guard<std::mutex> lk(mu);
      if (--active == 0) cv_done.notify_all();
    }
  }

 public:
  // false: a copy failed (the caller reports it)
  bool run(int dev, const std::vector<XferJob> &batch)
  {
    size_t total = 0;
    for (const XferJob &j : batch) total += j.bytes;
    if (total == 0) return true;
    static const int nthreads = [] {
      const char *e = getenv("FREGRID_HIP_XFER_THREADS");
      int n = e ? atoi(e) : 0;
      if (n <= 0) { const unsigned hc = std::thread::hardware_concurrency(); n = hc >= 16 ? 8 : (hc >= 8 ? 4 : 2); }
      return std::min(n, 16);
    }();
    if (total < 2 * CHUNK) {                               // small: not worth waking anybody
      if (hipSetDevice(dev) != hipSuccess) return false;
      for (const XferJob &j : batch)
        if (j.bytes && hipMemcpy(j.d2h ? j.host : j.dev, j.d2h ? j.dev : j.host, j.bytes, j.d2h ? hipMemcpyDeviceToHost : hipMemcpyHostToDevice) != hipSuccess) return false;
      return true;
    }
    std::lock_guard<std::mutex> bl(batch_mu);
    {
      std::lock_guard<std::mutex> lk(mu);
      if (threads.empty())
        for (int t = 0; t < nthreads; t++) { threads.emplace_back([this] { loop(); }); threads.back().detach(); }
      jobs = &batch; chunks.clear();
      for (size_t k = 0; k < batch.size(); k++)
        for (size_t off = 0; off < batch[k].bytes; off += CHUNK) chunks.push_back({(int)k, off});
      next.store(0); failed.store(0); device = dev; active = (int)threads.size(); gen++;
    }
    cv_work.notify_all();
    std::unique_lock<std::mutex> lk(mu);
    cv_done.wait(lk, [&] { return active == 0; });
    jobs = nullptr;
    return failed.load() == 0;
  }
};
XferPool &xfer_pool() { static XferPool *p = new XferPool(); return *p; }     // (never destroyed: its threads outlive main's statics)
}  // namespace

// ----------------------------------------------------------------------------- phase timing
// Optional HIP-event timing of the phases of a search / sweep, recorded on the plan's own
// stream (bench.py reads these for the roofline object; torch.cuda.Event would only see
// PyTorch's current stream).
static int g_profiling = 0;
extern "C" void fg_set_profiling(int on) { g_profiling = on ? 1 : 0; }
enum { PH_CELL_STRUCT = 0, PH_BINS, PH_CANDIDATES, PH_CLIP_QUAD, PH_CLIP_GENERAL, PH_COMPACT, PH_ROWS,
       PH_SEARCH_TOTAL, PH_FINALIZE, PH_APPLY, PH_COUNT };
// Streams and events are cached per process: creating and destroying a stream costs ~0.1 ms each on this runtime, which
// was a quarter of a millisecond per plan (scripts/host_time.py).
struct HandleCache {
  std::mutex mu;
  std::map<int, std::vector<hipStream_t>> streams;
  std::map<int, std::vector<hipEvent_t>> events;        // per device, like the streams: an event belongs to its device
  static int current_device() { int d = 0; (void)hipGetDevice(&d); return d; }
  hipStream_t get_stream(int dev)
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto &v = streams[dev];
      if (!v.empty()) { hipStream_t s = v.back(); v.pop_back(); return s; }
    }
    hipStream_t s = nullptr;
    if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) return nullptr;
    return s;
  }
  void put_stream(int dev, hipStream_t s)
  {
    if (!s) return;
    std::lock_guard<std::mutex> lk(mu);
    streams[dev].push_back(s);
  }
  hipEvent_t get_event()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto &v = events[current_device()];
      if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreate(&e);
    return e;
  }
  void put_event(hipEvent_t e)
  {
    if (!e) return;
    std::lock_guard<std::mutex> lk(mu);
    events[current_device()].push_back(e);
  }
  // ordering-only events (stream A -> stream B dependencies of the chunked search)
  std::map<int, std::vector<hipEvent_t>> sync_events;
  hipEvent_t get_sync_event()
  {
    {
      std::lock_guard<std::mutex> lk(mu);
      auto &v = sync_events[current_device()];
      if (!v.empty()) { hipEvent_t e = v.back(); v.pop_back(); return e; }
    }
    hipEvent_t e = nullptr;
    (void)hipEventCreateWithFlags(&e, hipEventDisableTiming);
    return e;
  }
  void put_sync_event(hipEvent_t e)
  {
    if (!e) return;
    std::lock_guard<std::mutex> lk(mu);
    sync_events[current_device()].push_back(e);
  }
};
static HandleCache g_handles;

struct PhaseTimer {
  bool on = false;
  hipStream_t st = nullptr;
  std::vector<std::pair<int, std::pair<hipEvent_t, hipEvent_t>>> spans;
  hipEvent_t open_ev = nullptr; int open_ph = -1;
  void start(bool enable, hipStream_t s) { on = enable; st = s; }
  void begin(int ph)
  {
    if (!on) return;
    open_ev = g_handles.get_event(); (void)hipEventRecord(open_ev, st); open_ph = ph;
  }
  void end()
  {
    if (!on || open_ph < 0) return;
    hipEvent_t e = g_handles.get_event(); (void)hipEventRecord(e, st);
    spans.push_back({open_ph, {open_ev, e}}); open_ph = -1;
  }
  // call after the stream has been synchronised; accumulates into ms[]
  void collect(float *ms)
  {
    for (auto &sp : spans) {
      float t = 0; (void)hipEventElapsedTime(&t, sp.second.first, sp.second.second);
      ms[sp.first] += t;
      g_handles.put_event(sp.second.first); g_handles.put_event(sp.second.second);
    }
    spans.clear();
  }
};

// ----------------------------------------------------------------------------- plan
struct fg_plan {
  float phase_ms[PH_COUNT] = {0};
  PhaseTimer apply_pt;
  int apply_spans = 0;
  int order = 0, device = 0;
  bool great_circle = false;   // exchange cells from create_xgrid_great_circle semantics (order 1 only)
  hipStream_t stream = nullptr;
  hipStream_t stream_b = nullptr;   // second stream of the chunked search (always ours)
  bool own_stream = true;
  int ntiles = 0;
  std::vector<int> nx_in, ny_in, cell_off;
  std::vector<FgTile> tiles_host;   // staging for the async descriptor upload
  int nsrc = 0;                 // flattened source cells
  int nx_out = 0, ny_out = 0, ndst = 0;
  long f_stride = 0;            // elements in one level of the source field array
  bool searched = false, finalized = false, have_geom = false;
  bool fused = false;            // finalized by its own search (fg_set_search_finalize): centroids from the plan's own per-cell sums

  std::vector<void *> owned;    // every pool block of this plan
  FgTile *tiles_dev = nullptr;
  double *mask_dev = nullptr;
  FgCells S{}, D{};
  bool rect = false;            // searched on the rectilinear path: D holds areas only, the cells are in rect_tab (FgRect)
  FgRect rect_tab{};
  FgPolyList polys{};           // npoly > 0: the source "cells" are this list of polygons (fg_plan_create_polylist)
  // exchange cells
  long nx = 0;
  int *x_src = nullptr, *x_dst = nullptr;
  double *x_area = nullptr, *x_c1 = nullptr, *x_c2 = nullptr;
  int *xoff = nullptr;
  int *x_rowpos = nullptr;       // slot of every exchange cell in its destination row, taken while compacting (search scratch)
  int *perm = nullptr;           // exchange cells grouped by destination row (unsorted inside a row) until fg_plan_finalize
  bool rows_built = false;       // csr.row_ptr and perm come from the search
  bool dist_pending = false;     // order 2: x_c1/x_c2 still hold the centroid integrals after finalize (fg_plan_get_xgrid applies di/dj)
  double *sums = nullptr, *cen = nullptr;
  // sweep
  FgCsr csr{};
  int *src_idx_f = nullptr;
  double *row_sum = nullptr, *red_partial = nullptr, *red_result = nullptr;
  long row_sum_cap = 0;
  double *il_f = nullptr, *il_out = nullptr, *il_rs = nullptr;   // [cell][8] scratch (order 1 field, output, row sums)
  double *il_m = nullptr;                                        // order 2: merged records [source cell][3][8] (k_merge3)
  long stats[FG_NSTATS] = {0};
  // monotone limiter scratch (fg_plan_mono_*): limited values per CSR entry, per-source-cell bounds, error word
  double *mono_x = nullptr, *mono_b = nullptr;     // mono_b: [4][nsrc] = f_bar_max | f_bar_min | f_max | f_min
  int *xerr = nullptr;
  bool mono_open = false;

  template <typename T> T *alloc(size_t count)
  {
    void *p = g_pool.get(device, count * sizeof(T));
    if (p) owned.push_back(p);
    return (T *)p;
  }
  void release(void *p)
  {
    if (!p) return;
    for (size_t k = 0; k < owned.size(); k++) if (owned[k] == p) { owned.erase(owned.begin() + k); break; }
    g_pool.put(p);
  }
};

static bool alloc_cells(fg_plan *pl, FgCells *c, size_t n)
{
  c->lat_min = pl->alloc<double>(n); c->lat_max = pl->alloc<double>(n);
  c->lon_min = pl->alloc<double>(n); c->lon_max = pl->alloc<double>(n);
  c->lon_avg = pl->alloc<double>(n); c->area = pl->alloc<double>(n);
  c->nv = pl->alloc<int>(n);
  c->verts = pl->alloc<double>(n * 16);
  return c->lat_min && c->lat_max && c->lon_min && c->lon_max && c->lon_avg && c->area && c->nv && c->verts;
}

static int plan_base(int order, int ntiles_in, const int *nx_in, const int *ny_in, int nx_out, int ny_out,
                     int device, fg_plan **out)
{
  if (order != FG_CONSERVE_ORDER1 && order != FG_CONSERVE_ORDER2) return fail(FG_ERR_ARG, "order must be 1 or 2");
  if (ntiles_in < 1 || !nx_in || !ny_in || nx_out < 1 || ny_out < 1 || !out) return fail(FG_ERR_ARG, "bad grid sizes");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(FG_ERR_HIP, "no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  if (device < 0 || device >= ndev) return fail(FG_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));
  fg_plan *pl = new fg_plan();
  pl->order = order; pl->device = device; pl->ntiles = ntiles_in;
  long off = 0, foff = 0;
  for (int m = 0; m < ntiles_in; m++) {
    if (nx_in[m] < 1 || ny_in[m] < 1) { delete pl; return fail(FG_ERR_ARG, "bad source tile size"); }
    pl->nx_in.push_back(nx_in[m]); pl->ny_in.push_back(ny_in[m]); pl->cell_off.push_back((int)off);
    off += (long)nx_in[m] * ny_in[m];
    foff += (order == 2) ? (long)(nx_in[m] + 2) * (ny_in[m] + 2) : (long)nx_in[m] * ny_in[m];
  }
  if (off > 2000000000L || (long)nx_out * ny_out > 2000000000L) { delete pl; return fail(FG_ERR_ARG, "grid too large for 32-bit cell indices"); }
  pl->nsrc = (int)off; pl->f_stride = foff;
  pl->nx_out = nx_out; pl->ny_out = ny_out; pl->ndst = nx_out * ny_out;
  pl->stream = g_handles.get_stream(device);
  if (!pl->stream) { delete pl; return fail(FG_ERR_HIP, "hipStreamCreate failed"); }
  *out = pl;
  return 0;
}

extern "C" void fg_plan_destroy(fg_plan *pl)
{
  if (!pl) return;
  (void)hipSetDevice(pl->device);
  if (pl->stream || !pl->own_stream) (void)hipStreamSynchronize(pl->stream);
  if (pl->stream && pl->own_stream) g_handles.put_stream(pl->device, pl->stream);
  if (pl->stream_b) { (void)hipStreamSynchronize(pl->stream_b); g_handles.put_stream(pl->device, pl->stream_b); }
  { float junk[PH_COUNT] = {0}; pl->apply_pt.collect(junk); }
  for (void *p : pl->owned) g_pool.put(p);
  delete pl;
}

static void choose_bins(const fg_plan *pl, double mean_dlat, double mean_dlon, FgBins *b)
{
  const double PI = 3.14159265358979323846;
  // a typical cell covers at most 2x2 bins.  Measured on C384 -> 0.25 deg: factors 1.1 / 1.25 / 1.5 / 2.0 give 0.247 / 0.257 /
  // 0.271 / 0.334 ms for the candidate phase (fewer records per scanned bin); 1.25 keeps a margin for cells larger than the mean
  double h = 1.25 * mean_dlat, w = 1.25 * mean_dlon;
  int nblat = (h > 0) ? (int)ceil(PI / h) : 1;
  int nblon = (w > 0) ? (int)ceil(2.0 * PI / w) : 1;
  if (nblat < 1) nblat = 1; if (nblat > 8192) nblat = 8192;
  if (nblon < 1) nblon = 1; if (nblon > 16384) nblon = 16384;
  (void)pl;
  b->nblat = nblat; b->nblon = nblon;
  b->inv_wlat = nblat / PI;
  b->inv_wlon = nblon / (2.0 * PI);
}

// Device-side counters of one search (FgCounters, xgrid_device.h): one block, zeroed by the search's one memset, read back with
// one copy into pinned host memory when the search has been queued.
static FgCounters *pinned_counters()
{
  static thread_local FgCounters *h = nullptr;
  if (!h) { if (hipHostMalloc((void **)&h, sizeof(FgCounters), hipHostMallocDefault) != hipSuccess) h = nullptr; }
  return h;
}

// the search proper; all grid pointers are device pointers
// Great-circle plans pass the corner unit vectors instead (gc_in[m] = {x, y, z} device pointers of source tile m,
// gc_out likewise); d_lon/d_lat are then unused.
struct GcXyz { const double *x, *y, *z; };
// 1-D x 2-D variants: the "source" tile of the search is the box grid expanded to corner arrays (d_lon_in[0]/d_lat_in[0]),
// the "destination" the quads; box holds the 1-D bounds the pair kernel clips against.
struct BoxMode { FgBox box; const double *mask_quad; int no_adjust; };
static const char *gc_clip_error(int code)
{
  switch (code) {
    case 3: return "firstIntersect is not in the grid1List";
    case 4: return " not found the next intersection ";
    case 5: return "not return back to the first intersection";
    case 6: return "After clipping, nintersect should be 0";
    case 7: return "inserAfter: point (x,y,z) is not found in the list";
    case 8: return "Error from create_xgrid.c: temp is not in list1";
    default: return "clip_2dx2d_great_circle: more intersections than two convex quadrilaterals can have (list capacity)";
  }
}

// Capacities of one attempt: bin-table records and entries per region of the pair list.  The default capacities (bin records
// 3*ndst, pairs 8*max(nsrc, ndst)) fit every remap between grids of comparable resolution; every kernel clamps its writes
// AND reads to them, the counters keep counting, and an attempt that outgrew one is repeated with the counted sizes.
static int g_search_cull = 0;
static int g_search_finalize = 0;     // fg_set_search_finalize: a search that can also queues the finalize work before its one synchronisation
// great-circle clip: 1 = three passes (k_gc_screen / k_gc_solve / k_gc_walk) with the one-kernel clip for the unusual pairs,
// 0 = the one-kernel clip for every pair (fg_set_gc_split; the tests compare the two)
static int g_gc_split = 1;
extern "C" void fg_set_gc_split(int on) { g_gc_split = on < 0 ? 0 : on; }      // 2: a task space 64 times too small (tests of the overflow path)
struct SearchCaps { unsigned long long entries; int regcap, nreg; bool rect; };
// Chunks of source cells per search (1 = everything on one stream, in sequence; fg_set_search_chunks / FREGRID_HIP_CHUNKS).  Measured at C384 -> 0.25 deg with 4 chunks: the
// kernels slow each other down by more than the overlap wins (clip 4 x 184 us against 482, step 1.48 ms against 1.30), so the
// default is ONE chunk; the machinery stays for grids where the balance differs.
static int g_search_chunks = 0;
extern "C" void fg_set_search_chunks(int k) { g_search_chunks = k < 0 ? 0 : (k > FG_MAX_CHUNKS ? FG_MAX_CHUNKS : k); }
static int choose_chunks(int nsrc, int nreg)
{
  static const int env_k = getenv("FREGRID_HIP_CHUNKS") ? atoi(getenv("FREGRID_HIP_CHUNKS")) : 0;
  int k = g_search_chunks ? g_search_chunks : (env_k > 0 ? env_k : 1);
  if (k > FG_MAX_CHUNKS) k = FG_MAX_CHUNKS;
  const bool forced = g_search_chunks > 0 || env_k > 0;
  while (k > 1 && (nreg / k < 1 || (!forced && nsrc / k < 4096))) k--;
  return k;
}
// Rectilinear destination grids (k_rect_tables, xgrid_kernels.hip): 1 = try the index-arithmetic path first and verify the grid on
// the device in the same stream (default), 0 = always the generic bins path.  A target that fails the check costs the attempt's
// launches (every kernel leaves at once) and one more synchronisation, ~0.1 ms; callers with host arrays are spared even that by
// a look at a few corners (host_says_not_rect).
static int g_search_rect = 1;
extern "C" void fg_set_search_rect(int on) { g_search_rect = on ? 1 : 0; }
// Longitude frame of the DESTINATION cells of a legacy search.  0 (default) = create_xgrid's: fix_lon(cell, pi), then per pair a
// shift of +-2pi towards the source cell's mean longitude (create_xgrid.c:1004,1062-1079).  1 = make_coupler_mosaic's: the cell is
// only unwrapped and then moved ONCE per pair, fix_lon(cell, mean longitude of the other cell) (make_coupler_mosaic.c:1452,1598).
// The two agree unless the first recentring moved the cell and the second moved it back -- (x + 2pi) - 2pi is not x in floating
// point -- as for an ocean grid given on -280..80 degrees.  Implementation: the recentring target of the destination records is
// NaN (every comparison of fix_lon's last step is then false), the per-pair shift is the same code.
static int g_dst_frame = 0;
extern "C" void fg_set_search_frame(int coupler) { g_dst_frame = coupler ? 1 : 0; }
static bool host_says_not_rect(int nx, int ny, const double *lon, const double *lat)
{
  const long nxp = nx + 1;
  const int js[3] = {ny, ny / 2, 1}, is[3] = {nx, nx / 2, 1};
  for (int a = 0; a < 3; a++)
    for (int b = 0; b < 3; b++) {
      const long e = (long)js[a] * nxp + is[b];
      if (memcmp(&lon[e], &lon[is[b]], sizeof(double)) || memcmp(&lat[e], &lat[(long)js[a] * nxp], sizeof(double))) return true;
    }
  return false;
}
#define FG_RETRY (-1000L)
static long plan_search_core(fg_plan *pl, const double *const *d_lon_in, const double *const *d_lat_in,
                        const double *const *d_mask_in, const double *d_lon_out, const double *d_lat_out,
                        double mean_dlat, double mean_dlon, const GcXyz *gc_in, const GcXyz *gc_out,
                        const BoxMode *boxm, SearchCaps *caps)
{
  // Queue order (legacy search, 11 launches + 1 readback; the host synchronises ONCE, at the end):
  //   memset | cell records + bin counts | bin scan | bin fill + heavy list | candidates | quad clip | general clip |
  //   compaction (+ big cells) | row scan | row slots -> perm | counters -> host
  const bool gc = gc_in != nullptr;
  pl->great_circle = gc;
  hipStream_t st = pl->stream;
  const int nsrc = pl->nsrc, ndst = pl->ndst, order = pl->order;
  FgCounters *hc = pinned_counters();
  if (!hc) return fail(FG_ERR_HIP, "hipHostMalloc failed");

  // tile descriptors: source tiles + the destination tile as entry [ntiles]
  std::vector<FgTile> &th = pl->tiles_host;  // owned by the plan: an async upload needs no sync
  th.resize(pl->ntiles + 1);
  for (int m = 0; m < pl->ntiles; m++) th[m] = FgTile{gc ? nullptr : d_lon_in[m], gc ? nullptr : d_lat_in[m], pl->nx_in[m], pl->ny_in[m], pl->cell_off[m]};
  th[pl->ntiles] = FgTile{d_lon_out, d_lat_out, pl->nx_out, pl->ny_out, 0};
  pl->tiles_dev = pl->alloc<FgTile>(pl->ntiles + 1);
  if (!pl->tiles_dev) return fail(FG_ERR_HIP, "out of device memory");
  FgTileSet ts; ts.n = 0;
  const bool by_value = !gc && pl->ntiles + 1 <= FG_TILESET_MAX;        // descriptors travel as a kernel argument
  if (by_value) { ts.n = pl->ntiles + 1; for (int m = 0; m <= pl->ntiles; m++) ts.t[m] = th[m]; }
  else HIPCHK(hipMemcpyAsync(pl->tiles_dev, th.data(), sizeof(FgTile) * th.size(), hipMemcpyHostToDevice, st));
  FgTileXyz *gct_dev = nullptr;
  std::vector<FgTileXyz> gct;                       // must outlive the async upload: synchronised at the readback
  if (gc) {
    gct.resize(pl->ntiles + 1);
    for (int m = 0; m < pl->ntiles; m++) gct[m] = FgTileXyz{gc_in[m].x, gc_in[m].y, gc_in[m].z, pl->nx_in[m], pl->ny_in[m], pl->cell_off[m]};
    gct[pl->ntiles] = FgTileXyz{gc_out->x, gc_out->y, gc_out->z, pl->nx_out, pl->ny_out, 0};
    gct_dev = pl->alloc<FgTileXyz>(pl->ntiles + 1);
    if (!gct_dev) return fail(FG_ERR_HIP, "out of device memory");
    HIPCHK(hipMemcpyAsync(gct_dev, gct.data(), sizeof(FgTileXyz) * gct.size(), hipMemcpyHostToDevice, st));
  }

  bool any_mask = false;
  if (d_mask_in) for (int m = 0; m < pl->ntiles; m++) if (d_mask_in[m]) any_mask = true;
  if (any_mask) {
    pl->mask_dev = pl->alloc<double>(nsrc);
    if (!pl->mask_dev) return fail(FG_ERR_HIP, "out of device memory");
    std::vector<double> ones;
    for (int m = 0; m < pl->ntiles; m++) {
      size_t nc = (size_t)pl->nx_in[m] * pl->ny_in[m];
      if (d_mask_in[m]) HIPCHK(hipMemcpyAsync(pl->mask_dev + pl->cell_off[m], d_mask_in[m], nc * sizeof(double), hipMemcpyDeviceToDevice, st));
      else {
        ones.assign(nc, 1.0);
        HIPCHK(hipMemcpyAsync(pl->mask_dev + pl->cell_off[m], ones.data(), nc * sizeof(double), hipMemcpyHostToDevice, st));
        HIPCHK(hipStreamSynchronize(st));
      }
    }
  }

  // --- sizes known up front
  const bool rect = caps->rect && !gc && !boxm;
  pl->rect = rect;
  const double dst_tlon = g_dst_frame ? (double)NAN : 3.14159265358979323846;
  FgBins bins;
  choose_bins(pl, mean_dlat, mean_dlon, &bins);
  if (rect) { bins.nblat = 0; bins.nblon = 0; }       // no bins on the rectilinear path
  const long nbins = (long)bins.nblat * bins.nblon;
  const long nslots = nbins + bins.nblat;             // regular bins + one wide list per bin row
  const unsigned long long nentries = rect ? 0ull : caps->entries;
  const int ecap = (int)std::min<unsigned long long>(nentries, 2147483647ull);     // records the buffer holds: writes and reads stop there
  FgPairSpace ps{};
  ps.nreg = caps->nreg; ps.regcap = caps->regcap;
  const long npairs = fgd_pairs_total(ps);            // capacity of the pair list
  const long nx_alloc = npairs;                       // nxgrid <= candidate pairs <= capacity

  if (!alloc_cells(pl, &pl->S, nsrc)) return fail(FG_ERR_HIP, "out of device memory");
  double *rect_blk = nullptr;
  if (rect) {
    pl->D = FgCells{};
    pl->D.area = pl->alloc<double>(ndst);
    rect_blk = pl->alloc<double>(8 + (size_t)(pl->ny_out + 1) + (size_t)(pl->nx_out + 1) + 8 * (size_t)pl->nx_out + 4 * (size_t)(pl->ny_out + 1));
    if (!pl->D.area || !rect_blk) return fail(FG_ERR_HIP, "out of device memory");
  } else if (!alloc_cells(pl, &pl->D, ndst)) return fail(FG_ERR_HIP, "out of device memory");
  // one zeroed block: [counters | region fill counters | tickets | look-back words of the three scans | bin counts |
  //                    bin fill cursors | destination-row counts | accepted pairs per source cell]
  const int K = rect ? 1 : choose_chunks(nsrc, ps.nreg);
  const long t_bins = fgd_scan_tiles(nslots), t_rows = fgd_scan_tiles(ndst), t_comp = fgd_scan_tiles(nsrc) + K;
  const size_t zc = (sizeof(FgCounters) + 127) / 128 * 128;
  const size_t zfill = (size_t)FG_NREG * FG_FILL_STRIDE * sizeof(unsigned);      // (nreg <= FG_NREG)
  const size_t ztick = 128;
  const size_t zlb = (size_t)(t_bins + t_rows + t_comp) * sizeof(unsigned long long);
  const size_t zints = ((size_t)(2 * (nslots + 1) + ndst + 1 + nsrc + 1) * sizeof(int) + 15) / 16 * 16;
  const size_t zgc = gc ? 2 * (size_t)K * FG_NREG * FG_FILL_STRIDE * sizeof(unsigned) : 0;    // great-circle clip: task counters, then the counters of the walk's pair lists
  const size_t zbytes = zc + zfill + ztick + zlb + zints + zgc;
  char *zero_blk = pl->alloc<char>(zbytes);
  int *bin_start = pl->alloc<int>(nslots + 1);
  int *heavy_list = pl->alloc<int>(nsrc + 1);
  int *big_list = pl->alloc<int>(nsrc + 1);
  int *pair_beg = pl->alloc<int>(nsrc + 1), *pair_cnt = pl->alloc<int>(nsrc + 1);
  FgBinEntry *bin_entries = pl->alloc<FgBinEntry>(nentries ? nentries : 1);
  ps.src = pl->alloc<int>(npairs + 1); ps.dst = pl->alloc<int>(npairs + 1);
  double *tmp_area = pl->alloc<double>(npairs + 1);
  double *tmp_clon = (order == 2) ? pl->alloc<double>(npairs + 1) : nullptr;
  double *tmp_clat = (order == 2) ? pl->alloc<double>(npairs + 1) : nullptr;
  int *defer_list = pl->alloc<int>(npairs + 1);
  // legacy clip on one chunk: the clip kernels take the destination-row slots, the row scan runs before the compaction and the
  // compaction stores the row lists itself (no k_csr_fill_pos, no returning atomics in the compaction)
  const bool early_rows = !gc && !boxm && K == 1;
  int *tmp_rowpos = early_rows ? pl->alloc<int>(npairs + 1) : nullptr;
  if (early_rows && !tmp_rowpos) return fail(FG_ERR_HIP, "out of device memory");
  // great-circle path, three-pass clip: per-pair words, and 3 tasks (edge pairs to solve) per pair of capacity -- 3.0 per LIVE
  // pair were counted at C384 -> 0.25 deg; pairs whose tasks do not fit go through the one-kernel clip instead
  const bool gc_split = gc && g_gc_split && npairs < (1L << 28);
  const long tcap_want = (g_gc_split == 2) ? npairs / 64 : 3 * npairs;
  const long tcap_reg = gc_split ? std::min<long>((tcap_want / K / FG_NREG + 255) / 256 * 256, 0x7ffffff0L / (K * FG_NREG)) : 0;   // tasks per region
  const long tcap_all = tcap_reg * K * FG_NREG;
  unsigned *gc_meta = gc_split ? pl->alloc<unsigned>(npairs + 1) : nullptr;
  int *gc_tbase = gc_split ? pl->alloc<int>(npairs + 1) : nullptr;
  int *gc_order = gc_split ? pl->alloc<int>(npairs + 1) : nullptr;
  unsigned *gc_task = gc_split ? pl->alloc<unsigned>(tcap_all + 1) : nullptr;
  double *gc_res = gc_split ? pl->alloc<double>(2 * (size_t)tcap_all + 2) : nullptr;
  if (gc_split && (!gc_meta || !gc_tbase || !gc_order || !gc_task || !gc_res)) return fail(FG_ERR_HIP, "out of device memory");
  pl->xoff = pl->alloc<int>(nsrc + 1);
  pl->x_src = pl->alloc<int>(nx_alloc + 1); pl->x_dst = pl->alloc<int>(nx_alloc + 1);
  pl->x_area = pl->alloc<double>(nx_alloc + 1);
  if (order == 2) { pl->x_c1 = pl->alloc<double>(nx_alloc + 1); pl->x_c2 = pl->alloc<double>(nx_alloc + 1); pl->sums = pl->alloc<double>(3 * (size_t)nsrc); }
  pl->x_rowpos = pl->alloc<int>(nx_alloc + 1);
  pl->perm = pl->alloc<int>(nx_alloc + 1);
  pl->csr.row_ptr = pl->alloc<int>(ndst + 1);
  if (!pl->src_idx_f) pl->src_idx_f = pl->alloc<int>(nsrc + 1);
  if (!zero_blk || !bin_start || !heavy_list || !big_list || !pair_beg || !pair_cnt || !bin_entries || !ps.src || !ps.dst || !tmp_area ||
      !defer_list || (order == 2 && (!tmp_clon || !tmp_clat || !pl->x_c1 || !pl->x_c2 || !pl->sums)) || !pl->xoff || !pl->x_src || !pl->x_dst ||
      !pl->x_area || !pl->x_rowpos || !pl->perm || !pl->csr.row_ptr || !pl->src_idx_f)
    return fail(FG_ERR_HIP, "out of device memory");
  FgCounters *dc = (FgCounters *)zero_blk;
  ps.fill = (unsigned *)(zero_blk + zc);
  unsigned *tickets = (unsigned *)(zero_blk + zc + zfill);                 // [0] bins [1] rows [2 + k] xoff scan of chunk k
  unsigned long long *lb_bins = (unsigned long long *)(zero_blk + zc + zfill + ztick), *lb_rows = lb_bins + t_bins, *lb_comp = lb_rows + t_rows;
  int *bin_cnt = (int *)(zero_blk + zc + zfill + ztick + zlb), *bin_fill = bin_cnt + (nslots + 1), *row_cnt = bin_fill + (nslots + 1);
  int *nacc = row_cnt + (ndst + 1);
  HIPCHK(hipMemsetAsync(zero_blk, 0, zbytes, st));

  PhaseTimer pt, ptot;
  pt.start(g_profiling != 0, st); ptot.start(g_profiling != 0, st);
  for (int k = 0; k < PH_COUNT; k++) pl->phase_ms[k] = 0;
  ptot.begin(PH_SEARCH_TOTAL);
  pt.begin(PH_CELL_STRUCT);
  if (gc) {
    // destination cells first: a culling search (a rank's band of the target) folds their latitude ranges into band_keys, and
    // the source launch drops the cells that cannot meet that range before their (expensive) spherical-excess area
    fgd_gc_cell_struct(gct_dev + pl->ntiles, 1, ndst, pl->D, st, g_search_cull ? dc->band_keys : nullptr, 1);
    fgd_gc_cell_struct(gct_dev, pl->ntiles, nsrc, pl->S, st, g_search_cull ? dc->band_keys : nullptr, 2);
    fgd_src_field_index(order, pl->tiles_dev, pl->ntiles, nsrc, pl->src_idx_f, st);
  } else if (rect) {
    // tables + on-device verification of the grid, then the source records, the heavy list and the destination AREAS in one launch
    FgRect &R = pl->rect_tab;
    R.hdr = rect_blk; R.lat_ax = rect_blk + 8; R.lon_ax = rect_blk + 8 + (pl->ny_out + 1); R.col = rect_blk + 8 + (pl->ny_out + 1) + (pl->nx_out + 1);
    double *rect_row = rect_blk + 8 + (pl->ny_out + 1) + (pl->nx_out + 1) + 8 * (size_t)pl->nx_out;
    R.row = rect_row;
    R.bad = &dc->rect_bad; R.nx = pl->nx_out; R.ny = pl->ny_out;
    fgd_rect_tables(d_lon_out, d_lat_out, pl->nx_out, pl->ny_out, rect_blk, rect_blk + 8, rect_blk + 8 + (pl->ny_out + 1),
                    rect_blk + 8 + (pl->ny_out + 1) + (pl->nx_out + 1), rect_row, &dc->rect_bad, dc->err, st, dst_tlon);
    if (pl->polys.npoly) fgd_polylist_records(pl->polys, pl->S, pl->src_idx_f, pl->sums, &R, heavy_list, &dc->heavy_cnt, dc->err, st);
    fgd_cell_struct2r(ts, pl->tiles_dev, pl->tiles_dev, pl->ntiles, pl->polys.npoly ? 0 : nsrc, ndst, pl->S, pl->D.area, R, pl->mask_dev, order,
                      pl->src_idx_f, pl->sums, dc->err, st, dc->band_keys, (g_search_cull && !pl->polys.npoly) ? 2 : 0, heavy_list, &dc->heavy_cnt);
  } else if (g_search_cull && !boxm && !pl->polys.npoly) {
    // the destination grid's latitude range from its corners (a 5 us reduction), then ONE record launch in which the source
    // blocks that cannot meet it leave early (round 2 first ran a destination launch, then a source launch: two latency floors)
    fgd_band_keys(d_lat_out, (long)(pl->nx_out + 1) * (pl->ny_out + 1), dc->band_keys, st);
    fgd_cell_struct2(ts, pl->tiles_dev, pl->tiles_dev, pl->ntiles, nsrc, ndst, pl->S, pl->D, bins, bin_cnt, order, pl->src_idx_f, pl->sums, dc->err, st, dc->band_keys, 2, dst_tlon);
  } else if (pl->polys.npoly) {
    fgd_polylist_records(pl->polys, pl->S, pl->src_idx_f, pl->sums, nullptr, nullptr, nullptr, dc->err, st);
    fgd_cell_struct2(ts, pl->tiles_dev, pl->tiles_dev, pl->ntiles, 0, ndst, pl->S, pl->D, bins, bin_cnt, order, pl->src_idx_f, pl->sums, dc->err, st, nullptr, 0, dst_tlon);
  } else
    fgd_cell_struct2(ts, pl->tiles_dev, pl->tiles_dev, pl->ntiles, nsrc, ndst, pl->S, pl->D, bins, bin_cnt, order, pl->src_idx_f, pl->sums, dc->err, st, nullptr, 0, dst_tlon);
  if (boxm) fgd_box_cell_boxes(boxm->box, pl->S, st);
  if (boxm && boxm->no_adjust) fgd_box_area_no_adjust(boxm->box, pl->S.area, st);      // create_xgrid.c:239-242
  pt.end();
  pl->have_geom = true;

  if (!rect) {
    pt.begin(PH_BINS);
    if (gc) fgd_bin_count(ndst, pl->D, bins, bin_cnt, st);
    fgd_exclusive_scan1(bin_cnt, nslots, bin_start, lb_bins, &tickets[0], &dc->total[0], dc->err, st);
    fgd_bin_fill(ndst, pl->D, bins, bin_fill, bin_start, bin_entries, ecap, nsrc, pl->S, pl->mask_dev, heavy_list, &dc->heavy_cnt, st);
    pt.end();
  }

  // --- per chunk of source cells: candidates (stream A) -> clip (stream B) -> scan + compaction (stream A again).  The clip of
  // chunk k is VALU bound, its neighbours in the schedule wait on memory: side by side they fill each other's gaps.
  hipStream_t sb = st;
  if (K > 1) {
    if (!pl->stream_b) pl->stream_b = g_handles.get_stream(pl->device);
    if (!pl->stream_b) return fail(FG_ERR_HIP, "hipStreamCreate failed");
    sb = pl->stream_b;
  }
  std::vector<hipEvent_t> ev_c(K, nullptr), ev_q(K, nullptr);
  hipEvent_t gc_e1 = nullptr, gc_e2 = nullptr;        // great-circle clip: the listed pairs run on stream B beside k_gc_walk
  struct EvReturn {                                   // back to the cache on every way out of this function (HIPCHK / fail returns too)
    hipEvent_t &a, &b;
    ~EvReturn() { g_handles.put_sync_event(a); g_handles.put_sync_event(b); a = b = nullptr; }
  } ev_return{gc_e1, gc_e2};
  if (gc_split && K == 1) {
    if (!pl->stream_b) pl->stream_b = g_handles.get_stream(pl->device);
    gc_e1 = g_handles.get_sync_event(); gc_e2 = g_handles.get_sync_event();
  }
  PhaseTimer ptb; ptb.start(g_profiling != 0, sb);
  int cb[FG_MAX_CHUNKS + 1];
  for (int k = 0; k <= K; k++) cb[k] = (k == K) ? nsrc : (int)(((long)nsrc * k / K) / 256 * 256);
  const int nreg_k = ps.nreg / K;                     // regions per chunk (choose_chunks keeps this >= 1)
  const long pcap_k = (long)nreg_k * ps.regcap;       // pairs per chunk
  auto chunk_ps = [&](int k) { FgPairSpace q = ps; q.nreg = nreg_k; q.src = ps.src + k * pcap_k; q.dst = ps.dst + k * pcap_k;
                               q.fill = ps.fill + (size_t)k * nreg_k * FG_FILL_STRIDE; return q; };
  pt.begin(PH_CANDIDATES);
  for (int k = 0; k < K; k++) {
    if (rect)
      fgd_candidates_rect(nsrc, pl->S, pl->mask_dev, pl->rect_tab, chunk_ps(k), pair_beg, pair_cnt, heavy_list, &dc->heavy_cnt, big_list, &dc->big_cnt[k], st);
    else
      fgd_candidates1(cb[k], cb[k + 1], pl->S, pl->mask_dev, bins, bin_start, bin_entries, ecap, chunk_ps(k), pair_beg, pair_cnt, heavy_list,
                      &dc->heavy_cnt, big_list + cb[k], &dc->big_cnt[k], st);
    if (K > 1) { ev_c[k] = g_handles.get_sync_event(); HIPCHK(hipEventRecord(ev_c[k], st)); }
  }
  pt.end();
  // --- clip, area, centroid integrals
  for (int k = 0; k < K; k++) {
    const FgPairSpace q = chunk_ps(k);
    double *ta = tmp_area + k * pcap_k, *tl = tmp_clon ? tmp_clon + k * pcap_k : nullptr, *tt = tmp_clat ? tmp_clat + k * pcap_k : nullptr;
    int *dl = defer_list + k * pcap_k;
    if (K > 1) HIPCHK(hipStreamWaitEvent(sb, ev_c[k], 0));
    if (boxm) {
      ptb.begin(PH_CLIP_GENERAL);
      fgd_clip_box(order, q, boxm->box, th[pl->ntiles], pl->S, pl->D, pl->mask_dev, boxm->mask_quad, ta, tl, tt, nacc, dc->stats, dc->err, sb);
      ptb.end();
    } else if (gc) {
      ptb.begin(PH_CLIP_GENERAL);
      if (gc_split) {
        const long tcap_k = tcap_reg * FG_NREG;
        unsigned *ntask = (unsigned *)(zero_blk + zbytes - zgc) + (size_t)k * FG_NREG * FG_FILL_STRIDE;
        GcSplit g{gc_meta + k * pcap_k, gc_tbase + k * pcap_k, gc_task + k * tcap_k, gc_res + 2 * k * tcap_k, (unsigned)tcap_reg,
                  ntask, dl, &dc->defer_cnt[k], &dc->gc_list2_cnt[k], pcap_k,
                  gc_order + k * pcap_k, ntask + (size_t)K * FG_NREG * FG_FILL_STRIDE};
        fgd_gc_clip_split(q, pl->S, pl->mask_dev, pl->D, ta, nacc, g, dc->stats, dc->err, sb, K == 1 ? pl->stream_b : nullptr, gc_e1, gc_e2);
      } else
        fgd_gc_clip(q, pl->S, pl->mask_dev, pl->D, ta, nacc, dl, &dc->defer_cnt[k], dc->stats, dc->err, sb);
      ptb.end();
    } else {
      ptb.begin(PH_CLIP_QUAD);
      fgd_clip_quad(order, q, pl->S, pl->mask_dev, pl->D, ta, tl, tt, nacc, dl, &dc->defer_cnt[k], dc->stats, dc->err, sb, rect ? &pl->rect_tab : nullptr,
                    early_rows ? row_cnt : nullptr, tmp_rowpos);
      ptb.end();
      ptb.begin(PH_CLIP_GENERAL);
      fgd_clip_general(order, q, pl->S, pl->mask_dev, pl->D, ta, tl, tt, nacc, dl, &dc->defer_cnt[k], dc->stats, dc->err, sb, rect ? &pl->rect_tab : nullptr,
                       early_rows ? row_cnt : nullptr, tmp_rowpos);
      ptb.end();
    }
    if (K > 1) { ev_q[k] = g_handles.get_sync_event(); HIPCHK(hipEventRecord(ev_q[k], sb)); }
  }
  // --- compaction into canonical order, per-source-cell sums, destination-row slots
  long tile0 = 0;
  for (int k = 0; k < K; k++) {
    const FgPairSpace q = chunk_ps(k);
    if (K > 1) HIPCHK(hipStreamWaitEvent(st, ev_q[k], 0));
    pt.begin(PH_COMPACT);
    FgCompactIo io{};
    io.pair_beg = pair_beg; io.pair_cnt = pair_cnt;
    io.tmp_area = tmp_area + k * pcap_k; io.tmp_clon = tmp_clon ? tmp_clon + k * pcap_k : nullptr; io.tmp_clat = tmp_clat ? tmp_clat + k * pcap_k : nullptr;
    io.xoff = pl->xoff; io.x_src = pl->x_src; io.x_dst = pl->x_dst; io.x_area = pl->x_area; io.x_c1 = pl->x_c1; io.x_c2 = pl->x_c2;
    io.row_cnt = row_cnt; io.x_rowpos = pl->x_rowpos; io.sums = pl->sums; io.big_list = big_list + cb[k]; io.big_cnt = &dc->big_cnt[k];
    io.tmp_rowpos = tmp_rowpos; io.row_ptr = pl->csr.row_ptr; io.perm = pl->perm;
    io.fill_all = (k == K - 1) ? ps.fill : nullptr; io.nreg_all = nreg_k * K;
    io.dc = dc; io.xcap = nx_alloc;
    const int nk = cb[k + 1] - cb[k];
    if (early_rows)       // accepted pairs per source cell -> xoff, destination-row counts -> row_ptr: one launch
      fgd_exclusive_scan2(nacc, nk, pl->xoff, lb_comp, &tickets[2], &dc->xtot[0], row_cnt, ndst, pl->csr.row_ptr, lb_rows, &tickets[1], &dc->rows_total, dc->err, st);
    else
      fgd_exclusive_scan1(nacc + cb[k], nk, pl->xoff + cb[k], lb_comp + tile0, &tickets[2 + k], &dc->xtot[k], dc->err, st, k ? &dc->xtot[k - 1] : nullptr);
    tile0 += fgd_scan_tiles(nk);
    fgd_compact(order, nsrc, q, io, st);
    pt.end();
  }
  if (!early_rows) {
    pt.begin(PH_ROWS);
    fgd_exclusive_scan1(row_cnt, ndst, pl->csr.row_ptr, lb_rows, &tickets[1], &dc->rows_total, dc->err, st);
    fgd_csr_fill_pos(nx_alloc, &dc->xtot[K - 1], pl->x_dst, pl->csr.row_ptr, pl->x_rowpos, pl->perm, st);
    pt.end();
  }
  ptot.end();
  // fg_set_search_finalize(1): the centroid pass and the CSR records of fg_plan_finalize queued HERE, behind the compaction and
  // before the one synchronisation -- a single-plan job (one destination tile, one rank: the plan's own per-cell sums are the
  // totals) then has no host round trip and no idle GPU between its search and its finalize (25-30 us of a 0.9 ms step; rocprofv3
  // kernel timeline, scripts/band_trace.sh).  The records are sized by the capacity of the exchange-cell arrays (the count is
  // not known yet); an attempt that has to be repeated drops them with everything else.
  const bool fuse = g_search_finalize && early_rows && !pl->polys.npoly;
  int *fused_tmp = nullptr;
  if (fuse) {
    pt.begin(PH_FINALIZE);
    if (order == 2) {
      pl->cen = pl->alloc<double>(2 * (size_t)nsrc);
      if (!pl->cen) return fail(FG_ERR_HIP, "out of device memory");
      fgd_centroids(nsrc, pl->S, pl->sums, pl->cen, st);
      pl->csr.e2 = pl->alloc<FgCsrEntry2>(nx_alloc + 1);
    } else pl->csr.e1 = pl->alloc<FgCsrEntry1>(nx_alloc + 1);
    if (!pl->csr.e1 && !pl->csr.e2) return fail(FG_ERR_HIP, "out of device memory");
    // (the row-length heuristic of the record kernel wants the exchange-cell count: nsrc + ndst is within a factor of it)
    int *sg_tmp = pl->alloc<int>(2 * (size_t)(nx_alloc + 1));   // (scratch for rows beyond the LDS staging; null = the serial last resort)
    fgd_csr_sortgather(order, ndst, (long)nsrc + ndst, pl->perm, pl->x_src, pl->x_area, pl->x_c1, pl->x_c2, pl->src_idx_f,
                       order == 2 ? pl->cen : nullptr, nsrc, pl->csr, st, sg_tmp, nx_alloc + 1);
    fused_tmp = sg_tmp;
    pt.end();
  }
  HIPCHK(hipMemcpyAsync(hc, dc, sizeof(FgCounters), hipMemcpyDeviceToHost, st));
  HIPCHK(hipStreamSynchronize(st));                              // the one synchronisation of a search (stream B's work is ordered before it)
  pt.collect(pl->phase_ms); ptot.collect(pl->phase_ms); ptb.collect(pl->phase_ms);   // (also hands the timing events back on every exit below)
  for (hipEvent_t e : ev_c) g_handles.put_sync_event(e);
  for (hipEvent_t e : ev_q) g_handles.put_sync_event(e);
  HIPCHK(hipGetLastError());
  if (rect && hc->rect_bad) { caps->rect = false; return FG_RETRY; }   // not a rectilinear grid after all: every kernel of this attempt left at once
  if (hc->total[0] > nentries) { caps->entries = hc->total[0]; return FG_RETRY; }
  if (hc->total[0] > 2000000000ull) return fail(FG_ERR_ARG, "bin table too large");
  if (hc->total[3] > (unsigned long long)ps.regcap) {
    if (hc->total[3] > 2000000000ull / (unsigned long long)ps.nreg) return fail(FG_ERR_CAPACITY, "candidate pair list exceeds 2^31 entries");
    caps->regcap = (int)((hc->total[3] + FG_REG_ALIGN - 1) / FG_REG_ALIGN * FG_REG_ALIGN);
    return FG_RETRY;
  }
  if (hc->err[0] & 8u) return fail(FG_ERR_ARG, "a grid corner latitude lies outside [-pi/2, pi/2] (radians expected)");
  if (hc->err[0] & 1u) return fail(FG_ERR_MAXV, "create_xgrid.c: n2_in is greater than MAX_V");
  if (hc->err[0] & 2u) return fail(FG_ERR_PARALLEL, "the line between <x1_0,y1_0> and  <x1_1,y1_1> should not parallel to "
                                                    "the line between <x2_0,y2_0> and  <x2_1,y2_1>");
  if (hc->err[0] & 4u) return fail(FG_ERR_MAXV, "clipped polygon exceeds 16 vertices");
  if (hc->err[0] & G_ERRBIT_GC_CONVEX1) return fail(FG_ERR_GEOM, "create_xgrid.c(clip_2dx2d_great_circle): grid box 1 is not convex");
  if (hc->err[0] & G_ERRBIT_GC_CONVEX2) return fail(FG_ERR_GEOM, "create_xgrid.c(clip_2dx2d_great_circle): grid box 2 is not convex");
  if (hc->err[0] & G_ERRBIT_GC_CLIP) return fail(FG_ERR_GEOM, "%s", gc_clip_error((int)hc->err[1]));
  if (hc->err[0] & G_ERRBIT_LOOKBACK) return fail(FG_ERR_HIP, "single-pass scan: a tile waited too long for its predecessor");
  pl->nx = (long)hc->xtot[K - 1];
  pl->stats[FG_STAT_PAIRS] = (long)hc->total[1];
  pl->stats[FG_STAT_NONEMPTY] = (long)(hc->xtot[K - 1] + hc->stats[FG_STAT_BELOW]);
  pl->stats[FG_STAT_NXGRID] = pl->nx;
  pl->stats[FG_STAT_BORDERLINE] = (long)hc->stats[FG_STAT_BORDERLINE];
  pl->stats[FG_STAT_BINS] = nbins;
  pl->stats[FG_STAT_BIN_ENTRIES] = (long)hc->total[0];
  pl->stats[FG_STAT_DEFERRED] = 0;
  for (int k = 0; k < K; k++) pl->stats[FG_STAT_DEFERRED] += hc->defer_cnt[k] + hc->gc_list2_cnt[k];
  pl->stats[FG_STAT_HEAVY] = hc->heavy_cnt;
  pl->stats[FG_STAT_BELOW] = (long)hc->stats[FG_STAT_BELOW];

  pl->rect_tab.bad = nullptr;                          // (lives in the scratch block released below; nothing reads it after the search)
  // scratch no longer needed
  void *scratch[] = {zero_blk, bin_start, bin_entries, heavy_list, big_list, pair_beg, pair_cnt, ps.src, ps.dst,
                     tmp_area, tmp_clon, tmp_clat, defer_list, gc_meta, gc_tbase, gc_order, gc_task, gc_res, tmp_rowpos};
  for (void *p : scratch) pl->release(p);
  pl->release(pl->x_rowpos); pl->x_rowpos = nullptr;
  pl->rows_built = true;
  pl->searched = true;
  if (fuse) {
    pl->release(pl->perm); pl->perm = nullptr; pl->release(fused_tmp);
    if (!pl->red_partial) { pl->red_partial = pl->alloc<double>(1024); pl->red_result = pl->alloc<double>(260); }
    if (!pl->red_partial || !pl->red_result) return fail(FG_ERR_HIP, "out of device memory");
    pl->dist_pending = (order == 2);
    pl->finalized = true; pl->fused = true;
  }
  return pl->nx;
}

// Default capacities first; an attempt that outgrew one (coarse -> very fine grids: a source cell with thousands of
// candidates) drops everything it allocated and is repeated with the sizes its counters report -- at most twice, since the
// candidate counts of an attempt whose bin table was cut short mean nothing.  fg_set_search_mode(1) /
// FREGRID_HIP_EXACT_SEARCH=1 starts from empty buffers, i.e. sizes everything by counting (tests; memory-tight callers).
static int g_search_exact = 0;
extern "C" void fg_set_search_mode(int exact) { g_search_exact = exact; }
// 1: source cells whose latitude range cannot meet the destination grid get no record (a rank of a banded multi-GPU job sees
// a fraction of the source cells); fg_plan_get_cell_area then returns 0 for them.  Results are unchanged.
extern "C" void fg_set_search_cull(int on) { g_search_cull = on ? 1 : 0; }
extern "C" void fg_set_search_finalize(int on) { g_search_finalize = on ? 1 : 0; }
static long plan_search(fg_plan *pl, const double *const *d_lon_in, const double *const *d_lat_in,
                        const double *const *d_mask_in, const double *d_lon_out, const double *d_lat_out,
                        double mean_dlat, double mean_dlon, const GcXyz *gc_in = nullptr, const GcXyz *gc_out = nullptr,
                        const BoxMode *boxm = nullptr, int rect_hint = 1)
{
  static const bool env_exact = getenv("FREGRID_HIP_EXACT_SEARCH") && atoi(getenv("FREGRID_HIP_EXACT_SEARCH")) != 0;
  const bool exact = g_search_exact || env_exact;
  const long big = std::max((long)pl->nsrc, (long)pl->ndst);
  // 32-bit prefixes (bin starts, exchange-cell offsets, CSR rows): the bin table holds up to 3 records per destination cell
  // (+ the wide lists) and the pair list 8 per cell, so grids beyond 2^28 cells could wrap an int before the host sees a
  // counter -- refused up front rather than guarded in every kernel
  if (big > (1L << 28)) return fail(FG_ERR_CAPACITY, "grid of %ld cells: more than 2^28 cells per search are not supported", big);
  SearchCaps caps;
  caps.entries = exact ? 0ull : 3ull * (unsigned long long)pl->ndst + 4096ull;
  const unsigned long long cap_pairs = std::min<unsigned long long>(8ull * (unsigned long long)big + 65536ull, 2000000000ull);
  // regions of the pair list: a run of 256 source cells appends to one region; at least four such runs per region on average, so
  // that small grids do not pile their pairs into a few small regions
  caps.nreg = (int)std::max(1L, std::min((long)FG_NREG, ((long)pl->nsrc + 1023) / 1024));
  caps.regcap = exact ? 0 : (int)((cap_pairs / caps.nreg + FG_REG_ALIGN - 1) / FG_REG_ALIGN * FG_REG_ALIGN);
  caps.rect = g_search_rect && rect_hint && !gc_in && !boxm && pl->nx_out <= 8192 && pl->nx_out >= 2;
  const size_t keep = pl->owned.size();              // blocks the caller staged before the search stay
  long rc = FG_RETRY;
  int attempts = 0;
  for (; attempts < 4 && rc == FG_RETRY; attempts++) {
    if (attempts) {
      (void)hipStreamSynchronize(pl->stream);
      while (pl->owned.size() > keep) { void *p = pl->owned.back(); pl->owned.pop_back(); g_pool.put(p); }
      pl->tiles_dev = nullptr; pl->mask_dev = nullptr; pl->S = FgCells{}; pl->D = FgCells{};
      pl->x_src = pl->x_dst = nullptr; pl->x_area = pl->x_c1 = pl->x_c2 = nullptr; pl->xoff = nullptr; pl->x_rowpos = nullptr;
      pl->perm = nullptr; pl->csr.row_ptr = nullptr; pl->csr.e1 = nullptr; pl->csr.e2 = nullptr; pl->cen = nullptr; pl->src_idx_f = nullptr; pl->sums = nullptr;
      pl->have_geom = false; pl->rect = false; pl->rect_tab = FgRect{};
    }
    rc = plan_search_core(pl, d_lon_in, d_lat_in, d_mask_in, d_lon_out, d_lat_out, mean_dlat, mean_dlon, gc_in, gc_out, boxm, &caps);
  }
  if (rc == FG_RETRY) return fail(FG_ERR_CAPACITY, "exchange-grid search: capacities did not settle");
  if (rc >= 0) pl->stats[FG_STAT_EXACT] = (attempts > 1) ? 1 : 0;
  return rc;
}

// mean cell extents from a strided sample of corner arrays (host or device-copied-to-host)
static void sample_extents(int nx, int ny, const double *lon, const double *lat, double *mdlat, double *mdlon)
{
  const double PI = 3.14159265358979323846;
  long ncell = (long)nx * ny;
  long step = ncell / 4096; if (step < 1) step = 1;
  double sl = 0, sw = 0; long cnt = 0;
  for (long c = 0; c < ncell; c += step) {
    int i = (int)(c % nx), j = (int)(c / nx);
    long n0 = (long)j * (nx + 1) + i, n1 = n0 + 1, n3 = n0 + nx + 1, n2 = n3 + 1;
    double y[4] = {lat[n0], lat[n1], lat[n2], lat[n3]}, x[4] = {lon[n0], lon[n1], lon[n2], lon[n3]};
    double ymin = y[0], ymax = y[0], w = 0;
    for (int k = 1; k < 4; k++) { if (y[k] < ymin) ymin = y[k]; if (y[k] > ymax) ymax = y[k]; }
    for (int k = 0; k < 4; k++) {
      double d = fabs(x[(k + 1) & 3] - x[k]);              // |remainder(dx, 2 pi)|: the subtraction is exact for pi <= |dx| <= 4 pi (Sterbenz),
      if (d > 3.0 * PI) d = fabs(remainder(d, 2.0 * PI)); else if (d > PI) d = fabs(2.0 * PI - d);   // so the same bits as the library call,
      if (d > w) w = d;                                    // which took 0.12 ms per plan for the 16 k differences of the sample
    }
    if (w > PI / 2) continue;      // polar caps: not representative
    sl += ymax - ymin; sw += w; cnt++;
  }
  if (cnt == 0) { *mdlat = PI / 180; *mdlon = PI / 180; return; }
  *mdlat = sl / cnt; *mdlon = sw / cnt;
  if (*mdlat < 1e-7) *mdlat = 1e-7;
  if (*mdlon < 1e-7) *mdlon = 1e-7;
}

// The same estimate for corner arrays that live on the device: the sampled cells' corners are gathered into a compact buffer
// (<= 4096 cells), copied and put through the host routine above -- a quarter of a megabyte instead of the whole grid (the whole
// 0.25-degree target copied into a fresh std::vector took 29 ms, thirty times the search it was preparing).
__global__ void k_sample_corners(int nx, int ny, long step, int nsamp, int narr, const double *a0, const double *a1, const double *a2, double *out)
{
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= nsamp) return;
  const long c = (long)k * step;
  const int i = (int)(c % nx), j = (int)(c / nx);
  const long n0 = (long)j * (nx + 1) + i, n1 = n0 + 1, n3 = n0 + nx + 1, n2 = n3 + 1;
  const double *arr[3] = {a0, a1, a2};
  for (int a = 0; a < narr; a++) {
    double *o = out + ((size_t)a * nsamp + k) * 4;
    o[0] = arr[a][n0]; o[1] = arr[a][n1]; o[2] = arr[a][n2]; o[3] = arr[a][n3];
  }
}
// (the buffer holds, per array, the four corners SW, SE, NE, NW of every sampled cell)
// *curvilinear = 1 if a sampled cell's west / east corners differ in longitude or its south / north corners in latitude (bit
// patterns, the test k_rect_tables applies to the whole grid): such a target cannot take the rectilinear path, and the search does
// not spend an attempt on finding that out (a cubed-sphere tile as target: 0.2 ms of kernels and a read-back per plan)
static int sample_extents_dev(fg_plan *pl, int nx, int ny, const double *d_lon, const double *d_lat, double *mdlat, double *mdlon,
                              int *curvilinear)
{
  *curvilinear = 0;
  const double PI = 3.14159265358979323846;
  const long ncell = (long)nx * ny;
  long step = ncell / 4096; if (step < 1) step = 1;
  const int nsamp = (int)((ncell + step - 1) / step);
  double *d_q = (double *)g_pool.get(pl->device, (size_t)nsamp * 8 * sizeof(double));
  if (!d_q) return fail(FG_ERR_HIP, "out of device memory");
  std::vector<double> q((size_t)nsamp * 8);
  k_sample_corners<<<(nsamp + 255) / 256, 256, 0, pl->stream>>>(nx, ny, step, nsamp, 2, d_lon, d_lat, nullptr, d_q);
  hipError_t e = hipMemcpyAsync(q.data(), d_q, q.size() * sizeof(double), hipMemcpyDeviceToHost, pl->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(pl->stream);
  g_pool.put(d_q);
  if (e != hipSuccess) return fail(FG_ERR_HIP, "sampling the destination corners failed: %s", hipGetErrorString(e));
  double sl = 0, sw = 0; long cnt = 0;
  for (int k = 0; k < nsamp; k++) {
    const double *x = &q[(size_t)k * 4], *y = &q[((size_t)nsamp + k) * 4];
    if (memcmp(&x[0], &x[3], 8) || memcmp(&x[1], &x[2], 8) || memcmp(&y[0], &y[1], 8) || memcmp(&y[3], &y[2], 8)) *curvilinear = 1;
    double ymin = y[0], ymax = y[0], w = 0;
    for (int m = 1; m < 4; m++) { if (y[m] < ymin) ymin = y[m]; if (y[m] > ymax) ymax = y[m]; }
    for (int m = 0; m < 4; m++) {                         // |remainder(dx, 2 pi)| without the library call where |dx| <= 3 pi (16 k calls per plan)
      double d = fabs(x[(m + 1) & 3] - x[m]);
      if (d > 3.0 * PI) d = fabs(remainder(d, 2.0 * PI)); else if (d > PI) d = fabs(2.0 * PI - d);
      if (d > w) w = d;
    }
    if (w > PI / 2) continue;      // polar caps: not representative
    sl += ymax - ymin; sw += w; cnt++;
  }
  if (cnt == 0) { *mdlat = PI / 180; *mdlon = PI / 180; return 0; }
  *mdlat = sl / cnt; *mdlon = sw / cnt;
  if (*mdlat < 1e-7) *mdlat = 1e-7;
  if (*mdlon < 1e-7) *mdlon = 1e-7;
  return 0;
}
static int sample_extents_xyz_dev(fg_plan *pl, int nx, int ny, const double *d_x, const double *d_y, const double *d_z, double *mdlat, double *mdlon)
{
  const double PI = 3.14159265358979323846;
  const long ncell = (long)nx * ny;
  long step = ncell / 4096; if (step < 1) step = 1;
  const int nsamp = (int)((ncell + step - 1) / step);
  double *d_q = (double *)g_pool.get(pl->device, (size_t)nsamp * 12 * sizeof(double));
  if (!d_q) return fail(FG_ERR_HIP, "out of device memory");
  std::vector<double> q((size_t)nsamp * 12);
  k_sample_corners<<<(nsamp + 255) / 256, 256, 0, pl->stream>>>(nx, ny, step, nsamp, 3, d_x, d_y, d_z, d_q);
  hipError_t e = hipMemcpyAsync(q.data(), d_q, q.size() * sizeof(double), hipMemcpyDeviceToHost, pl->stream);
  if (e == hipSuccess) e = hipStreamSynchronize(pl->stream);
  g_pool.put(d_q);
  if (e != hipSuccess) return fail(FG_ERR_HIP, "sampling the destination corners failed: %s", hipGetErrorString(e));
  double sl = 0, sw = 0; long cnt = 0;
  for (int k = 0; k < nsamp; k++) {                        // (corner 0 and the opposite corner 2, as sample_extents_xyz)
    const double *x = &q[(size_t)k * 4], *y = &q[((size_t)nsamp + k) * 4], *z = &q[((size_t)2 * nsamp + k) * 4];
    const double dx = x[0] - x[2], dy = y[0] - y[2], dz = z[0] - z[2];
    const double diag = sqrt(dx * dx + dy * dy + dz * dz);
    const double zc = 0.5 * (z[0] + z[2]);
    const double coslat = sqrt(fmax(0.0, 1.0 - zc * zc));
    if (coslat < 0.2) continue;
    sl += diag; sw += diag / coslat; cnt++;
  }
  if (cnt == 0) { *mdlat = PI / 180; *mdlon = PI / 180; return 0; }
  *mdlat = sl / cnt; *mdlon = sw / cnt;
  if (*mdlat < 1e-7) *mdlat = 1e-7;
  if (*mdlon < 1e-7) *mdlon = 1e-7;
  return 0;
}

extern "C" long fg_plan_create(int order, int ntiles_in, const int *nx_in, const int *ny_in,
                               const double *const *lon_in, const double *const *lat_in,
                               const double *const *mask_in,
                               int nx_out, int ny_out, const double *lon_out, const double *lat_out,
                               int device, fg_plan **plan_out)
{
  if (!lon_in || !lat_in || !lon_out || !lat_out) return fail(FG_ERR_ARG, "null grid pointer");
  fg_plan *pl = nullptr;
  int rc = plan_base(order, ntiles_in, nx_in, ny_in, nx_out, ny_out, device, &pl);
  if (rc) return rc;
  std::vector<const double *> dlon(ntiles_in), dlat(ntiles_in), dmask(ntiles_in, nullptr);
  std::vector<void *> staged;
  std::vector<XferJob> up_jobs;
  auto up = [&](const double *h, size_t n) -> const double * {
    double *d = pl->alloc<double>(n);
    if (!d) return nullptr;
    up_jobs.push_back(XferJob{(void *)h, d, n * sizeof(double), false});
    staged.push_back(d);
    return d;
  };
  bool ok = true;
  for (int m = 0; m < ntiles_in && ok; m++) {
    size_t np = (size_t)(nx_in[m] + 1) * (ny_in[m] + 1);
    dlon[m] = up(lon_in[m], np); dlat[m] = up(lat_in[m], np);
    if (mask_in && mask_in[m]) { dmask[m] = up(mask_in[m], (size_t)nx_in[m] * ny_in[m]); ok = ok && dmask[m]; }
    ok = ok && dlon[m] && dlat[m];
  }
  size_t npo = (size_t)(nx_out + 1) * (ny_out + 1);
  const double *dlo = ok ? up(lon_out, npo) : nullptr, *dla = ok ? up(lat_out, npo) : nullptr;
  if (!ok || !dlo || !dla || !xfer_pool().run(device, up_jobs)) { fg_plan_destroy(pl); return fail(FG_ERR_HIP, "grid upload failed (out of device memory?)"); }
  double mdlat, mdlon;
  sample_extents(nx_out, ny_out, lon_out, lat_out, &mdlat, &mdlon);
  long nx = plan_search(pl, dlon.data(), dlat.data(), mask_in ? dmask.data() : nullptr, dlo, dla, mdlat, mdlon, nullptr, nullptr, nullptr,
                        host_says_not_rect(nx_out, ny_out, lon_out, lat_out) ? 0 : 1);
  if (nx < 0) { fg_plan_destroy(pl); return nx; }
  for (void *p : staged) pl->release(p);
  *plan_out = pl;
  return nx;
}

// Same as fg_plan_create but every grid pointer is a DEVICE pointer (inputs already in HBM).
// mean_dlat/mean_dlon: typical destination cell extent in radians (<=0: derive by copying a
// strided sample of the destination corners to the host).
extern "C" long fg_plan_create_dev(int order, int ntiles_in, const int *nx_in, const int *ny_in,
                                   const double *const *d_lon_in, const double *const *d_lat_in,
                                   const double *const *d_mask_in,
                                   int nx_out, int ny_out, const double *d_lon_out, const double *d_lat_out,
                                   double mean_dlat, double mean_dlon, int device, void *stream,
                                   int use_caller_stream, fg_plan **plan_out)
{
  if (!d_lon_in || !d_lat_in || !d_lon_out || !d_lat_out) return fail(FG_ERR_ARG, "null grid pointer");
  fg_plan *pl = nullptr;
  int rc = plan_base(order, ntiles_in, nx_in, ny_in, nx_out, ny_out, device, &pl);
  if (rc) return rc;
  if (use_caller_stream) {
    g_handles.put_stream(pl->device, pl->stream);
    pl->stream = (hipStream_t)stream; pl->own_stream = false;
  }
  int curvilinear = 0;
  if (!(mean_dlat > 0) || !(mean_dlon > 0)) {
    if (hipSetDevice(pl->device) != hipSuccess || sample_extents_dev(pl, nx_out, ny_out, d_lon_out, d_lat_out, &mean_dlat, &mean_dlon, &curvilinear)) {
      fg_plan_destroy(pl); return FG_ERR_HIP;
    }
  }
  long nx = plan_search(pl, d_lon_in, d_lat_in, d_mask_in, d_lon_out, d_lat_out, mean_dlat, mean_dlon, nullptr, nullptr, nullptr, !curvilinear);
  if (nx < 0) { fg_plan_destroy(pl); return nx; }
  *plan_out = pl;
  return nx;
}

// ------------------------------------------------------------------------------------- great-circle plans
// typical destination cap-box extent from a strided sample of unit-vector corners (host copies)
static void sample_extents_xyz(int nx, int ny, const double *x, const double *y, const double *z, double *mdlat, double *mdlon)
{
  const double PI = 3.14159265358979323846;
  long ncell = (long)nx * ny;
  long step = ncell / 4096; if (step < 1) step = 1;
  double sl = 0, sw = 0; long cnt = 0;
  for (long c = 0; c < ncell; c += step) {
    int i = (int)(c % nx), j = (int)(c / nx);
    long n0 = (long)j * (nx + 1) + i, n2 = n0 + nx + 2;
    double dx = x[n0] - x[n2], dy = y[n0] - y[n2], dz = z[n0] - z[n2];
    double diag = sqrt(dx * dx + dy * dy + dz * dz);
    double zc = 0.5 * (z[n0] + z[n2]);
    double coslat = sqrt(fmax(0.0, 1.0 - zc * zc));
    if (coslat < 0.2) continue;
    sl += diag; sw += diag / coslat; cnt++;
  }
  if (cnt == 0) { *mdlat = PI / 180; *mdlon = PI / 180; return; }
  *mdlat = sl / cnt; *mdlon = sw / cnt;
  if (*mdlat < 1e-7) *mdlat = 1e-7;
  if (*mdlon < 1e-7) *mdlon = 1e-7;
}

extern "C" long fg_plan_create_great_circle_dev(int ntiles_in, const int *nx_in, const int *ny_in,
                                                const double *const *d_x_in, const double *const *d_y_in, const double *const *d_z_in,
                                                const double *const *d_mask_in, int nx_out, int ny_out,
                                                const double *d_x_out, const double *d_y_out, const double *d_z_out,
                                                double mean_dlat, double mean_dlon, int device, void *stream, int use_caller_stream,
                                                fg_plan **plan_out)
{
  if (!d_x_in || !d_y_in || !d_z_in || !d_x_out || !d_y_out || !d_z_out) return fail(FG_ERR_ARG, "null grid pointer");
  fg_plan *pl = nullptr;
  int rc = plan_base(FG_CONSERVE_ORDER1, ntiles_in, nx_in, ny_in, nx_out, ny_out, device, &pl);
  if (rc) return rc;
  if (use_caller_stream) {
    g_handles.put_stream(pl->device, pl->stream);
    pl->stream = (hipStream_t)stream; pl->own_stream = false;
  }
  if (!(mean_dlat > 0) || !(mean_dlon > 0)) {
    if (hipSetDevice(pl->device) != hipSuccess || sample_extents_xyz_dev(pl, nx_out, ny_out, d_x_out, d_y_out, d_z_out, &mean_dlat, &mean_dlon)) {
      fg_plan_destroy(pl); return FG_ERR_HIP;
    }
  }
  std::vector<GcXyz> gin(ntiles_in);
  for (int m = 0; m < ntiles_in; m++) gin[m] = GcXyz{d_x_in[m], d_y_in[m], d_z_in[m]};
  GcXyz gout{d_x_out, d_y_out, d_z_out};
  long nx = plan_search(pl, nullptr, nullptr, d_mask_in, nullptr, nullptr, mean_dlat, mean_dlon, gin.data(), &gout);
  if (nx < 0) { fg_plan_destroy(pl); return nx; }
  *plan_out = pl;
  return nx;
}

extern "C" long fg_plan_create_great_circle(int ntiles_in, const int *nx_in, const int *ny_in,
                                            const double *const *lon_in, const double *const *lat_in, const double *const *mask_in,
                                            int nx_out, int ny_out, const double *lon_out, const double *lat_out,
                                            int device, fg_plan **plan_out)
{
  if (!lon_in || !lat_in || !lon_out || !lat_out || !nx_in || !ny_in) return fail(FG_ERR_ARG, "null grid pointer");
  if (ntiles_in < 1 || nx_out < 1 || ny_out < 1) return fail(FG_ERR_ARG, "bad grid sizes");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(FG_ERR_HIP, "no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  if (device < 0 || device >= ndev) return fail(FG_ERR_ARG, "device %d out of range (0..%d)", device, ndev - 1);
  HIPCHK(hipSetDevice(device));
  // unit vectors on the host (libm, as the reference), then one upload per array
  std::vector<double *> dev_ptrs;
  auto cleanup = [&]() { for (double *p : dev_ptrs) g_pool.put(p); };
  auto to_dev_xyz = [&](int nx, int ny, const double *lon, const double *lat, GcXyz *out) -> bool {
    const size_t np = (size_t)(nx + 1) * (ny + 1);
    std::vector<double> h(3 * np);
    fg_latlon2xyz((long)np, lon, lat, h.data(), h.data() + np, h.data() + 2 * np);
    double *d = (double *)g_pool.get(device, 3 * np * sizeof(double));
    if (!d) return false;
    dev_ptrs.push_back(d);
    if (hipMemcpy(d, h.data(), 3 * np * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) return false;
    out->x = d; out->y = d + np; out->z = d + 2 * np;
    return true;
  };
  std::vector<GcXyz> gin(ntiles_in);
  std::vector<const double *> xs(ntiles_in), ys(ntiles_in), zs(ntiles_in), dmask(ntiles_in, nullptr);
  bool ok = true;
  for (int m = 0; m < ntiles_in && ok; m++) {
    if (nx_in[m] < 1 || ny_in[m] < 1) { cleanup(); return fail(FG_ERR_ARG, "bad source tile size"); }
    ok = to_dev_xyz(nx_in[m], ny_in[m], lon_in[m], lat_in[m], &gin[m]);
    xs[m] = gin[m].x; ys[m] = gin[m].y; zs[m] = gin[m].z;
    if (ok && mask_in && mask_in[m]) {
      const size_t nc = (size_t)nx_in[m] * ny_in[m];
      double *d = (double *)g_pool.get(device, nc * sizeof(double));
      ok = d && hipMemcpy(d, mask_in[m], nc * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
      if (d) dev_ptrs.push_back(d);
      dmask[m] = d;
    }
  }
  GcXyz gout{};
  ok = ok && to_dev_xyz(nx_out, ny_out, lon_out, lat_out, &gout);
  if (!ok) { cleanup(); return fail(FG_ERR_HIP, "grid upload failed (out of device memory?)"); }
  double mdlat, mdlon;
  sample_extents(nx_out, ny_out, lon_out, lat_out, &mdlat, &mdlon);
  long nx = fg_plan_create_great_circle_dev(ntiles_in, nx_in, ny_in, xs.data(), ys.data(), zs.data(), mask_in ? dmask.data() : nullptr,
                                            nx_out, ny_out, gout.x, gout.y, gout.z, mdlat, mdlon, device, nullptr, 0, plan_out);
  cleanup();
  return nx;
}

// A search whose SOURCE cells are a list of polygons (<= 8 vertices each), e.g. the atmosphere x land cells of a coupler mosaic
// against the ocean grid (make_coupler_mosaic.c:1659-1692: clip_2dx2d(atmxlnd polygon, ocean cell)).  Host arrays: n[npoly],
// lon / lat [npoly][8] in the longitude frame of the polygon's parent cell, lon_avg[npoly] = that cell's mean longitude (decides
// the +-2pi shift of a destination cell and is poly_ctrlon's reference longitude), area_ref[npoly] = the area the 1e-6 ratio
// test compares with (together with the destination cell's).  Exchange cells: i_in = polygon index, j_in = t_in = 0.
extern "C" long fg_plan_create_polylist(int order, int npoly, const int *n, const double *lon, const double *lat, const double *lon_avg,
                                        const double *area_ref, int nx_out, int ny_out, const double *lon_out, const double *lat_out,
                                        int device, fg_plan **plan_out)
{
  if (npoly < 1 || !n || !lon || !lat || !lon_avg || !area_ref || !lon_out || !lat_out) return fail(FG_ERR_ARG, "null argument");
  fg_plan *pl = nullptr;
  const int one = 1;
  int rc = plan_base(order, 1, &npoly, &one, nx_out, ny_out, device, &pl);
  if (rc) return rc;
  if (order == 2) pl->f_stride = npoly;                  // (no halo: the "field" of a polygon list is one value per polygon)
  const size_t npo = (size_t)(nx_out + 1) * (ny_out + 1);
  int *d_n = pl->alloc<int>(npoly);
  double *d_lon = pl->alloc<double>((size_t)npoly * 8), *d_lat = pl->alloc<double>((size_t)npoly * 8);
  double *d_avg = pl->alloc<double>(npoly), *d_area = pl->alloc<double>(npoly), *d_lo = pl->alloc<double>(npo), *d_la = pl->alloc<double>(npo);
  std::vector<XferJob> jobs = {{(void *)n, d_n, (size_t)npoly * sizeof(int), false}, {(void *)lon, d_lon, (size_t)npoly * 64, false},
                               {(void *)lat, d_lat, (size_t)npoly * 64, false}, {(void *)lon_avg, d_avg, (size_t)npoly * 8, false},
                               {(void *)area_ref, d_area, (size_t)npoly * 8, false}, {(void *)lon_out, d_lo, npo * 8, false}, {(void *)lat_out, d_la, npo * 8, false}};
  if (!d_n || !d_lon || !d_lat || !d_avg || !d_area || !d_lo || !d_la || !xfer_pool().run(device, jobs)) {
    fg_plan_destroy(pl); return fail(FG_ERR_HIP, "polygon upload failed (out of device memory?)");
  }
  pl->polys = FgPolyList{d_n, d_lon, d_lat, d_avg, d_area, npoly};
  double mdlat, mdlon;
  sample_extents(nx_out, ny_out, lon_out, lat_out, &mdlat, &mdlon);
  const double *none[1] = {nullptr};
  long nx = plan_search(pl, none, none, nullptr, d_lo, d_la, mdlat, mdlon, nullptr, nullptr, nullptr,
                        host_says_not_rect(nx_out, ny_out, lon_out, lat_out) ? 0 : 1);
  if (nx < 0) { fg_plan_destroy(pl); return nx; }
  pl->release(d_lo); pl->release(d_la);
  *plan_out = pl;
  return nx;
}

extern "C" int fg_plan_set_stream(fg_plan *pl, void *stream)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  if (pl->own_stream && pl->stream) g_handles.put_stream(pl->device, pl->stream);
  pl->stream = (hipStream_t)stream; pl->own_stream = false;
  return 0;
}

// Internal (sweep.hip): run the plan's launches on `stream` for the duration of one call and hand the previous stream back
// afterwards.  borrow: waits for the plan's own stream, then points it at `stream`; *saved / *saved_own receive what it had.
// give back: the caller has synchronised `stream`.  The plan never owns a borrowed stream.
int fg_plan_borrow_stream(fg_plan *pl, void *stream, void **saved, int *saved_own)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  *saved = (void *)pl->stream; *saved_own = pl->own_stream ? 1 : 0;
  pl->stream = (hipStream_t)stream; pl->own_stream = false;
  return 0;
}
void fg_plan_return_stream(fg_plan *pl, void *saved, int saved_own)
{
  if (!pl) return;
  pl->stream = (hipStream_t)saved; pl->own_stream = saved_own != 0;
}

// a plan without a search: exchange cells are supplied later by fg_plan_set_xgrid
extern "C" int fg_plan_create_empty(int order, int ntiles_in, const int *nx_in, const int *ny_in,
                                    int nx_out, int ny_out, int device, fg_plan **plan_out)
{
  fg_plan *pl = nullptr;
  int rc = plan_base(order, ntiles_in, nx_in, ny_in, nx_out, ny_out, device, &pl);
  if (rc) return rc;
  std::vector<FgTile> th(pl->ntiles);
  for (int m = 0; m < pl->ntiles; m++) th[m] = FgTile{nullptr, nullptr, pl->nx_in[m], pl->ny_in[m], pl->cell_off[m]};
  pl->tiles_dev = pl->alloc<FgTile>(pl->ntiles + 1);
  if (!pl->tiles_dev || hipMemcpy(pl->tiles_dev, th.data(), sizeof(FgTile) * th.size(), hipMemcpyHostToDevice) != hipSuccess) {
    fg_plan_destroy(pl); return fail(FG_ERR_HIP, "out of device memory");
  }
  *plan_out = pl;
  return 0;
}

// The search sizes the exchange-cell arrays by capacity (8*max(nsrc, ndst) entries), about twice what a remap between grids of
// similar resolution fills.  A caller that keeps many plans (one per output tile or variable set) can give the surplus back:
// the arrays are re-allocated at nxgrid entries (device-to-device copies, ~0.1 ms at C384 -> 0.25 deg).
extern "C" int fg_plan_trim(fg_plan *pl)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  if (!pl->searched) return fail(FG_ERR_STATE, "fg_plan_trim: plan holds no exchange cells");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  const size_t n = (size_t)pl->nx + 1;
  auto shrink = [&](auto *&ptr) -> int {
    if (!ptr) return 0;
    typedef typename std::remove_reference<decltype(*ptr)>::type T;
    T *q = pl->alloc<T>(n);
    if (!q) return fail(FG_ERR_HIP, "out of device memory");
    if (hipMemcpyAsync(q, ptr, n * sizeof(T), hipMemcpyDeviceToDevice, pl->stream) != hipSuccess) return fail(FG_ERR_HIP, "device copy failed");
    void *old = ptr; ptr = q;
    HIPCHK(hipStreamSynchronize(pl->stream));
    pl->release(old);
    return 0;
  };
  int rc = 0;
  if ((rc = shrink(pl->x_src)) || (rc = shrink(pl->x_dst)) || (rc = shrink(pl->x_area)) || (rc = shrink(pl->x_c1)) || (rc = shrink(pl->x_c2))) return rc;
  if (pl->perm && (rc = shrink(pl->perm))) return rc;
  return 0;
}

extern "C" long fg_plan_nxgrid(const fg_plan *pl) { return pl ? pl->nx : FG_ERR_ARG; }
extern "C" long fg_plan_ncells_in(const fg_plan *pl) { return pl ? pl->nsrc : FG_ERR_ARG; }
extern "C" long fg_plan_ncells_out(const fg_plan *pl) { return pl ? pl->ndst : FG_ERR_ARG; }
extern "C" int fg_plan_order(const fg_plan *pl) { return pl ? pl->order : FG_ERR_ARG; }
extern "C" int fg_plan_device(const fg_plan *pl) { return pl ? pl->device : FG_ERR_ARG; }
extern "C" double *fg_plan_cell_sums_dev(fg_plan *pl) { return (pl && pl->order == 2) ? pl->sums : nullptr; }
extern "C" int fg_plan_copy_cell_sums(fg_plan *pl, double *dst_dev)
{
  if (!pl || !dst_dev) return fail(FG_ERR_ARG, "null argument");
  if (pl->order != 2 || !pl->sums) return fail(FG_ERR_STATE, "plan holds no order-2 cell sums");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipMemcpyAsync(dst_dev, pl->sums, 3 * (size_t)pl->nsrc * sizeof(double), hipMemcpyDeviceToDevice, pl->stream));
  HIPCHK(hipStreamSynchronize(pl->stream));
  return 0;
}
// total_dev[3][ncells_in] (sum of area, of the clon integral, of the clat integral per source cell) += this plan's exchange cells,
// added one by one in exchange-cell order onto what total_dev already holds; cells_dev (may be null = all) restricts it to a list
// of source cells.  Before fg_plan_finalize only.
static int accumulate_cell_sums(fg_plan *pl, double *total_dev, const int *cells_dev, int ncells, bool sync)
{
  if (!pl || !total_dev) return fail(FG_ERR_ARG, "null argument");
  if (pl->order != 2 || !pl->searched || pl->finalized || !pl->xoff)
    return fail(FG_ERR_STATE, "fg_plan_accumulate_cell_sums: needs a searched, not yet finalized order-2 plan");
  HIPCHK(hipSetDevice(pl->device));
  if (pl->nx > 0)
    fgd_accumulate_cell_sums(cells_dev ? ncells : pl->nsrc, cells_dev, pl->nsrc, pl->xoff, pl->x_area, pl->x_c1, pl->x_c2, total_dev, pl->stream);
  if (sync) HIPCHK(hipStreamSynchronize(pl->stream));
  HIPCHK(hipGetLastError());
  return 0;
}
extern "C" int fg_plan_accumulate_cell_sums(fg_plan *pl, double *total_dev, const int *cells_dev, int ncells)
{
  return accumulate_cell_sums(pl, total_dev, cells_dev, ncells, true);
}
// the same, only queued on the plan's stream: for callers that put the plan on the stream their other device work is ordered on
// (fg_plan_set_stream / the stream argument of fg_plan_create_dev), so that a rank-to-rank hand-over needs no host round trip
extern "C" int fg_plan_accumulate_cell_sums_async(fg_plan *pl, double *total_dev, const int *cells_dev, int ncells)
{
  return accumulate_cell_sums(pl, total_dev, cells_dev, ncells, false);
}
extern "C" void *fg_plan_stream(fg_plan *pl) { return pl ? (void *)pl->stream : nullptr; }
extern "C" int fg_plan_sync(fg_plan *pl)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  HIPCHK(hipStreamSynchronize(pl->stream));
  return 0;
}
extern "C" int fg_plan_stats(const fg_plan *pl, long *stats, int n)
{
  if (!pl || !stats) return fail(FG_ERR_ARG, "null argument");
  for (int k = 0; k < n && k < FG_NSTATS; k++) stats[k] = pl->stats[k];
  return 0;
}

// Destination-row CSR.  A searched plan arrives with row_ptr and perm (rows grouped, unsorted inside a row) already built on
// the device by the search; a plan loaded from a remap file (fg_plan_set_xgrid) counts and fills its rows here.
// cen != null: order 2, x_c1/x_c2 hold the centroid integrals and di/dj are formed while the records are packed.
static int build_csr(fg_plan *pl, const double *cen)
{
  hipStream_t st = pl->stream;
  const int ndst = pl->ndst;
  const long nx = pl->nx;
  HIPCHK(hipSetDevice(pl->device));
  if (!pl->src_idx_f) {
    pl->src_idx_f = pl->alloc<int>(pl->nsrc + 1);
    if (!pl->src_idx_f) return fail(FG_ERR_HIP, "out of device memory");
    fgd_src_field_index(pl->order, pl->tiles_dev, pl->ntiles, pl->nsrc, pl->src_idx_f, st);
  }
  void *scratch = nullptr;
  if (!pl->rows_built) {
    const long t_rows = fgd_scan_tiles(ndst);
    const size_t zb = 256 + (size_t)t_rows * sizeof(unsigned long long) + 2 * ((size_t)ndst + 1) * sizeof(int);
    char *z = pl->alloc<char>(zb);                                    // [ticket, total, err | look-back words | counts | fill cursors]
    pl->perm = pl->alloc<int>(nx + 1);
    pl->csr.row_ptr = pl->alloc<int>(ndst + 1);
    if (!z || !pl->perm || !pl->csr.row_ptr) return fail(FG_ERR_HIP, "out of device memory");
    HIPCHK(hipMemsetAsync(z, 0, zb, st));
    unsigned long long *lb = (unsigned long long *)(z + 256);
    int *row_cnt = (int *)(z + 256 + (size_t)t_rows * sizeof(unsigned long long));
    fgd_csr_count(nx, pl->x_dst, row_cnt, st);
    fgd_exclusive_scan1(row_cnt, ndst, pl->csr.row_ptr, lb, (unsigned *)z, (unsigned long long *)(z + 8), (unsigned *)(z + 16), st);
    fgd_csr_fill(nx, pl->x_dst, pl->csr.row_ptr, row_cnt + ndst + 1, pl->perm, st);
    scratch = z;
  }
  if (pl->order == 2) pl->csr.e2 = pl->alloc<FgCsrEntry2>(nx + 1);
  else pl->csr.e1 = pl->alloc<FgCsrEntry1>(nx + 1);
  if (!pl->csr.e1 && !pl->csr.e2) return fail(FG_ERR_HIP, "out of device memory");
  // rows longer than the kernel's LDS staging are sorted through this -- also in plans whose MEAN row is short: the cells around a
  // pole of a cubed-sphere target hold a whole row of a lat-lon source each (1 440 exchange cells: a lane sorting that by insertion
  // took 17 ms)
  int *sg_tmp = pl->alloc<int>(2 * (size_t)(nx + 1));
  // a curvilinear target with cells that went to the general clip (pole vertices): its rows round the pole are long -- hundreds of
  // exchange cells -- under a mean of 6.6, and the 64-rows-per-wave mode ranks them one after another (0.25 deg -> C384 tile 3:
  // 0.23 ms of the tile's 1.1); the 16-rows-per-block mode costs the short rows of such a tile little
  const int long_rows = !pl->rect && !pl->great_circle && pl->stats[FG_STAT_DEFERRED] > 0 && nx > 4 * (long)ndst;
  fgd_csr_sortgather(pl->order, ndst, nx, pl->perm, pl->x_src, pl->x_area, pl->x_c1, pl->x_c2, pl->src_idx_f, cen, pl->nsrc, pl->csr, st,
                     sg_tmp, nx + 1, long_rows);
  HIPCHK(hipStreamSynchronize(st));
  HIPCHK(hipGetLastError());
  pl->release(scratch); pl->release(pl->perm); pl->perm = nullptr; pl->release(sg_tmp);
  if (!pl->red_partial) { pl->red_partial = pl->alloc<double>(1024); pl->red_result = pl->alloc<double>(260); }
  if (!pl->red_partial || !pl->red_result) return fail(FG_ERR_HIP, "out of device memory");
  return 0;
}

extern "C" int fg_plan_finalize(fg_plan *pl, const double *total_cell_sums_dev)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  if (!pl->searched) return fail(FG_ERR_STATE, "fg_plan_finalize: plan holds no search result");
  if (pl->finalized && pl->fused && !total_cell_sums_dev) return 0;         // its search did it (fg_set_search_finalize)
  if (pl->finalized) return fail(FG_ERR_STATE, pl->fused ? "fg_plan_finalize: the plan was finalized by its search (fg_set_search_finalize) from its own "
                                                          "per-cell sums; totals over several plans need the two-step sequence"
                                                         : "fg_plan_finalize: already finalized");
  HIPCHK(hipSetDevice(pl->device));
  PhaseTimer pt; pt.start(g_profiling != 0, pl->stream);
  pt.begin(PH_FINALIZE);
  if (pl->order == 2) {
    pl->cen = pl->alloc<double>(2 * (size_t)pl->nsrc);
    if (!pl->cen) return fail(FG_ERR_HIP, "out of device memory");
    const double *tot = total_cell_sums_dev ? total_cell_sums_dev : pl->sums;
    fgd_centroids(pl->nsrc, pl->S, tot, pl->cen, pl->stream);
    pl->dist_pending = true;                 // the CSR records get di/dj now; the exchange-cell arrays when somebody asks for them
  }
  int rc = build_csr(pl, pl->order == 2 ? pl->cen : nullptr);
  if (rc) return rc;
  pt.end();
  HIPCHK(hipStreamSynchronize(pl->stream));
  pl->phase_ms[PH_FINALIZE] = 0; pt.collect(pl->phase_ms);
  pl->finalized = true;
  return 0;
}

// phase times (ms) of the last search/finalize/apply when fg_set_profiling(1) was active:
// [0] cell records [1] binning [2] candidates [3] clip quad [4] clip general [5] compaction
// [6] destination rows [7] whole search (device span) [8] finalize [9] last apply
extern "C" int fg_plan_phase_ms(fg_plan *pl, float *ms, int n)
{
  if (!pl || !ms) return fail(FG_ERR_ARG, "null argument");
  if (pl->apply_spans > 0) {               // sweeps are timed without a sync; average them now
    HIPCHK(hipSetDevice(pl->device));
    HIPCHK(hipStreamSynchronize(pl->stream));
    float acc[PH_COUNT] = {0};
    pl->apply_pt.collect(acc);
    pl->phase_ms[PH_APPLY] = acc[PH_APPLY] / pl->apply_spans;
    pl->apply_spans = 0;
  }
  for (int k = 0; k < n && k < PH_COUNT; k++) ms[k] = pl->phase_ms[k];
  return 0;
}

extern "C" int fg_plan_get_xgrid(const fg_plan *pl, int *t_in, int *i_in, int *j_in, int *i_out, int *j_out,
                                 double *area, double *c1, double *c2)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  if (!pl->searched) return fail(FG_ERR_STATE, "plan holds no exchange cells");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  const long nx = pl->nx;
  if (nx == 0) return 0;
  if (pl->dist_pending) {                     // di = clon/area - centroid (conserve_interp.c:256-257,355-356), on first use
    fg_plan *m = const_cast<fg_plan *>(pl);
    fgd_distances(nx, pl->nsrc, pl->x_src, pl->x_area, pl->cen, m->x_c1, m->x_c2, pl->stream);
    HIPCHK(hipStreamSynchronize(pl->stream));
    m->dist_pending = false;
  }
  std::vector<XferJob> jobs;
  int *idx = nullptr;
  struct PutBack { int *&p; ~PutBack() { g_pool.put(p); } } put_back{idx};
  if (t_in || i_in || j_in || i_out || j_out) {
    // index decomposition on the device (the host loop with two divisions per exchange cell took longer than the transfer)
    idx = (int *)g_pool.get(pl->device, 5 * (size_t)nx * sizeof(int));
    if (!idx) return fail(FG_ERR_HIP, "out of device memory");
    fgd_xgrid_indices(nx, pl->x_src, pl->x_dst, pl->tiles_dev, pl->ntiles, pl->nx_out, idx, idx + nx, idx + 2 * nx, idx + 3 * nx,
                      idx + 4 * nx, pl->stream);
    HIPCHK(hipStreamSynchronize(pl->stream));
    int *dst[5] = {t_in, i_in, j_in, i_out, j_out};
    for (int q = 0; q < 5; q++) if (dst[q]) jobs.push_back(XferJob{dst[q], idx + (size_t)q * nx, (size_t)nx * sizeof(int), true});
  }
  if (area) jobs.push_back(XferJob{area, pl->x_area, (size_t)nx * sizeof(double), true});
  if (pl->order == 2) {
    if (c1) jobs.push_back(XferJob{c1, pl->x_c1, (size_t)nx * sizeof(double), true});
    if (c2) jobs.push_back(XferJob{c2, pl->x_c2, (size_t)nx * sizeof(double), true});
  }
  if (!xfer_pool().run(pl->device, jobs)) return fail(FG_ERR_HIP, "fg_plan_get_xgrid: device -> host copy failed");
  return 0;
}

// The clipped polygon of every exchange cell (canonical order): n_out[nx] vertex counts and [nx][maxv] vertex arrays -- legacy
// plans: v0 = lon, v1 = lat (v2 unused, may be null); great-circle plans: v0, v1, v2 = x, y, z.  Host arrays.  What clip_2dx2d /
// clip_2dx2d_great_circle returned for the pair inside create_xgrid (make_coupler_mosaic.c:1560-1577 keeps exactly these).
extern "C" int fg_plan_get_polygons(const fg_plan *pl, int maxv, int *n_out, double *v0, double *v1, double *v2)
{
  if (!pl || !n_out || !v0 || !v1) return fail(FG_ERR_ARG, "null argument");
  if (!pl->searched || !pl->have_geom || !pl->x_src) return fail(FG_ERR_STATE, "fg_plan_get_polygons: needs a plan from a search (cell records)");
  if (pl->great_circle && !v2) return fail(FG_ERR_ARG, "fg_plan_get_polygons: a great-circle plan returns x, y, z");
  if (maxv < 3 || maxv > 64) return fail(FG_ERR_ARG, "fg_plan_get_polygons: maxv must be 3..64");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  const long nx = pl->nx, CH = 1L << 20;
  const int dev = pl->device;
  for (long k0 = 0; k0 < nx; k0 += CH) {
    const long n = std::min(CH, nx - k0);
    const size_t nv = (size_t)n * maxv;
    int *d_n = (int *)g_pool.get(dev, n * sizeof(int));
    double *d_v = (double *)g_pool.get(dev, 3 * nv * sizeof(double));
    struct Put { void *a, *b; ~Put() { g_pool.put(a); g_pool.put(b); } } put{d_n, d_v};
    if (!d_n || !d_v) return fail(FG_ERR_HIP, "out of device memory");
    std::vector<XferJob> jobs;
    if (pl->great_circle) {
      double *a = (double *)g_pool.get(dev, (size_t)n * 12 * 8), *b = (double *)g_pool.get(dev, (size_t)n * 12 * 8);
      double *o = (double *)g_pool.get(dev, (size_t)n * FG_GC_POLY_CAP * 3 * 8), *ar = (double *)g_pool.get(dev, (size_t)n * 8);
      struct Put4 { void *p[4]; ~Put4() { for (void *q : p) g_pool.put(q); } } put4{{a, b, o, ar}};
      if (!a || !b || !o || !ar) return fail(FG_ERR_HIP, "out of device memory");
      fgd_xgrid_gather_gc(n, pl->x_src + k0, pl->x_dst + k0, pl->S, pl->D, a, b, pl->stream);
      fgd_gc_clip_batch((int)n, a, b, o, d_n, ar, pl->stream);
      fgd_split_xyz(n, maxv, o, d_v, d_v + nv, d_v + 2 * nv, pl->stream);
      HIPCHK(hipStreamSynchronize(pl->stream));
      jobs.push_back(XferJob{v2 + (size_t)k0 * maxv, d_v + 2 * nv, nv * sizeof(double), true});
    } else {
      fgd_xgrid_polygons(n, pl->x_src + k0, pl->x_dst + k0, pl->S, pl->D, pl->rect ? &pl->rect_tab : nullptr, maxv, d_n, d_v, d_v + nv, pl->stream);
      HIPCHK(hipStreamSynchronize(pl->stream));
    }
    HIPCHK(hipGetLastError());
    jobs.push_back(XferJob{n_out + k0, d_n, (size_t)n * sizeof(int), true});
    jobs.push_back(XferJob{v0 + (size_t)k0 * maxv, d_v, nv * sizeof(double), true});
    jobs.push_back(XferJob{v1 + (size_t)k0 * maxv, d_v + nv, nv * sizeof(double), true});
    if (!xfer_pool().run(dev, jobs)) return fail(FG_ERR_HIP, "fg_plan_get_polygons: device -> host copy failed");
  }
  return 0;
}

extern "C" int fg_plan_get_cell_area(const fg_plan *pl, double *area_in, double *area_out)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  if (!pl->have_geom) return fail(FG_ERR_STATE, "plan holds no cell records");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  if (area_in) HIPCHK(hipMemcpy(area_in, pl->S.area, (size_t)pl->nsrc * sizeof(double), hipMemcpyDeviceToHost));
  if (area_out) HIPCHK(hipMemcpy(area_out, pl->D.area, (size_t)pl->ndst * sizeof(double), hipMemcpyDeviceToHost));
  return 0;
}

// per-cell records of the source (which = 0) or destination (which = 1) cells, for the
// get_grid_cell_struct parity test.  Host arrays; vlon/vlat are [ncells][8].
extern "C" int fg_plan_get_cell_struct(const fg_plan *pl, int which, double *lat_min, double *lat_max, double *lon_min,
                                       double *lon_max, double *lon_avg, int *nvert, double *vlon, double *vlat)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  if (!pl->have_geom) return fail(FG_ERR_STATE, "plan holds no cell records");
  HIPCHK(hipSetDevice(pl->device));
  HIPCHK(hipStreamSynchronize(pl->stream));
  FgCells tmp{};
  std::vector<void *> tmp_blocks;
  struct Free { std::vector<void *> &v; ~Free() { for (void *p : v) g_pool.put(p); } } free_tmp{tmp_blocks};
  if (which && pl->rect) {                       // a rectilinear plan keeps tables, not records: spell them out for the caller
    const size_t nd = (size_t)pl->ndst;
    auto get = [&](size_t bytes) { void *p = g_pool.get(pl->device, bytes); if (p) tmp_blocks.push_back(p); return p; };
    tmp.lat_min = (double *)get(nd * 8); tmp.lat_max = (double *)get(nd * 8); tmp.lon_min = (double *)get(nd * 8);
    tmp.lon_max = (double *)get(nd * 8); tmp.lon_avg = (double *)get(nd * 8); tmp.nv = (int *)get(nd * 4); tmp.verts = (double *)get(nd * 128);
    tmp.area = pl->D.area;
    if (tmp_blocks.size() != 7) return fail(FG_ERR_HIP, "out of device memory");
    FgRect R = pl->rect_tab;
    fgd_rect_materialize(pl->ndst, R, tmp, pl->stream);
    HIPCHK(hipStreamSynchronize(pl->stream));
  }
  const FgCells &c = which ? (pl->rect ? tmp : pl->D) : pl->S;
  size_t n = which ? pl->ndst : pl->nsrc;
  if (lat_min) HIPCHK(hipMemcpy(lat_min, c.lat_min, n * sizeof(double), hipMemcpyDeviceToHost));
  if (lat_max) HIPCHK(hipMemcpy(lat_max, c.lat_max, n * sizeof(double), hipMemcpyDeviceToHost));
  if (lon_min) HIPCHK(hipMemcpy(lon_min, c.lon_min, n * sizeof(double), hipMemcpyDeviceToHost));
  if (lon_max) HIPCHK(hipMemcpy(lon_max, c.lon_max, n * sizeof(double), hipMemcpyDeviceToHost));
  if (lon_avg) HIPCHK(hipMemcpy(lon_avg, c.lon_avg, n * sizeof(double), hipMemcpyDeviceToHost));
  if (nvert) HIPCHK(hipMemcpy(nvert, c.nv, n * sizeof(int), hipMemcpyDeviceToHost));
  if (vlon || vlat) {
    std::vector<double> v(n * 16);
    HIPCHK(hipMemcpy(v.data(), c.verts, n * 16 * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t k = 0; k < n; k++)
      for (int l = 0; l < 8; l++) {
        if (vlon) vlon[k * 8 + l] = v[k * 16 + l];
        if (vlat) vlat[k * 8 + l] = v[k * 16 + 8 + l];
      }
  }
  return 0;
}

extern "C" int fg_plan_set_xgrid(fg_plan *pl, long nxgrid, const int *t_in, const int *i_in, const int *j_in,
                                 const int *i_out, const int *j_out, const double *area,
                                 const double *di_in, const double *dj_in)
{
  if (!pl) return fail(FG_ERR_ARG, "null plan");
  if (nxgrid < 0 || (nxgrid > 0 && (!t_in || !i_in || !j_in || !i_out || !j_out || !area))) return fail(FG_ERR_ARG, "null exchange-cell array");
  if (pl->order == 2 && nxgrid > 0 && (!di_in || !dj_in)) return fail(FG_ERR_ARG, "order 2 needs di_in/dj_in");
  if (pl->finalized || pl->searched) return fail(FG_ERR_STATE, "fg_plan_set_xgrid needs a plan from fg_plan_create_empty");
  HIPCHK(hipSetDevice(pl->device));
  std::vector<int> s(nxgrid), d(nxgrid);
  for (long k = 0; k < nxgrid; k++) {
    int m = t_in[k];
    if (m < 0 || m >= pl->ntiles || i_in[k] < 0 || i_in[k] >= pl->nx_in[m] || j_in[k] < 0 || j_in[k] >= pl->ny_in[m] ||
        i_out[k] < 0 || i_out[k] >= pl->nx_out || j_out[k] < 0 || j_out[k] >= pl->ny_out)
      return fail(FG_ERR_ARG, "exchange cell %ld has an index outside its grid", k);
    s[k] = pl->cell_off[m] + j_in[k] * pl->nx_in[m] + i_in[k];
    d[k] = j_out[k] * pl->nx_out + i_out[k];
  }
  pl->nx = nxgrid;
  pl->x_src = pl->alloc<int>(nxgrid + 1); pl->x_dst = pl->alloc<int>(nxgrid + 1);
  pl->x_area = pl->alloc<double>(nxgrid + 1);
  if (pl->order == 2) { pl->x_c1 = pl->alloc<double>(nxgrid + 1); pl->x_c2 = pl->alloc<double>(nxgrid + 1); }
  if (!pl->x_src || !pl->x_dst || !pl->x_area || (pl->order == 2 && (!pl->x_c1 || !pl->x_c2))) return fail(FG_ERR_HIP, "out of device memory");
  if (nxgrid > 0) {
    HIPCHK(hipMemcpy(pl->x_src, s.data(), nxgrid * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl->x_dst, d.data(), nxgrid * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(hipMemcpy(pl->x_area, area, nxgrid * sizeof(double), hipMemcpyHostToDevice));
    if (pl->order == 2) {
      HIPCHK(hipMemcpy(pl->x_c1, di_in, nxgrid * sizeof(double), hipMemcpyHostToDevice));
      HIPCHK(hipMemcpy(pl->x_c2, dj_in, nxgrid * sizeof(double), hipMemcpyHostToDevice));
    }
  }
  pl->searched = true;
  int rc = build_csr(pl, nullptr);
  if (rc) return rc;
  pl->finalized = true;
  return 0;
}

extern int g_apply_vec, g_apply_ep;
extern "C" void fg_set_apply_ep(int on) { g_apply_ep = on ? 1 : 0; }     // tuning hook: entry-parallel 8-level sweep on records

extern int g_apply_xcd;
extern "C" void fg_set_apply_xcd(int on) { g_apply_xcd = on < 0 ? 0 : on; }       // tuning hook: tile -> XCD mapping of the sweep (0 identity, 1 banded, C >= 2 chunked)
extern "C" void fg_set_apply_vec(int v) { g_apply_vec = (v >= 4) ? 4 : (v >= 2 ? 2 : (v == 1 ? 1 : 0)); }   // tuning hook (scripts/)

static int ensure_il_scratch(fg_plan *pl, bool want_rs)
{
  if (!pl->il_out) {
    pl->il_out = pl->alloc<double>((size_t)pl->ndst * 8);
    if (pl->order == 2) pl->il_m = pl->alloc<double>((size_t)pl->nsrc * 24);
    else pl->il_f = pl->alloc<double>((size_t)pl->f_stride * 8);
    if (!pl->il_out || (pl->order == 2 ? !pl->il_m : !pl->il_f)) return fail(FG_ERR_HIP, "out of device memory");
  }
  if (want_rs && !pl->il_rs) {
    pl->il_rs = pl->alloc<double>((size_t)pl->ndst * 16);
    if (!pl->il_rs) return fail(FG_ERR_HIP, "out of device memory");
  }
  return 0;
}

extern "C" int fg_plan_apply(fg_plan *pl, const double *data, const double *grad_x, const double *grad_y,
                             const int *grad_mask, int has_missing, double missing, int nz,
                             double *out, double *gsum_out)
{
  if (!pl || !data || !out) return fail(FG_ERR_ARG, "null argument");
  if (!pl->finalized) return fail(FG_ERR_STATE, "fg_plan_apply: call fg_plan_finalize first");
  if (nz < 1) return fail(FG_ERR_ARG, "nz must be >= 1");
  if (nz > 1 && has_missing) return fail(FG_ERR_ARG, "conserve_interp: has_missing should be false when nz > 1");
  if (pl->order == 2 && (!grad_x || !grad_y)) return fail(FG_ERR_ARG, "order 2 needs grad_x and grad_y");
  if (pl->order == 2 && has_missing && !grad_mask) return fail(FG_ERR_ARG, "order 2 with missing values needs grad_mask");
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = pl->stream;
  const double miss = has_missing ? missing : -1.e20;            // conserve_interp.c:541-542
  const int ndst = pl->ndst;
  const int nchunk = (nz + 7) / 8;
  if (nchunk > 256) return fail(FG_ERR_ARG, "nz too large for one call (max 2048 levels)");
  if (gsum_out && pl->row_sum_cap < ndst) {
    pl->release(pl->row_sum);
    pl->row_sum = pl->alloc<double>(ndst);
    if (!pl->row_sum) { pl->row_sum_cap = 0; return fail(FG_ERR_HIP, "out of device memory"); }
    pl->row_sum_cap = ndst;
  }
  if (nz > 1) { int rc = ensure_il_scratch(pl, gsum_out != nullptr); if (rc) return rc; }
  pl->apply_pt.start(g_profiling != 0 && pl->apply_spans < 256, st);     // (a long sweep loop must not pile up timing events)
  pl->apply_pt.begin(PH_APPLY);
  int nred = 0;
  for (int k0 = 0; k0 < nz; k0 += 8) {
    const int nbv = (nz - k0 < 8) ? nz - k0 : 8;
    const double *f = data + (size_t)k0 * pl->f_stride;
    const double *gx = grad_x ? grad_x + (size_t)k0 * pl->nsrc : nullptr;
    const double *gy = grad_y ? grad_y + (size_t)k0 * pl->nsrc : nullptr;
    double *o = out + (size_t)k0 * ndst;
    if (nbv == 1) {
      fgd_apply1(pl->order, ndst, pl->csr, f, gx, gy, grad_mask, has_missing, miss, o, gsum_out ? pl->row_sum : nullptr, st, pl->nx);
      if (gsum_out) fgd_reduce_sum(pl->row_sum, ndst, pl->red_partial, pl->red_result + nred++, st);
    } else {
      const int nbp = nbv > 4 ? 8 : (nbv > 2 ? 4 : 2);
      double *rs = gsum_out ? pl->il_rs : nullptr;
      if (pl->order == 2) {
        // field + gradients of the chunk -> one record per source cell, then the sweep writes level-major directly
        fgd_merge3(nbp, pl->nsrc, pl->src_idx_f, f, pl->f_stride, gx, gy, pl->nsrc, nbv, pl->il_m, st);
        fgd_apply_il_merged(nbp, ndst, pl->nx, pl->csr, pl->il_m, miss, o, rs, (long)ndst, nbv, st);
      } else {
        const double *ins[3] = {f, nullptr, nullptr};
        double *outs[3] = {pl->il_f, nullptr, nullptr};
        const long lds[3] = {pl->f_stride, 0, 0}, ns[3] = {pl->f_stride, 0, 0};
        fgd_interleave3(nbp, 1, ins, lds, ns, outs, nbv, st);
        fgd_apply_il(1, nbp, ndst, pl->csr, pl->il_f, nullptr, nullptr, miss, o, rs, (long)ndst, nbv, st, pl->nx);
      }
      if (gsum_out) fgd_reduce_sum(pl->il_rs, (long)ndst * nbp, pl->red_partial, pl->red_result + nred++, st);
    }
  }
  pl->apply_pt.end();
  if (pl->apply_pt.on) pl->apply_spans++;
  if (gsum_out) {
    double parts[256];
    HIPCHK(hipMemcpyAsync(parts, pl->red_result, nred * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double g = 0;
    for (int k = 0; k < nred; k++) g += parts[k];
    *gsum_out = g;
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// The order-2 sweep on records made by fg_c2l_gradient_records: no level-major detour, no merge pass.
extern "C" int fg_plan_apply_records(fg_plan *pl, int nz, const double *rec, double *out, double *gsum_out)
{
  if (!pl || !rec || !out) return fail(FG_ERR_ARG, "null argument");
  if (!pl->finalized) return fail(FG_ERR_STATE, "fg_plan_apply_records: call fg_plan_finalize first");
  if (pl->order != 2) return fail(FG_ERR_ARG, "fg_plan_apply_records: records carry gradients, the plan is first order");
  if (nz < 1 || nz > 8) return fail(FG_ERR_ARG, "fg_plan_apply_records: 1 to 8 levels per call");
  HIPCHK(hipSetDevice(pl->device));
  hipStream_t st = pl->stream;
  const int ndst = pl->ndst, nbp = nz > 4 ? 8 : (nz > 2 ? 4 : 2);
  if (gsum_out && !pl->il_rs) {
    pl->il_rs = pl->alloc<double>((size_t)ndst * 16);
    if (!pl->il_rs) return fail(FG_ERR_HIP, "out of device memory");
  }
  pl->apply_pt.start(g_profiling != 0 && pl->apply_spans < 256, st);     // (a long sweep loop must not pile up timing events)
  pl->apply_pt.begin(PH_APPLY);
  fgd_apply_il_merged(nbp, ndst, pl->nx, pl->csr, rec, -1.e20, out, gsum_out ? pl->il_rs : nullptr, (long)ndst, nz, st);
  if (gsum_out) fgd_reduce_sum(pl->il_rs, (long)ndst * nbp, pl->red_partial, pl->red_result, st);
  pl->apply_pt.end();
  if (pl->apply_pt.on) pl->apply_spans++;
  if (gsum_out) {
    HIPCHK(hipMemcpyAsync(gsum_out, pl->red_result, sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ------------------------------------------------------------------- the sweep with every option
static int ex_check(fg_plan *pl, const fg_apply_opts *o, const double *data, const double *gx, const double *gy, int nz, const char *who)
{
  if (!pl || !o || !data) return fail(FG_ERR_ARG, "%s: null argument", who);
  if (!pl->finalized) return fail(FG_ERR_STATE, "%s: call fg_plan_finalize first", who);
  if (nz < 1) return fail(FG_ERR_ARG, "nz must be >= 1");
  if (nz > 1 && o->has_missing) return fail(FG_ERR_ARG, "conserve_interp: has_missing should be false when nz > 1");
  if (nz > 1 && o->field_area) return fail(FG_ERR_ARG, "conserve_interp: cell_measures should be false when nz > 1");
  if (nz > 1 && o->cell_methods_sum) return fail(FG_ERR_ARG, "conserve_interp: cell_methods should not be sum when nz > 1");
  if (pl->order == 2 && (!gx || !gy)) return fail(FG_ERR_ARG, "order 2 needs grad_x and grad_y");
  if ((o->cell_methods_sum || o->field_area) && !o->cell_area_in && !pl->have_geom)
    return fail(FG_ERR_ARG, "%s: cell_area_in is required (this plan was loaded from a remap file and holds no cell areas)", who);
  return 0;
}

static int ex_error_word(fg_plan *pl)
{
  int e = 0;
  HIPCHK(hipMemcpyAsync(&e, pl->xerr, sizeof(int), hipMemcpyDeviceToHost, pl->stream));
  HIPCHK(hipStreamSynchronize(pl->stream));
  if (e & FG_XERR_AREA_MISSING) return fail(FG_ERR_DATA, "conserve_interp: data is not missing but area is missing");
  if (e & FG_XERR_ABOVE) return fail(FG_ERR_DATA, " xdata is greater than f_bar_max ");
  if (e & FG_XERR_BELOW) return fail(FG_ERR_DATA, " xdata is less than f_bar_min ");
  return 0;
}

static int ex_scratch(fg_plan *pl, bool mono)
{
  if (!pl->xerr) { pl->xerr = pl->alloc<int>(4); if (!pl->xerr) return fail(FG_ERR_HIP, "out of device memory"); }
  if (pl->row_sum_cap < pl->ndst) {
    pl->release(pl->row_sum);
    pl->row_sum = pl->alloc<double>(pl->ndst);
    if (!pl->row_sum) { pl->row_sum_cap = 0; return fail(FG_ERR_HIP, "out of device memory"); }
    pl->row_sum_cap = pl->ndst;
  }
  if (mono && !pl->mono_x) {
    pl->mono_x = pl->alloc<double>(pl->nx > 0 ? pl->nx : 1);
    pl->mono_b = pl->alloc<double>((size_t)pl->nsrc * 4);
    if (!pl->mono_x || !pl->mono_b) return fail(FG_ERR_HIP, "out of device memory");
  }
  return 0;
}

static FgApplyEx ex_pack(fg_plan *pl, const fg_apply_opts *o, const int *gmask)
{
  FgApplyEx x{};
  x.weight = o->weight; x.field_area = o->field_area; x.cell_area_out = o->cell_methods_sum ? nullptr : o->cell_area_out;
  x.cell_area = o->cell_area_in ? o->cell_area_in : (pl->have_geom ? pl->S.area : nullptr);
  x.gmask = gmask; x.area_missing = o->area_missing;
  x.has_missing = o->has_missing; x.missing = o->has_missing ? o->missing : -1.e20;   // conserve_interp.c:541-542
  x.sum = o->cell_methods_sum;
  return x;
}

extern "C" int fg_plan_mono_begin(fg_plan *pl, const fg_apply_opts *o, const double *data, const double *grad_x,
                                  const double *grad_y, const int *grad_mask)
{
  int rc = ex_check(pl, o, data, grad_x, grad_y, 1, "fg_plan_mono_begin");
  if (rc) return rc;
  if (pl->order != 2) return fail(FG_ERR_ARG, "the monotone limiter belongs to conserve_order2 (conserve_interp.c:527-531)");
  HIPCHK(hipSetDevice(pl->device));
  if ((rc = ex_scratch(pl, true))) return rc;
  const double miss = o->has_missing ? o->missing : -1.e20;
  double *b = pl->mono_b;
  const size_t n = pl->nsrc;
  HIPCHK(hipMemsetAsync(pl->xerr, 0, sizeof(int), pl->stream));
  fgd_mono_bounds(pl->tiles_dev, pl->ntiles, pl->nsrc, pl->src_idx_f, data, miss, b, b + n, b + 2 * n, b + 3 * n, pl->stream);
  fgd_mono_xdata(pl->nx, pl->csr, data, grad_x, grad_y, grad_mask, miss, pl->mono_x, b + 2 * n, b + 3 * n, pl->stream);
  HIPCHK(hipGetLastError());
  pl->mono_open = true;
  return 0;
}

extern "C" int fg_plan_mono_minmax_dev(fg_plan *pl, double **f_min, double **f_max)
{
  if (!pl || !pl->mono_open) return fail(FG_ERR_STATE, "fg_plan_mono_minmax_dev: call fg_plan_mono_begin first");
  if (f_max) *f_max = pl->mono_b + 2 * (size_t)pl->nsrc;
  if (f_min) *f_min = pl->mono_b + 3 * (size_t)pl->nsrc;
  return 0;
}

extern "C" int fg_plan_mono_copy_minmax(fg_plan *pl, int to_plan, double *f_min, double *f_max)
{
  if (!pl || !pl->mono_open || !f_min || !f_max) return fail(FG_ERR_STATE, "fg_plan_mono_copy_minmax: call fg_plan_mono_begin first");
  HIPCHK(hipSetDevice(pl->device));
  double *mx = pl->mono_b + 2 * (size_t)pl->nsrc, *mn = pl->mono_b + 3 * (size_t)pl->nsrc;
  const size_t bytes = (size_t)pl->nsrc * sizeof(double);
  HIPCHK(hipMemcpyAsync(to_plan ? mn : f_min, to_plan ? f_min : mn, bytes, hipMemcpyDeviceToDevice, pl->stream));
  HIPCHK(hipMemcpyAsync(to_plan ? mx : f_max, to_plan ? f_max : mx, bytes, hipMemcpyDeviceToDevice, pl->stream));
  HIPCHK(hipStreamSynchronize(pl->stream));
  return 0;
}

extern "C" int fg_plan_mono_end(fg_plan *pl, const fg_apply_opts *o, const double *data, double *out, double *gsum_out)
{
  if (!pl || !o || !data || !out) return fail(FG_ERR_ARG, "null argument");
  if (!pl->mono_open) return fail(FG_ERR_STATE, "fg_plan_mono_end: call fg_plan_mono_begin first");
  HIPCHK(hipSetDevice(pl->device));
  pl->mono_open = false;
  const double miss = o->has_missing ? o->missing : -1.e20;
  double *b = pl->mono_b;
  const size_t n = pl->nsrc;
  fgd_mono_limit(pl->nx, pl->csr, data, miss, b, b + n, b + 2 * n, b + 3 * n, pl->mono_x, pl->xerr, pl->stream);
  FgApplyEx x = ex_pack(pl, o, nullptr);
  x.xdata = pl->mono_x;
  fgd_apply_ex(2, pl->ndst, pl->csr, data, nullptr, nullptr, x, out, gsum_out ? pl->row_sum : nullptr, pl->xerr, pl->stream, pl->nx);
  if (gsum_out) {
    fgd_reduce_sum(pl->row_sum, pl->ndst, pl->red_partial, pl->red_result, pl->stream);
    HIPCHK(hipMemcpyAsync(gsum_out, pl->red_result, sizeof(double), hipMemcpyDeviceToHost, pl->stream));
  }
  HIPCHK(hipGetLastError());
  return ex_error_word(pl);
}

extern "C" int fg_plan_apply_ex(fg_plan *pl, const fg_apply_opts *o, const double *data, const double *grad_x,
                                const double *grad_y, const int *grad_mask, int nz, double *out, double *gsum_out)
{
  int rc = ex_check(pl, o, data, grad_x, grad_y, nz, "fg_plan_apply_ex");
  if (rc) return rc;
  if (!out) return fail(FG_ERR_ARG, "null argument");
  if (o->monotonic && pl->order == 2) {
    if (nz != 1) return fail(FG_ERR_ARG, "the monotone limiter works on one level per call (conserve_interp.c:648-651 index level 0 only)");
    if ((rc = fg_plan_mono_begin(pl, o, data, grad_x, grad_y, grad_mask))) return rc;
    return fg_plan_mono_end(pl, o, data, out, gsum_out);
  }
  HIPCHK(hipSetDevice(pl->device));
  if ((rc = ex_scratch(pl, false))) return rc;
  if (nz > 256) return fail(FG_ERR_ARG, "nz too large for one call (max 256 levels)");
  hipStream_t st = pl->stream;
  HIPCHK(hipMemsetAsync(pl->xerr, 0, sizeof(int), st));
  FgApplyEx x = ex_pack(pl, o, grad_mask);
  for (int k = 0; k < nz; k++) {
    fgd_apply_ex(pl->order, pl->ndst, pl->csr, data + (size_t)k * pl->f_stride, grad_x ? grad_x + (size_t)k * pl->nsrc : nullptr,
                 grad_y ? grad_y + (size_t)k * pl->nsrc : nullptr, x, out + (size_t)k * pl->ndst,
                 gsum_out ? pl->row_sum : nullptr, pl->xerr, st, pl->nx);
    if (gsum_out) fgd_reduce_sum(pl->row_sum, pl->ndst, pl->red_partial, pl->red_result + k, st);
  }
  HIPCHK(hipGetLastError());
  if (gsum_out) {
    double parts[256];
    HIPCHK(hipMemcpyAsync(parts, pl->red_result, nz * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(hipStreamSynchronize(st));
    double g = 0;
    for (int k = 0; k < nz; k++) g += parts[k];
    *gsum_out = g;
  }
  return ex_error_word(pl);
}

// The sweep on fields the caller already keeps interleaved: data_il [F][nb], grad_x_il/grad_y_il [ncells_in][nb],
// out_il [ndst][nb]; nb in {2, 4, 8}; no missing values.  Skips the two transposes of fg_plan_apply.
extern "C" int fg_plan_apply_interleaved(fg_plan *pl, int nb, const double *data_il, const double *grad_x_il,
                                         const double *grad_y_il, double *out_il, double *gsum_out)
{
  if (!pl || !data_il || !out_il) return fail(FG_ERR_ARG, "null argument");
  if (!pl->finalized) return fail(FG_ERR_STATE, "fg_plan_apply_interleaved: call fg_plan_finalize first");
  if (nb != 2 && nb != 4 && nb != 8 && nb != 16) return fail(FG_ERR_ARG, "nb must be 2, 4, 8 or 16");
  if (pl->order == 2 && (!grad_x_il || !grad_y_il)) return fail(FG_ERR_ARG, "order 2 needs grad_x and grad_y");
  HIPCHK(hipSetDevice(pl->device));
  if (gsum_out) { int rc = ensure_il_scratch(pl, true); if (rc) return rc; }
  pl->apply_pt.start(g_profiling != 0 && pl->apply_spans < 256, pl->stream);
  pl->apply_pt.begin(PH_APPLY);
  fgd_apply_il(pl->order, nb, pl->ndst, pl->csr, data_il, grad_x_il, grad_y_il, -1.e20, out_il,
               gsum_out ? pl->il_rs : nullptr, 0, nb, pl->stream, pl->nx);
  pl->apply_pt.end();
  if (pl->apply_pt.on) pl->apply_spans++;
  if (gsum_out) {
    fgd_reduce_sum(pl->il_rs, (long)pl->ndst * nb, pl->red_partial, pl->red_result, pl->stream);
    HIPCHK(hipMemcpyAsync(gsum_out, pl->red_result, sizeof(double), hipMemcpyDeviceToHost, pl->stream));
    HIPCHK(hipStreamSynchronize(pl->stream));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

// ----------------------------------------------------------------------------- (B1)
// libfrencutils drop-ins.  Fatal errors follow mosaic_util.c:57-65.
static void fatal(const char *msg)
{
  fprintf(stderr, "FATAL Error: %s\n", msg);
  exit(1);
}

#ifndef MAXXGRID
#define MAXXGRID 5e6          /* create_xgrid.h:22-28, serial build */
#endif

extern "C" int get_maxxgrid(void) { return MAXXGRID; }
extern "C" int get_maxxgrid_(void) { return get_maxxgrid(); }

static int b1_device(void)
{
  const char *e = getenv("FREGRID_HIP_DEVICE");
  return e ? atoi(e) : 0;
}

extern "C" void get_grid_area(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area)
{
  // a 1x1 dummy destination keeps the plan machinery uniform; only the source records are used
  int nx = *nlon, ny = *nlat, one = 1;
  const double dl[4] = {0, 0.01, 0, 0.01}, da[4] = {0, 0, 0.01, 0.01};
  const double *lons[1] = {lon}, *lats[1] = {lat};
  fg_plan *pl = nullptr;
  long rc = fg_plan_create(FG_CONSERVE_ORDER1, 1, &nx, &ny, lons, lats, nullptr, one, one, dl, da, b1_device(), &pl);
  if (rc < 0) fatal(fg_last_error());
  if (fg_plan_get_cell_area(pl, area, nullptr)) fatal(fg_last_error());
  fg_plan_destroy(pl);
}
extern "C" void get_grid_area_(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area)
{
  get_grid_area(nlon, nlat, lon, lat, area);
}

extern "C" int create_xgrid_great_circle(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                                         const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                                         const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out,
                                         double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  int nx1 = *nlon_in, ny1 = *nlat_in;
  const double *lons[1] = {lon_in}, *lats[1] = {lat_in}, *masks[1] = {mask_in};
  fg_plan *pl = nullptr;
  long nx = fg_plan_create_great_circle(1, &nx1, &ny1, lons, lats, masks, *nlon_out, *nlat_out, lon_out, lat_out, b1_device(), &pl);
  if (nx < 0) fatal(fg_last_error());
  if (nx > (long)MAXXGRID) fatal("nxgrid is greater than MAXXGRID, increase MAXXGRID");      // create_xgrid.c:1444
  if (fg_plan_get_xgrid(pl, nullptr, i_in, j_in, i_out, j_out, xgrid_area, nullptr, nullptr)) fatal(fg_last_error());
  for (long k = 0; k < nx; k++) { if (xgrid_clon) xgrid_clon[k] = 0; if (xgrid_clat) xgrid_clat[k] = 0; }   // :1446-1447
  fg_plan_destroy(pl);
  return (int)nx;
}
extern "C" int create_xgrid_great_circle_(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                                          const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                                          const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out,
                                          double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  return create_xgrid_great_circle(nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in,
                                   i_in, j_in, i_out, j_out, xgrid_area, xgrid_clon, xgrid_clat);
}

extern "C" void get_grid_great_circle_area(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area)
{
  int nx = *nlon, ny = *nlat, one = 1;
  const double dl[4] = {0, 0.01, 0, 0.01}, da[4] = {0, 0, 0.01, 0.01};      // dummy 1x1 destination
  const double *lons[1] = {lon}, *lats[1] = {lat};
  fg_plan *pl = nullptr;
  long rc = fg_plan_create_great_circle(1, &nx, &ny, lons, lats, nullptr, one, one, dl, da, b1_device(), &pl);
  if (rc < 0) fatal(fg_last_error());
  if (fg_plan_get_cell_area(pl, area, nullptr)) fatal(fg_last_error());
  fg_plan_destroy(pl);
}
extern "C" void get_grid_great_circle_area_(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area)
{
  get_grid_great_circle_area(nlon, nlat, lon, lat, area);
}

extern "C" int fg_gc_clip_batch(int npairs, const double *a, const double *b, double *out, int *n_out, double *area, int device)
{
  if (npairs < 0 || !a || !b || !out || !n_out || !area) return fail(FG_ERR_ARG, "null argument");
  if (npairs == 0) return 0;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(FG_ERR_HIP, "no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  HIPCHK(hipSetDevice(device));
  const size_t na = (size_t)npairs * 12, no = (size_t)npairs * FG_GC_POLY_CAP * 3;
  double *d_a = (double *)g_pool.get(device, (2 * na + no + npairs) * sizeof(double));
  int *d_n = (int *)g_pool.get(device, (size_t)npairs * sizeof(int));
  if (!d_a || !d_n) { if (d_a) g_pool.put(d_a); if (d_n) g_pool.put(d_n); return fail(FG_ERR_HIP, "out of device memory"); }
  double *d_b = d_a + na, *d_o = d_b + na, *d_ar = d_o + no;
  int rc = 0;
  if (hipMemcpy(d_a, a, na * sizeof(double), hipMemcpyHostToDevice) != hipSuccess ||
      hipMemcpy(d_b, b, na * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) rc = FG_ERR_HIP;
  if (!rc) {
    fgd_gc_clip_batch(npairs, d_a, d_b, d_o, d_n, d_ar, nullptr);
    if (hipMemcpy(out, d_o, no * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(n_out, d_n, (size_t)npairs * sizeof(int), hipMemcpyDeviceToHost) != hipSuccess ||
        hipMemcpy(area, d_ar, (size_t)npairs * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) rc = FG_ERR_HIP;
  }
  g_pool.put(d_a); g_pool.put(d_n);
  if (rc) return fail(rc, "fg_gc_clip_batch: HIP copy/launch failed: %s", hipGetErrorString(hipGetLastError()));
  return 0;
}

extern "C" double great_circle_area(int n, const double *x, const double *y, const double *z)
{
  if (n < 1 || n > 64) fatal("great_circle_area (HIP): 1 <= n <= 64");
  double h[64 * 3], area = 0;
  for (int k = 0; k < n; k++) { h[k * 3] = x[k]; h[k * 3 + 1] = y[k]; h[k * 3 + 2] = z[k]; }
  if (hipSetDevice(b1_device()) != hipSuccess) fatal("no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  double *d = (double *)g_pool.get(b1_device(), (64 * 3 + 2) * sizeof(double));
  if (!d) fatal("out of device memory");
  int *dn = (int *)(d + 64 * 3 + 1);
  bool ok = hipMemcpy(d, h, n * 3 * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(dn, &n, sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    fgd_gc_area_batch(1, 64, d, dn, d + 64 * 3, nullptr);
    ok = hipMemcpy(&area, d + 64 * 3, sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  }
  g_pool.put(d);
  if (!ok) fatal("great_circle_area: HIP copy/launch failed");
  return area;
}

// sin / cos of n host values with the device's latitude trig (sincos_glibc.h): parity probe for the tests
extern "C" int fg_sincos_batch(long n, const double *x, double *s, double *c, int device)
{
  if (n < 0 || !x || !s || !c) return fail(FG_ERR_ARG, "null argument");
  if (n == 0) return 0;
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(FG_ERR_HIP, "no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  HIPCHK(hipSetDevice(device));
  double *d = (double *)g_pool.get(device, 3 * (size_t)n * sizeof(double));
  if (!d) return fail(FG_ERR_HIP, "out of device memory");
  bool ok = hipMemcpy(d, x, n * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    fgd_sincos_probe(n, d, d + n, d + 2 * n, nullptr);
    ok = hipMemcpy(s, d + n, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
         hipMemcpy(c, d + 2 * n, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  }
  g_pool.put(d);
  if (!ok) return fail(FG_ERR_HIP, "fg_sincos_batch: HIP copy/launch failed");
  return 0;
}

// clip_2dx2d_great_circle / great_circle_area, one polygon pair per call (create_xgrid.h / mosaic_util.h prototypes)
extern "C" int clip_2dx2d_great_circle(const double x1_in[], const double y1_in[], const double z1_in[], int n1_in,
                                       const double x2_in[], const double y2_in[], const double z2_in[], int n2_in,
                                       double x_out[], double y_out[], double z_out[])
{
  if (n1_in != 4 || n2_in != 4) fatal("clip_2dx2d_great_circle (HIP): quadrilaterals only (n1_in == n2_in == 4, as every caller in the reference passes)");
  double a[12], b[12], out[FG_GC_POLY_CAP * 3], area;
  int n_out = 0;
  for (int k = 0; k < 4; k++) {
    a[k * 3] = x1_in[k]; a[k * 3 + 1] = y1_in[k]; a[k * 3 + 2] = z1_in[k];
    b[k * 3] = x2_in[k]; b[k * 3 + 1] = y2_in[k]; b[k * 3 + 2] = z2_in[k];
  }
  if (fg_gc_clip_batch(1, a, b, out, &n_out, &area, b1_device())) fatal(fg_last_error());
  if (n_out == -1) fatal("create_xgrid.c(clip_2dx2d_great_circle): grid box 1 is not convex");
  if (n_out == -2) fatal("create_xgrid.c(clip_2dx2d_great_circle): grid box 2 is not convex");
  if (n_out < 0) fatal(gc_clip_error(-n_out));
  for (int k = 0; k < n_out; k++) { x_out[k] = out[k * 3]; y_out[k] = out[k * 3 + 1]; z_out[k] = out[k * 3 + 2]; }
  return n_out;
}

// create_xgrid_1dx2d_order1/2 and create_xgrid_2dx1d_order1/2 (create_xgrid.c:208-591) through the shared search pipeline
static int b1_create_xgrid_box(int box_is_src, int order, int nxb, int nyb, const double *lon_b, const double *lat_b,
                               int nxq, int nyq, const double *lon_q, const double *lat_q, const double *mask,
                               int *i_in, int *j_in, int *i_out, int *j_out, double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  const int dev = b1_device();
  if (nxb < 1 || nyb < 1 || nxq < 1 || nyq < 1) fatal("create_xgrid: bad grid sizes");
  fg_plan *pl = nullptr;
  if (plan_base(order, 1, &nxb, &nyb, nxq, nyq, dev, &pl)) fatal(fg_last_error());
  const size_t npb = (size_t)(nxb + 1) * (nyb + 1), npq = (size_t)(nxq + 1) * (nyq + 1);
  std::vector<double> tx(npb), ty(npb);                                  // the reference's tmpx/tmpy, :229-236
  for (int j = 0; j <= nyb; j++) for (int i = 0; i <= nxb; i++) { tx[(size_t)j * (nxb + 1) + i] = lon_b[i]; ty[(size_t)j * (nxb + 1) + i] = lat_b[j]; }
  const size_t nmask = box_is_src ? (size_t)nxb * nyb : (size_t)nxq * nyq;
  double *d = pl->alloc<double>(2 * npb + 2 * npq + (nxb + 1) + (nyb + 1) + nmask);
  if (!d) fatal("out of device memory");
  double *d_tx = d, *d_ty = d_tx + npb, *d_lq = d_ty + npb, *d_aq = d_lq + npq, *d_lb = d_aq + npq, *d_ab = d_lb + (nxb + 1), *d_mask = d_ab + (nyb + 1);
  bool ok = hipMemcpy(d_tx, tx.data(), npb * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d_ty, ty.data(), npb * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d_lq, lon_q, npq * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d_aq, lat_q, npq * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d_lb, lon_b, (nxb + 1) * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d_ab, lat_b, (nyb + 1) * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d_mask, mask, nmask * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) fatal("grid upload failed");
  BoxMode bm;
  bm.box = FgBox{d_lb, d_ab, nxb, nyb};
  bm.mask_quad = box_is_src ? nullptr : d_mask;
  bm.no_adjust = (box_is_src && order == 1 && !(nxb > 1)) ? 1 : 0;
  const double *lons[1] = {d_tx}, *lats[1] = {d_ty}, *masks[1] = {box_is_src ? d_mask : nullptr};
  double mdlat, mdlon;
  sample_extents(nxq, nyq, lon_q, lat_q, &mdlat, &mdlon);
  long nx = plan_search(pl, lons, lats, box_is_src ? masks : nullptr, d_lq, d_aq, mdlat, mdlon, nullptr, nullptr, &bm);
  if (nx < 0) fatal(fg_last_error());
  if (nx > (long)MAXXGRID) fatal("nxgrid is greater than MAXXGRID, increase MAXXGRID");
  std::vector<int> bi(nx), bj(nx), qi(nx), qj(nx);
  if (fg_plan_get_xgrid(pl, nullptr, bi.data(), bj.data(), qi.data(), qj.data(), xgrid_area, order == 2 ? xgrid_clon : nullptr,
                        order == 2 ? xgrid_clat : nullptr)) fatal(fg_last_error());
  for (long k = 0; k < nx; k++) {
    if (box_is_src) { i_in[k] = bi[k]; j_in[k] = bj[k]; i_out[k] = qi[k]; j_out[k] = qj[k]; }
    else { i_in[k] = qi[k]; j_in[k] = qj[k]; i_out[k] = bi[k]; j_out[k] = bj[k]; }
  }
  fg_plan_destroy(pl);
  return (int)nx;
}

// clip / box_ctrlat / box_ctrlon / get_grid_area_no_adjust (create_xgrid.h:37-45)
extern "C" int clip(const double lon_in[], const double lat_in[], int n_in, double ll_lon, double ll_lat, double ur_lon, double ur_lat,
                    double lon_out[], double lat_out[])
{
  if (n_in < 1 || n_in > 12) fatal("clip (HIP): 1 <= n_in <= 12");
  const int dev = b1_device();
  if (hipSetDevice(dev) != hipSuccess) fatal("no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  double *d = (double *)g_pool.get(dev, (4 * 16 + 2) * sizeof(double));
  if (!d) fatal("out of device memory");
  int *dn = (int *)(d + 64);
  int n = 0;
  bool ok = hipMemcpy(d, lon_in, n_in * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d + 16, lat_in, n_in * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    fgd_clip_single(d, d + 16, n_in, ll_lon, ll_lat, ur_lon, ur_lat, d + 32, d + 48, dn, nullptr);
    ok = hipMemcpy(&n, dn, sizeof(int), hipMemcpyDeviceToHost) == hipSuccess;
    if (ok && n > 0) ok = hipMemcpy(lon_out, d + 32, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess &&
                          hipMemcpy(lat_out, d + 48, n * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  }
  g_pool.put(d);
  if (!ok) fatal("clip: HIP copy/launch failed");
  if (n < 0) fatal("clip (HIP): clipped polygon exceeds 16 vertices");
  return n;
}

static void b1_box_ctr(double ll_lon, double ll_lat, double ur_lon, double ur_lat, double clon, double out[2])
{
  const int dev = b1_device();
  if (hipSetDevice(dev) != hipSuccess) fatal("no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  double *d = (double *)g_pool.get(dev, 2 * sizeof(double));
  if (!d) fatal("out of device memory");
  fgd_box_ctr(ll_lon, ll_lat, ur_lon, ur_lat, clon, d, nullptr);
  bool ok = hipMemcpy(out, d, 2 * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  g_pool.put(d);
  if (!ok) fatal("box_ctr: HIP copy/launch failed");
}
extern "C" double box_ctrlat(double ll_lon, double ll_lat, double ur_lon, double ur_lat)
{
  double o[2]; b1_box_ctr(ll_lon, ll_lat, ur_lon, ur_lat, 0.0, o); return o[0];
}
extern "C" double box_ctrlon(double ll_lon, double ll_lat, double ur_lon, double ur_lat, double clon)
{
  double o[2]; b1_box_ctr(ll_lon, ll_lat, ur_lon, ur_lat, clon, o); return o[1];
}

extern "C" void get_grid_area_no_adjust(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area)
{
  const int nx = *nlon, ny = *nlat, dev = b1_device();
  if (nx < 1 || ny < 1) fatal("get_grid_area_no_adjust: bad grid sizes");
  if (hipSetDevice(dev) != hipSuccess) fatal("no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  const size_t np = (size_t)(nx + 1) * (ny + 1), nc = (size_t)nx * ny;
  double *d = (double *)g_pool.get(dev, (2 * np + nc) * sizeof(double));
  if (!d) fatal("out of device memory");
  bool ok = hipMemcpy(d, lon, np * sizeof(double), hipMemcpyHostToDevice) == hipSuccess &&
            hipMemcpy(d + np, lat, np * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (ok) {
    fgd_grid_area_no_adjust(nx, ny, d, d + np, d + 2 * np, nullptr);
    ok = hipMemcpy(area, d + 2 * np, nc * sizeof(double), hipMemcpyDeviceToHost) == hipSuccess;
  }
  g_pool.put(d);
  if (!ok) fatal("get_grid_area_no_adjust: HIP copy/launch failed");
}
extern "C" void get_grid_area_no_adjust_(const int *nlon, const int *nlat, const double *lon, const double *lat, double *area)
{
  get_grid_area_no_adjust(nlon, nlat, lon, lat, area);
}

#define B1_BOX_FN(NAME, BOXSRC, ORDER)                                                                                          \
  extern "C" int NAME(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out, const double *lon_in, \
                      const double *lat_in, const double *lon_out, const double *lat_out, const double *mask_in, int *i_in,     \
                      int *j_in, int *i_out, int *j_out, double *xgrid_area B1_BOX_EXTRA_##ORDER)                              \
  {                                                                                                                             \
    return (BOXSRC) ? b1_create_xgrid_box(1, ORDER, *nlon_in, *nlat_in, lon_in, lat_in, *nlon_out, *nlat_out, lon_out, lat_out, \
                                          mask_in, i_in, j_in, i_out, j_out, xgrid_area, B1_BOX_ARGS_##ORDER)                   \
                    : b1_create_xgrid_box(0, ORDER, *nlon_out, *nlat_out, lon_out, lat_out, *nlon_in, *nlat_in, lon_in, lat_in, \
                                          mask_in, i_in, j_in, i_out, j_out, xgrid_area, B1_BOX_ARGS_##ORDER);                  \
  }
#define B1_BOX_EXTRA_1
#define B1_BOX_EXTRA_2 , double *xgrid_clon, double *xgrid_clat
#define B1_BOX_ARGS_1 nullptr, nullptr
#define B1_BOX_ARGS_2 xgrid_clon, xgrid_clat
B1_BOX_FN(create_xgrid_1dx2d_order1, 1, 1)
B1_BOX_FN(create_xgrid_1dx2d_order1_, 1, 1)
B1_BOX_FN(create_xgrid_1dx2d_order2, 1, 2)
B1_BOX_FN(create_xgrid_1dx2d_order2_, 1, 2)
B1_BOX_FN(create_xgrid_2dx1d_order1, 0, 1)
B1_BOX_FN(create_xgrid_2dx1d_order1_, 0, 1)
B1_BOX_FN(create_xgrid_2dx1d_order2, 0, 2)
B1_BOX_FN(create_xgrid_2dx1d_order2_, 0, 2)

static int b1_create_xgrid(int order, const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                           const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                           const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out,
                           double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  int nx1 = *nlon_in, ny1 = *nlat_in;
  const double *lons[1] = {lon_in}, *lats[1] = {lat_in}, *masks[1] = {mask_in};
  fg_plan *pl = nullptr;
  long nx = fg_plan_create(order, 1, &nx1, &ny1, lons, lats, masks, *nlon_out, *nlat_out, lon_out, lat_out, b1_device(), &pl);
  if (nx < 0) fatal(fg_last_error());
  if (nx >= (long)MAXXGRID)                                     // create_xgrid.c:1087-1088
    fatal("nxgrid is greater than MAXXGRID/nthreads, increase MAXXGRID, decrease nthreads, or increase number of MPI ranks");
  if (fg_plan_get_xgrid(pl, nullptr, i_in, j_in, i_out, j_out, xgrid_area, xgrid_clon, xgrid_clat)) fatal(fg_last_error());
  fg_plan_destroy(pl);
  return (int)nx;
}

extern "C" int create_xgrid_2dx2d_order1(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                                         const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                                         const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out, double *xgrid_area)
{
  return b1_create_xgrid(FG_CONSERVE_ORDER1, nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in,
                         i_in, j_in, i_out, j_out, xgrid_area, nullptr, nullptr);
}
extern "C" int create_xgrid_2dx2d_order2(const int *nlon_in, const int *nlat_in, const int *nlon_out, const int *nlat_out,
                                         const double *lon_in, const double *lat_in, const double *lon_out, const double *lat_out,
                                         const double *mask_in, int *i_in, int *j_in, int *i_out, int *j_out,
                                         double *xgrid_area, double *xgrid_clon, double *xgrid_clat)
{
  return b1_create_xgrid(FG_CONSERVE_ORDER2, nlon_in, nlat_in, nlon_out, nlat_out, lon_in, lat_in, lon_out, lat_out, mask_in,
                         i_in, j_in, i_out, j_out, xgrid_area, xgrid_clon, xgrid_clat);
}
extern "C" int create_xgrid_2dx2d_order1_(const int *a, const int *b, const int *c, const int *d, const double *e, const double *f,
                                          const double *g, const double *h, const double *m, int *i1, int *j1, int *i2, int *j2, double *xa)
{
  return create_xgrid_2dx2d_order1(a, b, c, d, e, f, g, h, m, i1, j1, i2, j2, xa);
}
extern "C" int create_xgrid_2dx2d_order2_(const int *a, const int *b, const int *c, const int *d, const double *e, const double *f,
                                          const double *g, const double *h, const double *m, int *i1, int *j1, int *i2, int *j2,
                                          double *xa, double *xl, double *xt)
{
  return create_xgrid_2dx2d_order2(a, b, c, d, e, f, g, h, m, i1, j1, i2, j2, xa, xl, xt);
}

static int plan_apply_frac(fg_plan *pl, const double *data, double *out)
{
  fgd_apply_frac(pl->ndst, pl->csr, data, out, pl->stream);
  HIPCHK(hipStreamSynchronize(pl->stream));
  HIPCHK(hipGetLastError());
  return 0;
}

// interp.c:262-305: first-order remap with weights xarea / (sum of xarea over the destination cell)
extern "C" void conserve_interp(int nx_src, int ny_src, int nx_dst, int ny_dst, const double *x_src,
                                const double *y_src, const double *x_dst, const double *y_dst,
                                const double *mask_src, const double *data_src, double *data_dst)
{
  const double *lons[1] = {x_src}, *lats[1] = {y_src}, *masks[1] = {mask_src};
  fg_plan *pl = nullptr;
  long nx = fg_plan_create(FG_CONSERVE_ORDER1, 1, &nx_src, &ny_src, lons, lats, masks, nx_dst, ny_dst, x_dst, y_dst, b1_device(), &pl);
  if (nx < 0) fatal(fg_last_error());
  if (nx >= (long)MAXXGRID)
    fatal("The xgrid size is too large for resources.\n nxgrid is greater than MAXXGRID/nthreads; increase MAXXGRID,\n"
          " decrease nthreads, or increase number of MPI ranks.");
  if (fg_plan_finalize(pl, nullptr)) fatal(fg_last_error());
  size_t ns = (size_t)nx_src * ny_src, nd = (size_t)nx_dst * ny_dst;
  double *dsrc = nullptr, *ddst = nullptr;
  if (hipMalloc(&dsrc, ns * sizeof(double)) != hipSuccess || hipMalloc(&ddst, nd * sizeof(double)) != hipSuccess) fatal("hipMalloc failed");
  if (hipMemcpy(dsrc, data_src, ns * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) fatal("hipMemcpy failed");
  if (plan_apply_frac(pl, dsrc, ddst)) fatal(fg_last_error());
  if (hipMemcpy(data_dst, ddst, nd * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) fatal("hipMemcpy failed");
  (void)hipFree(dsrc); (void)hipFree(ddst);
  fg_plan_destroy(pl);
}

// interp.c:312 -- the same with the great-circle exchange grid
extern "C" void conserve_interp_great_circle(int nx_src, int ny_src, int nx_dst, int ny_dst, const double *x_src,
                                             const double *y_src, const double *x_dst, const double *y_dst,
                                             const double *mask_src, const double *data_src, double *data_dst)
{
  const double *lons[1] = {x_src}, *lats[1] = {y_src}, *masks[1] = {mask_src};
  fg_plan *pl = nullptr;
  long nx = fg_plan_create_great_circle(1, &nx_src, &ny_src, lons, lats, masks, nx_dst, ny_dst, x_dst, y_dst, b1_device(), &pl);
  if (nx < 0) fatal(fg_last_error());
  if (nx > (long)MAXXGRID) fatal("nxgrid is greater than MAXXGRID, increase MAXXGRID");
  if (fg_plan_finalize(pl, nullptr)) fatal(fg_last_error());
  size_t ns = (size_t)nx_src * ny_src, nd = (size_t)nx_dst * ny_dst;
  double *dsrc = nullptr, *ddst = nullptr;
  if (hipMalloc(&dsrc, ns * sizeof(double)) != hipSuccess || hipMalloc(&ddst, nd * sizeof(double)) != hipSuccess) fatal("hipMalloc failed");
  if (hipMemcpy(dsrc, data_src, ns * sizeof(double), hipMemcpyHostToDevice) != hipSuccess) fatal("hipMemcpy failed");
  if (plan_apply_frac(pl, dsrc, ddst)) fatal(fg_last_error());
  if (hipMemcpy(data_dst, ddst, nd * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) fatal("hipMemcpy failed");
  (void)hipFree(dsrc); (void)hipFree(ddst);
  fg_plan_destroy(pl);
}

// ----------------------------------------------------------------------------- polygon primitives
// Batched device versions + the single-polygon libfrencutils symbols built on them.
namespace {
struct DevBuf {
  void *p = nullptr;
  DevBuf(size_t n) { if (hipMalloc(&p, n ? n : 1) != hipSuccess) p = nullptr; }
  ~DevBuf() { if (p) (void)hipFree(p); }
};
}

// lon/lat arrays are [npoly][FG_POLY_STRIDE] (= 24) host arrays; n1/n2 <= 12 vertices each.
// n_out[p] = vertex count, 0 (empty), -1 (parallel edges: the reference is fatal here) or -2 (too many vertices).
extern "C" int fg_clip_2dx2d_batch(int npoly, const double *lon1, const double *lat1, const int *n1,
                                   const double *lon2, const double *lat2, const int *n2,
                                   double *lon_out, double *lat_out, int *n_out)
{
  if (npoly < 0 || !lon1 || !lat1 || !n1 || !lon2 || !lat2 || !n2 || !lon_out || !lat_out || !n_out) return fail(FG_ERR_ARG, "null argument");
  if (npoly == 0) return 0;
  HIPCHK(hipSetDevice(b1_device()));
  size_t nb = (size_t)npoly * FG_POLY_STRIDE * sizeof(double), ni = (size_t)npoly * sizeof(int);
  DevBuf a(nb), b(nb), c(nb), d(nb), xo(nb), yo(nb), m1(ni), m2(ni), mo(ni);
  if (!a.p || !b.p || !c.p || !d.p || !xo.p || !yo.p || !m1.p || !m2.p || !mo.p) return fail(FG_ERR_HIP, "out of device memory");
  HIPCHK(hipMemcpy(a.p, lon1, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(b.p, lat1, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(c.p, lon2, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(d.p, lat2, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m1.p, n1, ni, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(m2.p, n2, ni, hipMemcpyHostToDevice));
  fgd_poly_clip(npoly, (double *)a.p, (double *)b.p, (int *)m1.p, (double *)c.p, (double *)d.p, (int *)m2.p,
                (double *)xo.p, (double *)yo.p, (int *)mo.p, 0);
  HIPCHK(hipDeviceSynchronize());
  HIPCHK(hipMemcpy(lon_out, xo.p, nb, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(lat_out, yo.p, nb, hipMemcpyDeviceToHost));
  HIPCHK(hipMemcpy(n_out, mo.p, ni, hipMemcpyDeviceToHost));
  return 0;
}

// op 0 poly_area, 1 poly_ctrlon(clon), 2 poly_ctrlat, 3 fix_lon(tlon = clon) in place (lon/lat/n updated).
extern "C" int fg_poly_op_batch(int op, int npoly, double *lon, double *lat, int *n, const double *clon, double *result)
{
  if (op < 0 || op > 3 || npoly < 0 || !lon || !lat || !n) return fail(FG_ERR_ARG, "bad argument");
  if ((op == 1 || op == 3) && !clon) return fail(FG_ERR_ARG, "clon/tlon array required");
  if (op != 3 && !result) return fail(FG_ERR_ARG, "result array required");
  if (npoly == 0) return 0;
  HIPCHK(hipSetDevice(b1_device()));
  size_t nb = (size_t)npoly * FG_POLY_STRIDE * sizeof(double), ni = (size_t)npoly * sizeof(int), nd = (size_t)npoly * sizeof(double);
  DevBuf a(nb), b(nb), m(ni), cl(nd), r(nd);
  if (!a.p || !b.p || !m.p || !cl.p || !r.p) return fail(FG_ERR_HIP, "out of device memory");
  HIPCHK(hipMemcpy(a.p, lon, nb, hipMemcpyHostToDevice)); HIPCHK(hipMemcpy(b.p, lat, nb, hipMemcpyHostToDevice));
  HIPCHK(hipMemcpy(m.p, n, ni, hipMemcpyHostToDevice));
  if (clon) HIPCHK(hipMemcpy(cl.p, clon, nd, hipMemcpyHostToDevice));
  fgd_poly_op(op, npoly, (double *)a.p, (double *)b.p, (int *)m.p, (double *)cl.p, (double *)r.p, 0);
  HIPCHK(hipDeviceSynchronize());
  if (op == 3) {
    HIPCHK(hipMemcpy(lon, a.p, nb, hipMemcpyDeviceToHost)); HIPCHK(hipMemcpy(lat, b.p, nb, hipMemcpyDeviceToHost));
    HIPCHK(hipMemcpy(n, m.p, ni, hipMemcpyDeviceToHost));
  } else HIPCHK(hipMemcpy(result, r.p, nd, hipMemcpyDeviceToHost));
  return 0;
}

// single-polygon libfrencutils symbols (create_xgrid.h:35-46, mosaic_util.h:95-101)
extern "C" int clip_2dx2d(const double lon1_in[], const double lat1_in[], int n1_in, const double lon2_in[],
                          const double lat2_in[], int n2_in, double lon_out[], double lat_out[])
{
  double a[FG_POLY_STRIDE] = {0}, b[FG_POLY_STRIDE] = {0}, c[FG_POLY_STRIDE] = {0}, d[FG_POLY_STRIDE] = {0};
  double xo[FG_POLY_STRIDE], yo[FG_POLY_STRIDE];
  if (n1_in > 12 || n2_in > 12 || n1_in < 1 || n2_in < 1) fatal("clip_2dx2d: this build supports 1..12 vertices per polygon");
  for (int k = 0; k < n1_in; k++) { a[k] = lon1_in[k]; b[k] = lat1_in[k]; }
  for (int k = 0; k < n2_in; k++) { c[k] = lon2_in[k]; d[k] = lat2_in[k]; }
  int no = 0;
  if (fg_clip_2dx2d_batch(1, a, b, &n1_in, c, d, &n2_in, xo, yo, &no)) fatal(fg_last_error());
  if (no == -1) fatal("the line between <x1_0,y1_0> and  <x1_1,y1_1> should not parallel to "
                      "the line between <x2_0,y2_0> and  <x2_1,y2_1>");
  if (no < 0) fatal("clip_2dx2d: clipped polygon exceeds 24 vertices");
  for (int k = 0; k < no; k++) { lon_out[k] = xo[k]; lat_out[k] = yo[k]; }
  return no;
}
static double poly_single(int op, const double x[], const double y[], int n, double clon)
{
  double a[FG_POLY_STRIDE] = {0}, b[FG_POLY_STRIDE] = {0}, r = 0;
  if (n < 1 || n > FG_POLY_STRIDE) fatal("polygon primitive: this build supports 1..24 vertices");
  for (int k = 0; k < n; k++) { a[k] = x[k]; b[k] = y[k]; }
  if (fg_poly_op_batch(op, 1, a, b, &n, &clon, &r)) fatal(fg_last_error());
  return r;
}
extern "C" double poly_area(const double x[], const double y[], int n) { return poly_single(0, x, y, n, 0.0); }
extern "C" double poly_ctrlon(const double x[], const double y[], int n, double clon) { return poly_single(1, x, y, n, clon); }
extern "C" double poly_ctrlat(const double x[], const double y[], int n) { return poly_single(2, x, y, n, 0.0); }
extern "C" int fix_lon(double x[], double y[], int n, double tlon)
{
  double a[FG_POLY_STRIDE] = {0}, b[FG_POLY_STRIDE] = {0};
  if (n < 1 || n > 8) fatal("fix_lon: this build supports 1..8 vertices");
  for (int k = 0; k < n; k++) { a[k] = x[k]; b[k] = y[k]; }
  if (fg_poly_op_batch(3, 1, a, b, &n, &tlon, nullptr)) fatal(fg_last_error());
  if (n < 0) fatal("fix_lon: vertex capacity exceeded");
  for (int k = 0; k < n; k++) { x[k] = a[k]; y[k] = b[k]; }
  return n;
}
extern "C" void pimod(double x[], int nn)
{
  const double PI = 3.14159265358979323846;
  for (int i = 0; i < nn; i++) { if (x[i] < -PI) x[i] += 2.0 * PI; else if (x[i] > PI) x[i] -= 2.0 * PI; }
}

// ----------------------------------------------------------------------------- order-2 input preparation
// fg_c2l: per-mosaic state for get_input_data's order-2 branch on the device (fregrid_util.c:2137-2216):
// halo fill across tile contacts, grad_c2l, gradient mask.  Geometry (calc_c2l_grid_info) is computed once on
// the host in the reference's operation order (c2l_host.c) and kept in HBM.
extern "C" int fg_c2l_grid_info(int nx, int ny, const double *xt, const double *yt, const double *xc, const double *yc,
                                double *dx, double *dy, double *area, double *edge_w, double *edge_e, double *edge_s,
                                double *edge_n, double *en_n, double *en_e, double *vlon, double *vlat);
extern "C" int fg_halo_map(int ntiles, const int *nx, const int *ny, int ncontacts, const int *tile1, const int *tile2,
                           const int *istart1, const int *iend1, const int *jstart1, const int *jend1,
                           const int *istart2, const int *iend2, const int *jstart2, const int *jend2,
                           long *map_off, int *map);

struct fg_c2l {
  int device = 0, ntiles = 0;
  hipStream_t stream = nullptr;
  hipStream_t stream_b = nullptr;   // second stream of the chunked search (always ours)
  bool own_stream = true;
  long ncells = 0, F = 0;
  std::vector<int> nx, ny;
  std::vector<void *> owned;
  void *tiles_dev = nullptr;
  int *map_dev = nullptr;
  int *cell_of_dev = nullptr;                     // halo'd element -> unpadded cell whose value it holds after update_halo (-1: zero)
  double *geom_dev[11] = {nullptr};
  std::vector<double> lont_halo, latt_halo;       // host copies (tests / inspection)
  template <typename T> T *alloc(size_t n) { void *p = g_pool.get(device, n * sizeof(T)); if (p) owned.push_back(p); return (T *)p; }
};

extern "C" void fg_c2l_destroy(fg_c2l *h)
{
  if (!h) return;
  (void)hipSetDevice(h->device);
  (void)hipStreamSynchronize(h->stream);
  if (h->own_stream) g_handles.put_stream(h->device, h->stream);
  for (void *p : h->owned) g_pool.put(p);
  delete h;
}

extern "C" int fg_c2l_create(int ntiles, const int *nx, const int *ny, const double *const *lonc, const double *const *latc,
                             const double *const *lont, const double *const *latt,
                             int ncontacts, const int *tile1, const int *tile2,
                             const int *istart1, const int *iend1, const int *jstart1, const int *jend1,
                             const int *istart2, const int *iend2, const int *jstart2, const int *jend2,
                             int device, fg_c2l **out)
{
  if (ntiles < 1 || !nx || !ny || !lonc || !latc || !lont || !latt || !out) return fail(FG_ERR_ARG, "null argument");
  int ndev = 0;
  HIPCHK(hipGetDeviceCount(&ndev));
  if (ndev < 1) return fail(FG_ERR_HIP, "no HIP device visible: libfregrid_hip needs an MI355X-class GPU");
  if (device < 0 || device >= ndev) return fail(FG_ERR_ARG, "device out of range");
  HIPCHK(hipSetDevice(device));
  fg_c2l *h = new fg_c2l();
  h->device = device; h->ntiles = ntiles;
  std::vector<long> cell_off(ntiles + 1), f_off(ntiles + 1), dx_off(ntiles + 1), dy_off(ntiles + 1), ew_off(ntiles + 1), es_off(ntiles + 1);
  long co = 0, fo = 0, xo = 0, yo = 0, wo = 0, so = 0;
  for (int t = 0; t < ntiles; t++) {
    if (nx[t] < 2 || ny[t] < 2) { delete h; return fail(FG_ERR_ARG, "tiles need at least 2x2 cells for the gradient"); }
    h->nx.push_back(nx[t]); h->ny.push_back(ny[t]);
    cell_off[t] = co; f_off[t] = fo; dx_off[t] = xo; dy_off[t] = yo; ew_off[t] = wo; es_off[t] = so;
    co += (long)nx[t] * ny[t]; fo += (long)(nx[t] + 2) * (ny[t] + 2);
    xo += (long)nx[t] * (ny[t] + 1); yo += (long)(nx[t] + 1) * ny[t]; wo += ny[t] + 1; so += nx[t] + 1;
  }
  h->ncells = co; h->F = fo;
  // halo map (setup_boundary + update_halo semantics)
  std::vector<long> map_off(ntiles + 1);
  std::vector<int> map(fo);
  int rc = fg_halo_map(ntiles, nx, ny, ncontacts, tile1, tile2, istart1, iend1, jstart1, jend1, istart2, iend2, jstart2, jend2,
                       map_off.data(), map.data());
  if (rc) { delete h; return fail(rc, "fregrid_util: inconsistent contact description (size mismatch between the boundary)"); }
  // halo'd cell centres: interior copy, zero halo (init_halo), then the same halo update as the data
  h->lont_halo.assign(fo, 0.0); h->latt_halo.assign(fo, 0.0);
  for (int t = 0; t < ntiles; t++)
    for (int j = 0; j < ny[t]; j++) for (int i = 0; i < nx[t]; i++) {
      h->lont_halo[f_off[t] + (long)(j + 1) * (nx[t] + 2) + i + 1] = lont[t][(long)j * nx[t] + i];
      h->latt_halo[f_off[t] + (long)(j + 1) * (nx[t] + 2) + i + 1] = latt[t][(long)j * nx[t] + i];
    }
  for (long e = 0; e < fo; e++) if (map[e] >= 0) { h->lont_halo[e] = h->lont_halo[map[e]]; h->latt_halo[e] = h->latt_halo[map[e]]; }
  // geometry per tile
  std::vector<double> dx(xo), dy(yo), area(co), ew(wo), ee(wo), es(so), en(so), enn(3 * xo), ene(3 * yo), vlon(3 * co), vlat(3 * co);
  {
    // a thread per tile: the tiles' outputs are disjoint, the arithmetic per cell is the host routine's (calc_c2l_grid_info,
    // gradient_c2l.c:368-454) either way -- 0.38 s for a C384 mosaic on one core
    std::vector<int> rcs(ntiles, 0);
    std::vector<std::thread> workers;
    auto one = [&](int t) {
      rcs[t] = fg_c2l_grid_info(nx[t], ny[t], h->lont_halo.data() + f_off[t], h->latt_halo.data() + f_off[t], lonc[t], latc[t],
                                dx.data() + dx_off[t], dy.data() + dy_off[t], area.data() + cell_off[t], ew.data() + ew_off[t],
                                ee.data() + ew_off[t], es.data() + es_off[t], en.data() + es_off[t], enn.data() + 3 * dx_off[t],
                                ene.data() + 3 * dy_off[t], vlon.data() + 3 * cell_off[t], vlat.data() + 3 * cell_off[t]);
    };
    for (int t = 1; t < ntiles; t++) workers.emplace_back(one, t);
    one(0);
    for (std::thread &w : workers) w.join();
    for (int t = 0; t < ntiles; t++) if (rcs[t]) { rc = rcs[t]; delete h; return fail(rc, "fg_c2l_grid_info failed"); }
  }
  h->stream = g_handles.get_stream(device);
  if (!h->stream) { delete h; return fail(FG_ERR_HIP, "hipStreamCreate failed"); }
  std::vector<char> th(fgd_c2l_tile_size() * ntiles);
  for (int t = 0; t < ntiles; t++) fgd_c2l_tile_fill(th.data(), t, nx[t], ny[t], cell_off[t], f_off[t], dx_off[t], dy_off[t], ew_off[t], es_off[t]);
  const std::vector<double> *src[11] = {&dx, &dy, &area, &ew, &ee, &es, &en, &enn, &ene, &vlon, &vlat};
  bool ok = true;
  h->tiles_dev = h->alloc<char>(th.size()); h->map_dev = h->alloc<int>(fo); h->cell_of_dev = h->alloc<int>(fo);
  ok = ok && h->tiles_dev && h->map_dev && h->cell_of_dev;
  // the gather map composed with the packing: which unpadded cell ends up in each halo'd element
  std::vector<int> cell_of(fo, -1);
  for (int t = 0; t < ntiles; t++)
    for (int j = 0; j < ny[t]; j++) for (int i = 0; i < nx[t]; i++)
      cell_of[f_off[t] + (long)(j + 1) * (nx[t] + 2) + i + 1] = (int)(cell_off[t] + (long)j * nx[t] + i);
  for (long e = 0; e < fo; e++) if (map[e] >= 0) cell_of[e] = cell_of[map[e]];     // sources are interior elements
  ok = ok && hipMemcpy(h->cell_of_dev, cell_of.data(), fo * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  for (int k = 0; k < 11 && ok; k++) { h->geom_dev[k] = h->alloc<double>(src[k]->size()); ok = ok && h->geom_dev[k]; }
  if (!ok) { fg_c2l_destroy(h); return fail(FG_ERR_HIP, "out of device memory"); }
  ok = ok && hipMemcpy(h->tiles_dev, th.data(), th.size(), hipMemcpyHostToDevice) == hipSuccess;
  ok = ok && hipMemcpy(h->map_dev, map.data(), fo * sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
  for (int k = 0; k < 11 && ok; k++) ok = hipMemcpy(h->geom_dev[k], src[k]->data(), src[k]->size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
  if (!ok) { fg_c2l_destroy(h); return fail(FG_ERR_HIP, "upload of the gradient geometry failed"); }
  *out = h;
  return 0;
}

extern "C" long fg_c2l_ncells(const fg_c2l *h) { return h ? h->ncells : FG_ERR_ARG; }
extern "C" long fg_c2l_halo_size(const fg_c2l *h) { return h ? h->F : FG_ERR_ARG; }
extern "C" int fg_c2l_set_stream(fg_c2l *h, void *stream)
{
  if (!h) return fail(FG_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  if (h->own_stream) g_handles.put_stream(h->device, h->stream);
  h->stream = (hipStream_t)stream; h->own_stream = false;
  return 0;
}
int fg_c2l_borrow_stream(fg_c2l *h, void *stream, void **saved, int *saved_own)      // (see fg_plan_borrow_stream)
{
  if (!h) return fail(FG_ERR_ARG, "null handle");
  HIPCHK(hipSetDevice(h->device));
  HIPCHK(hipStreamSynchronize(h->stream));
  *saved = (void *)h->stream; *saved_own = h->own_stream ? 1 : 0;
  h->stream = (hipStream_t)stream; h->own_stream = false;
  return 0;
}
void fg_c2l_return_stream(fg_c2l *h, void *saved, int saved_own)
{
  if (!h) return;
  h->stream = (hipStream_t)saved; h->own_stream = saved_own != 0;
}
extern "C" int fg_c2l_sync(fg_c2l *h)
{
  if (!h) return fail(FG_ERR_ARG, "null handle");
  HIPCHK(hipStreamSynchronize(h->stream));
  return 0;
}
// host copies of the halo'd cell-centre arrays [F] (for tests)
extern "C" int fg_c2l_get_centres(const fg_c2l *h, double *lont_halo, double *latt_halo)
{
  if (!h) return fail(FG_ERR_ARG, "null handle");
  if (lont_halo) memcpy(lont_halo, h->lont_halo.data(), h->F * sizeof(double));
  if (latt_halo) memcpy(latt_halo, h->latt_halo.data(), h->F * sizeof(double));
  return 0;
}

// src [nz][ncells] (tiles back to back, no halo) -> halo_data [nz][F] with the halo filled from the neighbours.
// src == NULL: halo_data already holds the interiors, only the halo is filled.  Device pointers.
extern "C" int fg_c2l_fill_halo(fg_c2l *h, const double *src, double *halo_data, int nz)
{
  if (!h || !halo_data || nz < 1) return fail(FG_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(h->device));
  if (src) {
    HIPCHK(hipMemsetAsync(halo_data, 0, (size_t)nz * h->F * sizeof(double), h->stream));   // init_halo (corners stay 0)
    fgd_pack_interior(h->tiles_dev, h->ntiles, h->ncells, h->F, nz, src, halo_data, h->stream);
  }
  fgd_halo_gather(h->F, nz, h->map_dev, halo_data, h->stream);
  HIPCHK(hipGetLastError());
  return 0;
}

// grad_c2l for every tile and level: halo_data [nz][F] -> grad_x, grad_y [nz][ncells]; grad_mask (int [nz][ncells],
// may be NULL) is filled when has_missing (fregrid_util.c:2203-2216), zeroed otherwise.  Device pointers.
extern "C" int fg_c2l_gradient(fg_c2l *h, const double *halo_data, int nz, int has_missing, double missing,
                               double *grad_x, double *grad_y, int *grad_mask)
{
  if (!h || !halo_data || !grad_x || !grad_y || nz < 1) return fail(FG_ERR_ARG, "bad argument");
  HIPCHK(hipSetDevice(h->device));
  fgd_grad_c2l(h->tiles_dev, h->ntiles, h->ncells, h->F, nz, halo_data, (const double *const *)h->geom_dev, grad_x, grad_y, h->stream);
  if (grad_mask) {
    if (has_missing) fgd_grad_mask(h->tiles_dev, h->ntiles, h->ncells, h->F, nz, halo_data, missing, grad_mask, h->stream);
    else HIPCHK(hipMemsetAsync(grad_mask, 0, (size_t)nz * h->ncells * sizeof(int), h->stream));
  }
  HIPCHK(hipGetLastError());
  return 0;
}

static int records_nb_pad(int nz) { return nz > 4 ? 8 : (nz > 2 ? 4 : 2); }

extern "C" int fg_c2l_records(fg_c2l *h, const double *src, int nz, double *rec)
{
  if (!h || !src || !rec) return fail(FG_ERR_ARG, "bad argument");
  if (nz < 1 || nz > 8) return fail(FG_ERR_ARG, "fg_c2l_records: 1 to 8 levels per call");
  HIPCHK(hipSetDevice(h->device));
  fgd_c2l_records(h->tiles_dev, h->ntiles, h->ncells, nz, records_nb_pad(nz), src, h->cell_of_dev, (const double *const *)h->geom_dev, rec, h->stream);
  HIPCHK(hipGetLastError());
  return 0;
}

extern "C" int fg_c2l_gradient_records(fg_c2l *h, const double *halo_data, int nz, double *rec)
{
  if (!h || !halo_data || !rec) return fail(FG_ERR_ARG, "bad argument");
  if (nz < 1 || nz > 8) return fail(FG_ERR_ARG, "fg_c2l_gradient_records: 1 to 8 levels per call");
  HIPCHK(hipSetDevice(h->device));
  fgd_grad_c2l_rec(h->tiles_dev, h->ntiles, h->ncells, h->F, nz, records_nb_pad(nz), halo_data, (const double *const *)h->geom_dev, rec, h->stream);
  HIPCHK(hipGetLastError());
  return 0;
}
