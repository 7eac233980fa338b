// sweep.hip -- fg_sweep: fields streamed from host memory through the plans (include/fregrid_hip.h, "streamed sweep").
//
// fregrid's per-field loop (tools/fregrid/fregrid.c:1001-1075) reads one hyperslab per tile on the host, widens it to
// double, scales it (get_input_data, fregrid_util.c:2036-2165), remaps it level by level and writes the result back in the
// file's type (write_field_data, :2339-2418).  With the sweep on the GPU the remap of eight levels takes 0.1-0.25 ms while the
// same levels need ~1 ms each way on a PCIe Gen5 x16 link: the link is the bound, so the job of this driver is to keep it busy
// in both directions at once and to move as few bytes as the file holds:
//   * levels travel in the FILE type (NC_FLOAT: half the bytes of a double) and are widened / narrowed on the device;
//   * three streams -- copy-in, compute, copy-out -- and three buffer slots: the upload of chunk k+1, the sweep of chunk k and
//     the download of chunk k-1 overlap; events order them, the host only waits when it wants a slot back;
//   * page-locked user buffers (fg_host_alloc) are used in place; pageable ones are staged through pinned slots.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>
#include <string>
#include <vector>
#include "fregrid_hip.h"

extern "C" int fg_plan_order(const fg_plan *pl);
extern "C" long fg_plan_ncells_out(const fg_plan *pl);
extern "C" int fg_plan_device(const fg_plan *pl);
void fg_set_last_error(const char *msg);         // plan.hip
// run-scoped use of our compute stream by the plans and the gradient object (plan.hip)
int fg_plan_borrow_stream(fg_plan *pl, void *stream, void **saved, int *saved_own);
void fg_plan_return_stream(fg_plan *pl, void *saved, int saved_own);
int fg_c2l_borrow_stream(fg_c2l *h, void *stream, void **saved, int *saved_own);
void fg_c2l_return_stream(fg_c2l *h, void *saved, int saved_own);

static int sw_fail(int code, const char *msg) { fg_set_last_error(msg); return code; }
#define SWCHK(call) do { hipError_t e_ = (call); if (e_ != hipSuccess) { char b_[256]; \
  snprintf(b_, sizeof b_, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); return sw_fail(FG_ERR_HIP, b_); } } while (0)

namespace {
constexpr int NSLOT = 3;       // buffer slots in flight
constexpr int CHUNK = 8;       // levels per chunk: what fg_c2l_records / fg_plan_apply_records take per call

size_t type_size(int t) { return t == FG_NC_SHORT ? 2 : (t == FG_NC_INT || t == FG_NC_FLOAT) ? 4 : t == FG_NC_DOUBLE ? 8 : 0; }

// get_input_data's conversion (fregrid_util.c:2097-2123): widen, then `if (scale != 0) data *= scale` and
// `if (offset != 0) data += offset`, each only where the value at that point differs from missing_value
template <typename T>
__global__ __launch_bounds__(256) void k_widen(long n, const T *in, double scale, double offset, double missing, double *out)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = (double)in[i];
  if (scale != 0 && v != missing) v *= scale;
  if (offset != 0 && v != missing) v += offset;
  out[i] = v;
}
// write_field_data's conversion (fregrid_util.c:2376-2406): `-= offset`, `/= scale` where != missing_value, then the C cast
// to the file type (nc_put_vara_double's for NC_FLOAT, the explicit (short) / (int) casts of the reference)
template <typename T>
__global__ __launch_bounds__(256) void k_narrow(long n, const double *in, double scale, double offset, double missing, T *out)
{
  const long i = (long)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  double v = in[i];
  if (offset != 0 && v != missing) v -= offset;
  if (scale != 0 && v != missing) v /= scale;
  out[i] = (T)v;
}

bool is_pinned(const void *p)
{
  hipPointerAttribute_t a;
  if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
  return a.type == hipMemoryTypeHost;
}
}  // namespace

// get_input_data's / write_field_data's conversions on device buffers, for callers that run the field loop themselves
// (integration/field_io_hip.c).  Synchronous (null stream + device sync).
extern "C" int fg_dev_widen(int nc_type, long n, const void *raw_dev, double scale, double offset, double missing, double *out_dev)
{
  if (n < 0 || (n > 0 && (!raw_dev || !out_dev)) || !type_size(nc_type)) return sw_fail(FG_ERR_ARG, "fg_dev_widen: bad argument");
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256);
  switch (nc_type) {
    case FG_NC_SHORT: k_widen<int16_t><<<grid, 256>>>(n, (const int16_t *)raw_dev, scale, offset, missing, out_dev); break;
    case FG_NC_INT: k_widen<int32_t><<<grid, 256>>>(n, (const int32_t *)raw_dev, scale, offset, missing, out_dev); break;
    case FG_NC_FLOAT: k_widen<float><<<grid, 256>>>(n, (const float *)raw_dev, scale, offset, missing, out_dev); break;
    default: k_widen<double><<<grid, 256>>>(n, (const double *)raw_dev, scale, offset, missing, out_dev); break;
  }
  SWCHK(hipGetLastError());
  SWCHK(hipDeviceSynchronize());
  return 0;
}
extern "C" int fg_dev_narrow(int nc_type, long n, const double *in_dev, double scale, double offset, double missing, void *out_dev)
{
  if (n < 0 || (n > 0 && (!in_dev || !out_dev)) || !type_size(nc_type)) return sw_fail(FG_ERR_ARG, "fg_dev_narrow: bad argument");
  if (n == 0) return 0;
  const int grid = (int)((n + 255) / 256);
  switch (nc_type) {
    case FG_NC_SHORT: k_narrow<int16_t><<<grid, 256>>>(n, in_dev, scale, offset, missing, (int16_t *)out_dev); break;
    case FG_NC_INT: k_narrow<int32_t><<<grid, 256>>>(n, in_dev, scale, offset, missing, (int32_t *)out_dev); break;
    case FG_NC_FLOAT: k_narrow<float><<<grid, 256>>>(n, in_dev, scale, offset, missing, (float *)out_dev); break;
    default: k_narrow<double><<<grid, 256>>>(n, in_dev, scale, offset, missing, (double *)out_dev); break;
  }
  SWCHK(hipGetLastError());
  SWCHK(hipDeviceSynchronize());
  return 0;
}

struct fg_sweep {
  int device = 0, order = 1, in_type = FG_NC_DOUBLE, out_type = FG_NC_DOUBLE;
  size_t in_sz = 8, out_sz = 8;
  std::vector<fg_plan *> plans;
  fg_c2l *c2l = nullptr;
  long ncin = 0, ndst_total = 0;
  std::vector<long> ndst, doff;          // cells per plan, offset of the plan's block inside a slot's output (in cells x CHUNK)
  hipStream_t s_in = nullptr, s_comp = nullptr, s_out = nullptr;
  double *d_f64 = nullptr, *d_rec = nullptr, *d_tmp = nullptr;   // shared by the chunks (compute stream is in order)
  struct Slot {
    void *d_raw = nullptr, *d_fin = nullptr;     // device: levels as they came from the host; outputs as they go back
    void *pin_in = nullptr, *pin_out = nullptr;  // staging for pageable user memory (allocated on first need)
    hipEvent_t e_in = nullptr, e_comp = nullptr, e_out = nullptr;
    bool busy = false;
    // deferred copy of staged outputs to the user's pageable arrays
    long l0 = 0; int nl = 0; bool staged_out = false;
  } slot[NSLOT];
};

extern "C" void *fg_host_alloc(size_t bytes)
{
  void *p = nullptr;
  if (hipHostMalloc(&p, bytes ? bytes : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
  return p;
}
extern "C" void fg_host_free(void *p) { if (p) (void)hipHostFree(p); }

extern "C" void fg_sweep_destroy(fg_sweep *sw)
{
  if (!sw) return;
  (void)hipSetDevice(sw->device);
  for (hipStream_t s : {sw->s_in, sw->s_comp, sw->s_out}) if (s) (void)hipStreamSynchronize(s);
  // (the plans and the gradient object are not touched: they use our compute stream only inside fg_sweep_run)
  for (auto &sl : sw->slot) {
    if (sl.d_raw) (void)hipFree(sl.d_raw);
    if (sl.d_fin) (void)hipFree(sl.d_fin);
    if (sl.pin_in) (void)hipHostFree(sl.pin_in);
    if (sl.pin_out) (void)hipHostFree(sl.pin_out);
    for (hipEvent_t e : {sl.e_in, sl.e_comp, sl.e_out}) if (e) (void)hipEventDestroy(e);
  }
  for (double *d : {sw->d_f64, sw->d_rec, sw->d_tmp}) if (d) (void)hipFree(d);
  for (hipStream_t s : {sw->s_in, sw->s_comp, sw->s_out}) if (s) (void)hipStreamDestroy(s);
  delete sw;
}

extern "C" int fg_sweep_create(int nplans, fg_plan *const *plans, fg_c2l *c2l, int in_type, int out_type, fg_sweep **out)
{
  if (nplans < 1 || !plans || !out) return sw_fail(FG_ERR_ARG, "fg_sweep_create: null argument");
  if (!type_size(in_type) || !type_size(out_type)) return sw_fail(FG_ERR_ARG, "fg_sweep_create: types must be FG_NC_SHORT, FG_NC_INT, FG_NC_FLOAT or FG_NC_DOUBLE");
  fg_sweep *sw = new fg_sweep();
  sw->in_type = in_type; sw->out_type = out_type; sw->in_sz = type_size(in_type); sw->out_sz = type_size(out_type);
  sw->order = fg_plan_order(plans[0]); sw->device = fg_plan_device(plans[0]); sw->ncin = fg_plan_ncells_in(plans[0]);
  for (int p = 0; p < nplans; p++) {
    if (!plans[p] || fg_plan_order(plans[p]) != sw->order || fg_plan_ncells_in(plans[p]) != sw->ncin || fg_plan_device(plans[p]) != sw->device) {
      delete sw; return sw_fail(FG_ERR_ARG, "fg_sweep_create: the plans must share order, source grid and device");
    }
    sw->plans.push_back(plans[p]);
    sw->doff.push_back(sw->ndst_total * CHUNK);
    sw->ndst.push_back(fg_plan_ncells_out(plans[p]));
    sw->ndst_total += sw->ndst.back();
  }
  if (sw->order == 2 && !c2l) { delete sw; return sw_fail(FG_ERR_ARG, "fg_sweep_create: conserve_order2 plans need the gradient object (fg_c2l)"); }
  if (sw->order == 2 && fg_c2l_ncells(c2l) != sw->ncin) { delete sw; return sw_fail(FG_ERR_ARG, "fg_sweep_create: the gradient object belongs to another grid"); }
  sw->c2l = (sw->order == 2) ? c2l : nullptr;
  if (hipSetDevice(sw->device) != hipSuccess) { delete sw; return sw_fail(FG_ERR_HIP, "hipSetDevice failed"); }
  bool ok = hipStreamCreateWithFlags(&sw->s_in, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&sw->s_comp, hipStreamNonBlocking) == hipSuccess &&
            hipStreamCreateWithFlags(&sw->s_out, hipStreamNonBlocking) == hipSuccess;
  const size_t nin = (size_t)CHUNK * sw->ncin, nout = (size_t)CHUNK * sw->ndst_total;
  for (auto &sl : sw->slot) {
    ok = ok && hipMalloc(&sl.d_raw, nin * sw->in_sz) == hipSuccess && hipMalloc(&sl.d_fin, nout * sw->out_sz) == hipSuccess;
    ok = ok && hipEventCreateWithFlags(&sl.e_in, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&sl.e_comp, hipEventDisableTiming) == hipSuccess &&
         hipEventCreateWithFlags(&sl.e_out, hipEventDisableTiming) == hipSuccess;
  }
  ok = ok && hipMalloc((void **)&sw->d_f64, nin * 8) == hipSuccess && hipMalloc((void **)&sw->d_tmp, nout * 8) == hipSuccess;
  if (sw->order == 2) ok = ok && hipMalloc((void **)&sw->d_rec, (size_t)sw->ncin * 3 * 8 * 8) == hipSuccess;
  if (!ok) { (void)hipGetLastError(); fg_sweep_destroy(sw); return sw_fail(FG_ERR_HIP, "fg_sweep_create: out of device memory (or stream / event creation failed)"); }
  *out = sw;
  return 0;
}

// wait for a slot's chunk to be back on the host, then hand staged outputs to the user's pageable arrays
static int slot_retire(fg_sweep *sw, fg_sweep::Slot &sl, void *const *host_out)
{
  if (!sl.busy) return 0;
  SWCHK(hipEventSynchronize(sl.e_out));
  if (sl.staged_out)
    for (size_t p = 0; p < sw->plans.size(); p++)
      memcpy((char *)host_out[p] + (size_t)sl.l0 * sw->ndst[p] * sw->out_sz, (char *)sl.pin_out + (size_t)sw->doff[p] * sw->out_sz,
             (size_t)sl.nl * sw->ndst[p] * sw->out_sz);
  sl.busy = false;
  return 0;
}

extern "C" int fg_sweep_run(fg_sweep *sw, const void *host_in, long nlev, double scale, double offset, double missing,
                            void *const *host_out)
{
  if (!sw || !host_in || !host_out || nlev < 1) return sw_fail(FG_ERR_ARG, "fg_sweep_run: null argument");
  for (size_t p = 0; p < sw->plans.size(); p++) if (!host_out[p]) return sw_fail(FG_ERR_ARG, "fg_sweep_run: null output array");
  SWCHK(hipSetDevice(sw->device));
  // The plans and the gradient object launch on OUR compute stream for the duration of this call only: several fg_sweep objects
  // (one per pair of file types) may share the same plans, a plan may be used directly between two runs, and either may be
  // destroyed before the other.  On every way out -- errors included -- the three streams are drained, the borrowed streams
  // handed back and the slots cleared, so that a later run never copies a failed run's staged outputs.
  struct Borrow {
    fg_sweep *sw; std::vector<void *> saved; std::vector<int> own; void *c_saved = nullptr; int c_own = 0; bool c_on = false;
    ~Borrow()
    {
      for (hipStream_t s : {sw->s_in, sw->s_comp, sw->s_out}) (void)hipStreamSynchronize(s);
      for (size_t p = 0; p < saved.size(); p++) fg_plan_return_stream(sw->plans[p], saved[p], own[p]);
      if (c_on) fg_c2l_return_stream(sw->c2l, c_saved, c_own);
      for (auto &sl : sw->slot) { sl.busy = false; sl.staged_out = false; }
    }
  } borrow{sw};
  for (fg_plan *p : sw->plans) {
    void *sv = nullptr; int ow = 0;
    if (fg_plan_borrow_stream(p, sw->s_comp, &sv, &ow)) return FG_ERR_HIP;
    borrow.saved.push_back(sv); borrow.own.push_back(ow);
  }
  if (sw->c2l) { if (fg_c2l_borrow_stream(sw->c2l, sw->s_comp, &borrow.c_saved, &borrow.c_own)) return FG_ERR_HIP; borrow.c_on = true; }
  const bool in_pinned = is_pinned(host_in);
  bool out_pinned = true;
  for (size_t p = 0; p < sw->plans.size(); p++) out_pinned = out_pinned && is_pinned(host_out[p]);
  const size_t nin = (size_t)CHUNK * sw->ncin, nout = (size_t)CHUNK * sw->ndst_total;
  const bool widen = sw->in_type != FG_NC_DOUBLE || scale != 0 || offset != 0;
  const bool narrow = sw->out_type != FG_NC_DOUBLE || scale != 0 || offset != 0;
  long chunk = 0;
  for (long l0 = 0; l0 < nlev; l0 += CHUNK, chunk++) {
    const int nl = (int)((nlev - l0 < CHUNK) ? nlev - l0 : CHUNK);
    fg_sweep::Slot &sl = sw->slot[chunk % NSLOT];
    { int rc = slot_retire(sw, sl, host_out); if (rc) return rc; }
    // --- upload (copy-in stream)
    const size_t bytes_in = (size_t)nl * sw->ncin * sw->in_sz;
    const char *src = (const char *)host_in + (size_t)l0 * sw->ncin * sw->in_sz;
    if (!in_pinned) {
      if (!sl.pin_in) SWCHK(hipHostMalloc(&sl.pin_in, nin * sw->in_sz, hipHostMallocDefault));
      memcpy(sl.pin_in, src, bytes_in);
      src = (const char *)sl.pin_in;
    }
    SWCHK(hipMemcpyAsync(sl.d_raw, src, bytes_in, hipMemcpyHostToDevice, sw->s_in));
    SWCHK(hipEventRecord(sl.e_in, sw->s_in));
    // --- widen, (gradients,) sweep, narrow (compute stream)
    SWCHK(hipStreamWaitEvent(sw->s_comp, sl.e_in, 0));
    const long n_in = (long)nl * sw->ncin;
    const double *f64 = (const double *)sl.d_raw;
    if (widen) {
      const int grid = (int)((n_in + 255) / 256);
      switch (sw->in_type) {
        case FG_NC_SHORT: k_widen<int16_t><<<grid, 256, 0, sw->s_comp>>>(n_in, (const int16_t *)sl.d_raw, scale, offset, missing, sw->d_f64); break;
        case FG_NC_INT: k_widen<int32_t><<<grid, 256, 0, sw->s_comp>>>(n_in, (const int32_t *)sl.d_raw, scale, offset, missing, sw->d_f64); break;
        case FG_NC_FLOAT: k_widen<float><<<grid, 256, 0, sw->s_comp>>>(n_in, (const float *)sl.d_raw, scale, offset, missing, sw->d_f64); break;
        default: k_widen<double><<<grid, 256, 0, sw->s_comp>>>(n_in, (const double *)sl.d_raw, scale, offset, missing, sw->d_f64); break;
      }
      f64 = sw->d_f64;
    }
    double *res = narrow ? sw->d_tmp : (double *)sl.d_fin;         // [plan block][level][cell]
    if (sw->order == 2) { int rc = fg_c2l_records(sw->c2l, f64, nl, sw->d_rec); if (rc) return rc; }
    for (size_t p = 0; p < sw->plans.size(); p++) {
      int rc = (sw->order == 2) ? fg_plan_apply_records(sw->plans[p], nl, sw->d_rec, res + sw->doff[p], nullptr)
                                : fg_plan_apply(sw->plans[p], f64, nullptr, nullptr, nullptr, 0, 0.0, nl, res + sw->doff[p], nullptr);
      if (rc) return rc;
    }
    if (narrow) {
      for (size_t p = 0; p < sw->plans.size(); p++) {
        const long n = (long)nl * sw->ndst[p];
        const int grid = (int)((n + 255) / 256);
        const double *in = sw->d_tmp + sw->doff[p];
        char *o = (char *)sl.d_fin + (size_t)sw->doff[p] * sw->out_sz;
        switch (sw->out_type) {
          case FG_NC_SHORT: k_narrow<int16_t><<<grid, 256, 0, sw->s_comp>>>(n, in, scale, offset, missing, (int16_t *)o); break;
          case FG_NC_INT: k_narrow<int32_t><<<grid, 256, 0, sw->s_comp>>>(n, in, scale, offset, missing, (int32_t *)o); break;
          case FG_NC_FLOAT: k_narrow<float><<<grid, 256, 0, sw->s_comp>>>(n, in, scale, offset, missing, (float *)o); break;
          default: k_narrow<double><<<grid, 256, 0, sw->s_comp>>>(n, in, scale, offset, missing, (double *)o); break;
        }
      }
    }
    SWCHK(hipGetLastError());
    SWCHK(hipEventRecord(sl.e_comp, sw->s_comp));
    // --- download (copy-out stream)
    SWCHK(hipStreamWaitEvent(sw->s_out, sl.e_comp, 0));
    if (!out_pinned && !sl.pin_out) SWCHK(hipHostMalloc(&sl.pin_out, nout * sw->out_sz, hipHostMallocDefault));
    for (size_t p = 0; p < sw->plans.size(); p++) {
      const size_t b = (size_t)nl * sw->ndst[p] * sw->out_sz;
      char *dst = out_pinned ? (char *)host_out[p] + (size_t)l0 * sw->ndst[p] * sw->out_sz : (char *)sl.pin_out + (size_t)sw->doff[p] * sw->out_sz;
      SWCHK(hipMemcpyAsync(dst, (const char *)sl.d_fin + (size_t)sw->doff[p] * sw->out_sz, b, hipMemcpyDeviceToHost, sw->s_out));
    }
    SWCHK(hipEventRecord(sl.e_out, sw->s_out));
    sl.busy = true; sl.l0 = l0; sl.nl = nl; sl.staged_out = !out_pinned;
  }
  // drain in chunk order
  for (long c = (chunk > NSLOT ? chunk - NSLOT : 0); c < chunk; c++) { int rc = slot_retire(sw, sw->slot[c % NSLOT], host_out); if (rc) return rc; }
  return 0;
}
