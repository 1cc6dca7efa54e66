// slslam_amd/csrc/lba_api.hip — C ABI of the LBA path (include/slslam_hip.h): host orchestration
// of the kernels in lba_kernels.h.  No CPU fallback: without a HIP device every compute entry
// point fails with SLSLAM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/slslam_hip.h"
#include "lba_kernels.h"
#include "lba_eliminate_mfma.h"
#include "lba_eliminate_grouped.h"
#define SLSLAM_PO_FACTOR_ONLY
#include "po_kernels.h"
#include "lba_big.h"
#include "lba_big_solve.h"
#include "device_cache.h"
#include "lba_motion_only.h"
#include "lba_pack.h"
#include "lba_device_build.h"
#include "host_pool.h"
#include "pinned_registry.h"

#include <functional>
#include <memory>
#if defined(__SSE2__)
#include <emmintrin.h>
#endif
#include <thread>

using namespace slslam;

#define HIP_TRY(expr)                                                                   \
  do {                                                                                  \
    hipError_t _e = (expr);                                                             \
    if (_e != hipSuccess) {                                                             \
      std::fprintf(stderr, "slslam: %s failed: %s (%s:%d)\n", #expr, hipGetErrorString(_e), __FILE__, __LINE__); \
      return (_e == hipErrorNoDevice || _e == hipErrorInvalidDevice) ? SLSLAM_ERR_NO_DEVICE : SLSLAM_ERR_HIP; \
    }                                                                                   \
  } while (0)

extern "C" void slslam_default_options(slslam_solver_options* o) {
  if (!o) return;
  o->max_num_iterations = 10;                 // FLAGS_max_num_iter, reference src/main.cpp:23
  o->huber_delta = 1.0 / 406.05;              // reference src/lba_problem.cpp:78-80, FLAGS_robust = true
  o->baseline = 0.12;                         // reference src/lba_problem.h:101
  o->initial_trust_region_radius = 1e4;       // Ceres 1.7 defaults below
  o->max_trust_region_radius = 1e16;
  o->min_trust_region_radius = 1e-32;
  o->min_relative_decrease = 1e-3;
  o->min_lm_diagonal = 1e-6;
  o->max_lm_diagonal = 1e32;
  o->max_num_consecutive_invalid_steps = 5;
  o->function_tolerance = 1e-6;
  o->gradient_tolerance = 1e-10;
  o->parameter_tolerance = 1e-8;
  o->jacobi_scaling = 1;
  o->use_graph = 1;
  o->chunks_per_window = 0;
  o->reuse_elimination = 0;
  o->po_factor_fp32 = 0;
  o->po_dense_factor = 0;
  o->lba_fused_motion_only = 1;
  o->lba_elimination = 0;
  o->lba_keep_jacobian = 0;
  o->refill_headroom_percent = 0;
  o->host_threads = 0;
  o->reproducible = 0;
  o->lba_precision = 0;
  o->device_build = 0;
}

extern "C" void slslam_release_cached_memory(void) { DeviceBlockCache::drop(); }

extern "C" int slslam_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

extern "C" const char* slslam_version(void) { return "slslam_amd 0.1 (gfx950)"; }

extern "C" const char* slslam_status_string(int s) {
  switch (s) {
    case SLSLAM_OK: return "ok";
    case SLSLAM_ERR_INVALID_ARGUMENT: return "invalid argument";
    case SLSLAM_ERR_NO_DEVICE: return "no usable HIP device";
    case SLSLAM_ERR_HIP: return "HIP runtime error";
    case SLSLAM_ERR_UNSUPPORTED: return "problem shape not supported by the kernels";
    case SLSLAM_ERR_STATE: return "invalid call sequence";
    case SLSLAM_ERR_NO_MEMORY: return "host allocation failed";
    default: return "unknown status";
  }
}

namespace {

template <typename T>
struct DevBuf {
  T* p = nullptr;
  size_t n = 0;
  bool in_arena = false;            // carved out of a DeviceArena: the arena frees it
  int alloc(size_t count) {
    n = count; in_arena = false;
    if (count == 0) { p = nullptr; return SLSLAM_OK; }
    HIP_TRY(hipMalloc((void**)&p, count * sizeof(T)));
    return SLSLAM_OK;
  }
  void release() { if (p && !in_arena) (void)hipFree(p); p = nullptr; n = 0; in_arena = false; }
};

// Results on the host: pageable, or pinned for a batch that is refilled (its downloads are asynchronous copies).
template <typename T>
struct HostArr {
  T* p = nullptr;
  size_t n = 0;
  bool pinned = false;
  std::vector<T> v;
  int alloc(size_t count, bool pin) {
    release();
    n = count; pinned = pin && count > 0;
    if (pinned) { HIP_TRY(hipHostMalloc((void**)&p, count * sizeof(T), hipHostMallocDefault)); std::memset((void*)p, 0, count * sizeof(T)); }
    else { v.assign(count, T()); p = v.data(); }
    return SLSLAM_OK;
  }
  void release() { if (pinned && p) (void)hipHostFree(p); p = nullptr; n = 0; pinned = false; std::vector<T>().swap(v); }
  T* data() { return p; }
  const T* data() const { return p; }
  size_t size() const { return n; }
  bool empty() const { return n == 0; }
  T& operator[](size_t i) { return p[i]; }
  const T& operator[](size_t i) const { return p[i]; }
};

// All device arrays of a batch in ONE allocation, everything the host fills in ONE host-to-device copy: the build of
// a single window (slslam_lba_solve, the per-keyframe call of the reference) is dominated by per-call driver overheads
// otherwise (~35 hipMalloc + ~20 synchronous hipMemcpy).
// The uploaded arrays have a host IMAGE with the same offsets (`stage`): the batch build writes every window's slices straight into
// it (fill_window, one window per host thread) and the image goes up in one copy.  A batch that is to be refilled
// (slslam_solver_options.refill_headroom_percent > 0) keeps the image, in pinned memory: slslam_lba_batch_refill packs the next set of
// windows into it and uploads the arrays with asynchronous copies on the caller's stream.
struct DeviceArena {
  struct Item { void** pp; size_t bytes; const void* host; size_t host_bytes; int group; size_t off; };   // group 0 uploaded, 1 scratch, 2 zeroed
  std::vector<Item> items;
  char* base = nullptr;
  size_t bytes = 0;
  int device = 0;
  bool cached = false;              // one-shot solves: the block comes from / returns to the thread's DeviceBlockCache
  char* stage = nullptr;            // host image of the uploaded arrays (offsets as on the device)
  size_t stage_bytes = 0;
  bool stage_pinned = false;        // hipHostMalloc'ed and kept for refills; otherwise a vector that lives until push()
  bool keep_stage = false;
  std::vector<char> stage_vec;
  template <typename T>
  void add(DevBuf<T>& d, size_t count, const T* host, size_t host_count, int group) {
    d.n = count; d.in_arena = true; d.p = nullptr;
    items.push_back(Item{ (void**)&d.p, count * sizeof(T), host, host_count * sizeof(T), group, 0 });
  }
  template <typename T> void upload(DevBuf<T>& d, const std::vector<T>& h) { add(d, h.size(), h.data(), h.size(), 0); }
  // uploaded, `count` elements of room; the caller writes host_of(d) between layout() and push()
  template <typename T> void staged(DevBuf<T>& d, size_t count) { add(d, count, (const T*)nullptr, 0, 0); }
  template <typename T> void scratch(DevBuf<T>& d, size_t count) { add(d, count, (const T*)nullptr, 0, 1); }
  template <typename T> void zeroed(DevBuf<T>& d, size_t count) { add(d, count, (const T*)nullptr, 0, 2); }
  template <typename T> T* host_of(const DevBuf<T>& d) const { return reinterpret_cast<T*>(stage + ((char*)d.p - base)); }
  size_t upload_end = 0, zero_begin = 0;
  int layout() {
    size_t off = 0;
    upload_end = 0; zero_begin = 0;
    for (int group = 0; group < 3; ++group) {        // uploaded arrays first, then scratch, then the zero-initialised ones
      if (group == 2) zero_begin = off;
      for (Item& it : items) {
        if (it.group != group) continue;
        it.off = off;
        off += (it.bytes + 255) & ~(size_t)255;
      }
      if (group == 0) upload_end = off;
    }
    bytes = off;
    if (off == 0) { items.clear(); return SLSLAM_OK; }
    if (cached) HIP_TRY(DeviceBlockCache::acquire(off, device, &base));
    else HIP_TRY(hipMalloc((void**)&base, off));
    if (keep_stage) {
      HIP_TRY(hipHostMalloc((void**)&stage, std::max<size_t>(upload_end, 256), hipHostMallocDefault));
      stage_pinned = true;
      std::memset(stage, 0, upload_end);
    } else {
      stage_vec.assign(upload_end, 0);
      stage = stage_vec.data();
    }
    stage_bytes = upload_end;
    for (const Item& it : items) {
      *it.pp = base + it.off;
      if (it.host && it.host_bytes) std::memcpy(stage + it.off, it.host, it.host_bytes);
    }
    items.clear();
    return SLSLAM_OK;
  }
  int push() {
    if (bytes == 0) return SLSLAM_OK;
    if (upload_end) HIP_TRY(hipMemcpy(base, stage, upload_end, hipMemcpyHostToDevice));
    if (bytes > zero_begin) HIP_TRY(hipMemset(base + zero_begin, 0, bytes - zero_begin));
    if (!stage_pinned) { std::vector<char>().swap(stage_vec); stage = nullptr; }
    return SLSLAM_OK;
  }
  int commit() { const int rc = layout(); return rc != SLSLAM_OK ? rc : push(); }
  void release() {
    if (base) { if (cached) DeviceBlockCache::give_back(base, bytes, device); else (void)hipFree(base); }
    if (stage_pinned && stage) (void)hipHostFree(stage);
    stage = nullptr; stage_pinned = false; std::vector<char>().swap(stage_vec);
    base = nullptr; items.clear();
  }
};

Policy make_policy(const slslam_solver_options& o) {
  Policy p;
  p.huber_delta = o.huber_delta; p.baseline = o.baseline;
  p.initial_radius = o.initial_trust_region_radius; p.max_radius = o.max_trust_region_radius;
  p.min_radius = o.min_trust_region_radius; p.min_relative_decrease = o.min_relative_decrease;
  p.min_lm_diagonal = o.min_lm_diagonal; p.max_lm_diagonal = o.max_lm_diagonal;
  p.function_tolerance = o.function_tolerance; p.gradient_tolerance = o.gradient_tolerance;
  p.parameter_tolerance = o.parameter_tolerance;
  p.max_num_iterations = o.max_num_iterations; p.max_invalid = o.max_num_consecutive_invalid_steps;
  p.jacobi_scaling = o.jacobi_scaling; p.keep_jacobian = 0;      // (set by finalize for the sweep that implements it)
  p.store_f = o.reuse_elimination ? 1 : 0;
  const char* dbg = std::getenv("SLSLAM_DEBUG_ABLATE");    // timing experiments of the elimination sweep; never set in production
  p.debug_flags = dbg ? std::atoi(dbg) : 0;
  return p;
}

enum { FAM_LIN = 0, FAM_SOLVE = 1, FAM_BACKSUB = 2, FAM_TRIG = 3, FAM_COST = 4, FAM_UPDATE = 5, FAM_INIT = 6, FAM_N = 8 };

}  // namespace

struct slslam_lba_batch {
  int device = 0;
  bool finalized = false;
  std::vector<PackedWindow> wins;
  slslam_solver_options opt;
  Policy pol;
  // host mirrors
  std::vector<WinDesc> h_wins;
  std::vector<long long> h_param_off;     // per window offset into the exported parameter vector
  long long total_params = 0;
  std::vector<LMState> h_state0;
  HostArr<LMState> h_state;
  HostArr<IterRec> h_trace;
  HostArr<double> h_params;
  // refills (slslam_lba_batch_refill): the device arrays have room for `refill_headroom_percent` more than the first windows needed; the
  // kernels' view of the batch (BatchPtrs: pointers, strides, counts) and with it the captured graph never change
  bool refillable = false;
  long long used_ncam = 0, used_nline = 0, used_nobs = 0, used_tiles = 0, used_items = 0;   // of the room the device arrays have
  int cap_maxC = 0, cap_maxn = 0;            // the LDS sizes of the launches were made for these
  int auto_rounds = 0, auto_cpw = 0;         // the automatic chunk policy as finalize resolved it: a refill cuts its windows the same way
  std::vector<int> sys_map_off_of_cf;        // per free-camera count: its table in d_sys_map, or -1
  hipEvent_t ev_stage_free = nullptr;        // recorded behind the uploads of a refill: the host image may be written again
  hipEvent_t ev_results = nullptr;           // recorded behind the copies of an asynchronous download
  bool results_pending = false;
  HostPool* ext_pool = nullptr;              // host threads lent by a stream object; else own_pool, made on demand
  std::unique_ptr<HostPool> own_pool;
  std::vector<PackedWindow> wins_spare;      // what a refill packs into; swapped with `wins` when the refill is accepted
  std::vector<int> h_ob_orig_off;          // per window offset into d_ob_orig
  bool downloaded = false;
  // device
  DeviceArena arena;
  DevBuf<uint16_t> d_sys_map;
  DevBuf<WinDesc> d_wins; DevBuf<Tile> d_tiles; DevBuf<Chunk> d_chunks; DevBuf<uint8_t> d_items; DevBuf<uint16_t> d_lane_map; DevBuf<int32_t> d_lane_ctx;
  DevBuf<uint32_t> d_line_desc;
  DevBuf<unsigned long long> d_dbg_cycles;
  int elim_mode = 0, elim_waves = 1;     // see BatchPtrs
  bool elim_mixed = false;               // ... its steady sweeps in mixed precision (lba_precision = 1: k_eliminate_grouped<false, false, true>)
  bool elim_grouped = false;             // elim_mode 1 with group-local accumulators (lba_eliminate_grouped.h); the windows are packed with grouping = 1
  size_t lds_elim = 0;
  DevBuf<unsigned long long> d_iter_counter;
  DevBuf<unsigned int> d_active;
  DevBuf<double> d_cam_x, d_cam_x0, d_cam_scale, d_cam_tab; DevBuf<int> d_cam_cf, d_cam_win;
  DevBuf<double> d_line_x, d_line_x0, d_line_scale; DevBuf<int> d_line_ptr, d_line_flags, d_line_win, d_line_orig;
  DevBuf<double> d_ob; DevBuf<int> d_ob_cam, d_ob_orig;
  DevBuf<double> d_ob_raw;                   // refillable batches: a refill's observations as the caller holds them, permuted into d_ob on the device (k_permute_obs)
  // the build stage on the device (lba_device_build.h): a refill whose windows go up as the caller holds them
  DevBuf<RawWin> d_rawwin; DevBuf<BuildWin> d_buildwin; DevBuf<uint32_t> d_raw_idx, d_fmask; DevBuf<double> d_line_raw; DevBuf<uint8_t> d_lflags;
  DevBuf<int> d_item_base, d_totals, d_line_pos;
  DevBuf<uint32_t> d_mid_keys; DevBuf<BuildLine> d_mid_li; DevBuf<uint4> d_mid_rows; DevBuf<uint16_t> d_mid_next, d_mid_trows, d_mid_tptr; DevBuf<BuildMid> d_mid;
  RawWin* h_rawwin = nullptr;                // pinned [B]: what the device reads of every window (uploaded per refill)
  BuildWin* h_buildwin = nullptr;            // pinned [B]: what came of every window (downloaded with the results)
  WinDesc* h_wins_dl = nullptr;              // pinned [B]: the descriptors the device made
  int* h_totals = nullptr;                   // pinned [8]
  char* h_raw_stage = nullptr;               // pinned, made on demand: pageable inputs are copied here (indices narrowed on the way)
  size_t raw_stage_bytes = 0;
  char* d_stage_in = nullptr;                // device, made on demand: where the copy engine puts a refill's arrays (k_ingest reads them from here)
  size_t d_stage_bytes = 0;
  std::vector<RawWin> host_src;              // per window: host-readable pointers to what the device was given (the callers' page-locked arrays, or the staging copy)
  int ingest_mode = 0;                       // of the last device-built refill: 0 zero-copy kernel reads of the callers' page-locked arrays, 1 copy engine
  bool ingest_pinned = false;                // ... its observations and parameters were read where the caller holds them (page-locked), no staging copy
  bool device_built = false;                 // the batch's present windows were built on the device
  bool inplace_export = false;               // ... and their `parameters` arrays are pinned: results can be written straight into them
  bool results_inplace = false;              // the last download wrote them there
  std::vector<slslam_lba_window> src_windows;   // the callers' descriptors of a device-built refill (fallback of flagged windows, in-place export)
  std::vector<int> build_status;             // per window, after wait(): 0 or the SLSLAM_ERR_* a flagged window is reported with
  long long n_device_builds = 0, n_zero_copy = 0;
  // windows beyond the tiled sweeps (lba_big.h)
  bool big_mode = false;
  BigPtrs big;
  std::vector<long long> h_big_sys_off, h_big_linv_off;
  DevBuf<int> d_big_ob_line, d_big_cam_ptr, d_big_cam_obs, d_big_pair_ptr, d_big_pair_row, d_big_pair_col, d_big_pair_desc, d_big_flags;
  DevBuf<double> d_big_J, d_big_F, d_big_cost, d_big_camtab, d_big_line_acc, d_big_sys, d_big_scal, d_big_linv;
  DevBuf<long long> d_big_sys_off;
  DevBuf<double> d_slab_sum;
  long long slab_sum_stride = 0;           // > 0: k_slab_reduce runs ahead of the reduced solve
  bool slab_sum_image = false;             // ... and writes the LDS image of the reduced solve (BatchPtrs.slab_sum_image)
  int line_elim_stride = kLineElim;
  DevBuf<double> d_slab, d_bs_part, d_cost_part, d_ysys, d_params_out, d_fstore, d_line_elim, d_line_h;
  DevBuf<LMState> d_state; DevBuf<IterRec> d_trace; DevBuf<long long> d_param_off;
  BatchPtrs ptrs;
  int nchunk = 0, nline = 0, ncam = 0;
  long long nobs = 0;
  int num_cus = 256;
  bool fused_motion_only = false;
  size_t lds_motion_only = 0;
  size_t lds_lin = 0, lds_solve = 0, lds_bs = 0, lds_bs_stream = 0, lds_cost = 0;
  // graph
  hipGraphExec_t graph_exec = nullptr;
  hipStream_t capture_stream = nullptr;
  // A batch that mixes oversize windows with ordinary ones is solved as TWO batches side by side: part[0] holds the ordinary
  // windows (tiled sweeps), part[1] the oversize ones (lba_big.h), the second on a stream of its own between a fork and a join on
  // the caller's.  This object then only routes: window i of the caller is window route[i].second of part[route[i].first].
  slslam_lba_batch* part[2] = { nullptr, nullptr };
  std::vector<std::pair<int, int>> route;
  std::vector<int> h_win_graded;             // per window: 0, or the number of slot rounds its graded chunk sizes were made for (finalize)
  std::vector<long long> part_param_off[2];  // per window of a part: where its parameters go in the caller's export layout
  DevBuf<long long> d_part_off[2];           // [3 nwin] per window of a part: offset in the part's export | offset in the caller's | length (k_scatter_windows)
  hipStream_t part_stream = nullptr;
  hipEvent_t part_fork = nullptr, part_join = nullptr;
  DevBuf<double> d_part_out[2];
  int num_windows() const { return part[0] ? (int)route.size() : (int)wins.size(); }
  // profiling
  bool profiling = false;
  double fam_ms[FAM_N] = { 0 };
  int fam_launches[FAM_N] = { 0 };
  std::vector<hipEvent_t> ev_pool;            // created once, reused by every profiled solve
  std::vector<std::pair<int, int>> ev_used;   // (family, index of start event); stop = start + 1
  size_t ev_next = 0;                         // first unused event of the pool
  // folds the recorded event pairs into fam_ms / fam_launches and frees them for reuse (waits for the last one)
  void harvest_events() {
    if (ev_used.empty()) { ev_next = 0; return; }
    (void)hipEventSynchronize(ev_pool[ev_used.back().second + 1]);
    for (const auto& u : ev_used) {
      float ms = 0.f;
      if (hipEventElapsedTime(&ms, ev_pool[u.second], ev_pool[u.second + 1]) == hipSuccess) { fam_ms[u.first] += ms; fam_launches[u.first]++; }
    }
    ev_used.clear(); ev_next = 0;
  }

  void release() {
    d_sys_map.release(); d_wins.release(); d_tiles.release(); d_chunks.release(); d_items.release(); d_lane_map.release(); d_lane_ctx.release(); d_line_desc.release(); d_dbg_cycles.release();
    d_cam_x.release(); d_cam_x0.release(); d_cam_scale.release(); d_cam_tab.release(); d_cam_cf.release(); d_cam_win.release();
    d_line_x.release(); d_line_x0.release(); d_line_scale.release(); d_line_ptr.release(); d_line_flags.release();
    d_line_win.release(); d_line_orig.release(); d_ob.release(); d_ob_cam.release(); d_ob_orig.release(); d_ob_raw.release();
    d_slab.release(); d_bs_part.release(); d_cost_part.release(); d_ysys.release(); d_params_out.release();
    d_state.release(); d_trace.release(); d_param_off.release(); d_iter_counter.release(); d_active.release();
    d_fstore.release(); d_line_elim.release(); d_line_h.release(); d_slab_sum.release();
    d_rawwin.release(); d_buildwin.release(); d_raw_idx.release(); d_fmask.release(); d_line_raw.release(); d_lflags.release(); d_item_base.release(); d_totals.release(); d_line_pos.release();
    d_mid_keys.release(); d_mid_li.release(); d_mid_rows.release(); d_mid_next.release(); d_mid_trows.release(); d_mid_tptr.release(); d_mid.release();
    if (h_rawwin) (void)hipHostFree(h_rawwin); if (h_buildwin) (void)hipHostFree(h_buildwin); if (h_wins_dl) (void)hipHostFree(h_wins_dl);
    if (h_totals) (void)hipHostFree(h_totals); if (h_raw_stage) (void)hipHostFree(h_raw_stage);
    if (d_stage_in) (void)hipFree(d_stage_in);
    d_stage_in = nullptr; d_stage_bytes = 0;
    h_rawwin = nullptr; h_buildwin = nullptr; h_wins_dl = nullptr; h_totals = nullptr; h_raw_stage = nullptr; raw_stage_bytes = 0;
    d_big_ob_line.release(); d_big_cam_ptr.release(); d_big_cam_obs.release(); d_big_pair_ptr.release(); d_big_pair_row.release();
    d_big_pair_col.release(); d_big_pair_desc.release(); d_big_flags.release(); d_big_J.release(); d_big_F.release(); d_big_cost.release();
    d_big_camtab.release(); d_big_line_acc.release(); d_big_sys.release(); d_big_scal.release(); d_big_linv.release(); d_big_sys_off.release();
    arena.release();
    h_state.release(); h_trace.release(); h_params.release();
    if (ev_stage_free) { (void)hipEventDestroy(ev_stage_free); ev_stage_free = nullptr; }
    if (ev_results) { (void)hipEventDestroy(ev_results); ev_results = nullptr; }
    if (graph_exec) { (void)hipGraphExecDestroy(graph_exec); graph_exec = nullptr; }
    if (capture_stream) { (void)hipStreamDestroy(capture_stream); capture_stream = nullptr; }
    for (hipEvent_t e : ev_pool) (void)hipEventDestroy(e);
    ev_pool.clear();
  }
};

extern "C" int slslam_lba_batch_create(int device, slslam_lba_batch** out) {
  if (!out) return SLSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return SLSLAM_ERR_NO_DEVICE; }
  if (device >= ndev) return SLSLAM_ERR_INVALID_ARGUMENT;
  slslam_lba_batch* b = new (std::nothrow) slslam_lba_batch();
  if (!b) return SLSLAM_ERR_HIP;
  b->device = device;
  slslam_default_options(&b->opt);
  *out = b;
  return SLSLAM_OK;
}

extern "C" void slslam_lba_batch_destroy(slslam_lba_batch* b) {
  if (!b) return;
  (void)hipSetDevice(b->device);
  for (int h = 0; h < 2; ++h) if (b->part[h]) { slslam_lba_batch_destroy(b->part[h]); b->part[h] = nullptr; b->d_part_out[h].release(); }
  if (b->part_stream) { (void)hipStreamSynchronize(b->part_stream); (void)hipStreamDestroy(b->part_stream); }
  if (b->part_fork) (void)hipEventDestroy(b->part_fork);
  if (b->part_join) (void)hipEventDestroy(b->part_join);
  b->release();
  delete b;
}

extern "C" int slslam_lba_batch_add(slslam_lba_batch* b, const slslam_lba_window* w, int* index) {
  if (!b || !w) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->finalized) return SLSLAM_ERR_STATE;
  PackedWindow pw;
  const int rc = pack_window(w, &pw);
  if (rc != SLSLAM_OK) return rc;
  if (index) *index = (int)b->wins.size();
  b->wins.push_back(std::move(pw));
  return SLSLAM_OK;
}

namespace {
// A mixed batch: the ordinary windows and the oversize ones become two batches of their own (same device, same options).
int finalize_mixed_impl(slslam_lba_batch* b);
// (a failed attempt leaves the batch as it was before the call: the windows go back to it in their order, the parts are destroyed)
int finalize_mixed(slslam_lba_batch* b) {
  if (b->part[0] || b->part[1]) return SLSLAM_ERR_STATE;
  const int rc = finalize_mixed_impl(b);
  if (rc == SLSLAM_OK) return rc;
  std::vector<PackedWindow> back(b->route.size());
  for (size_t i = 0; i < b->route.size(); ++i) {
    slslam_lba_batch* pb = b->part[b->route[i].first];
    if (pb && (size_t)b->route[i].second < pb->wins.size()) back[i] = std::move(pb->wins[(size_t)b->route[i].second]);
  }
  if (b->wins.empty() && !b->route.empty()) b->wins = std::move(back);
  for (int h = 0; h < 2; ++h) {
    if (b->part[h]) { slslam_lba_batch_destroy(b->part[h]); b->part[h] = nullptr; }
    b->d_part_out[h].release(); b->d_part_off[h].release(); b->part_param_off[h].clear();
  }
  if (b->part_stream) { (void)hipStreamDestroy(b->part_stream); b->part_stream = nullptr; }
  if (b->part_fork) { (void)hipEventDestroy(b->part_fork); b->part_fork = nullptr; }
  if (b->part_join) { (void)hipEventDestroy(b->part_join); b->part_join = nullptr; }
  b->route.clear(); b->finalized = false;
  return rc;
}
int finalize_mixed_impl(slslam_lba_batch* b) {
  int rc;
  for (int h = 0; h < 2; ++h)
    if ((rc = slslam_lba_batch_create(b->device, &b->part[h])) != SLSLAM_OK) return rc;
  long long off = 0;
  b->route.clear();
  for (PackedWindow& P : b->wins) {
    const int h = P.big ? 1 : 0;
    b->route.push_back({ h, (int)b->part[h]->wins.size() });
    b->part_param_off[h].push_back(off);
    off += 6LL * P.C + 4LL * P.L;
    b->part[h]->wins.push_back(std::move(P));
  }
  b->total_params = off;
  b->wins.clear();
  for (int h = 0; h < 2; ++h) {
    if ((rc = slslam_lba_batch_finalize(b->part[h], &b->opt)) != SLSLAM_OK) return rc;
    if ((rc = b->d_part_out[h].alloc((size_t)std::max<long long>(1, b->part[h]->total_params))) != SLSLAM_OK) return rc;
    const slslam_lba_batch* pb = b->part[h];
    const size_t nw = pb->h_param_off.size();
    std::vector<long long> offs(3 * std::max<size_t>(1, nw), 0);
    for (size_t i = 0; i < nw; ++i) {
      offs[3 * i] = pb->h_param_off[i]; offs[3 * i + 1] = b->part_param_off[h][i];
      offs[3 * i + 2] = (i + 1 < nw ? pb->h_param_off[i + 1] : pb->total_params) - pb->h_param_off[i];
    }
    if ((rc = b->d_part_off[h].alloc(offs.size())) != SLSLAM_OK) return rc;
    HIP_TRY(hipMemcpy(b->d_part_off[h].p, offs.data(), offs.size() * sizeof(long long), hipMemcpyHostToDevice));
  }
  HIP_TRY(hipStreamCreateWithFlags(&b->part_stream, hipStreamNonBlocking));
  HIP_TRY(hipEventCreateWithFlags(&b->part_fork, hipEventDisableTiming));
  HIP_TRY(hipEventCreateWithFlags(&b->part_join, hipEventDisableTiming));
  b->finalized = true;
  return SLSLAM_OK;
}
// the oversize part runs on the batch's own stream, between a fork from and a join to the caller's
template <typename Fn>
int on_both_parts(slslam_lba_batch* b, hipStream_t s, Fn fn) {
  HIP_TRY(hipEventRecord(b->part_fork, s));
  HIP_TRY(hipStreamWaitEvent(b->part_stream, b->part_fork, 0));
  // (the ordinary windows first: a long solve synchronises its stream every 16 iterations, and the part enqueued second only
  // starts once the first has been enqueued to the end)
  int rc = fn(b->part[0], (void*)s);
  if (rc == SLSLAM_OK) rc = fn(b->part[1], (void*)b->part_stream);
  HIP_TRY(hipEventRecord(b->part_join, b->part_stream));
  HIP_TRY(hipStreamWaitEvent(s, b->part_join, 0));
  return rc;
}
}  // namespace

namespace {

// ------------------------------------------------------------------------------------------
// The batch-wide layout of a set of packed windows.  plan_layout: where every window's slices go (prefix sums) and how its tiles are
// cut into chunks - serial and cheap.  fill_window: the slices themselves, written into the host image of the device arrays; windows
// are independent, one per host thread.  Used by slslam_lba_batch_finalize and by slslam_lba_batch_refill.
struct LayoutPlan {
  std::vector<WinDesc> wins;
  std::vector<long long> param_off;             // per window: offset into the exported parameter vector
  std::vector<int> ob_orig_off, item_base, win_graded;
  std::vector<Chunk> chunks;                    // dispatch order
  long long ncam = 0, nline = 0, nobs = 0, ntiles = 0, nitems = 0, sys = 0, slab = 0, params = 0;
  int maxC = 1, maxn = 0, max_chunks = 0, max_sys = 0;
};

int plan_layout(slslam_lba_batch* b, const std::vector<PackedWindow>& wins, bool frozen, LayoutPlan* out) {
  LayoutPlan& L = *out;
  const int B = (int)wins.size();
  long long total_tiles = 0;
  for (const PackedWindow& P : wins) total_tiles += (long long)P.tiles.size();
  if (!frozen) {
    // the sweeps run 8 one-wave workgroups per CU (LDS): the same number of chunks for every window, chosen so that
    // the batch fills whole rounds of the chip's wave slots with about 36 tiles per chunk at most
    // (1024 bench windows: 6 chunks of 33 tiles = 6144 waves = 3 rounds of 256 CUs x 8)
    const long long slots = (8LL / b->elim_waves) * b->num_cus;      // chunk workgroups resident per round
    const long long per_round = 36LL * b->elim_waves * slots;
    const long long rounds = std::max<long long>(1, (total_tiles + per_round - 1) / per_round);
    b->auto_rounds = (int)std::min<long long>(rounds, 1 << 20);
    b->auto_cpw = (int)std::min<long long>(std::max<long long>(1, (slots * rounds) / std::max(1, B)), 1 << 20);
  }
  L.wins.resize((size_t)B); L.param_off.resize((size_t)B); L.ob_orig_off.resize((size_t)B); L.item_base.resize((size_t)B);
  L.win_graded.assign((size_t)B, 0);
  std::vector<int> chunk_rank;           // per chunk (window-major): 0 dispatched in the first class, 1 in the second (graded sizes)
  std::vector<Chunk> chunks;
  for (int wi = 0; wi < B; ++wi) {
    const PackedWindow& P = wins[wi];
    WinDesc& wd = L.wins[wi];
    std::memset(&wd, 0, sizeof(wd));
    wd.C = P.C; wd.Cf = P.Cf; wd.L = P.L; wd.M = P.M;
    wd.cam_off = (int)L.ncam; wd.line_off = (int)L.nline; wd.obs_off = (int)L.nobs;
    wd.tile_off = (int)L.ntiles; wd.ntiles = (int)P.tiles.size();
    wd.n = 6 * P.Cf; wd.sys_off = (int)L.sys; wd.nfree_params = P.nfree_params; wd.nkept = P.nkept;
    L.maxC = std::max(L.maxC, P.C); L.maxn = std::max(L.maxn, wd.n);
    L.param_off[wi] = L.params; L.params += 6LL * P.C + 4LL * P.L;
    L.ob_orig_off[wi] = (int)L.nobs;
    L.item_base[wi] = (int)L.nitems;
    // chunks: runs of tiles handled by one wave.  Few long chunks keep the per-chunk partial of
    // the reduced system (a slab in HBM) small against the observation stream; many short chunks
    // fill the chip when the batch is small.
    int per_chunk;
    int graded_chunks = 0, graded_rounds = 0;             // > 0: graded chunk sizes (below)
    if (b->opt.chunks_per_window < 0) {                   // the caller asks for the graded cut a batch reported (slslam_lba_batch_window_chunks < 0):
      graded_rounds = (-b->opt.chunks_per_window) / 1000; graded_chunks = (-b->opt.chunks_per_window) % 1000;        // -(1000 rounds + chunks)
      if (graded_rounds < 2 || graded_chunks < graded_rounds) return SLSLAM_ERR_INVALID_ARGUMENT;
      per_chunk = 1;
    } else
    if (b->opt.chunks_per_window > 0) per_chunk = std::max(1, (wd.ntiles + b->opt.chunks_per_window - 1) / b->opt.chunks_per_window);
    else {
      // automatic: the batch-wide numbers above - or, with slslam_solver_options.reproducible, a function of the WINDOW alone (what the
      // automatic choice gives the windows of a batch that fills the chip: chunks of at most 34 tiles, graded for three rounds of the
      // wave slots), so that a window is cut the same way whatever batch it sits in
      long long rounds = b->auto_rounds, cpw = b->auto_cpw;
      if (b->opt.reproducible) { cpw = std::max(1, (wd.ntiles + 33) / 34); rounds = 3; }
      // (down to ONE tile per wave when the chip has room: since the camera tables are shared through memory, the partials of a
      // many-chunk window summed chip-wide (k_slab_reduce) and the first tile's loads requested ahead of the set-up, a second tile
      // costs a resident window more than a second chunk does - 0.44 / 0.60 / 0.98 -> 0.38 / 0.50 / 0.89 ms at W = 5 / 10 / 20,
      // round 4; rounds 2-3 kept at least two tiles per wave)
      per_chunk = (int)std::max<long long>(b->elim_waves, (wd.ntiles + cpw - 1) / cpw);
      // GRADED sizes when every wave slot runs several chunks (rounds >= 2) of many tiles.  A launch ends when the slowest slot has
      // finished its LAST chunk; with equal chunks - exactly `rounds` per slot, so nothing is left to balance with - that wait was a
      // fifth of the sweep (tools/chunk_timeline.py: 2048 slots 81 % busy, chunk durations 206-442 us around 297).  Now the chunks a
      // window contributes to the first round of the slots carry 70 % of a slot's share of the tiles, those of the second round 20 %
      // (30 % when there are only two), the rest what is left, and the chunk array - the dispatch order - lists the classes one after
      // the other: a slot that is late with its long chunk takes fewer short ones.  The slab count per window - what the reduced
      // solve reads - does not change.
      if (rounds >= 2 && cpw >= rounds && cpw < 1000 && b->elim_waves == 1 && wd.ntiles >= 8 * cpw && !std::getenv("SLSLAM_EQUAL_CHUNKS")) {
        graded_chunks = (int)cpw; graded_rounds = (int)rounds;
      }
    }
    int weights[1000];
    for (int c = 0; c < graded_chunks; ++c) {
      const int q = (int)(((long long)c * graded_rounds) / graded_chunks);           // the round of the slots this chunk belongs to
      weights[c] = q == 0 ? 84 : q == 1 ? (graded_rounds == 2 ? 36 : 24) : std::max(1, 12 / (graded_rounds - 2));
    }
    if (const char* ws = std::getenv("SLSLAM_CHUNK_WEIGHTS")) {           // (experiments: comma-separated weights of the window's chunks)
      int c = 0;
      for (const char* q = ws; *q && c < graded_chunks; ++c) { weights[c] = std::max(1, std::atoi(q)); while (*q && *q != ',') ++q; if (*q == ',') ++q; }
    }
    std::vector<int> bounds = graded_chunks > 0 ? chunk_boundaries_graded(wd.ntiles, graded_chunks, weights) : chunk_boundaries(wd.ntiles, per_chunk);
    L.win_graded[wi] = (graded_chunks > 0 && (int)bounds.size() - 1 == graded_chunks) ? graded_rounds : 0;
    if (b->big_mode) { bounds.assign(2, 0); }            // no tiles: one chunk per window carries its step statistics
    wd.chunk_off = (int)chunks.size(); wd.nchunks = (int)bounds.size() - 1;
    wd.slab_off = (int)L.slab;
    const int nsys = b->elim_mode == 1 ? sys_doubles_mfma(wd.n) : sys_doubles(wd.n);
    const long long slab_stride = (long long)nsys + kSlabScalars;
    L.max_chunks = std::max(L.max_chunks, wd.nchunks); L.max_sys = std::max(L.max_sys, nsys);
    for (int c = 0; c < wd.nchunks; ++c) {
      Chunk ck; ck.win = wi; ck.tile_begin = wd.tile_off + bounds[c]; ck.tile_end = wd.tile_off + bounds[c + 1];
      ck.slab_off = (int)L.slab; L.slab += slab_stride;
      ck.id = (int)chunks.size();
      chunks.push_back(ck);
      chunk_rank.push_back(L.win_graded[wi] ? (int)(((long long)c * graded_rounds) / graded_chunks) : 0);
    }
    if (L.slab > 0x7fffffffLL) return SLSLAM_ERR_UNSUPPORTED;
    L.ncam += P.C; L.nline += P.L; L.nobs += P.M; L.sys += wd.n;
    L.ntiles += (long long)P.tiles.size(); L.nitems += (long long)(P.items.size() / 2);
    if (L.nobs > 0x7fffffffLL || L.nitems > 0x7fffffffLL) return SLSLAM_ERR_UNSUPPORTED;
  }
  // dispatch order: the long chunks of the graded windows (and every chunk of the others) first, the short ones after them; inside a class
  // the order of the ids (a stable partition: ids, slabs and partial sums keep their window-major places)
  {
    int max_rank = 0;
    for (int r : chunk_rank) max_rank = std::max(max_rank, r);
    L.chunks.clear(); L.chunks.reserve(chunks.size());
    for (int r = 0; r <= max_rank; ++r) for (size_t i = 0; i < chunks.size(); ++i) if (chunk_rank[i] == r) L.chunks.push_back(chunks[i]);
  }
  return SLSLAM_OK;
}

// The host image of the uploaded arrays (DeviceArena::stage), by array.
struct HostImage {
  WinDesc* wins; Tile* tiles; Chunk* chunks; uint8_t* items; uint16_t* lane_map; uint32_t* line_desc;
  double* cam_x0; int* cam_cf; int* cam_win;
  double* line_u0; int* line_ptr; int* line_flags; int* line_win; int* line_orig;
  double* ob; long long ob_stride; int* ob_cam; int* ob_orig;
  long long* param_off;
};
HostImage host_image(slslam_lba_batch* b) {
  const DeviceArena& ar = b->arena;
  HostImage m;
  m.wins = ar.host_of(b->d_wins); m.tiles = ar.host_of(b->d_tiles); m.chunks = ar.host_of(b->d_chunks); m.items = ar.host_of(b->d_items);
  m.lane_map = ar.host_of(b->d_lane_map); m.line_desc = ar.host_of(b->d_line_desc);
  m.cam_x0 = ar.host_of(b->d_cam_x0); m.cam_cf = ar.host_of(b->d_cam_cf); m.cam_win = ar.host_of(b->d_cam_win);
  m.line_u0 = ar.host_of(b->d_line_x0); m.line_ptr = ar.host_of(b->d_line_ptr); m.line_flags = ar.host_of(b->d_line_flags);
  m.line_win = ar.host_of(b->d_line_win); m.line_orig = ar.host_of(b->d_line_orig);
  m.ob = ar.host_of(b->d_ob); m.ob_stride = (long long)(b->d_ob.n / 8); m.ob_cam = ar.host_of(b->d_ob_cam); m.ob_orig = ar.host_of(b->d_ob_orig);
  m.param_off = ar.host_of(b->d_param_off);
  return m;
}

// Window wi's slices of every uploaded array (hw: the windows' descriptors with their sys_map tables resolved).
void fill_window(const slslam_lba_batch* b, const LayoutPlan& plan, const std::vector<PackedWindow>& wins, int wi, const HostImage& m, bool copy_observations) {
  const PackedWindow& P = wins[(size_t)wi];
  const WinDesc& wd = b->h_wins[(size_t)wi];
  m.wins[wi] = wd;
  m.param_off[wi] = plan.param_off[(size_t)wi];
  const int item_base = plan.item_base[(size_t)wi];
  for (size_t t = 0; t < P.tiles.size(); ++t) {
    Tile x = P.tiles[t];
    x.line_begin += wd.line_off; x.item_off += item_base;
    m.tiles[(size_t)wd.tile_off + t] = x;
  }
  if (!P.items.empty()) std::memcpy(m.items + 2 * (size_t)item_base, P.items.data(), P.items.size());
  if (!P.lane_map.empty()) std::memcpy(m.lane_map + 64 * (size_t)wd.tile_off, P.lane_map.data(), P.lane_map.size() * sizeof(uint16_t));
  if (P.L > 0) {
    std::memcpy(m.line_desc + wd.line_off, P.line_desc.data(), (size_t)P.L * sizeof(uint32_t));
    std::memcpy(m.line_u0 + 4 * (size_t)wd.line_off, P.line_u.data(), (size_t)4 * P.L * sizeof(double));
    std::memcpy(m.line_flags + wd.line_off, P.line_flags.data(), (size_t)P.L * sizeof(int));
    std::memcpy(m.line_orig + wd.line_off, P.line_order.data(), (size_t)P.L * sizeof(int));
    for (int s2 = 0; s2 < P.L; ++s2) { m.line_ptr[wd.line_off + s2] = wd.obs_off + P.line_ptr[s2]; m.line_win[wd.line_off + s2] = wi; }
  }
  for (int c = 0; c < P.C; ++c) {
    for (int a = 0; a < 6; ++a) m.cam_x0[6 * (size_t)(wd.cam_off + c) + a] = P.cam_x[6 * (size_t)c + a];
    m.cam_cf[wd.cam_off + c] = P.cam_cf[c]; m.cam_win[wd.cam_off + c] = wi;
  }
  if (P.M > 0) {
    std::memcpy(m.ob_cam + wd.obs_off, P.ob_cam.data(), (size_t)P.M * sizeof(int));
    std::memcpy(m.ob_orig + wd.obs_off, P.ob_orig.data(), (size_t)P.M * sizeof(int));
    if (copy_observations)       // planes of (x,y) pairs, batch-wide: ob[(plane * ob_stride + o) * 2 + {0, 1}] (the packer's layout per window)
      for (int pl = 0; pl < 4; ++pl)
        std::memcpy(m.ob + ((size_t)pl * (size_t)m.ob_stride + (size_t)wd.obs_off) * 2, P.ob.data() + (size_t)pl * 2 * (size_t)P.M, sizeof(double) * 2 * (size_t)P.M);
  }
}

// What is not a window's: the chunk array, the end of the line pointers, and the marks on the unused records (a record without a
// window is skipped by the per-line / per-camera kernels: k_line_trig, k_export).
void fill_tail(const slslam_lba_batch* b, const LayoutPlan& plan, const HostImage& m) {
  if (!plan.chunks.empty()) std::memcpy(m.chunks, plan.chunks.data(), plan.chunks.size() * sizeof(Chunk));
  for (size_t c = plan.chunks.size(); c < b->d_chunks.n && c < (size_t)b->nchunk; ++c) { Chunk z; z.win = -1; z.tile_begin = 0; z.tile_end = 0; z.slab_off = 0; z.id = 0; m.chunks[c] = z; }
  m.line_ptr[plan.nline] = (int)plan.nobs;
  for (size_t ls = (size_t)plan.nline; ls < b->d_line_win.n; ++ls) m.line_win[ls] = -1;
  for (size_t c = (size_t)plan.ncam; c < b->d_cam_win.n; ++c) { m.cam_win[c] = -1; m.cam_cf[c] = -1; }
  if (plan.ntiles == 0) for (int q = 0; q < 64; ++q) m.lane_map[q] = (uint16_t)0x00FF;
  if (plan.nline == 0) m.line_flags[0] = 1;
}

// fn(i) for i in [0, n) on the pool's threads (or the calling one); false when some fn(i) threw - an allocation failure in a window's
// vectors -, which is caught where it happens: nothing unwinds across threads or through the extern "C" entry points
bool run_all(HostPool* pool, int n, const std::function<void(int)>& fn) {
  if (pool) return pool->run(n, fn);
  bool ok = true;
  for (int i = 0; i < n; ++i) { try { fn(i); } catch (...) { ok = false; } }
  return ok;
}

HostPool* batch_pool(slslam_lba_batch* b, int threads) {
  if (threads <= 1) return nullptr;
  if (b->ext_pool) return b->ext_pool;
  const int hw = (int)std::thread::hardware_concurrency();
  if (hw > 0) threads = std::min(threads, hw);
  if (threads <= 1) return nullptr;
  if (!b->own_pool || b->own_pool->threads() != threads) b->own_pool.reset(new HostPool(threads));
  return b->own_pool.get();
}

// One 16-byte record per lane and tile (BatchPtrs.lane_ctx): sorted line | first observation of the line | position in the run (bits 0-5),
// run length (6-12), lane has a line (13), skew (14), the line's constant flag (15), the tile's flags (16-23), line slot (24-31) | the
// tile's lane-th line descriptor - lane_map, line_ptr, line_flags and line_desc resolved once per batch.  thread <-> (tile, lane).
// A refill's observations arrive in the caller's order (one linear copy per window on the host, one upload): thread <-> sorted observation o of
// window blockIdx.y takes observation ob_orig[o] of that window's raw block and lays its four (x, y) pairs into the planes the sweeps stream.
// What pack_window's gather does on the host for a fresh batch - the same bytes - at the device's memory rate instead of a host core's.
__global__ __launch_bounds__(256) void k_permute_obs(BatchPtrs p, const double* raw, const int* ob_orig, double* ob_planes, const RawWin* src_of = nullptr) {
  const WinDesc wd = p.wins[blockIdx.y];
  const int o = blockIdx.x * 256 + threadIdx.x;
  if (o >= wd.M) return;
  const long long g = (long long)wd.obs_off + o;
  // (src_of: the window's observations lie where the copy engine put them - a device-built refill of page-locked or staged arrays)
  const double* base = src_of ? src_of[blockIdx.y].obs : raw + (long long)wd.obs_off * 8;
  const double* sp = base + (long long)ob_orig[g] * 8;
  double2 a, b, c, d;
  if ((reinterpret_cast<uintptr_t>(sp) & 15u) == 0) {
    const double2* src = reinterpret_cast<const double2*>(sp);
    a = src[0]; b = src[1]; c = src[2]; d = src[3];
  } else {
    a = make_double2(sp[0], sp[1]); b = make_double2(sp[2], sp[3]); c = make_double2(sp[4], sp[5]); d = make_double2(sp[6], sp[7]);
  }
  double2* out = reinterpret_cast<double2*>(ob_planes);
  out[g] = a; out[p.ob_stride + g] = b; out[2 * p.ob_stride + g] = c; out[3 * p.ob_stride + g] = d;
}

__global__ __launch_bounds__(256) void k_build_lane_ctx(BatchPtrs p, int ntiles, const int* ntiles_dev, int32_t* out) {
  if (ntiles_dev) ntiles = min(ntiles, *ntiles_dev);      // (a device-built refill: the launch covers the room the arrays have, the count is the device's)
  const long long g = (long long)blockIdx.x * 256 + threadIdx.x;
  if (g >= (long long)ntiles * 64) return;
  const int t = (int)(g >> 6), lane = (int)(g & 63);
  const Tile tl = p.tiles[t];
  const unsigned m = p.lane_map[g];
  const unsigned tflags = ((unsigned)tl.flags & 0xffu) << 16;
  unsigned misc = ((m & 0xffu) << 24) | (1u << 15) | tflags;            // slot | "constant" for an idle lane | the tile's flags
  int r0 = 0, r1 = 0;
  if ((m & 0xffu) != 0xffu) {
    const int ls = tl.line_begin + (int)(m & 0xffu);
    const int o0 = p.line_ptr[ls], k = p.line_ptr[ls + 1] - o0;
    r0 = ls; r1 = o0;
    misc = ((m >> 8) & 0x3fu) | ((unsigned)k & 0x7fu) << 6 | 1u << 13 | ((m >> 15) & 1u) << 14 | ((unsigned)p.line_flags[ls] & 1u) << 15 | tflags | (m & 0xffu) << 24;
  }
  const int r3 = lane < tl.nlines ? (int)p.line_desc[tl.line_begin + lane] : 0;
  reinterpret_cast<int4*>(out)[g] = make_int4(r0, r1, (int)misc, r3);
}

// After the uploaded arrays are on the device: the tile contexts, and both parameter buffers of cameras and lines + the LM state from the
// initial values (k_reset).  On `s`.
int device_init_after_upload(slslam_lba_batch* b, hipStream_t s) {
  const long long nt = b->used_tiles;
  if (nt > 0 && !b->big_mode)
    hipLaunchKernelGGL(k_build_lane_ctx, dim3((unsigned)((nt * 64 + 255) / 256)), dim3(256), 0, s, b->ptrs, (int)nt, (const int*)(b->device_built ? b->d_totals.p : nullptr), b->d_lane_ctx.p);
  const long long total = 6LL * b->ncam + (long long)b->nline + b->ptrs.nwin;      // one thread per line / camera parameter / window state
  if (total > 0) hipLaunchKernelGGL(k_reset, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, b->ptrs, b->pol);
  HIP_TRY(hipGetLastError());
  return SLSLAM_OK;
}

}  // namespace

extern "C" int slslam_lba_batch_finalize(slslam_lba_batch* b, const slslam_solver_options* opt) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->finalized) return SLSLAM_ERR_STATE;
  if (opt) b->opt = *opt;
  if (b->opt.max_num_iterations < 0 || b->opt.max_num_iterations > 100000) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!(b->opt.initial_trust_region_radius > 0.0) || !(b->opt.baseline == b->opt.baseline)) return SLSLAM_ERR_INVALID_ARGUMENT;
  b->pol = make_policy(b->opt);
  HIP_TRY(hipSetDevice(b->device));
  {
    int cus = 0;
    if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, b->device) == hipSuccess && cus > 0) b->num_cus = cus;
  }
  const int B = (int)b->wins.size();

  // ---- which elimination sweep: the matrix-core one (lba_eliminate_mfma.h) needs the reduced system in the registers of
  // one workgroup (<= 10 free cameras) and one observation per (line, free camera); everything else takes the LDS-atomic sweep
  {
    bool mfma_ok = !b->opt.reuse_elimination && b->opt.max_num_iterations > 0 && B > 0;
    for (const PackedWindow& P : b->wins) if (P.Cf > kMfmaMaxFree || P.dup_free_obs) mfma_ok = false;
    int want = b->opt.lba_elimination;
    if (want < 0 || want > 4 || b->opt.lba_precision < 0 || b->opt.lba_precision > 1) return SLSLAM_ERR_INVALID_ARGUMENT;
    // mixed precision lives in the grouped matrix-core sweep: asked for, it is that sweep or nothing
    b->elim_mixed = b->opt.lba_precision == 1;
    if (b->elim_mixed) {
      if (want == 0) want = 4;
      if (want != 4 || b->opt.reuse_elimination || b->opt.max_num_iterations <= 0) return SLSLAM_ERR_UNSUPPORTED;
    }
    // automatic = the LDS-atomic sweep: the matrix-core sweep measures slower on MI355X - not because of the matrix pipe (a wave
    // issues a v_mfma_f64_16x16x4_f64 every 64 cycles, tools/micro/mfma_f64_bench.hip) but because of its operand path through
    // LDS and its front end
    b->big_mode = false;
    int nbig = 0;
    for (const PackedWindow& P : b->wins) if (P.big) { b->big_mode = true; ++nbig; }
    if (nbig > 0 && nbig < B && !b->opt.reuse_elimination) return finalize_mixed(b);       // oversize windows apart (see `part`)
    if (b->big_mode) { mfma_ok = false; if (b->opt.reuse_elimination) return SLSLAM_ERR_UNSUPPORTED; }
    if (b->elim_mixed && !mfma_ok) return SLSLAM_ERR_UNSUPPORTED;
    // automatic (0): the grouped matrix-core sweep (lba_eliminate_grouped.h) for a batch that fills the chip with long chunks -
    // there it measures faster than the LDS-atomic sweep (1.20 against 1.27 ms per launch on the 1024 x 2000-line batch, DESIGN.md
    // section 7d); a small batch cuts its windows into chunks of a tile or two, for which the grouped sweep's per-chunk work
    // (zeroing its 20 KB of accumulator tiles in memory, adding the group sums into them) is not worth it
    long long tiles_in_batch = 0;
    for (const PackedWindow& P : b->wins) tiles_in_batch += (long long)P.tiles.size();
    // (reproducible: no choice by the batch - 0 means the LDS-atomic sweep, and a requested matrix-core sweep is run or refused)
    if (b->opt.reproducible && want >= 2 && !mfma_ok) return SLSLAM_ERR_UNSUPPORTED;
    const bool auto_grouped = want == 0 && !b->opt.reproducible && mfma_ok && b->opt.chunks_per_window == 0 && tiles_in_batch >= 16LL * 8 * b->num_cus;
    b->elim_mode = ((want >= 2 || auto_grouped) && mfma_ok) ? 1 : 0;
    b->elim_waves = b->elim_mode == 0 ? 1 : (want == 3 ? 2 : 1);
    b->elim_grouped = b->elim_mode == 1 && (want == 4 || auto_grouped);
    b->pol.keep_jacobian = (b->elim_grouped && !b->elim_mixed && b->opt.lba_keep_jacobian && b->opt.max_num_iterations > 1) ? 1 : 0;
    {
      // the grouped sweep wants the lines of a window in the order of their first free camera: pack again (the default packing
      // deals rows to the tiles by pair-item count, which this sweep has no use for).  Both directions: a batch whose finalize failed
      // after this point keeps its windows as they are now, and a second attempt with another sweep must not read line descriptors in
      // the other layout
      const int want_grouping = b->elim_grouped ? 1 : 0;
      std::vector<int> todo;
      for (int wi = 0; wi < B; ++wi) if (b->wins[(size_t)wi].grouping != want_grouping && !b->wins[(size_t)wi].big) todo.push_back(wi);
      std::vector<int> st(todo.size(), SLSLAM_OK);
      auto one = [&](int q) {
        PackedWindow Q;
        st[(size_t)q] = repack_window(b->wins[(size_t)todo[(size_t)q]], want_grouping, &Q);
        if (st[(size_t)q] == SLSLAM_OK) b->wins[(size_t)todo[(size_t)q]] = std::move(Q);
      };
      HostPool* pool = batch_pool(b, (int)std::min<size_t>(todo.size(), b->opt.host_threads > 0 ? (size_t)b->opt.host_threads : (todo.size() >= 64 ? 8 : 1)));
      if (!run_all(pool, (int)todo.size(), one)) return SLSLAM_ERR_NO_MEMORY;
      for (int r : st) if (r != SLSLAM_OK) return r;
    }
  }

  // ---- global layout: where every window's slices go (serial: prefix sums and chunk cuts), then the slices themselves (fill_window)
  LayoutPlan plan;
  {
    const int prc = plan_layout(b, b->wins, /*frozen=*/false, &plan);
    if (prc != SLSLAM_OK) return prc;
  }
  const long long ncam = plan.ncam, nline = plan.nline, nobs = plan.nobs, sys = plan.sys, slab = plan.slab, param_off = plan.params;
  const int maxC = plan.maxC, maxn = plan.maxn;
  b->h_wins = plan.wins; b->h_param_off = plan.param_off; b->h_ob_orig_off = plan.ob_orig_off; b->h_win_graded = plan.win_graded;
  if (b->h_win_graded.empty()) b->h_win_graded.assign(1, 0);
  b->nobs = nobs;
  // a batch that is going to be refilled (slslam_lba_batch_refill) gets room beyond what its first windows need: the next windows'
  // observation, line, tile and item counts differ a little
  b->refillable = b->opt.refill_headroom_percent > 0 && !b->big_mode;
  const long long pct = b->refillable ? std::min(b->opt.refill_headroom_percent, 400) : 0;
  auto room = [&](long long used) -> size_t { return (size_t)(b->refillable ? used + (used * pct) / 100 + 64 : used); };
  const size_t cap_cam = std::max<size_t>(1, room(ncam)), cap_line = std::max<size_t>(1, room(nline)), cap_obs = std::max<size_t>(1, room(nobs));
  const size_t cap_tiles = std::max<size_t>(1, room(plan.ntiles)), cap_items = std::max<size_t>(1, room(plan.nitems));
  if (cap_obs > 0x7fffffffULL || cap_line > 0x7fffffffULL) return SLSLAM_ERR_UNSUPPORTED;
  // reduced solve: where each entry of a chunk partial goes in its LDS image, one table per distinct system order (a refillable batch:
  // one per free-camera count up to the largest of its first windows - the next windows may have any of them)
  std::vector<uint16_t> sys_map(1, (uint16_t)0xFFFF);
  b->sys_map_off_of_cf.assign((size_t)maxn / 6 + 1, -1);
  if (!b->big_mode) {
    for (int cf = 0; cf <= maxn / 6; ++cf) {
      bool wanted = b->refillable;
      for (const WinDesc& wd : b->h_wins) if (wd.n == 6 * cf) wanted = true;
      if (!wanted) continue;
      b->sys_map_off_of_cf[(size_t)cf] = (int)sys_map.size();
      sys_map.resize(sys_map.size() + (size_t)sys_doubles(6 * cf));
      sys_map_build(6 * cf, sys_map.data() + b->sys_map_off_of_cf[(size_t)cf]);
    }
  }
  for (WinDesc& wd : b->h_wins) { const int off = b->sys_map_off_of_cf[(size_t)wd.n / 6]; wd.map_off = off < 0 ? 0 : off; }
  // (the chunk array too: a refill may cut its windows into a few more chunks; the launches cover the whole array, unused entries are marked)
  b->total_params = param_off; b->nchunk = (int)room((long long)plan.chunks.size()); b->nline = (int)cap_line; b->ncam = (int)cap_cam;
  b->used_ncam = ncam; b->used_nline = nline; b->used_nobs = nobs; b->used_tiles = plan.ntiles; b->used_items = plan.nitems;
  b->cap_maxC = maxC; b->cap_maxn = maxn;

  // ---- initial LM state (Ceres: LevenbergMarquardtStrategy ctor)
  b->h_state0.assign(B, LMState());
  for (int wi = 0; wi < B; ++wi) {
    LMState& st = b->h_state0[wi];
    std::memset(&st, 0, sizeof(st));
    st.radius = b->pol.initial_radius; st.decrease_factor = 2.0; st.status = kRunning; st.fresh = 1;
  }

  // ---- one allocation, one upload
  int rc;
  DeviceArena& ar = b->arena;
  ar.device = b->device;
  ar.keep_stage = b->refillable;
  ar.staged(b->d_wins, (size_t)std::max(1, B));
  ar.upload(b->d_sys_map, sys_map);
  ar.staged(b->d_tiles, cap_tiles);
  ar.staged(b->d_chunks, (size_t)std::max(1, b->nchunk));
  ar.staged(b->d_items, 2 * cap_items);
  ar.staged(b->d_lane_map, 64 * cap_tiles);
  ar.staged(b->d_line_desc, cap_line);
  // everything a lane has to know about its place in a tile, resolved once per batch (lba_kernels.h::fetch_tile): the sweeps get it with
  // ONE 16-byte load whose address depends on the tile index only - no chain tile -> lane map -> line pointer inside their loops.
  // Built on the DEVICE from the lane maps (k_build_lane_ctx, round 5: the host loop and the upload of 1 KB per tile were a fifth of
  // a batch build)
  ar.scratch(b->d_lane_ctx, 64 * 4 * cap_tiles);
  // both pose buffers and both line parameter buffers are filled on the device from the initial values (k_reset)
  ar.scratch(b->d_cam_x, 12 * cap_cam);
  ar.staged(b->d_cam_x0, 6 * cap_cam);
  ar.zeroed(b->d_cam_scale, std::max<size_t>(6, (size_t)6 * cap_cam));
  // rotation / Jacobian tables of both pose buffers, shared by the chunks of a window (BatchPtrs.cam_tab); the default sweeps only:
  // the other paths (reuse_elimination, streamed F, matrix-core sweep) keep building their tables per sweep.  Small batches only:
  // there a chunk is a tile or two and the tables are a tenth of a sweep; in a batch that fills the chip a chunk is ~33 tiles, the
  // tables are nothing, and reading them would add 30 MB to a sweep's traffic
  const bool share_cam_tab = B <= b->num_cus && !b->big_mode && !b->fused_motion_only && b->elim_mode == 0 && !b->opt.reuse_elimination && !b->pol.store_f && b->opt.max_num_iterations > 0 && !(b->pol.debug_flags & 16384);
  ar.zeroed(b->d_cam_tab, share_cam_tab ? std::max<size_t>(1, (size_t)2 * kCamTab * cap_cam) : 1);
  ar.staged(b->d_cam_cf, cap_cam);
  ar.staged(b->d_cam_win, cap_cam);
  // both parameter buffers, buffer-major [2][nline][kLineRec]: (a, b, g, t) | sin/cos table (filled on the device)
  ar.scratch(b->d_line_x, 2 * cap_line * kLineRec);
  ar.staged(b->d_line_x0, 4 * cap_line);
  ar.zeroed(b->d_line_scale, std::max<size_t>(4, (size_t)4 * cap_line));
  ar.staged(b->d_line_ptr, cap_line + 1);
  ar.staged(b->d_line_flags, cap_line);
  ar.staged(b->d_line_win, cap_line);
  ar.staged(b->d_line_orig, cap_line);
  ar.staged(b->d_ob, 8 * cap_obs);
  ar.scratch(b->d_ob_raw, b->refillable ? 8 * cap_obs : 0);
  {
    // what the build stage on the device needs beside the batch's own arrays (lba_device_build.h)
    const bool db = b->refillable && b->opt.device_build >= 0;
    ar.scratch(b->d_rawwin, db ? (size_t)std::max(1, B) : 0);
    ar.scratch(b->d_buildwin, db ? (size_t)std::max(1, B) : 0);
    ar.scratch(b->d_raw_idx, db ? cap_obs : 0);
    ar.scratch(b->d_fmask, db ? cap_line : 0);
    ar.scratch(b->d_line_raw, db ? 4 * cap_line : 0);
    ar.scratch(b->d_lflags, db ? cap_line : 0);
    ar.scratch(b->d_item_base, db ? (size_t)std::max(1, B) : 0);
    ar.scratch(b->d_totals, db ? 8 : 0);
    ar.scratch(b->d_line_pos, db ? cap_line : 0);
    ar.scratch(b->d_mid_keys, db ? cap_line : 0); ar.scratch(b->d_mid_li, db ? cap_line : 0); ar.scratch(b->d_mid_rows, db ? cap_line : 0);
    ar.scratch(b->d_mid_next, db ? cap_line : 0); ar.scratch(b->d_mid_trows, db ? cap_line + 8 * (size_t)std::max(1, B) : 0);
    ar.scratch(b->d_mid_tptr, db ? cap_line + 8 * (size_t)std::max(1, B) : 0); ar.scratch(b->d_mid, db ? (size_t)std::max(1, B) : 0);
  }
  ar.staged(b->d_ob_cam, cap_obs);
  ar.staged(b->d_ob_orig, cap_obs);
  ar.scratch(b->d_slab, std::max<size_t>(1, room(slab)));
  {
    // windows cut into many chunks (small batches): their partials are summed by a kernel of their own, spread over the chip
    const int max_chunks = plan.max_chunks, max_sys = plan.max_sys;
    b->slab_sum_stride = (max_chunks > 8 && !b->opt.reuse_elimination && !b->big_mode) ? (long long)max_sys + kSlabScalars : 0;
    // default sweeps: the sums go straight into the LDS image of the reduced solve (zeros where nothing is mapped: written
    // once, here)
    b->slab_sum_image = b->slab_sum_stride && b->elim_mode == 0;
    if (b->slab_sum_image) {
      long long ext = 0;
      for (const WinDesc& wd : b->h_wins) { const int N = solve_pad(wd.n); ext = std::max<long long>(ext, (long long)N * solve_stride(wd.n) + 6LL * N); }
      b->slab_sum_stride = ((ext + kSlabScalars + 1) / 2) * 2;
      ar.zeroed(b->d_slab_sum, (size_t)b->slab_sum_stride * (size_t)B);
    } else
    ar.scratch(b->d_slab_sum, b->slab_sum_stride ? (size_t)b->slab_sum_stride * (size_t)B : 1);
  }
  ar.scratch(b->d_bs_part, std::max<size_t>(1, (size_t)b->nchunk * kBsStride));
  ar.scratch(b->d_cost_part, std::max<size_t>(1, (size_t)b->nchunk));
  ar.scratch(b->d_ysys, std::max<size_t>(1, room(sys)));
  ar.scratch(b->d_fstore, b->pol.keep_jacobian ? (size_t)(24 * 64) * cap_tiles        // [tile][12][64] double2
                          : b->opt.reuse_elimination ? (size_t)24 * cap_obs : 2);
  ar.scratch(b->d_line_h, b->pol.keep_jacobian ? (size_t)10 * cap_line : 2);
  b->line_elim_stride = (b->opt.reuse_elimination || b->big_mode) ? kLineElim : kLeU;      // K g is kept for those two paths only
  ar.scratch(b->d_line_elim, std::max<size_t>(1, cap_line * b->line_elim_stride));
  ar.scratch(b->d_params_out, std::max<size_t>(1, room(param_off)));
  ar.upload(b->d_state, b->h_state0);
  ar.zeroed(b->d_trace, std::max<size_t>(1, (size_t)B * kMaxTrace));
  ar.staged(b->d_param_off, (size_t)std::max(1, B));
  std::vector<int> big_ob_line, big_cam_ptr, big_cam_obs, big_pair_ptr, big_pair_row, big_pair_col, big_pair_desc;
  if (b->big_mode) {
    // gather lists of the global-memory path (lba_big.h): per camera its observations, per (window, camera pair r >= c) the
    // observation pairs of the lines both cameras see - every sum is walked in list order, so results are reproducible
    if (B > 0xffff) return SLSLAM_ERR_UNSUPPORTED;
    // a camera-pair descriptor packs (window, row camera, column camera) as wi | r << 16 | c << 24, decoded with & 0xff (lba_big.h):
    // more than 127 free cameras would spill into the next field (and shift into the sign bit)
    for (const PackedWindow& P : b->wins) if (P.Cf > 127) return SLSLAM_ERR_UNSUPPORTED;
    long long sys_cursor = 0, linv_cursor = 0, oc = 0, lc = 0;
    b->h_big_sys_off.resize(B); b->h_big_linv_off.resize(B);
    big_ob_line.reserve((size_t)nobs);
    big_cam_ptr.assign((size_t)ncam + 1, 0);
    big_cam_obs.resize((size_t)std::max<long long>(1, nobs));
    std::vector<int> pr_row, pr_col, pr_id;                  // unsorted items with their global pair id
    {
      size_t items = 0, pairs = 0;
      for (const PackedWindow& P : b->wins) {
        pairs += (size_t)(P.Cf * (P.Cf + 1)) / 2;
        for (int s2 = 0; s2 < P.L; ++s2) {
          if (P.line_flags[s2] & 1) continue;
          const int o0 = P.line_ptr[s2], k = P.line_ptr[s2 + 1] - o0;
          int kf = 0;
          while (kf < k && P.cam_cf[P.ob_cam[o0 + kf]] >= 0) ++kf;
          items += (size_t)(kf * (kf - 1)) / 2;
        }
      }
      pr_row.reserve(items); pr_col.reserve(items); pr_id.reserve(items); big_pair_desc.reserve(pairs);
    }
    long long pair_base = 0;
    for (int wi = 0; wi < B; ++wi) {
      const PackedWindow& P = b->wins[wi];
      const int cam_off = b->h_wins[wi].cam_off;
      b->h_big_sys_off[wi] = sys_cursor; sys_cursor += big_sys_doubles(6 * P.Cf);
      b->h_big_linv_off[wi] = linv_cursor; linv_cursor += (long long)((6 * P.Cf + kNB - 1) / kNB + 1) * kNB * kNB;
      for (int o = 0; o < P.M; ++o) big_cam_ptr[(size_t)cam_off + P.ob_cam[o] + 1]++;
      for (int r = 0; r < P.Cf; ++r)
        for (int c = 0; c <= r; ++c) big_pair_desc.push_back(wi | (r << 16) | (c << 24));
      for (int s2 = 0; s2 < P.L; ++s2) {
        const int o0 = P.line_ptr[s2], k = P.line_ptr[s2 + 1] - o0;
        for (int j = 0; j < k; ++j) big_ob_line.push_back((int)lc + s2);
        if (P.line_flags[s2] & 1) continue;
        int kf = 0;                                        // free-camera observations come first
        while (kf < k && P.cam_cf[P.ob_cam[o0 + kf]] >= 0) ++kf;
        for (int i = 0; i < kf; ++i)
          for (int j = i + 1; j < kf; ++j) {
            const int ci = P.cam_cf[P.ob_cam[o0 + i]], cj = P.cam_cf[P.ob_cam[o0 + j]];
            const bool jr = cj >= ci;                      // row: the camera with the larger free index (lower triangle)
            const int r = jr ? cj : ci, c = jr ? ci : cj;
            pr_row.push_back((int)(oc + o0 + (jr ? j : i))); pr_col.push_back((int)(oc + o0 + (jr ? i : j)));
            pr_id.push_back((int)(pair_base + (r * (r + 1)) / 2 + c));
          }
      }
      pair_base += (long long)(P.Cf * (P.Cf + 1)) / 2;
      oc += P.M; lc += P.L;
    }
    for (size_t c = 0; c < (size_t)ncam; ++c) big_cam_ptr[c + 1] += big_cam_ptr[c];
    {
      std::vector<int> fill(big_cam_ptr.begin(), big_cam_ptr.end() - 1);
      long long og = 0;
      for (int wi = 0; wi < B; ++wi) {
        const PackedWindow& P = b->wins[wi];
        for (int o = 0; o < P.M; ++o, ++og) big_cam_obs[(size_t)fill[(size_t)b->h_wins[wi].cam_off + P.ob_cam[o]]++] = (int)og;
      }
    }
    const size_t npairs = big_pair_desc.size();
    big_pair_ptr.assign(npairs + 1, 0);
    for (int id : pr_id) big_pair_ptr[(size_t)id + 1]++;
    for (size_t q = 0; q < npairs; ++q) big_pair_ptr[q + 1] += big_pair_ptr[q];
    big_pair_row.resize(std::max<size_t>(1, pr_id.size())); big_pair_col.resize(std::max<size_t>(1, pr_id.size()));
    {
      std::vector<int> fill(big_pair_ptr.begin(), big_pair_ptr.end() - 1);          // counting sort: items of a pair stay in line order
      for (size_t q = 0; q < pr_id.size(); ++q) { const int at = fill[(size_t)pr_id[q]]++; big_pair_row[(size_t)at] = pr_row[q]; big_pair_col[(size_t)at] = pr_col[q]; }
    }
    if (big_ob_line.empty()) big_ob_line.push_back(0);
    b->big.npairs = (long long)npairs;
    if (big_pair_desc.empty()) big_pair_desc.push_back(0);
    ar.upload(b->d_big_ob_line, big_ob_line);
    ar.upload(b->d_big_cam_ptr, big_cam_ptr);
    ar.upload(b->d_big_cam_obs, big_cam_obs);
    ar.upload(b->d_big_pair_ptr, big_pair_ptr);
    ar.upload(b->d_big_pair_row, big_pair_row);
    ar.upload(b->d_big_pair_col, big_pair_col);
    ar.upload(b->d_big_pair_desc, big_pair_desc);
    ar.upload(b->d_big_sys_off, b->h_big_sys_off);
    ar.scratch(b->d_big_J, (size_t)std::max<long long>(1, nobs) * kBigObs);
    ar.scratch(b->d_big_F, (size_t)std::max<long long>(1, nobs) * kBigF);
    ar.scratch(b->d_big_cost, (size_t)std::max<long long>(1, nobs) * 2);
    ar.scratch(b->d_big_camtab, (size_t)std::max<long long>(1, ncam) * 2 * kBigCam);
    ar.zeroed(b->d_big_line_acc, (size_t)std::max<long long>(1, nline) * kBigLine);
    ar.zeroed(b->d_big_sys, (size_t)std::max<long long>(1, sys_cursor));
    ar.zeroed(b->d_big_scal, (size_t)std::max(1, B) * kBgScal);
    ar.zeroed(b->d_big_flags, (size_t)std::max(1, B) * 2);
    ar.scratch(b->d_big_linv, (size_t)std::max<long long>(1, linv_cursor));
  }
  ar.zeroed(b->d_iter_counter, 1);
  ar.zeroed(b->d_dbg_cycles, b->pol.debug_flags ? std::max((size_t)32 * std::max(1, b->nchunk), (size_t)16 * std::max(1, B)) : 1);
  ar.zeroed(b->d_active, 1);
  if ((rc = ar.layout())) return rc;
  {
    // every window's slices into the host image, one window per host thread (they are independent), then ONE upload
    HostImage img = host_image(b);
    HostPool* pool = batch_pool(b, (int)std::min<long long>(B, b->opt.host_threads > 0 ? b->opt.host_threads : (B >= 64 ? 8 : 1)));
    auto one = [&](int wi) { fill_window(b, plan, b->wins, wi, img, /*copy_observations=*/true); };
    if (!run_all(pool, B, one)) return SLSLAM_ERR_NO_MEMORY;
    fill_tail(b, plan, img);
  }
  if ((rc = ar.push())) return rc;

  BatchPtrs& p = b->ptrs;
  p.wins = b->d_wins.p; p.tiles = b->d_tiles.p; p.chunks = b->d_chunks.p; p.items = b->d_items.p; p.lane_map = b->d_lane_map.p; p.lane_ctx = b->d_lane_ctx.p;
  p.cam_x = b->d_cam_x.p; p.cam_scale = b->d_cam_scale.p; p.cam_cf = b->d_cam_cf.p;
  p.cam_tab = share_cam_tab ? b->d_cam_tab.p : nullptr;
  p.line_x = b->d_line_x.p; p.line_scale = b->d_line_scale.p; p.line_ptr = b->d_line_ptr.p;
  p.line_flags = b->d_line_flags.p; p.line_win = b->d_line_win.p;
  p.ob = b->d_ob.p; p.ob_cam = b->d_ob_cam.p; p.ob_stride = (long long)cap_obs;
  p.slab_sum = b->slab_sum_stride ? b->d_slab_sum.p : nullptr; p.slab_sum_stride = b->slab_sum_stride; p.slab_sum_image = b->slab_sum_image ? 1 : 0;
  p.slab = b->d_slab.p; p.bs_part = b->d_bs_part.p; p.cost_part = b->d_cost_part.p; p.ysys = b->d_ysys.p;
  p.fstore = b->d_fstore.p; p.line_elim = b->d_line_elim.p; p.line_elim_stride = b->line_elim_stride; p.line_h = b->d_line_h.p;
  p.state = b->d_state.p; p.trace = b->d_trace.p;
  p.iter_counter = b->d_iter_counter.p; p.active_counter = b->d_active.p; p.cam_x0 = b->d_cam_x0.p; p.line_u0 = b->d_line_x0.p;
  p.nwin = B; p.nchunk = b->nchunk; p.nline = b->nline; p.ncam = b->ncam;
  p.dbg_cycles = b->d_dbg_cycles.p; p.sys_map = b->d_sys_map.p;
  if (b->big_mode) {
    BigPtrs& g = b->big;
    g.ob_line = b->d_big_ob_line.p; g.cam_win = b->d_cam_win.p; g.J = b->d_big_J.p; g.camtab = b->d_big_camtab.p;
    g.line_acc = b->d_big_line_acc.p; g.sys = b->d_big_sys.p; g.sys_off = b->d_big_sys_off.p; g.F = b->d_big_F.p; g.cost = b->d_big_cost.p;
    g.cam_ptr = b->d_big_cam_ptr.p; g.cam_obs = b->d_big_cam_obs.p; g.pair_ptr = b->d_big_pair_ptr.p; g.pair_row = b->d_big_pair_row.p;
    g.pair_col = b->d_big_pair_col.p; g.pair_desc = b->d_big_pair_desc.p; g.scal = b->d_big_scal.p; g.flags = b->d_big_flags.p; g.nobs = nobs;
  }
  p.line_desc = b->d_line_desc.p; p.elim_mode = b->elim_mode; p.elim_waves = b->elim_waves;
  b->lds_elim = b->elim_grouped ? (size_t)lds_bytes_eliminate_grouped(maxC, maxn) : (size_t)lds_bytes_eliminate_mfma(maxC, maxn, b->elim_waves);

  b->lds_lin = sizeof(double) * (size_t)lds_doubles_linearise(maxC, maxn);
  b->lds_solve = sizeof(double) * (size_t)lds_doubles_solve(maxn);
  // motion-only batches (one free camera, every line constant: SLAM::motion_only_ba): the whole solve in one launch
  b->fused_motion_only = b->opt.lba_fused_motion_only && !b->opt.reuse_elimination && B > 0;
  for (const PackedWindow& P : b->wins) if (P.Cf != 1 || P.nfree_params != 6) b->fused_motion_only = false;
  b->lds_motion_only = sizeof(double) * (size_t)lds_doubles_motion_only(maxC);
  b->lds_bs = sizeof(double) * (size_t)lds_doubles_backsub(maxC, maxn);
  b->lds_bs_stream = sizeof(double) * (size_t)lds_doubles_backsub_stream(maxC, maxn);
  b->lds_cost = sizeof(double) * (size_t)lds_doubles_cost(maxC, maxn);
  const size_t lds_max = std::max(std::max(b->lds_lin, b->lds_solve), std::max(b->lds_bs, b->lds_cost));
  if (lds_max > 160 * 1024 && !b->big_mode) return SLSLAM_ERR_UNSUPPORTED;
  if (b->lds_lin > 48 * 1024 && !b->big_mode) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_linearise_schur<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_lin));
    HIP_TRY(hipFuncSetAttribute((const void*)k_linearise_schur<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_lin));
    HIP_TRY(hipFuncSetAttribute((const void*)k_linearise_schur<false, 0>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_lin));
    HIP_TRY(hipFuncSetAttribute((const void*)k_linearise_schur<false, 1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_lin));
  }
  if (b->elim_mode == 1 && b->lds_elim > 48 * 1024) {
    if (b->lds_elim > 160 * 1024) return SLSLAM_ERR_UNSUPPORTED;
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_mfma<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_mfma<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_mfma<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_mfma<2, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_grouped<false>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_grouped<true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_grouped<false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
    HIP_TRY(hipFuncSetAttribute((const void*)k_eliminate_grouped<false, false, true>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_elim));
  }
  if (b->lds_solve > 48 * 1024 && !b->big_mode) {
    HIP_TRY(hipFuncSetAttribute((const void*)k_reduced_solve<4>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_solve));
    HIP_TRY(hipFuncSetAttribute((const void*)k_reduced_solve<1>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)b->lds_solve));
  }
  // on the device: the tile contexts from the lane maps, both parameter buffers of cameras and lines from the initial values
  if ((rc = device_init_after_upload(b, nullptr))) return rc;
  if (!ar.cached) HIP_TRY(hipStreamSynchronize(nullptr));      // (a one-shot solve stays on the null stream: ordered behind this anyway)
  if ((rc = b->h_state.alloc((size_t)B, b->refillable))) return rc;
  if ((rc = b->h_trace.alloc((size_t)B * kMaxTrace, b->refillable))) return rc;
  if ((rc = b->h_params.alloc(room(param_off), b->refillable))) return rc;
  b->finalized = true;
  return SLSLAM_OK;
}

namespace {

// Launch helper: optional event bracketing per kernel family.
struct Launcher {
  slslam_lba_batch* b;
  hipStream_t s;
  bool prof;
  int begin(int fam) {
    if (!prof) return SLSLAM_OK;
    if (b->ev_next + 2 > b->ev_pool.size()) {          // the pool only grows until it covers one harvest interval
      hipEvent_t e0, e1;
      if (hipEventCreate(&e0) != hipSuccess) return SLSLAM_ERR_HIP;
      if (hipEventCreate(&e1) != hipSuccess) { (void)hipEventDestroy(e0); return SLSLAM_ERR_HIP; }
      b->ev_pool.push_back(e0); b->ev_pool.push_back(e1);
    }
    b->ev_used.push_back({ fam, (int)b->ev_next });
    b->ev_next += 2;
    if (hipEventRecord(b->ev_pool[b->ev_used.back().second], s) != hipSuccess) return SLSLAM_ERR_HIP;
    return SLSLAM_OK;
  }
  int end() {
    if (!prof) return SLSLAM_OK;
    if (hipEventRecord(b->ev_pool[b->ev_used.back().second + 1], s) != hipSuccess) return SLSLAM_ERR_HIP;
    return SLSLAM_OK;
  }
};

// windows beyond the tiled sweeps whose reduced systems all fit k_big_solve (lba_big_solve.h): no memsets, no per-window launch
// loop - the solve is a fixed sequence of launches like the tiled path's, and can be captured
bool big_one_launch(const slslam_lba_batch* b) {
  if (!b->big_mode || (b->pol.debug_flags & 2048)) return false;
  for (const WinDesc& wd : b->h_wins) if (wd.n > kBsvMaxN) return false;
  return true;
}

int enqueue_solve(slslam_lba_batch* b, hipStream_t s, bool prof, bool capturing = false) {
  const BatchPtrs& p = b->ptrs;
  const Policy& pol = b->pol;
  const int B = p.nwin;
  if (B == 0) return SLSLAM_OK;
  Launcher L{ b, s, prof };
  const dim3 blk64(64), blk256(256);
  const dim3 g_chunk((unsigned)std::max(1, b->nchunk)), g_win((unsigned)B), g_line((unsigned)((b->nline + 255) / 256)),
      g_upd((unsigned)((B + 63) / 64));
  int rc;
#define LAUNCH(fam, ...)                                   \
  do {                                                     \
    if ((rc = L.begin(fam))) return rc;                    \
    __VA_ARGS__;                                           \
    if ((rc = L.end())) return rc;                         \
  } while (0)
  // (the sin / cos table of the accepted line parameters is k_reset's: every solve that has anything to do follows one - finalize, refill or reset)
  if (b->fused_motion_only && !b->big_mode) {
    LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_motion_only, g_win, blk64, b->lds_motion_only, s, p, pol));
    HIP_TRY(hipGetLastError());
    return SLSLAM_OK;
  }
  if (b->big_mode) {
    // windows beyond the tiled sweeps: everything in HBM / L2 (lba_big.h), the reduced system on the pose-graph path's
    // blocked MFMA Cholesky
    const BigPtrs& g = b->big;
    const dim3 blk128(128);
    const dim3 g_obs128((unsigned)((g.nobs + 127) / 128)), g_obs256((unsigned)((g.nobs + 255) / 256)), g_line128((unsigned)((b->nline + 127) / 128)), g_linew((unsigned)((b->nline + 3) / 4)),
        g_cam((unsigned)((b->ncam + 255) / 256)), g_pair((unsigned)std::max<long long>(1, g.npairs)), g_camwg((unsigned)std::max(1, b->ncam));
    const int iters = std::max(1, pol.max_num_iterations);      // (max_num_iterations = 0: the first sweep's initial evaluation only)
    // reduced systems of at most 256 unknowns (the reference's W = 40: 240) are factorised and solved by one workgroup in one
    // launch, from registers (lba_big_solve.h): nothing dirties the system in memory, which every sweep rebuilds by stores
    int max_n = 0;
    for (int wi = 0; wi < B; ++wi) max_n = std::max(max_n, b->h_wins[wi].n);
    const bool one_launch_solve = big_one_launch(b);
    for (int it = 0; it < iters; ++it) {
      if (!capturing && it > 0 && (it % 16) == 0) {
        unsigned int active = 0;
        HIP_TRY(hipMemcpyAsync(&active, b->d_active.p, sizeof(active), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        if (active == 0) break;
      }
      if (!capturing && ((it + 1) % 16) == 0) HIP_TRY(hipMemsetAsync(b->d_active.p, 0, sizeof(unsigned int), s));
      // (the reduced system is rebuilt by stores: only what the factorisation leaves behind - its upper triangle - and the
      // failure flags need clearing; the per-line / per-observation outputs are overwritten)
      if (!one_launch_solve) {
        HIP_TRY(hipMemsetAsync(b->d_big_sys.p, 0, b->d_big_sys.n * sizeof(double), s));
        HIP_TRY(hipMemsetAsync(b->d_big_flags.p, 0, b->d_big_flags.n * sizeof(int), s));
      }
      // (the launch-chain variant keeps the round-2 sequence of one kernel per stage; the default folds the stages a wave can
      // do for its own line into k_big_line / k_big_backsub_line and the table refresh into k_big_reduce: every kernel of this
      // path runs for a few microseconds and the launches are what an iteration is made of)
      const bool fold = one_launch_solve;
      if (it == 0 || !fold) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_cameras, g_cam, blk256, 0, s, p, g, 0));
      if (fold) {
        if (b->nline > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_line<true>, g_linew, blk256, 0, s, p, g, pol));
      } else {
        if (g.nobs > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_linearise, g_obs128, blk128, 0, s, p, g, pol));
        if (b->nline > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_line<false>, g_linew, blk256, 0, s, p, g, pol));
        if (it == 0 && g.nobs > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_rescale, g_obs256, blk256, 0, s, p, g));
        if (g.nobs > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_F, g_obs128, blk128, 0, s, p, g));
      }
      if (b->ncam > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_cam, g_camwg, blk256, 0, s, p, g));
      if (g.npairs > 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_big_pairs, g_pair, blk256, 0, s, p, g));
      LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_big_prepare, g_win, blk256, 0, s, p, g, pol, one_launch_solve ? 1 : 0));
      if (pol.max_num_iterations <= 0) break;
      if (one_launch_solve) {
        if (max_n <= 240) LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_big_solve<15>, g_win, dim3(64 * kBsvWaves), 0, s, p, g, pol));
        else LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_big_solve<kBsvSlots>, g_win, dim3(64 * kBsvWaves), 0, s, p, g, pol));
      }
      for (int wi = 0; wi < B && !one_launch_solve; ++wi) {
        const int n = b->h_wins[wi].n;
        if (n <= 0) continue;
        PoPtrs pp;
        std::memset(&pp, 0, sizeof(pp));
        pp.st = b->d_state.p + wi; pp.flags = b->d_big_flags.p + 2 * wi; pp.n = n; pp.ld = big_ld(n);
        pp.H = b->d_big_sys.p + b->h_big_sys_off[wi]; pp.y = pp.H + (long long)n * pp.ld + 3LL * n;
        double* linv = b->d_big_linv.p + b->h_big_linv_off[wi];
        const int nblk = (n + kNB - 1) / kNB;
        for (int bk = 0; bk < nblk; ++bk) {
          const int k0 = bk * kNB, rem = n - (k0 + kNB), tb = rem > 0 ? (rem + kNB - 1) / kNB : 0;
          LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_po_potrf_diag<double>, dim3(1), blk256, 0, s, pp, pp.H, linv + (size_t)bk * kNB * kNB, k0));
          if (tb > 0) {
            LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_po_panel_update<double>, dim3((unsigned)tb), blk256, 0, s, pp, pp.H, (const double*)(linv + (size_t)bk * kNB * kNB), k0, 0));
            LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_po_panel_update<double>, dim3((unsigned)(tb * (tb + 1) / 2)), blk256, 0, s, pp, pp.H, (const double*)(linv + (size_t)bk * kNB * kNB), k0, 1));
          }
        }
        LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_po_trisolve<double>, dim3(1), dim3(1024), 0, s, pp, (const double*)pp.H, (const double*)linv));
      }
      if (!one_launch_solve) LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_big_finish, g_win, blk64, 0, s, p, g));
      if (it == 0 && g.nobs > 0) LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_big_rescale_cameras, g_obs256, blk256, 0, s, p, g));
      if (!one_launch_solve) LAUNCH(FAM_BACKSUB, hipLaunchKernelGGL(k_big_cameras, g_cam, blk256, 0, s, p, g, 1));
      if (fold) {
        if (b->nline > 0) LAUNCH(FAM_BACKSUB, hipLaunchKernelGGL(k_big_backsub_line<true>, g_linew, blk256, 0, s, p, g, pol));
      } else {
        if (b->nline > 0) LAUNCH(FAM_BACKSUB, hipLaunchKernelGGL(k_big_backsub_line<false>, g_linew, blk256, 0, s, p, g, pol));
        if (g.nobs > 0) LAUNCH(FAM_BACKSUB, hipLaunchKernelGGL(k_big_cost, g_obs128, blk128, 0, s, p, g, pol));
      }
      LAUNCH(FAM_UPDATE, hipLaunchKernelGGL(k_big_reduce, g_win, blk256, 0, s, p, g, pol, fold ? 1 : 0));
    }
    HIP_TRY(hipGetLastError());
    return SLSLAM_OK;
  }
  // Ceres' initial evaluation (cost, gradient, column norms at x0 -> Jacobi scale, trace record 0) rides on the first
  // LM iteration's sweeps (LMState.fresh); only a solve that may not iterate at all needs it as a pass of its own
  if (pol.max_num_iterations <= 0 || pol.store_f) {     // (the streaming variant keeps F blocks in the first sweep's coordinates)
    if (b->nchunk > 0) LAUNCH(FAM_INIT, hipLaunchKernelGGL(k_linearise_schur<true>, g_chunk, blk64, b->lds_lin, s, p, pol));
    LAUNCH(FAM_UPDATE, hipLaunchKernelGGL(k_lm_init, g_win, blk64, 0, s, p, pol));
  }
  for (int it = 0; it < pol.max_num_iterations; ++it) {
    // long solves (the reference's max_num_iter = 1000 study): every 16 iterations ask the device
    // whether any window is still iterating and stop enqueueing when none is
    if (!capturing && it > 0 && (it % 16) == 0) {
      unsigned int active = 0;
      HIP_TRY(hipMemcpyAsync(&active, b->d_active.p, sizeof(active), hipMemcpyDeviceToHost, s));
      HIP_TRY(hipStreamSynchronize(s));
      if (active == 0) break;
    }
    if (!capturing && ((it + 1) % 16) == 0) HIP_TRY(hipMemsetAsync(b->d_active.p, 0, sizeof(unsigned int), s));
    if (b->nchunk > 0) {
      if (b->elim_grouped && it == 0) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_eliminate_grouped<true>, g_chunk, blk64, b->lds_elim, s, p, pol));
      else if (b->elim_grouped && b->elim_mixed) LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_eliminate_grouped<false, false, true>), g_chunk, blk64, b->lds_elim, s, p, pol));
      else if (b->elim_grouped && pol.keep_jacobian) LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_eliminate_grouped<false, true>), g_chunk, blk64, b->lds_elim, s, p, pol));
      else if (b->elim_grouped) LAUNCH(FAM_LIN, hipLaunchKernelGGL(k_eliminate_grouped<false>, g_chunk, blk64, b->lds_elim, s, p, pol));
      else if (b->elim_mode == 1 && pol.debug_flags) {          // timing experiments (SLSLAM_DEBUG_ABLATE)
        if (b->elim_waves == 2) LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_eliminate_mfma<2, true>), g_chunk, dim3(128), b->lds_elim, s, p, pol));
        else LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_eliminate_mfma<1, true>), g_chunk, blk64, b->lds_elim, s, p, pol));
      } else if (b->elim_mode == 1 && b->elim_waves == 2) LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_eliminate_mfma<2, false>), g_chunk, dim3(128), b->lds_elim, s, p, pol));
      else if (b->elim_mode == 1) LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_eliminate_mfma<1, false>), g_chunk, blk64, b->lds_elim, s, p, pol));
      // the first sweep of a solve doubles as Ceres' initial evaluation (every window fresh) unless the separate pass above ran
      else if (it == 0 && !(pol.max_num_iterations <= 0 || pol.store_f)) LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_linearise_schur<false, 1>), g_chunk, blk64, b->lds_lin, s, p, pol));
      else LAUNCH(FAM_LIN, hipLaunchKernelGGL((k_linearise_schur<false, 0>), g_chunk, blk64, b->lds_lin, s, p, pol));
    }
    if (b->slab_sum_stride)
      LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_slab_reduce, dim3((unsigned)B, (unsigned)((b->slab_sum_stride * kSlabReduceSub + 255) / 256)), blk256, 0, s, p));   // windows on grid.x (no 65535 limit)
    if (B <= b->num_cus) LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_reduced_solve<1>, g_win, blk256, b->lds_solve, s, p, pol));
    else LAUNCH(FAM_SOLVE, hipLaunchKernelGGL(k_reduced_solve<4>, g_win, blk256, b->lds_solve, s, p, pol));
    if (b->nchunk > 0) {
      if (pol.store_f) LAUNCH(FAM_BACKSUB, hipLaunchKernelGGL(k_backsub_stream, g_chunk, blk64, b->lds_bs_stream, s, p, pol));
      else LAUNCH(FAM_BACKSUB, hipLaunchKernelGGL(k_backsub, g_chunk, dim3(64 * b->elim_waves), b->lds_bs, s, p, pol));
    }
    if (pol.store_f) {   // the streaming variant has no observation in registers: separate sin/cos and cost sweeps
      if (b->nline > 0) LAUNCH(FAM_TRIG, hipLaunchKernelGGL(k_line_trig, g_line, blk256, 0, s, p, 1));
      if (b->nchunk > 0) LAUNCH(FAM_COST, hipLaunchKernelGGL(k_candidate_cost, g_chunk, blk64, b->lds_cost, s, p, pol));
    }
    if (b->slab_sum_stride) LAUNCH(FAM_UPDATE, hipLaunchKernelGGL(k_lm_update_wave, g_win, blk64, 0, s, p, pol));     // many chunks per window
    else LAUNCH(FAM_UPDATE, hipLaunchKernelGGL(k_lm_update, g_upd, blk64, 0, s, p, pol));
  }
#undef LAUNCH
  HIP_TRY(hipGetLastError());
  return SLSLAM_OK;
}

}  // namespace

extern "C" int slslam_lba_batch_solve(slslam_lba_batch* b, void* stream) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  b->downloaded = false;
  if (b->part[0]) return on_both_parts(b, s, [](slslam_lba_batch* pb, void* st) { return slslam_lba_batch_solve(pb, st); });
  if (b->profiling) {                                   // times accumulate until set_profiling()
    if (b->ev_next > 16384) b->harvest_events();        // long profiled runs: bounded pool (costs one synchronisation)
    return enqueue_solve(b, s, true);
  }
  if (!b->opt.use_graph || b->opt.max_num_iterations > 16 || (b->big_mode && !big_one_launch(b))) return enqueue_solve(b, s, false);
  if (!b->graph_exec) {
    // capture the whole solve (1 + 4 * max_iter launches) once; replay costs one host call
    HIP_TRY(hipStreamCreateWithFlags(&b->capture_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamBeginCapture(b->capture_stream, hipStreamCaptureModeThreadLocal));
    const int rc = enqueue_solve(b, b->capture_stream, false, true);
    hipGraph_t g = nullptr;
    const hipError_t e = hipStreamEndCapture(b->capture_stream, &g);
    if (rc != SLSLAM_OK || e != hipSuccess) {           // no half-built capture state survives a failed attempt
      if (g) (void)hipGraphDestroy(g);
      (void)hipStreamDestroy(b->capture_stream); b->capture_stream = nullptr;
      if (rc != SLSLAM_OK) return rc;
      HIP_TRY(e);
    }
    const hipError_t ei = hipGraphInstantiate(&b->graph_exec, g, nullptr, nullptr, 0);
    (void)hipGraphDestroy(g);
    // (the stream only served the capture: kept alive it would occupy one of the device's few hardware queues, which the streams of a process
    // share - a resident batch next to a stream of windows made uploads wait behind solves, tools/streamed_dbg.py)
    (void)hipStreamDestroy(b->capture_stream); b->capture_stream = nullptr;
    HIP_TRY(ei);
  }
  HIP_TRY(hipGraphLaunch(b->graph_exec, s));
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_reset(slslam_lba_batch* b, void* stream) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  if (b->part[0]) { b->downloaded = false; return on_both_parts(b, s, [](slslam_lba_batch* pb, void* st) { return slslam_lba_batch_reset(pb, st); }); }
  const long long total = 6LL * b->ncam + (long long)b->nline + b->ptrs.nwin;      // one thread per line / camera parameter / window state
  if (total > 0)
    hipLaunchKernelGGL(k_reset, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, b->ptrs, b->pol);
  HIP_TRY(hipGetLastError());
  b->downloaded = false;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_iterations(slslam_lba_batch* b, void* stream, long long* iterations, int clear) {
  if (!b || !iterations) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  if (b->part[0]) {
    long long a = 0, c = 0;
    int rc = slslam_lba_batch_iterations(b->part[0], stream, &a, clear);
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_iterations(b->part[1], stream, &c, clear);
    *iterations = a + c;
    return rc;
  }
  unsigned long long v = 0;
  HIP_TRY(hipMemcpyAsync(&v, b->d_iter_counter.p, sizeof(v), hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  if (clear) HIP_TRY(hipMemsetAsync(b->d_iter_counter.p, 0, sizeof(v), s));
  *iterations = (long long)v;
  return SLSLAM_OK;
}

namespace {
// mixed batch: window blockIdx.x of a part's export goes to its place in the caller's layout; offs = [source offset | destination offset | length] per window
__global__ __launch_bounds__(256) void k_scatter_windows(const double* src, double* dst, const long long* offs) {
  const long long so = offs[3 * (long long)blockIdx.x], dofs = offs[3 * (long long)blockIdx.x + 1], n = offs[3 * (long long)blockIdx.x + 2];
  for (long long q = threadIdx.x; q < n; q += 256) dst[dofs + q] = src[so + q];
}
}  // namespace

extern "C" int slslam_lba_batch_export_device(slslam_lba_batch* b, double* device_out, void* stream) {
  if (!b || !device_out) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  if (b->part[0]) {                                     // each part exports its windows, which are then put where the caller's order has them
    for (int h = 0; h < 2; ++h) {
      const slslam_lba_batch* pb = b->part[h];
      int rc = slslam_lba_batch_export_device(b->part[h], b->d_part_out[h].p, stream);
      if (rc != SLSLAM_OK) return rc;
      // one launch per part puts its windows where the caller's order has them (not one device-to-device copy per window)
      const unsigned nw = (unsigned)pb->h_param_off.size();
      if (nw > 0) hipLaunchKernelGGL(k_scatter_windows, dim3(nw), dim3(256), 0, s, (const double*)b->d_part_out[h].p, device_out, (const long long*)b->d_part_off[h].p);
      HIP_TRY(hipGetLastError());
    }
    return SLSLAM_OK;
  }
  const int total = b->ncam + b->nline;
  if (total > 0)
    hipLaunchKernelGGL(k_export, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, b->ptrs,
                       b->d_param_off.p, b->d_cam_win.p, b->d_line_orig.p, device_out);
  HIP_TRY(hipGetLastError());
  return SLSLAM_OK;
}

namespace {
// Results of a device-built refill whose `parameters` arrays are page-locked go straight where the caller wants them (reference
// src/slam.cpp:957-972 reads them there): thread <-> parameter block, the window's destination from the RawWin table.  A window that ended in
// NUMERICAL_FAILURE is left untouched, as Ceres leaves the user's parameters.
__global__ __launch_bounds__(256) void k_export_inplace(BatchPtrs p, const RawWin* raw, const int* line_pos, int nwin) {
  // thread <-> one double of a window's parameter vector, in the CALLER'S order: consecutive lanes write consecutive addresses
  // (the destination is host memory across the link: 8-byte stores scattered by the sorted line order ran at a sixth of its rate).
  // FEW workgroups walk the windows: the stores are posted and the link sets the pace (~1.3 ms per 67 MB) whatever the grid is, but a
  // wave that has stores on their way keeps its slot - a grid of thousands of short workgroups held slots all over the chip for that time,
  // and the solves of the stream's other batches (two 256-register waves per SIMD, no room beside them) waited for every one of them.
  for (int w = blockIdx.x; w < nwin; w += gridDim.x) {
    const WinDesc wd = p.wins[w];
    if (p.state[w].status == SLSLAM_NUMERICAL_FAILURE) continue;
    const int cur = p.state[w].cur, np = 6 * wd.C + 4 * wd.L;
    double* dst = raw[w].params;
    for (int q = threadIdx.x; q < np; q += 256) {
      double v;
      if (q < 6 * wd.C) { const int c = q / 6; v = p.cam_x[((long long)(wd.cam_off + c) * 2 + cur) * kCamRec + (q - 6 * c)]; }
      else { const int ql = q - 6 * wd.C, l = ql >> 2; v = p.line_x[line_rec(p, (long long)wd.line_off + line_pos[wd.line_off + l], cur) + (ql & 3)]; }
      dst[q] = v;
    }
  }
}

int download_async_impl(slslam_lba_batch* b, void* stream, bool allow_inplace) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  b->downloaded = false;
  if (b->part[0]) {
    int rc0 = download_async_impl(b->part[0], stream, false);
    if (rc0 == SLSLAM_OK) rc0 = download_async_impl(b->part[1], stream, false);
    return rc0;
  }
  const bool inplace = allow_inplace && b->device_built && b->inplace_export;
  b->results_inplace = inplace;
  if (inplace) {
    static const int export_wgs = std::getenv("SLSLAM_EXPORT_WORKGROUPS") ? std::max(1, std::atoi(std::getenv("SLSLAM_EXPORT_WORKGROUPS"))) : 64;
    if (b->total_params > 0 && !b->wins.empty())
      hipLaunchKernelGGL(k_export_inplace, dim3((unsigned)std::min<size_t>(b->wins.size(), (size_t)export_wgs)), dim3(256), 0, s, b->ptrs, (const RawWin*)b->d_rawwin.p,
                         (const int*)b->d_line_pos.p, (int)b->wins.size());
    HIP_TRY(hipGetLastError());
  } else {
    int rc = slslam_lba_batch_export_device(b, b->d_params_out.p, stream);
    if (rc) return rc;
    if (b->total_params > 0)
      HIP_TRY(hipMemcpyAsync(b->h_params.data(), b->d_params_out.p, (size_t)b->total_params * sizeof(double), hipMemcpyDeviceToHost, s));
  }
  if (!b->h_state.empty()) {
    HIP_TRY(hipMemcpyAsync(b->h_state.data(), b->d_state.p, b->h_state.size() * sizeof(LMState), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(b->h_trace.data(), b->d_trace.p, b->h_trace.size() * sizeof(IterRec), hipMemcpyDeviceToHost, s));
  }
  if (b->device_built && !b->wins.empty()) {
    // what the device made of the windows: descriptors (free cameras, kept residual blocks, chunk cut) and the per-window build status
    HIP_TRY(hipMemcpyAsync(b->h_wins_dl, b->d_wins.p, b->wins.size() * sizeof(WinDesc), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(b->h_buildwin, b->d_buildwin.p, b->wins.size() * sizeof(BuildWin), hipMemcpyDeviceToHost, s));
    HIP_TRY(hipMemcpyAsync(b->h_totals, b->d_totals.p, 8 * sizeof(int), hipMemcpyDeviceToHost, s));
  }
  if (!b->ev_results) HIP_TRY(hipEventCreateWithFlags(&b->ev_results, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(b->ev_results, s));
  b->results_pending = true;
  return SLSLAM_OK;
}

// after the copies of a download have arrived: the host mirrors of a device-built refill
void adopt_device_build(slslam_lba_batch* b) {
  if (!b->device_built) return;
  const size_t B = b->wins.size();
  for (size_t i = 0; i < B; ++i) {
    const BuildWin& bw = b->h_buildwin[i];
    b->h_wins[i] = b->h_wins_dl[i];
    b->build_status[i] = bw.status == 0 ? SLSLAM_OK : (bw.status & kBuildInvalid) ? SLSLAM_ERR_INVALID_ARGUMENT : SLSLAM_ERR_UNSUPPORTED;
    b->h_win_graded[i] = bw.graded;
    PackedWindow& P = b->wins[i];
    P.Cf = b->h_wins[i].Cf; P.nfree_params = b->h_wins[i].nfree_params; P.nkept = b->h_wins[i].nkept;
  }
  b->used_tiles = b->h_totals[0]; b->used_items = b->h_totals[1];
}
}  // namespace

extern "C" int slslam_lba_batch_download_async(slslam_lba_batch* b, void* stream) { return download_async_impl(b, stream, false); }

extern "C" int slslam_lba_batch_wait(slslam_lba_batch* b) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(b->device));
  if (b->part[0]) {
    int rc0 = slslam_lba_batch_wait(b->part[0]);
    if (rc0 == SLSLAM_OK) rc0 = slslam_lba_batch_wait(b->part[1]);
    b->downloaded = rc0 == SLSLAM_OK;
    return rc0;
  }
  if (!b->results_pending) return b->downloaded ? SLSLAM_OK : SLSLAM_ERR_STATE;
  // (a finished event is seen by a query at once; hipEventSynchronize on it was measured at ~2 ms per call in the streamed leg)
  if (hipEventQuery(b->ev_results) != hipSuccess) HIP_TRY(hipEventSynchronize(b->ev_results));
  b->results_pending = false;
  if (b->profiling) b->harvest_events();
  adopt_device_build(b);
  b->downloaded = true;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_download(slslam_lba_batch* b, void* stream) {
  const int rc = slslam_lba_batch_download_async(b, stream);
  return rc != SLSLAM_OK ? rc : slslam_lba_batch_wait(b);
}


namespace {

int ensure_build_host_buffers(slslam_lba_batch* b, int B) {
  if (!b->h_rawwin) HIP_TRY(hipHostMalloc((void**)&b->h_rawwin, sizeof(RawWin) * (size_t)std::max(1, B), hipHostMallocDefault));
  if (!b->h_buildwin) { HIP_TRY(hipHostMalloc((void**)&b->h_buildwin, sizeof(BuildWin) * (size_t)std::max(1, B), hipHostMallocDefault)); std::memset(b->h_buildwin, 0, sizeof(BuildWin) * (size_t)std::max(1, B)); }
  if (!b->h_wins_dl) HIP_TRY(hipHostMalloc((void**)&b->h_wins_dl, sizeof(WinDesc) * (size_t)std::max(1, B), hipHostMallocDefault));
  if (!b->h_totals) { HIP_TRY(hipHostMalloc((void**)&b->h_totals, sizeof(int) * 8, hipHostMallocDefault)); std::memset(b->h_totals, 0, sizeof(int) * 8); }
  return SLSLAM_OK;
}

// Copies `n` doubles (n even or odd) with the NaN / Inf test of lba_pack.cpp::all_finite; non-temporal stores when the destination allows.
unsigned long long copy_checked(double* dst, const double* src, size_t n) {
  unsigned long long bad = 0;
  size_t q = 0;
#if defined(__SSE2__)
  if (!(reinterpret_cast<uintptr_t>(dst) & 15u)) {
    const __m128i expo = _mm_set1_epi64x((long long)0x7ff0000000000000ull), one = _mm_set1_epi64x((long long)0x0010000000000000ull);
    __m128i acc = _mm_setzero_si128();
    for (; q + 2 <= n; q += 2) {
      const __m128d v = _mm_loadu_pd(src + q);
      _mm_stream_pd(dst + q, v);
      acc = _mm_or_si128(acc, _mm_add_epi64(_mm_and_si128(_mm_castpd_si128(v), expo), one));
    }
    _mm_sfence();
    unsigned long long lanes[2];
    _mm_storeu_si128(reinterpret_cast<__m128i*>(lanes), acc);
    bad |= (lanes[0] | lanes[1]) & 0x8000000000000000ull;
  }
#endif
  for (; q < n; ++q) {
    unsigned long long x;
    std::memcpy(&x, src + q, 8);
    dst[q] = src[q];
    bad |= ((x & 0x7ff0000000000000ull) + 0x0010000000000000ull) & 0x8000000000000000ull;
  }
  return bad;
}

// The LBAProblem::build stage of a refill ON THE DEVICE (lba_device_build.h).  SLSLAM_OK: taken; SLSLAM_ERR_UNSUPPORTED: not this path's
// business - nothing was touched, the caller goes on with the host packer; anything else: the refill failed.
// Two streams: the ingest (the host link's business: few workgroups for ~20 ms per 1024 x 2000-line batch) is enqueued on `s_in`, everything
// after it on `s` behind an event.  One stream for both (the plain refill entry point) is fine; a STREAM of windows hands ONE ingest stream
// and ONE solve stream to all its slots, so that batch k + 1 crosses the link while batch k is solved and neither shares its resource
// (batches that ingest and solve at the same time on streams of their own fall into step: three ingests share the link, then three
// solves share the chip - measured 37 ms per batch against 20).
// packed (optional): packed[i] != nullptr replaces window i's three index arrays by the narrowed form (slslam_pack_indices: one 32-bit word per
// observation) - 68 instead of 80 bytes per observation over the host link.
// the device block the copy engine fills (refill_device); the previous refill's kernels have read the old one: ev_stage_free / the results were waited for
int ensure_stage_block(slslam_lba_batch* b, size_t bytes) {
  if (bytes <= b->d_stage_bytes) return SLSLAM_OK;
  if (b->d_stage_in) { (void)hipFree(b->d_stage_in); b->d_stage_in = nullptr; b->d_stage_bytes = 0; }
  const size_t want = bytes + bytes / 8 + 4096;
  HIP_TRY(hipMalloc((void**)&b->d_stage_in, want));
  b->d_stage_bytes = want;
  return SLSLAM_OK;
}

int refill_device(slslam_lba_batch* b, const slslam_lba_window* windows, int B, hipStream_t s, hipStream_t s_in, const unsigned int* const* packed = nullptr) {
  if (b->opt.device_build < 0 || b->d_rawwin.n < (size_t)std::max(1, B) || !b->d_ob_raw.p) return SLSLAM_ERR_UNSUPPORTED;
  if (std::getenv("SLSLAM_CHUNK_WEIGHTS")) return SLSLAM_ERR_UNSUPPORTED;          // (an experiment knob of the host-side cut)
  // (k_build_layout dispatches graded chunks in at most kLayoutMaxRank classes: a batch cut for more rounds of the wave slots - several
  // thousand 2000-line windows - keeps the host packer)
  if ((b->opt.chunks_per_window < 0 ? (-b->opt.chunks_per_window) / 1000 : (b->opt.reproducible ? 3 : b->auto_rounds)) > (int)kLayoutMaxRank) return SLSLAM_ERR_UNSUPPORTED;
  static const bool timing = std::getenv("SLSLAM_REFILL_TIMING") != nullptr;
  const auto tt0 = std::chrono::steady_clock::now();
  // ---- what the host knows without reading an array: the windows' sizes, hence their places in the batch
  std::vector<int> cam_off((size_t)B + 1, 0), line_off((size_t)B + 1, 0);
  std::vector<long long> obs_off((size_t)B + 1, 0), par_off((size_t)B + 1, 0);
  int maxL = 0, maxM = 0, maxC = 1;
  for (int i = 0; i < B; ++i) {
    const slslam_lba_window& w = windows[i];
    if (w.num_cameras < 0 || w.num_lines < 0 || w.num_observations < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
    const bool pk = packed && packed[i];
    if (w.num_observations > 0 && ((!pk && (!w.camera_index || !w.line_index || !w.fixed_index)) || !w.observations)) return SLSLAM_ERR_INVALID_ARGUMENT;
    if ((w.num_cameras > 0 || w.num_lines > 0) && !w.parameters) return SLSLAM_ERR_INVALID_ARGUMENT;
    if (w.num_cameras > kMaxCams || w.num_lines > 0xfffe || w.num_observations >= (1 << 24)) return SLSLAM_ERR_UNSUPPORTED;
    maxL = std::max(maxL, w.num_lines); maxM = std::max(maxM, w.num_observations); maxC = std::max(maxC, w.num_cameras);
    cam_off[(size_t)i + 1] = cam_off[(size_t)i] + w.num_cameras; line_off[(size_t)i + 1] = line_off[(size_t)i] + w.num_lines;
    obs_off[(size_t)i + 1] = obs_off[(size_t)i] + w.num_observations;
    par_off[(size_t)i + 1] = par_off[(size_t)i] + 6LL * w.num_cameras + 4LL * w.num_lines;
  }
  size_t lds_tiles = build_tiles_lds_bytes(maxL, maxL);
  const size_t lds_build = build_lds_bytes(maxL), lds_tiles_staged = build_tiles_lds_bytes(maxL, maxL, maxM, b->elim_grouped ? 1 : 0);
  if (lds_build > 158 * 1024 || lds_tiles > 158 * 1024) return SLSLAM_ERR_UNSUPPORTED;
  const int tiles_stage_m = lds_tiles_staged <= 158 * 1024 ? maxM : -1;      // (k_build_tiles: the tile loop's inputs in LDS when they fit)
  if (tiles_stage_m >= 0) lds_tiles = lds_tiles_staged;
  const long long ncam = cam_off[(size_t)B], nline = line_off[(size_t)B], nobs = obs_off[(size_t)B], nparams = par_off[(size_t)B];
  const long long ob_stride = (long long)(b->d_ob.n / 8);
  if ((size_t)ncam > b->d_cam_cf.n || (size_t)nline > b->d_line_win.n || nobs > ob_stride || (size_t)nparams > b->d_params_out.n ||
      (size_t)nparams > b->h_params.size() || maxC > b->cap_maxC || nobs > 0x7fffffffLL)
    return SLSLAM_ERR_UNSUPPORTED;                      // (the host path says the same, before touching anything)
  int rc = ensure_build_host_buffers(b, B);
  if (rc != SLSLAM_OK) return rc;
  // ---- where the device reads the windows: where they are when all their arrays are page-locked (slslam_pinned_alloc /
  // slslam_pinned_register), else a pinned staging copy made here by the host threads (indices narrowed on the way)
  bool all_pinned = true, params_pinned = true;
  {
    const std::vector<PinnedRegistry::Range> rs = PinnedRegistry::get().snapshot();
    for (int i = 0; i < B && (all_pinned || params_pinned); ++i) {
      const slslam_lba_window& w = windows[i];
      const size_t M = (size_t)w.num_observations, np = (size_t)6 * w.num_cameras + (size_t)4 * w.num_lines;
      if (np && !PinnedRegistry::contains(rs, w.parameters, 8 * np)) { params_pinned = false; all_pinned = false; }
      const bool pk = packed && packed[i];
      if (M && all_pinned && !((pk ? PinnedRegistry::contains(rs, packed[i], 4 * M)
                                   : (PinnedRegistry::contains(rs, w.camera_index, 4 * M) && PinnedRegistry::contains(rs, w.line_index, 4 * M) && PinnedRegistry::contains(rs, w.fixed_index, 8 * M))) &&
                               PinnedRegistry::contains(rs, w.observations, 64 * M)))
        all_pinned = false;
    }
  }
  // the host image / the previous refill's sources may still be read by the device; results under way arrive first (and stay readable)
  if (b->ev_stage_free) HIP_TRY(hipEventSynchronize(b->ev_stage_free));
  if (b->results_pending) { HIP_TRY(hipEventSynchronize(b->ev_results)); b->results_pending = false; adopt_device_build(b); b->downloaded = true; if (b->profiling) b->harvest_events(); }
  const auto tt1 = std::chrono::steady_clock::now();
  RawWin* rw = b->h_rawwin;
  b->host_src.assign((size_t)B, RawWin());
  // copies the copy engine is to make before the ingest (host address, device offset in d_stage_in, bytes); empty: the ingest kernel reads
  // the callers' arrays itself (zero copy)
  struct Run { uintptr_t lo, hi; size_t dev_off; };
  std::vector<Run> runs;
  size_t dev_need = 0;
  int idx_run = -1;                   // the run that holds the indices the host threads narrow (copied after the others)
  bool early_dma = false;             // the other runs are on their way already
  if (all_pinned) {
    for (int i = 0; i < B; ++i) {
      const slslam_lba_window& w = windows[i];
      RawWin& r = b->host_src[(size_t)i];
      const bool pk = packed && packed[i];
      r.cam = pk ? nullptr : w.camera_index; r.line = pk ? nullptr : w.line_index; r.fixed = pk ? nullptr : w.fixed_index; r.packed = pk ? packed[i] : nullptr;
      r.obs = w.observations; r.params_in = w.parameters; r.params = w.parameters;
      rw[i] = r;
    }
    // Arrays that lie next to each other in host memory (a caller that carves its windows out of an arena) go up in a few large copies of
    // the copy engine, which does not travel through the shader's L2: a kernel that keeps the link full has ~0.5 MB of host reads
    // outstanding in the L2 channels, and every latency-bound kernel beside it waits behind them (the window build 2.7 -> 11.6 ms, k_build_tiles
    // 0.36 -> 6.4 ms measured).  Scattered arrays: zero-copy kernel reads.
    static const bool no_dma = std::getenv("SLSLAM_INGEST_ZERO_COPY") != nullptr;         // (measurement switch)
    // The three int32 index arrays are 16 of the 80 bytes an observation sends over the link, which is what a stream of page-locked windows is
    // bound by (1.085 GB per 1024 x 2000-line batch at 57 GB/s = 19 ms against 15 ms of solve).  The host threads narrow them to one word per
    // observation into a pinned block on the way (4 B; ~0.25 GB of reads per batch: a few milliseconds on two threads, beside the GPU's work) -
    // the observations and the parameters still go up from where they are.  Bad indices are reported now, as the host packer does.
    static const bool no_narrow = std::getenv("SLSLAM_NO_HOST_NARROW") != nullptr;          // (measurement switch)
    bool any_unpacked = false;
    for (int i = 0; i < B && !any_unpacked; ++i) any_unpacked = windows[i].num_observations > 0 && !(packed && packed[i]);
    const bool narrow = !no_dma && !no_narrow && nobs > 0 && any_unpacked;       // (a batch whose caller narrowed every window sends no block of ours)
    if (narrow) {
      const size_t need = 4 * (size_t)nobs + 64;
      if (need > b->raw_stage_bytes) {
        if (b->h_raw_stage) { (void)hipHostFree(b->h_raw_stage); b->h_raw_stage = nullptr; b->raw_stage_bytes = 0; }
        const size_t want = need + need / 8 + 4096;
        HIP_TRY(hipHostMalloc((void**)&b->h_raw_stage, want, hipHostMallocDefault));
        b->raw_stage_bytes = want;
      }
    }
    // the ranges the copy engine is to move: the callers' arrays and - a run of its own, never merged with a neighbour - the narrowed block
    struct Rg { uintptr_t lo, hi; bool idx; };
    std::vector<Rg> rg;
    rg.reserve((size_t)5 * B + 1);
    size_t payload = 0;
    if (narrow) { rg.push_back({ (uintptr_t)b->h_raw_stage, (uintptr_t)b->h_raw_stage + 4 * (size_t)nobs, true }); payload += 4 * (size_t)nobs; }
    for (int i = 0; i < B && !no_dma; ++i) {
      const slslam_lba_window& w = windows[i];
      const size_t M = (size_t)w.num_observations, np = (size_t)6 * w.num_cameras + (size_t)4 * w.num_lines;
      const bool pk = packed && packed[i];
      if (M && pk) rg.push_back({ (uintptr_t)packed[i], (uintptr_t)packed[i] + 4 * M, false });
      if (M && !pk && !narrow) { rg.push_back({ (uintptr_t)w.camera_index, (uintptr_t)w.camera_index + 4 * M, false }); rg.push_back({ (uintptr_t)w.line_index, (uintptr_t)w.line_index + 4 * M, false });
                                 rg.push_back({ (uintptr_t)w.fixed_index, (uintptr_t)w.fixed_index + 8 * M, false }); }
      if (M) rg.push_back({ (uintptr_t)w.observations, (uintptr_t)w.observations + 64 * M, false });
      if (np) rg.push_back({ (uintptr_t)w.parameters, (uintptr_t)w.parameters + 8 * np, false });
      payload += (pk ? 68 : narrow ? 64 : 80) * M + 8 * np;
    }
    std::sort(rg.begin(), rg.end(), [](const Rg& x, const Rg& y) { return x.lo < y.lo; });
    bool barrier = true;                                  // the next range starts a run whatever lies before it
    for (const Rg& g : rg) {
      if (g.idx) { idx_run = (int)runs.size(); runs.push_back(Run{ g.lo, g.hi, 0 }); barrier = true; continue; }
      if (!barrier && g.lo <= runs.back().hi + 4096) runs.back().hi = std::max(runs.back().hi, g.hi);
      else runs.push_back(Run{ g.lo, g.hi, 0 });
      barrier = false;
    }
    for (Run& r : runs) { r.dev_off = dev_need + (r.lo & 255); dev_need += ((r.lo & 255) + (r.hi - r.lo) + 255) & ~(size_t)255; }     // (device address = host address modulo 256)
    if (runs.size() > (size_t)std::max(8, B / 16) || dev_need > payload + payload / 8 + (1u << 20)) { runs.clear(); dev_need = 0; idx_run = -1; }     // scattered: zero copy
    // The callers' arrays start up the link NOW, the host threads narrow the indices meanwhile: a submit's chain is
    // max(narrowing, 0.87 GB of copies) + build, not their sum - with three batches in flight the sum (13 + 17 + 4 ms on a slow host thread)
    // left 2 ms of slack against two solve periods, and a stream at the mercy of the host's jitter (0.70 of resident measured on such a box).
    if (!runs.empty()) {
      rc = ensure_stage_block(b, dev_need);
      if (rc != SLSLAM_OK) return rc;
      for (size_t k = 0; k < runs.size(); ++k)
        if ((int)k != idx_run) HIP_TRY(hipMemcpyAsync(b->d_stage_in + runs[k].dev_off, (const void*)runs[k].lo, runs[k].hi - runs[k].lo, hipMemcpyHostToDevice, s_in));
      early_dma = true;
    }
    if (narrow) {
      std::vector<int> st((size_t)B, SLSLAM_OK);
      HostPool* pool = batch_pool(b, (int)std::min<long long>(B, b->opt.host_threads > 0 ? b->opt.host_threads : (B >= 64 ? 8 : 1)));
      auto narrow_one = [&](int i) {
        if (packed && packed[i]) return;
        const slslam_lba_window& w = windows[i];
        const size_t M = (size_t)w.num_observations;
        uint32_t* ix = reinterpret_cast<uint32_t*>(b->h_raw_stage) + obs_off[(size_t)i];
        const int C = w.num_cameras, L = w.num_lines;
        const int* cam = w.camera_index; const int* line = w.line_index; const int* fx = w.fixed_index;
        unsigned oob = 0;
        for (size_t q = 0; q < M; ++q) {
          const int c = cam[q], l = line[q];
          oob |= (unsigned)(c < 0) | (unsigned)(c >= C) | (unsigned)(l < 0) | (unsigned)(l >= L);
          ix[q] = ((uint32_t)l & 0xffffu) | ((uint32_t)c & 0xffu) << 16 | (fx[2 * q] ? 1u << 24 : 0u) | (fx[2 * q + 1] ? 1u << 25 : 0u);
        }
        if (oob) st[(size_t)i] = SLSLAM_ERR_INVALID_ARGUMENT;
        RawWin& r = b->host_src[(size_t)i];
        r.cam = nullptr; r.line = nullptr; r.fixed = nullptr; r.packed = ix;
        rw[i] = r;
      };
      int bad = run_all(pool, B, narrow_one) ? SLSLAM_OK : SLSLAM_ERR_NO_MEMORY;
      for (int v : st) if (bad == SLSLAM_OK && v != SLSLAM_OK) bad = v;
      if (bad != SLSLAM_OK) {
        if (early_dma) (void)hipStreamSynchronize(s_in);            // the copies under way read the callers' arrays: not beyond this call
        return bad;
      }
    }
  } else {
    // staging: per window [observations 64 M | narrowed indices 4 M | parameters], 64-byte aligned pieces
    std::vector<size_t> st_off((size_t)B + 1, 0);
    for (int i = 0; i < B; ++i) {
      const size_t M = (size_t)windows[i].num_observations, np = (size_t)6 * windows[i].num_cameras + (size_t)4 * windows[i].num_lines;
      st_off[(size_t)i + 1] = st_off[(size_t)i] + ((64 * M + 63) & ~(size_t)63) + ((4 * M + 63) & ~(size_t)63) + ((8 * np + 63) & ~(size_t)63);
    }
    if (st_off[(size_t)B] > b->raw_stage_bytes) {
      if (b->h_raw_stage) { (void)hipHostFree(b->h_raw_stage); b->h_raw_stage = nullptr; b->raw_stage_bytes = 0; }
      const size_t want = st_off[(size_t)B] + st_off[(size_t)B] / 8 + 4096;
      HIP_TRY(hipHostMalloc((void**)&b->h_raw_stage, want, hipHostMallocDefault));
      b->raw_stage_bytes = want;
    }
    std::vector<int> st((size_t)B, SLSLAM_OK);
    HostPool* pool = batch_pool(b, (int)std::min<long long>(B, b->opt.host_threads > 0 ? b->opt.host_threads : (B >= 64 ? 8 : 1)));
    auto stage_one = [&](int i) {
      const slslam_lba_window& w = windows[i];
      const size_t M = (size_t)w.num_observations, np = (size_t)6 * w.num_cameras + (size_t)4 * w.num_lines;
      char* base = b->h_raw_stage + st_off[(size_t)i];
      double* ob = reinterpret_cast<double*>(base);
      uint32_t* ix = reinterpret_cast<uint32_t*>(base + ((64 * M + 63) & ~(size_t)63));
      double* pr = reinterpret_cast<double*>(base + ((64 * M + 63) & ~(size_t)63) + ((4 * M + 63) & ~(size_t)63));
      unsigned long long bad = copy_checked(ob, w.observations, 8 * M);
      bad |= copy_checked(pr, w.parameters, np);
      const int C = w.num_cameras, L = w.num_lines;
      unsigned oob = 0;
      if (packed && packed[i]) {
        for (size_t q = 0; q < M; ++q) { const uint32_t v = packed[i][q]; oob |= (unsigned)((int)(v & 0xffffu) >= L) | (unsigned)((int)((v >> 16) & 0xffu) >= C) | (v >> 26); ix[q] = v; }
      } else
      for (size_t q = 0; q < M; ++q) {
        const int c = w.camera_index[q], l = w.line_index[q];
        oob |= (unsigned)(c < 0) | (unsigned)(c >= C) | (unsigned)(l < 0) | (unsigned)(l >= L);
        ix[q] = ((uint32_t)l & 0xffffu) | ((uint32_t)c & 0xffu) << 16 | (w.fixed_index[2 * q] ? 1u << 24 : 0u) | (w.fixed_index[2 * q + 1] ? 1u << 25 : 0u);
      }
      if (bad || oob) st[(size_t)i] = SLSLAM_ERR_INVALID_ARGUMENT;           // reported now, as the host packer does
      RawWin& r = b->host_src[(size_t)i];
      r.cam = nullptr; r.line = nullptr; r.fixed = nullptr; r.packed = ix; r.obs = ob; r.params_in = pr;
      r.params = params_pinned ? w.parameters : nullptr;                     // (written in place only where the GPU can reach the caller's array)
      rw[i] = r;
    };
    if (!run_all(pool, B, stage_one)) return SLSLAM_ERR_NO_MEMORY;
    for (int v : st) if (v != SLSLAM_OK) return v;
    // the whole staging copy goes up in ONE copy of the copy engine
    if (st_off[(size_t)B] > 0) { runs.push_back(Run{ (uintptr_t)b->h_raw_stage, (uintptr_t)b->h_raw_stage + st_off[(size_t)B], 0 }); dev_need = (st_off[(size_t)B] + 255) & ~(size_t)255; }
  }
  if (!runs.empty()) {
    rc = ensure_stage_block(b, dev_need);
    if (rc != SLSLAM_OK) return rc;
    // the device reads every array at its place in the block
    auto to_dev = [&](const void* p) -> const void* {
      if (!p) return nullptr;
      const uintptr_t a = (uintptr_t)p;
      auto it = std::upper_bound(runs.begin(), runs.end(), a, [](uintptr_t v, const Run& r) { return v < r.lo; });
      --it;
      return b->d_stage_in + it->dev_off + (a - it->lo);
    };
    for (int i = 0; i < B; ++i) {
      RawWin& r = rw[i];
      r.cam = (const int*)to_dev(r.cam); r.line = (const int*)to_dev(r.line); r.fixed = (const int*)to_dev(r.fixed); r.packed = (const uint32_t*)to_dev(r.packed);
      r.obs = (const double*)to_dev(r.obs); r.params_in = (const double*)to_dev(r.params_in);
    }
  }
  b->ingest_mode = runs.empty() ? 0 : 1;
  b->ingest_pinned = all_pinned;
  for (int i = 0; i < B; ++i) {
    RawWin& r = rw[i];
    r.param_off = par_off[(size_t)i]; r.C = windows[i].num_cameras; r.L = windows[i].num_lines; r.M = windows[i].num_observations;
    r.cam_off = cam_off[(size_t)i]; r.line_off = line_off[(size_t)i]; r.obs_off = (int)obs_off[(size_t)i]; r.pad = 0;
    RawWin& h = b->host_src[(size_t)i];
    h.param_off = r.param_off; h.C = r.C; h.L = r.L; h.M = r.M; h.cam_off = r.cam_off; h.line_off = r.line_off; h.obs_off = r.obs_off; h.pad = 0;
  }
  const auto tt2 = std::chrono::steady_clock::now();
  // ---- commit: the batch now IS the new windows (what the device makes of them comes back with the results: slslam_lba_batch_wait)
  b->wins_spare.resize((size_t)B);
  for (int i = 0; i < B; ++i) {
    PackedWindow& P = b->wins_spare[(size_t)i];
    P.C = windows[i].num_cameras; P.L = windows[i].num_lines; P.M = windows[i].num_observations; P.Cf = 0; P.nfree_params = 0; P.nkept = 0;
    P.big = false; P.dup_free_obs = false; P.grouping = b->elim_grouped ? 1 : 0;
    P.tiles.clear(); P.lane_map.clear(); P.items.clear(); P.ob.clear(); P.params0.clear(); P.line_order.clear(); P.ob_orig.clear(); P.ob_cam.clear();
  }
  b->wins.swap(b->wins_spare);
  b->h_wins.assign((size_t)B, WinDesc());
  b->h_param_off.assign(par_off.begin(), par_off.end() - 1);
  b->h_ob_orig_off.resize((size_t)B);
  for (int i = 0; i < B; ++i) {
    WinDesc& wd = b->h_wins[(size_t)i];
    std::memset(&wd, 0, sizeof(wd));
    wd.C = rw[i].C; wd.L = rw[i].L; wd.M = rw[i].M; wd.cam_off = rw[i].cam_off; wd.line_off = rw[i].line_off; wd.obs_off = rw[i].obs_off;
    b->h_ob_orig_off[(size_t)i] = rw[i].obs_off;
  }
  b->h_win_graded.assign((size_t)std::max(1, B), 0);
  b->total_params = nparams; b->nobs = nobs;
  b->used_ncam = ncam; b->used_nline = nline; b->used_nobs = nobs; b->used_tiles = (long long)b->d_tiles.n; b->used_items = (long long)(b->d_items.n / 2);
  b->device_built = true; b->inplace_export = params_pinned; b->results_inplace = false;
  b->src_windows.assign(windows, windows + B);
  b->build_status.assign((size_t)B, SLSLAM_OK);
  b->downloaded = false;
  ++b->n_device_builds; if (all_pinned) ++b->n_zero_copy;
  // ---- enqueue
  BuildPtrs P;
  std::memset(&P, 0, sizeof(P));
  P.raw = b->d_rawwin.p; P.bw = b->d_buildwin.p; P.nwin = B; P.grouping = b->elim_grouped ? 1 : 0;
  P.ob_raw = b->d_ob_raw.p; P.raw_idx = b->d_raw_idx.p; P.line_raw = b->d_line_raw.p; P.lflags = b->d_lflags.p; P.fmask = b->d_fmask.p; P.line_pos = b->d_line_pos.p;
  P.obs_in_place = runs.empty() ? 0 : 1;       // the copy engine's block is read where it is (k_permute_obs): no second copy of the observations
  P.mid_keys = b->d_mid_keys.p; P.mid_li = b->d_mid_li.p; P.mid_rows = b->d_mid_rows.p; P.mid_next = b->d_mid_next.p; P.mid_trows = b->d_mid_trows.p; P.mid_tptr = b->d_mid_tptr.p; P.mid = b->d_mid.p;
  P.wins = b->d_wins.p; P.tiles = b->d_tiles.p; P.chunks = b->d_chunks.p; P.items = b->d_items.p; P.lane_map = b->d_lane_map.p; P.line_desc = b->d_line_desc.p;
  P.cam_x0 = b->d_cam_x0.p; P.cam_cf = b->d_cam_cf.p; P.cam_win = b->d_cam_win.p;
  P.line_x0 = b->d_line_x0.p; P.line_ptr = b->d_line_ptr.p; P.line_flags = b->d_line_flags.p; P.line_win = b->d_line_win.p; P.line_orig = b->d_line_orig.p;
  P.ob_cam = b->d_ob_cam.p; P.ob_orig = b->d_ob_orig.p; P.param_off = b->d_param_off.p; P.item_base = b->d_item_base.p; P.totals = b->d_totals.p;
  LayoutArgs a;
  std::memset(&a, 0, sizeof(a));
  a.chunks_per_window = b->opt.chunks_per_window; a.reproducible = b->opt.reproducible; a.auto_rounds = b->auto_rounds; a.auto_cpw = b->auto_cpw;
  a.elim_waves = b->elim_waves; a.elim_mode = b->elim_mode; a.equal_chunks = std::getenv("SLSLAM_EQUAL_CHUNKS") ? 1 : 0;
  a.cap_tiles = (int)std::min<size_t>(b->d_tiles.n, 0x7fffffff); a.cap_items = (int)std::min<size_t>(b->d_items.n / 2, 0x7fffffff); a.cap_chunks = b->nchunk;
  a.cap_maxn = b->cap_maxn; a.slab_sum = b->slab_sum_stride ? 1 : 0; a.slab_sum_image = b->slab_sum_image ? 1 : 0;
  a.cap_slab = (long long)b->d_slab.n; a.cap_sys = (long long)b->d_ysys.n; a.slab_sum_stride = b->slab_sum_stride;
  a.nline = (int)nline; a.nobs = (int)nobs; a.max_free = b->elim_mode == 1 ? (int)kMfmaMaxFree : (int)kMaxFreeCams;
  for (int cf = 0; cf < kMaxFreeCams + 2; ++cf) a.sys_map_off[cf] = (size_t)cf < b->sys_map_off_of_cf.size() ? b->sys_map_off_of_cf[(size_t)cf] : -1;
  {
    static bool attr_set = false;
    if (!attr_set) {
      HIP_TRY(hipFuncSetAttribute((const void*)k_build_lines, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_build_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_build_order, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
      HIP_TRY(hipFuncSetAttribute((const void*)k_build_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024));
      attr_set = true;
    }
  }
  HIP_TRY(hipMemcpyAsync(b->d_rawwin.p, rw, sizeof(RawWin) * (size_t)B, hipMemcpyHostToDevice, s_in));
  HIP_TRY(hipMemsetAsync(b->d_buildwin.p, 0, sizeof(BuildWin) * (size_t)B, s_in));
  static const int ingest_wgs = std::getenv("SLSLAM_INGEST_WORKGROUPS") ? std::max(1, std::atoi(std::getenv("SLSLAM_INGEST_WORKGROUPS"))) : 32;
  for (size_t k = 0; k < runs.size(); ++k)
    if (!early_dma || (int)k == idx_run) HIP_TRY(hipMemcpyAsync(b->d_stage_in + runs[k].dev_off, (const void*)runs[k].lo, runs[k].hi - runs[k].lo, hipMemcpyHostToDevice, s_in));
  // zero copy: the ingest kernel IS the transfer (32 workgroups keep the link full: tools/micro/zero_copy_bench.hip; more only queue in the L2)
  if (B > 0 && runs.empty()) hipLaunchKernelGGL(k_ingest, dim3((unsigned)std::min(B, ingest_wgs)), dim3(256), 0, s_in, P);
  if (!b->ev_stage_free) HIP_TRY(hipEventCreateWithFlags(&b->ev_stage_free, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(b->ev_stage_free, s_in));
  if (s_in != s) HIP_TRY(hipStreamWaitEvent(s, b->ev_stage_free, 0));
  // copy engine: the ingest reads the block in HBM (a fraction of a millisecond), on the solve stream
  if (B > 0 && !runs.empty()) hipLaunchKernelGGL(k_ingest, dim3((unsigned)std::min(B, 1024)), dim3(256), 0, s, P);
  // the records beyond the refill's lines / cameras belong to no window
  if ((size_t)nline < b->d_line_win.n) HIP_TRY(hipMemsetAsync(b->d_line_win.p + nline, 0xFF, (b->d_line_win.n - (size_t)nline) * sizeof(int), s));
  if ((size_t)ncam < b->d_cam_win.n) {
    HIP_TRY(hipMemsetAsync(b->d_cam_win.p + ncam, 0xFF, (b->d_cam_win.n - (size_t)ncam) * sizeof(int), s));
    HIP_TRY(hipMemsetAsync(b->d_cam_cf.p + ncam, 0xFF, (b->d_cam_cf.n - (size_t)ncam) * sizeof(int), s));
  }
  if (B > 0) {
    hipLaunchKernelGGL(k_build_lines, dim3((unsigned)B), dim3(256), build_lines_lds_bytes(maxL), s, P);
    hipLaunchKernelGGL(k_build_rows, dim3((unsigned)B), dim3(128), build_rows_lds_bytes(maxL), s, P);
    hipLaunchKernelGGL(k_build_order, dim3((unsigned)B), dim3(256), lds_build, s, P);
    hipLaunchKernelGGL(k_build_layout, dim3(1), dim3(256), 0, s, P, a);
    hipLaunchKernelGGL(k_build_tiles, dim3((unsigned)B), dim3(256), lds_tiles, s, P, (const int*)b->d_cam_cf.p, tiles_stage_m);
  }
  HIP_TRY(hipGetLastError());
  // what a fresh batch finds zeroed
  HIP_TRY(hipMemsetAsync(b->d_cam_scale.p, 0, b->d_cam_scale.n * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(b->d_line_scale.p, 0, b->d_line_scale.n * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(b->d_cam_tab.p, 0, b->d_cam_tab.n * sizeof(double), s));
  if (b->slab_sum_image) HIP_TRY(hipMemsetAsync(b->d_slab_sum.p, 0, b->d_slab_sum.n * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(b->d_active.p, 0, sizeof(unsigned int), s));
  if (nobs > 0 && maxM > 0)
    hipLaunchKernelGGL(k_permute_obs, dim3((unsigned)((maxM + 255) / 256), (unsigned)B), dim3(256), 0, s, b->ptrs, (const double*)b->d_ob_raw.p, (const int*)b->d_ob_orig.p, b->d_ob.p,
                       (const RawWin*)(runs.empty() ? nullptr : b->d_rawwin.p));
  rc = device_init_after_upload(b, s);
  if (timing) {
    const auto tt3 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point x, std::chrono::steady_clock::time_point y) { return std::chrono::duration<double, std::milli>(y - x).count(); };
    std::fprintf(stderr, "slslam refill (device build, %s): sizes + wait %.2f  stage %.2f  commit + enqueue %.2f ms  (%d windows)\n",
                 all_pinned ? (runs.empty() ? "zero copy" : "pinned arrays, copy engine") : "staged", ms(tt0, tt1), ms(tt1, tt2), ms(tt2, tt3), B);
  }
  return rc;
}

}  // namespace

// ------------------------------------------------------------------------------------------
// A batch for a stream of windows: every window replaced, nothing allocated, nothing captured again (include/slslam_hip.h).
extern "C" int slslam_lba_batch_refill(slslam_lba_batch* b, const slslam_lba_window* windows, int n, void* stream) {
  if (!b || (!windows && n > 0)) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  if (!b->refillable || b->part[0] || b->big_mode || b->fused_motion_only || b->opt.reuse_elimination) return SLSLAM_ERR_UNSUPPORTED;
  const int B = (int)b->wins.size();
  if (n != B) return SLSLAM_ERR_UNSUPPORTED;              // the launches of the captured solve were made for this many windows
  HIP_TRY(hipSetDevice(b->device));
  hipStream_t s = (hipStream_t)stream;
  {
    // the build stage on the device when the windows allow it (lba_device_build.h); UNSUPPORTED: not that path's business, nothing touched
    const int drc = refill_device(b, windows, n, s, s);
    if (drc != SLSLAM_ERR_UNSUPPORTED) return drc;
  }
  static const bool timing = std::getenv("SLSLAM_REFILL_TIMING") != nullptr;      // host-side split of a refill on stderr
  const auto tt0 = std::chrono::steady_clock::now();
  if (b->ev_stage_free) HIP_TRY(hipEventSynchronize(b->ev_stage_free));      // the host image may still feed the previous refill's copies
  // (an asynchronous download still under way has arrived before the host image is touched - and stays readable: a refill that is refused
  // below leaves the batch as it was, its results included)
  if (b->results_pending) { HIP_TRY(hipEventSynchronize(b->ev_results)); b->results_pending = false; adopt_device_build(b); b->downloaded = true; if (b->profiling) b->harvest_events(); }
  const auto tt1 = std::chrono::steady_clock::now();
  // ---- pack (the LBAProblem::build stage, one window per host thread), the observations straight into the host image: where a window's
  // observations go only depends on the counts before it
  std::vector<long long> obs_off((size_t)B + 1, 0);
  for (int i = 0; i < B; ++i) {
    if (windows[i].num_observations < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
    obs_off[(size_t)i + 1] = obs_off[(size_t)i] + windows[i].num_observations;
  }
  const HostImage img = host_image(b);
  if (obs_off[(size_t)B] > img.ob_stride) return SLSLAM_ERR_UNSUPPORTED;
  const int grouping = b->elim_grouped ? 1 : 0;
  std::vector<PackedWindow>& wins = b->wins_spare;      // (the windows of the refill before last: their vectors are reused)
  wins.resize((size_t)B);
  std::vector<int> st((size_t)B, SLSLAM_OK);
  HostPool* pool = batch_pool(b, (int)std::min<long long>(B, b->opt.host_threads > 0 ? b->opt.host_threads : (B >= 64 ? 8 : 1)));
  // (the observation region of the pinned image holds the windows' RAW observations, one after the other, after a refill: the device permutes)
  static const bool host_gather = std::getenv("SLSLAM_REFILL_HOST_GATHER") != nullptr;      // (measurement switch: the gather on the host threads, as a fresh batch is built)
  const bool device_gather = b->d_ob_raw.n >= (size_t)8 * (size_t)obs_off[(size_t)B] && b->d_ob_raw.p && !host_gather;
  auto pack_one = [&](int i) {
    ObPlanes dest;
    for (int q = 0; q < 4; ++q) dest.plane[q] = img.ob + ((size_t)q * (size_t)img.ob_stride + (size_t)obs_off[(size_t)i]) * 2;
    if (device_gather) dest.raw = img.ob + (size_t)obs_off[(size_t)i] * 8;
    st[(size_t)i] = pack_window(&windows[i], &wins[(size_t)i], grouping, &dest);
  };
  if (!run_all(pool, B, pack_one)) return SLSLAM_ERR_NO_MEMORY;
  for (int r : st) if (r != SLSLAM_OK) return r;
  const auto tt2 = std::chrono::steady_clock::now();
  for (const PackedWindow& P : wins) {
    if (P.big) return SLSLAM_ERR_UNSUPPORTED;                                        // (would take the global-memory path)
    if (b->elim_mode == 1 && (P.Cf > kMfmaMaxFree || P.dup_free_obs)) return SLSLAM_ERR_UNSUPPORTED;   // (the batch's sweep cannot take it)
  }
  // ---- layout, cut like the batch's first windows, and does it fit
  LayoutPlan plan;
  int rc = plan_layout(b, wins, /*frozen=*/true, &plan);
  if (rc != SLSLAM_OK) return rc;
  bool fits = (size_t)plan.ncam <= b->d_cam_cf.n && (size_t)plan.nline <= b->d_line_win.n && plan.nobs <= img.ob_stride &&
              (size_t)plan.ntiles <= b->d_tiles.n && (size_t)plan.nitems * 2 <= b->d_items.n && (int)plan.chunks.size() <= b->nchunk &&
              (size_t)plan.slab <= b->d_slab.n && (size_t)plan.sys <= b->d_ysys.n && (size_t)plan.params <= b->d_params_out.n &&
              (size_t)plan.params <= b->h_params.size() && plan.maxC <= b->cap_maxC && plan.maxn <= b->cap_maxn;
  if (b->slab_sum_stride) {
    if (b->slab_sum_image) {
      long long ext = 0;
      for (const WinDesc& wd : plan.wins) { const int N = solve_pad(wd.n); ext = std::max<long long>(ext, (long long)N * solve_stride(wd.n) + 6LL * N); }
      fits = fits && ((ext + kSlabScalars + 1) / 2) * 2 <= b->slab_sum_stride;
    } else fits = fits && (long long)plan.max_sys + kSlabScalars <= b->slab_sum_stride;
  } else fits = fits && plan.max_chunks <= 8;            // (more chunks per window would need k_slab_reduce in the captured solve)
  for (WinDesc& wd : plan.wins) {
    const int off = (size_t)(wd.n / 6) < b->sys_map_off_of_cf.size() ? b->sys_map_off_of_cf[(size_t)wd.n / 6] : -1;
    if (off < 0) fits = false;
    wd.map_off = off < 0 ? 0 : off;
  }
  if (!fits) return SLSLAM_ERR_UNSUPPORTED;
  // ---- commit: the batch now IS the new windows
  b->h_wins = plan.wins; b->h_param_off = plan.param_off; b->h_ob_orig_off = plan.ob_orig_off; b->h_win_graded = plan.win_graded;
  if (b->h_win_graded.empty()) b->h_win_graded.assign(1, 0);
  b->total_params = plan.params; b->nobs = plan.nobs;
  b->used_ncam = plan.ncam; b->used_nline = plan.nline; b->used_nobs = plan.nobs; b->used_tiles = plan.ntiles; b->used_items = plan.nitems;
  b->wins.swap(wins);
  b->downloaded = false;
  b->device_built = false; b->inplace_export = false; b->results_inplace = false; b->src_windows.clear(); b->build_status.clear();
  auto fill_one = [&](int wi) { fill_window(b, plan, b->wins, wi, img, /*copy_observations=*/false); };
  (void)run_all(pool, B, fill_one);                      // (fill_window copies into the image: it allocates nothing)
  fill_tail(b, plan, img);
  const auto tt3 = std::chrono::steady_clock::now();
  // ---- upload what is used of every array, asynchronously from the pinned image
  const DeviceArena& ar = b->arena;
#define SLS_UP(buf, count) do { const size_t nb_ = (size_t)(count) * sizeof(*(buf).p); \
    if (nb_) HIP_TRY(hipMemcpyAsync((void*)(buf).p, (const void*)ar.host_of(buf), nb_, hipMemcpyHostToDevice, s)); } while (0)
  SLS_UP(b->d_wins, B); SLS_UP(b->d_tiles, plan.ntiles); SLS_UP(b->d_chunks, b->nchunk); SLS_UP(b->d_items, 2 * plan.nitems);
  SLS_UP(b->d_lane_map, 64 * plan.ntiles); SLS_UP(b->d_line_desc, plan.nline);
  SLS_UP(b->d_cam_x0, 6 * plan.ncam); SLS_UP(b->d_cam_cf, b->d_cam_cf.n); SLS_UP(b->d_cam_win, b->d_cam_win.n);
  SLS_UP(b->d_line_x0, 4 * plan.nline); SLS_UP(b->d_line_ptr, plan.nline + 1); SLS_UP(b->d_line_flags, std::max<long long>(1, plan.nline));
  SLS_UP(b->d_line_win, b->d_line_win.n); SLS_UP(b->d_line_orig, plan.nline);
  SLS_UP(b->d_ob_cam, plan.nobs); SLS_UP(b->d_ob_orig, plan.nobs); SLS_UP(b->d_param_off, B);
  if (device_gather && plan.nobs > 0) {
    int maxM = 0;
    for (int i = 0; i < B; ++i) maxM = std::max(maxM, windows[i].num_observations);
    HIP_TRY(hipMemcpyAsync(b->d_ob_raw.p, img.ob, (size_t)plan.nobs * 8 * sizeof(double), hipMemcpyHostToDevice, s));
    if (maxM > 0) hipLaunchKernelGGL(k_permute_obs, dim3((unsigned)((maxM + 255) / 256), (unsigned)B), dim3(256), 0, s, b->ptrs, (const double*)b->d_ob_raw.p, (const int*)b->d_ob_orig.p, b->d_ob.p);
  } else {
    for (int q = 0; q < 4 && plan.nobs > 0; ++q)
      HIP_TRY(hipMemcpyAsync(b->d_ob.p + (size_t)q * (size_t)img.ob_stride * 2, img.ob + (size_t)q * (size_t)img.ob_stride * 2, (size_t)plan.nobs * 2 * sizeof(double), hipMemcpyHostToDevice, s));
  }
#undef SLS_UP
  if (!b->ev_stage_free) HIP_TRY(hipEventCreateWithFlags(&b->ev_stage_free, hipEventDisableTiming));
  HIP_TRY(hipEventRecord(b->ev_stage_free, s));
  // what a fresh batch finds zeroed
  HIP_TRY(hipMemsetAsync(b->d_cam_scale.p, 0, b->d_cam_scale.n * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(b->d_line_scale.p, 0, b->d_line_scale.n * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(b->d_cam_tab.p, 0, b->d_cam_tab.n * sizeof(double), s));
  if (b->slab_sum_image) HIP_TRY(hipMemsetAsync(b->d_slab_sum.p, 0, b->d_slab_sum.n * sizeof(double), s));
  HIP_TRY(hipMemsetAsync(b->d_active.p, 0, sizeof(unsigned int), s));
  rc = device_init_after_upload(b, s);
  if (timing) {
    const auto tt4 = std::chrono::steady_clock::now();
    auto ms = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point c) { return std::chrono::duration<double, std::milli>(c - a).count(); };
    std::fprintf(stderr, "slslam refill: wait %.2f  pack %.2f  plan+fill %.2f  enqueue %.2f ms  (%d windows, %d threads)\n", ms(tt0, tt1), ms(tt1, tt2), ms(tt2, tt3), ms(tt3, tt4), B, pool ? pool->threads() : 1);
  }
  return rc;
}

extern "C" int slslam_lba_batch_get_parameters(const slslam_lba_batch* b, int index, double* parameters) {
  if (!b || !parameters || index < 0 || index >= b->num_windows()) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_lba_batch_get_parameters(b->part[b->route[index].first], b->route[index].second, parameters);
  if (!b->downloaded) return SLSLAM_ERR_STATE;
  const PackedWindow& P = b->wins[index];
  const size_t n = (size_t)6 * P.C + (size_t)4 * P.L;
  if (b->device_built) {
    // a window the device build flagged (bad input, a shape for the host path, no room) was emitted empty: it has no result here
    if (b->build_status[(size_t)index] != SLSLAM_OK) return b->build_status[(size_t)index];
    // (NUMERICAL_FAILURE: k_export hands back the initial values; in place the kernel left the caller's array alone)
    if (b->results_inplace) { if (parameters != b->src_windows[(size_t)index].parameters) std::memcpy(parameters, b->src_windows[(size_t)index].parameters, n * sizeof(double)); }
    else std::memcpy(parameters, b->h_params.data() + b->h_param_off[index], n * sizeof(double));
    return SLSLAM_OK;
  }
  // Ceres leaves the user's parameters untouched on NUMERICAL_FAILURE
  if (b->h_state[index].status == SLSLAM_NUMERICAL_FAILURE) std::memcpy(parameters, P.params0.data(), n * sizeof(double));
  else std::memcpy(parameters, b->h_params.data() + b->h_param_off[index], n * sizeof(double));
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_get_summary(const slslam_lba_batch* b, int index, slslam_summary* s) {
  if (!b || !s || index < 0 || index >= b->num_windows()) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_lba_batch_get_summary(b->part[b->route[index].first], b->route[index].second, s);
  if (!b->downloaded) return SLSLAM_ERR_STATE;
  if (b->device_built && b->build_status[(size_t)index] != SLSLAM_OK) return b->build_status[(size_t)index];
  const LMState& st = b->h_state[index];
  s->num_successful_steps = st.n_success;
  s->num_unsuccessful_steps = st.n_unsuccess;
  s->initial_cost = st.initial_cost;
  s->final_cost = st.min_cost < st.initial_cost ? st.min_cost : st.initial_cost;
  s->fixed_cost = st.fixed_cost;
  s->termination_type = st.status == kRunning ? SLSLAM_NO_CONVERGENCE : st.status;
  s->num_free_parameters = b->wins[index].nfree_params;
  s->num_residual_blocks = b->wins[index].nkept;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_get_trace(const slslam_lba_batch* b, int index, slslam_iteration* trace, int cap, int* len) {
  if (!b || index < 0 || index >= b->num_windows()) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_lba_batch_get_trace(b->part[b->route[index].first], b->route[index].second, trace, cap, len);
  if (!b->downloaded) return SLSLAM_ERR_STATE;
  if (b->device_built && b->build_status[(size_t)index] != SLSLAM_OK) return b->build_status[(size_t)index];
  const int n = std::min<int>(b->h_state[index].ntrace, kMaxTrace);
  if (len) *len = n;
  for (int i = 0; trace && i < n && i < cap; ++i) {
    const IterRec& r = b->h_trace[(size_t)index * kMaxTrace + i];
    slslam_iteration& o = trace[i];
    o.iteration = r.iteration; o.step_is_valid = r.step_is_valid; o.step_is_successful = r.step_is_successful;
    o.cost = r.cost; o.cost_change = r.cost_change; o.gradient_max_norm = r.gradient_max_norm;
    o.step_norm = r.step_norm; o.relative_decrease = r.relative_decrease;
    o.trust_region_radius = r.trust_region_radius; o.model_cost_change = r.model_cost_change;
  }
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_counts(const slslam_lba_batch* b, long long* nw, long long* nc, long long* nfc,
                                       long long* nl, long long* no) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  long long w = 0, c = 0, fc = 0, l = 0, o = 0;
  for (int h = 0; h < 2; ++h)
    if (b->part[h]) for (const PackedWindow& P : b->part[h]->wins) { ++w; c += P.C; fc += P.Cf; l += P.L; o += P.M; }
  for (const PackedWindow& P : b->wins) { ++w; c += P.C; fc += P.Cf; l += P.L; o += P.M; }
  if (nw) *nw = w; if (nc) *nc = c; if (nfc) *nfc = fc; if (nl) *nl = l; if (no) *no = o;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_elimination(const slslam_lba_batch* b, int* mode) {
  if (!b || !mode || !b->finalized) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_lba_batch_elimination(b->part[0], mode);
  *mode = (b->big_mode || b->fused_motion_only) ? 0 : b->elim_mode == 0 ? 1 : b->elim_grouped ? 4 : b->elim_waves == 2 ? 3 : 2;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_path(const slslam_lba_batch* b, int* path) {
  if (!b || !path || !b->finalized) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) { *path = SLSLAM_PATH_MIXED; return SLSLAM_OK; }
  *path = b->big_mode ? SLSLAM_PATH_GLOBAL_MEMORY : b->fused_motion_only ? SLSLAM_PATH_FUSED_MOTION_ONLY : SLSLAM_PATH_TILED;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_window_chunks(const slslam_lba_batch* b, int index, int* num_chunks) {
  if (!b || !num_chunks || !b->finalized || index < 0 || index >= b->num_windows()) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_lba_batch_window_chunks(b->part[b->route[index].first], b->route[index].second, num_chunks);
  // < 0: graded sizes, -(1000 x rounds + chunks): passed back as chunks_per_window it asks for the same cut
  *num_chunks = b->h_win_graded[(size_t)index] ? -(1000 * (int)b->h_win_graded[(size_t)index] + b->h_wins[index].nchunks) : b->h_wins[index].nchunks;
  return SLSLAM_OK;
}

// Timing experiments only (SLSLAM_DEBUG_ABLATE bit 8; not part of include/slslam_hip.h): per-phase wave cycles of the last
// matrix-core sweep, summed over all waves; out[16].
extern "C" int slslam_debug_phase_cycles(slslam_lba_batch* b, double* out) {
  if (!b || !out || !b->finalized) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_debug_phase_cycles(b->part[0], out);
  const size_t n = std::max((size_t)32 * std::max(1, b->nchunk), (size_t)16 * std::max<size_t>(1, b->wins.size()));
  if (!b->pol.debug_flags) return SLSLAM_ERR_STATE;
  std::vector<unsigned long long> h(n);
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(h.data(), b->d_dbg_cycles.p, n * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  for (int i = 0; i < 16; ++i) out[i] = 0.0;
  for (size_t q = 0; q < n; ++q) out[q % 16] += (double)h[q];
  return SLSLAM_OK;
}

extern "C" int slslam_debug_read_cycles(slslam_lba_batch* b, unsigned long long* out, long long n, long long* size) {
  if (!b || !b->finalized || !size) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (b->part[0]) return slslam_debug_read_cycles(b->part[0], out, n, size);
  if (!b->pol.debug_flags) return SLSLAM_ERR_STATE;
  const long long words = (long long)std::max((size_t)32 * std::max(1, b->nchunk), (size_t)16 * std::max<size_t>(1, b->wins.size()));
  *size = words;
  if (!out || n <= 0) return SLSLAM_OK;
  HIP_TRY(hipSetDevice(b->device));
  HIP_TRY(hipDeviceSynchronize());
  HIP_TRY(hipMemcpy(out, b->d_dbg_cycles.p, (size_t)std::min(n, words) * sizeof(unsigned long long), hipMemcpyDeviceToHost));
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_set_profiling(slslam_lba_batch* b, int enable) {
  if (!b) return SLSLAM_ERR_INVALID_ARGUMENT;
  for (int h = 0; h < 2; ++h) if (b->part[h]) (void)slslam_lba_batch_set_profiling(b->part[h], enable);
  b->profiling = enable != 0;
  b->ev_used.clear(); b->ev_next = 0;                   // the events themselves are kept for reuse
  for (int f = 0; f < FAM_N; ++f) { b->fam_ms[f] = 0.0; b->fam_launches[f] = 0; }
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_kernel_times(const slslam_lba_batch* b, double ms[8], int launches[8]) {
  if (!b || !ms || !launches) return SLSLAM_ERR_INVALID_ARGUMENT;
  for (int f = 0; f < FAM_N; ++f) { ms[f] = b->fam_ms[f]; launches[f] = b->fam_launches[f]; }
  for (int h = 0; h < 2; ++h)
    if (b->part[h]) for (int f = 0; f < FAM_N; ++f) { ms[f] += b->part[h]->fam_ms[f]; launches[f] += b->part[h]->fam_launches[f]; }
  return SLSLAM_OK;
}

extern "C" int slslam_lba_batch_linearise(slslam_lba_batch* b, int index, double* residuals, double* j_cam,
                                          double* j_line, double* cost) {
  if (!b || index < 0 || index >= b->num_windows() || !residuals || !j_cam || !j_line || !cost) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (!b->finalized) return SLSLAM_ERR_STATE;
  if (b->part[0]) return slslam_lba_batch_linearise(b->part[b->route[index].first], b->route[index].second, residuals, j_cam, j_line, cost);
  HIP_TRY(hipSetDevice(b->device));
  const PackedWindow& P = b->wins[index];
  const size_t M = (size_t)P.M;
  DevBuf<double> dr, djc, djl, dc;
  int rc;
  if ((rc = dr.alloc(std::max<size_t>(1, 4 * M))) || (rc = djc.alloc(std::max<size_t>(1, 24 * M))) ||
      (rc = djl.alloc(std::max<size_t>(1, 16 * M))) || (rc = dc.alloc(1))) {
    dr.release(); djc.release(); djl.release(); dc.release();
    return rc;
  }
  // make sure the trig table of the current buffer is valid
  if (b->nline > 0) hipLaunchKernelGGL(k_line_trig, dim3((unsigned)((b->nline + 255) / 256)), dim3(256), 0, 0, b->ptrs, 0);
  hipLaunchKernelGGL(k_debug_linearise, dim3(1), dim3(64), b->lds_cost, 0, b->ptrs, b->pol, index,
                     b->d_ob_orig.p, dr.p, djc.p, djl.p, dc.p);
  hipError_t e = hipDeviceSynchronize();
  if (e == hipSuccess && M > 0) {
    e = hipMemcpy(residuals, dr.p, 4 * M * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(j_cam, djc.p, 24 * M * sizeof(double), hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = hipMemcpy(j_line, djl.p, 16 * M * sizeof(double), hipMemcpyDeviceToHost);
  }
  if (e == hipSuccess) e = hipMemcpy(cost, dc.p, sizeof(double), hipMemcpyDeviceToHost);
  dr.release(); djc.release(); djl.release(); dc.release();
  HIP_TRY(e);
  return SLSLAM_OK;
}

// ------------------------------------------------------------------------------------------
// LBAProblem::build + set_options + ceres::Solve for one window (reference src/slam.cpp:924-944)
extern "C" int slslam_lba_solve(const slslam_lba_window* w, const slslam_solver_options* opt,
                                slslam_summary* summary, slslam_iteration* trace, int trace_cap, int* trace_len) {
  if (!w) return SLSLAM_ERR_INVALID_ARGUMENT;
  slslam_solver_options o;
  if (opt) o = *opt; else slslam_default_options(&o);
  if (o.max_num_iterations < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  PackedWindow pw;   // malformed input is reported as such on any machine, before the device is looked for
  {
    const int prc = pack_window(w, &pw);
    if (prc != SLSLAM_OK) return prc;
  }
  slslam_lba_batch* b = nullptr;
  int rc = slslam_lba_batch_create(-1, &b);
  if (rc) return rc;
  o.use_graph = 0;   // a single solve is replayed once: capture would only add latency
  b->wins.push_back(std::move(pw));
  b->arena.cached = true;
  if ((rc = slslam_lba_batch_finalize(b, &o)) == SLSLAM_OK &&
      (rc = slslam_lba_batch_solve(b, nullptr)) == SLSLAM_OK &&
      (rc = slslam_lba_batch_download(b, nullptr)) == SLSLAM_OK) {
    rc = slslam_lba_batch_get_parameters(b, 0, w->parameters);
    if (summary && rc == SLSLAM_OK) rc = slslam_lba_batch_get_summary(b, 0, summary);
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_get_trace(b, 0, trace, trace_cap, trace_len);
  }
  slslam_lba_batch_destroy(b);
  return rc;
}

// ------------------------------------------------------------------------------------------
// Page-locked host memory the GPU reads and writes in place (include/slslam_hip.h)
extern "C" int slslam_pinned_alloc(size_t bytes, void** out) {
  if (!out) return SLSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  HIP_TRY(PinnedRegistry::get().alloc(bytes, out));
  return SLSLAM_OK;
}
extern "C" int slslam_pinned_free(void* p) {
  if (!p) return SLSLAM_OK;
  return PinnedRegistry::get().free(p) == 0 ? SLSLAM_OK : SLSLAM_ERR_INVALID_ARGUMENT;
}
extern "C" int slslam_pinned_register(void* p, size_t bytes) {
  if (!p || !bytes) return SLSLAM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  HIP_TRY(PinnedRegistry::get().register_range(p, bytes));
  return SLSLAM_OK;
}
extern "C" int slslam_pinned_unregister(void* p) {
  if (!p) return SLSLAM_OK;
  return PinnedRegistry::get().unregister_range(p) == 0 ? SLSLAM_OK : SLSLAM_ERR_INVALID_ARGUMENT;
}
extern "C" int slslam_pinned_contains(const void* p, size_t bytes) { return PinnedRegistry::get().contains(p, bytes) ? 1 : 0; }
extern "C" int slslam_pack_indices(int n, const int* camera_index, const int* line_index, const int* fixed_index, unsigned int* packed) {
  if (n < 0 || (n > 0 && (!camera_index || !line_index || !fixed_index || !packed))) return SLSLAM_ERR_INVALID_ARGUMENT;
  for (int i = 0; i < n; ++i) {
    const int c = camera_index[i], l = line_index[i];
    if (c < 0 || c > 0xff || l < 0 || l > 0xfffe) return SLSLAM_ERR_UNSUPPORTED;
    packed[i] = (unsigned)l | (unsigned)c << 16 | (fixed_index[2 * i] ? 1u << 24 : 0u) | (fixed_index[2 * i + 1] ? 1u << 25 : 0u);
  }
  return SLSLAM_OK;
}

// Test hook (tests/test_gpu_device_build.py): ONE window through the device build alone - k_ingest, k_build_lines / _rows / _order, k_build_layout,
// k_build_tiles on temporary device arrays - everything pack_window emits comes back in the form tests/host_math::hm_pack_g returns the
// host packer's output in, so that the two are compared byte for byte.  status_out: the BuildWin.status bits (0: built).
extern "C" int slslam_debug_device_pack(const slslam_lba_window* w, int grouping, int* out_counts /*Cf, ntiles, nitems, nfree_params, nkept*/,
                                        int* line_order, int* line_ptr, int* ob_orig, int* ob_cam, int* tiles /*4 per tile: line_begin, nlines, flags, nitems*/,
                                        unsigned char* items, int* cam_cf, int max_tiles, int max_items, unsigned short* lane_map, unsigned* line_desc,
                                        int* status_out) {
  return slslam_debug_device_pack_timed(w, grouping, out_counts, line_order, line_ptr, ob_orig, ob_cam, tiles, items, cam_cf, max_tiles, max_items, lane_map, line_desc, status_out, nullptr);
}
extern "C" int slslam_debug_device_pack_timed(const slslam_lba_window* w, int grouping, int* out_counts, int* line_order, int* line_ptr, int* ob_orig, int* ob_cam, int* tiles,
                                              unsigned char* items, int* cam_cf, int max_tiles, int max_items, unsigned short* lane_map, unsigned* line_desc,
                                              int* status_out, unsigned long long* phase_clocks /*[16] or NULL*/) {
  if (!w || !out_counts || !status_out) return SLSLAM_ERR_INVALID_ARGUMENT;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  const int C = w->num_cameras, L = w->num_lines, M = w->num_observations;
  if (C < 0 || L < 0 || M < 0 || C > kMaxCams || L > 0xfffe || M >= (1 << 24)) return SLSLAM_ERR_UNSUPPORTED;
  size_t lds_tiles = build_tiles_lds_bytes(L, L);
  const size_t lds_build = build_lds_bytes(L), lds_tiles_staged = build_tiles_lds_bytes(L, L, M, grouping ? 1 : 0);
  if (lds_build > 158 * 1024 || lds_tiles > 158 * 1024) return SLSLAM_ERR_UNSUPPORTED;
  const bool no_stage = std::getenv("SLSLAM_BUILD_TILES_UNSTAGED") != nullptr;      // (tests: the form for windows whose observations do not fit the LDS)
  const int tiles_stage_m = (lds_tiles_staged <= 158 * 1024 && !no_stage) ? M : -1;
  if (tiles_stage_m >= 0) lds_tiles = lds_tiles_staged;
  const size_t Lq = (size_t)std::max(1, L), Mq = (size_t)std::max(1, M), Cq = (size_t)std::max(1, C), np = (size_t)6 * C + (size_t)4 * L;
  const size_t cap_tiles = Lq + 8, cap_items = (size_t)std::max(1, max_items);
  DeviceArena ar;
  DevBuf<int> d_cam, d_line, d_fixed, d_cam_cf, d_cam_win, d_line_ptr, d_line_flags, d_line_win, d_line_orig, d_ob_cam, d_ob_orig, d_item_base, d_totals;
  DevBuf<double> d_obs, d_params, d_ob_raw, d_line_raw, d_cam_x0, d_line_x0;
  DevBuf<uint32_t> d_raw_idx, d_fmask, d_line_desc; DevBuf<uint8_t> d_lflags, d_items; DevBuf<uint16_t> d_lane_map;
  DevBuf<RawWin> d_raw; DevBuf<BuildWin> d_bw; DevBuf<WinDesc> d_wins; DevBuf<Tile> d_tiles; DevBuf<Chunk> d_chunks; DevBuf<long long> d_param_off;
  DevBuf<unsigned long long> d_dbg;
  DevBuf<uint32_t> d_mid_keys; DevBuf<BuildLine> d_mid_li; DevBuf<uint4> d_mid_rows; DevBuf<uint16_t> d_mid_next, d_mid_trows, d_mid_tptr; DevBuf<BuildMid> d_mid;
  ar.add(d_cam, Mq, w->camera_index, (size_t)M, 0); ar.add(d_line, Mq, w->line_index, (size_t)M, 0); ar.add(d_fixed, 2 * Mq, w->fixed_index, 2 * (size_t)M, 0);
  ar.add(d_obs, 8 * Mq, w->observations, 8 * (size_t)M, 0); ar.add(d_params, std::max<size_t>(1, np), (const double*)w->parameters, np, 0);
  ar.scratch(d_ob_raw, 8 * Mq); ar.scratch(d_raw_idx, Mq); ar.scratch(d_line_raw, 4 * Lq); ar.scratch(d_lflags, Lq); ar.scratch(d_fmask, Lq);
  ar.scratch(d_wins, 1); ar.scratch(d_tiles, cap_tiles); ar.scratch(d_chunks, 16); ar.scratch(d_items, 2 * cap_items); ar.scratch(d_lane_map, 64 * cap_tiles);
  ar.scratch(d_line_desc, Lq); ar.scratch(d_cam_x0, 6 * Cq); ar.scratch(d_cam_cf, Cq); ar.scratch(d_cam_win, Cq);
  ar.scratch(d_line_x0, 4 * Lq); ar.scratch(d_line_ptr, Lq + 1); ar.scratch(d_line_flags, Lq); ar.scratch(d_line_win, Lq); ar.scratch(d_line_orig, Lq);
  ar.scratch(d_ob_cam, Mq); ar.scratch(d_ob_orig, Mq); ar.scratch(d_param_off, 1); ar.scratch(d_item_base, 1); ar.zeroed(d_totals, 8);
  ar.scratch(d_raw, 1); ar.zeroed(d_bw, 1); ar.zeroed(d_dbg, 16);
  ar.scratch(d_mid_keys, Lq); ar.scratch(d_mid_li, Lq); ar.scratch(d_mid_rows, Lq); ar.scratch(d_mid_next, Lq); ar.scratch(d_mid_trows, Lq + 8); ar.scratch(d_mid_tptr, Lq + 8); ar.scratch(d_mid, 1);
  int rc = ar.commit();
  if (rc != SLSLAM_OK) { ar.release(); return rc; }
  RawWin r;
  std::memset(&r, 0, sizeof(r));
  r.cam = d_cam.p; r.line = d_line.p; r.fixed = d_fixed.p; r.packed = nullptr; r.obs = d_obs.p; r.params_in = d_params.p; r.params = d_params.p;
  r.C = C; r.L = L; r.M = M;
  hipError_t e = hipMemcpy(d_raw.p, &r, sizeof(r), hipMemcpyHostToDevice);
  BuildPtrs P;
  std::memset(&P, 0, sizeof(P));
  P.raw = d_raw.p; P.bw = d_bw.p; P.nwin = 1; P.grouping = grouping ? 1 : 0;
  P.ob_raw = d_ob_raw.p; P.raw_idx = d_raw_idx.p; P.line_raw = d_line_raw.p; P.lflags = d_lflags.p; P.fmask = d_fmask.p;
  P.wins = d_wins.p; P.tiles = d_tiles.p; P.chunks = d_chunks.p; P.items = d_items.p; P.lane_map = d_lane_map.p; P.line_desc = d_line_desc.p;
  P.cam_x0 = d_cam_x0.p; P.cam_cf = d_cam_cf.p; P.cam_win = d_cam_win.p;
  P.line_x0 = d_line_x0.p; P.line_ptr = d_line_ptr.p; P.line_flags = d_line_flags.p; P.line_win = d_line_win.p; P.line_orig = d_line_orig.p;
  P.ob_cam = d_ob_cam.p; P.ob_orig = d_ob_orig.p; P.param_off = d_param_off.p; P.item_base = d_item_base.p; P.totals = d_totals.p;
  P.dbg = phase_clocks ? d_dbg.p : nullptr;
  P.mid_keys = d_mid_keys.p; P.mid_li = d_mid_li.p; P.mid_rows = d_mid_rows.p; P.mid_next = d_mid_next.p; P.mid_trows = d_mid_trows.p; P.mid_tptr = d_mid_tptr.p; P.mid = d_mid.p;
  LayoutArgs a;
  std::memset(&a, 0, sizeof(a));
  a.chunks_per_window = 1; a.elim_waves = 1; a.cap_tiles = (int)cap_tiles; a.cap_items = (int)cap_items; a.cap_chunks = 16; a.cap_maxn = 6 * kMaxFreeCams;
  a.slab_sum = 0; a.cap_slab = 1LL << 40; a.cap_sys = 1LL << 40; a.nline = L; a.nobs = M; a.max_free = kMaxFreeCams;
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_build_lines, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_build_rows, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_build_order, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  if (e == hipSuccess) e = hipFuncSetAttribute((const void*)k_build_tiles, hipFuncAttributeMaxDynamicSharedMemorySize, 158 * 1024);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(k_ingest, dim3(1), dim3(256), 0, 0, P);
    hipLaunchKernelGGL(k_build_lines, dim3(1), dim3(256), build_lines_lds_bytes(L), 0, P);
    hipLaunchKernelGGL(k_build_rows, dim3(1), dim3(128), build_rows_lds_bytes(L), 0, P);
    hipLaunchKernelGGL(k_build_order, dim3(1), dim3(256), lds_build, 0, P);
    hipLaunchKernelGGL(k_build_layout, dim3(1), dim3(256), 0, 0, P, a);
    hipLaunchKernelGGL(k_build_tiles, dim3(1), dim3(256), lds_tiles, 0, P, (const int*)d_cam_cf.p, tiles_stage_m);
    e = hipDeviceSynchronize();
  }
  BuildWin bw;
  WinDesc wd;
  std::memset(&bw, 0, sizeof(bw)); std::memset(&wd, 0, sizeof(wd));
  if (e == hipSuccess) e = hipMemcpy(&bw, d_bw.p, sizeof(bw), hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = hipMemcpy(&wd, d_wins.p, sizeof(wd), hipMemcpyDeviceToHost);
  *status_out = bw.status;
  if (e == hipSuccess && phase_clocks) e = hipMemcpy(phase_clocks, d_dbg.p, 16 * sizeof(unsigned long long), hipMemcpyDeviceToHost);
  if (e == hipSuccess && bw.status == 0) {
    out_counts[0] = bw.Cf; out_counts[1] = bw.ntiles; out_counts[2] = bw.nitems; out_counts[3] = bw.nfree_params; out_counts[4] = bw.nkept;
    if (bw.ntiles > max_tiles || bw.nitems > max_items) { ar.release(); return SLSLAM_ERR_UNSUPPORTED; }
    auto dl = [&](void* dst, const void* src, size_t bytes) { if (e == hipSuccess && dst && bytes) e = hipMemcpy(dst, src, bytes, hipMemcpyDeviceToHost); };
    dl(line_order, d_line_orig.p, sizeof(int) * (size_t)L); dl(line_ptr, d_line_ptr.p, sizeof(int) * ((size_t)L + 1));
    dl(ob_orig, d_ob_orig.p, sizeof(int) * (size_t)M); dl(ob_cam, d_ob_cam.p, sizeof(int) * (size_t)M); dl(cam_cf, d_cam_cf.p, sizeof(int) * (size_t)C);
    dl(items, d_items.p, 2 * (size_t)bw.nitems); dl(lane_map, d_lane_map.p, sizeof(uint16_t) * 64 * (size_t)bw.ntiles); dl(line_desc, d_line_desc.p, sizeof(uint32_t) * (size_t)L);
    std::vector<Tile> ht((size_t)bw.ntiles);
    dl(ht.data(), d_tiles.p, sizeof(Tile) * ht.size());
    for (size_t t = 0; t < ht.size() && tiles; ++t) { tiles[4 * t] = ht[t].line_begin; tiles[4 * t + 1] = ht[t].nlines; tiles[4 * t + 2] = ht[t].flags; tiles[4 * t + 3] = ht[t].nitems; }
  }
  ar.release();
  HIP_TRY(e);
  return SLSLAM_OK;
}

// ------------------------------------------------------------------------------------------
// A STREAM of windows (BASELINE config 4 taken literally: every window arrives as the five host arrays the reference builds per
// window, src/slam.cpp:899-921): `depth` refillable batches in flight, each on a HIP stream of its own - while the GPU solves batch k
// and its copy engine uploads batch k + 1, the host threads pack batch k + 2.
struct slslam_lba_stream {
  int device = 0;
  slslam_solver_options opt;
  int depth = 3;
  std::unique_ptr<HostPool> pool;
  struct Slot {
    slslam_lba_batch* batch = nullptr;
    hipStream_t stream = nullptr;
    int n = 0;                                  // windows of the batch in the slot
    long long ticket = -1;
    bool in_flight = false;
    std::vector<double*> out_params;            // the callers' parameter arrays (solved in place, written by collect)
  };
  std::vector<Slot> slots;
  hipStream_t ingest_stream = nullptr, solve_stream = nullptr;      // shared by the slots whose refills are built on the device (see refill_device)
  hipStream_t build_stream = nullptr;        // ... the build kernels of batch k + 1 run BESIDE the solve of batch k: they are latency-bound (one wave walks a window's lines) and
                                             // leave the chip mostly idle - and a chip that idles for 3 ms between two solves starts the next one at lower clocks (measured: the
                                             // first solve after a light-load gap 16.6 ms against 15.0 back to back, tools/first_solve_after_refill.py)
  hipStream_t result_stream = nullptr;       // ... and where their results leave: the export over the link and the state / trace copies of batch k
  hipEvent_t ev_solved = nullptr, ev_built = nullptr;
  long long next_ticket = 0;
  // host-side accounting (ms, wall clock of the calling thread)
  double ms_submit = 0, ms_collect_wait = 0, ms_collect_copy = 0;
  long long n_refills = 0, n_builds = 0, n_windows = 0, n_iterations = 0;
  long long n_device_builds = 0, n_zero_copy = 0, n_fallback_windows = 0;
};

extern "C" int slslam_lba_stream_create(int device, const slslam_solver_options* opt, int depth, slslam_lba_stream** out) {
  if (!out) return SLSLAM_ERR_INVALID_ARGUMENT;
  *out = nullptr;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return SLSLAM_ERR_NO_DEVICE;
  if (device < 0) { if (hipGetDevice(&device) != hipSuccess) return SLSLAM_ERR_NO_DEVICE; }
  if (device >= ndev || depth < 1 || depth > 8) return SLSLAM_ERR_INVALID_ARGUMENT;
  slslam_lba_stream* st = new (std::nothrow) slslam_lba_stream();
  if (!st) return SLSLAM_ERR_HIP;
  st->device = device; st->depth = depth;
  if (opt) st->opt = *opt; else slslam_default_options(&st->opt);
  if (st->opt.refill_headroom_percent <= 0) st->opt.refill_headroom_percent = 10;
  st->opt.use_graph = 1;
  int threads = st->opt.host_threads;
  if (threads <= 0) threads = std::max(1, std::min(16, (int)std::thread::hardware_concurrency()));
  st->opt.host_threads = threads;
  st->pool.reset(new HostPool(threads));
  st->slots.resize((size_t)depth);
  if (hipSetDevice(device) != hipSuccess) { delete st; return SLSLAM_ERR_NO_DEVICE; }
  // TWO streams serve every slot whose batches are built on the device: ingest (the host link) and build + solve + results.  A process has
  // few hardware queues (4 unless GPU_MAX_HW_QUEUES says otherwise) and streams that share one run one behind the other: the slots' own
  // streams - what the host-packer path overlaps its uploads with - are made when that path is first taken, not before.
  // The ingest stream gets the highest priority: its few workgroups must find wave slots while a solve's thousands are queued.
  int pr_lo = 0, pr_hi = 0;
  (void)hipDeviceGetStreamPriorityRange(&pr_lo, &pr_hi);
  // (experiment: SLSLAM_BUILD_CUS = n confines the build stream to n compute units spread over the chip)
  auto make_build_stream = [&](hipStream_t* out) -> hipError_t {
    const int ncu = std::getenv("SLSLAM_BUILD_CUS") ? std::atoi(std::getenv("SLSLAM_BUILD_CUS")) : 0;
    hipDeviceProp_t prop;
    if (ncu > 0 && hipGetDeviceProperties(&prop, device) == hipSuccess && ncu < prop.multiProcessorCount) {
      const int total = prop.multiProcessorCount;
      std::vector<uint32_t> mask((size_t)(total + 31) / 32, 0u);
      for (int k = 0; k < ncu; ++k) { const int cu = (int)((long long)k * total / ncu); mask[(size_t)cu / 32] |= 1u << (cu % 32); }
      return hipExtStreamCreateWithCUMask(out, (uint32_t)mask.size(), mask.data());
    }
    return hipStreamCreateWithPriority(out, hipStreamNonBlocking, std::getenv("SLSLAM_BUILD_STREAM_PRIORITY") ? std::atoi(std::getenv("SLSLAM_BUILD_STREAM_PRIORITY")) : 0);
  };
  if (hipStreamCreateWithPriority(&st->ingest_stream, hipStreamNonBlocking, pr_hi) != hipSuccess || hipStreamCreateWithFlags(&st->solve_stream, hipStreamNonBlocking) != hipSuccess ||
      make_build_stream(&st->build_stream) != hipSuccess ||
      hipStreamCreateWithFlags(&st->result_stream, hipStreamNonBlocking) != hipSuccess || hipEventCreateWithFlags(&st->ev_solved, hipEventDisableTiming) != hipSuccess ||
      hipEventCreateWithFlags(&st->ev_built, hipEventDisableTiming) != hipSuccess) {
    if (st->ingest_stream) (void)hipStreamDestroy(st->ingest_stream);
    if (st->solve_stream) (void)hipStreamDestroy(st->solve_stream);
    if (st->build_stream) (void)hipStreamDestroy(st->build_stream);
    if (st->result_stream) (void)hipStreamDestroy(st->result_stream);
    delete st;
    return SLSLAM_ERR_HIP;
  }
  *out = st;
  return SLSLAM_OK;
}

extern "C" void slslam_lba_stream_destroy(slslam_lba_stream* st) {
  if (!st) return;
  (void)hipSetDevice(st->device);
  if (st->ingest_stream) (void)hipStreamSynchronize(st->ingest_stream);
  if (st->solve_stream) (void)hipStreamSynchronize(st->solve_stream);
  if (st->build_stream) (void)hipStreamSynchronize(st->build_stream);
  if (st->result_stream) (void)hipStreamSynchronize(st->result_stream);
  for (auto& sl : st->slots) {
    if (sl.stream) (void)hipStreamSynchronize(sl.stream);
    if (sl.batch) { sl.batch->ext_pool = nullptr; slslam_lba_batch_destroy(sl.batch); }
    if (sl.stream) (void)hipStreamDestroy(sl.stream);
  }
  if (st->ingest_stream) (void)hipStreamDestroy(st->ingest_stream);
  if (st->solve_stream) (void)hipStreamDestroy(st->solve_stream);
  if (st->build_stream) (void)hipStreamDestroy(st->build_stream);
  if (st->result_stream) (void)hipStreamDestroy(st->result_stream);
  if (st->ev_solved) (void)hipEventDestroy(st->ev_solved);
  if (st->ev_built) (void)hipEventDestroy(st->ev_built);
  delete st;
}

namespace {
int stream_submit_impl(slslam_lba_stream* st, const slslam_lba_window* windows, const unsigned int* const* packed, int n, int* ticket);
}
extern "C" int slslam_lba_stream_submit(slslam_lba_stream* st, const slslam_lba_window* windows, int n, int* ticket) {
  return stream_submit_impl(st, windows, nullptr, n, ticket);
}
extern "C" int slslam_lba_stream_submit_packed(slslam_lba_stream* st, const slslam_lba_window* windows, const unsigned int* const* packed_index, int n, int* ticket) {
  return stream_submit_impl(st, windows, packed_index, n, ticket);
}
namespace {
int stream_submit_impl(slslam_lba_stream* st, const slslam_lba_window* windows, const unsigned int* const* packed, int n, int* ticket) {
  if (!st || !windows || n <= 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  const auto t0 = std::chrono::steady_clock::now();
  // windows whose indices came narrowed and that end up on the host packer (the slot's first batch, a refill the device build does not take)
  // get their three index arrays back first
  std::vector<slslam_lba_window> expanded;
  std::vector<std::vector<int>> exp_idx;
  auto expand = [&]() -> const slslam_lba_window* {
    if (!packed) return windows;
    if (!expanded.empty()) return expanded.data();
    expanded.assign(windows, windows + n);
    exp_idx.resize((size_t)n);
    for (int i = 0; i < n; ++i) {
      if (!packed[i]) continue;
      const size_t M = (size_t)std::max(0, windows[i].num_observations);
      std::vector<int>& v = exp_idx[(size_t)i];
      v.resize(4 * M);
      for (size_t q = 0; q < M; ++q) { const unsigned int x = packed[i][q]; v[q] = (int)((x >> 16) & 0xffu); v[M + q] = (int)(x & 0xffffu); v[2 * M + 2 * q] = (int)((x >> 24) & 1u); v[2 * M + 2 * q + 1] = (int)((x >> 25) & 1u); }
      expanded[(size_t)i].camera_index = v.data(); expanded[(size_t)i].line_index = v.data() + M; expanded[(size_t)i].fixed_index = v.data() + 2 * M;
    }
    return expanded.data();
  };
  auto& sl = st->slots[(size_t)(st->next_ticket % st->depth)];
  if (sl.in_flight) return SLSLAM_ERR_STATE;             // its results have not been collected
  HIP_TRY(hipSetDevice(st->device));
  int rc = SLSLAM_ERR_UNSUPPORTED;
  hipStream_t run = st->solve_stream;                     // where this ticket's solve and download go
  if (sl.batch && sl.n == n) {
    slslam_lba_batch* b = sl.batch;
    // the build stage on the device: ingest on the stream's ONE ingest stream, build + solve + results on its ONE solve stream
    if (b->finalized && b->refillable && !b->part[0] && !b->big_mode && !b->fused_motion_only && !b->opt.reuse_elimination && (int)b->wins.size() == n) {
      static const bool serial_build = std::getenv("SLSLAM_STREAM_SERIAL_BUILD") != nullptr;      // (measurement switch: build and solve on one stream)
      hipStream_t bs = serial_build ? st->solve_stream : st->build_stream;
      rc = refill_device(b, windows, n, bs, st->ingest_stream, packed);
      if (rc == SLSLAM_OK) {
        run = st->solve_stream;
        if (bs != run) { HIP_TRY(hipEventRecord(st->ev_built, bs)); HIP_TRY(hipStreamWaitEvent(run, st->ev_built, 0)); }
      }
    }
    if (rc == SLSLAM_ERR_UNSUPPORTED) {
      // the host packer (or refused: a new batch below): its uploads overlap the other slots' solves on a stream of the slot's own
      if (!sl.stream) HIP_TRY(hipStreamCreateWithFlags(&sl.stream, hipStreamNonBlocking));
      rc = slslam_lba_batch_refill(b, expand(), n, (void*)sl.stream);
      if (rc == SLSLAM_OK) run = sl.stream;
    }
  }
  if (rc == SLSLAM_OK) { ++st->n_refills; if (sl.batch->device_built) { ++st->n_device_builds; if (sl.batch->ingest_pinned) ++st->n_zero_copy; } }
  else if (rc == SLSLAM_ERR_UNSUPPORTED) {
    // the first batch of the slot, another number of windows, or windows that do not fit the room the slot's arrays have: a new batch
    if (sl.batch) { if (sl.stream) HIP_TRY(hipStreamSynchronize(sl.stream)); HIP_TRY(hipStreamSynchronize(st->solve_stream)); HIP_TRY(hipStreamSynchronize(st->build_stream)); HIP_TRY(hipStreamSynchronize(st->result_stream)); sl.batch->ext_pool = nullptr; slslam_lba_batch_destroy(sl.batch); sl.batch = nullptr; }
    slslam_lba_batch* b = nullptr;
    if ((rc = slslam_lba_batch_create(st->device, &b)) != SLSLAM_OK) return rc;
    b->ext_pool = st->pool.get();
    b->wins.resize((size_t)n);
    std::vector<int> ps((size_t)n, SLSLAM_OK);
    const slslam_lba_window* ew = expand();
    if (!st->pool->run(n, [&](int i) { ps[(size_t)i] = pack_window(&ew[i], &b->wins[(size_t)i]); })) rc = SLSLAM_ERR_NO_MEMORY;
    for (int r : ps) if (r != SLSLAM_OK) rc = r;
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_finalize(b, &st->opt);
    if (rc != SLSLAM_OK) { b->ext_pool = nullptr; slslam_lba_batch_destroy(b); return rc; }
    sl.batch = b; sl.n = n;
    ++st->n_builds;
  } else return rc;
  if ((rc = slslam_lba_batch_solve(sl.batch, (void*)run)) != SLSLAM_OK) return rc;
  if (run == st->solve_stream) {
    // the results leave on a stream of their own, behind the solve: the next batch's build does not wait for this batch's export
    HIP_TRY(hipEventRecord(st->ev_solved, run));
    HIP_TRY(hipStreamWaitEvent(st->result_stream, st->ev_solved, 0));
    run = st->result_stream;
  }
  if ((rc = download_async_impl(sl.batch, (void*)run, /*allow_inplace=*/true)) != SLSLAM_OK) return rc;
  sl.out_params.resize((size_t)n);
  for (int i = 0; i < n; ++i) sl.out_params[(size_t)i] = windows[i].parameters;
  sl.ticket = st->next_ticket; sl.in_flight = true;
  if (ticket) *ticket = (int)st->next_ticket;
  ++st->next_ticket; st->n_windows += n;
  st->ms_submit += std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
  return SLSLAM_OK;
}
}  // namespace

extern "C" int slslam_lba_stream_collect(slslam_lba_stream* st, int ticket, slslam_summary* summaries) {
  if (!st || ticket < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  auto& sl = st->slots[(size_t)(ticket % st->depth)];
  if (!sl.in_flight || sl.ticket != ticket) return SLSLAM_ERR_STATE;
  HIP_TRY(hipSetDevice(st->device));
  const auto t0 = std::chrono::steady_clock::now();
  int rc = slslam_lba_batch_wait(sl.batch);
  if (rc != SLSLAM_OK) return rc;
  const auto t1 = std::chrono::steady_clock::now();
  std::vector<int> rs((size_t)sl.n, SLSLAM_OK);
  // (the step counts through get_summary, which routes a window of a MIXED batch - oversize windows among ordinary ones - to the part
  // that solved it: the top-level batch of such a slot holds no LM states of its own)
  std::vector<int> steps((size_t)sl.n, 0);
  slslam_lba_batch* bt = sl.batch;
  bool dev = bt->device_built && !bt->part[0];
  std::vector<int> flagged;
  if (dev) for (int i = 0; i < sl.n; ++i) if (bt->build_status[(size_t)i] != SLSLAM_OK) flagged.push_back(i);
  // A refill that did not fit the room the slot's arrays have (only the device knows the tiles a set needs: k_build_layout flags the refill as
  // a whole) is what submit answers with a new batch when the HOST can see it: the same here, late - the set is packed by the host threads into
  // a batch of its own size (plus the stream's headroom), solved as ONE batch, and that batch takes the slot, so that the sets that follow fit.
  // (Solving 1024 flagged windows one by one would take a second, and the next set of that shape would be flagged again.)
  bool whole_nofit = dev && sl.n > 0 && (int)flagged.size() == sl.n && bt->h_buildwin && (int)bt->src_windows.size() == sl.n;
  for (int i = 0; whole_nofit && i < sl.n; ++i) if (!(bt->h_buildwin[i].status & kBuildNoFit)) whole_nofit = false;
  if (whole_nofit) {
    const int n = sl.n;
    std::vector<slslam_lba_window> ws(bt->src_windows);
    std::vector<std::vector<int>> idx((size_t)n);
    for (int i = 0; i < n; ++i) {
      const RawWin& r = bt->host_src[(size_t)i];
      ws[(size_t)i].parameters = sl.out_params[(size_t)i];
      if (ws[(size_t)i].num_observations > 0 && (!ws[(size_t)i].camera_index || !ws[(size_t)i].line_index || !ws[(size_t)i].fixed_index)) {
        if (!r.packed) return SLSLAM_ERR_STATE;
        const size_t M = (size_t)r.M;
        std::vector<int>& v = idx[(size_t)i];
        v.resize(4 * M);
        for (size_t q = 0; q < M; ++q) { const uint32_t x = r.packed[q]; v[q] = (int)((x >> 16) & 0xffu); v[M + q] = (int)(x & 0xffffu); v[2 * M + 2 * q] = (int)((x >> 24) & 1u); v[2 * M + 2 * q + 1] = (int)((x >> 25) & 1u); }
        ws[(size_t)i].camera_index = v.data(); ws[(size_t)i].line_index = v.data() + M; ws[(size_t)i].fixed_index = v.data() + 2 * M;
      }
    }
    slslam_lba_batch* nb = nullptr;
    if ((rc = slslam_lba_batch_create(st->device, &nb)) != SLSLAM_OK) return rc;
    nb->ext_pool = st->pool.get();
    nb->wins.resize((size_t)n);
    std::vector<int> ps((size_t)n, SLSLAM_OK);
    if (!st->pool->run(n, [&](int i) { ps[(size_t)i] = pack_window(&ws[(size_t)i], &nb->wins[(size_t)i]); })) rc = SLSLAM_ERR_NO_MEMORY;
    for (int r : ps) if (r != SLSLAM_OK) rc = r;
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_finalize(nb, &st->opt);
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_solve(nb, (void*)st->solve_stream);
    if (rc == SLSLAM_OK) rc = download_async_impl(nb, (void*)st->solve_stream, /*allow_inplace=*/false);
    if (rc == SLSLAM_OK) rc = slslam_lba_batch_wait(nb);
    if (rc != SLSLAM_OK) { nb->ext_pool = nullptr; slslam_lba_batch_destroy(nb); return rc; }
    if (sl.stream) HIP_TRY(hipStreamSynchronize(sl.stream));
    HIP_TRY(hipStreamSynchronize(st->solve_stream)); HIP_TRY(hipStreamSynchronize(st->build_stream)); HIP_TRY(hipStreamSynchronize(st->result_stream));
    bt->ext_pool = nullptr; slslam_lba_batch_destroy(bt);
    sl.batch = bt = nb;
    ++st->n_builds; st->n_fallback_windows += n;
    dev = false; flagged.clear();
  }
  auto one = [&](int i) {
    if (dev && bt->build_status[(size_t)i] != SLSLAM_OK) return;                 // below
    int r = SLSLAM_OK;
    if (!(dev && bt->results_inplace)) r = slslam_lba_batch_get_parameters(bt, i, sl.out_params[(size_t)i]);     // (in place: the device wrote them there)
    slslam_summary sm;
    if (r == SLSLAM_OK) r = slslam_lba_batch_get_summary(bt, i, &sm);
    if (r == SLSLAM_OK) { steps[(size_t)i] = sm.num_successful_steps + sm.num_unsuccessful_steps; if (summaries) summaries[i] = sm; }
    rs[(size_t)i] = r;
  };
  if (dev && bt->results_inplace) { for (int i = 0; i < sl.n; ++i) one(i); }      // (summaries only: not worth waking the pool)
  else st->pool->run(sl.n, one);
  // windows the device build flagged (a camera that sees a line twice, a line with more than 64 observations, more than 20 free cameras,
  // no room in the slot's arrays, bad input): solved here through the host path, one by one, from what the device read (the caller's
  // page-locked arrays, or the staging copy) - their status is whatever that path says
  for (int i : flagged) {
    const RawWin& r = bt->host_src[(size_t)i];
    std::vector<int> cam, line, fixed;
    slslam_lba_window w{};
    w.num_cameras = r.C; w.num_lines = r.L; w.num_observations = r.M; w.observations = r.obs; w.parameters = sl.out_params[(size_t)i];
    if (r.packed) {
      cam.resize((size_t)r.M); line.resize((size_t)r.M); fixed.resize(2 * (size_t)r.M);
      for (int q = 0; q < r.M; ++q) { const uint32_t v = r.packed[q]; line[(size_t)q] = (int)(v & 0xffffu); cam[(size_t)q] = (int)((v >> 16) & 0xffu); fixed[2 * (size_t)q] = (v >> 24) & 1u; fixed[2 * (size_t)q + 1] = (v >> 25) & 1u; }
      w.camera_index = cam.data(); w.line_index = line.data(); w.fixed_index = fixed.data();
    } else { w.camera_index = r.cam; w.line_index = r.line; w.fixed_index = r.fixed; }
    slslam_solver_options o = st->opt;
    o.refill_headroom_percent = 0; o.host_threads = 1; o.device_build = -1;
    slslam_summary sm;
    const int fr = slslam_lba_solve(&w, &o, &sm, nullptr, 0, nullptr);
    if (fr == SLSLAM_OK) { steps[(size_t)i] = sm.num_successful_steps + sm.num_unsuccessful_steps; if (summaries) summaries[i] = sm; }
    rs[(size_t)i] = fr;
    ++st->n_fallback_windows;
  }
  for (int r : rs) if (r != SLSLAM_OK) rc = r;
  for (int i = 0; i < sl.n; ++i) st->n_iterations += steps[(size_t)i];   // reference src/slam.cpp:949-950
  sl.in_flight = false;
  const auto t2 = std::chrono::steady_clock::now();
  st->ms_collect_wait += std::chrono::duration<double, std::milli>(t1 - t0).count();
  st->ms_collect_copy += std::chrono::duration<double, std::milli>(t2 - t1).count();
  return rc;
}

extern "C" int slslam_lba_stream_build_stats(const slslam_lba_stream* st, long long* device_builds, long long* zero_copy, long long* fallback_windows) {
  if (!st) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (device_builds) *device_builds = st->n_device_builds;
  if (zero_copy) *zero_copy = st->n_zero_copy;
  if (fallback_windows) *fallback_windows = st->n_fallback_windows;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_stream_batch(slslam_lba_stream* st, int ticket, slslam_lba_batch** batch) {
  if (!st || !batch || ticket < 0) return SLSLAM_ERR_INVALID_ARGUMENT;
  auto& sl = st->slots[(size_t)(ticket % st->depth)];
  if (sl.ticket != ticket || !sl.batch) return SLSLAM_ERR_STATE;
  *batch = sl.batch;
  return SLSLAM_OK;
}

extern "C" int slslam_lba_stream_stats(const slslam_lba_stream* st, double* ms_submit, double* ms_collect_wait, double* ms_collect_copy,
                                       long long* refills, long long* builds, long long* windows, long long* lm_iterations, int* host_threads) {
  if (!st) return SLSLAM_ERR_INVALID_ARGUMENT;
  if (ms_submit) *ms_submit = st->ms_submit;
  if (ms_collect_wait) *ms_collect_wait = st->ms_collect_wait;
  if (ms_collect_copy) *ms_collect_copy = st->ms_collect_copy;
  if (refills) *refills = st->n_refills;
  if (builds) *builds = st->n_builds;
  if (windows) *windows = st->n_windows;
  if (lm_iterations) *lm_iterations = st->n_iterations;
  if (host_threads) *host_threads = st->pool ? st->pool->threads() : 1;
  return SLSLAM_OK;
}
