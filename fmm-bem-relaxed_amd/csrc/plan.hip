// plan.hip -- implementation of the C ABI declared in include/fmmbem.h.
// Host orchestration only: builds the HostPlan, tabulates the translation operators, uploads
// everything once, and sequences the gfx950 kernels of kernels_near.hip / kernels_far.hip on a HIP
// stream.  There is NO CPU execution path: without a device, execute returns FMMBEM_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <complex>
#include <cstdio>
#include <cstring>
#include <cstdlib>
#include <future>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/fmmbem.h"
#include "device_launch.hpp"
#include "host_plan.hpp"
#include "host_tables.hpp"
#include "m2l_layout.hpp"
#include "m2l_rot.hpp"
#include "shift_ops.hpp"

using namespace fmmbem;
using namespace fmmbem::tables;

namespace {
thread_local std::string g_last_error;
}
namespace fmmbem {
int fail(int code, const std::string& msg) {          // also used by mesh_io.cpp
  g_last_error = msg;
  return code;
}
struct SolverWs;                                      // krylov.hip
void solver_ws_destroy(SolverWs* ws);
int plan_solver_info(fmmbem_plan* plan, int* device, int64_t* unknowns, int* p_max, SolverWs*** slot);
}  // namespace fmmbem

namespace {

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? FMMBEM_ERR_ALLOC : FMMBEM_ERR_HIP,                  \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                              \
  } while (0)

// The C ABI takes a device ordinal per plan; every entry point that touches the device switches to it and restores the
// caller's current device on the way out (a process may hold plans on several devices, and torch keeps its own idea of
// the current device).
struct DeviceGuard {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceGuard(int dev) {
    err = hipGetDevice(&prev);
    if (err != hipSuccess) { prev = -1; return; }
    if (prev != dev) err = hipSetDevice(dev); else prev = -1;
  }
  ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
  DeviceGuard(const DeviceGuard&) = delete;
  DeviceGuard& operator=(const DeviceGuard&) = delete;
};
#define DEVICE_SCOPE(dev)        \
  DeviceGuard device_guard_(dev); \
  HIP_TRY(device_guard_.err)

double now_ms() {
  using namespace std::chrono;
  return duration<double, std::milli>(steady_clock::now().time_since_epoch()).count();
}

}  // namespace

// the class tables of a plan are a few thousand independent rows of harmonics: a few threads take them in contiguous shares
template <class F>
static void parallel_rows(int64_t n, int64_t min_per_thread, F&& body) {
  const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency())), n / std::max<int64_t>(1, min_per_thread)));
  if (nt <= 1) { body((int64_t)0, n); return; }
  host_parallel(nt, [&](int t) { body(n * t / nt, n * (t + 1) / nt); });
}

// Tables that depend on the expansion ORDER alone -- the constant streams of the three rotation kernels, the lanes' tables of the
// one-pair-per-wavefront shifts, the lane and scatter maps of the double-sum M2L: ~30 ms of host arithmetic that used to sit on
// every plan's critical path.  Built once per process, on a thread of their own that fmmbem_plan_create starts BEFORE it builds
// the tree; to_device waits for them where it uploads them.
struct OrderTables {
  std::vector<double> rot_all, ups, dns;
  int rot_off[kRotPmax] = {}, shift_off[kRotPmax] = {};
  std::vector<double> urc, uxc, drc, dxc;
  std::vector<int32_t> urs, uxs, drs, dxs;
  size_t sl_rot_off[kShiftLanesPmax + 1] = {}, sl_ax_off[kShiftLanesPmax + 1] = {};
  std::vector<int32_t> lanes, scat;
  int scat_off[kPmax] = {};
  bool lanes_ok = true;
};
static std::shared_ptr<const OrderTables> build_order_tables() {
  auto t = std::make_shared<OrderTables>();
  std::vector<double> one;
  for (int q = 1; q <= kRotPmax; ++q) {
    t->rot_off[q - 1] = (int)t->rot_all.size();
    build_rot_stream(q, one); t->rot_all.insert(t->rot_all.end(), one.begin(), one.end());
    t->shift_off[q - 1] = (int)t->ups.size();
    build_rot_stream(q, one, kRotM2M); t->ups.insert(t->ups.end(), one.begin(), one.end());
    // both shifts have the same number of axial terms: one offset table serves the two streams
    build_rot_stream(q, one, kRotL2L); t->dns.insert(t->dns.end(), one.begin(), one.end());
  }
  ShiftLaneTables lt;
  for (int q = 1; q <= kShiftLanesPmax; ++q) {
    t->sl_rot_off[q] = t->urc.size(); t->sl_ax_off[q] = t->uxc.size();
    build_shift_lane_tables(q, kRotM2M, lt);
    t->urc.insert(t->urc.end(), lt.rot_c.begin(), lt.rot_c.end()); t->urs.insert(t->urs.end(), lt.rot_s.begin(), lt.rot_s.end());
    t->uxc.insert(t->uxc.end(), lt.ax_c.begin(), lt.ax_c.end()); t->uxs.insert(t->uxs.end(), lt.ax_s.begin(), lt.ax_s.end());
    build_shift_lane_tables(q, kRotL2L, lt);
    t->drc.insert(t->drc.end(), lt.rot_c.begin(), lt.rot_c.end()); t->drs.insert(t->drs.end(), lt.rot_s.begin(), lt.rot_s.end());
    t->dxc.insert(t->dxc.end(), lt.ax_c.begin(), lt.ax_c.end()); t->dxs.insert(t->dxs.end(), lt.ax_s.begin(), lt.ax_s.end());
  }
  std::vector<int32_t> m;
  for (int p = 1; p <= kPmax; ++p) {
    if (!m2l_lane_map(p, m)) t->lanes_ok = false;
    t->lanes.insert(t->lanes.end(), m.begin(), m.end());
    m2l_scatter_map(p, m);
    t->scat_off[p - 1] = (int)t->scat.size();
    t->scat.insert(t->scat.end(), m.begin(), m.end());
  }
  return t;
}
static std::shared_future<std::shared_ptr<const OrderTables>> order_tables() {
  static std::mutex mu;
  static std::shared_future<std::shared_ptr<const OrderTables>> fut;
  std::lock_guard<std::mutex> lock(mu);
  if (!fut.valid()) fut = std::async(std::launch::async, build_order_tables).share();
  return fut;
}

// What a plan holds that depends on the GEOMETRY and the options only -- the octree, the permutation, every pair list, the work
// items and run descriptors, the panels' points, the operator tables -- on the host and in HBM.  Plans of the same panels that
// differ in the boundary-condition flags alone (the drivers' right-hand-side plan beside the operator, examples/LaplaceBEM.cpp:
// 209-232, StokesBEM.cpp:266-270) share ONE of these through a reference count (fmmbem_plan_create_like, and fmmbem_plan_create
// itself when it recognises the geometry of a live plan); freed with the last plan that points to it.
struct PlanShared {
  HostPlan hp;
  std::vector<void*> allocs;
  int device = 0;
  bool on_device = false;
  uint64_t fingerprint[2] = {0, 0};                            // of the vertex bytes (original order), 0 0: not taken
  ~PlanShared() {
    if (!on_device) return;
    DeviceGuard guard(device);
    for (void* p : allocs) (void)hipFree(p);
  }
};

struct fmmbem_plan {
  fmmbem_options opts;
  std::shared_ptr<PlanShared> shared;
  HostPlan& hp;                                                // = shared->hp
  DevicePlan d;
  bool on_device = false;
  bool has_bc[2] = {false, false};                             // boundary-condition flags present among THIS plan's panels
  std::vector<void*> allocs;                                   // what this plan alone owns: everything that depends on the flags
  std::vector<void*>* alloc_list = nullptr;                    // where upload() / alloc() record: shared->allocs or allocs
  int64_t near_total_doubles = 0, sym_total_doubles = 0;       // sizes of the stored near blocks (allocated per plan)
  fmmbem_plan() : shared(std::make_shared<PlanShared>()), hp(shared->hp) { alloc_list = &shared->allocs; }
  explicit fmmbem_plan(std::shared_ptr<PlanShared> sh) : shared(std::move(sh)), hp(shared->hp) { alloc_list = &allocs; }   // the handle of a multi-device plan
  // A plan over SEVERAL devices of one process (fmmbem_options.n_devices > 1): this handle then owns one shard plan per device and
  // does the copies between them itself (MultiDevice, further down); everything else in this struct belongs to single-device plans
  std::shared_ptr<struct MultiDevice> multi;
  fmmbem_plan(const fmmbem_plan&) = default;                   // used by like(): shares `shared`; like() then replaces what must not be shared
  std::vector<std::pair<int, int>> m2m_launch, l2l_launch;     // (first, count) per level
  std::vector<std::pair<int, int>> m2m_shared_launch;          // sharded upward pass: parents spanning shards
  // the same launches for the rotation kernels (kernels_m2l_rot.hip with FMMBEM_ROT_OP = 1, 2): (first item, items, pairs)
  struct ShiftRot { int item_first = 0, n_items = 0, pairs = 0, level_boxes = 0, pair_first = 0, unit_first = 0, n_units = 0; };   // level_boxes: boxes of the child level in the WHOLE tree; units: parents (M2M) / pairs (L2L) of the one-pair-per-wavefront kernel
  std::vector<ShiftRot> m2m_rot, m2m_shared_rot, l2l_rot;
  const int *up_rsrc = nullptr, *up_rcls = nullptr, *up_rtgt = nullptr, *up_ritem = nullptr;
  const int *dn_rsrc = nullptr, *dn_rcls = nullptr, *dn_rtgt = nullptr, *dn_ritem = nullptr;
  // one pair per WAVEFRONT (kernels_shift.hip, shift_lanes.hpp): the same bits as the one-pair-per-lane kernels, ~5 us for a
  // level of a few thousand pairs where a pass of those takes ~20 whatever it holds; LDS-bound above (N = 1M, p = 10, L2L: 5 800
  // pairs 10.6 us against 16-20, 21 000 pairs 26 against 19.5, 39 000 pairs 42 against 23.6).  A launch of up to shift_lanes_max
  // pairs takes it -- counted on THIS plan's share of the level: the two kernels give the same bits, so a shard may choose for
  // itself (FMMBEM_SHIFT_LANES=0: never; FMMBEM_SHIFT_LANES_MAX).  Pairs are counted per live expansion slot
  // (Stokes: four, config 4 at p = 8 with the slots uncounted: M2M 0.09 -> 0.18 ms)
  // Thresholds from tools/shift_lanes_sweep.sh (profiles/r04k_shift_lanes_sweep.txt; M2M / L2L ms for one GPU and one rank of eight):
  // L2L gains at every order up to 16 384 pair-slots; M2M -- whose wavefronts also walk or share a parent's children -- only at
  // p = 9, 10, elsewhere the rotation kernel's pass is short enough and only launches of <= 2 048 pair-slots go over.
  bool shift_lanes = true;
  int shift_lanes_max = -1;                           // FMMBEM_SHIFT_LANES_MAX: one threshold for both passes and all orders (sweeps)
  int lanes_max(int p, bool m2m) const { return shift_lanes_max >= 0 ? shift_lanes_max : !m2m ? 16384 : p == 10 ? 16384 : p == 9 ? 8192 : 2048; }
  const int* up_unit_ptr = nullptr;
  const double *sl_up_class = nullptr, *sl_dn_class = nullptr, *sl_up_rc = nullptr, *sl_up_xc = nullptr, *sl_dn_rc = nullptr, *sl_dn_xc = nullptr;
  const int32_t *sl_up_rs = nullptr, *sl_up_xs = nullptr, *sl_dn_rs = nullptr, *sl_dn_xs = nullptr;
  size_t sl_rot_off[kShiftLanesPmax + 1] = {}, sl_ax_off[kShiftLanesPmax + 1] = {};
  const double *up_rec = nullptr, *dn_rec = nullptr, *up_stream = nullptr, *dn_stream = nullptr;
  int shift_stream_off[12] = {};
  int shift_rot_min = 2048;                                    // boxes on a tree level from which the rotation kernels take it
  bool shift_rot = true;
  // hipGraphs of the launch chain between the gather and the delivery of a matvec (those two take the caller's pointers),
  // one per (region of the execute, order p, exchange buffer): captured on own_stream the SECOND time a region runs at an
  // order (the first run goes out launch by launch, so that every one-time initialisation inside the launchers has happened),
  // then replayed on the caller's stream.  Off unless fmmbem_plan_set_graphs / FMMBEM_GRAPH=1: on this runtime dependent
  // launches already run back to back on the device (profiles/r03b), what a graph saves is host time per matvec.
  struct GraphEntry { int region, p; const void* buf; int runs; hipGraphExec_t exec; };
  std::vector<GraphEntry> graphs;
  bool use_graphs = false;
  int m2m_pass(int p, bool shared, hipStream_t s);
  int l2l_pass(int p, hipStream_t s);
  const DevicePlan* d_dev = nullptr;                            // copy of d in device memory
  bool split_upward = false;                                   // P2M/M2M sharded by owner, multipoles all-gathered by the caller
  unsigned pending_mask = 0;                                   // stages recorded by the upward half of a split execute
  bool pending_near = false;                                   // split execute: the near field already ran (fmmbem_plan_near_split_device)
  bool result_slices = false;                                  // execute delivers the owned rows in tree order (fmmbem_plan_set_result_slices)
  int64_t* d_cut = nullptr;                                    // tree-order row cuts of all shards, on the device
  std::vector<ShiftOpDev> up_ops, down_ops;                    // M2M / L2L operators, index p - 1
  int64_t near_bytes = 0;
  int64_t near_side_entries = 0;                               // matrix-free plans: listed near-regime pairs (12 bytes each)
  // hybrid near field (fmmbem_options.near_stream_fraction < 1): which leaves keep no matrix; host copies of the block offsets
  bool hybrid = false;
  std::vector<uint8_t> near_rec_host;                          // [leaf] 1 = recomputed
  std::vector<int64_t> near_off_host, sym_off_host;            // [leaf] offsets of the stored blocks (introspection)
  int64_t near_recomputed_pairs = 0;
  HybridStreams hyb;                                           // the recompute kernel and the listed entries run beside the streaming one
  int build_side_lists();
  int64_t n_classes = 0;
  double build_host_ms = 0, build_assemble_ms = 0;
  // execute state
  int timing = 0;                                              // 0 off, 1 events around every stage, 2 around the near-field kernel only
  int last_p = 0;
  static constexpr int kStages = 9, kRing = 64;                // gather spmv scatter p2m m2m mh m2l l2l l2p
  std::vector<hipEvent_t> ev;                                  // kRing sets of 2*kStages events (begin, end)
  unsigned ev_mask[kRing] = {};                                // stages actually recorded in each set
  int64_t ev_count = 0;                                        // executes recorded since timing was enabled
  double *stage_x = nullptr, *stage_y = nullptr;               // device staging for host-pointer execute
  fmmbem::SolverWs* solver_ws = nullptr;                       // workspace of fmmbem_gmres* on this plan (krylov.hip), kept between solves
  hipStream_t own_stream = nullptr;

  template <class T, class A>
  int upload(const std::vector<T, A>& v, const T** out) {
    *out = nullptr;
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(v.size(), 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    alloc_list->push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
    *out = static_cast<const T*>(p);
    return FMMBEM_OK;
  }
  template <class T>
  int alloc(size_t count, T** out, bool zero) {
    void* p = nullptr;
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    HIP_TRY(hipMalloc(&p, bytes));
    alloc_list->push_back(p);
    if (zero) HIP_TRY(hipMemset(p, 0, bytes));
    *out = static_cast<T*>(p);
    return FMMBEM_OK;
  }
  // which M2L an execute at order p takes: the rotation kernel for the orders it is instantiated for (p <= 12), the double-sum
  // kernels above; FMMBEM_M2L_ROT=0: the double sum at every order (A/B runs, tools/m2l_ab.py)
  int rot_max = kRotPmax;
  bool use_rot(int p) const { return p <= rot_max && m2l_rot_supported(p); }
  const double* create_vertices = nullptr;                   // fmmbem_plan_create: the caller's vertices while to_device runs (panel set-up on the device)
  // part 0: everything.  part 1: the near field's share -- panels, leaves, near lists, the assembly LAUNCHED -- as soon as the
  // host plan holds them (HostOptions::after_near_lists); part 2: the rest, when the host plan is complete
  int to_device(int part = 0);                               // the geometry's share (-> shared), then to_device_bc
  int to_device_bc(const uint8_t* bc_tree);
  int to_device_bc_begin(const uint8_t* bc_tree);
  int to_device_bc_end();
  hipEvent_t asm_ev[2] = {nullptr, nullptr};     // what depends on the boundary-condition flags (-> this plan)
  static int like(const fmmbem_plan& base, const uint8_t* bc, fmmbem_plan** out);
  static int like_finish(std::unique_ptr<fmmbem_plan> pl, const uint8_t* bc, fmmbem_plan** out);
  // phase 0: whole matvec; 1: upward half (gather, P2M, M2M of owned boxes, pack -> xbuf); 2: the rest (xbuf = gathered)
  int run(int p, const double* d_x, double* d_y, hipStream_t s, bool near_only, int phase = 0, double* xbuf = nullptr);
  ~fmmbem_plan() {
    if (solver_ws) fmmbem::solver_ws_destroy(solver_ws);
    if (on_device) {
      DeviceGuard guard(opts.device);
      for (void* p : allocs) (void)hipFree(p);
      for (auto& e : ev) (void)hipEventDestroy(e);
      for (auto& g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
      if (own_stream) (void)hipStreamDestroy(own_stream);
      if (hyb.recompute) (void)hipStreamDestroy(hyb.recompute);
      for (hipEvent_t e : {hyb.fork, hyb.join_recompute}) if (e) (void)hipEventDestroy(e);
    }
  }
};

#define TRY(expr) do { int rc_ = (expr); if (rc_ != FMMBEM_OK) return rc_; } while (0)

// Work items largest first, equal ones in their given order: a stable radix sort on the (descending) key -- the comparison
// sort of 60 000 items took 4-5 ms of every plan build
template <class T, class KeyFn>
static void stable_sort_largest_first(std::vector<T>& v, KeyFn key) {
  if (v.size() < 2) return;
  uint64_t kmax = 0;
  for (const T& x : v) kmax = std::max<uint64_t>(kmax, (uint64_t)key(x));
  std::vector<T> tmp(v.size());
  constexpr int kBits = 11, kB = 1 << kBits;
  for (int shift = 0; shift < 64 && (kmax >> shift) != 0; shift += kBits) {
    size_t count[kB + 1] = {0};
    for (const T& x : v) ++count[(((kmax - (uint64_t)key(x)) >> shift) & (kB - 1)) + 1];
    for (int b = 0; b < kB; ++b) count[b + 1] += count[b];
    for (const T& x : v) tmp[count[((kmax - (uint64_t)key(x)) >> shift) & (kB - 1)]++] = x;
    v.swap(tmp);
  }
}

// rot_tgt of a plan whose rotation pairs are its CSR lists: pair i belongs to the box b with m2l_ptr[b] <= i < m2l_ptr[b + 1]
__global__ void expand_csr_targets_kernel(const int* __restrict__ ptr, int nboxes, int* __restrict__ tgt) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= nboxes) return;
  for (int i = ptr[b]; i < ptr[b + 1]; ++i) tgt[i] = b;
}

int fmmbem_plan::to_device(int part) {
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FMMBEM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU execution path)");
  if (opts.device < 0 || opts.device >= ndev) return fail(FMMBEM_ERR_INVALID, "device ordinal out of range");
  DEVICE_SCOPE(opts.device);
  if (part != 2) {
    on_device = true;
    HIP_TRY(hipStreamCreateWithFlags(&own_stream, hipStreamNonBlocking));
    ev.assign((size_t)kRing * 2 * kStages, nullptr);
    for (auto& e : ev) HIP_TRY(hipEventCreate(&e));
  }

  const bool trace = std::getenv("FMMBEM_BUILD_TRACE") != nullptr;     // phase times on stderr (tuning aid)
  double t_last = now_ms();
  auto mark = [&](const char* what) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    const double now = now_ms();
    std::fprintf(stderr, "to_device %-28s %8.2f ms\n", what, now - t_last);
    t_last = now;
  };
  const HarmonicTables T;
  const int nl = hp.nleaves(), nb = hp.nboxes, pm = hp.opt.p_max;
  if (part != 2) {                                     // ======== the near field's share ========
  mark("streams, events, tables");
  d = DevicePlan{};
  d.n = hp.n; d.nq = hp.rule.n; d.nboxes = nb; d.nleaves = nl;
  d.p_max = pm; d.s_max = pm * (pm + 1) / 2; d.p2_max = pm * pm; d.y2_max = 4 * pm * pm;
  // stride of a panel's P2M record: S(p_max) rounded up to 8 complex = 128 bytes, so that every record of the streamed table
  // starts on a cache line (55 -> 56 at p_max = 10: 0.244 -> 0.220 ms).  Not for M and L: the M2L kernel gathers them lane by
  // lane and is 9 % SLOWER with line-aligned boxes (0.56 -> 0.61 ms at p = 10)
  d.p2m_stride = (d.s_max + 7) & ~7;
  if ((d.p2m_stride - d.s_max) * 32 > d.s_max) d.p2m_stride = d.s_max;      // more than 3 % of padding (36 -> 40 at p_max = 8) costs more than it saves
  if (opts.kernel != FMMBEM_KERNEL_STOKES_BEM) {
    // Laplace: packed records (device_plan.hpp p2m_packed) -- the zero imaginary parts of the m = 0 moments are not stored
    d.p2m_packed = 1;
    d.p2m_real_off = pm * (pm - 1) / 2;
    const int len = d.p2m_real_off + (pm + 1) / 2;                          // the reals two to a double2 slot
    d.p2m_stride = (len + 7) & ~7;
    if ((d.p2m_stride - len) * 32 > len) d.p2m_stride = len;                // the same 3 % rule: 50 slots = 800 bytes at p_max = 10, 72 = 1 152 at 12
  }
  d.leaf_begin = hp.leaf_begin; d.leaf_end = hp.leaf_end; d.row_begin = hp.row_begin; d.row_end = hp.row_end;
  for (int q = 0; q < hp.rule.n; ++q) d.qw[q] = hp.rule.w[q];
  d.kernel = opts.kernel;
  if (opts.kernel == FMMBEM_KERNEL_STOKES_BEM) {
    d.dof = 3; d.mu = opts.mu;
    QuadRule fine;
    if (!quad_rule(opts.quad_k_fine, fine)) return fail(FMMBEM_ERR_INVALID, "invalid K_fine (valid: 1 3 4 7 13 17 19 25 79)");
    d.nqf = fine.n;
    std::vector<double> qf((size_t)fine.n * 4);
    for (int q = 0; q < fine.n; ++q) { for (int k = 0; k < 3; ++k) qf[4 * q + k] = fine.pts[q][k]; qf[4 * q + 3] = fine.w[q]; }
    TRY(upload(qf, &d.qf));
  } else {
    d.dof = 1;
  }
  const int dof = d.dof;

  // panels + permutation
  TRY(upload(hp.perm, &d.perm));
  if (create_vertices) {
    // the panels' derived geometry, computed on the device from the caller's vertices (kernels_near.hip panel_setup: the host
    // form's arithmetic, bit for bit): 75 MB up instead of 0.2 GB written by the host and then uploaded
    const size_t nn = (size_t)hp.n;
    double *cx, *cy, *cz, *nx, *ny, *nz, *ar, *qd, *vt;
    TRY(alloc(nn, &cx, false)); TRY(alloc(nn, &cy, false)); TRY(alloc(nn, &cz, false));
    TRY(alloc(nn, &nx, false)); TRY(alloc(nn, &ny, false)); TRY(alloc(nn, &nz, false));
    TRY(alloc(nn, &ar, false)); TRY(alloc(nn * 3 * hp.rule.n, &qd, false)); TRY(alloc(nn * 9, &vt, false));
    double* v_orig = nullptr;
    HIP_TRY(hipMalloc(&v_orig, sizeof(double) * 9 * nn));
    std::vector<double> pts((size_t)hp.rule.n * 3);
    for (int q = 0; q < hp.rule.n; ++q) for (int k = 0; k < 3; ++k) pts[3 * q + k] = hp.rule.pts[q][k];
    const double* d_pts = nullptr;
    hipError_t e = hipMemcpy(v_orig, create_vertices, sizeof(double) * 9 * nn, hipMemcpyHostToDevice);
    int rc = e == hipSuccess ? upload(pts, &d_pts) : FMMBEM_ERR_HIP;
    if (rc == FMMBEM_OK) e = launch_panel_setup(hp.n, d.perm, v_orig, hp.rule.n, d_pts, cx, cy, cz, nx, ny, nz, ar, qd, vt, own_stream);
    if (rc == FMMBEM_OK && e == hipSuccess) e = hipStreamSynchronize(own_stream);
    (void)hipFree(v_orig);
    if (rc != FMMBEM_OK) return rc;
    if (e != hipSuccess) return fail(FMMBEM_ERR_HIP, std::string("panel setup: ") + hipGetErrorString(e));
    d.cx = cx; d.cy = cy; d.cz = cz; d.nx = nx; d.ny = ny; d.nz = nz; d.area = ar; d.quad = qd; d.vert = vt;
  } else {
    const PanelSoA& P = hp.panels;
    TRY(upload(P.cx, &d.cx)); TRY(upload(P.cy, &d.cy)); TRY(upload(P.cz, &d.cz));
    TRY(upload(P.nx, &d.nx)); TRY(upload(P.ny, &d.ny)); TRY(upload(P.nz, &d.nz));
    TRY(upload(P.area, &d.area)); TRY(upload(P.quad, &d.quad)); TRY(upload(P.vert, &d.vert));
  }

  mark("panels: upload + geometry");
  // leaves and the near block structure
  std::vector<int> leaf_row0(nl), leaf_nrows(nl), near_stride(nl), run_row0, run_off;
  std::vector<int64_t> near_off(nl, 0), run_ptr(nl + 1, 0);
  int64_t total = 0;
  int max_cols = 2, max_runs = 1;
  for (int l = 0; l < nl; ++l) {
    const int b = hp.leaf_box[l];
    leaf_nrows[l] = hp.box_body_end[b] - hp.box_body_begin[b];
  }
  // Hybrid near field: the leaves that keep NO matrix.  Chosen on the WHOLE tree (every shard builds the same tree and reaches the
  // same verdict for a leaf, so a shard's rows carry the bits of the single plan): the leaves with the most rows first -- a
  // recomputed pair costs the same arithmetic wherever it sits, but the source panel's 128 bytes are read once per item and
  // amortised over the item's rows -- until their pairs make up 1 - near_stream_fraction of all near pairs.
  std::vector<uint8_t> rec(nl, 0);
  {
    double f = opts.near_stream_fraction;
    if (const char* e = std::getenv("FMMBEM_NEAR_STREAM_FRACTION")) f = std::atof(e);
    bool rule_ok = hp.rule.n <= (dof == 3 ? 4 : 3);               // the far-regime points of a source live in registers
    for (int q = 2; q < hp.rule.n; ++q) rule_ok = rule_ok && hp.rule.w[q] == hp.rule.w[1];      // K = 1, 3, 4: two distinct weights at most
    const bool stokes_sym_on = !(std::getenv("FMMBEM_STOKES_SYM") && std::atoi(std::getenv("FMMBEM_STOKES_SYM")) == 0);
    // runs of consecutive source rows per leaf (the pipelined kernels prefetch an item's run descriptors one per thread): counted
    // by the shares of a few threads, as the runs themselves are filled in below
    std::vector<int> nruns_of(nl, 0);
    int max_runs_owned = 1;
    parallel_rows(nl, 4096, [&](int64_t l0, int64_t l1) {
      for (int64_t l = l0; l < l1; ++l) {
        int c = 0;
        for (int64_t i = hp.near_ptr[l]; i < hp.near_ptr[l + 1]; ++i) c += i == hp.near_ptr[l] || hp.near_src[i - 1] + 1 != hp.near_src[i];
        nruns_of[l] = c;
      }
    });
    for (int l = hp.leaf_begin; l < hp.leaf_end; ++l) max_runs_owned = std::max(max_runs_owned, nruns_of[l]);
    for (int l = 0; l < nl; ++l) run_ptr[l + 1] = run_ptr[l] + nruns_of[l];
    hybrid = f < 1.0 && opts.sparse_local && hp.opt.evaluator == 0 && rule_ok && (dof == 3 ? stokes_sym_on : max_runs_owned <= 256);
    if (hybrid) {
      if (!(f >= 0.0)) f = 0.0;
      std::vector<int> order(nl);
      for (int l = 0; l < nl; ++l) order[l] = l;
      std::stable_sort(order.begin(), order.end(), [&](int a, int b) { return leaf_nrows[a] > leaf_nrows[b]; });
      double all = 0;
      for (int l = 0; l < nl; ++l) all += (double)leaf_nrows[l] * hp.near_ncols[l];
      double acc = 0;
      for (int l : order) {
        if (acc >= (1.0 - f) * all) break;
        if (hp.near_ncols[l] > 8192) continue;        // a coarse leaf of an adaptive tree (10^4 columns): one item would run alone at the end
        if (nruns_of[l] > 256) continue;
        rec[l] = 1;
        acc += (double)leaf_nrows[l] * hp.near_ncols[l];
      }
      near_recomputed_pairs = 0;
      for (int l = hp.leaf_begin; l < hp.leaf_end; ++l) if (rec[l]) near_recomputed_pairs += (int64_t)leaf_nrows[l] * hp.near_ncols[l];
    }
  }
  run_row0.resize((size_t)run_ptr[nl]); run_off.resize((size_t)run_ptr[nl]);
  parallel_rows(nl, 4096, [&](int64_t l0, int64_t l1) {
    for (int64_t l = l0; l < l1; ++l) {
      const int b = hp.leaf_box[l];
      leaf_row0[l] = hp.box_body_begin[b];
      near_stride[l] = (dof * hp.near_ncols[l] + 1) & ~1;      // in unknowns (dof per panel), rows 16-B aligned
      // source leaves are ascending; leaves with consecutive indices own adjacent rows -> one run
      int col = 0;
      int64_t k = run_ptr[l];
      for (int64_t i = hp.near_ptr[l]; i < hp.near_ptr[l + 1]; ++i) {
        const int sl = hp.near_src[i], sb = hp.leaf_box[sl];
        if (i == hp.near_ptr[l] || hp.near_src[i - 1] + 1 != sl) {
          run_row0[(size_t)k] = hp.box_body_begin[sb];
          run_off[(size_t)k] = col;
          ++k;
        }
        col += hp.box_body_end[sb] - hp.box_body_begin[sb];
      }
    }
  });
  for (int l = hp.leaf_begin; l < hp.leaf_end; ++l) {
    near_off[l] = total;
    if (!rec[l]) total += (int64_t)dof * leaf_nrows[l] * near_stride[l];
    max_cols = std::max(max_cols, near_stride[l]);
    max_runs = std::max(max_runs, (int)(run_ptr[l + 1] - run_ptr[l]));
  }
  d.max_runs = max_runs;
  if (const char* e = getenv("FMMBEM_M2L_ROT")) { if (atoi(e) == 0) rot_max = 0; }
  if (const char* ge = getenv("FMMBEM_GRAPH")) use_graphs = atoi(ge) != 0;
  d.max_ncols = max_cols;
  near_bytes = total * (int64_t)sizeof(double);
  TRY(upload(leaf_row0, &d.leaf_row0)); TRY(upload(leaf_nrows, &d.leaf_nrows)); TRY(upload(hp.leaf_box, &d.leaf_box));
  TRY(upload(run_ptr, &d.near_ptr)); TRY(upload(run_row0, &d.near_run_row0)); TRY(upload(run_off, &d.near_run_off));
  TRY(upload(hp.near_ncols, &d.near_ncols)); TRY(upload(near_stride, &d.near_stride)); TRY(upload(near_off, &d.near_off));
  {
    // SpMV work items: a leaf's row block, cut into row ranges of <= kItemBytes (256 KB) so that no workgroup is left
    // streaming one coarse leaf alone (two-sphere N=1M: one leaf is 58 rows x 16 031 columns = 7.4 MB against
    // a mean of 75 KB), dealt round-robin to the persistent workgroups largest first.  Ranges shorter than 8
    // rows are processed with the columns split over the wavefronts instead of the rows.
    constexpr int64_t kItemBytes = 256 << 10;          // two-sphere N = 1M (near ms): 64 KB 0.81, 128 0.72, 192 0.74, 256 0.705, 320 0.735, 384 0.72, 512 0.72
    struct Item { int leaf, r0, nr; int64_t bytes; };
    std::vector<Item> items;
    // (a matrix-free plan runs the sweeps of kernels_near.hip over the same items, which count PANEL rows and pairs whatever
    // the number of unknowns per panel)
    const int idof = opts.sparse_local ? dof : 1;
    for (int l = hp.leaf_begin; l < hp.leaf_end; ++l) {
      const int nr = idof * leaf_nrows[l];
      const int64_t row_bytes = opts.sparse_local ? (int64_t)near_stride[l] * 8 : (int64_t)hp.near_ncols[l] * 8;
      if (nr == 0 || row_bytes == 0) continue;
      if (hybrid && (dof == 3 || rec[l])) continue;     // hybrid plans: Stokes builds its stream items below; a recomputed leaf has none
      int per = (int)std::max<int64_t>(1, kItemBytes / row_bytes);
      if (per >= 8) per &= ~7; else per = std::min(4, nr);
      const int cnt = (nr + per - 1) / per;
      per = (nr + cnt - 1) / cnt;
      if (per >= 8) per = (per + 7) & ~7;
      for (int r0 = 0; r0 < nr; r0 += per) { const int k = std::min(per, nr - r0); items.push_back({l, r0, k, k * row_bytes}); }
    }
    stable_sort_largest_first(items, [](const Item& a) { return a.bytes; });
    std::vector<int4> packed(items.size());
    for (size_t i = 0; i < items.size(); ++i) packed[i] = make_int4(items[i].leaf, items[i].r0, items[i].nr, items[i].nr < 8);
    d.near_nitems = (int)packed.size();
    TRY(upload(packed, &d.near_items));
    std::vector<NearItem> recs(items.size());
    for (size_t i = 0; i < items.size(); ++i) {
      const Item& it = items[i];
      NearItem& q = recs[i];
      q.val_off = near_off[it.leaf] + (int64_t)it.r0 * near_stride[it.leaf];
      q.run_begin = run_ptr[it.leaf];
      q.nruns = (int)(run_ptr[it.leaf + 1] - run_ptr[it.leaf]);
      q.yrow = dof * leaf_row0[it.leaf] + it.r0;
      q.nrows = it.nr;
      q.ncols = dof * hp.near_ncols[it.leaf];
      q.stride = near_stride[it.leaf];
      q.colsplit = it.nr < 8;
      q.pad[0] = q.pad[1] = 0;
    }
    TRY(upload(recs, &d.near_recs));
  }
  const bool stokes_sym = dof == 3 && opts.sparse_local && !(std::getenv("FMMBEM_STOKES_SYM") && std::atoi(std::getenv("FMMBEM_STOKES_SYM")) == 0);
  near_total_doubles = (opts.sparse_local && !stokes_sym) ? total : 0;      // allocated in to_device_bc; matrix-free mode keeps no matrix
  if (!near_total_doubles) near_bytes = 0;
  d.near_val = nullptr;
  {
    // Stokes: the near blocks in their symmetric 6-value form, the only copy (FMMBEM_STOKES_SYM=0: the 9-value rows instead)
    if (stokes_sym) {
      std::vector<int64_t> sym_off(nl, 0);
      int64_t sym_total = 0;
      constexpr int64_t kItemBytes = 512 << 10;        // red blood cell N = 524 288 (near ms): 128 KB 2.06-2.08, 256 KB 2.06-2.09, 512 KB 2.01-2.03
      struct Item { int leaf, r0, nr; int64_t bytes; };
      std::vector<Item> items;
      for (int l = hp.leaf_begin; l < hp.leaf_end; ++l) {
        const int nr = leaf_nrows[l], ncp = hp.near_ncols[l];
        sym_off[l] = sym_total;
        if (rec[l]) continue;                           // hybrid: no block, no stream item
        sym_total += (int64_t)6 * nr * ncp;
        const int64_t row_bytes = (int64_t)48 * ncp;
        if (nr == 0 || row_bytes == 0) continue;
          int per = (int)std::max<int64_t>(1, kItemBytes / row_bytes);
        if (per >= 8) per &= ~7; else per = std::min(4, nr);
        const int cnt = (nr + per - 1) / per;
        per = (nr + cnt - 1) / cnt;
        if (per >= 8) per = (per + 7) & ~7;
        for (int r0 = 0; r0 < nr; r0 += per) { const int k = std::min(per, nr - r0); items.push_back({l, r0, k, k * row_bytes}); }
      }
      stable_sort_largest_first(items, [](const Item& a) { return a.bytes; });
      std::vector<int4> packed(items.size());
      for (size_t i = 0; i < items.size(); ++i) packed[i] = make_int4(items[i].leaf, items[i].r0, items[i].nr, items[i].nr < 8);
      d.sym_nitems = (int)packed.size();
      TRY(upload(packed, &d.sym_items));
      TRY(upload(sym_off, &d.near_sym_off));
      sym_total_doubles = std::max<int64_t>(sym_total, 1);          // allocated in to_device_bc
      near_bytes = sym_total * (int64_t)sizeof(double);
      sym_off_host = sym_off;
    }
  }
  if (hybrid) {
    // recompute items: ranges of <= 20 (Stokes: four wavefronts x five rows) or 32 (Laplace: x eight) panel rows of the recomputed
    // leaves (kernels_near.hip near_recompute3 / near_recompute1)
    struct RItem { int leaf, r0, nr; int64_t pairs; };
    std::vector<RItem> ritems;
    for (int l = hp.leaf_begin; l < hp.leaf_end; ++l) {
      const int nr = leaf_nrows[l], ncp = hp.near_ncols[l];
      if (!rec[l] || nr == 0 || ncp == 0) continue;
      // Stokes, sources straight into LDS (near_recompute3g): a whole leaf of up to 60 rows is one item -- its source panels are
      // loaded once; otherwise four wavefronts x five (Laplace: eight) rows
      const int cap = dof == 3 ? rcg_item_rows() : 32;
        const int cnt = (nr + cap - 1) / cap;
      const int per = (nr + cnt - 1) / cnt;          // dealt evenly: the kernel gives a wavefront ceil(rows / 4) of an item's rows
      for (int r0 = 0; r0 < nr; r0 += per) { const int k = std::min(per, nr - r0); ritems.push_back({l, r0, k, (int64_t)k * ncp}); }
    }
    stable_sort_largest_first(ritems, [](const RItem& a) { return a.pairs; });
    // the side listing (mf_sweep COUNT / FILL) walks d.near_items: the recompute items, panel rows
    std::vector<int4> rpacked(ritems.size());
    std::vector<RcItem> rrecs(ritems.size());
    for (size_t i = 0; i < ritems.size(); ++i) {
      const RItem& it = ritems[i];
      rpacked[i] = make_int4(it.leaf, it.r0, it.nr, 0);
      RcItem& q = rrecs[i];
      q.prow0 = leaf_row0[it.leaf] + it.r0; q.nrows = it.nr; q.ncp = hp.near_ncols[it.leaf];
      q.run_begin = run_ptr[it.leaf]; q.nruns = (int)(run_ptr[it.leaf + 1] - run_ptr[it.leaf]); q.leaf = it.leaf; q.pad = 0;
    }
    d.near_nitems_stream = dof == 1 ? d.near_nitems : 0;   // (Laplace: the items built above are the streamed ones, near_recs)
      d.near_nitems = (int)rpacked.size();
    TRY(upload(rpacked, &d.near_items));
    d.rc_nitems = (int)rrecs.size();
    TRY(upload(rrecs, &d.rc_items));
    TRY(upload(rec, &d.near_rec));
    if (dof == 3) {
      // packed per-panel records of near_recompute3g_kernel (its sources go straight into LDS, 16 bytes at a time)
      double *rs = nullptr, *rn = nullptr;
      TRY(alloc((size_t)hp.n * 16, &rs, false)); TRY(alloc((size_t)hp.n * 4, &rn, false));
      HIP_TRY(launch_rc_pack(d, rs, rn, own_stream));
      HIP_TRY(hipStreamSynchronize(own_stream));
      d.rc_src = rs; d.rc_nrm = rn;
    }
  }
  near_rec_host = rec;
  near_off_host = near_off;

  mark("near lists + items");
  // boxes, expansions, tables
  TRY(upload(hp.box_center, &d.box_center));
  TRY(upload(T.A, &d.tabA)); TRY(upload(T.invA, &d.tabInvA)); TRY(upload(T.pref, &d.tabPref));
  {
    // per-step constants of the harmonic recurrences in the m-major order P2M visits (n,m) at order p:
    // pref = sqrt((n-m)!/(n+m)!) and the Legendre step P_{n+1}^m = c1 x P_n^m - c2 P_{n-1}^m (c1 = 2m+1, c2 = 0 at n = m)
    const int smax = kPmax * (kPmax + 1) / 2;
    std::vector<double> st((size_t)kPmax * (smax + 1) * 4, 0.0);
    for (int p = 1; p <= kPmax; ++p) {
      double* o = st.data() + (size_t)(p - 1) * (smax + 1) * 4;
      for (int m = 0; m < p; ++m)
        for (int n = m; n < p; ++n, o += 4) {
          o[0] = T.pref[n * n + n + m];
          o[1] = n == m ? (double)(2 * m + 1) : (double)(2 * n + 1) * (1.0 / (n - m + 1));
          o[2] = n == m ? 0.0 : (double)(n + m) * (1.0 / (n - m + 1));
        }
    }
    TRY(upload(st, &d.tabStep));
  }

  mark("boxes + harmonic tables");
  TRY(to_device_bc_begin(hp.panels.bc.data()));        // the near-matrix assembly runs on the GPU from here, under the host work below
  }                                                    // ======== the rest ========
  if (part == 1) return FMMBEM_OK;
  alloc_list = &shared->allocs;
  // far-field lists
  std::vector<int> p2m_leaf, l2p_leaf;
  for (int b : hp.p2m_leaves) p2m_leaf.push_back(hp.box_leaf_index[b]);
  for (int b : hp.l2p_leaves) l2p_leaf.push_back(hp.box_leaf_index[b]);
  d.n_p2m = (int)p2m_leaf.size(); d.n_l2p = (int)l2p_leaf.size();
  TRY(upload(p2m_leaf, &d.p2m_leaf)); TRY(upload(l2p_leaf, &d.l2p_leaf));
  {
    // L2P work groups: a leaf of this tree holds ~19 panels, a third of a wavefront; consecutive leaves are packed until
    // 64 rows or 8 leaves (Stokes: 4) (a leaf with more than 64 panels is a group of its own)
    std::vector<int> grp(1, 0);
    int rows = 0;
    const int max_leaves = l2p_group_leaves(d.kernel);
    for (size_t i = 0; i < l2p_leaf.size(); ++i) {
      const int nr = hp.box_body_end[hp.leaf_box[l2p_leaf[i]]] - hp.box_body_begin[hp.leaf_box[l2p_leaf[i]]];
      const int in_group = (int)i - grp.back();
      if (in_group > 0 && (rows + nr > 64 || in_group == max_leaves)) { grp.push_back((int)i); rows = 0; }
      rows += nr;
    }
    grp.push_back((int)l2p_leaf.size());
    d.n_l2p_grp = l2p_leaf.empty() ? 0 : (int)grp.size() - 1;
    TRY(upload(grp, &d.l2p_grp));
  }
  TRY(upload(hp.m2m_parents, &d.m2m_parent)); TRY(upload(hp.l2l_children, &d.l2l_child));
  TRY(upload(hp.box_child_begin, &d.box_child_begin)); TRY(upload(hp.box_child_end, &d.box_child_end));
  TRY(upload(hp.box_parent, &d.box_parent));
  for (size_t l = 0; l + 1 < hp.m2m_level_ptr.size(); ++l)
    if (hp.m2m_level_ptr[l + 1] > hp.m2m_level_ptr[l])
      m2m_launch.emplace_back(hp.m2m_level_ptr[l], hp.m2m_level_ptr[l + 1] - hp.m2m_level_ptr[l]);
  for (size_t l = 0; l + 1 < hp.l2l_level_ptr.size(); ++l)
    if (hp.l2l_level_ptr[l + 1] > hp.l2l_level_ptr[l])
      l2l_launch.emplace_back(hp.l2l_level_ptr[l], hp.l2l_level_ptr[l + 1] - hp.l2l_level_ptr[l]);
  for (size_t l = 0; l + 1 < hp.m2m_shared_ptr.size(); ++l)
    if (hp.m2m_shared_ptr[l + 1] > hp.m2m_shared_ptr[l])
      m2m_shared_launch.emplace_back(hp.m2m_shared_ptr[l], hp.m2m_shared_ptr[l + 1] - hp.m2m_shared_ptr[l]);
  // sharded upward pass: who sends which multipoles
  split_upward = hp.opt.shard_upward && hp.opt.shard_world > 1;
  d.xch_rank = hp.opt.shard_rank; d.xch_world = split_upward ? hp.opt.shard_world : 0; d.xch_max = 0;
  for (int r = 0; r < 9; ++r) d.xch_ptr[r] = 0;
  if (split_upward) {
    if (hp.opt.shard_world > 8) return fail(FMMBEM_ERR_UNSUPPORTED, "sharded upward pass: at most 8 shards");
    for (int r = 0; r <= hp.opt.shard_world; ++r) d.xch_ptr[r] = hp.xch_ptr[r];
    for (int r = 0; r < hp.opt.shard_world; ++r) d.xch_max = std::max(d.xch_max, hp.xch_ptr[r + 1] - hp.xch_ptr[r]);
  }
  TRY(upload(hp.xch_box, &d.xch_box));
  if (split_upward && hp.opt.shard_upward == 2) {      // selective exchange: only what the receiver reads
    TRY(upload(hp.xsel_send_box, &d.xsel_send_box)); TRY(upload(hp.xsel_recv_box, &d.xsel_recv_box));
    d.xsel_send_n = (int)hp.xsel_send_box.size(); d.xsel_recv_n = (int)hp.xsel_recv_box.size();
  }

  mark("far-field lists");
  // parent<->child translation classes and their regular-harmonic tables
  {
    std::vector<int> up_cls(nb, 0), down_cls(nb, 0);
    std::unordered_map<IVec3, int, IVec3Hash> seen;
    std::vector<cplx> up_tab, down_tab, h;
    std::vector<double> up_rec_h, dn_rec_h;              // records of the same classes for the rotation kernels
    for (int b = 1; b < nb; ++b) {
      const int par = hp.box_parent[b];
      const int32_t v[3] = {hp.box_icoord[3 * par] - hp.box_icoord[3 * b], hp.box_icoord[3 * par + 1] - hp.box_icoord[3 * b + 1],
                            hp.box_icoord[3 * par + 2] - hp.box_icoord[3 * b + 2]};
      const IVec3 key = {v[0], v[1], v[2]};
      auto [it, fresh] = seen.try_emplace(key, (int)seen.size());
      if (fresh) {
        double up[3], down[3];
        for (int k = 0; k < 3; ++k) { up[k] = 0.5 * hp.cell[k] * double(v[k]); down[k] = -up[k]; }
        up_rec_h.resize(up_rec_h.size() + 8); dn_rec_h.resize(dn_rec_h.size() + 8);
        rot_record(up, up_rec_h.data() + up_rec_h.size() - 8); rot_record(down, dn_rec_h.data() + dn_rec_h.size() - 8);
        // M2M: evalMultipole(rho, alpha, -beta) of (parent - child)   (LaplaceSpherical.hpp:253-254)
        SphHost s = cart2sph_host(up);
        harmonics(T, true, s.rho, s.alpha, -s.beta, pm, h);
        for (int n = 0; n < pm; ++n)
          for (int m = -n; m <= n; ++m) {
            const cplx y = h[(size_t)n * (n + 1) / 2 + std::abs(m)];
            up_tab.push_back(m < 0 ? std::conj(y) : y);
          }
        // L2L: evalMultipole(rho, alpha, +beta) of (child - parent)   (LaplaceSpherical.hpp:383-384)
        s = cart2sph_host(down);
        harmonics(T, true, s.rho, s.alpha, s.beta, pm, h);
        for (int n = 0; n < pm; ++n)
          for (int m = -n; m <= n; ++m) {
            const cplx y = h[(size_t)n * (n + 1) / 2 + std::abs(m)];
            down_tab.push_back(m < 0 ? std::conj(y) : y);
          }
      }
      up_cls[b] = down_cls[b] = it->second;
    }
    TRY(upload(up_cls, &d.up_cls)); TRY(upload(down_cls, &d.down_cls));
    // ---- M2M / L2L by rotation: pair lists per level launch, items, records, constant streams ----
    {
      TRY(upload(up_rec_h, &up_rec)); TRY(upload(dn_rec_h, &dn_rec));
      std::vector<int> rs, rc, rt, ri, len, uptr;
      auto level_boxes = [&](int l) { return l >= 0 && l < hp.nlevels ? hp.level_off[l + 1] - hp.level_off[l] : 0; };
      // items of the shifts: whole targets, ONE pass (at most 64 pairs) -- the shift kernels carry nothing between passes
      auto cut_single_pass = [](const std::vector<int>& seg_len, int pair_base, std::vector<int>& item_ptr, int lanes = 64) {
        item_ptr.push_back(pair_base);
        int fill = 0;
        for (int l : seg_len) {
          if (fill + l > lanes) { item_ptr.push_back(item_ptr.back() + fill); fill = 0; }
          fill += l;
        }
        if (fill > 0) item_ptr.push_back(item_ptr.back() + fill);
      };
      auto add_m2m = [&](const std::vector<std::pair<int, int>>& launches, std::vector<ShiftRot>& out) {
        for (auto [first, count] : launches) {
          ShiftRot sr;
          sr.item_first = (int)ri.size();
          const int base = (int)rs.size();
          sr.pair_first = base;
          sr.unit_first = (int)uptr.size();
          sr.n_units = count;
          len.clear();
          for (int i = first; i < first + count; ++i) {
            const int par = hp.m2m_parents[i];
            uptr.push_back((int)rs.size());
            for (int c = hp.box_child_begin[par]; c < hp.box_child_end[par]; ++c) { rs.push_back(c); rc.push_back(up_cls[c]); rt.push_back(par); }
            len.push_back(hp.box_child_end[par] - hp.box_child_begin[par]);
          }
          uptr.push_back((int)rs.size());               // one past the level's last parent
          cut_single_pass(len, base, ri);
          sr.n_items = (int)ri.size() - sr.item_first - 1;
          sr.pairs = (int)rs.size() - base;
          sr.level_boxes = count > 0 ? level_boxes(hp.box_level[hp.m2m_parents[first]] + 1) : 0;
          out.push_back(sr);
        }
      };
      add_m2m(m2m_launch, m2m_rot);
      add_m2m(m2m_shared_launch, m2m_shared_rot);
      TRY(upload(rs, &up_rsrc)); TRY(upload(rc, &up_rcls)); TRY(upload(rt, &up_rtgt)); TRY(upload(ri, &up_ritem));
      TRY(upload(uptr, &up_unit_ptr));
      rs.clear(); rc.clear(); rt.clear(); ri.clear();
      for (auto [first, count] : l2l_launch) {
        ShiftRot sr;
        sr.item_first = (int)ri.size();
        const int base = (int)rs.size();
        sr.pair_first = base;
        sr.n_units = count;
        for (int i = first; i < first + count; ++i) {
          const int c = hp.l2l_children[i];
          rs.push_back(hp.box_parent[c]); rc.push_back(down_cls[c]); rt.push_back(c);
        }
        len.assign((size_t)count, 1);
        cut_single_pass(len, base, ri);
        sr.n_items = (int)ri.size() - sr.item_first - 1;
        sr.pairs = count;
        sr.level_boxes = count > 0 ? level_boxes(hp.box_level[hp.l2l_children[first]]) : 0;
        l2l_rot.push_back(sr);
      }
      TRY(upload(rs, &dn_rsrc)); TRY(upload(rc, &dn_rcls)); TRY(upload(rt, &dn_rtgt)); TRY(upload(ri, &dn_ritem));
      const std::shared_ptr<const OrderTables> otp = order_tables().get();      // (built on a thread of their own since plan_create began)
      const OrderTables& ot = *otp;
      for (int q = 1; q <= kRotPmax; ++q) shift_stream_off[q - 1] = ot.shift_off[q - 1];
      TRY(upload(ot.ups, &up_stream)); TRY(upload(ot.dns, &dn_stream));
      {                                                // one pair per wavefront: class tables and the lanes' tables per order
        const int nclass = (int)up_rec_h.size() / 8, cs = sl_class_doubles(pm);
        std::vector<double> cu((size_t)nclass * cs), cd((size_t)nclass * cs);
        for (int c = 0; c < nclass; ++c) {
          sl_class_table(up_rec_h.data() + (size_t)c * 8, pm, kRotM2M, cu.data() + (size_t)c * cs);
          sl_class_table(dn_rec_h.data() + (size_t)c * 8, pm, kRotL2L, cd.data() + (size_t)c * cs);
        }
        TRY(upload(cu, &sl_up_class)); TRY(upload(cd, &sl_dn_class));
        for (int q = 1; q <= kShiftLanesPmax; ++q) { sl_rot_off[q] = ot.sl_rot_off[q]; sl_ax_off[q] = ot.sl_ax_off[q]; }
        TRY(upload(ot.urc, &sl_up_rc)); TRY(upload(ot.urs, &sl_up_rs)); TRY(upload(ot.uxc, &sl_up_xc)); TRY(upload(ot.uxs, &sl_up_xs));
        TRY(upload(ot.drc, &sl_dn_rc)); TRY(upload(ot.drs, &sl_dn_rs)); TRY(upload(ot.dxc, &sl_dn_xc)); TRY(upload(ot.dxs, &sl_dn_xs));
        if (const char* e = std::getenv("FMMBEM_SHIFT_LANES")) shift_lanes = std::atoi(e) != 0;
        if (const char* e = std::getenv("FMMBEM_SHIFT_LANES_MAX")) shift_lanes_max = std::atoi(e);
      }
      if (const char* e = std::getenv("FMMBEM_SHIFT_ROT")) shift_rot = std::atoi(e) != 0;
      if (const char* e = std::getenv("FMMBEM_SHIFT_ROT_MIN")) shift_rot_min = std::atoi(e);
    }
    const cplx *pu = nullptr, *pd = nullptr;
    TRY(upload(up_tab, &pu)); TRY(upload(down_tab, &pd));
    d.up_tab = reinterpret_cast<const double2*>(pu);
    d.down_tab = reinterpret_cast<const double2*>(pd);
    const ShiftOps ops = build_shift_ops(pm, T.A, kEps);
    auto up_op = [&](const VOp& v, ShiftOpDev& o) -> int {
      TRY(upload(v.src, &o.src)); TRY(upload(v.y, &o.y)); TRY(upload(v.real, &o.real));
      TRY(upload(v.npiece, &o.npiece)); TRY(upload(v.piece, &o.piece));
      o.T = v.T; o.V = v.V; o.maxp = v.maxp;
      return FMMBEM_OK;
    };
    up_ops.resize(pm); down_ops.resize(pm);
    for (int p = 1; p <= pm; ++p) { TRY(up_op(ops.up_v[p - 1], up_ops[p - 1])); TRY(up_op(ops.down_v[p - 1], down_ops[p - 1])); }
  }

  mark("shift classes + operators");
  // M2L: targets to run, sources whose Mh is needed, class tables Yh[r,c] = i^{|c|} EPS Y[r,c] / A[r,c]
  {
    std::vector<int> tgt, mh;
    std::vector<uint8_t> is_src(nb, 0);
    for (int b = 0; b < nb; ++b)
      if (hp.has_L[b] && hp.owned_L[b]) tgt.push_back(b);
    for (int s : hp.m2l_src) is_src[s] = 1;
    for (int b = 0; b < nb; ++b) if (is_src[b]) mh.push_back(b);
    d.n_m2l_tgt = (int)tgt.size(); d.n_mh = (int)mh.size();
    TRY(upload(tgt, &d.m2l_tgt)); TRY(upload(mh, &d.mh_box));
    TRY(upload(hp.m2l_ptr, &d.m2l_ptr)); TRY(upload(hp.m2l_src, &d.m2l_src)); TRY(upload(hp.m2l_cls, &d.m2l_cls));
    mark("m2l lists upload");
    n_classes = (int64_t)hp.m2l_class_rep.size() / 2;
    const int R = 2 * pm;
    d.g_max = m2l_entries(pm);
    std::vector<double> gtab((size_t)n_classes * d.g_max);
    std::vector<cplx> ztab((size_t)n_classes * pm);
    parallel_rows(n_classes, 64, [&](int64_t c_begin, int64_t c_end) {
    std::vector<cplx> h;
    for (int64_t c = c_begin; c < c_end; ++c) {
      // translation = c_target - c_source (executor/M2L.hpp:40), rebuilt from the exact integer class
      // vector (half-cell units) so that the table does not depend on which pair was seen first
      // (shards of one operator must produce bit-identical rows).
      double tr[3];
      for (int k = 0; k < 3; ++k) tr[k] = 0.5 * hp.cell[k] * double(hp.m2l_class_vec[3 * c + k]);
      const SphHost sp = cart2sph_host(tr);
      // Yh[r,c] = i^{|c|} EPS Y[r,c] / A[r,c] = Z^c gh[r,c]: tabulate the real part G (harmonics at beta = 0,
      // evalLocal to order 2P) and the phases Z^m = i^m e^{i m beta} separately (kernels_m2l.hip header)
      harmonics(T, false, sp.rho, sp.alpha, 0.0, R, h);
      double* out = gtab.data() + (size_t)c * d.g_max;
      for (int r = 0; r < R; ++r)
        for (int cc = 0; cc <= r; ++cc)
          out[r * (r + 1) / 2 + cc] = h[(size_t)r * (r + 1) / 2 + cc].real() * kEps / T.A[r * r + r + cc];
      for (int m = 0; m < pm; ++m) ztab[(size_t)c * pm + m] = i_pow(m) * std::exp(cplx(0, 1) * double(m * sp.beta));
    }
    });
    const cplx* pz = nullptr;
    TRY(upload(gtab, &d.m2l_g));
    TRY(upload(ztab, &pz));
    d.m2l_z = reinterpret_cast<const double2*>(pz);
    // lane -> output maps and table -> LDS scatter maps of the M2L kernel, every order up to p_max
    const std::shared_ptr<const OrderTables> otp = order_tables().get();
    if (!otp->lanes_ok) return fail(FMMBEM_ERR_INVALID, "internal: M2L lane dealing failed");
    for (int p = 1; p <= kPmax; ++p) d.m2l_scat_off[p - 1] = otp->scat_off[p - 1];
    TRY(upload(otp->lanes, &d.m2l_lane)); TRY(upload(otp->scat, &d.m2l_scat));
  }

  mark("m2l class tables");
  // M2L by rotation (kernels_m2l_rot.hip): the owned pairs in CSR order by target, cut into items; class records; constants
  {
    d.n_rot_items = (int)hp.rot_item_ptr.size() - 1;
    d.n_rot_empty = (int)hp.rot_empty.size();
    if (hp.rot_alias) {
      // one pair list for both M2L kernels; the target of pair i is the box whose CSR range holds i (written on the device)
      d.rot_src = d.m2l_src; d.rot_cls = d.m2l_cls;
      int* tgt = nullptr;
      TRY(alloc(hp.m2l_src.size(), &tgt, false));
      if (nb > 0) hipLaunchKernelGGL(expand_csr_targets_kernel, dim3((nb + 255) / 256), dim3(256), 0, own_stream, d.m2l_ptr, nb, tgt);
      HIP_TRY(hipGetLastError());
      d.rot_tgt = tgt;
    } else {
      TRY(upload(hp.rot_src, &d.rot_src)); TRY(upload(hp.rot_cls, &d.rot_cls)); TRY(upload(hp.rot_tgt, &d.rot_tgt));
    }
    TRY(upload(hp.rot_item_ptr, &d.rot_item_ptr)); TRY(upload(hp.rot_empty, &d.rot_empty));
    d.n_rot_items_long = (int)hp.rot_item_ptr_long.size() - 1;
    TRY(upload(hp.rot_item_ptr_long, &d.rot_item_ptr_long));
    std::vector<double> rec((size_t)n_classes * 8, 0.0);
    for (int64_t c = 0; c < n_classes; ++c) {
      double tr[3];
      for (int k = 0; k < 3; ++k) tr[k] = 0.5 * hp.cell[k] * double(hp.m2l_class_vec[3 * c + k]);
      rot_record(tr, rec.data() + (size_t)c * 8);
    }
    TRY(upload(rec, &d.rot_cls_rec));
    const std::shared_ptr<const OrderTables> otp = order_tables().get();
    for (int p = 1; p <= kRotPmax; ++p) d.rot_tab_off[p - 1] = otp->rot_off[p - 1];
    TRY(upload(otp->rot_all, &d.rot_tab));
  }

  mark("rotation lists");
  shared->on_device = true;
  shared->device = opts.device;
  return to_device_bc_end();
}

// Everything of a plan that depends on the boundary-condition flags: which expansion slots are live, the flags themselves, the
// near-matrix values (the TARGET's flag picks the kernel), the P2M moments (the SOURCE's flag picks them), the expansions and the
// work vectors.  bc_tree: the flags in tree order.  Allocations go to this plan, not to the shared block.
int fmmbem_plan::to_device_bc(const uint8_t* bc_tree) {
  TRY(to_device_bc_begin(bc_tree));
  return to_device_bc_end();
}

// First half: the flags, the near-matrix storage, and the assembly LAUNCHED (own_stream, asynchronous) -- to_device calls this as
// soon as the panels and the near lists are in HBM, so that the 20-50 ms of panel integrals run on the GPU while the host goes on
// tabulating and uploading the far-field operators.
int fmmbem_plan::to_device_bc_begin(const uint8_t* bc_tree) {
  DEVICE_SCOPE(opts.device);
  alloc_list = &allocs;
  d.n_act = 0;
  if (opts.kernel == FMMBEM_KERNEL_STOKES_BEM) {
    // StokesSphericalBEM: M[2][4] per box (kernel/StokesSphericalBEM.hpp:143-153).  The TARGET's flag picks the operator
    // (:377-389): velocity targets read the four potentials of the single layer (slots 0..3), TRACTION targets the seven of
    // the double layer (slots 4..10, kernels_far.hip p2m_apply_kernel<3>); every source feeds the groups that have readers
    d.stokes_velocity_targets = has_bc[0] ? 1 : 0;
    d.stokes_traction_targets = (has_bc[1] && hp.opt.evaluator == 0) ? 1 : 0;
    if (!d.stokes_velocity_targets && !d.stokes_traction_targets) d.stokes_velocity_targets = 1;
    d.nslots = d.stokes_traction_targets ? 11 : 8;
    if (d.stokes_velocity_targets) for (int s = 0; s < 4; ++s) d.act[d.n_act++] = s;
    if (d.stokes_traction_targets) for (int s = 4; s < 11; ++s) d.act[d.n_act++] = s;
  } else {
    d.nslots = 2;
    for (int s = 0; s < 2; ++s) if (has_bc[s]) d.act[d.n_act++] = s;
  }
  {
    uint8_t* dbc = nullptr;
    TRY(alloc((size_t)hp.n, &dbc, false));
    HIP_TRY(hipMemcpy(dbc, bc_tree, (size_t)hp.n, hipMemcpyHostToDevice));
    d.bc = dbc;
  }
  if (near_total_doubles) TRY(alloc((size_t)near_total_doubles, &d.near_val, false));
  if (sym_total_doubles) TRY(alloc((size_t)sym_total_doubles, &d.near_sym, false));
  if (hybrid) {
    HIP_TRY(hipStreamCreateWithFlags(&hyb.recompute, hipStreamNonBlocking));
    HIP_TRY(hipEventCreateWithFlags(&hyb.fork, hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&hyb.join_recompute, hipEventDisableTiming));
    TRY(alloc((size_t)hp.n * d.dof, &d.ys, true));
    if (d.rc_src) TRY(alloc((size_t)hp.n * 4, &d.xt4, true));
  }
  // near-field assembly on the device, in flight from here on (everything it reads is uploaded; to_device_bc_end waits for it)
  HIP_TRY(hipEventCreate(&asm_ev[0])); HIP_TRY(hipEventCreate(&asm_ev[1]));
  HIP_TRY(hipEventRecord(asm_ev[0], own_stream));
  if (opts.sparse_local) {
    if (opts.kernel == FMMBEM_KERNEL_STOKES_BEM) HIP_TRY(launch_near_assemble_stokes(d, own_stream));
    else HIP_TRY(launch_near_assemble(d, own_stream));
  }
  HIP_TRY(hipEventRecord(asm_ev[1], own_stream));
  return FMMBEM_OK;
}

int fmmbem_plan::to_device_bc_end() {
  DEVICE_SCOPE(opts.device);
  alloc_list = &allocs;
  const bool trace = std::getenv("FMMBEM_BUILD_TRACE") != nullptr;
  double t_last = now_ms();
  auto mark = [&](const char* what) {
    if (!trace) return;
    (void)hipDeviceSynchronize();
    const double now = now_ms();
    std::fprintf(stderr, "to_device %-28s %8.2f ms\n", what, now - t_last);
    t_last = now;
  };
  const int dof = d.dof, nb = hp.nboxes;
  TRY(alloc((size_t)nb * d.nslots * d.s_max, &d.M, true));
  TRY(alloc((size_t)nb * d.nslots * d.s_max, &d.L, true));
  TRY(alloc((size_t)nb * d.nslots * d.s_max, &d.Mh, true));
  TRY(alloc((size_t)hp.n * dof, &d.xt, true));
  TRY(alloc((size_t)hp.n * dof, &d.yt, true));
  TRY(alloc((size_t)hp.n * dof, &stage_x, true));
  TRY(alloc((size_t)hp.n * dof, &stage_y, true));

  // The zero-fills above ran on the NULL stream, which the plan's non-blocking streams do not wait for:
  // drain it before anything else touches those buffers (a late memset of x_tree / staging vectors would
  // otherwise race with the first execute).
  HIP_TRY(hipDeviceSynchronize());

  mark("vectors");
  // P2M as a stored operator: every panel's moments about its leaf centre, at p_max (FMMBEM_P2M_TABLE=0, or more than
  // 16 GB of them: the recurrences are run every matvec instead)
  {
    const char* e = std::getenv("FMMBEM_P2M_TABLE");
    const size_t ntab = opts.kernel == FMMBEM_KERNEL_STOKES_BEM ? 4 : 1;
    // the panels of the leaves this plan runs P2M on: all of them, or -- upward pass sharded by owner -- the shard's own
    // rows only (0.9 GB at N = 1M, p_max = 10 for the whole mesh: one eighth of that per GPU on an 8-GPU node)
    int64_t r0 = hp.n, r1 = 0;
    for (int b : hp.p2m_leaves) { r0 = std::min<int64_t>(r0, hp.box_body_begin[b]); r1 = std::max<int64_t>(r1, hp.box_body_end[b]); }
    if (r1 < r0) r0 = r1 = 0;
    d.p2m_tab_row0 = r0;
    const size_t count = (size_t)(r1 - r0) * ntab * d.p2m_stride;
    const bool stokes = opts.kernel == FMMBEM_KERNEL_STOKES_BEM;
    const size_t count_g = (size_t)(r1 - r0) * 3 * d.p2m_stride;
    if (stokes && d.stokes_traction_targets && d.n_p2m > 0 && !(e && std::atoi(e) == 0) && count_g * sizeof(double2) <= ((size_t)16 << 30)) {
      // the double layer's gradient records; without them (switch, or more than 16 GB) its P2M runs the recurrences per matvec
      double2* tab = nullptr;
      TRY(alloc(count_g, &tab, true));
      HIP_TRY(hipDeviceSynchronize());
      HIP_TRY(launch_p2m_table_grad(d, tab, own_stream));
      HIP_TRY(hipStreamSynchronize(own_stream));
      d.p2m_tab_g = tab;
    }
    const bool want = !stokes || d.stokes_velocity_targets;
    if (want && !(e && std::atoi(e) == 0) && hp.opt.evaluator == 0 && d.n_p2m > 0 && count * sizeof(double2) <= ((size_t)16 << 30)) {
      double2* tab = nullptr;
      TRY(alloc(count, &tab, true));
      HIP_TRY(hipDeviceSynchronize());                 // the zero-fill ran on the NULL stream
      HIP_TRY(launch_p2m_table(d, tab, own_stream));
      HIP_TRY(hipStreamSynchronize(own_stream));
      d.p2m_tab = tab;
    }
  }
  mark("p2m table");
  // the near-field assembly launched by to_device_bc_begin: its device time
  HIP_TRY(hipStreamSynchronize(own_stream));
  {
    float ms = 0;
    HIP_TRY(hipEventElapsedTime(&ms, asm_ev[0], asm_ev[1]));
    build_assemble_ms = ms;
    (void)hipEventDestroy(asm_ev[0]); (void)hipEventDestroy(asm_ev[1]);
    asm_ev[0] = asm_ev[1] = nullptr;
  }
  const double t0 = now_ms();
  if ((!opts.sparse_local || hybrid) && hp.row_end > hp.row_begin) TRY(build_side_lists());
  build_assemble_ms += now_ms() - t0;
  mark("near assembly (waited for)");
  {                                                    // the plan itself, readable from the device
    void* pd = nullptr;
    HIP_TRY(hipMalloc(&pd, sizeof(DevicePlan)));
    allocs.push_back(pd);
    HIP_TRY(hipMemcpy(pd, &d, sizeof(DevicePlan), hipMemcpyHostToDevice));
    d_dev = static_cast<const DevicePlan*>(pd);
  }
  return FMMBEM_OK;
}

// matrix-free plans (every owned row) and hybrid plans (the rows of the recomputed leaves; the others list nothing):
int fmmbem_plan::build_side_lists() {
  const int dof = d.dof;
  {
    // nothing is assembled for these rows, but the pairs of the near regimes -- the expensive 4.5 %, the same numbers every matvec
    // -- are listed per target row and evaluated once (kernels_near.hip, near_matfree third form): count, scan, fill, evaluate
    int* d_cnt = nullptr;
    TRY(alloc((size_t)hp.n, &d_cnt, true));
    HIP_TRY(hipDeviceSynchronize());
    HIP_TRY(launch_mf_side(d, 0, d_cnt, nullptr, nullptr, nullptr, nullptr, 0, own_stream));
    HIP_TRY(hipStreamSynchronize(own_stream));
    std::vector<int> cnt((size_t)hp.n);
    HIP_TRY(hipMemcpy(cnt.data(), d_cnt, sizeof(int) * cnt.size(), hipMemcpyDeviceToHost));
    std::vector<int64_t> ptr((size_t)hp.n + 1, 0);
    for (int64_t i = 0; i < hp.n; ++i) ptr[(size_t)i + 1] = ptr[(size_t)i] + cnt[(size_t)i];
    const int64_t nside = ptr.back();
    std::vector<int> row((size_t)nside);
    for (int64_t i = 0; i < hp.n; ++i)
      for (int64_t k = ptr[(size_t)i]; k < ptr[(size_t)i + 1]; ++k) row[(size_t)k] = (int)i;
    const int* d_row = nullptr;
    int* d_col = nullptr;
    double* d_val = nullptr;
    TRY(upload(ptr, &d.side_ptr)); TRY(upload(row, &d_row));
    TRY(alloc((size_t)nside, &d_col, false)); TRY(alloc((size_t)nside * (dof == 3 ? 9 : 1), &d_val, false));
    HIP_TRY(launch_mf_side(d, 1, nullptr, d.side_ptr, d_col, nullptr, nullptr, nside, own_stream));
    HIP_TRY(launch_mf_side(d, 2, nullptr, nullptr, d_col, d_row, d_val, nside, own_stream));
    HIP_TRY(hipStreamSynchronize(own_stream));
    d.side_col = d_col; d.side_val = d_val;
    near_side_entries = nside;
    if (hybrid) {
      // work items of near_side_items (kernels_near.hip): runs of consecutive rows that hold <= 256 listed entries together (a workgroup takes the
      // entries one per thread, then a thread per row adds the row's products in entry order); a row of more than 256 is an item
      // of its own, taken 256 at a time
      std::vector<int4> sitems;
      int64_t i = hp.row_begin;
      while (i < hp.row_end) {
        if (cnt[(size_t)i] == 0) { ++i; continue; }
        int64_t j = i + 1;
        while (j < hp.row_end && cnt[(size_t)j] > 0 && ptr[(size_t)j + 1] - ptr[(size_t)i] <= 256 && j - i < 256) ++j;
        sitems.push_back(make_int4((int)ptr[(size_t)i], (int)ptr[(size_t)j], (int)i, (int)j));
        i = j;
      }
      d.side_nitems = (int)sitems.size();
      TRY(upload(sitems, &d.side_items));
    }
    for (void* tmp : {(void*)d_cnt, (void*)const_cast<int*>(d_row)}) {        // creation-time scratch
      (void)hipFree(tmp);
      allocs.erase(std::find(allocs.begin(), allocs.end(), tmp));
    }
  }
  return FMMBEM_OK;
}

// One level of the upward / downward pass, p <= 12: the one-pair-per-WAVEFRONT kernel (kernels_shift.hip) for up to
// shift_lanes_max pairs of this plan, the one-pair-per-LANE rotation kernel (kernels_m2l_rot.hip, ~20 us a pass whatever it holds)
// above -- the two give the same bits, so every plan and every shard chooses by its own share.  p > 12 (and FMMBEM_SHIFT_ROT=0):
// the sparse-operator kernels of kernels_far.hip, which round differently -- there the choice is the same for all shards.
int fmmbem_plan::m2m_pass(int p, bool shared, hipStream_t s) {
  const auto& launches = shared ? m2m_shared_launch : m2m_launch;
  const auto& rots = shared ? m2m_shared_rot : m2m_rot;
  for (size_t i = 0; i < launches.size(); ++i) {
    const auto [first, count] = launches[i];
    const ShiftRot& sr = rots[i];
    if (shift_rot && shift_lanes && shift_lanes_supported(p) && (int64_t)sr.pairs * d.n_act <= lanes_max(p, true)) {
      ShiftLaneWork lw;
      lw.src = up_rsrc; lw.cls = up_rcls; lw.tgt = up_rtgt; lw.unit_ptr = up_unit_ptr + sr.unit_first; lw.n_units = sr.n_units;
      lw.class_tab = sl_up_class; lw.class_stride = sl_class_doubles(hp.opt.p_max); lw.p_max = hp.opt.p_max;
      lw.rot_c = sl_up_rc + sl_rot_off[p]; lw.rot_s = sl_up_rs + sl_rot_off[p]; lw.ax_c = sl_up_xc + sl_ax_off[p]; lw.ax_s = sl_up_xs + sl_ax_off[p];
      HIP_TRY(launch_m2m_lanes(d, lw, p, s));
    } else if (shift_rot && shift_rot_supported(p) && (shift_lanes || sr.level_boxes >= shift_rot_min)) {   // beside the wavefront kernel never the sparse operators: they round differently
      RotWork w;
      w.src = up_rsrc; w.cls = up_rcls; w.tgt = up_rtgt; w.rec = up_rec;
      w.item_ptr = up_ritem + sr.item_first; w.n_items = sr.n_items;
      w.stream = up_stream + shift_stream_off[p - 1];
      HIP_TRY(launch_m2m_rot(d, w, p, s));
    } else HIP_TRY(launch_m2m_level(d, up_ops[p - 1], p, first, count, s));
  }
  return FMMBEM_OK;
}
int fmmbem_plan::l2l_pass(int p, hipStream_t s) {
  for (size_t i = 0; i < l2l_launch.size(); ++i) {
    const auto [first, count] = l2l_launch[i];
    const ShiftRot& sr = l2l_rot[i];
    if (shift_rot && shift_lanes && shift_lanes_supported(p) && (int64_t)sr.pairs * d.n_act <= lanes_max(p, false)) {
      ShiftLaneWork lw;
      lw.src = dn_rsrc + sr.pair_first; lw.cls = dn_rcls + sr.pair_first; lw.tgt = dn_rtgt + sr.pair_first; lw.n_units = sr.n_units;
      lw.class_tab = sl_dn_class; lw.class_stride = sl_class_doubles(hp.opt.p_max); lw.p_max = hp.opt.p_max;
      lw.rot_c = sl_dn_rc + sl_rot_off[p]; lw.rot_s = sl_dn_rs + sl_rot_off[p]; lw.ax_c = sl_dn_xc + sl_ax_off[p]; lw.ax_s = sl_dn_xs + sl_ax_off[p];
      HIP_TRY(launch_l2l_lanes(d, lw, p, s));
    } else if (shift_rot && shift_rot_supported(p) && (shift_lanes || sr.level_boxes >= shift_rot_min)) {   // beside the wavefront kernel never the sparse operators: they round differently
      RotWork w;
      w.src = dn_rsrc; w.cls = dn_rcls; w.tgt = dn_rtgt; w.rec = dn_rec;
      w.item_ptr = dn_ritem + sr.item_first; w.n_items = sr.n_items;
      w.stream = dn_stream + shift_stream_off[p - 1];
      HIP_TRY(launch_l2l_rot(d, w, p, s));
    } else HIP_TRY(launch_l2l_level(d, down_ops[p - 1], p, first, count, s));
  }
  return FMMBEM_OK;
}

int fmmbem_plan::run(int p, const double* d_x, double* d_y, hipStream_t s, bool near_only, int phase, double* xbuf) {
  if (!on_device) return fail(FMMBEM_ERR_NO_DEVICE, "plan was built host-only; there is no CPU execution path");
  if (p < 1 || p > hp.opt.p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  if ((phase < 2 && !d_x) || (phase != 1 && !d_y)) return fail(FMMBEM_ERR_INVALID, "null vector");
  if (split_upward && phase == 0 && !near_only)
    return fail(FMMBEM_ERR_UNSUPPORTED, "plan shards the upward pass: use fmmbem_plan_upward_device / _downward_device");
  if (phase != 0 && (!split_upward || (!xbuf && phase != 3))) return fail(FMMBEM_ERR_INVALID, "split execute needs shard_upward and an exchange buffer");
  DEVICE_SCOPE(opts.device);
  const int tm = timing;
  const int64_t ring = ev_count % kRing;
  hipEvent_t* set = tm ? &ev[(size_t)ring * 2 * kStages] : nullptr;
  unsigned mask = phase >= 2 ? pending_mask : 0;
  // stage i runs on stream st: begin/end events bracket exactly that kernel (or level sequence).  An event record is a
  // packet of its own between two kernels (~5 us each on this part: a fully instrumented matvec is 85 us longer than a bare
  // one), so a throughput measurement brackets the one kernel it needs (mode 2: the near field, stage 1) and nothing else.
  auto rec = [&](int i) { return tm == 1 || (tm == 2 && i == 1); };
  auto begin = [&](int i, hipStream_t st) -> hipError_t { return rec(i) ? hipEventRecord(set[2 * i], st) : hipSuccess; };
  auto end = [&](int i, hipStream_t st) -> hipError_t {
    if (!rec(i)) return hipSuccess;
    mask |= 1u << i;
    return hipEventRecord(set[2 * i + 1], st);
  };
  // Stage order of the reference (EvalInteractionLazySparse.hpp:120-168): near-field SpMV, then P2M, M2M, M2L, L2L, L2P, all on
  // the caller's stream.  (The near field on a second stream beside the far field was measured in rounds 1, 2 and 3 in five
  // variants and is slower every time -- beside the saturated memory system of the near field every dependent access of the
  // latency-bound far kernels takes 10-70x longer; profiles/r03d_overlap_near_far.txt, DESIGN.md section 9 -- and is gone.)
  // a region of the chain, launch by launch or as a graph (see GraphEntry)
  const bool graph_ok = use_graphs && tm == 0;
  auto graphed = [&](int region, const void* buf, auto&& body) -> int {
    if (!graph_ok) return body(s);
    GraphEntry* ge = nullptr;
    for (auto& g : graphs) if (g.region == region && g.p == p && g.buf == buf) ge = &g;
    if (!ge) { graphs.push_back(GraphEntry{region, p, buf, 0, nullptr}); ge = &graphs.back(); }
    if (++ge->runs == 1) return body(s);
    if (!ge->exec) {
      // one capture at a time in the process: the shards of a device list are issued by threads of their own, and a capture on one
      // thread made another thread's hipStreamWaitEvent fail ("dependency created on uncaptured work in another stream")
      static std::mutex capture_mu;
      std::lock_guard<std::mutex> capture_lock(capture_mu);
      HIP_TRY(hipStreamBeginCapture(own_stream, hipStreamCaptureModeThreadLocal));
      const int rc = body(own_stream);
      hipGraph_t g = nullptr;
      const hipError_t e = hipStreamEndCapture(own_stream, &g);
      if (rc != FMMBEM_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
      HIP_TRY(e);
      const hipError_t ei = hipGraphInstantiate(&ge->exec, g, nullptr, nullptr, 0);
      (void)hipGraphDestroy(g);
      HIP_TRY(ei);
    }
    HIP_TRY(hipGraphLaunch(ge->exec, s));
    return FMMBEM_OK;
  };
  if (phase < 2) {
    HIP_TRY(begin(0, s));
    HIP_TRY(launch_gather_x(d, d_x, s));
    HIP_TRY(end(0, s));
  }
  if (phase == 1) {                                    // upward half: my leaves, my boxes, pack what the others need
    TRY(graphed(1, xbuf, [&](hipStream_t s) -> int {
      HIP_TRY(begin(3, s));
      if (d.kernel == FMMBEM_KERNEL_STOKES_BEM) HIP_TRY(launch_p2m_stokes(d, p, s)); else HIP_TRY(launch_p2m(d, p, s));
      HIP_TRY(end(3, s));
      HIP_TRY(begin(4, s));
      TRY(m2m_pass(p, false, s));
      HIP_TRY(launch_xch_pack(d, p, reinterpret_cast<double2*>(xbuf), s));
      HIP_TRY(end(4, s));
      return FMMBEM_OK;
    }));
    pending_mask = mask;
    pending_near = false;
    return FMMBEM_OK;
  }
  auto near_field = [&](hipStream_t ns) -> int {
    HIP_TRY(begin(1, ns));
    if (hybrid) HIP_TRY(launch_near_hybrid(d, ns, hyb));
    else if (opts.sparse_local) HIP_TRY(launch_near_spmv(d, ns)); else HIP_TRY(launch_near_matfree(d, ns));
    HIP_TRY(end(1, ns));
    return FMMBEM_OK;
  };
  // the result leaves the plan once, at the very end: the owned rows of y_tree (near + far field) scattered to the caller's
  // panel order (zeros elsewhere when the plan is a shard), or -- result_slices -- copied as they are, tree order, to the
  // head of d_y, for the caller's all-gather (fmmbem_plan_assemble_slices_device puts the gathered slices in panel order)
  auto deliver = [&](hipStream_t ns) -> int {
    HIP_TRY(begin(2, ns));
    if (result_slices) {
      HIP_TRY(hipMemcpyAsync(d_y, d.yt + d.row_begin * d.dof, sizeof(double) * (size_t)(d.row_end - d.row_begin) * d.dof,
                             hipMemcpyDeviceToDevice, ns));
    } else {
      if (hp.opt.shard_world > 1) HIP_TRY(hipMemsetAsync(d_y, 0, sizeof(double) * (size_t)hp.n * d.dof, ns));
      HIP_TRY(launch_scatter_y(d, d_y, ns));
    }
    HIP_TRY(end(2, ns));
    return FMMBEM_OK;
  };
  if (phase == 3) {                                    // the near field of a split execute, while the caller's all-gather is in flight
    TRY(near_field(s));
    pending_mask = mask;
    pending_near = true;
    return FMMBEM_OK;
  }
  const bool near_here = !(phase == 2 && pending_near);
  pending_near = false;
  TRY(graphed((phase == 2 ? 2 : 0) + (near_here ? 0 : 4) + (near_only ? 8 : 0), xbuf, [&](hipStream_t s) -> int {
  if (near_here) TRY(near_field(s));
  if (!near_only) {
    if (phase == 0) {
      HIP_TRY(begin(3, s));
      if (d.kernel == FMMBEM_KERNEL_STOKES_BEM) HIP_TRY(launch_p2m_stokes(d, p, s)); else HIP_TRY(launch_p2m(d, p, s));
      HIP_TRY(end(3, s));
      HIP_TRY(begin(4, s));
      TRY(m2m_pass(p, false, s));
      HIP_TRY(end(4, s));
    }
    HIP_TRY(begin(5, s));
    if (phase == 2) {                                  // the other shards' multipoles, then the boxes spanning shards
      HIP_TRY(launch_xch_unpack(d, p, reinterpret_cast<const double2*>(xbuf), s));
      TRY(m2m_pass(p, true, s));
    }
    const bool rot = use_rot(p);
    if (!rot) HIP_TRY(launch_mh_prep(d, p, s));      // the rotation kernel reads M itself
    HIP_TRY(end(5, s));
    HIP_TRY(begin(6, s));
    if (rot) HIP_TRY(launch_m2l_rot(d, d_dev, p, s));
    else HIP_TRY(launch_m2l(d, d_dev, p, s));
    HIP_TRY(end(6, s));
    HIP_TRY(begin(7, s));
    TRY(l2l_pass(p, s));
    HIP_TRY(end(7, s));
    HIP_TRY(begin(8, s));
    if (d.kernel == FMMBEM_KERNEL_STOKES_BEM) HIP_TRY(launch_l2p_stokes(d, p, d.yt, s));
    else HIP_TRY(launch_l2p(d, p, d.yt, s));
    HIP_TRY(end(8, s));
  }
  return FMMBEM_OK;
  }));
  TRY(deliver(s));
  last_p = p;
  if (tm) { ev_mask[ring] = mask; ++ev_count; }
  return FMMBEM_OK;
}

// ============================================ C ABI ============================================
extern "C" {

void fmmbem_options_default(fmmbem_options* o) {
  if (!o) return;
  std::memset(o, 0, sizeof(*o));
  o->kernel = FMMBEM_KERNEL_LAPLACE_BEM;
  o->p_max = 10;
  o->quad_k = 3;
  o->theta = 0.5;
  o->ncrit = 64;
  o->sparse_local = 1;
  o->shard_world = 1;
  o->quad_k_fine = 25;          // StokesSphericalBEM ctor default (kernel/StokesSphericalBEM.hpp:131)
  o->mu = 1e-3;
  o->near_stream_fraction = 1.0;
}

// ---- plans that share a geometry -----------------------------------------------------------------------------
// Two independent 64-bit hashes of the vertex bytes (the panels in the caller's order), a few threads: 75 MB at N = 1M in ~5 ms
static void fingerprint_vertices(const double* v, size_t n_panels, uint64_t out[2]) {
  const size_t words = n_panels * 9;
  const int nt = words < (1u << 20) ? 1 : (int)std::min<unsigned>(16, std::max(1u, std::thread::hardware_concurrency()));
  std::vector<uint64_t> part((size_t)nt * 2);
  auto work = [&](int t) {
    const size_t w0 = words * t / nt, w1 = words * (t + 1) / nt;
    uint64_t a = 0x9E3779B97F4A7C15ull + t, b = 0xC2B2AE3D27D4EB4Full ^ (uint64_t)t;
    for (size_t i = w0; i < w1; ++i) {
      uint64_t x;
      std::memcpy(&x, v + i, 8);
      a = (a ^ x) * 0x100000001B3ull; a ^= a >> 29;                       // FNV-style multiply-xor with a fold
      b += x * 0xFF51AFD7ED558CCDull; b = (b << 27) | (b >> 37); b *= 0xC4CEB9FE1A85EC53ull;
    }
    part[2 * t] = a; part[2 * t + 1] = b;
  };
  if (nt == 1) work(0);
  else {
    host_parallel(nt, [&](int t) { work(t); });
  }
  uint64_t a = n_panels, b = ~(uint64_t)n_panels;
  for (int t = 0; t < nt; ++t) { a = (a ^ part[2 * t]) * 0x100000001B3ull; b = (b + part[2 * t + 1]) * 0xC4CEB9FE1A85EC53ull; b ^= b >> 31; }
  out[0] = a; out[1] = b | 1;                                             // never 0 0
}

// the options that decide the geometry's share of a plan: everything but the flags (and mu, which only scales near values)
static bool same_geometry_options(const fmmbem_options& a, const fmmbem_options& b) {
  return a.kernel == b.kernel && a.p_max == b.p_max && a.quad_k == b.quad_k && a.theta == b.theta && a.ncrit == b.ncrit &&
         a.sparse_local == b.sparse_local && a.host_only == b.host_only && a.device == b.device && a.shard_rank == b.shard_rank &&
         a.shard_world == b.shard_world && a.quad_k_fine == b.quad_k_fine && a.evaluator == b.evaluator && a.mu == b.mu &&
         a.shard_upward == b.shard_upward && a.l2l_rule == b.l2l_rule && a.near_stream_fraction == b.near_stream_fraction;
}

// live plans that hold a geometry, most recent first (a handful: operator, right-hand side, preconditioner plans).  ONE mutex: a
// creation that finds its geometry here copies the holder's state under it (microseconds), fmmbem_plan_destroy takes it before it
// deletes -- so a plan is never read while it goes away -- and the long part of the creation runs outside.
static std::mutex g_cache_mu;
static std::vector<fmmbem_plan*> g_cache;

static void geometry_cache_add(fmmbem_plan* p) {
  std::lock_guard<std::mutex> lock(g_cache_mu);
  g_cache.insert(g_cache.begin(), p);
  if (g_cache.size() > 32) g_cache.resize(32);
}
static void geometry_cache_remove(fmmbem_plan* p) {
  std::lock_guard<std::mutex> lock(g_cache_mu);
  g_cache.erase(std::remove(g_cache.begin(), g_cache.end(), p), g_cache.end());
}

// A plan of `base`'s panels with other boundary-condition flags: shares base's PlanShared, builds what the flags decide.
int fmmbem_plan::like(const fmmbem_plan& base, const uint8_t* bc, fmmbem_plan** out) {
  *out = nullptr;
  std::unique_ptr<fmmbem_plan> pl(new (std::nothrow) fmmbem_plan(base));       // memberwise: d, the launch lists, the table pointers
  if (!pl) return fail(FMMBEM_ERR_ALLOC, "plan");
  return like_finish(std::move(pl), bc, out);
}

// `pl`: a memberwise copy of a plan that holds the geometry (base may be gone by now: the shared block is reference counted)
int fmmbem_plan::like_finish(std::unique_ptr<fmmbem_plan> pl, const uint8_t* bc, fmmbem_plan** out) {
  *out = nullptr;
  const double t0 = now_ms();
  // ... and now everything that must NOT be shared with base (the copy constructor copied the handles): fresh or empty
  pl->allocs.clear();
  pl->alloc_list = &pl->allocs;
  pl->ev.assign(pl->ev.size(), nullptr);
  pl->graphs.clear();
  pl->own_stream = nullptr; pl->hyb = HybridStreams{}; pl->d.ys = nullptr; pl->d.xt4 = nullptr;
  pl->asm_ev[0] = pl->asm_ev[1] = nullptr;
  pl->d_dev = nullptr; pl->stage_x = pl->stage_y = nullptr; pl->solver_ws = nullptr; pl->d_cut = nullptr;
  pl->multi.reset();
  pl->result_slices = false; pl->pending_mask = 0; pl->pending_near = false;
  pl->timing = 0; pl->last_p = 0; pl->ev_count = 0;
  for (auto& m : pl->ev_mask) m = 0;
  pl->near_side_entries = 0;
  pl->d.side_ptr = nullptr; pl->d.side_col = nullptr; pl->d.side_val = nullptr;
  pl->d.p2m_tab = nullptr; pl->d.p2m_tab_g = nullptr;
  pl->d.near_val = nullptr; pl->d.near_sym = nullptr;
  pl->build_host_ms = 0;
  DEVICE_SCOPE(pl->opts.device);
  HIP_TRY(hipStreamCreateWithFlags(&pl->own_stream, hipStreamNonBlocking));
  for (auto& e : pl->ev) HIP_TRY(hipEventCreate(&e));
  const HostPlan& h = pl->hp;
  std::vector<uint8_t> bc_tree((size_t)h.n, 0);
  pl->has_bc[0] = pl->has_bc[1] = false;
  for (int64_t i = 0; i < h.n; ++i) {
    const uint8_t f = bc ? (bc[h.perm[(size_t)i]] ? 1 : 0) : 0;
    bc_tree[(size_t)i] = f;
    pl->has_bc[f] = true;
  }
  const int rc = pl->to_device_bc(bc_tree.data());
  if (rc != FMMBEM_OK) return rc;
  pl->build_host_ms = now_ms() - t0 - pl->build_assemble_ms;
  if (!(std::getenv("FMMBEM_PLAN_SHARE") && std::atoi(std::getenv("FMMBEM_PLAN_SHARE")) == 0)) geometry_cache_add(pl.get());   // any holder of the geometry can stand in for the first
  *out = pl.release();
  return FMMBEM_OK;
}


// ---- one plan over several devices of ONE process (fmmbem_options.n_devices > 1; SURVEY.md section 8b "device list") ------
// The reference's drivers build one plan in one process (examples/LaplaceBEM.cpp:209; FMM_plan.hpp:34-43).  This is that plan with
// its target leaves sharded over the devices of the list: the handle owns one SHARD plan per device (the same shards the
// one-process-per-GPU form drives through torch.distributed, distributed.py) and moves the data between them itself --
//   x (on the first device)  -> a copy per device                                 hipMemcpyPeerAsync, one xGMI link each
//   upward pass by the boxes' owners -> the multipoles each shard's lists read     peer copies of fmmbem_plan_exchange_counts' segments
//   downward pass + near field of every shard -> its tree-order slice of y        -> peer copies into ONE buffer on the first device
//   slices -> y in panel order (fmmbem_plan_assemble_slices_device)
// every step on the shard's own stream, ordered by events: the host thread only enqueues.  xGMI is point to point, so plain peer
// copies ARE the collective here (one link per peer, no ring); the same bits as a single plan (shards sum bitwise).
// Not measured on more than one GPU (none has been available to this project); devices = {0, 0, ...} runs the whole path on one.
struct ShardDeleter { void operator()(fmmbem_plan* p) const { fmmbem_plan_destroy(p); } };   // through the C entry point: the shards are known to the geometry cache
struct MultiDevice {
  std::vector<std::unique_ptr<fmmbem_plan, ShardDeleter>> shards;
  std::vector<int> dev;
  std::vector<double*> x, slice;                     // per shard, on its device: the replica of x, the result slice
  double* gathered = nullptr;                        // on dev[0]: world * chunk doubles
  size_t chunk = 0;
  std::vector<int64_t> cut;
  hipEvent_t ev_x = nullptr;
  std::vector<hipEvent_t> ev_up, ev_done;
  bool split = false;
  struct Xch { std::vector<double*> send, recv; std::vector<std::vector<int64_t>> sc, rc; };
  std::map<int, Xch> xch;                            // per order p
  int world() const { return (int)shards.size(); }
  ~MultiDevice() {
    int prev = 0;
    (void)hipGetDevice(&prev);
    for (int r = 0; r < world(); ++r) {
      (void)hipSetDevice(dev[r]);
      if (r < (int)x.size() && x[r]) (void)hipFree(x[r]);
      if (r < (int)slice.size() && slice[r]) (void)hipFree(slice[r]);
      for (auto& kv : xch) { if (kv.second.send[r]) (void)hipFree(kv.second.send[r]); if (kv.second.recv[r]) (void)hipFree(kv.second.recv[r]); }
      if (r < (int)ev_up.size() && ev_up[r]) (void)hipEventDestroy(ev_up[r]);
      if (r < (int)ev_done.size() && ev_done[r]) (void)hipEventDestroy(ev_done[r]);
    }
    if (!dev.empty()) (void)hipSetDevice(dev[0]);
    if (gathered) (void)hipFree(gathered);
    if (ev_x) (void)hipEventDestroy(ev_x);
    (void)hipSetDevice(prev);
    shards.clear();
  }
};

static int multi_finish(fmmbem_plan* h, std::shared_ptr<MultiDevice> m) {
  const int W = m->world();
  const fmmbem_plan& s0 = *m->shards[0];
  const size_t nd = (size_t)s0.hp.n * s0.d.dof;
  m->cut.resize((size_t)W + 1);
  TRY(fmmbem_plan_shard_rows(m->shards[0].get(), m->cut.data()));
  for (int r = 0; r < W; ++r) m->chunk = std::max(m->chunk, (size_t)(m->cut[r + 1] - m->cut[r]) * s0.d.dof);
  m->chunk = std::max<size_t>(m->chunk, 1);
  m->split = s0.split_upward;
  m->x.assign(W, nullptr); m->slice.assign(W, nullptr); m->ev_up.assign(W, nullptr); m->ev_done.assign(W, nullptr);
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
  for (int r = 0; r < W; ++r) {
    HIP_TRY(hipSetDevice(m->dev[r]));
    for (int q = 0; q < W; ++q)                        // direct peer copies where the hardware has them (an error here only means "already on" or "not available")
      if (m->dev[q] != m->dev[r]) { if (hipDeviceEnablePeerAccess(m->dev[q], 0) != hipSuccess) (void)hipGetLastError(); }
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->x[r]), sizeof(double) * nd));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->slice[r]), sizeof(double) * m->chunk));
    HIP_TRY(hipEventCreateWithFlags(&m->ev_up[r], hipEventDisableTiming));
    HIP_TRY(hipEventCreateWithFlags(&m->ev_done[r], hipEventDisableTiming));
    m->shards[r]->result_slices = true;
    m->shards[r]->use_graphs = !(std::getenv("FMMBEM_GRAPH") && std::atoi(std::getenv("FMMBEM_GRAPH")) == 0);   // a shard's chain is short: the host must stay ahead of W of them
  }
  HIP_TRY(hipSetDevice(m->dev[0]));
  HIP_TRY(hipMalloc(reinterpret_cast<void**>(&m->gathered), sizeof(double) * m->chunk * (size_t)W));
  HIP_TRY(hipEventCreateWithFlags(&m->ev_x, hipEventDisableTiming));
  // the handle itself: what the host-pointer execute and the solver need on the first device
  h->opts.device = m->dev[0];
  h->on_device = true;
  h->d.dof = s0.d.dof; h->d.n = s0.d.n;
  h->has_bc[0] = s0.has_bc[0]; h->has_bc[1] = s0.has_bc[1];
  HIP_TRY(hipStreamCreateWithFlags(&h->own_stream, hipStreamNonBlocking));
  TRY(h->alloc(nd, &h->stage_x, true));
  TRY(h->alloc(nd, &h->stage_y, true));
  HIP_TRY(hipDeviceSynchronize());
  h->multi = std::move(m);
  return FMMBEM_OK;
}

static int multi_create(const fmmbem_options* opts, const std::vector<int>& devices, size_t n_panels, const double* vertices,
                        const uint8_t* bc, fmmbem_plan** out) {
  const int W = (int)devices.size();
  if (W > 8) return fail(FMMBEM_ERR_UNSUPPORTED, "at most 8 devices per plan");
  if (opts->shard_world > 1) return fail(FMMBEM_ERR_INVALID, "a plan over several devices is the whole operator: shard_world must be 1");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) return fail(FMMBEM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU execution path)");
  for (int d_ : devices) if (d_ < 0 || d_ >= ndev) return fail(FMMBEM_ERR_INVALID, "device ordinal out of range in the device list");
  auto m = std::make_shared<MultiDevice>();
  m->dev = devices;
  m->shards.resize(W);
  // the shards are built side by side: each is a plan_create of its own (host lists of the same tree, then its device's share)
  std::vector<int> rc(W, FMMBEM_OK);
  std::vector<std::string> err(W);
  std::vector<std::thread> pool;
  for (int r = 0; r < W; ++r)
    pool.emplace_back([&, r] {
      fmmbem_options o = *opts;
      o.n_devices = 0; o.device = devices[r]; o.shard_rank = r; o.shard_world = W;
      o.shard_upward = opts->shard_upward == 0 ? 0 : 2;            // 0: every device repeats the upward pass; else: owners + selective exchange
      if (std::getenv("FMMBEM_MULTI_UPWARD")) o.shard_upward = std::atoi(std::getenv("FMMBEM_MULTI_UPWARD")) ? 2 : 0;
      fmmbem_plan* p = nullptr;
      rc[r] = fmmbem_plan_create(&o, n_panels, vertices, bc, &p);
      if (rc[r] != FMMBEM_OK) err[r] = g_last_error;
      m->shards[r].reset(p);
    });
  for (auto& th : pool) th.join();
  for (int r = 0; r < W; ++r) if (rc[r] != FMMBEM_OK) return fail(rc[r], "shard " + std::to_string(r) + ": " + err[r]);
  std::unique_ptr<fmmbem_plan> h(new (std::nothrow) fmmbem_plan(m->shards[0]->shared));
  if (!h) return fail(FMMBEM_ERR_ALLOC, "plan");
  h->opts = *opts;
  TRY(multi_finish(h.get(), m));
  *out = h.release();
  return FMMBEM_OK;
}

// a multi-device plan of `base`'s panels with other flags: shard by shard (each shares its base shard's geometry)
static int multi_like(const fmmbem_plan& base, const uint8_t* bc, fmmbem_plan** out) {
  auto m = std::make_shared<MultiDevice>();
  m->dev = base.multi->dev;
  const int W = base.multi->world();
  m->shards.resize(W);
  for (int r = 0; r < W; ++r) {
    fmmbem_plan* p = nullptr;
    TRY(fmmbem_plan::like(*base.multi->shards[r], bc, &p));
    m->shards[r].reset(p);
  }
  std::unique_ptr<fmmbem_plan> h(new (std::nothrow) fmmbem_plan(m->shards[0]->shared));
  if (!h) return fail(FMMBEM_ERR_ALLOC, "plan");
  h->opts = base.opts;
  TRY(multi_finish(h.get(), m));
  *out = h.release();
  return FMMBEM_OK;
}

static int multi_xch(fmmbem_plan* h, int p, MultiDevice::Xch** out) {
  MultiDevice& m = *h->multi;
  auto it = m.xch.find(p);
  if (it != m.xch.end()) { *out = &it->second; return FMMBEM_OK; }
  const int W = m.world();
  MultiDevice::Xch x;
  x.send.assign(W, nullptr); x.recv.assign(W, nullptr); x.sc.resize(W); x.rc.resize(W);
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
  for (int r = 0; r < W; ++r) {
    x.sc[r].assign(W, 0); x.rc[r].assign(W, 0);
    TRY(fmmbem_plan_exchange_counts(m.shards[r].get(), p, x.sc[r].data(), x.rc[r].data()));
    int64_t ns = 0, nr = 0;
    for (int q = 0; q < W; ++q) { ns += x.sc[r][q]; nr += x.rc[r][q]; }
    HIP_TRY(hipSetDevice(m.dev[r]));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&x.send[r]), sizeof(double) * (size_t)std::max<int64_t>(ns, 1)));
    HIP_TRY(hipMalloc(reinterpret_cast<void**>(&x.recv[r]), sizeof(double) * (size_t)std::max<int64_t>(nr, 1)));
  }
  for (int r = 0; r < W; ++r)                          // what r sends q is what q expects from r, or the lists are broken
    for (int q = 0; q < W; ++q)
      if (x.sc[r][q] != x.rc[q][r]) return fail(FMMBEM_ERR_INVALID, "internal: exchange counts of the shards do not match");
  *out = &m.xch.emplace(p, std::move(x)).first->second;
  return FMMBEM_OK;
}

static int multi_execute_device(fmmbem_plan* h, int p, const double* d_x, double* d_y, hipStream_t s0) {
  MultiDevice& m = *h->multi;
  const int W = m.world();
  if (!d_x || !d_y) return fail(FMMBEM_ERR_INVALID, "null vector");
  const size_t nd = (size_t)h->hp.n * h->d.dof;
  int prev = 0;
  (void)hipGetDevice(&prev);
  struct Restore { int d; ~Restore() { (void)hipSetDevice(d); } } restore{prev};
  MultiDevice::Xch* xc = nullptr;
  if (m.split) TRY(multi_xch(h, p, &xc));
  HIP_TRY(hipSetDevice(m.dev[0]));
  HIP_TRY(hipEventRecord(m.ev_x, s0));
  // Every device's share of a phase is ISSUED by a thread of its own (host_parallel): one thread took 0.43 ms to issue a matvec
  // over eight devices -- as long as a shard's share of the work runs (FMMBEM_MULTI_ISSUE_THREADS=0: the one thread).  A phase's
  // events are all recorded when its fan-out returns, which is what the next phase waits on.
  const bool threads = W > 1 && !(std::getenv("FMMBEM_MULTI_ISSUE_THREADS") && std::atoi(std::getenv("FMMBEM_MULTI_ISSUE_THREADS")) == 0);
  std::vector<int> rcs((size_t)W, FMMBEM_OK);
  std::vector<std::string> errs((size_t)W);
  auto fan_out = [&](auto&& body) -> int {
    auto one = [&](int r) {
      rcs[(size_t)r] = body(r);
      if (rcs[(size_t)r] != FMMBEM_OK) errs[(size_t)r] = fmmbem_last_error();     // (the text is the issuing thread's own)
    };
    if (threads) host_parallel(W, one); else for (int r = 0; r < W; ++r) one(r);
    for (int r = 0; r < W; ++r) if (rcs[(size_t)r] != FMMBEM_OK) return fail(rcs[(size_t)r], errs[(size_t)r]);
    return FMMBEM_OK;
  };
  TRY(fan_out([&](int r) -> int {                      // x to every device; the owners' upward pass
    fmmbem_plan& sh = *m.shards[r];
    HIP_TRY(hipSetDevice(m.dev[r]));
    HIP_TRY(hipStreamWaitEvent(sh.own_stream, m.ev_x, 0));
    HIP_TRY(hipMemcpyPeerAsync(m.x[r], m.dev[r], d_x, m.dev[0], sizeof(double) * nd, sh.own_stream));
    if (m.split) {
      TRY(sh.run(p, m.x[r], nullptr, sh.own_stream, false, 1, xc->send[r]));
      HIP_TRY(hipEventRecord(m.ev_up[r], sh.own_stream));
    }
    return FMMBEM_OK;
  }));
  // (the waits on the other shards' events in a fan-out of their own: a shard whose downward pass is being CAPTURED into a graph
  // on one thread makes another thread's wait on an event of that stream fail -- "dependency created on uncaptured work")
  if (m.split)
    TRY(fan_out([&](int q) -> int {                    // the multipoles q's lists read
      fmmbem_plan& sh = *m.shards[q];
      HIP_TRY(hipSetDevice(m.dev[q]));
      int64_t roff = 0;
      for (int r = 0; r < W; ++r) {
        const int64_t cnt = xc->rc[q][r];
        if (cnt > 0) {
          int64_t soff = 0;
          for (int k = 0; k < q; ++k) soff += xc->sc[r][k];
          HIP_TRY(hipStreamWaitEvent(sh.own_stream, m.ev_up[r], 0));
          HIP_TRY(hipMemcpyPeerAsync(xc->recv[q] + roff, m.dev[q], xc->send[r] + soff, m.dev[r], sizeof(double) * (size_t)cnt, sh.own_stream));
        }
        roff += cnt;
      }
      return FMMBEM_OK;
    }));
  TRY(fan_out([&](int q) -> int {                      // q's downward pass and near field, its slice home
    fmmbem_plan& sh = *m.shards[q];
    HIP_TRY(hipSetDevice(m.dev[q]));
    if (m.split) TRY(sh.run(p, nullptr, m.slice[q], sh.own_stream, false, 2, xc->recv[q]));
    else TRY(sh.run(p, m.x[q], m.slice[q], sh.own_stream, false));
    const size_t rows = (size_t)(m.cut[q + 1] - m.cut[q]) * h->d.dof;
    if (rows) HIP_TRY(hipMemcpyPeerAsync(m.gathered + (size_t)q * m.chunk, m.dev[0], m.slice[q], m.dev[q], sizeof(double) * rows, sh.own_stream));
    HIP_TRY(hipEventRecord(m.ev_done[q], sh.own_stream));
    return FMMBEM_OK;
  }));
  HIP_TRY(hipSetDevice(m.dev[0]));
  for (int q = 0; q < W; ++q) HIP_TRY(hipStreamWaitEvent(s0, m.ev_done[q], 0));
  TRY(fmmbem_plan_assemble_slices_device(m.shards[0].get(), m.gathered, m.chunk, d_y, s0));
  h->last_p = p;
  return FMMBEM_OK;
}

int fmmbem_plan_create(const fmmbem_options* opts, size_t n_panels, const double* vertices, const uint8_t* bc,
                       fmmbem_plan** out) {
  if (!opts || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  *out = nullptr;
  if (opts->kernel != FMMBEM_KERNEL_LAPLACE_BEM && opts->kernel != FMMBEM_KERNEL_STOKES_BEM)
    return fail(FMMBEM_ERR_UNSUPPORTED, "unknown kernel id");
  if (opts->kernel == FMMBEM_KERNEL_STOKES_BEM) {
    if (!(opts->mu > 0)) return fail(FMMBEM_ERR_INVALID, "Stokes: viscosity mu must be positive");
    // TRACTION panels: the near blocks are eval_traction_integral (kernel/StokesSphericalBEM.hpp:160-258).  Their far field
    // is the double-layer decomposition of kernels_far.hip (seven dipole potentials), checked against the Direct sum -- the
    // reference's own far field for this operator disagrees with its Direct sum by 50-75 %, so there
    // is nothing of the reference's to be equal to beyond Direct.  Orders up to 12 take the rotation M2L kernel, 13 ... 16 the
    // double sum over the eleven slots, one slot per pass.
  }
  if (!vertices || n_panels == 0) return fail(FMMBEM_ERR_INVALID, "no panels");
  if (opts->l2l_rule != FMMBEM_L2L_COMPLETE && opts->l2l_rule != FMMBEM_L2L_REFERENCE) return fail(FMMBEM_ERR_INVALID, "unknown l2l_rule");
  if (opts->evaluator < FMMBEM_EVAL_FMM || opts->evaluator > FMMBEM_EVAL_BLOCK_DIAGONAL) return fail(FMMBEM_ERR_INVALID, "unknown evaluator");
  // The geometry of a live plan, recognised: same options, same panel count, same vertex bytes (two 64-bit hashes) -- only the
  // boundary-condition flags may differ.  The new plan then shares that plan's tree, lists and tables (PlanShared) and builds
  // only what the flags decide.  This is the drivers' second plan (examples/LaplaceBEM.cpp:218-232: the same panels with the
  // flags switched, for the right-hand side); FMMBEM_PLAN_SHARE=0 turns the recognition off.
  {
    // a device list: the options' own, or FMMBEM_DEVICES=0,1,2,... for callers that cannot say (the reference's unmodified drivers)
    std::vector<int> devices;
    if (opts->n_devices > 1) devices.assign(opts->devices, opts->devices + std::min(opts->n_devices, 8));
    else if (opts->n_devices == 0 && opts->shard_world <= 1 && !opts->host_only) {
      if (const char* e = std::getenv("FMMBEM_DEVICES")) {
        for (const char* c = e; *c;) { char* end = nullptr; const long v = std::strtol(c, &end, 10); if (end == c) break; devices.push_back((int)v); c = *end ? end + 1 : end; }
        if (devices.size() < 2) devices.clear();
      }
    }
    if (!devices.empty()) {
      if (opts->host_only) return fail(FMMBEM_ERR_INVALID, "a device list and host_only exclude each other");
      return multi_create(opts, devices, n_panels, vertices, bc, out);
    }
  }
  if (!opts->host_only) (void)order_tables();             // the order tables start building now, beside the tree
  uint64_t fp[2] = {0, 0};
  const bool share_on = !opts->host_only && !(std::getenv("FMMBEM_PLAN_SHARE") && std::atoi(std::getenv("FMMBEM_PLAN_SHARE")) == 0);
  if (share_on) {
    fingerprint_vertices(vertices, n_panels, fp);
    std::unique_ptr<fmmbem_plan> copy;
    {
      std::lock_guard<std::mutex> lock(g_cache_mu);
      for (const fmmbem_plan* p : g_cache)
        if ((size_t)p->hp.n == n_panels && p->shared->fingerprint[0] == fp[0] && p->shared->fingerprint[1] == fp[1] && same_geometry_options(p->opts, *opts)) {
          copy.reset(new (std::nothrow) fmmbem_plan(*p));
          break;
        }
    }
    if (copy) return fmmbem_plan::like_finish(std::move(copy), bc, out);
  }
  std::unique_ptr<fmmbem_plan> pl(new (std::nothrow) fmmbem_plan);
  if (!pl) return fail(FMMBEM_ERR_ALLOC, "plan");
  pl->opts = *opts;
  pl->shared->fingerprint[0] = fp[0]; pl->shared->fingerprint[1] = fp[1];
  HostOptions ho;
  ho.p_max = opts->p_max; ho.quad_k = opts->quad_k; ho.theta = opts->theta; ho.ncrit = opts->ncrit;
  ho.shard_rank = opts->shard_rank; ho.shard_world = opts->shard_world < 1 ? 1 : opts->shard_world;
  ho.evaluator = opts->evaluator;
  ho.shard_upward = opts->shard_upward < 0 ? 0 : opts->shard_upward > 2 ? 2 : opts->shard_upward;
  ho.reference_l2l = opts->l2l_rule == FMMBEM_L2L_REFERENCE;
  ho.panels_on_device = !opts->host_only && !(std::getenv("FMMBEM_PANELS_ON_HOST") && std::atoi(std::getenv("FMMBEM_PANELS_ON_HOST")) != 0);
  const double t0 = now_ms();
  // The near field goes to the device as soon as the host plan holds what it needs -- the panels' geometry, the leaves, the
  // near lists -- and its assembly runs on the GPU while the host builds the far-field lists (FMMBEM_BUILD_TWO_PHASE=0: after them)
  int near_rc = FMMBEM_OK;
  bool near_done = false;
  double near_ms = 0;
  if (!opts->host_only && ho.panels_on_device && !(std::getenv("FMMBEM_BUILD_TWO_PHASE") && std::atoi(std::getenv("FMMBEM_BUILD_TWO_PHASE")) == 0))
    ho.after_near_lists = [&]() -> std::string {
      const double t = now_ms();
      pl->has_bc[0] = pl->hp.has_bc[0]; pl->has_bc[1] = pl->hp.has_bc[1];
      pl->create_vertices = vertices;
      try {
        near_rc = pl->to_device(1);
      } catch (const std::bad_alloc&) {
        near_rc = fail(FMMBEM_ERR_ALLOC, "host allocation failed while tabulating operators");
      }
      pl->create_vertices = nullptr;
      near_done = true;
      near_ms = now_ms() - t;
      return near_rc == FMMBEM_OK ? std::string() : std::string("device");
    };
  std::string err;
  try {
    err = pl->hp.build(ho, (int64_t)n_panels, vertices, bc);
  } catch (const std::bad_alloc&) {
    return fail(FMMBEM_ERR_ALLOC, "host allocation failed while building the plan");
  }
  pl->hp.opt.after_near_lists = nullptr;                // (it refers to this frame)
  if (near_rc != FMMBEM_OK) return near_rc;
  if (!err.empty()) return fail(err.find("octree") != std::string::npos ? FMMBEM_ERR_TREE : FMMBEM_ERR_INVALID, err);
  pl->build_host_ms = now_ms() - t0 - near_ms;
  pl->has_bc[0] = pl->hp.has_bc[0]; pl->has_bc[1] = pl->hp.has_bc[1];
  if (!opts->host_only) {
    try {
      pl->create_vertices = ho.panels_on_device ? vertices : nullptr;
      const int rc = pl->to_device(near_done ? 2 : 0);
      pl->create_vertices = nullptr;
      if (rc != FMMBEM_OK) return rc;
    } catch (const std::bad_alloc&) {
      return fail(FMMBEM_ERR_ALLOC, "host allocation failed while tabulating operators");
    }
    // vertices are only needed on the device; keep the host copy small
    pl->hp.panels.vert.clear(); pl->hp.panels.vert.shrink_to_fit();
    if (share_on) geometry_cache_add(pl.get());
  }
  *out = pl.release();
  return FMMBEM_OK;
}

int fmmbem_plan_create_like(const fmmbem_plan* base, const uint8_t* bc, fmmbem_plan** out) {
  if (!base || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  *out = nullptr;
  if (!base->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "fmmbem_plan_create_like: the base plan was built host-only");
  if (base->multi) return multi_like(*base, bc, out);
  return fmmbem_plan::like(*base, bc, out);
}

void fmmbem_plan_destroy(fmmbem_plan* plan) {
  if (plan) geometry_cache_remove(plan);
  delete plan;
}

int fmmbem_plan_execute_device(fmmbem_plan* plan, int p, const double* d_x, double* d_y, void* stream) {
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  if (plan->multi) return multi_execute_device(plan, p, d_x, d_y, static_cast<hipStream_t>(stream));
  return plan->run(p, d_x, d_y, static_cast<hipStream_t>(stream), false);
}

// expansion slots an exchange carries per box, from the HOST lists (a host-only plan answers too): Laplace one per boundary
// condition present; Stokes four potentials for velocity targets, seven for TRACTION targets (to_device, d.act)
static int64_t active_slots(const fmmbem_plan* plan) {
  const HostPlan& h = plan->hp;
  if (plan->opts.kernel != FMMBEM_KERNEL_STOKES_BEM) return (int64_t)plan->has_bc[0] + (int64_t)plan->has_bc[1];
  const bool trac = plan->has_bc[1] && h.opt.evaluator == 0;
  const bool vel = plan->has_bc[0] || !trac;
  return (vel ? 4 : 0) + (trac ? 7 : 0);
}

int fmmbem_plan_exchange_doubles(const fmmbem_plan* plan, int p, size_t* per_shard) {
  if (!plan || !per_shard) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (p < 1 || p > plan->hp.opt.p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  // from the host lists, so that a host-only plan (CPU tests of the N > 1 path) answers too
  const HostPlan& h = plan->hp;
  size_t most = 0;
  for (size_t r = 0; r + 1 < h.xch_ptr.size(); ++r) most = std::max<size_t>(most, (size_t)(h.xch_ptr[r + 1] - h.xch_ptr[r]));
  const size_t n_act = (size_t)active_slots(plan);
  *per_shard = (h.opt.shard_upward && h.opt.shard_world > 1) ? most * n_act * (size_t)(p * (p + 1) / 2) * 2 : 0;
  return FMMBEM_OK;
}

int fmmbem_plan_exchange_counts(const fmmbem_plan* plan, int p, int64_t* send_doubles, int64_t* recv_doubles) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan || !send_doubles || !recv_doubles) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (p < 1 || p > plan->hp.opt.p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  const HostPlan& h = plan->hp;
  const int W = h.opt.shard_world;
  const bool sel = h.opt.shard_upward == 2 && W > 1;
  if (!sel) return fail(FMMBEM_ERR_INVALID, "plan was not created with shard_upward = 2");
  const int64_t n_act = active_slots(plan);
  const int64_t per = n_act * (int64_t)(p * (p + 1) / 2) * 2;
  for (int q = 0; q < W; ++q) {
    send_doubles[q] = (int64_t)(h.xsel_send_ptr[q + 1] - h.xsel_send_ptr[q]) * per;
    recv_doubles[q] = (int64_t)(h.xsel_recv_ptr[q + 1] - h.xsel_recv_ptr[q]) * per;
  }
  return FMMBEM_OK;
}

int fmmbem_plan_upward_device(fmmbem_plan* plan, int p, const double* d_x, double* d_send, void* stream) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  return plan->run(p, d_x, nullptr, static_cast<hipStream_t>(stream), false, 1, d_send);
}

int fmmbem_plan_downward_device(fmmbem_plan* plan, int p, const double* d_recv, double* d_y, void* stream) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  return plan->run(p, nullptr, d_y, static_cast<hipStream_t>(stream), false, 2, const_cast<double*>(d_recv));
}

int fmmbem_plan_near_split_device(fmmbem_plan* plan, double* d_y, void* stream) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  return plan->run(plan->last_p > 0 ? plan->last_p : 1, nullptr, d_y, static_cast<hipStream_t>(stream), false, 3, nullptr);
}

int fmmbem_plan_shard_rows(const fmmbem_plan* plan, int64_t* cut) {
  if (!plan || !cut) return fail(FMMBEM_ERR_INVALID, "null argument");
  const HostPlan& h = plan->hp;
  std::vector<int> leaf_cut;
  partition_leaves(h, h.opt.shard_world, leaf_cut);
  for (int r = 0; r <= h.opt.shard_world; ++r)
    cut[r] = leaf_cut[r] < h.nleaves() ? (int64_t)h.box_body_begin[h.leaf_box[leaf_cut[r]]] : h.n;
  return FMMBEM_OK;
}

int fmmbem_plan_set_result_slices(fmmbem_plan* plan, int enabled) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  plan->result_slices = enabled != 0;
  return FMMBEM_OK;
}

int fmmbem_plan_assemble_slices_device(fmmbem_plan* plan, const double* d_slices, size_t chunk_doubles, double* d_y, void* stream) {
  if (!plan || !d_slices || !d_y) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (!plan->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "plan was built host-only; there is no CPU execution path");
  const int world = plan->hp.opt.shard_world;
  DEVICE_SCOPE(plan->opts.device);
  if (!plan->d_cut) {
    std::vector<int64_t> cut((size_t)world + 1);
    TRY(fmmbem_plan_shard_rows(plan, cut.data()));
    for (int r = 0; r < world; ++r)
      if ((size_t)(cut[r + 1] - cut[r]) * plan->d.dof > chunk_doubles) return fail(FMMBEM_ERR_INVALID, "chunk smaller than a shard's slice");
    TRY(plan->alloc((size_t)world + 1, &plan->d_cut, false));
    HIP_TRY(hipMemcpy(plan->d_cut, cut.data(), sizeof(int64_t) * cut.size(), hipMemcpyHostToDevice));
  }
  HIP_TRY(launch_assemble_slices(plan->d, d_slices, d_y, world, plan->d_cut, (int64_t)chunk_doubles, static_cast<hipStream_t>(stream)));
  return FMMBEM_OK;
}

int fmmbem_plan_near_device(fmmbem_plan* plan, const double* d_x, double* d_y, void* stream) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  return plan->run(1, d_x, d_y, static_cast<hipStream_t>(stream), true);
}

int fmmbem_plan_execute(fmmbem_plan* plan, int p, const double* x, double* y) {
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  if (!plan->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "plan was built host-only; there is no CPU execution path");
  if (!x || !y) return fail(FMMBEM_ERR_INVALID, "null vector");
  if (plan->result_slices)         // the owned rows in tree order at the head of y are only meaningful to the device-side all-gather
    return fail(FMMBEM_ERR_INVALID, "plan delivers result slices (fmmbem_plan_set_result_slices): use the device entry points");
  DEVICE_SCOPE(plan->opts.device);
  const size_t bytes = sizeof(double) * (size_t)plan->hp.n * (plan->opts.kernel == FMMBEM_KERNEL_STOKES_BEM ? 3 : 1);
  hipStream_t s = plan->own_stream;
  HIP_TRY(hipMemcpyAsync(plan->stage_x, x, bytes, hipMemcpyHostToDevice, s));
  const int rc = plan->multi ? multi_execute_device(plan, p, plan->stage_x, plan->stage_y, s) : plan->run(p, plan->stage_x, plan->stage_y, s, false);
  if (rc != FMMBEM_OK) return rc;
  HIP_TRY(hipMemcpyAsync(y, plan->stage_y, bytes, hipMemcpyDeviceToHost, s));
  HIP_TRY(hipStreamSynchronize(s));
  return FMMBEM_OK;
}

int fmmbem_host_register(void* ptr, size_t bytes) {
  if (!ptr || !bytes) return fail(FMMBEM_ERR_INVALID, "null buffer");
  HIP_TRY(hipHostRegister(ptr, bytes, hipHostRegisterDefault));
  return FMMBEM_OK;
}

int fmmbem_host_unregister(void* ptr) {
  if (!ptr) return fail(FMMBEM_ERR_INVALID, "null buffer");
  HIP_TRY(hipHostUnregister(ptr));
  return FMMBEM_OK;
}

int fmmbem_plan_set_graphs(fmmbem_plan* plan, int enabled) {
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  if (plan->multi) { for (auto& sh : plan->multi->shards) sh->use_graphs = enabled != 0; return FMMBEM_OK; }
  plan->use_graphs = enabled != 0 && plan->on_device;
  return FMMBEM_OK;
}

int fmmbem_plan_set_timing(fmmbem_plan* plan, int enabled) {
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  if (plan->multi) { for (auto& sh : plan->multi->shards) TRY(fmmbem_plan_set_timing(sh.get(), enabled)); return FMMBEM_OK; }
  plan->timing = plan->on_device ? (enabled == 2 ? 2 : enabled != 0) : 0;
  plan->ev_count = 0;
  return FMMBEM_OK;
}

int fmmbem_plan_stats(const fmmbem_plan* plan, fmmbem_stats* o) {
  if (!plan || !o) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (plan->multi) {
    // the whole operator: what the shards own summed, the tree's figures from any of them, stage times the LONGEST shard's
    // (they run side by side), build times likewise
    const MultiDevice& m = *plan->multi;
    TRY(fmmbem_plan_stats(m.shards[0].get(), o));
    for (int r = 1; r < m.world(); ++r) {
      fmmbem_stats t;
      TRY(fmmbem_plan_stats(m.shards[r].get(), &t));
      o->near_nnz += t.near_nnz; o->m2l_pairs_owned += t.m2l_pairs_owned; o->near_bytes += t.near_bytes;
      o->near_side_entries += t.near_side_entries; o->near_recomputed_pairs += t.near_recomputed_pairs;
      o->m2m_ops = std::max(o->m2m_ops, t.m2m_ops); o->l2l_ops += t.l2l_ops; o->l2p_leaves += t.l2p_leaves; o->p2m_leaves += t.p2m_leaves;
      o->m2l_items += t.m2l_items; o->m2l_passes += t.m2l_passes;
      o->owned_leaf_end = t.owned_leaf_end; o->owned_row_end = t.owned_row_end;
      o->build_host_ms = std::max(o->build_host_ms, t.build_host_ms); o->build_assemble_ms = std::max(o->build_assemble_ms, t.build_assemble_ms);
      double* a[] = {&o->ms_total, &o->ms_gather, &o->ms_near, &o->ms_scatter, &o->ms_p2m, &o->ms_m2m, &o->ms_mh, &o->ms_m2l, &o->ms_l2l, &o->ms_l2p};
      const double b[] = {t.ms_total, t.ms_gather, t.ms_near, t.ms_scatter, t.ms_p2m, t.ms_m2m, t.ms_mh, t.ms_m2l, t.ms_l2l, t.ms_l2p};
      for (int k = 0; k < 10; ++k) *a[k] = std::max(*a[k], b[k]);
    }
    o->last_p = plan->last_p;
    o->n_devices = m.world();
    return FMMBEM_OK;
  }
  const HostPlan& h = plan->hp;
  std::memset(o, 0, sizeof(*o));
  o->n_panels = h.n; o->n_boxes = h.nboxes; o->n_leaves = h.nleaves(); o->n_levels = h.nlevels;
  o->near_nnz = h.near_nnz_owned; o->near_nnz_total = h.near_nnz_total;
  o->p2p_pairs = (int64_t)h.p2p_src.size(); o->m2l_pairs = (int64_t)h.lr_src.size(); o->m2l_pairs_owned = h.m2l_pairs_owned;
  o->m2m_ops = h.m2m_ops; o->l2l_ops = h.l2l_ops;
  o->l2l_reference_omitted = h.l2l_ref_omitted;
  o->p2m_leaves = (int64_t)h.p2m_leaves.size(); o->l2p_leaves = (int64_t)h.l2p_leaves.size();
  o->m2l_classes = (int64_t)h.m2l_class_rep.size() / 2;
  o->owned_leaf_begin = h.leaf_begin; o->owned_leaf_end = h.leaf_end;
  o->owned_row_begin = h.row_begin; o->owned_row_end = h.row_end;
  o->near_bytes = plan->near_bytes;
  const bool long_items = plan->last_p > 0 && m2l_rot_long_items(plan->last_p) && plan->use_rot(plan->last_p);   // the cut the last execute ran
  o->m2l_items = (int64_t)(long_items ? h.rot_item_ptr_long : h.rot_item_ptr).size() - 1;
  o->m2l_passes = long_items ? h.rot_passes_long : h.rot_passes;
  o->near_side_entries = plan->near_side_entries;
  o->near_recomputed_pairs = plan->near_recomputed_pairs;
  o->geometry_shared = (int32_t)plan->shared.use_count();
  o->n_devices = 1;
  o->expansion_slots = plan->on_device ? plan->d.nslots : (plan->opts.kernel == FMMBEM_KERNEL_STOKES_BEM ? 8 : 2);
  o->m2l_kernel = plan->last_p > 0 ? (plan->use_rot(plan->last_p) ? 1 : plan->last_p <= 4 ? 3 : 2) : 0;
  o->rot_nop_orders = (int64_t)rot_nop_orders_m2l() | ((int64_t)rot_nop_orders_m2m() << 16) | ((int64_t)rot_nop_orders_l2l() << 32);
  o->tree_coder_levels = h.tree_levels_max;
  o->expansions_active = (plan->has_bc[0] ? 1 : 0) | (plan->has_bc[1] ? 2 : 0);
  o->last_p = plan->last_p;
  o->build_host_ms = plan->build_host_ms; o->build_assemble_ms = plan->build_assemble_ms;
  if (plan->on_device && plan->ev_count > 0) {
    // mean stage times over the recorded executes (waits for the recorded events)
    constexpr int NS = fmmbem_plan::kStages, NR = fmmbem_plan::kRing;
    const int64_t have = plan->ev_count < NR ? plan->ev_count : NR;
    double sum[NS + 1] = {0};
    int64_t used = 0;
    DEVICE_SCOPE(plan->opts.device);
    for (int64_t i = 0; i < have; ++i) {
      const int64_t slot = (plan->ev_count - 1 - i) % NR;
      const hipEvent_t* set = &plan->ev[(size_t)slot * 2 * NS];
      const unsigned mask = plan->ev_mask[slot];
      int last = -1, first = -1;
      for (int k = 0; k < NS; ++k) if (mask & (1u << k)) { last = k; if (first < 0) first = k; }
      if (mask & 4u) last = 2;                          // the delivery of the result (stage 2) is the last thing an execute does
      if (last < 0) continue;
      float f = 0;
      for (int k = 0; k < NS; ++k) {
        if (!(mask & (1u << k))) continue;
        HIP_TRY(hipEventSynchronize(set[2 * k + 1]));
        HIP_TRY(hipEventElapsedTime(&f, set[2 * k], set[2 * k + 1]));
        sum[k] += f;
      }
      HIP_TRY(hipEventElapsedTime(&f, set[2 * first], set[2 * last + 1]));      // both on the caller's stream
      sum[NS] += f;
      ++used;
    }
    if (used) {
      const double inv = 1.0 / double(used);
      o->ms_gather = sum[0] * inv; o->ms_near = sum[1] * inv; o->ms_scatter = sum[2] * inv; o->ms_p2m = sum[3] * inv;
      o->ms_m2m = sum[4] * inv; o->ms_mh = sum[5] * inv; o->ms_m2l = sum[6] * inv; o->ms_l2l = sum[7] * inv;
      o->ms_l2p = sum[8] * inv; o->ms_total = sum[NS] * inv;
      o->timed_executes = used;
    }
  }
  return FMMBEM_OK;
}

int fmmbem_plan_get_perm(const fmmbem_plan* plan, uint32_t* out) {
  if (!plan || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  std::memcpy(out, plan->hp.perm.data(), sizeof(uint32_t) * (size_t)plan->hp.n);
  return FMMBEM_OK;
}

int fmmbem_plan_get_boxes(const fmmbem_plan* plan, double* center, double* side, int32_t* level, int32_t* is_leaf,
                          int32_t* parent, int32_t* body_begin, int32_t* body_end) {
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  const HostPlan& h = plan->hp;
  for (int b = 0; b < h.nboxes; ++b) {
    if (center) std::memcpy(center + 3 * b, &h.box_center[3 * b], sizeof(double) * 3);
    if (side) side[b] = h.box_side[b];
    if (level) level[b] = h.box_level[b];
    if (is_leaf) is_leaf[b] = h.box_leaf[b];
    if (parent) parent[b] = h.box_parent[b];
    if (body_begin) body_begin[b] = h.box_body_begin[b];
    if (body_end) body_end[b] = h.box_body_end[b];
  }
  return FMMBEM_OK;
}

int fmmbem_plan_get_pairs(const fmmbem_plan* plan, int which, int32_t* out, int64_t* n) {
  if (!plan || !n) return fail(FMMBEM_ERR_INVALID, "null argument");
  const HostPlan& h = plan->hp;
  std::vector<int32_t> flat;
  switch (which) {
    case 0: for (size_t i = 0; i < h.p2p_src.size(); ++i) { flat.push_back(h.p2p_src[i]); flat.push_back(h.p2p_tgt[i]); } break;
    case 1: for (size_t i = 0; i < h.lr_src.size(); ++i) { flat.push_back(h.lr_src[i]); flat.push_back(h.lr_tgt[i]); } break;
    case 2:
      for (int par : h.m2m_parents)
        for (int c = h.box_child_begin[par]; c < h.box_child_end[par]; ++c) { flat.push_back(c); flat.push_back(par); }
      break;
    case 3: for (int c : h.l2l_children) { flat.push_back(h.box_parent[c]); flat.push_back(c); } break;
    case 4:
      if (h.rot_alias) {
        for (int b = 0; b < h.nboxes; ++b)
          for (int i = h.m2l_ptr[b]; i < h.m2l_ptr[b + 1]; ++i) { flat.push_back(h.m2l_src[i]); flat.push_back(b); }
      } else
        for (size_t i = 0; i < h.rot_src.size(); ++i) { flat.push_back(h.rot_src[i]); flat.push_back(h.rot_tgt[i]); }
      break;
    case 5: for (size_t i = 0; i + 1 < h.rot_item_ptr.size(); ++i) { flat.push_back(h.rot_item_ptr[i]); flat.push_back(h.rot_item_ptr[i + 1]); } break;
    case 6: for (size_t i = 0; i + 1 < h.rot_item_ptr_long.size(); ++i) { flat.push_back(h.rot_item_ptr_long[i]); flat.push_back(h.rot_item_ptr_long[i + 1]); } break;
    default: return fail(FMMBEM_ERR_INVALID, "which must be 0..6");
  }
  *n = (int64_t)flat.size() / 2;
  if (out) std::memcpy(out, flat.data(), flat.size() * sizeof(int32_t));
  return FMMBEM_OK;
}

int fmmbem_plan_get_near_row(const fmmbem_plan* plan, int64_t row, uint32_t* cols, double* vals, int64_t* n) {
  if (!plan || !n) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (plan->multi) {
    const MultiDevice& m = *plan->multi;
    const int64_t prow = row / plan->d.dof;
    for (int r = 0; r < m.world(); ++r)
      if (prow >= m.cut[r] && prow < m.cut[r + 1]) return fmmbem_plan_get_near_row(m.shards[r].get(), row, cols, vals, n);
    return fail(FMMBEM_ERR_INVALID, "row out of range");
  }
  const HostPlan& h = plan->hp;
  const int dof = plan->opts.kernel == FMMBEM_KERNEL_STOKES_BEM ? 3 : 1;
  const int64_t prow = row / dof;                      // panel row (tree order); row counts unknowns
  const int comp = (int)(row % dof);
  if (prow < h.row_begin || prow >= h.row_end) return fail(FMMBEM_ERR_INVALID, "row not owned by this shard");
  // owning leaf: last leaf whose first row <= prow
  int lo = h.leaf_begin, hi = h.leaf_end - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) / 2;
    if (h.box_body_begin[h.leaf_box[mid]] <= prow) lo = mid; else hi = mid - 1;
  }
  const int leaf = lo, tb = h.leaf_box[leaf];
  const int ncols = dof * h.near_ncols[leaf];
  *n = ncols;
  if (cols) {
    int at = 0;
    for (int64_t s = h.near_ptr[leaf]; s < h.near_ptr[leaf + 1]; ++s) {
      const int sb = h.leaf_box[h.near_src[s]];
      for (int j = h.box_body_begin[sb]; j < h.box_body_end[sb]; ++j)
        for (int b = 0; b < dof; ++b) cols[at++] = (uint32_t)(dof * j + b);
    }
  }
  if (vals) {
    if (!plan->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "near values live on the device");
    if (!plan->opts.sparse_local) return fail(FMMBEM_ERR_INVALID, "matrix-free plan holds no near matrix");
    DEVICE_SCOPE(plan->opts.device);
    if (!plan->near_rec_host.empty() && plan->near_rec_host[leaf]) {
      // hybrid plan, recomputed leaf: no block is stored -- the row is evaluated now, by the entry functions of the assembly
      const int ncp = h.near_ncols[leaf];
      std::vector<int> pcol((size_t)ncp);
      int at = 0;
      for (int64_t s = h.near_ptr[leaf]; s < h.near_ptr[leaf + 1]; ++s) {
        const int sb = h.leaf_box[h.near_src[s]];
        for (int j = h.box_body_begin[sb]; j < h.box_body_end[sb]; ++j) pcol[at++] = j;
      }
      int* d_col = nullptr;
      double* d_val = nullptr;
      HIP_TRY(hipMalloc(&d_col, sizeof(int) * (size_t)ncp));
      if (hipMalloc(&d_val, sizeof(double) * (size_t)ncp * dof * dof) != hipSuccess) { (void)hipFree(d_col); return fail(FMMBEM_ERR_ALLOC, "device allocation failed"); }
      std::vector<double> blk((size_t)ncp * dof * dof);
      hipError_t e = hipMemcpy(d_col, pcol.data(), sizeof(int) * (size_t)ncp, hipMemcpyHostToDevice);
      if (e == hipSuccess) e = launch_near_row_eval(plan->d, prow, d_col, ncp, d_val, plan->own_stream);
      if (e == hipSuccess) e = hipStreamSynchronize(plan->own_stream);
      if (e == hipSuccess) e = hipMemcpy(blk.data(), d_val, sizeof(double) * blk.size(), hipMemcpyDeviceToHost);
      (void)hipFree(d_col); (void)hipFree(d_val);
      if (e != hipSuccess) return fail(FMMBEM_ERR_HIP, hipGetErrorString(e));
      for (int c = 0; c < ncp; ++c)
        for (int b = 0; b < dof; ++b) vals[dof * c + b] = blk[(size_t)c * dof * dof + comp * dof + b];
      return FMMBEM_OK;
    }
    const int64_t off = plan->near_off_host[leaf];
    if (plan->d.near_sym) {                            // Stokes, symmetric blocks: expand row `comp` of the panel row's 3x3 blocks
      const int ncp = h.near_ncols[leaf];
      const int64_t soff = plan->sym_off_host[leaf];
      std::vector<double> six((size_t)6 * ncp);
      HIP_TRY(hipMemcpy(six.data(), plan->d.near_sym + soff + (prow - h.box_body_begin[tb]) * 6 * ncp, sizeof(double) * six.size(), hipMemcpyDeviceToHost));
      const double *p0 = six.data(), *p1 = p0 + 2 * ncp, *p2 = p1 + 2 * ncp;      // (xx,xy) (xz,yy) (yz,zz) per source panel
      for (int c = 0; c < ncp; ++c) {
        const double m[3][3] = {{p0[2 * c], p0[2 * c + 1], p1[2 * c]}, {p0[2 * c + 1], p1[2 * c + 1], p2[2 * c]}, {p1[2 * c], p2[2 * c], p2[2 * c + 1]}};
        for (int b = 0; b < 3; ++b) vals[3 * c + b] = m[comp][b];
      }
      return FMMBEM_OK;
    }
    const int stride = (ncols + 1) & ~1;
    const int64_t r = (prow - h.box_body_begin[tb]) * dof + comp;
    HIP_TRY(hipMemcpy(vals, plan->d.near_val + off + r * stride, sizeof(double) * (size_t)ncols, hipMemcpyDeviceToHost));
  }
  return FMMBEM_OK;
}

int fmmbem_plan_get_diagonal(const fmmbem_plan* plan, double* out) {
  if (!plan || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (plan->multi) {                                   // every shard returns its own rows and zeros elsewhere
    const MultiDevice& m = *plan->multi;
    const size_t nd = (size_t)plan->hp.n * plan->d.dof;
    std::vector<double> part(nd);
    std::fill(out, out + nd, 0.0);
    for (int r = 0; r < m.world(); ++r) {
      TRY(fmmbem_plan_get_diagonal(m.shards[r].get(), part.data()));
      for (size_t i = 0; i < nd; ++i) out[i] += part[i];
    }
    return FMMBEM_OK;
  }
  if (!plan->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "near values live on the device");
  if (!plan->opts.sparse_local) return fail(FMMBEM_ERR_INVALID, "matrix-free plan holds no near matrix");
  const HostPlan& h = plan->hp;
  const int dof = plan->d.dof;
  DEVICE_SCOPE(plan->opts.device);
  std::vector<int> selfcol(h.nleaves(), 0);
  for (int l = h.leaf_begin; l < h.leaf_end; ++l) {
    int col = 0;
    bool found = false;
    for (int64_t s = h.near_ptr[l]; s < h.near_ptr[l + 1]; ++s) {
      if (h.near_src[s] == l) { found = true; break; }
      const int sb = h.leaf_box[h.near_src[s]];
      col += h.box_body_end[sb] - h.box_body_begin[sb];
    }
    if (!found) return fail(FMMBEM_ERR_INVALID, "internal: a leaf without its self block");
    selfcol[l] = dof * col;
  }
  int* d_sc = nullptr;
  double* d_out = nullptr;
  const size_t nb = sizeof(double) * (size_t)h.n * dof;
  HIP_TRY(hipMalloc(&d_sc, sizeof(int) * selfcol.size()));
  if (hipMalloc(&d_out, nb) != hipSuccess) { (void)hipFree(d_sc); return fail(FMMBEM_ERR_ALLOC, "device allocation failed"); }
  hipError_t e = hipMemcpy(d_sc, selfcol.data(), sizeof(int) * selfcol.size(), hipMemcpyHostToDevice);
  if (e == hipSuccess) e = hipMemset(d_out, 0, nb);                       // rows of other shards stay zero
  if (e == hipSuccess) e = launch_near_diag(plan->d, d_sc, d_out, plan->own_stream);
  if (e == hipSuccess) e = hipStreamSynchronize(plan->own_stream);
  if (e == hipSuccess) e = hipMemcpy(out, d_out, nb, hipMemcpyDeviceToHost);
  (void)hipFree(d_sc); (void)hipFree(d_out);
  if (e != hipSuccess) return fail(FMMBEM_ERR_HIP, hipGetErrorString(e));
  return FMMBEM_OK;
}

int fmmbem_plan_get_expansions(const fmmbem_plan* plan, int which, int p, double* out) {
  if (plan && plan->multi) return fail(FMMBEM_ERR_UNSUPPORTED, "not on a multi-device plan: it drives its shards itself");
  if (!plan || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (!plan->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "expansions live on the device");
  if (p < 1 || p > plan->hp.opt.p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  DEVICE_SCOPE(plan->opts.device);
  const DevicePlan& d = plan->d;
  const int ns = d.nslots;
  std::vector<double2> tmp((size_t)d.nboxes * ns * d.s_max);
  HIP_TRY(hipMemcpy(tmp.data(), which == 0 ? d.M : d.L, tmp.size() * sizeof(double2), hipMemcpyDeviceToHost));
  const int S = p * (p + 1) / 2;
  for (int b = 0; b < d.nboxes; ++b)
    for (int s = 0; s < ns; ++s)
      for (int i = 0; i < S; ++i) {
        const double2 v = tmp[((size_t)b * ns + s) * d.s_max + i];
        out[(((size_t)b * ns + s) * S + i) * 2] = v.x;
        out[(((size_t)b * ns + s) * S + i) * 2 + 1] = v.y;
      }
  return FMMBEM_OK;
}

int fmmbem_kernel_entries(const fmmbem_options* opts, size_t n, const double* target_vertices, const uint8_t* target_bc,
                          const double* source_vertices, double* out) {
  if (!opts || !target_vertices || !source_vertices || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (opts->kernel != FMMBEM_KERNEL_LAPLACE_BEM && opts->kernel != FMMBEM_KERNEL_STOKES_BEM)
    return fail(FMMBEM_ERR_UNSUPPORTED, "unknown kernel id");
  if (n == 0) return FMMBEM_OK;
  if (n > ((size_t)1 << 30)) return fail(FMMBEM_ERR_INVALID, "too many pairs");
  const bool stokes = opts->kernel == FMMBEM_KERNEL_STOKES_BEM;
  QuadRule rule, fine;
  if (!quad_rule(opts->quad_k, rule)) return fail(FMMBEM_ERR_INVALID, "invalid quadrature key (valid: 1 3 4 7 13 17 19 25 79)");
  if (stokes) {
    if (!quad_rule(opts->quad_k_fine, fine)) return fail(FMMBEM_ERR_INVALID, "invalid K_fine (valid: 1 3 4 7 13 17 19 25 79)");
    if (!(opts->mu > 0)) return fail(FMMBEM_ERR_INVALID, "Stokes: viscosity mu must be positive");
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FMMBEM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU execution path)");
  if (opts->device < 0 || opts->device >= ndev) return fail(FMMBEM_ERR_INVALID, "device ordinal out of range");
  DEVICE_SCOPE(opts->device);
  const int64_t N = 2 * (int64_t)n;
  PanelSoA P;
  try {
    alloc_panels(P, N, rule.n);
  } catch (const std::bad_alloc&) {
    return fail(FMMBEM_ERR_ALLOC, "host allocation failed");
  }
  for (size_t i = 0; i < n; ++i) {
    fill_panel(P, N, (int64_t)i, target_vertices + 9 * i, rule, target_bc ? (target_bc[i] ? 1 : 0) : 0);
    fill_panel(P, N, (int64_t)(n + i), source_vertices + 9 * i, rule, 0);
  }
  struct Frees {
    std::vector<void*> v;
    ~Frees() { for (void* p : v) (void)hipFree(p); }
  } frees;
  auto up = [&](const void* src, size_t bytes, const void** dst) -> hipError_t {
    void* p = nullptr;
    hipError_t e = hipMalloc(&p, bytes ? bytes : 1);
    if (e != hipSuccess) return e;
    frees.v.push_back(p);
    *dst = p;
    return bytes ? hipMemcpy(p, src, bytes, hipMemcpyHostToDevice) : hipSuccess;
  };
  DevicePlan d{};
  d.n = N; d.nq = rule.n; d.kernel = opts->kernel; d.dof = stokes ? 3 : 1; d.mu = opts->mu;
  for (int q = 0; q < rule.n; ++q) d.qw[q] = rule.w[q];
  std::vector<double> qf((size_t)fine.n * 4);
  if (stokes) {
    d.nqf = fine.n;
    for (int q = 0; q < fine.n; ++q) { for (int k = 0; k < 3; ++k) qf[4 * q + k] = fine.pts[q][k]; qf[4 * q + 3] = fine.w[q]; }
  }
#define UP(field, vec) HIP_TRY(up(vec.data(), vec.size() * sizeof(vec[0]), reinterpret_cast<const void**>(&d.field)))
  UP(cx, P.cx); UP(cy, P.cy); UP(cz, P.cz); UP(nx, P.nx); UP(ny, P.ny); UP(nz, P.nz);
  UP(area, P.area); UP(quad, P.quad); UP(vert, P.vert); UP(bc, P.bc);
  if (stokes) UP(qf, qf);
#undef UP
  const size_t per = stokes ? 9 : 1;
  double* d_out = nullptr;
  HIP_TRY(hipMalloc(&d_out, sizeof(double) * n * per));
  frees.v.push_back(d_out);
  HIP_TRY(launch_kernel_entries(d, (int)n, d_out, nullptr));
  HIP_TRY(hipMemcpy(out, d_out, sizeof(double) * n * per, hipMemcpyDeviceToHost));
  return FMMBEM_OK;
}

int fmmbem_mesh_unit_sphere(int recursions, double* vertices, size_t* n_panels) {
  if (recursions < 1 || recursions > 12 || !n_panels) return fail(FMMBEM_ERR_INVALID, "recursions must be 1..12");
  *n_panels = (size_t)unit_sphere(recursions, vertices);
  return FMMBEM_OK;
}

const char* fmmbem_status_string(int status) {
  switch (status) {
    case FMMBEM_OK: return "ok";
    case FMMBEM_ERR_INVALID: return "invalid argument";
    case FMMBEM_ERR_NO_DEVICE: return "no HIP device";
    case FMMBEM_ERR_HIP: return "HIP runtime error";
    case FMMBEM_ERR_ALLOC: return "allocation failed";
    case FMMBEM_ERR_TREE: return "octree too deep";
    case FMMBEM_ERR_UNSUPPORTED: return "unsupported option";
    case FMMBEM_ERR_IO: return "file input/output error";
    default: return "unknown status";
  }
}

const char* fmmbem_last_error(void) { return g_last_error.c_str(); }
int fmmbem_version(void) { return FMMBEM_VERSION; }

}  // extern "C"

// what the solver of krylov.hip needs to know about a plan; it reaches the matvec through the public entry point
int fmmbem::plan_solver_info(fmmbem_plan* plan, int* device, int64_t* unknowns, int* p_max, fmmbem::SolverWs*** slot) {
  if (!plan) return fail(FMMBEM_ERR_INVALID, "null plan");
  if (!plan->on_device) return fail(FMMBEM_ERR_NO_DEVICE, "plan was built host-only; there is no CPU execution path");
  if (!plan->multi && (plan->result_slices || plan->hp.opt.shard_world > 1))
    return fail(FMMBEM_ERR_UNSUPPORTED, "fmmbem_gmres runs on a whole operator; shards are driven by the caller's collectives (distributed.py)");
  *device = plan->opts.device;
  *unknowns = (int64_t)plan->hp.n * (plan->opts.kernel == FMMBEM_KERNEL_STOKES_BEM ? 3 : 1);
  *p_max = plan->hp.opt.p_max;
  *slot = &plan->solver_ws;
  return FMMBEM_OK;
}

