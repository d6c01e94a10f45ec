// ops.hip -- the translation operators ONE AT A TIME (fmmbem_ops_* of include/fmmbem.h): the member functions the reference's
// kernel classes offer beside operator() -- P2M, M2M, M2L, L2L, L2P of kernel/KernelSkeleton.hpp:62-212, as
// kernel/LaplaceSphericalBEM.hpp:307-476 and kernel/StokesSphericalBEM.hpp:391-530 implement them -- for callers that drive
// single operators (the reference's tests/single_level.cpp does).  Nothing here is a second implementation: every call builds the
// smallest DevicePlan that holds its operands (two boxes: 0 = source / the leaf, 1 = target; one pair; one class) and launches the
// SAME kernels a plan's matvec launches -- p2m_kernel, the rotation kernels of kernels_m2l_rot.hip (p <= 12), the sparse shift
// operators and the double-sum M2L above, l2p_kernel / l2p_stokes_kernel.  Operands and results are host buffers; a call costs a
// few launches and copies (~0.1 ms), which is what a checker pays, not what a matvec pays.
#include <hip/hip_runtime.h>

#include <cmath>
#include <complex>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <string>
#include <vector>

#include "../../include/fmmbem.h"
#include "device_launch.hpp"
#include "host_plan.hpp"
#include "host_tables.hpp"
#include "m2l_layout.hpp"
#include "m2l_rot.hpp"
#include "shift_ops.hpp"

namespace fmmbem {
int fail(int code, const std::string& msg);            // plan.hip
}
using namespace fmmbem;
using namespace fmmbem::tables;

namespace {

constexpr int kMaxSlots = 12;

#define HIP_TRY(expr)                                                                              \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess)                                                                          \
      return fail(e_ == hipErrorOutOfMemory ? FMMBEM_ERR_ALLOC : FMMBEM_ERR_HIP,                  \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                              \
  } while (0)
#define TRY(expr) do { int rc_ = (expr); if (rc_ != FMMBEM_OK) return rc_; } while (0)

struct DeviceScope {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceScope(int dev) {
    err = hipGetDevice(&prev);
    if (err != hipSuccess) { prev = -1; return; }
    if (prev != dev) err = hipSetDevice(dev); else prev = -1;
  }
  ~DeviceScope() { if (prev >= 0) (void)hipSetDevice(prev); }
};

// device memory of one call, freed when the call returns
struct Scratch {
  std::vector<void*> v;
  ~Scratch() { for (void* p : v) (void)hipFree(p); }
  template <class T>
  int up(const T* src, size_t count, const T** dst) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, count ? count * sizeof(T) : 1));
    v.push_back(p);
    if (count) HIP_TRY(hipMemcpy(p, src, count * sizeof(T), hipMemcpyHostToDevice));
    *dst = static_cast<const T*>(p);
    return FMMBEM_OK;
  }
  template <class T>
  int zeros(size_t count, T** dst, hipStream_t s) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, count ? count * sizeof(T) : 1));
    v.push_back(p);
    if (count) HIP_TRY(hipMemsetAsync(p, 0, count * sizeof(T), s));
    *dst = static_cast<T*>(p);
    return FMMBEM_OK;
  }
};

}  // namespace

struct fmmbem_ops {
  int kernel = 0, p_max = 0, device = 0;
  double mu = 1.0;
  QuadRule rule;
  HarmonicTables T;
  DevicePlan base{};                                   // the constant fields: orders, tables, the two boxes' storage
  std::vector<void*> allocs;
  hipStream_t stream = nullptr;
  // index lists of the one pair / the one leaf, uploaded once: ints[...]
  const int* ints = nullptr;
  enum { I_ZERO = 0, I_ONE = 1, I_PTR01 = 2, I_M2LPTR = 4, I_CHILD_BEGIN = 7, I_CHILD_END = 9, I_PARENT = 11, I_CLS = 13, I_COUNT = 16 };
  const double *up_stream = nullptr, *dn_stream = nullptr;
  int shift_stream_off[kRotPmax] = {};
  std::vector<ShiftOpDev> up_ops, down_ops;            // p > 12 only, built at the first call that needs them
  double2 *up_tab = nullptr, *down_tab = nullptr;      // [p_max^2], rewritten per call
  double* rec = nullptr;                               // [8] class record of the call's translation
  double* gtab = nullptr;                              // [g_max]  double-sum M2L: the call's class table
  double2* ztab = nullptr;                             // [p_max]
  double* centers = nullptr;                           // [2][3]
  DevicePlan* d_dev = nullptr;
  bool generic_ready = false;
  ~fmmbem_ops() {
    DeviceScope g(device);
    for (void* p : allocs) (void)hipFree(p);
    if (stream) (void)hipStreamDestroy(stream);
  }
  template <class T_>
  int upload(const std::vector<T_>& v, const T_** out) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(v.size(), 1) * sizeof(T_)));
    allocs.push_back(p);
    if (!v.empty()) HIP_TRY(hipMemcpy(p, v.data(), v.size() * sizeof(T_), hipMemcpyHostToDevice));
    *out = static_cast<const T_*>(p);
    return FMMBEM_OK;
  }
  template <class T_>
  int alloc(size_t count, T_** out) {
    void* p = nullptr;
    HIP_TRY(hipMalloc(&p, std::max<size_t>(count, 1) * sizeof(T_)));
    allocs.push_back(p);
    HIP_TRY(hipMemset(p, 0, std::max<size_t>(count, 1) * sizeof(T_)));
    *out = static_cast<T_*>(p);
    return FMMBEM_OK;
  }
  int slots_p2m() const { return kernel == FMMBEM_KERNEL_STOKES_BEM ? 4 : 2; }
  int init();
  int ensure_generic();
  int set_slots(DevicePlan& d, int n_slots) const;
  int put(double2* dst_box, int n_slots, int p, const double* src) const;     // host [slot][S(p)] -> device box [slot][s_max]
  int get_add(const double2* src_box, int n_slots, int p, double* dst) const; // dst += device box
  int panels(Scratch& sc, DevicePlan& d, size_t n, const double* vertices, const uint8_t* bc) const;
  int shift(int op, int p, int n_slots, const double* src, double* tgt, const double tr[3]);
};

int fmmbem_ops::init() {
  DevicePlan& d = base;
  const int pm = p_max;
  d.p_max = pm; d.s_max = pm * (pm + 1) / 2; d.p2_max = pm * pm; d.y2_max = 4 * pm * pm;
  d.nboxes = 2; d.nleaves = 1;
  d.kernel = kernel; d.dof = kernel == FMMBEM_KERNEL_STOKES_BEM ? 3 : 1; d.mu = mu;
  d.nq = rule.n;
  for (int q = 0; q < rule.n; ++q) d.qw[q] = rule.w[q];
  d.nslots = kMaxSlots;
  d.stokes_velocity_targets = 1; d.stokes_traction_targets = 0;
  HIP_TRY(hipStreamCreateWithFlags(&stream, hipStreamNonBlocking));
  TRY(upload(T.A, &d.tabA)); TRY(upload(T.invA, &d.tabInvA)); TRY(upload(T.pref, &d.tabPref));
  {
    // the recurrences' per-step constants (the same table plan.hip uploads: {pref, c1, c2, 0} per (order, step), m-major)
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
  TRY(alloc((size_t)2 * kMaxSlots * d.s_max, &d.M));
  TRY(alloc((size_t)2 * kMaxSlots * d.s_max, &d.L));
  TRY(alloc((size_t)2 * kMaxSlots * d.s_max, &d.Mh));
  TRY(alloc(6, &centers));
  d.box_center = centers;
  {
    std::vector<int> v(I_COUNT, 0);
    v[I_ONE] = 1;
    v[I_PTR01] = 0; v[I_PTR01 + 1] = 1;
    v[I_M2LPTR] = 0; v[I_M2LPTR + 1] = 0; v[I_M2LPTR + 2] = 1;       // CSR by target box: box 1 holds the one pair
    v[I_CHILD_BEGIN] = 0; v[I_CHILD_BEGIN + 1] = 0;                  // box 1's children: [0, 1)
    v[I_CHILD_END] = 0; v[I_CHILD_END + 1] = 1;
    v[I_PARENT] = 0; v[I_PARENT + 1] = 0;                            // box 1's parent: box 0
    TRY(upload(v, &ints));
  }
  // the one leaf (box 0), the one pair (box 0 -> box 1), the one class (0)
  d.leaf_box = ints + I_ZERO; d.leaf_row0 = ints + I_ZERO;
  d.p2m_leaf = ints + I_ZERO; d.n_p2m = 1;
  d.l2p_leaf = ints + I_ZERO; d.n_l2p = 1; d.l2p_grp = ints + I_PTR01; d.n_l2p_grp = 1;
  d.m2m_parent = ints + I_ONE; d.box_child_begin = ints + I_CHILD_BEGIN; d.box_child_end = ints + I_CHILD_END;
  d.l2l_child = ints + I_ONE; d.box_parent = ints + I_PARENT;
  d.up_cls = ints + I_CLS; d.down_cls = ints + I_CLS;
  d.mh_box = ints + I_ZERO; d.n_mh = 1;
  d.m2l_tgt = ints + I_ONE; d.n_m2l_tgt = 1;
  d.m2l_ptr = ints + I_M2LPTR; d.m2l_src = ints + I_ZERO; d.m2l_cls = ints + I_ZERO;
  d.rot_src = ints + I_ZERO; d.rot_cls = ints + I_ZERO; d.rot_tgt = ints + I_ONE;
  d.rot_item_ptr = ints + I_PTR01; d.n_rot_items = 1;
  d.rot_item_ptr_long = ints + I_PTR01; d.n_rot_items_long = 1;
  d.rot_empty = ints + I_ZERO; d.n_rot_empty = 0;
  TRY(alloc(8, &rec));
  d.rot_cls_rec = rec;
  {
    // constant streams of the rotation kernels, every order they exist for
    std::vector<double> all, ups, dns, one;
    for (int p = 1; p <= kRotPmax; ++p) {
      d.rot_tab_off[p - 1] = (int)all.size();
      build_rot_stream(p, one); all.insert(all.end(), one.begin(), one.end());
      shift_stream_off[p - 1] = (int)ups.size();
      build_rot_stream(p, one, kRotM2M); ups.insert(ups.end(), one.begin(), one.end());
      build_rot_stream(p, one, kRotL2L); dns.insert(dns.end(), one.begin(), one.end());
    }
    TRY(upload(all, &d.rot_tab)); TRY(upload(ups, &up_stream)); TRY(upload(dns, &dn_stream));
  }
  HIP_TRY(hipDeviceSynchronize());                     // the zero-fills ran on the NULL stream, which `stream` does not wait for
  return FMMBEM_OK;
}

// what the orders above the rotation kernels need: the sparse shift operators, the double-sum M2L's maps and class tables
int fmmbem_ops::ensure_generic() {
  if (generic_ready) return FMMBEM_OK;
  DevicePlan& d = base;
  const int pm = p_max;
  const ShiftOps ops = build_shift_ops(pm, T.A, kEps);
  auto up_op = [&](const VOp& v, ShiftOpDev& o) -> int {
    TRY(upload(v.src, &o.src)); TRY(upload(v.y, &o.y)); TRY(upload(v.real, &o.real));
    TRY(upload(v.npiece, &o.npiece)); TRY(upload(v.piece, &o.piece));
    o.T = v.T; o.V = v.V; o.maxp = v.maxp;
    return FMMBEM_OK;
  };
  up_ops.resize(pm); down_ops.resize(pm);
  for (int p = 1; p <= pm; ++p) { TRY(up_op(ops.up_v[p - 1], up_ops[p - 1])); TRY(up_op(ops.down_v[p - 1], down_ops[p - 1])); }
  TRY(alloc((size_t)d.p2_max, &up_tab)); TRY(alloc((size_t)d.p2_max, &down_tab));
  d.up_tab = up_tab; d.down_tab = down_tab;
  d.g_max = m2l_entries(pm);
  TRY(alloc((size_t)d.g_max, &gtab)); TRY(alloc((size_t)pm, &ztab));
  d.m2l_g = gtab; d.m2l_z = ztab;
  std::vector<int32_t> lanes, scat, one;
  for (int p = 1; p <= kPmax; ++p) {
    if (!m2l_lane_map(p, one)) return fail(FMMBEM_ERR_INVALID, "internal: M2L lane dealing failed");
    lanes.insert(lanes.end(), one.begin(), one.end());
    m2l_scatter_map(p, one);
    d.m2l_scat_off[p - 1] = (int)scat.size();
    scat.insert(scat.end(), one.begin(), one.end());
  }
  TRY(upload(lanes, &d.m2l_lane)); TRY(upload(scat, &d.m2l_scat));
  TRY(alloc(1, &d_dev));
  HIP_TRY(hipDeviceSynchronize());
  generic_ready = true;
  return FMMBEM_OK;
}

int fmmbem_ops::set_slots(DevicePlan& d, int n_slots) const {
  if (n_slots < 1 || n_slots > kMaxSlots) return fail(FMMBEM_ERR_INVALID, "n_slots outside [1, 12]");
  d.n_act = n_slots;
  for (int s = 0; s < n_slots; ++s) d.act[s] = s;
  return FMMBEM_OK;
}

int fmmbem_ops::put(double2* dst_box, int n_slots, int p, const double* src) const {
  const int S = p * (p + 1) / 2, SM = base.s_max;
  std::vector<double2> tmp((size_t)n_slots * SM, double2{0, 0});
  for (int s = 0; s < n_slots; ++s)
    for (int i = 0; i < S; ++i) tmp[(size_t)s * SM + i] = double2{src[((size_t)s * S + i) * 2], src[((size_t)s * S + i) * 2 + 1]};
  HIP_TRY(hipMemcpy(dst_box, tmp.data(), tmp.size() * sizeof(double2), hipMemcpyHostToDevice));
  return FMMBEM_OK;
}

int fmmbem_ops::get_add(const double2* src_box, int n_slots, int p, double* dst) const {
  const int S = p * (p + 1) / 2, SM = base.s_max;
  std::vector<double2> tmp((size_t)n_slots * SM);
  HIP_TRY(hipMemcpy(tmp.data(), src_box, tmp.size() * sizeof(double2), hipMemcpyDeviceToHost));
  for (int s = 0; s < n_slots; ++s)
    for (int i = 0; i < S; ++i) {
      dst[((size_t)s * S + i) * 2] += tmp[(size_t)s * SM + i].x;
      dst[((size_t)s * S + i) * 2 + 1] += tmp[(size_t)s * SM + i].y;
    }
  return FMMBEM_OK;
}

// the call's panels as the one leaf of box 0: derived geometry by the plan builder's own fill_panel, SoA of n
int fmmbem_ops::panels(Scratch& sc, DevicePlan& d, size_t n, const double* vertices, const uint8_t* bc) const {
  PanelSoA P;
  try {
    alloc_panels(P, (int64_t)n, rule.n);
  } catch (const std::bad_alloc&) {
    return fail(FMMBEM_ERR_ALLOC, "host allocation failed");
  }
  for (size_t i = 0; i < n; ++i) fill_panel(P, (int64_t)n, (int64_t)i, vertices + 9 * i, rule, bc ? (bc[i] ? 1 : 0) : 0);
  d.n = (int64_t)n;
  TRY(sc.up(P.cx.data(), n, &d.cx)); TRY(sc.up(P.cy.data(), n, &d.cy)); TRY(sc.up(P.cz.data(), n, &d.cz));
  TRY(sc.up(P.nx.data(), n, &d.nx)); TRY(sc.up(P.ny.data(), n, &d.ny)); TRY(sc.up(P.nz.data(), n, &d.nz));
  TRY(sc.up(P.area.data(), n, &d.area)); TRY(sc.up(P.quad.data(), P.quad.size(), &d.quad)); TRY(sc.up(P.vert.data(), P.vert.size(), &d.vert));
  TRY(sc.up(P.bc.data(), n, &d.bc));
  const int rows = (int)n;
  const int* d_rows = nullptr;
  TRY(sc.up(&rows, 1, &d_rows));
  d.leaf_nrows = d_rows;
  return FMMBEM_OK;
}

// op: 0 M2L (M of box 0 -> L of box 1), 1 M2M (M -> M), 2 L2L (L -> L).  tgt += Op(src)
int fmmbem_ops::shift(int op, int p, int n_slots, const double* src, double* tgt, const double tr[3]) {
  if (!src || !tgt || !tr) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (p < 1 || p > p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  DeviceScope guard(device);
  HIP_TRY(guard.err);
  DevicePlan d = base;
  TRY(set_slots(d, n_slots));
  const size_t box = (size_t)kMaxSlots * d.s_max;
  double2* in = op == 2 ? d.L : d.M;
  double2* out = (op == 1 ? d.M : d.L) + box;
  TRY(put(in, n_slots, p, src));
  HIP_TRY(hipMemsetAsync(out, 0, box * sizeof(double2), stream));   // (the L2L kernels add to what the child holds: zero, and the caller's is added on the way out)
  double r8[8];
  rot_record(tr, r8);
  // FMMBEM_OPS_GENERIC=1: the kernels of the orders above 12 at every order (tests)
  const bool force_generic = std::getenv("FMMBEM_OPS_GENERIC") && std::atoi(std::getenv("FMMBEM_OPS_GENERIC")) != 0;
  const bool rot = !force_generic && p <= kRotPmax && m2l_rot_supported(p) && shift_rot_supported(p);
  if (rot) {
    HIP_TRY(hipMemcpy(rec, r8, sizeof(r8), hipMemcpyHostToDevice));
    if (op == 0) HIP_TRY(launch_m2l_rot(d, nullptr, p, stream));
    else {
      RotWork w;
      w.src = ints + I_ZERO; w.cls = ints + I_ZERO; w.tgt = ints + I_ONE; w.item_ptr = ints + I_PTR01; w.n_items = 1;
      w.rec = rec; w.stream = (op == 1 ? up_stream : dn_stream) + shift_stream_off[p - 1];
      if (op == 1) HIP_TRY(launch_m2m_rot(d, w, p, stream)); else HIP_TRY(launch_l2l_rot(d, w, p, stream));
    }
  } else {
    TRY(ensure_generic());
    d = base;                                          // ensure_generic filled in the tables' addresses
    TRY(set_slots(d, n_slots));
    const int pm = p_max;
    const SphHost sp = cart2sph_host(tr);
    std::vector<cplx> h;
    if (op == 0) {
      // Yh[r,c] = i^{|c|} EPS Y[r,c] / A[r,c] = Z^c gh[r,c]: the real part G (evalLocal to order 2 p_max at beta = 0) and the phases Z^m
      const int R = 2 * pm;
      harmonics(T, false, sp.rho, sp.alpha, 0.0, R, h);
      std::vector<double> g((size_t)d.g_max, 0.0);
      for (int r = 0; r < R; ++r)
        for (int cc = 0; cc <= r; ++cc) g[(size_t)r * (r + 1) / 2 + cc] = h[(size_t)r * (r + 1) / 2 + cc].real() * kEps / T.A[r * r + r + cc];
      std::vector<cplx> z((size_t)pm);
      for (int m = 0; m < pm; ++m) z[(size_t)m] = i_pow(m) * std::exp(cplx(0, 1) * double(m * sp.beta));
      HIP_TRY(hipMemcpy(gtab, g.data(), g.size() * sizeof(double), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(ztab, z.data(), z.size() * sizeof(cplx), hipMemcpyHostToDevice));
      HIP_TRY(hipMemcpy(d_dev, &d, sizeof(DevicePlan), hipMemcpyHostToDevice));
      HIP_TRY(launch_mh_prep(d, p, stream));
      HIP_TRY(launch_m2l(d, d_dev, p, stream));
    } else {
      // M2M: evalMultipole(rho, alpha, -beta) of the translation (LaplaceSpherical.hpp:253-254); L2L: (rho, alpha, +beta) (:383-384)
      harmonics(T, true, sp.rho, sp.alpha, op == 1 ? -sp.beta : sp.beta, pm, h);
      std::vector<cplx> tab;
      for (int n = 0; n < pm; ++n)
        for (int m = -n; m <= n; ++m) {
          const cplx y = h[(size_t)n * (n + 1) / 2 + std::abs(m)];
          tab.push_back(m < 0 ? std::conj(y) : y);
        }
      HIP_TRY(hipMemcpy(op == 1 ? up_tab : down_tab, tab.data(), tab.size() * sizeof(cplx), hipMemcpyHostToDevice));
      if (op == 1) HIP_TRY(launch_m2m_level(d, up_ops[p - 1], p, 0, 1, stream));
      else HIP_TRY(launch_l2l_level(d, down_ops[p - 1], p, 0, 1, stream));
    }
  }
  HIP_TRY(hipStreamSynchronize(stream));
  return get_add(out, n_slots, p, tgt);
}

// ============================================ C ABI ============================================
extern "C" {

int fmmbem_ops_create(const fmmbem_options* opts, fmmbem_ops** out) {
  if (!opts || !out) return fail(FMMBEM_ERR_INVALID, "null argument");
  *out = nullptr;
  if (opts->kernel != FMMBEM_KERNEL_LAPLACE_BEM && opts->kernel != FMMBEM_KERNEL_STOKES_BEM)
    return fail(FMMBEM_ERR_UNSUPPORTED, "unknown kernel id");
  if (opts->p_max < 1 || opts->p_max > kPmax) return fail(FMMBEM_ERR_INVALID, "p_max outside [1, 16]");
  if (opts->kernel == FMMBEM_KERNEL_STOKES_BEM && !(opts->mu > 0)) return fail(FMMBEM_ERR_INVALID, "Stokes: viscosity mu must be positive");
  std::unique_ptr<fmmbem_ops> o(new (std::nothrow) fmmbem_ops);
  if (!o) return fail(FMMBEM_ERR_ALLOC, "ops");
  if (!quad_rule(opts->quad_k, o->rule)) return fail(FMMBEM_ERR_INVALID, "invalid quadrature key (valid: 1 3 4 7 13 17 19 25 79)");
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
    return fail(FMMBEM_ERR_NO_DEVICE, "no HIP device visible (this library has no CPU execution path)");
  if (opts->device < 0 || opts->device >= ndev) return fail(FMMBEM_ERR_INVALID, "device ordinal out of range");
  o->kernel = opts->kernel; o->p_max = opts->p_max; o->device = opts->device; o->mu = opts->mu;
  DeviceScope guard(o->device);
  HIP_TRY(guard.err);
  try {
    TRY(o->init());
  } catch (const std::bad_alloc&) {
    return fail(FMMBEM_ERR_ALLOC, "host allocation failed while tabulating operators");
  }
  *out = o.release();
  return FMMBEM_OK;
}

void fmmbem_ops_destroy(fmmbem_ops* ops) { delete ops; }

int fmmbem_ops_slots(const fmmbem_ops* ops) { return ops ? ops->slots_p2m() : 0; }

int fmmbem_ops_p2m(fmmbem_ops* ops, int p, size_t n, const double* vertices, const uint8_t* bc, const double* charges,
                   const double center[3], double* M) {
  if (!ops || !vertices || !charges || !center || !M) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (p < 1 || p > ops->p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  if (n == 0) return FMMBEM_OK;
  if (n > ((size_t)1 << 24)) return fail(FMMBEM_ERR_INVALID, "too many panels for one expansion");
  const bool stokes = ops->kernel == FMMBEM_KERNEL_STOKES_BEM;
  if (stokes && bc)
    for (size_t i = 0; i < n; ++i)
      if (bc[i]) return fail(FMMBEM_ERR_UNSUPPORTED, "Stokes P2M of a TRACTION source: the reference's stresslet moments (kernel/StokesSphericalBEM.hpp:438-466) "
                                                     "are not what this library's far field uses (include/fmmbem.h, FMMBEM_KERNEL_STOKES_BEM)");
  DeviceScope guard(ops->device);
  HIP_TRY(guard.err);
  try {
    Scratch sc;
    DevicePlan d = ops->base;
    TRY(ops->set_slots(d, ops->slots_p2m()));
    TRY(ops->panels(sc, d, n, vertices, bc));
    const double* xt = nullptr;
    TRY(sc.up(charges, n * d.dof, &xt));
    d.xt = const_cast<double*>(xt);
    HIP_TRY(hipMemcpy(ops->centers, center, 3 * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(hipMemsetAsync(d.M, 0, (size_t)kMaxSlots * d.s_max * sizeof(double2), ops->stream));
    if (stokes) HIP_TRY(launch_p2m_stokes(d, p, ops->stream)); else HIP_TRY(launch_p2m(d, p, ops->stream));
    HIP_TRY(hipStreamSynchronize(ops->stream));
    return ops->get_add(d.M, ops->slots_p2m(), p, M);
  } catch (const std::bad_alloc&) {
    return fail(FMMBEM_ERR_ALLOC, "host allocation failed");
  }
}

int fmmbem_ops_l2p(fmmbem_ops* ops, int p, const double* L, const double center[3], size_t n, const double* vertices,
                   const uint8_t* bc, double* result) {
  if (!ops || !L || !center || !vertices || !result) return fail(FMMBEM_ERR_INVALID, "null argument");
  if (p < 1 || p > ops->p_max) return fail(FMMBEM_ERR_INVALID, "p outside [1, p_max]");
  if (n == 0) return FMMBEM_OK;
  if (n > ((size_t)1 << 24)) return fail(FMMBEM_ERR_INVALID, "too many panels for one expansion");
  const bool stokes = ops->kernel == FMMBEM_KERNEL_STOKES_BEM;
  if (stokes && bc)
    for (size_t i = 0; i < n; ++i)
      if (bc[i]) return fail(FMMBEM_ERR_UNSUPPORTED, "Stokes L2P at a TRACTION target reads the double layer's seven potentials, which the reference's "
                                                     "local_type does not hold (include/fmmbem.h, FMMBEM_KERNEL_STOKES_BEM)");
  DeviceScope guard(ops->device);
  HIP_TRY(guard.err);
  try {
    Scratch sc;
    DevicePlan d = ops->base;
    TRY(ops->set_slots(d, ops->slots_p2m()));
    TRY(ops->panels(sc, d, n, vertices, bc));
    double* yt = nullptr;
    TRY(sc.zeros(n * d.dof, &yt, ops->stream));
    HIP_TRY(hipMemcpy(ops->centers, center, 3 * sizeof(double), hipMemcpyHostToDevice));
    TRY(ops->put(d.L, ops->slots_p2m(), p, L));
    if (stokes) HIP_TRY(launch_l2p_stokes(d, p, yt, ops->stream)); else HIP_TRY(launch_l2p(d, p, yt, ops->stream));
    HIP_TRY(hipStreamSynchronize(ops->stream));
    std::vector<double> y(n * d.dof);
    HIP_TRY(hipMemcpy(y.data(), yt, y.size() * sizeof(double), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < y.size(); ++i) result[i] += y[i];
    return FMMBEM_OK;
  } catch (const std::bad_alloc&) {
    return fail(FMMBEM_ERR_ALLOC, "host allocation failed");
  }
}

int fmmbem_ops_m2m(fmmbem_ops* ops, int p, int n_slots, const double* M_source, double* M_target, const double translation[3]) {
  if (!ops) return fail(FMMBEM_ERR_INVALID, "null argument");
  try { return ops->shift(1, p, n_slots, M_source, M_target, translation); }
  catch (const std::bad_alloc&) { return fail(FMMBEM_ERR_ALLOC, "host allocation failed"); }
}
int fmmbem_ops_m2l(fmmbem_ops* ops, int p, int n_slots, const double* M_source, double* L_target, const double translation[3]) {
  if (!ops) return fail(FMMBEM_ERR_INVALID, "null argument");
  try { return ops->shift(0, p, n_slots, M_source, L_target, translation); }
  catch (const std::bad_alloc&) { return fail(FMMBEM_ERR_ALLOC, "host allocation failed"); }
}
int fmmbem_ops_l2l(fmmbem_ops* ops, int p, int n_slots, const double* L_source, double* L_target, const double translation[3]) {
  if (!ops) return fail(FMMBEM_ERR_INVALID, "null argument");
  try { return ops->shift(2, p, n_slots, L_source, L_target, translation); }
  catch (const std::bad_alloc&) { return fail(FMMBEM_ERR_ALLOC, "host allocation failed"); }
}

}  // extern "C"
