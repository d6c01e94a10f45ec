// device_launch.hpp -- prototypes of the kernel launchers (kernels_near.hip, kernels_far.hip, kernels_m2l.hip and the three
// builds of kernels_m2l_rot.hip).  Kept apart from device_plan.hpp (the structs the kernels take by value) so that adding a
// launcher does not rebuild the rotation kernels, four minutes each.
#pragma once
#include "device_plan.hpp"
#include "shift_lanes.hpp"

namespace fmmbem {

// ---- launchers (kernels_near.hip / kernels_far.hip); all asynchronous on `s` ----
hipError_t launch_near_assemble(const DevicePlan& d, hipStream_t s);
hipError_t launch_gather_x(const DevicePlan& d, const double* x, hipStream_t s);
hipError_t launch_near_spmv(const DevicePlan& d, hipStream_t s);
struct HybridStreams { hipStream_t recompute = nullptr; hipEvent_t fork = nullptr, join_recompute = nullptr; };
hipError_t launch_near_hybrid(const DevicePlan& d, hipStream_t s, const HybridStreams& hs);   // near_stream_fraction < 1
hipError_t launch_kernel_entries(const DevicePlan& d, int m, double* out, hipStream_t s);   // panels [0,m) targets, [m,2m) sources
hipError_t launch_panel_setup(int64_t n, const uint32_t* perm, const double* v_orig, int nq, const double* pts, double* cx, double* cy,
                              double* cz, double* nx, double* ny, double* nz, double* area, double* quad, double* vert, hipStream_t s);
int rcg_item_rows();
hipError_t launch_rc_pack(const DevicePlan& d, double* rc_src, double* rc_nrm, hipStream_t s);
hipError_t launch_near_row_eval(const DevicePlan& d, int64_t prow, const int* cols, int n, double* out, hipStream_t s);
hipError_t launch_near_diag(const DevicePlan& d, const int* selfcol, double* out, hipStream_t s);
hipError_t launch_near_matfree(const DevicePlan& d, hipStream_t s);
hipError_t launch_mf_side(const DevicePlan& d, int phase, int* side_cnt, const int64_t* side_ptr, int* side_col, const int* side_row,
                          double* side_val, int64_t nside, hipStream_t s);
hipError_t launch_scatter_y(const DevicePlan& d, double* y, hipStream_t s);
hipError_t launch_assemble_slices(const DevicePlan& d, const double* slices, double* y, int world, const int64_t* d_cut, int64_t chunk,
                                  hipStream_t s);
hipError_t launch_p2m(const DevicePlan& d, int p, hipStream_t s);
hipError_t launch_p2m_table(const DevicePlan& d, double2* tab, hipStream_t s);
hipError_t launch_p2m_table_grad(const DevicePlan& d, double2* tab, hipStream_t s);     // one-off: fills DevicePlan::p2m_tab's storage
hipError_t launch_m2m_level(const DevicePlan& d, const ShiftOpDev& op, int p, int first, int count, hipStream_t s);
// multipoles of the boxes a shard owns -> send buffer [idx][active slot][S(p)]; all shards' buffers -> M (own slice skipped)
hipError_t launch_xch_pack(const DevicePlan& d, int p, double2* send, hipStream_t s);
hipError_t launch_xch_unpack(const DevicePlan& d, int p, const double2* recv, hipStream_t s);
hipError_t launch_mh_prep(const DevicePlan& d, int p, hipStream_t s);
// d_dev: a copy of d in device memory (the M2L kernel re-reads the plan fields it needs with scalar loads every source
// instead of keeping them alive in SGPRs across its FMA region)
hipError_t launch_m2l(const DevicePlan& d, const DevicePlan* d_dev, int p, hipStream_t s);
bool m2l_rot_supported(int p);
bool shift_rot_supported(int p);
hipError_t launch_m2m_rot(const DevicePlan& d, const RotWork& w, int p, hipStream_t s);   // M[tgt = parent] = sum over its children
hipError_t launch_l2l_rot(const DevicePlan& d, const RotWork& w, int p, hipStream_t s);   // L[tgt = child] += shift of L[src = parent]
hipError_t launch_m2l_rot(const DevicePlan& d, const DevicePlan* d_dev, int p, hipStream_t s);
bool m2l_rot_long_items(int p);
// kernels_shift.hip: the tree passes with one pair per wavefront (lane = coefficient); bit for bit the one-pair-per-lane kernels' results
bool shift_lanes_supported(int p);
hipError_t launch_m2m_lanes(const DevicePlan& d, const ShiftLaneWork& w, int p, hipStream_t s);
hipError_t launch_l2l_lanes(const DevicePlan& d, const ShiftLaneWork& w, int p, hipStream_t s);
// bit p - 1: orders of the four rotation objects that carry "s_nop 1" in front of their DPP FMAs (csrc/Makefile ROTBUILD)
unsigned rot_nop_orders_m2l();
unsigned rot_nop_orders_m2m();
unsigned rot_nop_orders_l2l();
int l2p_group_leaves(int kernel);                      // most leaves an L2P work group may hold (the kernels' LDS slice per wavefront)
hipError_t launch_m2l_rot_zero(const DevicePlan& d, int p, hipStream_t s);
hipError_t launch_l2l_level(const DevicePlan& d, const ShiftOpDev& op, int p, int first, int count, hipStream_t s);
hipError_t launch_l2p(const DevicePlan& d, int p, double* y, hipStream_t s);
hipError_t launch_near_assemble_stokes(const DevicePlan& d, hipStream_t s);
hipError_t launch_p2m_stokes(const DevicePlan& d, int p, hipStream_t s);
hipError_t launch_l2p_stokes(const DevicePlan& d, int p, double* y, hipStream_t s);


}  // namespace fmmbem
