// device_plan.hpp -- HBM-resident form of a plan and the launch interface of the gfx950 kernels.
//
// HBM layout (all arrays uploaded once at plan creation; DESIGN.md "Data layout"):
//   panels      SoA in TREE order: cx,cy,cz,nx,ny,nz,area [N]; quad [q][xyz][N]; vert [9][N]; bc [N] u8
//   perm        [N] u32                      tree index -> original index
//   leaves      leaf_row0/leaf_nrows [nl]; near_ptr [nl+1] -> near_run_row0/near_run_off (runs of consecutive
//               source rows: adjacent source leaves are merged; off = first column of the run);
//               near_ncols/near_stride [nl]; near_off [nl] (offset of the leaf's row block in near_val)
//   near_val    per owned target leaf a dense row-major block  nrows x stride  (stride = ncols rounded
//               up to even => every row 16-B aligned); the ONLY large array (8 B per near entry)
//   boxes       center [nb][3]
//   M, L        [nb][nslots][S_max] complex (S = p(p+1)/2); Laplace: slot 0 = G, slot 1 = dG/dn;
//               Stokes: slots 0-3 = the four harmonic potentials (f0,f1,f2,f.x) of the velocity group
//   Mh          [nb][2][S_max] complex: M rescaled and phase-rotated (orders m >= 0), the M2L input
//   m2l_*       CSR by target box: m2l_tgt [nt] (boxes to run), m2l_ptr [nb+1], m2l_src, m2l_cls
//   class tabs  m2l_g [classes][p_max(2 p_max+1)] real + m2l_z [classes][p_max] complex; up_tab/down_tab [classes][p_max^2] complex
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace fmmbem {

// One unit of near_spmv work with everything the pipelined kernel needs in one record (fetched two items ahead by
// scalar loads): rows [yrow, yrow+nrows) (in unknowns) of one target leaf's dense block.
struct NearItem {
  int64_t val_off;     // first entry of the row range in near_val
  int64_t run_begin;   // -> near_run_row0 / near_run_off
  int nruns;
  int yrow;            // first tree-order unknown of the range (index into yt)
  int nrows;
  int ncols, stride;   // columns in unknowns, row stride
  int colsplit;        // nrows < 8: wavefronts split the columns instead of the rows
  int pad[2];
};
static_assert(sizeof(NearItem) == 48, "NearItem is read as three 16-byte pieces");

// One unit of work of the recompute kernel (hybrid plans): rows [prow0, prow0 + nrows), nrows <= 20, of one recomputed leaf
// against the leaf's ncp source panels; self-contained like NearItem
struct RcItem {
  int prow0, nrows, ncp, nruns;
  int64_t run_begin;   // -> near_run_row0 / near_run_off
  int leaf, pad;
};
static_assert(sizeof(RcItem) == 32, "RcItem is read as two 16-byte pieces");

// M2M or L2L operator at one order p, terms dealt to virtual rows (shift_ops.hpp VOp); passed to the kernel by value.
struct ShiftOpDev {
  const uint16_t *src, *y;     // [i * V + v]
  const double* real;
  const uint16_t *npiece;      // [row]
  const uint16_t *piece;       // [k * S + row] -> v
  int T, V, maxp;
};

struct DevicePlan {
  int64_t n = 0;
  int nq = 0;
  int nboxes = 0, nleaves = 0;
  int p_max = 0, s_max = 0, p2_max = 0, y2_max = 0;   // S, P^2, (2P)^2 at p_max
  int p2m_stride = 0;                                 // double2 per record of p2m_tab: s_max rounded up to a cache line (classic), or the packed length
  // Laplace records are PACKED: the moments of order m = 0 are real (e^{i m beta} = 1; also their normal-derivative form), so a
  // record is [complex (n, m >= 1): n = 1 .., m = 1 .. n][real (n, 0): n = 0 ..] -- 16 p(p-1)/2 + 8 p bytes instead of 16 p(p+1)/2,
  // 800 instead of 880 (896 with the line padding) at p_max = 10; order p still reads a prefix of either segment.
  // p2m_real_off: double2 slot at which the reals begin (= p_max (p_max - 1) / 2); the Stokes tables keep the classic layout
  int p2m_packed = 0, p2m_real_off = 0;
  int leaf_begin = 0, leaf_end = 0;                   // owned target leaves
  int64_t row_begin = 0, row_end = 0;
  int max_ncols = 0;                                  // widest near row block (columns, padded even)
  int max_runs = 0;                                   // most source runs of any owned target leaf
  int n_act = 0;                                      // active expansion slots
  int act[12] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
  int nslots = 2;                                     // expansions per box: Laplace 2 (G, dG/dn), Stokes 2 x 4
  int kernel = 0;                                     // fmmbem_kernel
  int dof = 1;                                        // unknowns per panel (Stokes: 3, interleaved)
  double mu = 1.0;                                    // Stokes viscosity
  int nqf = 0;                                        // Stokes near-regime rule K_fine: barycentric points + weight
  const double* qf = nullptr;                         // [nqf][4] in device memory (by value it would not fit the kernel arguments at K_fine = 79)

  // panels
  const double *cx, *cy, *cz, *nx, *ny, *nz, *area, *quad, *vert;
  const uint8_t* bc;
  const uint32_t* perm;
  double qw[79];                                      // weights of the far-regime rule K (at most FMMBEM_MAX_QUAD points)
  // leaves
  const int *leaf_row0, *leaf_nrows, *leaf_box;
  const int64_t* near_ptr;
  const int *near_run_row0, *near_run_off;            // runs of consecutive source rows: first row, column offset
  const int *near_ncols, *near_stride;
  const int64_t* near_off;
  double* near_val;
  const int* xch_box;                                // sharded upward pass: private need_M boxes of every shard, by shard
  int xch_ptr[9];                                    // ... xch_box[xch_ptr[r] .. xch_ptr[r+1]) belongs to shard r (<= 8 shards)
  int xch_rank, xch_world, xch_max;                  // this shard, number of shards, largest per-shard count
  // shard_upward == 2 (selective exchange): the boxes this shard sends (all destinations, one after the other) and receives
  const int *xsel_send_box = nullptr, *xsel_recv_box = nullptr;
  int xsel_send_n = 0, xsel_recv_n = 0;
  const int4* near_items;                            // SpMV work items {leaf, first row, rows, column-split?}, largest first
  const NearItem* near_recs;                         // the same items as self-contained records (pipelined kernel)
  int near_nitems;
  // Stokes: the 3x3 block of a panel pair is symmetric ((delta_ij/r + x_i x_j/r^3) summed over the quadrature, the
  // products commute bit for bit), so the SpMV streams 6 values per pair instead of 9: per target PANEL row three planes
  // of ncols 16-byte pairs (xx,xy) (xz,yy) (yz,zz), written by the assembly; near_val is then not allocated.
  double* near_sym = nullptr;
  const int64_t* near_sym_off;                        // [nl] offset of a leaf's block in near_sym (doubles)
  const int4* sym_items;      int sym_nitems = 0;     // {leaf, first panel row, panel rows, column-split?}, largest first
  // Hybrid near field (near_stream_fraction < 1): near_rec[leaf] != 0 -> the leaf keeps no matrix block, its far-regime entries are
  // recomputed every matvec (near_recompute3_kernel over rc_items) and its near-regime pairs come from the side list; the other
  // leaves are streamed as ever (sym_items).  near_items = the recompute items as {leaf, first row, rows, 0} for the side listing
  const uint8_t* near_rec = nullptr;
  const RcItem* rc_items = nullptr;  int rc_nitems = 0;
  int near_nitems_stream = 0;                        // Laplace hybrid plans: the streamed items are near_recs[0 .. near_nitems_stream)
  // boxes / expansions
  const double* box_center;
  double2 *M, *L, *Mh;
  // tables
  const double *tabA, *tabInvA, *tabPref;             // [n^2+n+m], n < 2*kPmax
  const double *tabStep;                              // [p-1][step][4] = {pref, c1, c2, 0} of the harmonic recurrences at order p, m-major steps
  // far-field lists
  const int *p2m_leaf;        int n_p2m = 0;          // box ids
  // matrix-free near field (sparse_local = 0): the pairs of the near regimes (semi-analytic / fine-rule / self entries), evaluated
  // ONCE at plan creation; CSR by tree-order target row, columns ascending.  side_val: 1 double per entry (Laplace) or 9 (Stokes)
  const int64_t* side_ptr = nullptr;
  const int* side_col = nullptr;
  const double* side_val = nullptr;
  const int *l2p_leaf;        int n_l2p = 0;
  const int *l2p_grp;         int n_l2p_grp = 0;      // L2P work: consecutive l2p_leaf entries [grp[i], grp[i+1]) that fill one wavefront
  const int *m2m_parent;      const int* box_child_begin; const int* box_child_end;
  const int *l2l_child;       const int* box_parent;
  const int *up_cls, *down_cls;                       // per box: class of (parent-child) / (child-parent)
  const double2 *up_tab, *down_tab;                   // [cls][p2_max]
  // M2M / L2L as sparse operators in ELL form (shift_ops.hpp), rows = stored coefficient index at p_max
  const int *mh_box;          int n_mh = 0;           // boxes whose Mh is needed (M2L sources)
  const int *m2l_tgt;         int n_m2l_tgt = 0;
  const int *m2l_ptr, *m2l_src, *m2l_cls;
  const double* m2l_g;                                // [cls][g_max] real class tables G[r,a], entry r(r+1)/2+a, r < 2 p_max
  const double2* m2l_z;                               // [cls][p_max] class phases Z^m, Z = i e^{i beta}
  int g_max;                                          // p_max (2 p_max + 1)
  const int *m2l_lane;                                // [p-1][192] lane -> output map (m2l_layout.hpp)
  const int *m2l_scat;                                // per p: table entry -> its 4 LDS places (m2l_layout.hpp)
  int m2l_scat_off[16];                               // offset of order p's scatter map in m2l_scat
  // M2L by rotation / axial translation / rotation (kernels_m2l_rot.hip, m2l_rot.hpp): the owned pairs in CSR order by
  // target, cut into items (whole targets with <= 64 pairs in all, or one target with more); per class
  // {1/rho, cos alpha, sin alpha, cos beta, sin beta, 0, 0, 0}; per order p the constants in consumption order
  const int *rot_src = nullptr, *rot_cls = nullptr, *rot_tgt = nullptr, *rot_item_ptr = nullptr;
  int n_rot_items = 0;
  const int* rot_item_ptr_long = nullptr;  int n_rot_items_long = 0;   // the same pairs in long items, for the orders at one wavefront per SIMD
  const int* rot_empty = nullptr;   int n_rot_empty = 0;      // targets without a source of their own: L = 0
  const double* rot_cls_rec = nullptr;
  const double* rot_tab = nullptr;  int rot_tab_off[12] = {};
  // P2M as a precomputed operator: the multipole of a leaf is linear in the charges, M = sum_panels x_i * T_i with the
  // panel's own moments T_i = sum_q w_q A_i Ynm(q - c_leaf) (G or, for NORMAL_DERIV panels, gradient moments) independent
  // of x.  [panel (tree order)][ntab][p2m_stride] complex, ntab = 1 (Laplace) or 4 (Stokes: moments of 1, x_q, y_q, z_q);
  // built once at p_max (kernels_far.hip p2m_table); the coefficients of order p are a prefix of every record.
  const double2* p2m_tab = nullptr;
  // Stokes double layer (TRACTION targets): [panel][3][p2m_stride], the components of sum_q w_q A grad(rho^n Ynm)
  const double2* p2m_tab_g = nullptr;
  int stokes_velocity_targets = 1, stokes_traction_targets = 0;   // which groups of Stokes expansions are live (slots 0..3 / 4..10)
  int64_t p2m_tab_row0 = 0;                            // tree-order panel of the table's first record (a shard that runs P2M on
                                                      // its own leaves only keeps only their records)
  // scratch
  double *xt, *yt;                                    // tree-order x and near result
  const double *rc_src = nullptr, *rc_nrm = nullptr;  // hybrid Stokes plans: packed per-panel records of near_recompute3g_kernel ([n][16], [n][4])
  double* xt4 = nullptr;                              // ... and the tree-order charges padded to 4 doubles per panel (gather_x)
  double* ys = nullptr;                               // hybrid plans: the listed entries' sums of the recomputed rows (tree order)
  const int4* side_items = nullptr;  int side_nitems = 0;   // ... work of near_side_items: {first entry, one past last, first row, one past last}
};

// One launch of the rotation kernel (kernels_m2l_rot.hip, compiled per operator): pairs (source box, class, target box) sorted by
// target, cut into items of whole targets; class records of 8 doubles (1/rho, cos a, sin a, cos b, sin b, rho); the constant
// stream of the operator at this order (m2l_rot.hpp).
struct RotWork {
  const int *src = nullptr, *cls = nullptr, *tgt = nullptr, *item_ptr = nullptr;
  int n_items = 0;
  const double* rec = nullptr;
  const double* stream = nullptr;
};
constexpr int kShiftRotPmin = 1;                     // M2M / L2L by rotation from this order up

}  // namespace fmmbem
