// host_plan.hpp -- host side of a plan: panels, Morton octree, dual tree traversal and the
// flattened (CSR-like) operator lists that are uploaded once to HBM.
//
// What it reproduces from the reference (all host-only there as well):
//   * Panel geometry                 kernel/LaplaceSphericalBEM.hpp:64-97
//   * Octree                         include/tree/Octree.hpp:67-79, 118-129, 617-692, 334-355
//   * MAC                            include/FMMOptions.hpp:21-31
//   * dual traversal + lazy lists    include/executor/EvalInteractionLazySparse.hpp:68-115, 173-252
//   * near-matrix sparsity           include/executor/EvalP2P.hpp:47-98
// The layout is this build's own: everything is a flat array keyed by BFS box index or by
// tree-order panel index, grouped by TARGET so that one wavefront owns one target.
#pragma once
#include <cstdint>
#include <functional>
#include <string>
#include <vector>
#include <utility>
#include <memory>

namespace fmmbem {

constexpr int kPmax = 16;
constexpr int kMaxQuad = 79;       // FMMBEM_MAX_QUAD: the largest rule of examples/BEM/GaussQuadrature.hpp

// Fan-out of the plan builder's host work: body(t) for t = 0 .. nt - 1, t = 0 on the caller.  A few worker threads are kept for
// the life of the process (a plan build fans out ~40 times; starting sixteen threads each time cost more than some of the phases
// they shared: the octree 14 -> 8 ms at N = 1M); when the pool is busy (plans built concurrently by the shards of a device list),
// after a fork(), or for nt beyond the pool, plain threads are started as before.
void host_parallel(int nt, const std::function<void(int)>& body);

struct QuadRule {            // triangle Gauss rule: barycentric points + weights
  int n = 0;
  double pts[kMaxQuad][3];
  double w[kMaxQuad];
};
// examples/BEM/GaussQuadrature.hpp:19-185; false for an unknown key (the reference exits, :279-284)
bool quad_rule(int key, QuadRule& out);

// an exact integer translation vector (half-cells of the finest tree level) as the key of a translation class
struct IVec3 {
  int32_t x, y, z;
  bool operator==(const IVec3& o) const { return x == o.x && y == o.y && z == o.z; }
};
struct IVec3Hash {
  size_t operator()(const IVec3& v) const {
    uint64_t h = (uint64_t)(uint32_t)v.x * 0x9E3779B97F4A7C15ull;
    h ^= ((uint64_t)(uint32_t)v.y + 0x7F4A7C15ull) * 0xC2B2AE3D27D4EB4Full + (h << 6) + (h >> 2);
    h ^= ((uint64_t)(uint32_t)v.z + 0x165667B1ull) * 0xD6E8FEB86659FD93ull + (h << 6) + (h >> 2);
    return (size_t)(h ^ (h >> 29));
  }
};

struct HostOptions {
  // called by HostPlan::build once the tree, the leaves, the near lists, the shard's ranges and the flags (panels_on_device)
  // are final and before the far-field lists are built: the caller uploads them and starts the near-matrix assembly on the GPU
  // while the host goes on.  A non-empty string aborts the build with that text.
  std::function<std::string()> after_near_lists;
  int p_max = 10;
  int quad_k = 3;
  double theta = 0.5;
  unsigned ncrit = 64;
  int shard_rank = 0, shard_world = 1;
  int evaluator = 0;           // 0 FMM, 1 local only, 2 block diagonal (executor/make_executor.hpp:24-60)
  bool reference_l2l = false;  // keep only the parent->child L2L edges the reference's lazy rule keeps (see host_plan.cpp)
  int shard_upward = 0;        // shard_world > 1: 1, 2 = P2M/M2M only for boxes this shard owns (+ the few spanning shards); 2 = selective exchange lists
  bool panels_on_device = false;  // the caller derives the panels' geometry (centroid, normal, area, points) on the device from the
                               // vertices and the permutation (kernels_near.hip panel_setup): build() then fills only the flags
};

// Panels in TREE order, structure-of-arrays (what the kernels stream).
// std::vector whose resize() leaves the new elements uninitialised: the 240 MB of panel arrays at N = 1M are written
// exactly once, by several threads, and a zero-fill in front of that costs as much as the fill itself
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  template <class U> void construct(U* p) noexcept { ::new (static_cast<void*>(p)) U; }
  template <class U, class... A> void construct(U* p, A&&... a) { ::new (static_cast<void*>(p)) U(std::forward<A>(a)...); }
};
template <class T> using raw_vector = std::vector<T, default_init_allocator<T>>;

struct PanelSoA {
  raw_vector<double> cx, cy, cz;          // centroid
  raw_vector<double> nx, ny, nz;          // unit normal
  raw_vector<double> area;
  raw_vector<double> quad;                // [q][xyz][N]  stored quadrature points
  raw_vector<double> vert;                // [vertex*3+xyz][N]
  raw_vector<uint8_t> bc;
};

void alloc_panels(PanelSoA& P, int64_t n, int nq);
// one panel's derived geometry (kernel/LaplaceSphericalBEM.hpp:64-97) into slot i of arrays sized for n panels
void fill_panel(PanelSoA& P, int64_t n, int64_t i, const double* vertices9, const QuadRule& rule, uint8_t bc_flag);

struct HostPlan {
  HostOptions opt;
  int64_t n = 0;
  QuadRule rule;

  // ---- tree ----
  double pmin[3], cell[3];
  std::vector<uint32_t> perm;             // tree index -> original index
  int nboxes = 0, nlevels = 0;
  std::vector<int> level_off;             // nlevels+1, boxes of one level are contiguous (BFS order)
  std::vector<uint64_t> box_key;          // marker-bit Morton key
  int tree_levels_max = 10;               // 10: the reference's 32-bit coder built the tree; 21: the 64-bit coder had to
  std::vector<int> box_level, box_parent, box_child_begin, box_child_end;   // children empty for leaves
  std::vector<uint8_t> box_leaf;
  std::vector<int> box_body_begin, box_body_end;
  std::vector<double> box_center;         // [box][3]
  std::vector<double> box_side;
  std::vector<int32_t> box_icoord;        // [box][3] centre in half-finest-cell units (exact integers)

  // ---- raw traversal output (reference order) ----
  std::vector<int> p2p_src, p2p_tgt;      // leaf pairs
  std::vector<int> lr_src, lr_tgt;        // M2L pairs

  // ---- leaves ----
  std::vector<int> leaf_box;              // leaf index -> box, ascending body_begin (= tree order)
  std::vector<int> box_leaf_index;        // box -> leaf index or -1

  // ---- near field, grouped by target leaf ----
  std::vector<int64_t> near_ptr;          // nleaves+1 -> near_src
  std::vector<int> near_src;              // source LEAF indices, ascending body_begin
  std::vector<int> near_ncols;            // per target leaf: total columns
  int64_t near_nnz_total = 0;

  // ---- far field ----
  std::vector<uint8_t> need_M, has_L;     // per box
  std::vector<int> p2m_leaves, l2p_leaves;          // box ids
  std::vector<int> m2l_ptr;               // nboxes+1, by target box
  std::vector<int> m2l_src;               // source box
  std::vector<int> m2l_cls;               // translation class of the pair
  // work list of the rotation M2L kernel (kernels_m2l_rot.hip): the owned pairs, whole targets packed into ITEMS of at most
  // 64 pairs (one wavefront pass, lane = pair); a target with more than 64 sources is an item of its own (several passes)
  std::vector<int> rot_src, rot_cls, rot_tgt, rot_item_ptr, rot_empty;   // rot_empty: owned targets with no source at all
  bool rot_alias = false;              // the pair list IS m2l_src / m2l_cls (targets by m2l_ptr): rot_src / rot_cls / rot_tgt stay empty
  int64_t rot_passes = 0;
  // the same pairs cut into LONG items for the orders that run one wavefront per SIMD (p >= 9): there the chip holds 1 024
  // wavefronts at a time, the first pass of an item stands in the open, and two even rounds of long items beat seven of short ones
  std::vector<int> rot_item_ptr_long;
  int64_t rot_passes_long = 0;
  void build_rot_items();
  // cut a run of targets (pairs per target in seg_len) into items of about `nominal` pairs; appends the boundaries, as
  // positions in the pair list starting at pair_base, to item_ptr (n_items + 1 entries); returns the 64-pair passes
  static int64_t cut_rot_items(const std::vector<int>& seg_len, int nominal, int pair_base, std::vector<int>& item_ptr);
  std::vector<int32_t> m2l_class_vec;     // [class][3] integer translation (target - source), half-finest-cell units
  std::vector<int> m2l_class_rep;         // [class][2] representative (src,tgt) pair
  std::vector<int> m2m_parents;           // parents with need_M, deepest level first; m2m_level_ptr delimits levels
  std::vector<int> m2m_level_ptr;
  std::vector<int> l2l_children;          // children receiving L2L, top level first
  std::vector<int> l2l_level_ptr;
  int64_t m2m_ops = 0, l2l_ops = 0;
  int64_t l2l_ref_omitted = 0;            // edges (parent holds L, child too) the reference's lazy rule never queues
  std::vector<uint8_t> l2l_ref_edge;      // per box: the reference queues L2L parent(b) -> b

  // ---- shard (multi-GPU partition by target leaf) ----
  int leaf_begin = 0, leaf_end = 0;       // owned leaves
  int64_t row_begin = 0, row_end = 0;     // owned tree-order rows
  std::vector<uint8_t> owned_L;           // box is an owned leaf or an ancestor of one
  // upward pass sharded by owner (opt.shard_upward): a box is private to the shard whose rows contain all its bodies,
  // else shared.  p2m_leaves / m2m_parents[0 .. m2m_level_ptr.back()) then hold THIS shard's private boxes,
  // m2m_parents[m2m_shared_ptr[l] .. m2m_shared_ptr[l+1]) the shared parents of one level (deepest first), and
  // xch_box[xch_ptr[r] .. xch_ptr[r+1]) the private need_M boxes of shard r (what it sends in the all-gather).
  std::vector<int> m2m_shared_ptr, xch_ptr, xch_box;
  // shard_upward == 2: the same exchange, but a shard sends another only the boxes that shard's lists READ (the sources of the
  // M2L pairs of its targets, and the private children of the parents every shard translates): xsel_send_box[xsel_send_ptr[q]
  // .. xsel_send_ptr[q+1]) go from this shard to shard q, xsel_recv_box[xsel_recv_ptr[r] ..) come from shard r, ascending box ids
  // on both sides.  At N = 1M on 8 shards: 2-3 thousand boxes per shard instead of the 60 thousand of the all-gather.
  std::vector<int> xsel_send_ptr, xsel_send_box, xsel_recv_ptr, xsel_recv_box;
  int64_t near_nnz_owned = 0, m2l_pairs_owned = 0;

  PanelSoA panels;                        // tree order
  bool has_bc[2] = {false, false};

  // Builds everything above. Returns an empty string on success, else the error text.
  std::string build(const HostOptions& o, int64_t n_panels, const double* vertices, const uint8_t* bc);
  int nleaves() const { return (int)leaf_box.size(); }
};

// Triangulation::UnitSphere (examples/BEM/Triangulation.hpp:35-121)
int64_t unit_sphere(int recursions, double* vertices);

// Contiguous partition of leaves into `world` shards balanced by estimated work.
void partition_leaves(const HostPlan& hp, int world, std::vector<int>& cut);

}  // namespace fmmbem
