/* fmmbem.h -- C ABI of the MI355X-native FMM-BEM matvec (libfmmbem_hip.so).
 *
 * Drop-in boundary for the reference's hot path
 *     FMM_plan<LaplaceSphericalBEM>::FMM_plan / ::execute / ::kernel().set_p / ::options
 * (reference: include/FMM_plan.hpp:34-43, :75-90, :66-71, :94-96), i.e. everything that
 * include/executor/ExecutorSingleTree.hpp and EvalInteractionLazySparse.hpp do behind it.
 * Plain C types only: pointers, sizes, PODs.  No exit(), no stdout; every entry point returns
 * an fmmbem_status (0 = success) and fmmbem_last_error() gives the text of the last failure on
 * the calling thread.
 *
 * Threading: one in-flight execute per plan (as the reference: a plan is not re-entrant,
 * ExecutorSingleTree.hpp:120-129); distinct plans may run concurrently.
 */
#ifndef FMMBEM_H
#define FMMBEM_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define FMMBEM_VERSION 1
#define FMMBEM_PMAX 16          /* expansion orders 1..16 are pre-compiled */

typedef enum {
  FMMBEM_OK = 0,
  FMMBEM_ERR_INVALID = 1,       /* bad argument (null pointer, p out of range, bad quadrature key ...) */
  FMMBEM_ERR_NO_DEVICE = 2,     /* no usable HIP device, or the plan was built host-only            */
  FMMBEM_ERR_HIP = 3,           /* a HIP runtime call or kernel launch failed                         */
  FMMBEM_ERR_ALLOC = 4,         /* host or device allocation failed                                   */
  FMMBEM_ERR_TREE = 5,          /* octree deeper than 21 levels (64-bit Morton keys; the reference's 32-bit keys stop at 10:
                                 * trees that fit 10 levels are built by its rule, bit for bit, deeper ones by the wide coder) */
  FMMBEM_ERR_UNSUPPORTED = 6,   /* option combination not implemented                                 */
  FMMBEM_ERR_IO = 7             /* a mesh file could not be opened or parsed                          */
} fmmbem_status;

/* kernel ids: which reference Kernel class the plan stands in for */
typedef enum {
  FMMBEM_KERNEL_LAPLACE_BEM = 0, /* kernel/LaplaceSphericalBEM.hpp: 1 unknown per panel                  */
  FMMBEM_KERNEL_STOKES_BEM = 1   /* kernel/StokesSphericalBEM.hpp: 3 unknowns per panel, x/y hold Vec<3,double> per panel.
                                  * VELOCITY panels (stokeslet single layer, the operator the solve uses): everything.
                                  * TRACTION panels (bc flag 1, eval_traction_integral :160-258; the TARGET's flag picks the
                                  * operator, :377-389): near-matrix entries, fmmbem_kernel_entries, the near-field evaluators
                                  * (LOCAL, BLOCK_DIAGONAL), and through the FMM evaluator a double-layer far field checked
                                  * against the Direct sum (the reference's own disagrees with its Direct); p_max <= 12 */
} fmmbem_kernel;

/* boundary-condition flag per panel: LaplaceSphericalBEM::Panel::BoundaryType
 * (kernel/LaplaceSphericalBEM.hpp:40) */
enum { FMMBEM_BC_POTENTIAL = 0, FMMBEM_BC_NORMAL_DERIV = 1 };

/* Mirrors FMMOptions (include/FMMOptions.hpp:9-60) + the kernel constructor arguments
 * LaplaceSphericalBEM(p, k) (kernel/LaplaceSphericalBEM.hpp:131) + device placement.
 * Initialise with fmmbem_options_default(). */
/* executor/make_executor.hpp:24-60 (FMMOptions lazy_evaluation / local_evaluation / block_diagonal) */
typedef enum {
  FMMBEM_EVAL_FMM = 0,            /* EvalInteractionLazy(Sparse): near field + far field (the solver's operator)      */
  FMMBEM_EVAL_LOCAL = 1,          /* EvalLocal(Sparse): the near-field blocks only (Preconditioners::LocalInnerSolver,
                                   * examples/BEM/LocalPC.hpp:26-59)                                                  */
  FMMBEM_EVAL_BLOCK_DIAGONAL = 2  /* EvalDiagonalSparse: every leaf with itself (examples/BEM/BlockDiagonalPC.hpp:16-60)  */
} fmmbem_evaluator;

/* Downward pass.  The reference's lazy evaluator queues L2L parent->child only for children that were not an M2L
 * target EARLIER in the traversal than the parent (propagate_local / initialised_L,
 * executor/EvalInteractionLazySparse.hpp:199-237): on adaptive trees such a child never receives the far field its
 * ancestors collected and the result carries an error that does not decrease with p.  The reference's sphere meshes
 * produce no such edge, there the two rules are the same list.  fmmbem_stats.l2l_reference_omitted counts them. */
typedef enum {
  FMMBEM_L2L_COMPLETE = 0,        /* every child of a box that holds a local expansion (default)                      */
  FMMBEM_L2L_REFERENCE = 1        /* exactly the reference's list, omissions included (bit parity on adaptive trees)  */
} fmmbem_l2l_rule;

typedef struct {
  int32_t  kernel;            /* fmmbem_kernel                                                     */
  int32_t  p_max;             /* largest expansion order any execute() will ask for (1..16)        */
  int32_t  quad_k;            /* Gauss rule key K: 1,3,4,7,13,17,19,25 (GaussQuadrature.hpp)       */
  double   theta;             /* FMMOptions::set_mac_theta, default 0.5 (FMMOptions.hpp:45)        */
  uint32_t ncrit;             /* FMMOptions::set_max_per_box, default 64 (FMMOptions.hpp:46-47)    */
  int32_t  sparse_local;      /* 1: assembled near matrix (EvalInteractionLazySparse, what the BEM
                               *    drivers select); 0: matrix-free near field recomputed every
                               *    matvec (EvalInteractionLazy, `-disable_sparse`)                  */
  int32_t  host_only;         /* 1: build tree + lists on the host, touch no device (CPU tests)    */
  int32_t  device;            /* HIP device ordinal                                                */
  int32_t  shard_rank;        /* target-leaf shard owned by this plan: rank of world               */
  int32_t  shard_world;       /* 1 = whole operator                                                */
  int32_t  quad_k_fine;       /* Stokes: near-regime Gauss rule K_fine (StokesSphericalBEM::set_Kfine,
                               * kernel/StokesSphericalBEM.hpp:139-141; ctor default 25, driver 19)  */
  int32_t  evaluator;         /* fmmbem_evaluator: which branch of make_evaluators the plan takes   */
  double   mu;                /* Stokes: viscosity (StokesSphericalBEM(p,k,mu), :131)               */
  int32_t  shard_upward;      /* shard_world > 1: also shard P2M/M2M by owner; the multipoles are exchanged by ONE
                               * collective the caller performs between fmmbem_plan_upward_device and
                               * fmmbem_plan_downward_device: 1 = an all-gather of every shard's multipoles, 2 = an
                               * all-to-all of the ones each receiver reads (fmmbem_plan_exchange_counts);
                               * 0: every shard repeats the whole upward pass                                  */
  int32_t  l2l_rule;          /* fmmbem_l2l_rule: which parent->child L2L edges the downward pass applies              */
  double   near_stream_fraction; /* sparse_local = 1: share of the near-field pairs whose entries are STORED and streamed every
                               * matvec (P2P_Lazy::to_matrix + Matvec<>, EvalP2P.hpp:47-98, Matvec.hpp:14-33).  1.0 (default): all
                               * of them, the reference's assembled matrix.  0 <= f < 1, HYBRID: the target leaves with the most
                               * rows keep no matrix until they hold 1 - f of the pairs; their far-regime entries are recomputed
                               * every matvec from the quadrature points (what EvalInteractionLazy::eval_P2P_lists does,
                               * EvalInteractionLazy.hpp:239-252) by the SAME kernel between its streamed items -- arithmetic in
                               * the issue slots the HBM-bound stream leaves empty -- and their near-regime entries are evaluated
                               * once.  Footprint and HBM bytes fall by 1 - f; a row's sum is formed in a fixed order, results
                               * differ from f = 1 in the last bits.  Taken as 1 where the hybrid kernel does not apply (rules of
                               * more than 3 / 4 points, the LOCAL / BLOCK_DIAGONAL evaluators).  FMMBEM_NEAR_STREAM_FRACTION
                               * overrides it at creation (sweeps).                                                      */
  int32_t  n_devices;         /* > 1: ONE plan over the devices listed in `devices` (at most 8; SURVEY 8b "device list"): the target leaves
                               * are sharded over them inside the handle, which copies x to every device, exchanges the multipoles
                               * the shards' lists read and brings the result slices home by peer copies over xGMI; x and y of
                               * fmmbem_plan_execute_device / fmmbem_gmres_device live on devices[0].  The same bits as one device.
                               * shard_upward = 0 then makes every device repeat the upward pass instead of exchanging multipoles.
                               * 0 or 1: the single `device`.  With n_devices == 0 the environment variable FMMBEM_DEVICES=0,1,...
                               * supplies a list (for callers that cannot: the reference's unmodified drivers)                  */
  int32_t  devices[8];
} fmmbem_options;

/* Statistics of a plan and of its last execute (times in milliseconds, device-side HIP events). */
typedef struct {
  int64_t n_panels, n_boxes, n_leaves, n_levels;
  int64_t near_nnz;             /* entries of the assembled near matrix owned by this shard          */
  int64_t near_nnz_total;       /* ... of the whole operator                                         */
  int64_t p2p_pairs, m2l_pairs, m2l_pairs_owned, m2m_ops, l2l_ops, p2m_leaves, l2p_leaves;
  int64_t m2l_classes;          /* distinct M2L translation vectors                                  */
  int64_t owned_leaf_begin, owned_leaf_end, owned_row_begin, owned_row_end;
  int64_t near_bytes;           /* HBM bytes of the near-matrix values                               */
  int32_t expansions_active;    /* bit 0: G expansion, bit 1: dG/dn expansion                        */
  int32_t last_p;
  double  build_host_ms, build_assemble_ms;
  /* per-stage device times: MEAN over the executes recorded since timing was (re)enabled */
  double  ms_total, ms_gather, ms_near, ms_scatter, ms_p2m, ms_m2m, ms_mh, ms_m2l, ms_l2l, ms_l2p;
  int64_t timed_executes;       /* how many executes the means cover                                 */
  int64_t l2l_reference_omitted;/* L2L edges FMMBEM_L2L_REFERENCE leaves out of this tree (0: the rules coincide) */
  int64_t m2l_items, m2l_passes;/* rotation M2L: work items (one wavefront each) and 64-pair passes over them; pairs / (64 passes)
                                 * is the lane fill                                                   */
  int64_t near_side_entries;    /* matrix-free plans (sparse_local = 0): near-regime pairs evaluated once at creation and kept
                                 * (12 bytes each: column + value); 0 for assembled plans                */
  int32_t m2l_kernel;           /* the M2L an execute at last_p took: 1 rotation kernel, 2 double sum, 3 double sum (lanes = sources) */
  int32_t expansion_slots;             /* slots per box in M and L (fmmbem_plan_get_expansions): Laplace 2, Stokes 8, 11 with TRACTION targets */
  int64_t rot_nop_orders;       /* build record: orders p of the rotation kernels that were compiled with a wait state in front of
                                 * every DPP FMA because the build's ISA check found the DPP hazard in their code (csrc/Makefile,
                                 * tools/check_rot_isa.py): bit p-1 M2L, bit 16+p-1 M2M, bit 32+p-1 L2L.
                                 * 0 on the toolchain this was developed with; such an order runs ~30 % slower, not wrong */
  int32_t tree_coder_levels;    /* 10: the reference's 32-bit Morton coder built the tree; 21: the 64-bit coder had to (deeper tree) */
  int32_t n_devices;            /* devices this plan runs on (fmmbem_options.n_devices / FMMBEM_DEVICES); the figures above are then the whole
                                 * operator's: owned counts summed over the shards, stage times the longest shard's                    */
  int32_t geometry_shared;      /* plans alive that share this plan's tree, lists and tables (fmmbem_plan_create_like), itself included */
  int64_t near_recomputed_pairs;/* hybrid plans (near_stream_fraction < 1): panel pairs of this shard that are recomputed every matvec
                                 * instead of stored (near_nnz counts all of the shard's entries, near_bytes what is stored)         */
} fmmbem_stats;

typedef struct fmmbem_plan fmmbem_plan;

/* ---- lifecycle -------------------------------------------------------------------------- */
void fmmbem_options_default(fmmbem_options *opts);

/* FMM_plan(const Kernel&, const std::vector<source_type>&, FMMOptions&)  (FMM_plan.hpp:34-43).
 * vertices: n_panels x 3 vertices x 3 coordinates (the arguments of Panel(p0,p1,p2),
 * LaplaceSphericalBEM.hpp:64); bc: n_panels flags or NULL (all POTENTIAL).
 * Builds octree + interaction lists on the host, uploads them once, assembles the near matrix
 * on the device. */
int fmmbem_plan_create(const fmmbem_options *opts, size_t n_panels, const double *vertices,
                       const uint8_t *bc, fmmbem_plan **out);
/* A plan of the SAME panels, options and order as `base` with other boundary-condition flags -- the drivers' right-hand-side plan
 * (examples/LaplaceBEM.cpp:218-232: the panels with their flags switched; StokesBEM.cpp:266-270) and the preconditioners' plans.
 * Tree, permutation, pair lists, work items and operator tables are SHARED with base through a reference count (either plan may
 * be destroyed first); only what the flags decide is built: the near-matrix values, the P2M moments, the expansions.  0.05 s
 * instead of 0.28 s at N = 1M.  fmmbem_plan_create does the same on its own when it is handed the vertices and options of a live
 * plan (recognised by two 64-bit hashes of the vertex bytes; FMMBEM_PLAN_SHARE=0 disables that). */
int fmmbem_plan_create_like(const fmmbem_plan *base, const uint8_t *bc, fmmbem_plan **out);
void fmmbem_plan_destroy(fmmbem_plan *plan);

/* ---- the hot path ----------------------------------------------------------------------- */
/* results = plan.execute(charges) with kernel().set_p(p) applied first
 * (FMM_plan.hpp:75-90; GMRES.hpp:194-201).  x, y: n_panels x dof doubles (dof = 1 Laplace, 3 Stokes,
 * interleaved per panel as std::vector<Vec<3,double>>) in ORIGINAL panel order, HOST pointers; y is overwritten.  With shard_world > 1, y holds this shard's rows and zeros
 * elsewhere (sum over shards = full result). */
int fmmbem_plan_execute(fmmbem_plan *plan, int p, const double *x, double *y);

/* Optional, for callers that keep their vectors: page-lock a host buffer (hipHostRegister) so that the two copies of a
 * host-pointer execute run at the PCIe rate instead of through the runtime's staging of pageable memory (INTEGRATION.md gives
 * the measured difference).  The caller owns the memory: unregister BEFORE freeing or reallocating it -- which is why the
 * library never registers a caller's pointer on its own (a std::vector that the reference's solver reallocates would leave a
 * stale registration behind).  Not needed for fmmbem_gmres, whose vectors cross PCIe once per solve. */
int fmmbem_host_register(void *ptr, size_t bytes);
int fmmbem_host_unregister(void *ptr);

/* Same, with DEVICE pointers and an explicit hipStream_t (NULL = default stream); asynchronous. */
int fmmbem_plan_execute_device(fmmbem_plan *plan, int p, const double *d_x, double *d_y, void *stream);

/* Near-field part only: y = A_near x (what Matvec<> does in EvalInteractionLazySparse.hpp:136-148).
 * DEVICE pointers, asynchronous. */
int fmmbem_plan_near_device(fmmbem_plan *plan, const double *d_x, double *d_y, void *stream);

/* Toggle per-stage HIP-event timing.  While enabled every execute records HIP events around each
 * kernel on the stream it runs on (no synchronisation is added); fmmbem_plan_stats() waits for the
 * recorded events and reports the mean stage times of up to the last 64 executes.  Enabling resets
 * the record.  Default off. */
int fmmbem_plan_set_timing(fmmbem_plan *plan, int enabled);      /* 0 off; 1 every stage; 2 the near-field kernel only */

/* Replay the launch chain of an execute (everything between the gather of x and the delivery of y) as a hipGraph per order p,
 * captured the second time the plan runs at that order and launched on the caller's stream from then on: one host call
 * instead of ~16.  Same kernels, same order, same bits.  Off by default (also FMMBEM_GRAPH=1 at creation); executes with
 * stage timing on are not graphed.  Environment switches that the launchers read per launch are frozen into a captured
 * graph. */
int fmmbem_plan_set_graphs(fmmbem_plan *plan, int enabled);

int fmmbem_plan_stats(const fmmbem_plan *plan, fmmbem_stats *out);

/* ---- introspection (tests, partition checks) --------------------------------------------- */
/* Body::number() permutation, tree index -> original index (Octree.hpp:298-300). out: n_panels. */
int fmmbem_plan_get_perm(const fmmbem_plan *plan, uint32_t *out);
/* Boxes in BFS order. Any pointer may be NULL. center: n_boxes x 3. */
int fmmbem_plan_get_boxes(const fmmbem_plan *plan, double *center, double *side, int32_t *level,
                          int32_t *is_leaf, int32_t *parent, int32_t *body_begin, int32_t *body_end);
/* Pair lists; which: 0 = P2P (source leaf, target leaf), 1 = M2L (source box, target box),
 * 2 = M2M (child, parent), 3 = L2L (parent, child); the work list of the rotation M2L kernel: 4 = its OWNED pairs (source
 * box, target box) in the order the kernel takes them, 5 = its items as (first, one past last) positions in list 4 (the
 * short cut, orders with several wavefronts per SIMD), 6 = the long cut of the same list (orders p >= 9).
 * out may be NULL; returns the count via *n. */
int fmmbem_plan_get_pairs(const fmmbem_plan *plan, int which, int32_t *out, int64_t *n);
/* One assembled near-matrix row (tree-order index of the unknown, i.e. dof*panel + component; must be
 * owned): column indices (same numbering) and values, ascending columns.  cols/vals may be NULL; *n
 * receives the row length. */
int fmmbem_plan_get_near_row(const fmmbem_plan *plan, int64_t row, uint32_t *cols, double *vals,
                             int64_t *n);
/* Diagonal of the assembled near matrix = the self-interactions K(s,s) that Preconditioners::Diagonal inverts
 * (examples/BEM/Preconditioner.hpp:19-42; examples/LaplaceBEM.cpp:241-244): out[n_panels * dof], original order;
 * rows of other shards are returned as zero. */
int fmmbem_plan_get_diagonal(const fmmbem_plan *plan, double *out);

/* Kernel::operator()(target, source) for n independent panel pairs (kernel/LaplaceSphericalBEM.hpp:273-297,
 * kernel/StokesSphericalBEM.hpp:377-389): the panel integral of source i seen from the centroid of target i, evaluated on
 * the device by the code that assembles the near matrix.  Reads opts->kernel, quad_k, quad_k_fine, mu, device.
 * target_vertices / source_vertices: n x 9; target_bc: n flags or NULL (the TARGET's flag selects G vs dG/dn, :282-291);
 * out: n doubles (Laplace) or n row-major 3x3 blocks (Stokes).  No plan is needed: this is what Preconditioners::Diagonal
 * (examples/BEM/Preconditioner.hpp:24-33) calls as K(*it, *it) on the panels of FMM_plan::source_begin(). */
int fmmbem_kernel_entries(const fmmbem_options *opts, size_t n, const double *target_vertices, const uint8_t *target_bc,
                          const double *source_vertices, double *out);

/* ---- the translation operators one at a time: the KernelSkeleton contract (kernel/KernelSkeleton.hpp:62-212) --------------
 * What the reference's kernel classes offer beside operator(): P2M, M2M, M2L, L2L, L2P as kernel/LaplaceSphericalBEM.hpp:307-476
 * and kernel/StokesSphericalBEM.hpp:391-530 define them (each "+=" into its last argument), for code that drives single operators
 * (tests/single_level.cpp, ExpansionTraits<K>::is_valid_fmm of include/KernelTraits.hpp:188-194).  Every call runs the SAME device
 * kernels a plan's matvec runs, on a two-box plan that holds just the call's operands (csrc/ops.hip); operands and results are
 * host buffers.  M2P (the treecode's operator) is not offered.
 *   fmmbem_ops_create reads opts->kernel, p_max (1..16), quad_k, mu, device.  A handle is not thread-safe.
 *   An expansion: [slot][p (p + 1) / 2] complex as (re, im) pairs, coefficient (n, m >= 0) at n (n + 1) / 2 + m -- the reference's
 *   stored half (kernel/LaplaceSpherical.hpp:187-206).  fmmbem_ops_slots: the slots P2M writes and L2P reads -- Laplace 2
 *   (multipole_type[0] = G moments of POTENTIAL sources, [1] = dG/dn moments of NORMAL_DERIV sources, :323-344; L2P adds r0 at a
 *   POTENTIAL target and subtracts r1 at a NORMAL_DERIV target, :448-476), Stokes 4 (the stokeslet group M[0][0..3] of VELOCITY
 *   sources, StokesSphericalBEM.hpp:414-430; L2P at VELOCITY targets with the 1 / (2 mu), :512-522).  Stokes TRACTION sources
 *   and targets are FMMBEM_ERR_UNSUPPORTED here: the reference's stresslet moments (:432-466) are not the far field this library
 *   computes for those rows (see FMMBEM_KERNEL_STOKES_BEM above).
 *   M2M / M2L / L2L act on n_slots (1..12) expansions alike -- pass both groups of a Stokes multipole_type as 8.
 *   translation = centre of the target expansion - centre of the source expansion (executor/M2M.hpp, M2L.hpp:40, L2L.hpp).
 *   vertices: n x 9; bc: n flags or NULL (all 0); charges: n x dof; result: n x dof, added to. */
typedef struct fmmbem_ops fmmbem_ops;
int fmmbem_ops_create(const fmmbem_options *opts, fmmbem_ops **out);
void fmmbem_ops_destroy(fmmbem_ops *ops);
int fmmbem_ops_slots(const fmmbem_ops *ops);
int fmmbem_ops_p2m(fmmbem_ops *ops, int p, size_t n, const double *vertices, const uint8_t *bc, const double *charges,
                   const double center[3], double *M);
int fmmbem_ops_m2m(fmmbem_ops *ops, int p, int n_slots, const double *M_source, double *M_target, const double translation[3]);
int fmmbem_ops_m2l(fmmbem_ops *ops, int p, int n_slots, const double *M_source, double *L_target, const double translation[3]);
int fmmbem_ops_l2l(fmmbem_ops *ops, int p, int n_slots, const double *L_source, double *L_target, const double translation[3]);
int fmmbem_ops_l2p(fmmbem_ops *ops, int p, const double *L, const double center[3], size_t n, const double *vertices,
                   const uint8_t *bc, double *result);

/* ---- the orthogonalisation step of the callers above the matvec (examples/BEM/GMRES.hpp:203-212: modified Gram-Schmidt of
 * w against V_0 .. V_{ncols-1}, then the normalised next basis vector), device vectors, ONE call per Arnoldi column:
 *   for k < ncols:  h[k] = <w, V_k>;  w -= h[k] V_k;      h[ncols] = |w|;   vnext = w / h[ncols]
 * d_V: ncols vectors of n doubles, ldv apart; d_h: ncols + 1 doubles on the device; d_scratch:
 * fmmbem_mgs_scratch_doubles(largest ncols) doubles of work space (no initial contents assumed; reusable across calls and
 * sizes); asynchronous on `stream`.  The sums are formed in a fixed order.  Vectors that start on 16-byte boundaries with an
 * even ldv take the vector path; anything else is correct but scalar. */
int fmmbem_mgs_column_device(int64_t n, double *d_w, const double *d_V, int64_t ldv, int ncols, double *d_h, double *d_vnext,
                             double *d_scratch, void *stream);
int fmmbem_mgs_scratch_doubles(int max_cols);

/* ---- the caller of the hot path, resident on the device: restarted GMRES / flexible GMRES with the per-iteration relaxation
 * of the expansion order p (examples/BEM/GMRES.hpp:143-252 GMRES, :276-380 FGMRES; examples/BEM/GMRES_Stokes.hpp:173-320 the
 * same loops on Vec<3,double> values with the Stokes order rule; examples/BEM/SolverOptions.hpp:11-39).  The Krylov basis V
 * (and Z of FGMRES) lives in HBM, every matvec is fmmbem_plan_execute_device, every Arnoldi column one
 * fmmbem_mgs_column_device; the Hessenberg column comes to the host once per iteration (its only synchronisation) for the
 * Givens rotations, the residual estimate and predict_p; restart, back substitution and the solution update as the reference.
 * The same steps in the same order as the reference's loop -- order schedule, iteration counts and printed residuals are
 * the reference's (tests/golden/gmres_ref_r*.json) -- only the association inside a dot product differs. */
typedef enum { FMMBEM_RELAX_SIMONCINI = 0, FMMBEM_RELAX_BOURAS = 1 } fmmbem_relaxation;   /* SolverOptions::relaxation_type */
typedef enum {
  FMMBEM_ORDER_GMRES = 0,          /* GMRES.hpp:195         p = max(1, predict_p(|r|))                                    */
  FMMBEM_ORDER_GMRES_STOKES = 1,   /* GMRES_Stokes.hpp:229  p = max(p_min, predict_p(|r|) - 1)                            */
  FMMBEM_ORDER_FGMRES = 2,         /* GMRES.hpp:324         p = predict_p(|r|)  (raised to 1: the library has no order 0)   */
  FMMBEM_ORDER_FGMRES_STOKES = 3   /* GMRES_Stokes.hpp:373  p = max(5, predict_p(|r|))                                     */
} fmmbem_order_rule;

typedef struct {
  double  residual;        /* SolverOptions::residual: stop when |r| / |b| falls below it                               */
  int32_t max_iters;       /* SolverOptions::max_iters                                                                 */
  int32_t restart;         /* SolverOptions::restart (the drivers set it to max_iters, LaplaceBEM.cpp:162-163)          */
  int32_t max_p, p_min;    /* SolverOptions::max_p, p_min                                                              */
  int32_t variable_p;      /* SolverOptions::variable_p: 0 = every matvec at max_p                                     */
  int32_t relax_type;      /* fmmbem_relaxation                                                                        */
  int32_t order_rule;      /* fmmbem_order_rule: which call site's floor is applied to predict_p                       */
  int32_t flexible;        /* 0 GMRES (x += y_j M(V_j), the preconditioner applied again in the update, GMRES.hpp:237-241);
                            * 1 FGMRES (Z_j = M(V_j) kept, x += y_j Z_j, :318-320, :368-371)                           */
  int32_t initial_p;       /* the order of the FIRST matvec r0 = A x0 - b, run before any predict_p: the reference uses whatever
                            * the kernel object holds (its construction order); 0 = max_p                               */
} fmmbem_solver_options;
void fmmbem_solver_options_default(fmmbem_solver_options *opts);   /* SolverOptions(): 1e-5, 500, 500, 16, 5, true, BOURAS */

typedef enum {
  FMMBEM_PC_IDENTITY = 0,    /* Preconditioners::Identity (Preconditioner.hpp:8-17)                                       */
  FMMBEM_PC_DIAGONAL = 1,    /* Preconditioners::Diagonal (:19-42): z = reciprocals .* v, in whatever order the caller built them */
  FMMBEM_PC_INNER_PLAN = 2   /* Preconditioners::LocalInnerSolver / BlockDiagonal (LocalPC.hpp:26-59, BlockDiagonalPC.hpp:16-60):
                              * z = GMRES(inner_plan, 0, v, inner) on a plan created with evaluator LOCAL or BLOCK_DIAGONAL */
} fmmbem_preconditioner_kind;
typedef struct {
  int32_t kind;                      /* fmmbem_preconditioner_kind                                                     */
  const double *reciprocals;         /* DIAGONAL: n_panels * dof values; a DEVICE pointer for fmmbem_gmres_device, a host
                                      * pointer for fmmbem_gmres                                                       */
  fmmbem_plan *inner_plan;           /* INNER_PLAN: same panels, same device                                           */
  fmmbem_solver_options inner;       /* INNER_PLAN: LocalPC.hpp:52-54 uses residual 1e-1, variable_p 0, max_iters 1     */
} fmmbem_preconditioner;

typedef struct {
  int32_t iterations;                /* out: matvecs of the inner loops (the reference's `iter`)                        */
  double  residual;                  /* out: last |r| / |b| estimate                                                  */
  double  seconds;                   /* out: wall time of the call, the final synchronisation included                 */
  int32_t capacity;                  /* in: entries p[] / resid[] can take (0 or NULL arrays: no history)              */
  int32_t *p;                        /* out: order of iteration k (what the reference prints as fmm_req_p)             */
  double  *resid;                    /* out: |r| / |b| after iteration k                                               */
} fmmbem_solver_log;

/* x: initial guess in, solution out; b: right-hand side; n_panels * dof doubles each, ORIGINAL panel order, DEVICE pointers.
 * M == NULL: identity.  log may be NULL.  Synchronises `stream` once per iteration and before returning. */
int fmmbem_gmres_device(fmmbem_plan *plan, const fmmbem_solver_options *opts, double *d_x, const double *d_b,
                        const fmmbem_preconditioner *M, fmmbem_solver_log *log, void *stream);
/* The same with HOST pointers (x, b, M->reciprocals): the vectors cross PCIe once each way per solve, not per matvec. */
int fmmbem_gmres(fmmbem_plan *plan, const fmmbem_solver_options *opts, double *x, const double *b,
                 const fmmbem_preconditioner *M, fmmbem_solver_log *log);

/* ---- split execute of a plan created with shard_upward = 1 and shard_world > 1 (no reference counterpart:
 * the reference is single-node) ----------------------------------------------------------
 *   upward:   x -> P2M and M2M of the boxes this shard owns -> d_send (exchange_doubles(p) doubles)
 *   caller:   all-gather of the shards' d_send into d_recv (shard_world x exchange_doubles(p), rank order)
 *             [meanwhile, optionally: near_split -> y]
 *   downward: d_recv -> the remaining M2M, M2L, L2L, L2P and the near field -> y (zero outside the owned rows) */
int fmmbem_plan_exchange_doubles(const fmmbem_plan *plan, int p, size_t *per_shard);
/* shard_upward = 2: the same split execute, but the caller's collective is ONE all-to-all with uneven counts that carries,
 * for every pair of shards, only the multipoles the receiver's lists read (the M2L sources of its targets and the private
 * children of the few parents every shard translates): 2-3 thousand boxes per shard instead of 60 thousand at N = 1M on 8
 * shards.  send_doubles[q] / recv_doubles[q], q < shard_world: doubles this shard sends to / receives from shard q at order p
 * (0 for itself); d_send / d_recv of upward / downward hold the peers' segments one after the other, rank order -- the
 * layout of MPI_Alltoallv / torch.distributed.all_to_all_single with split sizes. */
int fmmbem_plan_exchange_counts(const fmmbem_plan *plan, int p, int64_t *send_doubles, int64_t *recv_doubles);
int fmmbem_plan_upward_device(fmmbem_plan *plan, int p, const double *d_x, double *d_send, void *stream);
int fmmbem_plan_downward_device(fmmbem_plan *plan, int p, const double *d_recv, double *d_y, void *stream);
/* Optional, between the two: the near field of this shard (y = A_near x of the x given to upward; zero outside the owned
 * rows), so that it runs while the caller's all-gather is in flight.  downward then skips it and adds the far field. */
int fmmbem_plan_near_split_device(fmmbem_plan *plan, double *d_y, void *stream);

/* The result as a slice instead of a zero-padded vector (the shards' rows are disjoint and, in tree
 * order, contiguous -- an all-gather moves half the bytes of the all-reduce).  With result slices enabled, every execute /
 * downward call writes only the rows this shard owns, in TREE order, to d_y[0 .. rows * dof); the caller all-gathers the
 * shards' slices into chunks of `chunk_doubles` (rank order; chunk >= the largest slice) and
 * fmmbem_plan_assemble_slices_device puts them into ORIGINAL panel order: y[n_panels * dof].  The values are the ones the
 * zero-padded form holds, bit for bit.  fmmbem_plan_shard_rows: cut[shard_world + 1], shard r owns tree rows
 * [cut[r], cut[r+1]) (every rank builds the same tree, so every rank knows all cuts). */
int fmmbem_plan_set_result_slices(fmmbem_plan *plan, int enabled);
int fmmbem_plan_shard_rows(const fmmbem_plan *plan, int64_t *cut);
int fmmbem_plan_assemble_slices_device(fmmbem_plan *plan, const double *d_slices, size_t chunk_doubles, double *d_y, void *stream);

/* Multipole (which=0) or local (which=1) coefficients of the last execute for every box:
 * out[box][slot][p(p+1)/2][re,im]; slots = fmmbem_stats.expansion_slots: Laplace 2 (G, dG/dn); Stokes 8 (M[2][4]), or 11 when
 * the plan has TRACTION targets (slots 4..10: the seven dipole potentials of the double layer). */
int fmmbem_plan_get_expansions(const fmmbem_plan *plan, int which, int p, double *out);

/* ---- mesh generator of the reference's drivers ------------------------------------------- */
/* Triangulation::UnitSphere(panels, recursions) (examples/BEM/Triangulation.hpp:105-121):
 * N = 2*4^recursions panels.  vertices == NULL: only returns N through *n_panels. */
int fmmbem_mesh_unit_sphere(int recursions, double *vertices, size_t *n_panels);
/* Triangulation::RedBloodCell(panels, recursions) with identity rotation and zero shift
 * (examples/BEM/Triangulation.hpp:184-255; examples/StokesBEM.cpp:111-113): same N, same calling convention. */
int fmmbem_mesh_red_blood_cell(int recursions, double *vertices, size_t *n_panels);
/* Triangulation::MultipleRedBloodCell(panels, recursions, cells) (examples/BEM/Triangulation.hpp:260-321): N =
 * cells * 2*4^recursions.  placement: cells x {alpha, beta, gamma, shift x, y, z} (RotationMatrix, :142-163), or NULL for
 * the reference's own drand48-driven orientations and offsets (see csrc/mesh_io.cpp for what that sequence assumes). */
int fmmbem_mesh_red_blood_cells(int recursions, int cells, const double *placement, double *vertices, size_t *n_panels);

/* Triangle Gauss rule `key` of examples/BEM/GaussQuadrature.hpp:15-274 (what BEMConfig hands the kernels): barycentric
 * points[n][3] and weights[n], n <= FMMBEM_MAX_QUAD (size the buffers for that); either array may be NULL.
 * Keys: 1 3 4 7 (alias of 4, :58-59) 13 17 (16 points) 19 25 79. */
#define FMMBEM_MAX_QUAD 79
int fmmbem_quadrature(int key, double *points, double *weights, int *n);

/* ---- mesh files of the reference's drivers ------------------------------------------------ */
/* All readers: vertices == NULL only counts (*n_panels out); otherwise *n_panels holds the capacity of
 * vertices (panels) on entry and the number read on return; vertices[panel][vertex][xyz].
 * MeshIO::readMsh (examples/BEM/MshReader.hpp:18-94): gmsh format 2 ASCII; elements of type 2 (triangles) only,
 * stored in element-number order with the winding swapped (v1, v3, v2) as the reference does. */
int fmmbem_mesh_read_msh(const char *path, double *vertices, size_t *n_panels);
/* MeshIO::ReadVertFace (examples/BEM/VertFaceReader.hpp:17-76): count line then "x y z" lines; count line then
 * 1-indexed "v1 v2 v3" lines; winding kept. */
int fmmbem_mesh_read_vert_face(const char *vert_path, const char *face_path, double *vertices, size_t *n_panels);
/* The .vert/.face pair the generators dump (examples/BEM/Triangulation.hpp:124-134), three vertices per
 * triangle, written WITH the count lines ReadVertFace expects and 17 significant digits. */
int fmmbem_mesh_write_vert_face(const char *vert_path, const char *face_path, const double *vertices, size_t n_panels);

/* ---- errors ------------------------------------------------------------------------------ */
const char *fmmbem_status_string(int status);
const char *fmmbem_last_error(void);
int fmmbem_version(void);

#ifdef __cplusplus
}
#endif
#endif /* FMMBEM_H */
