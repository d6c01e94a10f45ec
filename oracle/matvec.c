/* matvec.c -- CPU ORACLE (test infrastructure): near-field CSR assembly, the FMM matvec stage
 * driver of EvalInteractionLazySparse::execute, and the O(N^2) Direct sum.
 * See fmm_oracle.h for the rules. */
#include "fmm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#else
static double omp_get_wtime(void) { return 0; }
#endif

/* executor/EvalP2P.hpp:47-98 (P2P_Lazy::to_matrix): rows = target body, cols = source body, both in
 * tree order; columns sorted ascending per row; entry = K(target_i, source_j). */
int orc_build_near_pattern(orc_ctx *c) {
  if (c->row_ptr) return 0;
  const int n = c->n;
  int64_t *cnt = calloc((size_t)n + 1, sizeof(int64_t));
  /* per target leaf: list of source leaves */
  int *tptr = calloc((size_t)c->nboxes + 1, sizeof(int));
  for (int i = 0; i < c->n_p2p; ++i) tptr[c->p2p[i].second + 1]++;
  for (int b = 0; b < c->nboxes; ++b) tptr[b+1] += tptr[b];
  int *tsrc = malloc(sizeof(int)*(size_t)(c->n_p2p ? c->n_p2p : 1));
  int *fill = malloc(sizeof(int)*(size_t)c->nboxes);
  memcpy(fill, tptr, sizeof(int)*(size_t)c->nboxes);
  for (int i = 0; i < c->n_p2p; ++i) tsrc[fill[c->p2p[i].second]++] = c->p2p[i].first;
  free(fill);
  /* sort each target's source leaves by body_begin: equivalent to the per-row std::sort of columns
     (EvalP2P.hpp:87) because leaves own disjoint contiguous body ranges */
  for (int b = 0; b < c->nboxes; ++b) {
    int *s = tsrc + tptr[b]; int m = tptr[b+1] - tptr[b];
    for (int i = 1; i < m; ++i) {
      int v = s[i], j = i - 1;
      while (j >= 0 && c->boxes[s[j]].bb > c->boxes[v].bb) { s[j+1] = s[j]; --j; }
      s[j+1] = v;
    }
  }
  for (int b = 0; b < c->nboxes; ++b) {
    if (!c->boxes[b].leaf) continue;
    int64_t cols = 0;
    for (int i = tptr[b]; i < tptr[b+1]; ++i) cols += c->boxes[tsrc[i]].be - c->boxes[tsrc[i]].bb;
    for (uint32_t r = c->boxes[b].bb; r < c->boxes[b].be; ++r) cnt[r+1] = cols;
  }
  for (int i = 0; i < n; ++i) cnt[i+1] += cnt[i];
  c->row_ptr = cnt; c->nnz = cnt[n];
  c->col = malloc(sizeof(uint32_t)*(size_t)(c->nnz ? c->nnz : 1));
  if (!c->col) return -1;
  #pragma omp parallel for schedule(dynamic, 1)
  for (int b = 0; b < c->nboxes; ++b) {
    if (!c->boxes[b].leaf) continue;
    for (uint32_t r = c->boxes[b].bb; r < c->boxes[b].be; ++r) {
      int64_t at = c->row_ptr[r];
      for (int i = tptr[b]; i < tptr[b+1]; ++i) {
        const orc_box *sb = &c->boxes[tsrc[i]];
        for (uint32_t j = sb->bb; j < sb->be; ++j, ++at) c->col[at] = j;
      }
    }
  }
  free(tptr); free(tsrc);
  return 0;
}

int orc_build_near(orc_ctx *c) {
  if (c->val) return 0;
  if (orc_build_near_pattern(c)) return -1;
  c->val = malloc(sizeof(double)*(size_t)(c->nnz ? c->nnz : 1));
  if (!c->val) return -1;
  #pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < c->n; ++i) {
    const orc_panel *t = &c->panels[c->perm[i]];
    for (int64_t k = c->row_ptr[i]; k < c->row_ptr[i+1]; ++k)
      c->val[k] = orc_kernel(c, t, &c->panels[c->perm[c->col[k]]]);
  }
  return 0;
}

static void ensure_expansions(orc_ctx *c, int P) {
  if (c->pcap >= P) return;
  free(c->M); free(c->L);
  size_t sz = (size_t)c->nboxes * 2 * (size_t)(P*(P+1)/2);
  c->M = malloc(sizeof(cplx)*sz); c->L = malloc(sizeof(cplx)*sz);
  c->pcap = P;
}

/* y = A x, following executor/EvalInteractionLazySparse.hpp:120-168 stage by stage.
 * flags & ORC_FLAG_FAITHFUL: parallel structure and work of the reference: serial CSR SpMV
 *   (include/Matvec.hpp:25-32), OpenMP P2M, serial M2M, OpenMP M2L, serial L2L, OpenMP L2P, both
 *   expansions translated (kernel/LaplaceSphericalBEM.hpp:362-383,432-437).  The reference's M2L
 *   loop runs over PAIRS with a racy accumulation (:269-283); here the same work is grouped by
 *   target box so the sum is race-free.
 * otherwise ("tuned"): every stage threaded, and an expansion that no panel feeds is skipped.
 * stage_s: seconds for {init, near, P2M, M2M, M2L, L2L, L2P, total}. y is OVERWRITTEN. */
int orc_matvec(orc_ctx *c, int P, const double *x, double *y, int flags, double stage_s[8]) {
  if (P < 1 || P > ORC_PMAX) return -1;
  if (orc_build_near(c)) return -2;
  const int faithful = flags & ORC_FLAG_FAITHFUL;
  const int n = c->n, nb = c->nboxes, S = P*(P+1)/2;
  orc_tables *t = orc_tables_create(P);            /* K.set_p(p): LaplaceSpherical.hpp:119-128 */
  ensure_expansions(c, P);
  double T[9]; T[0] = omp_get_wtime();
  int use[2] = {1, 1};
  if (!faithful) {
    use[0] = use[1] = 0;
    for (int i = 0; i < n; ++i) use[c->panels[i].bc == ORC_POTENTIAL ? 0 : 1] = 1;
  }
  /* INITM / INITL (:124-131) */
  memset(c->M, 0, sizeof(cplx)*(size_t)nb*2*S);
  memset(c->L, 0, sizeof(cplx)*(size_t)nb*2*S);
  memset(y, 0, sizeof(double)*(size_t)n);
  T[1] = omp_get_wtime();
  /* near field (:136-148): gather to tree order, CSR SpMV, scatter-add to original order */
  double *xt = malloc(sizeof(double)*(size_t)n), *yt = malloc(sizeof(double)*(size_t)n);
  for (int i = 0; i < n; ++i) xt[i] = x[c->perm[i]];
  if (faithful) {
    for (int i = 0; i < n; ++i) {
      double r = 0;
      for (int64_t k = c->row_ptr[i]; k < c->row_ptr[i+1]; ++k) r += c->val[k] * xt[c->col[k]];
      yt[i] = r;
    }
  } else {
    #pragma omp parallel for schedule(static, 64)
    for (int i = 0; i < n; ++i) {
      double r = 0;
      for (int64_t k = c->row_ptr[i]; k < c->row_ptr[i+1]; ++k) r += c->val[k] * xt[c->col[k]];
      yt[i] = r;
    }
  }
  for (int i = 0; i < n; ++i) y[c->perm[i]] += yt[i];
  T[2] = omp_get_wtime();
  /* P2M (:254-260) */
  #pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < c->n_p2m; ++i) {
    const orc_box *b = &c->boxes[c->p2m[i]];
    cplx *M0 = c->M + ((size_t)c->p2m[i]*2 + 0)*S, *M1 = M0 + S;
    for (uint32_t j = b->bb; j < b->be; ++j)
      orc_p2m_panel(t, &c->panels[c->perm[j]], c->nq, c->qw, x[c->perm[j]], b->center, M0, M1);
  }
  T[3] = omp_get_wtime();
  /* M2M (:262-267): serial, post-order */
  for (int i = 0; i < c->n_m2m; ++i) {
    const int ch = c->m2m[i].first, pa = c->m2m[i].second;
    double tr[3]; for (int k = 0; k < 3; ++k) tr[k] = c->boxes[pa].center[k] - c->boxes[ch].center[k];
    for (int e = 0; e < 2; ++e)
      if (use[e]) orc_m2m(t, c->M + ((size_t)ch*2 + e)*S, c->M + ((size_t)pa*2 + e)*S, tr);
  }
  T[4] = omp_get_wtime();
  /* M2L (:269-283), grouped by target */
  #pragma omp parallel for schedule(dynamic, 8)
  for (int b = 0; b < nb; ++b) {
    for (int i = c->lr_ptr[b]; i < c->lr_ptr[b+1]; ++i) {
      const int s = c->lr_src[i];
      double tr[3]; for (int k = 0; k < 3; ++k) tr[k] = c->boxes[b].center[k] - c->boxes[s].center[k];
      for (int e = 0; e < 2; ++e)
        if (use[e]) orc_m2l(t, c->M + ((size_t)s*2 + e)*S, c->L + ((size_t)b*2 + e)*S, tr);
    }
  }
  T[5] = omp_get_wtime();
  /* L2L (:285-292): serial, pre-order */
  for (int i = 0; i < c->n_l2l; ++i) {
    const int pa = c->l2l[i].first, ch = c->l2l[i].second;
    double tr[3]; for (int k = 0; k < 3; ++k) tr[k] = c->boxes[ch].center[k] - c->boxes[pa].center[k];
    for (int e = 0; e < 2; ++e)
      if (use[e]) orc_l2l(t, c->L + ((size_t)pa*2 + e)*S, c->L + ((size_t)ch*2 + e)*S, tr);
  }
  T[6] = omp_get_wtime();
  /* L2P (:294-300) */
  #pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < c->n_l2p; ++i) {
    const orc_box *b = &c->boxes[c->l2p[i]];
    const cplx *L0 = c->L + ((size_t)c->l2p[i]*2 + 0)*S, *L1 = L0 + S;
    for (uint32_t j = b->bb; j < b->be; ++j)
      orc_l2p_panel(t, L0, L1, b->center, &c->panels[c->perm[j]], &y[c->perm[j]]);
  }
  T[7] = omp_get_wtime();
  if (stage_s) { for (int i = 0; i < 7; ++i) stage_s[i] = T[i+1] - T[i]; stage_s[7] = T[7] - T[0]; }
  free(xt); free(yt);
  orc_tables_destroy(t);
  return 0;
}

/* y = A_near x only (the sparse_local near matrix), original order */
void orc_near_only(const orc_ctx *c, const double *x, double *y) {
  const int n = c->n;
  #pragma omp parallel for schedule(static, 64)
  for (int i = 0; i < n; ++i) {
    double r = 0;
    for (int64_t k = c->row_ptr[i]; k < c->row_ptr[i+1]; ++k) r += c->val[k] * x[c->perm[c->col[k]]];
    y[c->perm[i]] = r;
  }
}

/* include/Direct.hpp:99-125 (asymmetric Direct::eval): r_i += sum_j K(t_i, s_j) c_j, original order.
 * Rows [row_begin,row_end) only, so large cases can be sampled. y[row] is OVERWRITTEN. */
void orc_direct(const orc_ctx *c, const double *x, double *y, int row_begin, int row_end) {
  #pragma omp parallel for schedule(dynamic, 4)
  for (int i = row_begin; i < row_end; ++i) {
    double r = 0;
    for (int j = 0; j < c->n; ++j) r += orc_kernel(c, &c->panels[i], &c->panels[j]) * x[j];
    y[i] = r;
  }
}

/* The same sum (include/Direct.hpp:99-125, error over chosen bodies as tests/scaling.cpp:56-74 forms it) on a LIST of
 * target rows, original panel order: y_out[k] = row rows[k].  Test tooling: lets a seeded sample cover the whole vector. */
void orc_direct_rows(const orc_ctx *c, const double *x, double *y_out, int nrows, const int32_t *rows) {
  #pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < nrows; ++k) {
    const int i = rows[k];
    double r = 0;
    for (int j = 0; j < c->n; ++j) r += orc_kernel(c, &c->panels[i], &c->panels[j]) * x[j];
    y_out[k] = r;
  }
}
