/* api.c -- CPU ORACLE (test infrastructure): flat-array accessors for the ctypes wrapper
 * (oracle/oracle.py).  No algorithm lives here.  See fmm_oracle.h for the rules. */
#include "fmm_oracle.h"
#include <string.h>

void orc_get_perm(const orc_ctx *c, uint32_t *out) { memcpy(out, c->perm, sizeof(uint32_t)*(size_t)c->n); }

void orc_get_boxes(const orc_ctx *c, double *center, double *side, int32_t *level, int32_t *leaf,
                   int32_t *parent, int32_t *cb, int32_t *ce, int32_t *bb, int32_t *be) {
  for (int b = 0; b < c->nboxes; ++b) {
    const orc_box *x = &c->boxes[b];
    memcpy(center + 3*b, x->center, sizeof(double)*3);
    side[b] = x->side; level[b] = x->level; leaf[b] = x->leaf; parent[b] = (int32_t)x->parent;
    cb[b] = (int32_t)x->cb; ce[b] = (int32_t)x->ce; bb[b] = (int32_t)x->bb; be[b] = (int32_t)x->be;
  }
}

/* which: 0 p2p(src,tgt) 1 lr(src,tgt) 2 m2m(child,parent) 3 l2l(parent,child) */
int orc_get_pairs(const orc_ctx *c, int which, int32_t *out) {
  const orc_pair *p; int n;
  switch (which) {
    case 0: p = c->p2p; n = c->n_p2p; break;
    case 1: p = c->lr;  n = c->n_lr;  break;
    case 2: p = c->m2m; n = c->n_m2m; break;
    case 3: p = c->l2l; n = c->n_l2l; break;
    default: return -1;
  }
  if (out) for (int i = 0; i < n; ++i) { out[2*i] = p[i].first; out[2*i+1] = p[i].second; }
  return n;
}
/* which: 0 p2m leaves, 1 l2p leaves */
int orc_get_list(const orc_ctx *c, int which, int32_t *out) {
  const int *p = which == 0 ? c->p2m : c->l2p; int n = which == 0 ? c->n_p2m : c->n_l2p;
  if (out) for (int i = 0; i < n; ++i) out[i] = p[i];
  return n;
}

int64_t orc_near_nnz(const orc_ctx *c) { return c->nnz; }
void orc_get_near(const orc_ctx *c, int64_t *row_ptr, uint32_t *col, double *val) {
  if (row_ptr) memcpy(row_ptr, c->row_ptr, sizeof(int64_t)*((size_t)c->n + 1));
  if (col) memcpy(col, c->col, sizeof(uint32_t)*(size_t)c->nnz);
  if (val) memcpy(val, c->val, sizeof(double)*(size_t)c->nnz);
}

/* expansions of the last matvec at order P: out = [nboxes][2][P(P+1)/2] complex as (re,im) pairs */
void orc_get_expansions(const orc_ctx *c, int P, int which, double *out) {
  const cplx *src = which == 0 ? c->M : c->L;
  memcpy(out, src, sizeof(cplx)*(size_t)c->nboxes*2*(size_t)(P*(P+1)/2));
}

void orc_get_panels(const orc_ctx *c, double *center, double *normal, double *area, double *quad) {
  for (int i = 0; i < c->n; ++i) {
    const orc_panel *p = &c->panels[i];
    memcpy(center + 3*i, p->c, sizeof(double)*3);
    memcpy(normal + 3*i, p->n, sizeof(double)*3);
    area[i] = p->area;
    if (quad) memcpy(quad + 3*(size_t)c->nq*i, p->q, sizeof(double)*3*(size_t)c->nq);
  }
}

/* K(t_i, s_j) for explicit panel index pairs (original order) */
void orc_kernel_entries(const orc_ctx *c, int npairs, const int32_t *ti, const int32_t *sj, double *out) {
  for (int k = 0; k < npairs; ++k) out[k] = orc_kernel(c, &c->panels[ti[k]], &c->panels[sj[k]]);
}

void orc_get_tables(const orc_tables *t, double *prefactor, double *Anm, double *Cnm) {
  const int P = t->P;
  if (prefactor) memcpy(prefactor, t->prefactor, sizeof(double)*4*P*P);
  if (Anm) memcpy(Anm, t->Anm, sizeof(double)*4*P*P);
  if (Cnm) memcpy(Cnm, t->Cnm, sizeof(cplx)*(size_t)P*P*P*P);
}

#ifdef _OPENMP
#include <omp.h>
int orc_num_threads(void) { return omp_get_max_threads(); }
/* torchrun exports OMP_NUM_THREADS=1 to every rank: the checker's Direct sum on rank 0 may still use the cores it has */
void orc_set_num_threads(int n) { if (n > 0) omp_set_num_threads(n); }
#else
int orc_num_threads(void) { return 1; }
void orc_set_num_threads(int n) { (void)n; }
#endif
