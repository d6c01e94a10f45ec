/* tree.c -- CPU ORACLE (test infrastructure): Morton octree, dual tree traversal and the lazy
 * operator lists of EvalInteractionLazySparse.  See fmm_oracle.h for the rules. */
#include "fmm_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define LEVELS 10u   /* include/tree/Octree.hpp:88 (MortonCoder::levels) */
/* NOT A REFERENCE RULE: when a box on level 10 still holds more than ncrit bodies the reference's coder is out of bits (its
 * shift 3*(levels - level - 1) wraps, Octree.hpp:649) and no tree exists.  The product then codes with 21 bits per dimension
 * in a 64-bit key; the oracle follows with the same switch so that the two can be compared on such meshes.  Every tree the
 * reference CAN build takes the 10-level path below unchanged. */
#define DEEP_LEVELS 21u

/* include/tree/Octree.hpp:143-159 (spread_bits, interleave), widened to 64 bits: on 10-bit inputs the values are the
 * reference's 32-bit ones */
static uint64_t spread_bits(uint64_t x) {
  x &= 0x1FFFFFull;
  x = (x | (x << 32)) & 0x1F00000000FFFFull;
  x = (x | (x << 16)) & 0x1F0000FF0000FFull;
  x = (x | (x <<  8)) & 0x100F00F00F00F00Full;
  x = (x | (x <<  4)) & 0x10C30C30C30C30C3ull;
  x = (x | (x <<  2)) & 0x1249249249249249ull;
  return x;
}
static uint64_t interleave(uint64_t x, uint64_t y, uint64_t z) {
  return spread_bits(x) | (spread_bits(y) << 1) | (spread_bits(z) << 2);
}
/* include/tree/Octree.hpp:167-174 (compact_bits) */
static uint64_t compact_bits(uint64_t x) {
  x &= 0x1249249249249249ull;
  x = (x | (x >>  2)) & 0x10C30C30C30C30C3ull;
  x = (x | (x >>  4)) & 0x100F00F00F00F00Full;
  x = (x | (x >>  8)) & 0x1F0000FF0000FFull;
  x = (x | (x >> 16)) & 0x1F00000000FFFFull;
  x = (x | (x >> 32)) & 0x1FFFFFull;
  return x;
}

static int key_level(uint64_t key) {          /* box_data::level(), Octree.hpp:226-238 */
  int hb = 63 - __builtin_clzll(key);
  return hb / 3;
}

typedef struct { uint64_t code; uint32_t idx; } code_pair;

#define VEC(T) struct { T *d; size_t n, cap; }
#define PUSH(v, x) do { if ((v).n == (v).cap) { (v).cap = (v).cap ? 2*(v).cap : 1024; \
      (v).d = realloc((v).d, (v).cap*sizeof(*(v).d)); } (v).d[(v).n++] = (x); } while (0)

static void box_geometry(orc_ctx *c, orc_box *b) {
  /* Box::center(), Octree.hpp:350-355, through MortonCoder::cell (:109-113) and
     box_data::get_mc_lower_bound (:243-248); Box::side_length (:334-336) */
  const unsigned L = c->levels;
  uint64_t m = b->key, top = (uint64_t)1 << (3*L);
  while (!(m & top)) m <<= 3;
  uint64_t lower = m & ~top;
  double ix[3] = { (double)compact_bits(lower), (double)compact_bits(lower >> 1), (double)compact_bits(lower >> 2) };
  for (int k = 0; k < 3; ++k) {
    double lo = c->pmin[k] + c->cell[k]*ix[k];
    double hi = lo + c->cell[k];
    double dim = hi - lo;
    b->center[k] = lo + dim * ldexp(1.0, (int)L - 1 - b->level);   /* = 1 << (10-level-1) for level <= 9 */
  }
  double bbmax0 = c->pmin[0] + ldexp(1.0, (int)L) * c->cell[0];   /* MortonCoder::bounding_box, :102-105 */
  b->side = (bbmax0 - c->pmin[0]) / (double)(1 << b->level);
}

/* include/FMMOptions.hpp:21-31 (DefaultMAC); radius = side/2 (Octree.hpp:340-342) */
static int mac(const orc_ctx *c, const orc_box *b1, const orc_box *b2) {
  double d[3] = { b1->center[0]-b2->center[0], b1->center[1]-b2->center[1], b1->center[2]-b2->center[2] };
  double r0 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
  double rhs = (b1->side/2.0 + b2->side/2.0) / c->theta;
  return r0 > rhs*rhs;
}

typedef VEC(orc_pair) pair_vec;
typedef VEC(int) int_vec;

/* EvalInteractionLazySparse.hpp:173-194 (resolve_multipole) */
static void resolve_multipole(orc_ctx *c, int b, char *initM, int_vec *p2m, pair_vec *m2m) {
  if (initM[b]) return;
  const orc_box *bx = &c->boxes[b];
  if (bx->leaf) {
    PUSH(*p2m, b);
  } else {
    for (uint32_t ch = bx->cb; ch < bx->ce; ++ch) {
      resolve_multipole(c, (int)ch, initM, p2m, m2m);
      orc_pair pr = { (int)ch, b };
      PUSH(*m2m, pr);
    }
  }
  initM[b] = 1;
}
/* EvalInteractionLazySparse.hpp:199-220 (propagate_local) */
static void propagate_local(orc_ctx *c, int b, char *initL, char *l2pset, int_vec *l2p, pair_vec *l2l) {
  const orc_box *bx = &c->boxes[b];
  if (bx->leaf) {
    if (!l2pset[b]) { PUSH(*l2p, b); l2pset[b] = 1; }
  } else {
    for (uint32_t ch = bx->cb; ch < bx->ce; ++ch) {
      if (!initL[ch]) {
        initL[ch] = 1;
        orc_pair pr = { b, (int)ch };
        PUSH(*l2l, pr);
        propagate_local(c, (int)ch, initL, l2pset, l2p, l2l);
      } else {
        c->n_l2l_skipped++;
      }
    }
  }
}

orc_ctx *orc_create(int n, const double *verts, const uint8_t *bc, int K, double theta, unsigned ncrit) {
  return orc_create_eval(n, verts, bc, K, theta, ncrit, ORC_EVAL_FMM);
}

/* evaluator: which of make_evaluators' branches (executor/make_executor.hpp:24-60) the plan takes:
 *   ORC_EVAL_FMM             EvalInteractionLazySparse (near matrix + far field)
 *   ORC_EVAL_LOCAL           EvalLocalSparse.hpp:34-86: same traversal, accepted multipoles ignored (:120-127)
 *   ORC_EVAL_BLOCK_DIAGONAL  EvalDiagonalSparse.hpp:33-49: every leaf with itself, in box order */
orc_ctx *orc_create_eval(int n, const double *verts, const uint8_t *bc, int K, double theta, unsigned ncrit, int evaluator) {
  double qp[ORC_MAXK][3];
  orc_ctx *c = calloc(1, sizeof(*c));
  c->n = n; c->K = K; c->theta = theta; c->ncrit = ncrit;
  c->nq = orc_quadrature(K, qp, c->qw);
  if (c->nq < 0 || n <= 0) { free(c); return NULL; }
  c->panels = malloc(sizeof(orc_panel)*(size_t)n);
  c->quad = malloc(sizeof(double)*3*(size_t)c->nq*(size_t)n);
  for (int i = 0; i < n; ++i)
    orc_panel_init(&c->panels[i], verts + 9*(size_t)i, bc ? bc[i] : 0, c->nq, qp, c->quad + 3*(size_t)c->nq*(size_t)i);

  /* ---- bounding box: Octree.hpp:67-79 ---- */
  double mn[3], mx[3];
  for (int k = 0; k < 3; ++k) mn[k] = mx[k] = c->panels[0].c[k];
  for (int i = 1; i < n; ++i)
    for (int k = 0; k < 3; ++k) {
      mn[k] = fmin(mn[k], c->panels[i].c[k]); mx[k] = fmax(mx[k], c->panels[i].c[k]);
    }
  double ext = fmax(fabs(mx[0]-mn[0]), fmax(fabs(mx[1]-mn[1]), fabs(mx[2]-mn[2])));   /* norm_inf(dimensions) */
  for (int k = 0; k < 3; ++k) {
    double a = mn[k] + ext*(1 + 1e-6);
    mx[k] = fmax(mx[k], a);
    c->pmin[k] = mn[k];
  }
  c->levels = LEVELS;
  code_pair *codes = malloc(sizeof(code_pair)*(size_t)n), *tmp = malloc(sizeof(code_pair)*(size_t)n);
  VEC(orc_box) boxes = {0};
  int_vec level_offset = {0};
rebuild:;
  const unsigned L = c->levels;
  for (int k = 0; k < 3; ++k) c->cell[k] = (mx[k] - mn[k]) / ldexp(1.0, (int)L);     /* MortonCoder ctor, :95-99 */

  /* ---- codes: MortonCoder::code, :118-129 ---- */
  for (int i = 0; i < n; ++i) {
    uint64_t s[3];
    for (int k = 0; k < 3; ++k) {
      double v = c->panels[i].c[k];
      v -= c->pmin[k]; v /= c->cell[k];
      s[k] = (uint64_t)(uint32_t)v;
    }
    codes[i].code = interleave(s[0], s[1], s[2]); codes[i].idx = (uint32_t)i;
  }

  /* ---- construct_tree: Octree.hpp:617-692 (incremental stable bucket sort, BFS box order) ---- */
  boxes.n = 0; level_offset.n = 0;
  orc_box root; memset(&root, 0, sizeof root);
  root.key = 1; root.parent = 0; root.cb = 0; root.ce = (uint32_t)n; root.level = 0; root.bb = 0; root.be = (uint32_t)n;
  PUSH(boxes, root);
  PUSH(level_offset, 0);
  int maxlevel = 0;
  for (size_t k = 0; k != boxes.n; ++k) {
    orc_box bk = boxes.d[k];
    if (bk.ce - bk.cb <= ncrit) { boxes.d[k].leaf = 1; continue; }       /* :641-644 */
    if (bk.level >= (int)L) {                                            /* 32-bit key limit, :85-92 */
      if (L == LEVELS) { c->levels = DEEP_LEVELS; goto rebuild; }        /* not a reference rule: see DEEP_LEVELS */
      fprintf(stderr, "oracle: octree deeper than %u levels\n", L);
      boxes.d[k].leaf = 1; continue;
    }
    unsigned shift = 3*(L - (unsigned)bk.level - 1);                    /* :649 */
    size_t cnt[9] = {0};
    for (uint32_t i = bk.cb; i < bk.ce; ++i) cnt[((codes[i].code >> shift) & 7) + 1]++;
    for (int b = 0; b < 8; ++b) cnt[b+1] += cnt[b];
    size_t pos[8]; for (int b = 0; b < 8; ++b) pos[b] = cnt[b];
    for (uint32_t i = bk.cb; i < bk.ce; ++i) tmp[bk.cb + pos[(codes[i].code >> shift) & 7]++] = codes[i];   /* stable */
    memcpy(codes + bk.cb, tmp + bk.cb, sizeof(code_pair)*(bk.ce - bk.cb));
    uint32_t first_child = (uint32_t)boxes.n, nchild = 0;
    for (int ch = 0; ch < 8; ++ch) {                                     /* :661-681 */
      uint32_t bch = bk.cb + (uint32_t)cnt[ch], ech = bk.cb + (uint32_t)cnt[ch+1];
      if (ech - bch > 0) {
        orc_box nb; memset(&nb, 0, sizeof nb);
        nb.key = (bk.key << 3) | (uint64_t)ch; nb.parent = (uint32_t)k;
        nb.cb = bch; nb.ce = ech; nb.bb = bch; nb.be = ech;
        nb.level = key_level(nb.key);
        if (nb.level > maxlevel) { maxlevel = nb.level; PUSH(level_offset, (int)boxes.n); }
        PUSH(boxes, nb);
        ++nchild;
      }
    }
    boxes.d[k].cb = first_child; boxes.d[k].ce = first_child + nchild;
  }
  PUSH(level_offset, (int)boxes.n);                                       /* :684 */
  c->nboxes = (int)boxes.n; c->boxes = boxes.d;
  c->nlevels = (int)level_offset.n - 1; c->level_offset = level_offset.d;
  c->perm = malloc(sizeof(uint32_t)*(size_t)n); c->code = malloc(sizeof(uint64_t)*(size_t)n);
  for (int i = 0; i < n; ++i) { c->perm[i] = codes[i].idx; c->code[i] = codes[i].code; }    /* :687-691 */
  free(codes); free(tmp);
  for (int b = 0; b < c->nboxes; ++b) box_geometry(c, &c->boxes[b]);

  /* ---- dual tree traversal: EvalInteractionLazySparse.hpp:68-110, interact :239-252 ---- */
  pair_vec q = {0}, p2p = {0}, lr = {0};
  size_t head = 0;
  orc_pair rr = {0, 0};
  if (evaluator == ORC_EVAL_BLOCK_DIAGONAL) {
    for (int b = 0; b < c->nboxes; ++b)
      if (c->boxes[b].leaf) { orc_pair pr = {b, b}; PUSH(p2p, pr); }
  } else PUSH(q, rr);
  while (head < q.n) {
    orc_pair pr = q.d[head++];
    const orc_box *b1 = &c->boxes[pr.first], *b2 = &c->boxes[pr.second];
    int split_first;
    if (b1->leaf) {
      if (b2->leaf) { PUSH(p2p, pr); continue; }
      split_first = 0;
    } else if (b2->leaf) split_first = 1;
    else split_first = (b1->side > b2->side);                          /* ties split b2, :98-108 */
    const orc_box *sp = split_first ? b1 : b2;
    for (uint32_t ch = sp->cb; ch < sp->ce; ++ch) {
      orc_pair np = split_first ? (orc_pair){ (int)ch, pr.second } : (orc_pair){ pr.first, (int)ch };
      if (mac(c, &c->boxes[np.first], &c->boxes[np.second])) { if (evaluator == ORC_EVAL_FMM) PUSH(lr, np); } else PUSH(q, np);
    }
    if (head > (1u << 20) && head*2 > q.n) {       /* compact the FIFO */
      memmove(q.d, q.d + head, sizeof(orc_pair)*(q.n - head)); q.n -= head; head = 0;
    }
  }
  free(q.d);
  c->p2p = p2p.d; c->n_p2p = (int)p2p.n; c->lr = lr.d; c->n_lr = (int)lr.n;

  /* ---- resolve_LR_interactions: :225-237 ---- */
  char *initM = calloc((size_t)c->nboxes, 1), *initL = calloc((size_t)c->nboxes, 1), *l2pset = calloc((size_t)c->nboxes, 1);
  int_vec p2m = {0}, l2p = {0}; pair_vec m2m = {0}, l2l = {0};
  for (int i = 0; i < c->n_lr; ++i) {
    resolve_multipole(c, c->lr[i].first, initM, &p2m, &m2m);
    if (!initL[c->lr[i].second]) {
      initL[c->lr[i].second] = 1;
      propagate_local(c, c->lr[i].second, initL, l2pset, &l2p, &l2l);
    }
  }
  free(initM); free(initL); free(l2pset);
  c->p2m = p2m.d; c->n_p2m = (int)p2m.n; c->m2m = m2m.d; c->n_m2m = (int)m2m.n;
  c->l2l = l2l.d; c->n_l2l = (int)l2l.n; c->l2p = l2p.d; c->n_l2p = (int)l2p.n;

  /* ---- LR list regrouped by target; sources kept in reference (traversal) order ---- */
  c->lr_ptr = calloc((size_t)c->nboxes + 1, sizeof(int));
  c->lr_src = malloc(sizeof(int)*(size_t)(c->n_lr ? c->n_lr : 1));
  for (int i = 0; i < c->n_lr; ++i) c->lr_ptr[c->lr[i].second + 1]++;
  for (int b = 0; b < c->nboxes; ++b) c->lr_ptr[b+1] += c->lr_ptr[b];
  int *fill = malloc(sizeof(int)*(size_t)c->nboxes);
  memcpy(fill, c->lr_ptr, sizeof(int)*(size_t)c->nboxes);
  for (int i = 0; i < c->n_lr; ++i) c->lr_src[fill[c->lr[i].second]++] = c->lr[i].first;
  free(fill);
  return c;
}

/* NOT a reference rule.  Replaces the L2L list of resolve_LR_interactions (which omits n_l2l_skipped edges, see
 * propagate_local above) by the complete one: every child of a box that holds a local expansion, parents first.  The
 * product's default downward pass (FMMBEM_L2L_COMPLETE) is checked against the oracle in this mode; with
 * n_l2l_skipped == 0 it is the same set of operations. */
void orc_complete_l2l(orc_ctx *c) {
  char *hasL = calloc((size_t)c->nboxes, 1);
  for (int i = 0; i < c->n_lr; ++i) hasL[c->lr[i].second] = 1;
  pair_vec l2l = {0};
  for (int b = 1; b < c->nboxes; ++b)            /* BFS box order: a parent precedes its children */
    if (hasL[c->boxes[b].parent]) {
      hasL[b] = 1;
      orc_pair pr = { (int)c->boxes[b].parent, b };
      PUSH(l2l, pr);
    }
  free(hasL); free(c->l2l);
  c->l2l = l2l.d; c->n_l2l = (int)l2l.n;
}

void orc_destroy(orc_ctx *c) {
  if (!c) return;
  free(c->panels); free(c->quad); free(c->boxes); free(c->perm); free(c->code); free(c->level_offset);
  free(c->p2p); free(c->lr); free(c->p2m); free(c->m2m); free(c->l2l); free(c->l2p);
  free(c->lr_ptr); free(c->lr_src); free(c->row_ptr); free(c->col); free(c->val);
  free(c->M); free(c->L); free(c->val9); free(c->MS); free(c->LS); free(c);
}

void orc_stats(const orc_ctx *c, int64_t out[16]) {
  int leaves = 0;
  for (int b = 0; b < c->nboxes; ++b) leaves += c->boxes[b].leaf;
  int64_t nnz = 0;
  for (int i = 0; i < c->n_p2p; ++i) {
    const orc_box *s = &c->boxes[c->p2p[i].first], *t = &c->boxes[c->p2p[i].second];
    nnz += (int64_t)(s->be - s->bb) * (int64_t)(t->be - t->bb);
  }
  out[0] = c->n; out[1] = c->nboxes; out[2] = leaves; out[3] = c->nlevels; out[4] = nnz;
  out[5] = c->n_lr; out[6] = c->n_m2m; out[7] = c->n_l2l; out[8] = c->n_p2m; out[9] = c->n_l2p;
  out[10] = c->n_p2p; out[11] = c->n_l2l_skipped; out[12] = c->nq;
  out[13] = out[14] = out[15] = 0;
}
