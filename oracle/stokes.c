/* stokes.c -- CPU ORACLE (test infrastructure): the StokesSphericalBEM path (velocity boundary condition:
 * stokeslet single layer, the operator the reference's StokesBEM solve uses).  The traction (stresslet)
 * operator is NOT restated: the reference's own FMM for it disagrees with its Direct sum (SURVEY.md 8a note).
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

#define CI (_Complex_I)
static double norm3(const double a[3]) { return sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }

/* examples/BEM/Triangulation.hpp:184-255 (ConvertRedBloodCellTriangle + RedBloodCell, identity rotation, no shift):
 * the unit-sphere triangulation with every vertex mapped onto the biconcave disc. verts: N x 9 in/out. */
static int sgn(double v) { return (0 < v) - (v < 0); }
void orc_red_blood_cell_map(long n, double *verts) {
  const double r = 3.91, C0 = 0.81, C2 = 7.83, C4 = -4.39;
  for (long i = 0; i < 3*n; ++i) {
    double *v = verts + 3*i;
    double x = v[0]*r, y = v[1]*r;
    double rho = sqrt(x*x + y*y);
    double ratio = rho / r;
    double z = sqrt(1 - ratio*ratio + 1e-12)*(C0 + C2*ratio*ratio + C4*ratio*ratio*ratio*ratio)*0.5*sgn(v[2]);
    v[0] = x; v[1] = y; v[2] = z;
  }
}

/* Self term: AnalyticalIntegral::FataAnalytical<STOKES>(y1,y2,y3,f,x=centroid,self=true,G)
 * (examples/BEM/FataAnalytical.hpp:414-690, self_interaction branch :535-539) followed by
 * Integration<STOKES>::integrate, type G (:273-341).  With the collocation point in the panel plane
 * (et = 0) and chi left at its initial {0,0,0} in the self branch, the routine reduces to the lines below. */
static void stokes_self(const orc_panel *s, double IU[9]) {
  const double *y1 = s->v[0], *y2 = s->v[1], *y3 = s->v[2], *x = s->c;
  const double pi = M_PI, gm_eps = 1e-10;
  double v1[3], v3[3], e1[3], e2[3], e3[3];
  for (int k = 0; k < 3; ++k) { v1[k] = y2[k] - y1[k]; v3[k] = y3[k] - y1[k]; }      /* :453-457 */
  /* ortho_comp_basis (:154-187) */
  double snrm = v1[0]*v1[0] + v1[1]*v1[1] + v1[2]*v1[2], nrm = sqrt(snrm);
  double al = (v1[0]*v3[0] + v1[1]*v3[1] + v1[2]*v3[2])/snrm;
  for (int i = 0; i < 3; ++i) e2[i] = v3[i] - al*v1[i];
  double nrx = norm3(e2);
  for (int i = 0; i < 3; ++i) { e1[i] = v1[i]/nrm; e2[i] = e2[i]/nrx; }
  e3[0] = e1[1]*e2[2] - e2[1]*e1[2]; e3[1] = e1[2]*e2[0] - e2[2]*e1[0]; e3[2] = e1[0]*e2[1] - e2[0]*e1[1];
  double bQ = v1[0]*e1[0] + v1[1]*e1[1] + v1[2]*e1[2];                                 /* :463-465 */
  double aQ = v3[0]*e2[0] + v3[1]*e2[1] + v3[2]*e2[2];
  double cQ = v3[0]*e1[0] + v3[1]*e1[1] + v3[2]*e1[2];
  double bmc = bQ - cQ, aQs = aQ*aQ;
  double theta0_ = acos(cQ/sqrt(cQ*cQ + aQs)), theta1_ = acos(bmc/sqrt(bmc*bmc + aQs));  /* :470 */
  double alpha2 = pi - theta1_, alpha3 = pi + theta0_;                                 /* :473 */
  double cscf2 = cos(alpha2), sncf2 = sin(alpha2), cscf3 = cos(alpha3), sncf3 = sin(alpha3);
  double r1[3]; for (int k = 0; k < 3; ++k) r1[k] = x[k] - y1[k];
  double xi = r1[0]*e1[0] + r1[1]*e1[1] + r1[2]*e1[2];
  double zt = r1[0]*e2[0] + r1[1]*e2[1] + r1[2]*e2[2];
  double eth = r1[0]*e3[0] + r1[1]*e3[1] + r1[2]*e3[2];
  double q[3], p11 = -xi, p12 = bQ - xi, et = -eth;
  q[0] = -zt;
  if (fabs(et) < gm_eps) et = 0.0;                                                     /* :496 */
  double x3 = cQ + p11, z3 = aQ + q[0];
  double p22 = p12*cscf2 + q[0]*sncf2, p23 = x3*cscf2 + z3*sncf2; q[1] = q[0]*cscf2 - p12*sncf2;
  double p31 = p11*cscf3 + q[0]*sncf3, p33 = x3*cscf3 + z3*sncf3; q[2] = q[0]*cscf3 - p11*sncf3;
  double ets = et*et;
  double Rhs0 = p11*p11 + q[0]*q[0], Rhs1 = p12*p12 + q[0]*q[0], Rhs2 = p33*p33 + q[2]*q[2];
  double rho[3] = { sqrt(Rhs0 + ets), sqrt(Rhs1 + ets), sqrt(Rhs2 + ets) };
  /* shc clamps (:510-519) and the vertex cases (:524-538) cannot trigger for the centroid (shc = 1/3) */
  double omega = q[0]*log((p11+rho[0])/(p12+rho[1])) + q[1]*log((p22+rho[1])/(p23+rho[2]))
               + q[2]*log((p33+rho[2])/(p31+rho[0]));                                  /* :538 */
  const double alpha[3] = {0., alpha2, alpha3};
  /* integrate (:297-341) with chi = 0, gam = 0; ThetGam only ever multiplies et (= 0 here) */
  double ThetGam = 0.0;
  double I1 = omega - et*ThetGam, etI3 = ThetGam;
  double rho_bar[3] = { rho[0]-rho[1], rho[1]-rho[2], rho[2]-rho[0] };
  double I3_xi = 0, I3_zeta = 0, I3_xi_xi = 0, I3_zeta_zeta = 0, I3_zeta_xi = 0;
  for (int i = 0; i < 3; ++i) {
    I3_xi_xi     += (rho_bar[i]*sin(alpha[i]))*cos(alpha[i]);
    I3_zeta_zeta += (-rho_bar[i]*cos(alpha[i]))*sin(alpha[i]);
    I3_zeta_xi   += (rho_bar[i]*sin(alpha[i]))*sin(alpha[i]);
  }
  I3_xi_xi -= et*ThetGam; I3_zeta_zeta -= et*ThetGam;
  double I3_xi_zeta = I3_zeta_xi;
  memset(IU, 0, sizeof(double)*9);
  const double *E[3] = { e1, e2, e3 };
  const double coef[3][3] = { { I1 + I3_xi_xi, I3_xi_zeta, et*I3_xi },
                              { I3_xi_zeta, I1 + I3_zeta_zeta, et*I3_zeta },
                              { et*I3_xi, et*I3_zeta, I1 + et*etI3 } };
  for (int a = 0; a < 3; ++a) for (int b = 0; b < 3; ++b)
    for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) IU[3*i + j] += coef[a][b]*E[a][i]*E[b][j];
}

/* Gauss rule on the stokeslet: res += w*A/r^3 * (r^2 I + d d^T), d = target - point
 * (kernel/StokesSphericalBEM.hpp:302-321 with K_fine, :352-369 with the K stored points) */
static void stokeslet_point(double res[9], double wA, const double t[3], const double pnt[3]) {
  double d[3] = { t[0]-pnt[0], t[1]-pnt[1], t[2]-pnt[2] };
  double r2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
  double invR2 = 1. / r2;
  if (r2 < 1e-8) invR2 = 0;
  double f = wA*invR2*sqrt(invR2);
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) res[3*i + j] += f*((i == j ? r2 : 0.0) + d[i]*d[j]);
}

/* one Gauss point of the traction (double-layer) integrand: res += w*A (d.n) d d^T / r^5, d = target - point
 * (kernel/StokesSphericalBEM.hpp:205-225 with K_fine, :236-252 with the K stored points) */
static void stresslet_point(double res[9], double wA, const double t[3], const double pnt[3], const double nrm[3]) {
  double d[3] = { t[0]-pnt[0], t[1]-pnt[1], t[2]-pnt[2] };
  double r2 = d[0]*d[0] + d[1]*d[1] + d[2]*d[2];
  double invR2 = 1. / r2;
  if (r2 < 1e-8) invR2 = 0;
  double invR5 = invR2*invR2*sqrt(invR2);
  double dn = d[0]*nrm[0] + d[1]*nrm[1] + d[2]*nrm[2];
  double f = wA*dn*invR5;
  for (int i = 0; i < 3; ++i) for (int j = 0; j < 3; ++j) res[3*i + j] += f*d[i]*d[j];
}

/* kernel/StokesSphericalBEM.hpp:160-258 (eval_traction_integral): self -> 2 pi I (:167-174); near -> K_fine rule; far -> the
 * K stored points; times -3 (:227, :254).  No 1/(2 mu): the reference's traction kernel carries none. */
static void stokes_traction_entry(const orc_ctx *c, const orc_panel *t, const orc_panel *s, double out[9]) {
  double dd[3] = { t->c[0]-s->c[0], t->c[1]-s->c[1], t->c[2]-s->c[2] };
  double d = norm3(dd);
  memset(out, 0, sizeof(double)*9);
  if (fabs(d) < 1e-8) { out[0] = out[4] = out[8] = 2*M_PI; return; }
  if (sqrt(2*s->area)/d >= 0.5) {
    for (int i = 0; i < c->nqf; ++i) {
      double pnt[3];
      for (int k = 0; k < 3; ++k) pnt[k] = s->v[0][k]*c->qfp[i][0] + s->v[1][k]*c->qfp[i][1] + s->v[2][k]*c->qfp[i][2];
      stresslet_point(out, c->qfw[i]*s->area, t->c, pnt, s->n);
    }
  } else {
    for (int i = 0; i < c->nq; ++i) stresslet_point(out, c->qw[i]*s->area, t->c, s->q[i], s->n);
  }
  for (int i = 0; i < 9; ++i) out[i] *= -3.;
}

/* kernel/StokesSphericalBEM.hpp:377-389 (operator(): the TARGET's flag picks the integral), :260-375 (eval_velocity_integral) */
void orc_stokes_entry(const orc_ctx *c, const orc_panel *t, const orc_panel *s, double out[9]) {
  if (t->bc) { stokes_traction_entry(c, t, s, out); return; }
  double dd[3] = { t->c[0]-s->c[0], t->c[1]-s->c[1], t->c[2]-s->c[2] };
  double d = norm3(dd);
  int self = d < 1e-8;
  memset(out, 0, sizeof(double)*9);
  if (sqrt(2*s->area)/d >= 0.5) {
    if (self) {
      stokes_self(s, out);
    } else {
      for (int i = 0; i < c->nqf; ++i) {
        double pnt[3];
        for (int k = 0; k < 3; ++k) pnt[k] = s->v[0][k]*c->qfp[i][0] + s->v[1][k]*c->qfp[i][1] + s->v[2][k]*c->qfp[i][2];
        stokeslet_point(out, c->qfw[i]*s->area, t->c, pnt);
      }
    }
  } else {
    for (int i = 0; i < c->nq; ++i) stokeslet_point(out, c->qw[i]*s->area, t->c, s->q[i]);
  }
  for (int i = 0; i < 9; ++i) out[i] *= 1./2/c->mu;
}

int orc_stokes_config(orc_ctx *c, double mu, int kfine) {
  c->mu = mu; c->kfine = kfine;
  c->nqf = orc_quadrature(kfine, c->qfp, c->qfw);
  return c->nqf < 0 ? -1 : 0;
}

/* near matrix of 3x3 blocks, same sparsity as the scalar case (executor/EvalP2P.hpp:47-98) */
int orc_stokes_build_near(orc_ctx *c) {
  if (c->val9) return 0;
  if (orc_build_near_pattern(c)) return -1;
  c->val9 = malloc(sizeof(double)*9*(size_t)(c->nnz ? c->nnz : 1));
  if (!c->val9) return -1;
  #pragma omp parallel for schedule(dynamic, 16)
  for (int i = 0; i < c->n; ++i) {
    const orc_panel *t = &c->panels[c->perm[i]];
    for (int64_t k = c->row_ptr[i]; k < c->row_ptr[i+1]; ++k)
      orc_stokes_entry(c, t, &c->panels[c->perm[c->col[k]]], c->val9 + 9*k);
  }
  return 0;
}

/* kernel/LaplaceSpherical.hpp:546-561 (sph2cart) */
static void sph2cart(double r, double theta, double phi, const double sp[3], double ca[3]) {
  ca[0] = sin(theta)*cos(phi)*sp[0] + cos(theta)*cos(phi)/r*sp[1] - sin(phi)/r/sin(theta)*sp[2];
  ca[1] = sin(theta)*sin(phi)*sp[0] + cos(theta)*sin(phi)/r*sp[1] + cos(phi)/r/sin(theta)*sp[2];
  ca[2] = cos(theta)*sp[0] - sin(theta)/r*sp[1];
}

/* kernel/StokesSphericalBEM.hpp:391-432 (P2M, VELOCITY branch): four harmonic moments f0,f1,f2,f.x */
static void stokes_p2m_panel(const orc_tables *t, const orc_ctx *c, const orc_panel *src, const double f[3],
                             const double center[3], cplx *M /* [4][S] */) {
  const int P = t->P, S = P*(P+1)/2;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  for (int i = 0; i < c->nq; ++i) {
    const double *qp = src->q[i];
    double dist[3] = { qp[0]-center[0], qp[1]-center[1], qp[2]-center[2] };
    double rho, alpha, beta;
    orc_cart2sph(&rho, &alpha, &beta, dist);
    orc_eval_multipole(t, rho, alpha, -beta, Ynm, YnmTheta);
    double f2[3] = { src->area*c->qw[i]*f[0], src->area*c->qw[i]*f[1], src->area*c->qw[i]*f[2] };
    double fdotx = f2[0]*qp[0] + f2[1]*qp[1] + f2[2]*qp[2];
    for (int n = 0; n != P; ++n)
      for (int m = 0; m <= n; ++m) {
        const int nm = n*(n+1) + m, nms = n*(n+1)/2 + m;
        M[0*S + nms] += f2[0]*Ynm[nm];
        M[1*S + nms] += f2[1]*Ynm[nm];
        M[2*S + nms] += f2[2]*Ynm[nm];
        M[3*S + nms] += fdotx*Ynm[nm];
      }
  }
}

/* kernel/StokesSpherical.hpp:318-401 (L2P, scale = 1) then StokesSphericalBEM.hpp:512-522 (x 1/(2 mu)) */
static void stokes_l2p_panel(const orc_tables *t, const orc_ctx *c, const cplx *L /* [4][S] */, const double center[3],
                             const orc_panel *tgt, double result[3]) {
  const int P = t->P, S = P*(P+1)/2;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  const double *target = tgt->c;
  double dist[3] = { target[0]-center[0], target[1]-center[1], target[2]-center[2] };
  double grad[4][3] = {{0}}, cart[3], res[3] = {0, 0, 0};
  double r, theta, phi;
  orc_cart2sph(&r, &theta, &phi, dist);
  orc_eval_multipole(t, r, theta, phi, Ynm, YnmTheta);
  for (int n = 0; n != P; ++n) {
    int nm = n*n + n, nms = n*(n+1)/2;
    for (int e = 0; e < 3; ++e) res[e] += creal(L[e*S + nms]*Ynm[nm]);
    double factor = 1. / r * n;
    for (int e = 0; e < 4; ++e) {
      grad[e][0] += creal(L[e*S + nms]*Ynm[nm])*factor;
      grad[e][1] += creal(L[e*S + nms]*YnmTheta[nm]);
    }
    for (int m = 1; m <= n; ++m) {
      nm = n*n + n + m; nms = n*(n+1)/2 + m;
      for (int e = 0; e < 3; ++e) res[e] += 2*creal(L[e*S + nms]*Ynm[nm]);
      for (int e = 0; e < 4; ++e) {
        grad[e][0] += 2*creal(L[e*S + nms]*Ynm[nm])*factor;
        grad[e][1] += 2*creal(L[e*S + nms]*YnmTheta[nm]);
        grad[e][2] += 2*creal(L[e*S + nms]*Ynm[nm]*CI)*m;
      }
    }
  }
  double g[4][3];
  for (int e = 0; e < 4; ++e) {
    sph2cart(r, theta, phi, grad[e], cart);
    for (int k = 0; k < 3; ++k) g[e][k] = e < 3 ? cart[k]*(-target[e]) : cart[k];
  }
  for (int k = 0; k < 3; ++k) res[k] += g[0][k] + g[1][k] + g[2][k] + g[3][k];
  for (int k = 0; k < 3; ++k) result[k] += 1./2/c->mu*res[k];
}

static void ensure_stokes_expansions(orc_ctx *c, int P) {
  if (c->spcap >= P) return;
  free(c->MS); free(c->LS);
  size_t sz = (size_t)c->nboxes*8*(size_t)(P*(P+1)/2);
  c->MS = malloc(sizeof(cplx)*sz); c->LS = malloc(sizeof(cplx)*sz);
  c->spcap = P;
}

/* Stokes matvec, stage order of executor/EvalInteractionLazySparse.hpp:120-168.  x, y: n x 3 (Vec<3,double> per
 * panel, original order).  Expansions [box][8][S]: slots 0-3 velocity group (M[0][0..3]), 4-7 traction group
 * (M[1][..], zero for velocity panels).  faithful: all 8 slots are translated, as the reference does. */
int orc_stokes_matvec(orc_ctx *c, int P, const double *x, double *y, int flags, double stage_s[8]) {
  if (P < 1 || P > ORC_PMAX) return -1;
  /* the far field below is the VELOCITY branch only; the reference's traction far field disagrees with its own Direct sum
   *  and is not restated: with traction panels only the near-field evaluators have an answer */
  if (c->n_lr > 0) for (int i = 0; i < c->n; ++i) if (c->panels[i].bc) return -3;
  if (orc_stokes_build_near(c)) return -2;
  const int faithful = flags & ORC_FLAG_FAITHFUL;
  const int n = c->n, nb = c->nboxes, S = P*(P+1)/2, E = faithful ? 8 : 4;
  orc_tables *t = orc_tables_create(P);
  ensure_stokes_expansions(c, P);
  double T[9]; T[0] = omp_get_wtime();
  memset(c->MS, 0, sizeof(cplx)*(size_t)nb*8*S);
  memset(c->LS, 0, sizeof(cplx)*(size_t)nb*8*S);
  memset(y, 0, sizeof(double)*3*(size_t)n);
  T[1] = omp_get_wtime();
  double *yt = malloc(sizeof(double)*3*(size_t)n);
  #pragma omp parallel for schedule(static, 64) if (!faithful)
  for (int i = 0; i < n; ++i) {
    double r[3] = {0, 0, 0};
    for (int64_t k = c->row_ptr[i]; k < c->row_ptr[i+1]; ++k) {
      const double *A = c->val9 + 9*k, *xs = x + 3*(size_t)c->perm[c->col[k]];
      r[0] += A[0]*xs[0] + A[1]*xs[1] + A[2]*xs[2];       /* Mat3 * Vec3, include/Mat3.hpp:60-66 */
      r[1] += A[3]*xs[0] + A[4]*xs[1] + A[5]*xs[2];
      r[2] += A[6]*xs[0] + A[7]*xs[1] + A[8]*xs[2];
    }
    yt[3*i] = r[0]; yt[3*i+1] = r[1]; yt[3*i+2] = r[2];
  }
  for (int i = 0; i < n; ++i) for (int k = 0; k < 3; ++k) y[3*(size_t)c->perm[i] + k] += yt[3*i + k];
  T[2] = omp_get_wtime();
  #pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < c->n_p2m; ++i) {
    const orc_box *b = &c->boxes[c->p2m[i]];
    cplx *M = c->MS + (size_t)c->p2m[i]*8*S;
    for (uint32_t j = b->bb; j < b->be; ++j)
      stokes_p2m_panel(t, c, &c->panels[c->perm[j]], x + 3*(size_t)c->perm[j], b->center, M);
  }
  T[3] = omp_get_wtime();
  for (int i = 0; i < c->n_m2m; ++i) {
    const int ch = c->m2m[i].first, pa = c->m2m[i].second;
    double tr[3]; for (int k = 0; k < 3; ++k) tr[k] = c->boxes[pa].center[k] - c->boxes[ch].center[k];
    for (int e = 0; e < E; ++e) orc_m2m(t, c->MS + ((size_t)ch*8 + e)*S, c->MS + ((size_t)pa*8 + e)*S, tr);
  }
  T[4] = omp_get_wtime();
  #pragma omp parallel for schedule(dynamic, 8)
  for (int b = 0; b < nb; ++b)
    for (int i = c->lr_ptr[b]; i < c->lr_ptr[b+1]; ++i) {
      const int s = c->lr_src[i];
      double tr[3]; for (int k = 0; k < 3; ++k) tr[k] = c->boxes[b].center[k] - c->boxes[s].center[k];
      for (int e = 0; e < E; ++e) orc_m2l(t, c->MS + ((size_t)s*8 + e)*S, c->LS + ((size_t)b*8 + e)*S, tr);
    }
  T[5] = omp_get_wtime();
  for (int i = 0; i < c->n_l2l; ++i) {
    const int pa = c->l2l[i].first, ch = c->l2l[i].second;
    double tr[3]; for (int k = 0; k < 3; ++k) tr[k] = c->boxes[ch].center[k] - c->boxes[pa].center[k];
    for (int e = 0; e < E; ++e) orc_l2l(t, c->LS + ((size_t)pa*8 + e)*S, c->LS + ((size_t)ch*8 + e)*S, tr);
  }
  T[6] = omp_get_wtime();
  #pragma omp parallel for schedule(dynamic, 4)
  for (int i = 0; i < c->n_l2p; ++i) {
    const orc_box *b = &c->boxes[c->l2p[i]];
    const cplx *L = c->LS + (size_t)c->l2p[i]*8*S;
    for (uint32_t j = b->bb; j < b->be; ++j)
      stokes_l2p_panel(t, c, L, b->center, &c->panels[c->perm[j]], y + 3*(size_t)c->perm[j]);
  }
  T[7] = omp_get_wtime();
  if (stage_s) { for (int i = 0; i < 7; ++i) stage_s[i] = T[i+1] - T[i]; stage_s[7] = T[7] - T[0]; }
  free(yt);
  orc_tables_destroy(t);
  return 0;
}

/* include/Direct.hpp:99-125 with the Mat3 kernel value: r_i += K(t_i, s_j) * c_j */
void orc_stokes_direct(const orc_ctx *c, const double *x, double *y, int row_begin, int row_end) {
  #pragma omp parallel for schedule(dynamic, 4)
  for (int i = row_begin; i < row_end; ++i) {
    double r[3] = {0, 0, 0}, A[9];
    for (int j = 0; j < c->n; ++j) {
      orc_stokes_entry(c, &c->panels[i], &c->panels[j], A);
      const double *xs = x + 3*(size_t)j;
      r[0] += A[0]*xs[0] + A[1]*xs[1] + A[2]*xs[2];
      r[1] += A[3]*xs[0] + A[4]*xs[1] + A[5]*xs[2];
      r[2] += A[6]*xs[0] + A[7]*xs[1] + A[8]*xs[2];
    }
    y[3*i] = r[0]; y[3*i+1] = r[1]; y[3*i+2] = r[2];
  }
}

/* the same on a list of target panels (original order): y_out[3k..3k+2] = rows of panel rows[k] */
void orc_stokes_direct_rows(const orc_ctx *c, const double *x, double *y_out, int nrows, const int32_t *rows) {
  #pragma omp parallel for schedule(dynamic, 4)
  for (int k = 0; k < nrows; ++k) {
    const int i = rows[k];
    double r[3] = {0, 0, 0}, A[9];
    for (int j = 0; j < c->n; ++j) {
      orc_stokes_entry(c, &c->panels[i], &c->panels[j], A);
      const double *xs = x + 3*(size_t)j;
      r[0] += A[0]*xs[0] + A[1]*xs[1] + A[2]*xs[2];
      r[1] += A[3]*xs[0] + A[4]*xs[1] + A[5]*xs[2];
      r[2] += A[6]*xs[0] + A[7]*xs[1] + A[8]*xs[2];
    }
    y_out[3*k] = r[0]; y_out[3*k+1] = r[1]; y_out[3*k+2] = r[2];
  }
}

void orc_stokes_get_near(const orc_ctx *c, double *val9) { memcpy(val9, c->val9, sizeof(double)*9*(size_t)c->nnz); }
void orc_stokes_get_expansions(const orc_ctx *c, int P, int which, double *out) {
  memcpy(out, which == 0 ? c->MS : c->LS, sizeof(cplx)*(size_t)c->nboxes*8*(size_t)(P*(P+1)/2));
}
void orc_stokes_entries(const orc_ctx *c, int npairs, const int32_t *ti, const int32_t *sj, double *out) {
  for (int k = 0; k < npairs; ++k) orc_stokes_entry(c, &c->panels[ti[k]], &c->panels[sj[k]], out + 9*k);
}

/* ---- single operators on caller-supplied panels (checkers of the product's fmmbem_ops_p2m / fmmbem_ops_l2p) ----
 * kernel == 0: LaplaceSphericalBEM::P2M / L2P (kernel/LaplaceSphericalBEM.hpp:307-352, 448-476), expansions [2][S];
 * kernel == 1: StokesSphericalBEM::P2M (VELOCITY branch, :391-432) / L2P (:512-522), expansions [4][S].
 * M and result are added to, as the reference's operators do.  Returns 0, or -1 for a bad P / K. */
int orc_single_p2m(int kernel, int P, int K, double mu, int n, const double *verts, const uint8_t *bc, const double *charges,
                   const double center[3], cplx *M) {
  (void)mu;
  orc_tables *t = orc_tables_create(P);
  orc_ctx *c = calloc(1, sizeof *c);
  double pts[ORC_MAXK][3], qstore[3*ORC_MAXK];
  if (!t || !c) { free(c); if (t) orc_tables_destroy(t); return -1; }
  c->nq = orc_quadrature(K, pts, c->qw);
  if (c->nq < 0) { free(c); orc_tables_destroy(t); return -1; }
  const int S = P*(P+1)/2;
  for (int i = 0; i < n; ++i) {
    orc_panel p;
    orc_panel_init(&p, verts + 9*(size_t)i, bc ? bc[i] : 0, c->nq, (const double (*)[3])pts, qstore);
    if (kernel == 0) orc_p2m_panel(t, &p, c->nq, c->qw, charges[i], center, M, M + S);
    else stokes_p2m_panel(t, c, &p, charges + 3*(size_t)i, center, M);
  }
  free(c); orc_tables_destroy(t);
  return 0;
}

int orc_single_l2p(int kernel, int P, int K, double mu, const cplx *L, const double center[3], int n, const double *verts,
                   const uint8_t *bc, double *result) {
  orc_tables *t = orc_tables_create(P);
  orc_ctx *c = calloc(1, sizeof *c);
  double pts[ORC_MAXK][3], qstore[3*ORC_MAXK];
  if (!t || !c) { free(c); if (t) orc_tables_destroy(t); return -1; }
  c->nq = orc_quadrature(K, pts, c->qw);
  if (c->nq < 0) { free(c); orc_tables_destroy(t); return -1; }
  c->mu = mu;
  const int S = P*(P+1)/2;
  for (int i = 0; i < n; ++i) {
    orc_panel p;
    orc_panel_init(&p, verts + 9*(size_t)i, bc ? bc[i] : 0, c->nq, (const double (*)[3])pts, qstore);
    if (kernel == 0) orc_l2p_panel(t, L, L + S, center, &p, result + i);
    else stokes_l2p_panel(t, c, L, center, &p, result + 3*(size_t)i);
  }
  free(c); orc_tables_destroy(t);
  return 0;
}
