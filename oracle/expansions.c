/* expansions.c -- CPU ORACLE (test infrastructure): spherical-harmonic Laplace expansions,
 * restated from kernel/LaplaceSpherical.hpp and the BEM P2M/L2P of kernel/LaplaceSphericalBEM.hpp.
 * Loop structure, index formulae and the EPS scaling follow the reference line by line so that the
 * rounding behaviour is the reference's; see fmm_oracle.h for the rules. */
#include "fmm_oracle.h"
#include <math.h>
#include <stdlib.h>

static const double EPS = ORC_EPS;
static inline int ODDEVEN(int n) { return ((n & 1) == 1) ? -1 : 1; }     /* LaplaceSpherical.hpp:34-36 */
#define CI (_Complex_I)

/* kernel/LaplaceSpherical.hpp:87-117 (precompute) */
orc_tables *orc_tables_create(int P) {
  if (P < 1 || P > ORC_PMAX) return NULL;
  orc_tables *t = malloc(sizeof *t);
  t->P = P;
  t->prefactor = malloc(sizeof(double)*4*P*P);
  t->Anm = malloc(sizeof(double)*4*P*P);
  t->Cnm = malloc(sizeof(cplx)*(size_t)P*P*P*P);
  for (int n = 0; n != 2*P; ++n) {
    for (int m = -n; m <= n; ++m) {
      int nm = n*n + n + m;
      int nabsm = abs(m);
      double fnmm = EPS; for (int i = 1; i <= n-m; ++i) fnmm *= i;
      double fnpm = EPS; for (int i = 1; i <= n+m; ++i) fnpm *= i;
      double fnma = 1.0; for (int i = 1; i <= n-nabsm; ++i) fnma *= i;
      double fnpa = 1.0; for (int i = 1; i <= n+nabsm; ++i) fnpa *= i;
      t->prefactor[nm] = sqrt(fnma/fnpa);
      t->Anm[nm] = ODDEVEN(n)/sqrt(fnmm*fnpm);
    }
  }
  for (int j = 0, jk = 0, jknm = 0; j != P; ++j) {
    for (int k = -j; k <= j; ++k, ++jk) {
      for (int n = 0, nm = 0; n != P; ++n) {
        for (int m = -n; m <= n; ++m, ++nm, ++jknm) {
          const int jnkm = (j+n)*(j+n) + j + n + m - k;
          t->Cnm[jknm] = cpow(CI, (double)(abs(k-m) - abs(k) - abs(m)))                      /* :111 */
                         * (double)(ODDEVEN(j)*t->Anm[nm]*t->Anm[jk]/t->Anm[jnkm]) * EPS;    /* :112 */
        }
      }
    }
  }
  return t;
}
void orc_tables_destroy(orc_tables *t) {
  if (!t) return;
  free(t->prefactor); free(t->Anm); free(t->Cnm); free(t);
}

/* kernel/LaplaceSpherical.hpp:528-541 (cart2sph) */
void orc_cart2sph(double *r, double *theta, double *phi, const double d[3]) {
  *r = sqrt(d[0]*d[0] + d[1]*d[1] + d[2]*d[2]) + EPS;
  *theta = acos(d[2] / *r);
  if (fabs(d[0]) + fabs(d[1]) < EPS) *phi = 0;
  else if (fabs(d[0]) < EPS) *phi = d[1] / fabs(d[1]) * M_PI * 0.5;
  else if (d[0] > 0) *phi = atan(d[1] / d[0]);
  else *phi = atan(d[1] / d[0]) + M_PI;
}

/* kernel/LaplaceSpherical.hpp:455-488 (evalMultipole): rho^n Y_n^m and theta derivative */
void orc_eval_multipole(const orc_tables *t, double rho, double alpha, double beta, cplx *Ynm, cplx *YnmTheta) {
  const int P = t->P; const double *prefactor = t->prefactor;
  double x = cos(alpha), y = sin(alpha);
  double fact = 1, pn = 1, rhom = 1;
  for (int m = 0; m != P; ++m) {
    cplx eim = cexp(CI * (double)(m * beta));
    double p = pn;
    int npn = m*m + 2*m, nmn = m*m;
    Ynm[npn] = rhom * p * prefactor[npn] * eim;
    Ynm[nmn] = conj(Ynm[npn]);
    double p1 = p;
    p = x * (2*m + 1) * p1;
    YnmTheta[npn] = rhom * (p - (m + 1) * x * p1) / y * prefactor[npn] * eim;
    rhom *= rho;
    double rhon = rhom;
    for (int n = m+1; n != P; ++n) {
      int npm = n*n + n + m, nmm = n*n + n - m;
      Ynm[npm] = rhon * p * prefactor[npm] * eim;
      Ynm[nmm] = conj(Ynm[npm]);
      double p2 = p1;
      p1 = p;
      p = (x * (2*n + 1) * p1 - (n + m) * p2) / (n - m + 1);
      YnmTheta[npm] = rhon * ((n - m + 1) * p - (n + 1) * x * p1) / y * prefactor[npm] * eim;
      rhon *= rho;
    }
    pn = -pn * fact * y;
    fact += 2;
  }
}

/* kernel/LaplaceSpherical.hpp:491-524 (evalLocal): rho^{-n-1} Y_n^m up to order 2P */
void orc_eval_local(const orc_tables *t, double rho, double alpha, double beta, cplx *Ynm, cplx *YnmTheta) {
  const int P = t->P; const double *prefactor = t->prefactor;
  double x = cos(alpha), y = sin(alpha);
  double fact = 1, pn = 1, rhom = 1.0 / rho;
  for (int m = 0; m != 2*P; ++m) {
    cplx eim = cexp(CI * (double)(m * beta));
    double p = pn;
    int npn = m*m + 2*m, nmn = m*m;
    Ynm[npn] = rhom * p * prefactor[npn] * eim;
    Ynm[nmn] = conj(Ynm[npn]);
    double p1 = p;
    p = x * (2*m + 1) * p1;
    YnmTheta[npn] = rhom * (p - (m + 1) * x * p1) / y * prefactor[npn] * eim;
    rhom /= rho;
    double rhon = rhom;
    for (int n = m+1; n != 2*P; ++n) {
      int npm = n*n + n + m, nmm = n*n + n - m;
      Ynm[npm] = rhon * p * prefactor[npm] * eim;
      Ynm[nmm] = conj(Ynm[npm]);
      double p2 = p1;
      p1 = p;
      p = (x * (2*n + 1) * p1 - (n + m) * p2) / (n - m + 1);
      YnmTheta[npm] = rhon * ((n - m + 1) * p - (n + 1) * x * p1) / y * prefactor[npm] * eim;
      rhon /= rho;
    }
    pn = -pn * fact * y;
    fact += 2;
  }
}

/* kernel/LaplaceSphericalBEM.hpp:307-352 (P2M for one source panel, all its quadrature points).
 * M0 receives the G moments (source BC POTENTIAL), M1 the dG/dn moments (NORMAL_DERIV). */
void orc_p2m_panel(const orc_tables *t, const orc_panel *src, int nq, const double *qw,
                   double charge, const double center[3], cplx *M0, cplx *M1) {
  const int P = t->P;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  for (int i = 0; i < nq; ++i) {
    double dist[3] = { src->q[i][0]-center[0], src->q[i][1]-center[1], src->q[i][2]-center[2] };
    double rho, alpha, beta;
    orc_cart2sph(&rho, &alpha, &beta, dist);
    orc_eval_multipole(t, rho, alpha, -beta, Ynm, YnmTheta);               /* :318 */
    for (int n = 0; n != P; ++n) {
      for (int m = 0; m <= n; ++m) {
        const int nm = n*n + n + m, nms = n*(n+1)/2 + m;
        if (src->bc == ORC_POTENTIAL) {
          M0[nms] += charge * qw[i] * src->area * Ynm[nm];                 /* :326 */
        } else {
          cplx brh = (double)n/rho*Ynm[nm];                                /* :331-333 */
          cplx bal = YnmTheta[nm];
          cplx bbe = -(0. + 1.*CI)*(double)m*Ynm[nm];
          cplx bxd = sin(alpha)*cos(beta)*brh + cos(alpha)*cos(beta)/rho*bal - sin(beta)/rho/sin(alpha)*bbe;
          cplx byd = sin(alpha)*sin(beta)*brh + cos(alpha)*sin(beta)/rho*bal + cos(beta)/rho/sin(alpha)*bbe;
          cplx bzd = cos(alpha)*brh - sin(alpha)/rho*bal;
          cplx mult_term = charge * qw[i] * src->area;                     /* :340-343 */
          M1[nms] += mult_term * src->n[0] * bxd;
          M1[nms] += mult_term * src->n[1] * byd;
          M1[nms] += mult_term * src->n[2] * bzd;
        }
      }
    }
  }
}

/* kernel/LaplaceSpherical.hpp:245-285 (M2M); translation = c_parent - c_child (executor/M2M.hpp:40) */
void orc_m2m(const orc_tables *t, const cplx *Ms, cplx *Mt, const double tr[3]) {
  const int P = t->P; const double *Anm = t->Anm;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  double rho, alpha, beta;
  orc_cart2sph(&rho, &alpha, &beta, tr);
  orc_eval_multipole(t, rho, alpha, -beta, Ynm, YnmTheta);
  for (int j = 0; j != P; ++j) {
    for (int k = 0; k <= j; ++k) {
      const int jk = j*j + j + k, jks = j*(j+1)/2 + k;
      cplx M = 0;
      for (int n = 0; n <= j; ++n) {
        int mmax = (k-1 < n) ? k-1 : n;
        for (int m = -n; m <= mmax; ++m) {
          if (j-n >= k-m) {
            const int jnkm = (j-n)*(j-n) + j - n + k - m;
            const int jnkms = (j-n)*(j-n+1)/2 + k - m;
            const int nm = n*n + n + m;
            M += Ms[jnkms] * cpow(CI, (double)(m - abs(m))) * Ynm[nm]
                 * (double)(ODDEVEN(n) * Anm[nm] * Anm[jnkm] / Anm[jk]);
          }
        }
        for (int m = k; m <= n; ++m) {
          if (j-n >= m-k) {
            const int jnkm = (j-n)*(j-n) + j - n + k - m;
            const int jnkms = (j-n)*(j-n+1)/2 - k + m;
            const int nm = n*n + n + m;
            M += conj(Ms[jnkms]) * Ynm[nm]
                 * (double)(ODDEVEN(k+n+m) * Anm[nm] * Anm[jnkm] / Anm[jk]);
          }
        }
      }
      Mt[jks] += M * EPS;
    }
  }
}

/* kernel/LaplaceSpherical.hpp:296-329 (M2L); translation = c_target - c_source (executor/M2L.hpp:40) */
void orc_m2l(const orc_tables *t, const cplx *Ms, cplx *Lt, const double tr[3]) {
  const int P = t->P; const cplx *Cnm = t->Cnm;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  double rho, alpha, beta;
  orc_cart2sph(&rho, &alpha, &beta, tr);
  orc_eval_local(t, rho, alpha, beta, Ynm, YnmTheta);
  for (int j = 0; j != P; ++j) {
    for (int k = 0; k <= j; ++k) {
      const int jk = j*j + j + k, jks = j*(j+1)/2 + k;
      cplx L = 0;
      for (int n = 0; n != P; ++n) {
        for (int m = -n; m < 0; ++m) {
          const int nm = n*n + n + m, nms = n*(n+1)/2 - m;
          const int jknm = jk*P*P + nm;
          const int jnkm = (j+n)*(j+n) + j + n + m - k;
          L += conj(Ms[nms]) * Cnm[jknm] * Ynm[jnkm];
        }
        for (int m = 0; m <= n; ++m) {
          const int nm = n*n + n + m, nms = n*(n+1)/2 + m;
          const int jknm = jk*P*P + nm;
          const int jnkm = (j+n)*(j+n) + j + n + m - k;
          L += Ms[nms] * Cnm[jknm] * Ynm[jnkm];
        }
      }
      Lt[jks] += L;
    }
  }
}

/* kernel/LaplaceSpherical.hpp:378-411 (L2L); translation = c_child - c_parent (executor/L2L.hpp:40) */
void orc_l2l(const orc_tables *t, const cplx *Ls, cplx *Lt, const double tr[3]) {
  const int P = t->P; const double *Anm = t->Anm;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  double rho, alpha, beta;
  orc_cart2sph(&rho, &alpha, &beta, tr);
  orc_eval_multipole(t, rho, alpha, beta, Ynm, YnmTheta);
  for (int j = 0; j != P; ++j) {
    for (int k = 0; k <= j; ++k) {
      const int jk = j*j + j + k, jks = j*(j+1)/2 + k;
      cplx L = 0;
      for (int n = j; n != P; ++n) {
        for (int m = j+k-n; m < 0; ++m) {
          const int jnkm = (n-j)*(n-j) + n - j + m - k;
          const int nm = n*n + n - m;
          const int nms = n*(n+1)/2 - m;
          L += conj(Ls[nms]) * Ynm[jnkm]
               * (double)(ODDEVEN(k) * Anm[jnkm] * Anm[jk] / Anm[nm]);
        }
        for (int m = 0; m <= n; ++m) {
          if (n-j >= abs(m-k)) {
            const int jnkm = (n-j)*(n-j) + n - j + m - k;
            const int nm = n*n + n + m;
            const int nms = n*(n+1)/2 + m;
            L += Ls[nms] * cpow(CI, (double)(m - k - abs(m-k)))
                 * Ynm[jnkm] * Anm[jnkm] * Anm[jk] / Anm[nm];
          }
        }
      }
      Lt[jks] += L * EPS;
    }
  }
}

/* kernel/LaplaceSphericalBEM.hpp:448-476 (L2P for one target panel): result += r0 (POTENTIAL) or -= r1 */
void orc_l2p_panel(const orc_tables *t, const cplx *L0, const cplx *L1, const double center[3],
                   const orc_panel *tgt, double *result) {
  const int P = t->P;
  cplx Ynm[4*ORC_PMAX*ORC_PMAX], YnmTheta[4*ORC_PMAX*ORC_PMAX];
  double r0 = 0., r1 = 0.;
  double dist[3] = { tgt->c[0]-center[0], tgt->c[1]-center[1], tgt->c[2]-center[2] };
  double r, theta, phi;
  orc_cart2sph(&r, &theta, &phi, dist);
  orc_eval_multipole(t, r, theta, phi, Ynm, YnmTheta);
  for (int n = 0; n != P; ++n) {
    int nm = n*n + n, nms = n*(n+1)/2;
    r0 += creal(L0[nms] * Ynm[nm]);
    r1 += creal(L1[nms] * Ynm[nm]);
    for (int m = 1; m <= n; ++m) {
      nm = n*n + n + m; nms = n*(n+1)/2 + m;
      r0 += 2 * creal(L0[nms] * Ynm[nm]);
      r1 += 2 * creal(L1[nms] * Ynm[nm]);
    }
  }
  if (tgt->bc == ORC_POTENTIAL) *result += r0; else *result -= r1;
}
