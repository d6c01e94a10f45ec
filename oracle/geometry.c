/* geometry.c -- CPU ORACLE (test infrastructure): mesh, panels, quadrature tables and
 * the near-field kernel entry of LaplaceSphericalBEM.  See fmm_oracle.h for the rules. */
#include "fmm_oracle.h"
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------
 * Triangle Gauss rules, barycentric points + weights.
 * Follows examples/BEM/GaussQuadrature.hpp:19-185 (tables are numeric data).
 * Quirks kept: key 7 aliases the 4-point rule (:58-59); key 17 holds 16 points (:102-115).
 * Key 79 (:186-273) is not restated (no configuration uses it) -> returns -1.
 * ---------------------------------------------------------------------------------- */
static int perm3(double pts[][3], double *w, int at, double a, double b, double c, double wt) {
  /* the 6 permutations in the order the reference lists them: (a,b,c),(a,c,b),(b,a,c),(b,c,a),(c,a,b),(c,b,a) */
  const double P[6][3] = {{a,b,c},{a,c,b},{b,a,c},{b,c,a},{c,a,b},{c,b,a}};
  for (int i = 0; i < 6; ++i) { memcpy(pts[at+i], P[i], sizeof(double)*3); w[at+i] = wt; }
  return at + 6;
}
static int rot3(double pts[][3], double *w, int at, double b1, double b2, double wt) {
  /* (b1,b2,b2),(b2,b1,b2),(b2,b2,b1) */
  const double P[3][3] = {{b1,b2,b2},{b2,b1,b2},{b2,b2,b1}};
  for (int i = 0; i < 3; ++i) { memcpy(pts[at+i], P[i], sizeof(double)*3); w[at+i] = wt; }
  return at + 3;
}
static int one(double pts[][3], double *w, int at, double wt) {
  pts[at][0] = pts[at][1] = pts[at][2] = 1./3; w[at] = wt; return at + 1;
}

int orc_quadrature(int key, double pts[][3], double *w) {
  int n = 0;
  switch (key) {
    case 1:  /* GaussQuadrature.hpp:19-20 */
      n = one(pts, w, n, 1.); break;
    case 3: { /* :22-25 */
      const double P[3][3] = {{0.5,0.5,0.},{0.,0.5,0.5},{0.5,0.,0.5}};
      for (int i = 0; i < 3; ++i) { memcpy(pts[i], P[i], sizeof(double)*3); w[i] = 1./3; }
      n = 3; break; }
    case 4: case 7: { /* :27-31, alias :58-59 */
      const double P[4][3] = {{1./3,1./3,1./3},{.6,.2,.2},{.2,.6,.2},{.2,.2,.6}};
      const double W[4] = {-27./48, 25./48, 25./48, 25./48};
      for (int i = 0; i < 4; ++i) { memcpy(pts[i], P[i], sizeof(double)*3); w[i] = W[i]; }
      n = 4; break; }
    case 13: /* :60-85 */
      n = one(pts, w, n, -0.149570044467682);
      n = rot3(pts, w, n, 0.479308067841920, 0.260345966079040, 0.175615257433208);
      n = rot3(pts, w, n, 0.869739794195568, 0.065130102902216, 0.053347235608838);
      n = perm3(pts, w, n, 0.048690315425316, 0.312865496004874, 0.638444188569810, 0.077113760890257);
      break;
    case 17: /* :86-116 (16 points) */
      n = one(pts, w, n, 0.144315607677787);
      n = rot3(pts, w, n, 0.081414823414554, 0.459292588292723, 0.095091634267285);
      n = rot3(pts, w, n, 0.658861384496480, 0.170569307751760, 0.103217370534718);
      n = rot3(pts, w, n, 0.898905543365938, 0.050547228317031, 0.032458497623198);
      n = perm3(pts, w, n, 0.008394777409958, 0.263112829634638, 0.728492392955404, 0.027230314174435);
      break;
    case 19: /* :117-150 */
      n = one(pts, w, n, 0.097135796282799);
      n = rot3(pts, w, n, 0.020634961602525, 0.489682519198738, 0.031334700227139);
      n = rot3(pts, w, n, 0.125820817014127, 0.437089591492937, 0.077827541004774);
      n = rot3(pts, w, n, 0.623592928761935, 0.188203535619033, 0.079647738927210);
      n = rot3(pts, w, n, 0.910540973211095, 0.044729513394453, 0.025577675658698);
      n = perm3(pts, w, n, 0.036838412054736, 0.221962989160766, 0.741198598784498, 0.043283539377289);
      break;
    case 25: /* :151-185 */
      n = one(pts, w, n, 0.090817990382754);
      n = rot3(pts, w, n, 0.028844733232685, 0.485577633383657, 0.036725957756467);
      n = rot3(pts, w, n, 0.781036849029926, 0.109481575485037, 0.045321059435528);
      n = perm3(pts, w, n, 0.141707219414880, 0.307939838764121, 0.550352941820999, 0.072757916845420);
      n = perm3(pts, w, n, 0.025003534762686, 0.246672560639903, 0.728323904597411, 0.028327242531057);
      n = perm3(pts, w, n, 0.009540815400299, 0.066803251012200, 0.923655933587500, 0.009421666963733);
      break;
    case 79: /* :186-272: centroid, ten rotation triples b..k, eight permutation sextets l..s, weights wa..ws */
      n = one(pts, w, n, 0.033057055541624);
      n = rot3(pts, w, n, -0.001900928704400, 0.500950464352200, 0.000867019185663);
      n = rot3(pts, w, n, 0.023574084130543, 0.488212957934729, 0.011660052716448);
      n = rot3(pts, w, n, 0.089726636099435, 0.455136681950283, 0.022876936356421);
      n = rot3(pts, w, n, 0.196007481363421, 0.401996259318289, 0.030448982673938);
      n = rot3(pts, w, n, 0.488214180481157, 0.255892909759421, 0.030624891725355);
      n = rot3(pts, w, n, 0.647023488009788, 0.176488255995106, 0.024368057676800);
      n = rot3(pts, w, n, 0.791658289326483, 0.104170855336758, 0.015997432032024);
      n = rot3(pts, w, n, 0.893862072318140, 0.053068963840930, 0.007698301815602);
      n = rot3(pts, w, n, 0.916762569607942, 0.041618715196029, -0.000632060497488);
      n = rot3(pts, w, n, 0.976836157186356, 0.011581921406822, 0.001751134301193);
      n = perm3(pts, w, n, 0.048741583664839, 0.344855770229001, 0.606402646106160, 0.016465839189576);
      n = perm3(pts, w, n, 0.006314115948605, 0.377843269594854, 0.615842614456541, 0.004839033540485);
      n = perm3(pts, w, n, 0.134316520547348, 0.306635479062357, 0.559048000390295, 0.025804906534650);
      n = perm3(pts, w, n, 0.013973893962392, 0.249419362774742, 0.736606743262866, 0.008471091054441);
      n = perm3(pts, w, n, 0.075549132909764, 0.212775724802802, 0.711675142287434, 0.018354914106280);
      n = perm3(pts, w, n, -0.008368153208227, 0.146965436053239, 0.861402717154987, 0.000704404677908);
      n = perm3(pts, w, n, 0.026686063258714, 0.137726978828923, 0.835586957912363, 0.010112684927462);
      n = perm3(pts, w, n, 0.010547719294141, 0.059696109149007, 0.929756171556853, 0.003573909385950);
      break;
    default: return -1;
  }
  return n;
}

static double norm3(const double a[3]) { return sqrt(a[0]*a[0] + a[1]*a[1] + a[2]*a[2]); }

/* kernel/LaplaceSphericalBEM.hpp:64-97 (Panel main constructor) */
void orc_panel_init(orc_panel *p, const double v[9], int bc, int nq, const double pts[][3], double *qstore) {
  memcpy(p->v, v, sizeof(double)*9);
  const double *p0 = p->v[0], *p1 = p->v[1], *p2 = p->v[2];
  for (int k = 0; k < 3; ++k) p->c[k] = (p0[k] + p1[k] + p2[k]) / 3;            /* :71 */
  double L0[3], L1[3];
  for (int k = 0; k < 3; ++k) { L0[k] = p2[k] - p0[k]; L1[k] = p1[k] - p0[k]; }  /* :74-75 */
  double c[3] = { L0[1]*L1[2] - L0[2]*L1[1],                                     /* :77-79 */
                  -(L0[0]*L1[2] - L0[2]*L1[0]),
                  L0[0]*L1[1] - L0[1]*L1[0] };
  p->area = 0.5 * norm3(c);                                                      /* :80 */
  for (int k = 0; k < 3; ++k) p->n[k] = c[k] / 2 / p->area;                      /* :81 */
  for (int i = 0; i < nq; ++i)                                                   /* :90-96 */
    for (int k = 0; k < 3; ++k)
      qstore[3*i+k] = p->v[0][k]*pts[i][0] + p->v[1][k]*pts[i][1] + p->v[2][k]*pts[i][2];
  p->q = (const double (*)[3])qstore;
  p->bc = bc;
}

/* examples/BEM/Triangulation.hpp:35-121 (UnitSphere): octahedron, recursions-1 four-way splits,
 * new vertices projected to the unit sphere.  verts_out: N x 9 doubles (v0,v1,v2). */
long orc_unit_sphere(int recursions, double *verts_out) {
  long n = 8;
  for (int i = 0; i < recursions - 1; ++i) n *= 4;
  if (!verts_out) return n;
  static const double ov[18] = {1,0,0, -1,0,0, 0,1,0, 0,-1,0, 0,0,1, 0,0,-1};       /* :59 */
  static const unsigned ot[24] = {0,4,2, 2,4,1, 1,4,3, 3,4,0, 0,2,5, 2,1,5, 1,3,5, 3,0,5}; /* :60 */
  double *cur = malloc(sizeof(double)*9*n), *nxt = malloc(sizeof(double)*9*n);
  for (int t = 0; t < 8; ++t)
    for (int j = 0; j < 3; ++j) memcpy(cur + 9*t + 3*j, ov + 3*ot[3*t+j], sizeof(double)*3);
  long m = 8;
  for (int it = 0; it < recursions - 1; ++it) {
    for (long t = 0; t < m; ++t) {                        /* triangle::split, :36-55 */
      const double *v0 = cur + 9*t, *v1 = v0 + 3, *v2 = v0 + 6;
      double a[3], b[3], c[3];
      for (int k = 0; k < 3; ++k) { a[k] = (v0[k]+v2[k])*0.5; b[k] = (v0[k]+v1[k])*0.5; c[k] = (v1[k]+v2[k])*0.5; }
      double na = norm3(a), nb = norm3(b), nc = norm3(c);
      for (int k = 0; k < 3; ++k) { a[k] /= na; b[k] /= nb; c[k] /= nc; }
      double *o = nxt + 36*t;
      const double *T[4][3] = {{v0,b,a},{b,v1,c},{a,b,c},{a,c,v2}};
      for (int s = 0; s < 4; ++s) for (int j = 0; j < 3; ++j) memcpy(o + 9*s + 3*j, T[s][j], sizeof(double)*3);
    }
    m *= 4; double *tmp = cur; cur = nxt; nxt = tmp;
  }
  memcpy(verts_out, cur, sizeof(double)*9*n);
  free(cur); free(nxt);
  return n;
}

/* ------------------------------------------------------------------------------------
 * Semi-analytical integral of G and dG/dn over a flat triangle.
 * Follows examples/BEM/SemiAnalytical.hpp: lineInt :13-71, intSide :81-145, SemiAnalytical :148-203
 * (LAPLACE branch only), with Mat3::multiply = row-major 3x3 times vector (include/Mat3.hpp:76-82).
 * ---------------------------------------------------------------------------------- */
static void line_int(double *G, double *dGdn, double z, double x, double v1, double v2) {
  double theta1 = atan2(v1, x), theta2 = atan2(v2, x);
  double dtheta = theta2 - theta1, thetam = (theta2 + theta1) / 2;
  double absZ = fabs(z), signZ;
  if (absZ < 1e-10) signZ = 0; else signZ = z / absZ;
  static const double xk[5] = { -9.06179846e-01, -5.38469310e-01, 1.78162900e-17, 9.06179846e-01, 5.38469310e-01 };
  static const double wk[5] = { 0.23692689, 0.47862867, 0.56888889, 0.23692689, 0.47862867 };
  for (int i = 0; i < 5; ++i) {
    double thetak = dtheta/2*xk[i] + thetam;
    double Rtheta = x / cos(thetak);
    double R = sqrt(Rtheta*Rtheta + z*z);
    *G    += wk[i]*(R - absZ) * dtheta/2;
    *dGdn += wk[i]*(z/R - signZ) * dtheta/2;
  }
}
static void mat3_mul(const double M[9], const double x[3], double r[3]) {
  r[0] = M[0]*x[0] + M[1]*x[1] + M[2]*x[2];
  r[1] = M[3]*x[0] + M[4]*x[1] + M[5]*x[2];
  r[2] = M[6]*x[0] + M[7]*x[1] + M[8]*x[2];
}
static void cross3(const double u[3], const double v[3], double r[3]) {
  r[0] = u[1]*v[2] - u[2]*v[1]; r[1] = u[2]*v[0] - u[0]*v[2]; r[2] = u[0]*v[1] - u[1]*v[0];
}
static void int_side(double *G, double *dGdn, const double v1[3], const double v2[3], double p) {
  double v21[3] = { v2[0]-v1[0], v2[1]-v1[1], v2[2]-v1[2] };
  double L21 = norm3(v21);
  double v21u[3] = { v21[0]/L21, v21[1]/L21, v21[2]/L21 };
  const double unit[3] = {0, 0, 1};
  double orthog[3]; cross3(unit, v21u, orthog);
  double R[9];
  for (int i = 0; i < 3; ++i) { R[i*3+0] = orthog[i]; R[i*3+1] = v21u[i]; R[i*3+2] = unit[i]; }   /* :100-108 */
  double v1new[3]; mat3_mul(R, v1, v1new);
  if (v1new[0] < 0) {                                                                             /* :112-119 */
    for (int i = 0; i < 9; ++i) R[i] = -R[i];
    R[8] = 1.;
    mat3_mul(R, v1, v1new);
  }
  double v2new[3]; mat3_mul(R, v2, v2new);
  double x = v1new[0];
  if ((v1new[1] > 0 && v2new[1] < 0) || (v1new[1] < 0 && v2new[1] > 0)) {                        /* :125-135 */
    double G1 = 0, dG1 = 0, G2 = 0, dG2 = 0;
    line_int(&G1, &dG1, p, x, 0, v1new[1]);
    line_int(&G2, &dG2, p, x, v2new[1], 0);
    *G += G1 + G2; *dGdn += dG1 + dG2;
  } else {                                                                                        /* :136-143 */
    double G1 = 0, dG1 = 0;
    line_int(&G1, &dG1, p, x, v1new[1], v2new[1]);
    *G -= G1; *dGdn -= dG1;
  }
}
void orc_semi_analytical(double *G, double *dGdn, const double y0[3], const double y1[3],
                         const double y2[3], const double x[3], int same) {
  double xp[3], y1p[3], y2p[3], y0p[3] = {0, 0, 0};
  for (int k = 0; k < 3; ++k) { xp[k] = x[k]-y0[k]; y1p[k] = y1[k]-y0[k]; y2p[k] = y2[k]-y0[k]; }  /* :154-157 */
  double X[3] = { y1p[0], y1p[1], y1p[2] }, Z[3], Y[3];
  cross3(y1p, y2p, Z);                                                                             /* :161 */
  double Xn = norm3(X), Zn = norm3(Z);
  for (int k = 0; k < 3; ++k) { X[k] /= Xn; Z[k] /= Zn; }
  cross3(Z, X, Y);                                                                                 /* :167 */
  double rot[9];
  for (int i = 0; i < 3; ++i) { rot[0*3+i] = X[i]; rot[1*3+i] = Y[i]; rot[2*3+i] = Z[i]; }         /* :171-175 */
  double p0[3], p1[3], p2[3], xpl[3];
  mat3_mul(rot, y0p, p0); mat3_mul(rot, y1p, p1); mat3_mul(rot, y2p, p2); mat3_mul(rot, xp, xpl);
  double f0[3], f1[3], f2[3];
  for (int k = 0; k < 3; ++k) { f0[k] = p0[k]-xpl[k]; f1[k] = p1[k]-xpl[k]; f2[k] = p2[k]-xpl[k]; } /* :183-185 */
  f0[2] = p0[2]; f1[2] = p1[2]; f2[2] = p2[2];                                                     /* :188-190 */
  int_side(G, dGdn, f0, f1, xpl[2]);                                                               /* :193-195 */
  int_side(G, dGdn, f1, f2, xpl[2]);
  int_side(G, dGdn, f2, f0, xpl[2]);
  if (same) *dGdn = 2*M_PI;                                                                        /* :197-201 */
}

/* kernel/LaplaceSphericalBEM.hpp:159-205 (eval_G): near -> semi-analytic, far -> K-point Gauss */
double orc_eval_G(const orc_panel *s, const double t[3], int nq, const double *w) {
  double d[3] = { t[0]-s->c[0], t[1]-s->c[1], t[2]-s->c[2] };
  double dist = norm3(d);
  if (sqrt(2*s->area)/dist >= 0.5) {                                    /* :163 */
    double G = 0., dGdn = 0.;
    orc_semi_analytical(&G, &dGdn, s->v[0], s->v[1], s->v[2], t, dist < 1e-10);   /* :173-175 */
    return G;
  }
  double r = 0.;
  for (int i = 0; i < nq; ++i) {                                        /* :198-203 */
    double e[3] = { t[0]-s->q[i][0], t[1]-s->q[i][1], t[2]-s->q[i][2] };
    r += w[i]*s->area/norm3(e);
  }
  return r;
}

/* the "K_fine" rule keyed 17 (16 points), BEMConfig::GaussPoints(17) (examples/BEM/BEMConfig.hpp:30-35) */
static double g17p[ORC_MAXK][3], g17w[ORC_MAXK]; static int g17n;
__attribute__((constructor)) static void init_g17(void) { g17n = orc_quadrature(17, g17p, g17w); }

/* kernel/LaplaceSphericalBEM.hpp:208-264 (eval_dGdn): self 2pi, near -> 16-pt rule (key 17), far -> K-pt */
double orc_eval_dGdn(const orc_panel *s, const double t[3], int nq, const double *w) {
  double d[3] = { t[0]-s->c[0], t[1]-s->c[1], t[2]-s->c[2] };
  double dist = norm3(d);
  if (dist < 1e-8) return 2*M_PI;                                       /* :211-213 */
  if (sqrt(2*s->area)/dist >= 0.5) {                                    /* :215, :229-249 */
    const double (*gp)[3] = g17p; const double *gw = g17w; const int gn = g17n;
    double r = 0.;
    for (int i = 0; i < gn; ++i) {
      double pt[3];
      for (int k = 0; k < 3; ++k) pt[k] = s->v[0][k]*gp[i][0] + s->v[1][k]*gp[i][1] + s->v[2][k]*gp[i][2];
      double dx[3] = { pt[0]-t[0], pt[1]-t[1], pt[2]-t[2] };
      double r2 = dx[0]*dx[0] + dx[1]*dx[1] + dx[2]*dx[2];
      double r3 = r2*sqrt(r2);
      r += gw[i]*s->area*(dx[0]*s->n[0] + dx[1]*s->n[1] + dx[2]*s->n[2])/r3;
    }
    return r;
  }
  double res = 0.;
  for (int i = 0; i < nq; ++i) {                                        /* :251-262 */
    double dx[3] = { s->q[i][0]-t[0], s->q[i][1]-t[1], s->q[i][2]-t[2] };
    double r2 = dx[0]*dx[0] + dx[1]*dx[1] + dx[2]*dx[2];
    double r3 = r2*sqrt(r2);
    res += w[i]*s->area*(dx[0]*s->n[0] + dx[1]*s->n[1] + dx[2]*s->n[2])/r3;
  }
  return res;
}

/* kernel/LaplaceSphericalBEM.hpp:273-297: K(t, s); the TARGET's BC picks G vs dG/dn */
double orc_kernel(const orc_ctx *c, const orc_panel *t, const orc_panel *s) {
  if (t->bc == ORC_POTENTIAL) return orc_eval_G(s, t->c, c->nq, c->qw);
  return orc_eval_dGdn(s, t->c, c->nq, c->qw);
}
