// capi_geometry_host.hip -- the in-tree remainder of the epipolar driver's sparse half: from the fundamental
// matrix to what the dense half and calc_cost_sgm need -- epipolar_geometry.m:40-96.  3x3 HOST algebra, no GPU work.
// (Feature detection / matching and the LMedS estimate of F, :130-149, are MATLAB toolbox calls outside the reference tree and
// randomised: not built.)  UNPINNED like every MATLAB-side row: MATLAB's svd is LAPACK's; this is a Jacobi eigen-solver on
// A'A.  The results the reference uses are invariant to the sign and rotation freedoms of an SVD BY CONSTRUCTION:
//   * the epipole is the null direction of F' divided by its third component (:41-43);
//   * {R1, R2} = {U W V', U W' V'} as a SET does not depend on the signs of the singular vectors (flipping a pair (u_i, v_i),
//     or u_3 alone when the matrix has rank 2, swaps R1 and R2 -- with the det < 0 negation of :52-55 applied), nor on a
//     rotation inside the plane of two equal singular values (W commutes with it);
//   * the choice between them is by the signs of the diagonal (:58-62): the one that is close to the identity.  When both or
//     neither qualify the reference's answer depends on its SVD's signs: `ambiguous` is set and R1 of THIS decomposition is used.
#include "capi_common.h"
#include <math.h>
#include <string.h>

using namespace fsgm;

namespace {

struct M3 { double a[3][3]; };

M3 mul(const M3& x, const M3& y) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[i][0] * y.a[0][j] + x.a[i][1] * y.a[1][j] + x.a[i][2] * y.a[2][j];
    return r;
}
M3 transpose(const M3& x) {
    M3 r;
    for (int i = 0; i < 3; i++)
        for (int j = 0; j < 3; j++) r.a[i][j] = x.a[j][i];
    return r;
}
double det(const M3& m) {
    return m.a[0][0] * (m.a[1][1] * m.a[2][2] - m.a[1][2] * m.a[2][1]) - m.a[0][1] * (m.a[1][0] * m.a[2][2] - m.a[1][2] * m.a[2][0]) +
           m.a[0][2] * (m.a[1][0] * m.a[2][1] - m.a[1][1] * m.a[2][0]);
}
bool inverse(const M3& m, M3& inv) {
    const double d = det(m);
    if (!(fabs(d) > 0.0) || !isfinite(d)) return false;
    const double (*a)[3] = m.a;
    inv.a[0][0] = (a[1][1] * a[2][2] - a[1][2] * a[2][1]) / d; inv.a[0][1] = (a[0][2] * a[2][1] - a[0][1] * a[2][2]) / d; inv.a[0][2] = (a[0][1] * a[1][2] - a[0][2] * a[1][1]) / d;
    inv.a[1][0] = (a[1][2] * a[2][0] - a[1][0] * a[2][2]) / d; inv.a[1][1] = (a[0][0] * a[2][2] - a[0][2] * a[2][0]) / d; inv.a[1][2] = (a[0][2] * a[1][0] - a[0][0] * a[1][2]) / d;
    inv.a[2][0] = (a[1][0] * a[2][1] - a[1][1] * a[2][0]) / d; inv.a[2][1] = (a[0][1] * a[2][0] - a[0][0] * a[2][1]) / d; inv.a[2][2] = (a[0][0] * a[1][1] - a[0][1] * a[1][0]) / d;
    return true;
}

// eigen-decomposition of a symmetric 3x3 by cyclic Jacobi rotations: s = V diag(w) V', eigenvalues sorted descending
void jacobi_eig(M3 s, double w[3], M3& V) {
    memset(&V, 0, sizeof V);
    V.a[0][0] = V.a[1][1] = V.a[2][2] = 1.0;
    for (int sweep = 0; sweep < 64; sweep++) {
        const double off = fabs(s.a[0][1]) + fabs(s.a[0][2]) + fabs(s.a[1][2]);
        const double dia = fabs(s.a[0][0]) + fabs(s.a[1][1]) + fabs(s.a[2][2]);
        if (off <= 1e-300 || off <= 1e-17 * dia) break;
        for (int p = 0; p < 2; p++)
            for (int q = p + 1; q < 3; q++) {
                if (s.a[p][q] == 0.0) continue;
                const double theta = (s.a[q][q] - s.a[p][p]) / (2.0 * s.a[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
                for (int k = 0; k < 3; k++) {                        // S <- J' S J, columns then rows
                    const double skp = s.a[k][p], skq = s.a[k][q];
                    s.a[k][p] = c * skp - sn * skq; s.a[k][q] = sn * skp + c * skq;
                }
                for (int k = 0; k < 3; k++) {
                    const double spk = s.a[p][k], sqk = s.a[q][k];
                    s.a[p][k] = c * spk - sn * sqk; s.a[q][k] = sn * spk + c * sqk;
                }
                for (int k = 0; k < 3; k++) {
                    const double vkp = V.a[k][p], vkq = V.a[k][q];
                    V.a[k][p] = c * vkp - sn * vkq; V.a[k][q] = sn * vkp + c * vkq;
                }
            }
    }
    for (int i = 0; i < 3; i++) w[i] = s.a[i][i];
    for (int i = 0; i < 2; i++)                                     // sort descending, columns of V along
        for (int j = i + 1; j < 3; j++)
            if (w[j] > w[i]) {
                const double t = w[i]; w[i] = w[j]; w[j] = t;
                for (int k = 0; k < 3; k++) { const double u = V.a[k][i]; V.a[k][i] = V.a[k][j]; V.a[k][j] = u; }
            }
}

// A = U diag(sv) V' with V from the eigenvectors of A'A; the columns of U that belong to (numerically) zero singular values
// are completed to an orthonormal basis
void svd3(const M3& A, M3& U, double sv[3], M3& V) {
    double w[3];
    jacobi_eig(mul(transpose(A), A), w, V);
    for (int i = 0; i < 3; i++) sv[i] = w[i] > 0 ? sqrt(w[i]) : 0.0;
    double u[3][3];                                                  // u[i] = column i
    int ok[3];
    for (int i = 0; i < 3; i++) {
        for (int r = 0; r < 3; r++) u[i][r] = A.a[r][0] * V.a[0][i] + A.a[r][1] * V.a[1][i] + A.a[r][2] * V.a[2][i];
        const double n = sqrt(u[i][0] * u[i][0] + u[i][1] * u[i][1] + u[i][2] * u[i][2]);
        ok[i] = n > 1e-12 * (sv[0] > 0 ? sv[0] : 1.0);
        if (ok[i]) for (int r = 0; r < 3; r++) u[i][r] /= n;
    }
    if (!ok[2] && ok[0] && ok[1]) {                                  // rank 2: u3 = u1 x u2
        u[2][0] = u[0][1] * u[1][2] - u[0][2] * u[1][1]; u[2][1] = u[0][2] * u[1][0] - u[0][0] * u[1][2]; u[2][2] = u[0][0] * u[1][1] - u[0][1] * u[1][0];
    }
    for (int i = 0; i < 3; i++)
        for (int r = 0; r < 3; r++) U.a[r][i] = u[i][r];
}

}  // namespace

extern "C" fsgm_status fsgm_epipolar_from_F(const double* F9, const double* K9, int32_t n_points, const double* pts1, const double* pts2,
                                           const uint8_t* inliers, fsgm_epi_geometry* out, int32_t* ambiguous) {
    FSGM_REQUIRE(F9 && K9 && out, "fsgm_epipolar_from_F: null argument");
    FSGM_REQUIRE(n_points >= 0 && (n_points == 0 || (pts1 && pts2)), "fsgm_epipolar_from_F: bad point list");
    M3 F, K, Kinv;
    memcpy(F.a, F9, sizeof F.a);
    memcpy(K.a, K9, sizeof K.a);
    for (int i = 0; i < 9; i++) FSGM_REQUIRE(isfinite(F9[i]) && isfinite(K9[i]), "fsgm_epipolar_from_F: F and K must be finite");
    FSGM_REQUIRE(inverse(K, Kinv), "fsgm_epipolar_from_F: K is singular");
    // epipole in image 2: F' e' = 0 (:38-43): the eigenvector of F F' for its smallest eigenvalue, divided by its third component
    double w[3];
    M3 V;
    jacobi_eig(mul(F, transpose(F)), w, V);                          // (F')'(F') = F F'
    const double e3 = V.a[2][2];
    FSGM_REQUIRE(fabs(e3) > 1e-300, "fsgm_epipolar_from_F: the epipole is at infinity (third component 0)");
    const double ex = V.a[0][2] / e3, ey = V.a[1][2] / e3;
    // E = K' F K, R1 = U W V', R2 = U W' V' (:46-55)
    const M3 E = mul(mul(transpose(K), F), K);
    M3 U, Ve;
    double sv[3];
    svd3(E, U, sv, Ve);
    FSGM_REQUIRE(sv[1] > 1e-12 * sv[0] && sv[0] > 0, "fsgm_epipolar_from_F: K'FK has rank < 2");
    const M3 W = {{{0, -1, 0}, {1, 0, 0}, {0, 0, 1}}};
    M3 R1 = mul(mul(U, W), transpose(Ve)), R2 = mul(mul(U, transpose(W)), transpose(Ve));
    if (det(R1) < 0)
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) { R1.a[i][j] = -R1.a[i][j]; R2.a[i][j] = -R2.a[i][j]; }
    const bool q1 = R1.a[0][0] > 0 && R1.a[1][1] > 0 && R1.a[2][2] > 0, q2 = R2.a[0][0] > 0 && R2.a[1][1] > 0 && R2.a[2][2] > 0;
    const M3 R = q1 ? R1 : R2;                                       // :58-62
    if (ambiguous) *ambiguous = (q1 == q2) ? 1 : 0;
    const M3 Hm = mul(mul(K, R), Kinv);                              // :65  H = K*R/K
    // expansion vote over the inlier matches (:68-96)
    long long expansion = 0, n_in = 0;
    for (int i = 0; i < n_points; i++) {
        if (inliers && !inliers[i]) continue;
        n_in++;
        const double x1 = pts1[2 * i], y1 = pts1[2 * i + 1], x2 = pts2[2 * i], y2 = pts2[2 * i + 1];
        const double p0 = Hm.a[0][0] * x1 + Hm.a[0][1] * y1 + Hm.a[0][2], p1 = Hm.a[1][0] * x1 + Hm.a[1][1] * y1 + Hm.a[1][2];
        const double p2 = Hm.a[2][0] * x1 + Hm.a[2][1] * y1 + Hm.a[2][2];
        const double xr = p0 / p2, yr = p1 / p2;
        const double d1 = sqrt((xr - ex) * (xr - ex) + (yr - ey) * (yr - ey)), d2 = sqrt((x2 - ex) * (x2 - ex) + (y2 - ey) * (y2 - ey));
        if (d2 > d1) expansion++;
    }
    memcpy(out->F, F9, sizeof out->F);
    memcpy(out->H, Hm.a, sizeof out->H);
    out->epipole[0] = ex; out->epipole[1] = ey;
    out->direction = (n_in > 0 && (double)expansion / (double)n_in > 0.5) ? 0 : 1;    // :92-96 (0/0 is NaN in MATLAB: not > 0.5 -> 1)
    return FSGM_OK;
}
