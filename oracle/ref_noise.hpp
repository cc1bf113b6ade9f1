/* ============================================================================
 * oracle/ref_noise.hpp  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 *
 * Small dense linear algebra in a chosen scalar type, used by the oracle's
 * "reference-noise" variants (cvo_oracle.h, ORC_VAR_*): what the reference
 * computes with Eigen in f32 where oracle and HIP path use exact double closed
 * forms.
 *
 *   - eigenvalues of the companion matrix of the step cubic (cvo.cpp:76-92:
 *     MatrixXf::eigenvalues(), root accepted only when imag()==0, cvo.cpp:326)
 *   - Matrix4f::log() (cvo.cpp:94-104, Eigen unsupported/MatrixFunctions)
 *
 * Eigen is a third-party dependency that is NOT under /root/reference (version
 * unpinned; README: ">= 3.1.0, tested 3.3.7").  Its published algorithms are
 * restated here: Householder Hessenberg reduction + Francis double-shift QR
 * (Golub & Van Loan, Matrix Computations, Alg. 7.4.2 / 7.5.1 / 7.5.2 -- the
 * EISPACK hqr2 scheme Eigen's RealSchur follows) and the inverse scaling and
 * squaring logarithm with Gauss-Legendre partial-fraction Pade approximants
 * (Higham, Functions of Matrices, Alg. 11.9; Eigen's MatrixLogarithm.h uses the
 * same node/weight form with degrees 3..5 in single precision).  These are NOT
 * bit-for-bit Eigen: they carry the same kind and size of rounding noise (all
 * arithmetic in the scalar type S, similarity transforms of a matrix with
 * entries ~1), which is what the noise envelope needs.  Instantiated in double
 * they are checked against numpy/scipy (tests/test_oracle_noise.py).
 * ========================================================================== */
#pragma once
#include <cmath>
#include <limits>

namespace refnoise {

template <class S, int N>
struct Mat {
    S a[N][N];
    static Mat identity() { Mat m; for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) m.a[i][j] = (i == j) ? S(1) : S(0); return m; }
};

template <class S, int N>
Mat<S, N> mul(const Mat<S, N>& A, const Mat<S, N>& B, int n) {
    Mat<S, N> C = Mat<S, N>::identity();
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) { S s = 0; for (int k = 0; k < n; ++k) s += A.a[i][k] * B.a[k][j]; C.a[i][j] = s; }
    return C;
}

// H <- P H P, U <- U P for the Householder reflector P = I - beta v v^T acting on rows/columns r0 .. r0+len-1
template <class S, int N>
void apply_reflector(Mat<S, N>& H, Mat<S, N>& U, int n, int r0, int len, const S* v, S beta) {
    for (int c = 0; c < n; ++c) {                       // rows r0.. of every column
        S s = 0; for (int k = 0; k < len; ++k) s += v[k] * H.a[r0 + k][c];
        s *= beta; for (int k = 0; k < len; ++k) H.a[r0 + k][c] -= s * v[k];
    }
    for (int r = 0; r < n; ++r) {                       // columns r0.. of every row
        S s = 0; for (int k = 0; k < len; ++k) s += H.a[r][r0 + k] * v[k];
        s *= beta; for (int k = 0; k < len; ++k) H.a[r][r0 + k] -= s * v[k];
        S t = 0; for (int k = 0; k < len; ++k) t += U.a[r][r0 + k] * v[k];
        t *= beta; for (int k = 0; k < len; ++k) U.a[r][r0 + k] -= t * v[k];
    }
}
// v, beta with (I - beta v v^T) x = -/+ |x| e1; false when x = 0 beyond its first entry (nothing to do)
template <class S>
bool make_reflector(const S* x, int len, S* v, S& beta) {
    S tail = 0; for (int k = 1; k < len; ++k) tail += x[k] * x[k];
    if (tail == S(0)) return false;
    const S nrm = std::sqrt(x[0] * x[0] + tail);
    const S alpha = (x[0] >= 0) ? -nrm : nrm;
    v[0] = x[0] - alpha; for (int k = 1; k < len; ++k) v[k] = x[k];
    S vv = 0; for (int k = 0; k < len; ++k) vv += v[k] * v[k];
    beta = S(2) / vv;
    return true;
}

// Real Schur form A = U T U^T (T quasi upper triangular, 2x2 blocks only for complex pairs), n <= N.
// Returns false if the QR iteration does not converge.
template <class S, int N>
bool real_schur(Mat<S, N>& T, Mat<S, N>& U, int n) {
    U = Mat<S, N>::identity();
    const S eps = std::numeric_limits<S>::epsilon();
    // Hessenberg reduction (GVL 7.4.2)
    for (int k = 0; k + 2 < n; ++k) {
        S x[N], v[N], beta;
        const int len = n - k - 1;
        for (int i = 0; i < len; ++i) x[i] = T.a[k + 1 + i][k];
        if (make_reflector(x, len, v, beta)) {
            apply_reflector(T, U, n, k + 1, len, v, beta);
            for (int i = k + 2; i < n; ++i) T.a[i][k] = 0;
        }
    }
    S norm = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) norm += std::fabs(T.a[i][j]);
    int hi = n - 1, iter = 0, total = 0;
    while (hi > 0) {
        int l = hi;
        while (l > 0) {
            S s = std::fabs(T.a[l - 1][l - 1]) + std::fabs(T.a[l][l]);
            if (s == S(0)) s = norm;
            if (std::fabs(T.a[l][l - 1]) <= eps * s) { T.a[l][l - 1] = 0; break; }
            --l;
        }
        if (l == hi) { --hi; iter = 0; continue; }
        if (l == hi - 1) {
            // 2x2 block: real eigenvalues -> rotate to upper triangular (first column of Q = eigenvector (lambda - d, c))
            const S a = T.a[hi - 1][hi - 1], b = T.a[hi - 1][hi], c = T.a[hi][hi - 1], d = T.a[hi][hi];
            const S p = S(0.5) * (a - d), q = p * p + b * c;
            if (q >= 0) {
                const S z = std::sqrt(q);
                const S w = (p >= 0) ? p + z : p - z;
                const S r = std::sqrt(w * w + c * c);
                const S cs = w / r, sn = c / r;
                for (int col = 0; col < n; ++col) {          // rows hi-1, hi  <- Q^T rows
                    const S t0 = T.a[hi - 1][col], t1 = T.a[hi][col];
                    T.a[hi - 1][col] = cs * t0 + sn * t1; T.a[hi][col] = -sn * t0 + cs * t1;
                }
                for (int row = 0; row < n; ++row) {          // columns hi-1, hi <- columns Q
                    const S t0 = T.a[row][hi - 1], t1 = T.a[row][hi];
                    T.a[row][hi - 1] = cs * t0 + sn * t1; T.a[row][hi] = -sn * t0 + cs * t1;
                    const S u0 = U.a[row][hi - 1], u1 = U.a[row][hi];
                    U.a[row][hi - 1] = cs * u0 + sn * u1; U.a[row][hi] = -sn * u0 + cs * u1;
                }
                T.a[hi][hi - 1] = 0;
            }
            hi -= 2; iter = 0; continue;
        }
        if (++total > 60 * n) return false;
        // Francis double-shift step on the active block l .. hi, EISPACK hqr form: the shift polynomial's first column is
        // built from differences (x - T[m][m]), so clustered eigenvalues (a pose update next to the identity) do not cancel
        S xx = T.a[hi][hi], yy = T.a[hi - 1][hi - 1], w = T.a[hi][hi - 1] * T.a[hi - 1][hi];
        if (iter == 10 || iter == 30) {                      // exceptional shift
            const S e = std::fabs(T.a[hi][hi - 1]) + std::fabs(T.a[hi - 1][hi - 2]);
            xx = yy = xx + S(0.75) * e; w = S(-0.4375) * e * e;
        }
        ++iter;
        int m = hi - 2;
        S p = 0, q = 0, r = 0;
        for (;; --m) {
            const S zz = T.a[m][m], rr = xx - zz, ss = yy - zz;
            p = (rr * ss - w) / T.a[m + 1][m] + T.a[m][m + 1];
            q = T.a[m + 1][m + 1] - zz - rr - ss;
            r = T.a[m + 2][m + 1];
            const S sc = std::fabs(p) + std::fabs(q) + std::fabs(r);
            if (sc != S(0)) { p /= sc; q /= sc; r /= sc; }
            if (m == l) break;
            const S lhs = std::fabs(T.a[m][m - 1]) * (std::fabs(q) + std::fabs(r));
            const S rhs = std::fabs(p) * (std::fabs(T.a[m - 1][m - 1]) + std::fabs(zz) + std::fabs(T.a[m + 1][m + 1]));
            if (lhs <= eps * rhs) break;
        }
        for (int k = m; k <= hi - 1; ++k) {
            const int len = (k == hi - 1) ? 2 : 3;
            S xv[3], v[3], beta;
            if (k == m) { xv[0] = p; xv[1] = q; xv[2] = r; }
            else { xv[0] = T.a[k][k - 1]; xv[1] = T.a[k + 1][k - 1]; xv[2] = (len == 3) ? T.a[k + 2][k - 1] : S(0); }
            if (!make_reflector(xv, len, v, beta)) continue;
            if (k == m && m > l) {
                // rows k.. of the columns from k on only; the negligible T[m][m-1] just changes sign (EISPACK hqr)
                const S keep = T.a[m][m - 1];
                T.a[m][m - 1] = 0;
                apply_reflector(T, U, n, k, len, v, beta);
                T.a[m][m - 1] = -keep;
            } else {
                apply_reflector(T, U, n, k, len, v, beta);
            }
            if (k > m) { T.a[k + 1][k - 1] = 0; if (len == 3) T.a[k + 2][k - 1] = 0; }
        }
    }
    return true;
}

// eigenvalues of a real n x n matrix; re/im of length n.  Real eigenvalues have im == 0 exactly (1x1 blocks and split 2x2 blocks).
template <class S, int N>
bool eigenvalues(const Mat<S, N>& A, int n, S* re, S* im) {
    Mat<S, N> T = A, U;
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (!std::isfinite(T.a[i][j])) return false;
    if (!real_schur(T, U, n)) return false;
    for (int i = 0; i < n;) {
        if (i == n - 1 || T.a[i + 1][i] == S(0)) { re[i] = T.a[i][i]; im[i] = 0; ++i; continue; }
        const S a = T.a[i][i], b = T.a[i][i + 1], c = T.a[i + 1][i], d = T.a[i + 1][i + 1];
        const S p = S(0.5) * (a - d), q = p * p + b * c;
        const S z = std::sqrt(std::fabs(q));
        re[i] = re[i + 1] = d + p; im[i] = z; im[i + 1] = -z;
        i += 2;
    }
    return true;
}

// X <- A^-1 B by Gaussian elimination with partial pivoting (all in S)
template <class S, int N>
Mat<S, N> solve(Mat<S, N> A, Mat<S, N> B, int n) {
    for (int c = 0; c < n; ++c) {
        int piv = c; for (int r = c + 1; r < n; ++r) if (std::fabs(A.a[r][c]) > std::fabs(A.a[piv][c])) piv = r;
        if (piv != c) for (int k = 0; k < n; ++k) { std::swap(A.a[c][k], A.a[piv][k]); std::swap(B.a[c][k], B.a[piv][k]); }
        for (int r = c + 1; r < n; ++r) {
            const S f = A.a[r][c] / A.a[c][c];
            for (int k = c; k < n; ++k) A.a[r][k] -= f * A.a[c][k];
            for (int k = 0; k < n; ++k) B.a[r][k] -= f * B.a[c][k];
        }
    }
    for (int c = n - 1; c >= 0; --c)
        for (int k = 0; k < n; ++k) {
            S s = B.a[c][k]; for (int j = c + 1; j < n; ++j) s -= A.a[c][j] * B.a[j][k];
            B.a[c][k] = s / A.a[c][c];
        }
    return B;
}

template <class S, int N>
S norm1_minus_identity(const Mat<S, N>& A, int n) {
    S best = 0;
    for (int c = 0; c < n; ++c) { S s = 0; for (int r = 0; r < n; ++r) s += std::fabs(A.a[r][c] - ((r == c) ? S(1) : S(0))); best = std::max(best, s); }
    return best;
}

// principal square root by the Denman-Beavers iteration (product-free form), all in S
template <class S, int N>
Mat<S, N> sqrt_db(const Mat<S, N>& A, int n) {
    Mat<S, N> Y = A, Z = Mat<S, N>::identity();
    const Mat<S, N> I = Mat<S, N>::identity();
    for (int it = 0; it < 50; ++it) {
        const Mat<S, N> Yi = solve(Y, I, n), Zi = solve(Z, I, n);
        S delta = 0;
        Mat<S, N> Yn = Y, Zn = Z;
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) {
            Yn.a[i][j] = S(0.5) * (Y.a[i][j] + Zi.a[i][j]); Zn.a[i][j] = S(0.5) * (Z.a[i][j] + Yi.a[i][j]);
            delta = std::max(delta, std::fabs(Yn.a[i][j] - Y.a[i][j]));
        }
        Y = Yn; Z = Zn;
        if (delta <= S(4) * std::numeric_limits<S>::epsilon()) break;
    }
    return Y;
}

// log(A) by inverse scaling and squaring on the real Schur form, Gauss-Legendre partial-fraction Pade of degree 3..5
// (single precision thresholds of Eigen's MatrixLogarithm.h); result = U log(T) U^T.
template <class S, int N>
bool logm(const Mat<S, N>& A, int n, Mat<S, N>& out) {
    Mat<S, N> T = A, U;
    if (!real_schur(T, U, n)) return false;
    static const double thr[3] = {2.5111573934555054e-1, 4.0535837411880493e-1, 5.3149729967117310e-1};
    static const double nodes[3][5] = {
        {0.1127016653792583114820734600217600, 0.5, 0.8872983346207416885179265399782400, 0, 0},
        {0.0694318442029737123880267555535953, 0.3300094782075718675986671204483777, 0.6699905217924281324013328795516223, 0.9305681557970262876119732444464048, 0},
        {0.0469100770306680036011865608503035, 0.2307653449471584544818427896498956, 0.5, 0.7692346550528415455181572103501044, 0.9530899229693319963988134391496965}};
    static const double weights[3][5] = {
        {0.2777777777777777777777777777777778, 0.4444444444444444444444444444444444, 0.2777777777777777777777777777777778, 0, 0},
        {0.1739274225687269286865319746109997, 0.3260725774312730713134680253890003, 0.3260725774312730713134680253890003, 0.1739274225687269286865319746109997, 0},
        {0.1184634425280945437571320203599587, 0.2393143352496832340206457574178191, 0.2844444444444444444444444444444444, 0.2393143352496832340206457574178191, 0.1184634425280945437571320203599587}};
    int roots = 0;
    S nrm = norm1_minus_identity(T, n);
    while (!(nrm < S(thr[2]))) {
        if (roots > 40 || !std::isfinite(nrm)) return false;
        T = sqrt_db(T, n); ++roots;
        nrm = norm1_minus_identity(T, n);
    }
    int deg = 3; while (deg < 5 && !(nrm <= S(thr[deg - 3]))) ++deg;
    Mat<S, N> X = T; for (int i = 0; i < n; ++i) X.a[i][i] -= S(1);
    Mat<S, N> L; for (int i = 0; i < N; ++i) for (int j = 0; j < N; ++j) L.a[i][j] = 0;
    for (int k = 0; k < deg; ++k) {
        Mat<S, N> M = Mat<S, N>::identity();
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) M.a[i][j] += S(nodes[deg - 3][k]) * X.a[i][j];
        const Mat<S, N> Y = solve(M, X, n);
        for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) L.a[i][j] += S(weights[deg - 3][k]) * Y.a[i][j];
    }
    const S scale = std::ldexp(S(1), roots);
    for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) L.a[i][j] *= scale;
    Mat<S, N> Ut = U; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) Ut.a[i][j] = U.a[j][i];
    out = mul(mul(U, L, n), Ut, n);
    return true;
}

}  // namespace refnoise
