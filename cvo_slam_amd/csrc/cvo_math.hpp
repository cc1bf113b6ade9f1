// cvo_math.hpp -- scalar math of one CVO iteration, written once for both the
// device epilogue (one lane per workgroup) and the host-side state helpers of the
// C ABI.  Compiled with -ffp-contract=off: every float expression below is
// evaluated exactly as written (products and sums round separately), which is what
// keeps the HIP path on the same float trajectory as the reference's Eigen code.
//
// Reference: thirdparty/cvo/src/cvo.cpp:76-110,239-334,763-821 and
// thirdparty/cvo/src/LieGroup.cpp:20-27,159-186 (each function cites its lines).
// 3x3 matrices are row-major float[9].  Fixed-size 3-term reductions use Eigen's
// unrolled order t0+(t1+t2); dynamic-size ones are sequential.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace cvohip {

#define CVO_HD __host__ __device__ __forceinline__

CVO_HD float sum3f(float t0, float t1, float t2) { return t0 + (t1 + t2); }
CVO_HD float dot3_seq(const float* a, const float* b) { return (a[0] * b[0] + a[1] * b[1]) + a[2] * b[2]; }
CVO_HD void mat3_mul(const float* A, const float* B, float* C) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j)
            C[i * 3 + j] = sum3f(A[i * 3 + 0] * B[0 * 3 + j], A[i * 3 + 1] * B[1 * 3 + j], A[i * 3 + 2] * B[2 * 3 + j]);
}
CVO_HD void mat3_vec(const float* A, const float* x, float* y) {
    for (int i = 0; i < 3; ++i) y[i] = sum3f(A[i * 3 + 0] * x[0], A[i * 3 + 1] * x[1], A[i * 3 + 2] * x[2]);
}
CVO_HD void cross3(const float* a, const float* b, float* c) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
CVO_HD void skew3(const float* v, float* M) {   // LieGroup.cpp:20-27
    M[0] = 0;     M[1] = -v[2]; M[2] = v[1];
    M[3] = v[2];  M[4] = 0;     M[5] = -v[0];
    M[6] = -v[1]; M[7] = v[0];  M[8] = 0;
}
CVO_HD float norm3f(const float* a) { return sqrtf(sum3f(a[0] * a[0], a[1] * a[1], a[2] * a[2])); }

// update_tf, cvo.cpp:106-110: M = [R^T | -R^T T], 3x4 row-major
CVO_HD void make_transform(const float* R, const float* T, float* M) {
    for (int r = 0; r < 3; ++r) {
        const float n0 = -R[0 * 3 + r], n1 = -R[1 * 3 + r], n2 = -R[2 * 3 + r];
        M[r * 4 + 0] = R[0 * 3 + r]; M[r * 4 + 1] = R[1 * 3 + r]; M[r * 4 + 2] = R[2 * 3 + r];
        M[r * 4 + 3] = sum3f(n0 * T[0], n1 * T[1], n2 * T[2]);
    }
}
// transform.linear()*p + transform.translation(), cvo.cpp:338
CVO_HD void apply_transform(const float* M, float p0, float p1, float p2, float& y0, float& y1, float& y2) {
    y0 = sum3f(M[0] * p0, M[1] * p1, M[2] * p2) + M[3];
    y1 = sum3f(M[4] * p0, M[5] * p1, M[6] * p2) + M[7];
    y2 = sum3f(M[8] * p0, M[9] * p1, M[10] * p2) + M[11];
}

// gates, cvo.cpp:125-126 (se_kernel: log(sp/s2)) and :395-396 (scores: log(sp/sigma/sigma)).
// std::log(float) is what the reference's overload resolution picks: the float logarithm of ITS libm, which differs between libms in the last bit (the
// device's OCML logf and glibc's differ on 39 % of a sample of arguments, by up to 2 ulps: tests/test_gpu_pair_values.py).  Device, host and oracle take the
// correctly rounded float -- the double routine rounded once -- as for sin / cos; for the reference's constants that is glibc's value too.
CVO_HD float log_f32_cr(float x) { return (float)log((double)x); }
CVO_HD float gate_d2_align(float l, float sp_thres, float s2) { return (float)(-2.0 * l * l * (double)log_f32_cr(sp_thres / s2)); }
CVO_HD float gate_d2_score(float l, float sp_thres, float sigma) { return (float)(-2.0 * l * l * (double)log_f32_cr(sp_thres / sigma / sigma)); }
CVO_HD float gate_d2c(float c_ell, float sp_thres, float c_sigma) { return (float)(-2.0 * c_ell * c_ell * (double)log_f32_cr(sp_thres / c_sigma / c_sigma)); }

// poly_solver + root selection for 4E t^3 + 3D t^2 + 2C t + B (cvo.cpp:76-92,317-333).
// The reference takes f32 eigenvalues of the companion matrix of the monic cubic
// and keeps those with imag()==0; the real roots are computed here in closed form
// (double) from the same f32 monic coefficients.
// Real roots of the monic cubic t^3 + a t^2 + b t + c (double).  One root by Newton
// from 0, falling back to a bracketed Newton iteration (always converges: f(-R) < 0 < f(R)
// for the Cauchy bound R), the other two from the deflated quadratic (deflation direction chosen
// by the root's size so no cancellation), each polished on the full cubic.  Plain
// Cardano loses the sign of the discriminant when the roots differ by many orders
// of magnitude.
CVO_HD int cubic_real_roots(double a, double b, double c, double* roots) {
    // fast path: plain Newton from t = 0 (the step the line search wants is normally the small
    // root next to 0); accepted only if it converges, otherwise the bracketed iteration below.
    double x = 0.0;
    bool conv = false;
    for (int it = 0; it < 12; ++it) {
        const double f = ((x + a) * x + b) * x + c, df = (3.0 * x + 2.0 * a) * x + b;
        if (f == 0.0) { conv = true; break; }
        const double xn = x - f / df;
        if (!(df != 0.0) || !isfinite(xn)) break;
        if (fabs(xn - x) <= 1e-15 * fabs(xn)) { x = xn; conv = true; break; }
        x = xn;
    }
    if (!conv) {
        const double R = 1.0 + fmax(fabs(a), fmax(fabs(b), fabs(c)));
        double lo = -R, hi = R;
        x = 0.0;
        for (int it = 0; it < 200; ++it) {
            const double f = ((x + a) * x + b) * x + c, df = (3.0 * x + 2.0 * a) * x + b;
            if (f == 0.0) break;
            if (f < 0) lo = x; else hi = x;
            double xn = x - f / df;
            if (!(df != 0.0) || !(xn > lo && xn < hi)) xn = 0.5 * (lo + hi);
            if (fabs(xn - x) <= 1e-16 * fabs(xn) || xn == x) { x = xn; break; }
            x = xn;
        }
    }
    const double r = x;
    roots[0] = r;
    double p, q;                                   // t^2 + p t + q
    if (fabs(r) * fabs(r) * fabs(r) > fabs(c)) { q = -c / r; p = (q - b) / r; }   // large root: divide from the constant term up
    else { p = a + r; q = b + p * r; }                                              // small root: synthetic division from the top
    const double disc = p * p - 4.0 * q;
    if (!(disc >= 0.0)) return 1;
    const double s = -0.5 * (p + copysign(sqrt(disc), p));
    double r2 = s, r3 = (s != 0.0) ? q / s : 0.0;
    double* rr[2] = {&r2, &r3};
    for (int k = 0; k < 2; ++k) {
        double t = *rr[k];
        // The caller keeps the smallest root > 0 and clamps it to 0.8 (cvo.cpp:326-333): a root that is clearly negative or above 1 cannot
        // change the step whatever its last digits are (the deflated value is good to ~1e-15 relative), so it is not polished.
        if (t < -1e-6 || t > 1.0) continue;
        for (int it = 0; it < 2; ++it) {          // the deflated roots are already good to ~1e-15; two Newton steps on the full cubic
            const double f = ((t + a) * t + b) * t + c, df = (3.0 * t + 2.0 * a) * t + b;
            const double tn = t - f / df;
            if (!(df != 0.0) || !isfinite(tn)) break;
            t = tn;
        }
        *rr[k] = t;
    }
    roots[1] = r2; roots[2] = r3;
    return 3;
}

CVO_HD float cubic_step(float c3, float c2, float c1, float c0, float min_step) {
    const float p1f = c2 / c3, p2f = c1 / c3, p3f = c0 / c3;       // monic f32 coefficients, cvo.cpp:86
    const float FMAX = 3.402823466e+38f;
    float best = FMAX;
    if (isfinite(p1f) && isfinite(p2f) && isfinite(p3f)) {
        double roots[3];
        const int nr = cubic_real_roots(p1f, p2f, p3f, roots);
        for (int k = 0; k < nr; ++k) {
            const float tf = (float)roots[k];
            if (tf > 0 && tf < best) best = tf;                    // cvo.cpp:326-327
        }
    }
    float step = (best == FMAX) ? min_step : best;                 // cvo.cpp:330
    step = step > 0.8 ? (float)0.8 : step;                         // cvo.cpp:333
    return step;
}

// sin(float), cos(float) as the correctly rounded float: the reference calls its libm's float routines (LieGroup.cpp:174-175), whose last bit differs
// between libms (glibc 2.35's sinf is not the correctly rounded value for 1 % of the arguments above 0.03, OCML's for another 1.4 %:
// tests/test_gpu_pair_values.py); oracle and device both take the value of the double routine rounded once.  Below 0.5 -- every argument a converging
// alignment produces (step <= 0.8 times |omega| of a few hundredths) -- the series to x^17 (truncation < 1e-20) stands in for the double routine.
CVO_HD float sin_f32_cr(float x) {
    const double t = (double)x;
    if (!(fabs(t) < 0.5)) return (float)sin(t);
    const double z = t * t;
    double p = -1.0 / 355687428096000.0;                            // -1/17!
    p = fma(p, z, 1.0 / 1307674368000.0);                           //  1/15!
    p = fma(p, z, -1.0 / 6227020800.0);
    p = fma(p, z, 1.0 / 39916800.0);
    p = fma(p, z, -1.0 / 362880.0);
    p = fma(p, z, 1.0 / 5040.0);
    p = fma(p, z, -1.0 / 120.0);
    p = fma(p, z, 1.0 / 6.0);
    return (float)fma(-t * z, p, t);                                // t - t^3 (1/6 - ...)
}
CVO_HD float cos_f32_cr(float x) {
    const double t = (double)x;
    if (!(fabs(t) < 0.5)) return (float)cos(t);
    const double z = t * t;
    double p = 1.0 / 20922789888000.0;                              // 1/16!
    p = fma(p, z, -1.0 / 87178291200.0);                            // -1/14!
    p = fma(p, z, 1.0 / 479001600.0);
    p = fma(p, z, -1.0 / 3628800.0);
    p = fma(p, z, 1.0 / 40320.0);
    p = fma(p, z, -1.0 / 720.0);
    p = fma(p, z, 1.0 / 24.0);
    p = fma(p, z, -0.5);
    return (float)fma(z, p, 1.0);
}

// Exp_SEK3 (K=1), LieGroup.cpp:159-186, including the theta<1e-6 branch R=I, Jl=I (Q3)
CVO_HD void exp_sek3(const float* omega, const float* v, float dt, float* dR, float* dT) {
    const float I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    float Jl[9];
    const float theta = norm3f(omega);
    if (theta < 1e-6f) {
        for (int i = 0; i < 9; ++i) { dR[i] = I[i]; Jl[i] = I[i]; }
    } else {
        float A[9]; skew3(omega, A);
        const float theta2 = theta * theta;
        const float stheta = sin_f32_cr(dt * theta);
        const float ctheta = cos_f32_cr(dt * theta);
        const float oneMinusCosTheta2 = (1 - ctheta) / (theta2);
        float A2[9]; mat3_mul(A, A, A2);
        const float s1 = stheta / theta;
        const float s3 = (dt * theta - stheta) / (theta2 * theta);
        for (int i = 0; i < 9; ++i) {
            dR[i] = (I[i] + s1 * A[i]) + oneMinusCosTheta2 * A2[i];
            Jl[i] = (dt * I[i] + oneMinusCosTheta2 * A[i]) + s3 * A2[i];
        }
    }
    mat3_vec(Jl, v, dT);
}

// dist_se3, cvo.cpp:94-104: Frobenius norm of logm([dR dT; 0 1]) in closed form,
// sqrt(2 theta^2 + |V^-1 dT|^2), V^-1 = LeftJacobianInverse_SO3 (LieGroup.cpp:61-69);
// theta from atan2 so that the 1e-6-sized angles that decide the stop test survive.
CVO_HD float dist_se3(const float* dR, const float* dT) {
    const double w[3] = {0.5 * ((double)dR[7] - dR[5]), 0.5 * ((double)dR[2] - dR[6]), 0.5 * ((double)dR[3] - dR[1])};
    const double s = sqrt(w[0] * w[0] + w[1] * w[1] + w[2] * w[2]);
    const double c = 0.5 * ((double)dR[0] + dR[4] + dR[8] - 1.0);
    // theta = atan2(s, c) and the V^-1 coefficient 1/theta^2 - (1+cos)/(2 theta sin): for the small
    // per-iteration rotations (s < 0.1, c > 0) both come from their series (error < 1e-13), no trig
    double theta, coef;
    if (s < 0.1 && c > 0.0) {
        const double s2 = s * s;
        theta = s * (1.0 + s2 * (1.0 / 6.0 + s2 * (3.0 / 40.0 + s2 * (15.0 / 336.0 + s2 * (105.0 / 3456.0)))));   // asin(s)
        const double t2 = theta * theta;
        coef = 1.0 / 12.0 + t2 * (1.0 / 720.0 + t2 * (1.0 / 30240.0 + t2 * (1.0 / 1209600.0)));
    } else {
        theta = atan2(s, c);
        coef = (theta < 1e-4) ? 1.0 / 12.0 + theta * theta / 720.0
                              : 1.0 / (theta * theta) - (1.0 + cos(theta)) / (2.0 * theta * sin(theta));
    }
    double phi[3] = {w[0], w[1], w[2]};
    if (s > 1e-300) { const double f = theta / s; for (int k = 0; k < 3; ++k) phi[k] *= f; }
    const double t[3] = {dT[0], dT[1], dT[2]};
    const double pxt[3] = {phi[1] * t[2] - phi[2] * t[1], phi[2] * t[0] - phi[0] * t[2], phi[0] * t[1] - phi[1] * t[0]};
    const double ppxt[3] = {phi[1] * pxt[2] - phi[2] * pxt[1], phi[2] * pxt[0] - phi[0] * pxt[2], phi[0] * pxt[1] - phi[1] * pxt[0]};
    double u2 = 0;
    for (int k = 0; k < 3; ++k) { const double u = t[k] - 0.5 * pxt[k] + coef * ppxt[k]; u2 += u * u; }
    return (float)sqrt(2.0 * theta * theta + u2);
}

}  // namespace cvohip
