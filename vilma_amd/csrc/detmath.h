// A natural logarithm that gives the SAME BITS on the host and on the device.
//
// The sweep's decisions are taken twice -- by sweep_decide_kernel on the device, which runs ahead,
// and by the host code that reports them (sweep.hip) -- and the two must never disagree.  Their
// arithmetic is IEEE double add / multiply / divide without fused multiply-add on both sides,
// which is bit-reproducible; the library logarithms are not (ocml's log on the device and glibc's
// on the host may differ in the last place).  Two logarithms sit on that path: log hyper_delta in
// the table lh = log h - 0.5 log_det (fast_vi_delta_grad, reference numerics.py:149-164), built by
// the device's M-step or by the host's vilma_set_hyper, and 0.5 rank log tau in the likelihood
// (fast_likelihood, numerics.py:31-46), where tau comes from the device's or the host's
// error-scaling update.  Both go through det_log.
//
// Algorithm and coefficients: the classic argument reduction x = 2^k (1 + f), sqrt(1/2) < 1 + f <
// sqrt(2), with log(1 + f) = f - f^2/2 + s (f^2/2 + R(s^2)), s = f / (2 + f), R the degree-14
// minimax polynomial of fdlibm's e_log.c (Sun Microsystems, 1993: "Permission to use, copy,
// modify, and distribute this software is freely granted, provided that this notice is
// preserved"); error below 1 ulp.  Written with explicit temporaries so that no compiler
// contraction or reassociation can apply.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__) || defined(__CUDACC__)
#define VILMA_HD __host__ __device__
#else
#define VILMA_HD
#endif

static VILMA_HD inline double det_log(double x) {
#pragma clang fp contract(off)
    const double ln2_hi = 6.93147180369123816490e-01, ln2_lo = 1.90821492927058770002e-10;
    const double Lg1 = 6.666666666666735130e-01, Lg2 = 3.999999999940941908e-01,
                 Lg3 = 2.857142874366239149e-01, Lg4 = 2.222219843214978396e-01,
                 Lg5 = 1.818357216161805012e-01, Lg6 = 1.531383769920937332e-01,
                 Lg7 = 1.479819860511658591e-01;
    uint64_t u;
    __builtin_memcpy(&u, &x, 8);
    int32_t hx = (int32_t)(u >> 32);
    const uint32_t lx = (uint32_t)u;
    int32_t k = 0;
    if (hx < 0x00100000) {                          // zero, subnormal or negative
        if (((hx & 0x7fffffff) | (int32_t)lx) == 0) return -__builtin_huge_val();
        if (hx < 0) return __builtin_nan("");
        k -= 54;
        x = x * 18014398509481984.0;                // 2^54
        __builtin_memcpy(&u, &x, 8);
        hx = (int32_t)(u >> 32);
    }
    if (hx >= 0x7ff00000) return x + x;             // inf or nan
    k += (hx >> 20) - 1023;
    hx &= 0x000fffff;
    const int32_t i0 = (hx + 0x95f64) & 0x100000;
    u = (u & 0xffffffffull) | ((uint64_t)(uint32_t)(hx | (i0 ^ 0x3ff00000)) << 32);   // x or x / 2
    __builtin_memcpy(&x, &u, 8);
    k += i0 >> 20;
    const double f = x - 1.0;
    const double dk = (double)k;
    if ((0x000fffff & (2 + hx)) < 3) {              // |f| < 2^-20
        if (f == 0.0) {
            if (k == 0) return 0.0;
            const double a = dk * ln2_hi, b = dk * ln2_lo;
            return a + b;
        }
        const double t = 0.33333333333333333 * f;
        const double ff = f * f;
        const double R = ff * (0.5 - t);
        if (k == 0) return f - R;
        const double a = dk * ln2_hi, b = dk * ln2_lo;
        return a - ((R - b) - f);
    }
    const double s = f / (2.0 + f);
    const double z = s * s;
    const int32_t i1 = hx - 0x6147a;
    const double w = z * z;
    const int32_t j1 = 0x6b851 - hx;
    double p1 = w * Lg6;
    p1 = Lg4 + p1;
    p1 = w * p1;
    p1 = Lg2 + p1;
    const double t1 = w * p1;
    double p2 = w * Lg7;
    p2 = Lg5 + p2;
    p2 = w * p2;
    p2 = Lg3 + p2;
    p2 = w * p2;
    p2 = Lg1 + p2;
    const double t2 = z * p2;
    const double R = t2 + t1;
    if ((i1 | j1) > 0) {
        const double hf = 0.5 * f;
        const double hfsq = hf * f;
        const double q = s * (hfsq + R);
        if (k == 0) return f - (hfsq - q);
        const double a = dk * ln2_hi, b = dk * ln2_lo;
        return a - ((hfsq - (q + b)) - f);
    }
    const double q = s * (f - R);
    if (k == 0) return f - q;
    const double a = dk * ln2_hi, b = dk * ln2_lo;
    return a - ((q - b) - f);
}

// lh = log h - 0.5 log_det (fast_vi_delta_grad, reference numerics.py:149-164), no contraction
static VILMA_HD inline double det_lh(double h, double log_det) {
#pragma clang fp contract(off)
    const double half = 0.5 * log_det;
    const double lg = det_log(h);
    return lg - half;
}
// 0.5 * ld_rank * log(tau): the cohort's constant in fast_likelihood (numerics.py:31-46)
static VILMA_HD inline double det_hrl(double rank, double tau) {
#pragma clang fp contract(off)
    const double a = 0.5 * rank;
    const double lg = det_log(tau);
    return a * lg;
}
// _update_error_scaling (variational_inference.py:472-486) from a cohort's sums
//   lin = sum m adj, quad = z^T R z, var = sum d v
static VILMA_HD inline double det_tau(double chi, double lin, double quad, double var, double rank) {
#pragma clang fp contract(off)
    const double two_lin = 2.0 * lin;
    double t = chi - two_lin;
    t = t + quad;
    t = t + var;
    return t / rank;
}
// the objective from the 3P+2 sums (include/vilma_hip.h): fast_likelihood (numerics.py:31-46) minus
// _beta_KL (variational_inference.py:873-885); hrl[p] = det_hrl(rank_p, tau_p)
static VILMA_HD inline double det_objective(int P, const double *chi, const double *tau,
                                            const double *hrl, const double *t) {
#pragma clang fp contract(off)
    double lik = 0.0;
    for (int p = 0; p < P; ++p) {
        const double sq = t[P + p] + t[2 * P + p];
        const double hsq = -0.5 * sq;
        const double num = hsq + t[p];
        const double hchi = 0.5 * chi[p];
        const double inner = (num - hchi) / tau[p];
        const double term = inner - hrl[p];
        lik = lik + term;
    }
    const double kl = t[3 * P] + t[3 * P + 1];
    return lik - kl;
}
