// Function API: the reference's 20 `vilma.numerics.*` functions as stand-alone device calls.
//
// The fit itself never comes through here -- its per-SNP work is the fused snp_pass / delta /
// init kernels of kernels.hip.  These entry points keep the third depth of the reference's
// boundary (SURVEY 8b "Function API", /root/reference/src/vilma/numerics.py:11-290): pure
// functions on whole arrays in the reference's layouts (vi_mu [M,P,N], vi_delta [N,M],
// vi_sigma / nat_sigma [M,P,P,N], hyper_delta [A,M]; float64 / int64, C order), returning
// fresh arrays.  Pointers may be host or device memory (copies go through hipMemcpyDefault);
// every call is synchronous.  vilma_amd/numerics.py is the Python face with the reference's
// names; include/vilma_numerics.h declares the C-ABI.
//
// Layout notes for gfx950: [.,N]-minor arrays are read one SNP per lane (coalesced).  The
// [N,M] arrays (vi_delta and friends) are row-major with the SNP index slowest, so per-SNP
// kernels bring them in through an LDS tile (128 SNPs x 16 components, rows read as 128-byte
// segments) instead of 64 lanes striding by M.  Reductions are two-stage with a fixed
// combination order (no atomics): results do not depend on scheduling.
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "../../include/vilma_numerics.h"

namespace {

constexpr int TB = 128;          // threads per workgroup (two staging tiles fit 64 KB of LDS)
constexpr int KT = 16;           // tile width (components) of the [N,M] staging tile
constexpr double EPSILON = 1e-100;   // numerics.py:8

thread_local std::string g_err;

int fail(const char *what, hipError_t e) {
    g_err = std::string(what) + ": " + hipGetErrorString(e);
    return 1;
}
int fail(const std::string &what) {
    g_err = what;
    return 1;
}

// device buffer that frees itself; upload() takes host or device pointers
struct Dev {
    void *p = nullptr;
    size_t bytes = 0;
    ~Dev() {
        if (p) (void)hipFree(p);
    }
    hipError_t alloc(size_t n) {
        bytes = n;
        return hipMalloc(&p, n ? n : 8);
    }
    hipError_t upload(const void *src, size_t n) {
        hipError_t e = alloc(n);
        if (e != hipSuccess || !n) return e;
        return hipMemcpy(p, src, n, hipMemcpyDefault);
    }
    hipError_t download(void *dst) const {
        return bytes ? hipMemcpy(dst, p, bytes, hipMemcpyDefault) : hipSuccess;
    }
    double *d() const { return (double *)p; }
    int64_t *i() const { return (int64_t *)p; }
};

#define TRY(expr)                                          \
    do {                                                   \
        hipError_t e_ = (expr);                            \
        if (e_ != hipSuccess) return fail(#expr, e_);      \
    } while (0)

int finish(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(what, e);
    e = hipDeviceSynchronize();
    if (e != hipSuccess) return fail(what, e);
    return 0;
}

inline int blocks_for(int64_t n, int per = TB) {
    int64_t b = (n + per - 1) / per;
    return (int)(b < 1 ? 1 : b);
}
// grid for a grid-stride pass / a first reduction stage: enough to fill 256 CUs several times
inline int stride_grid(int64_t n) {
    int64_t b = (n + TB - 1) / TB;
    return (int)(b < 1 ? 1 : (b > 4096 ? 4096 : b));
}

// ---- workgroup sum, fixed order: lanes by xor-shuffle, waves in index order ----------------
__device__ inline double block_sum(double v) {
    __shared__ double part[TB / 64];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    const int w = threadIdx.x >> 6;
    __syncthreads();                       // part[] may still be read from a previous call
    if ((threadIdx.x & 63) == 0) part[w] = v;
    __syncthreads();
    double s = part[0];
#pragma unroll
    for (int ww = 1; ww < TB / 64; ++ww) s += part[ww];
    return s;
}

// out[c] = sum_r partials[c * rows + r], one workgroup per column, rows in a fixed order
__global__ __launch_bounds__(TB) void reduce_rows_kernel(const double *__restrict__ partials,
                                                         int rows, double *__restrict__ out) {
    const double *col = partials + (int64_t)blockIdx.x * rows;
    double s = 0.0;
    for (int r = threadIdx.x; r < rows; r += TB) s += col[r];
    s = block_sum(s);
    if (threadIdx.x == 0) out[blockIdx.x] = s;
}

// ---- elementwise (numerics.py:11-28) -------------------------------------------------------
__global__ __launch_bounds__(TB) void sum_betas_kernel(const double *__restrict__ old_beta,
                                                       const double *__restrict__ new_beta,
                                                       double step, int64_t n,
                                                       double *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TB)
        out[t] = step * new_beta[t] + (1. - step) * old_beta[t];
}
__global__ __launch_bounds__(TB) void divide_kernel(const double *__restrict__ x,
                                                    const double *__restrict__ y, int64_t n,
                                                    double *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TB)
        out[t] = x[t] / y[t];
}
__global__ __launch_bounds__(TB) void linked_ests_kernel(const double *__restrict__ w,
                                                         const double *__restrict__ x,
                                                         const double *__restrict__ y,
                                                         const double *__restrict__ z, int64_t n,
                                                         double *__restrict__ out) {
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TB)
        out[t] = w[t] / x[t] - y[t] * z[t];
}

// ---- fast_likelihood (numerics.py:31-46): grid (chunks, P) -> partials[p][chunk] -----------
__global__ __launch_bounds__(TB) void likelihood_kernel(
    const double *__restrict__ post_means, const double *__restrict__ post_vars,
    const double *__restrict__ scaled_mu, const double *__restrict__ scaled_ld_diags,
    const double *__restrict__ linked_ests, const double *__restrict__ adj_marginal, int64_t N,
    double *__restrict__ partials) {
    const int64_t o = (int64_t)blockIdx.y * N;
    double s = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * TB + threadIdx.x; i < N; i += (int64_t)gridDim.x * TB)
        s += -0.5 * (scaled_ld_diags[o + i] * post_vars[o + i] + linked_ests[o + i] * scaled_mu[o + i])
             + post_means[o + i] * adj_marginal[o + i];
    s = block_sum(s);
    if (threadIdx.x == 0) partials[(int64_t)blockIdx.y * gridDim.x + blockIdx.x] = s;
}
__global__ void likelihood_final_kernel(const double *__restrict__ lik, const double *__restrict__ chi,
                                        const double *__restrict__ ranks,
                                        const double *__restrict__ tau, int P,
                                        double *__restrict__ out) {
    if (threadIdx.x || blockIdx.x) return;
    double total = 0.0;
    for (int p = 0; p < P; ++p)
        total += (lik[p] - 0.5 * chi[p]) / tau[p] - 0.5 * ranks[p] * log(tau[p]);
    out[0] = total;
}

// ---- the [N,M] staging tile ----------------------------------------------------------------
struct Tile {
    double v[TB][KT + 1];      // +1: rows land in different LDS banks
};
// rows i0 .. i0+TB-1 (zero beyond N), columns k0 .. k0+kn-1 of a row-major [N, ld] array
__device__ inline void tile_load(Tile &t, const double *__restrict__ src, int64_t i0, int64_t N,
                                 int ld, int k0, int kn) {
    for (int e = threadIdx.x; e < TB * KT; e += TB) {
        const int r = e / KT, c = e % KT;
        const int64_t i = i0 + r;
        t.v[r][c] = (i < N && c < kn) ? src[i * ld + k0 + c] : 0.0;
    }
}
__device__ inline void tile_store(const Tile &t, double *__restrict__ dst, int64_t i0, int64_t N,
                                  int ld, int k0, int kn) {
    for (int e = threadIdx.x; e < TB * KT; e += TB) {
        const int r = e / KT, c = e % KT;
        const int64_t i = i0 + r;
        if (i < N && c < kn) dst[i * ld + k0 + c] = t.v[r][c];
    }
}

// ---- fast_posterior_mean / fast_pmv (numerics.py:49-65): grid (SNP tiles, P) ---------------
template <bool PMV>
__global__ __launch_bounds__(TB) void posterior_kernel(const double *__restrict__ vi_mu,
                                                       const double *__restrict__ vi_delta,
                                                       const double *__restrict__ temp,
                                                       const double *__restrict__ mean, int M,
                                                       int P, int64_t N,
                                                       double *__restrict__ out) {
    __shared__ Tile tile;
    const int64_t i0 = (int64_t)blockIdx.x * TB, i = i0 + threadIdx.x;
    const int p = blockIdx.y;
    const bool live = i < N;
    const int64_t ii = live ? i : N - 1;
    double s = 0.0;
    for (int k0 = 0; k0 < M; k0 += KT) {
        const int kn = min(KT, M - k0);
        __syncthreads();
        tile_load(tile, vi_delta, i0, N, M, k0, kn);
        __syncthreads();
        for (int c = 0; c < kn; ++c) {
            const int64_t at = ((int64_t)(k0 + c) * P + p) * N + ii;
            const double mu = vi_mu[at];
            const double x = PMV ? temp[at] + mu * mu : mu;
            s += x * tile.v[threadIdx.x][c];
        }
    }
    if (live) {
        const int64_t at = (int64_t)p * N + i;
        out[at] = PMV ? s - mean[at] * mean[at] : s;
    }
}

// ---- fast_nat_inner_product(_m2) (numerics.py:68-95): one thread per (s, p, i) -------------
__global__ __launch_bounds__(TB) void nat_inner_kernel(const double *__restrict__ vi_mu,
                                                       const double *__restrict__ nat_sigma, int M,
                                                       int P, int64_t N, double scale,
                                                       double *__restrict__ out) {
    const int64_t total = (int64_t)M * P * N;
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * TB) {
        const int64_t i = t % N, sp = t / N;
        const int64_t s = sp / P;
        double acc = 0.0;
        for (int q = 0; q < P; ++q)
            acc += nat_sigma[(sp * P + q) * N + i] * vi_mu[(s * P + q) * N + i];
        out[t] = scale * acc;
    }
}

// ---- fast_inner_product_comp (numerics.py:98-115) -------------------------------------------
__global__ __launch_bounds__(TB) void inner_product_comp_kernel(
    const double *__restrict__ vi_mu, const double *__restrict__ mixture_prec,
    const double *__restrict__ vi_delta, int M, int P, int64_t N, double *__restrict__ partials) {
    __shared__ Tile tile;
    const int64_t i0 = (int64_t)blockIdx.x * TB, i = i0 + threadIdx.x;
    const bool live = i < N;
    const int64_t ii = live ? i : N - 1;
    double total = 0.0;
    for (int k0 = 0; k0 < M; k0 += KT) {
        const int kn = min(KT, M - k0);
        __syncthreads();
        tile_load(tile, vi_delta, i0, N, M, k0, kn);
        __syncthreads();
        for (int c = 0; c < kn; ++c) {
            const int k = k0 + c;
            double t = 0.0;
            for (int p = 0; p < P; ++p) {
                const double mp = vi_mu[((int64_t)k * P + p) * N + ii];
                for (int q = 0; q < P; ++q)
                    t += mp * vi_mu[((int64_t)k * P + q) * N + ii]
                         * mixture_prec[((int64_t)k * P + q) * P + p];
            }
            total += t * tile.v[threadIdx.x][c];
        }
    }
    total = block_sum(live ? total : 0.0);
    if (threadIdx.x == 0) partials[blockIdx.x] = total;
}
__global__ void scale_scalar_kernel(double *x, double f) {
    if (!threadIdx.x && !blockIdx.x) x[0] *= f;
}

// ---- sum_annotations (numerics.py:118-129) ---------------------------------------------------
// grid (row chunks, A, column tiles of 64); lanes along components, the waves take interleaved
// rows; partials[(a * M + k) * chunks + chunk]
__global__ __launch_bounds__(TB) void sum_annotations_kernel(
    const double *__restrict__ deltas, const int64_t *__restrict__ annotations, int M, int64_t N,
    int64_t rows_per_chunk, double *__restrict__ partials) {
    __shared__ double part[TB / 64][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int k = blockIdx.z * 64 + lane;
    const int64_t a = blockIdx.y;
    const int64_t r0 = (int64_t)blockIdx.x * rows_per_chunk;
    const int64_t r1 = min(N, r0 + rows_per_chunk);
    double s = 0.0;
    if (k < M)
        for (int64_t i = r0 + w; i < r1; i += TB / 64)
            if (annotations[i] == a) s += deltas[i * M + k];
    part[w][lane] = s;
    __syncthreads();
    if (w == 0 && k < M) {
        double t = part[0][lane];
#pragma unroll
        for (int ww = 1; ww < TB / 64; ++ww) t += part[ww][lane];
        partials[(a * M + k) * gridDim.x + blockIdx.x] = t;
    }
}

// ---- tables [A,M]: log hyper, and log hyper - 0.5 log_det ----------------------------------
__global__ void log_table_kernel(const double *__restrict__ hyper, const double *__restrict__ log_det,
                                 int A, int M, double *__restrict__ out) {
    for (int t = blockIdx.x * blockDim.x + threadIdx.x; t < A * M; t += gridDim.x * blockDim.x)
        out[t] = log(hyper[t]) + (log_det ? -0.5 * log_det[t % M] : 0.0);
}

// ---- fast_delta_kl (numerics.py:132-141): flat over [N,M] ----------------------------------
__global__ __launch_bounds__(TB) void delta_kl_kernel(const double *__restrict__ vi_delta,
                                                      const double *__restrict__ log_hyper,
                                                      const int64_t *__restrict__ annotations,
                                                      int M, int64_t N,
                                                      double *__restrict__ partials) {
    const int64_t total = N * M;
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * TB) {
        const int64_t i = t / M;
        const int k = (int)(t - i * M);
        const double d = vi_delta[t];
        s += d * (log(d) - log_hyper[annotations[i] * M + k]);
    }
    s = block_sum(s);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ---- fast_beta_kl (numerics.py:144-146) ------------------------------------------------------
__global__ __launch_bounds__(TB) void dot_kernel(const double *__restrict__ x,
                                                 const double *__restrict__ y, int64_t n,
                                                 double *__restrict__ partials) {
    double s = 0.0;
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < n; t += (int64_t)gridDim.x * TB)
        s += x[t] * y[t];
    s = block_sum(s);
    if (threadIdx.x == 0) partials[blockIdx.x] = s;
}

// ---- fast_vi_delta_grad (numerics.py:149-164): flat over [N, M-1] --------------------------
__global__ __launch_bounds__(TB) void vi_delta_grad_kernel(const double *__restrict__ full,
                                                           const int64_t *__restrict__ annotations,
                                                           int M, int64_t N,
                                                           double *__restrict__ out) {
    const int K = M - 1;
    const int64_t total = N * K;
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * TB) {
        const int64_t i = t / K;
        const int k = (int)(t - i * K);
        const double *row = full + annotations[i] * M;
        out[t] = row[k] - row[M - 1];
    }
}

// ---- map_to_nat_cat_2D (numerics.py:167-176): flat over [N, K-1] ---------------------------
__global__ __launch_bounds__(TB) void map_to_nat_kernel(const double *__restrict__ probs, int K,
                                                        int64_t N, double *__restrict__ out) {
    const int Ko = K - 1;
    const int64_t total = N * Ko;
    for (int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x; t < total;
         t += (int64_t)gridDim.x * TB) {
        const int64_t i = t / Ko;
        const int k = (int)(t - i * Ko);
        out[t] = log(probs[i * K + k]) - log(probs[i * K + K - 1]);
    }
}

// ---- invert_nat_cat_2D (numerics.py:179-195): probs [N,K] -> out [N,K+1] --------------------
// one thread per row; three sweeps over the row's tiles (max, denominator, outputs)
__global__ __launch_bounds__(TB) void invert_nat_kernel(const double *__restrict__ probs, int K,
                                                        int64_t N, double *__restrict__ out) {
    __shared__ Tile tile;
    const int64_t i0 = (int64_t)blockIdx.x * TB;
    const int r = threadIdx.x;
    double max_p = 0.0;                              // np.maximum(np.max(probs[i]), 0)
    for (int k0 = 0; k0 < K; k0 += KT) {
        const int kn = min(KT, K - k0);
        __syncthreads();
        tile_load(tile, probs, i0, N, K, k0, kn);
        __syncthreads();
        for (int c = 0; c < kn; ++c) max_p = fmax(max_p, tile.v[r][c]);
    }
    const double last_p = exp(-max_p);
    double denom = last_p;
    for (int k0 = 0; k0 < K; k0 += KT) {
        const int kn = min(KT, K - k0);
        __syncthreads();
        tile_load(tile, probs, i0, N, K, k0, kn);
        __syncthreads();
        for (int c = 0; c < kn; ++c) denom += exp(tile.v[r][c] - max_p);
    }
    for (int k0 = 0; k0 < K; k0 += KT) {
        const int kn = min(KT, K - k0);
        __syncthreads();
        tile_load(tile, probs, i0, N, K, k0, kn);
        __syncthreads();
        double o[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) o[c] = fmax(exp(tile.v[r][c] - max_p) / denom, EPSILON);
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KT; ++c) tile.v[r][c] = o[c];
        __syncthreads();
        tile_store(tile, out, i0, N, K + 1, k0, kn);
    }
    if (i0 + r < N) out[(i0 + r) * (K + 1) + K] = fmax(last_p / denom, EPSILON);
}

// ---- the logits of fast_invert_nat_vi_delta (numerics.py:198-211): to_invert [N, M-1] -------
__global__ __launch_bounds__(TB) void nat_vi_delta_logits_kernel(
    const double *__restrict__ new_mu, const double *__restrict__ nat_mu,
    const double *__restrict__ const_part, const double *__restrict__ nat_vi_delta, int M, int P,
    int64_t N, double *__restrict__ to_invert) {
    __shared__ Tile tc, tn;
    const int64_t i0 = (int64_t)blockIdx.x * TB, i = i0 + threadIdx.x;
    const int r = threadIdx.x;
    const bool live = i < N;
    const int64_t ii = live ? i : N - 1;
    double last = const_part[ii * M + (M - 1)];
    for (int j = 0; j < P; ++j) {
        const int64_t at = ((int64_t)(M - 1) * P + j) * N + ii;
        last += new_mu[at] * nat_mu[at];
    }
    const int K = M - 1;
    for (int k0 = 0; k0 < K; k0 += KT) {
        const int kn = min(KT, K - k0);
        __syncthreads();
        tile_load(tc, const_part, i0, N, M, k0, kn);
        tile_load(tn, nat_vi_delta, i0, N, K, k0, kn);
        __syncthreads();
        double o[KT];
#pragma unroll
        for (int c = 0; c < KT; ++c) {
            const int k = min(k0 + c, K - 1);
            double add = tc.v[r][c];
            for (int j = 0; j < P; ++j) {
                const int64_t at = ((int64_t)k * P + j) * N + ii;
                add += new_mu[at] * nat_mu[at];
            }
            o[c] = 0.5 * (add - last) + tn.v[r][c];
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < KT; ++c) tc.v[r][c] = o[c];
        __syncthreads();
        tile_store(tc, to_invert, i0, N, K, k0, kn);
    }
}

// ---- batched P x P inverse / log|det| (numerics.py:216-290) ---------------------------------
// matrix t = (t / inner, t % inner) sits at base = (t / inner) * outer + (t % inner) with its
// elements `es` apart: [n,P,P] is (inner 1, outer P*P, es 1); [M,P,P,N] is (inner N, outer
// P*P*N, es N), which keeps lanes along the SNP axis.
struct MatLayout {
    int64_t count, inner, outer, es;
};
__device__ inline int64_t mat_base(const MatLayout &l, int64_t t) {
    return (t / l.inner) * l.outer + (t % l.inner);
}

// closed forms of _matrix_invert_4d_numba for P <= 2 (the 2 x 2 result is written symmetric,
// numerics.py:231-232); Gauss-Jordan with partial pivoting otherwise (np.linalg.inv's LU)
template <int P, bool CLOSED>
__global__ __launch_bounds__(TB) void mat_inverse_kernel(const double *__restrict__ m,
                                                         MatLayout l, double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= l.count) return;
    const int64_t b = mat_base(l, t);
    double a[P][P];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < P; ++q) a[p][q] = m[b + (p * P + q) * l.es];
    double inv[P][P];
    if (CLOSED && P == 1) {
        inv[0][0] = 1. / a[0][0];
    } else if (CLOSED && P == 2) {
        const double det = 1. / (a[0][0] * a[P - 1][P - 1] - a[0][P - 1] * a[P - 1][0]);
        inv[0][0] = a[P - 1][P - 1] * det;
        inv[P - 1][P - 1] = a[0][0] * det;
        inv[0][P - 1] = -a[0][P - 1] * det;
        inv[P - 1][0] = inv[0][P - 1];
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int q = 0; q < P; ++q) inv[p][q] = p == q ? 1.0 : 0.0;
#pragma unroll
        for (int c = 0; c < P; ++c) {
#pragma unroll
            for (int r = c + 1; r < P; ++r) {          // bring the largest |a[r][c]| to row c
                const bool swap = fabs(a[r][c]) > fabs(a[c][c]);
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const double x = a[c][q], y = a[r][q], u = inv[c][q], v = inv[r][q];
                    a[c][q] = swap ? y : x;
                    a[r][q] = swap ? x : y;
                    inv[c][q] = swap ? v : u;
                    inv[r][q] = swap ? u : v;
                }
            }
            const double piv = 1. / a[c][c];
#pragma unroll
            for (int q = 0; q < P; ++q) {
                a[c][q] *= piv;
                inv[c][q] *= piv;
            }
#pragma unroll
            for (int r = 0; r < P; ++r) {
                if (r == c) continue;
                const double f = a[r][c];
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    a[r][q] -= f * a[c][q];
                    inv[r][q] -= f * inv[c][q];
                }
            }
        }
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < P; ++q) out[b + (p * P + q) * l.es] = inv[p][q];
}

// closed forms of _matrix_log_det_4d_numba for P <= 2 (log of the signed determinant: NaN for
// a negative one, as np.log gives); log|det| by pivoted elimination otherwise (slogdet()[1])
template <int P, bool CLOSED>
__global__ __launch_bounds__(TB) void mat_log_det_kernel(const double *__restrict__ m, MatLayout l,
                                                         double *__restrict__ out) {
    const int64_t t = (int64_t)blockIdx.x * TB + threadIdx.x;
    if (t >= l.count) return;
    const int64_t b = mat_base(l, t);
    double a[P][P];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < P; ++q) a[p][q] = m[b + (p * P + q) * l.es];
    double res;
    if (CLOSED && P == 1) {
        res = log(a[0][0]);
    } else if (CLOSED && P == 2) {
        res = log(a[0][0] * a[P - 1][P - 1] - a[0][P - 1] * a[P - 1][0]);
    } else {
        res = 0.0;
#pragma unroll
        for (int c = 0; c < P; ++c) {
#pragma unroll
            for (int r = c + 1; r < P; ++r) {
                const bool swap = fabs(a[r][c]) > fabs(a[c][c]);
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    const double x = a[c][q], y = a[r][q];
                    a[c][q] = swap ? y : x;
                    a[r][q] = swap ? x : y;
                }
            }
            res += log(fabs(a[c][c]));
            const double piv = 1. / a[c][c];
#pragma unroll
            for (int r = c + 1; r < P; ++r) {
                const double f = a[r][c] * piv;
#pragma unroll
                for (int q = 0; q < P; ++q) a[r][q] -= f * a[c][q];
            }
        }
    }
    out[t] = res;      // [count] in matrix order: [n] or [M,N]
}

template <bool CLOSED>
int launch_inverse(int P, const double *m, const MatLayout &l, double *out) {
    const dim3 grid(blocks_for(l.count)), block(TB);
    switch (P) {
        case 1: hipLaunchKernelGGL((mat_inverse_kernel<1, CLOSED>), grid, block, 0, 0, m, l, out); break;
        case 2: hipLaunchKernelGGL((mat_inverse_kernel<2, CLOSED>), grid, block, 0, 0, m, l, out); break;
        case 3: hipLaunchKernelGGL((mat_inverse_kernel<3, false>), grid, block, 0, 0, m, l, out); break;
        case 4: hipLaunchKernelGGL((mat_inverse_kernel<4, false>), grid, block, 0, 0, m, l, out); break;
        case 5: hipLaunchKernelGGL((mat_inverse_kernel<5, false>), grid, block, 0, 0, m, l, out); break;
        case 6: hipLaunchKernelGGL((mat_inverse_kernel<6, false>), grid, block, 0, 0, m, l, out); break;
        case 7: hipLaunchKernelGGL((mat_inverse_kernel<7, false>), grid, block, 0, 0, m, l, out); break;
        case 8: hipLaunchKernelGGL((mat_inverse_kernel<8, false>), grid, block, 0, 0, m, l, out); break;
        default: return fail("matrix inverse: P must be 1..8");
    }
    return 0;
}
template <bool CLOSED>
int launch_log_det(int P, const double *m, const MatLayout &l, double *out) {
    const dim3 grid(blocks_for(l.count)), block(TB);
    switch (P) {
        case 1: hipLaunchKernelGGL((mat_log_det_kernel<1, CLOSED>), grid, block, 0, 0, m, l, out); break;
        case 2: hipLaunchKernelGGL((mat_log_det_kernel<2, CLOSED>), grid, block, 0, 0, m, l, out); break;
        case 3: hipLaunchKernelGGL((mat_log_det_kernel<3, false>), grid, block, 0, 0, m, l, out); break;
        case 4: hipLaunchKernelGGL((mat_log_det_kernel<4, false>), grid, block, 0, 0, m, l, out); break;
        case 5: hipLaunchKernelGGL((mat_log_det_kernel<5, false>), grid, block, 0, 0, m, l, out); break;
        case 6: hipLaunchKernelGGL((mat_log_det_kernel<6, false>), grid, block, 0, 0, m, l, out); break;
        case 7: hipLaunchKernelGGL((mat_log_det_kernel<7, false>), grid, block, 0, 0, m, l, out); break;
        case 8: hipLaunchKernelGGL((mat_log_det_kernel<8, false>), grid, block, 0, 0, m, l, out); break;
        default: return fail("matrix log-determinant: P must be 1..8");
    }
    return 0;
}

// second stage of a scalar reduction: partials[rows] -> scalar[0]
void reduce_scalar(const double *partials, int rows, double *scalar) {
    hipLaunchKernelGGL(reduce_rows_kernel, dim3(1), dim3(TB), 0, 0, partials, rows, scalar);
}

int check_annotations(int64_t N, int A, const Dev &dev) {
    // an annotation outside [0, A) would index past the [A,M] tables on the device
    std::vector<int64_t> h((size_t)N);
    if (N && hipMemcpy(h.data(), dev.p, (size_t)N * 8, hipMemcpyDefault) != hipSuccess)
        return fail("annotations: copy failed");
    for (int64_t i = 0; i < N; ++i)
        if (h[i] < 0 || h[i] >= A) return fail("annotations: index out of range");
    return 0;
}

}  // namespace

extern "C" {

const char *vilma_num_last_error(void) { return g_err.c_str(); }

int vilma_num_sum_betas(const double *old_beta, const double *new_beta, double step_size,
                        int64_t n, double *out) {
    if (n <= 0) return 0;
    Dev a, b, o;
    TRY(a.upload(old_beta, n * 8));
    TRY(b.upload(new_beta, n * 8));
    TRY(o.alloc(n * 8));
    hipLaunchKernelGGL(sum_betas_kernel, dim3(stride_grid(n)), dim3(TB), 0, 0, a.d(), b.d(),
                       step_size, n, o.d());
    if (finish("sum_betas")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_divide(const double *x, const double *y, int64_t n, double *out) {
    if (n <= 0) return 0;
    Dev a, b, o;
    TRY(a.upload(x, n * 8));
    TRY(b.upload(y, n * 8));
    TRY(o.alloc(n * 8));
    hipLaunchKernelGGL(divide_kernel, dim3(stride_grid(n)), dim3(TB), 0, 0, a.d(), b.d(), n, o.d());
    if (finish("fast_divide")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_linked_ests(const double *w, const double *x, const double *y, const double *z,
                          int64_t n, double *out) {
    if (n <= 0) return 0;
    Dev a, b, c, d, o;
    TRY(a.upload(w, n * 8));
    TRY(b.upload(x, n * 8));
    TRY(c.upload(y, n * 8));
    TRY(d.upload(z, n * 8));
    TRY(o.alloc(n * 8));
    hipLaunchKernelGGL(linked_ests_kernel, dim3(stride_grid(n)), dim3(TB), 0, 0, a.d(), b.d(),
                       c.d(), d.d(), n, o.d());
    if (finish("fast_linked_ests")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_likelihood(const double *post_means, const double *post_vars,
                         const double *scaled_mu, const double *scaled_ld_diags,
                         const double *linked_ests, const double *adj_marginal,
                         const double *chi_stat, const double *ld_ranks,
                         const double *error_scaling, int P, int64_t N, double *out) {
    if (P < 0 || N < 0) return fail("fast_likelihood: negative size");
    Dev pm, pv, sm, sd, le, am, chi, rk, tau, partials, lik, res;
    const size_t pn = (size_t)P * N * 8;
    TRY(pm.upload(post_means, pn));
    TRY(pv.upload(post_vars, pn));
    TRY(sm.upload(scaled_mu, pn));
    TRY(sd.upload(scaled_ld_diags, pn));
    TRY(le.upload(linked_ests, pn));
    TRY(am.upload(adj_marginal, pn));
    TRY(chi.upload(chi_stat, (size_t)P * 8));
    TRY(rk.upload(ld_ranks, (size_t)P * 8));
    TRY(tau.upload(error_scaling, (size_t)P * 8));
    const int chunks = stride_grid(N);
    TRY(partials.alloc((size_t)(P ? P : 1) * chunks * 8));
    TRY(lik.alloc((size_t)(P ? P : 1) * 8));
    TRY(res.alloc(8));
    if (P > 0) {
        hipLaunchKernelGGL(likelihood_kernel, dim3(chunks, P), dim3(TB), 0, 0, pm.d(), pv.d(),
                           sm.d(), sd.d(), le.d(), am.d(), N, partials.d());
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(P), dim3(TB), 0, 0, partials.d(), chunks,
                           lik.d());
    }
    hipLaunchKernelGGL(likelihood_final_kernel, dim3(1), dim3(1), 0, 0, lik.d(), chi.d(), rk.d(),
                       tau.d(), P, res.d());
    if (finish("fast_likelihood")) return 1;
    TRY(res.download(out));
    return 0;
}

static int posterior_common(bool pmv, const double *mean, const double *vi_mu,
                            const double *vi_delta, const double *temp, int M, int P, int64_t N,
                            double *out) {
    if ((int64_t)P * N <= 0) return 0;
    Dev mu, dl, tp, mn, o;
    const size_t mpn = (size_t)M * P * N * 8;
    TRY(mu.upload(vi_mu, mpn));
    TRY(dl.upload(vi_delta, (size_t)N * M * 8));
    if (pmv) {
        TRY(tp.upload(temp, mpn));
        TRY(mn.upload(mean, (size_t)P * N * 8));
    }
    TRY(o.alloc((size_t)P * N * 8));
    const dim3 grid(blocks_for(N), P), block(TB);
    if (pmv)
        hipLaunchKernelGGL((posterior_kernel<true>), grid, block, 0, 0, mu.d(), dl.d(), tp.d(),
                           mn.d(), M, P, N, o.d());
    else
        hipLaunchKernelGGL((posterior_kernel<false>), grid, block, 0, 0, mu.d(), dl.d(),
                           (const double *)nullptr, (const double *)nullptr, M, P, N, o.d());
    if (finish(pmv ? "fast_pmv" : "fast_posterior_mean")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_posterior_mean(const double *vi_mu, const double *vi_delta, int M, int P, int64_t N,
                             double *out) {
    return posterior_common(false, nullptr, vi_mu, vi_delta, nullptr, M, P, N, out);
}

int vilma_num_pmv(const double *mean, const double *vi_mu, const double *vi_delta,
                  const double *temp, int M, int P, int64_t N, double *out) {
    return posterior_common(true, mean, vi_mu, vi_delta, temp, M, P, N, out);
}

int vilma_num_nat_inner_product(const double *vi_mu, const double *nat_sigma, int M, int P,
                                int64_t N, double scale, double *out) {
    const int64_t total = (int64_t)M * P * N;
    if (total <= 0) return 0;
    Dev mu, ns, o;
    TRY(mu.upload(vi_mu, (size_t)total * 8));
    TRY(ns.upload(nat_sigma, (size_t)total * P * 8));
    TRY(o.alloc((size_t)total * 8));
    hipLaunchKernelGGL(nat_inner_kernel, dim3(stride_grid(total)), dim3(TB), 0, 0, mu.d(), ns.d(),
                       M, P, N, scale, o.d());
    if (finish("fast_nat_inner_product")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_inner_product_comp(const double *vi_mu, const double *mixture_prec,
                                 const double *vi_delta, int M, int P, int64_t N, double *out) {
    Dev mu, pr, dl, partials, res;
    const int nb = blocks_for(N);
    TRY(mu.upload(vi_mu, (size_t)M * P * N * 8));
    TRY(pr.upload(mixture_prec, (size_t)M * P * P * 8));
    TRY(dl.upload(vi_delta, (size_t)N * M * 8));
    TRY(partials.alloc((size_t)nb * 8));
    TRY(res.alloc(8));
    if (N > 0) {
        hipLaunchKernelGGL(inner_product_comp_kernel, dim3(nb), dim3(TB), 0, 0, mu.d(), pr.d(),
                           dl.d(), M, P, N, partials.d());
        reduce_scalar(partials.d(), nb, res.d());
        hipLaunchKernelGGL(scale_scalar_kernel, dim3(1), dim3(1), 0, 0, res.d(), 0.5);
    } else {
        TRY(hipMemset(res.p, 0, 8));
    }
    if (finish("fast_inner_product_comp")) return 1;
    TRY(res.download(out));
    return 0;
}

int vilma_num_sum_annotations(const double *deltas, const int64_t *annotations, int A, int M,
                              int64_t N, double *out) {
    if ((int64_t)A * M <= 0) return 0;
    Dev dl, an, partials, o;
    TRY(dl.upload(deltas, (size_t)N * M * 8));
    TRY(an.upload(annotations, (size_t)N * 8));
    TRY(o.alloc((size_t)A * M * 8));
    if (N > 0) {
        int64_t rows = (N + 1023) / 1024;
        if (rows < 64) rows = 64;
        const int chunks = (int)((N + rows - 1) / rows);
        TRY(partials.alloc((size_t)A * M * chunks * 8));
        hipLaunchKernelGGL(sum_annotations_kernel, dim3(chunks, A, (M + 63) / 64), dim3(TB), 0, 0,
                           dl.d(), an.i(), M, N, rows, partials.d());
        hipLaunchKernelGGL(reduce_rows_kernel, dim3(A * M), dim3(TB), 0, 0, partials.d(), chunks,
                           o.d());
    } else {
        TRY(hipMemset(o.p, 0, o.bytes));
    }
    if (finish("sum_annotations")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_delta_kl(const double *vi_delta, const double *hyper_delta,
                       const int64_t *annotations, int A, int M, int64_t N, double *out) {
    Dev dl, hy, an, lh, partials, res;
    TRY(dl.upload(vi_delta, (size_t)N * M * 8));
    TRY(hy.upload(hyper_delta, (size_t)A * M * 8));
    TRY(an.upload(annotations, (size_t)N * 8));
    if (check_annotations(N, A, an)) return 1;
    TRY(lh.alloc((size_t)A * M * 8));
    TRY(res.alloc(8));
    const int64_t total = N * M;
    if (total > 0) {
        const int nb = stride_grid(total);
        TRY(partials.alloc((size_t)nb * 8));
        hipLaunchKernelGGL(log_table_kernel, dim3(blocks_for((int64_t)A * M)), dim3(TB), 0, 0,
                           hy.d(), (const double *)nullptr, A, M, lh.d());
        hipLaunchKernelGGL(delta_kl_kernel, dim3(nb), dim3(TB), 0, 0, dl.d(), lh.d(), an.i(), M, N,
                           partials.d());
        reduce_scalar(partials.d(), nb, res.d());
    } else {
        TRY(hipMemset(res.p, 0, 8));
    }
    if (finish("fast_delta_kl")) return 1;
    TRY(res.download(out));
    return 0;
}

int vilma_num_beta_kl(const double *sigma_summary, const double *vi_delta, int64_t n,
                      double *out) {
    Dev a, b, partials, res;
    TRY(a.upload(sigma_summary, (size_t)(n > 0 ? n : 0) * 8));
    TRY(b.upload(vi_delta, (size_t)(n > 0 ? n : 0) * 8));
    TRY(res.alloc(8));
    if (n > 0) {
        const int nb = stride_grid(n);
        TRY(partials.alloc((size_t)nb * 8));
        hipLaunchKernelGGL(dot_kernel, dim3(nb), dim3(TB), 0, 0, a.d(), b.d(), n, partials.d());
        reduce_scalar(partials.d(), nb, res.d());
        hipLaunchKernelGGL(scale_scalar_kernel, dim3(1), dim3(1), 0, 0, res.d(), 0.5);
    } else {
        TRY(hipMemset(res.p, 0, 8));
    }
    if (finish("fast_beta_kl")) return 1;
    TRY(res.download(out));
    return 0;
}

int vilma_num_vi_delta_grad(const double *hyper_delta, const double *log_det,
                            const int64_t *annotations, int A, int M, int64_t N, double *out) {
    const int64_t total = N * (M - 1);
    if (total <= 0) return 0;
    Dev hy, ld, an, full, o;
    TRY(hy.upload(hyper_delta, (size_t)A * M * 8));
    TRY(ld.upload(log_det, (size_t)M * 8));
    TRY(an.upload(annotations, (size_t)N * 8));
    if (check_annotations(N, A, an)) return 1;
    TRY(full.alloc((size_t)A * M * 8));
    TRY(o.alloc((size_t)total * 8));
    hipLaunchKernelGGL(log_table_kernel, dim3(blocks_for((int64_t)A * M)), dim3(TB), 0, 0, hy.d(),
                       ld.d(), A, M, full.d());
    hipLaunchKernelGGL(vi_delta_grad_kernel, dim3(stride_grid(total)), dim3(TB), 0, 0, full.d(),
                       an.i(), M, N, o.d());
    if (finish("fast_vi_delta_grad")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_map_to_nat_cat(const double *probs, int64_t N, int K, double *out) {
    const int64_t total = N * (K - 1);
    if (total <= 0) return 0;
    Dev pr, o;
    TRY(pr.upload(probs, (size_t)N * K * 8));
    TRY(o.alloc((size_t)total * 8));
    hipLaunchKernelGGL(map_to_nat_kernel, dim3(stride_grid(total)), dim3(TB), 0, 0, pr.d(), K, N,
                       o.d());
    if (finish("map_to_nat_cat_2D")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_invert_nat_cat(const double *probs, int64_t N, int K, double *out) {
    if (N <= 0) return 0;
    Dev pr, o;
    TRY(pr.upload(probs, (size_t)N * K * 8));
    TRY(o.alloc((size_t)N * (K + 1) * 8));
    hipLaunchKernelGGL(invert_nat_kernel, dim3(blocks_for(N)), dim3(TB), 0, 0, pr.d(), K, N, o.d());
    if (finish("invert_nat_cat_2D")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_invert_nat_vi_delta(const double *new_mu, const double *nat_mu,
                                  const double *const_part, const double *nat_vi_delta, int M,
                                  int P, int64_t N, double *out) {
    if (N <= 0) return 0;
    if (M < 1) return fail("fast_invert_nat_vi_delta: M must be >= 1");
    Dev nm, na, cp, nv, logits, o;
    const size_t mpn = (size_t)M * P * N * 8;
    TRY(nm.upload(new_mu, mpn));
    TRY(na.upload(nat_mu, mpn));
    TRY(cp.upload(const_part, (size_t)N * M * 8));
    TRY(nv.upload(nat_vi_delta, (size_t)N * (M - 1) * 8));
    TRY(logits.alloc((size_t)N * (M - 1) * 8));
    TRY(o.alloc((size_t)N * M * 8));
    if (M > 1)
        hipLaunchKernelGGL(nat_vi_delta_logits_kernel, dim3(blocks_for(N)), dim3(TB), 0, 0, nm.d(),
                           na.d(), cp.d(), nv.d(), M, P, N, logits.d());
    hipLaunchKernelGGL(invert_nat_kernel, dim3(blocks_for(N)), dim3(TB), 0, 0, logits.d(), M - 1, N,
                       o.d());
    if (finish("fast_invert_nat_vi_delta")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_matrix_invert(const double *mats, int64_t n, int P, int closed_form, double *out) {
    if (n <= 0 || P <= 0) return 0;
    Dev m, o;
    TRY(m.upload(mats, (size_t)n * P * P * 8));
    TRY(o.alloc((size_t)n * P * P * 8));
    const MatLayout l{n, 1, (int64_t)P * P, 1};
    if (closed_form ? launch_inverse<true>(P, m.d(), l, o.d())
                    : launch_inverse<false>(P, m.d(), l, o.d()))
        return 1;
    if (finish("matrix_invert")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_matrix_log_det(const double *mats, int64_t n, int P, int closed_form, double *out) {
    if (n <= 0) return 0;
    if (P <= 0) return fail("matrix_log_det: P must be >= 1");
    Dev m, o;
    TRY(m.upload(mats, (size_t)n * P * P * 8));
    TRY(o.alloc((size_t)n * 8));
    const MatLayout l{n, 1, (int64_t)P * P, 1};
    if (closed_form ? launch_log_det<true>(P, m.d(), l, o.d())
                    : launch_log_det<false>(P, m.d(), l, o.d()))
        return 1;
    if (finish("matrix_log_det")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_vi_sigma_inv(const double *matrices, int M, int P, int64_t N, double *out) {
    const int64_t count = (int64_t)M * N;
    if (count <= 0 || P <= 0) return 0;
    Dev m, o;
    TRY(m.upload(matrices, (size_t)count * P * P * 8));
    TRY(o.alloc((size_t)count * P * P * 8));
    const MatLayout l{count, N, (int64_t)P * P * N, N};
    if (launch_inverse<true>(P, m.d(), l, o.d())) return 1;
    if (finish("vi_sigma_inv")) return 1;
    TRY(o.download(out));
    return 0;
}

int vilma_num_vi_sigma_log_det(const double *matrices, int M, int P, int64_t N, double *out) {
    const int64_t count = (int64_t)M * N;
    if (count <= 0) return 0;
    if (P <= 0) return fail("vi_sigma_log_det: P must be >= 1");
    Dev m, o;
    TRY(m.upload(matrices, (size_t)count * P * P * 8));
    TRY(o.alloc((size_t)count * 8));
    const MatLayout l{count, N, (int64_t)P * P * N, N};
    if (launch_log_det<true>(P, m.d(), l, o.d())) return 1;
    if (finish("vi_sigma_log_det")) return 1;
    TRY(o.download(out));
    return 0;
}

}  // extern "C"
