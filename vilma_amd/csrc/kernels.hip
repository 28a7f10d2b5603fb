// Hand-written gfx950 (CDNA4, wave64) kernels for the `vilma fit` hot path.
//
//   ld_tile_kernel     block-diagonal LD product on dense symmetric blocks: streams only the
//                      lower triangle (128-column slab panels), forming column sums and row sums
//                      in one pass; one workgroup takes the rows of up to four neighbouring slabs,
//                      merges their partial sums in LDS and stores once (a block that fits one
//                      item writes y itself); ld_tile_combine_kernel adds the partials of larger
//                      blocks in a fixed order -- replaces BlockDiagonalMatrix.dot (reference
//                      matrix_structures.py:389-408).  HBM-streaming, 16-byte coalesced loads.
//                      ld_sym_kernel + ld_sym_combine_kernel: rounds 1 - 4's form of the same
//                      product, one workgroup per slab chunk (VILMA_LD_TILE=0: the A/B baseline).
//   ld_eig_fused_kernel  eigen-form blocks, y = U (s * (U^T x)) with U read once: a slab of columns
//                      of the column-major U stays in registers between its two uses --
//                      LowRankMatrix.dot (matrix_structures.py:148-152); ld_eig_wave_kernel
//                      (blocks up to 512 SNPs, a wave per slab) and ld_eig_tall_kernel (3 073 ..
//                      6 144 SNPs, 512 threads) are the same product.  ld_colsum_kernel +
//                      ld_rowsum_kernel: two passes over a row-major U, for blocks too tall for
//                      the fused kernels (> 6 144 SNPs).
//   snp_pass_kernel    fused per-SNP pass: natural-gradient blend, new_mu, mixture
//                      responsibilities (softmax against a fixed shift), posterior moments, KL and
//                      likelihood partial sums, the M-step's responsibility sums -- replaces
//                      numerics.py:11-146, 179-213 and variational_inference.py:762-823, 873-885
//                      for one or two candidate points.
//   delta_kernel       responsibilities of a state reduced per annotation (numerics.py:118-129)
//                      or written out; with MAT the state lazy trials reached, a (stored vi_mu) +
//                      Sig c, derived on the way (and stored, when somebody needs the array).
//                      mstep_kernel: the M-step table from those sums.
//   sweep_decide_kernel  the decisions of the sweep loop on the device (decide.h).
//   finalize / reduce / mean_diff   deterministic fixed-order reductions of per-workgroup
//                      partials and the convergence statistics.  No atomics anywhere.
//
// All arithmetic is IEEE double.  vi_sigma, nat_sigma, vi_sigma_log_det, vi_sigma_matches and
// sigma_summary ([M,P,P,N] / [N,M] arrays in the reference, variational_inference.py:712-733)
// are never materialised: they are recomputed per (component, SNP) from mixture_prec [M,P,P]
// and scaled_ld_diags/tau, which turns ~5 full-array streams per pass into ALU work.
#include "kernels.h"
#include "detmath.h"
#include "decide.h"

#include <algorithm>
#include <cstdlib>
#include <type_traits>

#define NEG_INF (-__builtin_huge_val())

static __device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
// sum over aligned groups of W consecutive lanes (W = 64: the whole wave)
template <int W>
static __device__ __forceinline__ double seg_sum(double v) {
#pragma unroll
    for (int o = W / 2; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
static __device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}

// --------------------------------------------------------------------------------------------
// LD product: out[c] = sum_j a[j][c] * x[j] for a 128-column slab.  Lanes run along the columns
// (2 doubles = 16 B per lane, 1 KiB per wave-load, fully coalesced), the 4 waves of the
// workgroup take interleaved groups of 8 rows, x[j] is wave-uniform (scalar loads), and there
// is no cross-lane reduction at all: each lane owns its two output columns.  The four per-wave
// partials are combined through LDS in a fixed order, so the result is deterministic.
// --------------------------------------------------------------------------------------------
#define CS_WAVES 4
#define CS_ROWS 8

// the matrix pointer comes out of the item table, so hipcc would treat it as a generic (flat)
// pointer; pin it to the global address space to get global_load_dwordx4
typedef double v2d __attribute__((ext_vector_type(2)));
typedef const v2d __attribute__((address_space(1))) *gd2_ptr;
// The LD stream is read exactly once per product: its loads are non-temporal (`nt`), so 7 GB of
// once-read lines do not churn through the 256 MB Infinity Cache.  Measured on ld_sym_kernel
// alone at C3: 1.14-1.18 ms -> 1.01 ms (6.1 -> 6.9 TB/s on the stored bytes).
#ifndef LD_NT
#define LD_NT 1
#endif
#if LD_NT
#define LD_STREAM_LOAD(p) __builtin_nontemporal_load((gd2_ptr)(p))
#else
#define LD_STREAM_LOAD(p) (*(gd2_ptr)(p))
#endif
// How ld_sym_kernel's partial sums leave the workgroup.  They are 0.8 % of the bytes the kernel
// reads (55 MB beside 6.4 GB at C3) and cost it 8 - 12 % when written the obvious way: a thin stream
// of small plain stores beside a read stream that saturates HBM (profiles/r04a_ld_levels_probe.txt:
// without the stores the kernel reads at the bare read's rate; a bare read with the same stores
// beside it slows down as much).  What helps is to write each chunk's row sums ONCE, whole, when
// the workgroup has finished streaming (staged in LDS meanwhile), with write-through stores (sc1:
// the lines do not linger dirty in L2 until the read stream evicts them): 1.03 -> 0.92 - 1.00 ms per
// launch at C3 (profiles/r04c_ld_partial_stores.txt; either half alone buys nothing).
// -DLD_STORE_MODE=k makes diagnostic builds (profiles/ld_levels_probe.py --build-variants):
// 1 = NO stores (results wrong: timing only); 2 = plain stores as they come (rounds 1 - 3);
// 3 = 2 non-temporal; 4 = staged, plain stores; 5 = as they come, write-through; 6 = staged,
// write-through 8-byte stores (the product uses 16-byte ones).
#ifndef LD_STORE_MODE
#define LD_STORE_MODE 0
#endif
#define LD_STAGED (LD_STORE_MODE == 0 || LD_STORE_MODE == 4 || LD_STORE_MODE == 6)
#define LD_MAX_CHUNK_ROWS 512       // rows of a work item (vilma_ctx::chunk_rows is capped at this)
// LD_PARTIAL_STORE2(p, a, b): two consecutive doubles at a 16-byte aligned p.  Write-through
// stores narrower than 16 bytes go out as one fabric write each (MI355X_MICROARCH.md), so the product
// writes its partial sums as global_store_dwordx4 ... sc1.
typedef int ld_v4i __attribute__((ext_vector_type(4)));
static __device__ __forceinline__ void ld_store2_wt(double *p, double a, double b) {
    const ld_v4i v = {__double2loint(a), __double2hiint(a), __double2loint(b), __double2hiint(b)};
    // A store of more than 8 bytes reads its data registers late: a vector-ALU write to them within two
    // wait states of the store reaches memory instead of the value stored (the compiler pads its own
    // stores; it cannot see into this one and is free to reuse the registers at once -- seen in round 5:
    // the first right-hand side's partial sums overwritten by the second one's arithmetic, a few
    // entries per launch).  The s_nop keeps the slots empty.
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v) : "memory");
}
#if LD_STORE_MODE == 1
#define LD_PARTIAL_STORE(p, v) do { if ((v) == 1.2345e300) *(p) = (v); } while (0)
#define LD_PARTIAL_STORE2(p, a, b) do { if ((a) == 1.2345e300) { (p)[0] = (a); (p)[1] = (b); } } while (0)
#elif LD_STORE_MODE == 2 || LD_STORE_MODE == 4
#define LD_PARTIAL_STORE(p, v) (*(p) = (v))
#define LD_PARTIAL_STORE2(p, a, b) (*(v2d *)(p) = v2d{(a), (b)})
#elif LD_STORE_MODE == 3
#define LD_PARTIAL_STORE(p, v) __builtin_nontemporal_store((v), (p))
#define LD_PARTIAL_STORE2(p, a, b) __builtin_nontemporal_store(v2d{(a), (b)}, (v2d *)(p))
#elif LD_STORE_MODE == 5 || LD_STORE_MODE == 6
#define LD_PARTIAL_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LD_PARTIAL_STORE2(p, a, b) do { LD_PARTIAL_STORE((p), (a)); LD_PARTIAL_STORE((p) + 1, (b)); } while (0)
#else
#define LD_PARTIAL_STORE(p, v) __hip_atomic_store((p), (v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)
#define LD_PARTIAL_STORE2(p, a, b) ld_store2_wt((p), (a), (b))
#endif
// Wave-uniform reads of the small per-component tables (mixture precisions, log-weights): through
// the constant address space the compiler may use scalar loads (SGPR results, scalar cache) even
// in kernels that also store to global memory -- from a generic pointer it must assume the
// stores alias the table, falls back to vector loads and then waits on vmcnt(0), i.e. on every
// outstanding vi_mu load AND store, once per component.  The tables are written only by other
// kernels (mstep, uploads), never by the kernel reading them.
// vi_mu streams of the per-SNP kernels.  The new vi_mu of a trial (0.67 GB at C3) is stored
// non-temporally: written back normally it is still draining from the caches while the LD
// product that follows streams, and takes bandwidth from it (sweep -2.5 % at C3 on top of the
// LD loads; the trial pass itself -7 % at M = 40).  Non-temporal vi_mu LOADS measured neutral.
#ifndef MU_NT
#define MU_NT 0
#endif
#ifndef MUOUT_NT
#define MUOUT_NT 1
#endif
#if MU_NT
#define MU_LOAD(p) __builtin_nontemporal_load(p)
#else
#define MU_LOAD(p) (*(p))
#endif
#if MUOUT_NT
#define MU_STORE(p, v) __builtin_nontemporal_store((v), (p))
#else
#define MU_STORE(p, v) (*(p) = (v))
#endif
// Layout of the vi_mu buffers in HBM.  The reference's array is [M][P][N] (SNP axis contiguous):
// a wave that walks the components of its 64 SNPs then touches M*P rows 8 N bytes apart, 512 B in
// each -- M*P DRAM pages and address translations per tile.  The buffers hold the same numbers tile
// by tile instead, so that the M*P pieces of a tile of 64 SNPs are one contiguous run and every wave
// of the per-SNP kernels reads (and writes) a linear stream:
//   MU_TILED 1   [ceil(N/64)][M*P][64]            8 bytes per lane and row (r05a)
//   MU_TILED 2   [ceil(N/64)][ceil(M*P/2)][64][2]  rows r = k P + p in PAIRS (2j, 2j+1): a lane holds
//                both rows of a pair side by side, so one global_load_dwordx4 / global_store_dwordx4
//                moves 16 bytes per lane, 1 KiB per wave -- half the memory instructions
//   MU_TILED 0   the reference's layout (diagnostic builds)
// vilma_set_mu / vilma_get_mu convert at the boundary (mu_tile_kernel), so nothing outside the
// kernels sees the difference.
#ifndef MU_TILED
#define MU_TILED 2
#endif
#define MU_TILE 64
// element (row r = k P + p) of SNP ii: base + MU_ROW(r), base = MU_BASE(ii, M, P, N)
#if MU_TILED == 2
#define MU_PAIRS(M, P) (((int64_t)(M) * (P) + 1) / 2)
#define MU_BASE(ii, M, P, N64) ((int64_t)((ii) >> 6) * (MU_PAIRS(M, P) * 2 * MU_TILE) + 2 * ((ii) & 63))
#define MU_ROW(r, N64) ((int64_t)((r) >> 1) * (2 * MU_TILE) + ((r) & 1))
// pair j (rows 2j, 2j+1) of the SNP whose base is `base`: a 16-byte aligned v2d
#define MU_PAIR(j) ((int64_t)(j) * (2 * MU_TILE))
#elif MU_TILED == 1
#define MU_BASE(ii, M, P, N64) ((int64_t)((ii) >> 6) * ((int64_t)(M) * (P) * MU_TILE) + ((ii) & 63))
#define MU_ROW(r, N64) ((int64_t)(r) * MU_TILE)
#else
#define MU_BASE(ii, M, P, N64) ((int64_t)(ii))
#define MU_ROW(r, N64) ((int64_t)(r) * (N64))
#endif
#define MU_PAIRED (MU_TILED == 2)
#if MU_PAIRED
#if MU_NT
#define MU_LOAD2(p) __builtin_nontemporal_load((const v2d *)(p))
#else
#define MU_LOAD2(p) (*(const v2d *)(p))
#endif
#if MUOUT_NT
#define MU_STORE2(p, a, b) __builtin_nontemporal_store(v2d{(a), (b)}, (v2d *)(p))
#else
#define MU_STORE2(p, a, b) (*(v2d *)(p) = v2d{(a), (b)})
#endif
#endif
int64_t mu_buffer_elems(int64_t N, int M, int P) {
#if MU_TILED == 2
    return (N + MU_TILE - 1) / MU_TILE * MU_TILE * 2 * MU_PAIRS(M, P);
#elif MU_TILED == 1
    return (N + MU_TILE - 1) / MU_TILE * MU_TILE * (int64_t)M * P;
#else
    return N * (int64_t)M * P;
#endif
}
// rows [r0, r0 + R) of the natural [M*P][N] array (`nat`: a staging chunk [R][N]) <-> the buffer's
// layout; to_tiled: nat -> buf (lanes past N fill the last tile with zeros), else buf -> nat
__global__ __launch_bounds__(256) void mu_tile_kernel(double *__restrict__ buf, double *__restrict__ nat,
                                                       int64_t N, int MP, int r0, int R, bool to_tiled) {
    const int64_t npad = (N + MU_TILE - 1) / MU_TILE * MU_TILE;
    const int64_t total = npad * R;
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < total; o += (int64_t)gridDim.x * 256) {
        const int rr = (int)(o / npad);
        const int64_t i = o % npad;
        double *b = buf + MU_BASE(i, MP, 1, N) + MU_ROW(r0 + rr, N);
        if (to_tiled) {
            *b = i < N ? nat[(int64_t)rr * N + i] : 0.0;
#if MU_PAIRED
            if (r0 + rr == MP - 1 && (MP & 1)) b[1] = 0.0;       // the unpaired last row's partner
#endif
        } else if (i < N) nat[(int64_t)rr * N + i] = *b;
    }
}
bool mu_is_tiled() { return MU_TILED != 0; }
void launch_mu_tile(double *buf, double *nat, int64_t N, int MP, int r0, int R, bool to_tiled, hipStream_t s) {
    const int64_t total = (N + MU_TILE - 1) / MU_TILE * MU_TILE * R;
    const int grid = (int)std::min<int64_t>((total + 255) / 256, 1 << 16);
    hipLaunchKernelGGL(mu_tile_kernel, dim3(grid), dim3(256), 0, s, buf, nat, N, MP, r0, R, to_tiled);
}
// (Prec + D)^-1 by Cholesky with IEEE sqrt and divide -- for outputs (vi_sigma_kernel), where the
// per-SNP passes' fast reciprocal square root (pass_rsqrt, ~2 ulp) has no business
template <int P>
static __device__ __forceinline__ void spd_inverse_exact(const double (&lam)[P][P], double (&sig)[P][P]) {
    double G[P][P], Gi[P][P];
#pragma unroll
    for (int j = 0; j < P; ++j) {
        double s = lam[j][j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= G[j][k] * G[j][k];
        G[j][j] = sqrt(s);
        Gi[j][j] = 1.0 / G[j][j];
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double t = lam[i][j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= G[i][k] * G[j][k];
            G[i][j] = t * Gi[j][j];
        }
    }
#pragma unroll
    for (int j = 0; j < P; ++j) {
#pragma unroll
        for (int i = j + 1; i < P; ++i) {
            double t = 0.0;
#pragma unroll
            for (int k = j; k < i; ++k) t += G[i][k] * Gi[k][j];
            Gi[i][j] = -t * Gi[i][i];
        }
    }
#pragma unroll
    for (int a = 0; a < P; ++a) {
#pragma unroll
        for (int b = 0; b <= a; ++b) {
            double t = 0.0;
#pragma unroll
            for (int k = a; k < P; ++k) t += Gi[k][a] * Gi[k][b];
            sig[a][b] = t;
            sig[b][a] = t;
        }
    }
}

// out[(i - i0) * M + k] = in[k * N + i] for i in [i0, i0 + n): the [M][N] responsibilities the delta
// pass writes, turned into the reference's [N][M] rows for the host (32 x 32 tiles through LDS, both
// sides coalesced)
__global__ __launch_bounds__(256) void transpose_km_kernel(const double *__restrict__ in, double *__restrict__ out,
                                                            int64_t N, int M, int64_t i0, int n) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;        // 32 x 8
    const int64_t ib = (int64_t)blockIdx.x * 32;
    const int kb = blockIdx.y * 32;
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int k = kb + ty + r;
        const int64_t i = ib + tx;
        tile[ty + r][tx] = (k < M && i < n) ? in[(int64_t)k * N + i0 + i] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int r = 0; r < 32; r += 8) {
        const int64_t i = ib + ty + r;
        const int k = kb + tx;
        if (i < n && k < M) out[i * M + k] = tile[tx][ty + r];
    }
}
void launch_transpose_km(const double *in, double *out, int64_t N, int M, int64_t i0, int n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(transpose_km_kernel, dim3((n + 31) / 32, (M + 31) / 32), dim3(256), 0, s, in, out, N, M, i0, n);
}

// vi_sigma [M][P][P][N] for the outputs (`vilma fit` writes it into the .npz; reference
// variational_inference.py:712-724, numerics.py:216-290): Sig_ki = (Prec_k + diag(sld_i / tau))^-1 for
// components [k0, k0 + nk) into out [nk][P][P][N].  Off the sweep path.  One and two cohorts use the
// reference's closed forms in plain IEEE operations (no contraction): the same bits numpy gives.
template <int P>
__global__ __launch_bounds__(256) void vi_sigma_kernel(const double *__restrict__ prec, const double *__restrict__ sld,
                                                        const TauArg tau, int64_t N, int k0, double *__restrict__ out) {
#pragma clang fp contract(off)
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= N) return;
    const int kk = blockIdx.y, k = k0 + kk;
    double lam[P][P], sig[P][P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
#pragma unroll
        for (int q = 0; q < P; ++q) lam[p][q] = prec[((int64_t)k * P + p) * P + q];
        const double d = sld[p * N + i] / tau.v[p];
        lam[p][p] = d + lam[p][p];
    }
    if constexpr (P == 1) {
        sig[0][0] = 1.0 / lam[0][0];
    } else if constexpr (P == 2) {
        const double ad = lam[0][0] * lam[1][1];
        const double bc = lam[0][1] * lam[1][0];
        const double r = 1.0 / (ad - bc);
        sig[0][0] = lam[1][1] * r;
        sig[1][1] = lam[0][0] * r;
        sig[1][0] = -lam[1][0] * r;
        sig[0][1] = sig[1][0];
    } else {
        spd_inverse_exact<P>(lam, sig);
    }
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < P; ++q) out[(((int64_t)kk * P + p) * P + q) * N + i] = sig[p][q];
}
template <int P>
static void launch_vi_sigma_p(const double *prec, const double *sld, const TauArg &tau, int64_t N, int k0, int nk,
                              double *out, hipStream_t s) {
    hipLaunchKernelGGL(vi_sigma_kernel<P>, dim3((unsigned)((N + 255) / 256), nk), dim3(256), 0, s, prec, sld, tau, N, k0, out);
}
void launch_vi_sigma(int P, const double *prec, const double *sld, const TauArg &tau, int64_t N, int k0, int nk,
                     double *out, hipStream_t s) {
    if (nk <= 0) return;
    switch (P) {
        case 1: launch_vi_sigma_p<1>(prec, sld, tau, N, k0, nk, out, s); break;
        case 2: launch_vi_sigma_p<2>(prec, sld, tau, N, k0, nk, out, s); break;
        case 3: launch_vi_sigma_p<3>(prec, sld, tau, N, k0, nk, out, s); break;
        case 4: launch_vi_sigma_p<4>(prec, sld, tau, N, k0, nk, out, s); break;
        case 5: launch_vi_sigma_p<5>(prec, sld, tau, N, k0, nk, out, s); break;
        case 6: launch_vi_sigma_p<6>(prec, sld, tau, N, k0, nk, out, s); break;
        case 7: launch_vi_sigma_p<7>(prec, sld, tau, N, k0, nk, out, s); break;
        case 8: launch_vi_sigma_p<8>(prec, sld, tau, N, k0, nk, out, s); break;
        default: break;
    }
}

// Predicated launches: when the host queues work ahead of a decision that a device kernel takes
// (decide_kernel), every kernel of that work starts by reading the decision's flag and exits if
// it is 0 -- mis-speculated work costs a few microseconds of empty launches and touches nothing.
// The flag pointer is a per-thread launch attribute (set_launch_predicate), nullptr = always run.
static thread_local const int *g_pred = nullptr;
void set_launch_predicate(const int *flag) { g_pred = flag; }
// Buffers of a queued sweep's phase (device-resident sweep, kernels.h): the LD kernels take their
// vector pools, the per-SNP pass all of its state pointers and step sizes from *g_phase.
static thread_local const PhasePtrs *g_phase = nullptr;
void set_launch_phase(const PhasePtrs *pp) { g_phase = pp; }
// The block is written by an EARLIER kernel (the decision) and only read here: through the
// constant address space its fields come in by scalar loads and stay in SGPRs -- read through a
// generic pointer they are vector loads, and every pointer derived from them costs two VGPRs per
// lane (ld_sym_kernel<2>: 108 -> 140 VGPRs, one wave per SIMD less).
typedef const PhasePtrs __attribute__((address_space(4))) *phase_tab;
#define PHASE(pp) ((phase_tab)(pp))
#define PRED_EXIT(pred) do { if ((pred) != nullptr && *(pred) == 0) return; } while (0)

// VILMA_DEBUG_POISON=1 (read by vilma_create): what a beta trial is about to write -- its result slots,
// its candidates' vi_mu -- is filled with NaN first, so that anything a kernel leaves unwritten, or a
// decision reads without its having been produced, surfaces at once as a non-finite objective ("Encountered
// a numerical error.") instead of as whatever the buffer held before.  which = 1 / 2: the vi_mu buffer
// the launch phase assigns to candidate A / B (roles of a sweep queued ahead live on the device).
__global__ __launch_bounds__(256) void poison_kernel(double *p, int64_t n, const int *pred,
                                                      const PhasePtrs *pp, int which) {
    PRED_EXIT(pred);
    if (pp != nullptr && which == 1) p = PHASE(pp)->mu_out;
    if (pp != nullptr && which == 2) p = PHASE(pp)->mu_out2;
    if (p == nullptr) return;
    const double nan = __builtin_nan("");
    for (int64_t o = (int64_t)blockIdx.x * 256 + threadIdx.x; o < n; o += (int64_t)gridDim.x * 256) p[o] = nan;
}
void launch_poison(double *p, int64_t n, int which, hipStream_t s) {
    if (n <= 0) return;
    const int grid = (int)std::min<int64_t>((n + 255) / 256, 1 << 14);
    hipLaunchKernelGGL(poison_kernel, dim3(grid), dim3(256), 0, s, p, n, g_pred, which ? g_phase : nullptr, which);
}

typedef const double __attribute__((address_space(4))) *const_tab;
static __device__ __forceinline__ const_tab as_table(const double *p) { return (const_tab)p; }

struct PoolPair { const double *p[2]; };
struct PoolPairRW { double *p[2]; };

// NR right-hand sides per pass over U (1, or 2 for the two candidates of a beta trial): x of side r
// is read from pool r and its t' written there; every element of U is loaded once for both.
template <bool KEEP, int NR>
__global__ __launch_bounds__(CS_WAVES * 64) void ld_colsum_kernel(
    const LdItem *__restrict__ items, const PoolPairRW pools_arg, const int *pred,
    const PhasePtrs *pp) {
    __shared__ double red[NR][CS_WAVES][128];
    PRED_EXIT(pred);
    PoolPairRW pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const LdItem it = items[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int c = it.col0 + 2 * lane;
    const int rows = it.rows;
    const int64_t ld = it.ld;
    const double *xp[NR];
    double acc0[NR], acc1[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) { xp[r] = pools.p[r] + it.x_off; acc0[r] = 0.0; acc1[r] = 0.0; }
    if (c < it.ncols) {
        const double *ap = it.a + c;
        for (int j = w * CS_ROWS; j < rows; j += CS_WAVES * CS_ROWS) {
            if (j + CS_ROWS <= rows) {
                v2d v[CS_ROWS];
#pragma unroll
                for (int u = 0; u < CS_ROWS; ++u)
                    v[u] = KEEP ? *(gd2_ptr)(ap + (int64_t)(j + u) * ld)
                                : LD_STREAM_LOAD(ap + (int64_t)(j + u) * ld);
#pragma unroll
                for (int r = 0; r < NR; ++r) {
#pragma unroll
                    for (int u = 0; u < CS_ROWS; ++u) {
                        const double xv = xp[r][j + u];
                        acc0[r] = fma(v[u].x, xv, acc0[r]);
                        acc1[r] = fma(v[u].y, xv, acc1[r]);
                    }
                }
            } else {
                for (int jj = j; jj < rows; ++jj) {
                    const v2d v = KEEP ? *(gd2_ptr)(ap + (int64_t)jj * ld)
                                       : LD_STREAM_LOAD(ap + (int64_t)jj * ld);
#pragma unroll
                    for (int r = 0; r < NR; ++r) {
                        const double xv = xp[r][jj];
                        acc0[r] = fma(v.x, xv, acc0[r]);
                        acc1[r] = fma(v.y, xv, acc1[r]);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        red[r][w][2 * lane] = acc0[r];
        red[r][w][2 * lane + 1] = acc1[r];
    }
    __syncthreads();
    if (threadIdx.x < 128) {
        const int col = it.col0 + threadIdx.x;
        if (col < it.ncols) {
            const double scale = it.scale != nullptr ? it.scale[col] : 1.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double s = red[r][0][threadIdx.x];
#pragma unroll
                for (int ww = 1; ww < CS_WAVES; ++ww) s += red[r][ww][threadIdx.x];
                pools.p[r][it.y_off + col] = it.scale != nullptr ? s * scale : s;
            }
        }
    }
}

// pool1 == nullptr: one right-hand side
void launch_ld_colsum(const LdItem *items, int n_items, double *pool0, double *pool1, bool keep,
                      hipStream_t s) {
    if (n_items <= 0) return;
    PoolPairRW pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    const dim3 grid(n_items), block(CS_WAVES * 64);
    if (pool1) {
        if (keep) hipLaunchKernelGGL((ld_colsum_kernel<true, 2>), grid, block, 0, s, items, pp, g_pred, g_phase);
        else hipLaunchKernelGGL((ld_colsum_kernel<false, 2>), grid, block, 0, s, items, pp, g_pred, g_phase);
    } else {
        if (keep) hipLaunchKernelGGL((ld_colsum_kernel<true, 1>), grid, block, 0, s, items, pp, g_pred, g_phase);
        else hipLaunchKernelGGL((ld_colsum_kernel<false, 1>), grid, block, 0, s, items, pp, g_pred, g_phase);
    }
}

// --------------------------------------------------------------------------------------------
// Symmetric dense LD product reading only the lower triangle (by 128-column slabs): for slab J the
// panel P = R[j0.., slab] (rows from the diagonal tile down) is streamed ONCE and used twice,
//   column sums  cs[c] = sum_j P[j][c] x[j]        -> y[slab]  (R symmetric: = R[slab, j0..] x)
//   row sums     rs[j] = sum_c P[j][c] x[slab c]    -> y[j] for the rows below the diagonal tile
// so a block costs ~n^2/2 + 64 n elements instead of n^2.  Row sums need a cross-lane reduction:
// 8 rows at a time with a halving butterfly (4+2+1 exchanges, then 3 full steps = 10 shuffles per
// 8 KiB streamed).  Both kinds of partial go to a scratch S[slab][j]; ld_sym_combine_kernel adds
// them in slab order (fixed order => deterministic) and forms the y.z partials.
// --------------------------------------------------------------------------------------------
// Cross-lane moves without the LDS crossbar.  __shfl_xor compiles to ds_bpermute_b32 (two per
// double, each a round trip through the LDS); gfx950 can do what the butterfly below needs in the
// vector ALU: v_permlane32_swap / v_permlane16_swap exchange halves of two registers, DPP moves
// permute within a row of 16 lanes.  (Semantics checked on the device: profiles/README.md.)
static __device__ __forceinline__ void swap_halves32(double &a, double &b) {
    // a' = {a[0:31], b[0:31]}, b' = {a[32:63], b[32:63]}
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a);
    const unsigned blo = __double2loint(b), bhi = __double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane32_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(ahi, bhi, false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
static __device__ __forceinline__ void swap_rows16(double &a, double &b) {
    // per 32 lanes: a' = {a[0:15], b[0:15]}, b' = {a[16:31], b[16:31]}
    const unsigned alo = __double2loint(a), ahi = __double2hiint(a);
    const unsigned blo = __double2loint(b), bhi = __double2hiint(b);
    const auto lo = __builtin_amdgcn_permlane16_swap(alo, blo, false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(ahi, bhi, false, false);
    a = __hiloint2double(hi[0], lo[0]);
    b = __hiloint2double(hi[1], lo[1]);
}
#define DPP_XOR1 0xB1          // quad_perm [1,0,3,2]
#define DPP_XOR2 0x4E          // quad_perm [2,3,0,1]
#define DPP_HALF_MIRROR 0x141  // lane i <-> 7 - i within 8 lanes
#define DPP_ROR8 0x128         // row_ror:8 = lane ^ 8 within 16 lanes
template <int CTRL>
static __device__ __forceinline__ double dpp_move(double v) {
    const unsigned lo = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2loint(v), CTRL, 0xf, 0xf, false);
    const unsigned hi = __builtin_amdgcn_update_dpp(0u, (unsigned)__double2hiint(v), CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}

// halving butterfly over 8 row partials: after the three exchange steps lane l holds row
// (bit5*4 + bit4*2 + bit3) summed over the 8 lanes that differ from it in bits 5,4,3; three
// plain steps finish bits 2..0.  Returns the total of that row in every lane of its group.
static __device__ __forceinline__ double sym_rowsum8(const double (&p)[CS_ROWS], int lane, int &row) {
    double q[4], r2[2];
#pragma unroll
    for (int u = 0; u < 4; ++u) {           // bit 5: lanes < 32 end up with rows u, the others u + 4
        double a = p[u], b = p[u + 4];
        swap_halves32(a, b);
        q[u] = a + b;
    }
#pragma unroll
    for (int u = 0; u < 2; ++u) {           // bit 4: odd rows of 16 lanes take rows u + 2
        double a = q[u], b = q[u + 2];
        swap_rows16(a, b);
        r2[u] = a + b;
    }
    const bool h3 = lane & 8;               // bit 3: the upper 8 lanes of a row take row + 1
    const double send = h3 ? r2[0] : r2[1];
    const double keep = h3 ? r2[1] : r2[0];
    double t1 = keep + dpp_move<DPP_ROR8>(send);
    t1 += dpp_move<DPP_XOR1>(t1);           // bits 0, 1, then 2 (the quad is uniform by then, so
    t1 += dpp_move<DPP_XOR2>(t1);           // the mirror within 8 lanes pairs the two quads)
    t1 += dpp_move<DPP_HALF_MIRROR>(t1);
    row = ((lane & 32) ? 4 : 0) | ((lane & 16) ? 2 : 0) | (h3 ? 1 : 0);
    return t1;
}

// one group of 8 panel rows BELOW the diagonal tile: every element feeds a column sum (with the
// row's x) and a row sum (with the column's x); row sums go to scratch
template <bool FULL>
static __device__ __forceinline__ void sym_group(const v2d (&v)[CS_ROWS], const double *__restrict__ xrow,
                                                 int r0, int rows, double xs0, double xs1,
                                                 double &acc0, double &acc1, int lane,
                                                 double *__restrict__ srow) {
    double p[CS_ROWS];
#pragma unroll
    for (int u = 0; u < CS_ROWS; ++u) {
        const double xv = xrow[FULL ? r0 + u : min(r0 + u, rows - 1)];
        const double xz = (FULL || r0 + u < rows) ? xv : 0.0;
        acc0 = fma(v[u].x, xz, acc0);
        acc1 = fma(v[u].y, xz, acc1);
        p[u] = fma(v[u].x, xs0, v[u].y * xs1);
    }
    int rsub;
    const double t1 = sym_rowsum8(p, lane, rsub);
    const int rr = r0 + rsub;
#if LD_STAGED
    if ((lane & 7) == 0 && (FULL || rr < rows)) srow[rr] = t1;          // into the LDS stage
#else
    if ((lane & 7) == 0 && (FULL || rr < rows)) LD_PARTIAL_STORE(&srow[rr], t1);
#endif
}

// one group of 8 rows INSIDE the diagonal tile (row r, column c of the tile): only the lower
// triangle is used -- c <= r feeds the column sum, c < r the row sum -- so for these rows only
// the 128-byte lines up to the diagonal are loaded at all (lanes past them are masked off):
// 56 % of a 128 x 128 tile.  Row sums of the tile go to LDS (they belong to the same output
// entries as the tile's column sums).
static __device__ __forceinline__ void sym_load_diag(v2d (&v)[CS_ROWS], const double *__restrict__ rp,
                                                     int64_t ld, int r0, int rows, int lane) {
    // columns needed by rows r0..r0+7: c <= r0+7, rounded up to whole 16-double lines
    const int lim = 8 * ((r0 + CS_ROWS - 1) / 16 + 1);           // lanes (2 columns each)
    if (lane < lim) {
#pragma unroll
        for (int u = 0; u < CS_ROWS; ++u)
            v[u] = LD_STREAM_LOAD(rp + (int64_t)min(u, rows - 1 - r0) * ld);
    } else {
#pragma unroll
        for (int u = 0; u < CS_ROWS; ++u) v[u] = v2d{0.0, 0.0};
    }
}
static __device__ __forceinline__ void sym_group_diag(const v2d (&v)[CS_ROWS],
                                                      const double *__restrict__ xrow, int r0,
                                                      int rows, int cl, double xs0, double xs1,
                                                      double &acc0, double &acc1, int lane,
                                                      double *__restrict__ rs_lds) {
    double p[CS_ROWS];
#pragma unroll
    for (int u = 0; u < CS_ROWS; ++u) {
        const int r = r0 + u;
        const double xv = xrow[min(r, rows - 1)];
        const double xz = r < rows ? xv : 0.0;
        const double a0 = cl <= r ? v[u].x : 0.0, a1 = cl + 1 <= r ? v[u].y : 0.0;   // c <= r
        acc0 = fma(a0, xz, acc0);
        acc1 = fma(a1, xz, acc1);
        const double b0 = cl < r ? v[u].x : 0.0, b1 = cl + 1 < r ? v[u].y : 0.0;     // c <  r
        p[u] = fma(b0, xs0, b1 * xs1);
    }
    int rsub;
    const double t1 = sym_rowsum8(p, lane, rsub);
    const int rr = r0 + rsub;
    if ((lane & 7) == 0 && rr < rows) rs_lds[rr] = t1;
}

// NR right-hand sides per pass over the panel (1, or 2 for the two candidates of a beta trial):
// every element is loaded once and used for each of them -- the kernel is HBM-bound with the
// vector ALUs a few percent busy, so the second product is free.  Right-hand side r reads its x
// from pool r and writes its partials at scratch + r * s_stride; each goes through exactly the
// arithmetic of the NR = 1 kernel.
#ifndef SYM_PIPELINE
#define SYM_PIPELINE 1
#endif
// -DLD_TRACE=1 (profiles/ld_levels_probe.py builds that variant next to the library): every
// workgroup of ld_sym_kernel leaves {start, end, XCC id, bytes} in a trace buffer
#ifndef LD_TRACE
#define LD_TRACE 0
#endif
// -DSNP_TRACE=1 (profiles/snp_pass_timeline.py): every WAVE of snp_pass_kernel leaves a row of 12 in
// the same buffer: seven core-clock stamps (entry, loop start, loop end, normaliser known, tile
// sums written, hand-over done, exit), the XCC id, entry and exit on the 100 MHz clock the whole
// chip shares, and HW_ID (which CU, SIMD and wave slot it ran in)
#ifndef SNP_TRACE
#define SNP_TRACE 0
#endif

#if LD_TRACE || SNP_TRACE
__device__ double *g_ld_trace = nullptr;
__device__ long long g_ld_trace_cap = 0;
int set_ld_trace(double *buf_dev, int64_t capacity_rows) {
    long long cap = capacity_rows;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ld_trace), &buf_dev, sizeof(buf_dev)) != hipSuccess) return 2;
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_ld_trace_cap), &cap, sizeof(cap)) != hipSuccess) return 2;
    return 0;
}
#else
int set_ld_trace(double *, int64_t) { return 1; }
#endif
template <int NR>
__global__ __launch_bounds__(CS_WAVES * 64) void ld_sym_kernel(
    const SymItem *__restrict__ items, const PoolPair pools_arg, double *__restrict__ scratch,
    int64_t s_stride, const int *pred, const PhasePtrs *pp) {
    __shared__ double red[NR][CS_WAVES][128];
    __shared__ double rs_diag[NR][128];
#if LD_STAGED
    __shared__ double rs_stage[NR][LD_MAX_CHUNK_ROWS];
#endif
    PRED_EXIT(pred);
#if LD_TRACE
    const unsigned long long trace_t0 = __builtin_amdgcn_s_memrealtime();
    const unsigned long long trace_c0 = __builtin_amdgcn_s_memtime();
#endif
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const SymItem it = items[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cl = 2 * lane;
    const bool active = cl < it.w;
    const int rows = it.rows;
    const int64_t ld = it.ld;
    const bool has_diag = it.r0 == 0;                                // chunk 0 holds the diagonal tile
    const double *xrow[NR];
    double *srow[NR];
    double xs0[NR], xs1[NR], acc0[NR], acc1[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double *xcol = pools.p[r] + it.x_off + it.j0;          // x of the slab's columns
        xrow[r] = xcol + it.r0;                                      // x of this chunk's rows
        // x of this lane's two columns (the slab's columns are rows j0.. of the same vector)
        xs0[r] = active ? xcol[cl] : 0.0;
        xs1[r] = (cl + 1 < it.w) ? xcol[cl + 1] : 0.0;
#if LD_STAGED
        srow[r] = rs_stage[r];          // this chunk's row sums, written out whole at the end
#else
        srow[r] = scratch + r * s_stride + it.s_off + it.j0 + it.r0;
#endif
        acc0[r] = 0.0;
        acc1[r] = 0.0;
    }
    // loads are unconditional within a group (a select around a load makes hipcc branch and wait
    // per element): lanes beyond the slab read column 0 and are neutralised by xs = 0
    const double *ap = it.a + (active ? cl : 0);
    const int ngroups = (rows + CS_ROWS - 1) / CS_ROWS;     // group g belongs to wave g % 4
    const int ndiag = has_diag ? (it.w + CS_ROWS - 1) / CS_ROWS : 0;   // groups inside the diagonal tile
    const int nfull = rows / CS_ROWS;
    const int64_t gstride = (int64_t)CS_WAVES * CS_ROWS * ld;
    const double *rp = ap + (int64_t)w * CS_ROWS * ld;
    int g = w;
    v2d v[CS_ROWS];
    // (1) the diagonal tile: lower triangle only
    for (; g < ndiag && g < ngroups; g += CS_WAVES, rp += gstride) {
        sym_load_diag(v, rp, ld, g * CS_ROWS, rows, lane);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            sym_group_diag(v, xrow[r], g * CS_ROWS, rows, active ? cl : 2 * 64, xs0[r], xs1[r],
                           acc0[r], acc1[r], lane, rs_diag[r]);
    }
    // (2) the rows below it.  One right-hand side: plain loop, 8 x 1 KiB loads in flight per wave,
    // latency hidden by occupancy.  Two: the loop is unrolled by two with the next group's loads
    // written ahead of the current group's arithmetic -- but they sit in a wave-uniform `if`, so the
    // compiler cannot count them and waits for vmcnt(0) before the arithmetic: in effect the plain
    // loop (SYM_PIPELINE=0 measures the same).  A TRUE double buffer (unconditional, clamped
    // prefetch: 16 KiB in flight per wave) was measured in round 3 and is 2 % SLOWER, for one and
    // for two right-hand sides (gpurun_out/ab36.txt): this kernel does not want more in flight.
    if (SYM_PIPELINE && NR == 2) {
        v2d vb[CS_ROWS];
        if (g < nfull) {
#pragma unroll
            for (int u = 0; u < CS_ROWS; ++u) v[u] = LD_STREAM_LOAD(rp + (int64_t)u * ld);
        }
        while (g < nfull) {
            const int g2 = g + CS_WAVES;
            if (g2 < nfull) {
#pragma unroll
                for (int u = 0; u < CS_ROWS; ++u) vb[u] = LD_STREAM_LOAD(rp + gstride + (int64_t)u * ld);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
                sym_group<true>(v, xrow[r], g * CS_ROWS, rows, xs0[r], xs1[r], acc0[r], acc1[r],
                                lane, srow[r]);
            if (g2 >= nfull) { g = g2; rp += gstride; break; }
            const int g3 = g2 + CS_WAVES;
            if (g3 < nfull) {
#pragma unroll
                for (int u = 0; u < CS_ROWS; ++u) v[u] = LD_STREAM_LOAD(rp + 2 * gstride + (int64_t)u * ld);
            }
#pragma unroll
            for (int r = 0; r < NR; ++r)
                sym_group<true>(vb, xrow[r], g2 * CS_ROWS, rows, xs0[r], xs1[r], acc0[r], acc1[r],
                                lane, srow[r]);
            g = g3;
            rp += 2 * gstride;
        }
    } else {
        for (; g < nfull; g += CS_WAVES, rp += gstride) {
#pragma unroll
            for (int u = 0; u < CS_ROWS; ++u) v[u] = LD_STREAM_LOAD(rp + (int64_t)u * ld);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                sym_group<true>(v, xrow[r], g * CS_ROWS, rows, xs0[r], xs1[r], acc0[r], acc1[r],
                                lane, srow[r]);
        }
    }
    if (g == nfull && g < ngroups) {                        // the one partial group, below the tile
        const int r0 = nfull * CS_ROWS;
#pragma unroll
        for (int u = 0; u < CS_ROWS; ++u)     // rows past the end re-read the last row
            v[u] = LD_STREAM_LOAD(ap + (int64_t)min(r0 + u, rows - 1) * ld);
#pragma unroll
        for (int r = 0; r < NR; ++r)
            sym_group<false>(v, xrow[r], r0, rows, xs0[r], xs1[r], acc0[r], acc1[r], lane, srow[r]);
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        red[r][w][2 * lane] = acc0[r];
        red[r][w][2 * lane + 1] = acc1[r];
    }
    __syncthreads();
#if LD_STAGED
    // the rows below the diagonal tile (its own row sums went into rs_diag), two per store
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        double *dst = scratch + r * s_stride + it.s_off + it.j0 + it.r0;
        int t = ndiag * CS_ROWS + 2 * (int)threadIdx.x;
        for (; t + 1 < rows; t += 2 * CS_WAVES * 64) LD_PARTIAL_STORE2(&dst[t], rs_stage[r][t], rs_stage[r][t + 1]);
        if (t + 1 == rows) LD_PARTIAL_STORE(&dst[t], rs_stage[r][t]);
    }
#endif
    // this chunk's share of the slab's own entries: its column sums (+ the diagonal tile's row
    // sums, which belong to the same entries), two columns per thread
    const int c0 = 2 * (int)threadIdx.x;
    if (c0 < it.w) {
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double s0 = red[r][0][c0], s1 = red[r][0][c0 + 1];
#pragma unroll
            for (int ww = 1; ww < CS_WAVES; ++ww) { s0 += red[r][ww][c0]; s1 += red[r][ww][c0 + 1]; }
            if (has_diag) { s0 += rs_diag[r][c0]; s1 += rs_diag[r][c0 + 1]; }
            double *dst = scratch + r * s_stride + it.c_off + c0;
            if (c0 + 1 < it.w) LD_PARTIAL_STORE2(dst, s0, s1);
            else LD_PARTIAL_STORE(dst, s0);
        }
    }
#if LD_TRACE
    if (threadIdx.x == 0 && g_ld_trace != nullptr && (long long)blockIdx.x < g_ld_trace_cap) {
        double *row = g_ld_trace + 5 * (long long)blockIdx.x;
        row[0] = (double)trace_t0;
        row[1] = (double)__builtin_amdgcn_s_memrealtime();
        row[2] = (double)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xf);    // HW_REG_XCC_ID[3:0]
        row[3] = 8.0 * (double)it.rows * (double)it.ld;
        row[4] = (double)(__builtin_amdgcn_s_memtime() - trace_c0);      // core-clock cycles
    }
#endif
}

// y[j] for 256 columns of one block per workgroup: the row sums S[J][j] of the slabs to the left
// (J < slab(j), slab order) then the column-sum chunks of j's own slab (chunk order) -- a fixed
// order -- with the chunk's y.z partial.  The slab index is wave-uniform; eight loads in flight.
// blockIdx.y = right-hand side (its own pool, scratch and partials behind the first one's).
// Partials are read in groups of COMB_GROUP independent loads (indices clamped, extras masked: no
// branch around a load).  A typical block has 2 - 5 slabs and 1 - 2 chunks per slab, so groups of 8
// issued mostly duplicates: 4 takes the C3 combine from 38 to 31 us; the order of the additions is
// the same for any group size.
#ifndef COMB_GROUP
#define COMB_GROUP 4
#endif
__global__ __launch_bounds__(256) void ld_sym_combine_kernel(
    const SymCombItem *__restrict__ items, const PoolPairRW pools_arg,
    const double *__restrict__ scratch0, int64_t s_stride, double *__restrict__ dot_partials0,
    int dot_stride, const int *pred, const PhasePtrs *pp) {
    __shared__ double dred[4];
    PRED_EXIT(pred);
    const int rhs = blockIdx.y;
    // (selected with a conditional: indexing a local copy of the pair would put it in scratch)
    double *const pool_r = pp != nullptr ? (rhs == 0 ? PHASE(pp)->pool_out : PHASE(pp)->pool_out2)
                                         : (rhs == 0 ? pools_arg.p[0] : pools_arg.p[1]);
    const double *__restrict__ xpool = pool_r;
    double *__restrict__ ypool = pool_r;
    const double *__restrict__ scratch = scratch0 + rhs * s_stride;
    double *__restrict__ dot_partials = dot_partials0 + (int64_t)rhs * dot_stride;
    const SymCombItem it = items[blockIdx.x];
    const int j = it.j0 + threadIdx.x;
    const bool live = j < it.n;
    const int jj = live ? j : it.n - 1;
    const int slab = jj >> 7;
    const int sn = (it.n + 1) & ~1;                 // rows of S are pad2(n) apart
    const double *sj = scratch + it.s_base + jj;
    const double xj = xpool[it.dot_off + jj];
    double s = 0.0;
    for (int J = 0; J < slab; J += COMB_GROUP) {
        double t[COMB_GROUP];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) t[u] = sj[(int64_t)min(J + u, slab - 1) * sn];   // no branch
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) s += (J + u < slab) ? t[u] : 0.0;
    }
    const int nch = (it.n - 128 * slab + it.chunk_rows - 1) / it.chunk_rows;
    const double *cj = scratch + it.c_base + (int64_t)slab * it.nch_max * 128 + (jj & 127);
    for (int c = 0; c < nch; c += COMB_GROUP) {
        double t[COMB_GROUP];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) t[u] = cj[(int64_t)min(c + u, nch - 1) * 128];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) s += (c + u < nch) ? t[u] : 0.0;
    }
    if (live) ypool[it.y_off + j] = s;
    double dv = wave_sum(live ? s * xj : 0.0);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = dv;
    __syncthreads();
    if (threadIdx.x == 0) dot_partials[it.dot_slot] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

// pool1 == nullptr: one right-hand side; else two (pool0 and pool1 hold the two candidates' x)
void launch_ld_sym(const SymItem *items, int n_items, const double *pool0, const double *pool1,
                   double *scratch, int64_t s_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    if (pool1)
        hipLaunchKernelGGL(ld_sym_kernel<2>, dim3(n_items), dim3(CS_WAVES * 64), 0, s, items, pp,
                           scratch, s_stride, g_pred, g_phase);
    else
        hipLaunchKernelGGL(ld_sym_kernel<1>, dim3(n_items), dim3(CS_WAVES * 64), 0, s, items, pp,
                           scratch, s_stride, g_pred, g_phase);
}

void launch_ld_sym_combine(const SymCombItem *items, int n_items, double *pool0, double *pool1,
                           const double *scratch, int64_t s_stride, double *dot_partials,
                           int dot_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPairRW pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    hipLaunchKernelGGL(ld_sym_combine_kernel, dim3(n_items, pool1 ? 2 : 1), dim3(256), 0, s, items,
                       pp, scratch, s_stride, dot_partials, dot_stride, g_pred, g_phase);
}

// --------------------------------------------------------------------------------------------
// Tiled symmetric product (round 5): ONE workgroup takes the rows [R0, R1) of up to LD_TILE_MAX_SLABS
// neighbouring 128-column slabs of a block -- what ld_sym_kernel gives to as many workgroups -- and
// stores its partial sums once, merged:
//   * a row of the block is handled by the same lane of the same wave in every slab (panel rows
//     start a multiple of 128 below each other and groups of 8 rows go round the 4 waves), so the
//     row sums of the slabs are ADDED in LDS without any synchronisation, in slab order;
//   * each wave keeps a slab's column sums in registers and parks them in LDS when the slab ends;
//     no barrier between slabs, one at the end;
//   * an item on the diagonal (its columns are among its rows) writes rows and columns as ONE run
//     y_part[j] = row sum (strictly lower part) + column sum: the two belong to the same entries.
// What leaves per item: (rows + columns) doubles instead of (slabs x rows + slabs x 128 per chunk).
// Scratch: S[slot][pad2(n)] per block; entry j (column strip k = j / CW, row strip g = j / TR, G row
// strips) receives its row partials of strips 0 .. k in slots 0 .. k and its column partials of the
// row strips g + 1 .. G - 1 in slots k + 1 .. k + G - 1 - g; ld_tile_combine_kernel adds slots
// 0 .. k + G - g - 1 in order.  Fixed order throughout: results do not depend on scheduling.
// --------------------------------------------------------------------------------------------
#define LD_TILE_MAX_SLABS 4
#define LD_TILE_MAX_ROWS 512

template <bool FULL>
static __device__ __forceinline__ void tile_group(const v2d (&v)[CS_ROWS], const double *__restrict__ xrow,
                                                  int r0, int rows, double xs0, double xs1,
                                                  double &acc0, double &acc1, int lane,
                                                  double *__restrict__ srow, bool first) {
    double p[CS_ROWS];
#pragma unroll
    for (int u = 0; u < CS_ROWS; ++u) {
        const double xv = xrow[FULL ? r0 + u : min(r0 + u, rows - 1)];
        const double xz = (FULL || r0 + u < rows) ? xv : 0.0;
        acc0 = fma(v[u].x, xz, acc0);
        acc1 = fma(v[u].y, xz, acc1);
        p[u] = fma(v[u].x, xs0, v[u].y * xs1);
    }
    int rsub;
    const double t1 = sym_rowsum8(p, lane, rsub);
    const int rr = r0 + rsub;
    if ((lane & 7) == 0 && (FULL || rr < rows)) srow[rr] = first ? t1 : srow[rr] + t1;
}
static __device__ __forceinline__ void tile_group_diag(const v2d (&v)[CS_ROWS],
                                                       const double *__restrict__ xrow, int r0,
                                                       int rows, int cl, double xs0, double xs1,
                                                       double &acc0, double &acc1, int lane,
                                                       double *__restrict__ srow, bool first) {
    double p[CS_ROWS];
#pragma unroll
    for (int u = 0; u < CS_ROWS; ++u) {
        const int r = r0 + u;
        const double xv = xrow[min(r, rows - 1)];
        const double xz = r < rows ? xv : 0.0;
        const double a0 = cl <= r ? v[u].x : 0.0, a1 = cl + 1 <= r ? v[u].y : 0.0;   // c <= r
        acc0 = fma(a0, xz, acc0);
        acc1 = fma(a1, xz, acc1);
        const double b0 = cl < r ? v[u].x : 0.0, b1 = cl + 1 < r ? v[u].y : 0.0;     // c <  r
        p[u] = fma(b0, xs0, b1 * xs1);
    }
    int rsub;
    const double t1 = sym_rowsum8(p, lane, rsub);
    const int rr = r0 + rsub;
    if ((lane & 7) == 0 && rr < rows) srow[rr] = first ? t1 : srow[rr] + t1;
}

// element offset of panel J inside a block's store: panels 0 .. J-1 are full slabs (ld = 128)
static __device__ __forceinline__ int64_t sym_panel_off(int n, int J) {
    return 128 * ((int64_t)J * n - 64 * (int64_t)J * (J - 1));
}

template <int NR>
__global__ __launch_bounds__(CS_WAVES * 64) void ld_tile_kernel(
    const SymTile *__restrict__ items, const PoolPair pools_arg, double *__restrict__ scratch,
    int64_t s_stride, PoolPairRW ypools_arg, double *__restrict__ dot_partials, int dot_stride,
    int max_slabs, const int *pred, const PhasePtrs *pp) {
    // rs[NR][LD_TILE_MAX_ROWS] then red[NR][max_slabs][CS_WAVES][128]
    extern __shared__ double tile_lds[];
    PRED_EXIT(pred);
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const SymTile it = items[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cl = 2 * lane;
    const int n = it.n;
    const bool diag = it.col_off < 0;                 // the item's columns are among its rows
    double *const red = tile_lds + NR * LD_TILE_MAX_ROWS;
    const double *xblk[NR];
    double xs0n[NR], xs1n[NR], acc0[NR], acc1[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        xblk[r] = pools.p[r] + it.x_off;
        // x of this lane's two columns in the first slab (lanes past the block's end read its last entry)
        const int c = 128 * it.J0 + cl;
        xs0n[r] = xblk[r][min(c, n - 1)];
        xs1n[r] = xblk[r][min(c + 1, n - 1)];
    }
    for (int Jl = 0; Jl < it.nJ; ++Jl) {
        const int J = it.J0 + Jl, j0 = 128 * J;
        const int wJ = min(128, n - j0);
        const int64_t ld = (wJ + 15) & ~15;
        const bool active = cl < wJ;
        const bool first = Jl == 0;
        const int rstart = diag ? j0 : it.R0;         // block row of the first panel row taken
        const int rows = it.R1 - rstart;
        double xs0[NR], xs1[NR];
        const double *xrow[NR];
        double *srow[NR];
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            xs0[r] = active ? xs0n[r] : 0.0;
            xs1[r] = (cl + 1 < wJ) ? xs1n[r] : 0.0;
            // the next slab's x, a slab ahead of its use
            const int c = j0 + 128 + cl;
            xs0n[r] = xblk[r][min(c, n - 1)];
            xs1n[r] = xblk[r][min(c + 1, n - 1)];
            xrow[r] = xblk[r] + rstart;
            srow[r] = tile_lds + r * LD_TILE_MAX_ROWS + (rstart - it.out0);
            acc0[r] = 0.0;
            acc1[r] = 0.0;
        }
        const double *ap = it.a + sym_panel_off(n, J) + (int64_t)(rstart - j0) * ld + (active ? cl : 0);
        const int ngroups = (rows + CS_ROWS - 1) / CS_ROWS;     // group g belongs to wave g % 4
        const int ndiag = diag ? (wJ + CS_ROWS - 1) / CS_ROWS : 0;   // groups inside the diagonal tile
        const int nfull = rows / CS_ROWS;
        const int64_t gstride = (int64_t)CS_WAVES * CS_ROWS * ld;
        const double *rp = ap + (int64_t)w * CS_ROWS * ld;
        int g = w;
        v2d v[CS_ROWS];
        for (; g < ndiag && g < ngroups; g += CS_WAVES, rp += gstride) {
            sym_load_diag(v, rp, ld, g * CS_ROWS, rows, lane);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                tile_group_diag(v, xrow[r], g * CS_ROWS, rows, active ? cl : 2 * 64, xs0[r], xs1[r],
                                acc0[r], acc1[r], lane, srow[r], first);
        }
        for (; g < nfull; g += CS_WAVES, rp += gstride) {
#pragma unroll
            for (int u = 0; u < CS_ROWS; ++u) v[u] = LD_STREAM_LOAD(rp + (int64_t)u * ld);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                tile_group<true>(v, xrow[r], g * CS_ROWS, rows, xs0[r], xs1[r], acc0[r], acc1[r],
                                 lane, srow[r], first);
        }
        if (g == nfull && g < ngroups) {                        // the one partial group, below the tile
            const int r0 = nfull * CS_ROWS;
#pragma unroll
            for (int u = 0; u < CS_ROWS; ++u)     // rows past the end re-read the last row
                v[u] = LD_STREAM_LOAD(ap + (int64_t)min(r0 + u, rows - 1) * ld);
#pragma unroll
            for (int r = 0; r < NR; ++r)
                tile_group<false>(v, xrow[r], r0, rows, xs0[r], xs1[r], acc0[r], acc1[r], lane,
                                  srow[r], first);
        }
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double *rd = red + (((int64_t)r * max_slabs + Jl) * CS_WAVES + w) * 128;
            *(v2d *)(rd + cl) = v2d{acc0[r], acc1[r]};
        }
    }
    __syncthreads();
    // one store phase: two entries per thread and pass
    const int ncols = min(128 * it.nJ, n - 128 * it.J0);
    const int nrows = it.R1 - it.out0;
    const int t = 2 * (int)threadIdx.x;
    double ydot[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double *rs = tile_lds + r * LD_TILE_MAX_ROWS;
        const double *rd = red + (int64_t)r * max_slabs * CS_WAVES * 128;
        double c0 = 0.0, c1 = 0.0;
        if (t < ncols) {
            const double *q = rd + (int64_t)(t >> 7) * CS_WAVES * 128 + (t & 127);
            c0 = q[0]; c1 = q[1];
#pragma unroll
            for (int ww = 1; ww < CS_WAVES; ++ww) { c0 += q[ww * 128]; c1 += q[ww * 128 + 1]; }
        }
        ydot[r] = 0.0;
        if (it.direct) {
            // the item is the whole block: these ARE y's entries (row_off is y's pool offset), and
            // the block's share of y.z is formed here -- no scratch, no combine item
            double *yp = (pp != nullptr ? (r == 0 ? PHASE(pp)->pool_out : PHASE(pp)->pool_out2)
                                        : (r == 0 ? ypools_arg.p[0] : ypools_arg.p[1])) + it.row_off;
            if (t < nrows) {
                const double y0 = rs[t] + c0;
                yp[t] = y0;
                ydot[r] = y0 * xblk[r][t];
            }
            if (t + 1 < nrows) {
                const double y1 = rs[t + 1] + c1;
                yp[t + 1] = y1;
                ydot[r] = fma(y1, xblk[r][t + 1], ydot[r]);
            }
            continue;
        }
        double *dst = scratch + r * s_stride + it.row_off;
        if (diag) {
            if (t + 1 < nrows) LD_PARTIAL_STORE2(&dst[t], rs[t] + c0, rs[t + 1] + c1);
            else if (t < nrows) LD_PARTIAL_STORE(&dst[t], rs[t] + c0);
        } else {
            if (t + 1 < nrows) LD_PARTIAL_STORE2(&dst[t], rs[t], rs[t + 1]);
            else if (t < nrows) LD_PARTIAL_STORE(&dst[t], rs[t]);
            double *cd = scratch + r * s_stride + it.col_off;
            if (t + 1 < ncols) LD_PARTIAL_STORE2(&cd[t], c0, c1);
            else if (t < ncols) LD_PARTIAL_STORE(&cd[t], c0);
        }
    }
    if (it.direct) {                    // (uniform over the workgroup)
        double *dred = red;             // the column sums have been consumed by their own threads only
        __syncthreads();
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            const double dv = wave_sum(ydot[r]);
            if (lane == 0) dred[r * CS_WAVES + w] = dv;
        }
        __syncthreads();
        if (threadIdx.x < NR) {
            const double *d = dred + threadIdx.x * CS_WAVES;
            dot_partials[(int64_t)threadIdx.x * dot_stride + it.direct - 1] = (d[0] + d[1]) + (d[2] + d[3]);
        }
    }
}

// y[j] = S[0][j] + ... + S[nt - 1][j], nt = j / CW + G - j / TR, for 256 entries of one block, with the
// chunk's y.z partial.  blockIdx.y = right-hand side.
__global__ __launch_bounds__(256) void ld_tile_combine_kernel(
    const TileCombItem *__restrict__ items, const PoolPairRW pools_arg,
    const double *__restrict__ scratch0, int64_t s_stride, double *__restrict__ dot_partials0,
    int dot_stride, const int *pred, const PhasePtrs *pp) {
    __shared__ double dred[4];
    PRED_EXIT(pred);
    const int rhs = blockIdx.y;
    double *const pool_r = pp != nullptr ? (rhs == 0 ? PHASE(pp)->pool_out : PHASE(pp)->pool_out2)
                                         : (rhs == 0 ? pools_arg.p[0] : pools_arg.p[1]);
    const double *__restrict__ xpool = pool_r;
    double *__restrict__ ypool = pool_r;
    const double *__restrict__ scratch = scratch0 + rhs * s_stride;
    double *__restrict__ dot_partials = dot_partials0 + (int64_t)rhs * dot_stride;
    const TileCombItem it = items[blockIdx.x];
    const int j = it.j0 + threadIdx.x;
    const bool live = j < it.n;
    const int jj = live ? j : it.n - 1;
    const int sn = (it.n + 1) & ~1;
    const int nt = jj / it.cw + it.G - jj / it.tr;
    const double *sj = scratch + it.s_base + jj;
    const double xj = xpool[it.dot_off + jj];
    double s = 0.0;
    for (int T = 0; T < nt; T += COMB_GROUP) {
        double t[COMB_GROUP];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) t[u] = sj[(int64_t)min(T + u, nt - 1) * sn];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) s += (T + u < nt) ? t[u] : 0.0;
    }
    if (live) ypool[it.y_off + j] = s;
    double dv = wave_sum(live ? s * xj : 0.0);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = dv;
    __syncthreads();
    if (threadIdx.x == 0) dot_partials[it.dot_slot] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

void launch_ld_tile(const SymTile *items, int n_items, int max_slabs, double *pool0, double *pool1,
                    double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                    hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    PoolPairRW yp;
    yp.p[0] = pool0;
    yp.p[1] = pool1 ? pool1 : pool0;
    const int nr = pool1 ? 2 : 1;
    const size_t lds = sizeof(double) * nr * (LD_TILE_MAX_ROWS + (size_t)max_slabs * CS_WAVES * 128);
    if (pool1)
        hipLaunchKernelGGL(ld_tile_kernel<2>, dim3(n_items), dim3(CS_WAVES * 64), lds, s, items, pp,
                           scratch, s_stride, yp, dot_partials, dot_stride, max_slabs, g_pred, g_phase);
    else
        hipLaunchKernelGGL(ld_tile_kernel<1>, dim3(n_items), dim3(CS_WAVES * 64), lds, s, items, pp,
                           scratch, s_stride, yp, dot_partials, dot_stride, max_slabs, g_pred, g_phase);
}

void launch_ld_tile_combine(const TileCombItem *items, int n_items, double *pool0, double *pool1,
                            const double *scratch, int64_t s_stride, double *dot_partials,
                            int dot_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPairRW pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    hipLaunchKernelGGL(ld_tile_combine_kernel, dim3(n_items, pool1 ? 2 : 1), dim3(256), 0, s, items,
                       pp, scratch, s_stride, dot_partials, dot_stride, g_pred, g_phase);
}

// --------------------------------------------------------------------------------------------
// Second pass of the eigen form on the same U: y = U t'.  Lanes along the (up to 128) columns of a
// slab, 4 waves x interleaved 8-row groups, the row sums of each group by the halving butterfly
// of the symmetric kernel; partial row sums per column slab go to scratch, the combine kernel
// adds the slabs in order and forms the y.z partial.  Loads are non-temporal (last use).
// --------------------------------------------------------------------------------------------
template <int NR>
__global__ __launch_bounds__(CS_WAVES * 64) void ld_rowsum_kernel(
    const RowItem *__restrict__ items, const PoolPair pools_arg, double *__restrict__ scratch,
    int64_t s_stride, const int *pred, const PhasePtrs *pp) {
    PRED_EXIT(pred);
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const RowItem it = items[blockIdx.x];
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int cl = 2 * lane;
    const bool active = cl < it.w;
    const int rows = it.rows;
    const int64_t ld = it.ld;
    double ts0[NR], ts1[NR];
    double *srow[NR];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double *tp = pools.p[r] + it.t_off;
        ts0[r] = active ? tp[cl] : 0.0;
        ts1[r] = (cl + 1 < it.w) ? tp[cl + 1] : 0.0;
        srow[r] = scratch + r * s_stride + it.s_off;
    }
    const double *ap = it.a + (active ? cl : 0);          // idle lanes re-read column 0, times 0
    const int ngroups = (rows + CS_ROWS - 1) / CS_ROWS;
    for (int g = w; g < ngroups; g += CS_WAVES) {
        const int r0 = g * CS_ROWS;
        v2d v[CS_ROWS];
#pragma unroll
        for (int u = 0; u < CS_ROWS; ++u)
            v[u] = LD_STREAM_LOAD(ap + (int64_t)min(r0 + u, rows - 1) * ld);
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double p[CS_ROWS];
#pragma unroll
            for (int u = 0; u < CS_ROWS; ++u) p[u] = fma(v[u].x, ts0[r], v[u].y * ts1[r]);
            int rsub;
            const double t1 = sym_rowsum8(p, lane, rsub);
            const int rr = r0 + rsub;
            if ((lane & 7) == 0 && rr < rows) srow[r][rr] = t1;
        }
    }
}

// blockIdx.y = right-hand side
__global__ __launch_bounds__(256) void ld_rowsum_combine_kernel(
    const RowCombItem *__restrict__ items, const PoolPairRW pools_arg,
    const double *__restrict__ scratch0, int64_t s_stride, double *__restrict__ dot_partials0,
    int dot_stride, const int *pred, const PhasePtrs *pp) {
    __shared__ double dred[4];
    PRED_EXIT(pred);
    const int rhs = blockIdx.y;
    // (selected with a conditional: indexing a local copy of the pair would put it in scratch)
    double *const pool_r = pp != nullptr ? (rhs == 0 ? PHASE(pp)->pool_out : PHASE(pp)->pool_out2)
                                         : (rhs == 0 ? pools_arg.p[0] : pools_arg.p[1]);
    const double *__restrict__ xpool = pool_r;
    double *__restrict__ ypool = pool_r;
    const double *__restrict__ scratch = scratch0 + rhs * s_stride;
    double *__restrict__ dot_partials = dot_partials0 + (int64_t)rhs * dot_stride;
    const RowCombItem it = items[blockIdx.x];
    const int i = it.i0 + threadIdx.x;
    const bool live = i < it.n;
    const int ii = live ? i : it.n - 1;
    const int sn = (it.n + 1) & ~1;                 // rows of S are pad2(n) apart
    const double *si = scratch + it.s_base + ii;
    const double xi = xpool[it.dot_off + ii];
    double s = 0.0;
    for (int J = 0; J < it.ns; J += COMB_GROUP) {
        double t[COMB_GROUP];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) t[u] = si[(int64_t)min(J + u, it.ns - 1) * sn];
#pragma unroll
        for (int u = 0; u < COMB_GROUP; ++u) s += (J + u < it.ns) ? t[u] : 0.0;
    }
    if (live) ypool[it.y_off + i] = s;
    double dv = wave_sum(live ? s * xi : 0.0);
    if ((threadIdx.x & 63) == 0) dred[threadIdx.x >> 6] = dv;
    __syncthreads();
    if (threadIdx.x == 0) dot_partials[it.dot_slot] = (dred[0] + dred[1]) + (dred[2] + dred[3]);
}

void launch_ld_rowsum(const RowItem *items, int n_items, const double *pool0, const double *pool1,
                      double *scratch, int64_t s_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    if (pool1)
        hipLaunchKernelGGL(ld_rowsum_kernel<2>, dim3(n_items), dim3(CS_WAVES * 64), 0, s, items, pp,
                           scratch, s_stride, g_pred, g_phase);
    else
        hipLaunchKernelGGL(ld_rowsum_kernel<1>, dim3(n_items), dim3(CS_WAVES * 64), 0, s, items, pp,
                           scratch, s_stride, g_pred, g_phase);
}

void launch_ld_rowsum_combine(const RowCombItem *items, int n_items, double *pool0, double *pool1,
                              const double *scratch, int64_t s_stride, double *dot_partials,
                              int dot_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPairRW pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    hipLaunchKernelGGL(ld_rowsum_combine_kernel, dim3(n_items, pool1 ? 2 : 1), dim3(256), 0, s,
                       items, pp, scratch, s_stride, dot_partials, dot_stride, g_pred, g_phase);
}

// --------------------------------------------------------------------------------------------
// Fused eigen-form product: y_partial = U_slab (s * (U_slab^T x)) with U read from HBM ONCE and
// held in REGISTERS between the two uses.  U is stored column-major (column stride ldc = n rounded
// up to even, the pad row zero).  One workgroup of 256 threads owns a slab of columns of one
// block; thread t owns rows 512 i + 2 t, + 1 (i < R / 2): its x and its partial y live in
// registers for the whole slab.  Columns go in batches of C:
//   1. the batch is loaded (16 B per lane, 1 KiB per wave-load, consecutive rows of one column);
//      the NEXT batch's loads are issued before this one is used, so the stream never drains;
//   2. each thread's partial dot products with x (C values per right-hand side) are summed over
//      the wave by the halving butterfly of the symmetric kernel (8 values per butterfly, DPP /
//      permlane moves only) and over the 4 waves through LDS (double-buffered: one barrier per
//      batch); scaled by the eigenvalues they are t'[c];
//   3. y += U[:, c] t'[c] from the registers that still hold the batch.
// The partial y of the slab goes to scratch S[slab][i]; ld_rowsum_combine_kernel adds the slabs
// in order.  Workgroups never wait for each other; summation orders are fixed.
// --------------------------------------------------------------------------------------------
#define EIG_THREADS 256
#define EIG_TALL_THREADS 512                // blocks of 3 073 .. 6 144 rows: twice the threads, R = 12
#define EIG_MAX_ROWS (EIG_TALL_THREADS * 12)     // tallest block the fused kernels take

// rows per thread: 2, 4, 8, 12 (256 threads); 24 = the tall class (12 rows per thread of 512); or
// 0: too tall, two-pass kernels
int eig_rows_per_thread(int n) {
    if (n <= 2 * EIG_THREADS) return 2;
    if (n <= 4 * EIG_THREADS) return 4;
    if (n <= 8 * EIG_THREADS) return 8;
    if (n <= 12 * EIG_THREADS) return 12;
    if (n <= EIG_MAX_ROWS) return 24;
    return 0;
}
// (Measured and not adopted, round 4: batches of 4 / 4 / 2 / 1 columns with a register budget of
// three workgroups per CU for the one-launch kernel -- 126 / 168 registers instead of 174 / 230:
// C4 0.572 / 0.566 ms per product against 0.603 / 0.567, C4f 0.521 / 0.551 against 0.517 / 0.526,
// alternating on one box: inside the spread between processes.  gpurun_out/r04w.)
int eig_batch_cols(int R) { return R <= 2 ? 8 : (R <= 4 ? 4 : 2); }

#define EIG_RED_SLOTS 16                      // values per batch, at most (R = 2, two right-hand sides)
template <int R, int NR, int T = EIG_THREADS>
static __device__ __forceinline__ void eig_fused_body(
    const EigItem &it, const PoolPair &pools, double *__restrict__ scratch, int64_t s_stride,
    double *__restrict__ dot_partials, int dot_stride, double (&red)[2][T / 64][EIG_RED_SLOTS]) {
    constexpr int H = R / 2;                       // 16-byte loads per column per thread
    constexpr int C = R <= 2 ? 8 : (R <= 4 ? 4 : 2);   // columns per batch: <= 12 loads in flight
    constexpr int V = C * NR;                      // values reduced per batch
    constexpr int G = (V + 7) / 8;                 // butterflies per batch
    constexpr int NW = T / 64;
    const int n = it.n, ncols = it.ncols;
    const int64_t ldc = it.ldc;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int row0 = 2 * threadIdx.x;              // + 512 i
    double x[NR][R], y[NR][R];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double *xg = pools.p[r] + it.x_off;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const int row = row0 + 2 * T * i;
            x[r][2 * i] = row < n ? xg[row] : 0.0;
            x[r][2 * i + 1] = row + 1 < n ? xg[row + 1] : 0.0;
            y[r][2 * i] = y[r][2 * i + 1] = 0.0;
        }
    }
    // Loads are UNCONDITIONAL, at addresses clamped into the slab (a row pair past the block's end
    // re-reads the column's last pair, a column past the slab's end its last column): what they
    // return there is finite and meets x = 0 (rows) or a zero eigenvalue factor (columns), and y of
    // a row past the end is never stored.  Guarded loads (`ok ? load : 0`) put every load in a
    // basic block of its own; the compiler then cannot count what is in flight and waits for
    // vmcnt(0) -- for the NEXT batch too -- before it touches this one.
    int roff[H];
#pragma unroll
    for (int i = 0; i < H; ++i) roff[i] = min(row0 + 2 * T * i, (int)ldc - 2);
    // eigenvalue factors: wave-uniform, written at load time only -> scalar loads.  As vector loads
    // they share vmcnt with the stream, and waiting for one of them is waiting for the next batch.
    const const_tab scale_tab = as_table(it.scale);
    v2d nxt[C][H];
    auto issue = [&](int c0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double *cp = it.a + (int64_t)min(c0 + c, ncols - 1) * ldc;
#pragma unroll
            for (int i = 0; i < H; ++i) nxt[c][i] = LD_STREAM_LOAD(cp + roff[i]);
        }
    };
    issue(0);
    int buf = 0;
    for (int c0 = 0; c0 < ncols; c0 += C, buf ^= 1) {
        v2d cur[C][H];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int i = 0; i < H; ++i) cur[c][i] = nxt[c][i];
        issue(c0 + C);                             // nothing is loaded past the slab's last column
        double p[G * 8];
#pragma unroll
        for (int u = 0; u < G * 8; ++u) p[u] = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    sacc = fma(cur[c][i].x, x[r][2 * i], sacc);
                    sacc = fma(cur[c][i].y, x[r][2 * i + 1], sacc);
                }
                p[c * NR + r] = sacc;
            }
#pragma unroll
        for (int g = 0; g < G; ++g) {
            double q8[CS_ROWS];
#pragma unroll
            for (int u = 0; u < 8; ++u) q8[u] = p[8 * g + u];
            int slot;
            const double tot = sym_rowsum8(q8, lane, slot);
            if ((lane & 7) == 0) red[buf][w][8 * g + slot] = tot;
        }
        __syncthreads();
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double sc = c0 + c < ncols ? scale_tab[c0 + c] : 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double t = red[buf][0][c * NR + r];
#pragma unroll
                for (int ww = 1; ww < NW; ++ww) t += red[buf][ww][c * NR + r];
                t *= sc;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    y[r][2 * i] = fma(cur[c][i].x, t, y[r][2 * i]);
                    y[r][2 * i + 1] = fma(cur[c][i].y, t, y[r][2 * i + 1]);
                }
            }
        }
    }
    if (it.direct) {
        // the slab is the whole block (uniform over the workgroup): y goes to the pool, and the
        // block's y.z partial is formed from the x still in registers -- no scratch, no combine item
        __syncthreads();                            // the last batch's column sums have been read
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double *yo = const_cast<double *>(pools.p[r]) + it.s_off;
            double d = 0.0;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const int row = row0 + 2 * T * i;
                if (row < n) { yo[row] = y[r][2 * i]; d = fma(y[r][2 * i], x[r][2 * i], d); }
                if (row + 1 < n) { yo[row + 1] = y[r][2 * i + 1]; d = fma(y[r][2 * i + 1], x[r][2 * i + 1], d); }
            }
            d = wave_sum(d);
            if (lane == 0) red[0][w][r] = d;
        }
        __syncthreads();
        if (threadIdx.x < NR) {
            double d = red[0][0][threadIdx.x];
#pragma unroll
            for (int ww = 1; ww < NW; ++ww) d += red[0][ww][threadIdx.x];
            dot_partials[(int64_t)threadIdx.x * dot_stride + it.direct - 1] = d;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        double *so = scratch + r * s_stride + it.s_off;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const int row = row0 + 2 * T * i;
            // (write-through, like ld_sym_kernel's partial sums: see LD_PARTIAL_STORE; row is even
            // and so is the record's offset)
            if (row + 1 < n) LD_PARTIAL_STORE2(&so[row], y[r][2 * i], y[r][2 * i + 1]);
            else if (row < n) LD_PARTIAL_STORE(&so[row], y[r][2 * i]);
        }
    }
}

// Blocks of up to 512 rows: ONE WAVE owns a slab of columns (lane l holds rows 128 i + 2 l, + 1,
// i < 4), so the column sums need the in-wave butterfly only -- no LDS, no workgroup barrier -- and
// the four waves of a workgroup work on four different slabs.  With the workgroup-wide kernel a
// batch of this class is 8 columns x <= 512 rows = 32 KB between barriers, and with two
// right-hand sides it needs two butterflies, an LDS exchange and a barrier per batch: 277 us
// against 102 us with one right-hand side at C4 (profiles/r02q_c4_stats_kernel_stats.csv), slower
// than two separate products.  Here a batch is 2 columns (8 loads of 1 KiB in flight per wave,
// the next batch's issued before this one is used), its 2 NR dot products go through one
// butterfly and come back as wave-uniform scalars (v_readlane).
#define EIGW_ROWS 512
#define EIGW_C 2
template <int NR>
static __device__ __forceinline__ void eig_wave_body(const EigItem &it, const PoolPair &pools,
                                                     double *__restrict__ scratch, int64_t s_stride,
                                                     double *__restrict__ dot_partials, int dot_stride) {
    constexpr int H = EIGW_ROWS / 128;             // 16-byte loads per column per lane
    constexpr int C = EIGW_C;
    const int n = it.n, ncols = it.ncols;
    const int64_t ldc = it.ldc;
    const int lane = threadIdx.x & 63;
    const int row0 = 2 * lane;                     // + 128 i
    double x[NR][2 * H], y[NR][2 * H];
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        const double *xg = pools.p[r] + it.x_off;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const int row = row0 + 128 * i;
            x[r][2 * i] = row < n ? xg[row] : 0.0;
            x[r][2 * i + 1] = row + 1 < n ? xg[row + 1] : 0.0;
            y[r][2 * i] = y[r][2 * i + 1] = 0.0;
        }
    }
    int roff[H];                                   // unconditional, clamped loads: see eig_fused_body
#pragma unroll
    for (int i = 0; i < H; ++i) roff[i] = min(row0 + 128 * i, (int)ldc - 2);
    const const_tab scale_tab = as_table(it.scale);     // scalar loads: see eig_fused_body
    v2d nxt[C][H];
    auto issue = [&](int c0) {
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double *cp = it.a + (int64_t)min(c0 + c, ncols - 1) * ldc;
#pragma unroll
            for (int i = 0; i < H; ++i) nxt[c][i] = LD_STREAM_LOAD(cp + roff[i]);
        }
    };
    issue(0);
    for (int c0 = 0; c0 < ncols; c0 += C) {
        v2d cur[C][H];
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int i = 0; i < H; ++i) cur[c][i] = nxt[c][i];
        issue(c0 + C);                             // nothing is loaded past the slab's last column
        double p[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) p[u] = 0.0;
#pragma unroll
        for (int c = 0; c < C; ++c)
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                double sacc = 0.0;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    sacc = fma(cur[c][i].x, x[r][2 * i], sacc);
                    sacc = fma(cur[c][i].y, x[r][2 * i + 1], sacc);
                }
                p[c * NR + r] = sacc;
            }
        int slot;
        const double tot = sym_rowsum8(p, lane, slot);
        // value v sits in the lanes whose bits 5..3 spell v: lane 8 * perm(v); read each back as a
        // wave-uniform scalar
        const int lo = __double2loint(tot), hi = __double2hiint(tot);
#pragma unroll
        for (int c = 0; c < C; ++c) {
            const double sc = c0 + c < ncols ? scale_tab[c0 + c] : 0.0;
#pragma unroll
            for (int r = 0; r < NR; ++r) {
                const int v = c * NR + r;
                // slot = bit5*4 + bit4*2 + bit3  =>  lane with slot v and bits 2..0 = 0
                const int src = ((v & 4) ? 32 : 0) | ((v & 2) ? 16 : 0) | ((v & 1) ? 8 : 0);
                const double t = __hiloint2double(__builtin_amdgcn_readlane(hi, src),
                                                  __builtin_amdgcn_readlane(lo, src)) * sc;
#pragma unroll
                for (int i = 0; i < H; ++i) {
                    y[r][2 * i] = fma(cur[c][i].x, t, y[r][2 * i]);
                    y[r][2 * i + 1] = fma(cur[c][i].y, t, y[r][2 * i + 1]);
                }
            }
        }
    }
    if (it.direct) {                                // the slab is the whole block: see eig_fused_body
#pragma unroll
        for (int r = 0; r < NR; ++r) {
            double *yo = const_cast<double *>(pools.p[r]) + it.s_off;
            double d = 0.0;
#pragma unroll
            for (int i = 0; i < H; ++i) {
                const int row = row0 + 128 * i;
                if (row < n) { yo[row] = y[r][2 * i]; d = fma(y[r][2 * i], x[r][2 * i], d); }
                if (row + 1 < n) { yo[row + 1] = y[r][2 * i + 1]; d = fma(y[r][2 * i + 1], x[r][2 * i + 1], d); }
            }
            d = wave_sum(d);
            if (lane == 0) dot_partials[(int64_t)r * dot_stride + it.direct - 1] = d;
        }
        return;
    }
#pragma unroll
    for (int r = 0; r < NR; ++r) {
        double *so = scratch + r * s_stride + it.s_off;
#pragma unroll
        for (int i = 0; i < H; ++i) {
            const int row = row0 + 128 * i;
            if (row + 1 < n) LD_PARTIAL_STORE2(&so[row], y[r][2 * i], y[r][2 * i + 1]);
            else if (row < n) LD_PARTIAL_STORE(&so[row], y[r][2 * i]);
        }
    }
}

template <int NR>
__global__ __launch_bounds__(256) void ld_eig_wave_kernel(
    const EigItem *__restrict__ items, int n_items, const PoolPair pools_arg,
    double *__restrict__ scratch, int64_t s_stride, double *__restrict__ dot_partials, int dot_stride,
    const int *pred, const PhasePtrs *pp) {
    PRED_EXIT(pred);
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int idx = blockIdx.x * 4 + w;            // one slab per wave
    if (idx >= n_items) return;
    const EigItem it = items[idx];
    eig_wave_body<NR>(it, pools, scratch, s_stride, dot_partials, dot_stride);
}

void launch_ld_eig_wave(const EigItem *items, int n_items, const double *pool0, const double *pool1,
                        double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                        hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    const dim3 grid((n_items + 3) / 4), block(256);
    if (pool1)
        hipLaunchKernelGGL((ld_eig_wave_kernel<2>), grid, block, 0, s, items, n_items, pp, scratch,
                           s_stride, dot_partials, dot_stride, g_pred, g_phase);
    else
        hipLaunchKernelGGL((ld_eig_wave_kernel<1>), grid, block, 0, s, items, n_items, pp, scratch,
                           s_stride, dot_partials, dot_stride, g_pred, g_phase);
}

// One launch per block-height class present (R = rows per thread).  A single kernel switching on
// the class at run time was measured and is slower (C4, one right-hand side: 695 us against
// 620 us for the four launches together): it runs every class with the register budget of the
// tallest (172 VGPRs: 2 waves per SIMD instead of 3-4).
template <int R, int NR>
__global__ __launch_bounds__(EIG_THREADS) void ld_eig_fused_kernel(
    const EigItem *__restrict__ items, const PoolPair pools_arg, double *__restrict__ scratch,
    int64_t s_stride, double *__restrict__ dot_partials, int dot_stride, const int *pred,
    const PhasePtrs *pp) {
    __shared__ double red[2][EIG_THREADS / 64][EIG_RED_SLOTS];
    PRED_EXIT(pred);
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const EigItem it = items[blockIdx.x];
    eig_fused_body<R, NR>(it, pools, scratch, s_stride, dot_partials, dot_stride, red);
}

// The tall class (3 073 .. 6 144 rows): the same body with 512 threads, 12 rows each, so U is read
// once here too (228 registers with two right-hand sides: two waves per SIMD, which 512 threads
// are).  Taller blocks still take the two-pass kernels.
template <int NR>
__global__ __launch_bounds__(EIG_TALL_THREADS) void ld_eig_tall_kernel(
    const EigItem *__restrict__ items, const PoolPair pools_arg, double *__restrict__ scratch,
    int64_t s_stride, double *__restrict__ dot_partials, int dot_stride, const int *pred,
    const PhasePtrs *pp) {
    __shared__ double red[2][EIG_TALL_THREADS / 64][EIG_RED_SLOTS];
    PRED_EXIT(pred);
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const EigItem it = items[blockIdx.x];
    eig_fused_body<12, NR, EIG_TALL_THREADS>(it, pools, scratch, s_stride, dot_partials, dot_stride, red);
}

void launch_ld_eig_tall(const EigItem *items, int n_items, const double *pool0, const double *pool1,
                        double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                        hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    const dim3 grid(n_items), block(EIG_TALL_THREADS);
    if (pool1)
        hipLaunchKernelGGL((ld_eig_tall_kernel<2>), grid, block, 0, s, items, pp, scratch, s_stride,
                           dot_partials, dot_stride, g_pred, g_phase);
    else
        hipLaunchKernelGGL((ld_eig_tall_kernel<1>), grid, block, 0, s, items, pp, scratch, s_stride,
                           dot_partials, dot_stride, g_pred, g_phase);
}

// All classes in one launch: the item says how many rows per thread its block needs.  Every class
// then runs with the register budget of the tallest, which costs ~10 % on a full-size product --
// but a small shard (an 8-GPU rank: ~1 300 items in all) gains more from one ramp and one tail
// instead of four; launch_ld_eig_fused_all is used below EIG_MERGE_BELOW items.
template <int NR>
__global__ __launch_bounds__(EIG_THREADS) void ld_eig_fused_all_kernel(
    const EigItem *__restrict__ items, const PoolPair pools_arg, double *__restrict__ scratch,
    int64_t s_stride, double *__restrict__ dot_partials, int dot_stride, const int *pred,
    const PhasePtrs *pp) {
    __shared__ double red[2][EIG_THREADS / 64][EIG_RED_SLOTS];
    PRED_EXIT(pred);
    PoolPair pools = pools_arg;
    if (pp != nullptr) { pools.p[0] = PHASE(pp)->pool_out; pools.p[1] = NR == 2 ? PHASE(pp)->pool_out2 : PHASE(pp)->pool_out; }
    const EigItem it = items[blockIdx.x];
    if (it.n <= 2 * EIG_THREADS) eig_fused_body<2, NR>(it, pools, scratch, s_stride, dot_partials, dot_stride, red);
    else if (it.n <= 4 * EIG_THREADS) eig_fused_body<4, NR>(it, pools, scratch, s_stride, dot_partials, dot_stride, red);
    else if (it.n <= 8 * EIG_THREADS) eig_fused_body<8, NR>(it, pools, scratch, s_stride, dot_partials, dot_stride, red);
    else eig_fused_body<12, NR>(it, pools, scratch, s_stride, dot_partials, dot_stride, red);
}

void launch_ld_eig_fused_all(const EigItem *items, int n_items, const double *pool0,
                             const double *pool1, double *scratch, int64_t s_stride,
                             double *dot_partials, int dot_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    const dim3 grid(n_items), block(EIG_THREADS);
    if (pool1)
        hipLaunchKernelGGL((ld_eig_fused_all_kernel<2>), grid, block, 0, s, items, pp, scratch,
                           s_stride, dot_partials, dot_stride, g_pred, g_phase);
    else
        hipLaunchKernelGGL((ld_eig_fused_all_kernel<1>), grid, block, 0, s, items, pp, scratch,
                           s_stride, dot_partials, dot_stride, g_pred, g_phase);
}

template <int R>
static void launch_eig_r(const EigItem *items, int n_items, const PoolPair &pp, bool two,
                         double *scratch, int64_t s_stride, double *dot_partials, int dot_stride,
                         hipStream_t s) {
    const dim3 grid(n_items), block(EIG_THREADS);
    if (two)
        hipLaunchKernelGGL((ld_eig_fused_kernel<R, 2>), grid, block, 0, s, items, pp, scratch,
                           s_stride, dot_partials, dot_stride, g_pred, g_phase);
    else
        hipLaunchKernelGGL((ld_eig_fused_kernel<R, 1>), grid, block, 0, s, items, pp, scratch,
                           s_stride, dot_partials, dot_stride, g_pred, g_phase);
}

void launch_ld_eig_fused(const EigItem *items, int n_items, int R, const double *pool0,
                         const double *pool1, double *scratch, int64_t s_stride,
                         double *dot_partials, int dot_stride, hipStream_t s) {
    if (n_items <= 0) return;
    PoolPair pp;
    pp.p[0] = pool0;
    pp.p[1] = pool1 ? pool1 : pool0;
    switch (R) {
        case 2: launch_eig_r<2>(items, n_items, pp, pool1 != nullptr, scratch, s_stride, dot_partials, dot_stride, s); break;
        case 4: launch_eig_r<4>(items, n_items, pp, pool1 != nullptr, scratch, s_stride, dot_partials, dot_stride, s); break;
        case 8: launch_eig_r<8>(items, n_items, pp, pool1 != nullptr, scratch, s_stride, dot_partials, dot_stride, s); break;
        case 12: launch_eig_r<12>(items, n_items, pp, pool1 != nullptr, scratch, s_stride, dot_partials, dot_stride, s); break;
        default: break;
    }
}

// row-major src [n x r] -> column-major dst [r][ldc], rows n .. ldc-1 zero (load time).  A 32 x 32
// tile goes through LDS so both sides are coalesced.
__global__ __launch_bounds__(256) void repack_columns_kernel(const double *__restrict__ src, int n,
                                                             int r, int64_t ldc,
                                                             double *__restrict__ dst) {
    __shared__ double tile[32][33];
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;       // 32 x 8
    const int j0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int j = j0 + ty + 8 * k, c = c0 + tx;
        tile[ty + 8 * k][tx] = (j < n && c < r) ? src[(int64_t)j * r + c] : 0.0;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int c = c0 + ty + 8 * k, j = j0 + tx;
        if (c < r && j < ldc) dst[(int64_t)c * ldc + j] = tile[tx][ty + 8 * k];
    }
}
void launch_repack_columns(const double *src, int n, int r, int64_t ldc, double *dst,
                           hipStream_t s) {
    const dim3 grid((unsigned)((ldc + 31) / 32), (unsigned)((r + 31) / 32));
    hipLaunchKernelGGL(repack_columns_kernel, grid, dim3(256), 0, s, src, n, r, ldc, dst);
}

// --------------------------------------------------------------------------------------------
// exp and 1/sqrt of the per-SNP passes.  These passes are co-bound by the vector ALU (profiles/
// r03o_c3_pmc_sq.txt: ~70 % busy), so what the library versions spend on inputs that cannot occur
// here is worth removing: exp() selects its result for +-inf and out-of-range arguments (two
// compares, three selects per call), rsqrt() tests the class of its seed (a compare, two selects).
//   pass_exp(x): x = n ln 2 + r, |r| <= ln 2 / 2; e^r = 1 + r + r^2 q(r), q of degree 9 fitted at
//     the Chebyshev nodes of the interval (truncation 1.6e-17 relative with the coefficients
//     rounded to double); the result is scaled by 2^n with ldexp, which overflows to inf and
//     underflows through the denormals to 0 by itself (n saturates in the conversion).  NaN in,
//     NaN out; +-inf (never formed from finite logits) gives NaN, which the normaliser check of
//     snp_pass_kernel treats like any other Z out of range.
//   pass_rsqrt(x): the hardware seed (v_rsq_f64, ~2^-23) and one third-order step,
//     y (1 + e/2 + 3 e^2/8) with e = 1 - x y^2: below 1 ulp for normal x > 0 (pivots of SPD
//     matrices with entries of 1e0 .. 1e12).
// --------------------------------------------------------------------------------------------
// The coefficients are pinned to scalar registers once per kernel (exp_consts()): as
// vector-register operands the compiler ties each to the accumulator of a two-address v_fmac and
// copies it first -- one v_mov_b64 per Horner step.
struct ExpConsts { double c[9]; };
static __device__ __forceinline__ ExpConsts exp_consts() {
    ExpConsts k = {{0x1.28918e390a71ep-22, 0x1.71de0da9d06acp-19, 0x1.a019b905bafb1p-16,
                    0x1.a01a01a7c54aep-13, 0x1.6c16c1788dc33p-10, 0x1.11111111109b2p-7,
                    0x1.5555555553d5ep-5, 0x1.5555555555556p-3, 0x1.0000000000001p-1}};
#pragma unroll
    for (int j = 0; j < 9; ++j) asm("" : "+s"(k.c[j]));
    return k;
}
static __device__ __forceinline__ double pass_exp(double x, const ExpConsts &k) {
    const double n = __builtin_rint(x * 0x1.71547652b82fep+0);
    double r = fma(n, -0x1.62e42fefa39efp-1, x);
    r = fma(n, -0x1.abc9e3b39803fp-56, r);
    double t = 0x1.af38b4925be09p-26;
#pragma unroll
    for (int j = 0; j < 9; ++j) t = fma(t, r, k.c[j]);
    t = fma(t, r, 1.0);
    t = fma(t, r, 1.0);
    return __builtin_ldexp(t, (int)n);
}
static __device__ __forceinline__ double pass_rsqrt(double x) {
    const double y = __builtin_amdgcn_rsq(x);
    const double e = fma(-x * y, y, 1.0);
    return fma(y * e, fma(e, 0.375, 0.5), y);
}

// --------------------------------------------------------------------------------------------
// small SPD helpers, fully unrolled so everything stays in registers
// --------------------------------------------------------------------------------------------
// det(lam)^(-1/2) of a symmetric positive definite matrix
template <int P>
static __device__ __forceinline__ double spd_rsqrt_det(const double (&lam)[P][P]) {
    if constexpr (P == 1) {
        return pass_rsqrt(lam[0][0]);
    } else if constexpr (P == 2) {
        return pass_rsqrt(lam[0][0] * lam[1][1] - lam[0][1] * lam[1][0]);
    } else {
        double G[P][P];
        double w = 1.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            double s = lam[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= G[j][k] * G[j][k];
            const double rj = pass_rsqrt(s);
            G[j][j] = s * rj;
            w *= rj;
#pragma unroll
            for (int i = j + 1; i < P; ++i) {
                double t = lam[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) t -= G[i][k] * G[j][k];
                G[i][j] = t * rj;
            }
        }
        return w;
    }
}

// sig = inverse(lam), returns det(lam)^(-1/2); lam symmetric positive definite
template <int P>
static __device__ __forceinline__ double spd_inverse(const double (&lam)[P][P], double (&sig)[P][P]) {
    if constexpr (P == 1) {
        const double w = pass_rsqrt(lam[0][0]);
        sig[0][0] = w * w;
        return w;
    } else if constexpr (P == 2) {
        // closed form of the reference's 2x2 helper (numerics.py:223-232); 1/det = w^2
        const double det = lam[0][0] * lam[1][1] - lam[0][1] * lam[1][0];
        const double w = pass_rsqrt(det);
        const double r = w * w;
        sig[0][0] = lam[1][1] * r;
        sig[1][1] = lam[0][0] * r;
        sig[0][1] = -lam[1][0] * r;
        sig[1][0] = sig[0][1];
        return w;
    } else {
        // Cholesky lam = G G^T; Gi = G^-1 (its diagonal entries are 1/G_jj, so det^-1/2 is
        // their product); sig = Gi^T Gi
        double G[P][P], Gi[P][P];
        double w = 1.0;
#pragma unroll
        for (int j = 0; j < P; ++j) {
            double s = lam[j][j];
#pragma unroll
            for (int k = 0; k < j; ++k) s -= G[j][k] * G[j][k];
            const double rj = pass_rsqrt(s);
            G[j][j] = s * rj;
            Gi[j][j] = rj;
            w *= rj;
#pragma unroll
            for (int i = j + 1; i < P; ++i) {
                double t = lam[i][j];
#pragma unroll
                for (int k = 0; k < j; ++k) t -= G[i][k] * G[j][k];
                G[i][j] = t * rj;
            }
        }
#pragma unroll
        for (int j = 0; j < P; ++j) {
#pragma unroll
            for (int i = j + 1; i < P; ++i) {
                double t = 0.0;
#pragma unroll
                for (int k = j; k < i; ++k) t += G[i][k] * Gi[k][j];
                Gi[i][j] = -t * Gi[i][i];
            }
        }
#pragma unroll
        for (int a = 0; a < P; ++a) {
#pragma unroll
            for (int b = 0; b <= a; ++b) {
                double t = 0.0;
#pragma unroll
                for (int k = a; k < P; ++k) t += Gi[k][a] * Gi[k][b];
                sig[a][b] = t;
                sig[b][a] = t;
            }
        }
        return w;
    }
}

// --------------------------------------------------------------------------------------------
// fused per-SNP pass.  A workgroup of four waves takes a TILE of 64 SNPs (lane = SNP, coalesced
// along the SNP axis of vi_mu [M][P][N]); the waves split the mixture
// components of the tile between them (wave w owns components [w Q, (w+1) Q), Q = ceil(M / 4)), so
// the per-component tables stay wave-uniform (scalar loads) while a small shard still gets four
// times the waves of a thread-per-SNP layout.
//
// Softmax over components against a FIXED shift.  Responsibilities delta_k ~ exp(u_k),
// u_k = 0.5 (quad_k - log det Lam_k) + lh_k, are accumulated as e_k = w_k exp(a_k - s) with
// a_k = 0.5 quad_k + lh_k, w_k = det^-1/2 (an rsqrt instead of a log per (component, SNP)) and
// s = the log-normaliser of the CURRENT accepted state of that SNP: the candidate's own
// normaliser is Z = sum_k e_k and delta_k = e_k / Z exactly, with no running maximum, no
// rescaling, and every term final the moment it is formed.  That is what lets the e_k be stashed
// (LDS, one double per (component, candidate, SNP)) and turned into the per-annotation
// responsibility sums -- the M-step statistic, sum_annotations (numerics.py:118-129) -- at the
// end of the same pass, for BOTH candidates of a two-step trial, instead of re-reading the
// candidate's vi_mu in a second kernel.  Safety: each wave also tracks max_k a_k; if Z leaves
// [1e-150, 1e150] for any SNP of the tile (a state that moved by hundreds of log units, or no
// reference yet) the tile is redone once with s = max_k a_k, for which Z is within
// [min w, M max w].  The decision is uniform over the workgroup and depends on the data only, so
// results are reproducible.  In the KL terms the log-determinants of fast_delta_kl and
// fast_beta_kl cancel, so their sum needs only quad_k and tr(Prec_k Sig_k).
//
// --------------------------------------------------------------------------------------------
#define SNP_THREADS 256
#ifndef SNP_SPLIT
#define SNP_SPLIT 4                 // waves sharing a tile's components (1: experiment, no stash)
#endif
#define SNP_TILE (SNP_THREADS / SNP_SPLIT)
// components per vi_mu batch (P <= 2; more cohorts always take 2).  2 against 4, final round-3
// kernels, same box: evaluation pass 0.200 -> 0.182 ms, two-step trial 0.482 -> 0.475 ms
// (gpurun_out/ab33.txt): the smaller batch leaves registers for the scheduler, and with the
// branch-free stores nothing needs a deeper batch to hide.
#ifndef KU
#define KU 2
#endif
// batches of a plain evaluation requested ahead of the one being folded in, plus one (2: double
// buffer, as the trials; 3: two ahead -- 100 VGPRs instead of 88, four waves per SIMD instead of
// five, 6 KiB in flight per wave instead of 4).  Measured in round 5 and the same to the percent
// (C3 evaluation 0.1753 against 0.1769 ms, M = 582 1.838 against 1.829 ms; forced back to five waves it
// spills and takes 0.192 ms: profiles/r05q_eval_pass_prefetch_depth.txt): neither what a wave has in
// flight nor the number of waves is what holds the evaluation at 3.8 - 4.1 TB/s.
#ifndef SNP_EVAL_DEPTH
#define SNP_EVAL_DEPTH 2
#endif
// Few components per wave (M = 40: ten): measured and NOT adopted -- batches of 5, so that all of a
// wave's vi_mu is requested up front.  The loop shrinks (6.0 k -> 4.7 k cycles) but the kernel
// needs 129 - 166 registers instead of 92 - 128, fewer workgroups are in flight per CU and both
// passes get slower (evaluation 0.155 -> 0.166 ms, two-step trial 0.474 -> 0.503 ms, one box:
// profiles/r04e_snp_pass_timeline.txt).

int snp_pass_grid(int64_t N) { return (int)((N + SNP_THREADS - 1) / SNP_THREADS); }
// workgroups (= tiles) of launch_snp_pass = rows of its partials
int snp_tile_grid(int64_t N) { return (int)((N + 63) / 64); }
int snp_sum_rows(int64_t N, int A) { (void)A; return snp_tile_grid(N); }

// components per wave = stash slots per wave
// (with vi_mu in row pairs, an odd number of cohorts pairs rows of two components: every wave's range
// then starts at an even component)
static __host__ __device__ inline int snp_wave_comps(int M, int P) {
    const int q = (M + 3) / 4;
    return (MU_PAIRED && (P & 1)) ? (q + 1) / 2 * 2 : q;
}
static inline int snp_slots(int M, int P) { return snp_wave_comps(M, P); }
// accumulators a wave hands to wave 0 per candidate: Skl, Sip, Sm[P], S2[P]
static inline int snp_nacc(int P) { return 2 + 2 * P; }
size_t snp_pass_lds_bytes(int M, int P, int ns, bool stash) {
    const size_t zx = (size_t)4 * ns * 2 * SNP_TILE * sizeof(double);         // every wave's Z, max a
    const size_t hand = (size_t)3 * ns * snp_nacc(P) * SNP_TILE * sizeof(double);   // waves 1..3 -> 0
    const size_t st = stash ? (size_t)snp_slots(M, P) * ns * SNP_THREADS * sizeof(double) : 0;
    return zx + std::max(hand, st);      // the hand-over reuses the stash once the sums are formed
}
// The stash is used when it leaves room for at least two workgroups in a CU's 160 KB of LDS (one
// is not enough to hide the pass's latencies: profiles/r02s_ab_snp_pass_occupancy.txt); beyond
// that the sums come from delta_kernel as before.
bool snp_pass_can_stash(int M, int P, int ns) {
    return SNP_SPLIT == 4 && P <= 4 && snp_pass_lds_bytes(M, P, ns, true) <= (size_t)78 * 1024;
}

// sum over the 64 lanes of 8 values at once (halving butterfly, DPP / permlane moves only): the
// total of value `row` ends up in every lane whose bits 5..3 spell `row`
static __device__ __forceinline__ double tile_sum8(const double (&p)[8], int lane, int &row) {
    return sym_rowsum8(p, lane, row);
}

// Register budget: three waves per SIMD up to four cohorts.  For up to two that is what the kernels
// need anyway (92 - 128 VGPRs since round 4).  With the Cholesky of four cohorts the budget of 168
// spills 10 - 16 registers (round 3: dozens, and two waves were better; after round 4's cheaper
// arithmetic three are: C5 evaluation 3.34 -> 3.23 ms, trial 7.55 - 7.70 -> 7.26 - 7.40 ms, alternating
// on one box, gpurun_out/r04G).
#ifndef SNP_MIN_WAVES
#define SNP_MIN_WAVES(P) ((P) <= 4 ? 3 : 1)
#endif
// (a plain evaluation is a chain of HBM round trips per tile: what it has in flight is waves x batch)
#ifndef SNP_EVAL_WAVES
#define SNP_EVAL_WAVES(P) SNP_MIN_WAVES(P)
#endif
// Workgroup barrier that orders LDS traffic only.  __syncthreads() also waits for every global
// store of the wave to be acknowledged (vmcnt(0)): with the pass's vi_mu stores in flight that is
// microseconds per barrier, and a tile has three.  Nothing in this kernel is handed between
// waves through global memory.
static __device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <int P, bool BLEND, bool ONE_ANNOT, int NS, bool STASH, bool NOSTORE = false, bool NOBASE = false>
__global__ __launch_bounds__(SNP_THREADS, BLEND ? SNP_MIN_WAVES(P) : SNP_EVAL_WAVES(P)) void snp_pass_kernel(const SnpKernelArgs a) {
    static_assert(NS == 1 || BLEND, "two candidates only make sense for a beta trial");
    // NOSTORE with BLEND: a lazy trial; without: a plain evaluation of a lazy state (both derive
    // mu_k = a (stored vi_mu) + Sig_k c component by component and store no vi_mu).  A lazy trial
    // may keep the stash (mixtures that fit it): its candidates' responsibility sums then come out
    // of this pass as a storing trial's do.
    static_assert(!NOSTORE || !STASH || BLEND, "only trials stash");
    // NOBASE: a lazy pass that is KNOWN (by the host, when it queues it: SnpKernelArgs::no_store == 2) to
    // work on a state with a == 0 -- no vi_mu load is compiled in, nor the arithmetic on what it would
    // have returned (the passes are bound by the vector ALU there).  The other lazy passes find
    // a == 0 out at run time and point their loads at one L2-resident tile (below).
    static_assert(!NOBASE || NOSTORE, "base-free passes are lazy passes");
    constexpr int NT = 2 * P + 2;
    constexpr int NTP = (NT + 7) / 8 * 8;
    constexpr int NACC = 2 + 2 * P;
    constexpr int KB = P <= 2 ? KU : (KU > 2 ? 2 : KU);
    constexpr int TB = KB < 2 ? KB : 2;            // components whose tables are fetched together
    extern __shared__ double lds[];
    PRED_EXIT(a.pred);
    // the buffers and step sizes: from the arguments, or -- for a sweep queued ahead of the
    // decision that assigns the buffers their roles -- from the device-resident phase block
    PhasePtrs q;
    if (a.pp != nullptr) {
        const phase_tab t = PHASE(a.pp);
        q.mu_in = t->mu_in; q.mu_out = t->mu_out; q.mu_out2 = t->mu_out2;
        q.pool_cur = t->pool_cur; q.m_cur = t->m_cur; q.lse_ref = t->lse_ref;
        q.pool_out = t->pool_out; q.m_out = t->m_out; q.v_out = t->v_out; q.lse_out = t->lse_out;
        q.pool_out2 = t->pool_out2; q.m_out2 = t->m_out2; q.v_out2 = t->v_out2; q.lse_out2 = t->lse_out2;
        q.step = t->step; q.step2 = t->step2;
        q.snap_in = t->snap_in; q.snap_out = t->snap_out;
        q.c_cur = t->c_cur; q.c_out = t->c_out; q.c_out2 = t->c_out2;
        q.a_def = t->a_def; q.c_zero = t->c_zero;
#pragma unroll
        for (int p = 0; p < P; ++p) q.tau[p] = t->tau[p];
    } else {
        q.mu_in = a.mu_in; q.mu_out = a.mu_out; q.mu_out2 = a.mu_out2;
        q.pool_cur = a.pool_cur; q.m_cur = a.m_cur; q.lse_ref = a.lse_ref;
        q.pool_out = a.pool_out; q.m_out = a.m_out; q.v_out = a.v_out; q.lse_out = a.lse_out;
        q.pool_out2 = a.pool_out2; q.m_out2 = a.m_out2; q.v_out2 = a.v_out2; q.lse_out2 = a.lse_out2;
        q.step = a.step; q.step2 = a.step2;
        q.snap_in = a.snapshot; q.snap_out = a.snapshot_out;
        q.c_cur = nullptr; q.c_out = nullptr; q.c_out2 = nullptr;      // (lazy trials are queued ones)
        q.a_def = 1.0; q.c_zero = 1;
#pragma unroll
        for (int p = 0; p < P; ++p) q.tau[p] = a.tau.v[p];
    }
#if SNP_TRACE
    long long stamp[7];
#define SNP_STAMP(j) stamp[j] = __builtin_readcyclecounter()
    const long long real0 = (long long)__builtin_amdgcn_s_memrealtime();     // 100 MHz, one clock for the chip
#else
#define SNP_STAMP(j)
#endif
    SNP_STAMP(0);
    const ExpConsts expk = exp_consts();
    const int N = a.N, M = a.M;
    const int64_t N64 = N;
    const int lane = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    // this wave's components
    const int Q = SNP_SPLIT == 4 ? snp_wave_comps(M, P) : M;
    const int kbeg = SNP_SPLIT == 4 ? w * Q : 0, kend = min(M, kbeg + Q);
    const int rowid = SNP_SPLIT == 4 ? blockIdx.x : blockIdx.x * 4 + w;     // row of the partials
    const int nrows = SNP_SPLIT == 4 ? gridDim.x : (N + 63) / 64;
    // LDS: [4 waves][NS][2][64] Z and max a | stash [Q][NS][256], later the hand-over [3][NS][NACC][64]
    double *zx_lds = lds;
    double *stash = zx_lds + 4 * NS * 2 * SNP_TILE;
    double *hand_lds = stash;

    const int i = SNP_SPLIT == 4 ? blockIdx.x * 64 + lane : blockIdx.x * 256 + threadIdx.x;
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    // A lazy pass whose state has a == 0 -- mu_k = Sig_k c: what _initialize builds and every blend of
    // it stays (a' = (1 - s) a) -- needs no stored vi_mu at all.  Its loads are not compiled out but
    // pointed at tile 0 of the buffer (40 KB that stay in L2; 0 * finite = 0): no vi_mu byte comes
    // from HBM.
    const bool nobase = NOBASE || (NOSTORE && !q.c_zero && q.a_def == 0.0);   // uniform over the launch
    const int64_t mu_base = MU_BASE(nobase ? (ii & (MU_TILE - 1)) : ii, M, P, N64);

    // vi_mu is read in batches of KB components (KB*P independent 512-B wave loads), double
    // buffered (see below).  A plain evaluation requests its first batch before anything else, so
    // the per-SNP constants and the divisions that follow do not add a trip to HBM in front of it
    // (C3: 0.207 -> 0.198 ms).  A trial does not: the same move costs it 0.02 ms
    // (gpurun_out/ab23.txt of round 3; it is bound by its stores, not by latency).
    auto fetch_mu = [&](double (&dst)[KB][P], int k0) {
        if constexpr (NOBASE) {
#pragma unroll
            for (int kk = 0; kk < KB; ++kk)
#pragma unroll
                for (int p = 0; p < P; ++p) dst[kk][p] = 0.0;
            (void)k0;
            return;
        }
#if MU_PAIRED
        // the batch's KB P rows are KB P / 2 row pairs: 16 bytes per lane and load (k0 P is even)
        static_assert(KB % 2 == 0, "row pairs of an odd number of cohorts span two components");
        const int j0 = (k0 * P) >> 1, jmax = (int)MU_PAIRS(M, P) - 1;
#pragma unroll
        for (int t = 0; t < KB * P / 2; ++t) {
            const v2d v = MU_LOAD2(&q.mu_in[mu_base + MU_PAIR(min(j0 + t, jmax))]);   // unconditional; extras ignored
            dst[(2 * t) / P][(2 * t) % P] = v.x;
            dst[(2 * t + 1) / P][(2 * t + 1) % P] = v.y;
        }
#else
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) {
            const int kc = min(k0 + kk, M - 1);       // unconditional loads; extras are ignored
#pragma unroll
            for (int p = 0; p < P; ++p) dst[kk][p] = MU_LOAD(&q.mu_in[mu_base + MU_ROW(kc * P + p, N64)]);
        }
#endif
    };
    double bufA[KB][P], bufB[KB][P], lhA[KB], lhB[KB];
    constexpr bool DEEP = !BLEND && P <= 2 && SNP_EVAL_DEPTH == 3;
    double bufC[DEEP ? KB : 1][P], lhC[DEEP ? KB : 1];
    (void)bufC; (void)lhC;
    constexpr bool FETCH_FIRST = !BLEND;
    if (FETCH_FIRST) fetch_mu(bufA, kbeg);
    const double s0 = q.lse_ref != nullptr ? q.lse_ref[ii] : 0.0;

    double d[P], se[P], adj[P], sld[P], g[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        se[p] = a.se[p * N64 + ii];
        adj[p] = a.adj[p * N64 + ii];
        sld[p] = a.sld[p * N64 + ii];
        d[p] = sld[p] / q.tau[p];
        g[p] = 0.0;
    }
    // Everything the END of the tile needs from memory is requested here, by the wave that will
    // use it: a load issued after the main loop would wait (vmcnt is in order) for every vi_mu
    // store of the loop to be acknowledged first -- microseconds, with the other three waves of
    // the workgroup already gone (profiles/r04e_snp_pass_timeline.txt: the last stretch of wave 0
    // was 24 - 27 % of the workgroup's life).
    int pos[P];
    double scal_t[P], snap_t[P];
    const bool tail_wave = SNP_SPLIT != 4 || w == 0;
    if (BLEND || tail_wave) {
#pragma unroll
        for (int p = 0; p < P; ++p) pos[p] = a.invperm[p * N64 + ii];
    }
    if (!BLEND && a.diff && tail_wave) {
#pragma unroll
        for (int p = 0; p < P; ++p) {
            scal_t[p] = a.scal[p * N64 + ii];
            snap_t[p] = q.snap_in[p * N64 + ii];
        }
    }
    if (BLEND) {
        // _nat_grad_beta (variational_inference.py:804-823); identical for every component k
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double linked = q.pool_cur[(int64_t)(P + p) * N64 + pos[p]];
            const double m = q.m_cur[p * N64 + ii];
            g[p] = (adj[p] - (linked / se[p] - m * sld[p])) / q.tau[p];
        }
    }
    const double *lh = a.lh + (ONE_ANNOT ? 0 : (int64_t)a.annot[ii] * M);
    auto fetch_lh = [&](double (&lhv)[KB], int k0) {
#pragma unroll
        for (int kk = 0; kk < KB; ++kk) lhv[kk] = ONE_ANNOT ? 0.0 : lh[min(k0 + kk, M - 1)];
    };
    auto fetch = [&](double (&dst)[KB][P], double (&lhv)[KB], int k0) {
        fetch_mu(dst, k0);
        fetch_lh(lhv, k0);
    };
    if (FETCH_FIRST) fetch_lh(lhA, kbeg);
    double step[NS];
    double *mu_out[NS];
    step[0] = q.step;
    mu_out[0] = q.mu_out;
    if (NS == 2) { step[NS - 1] = q.step2; mu_out[NS - 1] = q.mu_out2; }
    // A lazy trial (PhasePtrs): the state it starts from is a0 (stored vi_mu) + Sig c0, its candidates
    // are ac (stored vi_mu) + Sig cc with ac = (1 - s) a0, cc = (1 - s) c0 + s g -- the blend
    // Sig (s g + (1 - s) Lam mu) written in the two numbers that describe it.  The candidates' cc go
    // beside their moments (one wave per tile writes them).
    double ac[NS], cc[NS][P];
    if (NOSTORE && BLEND) {
        const double a0 = q.c_zero ? 1.0 : q.a_def;
#pragma unroll
        for (int c = 0; c < NS; ++c) ac[c] = a0 * (1.0 - step[c]);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            const double c0 = q.c_zero ? 0.0 : q.c_cur[p * N64 + ii];
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                cc[c][p] = (1.0 - step[c]) * c0 + step[c] * g[p];
                if ((SNP_SPLIT != 4 || w == 0) && live) (c == 0 ? q.c_out : q.c_out2)[p * N64 + i] = cc[c][p];
            }
        }
    } else if (NOSTORE) {
        // a plain evaluation of the lazy state itself: (a, c) as they stand
        ac[0] = q.c_zero ? 1.0 : q.a_def;
#pragma unroll
        for (int p = 0; p < P; ++p) cc[0][p] = q.c_zero ? 0.0 : q.c_cur[p * N64 + ii];
    }
    if (NOBASE && (q.c_zero || q.a_def != 0.0)) {
        // queued for a state it does not find (the host's bookkeeping and the device's disagree):
        // nothing of this pass may look like a result -- every sum it leaves is NaN, the decision
        // behind it gives the sweep back to the host, which reports the failure
#pragma unroll
        for (int c = 0; c < NS; ++c)
#pragma unroll
            for (int p = 0; p < P; ++p) cc[c][p] = __builtin_nan("");
    }

    const const_tab prec_tab = as_table(a.prec);
    const const_tab lh_tab = as_table(a.lh);          // one annotation: the row is wave-uniform
    // tables of a whole batch are fetched up front too (P <= 2: 5 doubles per component fit the
    // scalar registers), so the batch computes without a scalar-load stall per component
    constexpr bool TAB_AHEAD = P <= 2;

    // Running sums per candidate, e = the softmax term of a component: Z = sum e, Sq = sum e quad,
    // Sm = sum e mu, Smm = sum e mu^2, Ssig = sum e Sig_pp.  Everything the objective needs is
    // linear in these (see below the loop), so tr(Prec Sig) and mu^T Prec mu are never formed per
    // component: 10 operations per (component, candidate) where the direct sums took 17.
    double shift[NS], Z[NS], Sq[NS], amax[NS], Sm[NS][P], Smm[NS][P], Ssig[NS][P];
#pragma unroll
    for (int c = 0; c < NS; ++c) shift[c] = s0;

    // Double buffering: the loads of batch b+1 are issued BEFORE batch b is folded in and its new
    // vi_mu stored.  On gfx9 loads and stores retire through one in-order counter (vmcnt), so a wave that
    // stores and then loads waits for its own stores to reach HBM before it sees the loaded data;
    // with the next loads ahead of the stores it only ever waits for loads.  With several
    // annotations the log-weight row differs per lane: those (vector) loads travel with the batch.
    // `maxonly`: form the logits a_k only and keep their maximum (the rare second look at a tile
    // whose normaliser left the representable range, see below)
    auto fold = [&](const double (&mul)[KB][P], const double (&lhv)[KB], int k0, auto maxonly) {
        constexpr bool MAXONLY = decltype(maxonly)::value;
        double carry[NS][P];        // (odd P, paired rows: the even component of a pair waiting to be stored)
        (void)carry;
#pragma unroll
        for (int t0 = 0; t0 < KB; t0 += TB) {
        if (k0 + t0 >= kend) break;                    // wave-uniform
        double prt[TAB_AHEAD ? TB : 1][P][P], lht[TAB_AHEAD ? TB : 1];
        if (TAB_AHEAD) {
#pragma unroll
            for (int tt = 0; tt < TB; ++tt) {
                if (t0 + tt >= KB) break;
                const int kc = min(k0 + t0 + tt, M - 1);
#pragma unroll
                for (int e = 0; e < P * P; ++e) prt[tt][e / P][e % P] = prec_tab[(int64_t)kc * P * P + e];
                lht[tt] = ONE_ANNOT ? lh_tab[kc] : lhv[t0 + tt];
            }
        }
#pragma unroll
        for (int tt = 0; tt < TB; ++tt) {
            const int kk = t0 + tt;
            if (kk >= KB) break;
            const int k = k0 + kk;
            if (k >= kend) break;                      // wave-uniform
            double pr[P][P], lam[P][P], sig[P][P], told[P];
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int q = 0; q < P; ++q) {
                    pr[p][q] = TAB_AHEAD ? prt[tt][p][q] : prec_tab[(int64_t)k * P * P + p * P + q];
                    lam[p][q] = pr[p][q];
                }
                lam[p][p] += d[p];
            }
            const double lhk = TAB_AHEAD ? lht[tt] : (ONE_ANNOT ? lh_tab[k] : lhv[kk]);
            const double wk = spd_inverse<P>(lam, sig);
#pragma unroll
            for (int p = 0; p < P; ++p) {              // Lam_k mu_k: the old natural parameter
                double t = 0.0;
                if constexpr (!NOBASE) {
#pragma unroll
                    for (int q = 0; q < P; ++q) t += lam[p][q] * mul[kk][q];
                }
                told[p] = t;
            }
#pragma unroll
            for (int c = 0; c < NS; ++c) {
                double nat[P], mun[P];
#pragma unroll
                for (int p = 0; p < P; ++p)
                    nat[p] = NOBASE ? cc[c][p]                              // (a == 0: Lam Sig cc)
                             : NOSTORE ? (ac[c] * told[p] + cc[c][p])      // Lam (ac mu + Sig cc)
                             : !BLEND ? told[p]
                                      : (step[c] * g[p] + (1.0 - step[c]) * told[p]);
                double quad = 0.0;
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    double t = mul[kk][p];
                    if (BLEND || NOSTORE) {
                        t = (NOSTORE && !NOBASE) ? ac[c] * mul[kk][p] : 0.0;
#pragma unroll
                        for (int q = 0; q < P; ++q) t += sig[p][q] * (NOSTORE ? cc[c][q] : nat[q]);
#if !MU_PAIRED
                        if (!NOSTORE && !MAXONLY) {
#ifndef SNP_DIAG_NOSTORE                 // (diagnostic builds of profiles/microbench_snp.py only)
                            // no `if (live)`: a lane past the end works on a copy of SNP N-1 and
                            // stores the value that SNP's own lane stores, to the same place --
                            // cheaper than four exec-mask branches per component (C3 trial pass
                            // 0.48 -> 0.44 ms)
                            MU_STORE(&mu_out[c][mu_base + MU_ROW(k * P + p, N64)], t);
#endif
                        }
#endif
                    }
                    mun[p] = t;
                    quad += t * nat[p];
                }
#if MU_PAIRED && !defined(SNP_DIAG_NOSTORE)
                if (BLEND && !NOSTORE && !MAXONLY) {
                    // 16 bytes per lane and store: a component's rows in pairs (no `if (live)`, as above)
                    if constexpr (P % 2 == 0) {
#pragma unroll
                        for (int t = 0; t < P / 2; ++t)
                            MU_STORE2(&mu_out[c][mu_base + MU_PAIR(((k * P) >> 1) + t)], mun[2 * t], mun[2 * t + 1]);
                    } else if ((kk & 1) == 0) {
                        // an odd number of cohorts: the rows of components (k, k + 1), k even, pair up;
                        // the even one waits for its partner -- or, as the mixture's last component, goes
                        // out with the buffer's padding row
#pragma unroll
                        for (int p = 0; p < P; ++p) carry[c][p] = mun[p];
                        if (k + 1 >= kend) {                  // wave-uniform; then k + 1 == M
#pragma unroll
                            for (int t = 0; t < (P + 1) / 2; ++t)
                                MU_STORE2(&mu_out[c][mu_base + MU_PAIR(((k * P) >> 1) + t)], carry[c][2 * t],
                                          2 * t + 1 < P ? carry[c][(2 * t + 1) % P] : 0.0);
                        }
                    } else {
#pragma unroll
                        for (int t = 0; t < P; ++t) {
                            const double lo = 2 * t < P ? carry[c][(2 * t) % P] : mun[(2 * t - P) % P];
                            const double hi = 2 * t + 1 < P ? carry[c][(2 * t + 1) % P] : mun[(2 * t + 1 - P) % P];
                            MU_STORE2(&mu_out[c][mu_base + MU_PAIR((((k - 1) * P) >> 1) + t)], lo, hi);
                        }
                    }
                }
#endif
                const double ak = 0.5 * quad + lhk;
                if (MAXONLY) {
                    amax[c] = fmax(amax[c], ak);
                    continue;
                }
                const double e = wk * pass_exp(ak - shift[c], expk);
                Z[c] += e;
                Sq[c] = fma(e, quad, Sq[c]);
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    Sm[c][p] = fma(e, mun[p], Sm[c][p]);
                    Smm[c][p] = fma(e, mun[p] * mun[p], Smm[c][p]);
                    Ssig[c][p] = fma(e, sig[p][p], Ssig[c][p]);
                }
                if (STASH) stash[((k - kbeg) * NS + c) * SNP_THREADS + threadIdx.x] = e;
            }
        }
        }
    };

    const std::false_type all_sums{};
    const std::true_type logit_max{};
    double invZ[NS], Zt[NS];
    for (int attempt = 0;; ++attempt) {
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            Z[c] = 0.0; Sq[c] = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) { Sm[c][p] = 0.0; Smm[c][p] = 0.0; Ssig[c][p] = 0.0; }
        }
        if (!FETCH_FIRST || attempt > 0) fetch(bufA, lhA, kbeg);
        SNP_STAMP(1);
        if constexpr (DEEP) {
            // a plain evaluation only reads: TWO batches requested ahead of the one being folded in
            // (three buffers in rotation).  With one ahead a wave has 2 - 4 KiB in flight, 20 waves a
            // CU 40 - 80 KiB: short of what 8 TB/s times the loaded latency asks of a CU.
            fetch(bufB, lhB, kbeg + KB);
            for (int k0 = kbeg; k0 < kend; k0 += 3 * KB) {
                fetch(bufC, lhC, k0 + 2 * KB);
                fold(bufA, lhA, k0, all_sums);
                if (k0 + KB >= kend) break;        // wave-uniform
                fetch(bufA, lhA, k0 + 3 * KB);
                fold(bufB, lhB, k0 + KB, all_sums);
                if (k0 + 2 * KB >= kend) break;
                fetch(bufB, lhB, k0 + 4 * KB);
                fold(bufC, lhC, k0 + 2 * KB, all_sums);
            }
        } else {
        for (int k0 = kbeg; k0 < kend; k0 += 2 * KB) {
            fetch(bufB, lhB, k0 + KB);         // past the end the clamped loads re-read component M-1
            fold(bufA, lhA, k0, all_sums);
            if (k0 + KB >= kend) break;        // wave-uniform
            fetch(bufA, lhA, k0 + 2 * KB);
            fold(bufB, lhB, k0 + KB, all_sums);
        }
        }
        SNP_STAMP(2);
        // every wave publishes its part of the normaliser; every wave adds the four parts in wave
        // order, so all of them hold the same Z and take the same decision
        if (SNP_SPLIT == 4) {
#pragma unroll
            for (int c = 0; c < NS; ++c) zx_lds[((w * NS + c) * 2 + 0) * 64 + lane] = Z[c];
            lds_barrier();
        }
        bool bad = false;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double z = SNP_SPLIT == 4 ? zx_lds[((0 * NS + c) * 2 + 0) * 64 + lane] : Z[c];
#pragma unroll
            for (int ww = 1; ww < SNP_SPLIT; ++ww) z += zx_lds[((ww * NS + c) * 2 + 0) * 64 + lane];
            Zt[c] = z;
            bad = bad || !(z >= 1e-150 && z <= 1e150);               // true for NaN too
        }
        // a tile is redone at most once: with the exact maximum as the shift Z is in range unless
        // the problem itself is not finite
        const bool retry = attempt == 0 && __builtin_amdgcn_ballot_w64(bad) != 0;   // the same in all four waves
        if (!retry) break;
        // The largest logit of every candidate, from a pass over the tile that forms no exponentials
        // and stores nothing (the running maximum used to ride along in the main loop: two
        // operations per component and candidate for a tile in a million).
#pragma unroll
        for (int c = 0; c < NS; ++c) amax[c] = NEG_INF;
        fetch(bufA, lhA, kbeg);
        for (int k0 = kbeg; k0 < kend; k0 += 2 * KB) {
            fetch(bufB, lhB, k0 + KB);
            fold(bufA, lhA, k0, logit_max);
            if (k0 + KB >= kend) break;
            fetch(bufA, lhA, k0 + 2 * KB);
            fold(bufB, lhB, k0 + KB, logit_max);
        }
        if (SNP_SPLIT == 4) {
#pragma unroll
            for (int c = 0; c < NS; ++c) zx_lds[((w * NS + c) * 2 + 1) * 64 + lane] = amax[c];
            lds_barrier();      // (also: every wave has read the Z parts, which the next round rewrites)
        }
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double m = SNP_SPLIT == 4 ? zx_lds[((0 * NS + c) * 2 + 1) * 64 + lane] : amax[c];
#pragma unroll
            for (int ww = 1; ww < SNP_SPLIT; ++ww) m = fmax(m, zx_lds[((ww * NS + c) * 2 + 1) * 64 + lane]);
            const bool ok = Zt[c] >= 1e-150 && Zt[c] <= 1e150;
            if (!ok && m > NEG_INF && m < -NEG_INF) shift[c] = m;
        }
    }
    SNP_STAMP(3);
    // the sums the objective is written in, from the linear ones: with Lam = Prec + D (D diagonal)
    // and Sig = Lam^-1,
    //   sum e (quad + tr(Prec Sig)) / 2,  tr(Prec Sig) = P - sum_p d_p Sig_pp
    //   sum e mu^T Prec mu,               mu^T Prec mu = quad - sum_p d_p mu_p^2
    //   sum e (Sig_pp + mu_p^2)
    // (the cancellations cost ~P eps of Z absolute, as they did term by term)
    double Skl[NS], Sip[NS], S2[NS][P];
#pragma unroll
    for (int c = 0; c < NS; ++c) {
        double trs = (double)P * Z[c], ips = Sq[c];
#pragma unroll
        for (int p = 0; p < P; ++p) {
            trs = fma(-d[p], Ssig[c][p], trs);
            ips = fma(-d[p], Smm[c][p], ips);
            S2[c][p] = Smm[c][p] + Ssig[c][p];
        }
        Skl[c] = 0.5 * (Sq[c] + trs);
        Sip[c] = ips;
    }
#pragma unroll
    for (int c = 0; c < NS; ++c) invZ[c] = 1.0 / Zt[c];

    if (STASH) {
        // responsibility sums of the tile: delta_k = max(e_k / Z, 1e-100) (invert_nat_cat_2D's clamp,
        // numerics.py:192-194), summed over the tile's SNPs per annotation, for this wave's
        // components and every candidate; one row of partials per tile:
        // [candidate][tile][annotation][component]
        const int A = a.A;
        const int ann = ONE_ANNOT ? 0 : a.annot[ii];
        const int AM = A * M;
        for (int c = 0; c < NS; ++c) {
            double *prow = a.sum_partials + ((int64_t)c * gridDim.x + blockIdx.x) * AM;
            for (int s0 = 0; s0 < Q; s0 += 8) {
                double e8[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) {
                    const int slot = s0 + u;
                    const bool have = kbeg + slot < kend && slot < Q;
                    const double e = have ? stash[(slot * NS + c) * SNP_THREADS + threadIdx.x] : 0.0;
                    e8[u] = (live && have) ? fmax(e * invZ[c], 1e-100) : 0.0;
                }
                if (ONE_ANNOT) {
                    int row;
                    const double tot = tile_sum8(e8, lane, row);
                    const int k = kbeg + s0 + row;
                    if ((lane & 7) == 0 && s0 + row < Q && k < kend) prow[k] = tot;
                } else {
                    for (int aa = 0; aa < A; ++aa) {
                        double m8[8];
#pragma unroll
                        for (int u = 0; u < 8; ++u) m8[u] = ann == aa ? e8[u] : 0.0;
                        int row;
                        const double tot = tile_sum8(m8, lane, row);
                        const int k = kbeg + s0 + row;
                        if ((lane & 7) == 0 && s0 + row < Q && k < kend) prow[(int64_t)aa * M + k] = tot;
                    }
                }
            }
        }
        lds_barrier();                          // the stash is dead: it carries the hand-over now
    }
    SNP_STAMP(4);
    // waves 1..3 hand their remaining sums to wave 0, which adds them in wave order
    if (SNP_SPLIT == 4 && w > 0) {
        double *dst = hand_lds + (w - 1) * NS * NACC * SNP_TILE + lane;
#pragma unroll
        for (int c = 0; c < NS; ++c) {
            double *dc = dst + c * NACC * SNP_TILE;
            dc[0 * SNP_TILE] = Skl[c]; dc[1 * SNP_TILE] = Sip[c];
#pragma unroll
            for (int p = 0; p < P; ++p) {
                dc[(2 + p) * SNP_TILE] = Sm[c][p];
                dc[(2 + P + p) * SNP_TILE] = S2[c][p];
            }
        }
    }
    if (SNP_SPLIT == 4) lds_barrier();
    SNP_STAMP(5);
    if (SNP_SPLIT != 4 || w == 0) {
        // per-SNP results and the tile's contributions to the objective sums
        const bool owner = live;
        double mpost[P];
#pragma unroll
        for (int c = 0; c < NS; ++c) {
#pragma unroll
            for (int ww = 0; ww < SNP_SPLIT - 1; ++ww) {
                const double *sc = hand_lds + (ww * NS + c) * NACC * SNP_TILE + lane;
                Skl[c] += sc[0 * SNP_TILE]; Sip[c] += sc[1 * SNP_TILE];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    Sm[c][p] += sc[(2 + p) * SNP_TILE];
                    S2[c][p] += sc[(2 + P + p) * SNP_TILE];
                }
            }
            double *m_out = c == 0 ? q.m_out : q.m_out2, *v_out = c == 0 ? q.v_out : q.v_out2;
            double *pool_out = c == 0 ? q.pool_out : q.pool_out2;
            double *lse_out = c == 0 ? q.lse_out : q.lse_out2;
            const double lse = shift[c] + log(Zt[c]);
            double vals[NTP];
#pragma unroll
            for (int t = 0; t < NTP; ++t) vals[t] = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                const double m = Sm[c][p] * invZ[c];
                const double v = S2[c][p] * invZ[c] - m * m;
                if (c == 0) mpost[p] = m;
                if (owner) {
                    m_out[p * N64 + i] = m;
                    v_out[p * N64 + i] = v;
                    pool_out[p * N64 + pos[p]] = m / se[p];
                }
                vals[p] = owner ? m * adj[p] : 0.0;
                vals[P + p] = owner ? sld[p] * v : 0.0;
            }
            if (owner) lse_out[i] = lse;
            vals[2 * P] = owner ? (Skl[c] * invZ[c] - lse) : 0.0;
            vals[2 * P + 1] = owner ? 0.5 * Sip[c] * invZ[c] : 0.0;
            // candidate c's columns sit behind the first candidate's (and the 6 statistics columns)
            double *col = a.partials + (int64_t)c * (NT + 6) * nrows + rowid;
#pragma unroll
            for (int h = 0; h < NTP / 8; ++h) {
                double p8[8];
#pragma unroll
                for (int t = 0; t < 8; ++t) p8[t] = vals[8 * h + t];
                int row;
                const double tot = tile_sum8(p8, lane, row);
                if ((lane & 7) == 0 && 8 * h + row < NT) col[(int64_t)(8 * h + row) * nrows] = tot;
            }
        }
        // Convergence statistics of real_posterior_mean (variational_inference.py:374-382, 292-314)
        // fused into an evaluation the caller accepts unconditionally (the one after the M-step):
        // the new posterior means are compared with the snapshot and become the snapshot, so no
        // separate pass over [P][N] (mean_diff_kernel) and no second stream are needed per sweep.
        if (!BLEND && a.diff) {                       // kernel-uniform
            double dv[6] = {0, 0, 0, 0, 0, 0};
            if (owner) {
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    const double nw = mpost[p] * scal_t[p];
                    const double od = snap_t[p];
                    const double df = fabs(nw - od);
                    dv[0] += (df <= 1e-6 + 1e-6 * fabs(od)) ? 0.0 : 1.0;
                    dv[1] += df;
                    dv[2] += df * df;
                    dv[3] = fmax(dv[3], fabs(nw));
                    dv[4] = fmax(dv[4], df);
                    dv[5] = fmax(dv[5], fabs((nw - od) / (od + 1e-100)));
                    q.snap_out[p * N64 + i] = nw;
                }
            }
#pragma unroll
            for (int q = 0; q < 6; ++q) {
                const double t = q < 3 ? wave_sum(dv[q]) : wave_max(dv[q]);
                if (lane == 0) a.partials[(int64_t)(NT + q) * nrows + rowid] = t;
            }
        }
    }
#if SNP_TRACE
    SNP_STAMP(6);
    if (lane == 0 && g_ld_trace != nullptr && (long long)blockIdx.x * 4 + w < g_ld_trace_cap) {
        double *row = g_ld_trace + 12 * ((long long)blockIdx.x * 4 + w);
        for (int j = 0; j < 7; ++j) row[j] = (double)stamp[j];
        row[7] = (double)(__builtin_amdgcn_s_getreg((20) | (0 << 6) | (3 << 11)) & 0xf);    // HW_REG_XCC_ID[3:0]
        row[8] = (double)real0;
        row[9] = (double)(long long)__builtin_amdgcn_s_memrealtime();
        row[10] = (double)__builtin_amdgcn_s_getreg((4) | (0 << 6) | (31 << 11));           // HW_REG_HW_ID
        row[11] = 0.0;
    }
#endif
}

template <int P, bool BLEND, bool ONE_ANNOT, int NS>
static void launch_snp_pass_s(const SnpKernelArgs &a, bool stash, hipStream_t s) {
    const dim3 grid(SNP_SPLIT == 4 ? snp_tile_grid(a.N) : snp_pass_grid(a.N)), block(SNP_THREADS);
    const size_t lds = snp_pass_lds_bytes(a.M, P, NS, stash);
    if constexpr (P <= 4) {
        if (stash) {
            auto launch = [&](auto kern, bool &raised) {
                if (!raised) {                  // more than 64 KB of dynamic LDS needs the attribute
                    (void)hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize,
                                              96 * 1024);
                    raised = true;
                }
                hipLaunchKernelGGL(kern, grid, block, lds, s, a);
            };
            static bool raised = false, raised_lazy = false, raised_nobase = false;
            if constexpr (BLEND) {
                if constexpr (P <= 2) {
                    if (a.no_store == 2) {      // ... of a state known to have a == 0
                        launch(snp_pass_kernel<P, BLEND, ONE_ANNOT, NS, true, true, true>, raised_nobase);
                        return;
                    }
                }
                if (a.no_store) {       // a lazy trial that keeps the stash
                    launch(snp_pass_kernel<P, BLEND, ONE_ANNOT, NS, true, true>, raised_lazy);
                    return;
                }
            }
            launch(snp_pass_kernel<P, BLEND, ONE_ANNOT, NS, true>, raised);
            return;
        }
    }
    if constexpr (P <= 2) {
        if (a.no_store == 2) {  // a lazy pass of a state known to have a == 0: no vi_mu load compiled in
            hipLaunchKernelGGL((snp_pass_kernel<P, BLEND, ONE_ANNOT, NS, false, true, true>), grid, block, lds, s, a);
            return;
        }
    }
    if constexpr (BLEND || P <= 4) {
        if (a.no_store) {       // lazy trial: no candidate vi_mu stored; lazy evaluation: the state derived
            hipLaunchKernelGGL((snp_pass_kernel<P, BLEND, ONE_ANNOT, NS, false, true>), grid, block, lds, s, a);
            return;
        }
    }
    hipLaunchKernelGGL((snp_pass_kernel<P, BLEND, ONE_ANNOT, NS, false>), grid, block, lds, s, a);
}

template <int P>
static void launch_snp_pass_p(const SnpKernelArgs &a, bool blend, int ns, bool stash, hipStream_t s) {
    const bool one = a.A == 1;
    if (blend && ns == 2) {
        if (one) launch_snp_pass_s<P, true, true, 2>(a, stash, s);
        else launch_snp_pass_s<P, true, false, 2>(a, stash, s);
    } else if (blend) {
        if (one) launch_snp_pass_s<P, true, true, 1>(a, stash, s);
        else launch_snp_pass_s<P, true, false, 1>(a, stash, s);
    } else {
        if (one) launch_snp_pass_s<P, false, true, 1>(a, stash, s);
        else launch_snp_pass_s<P, false, false, 1>(a, stash, s);
    }
}

template <int P>
static void launch_snp_pass_big(const SnpKernelArgs &a, bool blend, hipStream_t s) {
    const bool one = a.A == 1;
    if (blend) {
        if (one) launch_snp_pass_s<P, true, true, 1>(a, false, s);
        else launch_snp_pass_s<P, true, false, 1>(a, false, s);
    } else {
        if (one) launch_snp_pass_s<P, false, true, 1>(a, false, s);
        else launch_snp_pass_s<P, false, false, 1>(a, false, s);
    }
}

// ns = 1: one candidate (or a plain evaluation); ns = 2: a beta trial at a.step and a.step2.
// With a.sum_partials != nullptr (and snp_pass_can_stash) the per-tile responsibility sums of every
// candidate are written there: [candidate][tile][A*M].
void launch_snp_pass(const SnpKernelArgs &args, bool blend, int ns, hipStream_t s) {
    SnpKernelArgs a = args;
    a.pred = g_pred;
    a.pp = g_phase;
    const bool stash = a.sum_partials != nullptr && snp_pass_can_stash(a.M, a.P, ns);
    switch (a.P) {
        case 1: launch_snp_pass_p<1>(a, blend, ns, stash, s); break;
        case 2: launch_snp_pass_p<2>(a, blend, ns, stash, s); break;
        case 3: launch_snp_pass_p<3>(a, blend, ns, stash, s); break;
        case 4: launch_snp_pass_p<4>(a, blend, ns, stash, s); break;
        // five to eight cohorts: the same kernel with the P x P Cholesky unrolled (it spills; the
        // reference's own general branch, numerics.py:238-290, is numpy's inv / slogdet per matrix);
        // one candidate per trial, sums from delta_kernel
        case 5: launch_snp_pass_big<5>(a, blend, s); break;
        case 6: launch_snp_pass_big<6>(a, blend, s); break;
        case 7: launch_snp_pass_big<7>(a, blend, s); break;
        case 8: launch_snp_pass_big<8>(a, blend, s); break;
        default: break;   // rejected in vilma_create
    }
}

// --------------------------------------------------------------------------------------------
// responsibilities of the current state: delta_ik = max(exp(u_ik - lse_i), 1e-100)
// (invert_nat_cat_2D's clamp, numerics.py:192-194)
// --------------------------------------------------------------------------------------------
// KS = lanes that share one SNP (1 or 4).  With KS = 4 the quarter-waves take components
// k = ks, ks+4, ... of the same 16 SNPs: 4x the waves for the same work, which small shards (an
// 8-GPU rank holds ~130 k SNPs = 2 waves per SIMD at KS = 1) need to hide latency, and the
// per-component sum over SNPs becomes a 16-lane shuffle reduction (29 vs 39 us at 131 k SNPs).
// MAT: the state is not stored yet -- it is the candidate a lazy beta trial's decision accepted,
// mu_k' = Sig_k (step g + (1 - step) Lam_k mu_k) in terms of the state the trial started from (the
// very expressions of snp_pass_kernel's blend, so the same bits) -- and this pass stores it on its
// way (the one vi_mu array written per accepted update).
template <int P, bool ONE_ANNOT, bool WRITE, int KS, bool MAT = false>
__global__ __launch_bounds__(SNP_THREADS) void delta_kernel(const DeltaArgs a) {
    constexpr int SPW = 64 / KS;
    PRED_EXIT(a.pred);
    const ExpConsts expk = exp_consts();
    const int N = a.N, M = a.M, A = a.A;
    const int64_t N64 = N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int ks = lane / SPW;
    const int i = (blockIdx.x * (SNP_THREADS / 64) + w) * SPW + (lane % SPW);
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    const int64_t mu_base = MU_BASE(ii, M, P, N64);
    // the state: from the arguments, or -- behind a queued sweep's decision -- the one its EVAL
    // phase starts from (the candidate the decision accepted)
    const double *mu_state = a.mu, *lse_state = a.lse;
    double *mu_mat = a.mu_mat;
    const double *c_state = a.cvec;
    double acoef = a.acoef;
    double d[P];
    if (a.pp != nullptr) {
        const phase_tab t = PHASE(a.pp);
        mu_state = t->mu_in;
        lse_state = t->lse_ref;
        mu_mat = t->mu_mat;
        c_state = t->c_pend;
        acoef = t->a_pend;
#pragma unroll
        for (int p = 0; p < P; ++p) d[p] = a.sld[p * N64 + ii] / t->tau[p];
    } else {
#pragma unroll
        for (int p = 0; p < P; ++p) d[p] = a.sld[p * N64 + ii] / a.tau.v[p];
    }
    // (acoef == 0: the state needs no stored vi_mu -- loads pointed at tile 0, see snp_pass_kernel)
    const int64_t ld_base = (MAT && acoef == 0.0) ? MU_BASE(ii & (MU_TILE - 1), M, P, N64) : mu_base;
    double cv[P];            // MAT: the state is acoef (stored vi_mu) + Sig cv (PhasePtrs)
#pragma unroll
    for (int p = 0; p < P; ++p) cv[p] = MAT ? c_state[p * N64 + ii] : 0.0;
    const int ann = ONE_ANNOT ? 0 : a.annot[ii];
    const double *lh = a.lh + (int64_t)ann * M;
    const double lse = lse_state[ii];
    double *prow = WRITE ? nullptr : a.out + ((int64_t)blockIdx.x * (SNP_THREADS / 64) + w) * A * M;
    // with KS = 4 the quarter-waves take components k = ks, ks+4, ... of the same 16 SNPs; the
    // per-component sum over SNPs is then a 16-lane (segmented) shuffle reduction
    // KS == 1: k is wave-uniform, the tables go through the constant address space (scalar loads);
    // KS == 4: the quarter-waves read different components, plain vector loads.
    // Components go in batches of KD: all vi_mu loads of a batch first, then the arithmetic, then
    // the (few) stores together, so a wave waits on its own stores once per batch, not per component.
    constexpr int KD = 4;
    const const_tab prec_tab = as_table(a.prec);
    const const_tab lh_tab = as_table(a.lh);
    // Double buffering as in snp_pass_kernel: the loads of the next batch are issued BEFORE this
    // batch's stores (on gfx9 loads and stores retire through one in-order counter, so a wave that
    // stores and then loads waits for its own stores to reach memory before it sees the loaded data)
    // vi_mu in row pairs (MU_TILED 2): 16 bytes per lane and access where the rows a lane works on are
    // whole pairs -- a batch of KD consecutive components (KS == 1: k0 is a multiple of KD), or the
    // rows of one component with an even number of cohorts
    constexpr bool PAIR_BATCH = MU_PAIRED && KS == 1 && (KD * P) % 2 == 0 && KD % 2 == 0;
    constexpr bool PAIR_COMP = MU_PAIRED && !PAIR_BATCH && P % 2 == 0;
    auto fetch = [&](double (&dst)[KD][P], int k0) {
#if MU_PAIRED
        if constexpr (PAIR_BATCH) {
            const int j0 = (k0 * P) >> 1, jmax = (int)MU_PAIRS(M, P) - 1;
#pragma unroll
            for (int t = 0; t < KD * P / 2; ++t) {
                const v2d v = MU_LOAD2(&mu_state[ld_base + MU_PAIR(min(j0 + t, jmax))]);
                dst[(2 * t) / P][(2 * t) % P] = v.x;
                dst[(2 * t + 1) / P][(2 * t + 1) % P] = v.y;
            }
            return;
        } else if constexpr (PAIR_COMP) {
#pragma unroll
            for (int u = 0; u < KD; ++u) {
                const int kc = min(k0 + u * KS, M - 1);
#pragma unroll
                for (int t = 0; t < P / 2; ++t) {
                    const v2d v = MU_LOAD2(&mu_state[ld_base + MU_PAIR(((kc * P) >> 1) + t)]);
                    dst[u][2 * t] = v.x;
                    dst[u][2 * t + 1] = v.y;
                }
            }
            return;
        }
#endif
#pragma unroll
        for (int u = 0; u < KD; ++u) {
            const int kc = min(k0 + u * KS, M - 1);       // unconditional loads; extras ignored
#pragma unroll
            for (int p = 0; p < P; ++p) dst[u][p] = MU_LOAD(&mu_state[ld_base + MU_ROW(kc * P + p, N64)]);
        }
    };
    auto work = [&](double (&mu)[KD][P], int k0) {
        double delta[KD];
#pragma unroll
        for (int u = 0; u < KD; ++u) {
            const int k = min(k0 + u * KS, M - 1);
            double lam[P][P];
            const double *pk = a.prec + (int64_t)k * P * P;
#pragma unroll
            for (int p = 0; p < P; ++p) {
#pragma unroll
                for (int q = 0; q < P; ++q)
                    lam[p][q] = KS == 1 ? prec_tab[(int64_t)k * P * P + p * P + q] : pk[p * P + q];
                lam[p][p] += d[p];
            }
            const double lhk = (KS == 1 && ONE_ANNOT) ? lh_tab[k] : lh[k];
            double wdet;
            if (MAT) {
                double sig[P][P], base[P];
                wdet = spd_inverse<P>(lam, sig);
#pragma unroll
                for (int p = 0; p < P; ++p) base[p] = mu[u][p];
#pragma unroll
                for (int p = 0; p < P; ++p) {
                    double t = acoef * base[p];
#pragma unroll
                    for (int q = 0; q < P; ++q) t += sig[p][q] * cv[q];
                    mu[u][p] = t;
                    if constexpr (!PAIR_BATCH && !PAIR_COMP) {
                        if (mu_mat != nullptr) MU_STORE(&mu_mat[mu_base + MU_ROW(k * P + p, N64)], t);
                    }
                }
#if MU_PAIRED
                if constexpr (PAIR_COMP) {
                    if (mu_mat != nullptr) {
#pragma unroll
                        for (int t = 0; t < P / 2; ++t)
                            MU_STORE2(&mu_mat[mu_base + MU_PAIR(((k * P) >> 1) + t)], mu[u][2 * t], mu[u][2 * t + 1]);
                    }
                }
#endif
            } else {
                wdet = spd_rsqrt_det<P>(lam);
            }
            double quad = 0.0;
#pragma unroll
            for (int p = 0; p < P; ++p) {
                double t = 0.0;
#pragma unroll
                for (int q = 0; q < P; ++q) t += lam[p][q] * mu[u][q];
                quad += mu[u][p] * t;
            }
            delta[u] = fmax(wdet * pass_exp(0.5 * quad + lhk - lse, expk), 1e-100);
        }
#if MU_PAIRED
        // (mu_mat == nullptr, uniform over the launch: a persistent lazy state -- the pass derives the
        // state for its sums and stores nothing)
        if constexpr (MAT && PAIR_BATCH) if (mu_mat != nullptr) {
            // the batch's new rows, pair by pair (a pair whose first row lies beyond the mixture was
            // computed from a clamped load: not stored; one that ends in the padding row stores zero there)
            const int j0 = (k0 * P) >> 1, MP = M * P;
#pragma unroll
            for (int t = 0; t < KD * P / 2; ++t) {
                if (2 * (j0 + t) >= MP) break;              // uniform over the launch
                const double lo = mu[(2 * t) / P][(2 * t) % P];
                const double hi = 2 * (j0 + t) + 1 < MP ? mu[(2 * t + 1) / P][(2 * t + 1) % P] : 0.0;
                MU_STORE2(&mu_mat[mu_base + MU_PAIR(j0 + t)], lo, hi);
            }
        }
#endif
        if (WRITE) {
#pragma unroll
            for (int u = 0; u < KD; ++u) {
                const int k = k0 + u * KS;
                if (live && k < M) a.out[(int64_t)k * N64 + i] = delta[u];
            }
        } else if (ONE_ANNOT) {
            double sums[KD];
#pragma unroll
            for (int u = 0; u < KD; ++u) sums[u] = seg_sum<SPW>(live ? delta[u] : 0.0);
#pragma unroll
            for (int u = 0; u < KD; ++u) {
                const int k = k0 + u * KS;
                if ((lane % SPW) == 0 && k < M) prow[k] = sums[u];
            }
        } else {
            for (int aa = 0; aa < A; ++aa) {
                double sums[KD];
#pragma unroll
                for (int u = 0; u < KD; ++u)
                    sums[u] = seg_sum<SPW>((live && ann == aa) ? delta[u] : 0.0);
#pragma unroll
                for (int u = 0; u < KD; ++u) {
                    const int k = k0 + u * KS;
                    if ((lane % SPW) == 0 && k < M) prow[(int64_t)aa * M + k] = sums[u];
                }
            }
        }
    };
    double bufA[KD][P], bufB[KD][P];
    fetch(bufA, ks);
    for (int k0 = ks; k0 < M; k0 += 2 * KS * KD) {
        fetch(bufB, k0 + KS * KD);          // (past the end the clamped loads re-read component M-1)
        work(bufA, k0);
        if (k0 + KS * KD >= M) break;
        fetch(bufA, k0 + 2 * KS * KD);
        work(bufB, k0 + KS * KD);
    }
}

#ifndef KS_SPLIT_BELOW
#define KS_SPLIT_BELOW 400000       // shards below this many SNPs use 4 lanes per SNP
#endif
static inline int delta_ks(int64_t N) { return N < KS_SPLIT_BELOW ? 4 : 1; }

int delta_grid(int64_t N) {
    const int64_t per_block = SNP_THREADS / delta_ks(N);
    return (int)((N + per_block - 1) / per_block);
}

// out[chunk][c] = sum over the chunk's rows of in[r][c]: lanes along columns, the 4 waves take
// interleaved rows with 4 independent accumulators each, fixed combination order.  Launched
// twice (rows -> RC_CHUNKS partial rows -> 1 row) so the long reduction is spread over the chip.
#define RC_CHUNK_ROWS 256
// blockIdx.z = independent problem (candidate): its input / output sit in_zstride / out_zstride
// doubles behind the first one's.
__global__ __launch_bounds__(256) void reduce_cols_kernel(const double *__restrict__ in, int rows,
                                                           int ncols, double *__restrict__ out,
                                                           int64_t in_zstride, int64_t out_zstride,
                                                           const int *pred) {
    __shared__ double red[4][64];
    PRED_EXIT(pred);
    in += blockIdx.z * in_zstride;
    out += blockIdx.z * out_zstride;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * RC_CHUNK_ROWS;
    const int r1 = min(rows, r0 + RC_CHUNK_ROWS);
    double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
    if (c < ncols) {
        int r = r0 + w;
        for (; r + 12 < r1; r += 16) {
            s0 += in[(int64_t)r * ncols + c];
            s1 += in[(int64_t)(r + 4) * ncols + c];
            s2 += in[(int64_t)(r + 8) * ncols + c];
            s3 += in[(int64_t)(r + 12) * ncols + c];
        }
        for (; r < r1; r += 4) s0 += in[(int64_t)r * ncols + c];
    }
    red[w][lane] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (w == 0 && c < ncols)
        out[(int64_t)blockIdx.y * ncols + c] = (red[0][lane] + red[1][lane]) + (red[2][lane] + red[3][lane]);
}

// rows of scratch reduce_cols needs behind `rows` partial rows: the chunk rows of every pass but
// the last (which writes the result)
int64_t reduce_cols_scratch_rows(int64_t rows) {
    int64_t total = 0;
    while (rows > RC_CHUNK_ROWS) {
        rows = (rows + RC_CHUNK_ROWS - 1) / RC_CHUNK_ROWS;
        total += rows;
    }
    return total;
}

int64_t delta_partial_rows(int64_t N) {
    const int64_t rows = (int64_t)delta_grid(N) * (SNP_THREADS / 64);
    return rows + reduce_cols_scratch_rows(rows);
}

// nz independent problems in one launch sequence: problem z reads in + z * in_zstride and writes
// out + z * out_zstride; scratch must hold nz * reduce_cols_scratch_rows(rows) * ncols doubles
static void reduce_cols(const double *in, int rows, int ncols, double *scratch, double *out,
                        hipStream_t s, int nz = 1, int64_t in_zstride = 0, int64_t out_zstride = 0) {
    const int colblocks = (ncols + 63) / 64;
    const int64_t sz = reduce_cols_scratch_rows(rows) * ncols;      // per-problem scratch
    while (rows > RC_CHUNK_ROWS) {
        const int chunks = (rows + RC_CHUNK_ROWS - 1) / RC_CHUNK_ROWS;
        hipLaunchKernelGGL(reduce_cols_kernel, dim3(colblocks, chunks, nz), dim3(256), 0, s, in, rows,
                           ncols, scratch, in_zstride, sz, g_pred);
        in = scratch;
        in_zstride = sz;
        scratch = scratch + (int64_t)chunks * ncols;
        rows = chunks;
    }
    hipLaunchKernelGGL(reduce_cols_kernel, dim3(colblocks, 1, nz), dim3(256), 0, s, in, rows, ncols,
                       out, in_zstride, out_zstride, g_pred);
}

template <int P, bool WRITE, int KS>
static void launch_delta_pk(const DeltaArgs &a, hipStream_t s) {
    const dim3 grid(delta_grid(a.N)), block(SNP_THREADS);
    if constexpr (!WRITE) {
        if (a.mat) {
            if (a.A == 1) hipLaunchKernelGGL((delta_kernel<P, true, false, KS, true>), grid, block, 0, s, a);
            else hipLaunchKernelGGL((delta_kernel<P, false, false, KS, true>), grid, block, 0, s, a);
            return;
        }
    }
    if (a.A == 1) hipLaunchKernelGGL((delta_kernel<P, true, WRITE, KS>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((delta_kernel<P, false, WRITE, KS>), grid, block, 0, s, a);
}

template <int P, bool WRITE>
static void launch_delta_p(const DeltaArgs &a, hipStream_t s) {
    if (delta_ks(a.N) == 4) launch_delta_pk<P, WRITE, 4>(a, s);
    else launch_delta_pk<P, WRITE, 1>(a, s);
}

template <bool WRITE>
static void launch_delta_any(const DeltaArgs &args, hipStream_t s) {
    DeltaArgs a = args;
    a.pred = g_pred;
    a.pp = g_phase;
    switch (a.P) {
        case 1: launch_delta_p<1, WRITE>(a, s); break;
        case 2: launch_delta_p<2, WRITE>(a, s); break;
        case 3: launch_delta_p<3, WRITE>(a, s); break;
        case 4: launch_delta_p<4, WRITE>(a, s); break;
        case 5: launch_delta_p<5, WRITE>(a, s); break;
        case 6: launch_delta_p<6, WRITE>(a, s); break;
        case 7: launch_delta_p<7, WRITE>(a, s); break;
        case 8: launch_delta_p<8, WRITE>(a, s); break;
        default: break;
    }
}

void launch_delta_sums(const DeltaArgs &a, double *sums_out, hipStream_t s) {
    launch_delta_any<false>(a, s);
    const int ncols = a.A * a.M;
    const int rows = delta_grid(a.N) * (SNP_THREADS / 64);
    reduce_cols(a.out, rows, ncols, a.out + (int64_t)rows * ncols, sums_out, s);
}

void launch_delta_write(const DeltaArgs &a, hipStream_t s) { launch_delta_any<true>(a, s); }

// Responsibility sums from the partial rows a stashing snp_pass left behind ([candidate][row][AM],
// launch_snp_pass): one workgroup per (column, candidate) adds the rows in a fixed order.  The rows
// were written a moment ago by the pass; a column is rows x 8 B spread over rows lines.
static __device__ __forceinline__ double block_column_sum(const double *__restrict__ src, int rows,
                                                          int64_t stride, double *sh);
__global__ __launch_bounds__(256) void tile_sums_kernel(const double *__restrict__ in, int rows, int AM,
                                                         double *__restrict__ out, int64_t out_zstride,
                                                         const int *pred) {
    __shared__ double sh[4];
    PRED_EXIT(pred);
    const int col = blockIdx.x, z = blockIdx.y;
    const double tot = block_column_sum(in + (int64_t)z * rows * AM + col, rows, AM, sh);
    if (threadIdx.x == 0) out[(int64_t)z * out_zstride + col] = tot;
}
int64_t tile_sums_elems(int64_t N, int A, int M, int ns) {
    return (int64_t)ns * snp_sum_rows(N, A) * A * M;
}
void launch_tile_sums(const double *rows_dev, int64_t N, int A, int M, int ns, double *out,
                      int64_t out_zstride, hipStream_t s) {
    hipLaunchKernelGGL(tile_sums_kernel, dim3(A * M, ns), dim3(256), 0, s, rows_dev,
                       snp_sum_rows(N, A), A * M, out, out_zstride, g_pred);
}

// --------------------------------------------------------------------------------------------
// _initialize on the device (variational_inference.py:658-692): from the jittered start fake_mu
// [P][N] (drawn on the host with the reference's legacy RNG call) the heuristic responsibilities
//   delta_ik ~ exp(-0.5 (1.6^2 f^T Prec_k f + tr(Prec_k Sig_ki) - log_det_k)),  clamped at 1e-100,
// their per-annotation sums (-> hyper_delta on the host), avg_i = sum_k delta_ik Sig_ki and
//   vi_mu_ki = Sig_ki avg_i^-1 f_i
// written straight into the vi_mu buffer: no [M,P,N] array is ever built on the host or crosses
// PCIe.  Thread per SNP; two passes over the components (normaliser, then outputs), everything
// else in registers.  Partial sums per wave row, reduced by reduce_cols like delta_kernel's.
// --------------------------------------------------------------------------------------------
template <int P, bool ONE_ANNOT>
__global__ __launch_bounds__(SNP_THREADS) void init_state_kernel(const InitArgs a) {
    const int N = a.N, M = a.M, A = a.A;
    const int64_t N64 = N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * SNP_THREADS + threadIdx.x;
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    double d[P], f[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        d[p] = a.sld[p * N64 + ii] / a.tau.v[p];
        f[p] = a.fake_mu[p * N64 + ii];
    }
    const int ann = ONE_ANNOT ? 0 : a.annot[ii];
    const const_tab prec_tab = as_table(a.prec);
    const const_tab ld_tab = as_table(a.log_det);
    // pass 1: x_k = -0.5 probs_k, running maximum, Z = sum exp(x_k - max), avg ~ sum e_k Sig_k
    auto half_neg_probs = [&](int k, double (&sig)[P][P]) {
        double pr[P][P], lam[P][P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                pr[p][q] = prec_tab[(int64_t)k * P * P + p * P + q];
                lam[p][q] = pr[p][q];
            }
            lam[p][p] += d[p];
        }
        spd_inverse<P>(lam, sig);
        double quad = 0.0, tr = (double)P;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < P; ++q) t += pr[p][q] * (1.6 * f[q]);
            quad += (1.6 * f[p]) * t;
            tr = fma(-d[p], sig[p][p], tr);          // tr(Prec Sig) = P - tr(D Sig)
        }
        return -0.5 * (quad + tr - ld_tab[k]);
    };
    double mx = NEG_INF, Z = 0.0, avg[P][P];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < P; ++q) avg[p][q] = 0.0;
    for (int k = 0; k < M; ++k) {
        double sig[P][P];
        const double x = half_neg_probs(k, sig);
        const double dk = x - mx;
        const double t = exp(-fabs(dk));
        const bool up = dk > 0.0;
        const double sc = up ? t : 1.0, e = up ? 1.0 : t;
        mx = up ? x : mx;
        Z = fma(Z, sc, e);
#pragma unroll
        for (int p = 0; p < P; ++p)
#pragma unroll
            for (int q = 0; q < P; ++q) avg[p][q] = fma(avg[p][q], sc, e * sig[p][q]);
    }
    const double invZ = 1.0 / Z;
    double avgn[P][P], iavg[P][P], nat[P];
#pragma unroll
    for (int p = 0; p < P; ++p)
#pragma unroll
        for (int q = 0; q < P; ++q) avgn[p][q] = avg[p][q] * invZ;
    spd_inverse<P>(avgn, iavg);
#pragma unroll
    for (int p = 0; p < P; ++p) {
        double t = 0.0;
#pragma unroll
        for (int q = 0; q < P; ++q) t += iavg[p][q] * f[q];
        nat[p] = t;
        if (a.c_out != nullptr && live) a.c_out[p * N64 + i] = t;
    }
    // pass 2: vi_mu_k = Sig_k nat and the responsibility sums
    double *prow = a.partials + ((int64_t)blockIdx.x * (SNP_THREADS / 64) + w) * A * M;
    for (int k = 0; k < M; ++k) {
        double sig[P][P];
        const double x = half_neg_probs(k, sig);
        const double delta = fmax(exp(x - mx) * invZ, 1e-100);
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double t = 0.0;
#pragma unroll
            for (int q = 0; q < P; ++q) t += sig[p][q] * nat[q];
            if (live) a.mu_out[MU_BASE(i, M, P, N64) + MU_ROW(k * P + p, N64)] = t;
        }
        if (ONE_ANNOT) {
            const double s = wave_sum(live ? delta : 0.0);
            if (lane == 0) prow[k] = s;
        } else {
            for (int aa = 0; aa < A; ++aa) {
                const double s = wave_sum((live && ann == aa) ? delta : 0.0);
                if (lane == 0) prow[(int64_t)aa * M + k] = s;
            }
        }
    }
}

int64_t init_partial_rows(int64_t N) {
    const int64_t rows = (int64_t)snp_pass_grid(N) * (SNP_THREADS / 64);
    return rows + reduce_cols_scratch_rows(rows);
}

template <int P>
static void launch_init_p(const InitArgs &a, hipStream_t s) {
    const dim3 grid(snp_pass_grid(a.N)), block(SNP_THREADS);
    if (a.A == 1) hipLaunchKernelGGL((init_state_kernel<P, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((init_state_kernel<P, false>), grid, block, 0, s, a);
}

void launch_init_state(const InitArgs &a, double *sums_out, hipStream_t s) {
    switch (a.P) {
        case 1: launch_init_p<1>(a, s); break;
        case 2: launch_init_p<2>(a, s); break;
        case 3: launch_init_p<3>(a, s); break;
        case 4: launch_init_p<4>(a, s); break;
        case 5: launch_init_p<5>(a, s); break;
        case 6: launch_init_p<6>(a, s); break;
        case 7: launch_init_p<7>(a, s); break;
        case 8: launch_init_p<8>(a, s); break;
        default: break;
    }
    const int ncols = a.A * a.M;
    const int rows = snp_pass_grid(a.N) * (SNP_THREADS / 64);
    reduce_cols(a.partials, rows, ncols, a.partials + (int64_t)rows * ncols, sums_out, s);
}

// --------------------------------------------------------------------------------------------
// Objective pieces of (current vi_mu, a vi_delta GIVEN by the caller, current hyper / tau): what
// elbo(params), real_posterior_mean/variance compute in the reference when they are handed a
// vi_delta that is not the coordinate-ascent fixed point of (vi_mu, hyper_delta, error_scaling)
// (variational_inference.py:412-417, 740-760, 873-885; numerics.py:49-65, 98-146).  Off the sweep
// path: logs per (component, SNP), no cancellation tricks.  delta comes component-major [M][N].
// Also reports max |delta_given - delta_derived| (derived from the lse of the accepted state).
// --------------------------------------------------------------------------------------------
template <int P, bool ONE_ANNOT>
__global__ __launch_bounds__(SNP_THREADS) void snp_given_delta_kernel(const SnpKernelArgs a,
                                                                      const double *__restrict__ delta_km,
                                                                      const double *__restrict__ lse_cur,
                                                                      double *__restrict__ maxdev_partials) {
    constexpr int NT = 2 * P + 2;
    __shared__ double red[SNP_THREADS / 64][NT + 1];
    const int N = a.N, M = a.M;
    const int64_t N64 = N;
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int i = blockIdx.x * SNP_THREADS + threadIdx.x;
    const bool live = i < N;
    const int ii = live ? i : N - 1;
    double d[P], se[P], adj[P], sld[P];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        se[p] = a.se[p * N64 + ii];
        adj[p] = a.adj[p * N64 + ii];
        sld[p] = a.sld[p * N64 + ii];
        d[p] = sld[p] / a.tau.v[p];
    }
    const double *lh = a.lh + (ONE_ANNOT ? 0 : (int64_t)a.annot[ii] * M);
    const double lse = lse_cur[ii];
    double Skl = 0.0, Sip = 0.0, dev = 0.0, Sm[P], S2[P];
#pragma unroll
    for (int p = 0; p < P; ++p) { Sm[p] = 0.0; S2[p] = 0.0; }
    for (int k = 0; k < M; ++k) {
        double pr[P][P], lam[P][P], sig[P][P], mu[P];
#pragma unroll
        for (int p = 0; p < P; ++p) {
#pragma unroll
            for (int q = 0; q < P; ++q) {
                pr[p][q] = a.prec[(int64_t)k * P * P + p * P + q];
                lam[p][q] = pr[p][q];
            }
            lam[p][p] += d[p];
            mu[p] = a.mu_in[MU_BASE(ii, a.M, P, N64) + MU_ROW(k * P + p, N64)];
        }
        const double wk = spd_inverse<P>(lam, sig);            // det(lam)^-1/2
        const double dl = delta_km[(int64_t)k * N64 + ii];
        double quad = 0.0, ip = 0.0, tr = 0.0;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            double t = 0.0, u = 0.0;
#pragma unroll
            for (int q = 0; q < P; ++q) {
                t += lam[p][q] * mu[q];
                u += pr[p][q] * mu[q];
                tr += pr[p][q] * sig[q][p];
            }
            quad += mu[p] * t;
            ip += mu[p] * u;
        }
        // log h_k = lh_k + 0.5 log_det_k;  log det Sig_ki = 2 log wk
        const double log_h = lh[k] + 0.5 * a.log_det[k];
        const double sigma_summary = a.log_det[k] - 2.0 * log(wk) + tr;
        Skl += dl * (log(dl) - log_h) + 0.5 * sigma_summary * dl;
        Sip += 0.5 * dl * ip;
#pragma unroll
        for (int p = 0; p < P; ++p) {
            Sm[p] = fma(dl, mu[p], Sm[p]);
            S2[p] = fma(dl, sig[p][p] + mu[p] * mu[p], S2[p]);
        }
        const double derived = fmax(wk * exp(0.5 * quad + lh[k] - lse), 1e-100);
        dev = fmax(dev, fabs(dl - derived));
    }
    double part[NT + 1];
#pragma unroll
    for (int p = 0; p < P; ++p) {
        const double m = Sm[p];
        const double v = S2[p] - m * m;
        if (live) {
            a.m_out[p * N64 + i] = m;
            a.v_out[p * N64 + i] = v;
            a.pool_out[p * N64 + a.invperm[p * N64 + i]] = m / se[p];
        }
        part[p] = live ? m * adj[p] : 0.0;
        part[P + p] = live ? sld[p] * v : 0.0;
    }
    part[2 * P] = live ? Skl : 0.0;
    part[2 * P + 1] = live ? Sip : 0.0;
    part[NT] = live ? dev : 0.0;
#pragma unroll
    for (int t = 0; t <= NT; ++t) {
        const double s = t < NT ? wave_sum(part[t]) : wave_max(part[t]);
        if (lane == 0) red[w][t] = s;
    }
    __syncthreads();
    if (threadIdx.x <= NT) {
        const int t = threadIdx.x;
        double s = red[0][t];
#pragma unroll
        for (int ww = 1; ww < SNP_THREADS / 64; ++ww) s = t < NT ? s + red[ww][t] : fmax(s, red[ww][t]);
        if (t < NT) a.partials[(int64_t)t * gridDim.x + blockIdx.x] = s;
        else maxdev_partials[blockIdx.x] = s;
    }
}

__global__ __launch_bounds__(256) void max_reduce_kernel(const double *__restrict__ v, int n,
                                                          double *__restrict__ out) {
    __shared__ double sh[4];
    double acc = 0.0;
    for (int r = threadIdx.x; r < n; r += 256) acc = fmax(acc, v[r]);
    acc = wave_max(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) *out = fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]));
}

template <int P>
static void launch_given_p(const SnpKernelArgs &a, const double *delta_km, const double *lse_cur,
                           double *maxdev_partials, hipStream_t s) {
    const dim3 grid(snp_pass_grid(a.N)), block(SNP_THREADS);
    if (a.A == 1)
        hipLaunchKernelGGL((snp_given_delta_kernel<P, true>), grid, block, 0, s, a, delta_km, lse_cur,
                           maxdev_partials);
    else
        hipLaunchKernelGGL((snp_given_delta_kernel<P, false>), grid, block, 0, s, a, delta_km, lse_cur,
                           maxdev_partials);
}

void launch_snp_given_delta(const SnpKernelArgs &a, const double *delta_km, const double *lse_cur,
                            double *maxdev_partials, double *maxdev_out, hipStream_t s) {
    switch (a.P) {
        case 1: launch_given_p<1>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 2: launch_given_p<2>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 3: launch_given_p<3>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 4: launch_given_p<4>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 5: launch_given_p<5>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 6: launch_given_p<6>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 7: launch_given_p<7>(a, delta_km, lse_cur, maxdev_partials, s); break;
        case 8: launch_given_p<8>(a, delta_km, lse_cur, maxdev_partials, s); break;
        default: break;
    }
    hipLaunchKernelGGL(max_reduce_kernel, dim3(1), dim3(256), 0, s, maxdev_partials,
                       snp_pass_grid(a.N), maxdev_out);
}

// --------------------------------------------------------------------------------------------
// deterministic block-wide sum helper for single-workgroup finalisers
// --------------------------------------------------------------------------------------------
static __device__ double block_sum_1024(double v, double *sh /*[16]*/) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_sum(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = sh[0];
#pragma unroll
    for (int ww = 1; ww < 16; ++ww) t += sh[ww];
    return t;
}
static __device__ double block_max_1024(double v, double *sh) {
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    v = wave_max(v);
    __syncthreads();
    if (lane == 0) sh[w] = v;
    __syncthreads();
    double t = sh[0];
#pragma unroll
    for (int ww = 1; ww < 16; ++ww) t = fmax(t, sh[ww]);
    return t;
}

// sum of v[0..n) with stride `stride`, by a 1024-thread workgroup: every thread issues up to 8
// independent loads per pass (clamped index + select, so no branch sits around a load and the
// loads pipeline), then a fixed-order wave / LDS combination.
static __device__ double block_strided_sum_1024(const double *__restrict__ v, int n, int stride,
                                                double *sh) {
    double acc = 0.0;
    for (int r0 = threadIdx.x; r0 < n; r0 += 8 * 1024) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u * 1024;
            const double x = v[(int64_t)min(r, n - 1) * stride];
            t[u] = r < n ? x : 0.0;
        }
        acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    return block_sum_1024(acc, sh);
}

// One 256-thread workgroup per output.  For each of `ncand` candidates (one, or the two of a
// two-step trial): workgroups reduce a column of the per-SNP partials (stored column-major, so the
// loads are contiguous) or a cohort's y.z partials; then the six convergence statistics (fused
// diff, evaluations only); then -- when the pass stashed them -- the responsibility sums, one
// workgroup per (component, candidate) over the per-tile rows.  Eight independent loads per thread
// per pass, fixed-order combination; everything in ONE launch.
struct DotStart { int32_t v[VILMA_MAX_P + 1]; };
struct FinalizeArgs {
    const double *snp_partials;     // [ncand][NT + 6][rows]
    int32_t snp_rows, P, ncand;
    const double *dot_partials;     // candidate c at + c * dot_stride
    int32_t dot_stride;
    DotStart dot_start;
    double *totals[2];
    double *dsum, *dmax;            // both or neither
    const double *sum_rows;         // [ncand][tile rows][AM], or nullptr
    int32_t sum_nrows, AM;
    double *sums[2];
    const int *pred;
};

// column sum over `rows` rows spaced `stride` doubles by a 256-thread workgroup (shared by
// tile_sums_kernel and finalize_kernel so both produce the same bits)
static __device__ __forceinline__ double block_column_sum(const double *__restrict__ src, int rows,
                                                          int64_t stride, double *sh /*[4]*/) {
    double acc = 0.0;
    for (int r0 = threadIdx.x; r0 < rows; r0 += 8 * 256) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[(int64_t)min(r0 + u * 256, rows - 1) * stride];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (r0 + u * 256 < rows) ? t[u] : 0.0;
        acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    return (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

__global__ __launch_bounds__(256) void finalize_kernel(const FinalizeArgs a) {
    __shared__ double sh[4];
    PRED_EXIT(a.pred);
    const int P = a.P, NT = 2 * P + 2, per_cand = NT + P;
    int c = blockIdx.x;
    if (c >= a.ncand * per_cand + (a.dsum != nullptr ? 6 : 0)) {
        // responsibility sums
        c -= a.ncand * per_cand + (a.dsum != nullptr ? 6 : 0);
        const int z = c / a.AM, col = c % a.AM;
        const double tot = block_column_sum(a.sum_rows + (int64_t)z * a.sum_nrows * a.AM + col,
                                            a.sum_nrows, a.AM, sh);
        if (threadIdx.x == 0) a.sums[z][col] = tot;
        return;
    }
    const double *src;
    double *dst;
    int n;
    bool is_max = false;
    if (c < a.ncand * per_cand) {
        const int z = c / per_cand;
        c -= z * per_cand;
        double *totals = a.totals[z];
        if (c < NT) {
            src = a.snp_partials + ((int64_t)z * (NT + 6) + c) * a.snp_rows;
            n = a.snp_rows;
            dst = totals + (c < 2 * P ? c : 3 * P + (c - 2 * P));
        } else {
            const int p = c - NT;
            src = a.dot_partials + (int64_t)z * a.dot_stride + a.dot_start.v[p];
            n = a.dot_start.v[p + 1] - a.dot_start.v[p];
            dst = totals + 2 * P + p;
        }
    } else {                                    // the six convergence statistics (fused diff)
        const int q = c - a.ncand * per_cand;
        src = a.snp_partials + (int64_t)(NT + q) * a.snp_rows;
        n = a.snp_rows;
        is_max = q >= 3;
        dst = q < 3 ? a.dsum + q : a.dmax + (q - 3);
    }
    double acc = 0.0;                           // every maximum here is of non-negative numbers
    for (int r0 = threadIdx.x; r0 < n; r0 += 8 * 256) {
        double t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = src[min(r0 + u * 256, n - 1)];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = (r0 + u * 256 < n) ? t[u] : 0.0;
        if (is_max)
            acc = fmax(acc, fmax(fmax(fmax(t[0], t[1]), fmax(t[2], t[3])),
                                 fmax(fmax(t[4], t[5]), fmax(t[6], t[7]))));
        else
            acc += ((t[0] + t[1]) + (t[2] + t[3])) + ((t[4] + t[5]) + (t[6] + t[7]));
    }
    acc = is_max ? wave_max(acc) : wave_sum(acc);
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0)
        *dst = is_max ? fmax(fmax(sh[0], sh[1]), fmax(sh[2], sh[3]))
                      : (sh[0] + sh[1]) + (sh[2] + sh[3]);
}

void launch_finalize(const double *snp_partials, int snp_rows, int P, const double *dot_partials,
                     int dot_stride, const int32_t *dot_start, int ncand, double *totals_a,
                     double *totals_b, double *dsum, double *dmax, const double *sum_rows,
                     int sum_nrows, int AM, double *sums_a, double *sums_b, hipStream_t s) {
    FinalizeArgs a;
    a.snp_partials = snp_partials; a.snp_rows = snp_rows; a.P = P; a.ncand = ncand;
    a.dot_partials = dot_partials; a.dot_stride = dot_stride;
    for (int p = 0; p <= VILMA_MAX_P; ++p) a.dot_start.v[p] = p <= P ? dot_start[p] : 0;
    a.totals[0] = totals_a; a.totals[1] = totals_b;
    const bool diff = dsum != nullptr && dmax != nullptr;
    a.dsum = diff ? dsum : nullptr; a.dmax = diff ? dmax : nullptr;
    a.sum_rows = sum_rows; a.sum_nrows = sum_nrows; a.AM = AM;
    a.sums[0] = sums_a; a.sums[1] = sums_b;
    a.pred = g_pred;
    const int blocks = ncand * (3 * P + 2) + (diff ? 6 : 0) + (sum_rows != nullptr ? ncand * AM : 0);
    hipLaunchKernelGGL(finalize_kernel, dim3(blocks), dim3(256), 0, s, a);
}

// --------------------------------------------------------------------------------------------
// perm gather / scatter for the stand-alone BlockDiagonalMatrix.dot entry point
// --------------------------------------------------------------------------------------------
__global__ void gather_x_kernel(const double *__restrict__ x, const int32_t *__restrict__ invperm,
                                double *__restrict__ pool_x, int N, int64_t PN) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < PN) {
        const int64_t p = t / N;
        pool_x[p * N + invperm[t]] = x[t];
    }
}
__global__ void scatter_y_kernel(const double *__restrict__ pool_y, const int32_t *__restrict__ invperm,
                                 double *__restrict__ y, int N, int64_t PN) {
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t < PN) {
        const int64_t p = t / N;
        y[t] = pool_y[p * N + invperm[t]];
    }
}
void launch_gather_x(const double *x_snp, const int32_t *invperm, double *pool_x, int N, int P,
                     hipStream_t s) {
    const int64_t PN = (int64_t)N * P;
    hipLaunchKernelGGL(gather_x_kernel, dim3((unsigned)((PN + 255) / 256)), dim3(256), 0, s, x_snp,
                       invperm, pool_x, N, PN);
}
void launch_scatter_y(const double *pool_y, const int32_t *invperm, double *y_snp, int N, int P,
                      hipStream_t s) {
    const int64_t PN = (int64_t)N * P;
    hipLaunchKernelGGL(scatter_y_kernel, dim3((unsigned)((PN + 255) / 256)), dim3(256), 0, s,
                       pool_y, invperm, y_snp, N, PN);
}

// --------------------------------------------------------------------------------------------
// convergence statistics of real_posterior_mean (variational_inference.py:374-382, 292-314)
// --------------------------------------------------------------------------------------------
#define MD_PER_THREAD 4
__global__ __launch_bounds__(256) void mean_diff_kernel(const double *__restrict__ m_cur,
                                                         const double *__restrict__ scalings,
                                                         double *__restrict__ snapshot, int64_t PN,
                                                         double *__restrict__ partials, int compare,
                                                         const int *pred) {
    __shared__ double red[4][6];
    PRED_EXIT(pred);
    double v[6] = {0, 0, 0, 0, 0, 0};
    const int64_t base = (int64_t)blockIdx.x * 256 * MD_PER_THREAD + threadIdx.x;
#pragma unroll
    for (int u = 0; u < MD_PER_THREAD; ++u) {
        const int64_t t = base + (int64_t)u * 256;
        if (t < PN) {
            const double nw = m_cur[t] * scalings[t];
            if (compare) {
                const double od = snapshot[t];
                const double df = fabs(nw - od);
                v[0] += (df <= 1e-6 + 1e-6 * fabs(od)) ? 0.0 : 1.0;
                v[1] += df;
                v[2] += df * df;
                v[3] = fmax(v[3], fabs(nw));
                v[4] = fmax(v[4], df);
                v[5] = fmax(v[5], fabs((nw - od) / (od + 1e-100)));
            }
            snapshot[t] = nw;
        }
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double s = c < 3 ? wave_sum(v[c]) : wave_max(v[c]);
        if (lane == 0) red[w][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        double s = red[0][c];
        for (int ww = 1; ww < 4; ++ww) s = c < 3 ? s + red[ww][c] : fmax(s, red[ww][c]);
        partials[(int64_t)blockIdx.x * 6 + c] = s;
    }
}

__global__ __launch_bounds__(1024) void mean_diff_final_kernel(const double *__restrict__ partials,
                                                                int rows, double *__restrict__ out_sum,
                                                                double *__restrict__ out_max,
                                                                const int *pred) {
    __shared__ double sh[16][6];
    PRED_EXIT(pred);
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    double acc[6] = {0, 0, 0, 0, 0, 0};
    for (int r = threadIdx.x; r < rows; r += 1024) {
        const double *row = partials + (int64_t)r * 6;
#pragma unroll
        for (int c = 0; c < 3; ++c) acc[c] += row[c];
#pragma unroll
        for (int c = 3; c < 6; ++c) acc[c] = fmax(acc[c], row[c]);
    }
#pragma unroll
    for (int c = 0; c < 6; ++c) {
        const double s = c < 3 ? wave_sum(acc[c]) : wave_max(acc[c]);
        if (lane == 0) sh[w][c] = s;
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int c = threadIdx.x;
        double s = sh[0][c];
        for (int ww = 1; ww < 16; ++ww) s = c < 3 ? s + sh[ww][c] : fmax(s, sh[ww][c]);
        if (c < 3) out_sum[c] = s; else out_max[c - 3] = s;
    }
}

int mean_diff_grid(int64_t PN) {
    return (int)((PN + 256 * MD_PER_THREAD - 1) / (256 * MD_PER_THREAD));
}

void launch_mean_diff(const double *m_cur, const double *scalings, double *snapshot, int64_t PN,
                      double *partials, double *out_sum3, double *out_max3, bool compare,
                      hipStream_t s) {
    const int grid = mean_diff_grid(PN);
    hipLaunchKernelGGL(mean_diff_kernel, dim3(grid), dim3(256), 0, s, m_cur, scalings, snapshot, PN,
                       partials, compare ? 1 : 0, g_pred);
    if (compare)
        hipLaunchKernelGGL(mean_diff_final_kernel, dim3(1), dim3(1024), 0, s, partials, grid,
                           out_sum3, out_max3, g_pred);
}

// --------------------------------------------------------------------------------------------
// device M-step: hyper = normalise(max(S / (count + 1e-100), 1e-100)) per annotation
// (variational_inference.py:832-842) and lh = log hyper - 0.5 log_det (numerics.py:149-164).
// One workgroup per annotation row; fixed-order sum.
// --------------------------------------------------------------------------------------------
// one annotation row of the M-step by a 256-thread workgroup (shared by mstep_kernel and the
// device-resident sweep's decision kernel, so both give the same bits)
static __device__ __forceinline__ void mstep_row(const double *__restrict__ sums,
                                                 const double *__restrict__ counts,
                                                 const double *__restrict__ log_det, int M, int a,
                                                 double *__restrict__ hyper, double *__restrict__ lh,
                                                 double *red /*[4] shared*/) {
    const double inv = 1.0 / (counts[a] + 1e-100);
    double part = 0.0;
    for (int k = threadIdx.x; k < M; k += 256) part += fmax(sums[(int64_t)a * M + k] * inv, 1e-100);
    part = wave_sum(part);
    __syncthreads();                                // red[] may still be read from the previous row
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = part;
    __syncthreads();
    const double total = (red[0] + red[1]) + (red[2] + red[3]);
    for (int k = threadIdx.x; k < M; k += 256) {
        const double h = fmax(sums[(int64_t)a * M + k] * inv, 1e-100) / total;
        hyper[(int64_t)a * M + k] = h;
        // det_lh: the host rebuilds this table (vilma_set_hyper) to the same bits
        lh[(int64_t)a * M + k] = det_lh(h, log_det[k]);
    }
}

__global__ __launch_bounds__(256) void mstep_kernel(const double *__restrict__ sums,
                                                     const double *__restrict__ counts,
                                                     const double *__restrict__ log_det, int M,
                                                     double *__restrict__ hyper,
                                                     double *__restrict__ lh, const int *pred) {
    __shared__ double red[4];
    PRED_EXIT(pred);
    mstep_row(sums, counts, log_det, M, blockIdx.x, hyper, lh, red);
}

void launch_mstep(const double *sums, const double *counts, const double *log_det, int A, int M,
                  double *hyper, double *lh, hipStream_t s) {
    hipLaunchKernelGGL(mstep_kernel, dim3(A), dim3(256), 0, s, sums, counts, log_det, M, hyper, lh,
                       g_pred);
}

// --------------------------------------------------------------------------------------------
// The decisions of a device-resident sweep (sweep.hip).  One workgroup; everything it needs is in
// the control block (SweepCtl) and the context's result vector, everything it decides goes back
// into the control block -- the kernels queued behind read their buffers, step sizes, tau and
// whether to run at all from there -- and, as a snapshot, straight into host memory.
//
// The host queues groups  [beta trial (two candidates) -> all-reduce -> TRIAL decision -> evaluation
// (-> all-reduce -> EVAL decision -> re-evaluation, with --learn-scaling)]  and the device walks
// the reference's loop through them (variational_inference.py:419-450):
//   TRIAL decision:
//   - an evaluation that ran since the last decision (the one after the previous sweep's M-step or
//     tau update) is looked at first: its objective closes that sweep -- _nat_grad_step's delta_sum,
//     _optimize_step's running ELBO change (:403-409), optimize()'s "no posterior mean moved" stop
//     (:374-382) as a veto;
//   - the line search of _update_beta (:777-800) on the two candidates: A if it passes the accept
//     test, else B; neither -> the next group's trial runs at the next larger L (this group's
//     evaluation is skipped);
//   - _nat_grad_step's break rule after an accepted step (:432-438): the inner loop goes on -> the
//     next group's trial starts from the accepted candidate (evaluation skipped); it ends -> the
//     M-step of _update_hyper_delta (:837-848) from the accepted candidate's responsibility sums
//     and the group's evaluation runs;
//   - in every case the buffers swap roles (vilma_accept's bookkeeping) and the next trial's step
//     sizes are set (L / 1.25 floored at 1, :430).
//   EVAL decision (--learn-scaling): looks at the evaluation after the M-step; if the sweep has
//     gained less than EM_TOL so far, _update_error_scaling (:472-486): tau from the sums, and the
//     re-evaluation queued behind runs.
// What the device will not decide -- L beyond L_MAX, a non-positive tau, the veto -- turns the block
// dead (every kernel queued behind exits) and the host carries on from the snapshot.  All of this
// in the host's operation order without fused multiply-add (detmath.h), so the host, which replays
// every decision from the snapshot's sums, can never disagree.
// --------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void sweep_decide_kernel(const SweepDecideArgs a) {
    __shared__ double red[4];
    __shared__ double sh_x[VILMA_SNAP_EXTRA];
    __shared__ int sh_mstep;
    if (threadIdx.x == 0) {
        DecideReport rep;
        decide_core(a, a.ctl, a.results, rep);
        sh_mstep = rep.mstep;
        decide_snapshot_scalars(a, a.ctl, rep, sh_x);
    }
    __syncthreads();
    // snapshot for the host: the result vector as the decision saw it (hyper_delta still the one
    // of the sweep just completed), then the block's scalars after the decision
    for (int t = threadIdx.x; t < a.n_results; t += blockDim.x) a.snap[t] = a.results[t];
    if (threadIdx.x < VILMA_SNAP_EXTRA - 1) a.snap[a.n_results + threadIdx.x] = sh_x[threadIdx.x];
    // the snapshot lives in host memory the device writes directly: data first, then the serial
    // number the host polls for
    __threadfence_system();
    __syncthreads();
    if (threadIdx.x == 0) {
        *(volatile double *)(a.snap + a.n_results + SNAP_SERIAL) = a.serial;
        __threadfence_system();
    }
    const int mstep = sh_mstep;
    if (mstep == 0) return;
    const double *sums = a.results + (mstep == 1 ? a.o_sa : a.o_sb);
    for (int an = 0; an < a.A; ++an) mstep_row(sums, a.counts, a.log_det, a.M, an, a.hyper, a.lh, red);
}

void launch_sweep_decide(const SweepDecideParams &p, hipStream_t s) {
    const SweepDecideArgs a = decide_args(p);
    hipLaunchKernelGGL(sweep_decide_kernel, dim3(1), dim3(256), 0, s, a);
}
