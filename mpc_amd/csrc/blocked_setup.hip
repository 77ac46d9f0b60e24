// Shared set-up of the general dense / constraint-space QP kernels for large n, spread over the whole GPU.
//
// What it replaces: phase 1 of dense_qp_kernel, ONE workgroup that scales the Hessian and the rows, factors Ps + sigma I
// column by column and inverts it -- 0.12 s for the 600 x 600 Hessian and 13,200 rows of BASELINE.json configs[4], once per
// model (reference: the per-call factorisation inside cvxopt.solvers.qp, LaunchPlannerQPFormulation.py:173).
//
// Here the same quantities, written to the same places of the workspace of "QP 0", come from
//   * three scaling kernels (column scaling d, Ps / W = Ps + sigma I over an n^2 grid, one wave per constraint row),
//   * a right-looking blocked Cholesky with 32-wide panels: the diagonal block is factored AND inverted by one wave in LDS,
//     the panel below it is a product with that inverse (L21 = W21 D^-T), the trailing update a symmetric rank-32 product --
//     both on the fp64 MFMA GEMM of rows_qp.hip,
//   * T = L^-1 by block rows, stored transposed (Tt, upper triangular) so that every product is "A B'": two GEMMs per block row,
//   * Hi = T'T = Tt Tt' as one GEMM.
// Every step is an ordinary launch on the caller's stream, nothing synchronises.
#include <hip/hip_runtime.h>
#include <algorithm>
#include "../../include/mpcqp.h"

int gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K, double alpha,
            double beta, hipStream_t st);

namespace {

constexpr int NB = 32;

struct ScaleParams {
    int n, m, pe;
    const double *P, *G, *A;
    double *d, *rn, *Ps, *W, *Cs, *sigma;
    double sigma_rel;
};

// d[j] = 1/sqrt(P_jj) (columns with an empty diagonal: 1 / largest row entry), sigma = sigma_rel max_j |Ps_jj|
__global__ __launch_bounds__(256) void scale_cols_kernel(ScaleParams p) {
    __shared__ double red[256];
    const int tid = threadIdx.x, n = p.n;
    double dmax = 0.0;
    for (int j = tid; j < n; j += 256) {
        const double dg = p.P[(size_t)j * n + j];
        double dj;
        if (dg > 0.0) dj = 1.0 / sqrt(dg);
        else {
            double cm = 0.0;
            for (int r = 0; r < p.pe; ++r) cm = fmax(cm, fabs(p.A[(size_t)r * n + j]));
            for (int r = 0; r < p.m; ++r) cm = fmax(cm, fabs(p.G[(size_t)r * n + j]));
            dj = cm > 0.0 ? 1.0 / cm : 1.0;
        }
        p.d[j] = dj;
        dmax = fmax(dmax, fabs(dg * dj * dj));
    }
    red[tid] = dmax;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) red[tid] = fmax(red[tid], red[tid + s]);
        __syncthreads();
    }
    if (tid == 0) *p.sigma = p.sigma_rel * (red[0] > 0.0 ? red[0] : 1.0);
}

__global__ __launch_bounds__(256) void scale_hessian_kernel(ScaleParams p) {
    const int n = p.n;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)n * n) return;
    const int i = (int)(idx / n), j = (int)(idx % n);
    const double v = 0.5 * (p.P[idx] + p.P[(size_t)j * n + i]) * p.d[i] * p.d[j];
    p.Ps[idx] = v;
    p.W[idx] = v + (i == j ? *p.sigma : 0.0);
}

// one wave per row: rn = |row . d|_2, Cs row = row . d / rn   (equality rows first, as dense_qp_kernel stacks them)
__global__ __launch_bounds__(256) void scale_rows_kernel(ScaleParams p) {
    const int lane = threadIdx.x & 63, n = p.n;
    const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (r >= p.m + p.pe) return;
    const double* src = r < p.pe ? p.A + (size_t)r * n : p.G + (size_t)(r - p.pe) * n;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) { const double v = src[j] * p.d[j]; s += v * v; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o);
    s = sqrt(s);
    const double rn = s > 0.0 ? s : 1.0;
    if (lane == 0) p.rn[r] = rn;
    double* dst = p.Cs + (size_t)r * n;
    for (int j = lane; j < n; j += 64) dst[j] = src[j] * p.d[j] / rn;
}

// Diagonal block k0 .. k0+kb of W (already carrying the trailing updates): Cholesky factor -> Lout, its inverse ->
// Dinv (32 x 32, zero padded) and, transposed, into the diagonal block of Tt.  One wave; flag = first failing pivot.
__global__ __launch_bounds__(64) void diag_block_kernel(const double* W, int n, int k0, int kb, double* Lout, double* Dinv,
                                                        double* Tt, int* flag, int flag_bits) {
    __shared__ double a[NB][NB + 1];
    __shared__ double t[NB][NB + 1];
    const int lane = threadIdx.x;
    for (int idx = lane; idx < NB * NB; idx += 64) {
        const int i = idx / NB, j = idx % NB;
        a[i][j] = (i < kb && j < kb) ? W[(size_t)(k0 + i) * n + k0 + j] : (i == j ? 1.0 : 0.0);
        t[i][j] = 0.0;
    }
    __syncthreads();
    for (int j = 0; j < kb; ++j) {
        double dj = a[j][j];
        if (!(dj > 0.0)) {                      // not positive definite: record it and keep the arithmetic finite
            if (lane == 0) { if (flag_bits) *flag |= 1; else if (*flag < 0) *flag = k0 + j; }
            dj = 1.0;
        }
        const double ljj = sqrt(dj), inv = 1.0 / ljj;
        __syncthreads();
        if (lane == 0) a[j][j] = ljj;
        if (lane > j && lane < kb) a[lane][j] *= inv;
        __syncthreads();
        const int rem = kb - j - 1;
        for (int idx = lane; idx < rem * rem; idx += 64) {
            const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
            if (c <= i) a[i][c] -= a[i][j] * a[c][j];
        }
        __syncthreads();
    }
    // lane c: column c of the inverse by forward substitution
    if (lane < kb) {
        const int c = lane;
        t[c][c] = 1.0 / a[c][c];
        for (int i = c + 1; i < kb; ++i) {
            double s = 0.0;
            for (int k = c; k < i; ++k) s += a[i][k] * t[k][c];
            t[i][c] = -s / a[i][i];
        }
    }
    __syncthreads();
    for (int idx = lane; idx < NB * NB; idx += 64) {
        const int i = idx / NB, j = idx % NB;
        const bool in = i < kb && j < kb;
        Dinv[idx] = (in && j <= i) ? t[i][j] : 0.0;
        if (in) {
            Lout[(size_t)(k0 + i) * n + k0 + j] = j <= i ? a[i][j] : 0.0;
            Tt[(size_t)(k0 + i) * n + k0 + j] = j >= i ? t[j][i] : 0.0;
        }
    }
}

__global__ void finish_kernel(double* W, const double* sigma, const int* flag) {
    if (threadIdx.x == 0) { W[0] = *sigma; W[1] = *flag >= 0 ? 3.0 : 0.0; }
}

__global__ void clear_flag_kernel(int* flag) { if (threadIdx.x == 0) *flag = -1; }

}  // namespace

// X = W^-1 for a symmetric positive definite W (n x n, destroyed).  Lout, Tt: n x n scratch; blk: blocked_inverse_scratch(n)
// doubles.  A failing pivot is recorded in *flag (flag_bits: OR 1 into it; else the pivot's index if *flag < 0) and replaced
// by 1 so that the arithmetic stays finite.  About 5 n / 32 launches on the caller's stream, no synchronisation.
long blocked_inverse_scratch(int n) { const long nb = (n + NB - 1) / NB; return nb * NB * NB + (long)n * NB; }

int blocked_spd_inverse(double* W, int n, double* Lout, double* Tt, double* blk, double* X, int* flag, int flag_bits,
                        hipStream_t st) {
    const long nn = (long)n * n;
    const int nb = (n + NB - 1) / NB;
    double* Dinv = blk;                                   // nb blocks of 32 x 32
    double* St = blk + (size_t)nb * NB * NB;              // (n, 32) panel
    if (hipMemsetAsync(Tt, 0, nn * sizeof(double), st) != hipSuccess) return MPCQP_EHIP;
    int rc = MPCQP_OK;
    for (int b = 0; b < nb && rc == MPCQP_OK; ++b) {
        const int k0 = b * NB, kb = std::min(NB, n - k0), rem = n - k0 - kb;
        double* Db = Dinv + (size_t)b * NB * NB;
        hipLaunchKernelGGL(diag_block_kernel, dim3(1), dim3(64), 0, st, W, n, k0, kb, Lout, Db, Tt, flag, flag_bits);
        if (rem > 0) {
            double* L21 = Lout + (size_t)(k0 + kb) * n + k0;
            rc = gemm_nt(W + (size_t)(k0 + kb) * n + k0, n, Db, NB, L21, n, rem, kb, kb, 1.0, 0.0, st);       // L21 = W21 D^-T
            if (rc == MPCQP_OK)
                rc = gemm_nt(L21, n, L21, n, W + (size_t)(k0 + kb) * n + k0 + kb, n, rem, rem, kb, -1.0, 1.0, st);   // W22 -= L21 L21'
        }
    }
    for (int b = 1; b < nb && rc == MPCQP_OK; ++b) {
        const int k0 = b * NB, kb = std::min(NB, n - k0);
        const double* Db = Dinv + (size_t)b * NB * NB;
        // block row b of T = L^-1:  T_b,: = -D_b^-1 (L_b,0:k0 T_0:k0,:), transposed throughout
        rc = gemm_nt(Tt, n, Lout + (size_t)k0 * n, n, St, NB, k0, kb, k0, 1.0, 0.0, st);
        if (rc == MPCQP_OK) rc = gemm_nt(St, NB, Db, NB, Tt + k0, n, k0, kb, kb, -1.0, 0.0, st);
    }
    if (rc == MPCQP_OK) rc = gemm_nt(Tt, n, Tt, n, X, n, n, n, n, 1.0, 0.0, st);       // X = T'T
    if (rc == MPCQP_OK && hipGetLastError() != hipSuccess) rc = MPCQP_EHIP;
    return rc;
}

// The blocked path needs room for the 32 x 32 inverses and one n x 32 panel inside an n x n scratch block
bool blocked_setup_fits(int n) { return n >= 2 * NB + NB; }

// Same contract as phase 1 of dense_qp_kernel: into the workspace of QP 0 go Ps, Hi = (Ps + sigma I)^-1, Cs, d, rn and
// (sigma, status) in the first two entries of its W block.  Scratch: the W, T, Y and L blocks of QP 0 (n x n each).
int blocked_shared_setup(int n, int m, int pe, const double* P, const double* G, const double* A, double* ws, int* iws,
                         double sigma_rel, hipStream_t st) {
    const long nn = (long)n * n, mc = (long)m + pe, V = std::max<long>(std::max<long>(n, mc), 1);
    double* Ps = ws; double* Hi = Ps + nn; double* W = Hi + nn; double* Tt = W + nn;
    double* Cs = Tt + nn; double* Lout = Cs + mc * n; double* Lblk = Lout + nn;
    double* vec = Lblk + nn;
    double* d = vec; double* rn = vec + 3 * V; double* sigma = vec + 13 * V;       // sigma parks in QP 0's t1 until the end
    int* flag = iws;
    ScaleParams sp{n, m, pe, P, G, A, d, rn, Ps, W, Cs, sigma, sigma_rel};
    hipLaunchKernelGGL(clear_flag_kernel, dim3(1), dim3(64), 0, st, flag);
    hipLaunchKernelGGL(scale_cols_kernel, dim3(1), dim3(256), 0, st, sp);
    hipLaunchKernelGGL(scale_hessian_kernel, dim3((unsigned)((nn + 255) / 256)), dim3(256), 0, st, sp);
    if (mc > 0) hipLaunchKernelGGL(scale_rows_kernel, dim3((unsigned)((mc + 3) / 4)), dim3(256), 0, st, sp);
    if (hipGetLastError() != hipSuccess) return MPCQP_EHIP;
    const int rc = blocked_spd_inverse(W, n, Lout, Tt, Lblk, Hi, flag, 0, st);
    if (rc != MPCQP_OK) return rc;
    hipLaunchKernelGGL(finish_kernel, dim3(1), dim3(64), 0, st, W, sigma, flag);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}
