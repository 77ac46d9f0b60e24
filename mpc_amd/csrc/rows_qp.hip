// Shared-matrix QPs with many inequality rows, solved in CONSTRAINT space.
//
//     min 1/2 x'Px + q_b'x   s.t.  G x <= h_b        b = 0 .. batch-1,  one (P, G) for the batch, P positive definite
//
// This is the condensed MPC problem with state-box / polytopic rows (reference: rows of the
// prediction matrix stacked under the input box, LaunchPlannerQPFormulation.py:109-114, Launch3DPlanner.py:134-142,
// handed to cvxopt.solvers.qp at LaunchPlannerQPFormulation.py:173): n = N nu variables, m = 2 N (nu + nx + nc) rows
// (13,200 x 600 for BASELINE.json config 5), a few of which bind.
//
// dense_qp.hip re-evaluates all m rows against x after every active-set change: m*n doubles per step per instance
// (63 MB for config 5) -- that pass is ~all of its time.  Here the set-up forms, once per (P, G),
//     CH   = Cs Hi          (m x n)    row r = the step direction of row r
//     Gram = Cs Hi Cs'      (m x m)    the dual Hessian
// (Cs = scaled rows, Hi = inverse scaled Hessian: both from the dense kernel's shared set-up) with an fp64 MFMA
// GEMM, and each instance keeps its row residuals res = Cs x - cs up to date through Gram rows:
//     x <- x - t Hi (n_p - N'r)      =>      res <- res - t (Gram[p,:] - sum_a r_a Gram[A_a,:])
// i.e. (k+1) rows of m doubles instead of the m x n pass, and the Schur complement of the active rows is a gather
// S = Gram[A, A].  The iteration itself is the same Goldfarb-Idnani dual active-set method as dense_qp.hip (same
// pivoting rules, same thresholds), so both paths land on the same vertex; the tests compare them.
//
// Per solve: scale (q, h) -> GEMM xu = -Qs Hi -> GEMM res = Xu Cs' - cs -> one 1024-thread workgroup per instance.
#include "common.h"
#include <atomic>
#include <algorithm>
#include <cstdint>

int dense_shared_setup_sigma(int n, int m, const double* P, const double* G, double* workspace, int* int_workspace,
                             double sigma_rel, hipStream_t stream);

namespace {

__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// ---- C (M x N) = alpha A (M x K) B' (B is N x K) + beta C : fp64 MFMA, 64 x 64 per workgroup, 32 x 32 per wave ----------
struct GemmParams {
    const double *A, *B;
    double* C;
    long lda, ldb, ldc;
    int M, N, K, vec;
    double alpha, beta;
    // epilogue: C = (alpha acc + bias[j]) (* or /) cscale[j] + beta Cin   (bias, cscale per column; NULL = none; Cin NULL = C)
    const double *bias, *cscale, *Cin;
    int cdiv;
    // gemm_nt_lds_kernel only: column j of C is the product with row 2 bpairs[j / 2] + (j & 1) of B, for the first 2 bcount[0]
    // columns (the count lives on the device: workgroups beyond it leave at once); nullptr: column j <-> row j
    const int *bpairs, *bcount;
};

__global__ __launch_bounds__(256) void gemm_nt_kernel(GemmParams g) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.y * 64 + (wave >> 1) * 32, j0 = blockIdx.x * 64 + (wave & 1) * 32;
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
    const double* ar[2]; const double* br[2];
    bool av[2], bv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = i0 + 16 * t + col, j = j0 + 16 * t + col;
        av[t] = i < g.M; bv[t] = j < g.N;
        ar[t] = g.A + (size_t)min(i, g.M - 1) * g.lda;
        br[t] = g.B + (size_t)min(j, g.N - 1) * g.ldb;
    }
    // each lane owns 4 consecutive k of every 16 (the k order inside a product is free as long as A and B agree)
    for (int k0 = 0; k0 < g.K; k0 += 16) {
        double a[2][4], b[2][4];
        const int kb = k0 + 4 * kq;
        if (g.vec && k0 + 16 <= g.K) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double2 a0 = *reinterpret_cast<const double2*>(ar[t] + kb), a1 = *reinterpret_cast<const double2*>(ar[t] + kb + 2);
                const double2 b0 = *reinterpret_cast<const double2*>(br[t] + kb), b1 = *reinterpret_cast<const double2*>(br[t] + kb + 2);
                a[t][0] = a0.x; a[t][1] = a0.y; a[t][2] = a1.x; a[t][3] = a1.y;
                b[t][0] = b0.x; b[t][1] = b0.y; b[t][2] = b1.x; b[t][3] = b1.y;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kb + e;
                    a[t][e] = k < g.K ? ar[t][k] : 0.0;
                    b[t][e] = k < g.K ? br[t][k] : 0.0;
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!av[t]) a[t][e] = 0.0;
                if (!bv[t]) b[t][e] = 0.0;
            }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = mfma_f64(a[ti][e], b[tj][e], acc[ti][tj]);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int i = i0 + 16 * ti + kq + 4 * r4, j = j0 + 16 * tj + col;
                if (i < g.M && j < g.N) {
                    double* c = g.C + (size_t)i * g.ldc + j;
                    double v = g.alpha * acc[ti][tj][r4];
                    if (g.bias) v += g.bias[j];
                    if (g.cscale) v = g.cdiv ? v / g.cscale[j] : v * g.cscale[j];
                    if (g.beta != 0.0) v += g.beta * (g.Cin ? g.Cin[(size_t)i * g.ldc + j] : *c);
                    *c = v;
                }
            }
}

// The same tile with the operands staged through LDS: a 64 x 32 chunk of A and of B per step, fetched by the whole workgroup
// with row-contiguous 16-byte loads (16 lanes per row: every line once per workgroup -- the kernel above has each wave fetch
// its own 32 rows, lane by lane 16 rows apart, so every operand line crosses the CU's L1 twice and its address unit sixteen
// lines per instruction), held in registers while the previous chunk is multiplied, then written to LDS rows 34 doubles apart
// (row r starts on bank 4 r mod 64: the 16 rows of an MFMA operand read, 16 bytes each, cover all 64 banks once).
// Needs 16-byte aligned operands and an even K (launch_gemm_nt checks); the k order inside a 16-step is the same as above.
__global__ __launch_bounds__(256) void gemm_nt_lds_kernel(GemmParams g) {
    constexpr int KC = 32, LD = KC + 2;
    __shared__ __attribute__((aligned(16))) double sA[64 * LD];
    __shared__ __attribute__((aligned(16))) double sB[64 * LD];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int bi = blockIdx.y * 64, bj = blockIdx.x * 64;
    const int wi = (wave >> 1) * 32, wj = (wave & 1) * 32;
    const int N = g.bpairs ? min(g.N, 2 * g.bcount[0]) : g.N;
    if (bj >= N) return;                            // uniform over the workgroup, before any barrier
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
    // staging: thread -> (row = tid / 16 + 16 u, doubles 2 (tid % 16), +1) of the chunk, u = 0..3
    const int sr = tid >> 4, sc = 2 * (tid & 15);
    const double* ga[4]; const double* gb[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        ga[u] = g.A + (size_t)min(bi + sr + 16 * u, g.M - 1) * g.lda + sc;      // rows beyond the matrix: a copy of the last row,
        const int j = min(bj + sr + 16 * u, N - 1);                             // whose products are never stored
        gb[u] = g.B + (size_t)(g.bpairs ? 2 * g.bpairs[j >> 1] + (j & 1) : j) * g.ldb + sc;
    }
    double2 ra[4], rb[4];
    auto fetch = [&](int k0) {
        const bool on = k0 + sc < g.K;          // K is even: a pair is inside or outside as a whole
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            ra[u] = on ? *reinterpret_cast<const double2*>(ga[u] + k0) : double2{0.0, 0.0};
            rb[u] = on ? *reinterpret_cast<const double2*>(gb[u] + k0) : double2{0.0, 0.0};
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < g.K; k0 += KC) {
        __syncthreads();                        // the previous chunk has been read by every wave
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            *reinterpret_cast<double2*>(&sA[(sr + 16 * u) * LD + sc]) = ra[u];
            *reinterpret_cast<double2*>(&sB[(sr + 16 * u) * LD + sc]) = rb[u];
        }
        __syncthreads();
        if (k0 + KC < g.K) fetch(k0 + KC);      // in flight while this chunk is multiplied
#pragma unroll
        for (int ks = 0; ks < KC; ks += 16) {
            double2 a[2][2], b[2][2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                a[t][0] = *reinterpret_cast<const double2*>(&sA[(wi + 16 * t + col) * LD + ks + 4 * kq]);
                a[t][1] = *reinterpret_cast<const double2*>(&sA[(wi + 16 * t + col) * LD + ks + 4 * kq + 2]);
                b[t][0] = *reinterpret_cast<const double2*>(&sB[(wj + 16 * t + col) * LD + ks + 4 * kq]);
                b[t][1] = *reinterpret_cast<const double2*>(&sB[(wj + 16 * t + col) * LD + ks + 4 * kq + 2]);
            }
#pragma unroll
            for (int e = 0; e < 4; ++e)
#pragma unroll
                for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                    for (int tj = 0; tj < 2; ++tj) {
                        const double av = (e & 1) ? a[ti][e >> 1].y : a[ti][e >> 1].x;
                        const double bv = (e & 1) ? b[tj][e >> 1].y : b[tj][e >> 1].x;
                        acc[ti][tj] = mfma_f64(av, bv, acc[ti][tj]);
                    }
        }
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int i = bi + wi + 16 * ti + kq + 4 * r4, j = bj + wj + 16 * tj + col;
                if (i < g.M && j < N) {
                    double* c = g.C + (size_t)i * g.ldc + j;
                    double v = g.alpha * acc[ti][tj][r4];
                    if (g.bias) v += g.bias[j];
                    if (g.cscale) v = g.cdiv ? v / g.cscale[j] : v * g.cscale[j];
                    if (g.beta != 0.0) v += g.beta * (g.Cin ? g.Cin[(size_t)i * g.ldc + j] : *c);
                    *c = v;
                }
            }
}

// The same product when C is small and K long (Xu = -Qs Hi: 256 x 600 over K = 600 is 40 of the tiles above, each walking
// all of K): 32 x 32 per workgroup and the FOUR WAVES SPLIT K, each accumulating the whole tile over its quarter; the
// partial tiles meet in LDS in a fixed order (deterministic), wave w finishes rows 8w .. 8w+7.  Four times the workgroups and
// a quarter of the chain.
__global__ __launch_bounds__(256) void gemm_nt_splitk_kernel(GemmParams g) {
    __shared__ double part[4][32][33];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int i0 = blockIdx.y * 32, j0 = blockIdx.x * 32;
    const int kblocks = (g.K + 15) / 16, per = (kblocks + 3) / 4;
    const int kbeg = wave * per * 16, kend = min(g.K, (wave + 1) * per * 16);
    d4 acc[2][2];
#pragma unroll
    for (int a = 0; a < 2; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b) acc[a][b] = d4{0, 0, 0, 0};
    const double* ar[2]; const double* br[2];
    bool av[2], bv[2];
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int i = i0 + 16 * t + col, j = j0 + 16 * t + col;
        av[t] = i < g.M; bv[t] = j < g.N;
        ar[t] = g.A + (size_t)min(i, g.M - 1) * g.lda;
        br[t] = g.B + (size_t)min(j, g.N - 1) * g.ldb;
    }
    for (int k0 = kbeg; k0 < kend; k0 += 16) {
        double a[2][4], b[2][4];
        const int kb = k0 + 4 * kq;
        if (g.vec && k0 + 16 <= g.K) {
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const double2 a0 = *reinterpret_cast<const double2*>(ar[t] + kb), a1 = *reinterpret_cast<const double2*>(ar[t] + kb + 2);
                const double2 b0 = *reinterpret_cast<const double2*>(br[t] + kb), b1 = *reinterpret_cast<const double2*>(br[t] + kb + 2);
                a[t][0] = a0.x; a[t][1] = a0.y; a[t][2] = a1.x; a[t][3] = a1.y;
                b[t][0] = b0.x; b[t][1] = b0.y; b[t][2] = b1.x; b[t][3] = b1.y;
            }
        } else {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const int k = kb + e;
                    a[t][e] = k < g.K ? ar[t][k] : 0.0;
                    b[t][e] = k < g.K ? br[t][k] : 0.0;
                }
        }
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!av[t]) a[t][e] = 0.0;
                if (!bv[t]) b[t][e] = 0.0;
            }
#pragma unroll
        for (int e = 0; e < 4; ++e)
#pragma unroll
            for (int ti = 0; ti < 2; ++ti)
#pragma unroll
                for (int tj = 0; tj < 2; ++tj) acc[ti][tj] = mfma_f64(a[ti][e], b[tj][e], acc[ti][tj]);
    }
#pragma unroll
    for (int ti = 0; ti < 2; ++ti)
#pragma unroll
        for (int tj = 0; tj < 2; ++tj)
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) part[wave][16 * ti + kq + 4 * r4][16 * tj + col] = acc[ti][tj][r4];
    __syncthreads();
    for (int e = threadIdx.x; e < 32 * 32; e += 256) {
        const int r = e >> 5, c = e & 31;
        const int i = i0 + r, j = j0 + c;
        if (i < g.M && j < g.N) {
            double* cp = g.C + (size_t)i * g.ldc + j;
            double v = g.alpha * (((part[0][r][c] + part[1][r][c]) + part[2][r][c]) + part[3][r][c]);
            if (g.bias) v += g.bias[j];
            if (g.cscale) v = g.cdiv ? v / g.cscale[j] : v * g.cscale[j];
            if (g.beta != 0.0) v += g.beta * (g.Cin ? g.Cin[(size_t)i * g.ldc + j] : *cp);
            *cp = v;
        }
    }
}

int launch_gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K,
                   double alpha, double beta, hipStream_t st, const double* bias = nullptr, const double* cscale = nullptr,
                   int cdiv = 0, const double* Cin = nullptr) {
    GemmParams g;
    g.bias = bias; g.cscale = cscale; g.cdiv = cdiv; g.Cin = Cin; g.bpairs = nullptr; g.bcount = nullptr;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = alpha; g.beta = beta;
    g.vec = (lda % 2 == 0 && ldb % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0);
    if ((long)ceil_div(N, 64) * ceil_div(M, 64) < 128 && K >= 256)      // few tiles, long K: split K over the waves
        hipLaunchKernelGGL(gemm_nt_splitk_kernel, dim3(ceil_div(N, 32), ceil_div(M, 32)), dim3(256), 0, st, g);
    else if (g.vec && K % 2 == 0 && K >= 64 && M >= 64 && N >= 64)
        hipLaunchKernelGGL(gemm_nt_lds_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, g);
    else
        hipLaunchKernelGGL(gemm_nt_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, g);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

// C (M x 2 count) = A (M x K) Bsel' where Bsel takes the row pairs bpairs[0 .. count) of B (count on the device, at most N / 2);
// false if the shapes do not fit the LDS-staged kernel (the caller then forms the whole product)
bool launch_gemm_nt_pairs(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K,
                          const int* bpairs, const int* bcount, hipStream_t st) {
    const bool vec = lda % 2 == 0 && ldb % 2 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0 && reinterpret_cast<uintptr_t>(B) % 16 == 0;
    if (!(vec && K % 2 == 0 && K >= 64 && M >= 64 && N >= 64 && N % 2 == 0)) return false;
    GemmParams g;
    g.bias = nullptr; g.cscale = nullptr; g.cdiv = 0; g.Cin = nullptr; g.bpairs = bpairs; g.bcount = bcount;
    g.A = A; g.B = B; g.C = C; g.lda = lda; g.ldb = ldb; g.ldc = ldc; g.M = M; g.N = N; g.K = K; g.alpha = 1.0; g.beta = 0.0; g.vec = 1;
    hipLaunchKernelGGL(gemm_nt_lds_kernel, dim3(ceil_div(N, 64), ceil_div(M, 64)), dim3(256), 0, st, g);
    return true;
}

// ---- per-instance scaling of (q, h) ------------------------------------------------------------------------------------
struct Layout {
    // shared block: [dense set-up of "QP 0" | CH | Gram]
    long nd0;
    // per-instance block
    long o_res, o_cs, o_qs, o_xu, o_x, o_t1, o_dv, o_slam, o_sact, o_wact, stride;      // slam/sact: active set handed to pass 2; wact: the final set, kept for a warm start
};

// capacity of the active set: first pass (several workgroups per CU), second pass (one per CU, the LDS nearly full)
int rows_kl1(int n, int m) { return std::max(1, std::min(std::min(n, m), 96)); }
int rows_kl2(int n, int m) { return std::max(1, std::min(std::min(n, m), 176)); }

Layout make_layout(int n, int m) {
    Layout l;
    l.nd0 = mpcqp_dense_workspace(n, m, 0, nullptr);
    l.nd0 += l.nd0 & 1;
    const long me = m + (m & 1), ne = n + (n & 1);
    l.o_res = 0; l.o_cs = me; l.o_qs = 2 * me; l.o_xu = l.o_qs + ne; l.o_x = l.o_xu + ne; l.o_t1 = l.o_x + ne; l.o_dv = l.o_t1 + ne;
    const long K2 = rows_kl2(n, m) + 3;
    l.o_slam = l.o_dv + ne; l.o_sact = l.o_slam + K2 + (K2 & 1);
    l.o_wact = l.o_sact + (K2 + 1) / 2 + 1;
    l.stride = l.o_wact + (K2 + 1) / 2 + 1;
    l.stride += l.stride & 1;
    return l;
}

// ints behind [dense set-up | CH | Gram]: mirror row of every row (m) | mirror pair of every pair (m / 2 + 1) | ok flag, number of
// working pairs (4) | the working pairs in ascending order (m / 2 + 1)
inline long mirror_doubles(int m) { return ((long)m + 2 * (m / 2 + 1) + 4 + 1) / 2 + 1; }

struct SharedPtrs { const double *Ps, *Hi, *W0, *Cs, *d, *rn; };

__host__ __device__ inline SharedPtrs shared_ptrs(const double* ws0, int n, int m) {
    // the layout of dense_qp.hip's workspace for one QP (p = 0)
    SharedPtrs s;
    const long V = n > m ? n : (m > 1 ? m : 1);
    s.Ps = ws0; s.Hi = ws0 + (size_t)n * n; s.W0 = ws0 + (size_t)2 * n * n; s.Cs = ws0 + (size_t)4 * n * n;
    const double* v0 = s.Cs + (size_t)m * n + (size_t)2 * n * n;
    s.d = v0; s.rn = v0 + 3 * V;
    return s;
}

struct PrepParams { const double *q, *h, *ws0; double* inst; Layout l; int n, m, B; };

__global__ __launch_bounds__(256) void rows_prep_kernel(PrepParams p) {
    const SharedPtrs s = shared_ptrs(p.ws0, p.n, p.m);
    const long per = (long)p.n + p.m, total = per * p.B;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long)gridDim.x * 256) {
        const int b = (int)(i / per), e = (int)(i % per);
        double* c = p.inst + (size_t)b * p.l.stride;
        if (e < p.n) c[p.l.o_qs + e] = p.q[(size_t)b * p.n + e] * s.d[e];
        else {
            const int r = e - p.n;
            const double v = p.h[(size_t)b * p.m + r] / s.rn[r];
            c[p.l.o_cs + r] = v;
        }
    }
}

// ---- the active-set iteration, one workgroup per instance ---------------------------------------------------------------
constexpr int RT = 1024;

struct RowsParams {
    const double *ws0, *CH, *Gram;
    double* inst;
    Layout l;
    double *x, *z, *obj;
    int *status, *iters;
    int n, m, B, KL, max_steps, refine, pass, warm;
    const int* mir2;    // per PAIR of rows: the pair holding their negatives (-1: none), valid where *mirok != 0 (rows_mirror_* below)
    const int* mirok;   // [0] ok, [1] number of working pairs (a pair and the pair of its negatives count once)
    const int* mlist;   // the working pairs, ascending
    int res_compact;    // with res_lds: the instance's residual block holds the products Xu Cs' of the working pairs only, in list order
                        // (launch_gemm_nt_pairs); the rows' residuals -- and their negatives' -- are formed while they are loaded
    int res_lds;        // 1: the residual vector lives in LDS behind the other arrays for the length of the iteration (first pass, one workgroup per CU)
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

__device__ double blk_sum(double v, double* red) {
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    double s = 0.0;
    for (int w = 0; w < RT / 64; ++w) s += red[w];
    return s;
}

// the largest v over candidates (idx >= 0), ties to the lowest index
__device__ void blk_argmax(double v, int idx, double* red, int* redi, double& vout, int& iout) {
#pragma unroll
    for (int o = 32; o; o >>= 1) {
        const double ov = __shfl_xor(v, o); const int oi = __shfl_xor(idx, o);
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { red[threadIdx.x >> 6] = v; redi[threadIdx.x >> 6] = idx; }
    __syncthreads();
    v = red[0]; idx = redi[0];
    for (int w = 1; w < RT / 64; ++w) {
        const double ov = red[w]; const int oi = redi[w];
        if (oi >= 0 && (idx < 0 || ov > v || (ov == v && oi < idx))) { v = ov; idx = oi; }
    }
    vout = v; iout = idx;
}

// y = M v for the symmetric n x n matrix M (row-major, read by columns: coalesced), v staged in LDS (vs), all RT threads:
// RT/128 groups of 128 column lanes split the reduction range, partial sums meet in LDS (part: RT doubles)
__device__ void blk_symv(double* y, const double* M, const double* v, int n, double* vs, double* part) {
    const int tid = threadIdx.x, g = tid >> 7, jl = tid & 127;
    __syncthreads();
    for (int i = tid; i < n; i += RT) vs[i] = v[i];
    __syncthreads();
    const int q = (n + RT / 128 - 1) / (RT / 128), i0 = g * q, i1 = min(n, i0 + q);
    for (int j0 = 0; j0 < n; j0 += 128) {
        const int j = j0 + jl;
        double acc = 0.0;
        if (j < n) {
            // every instance reads the same shared matrix: start each one at a different row so that the workgroups of
            // an XCD do not queue on the same L2 lines at the same time
            const double* mj = M + j;
            const int im = i0 + (int)((blockIdx.x * 11u) % (unsigned)max(i1 - i0, 1));
#pragma unroll 8
            for (int i = im; i < i1; ++i) acc += mj[(size_t)i * n] * vs[i];
#pragma unroll 8
            for (int i = i0; i < im; ++i) acc += mj[(size_t)i * n] * vs[i];
        }
        part[tid] = acc;
        __syncthreads();
        if (g == 0 && j < n) {
            double sum = 0.0;
#pragma unroll
            for (int e = 0; e < RT / 128; ++e) sum += part[128 * e + jl];
            y[j] = sum;
        }
        __syncthreads();
    }
}

// the Cholesky factor of S is stored packed: row i holds its i+1 entries at LIX(i, 0)
#define LIX(i, j) ((i) * ((i) + 1) / 2 + (j))

// wave 0, vector in LDS: v <- L^-1 v / v <- L^-T v for the k x k lower factor L (packed)
__device__ void w0_fwd(const double* L, int k, double* v, int lane) {
    for (int c = 0; c < k; ++c) {
        const double yc = v[c] / L[LIX(c, c)];
        wave_sync();
        if (lane == 0) v[c] = yc;
        for (int i = c + 1 + lane; i < k; i += 64) v[i] -= L[LIX(i, c)] * yc;
        wave_sync();
    }
}

__device__ void w0_bwd(const double* L, int k, double* v, int lane) {
    for (int c = k - 1; c >= 0; --c) {
        const double yc = v[c] / L[LIX(c, c)];
        wave_sync();
        if (lane == 0) v[c] = yc;
        for (int i = lane; i < c; i += 64) v[i] -= L[LIX(c, i)] * yc;
        wave_sync();
    }
}

// wave 0: remove member j of the active list
__device__ void w0_remove(int* act, double* lam, int j, int k, int lane) {
    for (int base = j; base < k - 1; base += 64) {
        const int a = base + lane;
        int na = 0; double nl = 0.0;
        if (a < k - 1) { na = act[a + 1]; nl = lam[a + 1]; }
        wave_sync();
        if (a < k - 1) { act[a] = na; lam[a] = nl; }
        wave_sync();
    }
}

// L <- chol(Gram[act, act]); a member whose pivot falls below 1e-12 of its diagonal is dependent on the ones before
// it and is removed.  Returns the (possibly reduced) k.
__device__ int rows_rebuild(const double* Gram, int m, double* L, int* act, double* lam, int k, int* sflag, double thr = 1e-12) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (;;) {
        __syncthreads();
        for (int idx = tid; idx < k * k; idx += RT) {
            const int a = idx / k, c = idx % k;
            if (c <= a) L[LIX(a, c)] = Gram[(size_t)act[a] * m + act[c]];
        }
        __syncthreads();
        if (wave == 0) {
            int bad = -1;
            for (int j = 0; j < k; ++j) {
                const double dj = L[LIX(j, j)];
                const double d0 = Gram[(size_t)act[j] * m + act[j]];
                if (!(dj > thr * d0)) { bad = j; break; }
                const double sj = sqrt(dj), inv = 1.0 / sj;
                wave_sync();
                if (lane == 0) L[LIX(j, j)] = sj;
                for (int i = j + 1 + lane; i < k; i += 64) L[LIX(i, j)] *= inv;
                wave_sync();
                const int rem = k - j - 1;
                for (int idx = lane; idx < rem * rem; idx += 64) {
                    const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
                    if (c <= i) L[LIX(i, c)] -= L[LIX(i, j)] * L[LIX(c, j)];
                }
                wave_sync();
            }
            if (lane == 0) *sflag = bad;
        }
        __syncthreads();
        const int bad = *sflag;
        if (bad < 0) return k;
        if (wave == 0) w0_remove(act, lam, bad, k, lane);
        --k;
        if (k == 0) { __syncthreads(); return 0; }
    }
}

__global__ __launch_bounds__(RT) void rows_qp_kernel(RowsParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[RT / 64];
    __shared__ int redi[RT / 64];
    __shared__ int sflag;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = p.n, m = p.m, KL = p.KL, qi = blockIdx.x;
    double* L = reinterpret_cast<double*>(smem);        // packed lower Cholesky factor of S = Gram[act, act], up to KL rows
    double* lv = L + (size_t)KL * (KL + 1) / 2 + (((size_t)KL * (KL + 1) / 2) & 1);
    double* rv = lv + KL;
    double* lam = rv + KL;
    double* dl = lam + KL;
    double* vs = dl + KL;                                // n: matvec operand
    double* part = vs + n + (n & 1);                     // RT: matvec partial sums
    int* act = reinterpret_cast<int*>(part + RT);
    const SharedPtrs s = shared_ptrs(p.ws0, n, m);
    double* c = p.inst + (size_t)qi * p.l.stride;
    double* res = c + p.l.o_res; const double* cs = c + p.l.o_cs; const double* qs = c + p.l.o_qs;
    double* const res_global = res;
    if (p.res_lds) {
        // The residuals are read and rewritten by every step: two of the (k + 4) rows of m doubles a step streams through this
        // one CU.  When the launch is one workgroup per CU anyway they fit into the LDS the first pass leaves free.
        double* rl = reinterpret_cast<double*>(smem + (((size_t)((char*)(act + KL) - smem) + 15) & ~(size_t)15));
        if (p.res_compact) {
            // res_r = (Cs xu)_r - cs_r; the row stacked as the negative of r: res_r' = -(Cs xu)_r - cs_r'
            const int nw = p.mirok[1]; const bool um = p.mirok[0] != 0;
            const double2* pr2 = reinterpret_cast<const double2*>(res_global);
            const double2* cs2 = reinterpret_cast<const double2*>(cs);
            double2* rl2 = reinterpret_cast<double2*>(rl);
            for (int i = tid; i < nw; i += RT) {
                const int q = p.mlist[i];
                const double2 v = pr2[i], c0 = cs2[q];
                rl2[q] = double2{v.x - c0.x, v.y - c0.y};
                const int mq = um ? p.mir2[q] : -1;
                if (mq >= 0) { const double2 c1 = cs2[mq]; rl2[mq] = double2{-v.x - c1.x, -v.y - c1.y}; }
            }
        } else
        for (int r = tid; r < m; r += RT) rl[r] = res_global[r];
        res = rl;
        __syncthreads();
    }
    const double* xu = c + p.l.o_xu; double* x = c + p.l.o_x; double* t1 = c + p.l.o_t1; double* dv = c + p.l.o_dv;

#ifdef ROWS_PROF
    long long tk0 = wall_clock64(), t_arg = 0, t_dir = 0, t_upd = 0, t_book = 0, tk;
#define TICK(acc) { tk = wall_clock64(); acc += tk - tk0; tk0 = tk; }
#else
#define TICK(acc)
#endif
    int status = (int)s.W0[1];      // 0 optimal, 1 step cap, 2 infeasible, 3 numerical (P not positive definite, or more than KL rows bind)
    int k = 0, steps = 0;
    // hand-over between the passes: sact[0] = members saved by pass 1 (-1: none), sact[1] = its step count, sact[2..] = rows
    double* slam = c + p.l.o_slam;
    int* sact = reinterpret_cast<int*>(c + p.l.o_sact);
    // the set the previous solve of this instance ended with (wact[0] members, -1: none): the start of a warm solve
    int* wact = reinterpret_cast<int*>(c + p.l.o_wact);
    bool handed_over = false;
    if (p.pass == 1 && p.warm && status == 0 && wact[0] > 0) {
        // Warm start: take the kept set as the working set.  lam = S^-1 (N xu - cs)_A is the multiplier of "these rows hold with
        // equality"; members it gives a wrong sign would be pulled the wrong way and leave (all at once, then again): what is left
        // is the minimiser over a set of equalities with lam >= 0, a state the iteration below can continue from exactly as
        // from one it built itself.  The residuals of all rows follow in one sweep, like one step's update.
        k = min(wact[0], KL);
        for (int a = tid; a < k; a += RT) { act[a] = wact[1 + a]; lam[a] = 0.0; }
        for (;;) {
            // a kept set may sit on a degenerate vertex (the iteration accepts pivots down to 1e-13): as a START only members
            // that are clearly independent are taken (pivot > 1e-8 of the row's diagonal) -- the iteration adds what is missing
            k = rows_rebuild(p.Gram, m, L, act, lam, k, &sflag, 1e-8);
            if (k == 0) break;
            for (int a = tid; a < k; a += RT) lam[a] = res[act[a]];
            __syncthreads();
            if (wave == 0) {
                w0_fwd(L, k, lam, lane); w0_bwd(L, k, lam, lane);
            }
            __syncthreads();
            // the multipliers must reproduce the right-hand side (S lam = rhs to 1e-9): if they do not, start cold
            double bad = 0.0, big = 0.0;
            for (int a = tid; a < k; a += RT) {
                double v = 0.0;
                for (int b = 0; b < k; ++b) v += p.Gram[(size_t)act[a] * m + act[b]] * lam[b];
                bad = fabs(v - res[act[a]]); big = fabs(res[act[a]]);
            }
            double o; int oi;
            blk_argmax(bad, 0, red, redi, o, oi); bad = o;
            blk_argmax(big, 0, red, redi, o, oi); big = o;
            __syncthreads();
            if (!(bad <= 1e-9 * fmax(big, 1e-300))) { k = 0; break; }
            if (wave == 0 && lane == 0) {
                int kk = 0;
                for (int a = 0; a < k; ++a) if (lam[a] > 0.0) { act[kk] = act[a]; ++kk; }
                sflag = kk;
            }
            __syncthreads();
            const int kk = sflag;
            __syncthreads();
            if (kk == k) break;
            k = kk;
            if (k == 0) break;
        }
        if (k > 0)
            for (int r = tid; r < m; r += RT) {
                double v = res[r];
                for (int a = 0; a < k; ++a) v -= lam[a] * p.Gram[(size_t)act[a] * m + r];
                res[r] = v;
            }
        __syncthreads();
    }
    if (p.pass == 2) {
        if (status != 0 || p.status[qi] != 3 || sact[0] < 0) return;       // uniform over the workgroup
        k = sact[0]; steps = sact[1];
        for (int a = tid; a < k; a += RT) { act[a] = sact[2 + a]; lam[a] = slam[a]; }
        k = rows_rebuild(p.Gram, m, L, act, lam, k, &sflag);
    }

    // the scan for the most violated row rides on the residual update of a full step (fbest, fbi: this thread's candidate
    // among the rows it has just rewritten), so only the first one reads the residuals for its own sake
    bool fused = false;
    double fbest = 0.0; int fbi = -1;
    const bool usem = (m & 1) == 0 && p.mirok[0] != 0;      // uniform: rows stacked with their negatives, pair by pair
    const int nwork = (m & 1) == 0 ? p.mirok[1] : 0;       // pairs a sweep visits (all m / 2 without mirrors)
    while (status == 0) {
        // the most violated row that is not in the active set
        double best = 0.0; int bi = -1;
        if (fused) { best = fbest; bi = fbi; }
        else
            for (int r = tid; r < m; r += RT) {
                const double v = res[r];
                if (v > 1e-12 * fmax(1.0, fabs(cs[r])) && (bi < 0 || v > best)) {
                    bool member = false;
                    for (int a = 0; a < k; ++a) member |= act[a] == r;
                    if (!member) { best = v; bi = r; }
                }
            }
        fused = false;
        double v0; int pp;
        blk_argmax(best, bi, red, redi, v0, pp);
        TICK(t_arg)
        if (pp < 0) break;
        const double* Grow = p.Gram + (size_t)pp * m;
        const double nHn = Grow[pp];
        double accl = 0.0;
        bool added = false;
        while (!added) {
            if (++steps > p.max_steps) { status = 1; break; }
            __syncthreads();
            for (int a = tid; a < k; a += RT) lv[a] = Grow[act[a]];
            __syncthreads();
            if (wave == 0) {
                w0_fwd(L, k, lv, lane);
                for (int a = lane; a < k; a += 64) rv[a] = lv[a];
                wave_sync();
                w0_bwd(L, k, rv, lane);
            }
            __syncthreads();
            // every wave evaluates the step lengths redundantly (k <= KL): identical arithmetic, uniform control flow
            double part = 0.0;
            for (int a = lane; a < k; a += 64) part += lv[a] * lv[a];
            const double d2 = nHn - wave_sum(part);
            const bool dep = !(d2 > 1e-13 * nHn);
            const double t2 = dep ? 1e300 : v0 / d2;
            double tr = 1e300; int jb = -1;
            for (int a = lane; a < k; a += 64)
                if (rv[a] > 0.0) {
                    const double t = lam[a] / rv[a];
                    if (t < tr) { tr = t; jb = a; }
                }
#pragma unroll
            for (int o = 32; o; o >>= 1) {
                const double ot = __shfl_xor(tr, o); const int oj = __shfl_xor(jb, o);
                if (oj >= 0 && (jb < 0 || ot < tr || (ot == tr && oj < jb))) { tr = ot; jb = oj; }
            }
            TICK(t_dir)
            if (jb < 0 && dep) { status = 2; break; }           // no step possible: infeasible
            const bool full = !dep && t2 <= tr;
            const double t = full ? t2 : tr;
            if (!dep) {
                // after a full step with room for p the active set is act + {p}: the candidates of the next scan
                const bool scan = full && k < KL;
                fbest = 0.0; fbi = -1;
                if ((m & 1) == 0) {
                    // The sweep is a stream of (k + 2) rows of 8 m bytes through ONE CU, and what bounds it is how many bytes that CU
                    // has in flight: with 8-byte loads and one active row at a time, 64 KB -- 25-40 GB/s at the latency of a
                    // miss, 30 us per step at full size.  Here: 16-byte loads (row pairs; rows of an even m are 16-byte aligned)
                    // and FOUR active rows' loads issued before the first is used: 256 KB in flight.  Same sums in the same order.
                    const double2* Grow2 = reinterpret_cast<const double2*>(Grow);
                    double2* res2 = reinterpret_cast<double2*>(res);
                    // (over the working pairs: of a pair and the pair of its negatives the one with the lower index reads Gram and
                    // updates both)
                    for (int i0 = 0; i0 < nwork; i0 += 4 * RT) {
                        double2 sv[4]; int q[4], qa[4], mq[4]; bool work[4];
#pragma unroll
                        for (int e = 0; e < 4; ++e) {
                            const int i = i0 + e * RT + tid;
                            work[e] = i < nwork;
                            q[e] = p.mlist[min(i, nwork - 1)];
                            mq[e] = usem ? p.mir2[q[e]] : -1;
                            qa[e] = q[e];
                            sv[e] = Grow2[qa[e]];
                        }
                        int a = 0;
                        for (; a + 4 <= k; a += 4) {
                            double2 g[4][4];
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const double2* ga = reinterpret_cast<const double2*>(p.Gram + (size_t)act[a + u] * m);
#pragma unroll
                                for (int e = 0; e < 4; ++e) g[u][e] = ga[qa[e]];
                            }
#pragma unroll
                            for (int u = 0; u < 4; ++u) {
                                const double cf = rv[a + u];
#pragma unroll
                                for (int e = 0; e < 4; ++e) { sv[e].x -= cf * g[u][e].x; sv[e].y -= cf * g[u][e].y; }
                            }
                        }
                        for (; a < k; ++a) {
                            const double cf = rv[a];
                            const double2* ga = reinterpret_cast<const double2*>(p.Gram + (size_t)act[a] * m);
                            double2 g[4];
#pragma unroll
                            for (int e = 0; e < 4; ++e) g[e] = ga[qa[e]];
#pragma unroll
                            for (int e = 0; e < 4; ++e) { sv[e].x -= cf * g[e].x; sv[e].y -= cf * g[e].y; }
                        }
                        // the rows this thread has just rewritten are its candidates for the next scan
                        auto cand = [&](double v, int r) {
                            // (the row's tolerance 1e-12 max(1, |cs_r|) is looked up only for a row that passes the others:
                            // a handful per sweep instead of one more stream of m doubles)
                            if (scan && v > 1e-12 && (fbi < 0 || v > fbest) && r != pp && v > 1e-12 * fabs(cs[r])) {
                                bool member = false;
                                for (int a2 = 0; a2 < k; ++a2) member |= act[a2] == r;
                                if (!member) { fbest = v; fbi = r; }
                            }
                        };
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (work[e]) {
                                const double2 rr = res2[q[e]];
                                const double v0 = rr.x - t * sv[e].x, v1 = rr.y - t * sv[e].y;
                                res2[q[e]] = double2{v0, v1};
                                cand(v0, 2 * q[e]); cand(v1, 2 * q[e] + 1);
                                if (mq[e] >= 0) {           // their negatives: the same update with the other sign
                                    const double2 rm = res2[mq[e]];
                                    const double w0 = rm.x + t * sv[e].x, w1 = rm.y + t * sv[e].y;
                                    res2[mq[e]] = double2{w0, w1};
                                    cand(w0, 2 * mq[e]); cand(w1, 2 * mq[e] + 1);
                                }
                            }
                    }
                } else
                for (int r0 = 0; r0 < m; r0 += 8 * RT) {
                    double sv[8]; int r[8];
#pragma unroll
                    for (int e = 0; e < 8; ++e) { r[e] = r0 + e * RT + tid; sv[e] = r[e] < m ? Grow[r[e]] : 0.0; }
                    for (int a = 0; a < k; ++a) {
                        const double cf = rv[a];
                        const double* ga = p.Gram + (size_t)act[a] * m;
#pragma unroll
                        for (int e = 0; e < 8; ++e) if (r[e] < m) sv[e] -= cf * ga[r[e]];
                    }
#pragma unroll
                    for (int e = 0; e < 8; ++e)
                        if (r[e] < m) {
                            const double v = res[r[e]] - t * sv[e];
                            res[r[e]] = v;
                            if (scan && v > 1e-12 * fmax(1.0, fabs(cs[r[e]])) && (fbi < 0 || v > fbest) && r[e] != pp) {
                                bool member = false;
                                for (int a = 0; a < k; ++a) member |= act[a] == r[e];
                                if (!member) { fbest = v; fbi = r[e]; }
                            }
                        }
                }
                fused = scan;
            }
            __syncthreads();
            TICK(t_upd)
            if (wave == 0) {
                for (int a = lane; a < k; a += 64) lam[a] -= t * rv[a];
                wave_sync();
            }
            accl += t;
            if (full) {
                if (k >= KL) {
                    status = 3;
                    if (p.pass == 1) {      // the active set, with the member that did not fit, goes to the second pass
                        __syncthreads();
                        for (int a = tid; a < k; a += RT) { sact[2 + a] = act[a]; slam[a] = lam[a]; }
                        if (tid == 0) { sact[2 + k] = pp; slam[k] = accl; sact[0] = k + 1; sact[1] = steps; }
                        if (p.res_lds) for (int r = tid; r < m; r += RT) res_global[r] = res[r];      // the second pass reads them from memory
                        handed_over = true;
                    }
                    break;
                }
                if (wave == 0) {
                    for (int a = lane; a < k; a += 64) L[LIX(k, a)] = lv[a];
                    if (lane == 0) { L[LIX(k, k)] = sqrt(d2); act[k] = pp; lam[k] = accl; }
                }
                ++k;
                added = true;
                __syncthreads();
            } else {
                if (!dep) v0 -= t * d2;
                if (wave == 0) w0_remove(act, lam, jb, k, lane);
                --k;
                k = rows_rebuild(p.Gram, m, L, act, lam, k, &sflag);
            }
            TICK(t_book)
        }
    }
    __syncthreads();
#ifdef ROWS_PROF
    long long t_fin0 = wall_clock64(), t_x = 0, t_sv = 0, t_nl = 0, t_rows = 0, t_out = 0;
    tk0 = t_fin0;
#endif

    // ---- x = xu - CH[act,:]' lam, then refinement on the final active set with Ps / Cs themselves ----------------------
    for (int j = tid; j < n; j += RT) {
        double v = xu[j];
        for (int a = 0; a < k; ++a) v -= lam[a] * p.CH[(size_t)act[a] * n + j];
        x[j] = v;
    }
    __syncthreads();
    TICK(t_x)
    // Is (x, lam) already optimal to rounding level?  x = xu - Hi N_A' lam holds by construction, so the stationarity
    // residual is g = (I - Ps Hi)(qs + N' lam): bounded by eps |qs + N' lam|_inf with eps = |I - Ps Hi|_inf from the set-up,
    // and the rows of the active set can be checked directly -- no matvec with a shared n x n matrix, which is what this
    // tail used to spend its time on (every workgroup of the batch streaming the same 2.9 MB matrices through L2, three
    // times).  Only if the bound does not certify it is g itself formed (one matvec), and only if g is not small either does
    // a refinement round run (two more).
    const double eps = s.W0[2], tol = 1e-11;
    int rounds = 0;
    bool have_t1 = false;
    double rm_last = 0.0, cm_last = 0.0, qm_last = 0.0;
    for (;;) {
        double wm = 0.0, qm = 0.0;
        for (int j = tid; j < n; j += RT) {                 // w = qs + N' lam
            double v = qs[j];
            for (int a = 0; a < k; ++a) v += s.Cs[(size_t)act[a] * n + j] * lam[a];
            dv[j] = v;
            wm = fmax(wm, fabs(v)); qm = fmax(qm, fabs(qs[j]));
        }
        double rm = 0.0, cm = 0.0;
        for (int a = wave; a < k; a += RT / 64) {           // N x - cs on the active rows; lam_a (N xu - cs)_a for the objective
            const double* na = s.Cs + (size_t)act[a] * n;
            double v = 0.0, v0 = 0.0;
            for (int j = lane; j < n; j += 64) { v += na[j] * x[j]; v0 += na[j] * xu[j]; }
            v = wave_sum(v) - cs[act[a]]; v0 = wave_sum(v0) - cs[act[a]];
            rm = fmax(rm, fabs(v)); cm = fmax(cm, fabs(cs[act[a]]));
            if (lane == 0) { dl[a] = v; rv[a] = lam[a] * v0; }
        }
        // block maxima through the arg-max reduction (values only)
        double o; int oi;
        blk_argmax(wm, 0, red, redi, o, oi); wm = o;
        blk_argmax(qm, 0, red, redi, o, oi); qm = o;
        blk_argmax(rm, 0, red, redi, o, oi); rm = o;
        blk_argmax(cm, 0, red, redi, o, oi); cm = o;
        __syncthreads();
        rm_last = rm; cm_last = cm; qm_last = qm;
        TICK(t_nl)
        if (status != 0 || p.refine == 0) break;
        const bool rows_ok = rm <= tol * fmax(1.0, cm);
        if (rows_ok && eps * wm <= tol * fmax(1.0, qm)) break;                   // certified by the bound
        blk_symv(t1, s.Ps, x, n, vs, part);
        have_t1 = true;
        TICK(t_sv)
        double gm = 0.0;
        for (int j = tid; j < n; j += RT) { const double v = t1[j] + dv[j]; dv[j] = v; gm = fmax(gm, fabs(v)); }     // g itself
        blk_argmax(gm, 0, red, redi, o, oi); gm = o;
        __syncthreads();
        if ((rows_ok && gm <= tol * fmax(1.0, qm)) || rounds >= p.refine) break;
        ++rounds;
        have_t1 = false;
        blk_symv(dv, s.Hi, dv, n, vs, part);                // dv = Hi g  (the operand is staged in LDS before dv is written)
        TICK(t_sv)
        for (int a = wave; a < k; a += RT / 64) {           // N (x - dv) - cs = (N x - cs) - N dv on the active rows
            const double* na = s.Cs + (size_t)act[a] * n;
            double v = 0.0;
            for (int j = lane; j < n; j += 64) v += na[j] * dv[j];
            v = wave_sum(v);
            if (lane == 0) dl[a] -= v;
        }
        __syncthreads();
        if (wave == 0) { w0_fwd(L, k, dl, lane); w0_bwd(L, k, dl, lane); }
        __syncthreads();
        for (int j = tid; j < n; j += RT) {
            double v = -dv[j];
            for (int a = 0; a < k; ++a) v -= p.CH[(size_t)act[a] * n + j] * dl[a];
            x[j] += v;
        }
        __syncthreads();
        if (wave == 0) { for (int a = lane; a < k; a += 64) lam[a] += dl[a]; }
        __syncthreads();
        TICK(t_rows)
    }

    // What is about to be called SOLVED must hold its own active rows: a set that came to include a (numerically) dependent
    // row -- an infeasible instance whose last row passed the dependency test by a hair -- carries multipliers of 1e15, and
    // x = xu - CH' lam is then rounding noise that no refinement round repairs.  Such an instance is reported infeasible when
    // its multipliers have run away, as a numerical failure otherwise; never as solved.
    if (status == 0 && k > 0 && !(rm_last <= 1e-7 * fmax(1.0, cm_last))) {
        double lm = 0.0;
        for (int a = tid; a < k; a += RT) lm = fmax(lm, fabs(lam[a]));
        double o; int oi;
        blk_argmax(lm, 0, red, redi, o, oi);
        __syncthreads();
        status = o > 1e9 * fmax(1.0, fmax(qm_last, cm_last)) ? 2 : 3;
    }

    // ---- outputs ----------------------------------------------------------------------------------------------------------
    double* xo = p.x + (size_t)qi * n;
    for (int j = tid; j < n; j += RT) xo[j] = status == 3 ? 0.0 : x[j] * s.d[j];
    if (p.z) for (int r = tid; r < m; r += RT) p.z[(size_t)qi * m + r] = 0.0;
    __syncthreads();
    if (p.z) for (int a = tid; a < k; a += RT) p.z[(size_t)qi * m + act[a]] = lam[a] / s.rn[act[a]];
    __syncthreads();
    TICK(t_out)
    // objective: at an untouched active-set solution f(x) = f(xu) + 1/2 lam' S lam = 1/2 qs' xu + 1/2 sum_a lam_a (N xu - cs)_a;
    // after a refinement round, or for an instance that did not finish, from x itself
    if (!have_t1 && (status != 0 || rounds > 0)) { blk_symv(t1, s.Ps, x, n, vs, part); have_t1 = true; }
    double po = 0.0;
    if (have_t1) { for (int j = tid; j < n; j += RT) po += x[j] * (0.5 * t1[j] + qs[j]); }
    else {
        for (int j = tid; j < n; j += RT) po += 0.5 * qs[j] * xu[j];
        for (int a = tid; a < k; a += RT) po += 0.5 * rv[a];
    }
    const double obj = blk_sum(po, red);
#ifdef ROWS_PROF
    if (tid == 0 && (qi & 63) == 0)
        printf("qp %d steps %d k %d: argmax %lld dir %lld update %lld book %lld final %lld = x %lld symv %lld nl %lld rows %lld out %lld (x10 ns)\n",
               qi, steps, k, t_arg, t_dir, t_upd, t_book, wall_clock64() - t_fin0, t_x, t_sv, t_nl, t_rows, t_out);
#endif
    if (!handed_over) {         // the final working set, for a warm start of the next solve (a handed-over instance: pass 2 writes it)
        for (int a = tid; a < k; a += RT) wact[1 + a] = act[a];
        if (tid == 0) wact[0] = status == 0 ? k : -1;
    }
    if (tid == 0) {
        if (!handed_over) sact[0] = -1;
        if (p.obj) p.obj[qi] = obj;
        if (p.status) p.status[qi] = status;
        if (p.iters) p.iters[qi] = steps;
    }
}

// ---- rows that come with their negatives -----------------------------------------------------------------------------
// The reference stacks every two-sided bound as a row and its negative (LaunchPlannerQPFormulation.py:96-118: +-I, +-rows of
// the prediction matrix), so half of Gram's columns are the other half with the sign flipped: Gram[a][r'] = -Gram[a][r]
// bit for bit when Cs[r'] = -Cs[r] (same products, same order).  The sweep of a step then needs only ONE of the two
// columns: half the bytes of what a step streams.  Found from Gram itself, so any caller of the set-up gets it:
// Cs[r'] = -Cs[r]  <=>  (Cs[r] + Cs[r'])' Hi (Cs[r] + Cs[r']) = G_rr + 2 G_rr' + G_r'r' = 0 (Hi positive definite), tested
// as G_rr' == -G_rr and G_r'r' == G_rr exactly, then confirmed on the two rows of Cs entry by entry -- a pair that misses by
// rounding is simply not used.
__global__ __launch_bounds__(256) void rows_mirror_find_kernel(const double* Gram, const double* Cs, int n, int m, int* mir) {
    __shared__ int best, differs;
    const int r = blockIdx.x;
    if (threadIdx.x == 0) { best = 0x7fffffff; differs = 0; }
    __syncthreads();
    const double* g = Gram + (size_t)r * m;
    const double drr = g[r];
    if (drr > 0.0)
        for (int c = threadIdx.x; c < m; c += 256)
            if (c != r && g[c] == -drr && Gram[(size_t)c * m + c] == drr) atomicMin(&best, c);
    __syncthreads();
    const int c = best;
    if (c != 0x7fffffff)        // the candidate's row itself, entry by entry: nothing is taken on the strength of two equal sums
        for (int j = threadIdx.x; j < n; j += 256)
            if (Cs[(size_t)c * n + j] != -Cs[(size_t)r * n + j]) differs = 1;
    __syncthreads();
    if (threadIdx.x == 0) mir[r] = (c == 0x7fffffff || differs) ? -1 : c;
}
// the sweep works on row PAIRS (16-byte loads): usable only if mirrors pair up the same way -- (2q, 2q+1) <-> (2q', 2q'+1),
// mutually -- for every pair; otherwise *ok = 0 and the sweep reads every column
__global__ __launch_bounds__(256) void rows_mirror_pairs_kernel(const int* mir, int m, int* mir2, int* ok) {
    const int q = blockIdx.x * 256 + threadIdx.x;
    if (q >= m / 2) return;
    const int a = mir[2 * q], b = mir[2 * q + 1];
    const bool none = a < 0 && b < 0;
    const bool pair = a >= 0 && (a & 1) == 0 && a + 1 < m && b == a + 1 && mir[a] == 2 * q && mir[a + 1] == 2 * q + 1;
    mir2[q] = pair ? a / 2 : -1;
    if (!none && !pair) atomicAnd(ok, 0);
}

// the pairs a sweep has to visit (every pair without a mirror, and of two mirrored pairs the lower one), ascending, and
// their number in okc[1]: one workgroup, every thread a contiguous run of pairs
__global__ __launch_bounds__(1024) void rows_mirror_list_kernel(const int* mir2, int m2, int* okc, int* list) {
    __shared__ int cnt[1024];
    const int tid = threadIdx.x, per = (m2 + 1023) / 1024, lo = min(tid * per, m2), hi = min(lo + per, m2);
    const bool usem = okc[0] != 0;
    int c = 0;
    for (int q = lo; q < hi; ++q) c += !(usem && mir2[q] >= 0 && mir2[q] < q);
    cnt[tid] = c;
    __syncthreads();
    for (int s = 1; s < 1024; s <<= 1) {
        const int v = tid >= s ? cnt[tid - s] : 0;
        __syncthreads();
        cnt[tid] += v;
        __syncthreads();
    }
    int o = cnt[tid] - c;
    for (int q = lo; q < hi; ++q) if (!(usem && mir2[q] >= 0 && mir2[q] < q)) list[o++] = q;
    if (tid == 1023) okc[1] = cnt[1023];
}

// out = max_i sum_j |E_ij - delta_ij| (one wave per row; non-negative doubles order like their bit patterns)
__global__ __launch_bounds__(256) void inv_resid_kernel(const double* E, int n, double* out) {
    const int lane = threadIdx.x & 63, i = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (i >= n) return;
    double s = 0.0;
    for (int j = lane; j < n; j += 64) s += fabs(E[(size_t)i * n + j] - (i == j ? 1.0 : 0.0));
    s = wave_sum(s);
    if (!(s == s)) s = 1e300;
    if (lane == 0) atomicMax(reinterpret_cast<unsigned long long*>(out), (unsigned long long)__double_as_longlong(s));
}

}  // namespace

// general.hip: C (M, N) = alpha A (M, K) B (N, K)' + bias[j]
int gemm_nt_bias(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K, double alpha,
                 const double* bias, hipStream_t st) {
    return launch_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, alpha, 0.0, st, bias);
}

// blocked_setup.hip: C (M, N) = alpha A (M, K) B (N, K)' + beta C
int gemm_nt(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K, double alpha,
            double beta, hipStream_t st) {
    return launch_gemm_nt(A, lda, B, ldb, C, ldc, M, N, K, alpha, beta, st);
}

extern "C" {

long mpcqp_rows_workspace(int n, int m, long* per_instance) {
    if (n < 1 || m < 1) return 0;
    const Layout l = make_layout(n, m);
    if (per_instance) *per_instance = l.stride;
    return l.nd0 + (long)m * n + (long)m * m + mirror_doubles(m);
}

int mpcqp_rows_setup(int device_id, int n, int m, const double* P, const double* G, double* shared_ws, int* int_ws,
                     void* stream) {
    if (n < 1 || m < 1 || !P || !G || !shared_ws || !int_ws) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Layout l = make_layout(n, m);
    int rc = dense_shared_setup_sigma(n, m, P, G, shared_ws, int_ws, 0.0, st);
    if (rc != MPCQP_OK) return rc;
    const SharedPtrs s = shared_ptrs(shared_ws, n, m);
    double* CH = shared_ws + l.nd0;
    double* Gram = CH + (size_t)m * n;
    rc = launch_gemm_nt(s.Cs, n, s.Hi, n, CH, n, m, n, n, 1.0, 0.0, st);          // CH = Cs Hi   (Hi symmetric)
    if (rc != MPCQP_OK) return rc;
    rc = launch_gemm_nt(CH, n, s.Cs, n, Gram, m, m, m, n, 1.0, 0.0, st);            // Gram = CH Cs'
    if (rc != MPCQP_OK) return rc;
    {   // rows stacked with their negatives (one more pass over Gram, once per model)
        int* mir = reinterpret_cast<int*>(Gram + (size_t)m * m);
        int* mir2 = mir + m;
        int* ok = mir2 + m / 2 + 1;
        const int one = (m & 1) ? 0 : 1;
        if (hipMemcpyAsync(ok, &one, sizeof(int), hipMemcpyHostToDevice, st) != hipSuccess) return MPCQP_EHIP;
        hipLaunchKernelGGL(rows_mirror_find_kernel, dim3(m), dim3(256), 0, st, Gram, s.Cs, n, m, mir);
        if (m >= 2) hipLaunchKernelGGL(rows_mirror_pairs_kernel, dim3(ceil_div(m / 2, 256)), dim3(256), 0, st, mir, m, mir2, ok);
        hipLaunchKernelGGL(rows_mirror_list_kernel, dim3(1), dim3(1024), 0, st, mir2, m / 2, ok, ok + 4);
        if (hipGetLastError() != hipSuccess) return MPCQP_EHIP;
    }
    // eps = |I - Ps Hi|_inf into the third entry of QP 0's W block (beside sigma and the status): what the solve's tail
    // needs to certify stationarity without a matvec.  The product goes to QP 0's T block, free once the inverse exists
    double* W0 = shared_ws + (size_t)2 * n * n;
    double* E = W0 + (size_t)n * n;
    rc = launch_gemm_nt(s.Ps, n, s.Hi, n, E, n, n, n, n, 1.0, 0.0, st);
    if (rc != MPCQP_OK) return rc;
    if (hipMemsetAsync(W0 + 2, 0, sizeof(double), st) != hipSuccess) return MPCQP_EHIP;
    hipLaunchKernelGGL(inv_resid_kernel, dim3(ceil_div(n, 4)), dim3(256), 0, st, E, n, W0 + 2);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

struct MpcMaps { const double *x0, *F, *f0, *E, *h0; int nx; };

static int rows_solve_impl(int device_id, int batch, int n, int m, const double* q, const double* h, const MpcMaps* mp,
                           double* x, double* z, double* obj, int* status, int* iters, const double* shared_ws,
                           double* instance_ws, void* stream, int warm = 0) {
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const Layout l = make_layout(n, m);
    const SharedPtrs s = shared_ptrs(shared_ws, n, m);
    const double* CH = shared_ws + l.nd0;
    const double* Gram = CH + (size_t)m * n;
    int rc;
    if (mp) {
        // qs = (x0 F' + f0) d ;  cs = (h0 - x0 E') / rn : two thin GEMMs (K = nx) with the scaling in their epilogue
        rc = launch_gemm_nt(mp->x0, mp->nx, mp->F, mp->nx, instance_ws + l.o_qs, l.stride, batch, n, mp->nx, 1.0, 0.0, st,
                            mp->f0, s.d, 0);
        if (rc != MPCQP_OK) return rc;
        rc = launch_gemm_nt(mp->x0, mp->nx, mp->E, mp->nx, instance_ws + l.o_cs, l.stride, batch, m, mp->nx, -1.0, 0.0, st,
                            mp->h0, s.rn, 1);
        if (rc != MPCQP_OK) return rc;
    } else {
        PrepParams pp;
        pp.q = q; pp.h = h; pp.ws0 = shared_ws; pp.inst = instance_ws; pp.l = l; pp.n = n; pp.m = m; pp.B = batch;
        const long total = ((long)n + m) * batch;
        hipLaunchKernelGGL(rows_prep_kernel, dim3((unsigned)std::min<long>((total + 255) / 256, 4096)), dim3(256), 0, st, pp);
        if (hipGetLastError() != hipSuccess) return MPCQP_EHIP;
    }
    RowsParams r;
    r.ws0 = shared_ws; r.CH = CH; r.Gram = Gram; r.inst = instance_ws; r.l = l;
    r.x = x; r.z = z; r.obj = obj; r.status = status; r.iters = iters;
    r.n = n; r.m = m; r.B = batch; r.max_steps = 20 * (n + m) + 100;
    r.warm = warm ? 1 : 0;
    r.mir2 = reinterpret_cast<const int*>(Gram + (size_t)m * m) + m; r.mirok = r.mir2 + m / 2 + 1; r.mlist = r.mirok + 4;
    r.refine = 1;       // Hi is the exact-arithmetic inverse here (no proximal term): one round squares its rounding error
    auto lds_of = [&](int KL) {
        const size_t tri = (size_t)KL * (KL + 1) / 2;
        return (tri + (tri & 1) + 4 * (size_t)KL + n + (n & 1) + RT) * sizeof(double) + (size_t)KL * sizeof(int);
    };
    const int KL1 = rows_kl1(n, m), KL2 = rows_kl2(n, m);
    size_t lds1 = lds_of(KL1);
    const size_t lds2 = lds_of(KL2);
    if (lds1 > 160 * 1024) return MPCQP_ENOTIMPL;
    // residuals in LDS: only when the launch is one round of one workgroup per CU either way (a larger batch wants the second
    // workgroup per CU more than the shorter sweep) and they fit behind the first pass's arrays
    static std::atomic<int> ncu_of[64];     // CUs per device, asked once (0: not yet; a failed query leaves the residuals in memory)
    if (device_id < 64 && ncu_of[device_id].load(std::memory_order_relaxed) == 0) {
        int v = 0;
        ncu_of[device_id].store((hipDeviceGetAttribute(&v, hipDeviceAttributeMultiprocessorCount, device_id) == hipSuccess && v > 0) ? v : -1,
                                std::memory_order_relaxed);
    }
    const int ncu = device_id < 64 ? ncu_of[device_id].load(std::memory_order_relaxed) : -1;
    const size_t lds1r = ((lds1 + 15) & ~(size_t)15) + (size_t)m * sizeof(double);
    r.res_lds = (batch <= ncu && lds1r <= 160 * 1024) ? 1 : 0;
    if (r.res_lds) lds1 = lds1r;
    // xu = -Hi qs ;  res = Cs xu - cs   (both over the whole batch: the shared matrices are read once, not once per QP)
    rc = launch_gemm_nt(instance_ws + l.o_qs, l.stride, s.Hi, n, instance_ws + l.o_xu, l.stride, batch, n, n, -1.0, 0.0, st);
    if (rc != MPCQP_OK) return rc;
    // with the residuals loaded into LDS by the kernel itself, the product is formed for the working row pairs only (rows
    // stacked with their negatives: half of it) and "- cs" moves into that load
    r.res_compact = (r.res_lds && launch_gemm_nt_pairs(instance_ws + l.o_xu, l.stride, s.Cs, n, instance_ws + l.o_res, l.stride, batch, m,
                                                       n, r.mlist, r.mirok + 1, st)) ? 1 : 0;
    if (!r.res_compact) {
        rc = launch_gemm_nt(instance_ws + l.o_xu, l.stride, s.Cs, n, instance_ws + l.o_res, l.stride, batch, m, n, 1.0, -1.0, st,
                            nullptr, nullptr, 0, instance_ws + l.o_cs);
        if (rc != MPCQP_OK) return rc;
    }
    if (hipGetLastError() != hipSuccess) return MPCQP_EHIP;
    const bool second = KL2 > KL1 && lds2 <= 160 * 1024;
    if (ensure_dyn_lds(device_id, reinterpret_cast<const void*>(&rows_qp_kernel), std::max(lds1, second ? lds2 : lds1)) != hipSuccess) return MPCQP_EHIP;
    r.KL = KL1; r.pass = 1;
    hipLaunchKernelGGL(rows_qp_kernel, dim3(batch), dim3(RT), lds1, st, r);
    if (hipGetLastError() != hipSuccess) return MPCQP_EHIP;
    if (second) {       // instances whose active set outgrew the first pass continue where they stopped; the others return at once
        r.KL = KL2; r.pass = 2; r.res_lds = 0; r.res_compact = 0;
        hipLaunchKernelGGL(rows_qp_kernel, dim3(batch), dim3(RT), lds2, st, r);
    }
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

int mpcqp_rows_solve(int device_id, int batch, int n, int m, const double* q, const double* h, double* x, double* z,
                     double* obj, int* status, int* iters, const double* shared_ws, double* instance_ws, void* stream) {
    if (batch < 1 || n < 1 || m < 1 || !q || !h || !x || !status || !shared_ws || !instance_ws) return MPCQP_EINVAL;
    return rows_solve_impl(device_id, batch, n, m, q, h, nullptr, x, z, obj, status, iters, shared_ws, instance_ws, stream);
}

int mpcqp_rows_solve_mpc(int device_id, int batch, int n, int m, int nx, const double* x0, const double* F, const double* f0,
                         const double* E, const double* h0, double* x, double* z, double* obj, int* status, int* iters,
                         const double* shared_ws, double* instance_ws, void* stream) {
    if (batch < 1 || n < 1 || m < 1 || nx < 1 || !x0 || !F || !f0 || !E || !h0 || !x || !status || !shared_ws || !instance_ws)
        return MPCQP_EINVAL;
    const MpcMaps mp{x0, F, f0, E, h0, nx};
    return rows_solve_impl(device_id, batch, n, m, nullptr, nullptr, &mp, x, z, obj, status, iters, shared_ws, instance_ws, stream);
}

int mpcqp_rows_solve_mpc_warm(int device_id, int batch, int n, int m, int nx, const double* x0, const double* F, const double* f0,
                              const double* E, const double* h0, double* x, double* z, double* obj, int* status, int* iters,
                              const double* shared_ws, double* instance_ws, void* stream) {
    if (batch < 1 || n < 1 || m < 1 || nx < 1 || !x0 || !F || !f0 || !E || !h0 || !x || !status || !shared_ws || !instance_ws)
        return MPCQP_EINVAL;
    const MpcMaps mp{x0, F, f0, E, h0, nx};
    return rows_solve_impl(device_id, batch, n, m, nullptr, nullptr, &mp, x, z, obj, status, iters, shared_ws, instance_ws, stream, 1);
}

}  // extern "C"
