// Set-up for the shared-Hessian box-QP solver: Jacobi scaling, the "cached dense Cholesky of
// (H + rho I)" the north star names, turned into explicit inverses so that the per-iteration
// triangular solves become one MFMA GEMM for the whole batch, and the MFMA fragment images.
//
// Reference counterpart: none as code -- cvxopt factors its KKT matrix inside every call of
// solvers.qp (LaunchPlannerQPFormulation.py:173); here the factorisation is done once per model.
//
// One 256-thread workgroup; this runs once per model, so clarity beats speed:
//   d = diag(H)^-1/2, Hs = d H d,  Hi = Hs^-1,  rho (given or 0.1 sqrt(lmax lmin) from norm bounds),
//   M = (Hs + rho I)^-1,  Fs1 = [d F | d f0],  Ku1 = -Hi Fs1,  Kb1 = -M Fs1,  lbs = lb/d, ubs = ub/d.
#include "common.h"
#include <algorithm>

namespace {

struct FacParams {
    const double *H, *F, *f0, *umin, *umax;
    double *d, *Hs, *M, *Hi, *W1, *W2, *Fs1, *Ku1, *Cq, *FsT, *KuT, *Mfrag, *Kbfrag, *Kufrag, *lbs, *ubs, *rho_dev;
    int* flag;
    int n, nx, nu, NT, S, SX, per_stage;
    double rho_in;
};

// in-place lower Cholesky of the n x n row-major W (upper part ignored); flag[0] |= 1 on failure
__device__ void chol_inplace(double* W, int n, int* flag) {
    const int tid = threadIdx.x;
    for (int j = 0; j < n; ++j) {
        if (tid == 0) {
            const double dj = W[(size_t)j * n + j];
            if (!(dj > 0.0)) { flag[0] |= 1; W[(size_t)j * n + j] = 1.0; }
            else W[(size_t)j * n + j] = sqrt(dj);
        }
        __syncthreads();
        const double inv = 1.0 / W[(size_t)j * n + j];
        for (int i = j + 1 + tid; i < n; i += 256) W[(size_t)i * n + j] *= inv;
        __syncthreads();
        for (int i = j + 1 + (tid >> 6); i < n; i += 4) {       // a wave per row, lanes along the columns
            const double lij = W[(size_t)i * n + j];
            for (int k = j + 1 + (tid & 63); k <= i; k += 64) W[(size_t)i * n + k] -= lij * W[(size_t)k * n + j];
        }
        __syncthreads();
    }
}

// X (n x n, symmetric) = (L L')^-1 given lower L in W; T is scratch for L^-1
__device__ void inverse_from_chol(const double* W, double* T, double* X, int n) {
    const int tid = threadIdx.x;
    for (int i = 0; i < n; ++i) {           // row i of L^-1 from rows < i
        const double inv = 1.0 / W[(size_t)i * n + i];
        for (int c = tid; c <= i; c += 256) {
            double s = (c == i) ? 1.0 : 0.0;
            for (int k = c; k < i; ++k) s -= W[(size_t)i * n + k] * T[(size_t)k * n + c];
            T[(size_t)i * n + c] = s * inv;
        }
        __syncthreads();
    }
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx % n;
        if (j > i) continue;
        double s = 0.0;
        for (int k = i; k < n; ++k) s += T[(size_t)k * n + i] * T[(size_t)k * n + j];
        X[(size_t)i * n + j] = s;
        X[(size_t)j * n + i] = s;
    }
    __syncthreads();
}

// ---- shared model: the set-up as a chain of small grids (it was ONE 256-thread workgroup working in global memory behind
// ~1,000 barriers: 4.0 ms at n = 120).  The two inverses run blocked on the MFMA GEMM
// (blocked_setup.hip); everything else is element-wise or a dot product per output element.

// d, Hs = d H d (symmetrised so that rounding in the assembly cannot leave it unsymmetric), W1 = Hs
__global__ __launch_bounds__(256) void fac_scale_kernel(FacParams f) {
    const int n = f.n;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    const int i = idx / n, j = idx - i * n;
    const double hii = f.H[(size_t)i * n + i], hjj = f.H[(size_t)j * n + j];
    const double di = hii > 0.0 ? 1.0 / sqrt(hii) : 1.0, dj = hjj > 0.0 ? 1.0 / sqrt(hjj) : 1.0;
    if (i == j) { f.d[i] = di; if (!(hii > 0.0)) atomicOr(f.flag, 1); }
    const double v = 0.5 * (f.H[idx] + f.H[(size_t)j * n + i]) * di * dj;
    f.Hs[idx] = v;
    f.W1[idx] = v;
}

// spectrum bounds lmax <= |Hs|_inf, lmin >= 1 / |Hi|_inf  ->  rho;  rho_dev = (rho, lmax, lmin, 0)
__global__ __launch_bounds__(256) void fac_rho_kernel(FacParams f) {
    __shared__ double ra[4], rb[4];
    const int n = f.n, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    double a = 0.0, b = 0.0;
    for (int i = wave; i < n; i += 4) {         // a wave per row, lanes along the columns
        double sa = 0.0, sb = 0.0;
        for (int j = lane; j < n; j += 64) { sa += fabs(f.Hs[(size_t)i * n + j]); sb += fabs(f.Hi[(size_t)i * n + j]); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { sa += __shfl_xor(sa, o); sb += __shfl_xor(sb, o); }
        a = fmax(a, sa); b = fmax(b, sb);
    }
    if (lane == 0) { ra[wave] = a; rb[wave] = b; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double lmax = fmax(fmax(ra[0], ra[1]), fmax(ra[2], ra[3]));
        const double lmin = 1.0 / fmax(fmax(rb[0], rb[1]), fmax(rb[2], rb[3]));
        f.rho_dev[0] = f.rho_in > 0.0 ? f.rho_in : 0.1 * sqrt(lmax * lmin);
        f.rho_dev[1] = lmax; f.rho_dev[2] = lmin; f.rho_dev[3] = 0.0;
    }
}

__global__ __launch_bounds__(256) void fac_shift_kernel(FacParams f) {
    const int n = f.n;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * n) return;
    f.W1[idx] = f.Hs[idx] + ((idx / n == idx % n) ? f.rho_dev[0] : 0.0);
}

// affine maps from [x0; 1]: Fs1 = [d F | d f0], Ku1 = -Hi Fs1, Kb1 = -M Fs1 (into W2); one thread per element
__global__ __launch_bounds__(256) void fac_maps_kernel(FacParams f) {
    const int n = f.n, nx = f.nx, nx1 = nx + 1;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n * nx1) return;
    const int i = idx / nx1, c = idx - i * nx1;
    double su = 0.0, sb = 0.0;
    for (int k = 0; k < n; ++k) {
        const double fv = f.d[k] * (c < nx ? f.F[(size_t)k * nx + c] : f.f0[k]);
        su -= f.Hi[(size_t)i * n + k] * fv;
        sb -= f.M[(size_t)i * n + k] * fv;
    }
    f.Fs1[idx] = f.d[i] * (c < nx ? f.F[(size_t)i * nx + c] : f.f0[i]);
    f.Ku1[idx] = su;
    f.W2[idx] = sb;
}

// how far the explicit inverse is from exact: max |Hs Hi - I| -> rho_dev[3] (the active-set stage refines if this is large)
__global__ __launch_bounds__(256) void fac_resid_kernel(FacParams f) {
    __shared__ double red[256];
    const int n = f.n, tid = threadIdx.x;
    const int idx = blockIdx.x * 256 + tid;
    double e = 0.0;
    if (idx < n * n) {
        const int i = idx / n, j = idx - i * n;
        double sacc = (i == j) ? -1.0 : 0.0;
        for (int k = 0; k < n; ++k) sacc += f.Hs[(size_t)i * n + k] * f.Hi[(size_t)k * n + j];
        e = fabs(sacc);
        if (!(e == e)) e = 1e300;       // NaN (a failed factorisation): report it as a large residual
    }
    red[tid] = e; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
    // non-negative doubles order like their bit patterns
    if (tid == 0) atomicMax(reinterpret_cast<unsigned long long*>(f.rho_dev + 3), (unsigned long long)__double_as_longlong(red[0]));
}

// Cq = Fs1' Ku1 (the active-set kernels get qs'x_unc as a quadratic form in [x0; 1]), the transposed maps, the scaled
// bounds and the MFMA A-operand fragment images: frag (t, s), lane l  <-  X[16t + (l&15)][4s + (l>>4)]
__global__ __launch_bounds__(256) void fac_tail_kernel(FacParams f) {
    const int n = f.n, nx1 = f.nx + 1, NP = 16 * f.NT;
    const int gid = blockIdx.x * 256 + threadIdx.x, gsz = gridDim.x * 256;
    const double rho = f.rho_dev[0];
    const double* Kb1 = f.W2;
    for (int idx = gid; idx < nx1 * nx1; idx += gsz) {
        const int a = idx / nx1, b = idx % nx1;
        double sacc = 0.0;
        for (int i = 0; i < n; ++i) sacc += f.Fs1[(size_t)i * nx1 + a] * f.Ku1[(size_t)i * nx1 + b];
        f.Cq[idx] = sacc;
    }
    for (int idx = gid; idx < NP * nx1; idx += gsz) {
        const int c = idx / NP, i = idx % NP;
        f.FsT[idx] = i < n ? f.Fs1[(size_t)i * nx1 + c] : 0.0;
        f.KuT[idx] = i < n ? f.Ku1[(size_t)i * nx1 + c] : 0.0;
    }
    for (int i = gid; i < NP; i += gsz) {
        double lo = 0.0, hi = 0.0;
        if (i < n) {
            const int j = f.per_stage ? i : (i % f.nu);
            lo = f.umin[j] / f.d[i];
            hi = f.umax[j] / f.d[i];
        }
        f.lbs[i] = lo; f.ubs[i] = hi;
    }
    for (int idx = gid; idx < f.NT * f.S * 64; idx += gsz) {
        const int l = idx & 63, ts = idx >> 6, t = ts / f.S, s = ts % f.S;
        const int i = 16 * t + (l & 15), k = 4 * s + (l >> 4);
        f.Mfrag[idx] = (i < n && k < n) ? rho * f.M[(size_t)i * n + k] : 0.0;
    }
    for (int idx = gid; idx < f.NT * f.SX * 64; idx += gsz) {
        const int l = idx & 63, ts = idx >> 6, t = ts / f.SX, s = ts % f.SX;
        const int i = 16 * t + (l & 15), k = 4 * s + (l >> 4);
        const bool in = (i < n && k < nx1);
        f.Kbfrag[idx] = in ? Kb1[(size_t)i * nx1 + k] : 0.0;
        f.Kufrag[idx] = in ? f.Ku1[(size_t)i * nx1 + k] : 0.0;
    }
}

// ---- one model per instance (n_models > 1): no shared factor, so no ADMM GEMM form; each workgroup prepares what
// the active-set kernel needs for ITS model: d, Hs, Hi = Hs^-1, [Fs|f0s]^T, [Ku|ku]^T, scaled bounds.
struct FacBatchParams {
    const double *H, *F, *f0, *umin, *umax;
    double *d, *Hs, *Hi, *W1, *W2, *Fs1, *Ku1, *FsT, *KuT, *lbs, *ubs, *rho_dev;
    int *flag, *mflag;                  // any model failed / per model
    int n, nx, nu, NP, per_stage, use_lds, probe;
    long bnd_stride;
};

__global__ __launch_bounds__(256) void factor_batch_kernel(FacBatchParams f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ double red[256];
    __shared__ int lflag[1];
    const int n = f.n, nx = f.nx, nx1 = f.nx + 1, NP = f.NP, tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const size_t m = blockIdx.x;
    const double* H = f.H + m * n * n;
    const double* F = f.F + m * n * nx;
    const double* f0 = f.f0 + m * n;
    double* d = f.d + m * n;
    double* gHs = f.Hs + m * n * n;
    double* gHi = f.Hi + m * n * n;
    double* Fs1 = f.Fs1 + m * n * nx1;
    double* Ku1 = f.Ku1 + m * n * nx1;
    // LDS: Hs, L (in place), L^-1, Hi, d  -- or, when 4 n^2 doubles do not fit, the global scratch and outputs
    double* sm = reinterpret_cast<double*>(smem);
    double* Hs = f.use_lds ? sm : gHs;
    double* W1 = f.use_lds ? sm + (size_t)n * n : f.W1 + m * n * n;
    double* W2 = f.use_lds ? sm + (size_t)2 * n * n : f.W2 + m * n * n;
    double* Hi = f.use_lds ? sm + (size_t)3 * n * n : gHi;
    if (tid == 0) lflag[0] = 0;
    __syncthreads();
    for (int i = tid; i < n; i += 256) {
        const double hii = H[(size_t)i * n + i];
        if (!(hii > 0.0)) { lflag[0] |= 1; d[i] = 1.0; }
        else d[i] = 1.0 / sqrt(hii);
    }
    __syncthreads();
    for (int i = wave; i < n; i += 4)
        for (int j = lane; j < n; j += 64) {
            const double v = 0.5 * (H[(size_t)i * n + j] + H[(size_t)j * n + i]) * d[i] * d[j];
            Hs[i * n + j] = v;
            W1[i * n + j] = v;
        }
    __syncthreads();
    chol_inplace(W1, n, lflag);
    inverse_from_chol(W1, W2, Hi, n);
    for (int idx = tid; idx < n * nx1; idx += 256) {
        const int i = idx / nx1, c = idx - i * nx1;
        Fs1[idx] = d[i] * (c < nx ? F[(size_t)i * nx + c] : f0[i]);
    }
    __syncthreads();
    for (int idx = tid; idx < n * nx1; idx += 256) {
        const int i = idx / nx1, c = idx - i * nx1;
        double su = 0.0;
        for (int k = 0; k < n; ++k) su -= Hi[i * n + k] * Fs1[(size_t)k * nx1 + c];
        Ku1[idx] = su;
    }
    {   // max |Hs Hi - I| over all models (non-negative doubles order like their bit patterns)
        double e = 0.0;
        for (int i = wave; i < n; i += 4)
            for (int j = lane; j < n; j += 64) {
                double sacc = (i == j) ? -1.0 : 0.0;
                for (int k = 0; k < n; ++k) sacc += Hs[i * n + k] * Hi[k * n + j];
                e = fmax(e, fabs(sacc));
            }
        red[tid] = e; __syncthreads();
        for (int sft = 128; sft > 0; sft >>= 1) { if (tid < sft) red[tid] = fmax(red[tid], red[tid + sft]); __syncthreads(); }
        if (tid == 0) {
            atomicMax(reinterpret_cast<unsigned long long*>(f.rho_dev + 3), (unsigned long long)__double_as_longlong(red[0]));
            if (lflag[0]) atomicOr(f.flag, 1);
            f.mflag[m] = lflag[0];
        }
        __syncthreads();
    }
    if (f.use_lds)
        for (int idx = tid; idx < n * n; idx += 256) { gHs[idx] = Hs[idx]; gHi[idx] = Hi[idx]; }
    double* FsT = f.FsT + m * NP * nx1;
    double* KuT = f.KuT + m * NP * nx1;
    for (int c = wave; c < nx1; c += 4)
        for (int i = lane; i < NP; i += 64) {
            FsT[c * NP + i] = i < n ? Fs1[(size_t)i * nx1 + c] : 0.0;
            KuT[c * NP + i] = i < n ? Ku1[(size_t)i * nx1 + c] : 0.0;
        }
    const double* umin = f.umin + m * f.bnd_stride;
    const double* umax = f.umax + m * f.bnd_stride;
    for (int i = tid; i < NP; i += 256) {
        double lo = 0.0, hi = 0.0;
        if (i < n) {
            const int j = f.per_stage ? i : (i % f.nu);
            lo = umin[j] / d[i];
            hi = umax[j] / d[i];
        }
        f.lbs[m * NP + i] = lo; f.ubs[m * NP + i] = hi;
    }
}

// n <= 64: one WAVE per model, no workgroup barriers (the 256-thread kernel above spends its time in ~200 barriers per
// model).  Blocked right-looking Cholesky on 16 x 16 tiles: the diagonal tile is factored and inverted in REGISTERS
// (lane r owns row r, v_readlane broadcasts), every off-diagonal product -- panel L_ib = S_ib T_bb', trailing update
// S_ij -= L_ib L_jb', T = L^-1 by block back-substitution, Hi = T'T -- is a v_mfma_f64_16x16x4_f64 tile product with
// operands read once from LDS.  LDS per wave: two (16 nb) x (16 nb + 1) matrices.
__device__ __forceinline__ void wv_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// 1 / sqrt(x) from v_rsq_f64 and two Newton steps (the division-free form of the pivot scaling)
__device__ __forceinline__ double frsqrt(double x) {
    double y = __builtin_amdgcn_rsq(x);
    y = y * (1.5 - 0.5 * x * y * y);
    y = y * (1.5 - 0.5 * x * y * y);
    return y;
}

template <int SRC>
__device__ __forceinline__ double readlane_d(double v) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), SRC);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), SRC);
    return __hiloint2double(hi, lo);
}

// acc += (lane C of the row of 16 lanes)'s src * w in ONE instruction: the 64-bit FMA takes its first operand through the DPP
// row broadcast (gfx90a+), where v_readlane needs two reads into scalar registers and the product a third instruction.
// The compiler does not see inside the statement, so the DPP hazard (a VALU write of the broadcast register within the two
// instructions before) is kept off by hand: dpp_guard() after the value is computed, before its first broadcast.
template <int C>
__device__ __forceinline__ void fmac_bc(double& acc, double src, double w) {
    asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(C));
}
__device__ __forceinline__ void dpp_guard(double& v) { asm volatile("s_nop 1" : "+v"(v)); }

template <int J>
struct DiagStep {
    // column J of the tile's Cholesky factor and the rank-1 update of the columns right of it
    static __device__ __forceinline__ void run(double (&a)[16], double (&dinv)[16], int r, bool& bad) {
        const double piv = readlane_d<J>(a[J]);
        if (!(piv > 0.0)) bad = true;
        const double rs = piv > 0.0 ? frsqrt(piv) : 1.0;
        dinv[J] = rs;
        a[J] = r < J ? 0.0 : (r == J ? piv * rs : a[J] * rs);
        const double na = -a[J];
        dpp_guard(a[J]);
        update<J + 1>(a, na);
        DiagStep<J + 1>::run(a, dinv, r, bad);
    }
    template <int C>
    static __device__ __forceinline__ void update(double (&a)[16], double na) {
        if constexpr (C < 16) {
            fmac_bc<C>(a[C], a[J], na);                  // a_rC -= l_rJ l_CJ (meaningful for r >= C; zeroed at step C otherwise)
            update<C + 1>(a, na);
        }
    }
};
template <>
struct DiagStep<16> {
    static __device__ __forceinline__ void run(double (&)[16], double (&)[16], int, bool&) {}
};

template <int RR>
struct InvStep {
    // entry RR of column r of T = L^-1: t_RR = (delta - sum_{k<RR} l_RR,k t_k) / l_RR,RR, row RR of L broadcast from lane RR
    // (nt = -t is what the products take)
    static __device__ __forceinline__ void run(const double (&a)[16], const double (&dinv)[16], double (&t)[16], double (&nt)[16], int r) {
        double acc = (RR == r) ? 1.0 : 0.0;
        accum<0>(a, nt, acc);
        t[RR] = RR >= r ? acc * dinv[RR] : 0.0;
        nt[RR] = -t[RR];
        InvStep<RR + 1>::run(a, dinv, t, nt, r);
    }
    template <int K>
    static __device__ __forceinline__ void accum(const double (&a)[16], const double (&nt)[16], double& acc) {
        if constexpr (K < RR) {
            fmac_bc<RR>(acc, a[K], nt[K]);
            accum<K + 1>(a, nt, acc);
        }
    }
};
template <>
struct InvStep<16> {
    static __device__ __forceinline__ void run(const double (&)[16], const double (&)[16], double (&)[16], double (&)[16], int) {}
};

// ONE n x (16 NB + 1) matrix in LDS holds everything: Hs -> (lower off-diagonal tiles) L_ij, (diagonal tiles) T_bb =
// L_bb^-1, (upper off-diagonal tile (b, i)) T_ib' ; the tiles of Hi = T'T are accumulated in registers and overwrite it.
// The identity padding of the last row block (rows n .. 16 NB - 1) is not stored: every step leaves it what it was (L, T and
// Hi of the padding are the identity, its off-diagonal tiles zero), so reads of those rows are answered with the constant
// (PADL) and writes to them dropped -- 15.7 instead of 18.8 KB per wave at n = 40, nine waves per CU instead of seven.
#ifndef FACW
#define FACW 2
#endif
template <int NB>
__global__ __launch_bounds__(64, FACW) void factor_wave_kernel(FacBatchParams f) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int n = f.n, nx = f.nx, nx1 = f.nx + 1, NP = f.NP, i = threadIdx.x;
    constexpr int NPD = 16 * NB, LDW = NPD + 1;
    const int col = i & 15, kq = i >> 4;
    const size_t m = blockIdx.x;
    const double* __restrict__ H = f.H + m * n * n;
    const double* __restrict__ F = f.F + m * n * nx;
    const double* __restrict__ f0 = f.f0 + m * n;
    double* __restrict__ gHs = f.Hs + m * n * n;
    double* __restrict__ gHi = f.Hi + m * n * n;
    double* S = reinterpret_cast<double*>(smem);
    double* sHi = S;
    double* sd = S + (size_t)n * LDW;                     // d (n)
    // rows of the LAST row block may be padding (LASTB: a compile-time fact after unrolling; the test on r is then the only cost)
    auto PADL = [&](bool lastb, int r, int c) -> double { return (!lastb || r < n) ? S[r * LDW + c] : (r == c ? 1.0 : 0.0); };
    auto PADS = [&](bool lastb, int r, int c, double v) { if (!lastb || r < n) S[r * LDW + c] = v; };
    double* sP = sd + n;                                  // two probe vectors (synchronous calls only: the launch sizes the LDS)
    const bool in = i < n;
    bool bad = false;
    double di = 1.0;
#ifdef FAC_PROF
    long long tk0 = wall_clock64(), t_ld = 0, t_chol = 0, t_inv = 0, t_hi = 0, t_ku = 0, t_pr = 0;
#endif
    // Everything the wave needs from memory is asked for before anything waits: the diagonal entry of the lane's row (the
    // Jacobi scale, without a trip through LDS), [F | f0] in the layout the Ku product at the end wants its B operand in
    // (lane (c, kq) holds rows 16 kb + 4 s + kq of column c: registers, no LDS), and H as one flat array, 32 coalesced loads
    // in flight per lane (row by row the 40 loads of a bicycle model came back in 5 rounds of memory latency).
    const double hii = in ? H[(size_t)i * n + i] : 1.0;
    double fb[NB][4];
    auto loadF = [&](int c0) {
        const int c = c0 + col;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = 16 * kb + 4 * s + kq;
                fb[kb][s] = (kk < n && c < nx1) ? (c < nx ? F[(size_t)kk * nx + c] : f0[kk]) : 0.0;
            }
    };
    loadF(0);
    {
        const int tot = n * n;
        for (int base = 0; base < tot; base += 32 * 64) {
            double hv[32];
#pragma unroll
            for (int u = 0; u < 32; ++u) { const int idx = base + 64 * u + i; hv[u] = idx < tot ? H[idx] : 0.0; }
#pragma unroll
            for (int u = 0; u < 32; ++u) {
                const int idx = base + 64 * u + i;
                if (idx < tot) { const int r = idx / n; S[r * LDW + (idx - r * n)] = hv[u]; }
            }
        }
    }
    for (int idx = i; idx < n * (NPD - n); idx += 64) { const int r = idx / (NPD - n); S[r * LDW + n + (idx - r * (NPD - n))] = 0.0; }
    if (in) {
        if (!(hii > 0.0)) bad = true; else di = 1.0 / sqrt(hii);
        sd[i] = di;
    }
    wv_sync();
    double* __restrict__ Fs1 = f.Fs1 + m * n * nx1;
    double* __restrict__ FsT = f.FsT + m * NP * nx1;
    // [Fs | f0s] = d [F | f0], also transposed (nx+1, NP): straight from the registers (first 16 columns; the rest with Ku below)
    auto scaleF = [&](int c0) {
        const int c = c0 + col;
#pragma unroll
        for (int kb = 0; kb < NB; ++kb)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int kk = 16 * kb + 4 * s + kq;
                if (kk < n && c < nx1) {
                    fb[kb][s] *= sd[kk];
                    Fs1[(size_t)kk * nx1 + c] = fb[kb][s]; FsT[c * NP + kk] = fb[kb][s];
                }
            }
    };
    scaleF(0);
    // Hs = d (H + H')/2 d in two sweeps whose iterations do not depend on each other (eight rows' reads in flight; the
    // row-by-row form with its mirrored store was a chain of n LDS round trips): the upper triangle from the raw entries --
    // nothing below the diagonal is written in this sweep -- then the mirror image and the copy to global memory
    for (int r0 = 0; r0 < n; r0 += 8) {
        double av[8], bv[8], dv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            const bool on = in && r < n && i >= r;
            av[u] = on ? S[r * LDW + i] : 0.0; bv[u] = on ? S[i * LDW + r] : 0.0; dv[u] = on ? sd[r] : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            if (in && r < n && i >= r) S[r * LDW + i] = 0.5 * (av[u] + bv[u]) * dv[u] * di;
        }
    }
    wv_sync();
    for (int r0 = 0; r0 < n; r0 += 8) {
        double v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            v[u] = (in && r < n) ? (i >= r ? S[r * LDW + i] : S[i * LDW + r]) : 0.0;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int r = r0 + u;
            if (in && r < n) { if (i < r) S[r * LDW + i] = v[u]; gHs[(size_t)r * n + i] = v[u]; }
        }
    }
#ifdef FAC_PROF
    { long long tk = wall_clock64(); t_ld += tk - tk0; tk0 = tk; }
#endif
    // ---- blocked Cholesky, L^-1 of the diagonal tiles on the way -----------------------------------------------
#pragma unroll
    for (int b = 0; b < NB; ++b) {
        const int o = 16 * b;
        double a[16], t[16], nt[16], dinv[16];
#pragma unroll
        for (int j = 0; j < 16; ++j) a[j] = PADL(b == NB - 1, o + col, o + j);
        DiagStep<0>::run(a, dinv, col, bad);
#pragma unroll
        for (int j = 0; j < 16; ++j) dpp_guard(a[j]);
        InvStep<0>::run(a, dinv, t, nt, col);
        wv_sync();
        if (i < 16) {
#pragma unroll
            for (int j = 0; j < 16; ++j) PADS(b == NB - 1, o + j, o + col, t[j]);     // column col of T_bb replaces the tile
        }
        wv_sync();
#pragma unroll
        for (int ib = b + 1; ib < NB; ++ib) {                   // panel: L_ib = S_ib T_bb'
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < 4; ++s)
                acc = mfma_f64(PADL(ib == NB - 1, 16 * ib + col, o + 4 * s + kq), S[(o + col) * LDW + o + 4 * s + kq], acc);
            wv_sync();
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) PADS(ib == NB - 1, 16 * ib + kq + 4 * r4, o + col, acc[r4]);
        }
        wv_sync();
#pragma unroll
        for (int ib = b + 1; ib < NB; ++ib)                     // trailing update: S_ij -= L_ib L_jb'
#pragma unroll
            for (int jb = b + 1; jb <= ib; ++jb) {
                d4 acc;
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) acc[r4] = PADL(ib == NB - 1, 16 * ib + kq + 4 * r4, 16 * jb + col);
#pragma unroll
                for (int s = 0; s < 4; ++s)
                    acc = mfma_f64(-PADL(ib == NB - 1, 16 * ib + col, o + 4 * s + kq), PADL(jb == NB - 1, 16 * jb + col, o + 4 * s + kq), acc);
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) PADS(ib == NB - 1, 16 * ib + kq + 4 * r4, 16 * jb + col, acc[r4]);
            }
        wv_sync();
    }
#ifdef FAC_PROF
    { long long tk = wall_clock64(); t_chol += tk - tk0; tk0 = tk; }
#endif
    // ---- T = L^-1 below the diagonal tiles: T_ib = -T_ii (sum_{b <= j < i} L_ij T_jb), stored transposed in tile (b, i) ----
    // T_jb[k][c]: j == b -> the diagonal tile S[(16b + k)][16b + c];  j > b -> S[(16b + c)][16j + k]
#pragma unroll
    for (int b = 0; b < NB; ++b)
#pragma unroll
        for (int ib = b + 1; ib < NB; ++ib) {
            d4 w = {0, 0, 0, 0};
#pragma unroll
            for (int jb = b; jb < ib; ++jb)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kk = 4 * s + kq;
                    const double tv = jb == b ? S[(16 * b + kk) * LDW + 16 * b + col] : S[(16 * b + col) * LDW + 16 * jb + kk];
                    w = mfma_f64(PADL(ib == NB - 1, 16 * ib + col, 16 * jb + kk), tv, w);
                }
            d4 o4 = {0, 0, 0, 0};
#pragma unroll
            for (int r = 0; r < 4; ++r)         // the D registers of w are the B operand of k-step r with k = kq + 4 r
                o4 = mfma_f64(-PADL(ib == NB - 1, 16 * ib + col, 16 * ib + kq + 4 * r), w[r], o4);
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) S[(16 * b + col) * LDW + 16 * ib + kq + 4 * r4] = o4[r4];
            wv_sync();
        }
#ifdef FAC_PROF
    { long long tk = wall_clock64(); t_inv += tk - tk0; tk0 = tk; }
#endif
    // ---- Hi = T'T: every lower tile in registers first, then over the matrix (mirrored) ------------------------------------
    // T_kc[kk][x] for k >= c: k == c -> S[(16c + kk)][16c + x];  k > c -> S[(16c + x)][16k + kk]
    {
        d4 hacc[NB * (NB + 1) / 2];
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb) {
                d4 acc = {0, 0, 0, 0};
#pragma unroll
                for (int kb = ib; kb < NB; ++kb)
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int kk = 4 * s + kq;
                        const double av = kb == ib ? PADL(ib == NB - 1, 16 * ib + kk, 16 * ib + col) : S[(16 * ib + col) * LDW + 16 * kb + kk];
                        const double bv = kb == jb ? PADL(jb == NB - 1, 16 * jb + kk, 16 * jb + col) : S[(16 * jb + col) * LDW + 16 * kb + kk];
                        acc = mfma_f64(av, bv, acc);
                    }
                hacc[ib * (ib + 1) / 2 + jb] = acc;
            }
        wv_sync();
#pragma unroll
        for (int ib = 0; ib < NB; ++ib)
#pragma unroll
            for (int jb = 0; jb <= ib; ++jb)
#pragma unroll
                for (int r4 = 0; r4 < 4; ++r4) {
                    const double v = hacc[ib * (ib + 1) / 2 + jb][r4];
                    PADS(ib == NB - 1, 16 * ib + kq + 4 * r4, 16 * jb + col, v);
                    PADS(jb == NB - 1, 16 * jb + col, 16 * ib + kq + 4 * r4, v);
                }
    }
    wv_sync();
    for (int r = 0; r < n; ++r) if (in) gHi[(size_t)r * n + i] = sHi[r * LDW + i];
#ifdef FAC_PROF
    { long long tk = wall_clock64(); t_hi += tk - tk0; tk0 = tk; }
#endif
    // [Ku | ku], also transposed (nx+1, NP)
    double* __restrict__ Ku1 = f.Ku1 + m * n * nx1;
    double* __restrict__ KuT = f.KuT + m * NP * nx1;
    // Ku = -Hi [Fs | f0s] as MFMA tiles, 16 columns of the map at a time: A = a row block of Hi, B' = [Fs | f0s]' (zero
    // beyond its nx + 1 columns); the padding rows of the last block answer with the identity and are not stored
    for (int c0 = 0; c0 < nx1; c0 += 16) {
        const int c = c0 + col;
        if (c0 > 0) { loadF(c0); scaleF(c0); }
#pragma unroll
        for (int ib = 0; ib < NB; ++ib) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int kb = 0; kb < NB; ++kb)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int kk = 16 * kb + 4 * s + kq;
                    acc = mfma_f64(PADL(ib == NB - 1, 16 * ib + col, kk), fb[kb][s], acc);
                }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = 16 * ib + kq + 4 * r4;
                if (r < n && c < nx1) { Ku1[(size_t)r * nx1 + c] = -acc[r4]; KuT[c * NP + r] = -acc[r4]; }
            }
        }
    }
    for (int c = 0; c < nx1; ++c)
        for (int r = n + i; r < NP; r += 64) { FsT[c * NP + r] = 0.0; KuT[c * NP + r] = 0.0; }
#ifdef FAC_PROF
    { long long tk = wall_clock64(); t_ku += tk - tk0; tk0 = tk; }
#endif
    // accuracy of the explicit inverse, estimated with two probe vectors instead of the full n^2 product (this
    // kernel is instruction bound): max_i |(Hs Hi v - v)_i| for v = ones and v = (+1,-1,...); it only decides
    // whether the solve refines with Hs itself
    double e = 0.0;
    if (f.probe) {
        double v1 = 0.0, v2 = 0.0;
        if (in) {
#pragma unroll 4
            for (int j = 0; j < n; ++j) { const double hv = sHi[i * LDW + j]; v1 += hv; v2 += (j & 1) ? -hv : hv; }
        }
        wv_sync();
        double* p1 = sP;
        double* p2 = sP + n;
        if (in) { p1[i] = v1; p2[i] = v2; }
        wv_sync();
        if (in) {
            double w1 = 0.0, w2 = 0.0;
#pragma unroll 4
            for (int k = 0; k < n; ++k) { const double hs = gHs[(size_t)k * n + i]; w1 += hs * p1[k]; w2 += hs * p2[k]; }
            e = fmax(fabs(w1 - 1.0), fabs(w2 - ((i & 1) ? -1.0 : 1.0)));
        }
    }
    for (int off = 32; off > 0; off >>= 1) e = fmax(e, __shfl_xor(e, off, 64));
    const bool anybad = __any(bad);
    if (i == 0) {
        if (f.probe) atomicMax(reinterpret_cast<unsigned long long*>(f.rho_dev + 3), (unsigned long long)__double_as_longlong(e));
        if (anybad && f.probe) atomicOr(f.flag, 1);     // asynchronous calls report per model only (mflag): nobody reads the sum
        f.mflag[m] = anybad ? 1 : 0;
    }
    const double* umin = f.umin + m * f.bnd_stride;
    const double* umax = f.umax + m * f.bnd_stride;
    for (int r = i; r < NP; r += 64) {
        double lo = 0.0, hi = 0.0;
        if (r < n) {
            const int j = f.per_stage ? r : (r % f.nu);
            lo = umin[j] / sd[r];
            hi = umax[j] / sd[r];
        }
        f.lbs[m * NP + r] = lo; f.ubs[m * NP + r] = hi;
    }
    if (in) f.d[m * n + i] = di;
#ifdef FAC_PROF
    { long long tk = wall_clock64(); t_pr += tk - tk0; }
    if (i == 0 && (m % 2048) == 0)
        printf("model %d: load %lld chol %lld inv %lld Hi %lld Fs/Ku %lld probe/bounds %lld (x10 ns)\n", (int)m, t_ld, t_chol, t_inv, t_hi, t_ku, t_pr);
#endif
}

}  // namespace

int launch_factor(mpcqp_handle h, double rho, hipStream_t st) {
    FacParams f;
    f.H = h->H; f.F = h->F; f.f0 = h->f0; f.umin = h->umin; f.umax = h->umax;
    f.d = h->d; f.Hs = h->Hs; f.M = h->M; f.Hi = h->Hi; f.W1 = h->W1; f.W2 = h->W2;
    f.Fs1 = h->Fs1; f.Ku1 = h->Ku1; f.Cq = h->Cq; f.FsT = h->FsT; f.KuT = h->KuT; f.Mfrag = h->Mfrag; f.Kbfrag = h->Kbfrag; f.Kufrag = h->Kufrag;
    f.lbs = h->lbs; f.ubs = h->ubs; f.rho_dev = h->rho_dev; f.flag = h->flag_dev;
    f.n = h->n; f.nx = h->dims.nx; f.nu = h->dims.nu; f.NT = h->NT; f.S = h->S; f.SX = h->SX;
    f.per_stage = h->per_stage; f.rho_in = rho;
    const int n = h->n, nn = n * n, nx1 = h->dims.nx + 1;
    auto fail = [&](const char* what, hipError_t e) {
        h->err = std::string("factor: ") + what + ": " + hipGetErrorString(e); return MPCQP_EHIP;
    };
    hipError_t e = hipMemsetAsync(h->flag_dev, 0, sizeof(int), st);
    if (e != hipSuccess) return fail("memset", e);
    hipLaunchKernelGGL(fac_scale_kernel, dim3(ceil_div(nn, 256)), dim3(256), 0, st, f);
    int rc = blocked_spd_inverse(h->W1, n, h->facL, h->W2, h->facBlk, h->Hi, h->flag_dev, 1, st);       // Hi = Hs^-1
    if (rc != MPCQP_OK) { h->err = "factor: blocked inverse of Hs failed to launch"; return rc; }
    hipLaunchKernelGGL(fac_rho_kernel, dim3(1), dim3(256), 0, st, f);
    hipLaunchKernelGGL(fac_shift_kernel, dim3(ceil_div(nn, 256)), dim3(256), 0, st, f);
    rc = blocked_spd_inverse(h->W1, n, h->facL, h->W2, h->facBlk, h->M, h->flag_dev, 1, st);            // M = (Hs + rho I)^-1
    if (rc != MPCQP_OK) { h->err = "factor: blocked inverse of Hs + rho I failed to launch"; return rc; }
    hipLaunchKernelGGL(fac_maps_kernel, dim3(ceil_div(n * nx1, 256)), dim3(256), 0, st, f);
    hipLaunchKernelGGL(fac_resid_kernel, dim3(ceil_div(nn, 256)), dim3(256), 0, st, f);
    hipLaunchKernelGGL(fac_tail_kernel, dim3(std::max(1, std::min(64, ceil_div(h->NT * h->S * 64, 256)))), dim3(256), 0, st, f);
    e = hipGetLastError();
    if (e != hipSuccess) return fail("launch", e);
    return MPCQP_OK;
}

// probe: estimate the accuracy of the explicit inverses (read back by a synchronous mpcqp_factor; a no_sync call has no use
// for it -- the solve keeps the decision of the last synchronous one -- and it costs a sixth of the kernel)
int launch_factor_batch(mpcqp_handle h, hipStream_t st, bool probe) {
    FacBatchParams f;
    f.probe = probe ? 1 : 0;
    f.H = h->H; f.F = h->F; f.f0 = h->f0; f.umin = h->umin; f.umax = h->umax;
    f.d = h->d; f.Hs = h->Hs; f.Hi = h->Hi; f.W1 = h->W1; f.W2 = h->W2; f.Fs1 = h->Fs1; f.Ku1 = h->Ku1;
    f.FsT = h->FsT; f.KuT = h->KuT; f.lbs = h->lbs; f.ubs = h->ubs; f.rho_dev = h->rho_dev; f.flag = h->flag_dev;
    f.n = h->n; f.nx = h->dims.nx; f.nu = h->dims.nu; f.NP = h->NP; f.per_stage = h->per_stage;
    f.bnd_stride = h->bnd_stride;
    const size_t lds = (size_t)4 * h->n * h->n * sizeof(double);
    f.use_lds = lds <= 128 * 1024;
    f.mflag = h->mflag;
    hipError_t e = probe ? hipMemsetAsync(h->rho_dev, 0, 4 * sizeof(double), st) : hipSuccess;      // only the probe writes it
    if (e == hipSuccess && (probe || h->n > 64)) e = hipMemsetAsync(h->flag_dev, 0, sizeof(int), st);   // read back by synchronous calls only
    if (e == hipSuccess && f.use_lds)
        e = ensure_dyn_lds(h->dev, reinterpret_cast<const void*>(&factor_batch_kernel), lds);
    if (e != hipSuccess) { h->err = std::string("factor (batch): ") + hipGetErrorString(e); return MPCQP_EHIP; }
    if (h->n <= 64) {       // one wave per model
        const int nb = ceil_div(h->n, 16);
        const size_t npd = 16 * (size_t)nb;
        const size_t wl = ((size_t)h->n * (npd + 1) + h->n + (probe ? 2 * (size_t)h->n : 0)) * sizeof(double);
        auto kern = nb == 1 ? &factor_wave_kernel<1> : nb == 2 ? &factor_wave_kernel<2> : nb == 3 ? &factor_wave_kernel<3> : &factor_wave_kernel<4>;
        e = ensure_dyn_lds(h->dev, reinterpret_cast<const void*>(kern), wl);
        if (e != hipSuccess) { h->err = std::string("factor (batch): ") + hipGetErrorString(e); return MPCQP_EHIP; }
        hipLaunchKernelGGL(kern, dim3(h->dims.n_models), dim3(64), wl, st, f);
    } else
    hipLaunchKernelGGL(factor_batch_kernel, dim3(h->dims.n_models), dim3(256), f.use_lds ? lds : 0, st, f);
    e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("factor (batch) launch: ") + hipGetErrorString(e); return MPCQP_EHIP; }
    return MPCQP_OK;
}
