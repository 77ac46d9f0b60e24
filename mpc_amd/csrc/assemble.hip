// Condensed-QP assembly: horizon rollout of (A_k, B_k) into the prediction matrices Gamma / Phi,
// then H = Gamma' Qbar Gamma + Rbar, F = Gamma' Qbar Phi and the constant gradient part f0.
//
// Reference counterparts (NumPy, one problem at a time):
//   rollout        LaunchPlannerQPFormulation.py:49-89 (vi/xi stencil + prefix sums),
//                  Launch3DPlanner.py:61-108
//   H, gradient    HelperFuncs.py:6-11 getQuadProgCostFunction (Q = J'WJ, f = -J'Wc) with
//                  J = [Gamma; I], W = blkdiag(Qbar, Rbar); LaunchPlannerQPFormulation.py:151-162
//
// One 256-thread workgroup per model when there are many models; ONE shared model is spread over the GPU instead: the
// rollout's chain is independent per column of Gamma, so gridDim.y workgroups walk the horizon on their own column tiles,
// and a second launch spreads the tiles of H and F (it starts when all of Gamma is there).  Phase 1 walks the horizon: row block k of Gamma is
// A_k times row block k-1 with B_k dropped on the block diagonal (the state-space form of the
// reference's prefix sums); Qbar Gamma is formed on the fly.  Phase 2 is the SYRK-shaped
// contraction Gamma'(Qbar Gamma) on v_mfma_f64_16x16x4_f64, lower tiles only and mirrored, with
// the all-zero leading k-steps of the block-lower-triangular Gamma skipped.
#include "common.h"
#include <algorithm>

namespace {

struct AsmParams {
    const double *A, *B, *Q, *R, *P, *xref, *uref;
    int rows_in_lds, ab_in_lds, q_in_lds;
    long xref_stride, uref_stride;       // doubles between the references of consecutive models (0: shared)
    double *Gam, *QG, *Phi, *H, *F, *f0;
    int nx, nu, N, n, p, tv;
    int phases;     // bit 0: the rollout; bit 1: H, F, f0.  One model: two launches, each spread over gridDim.y workgroups
};

// barrier that orders LDS traffic only: the rollout's global stores (Gamma, Qbar Gamma, Phi rows) stay in flight across
// it -- nothing reads them back before the full barrier that ends the rollout
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__global__ __launch_bounds__(256) void assemble_kernel(AsmParams a) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int nx = a.nx, nu = a.nu, N = a.N, n = a.n, p = a.p;
    const int m = blockIdx.x;                       // model index
    const size_t strideA = (size_t)(a.tv ? N : 1) * nx * nx, strideB = (size_t)(a.tv ? N : 1) * nx * nu;
    const double* Am = a.A + m * strideA;
    const double* Bm = a.B + m * strideB;
    double* Gam = a.Gam + (size_t)m * p * n;
    double* QG = a.QG + (size_t)m * p * n;
    double* Phi = a.Phi + (size_t)m * p * nx;
    double* H = a.H + (size_t)m * n * n;
    double* F = a.F + (size_t)m * n * nx;
    double* f0 = a.f0 + (size_t)m * n;

    // row-block buffers of the rollout: LDS ping-pong, or (long horizons, a.rows_in_lds == 0) the rows of Gamma
    // themselves in global memory -- block k-1 is complete and visible to the workgroup when block k reads it
    double* rowP = reinterpret_cast<double*>(smem);     // 2 x (nx, nx)
    double* rowG = rowP + 2 * nx * nx;                  // 2 x (nx, n) ping-pong
    double* sQ = rowG + (a.rows_in_lds ? 2 * nx * n : 0);   // Q, P (nx, nx) each
    double* sA = sQ + (a.q_in_lds ? 2 * nx * nx : 0);   // the model's A_k, B_k (a.ab_in_lds): the rollout is a chain of
    double* sB = sA + (a.tv ? N : 1) * nx * nx;         // N dependent steps, so nothing in it should wait on global memory
    const int tid = threadIdx.x;
    const int Y = gridDim.y, y = blockIdx.y;
    if (a.phases & 1) {
    if (a.q_in_lds) for (int i = tid; i < nx * nx; i += 256) { sQ[i] = a.Q[i]; sQ[nx * nx + i] = a.P[i]; }
    if (a.ab_in_lds) {
        for (int i = tid; i < (int)strideA; i += 256) sA[i] = Am[i];
        for (int i = tid; i < (int)strideB; i += 256) sB[i] = Bm[i];
    }

    // ---- phase 1: rollout --------------------------------------------------------------
    if (a.rows_in_lds) for (int i = tid; i < nx * n; i += 256) rowG[i] = 0.0;
    for (int i = tid; i < nx * nx; i += 256) rowP[i] = (i / nx == i % nx) ? 1.0 : 0.0;
    __syncthreads();
    // every product of the chain (A_k x previous row block, A_k x Phi block, Q x new row block) is a set of 16 x 16 MFMA
    // tiles with K = nx, spread over the four waves: ~5x fewer instructions than one thread per element, which is what
    // bounds this kernel when there is one small model per instance
    const int lane_ = tid & 63, wave_ = tid >> 6, col_ = lane_ & 15, kq_ = lane_ >> 4;
    const int RTn = (nx + 15) / 16, CTn = (n + 15) / 16, KS = (nx + 3) / 4;
    // this workgroup's column tiles of Gamma: y, y + Y, ...; the Phi chain belongs to workgroup 0
    const int nct = y < CTn ? (CTn - y + Y - 1) / Y : 0;
    const int ntG = RTn * nct, ntP = y == 0 ? RTn * RTn : 0;
    for (int k = 0; k < N; ++k) {
        const double* Ak = (a.ab_in_lds ? sA : Am) + (a.tv ? (size_t)k * nx * nx : 0);
        const double* Bk = (a.ab_in_lds ? sB : Bm) + (a.tv ? (size_t)k * nx * nu : 0);
        // (more than 71 states: the weights stay in memory -- the Phi ping-pong alone then takes most of the LDS)
        const double* Qk = a.q_in_lds ? ((k == N - 1) ? sQ + nx * nx : sQ) : ((k == N - 1) ? a.P : a.Q);
        const double* gin = a.rows_in_lds ? rowG + (k & 1) * nx * n : Gam + (size_t)(k > 0 ? k - 1 : 0) * nx * n;
        double* gout = a.rows_in_lds ? rowG + ((k + 1) & 1) * nx * n : Gam + (size_t)k * nx * n;
        const double* pin = rowP + (k & 1) * nx * nx;
        double* pout = rowP + ((k + 1) & 1) * nx * nx;
        const int cB = k * nu;                          // first column of stage k
        // tiles 0 .. RTn*CTn-1: Gamma row block;  the next RTn*RTn: Phi row block
        for (int t = wave_; t < ntG + ntP; t += 4) {
            const bool isP = t >= ntG;
            const int tt = isP ? t - ntG : t;
            const int wdt = isP ? nx : n;
            const int rt = isP ? tt / RTn : tt / nct, ct = isP ? tt % RTn : y + Y * (tt % nct);
            const double* src = isP ? pin : gin;
            const int ra = 16 * rt + col_, cb = 16 * ct + col_;
            d4 acc = {0, 0, 0, 0};
            if (k > 0 || isP)
                for (int s = 0; s < KS; ++s) {
                    const int kk = 4 * s + kq_;
                    const double av = (ra < nx && kk < nx) ? Ak[ra * nx + kk] : 0.0;
                    const double bv = (kk < nx && cb < wdt) ? src[kk * wdt + cb] : 0.0;
                    acc = mfma_f64(av, bv, acc);
                }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = 16 * rt + kq_ + 4 * r4;
                if (r < nx && cb < wdt) {
                    double v = acc[r4];
                    if (isP) {
                        pout[r * nx + cb] = v;
                        Phi[(size_t)(k * nx + r) * nx + cb] = v;
                    } else {
                        if (cb >= cB) v = cb < cB + nu ? Bk[r * nu + (cb - cB)] : 0.0;
                        gout[r * n + cb] = v;
                        if (a.rows_in_lds) Gam[(size_t)(k * nx + r) * n + cb] = v;
                    }
                }
            }
        }
        if (a.rows_in_lds) lds_barrier(); else { __threadfence_block(); __syncthreads(); }
        for (int t = wave_; t < ntG; t += 4) {                  // Qbar Gamma row block
            const int rt = t / nct, ct = y + Y * (t % nct);
            const int ra = 16 * rt + col_, cb = 16 * ct + col_;
            d4 acc = {0, 0, 0, 0};
            for (int s = 0; s < KS; ++s) {
                const int kk = 4 * s + kq_;
                const double av = (ra < nx && kk < nx) ? Qk[ra * nx + kk] : 0.0;
                const double bv = (kk < nx && cb < n) ? gout[kk * n + cb] : 0.0;
                acc = mfma_f64(av, bv, acc);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int r = 16 * rt + kq_ + 4 * r4;
                if (r < nx && cb < n) QG[(size_t)(k * nx + r) * n + cb] = acc[r4];
            }
        }
        if (a.rows_in_lds) lds_barrier(); else { __threadfence_block(); __syncthreads(); }
    }
    }
    if (!(a.phases & 2)) return;
    __threadfence_block();
    __syncthreads();

    // ---- phase 2: H = Gam' QG + Rbar on fp64 MFMA, lower tiles, mirrored ----------------------
    const int lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int NT = (n + 15) / 16;
    const int ksteps = (p + 3) / 4;
    int tile = 0;
    for (int ti = 0; ti < NT; ++ti)
        for (int tj = 0; tj <= ti; ++tj, ++tile) {
            if (tile % (4 * Y) != wave + 4 * y) continue;
            // rows of Gamma above stage(first column of the later tile) * nx are zero in that tile
            const int first_stage = (16 * ti) / nu;
            const int s0 = (first_stage * nx) / 4;
            d4 acc = {0, 0, 0, 0};
            const int ci = 16 * ti + col, cj = 16 * tj + col;
            for (int s = s0; s < ksteps; ++s) {
                const int r = 4 * s + kq;
                double av = 0.0, bv = 0.0;
                if (r < p) {
                    if (ci < n) av = Gam[(size_t)r * n + ci];
                    if (cj < n) bv = QG[(size_t)r * n + cj];
                }
                acc = mfma_f64(av, bv, acc);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int i = 16 * ti + kq + 4 * r4, j = 16 * tj + col;
                if (i < n && j < n && j <= i) {
                    double v = acc[r4];
                    if (i / nu == j / nu) v += a.R[(i % nu) * nu + (j % nu)];
                    H[(size_t)i * n + j] = v;
                    H[(size_t)j * n + i] = v;
                }
            }
        }

    // ---- phase 3: F = QG' Phi (n, nx) and f0 = -QG' xr - Rbar ur --------------------------------
    {
        const int NXT = (nx + 15) / 16;
        for (int t = wave + 4 * y; t < NT * NXT; t += 4 * Y) {
            const int ti = t / NXT, tj = t % NXT;
            d4 acc = {0, 0, 0, 0};
            const int ci = 16 * ti + col, cj = 16 * tj + col;
            for (int s = 0; s < ksteps; ++s) {
                const int r = 4 * s + kq;
                double av = 0.0, bv = 0.0;
                if (r < p) {
                    if (ci < n) av = QG[(size_t)r * n + ci];
                    if (cj < nx) bv = Phi[(size_t)r * nx + cj];
                }
                acc = mfma_f64(av, bv, acc);
            }
#pragma unroll
            for (int r4 = 0; r4 < 4; ++r4) {
                const int i = 16 * ti + kq + 4 * r4, j = 16 * tj + col;
                if (i < n && j < nx) F[(size_t)i * nx + j] = acc[r4];
            }
        }
        // f0: wave w sums its quarter of the rows, the quarters meet in LDS (the row buffers of the rollout are free now)
        __syncthreads();
        double* part = reinterpret_cast<double*>(smem);         // 4 x 64
        const int pq = (p + 3) / 4, r0 = wave * pq, r1 = min(p, r0 + pq);
        for (int i0 = 64 * y; i0 < n; i0 += 64 * Y) {
            const int i = i0 + lane;
            double s = 0.0;
            if (i < n && a.xref) {
                const double* xr = a.xref + (size_t)m * a.xref_stride;
#pragma unroll 4
                for (int r = r0; r < r1; ++r) s -= QG[(size_t)r * n + i] * xr[r];
            }
            part[wave * 64 + lane] = s;
            __syncthreads();
            if (wave == 0 && i < n) {
                s = part[lane] + part[64 + lane] + part[128 + lane] + part[192 + lane];
                if (a.uref) {
                    const int st = i / nu;
                    for (int j = 0; j < nu; ++j) s -= a.R[(i % nu) * nu + j] * a.uref[(size_t)m * a.uref_stride + st * nu + j];
                }
                f0[i] = s;
            }
            __syncthreads();
        }
    }
}

// orders the LDS traffic of one wave only: the global stores of a step (rows of Phi, F, f0) stay in flight
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

// ---- one model per instance, small nx: H, F, f0 WITHOUT the prediction matrices ----------------------------------------
// H = Gamma' Qbar Gamma + Rbar has the structure of the dynamics: with Psi(k, j) = A_k ... A_{j+1} and
//     P_k = Q_k + A_{k+1}' P_{k+1} A_{k+1}     (backwards from P_{N-1} = terminal weight)
// the block (i, j), i <= j, is  B_i' Psi(j, i)' P_j B_j  (+ R on the diagonal): for every column j one chain
//     V = P_j B_j;   for i = j .. 0:   H_ij = B_i' V,   V <- A_i' V.
// Likewise F_i = B_i' P_i Phi_i = V_i' Phi_i (forward, after the P_k) and f0_i = -B_i' s_i - R uref_i with
// s_k = Q_k xref_k + A_{k+1}' s_{k+1}.  O(N^2) small products instead of the (N nx) x n x n contraction, and neither Gamma nor
// Qbar Gamma is ever written: the 440 MB that kernel moved per step of BASELINE configs[3] (8,192 models) and 0.44 of the
// 0.89 ms the step took.  One WAVE per model: the model and the V_k live in LDS, lane (j, c) owns column j nu + c of H
// with its V column in registers.  Same quantities as assemble_kernel up to the order of summation.
// Measured for 8,192 bicycle models (nx 4, nu 2, N 20): 0.14 ms against 0.44 ms.  It is bound by instruction issue, not by
// memory or latency (~5,000 wave instructions per model, 16 of 64 lanes busy in the recursions, the triangle of chains half
// masked, 8 waves per SIMD in turn): staging H through LDS for whole-row stores (0.21 ms: LDS costs occupancy), the
// references in LDS instead of loads inside the chain, and dropping the mirrored stores (0.137 ms) all left it where it is.
// What would help is several models per wave, which the 10 KB of LDS per model do not allow at a useful occupancy.
// (Reference: the same cost and prediction model, LaunchPlannerQPFormulation.py:49-89,130-162; HelperFuncs.py:6-11.)
// MPW models per wave (4 when the NX x NX recursions need 16 lanes or fewer: they are the longer part of the kernel and kept
// 16 of 64 lanes busy with one model per wave), LPM = 64 / MPW lanes per model; `per` doubles of LDS per model.
// NU: the number of inputs when it is 1, 2 or 4 (loops over it unroll and the divisions by it are by a constant), 0 = a.nu
template <int NX, int MPW, int NU>
__global__ __launch_bounds__(64) void assemble_wave_kernel(AsmParams a, int n_models, int per) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int XX = NX * NX, LPM = 64 / MPW;
    static_assert(XX <= LPM, "the NX x NX recursions take one lane per element");
    const int q = threadIdx.x / LPM, lane = threadIdx.x % LPM;
    const int m = blockIdx.x * MPW + q;
    const bool live = m < n_models;
    const int mm = live ? m : 0;                        // a dead group computes model 0 again and stores nothing
    const int nu = NU ? NU : a.nu, N = a.N, n = a.n, nA = a.tv ? N : 1, XU = NX * nu;
    const double* Am = a.A + (size_t)mm * nA * XX;
    const double* Bm = a.B + (size_t)mm * nA * XU;
    double* Phi = a.Phi + (size_t)mm * a.p * NX;
    double* H = a.H + (size_t)mm * n * n;
    double* F = a.F + (size_t)mm * n * NX;
    double* f0 = a.f0 + (size_t)mm * n;
    double* sA = reinterpret_cast<double*>(smem) + (size_t)q * per;       // nA x (NX, NX)
    double* sB = sA + nA * XX;                          // nA x (NX, nu)
    double* sQ = sB + nA * XU;                          // Q, terminal P
    double* sR = sQ + 2 * XX;                           // (nu, nu)
    double* sV = sR + nu * nu;                          // N x (NX, nu): V_k = P_k B_k
    double* sP = sV + N * XU;                           // 2 x (NX, NX): P_k, ping-pong; later Phi_k, ping-pong
    double* sT = sP + 2 * XX;                           // (NX, NX): P_{k+1} A_{k+1}
    double* ss = sT + XX;                               // 2 x NX: s_k, ping-pong
    double* sxr = ss + 2 * NX;                          // (N, NX), (N, nu): the model's references -- nothing inside the
    double* sur = sxr + N * NX;                         // backward chain may wait on global memory
    const double* xr = a.xref ? a.xref + (size_t)mm * a.xref_stride : nullptr;
    const double* ur = a.uref ? a.uref + (size_t)mm * a.uref_stride : nullptr;
    {   // the model into LDS with every load in flight before the first LDS write waits for one: as plain loops the compiler
        // waited for each element before loading the next (38 round trips to memory per model with 16 lanes, a quarter of
        // the kernel's time with 64)
        constexpr int UA = 24, UB = 16, UX = 8, UU = 4;
        const int cA = nA * XX, cB = nA * XU, cX = N * NX, cU = N * nu;
        double vA[UA], vB[UB], vX[UX], vU[UU];
#pragma unroll
        for (int u = 0; u < UA; ++u) { const int i = lane + u * LPM; vA[u] = i < cA ? Am[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < UB; ++u) { const int i = lane + u * LPM; vB[u] = i < cB ? Bm[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < UX; ++u) { const int i = lane + u * LPM; vX[u] = (xr && i < cX) ? xr[i] : 0.0; }
#pragma unroll
        for (int u = 0; u < UU; ++u) { const int i = lane + u * LPM; vU[u] = (ur && i < cU) ? ur[i] : 0.0; }
        double vq = 0.0, vp = 0.0, vr = 0.0;
        if (lane < XX) { vq = a.Q[lane]; vp = a.P[lane]; }
        if (lane < nu * nu) vr = a.R[lane];
#pragma unroll
        for (int u = 0; u < UA; ++u) { const int i = lane + u * LPM; if (i < cA) sA[i] = vA[u]; }
#pragma unroll
        for (int u = 0; u < UB; ++u) { const int i = lane + u * LPM; if (i < cB) sB[i] = vB[u]; }
#pragma unroll
        for (int u = 0; u < UX; ++u) { const int i = lane + u * LPM; if (i < cX) sxr[i] = vX[u]; }
#pragma unroll
        for (int u = 0; u < UU; ++u) { const int i = lane + u * LPM; if (i < cU) sur[i] = vU[u]; }
        if (lane < XX) { sQ[lane] = vq; sQ[XX + lane] = vp; }
        if (lane < nu * nu) sR[lane] = vr;
        // what the batches above do not cover (long horizons)
        for (int i = lane + UA * LPM; i < cA; i += LPM) sA[i] = Am[i];
        for (int i = lane + UB * LPM; i < cB; i += LPM) sB[i] = Bm[i];
        for (int i = lane + UX * LPM; i < cX; i += LPM) sxr[i] = xr ? xr[i] : 0.0;
        for (int i = lane + UU * LPM; i < cU; i += LPM) sur[i] = ur ? ur[i] : 0.0;
        for (int i = lane + LPM; i < nu * nu; i += LPM) sR[i] = a.R[i];
    }
    wave_sync();
    const int er = lane / NX, ec = lane % NX;           // lane < XX: element (er, ec) of the NX x NX recursions
    // ---- backward: P_k, V_k = P_k B_k, s_k; rows of f0 of stage k as soon as s_k exists ---------------------------------------
    for (int k = N - 1; k >= 0; --k) {
        const double* Qk = (k == N - 1) ? sQ + XX : sQ;
        const double* A1 = sA + (a.tv ? (k + 1) * XX : 0);          // A_{k+1} (k < N-1 only)
        double* Pk = sP + (k & 1) * XX; const double* P1 = sP + ((k + 1) & 1) * XX;
        double* sk = ss + (k & 1) * NX; const double* s1 = ss + ((k + 1) & 1) * NX;
        if (k < N - 1 && lane < XX) {
            double t = 0.0;
#pragma unroll
            for (int b = 0; b < NX; ++b) t += P1[er * NX + b] * A1[b * NX + ec];
            sT[lane] = t;
        }
        wave_sync();
        if (lane < XX) {
            double pv = Qk[lane];
            if (k < N - 1) {
#pragma unroll
                for (int b = 0; b < NX; ++b) pv += A1[b * NX + er] * sT[b * NX + ec];
            }
            Pk[lane] = pv;
        }
        if (lane < NX) {
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < NX; ++b) v += Qk[lane * NX + b] * sxr[k * NX + b];
            if (k < N - 1) {
#pragma unroll
                for (int b = 0; b < NX; ++b) v += A1[b * NX + lane] * s1[b];
            }
            sk[lane] = v;
        }
        wave_sync();
        const double* Bk = sB + (a.tv ? k * XU : 0);
        for (int e = lane; e < XU; e += LPM) {                      // V_k = P_k B_k  (NX, nu)
            const int r = e / nu, c = e - r * nu;
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < NX; ++b) v += Pk[r * NX + b] * Bk[b * nu + c];
            sV[k * XU + e] = v;
        }
        for (int r = lane; r < nu; r += LPM) {                      // f0 rows of stage k
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < NX; ++b) v -= Bk[b * nu + r] * sk[b];
            for (int j = 0; j < nu; ++j) v -= sR[r * nu + j] * sur[k * nu + j];
            if (live) f0[k * nu + r] = v;
        }
    }
    wave_sync();
    // ---- forward: Phi_k, and with it the rows of F of stage k: S_k = sum_{l >= k} Psi(l,k)' Q_l Phi_l = P_k Phi_k because
    // Phi_l = Psi(l,k) Phi_k, so F_k = B_k' S_k = V_k' Phi_k -- no second recursion and no Phi_k kept ------------------------------
    double* sPhi = sP;
    for (int k = 0; k < N; ++k) {
        const double* Ak = sA + (a.tv ? k * XX : 0);
        double* pk = sPhi + (k & 1) * XX; const double* p1 = sPhi + ((k + 1) & 1) * XX;
        if (lane < XX) {
            double v;
            if (k == 0) v = Ak[lane];
            else {
                v = 0.0;
#pragma unroll
                for (int b = 0; b < NX; ++b) v += Ak[er * NX + b] * p1[b * NX + ec];
            }
            pk[lane] = v;
            if (live) Phi[(size_t)(k * NX + er) * NX + ec] = v;
        }
        wave_sync();
        const double* Vk = sV + k * XU;
        for (int e = lane; e < nu * NX; e += LPM) {                 // F rows of stage k: V_k' Phi_k
            const int r = e / NX, c = e % NX;
            double v = 0.0;
#pragma unroll
            for (int b = 0; b < NX; ++b) v += Vk[b * nu + r] * pk[b * NX + c];
            if (live) F[(size_t)(k * nu + r) * NX + c] = v;
        }
    }
    wave_sync();
    // ---- columns of H -----------------------------------------------------------------------------------------------------
    // Every lane of a model walks the stages in the SAME order i = jtop .. 0 and joins at its own stage j: the reads of
    // A_i, B_i are then one address per model (a broadcast) instead of one matrix per lane on two sets of banks
    // (a chain per lane, i = j .. 0, had the lanes of a wave in 20 different stages at once: 4 M conflict cycles per launch)
    for (int col0 = 0; col0 < n; col0 += LPM) {
        const int col = col0 + lane;
        const bool on = col < n;
        const int j = on ? col / nu : -1, c = on ? col - j * nu : 0;
        double v[NX];
#pragma unroll
        for (int r = 0; r < NX; ++r) v[r] = 0.0;
        const int jtop = (min(n, col0 + LPM) - 1) / nu;
        for (int i = jtop; i >= 0; --i) {
            const double* Bi = sB + (a.tv ? i * XU : 0);
            const double* Ai = sA + (a.tv ? i * XX : 0);
            if (i == j) {
#pragma unroll
                for (int r = 0; r < NX; ++r) v[r] = sV[j * XU + r * nu + c];
            }
            // the diagonal block is written from its upper triangle only, so that every entry of H has one writer and the
            // matrix is exactly symmetric
            const int rmax = i == j ? c : (i < j ? nu - 1 : -1);
            for (int r = 0; r < nu; ++r) {
                double hv = 0.0;
#pragma unroll
                for (int b = 0; b < NX; ++b) hv += Bi[b * nu + r] * v[b];
                if (r <= rmax && live) {
                    if (i == j) hv += sR[r * nu + c];
                    const int row = i * nu + r;
                    H[(size_t)row * n + col] = hv;
                    H[(size_t)col * n + row] = hv;
                }
            }
            if (i > 0) {
                double w[NX];
#pragma unroll
                for (int r = 0; r < NX; ++r) {
                    double t = 0.0;
#pragma unroll
                    for (int b = 0; b < NX; ++b) t += Ai[b * NX + r] * v[b];
                    w[r] = t;
                }
#pragma unroll
                for (int r = 0; r < NX; ++r) v[r] = w[r];
            }
        }
    }
}

template <int NX, int NU>
int launch_assemble_wave_nu(mpcqp_handle h, const AsmParams& a, size_t lds_model, hipStream_t st) {
    constexpr int MPW = NX * NX <= 16 ? 4 : (NX * NX <= 32 ? 2 : 1);
    // regions 16 (mod 32) doubles apart: the two models a 32-lane LDS cycle serves (lanes 0-31, 32-63) read the same offsets
    // of their regions, 128 bytes wide, and must fall on the two halves of the 256-byte bank line
    int per = (int)(lds_model / sizeof(double));
    per += (16 - per % 32 + 32) % 32;
    const size_t lds = (size_t)MPW * per * sizeof(double);
    hipError_t e = ensure_dyn_lds(h->dev, reinterpret_cast<const void*>(&assemble_wave_kernel<NX, MPW, NU>), lds);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return MPCQP_EHIP; }
    hipLaunchKernelGGL((assemble_wave_kernel<NX, MPW, NU>), dim3(ceil_div(h->dims.n_models, MPW)), dim3(64), lds, st, a, h->dims.n_models, per);
    return MPCQP_OK;
}

template <int NX>
int launch_assemble_wave_nx(mpcqp_handle h, const AsmParams& a, size_t lds_model, hipStream_t st) {
    switch (a.nu) {
        case 1: return launch_assemble_wave_nu<NX, 1>(h, a, lds_model, st);
        case 2: return launch_assemble_wave_nu<NX, 2>(h, a, lds_model, st);
        case 4: return launch_assemble_wave_nu<NX, 4>(h, a, lds_model, st);
        default: return launch_assemble_wave_nu<NX, 0>(h, a, lds_model, st);
    }
}

}  // namespace

// What the condensed QP needs without the prediction matrices (one model per instance, small nx): -1 if the shape is not
// covered, else the launch's return code
static int launch_assemble_wave(mpcqp_handle h, hipStream_t st) {
    const int nx = h->dims.nx, nu = h->dims.nu, N = h->dims.N;
    if (h->dims.n_models < 2 || nx > 8) return -1;
    const int nA = h->dims.time_varying ? N : 1, XX = nx * nx;
    const size_t lds = ((size_t)nA * (XX + nx * nu) + 2 * XX + nu * nu + (size_t)N * nx * nu + 3 * XX + 2 * nx +
                        (size_t)N * (nx + nu)) * sizeof(double);
    if (lds > 40 * 1024) return -1;
    AsmParams a;
    a.A = h->A; a.B = h->B; a.Q = h->Q; a.R = h->R; a.P = h->P; a.xref = h->xref; a.uref = h->uref;
    a.xref_stride = h->xref_stride; a.uref_stride = h->uref_stride;
    a.Gam = nullptr; a.QG = nullptr; a.Phi = h->Phi; a.H = h->H; a.F = h->F; a.f0 = h->f0;
    a.nx = nx; a.nu = nu; a.N = N; a.n = h->n; a.p = h->p; a.tv = h->dims.time_varying;
    a.rows_in_lds = 0; a.ab_in_lds = 0; a.q_in_lds = 0; a.phases = 0;
    int rc = MPCQP_OK;
    switch (nx) {
        case 1: rc = launch_assemble_wave_nx<1>(h, a, lds, st); break;
        case 2: rc = launch_assemble_wave_nx<2>(h, a, lds, st); break;
        case 3: rc = launch_assemble_wave_nx<3>(h, a, lds, st); break;
        case 4: rc = launch_assemble_wave_nx<4>(h, a, lds, st); break;
        case 5: rc = launch_assemble_wave_nx<5>(h, a, lds, st); break;
        case 6: rc = launch_assemble_wave_nx<6>(h, a, lds, st); break;
        case 7: rc = launch_assemble_wave_nx<7>(h, a, lds, st); break;
        default: rc = launch_assemble_wave_nx<8>(h, a, lds, st); break;
    }
    if (rc != MPCQP_OK) return rc;
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("assemble (wave) launch: ") + hipGetErrorString(e); return MPCQP_EHIP; }
    return MPCQP_OK;
}

// with_gamma: the caller needs Gamma / Qbar Gamma in memory (state or polytopic rows, mpcqp_get of Gamma)
// gamma_only (one model per instance): only the rollout phase -- Gamma, Qbar Gamma, Phi -- so that H, F, f0 keep the bits the
// factorisation was made from (an introspection call must not change what a later solve or mpcqp_get(H) sees)
int launch_assemble(mpcqp_handle h, hipStream_t st, bool with_gamma, bool gamma_only) {
    if (!with_gamma) {
        const int rc = launch_assemble_wave(h, st);
        if (rc >= 0) { if (rc == MPCQP_OK) h->gam_valid = false; return rc; }
    }
    h->gam_valid = true;
    AsmParams a;
    a.A = h->A; a.B = h->B; a.Q = h->Q; a.R = h->R; a.P = h->P; a.xref = h->xref; a.uref = h->uref;
    a.xref_stride = h->xref_stride; a.uref_stride = h->uref_stride;
    a.Gam = h->Gam; a.QG = h->QG; a.Phi = h->Phi; a.H = h->H; a.F = h->F; a.f0 = h->f0;
    a.nx = h->dims.nx; a.nu = h->dims.nu; a.N = h->dims.N; a.n = h->n; a.p = h->p; a.tv = h->dims.time_varying;
    const size_t pp = (size_t)(2 * a.nx * a.nx) * sizeof(double);           // Phi ping-pong; Q, P beside it while they fit
    a.q_in_lds = 2 * pp <= 160 * 1024;
    const size_t fixed = a.q_in_lds ? 2 * pp : pp;
    const size_t rows = (size_t)(2 * a.nx * a.n) * sizeof(double);
    const size_t ab = (size_t)(a.tv ? a.N : 1) * a.nx * (a.nx + a.nu) * sizeof(double);
    a.rows_in_lds = fixed + rows <= 160 * 1024;
    size_t lds = fixed + (a.rows_in_lds ? rows : 0);
    // the staged model is only worth LDS that does not cost occupancy (many small models) or that is tiny
    a.ab_in_lds = lds + ab <= (h->dims.n_models > 1 ? 16 * 1024 : 160 * 1024);
    if (a.ab_in_lds) lds += ab;
    lds = std::max<size_t>(lds, 256 * sizeof(double));                     // the f0 partial sums reuse the front of it
    if (lds > 160 * 1024) { h->err = "assemble: nx too large for the LDS buffers"; return MPCQP_ENOTIMPL; }
    hipError_t e = ensure_dyn_lds(h->dev, reinterpret_cast<const void*>(&assemble_kernel), lds);
    if (e != hipSuccess) { h->err = hipGetErrorString(e); return MPCQP_EHIP; }
    if (h->dims.n_models > 1) {
        a.phases = gamma_only ? 1 : 3;
        hipLaunchKernelGGL(assemble_kernel, dim3(h->dims.n_models), dim3(256), lds, st, a);
    } else {
        // one shared model: the rollout over up to 32 workgroups (one or more column tiles each), then the tiles of H and F
        // over enough workgroups that a wave has a handful of them
        const int CTn = ceil_div(a.n, 16), tiles = CTn * (CTn + 1) / 2;
        a.phases = 1;
        hipLaunchKernelGGL(assemble_kernel, dim3(1, std::min(CTn, 32)), dim3(256), lds, st, a);
        a.phases = 2;
        hipLaunchKernelGGL(assemble_kernel, dim3(1, std::max(1, std::min(256, ceil_div(tiles, 8)))), dim3(256), lds, st, a);
    }
    e = hipGetLastError();
    if (e != hipSuccess) { h->err = std::string("assemble launch: ") + hipGetErrorString(e); return MPCQP_EHIP; }
    return MPCQP_OK;
}
