// General dense convex QP on the device:  min 1/2 x'Px + q'x  s.t.  Gx <= h,  Ax = b.
//
// This backs the reference's literal boundary, cvxopt.solvers.qp(P, q, G, h[, A, b])
// (call sites CollocationBasedPlanner.py:59, LaunchPlannerQPFormulation.py:173,
// Launch3DPlanner.py:201, CollocationBasedPlanner2D.py:359, SE3Planner.py:181,183), for a batch
// of independent QPs of equal shape -- one 256-thread workgroup per QP, matrices in a global
// workspace (L2 resident; the reference's sizes are n = 2 .. 280).
//
// Algorithm (same family as the MPC polish, generalised to dense rows):
//   * Jacobi scaling of the variables, unit-norm constraint rows;
//   * proximal regularisation H = Ps + sigma I (P may be singular: CollocationBasedPlanner.py:46),
//     explicit Hi = H^-1 by Cholesky, outer proximal-point iterations x_{k+1} = argmin ... + sigma/2|x - x_k|^2;
//   * each proximal subproblem by the Goldfarb-Idnani dual active-set method on the Schur
//     complement S = N Hi N' of the active rows N (equalities first, never released); linearly
//     dependent rows are handled by dual-only steps;
//   * refinement steps on the final active set with Ps itself.
// Finite, needs no tolerance tuning, and lands on the vertex-exact minimiser (KKT ~1e-12), which is
// what the 1e-6 / 1e-8 parity bar needs; an interior-point or ADMM iteration would not on these
// badly scaled problems (LaunchPlannerQPFormulation.py: eig(P) = 1e-8 .. 3e-5).
#include "common.h"
#include <algorithm>
#include <mutex>

namespace {

struct DenseParams {
    const double *P, *q, *G, *h, *A, *b;       // [B][n*n], [B][n], [B][m*n], [B][m], [B][p*n], [B][p]
    long sP, sq, sG, sh, sA, sb;               // doubles between consecutive QPs (0: one array shared by all)
    double *x, *z, *y, *obj;                   // [B][n], [B][m], [B][p], [B]
    int *status, *iters;                       // [B]
    double* ws; long ws_stride;                // doubles per QP
    int* iws; long iws_stride;
    int n, m, p, B, max_steps, prox_iters, refine;
    int phase;      // 0: everything per QP; 1: shared set-up only (one Hessian / constraint matrix for the batch,
                    //    grid of 1, into the workspace of QP 0); 2: per-QP solve on that shared set-up
    double sigma_rel;
};

struct Ws {
    double *Ps, *Hi, *W, *T, *Cs, *Y, *L;      // n*n x4, (p+m)*n, K*n, K*K
    double *d, *qs, *cs, *rn, *xk, *x, *xu, *qk, *dv, *zv, *lv, *rv, *lam, *t1, *t2;
    int* act;
};

__device__ double wg_max(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] = fmax(red[tid], red[tid + s]); __syncthreads(); }
    const double r = red[0]; __syncthreads();
    return r;
}

__device__ double wg_sum(double v, double* red) {
    const int tid = threadIdx.x;
    red[tid] = v; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) { if (tid < s) red[tid] += red[tid + s]; __syncthreads(); }
    const double r = red[0]; __syncthreads();
    return r;
}

// arg-extreme: the lowest index attaining the maximum of v (idx < 0 = no candidate)
__device__ void wg_argmax(double v, int idx, double* red, int* redi, double& vout, int& iout) {
    const int tid = threadIdx.x;
    red[tid] = v; redi[tid] = idx; __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (tid < s) {
            const double o = red[tid + s]; const int oi = redi[tid + s];
            if (oi >= 0 && (redi[tid] < 0 || o > red[tid] || (o == red[tid] && oi < redi[tid]))) { red[tid] = o; redi[tid] = oi; }
        }
        __syncthreads();
    }
    vout = red[0]; iout = redi[0]; __syncthreads();
}

// in-place lower Cholesky of the k x k matrix W (leading dimension ld).  Returns -1 on success or the
// index of the first non-positive pivot.
__device__ int wg_chol(double* W, int ld, int k, int* flag) {
    const int tid = threadIdx.x;
    if (tid == 0) *flag = -1;
    __syncthreads();
    for (int j = 0; j < k; ++j) {
        if (tid == 0) {
            const double dj = W[(size_t)j * ld + j];
            if (!(dj > 0.0)) *flag = j;
            else W[(size_t)j * ld + j] = sqrt(dj);
        }
        __syncthreads();
        if (*flag >= 0) break;
        const double inv = 1.0 / W[(size_t)j * ld + j];
        for (int i = j + 1 + tid; i < k; i += 256) W[(size_t)i * ld + j] *= inv;
        __syncthreads();
        const int rem = k - j - 1;
        for (int idx = tid; idx < rem * rem; idx += 256) {
            const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
            if (c <= i) W[(size_t)i * ld + c] -= W[(size_t)i * ld + j] * W[(size_t)c * ld + j];
        }
        __syncthreads();
    }
    const int r = *flag;
    __syncthreads();
    return r;
}

// v <- L^-1 v  (forward) then optionally v <- L^-T v (backward); v has k entries in global memory
__device__ void wg_trsv(const double* L, int ld, int k, double* v, bool forward, bool backward) {
    const int tid = threadIdx.x;
    if (forward)
        for (int c = 0; c < k; ++c) {
            if (tid == 0) v[c] /= L[(size_t)c * ld + c];
            __syncthreads();
            const double yc = v[c];
            for (int i = c + 1 + tid; i < k; i += 256) v[i] -= L[(size_t)i * ld + c] * yc;
            __syncthreads();
        }
    if (backward)
        for (int c = k - 1; c >= 0; --c) {
            if (tid == 0) v[c] /= L[(size_t)c * ld + c];
            __syncthreads();
            const double yc = v[c];
            for (int i = tid; i < c; i += 256) v[i] -= L[(size_t)c * ld + i] * yc;
            __syncthreads();
        }
}

// X = (L L')^-1 for lower L (n x n, ld n) using T as scratch
__device__ void wg_inverse_from_chol(const double* W, double* T, double* X, int n) {
    const int tid = threadIdx.x;
    for (int i = 0; i < n; ++i) {
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

// out[i] = sum_j M[i][j] v[j]  for i < rows (M row-major, ld)
__device__ void wg_matvec(double* out, const double* M, int ld, int rows, int cols, const double* v) {
    for (int i = threadIdx.x; i < rows; i += 256) {
        double s = 0.0;
        for (int j = 0; j < cols; ++j) s += M[(size_t)i * ld + j] * v[j];
        out[i] = s;
    }
    __syncthreads();
}

// Y row a = Hi * Cs[act[a]]  for a in [a0, k);  S = N Y' lower -> L; Cholesky.  On a failing pivot the
// offending member is removed and the factorisation is redone.  Returns the (possibly reduced) k.
__device__ int rebuild(Ws& w, int n, int k, int a0, double* lamv, int* flag) {
    const int tid = threadIdx.x;
    for (;;) {
        for (int a = a0; a < k; ++a) wg_matvec(w.Y + (size_t)a * n, w.Hi, n, n, n, w.Cs + (size_t)w.act[a] * n);
        for (int idx = tid; idx < k * k; idx += 256) {
            const int a = idx / k, c = idx % k;
            if (c > a) continue;
            const double* na = w.Cs + (size_t)w.act[a] * n;
            const double* yc = w.Y + (size_t)c * n;
            double s = 0.0;
            for (int j = 0; j < n; ++j) s += na[j] * yc[j];
            w.L[(size_t)a * n + c] = s;
        }
        __syncthreads();
        // relative pivot floor: a pivot below 1e-12 of its own diagonal means a dependent row
        for (int a = tid; a < k; a += 256) w.t2[a] = w.L[(size_t)a * n + a];
        __syncthreads();
        int bad = -1;
        {
            // Cholesky with the floor test folded in
            if (tid == 0) *flag = -1;
            __syncthreads();
            for (int j = 0; j < k; ++j) {
                if (tid == 0) {
                    const double dj = w.L[(size_t)j * n + j];
                    if (!(dj > 1e-12 * w.t2[j])) { if (*flag < 0) *flag = j; }
                    else w.L[(size_t)j * n + j] = sqrt(dj);
                }
                __syncthreads();
                if (*flag >= 0) break;
                const double inv = 1.0 / w.L[(size_t)j * n + j];
                for (int i = j + 1 + tid; i < k; i += 256) w.L[(size_t)i * n + j] *= inv;
                __syncthreads();
                const int rem = k - j - 1;
                for (int idx = tid; idx < rem * rem; idx += 256) {
                    const int i = j + 1 + idx / rem, c = j + 1 + idx % rem;
                    if (c <= i) w.L[(size_t)i * n + c] -= w.L[(size_t)i * n + j] * w.L[(size_t)c * n + j];
                }
                __syncthreads();
            }
            bad = *flag;
            __syncthreads();
        }
        if (bad < 0) return k;
        // drop member `bad` (dependent on the ones before it): compact act, lam and Y
        for (int a = bad; a + 1 < k; ++a) {
            for (int j = tid; j < n; j += 256) w.Y[(size_t)a * n + j] = w.Y[(size_t)(a + 1) * n + j];
            if (tid == 0) { w.act[a] = w.act[a + 1]; lamv[a] = lamv[a + 1]; }
            __syncthreads();
        }
        --k;
        a0 = k;                     // every remaining Y row is valid
        if (k == 0) return 0;
    }
}

__global__ __launch_bounds__(256) void dense_qp_kernel(DenseParams p) {
    __shared__ double red[256];
    __shared__ int redi[256];
    __shared__ int sflag;
    const int tid = threadIdx.x, n = p.n, m = p.m, pe = p.p, mc = p.m + p.p, K = p.n;
    const int qi = blockIdx.x;
    const double* P = p.P + (size_t)qi * p.sP;
    const double* q = p.q + (size_t)qi * p.sq;
    const double* G = p.G ? p.G + (size_t)qi * p.sG : nullptr;
    const double* h = p.h ? p.h + (size_t)qi * p.sh : nullptr;
    const double* A = p.A ? p.A + (size_t)qi * p.sA : nullptr;
    const double* b = p.b ? p.b + (size_t)qi * p.sb : nullptr;
    Ws w;
    {
        double* c = p.ws + (size_t)qi * p.ws_stride;
        const int V = max(max(n, mc), 1);
        w.Ps = c; c += (size_t)n * n; w.Hi = c; c += (size_t)n * n; w.W = c; c += (size_t)n * n; w.T = c; c += (size_t)n * n;
        w.Cs = c; c += (size_t)mc * n; w.Y = c; c += (size_t)K * n; w.L = c; c += (size_t)K * n;
        double** vs[] = {&w.d, &w.qs, &w.cs, &w.rn, &w.xk, &w.x, &w.xu, &w.qk, &w.dv, &w.zv, &w.lv, &w.rv, &w.lam, &w.t1, &w.t2};
        for (auto v : vs) { *v = c; c += V; }
        w.act = p.iws + (size_t)qi * p.iws_stride;
    }
    // shared set-up (phase 2): scaling, scaled Hessian and rows, inverse of Ps + sigma I live in the workspace of QP 0;
    // this QP keeps its own vectors, Y, L and active list
    double* sh_d = nullptr; double* sh_rn = nullptr; double* sh_t2 = nullptr;
    {
        double* c0 = p.ws;
        const int V = max(max(n, mc), 1);
        double* Ps0 = c0; double* Hi0 = Ps0 + (size_t)n * n; double* Cs0 = Hi0 + (size_t)3 * n * n;
        double* v0 = Cs0 + (size_t)mc * n + (size_t)2 * K * n;
        sh_d = v0; sh_rn = v0 + (size_t)3 * V;
        sh_t2 = Hi0 + (size_t)n * n;        // the W block of QP 0: free once the inverse exists, never written in phase 2
        if (p.phase == 2) { w.Ps = Ps0; w.Hi = Hi0; w.Cs = Cs0; }
    }
    int status = 0;     // 0 optimal, 1 unknown (step cap), 2 infeasible, 3 numerical
    double sigma = 0.0;
    if (p.phase == 2) {
        for (int j = tid; j < n; j += 256) { w.d[j] = sh_d[j]; w.qs[j] = q[j] * sh_d[j]; w.xk[j] = 0.0; }
        for (int r = tid; r < mc; r += 256) { w.rn[r] = sh_rn[r]; w.cs[r] = (r < pe ? b[r] : h[r - pe]) / sh_rn[r]; }
        sigma = sh_t2[0];
        status = (int)sh_t2[1];
        __syncthreads();
    } else {
    // ---- scaling -------------------------------------------------------------------------------------
    for (int j = tid; j < n; j += 256) {
        const double dg = P[(size_t)j * n + j];
        double dj;
        if (dg > 0.0) dj = 1.0 / sqrt(dg);
        else {
            double cm = 0.0;
            for (int r = 0; r < pe; ++r) cm = fmax(cm, fabs(A[(size_t)r * n + j]));
            for (int r = 0; r < m; ++r) cm = fmax(cm, fabs(G[(size_t)r * n + j]));
            dj = cm > 0.0 ? 1.0 / cm : 1.0;
        }
        w.d[j] = dj;
    }
    __syncthreads();
    double dmax = 0.0;
    for (int idx = tid; idx < n * n; idx += 256) {
        const int i = idx / n, j = idx % n;
        const double v = 0.5 * (P[idx] + P[(size_t)j * n + i]) * w.d[i] * w.d[j];
        w.Ps[idx] = v;
        if (i == j) dmax = fmax(dmax, fabs(v));
    }
    for (int j = tid; j < n; j += 256) { w.qs[j] = q[j] * w.d[j]; w.xk[j] = 0.0; }
    dmax = wg_max(dmax, red);
    for (int r = tid; r < mc; r += 256) {
        const double* src = r < pe ? A + (size_t)r * n : G + (size_t)(r - pe) * n;
        double s = 0.0;
        for (int j = 0; j < n; ++j) { const double v = src[j] * w.d[j]; s += v * v; }
        s = sqrt(s);
        w.rn[r] = s > 0.0 ? s : 1.0;
    }
    __syncthreads();
    for (int idx = tid; idx < mc * n; idx += 256) {
        const int r = idx / n, j = idx % n;
        const double* src = r < pe ? A + (size_t)r * n : G + (size_t)(r - pe) * n;
        w.Cs[idx] = src[j] * w.d[j] / w.rn[r];
    }
    for (int r = tid; r < mc; r += 256) w.cs[r] = (r < pe ? b[r] : h[r - pe]) / w.rn[r];
    sigma = p.sigma_rel * (dmax > 0.0 ? dmax : 1.0);     // P = 0 (an LP): the scaled rows set the unit
    for (int idx = tid; idx < n * n; idx += 256) w.W[idx] = w.Ps[idx] + ((idx / n == idx % n) ? sigma : 0.0);
    __syncthreads();
    if (wg_chol(w.W, n, n, &sflag) >= 0) status = 3;     // P is not positive semidefinite
    if (status == 0) wg_inverse_from_chol(w.W, w.T, w.Hi, n);
    if (p.phase == 1) {     // leave sigma and the status for the per-QP launch
        __syncthreads();
        if (tid == 0) { w.W[0] = sigma; w.W[1] = (double)status; }
        return;
    }
    }

    // ---- proximal outer loop ---------------------------------------------------------------------------
    int k = 0, steps = 0, outer = 0;
    if (status == 0) {
        for (int a = tid; a < pe; a += 256) { w.act[a] = a; w.lam[a] = 0.0; }
        __syncthreads();
        k = rebuild(w, n, pe, 0, w.lam, &sflag);
    }
    for (; status == 0 && outer < p.prox_iters; ++outer) {
        for (int j = tid; j < n; j += 256) w.qk[j] = w.qs[j] - sigma * w.xk[j];
        __syncthreads();
        wg_matvec(w.xu, w.Hi, n, n, n, w.qk);
        for (int j = tid; j < n; j += 256) w.xu[j] = -w.xu[j];
        __syncthreads();
        // warm phase: solve on the current set, release wrong-signed inequality multipliers
        for (;;) {
            for (int a = tid; a < k; a += 256) {
                const double* na = w.Cs + (size_t)w.act[a] * n;
                double s = -w.cs[w.act[a]];
                for (int j = 0; j < n; ++j) s += na[j] * w.xu[j];
                w.lam[a] = s;
            }
            __syncthreads();
            wg_trsv(w.L, n, k, w.lam, true, true);
            int nneg = 0;
            for (int a = tid; a < k; a += 256) if (w.act[a] >= pe && w.lam[a] < 0.0) ++nneg;
            const double tot = wg_sum((double)nneg, red);
            if (tot == 0.0) break;
            if (tid == 0) {         // compact (k is small; serial is fine)
                int kk = 0;
                for (int a = 0; a < k; ++a)
                    if (!(w.act[a] >= pe && w.lam[a] < 0.0)) { w.act[kk] = w.act[a]; ++kk; }
                sflag = kk;
            }
            __syncthreads();
            k = sflag;
            __syncthreads();
            steps += (int)tot;
            k = rebuild(w, n, k, 0, w.lam, &sflag);
        }
        for (int j = tid; j < n; j += 256) {
            double s = w.xu[j];
            for (int a = 0; a < k; ++a) s -= w.Y[(size_t)a * n + j] * w.lam[a];
            w.x[j] = s;
        }
        __syncthreads();

        // main phase
        for (;;) {
            double best = 0.0; int bi = -1;
            for (int r = tid; r < m; r += 256) {
                const double* g = w.Cs + (size_t)(pe + r) * n;
                double s = -w.cs[pe + r];
                for (int j = 0; j < n; ++j) s += g[j] * w.x[j];
                w.t1[r] = s;
            }
            __syncthreads();
            for (int a = tid; a < k; a += 256) if (w.act[a] >= pe) w.t1[w.act[a] - pe] = -1.0;
            __syncthreads();
            for (int r = tid; r < m; r += 256) {
                const double v = w.t1[r];
                if (v > 1e-12 * fmax(1.0, fabs(w.cs[pe + r])) && (bi < 0 || v > best)) { best = v; bi = r; }
            }
            double v0; int pr;
            wg_argmax(best, bi, red, redi, v0, pr);
            if (pr < 0) break;
            const int pp = pe + pr;
            double accl = 0.0;
            bool added = false;
            while (!added && status == 0) {
                if (++steps > p.max_steps) { status = 1; break; }
                const double* np_ = w.Cs + (size_t)pp * n;
                wg_matvec(w.dv, w.Hi, n, n, n, np_);
                double part = 0.0;
                for (int j = tid; j < n; j += 256) part += np_[j] * w.dv[j];
                const double nHn = wg_sum(part, red);
                for (int a = tid; a < k; a += 256) {
                    const double* na = w.Cs + (size_t)w.act[a] * n;
                    double s = 0.0;
                    for (int j = 0; j < n; ++j) s += na[j] * w.dv[j];
                    w.lv[a] = s;
                }
                __syncthreads();
                wg_trsv(w.L, n, k, w.lv, true, false);
                part = 0.0;
                for (int a = tid; a < k; a += 256) { part += w.lv[a] * w.lv[a]; w.rv[a] = w.lv[a]; }
                const double d2 = nHn - wg_sum(part, red);
                wg_trsv(w.L, n, k, w.rv, false, true);
                const bool dep = !(d2 > 1e-13 * nHn);
                const double t2 = dep ? 1e300 : v0 / d2;
                double t1 = 1e300; int jb = -1;
                for (int a = tid; a < k; a += 256)
                    if (w.act[a] >= pe && w.rv[a] > 0.0) {
                        const double t = w.lam[a] / w.rv[a];
                        if (t < t1) { t1 = t; jb = a; }
                    }
                double nt1; int njb;
                wg_argmax(-t1, jb, red, redi, nt1, njb);
                t1 = njb >= 0 ? -nt1 : 1e300; jb = njb;
                if (t1 >= 1e300 && t2 >= 1e300) { status = 2; break; }       // infeasible
                const bool full = !dep && t2 <= t1;
                const double t = full ? t2 : t1;
                if (!dep)
                    for (int j = tid; j < n; j += 256) {
                        double zj = w.dv[j];
                        for (int a = 0; a < k; ++a) zj -= w.Y[(size_t)a * n + j] * w.rv[a];
                        w.x[j] -= t * zj;
                    }
                for (int a = tid; a < k; a += 256) w.lam[a] -= t * w.rv[a];
                accl += t;
                __syncthreads();
                if (full) {
                    if (k >= K) { status = 3; break; }
                    for (int j = tid; j < n; j += 256) w.Y[(size_t)k * n + j] = w.dv[j];
                    for (int a = tid; a < k; a += 256) w.L[(size_t)k * n + a] = w.lv[a];
                    if (tid == 0) { w.L[(size_t)k * n + k] = sqrt(d2); w.act[k] = pp; w.lam[k] = accl; }
                    __syncthreads();
                    ++k;
                    added = true;
                } else {
                    if (!dep) v0 -= t * d2;
                    // release member jb
                    for (int a = jb; a + 1 < k; ++a) {
                        for (int j = tid; j < n; j += 256) w.Y[(size_t)a * n + j] = w.Y[(size_t)(a + 1) * n + j];
                        if (tid == 0) { w.act[a] = w.act[a + 1]; w.lam[a] = w.lam[a + 1]; }
                        __syncthreads();
                    }
                    --k;
                    k = rebuild(w, n, k, k, w.lam, &sflag);
                }
            }
            if (status != 0) break;
        }
        if (status != 0) break;
        // proximal convergence
        double dxm = 0.0, xm = 0.0;
        for (int j = tid; j < n; j += 256) { dxm = fmax(dxm, fabs(w.x[j] - w.xk[j])); xm = fmax(xm, fabs(w.x[j])); w.xk[j] = w.x[j]; }
        dxm = wg_max(dxm, red); xm = wg_max(xm, red);
        if (dxm <= 1e-13 * fmax(xm, 1e-300)) { ++outer; break; }
    }

    // ---- refinement on the final active set with Ps itself ------------------------------------------------
    if (status == 0)
        for (int itr = 0; itr < p.refine; ++itr) {
            // g = Ps x + qs + N' lam ; rc = N x - cs
            for (int j = tid; j < n; j += 256) {
                double s = w.qs[j];
                for (int c = 0; c < n; ++c) s += w.Ps[(size_t)j * n + c] * w.x[c];
                for (int a = 0; a < k; ++a) s += w.Cs[(size_t)w.act[a] * n + j] * w.lam[a];
                w.t1[j] = s;
            }
            __syncthreads();
            wg_matvec(w.dv, w.Hi, n, n, n, w.t1);                 // dv = Hi g ; dx0 = -dv
            for (int a = tid; a < k; a += 256) {
                const double* na = w.Cs + (size_t)w.act[a] * n;
                double s = -w.cs[w.act[a]];
                for (int j = 0; j < n; ++j) s += na[j] * (w.x[j] - w.dv[j]);
                w.lv[a] = s;                                          // N dx0 + rc
            }
            __syncthreads();
            wg_trsv(w.L, n, k, w.lv, true, true);                 // dl
            for (int j = tid; j < n; j += 256) {
                double s = -w.dv[j];
                for (int a = 0; a < k; ++a) s -= w.Y[(size_t)a * n + j] * w.lv[a];
                w.x[j] += s;
            }
            for (int a = tid; a < k; a += 256) w.lam[a] += w.lv[a];
            __syncthreads();
        }

    // dropped (dependent) equalities must still hold, otherwise the problem is infeasible
    if (status == 0 && pe > 0) {
        double worst = 0.0;
        for (int r = tid; r < pe; r += 256) {
            const double* na = w.Cs + (size_t)r * n;
            double s = -w.cs[r];
            for (int j = 0; j < n; ++j) s += na[j] * w.x[j];
            worst = fmax(worst, fabs(s) / fmax(1.0, fabs(w.cs[r])));
        }
        worst = wg_max(worst, red);
        if (worst > 1e-8) status = 2;
    }

    // ---- outputs -----------------------------------------------------------------------------------------
    double* xo = p.x + (size_t)qi * n;
    for (int j = tid; j < n; j += 256) xo[j] = status == 3 ? 0.0 : w.x[j] * w.d[j];
    if (p.z) for (int r = tid; r < m; r += 256) p.z[(size_t)qi * m + r] = 0.0;
    if (p.y) for (int r = tid; r < pe; r += 256) p.y[(size_t)qi * pe + r] = 0.0;
    __syncthreads();
    for (int a = tid; a < k; a += 256) {
        const int r = w.act[a];
        const double v = w.lam[a] / w.rn[r];
        if (r >= pe) { if (p.z) p.z[(size_t)qi * m + (r - pe)] = v; }
        else if (p.y) p.y[(size_t)qi * pe + r] = v;
    }
    double part = 0.0;
    for (int j = tid; j < n; j += 256) {
        double s = 0.0;
        for (int c = 0; c < n; ++c) s += w.Ps[(size_t)j * n + c] * w.x[c];
        part += w.x[j] * (0.5 * s + w.qs[j]);
    }
    const double obj = wg_sum(part, red);
    if (tid == 0) {
        if (p.obj) p.obj[qi] = obj;
        if (p.status) p.status[qi] = status;
        if (p.iters) p.iters[qi] = steps + 1000 * outer;
    }
}

}  // namespace

// blocked_setup.hip: the same set-up spread over the GPU (scaling kernels, blocked Cholesky and inverse on the MFMA GEMM)
bool blocked_setup_fits(int n);
int blocked_shared_setup(int n, int m, int pe, const double* P, const double* G, const double* A, double* ws, int* iws,
                         double sigma_rel, hipStream_t st);

static int shared_setup(int n, int m, int p, const double* P, const double* G, const double* A, double* workspace,
                        int* int_workspace, double sigma_rel, hipStream_t stream) {
    if (blocked_setup_fits(n))
        return blocked_shared_setup(n, m, p, P, m ? G : nullptr, p ? A : nullptr, workspace, int_workspace, sigma_rel, stream);
    DenseParams d{};
    d.P = P; d.q = workspace; d.G = m ? G : nullptr; d.h = m ? workspace : nullptr; d.A = p ? A : nullptr; d.b = p ? workspace : nullptr;
    long ic = 0;
    d.ws = workspace; d.ws_stride = mpcqp_dense_workspace(n, m, p, &ic);
    d.iws = int_workspace; d.iws_stride = ic;
    d.n = n; d.m = m; d.p = p; d.B = 1;
    d.max_steps = 0; d.prox_iters = 0; d.refine = 0; d.sigma_rel = sigma_rel; d.phase = 1;
    hipLaunchKernelGGL(dense_qp_kernel, dim3(1), dim3(256), 0, stream, d);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

// rows_qp.hip: the same shared set-up without the proximal term (P positive definite), inequality rows only
int dense_shared_setup_sigma(int n, int m, const double* P, const double* G, double* workspace, int* int_workspace,
                             double sigma_rel, hipStream_t stream) {
    return shared_setup(n, m, 0, P, G, nullptr, workspace, int_workspace, sigma_rel, stream);
}

extern "C" {

long mpcqp_dense_workspace(int n, int m, int p, long* int_count) {
    const long mc = (long)m + p, V = std::max<long>(std::max<long>(n, mc), 1);
    if (int_count) *int_count = n > 0 ? n : 1;
    return 4L * n * n + mc * n + 2L * n * n + 15L * V;
}

void mpcqp_default_dense_opts(mpcqp_dense_opts* o) {
    if (!o) return;
    o->max_steps = 0; o->prox_iters = 30; o->refine = 2; o->sigma_rel = 1e-6;
}

int mpcqp_solve_dense_strided(int device_id, int batch, int n, int m, int p, const double* P, long sP, const double* q,
                              long sq, const double* G, long sG, const double* h, long sh, const double* A, long sA,
                              const double* b, long sb, double* x, double* z, double* y, double* obj, int* status,
                              int* iters, double* workspace, int* int_workspace, void* stream) {
    return mpcqp_solve_dense_opts(device_id, batch, n, m, p, P, sP, q, sq, G, sG, h, sh, A, sA, b, sb, x, z, y, obj, status,
                                  iters, workspace, int_workspace, nullptr, stream);
}

int mpcqp_solve_dense_opts(int device_id, int batch, int n, int m, int p, const double* P, long sP, const double* q,
                           long sq, const double* G, long sG, const double* h, long sh, const double* A, long sA,
                           const double* b, long sb, double* x, double* z, double* y, double* obj, int* status,
                           int* iters, double* workspace, int* int_workspace, const mpcqp_dense_opts* opts, void* stream) {
    mpcqp_dense_opts o;
    if (opts) o = *opts; else mpcqp_default_dense_opts(&o);
    if (o.prox_iters < 1 || o.refine < 0 || !(o.sigma_rel > 0.0)) return MPCQP_EINVAL;
    if (batch < 1 || n < 1 || m < 0 || p < 0 || !P || !q || !x || !workspace || !int_workspace) return MPCQP_EINVAL;
    if ((m > 0 && (!G || !h)) || (p > 0 && (!A || !b))) return MPCQP_EINVAL;
    if (sP < 0 || sq < 0 || sG < 0 || sh < 0 || sA < 0 || sb < 0) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    DenseParams d;
    d.P = P; d.q = q; d.G = m ? G : nullptr; d.h = m ? h : nullptr; d.A = p ? A : nullptr; d.b = p ? b : nullptr;
    d.sP = sP; d.sq = sq; d.sG = sG; d.sh = sh; d.sA = sA; d.sb = sb;
    d.x = x; d.z = z; d.y = y; d.obj = obj; d.status = status; d.iters = iters;
    long ic = 0;
    d.ws = workspace; d.ws_stride = mpcqp_dense_workspace(n, m, p, &ic);
    d.iws = int_workspace; d.iws_stride = ic;
    d.n = n; d.m = m; d.p = p; d.B = batch;
    d.max_steps = o.max_steps > 0 ? o.max_steps : 20 * (n + m + p) + 100;
    d.prox_iters = o.prox_iters; d.refine = o.refine; d.sigma_rel = o.sigma_rel;
    // one Hessian and one set of rows for the whole batch: scale and factor them ONCE (a grid of one workgroup into the
    // workspace of QP 0), then every QP solves on that shared set-up with its own q, h, b
    const bool shared = batch > 1 && sP == 0 && (m == 0 || sG == 0) && (p == 0 || sA == 0);
    if (shared) {
        const int rc = shared_setup(n, m, p, P, G, A, workspace, int_workspace, o.sigma_rel, static_cast<hipStream_t>(stream));
        if (rc != MPCQP_OK) return rc;
        d.phase = 2;
    } else d.phase = 0;
    hipLaunchKernelGGL(dense_qp_kernel, dim3(batch), dim3(256), 0, static_cast<hipStream_t>(stream), d);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

int mpcqp_dense_shared_setup(int device_id, int n, int m, int p, const double* P, const double* G, const double* A,
                             double* workspace, int* int_workspace, void* stream) {
    if (n < 1 || m < 0 || p < 0 || !P || !workspace || !int_workspace || (m > 0 && !G) || (p > 0 && !A)) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    return shared_setup(n, m, p, P, G, A, workspace, int_workspace, 1e-6, static_cast<hipStream_t>(stream));
}

int mpcqp_dense_shared_solve(int device_id, int batch, int n, int m, int p, const double* q, const double* h, const double* b,
                             double* x, double* z, double* y, double* obj, int* status, int* iters,
                             double* workspace, int* int_workspace, void* stream) {
    if (batch < 1 || n < 1 || m < 0 || p < 0 || !q || !x || !workspace || !int_workspace || (m > 0 && !h) || (p > 0 && !b))
        return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    DenseParams d{};
    d.P = nullptr; d.q = q; d.G = nullptr; d.h = m ? h : nullptr; d.A = nullptr; d.b = p ? b : nullptr;
    d.sq = n; d.sh = m; d.sb = p;
    d.x = x; d.z = z; d.y = y; d.obj = obj; d.status = status; d.iters = iters;
    long ic = 0;
    d.ws = workspace; d.ws_stride = mpcqp_dense_workspace(n, m, p, &ic);
    d.iws = int_workspace; d.iws_stride = ic;
    d.n = n; d.m = m; d.p = p; d.B = batch;
    d.max_steps = 20 * (n + m + p) + 100; d.prox_iters = 30; d.refine = 2; d.sigma_rel = 1e-6; d.phase = 2;
    hipLaunchKernelGGL(dense_qp_kernel, dim3(batch), dim3(256), 0, static_cast<hipStream_t>(stream), d);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

int mpcqp_solve_dense(int device_id, int batch, int n, int m, int p, const double* P, const double* q,
                      const double* G, const double* h, const double* A, const double* b,
                      double* x, double* z, double* y, double* obj, int* status, int* iters,
                      double* workspace, int* int_workspace, void* stream) {
    return mpcqp_solve_dense_strided(device_id, batch, n, m, p, P, (long)n * n, q, n, G, (long)m * n, h, m, A, (long)p * n,
                                     b, p, x, z, y, obj, status, iters, workspace, int_workspace, stream);
}

}  // extern "C"

// ---- the same solve for callers that cannot allocate (HIP graph capture, serving loops, harnesses that forbid hipMalloc):
// the workspace is a file-scope device array instead of an argument.  Every GPU holds its own copy of the array at its own
// address; the name `scratch` is only a host-side stand-in for it, so host code goes through the per-device pointer that
// hipGetSymbolAddress returns and never through the name.
namespace {

constexpr long STATIC_WS_DOUBLES = MPCQP_DENSE_STATIC_DOUBLES;
constexpr long STATIC_WS_INTS = MPCQP_DENSE_STATIC_INTS;
constexpr int STATIC_MAX_DEVICES = 64;

static __device__ __align__(256) double scratch[STATIC_WS_DOUBLES + STATIC_WS_INTS / 2];

double* g_scratch_ptr[STATIC_MAX_DEVICES];      // device address of `scratch` on each GPU, resolved once per device
std::mutex g_scratch_mutex;

// device_id must be the current device
int static_scratch(int device_id, double** out) {
    if (device_id < 0 || device_id >= STATIC_MAX_DEVICES) return MPCQP_EINVAL;
    std::lock_guard<std::mutex> lock(g_scratch_mutex);
    if (!g_scratch_ptr[device_id]) {
        void* ptr = nullptr;
        if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch)) != hipSuccess || !ptr) return MPCQP_EHIP;
        g_scratch_ptr[device_id] = static_cast<double*>(ptr);
    }
    *out = g_scratch_ptr[device_id];
    return MPCQP_OK;
}

}  // namespace

extern "C" {

int mpcqp_solve_dense_static(int device_id, int batch, int n, int m, int p, const double* P, const double* q,
                             const double* G, const double* h, const double* A, const double* b,
                             double* x, double* z, double* y, double* obj, int* status, int* iters, void* stream) {
    if (batch < 1 || n < 1 || m < 0 || p < 0) return MPCQP_EINVAL;
    long ic = 0;
    const long per = mpcqp_dense_workspace(n, m, p, &ic);
    // sizes beyond the static array are an error, never a silent allocation
    if (per > STATIC_WS_DOUBLES / batch || ic > STATIC_WS_INTS / batch) return MPCQP_ENOTIMPL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    double* ws = nullptr;
    const int rc = static_scratch(device_id, &ws);
    if (rc != MPCQP_OK) return rc;
    int* iws = reinterpret_cast<int*>(ws + STATIC_WS_DOUBLES);
    // the kernel writes every workspace entry before it reads it (callers of mpcqp_solve_dense pass uninitialised
    // memory), so the contents the previous call left behind need no clearing
    return mpcqp_solve_dense(device_id, batch, n, m, p, P, q, G, h, A, b, x, z, y, obj, status, iters, ws, iws, stream);
}

}  // extern "C"
