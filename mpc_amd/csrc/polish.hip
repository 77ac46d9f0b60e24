// Polish / finalise: turn the ADMM iterate of every instance into the exact minimiser of its
// box-constrained QP and write the caller-visible outputs.
//
// Reference counterpart: the last interior-point iterations of cvxopt.solvers.qp, i.e. what makes
// soln['x'] (LaunchPlannerQPFormulation.py:174) accurate to the solver tolerance.  ADMM alone
// converges too slowly for the 1e-6 / 1e-8 parity bar on condensed MPC Hessians (cond ~ 1e7), so,
// as in OSQP, the iterate is only used to GUESS the active set; the guess is then corrected by a
// dual active-set method (Goldfarb-Idnani) that works on the Schur complement S = Hi[A,A] of the
// shared inverse Hi = Hs^-1:
//     lam_A = S^-1 (x_unc[A] - bnd[A]),     x = x_unc - Hi[:,A] lam_A
//   * warm phase: constraints of the guess whose multiplier has the wrong sign are released;
//   * main phase: the most violated bound p is added; the step along
//       z = sp Hi[:,p] - Hi[:,A] r,  r = sp S^-1 Hi[A,p]
//     is cut short (and a constraint dropped) if a multiplier would change sign.
// Finite termination for positive definite H; with a good guess it is one solve and no steps.
// An optional refinement step uses Hs itself so that the answer does not inherit the rounding of
// the explicit inverse.
//
// Mapping: one wave per instance, lane j owns rows j and j+64 (n <= 128); the packed Cholesky factor
// of S lives in LDS (KMAX = 32 rows in the fast pass; instances that need more are redone by a
// second launch with KMAX = 128 and one wave per workgroup).  No workgroup barriers.
#include "common.h"
#include "polish.h"
#include <cstdlib>

namespace {


__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

// broadcast from a WAVE-UNIFORM source lane: two v_readlane_b32 instead of two ds_bpermute round trips
__device__ __forceinline__ double bcast(double v, int src) {
    const int sl = __builtin_amdgcn_readfirstlane(src);
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), sl);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), sl);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int bcast_i(int v, int src) {
    return __builtin_amdgcn_readlane(v, __builtin_amdgcn_readfirstlane(src));
}

// value owned by "row" i of a 2-slot per-lane vector (rows lane, lane+64)
__device__ __forceinline__ double row_get(double v0, double v1, int i) {
    return i < 64 ? bcast(v0, i) : bcast(v1, i - 64);
}

template <int KMAX>
struct WaveLds {
    double L[KMAX * (KMAX + 1) / 2];
    double xs[128];        // current x (scaled), indexable by row
    double gs[128];        // scratch vector indexable by row
    double di[KMAX];       // 1 / L[c][c]
    int A[KMAX];           // active rows, signed: +(row+1) upper, -(row+1) lower
};

template <int KMAX>
__device__ __forceinline__ double& Lat(WaveLds<KMAX>& s, int i, int m) { return s.L[i * (i + 1) / 2 + m]; }

// Cholesky of S = Hi[A,A] (k x k) into s.L.  Returns false on a non-positive pivot.
template <int KMAX>
__device__ bool chol_S(WaveLds<KMAX>& s, const double* Hi, int n, int k, int lane) {
    // stage the lower triangle of S in place first, with every load in flight at once (one
    // dependent L2 round trip instead of one per column)
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int i = lane + 64 * q;
        if (i < k) {
            const size_t ri = (size_t)(abs(s.A[i]) - 1) * n;
            for (int c = 0; c <= i; ++c) Lat(s, i, c) = Hi[ri + (abs(s.A[c]) - 1)];
        }
    }
    wave_sync();
    bool ok = true;
    for (int c = 0; c < k; ++c) {
        double t[2];
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lane + 64 * q;
            t[q] = 0.0;
            if (i >= c && i < k) {
                double v = Lat(s, i, c);
                for (int m = 0; m < c; ++m) v -= Lat(s, i, m) * Lat(s, c, m);
                t[q] = v;
            }
        }
        const double dcc = row_get(t[0], t[1], c);
        if (!(dcc > 0.0)) { ok = false; break; }
        const double dinv = 1.0 / sqrt(dcc);
        wave_sync();        // every lane has read row c's old entries before they are overwritten
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = lane + 64 * q;
            if (i >= c && i < k) Lat(s, i, c) = (i == c) ? sqrt(dcc) : t[q] * dinv;
            if (i == c) s.di[c] = dinv;
        }
        wave_sync();
    }
    return ok;
}

// in-register forward solve L y = v (rows lane, lane+64)
template <int KMAX>
__device__ __forceinline__ void fwd_solve(WaveLds<KMAX>& s, int k, int lane, double& v0, double& v1) {
    for (int c = 0; c < k; ++c) {
        const double yc = row_get(v0, v1, c) * s.di[c];
        if (lane == c) v0 = yc;
        if (lane + 64 == c) v1 = yc;
        if (lane > c && lane < k) v0 -= Lat(s, lane, c) * yc;
        if (lane + 64 > c && lane + 64 < k) v1 -= Lat(s, lane + 64, c) * yc;
    }
}

// in-register backward solve L' y = v
template <int KMAX>
__device__ __forceinline__ void bwd_solve(WaveLds<KMAX>& s, int k, int lane, double& v0, double& v1) {
    for (int c = k - 1; c >= 0; --c) {
        const double yc = row_get(v0, v1, c) * s.di[c];
        if (lane == c) v0 = yc;
        if (lane + 64 == c) v1 = yc;
        if (lane < c) v0 -= Lat(s, c, lane) * yc;
        if (lane + 64 < c) v1 -= Lat(s, c, lane + 64) * yc;
    }
}


// acc_q -= sum_a Hi[row(A_a)][lane + 64 q] * coef_a, with the list (rows) and the coefficients held in
// list order in registers (lane a <-> member a): row index and coefficient become scalars by readlane,
// so the loop is scalar address arithmetic + coalesced row loads + FMAs, with no LDS in it.
template <int KMAX>
__device__ __forceinline__ void rows_accum(const WaveLds<KMAX>& s, const double* Hi, int n, int k, int lane,
                                           double c0, double c1, double& acc0, double& acc1) {
    const int a0 = lane < k ? abs(s.A[lane]) - 1 : 0;
    const int a1 = lane + 64 < k ? abs(s.A[lane + 64]) - 1 : 0;
    const bool in0 = lane < n, in1 = lane + 64 < n;
    for (int a = 0; a < k; ++a) {
        const int r = a < 64 ? bcast_i(a0, a) : bcast_i(a1, a - 64);
        const double cf = a < 64 ? bcast(c0, a) : bcast(c1, a - 64);
        const double* row = Hi + (size_t)r * n;
        if (in0) acc0 -= row[lane] * cf;
        if (in1) acc1 -= row[lane + 64] * cf;
    }
}

// rebuild the ordered active list from per-row side flags; returns k (may exceed KMAX: then
// nothing beyond KMAX is stored)
template <int KMAX>
__device__ int build_list(WaveLds<KMAX>& s, const int (&side)[2], int lane) {
    int k = 0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const unsigned long long m = __ballot(side[q] != 0);
        const int pos = k + __popcll(m & ((1ull << lane) - 1ull));
        if (side[q] != 0 && pos < KMAX) s.A[pos] = side[q] * (lane + 64 * q + 1);
        k += __popcll(m);
    }
    wave_sync();
    return k;
}

// one instance, one wave
template <int KMAX>
__device__ void polish_wave(const PolParams& pin, const int inst, WaveLds<KMAX>& s, const int lane) {
    PolParams p = pin;
    if (p.st_nn) {          // this instance's own model
        p.Hs += (size_t)inst * p.st_nn; p.Hi += (size_t)inst * p.st_nn;
        p.FsT += (size_t)inst * p.st_k; p.KuT += (size_t)inst * p.st_k;
        p.lbs += (size_t)inst * p.st_np; p.ubs += (size_t)inst * p.st_np;
        p.d += (size_t)inst * p.st_n;
    }
    const int n = p.n;
    const double* Hi = p.Hi;

    // ---- per-row data ---------------------------------------------------------------------
    double xu[2], qs[2], x[2], lo[2], hi[2], lam[2];
    int side[2];
    double* zb = p.z + (size_t)inst * p.NP;
    double* wb = p.w + (size_t)inst * p.NP;
    // x0 of this instance, one element per lane, broadcast on use
    const double x0l = (lane < p.nx) ? p.x0[(size_t)inst * p.nx + lane] : 0.0;
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        xu[q] = qs[q] = x[q] = lo[q] = hi[q] = lam[q] = 0.0;
        side[q] = 0;
    }
    {
        const bool in0 = lane < n, in1 = lane + 64 < n;
        double su0 = in0 ? p.KuT[(size_t)p.nx * p.NP + lane] : 0.0, su1 = in1 ? p.KuT[(size_t)p.nx * p.NP + lane + 64] : 0.0;
        double sq0 = in0 ? p.FsT[(size_t)p.nx * p.NP + lane] : 0.0, sq1 = in1 ? p.FsT[(size_t)p.nx * p.NP + lane + 64] : 0.0;
        for (int c = 0; c < p.nx; ++c) {
            const double xc = bcast(x0l, c);
            if (in0) { su0 += p.KuT[(size_t)c * p.NP + lane] * xc; sq0 += p.FsT[(size_t)c * p.NP + lane] * xc; }
            if (in1) { su1 += p.KuT[(size_t)c * p.NP + lane + 64] * xc; sq1 += p.FsT[(size_t)c * p.NP + lane + 64] * xc; }
        }
        xu[0] = su0; xu[1] = su1; qs[0] = sq0; qs[1] = sq1;
    }
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < n) {
            lo[q] = p.lbs[j]; hi[q] = p.ubs[j];
            const bool guess = !p.cold || (p.hand_zw && p.need_big[inst] == 3);     // a kept iterate, or the set the last pass left
            const double zz = guess ? zb[j] : 0.0, ww = guess ? wb[j] : 0.0;
            x[q] = zz;
            lam[q] = p.rho * ww;
            // (z, w) are recovered from the kernel's (v, c) state, so "at the bound" and "w != 0"
            // hold up to rounding only
            const double tz = 1e-9 * bound_scale(lo[q], hi[q]);
            const double tw = 1e-12 * fmax(1.0, fabs(zz));
            side[q] = (zz >= hi[q] - tz && ww > tw) ? 1 : ((zz <= lo[q] + tz && ww < -tw) ? -1 : 0);
        }
    }

    int status = MPCQP_SOLVED_INACCURATE;
    int steps = 0;
    // a failed factorisation (this instance's own model, or the one shared model: the flag a factor call with no_sync
    // leaves on the device) is reported as status MPCQP_MAX_ITER, never as a solution
    const bool mfail = (p.mflag != nullptr) && p.mflag[p.st_nn ? inst : 0] != 0;
    bool overflow = false, fail = mfail;

    if (p.polish && !fail) {
        // ---- warm phase: solve on the guessed set, release wrong-signed multipliers -------------
        int k = 0;
        for (;;) {
            k = build_list(s, side, lane);
            if (k > KMAX) { overflow = true; break; }
            lam[0] = lam[1] = 0.0;
            if (k == 0) { x[0] = xu[0]; x[1] = xu[1]; break; }
            if (!chol_S(s, Hi, n, k, lane)) { fail = true; break; }
            // rhs_i = xu[A_i] - bnd_i, in list order (rows lane, lane+64 of the LIST)
            s.xs[lane] = xu[0]; s.xs[lane + 64] = xu[1];
            wave_sync();
            double v0 = 0.0, v1 = 0.0;
            {
                if (lane < k) { const int a = s.A[lane], r = abs(a) - 1; v0 = s.xs[r] - (a > 0 ? p.ubs[r] : p.lbs[r]); }
                if (lane + 64 < k) { const int a = s.A[lane + 64], r = abs(a) - 1; v1 = s.xs[r] - (a > 0 ? p.ubs[r] : p.lbs[r]); }
            }
            fwd_solve(s, k, lane, v0, v1);
            bwd_solve(s, k, lane, v0, v1);
            // multipliers live in list order in (v0, v1); scatter to row order through LDS
            s.gs[lane] = 0.0; s.gs[lane + 64] = 0.0;
            wave_sync();
            bool neg0 = false, neg1 = false;
            if (lane < k) { const int a = s.A[lane]; s.gs[abs(a) - 1] = v0; neg0 = (a > 0 ? v0 : -v0) < 0.0; }
            if (lane + 64 < k) { const int a = s.A[lane + 64]; s.gs[abs(a) - 1] = v1; neg1 = (a > 0 ? v1 : -v1) < 0.0; }
            wave_sync();
            lam[0] = s.gs[lane]; lam[1] = s.gs[lane + 64];
            const bool anyneg = __any(neg0 || neg1);
            if (!anyneg) {
                // x = xu - Hi[:,A] lam_A, exact bounds on A
                double acc0 = xu[0], acc1 = xu[1];
                rows_accum(s, Hi, n, k, lane, v0, v1, acc0, acc1);
                x[0] = side[0] > 0 ? hi[0] : (side[0] < 0 ? lo[0] : acc0);
                x[1] = side[1] > 0 ? hi[1] : (side[1] < 0 ? lo[1] : acc1);
                break;
            }
            // release: row-order test on lam sign
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const bool rel = (side[q] != 0 && lam[q] * side[q] < 0.0);
                if (rel) side[q] = 0;
                steps += __popcll(__ballot(rel));
            }
            wave_sync();
        }

        // ---- main phase: add the most violated bound until none is violated ---------------------
        int guard = 0;
        while (!overflow && !fail) {
            double vb[2];
            int best_row = -1; double best = 0.0;
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int j = lane + 64 * q;
                vb[q] = -1.0;
                if (j < n && side[q] == 0) {
                    const double tol = 1e-12 * bound_scale(lo[q], hi[q]);
                    const double v = fmax(x[q] - hi[q], lo[q] - x[q]);
                    if (v > tol) vb[q] = v;
                }
                if (vb[q] > best) { best = vb[q]; best_row = j; }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ob = __shfl_xor(best, off, 64);
                const int orow = __shfl_xor(best_row, off, 64);
                if (ob > best || (ob == best && orow >= 0 && (best_row < 0 || orow < best_row))) { best = ob; best_row = orow; }
            }
            if (best_row < 0) break;                       // primal feasible: optimal
            if (++guard > p.max_steps) { fail = true; break; }
            const int pr = best_row;                       // wave-uniform
            const double xp = row_get(x[0], x[1], pr);
            const double hip_ = p.ubs[pr], lop_ = p.lbs[pr];
            const int sp = (xp > hip_) ? 1 : -1;
            double v0viol = (sp > 0) ? xp - hip_ : lop_ - xp;
            double accl = 0.0;                              // multiplier of p accumulated over partial steps
            for (;;) {
                // l = L^-1 Hi[A,p]  (list order), d2 = Hi[p,p] - l'l, r = sp L^-T l
                double l0 = 0.0, l1 = 0.0;
                if (lane < k) l0 = Hi[(size_t)(abs(s.A[lane]) - 1) * n + pr];
                if (lane + 64 < k) l1 = Hi[(size_t)(abs(s.A[lane + 64]) - 1) * n + pr];
                fwd_solve(s, k, lane, l0, l1);
                double ss = (lane < k ? l0 * l0 : 0.0) + (lane + 64 < k ? l1 * l1 : 0.0);
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) ss += __shfl_xor(ss, off, 64);
                const double d2 = Hi[(size_t)pr * n + pr] - ss;
                if (!(d2 > 0.0)) { fail = true; break; }
                double r0 = l0, r1 = l1;
                bwd_solve(s, k, lane, r0, r1);
                r0 *= sp; r1 *= sp;
                // scatter r to row order
                s.gs[lane] = 0.0; s.gs[lane + 64] = 0.0;
                wave_sync();
                if (lane < k) s.gs[abs(s.A[lane]) - 1] = r0;
                if (lane + 64 < k) s.gs[abs(s.A[lane + 64]) - 1] = r1;
                wave_sync();
                const double rr[2] = {s.gs[lane], s.gs[lane + 64]};
                // step lengths
                const double t2 = v0viol / d2;
                double t1 = 1e300; int jb = -1;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (side[q] != 0 && side[q] * rr[q] > 0.0) {
                        const double t = lam[q] / rr[q];
                        if (t < t1) { t1 = t; jb = lane + 64 * q; }
                    }
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) {
                    const double ot = __shfl_xor(t1, off, 64);
                    const int oj = __shfl_xor(jb, off, 64);
                    if (ot < t1 || (ot == t1 && oj >= 0 && (jb < 0 || oj < jb))) { t1 = ot; jb = oj; }
                }
                const bool full = (t2 <= t1);
                const double t = full ? t2 : t1;
                // primal direction z_j = sp Hi[p][j] - sum_a Hi[A_a][j] r_a
                {
                    double z0 = lane < n ? sp * Hi[(size_t)pr * n + lane] : 0.0;
                    double z1 = lane + 64 < n ? sp * Hi[(size_t)pr * n + lane + 64] : 0.0;
                    rows_accum(s, Hi, n, k, lane, r0, r1, z0, z1);
                    if (side[0] == 0) x[0] -= t * z0;
                    if (side[1] == 0) x[1] -= t * z1;
                    lam[0] -= t * rr[0];
                    lam[1] -= t * rr[1];
                }
                accl += sp * t;
                ++steps;
                if (full) {
                    if (k + 1 > KMAX) { overflow = true; break; }
                    // append row k of the factor: [l, sqrt(d2)]
                    if (lane < k) Lat(s, k, lane) = l0;
                    if (lane + 64 < k) Lat(s, k, lane + 64) = l1;
                    if (lane == 0) { Lat(s, k, k) = sqrt(d2); s.di[k] = 1.0 / sqrt(d2); s.A[k] = sp * (pr + 1); }
                    ++k;
#pragma unroll
                    for (int q = 0; q < 2; ++q)
                        if (lane + 64 * q == pr) { side[q] = sp; lam[q] = accl; x[q] = sp > 0 ? hi[q] : lo[q]; }
                    wave_sync();
                    break;
                }
                // partial step: release row jb and refactor
                v0viol -= t * d2;
#pragma unroll
                for (int q = 0; q < 2; ++q)
                    if (lane + 64 * q == jb) { side[q] = 0; lam[q] = 0.0; }
                wave_sync();
                k = build_list(s, side, lane);
                if (k > 0 && !chol_S(s, Hi, n, k, lane)) { fail = true; break; }
                if (++guard > p.max_steps) { fail = true; break; }
            }
        }

        // ---- refinement with Hs itself --------------------------------------------------------
        if (!overflow && !fail) {
            for (int itr = 0; itr < p.refine; ++itr) {
                s.xs[lane] = x[0]; s.xs[lane + 64] = x[1];
                wave_sync();
                double g[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = lane + 64 * q;
                    g[q] = 0.0;
                    if (j < n) {
                        double acc = qs[q];
                        for (int m = 0; m < n; ++m) acc += p.Hs[(size_t)m * n + j] * s.xs[m];
                        if (side[q] == 0) g[q] = acc;   // stationarity residual of a free row
                        else lam[q] = -acc;             // the multiplier this x really has
                    }
                }
                s.gs[lane] = g[0]; s.gs[lane + 64] = g[1];
                wave_sync();
                double dx[2];
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = lane + 64 * q;
                    dx[q] = 0.0;
                    if (j < n) {
                        double acc = 0.0;
                        for (int m = 0; m < n; ++m) acc -= Hi[(size_t)m * n + j] * s.gs[m];
                        dx[q] = acc;
                    }
                }
                if (k > 0) {
                    wave_sync();
                    s.gs[lane] = dx[0]; s.gs[lane + 64] = dx[1];
                    wave_sync();
                    double v0 = 0.0, v1 = 0.0;
                    if (lane < k) v0 = s.gs[abs(s.A[lane]) - 1];
                    if (lane + 64 < k) v1 = s.gs[abs(s.A[lane + 64]) - 1];
                    fwd_solve(s, k, lane, v0, v1);
                    bwd_solve(s, k, lane, v0, v1);
                    wave_sync();
                    s.gs[lane] = 0.0; s.gs[lane + 64] = 0.0;
                    wave_sync();
                    if (lane < k) s.gs[abs(s.A[lane]) - 1] = v0;
                    if (lane + 64 < k) s.gs[abs(s.A[lane + 64]) - 1] = v1;
                    wave_sync();
                    {
                        double a0 = dx[0], a1 = dx[1];
                        rows_accum(s, Hi, n, k, lane, v0, v1, a0, a1);
                        if (side[0] == 0) x[0] += a0; else lam[0] += s.gs[lane];
                        if (side[1] == 0) x[1] += a1; else lam[1] += s.gs[lane + 64];
                    }
                } else {
                    x[0] += dx[0]; x[1] += dx[1];
                }
                wave_sync();
            }
            status = MPCQP_SOLVED;
            if (p.refine > 0) {
                // The refinement moves the free rows with Hs itself; nothing has checked the moved point yet.  When the
                // explicit inverse is as poor as a condition number of 1e13 makes it, the corrected rows can leave the box
                // or the corrected multipliers change sign: such an answer is not SOLVED.  Overshoots at rounding level are
                // clipped, so that U is always inside the bounds.
                bool off = false;
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    const int j = lane + 64 * q;
                    if (j < n) {
                        if (side[q] == 0) {
                            const double tol = 1e-9 * bound_scale(lo[q], hi[q]);
                            off = off || x[q] > hi[q] + tol || x[q] < lo[q] - tol;
                            x[q] = fmin(fmax(x[q], lo[q]), hi[q]);
                        } else {
                            off = off || lam[q] * side[q] < -1e-9 * fmax(1.0, fabs(qs[q]));
                        }
                    }
                }
                if (__any(off)) status = MPCQP_SOLVED_INACCURATE;
            }
        }
    }

    if (overflow && p.qout) {               // queue mode: hand the instance to the next, larger pass
        if (lane == 0) p.qout[atomicAdd(p.qout_count, 1)] = inst;
        return;
    }
    if (overflow && p.pass == 0) {          // redo in the big pass
        if (lane == 0) p.need_big[inst] = 1;
        return;
    }
    if (p.pass == 0 && lane == 0) p.need_big[inst] = 0;

    if (!p.polish || fail || overflow) {
        // ADMM answer as is: x = z (feasible), lam = rho w
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < n) { x[q] = fmin(fmax(zb[j], lo[q]), hi[q]); lam[q] = p.rho * wb[j]; }
        }
        status = (p.admm_ran && p.cv_ws[inst] && !mfail) ? MPCQP_SOLVED_INACCURATE : MPCQP_MAX_ITER;
    }

    // ---- objective 1/2 x'Hs x + qs'x ---------------------------------------------------------------
    double obj;
    if (status == MPCQP_SOLVED) {
        // stationarity Hs x + qs + lam = 0  =>  obj = 1/2 (qs - lam)'x
        obj = 0.5 * ((qs[0] - lam[0]) * x[0] + (qs[1] - lam[1]) * x[1]);
    } else {
        s.xs[lane] = x[0]; s.xs[lane + 64] = x[1];
        wave_sync();
        obj = 0.0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int j = lane + 64 * q;
            if (j < n) {
                double acc = 0.0;
                for (int m = 0; m < n; ++m) acc += p.Hs[(size_t)m * n + j] * s.xs[m];
                obj += x[q] * (0.5 * acc + qs[q]);
            }
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) obj += __shfl_xor(obj, off, 64);

    // ---- outputs ------------------------------------------------------------------------------------
#pragma unroll
    for (int q = 0; q < 2; ++q) {
        const int j = lane + 64 * q;
        if (j < n) {
            const double dj = p.d[j];
            p.U[(size_t)inst * n + j] = dj * x[q];
            if (p.info.lam) p.info.lam[(size_t)inst * n + j] = lam[q] / dj;
            if (status == MPCQP_SOLVED && p.keep_state) {       // keep the polished point as warm-start state
                zb[j] = x[q];
                wb[j] = lam[q] / p.rho;
            }
        }
    }
    if (lane == 0) {
        if (p.info.status) p.info.status[inst] = status;
        if (p.info.obj) p.info.obj[inst] = obj;
        if (p.info.iters) p.info.iters[inst] = p.admm_ran ? p.it_ws[inst] : 0;
        if (p.info.prim_res) p.info.prim_res[inst] = p.admm_ran ? p.rp_ws[inst] : 0.0;
        if (p.info.dual_res) p.info.dual_res[inst] = p.admm_ran ? p.rd_ws[inst] : 0.0;
        if (p.info.polish_steps) p.info.polish_steps[inst] = steps;
    }
}

template <int KMAX, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void polish_box_kernel(PolParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    WaveLds<KMAX>& s = reinterpret_cast<WaveLds<KMAX>*>(smem)[wave];
    if (p.qin) {            // leftovers of the fast path: grid-stride over the device-side queue
        const int cnt = *p.qin_count;
        for (int qi = blockIdx.x * WAVES + wave; qi < cnt; qi += gridDim.x * WAVES) {
            polish_wave<KMAX>(p, p.qin[qi], s, lane);
            wave_sync();
        }
        return;
    }
    const int inst = blockIdx.x * WAVES + wave;
    if (inst >= p.B) return;
    if (p.pass == 1 && p.need_big[inst] == 0) return;
    polish_wave<KMAX>(p, inst, s, lane);
}

// Second pass on a short grid (a kernel of its own: one more way into polish_box_kernel cost its first-pass instantiation a
// VGPR and with it a wave per SIMD): every wave looks through its share of the flags the first pass left, 64 per load, and takes
// the instances it finds one after the other -- usually none, and then the launch is a few loads long instead of one wave per
// instance that starts, reads one flag and ends
template <int KMAX>
__global__ __launch_bounds__(64) void polish_scan_kernel(PolParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    WaveLds<KMAX>& s = reinterpret_cast<WaveLds<KMAX>*>(smem)[0];
    const int nw = (int)gridDim.x, per = ((p.B + nw - 1) / nw + 63) & ~63;
    const int lo = blockIdx.x * per, hi = min(lo + per, p.B);
    for (int base = lo; base < hi; base += 64) {
        unsigned long long m = __ballot(base + lane < hi && p.need_big[base + lane] != 0);
        while (m) {
            const int j = __builtin_ctzll(m);
            m &= m - 1;
            polish_wave<KMAX>(p, base + j, s, lane);
            wave_sync();
        }
    }
}

template <int KMAX>
int launch_pol_scan(const PolParams& p, hipStream_t st) {
    const size_t lds = sizeof(WaveLds<KMAX>);
    hipError_t e = ensure_dyn_lds(p.dev, reinterpret_cast<const void*>(&polish_scan_kernel<KMAX>), lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((polish_scan_kernel<KMAX>), dim3(std::min(ceil_div(p.B, 64), 1024)), dim3(64), lds, st, p);
    return (int)hipGetLastError();
}

template <int KMAX, int WAVES>
int launch_pol(const PolParams& p, hipStream_t st, int blocks = 0) {
    const size_t lds = sizeof(WaveLds<KMAX>) * WAVES;
    hipError_t e = ensure_dyn_lds(p.dev, reinterpret_cast<const void*>(&polish_box_kernel<KMAX, WAVES>), lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((polish_box_kernel<KMAX, WAVES>), dim3(blocks > 0 ? blocks : ceil_div(p.B, WAVES)), dim3(64 * WAVES), lds, st, p);
    return (int)hipGetLastError();
}

}  // namespace

// After the passes of a component-split solve (split.hip), ONE launch of one-wave workgroups with two kinds of work:
//   * workgroups [0, redo_blocks): the instances the last pass of the cascade queued (a component with more active rows than
//     that pass holds, or a numerical guard) are solved again as a whole, from the empty set, by the any-size method
//     (polish_wave, one wave per instance; an instance queued by several of its components is taken once);
//   * the others, 64 instances each: one lane per instance adds up its components' parts of the objective and their step
//     counts, in component order, and writes the instance's record -- unless a component is unfinished (need != 0): that
//     instance's record is the redo's.  The first of them also zeroes the counters for the next solve (no memset in front of
//     every solve): all but the redo queue's, which the other kind reads -- split_key_kernel clears that one.
// The two do not depend on each other (the queue was written by the launch before), so they share a launch: the redo is
// nearly always empty, and as a launch of its own it cost a small batch 4 us of its 74.
// p is the UNSPLIT parameter set plus sp_C / sp_part / sp_steps, qin / qin_count = the redo queue.
__global__ __launch_bounds__(64) void split_tail_kernel(PolParams p, const int* need, const double* c0, int* zero, int nzero,
                                                        int redo_blocks) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int lane = threadIdx.x;
    if ((int)blockIdx.x < redo_blocks) {
        WaveLds<128>& s = reinterpret_cast<WaveLds<128>*>(smem)[0];
        const int cnt = *p.qin_count;
        for (int qi = blockIdx.x; qi < cnt; qi += redo_blocks) {
            const int inst = p.qin[qi];
            bool seen = false;                          // queued before by another of its components?
            for (int j = lane; j < qi; j += 64) seen = seen || p.qin[j] == inst;
            if (__any(seen)) continue;
            polish_wave<128>(p, inst, s, lane);
            wave_sync();
        }
        return;
    }
    const int fb = blockIdx.x - redo_blocks;
    if (fb == 0) for (int i = lane; i < nzero; i += 64) if (zero + i != p.qin_count) zero[i] = 0;
    const int inst = fb * 64 + lane, C = p.sp_C;
    if (inst >= p.B) return;
    constexpr int CM = 8;
    double pv[CM]; int nv[CM], sv[CM];
    double o = c0[inst];        // the unconstrained part of the objective, 1/2 [x0;1]' Cq [x0;1] (split_key_kernel)
#pragma unroll
    for (int c = 0; c < CM; ++c) {      // every load of the lane first: they are independent, and each is a trip to memory
        const bool in = c < C;
        nv[c] = in ? need[(size_t)inst * C + c] : 0;
        pv[c] = in ? p.sp_part[(size_t)inst * C + c] : 0.0;
        sv[c] = in ? p.sp_steps[(size_t)inst * C + c] : 0;
    }
    bool ok = true; int steps = 0;
#pragma unroll
    for (int c = 0; c < CM; ++c) { ok = ok && nv[c] == 0; o += pv[c]; steps += sv[c]; }
    if (!ok) return;            // in the redo queue (the last pass put it there)
    if (p.info.status) p.info.status[inst] = MPCQP_SOLVED;
    if (p.info.obj) p.info.obj[inst] = o;
    if (p.info.iters) p.info.iters[inst] = 0;
    if (p.info.prim_res) p.info.prim_res[inst] = 0.0;
    if (p.info.dual_res) p.info.dual_res[inst] = 0.0;
    if (p.info.polish_steps) p.info.polish_steps[inst] = steps;
}

int launch_split_finish(const PolParams& pin, const int* need, const double* c0, int* redo, int* redo_count, int* zero, int nzero,
                        hipStream_t st) {
    PolParams p = pin;
    p.qin = redo; p.qin_count = redo_count;
    constexpr int REDO_BLOCKS = 64;     // a short grid: the queue is empty unless a component outgrew the cascade or hit a numerical guard
    const size_t lds = sizeof(WaveLds<128>);
    hipError_t e = ensure_dyn_lds(p.dev, reinterpret_cast<const void*>(&split_tail_kernel), lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL(split_tail_kernel, dim3(REDO_BLOCKS + ceil_div(pin.B, 64)), dim3(64), lds, st, p, need, c0, zero, nzero, REDO_BLOCKS);
    return (int)hipGetLastError();
}

int launch_polish_box(mpcqp_handle h, const double* x0, int B, double* U, const mpcqp_info* info,
                      const mpcqp_solve_opts& o, hipStream_t st) {
    PolParams p{};
    p.Hs = h->Hs; p.Hi = h->Hi; p.FsT = h->FsT; p.KuT = h->KuT; p.lbs = h->lbs; p.ubs = h->ubs; p.d = h->d;
    p.x0 = x0; p.z = h->z; p.w = h->w; p.it_ws = h->it_ws; p.rp_ws = h->rp_ws; p.rd_ws = h->rd_ws;
    p.need_big = h->it_ws + h->Bcap;      // int workspace: [iters | need_big | converged]
    p.cv_ws = h->it_ws + 2 * (size_t)h->Bcap;
    p.U = U;
    if (info) p.info = *info; else p.info = mpcqp_info{nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
    p.B = B; p.n = h->n; p.nx = h->dims.nx; p.NP = h->NP; p.dev = h->dev;
    p.polish = o.polish;
    // refine < 0: automatic -- one step with Hs itself only if the explicit inverse is not accurate enough
    p.refine = o.polish_refine >= 0 ? o.polish_refine : (h->inv_resid > 1e-9 ? 1 : 0); p.max_steps = o.polish_max_steps > 0 ? o.polish_max_steps : 4 * h->n + 16;
    p.rho = h->rho;
    p.keep_state = o.keep_state ? 1 : 0;
    p.prof = nullptr;
#ifdef POLISH_PROF
    static unsigned long long* profbuf = nullptr;
    if (!profbuf) { (void)hipMalloc(&profbuf, 32 * 8); }
    (void)hipMemsetAsync(profbuf, 0, 32 * 8, st);
    p.prof = profbuf;
#endif
    p.pass = 0;
    const bool per_model = h->dims.n_models > 1;
    p.st_nn = per_model ? (long)h->n * h->n : 0; p.st_k = per_model ? (long)h->NP * (h->dims.nx + 1) : 0;
    p.st_np = per_model ? h->NP : 0; p.st_n = per_model ? h->n : 0;
    // cold: no ADMM iterate to start from (one model per instance, or max_iter = 0 without a kept state): the kernels
    // start from the empty active set
    const bool no_admm = o.max_iter == 0 && o.polish && (o.stages & 1);     // api.hip skipped the ADMM launch
    p.cold = ((per_model || no_admm) && !(o.warm_start && h->has_state)) ? 1 : 0;
    p.admm_ran = (!per_model && !no_admm) ? 1 : 0;      // iters / residuals / converged flags are the ADMM kernel's
    p.Cq = per_model ? nullptr : h->Cq;
    p.mflag = per_model ? h->mflag : h->flag_dev;
    p.qin = nullptr; p.qin_count = nullptr; p.qout = nullptr; p.qout_count = nullptr;
    p.hand_in = nullptr; p.hand_out = nullptr; p.hand_out_cap = 0; p.hand_zw = 0;
    p.hand_buf1 = h->hbuf1; p.hand_buf2 = h->hbuf2; p.hand_cap2 = h->Bcap / 4;
    if (h->dims.nx > 64) { h->err = "polish: nx > 64 not supported in this build"; return MPCQP_ENOTIMPL; }
    // int workspace beyond [iters | need_big | converged]: two instance queues and their counters
    int* q1 = h->it_ws + 3 * (size_t)h->Bcap;
    int* q2 = q1 + h->Bcap;
    int* counters = q2 + h->Bcap;
    // small active sets (the usual case): a few lanes per instance with Hs^-1 in LDS, a cascade of passes (polish_quad.hip);
    // what those cannot finish is queued in q2 for the wave-per-instance kernel.  Shapes the fast path does not
    // take (refinement requested, polish off, n > 128) run the wave-per-instance kernel on the whole batch.
    // diagnostics for bench.py: stages bit 2 stops after the first pass, bit 3 after the cold-start ordering kernels
    const int stop_after = (o.stages & 8) ? 2 : ((o.stages & 4) ? 1 : 0);
    // a model whose Hessian falls into independent blocks: the cold solve runs per block (split.hip); -1 = does not apply
    int rc = per_model ? -1 : launch_polish_split(h, p, st, stop_after);
#ifdef POLISH_PROF
    if (rc == 0 && getenv("MPCQP_DEBUG_POLISH")) {
        unsigned long long pr[32];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(pr, profbuf, sizeof(pr), hipMemcpyDeviceToHost);
        const char* nm[16] = {"stage", "xu", "warm-build", "warm-lam", "xeval", "refine", "out", "#build", "#inner", "scan", "direction",
                              "x-update", "append", "drop", "#drops", ""};
        for (int k = 0; k < 2; ++k) {
            fprintf(stderr, "[split prof] %s-lane passes, cycle sums over all waves:", k ? "16" : "8");
            for (int i = 0; i < 15; ++i) fprintf(stderr, " %s %llu", nm[i], pr[16 * k + i]);
            fprintf(stderr, " | longest wave %llu cycles\n", pr[16 * k + 15]);
        }
    }
#endif
    if (rc == 0) return MPCQP_OK;
    if (rc < 0 && !per_model) {
        if (p.cold && p.polish && p.refine == 0 && h->n <= 128) {       // the cascade hands working states from pass to pass
            const int ra = ensure_handover(h);
            if (ra != MPCQP_OK) return ra;
            p.hand_buf1 = h->hbuf1; p.hand_buf2 = h->hbuf2;
        }
        rc = launch_polish_quad(p, q1, q2, counters, st, stop_after);
    }
    if (rc == 0 && (o.stages & 12)) return MPCQP_OK;
#ifdef POLISH_PROF
    if (rc == 0 && getenv("MPCQP_DEBUG_POLISH")) {
        unsigned long long pr[32]; int cn[4];
        (void)hipStreamSynchronize(st);
        (void)hipMemcpy(pr, profbuf, sizeof(pr), hipMemcpyDeviceToHost);
        (void)hipMemcpy(cn, counters, sizeof(cn), hipMemcpyDeviceToHost);
        const char* nm[16] = {"stage", "xu", "warm-build", "warm-lam", "xeval", "refine", "out", "#build", "#inner", "scan", "direction",
                              "x-update", "append", "drop", "#drops", ""};
        for (int k = 0; k < 2; ++k) {
            const double nw = k == 0 ? (double)ceil_div(cn[2] > 0 ? cn[2] : B, 8) : (double)ceil_div(cn[0], 2);
            fprintf(stderr, "[polish prof] pass %c: %d instances, %.0f waves; per wave (shader cycles):", k ? 'B' : 'A', k ? cn[0] : (cn[2] > 0 ? cn[2] : B), nw);
            for (int i = 0; i < 15; ++i) fprintf(stderr, " %s %.1f", nm[i], pr[16 * k + i] / nw);
            fprintf(stderr, " | longest wave %llu cycles", pr[16 * k + 15]);
            fprintf(stderr, "\n");
        }
    }
#endif
    if (rc == 0) {
        // what the two fast passes could not finish: up to 64 active rows with four waves per workgroup (eight waves per
        // CU), the rest (queue 1 is free again) with one wave per workgroup and room for every row
        p.pass = 1; p.qin = q2; p.qin_count = counters + 1; p.qout = q1; p.qout_count = counters + 3;
        rc = launch_polish_oct64(p, st);
        if (rc < 0) rc = launch_pol<64, 4>(p, st, 512);
        if (rc == 0) {
            p.qin = q1; p.qin_count = counters + 3; p.qout = nullptr; p.qout_count = nullptr;
            p.hand_zw = p.cold;         // the 64-row pass leaves its working set as the guess
            rc = launch_pol<128, 1>(p, st, 512);
        }
    } else if (rc < 0) {
        rc = launch_pol<32, 4>(p, st);
        if (rc == 0 && o.polish) { p.pass = 1; rc = launch_pol_scan<128>(p, st); }
    }
    if (rc != 0) { h->err = std::string("polish launch: ") + hipGetErrorString((hipError_t)rc); return MPCQP_EHIP; }
    return MPCQP_OK;
}
