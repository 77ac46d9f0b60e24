/* qp_ipm.c -- CPU restatement in C of the reference's solver algorithm class,
 * TEST/BASELINE INFRASTRUCTURE ONLY: only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load the library built from this file.
 *
 * The reference solves each QP with cvxopt.solvers.qp(P, q, G, h)
 * (LaunchPlannerQPFormulation.py:173, Launch3DPlanner.py:201, SE3Planner.py:183):
 * a dense primal-dual interior-point method (Mehrotra predictor-corrector,
 * one dense Cholesky of P + G' diag(z/s) G per iteration), one problem at a
 * time.  cvxopt is absent from this image, so this file restates that
 * algorithm (same structure as oracle/qp_oracle.py::solve_qp, inequality-only
 * case) with cvxopt's default stopping tolerances, in plain C so that it can
 * be timed on the GPU box's host cores: one QP per thread, OpenMP over
 * instances.  It is a "port" baseline, not cvxopt itself (no BLAS here; dense
 * G is treated as dense, as cvxopt does).
 *
 * Build: see oracle/Makefile  (gcc -O3 -march=native -fopenmp -shared -fPIC).
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* in-place lower Cholesky of n x n row-major A; returns 0 on success */
static int chol(double* A, int n) {
    for (int j = 0; j < n; ++j) {
        double d = A[j * n + j];
        for (int k = 0; k < j; ++k) d -= A[j * n + k] * A[j * n + k];
        if (!(d > 0.0)) return 1;
        d = sqrt(d);
        A[j * n + j] = d;
        const double inv = 1.0 / d;
        for (int i = j + 1; i < n; ++i) {
            double s = A[i * n + j];
            const double* ai = A + i * n; const double* aj = A + j * n;
            for (int k = 0; k < j; ++k) s -= ai[k] * aj[k];
            A[i * n + j] = s * inv;
        }
    }
    return 0;
}

static void chol_solve(const double* L, int n, double* b) {
    for (int i = 0; i < n; ++i) {
        double s = b[i];
        for (int k = 0; k < i; ++k) s -= L[i * n + k] * b[k];
        b[i] = s / L[i * n + i];
    }
    for (int i = n - 1; i >= 0; --i) {
        double s = b[i];
        for (int k = i + 1; k < n; ++k) s -= L[k * n + i] * b[k];
        b[i] = s / L[i * n + i];
    }
}

static double max_step(const double* v, const double* dv, int m) {
    double a = 1e300;
    for (int i = 0; i < m; ++i)
        if (dv[i] < 0.0) { const double t = -v[i] / dv[i]; if (t < a) a = t; }
    return a;
}

/* workspace size in doubles for one problem */
long qp_ipm_worksize(int n, int m) { return (long)2 * n * n + (long)m * n + 12L * m + 8L * n + 64; }

/* Solve min 1/2 x'Px + q'x s.t. Gx <= h.  P (n,n) symmetric PSD, G (m,n), row-major.
 * Returns 0 optimal, 1 max iterations, 2 numerical failure. */
int qp_ipm_solve(int n, int m, const double* P, const double* q, const double* G, const double* h,
                 double* x, double* z, int maxiters, double abstol, double reltol, double feastol,
                 int* iters_out, double* work) {
    double* Hd = work;                 /* n*n */
    double* GD = Hd + (long)n * n;     /* m*n: diag(d) G */
    double* s = GD + (long)m * n;
    double* d = s + m;   double* rz = d + m;  double* t = rz + m;  double* dz = t + m;  double* ds = dz + m;
    double* dza = ds + m; double* dsa = dza + m; double* rc = dsa + m; double* Gx = rc + m; double* tmpm = Gx + m;
    double* rx = tmpm + m; double* dx = rx + n; double* rhs = dx + n; double* Px = rhs + n;
    int it, status = 1;

    /* initial point: x solves (P + G'G) x = -q + G'h ; s, z from the residual (cvxopt-style) */
    for (int i = 0; i < n; ++i)
        for (int j = 0; j <= i; ++j) {
            double a = P[i * n + j];
            for (int k = 0; k < m; ++k) a += G[k * n + i] * G[k * n + j];
            Hd[i * n + j] = a;
        }
    if (chol(Hd, n)) return 2;
    for (int i = 0; i < n; ++i) { double a = -q[i]; for (int k = 0; k < m; ++k) a += G[k * n + i] * h[k]; x[i] = a; }
    chol_solve(Hd, n, x);
    {
        double amax = -1e300, amin = 1e300, nrm = 0.0;
        for (int k = 0; k < m; ++k) {
            double a = -h[k]; for (int j = 0; j < n; ++j) a += G[k * n + j] * x[j];
            tmpm[k] = a; if (a > amax) amax = a; if (a < amin) amin = a; nrm += a * a;
        }
        nrm = sqrt(nrm); if (nrm < 1.0) nrm = 1.0;
        for (int k = 0; k < m; ++k) { s[k] = -tmpm[k]; z[k] = tmpm[k]; }
        if (amax >= -1e-8 * nrm) for (int k = 0; k < m; ++k) s[k] += 1.0 + amax;
        if (-amin >= -1e-8 * nrm) for (int k = 0; k < m; ++k) z[k] += 1.0 - amin;
    }
    double nq = 0.0, nh = 0.0;
    for (int i = 0; i < n; ++i) nq += q[i] * q[i];
    for (int k = 0; k < m; ++k) nh += h[k] * h[k];
    nq = fmax(1.0, sqrt(nq)); nh = fmax(1.0, sqrt(nh));

    for (it = 0; it < maxiters; ++it) {
        /* residuals */
        double pobj = 0.0, gap = 0.0, nrx = 0.0, nrz = 0.0, zr = 0.0;
        for (int i = 0; i < n; ++i) { double a = 0.0; for (int j = 0; j < n; ++j) a += P[i * n + j] * x[j]; Px[i] = a; }
        for (int i = 0; i < n; ++i) { rx[i] = Px[i] + q[i]; pobj += x[i] * (0.5 * Px[i] + q[i]); }
        for (int k = 0; k < m; ++k) {
            double a = 0.0; const double* g = G + (long)k * n;
            for (int j = 0; j < n; ++j) a += g[j] * x[j];
            Gx[k] = a; rz[k] = a + s[k] - h[k]; gap += s[k] * z[k]; nrz += rz[k] * rz[k]; zr += z[k] * (a - h[k]);
            const double zk = z[k];
            for (int j = 0; j < n; ++j) rx[j] += g[j] * zk;
        }
        for (int i = 0; i < n; ++i) nrx += rx[i] * rx[i];
        const double dobj = pobj + zr;
        const double pres = sqrt(nrz) / nh, dres = sqrt(nrx) / nq;
        double relgap = 1e300;
        if (pobj < 0.0) relgap = gap / -pobj; else if (dobj > 0.0) relgap = gap / dobj;
        if (pres <= feastol && dres <= feastol && (gap <= abstol || relgap <= reltol)) { status = 0; break; }

        /* Hd = P + G' diag(z/s) G, Cholesky */
        for (int k = 0; k < m; ++k) {
            d[k] = z[k] / s[k];
            const double* g = G + (long)k * n; double* gd = GD + (long)k * n;
            for (int j = 0; j < n; ++j) gd[j] = d[k] * g[j];
        }
        for (int i = 0; i < n; ++i) for (int j = 0; j <= i; ++j) Hd[i * n + j] = P[i * n + j];
        for (int k = 0; k < m; ++k) {
            const double* g = G + (long)k * n; const double* gd = GD + (long)k * n;
            for (int i = 0; i < n; ++i) {
                const double gi = gd[i];
                if (gi == 0.0) continue;        /* sparsity of a row costs nothing to skip */
                double* hrow = Hd + (long)i * n;
                for (int j = 0; j <= i; ++j) hrow[j] += gi * g[j];
            }
        }
        if (chol(Hd, n)) return 2;

        /* predictor: rc = s*z, eta = 1 */
        double mu = gap / m, sigma = 0.0, eta = 1.0;
        for (int pass = 0; pass < 2; ++pass) {
            for (int k = 0; k < m; ++k) {
                const double rck = pass == 0 ? s[k] * z[k] : s[k] * z[k] + dsa[k] * dza[k] - sigma * mu;
                rc[k] = rck;
                t[k] = (-rck + eta * z[k] * rz[k]) / s[k];
            }
            for (int i = 0; i < n; ++i) rhs[i] = -eta * rx[i];
            for (int k = 0; k < m; ++k) {
                const double* g = G + (long)k * n; const double tk = t[k];
                for (int j = 0; j < n; ++j) rhs[j] -= g[j] * tk;
            }
            memcpy(dx, rhs, sizeof(double) * n);
            chol_solve(Hd, n, dx);
            for (int k = 0; k < m; ++k) {
                double a = 0.0; const double* g = G + (long)k * n;
                for (int j = 0; j < n; ++j) a += g[j] * dx[j];
                dz[k] = t[k] + d[k] * a;
                ds[k] = -eta * rz[k] - a;
            }
            if (pass == 0) {
                double a = fmin(1.0, fmin(max_step(s, ds, m), max_step(z, dz, m)));
                sigma = (1.0 - a) * (1.0 - a) * (1.0 - a);
                eta = 1.0 - sigma;
                memcpy(dsa, ds, sizeof(double) * m); memcpy(dza, dz, sizeof(double) * m);
            }
        }
        const double a = fmin(1.0, 0.99 * fmin(max_step(s, ds, m), max_step(z, dz, m)));
        for (int i = 0; i < n; ++i) x[i] += a * dx[i];
        for (int k = 0; k < m; ++k) { s[k] += a * ds[k]; z[k] += a * dz[k]; }
    }
    if (iters_out) *iters_out = it;
    return status;
}

/* Batched condensed-MPC box QPs sharing H:  q_b = F x0_b + f0,  lb <= u <= ub, written in the
 * reference's one-sided form G = [I; -I], h = [ub; -lb] (LaunchPlannerQPFormulation.py:96-107).
 * One instance per thread.  U (batch, n).  Returns the number of instances not reported optimal. */
int qp_ipm_batch_box(int n, int nx, int batch, const double* H, const double* F, const double* f0,
                     const double* lb, const double* ub, const double* x0, double* U, int* iters,
                     int nthreads, double abstol, double reltol, double feastol) {
    const int m = 2 * n;
    double* G = (double*)calloc((size_t)m * n, sizeof(double));
    double* h = (double*)malloc(sizeof(double) * m);
    for (int i = 0; i < n; ++i) { G[i * n + i] = 1.0; h[i] = ub[i]; G[(n + i) * n + i] = -1.0; h[n + i] = -lb[i]; }
    int bad = 0;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#endif
#pragma omp parallel reduction(+ : bad)
    {
        double* work = (double*)malloc(sizeof(double) * qp_ipm_worksize(n, m));
        double* q = (double*)malloc(sizeof(double) * n);
        double* z = (double*)malloc(sizeof(double) * m);
#pragma omp for schedule(dynamic, 4)
        for (int b = 0; b < batch; ++b) {
            for (int i = 0; i < n; ++i) {
                double a = f0[i];
                for (int c = 0; c < nx; ++c) a += F[i * nx + c] * x0[(size_t)b * nx + c];
                q[i] = a;
            }
            int it = 0;
            const int st = qp_ipm_solve(n, m, H, q, G, h, U + (size_t)b * n, z, 100, abstol, reltol, feastol, &it, work);
            if (iters) iters[b] = it;
            if (st != 0) ++bad;
        }
        free(work); free(q); free(z);
    }
    free(G); free(h);
    return bad;
}
