// General-constraint MPC path behind the handle API: state-box and polytopic rows on x_1..x_N (and horizons with more
// than 128 inputs), so that a C caller of mpcqp_factor / mpcqp_solve gets what the Python Planner gets.
//
// Reference counterpart: the stacking of the inequality rows and their right-hand sides on the host --
// input box rows +-I (LaunchPlannerQPFormulation.py:96-107), state rows = rows of the prediction matrix with the affine
// rollout terms moved to the right-hand side (:109-118), polytopic rows (Launch3DPlanner.py:134-156), then
// Ain / Bin = vstack (:120-121) -- and the pruning of rows that constrain nothing (HelperFuncs.py:41-54).
//
// factor: build [I; -I; Gamma; -Gamma; C Gamma; -C Gamma] (G, m_full x n), the matching maps of x0 (E, m_full x nx) and
// bounds (h0) on the device, drop the rows with an infinite bound (flag / scan / ordered copy), then set up the
// constraint-space kernel (rows_qp.hip) or, if its m x m dual Hessian does not fit the memory limit, the general dense
// kernel (dense_qp.hip).  solve: q = F x0 + f0 and h = h0 - E x0 are formed on the device by the solver's own thin GEMMs.
#include "common.h"
#include <algorithm>
#include <cmath>

int gemm_nt_bias(const double* A, long lda, const double* B, long ldb, double* C, long ldc, int M, int N, int K, double alpha,
                 const double* bias, hipStream_t st);        // rows_qp.hip

namespace {

struct RowsParams {
    const double *Gam, *Phi;                    // (p, n), (p, nx)
    const double *umin, *umax; int per_stage;
    const double *xmin, *xmax;                  // (N, nx) or nullptr
    const double *C, *cmin, *cmax; int nc;      // (N, nc, nx), (N, nc) or nullptr
    double *G, *E, *h;                          // (m_full, n), (m_full, nx), (m_full); with `pos`: (m, .) kept rows only
    int n, nx, nu, N, p, xbox, mfull;
    const int* pos;                             // nullptr: every row at its own index; else row -> kept index or -1 (dropped)
    int m;                                      // rows kept (with pos)
    long bnd_stride;                            // one model per instance (blockIdx.z): doubles between the input bounds of models
    int mdl0;                                   // first model of this launch (grid.z holds at most 65,535: launches go by chunks)
};

// the bound of row `row` of [I; -I; Gamma; -Gamma; C Gamma; -C Gamma] U <= bound - E x0 for model `mdl` (only the input
// bounds can differ between models)
__device__ __forceinline__ double row_bound(const RowsParams& r, int row, size_t mdl) {
    const int n = r.n, p = r.p;
    if (row < 2 * n) {
        const bool up = row < n;
        const int i = up ? row : row - n;
        const int j = r.per_stage ? i : (i % r.nu);
        return up ? r.umax[mdl * r.bnd_stride + j] : -r.umin[mdl * r.bnd_stride + j];
    }
    int rr = row - 2 * n;
    const int xrows = r.xbox ? 2 * p : 0;
    if (rr < xrows) { const bool up = rr < p; const int i = up ? rr : rr - p; return up ? r.xmax[i] : -r.xmin[i]; }
    rr -= xrows;
    const int half = r.N * r.nc;
    const bool up = rr < half;
    const int i = up ? rr : rr - half;
    return up ? r.cmax[i] : -r.cmin[i];
}

// pass 1: keep[row] = 1 if ANY model has a finite bound on the row (a row open in model 0 but bounded in model b must
// stay; where it is open, its infinite bound never binds)
__global__ __launch_bounds__(256) void finite_rows_kernel(RowsParams r, int nm, int* keep) {
    const int row = blockIdx.x;
    if (row >= r.mfull) return;
    const int nmr = (row < 2 * r.n && r.bnd_stride) ? nm : 1;      // rows beyond the input box have one bound for all models
    bool any = false;
    for (int mdl = blockIdx.y * 256 + threadIdx.x; mdl < nmr; mdl += gridDim.y * 256) any = any || isfinite(row_bound(r, row, (size_t)mdl));
    if (any) keep[row] = 1;
}

// one thread per (row, column of [G | E]); column 0 also writes the bound
__global__ __launch_bounds__(256) void build_rows_kernel(RowsParams r) {
    const int row = blockIdx.x, c = blockIdx.y * 256 + threadIdx.x;
    const int n = r.n, nx = r.nx, p = r.p;
    if (row >= r.mfull || c >= n + nx) return;
    const int orow = r.pos ? r.pos[row] : row;          // where the row lands (pruned layout), -1: dropped
    if (orow < 0) return;
    const size_t mdl = (size_t)r.mdl0 + blockIdx.z;     // model (0 for a shared model)
    const int mrows = r.pos ? r.m : r.mfull;
    r.Gam += mdl * (size_t)p * n; r.Phi += mdl * (size_t)p * nx;
    r.umin += mdl * r.bnd_stride; r.umax += mdl * r.bnd_stride;
    r.G += mdl * (size_t)mrows * n; r.E += mdl * (size_t)mrows * nx; r.h += mdl * (size_t)mrows;
    int rr = row;
    double val = 0.0, bound = 0.0;
    const bool isG = c < n;
    const int cc = isG ? c : c - n;
    if (rr < 2 * n) {                                   // input box: +-I, nothing from x0
        const bool up = rr < n;
        const int i = up ? rr : rr - n;
        const int j = r.per_stage ? i : (i % r.nu);
        val = (isG && cc == i) ? (up ? 1.0 : -1.0) : 0.0;
        bound = up ? r.umax[j] : -r.umin[j];
    } else {
        rr -= 2 * n;
        const int xrows = r.xbox ? 2 * p : 0;
        if (rr < xrows) {                               // state box: +-Gamma rows, +-Phi rows
            const bool up = rr < p;
            const int i = up ? rr : rr - p;
            const double v = isG ? r.Gam[(size_t)i * n + cc] : r.Phi[(size_t)i * nx + cc];
            val = up ? v : -v;
            bound = up ? r.xmax[i] : -r.xmin[i];
        } else {                                        // polytopic rows: +-(C_k Gamma_k), +-(C_k Phi_k)
            rr -= xrows;
            const int half = r.N * r.nc;
            const bool up = rr < half;
            const int i = up ? rr : rr - half;
            const int k = i / r.nc, a = i - k * r.nc;
            const double* crow = r.C + ((size_t)k * r.nc + a) * nx;
            double s = 0.0;
            for (int j = 0; j < nx; ++j)
                s += crow[j] * (isG ? r.Gam[(size_t)(k * nx + j) * n + cc] : r.Phi[(size_t)(k * nx + j) * nx + cc]);
            val = up ? s : -s;
            bound = up ? r.cmax[i] : -r.cmin[i];
        }
    }
    if (isG) r.G[(size_t)orow * n + cc] = val; else r.E[(size_t)orow * nx + cc] = val;
    if (c == 0) r.h[orow] = bound;
}

// pos[i] = kept rows before row i (-1: dropped); *count = their number (one workgroup, chunked scan)
__global__ __launch_bounds__(1024) void finite_scan_kernel(const int* keep, int rows, int* pos, int* count) {
    __shared__ int part[1024];
    __shared__ int carry;
    const int tid = threadIdx.x;
    if (tid == 0) carry = 0;
    __syncthreads();
    for (int base = 0; base < rows; base += 1024) {
        const int i = base + tid;
        const int k = (i < rows && keep[i]) ? 1 : 0;
        part[tid] = k;
        __syncthreads();
        for (int s = 1; s < 1024; s <<= 1) {
            const int v = tid >= s ? part[tid - s] : 0;
            __syncthreads();
            part[tid] += v;
            __syncthreads();
        }
        if (i < rows) pos[i] = k ? carry + part[tid] - 1 : -1;
        __syncthreads();
        if (tid == 1023) carry += part[1023];
        __syncthreads();
    }
    if (tid == 0) *count = carry;
}

__global__ __launch_bounds__(256) void compact_rows_kernel(const double* Gf, const double* Ef, const double* hf, const int* pos,
                                                           int rows, int n, int nx, double* G, double* E, double* h0) {
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (row >= rows) return;
    const int o = pos[row];
    if (o < 0) return;
    for (int j = lane; j < n; j += 64) G[(size_t)o * n + j] = Gf[(size_t)row * n + j];
    for (int j = lane; j < nx; j += 64) E[(size_t)o * nx + j] = Ef[(size_t)row * nx + j];
    if (lane == 0) h0[o] = hf[row];
}

// status of the row kernels (0 optimal | 1 step cap | 2 infeasible | 3 numerical) -> MPCQP_SOLVED | MPCQP_MAX_ITER | 3 | 4,
// and the remaining info fields
__global__ void general_info_kernel(const int* st, const int* it, int B, mpcqp_info info) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    if (info.status) info.status[i] = st[i] == 0 ? MPCQP_SOLVED : st[i] + 1;
    if (info.polish_steps) info.polish_steps[i] = it[i] % 1000;
    if (info.iters) info.iters[i] = 0;
    if (info.prim_res) info.prim_res[i] = 0.0;
    if (info.dual_res) info.dual_res[i] = 0.0;
}

// one model per instance: q_b = F_b x0_b + f0_b (n), h_b = h0_b - E_b x0_b (m)
__global__ __launch_bounds__(256) void affine_per_model_kernel(const double* x0, const double* F, const double* f0, const double* E,
                                                               const double* h0, double* q, double* hh, int n, int m, int nx, int b0) {
    const size_t b = (size_t)b0 + blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n + m) return;
    const double* xb = x0 + b * nx;
    if (i < n) {
        const double* fr = F + (b * n + i) * nx;
        double s = f0[b * n + i];
        for (int c = 0; c < nx; ++c) s += fr[c] * xb[c];
        q[b * n + i] = s;
    } else {
        const int r = i - n;
        const double* er = E + (b * m + r) * nx;
        double s = h0[b * m + r];
        for (int c = 0; c < nx; ++c) s -= er[c] * xb[c];
        hh[b * m + r] = s;
    }
}

template <typename T>
int grow(mpcqp_handle h, T** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
    return MPCQP_OK;
}

#define TRYG(x) do { int rc_ = (x); if (rc_ != MPCQP_OK) return rc_; } while (0)

}  // namespace

void general_free(mpcqp_handle h) {
    void* ptrs[] = {h->gG, h->gE, h->gh0, h->g_shared, h->g_iws, h->g_inst, h->g_q, h->g_hh, h->g_z, h->g_obj, h->g_st, h->g_it,
                    h->g_dws, h->g_diws};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->gG = h->gE = h->gh0 = h->g_shared = h->g_inst = h->g_q = h->g_hh = h->g_z = h->g_obj = h->g_dws = nullptr;
    h->g_iws = h->g_st = h->g_it = h->g_diws = nullptr;
    h->g_m = 0; h->g_bcap = 0; h->g_dcap = 0; h->general = false; h->g_wvalid = 0;
}

// synchronous (it is a set-up call): the number of rows that survive the pruning decides every later allocation
int factor_general(mpcqp_handle h, hipStream_t st) {
    const int n = h->n, nx = h->dims.nx, p = h->p, N = h->dims.N, nm = h->dims.n_models;
    const bool xbox = h->g_xmin || h->g_xmax;
    if (xbox && !(h->g_xmin && h->g_xmax)) { h->err = "factor: x_min and x_max must both be given (use +-inf for an open side)"; return MPCQP_EINVAL; }
    const int nc = h->g_C ? h->g_nc : 0;
    const long mfull = 2L * n + (xbox ? 2L * p : 0) + 2L * N * nc;
    general_free(h);
    // the rows are built from Gamma: assemble it if the last assembly took the route that skips it
    if (!h->gam_valid && (xbox || nc > 0)) { const int ra = launch_assemble(h, st, true); if (ra != MPCQP_OK) return ra; }
    int* keep = nullptr; int* pos = nullptr;
    auto cleanup = [&]() { for (void* q : {(void*)keep, (void*)pos}) if (q) (void)hipFree(q); };
    int rc = grow(h, &keep, (size_t)mfull);
    if (rc == MPCQP_OK) rc = grow(h, &pos, (size_t)mfull + 1);
    if (rc == MPCQP_OK && hipMemsetAsync(keep, 0, (size_t)mfull * sizeof(int), st) != hipSuccess) rc = MPCQP_EHIP;
    if (rc != MPCQP_OK) { cleanup(); return rc; }
    RowsParams r;
    r.Gam = h->Gam; r.Phi = h->Phi; r.umin = h->umin; r.umax = h->umax; r.per_stage = h->per_stage;
    r.xmin = h->g_xmin; r.xmax = h->g_xmax; r.C = h->g_C; r.cmin = h->g_cmin; r.cmax = h->g_cmax; r.nc = nc;
    r.n = n; r.nx = nx; r.nu = h->dims.nu; r.N = N; r.p = p; r.xbox = xbox ? 1 : 0; r.mfull = (int)mfull;
    r.bnd_stride = h->bnd_stride; r.mdl0 = 0;
    // pass 1: the bounds alone decide which rows exist -- a row whose bound is infinite in EVERY model constrains nothing
    // (one model per instance: a kept row whose own bound is infinite never binds in that model)
    r.G = nullptr; r.E = nullptr; r.h = nullptr; r.pos = nullptr; r.m = 0;
    hipLaunchKernelGGL(finite_rows_kernel, dim3((unsigned)mfull, (unsigned)std::min(64, ceil_div(nm, 256))), dim3(256), 0, st, r, nm, keep);
    hipLaunchKernelGGL(finite_scan_kernel, dim3(1), dim3(1024), 0, st, keep, (int)mfull, pos, pos + mfull);
    int m = 0;
    hipError_t e = hipGetLastError();
    if (e == hipSuccess) e = hipMemcpyAsync(&m, pos + mfull, sizeof(int), hipMemcpyDeviceToHost, st);
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    if (e != hipSuccess) { cleanup(); h->err = std::string("factor (general rows): ") + hipGetErrorString(e); return MPCQP_EHIP; }
    if (m < 1) { cleanup(); h->err = "factor: no finite constraint row is left"; return MPCQP_EINVAL; }
    rc = grow(h, &h->gG, (size_t)nm * m * n);
    if (rc == MPCQP_OK) rc = grow(h, &h->gE, (size_t)nm * m * nx);
    if (rc == MPCQP_OK) rc = grow(h, &h->gh0, (size_t)nm * m);
    if (rc != MPCQP_OK) { cleanup(); return rc; }
    // pass 2: the kept rows of every model, written at their pruned positions
    r.G = h->gG; r.E = h->gE; r.h = h->gh0; r.pos = pos; r.m = m;
    for (int m0 = 0; m0 < nm; m0 += 65535) {           // grid.z holds at most 65,535 models per launch
        r.mdl0 = m0;
        hipLaunchKernelGGL(build_rows_kernel, dim3((unsigned)mfull, ceil_div(n + nx, 256), (unsigned)std::min(65535, nm - m0)), dim3(256), 0, st, r);
    }
    e = hipGetLastError();
    if (e == hipSuccess) e = hipStreamSynchronize(st);
    cleanup();
    if (e != hipSuccess) { h->err = std::string("factor (general rows): ") + hipGetErrorString(e); return MPCQP_EHIP; }
    h->g_m = m;
    // shared model: constraint-space kernel if its shared workspace (the m x m dual Hessian) fits the limit, else the general
    // dense kernel on the shared set-up.  One model per instance: every instance is a dense QP of its own (P_b, G_b).
    long per = 0;
    const long ns = mpcqp_rows_workspace(n, m, &per);
    h->g_rows = nm == 1 && (h->g_path != 1) && (8.0 * (double)ns <= (double)h->g_mem_limit) && (h->g_xmin || h->g_C || h->g_path == 2);
    if (h->g_rows) {
        TRYG(grow(h, &h->g_shared, (size_t)ns));
        TRYG(grow(h, &h->g_iws, (size_t)n));
        h->g_per = per;
        rc = mpcqp_rows_setup(h->dev, n, m, h->H, h->gG, h->g_shared, h->g_iws, st);
        if (rc != MPCQP_OK) { h->err = "factor: mpcqp_rows_setup failed"; return rc; }
    }
    h->general = true; h->factored = true; h->has_state = false; h->rho = 0.0; h->g_wvalid = 0;
    return MPCQP_OK;
}

int solve_general(mpcqp_handle h, const double* x0, int B, double* U, const mpcqp_info* info, hipStream_t st, int warm) {
    const int n = h->n, m = h->g_m, nx = h->dims.nx;
    if (B > h->g_bcap) {
        h->g_wvalid = 0;        // per-batch buffers (grow only; like ensure_batch this may allocate on the first larger batch)
        if (h->pinned) { h->err = "solve: a larger batch would re-allocate workspaces that a recorded graph points into (mpcqp_pin)"; return MPCQP_ESTATE; }
        TRYG(grow(h, &h->g_z, (size_t)B * m));
        TRYG(grow(h, &h->g_obj, (size_t)B));
        TRYG(grow(h, &h->g_st, (size_t)B));
        TRYG(grow(h, &h->g_it, (size_t)B));
        if (h->g_rows) TRYG(grow(h, &h->g_inst, (size_t)B * h->g_per));
        else { TRYG(grow(h, &h->g_q, (size_t)B * n)); TRYG(grow(h, &h->g_hh, (size_t)B * m)); }
        h->g_bcap = B;
    }
    double* z = (info && info->lam) ? info->lam : h->g_z;           // general mode: lam is (batch, m), one value per kept row
    double* obj = (info && info->obj) ? info->obj : h->g_obj;
    int rc;
    if (h->dims.n_models > 1) {
        // one model per instance: B dense QPs with their own Hessian and rows (the kernel of the literal solvers.qp call)
        if (B != h->dims.n_models) { h->err = "solve: with one model per instance the batch must equal n_models"; return MPCQP_EINVAL; }
        long ni = 0;
        const long nd = mpcqp_dense_workspace(n, m, 0, &ni);
        if (B > h->g_dcap) {
            TRYG(grow(h, &h->g_dws, (size_t)B * nd));
            TRYG(grow(h, &h->g_diws, (size_t)B * ni));
            h->g_dcap = B;
        }
        for (int b0 = 0; b0 < B; b0 += 65535)           // grid.y holds at most 65,535 instances per launch
            hipLaunchKernelGGL(affine_per_model_kernel, dim3(ceil_div(n + m, 256), (unsigned)std::min(65535, B - b0)), dim3(256), 0, st, x0,
                               h->F, h->f0, h->gE, h->gh0, h->g_q, h->g_hh, n, m, nx, b0);
        HIPCHK(h, hipGetLastError());
        rc = mpcqp_solve_dense_opts(h->dev, B, n, m, 0, h->H, (long)n * n, h->g_q, n, h->gG, (long)m * n, h->g_hh, m, nullptr, 0,
                                    nullptr, 0, U, z, nullptr, obj, h->g_st, h->g_it, h->g_dws, h->g_diws, nullptr, st);
        if (rc != MPCQP_OK) { h->err = "solve: dense path (one model per instance) failed"; return rc; }
    } else if (h->g_rows) {
        // warm: from the working sets the previous solve of the same batch left in the instance workspace (rows_qp.hip)
        const bool w = warm && h->g_wvalid == B;
        rc = (w ? mpcqp_rows_solve_mpc_warm : mpcqp_rows_solve_mpc)(h->dev, B, n, m, nx, x0, h->F, h->f0, h->gE, h->gh0, U, z, obj, h->g_st,
                                                                     h->g_it, h->g_shared, h->g_inst, st);
        if (rc != MPCQP_OK) { h->err = "solve: mpcqp_rows_solve_mpc failed"; return rc; }
        h->g_wvalid = B;
    } else {
        long ni = 0;
        const long nd = mpcqp_dense_workspace(n, m, 0, &ni);
        if (B > h->g_dcap) {    // the shared set-up lives in the workspace of QP 0: redo it when the workspace moves
            TRYG(grow(h, &h->g_dws, (size_t)B * nd));
            TRYG(grow(h, &h->g_diws, (size_t)B * ni));
            h->g_dcap = B;
            rc = mpcqp_dense_shared_setup(h->dev, n, m, 0, h->H, h->gG, nullptr, h->g_dws, h->g_diws, st);
            if (rc != MPCQP_OK) { h->err = "solve: mpcqp_dense_shared_setup failed"; return rc; }
        }
        // q = x0 F' + f0,  h = h0 - x0 E'
        rc = gemm_nt_bias(x0, nx, h->F, nx, h->g_q, n, B, n, nx, 1.0, h->f0, st);
        if (rc == MPCQP_OK) rc = gemm_nt_bias(x0, nx, h->gE, nx, h->g_hh, m, B, m, nx, -1.0, h->gh0, st);
        if (rc == MPCQP_OK)
            rc = mpcqp_dense_shared_solve(h->dev, B, n, m, 0, h->g_q, h->g_hh, nullptr, U, z, nullptr, obj, h->g_st, h->g_it,
                                          h->g_dws, h->g_diws, st);
        if (rc != MPCQP_OK) { h->err = "solve: dense path failed"; return rc; }
    }
    if (info) {
        hipLaunchKernelGGL(general_info_kernel, dim3(ceil_div(B, 256)), dim3(256), 0, st, h->g_st, h->g_it, B, *info);
        HIPCHK(h, hipGetLastError());
    }
    h->has_state = false; h->Blast = B;
    return MPCQP_OK;
}

extern "C" {

int mpcqp_set_state_constraints(mpcqp_handle h, const double* x_min, const double* x_max, const double* C, const double* c_min,
                                const double* c_max, int nc) {
    if (!h) return MPCQP_EINVAL;
    if ((x_min == nullptr) != (x_max == nullptr)) { h->err = "set_state_constraints: x_min and x_max go together (+-inf opens a side)"; return MPCQP_EINVAL; }
    if (C && (nc < 1 || !c_min || !c_max)) { h->err = "set_state_constraints: C needs nc >= 1, c_min and c_max"; return MPCQP_EINVAL; }
    if (!C && (c_min || c_max)) { h->err = "set_state_constraints: c_min / c_max need C"; return MPCQP_EINVAL; }
    h->g_xmin = x_min; h->g_xmax = x_max; h->g_C = C; h->g_cmin = c_min; h->g_cmax = c_max; h->g_nc = C ? nc : 0;
    h->factored = false;
    return MPCQP_OK;
}

int mpcqp_set_general_opts(mpcqp_handle h, int path, long mem_limit_bytes) {
    if (!h) return MPCQP_EINVAL;
    if (path < 0 || path > 2 || mem_limit_bytes < 0) { h->err = "set_general_opts: path in {0, 1, 2}, mem_limit_bytes >= 0"; return MPCQP_EINVAL; }
    h->g_path = path;
    if (mem_limit_bytes > 0) h->g_mem_limit = mem_limit_bytes;
    h->factored = false;
    return MPCQP_OK;
}

int mpcqp_general_rows(mpcqp_handle h, int* m, int* uses_rows_kernel) {
    if (!h || !m) return MPCQP_EINVAL;
    *m = h->general ? h->g_m : 0;
    if (uses_rows_kernel) *uses_rows_kernel = (h->general && h->g_rows) ? 1 : 0;
    return MPCQP_OK;
}

}  // extern "C"
