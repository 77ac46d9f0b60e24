// extern "C" surface of libmpcqp.so (declared in include/mpcqp.h).
#include "common.h"
#include <algorithm>
#include <cmath>
#include <cstring>
#include <new>

#include <map>
#include <mutex>
#include <utility>

hipError_t ensure_dyn_lds(int device, const void* kernel, size_t bytes) {
    static std::mutex mu;
    static std::map<std::pair<int, const void*>, size_t> set_to;
    std::lock_guard<std::mutex> lock(mu);
    size_t& have = set_to[{device, kernel}];
    if (have >= bytes && have > 0) return hipSuccess;
    const hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes);
    if (e == hipSuccess) have = bytes;
    return e;
}

namespace {

std::string g_create_err;

template <typename T>
int dev_alloc(mpcqp_handle h, T** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    HIPCHK(h, hipMemset(*p, 0, count * sizeof(T)));
    return MPCQP_OK;
}

#define TRY(x) do { int rc_ = (x); if (rc_ != MPCQP_OK) return rc_; } while (0)

int ensure_batch(mpcqp_handle h, int B) {
    const int need = ceil_div(B, 64) * 64;
    if (need <= h->Bcap) return MPCQP_OK;
    if (h->pinned) { h->err = "solve: a larger batch would re-allocate workspaces that a recorded graph points into (mpcqp_pin)"; return MPCQP_ESTATE; }
    TRY(dev_alloc(h, &h->z, (size_t)need * h->NP));
    TRY(dev_alloc(h, &h->w, (size_t)need * h->NP));
    TRY(dev_alloc(h, &h->b0ws, (size_t)need * h->NP));
    TRY(dev_alloc(h, &h->it_ws, (size_t)need * (5 + 32) + 96 + 1024));   // iters | need_big | converged | queue 1 | queue 2 | counters | bucket sizes (32 x 32) | the 32 buckets of the cold-start order
    TRY(dev_alloc(h, &h->rp_ws, (size_t)need));
    TRY(dev_alloc(h, &h->rd_ws, (size_t)need * 3));      // dual residuals | what the two residuals are measured against (mpcqp_adapt_rho)
    // the hand-over records of the whole-instance cascade (2.6 KB per instance) are allocated by the first solve that takes
    // that cascade (ensure_handover): models solved component by component, one model per instance, warm-started and
    // ADMM-first solves never touch them
    if (h->hbuf1) { (void)hipFree(h->hbuf1); h->hbuf1 = nullptr; }
    if (h->hbuf2) { (void)hipFree(h->hbuf2); h->hbuf2 = nullptr; }
    h->hcap = 0;
    h->Bcap = need;
    h->has_state = false;
    return MPCQP_OK;
}

}  // namespace

int ensure_handover(mpcqp_handle h) {
    if (h->hcap >= h->Bcap) return MPCQP_OK;
    if (h->pinned) { h->err = "solve: this solve needs buffers that were not allocated when the handle was pinned (mpcqp_pin)"; return MPCQP_ESTATE; }
    TRY(dev_alloc(h, &h->hbuf1, (size_t)h->Bcap * (16 * 16 + 4 * 16 + 2)));
    TRY(dev_alloc(h, &h->hbuf2, (size_t)(h->Bcap / 4) * (32 * 32 + 4 * 32 + 2)));
    h->hcap = h->Bcap;
    return MPCQP_OK;
}

namespace {

__global__ void shift_kernel(double* z, double* w, int nblk, int NP, int n, int nu, int zero_z) {
    // one 64-lane wave per instance; rows move forward by one stage (nu rows), the last stage repeats
    const int inst = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (inst >= nblk) return;
    double* zb = z + (size_t)inst * NP;
    double* wb = w + (size_t)inst * NP;
    double zv[2], wv[2];
    for (int q = 0; q < 2; ++q) {
        const int r = lane + 64 * q;
        const int src = (r + nu < n) ? r + nu : r;
        zv[q] = (r < n && !zero_z) ? zb[src] : 0.0;
        wv[q] = r < n ? wb[src] : 0.0;
    }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
    for (int q = 0; q < 2; ++q) {
        const int r = lane + 64 * q;
        if (r < n) { zb[r] = zv[q]; wb[r] = wv[q]; }
    }
}

__global__ void ints_to_doubles_kernel(const int* in, double* out, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < count) out[i] = (double)in[i];
}

__global__ void unblock_kernel(const double* zb, double* out, int B, int NP, int n) {
    const size_t idx = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (size_t)B * n) return;
    const int inst = (int)(idx / n), j = (int)(idx % n);
    out[idx] = zb[(size_t)inst * NP + j];
}

// sum over the batch of log(r_prim) - log(r_dual) (instances with both residuals > 0) and their number: one workgroup
__global__ __launch_bounds__(256) void rho_stats_kernel(const double* rp, const double* rd, const double* pn, const double* dn, int B, double* out) {
    __shared__ double ssum[256];
    __shared__ int scnt[256];
    double a = 0.0; int c = 0;
    for (int i = threadIdx.x; i < B; i += 256) {
        // OSQP's normalised residuals: r_prim / max(|x|, |z|), r_dual / (rho |w|), the scales the kernel's own stopping test uses
        const double p = rp[i] / pn[i], d = rd[i] / dn[i];
        if (p > 0.0 && d > 0.0 && isfinite(p) && isfinite(d)) { a += log(p) - log(d); ++c; }
    }
    ssum[threadIdx.x] = a; scnt[threadIdx.x] = c;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) { ssum[threadIdx.x] += ssum[threadIdx.x + s]; scnt[threadIdx.x] += scnt[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { out[0] = ssum[0]; out[1] = (double)scnt[0]; }
}

__global__ __launch_bounds__(256) void scale_vec_kernel(double* v, size_t count, double f) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) v[i] *= f;
}

}  // namespace

extern "C" {

const char* mpcqp_last_error(mpcqp_handle h) { return h ? h->err.c_str() : g_create_err.c_str(); }

void mpcqp_default_solve_opts(mpcqp_solve_opts* o) {
    if (!o) return;
    o->max_iter = 0; o->min_iter = 5; o->check_every = 5;
    o->eps_abs = 1e-6; o->eps_rel = 1e-6; o->alpha = 1.6;
    o->polish = 1; o->polish_refine = -1; o->polish_max_steps = 0; o->warm_start = 0; o->stages = 3; o->keep_state = 1;
}

int mpcqp_create(mpcqp_handle* out, const mpcqp_dims* dims, int device_id) {
    if (!out || !dims) { g_create_err = "null argument"; return MPCQP_EINVAL; }
    *out = nullptr;
    if (dims->nx < 1 || dims->nu < 1 || dims->N < 1 || dims->n_models < 1) {
        g_create_err = "nx, nu, N, n_models must be >= 1"; return MPCQP_EINVAL;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) { g_create_err = "no HIP device"; return MPCQP_EHIP; }
    if (device_id < 0 || device_id >= ndev) { g_create_err = "device_id out of range"; return MPCQP_EINVAL; }
    mpcqp_handle h = new (std::nothrow) mpcqp_handle_s;
    if (!h) { g_create_err = "out of host memory"; return MPCQP_EHIP; }
    h->dev = device_id; h->dims = *dims;
    h->n = dims->N * dims->nu; h->p = dims->N * dims->nx;
    h->NT = ceil_div(h->n, 16); h->NP = 16 * h->NT; h->S = ceil_div(h->n, 4); h->SX = ceil_div(dims->nx + 1, 4);
    hipError_t e = hipSetDevice(device_id);
    if (e != hipSuccess) { g_create_err = hipGetErrorString(e); delete h; return MPCQP_EHIP; }
    const size_t nm = dims->n_models, n = h->n, p = h->p, nx = dims->nx;
    int rc = MPCQP_OK;
    auto A_ = [&](auto** ptr, size_t cnt) { if (rc == MPCQP_OK) rc = dev_alloc(h, ptr, cnt); };
    A_(&h->Gam, nm * p * n); A_(&h->QG, nm * p * n); A_(&h->Phi, nm * p * nx);
    A_(&h->H, nm * n * n); A_(&h->F, nm * n * nx); A_(&h->f0, nm * n);
    // solver data: one copy for a shared model, one per model otherwise (W1, W2 are then only used when the
    // factorisation does not fit LDS)
    const bool w_in_lds = nm > 1 && 4 * n * n * sizeof(double) <= 128 * 1024;
    A_(&h->d, nm * n); A_(&h->Hs, nm * n * n); A_(&h->M, n * n); A_(&h->Hi, nm * n * n);
    A_(&h->W1, (w_in_lds ? 1 : nm) * n * n);
    A_(&h->W2, (w_in_lds ? 1 : nm) * n * std::max(n, nx + 1));
    if (nm == 1) { A_(&h->facL, n * n); A_(&h->facBlk, (size_t)blocked_inverse_scratch((int)n)); }
    A_(&h->Fs1, nm * n * (nx + 1)); A_(&h->Ku1, nm * n * (nx + 1)); A_(&h->Cq, (nx + 1) * (nx + 1));
    A_(&h->FsT, nm * h->NP * (nx + 1)); A_(&h->KuT, nm * h->NP * (nx + 1));
    A_(&h->Mfrag, (size_t)h->NT * h->S * 64); A_(&h->Kbfrag, (size_t)h->NT * h->SX * 64);
    A_(&h->Kufrag, (size_t)h->NT * h->SX * 64);
    A_(&h->lbs, nm * h->NP); A_(&h->ubs, nm * h->NP); A_(&h->rho_dev, 4); A_(&h->flag_dev, 4); A_(&h->mflag, nm);
    if (rc != MPCQP_OK) { g_create_err = h->err; mpcqp_destroy(h); return rc; }
    *out = h;
    return MPCQP_OK;
}

int mpcqp_destroy(mpcqp_handle h) {
    if (!h) return MPCQP_OK;
    (void)hipSetDevice(h->dev);
    void* ptrs[] = {h->Gam, h->QG, h->Phi, h->H, h->F, h->f0, h->d, h->Hs, h->M, h->Hi, h->W1, h->W2, h->Fs1,
                    h->Ku1, h->Cq, h->FsT, h->KuT, h->Mfrag, h->Kbfrag, h->Kufrag, h->lbs, h->ubs, h->rho_dev, h->flag_dev, h->mflag,
                    h->z, h->w, h->b0ws, h->it_ws, h->rp_ws, h->rd_ws, h->hbuf1, h->hbuf2, h->facL, h->facBlk, h->adapt_ws};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    general_free(h);
    split_free(h);
    delete h;
    return MPCQP_OK;
}

int mpcqp_set_model(mpcqp_handle h, const double* A, const double* B) {
    if (!h) return MPCQP_EINVAL;
    if (!A || !B) { h->err = "set_model: null pointer"; return MPCQP_EINVAL; }
    h->A = A; h->B = B; h->assembled = h->factored = false;
    return MPCQP_OK;
}

int mpcqp_set_cost(mpcqp_handle h, const double* Q, const double* R, const double* P,
                   const double* x_ref, const double* u_ref) {
    if (!h) return MPCQP_EINVAL;
    if (!Q || !R || !P) { h->err = "set_cost: Q, R, P are required"; return MPCQP_EINVAL; }
    h->Q = Q; h->R = R; h->P = P; h->xref = x_ref; h->uref = u_ref;
    h->assembled = h->factored = false;
    return MPCQP_OK;
}

int mpcqp_set_constraints(mpcqp_handle h, const double* u_min, const double* u_max, int per_stage) {
    if (!h) return MPCQP_EINVAL;
    if (!u_min || !u_max) { h->err = "set_constraints: u_min and u_max are required"; return MPCQP_EINVAL; }
    h->umin = u_min; h->umax = u_max; h->per_stage = per_stage ? 1 : 0;
    h->factored = false;
    return MPCQP_OK;
}

int mpcqp_set_strides(mpcqp_handle h, long xref_stride, long uref_stride, long bound_stride) {
    if (!h) return MPCQP_EINVAL;
    if (xref_stride < 0 || uref_stride < 0 || bound_stride < 0) { h->err = "set_strides: strides must be >= 0"; return MPCQP_EINVAL; }
    if (h->dims.n_models == 1 && (xref_stride || uref_stride || bound_stride)) {
        h->err = "set_strides: per-model data needs n_models > 1"; return MPCQP_EINVAL;
    }
    h->xref_stride = xref_stride; h->uref_stride = uref_stride; h->bnd_stride = bound_stride;
    h->assembled = h->factored = false;
    return MPCQP_OK;
}

int mpcqp_assemble(mpcqp_handle h, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!h->A || !h->Q) { h->err = "assemble: set_model and set_cost first"; return MPCQP_ESTATE; }
    if (h->pinned && h->dims.n_models == 1) { h->err = "assemble: a recorded graph reads this model's matrices (mpcqp_pin): unpin first"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    TRY(launch_assemble(h, static_cast<hipStream_t>(stream), h->g_xmin || h->g_C));
    h->assembled = true; h->factored = false;
    return MPCQP_OK;
}

int mpcqp_factor(mpcqp_handle h, const mpcqp_factor_opts* opts, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!h->assembled) { h->err = "factor: assemble first"; return MPCQP_ESTATE; }
    if (!h->umin) { h->err = "factor: set_constraints first"; return MPCQP_ESTATE; }
    if (h->pinned && h->dims.n_models == 1) { h->err = "factor: a recorded graph points into this factorisation's buffers (mpcqp_pin): unpin first"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    hipStream_t st = static_cast<hipStream_t>(stream);
    h->sp_C = 0;            // the component split belongs to the factorisation it was found in (split_setup below)
    // state-box / polytopic rows, or more inputs than the box kernels hold: the general-constraint path (general.hip)
    // (more than 64 states: the box kernels stage x0 in 64 lanes; the general path forms q = F x0 + f0 with a GEMM of any depth)
    if (h->g_xmin || h->g_C || h->g_path == 2 || !admm_shared_box_supported(h->NT, h->S) || h->dims.nx > 64) return factor_general(h, st);
    if (h->general) general_free(h);
    if (h->dims.n_models > 1) TRY(launch_factor_batch(h, st, !(opts && opts->no_sync)));      // one model per instance: active-set path only
    else TRY(launch_factor(h, opts ? opts->rho : 0.0, st));
    if (opts && opts->no_sync && (h->dims.n_models > 1 || opts->rho > 0.0)) {
        // closed-loop use: no host round trip.  A failed factorisation is then reported per instance by the
        // solve (status MPCQP_MAX_ITER); whether the polish refines follows the last synchronous factor call.
        h->rho = h->dims.n_models > 1 ? 1.0 : opts->rho;
        h->factored = true; h->has_state = h->has_state && opts->keep_state;
        return MPCQP_OK;
    }
    // set-up call: fetch rho and the failure flag (the only host synchronisation besides create)
    double r3[4]; int flag = 0;
    HIPCHK(h, hipMemcpyAsync(r3, h->rho_dev, sizeof(r3), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(&flag, h->flag_dev, sizeof(int), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    if (flag) { h->err = "factor: H is not positive definite (Cholesky pivot <= 0)"; return MPCQP_ENUMERIC; }
    h->rho = h->dims.n_models > 1 ? 1.0 : r3[0];
    h->inv_resid = r3[3];
    h->factored = true; h->has_state = h->has_state && opts && opts->keep_state;
    // a shared model whose Hessian is block diagonal under a permutation is solved block by block (split.hip)
    if (h->dims.n_models == 1) TRY(split_setup(h, st));
    return MPCQP_OK;
}

int mpcqp_solve(mpcqp_handle h, const double* x0, int batch, double* U, const mpcqp_info* info,
                const mpcqp_solve_opts* opts, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!h->factored) { h->err = "solve: factor first"; return MPCQP_ESTATE; }
    if (!x0 || !U || batch < 1) { h->err = "solve: x0, U must be non-null and batch >= 1"; return MPCQP_EINVAL; }
    mpcqp_solve_opts o;
    if (opts) o = *opts; else mpcqp_default_solve_opts(&o);
    if (!(o.alpha > 0.0 && o.alpha < 2.0) || o.max_iter < 0) { h->err = "solve: alpha must be in (0,2), max_iter >= 0"; return MPCQP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->dev));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (h->general) {
        // warm start: the constraint-space kernel continues from the working sets of the previous solve of the same batch (a cold
        // solve if there is none); the dense kernel keeps no state
        if (o.warm_start && !h->g_rows) { h->err = "solve: warm start in the general-constraint mode needs the constraint-space kernel (the dense kernel keeps no state)"; return MPCQP_EINVAL; }
        return solve_general(h, x0, batch, U, info, st, o.warm_start);
    }
    TRY(ensure_batch(h, batch));
    if (o.warm_start && h->has_state && batch != h->Blast) { h->err = "solve: warm start needs the same batch as the previous solve"; return MPCQP_EINVAL; }
    if (h->dims.n_models == 1 && (o.stages & 2) && !(o.stages & 1) && !(h->has_state && batch == h->Blast)) {
        h->err = "solve: stages=2 needs the state of a previous solve of the same batch"; return MPCQP_ESTATE;
    }
    if (h->dims.n_models > 1) {
        if (batch != h->dims.n_models) { h->err = "solve: with one model per instance the batch must equal n_models"; return MPCQP_EINVAL; }
        if (!o.polish) { h->err = "solve: one model per instance is solved by the active-set kernel; polish cannot be off"; return MPCQP_EINVAL; }
        o.stages = 2;       // no shared factor, hence no ADMM GEMM: the dual active-set kernel does the whole solve
    }
    // max_iter = 0 with the polish on: nothing for the ADMM kernel to do -- the active-set kernels start cold (empty set), or
    // from the kept (z, w) state when warm_start is set
    const bool skip_admm = o.max_iter == 0 && o.polish && (o.stages & 2);   // polish.hip: no_admm
    h->admm_last = false;
    if ((o.stages & 1) && h->dims.n_models == 1 && !skip_admm) { TRY(launch_admm_shared_box(h, x0, batch, o, st)); h->admm_last = true; }
    if (o.stages & 2) TRY(launch_polish_box(h, x0, batch, U, info, o, st));
    h->has_state = o.keep_state != 0; h->Blast = batch;
    return MPCQP_OK;
}

int mpcqp_adapt_rho(mpcqp_handle h, double threshold, double* rho_out, int* refactored, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (refactored) *refactored = 0;
    if (h->dims.n_models != 1 || h->general) { h->err = "adapt_rho: shared models with box bounds only (the ADMM kernel's regime)"; return MPCQP_ENOTIMPL; }
    if (!h->factored || !h->admm_last || h->Blast < 1) { h->err = "adapt_rho: needs a solve that ran ADMM iterations (max_iter > 0) on the current factorisation"; return MPCQP_ESTATE; }
    if (h->pinned) { h->err = "adapt_rho: a recorded graph points into this factorisation's buffers (mpcqp_pin): unpin first"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (!h->adapt_ws) HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->adapt_ws), 2 * sizeof(double)));
    hipLaunchKernelGGL(rho_stats_kernel, dim3(1), dim3(256), 0, st, h->rp_ws, h->rd_ws, h->rd_ws + h->Bcap, h->rd_ws + 2 * (size_t)h->Bcap, h->Blast, h->adapt_ws);
    HIPCHK(h, hipGetLastError());
    double sums[2] = {0.0, 0.0};
    HIPCHK(h, hipMemcpyAsync(sums, h->adapt_ws, sizeof(sums), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    const double thr = threshold > 1.0 ? threshold : 5.0;
    double rho_new = h->rho;
    if (sums[1] >= 1.0) rho_new = h->rho * std::exp(0.5 * sums[0] / sums[1]);      // rho sqrt(r_prim / r_dual), geometric mean
    rho_new = std::min(std::max(rho_new, h->rho * 1e-3), h->rho * 1e3);             // one update never moves it further than that
    if (!(rho_new > h->rho * thr) && !(rho_new < h->rho / thr)) { if (rho_out) *rho_out = h->rho; return MPCQP_OK; }
    const double rho_old = h->rho;
    const bool had_state = h->has_state;
    mpcqp_factor_opts fo{rho_new, 0, 1};
    TRY(mpcqp_factor(h, &fo, stream));
    h->admm_last = false;               // the residuals on record belong to the old penalty
    if (had_state && h->w && h->Blast > 0) {
        const size_t count = (size_t)h->Blast * h->NP;
        hipLaunchKernelGGL(scale_vec_kernel, dim3((unsigned)std::min<size_t>((count + 255) / 256, 4096)), dim3(256), 0, st, h->w, count, rho_old / h->rho);
        HIPCHK(h, hipGetLastError());
    }
    if (rho_out) *rho_out = h->rho;
    if (refactored) *refactored = 1;
    return MPCQP_OK;
}

int mpcqp_pin(mpcqp_handle h, int on) {
    if (!h) return MPCQP_EINVAL;
    if (!on && h->pinned == 0) { h->err = "pin: not pinned"; return MPCQP_ESTATE; }
    h->pinned += on ? 1 : -1;
    return MPCQP_OK;
}

int mpcqp_set_split(mpcqp_handle h, int enable, double tol) {
    if (!h) return MPCQP_EINVAL;
    if (tol < 0.0 || tol > 1e-6) { h->err = "set_split: tol must be in [0, 1e-6] (0 keeps the current value)"; return MPCQP_EINVAL; }
    h->sp_enable = enable ? 1 : 0;
    if (tol > 0.0) h->sp_tol = tol;
    h->factored = false;        // takes effect with the next factorisation
    return MPCQP_OK;
}

int mpcqp_split_info(mpcqp_handle h, int* components, int* rows_per_component, int* row_map, int row_map_len) {
    if (!h) return MPCQP_EINVAL;
    if (components) *components = h->sp_C;
    if (rows_per_component) *rows_per_component = h->sp_C ? h->sp_ns : 0;
    if (row_map) {
        if (row_map_len < h->n) { h->err = "split_info: row_map needs N*nu entries"; return MPCQP_EINVAL; }
        if (!h->sp_C) { for (int i = 0; i < h->n; ++i) row_map[i] = 0; return MPCQP_OK; }
        HIPCHK(h, hipSetDevice(h->dev));
        HIPCHK(h, hipMemcpy(row_map, h->sp_rowmap, (size_t)h->n * sizeof(int), hipMemcpyDeviceToHost));
        for (int i = 0; i < h->n; ++i) row_map[i] >>= 16;
    }
    return MPCQP_OK;
}

int mpcqp_shift(mpcqp_handle h, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!h->has_state) { h->err = "shift: no solution state yet"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    hipLaunchKernelGGL(shift_kernel, dim3(ceil_div(h->Blast, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h->z, h->w, h->Blast, h->NP, h->n, h->dims.nu, 0);
    HIPCHK(h, hipGetLastError());
    return MPCQP_OK;
}

int mpcqp_shift_recenter(mpcqp_handle h, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!h->has_state) { h->err = "shift: no solution state yet"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    hipLaunchKernelGGL(shift_kernel, dim3(ceil_div(h->Blast, 4)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       h->z, h->w, h->Blast, h->NP, h->n, h->dims.nu, 1);
    HIPCHK(h, hipGetLastError());
    return MPCQP_OK;
}

int mpcqp_ltv_bicycle_advance(mpcqp_handle h, double dt, double L, const double* plan_in, const double* dU, double* plan_out,
                              double* u0, const double* x, double* x_next, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (h->dims.nu != 2 || h->dims.nx != 4) { h->err = "ltv_bicycle_advance: the handle is not a bicycle's (nx 4, nu 2)"; return MPCQP_EINVAL; }
    if (!plan_in || !dU || !plan_out || !u0 || plan_in == plan_out || (!x) != (!x_next) || !(dt > 0.0) || !(L > 0.0)) {
        h->err = "ltv_bicycle_advance: plan_in, dU, plan_out (not aliasing plan_in), u0 are required; x and x_next both or neither"; return MPCQP_EINVAL;
    }
    if (!h->has_state) { h->err = "shift: no solution state yet"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    if (launch_bicycle_advance(h->Blast, h->dims.N, dt, L, plan_in, dU, plan_out, u0, x, x_next, h->z, h->w, h->NP, h->n, h->dims.nu,
                               static_cast<hipStream_t>(stream)) != MPCQP_OK) { h->err = "ltv_bicycle_advance: launch failed"; return MPCQP_EHIP; }
    return MPCQP_OK;
}

int mpcqp_recenter(mpcqp_handle h, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!h->has_state) { h->err = "recenter: no solution state yet"; return MPCQP_ESTATE; }
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipMemsetAsync(h->z, 0, (size_t)h->Blast * h->NP * sizeof(double), static_cast<hipStream_t>(stream)));
    return MPCQP_OK;
}

int mpcqp_get(mpcqp_handle h, int which, int model, double* out, long out_len, void* stream) {
    if (!h || !out) return MPCQP_EINVAL;
    const size_t n = h->n, p = h->p, nx = h->dims.nx;
    if (model < 0 || model >= h->dims.n_models) { h->err = "get: model index out of range"; return MPCQP_EINVAL; }
    const double* src = nullptr; size_t len = 0;
    switch (which) {
        case 0: src = h->H + model * n * n; len = n * n; break;
        case 1: src = h->F + model * n * nx; len = n * nx; break;
        case 2: src = h->f0 + model * n; len = n; break;
        case 3:
            if (!h->assembled) { h->err = "get: assemble first"; return MPCQP_ESTATE; }
            // (one model per instance, structured assembly: the rollout phase alone -- H, F, f0 keep the bits that were factored)
            if (!h->gam_valid) { HIPCHK(h, hipSetDevice(h->dev)); TRY(launch_assemble(h, static_cast<hipStream_t>(stream), true, true)); }
            src = h->Gam + model * p * n; len = p * n; break;
        case 4: src = h->Phi + model * p * nx; len = p * nx; break;
        case 5: src = h->Hs + (h->dims.n_models > 1 ? model * n * n : 0); len = n * n; break;
        case 6: src = h->M; len = n * n; break;
        case 7: src = h->Hi + (h->dims.n_models > 1 ? model * n * n : 0); len = n * n; break;
        case 8: src = h->d + (h->dims.n_models > 1 ? model * n : 0); len = n; break;
        case 9: src = h->rho_dev; len = 3; break;
        case 10: case 11: {
            if (!h->has_state) { h->err = "get: no solution state yet"; return MPCQP_ESTATE; }
            len = (size_t)h->Blast * n;
            if ((size_t)out_len < len) { h->err = "get: output buffer too small"; return MPCQP_EINVAL; }
            HIPCHK(h, hipSetDevice(h->dev));
            hipLaunchKernelGGL(unblock_kernel, dim3((unsigned)((len + 255) / 256)), dim3(256), 0,
                               static_cast<hipStream_t>(stream), which == 10 ? h->z : h->w, out, h->Blast, h->NP, h->n);
            HIPCHK(h, hipGetLastError());
            return MPCQP_OK;
        }
        case 13: case 14: case 15: {
            if (!h->general) { h->err = "get: no general-constraint rows (set_state_constraints + factor first)"; return MPCQP_ESTATE; }
            src = which == 13 ? h->gG : (which == 14 ? h->gE : h->gh0);
            len = which == 13 ? (size_t)h->g_m * n : (which == 14 ? (size_t)h->g_m * nx : (size_t)h->g_m);
            break;
        }
        case 12: {
            if (!h->has_state) { h->err = "get: no solution state yet"; return MPCQP_ESTATE; }
            if (out_len < 96) { h->err = "get: output buffer too small"; return MPCQP_EINVAL; }
            HIPCHK(h, hipSetDevice(h->dev));
            hipLaunchKernelGGL(ints_to_doubles_kernel, dim3(1), dim3(96), 0, static_cast<hipStream_t>(stream),
                               h->it_ws + 5 * (size_t)h->Bcap, out, 96);
            HIPCHK(h, hipGetLastError());
            return MPCQP_OK;
        }
        default: h->err = "get: unknown selector"; return MPCQP_EINVAL;
    }
    if ((size_t)out_len < len) { h->err = "get: output buffer too small"; return MPCQP_EINVAL; }
    HIPCHK(h, hipSetDevice(h->dev));
    HIPCHK(h, hipMemcpyAsync(out, src, len * sizeof(double), hipMemcpyDeviceToDevice, static_cast<hipStream_t>(stream)));
    return MPCQP_OK;
}

}  // extern "C"
