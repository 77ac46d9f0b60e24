// Scenario physics of BASELINE.json config 4 on the device: the kinematic bicycle's nominal rollout along the
// shifted plan and its Jacobians, plus the deviation-form references and bounds of the next QP -- one thread per
// rollout, so that a closed-loop step is a handful of launches instead of ~300 small tensor ops.
//
// Reference counterpart: the per-iteration model code inside the sequential-QP loops (the finite-horizon rollouts and
// Jacobian blocks of CollocationBasedPlanner2D.py:88-347, SE3Planner.py:136-177); the reference has no bicycle, the
// model is the one SURVEY.md section 8(d) fixes for config 4.
#include "common.h"
#include <algorithm>

namespace {

// One thread per (rollout, stage); R = blockDim.x / N rollouts per workgroup.  The nominal rollout is a chain in (psi, v)
// only through two prefix sums: v_k = v_0 + dt sum a, psi_k = psi_0 + dt/L sum v tan(delta), and (px, py) through two more
// of dt v cos(psi), dt v sin(psi) -- so the transcendentals of all stages are evaluated side by side and only the short sums
// are serial (thread k = 0 of each rollout, N additions each).  The first version walked the N stages in one thread per
// rollout, sin / cos / tan in the chain: 38 us for 8,192 rollouts of 20 stages.
__global__ __launch_bounds__(256) void bicycle_linearize_kernel(int B, int N, int R, double dt, double L, const double* __restrict__ x,
                                         const double* __restrict__ plan, const double* __restrict__ x_ref,
                                         const double* __restrict__ u_min, const double* __restrict__ u_max,
                                         double* __restrict__ xbar, double* __restrict__ A, double* __restrict__ Bm,
                                         double* __restrict__ xref_dev, double* __restrict__ uref_dev,
                                         double* __restrict__ lb, double* __restrict__ ub) {
    extern __shared__ double sm[];
    const int tid = threadIdx.x, rl = tid / N, k = tid - rl * N;
    const int i = blockIdx.x * R + rl;
    const bool on = rl < R && i < B;
    // per rollout: N + 1 values each of v, psi, px, py and N of tan(delta)
    double* sv = sm + (size_t)rl * (5 * (N + 1));
    double* spsi = sv + (N + 1); double* spx = spsi + (N + 1); double* spy = spx + (N + 1); double* st = spy + (N + 1);
    const size_t ik = (size_t)i * N + k;
    double a = 0.0, delta = 0.0, t = 0.0, cd = 1.0;
    if (on) {
        a = plan[ik * 2]; delta = plan[ik * 2 + 1];
        t = tan(delta); cd = cos(delta);
        st[k] = t;
        sv[k + 1] = a;                  // increments; turned into v_k below
    }
    __syncthreads();
    if (on && k == 0) {
        double v = x[4 * i + 3], psi = x[4 * i + 2];
        sv[0] = v; spsi[0] = psi;
        for (int j = 0; j < N; ++j) {
            const double aj = sv[j + 1];
            psi += dt * v / L * st[j];
            v += dt * aj;
            sv[j + 1] = v; spsi[j + 1] = psi;
        }
    }
    __syncthreads();
    double s = 0.0, c = 1.0, v = 0.0;
    if (on) {
        v = sv[k];
        const double psi = spsi[k];
        s = sin(psi); c = cos(psi);
        spx[k + 1] = dt * v * c; spy[k + 1] = dt * v * s;       // increments
    }
    __syncthreads();
    if (on && k == 0) {
        double px = x[4 * i], py = x[4 * i + 1];
        spx[0] = px; spy[0] = py;
        for (int j = 0; j < N; ++j) { px += spx[j + 1]; py += spy[j + 1]; spx[j + 1] = px; spy[j + 1] = py; }
    }
    __syncthreads();
    if (!on) return;
    double* Ak = A + ik * 16;
    double* Bk = Bm + ik * 8;
    Ak[0] = 1.0; Ak[1] = 0.0; Ak[2] = -dt * v * s; Ak[3] = dt * c;
    Ak[4] = 0.0; Ak[5] = 1.0; Ak[6] = dt * v * c;  Ak[7] = dt * s;
    Ak[8] = 0.0; Ak[9] = 0.0; Ak[10] = 1.0;        Ak[11] = dt * t / L;
    Ak[12] = 0.0; Ak[13] = 0.0; Ak[14] = 0.0;      Ak[15] = 1.0;
    Bk[0] = 0.0; Bk[1] = 0.0; Bk[2] = 0.0; Bk[3] = 0.0;
    Bk[4] = 0.0; Bk[5] = dt * v / (L * cd * cd);
    Bk[6] = dt;  Bk[7] = 0.0;
    double* xb = xbar + (size_t)i * (N + 1) * 4;
    if (k == 0) { xb[0] = spx[0]; xb[1] = spy[0]; xb[2] = spsi[0]; xb[3] = sv[0]; }
    const double npx = spx[k + 1], npy = spy[k + 1], npsi = spsi[k + 1], nv = sv[k + 1];
    double* xn = xb + 4 * (k + 1);
    xn[0] = npx; xn[1] = npy; xn[2] = npsi; xn[3] = nv;
    const double* xr = x_ref + ik * 4;
    double* xd = xref_dev + ik * 4;
    xd[0] = xr[0] - npx; xd[1] = xr[1] - npy; xd[2] = xr[2] - npsi; xd[3] = xr[3] - nv;
    uref_dev[ik * 2] = -a; uref_dev[ik * 2 + 1] = -delta;
    lb[ik * 2] = u_min[0] - a; lb[ik * 2 + 1] = u_min[1] - delta;
    ub[ik * 2] = u_max[0] - a; ub[ik * 2 + 1] = u_max[1] - delta;
}

// the serial form, for horizons longer than a workgroup
__global__ void bicycle_linearize_serial_kernel(int B, int N, double dt, double L, const double* __restrict__ x,
                                         const double* __restrict__ plan, const double* __restrict__ x_ref,
                                         const double* __restrict__ u_min, const double* __restrict__ u_max,
                                         double* __restrict__ xbar, double* __restrict__ A, double* __restrict__ Bm,
                                         double* __restrict__ xref_dev, double* __restrict__ uref_dev,
                                         double* __restrict__ lb, double* __restrict__ ub) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= B) return;
    double px = x[4 * i], py = x[4 * i + 1], psi = x[4 * i + 2], v = x[4 * i + 3];
    double* xb = xbar + (size_t)i * (N + 1) * 4;
    xb[0] = px; xb[1] = py; xb[2] = psi; xb[3] = v;
    const double umin0 = u_min[0], umin1 = u_min[1], umax0 = u_max[0], umax1 = u_max[1];
    for (int k = 0; k < N; ++k) {
        const size_t ik = (size_t)i * N + k;
        const double a = plan[ik * 2], delta = plan[ik * 2 + 1];
        const double s = sin(psi), c = cos(psi), t = tan(delta), cd = cos(delta);
        double* Ak = A + ik * 16;
        double* Bk = Bm + ik * 8;
        Ak[0] = 1.0; Ak[1] = 0.0; Ak[2] = -dt * v * s; Ak[3] = dt * c;
        Ak[4] = 0.0; Ak[5] = 1.0; Ak[6] = dt * v * c;  Ak[7] = dt * s;
        Ak[8] = 0.0; Ak[9] = 0.0; Ak[10] = 1.0;        Ak[11] = dt * t / L;
        Ak[12] = 0.0; Ak[13] = 0.0; Ak[14] = 0.0;      Ak[15] = 1.0;
        Bk[0] = 0.0; Bk[1] = 0.0; Bk[2] = 0.0; Bk[3] = 0.0;
        Bk[4] = 0.0; Bk[5] = dt * v / (L * cd * cd);
        Bk[6] = dt;  Bk[7] = 0.0;
        const double npx = px + dt * v * c, npy = py + dt * v * s, npsi = psi + dt * v / L * t, nv = v + dt * a;
        px = npx; py = npy; psi = npsi; v = nv;
        double* xn = xb + 4 * (k + 1);
        xn[0] = px; xn[1] = py; xn[2] = psi; xn[3] = v;
        const double* xr = x_ref + ik * 4;
        double* xd = xref_dev + ik * 4;
        xd[0] = xr[0] - px; xd[1] = xr[1] - py; xd[2] = xr[2] - psi; xd[3] = xr[3] - v;
        uref_dev[ik * 2] = -a; uref_dev[ik * 2 + 1] = -delta;
        lb[ik * 2] = umin0 - a; lb[ik * 2 + 1] = umin1 - delta;
        ub[ik * 2] = umax0 - a; ub[ik * 2 + 1] = umax1 - delta;
    }
}

__device__ __forceinline__ void bicycle_f(const double* x, double a, double delta, double dt, double L, double* xn) {
    const double px = x[0], py = x[1], psi = x[2], v = x[3];
    xn[0] = px + dt * v * cos(psi); xn[1] = py + dt * v * sin(psi); xn[2] = psi + dt * v / L * tan(delta); xn[3] = v + dt * a;
}

// apply + shift in one launch: plan = plan_in + dU; u0 = its first stage; plan_out = the plan moved up one stage with the last
// one repeated; optionally the plant itself one step on with u0.  One thread per (rollout, stage).
__global__ __launch_bounds__(256) void bicycle_apply_kernel(int B, int N, double dt, double L, const double* __restrict__ plan_in,
                                                            const double* __restrict__ dU, double* __restrict__ plan_out,
                                                            double* __restrict__ u0, const double* __restrict__ x,
                                                            double* __restrict__ x_next) {
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * N) return;
    const int i = (int)(e / N), k = (int)(e - (long)i * N);
    const int ks = k < N - 1 ? k + 1 : N - 1;
    const size_t src = ((size_t)i * N + ks) * 2, dst = ((size_t)i * N + k) * 2;
    plan_out[dst] = plan_in[src] + dU[src];
    plan_out[dst + 1] = plan_in[src + 1] + dU[src + 1];
    if (k == 0) {
        const size_t f = (size_t)i * N * 2;
        const double a = plan_in[f] + dU[f], delta = plan_in[f + 1] + dU[f + 1];
        u0[2 * i] = a; u0[2 * i + 1] = delta;
        if (x && x_next) bicycle_f(x + 4 * (size_t)i, a, delta, dt, L, x_next + 4 * (size_t)i);
    }
}

// bicycle_apply_kernel and the shift + re-centring of the handle's kept (z, w) (api.hip: shift_kernel with zero_z) as ONE
// launch: what a re-linearising closed loop does between two solves (mpcqp_ltv_bicycle_advance).  Block b: the shift of
// instances 4 b .. 4 b + 3 (one wave each), then elements 256 b .. of the plan.
__global__ __launch_bounds__(256) void bicycle_advance_kernel(int B, int N, double dt, double L, const double* __restrict__ plan_in,
                                                              const double* __restrict__ dU, double* __restrict__ plan_out,
                                                              double* __restrict__ u0, const double* __restrict__ x,
                                                              double* __restrict__ x_next, double* __restrict__ z,
                                                              double* __restrict__ w, int NP, int n, int nu) {
    {
        const int inst = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
        if (inst < B) {
            double* zb = z + (size_t)inst * NP;
            double* wb = w + (size_t)inst * NP;
            double wv[2];
            for (int q = 0; q < 2; ++q) {
                const int r = lane + 64 * q;
                const int src = (r + nu < n) ? r + nu : r;
                wv[q] = r < n ? wb[src] : 0.0;
            }
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
            for (int q = 0; q < 2; ++q) {
                const int r = lane + 64 * q;
                if (r < n) { zb[r] = 0.0; wb[r] = wv[q]; }
            }
        }
    }
    const long e = (long)blockIdx.x * 256 + threadIdx.x;
    if (e >= (long)B * N) return;
    const int i = (int)(e / N), k = (int)(e - (long)i * N);
    const int ks = k < N - 1 ? k + 1 : N - 1;
    const size_t src = ((size_t)i * N + ks) * 2, dst = ((size_t)i * N + k) * 2;
    plan_out[dst] = plan_in[src] + dU[src];
    plan_out[dst + 1] = plan_in[src + 1] + dU[src + 1];
    if (k == 0) {
        const size_t f = (size_t)i * N * 2;
        const double a = plan_in[f] + dU[f], delta = plan_in[f + 1] + dU[f + 1];
        u0[2 * i] = a; u0[2 * i + 1] = delta;
        if (x && x_next) bicycle_f(x + 4 * (size_t)i, a, delta, dt, L, x_next + 4 * (size_t)i);
    }
}

__global__ __launch_bounds__(256) void bicycle_step_kernel(int B, double dt, double L, const double* __restrict__ x,
                                                           const double* __restrict__ u, double* __restrict__ x_next) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= B) return;
    bicycle_f(x + 4 * (size_t)i, u[2 * i], u[2 * i + 1], dt, L, x_next + 4 * (size_t)i);
}

}  // namespace

extern "C" int mpcqp_ltv_bicycle(int device_id, int batch, int N, double dt, double L, const double* x,
                                 const double* plan, const double* x_ref, const double* u_min, const double* u_max,
                                 double* xbar, double* A, double* B, double* xref_dev, double* uref_dev,
                                 double* lb, double* ub, void* stream) {
    if (batch < 1 || N < 1 || !x || !plan || !x_ref || !u_min || !u_max || !xbar || !A || !B || !xref_dev ||
        !uref_dev || !lb || !ub || !(dt > 0.0) || !(L > 0.0)) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (N <= 128) {
        const int R = 256 / N;
        const size_t lds = (size_t)R * 5 * (N + 1) * sizeof(double);
        hipLaunchKernelGGL(bicycle_linearize_kernel, dim3(ceil_div(batch, R)), dim3(256), lds, st,
                           batch, N, R, dt, L, x, plan, x_ref, u_min, u_max, xbar, A, B, xref_dev, uref_dev, lb, ub);
    } else
        hipLaunchKernelGGL(bicycle_linearize_serial_kernel, dim3(ceil_div(batch, 64)), dim3(64), 0, st,
                           batch, N, dt, L, x, plan, x_ref, u_min, u_max, xbar, A, B, xref_dev, uref_dev, lb, ub);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

// apply the solution of the increment QP and move the plan on by one stage (ClosedLoopLTV.step): plan_out may not alias
// plan_in; x / x_next (both or neither): the bicycle itself one step on under the applied input
extern "C" int mpcqp_ltv_bicycle_apply(int device_id, int batch, int N, double dt, double L, const double* plan_in,
                                       const double* dU, double* plan_out, double* u0, const double* x, double* x_next,
                                       void* stream) {
    if (batch < 1 || N < 1 || !plan_in || !dU || !plan_out || !u0 || plan_in == plan_out || (!x) != (!x_next) ||
        !(dt > 0.0) || !(L > 0.0)) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipLaunchKernelGGL(bicycle_apply_kernel, dim3((unsigned)(((long)batch * N + 255) / 256)), dim3(256), 0,
                       static_cast<hipStream_t>(stream), batch, N, dt, L, plan_in, dU, plan_out, u0, x, x_next);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

// api.hip (mpcqp_ltv_bicycle_advance): the launch itself; the handle's state is checked there
int launch_bicycle_advance(int B, int N, double dt, double L, const double* plan_in, const double* dU, double* plan_out, double* u0,
                           const double* x, double* x_next, double* z, double* w, int NP, int n, int nu, hipStream_t st) {
    const long blocks = std::max<long>(((long)B * N + 255) / 256, (B + 3) / 4);
    hipLaunchKernelGGL(bicycle_advance_kernel, dim3((unsigned)blocks), dim3(256), 0, st, B, N, dt, L, plan_in, dU, plan_out, u0, x, x_next,
                       z, w, NP, n, nu);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}

extern "C" int mpcqp_ltv_bicycle_step(int device_id, int batch, double dt, double L, const double* x, const double* u,
                                      double* x_next, void* stream) {
    if (batch < 1 || !x || !u || !x_next || !(dt > 0.0) || !(L > 0.0)) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipLaunchKernelGGL(bicycle_step_kernel, dim3(ceil_div(batch, 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       batch, dt, L, x, u, x_next);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}
