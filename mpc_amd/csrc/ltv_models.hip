// Scenario physics of BASELINE.json config 4 on the device: the kinematic bicycle's nominal rollout along the
// shifted plan and its Jacobians, plus the deviation-form references and bounds of the next QP -- one thread per
// rollout, so that a closed-loop step is a handful of launches instead of ~300 small tensor ops.
//
// Reference counterpart: the per-iteration model code inside the sequential-QP loops (the finite-horizon rollouts and
// Jacobian blocks of CollocationBasedPlanner2D.py:88-347, SE3Planner.py:136-177); the reference has no bicycle, the
// model is the one SURVEY.md section 8(d) fixes for config 4.
#include "common.h"

namespace {

__global__ void bicycle_linearize_kernel(int B, int N, double dt, double L, const double* __restrict__ x,
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

}  // namespace

extern "C" int mpcqp_ltv_bicycle(int device_id, int batch, int N, double dt, double L, const double* x,
                                 const double* plan, const double* x_ref, const double* u_min, const double* u_max,
                                 double* xbar, double* A, double* B, double* xref_dev, double* uref_dev,
                                 double* lb, double* ub, void* stream) {
    if (batch < 1 || N < 1 || !x || !plan || !x_ref || !u_min || !u_max || !xbar || !A || !B || !xref_dev ||
        !uref_dev || !lb || !ub || !(dt > 0.0) || !(L > 0.0)) return MPCQP_EINVAL;
    if (hipSetDevice(device_id) != hipSuccess) return MPCQP_EHIP;
    hipLaunchKernelGGL(bicycle_linearize_kernel, dim3(ceil_div(batch, 64)), dim3(64), 0, static_cast<hipStream_t>(stream),
                       batch, N, dt, L, x, plan, x_ref, u_min, u_max, xbar, A, B, xref_dev, uref_dev, lb, ub);
    return hipGetLastError() == hipSuccess ? MPCQP_OK : MPCQP_EHIP;
}
