// Trajectory recovery: the states x_1..x_N that the optimal inputs produce, for the whole batch.
//
// Reference counterpart: `x_traj = Gamma u* + affine terms` after the solve (LaunchPlannerQPFormulation.py:176-183,
// Launch3DPlanner.py:202-211).  On the device the prediction matrix is not needed: every instance simply runs its
// own model forward, x_{k+1} = A_k x_k + B_k u_k -- 2 N nx (nx + nu) flops per instance instead of 2 N nx n, and it
// also covers one model per instance.  Bytes per instance: 8 (nx + N nu) in, 8 N nx out: HBM-streaming.
//
// Mapping: NXP = next power of two >= nx lanes per instance (lane r owns state r), 64 / NXP instances per wave;
// the row of A (and of B) a lane needs stays in registers for LTI models; x_j of the other lanes comes by ds_bpermute.
#include "common.h"

namespace {

struct PredParams {
    const double *A, *B, *x0, *U;
    double* X;
    int batch, nx, nu, N, tv;
    long model_stride_A, model_stride_B;     // 0: shared model
};

template <int NXP>
__global__ __launch_bounds__(256) void predict_kernel(PredParams p) {
    constexpr int IPW = 64 / NXP;                       // instances per wave
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int r = lane & (NXP - 1), gi = lane / NXP;
    const int inst = (blockIdx.x * 4 + wave) * IPW + gi;
    const bool live = inst < p.batch && r < p.nx;
    const int ii = inst < p.batch ? inst : p.batch - 1;
    const int nx = p.nx, nu = p.nu;
    const double* Am = p.A + (size_t)ii * p.model_stride_A;
    const double* Bm = p.B + (size_t)ii * p.model_stride_B;
    double a[NXP];
    auto load_a = [&](const double* Ak) {
#pragma unroll
        for (int j = 0; j < NXP; ++j) a[j] = (live && j < nx) ? Ak[r * nx + j] : 0.0;
    };
    if (!p.tv) load_a(Am);
    double x = live ? p.x0[(size_t)ii * nx + r] : 0.0;
    const int base = lane & ~(NXP - 1);
    for (int k = 0; k < p.N; ++k) {
        if (p.tv) load_a(Am + (size_t)k * nx * nx);
        const double* Bk = Bm + (p.tv ? (size_t)k * nx * nu : 0);
        const double* uk = p.U + ((size_t)ii * p.N + k) * nu;
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < NXP; ++j) s += a[j] * __shfl(x, base + j, 64);
        if (live)
            for (int j = 0; j < nu; ++j) s += Bk[r * nu + j] * uk[j];
        x = s;
        if (live) p.X[((size_t)inst * p.N + k) * nx + r] = x;
    }
}

template <int NXP>
int launch_pred(const PredParams& p, hipStream_t st) {
    const int per_block = 4 * (64 / NXP);
    hipLaunchKernelGGL(predict_kernel<NXP>, dim3(ceil_div(p.batch, per_block)), dim3(256), 0, st, p);
    return (int)hipGetLastError();
}

}  // namespace

extern "C" int mpcqp_predict(mpcqp_handle h, const double* x0, const double* U, int batch, double* X, void* stream) {
    if (!h) return MPCQP_EINVAL;
    if (!x0 || !U || !X || batch < 1) { h->err = "predict: x0, U, X must be non-null and batch >= 1"; return MPCQP_EINVAL; }
    if (!h->A || !h->B) { h->err = "predict: set_model first"; return MPCQP_ESTATE; }
    if (h->dims.n_models > 1 && batch != h->dims.n_models) { h->err = "predict: batch must equal n_models"; return MPCQP_EINVAL; }
    if (h->dims.nx > 64) { h->err = "predict: nx > 64 is not implemented in this build"; return MPCQP_ENOTIMPL; }
    HIPCHK(h, hipSetDevice(h->dev));
    PredParams p;
    p.A = h->A; p.B = h->B; p.x0 = x0; p.U = U; p.X = X;
    p.batch = batch; p.nx = h->dims.nx; p.nu = h->dims.nu; p.N = h->dims.N; p.tv = h->dims.time_varying;
    const long per = h->dims.time_varying ? h->dims.N : 1;
    p.model_stride_A = h->dims.n_models > 1 ? per * p.nx * p.nx : 0;
    p.model_stride_B = h->dims.n_models > 1 ? per * p.nx * p.nu : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    int rc;
    if (p.nx <= 4) rc = launch_pred<4>(p, st);
    else if (p.nx <= 8) rc = launch_pred<8>(p, st);
    else if (p.nx <= 16) rc = launch_pred<16>(p, st);
    else if (p.nx <= 32) rc = launch_pred<32>(p, st);
    else rc = launch_pred<64>(p, st);
    if (rc != 0) { h->err = std::string("predict launch: ") + hipGetErrorString((hipError_t)rc); return MPCQP_EHIP; }
    return MPCQP_OK;
}
