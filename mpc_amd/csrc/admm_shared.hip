// ADMM for box-constrained condensed-MPC QPs that share one Hessian (LTI / shared LTV model,
// only x0 differs between instances) -- the "shared-H" regime of SURVEY.md section 7.
//
// Replaces, for the whole batch at once, the interior-point iterations inside
// cvxopt.solvers.qp at LaunchPlannerQPFormulation.py:173 / Launch3DPlanner.py:201 for QPs whose
// only inequalities are the input box rows of LaunchPlannerQPFormulation.py:96-107.
//
// Design (gfx950): because (Hs + rho I)^-1 is the same for every instance, one ADMM iteration of a
// 16-instance block is the 16-column GEMM  X = (rho M) V + B0  followed by element-wise clipping.
// One wave owns 16 instances and ALL rows: with v_mfma_f64_16x16x4_f64 the accumulator registers
// of row tile t, register r are exactly the B operand of k-step 4t+r of the next iteration's
// product (see common.h), so the iterates never leave registers -- no LDS traffic, no barrier in
// the loop.  rho*M sits in LDS once per workgroup, stored in A-fragment order so that every
// ds_read_b64 is lane-linear (conflict free).  HBM traffic per solve: x0 in, (z, w) out.
#include "common.h"

namespace {

struct AdmmParams {
    const double* Mfrag;    // [NT*S][64]   rho*M, A-fragment order
    const double* Kbfrag;   // [NT*SX][64]  [Kb | kb]:  b0 = -M qs = Kb x0 + kb
    const double* Kufrag;   // [NT*SX][64]  [Ku | ku]:  x_unc = Ku x0 + ku
    const double* lbs;      // [NP]
    const double* ubs;      // [NP]
    const double* x0;       // [B][nx]
    double* z;              // [Bcap][NP] row-major (Bcap is a multiple of 64: no bounds checks)
    double* w;
    int* iters;             // [B]
    int* conv;              // [B] 1 if the eps criterion was met
    double* prim;           // [B]
    double* dual;           // [B]
    int B, nx, SX;
    int max_iter, min_iter, check_every, warm, stop_on_stable;
    double eps_abs, eps_rel, alpha, rho;
};

constexpr int SXMAX = 9;    // nx + 1 <= 36

struct Stats {
    double rp = 0.0, rd = 0.0, x = 0.0, z = 0.0, w = 0.0;
    unsigned mask = 0;
};

// One ADMM iteration for the 16 instances of a wave.  v = z - w (the next product's B operand),
// w = scaled dual.  X = (rho M) V + b0 is accumulated two row tiles at a time so that dependent
// MFMAs are two issues apart; the element-wise update of a finished pair is independent of the
// next pair's MFMAs (it writes vn, never v).
template <int NT, int S, bool CHECK>
__device__ __forceinline__ void admm_iteration(d4 (&v)[NT], d4 (&w)[NT], const d4 (&b0)[NT],
                                               const double* sM, const double* sLb, const double* sUb,
                                               double alpha, double oma, int lane, int kq, Stats& st) {
    d4 vn[NT];
#pragma unroll
    for (int t0 = 0; t0 < NT; t0 += 2) {
        d4 acc0 = b0[t0];
        d4 acc1 = (t0 + 1 < NT) ? b0[t0 + 1 < NT ? t0 + 1 : t0] : d4{0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < S; ++s) {
            const double bf = v[s >> 2][s & 3];
            acc0 = mfma_f64(sM[(t0 * S + s) * 64 + lane], bf, acc0);
            if (t0 + 1 < NT) acc1 = mfma_f64(sM[((t0 + 1) * S + s) * 64 + lane], bf, acc1);
        }
#pragma unroll
        for (int tt = 0; tt < 2; ++tt) {
            const int t = t0 + tt;
            if (t < NT) {
                const d4 acc = tt ? acc1 : acc0;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * t + 4 * r + kq;
                    const double x = acc[r];
                    const double wo = w[t][r];
                    const double zo = v[t][r] + wo;
                    const double tq = alpha * x + oma * zo + wo;
                    const double zn = fmin(fmax(tq, sLb[row]), sUb[row]);
                    const double wn = tq - zn;
                    w[t][r] = wn;
                    vn[t][r] = zn - wn;
                    if (CHECK) {
                        st.rp = fmax(st.rp, fabs(x - zn));
                        st.rd = fmax(st.rd, fabs(zn - zo));
                        st.x = fmax(st.x, fabs(x));
                        st.z = fmax(st.z, fabs(zn));
                        st.w = fmax(st.w, fabs(wn));
                        st.mask |= (wn != 0.0 ? 1u : 0u) << (4 * t + r);
                    }
                }
            }
        }
    }
#pragma unroll
    for (int t = 0; t < NT; ++t) v[t] = vn[t];
}

template <int NT, int S>
__global__ __launch_bounds__(256, 1) void admm_shared_box_kernel(AdmmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = 16 * NT;
    double* sM = reinterpret_cast<double*>(smem);   // NT*S*64
    double* sLb = sM + NT * S * 64;                 // NP
    double* sUb = sLb + NP;                         // NP

    {   // stage rho*M and the bounds: 16 B per lane, lane-linear
        const double2* src = reinterpret_cast<const double2*>(p.Mfrag);
        double2* dst = reinterpret_cast<double2*>(sM);
        for (int i = threadIdx.x; i < NT * S * 32; i += 256) dst[i] = src[i];
        for (int i = threadIdx.x; i < NP; i += 256) { sLb[i] = p.lbs[i]; sUb[i] = p.ubs[i]; }
    }
    __syncthreads();

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int blk = blockIdx.x * 4 + wave;          // 16-instance block of this wave
    if (blk * 16 >= p.B) return;                    // wave-uniform, after the only barrier
    const int col = lane & 15;
    const int kq = lane >> 4;
    const int inst = blk * 16 + col;
    const bool live = inst < p.B;

    // ---- prologue: b0 and the cold start from x0 -------------------------------------
    double xf[SXMAX];
#pragma unroll
    for (int s = 0; s < SXMAX; ++s) {
        const int k = 4 * s + kq;
        double v = 0.0;
        if (live && k < p.nx) v = p.x0[(size_t)inst * p.nx + k];
        if (k == p.nx) v = 1.0;
        xf[s] = v;
    }
    d4 b0[NT], v[NT], w[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < SXMAX; ++s)
            if (s < p.SX) acc = mfma_f64(p.Kbfrag[(size_t)(t * p.SX + s) * 64 + lane], xf[s], acc);
        b0[t] = acc;
    }
    double* zrow = p.z + (size_t)inst * NP;     // instance-major: what the polish reads coalesced
    double* wrow = p.w + (size_t)inst * NP;
    if (p.warm) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + kq;
                const double zz = zrow[row], ww = wrow[row];
                w[t][r] = ww;
                v[t][r] = zz - ww;
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < SXMAX; ++s)
                if (s < p.SX) acc = mfma_f64(p.Kufrag[(size_t)(t * p.SX + s) * 64 + lane], xf[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + kq;
                v[t][r] = fmin(fmax(acc[r], sLb[row]), sUb[row]);
                w[t][r] = 0.0;
            }
        }
    }

    const double alpha = p.alpha, oma = 1.0 - p.alpha, rho = p.rho;
    int it = 0;
    double rp = 0.0, rd = 0.0;
    unsigned mask_prev = 0;
    bool have_prev = false, conv_last = false;
    int next_check = p.min_iter > 0 ? p.min_iter : 1;

    while (it < p.max_iter) {
        ++it;
        const bool check = (it >= next_check) || (it == p.max_iter);
        if (!check) {
            Stats st;
            admm_iteration<NT, S, false>(v, w, b0, sM, sLb, sUb, alpha, oma, lane, kq, st);
            continue;
        }
        Stats st;
        admm_iteration<NT, S, true>(v, w, b0, sM, sLb, sUb, alpha, oma, lane, kq, st);
        next_check = it + p.check_every;
        // the 4 lanes {c, c+16, c+32, c+48} hold one instance
#pragma unroll
        for (int off = 16; off < 64; off <<= 1) {
            st.rp = fmax(st.rp, __shfl_xor(st.rp, off));
            st.rd = fmax(st.rd, __shfl_xor(st.rd, off));
            st.x = fmax(st.x, __shfl_xor(st.x, off));
            st.z = fmax(st.z, __shfl_xor(st.z, off));
            st.w = fmax(st.w, __shfl_xor(st.w, off));
        }
        rp = st.rp;
        rd = rho * st.rd;
        const bool conv = (rp <= p.eps_abs + p.eps_rel * fmax(st.x, st.z)) &&
                          (rd <= p.eps_abs + p.eps_rel * rho * st.w);
        // active-set pattern unchanged since the previous check: good enough for the polish
        const bool stable = p.stop_on_stable && have_prev && (st.mask == mask_prev);
        mask_prev = st.mask;
        have_prev = true;
        conv_last = conv;
        if (__all(conv || stable || !live)) break;
    }

    // ---- epilogue: keep (z, w) for the polish / warm start ----------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * r + kq;
            zrow[row] = v[t][r] + w[t][r];
            wrow[row] = w[t][r];
        }
    if (kq == 0 && live) {
        p.iters[inst] = it;
        p.conv[inst] = conv_last ? 1 : 0;
        p.prim[inst] = rp;
        p.dual[inst] = rd;
    }
}

template <int NT, int S>
int launch_one(const AdmmParams& p, hipStream_t st) {
    const size_t lds = (size_t)(NT * S * 64 + 2 * 16 * NT) * sizeof(double);
    static bool attr_set = false;
    if (!attr_set) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&admm_shared_box_kernel<NT, S>),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (e != hipSuccess) return (int)e;
        attr_set = true;
    }
    const int nblk16 = ceil_div(p.B, 16);
    const int grid = ceil_div(nblk16, 4);
    hipLaunchKernelGGL((admm_shared_box_kernel<NT, S>), dim3(grid), dim3(256), lds, st, p);
    return (int)hipGetLastError();
}

// S in [4NT-3, 4NT] for NT = 1..8 (n <= 128)
#define FOR_S(NT_, X) X(NT_, 4 * NT_ - 3) X(NT_, 4 * NT_ - 2) X(NT_, 4 * NT_ - 1) X(NT_, 4 * NT_)
#define FOR_ALL(X) FOR_S(1, X) FOR_S(2, X) FOR_S(3, X) FOR_S(4, X) FOR_S(5, X) FOR_S(6, X) FOR_S(7, X) FOR_S(8, X)

}  // namespace

bool admm_shared_box_supported(int NT, int S) {
    return NT >= 1 && NT <= 8 && S >= 4 * NT - 3 && S <= 4 * NT && S >= 1;
}

int launch_admm_shared_box(mpcqp_handle h, const double* x0, int B, const mpcqp_solve_opts& o, hipStream_t st) {
    AdmmParams p;
    p.Mfrag = h->Mfrag; p.Kbfrag = h->Kbfrag; p.Kufrag = h->Kufrag;
    p.lbs = h->lbs; p.ubs = h->ubs; p.x0 = x0;
    p.z = h->z; p.w = h->w; p.iters = h->it_ws; p.conv = h->it_ws + 2 * (size_t)h->Bcap; p.prim = h->rp_ws; p.dual = h->rd_ws;
    p.B = B; p.nx = h->dims.nx; p.SX = h->SX;
    p.max_iter = o.max_iter; p.min_iter = o.min_iter; p.check_every = o.check_every > 0 ? o.check_every : 1;
    p.warm = (o.warm_start && h->has_state) ? 1 : 0;
    p.stop_on_stable = o.polish ? 1 : 0;
    p.eps_abs = o.eps_abs; p.eps_rel = o.eps_rel; p.alpha = o.alpha; p.rho = h->rho;
    if (h->SX > SXMAX) { h->err = "nx too large for this build (nx + 1 <= 36)"; return MPCQP_ENOTIMPL; }
    int rc = -1;
#define DISPATCH(NT_, S_) if (h->NT == NT_ && h->S == S_) rc = launch_one<NT_, S_>(p, st);
    FOR_ALL(DISPATCH)
#undef DISPATCH
    if (rc == -1) { h->err = "shared-H ADMM kernel: N*nu > 128 not instantiated in this build"; return MPCQP_ENOTIMPL; }
    if (rc != 0) { h->err = std::string("admm launch: ") + hipGetErrorString((hipError_t)rc); return MPCQP_EHIP; }
    return MPCQP_OK;
}
