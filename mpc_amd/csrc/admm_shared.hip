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
    double* b0ws;           // [Bcap/16][NT*4][64]: per-wave b0 in accumulator-fragment order (L2 resident)
    double* z;              // [Bcap][NP] row-major (Bcap is a multiple of 64: no bounds checks)
    double* w;
    int* iters;             // [B]
    int* conv;              // [B] 1 if the eps criterion was met
    double* prim;           // [B]
    double* dual;           // [B]
    double *pnorm, *dnorm;  // [B]: max(|x|, |z|) and rho |w| at the last check -- what the residuals are measured against
    int B, nx, SX, dev;
    int k_in_lds;           // 1: Kbfrag, Kufrag are staged behind M in LDS
    int max_iter, min_iter, check_every, warm, stop_on_stable;
    double eps_abs, eps_rel, alpha, rho;
};

constexpr int SXMAX = 9;    // nx + 1 <= 36

typedef unsigned u2 __attribute__((ext_vector_type(2)));

struct Stats {
    double rp = 0.0, rd = 0.0, x = 0.0, z = 0.0, w = 0.0;
    unsigned mask = 0;
};

// ---- building blocks -------------------------------------------------------------------------------
// Per element the ADMM state is (v, c):  v = z - w is the B operand of the next product,
// c = (1 - alpha) z + w is everything the next update needs from the past, so the update
//     t = alpha x + c;  z+ = clip(t);  w+ = t - z+;  v+ = z+ - w+;  c+ = (1 - alpha) z+ + w+
// runs IN PLACE on the accumulator registers of a finished row tile (they become v+), and
// (z, w) are recovered at the end from z = (c + v) / (2 - alpha), w = z - v.

// element-wise update of element (T, R), in place on Rw (raw x -> v+).  CHECK also gathers the
// residuals (needs Vold = the B operand that produced Rw, still intact on a check iteration).
template <int NT, int T, int R, bool CHECK>
__device__ __forceinline__ void post_elem(d4 (&Rw)[NT], const d4 (&Vold)[NT], d4 (&c)[NT], const double2* sB,
                                          double alpha, double oma, double inv2ma, int kq, Stats& st) {
    const int row = 16 * T + 4 * R + kq;
    const double x = Rw[T][R];
    const double co = c[T][R];
    const double tq = alpha * x + co;
    const double2 bd = sB[row];
    const double zn = fmin(fmax(tq, bd.x), bd.y);
    const double wn = tq - zn;
    if (CHECK) {
        const double zo = (co + Vold[T][R]) * inv2ma;
        st.rp = fmax(st.rp, fabs(x - zn));
        st.rd = fmax(st.rd, fabs(zn - zo));
        st.x = fmax(st.x, fabs(x));
        st.z = fmax(st.z, fabs(zn));
        st.w = fmax(st.w, fabs(wn));
        st.mask |= (wn != 0.0 ? 1u : 0u) << (4 * T + R);
    }
    Rw[T][R] = zn - wn;
    c[T][R] = oma * zn + wn;
}

// element-wise update with the bounds already in registers
template <int NT, int T, int R>
__device__ __forceinline__ void post_elem_bd(d4 (&Rw)[NT], d4 (&c)[NT], double2 bd, double alpha, double oma) {
    const double tq = alpha * Rw[T][R] + c[T][R];
    const double zn = fmin(fmax(tq, bd.x), bd.y);
    const double wn = tq - zn;
    Rw[T][R] = zn - wn;
    c[T][R] = oma * zn + wn;
}

// One k-step (s = 4T + R) of  Out = (rho M) In + b0  for all row tiles; In[T][R] is the B operand.
// Software pipeline, pinned with sched_barrier(0) because the compiler otherwise issues every LDS
// read right in front of its MFMA (one wave per SIMD: nothing else hides the LDS latency):
//   [reads: A fragments of step s+1, bounds of element (T+1,R)] | [8 MFMAs of step s] |
//   [PIPE: update of element (T+1,R), ~12 fp64 VALU ops in the shadow of the last MFMA]
template <int NT, int S, int T, int R, bool PIPE>
__device__ __forceinline__ void kstep(d4 (&In)[NT], d4 (&Out)[NT], d4 (&c)[NT], const __amdgpu_buffer_rsrc_t b0,
                                      const double* sM, const double2* sB, double alpha, double oma, int lane,
                                      int kq, double (&af)[2][NT], double2 (&bd)[2]) {
    constexpr int s = 4 * T + R;
    constexpr int cb = s & 1, nb = cb ^ 1;
    constexpr int Tn = (R < 3) ? T : T + 1, Rn = (R < 3) ? R + 1 : 0;    // the next k-step
    if constexpr (s == 0) {
        // the product starts from b0 (this wave's own fragment-ordered copy, L2 resident): keeping
        // it in registers instead costs 64 VGPRs and spills the pipelined loop.  Buffer loads: one
        // lane-offset VGPR + scalar offsets (plain global loads kept a 64-bit address pair per load)
#pragma unroll
        for (int tt = 0; tt < NT; ++tt)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const u2 raw = __builtin_amdgcn_raw_buffer_load_b64(b0, lane * 8, (tt * 4 + q) * 512, 0);
                Out[tt][q] = __builtin_bit_cast(double, raw);
            }
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) af[cb][tt] = sM[(tt * S + s) * 64 + lane];
    }
    if constexpr (s + 1 < S) {
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) af[nb][tt] = sM[(tt * S + s + 1) * 64 + lane];
    }
    if constexpr (PIPE && s == 0 && T + 1 < NT) bd[cb] = sB[16 * (T + 1) + 4 * R + kq];
    if constexpr (PIPE && Tn + 1 < NT) bd[nb] = sB[16 * (Tn + 1) + 4 * Rn + kq];   // one step ahead
    __builtin_amdgcn_sched_barrier(0);
    if constexpr (s < S) {
        const double bf = In[T][R];
#pragma unroll
        for (int tt = 0; tt < NT; ++tt) Out[tt] = mfma_f64(af[cb][tt], bf, Out[tt]);
    }
    if constexpr (PIPE && T + 1 < NT) post_elem_bd<NT, T + 1, R>(In, c, bd[cb], alpha, oma);
    __builtin_amdgcn_sched_barrier(0);
}

template <int NT, int S, int T, bool PIPE>
struct Gemm {
    static __device__ __forceinline__ void run(d4 (&cur)[NT], d4 (&oth)[NT], d4 (&c)[NT], const __amdgpu_buffer_rsrc_t b0,
                                               const double* sM, const double2* sB, double alpha, double oma,
                                               int lane, int kq, double (&af)[2][NT], double2 (&bd)[2]) {
        kstep<NT, S, T, 0, PIPE>(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
        kstep<NT, S, T, 1, PIPE>(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
        kstep<NT, S, T, 2, PIPE>(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
        kstep<NT, S, T, 3, PIPE>(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
        if constexpr (T + 1 < NT) Gemm<NT, S, T + 1, PIPE>::run(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
    }
};

template <int NT, int T, bool CHECK>
struct PostTile {
    static __device__ __forceinline__ void run(d4 (&cur)[NT], const d4 (&oth)[NT], d4 (&c)[NT], const double2* sB,
                                               double alpha, double oma, double inv2ma, int kq, Stats& st) {
        post_elem<NT, T, 0, CHECK>(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
        post_elem<NT, T, 1, CHECK>(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
        post_elem<NT, T, 2, CHECK>(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
        post_elem<NT, T, 3, CHECK>(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
    }
};

template <int NT, int T>
struct PostAll {
    static __device__ __forceinline__ void run(d4 (&cur)[NT], const d4 (&oth)[NT], d4 (&c)[NT], const double2* sB,
                                               double alpha, double oma, double inv2ma, int kq, Stats& st) {
        PostTile<NT, T, true>::run(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
        if constexpr (T + 1 < NT) PostAll<NT, T + 1>::run(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
    }
};

template <int NT, int S>
__global__ __launch_bounds__(256, 1) void admm_shared_box_kernel(AdmmParams p) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NP = 16 * NT;
    // bounds first: their addresses then fit the 16-bit ds offset field off ONE base register
    // (behind the 120 KB of M the compiler keeps one address VGPR per element and spills)
    double2* sB = reinterpret_cast<double2*>(smem);             // NP x {lb, ub}
    double* sM = reinterpret_cast<double*>(smem) + 2 * NP;      // NT*S*64

    {   // stage rho*M and the bounds: 16 B per lane, lane-linear
        const double2* src = reinterpret_cast<const double2*>(p.Mfrag);
        double2* dst = reinterpret_cast<double2*>(sM);
        for (int i = threadIdx.x; i < NT * S * 32; i += 256) dst[i] = src[i];
        for (int i = threadIdx.x; i < NP; i += 256) sB[i] = double2{p.lbs[i], p.ubs[i]};
        if (p.k_in_lds) {   // the two prologue maps too: 2 x NT*SX fragments behind M (one bulk copy instead of
                            // 2 NT SX dependent L2 round trips per wave)
            const double2* ks = reinterpret_cast<const double2*>(p.Kbfrag);
            const double2* us = reinterpret_cast<const double2*>(p.Kufrag);
            double2* kd = reinterpret_cast<double2*>(sM + NT * S * 64);
            const int cnt = NT * p.SX * 32;
            for (int i = threadIdx.x; i < cnt; i += 256) { kd[i] = ks[i]; kd[cnt + i] = us[i]; }
        }
    }
    __syncthreads();
    const double* Kb = p.k_in_lds ? sM + NT * S * 64 : p.Kbfrag;
    const double* Ku = p.k_in_lds ? sM + NT * S * 64 + NT * p.SX * 64 : p.Kufrag;

    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    // 16-instance block of this wave; readfirstlane makes it provably wave-uniform so that the b0
    // workspace is addressed SGPR base + one lane offset instead of one address pair per load
    const int blk = __builtin_amdgcn_readfirstlane(blockIdx.x * 4 + wave);
    if (blk * 16 >= p.B) return;                    // wave-uniform, after the only barrier
    const int col = lane & 15;
    const int kq = lane >> 4;
    const int inst = blk * 16 + col;
    const bool live = inst < p.B;
    const double alpha = p.alpha, oma = 1.0 - p.alpha, inv2ma = 1.0 / (2.0 - p.alpha), rho = p.rho;

    // ---- prologue: b0 and the start point from x0 --------------------------------------
    double xf[SXMAX];
#pragma unroll
    for (int s = 0; s < SXMAX; ++s) {
        const int k = 4 * s + kq;
        double v = 0.0;
        if (live && k < p.nx) v = p.x0[(size_t)inst * p.nx + k];
        if (k == p.nx) v = 1.0;
        xf[s] = v;
    }
    d4 X[NT], Y[NT], c[NT];
    double* b0p = p.b0ws + (size_t)blk * (NT * 4 * 64);
    const __amdgpu_buffer_rsrc_t b0 = __builtin_amdgcn_make_buffer_rsrc(b0p, 0, NT * 4 * 64 * 8, 0x00020000);
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        d4 acc = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < SXMAX; ++s)
            if (s < p.SX) acc = mfma_f64(Kb[(t * p.SX + s) * 64 + lane], xf[s], acc);
#pragma unroll
        for (int q = 0; q < 4; ++q) b0p[(t * 4 + q) * 64 + lane] = acc[q];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    double* zrow = p.z + (size_t)inst * NP;     // instance-major: what the polish reads coalesced
    double* wrow = p.w + (size_t)inst * NP;
    if (p.warm) {
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + kq;
                const double zz = zrow[row], ww = wrow[row];
                Y[t][r] = zz - ww;
                c[t][r] = oma * zz + ww;
            }
    } else {
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            d4 acc = {0, 0, 0, 0};
#pragma unroll
            for (int s = 0; s < SXMAX; ++s)
                if (s < p.SX) acc = mfma_f64(Ku[(t * p.SX + s) * 64 + lane], xf[s], acc);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + 4 * r + kq;
                const double2 bd = sB[row];
                const double z0 = fmin(fmax(acc[r], bd.x), bd.y);
                Y[t][r] = z0;                   // w = 0
                c[t][r] = oma * z0;
            }
        }
    }

    int it = 0;
    double rp = 0.0, rd = 0.0, pnv = 1.0, dnv = 1.0;
    unsigned mask_prev = 0;
    bool have_prev = false, conv_last = false;
    int next_check = p.min_iter > 0 ? p.min_iter : 1;
    bool final_in_X = false;

    if (p.max_iter > 0) {
        // x of iteration 1
        double af[2][NT];
        double2 bd[2];
        Gemm<NT, S, 0, false>::run(Y, X, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
        it = 1;
        // `cur` holds the raw x of iteration `it`, `oth` the v that produced it.  Returns true when done.
        auto step = [&](d4 (&cur)[NT], d4 (&oth)[NT]) -> bool {
            const bool check = (it >= next_check) || (it >= p.max_iter);
            if (!check) {
                Stats dummy;
                PostTile<NT, 0, false>::run(cur, oth, c, sB, alpha, oma, 0.0, kq, dummy);
                Gemm<NT, S, 0, true>::run(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
                ++it;
                return false;
            }
            Stats st;
            PostAll<NT, 0>::run(cur, oth, c, sB, alpha, oma, inv2ma, kq, st);
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
            pnv = fmax(st.x, st.z); dnv = rho * st.w;
            const bool conv = (rp <= p.eps_abs + p.eps_rel * fmax(st.x, st.z)) &&
                              (rd <= p.eps_abs + p.eps_rel * rho * st.w);
            // active-set pattern unchanged since the previous check: good enough for the polish
            const bool stable = p.stop_on_stable && have_prev && (st.mask == mask_prev);
            mask_prev = st.mask;
            have_prev = true;
            conv_last = conv;
            if (__all(conv || stable || !live) || it >= p.max_iter) return true;
            Gemm<NT, S, 0, false>::run(cur, oth, c, b0, sM, sB, alpha, oma, lane, kq, af, bd);
            ++it;
            return false;
        };
        for (;;) {
            if (step(X, Y)) { final_in_X = true; break; }
            if (step(Y, X)) { final_in_X = false; break; }
        }
    }

    // ---- epilogue: keep (z, w) for the polish / warm start ----------------------------
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 16 * t + 4 * r + kq;
            const double v = final_in_X ? X[t][r] : Y[t][r];
            const double z = (c[t][r] + v) * inv2ma;
            zrow[row] = z;
            wrow[row] = z - v;
        }
    if (kq == 0 && live) {
        p.iters[inst] = it;
        p.conv[inst] = conv_last ? 1 : 0;
        p.prim[inst] = rp;
        p.dual[inst] = rd;
        p.pnorm[inst] = pnv; p.dnorm[inst] = dnv;
    }
}

template <int NT, int S>
int launch_one(const AdmmParams& p, hipStream_t st) {
    size_t lds = (size_t)(NT * S * 64 + 2 * 16 * NT) * sizeof(double);
    const size_t klds = (size_t)2 * NT * p.SX * 64 * sizeof(double);
    AdmmParams q = p;
    q.k_in_lds = (lds + klds <= 160 * 1024) ? 1 : 0;
    if (q.k_in_lds) lds += klds;
    const hipError_t e = ensure_dyn_lds(p.dev, reinterpret_cast<const void*>(&admm_shared_box_kernel<NT, S>), lds);
    if (e != hipSuccess) return (int)e;
    const int nblk16 = ceil_div(p.B, 16);
    const int grid = ceil_div(nblk16, 4);
    hipLaunchKernelGGL((admm_shared_box_kernel<NT, S>), dim3(grid), dim3(256), lds, st, q);
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
    p.b0ws = h->b0ws; p.z = h->z; p.w = h->w; p.iters = h->it_ws; p.conv = h->it_ws + 2 * (size_t)h->Bcap; p.prim = h->rp_ws; p.dual = h->rd_ws; p.pnorm = h->rd_ws + h->Bcap; p.dnorm = h->rd_ws + 2 * (size_t)h->Bcap;
    p.B = B; p.nx = h->dims.nx; p.SX = h->SX; p.dev = h->dev;
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
