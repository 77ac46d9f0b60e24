// Active-set stage, fast path ("polish" in the names: it began as the step that made an ADMM iterate exact and is now the
// whole default solve): the same dual active-set method as polish.hip (Goldfarb-Idnani on the Schur complement
// S = Hi[A,A] of the shared inverse Hi = Hs^-1), started from the EMPTY set (cold, the default) or from the guess a kept /
// ADMM iterate (z, w) provides (warm), laid out for what the workload looks like.  On condensed MPC QPs the optimal active
// set is small (config 3, measured: mean 7.4 of 120 bounds, 97.7 % of instances <= 16, 99.98 % <= 24), so the
// per-instance arithmetic is a few thousand flops and a one-wave-per-instance kernel would spend its time issuing 64-wide
// instructions for 7-element vectors and walking Cholesky columns one by one.
//
// Mapping: G lanes per instance, 512-thread workgroups; the cascade is G = 8 / 16 slots (every instance) -> G = 32 or 16 /
// 32 slots (what outgrew the first; 16 lanes when that queue is long) -> G = 64 / 64 slots -> polish.hip's any-size kernel,
// each pass handing its whole working state to the next.  A cold batch of 8,192 instances or fewer is ONE pass with
// G = 16 / 32 slots (launch_polish_quad).
//   * Hi (n x n, odd row stride) is staged ONCE per workgroup into LDS and serves all its instances;
//     it is symmetric, so "column p restricted to my rows" is a contiguous row read.
//   * lane g of the group owns the row pairs (2 G j + 2 g, + 1), j < RP, of x in registers.
//   * Instead of a Cholesky factor the group keeps W = S^-1 itself (k <= KMAX), rows G i + g in lane g,
//     embedded in a KMAX x KMAX matrix whose unused rows/columns are zero.  Every update is then a
//     matrix-vector product or a rank-one update with compile-time register indices and no
//     column-by-column dependency chain:
//         add p   :  r = W h,  d2 = Hi[p,p] - h'r,  W += u u'/d2  with u = [r; -1 at the new slot]
//         drop j  :  W -= W[:,j] W[j,:] / W[j,j], then row/column j are zeroed and the slot is free
//     (block-inverse / Sherman-Morrison forms of what polish.hip does with its factor; S is a
//     principal submatrix of Hi, cond <= cond(Hs), and there are at most a few dozen updates).
//     Vectors a whole group needs (r, W[j,:], the row list) are exchanged through a per-instance LDS
//     scratch; scalars are reduced with DPP (quad_perm, row_half_mirror, row_ror) and one ds_bpermute for G = 32.
//     "Which row / which slot" reductions carry the index in the low mantissa bits of the value (key_pack): one v_max_f64 or
//     v_min_f64 per butterfly step.  Where a vector held one element per lane meets per-lane rows (r = W h; the rank-one
//     update for G = 16) the 64-bit FMA takes the element through the DPP row broadcast (fmac_grp) instead of LDS.
//   * the instances of a wave advance in lock-step (one add/drop per round, predicated).
// Instances whose active set outgrows KMAX (or that hit a numerical guard) are queued on the device for the next,
// larger pass; the last resort is the wave-per-instance kernel (polish.hip).
//
// Reference counterpart: as polish.hip -- the accuracy of soln['x'] (LaunchPlannerQPFormulation.py:174).
#include <type_traits>
#include "polish.h"
#include <cstdlib>
#include <algorithm>

#ifndef OCT_CH32
#define OCT_CH32 4
#define OCT_XCH32 2
#endif

namespace {

constexpr int DPP_XOR1 = 0xB1, DPP_XOR2 = 0x4E;     // quad_perm [1,0,3,2], [2,3,0,1]
constexpr int DPP_HMIRROR = 0x141;                  // row_half_mirror: lane i <-> 7-i of each 8

// The lane exchanges below are all inside a row of 16 lanes and run with every lane of the wave active, so the value a
// disabled or out-of-row source would leave (bound_ctrl: zero) never shows; without an "old" operand the compiler emits
// the two v_mov_b32_dpp alone (with update_dpp(v, v, ..) it copied the pair first: 5 instructions a step, not 2).
template <int CTRL>
__device__ __forceinline__ int dpp_i(int v) { return __builtin_amdgcn_mov_dpp(v, CTRL, 0xF, 0xF, true); }
// v_max_f64 / v_min_f64 as they are: fmax() / fmin() first canonicalise every operand the compiler cannot prove quiet
// (one more v_max_f64 each); the keys below are bit patterns put together by integer instructions, never NaN
__device__ __forceinline__ double fmax_raw(double a, double b) { double r; asm("v_max_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ double fmin_raw(double a, double b) { double r; asm("v_min_f64 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
template <int CTRL>
__device__ __forceinline__ double dpp_d(double v) {
    return __hiloint2double(dpp_i<CTRL>(__double2hiint(v)), dpp_i<CTRL>(__double2loint(v)));
}
__device__ __forceinline__ void wsync() {
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    __builtin_amdgcn_wave_barrier();
}
constexpr int DPP_ROR8 = 0x128;                     // row_ror:8: lane i <-> i+8 of each 16
// sum over the G (4 .. 64) lanes of a group, result in all of them
template <int G>
__device__ __forceinline__ double grp_sum(double v) {
    v += dpp_d<DPP_XOR1>(v);
    v += dpp_d<DPP_XOR2>(v);
    if (G >= 8) v += dpp_d<DPP_HMIRROR>(v);
    if (G >= 16) v += dpp_d<DPP_ROR8>(v);
    if (G >= 32) v += __shfl_xor(v, 16);        // across the DPP rows of the group
    if (G == 64) v += __shfl_xor(v, 32);
    return v;
}
template <int CTRL>
__device__ __forceinline__ void imin_step(int& j) { const int oj = dpp_i<CTRL>(j); j = oj < j ? oj : j; }
template <int G>
__device__ __forceinline__ int grp_imin(int j) {
    imin_step<DPP_XOR1>(j); imin_step<DPP_XOR2>(j);
    if (G >= 8) imin_step<DPP_HMIRROR>(j);
    if (G >= 16) imin_step<DPP_ROR8>(j);
    if (G >= 32) { const int oj = __shfl_xor(j, 16); j = oj < j ? oj : j; }
    if (G == 64) { const int oj = __shfl_xor(j, 32); j = oj < j ? oj : j; }
    return j;
}
constexpr int BIGI = 1 << 20;

// Reductions that carry an index in the value.  A positive double orders like its bit pattern, so with the low KEYBITS bits
// of the mantissa replaced by an index the maximum (minimum) of the KEYS is the value AND its owner in one v_max_f64
// (v_min_f64) per step of the butterfly -- no comparison chain, no second register riding along.  The key differs from the
// value by < 2^-42 of it: far inside what "most violated row" or "first blocking multiplier" mean, and the quantities that
// are derived from it (a step length) are restored to full accuracy by the refinement every solve ends with.
constexpr int KEYBITS = 10;
__device__ __forceinline__ double key_pack(double v, int idx) {
    return __hiloint2double(__double2hiint(v), (__double2loint(v) & ~((1 << KEYBITS) - 1)) | idx);
}
__device__ __forceinline__ int key_index(double k) { return __double2loint(k) & ((1 << KEYBITS) - 1); }
// acc += (lane C of my group of G lanes)'s src * w without a trip through LDS: the 64-bit FMA takes its first operand through
// the DPP row broadcast (a "row" is 16 lanes; gfx90a+).  A group of 8 lanes is half a row: two instructions, each writing
// the half of the row (bank_mask) whose lane it broadcasts.  Every lane of the wave must be active.  The compiler does not
// see inside the statement: dpp_guard() keeps the DPP hazard (a VALU write of the broadcast register within the two
// instructions before) off, and tools/check_dpp_hazard.py checks the listing.
template <int G, int C>
__device__ __forceinline__ void fmac_grp(double& acc, double src, double w) {
    static_assert(G == 8 || G == 16, "groups of half a DPP row or a whole one");
    if constexpr (G == 16) {
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(w), "n"(C));
    } else {
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0x3" : "+v"(acc) : "v"(src), "v"(w), "n"(C));
        asm("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xc" : "+v"(acc) : "v"(src), "v"(w), "n"(C + 8));
    }
}
__device__ __forceinline__ void dpp_guard(double& v) { asm volatile("s_nop 1" : "+v"(v)); }
template <int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (N > 0) { static_for<N - 1>(f); f(std::integral_constant<int, N - 1>{}); }
}

template <int G>
__device__ __forceinline__ double grp_fmax(double v) {
    v = fmax_raw(v, dpp_d<DPP_XOR1>(v));
    v = fmax_raw(v, dpp_d<DPP_XOR2>(v));
    if (G >= 8) v = fmax_raw(v, dpp_d<DPP_HMIRROR>(v));
    if (G >= 16) v = fmax_raw(v, dpp_d<DPP_ROR8>(v));
    if (G >= 32) v = fmax_raw(v, __shfl_xor(v, 16));
    if (G == 64) v = fmax_raw(v, __shfl_xor(v, 32));
    return v;
}
template <int G>
__device__ __forceinline__ double grp_fmin(double v) {
    v = fmin_raw(v, dpp_d<DPP_XOR1>(v));
    v = fmin_raw(v, dpp_d<DPP_XOR2>(v));
    if (G >= 8) v = fmin_raw(v, dpp_d<DPP_HMIRROR>(v));
    if (G >= 16) v = fmin_raw(v, dpp_d<DPP_ROR8>(v));
    if (G >= 32) v = fmin_raw(v, __shfl_xor(v, 16));
    if (G == 64) v = fmin_raw(v, __shfl_xor(v, 32));
    return v;
}

// 1/x for finite, normal x: v_rcp_f64 plus two Newton steps (full fp64 accuracy, a fifth of the instructions and of the
// dependent latency of the IEEE division sequence; the operands here are pivots checked to be > 0, multipliers, rho, d)
__device__ __forceinline__ double frcp(double x) {
    double r = __builtin_amdgcn_rcp(x);
    r = fma(r, fma(-x, r, 1.0), r);
    r = fma(r, fma(-x, r, 1.0), r);
    return r;
}

#ifdef POLISH_PROF
// wave-uniform accumulators (scalar registers), one atomic per counter at the end of the kernel
#define PROF_MARK(i) do { const long long t_ = clock64(); pacc[i] += (unsigned long long)(t_ - tprof); tprof = t_; } while (0)
#define PROF_CNT(i) do { pacc[i] += 1ull; } while (0)
#else
#define PROF_MARK(i)
#define PROF_CNT(i)
#endif

// row of the lane's j-th register element: pairs (16 jj + 2 g, +1) -- two adjacent rows per lane so that LDS
// reads pair up (ds_read2_b64) and global loads/stores are 16 bytes per lane, 128 bytes per instance
// With 8 lanes per instance a 16-byte LDS read of one row pair per lane covers 128 bytes per instance, and the hardware
// serves such a read 16 lanes at a time: {0-3, 12-15, 20-27} = the same halves of instances 0/3 and of 1/2.  Every row of Hi
// starts on bank 0 (stride 1 KiB), so they would collide two ways; instances 2, 3 (6, 7) of a wave therefore keep their
// register pairs in the order jj ^ 1 (SW = 1): the same rows, the other 128-byte half of the 256-byte bank line.
#define ROWJ(j) (((((j) >> 1) & 1) ? rbO : rbE) + 2 * G * ((j) >> 1) + ((j) & 1))
#define ROW_OWNER(r) ((((r) >> 1) & (G - 1)) == g)
#define ROW_BIT(r) (((((r) / (2 * G)) ^ SW) << 1) | ((r) & 1))

constexpr int NBIN = 32;
// every bucket's count and cursor on a cache line of their own: the 1,024 workgroups of the key kernel add to them at the
// end, and ~20 k atomics on ONE line were what the launch waited for after its last workgroup had finished computing
constexpr int BIN_STRIDE = 32;

#ifndef OCT_DPP
#define OCT_DPP 1
#endif
#define OCT_LAYOUT \
    extern __shared__ __attribute__((aligned(16))) char smem[]; \
    constexpr int IPW = THREADS / G; \
    constexpr int NS = KMAX / G; \
    constexpr int RPL = 2 * RP; \
    constexpr int NQ = 2 * G * RP; \
    constexpr int LD = NQ; \
    constexpr int CH = (G >= 32) ? OCT_CH32 : 4, XCH = (G >= 32) ? OCT_XCH32 : 2; \
    [[maybe_unused]] constexpr bool DPPH = OCT_DPP && (G == 16 || (G == 8 && KMAX == 8)), DPPR = OCT_DPP && G == 16; \
    const int n = p.n, nx = p.nx; \
    [[maybe_unused]] const int CN = SPLIT ? p.sp_C : 1; \
    const int tid = threadIdx.x, g = tid & (G - 1), li = tid / G; \
    [[maybe_unused]] const int lane = tid & 63; \
    const int SW = (G == 8 && !(RP & 1)) ? ((li >> 1) & 1) : 0; \
    [[maybe_unused]] const int rbE = 2 * g + 2 * G * SW, rbO = 2 * g - 2 * G * SW; \
    double* sHi = reinterpret_cast<double*>(smem); \
    double* sLb = sHi + (size_t)CN * n * LD; \
    double* sUb = sLb + CN * NQ; \
    double* sD = sUb + CN * NQ; \
    double2* sBT = reinterpret_cast<double2*>(sD + CN * NQ); \
    [[maybe_unused]] int* sPerm = reinterpret_cast<int*>(sD + 3 * CN * NQ); \
    double* sKu = sD + 3 * CN * NQ + (SPLIT ? CN * NQ / 2 : 0); \
    double* sStage = sKu + (size_t)CN * (nx + 1) * NQ; \
    double* sX0 = sStage; \
    constexpr int KST = KMAX + 4; \
    double* myR = sKu + (size_t)li * KST; \
    double* myV = sKu + (size_t)(IPW + li) * KST; \
    int* myA = reinterpret_cast<int*>(sKu + (size_t)2 * IPW * KST) + li * KST; \


// one block of IPW instances: the whole active-set solve
// SPLIT: the work items are SUB-instances (instance `inst`, component `comp`, id `sub` = inst * components + comp) of a model
// whose Hessian is block diagonal under a permutation (split.hip): every component is a small box QP of its own, with its
// block of Hi, its rows of the map and its bounds at an offset in LDS; outputs go to the rows of the whole problem through
// the component's row map, and the last component of an instance to finish adds up the instance's objective and step count.
template <int G, int RP, int KMAX, int THREADS, bool COLD, bool SPLIT>
__device__ __forceinline__ void oct_block(const PolParams& p, const int slot, const int total, unsigned long long* pacc_out,
                                          const int inst, const double (&xpre)[64 / G], const int sub, const int comp) {
    static_assert(!SPLIT || COLD, "the component split is a cold-start path");
    OCT_LAYOUT
    const int hiB = SPLIT ? comp * n * LD : 0;              // my component's block of Hi, its bounds, its rows of the map
    const int bB = SPLIT ? comp * NQ : 0;
    const int kuB = SPLIT ? comp * (nx + 1) * NQ : 0;
#ifdef POLISH_PROF
    unsigned long long pacc[16] = {};
    long long tprof = clock64();
#endif
    const bool valid = slot < total;       // slot: my place in the batch / the queue / the cold-start order
    // the ADMM iterate (z, w) of my rows: issued first so that the HBM latency hides behind the staging below
    double zz[RPL], ww[RPL];
#pragma unroll
    for (int j = 0; j < RPL; ++j) { zz[j] = 0.0; ww[j] = 0.0; }
    double* zb = p.z + (size_t)inst * p.NP;
    double* wb = p.w + (size_t)inst * p.NP;
    if (!COLD && valid) {
        if (!(n & 1)) {     // pairs never straddle n: 16-byte loads
#pragma unroll
            for (int jj = 0; jj < RP; ++jj) {
                const int row = ROWJ(2 * jj);
                double2 a = make_double2(0.0, 0.0), b = a;
                if (row < n) { a = *reinterpret_cast<const double2*>(zb + row); b = *reinterpret_cast<const double2*>(wb + row); }
                zz[2 * jj] = a.x; zz[2 * jj + 1] = a.y; ww[2 * jj] = b.x; ww[2 * jj + 1] = b.y;
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int row = ROWJ(j);
                zz[j] = (row < n) ? zb[row] : 0.0;
                ww[j] = (row < n) ? wb[row] : 0.0;
            }
        }
    }

    // x0 of my instance: fetched by the group's own lanes before the staging (oct_body)
#pragma unroll
    for (int u = 0; u < 64 / G; ++u) { const int c = g + u * G; if (c < nx) sX0[li * nx + c] = xpre[u]; }
    __syncthreads();
    PROF_MARK(0);

    // ---- unconstrained minimiser xu = Ku1 [x0;1] on my rows, and c0 = qs'xu ----------------------
    double x[RPL];
    double c0 = 0.0;
    {
#pragma unroll
        for (int j = 0; j < RPL; ++j) x[j] = sKu[kuB + nx * NQ + ROWJ(j)];
        // four columns of the map at a time, their LDS reads (16 bytes: a lane's rows come in pairs) issued before the first
        // product: as a plain loop over a run-time nx every column waited for its own reads (11 % of the first pass of a
        // component-split solve, in-kernel counters).  Same sums in the same order.
        for (int c0 = 0; c0 < nx; c0 += 4) {
            double xc[4]; double2 kv[4][RP];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const bool in = c0 + u < nx;
                const int c = in ? c0 + u : nx;             // row nx of the map (the constant term) is always there
                xc[u] = in ? sX0[li * nx + c] : 0.0;
#pragma unroll
                for (int jj = 0; jj < RP; ++jj) kv[u][jj] = *reinterpret_cast<const double2*>(sKu + kuB + c * NQ + ROWJ(2 * jj));
            }
#pragma unroll
            for (int u = 0; u < 4; ++u)
                if (c0 + u < nx) {
#pragma unroll
                    for (int jj = 0; jj < RP; ++jj) { x[2 * jj] += kv[u][jj].x * xc[u]; x[2 * jj + 1] += kv[u][jj].y * xc[u]; }
                }
        }
        if constexpr (!SPLIT) {     // c0 = [x0;1]' Cq [x0;1], rows a = g, g+G, .. of the form in lane g (split: split_finish_kernel adds it)
            const int nx1 = nx + 1;
            for (int a = g; a < nx1; a += G) {
                const double* cr = p.Cq + (size_t)a * nx1;
                double sa = cr[nx];
                for (int b = 0; b < nx; ++b) sa += cr[b] * sX0[li * nx + b];
                c0 += sa * (a < nx ? sX0[li * nx + a] : 1.0);
            }
        }
        if constexpr (!SPLIT) c0 = grp_sum<G>(c0);
    }

    // ---- active-set guess from the ADMM iterate (z at a bound and the scaled dual pushing outwards) ---
    unsigned mU = 0u, mL = 0u;
    if (!COLD && valid) {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int row = ROWJ(j);
            const double hi = sUb[row], lo = sLb[row];
            const double tz = 1e-9 * bound_scale(lo, hi);
            const double tw = 1e-12 * fmax(1.0, fabs(zz[j]));
            if (row < n) {
                // cold (no ADMM iterate): start from the empty set.  Guessing the bounds the unconstrained minimiser
                // violates was measured and is slower (0.67 vs 0.58 ms per batch): the guess overshoots, the warm phase
                // has to factor and prune it and more instances outgrow the first pass
                // (the same guess for the heavy instances of the second pass only: also slower, 138 vs 146 M solves/s)
                if (zz[j] >= hi - tz && ww[j] > tw) mU |= 1u << j;
                else if (zz[j] <= lo + tz && ww[j] < -tw) mL |= 1u << j;
            }
        }
    }
    __syncthreads();       // the map and x0 are dead: their space becomes the exchange scratch
#pragma unroll
    for (int i = 0; i < NS; ++i) { myR[G * i + g] = 0.0; myV[G * i + g] = 0.0; myA[G * i + g] = 0; }
    wsync();

    // ---- per-instance state -------------------------------------------------------------------------
    double W[NS][KMAX];                 // rows 8i+g of S^-1 (zero where a slot is free)
    double lam[NS], rhs[NS], sgO[NS];   // my slots: multiplier (signed), xu[row]-bound, side (+1 upper, -1 lower, 0 free)
    int raO[NS];                        // my slots: row index (0 if free)
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        lam[i] = rhs[i] = sgO[i] = 0.0; raO[i] = 0;
#pragma unroll
        for (int c = 0; c < KMAX; ++c) W[i][c] = 0.0;
    }
    unsigned long long live = 0ull;     // occupied slots (same in the G lanes)
    int khi = 0;                        // 1 + highest slot ever occupied
    bool over = false, fail = false;
    int steps = 0, why = 0;
    if (p.mflag && p.mflag[0]) { fail = true; why = 8; }      // the shared factorisation failed: the wave kernel reports it

    // r = W h with h_c = Hi[p][row_c]; also h at my own slots
    auto sinv_h = [&](int po, int lim, double (&r)[NS], double (&hown)[NS]) {
#pragma unroll
        for (int i = 0; i < NS; ++i) r[i] = 0.0;
        if constexpr (DPPH) {
            // h at a slot is what the slot's owner reads for itself: broadcast from there (no list look-up, no gather)
#pragma unroll
            for (int i = 0; i < NS; ++i) { hown[i] = sHi[po + raO[i]]; dpp_guard(hown[i]); }
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if (s == 0 || __any(G * s < lim)) {
                    static_for<G>([&](auto gc) {
                        constexpr int c = decltype(gc)::value;
#pragma unroll
                        for (int i = 0; i < NS; ++i) fmac_grp<G, c>(r[i], hown[s], W[i][G * s + c]);
                    });
                }
            });
            return;
        }
#pragma unroll
        for (int cb = 0; cb < KMAX; cb += CH) {
            if (!__any(cb < lim)) break;
            int rw[CH]; double h[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) rw[u] = myA[cb + u];
#pragma unroll
            for (int u = 0; u < CH; ++u) h[u] = sHi[po + rw[u]];
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) r[i] += W[i][cb + u] * h[u];
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) hown[i] = sHi[po + raO[i]];
    };
    // W += coef_a * myR[c]
    auto rank1 = [&](const double (&coef)[NS], int lim, double (&uo)[NS]) {
        if constexpr (DPPR) {           // uo: the vector myR holds, at its owners
#pragma unroll
            for (int i = 0; i < NS; ++i) dpp_guard(uo[i]);
            static_for<NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                if (s == 0 || __any(G * s < lim)) {
                    static_for<G>([&](auto gc) {
                        constexpr int c = decltype(gc)::value;
#pragma unroll
                        for (int i = 0; i < NS; ++i) fmac_grp<G, c>(W[i][G * s + c], uo[s], coef[i]);
                    });
                }
            });
            return;
        }
#pragma unroll
        for (int cb = 0; cb < KMAX; cb += CH) {
            if (!__any(cb < lim)) break;
            double uu[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) uu[u] = myR[cb + u];
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) W[i][cb + u] += coef[i] * uu[u];
        }
    };
    // remove slot jb (predicate d): W -= W[:,jb] W[jb,:] / W[jb,jb]
    auto drop = [&](int jb, bool d) {
        int row = 0;
        if (d) row = myA[jb];
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (d && G * i + g == jb) {
#pragma unroll
                for (int c = 0; c < KMAX; ++c) myV[c] = W[i][c];
            }
        wsync();
        const double vjj = d ? myV[jb] : 1.0;
        const double al = d ? -frcp(vjj) : 0.0;
        double ai[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) ai[i] = myV[G * i + g] * al;
#pragma unroll
        for (int cb = 0; cb < KMAX; cb += CH) {
            if (!__any(cb < khi)) break;
            double vc[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) vc[u] = myV[cb + u];
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    W[i][cb + u] += ai[i] * vc[u];
                    if (d && (cb + u == jb || G * i + g == jb)) W[i][cb + u] = 0.0;
                }
        }
        if (d) {
            live &= ~(1ull << jb);
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (G * i + g == jb) { lam[i] = 0.0; rhs[i] = 0.0; sgO[i] = 0.0; raO[i] = 0; }
            if (ROW_OWNER(row)) { mU &= ~(1u << ROW_BIT(row)); mL &= ~(1u << ROW_BIT(row)); }
        }
        wsync();
        if (d && (jb & (G - 1)) == g) myA[jb] = 0;
        wsync();
    };

    PROF_MARK(1);
    // ---- warm phase (skipped by a wave none of whose instances has a guess: every cold start) -----------
    if (!COLD && __any(valid && (mU | mL) != 0u)) {
        // ordered list of the guessed set: lane g's rows after those of lanes < g
        const unsigned m = mU | mL;
        int* cntp = reinterpret_cast<int*>(myR);
        cntp[g] = __popc(m);
        wsync();
        int off = 0, k0 = 0;
#pragma unroll
        for (int gg = 0; gg < G; ++gg) { const int cg = cntp[gg]; if (gg < g) off += cg; k0 += cg; }
        wsync();
        if (g < G / 2) myR[g] = 0.0;     // the G counters overlaid myR[0..G/2)
        wsync();
        bool wact = valid;
        if (wact && k0 > KMAX) { over = true; why = 1; wact = false; }
        if (wact) {
            // list positions in ROW order, whatever the order of the lane's registers (SW swaps its row pairs two by two):
            // the order of the list is the order of the bordering, and the result must not depend on the instance's place
            // in the wave
            const unsigned mo = SW ? (((m & 0x33333333u) << 2) | ((m >> 2) & 0x33333333u)) : m;
#pragma unroll
            for (int j = 0; j < RPL; ++j)
                if ((m >> j) & 1u) {
                    const int row = ROWJ(j);
                    const bool up = (mU >> j) & 1u;
                    const int bj = SW ? (j ^ 2) : j;
                    const int pos = off + __popc(mo & ((1u << bj) - 1u));
                    myA[pos] = row;
                    myR[pos] = up ? 1.0 : -1.0;
                    myV[pos] = x[j] - (up ? sUb[row] : sLb[row]);
                }
        }
        wsync();
        if (!wact) k0 = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            const int a = G * i + g;
            const bool in = a < k0;
            raO[i] = in ? myA[a] : 0;
            sgO[i] = in ? myR[a] : 0.0;
            rhs[i] = in ? myV[a] : 0.0;
        }
        live = k0 >= 64 ? ~0ull : ((1ull << k0) - 1ull);
        khi = k0;
        wsync();
        // W = S^-1 by successive bordering in list order
        for (int t = 0; t < KMAX; ++t) {
            PROF_CNT(7);
            const bool actt = wact && t < k0;
            if (!__any(actt)) break;
            const int pr = actt ? myA[t] : 0;
            const int po = hiB + pr * LD;
            double r[NS], hown[NS];
            sinv_h(po, min(t, k0), r, hown);
            double part = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) part += hown[i] * r[i];
            const double d2 = sHi[po + pr] - grp_sum<G>(part);
            bool go = actt;
            if (go && !(d2 > 0.0)) { fail = true; why = 2; wact = false; go = false; }
            const double rd2 = go ? frcp(d2) : 0.0;
            double coef[NS], uo[NS];
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double u = (G * i + g == t) ? -1.0 : r[i];
                if constexpr (!DPPR) myR[G * i + g] = u;
                uo[i] = u;
                coef[i] = u * rd2;
            }
            wsync();
            rank1(coef, min(t + 1, k0), uo);
            wsync();
        }
        if (fail) { live = 0ull; }
        PROF_MARK(2);
        // multipliers on the guessed set; release the wrong-signed ones and solve again
        for (;;) {
#pragma unroll
            for (int i = 0; i < NS; ++i) myV[G * i + g] = rhs[i];
            wsync();
#pragma unroll
            for (int i = 0; i < NS; ++i) lam[i] = 0.0;
#pragma unroll
            for (int cb = 0; cb < KMAX; cb += CH) {
                if (!__any(cb < khi)) break;
                double v[CH];
#pragma unroll
                for (int u = 0; u < CH; ++u) v[u] = myV[cb + u];
#pragma unroll
                for (int u = 0; u < CH; ++u)
#pragma unroll
                    for (int i = 0; i < NS; ++i) lam[i] += W[i][cb + u] * v[u];
            }
            wsync();
            bool bad[NS]; bool anyb = false;
#pragma unroll
            for (int i = 0; i < NS; ++i) { bad[i] = wact && sgO[i] * lam[i] < 0.0; anyb = anyb || bad[i]; }
            if (!__any(anyb)) break;
            for (;;) {
                int cand = BIGI;
#pragma unroll
                for (int i = NS - 1; i >= 0; --i) if (bad[i]) cand = G * i + g;
                const int jb = grp_imin<G>(cand);
                const bool d = jb < BIGI;
                if (!__any(d)) break;
                drop(d ? jb : 0, d);
                if (d) ++steps;
#pragma unroll
                for (int i = 0; i < NS; ++i) if (G * i + g == jb) bad[i] = false;
            }
        }
    }

    PROF_MARK(3);
    // x -= Hi[:,A] c for the list-ordered coefficients c (my slots in cown)
    auto x_minus = [&](const double (&cown)[NS]) {
#pragma unroll
        for (int i = 0; i < NS; ++i) myR[G * i + g] = cown[i];
        wsync();
#pragma unroll
        for (int cb = 0; cb < KMAX; cb += XCH) {
            if (!__any(cb < khi)) break;
            double la[XCH]; int oa[XCH];
#pragma unroll
            for (int u = 0; u < XCH; ++u) { la[u] = myR[cb + u]; oa[u] = hiB + myA[cb + u] * LD; }
#pragma unroll
            for (int u = 0; u < XCH; ++u)
#pragma unroll
                for (int jj = 0; jj < RP; ++jj) {
                    const double2 hv = *reinterpret_cast<const double2*>(sHi + oa[u] + ROWJ(2 * jj));
                    x[2 * jj] -= hv.x * la[u]; x[2 * jj + 1] -= hv.y * la[u];
                }
        }
        wsync();
    };
    // e = rhs - S lam on my slots, with S read from Hi itself (not from the updated inverse W)
    auto resid = [&](double (&e)[NS]) {
#pragma unroll
        for (int i = 0; i < NS; ++i) { myR[G * i + g] = lam[i]; e[i] = rhs[i]; }
        wsync();
        int ro[NS];
#pragma unroll
        for (int i = 0; i < NS; ++i) ro[i] = raO[i];
#pragma unroll
        for (int cb = 0; cb < KMAX; cb += CH) {
            if (!__any(cb < khi)) break;
            int rw[CH]; double lv[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) { rw[u] = myA[cb + u]; lv[u] = myR[cb + u]; }
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) e[i] -= sHi[hiB + rw[u] * LD + ro[i]] * lv[u];      // Hi is symmetric
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) if (sgO[i] == 0.0) e[i] = 0.0;
        wsync();
    };
    // dl = W e
    auto w_times = [&](const double (&e)[NS], double (&dl)[NS]) {
#pragma unroll
        for (int i = 0; i < NS; ++i) { myV[G * i + g] = e[i]; dl[i] = 0.0; }
        wsync();
#pragma unroll
        for (int cb = 0; cb < KMAX; cb += CH) {
            if (!__any(cb < khi)) break;
            double v[CH];
#pragma unroll
            for (int u = 0; u < CH; ++u) v[u] = myV[cb + u];
#pragma unroll
            for (int u = 0; u < CH; ++u)
#pragma unroll
                for (int i = 0; i < NS; ++i) dl[i] += W[i][cb + u] * v[u];
        }
        wsync();
    };

    // one refinement of the multipliers against S itself (W is a product of rank-one updates), then
    // x = xu - Hi[:,A] lam on every row (rows in A land on their bound up to rounding; they are
    // reported as the bound itself and never tested for violation)
    if (!COLD) {
        double e[NS], dl[NS];
        resid(e);
        w_times(e, dl);
#pragma unroll
        for (int i = 0; i < NS; ++i) lam[i] += dl[i];
        x_minus(lam);
    }

    // ---- hand-over: continue from the state the previous pass left when its slot list was full ----------------------
    if (p.hand_in) {
        constexpr int HK = KMAX / 2;                    // the previous pass has half the slots
        constexpr int HENT = HK * HK + 4 * HK + 2;
        const bool handed = valid && !fail && p.need_big[sub] == 3;
        if (__any(handed)) {
            if (handed) {
                const double* hb = p.hand_in + (size_t)slot * HENT;
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    const int a = G * i + g;
                    if (a < HK) {
#pragma unroll
                        for (int c = 0; c < HK; c += 2) {
                            const double2 v = *reinterpret_cast<const double2*>(hb + a * HK + c);
                            W[i][c] = v.x; W[i][c + 1] = v.y;
                        }
                        lam[i] = hb[HK * HK + a]; rhs[i] = hb[HK * HK + HK + a]; sgO[i] = hb[HK * HK + 2 * HK + a];
                        raO[i] = (int)hb[HK * HK + 3 * HK + a];
                        myA[a] = raO[i]; myV[a] = sgO[i];
                    }
                }
                steps = (int)hb[HK * HK + 4 * HK];
                live = (1ull << HK) - 1ull; khi = HK;
            }
            wsync();
            if (handed) {
                for (int a = 0; a < HK; ++a) {
                    const int row = myA[a];
                    if (ROW_OWNER(row)) { if (myV[a] > 0.0) mU |= 1u << ROW_BIT(row); else mL |= 1u << ROW_BIT(row); }
                }
            }
            wsync();
            x_minus(lam);           // x = xu - Hi[:,A] lam (lam = 0 and no rows where nothing was handed over)
        }
    }

    PROF_MARK(4);
    // ---- main phase: add the most violated bound until none is violated ----------------------------
    bool run = valid && !over && !fail;
    int guard = 0;
    for (;;) {
        static_assert(2 * NQ <= (1 << KEYBITS), "2 row + side must fit the key's index bits");
        double bkey = 0.0;
        const unsigned m = mU | mL;
#pragma unroll
        for (int j = 0; j < RPL; ++j) {     // violation beyond the tolerance: x - (hi + tol) or (lo - tol) - x, as a key with (2 row + side)
            const int row = ROWJ(j);
            const double2 bt = sBT[bB + row];
            const double kv = fmax_raw(key_pack(x[j] - bt.y, 2 * row + 1), key_pack(bt.x - x[j], 2 * row));
            // a row of the working set gets the sign bit: never the maximum unless nothing is violated (and then it is not positive)
            const int sgn = (int)((m >> j) << 31);
            const double km = __hiloint2double(__double2hiint(kv) | sgn, __double2loint(kv));
            bkey = j == 0 ? km : fmax_raw(bkey, km);
        }
        bkey = grp_fmax<G>(bkey);
        const int code = bkey > 0.0 ? key_index(bkey) : -1;
        bool has = run && code >= 0;
        PROF_MARK(9);
        // the violation itself: add the row's tolerance back (the violated side's, which is finite): |staged bound - bound|,
        // read without a branch on the side (no violation: row 0's, unused)
        const int vrow = max(code, 0) >> 1, vside = code & 1;
        const double best = bkey + fabs(reinterpret_cast<const double*>(sBT)[2 * (bB + vrow) + vside] - sLb[vside * (CN * NQ) + bB + vrow]);
        if (!__any(has)) break;
        if (has && ++guard > p.max_steps) { fail = true; why = 5; run = false; has = false; }
        const int pr = has ? (code >> 1) : 0;
        const int po = hiB + pr * LD;
        const bool upP = code & 1;
        const double sp = upP ? 1.0 : -1.0;
        const double hpp = sHi[po + pr];
        double viol = best, rp = 0.0, accl = 0.0;
        bool first = true;
        bool inner = has;
        while (__any(inner)) {
            PROF_CNT(8);
            double r[NS], hown[NS];
            sinv_h(po, khi, r, hown);
            double part = 0.0, part2 = 0.0;
#pragma unroll
            for (int i = 0; i < NS; ++i) { part += hown[i] * r[i]; part2 += hown[i] * lam[i]; }
            const double d2 = hpp - grp_sum<G>(part);
            if (first) { rp = sp * viol + grp_sum<G>(part2); first = false; }   // xu_p - bnd_p = (x_p - bnd_p) + (Hi lam)_p
            if (inner && !(d2 > 0.0)) { fail = true; why = 4; run = false; inner = false; }
            const double rd2 = frcp(d2);
            const double t2 = viol * rd2;
            static_assert(KMAX <= (1 << KEYBITS), "a slot index must fit the key's index bits");
            double t1 = 1e300;              // the first blocking multiplier, as a key with its slot
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const double rr = sp * r[i];
                const double t = key_pack(fmin(lam[i] * frcp(rr), 1e300), G * i + g);     // finite: an infinity with index bits would be a NaN
                t1 = (sgO[i] * rr > 0.0) ? fmin_raw(t1, t) : t1;
            }
            t1 = grp_fmin<G>(t1);
            const int jb = t1 < 1e299 ? key_index(t1) : BIGI;
            const bool full = (t2 <= t1);
            // a full step needs a free slot for p: without one stop BEFORE the step, so that (x, lam, W) stay the consistent
            // state of the current working set -- that state is what the next pass is handed
            if (inner && full && live == (KMAX == 64 ? ~0ull : ((1ull << (KMAX & 63)) - 1ull))) { over = true; why = 3; run = false; inner = false; }
            const double t = inner ? (full ? t2 : t1) : 0.0;
            PROF_MARK(10);
            // x -= t (sp Hi[:,p] - Hi[:,A] (sp r)),  lam_A -= t sp r
#pragma unroll
            for (int i = 0; i < NS; ++i) myR[G * i + g] = r[i];
            wsync();
            const double cp = t * sp;
            {
#pragma unroll
                for (int jj = 0; jj < RP; ++jj) {
                    const double2 hv = *reinterpret_cast<const double2*>(sHi + po + ROWJ(2 * jj));
                    x[2 * jj] -= cp * hv.x; x[2 * jj + 1] -= cp * hv.y;
                }
            }
#pragma unroll
            for (int cb = 0; cb < KMAX; cb += XCH) {
                if (!__any(cb < khi)) break;
                double ca[XCH]; int oa[XCH];
#pragma unroll
                for (int u = 0; u < XCH; ++u) { ca[u] = cp * myR[cb + u]; oa[u] = hiB + myA[cb + u] * LD; }
#pragma unroll
                for (int u = 0; u < XCH; ++u)
#pragma unroll
                    for (int jj = 0; jj < RP; ++jj) {
                        const double2 hv = *reinterpret_cast<const double2*>(sHi + oa[u] + ROWJ(2 * jj));
                        x[2 * jj] += ca[u] * hv.x; x[2 * jj + 1] += ca[u] * hv.y;
                    }
            }
#pragma unroll
            for (int i = 0; i < NS; ++i) lam[i] -= cp * r[i];
            accl += cp;
            PROF_MARK(11);
            if (inner) ++steps;
            bool app = inner && full, drp = inner && !full;
            int pos = 0;
            if (app) {
                pos = __ffsll((long long)~live) - 1;
                if (pos < 0 || pos >= KMAX) { over = true; why = 3; run = false; app = false; pos = 0; }
                inner = false;
            }
            if (__any(app)) {
                wsync();
                if (app && (pos & (G - 1)) == g) { myR[pos] = -1.0; myA[pos] = pr; }
                if (app) khi = max(khi, pos + 1);
                wsync();
                double coef[NS], uo[NS];
#pragma unroll
                for (int i = 0; i < NS; ++i) {
                    uo[i] = (G * i + g == pos) ? -1.0 : r[i];
                    coef[i] = app ? ((G * i + g == pos) ? -rd2 : r[i] * rd2) : 0.0;
                }
                rank1(coef, khi, uo);
                if (app) {
#pragma unroll
                    for (int i = 0; i < NS; ++i)
                        if (G * i + g == pos) { lam[i] = accl; rhs[i] = rp; raO[i] = pr; sgO[i] = sp; }
                    live |= 1ull << pos;
                    if (ROW_OWNER(pr)) { if (upP) mU |= 1u << ROW_BIT(pr); else mL |= 1u << ROW_BIT(pr); }
                }
            }
            PROF_MARK(12);
            if (__any(drp)) {
                PROF_CNT(14);
                if (drp) viol -= t * d2;
                drop(drp ? jb : 0, drp);
                if (drp && ++guard > p.max_steps) { fail = true; why = 6; run = false; inner = false; }
            }
            wsync();
            PROF_MARK(13);
        }
    }

    // ---- refinement: the active rows of x must sit ON their bounds, i.e. S lam = rhs with the true S -----
    for (int itr = 0; itr < 3; ++itr) {
        double e[NS], dl[NS];
        resid(e);
        bool need = false;
#pragma unroll
        for (int i = 0; i < NS; ++i) need = need || fabs(e[i]) > 1e-13 * fmax(1.0, fabs(rhs[i]));
        if (!__any(need && run)) break;
        // the correction is the instance's own business: one that does not need it must not receive a rounding-level one because
        // a neighbour in the wave does (which instances share a wave changes from solve to solve with the cold-start order,
        // and the results are meant to be reproducible bit for bit)
        const unsigned long long nb = __ballot(need && run);
        const bool mine = G == 64 ? nb != 0ull : ((nb >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1ull)) != 0ull;
        if (!mine) {
#pragma unroll
            for (int i = 0; i < NS; ++i) e[i] = 0.0;
        }
        w_times(e, dl);
#pragma unroll
        for (int i = 0; i < NS; ++i) lam[i] += dl[i];
        x_minus(dl);
    }

    PROF_MARK(5);
    // ---- outputs --------------------------------------------------------------------------------------
    // The keyed scan trusts its operands to be finite: a NaN key is dropped by v_max_f64 and would read as "nothing violated".
    // The staged bounds are finite by construction; an x that is not (overflow, NaN data) fails the instance here, once per
    // solve, and the next pass / the any-size kernel (which has no keys) takes it.
    {
        bool xbad = false;
#pragma unroll
        for (int j = 0; j < RPL; ++j) xbad = xbad || !(fabs(x[j]) < 1e300);
        const unsigned long long nb = __ballot(xbad);
        if (G == 64 ? nb != 0ull : ((nb >> (lane & ~(G - 1))) & ((1ull << (G & 63)) - 1ull)) != 0ull) { fail = true; if (!why) why = 10; }
    }
    const bool okk = valid && !over && !fail;
    int qslot = -1;
    if (valid && g == 0) {
        if (!okk && p.qout) { qslot = atomicAdd(p.qout_count, 1); p.qout[qslot] = sub; }      // queue for the next, larger pass
        else if (SPLIT && !okk && p.redo_q) p.redo_q[atomicAdd(p.redo_count, 1)] = inst;     // no pass left: the instance as a whole
        const bool give = over && why == 3 && qslot >= 0 && ((p.hand_out && qslot < p.hand_out_cap) || p.hand_zw);
        p.need_big[sub] = okk ? 0 : (why == 3 && !give ? 9 : (why ? why : 7));       // 3 = handed over, 9 = list full, start again
    }
    double dual = 0.0;
#pragma unroll
    for (int i = 0; i < NS; ++i) dual += lam[i] * rhs[i];
    dual = grp_sum<G>(dual);
    const double obj = 0.5 * c0 + 0.5 * dual;       // f(xu) + 1/2 lam'S lam
    if constexpr (SPLIT) {
        if (okk) {          // my component's rows of the instance, through the row map (8-byte stores)
            double* Ub = p.U + (size_t)inst * p.sp_n;
            double* lamb = p.info.lam ? p.info.lam + (size_t)inst * p.sp_n : nullptr;
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int row = ROWJ(j);
                const int gr = sPerm[bB + row];
                if (gr >= 0) {
                    const bool act = ((mU | mL) >> j) & 1u;
                    const double xo = ((mU >> j) & 1u) ? sUb[bB + row] : (((mL >> j) & 1u) ? sLb[bB + row] : x[j]);
                    Ub[gr] = sD[bB + row] * xo;
                    if (p.keep_state) { zb[gr] = xo; if (!act) wb[gr] = 0.0; }
                    if (lamb && !act) lamb[gr] = 0.0;
                }
            }
#pragma unroll
            for (int i = 0; i < NS; ++i)
                if (sgO[i] != 0.0) {
                    const int row = raO[i];
                    const int gr = sPerm[bB + row];
                    if (p.keep_state) wb[gr] = lam[i] / p.rho;
                    if (lamb) lamb[gr] = lam[i] / sD[bB + row];
                }
        }
        // the instance's record is put together from its components' parts by split_finish_kernel (polish.hip), in component
        // order, after the last pass: no cross-workgroup hand-shake here (device-scope fences cost an L2 write-back each)
        if (valid && g == 0 && okk) { p.sp_part[sub] = obj; p.sp_steps[sub] = steps; }
    } else
#ifdef POLISH_NOOUT
    if (okk && p.B < 0) {
#else
    if (okk) {
#endif
        if (!(n & 1)) {     // 16-byte stores; a pair holding an active row writes its multipliers one by one
#pragma unroll
            for (int jj = 0; jj < RP; ++jj) {
                const int row = ROWJ(2 * jj);
                if (row < n) {
                    const unsigned a2 = ((mU | mL) >> (2 * jj)) & 3u;
                    double2 xo;
                    xo.x = ((mU >> (2 * jj)) & 1u) ? sUb[row] : (((mL >> (2 * jj)) & 1u) ? sLb[row] : x[2 * jj]);
                    xo.y = ((mU >> (2 * jj + 1)) & 1u) ? sUb[row + 1] : (((mL >> (2 * jj + 1)) & 1u) ? sLb[row + 1] : x[2 * jj + 1]);
                    *reinterpret_cast<double2*>(p.U + (size_t)inst * n + row) = make_double2(sD[row] * xo.x, sD[row + 1] * xo.y);
                    if (p.keep_state) *reinterpret_cast<double2*>(zb + row) = xo;          // the polished point is the warm-start state
                    if (a2 == 0u) {
                        if (p.keep_state) *reinterpret_cast<double2*>(wb + row) = make_double2(0.0, 0.0);
                        if (p.info.lam) *reinterpret_cast<double2*>(p.info.lam + (size_t)inst * n + row) = make_double2(0.0, 0.0);
                    } else {
                        if (!(a2 & 1u)) { if (p.keep_state) wb[row] = 0.0; if (p.info.lam) p.info.lam[(size_t)inst * n + row] = 0.0; }
                        if (!(a2 & 2u)) { if (p.keep_state) wb[row + 1] = 0.0; if (p.info.lam) p.info.lam[(size_t)inst * n + row + 1] = 0.0; }
                    }
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < RPL; ++j) {
                const int row = ROWJ(j);
                if (row < n) {
                    const bool act = ((mU | mL) >> j) & 1u;
                    const double xo = ((mU >> j) & 1u) ? sUb[row] : (((mL >> j) & 1u) ? sLb[row] : x[j]);
                    p.U[(size_t)inst * n + row] = sD[row] * xo;
                    if (p.keep_state) zb[row] = xo;
                    if (!act) {                 // rows in the active set get their multiplier from the slot owner below
                        if (p.keep_state) wb[row] = 0.0;
                        if (p.info.lam) p.info.lam[(size_t)inst * n + row] = 0.0;
                    }
                }
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i)
            if (sgO[i] != 0.0) {
                const int row = raO[i];
                if (p.keep_state) wb[row] = lam[i] / p.rho;
                if (p.info.lam) p.info.lam[(size_t)inst * n + row] = lam[i] / sD[row];
            }
        if (g == 0) {
            if (p.info.status) p.info.status[inst] = MPCQP_SOLVED;
            if (p.info.obj) p.info.obj[inst] = obj;
            if (p.info.iters) p.info.iters[inst] = p.admm_ran ? p.it_ws[inst] : 0;
            if (p.info.prim_res) p.info.prim_res[inst] = p.admm_ran ? p.rp_ws[inst] : 0.0;
            if (p.info.dual_res) p.info.dual_res[inst] = p.admm_ran ? p.rd_ws[inst] : 0.0;
            if (p.info.polish_steps) p.info.polish_steps[inst] = steps;
        }
    }
    if (p.hand_out && p.qout && valid && over && why == 3) {
        // the slot list is full but consistent (a dual-feasible working set with its inverse): hand the state to the next pass
        constexpr int HENT = KMAX * KMAX + 4 * KMAX + 2;
        qslot = __shfl(qslot, lane & ~(G - 1));
        if (qslot < p.hand_out_cap) {
            double* hb = p.hand_out + (size_t)qslot * HENT;
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const int a = G * i + g;
#pragma unroll
                for (int c = 0; c < KMAX; c += 2) *reinterpret_cast<double2*>(hb + a * KMAX + c) = make_double2(W[i][c], W[i][c + 1]);
                hb[KMAX * KMAX + a] = lam[i]; hb[KMAX * KMAX + KMAX + a] = rhs[i]; hb[KMAX * KMAX + 2 * KMAX + a] = sgO[i];
                hb[KMAX * KMAX + 3 * KMAX + a] = (double)raO[i];
            }
            if (g == 0) hb[KMAX * KMAX + 4 * KMAX] = (double)steps;
        }
    }
    if (!SPLIT && p.hand_zw && p.qout && valid && over && why == 3) {
#pragma unroll
        for (int j = 0; j < RPL; ++j) {
            const int row = ROWJ(j);
            if (row < n) {
                const bool up = (mU >> j) & 1u, lo = (mL >> j) & 1u;
                zb[row] = up ? sUb[row] : (lo ? sLb[row] : 0.0);
                wb[row] = up ? 1.0 : (lo ? -1.0 : 0.0);
            }
        }
    }
    PROF_MARK(6);
#ifdef POLISH_PROF
    if (lane == 0) for (int i = 0; i < 16; ++i) pacc_out[i] += pacc[i];
#endif
}

// COLD: no (z, w) to start from -- the instantiation every default solve uses (it carries neither the iterate registers nor
// the warm-phase code).
template <int G, int RP, int KMAX, int THREADS, bool COLD, bool SPLIT>
__device__ __forceinline__ void oct_body(const PolParams& p) {
    OCT_LAYOUT
    (void)sLb; (void)sUb; (void)sD; (void)sBT; (void)sX0; (void)myR; (void)myV; (void)myA; (void)g; (void)li; (void)CH; (void)XCH; (void)NS; (void)RPL;
    unsigned long long pacc[16] = {};
    // one block of IPW instances per workgroup (a contiguous block of the batch, or entries of a device-side queue; launches
    // over a queue are sized for the whole batch, so most of their workgroups leave here).
    // Persistent workgroups that stage once and walk several blocks were measured twice (round 1 with prefetch, round 2 as
    // a plain loop): the loop around this much code costs 30 % in register allocation, more than the staging it saves.
    static_assert(!SPLIT, "split models: oct_body_split");
    const int total = p.qin ? *p.qin_count : p.B;
    if ((int)blockIdx.x * IPW >= total) return;
    if ((p.q_gate > 0 && total < p.q_gate) || (p.q_gate < 0 && total >= -p.q_gate)) return;     // not this instantiation's regime
    // my instance and its x0 BEFORE the staging: the chain place -> (bucket sizes ->) instance index -> x0 is two or three
    // round trips to memory, which now pass while the 115 KB of Hi arrive instead of after them
    const int blk = blockIdx.x;
    const int slot0 = blk * IPW;
    const bool valid = slot0 + li < total;
    int sub = valid ? (p.qin ? p.qin[slot0 + li] : slot0 + li) : 0;
    if (p.seg) {        // my place in the cold-start order (largest bucket first) -> bucket and offset, from the buckets' sizes
        const int pos = slot0 + li;
        int bk = 0, accb = 0, acc = 0;      // (as in oct_body_split: the last bucket, in this order, whose start the place has reached)
#pragma unroll
        for (int k = NBIN - 1; k >= 0; --k) {
            const int c = p.seg_counts[k * BIN_STRIDE];
            const bool ge = pos >= acc;
            bk = ge ? k : bk; accb = ge ? acc : accb;
            acc += c;
        }
        const int off = pos - accb;
        sub = valid ? p.seg[(size_t)bk * p.B + off] : 0;
    }
    const int inst = sub, comp = 0;
    double xpre[64 / G];
#pragma unroll
    for (int u = 0; u < 64 / G; ++u) { const int c = g + u * G; xpre[u] = (valid && c < nx) ? p.x0[(size_t)inst * nx + c] : 0.0; }
    // ---- stage the shared data --------------------------------------------------------------
    {   // Hi -> LDS.  The workgroup can do nothing else until this has landed, so the loads are as wide as they can be (16 bytes,
        // contiguous over the workgroup) and all in flight before the first LDS write waits for one
        const int nrH = n;
        if (!(n & 1)) {
            const int hn = n >> 1, tot2 = nrH * hn;         // double2 elements
            constexpr int U = 8;
            for (int base = tid; base < tot2; base += U * THREADS) {
                double2 v[U]; int off[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int idx = base + u * THREADS;
                    const bool in = idx < tot2;
                    const int r = in ? idx / hn : 0;
                    off[u] = in ? r * LD + 2 * (idx - r * hn) : -1;
                    v[u] = in ? *reinterpret_cast<const double2*>(p.Hi + (size_t)2 * idx) : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (off[u] >= 0) { sHi[off[u]] = v[u].x; sHi[off[u] + 1] = v[u].y; }
            }
        } else {
            constexpr int TR = THREADS / 128;
            const int tc = tid & 127, tr = tid >> 7;
            for (int r0 = tr; r0 < nrH; r0 += 6 * TR) {
                double v[6];
#pragma unroll
                for (int u = 0; u < 6; ++u) { const int r = r0 + TR * u; v[u] = (r < nrH && tc < n) ? p.Hi[(size_t)r * n + tc] : 0.0; }
#pragma unroll
                for (int u = 0; u < 6; ++u) { const int r = r0 + TR * u; if (r < nrH && tc < n) sHi[r * LD + tc] = v[u]; }
            }
        }
        const int npad = LD - n;            // zero columns n .. LD-1
        for (int idx = tid; idx < nrH * npad; idx += THREADS) { const int r = idx / npad; sHi[r * LD + n + (idx - r * npad)] = 0.0; }
    }
    for (int j = tid; j < NQ; j += THREADS) {
        const double lo = (j < n) ? p.lbs[j] : -1e300, hi = (j < n) ? p.ubs[j] : 1e300;
        sLb[j] = lo;
        sUb[j] = hi;
        sD[j] = (j < n) ? p.d[j] : 0.0;
        const double tol = 1e-12 * bound_scale(lo, hi);
        sBT[j] = make_double2(fmax(lo - tol, -1e300), fmin(hi + tol, 1e300));   // finite: the keyed scan packs bits into x - bound
    }
    {   // the map [Ku | ku]': (nx+1) rows of NQ, 16-byte loads
        // (NP and NQ are multiples of 16), four in flight per thread
        const int hq = NQ / 2, tot2 = (nx + 1) * hq;
        constexpr int U = 4;
        for (int base = tid; base < tot2; base += U * THREADS) {
            double2 vk[U]; int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * THREADS;
                const bool in = idx < tot2;
                const int c = in ? idx / hq : 0, j = 2 * (idx - c * hq);
                off[u] = in ? c * NQ + j : -1;
                const bool ld = in && j < p.NP;
                vk[u] = ld ? *reinterpret_cast<const double2*>(p.KuT + (size_t)c * p.NP + j) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (off[u] >= 0) *reinterpret_cast<double2*>(sKu + off[u]) = vk[u];
        }
    }
    oct_block<G, RP, KMAX, THREADS, COLD, SPLIT>(p, slot0 + li, total, pacc, inst, xpre, sub, comp);
#ifdef POLISH_PROF
    if (lane == 0 && p.prof) {
        unsigned long long tot = 0;
        for (int i = 0; i < 16; ++i) { atomicAdd(p.prof + (G >= 16 ? 16 : 0) + i, pacc[i]); if (i != 7 && i != 8 && i != 14) tot += pacc[i]; }
        atomicMax(p.prof + (G >= 16 ? 16 : 0) + 15, tot);       // longest wave: calibrates the tick against the kernel time
    }
#endif
}

// Component-split models (split.hip): the work items are sub-instances (instance, component), a few hundred flops each, so
// what a workgroup costs is its start: everything it stages is ONE image in the layout of this LDS (split_setup builds it
// for LD = NQ = sp_ld, which every 4- and 8-lane pass has), and the first pieces of it are in flight before the chain bucket
// sizes -> my sub-instance -> its x0 is walked.  A queue-fed pass (p.qin: what outgrew the pass before) looks at its queue first.
template <int G, int RP, int KMAX, int THREADS>
__device__ __forceinline__ void oct_body_split(const PolParams& p) {
    constexpr bool SPLIT = true;
    OCT_LAYOUT
    (void)sLb; (void)sUb; (void)sD; (void)sBT; (void)sX0; (void)myR; (void)myV; (void)myA; (void)g; (void)li; (void)CH; (void)XCH; (void)NS; (void)RPL;
    unsigned long long pacc[16] = {};
    int total = p.B;
    if (p.qin) { total = *p.qin_count; if ((int)blockIdx.x * IPW >= total) return; }
    const int sl = p.sp_ld;
    const int img2 = (CN * n * LD + 5 * CN * NQ + CN * NQ / 2 + CN * (nx + 1) * NQ) >> 1;      // double2 pieces of the image
    constexpr int IMG_U = 4;
    double2 img[IMG_U];
    if (sl == NQ) {
#pragma unroll
        for (int u = 0; u < IMG_U; ++u) {
            const int idx = tid + u * THREADS;
            img[u] = idx < img2 ? reinterpret_cast<const double2*>(p.sp_img)[idx] : make_double2(0.0, 0.0);
        }
    }
    const int slot = blockIdx.x * IPW + li;
    int sub = 0;
    if (!p.qin && p.seg) {      // my place in the cold-start order (largest bucket first) -> bucket and offset, from the buckets' sizes;
        // (the running sum `acc` is uniform -- scalar loads and adds; per lane one comparison and two selects per bucket: the
        // place belongs to the LAST bucket, in this order, whose start it has reached -- an empty bucket starts where the next
        // one does and loses to it)
        int bk = 0, accb = 0, acc = 0;      // split_key_kernel queued only the components with something to do
#pragma unroll
        for (int k = NBIN - 1; k >= 0; --k) {        // only the buckets this launch is given (the others run beside it)
            const int c = (k >= p.seg_lo && k <= p.seg_hi) ? p.seg_counts[k * BIN_STRIDE] : 0;
            const bool ge = slot >= acc;
            bk = ge ? k : bk; accb = ge ? acc : accb;
            acc += c;
        }
        const int off = slot - accb;
        total = acc;
        if ((int)blockIdx.x * IPW >= total) return;
        sub = slot < total ? p.seg[(size_t)bk * p.B + off] : 0;
    } else if (slot < total) sub = p.qin ? p.qin[slot] : slot;
    const int inst = sub / CN, comp = sub - inst * CN;
    double xpre[64 / G];
#pragma unroll
    for (int u = 0; u < 64 / G; ++u) { const int c = g + u * G; xpre[u] = (slot < total && c < nx) ? p.x0[(size_t)inst * nx + c] : 0.0; }
    if (sl == NQ) {
#pragma unroll
        for (int u = 0; u < IMG_U; ++u) { const int idx = tid + u * THREADS; if (idx < img2) reinterpret_cast<double2*>(sHi)[idx] = img[u]; }
        for (int base = tid + IMG_U * THREADS; base < img2; base += IMG_U * THREADS) {
            double2 v[IMG_U];
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) { const int idx = base + u * THREADS; v[u] = idx < img2 ? reinterpret_cast<const double2*>(p.sp_img)[idx] : make_double2(0.0, 0.0); }
#pragma unroll
            for (int u = 0; u < IMG_U; ++u) { const int idx = base + u * THREADS; if (idx < img2) reinterpret_cast<double2*>(sHi)[idx] = v[u]; }
        }
    } else {                // a pass whose LDS rows are wider than the packed ones (16 lanes, components of <= 16 rows)
        for (int idx = tid; idx < CN * n * LD; idx += THREADS) { const int r = idx / LD, j = idx - r * LD; sHi[idx] = j < sl ? p.Hi[(size_t)r * sl + j] : 0.0; }
        for (int idx = tid; idx < CN * (nx + 1) * NQ; idx += THREADS) { const int r = idx / NQ, j = idx - r * NQ; sKu[idx] = j < sl ? p.KuT[(size_t)r * sl + j] : 0.0; }
        for (int idx = tid; idx < CN * NQ; idx += THREADS) {
            const int c = idx / NQ, j = idx - c * NQ;
            const double lo = j < sl ? p.lbs[c * sl + j] : -1e300, hi = j < sl ? p.ubs[c * sl + j] : 1e300;
            sLb[idx] = lo; sUb[idx] = hi; sD[idx] = j < sl ? p.d[c * sl + j] : 0.0;
            const double tol = 1e-12 * bound_scale(lo, hi);
            sBT[idx] = make_double2(fmax(lo - tol, -1e300), fmin(hi + tol, 1e300));     // finite: the keyed scan packs bits into x - bound
            sPerm[idx] = j < sl ? p.sp_perm[c * sl + j] : -1;
        }
    }
    oct_block<G, RP, KMAX, THREADS, true, true>(p, slot, total, pacc, inst, xpre, sub, comp);
#ifdef POLISH_PROF
    // (every 16th workgroup reports: with all of them, 350 k atomics on 16 addresses back the memory pipeline up and the waves'
    // own stores -- phase "out" -- appear to take three quarters of the launch)
    if (lane == 0 && p.prof && (blockIdx.x & 15) == 0) {
        unsigned long long tot = 0;
        for (int i = 0; i < 16; ++i) { atomicAdd(p.prof + (G >= 16 ? 16 : 0) + i, pacc[i]); if (i != 7 && i != 8 && i != 14) tot += pacc[i]; }
        atomicMax(p.prof + (G >= 16 ? 16 : 0) + 15, tot);
    }
#endif
}

// WPS: waves per SIMD the register allocation has to leave room for (HIP's second launch bound); the split passes want two
// workgroups per CU so that one's start-up latency (bucket sizes -> sub-instance -> x0, the staging) hides behind the other's work
template <int G, int RP, int KMAX, int THREADS, bool COLD, bool SPLIT = false, int WPS = THREADS / 256>
__global__ __launch_bounds__(THREADS, WPS) void polish_oct_kernel(PolParams p) {
    if constexpr (SPLIT) oct_body_split<G, RP, KMAX, THREADS>(p);
    else oct_body<G, RP, KMAX, THREADS, COLD, false>(p);
}

template <int G, int RP, int KMAX, int THREADS, bool COLD, bool SPLIT = false, int WPS = THREADS / 256>
int launch_oct_c(const PolParams& p, int ninst, hipStream_t st) {
    constexpr int NQ = 2 * G * RP, LD = NQ, IPW = THREADS / G;
    if (!p.Cq) return -1;
    const size_t CN = SPLIT ? p.sp_C : 1;
    const size_t x0s = (size_t)IPW * p.nx * 8;
    // the exchange arrays of the instances (r, v: doubles, the row list: ints) take the place of the map Ku1 and of x0 once the
    // unconstrained minimiser is formed; KMAX + 4 entries apart, so that the instances of a wave read their (broadcast)
    // entries from different banks -- KMAX apart, every other instance of the wave sat on the same ones
    const size_t xch = (size_t)IPW * (KMAX + 4) * 20;
    const size_t kux = CN * (p.nx + 1) * NQ * 8 + x0s;
    const size_t lds = (CN * p.n * LD + 5 * CN * NQ + (SPLIT ? CN * NQ / 2 : 0)) * 8 + (kux > xch ? kux : xch);
    if (lds > 160 * 1024) return -1;
    hipError_t e = ensure_dyn_lds(p.dev, reinterpret_cast<const void*>(&polish_oct_kernel<G, RP, KMAX, THREADS, COLD, SPLIT, WPS>), lds);
    if (e != hipSuccess) return (int)e;
    hipLaunchKernelGGL((polish_oct_kernel<G, RP, KMAX, THREADS, COLD, SPLIT, WPS>), dim3(ceil_div(ninst, IPW)), dim3(THREADS), lds, st, p);
    return (int)hipGetLastError();
}

template <int G, int RP, int KMAX, int THREADS>
int launch_oct(const PolParams& p, int ninst, hipStream_t st) {
    return p.cold ? launch_oct_c<G, RP, KMAX, THREADS, true>(p, ninst, st) : launch_oct_c<G, RP, KMAX, THREADS, false>(p, ninst, st);
}

// row pairs per lane: n <= 2 G RP
template <int G, int KMAX, int THREADS>
int launch_oct_rp(const PolParams& p, int ninst, hipStream_t st) {
    const int rp = ceil_div(p.n, 2 * G);
    if (rp <= 1) return launch_oct<G, 1, KMAX, THREADS>(p, ninst, st);
    if (rp <= 2) return launch_oct<G, 2, KMAX, THREADS>(p, ninst, st);
    if (rp <= 4) return launch_oct<G, 4, KMAX, THREADS>(p, ninst, st);
    if constexpr (G == 8) {
        if (rp <= 6) return launch_oct<G, 6, KMAX, THREADS>(p, ninst, st);
        if (rp <= 8) return launch_oct<G, 8, KMAX, THREADS>(p, ninst, st);
    }
    return -1;
}

}  // namespace

// One pass of the split cascade (split.hip): `slots` = 8 (8 lanes per sub-instance), 16 (8 lanes) or 32 (16 lanes) working-set
// slots; rows per lane from the component size p.n.  -1 if the shape has no instantiation.
int launch_oct_split(const PolParams& p, int slots, hipStream_t st) {
    const int n = p.n;
    // first pass: 8 lanes per sub-instance and room for two workgroups per CU (one's start-up hides behind the other's work).
    // Measured and dropped (65,536 instances of configs[2], first pass 136 us): 4 lanes per sub-instance (16 per wave: 181 VGPRs,
    // or 120 spilled at the 128 this occupancy needs: 144 / 187 us); 256-thread workgroups (154 us: twice the staging);
    // wave-persistent workgroups with the next unit's data prefetched (146 - 184 us: the loop around the solve costs more
    // registers than its prologue saves); 640-thread workgroups at five waves per SIMD after the instruction-level pass had
    // brought the body to 103 VGPRs (96 + 72 bytes of scratch: 232 us per solve against 179)
    if (slots == 8) {
        if (n <= 16) return launch_oct_c<8, 1, 8, 512, true, true, 4>(p, p.B, st);
        if (n <= 32) return launch_oct_c<8, 2, 8, 512, true, true, 4>(p, p.B, st);
        if (n <= 64) return launch_oct_c<8, 4, 8, 512, true, true>(p, p.B, st);
    } else if (slots == 16) {
        if (n <= 16) return launch_oct_c<8, 1, 16, 512, true, true>(p, p.B, st);
        if (n <= 32) return launch_oct_c<8, 2, 16, 512, true, true>(p, p.B, st);
        if (n <= 64) return launch_oct_c<8, 4, 16, 512, true, true>(p, p.B, st);
    } else if (slots == 17) {           // 16 slots, 16 lanes per sub-instance: the first pass of the smallest batches (split.hip)
        if (n <= 32) return launch_oct_c<16, 1, 16, 512, true, true>(p, p.B, st);
    } else if (slots == 32) {
        if (n <= 32) return launch_oct_c<16, 1, 32, 512, true, true>(p, p.B, st);
        if (n <= 64) return launch_oct_c<16, 2, 32, 512, true, true>(p, p.B, st);
    }
    return -1;
}

// ---- cold start: order the instances by expected work ---------------------------------------------------------------------
// The lanes of a wave walk their 8 (16) instances in lock-step, so a wave takes as long as its slowest instance.  Without
// an ADMM iterate the number of bounds the unconstrained minimiser violates predicts the number of active-set steps:
// instances are bucketed by that count (32 buckets, largest first) and the first pass reads them through the queue.
constexpr int SMALL_BATCH = 8192;

// xu = Ku1 [x0; 1] for 16 instances per wave on fp64 MFMA tiles (A = 16 rows of Ku1, B = [x0; 1] of 16 instances), the
// violated rows counted straight from the D registers: lane l holds rows (l>>4)+4r of instance l&15
__global__ __launch_bounds__(256) void cold_key_kernel(PolParams p, int* keys, int* bins, int* seg) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int hist[NBIN];
    const int n = p.n, nx = p.nx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int KS = (nx + 1 + 3) / 4, KP = 4 * KS, NQ = 16 * ((n + 15) / 16);
    const int XS = KP + 1;      // odd stride of the [x0; 1] rows: the 16 instances of a tile read theirs from different banks
    double* sKu = reinterpret_cast<double*>(smem);      // (KP, NQ): Ku1 transposed, zero padded
    double* sLb = sKu + (size_t)KP * NQ;                // (NQ) padded rows can never be violated
    double* sUb = sLb + NQ;
    double* sX = sUb + NQ;                              // (64, KP): [x0; 1; 0..] of this block's instances
    {   // stage with every load in flight before the first LDS write: 16-byte loads of the map (NP, NQ multiples of 16)
        const int hq = NQ / 2, tot2 = KP * hq;
        constexpr int U = 4;
        for (int base = tid; base < tot2; base += U * 256) {
            double2 v[U]; int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                const bool in = idx < tot2;
                const int k = in ? idx / hq : 0, j = 2 * (idx - k * hq);
                off[u] = in ? k * NQ + j : -1;
                v[u] = (in && k <= nx && j < p.NP) ? *reinterpret_cast<const double2*>(p.KuT + (size_t)k * p.NP + j) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (off[u] >= 0) *reinterpret_cast<double2*>(sKu + off[u]) = v[u];
        }
    }
    for (int j = tid; j < NQ; j += 256) { sLb[j] = j < n ? p.lbs[j] : -1e300; sUb[j] = j < n ? p.ubs[j] : 1e300; }
    {
        constexpr int U = 4;
        for (int base = tid; base < 64 * KP; base += U * 256) {
            double v[U]; int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = base + u * 256;
                const int l = idx / KP, k = idx - l * KP;
                const int i2 = blockIdx.x * 64 + l;
                off[u] = l * XS + k;
                v[u] = (idx < 64 * KP && i2 < p.B) ? (k < nx ? p.x0[(size_t)i2 * nx + k] : (k == nx ? 1.0 : 0.0)) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (base + u * 256 < 64 * KP) sX[off[u]] = v[u];
        }
    }
    if (tid < NBIN) hist[tid] = 0;
    __syncthreads();
    const int inst = blockIdx.x * 64 + wave * 16 + col;
    const double* xrow = sX + (size_t)(wave * 16 + col) * XS;
    int cnt = 0;
    // all (up to 8) row tiles of the map at once: one read of [x0; 1] per k-step serves them, their LDS reads are issued
    // together and the MFMAs follow each other on independent accumulators (tile by tile every MFMA waited for its own two
    // LDS reads: 17 k of the kernel's 25 k cycles per workgroup)
    const int NT = NQ / 16;
    d4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = d4{0, 0, 0, 0};
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + kq;
        const double xv = xrow[k];
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < NT) acc[t] = mfma_f64(sKu[k * NQ + 16 * t + col], xv, acc[t]);
    }
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < NT) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 16 * t + kq + 4 * r;
                cnt += (acc[t][r] > sUb[row]) | (acc[t][r] < sLb[row]);
            }
        }
    cnt += __shfl_xor(cnt, 16);
    cnt += __shfl_xor(cnt, 32);
    // the instance goes straight into its bucket's own array (place = the bucket's running count: one atomic per bucket and
    // workgroup); the first pass finds "place p of the order" from the buckets' final sizes itself -- no scatter pass
    __shared__ int base[NBIN];
    const bool mine = kq == 0 && inst < p.B;
    int key = 0, local = 0;
    if (mine) {
        key = min(cnt, NBIN - 1);
        keys[inst] = key;
        local = atomicAdd(&hist[key], 1);
    }
    __syncthreads();
    if (tid < NBIN && hist[tid]) base[tid] = atomicAdd(&bins[tid * BIN_STRIDE], hist[tid]);
    __syncthreads();
    if (mine) seg[(size_t)key * p.B + base[key] + local] = inst;
}

// First launch of a component-split solve (split.hip).  x_unc = Ku1 [x0; 1] of 16 instances per wave as X Ku1' on MFMA tiles
// (A = [x0; 1] of the instances, B = 16 rows of the map), so that lane l holds rows 16 t + (l & 15) of instances (l >> 4) + 4 r:
// 16 lanes cover 16 consecutive rows of an instance and every store below is a run of 128 contiguous bytes.
//   * a component none of whose rows x_unc violates is solved by x_unc itself: its rows of U (and of the kept state, the
//     multipliers) are written here, its part of the objective is zero, and it never reaches the active-set kernels;
//   * the others are put into the bucket of their number of violated rows (what predicts their work), largest first, as
//     cold_key_kernel does for whole instances.
//   * the unconstrained part of the instance's objective, 1/2 [x0;1]' Cq [x0;1], comes off the same tiles (X Cq' on MFMA, then a
//     dot product with [x0;1] across the 16 lanes of a row) and is left in c0out for split_finish_kernel.
// p is the UNSPLIT parameter set (map, bounds, scaling of the whole problem; B instances) plus sp_C, sp_part, sp_steps.
__global__ __launch_bounds__(256) void split_key_kernel(PolParams p, const int* rowmap, int* need, int* bins, int* seg, double* c0out,
                                                        int* redo_count) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    __shared__ int hist[NBIN];
    __shared__ int base[NBIN];
    const int n = p.n, nx = p.nx, C = p.sp_C, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int col = lane & 15, kq = lane >> 4;
    const int KS = (nx + 1 + 3) / 4, KP = 4 * KS, NQ = 16 * ((n + 15) / 16);
    const int XS = KP + 1;
    double* sKu = reinterpret_cast<double*>(smem);      // (KP, NQ): Ku1 transposed, zero padded
    double* sLb = sKu + (size_t)KP * NQ;
    double* sUb = sLb + NQ;
    double* sDd = sUb + NQ;
    double* sX = sDd + NQ;                              // (64, XS): [x0; 1; 0..] of this block's instances
    const int CA = 16 * ((nx + 1 + 15) / 16), CQS = KP + 1;
    double* sCq = sX + 64 * XS;                         // (CA, CQS): Cq, zero padded (odd stride: 16 rows on 16 banks)
    int* sComp = reinterpret_cast<int*>(sCq + (size_t)CA * CQS);  // (NQ): component of the row
    {
        const int hq = NQ / 2, tot2 = KP * hq;
        constexpr int U = 4;
        for (int b0 = tid; b0 < tot2; b0 += U * 256) {
            double2 v[U]; int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = b0 + u * 256;
                const bool in = idx < tot2;
                const int k = in ? idx / hq : 0, j = 2 * (idx - k * hq);
                off[u] = in ? k * NQ + j : -1;
                v[u] = (in && k <= nx && j < p.NP) ? *reinterpret_cast<const double2*>(p.KuT + (size_t)k * p.NP + j) : make_double2(0.0, 0.0);
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (off[u] >= 0) *reinterpret_cast<double2*>(sKu + off[u]) = v[u];
        }
    }
    for (int j = tid; j < NQ; j += 256) {
        sLb[j] = j < n ? p.lbs[j] : -1e300; sUb[j] = j < n ? p.ubs[j] : 1e300; sDd[j] = j < n ? p.d[j] : 0.0;
        sComp[j] = j < n ? (rowmap[j] >> 16) : 0;
    }
    {
        constexpr int U = 4;
        for (int b0 = tid; b0 < 64 * KP; b0 += U * 256) {
            double v[U]; int off[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int idx = b0 + u * 256;
                const int l = idx / KP, k = idx - l * KP;
                const int i2 = blockIdx.x * 64 + l;
                off[u] = l * XS + k;
                v[u] = (idx < 64 * KP && i2 < p.B) ? (k < nx ? p.x0[(size_t)i2 * nx + k] : (k == nx ? 1.0 : 0.0)) : 0.0;
            }
#pragma unroll
            for (int u = 0; u < U; ++u) if (b0 + u * 256 < 64 * KP) sX[off[u]] = v[u];
        }
    }
    // the quadratic form's matrix rides along with the staging: read from global memory inside the product loop, its latency
    // was on the critical path of every workgroup (in-kernel stamps: 6 of a workgroup's 16 us between staging and counts)
    for (int idx = tid; idx < CA * CQS; idx += 256) {
        const int a = idx / CQS, k = idx - a * CQS;
        sCq[idx] = (a <= nx && k <= nx) ? p.Cq[(size_t)a * (nx + 1) + k] : 0.0;
    }
    if (tid < NBIN) hist[tid] = 0;
    if (blockIdx.x == 0 && tid == 0) *redo_count = 0;       // the one counter the last solve's finish kernel could not clear
    __syncthreads();
    const double* xrow = sX + (size_t)(wave * 16 + col) * XS;
    const int NT = NQ / 16;
    d4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; ++t) acc[t] = d4{0, 0, 0, 0};
    for (int s = 0; s < KS; ++s) {
        const int k = 4 * s + kq;
        const double xv = xrow[k];
#pragma unroll
        for (int t = 0; t < 8; ++t)
            if (t < NT) acc[t] = mfma_f64(xv, sKu[k * NQ + 16 * t + col], acc[t]);
    }
    {   // 1/2 v' Cq v, v = [x0; 1]: D[i][a] = sum_k X[i][k] Cq[a][k] for 16 values of a at a time, times v_i[a], summed over a
        const int nx1 = nx + 1;
        double cr[4] = {0.0, 0.0, 0.0, 0.0};
        for (int ta = 0; 16 * ta < nx1; ++ta) {
            const int a = 16 * ta + col;
            d4 cacc = d4{0, 0, 0, 0};
            for (int s = 0; s < KS; ++s) {
                const int k = 4 * s + kq;
                cacc = mfma_f64(xrow[k], sCq[a * CQS + k], cacc);
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) cr[r] += cacc[r] * (a < nx1 ? sX[(size_t)(wave * 16 + kq + 4 * r) * XS + a] : 0.0);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            // over the 16 lanes of a DPP row by rotations (as the counts below): no trips through the LDS crossbar
            cr[r] += dpp_d<0x128>(cr[r]); cr[r] += dpp_d<0x124>(cr[r]); cr[r] += dpp_d<0x122>(cr[r]); cr[r] += dpp_d<0x121>(cr[r]);
            const int inst = blockIdx.x * 64 + wave * 16 + kq + 4 * r;
            if (col == 0 && inst < p.B) c0out[inst] = 0.5 * cr[r];
        }
    }
    // violated rows per (instance, component): 8-bit fields of one 64-bit word per instance (at most 8 components, 128 rows)
    unsigned long long cnt[4] = {0ull, 0ull, 0ull, 0ull};
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < NT) {
            const int row = 16 * t + col;
            const double lo = sLb[row], hi = sUb[row];
            const int sh = 8 * sComp[row];
#pragma unroll
            for (int r = 0; r < 4; ++r) cnt[r] += (unsigned long long)((acc[t][r] > hi) | (acc[t][r] < lo)) << sh;
        }
    // sums over the 16 lanes of an instance = one DPP row: rotations instead of trips through the LDS crossbar (the 8-bit fields
    // cannot carry into each other: a component has at most 64 rows)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        int lo = (int)(unsigned)cnt[r], hi = (int)(unsigned)(cnt[r] >> 32);
        lo += dpp_i<0x128>(lo); hi += dpp_i<0x128>(hi);      // row_ror:8
        lo += dpp_i<0x124>(lo); hi += dpp_i<0x124>(hi);      // row_ror:4
        lo += dpp_i<0x122>(lo); hi += dpp_i<0x122>(hi);      // row_ror:2
        lo += dpp_i<0x121>(lo); hi += dpp_i<0x121>(hi);      // row_ror:1
        cnt[r] = ((unsigned long long)(unsigned)hi << 32) | (unsigned)lo;
    }
    const int inst0 = blockIdx.x * 64 + wave * 16;
    // one lane per (instance, component): done, or into its bucket
    int mk[2] = {0, 0}, ml[2] = {0, 0}, ms[2] = {0, 0};
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        const int idx = col + 16 * u;
        if (idx < 4 * C) {
            const int r = idx / C, comp = idx - r * C;
            const int inst = inst0 + kq + 4 * r;
            const unsigned long long cr = r == 0 ? cnt[0] : (r == 1 ? cnt[1] : (r == 2 ? cnt[2] : cnt[3]));
            const int key = (int)((cr >> (8 * comp)) & 0xFFull);
            if (inst < p.B) {
                const int sub = inst * C + comp;
                if (key == 0) { p.sp_part[sub] = 0.0; p.sp_steps[sub] = 0; need[sub] = 0; }
                else { mk[u] = min(key, NBIN - 1); ms[u] = sub; ml[u] = atomicAdd(&hist[mk[u]], 1); }
            }
        }
    }
    __syncthreads();
    // the workgroup's places in the buckets: asked for here, needed after the stores below (the atomic's round trip to the
    // memory side, ~2 us, hides behind them)
    const bool asks = tid < NBIN && hist[tid] != 0;
    int got = 0;
    if (asks) got = atomicAdd(&bins[tid * BIN_STRIDE], hist[tid]);
    // rows of the components that x_unc solves
#pragma unroll
    for (int t = 0; t < 8; ++t)
        if (t < NT) {
            const int row = 16 * t + col;
            const int sh = 8 * sComp[row];
            const double dd = sDd[row];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int inst = inst0 + kq + 4 * r;
                if (row < n && inst < p.B && ((cnt[r] >> sh) & 0xFFull) == 0ull) {
                    const double xv = acc[t][r];
                    p.U[(size_t)inst * n + row] = dd * xv;
                    if (p.keep_state) { p.z[(size_t)inst * p.NP + row] = xv; p.w[(size_t)inst * p.NP + row] = 0.0; }
                    if (p.info.lam) p.info.lam[(size_t)inst * n + row] = 0.0;
                }
            }
        }
    if (asks) base[tid] = got;
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 2; ++u)
        if (mk[u]) seg[(size_t)mk[u] * ((size_t)p.B * C) + base[mk[u]] + ml[u]] = ms[u];
}

int launch_split_key(const PolParams& p, const int* rowmap, int* need, int* bins, int* seg, double* c0out, int* redo_count,
                     hipStream_t st) {
    if (p.sp_C > 8 || p.n > 128) return -1;
    const int kp = 4 * ((p.nx + 4) / 4), nq = 16 * ((p.n + 15) / 16);
    const int ca = 16 * ((p.nx + 1 + 15) / 16);
    const size_t kl = ((size_t)kp * nq + 3 * nq + 64 * (size_t)(kp + 1) + (size_t)ca * (kp + 1)) * sizeof(double) + nq * sizeof(int);
    if (kl > 64 * 1024) return -1;
    hipLaunchKernelGGL(split_key_kernel, dim3(ceil_div(p.B, 64)), dim3(256), kl, st, p, rowmap, need, bins, seg, c0out, redo_count);
    return (int)hipGetLastError();
}

// The two fast passes: pass A (8 lanes per instance, KMAX = 16) takes the whole batch (large cold batches: in cold-start
// order); what outgrows it is queued on the device (q1) with its working state and taken by pass B (KMAX = 32; 32 or 16
// lanes per instance by the length of the queue); what outgrows that is queued in q2 for the 64-row pass.
// Returns 0 on launch, -1 if the shape does not fit this path, a hipError_t (> 0) otherwise.
int launch_polish_quad(const PolParams& p, int* q1, int* q2, int* counters, hipStream_t st, int stop_after) {
    if (!p.polish || p.refine > 0 || p.nx > 64 || p.n > 128) return -1;
    hipError_t e = hipMemsetAsync(counters, 0, (96 + NBIN * BIN_STRIDE) * sizeof(int), st);       // queue counters, buckets, cursors
    if (e != hipSuccess) return (int)e;
    PolParams a = p;
    a.qin = nullptr; a.qin_count = nullptr; a.qout = q1; a.qout_count = counters;
    a.hand_in = nullptr; a.hand_out = p.cold ? p.hand_buf1 : nullptr; a.hand_out_cap = p.B; a.hand_zw = 0;
    const int kp = 4 * ((p.nx + 4) / 4), nq = 16 * ((p.n + 15) / 16);
    const size_t kl = ((size_t)kp * nq + 2 * nq + 64 * (size_t)(kp + 1)) * sizeof(double);
    // the ordering pays once the device is full (lock-step efficiency); below that a launch lasts as long as its heaviest
    // block either way and the two extra kernels only add to it (measured, per solve: 32,768 instances 276 us with / 322 us
    // without; 16,384: 242 / 223; 8,192: 219 / 200)
    // (round 2, late: the ordering kernel is one launch and 30 us now, and pays as soon as the first pass needs more than
    // ONE round of workgroups -- 20,000 instances: 0.227 ms unordered, 0.18 ordered; 16,384, one round: 0.157 / 0.177)
    if (p.cold && p.B > 64 * 256 && kl <= 64 * 1024) {
        // keys in q2 (written again only by the second pass), buckets / cursors / count in the counter block,
        // the order in the last slot of the int workspace
        int* keys = q2; int* seg = counters + 96 + NBIN * BIN_STRIDE;
        int* bins = counters + 96;
        hipLaunchKernelGGL(cold_key_kernel, dim3(ceil_div(p.B, 64)), dim3(256), kl, st, p, keys, bins, seg);
        // (Round 1 routed the heaviest buckets past the first pass, because an instance that outgrew it had to start again;
        // with the hand-over every instance starts in the first pass and continues where it stopped: 402 us routed, 375 us not.)
        if ((e = hipGetLastError()) != hipSuccess) return (int)e;
        a.seg = seg; a.seg_counts = bins;
    }
    if (stop_after == 2) return 0;
    // A small cold batch (a 1/8 shard of the headline batch and below) lasts as long as its slowest instance, whatever the
    // launch holds: ONE pass with 16 lanes per instance and 32 slots takes it all the way (32 instances per workgroup: 8,192
    // instances are one workgroup per CU; shorter steps than 8 lanes, and ~no instance outgrows 32 rows), its rare overflow
    // goes straight to the 64-row pass with the state record that pass B would have written.  Two launches fewer, and no
    // second staging + latency chain behind the first (8,192 instances: 155 -> see DESIGN.md section 7).
    if (p.cold && p.B <= SMALL_BATCH && p.hand_buf2 && stop_after != 2) {
        PolParams s16 = p;
        s16.qin = nullptr; s16.qin_count = nullptr; s16.qout = q2; s16.qout_count = counters + 1;
        s16.hand_in = nullptr; s16.hand_out = p.hand_buf2; s16.hand_out_cap = p.hand_cap2; s16.hand_zw = 0;
        const int rc16 = launch_oct_rp<16, 32, 512>(s16, p.B, st);
        if (rc16 >= 0) return rc16;
    }
    // fewer blocks of 64 instances than the device has CUs (a 1/8 shard of the headline batch): half-size workgroups, so that
    // the launch covers twice as many CUs with one wave per SIMD -- the latency of the heaviest block is what such a launch lasts
    // (half-size workgroups only while they still fit the device in ONE round, 32 instances x 256 CUs: between 8,193 and 16,383
    // instances they were two rounds -- 12,288 instances took 0.220 ms, longer than 16,384 -- where the full-size ones are one)
    int rc = p.B <= 32 * 256 ? launch_oct_rp<8, 16, 256>(a, p.B, st) : launch_oct_rp<8, 16, 512>(a, p.B, st);
    if (rc != 0 || stop_after == 1) return rc;
    PolParams b = p;
    b.qin = q1; b.qin_count = counters; b.qout = q2; b.qout_count = counters + 1;
    b.hand_in = a.hand_out; b.hand_out = p.cold ? p.hand_buf2 : nullptr; b.hand_out_cap = p.hand_cap2; b.hand_zw = 0;
    // warm solves keep the 32-lane kernel alone: with the iterate registers the 16-lane body spills (166 VGPRs)
    if (!p.cold) return launch_oct_rp<32, 32, 512>(b, p.B, st);
    // The length of the queue is only known on the device, so BOTH instantiations are launched and each one gates itself on
    // it (the other's workgroups leave after one load).  A long queue (the device is full several times over: many binding
    // rows per instance) is taken 16 lanes per instance, four instances per wave -- throughput; a short one (the headline:
    // 2-3 % of the batch) 32 lanes per instance -- latency of the slowest instance.  Both share the 32-slot working set and
    // the state records.  Measured, 65,536 instances, x0 scaled by 1.0 / 1.5 / 2.5: 32 lanes only 0.385 / 0.941 / 3.39 ms,
    // gated pair 0.383 / 0.877 / 3.16 ms.  (Both bodies inlined into ONE kernel behind a uniform branch were measured too: the
    // 16-lane body then loses what it gains, 0.941 / 3.37 ms, to the joint register allocation.)
    constexpr int LONG_QUEUE = 8192;
    PolParams b16 = b;
    b16.q_gate = LONG_QUEUE;
    const int rc16 = launch_oct_rp<16, 32, 512>(b16, p.B, st);
    if (rc16 != 0) return rc16;
    b.q_gate = -LONG_QUEUE;
    return launch_oct_rp<32, 32, 512>(b, p.B, st);
}
#ifdef OCT_PROBE
namespace { 
template __global__ void polish_oct_kernel<8, 8, 16, 768, true>(PolParams);
template __global__ void polish_oct_kernel<8, 8, 16, 1024, true>(PolParams);
template __global__ void polish_oct_kernel<32, 2, 32, 1024, true>(PolParams);
template __global__ void polish_oct_kernel<16, 4, 16, 1024, true>(PolParams);
template __global__ void polish_oct_kernel<16, 4, 32, 1024, true>(PolParams);
}
#endif

int launch_polish_oct64(const PolParams& p, hipStream_t st) {
    if (!p.polish || p.refine > 0 || p.nx > 64 || p.n > 128) return -1;
    PolParams c = p;
    c.hand_in = p.cold ? p.hand_buf2 : nullptr; c.hand_out = nullptr; c.hand_out_cap = 0; c.hand_zw = p.cold;
    return launch_oct_rp<64, 64, 512>(c, p.B, st);
}
