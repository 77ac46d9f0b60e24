// Component split of the shared-model box QP.
//
// The condensed Hessian H = Gamma'Qbar Gamma + Rbar couples two inputs only if the dynamics and the weights do.  Models
// built from independent axes -- the reference's own planners are: Launch3DPlanner.py:61-108 rolls x, y and z out as three
// separate double integrators, and the quadrotor hover linearisation of BASELINE.json configs[2] is four chains (thrust /
// height, roll / y, pitch / x, yaw) -- give an H that is block diagonal under a permutation of the rows, and then
//      min 1/2 U'HU + q'U,  lb <= U <= ub
// IS a set of independent box QPs, one per block: the minimiser is the stack of theirs, the objective the sum.  The dual
// active-set method costs O(rows x active bounds) per step and its steps form a dependent chain, so C components of n/C
// rows are C times fewer flops per step, C times shorter chains, and a Hi that is C times smaller in LDS.
//
// split_setup (after the factorisation of a shared model, on the synchronous set-up call): the coupling graph of the
// Jacobi-scaled Hessian Hs and of its inverse Hi (entries at or below sp_tol relative to the unit diagonal -- rounding
// noise of the assembly, 1e-15 on configs[2] against couplings >= 0.25 -- do not count), its connected components on the
// host (n <= 128: a union-find over 16 k entries), then the per-component blocks of Hi, rows of the map x_unc = Ku [x0;1],
// bounds and scaling packed on the device.  launch_polish_split runs the cold solve on batch x C SUB-instances with the
// SPLIT instantiations of polish_oct_kernel (polish_quad.hip): 8 working-set slots -> 16 -> 32, each pass handing its
// state to the next; a component that cannot be finished that way sends its whole instance to the any-size kernel.
//
// Reference counterpart: none as code (cvxopt.solvers.qp sees one dense P, LaunchPlannerQPFormulation.py:173); the
// structure it exploits is the reference's per-axis prediction matrices.
#include "polish.h"
#include <algorithm>
#include <numeric>
#include <vector>
#include <cmath>

// polish.hip: per-instance record, whole-instance redo, and the next solve's counters zeroed
int launch_split_finish(const PolParams& p, const int* need, const double* c0, int* redo, int* redo_count, int* zero, int nzero,
                        hipStream_t st);

namespace {

constexpr int NBIN = 32, BIN_STRIDE = 32;       // as polish_quad.hip (cold-start buckets)

// Hi, the map, the bounds and the scaling of every component, gathered through perm [C][ld] (-1 = padding row)
__global__ void split_pack_kernel(const double* Hi, const double* KuT, const double* lbs, const double* ubs, const double* d,
                                  const int* perm, int C, int ns, int ld, int n, int NP, int nx1, double* oHi, double* oKuT,
                                  double* olb, double* oub, double* od, double* img) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x, gsz = gridDim.x * blockDim.x;
    for (int idx = gid; idx < C * ns * ld; idx += gsz) {        // rows of ld doubles, padding columns zero
        const int c = idx / (ns * ld), r = (idx / ld) % ns, j = idx % ld;
        const int gr = perm[c * ld + r], gj = perm[c * ld + j];
        oHi[idx] = (gr >= 0 && gj >= 0) ? Hi[(size_t)gr * n + gj] : 0.0;
    }
    for (int idx = gid; idx < C * nx1 * ld; idx += gsz) {
        const int c = idx / (nx1 * ld), k = (idx / ld) % nx1, j = idx % ld;
        const int gj = perm[c * ld + j];
        oKuT[idx] = gj >= 0 ? KuT[(size_t)k * NP + gj] : 0.0;
    }
    for (int idx = gid; idx < C * ld; idx += gsz) {
        const int gj = perm[idx];
        olb[idx] = gj >= 0 ? lbs[gj] : -1e300;      // padding rows: never violated, never written
        oub[idx] = gj >= 0 ? ubs[gj] : 1e300;
        od[idx] = gj >= 0 ? d[gj] : 0.0;
    }
    // the image: Hi (C ns ld) | lbs | ubs | d (C ld each) | (lo - tol, hi + tol) pairs (2 C ld) | perm as ints (C ld / 2 doubles) |
    // KuT (C nx1 ld) -- OCT_LAYOUT of polish_quad.hip with LD = NQ = ld
    double* iLb = img + (size_t)C * ns * ld;
    double* iUb = iLb + C * ld;
    double* iD = iUb + C * ld;
    double* iBT = iD + C * ld;
    int* iPerm = reinterpret_cast<int*>(iBT + 2 * C * ld);
    double* iKu = iBT + 2 * C * ld + C * ld / 2;
    for (int idx = gid; idx < C * ns * ld; idx += gsz) {
        const int c = idx / (ns * ld), r = (idx / ld) % ns, j = idx % ld;
        const int gr = perm[c * ld + r], gj = perm[c * ld + j];
        img[idx] = (gr >= 0 && gj >= 0) ? Hi[(size_t)gr * n + gj] : 0.0;
    }
    for (int idx = gid; idx < C * ld; idx += gsz) {
        const int gj = perm[idx];
        const double lo = gj >= 0 ? lbs[gj] : -1e300, hi = gj >= 0 ? ubs[gj] : 1e300;
        iLb[idx] = lo; iUb[idx] = hi; iD[idx] = gj >= 0 ? d[gj] : 0.0;
        const double tol = 1e-12 * bound_scale(lo, hi);
        iBT[2 * idx] = fmax(lo - tol, -1e300); iBT[2 * idx + 1] = fmin(hi + tol, 1e300);    // finite: the keyed scan packs bits into x - bound (polish_quad.hip)
        iPerm[idx] = gj;
    }
    for (int idx = gid; idx < C * nx1 * ld; idx += gsz) {
        const int c = idx / (nx1 * ld), k = (idx / ld) % nx1, j = idx % ld;
        const int gj = perm[c * ld + j];
        iKu[idx] = gj >= 0 ? KuT[(size_t)k * NP + gj] : 0.0;
    }
}

template <typename T>
int sp_alloc(mpcqp_handle h, T** p, size_t count) {
    if (*p) { (void)hipFree(*p); *p = nullptr; }
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(p), count * sizeof(T)));
    HIPCHK(h, hipMemset(*p, 0, count * sizeof(T)));
    return MPCQP_OK;
}

// int workspace for BC sub-instances of B instances:
//   need[BC] | queue 1 [BC] | queue 2 [BC] | steps[BC] | counters (96 + 32 x 32) | buckets [32][BC]
// No memset in front of a solve: every solve leaves the counters zero for the next one -- the finish kernel clears all of
// them but the redo queue's (which the kernel after it still reads), the key kernel of the next solve clears that one.  The
// sequence of launches therefore does not depend on host state and can be replayed from a HIP graph.
constexpr int NCNT = 96 + NBIN * BIN_STRIDE;
constexpr int REDO = 2;         // counters[REDO]: length of the redo queue
struct SplitWs { int *need, *q1, *q2, *steps, *counters, *seg; };
SplitWs split_ws(mpcqp_handle h) {
    const size_t BC = h->sp_bcap;
    SplitWs w;
    w.need = h->sp_iws; w.q1 = w.need + BC; w.q2 = w.q1 + BC; w.steps = w.q2 + BC;
    w.counters = w.steps + BC; w.seg = w.counters + NCNT;
    return w;
}

int split_ensure_batch(mpcqp_handle h, int B) {
    const int need = ceil_div(B, 64) * 64 * h->sp_C;
    if (need <= h->sp_bcap) return MPCQP_OK;
    if (h->pinned) { h->err = "solve: a larger batch would re-allocate workspaces that a recorded graph points into (mpcqp_pin)"; return MPCQP_ESTATE; }
    const size_t BC = need;
    int rc = sp_alloc(h, &h->sp_iws, 4 * BC + NCNT + (size_t)NBIN * BC);            // zeroed: the counters start clean
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_part, BC + BC / h->sp_C);              // parts per sub-instance | c0 per instance
    // hand-over records (polish_quad.hip): K*K + 4K + 2 doubles for K slots; sized for half (a quarter) of the sub-instances
    // outgrowing 8 (16) slots -- beyond that a sub-instance starts its next pass from the empty set
    h->sp_cap1 = (int)(BC / 2); h->sp_cap2 = (int)(BC / 4);
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_hb1, (size_t)h->sp_cap1 * (8 * 8 + 4 * 8 + 2));
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_hb2, (size_t)h->sp_cap2 * (16 * 16 + 4 * 16 + 2));
    if (rc != MPCQP_OK) { h->sp_bcap = 0; return rc; }
    h->sp_bcap = need; h->sp_wsC = h->sp_C; h->sp_wsld = h->sp_ld;
    return MPCQP_OK;
}

}  // namespace

void split_free(mpcqp_handle h) {
    void* ptrs[] = {h->sp_perm, h->sp_rowmap, h->sp_Hi, h->sp_KuT, h->sp_lbs, h->sp_ubs, h->sp_d, h->sp_img, h->sp_iws, h->sp_part, h->sp_hb1, h->sp_hb2};
    for (void* p : ptrs) if (p) (void)hipFree(p);
    h->sp_perm = h->sp_rowmap = h->sp_iws = nullptr;
    h->sp_Hi = h->sp_KuT = h->sp_lbs = h->sp_ubs = h->sp_d = h->sp_img = h->sp_part = h->sp_hb1 = h->sp_hb2 = nullptr;
    h->sp_C = h->sp_ns = h->sp_ld = h->sp_bcap = 0;
}

int split_setup(mpcqp_handle h, hipStream_t st) {
    const int n = h->n, nx1 = h->dims.nx + 1;
    h->sp_C = 0;
    if (!h->sp_enable || h->dims.n_models != 1 || n < 4 || n > 128) return MPCQP_OK;
    std::vector<double> Hs((size_t)n * n), Hi((size_t)n * n);
    HIPCHK(h, hipMemcpyAsync(Hs.data(), h->Hs, Hs.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(Hi.data(), h->Hi, Hi.size() * sizeof(double), hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipStreamSynchronize(st));
    // connected components of the coupling graph (Hs has a unit diagonal; Hi is compared with its own diagonal)
    std::vector<int> parent(n);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    const double tol = h->sp_tol;
    for (int i = 0; i < n; ++i)
        for (int j = i + 1; j < n; ++j) {
            const double a = std::fabs(Hs[(size_t)i * n + j]), b = std::fabs(Hs[(size_t)j * n + i]);
            const double s = std::sqrt(std::fabs(Hi[(size_t)i * n + i] * Hi[(size_t)j * n + j]));
            const double c = std::fabs(Hi[(size_t)i * n + j]), e = std::fabs(Hi[(size_t)j * n + i]);
            const bool coupled = !(a <= tol) || !(b <= tol) || !(c <= tol * s) || !(e <= tol * s);   // NaN counts as coupled
            if (coupled) { const int ri = find(i), rj = find(j); if (ri != rj) parent[std::max(ri, rj)] = std::min(ri, rj); }
        }
    std::vector<int> comp_of(n, -1), size;
    int C = 0;
    for (int i = 0; i < n; ++i) {               // components numbered by their first row
        const int r = find(i);
        if (comp_of[r] < 0) { comp_of[r] = C++; size.push_back(0); }
        comp_of[i] = comp_of[r];
        ++size[comp_of[i]];
    }
    if (C < 2 || C > 8) return MPCQP_OK;        // split_key_kernel counts violated rows per component in 8-bit fields of one word
    int ns = *std::max_element(size.begin(), size.end());
    ns += ns & 1;                               // even: 16-byte rows in the staging loops
    if (ns > 64) return MPCQP_OK;               // what the SPLIT instantiations hold (polish_quad.hip: launch_oct_split)
    const int ld = ns <= 16 ? 16 : (ns <= 32 ? 32 : 64);     // = the LDS row of the 8-lane passes (polish_quad.hip: NQ), so that staging is a flat copy
    {   // LDS of the widest pass must fit (the formula of launch_oct_c): otherwise the split is not worth having
        const size_t NQ = ns <= 16 ? 16 : (ns <= 32 ? 32 : 64);
        const size_t kux = (size_t)C * nx1 * NQ * 8 + 128 * (size_t)h->dims.nx * 8, xch = 128 * 12 * 20;   // 4-lane pass: 128 sub-instances per workgroup
        const size_t lds = ((size_t)C * ns * NQ + 5 * C * NQ + C * NQ / 2) * 8 + std::max(kux, xch);
        if (lds > 150 * 1024) return MPCQP_OK;
    }
    std::vector<int> perm((size_t)C * ld, -1), rowmap(h->NP, -1), fill(C, 0);
    for (int i = 0; i < n; ++i) {
        const int c = comp_of[i], l = fill[c]++;
        perm[(size_t)c * ld + l] = i;
        rowmap[i] = (c << 16) | l;
    }
    if (h->sp_perm) { (void)hipFree(h->sp_perm); h->sp_perm = nullptr; }
    if (h->sp_rowmap) { (void)hipFree(h->sp_rowmap); h->sp_rowmap = nullptr; }
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->sp_perm), perm.size() * sizeof(int)));
    HIPCHK(h, hipMalloc(reinterpret_cast<void**>(&h->sp_rowmap), rowmap.size() * sizeof(int)));
    HIPCHK(h, hipMemcpy(h->sp_perm, perm.data(), perm.size() * sizeof(int), hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->sp_rowmap, rowmap.data(), rowmap.size() * sizeof(int), hipMemcpyHostToDevice));
    int rc = sp_alloc(h, &h->sp_Hi, (size_t)C * ns * ld);
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_KuT, (size_t)C * nx1 * ld);
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_lbs, (size_t)C * ld);
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_ubs, (size_t)C * ld);
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_d, (size_t)C * ld);
    if (rc == MPCQP_OK) rc = sp_alloc(h, &h->sp_img, (size_t)C * ns * ld + 5 * (size_t)C * ld + (size_t)C * ld / 2 + (size_t)C * nx1 * ld + 2);
    if (rc != MPCQP_OK) return rc;
    hipLaunchKernelGGL(split_pack_kernel, dim3(64), dim3(256), 0, st, h->Hi, h->KuT, h->lbs, h->ubs, h->d, h->sp_perm, C, ns,
                       ld, n, h->NP, nx1, h->sp_Hi, h->sp_KuT, h->sp_lbs, h->sp_ubs, h->sp_d, h->sp_img);
    HIPCHK(h, hipGetLastError());
    h->sp_C = C; h->sp_ns = ns; h->sp_ld = ld;
    return MPCQP_OK;
}

int launch_polish_split(mpcqp_handle h, const PolParams& pin, hipStream_t st, int stop_after) {
    if (h->sp_C < 2 || !pin.cold || !pin.polish || pin.refine > 0 || pin.st_nn || !pin.Cq) return -1;
    const int C = h->sp_C, Binst = pin.B;
    if ((long)Binst * C > (1L << 30)) return -1;
    if (h->sp_bcap && (h->sp_wsC != C || h->sp_wsld != h->sp_ld)) h->sp_bcap = 0;     // workspace of another split: size it again
    int rc = split_ensure_batch(h, Binst);
    if (rc != MPCQP_OK) return (int)hipErrorOutOfMemory;
    const SplitWs w = split_ws(h);
    double* c0 = h->sp_part + h->sp_bcap;
    // leaving early (a failed launch; the timing modes of bench.py that stop before the finish kernel): the counters are
    // zeroed here instead of by that kernel
    auto leave = [&](int code) {
        (void)hipMemsetAsync(w.counters, 0, NCNT * sizeof(int), st);
        return code;
    };
    PolParams p = pin;
    p.sp_C = C; p.sp_ld = h->sp_ld; p.sp_n = pin.n; p.sp_NP = pin.NP; p.sp_perm = h->sp_perm; p.sp_img = h->sp_img;
    p.sp_part = h->sp_part; p.sp_steps = w.steps;
    p.Hi = h->sp_Hi; p.KuT = h->sp_KuT; p.lbs = h->sp_lbs; p.ubs = h->sp_ubs; p.d = h->sp_d;
    p.n = h->sp_ns; p.B = Binst * C;
    p.need_big = w.need;
    p.seg = nullptr; p.seg_counts = nullptr; p.q_gate = 0; p.hand_zw = 0; p.redo_q = nullptr; p.redo_count = nullptr;
    // x_unc for everyone: components it solves are finished there, the rest ordered by the number of rows it violates
    {
        PolParams k = pin;
        k.sp_C = C; k.sp_part = h->sp_part; k.sp_steps = w.steps;
        rc = launch_split_key(k, h->sp_rowmap, w.need, w.counters + 96, w.seg, c0, w.counters + REDO, st);
        if (rc != 0) return leave(rc);
    }
    if (stop_after == 2) return leave(0);
    // 8 slots -> 16 -> 32, each pass taking the queue (and the working states) the one before left.
    // (Measured and dropped: components that violate 6 rows or more -- they outgrow 8 slots almost surely -- started in the 16-slot
    // pass on a second stream BESIDE the first pass: the fork / join through events cost ~40 us of gaps per solve, more than the
    // 35 us the queue-fed 16-slot pass lasts.)
    // A small batch (one round of workgroups in any pass) lasts as long as its longest chain of dependent steps, whatever
    // the launch holds: it starts in the 16-slot pass -- one launch and one hand-over fewer for the components that outgrow 8
    const bool small = (long)Binst * C <= h->sp_small;
    PolParams a = p;
    a.seg = w.seg; a.seg_counts = w.counters + 96; a.seg_lo = 1; a.seg_hi = NBIN - 1;
    a.qin = nullptr; a.qin_count = nullptr; a.hand_in = nullptr;
    if (small) { a.qout = w.q2; a.qout_count = w.counters + 1; a.hand_out = h->sp_hb2; a.hand_out_cap = h->sp_cap2; }
    else { a.qout = w.q1; a.qout_count = w.counters; a.hand_out = h->sp_hb1; a.hand_out_cap = h->sp_cap1; }
    // up to 4,096 x 4 sub-instances the 16-slot pass runs with 16 lanes per sub-instance (two rows per lane: a shorter step, four
    // per wave): 4,096 instances 70 -> 63 us, 1,024: 62 -> 57; equal at 8,192 x 4, slower beyond (12,288: 79 -> 93 us)
    const bool wide = small && (long)Binst * C <= 20000 && h->sp_ld == 32;
    rc = launch_oct_split(a, small ? (wide ? 17 : 16) : 8, st);
    if (rc != 0) return leave(rc);
    if (stop_after == 1) return leave(0);      // timing of the first pass alone (bench.py): the batch is left partly solved
    if (!small) {
        PolParams b = p;
        b.qin = w.q1; b.qin_count = w.counters; b.qout = w.q2; b.qout_count = w.counters + 1;
        b.hand_in = h->sp_hb1; b.hand_out = h->sp_hb2; b.hand_out_cap = h->sp_cap2;
        // the queue of the 8-slot pass is a few thousand long chains: 16 lanes per sub-instance shorten their steps (65,536
        // instances: 207 -> 199.5 us per solve; 32,768: 136.5 -> 130.3)
        rc = launch_oct_split(b, h->sp_ld == 32 ? 17 : 16, st);
        if (rc != 0) return leave(rc);
    }
    PolParams c = p;
    c.qin = w.q2; c.qin_count = w.counters + 1; c.qout = nullptr; c.qout_count = nullptr;
    c.hand_in = h->sp_hb2; c.hand_out = nullptr; c.hand_out_cap = 0;
    c.redo_q = w.q1; c.redo_count = w.counters + REDO;        // queue 1 is free again: what this pass cannot finish is solved again as a whole
    rc = launch_oct_split(c, 32, st);
    if (rc != 0) return leave(rc);
    // the instances' records from their components' parts; an instance with a component the cascade could not finish (more than
    // 32 active rows in a component of more than 32 rows, or a numerical guard: the last pass queued it) is solved again as a
    // whole, unsplit, from the empty set, by the any-size method -- both in one launch (split_tail_kernel)
    PolParams f = pin;
    f.pass = 1; f.qin = nullptr; f.qin_count = nullptr; f.qout = nullptr; f.qout_count = nullptr;
    f.hand_in = nullptr; f.hand_out = nullptr; f.hand_zw = 0; f.seg = nullptr;
    f.sp_C = C; f.sp_part = h->sp_part; f.sp_steps = w.steps;
    rc = launch_split_finish(f, w.need, c0, w.q1, w.counters + REDO, w.counters, NCNT, st);
    if (rc != 0) return leave(rc);
    return 0;
}
