// Parameters shared by the two polish kernels (polish.hip: one wave per instance, any active-set size;
// polish_quad.hip: four lanes per instance, Hs^-1 resident in LDS, small active sets).
#pragma once
#include "common.h"

// magnitude of a row's FINITE bounds (>= 1): what its tolerances scale with.  An infinite side (u_min = -inf: the row is
// bounded on one side only) must not turn the tolerance of the finite side into infinity
__device__ __forceinline__ double bound_scale(double lo, double hi) {
    const double a = isfinite(hi) ? fabs(hi) : 0.0, b = isfinite(lo) ? fabs(lo) : 0.0;
    return fmax(1.0, fmax(a, b));
}

struct PolParams {
    const double *Hs, *Hi, *FsT, *KuT, *lbs, *ubs, *d;     // FsT, KuT: (nx+1, NP)
    const double* x0;
    double *z, *w;                 // [Bcap][NP] row-major
    const int* it_ws; const double *rp_ws, *rd_ws;
    const int* cv_ws;              // [B]: 1 if ADMM met its eps criterion
    int* need_big;                 // [B]: 1 if the fast pass overflowed KMAX
    double* U;                     // [B][n]
    mpcqp_info info;
    int B, n, nx, NP, dev;
    int polish, refine, max_steps, pass;   // pass 0: fast, 1: big
    double rho;
    long st_nn, st_k, st_np, st_n;  // one model per instance: element strides of (Hs,Hi), (FsT,KuT), (lbs,ubs), d; 0 = shared
    const int* mflag;              // factorisation failure flags: one per model (st_nn != 0) or the shared model's single flag
    // hand-over between the passes of a cold solve: an instance whose slot list is full (need_big == 3) leaves its whole
    // working state -- S^-1, multipliers, right-hand sides, rows, step count: K*K + 4K + 2 doubles for K slots -- in hand_out
    // at the index of its place in the output queue; the next pass (twice the slots) reads it from hand_in at the index of
    // its place in the input queue and continues instead of starting again.  nullptr = no hand-over.
    double* hand_out; int hand_out_cap;     // entries available in hand_out (beyond that the next pass starts cold)
    const double* hand_in;
    int hand_zw;                   // giver: leave the full working SET in (z, w) (z on the bound, w = +-1) for the wave-per-instance
                                   // kernel, whose factorisation is a Cholesky factor; receiver (that kernel): instances with
                                   // need_big == 3 start from the guess in (z, w) although the solve is cold
    double *hand_buf1, *hand_buf2; int hand_cap2;      // the handle's buffers (first -> second pass, second -> third pass)
    int q_gate;                    // > 0: run only if the queue holds at least this many; < 0: only if fewer than -q_gate; 0: always
    int cold; int admm_ran; const double* Cq;                      // 1: no (z, w) to start from -- begin with the empty active set
    const int *qin, *qin_count;    // instance queue to take the work from (nullptr: instances 0..B-1)
    const int *seg, *seg_counts;   // cold-start order without a scatter pass: bucket k's instances in seg[k * B ..], its size in
                                   // seg_counts[32 k]; the place p of the order (largest bucket first) is looked up by the reader
    int *qout, *qout_count;        // queue for the instances this pass could not finish (nullptr: none)
    unsigned long long* prof;      // debugging only (POLISH_PROF builds): per-phase cycle sums, else nullptr
    int keep_state;                // 1: leave the solution in (z, w) for a later warm start / shift; 0: skip those writes
    // component split (split.hip; SPLIT instantiations of polish_oct_kernel): B counts SUB-instances (instance * sp_C +
    // component), n is the (uniform, padded) number of rows of a component, and Hi [C][n][sp_ld], KuT [C][nx+1][sp_ld], lbs / ubs /
    // d [C][sp_ld] are packed per component; x0, U, z, w, info stay per instance and in the row order of the whole problem
    int sp_C, sp_ld, sp_n, sp_NP;  // components (0: not split), row stride of the packed vectors, rows / padded rows of the whole problem
    const int* sp_perm;            // [C][sp_ld]: component row -> row of the whole problem, -1 = padding
    const double* sp_img;          // all of it (Hi | lbs | ubs | d | bounds -+ tolerance | perm | KuT) laid out as the kernels' LDS for rows of sp_ld
    double* sp_part; int* sp_steps;                // per sub-instance: its part of the objective, its steps
    int seg_lo, seg_hi;            // split: the buckets (keys) a launch over the cold-start order takes
    int *redo_q, *redo_count;      // split, LAST pass of the cascade: the instance of a component it could not finish is appended here
                                   // (once per failing component; split_tail_kernel skips the repeats) -- solved again as a whole
};

// fast path; returns MPCQP_ENOTIMPL if the shape does not fit it (the caller then uses the wave kernel)
int launch_polish_quad(const PolParams& p, int* q1, int* q2, int* counters, hipStream_t st, int stop_after = 0);   // diagnostics: 1 = after the first pass, 2 = after the cold-start ordering
// third chance: a whole wave per instance and up to 64 active rows with the same explicit-inverse kernel (the queue of the
// second pass in p.qin, what outgrows it into p.qout); -1 if the shape does not fit
int launch_polish_oct64(const PolParams& p, hipStream_t st);
// split.hip: the cold solve of a model with independent components; -1 if it does not apply (the caller goes on unsplit)
int launch_polish_split(mpcqp_handle h, const PolParams& p, hipStream_t st, int stop_after);
// polish_quad.hip: one pass of the split cascade over p.B sub-instances (or p.qin): slots = 8, 16 or 32
int launch_oct_split(const PolParams& p, int slots, hipStream_t st);
// polish_quad.hip: x_unc for every instance, the components it solves written out, the others bucketed by expected work
int launch_split_key(const PolParams& p, const int* rowmap, int* need, int* bins, int* seg, double* c0out, int* redo_count,
                     hipStream_t st);
