// Internal definitions shared by the libmpcqp translation units (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <string>
#include <cstdio>
#include "../../include/mpcqp.h"

typedef double d4 __attribute__((ext_vector_type(4)));

// v_mfma_f64_16x16x4_f64: D(16x16) = A(16x4) B(4x16) + C.
//   lane l holds A[l&15][l>>4], B[l>>4][l&15]; D reg r of lane l = D[(l>>4) + 4r][l&15].
// Consequence used throughout: the D registers of row tile t, reg r ARE the B operand of
// k-step 4t+r of a following product over the same 16 columns -- no lane movement.
__device__ __forceinline__ d4 mfma_f64(double a, double b, d4 c) {
    return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
}

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }

struct mpcqp_handle_s {
    int dev = 0;
    mpcqp_dims dims{};
    int n = 0, p = 0;        // decision size N*nu, stacked state size N*nx
    int NT = 0, NP = 0;      // row tiles of 16, padded rows
    int S = 0;               // k-steps of 4 over the n columns
    int SX = 0;              // k-steps over [x0; 1]
    // caller-owned problem data
    const double *A = nullptr, *B = nullptr, *Q = nullptr, *R = nullptr, *P = nullptr;
    const double *xref = nullptr, *uref = nullptr, *umin = nullptr, *umax = nullptr;
    int per_stage = 0;
    long xref_stride = 0, uref_stride = 0, bnd_stride = 0;   // per-model data (n_models > 1): doubles between models, 0 = shared
    // per-model workspace (n_models copies)
    double *Gam = nullptr, *QG = nullptr, *Phi = nullptr;   // (p,n) (p,n) (p,nx)
    double *H = nullptr, *F = nullptr, *f0 = nullptr;       // (n,n) (n,nx) (n)
    // shared-model solver data (model 0)
    double *d = nullptr, *Hs = nullptr, *M = nullptr, *Hi = nullptr, *W1 = nullptr, *W2 = nullptr;
    bool gam_valid = false;        // Gam / QG hold the prediction matrices of the current model (the per-instance fast assembly skips them)
    double *facL = nullptr, *facBlk = nullptr;     // shared model: scratch of the blocked inverses (blocked_setup.hip)
    double *Fs1 = nullptr;      // (n, nx+1): [Fs | f0s]   qs = Fs1 [x0;1]
    double *Cq = nullptr;       // (nx+1, nx+1): Fs1' Ku1 (shared model): qs'x_unc = [x0;1]' Cq [x0;1]
    double *Ku1 = nullptr;      // (n, nx+1): [Ku | ku]    x_unc = Ku1 [x0;1]
    double *FsT = nullptr, *KuT = nullptr;   // the same two maps transposed, (nx+1, NP): coalesced per-row reads
    double *Mfrag = nullptr, *Kbfrag = nullptr, *Kufrag = nullptr;
    double *lbs = nullptr, *ubs = nullptr;                  // (NP) scaled bounds, padded rows = 0
    double *rho_dev = nullptr;  // [0] rho, [1] lambda_max est, [2] lambda_min est
    int *flag_dev = nullptr;    // [0] != 0: factorisation failed
    int *mflag = nullptr;       // [n_models]: this model's factorisation failed (its instances report MPCQP_MAX_ITER)
    double rho = 0.0;
    double inv_resid = 0.0;     // max |Hs Hi - I|: decides whether the polish refines by default
    // per-batch workspace (grow only)
    int Bcap = 0;               // capacity in instances (multiple of 64)
    int Blast = 0;              // batch of the last solve (for shift / warm start)
    double *z = nullptr, *w = nullptr;     // [Bcap][NP] row-major, scaled variables
    double *b0ws = nullptr;                // [Bcap][NP]: ADMM offsets b0, fragment order per 16 instances
    int *it_ws = nullptr;                  // [6 Bcap + 96]: iters | need_big | converged | polish queue 1 | queue 2 | counters, buckets (96) | cold-start order
    double *rp_ws = nullptr, *rd_ws = nullptr;
    double *hbuf1 = nullptr, *hbuf2 = nullptr;   // hand-over of the working state between active-set passes: [Bcap][16*16+66], [Bcap/4][32*32+130]
    int hcap = 0;                                // batch capacity they were allocated for (0: not yet -- ensure_handover, api.hip)
    // general-constraint path (general.hip): state-box / polytopic rows, or N*nu > 128
    const double *g_xmin = nullptr, *g_xmax = nullptr, *g_C = nullptr, *g_cmin = nullptr, *g_cmax = nullptr;   // caller-owned
    int g_nc = 0;
    int g_path = 0;                         // 0 auto | 1 dense kernel | 2 constraint-space kernel
    long g_mem_limit = 32L << 30;           // bytes of shared workspace (the m x m dual Hessian) the constraint-space kernel may take
    bool general = false, g_rows = false;   // the last factorisation took the general path / the constraint-space kernel
    int g_m = 0, g_bcap = 0, g_dcap = 0; long g_per = 0;
    int g_wvalid = 0;                            // batch whose final working sets the rows kernel's instance workspace holds (0: none)
    double *gG = nullptr, *gE = nullptr, *gh0 = nullptr;        // (m, n), (m, nx), (m): rows G U <= h0 - E x0, infinite bounds dropped
    double *g_shared = nullptr, *g_inst = nullptr, *g_q = nullptr, *g_hh = nullptr, *g_z = nullptr, *g_obj = nullptr, *g_dws = nullptr;
    int *g_iws = nullptr, *g_st = nullptr, *g_it = nullptr, *g_diws = nullptr;
    // component split (split.hip): the scaled Hessian of the shared model is block diagonal under a permutation, so every
    // instance is sp_C independent box QPs; the cold active-set solve then runs on batch * sp_C sub-instances of sp_ns rows
    int sp_C = 0, sp_ns = 0, sp_ld = 0;     // components (0 = no split), rows per component (uniform, padded, even), row stride of the packed arrays (16 | 32 | 64)
    int sp_enable = 1;                      // mpcqp_set_split: 0 = never split
    double sp_tol = 1e-14;                  // |Hs_ij|, |Hi_ij| (unit-diagonal scaling) at or below this count as no coupling
    int* sp_perm = nullptr;                 // [C][ld]: component row -> row of the whole problem, -1 = padding
    int* sp_rowmap = nullptr;               // [NP]: row of the whole problem -> component << 16 | component row
    double *sp_Hi = nullptr, *sp_KuT = nullptr, *sp_lbs = nullptr, *sp_ubs = nullptr, *sp_d = nullptr;   // packed per component
    double* sp_img = nullptr;               // the same as ONE image in the layout of the active-set kernels' LDS (rows of sp_ld)
    int sp_small = 65536;                    // sub-instances up to which a batch starts in the 16-slot pass (split.hip)
    int sp_bcap = 0, sp_wsC = 0, sp_wsld = 0;   // sub-instance capacity of the workspace below (grow only) and the split it was sized for
    int* sp_iws = nullptr; double* sp_part = nullptr;           // int / double workspace of the split solve (split.hip)
    double *sp_hb1 = nullptr, *sp_hb2 = nullptr; int sp_cap1 = 0, sp_cap2 = 0;   // hand-over records 8 -> 16 and 16 -> 32 slots
    bool assembled = false, factored = false, has_state = false;
    bool admm_last = false;                      // the last solve ran ADMM iterations (their residuals are in rp_ws / rd_ws): mpcqp_adapt_rho
    double* adapt_ws = nullptr;                  // [2]: sum of log(r_prim / r_dual), number of instances in it (mpcqp_adapt_rho)
    int pinned = 0;                              // > 0: a recorded graph holds this handle's device pointers (mpcqp_pin): nothing may be freed or re-grown
    std::string err;
};

#define HIPCHK(h, call)                                                              \
    do {                                                                             \
        hipError_t e_ = (call);                                                      \
        if (e_ != hipSuccess) {                                                      \
            (h)->err = std::string(#call) + ": " + hipGetErrorString(e_);            \
            return MPCQP_EHIP;                                                       \
        }                                                                            \
    } while (0)

// hipFuncAttributeMaxDynamicSharedMemorySize is a per-device property of a kernel: set it once per (device, kernel) and
// whenever a launch needs more than was set, not on every launch of the timed path (api.hip; mutex-guarded)
hipError_t ensure_dyn_lds(int device, const void* kernel, size_t bytes);

// ---- launchers implemented in the kernel translation units ----------------------
int launch_assemble(mpcqp_handle h, hipStream_t st, bool with_gamma, bool gamma_only = false);
int launch_factor(mpcqp_handle h, double rho, hipStream_t st);
int launch_factor_batch(mpcqp_handle h, hipStream_t st, bool probe);
int launch_admm_shared_box(mpcqp_handle h, const double* x0, int B, const mpcqp_solve_opts& o, hipStream_t st);
int launch_polish_box(mpcqp_handle h, const double* x0, int B, double* U, const mpcqp_info* info,
                      const mpcqp_solve_opts& o, hipStream_t st);
int launch_shift(mpcqp_handle h, hipStream_t st);
bool admm_shared_box_supported(int NT, int S);
// blocked_setup.hip: X = W^-1 (W symmetric positive definite, destroyed) spread over the GPU; scratch sizes in doubles
long blocked_inverse_scratch(int n);
int blocked_spd_inverse(double* W, int n, double* Lout, double* Tt, double* blk, double* X, int* flag, int flag_bits,
                        hipStream_t st);
int ensure_handover(mpcqp_handle h);      // api.hip: the whole-instance cascade's hand-over records, on first use
int factor_general(mpcqp_handle h, hipStream_t st);
// split.hip: look for independent components in the factored shared model (synchronises `st`); free its buffers
int split_setup(mpcqp_handle h, hipStream_t st);
void split_free(mpcqp_handle h);
int solve_general(mpcqp_handle h, const double* x0, int B, double* U, const mpcqp_info* info, hipStream_t st, int warm);
void general_free(mpcqp_handle h);
// ltv_models.hip: apply + plant + shift / re-centre of (z, w) in one launch
int launch_bicycle_advance(int B, int N, double dt, double L, const double* plan_in, const double* dU, double* plan_out, double* u0,
                           const double* x, double* x_next, double* z, double* w, int NP, int n, int nu, hipStream_t st);
