/* mpcqp.h -- C ABI of libmpcqp.so: batched condensed-MPC QP assembly and solve on
 * AMD MI355X (gfx950), fp64.
 *
 * What this boundary replaces in the reference (AOE-khkhan/MPC, src/Planner/):
 *   - the NumPy assembly of (P, q, G, h) from the dynamics: prediction-matrix
 *     rollout (LaunchPlannerQPFormulation.py:49-89, Launch3DPlanner.py:61-108),
 *     Q = J'WJ, f = -J'Wc (HelperFuncs.py:6-11; LaunchPlannerQPFormulation.py:151-162),
 *     box rows (LaunchPlannerQPFormulation.py:94-107);
 *   - cvxopt.matrix(..., tc='d') + cvxopt.solvers.qp(P, q, G, h[, A, b])
 *     (call sites CollocationBasedPlanner.py:59, LaunchPlannerQPFormulation.py:173,
 *     Launch3DPlanner.py:201, CollocationBasedPlanner2D.py:359, SE3Planner.py:181,183);
 *   - the read-back of soln['x'] (LaunchPlannerQPFormulation.py:174).
 *
 * Conventions
 *   - every function returns int: 0 = MPCQP_OK, otherwise an MPCQP_E* code; no C++
 *     exception crosses this boundary; mpcqp_last_error(h) gives the text.
 *   - all matrix/vector arguments are DEVICE pointers to fp64, row-major, owned by
 *     the caller (PyTorch); the handle owns only its workspace.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All calls
 *     are asynchronous w.r.t. the host except create/destroy and workspace growth
 *     on the first solve with a larger batch.
 *   - a handle is not thread-safe; distinct handles are independent.
 *
 * Problem solved per instance b (N-step horizon, nx states, nu inputs):
 *   minimise  sum_{k=1}^{N-1} 1/2 (x_k-xr_k)'Q(x_k-xr_k) + 1/2 (x_N-xr_N)'P(x_N-xr_N)
 *           + sum_{k=0}^{N-1} 1/2 (u_k-ur_k)'R(u_k-ur_k)
 *   s.t.      x_{k+1} = A_k x_k + B_k u_k,   x_0 given,   u_min <= u_k <= u_max
 * condensed to   min 1/2 U'HU + (F x0 + f0)'U,  lb <= U <= ub,  U = [u_0;...;u_{N-1}].
 */
#ifndef MPCQP_H
#define MPCQP_H

#ifdef __cplusplus
extern "C" {
#endif

typedef struct mpcqp_handle_s* mpcqp_handle;

enum {
    MPCQP_OK = 0,
    MPCQP_EINVAL = 1,     /* bad shape / rank / null pointer          -> ValueError       */
    MPCQP_ENUMERIC = 2,   /* factorisation failed (H not PD)          -> ArithmeticError  */
    MPCQP_EHIP = 3,       /* HIP runtime error                        -> RuntimeError     */
    MPCQP_ESTATE = 4,     /* call order (solve before factor ...)     -> RuntimeError     */
    MPCQP_ENOTIMPL = 5    /* size outside what this build instantiates -> NotImplementedError */
};

/* per-instance status written to info.status */
enum {
    MPCQP_SOLVED = 0,            /* polished, KKT verified on device              */
    MPCQP_SOLVED_INACCURATE = 1, /* ADMM met eps but polish was off / gave up      */
    MPCQP_MAX_ITER = 2           /* ADMM hit max_iter without meeting eps (no polish) */
};

typedef struct mpcqp_dims {
    int nx;            /* states  */
    int nu;            /* inputs  */
    int N;             /* horizon */
    int n_models;      /* 1 = one model shared by the whole batch (LTI/LTV), B = one per instance */
    int time_varying;  /* 0: A,B are (n_models, nx, nx|nu); 1: (n_models, N, nx, nx|nu)          */
} mpcqp_dims;

typedef struct mpcqp_factor_opts {
    double rho;        /* ADMM penalty in Jacobi-scaled variables; <= 0: pick from the spectrum */
    int    no_sync;    /* 1: do not synchronise the stream to report rho / a failed factorisation to the host
                          (needs rho > 0 or n_models > 1); failures then show up as per-instance status
                          MPCQP_MAX_ITER of the following solves (shared model: every instance)              */
    int    keep_state; /* 1: the kept (z, w) stays valid across this re-factorisation (closed loop, warm start) */
} mpcqp_factor_opts;

typedef struct mpcqp_solve_opts {
    int    max_iter;      /* ADMM iterations before the active-set stage (default 0: the active-set kernels alone are
                             fastest when few bounds bind -- measured table in DESIGN.md section 2 item 5;
                             a few iterations pay off when large active sets must be guessed)                */
    int    min_iter;      /* iterations before the first convergence check        (default 5)   */
    int    check_every;   /* residual / active-set check period                   (default 5)   */
    double eps_abs;       /* OSQP-style absolute tolerance on scaled residuals    (default 1e-6)*/
    double eps_rel;       /* relative tolerance                                   (default 1e-6)*/
    double alpha;         /* over-relaxation in (0,2)                             (default 1.6) */
    int    polish;        /* 1: active-set KKT solve after ADMM (exact minimiser) (default 1)   */
    int    polish_refine; /* iterative-refinement steps with H itself; -1 = one step only if
                             max|Hs Hs^-1 - I| > 1e-9 at factor time              (default -1)  */
    int    polish_max_steps; /* cap on add/drop steps per instance               (default 4n)  */
    int    warm_start;    /* 1: start from the (z, w) kept in the handle          (default 0)   */
    int    stages;        /* bit 0: run the ADMM kernel, bit 1: run polish/finalise (default 3);
                             1 or 2 alone are for per-kernel timing and reuse the kept (z, w); bits 2 / 3 (with 3) stop
                             after the first active-set pass / after the cold-start ordering kernels: timing only, the
                             batch is left partly solved */
    int    keep_state;    /* 1 (default): leave the solution in the handle's (z, w) for a later warm start / mpcqp_shift;
                             0: skip those writes (two thirds of a cold solve's HBM traffic) -- a following warm_start then
                             starts cold and mpcqp_shift reports that there is no state */
} mpcqp_solve_opts;

/* all members are device pointers of length batch, each may be NULL */
typedef struct mpcqp_info {
    int*    iters;        /* ADMM iterations executed                                */
    int*    status;       /* MPCQP_SOLVED ...                                        */
    double* obj;          /* 1/2 U'HU + q'U  (same constant-free objective cvxopt
                             reports as 'primal objective')                          */
    double* prim_res;     /* ||x - z||_inf of the last ADMM iterate (scaled)         */
    double* dual_res;     /* rho ||z - z_prev||_inf of the last ADMM iterate (scaled) */
    int*    polish_steps; /* constraints added + dropped by the polish              */
    double* lam;          /* (batch, N*nu) signed multipliers: >0 upper, <0 lower; NULL to skip */
} mpcqp_info;

/* -- lifetime ------------------------------------------------------------------ */
int  mpcqp_create(mpcqp_handle* h, const mpcqp_dims* dims, int device_id);
int  mpcqp_destroy(mpcqp_handle h);
const char* mpcqp_last_error(mpcqp_handle h);   /* h may be NULL: last create error */
void mpcqp_default_solve_opts(mpcqp_solve_opts* o);

/* -- problem data (pointers are remembered, read by mpcqp_assemble) ---------------- */
/* A: (n_models, [N,] nx, nx), B: (n_models, [N,] nx, nu), contiguous. */
int mpcqp_set_model(mpcqp_handle h, const double* A, const double* B);
/* Q (nx,nx), R (nu,nu), P (nx,nx) terminal; x_ref (N,nx) for x_1..x_N or NULL; u_ref (N,nu) or NULL.
 * Shared by all models. */
int mpcqp_set_cost(mpcqp_handle h, const double* Q, const double* R, const double* P,
                   const double* x_ref, const double* u_ref);
/* u_min, u_max: (nu) applied at every stage (per_stage = 0) or (N, nu) (per_stage = 1). */
int mpcqp_set_constraints(mpcqp_handle h, const double* u_min, const double* u_max, int per_stage);

/* State box x_min <= x_k <= x_max and polytopic rows c_min <= C_k x_k <= c_max on x_1..x_N -- the reference's state rows
 * (rows of the prediction matrix, LaunchPlannerQPFormulation.py:109-118) and friction-pyramid rows
 * (Launch3DPlanner.py:134-156).  x_min, x_max: (N, nx) or both NULL; C: (N, nc, nx) with c_min, c_max (N, nc), or NULL.
 * +-infinity opens a side: rows with an infinite bound are dropped on the device, as HelperFuncs.removeNullConstraints
 * (HelperFuncs.py:41-54) drops rows that constrain nothing.  With any of them set -- or with N*nu > 128 -- mpcqp_factor
 * builds the condensed rows [I; -I; Gamma; -Gamma; C Gamma; -C Gamma] U <= h0 - E x0 and sets up the constraint-space
 * kernel (mpcqp_rows_*) or, when its m x m dual Hessian exceeds the memory limit, the general dense kernel
 * (mpcqp_dense_shared_*); mpcqp_solve then runs that kernel.  With one model per instance (n_models = batch) every
 * instance is a dense QP of its own -- its Hessian, its rows [..Gamma_b..], q_b = F_b x0_b + f0_b -- solved by the kernel
 * of mpcqp_solve_dense; a row is kept when its bound is finite in ANY model.  In this mode: no
 * warm start / shift, mpcqp_factor synchronises, info.lam (if given) holds (batch, m) multipliers of the kept rows
 * (m from mpcqp_general_rows; mpcqp_get 13 / 14 / 15 return G (m,n), E (m,nx), h0 (m)), status 3 = infeasible,
 * 4 = numerical failure. */
int mpcqp_set_state_constraints(mpcqp_handle h, const double* x_min, const double* x_max, const double* C,
                                const double* c_min, const double* c_max, int nc);
/* path: 0 = constraint-space kernel when rows beyond the input box exist and its workspace fits (default), 1 = always
 * the general dense kernel, 2 = constraint-space kernel even for a plain input box; mem_limit_bytes: workspace the
 * constraint-space kernel may take (0 keeps the current value; default 32 GiB). */
int mpcqp_set_general_opts(mpcqp_handle h, int path, long mem_limit_bytes);
/* rows kept by the last mpcqp_factor in the general mode (0: the box kernels are in use) */
int mpcqp_general_rows(mpcqp_handle h, int* m, int* uses_rows_kernel);

/* One model per instance (dims.n_models = batch > 1; e.g. an LTV re-linearisation about each instance's own
 * trajectory): x_ref, u_ref and the bounds may then differ per model too.  Strides are in doubles between
 * consecutive models, 0 = shared by all (the default).  In this regime there is no shared factor, so
 * mpcqp_solve skips the ADMM GEMM kernel and the dual active-set kernel does the whole solve (cold, or from the
 * kept state with warm_start = 1); batch must equal n_models. */
int mpcqp_set_strides(mpcqp_handle h, long xref_stride, long uref_stride, long bound_stride);

/* Component split.  When the Jacobi-scaled Hessian of a SHARED model (n_models = 1, input box only, N*nu <= 128) is block
 * diagonal under a permutation of the rows -- models built from independent axes, as the reference's
 * Launch3DPlanner.py:61-108 builds x, y, z -- every instance is a set of independent box QPs, and a cold mpcqp_solve runs
 * the dual active-set kernels per block (same minimiser, same outputs; fewer flops per step, shorter dependent chains).
 * mpcqp_factor looks for the blocks on its synchronous set-up call: entries of Hs and of Hs^-1 at or below `tol` relative
 * to the unit diagonal count as no coupling (default 1e-14: rounding noise of the assembly; 0 keeps the current value).
 * enable = 0 switches the search off.  Takes effect with the next mpcqp_factor.  Warm-started solves, ADMM iterations
 * first, one model per instance and the general-constraint mode do not split. */
int mpcqp_set_split(mpcqp_handle h, int enable, double tol);
/* what the last mpcqp_factor found: components (0 = not split), rows per component (the largest, padded to even);
 * row_map (host, N*nu ints, may be NULL): component of every row */
int mpcqp_split_info(mpcqp_handle h, int* components, int* rows_per_component, int* row_map, int row_map_len);

/* Recording mpcqp_solve into a HIP graph (hipStreamBeginCapture on the solve's stream).  The launch chain of a solve
 * carries no host state -- every solve leaves its device-side counters zeroed for the next, nothing is read back -- so it
 * can be captured and replayed, PROVIDED the device pointers the recorded kernels hold stay valid: run one eager solve of
 * the same batch size and options first (workspaces grow, and are zero-filled with synchronous calls, on the first solve of
 * a larger batch and on the first solve that takes a given kernel cascade), then pin the handle for as long as the graph
 * lives.  A pinned handle (on = 1; calls nest, on = 0 undoes one) answers MPCQP_ESTATE instead of freeing or re-growing
 * anything a graph may point into: a solve of a larger batch, a solve that needs buffers not yet allocated, mpcqp_assemble
 * and mpcqp_factor of a shared model (new matrices under recorded kernels; the split images are re-packed).  x0, U and the info arrays are the caller's: the graph reads and
 * writes the addresses given at capture.  (Replaces nothing in the reference, whose solves are host calls.) */
int mpcqp_pin(mpcqp_handle h, int on);

/* -- kernels ----------------------------------------------------------------------- */
/* rollout Gamma/Phi, H = Gamma'Qbar Gamma + Rbar, F = Gamma'Qbar Phi, f0  (fp64 MFMA) */
int mpcqp_assemble(mpcqp_handle h, void* stream);
/* Jacobi scaling, Cholesky of Hs + rho I and of Hs, explicit inverses, MFMA fragment images */
int mpcqp_factor(mpcqp_handle h, const mpcqp_factor_opts* opts, void* stream);
/* x0 (batch, nx) -> U (batch, N*nu).  ADMM (+ polish). */
int mpcqp_solve(mpcqp_handle h, const double* x0, int batch, double* U,
                const mpcqp_info* info, const mpcqp_solve_opts* opts, void* stream);
/* rho update of the ADMM stage (BASELINE.json north_star, subsystem 2: "rho updates"; the reference has no counterpart -- cvxopt's
 * interior point has no penalty parameter).  One (Hs + rho I)^-1 serves the whole batch, so rho changes where the factorisation
 * does: between solves.  After a solve that ran ADMM iterations (max_iter > 0) this call forms OSQP's ratio from the
 * batch's final residuals -- rho_new = rho sqrt(r_prim / r_dual), geometric mean over the instances with both residuals
 * non-zero -- and, when rho_new is more than `threshold` (> 1; <= 0: OSQP's 5) away from rho either way, factors again with
 * it (synchronous, like a set-up mpcqp_factor) and rescales the kept scaled duals w = y / rho, so that a warm start stays
 * valid.  *rho_out = the penalty in force afterwards, *refactored = 1 if it changed.  MPCQP_ESTATE without such a solve;
 * shared models with box bounds only (the ADMM kernel's regime). */
int mpcqp_adapt_rho(mpcqp_handle h, double threshold, double* rho_out, int* refactored, void* stream);
/* receding horizon: shift the kept (z, w) one stage forward, repeating the last stage */
int mpcqp_shift(mpcqp_handle h, void* stream);

/* trajectory recovery (`x_traj = Gamma u* + ...`, LaunchPlannerQPFormulation.py:176-183, Launch3DPlanner.py:202-211):
 * X (batch, N, nx) = x_1..x_N of x_{k+1} = A_k x_k + B_k u_k from x0 (batch, nx) and U (batch, N*nu), each instance
 * with the model the handle holds for it */
int mpcqp_predict(mpcqp_handle h, const double* x0, const double* U, int batch, double* X, void* stream);

/* deviation-form re-linearisation (the reference's `D = D + optX` / `f += delf` pattern, CollocationBasedPlanner2D.py:363,
 * SE3Planner.py:194-195): when the next QP is posed in increments about the shifted previous plan, the kept primal
 * state is zeroed (the plan itself is the new origin) while the kept multipliers still mark the active bounds */
int mpcqp_recenter(mpcqp_handle h, void* stream);
/* mpcqp_shift followed by mpcqp_recenter as one launch (what a re-linearising closed loop does every step) */
int mpcqp_shift_recenter(mpcqp_handle h, void* stream);

/* -- introspection (device -> caller's device buffers; for tests and parity reports) --- */
/* which: 0 H (n,n)  1 F (n,nx)  2 f0 (n)  3 Gamma (N nx, n)  4 Phi (N nx, nx)
 *        5 Hs (n,n) 6 M=(Hs+rho I)^-1 (n,n)  7 Hi=Hs^-1 (n,n)  8 d (n)  9 rho (1)
 *        10 z (batch,n) scaled  11 w (batch,n) scaled
 *        12 diagnostics of the last solve (96 values): [0] instances handed to the second active-set pass (routed by the
 *           cold-start ordering + overflow of the first), [1] handed on to the 64-row pass, [2] instances the first pass
 *           took, [3] handed to the any-size pass, [16..48) histogram of the cold-start keys
 *        13 G (m,n)  14 E (m,nx)  15 h0 (m): the condensed rows of the general-constraint mode
 * Gamma: with one model per instance and nx <= 8 the assembly does not form the prediction matrices (H, F, f0 come from the
 * backward recursion of the dynamics); selector 3 then runs the prediction-matrix assembly first, on `stream`.
 * Diagnostics [12]: a cold batch of 8,192 instances or fewer is one 32-slot pass, so [0] and [2] stay 0 for it. */
int mpcqp_get(mpcqp_handle h, int which, int model, double* out, long out_len, void* stream);

/* -- dense QP: the reference's literal call  cvxopt.solvers.qp(P, q, G, h[, A, b]) ---------------------
 * (CollocationBasedPlanner.py:59, LaunchPlannerQPFormulation.py:173, Launch3DPlanner.py:201,
 *  CollocationBasedPlanner2D.py:359, SE3Planner.py:181,183) for `batch` independent QPs of equal shape:
 *      min 1/2 x'Px + q'x   s.t.  Gx <= h (m rows),  Ax = b (p rows),   P symmetric PSD.
 * All pointers are device pointers, row-major, instance-major ([batch][n*n], [batch][m*n] ...).
 * Outputs: x [batch][n]; z [batch][m] (>= 0, may be NULL); y [batch][p] (may be NULL); obj [batch];
 * status [batch]: 0 optimal, 1 step cap reached, 2 infeasible, 3 numerical failure (P not PSD);
 * iters [batch]: active-set steps + 1000 * proximal iterations.
 * The caller provides the scratch: `workspace` of batch * mpcqp_dense_workspace(n,m,p,&ni) doubles and
 * `int_workspace` of batch * ni ints.  No handle, no allocation, asynchronous on `stream`. */
long mpcqp_dense_workspace(int n, int m, int p, long* int_count);
int  mpcqp_solve_dense(int device_id, int batch, int n, int m, int p, const double* P, const double* q,
                       const double* G, const double* h, const double* A, const double* b,
                       double* x, double* z, double* y, double* obj, int* status, int* iters,
                       double* workspace, int* int_workspace, void* stream);

/* the same with the solver's knobs exposed (what cvxopt.solvers.options carries for the reference:
 * CollocationBasedPlanner2D.py:16-20 sets maxiters and refinement): */
typedef struct mpcqp_dense_opts {
    int    max_steps;   /* cap on active-set steps per QP, status 1 when reached; <= 0: 20 (n+m+p) + 100              */
    int    prox_iters;  /* cap on proximal-point outer iterations (only singular P needs more than one) (default 30) */
    int    refine;      /* refinement rounds on the final active set with P itself                       (default 2)  */
    double sigma_rel;   /* proximal weight relative to max diag of the scaled P                          (default 1e-6) */
} mpcqp_dense_opts;
void mpcqp_default_dense_opts(mpcqp_dense_opts* o);
/* strides as in mpcqp_solve_dense_strided below; opts may be NULL (defaults) */
int  mpcqp_solve_dense_opts(int device_id, int batch, int n, int m, int p, const double* P, long sP, const double* q,
                            long sq, const double* G, long sG, const double* h, long sh, const double* A, long sA,
                            const double* b, long sb, double* x, double* z, double* y, double* obj, int* status,
                            int* iters, double* workspace, int* int_workspace, const mpcqp_dense_opts* opts, void* stream);

/* mpcqp_solve_dense for callers that cannot allocate.  NO COUNTERPART IN THE REFERENCE: it exists for callers of this
 * boundary that may not call hipMalloc (code under HIP graph capture, tight serving loops, harnesses that forbid
 * allocation) and therefore cannot bring the workspace mpcqp_solve_dense asks for.  The workspace is a static device
 * array inside the library (MPCQP_DENSE_STATIC_DOUBLES doubles + MPCQP_DENSE_STATIC_INTS ints, about 16.3 MB, held by
 * every GPU the library runs on until the process ends); the call allocates nothing and frees nothing.
 * Limit: batch * mpcqp_dense_workspace(n,m,p,&ni) <= MPCQP_DENSE_STATIC_DOUBLES and batch * ni <= MPCQP_DENSE_STATIC_INTS,
 * otherwise MPCQP_ENOTIMPL and nothing is launched (e.g. 3 QPs of CollocationBasedPlanner2D.py:359's size n=280, m=236,
 * p=188, or 860 of SE3Planner.py:183's n=18, m=12).  All calls on one GPU share that GPU's array: calls that overlap on
 * one GPU (two streams, two threads) race; calls on one stream are ordered and safe.  Same arguments, outputs and
 * status codes as mpcqp_solve_dense otherwise. */
#define MPCQP_DENSE_STATIC_DOUBLES (2L << 20)
#define MPCQP_DENSE_STATIC_INTS    (64L << 10)
int  mpcqp_solve_dense_static(int device_id, int batch, int n, int m, int p, const double* P, const double* q,
                              const double* G, const double* h, const double* A, const double* b,
                              double* x, double* z, double* y, double* obj, int* status, int* iters, void* stream);

/* the same with explicit strides (doubles between consecutive QPs; 0 = one array shared by the whole batch): the MPC
 * case with state / polytopic rows has ONE Hessian and ONE constraint matrix [I; Gamma rows; C Gamma rows]
 * (LaunchPlannerQPFormulation.py:94-121, Launch3DPlanner.py:115-160) and per-instance q = F x0 + f0, h = h0 - E x0 */
int  mpcqp_solve_dense_strided(int device_id, int batch, int n, int m, int p, const double* P, long sP, const double* q,
                               long sq, const double* G, long sG, const double* h, long sh, const double* A, long sA,
                               const double* b, long sb, double* x, double* z, double* y, double* obj, int* status,
                               int* iters, double* workspace, int* int_workspace, void* stream);

/* the shared case split like factor / solve: set-up once per model (scaling, proximal Hessian, its inverse, scaled rows --
 * into the workspace of QP 0), then any number of solves with new (q, h, b) on the SAME workspace
 * (batch * mpcqp_dense_workspace doubles; the batch of the solves may not exceed what the workspace was sized for) */
int  mpcqp_dense_shared_setup(int device_id, int n, int m, int p, const double* P, const double* G, const double* A,
                              double* workspace, int* int_workspace, void* stream);
int  mpcqp_dense_shared_solve(int device_id, int batch, int n, int m, int p, const double* q, const double* h,
                              const double* b, double* x, double* z, double* y, double* obj, int* status, int* iters,
                              double* workspace, int* int_workspace, void* stream);

/* the same shared case when MANY rows meet few active ones (state / polytopic rows over a long horizon: BASELINE.json
 * configs[4], 13,200 rows x 600 variables; rows as in LaunchPlannerQPFormulation.py:109-114, Launch3DPlanner.py:134-142):
 * the set-up additionally forms CH = Cs Hi (m x n) and the dual Hessian Cs Hi Cs' (m x m), and each solve keeps the row
 * residuals up to date through rows of that matrix instead of re-evaluating G x.  Needs P positive definite (the
 * condensed MPC Hessian is: Rbar > 0) and inequality rows only; at most 176 rows may be active at once (status 3
 * beyond; instances with more than 96 finish in a second launch).  Same outputs and status codes as mpcqp_solve_dense
 * (`status` is required here); iters = active-set steps.
 * shared_ws: mpcqp_rows_workspace(n, m, &per) doubles, written by the set-up, read-only afterwards;
 * int_ws: n ints (set-up only); instance_ws: batch * per doubles. */
long mpcqp_rows_workspace(int n, int m, long* per_instance);
int  mpcqp_rows_setup(int device_id, int n, int m, const double* P, const double* G, double* shared_ws, int* int_ws,
                      void* stream);
int  mpcqp_rows_solve(int device_id, int batch, int n, int m, const double* q, const double* h, double* x, double* z,
                      double* obj, int* status, int* iters, const double* shared_ws, double* instance_ws, void* stream);
/* the same with the MPC data maps applied on the device: q = F x0 + f0 (F (n,nx), f0 (n)), h = h0 - E x0 (E (m,nx), h0 (m)),
 * x0 (batch,nx) -- what LaunchPlannerQPFormulation.py:94-121,151-162 evaluate on the host for its one initial state */
int  mpcqp_rows_solve_mpc(int device_id, int batch, int n, int m, int nx, const double* x0, const double* F,
                          const double* f0, const double* E, const double* h0, double* x, double* z, double* obj,
                          int* status, int* iters, const double* shared_ws, double* instance_ws, void* stream);
/* the same, started from the working set each instance's PREVIOUS solve in this instance_ws ended with (every solve leaves
 * it there): for loops that solve the same model again from nearby data -- the reference's sequential QPs,
 * CollocationBasedPlanner2D.py:85-365, SE3Planner.py:108-195.  Members of the kept set whose multiplier comes out with the
 * wrong sign for the new data are released first; the minimiser found is the same as a cold solve's.  Only valid when the
 * last call on this instance_ws was a solve of the same batch on the same shared_ws (mpcqp_solve with warm_start = 1 in
 * the general-constraint mode checks that and falls back to a cold solve otherwise). */
int  mpcqp_rows_solve_mpc_warm(int device_id, int batch, int n, int m, int nx, const double* x0, const double* F,
                               const double* f0, const double* E, const double* h0, double* x, double* z, double* obj,
                               int* status, int* iters, const double* shared_ws, double* instance_ws, void* stream);

/* -- pre-solve row hygiene: removeNullConstraints(J, c, threshold) of HelperFuncs.py:41-54 (called at
 * CollocationBasedPlanner2D.py:338 on the stacked equality rows before they reach solvers.qp) -----------------------
 * Rows of [J | c] whose coefficients J[i,:] are ALL below `threshold` in magnitude are dropped; the others keep their
 * order.  J (rows, cols), c (rows): device, row-major.  J_out (rows, cols), c_out (rows): device, capacity of the input,
 * not aliasing it; the first counts[0] rows are valid afterwards.  counts: device int[2] = {rows kept, rows dropped
 * although |c| > threshold (the reference prints a warning for each)}.  int_ws: 2 * rows ints.  Asynchronous: read
 * `counts` after synchronising the stream. */
int mpcqp_remove_null_rows(int device_id, int rows, int cols, const double* J, const double* c, double threshold,
                           double* J_out, double* c_out, int* counts, int* int_ws, void* stream);

/* -- scenario model for the closed-loop LTV configuration (BASELINE.json configs[3]; the reference has no counterpart:
 * its scripts keep the model code inline, e.g. CollocationBasedPlanner2D.py:88-347) --------------------------------
 * Kinematic bicycle, state (x, y, psi, v), input (a, delta), explicit Euler.  For every rollout: nominal trajectory
 * xbar (batch, N+1, 4) of `plan` (batch, N, 2) from x (batch, 4); Jacobians A (batch, N, 4, 4), B (batch, N, 4, 2);
 * and the data of the QP in input increments: xref_dev = x_ref - xbar[1:], uref_dev = -plan, lb/ub = u_min/u_max - plan
 * (all (batch, N, .)).  Feed them to mpcqp_set_model / set_cost / set_constraints / set_strides with n_models = batch. */
int mpcqp_ltv_bicycle(int device_id, int batch, int N, double dt, double L, const double* x, const double* plan,
                      const double* x_ref, const double* u_min, const double* u_max, double* xbar, double* A,
                      double* B, double* xref_dev, double* uref_dev, double* lb, double* ub, void* stream);
/* after the solve of that QP: plan = plan_in + dU; u0 (batch, 2) = its first stage (the applied input); plan_out = the plan
 * moved up one stage with the last one repeated (may not alias plan_in); with x, x_next (batch, 4; both or neither) the
 * bicycle itself one step on under u0.  One launch for the apply / shift of a receding-horizon step. */
int mpcqp_ltv_bicycle_apply(int device_id, int batch, int N, double dt, double L, const double* plan_in, const double* dU,
                            double* plan_out, double* u0, const double* x, double* x_next, void* stream);
/* mpcqp_ltv_bicycle_apply for the batch of the handle's last solve AND mpcqp_shift_recenter of its kept state, as one launch:
 * everything a re-linearising closed loop does between two solves (same arguments; batch and N are the handle's) */
int mpcqp_ltv_bicycle_advance(mpcqp_handle h, double dt, double L, const double* plan_in, const double* dU, double* plan_out,
                              double* u0, const double* x, double* x_next, void* stream);
/* x_next = f(x, u): the plant of the closed loop (x, x_next (batch, 4); u (batch, 2)) */
int mpcqp_ltv_bicycle_step(int device_id, int batch, double dt, double L, const double* x, const double* u, double* x_next,
                           void* stream);

/* -- multi-GPU: there are NO mpcqp_comm_* entry points, on purpose -------------------------------------------------
 * SURVEY.md section 8(b2) sketched mpcqp_comm_init_all / mpcqp_allgather_solutions / mpcqp_allreduce_flag over a
 * single-process ncclCommInitAll.  They are not part of this library: MPC instances are independent, so a rank never
 * needs another rank's data to solve its shard, and the deployment model of this boundary is ONE PROCESS PER GPU
 * (each with its own handle, created with its device_id) -- the process launcher and the two optional collectives
 * (all-gather of U, all-reduce of the worst status) belong to the host framework that already owns the ranks:
 * torch.distributed / RCCL in mpc_amd/distributed.py and bench.py, MPI or plain RCCL for a C or C++ caller (the
 * buffers are ordinary device pointers: ncclAllGather(U_shard, U_all, rows * N*nu, ncclDouble, comm, stream) and
 * ncclAllReduce(worst_status, ..., 1, ncclInt32, ncclMax, comm, stream) on the stream given to mpcqp_solve).  A second,
 * single-process communicator inside the library would duplicate that and tie the handle to one process owning all
 * devices.  The reference has no counterpart either way (SE3Planner.py:108-195 couples QPs only in time). */

#ifdef __cplusplus
}
#endif
#endif /* MPCQP_H */
