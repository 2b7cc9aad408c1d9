/*
 * include/lmpc_hip.h -- C ABI of liblmpc_hip.so (MI355X / gfx950).
 *
 * Drop-in boundary for the per-time-step LMPC hot path of urosolia/RacingLMPC.  The reference has
 * no FFI: its seam is Python duck typing (Simulator.sim -> Controller.solve(x0), SysModel.py:34).
 * Each entry point below replaces the reference interface cited next to it (paths under
 * /root/reference/src/fnc/controller/).  The Python side that binds these symbols with ctypes is
 * racinglmpc_amd/_capi.py; the drop-in modules are racinglmpc_amd/PredictiveControllers.py and
 * racinglmpc_amd/PredictiveModel.py.
 *
 * Conventions: extern "C"; every function returns 0 on success or a negative LMPC_E_* code and
 * never throws; plain pointers and sizes only; all matrices row-major FP64 (NumPy C-contiguous);
 * host pointers unless the name ends in _dev; caller owns every buffer; one lmpc_ctx per Python
 * controller object; calls on one ctx are serialised by the caller.
 */
#ifndef LMPC_HIP_H
#define LMPC_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define LMPC_NX 6
#define LMPC_NU 2
#define LMPC_MAX_TRACK_ROWS 16
#define LMPC_MAX_USED_LAPS 32      /* trToUse / numSS_it upper bound (the reference takes any: PredictiveControllers.py:293-311, PredictiveModel.py:31) */
#define LMPC_MAX_N 64
#define LMPC_MAX_SS_POINTS 384     /* numSS_points upper bound = 32 laps x 12 points (beyond 58 the terminal block takes several columns per lane) */

/* error codes (function return values) */
#define LMPC_OK 0
#define LMPC_E_ARG (-1)            /* bad argument / unsupported configuration */
#define LMPC_E_HIP (-2)            /* a HIP runtime call failed (see lmpc_last_error) */
#define LMPC_E_CAPACITY (-3)       /* (no longer returned: the lap stores grow on demand; kept so that the numbering of the codes is stable) */
#define LMPC_E_STATE (-4)          /* call not valid in the current state (e.g. no laps stored) */
#define LMPC_E_VARIANT (-5)        /* lmpc_create: the solve kernels for this (N, numSS_points) are not part of the library and their shared object
                                      liblmpc_var_N<N>_S<S>.so (next to liblmpc_hip.so) has not been built yet; racinglmpc_amd builds it on demand */

/* per-problem status bits (status[] outputs); 0 == solved */
#define LMPC_ST_MAXITER 1          /* interior-point iteration limit hit before tolerances met */
#define LMPC_ST_REG_SINGULAR 2     /* local regression normal matrix not positive definite (reference: cvxopt raises) */
#define LMPC_ST_NO_SEGMENT 4       /* curvature(): s on no track segment (reference: int(np.where) raises, Track.py:307) */
#define LMPC_ST_WINDOW 8           /* safe-set window runs past the end of a stored lap (reference: IndexError, :497) */
#define LMPC_ST_NUMERIC 16         /* NaN / non-positive pivot inside the KKT factorisation */
#define LMPC_ST_NOT_INTERIOR 32    /* u = 0 is not strictly inside Fu u <= bu (solver start point) */
#define LMPC_ST_INFEASIBLE 128      /* lmpc_config.slacks = 0 (MPCParams.slacks = False): a hard lane row is exceeded by more than 1e-8 at the optimum of the penalised
                                    * problem, i.e. the hard problem has no feasible point (e.g. x0 already outside the lane: its k = 0 row is fixed by the
                                    * equality).  Reference: OSQP reports primal infeasible and MPC.solve sets feasible = 0 (PredictiveControllers.py:277-280). */
#define LMPC_ST_INEXACT 64         /* returned iterate is optimal to working accuracy only (gap < 1e-9, dual residual < 1e-5 rel., equality
                                      residual < 1e-7) because the factorisation broke down or the iteration limit was hit at that point;
                                      the solution is usable (the reference's own solver tolerance is 1e-3) */

typedef struct lmpc_ctx lmpc_ctx;

/* Numeric content of MPCParams (PredictiveControllers.py:24-51), of PredictiveModel.__init__
 * (PredictiveModel.py:12-32), of the LMPC constructor (PredictiveControllers.py:293-311) and of the
 * track table Map.PointAndTangent (Track.py:54-133).                                              */
typedef struct {
    int N;                      /* horizon (reference main.py:43 uses 14; BASELINE metric uses 12) */
    int numSS_it;               /* laps in the safe set per solve; 0 => plain MPC/LTV-MPC, no terminal set */
    int numSS_points;           /* total safe-set columns (reference: 12 * numSS_it = 48); any multiple of numSS_it up to LMPC_MAX_SS_POINTS, at most 63 per lap */
    int trToUse;                /* laps used by the regression (PredictiveModel usedIt) */
    int maxNumPoint;            /* 7   (PredictiveModel.py:18) */
    double h, lamb, dt;         /* 5, 0.0, 0.1 (PredictiveModel.py:19-21) */
    double scaling[5];          /* diag of PredictiveModel.py:22-26 */
    double Q[36], R[4], Qf[36], dR[2], Qslack[2], QtermSlack[36], xRef[6];
    double Fx[12], bx[2], Fu[8], bu[4];          /* 2 state rows, 4 input rows (reference shapes) */
    double track[LMPC_MAX_TRACK_ROWS * 6]; int track_rows; double trackLength;
    int device;                 /* HIP device ordinal */
    int max_batch;              /* capacity of internal work buffers */
    int max_laps, max_lap_len;  /* INITIAL lap-store capacity (rows per lap include addPoint extensions).  The reference's stores are Python lists
                                 * (PredictiveControllers.py:418-445, :466-474; PredictiveModel.py:35-46): when a lap or a row does not fit, the
                                 * device stores are reallocated at >= twice the size (lmpc_*_add_trajectory, lmpc_ss_add_point, lmpc_ss_extend_lap) */
    /* solver: structure-exploiting primal-dual interior point on the block-banded KKT system */
    double tol_gap, tol_res, reg_lambda; int max_iter;
    int slacks;                 /* MPCParams.slacks (:184-198, 218-221, 248-254): 1 = lane rows softened by slack variables (main.py always);
                                   0 = hard lane rows Fx x_k <= bx, no slack variables in the QP (plain MPC only: numSS_it = 0).  Solved with an internal
                                   quadratic slack weight of 1e12: a hard row may be exceeded by mu / 2e12 ~ 1e-11; the `slack` output holds that excess. */
} lmpc_config;

typedef struct {
    double ms_regress, ms_solve;      /* accumulated HIP-event time of the two kernels (profiling on) */
    long long n_regress, n_solve;     /* launches accumulated */
    long long qp_solved;              /* problems passed through the solve kernel */
    long long ipm_iters;              /* interior-point iterations accumulated by the host-buffer entry points (lmpc_step_batch, lmpc_qp_solve_batch) */
    long long n_regress_timed, n_solve_timed;   /* launches that carried events: ms_regress / n_regress_timed is the average kernel duration */
    long long n_retry;                /* retry passes launched on demand (a problem of the batch ended at the iteration limit / broke down; see lmpc_capi.hip) */
} lmpc_stats;

int lmpc_config_default(lmpc_config *cfg);                       /* reference defaults, N = 12 */
int lmpc_create(const lmpc_config *cfg, lmpc_ctx **out);         /* MPC/LMPC.__init__, :63-107, :293-338 */
int lmpc_create_ex(const lmpc_config *cfg, unsigned flags, lmpc_ctx **out);
        /* lmpc_create with flags.  LMPC_CREATE_RUNTIME_KERNEL: an (N, numSS_points) pair that is neither built in nor available as a variant library is served by
         * the runtime-(N, S) solve kernel (any horizon without a compiler on the box, several times slower) instead of LMPC_E_VARIANT -- MPCParams.N is a plain
         * parameter in the reference (:63-107, main.py:43).  LMPC_CREATE_FORCE_RUNTIME_KERNEL: use that kernel even where a fast one exists (tests) */
#define LMPC_CREATE_RUNTIME_KERNEL 1u
#define LMPC_CREATE_FORCE_RUNTIME_KERNEL 2u
int lmpc_solver_kind(lmpc_ctx *);                                /* 0: built-in solve kernels, 1: variant library, 2: runtime-(N, S) kernel */
int lmpc_destroy(lmpc_ctx *ctx);
const char *lmpc_last_error(void);
const char *lmpc_active_knobs(void);                             /* developer environment variables this process has acted on ("NAME=value;..."; "" = none): LMPC_K1_QG,
                                                                    LMPC_K1_RPL16, LMPC_MW_MAX_BATCH, LMPC_MW2_MAX_BATCH, LMPC_FUSE, LMPC_CD, LMPC_NO_ABG -- route / grid choices
                                                                    with identical results, each announced once on stderr; no reference counterpart */
int lmpc_version(void);
int lmpc_device_memory(int device, unsigned long long *free_bytes, unsigned long long *total_bytes);
        /* hipMemGetInfo of `device`: what is left of the 288 GB for max_batch, the lap stores (lmpc_config: max_laps x max_points) and rollout sessions -- and what a
         * destroyed context has given back (tests/test_gpu_stores.py).  No reference counterpart (the reference holds its stores in Python lists) */

/* ---- lap stores -------------------------------------------------------------------------------
 * Two separate stores, as in the reference: the regression store (PredictiveModel.xStored/uStored)
 * and the safe set (LMPC.SS/uSS/Qfun).  Device layout: [lap][column][row], FP64, row stride
 * max_lap_len, so that a wave scanning one column of one lap reads contiguous memory.           */
int lmpc_model_add_trajectory(lmpc_ctx *, const double *x /*T x 6*/, const double *u /*T x 2*/, int T);
        /* PredictiveModel.addTrajectory, PredictiveModel.py:35-46: sorted insert, ascending T, ties append */
int lmpc_model_num_laps(lmpc_ctx *, int *n);
int lmpc_model_replace_lap(lmpc_ctx *, int pos /*position in sorted order*/, const double *x, const double *u, int T);
int lmpc_ss_add_trajectory(lmpc_ctx *, const double *x, const double *u, int T);
        /* LMPC.addTrajectory + computeCost, PredictiveControllers.py:418-464 (it += 1) */
int lmpc_ss_add_point(lmpc_ctx *, const double *x /*6*/, const double *u /*2*/);
        /* LMPC.addPoint, :466-474: append x + [0,0,0,0,TrackLength,0] to lap it-1, Qfun[-1]-1 */
int lmpc_ss_replace_lap(lmpc_ctx *, int lap, const double *x, const double *u, const double *qfun, int T);
int lmpc_ss_set_selected(lmpc_ctx *, const int *laps, int n);
        /* override of argsort(LapTime)[0:numSS_it] (:395,402); n = 0 restores the built-in stable sort */
int lmpc_ss_num_laps(lmpc_ctx *, int *n);
int lmpc_ss_get_qfun(lmpc_ctx *, int lap, double *qfun /*T*/, int *T);
int lmpc_store_read_lap(lmpc_ctx *, int store /*0: regression store (sorted position), 1: safe set*/, int lap, double *x /*T x 6*/, double *u /*T x 2*/, double *qfun /*T, safe set only*/, int *T);
        /* checkpoint / resume: one stored lap back on the host, safe-set laps with the rows LMPC.addPoint appended (the reference imports pickle for this and never uses it,
         * main.py:36; racinglmpc_amd._capi.Context.save_stores / restore_stores write and read an .npz).  Any of x, u, qfun may be NULL */
int lmpc_ss_get_laptime(lmpc_ctx *, int lap, int *T);
        /* LMPC.LapTime[lap] (:420): rows of the lap when it was added (without addPoint extensions) -- what argsort(LapTime) sorts */

/* ---- batched compute, host buffers --------------------------------------------------------- */
int lmpc_regress_batch(lmpc_ctx *, int B, const double *xLin /*B x N x 6 (first N rows used)*/, int xLinRowStride /* (N+1)*6 or N*6 */,
                       const double *uLin /*B x N x 2*/,
                       double *A /*B x N x 36*/, double *Bm /*B x N x 12*/, double *C /*B x N x 6*/, int *status /*B x N*/);
int lmpc_regress_points(lmpc_ctx *, int n, const double *x /*n x 6*/, const double *u /*n x 2*/, double *A /*n x 6 x 6*/, double *B /*n x 6 x 2*/, double *C /*n x 6*/, int *status /*n*/);
        /* PredictiveModel.regressionAndLinearization, PredictiveModel.py:48-197, for n independent points (the reference calls it with one): no horizon around them */
        /* MPC.computeLTVdynamics -> PredictiveModel.regressionAndLinearization, :140-145 / PredictiveModel.py:48-197 */

int lmpc_select_batch(lmpc_ctx *, int B, const double *x0 /*B x 6*/, const double *zt /*B x 6*/,
                      const double *xPredPrev /*B x (N+1) x 6*/, const int *hasPred /*B*/, const int *timeStep /*B*/,
                      double *ssSel /*B x S x 6*/, double *qSel /*B x S*/, double *succ /*B x S x 6*/, double *succU /*B x S x 2*/,
                      double *ztUsed /*B x 6*/, int *selStart /*B x numSS_it: first row of each lap's window, or NULL*/, int *status /*B*/);
        /* LMPC.addTerminalComponents (selection part) + selectPoints, :386-412, :478-514.  xPredPrev, hasPred, timeStep may be NULL (a first step);
         * a set hasPred[b] without xPredPrev is LMPC_E_ARG -- here, in lmpc_step_batch and (any hasPred array) in lmpc_step_batch_dev */

int lmpc_qp_solve_batch(lmpc_ctx *, int B, const double *A, const double *Bm, const double *C,
                        const double *x0 /*B x 6*/, const double *uOld /*B x 2*/,
                        const double *ssSel /*B x S x 6 or NULL*/, const double *qSel /*B x S or NULL*/,
                        double *xPred /*B x (N+1) x 6*/, double *uPred /*B x N x 2*/, double *slack /*B x 2N*/,
                        double *lambda /*B x S*/, double *sTerm /*B x 6*/, double *mu /*B x (8N+S) ineq duals, reference row order*/,
                        int *status /*B*/, int *iters /*B*/, double *resid /*B x 3: gap, dual res, eq res*/);
        /* buildCost/buildEqConstr/addSafeSet* + osqp_solve_qp + unpackSolution, :200-283, :340-379 */

int lmpc_step_batch(lmpc_ctx *, int B, const double *x0, const double *xLin /*B x (N+1) x 6*/, const double *uLin /*B x N x 2*/,
                    const double *uOld, const double *zt, const double *xPredPrev, const int *hasPred, const int *timeStep,
                    double *xPred, double *uPred, double *slack, double *lambda, double *sTerm,
                    double *ztNext /*B x 6*/, double *ztuNext /*B x 2*/, double *ssSel /*B x S x 6*/,
                    double *qSel /*B x S: Qfun_SelectedTot, :404-412, or NULL*/, double *mu /*B x (8N+S) inequality duals, reference row order, or NULL*/,
                    double *Aout /*B x N x 36 or NULL*/, double *Bout, double *Cout,
                    int *status, int *iters, double *resid);
        /* one full MPC.solve(x0) per problem, :110-137 (a3 -> a19 of SURVEY section 8).  status[b] carries the solver bits and the
         * OR of the regression kernel's per-point bits (LMPC_ST_REG_SINGULAR, LMPC_ST_NO_SEGMENT) of the problem's N points. */

int lmpc_assemble_batch(lmpc_ctx *, int B, const double *A, const double *Bm, const double *C, const double *x0, const double *uOld,
                        const double *ssSel, const double *qSel,
                        double *Pdense /*B x nz x nz*/, double *q /*B x nz*/, double *Adense /*B x m x nz*/, double *l /*B x m*/, double *u /*B x m*/);
        /* the reference's explicit OSQP-form matrices (H_FTOCP,q_FTOCP,[F;G],l,u of :259-273), for parity checks only */
int lmpc_qp_dims(lmpc_ctx *, int *nz, int *m_ineq, int *m_eq);

/* ---- device-resident path (bench / rollouts): buffers live in HBM ---------------------------- */
int lmpc_dev_alloc(lmpc_ctx *, long long bytes, void **dptr);
int lmpc_dev_free(lmpc_ctx *, void *dptr);
int lmpc_dev_upload(lmpc_ctx *, void *dptr, const void *host, long long bytes);
int lmpc_dev_download(lmpc_ctx *, void *host, const void *dptr, long long bytes);
int lmpc_dev_sync(lmpc_ctx *);
typedef struct {          /* all device pointers, layouts as in lmpc_step_batch.  Optional (NULL = not wanted): slack, lambda, sTerm, ztNext, ztuNext,
                             ssSel, A, Bm, C, mu, resid, qSel.  With LMPC_FUSE=1 in the environment, batches that run one wavefront per QP take
                             the step as ONE kernel: each wave first runs the regression (MPC.computeLTVdynamics, :140-145) of its own QP,
                             A_i / B_i / C_i stay in LDS and are copied out only if A / Bm / C are given (bit-identical results; measured
                             slower than the two-kernel step, hence off by default). */
    const double *x0, *xLin, *uLin, *uOld, *zt, *xPredPrev; const int *hasPred, *timeStep;
    double *xPred, *uPred, *slack, *lambda, *sTerm, *ztNext, *ztuNext, *ssSel, *A, *Bm, *C, *mu, *resid;
    int *status, *iters;
    double *qSel;          /* B x S or NULL */
} lmpc_step_dev_args;
int lmpc_step_batch_dev(lmpc_ctx *, int B, const lmpc_step_dev_args *args);   /* async on the ctx stream; status as in lmpc_step_batch */

/* ---- caller side of the path, next row of SURVEY 8(f): plant integrator + device-resident closed-loop laps ---- */
/* Map.getGlobalPosition (Track.py:135-189), batched: curvilinear (s, ey) -> inertial (X, Y) for n points on the track given
 * in lmpc_config.track (plotting / logging export, plot.py:50-175).  status[e] = LMPC_ST_NO_SEGMENT where the reference raises. */
int lmpc_global_position_batch(lmpc_ctx *, int n, const double *s /*n*/, const double *ey /*n*/, double *xy /*n x 2*/, int *status /*n*/);

int lmpc_plant_step_batch(lmpc_ctx *, int B, const double *x /*B x 6*/, const double *x_glob /*B x 6*/, const double *u /*B x 2*/,
                          const double *noise /*B x 3 N(0,1) draws*/, double *x_next, double *x_glob_next, int *status);
        /* Simulator.dynModel, fnc/simulator/SysModel.py:56-147 (100 Euler sub-steps, clipped noise) */
int lmpc_rollout_begin(lmpc_ctx *, int B, int T_max, const double *x0 /*B x 6*/, const double *xglob0 /*B x 6*/,
                       const double *xLin0 /*B x (N+1) x 6*/, const double *uLin0 /*B x N x 2*/, const double *noise /*T_max x B x 3*/);
int lmpc_rollout_run(lmpc_ctx *, int max_steps, int *steps_total, int *n_done);
int lmpc_rollout_fetch(lmpc_ctx *, int t0, int t1, double *X /*(t1-t0) x B x 6*/, double *U /*.. x B x 2*/, double *Xglob,
                       int *doneAt /*B*/, int *status /*B*/, double *finalX /*B x 6*/, double *finalXglob /*B x 6*/);
int lmpc_rollout_end(lmpc_ctx *);
        /* Simulator.sim with one LMPC controller per rollout, SysModel.py:22-54, state resident on the device.  The session's device buffers are kept for the next
         * lap of the same shape (a generation loop begins one per lap); lmpc_rollout_release or lmpc_destroy frees them */
int lmpc_rollout_release(lmpc_ctx *);
        /* give the kept session buffers back (55 MB at 1024 rollouts x 400 steps) without destroying the context; not valid between begin and end.  No reference counterpart */
int lmpc_ss_extend_lap(lmpc_ctx *, int lap, const double *x /*n x 6*/, const double *u /*n x 2*/, int n);
/* Undo extensions: keep the first T rows of stored lap `lap` (LapTime <= T <= current rows).  rollout.LmpcGeneration rolls a failed
 * generation back with it, so that a generation either completes or leaves the safe set as it found it.  No reference counterpart. */
int lmpc_ss_truncate_lap(lmpc_ctx *, int lap, int T);
        /* LMPC.addPoint (:466-474) applied to any stored lap: n points appended with s + TrackLength, Qfun counting down */
int lmpc_lti_regression(int device, const double *x /*T x 6*/, const double *u /*T x 2*/, int T, double lamb,
                        double *A /*6 x 6*/, double *B /*6 x 2*/, double *Error /*2 x 6: max; min of the fit residual*/, int *status /*or NULL*/);
        /* Utilities.Regression, fnc/Utilities.py:5-28 (main.py:74-77: the LTI model of the path-following MPC); ridge least squares over one lap */

/* ---- multi-GPU: one process per GPU, RCCL over xGMI.  The QPs / rollouts of a batch are independent given the
 * read-only safe set, so the data path has no collective; the ONE exchange is per lap.  The reference has no counterpart (single
 * process); what is exchanged are the arguments of LMPC.addTrajectory / PredictiveModel.addTrajectory (:418-445, PredictiveModel.py:35-46).
 * The 128-byte id is created on rank 0 and handed to the other processes by the caller (racinglmpc_amd/parallel.py). */
#define LMPC_COMM_ID_BYTES 128
int lmpc_comm_unique_id(unsigned char *id /*LMPC_COMM_ID_BYTES*/);                       /* ncclGetUniqueId */
int lmpc_comm_init(lmpc_ctx *, const unsigned char *id, int rank, int world);            /* ncclCommInitRank on the ctx's device */
int lmpc_comm_destroy(lmpc_ctx *);
int lmpc_comm_info(lmpc_ctx *, int *rank, int *world, int *is_rccl);
int lmpc_comm_allgather_dev(lmpc_ctx *, const void *send_dev, void *recv_dev /*world x bytes*/, long long bytes);   /* ncclAllGather, async on the ctx stream */
int lmpc_comm_allgather(lmpc_ctx *, const void *send_host, void *recv_host /*world x bytes*/, long long bytes);     /* same for small host-side control data */
int lmpc_comm_allreduce_max(lmpc_ctx *, double *v /*n, in/out, host*/, int n);
int lmpc_comm_barrier(lmpc_ctx *);                                                       /* stream drained on every rank */
int lmpc_rollout_exchange(lmpc_ctx *, int K, int T_max, double *records /*world x K x (T_max+1) x 14*/, long long *lens /*world x K, -1 = empty*/,
                          int *n_valid_local);
        /* per-lap exchange of the current rollout session: the rank's K fastest valid laps (finished, <= T_max steps, status clean) are
         * packed on the device from the session logs -- rows t < T: x_t | u_t | x_glob_t, row T_max: state and global state right after the
         * finish line, rollout index -- and all-gathered; every rank receives the same world x K records */

int lmpc_selftest(lmpc_ctx *);                   /* device self test of the cross-lane reduction primitives */
/* Developer switch (environment, read at lmpc_create): LMPC_MW_MAX_BATCH=<n> overrides the largest batch that runs four waves per
 * QP (default: the number of CUs; 0 forces the one-wave kernel everywhere).  Results do not depend on it beyond summation order. */
int lmpc_solver_waves(lmpc_ctx *, int B);         /* wavefronts per QP the solve kernel uses for a batch of B (4 or 2: lmpc_solve_kernel_mw, 1: lmpc_solve_kernel) */
int lmpc_set_profiling(lmpc_ctx *, int every);    /* 0: off; k > 0: HIP events on the launch stream around every k-th launch of each kernel (an event
                                                     record costs ~4 us of stream time: k = 1 adds 15 us to a two-kernel step) */
int lmpc_get_stats(lmpc_ctx *, lmpc_stats *out);  /* drains pending events */
int lmpc_reset_stats(lmpc_ctx *);

/* ---- developer entry points.  They exist in every build, but only the flavours built with the named macro carry the code behind them
 *      (racinglmpc_amd.build.build_flavour); in the product library both return LMPC_E_ARG and say so in lmpc_last_error. ---- */
int lmpc_debug_set_trace(lmpc_ctx *, double *dev_rows /* max_batch x 48 x 6 doubles of device memory, or NULL */);
        /* -DLMPC_TRACE: every later solve launch records, per problem and interior-point iteration, (gap, r_d, r_e | sigma, alpha_p, alpha_d) --
         * the columns of tests/ipm_model.py's trace, for a differential comparison of kernel and model (tools/n40_model.py) */
int lmpc_debug_rollout_peek(lmpc_ctx *, double *xLin /*B x (N+1) x 6*/, double *uLin /*B x N x 2*/, int *status /*B*/, int *rstatus /*B x N*/);
        /* (every build) the rollout session's current linearisation trajectories -- the queries of the NEXT step's regression -- and the status words of the step
         * just taken; any pointer may be NULL.  tools/robustness_sweep.py captures the inputs of a flagged closed-loop regression with it */
int lmpc_debug_rollout_capture(lmpc_ctx *, int on);
        /* (every build) sessions begun after this call also keep the selected safe-set points / Q-values of the step just taken (B x S x 6, B x S): off by default,
         * the rollout step then does not write them */
int lmpc_debug_rollout_qp(lmpc_ctx *, int b0, int n, double *A, double *B, double *C, double *xPred, double *uPred, double *ssSel, double *qSel, double *succ, double *succU,
                          double *lambda, double *ztNext, double *ztuNext, int *iters, int *status);
        /* (every build) the QP the LAST simulated step solved for rollouts b0 .. b0 + n - 1: the regression's A, B, C, the kernel's answer (xPred, uPred, lambda, zt / zt_u),
         * and (capture on) the selection with its successor rows -- what tests/closed_loop_probe.py hands to the oracle; x0 / uOld are rows t - 1 / t - 2 of the
         * session's logs.  Any pointer may be NULL */
int lmpc_debug_exec_audit(lmpc_ctx *, unsigned long long *out16, int reset);
        /* -DLMPC_EXEC_AUDIT: out16[site] = calls of a cross-lane primitive (DPP / permlane / bpermute reductions, MFMA stages) of the built-in
         * kernels that found an incomplete EXEC mask, out16[8 + site] = calls; sites are listed in csrc/lmpc_kernels.hip.h */

#ifdef __cplusplus
}
#endif
#endif /* LMPC_HIP_H */
