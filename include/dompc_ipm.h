/*
 * dompc_ipm.h - C ABI of the MI355X-native structured interior-point backend that replaces
 * the `nlpsol('ipopt', ...)` call on do-mpc's MPC.make_step() path.
 *
 * Reference interfaces replaced (paths relative to the do-mpc source tree):
 *   do_mpc/controller/_mpc.py:1326-1328   self.S = castools.nlpsol('S', 'ipopt', nlp, opts)   -> dompc_create
 *   do_mpc/optimizer.py:754-770           r = self.S(x0=, lbx=, ubx=, lbg=, ubg=, p=, ...)    -> dompc_solve
 *   do_mpc/optimizer.py:772-778           r['x'], r['g'], r['lam_g'], r['lam_x'], S.stats()   -> outputs + dompc_stats
 *   do_mpc/sampling/_sampler.py:198-228   one make_step per sample, fanned out by processes   -> dompc_solve_batch*
 * The NLP itself is *not* passed as a symbolic graph: the model functions were lowered to a
 * gfx950 code object at setup() (do_mpc_amd/lowering.py) and the multi-stage structure
 * (do_mpc/optimizer.py:998-1048, do_mpc/controller/_mpc.py:1126-1245) is described by the
 * integer tables below.
 *
 * Conventions
 *   - all vectors are contiguous float64 in the reference's canonical order (opt_x, opt_p, g);
 *   - multipliers follow CasADi's sign convention  L = f + lam_g'g + lam_x'x;
 *   - return code 0 = ok, != 0 = infrastructure error (HIP, allocation, bad argument); the text is
 *     available from dompc_last_error().  Solver non-convergence is NOT an error: it is reported
 *     in dompc_stats.success / return_status, results are still written (matches optimizer.py:770-778);
 *   - the caller owns every buffer it passes; the library owns device memory, streams and the loaded
 *     code object inside the handle; a handle is not thread-safe, distinct handles are independent;
 *   - create the handle after fork(): no HIP state is touched before dompc_create().
 */
#ifndef DOMPC_IPM_H
#define DOMPC_IPM_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct dompc_handle dompc_handle;

/* IPOPT-named algorithm options (defaults = IPOPT 3.14 defaults, see dompc_default_options). */
typedef struct dompc_options {
  double tol;                /* ipopt.tol                      1e-8  */
  double dual_inf_tol;       /* ipopt.dual_inf_tol             1.0   */
  double constr_viol_tol;    /* ipopt.constr_viol_tol          1e-4  */
  double compl_inf_tol;      /* ipopt.compl_inf_tol            1e-4  */
  double acceptable_tol;     /* ipopt.acceptable_tol           1e-6  */
  double mu_init;            /* ipopt.mu_init                  0.1   */
  double kappa_mu;           /* mu_linear_decrease_factor      0.2   */
  double theta_mu;           /* mu_superlinear_decrease_power  1.5   */
  double kappa_eps;          /* barrier_tol_factor             10    */
  double tau_min;            /* ipopt.tau_min                  0.99  */
  double bound_push;         /* ipopt.bound_push               0.01  */
  double bound_frac;         /* ipopt.bound_frac               0.01  */
  double bound_relax_factor; /* ipopt.bound_relax_factor       1e-8  */
  double nlp_scaling_max_gradient; /*                          100   */
  double delta_w_0;          /* first_hessian_perturbation     1e-4  */
  double delta_w_min;        /* min_hessian_perturbation       1e-20 */
  double delta_w_max;        /* max_hessian_perturbation       1e20  */
  double kappa_w_minus;      /* perturb_dec_fact               1/3   */
  double kappa_w_plus;       /* perturb_inc_fact               8     */
  double kappa_w_plus_bar;   /* perturb_inc_fact_first         100   */
  int32_t max_iter;          /* ipopt.max_iter                 3000  */
  int32_t acceptable_iter;   /* ipopt.acceptable_iter          15    */
  int32_t obj_scaling;       /* gradient-based objective scaling on/off (1) */
  int32_t max_soc;           /* ipopt.max_soc: second-order correction attempts per iteration   4 (0 = off) */
  double constr_mult_init_max; /* ipopt.constr_mult_init_max: least-squares multiplier estimate at the starting point,
                                  discarded if its max-norm is above this value   1000 (0 = start from lambda = 0);
                                  models with nl_cons rows always start from lambda = 0 */
  int32_t watchdog_shortened_iter_trigger; /* ipopt.watchdog_shortened_iter_trigger: consecutive iterations with a shortened step
                                              after which the watchdog procedure starts   10 (0 = off) */
  int32_t watchdog_trial_iter_max;         /* ipopt.watchdog_trial_iter_max: full steps it may take without the filter   3 */
} dompc_options;

/* Description of one multi-stage problem class (fixed at MPC.setup()). All pointers are host
 * pointers and are copied by dompc_create(). */
typedef struct dompc_problem_desc {
  int32_t nx, nu, np, ntvp, ne, ns;     /* model dims; ne = nl_cons rows/edge, ns = slacks/(stage,scenario) */
  int32_t deg, ni, M;                   /* collocation; M = ni*(deg+1) stored slots, 0 = discrete model     */
  int32_t N;                            /* horizon                                                         */
  int32_t n_opt_x, n_opt_p, n_g;        /* canonical vector sizes                                          */
  int32_t n_nodes, n_edges, n_dummy;
  int32_t p_off_tvp, p_off_p, p_off_uprev; /* offsets inside opt_p (= [_x0 ; _tvp ; _p ; _u_prev])       */
  const int32_t* level_node_start;      /* [N+2]  nodes are ordered by (stage k, scenario s)               */
  const int32_t* node_level;            /* [n_nodes]                                                       */
  const int32_t* node_x_off;            /* [n_nodes] offset of _x[k,s,-1] in opt_x                          */
  const int32_t* node_u_off;            /* [n_nodes] offset of _u[k,s]   (-1 at stage N)                    */
  const int32_t* node_eps_off;          /* [n_nodes] offset of _eps[k,s] (-1 if none)                       */
  const int32_t* node_child_start;      /* [n_nodes] first outgoing edge                                   */
  const int32_t* node_child_count;      /* [n_nodes]                                                       */
  const int32_t* node_parent;           /* [n_nodes] parent node (-1 root)                                 */
  const int32_t* node_in_edge;          /* [n_nodes] incoming edge (-1 root)                                */
  const int32_t* edge_parent;           /* [n_edges]                                                       */
  const int32_t* edge_child;            /* [n_edges]                                                       */
  const int32_t* edge_pidx;             /* [n_edges] row of opt_p['_p'] used on this edge                   */
  const int32_t* edge_w_off;            /* [n_edges] offset of _x[k+1,c,0] (collocation slots of the edge)  */
  const int32_t* edge_row0;             /* [n_edges] first constraint row of the edge in g                  */
  const int32_t* edge_level;            /* [n_edges] stage k of the parent                                  */
  const double*  edge_omega;            /* [n_edges] scenario weight 1/n_scenarios[k+1]                     */
  const int32_t* dummy_idx;             /* [n_dummy] opt_x entries used by no constraint/cost               */
  const char*    code_object_path;      /* gfx950 code object built from the lowered model                  */
  const char*    model_hash;            /* must equal the hash embedded in the code object                  */
  int32_t device;                       /* HIP device ordinal                                              */
  int32_t max_batch;                    /* largest batch that will be passed to *_batch calls               */
  int32_t n_slots;                      /* concurrent problem slots (workgroups); 0 = what the device keeps resident */
  int32_t block_threads;                /* threads per problem in batch mode: 64, 128 or 256; 0 = choose from max_batch */
  dompc_options opts;
} dompc_problem_desc;

typedef struct dompc_stats {
  int32_t success;          /* 1 = converged to tol (or acceptable level)                                  */
  int32_t status;           /* 0 Solve_Succeeded, 1 Solved_To_Acceptable_Level, 2 Maximum_Iterations_Exceeded,
                               3 Error_In_Step_Computation, 4 Invalid_Number_Detected, 5 Internal_Error (a peer
                               workgroup / rank never arrived), 6 User_Requested_Stop (dompc_abort / watchdog)  */
  int32_t iter_count;
  int32_t n_reg;            /* iterations that needed Hessian regularisation                               */
  int32_t n_ls_fail;        /* line searches that hit alpha_min (no restoration phase)                     */
  int32_t n_sweeps;         /* derivative sweeps executed (model evaluation + condensing of every edge)        */
  int32_t n_trials;         /* function-only trial sweeps of the line search                                 */
  int32_t n_soc;            /* second-order correction solves (each one more sweep + Riccati pass)            */
  int32_t n_watchdog;       /* watchdog procedures started (IPOPT: watchdog_shortened_iter_trigger)            */
  int32_t reserved0;
  double  mu;
  double  obj;              /* unscaled objective                                                          */
  double  inf_pr, inf_du, inf_compl; /* scaled errors at exit                                                  */
  double  obj_scaling;
  double  t_wall_total;     /* filled by the host side: wall time of the call / batch                      */
} dompc_stats;

void dompc_default_options(dompc_options* opts);

int  dompc_create(const dompc_problem_desc* desc, dompc_handle** out);
void dompc_destroy(dompc_handle* h);
const char* dompc_last_error(const dompc_handle* h);   /* h may be NULL: error of the last failed create */
const char* dompc_status_string(int32_t status);

/* One solve, host buffers (what Optimizer.solve does). lam_x0/lam_g0 may be NULL (they are ignored,
 * like IPOPT with warm_start_init_point=no). Any output pointer may be NULL. */
int dompc_solve(dompc_handle* h,
                const double* x0, const double* lbx, const double* ubx,
                const double* lbg, const double* ubg, const double* p,
                const double* lam_x0, const double* lam_g0,
                double* x, double* g, double* lam_x, double* lam_g, double* f,
                dompc_stats* stats);

/* B independent solves, host buffers.  x0 and p are [B][n]; bounds are shared by the batch. */
int dompc_solve_batch(dompc_handle* h, int32_t B,
                      const double* x0, const double* lbx, const double* ubx,
                      const double* lbg, const double* ubg, const double* p,
                      double* x, double* g, double* lam_x, double* lam_g, double* f,
                      dompc_stats* stats);

/* Same with DEVICE buffers (inputs already resident in HBM), asynchronous on `stream`
 * (a hipStream_t passed as void*; NULL = default stream).  stats is a device pointer [B]. */
int dompc_solve_batch_device(dompc_handle* h, int32_t B,
                             const double* x0, const double* lbx, const double* ubx,
                             const double* lbg, const double* ubg, const double* p,
                             double* x, double* g, double* lam_x, double* lam_g, double* f,
                             dompc_stats* stats, void* stream);

/* Model-evaluation sweep only (the "Jacobian sweep" of the IPM iteration): for B iterates evaluate
 * g(x), and per edge the linearised dynamics [A|B], c and the condensed Lagrangian-Hessian block.
 * DEVICE buffers: x [B][n_opt_x], lam [B][n_g], p [B][n_opt_p]; outputs g [B][n_g],
 * blocks [B][n_edges][dompc_sweep_block_doubles()] (may be NULL: the sweep runs, only g is copied out - the
 * sweep-only roofline of bench.py).  Launched in the shape of dompc_solve_batch_device for the same B.
 * Used for roofline measurement and parity. */
int dompc_sweep_batch_device(dompc_handle* h, int32_t B,
                             const double* x, const double* lam, const double* p,
                             double* g, double* blocks, void* stream);
int64_t dompc_sweep_block_doubles(const dompc_handle* h);

/* Debug/parity entry: one Newton direction at (x, lam_g, z_l, z_u, mu) exactly as given (no push,
 * no scaling).  Host buffers.  dx [n_opt_x], dlam [n_g].  delta_w = primal regularisation to apply. */
int dompc_debug_newton_step(dompc_handle* h,
                            const double* x, const double* lam_g, const double* zl, const double* zu,
                            const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                            const double* p, double mu, double delta_w,
                            double* dx, double* dlam, double* rd, double* c);

/* Newton direction of the primal-dual system at a CONVERGED point of the barrier problem (x, lam_g, z_l, z_u at barrier
 * parameter mu; bounds as the solver relaxed them): like dompc_debug_newton_step, but the slack variables of the nl_cons
 * rows take their values at the point (s = d(x), multipliers mu / distance to the relaxed row bounds) instead of the pushed
 * starting values.  The building block of the parametric sensitivities (do_mpc/differentiator/_nlpdifferentiator.py:
 * 792-841 solves the same system densely): dv/dp_j = [d(p + h e_j) - d(p)] / h.  Host buffers. */
int dompc_newton_step_at_solution(dompc_handle* h,
                                  const double* x, const double* lam_g, const double* zl, const double* zu,
                                  const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                  const double* p, double mu, double* dx, double* dlam);

/* The same for B parameter vectors p (row-major B x n_opt_p) at ONE point: one workgroup per vector, B directions in one launch
 * (dx: B x n_opt_x, dlam: B x n_g) - all columns of a sensitivity matrix at once (do_mpc_amd/differentiator.py; the reference
 * factorises the dense KKT matrix once and solves for all right-hand sides, _nlpdifferentiator.py:792-841). */
int dompc_newton_steps_at_solution(dompc_handle* h, int32_t B,
                                   const double* x, const double* lam_g, const double* zl, const double* zu,
                                   const double* lbx, const double* ubx, const double* lbg, const double* ubg,
                                   const double* p, double mu, double* dx, double* dlam);

/* Iteration trace of problem 0 of the last solve: rows of 8 doubles (it, mu, E0, inf_pr, inf_du,
 * +-alpha (negative: line search failed), delta_w, obj). */
int dompc_debug_get_trace(dompc_handle* h, double* out, int32_t max_rows);

/* Stop request: running and queued solves of this handle leave their IPM loop at the next iteration and report
 * status 6 (User_Requested_Stop, success = 0); may be called from another thread while a solve is in flight.
 * stop = 0 re-arms the handle.  The blocking entry points (dompc_solve, dompc_solve_batch, the service loop of a
 * sharded solve) raise it themselves when the device has not finished within the watchdog time (environment
 * variable DOMPC_WATCHDOG_S, default 600 s) and return an error if the kernel does not react. */
int dompc_abort(dompc_handle* h, int32_t stop);

int64_t dompc_workspace_bytes(const dompc_handle* h);
int32_t dompc_num_slots(const dompc_handle* h);

/* ---- tree sharding of ONE problem over several ranks / GPUs --------------------
 * Replaces nothing in the reference (IPOPT/MUMPS is single-process); it is the multi-GPU path the
 * north star asks for: the sub-trees below `cut_level` of the scenario tree built by
 * do_mpc/optimizer.py:998-1048 (_setup_scenario_tree) go to the ranks in contiguous blocks, the stages
 * above are replicated, and the ranks exchange (a) the condensed contributions of the cut edges to
 * their parent nodes in the Riccati recursion and (b) the scalar reductions of the IPM (norms, step
 * sizes, filter quantities) - all as element-wise SUMs over a small exchange buffer.
 * The collective itself is supplied by the caller (RCCL all-reduce through torch.distributed in
 * do_mpc_amd/solver.py): `allreduce(ctx, buf, count)` must return when buf[0..count) holds the sum
 * over all ranks.  While a sharded solve runs the library serves the kernel's exchange requests from
 * the calling thread, so dompc_solve* returns only when the solve is finished.
 * Masks (host arrays, copied): 0 = another rank's, 1 = mine, 2 = replicated on every rank. */
typedef void (*dompc_allreduce_fn)(void* ctx, double* buf, int32_t count);
typedef struct dompc_shard_desc {
  int32_t rank, world, cut_level, n_cut;
  const int8_t* x_mask;      /* n_opt_x */
  const int8_t* g_mask;      /* n_g */
  const int8_t* edge_mask;   /* n_edges */
  const int8_t* node_mask;   /* n_nodes */
  const int32_t* node_cut;   /* n_nodes: index of a cut parent among the cut parents, else -1 */
  double* xbuf;              /* exchange buffer, DEVICE memory owned by the caller */
  int64_t xbuf_doubles;      /* its length; must be >= dompc_exchange_doubles(h, world, n_cut) */
  dompc_allreduce_fn allreduce;
  void* ctx;
} dompc_shard_desc;
/* Native collective: with allreduce == NULL in dompc_shard_desc the runtime calls RCCL itself
 * (ncclAllReduce, double, sum, on its own communicator and stream) from the service loop - no Python in the
 * exchange path.  The library is opened with dlopen (pass the path of the librccl already loaded in the process,
 * e.g. torch/lib/librccl.so, so that one copy is used).  Rank 0 creates the 128-byte unique id, the caller
 * distributes it (any transport), every rank joins with dompc_rccl_init before dompc_set_sharding. */
int dompc_rccl_unique_id(dompc_handle* h, const char* librccl_path, uint8_t id[128]);
int dompc_rccl_init(dompc_handle* h, const char* librccl_path, const uint8_t id[128], int32_t rank, int32_t world);

/* doubles the exchange buffer needs for (world, n_cut) */
int64_t dompc_exchange_doubles(const dompc_handle* h, int32_t world, int32_t n_cut);
/* desc == NULL switches sharding off again */
int dompc_set_sharding(dompc_handle* h, const dompc_shard_desc* desc);
/* exchanges (element-wise SUMs over the exchange buffer) the last sharded solve asked for; divided by its iteration count:
 * the collectives per interior-point iteration */
int64_t dompc_last_exchange_count(const dompc_handle* h);
/* the launch-shape-specific sibling code object `<name>_batch.hsaco` of the handle: 0 = none next to the general object, 1 = loaded and
 * launched for batches of one wavefront per problem, 2 = found but built from other sources or for another model - not used */
int dompc_batch_object_state(const dompc_handle* h);
/* edges a wavefront handles at a time in the derivative sweep of this model class: 4 = the quad sweep (csrc/dompc_quad.h: single finite
 * element, no nl_cons rows, at most 14 stage variables), 1 = the wavefront-per-edge paths */
int dompc_edges_per_wavefront(const dompc_handle* h);

/* ---- batched plant integration ---------------------------------------------------------
 * Replaces the integrator object of do_mpc.simulator.Simulator (do_mpc/simulator.py:381-416:
 * casadi.integrator('simulator', 'cvodes', dae, t0, t_step, {abstol, reltol}); discrete models: the `simulator`
 * Function of simulator.py:363-378) and its call in Simulator.make_step (simulator.py:757-850): x, u, tvp, p, w in,
 * x_next and y = meas(x_next, u, tvp, p) + v out, in physical units - for B samples at once, one GPU thread per sample,
 * so that an x0 batch stays resident in HBM between the controller's make_step calls.  The model's right-hand side and
 * measurement function come from a per-model gfx950 code object (do_mpc_amd/lowering.py:lower_plant).
 * Methods (dompc_plant_set_method): explicit Dormand-Prince 5(4) and the implicit SDIRK 4(3) of Hairer & Wanner, both with
 * per-sample step-size control (local error at 1/100 of abstol / reltol); default 0 = explicit, a sample that needs more than
 * `explicit_limit` explicit steps (stiff) repeats its interval with the implicit method - CVODES / IDAS of the reference are
 * implicit.  Algebraic states (semi-explicit index-1 DAE): solved for by Newton's method inside every right-hand-side
 * evaluation, per-sample start values carried from call to call (seed: dompc_plant_set_z0, the reference's
 * simulator.set_initial_guess / sim_z_num, simulator.py:603-620).  status[b]: bit 0 = step limit reached, NaN right-hand
 * side or a failed algebraic / stage solve (x_next is the state reached so far), bit 1 = the implicit method produced the
 * result, steps taken = status[b] >> 8.
 * shared_mask: bit 0/1/2/3/4 set = u/tvp/p/w/v is ONE row shared by all samples instead of [B][n]. */
typedef struct dompc_plant dompc_plant;
typedef struct dompc_plant_desc {
  int32_t nx, nu, np, ntvp, nw, nv, ny;
  int32_t discrete;                  /* 1: x_next = rhs(x, u, tvp, p, w) (no integration)                     */
  const char* code_object_path;      /* gfx950 code object built from the lowered plant                       */
  const char* model_hash;            /* must equal the hash embedded in the code object (NULL = no check)      */
  int32_t device;
  int32_t max_steps;                 /* per sample and call; 0 = 200000                                        */
  double t_step, reltol, abstol;     /* settings.t_step / reltol / abstol of the reference's SimulatorSettings */
} dompc_plant_desc;
int  dompc_plant_create(const dompc_plant_desc* desc, dompc_plant** out);
void dompc_plant_destroy(dompc_plant* h);
/* method: 0 explicit with implicit repeat for stiff samples (default), 1 explicit only, 2 implicit only;
 * explicit_limit: explicit steps per interval after which a sample counts as stiff (0 = keep, default 4000) */
int  dompc_plant_set_method(dompc_plant* h, int32_t method, int32_t explicit_limit);
/* algebraic states: their number, and the Newton start for every sample (host array of that many values, NULL = zeros).
 * Carry-over: ROW b of a call continues from the values row b found at the end of the previous call - meaningful only when the rows of
 * consecutive calls are the same trajectories.  dompc_plant_step_batch_device (resident closed loops) always carries; the host entry
 * dompc_plant_step_batch starts every call from z0 unless dompc_plant_set_z_carry(h, 1) says its rows correspond across calls
 * (Simulator.make_step: one trajectory).  A batch larger than any before starts all rows from z0. */
int32_t dompc_plant_num_alg_states(const dompc_plant* h);
int  dompc_plant_set_z0(dompc_plant* h, const double* z0);
int  dompc_plant_set_z_carry(dompc_plant* h, int32_t on);
const char* dompc_plant_last_error(const dompc_plant* h);     /* h may be NULL: error of the last failed create */
/* host buffers; w, v, y, status may be NULL */
int dompc_plant_step_batch(dompc_plant* h, int32_t B, const double* x, const double* u, const double* tvp, const double* p,
                           const double* w, const double* v, int32_t shared_mask, double* x_next, double* y, int32_t* status);
/* DEVICE buffers, asynchronous on `stream` (hipStream_t as void*) */
int dompc_plant_step_batch_device(dompc_plant* h, int32_t B, const double* x, const double* u, const double* tvp,
                                  const double* p, const double* w, const double* v, int32_t shared_mask, double* x_next,
                                  double* y, int32_t* status, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DOMPC_IPM_H */
