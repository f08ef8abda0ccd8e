/*
 * sai2_oracle.h — CPU oracle for the batched operational-space controller.
 *
 * TEST INFRASTRUCTURE ONLY. This is a plain-C, FP64, one-robot-at-a-time restatement of the
 * reference's algorithm (mikael-jorda/sai2-primitives-perso @ 2024-12-18) for the hot path
 * RobotController::updateControllerTaskModels / computeControlTorques and the tasks under it.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use it; the product
 * (sai2-primitives-perso_amd/) never links, imports or calls anything in this directory.
 *
 * PARITY STATUS: "parity unpinned" at the sai2-model / Eigen boundary (the internal OTG's planner is
 * the exception: otg_oracle.h, pinned against the reference's own ruckig core in oracle/_ref). The reference cannot be
 * compiled here (Eigen3 and sai2-model are absent, no network: SURVEY.md §8(c)) and ships no
 * tests, fixtures or golden vectors for src/. The oracle is pinned instead by
 *   (1) an independent numpy/float64 restatement (tests/golden/make_golden.py) whose outputs are
 *       committed under tests/golden/, and
 *   (2) analytic known-answers (SURVEY.md App. A-KA), checked in tests/test_oracle.py.
 * sai2-model helper semantics (operationalSpaceMatrices, matrixRangeBasis, computePseudoInverse,
 * orientationError) are DEFINED here as in SURVEY.md App. D.
 *
 * It shares only the POD configuration structs of include/sai2b.h with the product, so the same
 * inputs can be fed to both. Array layout is the product's: SoA, batch-minor, [C][B].
 */
#ifndef SAI2_ORACLE_H_
#define SAI2_ORACLE_H_

#include "../include/sai2b.h"

#ifdef __cplusplus
extern "C" {
#endif

typedef struct oracle_ctx oracle_ctx;

/* independent restatements of the host-side config helpers (compared with the product's in
 * tests/test_host_config.py) */
int oracle_panda_model(sai2b_robot_model* model);
int oracle_model_merge_fixed_body(sai2b_robot_model* model, int link, const double xyz[3],
								  const double rpy[3], double mass, const double com[3],
								  const double inertia[6]);
int oracle_default_joint_task(sai2b_task_config* cfg, const char* name, int task_dof,
							  const double* selection);
int oracle_default_motion_force_task(sai2b_task_config* cfg, const char* name, int link,
									 const double frame_pos[3], const double* frame_rot,
									 int n_trans, const double* dirs_trans, int n_rot,
									 const double* dirs_rot);

oracle_ctx* oracle_create(const sai2b_robot_model* model, const sai2b_task_config* tasks,
						  int n_tasks, int batch);
void oracle_destroy(oracle_ctx* ctx);
const char* oracle_last_error(void);
/* number of OpenMP threads for the batch loop (1 = the reference's execution model) */
void oracle_set_threads(oracle_ctx* ctx, int n_threads);

int oracle_update_task_config(oracle_ctx* ctx, int task, const sai2b_task_config* cfg);
int oracle_enable_gravity_compensation(oracle_ctx* ctx, int enable);
int oracle_set_state(oracle_ctx* ctx, const double* q, const double* dq);
int oracle_set_mft_goals(oracle_ctx* ctx, int task, const double* pos, const double* rot,
						 const double* lin_vel, const double* ang_vel, const double* lin_acc,
						 const double* ang_acc);
int oracle_set_mft_goal_wrench(oracle_ctx* ctx, int task, const double* force,
							   const double* moment);
int oracle_set_mft_sensed_wrench(oracle_ctx* ctx, int task, const double* force,
								 const double* moment);
int oracle_set_jt_goals(oracle_ctx* ctx, int task, const double* q_goal, const double* dq_goal,
						const double* ddq_goal);
int oracle_reinitialize(oracle_ctx* ctx);
int oracle_update_task_models(oracle_ctx* ctx);
int oracle_compute_control_torques(oracle_ctx* ctx, double* tau, int with_compensation);
int oracle_tick(oracle_ctx* ctx, double* tau);

/* TemplateTask virtuals on one task (TemplateTask.h:42-88): N_prec [49][B] or NULL = identity; tau_prec [7][B] or
 * NULL = the no-argument computeTorques(); like the reference, computeTorques uses the model cached by the last
 * updateTaskModel and the robot model of the current state */
int oracle_task_update_model(oracle_ctx* ctx, int task, const double* N_prec);
int oracle_task_compute_torques(oracle_ctx* ctx, int task, const double* tau_prec, double* tau);
int oracle_task_reinitialize(oracle_ctx* ctx, int task);
int oracle_task_get_nullspaces(oracle_ctx* ctx, int task, double* N, double* N_prec, double* N_total);

int oracle_get_task_nullspace(oracle_ctx* ctx, int task, double* N_total);
int oracle_get_task_torques(oracle_ctx* ctx, int task, double* tau_task);
int oracle_get_mft_singularity(oracle_ctx* ctx, int task, double* sigma, double* alpha,
							   double* ns_rank);
int oracle_get_model(oracle_ctx* ctx, int task, double* M, double* J, double* pos, double* rot);
/* extra introspection for fixtures: M^-1 [49][B]; gravity vector [7][B] */
int oracle_get_minv(oracle_ctx* ctx, double* Minv);
int oracle_get_gravity(oracle_ctx* ctx, double* g);
/* MFT: Lambda_ns embedded as U_ns Lambda_ns U_ns^T [36][B], same for the modified one */
int oracle_get_mft_lambda(oracle_ctx* ctx, int task, double* Lambda_ns_full,
						  double* Lambda_ns_mod_full);
/* MFT: F_unit and F_force of the last computeTorques, [6][B] each */
int oracle_get_mft_task_forces(oracle_ctx* ctx, int task, double* F_unit, double* F_force);
/* simulation harness (the reference uses the external sai2-simulation,
 * examples/05-...cpp:215-236, so these are definitions of ours): advance every robot by one control
 * period dt under joint torques tau ([7][B], NULL = zero) held constant, with `substeps`
 * semi-implicit Euler steps of the rigid-body dynamics M qdd + C dq (+ g) = tau */
int oracle_sim_step(oracle_ctx* ctx, const double* tau, double dt, int substeps, int with_gravity);
int oracle_get_state(oracle_ctx* ctx, double* q, double* dq);
int oracle_get_bias(oracle_ctx* ctx, int with_gravity, double* bias);
int oracle_get_mft_status(oracle_ctx* ctx, int task, double* pos, double* rot, double* sensed_force_world,
						  double* sensed_moment_world, double* pos_error, double* ori_error, double* pos_error_norm,
						  double* ori_error_norm);
int oracle_reset_integrators(oracle_ctx* ctx, int task, int which);
/* desired state of the last computeTorques = goal, or the internal OTG's next state
 * (JointTask.h:182-198 getDesired*, MotionForceTask.h getDesired*); any pointer may be NULL */
int oracle_get_mft_integrators(oracle_ctx* c, int task, double* out);
int oracle_get_jt_desired(oracle_ctx* ctx, int task, double* q, double* dq, double* ddq);
int oracle_get_mft_desired(oracle_ctx* ctx, int task, double* pos, double* rot, double* lin_vel,
						   double* ang_vel, double* lin_acc, double* ang_acc);
/* internal OTG flags per robot, as doubles [B]: isGoalReached(), last ruckig Result */
int oracle_get_otg_status(oracle_ctx* ctx, int task, double* goal_reached, double* result);
/* MFT singularity classification of the last update: type per robot (0 none, 1 type-1, 2 type-2
 * of the first singular column), type-1 and type-2 counters; as doubles, [B] each */
int oracle_get_mft_sh_state(oracle_ctx* ctx, int task, double* first_type, double* c1, double* c2);
/* JT: R M_partial R^T and R M_partial_mod R^T, [k0*k0][B] */
int oracle_get_jt_inertia(oracle_ctx* ctx, int task, double* M_partial_full,
						  double* M_partial_mod_full);

/* small dense kernels exposed for unit tests */
void oracle_svd(int m, int n, const double* A, double* U, double* s, double* V);
int oracle_inverse(int n, const double* A, double* Ainv);
int oracle_range_basis(int m, int n, const double* A, double tol, double* R);

#ifdef __cplusplus
}
#endif
#endif
