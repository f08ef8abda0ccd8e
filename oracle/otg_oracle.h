/*
 * otg_oracle.h — CPU oracle for the internal online trajectory generation (OTG) of the tasks.
 *
 * TEST INFRASTRUCTURE ONLY (see sai2_oracle.h). Plain-C restatement of
 *   (1) the acceleration-limited ("second-order", max_jerk = inf) position interface of the
 *       reference's vendored ruckig 0.10.1 core: brake pre-trajectory, step 1 extremal profiles,
 *       blocked intervals, synchronisation (phase, with time fallback), step 2, sampling and
 *       Ruckig::update — ruckig/include/ruckig/{ruckig,calculator_target,block,profile,brake,
 *       trajectory}.hpp, ruckig/src/ruckig/{brake,position-second-step1,position-second-step2}.cpp;
 *   (2) the two sai2 wrappers, src/helper_modules/OTG_joints.cpp and OTG_6dof_cartesian.cpp.
 *
 * PARITY STATUS: (1) is PINNED against the reference's own code: the ruckig core builds here
 * without Eigen (oracle/Makefile `ref` -> oracle/_ref/libruckig_ref.so) and tests/test_otg_oracle.py
 * compares the two on ruckig's own known answers (ruckig/test/test-target-known.cpp:263-299), on its
 * random second-order test distribution (ruckig/test/test-target.cpp:21-23,1247-1283) and along
 * stepped trajectories; committed fixtures tests/golden/otg_*.npz hold the reference's outputs.
 * (2) needs Eigen and cannot be built: "parity unpinned" there, checked by an independent numpy
 * restatement of the wrappers driving the real ruckig core (tests/golden/make_otg_golden.py).
 * (3) The jerk-limited (third-order) planner is NOT restated: for a generator with a finite max_jerk
 * (enableInternalOtgJerkLimited: JointTask.cpp:383-406, MotionForceTask.cpp:525-538) otg_update calls the REFERENCE's
 * own ruckig — rref_plan_jerk of oracle/_ref/libruckig_ref.so, compiled from the reference's sources (Makefile `ref`),
 * loaded on first use — and samples the trajectory it returns. Stronger than a restatement where it applies: the
 * oracle's planner there is the reference's code. Without oracle/_ref a jerk-limited generator reports
 * OTG_ERROR_NO_REFERENCE_PLANNER (tests skip). The restated parts around it stay: Ruckig::update, sampling, wrappers.
 */
#ifndef OTG_ORACLE_H_
#define OTG_ORACLE_H_

#ifdef __cplusplus
extern "C" {
#endif

#ifndef OTG_MAX_DOF
#define OTG_MAX_DOF 7 /* DoFs of the largest generator; the 8-joint build of the oracle sets 8 */
#endif

enum {
	OTG_WORKING = 0,
	OTG_FINISHED = 1,
	OTG_ERROR_INVALID_INPUT = -100,
	OTG_ERROR_TRAJECTORY_DURATION = -101,
	OTG_ERROR_EXECUTION_TIME_CALCULATION = -110,
	OTG_ERROR_SYNCHRONIZATION_CALCULATION = -111,
	OTG_ERROR_NO_REFERENCE_PLANNER = -900 /* ours: jerk-limited planning needs oracle/_ref/libruckig_ref.so */
};
enum { OTG_SYNC_TIME = 0, OTG_SYNC_PHASE = 2 };

/* BrakeProfile, second-order part (brake.hpp:16-79) */
typedef struct {
	double duration, t, a, p, v;
} otg_brake;

/* Profile (profile.hpp:33-56), second-order members */
typedef struct {
	double t[7], t_sum[7], a[8], v[8], p[8];
	otg_brake brake;
	double pf, vf, af;
	int limits, direction, control_signs; /* direction: 0 UP, 1 DOWN */
} otg_profile;

/* InputParameter subset (input_parameter.hpp:73-104) */
typedef struct {
	int n, synchronization;
	double cp[OTG_MAX_DOF], cv[OTG_MAX_DOF], ca[OTG_MAX_DOF];
	double tp[OTG_MAX_DOF], tv[OTG_MAX_DOF];
	double vmax[OTG_MAX_DOF], amax[OTG_MAX_DOF];
	double jmax[OTG_MAX_DOF]; /* INFINITY: acceleration-limited (the default); finite: the jerk-limited interface */
} otg_input;

/* a third-order profile as the reference's planner returns it (ruckig_ref_harness.cpp: rref_plan_jerk) */
#define OTG_PROF3_LEN 49
typedef struct {
	otg_profile prof[OTG_MAX_DOF];
	double duration;
	int third_order;
	double prof3[OTG_MAX_DOF][OTG_PROF3_LEN];
} otg_traj;

/* OutputParameter subset (output_parameter.hpp:27-52) */
typedef struct {
	double np[OTG_MAX_DOF], nv[OTG_MAX_DOF], na[OTG_MAX_DOF];
	double time;
	int new_calculation;
	otg_traj traj;
} otg_output;

/* Ruckig<> instance state (ruckig.hpp:24-33) */
typedef struct {
	double delta_time;
	int current_input_initialized;
	otg_input current_input;
} otg_ruckig;

/* TargetCalculator::calculate (calculator_target.hpp:249-532) */
int otg_calculate(const otg_input* inp, otg_traj* traj);
/* Trajectory::at_time (trajectory.hpp:65-142,182-193) */
void otg_at_time(const otg_traj* traj, int n, double time, double* p, double* v, double* a);
/* Ruckig::update (ruckig.hpp:180-216) */
int otg_update(otg_ruckig* otg, const otg_input* inp, otg_output* out);

/* ---- OTG_joints (OTG_joints.h / OTG_joints.cpp) ---- */
typedef struct {
	int dim, goal_reached, result_value, target_set;
	otg_ruckig otg;
	otg_input input;
	otg_output output;
} otg_joints;

void otg_joints_init(otg_joints* o, int dim, const double* initial_position, double loop_time);
void otg_joints_reinitialize(otg_joints* o, const double* initial_position);
void otg_joints_set_limits(otg_joints* o, const double* max_velocity, const double* max_acceleration);
void otg_joints_disable_jerk_limits(otg_joints* o);
void otg_joints_set_max_jerk(otg_joints* o, const double* max_jerk); /* OTG_joints::setMaxJerk (OTG_joints.cpp:73-86) */
int otg_jerk_planner_available(void);
void otg_joints_set_goal(otg_joints* o, const double* goal_position, const double* goal_velocity);
void otg_joints_update(otg_joints* o);

/* ---- OTG_6dof_cartesian (OTG_6dof_cartesian.h / .cpp); rotations row-major 3x3 ---- */
typedef struct {
	int goal_reached, result_value, target_pos_set, goal_ori_set;
	double reference_frame[9], goal_orientation[9], goal_angular_velocity[3];
	otg_ruckig otg;
	otg_input input;
	otg_output output;
} otg_cartesian;

void otg_cartesian_init(otg_cartesian* o, const double* initial_position,
						const double* initial_orientation, double loop_time);
void otg_cartesian_reinitialize(otg_cartesian* o, const double* position, const double* orientation);
void otg_cartesian_reinitialize_linear(otg_cartesian* o, const double* position);
void otg_cartesian_reinitialize_angular(otg_cartesian* o, const double* orientation);
void otg_cartesian_set_limits(otg_cartesian* o, double max_lin_vel, double max_lin_acc,
							  double max_ang_vel, double max_ang_acc);
/* setMaxJerk / disableJerkLimits (OTG_6dof_cartesian.cpp:126-136, .h:167-169); <= 0 or inf: disabled */
void otg_cartesian_set_max_jerk(otg_cartesian* o, double max_lin_jerk, double max_ang_jerk);
void otg_cartesian_set_goal_position(otg_cartesian* o, const double* pos, const double* lin_vel);
void otg_cartesian_set_goal_orientation(otg_cartesian* o, const double* rot, const double* ang_vel);
void otg_cartesian_update(otg_cartesian* o);
void otg_cartesian_next_orientation(const otg_cartesian* o, double* rot);
void otg_cartesian_next_angular(const otg_cartesian* o, double* ang_vel, double* ang_acc);

/* rotation helpers (Eigen AngleAxisd semantics, restated from the published algorithm) */
void otg_rot_to_angle_axis_vec(const double* R, double* v);
void otg_angle_axis_vec_to_rot(const double* v, double* R);

#ifdef __cplusplus
}
#endif
#endif
