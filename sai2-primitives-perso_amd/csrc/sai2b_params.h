// sai2b_params.h — device-visible parameter block shared by the host code and the kernels.
// One block per ctx lives in device memory; every field is batch-uniform, so the kernels read it
// with scalar loads.
#pragma once
#include "../../include/sai2b.h"

namespace sai2b {

// joints of the robots this build of the library serves: every translation unit is compiled once per supported
// value (Makefile: 4, 6, 7, 8) with its symbols suffixed (sai2b_dof_rename.h); sai2b_dispatch.cpp routes the
// public entry points by the model's / the context's dof
#ifndef SAI2B_N
#define SAI2B_N 7
#endif
constexpr int N = SAI2B_N;
static_assert(N >= 1 && N <= SAI2B_MAX_DOF, "SAI2B_N");

// sai2-model subset: constant part of the kinematic/dynamic model
struct DevModel {
	double E[N][9];	  // joint frame rotation in the parent link frame (from URDF rpy)
	double xyz[N][3]; // joint frame origin in the parent link frame
	double mass[N];
	double com[N][3];
	double inertia[N][6]; // ixx iyy izz ixy ixz iyz at the COM, link axes
	double q_lower[N], q_upper[N], effort[N];
	double gravity[3];
	int jtype[N];  // enum sai2b_joint_type: 0 = revolute about, 1 = prismatic along the joint frame's z
};

constexpr int OTG_MD = N > 7 ? N : 7;  // DoFs of the largest generator (a full JointTask; the Cartesian one has 6)

// batch-uniform task parameters (sai2b_task_config flattened for the device)
struct DevTask {
	int type;
	int decoupling;
	double bie_threshold;
	double dt;
	// JointTask
	int k0;
	int full_selection; // S == I
	double S[N * N];	// rows >= k0 are zero
	double kp[N], kv[N], ki[N];
	int use_vsat;
	double vsat[N];
	// MotionForceTask
	int link;
	double frame_pos[3], frame_rot[9];
	int frame_rigid;  // frame_rot is a rotation matrix to 1e-12 (host: fill_dev_task)
	int full_projection; // P == I
	int plain_motion;	 // full task, world frame, no force/moment space, no velocity saturation
	double P[36];
	double PU[36];	// orthonormal basis of range(P) in its first `rank` columns (row-major 6x6), rest zero
	int rank; // pos_range + ori_range
	int p_lead;	 // P == diag(1 x p_lead, 0 ...) exactly (then == rank); else 6: lets the 6 x 6 eliminations stop early
	int in_frame;
	double kp_pos[3], kv_pos[3], ki_pos[3], kp_ori[3], kv_ori[3], ki_ori[3];
	double kp_f[3], kv_f[3], ki_f[3], kp_m[3], kv_m[3], ki_m[3];
	double kff_f, kff_m, max_f, max_m;
	int cl_force, cl_moment, fdim, mdim;
	int passivity;	// POPC enabled on the closed-loop force term (POPCExplicitForceControl.cpp:37-95)
	double faxis[3], maxis[3];
	double sig[4][9];  // sigmaForce, sigmaPosition, sigmaMoment, sigmaOrientation when !in_frame
	double lin_vsat, ang_vsat;
	double sensor_rot[9], sensor_pos[3];
	// SingularityHandler
	double s_min, s_max, s_abs_tol, type_1_tol, t2_ratio, t2_angle, perturb;
	int sh_cap;
	double kp1, kv1, kv2;
	int enforce_t1, enforce;
	int sv_sign;  // enum sai2b_singular_vector_sign (include/sai2b.h): classifySingularity's perturbation direction
	// internal OTG (JointTask.h:38-42, MotionForceTask.h:67-74); limits per OTG DoF (JT: task dof;
	// MFT: 3 linear then 3 angular)
	int otg_on, otg_n;
	int otg_out_is_desired;	 // full JointTask: the generator's output rows already have the goals' layout, no copy kept
	// JointTask whose range can be empty for some robots and ticks (JointTask.cpp:302-306 returns before
	// the generator is touched): a model-only pass ahead of otg_kernel writes OTG_ACTIVE per robot and
	// the generator of an inactive robot is left alone for that tick
	int otg_gated;
	double otg_vmax[OTG_MD], otg_amax[OTG_MD];
	double otg_epoch;  // bumped when the limits change: every moving robot re-plans on its next tick
	// jerk-limited generator (enableInternalOtgJerkLimited: JointTask.cpp:383-406, MotionForceTask.cpp:525-538): ruckig's
	// third-order interface (sai2b_otg3_core.hpp); its stored trajectory lives in a buffer of its own
	int otg_jerk;
	double otg_jmax[OTG_MD];
	double* otg3_traj;	// [OTG_MD * OTG3_STRIDE][B] or NULL while the task has never been jerk-limited
	// device buffers of this task
	double* goals;	// MFT [30][B]: pos3 rot9 v3 w3 a3 alpha3 f3 m3 ; JT [3*k0][B]: q dq ddq
	double* law_goals;	 // what the control law tracks: `goals`, or `otg_desired` when the OTG is on
	double* otg_desired; // same layout as goals; written by otg_kernel (MFT: rows 0..23)
	double* otg_state;	 // [OTG_ROWS][B], see the OTG_* row constants
	double* sensed; // MFT [6][B]
	double* state;	// MFT [33][B]: integ pos3 ori3 f3 m3, q_prior7, dq_prior7, t2dir7 ; JT [k0][B]
	int* istate;	// MFT [12][B]: hist words 0..6, n_types, count, size, c1, c2
	double* popc_f; // MFT, passivity only: [4][B] PO value, E_correction, sum |vcl|^2, Rc
	int* popc_i;	// [3][B] PO counter, ring head, ring size
	double* popc_q; // [POPC_RING][B] windowed power samples
	// optional introspection outputs (NULL unless debug outputs are enabled)
	double* dbg_tau;	// [7][B]
	double* dbg_N;		// [49][B] N * N_prec
	double* dbg_sigma;	// [8][B] sigma0..5, alpha, ns
	double* dbg_J;		// [42][B] JWorldFrame
	double* dbg_pose;	// [12][B] pos3 rot9
	double* dbg_F;		// [12][B] F_unit 6, F_force 6 (MotionForceTask.cpp:478-487)
};

constexpr int MFT_GOAL_ROWS = 30;
constexpr int MFT_MOTION_GOAL_ROWS = 24;  // pos rot v w a alpha: the rows the OTG replaces
// rows of otg_state (one OTG_joints / OTG_6dof_cartesian object per robot; sai2b_otg_core.hpp: Gen)
constexpr int OTG_IN = 0;	  // wrapper _input: cp cv ca tp tv, OTG_MD rows each
constexpr int OTG_CI = 5 * OTG_MD;	  // Ruckig current_input: cp cv ca tp tv
constexpr int OTG_OUT = 10 * OTG_MD;	  // _output: new position, velocity, acceleration
constexpr int OTG_TIME = 13 * OTG_MD, OTG_DURATION = OTG_TIME + 1, OTG_GOAL_REACHED = OTG_TIME + 2, OTG_RESULT = OTG_TIME + 3,
			  OTG_TARGET_SET = OTG_TIME + 4, OTG_CI_INIT = OTG_TIME + 5, OTG_CI_EPOCH = OTG_TIME + 6, OTG_CONSTRUCTED = OTG_TIME + 7;
// OTG_IN_SYNC != 0: the wrapper's input state and Ruckig's stored one both equal the output (the normal
// case after a step along the trajectory: pass_to_input twice, OTG_joints.cpp:137, ruckig.hpp:209) and
// the stored targets are equal; their rows (IN c*, CI c*, CI t*) are then not kept up to date
constexpr int OTG_TRAJ = OTG_TIME + 8;  // per DoF: brake t a p v, p0 v0, t0 t1 t2 t6, a0 a2 a6
constexpr int OTG_TRAJ_STRIDE = 13;
// rows of otg3_traj per DoF (a third-order Profile as Trajectory::at_time needs it, profile.hpp:46-50, brake.hpp:27):
// brake duration, t[2], j[2], a[2], v[2], p[2]; t_sum[7]; j[7]; a[8]; v[8]; p[8]
constexpr int OTG3_BRAKE = 0, OTG3_TSUM = 11, OTG3_J = 18, OTG3_A = 25, OTG3_V = 33, OTG3_P = 41, OTG3_STRIDE = 49;
constexpr int OTG_CART = OTG_TRAJ + OTG_MD * OTG_TRAJ_STRIDE;  // reference frame 9, goal orientation 9, goal angular velocity 3
constexpr int OTG_IN_SYNC = OTG_CART + 21;
constexpr int OTG_ACTIVE = OTG_CART + 22;  // gated JointTask only: 1 = the task has a non-empty range this tick
constexpr int OTG_ROWS = OTG_CART + 23;
static_assert(N > 7 || (OTG_TRAJ == 99 && OTG_CART == 190 && OTG_IN_SYNC == 211), "row layout of the 7-DoF builds");
// rows of a MotionForceTask's state: integrators pos 3 ori 3 force 3 moment 3, then per joint q_prior, dq_prior and the
// type-2 direction (SingularityHandler.h:218,227)
constexpr int MFT_QPRIOR = 12, MFT_DQPRIOR = 12 + N, MFT_T2DIR = 12 + 2 * N;
constexpr int MFT_STATE_ROWS = 12 + 3 * N;
constexpr int MFT_ISTATE_ROWS = 12;
constexpr int POPC_RING = 1024;	 // capacity of the PO window ring (the reference queue is unbounded)
constexpr int POPC_WINDOW = 250, POPC_MAX_COUNTER = 50;	 // POPCExplicitForceControl.h:38-39
// istate rows
constexpr int IS_NTYPES = 7, IS_COUNT = 8, IS_SIZE = 9, IS_C1 = 10, IS_C2 = 11;

struct DevParams {
	int B;
	int n_tasks;
	int gravity_comp;
	int any_bie;	  // some task uses BOUNDED_INERTIA_ESTIMATES (its shared threshold: bie_thr); set by upload_params
	const double* q;  // [7][B]
	const double* dq; // [7][B]
	double* tau;	  // [7][B]
	double* dbg_M;	  // [49][B] or NULL
	double bie_thr;
	DevModel model;
	DevTask task[SAI2B_MAX_TASKS];
};

}  // namespace sai2b
