// sai2b_host.cpp — host side of the C ABI in include/sai2b.h: configuration helpers (no GPU
// needed), context/buffer management and kernel launches. The numerical work is in
// sai2b_kernels.hip; there is no CPU compute path here — without a GPU sai2b_create() fails.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <string>
#include <vector>

#include "sai2b_launch.h"
#include "sai2b_model_host.h"
#include "sai2b_baked_panda.h"

using sai2b::DevModel;
using sai2b::rot_from_rpy;
using sai2b::sym3_from6;
using sai2b::DevParams;
using sai2b::DevTask;
constexpr int N = SAI2B_N;  // joints of the robots this build serves (sai2b_params.h)

static thread_local std::string g_error;

// defined once for the whole library by sai2b_dispatch.cpp: the error text of calls that have no context
extern "C" void sai2b_shared_error_set(const char* msg);

struct sai2b_ctx {
	int dof_tag = N;  // FIRST member: sai2b_dispatch.cpp reads it to route a call to the build for this robot size
	int B = 0, T = 0, device = 0;
	bool introspection = false;
	bool models_fresh = false;	// update_task_models() ran for the current state
	bool params_dirty = true;
	bool baked_model = false;  // the ctx model is bit-equal to the compile-time Panda constants
	bool no_fast_path = false;	// SAI2B_NO_FAST_PATH=1 in the environment: always run the generic kernel
	bool blocking_sync = false;	// SAI2B_BLOCKING_SYNC=1: sai2b_synchronize() blocks without polling first
	bool no_cert_path = false;	// SAI2B_NO_CERT_PATH=1: no SVD-free kernel for general hierarchies (sai2b_cert.hpp)
	bool no_inlane_singular = false;  // SAI2B_NO_INLANE_SINGULAR=1: singular MotionForceTasks of tick_cert_kernel<3> go to the work list
	bool prefer_cert = false;	// SAI2B_PREFER_CERT=1 (diagnostic): sai2b_cert.hpp also where sai2b_fast.hpp applies
	// lanes per robot of the generic kernel: SAI2B_GENERIC_LANES = 16 / 8 / 1 (1: the one-lane-per-robot kernel),
	// default 0 = by the amount of work (generic_lanes())
	int generic_lanes_env = 0;
	// [4]: [0..1] robots the SVD-free kernel handed to the generic one (alternating by fb_parity), [2..3] robots that went through
	// the in-lane singular branch of tick_cert_kernel (same alternation)
	int* fb_counts = nullptr;
	int* fb_list = nullptr;		// [B] their indices
	int* rg_counts = nullptr;	// [2], rg_list [B]: the same for the range pass ahead of the trajectory generators
	int* rg_list = nullptr;
	int rg_parity = 0;
	int fb_parity = 0;			// counter set of the last SVD-free launch
	// How many robots the SVD-free kernel for general hierarchies keeps is a property of the workload (a 6-DOF task
	// behind a partial JointTask is inside a blending region most of the time): every 8th such tick its count of
	// declined robots comes back to the host (pinned word, never waited for); above 40 % of the batch the next 64
	// ticks run the generic kernel alone, then the SVD-free kernel is tried again. Results are the same either way.
	int* fb_seen = nullptr;		// pinned host words: [0] declined, [1] through the in-lane singular branch
	// many robots inside a blending region of a 4- to 6-row MotionForceTask: the 6-row SVD-free kernel with the singular branch in
	// the lane runs instead of the hierarchy's usual first kernel (launch_tick); SAI2B_NO_SING6=1 switches the mode off
	bool sing_mode = false, no_sing6 = false, force_sing6 = false;
	hipEvent_t fb_seen_ev = nullptr;
	bool fb_seen_pending = false;
	int cert_probe = 0, cert_backoff = 0;
	int fb_last_seen = -1;		// length of the work list when the host last looked (-1: never)
	bool last_tick_generic_only = false;
	// sai2b_update_task_models() is deferred: the reference's loop is update -> goal setters -> computeControlTorques,
	// and the fused tick (with the SVD-free kernels in front) does both at once. Any other call in between runs the
	// pending model update first (flush_update), so nothing observable changes.
	bool update_pending = false;
	// The tasks' _current_position / _current_orientation are those of the last torque computation (or
	// re-initialisation), and enabling an OTG starts its generator there (JointTask.cpp:374-376). While
	// the state buffer still holds that state nothing is kept; the first write to it afterwards
	// (set_state, sim_step) saves q here first.
	double* q_pose = nullptr;	// [7][B]
	bool q_is_pose = true;
	int* otg_counts = nullptr;	// [2][MAX_TASKS] work-list counters of the trajectory planner (sai2b_otg.hip)
	int* otg_list = nullptr;	// [MAX_TASKS][B] robots that need the planner this tick
	int otg_parity = 0;
	// bit t set: task t's goals may have changed since the last OTG update (setters, reinitialize, config
	// updates); goals_exposed: the caller holds the device pointer of some goals buffer, so always assume it
	unsigned goals_dirty = ~0u;
	// Every generator idle (goal reached, goals untouched since): an update is a no-op for every robot, and stays one until the
	// host touches a goal or a generator's configuration — the generator kernels are not launched at all then. Known from the
	// count of non-idle robots otg_kernel leaves behind (read back asynchronously every 8th tick, never waited for) of a tick
	// launched with the goals as they still are (goals_epoch). SAI2B_NO_OTG_IDLE_SKIP=1 switches it off.
	bool otg_all_idle = false, no_otg_idle_skip = false, otg_seen_pending = false;
	unsigned long long goals_epoch = 0, otg_obs_epoch = 0;
	int* otg_busy_seen = nullptr;  // pinned host word
	hipEvent_t otg_seen_ev = nullptr;
	unsigned otg_probe = 0;
	bool goals_exposed = false;
	sai2b_robot_model model;
	sai2b_task_config cfg[SAI2B_MAX_TASKS];
	DevParams h_params;
	DevParams* d_params = nullptr;
	hipStream_t stream = nullptr;
	// device-pointer arguments (on_device != 0) are produced / consumed on the caller's stream: ordered against the
	// ctx stream with two events (sai2b_set_caller_stream; default: the legacy default stream)
	hipStream_t caller_stream = nullptr;
	hipEvent_t ev_in = nullptr, ev_out = nullptr;
	double *q = nullptr, *dq = nullptr, *tau = nullptr;
	double* status_buf = nullptr;  // [68][B] scratch of sai2b_get_mft_status
	double* sim_tau = nullptr;	// staging for host torques / bias read-back of the simulation harness
	// task-level calls (TemplateTask.h:42-88): per task the caller's N_prec, the task's N and N * N_prec of the
	// last sai2b_task_update_model, its torques and a staging copy of a host tau_prec; created on first use
	int last_call_task = 0;	  // the last launch sequence was a task-level call: 1 = task_cert_kernel + work list, 2 = the generic task kernel alone
	int* tk_count = nullptr;  // work list of the task-level SVD-free kernel (task_cert_kernel): 2 alternating counters + B robot indices
	int* tk_list = nullptr;
	int tk_parity = 0;
	bool no_task_cert = false;	// SAI2B_NO_TASK_CERT=1: the generic task_kernel for every robot (A/B)
	struct TaskIO {
		double *Nprec = nullptr, *N = nullptr, *Ntot = nullptr, *tau = nullptr, *tau_prec = nullptr;
		bool nprec_given = false;  // false: identity (the value a task is constructed with)
		bool model_fresh = false;  // sai2b_task_update_model ran for the current state
		bool standalone = false;   // the task is being driven through the task-level calls
	} tio[SAI2B_MAX_TASKS];
	std::vector<void*> allocs;
	long long launches = 0, ticks = 0;
	std::string error;
};

static int flush_update(sai2b_ctx* ctx);  // runs a deferred sai2b_update_task_models() now

static int set_error(sai2b_ctx* ctx, int code, const std::string& msg) {
	g_error = msg;
	sai2b_shared_error_set(msg.c_str());
	if (ctx) ctx->error = msg;
	return code;
}
#define HIP_TRY(ctx, expr)                                                                         \
	do {                                                                                           \
		hipError_t e_ = (expr);                                                                    \
		if (e_ != hipSuccess)                                                                      \
			return set_error(ctx, SAI2B_RUNTIME_ERROR, std::string(#expr) + ": " + hipGetErrorString(e_)); \
	} while (0)

// ------------------------------------------------------------------------------------------------
// small host-side linear algebra for the configuration helpers
// ------------------------------------------------------------------------------------------------
// cyclic Jacobi eigen-decomposition of a symmetric n x n matrix (n <= 7): A = V diag(w) V^T
static void sym_eig(int n, const double* A_in, double* w, double* V) {
	double A[49];
	std::memcpy(A, A_in, sizeof(double) * n * n);
	for (int i = 0; i < n * n; i++) V[i] = 0;
	for (int i = 0; i < n; i++) V[i * n + i] = 1;
	for (int sweep = 0; sweep < 64; sweep++) {
		double off = 0;
		for (int i = 0; i < n; i++)
			for (int j = i + 1; j < n; j++) off += A[i * n + j] * A[i * n + j];
		if (off < 1e-300) break;
		for (int p = 0; p < n; p++)
			for (int q = p + 1; q < n; q++) {
				const double apq = A[p * n + q];
				if (std::fabs(apq) < 1e-300) continue;
				const double theta = (A[q * n + q] - A[p * n + p]) / (2 * apq);
				const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1));
				const double c = 1 / std::sqrt(t * t + 1), s = t * c;
				for (int k = 0; k < n; k++) {
					const double akp = A[k * n + p], akq = A[k * n + q];
					A[k * n + p] = c * akp - s * akq;
					A[k * n + q] = s * akp + c * akq;
				}
				for (int k = 0; k < n; k++) {
					const double apk = A[p * n + k], aqk = A[q * n + k];
					A[p * n + k] = c * apk - s * aqk;
					A[q * n + k] = s * apk + c * aqk;
				}
				for (int k = 0; k < n; k++) {
					const double vkp = V[k * n + p], vkq = V[k * n + q];
					V[k * n + p] = c * vkp - s * vkq;
					V[k * n + q] = s * vkp + c * vkq;
				}
			}
	}
	for (int i = 0; i < n; i++) w[i] = A[i * n + i];
}
// Orthogonal projector onto the range of the 3 x nd matrix whose columns are the given directions,
// with the rank rule of Sai2Model::matrixRangeBasis (sigma_i / sigma_0 >= 1e-3, zero if
// sigma_0 < 1e-3). Returns the rank. (MotionForceTask.cpp:55-87,146-152)
static int direction_projector(int nd, const double* dirs, double* P) {
	for (int i = 0; i < 9; i++) P[i] = 0;
	if (nd <= 0) return 0;
	double G[9] = {0};
	for (int d = 0; d < nd; d++)
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) G[3 * i + j] += dirs[3 * d + i] * dirs[3 * d + j];
	double w[3], V[9];
	sym_eig(3, G, w, V);
	double s[3], s0 = 0;
	for (int i = 0; i < 3; i++) {
		s[i] = std::sqrt(std::max(w[i], 0.0));
		s0 = std::max(s0, s[i]);
	}
	if (s0 < 1e-3) return 0;
	int rank = 0;
	const int p = std::min(3, nd);
	// keep at most min(3, nd) directions, those above the relative tolerance
	int order[3] = {0, 1, 2};
	std::sort(order, order + 3, [&](int a, int b) { return s[a] > s[b]; });
	for (int k = 0; k < p; k++)
		if (k == 0 || s[order[k]] / s0 >= 1e-3) rank++;
	if (rank == 3) {
		P[0] = P[4] = P[8] = 1;
		return 3;
	}
	for (int k = 0; k < rank; k++) {
		const int c = order[k];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) P[3 * i + j] += V[3 * i + c] * V[3 * j + c];
	}
	return rank;
}
// rank of a k x 7 matrix by Gaussian elimination with full pivoting (Eigen::FullPivLU::rank(),
// JointTask.cpp:34-35)
static int full_pivot_rank(int rows, int cols, const double* A_in) {
	std::vector<double> A(A_in, A_in + rows * cols);
	int rank = 0;
	double maxpiv = 0;
	const int p = std::min(rows, cols);
	std::vector<int> rused(rows, 0), cused(cols, 0);
	for (int step = 0; step < p; step++) {
		int pr = -1, pc = -1;
		double best = 0;
		for (int r = 0; r < rows; r++)
			if (!rused[r])
				for (int c = 0; c < cols; c++)
					if (!cused[c] && std::fabs(A[r * cols + c]) > best) {
						best = std::fabs(A[r * cols + c]);
						pr = r;
						pc = c;
					}
		if (pr < 0) break;
		if (step == 0) maxpiv = best;
		if (best <= maxpiv * 2.220446049250313e-16 * p) break;
		rank++;
		rused[pr] = cused[pc] = 1;
		for (int r = 0; r < rows; r++) {
			if (rused[r]) continue;
			const double f = A[r * cols + pc] / A[pr * cols + pc];
			for (int c = 0; c < cols; c++) A[r * cols + c] -= f * A[pr * cols + c];
		}
	}
	return rank;
}

// ------------------------------------------------------------------------------------------------
// configuration helpers (host only)
// ------------------------------------------------------------------------------------------------

extern "C" int sai2b_model_merge_fixed_body(sai2b_robot_model* md, int link, const double xyz[3], const double rpy[3],
											double mass, const double com[3], const double inertia[6]) {
	if (!md || link < 0 || link >= N || !xyz || !rpy || !com || !inertia)
		return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_model_merge_fixed_body: bad arguments");
	sai2b::host_merge_fixed_body(md, link, xyz, rpy, mass, com, inertia);
	return SAI2B_OK;
}

extern "C" int sai2b_panda_model(sai2b_robot_model* md) {
	if (!md) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_panda_model: null");
	sai2b::host_panda_model(md);
	return SAI2B_OK;
}

static void singularity_defaults(sai2b_task_config* c) {
	c->s_min = 6e-3, c->s_max = 6e-2;  // MotionForceTask.cpp:197
	c->s_abs_tol = 1e-3;			   // SingularityHandler.cpp:11-19
	c->type_1_tol = 0.5;
	c->type_2_torque_ratio = 1e-2;
	c->type_2_angle_threshold = 5 * M_PI / 180;
	c->perturb_step_size = 5;
	c->sh_buffer_size = SAI2B_SH_HISTORY;
	c->kp_type_1 = 50, c->kv_type_1 = 14, c->kv_type_2 = 5;
	c->enforce_type_1_strategy = 0;	 // SingularityHandler.cpp:63-64
	c->enforce_handling_strategy = 1;
}

extern "C" int sai2b_default_joint_task(sai2b_task_config* c, const char* name, int task_dof, const double* selection) {
	if (!c) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_default_joint_task: null config");
	std::memset(c, 0, sizeof(*c));
	c->type = SAI2B_JOINT_TASK;
	std::snprintf(c->name, sizeof(c->name), "%s", name ? name : "joint_task");
	c->loop_timestep = 0.001;
	c->dynamic_decoupling_type = SAI2B_BOUNDED_INERTIA_ESTIMATES;
	c->bie_threshold = 0.1;
	if (!selection) {
		c->task_dof = N;
		for (int i = 0; i < N; i++) c->joint_selection[i * N + i] = 1;
	} else {
		if (task_dof < 1 || task_dof > N)
			return set_error(nullptr, SAI2B_INVALID_ARGUMENT,
							 "joint selection matrix size not consistent with robot dof in JointTask constructor\n");
		if (full_pivot_rank(task_dof, N, selection) != task_dof)
			return set_error(nullptr, SAI2B_INVALID_ARGUMENT,
							 "joint selection matrix is not full rank in JointTask constructor\n");
		c->task_dof = task_dof;
		std::memcpy(c->joint_selection, selection, sizeof(double) * task_dof * N);
	}
	for (int i = 0; i < N; i++) {
		c->kp[i] = 50.0, c->kv[i] = 14.0, c->ki[i] = 0.0;  // JointTask.h:32-34
		c->saturation_velocity[i] = M_PI / 3.0;				 // JointTask.h:44
		c->otg_max_velocity[i] = M_PI / 3.0;				 // JointTask.h:40
		c->otg_max_acceleration[i] = 2.0 * M_PI;			 // JointTask.h:41
	}
	c->use_internal_otg = 1;  // JointTask.h:38-39: on, acceleration-limited
	c->internal_otg_jerk_limited = 0;
	for (int i = 0; i < SAI2B_MAX_DOF; i++) c->otg_max_jerk[i] = 10.0 * M_PI;  // JointTask.h:42
	c->robot_dof = N;
	return SAI2B_OK;
}

extern "C" int sai2b_default_motion_force_task(sai2b_task_config* c, const char* name, int link,
											   const double frame_pos[3], const double* frame_rot, int n_trans,
											   const double* dirs_trans, int n_rot, const double* dirs_rot) {
	if (!c) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_default_motion_force_task: null config");
	std::memset(c, 0, sizeof(*c));
	c->type = SAI2B_MOTION_FORCE_TASK;
	const bool partial = !(n_trans < 0 && n_rot < 0);
	std::snprintf(c->name, sizeof(c->name), "%s", name ? name : (partial ? "partial_motion_force_task" : "motion_force_task"));
	c->loop_timestep = 0.001;
	if (link < 0 || link >= N) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "MotionForceTask: link index out of range");
	c->link = link;
	for (int i = 0; i < 3; i++) c->frame_pos[i] = frame_pos ? frame_pos[i] : 0.0;
	for (int i = 0; i < 9; i++) c->frame_rot[i] = frame_rot ? frame_rot[i] : (i % 4 == 0 ? 1.0 : 0.0);
	const char* empty_msg =
		"controlled_directions_translation and controlled_directions_rotation cannot both be empty in "
		"MotionForceTask::MotionForceTask\n";
	if (!partial) {
		for (int i = 0; i < 6; i++) c->partial_projection[i * 6 + i] = 1;
		c->pos_range = c->ori_range = 3;
	} else {
		if (std::max(n_trans, 0) + std::max(n_rot, 0) == 0) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, empty_msg);
		if ((n_trans > 0 && !dirs_trans) || (n_rot > 0 && !dirs_rot))
			return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "MotionForceTask: null direction array");
		double Pp[9], Po[9];
		c->pos_range = direction_projector(n_trans, dirs_trans, Pp);
		c->ori_range = direction_projector(n_rot, dirs_rot, Po);
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				c->partial_projection[i * 6 + j] = Pp[3 * i + j];
				c->partial_projection[(3 + i) * 6 + 3 + j] = Po[3 * i + j];
			}
		if (c->pos_range + c->ori_range == 0) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, empty_msg);
	}
	c->dynamic_decoupling_type = SAI2B_BOUNDED_INERTIA_ESTIMATES;
	c->bie_threshold = 0.1;
	for (int i = 0; i < 3; i++) {  // MotionForceTask.h:44-55
		c->kp_pos[i] = 100.0, c->kv_pos[i] = 20.0, c->ki_pos[i] = 0.0;
		c->kp_ori[i] = 200.0, c->kv_ori[i] = 28.3, c->ki_ori[i] = 0.0;
		c->kp_force[i] = 0.7, c->kv_force[i] = 10.0, c->ki_force[i] = 1.3;
		c->kp_moment[i] = 0.7, c->kv_moment[i] = 10.0, c->ki_moment[i] = 1.3;
	}
	c->kff_force = c->kff_moment = 0.95;
	c->max_force_feedback = 20.0, c->max_moment_feedback = 10.0;
	c->force_axis[2] = c->moment_axis[2] = 1.0;
	c->linear_saturation_velocity = 0.3, c->angular_saturation_velocity = M_PI / 3;
	c->sensor_rot[0] = c->sensor_rot[4] = c->sensor_rot[8] = 1.0;
	singularity_defaults(c);
	c->use_internal_otg = 1;  // MotionForceTask.h:67-72: on, acceleration-limited
	c->internal_otg_jerk_limited = 0;
	c->otg_max_linear_jerk = 10.0, c->otg_max_angular_jerk = 10.0 * M_PI;  // MotionForceTask.h:73-74
	c->otg_max_linear_velocity = 0.3, c->otg_max_linear_acceleration = 2.0;
	c->otg_max_angular_velocity = M_PI / 3, c->otg_max_angular_acceleration = 2.0 * M_PI;
	c->robot_dof = N;
	return SAI2B_OK;
}

extern "C" int sai2b_validate_tasks(const sai2b_task_config* tasks, int n_tasks, char* msg, int msg_len) {
	std::string err;
	if (!tasks || n_tasks <= 0)
		err = "RobotController must have at least one task";
	else if (n_tasks > SAI2B_MAX_TASKS)
		err = "too many tasks for this build (SAI2B_MAX_TASKS)";
	bool closed = false;
	for (int i = 0; err.empty() && i < n_tasks; i++) {
		const sai2b_task_config& t = tasks[i];
		if (t.type != SAI2B_JOINT_TASK && t.type != SAI2B_MOTION_FORCE_TASK)
			err = "task type must be JOINT_TASK or MOTION_FORCE_TASK";
		else if ((t.robot_dof ? t.robot_dof : SAI2B_DOF) != N)
			err = "task was configured for a robot with another number of joints (sai2b_task_config.robot_dof)";
		else if (t.loop_timestep != tasks[0].loop_timestep)
			err = "All tasks must have the same loop timestep in RobotController";
		for (int j = 0; err.empty() && j < i; j++)
			if (std::strncmp(t.name, tasks[j].name, sizeof(t.name)) == 0) err = "Tasks in RobotController must have unique names";
		if (err.empty() && closed)
			err = std::string("task [") + t.name +
				  "] cannot be added to the controller because it is in the nullspace of a full joint task";
		if (err.empty() && t.type == SAI2B_JOINT_TASK) {
			if (t.task_dof < 1 || t.task_dof > N)
				err = "joint selection matrix size not consistent with robot dof in JointTask constructor\n";
			else if (full_pivot_rank(t.task_dof, N, t.joint_selection) != t.task_dof)
				err = "joint selection matrix is not full rank in JointTask constructor\n";
			else if (t.task_dof == N)
				closed = true;	// isFullJointTask (RobotController.cpp:44-49)
			for (int k = 0; err.empty() && k < t.task_dof && !t.unsafe_motion_gains; k++)	// (setGainsUnsafe: JointTask.cpp:136-156)
				if (t.kp[k] < 0 || t.kv[k] < 0 || t.ki[k] < 0) err = "gains must be positive or zero in JointTask::setGains\n";
		}
		if (err.empty() && t.type == SAI2B_MOTION_FORCE_TASK) {
			if (t.link < 0 || t.link >= N) err = "MotionForceTask: link index out of range";
			if (t.force_space_dimension < 0 || t.force_space_dimension > 3)
				err = "Force space dimension should be between 0 and 3 in MotionForceTask::parametrizeForceMotionSpaces\n";
			if (t.moment_space_dimension < 0 || t.moment_space_dimension > 3)
				err = "Moment space dimension should be between 0 and 3 in MotionForceTask::parametrizeMomentRotMotionSpaces\n";
			if (t.sh_buffer_size < 1 || t.sh_buffer_size > SAI2B_SH_HISTORY) err = "singularity history size must be in [1, 200]";
			for (int k = 0; err.empty() && k < 3 && !t.unsafe_motion_gains; k++)
				if (t.kp_pos[k] < 0 || t.kv_pos[k] < 0 || t.ki_pos[k] < 0 || t.kp_ori[k] < 0 || t.kv_ori[k] < 0 || t.ki_ori[k] < 0)
					err = "all gains should be positive or zero in MotionForceTask::setPosControlGains\n";
			// the single axis of a 1- or 2-dimensional force / moment space must be a direction (MotionForceTask.cpp:840-848,868-876)
			auto axis_norm = [](const double* a) { return std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]); };
			if (err.empty() && (t.force_space_dimension == 1 || t.force_space_dimension == 2) && !(axis_norm(t.force_axis) >= 1e-2))
				err = "Force or motion axis should be a non singular vector in MotionForceTask::parametrizeForceMotionSpaces\n";
			if (err.empty() && (t.moment_space_dimension == 1 || t.moment_space_dimension == 2) && !(axis_norm(t.moment_axis) >= 1e-2))
				err = "Moment or rot motion axis should be a non singular vector in MotionForceTask::parametrizeMomentRotMotionSpaces\n";
			// pos_range / ori_range are the ranks of the two diagonal blocks of the projection (MotionForceTask.cpp:146-152)
			if (err.empty()) {
				double w[6], V[36];
				sym_eig(6, t.partial_projection, w, V);
				int rank = 0;
				bool projector = true;
				for (int k = 0; k < 6; k++) {
					if (std::fabs(w[k] - 1.0) < 1e-9)
						rank++;
					else if (std::fabs(w[k]) > 1e-9)
						projector = false;
				}
				if (!projector || rank != t.pos_range + t.ori_range || t.pos_range < 0 || t.pos_range > 3 || t.ori_range < 0 || t.ori_range > 3 ||
					rank == 0)
					err = "MotionForceTask: partial_projection must be an orthogonal projector of rank pos_range + ori_range >= 1";
			}
		}
		if (err.empty() && (t.singular_vector_sign < SAI2B_SV_SIGN_V_MAX_POSITIVE || t.singular_vector_sign > SAI2B_SV_SIGN_BOTH))
			err = "singular_vector_sign must be one of enum sai2b_singular_vector_sign";
		if (err.empty() && (t.dynamic_decoupling_type < SAI2B_FULL_DYNAMIC_DECOUPLING || t.dynamic_decoupling_type > SAI2B_IMPEDANCE))
			err = "dynamic_decoupling_type must be FULL_DYNAMIC_DECOUPLING, BOUNDED_INERTIA_ESTIMATES or IMPEDANCE";
		if (err.empty() && t.use_internal_otg && t.internal_otg_jerk_limited) {	 // setMaxJerk (OTG_joints.cpp:73-86, OTG_6dof_cartesian.cpp:126-136)
			if (t.type == SAI2B_JOINT_TASK) {
				for (int k = 0; err.empty() && k < t.task_dof; k++)
					if (!(t.otg_max_jerk[k] > 0) || std::isinf(t.otg_max_jerk[k]))
						err = "max jerk cannot be 0 or negative in any directions in OTG_joints::setMaxJerk\n";
			} else if (!(t.otg_max_linear_jerk > 0) || !(t.otg_max_angular_jerk > 0) || std::isinf(t.otg_max_linear_jerk) ||
					   std::isinf(t.otg_max_angular_jerk)) {
				err = "max jerk set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxJerk\n";
			}
		}
		if (err.empty() && t.use_internal_otg) {
			if (t.type == SAI2B_JOINT_TASK) {
				for (int k = 0; err.empty() && k < t.task_dof; k++) {
					if (!(t.otg_max_velocity[k] > 0))
						err = "max velocity cannot be 0 or negative in any directions in OTG_joints::setMaxVelocity\n";
					else if (!(t.otg_max_acceleration[k] > 0))
						err = "max acceleration cannot be 0 or negative in any directions in OTG_joints::setMaxAcceleration\n";
				}
			} else if (!(t.otg_max_linear_velocity > 0) || !(t.otg_max_angular_velocity > 0)) {
				err = "max velocity set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxLinearVelocity\n";
			} else if (!(t.otg_max_linear_acceleration > 0) || !(t.otg_max_angular_acceleration > 0)) {
				err = "max acceleration set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxLinearAcceleration\n";
			}
		}
		if (err.empty() && t.use_velocity_saturation) {
			if (t.type == SAI2B_MOTION_FORCE_TASK && (t.linear_saturation_velocity <= 0 || t.angular_saturation_velocity <= 0))
				err = "Velocity saturation values should be strictly positive or zero in MotionForceTask::enableVelocitySaturation\n";
			if (t.type == SAI2B_JOINT_TASK)
				for (int k = 0; k < t.task_dof; k++)
					if (t.saturation_velocity[k] <= 0) err = "saturation velocity must be positive in JointTask::enableVelocitySaturation\n";
		}
	}
	// implementation restriction: one shared bounded-inertia estimate per tick
	double thr = -1;
	for (int i = 0; err.empty() && i < n_tasks; i++)
		if (tasks[i].dynamic_decoupling_type == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
			if (thr >= 0 && tasks[i].bie_threshold != thr)
				err = "tasks using BOUNDED_INERTIA_ESTIMATES must share one threshold in this build";
			thr = tasks[i].bie_threshold;
		}
	if (msg && msg_len > 0) std::snprintf(msg, msg_len, "%s", err.c_str());
	if (!err.empty()) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, err);
	return SAI2B_OK;
}

// ------------------------------------------------------------------------------------------------
// context
// ------------------------------------------------------------------------------------------------
// the reference's setters keep the single axis normalised (MotionForceTask.cpp:840-848,868-876): C callers need not
static void normalise_axes(sai2b_task_config& c) {
	if (c.type != SAI2B_MOTION_FORCE_TASK) return;
	for (double* a : {c.force_axis, c.moment_axis}) {
		const double n = std::sqrt(a[0] * a[0] + a[1] * a[1] + a[2] * a[2]);
		if (n >= 1e-2)
			for (int k = 0; k < 3; k++) a[k] /= n;
	}
}

static void fill_dev_task(const sai2b_task_config& c, DevTask& d) {
	d.type = c.type;
	d.decoupling = c.dynamic_decoupling_type;
	d.bie_threshold = c.bie_threshold;
	d.dt = c.loop_timestep;
	d.k0 = c.type == SAI2B_JOINT_TASK ? c.task_dof : 0;
	for (int i = 0; i < N * N; i++) d.S[i] = 0;
	bool ident = (d.k0 == N);
	for (int i = 0; i < d.k0; i++)
		for (int j = 0; j < N; j++) {
			d.S[i * N + j] = c.joint_selection[i * N + j];
			if (d.S[i * N + j] != (i == j ? 1.0 : 0.0)) ident = false;
		}
	d.full_selection = ident ? 1 : 0;
	for (int i = 0; i < N; i++) d.kp[i] = c.kp[i], d.kv[i] = c.kv[i], d.ki[i] = c.ki[i], d.vsat[i] = c.saturation_velocity[i];
	d.use_vsat = c.use_velocity_saturation;
	d.link = c.link;
	std::memcpy(d.frame_pos, c.frame_pos, sizeof(d.frame_pos));
	std::memcpy(d.frame_rot, c.frame_rot, sizeof(d.frame_rot));
	{
		// is the compliant frame's linear part a rotation (orthonormal, det +1)? Then a pose of the control frame is its
		// position and two columns of its orientation (cert::singular_part keeps it that way); anything else the
		// reference would accept too, and it is carried as the nine numbers it is
		const double* R = c.frame_rot;
		double worst = 0;
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				double g = 0;
				for (int k = 0; k < 3; k++) g += R[3 * k + i] * R[3 * k + j];
				worst = std::fmax(worst, std::fabs(g - (i == j ? 1.0 : 0.0)));
			}
		const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
		d.frame_rigid = (worst < 1e-12 && det > 0) ? 1 : 0;
	}
	std::memcpy(d.P, c.partial_projection, sizeof(d.P));
	bool pid = true;
	for (int i = 0; i < 36; i++)
		if (c.partial_projection[i] != ((i % 7 == 0) ? 1.0 : 0.0)) pid = false;
	d.full_projection = pid ? 1 : 0;
	d.rank = c.pos_range + c.ori_range;
	{
		int lead = 0;
		bool diag01 = true;
		for (int i = 0; i < 6; i++)
			for (int j = 0; j < 6; j++) {
				const double v = c.partial_projection[i * 6 + j];
				if (i != j && v != 0.0) diag01 = false;
				if (i == j && v != 0.0 && v != 1.0) diag01 = false;
			}
		while (lead < 6 && c.partial_projection[lead * 7] == 1.0) lead++;
		for (int i = lead; i < 6; i++)
			if (c.partial_projection[i * 7] != 0.0) diag01 = false;
		d.p_lead = diag01 ? lead : 6;
	}
	{  // basis of range(P): eigenvectors of the projector with eigenvalue 1
		double w[6], V[36];
		sym_eig(6, c.partial_projection, w, V);
		for (int i = 0; i < 36; i++) d.PU[i] = 0;
		int col = 0;
		for (int j = 0; j < 6 && col < d.rank; j++)
			if (w[j] > 0.5) {
				for (int i = 0; i < 6; i++) d.PU[i * 6 + col] = V[i * 6 + j];
				col++;
			}
		if (pid)
			for (int i = 0; i < 36; i++) d.PU[i] = (i % 7 == 0) ? 1.0 : 0.0;
	}
	d.in_frame = c.parametrization_in_compliant_frame;
	for (int i = 0; i < 3; i++) {
		d.kp_pos[i] = c.kp_pos[i], d.kv_pos[i] = c.kv_pos[i], d.ki_pos[i] = c.ki_pos[i];
		d.kp_ori[i] = c.kp_ori[i], d.kv_ori[i] = c.kv_ori[i], d.ki_ori[i] = c.ki_ori[i];
		d.kp_f[i] = c.kp_force[i], d.kv_f[i] = c.kv_force[i], d.ki_f[i] = c.ki_force[i];
		d.kp_m[i] = c.kp_moment[i], d.kv_m[i] = c.kv_moment[i], d.ki_m[i] = c.ki_moment[i];
		d.faxis[i] = c.force_axis[i], d.maxis[i] = c.moment_axis[i];
		d.sensor_pos[i] = c.sensor_pos[i];
	}
	std::memcpy(d.sensor_rot, c.sensor_rot, sizeof(d.sensor_rot));
	d.kff_f = c.kff_force, d.kff_m = c.kff_moment, d.max_f = c.max_force_feedback, d.max_m = c.max_moment_feedback;
	d.cl_force = c.closed_loop_force, d.cl_moment = c.closed_loop_moment;
	d.passivity = c.passivity_enabled;
	d.fdim = c.force_space_dimension, d.mdim = c.moment_space_dimension;
	d.lin_vsat = c.linear_saturation_velocity, d.ang_vsat = c.angular_saturation_velocity;
	d.plain_motion = (d.full_projection && d.fdim == 0 && d.mdim == 0 && !d.use_vsat) ? 1 : 0;
	// MotionForceTask.cpp:892-971 with rotation = identity (world-frame parametrisation)
	for (int blk = 0; blk < 2; blk++) {
		const int dim = blk ? c.moment_space_dimension : c.force_space_dimension;
		const double* ax = blk ? c.moment_axis : c.force_axis;
		double Pb[9], A[9], T[9], sf[9], sp[9];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				Pb[3 * i + j] = c.partial_projection[(3 * blk + i) * 6 + 3 * blk + j];
				const double aa = ax[i] * ax[j], id = i == j ? 1.0 : 0.0;
				A[3 * i + j] = dim == 1 ? aa : (dim == 2 ? id - aa : (dim == 3 ? id : 0.0));
			}
		auto mul = [](const double* X, const double* Y, bool yt, double* Z) {
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += X[3 * i + k] * (yt ? Y[3 * j + k] : Y[3 * k + j]);
					Z[3 * i + j] = v;
				}
		};
		mul(Pb, A, false, T);
		mul(T, Pb, true, sf);
		if (dim == 3) std::memcpy(sf, Pb, sizeof(sf));
		for (int i = 0; i < 9; i++) A[i] = (i % 4 == 0 ? 1.0 : 0.0) - sf[i];
		mul(Pb, A, false, T);
		mul(T, Pb, true, sp);
		std::memcpy(d.sig[2 * blk], sf, sizeof(sf));
		std::memcpy(d.sig[2 * blk + 1], sp, sizeof(sp));
	}
	d.s_min = c.s_min, d.s_max = c.s_max, d.s_abs_tol = c.s_abs_tol, d.type_1_tol = c.type_1_tol;
	d.t2_ratio = c.type_2_torque_ratio, d.t2_angle = c.type_2_angle_threshold, d.perturb = c.perturb_step_size;
	d.sh_cap = c.sh_buffer_size;
	d.kp1 = c.kp_type_1, d.kv1 = c.kv_type_1, d.kv2 = c.kv_type_2;
	d.enforce_t1 = c.enforce_type_1_strategy, d.enforce = c.enforce_handling_strategy;
	d.sv_sign = c.singular_vector_sign;
	// internal OTG: one generator DoF per task dof (JT) or 3 linear + 3 angular (MFT)
	d.otg_on = c.use_internal_otg ? 1 : 0;
	d.otg_jerk = (c.use_internal_otg && c.internal_otg_jerk_limited) ? 1 : 0;
	d.otg_n = c.type == SAI2B_JOINT_TASK ? c.task_dof : 6;
	for (int i = 0; i < sai2b::OTG_MD; i++) {
		if (c.type == SAI2B_JOINT_TASK) {
			d.otg_vmax[i] = i < c.task_dof ? c.otg_max_velocity[i] : 0.0;
			d.otg_amax[i] = i < c.task_dof ? c.otg_max_acceleration[i] : 0.0;
			d.otg_jmax[i] = i < c.task_dof ? c.otg_max_jerk[i] : 0.0;
		} else {
			d.otg_vmax[i] = i < 3 ? c.otg_max_linear_velocity : i < 6 ? c.otg_max_angular_velocity : 0.0;
			d.otg_amax[i] = i < 3 ? c.otg_max_linear_acceleration : i < 6 ? c.otg_max_angular_acceleration : 0.0;
			d.otg_jmax[i] = i < 3 ? c.otg_max_linear_jerk : i < 6 ? c.otg_max_angular_jerk : 0.0;
		}
	}
}

template <class T>
static int dev_alloc(sai2b_ctx* ctx, T** p, size_t count) {
	void* v = nullptr;
	HIP_TRY(ctx, hipMalloc(&v, std::max<size_t>(count, 1) * sizeof(T)));
	HIP_TRY(ctx, hipMemsetAsync(v, 0, std::max<size_t>(count, 1) * sizeof(T), ctx->stream));
	ctx->allocs.push_back(v);
	*p = (T*)v;
	return SAI2B_OK;
}

// POPCExplicitForceControl::reInitialize (POPCExplicitForceControl.cpp:10-22) for every robot of a task;
// buffers are created on first use (1024-sample window ring per robot)
static int popc_reinit(sai2b_ctx* ctx, int task) {
	DevTask& d = ctx->h_params.task[task];
	const size_t B = ctx->B;
	int rc;
	if (!d.popc_f) {
		if ((rc = dev_alloc(ctx, &d.popc_f, 4 * B))) return rc;
		if ((rc = dev_alloc(ctx, &d.popc_i, 3 * B))) return rc;
		if ((rc = dev_alloc(ctx, &d.popc_q, (size_t)sai2b::POPC_RING * B))) return rc;
		ctx->params_dirty = true;
	}
	std::vector<double> f(4 * B, 0.0);
	std::fill(f.begin() + 3 * B, f.end(), 1.0);	 // Rc = 1
	std::vector<int> iv(3 * B, 0);
	std::fill(iv.begin(), iv.begin() + B, sai2b::POPC_MAX_COUNTER);
	HIP_TRY(ctx, hipMemcpyAsync(d.popc_f, f.data(), f.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipMemcpyAsync(d.popc_i, iv.data(), iv.size() * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}

static int upload_params(sai2b_ctx* ctx) {
	if (!ctx->params_dirty) return SAI2B_OK;
	ctx->h_params.any_bie = 0, ctx->h_params.bie_thr = 0;
	for (int t = 0; t < ctx->h_params.n_tasks; t++)
		if (ctx->h_params.task[t].decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
			ctx->h_params.any_bie = 1;
			ctx->h_params.bie_thr = ctx->h_params.task[t].bie_threshold;
		}
	// stream-ordered so that kernels already enqueued keep the parameters they were launched with
	HIP_TRY(ctx, hipMemcpyAsync(ctx->d_params, &ctx->h_params, sizeof(DevParams), hipMemcpyHostToDevice, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // h_params may be edited again right after
	ctx->params_dirty = false;
	return SAI2B_OK;
}

static int create_impl(sai2b_ctx* ctx, const sai2b_robot_model* model, const sai2b_task_config* tasks, int n_tasks,
					   int batch, int device) {
	int ndev = 0;
	if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0)
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "sai2b_create: no HIP device available (this library has no CPU path)");
	if (device < 0 || device >= ndev) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_create: bad device index");
	HIP_TRY(ctx, hipSetDevice(device));
	HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
	HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_in, hipEventDisableTiming));
	HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_out, hipEventDisableTiming));
	ctx->B = batch, ctx->T = n_tasks, ctx->device = device;
	ctx->model = *model;
	const char* nf = std::getenv("SAI2B_NO_FAST_PATH");
	ctx->no_fast_path = nf && nf[0] == '1';
	const char* bs = std::getenv("SAI2B_BLOCKING_SYNC");
	ctx->blocking_sync = bs && bs[0] == '1';
	const char* nc = std::getenv("SAI2B_NO_CERT_PATH");
	ctx->no_cert_path = nc && nc[0] == '1';
	const char* ntc = std::getenv("SAI2B_NO_TASK_CERT");
	ctx->no_task_cert = ntc && ntc[0] == '1';
	const char* nis = std::getenv("SAI2B_NO_INLANE_SINGULAR");
	ctx->no_inlane_singular = nis && nis[0] == '1';
	const char* nois = std::getenv("SAI2B_NO_OTG_IDLE_SKIP");
	ctx->no_otg_idle_skip = nois && nois[0] == '1';
	const char* ns6 = std::getenv("SAI2B_NO_SING6");
	ctx->no_sing6 = ns6 && ns6[0] == '1';
	const char* fs6 = std::getenv("SAI2B_FORCE_SING6");	 // testing aid: that kernel from the first tick on, whatever the counts
	ctx->force_sing6 = fs6 && fs6[0] == '1';
	ctx->sing_mode = ctx->force_sing6;
	const char* pc = std::getenv("SAI2B_PREFER_CERT");
	ctx->prefer_cert = pc && pc[0] == '1';
	if (const char* gl = std::getenv("SAI2B_GENERIC_LANES")) ctx->generic_lanes_env = std::atoi(gl);
	DevParams& hp = ctx->h_params;
	std::memset(&hp, 0, sizeof(hp));
	hp.B = batch, hp.n_tasks = n_tasks;
	sai2b::host_fill_dev_model(*model, hp.model);
	if constexpr (N == 7) {
		using PB = sai2b::PandaBaked;
		const DevModel& m = hp.model;
		ctx->baked_model = std::memcmp(m.E, PB::E, sizeof(m.E)) == 0 && std::memcmp(m.xyz, PB::xyz, sizeof(m.xyz)) == 0 &&
						   std::memcmp(m.mass, PB::mass, sizeof(m.mass)) == 0 && std::memcmp(m.com, PB::com, sizeof(m.com)) == 0 &&
						   std::memcmp(m.inertia, PB::inertia, sizeof(m.inertia)) == 0 &&
						   std::memcmp(m.gravity, PB::gravity, sizeof(m.gravity)) == 0 &&
						   std::memcmp(m.jtype, PB::jtype, sizeof(m.jtype)) == 0;
		if (const char* e = std::getenv("SAI2B_NO_BAKED_MODEL"))
			if (e[0] == '1') ctx->baked_model = false;
	}
	const size_t Bs = (size_t)batch;
	int rc;
	if ((rc = dev_alloc(ctx, &ctx->d_params, 1))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->q, N * Bs))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->dq, N * Bs))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->tau, N * Bs))) return rc;
	hp.q = ctx->q, hp.dq = ctx->dq, hp.tau = ctx->tau;
	if ((rc = dev_alloc(ctx, &ctx->q_pose, (size_t)N * Bs))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->fb_counts, 4))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->fb_list, Bs))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->rg_counts, 2))) return rc;
	if ((rc = dev_alloc(ctx, &ctx->rg_list, Bs))) return rc;
	HIP_TRY(ctx, hipHostMalloc((void**)&ctx->fb_seen, 2 * sizeof(int), hipHostMallocDefault));
	ctx->fb_seen[0] = ctx->fb_seen[1] = 0;
	HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->fb_seen_ev, hipEventDisableTiming));
	if ((rc = dev_alloc(ctx, &ctx->otg_counts, 2 * SAI2B_MAX_TASKS + 2))) return rc;  // (+ 2: non-idle robots of a tick, alternating)
	HIP_TRY(ctx, hipHostMalloc((void**)&ctx->otg_busy_seen, sizeof(int), hipHostMallocDefault));
	*ctx->otg_busy_seen = 1;
	HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->otg_seen_ev, hipEventDisableTiming));
	if ((rc = dev_alloc(ctx, &ctx->otg_list, SAI2B_MAX_TASKS * Bs))) return rc;
	for (int t = 0; t < n_tasks; t++) {
		ctx->cfg[t] = tasks[t];
		normalise_axes(ctx->cfg[t]);
		DevTask& d = hp.task[t];
		fill_dev_task(ctx->cfg[t], d);
		if (tasks[t].type == SAI2B_MOTION_FORCE_TASK) {
			if ((rc = dev_alloc(ctx, &d.goals, sai2b::MFT_GOAL_ROWS * Bs))) return rc;
			if ((rc = dev_alloc(ctx, &d.sensed, 6 * Bs))) return rc;
			if ((rc = dev_alloc(ctx, &d.state, sai2b::MFT_STATE_ROWS * Bs))) return rc;
			if ((rc = dev_alloc(ctx, &d.istate, sai2b::MFT_ISTATE_ROWS * Bs))) return rc;
		} else {
			if ((rc = dev_alloc(ctx, &d.goals, 3 * (size_t)d.k0 * Bs))) return rc;
			if ((rc = dev_alloc(ctx, &d.state, (size_t)d.k0 * Bs))) return rc;
		}
		// the OTG objects exist (and follow reinitialize) whether or not the OTG is enabled
		const size_t goal_rows = tasks[t].type == SAI2B_MOTION_FORCE_TASK ? (size_t)sai2b::MFT_GOAL_ROWS : 3 * (size_t)d.k0;
		if ((rc = dev_alloc(ctx, &d.otg_desired, goal_rows * Bs))) return rc;
		if ((rc = dev_alloc(ctx, &d.otg_state, (size_t)sai2b::OTG_ROWS * Bs))) return rc;
		d.otg3_traj = nullptr;
		if (d.otg_jerk && (rc = dev_alloc(ctx, &d.otg3_traj, (size_t)sai2b::OTG_MD * sai2b::OTG3_STRIDE * Bs))) return rc;
		d.otg_epoch = 0.0;
		d.otg_out_is_desired = (tasks[t].type == SAI2B_JOINT_TASK && d.k0 == N && N == sai2b::OTG_MD) ? 1 : 0;  // same row stride
		d.law_goals = d.otg_on ? (d.otg_out_is_desired ? d.otg_state + (size_t)sai2b::OTG_OUT * Bs : d.otg_desired) : d.goals;
	}
	for (int t = 0; t < n_tasks; t++)
		if (tasks[t].type == SAI2B_MOTION_FORCE_TASK && tasks[t].passivity_enabled && (rc = popc_reinit(ctx, t))) return rc;
	ctx->params_dirty = true;
	if ((rc = upload_params(ctx))) return rc;
	// the reference constructs tasks from the model's current state (q = 0 until set_state)
	if (sai2b_launch_reinit(ctx->d_params, ctx->B, -1, ctx->stream)) return set_error(ctx, SAI2B_RUNTIME_ERROR, "reinit launch failed");
	if (sai2b_launch_otg_reinit(ctx->d_params, ctx->B, -1, 0, ctx->q, ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "OTG reinit launch failed");
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}

extern "C" sai2b_ctx* sai2b_create(const sai2b_robot_model* model, const sai2b_task_config* tasks, int n_tasks, int batch,
								   int device) {
	if (!model || model->dof != N) {
		set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_create: robot model has a number of joints this library was not built for");
		return nullptr;
	}
	for (int i = 0; i < N; i++)
		if (model->joint_type[i] != SAI2B_REVOLUTE && model->joint_type[i] != SAI2B_PRISMATIC) {
			set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_create: joint_type must be SAI2B_REVOLUTE or SAI2B_PRISMATIC");
			return nullptr;
		}
	if (batch < 1) {
		set_error(nullptr, SAI2B_INVALID_ARGUMENT, "sai2b_create: batch must be >= 1");
		return nullptr;
	}
	char msg[256];
	if (sai2b_validate_tasks(tasks, n_tasks, msg, sizeof(msg))) return nullptr;
	sai2b_ctx* ctx = new sai2b_ctx();
	if (create_impl(ctx, model, tasks, n_tasks, batch, device) != SAI2B_OK) {
		std::string keep = ctx->error;
		sai2b_destroy(ctx);
		g_error = keep;
		return nullptr;
	}
	return ctx;
}

extern "C" void sai2b_destroy(sai2b_ctx* ctx) {
	if (!ctx) return;
	if (ctx->stream) (void)hipStreamSynchronize(ctx->stream);
	for (void* p : ctx->allocs) (void)hipFree(p);
	if (ctx->ev_in) (void)hipEventDestroy(ctx->ev_in);
	if (ctx->ev_out) (void)hipEventDestroy(ctx->ev_out);
	if (ctx->fb_seen_ev) (void)hipEventDestroy(ctx->fb_seen_ev);
	if (ctx->otg_seen_ev) (void)hipEventDestroy(ctx->otg_seen_ev);
	if (ctx->otg_busy_seen) (void)hipHostFree(ctx->otg_busy_seen);
	if (ctx->fb_seen) (void)hipHostFree(ctx->fb_seen);
	if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
	delete ctx;
}

extern "C" const char* sai2b_last_error(const sai2b_ctx* ctx) { return ctx ? ctx->error.c_str() : g_error.c_str(); }
extern "C" int sai2b_batch(const sai2b_ctx* ctx) { return ctx ? ctx->B : 0; }
extern "C" int sai2b_num_tasks(const sai2b_ctx* ctx) { return ctx ? ctx->T : 0; }
extern "C" int sai2b_num_joints(const sai2b_ctx* ctx) { return ctx ? N : 0; }

// `reset` of MotionForceTask::parametrizeForceMotionSpaces / parametrizeMomentRotMotionSpaces
// (MotionForceTask.cpp:838-848,866-878): the dimension changed, or, for dimension 1 or 2, the normalised
// axis is not isApprox (Eigen default 1e-12) the one in use
static bool space_changed(int dim, const double* axis, int old_dim, const double* old_axis) {
	if (dim != old_dim) return true;
	if (dim != 1 && dim != 2) return false;
	double na = 0, nb = 0, d2 = 0, a2 = 0, b2 = 0;
	for (int i = 0; i < 3; i++) na += axis[i] * axis[i], nb += old_axis[i] * old_axis[i];
	na = std::sqrt(na), nb = std::sqrt(nb);
	for (int i = 0; i < 3; i++) {
		const double a = axis[i] / na, b = old_axis[i] / nb;
		d2 += (a - b) * (a - b), a2 += a * a, b2 += b * b;
	}
	return !(d2 <= 1e-24 * std::min(a2, b2));
}

extern "C" int sai2b_update_task_config(sai2b_ctx* ctx, int task, const sai2b_task_config* cfg) {
	if (!ctx || !cfg || task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_update_task_config: bad arguments");
	if (int rc_ = flush_update(ctx)) return rc_;
	const sai2b_task_config& old = ctx->cfg[task];
	if (cfg->type != old.type || cfg->task_dof != old.task_dof || cfg->link != old.link ||
		std::memcmp(cfg->joint_selection, old.joint_selection, sizeof(old.joint_selection)) != 0 ||
		std::memcmp(cfg->partial_projection, old.partial_projection, sizeof(old.partial_projection)) != 0)
		return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_update_task_config: structural fields must not change");
	sai2b_task_config all[SAI2B_MAX_TASKS];
	for (int t = 0; t < ctx->T; t++) all[t] = ctx->cfg[t];
	all[task] = *cfg;
	char msg[256];
	if (sai2b_validate_tasks(all, ctx->T, msg, sizeof(msg))) return set_error(ctx, SAI2B_INVALID_ARGUMENT, msg);
	const bool popc_toggle = old.type == SAI2B_MOTION_FORCE_TASK && (cfg->passivity_enabled != 0) != (old.passivity_enabled != 0);
	int reparam = 0;  // flags of mft_reparam_kernel
	if (old.type == SAI2B_MOTION_FORCE_TASK) {
		const bool lin = space_changed(cfg->force_space_dimension, cfg->force_axis, old.force_space_dimension, old.force_axis);
		const bool ang = space_changed(cfg->moment_space_dimension, cfg->moment_axis, old.moment_space_dimension, old.moment_axis);
		const bool cl_f = (cfg->closed_loop_force != 0) != (old.closed_loop_force != 0);
		const bool cl_m = (cfg->closed_loop_moment != 0) != (old.closed_loop_moment != 0);
		reparam = (lin ? 1 | 4 : 0) | (ang ? 2 | 8 : 0) | (cl_f ? 4 : 0) | (cl_m ? 8 : 0);
	}
	ctx->cfg[task] = *cfg;
	normalise_axes(ctx->cfg[task]);
	DevTask& d = ctx->h_params.task[task];
	DevTask keep = d;
	fill_dev_task(ctx->cfg[task], d);
	d.popc_f = keep.popc_f, d.popc_i = keep.popc_i, d.popc_q = keep.popc_q;
	// enable(): just switches on; disable(): also reinitialises (POPCExplicitForceControl.cpp:24-28).
	// Buffers must exist before the first enabled tick.
	if (popc_toggle || (cfg->passivity_enabled && !d.popc_f)) {
		int rc2 = popc_reinit(ctx, task);
		if (rc2) return rc2;
	}
	d.goals = keep.goals, d.sensed = keep.sensed, d.state = keep.state, d.istate = keep.istate;
	d.dbg_tau = keep.dbg_tau, d.dbg_N = keep.dbg_N, d.dbg_sigma = keep.dbg_sigma, d.dbg_J = keep.dbg_J, d.dbg_pose = keep.dbg_pose;
	d.dbg_F = keep.dbg_F;
	d.otg_desired = keep.otg_desired, d.otg_state = keep.otg_state, d.otg_epoch = keep.otg_epoch;
	d.otg3_traj = keep.otg3_traj;
	if (d.otg_jerk && !d.otg3_traj) {
		int rc5 = dev_alloc(ctx, &d.otg3_traj, (size_t)sai2b::OTG_MD * sai2b::OTG3_STRIDE * (size_t)ctx->B);
		if (rc5) return rc5;
	}
	d.otg_out_is_desired = keep.otg_out_is_desired, d.otg_gated = keep.otg_gated;
	d.law_goals = d.otg_on ? (d.otg_out_is_desired ? d.otg_state + (size_t)sai2b::OTG_OUT * ctx->B : d.otg_desired) : d.goals;
	ctx->params_dirty = true;
	ctx->goals_dirty |= 1u << task;
	ctx->goals_epoch++, ctx->otg_all_idle = false;
	// enableInternalOtgAccelerationLimited (JointTask.cpp:360-381, MotionForceTask.cpp:511-523) is
	// applied when the OTG fields change: new limits make every moving robot re-plan
	// (InputParameter::operator!=, input_parameter.hpp:362-394); a generator that was off is
	// re-initialised at the current state first
	// enableInternalOtgJerkLimited (JointTask.cpp:383-406, MotionForceTask.cpp:525-538) likewise; a generator that is on and
	// changes its kind of limitation (getJerkLimitEnabled() differs) restarts at the task's current pose too, and
	// setMaxJerk, unlike disableJerkLimits, does not touch the input acceleration
	const bool mode_changed = keep.otg_on && d.otg_jerk != keep.otg_jerk;
	const bool limits_changed = std::memcmp(d.otg_vmax, keep.otg_vmax, sizeof(d.otg_vmax)) != 0 ||
								std::memcmp(d.otg_amax, keep.otg_amax, sizeof(d.otg_amax)) != 0 ||
								(d.otg_jerk && std::memcmp(d.otg_jmax, keep.otg_jmax, sizeof(d.otg_jmax)) != 0);
	if (d.otg_on && (limits_changed || mode_changed || !keep.otg_on)) {
		if (limits_changed || mode_changed) d.otg_epoch += 1.0;
		const int reinit_mode = (!keep.otg_on || mode_changed) ? 1 : (d.otg_jerk ? -1 : 2);
		int rc3 = upload_params(ctx);
		if (rc3) return rc3;
		if (reinit_mode >= 0) {
			if (sai2b_launch_otg_reinit(ctx->d_params, ctx->B, task, reinit_mode, ctx->q_is_pose ? ctx->q : ctx->q_pose, ctx->stream))
				return set_error(ctx, SAI2B_RUNTIME_ERROR, "OTG enable launch failed");
			ctx->launches++;
		}
	}
	if (reparam) {
		int rc4 = upload_params(ctx);
		if (rc4) return rc4;
		if (sai2b_launch_mft_reparam(ctx->d_params, ctx->B, task, reparam, ctx->q_is_pose ? ctx->q : ctx->q_pose, ctx->stream))
			return set_error(ctx, SAI2B_RUNTIME_ERROR, "re-parametrisation launch failed");
		ctx->launches++;
	}
	return SAI2B_OK;
}

extern "C" int sai2b_enable_gravity_compensation(sai2b_ctx* ctx, int enable) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	ctx->h_params.gravity_comp = enable ? 1 : 0;
	ctx->params_dirty = true;
	return SAI2B_OK;
}

// Ordering of device-pointer arguments against the caller's stream. before_read: work the caller enqueued on its
// stream so far (the producer of the argument) completes before what the ctx stream does next; after_read: what
// the ctx stream has been given so far (the read of the argument, or the write of a device result) completes
// before anything the caller enqueues on its stream afterwards — so the caller may overwrite an input, or
// consume a result, right after the call returns without synchronising.
static int caller_before_read(sai2b_ctx* ctx) {
	HIP_TRY(ctx, hipEventRecord(ctx->ev_in, ctx->caller_stream));
	HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->ev_in, 0));
	return SAI2B_OK;
}
static int caller_after_read(sai2b_ctx* ctx) {
	HIP_TRY(ctx, hipEventRecord(ctx->ev_out, ctx->stream));
	HIP_TRY(ctx, hipStreamWaitEvent(ctx->caller_stream, ctx->ev_out, 0));
	return SAI2B_OK;
}

static int copy_rows(sai2b_ctx* ctx, double* dst, const double* src, size_t rows, int on_device) {
	if (!src) return SAI2B_OK;
	int rc;
	if (on_device && (rc = caller_before_read(ctx))) return rc;
	HIP_TRY(ctx, hipMemcpyAsync(dst, src, rows * (size_t)ctx->B * sizeof(double),
								on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, ctx->stream));
	if (on_device) return caller_after_read(ctx);
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // pageable host memory may be reused by the caller
	return SAI2B_OK;
}

// called before the state buffers are overwritten
static int keep_pose(sai2b_ctx* ctx) {
	if (!ctx->q_is_pose) return SAI2B_OK;
	HIP_TRY(ctx, hipMemcpyAsync(ctx->q_pose, ctx->q, sizeof(double) * N * (size_t)ctx->B, hipMemcpyDeviceToDevice, ctx->stream));
	ctx->q_is_pose = false;
	return SAI2B_OK;
}

extern "C" int sai2b_set_state(sai2b_ctx* ctx, const double* q, const double* dq, int on_device) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc;
	if (q && (rc = keep_pose(ctx))) return rc;
	if ((rc = copy_rows(ctx, ctx->q, q, N, on_device))) return rc;
	if ((rc = copy_rows(ctx, ctx->dq, dq, N, on_device))) return rc;
	ctx->models_fresh = false;
	for (int t = 0; t < ctx->T; t++) ctx->tio[t].model_fresh = false;
	return SAI2B_OK;
}

static int mft_task_check(sai2b_ctx* ctx, int task, const char* fn) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (task < 0 || task >= ctx->T || ctx->cfg[task].type != SAI2B_MOTION_FORCE_TASK)
		return set_error(ctx, SAI2B_INVALID_ARGUMENT, std::string(fn) + ": task is not a MotionForceTask");
	return hipSetDevice(ctx->device) == hipSuccess ? SAI2B_OK : set_error(ctx, SAI2B_RUNTIME_ERROR, "hipSetDevice failed");
}

extern "C" int sai2b_set_mft_goals(sai2b_ctx* ctx, int task, const double* pos, const double* rot, const double* lin_vel,
								   const double* ang_vel, const double* lin_acc, const double* ang_acc, int on_device) {
	int rc = mft_task_check(ctx, task, "sai2b_set_mft_goals");
	if (rc) return rc;
	double* G = ctx->h_params.task[task].goals;
	const size_t B = ctx->B;
	ctx->goals_dirty |= 1u << task;
	ctx->goals_epoch++, ctx->otg_all_idle = false;
	if ((rc = copy_rows(ctx, G, pos, 3, on_device))) return rc;
	if ((rc = copy_rows(ctx, G + 3 * B, rot, 9, on_device))) return rc;
	if ((rc = copy_rows(ctx, G + 12 * B, lin_vel, 3, on_device))) return rc;
	if ((rc = copy_rows(ctx, G + 15 * B, ang_vel, 3, on_device))) return rc;
	if ((rc = copy_rows(ctx, G + 18 * B, lin_acc, 3, on_device))) return rc;
	return copy_rows(ctx, G + 21 * B, ang_acc, 3, on_device);
}

extern "C" int sai2b_set_mft_goal_wrench(sai2b_ctx* ctx, int task, const double* force, const double* moment, int on_device) {
	int rc = mft_task_check(ctx, task, "sai2b_set_mft_goal_wrench");
	if (rc) return rc;
	double* G = ctx->h_params.task[task].goals;
	const size_t B = ctx->B;
	if ((rc = copy_rows(ctx, G + 24 * B, force, 3, on_device))) return rc;
	return copy_rows(ctx, G + 27 * B, moment, 3, on_device);
}

extern "C" int sai2b_set_mft_sensed_wrench(sai2b_ctx* ctx, int task, const double* force, const double* moment, int on_device) {
	int rc = mft_task_check(ctx, task, "sai2b_set_mft_sensed_wrench");
	if (rc) return rc;
	double* S = ctx->h_params.task[task].sensed;
	if ((rc = copy_rows(ctx, S, force, 3, on_device))) return rc;
	return copy_rows(ctx, S + 3 * (size_t)ctx->B, moment, 3, on_device);
}

extern "C" int sai2b_set_jt_goals(sai2b_ctx* ctx, int task, const double* q_goal, const double* dq_goal, const double* ddq_goal,
								  int on_device) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (task < 0 || task >= ctx->T || ctx->cfg[task].type != SAI2B_JOINT_TASK)
		return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_set_jt_goals: task is not a JointTask");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	double* G = ctx->h_params.task[task].goals;
	const size_t B = ctx->B, k0 = ctx->cfg[task].task_dof;
	ctx->goals_dirty |= 1u << task;
	ctx->goals_epoch++, ctx->otg_all_idle = false;
	int rc;
	if ((rc = copy_rows(ctx, G, q_goal, k0, on_device))) return rc;
	if ((rc = copy_rows(ctx, G + k0 * B, dq_goal, k0, on_device))) return rc;
	return copy_rows(ctx, G + 2 * k0 * B, ddq_goal, k0, on_device);
}

extern "C" int sai2b_reinitialize(sai2b_ctx* ctx) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = upload_params(ctx);
	if (rc) return rc;
	if (sai2b_launch_reinit(ctx->d_params, ctx->B, -1, ctx->stream)) return set_error(ctx, SAI2B_RUNTIME_ERROR, "reinit launch failed");
	if (sai2b_launch_otg_reinit(ctx->d_params, ctx->B, -1, 0, ctx->q, ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "OTG reinit launch failed");
	ctx->goals_dirty = ~0u;
	ctx->goals_epoch++, ctx->otg_all_idle = false;
	ctx->launches += 2;
	ctx->models_fresh = false;
	ctx->q_is_pose = true;	// reInitializeTask reads the pose of the current state
	return SAI2B_OK;
}

// eligibility of the SVD-free path (sai2b_fast.hpp): [full MFT] or [full MFT, full JT] (any batch size: a robot
// the kernel declines goes to the generic kernel on its own, lanes past the batch just exit)
// 3 + r: the SVD-free kernel for general hierarchies (sai2b_cert.hpp): any robot size and joint type, any sequence
// of tasks whose rows can all be independent (at most N of them, a full JointTask only at the bottom); r = most
// rows a partial task brings (selects the kernel's instantiation)
static int cert_kind(const sai2b_ctx* ctx) {
	if (ctx->no_fast_path || ctx->no_cert_path || ctx->T < 1) return 0;
	int rows = 0, max_rows = 0, slots = N;  // deferred stores per robot (sai2b_cert.hpp: PEND_SLOTS = 36)
	for (int t = 0; t < ctx->T; t++) {
		const DevTask& d = ctx->h_params.task[t];
		if (d.type == SAI2B_MOTION_FORCE_TASK) {
			// the passivity observer mutates per-robot state inside the law: generic kernel only
			if (ctx->cfg[t].passivity_enabled && ctx->cfg[t].closed_loop_force) return 0;
			rows += d.rank;
			max_rows = std::max(max_rows, d.rank);
			slots += 12;
		} else {
			if (d.full_selection && t != ctx->T - 1) return 0;
			rows += d.full_selection ? 0 : d.k0;
			if (!d.full_selection) max_rows = std::max(max_rows, d.k0);
			slots += d.k0;
		}
	}
	// (the kernel is instantiated for partial tasks of up to 6 rows)
	return (rows <= N && slots <= 36 && max_rows <= 6) ? 3 + max_rows : 0;
}
static int fast_kind(const sai2b_ctx* ctx) {
	// these two kernels are written for 7 revolute joints (a 6-DOF task leaves a one-dimensional nullspace)
	bool arm7 = N == 7;
	for (int i = 0; i < N; i++)
		if (ctx->model.joint_type[i] != SAI2B_REVOLUTE) arm7 = false;
	if (!arm7 || ctx->no_fast_path || ctx->T > 2 || ctx->cfg[0].type != SAI2B_MOTION_FORCE_TASK ||
		!ctx->h_params.task[0].full_projection || ctx->h_params.task[0].rank != 6 ||
		(ctx->cfg[0].passivity_enabled && ctx->cfg[0].closed_loop_force))
		return cert_kind(ctx);
	if (ctx->prefer_cert && cert_kind(ctx)) return cert_kind(ctx);
	if (ctx->T == 1) return 1;
	return (ctx->cfg[1].type == SAI2B_JOINT_TASK && ctx->h_params.task[1].full_selection) ? 2 : cert_kind(ctx);
}

// The kernel for batches with many robots inside a blending region of a 4- to 6-row MotionForceTask: tick_cert_kernel<6, S6>
// (cert::singular_streamed); 0 when the hierarchy has no such task or is not the SVD-free kernel's
static int sing6_kind(const sai2b_ctx* ctx) {
	if (ctx->no_inlane_singular || ctx->no_sing6) return 0;
	const int ck = cert_kind(ctx);
	return ck >= 3 + 4 ? ck : 0;
}

// How many lanes a robot gets in the generic kernel (sai2b_group.hip). 16 = one DPP row per robot: the shortest
// critical path, what the (usually short) work list behind the SVD-free kernel wants; 8 = two robots per row:
// fewer idle lanes, better when the whole batch runs the generic kernel and fills the machine anyway.
static int generic_lanes(const sai2b_ctx* ctx, bool whole_batch) {
	if (ctx->generic_lanes_env == 1) return 0;
	if (ctx->generic_lanes_env == 8 || ctx->generic_lanes_env == 16) return ctx->generic_lanes_env;
	return (whole_batch && ctx->B >= 16384) ? 8 : 16;
}

// tasks whose generator is jerk-limited: their planner work lists take a launch of their own (sai2b_otg.hip)
static int jerk_mask(const sai2b_ctx* ctx) {
	int m = 0;
	for (int t = 0; t < ctx->T; t++)
		if (ctx->h_params.task[t].otg_on && ctx->h_params.task[t].otg_jerk) m |= 1 << t;
	return m;
}
static bool any_otg(const sai2b_ctx* ctx) {
	for (int t = 0; t < ctx->T; t++)
		if (ctx->h_params.task[t].otg_on) return true;
	return false;
}

// Which JointTasks need their generator gated per robot and tick: one whose range can come out empty
// (JointTask.cpp:233-239,302-306: the task then returns before setGoal / update of its OTG). Never the
// case for the first task (range of S) nor for an invertible selection behind fewer than 7 task DoF
// (S N_prec has the rank of N_prec >= 1, and a non-zero projector has a singular value >= 1).
static unsigned refresh_otg_gating(sai2b_ctx* ctx) {
	unsigned mask = 0;
	int dof_above = 0;
	for (int t = 0; t < ctx->h_params.n_tasks; t++) {
		DevTask& d = ctx->h_params.task[t];
		int gated = 0;
		if (d.type == SAI2B_JOINT_TASK) {
			gated = (d.otg_on && t > 0 && !(d.k0 == N && dof_above < N)) ? 1 : 0;
			// driven on its own behind a caller-supplied N_prec: nothing is known about its range
			if (ctx->tio[t].standalone) gated = (d.otg_on && ctx->tio[t].nprec_given) ? 1 : 0;
			dof_above += d.k0;
		} else {
			dof_above += d.rank;
		}
		if (d.otg_gated != gated) {
			d.otg_gated = gated;
			ctx->params_dirty = true;
		}
		if (gated) mask |= 1u << t;
	}
	return mask;
}

static int fetch_rows(sai2b_ctx* ctx, const double* src, size_t row0, size_t rows, double* dst);

static int launch_tick(sai2b_ctx* ctx, int commit_sh, int with_comp, int do_torque) {
	ctx->last_call_task = 0;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	for (int t = 0; t < ctx->T; t++) ctx->tio[t].standalone = ctx->tio[t].model_fresh = false;
	const unsigned gated = refresh_otg_gating(ctx);
	int rc = upload_params(ctx);
	if (rc) return rc;
	const int fast = fast_kind(ctx);
	if (do_torque && gated) {
		// which robots' gated tasks are active this tick: the task models of the current state, nothing committed.
		// Where the SVD-free kernel for general hierarchies applies (and keeps most of the batch), its cascade decides
		// for the robots it can certify and only the others take the generic kernel's range pass
		const int ck = (ctx->introspection || ctx->cert_backoff > 0) ? 0 : cert_kind(ctx);
		if (ck) {
			if (sai2b_launch_range_cert(ctx->d_params, ctx->B, ck - 3, ctx->rg_counts, ctx->rg_list, ctx->rg_parity, ctx->no_inlane_singular ? 0 : 1, ctx->stream) ||
				sai2b_launch_tick_group(ctx->d_params, ctx->B, 16, 1, 0, with_comp, 0, ctx->rg_counts + ctx->rg_parity, ctx->rg_list, ctx->stream))
				return set_error(ctx, SAI2B_RUNTIME_ERROR, "task-range pass launch failed");
			ctx->rg_parity ^= 1;
			ctx->launches += 2;
		} else {
			if (sai2b_launch_range_pass(ctx->d_params, ctx->B, ctx->introspection ? 1 : 0, with_comp, generic_lanes(ctx, true), ctx->stream))
				return set_error(ctx, SAI2B_RUNTIME_ERROR, "task-range pass launch failed");
			ctx->launches++;
		}
	}
	if (do_torque && any_otg(ctx)) {  // the generators advance once per torque computation, before the law
		if (ctx->otg_seen_pending && hipEventQuery(ctx->otg_seen_ev) == hipSuccess) {
			ctx->otg_seen_pending = false;
			if (*ctx->otg_busy_seen == 0 && ctx->otg_obs_epoch == ctx->goals_epoch) ctx->otg_all_idle = true;
		}
		// (see otg_all_idle: exact, not a heuristic — an idle generator's update neither reads nor writes anything)
		const bool idle_skip = ctx->otg_all_idle && !gated && !ctx->goals_exposed && ctx->goals_dirty == 0 && !ctx->no_otg_idle_skip;
		if (!idle_skip) {
			// (a gated task keeps reading its goals: a robot skipped while they changed must see them later)
			const int clean_mask = ctx->goals_exposed ? 0 : (int)(~ctx->goals_dirty & ~gated & ((1u << SAI2B_MAX_TASKS) - 1u));
			ctx->goals_dirty = 0;
			if (sai2b_launch_otg(ctx->d_params, ctx->B, ctx->otg_counts, ctx->otg_list, ctx->otg_parity, clean_mask, ~0, jerk_mask(ctx), ctx->stream)) return set_error(ctx, SAI2B_RUNTIME_ERROR, "OTG launch failed");
			if (!gated && !ctx->goals_exposed && !ctx->otg_seen_pending && (ctx->otg_probe++ & 7) == 0) {
				HIP_TRY(ctx, hipMemcpyAsync(ctx->otg_busy_seen, ctx->otg_counts + 2 * SAI2B_MAX_TASKS + ctx->otg_parity, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
				HIP_TRY(ctx, hipEventRecord(ctx->otg_seen_ev, ctx->stream));
				ctx->otg_seen_pending = true;
				ctx->otg_obs_epoch = ctx->goals_epoch;
			}
			ctx->otg_parity ^= 1;
			ctx->launches += 2;
		}
	}
	int fast_now = fast, cert_bits = ctx->no_inlane_singular ? 2 : 0;
	const bool fast_wanted = fast != 0 && !ctx->introspection && do_torque && commit_sh;
	if (fast_wanted) {
		// Which first kernel, by what the last look at the counters said (every 8th tick they come back through two pinned
		// words, never waited for): the hierarchy's own SVD-free kernel; while more than 20 480 robots leave it for the work list
		// and the hierarchy has a 4- to 6-row MotionForceTask, the 6-row kernel with the singular branch in the lane (until
		// fewer than 15 360 either take that branch or leave); and the generic kernel alone for 64 ticks whenever a kernel for
		// general hierarchies keeps less than 60 % of the batch. The numbers are the measured break-even on the Panda
		// (scripts/micro/sing6_crossover.py: 65 536 robots, 14 890 declined: 173 us against 215 us with the 6-row kernel;
		// 22 382 declined: 238 against 195): the in-lane 6-row branch costs every wavefront ~100 us whatever the batch, a pass
		// over the work list ~60 us per round of 4 096 robots (16 lanes each) or ~450 us per 65 536 in its throughput form.
		const int alt = sing6_kind(ctx);
		if (ctx->fb_seen_pending && hipEventQuery(ctx->fb_seen_ev) == hipSuccess) {
			ctx->fb_seen_pending = false;
			const long long d = ctx->fb_seen[0], took = ctx->fb_seen[1];
			ctx->fb_last_seen = (int)d;
			if (ctx->sing_mode) {
				if (d * 5 > (long long)ctx->B * 2)
					ctx->cert_backoff = 64;
				else if (d + took < 15360 && !ctx->force_sing6)
					ctx->sing_mode = false;
			} else if (alt && d > 20480) {
				ctx->sing_mode = true;
			} else if (fast >= 3 && d * 5 > (long long)ctx->B * 2) {
				ctx->cert_backoff = 64;
			}
		}
		if (ctx->sing_mode && alt) fast_now = alt, cert_bits |= 4;
		if (fast_now >= 3 && ctx->cert_backoff > 0) {
			ctx->cert_backoff--;
			fast_now = 0;
		}
	}
	const bool fast_launch = fast_wanted && fast_now != 0;
	ctx->last_tick_generic_only = do_torque && commit_sh && !fast_launch && !ctx->introspection;
	if (fast_launch) ctx->fb_parity ^= 1;
	// a long work list (thousands of robots) is throughput, not latency: two robots per DPP row, as for a whole batch
	const bool long_list = fast_launch && ctx->fb_last_seen > 4096;	 // (16 lanes: 4 robots x 1024 wavefronts in one round)
	if (sai2b_launch_tick(ctx->d_params, ctx->B, ctx->introspection ? 1 : 0, fast_now, ctx->baked_model ? 1 : 0, commit_sh, (with_comp ? 1 : 0) | cert_bits, do_torque, ctx->fb_counts, ctx->fb_list, ctx->fb_parity, generic_lanes(ctx, !fast_launch || long_list), ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "tick launch failed");
	ctx->launches++;
	if (fast_launch && !ctx->fb_seen_pending && (ctx->cert_probe++ & 7) == 0) {
		HIP_TRY(ctx, hipMemcpyAsync(ctx->fb_seen, ctx->fb_counts + ctx->fb_parity, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipMemcpyAsync(ctx->fb_seen + 1, ctx->fb_counts + 2 + ctx->fb_parity, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipEventRecord(ctx->fb_seen_ev, ctx->stream));
		ctx->fb_seen_pending = true;
	}
	if (do_torque) ctx->q_is_pose = true;  // computeTorques caches the tasks' current pose
	return SAI2B_OK;
}

static int fetch_tau(sai2b_ctx* ctx, double* tau, int on_device) {
	if (!tau) return SAI2B_OK;
	// a device result is written on the ctx stream: what the caller's stream still does with that buffer comes first
	if (on_device)
		if (int rc = caller_before_read(ctx)) return rc;
	HIP_TRY(ctx, hipMemcpyAsync(tau, ctx->tau, (size_t)N * ctx->B * sizeof(double),
								on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));  // a returned tau is complete, host or device
	return SAI2B_OK;
}

// The split API recomputes the (deterministic) task models inside compute_control_torques; what
// update_task_models() adds is the once-per-tick singularity bookkeeping, which is why it is
// committed there and not again by the torque pass that follows it (DESIGN.md "split API").
static int flush_update(sai2b_ctx* ctx) {
	if (!ctx || !ctx->update_pending) return SAI2B_OK;
	ctx->update_pending = false;
	int rc = launch_tick(ctx, /*commit_sh=*/1, /*with_comp=*/1, /*do_torque=*/0);
	if (rc) return rc;
	ctx->models_fresh = true;
	return SAI2B_OK;
}
extern "C" int sai2b_update_task_models(sai2b_ctx* ctx) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;  // an update still pending: it happens now, this one is the new pending one
	if (ctx->introspection) {  // the introspection outputs of the model pass are read between the two calls
		int rc = launch_tick(ctx, /*commit_sh=*/1, /*with_comp=*/1, /*do_torque=*/0);
		if (rc) return rc;
		ctx->models_fresh = true;
		return SAI2B_OK;
	}
	ctx->update_pending = true;
	return SAI2B_OK;
}

extern "C" int sai2b_compute_control_torques_ex(sai2b_ctx* ctx, double* tau, int on_device, int with_compensation) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	// a pending update is consumed here: update + torques as ONE fused tick (once-per-update singularity bookkeeping
	// committed by it, exactly as by the model pass it replaces)
	const bool fused = ctx->update_pending;
	ctx->update_pending = false;
	int rc = launch_tick(ctx, (fused || !ctx->models_fresh) ? 1 : 0, with_compensation, 1);
	if (rc) return rc;
	ctx->models_fresh = false;
	ctx->ticks += ctx->B;
	return fetch_tau(ctx, tau, on_device);
}
extern "C" int sai2b_compute_control_torques(sai2b_ctx* ctx, double* tau, int on_device) {
	return sai2b_compute_control_torques_ex(ctx, tau, on_device, 1);
}

extern "C" int sai2b_tick(sai2b_ctx* ctx, double* tau, int on_device) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	int rc = launch_tick(ctx, 1, 1, 1);
	if (rc) return rc;
	ctx->models_fresh = false;
	ctx->ticks += ctx->B;
	return fetch_tau(ctx, tau, on_device);
}

// ------------------------------------------------------------------------------------------------
// task-level plugin interface (TemplateTask.h:42-88): one task driven on its own
// ------------------------------------------------------------------------------------------------
static int task_io(sai2b_ctx* ctx, int task, const char* fn) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, std::string(fn) + ": bad task index");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (int rc_ = flush_update(ctx)) return rc_;
	sai2b_ctx::TaskIO& io = ctx->tio[task];
	if (!io.N) {
		const size_t B = ctx->B;
		int rc;
		if ((rc = dev_alloc(ctx, &io.Nprec, N * N * B))) return rc;
		if ((rc = dev_alloc(ctx, &io.N, N * N * B))) return rc;
		if ((rc = dev_alloc(ctx, &io.Ntot, N * N * B))) return rc;
		if ((rc = dev_alloc(ctx, &io.tau, N * B))) return rc;
		if ((rc = dev_alloc(ctx, &io.tau_prec, N * B))) return rc;
	}
	io.standalone = true;
	return SAI2B_OK;
}

// Rows of a task when the SVD-free task-level kernel (sai2b_cert.hip: task_cert_kernel) can serve it, else 0: the
// eligibility of cert_kind() for this one task. Not with introspection (its outputs come from the generic kernel) nor
// a passivity observer (it mutates state inside the law).
static int task_cert_rows(const sai2b_ctx* ctx, int task) {
	if (ctx->no_task_cert || ctx->no_fast_path || ctx->no_cert_path || ctx->introspection) return 0;
	const DevTask& d = ctx->h_params.task[task];
	if (d.type == SAI2B_MOTION_FORCE_TASK) {
		if (ctx->cfg[task].passivity_enabled && ctx->cfg[task].closed_loop_force) return 0;
		return (d.rank >= 1 && d.rank <= 6) ? d.rank : 0;
	}
	if (d.full_selection) return 1;	 // (a full JointTask brings no level of its own: the small instantiation)
	return d.k0 <= 6 ? d.k0 : 0;
}
// one TemplateTask call on the device: the SVD-free kernel with the generic one over the robots it declined, or the
// generic one for every robot
// the generic form of a task-level call: a robot spread over 16 lanes (a work list) or 8 (a whole batch), or — SAI2B_GENERIC_LANES=1,
// the A/B switch — the one-lane-per-robot task_kernel of rounds 1 and 2
static int launch_task_generic(sai2b_ctx* ctx, int task, const double* Np, const double* tp, double* tau_out, double* N_out, double* Ntot_out,
							   int commit_sh, int do_torque, const int* count, const int* list) {
	// (introspection: the one-lane kernel is the instantiation that fills the per-task debug outputs)
	// A whole batch stays on the one-lane kernel unless SAI2B_GENERIC_LANES asks otherwise: without the row-space chain of a
	// controller tick every JointTask behind another task takes the 7 x 7 Jacobi, and 65 536 robots through the hand-chained
	// [MFT, JT] cost 1 111 us per period with 8 lanes per robot against 692 us with one (profiles/r03_bench_task_level.txt)
	int lanes = ctx->introspection ? 0 : generic_lanes(ctx, count == nullptr);
	if (!count && ctx->generic_lanes_env == 0) lanes = 0;
	if (lanes)
		return sai2b_launch_task_group(ctx->d_params, ctx->B, lanes, task, Np, tp, tau_out, N_out, Ntot_out, commit_sh, do_torque, count, list,
									   ctx->stream);
	return sai2b_launch_task(ctx->d_params, ctx->B, task, Np, tp, tau_out, N_out, Ntot_out, commit_sh, do_torque, count, list, ctx->stream);
}

static int launch_task_call(sai2b_ctx* ctx, int task, const double* Np, const double* tp, double* tau_out, double* N_out, double* Ntot_out,
							int commit_sh, int do_torque) {
	const int rows = task_cert_rows(ctx, task);
	if (rows) {
		if (!ctx->tk_count) {
			int rc;
			if ((rc = dev_alloc(ctx, &ctx->tk_count, 2))) return rc;
			if ((rc = dev_alloc(ctx, &ctx->tk_list, (size_t)ctx->B))) return rc;
		}
		ctx->tk_parity ^= 1;
		if (sai2b_launch_task_cert(ctx->d_params, ctx->B, task, rows, Np, tp, tau_out, N_out, Ntot_out,
								   (do_torque ? 1 : 0) | (commit_sh ? 2 : 0) | (ctx->no_inlane_singular ? 4 : 0), ctx->tk_count, ctx->tk_list,
								   ctx->tk_parity, ctx->stream) ||
			launch_task_generic(ctx, task, Np, tp, tau_out, N_out, Ntot_out, commit_sh, do_torque, ctx->tk_count + ctx->tk_parity, ctx->tk_list))
			return set_error(ctx, SAI2B_RUNTIME_ERROR, "task launch failed");
		ctx->launches += 2;
		ctx->last_call_task = 1;
		return SAI2B_OK;
	}
	if (launch_task_generic(ctx, task, Np, tp, tau_out, N_out, Ntot_out, commit_sh, do_torque, nullptr, nullptr))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "task launch failed");
	ctx->last_call_task = 2;
	ctx->launches++;
	return SAI2B_OK;
}

extern "C" int sai2b_task_update_model(sai2b_ctx* ctx, int task, const double* N_prec, int on_device) {
	int rc = task_io(ctx, task, "sai2b_task_update_model");
	if (rc) return rc;
	sai2b_ctx::TaskIO& io = ctx->tio[task];
	io.nprec_given = N_prec != nullptr;
	if ((rc = copy_rows(ctx, io.Nprec, N_prec, N * N, on_device))) return rc;
	refresh_otg_gating(ctx);
	if ((rc = upload_params(ctx))) return rc;
	if ((rc = launch_task_call(ctx, task, io.nprec_given ? io.Nprec : nullptr, nullptr, nullptr, io.N, io.Ntot, /*commit_sh=*/1, /*do_torque=*/0)))
		return rc;
	io.model_fresh = true;
	return SAI2B_OK;
}

extern "C" int sai2b_task_compute_torques(sai2b_ctx* ctx, int task, const double* tau_prec, double* tau, int on_device) {
	int rc = task_io(ctx, task, "sai2b_task_compute_torques");
	if (rc) return rc;
	sai2b_ctx::TaskIO& io = ctx->tio[task];
	const unsigned gated = refresh_otg_gating(ctx);
	if ((rc = upload_params(ctx))) return rc;
	const double* Np = io.nprec_given ? io.Nprec : nullptr;
	const DevTask& d = ctx->h_params.task[task];
	if (d.otg_on) {	 // the task's generator advances once per torque computation, before the law
		if (((gated >> task) & 1) && !io.model_fresh) {
			// is the JointTask's range empty for this robot now (JointTask.cpp:302-306)? models of the current state, nothing committed
			if (launch_task_generic(ctx, task, Np, nullptr, nullptr, io.N, io.Ntot, 0, 0, nullptr, nullptr))
				return set_error(ctx, SAI2B_RUNTIME_ERROR, "task-range pass launch failed");
			ctx->launches++;
		}
		const int clean = ctx->goals_exposed ? 0 : (int)(~ctx->goals_dirty & ~gated & (1u << task));
		ctx->goals_dirty &= ~(1u << task);
		if (sai2b_launch_otg(ctx->d_params, ctx->B, ctx->otg_counts, ctx->otg_list, ctx->otg_parity, clean, 1 << task, jerk_mask(ctx), ctx->stream))
			return set_error(ctx, SAI2B_RUNTIME_ERROR, "OTG launch failed");
		ctx->otg_parity ^= 1;
		ctx->launches += 2;
	}
	const double* tp = nullptr;
	if (tau_prec && on_device) {
		tp = tau_prec;
		if ((rc = caller_before_read(ctx))) return rc;
	} else if (tau_prec) {
		if ((rc = copy_rows(ctx, io.tau_prec, tau_prec, N, 0))) return rc;
		tp = io.tau_prec;
	}
	// (behind updateTaskModel of the same state the nullspaces are in place: only the torques are produced)
	if ((rc = launch_task_call(ctx, task, Np, tp, io.tau, io.model_fresh ? nullptr : io.N, io.model_fresh ? nullptr : io.Ntot, io.model_fresh ? 0 : 1, 1)))
		return rc;
	ctx->ticks += ctx->B;
	io.model_fresh = false;
	ctx->q_is_pose = true;	// computeTorques caches the task's current pose
	// a device tau_prec was read in place: the caller may overwrite it once this call has returned
	if (!tau) return (tau_prec && on_device) ? caller_after_read(ctx) : SAI2B_OK;
	if (on_device && (rc = caller_before_read(ctx))) return rc;
	HIP_TRY(ctx, hipMemcpyAsync(tau, io.tau, (size_t)N * ctx->B * sizeof(double), on_device ? hipMemcpyDeviceToDevice : hipMemcpyDeviceToHost,
								ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}

extern "C" int sai2b_task_reinitialize(sai2b_ctx* ctx, int task) {
	if (!ctx || task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_task_reinitialize: bad arguments");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = upload_params(ctx);
	if (rc) return rc;
	if (sai2b_launch_reinit(ctx->d_params, ctx->B, task, ctx->stream)) return set_error(ctx, SAI2B_RUNTIME_ERROR, "reinit launch failed");
	if (sai2b_launch_otg_reinit(ctx->d_params, ctx->B, task, 0, ctx->q, ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "OTG reinit launch failed");
	ctx->goals_dirty |= 1u << task;
	ctx->goals_epoch++, ctx->otg_all_idle = false;
	ctx->launches += 2;
	ctx->models_fresh = false;
	ctx->tio[task].model_fresh = false;
	// reInitializeTask reads the pose of the current state; the other tasks keep theirs
	if (ctx->T == 1) ctx->q_is_pose = true;
	return SAI2B_OK;
}

extern "C" int sai2b_task_get_nullspaces(sai2b_ctx* ctx, int task, double* N_task, double* N_prec, double* N_total) {
	int rc = task_io(ctx, task, "sai2b_task_get_nullspaces");
	if (rc) return rc;
	sai2b_ctx::TaskIO& io = ctx->tio[task];
	if ((rc = fetch_rows(ctx, io.N, 0, N * N, N_task))) return rc;
	if ((rc = fetch_rows(ctx, io.Ntot, 0, N * N, N_total))) return rc;
	if (N_prec && !io.nprec_given) {  // identity, the value a task is constructed with (JointTask.cpp:62, MotionForceTask.cpp:138)
		const size_t B = ctx->B;
		for (int i = 0; i < N * N; i++) std::fill(N_prec + i * B, N_prec + (i + 1) * B, (i % (N + 1) == 0) ? 1.0 : 0.0);
		return SAI2B_OK;
	}
	return fetch_rows(ctx, io.Nprec, 0, N * N, N_prec);
}

extern "C" int sai2b_synchronize(sai2b_ctx* ctx) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	// poll for up to ~2 ms before blocking: a control loop waits for ticks of tens of microseconds, and the wake-up
	// of a blocking wait is of that order (SAI2B_BLOCKING_SYNC=1: block at once)
	if (!ctx->blocking_sync) {
		const auto t0 = std::chrono::steady_clock::now();
		for (int spin = 0;; spin++) {
			const hipError_t e = hipStreamQuery(ctx->stream);
			if (e == hipSuccess) return SAI2B_OK;
			if (e != hipErrorNotReady) break;  // let the blocking call report it
			if ((spin & 63) == 63 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) break;
		}
	}
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}
// (a deferred model update is enqueued before the stream is handed out; a caller that keeps the handle, or a
// buffer pointer, across later sai2b_update_task_models() calls flushes with sai2b_synchronize() — INTEGRATION.md §4)
extern "C" void* sai2b_stream(sai2b_ctx* ctx) {
	if (!ctx) return nullptr;
	flush_update(ctx);
	return (void*)ctx->stream;
}
extern "C" int sai2b_set_caller_stream(sai2b_ctx* ctx, void* stream) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	ctx->caller_stream = (hipStream_t)stream;
	return SAI2B_OK;
}

extern "C" void* sai2b_device_buffer(sai2b_ctx* ctx, int which, int task) {
	if (!ctx) return nullptr;
	if (flush_update(ctx)) return nullptr;
	const bool task_ok = task >= 0 && task < ctx->T;
	switch (which) {
		case SAI2B_BUF_Q: return ctx->q;
		case SAI2B_BUF_DQ: return ctx->dq;
		case SAI2B_BUF_TAU: return ctx->tau;
		case SAI2B_BUF_GOALS:
			if (task_ok) ctx->goals_exposed = true;	 // the caller may now write goals behind the library's back
			return task_ok ? ctx->h_params.task[task].goals : nullptr;
		case SAI2B_BUF_SENSED: return task_ok ? ctx->h_params.task[task].sensed : nullptr;
		case SAI2B_BUF_STATE: return task_ok ? ctx->h_params.task[task].state : nullptr;
		case SAI2B_BUF_TASK_N: return task_ok ? ctx->tio[task].N : nullptr;
		case SAI2B_BUF_TASK_N_TOTAL: return task_ok ? ctx->tio[task].Ntot : nullptr;
	}
	return nullptr;
}

extern "C" int sai2b_enable_introspection(sai2b_ctx* ctx, int enable) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (enable && !ctx->h_params.dbg_M) {
		const size_t B = ctx->B;
		int rc;
		if ((rc = dev_alloc(ctx, &ctx->h_params.dbg_M, N * N * B))) return rc;
		for (int t = 0; t < ctx->T; t++) {
			DevTask& d = ctx->h_params.task[t];
			if ((rc = dev_alloc(ctx, &d.dbg_tau, N * B))) return rc;
			if ((rc = dev_alloc(ctx, &d.dbg_N, N * N * B))) return rc;
			if (d.type == SAI2B_MOTION_FORCE_TASK) {
				if ((rc = dev_alloc(ctx, &d.dbg_sigma, 8 * B))) return rc;
				if ((rc = dev_alloc(ctx, &d.dbg_J, 6 * N * B))) return rc;
				if ((rc = dev_alloc(ctx, &d.dbg_pose, 12 * B))) return rc;
				if ((rc = dev_alloc(ctx, &d.dbg_F, 12 * B))) return rc;
			}
		}
		ctx->params_dirty = true;
	}
	ctx->introspection = enable != 0;
	return SAI2B_OK;
}

static int fetch_dbg(sai2b_ctx* ctx, double* dst, const double* src, size_t rows) {
	if (!dst) return SAI2B_OK;
	if (!ctx->introspection || !src)
		return set_error(ctx, SAI2B_INVALID_ARGUMENT, "introspection is not enabled: call sai2b_enable_introspection() before the tick");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	HIP_TRY(ctx, hipMemcpy(dst, src, rows * (size_t)ctx->B * sizeof(double), hipMemcpyDeviceToHost));
	return SAI2B_OK;
}

extern "C" int sai2b_get_task_nullspace(sai2b_ctx* ctx, int task, double* N_total) {
	if (!ctx || task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "bad task index");
	if (int rc_ = flush_update(ctx)) return rc_;
	return fetch_dbg(ctx, N_total, ctx->h_params.task[task].dbg_N, N * N);
}
extern "C" int sai2b_get_task_torques(sai2b_ctx* ctx, int task, double* tau_task) {
	if (!ctx || task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "bad task index");
	if (int rc_ = flush_update(ctx)) return rc_;
	return fetch_dbg(ctx, tau_task, ctx->h_params.task[task].dbg_tau, N);
}
extern "C" int sai2b_get_mft_singularity(sai2b_ctx* ctx, int task, double* sigma, double* alpha, double* ns_rank) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_singularity");
	if (rc) return rc;
	if ((rc = flush_update(ctx))) return rc;
	const double* s = ctx->h_params.task[task].dbg_sigma;
	const size_t B = ctx->B;
	if ((rc = fetch_dbg(ctx, sigma, s, 6))) return rc;
	if ((rc = fetch_dbg(ctx, alpha, s ? s + 6 * B : nullptr, 1))) return rc;
	return fetch_dbg(ctx, ns_rank, s ? s + 7 * B : nullptr, 1);
}
// SingularityHandler members between ticks (SingularityHandler.h:211-215): how many singular directions the
// last model update found (_singularity_types.size()) and the type-1 / type-2 counters of the history
extern "C" int sai2b_get_mft_singularity_state(sai2b_ctx* ctx, int task, int* n_singular, int* type_1_count, int* type_2_count) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_singularity_state");
	if (rc) return rc;
	if ((rc = flush_update(ctx))) return rc;
	const int* IS = ctx->h_params.task[task].istate;
	const size_t B = ctx->B;
	int* dst[3] = {n_singular, type_1_count, type_2_count};
	const int row[3] = {sai2b::IS_NTYPES, sai2b::IS_C1, sai2b::IS_C2};
	for (int k = 0; k < 3; k++)
		if (dst[k]) HIP_TRY(ctx, hipMemcpyAsync(dst[k], IS + row[k] * B, B * sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}
extern "C" int sai2b_get_mft_task_forces(sai2b_ctx* ctx, int task, double* F_unit, double* F_force) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_task_forces");
	if (rc) return rc;
	if ((rc = flush_update(ctx))) return rc;
	const double* F = ctx->h_params.task[task].dbg_F;
	if ((rc = fetch_dbg(ctx, F_unit, F, 6))) return rc;
	return fetch_dbg(ctx, F_force, F ? F + 6 * (size_t)ctx->B : nullptr, 6);
}
// rows of a device buffer to host arrays (any destination may be NULL)
static int fetch_rows(sai2b_ctx* ctx, const double* src, size_t row0, size_t rows, double* dst) {
	if (!dst) return SAI2B_OK;
	HIP_TRY(ctx, hipMemcpyAsync(dst, src + row0 * (size_t)ctx->B, rows * (size_t)ctx->B * sizeof(double), hipMemcpyDeviceToHost,
								ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}
// ---- simulation harness ----
extern "C" int sai2b_sim_step(sai2b_ctx* ctx, const double* tau, int on_device, double dt, int substeps, int with_gravity) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	if (!(dt > 0) || substeps < 1 || substeps > 1000) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_sim_step: dt must be > 0 and substeps in [1, 1000]");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = upload_params(ctx);
	if (rc) return rc;
	const double* t = ctx->tau;	 // default: the torques of the last computeControlTorques
	const bool caller_tau = tau && on_device;
	if (caller_tau) {
		t = tau;
		if ((rc = caller_before_read(ctx))) return rc;
	} else if (tau) {
		if (!ctx->sim_tau && (rc = dev_alloc(ctx, &ctx->sim_tau, (size_t)N * ctx->B))) return rc;
		if ((rc = copy_rows(ctx, ctx->sim_tau, tau, N, 0))) return rc;
		t = ctx->sim_tau;
	}
	// the kernel saves the pose the tasks cached (q_pose) on its way in, when the state buffer still holds it
	double* q_keep = ctx->q_is_pose ? ctx->q_pose : nullptr;
	ctx->q_is_pose = false;
	if (sai2b_launch_sim(ctx->d_params, ctx->B, t, dt, substeps, with_gravity, nullptr, q_keep, ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "simulation launch failed");
	ctx->launches++;
	ctx->models_fresh = false;
	for (int k = 0; k < ctx->T; k++) ctx->tio[k].model_fresh = false;
	if (caller_tau) return caller_after_read(ctx);
	return SAI2B_OK;
}
extern "C" int sai2b_get_state(sai2b_ctx* ctx, double* q, double* dq) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = fetch_rows(ctx, ctx->q, 0, N, q);
	if (rc) return rc;
	return fetch_rows(ctx, ctx->dq, 0, N, dq);
}
extern "C" int sai2b_get_bias(sai2b_ctx* ctx, int with_gravity, double* bias) {
	if (!ctx || !bias) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_get_bias: bad arguments");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = upload_params(ctx);
	if (rc) return rc;
	if (!ctx->sim_tau && (rc = dev_alloc(ctx, &ctx->sim_tau, (size_t)N * ctx->B))) return rc;
	// a zero-length step leaves the state as it is and writes the bias vector of the current state
	if (sai2b_launch_sim(ctx->d_params, ctx->B, nullptr, 0.0, 1, with_gravity, ctx->sim_tau, nullptr, ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "simulation launch failed");
	return fetch_rows(ctx, ctx->sim_tau, 0, N, bias);
}

static int run_status(sai2b_ctx* ctx, int task) {
	int rc = flush_update(ctx);	 // a deferred model update (and its launch errors) happens before anything is observed
	if (rc) return rc;
	rc = upload_params(ctx);
	if (rc) return rc;
	if (!ctx->status_buf && (rc = dev_alloc(ctx, &ctx->status_buf, 68 * (size_t)ctx->B))) return rc;
	if (sai2b_launch_mft_status(ctx->d_params, ctx->B, task, ctx->status_buf, ctx->stream))
		return set_error(ctx, SAI2B_RUNTIME_ERROR, "status launch failed");
	return SAI2B_OK;
}
extern "C" int sai2b_get_mft_velocity(sai2b_ctx* ctx, int task, double* linear_velocity, double* angular_velocity) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_velocity");
	if (rc) return rc;
	if ((rc = run_status(ctx, task))) return rc;
	if ((rc = fetch_rows(ctx, ctx->status_buf, 26, 3, linear_velocity))) return rc;
	return fetch_rows(ctx, ctx->status_buf, 29, 3, angular_velocity);
}
extern "C" int sai2b_get_mft_sigma(sai2b_ctx* ctx, int task, double* sigma_force, double* sigma_position, double* sigma_moment,
								   double* sigma_orientation) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_sigma");
	if (rc) return rc;
	if ((rc = run_status(ctx, task))) return rc;
	if ((rc = fetch_rows(ctx, ctx->status_buf, 32, 9, sigma_force))) return rc;
	if ((rc = fetch_rows(ctx, ctx->status_buf, 41, 9, sigma_position))) return rc;
	if ((rc = fetch_rows(ctx, ctx->status_buf, 50, 9, sigma_moment))) return rc;
	return fetch_rows(ctx, ctx->status_buf, 59, 9, sigma_orientation);
}
// SingularityHandler::setType1Posture (SingularityHandler.h:140-142, via MotionForceTask.h:706): _q_prior := q_des.
// Like the reference's member it holds until the handler next refreshes it (on entering a singular region, or
// while type-2 classifications dominate: SingularityHandler.cpp:233-236).
extern "C" int sai2b_set_mft_type1_posture(sai2b_ctx* ctx, int task, const double* q_des, int on_device) {
	int rc = mft_task_check(ctx, task, "sai2b_set_mft_type1_posture");
	if (rc) return rc;
	if ((rc = flush_update(ctx))) return rc;
	if (!q_des) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_set_mft_type1_posture: null posture");
	return copy_rows(ctx, ctx->h_params.task[task].state + (size_t)sai2b::MFT_QPRIOR * ctx->B, q_des, N, on_device);
}
extern "C" int sai2b_get_mft_status(sai2b_ctx* ctx, int task, double* pos, double* rot, double* sensed_force_world,
									double* sensed_moment_world, double* pos_error, double* ori_error, double* pos_error_norm,
									double* ori_error_norm) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_status");
	if (rc) return rc;
	if ((rc = run_status(ctx, task))) return rc;
	const double* S = ctx->status_buf;
	if ((rc = fetch_rows(ctx, S, 0, 3, pos))) return rc;
	if ((rc = fetch_rows(ctx, S, 3, 9, rot))) return rc;
	if ((rc = fetch_rows(ctx, S, 12, 3, sensed_force_world))) return rc;
	if ((rc = fetch_rows(ctx, S, 15, 3, sensed_moment_world))) return rc;
	if ((rc = fetch_rows(ctx, S, 18, 3, pos_error))) return rc;
	if ((rc = fetch_rows(ctx, S, 21, 3, ori_error))) return rc;
	if ((rc = fetch_rows(ctx, S, 24, 1, pos_error_norm))) return rc;
	return fetch_rows(ctx, S, 25, 1, ori_error_norm);
}
extern "C" int sai2b_get_mft_goals(sai2b_ctx* ctx, int task, double* pos, double* rot, double* lin_vel, double* ang_vel,
								   double* lin_acc, double* ang_acc, double* force, double* moment) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_goals");
	if (rc) return rc;
	const double* G = ctx->h_params.task[task].goals;
	double* dst[8] = {pos, rot, lin_vel, ang_vel, lin_acc, ang_acc, force, moment};
	const size_t row0[8] = {0, 3, 12, 15, 18, 21, 24, 27}, rows[8] = {3, 9, 3, 3, 3, 3, 3, 3};
	for (int k = 0; k < 8; k++)
		if ((rc = fetch_rows(ctx, G, row0[k], rows[k], dst[k]))) return rc;
	return SAI2B_OK;
}
extern "C" int sai2b_get_jt_goals(sai2b_ctx* ctx, int task, double* q, double* dq, double* ddq) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (task < 0 || task >= ctx->T || ctx->cfg[task].type != SAI2B_JOINT_TASK)
		return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_get_jt_goals: task is not a JointTask");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const double* G = ctx->h_params.task[task].goals;
	const size_t k0 = ctx->cfg[task].task_dof;
	int rc;
	if ((rc = fetch_rows(ctx, G, 0, k0, q))) return rc;
	if ((rc = fetch_rows(ctx, G, k0, k0, dq))) return rc;
	return fetch_rows(ctx, G, 2 * k0, k0, ddq);
}

extern "C" int sai2b_reset_integrators(sai2b_ctx* ctx, int task, int which) {
	if (!ctx || task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_reset_integrators: bad arguments");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	double* S = ctx->h_params.task[task].state;
	const size_t B = ctx->B, row = B * sizeof(double);
	if (ctx->cfg[task].type == SAI2B_JOINT_TASK) {
		HIP_TRY(ctx, hipMemsetAsync(S, 0, row * ctx->cfg[task].task_dof, ctx->stream));
		return SAI2B_OK;
	}
	// MFT state rows: integral of position 0-2, orientation 3-5, force 6-8, moment 9-11
	if (which == 0 || which == 1) {
		HIP_TRY(ctx, hipMemsetAsync(S, 0, row * 3, ctx->stream));
		HIP_TRY(ctx, hipMemsetAsync(S + 6 * B, 0, row * 3, ctx->stream));
	}
	if (which == 0 || which == 2) {
		HIP_TRY(ctx, hipMemsetAsync(S + 3 * B, 0, row * 3, ctx->stream));
		HIP_TRY(ctx, hipMemsetAsync(S + 9 * B, 0, row * 3, ctx->stream));
	}
	return SAI2B_OK;
}

extern "C" int sai2b_get_jt_desired(sai2b_ctx* ctx, int task, double* q, double* dq, double* ddq) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (task < 0 || task >= ctx->T || ctx->cfg[task].type != SAI2B_JOINT_TASK)
		return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_get_jt_desired: task is not a JointTask");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	const double* G = ctx->h_params.task[task].law_goals;
	const size_t k0 = ctx->cfg[task].task_dof;
	int rc;
	if ((rc = fetch_rows(ctx, G, 0, k0, q))) return rc;
	if ((rc = fetch_rows(ctx, G, k0, k0, dq))) return rc;
	return fetch_rows(ctx, G, 2 * k0, k0, ddq);
}
extern "C" int sai2b_get_mft_desired(sai2b_ctx* ctx, int task, double* pos, double* rot, double* lin_vel, double* ang_vel,
									 double* lin_acc, double* ang_acc) {
	int rc = mft_task_check(ctx, task, "sai2b_get_mft_desired");
	if (rc) return rc;
	const double* G = ctx->h_params.task[task].law_goals;
	if ((rc = fetch_rows(ctx, G, 0, 3, pos))) return rc;
	if ((rc = fetch_rows(ctx, G, 3, 9, rot))) return rc;
	if ((rc = fetch_rows(ctx, G, 12, 3, lin_vel))) return rc;
	if ((rc = fetch_rows(ctx, G, 15, 3, ang_vel))) return rc;
	if ((rc = fetch_rows(ctx, G, 18, 3, lin_acc))) return rc;
	return fetch_rows(ctx, G, 21, 3, ang_acc);
}
extern "C" int sai2b_get_otg_status(sai2b_ctx* ctx, int task, double* goal_reached, double* result) {
	if (!ctx || task < 0 || task >= ctx->T) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_get_otg_status: bad arguments");
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = flush_update(ctx);
	if (rc) return rc;
	const double* S = ctx->h_params.task[task].otg_state;
	if ((rc = fetch_rows(ctx, S, sai2b::OTG_GOAL_REACHED, 1, goal_reached))) return rc;
	return fetch_rows(ctx, S, sai2b::OTG_RESULT, 1, result);
}

extern "C" int sai2b_get_model(sai2b_ctx* ctx, int task, double* M, double* J, double* pos, double* rot) {
	if (!ctx) return set_error(nullptr, SAI2B_INVALID_ARGUMENT, "null ctx");
	if (int rc_ = flush_update(ctx)) return rc_;
	int rc;
	if ((rc = fetch_dbg(ctx, M, ctx->h_params.dbg_M, N * N))) return rc;
	if (!J && !pos && !rot) return SAI2B_OK;
	if ((rc = mft_task_check(ctx, task, "sai2b_get_model"))) return rc;
	const DevTask& d = ctx->h_params.task[task];
	const size_t B = ctx->B;
	if ((rc = fetch_dbg(ctx, J, d.dbg_J, 6 * N))) return rc;
	if ((rc = fetch_dbg(ctx, pos, d.dbg_pose, 3))) return rc;
	return fetch_dbg(ctx, rot, d.dbg_pose ? d.dbg_pose + 3 * B : nullptr, 9);
}

// Bench bookkeeping: run `steps` fused ticks with HIP events around EACH kernel launch of the tick on
// the ctx stream and return the average duration per launch in milliseconds: first kernel of the tick
// (the SVD-free kernel when the hierarchy is eligible, else the generic kernel) and, when there is
// one, the generic kernel over its work list behind it (0 otherwise).
extern "C" int sai2b_profile_tick(sai2b_ctx* ctx, int steps, double* first_ms, double* second_ms) {
	if (!ctx || steps < 1) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_profile_tick: bad arguments");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	int rc = upload_params(ctx);
	if (rc) return rc;
	int fast = fast_kind(ctx), cert_bits = ctx->no_inlane_singular ? 2 : 0;
	if (ctx->sing_mode && sing6_kind(ctx)) fast = sing6_kind(ctx), cert_bits |= 4;	// (the kernel the ticks are running now)
	const bool two = fast != 0 && !ctx->introspection;
	// ONE event pair around `steps` back-to-back launches (an event pair per launch costs ~4 us of its own, which made
	// the two parts add up to more than the step): first the first kernel alone, then the sequence of a tick. The
	// second figure is what the work-list pass adds to a step, so the two sum to the step by construction.
	hipEvent_t e0, e1;
	HIP_TRY(ctx, hipEventCreate(&e0));
	HIP_TRY(ctx, hipEventCreate(&e1));
	double part_ms[2] = {0, 0};
	for (int phase = 0; phase < (two ? 2 : 1); phase++) {
		for (int rep = 0; rep < 2; rep++) {	 // (the first repetition warms the sequence up)
			HIP_TRY(ctx, hipEventRecord(e0, ctx->stream));
			for (int s = 0; s < steps; s++) {
				if (two) ctx->fb_parity ^= 1;
				if (sai2b_launch_tick_part(ctx->d_params, ctx->B, ctx->introspection ? 1 : 0, fast, ctx->baked_model ? 1 : 0, 0, 1 | cert_bits, ctx->fb_counts, ctx->fb_list, ctx->fb_parity, generic_lanes(ctx, !two), ctx->stream))
					return set_error(ctx, SAI2B_RUNTIME_ERROR, "tick launch failed");
				if (phase == 1 && sai2b_launch_tick_part(ctx->d_params, ctx->B, 0, fast, ctx->baked_model ? 1 : 0, 1, 1 | cert_bits, ctx->fb_counts, ctx->fb_list, ctx->fb_parity, generic_lanes(ctx, false), ctx->stream))
					return set_error(ctx, SAI2B_RUNTIME_ERROR, "tick launch failed");
			}
			HIP_TRY(ctx, hipEventRecord(e1, ctx->stream));
			HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
			float ms = 0;
			HIP_TRY(ctx, hipEventElapsedTime(&ms, e0, e1));
			part_ms[phase] = ms / steps;
		}
	}
	(void)hipEventDestroy(e0);
	(void)hipEventDestroy(e1);
	ctx->launches += (two ? 6 : 2) * (long long)steps;
	ctx->ticks += (two ? 4 : 2) * (long long)steps * ctx->B;
	if (first_ms) *first_ms = part_ms[0];
	if (second_ms) *second_ms = two ? std::max(0.0, part_ms[1] - part_ms[0]) : 0.0;
	return SAI2B_OK;
}

extern "C" int sai2b_get_fallback_count(sai2b_ctx* ctx, int* robots) {
	if (!ctx || !robots) return set_error(ctx, SAI2B_INVALID_ARGUMENT, "sai2b_get_fallback_count: bad arguments");
	if (int rc_ = flush_update(ctx)) return rc_;
	HIP_TRY(ctx, hipSetDevice(ctx->device));
	if (ctx->last_call_task) {	// a TemplateTask call came last: its own work list
		if (ctx->last_call_task == 2) {
			*robots = ctx->B;
			return SAI2B_OK;
		}
		HIP_TRY(ctx, hipMemcpyAsync(robots, ctx->tk_count + ctx->tk_parity, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
		HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
		return SAI2B_OK;
	}
	if (ctx->last_tick_generic_only) {	// no SVD-free kernel ran in front: every robot took the generic kernel
		*robots = ctx->B;
		return SAI2B_OK;
	}
	HIP_TRY(ctx, hipMemcpyAsync(robots, ctx->fb_counts + ctx->fb_parity, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
	HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
	return SAI2B_OK;
}

extern "C" int sai2b_device_count(void) {
	int n = 0;
	return hipGetDeviceCount(&n) == hipSuccess ? n : 0;
}

extern "C" int sai2b_counters(const sai2b_ctx* ctx, long long* launches, long long* ticks) {
	if (!ctx) return SAI2B_INVALID_ARGUMENT;
	if (launches) *launches = ctx->launches;
	if (ticks) *ticks = ctx->ticks;
	return SAI2B_OK;
}
