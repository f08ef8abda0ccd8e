/*
 * sai2b.h — C ABI of the batched operational-space controller for MI355X.
 *
 * This is the drop-in boundary for the hot path
 *     RobotController::updateControllerTaskModels()   (reference src/RobotController.cpp:53-60)
 *     RobotController::computeControlTorques()        (reference src/RobotController.cpp:62-74)
 * and the task objects they drive (MotionForceTask, SingularityHandler, JointTask).
 * The reference has no C ABI (it is a static C++ library, CMakeLists.txt:70); the entry points
 * below are what a C++ adapter inside sai2-primitives would bind, and each one
 * cites the reference member function it replaces.
 *
 * Conventions
 *   - all numeric data is IEEE double
 *   - batched arrays are SoA, batch-minor:  a[c * B + b]  = component c of robot b  ("[C][B]")
 *   - matrices inside a component index are row-major (R[3*i+j], N[7*i+j], ...)
 *   - a ctx owns every device buffer; callers copy in/out with sai2b_set_ / sai2b_get_, or write
 *     device-resident data straight into the ctx buffers returned by sai2b_device_buffer()
 *   - every function returns 0 on success, nonzero on error; sai2b_last_error() has the text.
 *     Argument errors correspond to the reference's std::invalid_argument throws.
 */
#ifndef SAI2B_H_
#define SAI2B_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SAI2B_MAX_DOF 8	  /* joints of the largest supported robot; the library holds builds for 4, 6, 7 and 8 */
#define SAI2B_DOF 7		  /* the Panda's, the default of the helpers that take no robot (source compatibility) */
#define SAI2B_MAX_TASKS 4 /* tasks in one controller hierarchy */
#define SAI2B_SH_HISTORY 200 /* SingularityHandler.cpp:16 BUFFER_SIZE */

/* reference src/tasks/TemplateTask.h:19-23 */
enum sai2b_task_type {
	SAI2B_UNDEFINED = 0,
	SAI2B_JOINT_TASK = 1,
	SAI2B_MOTION_FORCE_TASK = 2
};

/* reference src/helper_modules/Sai2PrimitivesCommonDefinitions.h:9-15 */
enum sai2b_decoupling {
	SAI2B_FULL_DYNAMIC_DECOUPLING = 0,
	SAI2B_BOUNDED_INERTIA_ESTIMATES = 1,
	SAI2B_IMPEDANCE = 2
};

/* Orientation of the singular vectors the singularity classification perturbs along. The reference classifies a
 * singular direction by forward kinematics at q + 5 * V_s[:, i] (SingularityHandler.cpp:253-265) where V_s comes out
 * of Eigen::JacobiSVD (:78-81), and a singular vector is only defined up to its sign: FK(q + 5 v) and FK(q - 5 v) are
 * different poses, so "type 1" vs "type 2" — two different control strategies (:328-351) — can depend on a sign Eigen
 * does not specify and this library cannot reproduce without Eigen. The convention is therefore DEFINED here and
 * selectable per MotionForceTask; how many robots it matters for is measured in tests/test_gpu_svd_sign.py and
 * recorded in DESIGN.md §2 / INTEGRATION.md §6. Nothing else on the path depends on the sign (every other use of
 * U_s, V_s is a product v v^T or an absolute value). */
enum sai2b_singular_vector_sign {
	SAI2B_SV_SIGN_V_MAX_POSITIVE = 0, /* default: the largest-magnitude component of V_s[:, i] is positive */
	SAI2B_SV_SIGN_V_MAX_NEGATIVE = 1, /* the opposite orientation */
	SAI2B_SV_SIGN_EITHER = 2,		  /* sign-free: type 1 if the perturbation along +v OR -v moves the task */
	SAI2B_SV_SIGN_BOTH = 3			  /* sign-free: type 1 only if both do */
};

/* error codes */
enum sai2b_status {
	SAI2B_OK = 0,
	SAI2B_INVALID_ARGUMENT = 1, /* reference: std::invalid_argument */
	SAI2B_RUNTIME_ERROR = 2,	/* HIP failure */
	SAI2B_UNSUPPORTED = 3
};

/* reference: the joint types sai2-model reads from a URDF that the examples use (revolute / continuous,
 * prismatic: examples/06-partial_joint_task/panda_arm_sliding_base.urdf) */
enum sai2b_joint_type {
	SAI2B_REVOLUTE = 0,
	SAI2B_PRISMATIC = 1
};

/*
 * Rigid-body model of a fixed-base serial chain with `dof` joints (4, 6, 7 or 8), each moving about (revolute)
 * or along (prismatic) the z axis of its joint frame (the subset of sai2-model the path needs: reference call
 * sites SURVEY §8(c)). A URDF joint with another <axis> is brought to this form by rotating its joint frame and
 * re-expressing what hangs on it, which sai2b_model_from_urdf() does.
 * Joint i connects link i-1 (parent) to link i; link -1 is the world/base.
 * Fixed children (e.g. the Panda "end-effector" body, panda_arm.urdf:105-116,179-183) must be merged
 * into their parent's inertial parameters with sai2b_model_merge_fixed_body().
 * Arrays are sized for SAI2B_MAX_DOF; entries >= dof are ignored.
 */
typedef struct sai2b_robot_model {
	int dof;
	double joint_xyz[SAI2B_MAX_DOF][3];				   /* URDF <origin xyz>, in parent link frame */
	double joint_rpy[SAI2B_MAX_DOF][3];				   /* URDF <origin rpy>  (R = Rz(y) Ry(p) Rx(r)) */
	double link_mass[SAI2B_MAX_DOF];				   /* child link of joint i */
	double link_com[SAI2B_MAX_DOF][3];				   /* COM in link frame */
	double link_inertia[SAI2B_MAX_DOF][6];			   /* ixx iyy izz ixy ixz iyz at the COM, link axes */
	double q_lower[SAI2B_MAX_DOF], q_upper[SAI2B_MAX_DOF]; /* joint limits (SingularityHandler.cpp:43-51) */
	double effort[SAI2B_MAX_DOF];
	double gravity[3]; /* world gravity used by jointGravityVector (RobotController.cpp:71) */
	int joint_type[SAI2B_MAX_DOF]; /* enum sai2b_joint_type */
} sai2b_robot_model;

/*
 * One task of the hierarchy. Field defaults are filled by sai2b_default_joint_task() /
 * sai2b_default_motion_force_task() and mirror JointTask.h:31-45, MotionForceTask.h:40-75,
 * MotionForceTask.cpp:197 and SingularityHandler.cpp:10-20. Everything here is batch-uniform.
 */
typedef struct sai2b_task_config {
	int type; /* enum sai2b_task_type */
	char name[64];
	double loop_timestep;
	int dynamic_decoupling_type; /* enum sai2b_decoupling */
	double bie_threshold;

	/* ---- JointTask (JointTask.cpp:14-89) ---- */
	int task_dof;								  /* rows of the selection matrix (the robot's dof if full) */
	double joint_selection[SAI2B_MAX_DOF * SAI2B_MAX_DOF]; /* row-major task_dof x robot_dof, packed (row stride = robot_dof) */
	double kp[SAI2B_MAX_DOF], kv[SAI2B_MAX_DOF], ki[SAI2B_MAX_DOF];
	int use_velocity_saturation; /* shared flag name for both task types */
	double saturation_velocity[SAI2B_MAX_DOF];

	/* ---- MotionForceTask (MotionForceTask.cpp:16-202) ---- */
	int link;						/* 0-based moving link the compliant frame is attached to */
	double frame_pos[3];			/* compliant frame in link frame: translation */
	double frame_rot[9];			/*                               rotation (row-major) */
	double partial_projection[36];	/* _partial_task_projection, blkdiag(P_pos, P_ori) */
	int pos_range, ori_range;		/* ranks of the two 3x3 blocks (MotionForceTask.cpp:151-152) */
	int parametrization_in_compliant_frame;
	double kp_pos[3], kv_pos[3], ki_pos[3];
	double kp_ori[3], kv_ori[3], ki_ori[3];
	double kp_force[3], kv_force[3], ki_force[3];
	double kp_moment[3], kv_moment[3], ki_moment[3];
	double kff_force, kff_moment;
	double max_force_feedback, max_moment_feedback;
	int closed_loop_force, closed_loop_moment;
	int passivity_enabled; /* MotionForceTask::enablePassivity (MotionForceTask.h:629): POPC on the force loop */
	int force_space_dimension, moment_space_dimension;
	double force_axis[3], moment_axis[3];
	double linear_saturation_velocity, angular_saturation_velocity;
	double sensor_rot[9], sensor_pos[3]; /* _T_control_to_sensor (MotionForceTask.cpp:793-803) */

	/* ---- SingularityHandler (SingularityHandler.cpp:10-73, MotionForceTask.cpp:197) ---- */
	double s_min, s_max, s_abs_tol;
	double type_1_tol, type_2_torque_ratio, type_2_angle_threshold, perturb_step_size;
	int sh_buffer_size;
	double kp_type_1, kv_type_1, kv_type_2;
	int enforce_type_1_strategy, enforce_handling_strategy;
	int singular_vector_sign; /* enum sai2b_singular_vector_sign: classifySingularity's perturbation direction */

	/* ---- internal online trajectory generation (JointTask.h:38-42,294-324;
	 * MotionForceTask.h:67-74,387-427): on by default, acceleration-limited. The desired state fed
	 * to the control law is the OTG's next state instead of the goal. ---- */
	int use_internal_otg;		   /* enableInternalOtgAccelerationLimited / ...JerkLimited / disableInternalOtg */
	int internal_otg_jerk_limited; /* enableInternalOtgJerkLimited (JointTask.h:295-310, MotionForceTask.h:416): ruckig's
									* third-order interface with the otg_max_*jerk limits below; switching between the two
									* modes re-initialises the generator at the task's current pose, as the reference does
									* (JointTask.cpp:374,399; MotionForceTask.cpp:514,529) */
	double otg_max_velocity[SAI2B_MAX_DOF], otg_max_acceleration[SAI2B_MAX_DOF]; /* JointTask, per task dof */
	double otg_max_linear_velocity, otg_max_linear_acceleration;		 /* MotionForceTask */
	double otg_max_angular_velocity, otg_max_angular_acceleration;
	double otg_max_jerk[SAI2B_MAX_DOF];							 /* JointTask, per task dof (JointTask.h:42: 10 pi) */
	double otg_max_linear_jerk, otg_max_angular_jerk;			 /* MotionForceTask (MotionForceTask.h:73-74: 10, 10 pi) */

	/* MotionForceTask::setPosControlGainsUnsafe / setOriControlGainsUnsafe (MotionForceTask.h:283,304;
	 * MotionForceTask.cpp:630-649) and JointTask::setGainsUnsafe (JointTask.h:256, JointTask.cpp:136-156): nonzero
	 * skips the sign check of the motion gains */
	int unsafe_motion_gains;

	/* joints of the robot the task is for (filled by the sai2b_default_* helpers; 0 is read as SAI2B_DOF) */
	int robot_dof;
} sai2b_task_config;

typedef struct sai2b_ctx sai2b_ctx;

/* ------------------------------------------------------------------ model / config helpers
 * (host-only, no GPU needed) */

/* Panda arm constants (examples/15-haptic_control_impedance_type/panda_arm.urdf:4-184), with the
 * fixed "end-effector" body merged into link 7 (dof = 7). */
int sai2b_panda_model(sai2b_robot_model* model);

/* Merge a fixed child body into link `link` (what RBDL does for URDF fixed joints). */
int sai2b_model_merge_fixed_body(sai2b_robot_model* model, int link, const double xyz[3],
								 const double rpy[3], double mass, const double com[3],
								 const double inertia[6]);

/* URDF ingestion (host-only). The reference loads robots from URDF through sai2-model (e.g.
 * examples/05-using_robot_controller/05-using_robot_controller.cpp:45-47 with panda_arm.urdf): this
 * reads the same files into a sai2b_robot_model. Scope: one serial chain of 4, 6, 7 or 8 revolute / continuous /
 * prismatic joints with any <axis> (panda_arm.urdf; examples/06-partial_joint_task/panda_arm_sliding_base.urdf;
 * examples/11-planar_robot_controller/rrrrbot.urdf), any fixed joints (the bodies behind them are
 * merged into the link they hang on, as RBDL does), rotated <inertial> frames. `urdf` is a file name
 * (is_file != 0) or the XML text. `links` (may be NULL) receives, for every URDF link, the moving
 * link it is rigidly attached to (-1: the world) and its fixed pose there. */
#define SAI2B_URDF_MAX_LINKS 32
typedef struct sai2b_urdf_links {
	int n_links;
	char name[SAI2B_URDF_MAX_LINKS][64];
	int moving_link[SAI2B_URDF_MAX_LINKS];
	double pos[SAI2B_URDF_MAX_LINKS][3];
	double rot[SAI2B_URDF_MAX_LINKS][9]; /* row-major */
} sai2b_urdf_links;
int sai2b_model_from_urdf(const char* urdf, int is_file, sai2b_robot_model* model,
						  sai2b_urdf_links* links);
/* Sai2Model::setTRobotBase (examples/05-using_robot_controller/05-using_robot_controller.cpp:69): the pose of the
 * robot's base in the world. The reference's tasks work in the WORLD frame (MotionForceTask.cpp:100-103,262,
 * 286-289: positionInWorld / rotationInWorld / JWorldFrame) and the model's gravity is a world vector, so the
 * transform is folded into the first joint's origin: link -1 of the model is the world afterwards, goals, poses,
 * forces and Jacobians of a MotionForceTask are world quantities, jointGravityVector sees the rotated base.
 * Host-only; call it on the model BEFORE sai2b_create() (a context copies the model); a second call composes on
 * top of the first. rot: row-major rotation matrix, NULL = identity. */
int sai2b_model_set_base_transform(sai2b_robot_model* model, const double pos[3], const double* rot);
/* MotionForceTask takes a link NAME and a compliant frame in that link (MotionForceTask.h:96-101,
 * e.g. "end-effector", a body on a fixed joint of link7): resolve them to the moving link index and
 * the frame in it that sai2b_default_motion_force_task() takes. rot_in_link / frame_rot may be NULL. */
int sai2b_urdf_resolve_frame(const sai2b_urdf_links* links, const char* link_name,
							 const double pos_in_link[3], const double* rot_in_link, int* moving_link,
							 double frame_pos[3], double frame_rot[9]);

/* JointTask::JointTask + initialSetup defaults (JointTask.cpp:14-89, JointTask.h:31-45).
 * selection == NULL -> full joint task; else row-major task_dof x 7, must be full row rank
 * (JointTask.cpp:34-39). */
int sai2b_default_joint_task(sai2b_task_config* cfg, const char* name, int task_dof,
							 const double* selection);
/* the same for a robot with `robot_dof` joints (selection: row-major task_dof x robot_dof) */
int sai2b_default_joint_task_dof(sai2b_task_config* cfg, const char* name, int robot_dof, int task_dof,
								 const double* selection);

/* MotionForceTask::MotionForceTask + initialSetup defaults (MotionForceTask.cpp:16-202).
 * n_trans/n_rot < 0 -> full 6-DOF task (first ctor); otherwise the partial-task ctor with the given
 * controlled directions (row-major n x 3). frame_rot may be NULL (identity). */
int sai2b_default_motion_force_task(sai2b_task_config* cfg, const char* name, int link,
									const double frame_pos[3], const double* frame_rot,
									int n_trans, const double* dirs_trans, int n_rot,
									const double* dirs_rot);
/* the same for a robot with `robot_dof` joints (link < robot_dof) */
int sai2b_default_motion_force_task_dof(sai2b_task_config* cfg, const char* name, int robot_dof, int link,
										const double frame_pos[3], const double* frame_rot,
										int n_trans, const double* dirs_trans, int n_rot,
										const double* dirs_rot);

/* RobotController ctor checks (RobotController.cpp:8-51): at least one task, same loop timestep,
 * unique names, nothing after a full joint task. Returns SAI2B_INVALID_ARGUMENT with `msg` filled. */
int sai2b_validate_tasks(const sai2b_task_config* tasks, int n_tasks, char* msg, int msg_len);

/* ------------------------------------------------------------------ controller (needs a GPU) */

/* RobotController::RobotController (RobotController.cpp:8-51) for `batch` robot instances on HIP
 * device `device`. Validates like the reference; allocates all device buffers; goals are
 * initialised as reInitializeTask() does once a state has been set. Returns NULL on failure
 * (see sai2b_last_error(NULL)). */
sai2b_ctx* sai2b_create(const sai2b_robot_model* model, const sai2b_task_config* tasks, int n_tasks,
						int batch, int device);
void sai2b_destroy(sai2b_ctx* ctx);
const char* sai2b_last_error(const sai2b_ctx* ctx);

int sai2b_batch(const sai2b_ctx* ctx);
int sai2b_num_tasks(const sai2b_ctx* ctx);
int sai2b_num_joints(const sai2b_ctx* ctx); /* joints of the context's robot */

/* Re-configure batch-uniform task parameters (gains, decoupling, force-space parametrisation,
 * flags) after creation — the reference's setters (MotionForceTask.h:272-328,576-623,669-753,
 * JointTask.h:234-259,360-384). Structural fields (type, task_dof, selection, link, projection)
 * must not change. The setters' side effects follow the fields that changed:
 *  - use_internal_otg / otg_max_*: enableInternalOtgAccelerationLimited (JointTask.cpp:360-381,
 *    MotionForceTask.cpp:511-523): a generator that was off starts at the task's current pose;
 *  - force_space_dimension / force_axis (axis compared normalised, for dimension 1 or 2):
 *    parametrizeForceMotionSpaces (MotionForceTask.cpp:830-858): goal position := current position, goal
 *    linear velocity / acceleration := 0, the linear half of the generator re-initialised there, position
 *    and force integrators reset; moment_space_dimension / moment_axis: the angular counterpart (:860-890);
 *  - closed_loop_force / closed_loop_moment: the corresponding integrators reset (:973-986);
 *  - passivity_enabled: the observer re-initialised (POPCExplicitForceControl.cpp:24-28).
 * "Current" pose = the task's cached one: that of the last torque computation or re-initialisation. */
int sai2b_update_task_config(sai2b_ctx* ctx, int task, const sai2b_task_config* cfg);

/* RobotController::enableGravityCompensation (RobotController.h:31-33) */
int sai2b_enable_gravity_compensation(sai2b_ctx* ctx, int enable);

/* Sai2Model::setQ / setDq + updateModel (examples/05-using_robot_controller.cpp:143-145).
 * q, dq: [dof][B] (here and below "7" in a shape stands for the robot's dof). on_device != 0 -> the pointers are device memory. */
int sai2b_set_state(sai2b_ctx* ctx, const double* q, const double* dq, int on_device);

/* MotionForceTask::setGoalPosition/Orientation/LinearVelocity/AngularVelocity/
 * LinearAcceleration/AngularAcceleration (MotionForceTask.h:211-247). Any pointer may be NULL
 * (left unchanged). pos,v,w,a,alpha: [3][B]; rot: [9][B] row-major. */
int sai2b_set_mft_goals(sai2b_ctx* ctx, int task, const double* pos, const double* rot,
						const double* lin_vel, const double* ang_vel, const double* lin_acc,
						const double* ang_acc, int on_device);
/* MotionForceTask::setGoalForce / setGoalMoment (MotionForceTask.h:590-623): [3][B] each */
int sai2b_set_mft_goal_wrench(sai2b_ctx* ctx, int task, const double* force, const double* moment,
							  int on_device);
/* MotionForceTask::updateSensedForceAndMoment (MotionForceTask.cpp:805-828): sensor-frame values,
 * [3][B] each; resolved to the world frame with the current state when the tick runs. */
int sai2b_set_mft_sensed_wrench(sai2b_ctx* ctx, int task, const double* force, const double* moment,
								int on_device);
/* JointTask::setGoalPosition/Velocity/Acceleration (JointTask.h:137-179): [task_dof][B] each */
int sai2b_set_jt_goals(sai2b_ctx* ctx, int task, const double* q_goal, const double* dq_goal,
					   const double* ddq_goal, int on_device);

/* RobotController::reinitializeTasks (RobotController.cpp:76-80): goals <- current state,
 * integrators and singularity history cleared. */
int sai2b_reinitialize(sai2b_ctx* ctx);

/* RobotController::updateControllerTaskModels (RobotController.cpp:53-60) */
int sai2b_update_task_models(sai2b_ctx* ctx);
/* RobotController::computeControlTorques (RobotController.cpp:62-74). tau: [7][B]; may be NULL
 * (result stays in the ctx torque buffer). with_compensation == 0 reproduces the manual flow of
 * examples 04/18 (no-argument computeTorques(), torques summed). */
int sai2b_compute_control_torques(sai2b_ctx* ctx, double* tau, int on_device);
int sai2b_compute_control_torques_ex(sai2b_ctx* ctx, double* tau, int on_device,
									 int with_compensation);
/* update_task_models + compute_control_torques for the current state in ONE fused launch: the
 * batched hot path. Enqueued on the ctx stream; does not synchronise when tau == NULL. */
int sai2b_tick(sai2b_ctx* ctx, double* tau, int on_device);
/* ------------------------------------------------------------------ task-level plugin interface
 * The reference's tasks can be driven one by one, without a RobotController, through the virtuals of
 * TemplateTask (src/tasks/TemplateTask.h:42-88) — examples/01-joint_control.cpp:131-191,
 * examples/04-task_and_redundancy.cpp:141-150,188-206, examples/18. The same calls on task `task` of a ctx
 * (a ctx with ONE task is a standalone task object; tasks of one ctx share the robot state of
 * sai2b_set_state). The caller chains the nullspaces itself:
 *     sai2b_task_update_model(ctx, 0, NULL, 0);                 // N_prec = identity
 *     sai2b_task_get_nullspaces(ctx, 0, NULL, NULL, N01);       // getTaskAndPreviousNullspace()
 *     sai2b_task_update_model(ctx, 1, N01, 0);
 *     sai2b_task_compute_torques(ctx, 0, NULL, tau0, 0);        // computeTorques()
 *     sai2b_task_compute_torques(ctx, 1, tau0, tau1, 0);        // computeTorques(tau_prec)
 * Nothing is assumed about the tasks above: range decisions always take the SVD path. */
/* TemplateTask::updateTaskModel(N_prec) (TemplateTask.h:42; JointTask.cpp:218-283, MotionForceTask.cpp:247-268).
 * N_prec: [n*n][B] row-major inside the component index, NULL = identity. The singularity bookkeeping of a
 * MotionForceTask (SingularityHandler.cpp:230-295) advances once per call, as in the reference. */
int sai2b_task_update_model(sai2b_ctx* ctx, int task, const double* N_prec, int on_device);
/* TemplateTask::computeTorques() (tau_prec == NULL, TemplateTask.h:49) and computeTorques(tau_prec) (:58): the
 * task's own torques [n][B] under the N_prec of the last sai2b_task_update_model (identity when there was none,
 * the value a task is constructed with). Integrators and the task's internal OTG advance as in the reference.
 * A MotionForceTask ignores tau_prec (its compensation term is identically zero: MotionForceTask.cpp:270-276
 * with the never-assigned _Lambda, :140); a JointTask subtracts Jp^T R M_partial R^T S M^-1 tau_prec
 * (JointTask.cpp:285-292). Task models are those of the CURRENT state (never stale, see DESIGN.md "split API").
 * tau may be NULL (the result stays on the device). */
int sai2b_task_compute_torques(sai2b_ctx* ctx, int task, const double* tau_prec, double* tau, int on_device);
/* TemplateTask::reInitializeTask (TemplateTask.h:65; JointTask.cpp:91-107, MotionForceTask.cpp:204-245) of one task */
int sai2b_task_reinitialize(sai2b_ctx* ctx, int task);
/* TemplateTask::getTaskNullspace / getPreviousTasksNullspace / getTaskAndPreviousNullspace (TemplateTask.h:73-88)
 * of the last sai2b_task_update_model / sai2b_task_compute_torques of that task: host arrays [n*n][B], any NULL */
int sai2b_task_get_nullspaces(sai2b_ctx* ctx, int task, double* N_task, double* N_prec, double* N_total);

/* wait for everything enqueued on the ctx stream */
int sai2b_synchronize(sai2b_ctx* ctx);
/* the hipStream_t the ctx launches on (as void*) */
void* sai2b_stream(sai2b_ctx* ctx);
/* Stream contract for DEVICE-pointer arguments (every `on_device != 0` above and below). The ctx works on its own
 * non-blocking stream. A device input is read after everything the caller has enqueued on ITS stream up to the
 * call (the producer kernels) and the read is finished before anything the caller enqueues on that stream after
 * the call returns (so the buffer may be overwritten at once); a device result (tau with on_device) is complete
 * when the call returns. The caller's stream is the legacy default stream unless set here (pass the hipStream_t,
 * e.g. torch.cuda.current_stream().cuda_stream). Host-pointer arguments are always complete on return.
 * Buffers handed out by sai2b_device_buffer() are outside this contract: order them with sai2b_stream(). */
int sai2b_set_caller_stream(sai2b_ctx* ctx, void* stream);

/* Device-resident ctx buffers, for zero-copy producers/consumers. `which`: */
enum sai2b_buffer {
	SAI2B_BUF_Q = 0,	  /* [7][B]  */
	SAI2B_BUF_DQ = 1,	  /* [7][B]  */
	SAI2B_BUF_TAU = 2,	  /* [7][B]  */
	SAI2B_BUF_GOALS = 3,  /* per task; MFT: [30][B] = pos3 rot9 v3 w3 a3 alpha3 f3 m3; JT: [3k][B] */
	SAI2B_BUF_SENSED = 4, /* per MFT task: [6][B] sensor-frame force, moment */
	SAI2B_BUF_STATE = 5,  /* per task persistent state, see DESIGN.md */
	/* outputs of the task-level calls (sai2b_task_update_model), [n*n][B]: a manual hierarchy chains them on the device —
	 * sai2b_task_update_model(ctx, next, sai2b_device_buffer(ctx, SAI2B_BUF_TASK_N_TOTAL, task), 1) — instead of through
	 * sai2b_task_get_nullspaces() and the host. NULL before the task's first task-level call. */
	SAI2B_BUF_TASK_N = 6,		/* the task's nullspace N (getTaskNullspace) */
	SAI2B_BUF_TASK_N_TOTAL = 7	/* N * N_prec (getTaskAndPreviousNullspace) */
};
/* (A producer that writes q through SAI2B_BUF_Q bypasses the bookkeeping of the tasks' cached pose: an OTG
 * enabled / a space re-parametrised after such a write and before the next tick starts from the state as
 * written, not from the pose of the last torque computation.) */
void* sai2b_device_buffer(sai2b_ctx* ctx, int which, int task);

/* Optional per-robot outputs of the last tick (debug / observers such as
 * POPCBilateralTeleoperation.cpp:81-92). Introspection must be enabled BEFORE the tick whose values
 * are wanted (it selects a kernel variant that also stores these arrays); the getters fail with
 * SAI2B_INVALID_ARGUMENT otherwise. Arrays are host pointers; any may be NULL. */
int sai2b_enable_introspection(sai2b_ctx* ctx, int enable);
/* TemplateTask::getTaskAndPreviousNullspace (TemplateTask.h:88): [49][B] */
int sai2b_get_task_nullspace(sai2b_ctx* ctx, int task, double* N_total);
/* per-task torque contribution of the last tick: [7][B] */
int sai2b_get_task_torques(sai2b_ctx* ctx, int task, double* tau_task);
/* MFT: singular values [6][B], blending alpha [B], split index (non-singular rank) [B] as doubles */
int sai2b_get_mft_singularity(sai2b_ctx* ctx, int task, double* sigma, double* alpha,
							  double* ns_rank);
/* MFT, no introspection needed: the SingularityHandler's state after the last model update
 * (SingularityHandler.h:211-215) per robot, int [B] each, any may be NULL: number of singular directions
 * (_singularity_types.size(), 0 = fully non-singular), _type_1_counter, _type_2_counter. */
int sai2b_get_mft_singularity_state(sai2b_ctx* ctx, int task, int* n_singular, int* type_1_count,
									int* type_2_count);
/* MFT: unit-mass motion force and force-related terms of the last tick, [6][B] each — what
 * MotionForceTask::getUnitMassForce and the observers of POPCBilateralTeleoperation.cpp:81-92,172-182
 * read (MotionForceTask.cpp:478-487) */
int sai2b_get_mft_task_forces(sai2b_ctx* ctx, int task, double* F_unit, double* F_force);
/* ------------------------------------------------------------------ simulation harness
 * What the reference's examples obtain from the external sai2-simulation (examples/05-...cpp:215-236:
 * setJointTorques / integrate / getJointPositions, getJointVelocities): one control period of
 * rigid-body dynamics  M qdd + C dq (+ g) = tau  for every robot, with the state staying in device
 * memory between ticks. tau [7][B] (host, or device when on_device != 0; NULL = the torques of the
 * last sai2b_compute_control_torques / sai2b_tick) is held over dt, which is split into `substeps`
 * semi-implicit Euler steps. with_gravity uses sai2b_robot_model.gravity (the example worlds have
 * none). The state buffers (SAI2B_BUF_Q / SAI2B_BUF_DQ) are updated in place. */
int sai2b_sim_step(sai2b_ctx* ctx, const double* tau, int on_device, double dt, int substeps,
				   int with_gravity);
/* current joint positions / velocities to host arrays [7][B] (either may be NULL) */
int sai2b_get_state(sai2b_ctx* ctx, double* q, double* dq);
/* bias vector C(q, dq) dq (+ g(q)) of the current state, host [7][B] (Sai2Model::coriolisForce /
 * jointGravityVector of sai2-model) */
int sai2b_get_bias(sai2b_ctx* ctx, int with_gravity, double* bias);

/* Observers the reference's callers use between ticks, computed from the state and goal buffers as
 * they are now (the reference caches them at the last computeTorques): MotionForceTask::
 * getCurrentPosition / getCurrentOrientation (MotionForceTask.h:121-140), getSensedForce/
 * MomentControlWorldFrame (:154-165), getPositionError / getOrientationError
 * (MotionForceTask.cpp:540-546), and the norms sqrt(e^T sigma e) that goalPositionReached /
 * goalOrientationReached compare with their tolerance (:548-579). Host arrays [rows][B], any NULL. */
int sai2b_get_mft_status(sai2b_ctx* ctx, int task, double* pos, double* rot, double* sensed_force_world,
						 double* sensed_moment_world, double* pos_error, double* ori_error,
						 double* pos_error_norm, double* ori_error_norm);
/* MotionForceTask::getCurrentLinearVelocity / getCurrentAngularVelocity (MotionForceTask.h:127-146; J dq of
 * the state as it is now, MotionForceTask.cpp:293-298): host [3][B] each, any NULL */
int sai2b_get_mft_velocity(sai2b_ctx* ctx, int task, double* linear_velocity, double* angular_velocity);
/* MotionForceTask::sigmaForce / sigmaPosition / sigmaMoment / sigmaOrientation (MotionForceTask.h:610-613,
 * MotionForceTask.cpp:892-971) for the state as it is now (they depend on the robot only when the spaces are
 * parametrised in the compliant frame): host [9][B] row-major each, any NULL */
int sai2b_get_mft_sigma(sai2b_ctx* ctx, int task, double* sigma_force, double* sigma_position,
						double* sigma_moment, double* sigma_orientation);
/* MotionForceTask::setType1Posture (MotionForceTask.h:706 -> SingularityHandler.h:140-142): the posture the
 * type-1 singularity strategy holds, [n][B]; kept until the handler next refreshes it
 * (SingularityHandler.cpp:233-236), as in the reference */
int sai2b_set_mft_type1_posture(sai2b_ctx* ctx, int task, const double* q_des, int on_device);
/* getGoalPosition ... getGoalMoment (MotionForceTask.h:214-247,  JointTask.h:144-160) */
int sai2b_get_mft_goals(sai2b_ctx* ctx, int task, double* pos, double* rot, double* lin_vel, double* ang_vel,
						double* lin_acc, double* ang_acc, double* force, double* moment);
int sai2b_get_jt_goals(sai2b_ctx* ctx, int task, double* q, double* dq, double* ddq);

/* MotionForceTask::resetIntegrators / resetIntegratorsLinear / resetIntegratorsAngular
 * (MotionForceTask.cpp:988-1001) and JointTask::resetIntegrators: which = 0 all, 1 linear (position
 * and force integrals), 2 angular (orientation and moment integrals); JointTask: any value. */
int sai2b_reset_integrators(sai2b_ctx* ctx, int task, int which);

/* Desired state the control law tracked in the last torque computation: the goal, or with the
 * internal OTG on its next state (JointTask::getDesiredPosition/Velocity/Acceleration,
 * JointTask.h:182-198; MotionForceTask::getDesired*). Host arrays, SoA [rows][B]; any may be NULL.
 * No introspection switch needed. */
int sai2b_get_jt_desired(sai2b_ctx* ctx, int task, double* q, double* dq, double* ddq);
int sai2b_get_mft_desired(sai2b_ctx* ctx, int task, double* pos, double* rot, double* lin_vel,
						  double* ang_vel, double* lin_acc, double* ang_acc);
/* Internal OTG flags per robot as doubles [B]: OTG_joints/OTG_6dof_cartesian::isGoalReached()
 * (OTG_joints.h:152, OTG_6dof_cartesian.h:229) and the last ruckig Result (0 working, 1 finished,
 * < 0 error: ruckig/include/ruckig/result.hpp:6-18) */
int sai2b_get_otg_status(sai2b_ctx* ctx, int task, double* goal_reached, double* result);
/* Sai2Model::M(): [49][B]; J of MFT `task` (JWorldFrame): [42][B]; position [3][B], rotation [9][B] */
int sai2b_get_model(sai2b_ctx* ctx, int task, double* M, double* J, double* pos, double* rot);

/* Bench bookkeeping (on the ctx stream, ONE HIP event pair around `steps` back-to-back launches, so no event sits
 * between two kernels): first the first kernel of the tick alone — its average duration per launch in ms —, then the
 * tick's whole launch sequence; second_kernel_ms = what the generic kernel over the work list behind an SVD-free
 * kernel adds to a step (0 when the hierarchy has no such pass). The two add up to the step. The ticks run for real
 * (integrators, generators and singularity histories advance; in the first phase the robots the SVD-free kernel
 * declines are not served): a profiling call, not part of a control loop. */
int sai2b_profile_tick(sai2b_ctx* ctx, int steps, double* first_kernel_ms, double* second_kernel_ms);

/* HIP devices visible to this process (0 when there is none or the runtime fails): what a host that shards a batch
 * over the GPUs of a node creates one context each for (Sai2PrimitivesBatched.h: ShardedRobotController) */
int sai2b_device_count(void);

/* How many robots of the last tick — or of the last task-level call (sai2b_task_update_model / _compute_torques), whichever
 * came last — the SVD-free kernel handed to the generic (Jacobi-SVD) kernel: robots inside or leaving a
 * singularity-blending region (SingularityHandler.cpp:66-160) that the SVD-free kernel does not handle itself, levels it
 * cannot certify. The whole batch when the generic kernel ran alone. Waits for the ctx stream. */
int sai2b_get_fallback_count(sai2b_ctx* ctx, int* robots);

/* number of kernel launches and robots processed since creation (bench bookkeeping) */
int sai2b_counters(const sai2b_ctx* ctx, long long* launches, long long* ticks);

#ifdef __cplusplus
}
#endif
#endif /* SAI2B_H_ */
