"""ctypes view of the C ABI declared in include/sai2b.h.

Only POD structs and the loader live here. The product library is
``sai2-primitives-perso_amd/csrc/libsai2b.so`` (built by ``__graft_entry__.build()``); loading it
fails loudly when it has not been built — there is no CPU fallback.
"""
import ctypes as C
import os

DOF = 7  # the Panda's: default of the helpers that take no robot
MAX_DOF = 8
MAX_TASKS = 4
SH_HISTORY = 200

# enum sai2b_task_type (reference src/tasks/TemplateTask.h:19-23)
UNDEFINED, JOINT_TASK, MOTION_FORCE_TASK = 0, 1, 2
# enum sai2b_decoupling (reference src/helper_modules/Sai2PrimitivesCommonDefinitions.h:9-15)
FULL_DYNAMIC_DECOUPLING, BOUNDED_INERTIA_ESTIMATES, IMPEDANCE = 0, 1, 2
# enum sai2b_singular_vector_sign (not in the reference: the orientation of the singular vector classifySingularity
# perturbs along, SingularityHandler.cpp:253-265)
SV_SIGN_V_MAX_POSITIVE, SV_SIGN_V_MAX_NEGATIVE, SV_SIGN_EITHER, SV_SIGN_BOTH = 0, 1, 2, 3
# enum sai2b_status
OK, INVALID_ARGUMENT, RUNTIME_ERROR, UNSUPPORTED = 0, 1, 2, 3
# enum sai2b_buffer
BUF_Q, BUF_DQ, BUF_TAU, BUF_GOALS, BUF_SENSED, BUF_STATE, BUF_TASK_N, BUF_TASK_N_TOTAL = 0, 1, 2, 3, 4, 5, 6, 7

_d = C.c_double
_i = C.c_int


class RobotModel(C.Structure):
    """struct sai2b_robot_model"""

    _fields_ = [
        ("dof", _i),
        ("joint_xyz", (_d * 3) * MAX_DOF),
        ("joint_rpy", (_d * 3) * MAX_DOF),
        ("link_mass", _d * MAX_DOF),
        ("link_com", (_d * 3) * MAX_DOF),
        ("link_inertia", (_d * 6) * MAX_DOF),
        ("q_lower", _d * MAX_DOF),
        ("q_upper", _d * MAX_DOF),
        ("effort", _d * MAX_DOF),
        ("gravity", _d * 3),
        ("joint_type", _i * MAX_DOF),
    ]


class TaskConfig(C.Structure):
    """struct sai2b_task_config"""

    _fields_ = [
        ("type", _i),
        ("name", C.c_char * 64),
        ("loop_timestep", _d),
        ("dynamic_decoupling_type", _i),
        ("bie_threshold", _d),
        # JointTask
        ("task_dof", _i),
        ("joint_selection", _d * (MAX_DOF * MAX_DOF)),
        ("kp", _d * MAX_DOF),
        ("kv", _d * MAX_DOF),
        ("ki", _d * MAX_DOF),
        ("use_velocity_saturation", _i),
        ("saturation_velocity", _d * MAX_DOF),
        # MotionForceTask
        ("link", _i),
        ("frame_pos", _d * 3),
        ("frame_rot", _d * 9),
        ("partial_projection", _d * 36),
        ("pos_range", _i),
        ("ori_range", _i),
        ("parametrization_in_compliant_frame", _i),
        ("kp_pos", _d * 3),
        ("kv_pos", _d * 3),
        ("ki_pos", _d * 3),
        ("kp_ori", _d * 3),
        ("kv_ori", _d * 3),
        ("ki_ori", _d * 3),
        ("kp_force", _d * 3),
        ("kv_force", _d * 3),
        ("ki_force", _d * 3),
        ("kp_moment", _d * 3),
        ("kv_moment", _d * 3),
        ("ki_moment", _d * 3),
        ("kff_force", _d),
        ("kff_moment", _d),
        ("max_force_feedback", _d),
        ("max_moment_feedback", _d),
        ("closed_loop_force", _i),
        ("closed_loop_moment", _i),
        ("passivity_enabled", _i),
        ("force_space_dimension", _i),
        ("moment_space_dimension", _i),
        ("force_axis", _d * 3),
        ("moment_axis", _d * 3),
        ("linear_saturation_velocity", _d),
        ("angular_saturation_velocity", _d),
        ("sensor_rot", _d * 9),
        ("sensor_pos", _d * 3),
        # SingularityHandler
        ("s_min", _d),
        ("s_max", _d),
        ("s_abs_tol", _d),
        ("type_1_tol", _d),
        ("type_2_torque_ratio", _d),
        ("type_2_angle_threshold", _d),
        ("perturb_step_size", _d),
        ("sh_buffer_size", _i),
        ("kp_type_1", _d),
        ("kv_type_1", _d),
        ("kv_type_2", _d),
        ("enforce_type_1_strategy", _i),
        ("enforce_handling_strategy", _i),
        ("singular_vector_sign", _i),
        # internal OTG
        ("use_internal_otg", _i),
        ("internal_otg_jerk_limited", _i),
        ("otg_max_velocity", _d * MAX_DOF),
        ("otg_max_acceleration", _d * MAX_DOF),
        ("otg_max_linear_velocity", _d),
        ("otg_max_linear_acceleration", _d),
        ("otg_max_angular_velocity", _d),
        ("otg_max_angular_acceleration", _d),
        ("otg_max_jerk", _d * MAX_DOF),
        ("otg_max_linear_jerk", _d),
        ("otg_max_angular_jerk", _d),
        ("unsafe_motion_gains", _i),
        ("robot_dof", _i),
    ]


URDF_MAX_LINKS = 32


class UrdfLinks(C.Structure):
    """sai2b_urdf_links"""

    _fields_ = [
        ("n_links", _i),
        ("name", (C.c_char * 64) * URDF_MAX_LINKS),
        ("moving_link", _i * URDF_MAX_LINKS),
        ("pos", (_d * 3) * URDF_MAX_LINKS),
        ("rot", (_d * 9) * URDF_MAX_LINKS),
    ]


def struct_to_dict(s):
    """Flatten a ctypes struct into plain python values (for comparisons in tests)."""
    out = {}
    for name, _ in s._fields_:
        v = getattr(s, name)
        if isinstance(v, C.Array):
            if isinstance(v, bytes):
                out[name] = v
            else:
                flat = []

                def rec(a):
                    for x in a:
                        if isinstance(x, C.Array):
                            rec(x)
                        else:
                            flat.append(x)

                rec(v)
                out[name] = flat
        else:
            out[name] = v
    return out


PKG_DIR = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("SAI2B_LIB") or os.path.join(PKG_DIR, "csrc", "libsai2b.so")

# every symbol include/sai2b.h declares (tests check that the built library exports all of them)
EXPORTS = [
    "sai2b_panda_model",
    "sai2b_model_merge_fixed_body",
    "sai2b_model_from_urdf",
    "sai2b_urdf_resolve_frame",
    "sai2b_model_set_base_transform",
    "sai2b_default_joint_task",
    "sai2b_default_joint_task_dof",
    "sai2b_default_motion_force_task",
    "sai2b_default_motion_force_task_dof",
    "sai2b_validate_tasks",
    "sai2b_create",
    "sai2b_destroy",
    "sai2b_last_error",
    "sai2b_batch",
    "sai2b_num_tasks",
    "sai2b_num_joints",
    "sai2b_update_task_config",
    "sai2b_enable_gravity_compensation",
    "sai2b_set_state",
    "sai2b_set_mft_goals",
    "sai2b_set_mft_goal_wrench",
    "sai2b_set_mft_sensed_wrench",
    "sai2b_set_jt_goals",
    "sai2b_reinitialize",
    "sai2b_update_task_models",
    "sai2b_compute_control_torques",
    "sai2b_compute_control_torques_ex",
    "sai2b_tick",
    "sai2b_task_update_model",
    "sai2b_task_compute_torques",
    "sai2b_task_reinitialize",
    "sai2b_task_get_nullspaces",
    "sai2b_get_mft_singularity_state",
    "sai2b_get_mft_velocity",
    "sai2b_get_mft_sigma",
    "sai2b_set_mft_type1_posture",
    "sai2b_synchronize",
    "sai2b_stream",
    "sai2b_set_caller_stream",
    "sai2b_device_buffer",
    "sai2b_enable_introspection",
    "sai2b_get_task_nullspace",
    "sai2b_get_task_torques",
    "sai2b_get_mft_singularity",
    "sai2b_get_mft_task_forces",
    "sai2b_get_mft_status",
    "sai2b_get_mft_goals",
    "sai2b_get_jt_goals",
    "sai2b_reset_integrators",
    "sai2b_sim_step",
    "sai2b_get_state",
    "sai2b_get_bias",
    "sai2b_get_jt_desired",
    "sai2b_get_mft_desired",
    "sai2b_get_otg_status",
    "sai2b_get_model",
    "sai2b_profile_tick",
    "sai2b_device_count",
    "sai2b_get_fallback_count",
    "sai2b_counters",
]

_lib = None


def load_library():
    """Load csrc/libsai2b.so (the HIP product library). Raises if it is missing."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build the HIP extension first "
            "(python -c 'import __graft_entry__ as g; g.build()'). There is no CPU fallback."
        )
    # torch (device memory / streams / torch.distributed plumbing) bundles its own HIP runtime with the
    # same SONAME as /opt/rocm's; two HIP runtimes in one process cannot both own the GPU, so make
    # sure torch's copy is the one already mapped before libsai2b.so resolves libamdhip64.so.7.
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    lib = C.CDLL(LIB_PATH)
    P = C.POINTER
    dp = P(_d)
    vp = C.c_void_p
    lib.sai2b_panda_model.argtypes = [P(RobotModel)]
    lib.sai2b_model_merge_fixed_body.argtypes = [P(RobotModel), _i, dp, dp, _d, dp, dp]
    lib.sai2b_model_from_urdf.argtypes = [C.c_char_p, _i, P(RobotModel), P(UrdfLinks)]
    lib.sai2b_urdf_resolve_frame.argtypes = [P(UrdfLinks), C.c_char_p, dp, dp, P(_i), dp, dp]
    lib.sai2b_model_set_base_transform.argtypes = [P(RobotModel), dp, dp]
    lib.sai2b_default_joint_task.argtypes = [P(TaskConfig), C.c_char_p, _i, dp]
    lib.sai2b_default_joint_task_dof.argtypes = [P(TaskConfig), C.c_char_p, _i, _i, dp]
    lib.sai2b_default_motion_force_task_dof.argtypes = [P(TaskConfig), C.c_char_p, _i, _i, dp, dp, _i, dp, _i, dp]
    lib.sai2b_default_motion_force_task.argtypes = [P(TaskConfig), C.c_char_p, _i, dp, dp, _i, dp, _i, dp]
    lib.sai2b_validate_tasks.argtypes = [P(TaskConfig), _i, C.c_char_p, _i]
    lib.sai2b_create.argtypes = [P(RobotModel), P(TaskConfig), _i, _i, _i]
    lib.sai2b_create.restype = vp
    lib.sai2b_destroy.argtypes = [vp]
    lib.sai2b_destroy.restype = None
    lib.sai2b_last_error.argtypes = [vp]
    lib.sai2b_last_error.restype = C.c_char_p
    lib.sai2b_batch.argtypes = [vp]
    lib.sai2b_num_tasks.argtypes = [vp]
    lib.sai2b_num_joints.argtypes = [vp]
    lib.sai2b_update_task_config.argtypes = [vp, _i, P(TaskConfig)]
    lib.sai2b_enable_gravity_compensation.argtypes = [vp, _i]
    lib.sai2b_set_state.argtypes = [vp, vp, vp, _i]
    lib.sai2b_set_mft_goals.argtypes = [vp, _i, vp, vp, vp, vp, vp, vp, _i]
    lib.sai2b_set_mft_goal_wrench.argtypes = [vp, _i, vp, vp, _i]
    lib.sai2b_set_mft_sensed_wrench.argtypes = [vp, _i, vp, vp, _i]
    lib.sai2b_set_jt_goals.argtypes = [vp, _i, vp, vp, vp, _i]
    lib.sai2b_reinitialize.argtypes = [vp]
    lib.sai2b_update_task_models.argtypes = [vp]
    lib.sai2b_compute_control_torques.argtypes = [vp, vp, _i]
    lib.sai2b_compute_control_torques_ex.argtypes = [vp, vp, _i, _i]
    lib.sai2b_tick.argtypes = [vp, vp, _i]
    lib.sai2b_task_update_model.argtypes = [vp, _i, vp, _i]
    lib.sai2b_task_compute_torques.argtypes = [vp, _i, vp, vp, _i]
    lib.sai2b_task_reinitialize.argtypes = [vp, _i]
    lib.sai2b_task_get_nullspaces.argtypes = [vp, _i, vp, vp, vp]
    lib.sai2b_get_mft_singularity_state.argtypes = [vp, _i, vp, vp, vp]
    lib.sai2b_get_mft_velocity.argtypes = [vp, _i, vp, vp]
    lib.sai2b_get_mft_sigma.argtypes = [vp, _i, vp, vp, vp, vp]
    lib.sai2b_set_mft_type1_posture.argtypes = [vp, _i, vp, _i]
    lib.sai2b_synchronize.argtypes = [vp]
    lib.sai2b_stream.argtypes = [vp]
    lib.sai2b_stream.restype = vp
    lib.sai2b_set_caller_stream.argtypes = [vp, vp]
    lib.sai2b_device_buffer.argtypes = [vp, _i, _i]
    lib.sai2b_device_buffer.restype = vp
    lib.sai2b_enable_introspection.argtypes = [vp, _i]
    lib.sai2b_get_task_nullspace.argtypes = [vp, _i, vp]
    lib.sai2b_get_task_torques.argtypes = [vp, _i, vp]
    lib.sai2b_get_mft_singularity.argtypes = [vp, _i, vp, vp, vp]
    lib.sai2b_get_mft_task_forces.argtypes = [vp, _i, vp, vp]
    lib.sai2b_get_mft_status.argtypes = [vp, _i] + [vp] * 8
    lib.sai2b_get_mft_goals.argtypes = [vp, _i] + [vp] * 8
    lib.sai2b_get_jt_goals.argtypes = [vp, _i, vp, vp, vp]
    lib.sai2b_reset_integrators.argtypes = [vp, _i, _i]
    lib.sai2b_sim_step.argtypes = [vp, vp, _i, _d, _i, _i]
    lib.sai2b_get_state.argtypes = [vp, vp, vp]
    lib.sai2b_get_bias.argtypes = [vp, _i, vp]
    lib.sai2b_get_jt_desired.argtypes = [vp, _i, vp, vp, vp]
    lib.sai2b_get_mft_desired.argtypes = [vp, _i, vp, vp, vp, vp, vp, vp]
    lib.sai2b_get_otg_status.argtypes = [vp, _i, vp, vp]
    lib.sai2b_get_model.argtypes = [vp, _i, vp, vp, vp, vp]
    lib.sai2b_profile_tick.argtypes = [vp, _i, P(_d), P(_d)]
    lib.sai2b_get_fallback_count.argtypes = [vp, P(_i)]
    lib.sai2b_counters.argtypes = [vp, P(C.c_longlong), P(C.c_longlong)]
    _lib = lib
    return lib
