"""Batched operational-space controller for MI355X (gfx950).

Host-side mirror of the sai2-primitives RobotController / MotionForceTask / JointTask API over the
C ABI of include/sai2b.h, whose implementation is hand-written HIP (csrc/). See DESIGN.md.
"""
from . import _abi, sharding, workloads  # noqa: F401
from ._abi import (  # noqa: F401
    BOUNDED_INERTIA_ESTIMATES,
    DOF,
    FULL_DYNAMIC_DECOUPLING,
    IMPEDANCE,
    JOINT_TASK,
    MOTION_FORCE_TASK,
    RobotModel,
    SV_SIGN_BOTH,
    SV_SIGN_EITHER,
    SV_SIGN_V_MAX_NEGATIVE,
    SV_SIGN_V_MAX_POSITIVE,
    TaskConfig,
)
from .controller import (  # noqa: F401,E402
    BatchedRobotModel,
    BatchedSimulation,
    Controller,
    JointTask,
    MotionForceTask,
    RobotController,
    joint_task_config,
    model_from_urdf,
    resolve_link_frame,
    with_base_transform,
    motion_force_task_config,
    panda_model,
    task_configs,
)
