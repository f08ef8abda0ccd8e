"""Host-side mirror of the reference's controller interface over the C ABI (include/sai2b.h).

``Controller`` is the thin array-level binding (SoA numpy arrays or raw device pointers in, torques
out). The classes ``BatchedRobotModel``, ``JointTask``, ``MotionForceTask`` and ``RobotController``
keep the reference's names, argument meaning and error behaviour
(reference src/RobotController.h:25-40, src/tasks/JointTask.h:56-384,
src/tasks/MotionForceTask.h:96-753) but every vector/matrix is batched: shape ``[C, B]``
(component-major, batch-minor). std::invalid_argument becomes ValueError; HIP failures RuntimeError.

All numerical work happens in csrc/libsai2b.so (HIP, gfx950). There is no CPU fallback: creating a
controller without the library or without a GPU raises.
"""
import ctypes as C

import numpy as np

from . import _abi
from ._abi import DOF, MOTION_FORCE_TASK, RobotModel, TaskConfig
from .workloads import EE_FRAME_POS, EE_LINK


def _check(lib, handle, rc):
    if rc == _abi.OK:
        return
    msg = lib.sai2b_last_error(handle)
    msg = msg.decode() if msg else "unknown error"
    if rc == _abi.INVALID_ARGUMENT:
        raise ValueError(msg)
    raise RuntimeError(msg)


def _dp(a):
    return None if a is None else a.ctypes.data_as(C.POINTER(C.c_double))


def panda_model():
    """sai2b_panda_model(): Panda constants with the fixed end-effector body merged into link 7."""
    lib = _abi.load_library()
    m = RobotModel()
    _check(lib, None, lib.sai2b_panda_model(C.byref(m)))
    return m


def model_from_urdf(urdf, is_file=True):
    """sai2b_model_from_urdf(): (RobotModel, UrdfLinks) from a URDF file name or XML text"""
    lib = _abi.load_library()
    m, links = RobotModel(), _abi.UrdfLinks()
    _check(lib, None, lib.sai2b_model_from_urdf(urdf.encode(), 1 if is_file else 0, C.byref(m), C.byref(links)))
    return m, links


def with_base_transform(model, pos, rot=None):
    """sai2b_model_set_base_transform() on a COPY of the model: Sai2Model::setTRobotBase (examples/05-...cpp:69).
    The tasks work in the world frame (MotionForceTask.cpp:100-103, 262: positionInWorld / JWorldFrame), so the base
    pose is part of the model a Controller is built from."""
    lib = _abi.load_library()
    m = RobotModel()
    C.memmove(C.byref(m), C.byref(model), C.sizeof(RobotModel))
    p = np.ascontiguousarray(pos, dtype=np.float64).reshape(3)
    r = None if rot is None else np.ascontiguousarray(rot, dtype=np.float64).reshape(9)
    _check(lib, None, lib.sai2b_model_set_base_transform(C.byref(m), _dp(p), _dp(r)))
    return m


def resolve_link_frame(links, link_name, pos_in_link=(0.0, 0.0, 0.0), rot_in_link=None):
    """sai2b_urdf_resolve_frame(): link name + frame in that link -> (moving link index, frame_pos, frame_rot)"""
    lib = _abi.load_library()
    p = np.ascontiguousarray(pos_in_link, dtype=np.float64)
    r = None if rot_in_link is None else np.ascontiguousarray(rot_in_link, dtype=np.float64)
    link, fp, fr = C.c_int(), np.zeros(3), np.zeros(9)
    _check(lib, None, lib.sai2b_urdf_resolve_frame(C.byref(links), link_name.encode(), _dp(p), _dp(r), C.byref(link), _dp(fp), _dp(fr)))
    return link.value, fp, fr.reshape(3, 3)


def joint_task_config(name=None, selection=None, internal_otg=False, robot_dof=DOF):
    """sai2b_default_joint_task(): JointTask ctor + defaults (JointTask.cpp:14-89).
    The library default is the reference's: internal OTG on (JointTask.h:38). This helper turns it
    off unless internal_otg=True, the way the reference's examples call disableInternalOtg() after
    construction (examples/05-...cpp:117) and as BASELINE's workloads are defined."""
    lib = _abi.load_library()
    c = TaskConfig()
    sel = None if selection is None else np.ascontiguousarray(selection, dtype=np.float64)
    if sel is not None and (sel.ndim != 2 or sel.shape[1] != robot_dof):
        raise ValueError("joint selection matrix size not consistent with robot dof in JointTask constructor\n")
    rc = lib.sai2b_default_joint_task_dof(C.byref(c), name.encode() if name else None, int(robot_dof), 0 if sel is None else sel.shape[0],
                                          _dp(sel))
    _check(lib, None, rc)
    if not internal_otg:
        c.use_internal_otg = 0
    return c


def motion_force_task_config(name=None, link=EE_LINK, frame_pos=EE_FRAME_POS, frame_rot=None, partial=None,
                             internal_otg=False, robot_dof=DOF):
    """sai2b_default_motion_force_task(): MotionForceTask ctors + defaults (MotionForceTask.cpp:16-202).
    partial = (translation directions [n,3], rotation directions [m,3]) selects the partial-task ctor.
    internal_otg: see joint_task_config (library default on, this helper's default off)."""
    lib = _abi.load_library()
    c = TaskConfig()
    fp = np.ascontiguousarray(frame_pos, dtype=np.float64)
    fr = None if frame_rot is None else np.ascontiguousarray(frame_rot, dtype=np.float64)
    if partial is None:
        nt, nr, dt, dr = -1, -1, None, None
    else:
        dt = np.ascontiguousarray(partial[0], dtype=np.float64).reshape(-1, 3)
        dr = np.ascontiguousarray(partial[1], dtype=np.float64).reshape(-1, 3)
        nt, nr = dt.shape[0], dr.shape[0]
    rc = lib.sai2b_default_motion_force_task_dof(
        C.byref(c), name.encode() if name else None, int(robot_dof), link, _dp(fp), _dp(fr),
        nt, _dp(dt) if nt and nt > 0 else None, nr, _dp(dr) if nr and nr > 0 else None,
    )
    _check(lib, None, rc)
    if not internal_otg:
        c.use_internal_otg = 0
    return c


def task_configs(tasks):
    """workloads.make_inputs()['tasks'] -> list of TaskConfig built by the product's helpers"""
    out = []
    for t, (kind, prm) in enumerate(tasks):
        if kind == "jt":
            out.append(joint_task_config(f"joint_task_{t}", prm.get("selection")))
        else:
            out.append(motion_force_task_config(f"motion_force_task_{t}", partial=prm.get("partial")))
    return out


class Controller:
    """Array-level binding of one sai2b_ctx (one GPU, one batch)."""

    def __init__(self, model, tasks, batch, device=0, introspection=False):
        self.lib = _abi.load_library()
        self.B = int(batch)
        self.dof = int(model.dof)  # joints of the robot: the library routes to its build for that size
        self.tasks = list(tasks)
        arr = (TaskConfig * len(self.tasks))(*self.tasks)
        self.h = self.lib.sai2b_create(C.byref(model), arr, len(self.tasks), self.B, int(device))
        if not self.h:
            msg = self.lib.sai2b_last_error(None).decode()
            if "hip" in msg.lower():
                raise RuntimeError(msg)
            raise ValueError(msg)
        self._caller_stream = 0  # the legacy default stream
        if introspection:
            self.enable_introspection(True)

    # -- lifetime
    def close(self):
        if getattr(self, "h", None):
            self.lib.sai2b_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _rc(self, rc):
        _check(self.lib, self.h, rc)

    def _in(self, a, rows):
        """-> (pointer, on_device, keepalive) for a numpy array, a torch CUDA tensor, or None"""
        if a is None:
            return None, None
        if hasattr(a, "data_ptr"):  # torch tensor
            if not a.is_cuda or not a.is_contiguous() or tuple(a.shape) != (rows, self.B) or str(a.dtype) != "torch.float64":
                raise ValueError(f"expected a contiguous float64 CUDA tensor of shape ({rows}, {self.B})")
            # the tensor was (or is being) produced on torch's current stream: sai2b.h "stream contract"
            import torch

            s = torch.cuda.current_stream(a.device).cuda_stream
            if s != self._caller_stream:
                self._rc(self.lib.sai2b_set_caller_stream(self.h, C.c_void_p(s)))
                self._caller_stream = s
            return C.c_void_p(a.data_ptr()), a
        arr = np.ascontiguousarray(a, dtype=np.float64)
        if arr.shape != (rows, self.B):
            raise ValueError(f"expected an array of shape ({rows}, {self.B}), got {arr.shape}")
        return C.c_void_p(arr.ctypes.data), arr

    @staticmethod
    def _dev(*objs):
        kinds = {hasattr(o, "data_ptr") for o in objs if o is not None}
        if len(kinds) > 1:
            raise ValueError("mixing host arrays and device tensors in one call")
        return 1 if kinds == {True} else 0

    # -- configuration
    def update_task_config(self, task, cfg):
        self._rc(self.lib.sai2b_update_task_config(self.h, task, C.byref(cfg)))
        self.tasks[task] = cfg

    def enable_gravity_compensation(self, on=True):
        self._rc(self.lib.sai2b_enable_gravity_compensation(self.h, int(on)))

    def enable_introspection(self, on=True):
        self._rc(self.lib.sai2b_enable_introspection(self.h, int(on)))

    # -- inputs
    def set_state(self, q=None, dq=None):
        pq, kq = self._in(q, self.dof)
        pd, kd = self._in(dq, self.dof)
        self._rc(self.lib.sai2b_set_state(self.h, pq, pd, self._dev(q, dq)))

    def set_mft_goals(self, task, pos=None, rot=None, v=None, w=None, a=None, alpha=None):
        objs = (pos, rot, v, w, a, alpha)
        ins = [self._in(o, r) for o, r in zip(objs, (3, 9, 3, 3, 3, 3))]
        self._rc(self.lib.sai2b_set_mft_goals(self.h, task, *[p for p, _ in ins], self._dev(*objs)))

    def set_mft_goal_wrench(self, task, f=None, m=None):
        ins = [self._in(f, 3), self._in(m, 3)]
        self._rc(self.lib.sai2b_set_mft_goal_wrench(self.h, task, ins[0][0], ins[1][0], self._dev(f, m)))

    def set_mft_sensed_wrench(self, task, f=None, m=None):
        ins = [self._in(f, 3), self._in(m, 3)]
        self._rc(self.lib.sai2b_set_mft_sensed_wrench(self.h, task, ins[0][0], ins[1][0], self._dev(f, m)))

    def set_jt_goals(self, task, q=None, dq=None, ddq=None):
        if not (0 <= task < len(self.tasks)) or self.tasks[task].type != _abi.JOINT_TASK:
            raise ValueError("sai2b_set_jt_goals: task is not a JointTask")
        k0 = self.tasks[task].task_dof
        ins = [self._in(o, k0) for o in (q, dq, ddq)]
        self._rc(self.lib.sai2b_set_jt_goals(self.h, task, *[p for p, _ in ins], self._dev(q, dq, ddq)))

    # -- the path
    def reinitialize(self):
        self._rc(self.lib.sai2b_reinitialize(self.h))

    def update_task_models(self):
        self._rc(self.lib.sai2b_update_task_models(self.h))

    def compute_control_torques(self, with_compensation=True, out=None):
        return self._torques(lambda p, dev: self.lib.sai2b_compute_control_torques_ex(self.h, p, dev, int(with_compensation)), out)

    def tick(self, out=None, want_output=True):
        """fused update_task_models + compute_control_torques; with want_output=False only enqueues"""
        if not want_output:
            self._rc(self.lib.sai2b_tick(self.h, None, 0))
            return None
        return self._torques(lambda p, dev: self.lib.sai2b_tick(self.h, p, dev), out)

    def _torques(self, call, out):
        if out is None:
            out = np.empty((self.dof, self.B))
        p, keep = self._in(out, self.dof)
        if not hasattr(out, "data_ptr") and keep is not out:
            raise ValueError("out must be a C-contiguous float64 array")
        self._rc(call(p, self._dev(out)))
        return out

    # -- task-level plugin interface (TemplateTask.h:42-88): one task driven on its own
    def task_update_model(self, task, N_prec=None):
        """TemplateTask::updateTaskModel(N_prec); N_prec [n*n][B] (numpy or torch CUDA), None = identity"""
        p, _ = self._in(N_prec, self.dof * self.dof)
        self._rc(self.lib.sai2b_task_update_model(self.h, task, p, self._dev(N_prec)))

    def task_update_model_behind(self, task, previous_task):
        """updateTaskModel(previous->getTaskAndPreviousNullspace()) without the host round trip of the [n*n][B] matrix:
        the previous task's N * N_prec is read where its own update left it on the device (same context, same stream)"""
        p = self.lib.sai2b_device_buffer(self.h, _abi.BUF_TASK_N_TOTAL, int(previous_task))
        if not p:
            raise ValueError("task_update_model_behind: the previous task has no task-level model yet")
        self._rc(self.lib.sai2b_task_update_model(self.h, task, C.c_void_p(p), 1))

    def task_compute_torques(self, task, tau_prec=None, out=None):
        """TemplateTask::computeTorques() / computeTorques(tau_prec): the task's own torques [n][B]"""
        p, _ = self._in(tau_prec, self.dof)
        if out is None:
            out = np.empty((self.dof, self.B))
        po, keep = self._in(out, self.dof)
        if tau_prec is not None and self._dev(tau_prec) != self._dev(out):
            raise ValueError("tau_prec and out must both be host arrays or both device tensors")
        self._rc(self.lib.sai2b_task_compute_torques(self.h, task, p, po, self._dev(out)))
        return out

    def task_reinitialize(self, task):
        self._rc(self.lib.sai2b_task_reinitialize(self.h, task))

    def task_nullspaces(self, task):
        """(N, N_prec, N * N_prec) of the task's last model update, [n*n][B] each"""
        out = [np.empty((self.dof * self.dof, self.B)) for _ in range(3)]
        self._rc(self.lib.sai2b_task_get_nullspaces(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return tuple(out)

    def get_mft_velocity(self, task):
        v, w = np.empty((3, self.B)), np.empty((3, self.B))
        self._rc(self.lib.sai2b_get_mft_velocity(self.h, task, C.c_void_p(v.ctypes.data), C.c_void_p(w.ctypes.data)))
        return v, w

    def get_mft_sigma(self, task):
        """sigmaForce, sigmaPosition, sigmaMoment, sigmaOrientation: [9][B] row-major each"""
        out = [np.empty((9, self.B)) for _ in range(4)]
        self._rc(self.lib.sai2b_get_mft_sigma(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return tuple(out)

    def set_mft_type1_posture(self, task, q_des):
        p, _ = self._in(q_des, self.dof)
        self._rc(self.lib.sai2b_set_mft_type1_posture(self.h, task, p, self._dev(q_des)))

    def get_singularity_types_count(self, task):
        """per robot: singular directions found by the last model update (SingularityHandler.h:211), no introspection needed"""
        n = np.empty(self.B, dtype=np.int32)
        self._rc(self.lib.sai2b_get_mft_singularity_state(self.h, task, C.c_void_p(n.ctypes.data), None, None))
        return n

    def get_mft_singularity_state(self, task):
        """per robot: singular directions of the last model update, type-1 and type-2 counts of the handler's history
        (SingularityHandler.h:211-215); no introspection needed"""
        n, c1, c2 = (np.empty(self.B, dtype=np.int32) for _ in range(3))
        self._rc(self.lib.sai2b_get_mft_singularity_state(self.h, task, *[C.c_void_p(x.ctypes.data) for x in (n, c1, c2)]))
        return n, c1, c2

    def synchronize(self):
        self._rc(self.lib.sai2b_synchronize(self.h))

    def stream(self):
        return self.lib.sai2b_stream(self.h)

    def device_buffer(self, which, task=-1):
        return self.lib.sai2b_device_buffer(self.h, which, task)

    def profile_tick(self, steps):
        """-> (first_kernel_ms, second_kernel_ms): HIP-event average duration per launch of each kernel of
        the fused tick, measured on the ctx stream"""
        a, b = C.c_double(), C.c_double()
        self._rc(self.lib.sai2b_profile_tick(self.h, int(steps), C.byref(a), C.byref(b)))
        return a.value, b.value

    def fallback_count(self):
        """robots of the last tick that ran the generic (Jacobi-SVD) kernel behind the SVD-free one"""
        n = C.c_int()
        self._rc(self.lib.sai2b_get_fallback_count(self.h, C.byref(n)))
        return n.value

    def counters(self):
        a, b = C.c_longlong(), C.c_longlong()
        self.lib.sai2b_counters(self.h, C.byref(a), C.byref(b))
        return a.value, b.value

    # -- introspection of the last tick
    def get_task_nullspace(self, task):
        out = np.empty((self.dof * self.dof, self.B))
        self._rc(self.lib.sai2b_get_task_nullspace(self.h, task, C.c_void_p(out.ctypes.data)))
        return out

    def get_task_torques(self, task):
        out = np.empty((self.dof, self.B))
        self._rc(self.lib.sai2b_get_task_torques(self.h, task, C.c_void_p(out.ctypes.data)))
        return out

    def get_mft_singularity(self, task):
        s, a, r = np.empty((6, self.B)), np.empty(self.B), np.empty(self.B)
        self._rc(self.lib.sai2b_get_mft_singularity(self.h, task, *[C.c_void_p(x.ctypes.data) for x in (s, a, r)]))
        return s, a, r

    def get_mft_task_forces(self, task):
        fu, ff = np.empty((6, self.B)), np.empty((6, self.B))
        self._rc(self.lib.sai2b_get_mft_task_forces(self.h, task, C.c_void_p(fu.ctypes.data), C.c_void_p(ff.ctypes.data)))
        return fu, ff

    def get_mft_status(self, task):
        """dict: pos [3,B], rot [9,B], sensed_force / sensed_moment (world frame) [3,B], pos_error, ori_error [3,B],
        pos_error_norm, ori_error_norm [B] (what goalPositionReached / goalOrientationReached compare)"""
        B = self.B
        names = ("pos", "rot", "sensed_force", "sensed_moment", "pos_error", "ori_error", "pos_error_norm", "ori_error_norm")
        out = [np.empty((r, B)) for r in (3, 9, 3, 3, 3, 3)] + [np.empty(B), np.empty(B)]
        self._rc(self.lib.sai2b_get_mft_status(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return dict(zip(names, out))

    def get_mft_goals(self, task):
        B = self.B
        out = [np.empty((r, B)) for r in (3, 9, 3, 3, 3, 3, 3, 3)]
        self._rc(self.lib.sai2b_get_mft_goals(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return tuple(out)

    def get_jt_goals(self, task):
        k0 = self.tasks[task].task_dof
        out = [np.empty((k0, self.B)) for _ in range(3)]
        self._rc(self.lib.sai2b_get_jt_goals(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return tuple(out)

    def reset_integrators(self, task, which=0):
        """0 all, 1 linear (position / force), 2 angular (orientation / moment); JointTask: all"""
        self._rc(self.lib.sai2b_reset_integrators(self.h, task, int(which)))

    # -- simulation harness
    def sim_step(self, tau=None, dt=0.001, substeps=1, with_gravity=False):
        """one control period of rigid-body dynamics under `tau` (None = the last computed torques; numpy
        [7][B] or a torch CUDA tensor), state updated in place on the device"""
        p, _ = self._in(tau, self.dof)
        self._rc(self.lib.sai2b_sim_step(self.h, p, self._dev(tau), float(dt), int(substeps), int(with_gravity)))

    def get_state(self):
        q, dq = np.empty((self.dof, self.B)), np.empty((self.dof, self.B))
        self._rc(self.lib.sai2b_get_state(self.h, C.c_void_p(q.ctypes.data), C.c_void_p(dq.ctypes.data)))
        return q, dq

    def get_bias(self, with_gravity=False):
        out = np.empty((self.dof, self.B))
        self._rc(self.lib.sai2b_get_bias(self.h, int(with_gravity), C.c_void_p(out.ctypes.data)))
        return out

    def get_jt_desired(self, task):
        """desired q, dq, ddq of a JointTask: the goal, or the internal OTG's next state"""
        k0 = self.tasks[task].task_dof
        out = [np.empty((k0, self.B)) for _ in range(3)]
        self._rc(self.lib.sai2b_get_jt_desired(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return tuple(out)

    def get_mft_desired(self, task):
        """desired position, orientation, linear/angular velocity, linear/angular acceleration"""
        B = self.B
        out = [np.empty((r, B)) for r in (3, 9, 3, 3, 3, 3)]
        self._rc(self.lib.sai2b_get_mft_desired(self.h, task, *[C.c_void_p(x.ctypes.data) for x in out]))
        return tuple(out)

    def get_otg_status(self, task):
        """per robot: isGoalReached() and the last ruckig Result of the task's internal OTG"""
        a, b = np.empty(self.B), np.empty(self.B)
        self._rc(self.lib.sai2b_get_otg_status(self.h, task, C.c_void_p(a.ctypes.data), C.c_void_p(b.ctypes.data)))
        return a, b

    def get_model(self, task=-1):
        M = np.empty((self.dof * self.dof, self.B))
        if task < 0:
            self._rc(self.lib.sai2b_get_model(self.h, -1, C.c_void_p(M.ctypes.data), None, None, None))
            return M
        J, x, R = np.empty((6 * self.dof, self.B)), np.empty((3, self.B)), np.empty((9, self.B))
        self._rc(self.lib.sai2b_get_model(self.h, task, *[C.c_void_p(a.ctypes.data) for a in (M, J, x, R)]))
        return M, J, x, R


# --------------------------------------------------------------------------------------------------
# reference-named, batch-aware facade
# --------------------------------------------------------------------------------------------------
class BatchedRobotModel:
    """Stands where the reference takes ``std::shared_ptr<Sai2Model::Sai2Model>``: the constant
    model plus the batch size and the device; q/dq are set on it as on the reference's model
    (examples/05-using_robot_controller.cpp:143-145)."""

    def __init__(self, batch, model=None, device=0, urdf_file=None):
        self.batch = int(batch)
        self.device = int(device)
        self.links = None
        if urdf_file is not None:  # Sai2Model::Sai2Model(urdf_file) (examples/05-...cpp:96-97)
            model, self.links = model_from_urdf(urdf_file)
        self.model = model if model is not None else panda_model()
        self._q = np.zeros((self.model.dof, self.batch))
        self._dq = np.zeros((self.model.dof, self.batch))
        self._controller = None
        self._standalone = []  # tasks driven on their own (TemplateTask-level calls): each has its 1-task context

    def dof(self):
        return int(self.model.dof)

    def setTRobotBase(self, pos, rot=None):
        """Sai2Model::setTRobotBase (examples/05-using_robot_controller.cpp:69): the robot base in the world; replaces an
        earlier one. Part of the model the kernels see (with_base_transform): before a controller or a task runs on it."""
        if self._controller is not None or self._standalone:
            raise ValueError("setTRobotBase must be called before a controller or a task runs on this robot model")
        if not hasattr(self, "_model_in_base"):
            self._model_in_base = self.model
        self.model = with_base_transform(self._model_in_base, pos, rot)
        self._base = (np.array(pos, dtype=np.float64).reshape(3), np.eye(3) if rot is None else np.array(rot, dtype=np.float64).reshape(3, 3))

    def TRobotBase(self):
        """(position, rotation) of the base in the world"""
        return getattr(self, "_base", (np.zeros(3), np.eye(3)))

    def setQ(self, q):
        self._q = q
        self._push()

    def setDq(self, dq):
        self._dq = dq
        self._push()

    def q(self):
        return self._q

    def dq(self):
        return self._dq

    def updateModel(self):
        """kept for call-order compatibility: the model update is fused into the tick kernel"""
        self._push()

    def _push(self):
        if self._controller is not None:
            self._controller._ctrl.set_state(self._q, self._dq)
        for own in self._standalone:
            own._ctrl.set_state(self._q, self._dq)


class _StandaloneOwner:
    """What a task that is not attached to a RobotController runs on: a context holding that one task
    (sai2b.h "task-level plugin interface"). Created on the first call that needs the device."""

    def __init__(self, robot, cfg, q_construction):
        self._ctrl = Controller(robot.model, [cfg], robot.batch, robot.device)
        # the reference constructs a task at the model's state of that moment (goals := current pose)
        self._ctrl.set_state(q_construction, np.zeros_like(q_construction))
        self._ctrl.reinitialize()
        self._ctrl.set_state(robot.q(), robot.dq())
        robot._standalone.append(self)


class _InternalOtgView:
    """read-only side of the task's OTG_joints / OTG_6dof_cartesian object (JointTask.h:324 `getInternalOtg()`,
    OTG_joints.h:110-164, OTG_6dof_cartesian.h:171-243): one answer per robot"""

    def __init__(self, task):
        self._t = task

    def isGoalReached(self):
        rc, idx = self._t._require_owner()
        return rc._ctrl.get_otg_status(idx)[0] != 0

    def getJerkLimitEnabled(self):
        return bool(self._t._cfg.use_internal_otg and self._t._cfg.internal_otg_jerk_limited)

    def getNextPosition(self):
        return self._t.getDesiredPosition()

    def getNextVelocity(self):
        return self._t.getDesiredVelocity()

    def getNextAcceleration(self):
        return self._t.getDesiredAcceleration()

    def getNextOrientation(self):
        return self._t.getDesiredOrientation()

    def getNextLinearVelocity(self):
        return self._t.getDesiredLinearVelocity()

    def getNextAngularVelocity(self):
        return self._t.getDesiredAngularVelocity()

    def getNextLinearAcceleration(self):
        return self._t.getDesiredLinearAcceleration()

    def getNextAngularAcceleration(self):
        return self._t.getDesiredAngularAcceleration()


class _TaskBase:
    def __init__(self, robot, cfg):
        self._robot = robot
        self._cfg = cfg
        self._owner = None  # (RobotController, index) once attached; (_StandaloneOwner, 0) when driven on its own
        self._pending = {}
        self._q_construction = np.array(robot.q(), dtype=np.float64, copy=True) if not hasattr(robot.q(), "data_ptr") else robot.q().cpu().numpy()
        self._task_level = False  # the last model update came through updateTaskModel()

    # TemplateTask accessors (reference src/tasks/TemplateTask.h:95-115)
    def getInternalOtg(self):
        return _InternalOtgView(self)

    def getTaskName(self):
        return self._cfg.name.decode()

    def getTaskType(self):
        return self._cfg.type

    def getLoopTimestep(self):
        return self._cfg.loop_timestep

    def getConstRobotModel(self):
        return self._robot

    def setDynamicDecouplingType(self, t):
        self._cfg.dynamic_decoupling_type = int(t)
        self._sync_cfg()

    def setBoundedInertiaEstimateThreshold(self, thr):
        self._cfg.bie_threshold = max(float(thr), 0.0)  # JointTask.h:369-375
        self._sync_cfg()

    def _sync_cfg(self):
        if self._owner:
            rc, idx = self._owner
            rc._ctrl.update_task_config(idx, self._cfg)

    def _goal(self, key, value, rows):
        B = self._robot.batch
        if not hasattr(value, "data_ptr"):
            value = np.ascontiguousarray(value, dtype=np.float64)
            if value.shape != (rows, B):
                raise ValueError(f"goal must have shape ({rows}, {B})")
        self._pending[key] = value
        self._flush()

    # ---- the TemplateTask virtuals (reference src/tasks/TemplateTask.h:42-88): a task driven on its own, the
    # caller chaining the nullspaces as in examples/04-task_and_redundancy.cpp:141-150,188-189
    def updateTaskModel(self, N_prec=None):
        """N_prec [n*n][B] (row-major inside the component index), None = identity"""
        rc, idx = self._require_owner()
        rc._ctrl.task_update_model(idx, N_prec)
        self._task_level = True

    def computeTorques(self, tau_prec=None, out=None):
        """computeTorques() / computeTorques(tau_prec): this task's torques [n][B]"""
        rc, idx = self._require_owner()
        return rc._ctrl.task_compute_torques(idx, tau_prec, out)

    def reInitializeTask(self):
        rc, idx = self._require_owner()
        rc._ctrl.task_reinitialize(idx)

    def getTaskNullspace(self):
        rc, idx = self._require_owner()
        return rc._ctrl.task_nullspaces(idx)[0]

    def getPreviousTasksNullspace(self):
        rc, idx = self._require_owner()
        return rc._ctrl.task_nullspaces(idx)[1]

    def getTaskAndPreviousNullspace(self):
        rc, idx = self._require_owner()
        if self._task_level:
            return rc._ctrl.task_nullspaces(idx)[2]
        return rc._ctrl.get_task_nullspace(idx)  # of the controller's last tick (introspection on)

    def _require_owner(self):
        if not self._owner:
            self._owner = (_StandaloneOwner(self._robot, self._cfg, self._q_construction), 0)
            self._flush()
        return self._owner


class JointTask(_TaskBase):
    """reference src/tasks/JointTask.h:56-75 (ctors), :137-179 (goals), :234-259 (gains)"""

    def __init__(self, robot, joint_selection_matrix=None, task_name="joint_task", loop_timestep=0.001):
        cfg = joint_task_config(task_name, joint_selection_matrix, internal_otg=True, robot_dof=robot.dof())  # JointTask.h:38
        cfg.loop_timestep = loop_timestep
        super().__init__(robot, cfg)

    def isFullJointTask(self):
        return self._cfg.task_dof == self._robot.dof()

    def setGoalPosition(self, q):
        self._goal("q", q, self._cfg.task_dof)

    def setGoalVelocity(self, dq):
        self._goal("dq", dq, self._cfg.task_dof)

    def setGoalAcceleration(self, ddq):
        self._goal("ddq", ddq, self._cfg.task_dof)

    def setGains(self, kp, kv, ki=0.0, _checked=True):
        """JointTask.h:225-257 (JointTask.cpp:136-205): scalars / size-1 vectors = isotropic, else one gain per task
        coordinate; vectors are rejected only when every entry is negative (JointTask.cpp:171)"""
        k0 = self._cfg.task_dof
        a = [np.atleast_1d(np.asarray(x, dtype=float)) for x in (kp, kv, ki)]
        iso = all(x.size == 1 for x in a)
        if not iso:
            a[2] = np.zeros(k0) if np.ndim(ki) == 0 and ki == 0.0 else a[2]  # the two-vector overload: ki = 0
            if any(x.size != k0 for x in a):
                raise ValueError("size of gain vectors inconsistent with number of task dofs in JointTask::setGains\n")
        if _checked and (any(x[0] < 0 for x in a) if iso else any(x.max() < 0 for x in a)):
            raise ValueError("gains must be positive or zero in JointTask::setGains\n")
        kp, kv, ki = (np.broadcast_to(x, (k0,)) for x in a)
        for i in range(k0):
            self._cfg.kp[i], self._cfg.kv[i], self._cfg.ki[i] = kp[i], kv[i], ki[i]
        if not _checked:
            self._cfg.unsafe_motion_gains = 1
        self._sync_cfg()

    def setGainsUnsafe(self, kp, kv, ki):
        """JointTask.h:256: no sign check"""
        self.setGains(kp, kv, ki, _checked=False)

    def getJointSelectionMatrix(self):
        """JointTask.h:120: [task_dof, dof]"""
        n = self._robot.dof()
        return np.array(self._cfg.joint_selection[: self._cfg.task_dof * n]).reshape(self._cfg.task_dof, n)

    def getVelocitySaturationMaxVelocity(self):
        """JointTask.h:352"""
        return np.array(self._cfg.saturation_velocity[: self._cfg.task_dof])

    def enableVelocitySaturation(self, saturation_velocity):
        v = np.broadcast_to(np.asarray(saturation_velocity, dtype=float), (self._cfg.task_dof,))
        if v.min() <= 0:
            raise ValueError("saturation velocity must be positive in JointTask::enableVelocitySaturation\n")
        self._cfg.use_velocity_saturation = 1
        for i in range(self._cfg.task_dof):
            self._cfg.saturation_velocity[i] = v[i]
        self._sync_cfg()

    def disableVelocitySaturation(self):
        self._cfg.use_velocity_saturation = 0
        self._sync_cfg()

    def getGoalPosition(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_jt_goals(idx)[0]

    def getGoalVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_jt_goals(idx)[1]

    def getGoalAcceleration(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_jt_goals(idx)[2]

    def getCurrentPosition(self):
        """S q (JointTask.h:130), from the controller's state buffer"""
        rc, idx = self._require_owner()
        S = np.array(self._cfg.joint_selection[: self._cfg.task_dof * self._robot.dof()]).reshape(self._cfg.task_dof, self._robot.dof())
        return S @ rc._ctrl.get_state()[0]

    def getCurrentVelocity(self):
        rc, idx = self._require_owner()
        S = np.array(self._cfg.joint_selection[: self._cfg.task_dof * self._robot.dof()]).reshape(self._cfg.task_dof, self._robot.dof())
        return S @ rc._ctrl.get_state()[1]

    def getGains(self):
        """one (kp, kv, ki) when the gains are isotropic, one per task coordinate otherwise (JointTask.cpp:207-216)"""
        k0 = self._cfg.task_dof
        g = [(self._cfg.kp[i], self._cfg.kv[i], self._cfg.ki[i]) for i in range(k0)]
        return g[:1] if all(x == g[0] for x in g) else g

    def getTaskDof(self):
        return self._cfg.task_dof

    def getVelocitySaturationEnabled(self):
        return bool(self._cfg.use_velocity_saturation)

    def getBoundedInertiaEstimateThreshold(self):
        return self._cfg.bie_threshold

    def resetIntegrators(self):
        rc, idx = self._require_owner()
        rc._ctrl.reset_integrators(idx)

    def enableInternalOtgAccelerationLimited(self, max_velocity, max_acceleration):
        """JointTask.cpp:360-381 (scalars or per-task-dof vectors)"""
        k0 = self._cfg.task_dof
        v, a = (np.broadcast_to(np.asarray(x, dtype=float), (k0,)) for x in (max_velocity, max_acceleration))
        if v.min() <= 0:
            raise ValueError("max velocity cannot be 0 or negative in any directions in OTG_joints::setMaxVelocity\n")
        if a.min() <= 0:
            raise ValueError("max acceleration cannot be 0 or negative in any directions in OTG_joints::setMaxAcceleration\n")
        for i in range(k0):
            self._cfg.otg_max_velocity[i], self._cfg.otg_max_acceleration[i] = v[i], a[i]
        self._cfg.use_internal_otg, self._cfg.internal_otg_jerk_limited = 1, 0
        self._sync_cfg()

    def enableInternalOtgJerkLimited(self, max_velocity, max_acceleration, max_jerk):
        """JointTask.cpp:383-406 (scalars or per-task-dof vectors): ruckig's jerk-limited interface"""
        k0 = self._cfg.task_dof
        v, a, j = (np.broadcast_to(np.asarray(x, dtype=float), (k0,)) for x in (max_velocity, max_acceleration, max_jerk))
        if v.min() <= 0:
            raise ValueError("max velocity cannot be 0 or negative in any directions in OTG_joints::setMaxVelocity\n")
        if a.min() <= 0:
            raise ValueError("max acceleration cannot be 0 or negative in any directions in OTG_joints::setMaxAcceleration\n")
        if j.min() <= 0:
            raise ValueError("max jerk cannot be 0 or negative in any directions in OTG_joints::setMaxJerk\n")
        for i in range(k0):
            self._cfg.otg_max_velocity[i], self._cfg.otg_max_acceleration[i], self._cfg.otg_max_jerk[i] = v[i], a[i], j[i]
        self._cfg.use_internal_otg, self._cfg.internal_otg_jerk_limited = 1, 1
        self._sync_cfg()

    def disableInternalOtg(self):
        """JointTask.h:320: desired state = goal state"""
        self._cfg.use_internal_otg = 0
        self._sync_cfg()

    def getInternalOtgEnabled(self):
        return bool(self._cfg.use_internal_otg)

    def getDesiredPosition(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_jt_desired(idx)[0]

    def getDesiredVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_jt_desired(idx)[1]

    def getDesiredAcceleration(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_jt_desired(idx)[2]

    def _flush(self):
        if self._owner and self._pending:
            rc, idx = self._owner
            rc._ctrl.set_jt_goals(idx, self._pending.get("q"), self._pending.get("dq"), self._pending.get("ddq"))
            self._pending = {}


class MotionForceTask(_TaskBase):
    """reference src/tasks/MotionForceTask.h:96-110 (ctors), :211-247 (goals), :272-328 (gains),
    :576-623 (force space)"""

    def __init__(self, robot, link=EE_LINK, compliant_frame_pos=EE_FRAME_POS, compliant_frame_rot=None,
                 controlled_directions_translation=None, controlled_directions_rotation=None,
                 task_name=None, is_force_motion_parametrization_in_compliant_frame=False, loop_timestep=0.001):
        partial = None
        if controlled_directions_translation is not None or controlled_directions_rotation is not None:
            partial = (
                np.zeros((0, 3)) if controlled_directions_translation is None else controlled_directions_translation,
                np.zeros((0, 3)) if controlled_directions_rotation is None else controlled_directions_rotation,
            )
        if isinstance(link, str):  # a link NAME as in the reference: needs a robot built from a URDF file
            if robot.links is None:
                raise ValueError("link names need a robot model built from a URDF file")
            link, compliant_frame_pos, compliant_frame_rot = resolve_link_frame(robot.links, link, compliant_frame_pos, compliant_frame_rot)
        cfg = motion_force_task_config(task_name, link, compliant_frame_pos, compliant_frame_rot, partial,
                                       internal_otg=True, robot_dof=robot.dof())  # MotionForceTask.h:67
        cfg.parametrization_in_compliant_frame = int(is_force_motion_parametrization_in_compliant_frame)
        cfg.loop_timestep = loop_timestep
        super().__init__(robot, cfg)

    def setGoalPosition(self, x):
        self._goal("pos", x, 3)

    def setGoalOrientation(self, R):
        self._goal("rot", R, 9)

    def setGoalLinearVelocity(self, v):
        self._goal("v", v, 3)

    def setGoalAngularVelocity(self, w):
        self._goal("w", w, 3)

    def setGoalLinearAcceleration(self, a):
        self._goal("a", a, 3)

    def setGoalAngularAcceleration(self, a):
        self._goal("alpha", a, 3)

    def setGoalForce(self, f):
        self._goal("f", f, 3)

    def setGoalMoment(self, m):
        self._goal("m", m, 3)

    def updateSensedForceAndMoment(self, f, m):
        self._goal("sf", f, 3)
        self._goal("sm", m, 3)
        self._sensed_sensor = (f, m)

    def getSensedForceSensor(self):
        """MotionForceTask.h:173,183: the last sensor-frame readings handed to updateSensedForceAndMoment"""
        return getattr(self, "_sensed_sensor", (np.zeros((3, self._robot.batch)),) * 2)[0]

    def getSensedMomentSensor(self):
        return getattr(self, "_sensed_sensor", (np.zeros((3, self._robot.batch)),) * 2)[1]

    def _set3(self, names, values, where):
        vals = [np.broadcast_to(np.asarray(v, dtype=float), (3,)) for v in values]
        if min(v.min() for v in vals) < 0:
            raise ValueError(f"all gains should be positive or zero in MotionForceTask::{where}\n")
        for n, v in zip(names, vals):
            for i in range(3):
                getattr(self._cfg, n)[i] = v[i]
        self._sync_cfg()

    def setPosControlGains(self, kp, kv, ki=0.0):
        self._set3(("kp_pos", "kv_pos", "ki_pos"), (kp, kv, ki), "setPosControlGains")

    def setOriControlGains(self, kp, kv, ki=0.0):
        self._set3(("kp_ori", "kv_ori", "ki_ori"), (kp, kv, ki), "setOriControlGains")

    def setForceControlGains(self, kp, kv, ki):
        self._set3(("kp_force", "kv_force", "ki_force"), (kp, kv, ki), "setForceControlGains")

    def setMomentControlGains(self, kp, kv, ki):
        self._set3(("kp_moment", "kv_moment", "ki_moment"), (kp, kv, ki), "setMomentControlGains")

    def _parametrize(self, dim_field, axis_field, dim, axis, dim_msg, axis_msg):
        """-> whether the space changed (`reset` of MotionForceTask.cpp:838-848); the library then moves that half of
        the goal to the current pose, restarts that half of the internal OTG there and resets its integrators"""
        if not 0 <= dim <= 3:
            raise ValueError(dim_msg)
        old_dim, old_axis = getattr(self._cfg, dim_field), np.array(getattr(self._cfg, axis_field)[:])
        if dim in (1, 2):
            a = np.asarray(axis, dtype=float)
            if np.linalg.norm(a) < 1e-2:
                raise ValueError(axis_msg)
            a = a / np.linalg.norm(a)
            for i in range(3):
                getattr(self._cfg, axis_field)[i] = a[i]
        setattr(self._cfg, dim_field, int(dim))
        self._sync_cfg()
        if dim != old_dim:
            return True
        if dim not in (1, 2):
            return False
        new_axis = np.array(getattr(self._cfg, axis_field)[:])
        return not (np.sum((new_axis - old_axis) ** 2) <= 1e-24 * min(np.sum(new_axis ** 2), np.sum(old_axis ** 2)))

    def parametrizeForceMotionSpaces(self, force_space_dimension, axis=(0, 0, 1)):
        return self._parametrize("force_space_dimension", "force_axis", force_space_dimension, axis,
                                 "Force space dimension should be between 0 and 3 in MotionForceTask::parametrizeForceMotionSpaces\n",
                                 "Force or motion axis should be a non singular vector in MotionForceTask::parametrizeForceMotionSpaces\n")

    def parametrizeMomentRotMotionSpaces(self, moment_space_dimension, axis=(0, 0, 1)):
        return self._parametrize("moment_space_dimension", "moment_axis", moment_space_dimension, axis,
                                 "Moment space dimension should be between 0 and 3 in MotionForceTask::parametrizeMomentRotMotionSpaces\n",
                                 "Moment or rot motion axis should be a non singular vector in MotionForceTask::parametrizeMomentRotMotionSpaces\n")

    def setClosedLoopForceControl(self, on=True):
        self._cfg.closed_loop_force = int(on)
        self._sync_cfg()

    def enablePassivity(self):
        self._cfg.passivity_enabled = 1
        self._sync_cfg()

    def disablePassivity(self):
        self._cfg.passivity_enabled = 0
        self._sync_cfg()

    def setClosedLoopMomentControl(self, on=True):
        self._cfg.closed_loop_moment = int(on)
        self._sync_cfg()

    def enableVelocitySaturation(self, linear_vel_sat=0.3, angular_vel_sat=np.pi / 3):
        if linear_vel_sat <= 0 or angular_vel_sat <= 0:
            raise ValueError("Velocity saturation values should be strictly positive or zero in MotionForceTask::enableVelocitySaturation\n")
        self._cfg.use_velocity_saturation = 1
        self._cfg.linear_saturation_velocity, self._cfg.angular_saturation_velocity = linear_vel_sat, angular_vel_sat
        self._sync_cfg()

    def disableVelocitySaturation(self):
        self._cfg.use_velocity_saturation = 0
        self._sync_cfg()

    def enableInternalOtgAccelerationLimited(self, max_linear_velocity, max_linear_acceleration, max_angular_velocity,
                                             max_angular_acceleration):
        """MotionForceTask.cpp:511-523"""
        c = self._cfg
        if min(max_linear_velocity, max_angular_velocity) <= 0:
            raise ValueError("max velocity set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxLinearVelocity\n")
        if min(max_linear_acceleration, max_angular_acceleration) <= 0:
            raise ValueError("max acceleration set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxLinearAcceleration\n")
        c.otg_max_linear_velocity, c.otg_max_linear_acceleration = float(max_linear_velocity), float(max_linear_acceleration)
        c.otg_max_angular_velocity, c.otg_max_angular_acceleration = float(max_angular_velocity), float(max_angular_acceleration)
        c.use_internal_otg, c.internal_otg_jerk_limited = 1, 0
        self._sync_cfg()

    def enableInternalOtgJerkLimited(self, max_linear_velocity, max_linear_acceleration, max_linear_jerk, max_angular_velocity,
                                     max_angular_acceleration, max_angular_jerk):
        """MotionForceTask.cpp:525-538: ruckig's jerk-limited interface"""
        c = self._cfg
        if min(max_linear_velocity, max_angular_velocity) <= 0:
            raise ValueError("max velocity set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxLinearVelocity\n")
        if min(max_linear_acceleration, max_angular_acceleration) <= 0:
            raise ValueError("max acceleration set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxLinearAcceleration\n")
        if min(max_linear_jerk, max_angular_jerk) <= 0:
            raise ValueError("max jerk set to 0 or negative value in some directions in OTG_6dof_cartesian::setMaxJerk\n")
        c.otg_max_linear_velocity, c.otg_max_linear_acceleration = float(max_linear_velocity), float(max_linear_acceleration)
        c.otg_max_angular_velocity, c.otg_max_angular_acceleration = float(max_angular_velocity), float(max_angular_acceleration)
        c.otg_max_linear_jerk, c.otg_max_angular_jerk = float(max_linear_jerk), float(max_angular_jerk)
        c.use_internal_otg, c.internal_otg_jerk_limited = 1, 1
        self._sync_cfg()

    def disableInternalOtg(self):
        """MotionForceTask.h:423: desired state = goal state"""
        self._cfg.use_internal_otg = 0
        self._sync_cfg()

    def getInternalOtgEnabled(self):
        return bool(self._cfg.use_internal_otg)

    def getDesiredPosition(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_desired(idx)[0]

    def getDesiredOrientation(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_desired(idx)[1]

    def getDesiredLinearVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_desired(idx)[2]

    def getDesiredAngularVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_desired(idx)[3]

    def getDesiredLinearAcceleration(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_desired(idx)[4]

    def getDesiredAngularAcceleration(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_desired(idx)[5]

    def setSingularityHandlingBounds(self, s_min, s_max):
        self._cfg.s_min, self._cfg.s_max = float(s_min), float(s_max)
        self._sync_cfg()

    def _status(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_status(idx)

    def getCurrentPosition(self):
        return self._status()["pos"]

    def getCurrentOrientation(self):
        return self._status()["rot"]

    def getSensedForceControlWorldFrame(self):
        return self._status()["sensed_force"]

    def getSensedMomentControlWorldFrame(self):
        return self._status()["sensed_moment"]

    def getPositionError(self):
        return self._status()["pos_error"]

    def getOrientationError(self):
        return self._status()["ori_error"]

    def goalPositionReached(self, tolerance):
        """per robot (MotionForceTask.cpp:548-563)"""
        return self._status()["pos_error_norm"] < tolerance

    def goalOrientationReached(self, tolerance):
        return self._status()["ori_error_norm"] < tolerance

    def getGoalPosition(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_goals(idx)[0]

    def getGoalOrientation(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_goals(idx)[1]

    def _world_wrench(self, which):
        """MotionForceTask.cpp:755-769: a goal given in the compliant frame comes back in the WORLD frame"""
        rc, idx = self._require_owner()
        g = rc._ctrl.get_mft_goals(idx)[which]
        if not self._cfg.parametrization_in_compliant_frame:
            return g
        R = self._status()["rot"].T.reshape(-1, 3, 3)
        return np.ascontiguousarray(np.einsum("bij,jb->ib", R, g))

    def getGoalForce(self):
        return self._world_wrench(6)

    def getGoalMoment(self):
        return self._world_wrench(7)

    def getGoalLinearVelocity(self):
        """MotionForceTask.h:224-247"""
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_goals(idx)[2]

    def getGoalAngularVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_goals(idx)[3]

    def getGoalLinearAcceleration(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_goals(idx)[4]

    def getGoalAngularAcceleration(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_goals(idx)[5]

    def getLinearSaturationVelocity(self):
        """MotionForceTask.h:437-440"""
        return self._cfg.linear_saturation_velocity

    def getAngularSaturationVelocity(self):
        return self._cfg.angular_saturation_velocity

    def posSelectionProjector(self):
        """MotionForceTask.h:653-659: 3 x 3 diagonal blocks of the partial-task projection"""
        return np.array(self._cfg.partial_projection[:]).reshape(6, 6)[:3, :3].copy()

    def oriSelectionProjector(self):
        return np.array(self._cfg.partial_projection[:]).reshape(6, 6)[3:, 3:].copy()

    def setSingularityHandlingGains(self, kp_type_1, kv_type_1, kv_type_2):
        """MotionForceTask.h:748-753"""
        self._cfg.kp_type_1, self._cfg.kv_type_1, self._cfg.kv_type_2 = float(kp_type_1), float(kv_type_1), float(kv_type_2)
        self._sync_cfg()

    def enableSingularityHandling(self):
        self._cfg.enforce_handling_strategy = 1
        self._sync_cfg()

    def disableSingularityHandling(self):
        self._cfg.enforce_handling_strategy = 0
        self._sync_cfg()

    def setSingularVectorSign(self, convention):
        """Not in the reference: which way classifySingularity perturbs along a singular vector
        (SingularityHandler.cpp:253-265 uses V_s as Eigen's JacobiSVD left it, a sign Eigen does not specify).
        0 = largest-magnitude component of V_s[:, i] positive (default), 1 = the opposite, 2 = type 1 if either
        direction moves the task, 3 = only if both do (enum sai2b_singular_vector_sign)."""
        self._cfg.singular_vector_sign = int(convention)
        self._sync_cfg()

    def setFeedforwardForceGain(self, kff):
        self._cfg.kff_force = float(kff)
        self._sync_cfg()

    def getFeedforwardForceGain(self):
        return self._cfg.kff_force

    def setFeedforwardmomentGain(self, kff):
        self._cfg.kff_moment = float(kff)
        self._sync_cfg()

    def getFeedforwardmomentGain(self):
        return self._cfg.kff_moment

    def setMaxForceControlFeedbackOutput(self, v):
        self._cfg.max_force_feedback = float(v)
        self._sync_cfg()

    def getMaxForceControlFeedbackOutput(self):
        return self._cfg.max_force_feedback

    def setMaxMomentControlFeedbackOutput(self, v):
        self._cfg.max_moment_feedback = float(v)
        self._sync_cfg()

    def getMaxMomentControlFeedbackOutput(self):
        return self._cfg.max_moment_feedback

    def setForceSensorFrame(self, *args):
        """MotionForceTask::setForceSensorFrame (MotionForceTask.cpp:794-803). Two forms:
        (link, sensor_pos_in_link, sensor_rot_in_link=None) — the reference's: link index or NAME, the sensor frame in
        that link, which must be the control frame's; _T_control_to_sensor = compliant_frame^-1 * transformation_in_link;
        (sensor_pos_in_control_frame, sensor_rot_in_control_frame=None) — _T_control_to_sensor given directly."""
        if isinstance(args[0], (int, np.integer, str)):
            link, ps = args[0], np.asarray(args[1], dtype=float)
            Rs = np.eye(3) if len(args) < 3 or args[2] is None else np.asarray(args[2], dtype=float).reshape(3, 3)
            if isinstance(link, str):
                if self._robot.links is None:
                    raise ValueError("link names need a robot model built from a URDF file")
                link, ps, Rs = resolve_link_frame(self._robot.links, link, ps, Rs)
            if int(link) != self._cfg.link:
                raise ValueError("The link to which is attached the sensor should be the same as the link to which is attached "
                                 "the control frame in MotionForceTask::setForceSensorFrame\n")
            Rc, pc = np.array(self._cfg.frame_rot[:]).reshape(3, 3), np.array(self._cfg.frame_pos[:])
            return self.setForceSensorFrame(Rc.T @ (np.asarray(ps) - pc), Rc.T @ Rs)
        sensor_pos_in_control_frame = args[0]
        sensor_rot_in_control_frame = args[1] if len(args) > 1 else None
        p = np.asarray(sensor_pos_in_control_frame, dtype=float)
        R = np.eye(3) if sensor_rot_in_control_frame is None else np.asarray(sensor_rot_in_control_frame, dtype=float)
        for i in range(3):
            self._cfg.sensor_pos[i] = p[i]
        for i in range(9):
            self._cfg.sensor_rot[i] = R.ravel()[i]
        self._sync_cfg()

    def getForceSpaceDimension(self):
        return self._cfg.force_space_dimension

    def getMomentSpaceDimension(self):
        return self._cfg.moment_space_dimension

    def getPosControlGains(self):
        return [(self._cfg.kp_pos[i], self._cfg.kv_pos[i], self._cfg.ki_pos[i]) for i in range(3)]

    def getOriControlGains(self):
        return [(self._cfg.kp_ori[i], self._cfg.kv_ori[i], self._cfg.ki_ori[i]) for i in range(3)]

    def getVelocitySaturationEnabled(self):
        return bool(self._cfg.use_velocity_saturation)

    def getBoundedInertiaEstimateThreshold(self):
        return self._cfg.bie_threshold

    def resetIntegrators(self):
        rc, idx = self._require_owner()
        rc._ctrl.reset_integrators(idx, 0)

    def resetIntegratorsLinear(self):
        rc, idx = self._require_owner()
        rc._ctrl.reset_integrators(idx, 1)

    def resetIntegratorsAngular(self):
        rc, idx = self._require_owner()
        rc._ctrl.reset_integrators(idx, 2)

    def getUnitMassForce(self):
        """MotionForceTask.h:266, of the last tick (introspection on)"""
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_task_forces(idx)[0]

    def getCurrentLinearVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_velocity(idx)[0]

    def getCurrentAngularVelocity(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_velocity(idx)[1]

    def sigmaForce(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_sigma(idx)[0]

    def sigmaPosition(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_sigma(idx)[1]

    def sigmaMoment(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_sigma(idx)[2]

    def sigmaOrientation(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_sigma(idx)[3]

    def setPosControlGainsUnsafe(self, kp, kv, ki):
        """MotionForceTask.h:283 (MotionForceTask.cpp: no sign check, per-axis values)"""
        for n, v in zip(("kp_pos", "kv_pos", "ki_pos"), (kp, kv, ki)):
            v = np.broadcast_to(np.asarray(v, dtype=float), (3,))
            for i in range(3):
                getattr(self._cfg, n)[i] = v[i]
        self._sync_cfg()

    def setOriControlGainsUnsafe(self, kp, kv, ki):
        for n, v in zip(("kp_ori", "kv_ori", "ki_ori"), (kp, kv, ki)):
            v = np.broadcast_to(np.asarray(v, dtype=float), (3,))
            for i in range(3):
                getattr(self._cfg, n)[i] = v[i]
        self._sync_cfg()

    def getForceControlGains(self):
        return [(self._cfg.kp_force[0], self._cfg.kv_force[0], self._cfg.ki_force[0])]

    def getMomentControlGains(self):
        return [(self._cfg.kp_moment[0], self._cfg.kv_moment[0], self._cfg.ki_moment[0])]

    def handleAllSingularitiesAsType1(self, flag=True):
        """MotionForceTask.h:697"""
        self._cfg.enforce_type_1_strategy = int(bool(flag))
        self._sync_cfg()

    def setType1Posture(self, q_des):
        """MotionForceTask.h:706; q_des [n][B]"""
        rc, idx = self._require_owner()
        rc._ctrl.set_mft_type1_posture(idx, q_des)

    def getForceMotionSingleAxis(self):
        return np.array(self._cfg.force_axis[:])

    def getMomentRotMotionSingleAxis(self):
        return np.array(self._cfg.moment_axis[:])

    def getSigmaValues(self):
        rc, idx = self._require_owner()
        return rc._ctrl.get_mft_singularity(idx)[0]

    def _flush(self):
        if self._owner and self._pending:
            rc, idx = self._owner
            p = self._pending
            if any(k in p for k in ("pos", "rot", "v", "w", "a", "alpha")):
                rc._ctrl.set_mft_goals(idx, p.get("pos"), p.get("rot"), p.get("v"), p.get("w"), p.get("a"), p.get("alpha"))
            if "f" in p or "m" in p:
                rc._ctrl.set_mft_goal_wrench(idx, p.get("f"), p.get("m"))
            if "sf" in p or "sm" in p:
                rc._ctrl.set_mft_sensed_wrench(idx, p.get("sf"), p.get("sm"))
            self._pending = {}


class RobotController:
    """reference src/RobotController.h:25-40: same ctor checks, same entry points, batched."""

    def __init__(self, robot, tasks, introspection=False):
        if len(tasks) == 0:
            raise ValueError("RobotController must have at least one task")
        for t in tasks:
            if t.getConstRobotModel() is not robot:
                raise ValueError("All tasks must have the same robot model in RobotController")
        self._robot = robot
        self._tasks = list(tasks)
        self._ctrl = Controller(robot.model, [t._cfg for t in tasks], robot.batch, robot.device, introspection)
        robot._controller = self
        self._ctrl.set_state(robot.q(), robot.dq())
        self._ctrl.reinitialize()  # tasks are constructed at the model's current state
        for i, t in enumerate(self._tasks):
            t._owner = (self, i)
            t._flush()

    def updateControllerTaskModels(self):
        self._ctrl.update_task_models()

    def computeControlTorques(self, out=None):
        return self._ctrl.compute_control_torques(True, out)

    def tick(self, out=None):
        """fused updateControllerTaskModels() + computeControlTorques(): one kernel launch"""
        return self._ctrl.tick(out)

    def enableGravityCompensation(self, enable=True):
        self._ctrl.enable_gravity_compensation(enable)

    def reinitializeTasks(self):
        self._ctrl.reinitialize()

    def getTaskNames(self):
        return [t.getTaskName() for t in self._tasks]

    def _by_name(self, name, kind, label):
        for t in self._tasks:
            if t.getTaskName() == name:
                if t.getTaskType() != kind:
                    raise ValueError(f"Task {name} is not a {label}, and cannot be casted as such in RobotController::GetTaskByName")
                return t
        raise ValueError(f"Task {name} not found in RobotController::GetTaskByName")

    def getJointTaskByName(self, name):
        return self._by_name(name, _abi.JOINT_TASK, "JointTask")

    def getMotionForceTaskByName(self, name):
        return self._by_name(name, MOTION_FORCE_TASK, "MotionForceTask")


class BatchedSimulation:
    """Stands in for Sai2Simulation in the examples' loops (examples/05-...cpp:215-236): rigid-body dynamics
    of the batch, state resident on the device. Without setJointTorques, integrate() consumes the torques
    of the controller's last computeControlTorques() / tick() without a host round trip."""

    def __init__(self, controller, timestep=0.001, substeps=1):
        if timestep <= 0 or substeps < 1:
            raise ValueError("simulation timestep must be positive")
        self._c, self._dt, self._substeps, self._gravity, self._tau = controller, float(timestep), int(substeps), False, None

    def setTimestep(self, dt):
        if dt <= 0:
            raise ValueError("simulation timestep must be positive")
        self._dt = float(dt)

    def enableGravity(self, on=True):
        self._gravity = bool(on)

    def setJointTorques(self, tau):
        self._tau = tau

    def integrate(self):
        self._c._ctrl.sim_step(self._tau, self._dt, self._substeps, self._gravity)
        self._tau = None

    def getJointPositions(self):
        return self._c._ctrl.get_state()[0]

    def getJointVelocities(self):
        return self._c._ctrl.get_state()[1]
