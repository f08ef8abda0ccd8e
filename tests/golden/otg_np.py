"""numpy restatement of the sai2 OTG wrappers, driving the REAL ruckig core of the reference.

Generation-time only (runs in the build container, where `make -C oracle ref` has compiled the
reference's vendored ruckig sources into oracle/_ref/libruckig_ref.so). It is the independent
check of oracle/otg_oracle.c part 2: the wrappers (src/helper_modules/OTG_joints.cpp,
OTG_6dof_cartesian.cpp) need Eigen and cannot be built, so they are restated here in Python on top
of the reference's own trajectory generator, with scipy's rotation-vector conversions standing in
for Eigen's AngleAxisd.
"""
import ctypes as C
import os

import numpy as np
from scipy.spatial.transform import Rotation

_REF = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "..", "oracle", "_ref", "libruckig_ref.so")
_dp = C.POINTER(C.c_double)

WORKING, FINISHED = 0, 1
SYNC_TIME, SYNC_PHASE = 0, 2


def ref_available():
    return os.path.exists(_REF)


def load_ref():
    L = C.CDLL(_REF)
    L.rref_create.restype = C.c_void_p
    L.rref_create.argtypes = [C.c_int, C.c_double]
    L.rref_destroy.argtypes = [C.c_void_p]
    L.rref_set_synchronization.argtypes = [C.c_void_p, C.c_int]
    L.rref_set_limits.argtypes = [C.c_void_p, _dp, _dp]
    L.rref_set_current.argtypes = [C.c_void_p, _dp, _dp, _dp]
    L.rref_set_target.argtypes = [C.c_void_p, _dp, _dp]
    L.rref_update.argtypes = [C.c_void_p]
    L.rref_update.restype = C.c_int
    L.rref_get_output.argtypes = [C.c_void_p, _dp, _dp, _dp, _dp, _dp, C.POINTER(C.c_int)]
    L.rref_calculate_and_sample.restype = C.c_int
    return L


def _p(a):
    return a.ctypes.data_as(_dp)


class RuckigRef:
    """Ruckig<DynamicDOFs> + InputParameter + OutputParameter of the reference, through ctypes."""

    def __init__(self, n, dt, lib=None):
        self.L = lib or load_ref()
        self.n = n
        self.h = self.L.rref_create(n, dt)
        self.L.rref_set_synchronization(self.h, SYNC_PHASE)
        # mirrors of the InputParameter / OutputParameter fields the wrappers touch
        self.cp, self.cv, self.ca = np.zeros(n), np.zeros(n), np.zeros(n)
        self.tp, self.tv = np.zeros(n), np.zeros(n)
        self.vmax, self.amax = np.ones(n), np.ones(n)
        self.np_, self.nv, self.na = np.zeros(n), np.zeros(n), np.zeros(n)

    def __del__(self):
        try:
            self.L.rref_destroy(self.h)
        except Exception:
            pass

    def update(self):
        """_otg->update(_input, _output)"""
        L, h = self.L, self.h
        L.rref_set_limits(h, _p(self.vmax), _p(self.amax))
        L.rref_set_current(h, _p(self.cp), _p(self.cv), _p(self.ca))
        L.rref_set_target(h, _p(self.tp), _p(self.tv))
        r = L.rref_update(h)
        if r in (WORKING, FINISHED):
            t, d, nc = C.c_double(), C.c_double(), C.c_int()
            L.rref_get_output(h, _p(self.np_), _p(self.nv), _p(self.na), C.byref(t), C.byref(d), C.byref(nc))
        return r

    def pass_to_input(self):
        self.cp, self.cv, self.ca = self.np_.copy(), self.nv.copy(), self.na.copy()


def is_approx(a, b, prec):
    """Eigen DenseBase::isApprox"""
    a, b = np.asarray(a, float).ravel(), np.asarray(b, float).ravel()
    return np.sum((a - b) ** 2) <= prec * prec * min(np.sum(a * a), np.sum(b * b))


class JointOTGNP:
    """OTG_joints (OTG_joints.h, OTG_joints.cpp)"""

    def __init__(self, x0, dt, lib=None):
        self.dim = len(x0)
        self.r = RuckigRef(self.dim, dt, lib)
        self.r.amax[:] = np.inf
        self.goal_reached = False
        self.target_set = False
        self.result = FINISHED
        self.reinitialize(x0)

    def reinitialize(self, x0):
        self.set_goal(x0, np.zeros(self.dim))
        r = self.r
        r.np_, r.nv, r.na = np.array(x0, float), np.zeros(self.dim), np.zeros(self.dim)
        r.pass_to_input()

    def set_limits(self, vmax, amax):
        self.r.vmax = np.broadcast_to(np.asarray(vmax, float), (self.dim,)).copy()
        self.r.amax = np.broadcast_to(np.asarray(amax, float), (self.dim,)).copy()

    def disable_jerk_limits(self):
        self.r.ca = np.zeros(self.dim)

    def set_goal(self, gp, gv):
        r = self.r
        if self.target_set and is_approx(gp, r.tp, 1e-12) and is_approx(gv, r.tv, 1e-12):
            return
        self.goal_reached = False
        self.target_set = True
        r.tp, r.tv = np.array(gp, float), np.array(gv, float)

    def update(self):
        if self.goal_reached:
            return
        r = self.r
        prev = (r.np_.copy(), r.nv.copy(), r.na.copy())
        self.result = r.update()
        if self.result == FINISHED:
            if np.linalg.norm(r.nv) < 1e-3:
                self.goal_reached = True
            else:  # intent of OTG_joints.cpp:129 (see oracle/otg_oracle.c header)
                self.set_goal(r.tp.copy(), np.zeros(self.dim))
            return
        if self.result == WORKING:
            r.pass_to_input()
            return
        r.np_, r.nv, r.na = prev
        r.cv, r.ca = np.zeros(self.dim), np.zeros(self.dim)

    def next(self):
        return self.r.np_.copy(), self.r.nv.copy(), self.r.na.copy()


class CartesianOTGNP:
    """OTG_6dof_cartesian (OTG_6dof_cartesian.h, OTG_6dof_cartesian.cpp)"""

    def __init__(self, pos, rot, dt, lib=None):
        self.r = RuckigRef(6, dt, lib)
        self.r.amax[:] = np.inf
        self.goal_reached = False
        self.result = FINISHED
        self.pos_set = self.ori_set = False
        self.ref = np.array(rot, float)
        self.goal_R = np.zeros((3, 3))
        self.goal_w = np.zeros(3)
        self.reinitialize(pos, rot)

    def set_limits(self, lv, la, av, aa):
        self.r.vmax = np.array([lv] * 3 + [av] * 3, float)
        self.r.amax = np.array([la] * 3 + [aa] * 3, float)

    def next_orientation(self):
        v = self.r.np_[3:]
        if np.linalg.norm(v) < 1e-3:
            return self.ref.copy()
        return self.ref @ Rotation.from_rotvec(v).as_matrix()

    def set_goal_position(self, gp, gv):
        r = self.r
        if self.pos_set and is_approx(gp, r.tp[:3], 1e-3) and is_approx(gv, r.tv[:3], 1e-3):
            return
        self.goal_reached = False
        self.pos_set = True
        r.tp[:3], r.tv[:3] = gp, gv

    def set_goal_orientation(self, gR, gw):
        r = self.r
        if self.ori_set and is_approx(self.goal_R, gR, 1e-3) and is_approx(self.goal_w, gw, 1e-3):
            return
        self.goal_reached = False
        self.ori_set = True
        new_ref = self.next_orientation()
        R_new_to_prev = new_ref.T @ self.ref
        self.ref = new_ref
        self.goal_R, self.goal_w = np.array(gR, float), np.array(gw, float)
        r.np_[3:] = 0
        r.nv[3:] = R_new_to_prev @ r.nv[3:]
        r.na[3:] = R_new_to_prev @ r.na[3:]
        r.pass_to_input()
        r.tp[3:] = Rotation.from_matrix(self.ref.T @ self.goal_R).as_rotvec()
        r.tv[3:] = self.ref.T @ self.goal_w

    def reinitialize(self, pos, rot):
        self.set_goal_position(pos, np.zeros(3))
        self.set_goal_orientation(rot, np.zeros(3))
        r = self.r
        r.cp, r.cv, r.ca = r.tp.copy(), np.zeros(6), np.zeros(6)
        r.np_, r.nv, r.na = r.tp.copy(), np.zeros(6), np.zeros(6)

    def update(self):
        if self.goal_reached:
            return
        r = self.r
        prev = (r.np_.copy(), r.nv.copy(), r.na.copy())
        self.result = r.update()
        if self.result == FINISHED:
            if np.linalg.norm(r.nv) < 1e-3:
                self.goal_reached = True
            else:
                self.set_goal_position(r.tp[:3].copy(), np.zeros(3))
                self.set_goal_orientation(self.goal_R.copy(), np.zeros(3))
            return
        if self.result == WORKING:
            r.pass_to_input()
            return
        r.np_, r.nv, r.na = prev
        r.cv, r.ca = np.zeros(6), np.zeros(6)

    def next(self):
        """position, orientation, linear/angular velocity, linear/angular acceleration"""
        r = self.r
        return (r.np_[:3].copy(), self.next_orientation(), r.nv[:3].copy(), self.ref @ r.nv[3:],
                r.na[:3].copy(), self.ref @ r.na[3:])
