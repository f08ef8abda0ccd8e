"""An independent reading of a URDF for the tests (test infrastructure): xml.etree for the parsing, the URDF's own
semantics for the kinematics (joint origin, then a rotation about / translation along the joint's <axis>), numpy for
everything — nothing of the product's or the oracle's model code. Gives world poses of every link, the joint-space
mass matrix M = sum_k m_k Jv_k^T Jv_k + Jw_k^T I_k Jw_k and the gravity vector for a serial chain, fixed joints
included (their bodies move with the link they hang on)."""
import re
import xml.etree.ElementTree as ET

import numpy as np


def _nums(text, n, default):
    if text is None:
        return np.array(default, dtype=float)
    out = []
    for tok in text.split():
        m = re.match(r"[-+]?(\d+\.?\d*([eE][-+]?\d+)?|\.\d+([eE][-+]?\d+)?)", tok)  # numeric prefix, like std::stod
        out.append(float(m.group(0)))
    assert len(out) == n, text
    return np.array(out)


def _rpy(r):
    cr, sr, cp, sp, cy, sy = np.cos(r[0]), np.sin(r[0]), np.cos(r[1]), np.sin(r[1]), np.cos(r[2]), np.sin(r[2])
    return np.array([[cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr], [sy * cp, sy * sp * sr + cy * cr, sy * sp * cr - cy * sr],
                     [-sp, cp * sr, cp * cr]])


def _axis_rot(a, q):
    a = a / np.linalg.norm(a)
    K = np.array([[0, -a[2], a[1]], [a[2], 0, -a[0]], [-a[1], a[0], 0]])
    return np.eye(3) + np.sin(q) * K + (1 - np.cos(q)) * K @ K


class Chain:
    def __init__(self, path, is_file=True):
        root = ET.parse(path).getroot() if is_file else ET.fromstring(path)
        self.links = {}
        for l in root.findall("link"):
            inn = l.find("inertial")
            if inn is None:
                self.links[l.get("name")] = None
                continue
            o = inn.find("origin")
            I = inn.find("inertia")
            g = lambda k: float(I.get(k, "0")) if I is not None else 0.0
            Il = np.array([[g("ixx"), g("ixy"), g("ixz")], [g("ixy"), g("iyy"), g("iyz")], [g("ixz"), g("iyz"), g("izz")]])
            Ro = _rpy(_nums(o.get("rpy") if o is not None else None, 3, [0, 0, 0]))
            self.links[l.get("name")] = dict(m=float(inn.find("mass").get("value")), com=_nums(o.get("xyz") if o is not None else None, 3, [0, 0, 0]),
                                             I=Ro @ Il @ Ro.T)
        self.joints = []
        children = set()
        for j in root.findall("joint"):
            o, ax = j.find("origin"), j.find("axis")
            self.joints.append(dict(type=j.get("type"), parent=j.find("parent").get("link"), child=j.find("child").get("link"),
                                    xyz=_nums(o.get("xyz") if o is not None else None, 3, [0, 0, 0]),
                                    R=_rpy(_nums(o.get("rpy") if o is not None else None, 3, [0, 0, 0])),
                                    axis=_nums(ax.get("xyz") if ax is not None else None, 3, [1, 0, 0])))
            children.add(self.joints[-1]["child"])
        (self.root,) = [n for n in self.links if n not in children]
        # moving joints in chain order
        self.moving = []
        cur = [self.root]
        while cur:
            nxt = []
            for j in self.joints:
                if j["parent"] in cur:
                    if j["type"] != "fixed":
                        self.moving.append(j)
                    nxt.append(j["child"])
            cur = nxt
        self.dof = len(self.moving)

    def poses(self, q):
        """{link: (R, p, index of the last moving joint before it or -1)} and per moving joint (axis_world, origin_world, type)"""
        out = {self.root: (np.eye(3), np.zeros(3), -1)}
        jinfo = [None] * self.dof
        cur = [self.root]
        while cur:
            nxt = []
            for j in self.joints:
                if j["parent"] not in cur:
                    continue
                Rp, pp, mv = out[j["parent"]]
                R, p = Rp @ j["R"], pp + Rp @ j["xyz"]
                if j["type"] != "fixed":
                    k = self.moving.index(j)
                    a = j["axis"] / np.linalg.norm(j["axis"])
                    aw = R @ a
                    if j["type"] == "prismatic":
                        p = p + aw * q[k]
                    else:
                        R = R @ _axis_rot(a, q[k])
                    jinfo[k] = (aw, p.copy(), j["type"])
                    mv = k
                out[j["child"]] = (R, p, mv)
                nxt.append(j["child"])
            cur = nxt
        return out, jinfo

    def jacobian(self, q, link, point_in_link):
        pos, jinfo = self.poses(q)
        R, p, mv = pos[link]
        x = p + R @ np.asarray(point_in_link, dtype=float)
        J = np.zeros((6, self.dof))
        for k in range(mv + 1):
            aw, o, typ = jinfo[k]
            if typ == "prismatic":
                J[:3, k] = aw
            else:
                J[:3, k] = np.cross(aw, x - o)
                J[3:, k] = aw
        return J, x, R

    def mass_matrix_and_gravity(self, q, gravity=(0, 0, -9.81)):
        pos, _ = self.poses(q)
        M, g = np.zeros((self.dof, self.dof)), np.zeros(self.dof)
        for name, body in self.links.items():
            if body is None or pos[name][2] < 0:
                continue
            J, _, R = self.jacobian(q, name, body["com"])
            Iw = R @ body["I"] @ R.T
            M += body["m"] * J[:3].T @ J[:3] + J[3:].T @ Iw @ J[3:]
            g -= body["m"] * J[:3].T @ np.asarray(gravity, dtype=float)
        return M, g
