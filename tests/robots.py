"""Robot descriptions of the reference's examples other than the Panda, as NUMBERS (test fixtures) and a small URDF
writer of ours — the GPU box has no /root/reference. tests/test_urdf.py checks, where the reference tree is present,
that these load into the same model as the reference's own files:
  sliding_base : examples/06-partial_joint_task/panda_arm_sliding_base.urdf — the Panda (the library's own constants)
                 on a prismatic joint along y (effort 150, travel -1..1) under a 4 kg base link
  planar_4r    : examples/11-planar_robot_controller/rrrrbot.urdf — four 0.5 m links, 1 kg each, joints about z
  six_r        : a 6R arm of PUMA-like geometry (wrist links heavier than a PUMA's: a closed loop on 1e-5 kg m^2 inertias amplifies rounding by the period) (the robot of examples/01-joint_control; its URDF lives in sai2-model's
                 resources, outside the reference tree: geometry and inertias here are ours), joint axes about z, y, y,
                 x, y, x of the link frames — every non-z axis goes through the loader's axis folding"""
import sai2_primitives_perso_amd as pkg

PANDA_LINKS = [dict(m=3, c=(0, 0, -0.07), i=(0.3, 0.3, 0.3)), dict(m=3, c=(0, -0.1, 0), i=(0.3, 0.3, 0.3)),
               dict(m=2, c=(0.04, 0, -0.05), i=(0.2, 0.2, 0.2)), dict(m=2, c=(-0.04, 0.05, 0), i=(0.2, 0.2, 0.2)),
               dict(m=2, c=(0, 0, -0.15), i=(0.2, 0.2, 0.2)), dict(m=1.5, c=(0.06, 0, 0), i=(0.1, 0.1, 0.1)),
               dict(m=1.8, c=(0, 0, 0.17), i=(0.09, 0.05, 0.07))]


def _link(name, m, c, i):
    return (f'<link name="{name}"><inertial><origin xyz="{c[0]!r} {c[1]!r} {c[2]!r}" rpy="0 0 0"/><mass value="{m!r}"/>'
            f'<inertia ixx="{i[0]!r}" iyy="{i[1]!r}" izz="{i[2]!r}" ixy="0" ixz="0" iyz="0"/></inertial></link>')


def _joint(name, typ, parent, child, xyz, rpy, axis, lower, upper, effort):
    return (f'<joint name="{name}" type="{typ}"><origin rpy="{rpy[0]!r} {rpy[1]!r} {rpy[2]!r}" xyz="{xyz[0]!r} {xyz[1]!r} {xyz[2]!r}"/>'
            f'<parent link="{parent}"/><child link="{child}"/><axis xyz="{axis[0]} {axis[1]} {axis[2]}"/>'
            f'<limit effort="{effort!r}" lower="{lower!r}" upper="{upper!r}" velocity="2"/></joint>')


def sliding_base_urdf():
    """8 joints: prismatic base along y + the Panda's seven"""
    p = pkg.panda_model()
    out = ['<robot name="panda_sliding_base">', _link("slider_link", 4, (0, 0, 0.05), (0.4, 0.4, 0.4)), _link("link0", 4, (0, 0, 0.05), (0.4, 0.4, 0.4))]
    for k, r in enumerate(PANDA_LINKS):
        out.append(_link(f"link{k + 1}", r["m"], r["c"], r["i"]))
    out.append(_link("end-effector", 0.2, (0, 0, 0), (0.01, 0.01, 0.01)))
    out.append(_joint("joint0", "prismatic", "slider_link", "link0", (0, 0, 0), (0, 0, 0), (0, 1, 0), -1.0, 1.0, 150.0))
    for k in range(7):
        out.append(_joint(f"joint{k + 1}", "revolute", f"link{k}", f"link{k + 1}", list(p.joint_xyz[k]), list(p.joint_rpy[k]), (0, 0, 1),
                          p.q_lower[k], p.q_upper[k], p.effort[k]))
    out.append('<joint name="joint_ee" type="fixed"><origin rpy="0 0 0" xyz="0 0 0.15"/><parent link="link7"/><child link="end-effector"/></joint>')
    out.append("</robot>")
    return "\n".join(out)


def planar_4r_urdf():
    inertia = (0.084167, 0.083467, 0.000967)
    out = ['<robot name="RRRRBot">', _link("link0", 1, (0.0, 0.0, 0.0), inertia)]
    for k in range(1, 5):
        out.append(_link(f"link{k}", 1, (0.25, 0.0, 0.0), inertia))
    names = ["j0", "j1", "j3", "j4"]
    for k in range(4):
        out.append(_joint(names[k], "revolute", f"link{k}", f"link{k + 1}", (0.0 if k == 0 else 0.5, 0, 0), (0, 0, 0), (0, 0, 1), -2.9, 2.9, 176.0))
    out.append("</robot>")
    return "\n".join(out)


def six_r_urdf():
    links = [dict(m=12.0, c=(0, 0, 0.05), i=(0.4, 0.4, 0.35)), dict(m=17.4, c=(0.07, 0, 0.2), i=(0.13, 0.52, 0.54)),
             dict(m=4.8, c=(0.01, 0.02, 0.15), i=(0.066, 0.086, 0.0125)), dict(m=1.6, c=(0.02, 0, 0), i=(0.012, 0.009, 0.012)),
             dict(m=1.1, c=(0, 0.01, 0), i=(0.006, 0.008, 0.006)), dict(m=0.8, c=(0.03, 0, 0), i=(0.004, 0.006, 0.006))]
    xyz = [(0, 0, 0.67), (0, 0.15, 0), (0, 0, 0.43), (0.02, -0.02, 0.43), (0.1, 0, 0), (0.06, 0, 0)]
    rpy = [(0, 0, 0), (0, 0, 0), (0, 0.1, 0), (0, 0, 0), (0.05, 0, 0), (0, 0, 0)]
    axes = [(0, 0, 1), (0, 1, 0), (0, 1, 0), (1, 0, 0), (0, 1, 0), (1, 0, 0)]
    lim = [(-2.8, 2.8), (-3.9, 0.8), (-0.9, 3.9), (-1.9, 2.9), (-1.7, 1.7), (-4.6, 4.6)]
    eff = [97.6, 186.4, 89.4, 24.2, 20.1, 21.3]
    out = ['<robot name="six_r">', _link("base", 10, (0, 0, 0.3), (0.5, 0.5, 0.2))]
    for k, r in enumerate(links):
        out.append(_link(f"link{k + 1}", r["m"], r["c"], r["i"]))
    for k in range(6):
        out.append(_joint(f"j{k + 1}", "revolute", "base" if k == 0 else f"link{k}", f"link{k + 1}", xyz[k], rpy[k], axes[k], lim[k][0], lim[k][1], eff[k]))
    out.append("</robot>")
    return "\n".join(out)


TEXT = {"sliding_base": sliding_base_urdf, "planar_4r": planar_4r_urdf, "six_r": six_r_urdf}
