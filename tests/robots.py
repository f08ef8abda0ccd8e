"""Robot descriptions of the reference's examples other than the Panda, as NUMBERS (test fixtures) and a small URDF
writer of ours — the GPU box has no /root/reference. tests/test_urdf.py checks, where the reference tree is present,
that these load into the same model as the reference's own files:
  sliding_base : examples/06-partial_joint_task/panda_arm_sliding_base.urdf — the Panda (the library's own constants)
                 on a prismatic joint along y (effort 150, travel -1..1) under a 4 kg base link
  planar_4r    : examples/11-planar_robot_controller/rrrrbot.urdf — four 0.5 m links, 1 kg each, joints about z"""
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


TEXT = {"sliding_base": sliding_base_urdf, "planar_4r": planar_4r_urdf}
