// sai2b_model_host.h — host-side model helpers shared by sai2b_host.cpp and the build-time generator of
// the baked Panda constants (tools/gen_baked_model.cpp), so that both produce bit-identical numbers.
#pragma once
#include <cmath>
#include <cstring>

#include "sai2b_params.h"

namespace sai2b {

inline void rot_from_rpy(const double* rpy, double* R) {
	const double cr = std::cos(rpy[0]), sr = std::sin(rpy[0]);
	const double cp = std::cos(rpy[1]), sp = std::sin(rpy[1]);
	const double cy = std::cos(rpy[2]), sy = std::sin(rpy[2]);
	const double Rm[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr,
						  sy * sp * cr - cy * sr, -sp,		cp * sr,				 cp * cr};
	std::memcpy(R, Rm, sizeof(Rm));
}
inline void sym3_from6(const double* v, double* I) {
	I[0] = v[0], I[4] = v[1], I[8] = v[2];
	I[1] = I[3] = v[3];
	I[2] = I[6] = v[4];
	I[5] = I[7] = v[5];
}

// Merge a fixed child body into link `link` (what RBDL does for URDF fixed joints)
inline void host_merge_fixed_body(sai2b_robot_model* md, int link, const double xyz[3], const double rpy[3], double mass,
								  const double com[3], const double inertia[6]) {
	double Rf[9], Ic[9], Iw[9] = {0};
	rot_from_rpy(rpy, Rf);
	sym3_from6(inertia, Ic);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++)
			for (int a = 0; a < 3; a++)
				for (int b = 0; b < 3; b++) Iw[3 * i + j] += Rf[3 * i + a] * Ic[3 * a + b] * Rf[3 * j + b];
	double c_child[3];
	for (int i = 0; i < 3; i++) c_child[i] = xyz[i] + Rf[3 * i] * com[0] + Rf[3 * i + 1] * com[1] + Rf[3 * i + 2] * com[2];
	const double m_parent = md->link_mass[link], m_total = m_parent + mass;
	double c_parent[3], c_new[3], I_parent[9], I_new[9];
	for (int i = 0; i < 3; i++) {
		c_parent[i] = md->link_com[link][i];
		c_new[i] = (m_parent * c_parent[i] + mass * c_child[i]) / m_total;
	}
	sym3_from6(md->link_inertia[link], I_parent);
	auto shifted = [&](const double* I, double m, const double* c, double* out) {
		const double d[3] = {c[0] - c_new[0], c[1] - c_new[1], c[2] - c_new[2]};
		const double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) out[3 * i + j] += I[3 * i + j] + m * ((i == j ? d2 : 0.0) - d[i] * d[j]);
	};
	for (int i = 0; i < 9; i++) I_new[i] = 0;
	shifted(I_parent, m_parent, c_parent, I_new);
	shifted(Iw, mass, c_child, I_new);
	md->link_mass[link] = m_total;
	for (int i = 0; i < 3; i++) md->link_com[link][i] = c_new[i];
	double* o = md->link_inertia[link];
	o[0] = I_new[0], o[1] = I_new[4], o[2] = I_new[8], o[3] = I_new[1], o[4] = I_new[2], o[5] = I_new[5];
}

// Panda arm constants (reference examples/15-haptic_control_impedance_type/panda_arm.urdf:4-184)
inline void host_panda_model(sai2b_robot_model* md) {
	constexpr int N = 7;
	std::memset(md, 0, sizeof(*md));
	md->dof = N;
	// examples/15-haptic_control_impedance_type/panda_arm.urdf:118-178 (joint origins / limits)
	struct Jt {
		double x, y, z, roll, lo, hi, effort;
	};
	const Jt joints[N] = {{0, 0, 0.333, 0, -2.8973, 2.8973, 87},
						  {0, 0, 0, -1.57079632679, -1.7628, 1.7628, 87},
						  {0, -0.316, 0, 1.57079632679, -2.8973, 2.8973, 87},
						  {0.0825, 0, 0, 1.57079632679, -3.0718, -0.0698, 87},
						  {-0.0825, 0.384, 0, -1.57079632679, -2.8973, 2.8973, 12},
						  {0, 0, 0, 1.57079632679, -0.0175, 3.7525, 12},
						  {0.088, 0, 0, 1.57079632679, -2.8973, 2.8973, 12}};
	// panda_arm.urdf:17-104 (link inertials: mass, COM, diagonal inertia)
	struct Lk {
		double m, cx, cy, cz, ixx, iyy, izz;
	};
	const Lk links[N] = {{3, 0, 0, -0.07, 0.3, 0.3, 0.3},	 {3, 0, -0.1, 0, 0.3, 0.3, 0.3},
						 {2, 0.04, 0, -0.05, 0.2, 0.2, 0.2}, {2, -0.04, 0.05, 0, 0.2, 0.2, 0.2},
						 {2, 0, 0, -0.15, 0.2, 0.2, 0.2},	 {1.5, 0.06, 0, 0, 0.1, 0.1, 0.1},
						 {1.8, 0, 0, 0.17, 0.09, 0.05, 0.07}};
	for (int i = 0; i < N; i++) {
		md->joint_xyz[i][0] = joints[i].x, md->joint_xyz[i][1] = joints[i].y, md->joint_xyz[i][2] = joints[i].z;
		md->joint_rpy[i][0] = joints[i].roll;
		md->q_lower[i] = joints[i].lo, md->q_upper[i] = joints[i].hi, md->effort[i] = joints[i].effort;
		md->link_mass[i] = links[i].m;
		md->link_com[i][0] = links[i].cx, md->link_com[i][1] = links[i].cy, md->link_com[i][2] = links[i].cz;
		md->link_inertia[i][0] = links[i].ixx, md->link_inertia[i][1] = links[i].iyy, md->link_inertia[i][2] = links[i].izz;
	}
	md->gravity[2] = -9.81;
	// fixed "end-effector" body (panda_arm.urdf:105-116) on joint_ee (:179-183)
	const double xyz[3] = {0, 0, 0.15}, zero[3] = {0, 0, 0}, inertia[6] = {0.01, 0.01, 0.01, 0, 0, 0};
	host_merge_fixed_body(md, 6, xyz, zero, 0.2, zero, inertia);
}

// constant part of the device parameter block from the C-ABI model
inline void host_fill_dev_model(const sai2b_robot_model& model, DevModel& dm) {
	for (int i = 0; i < N; i++) {
		dm.jtype[i] = model.joint_type[i];
		rot_from_rpy(model.joint_rpy[i], dm.E[i]);
		for (int k = 0; k < 3; k++) dm.xyz[i][k] = model.joint_xyz[i][k], dm.com[i][k] = model.link_com[i][k];
		for (int k = 0; k < 6; k++) dm.inertia[i][k] = model.link_inertia[i][k];
		dm.mass[i] = model.link_mass[i];
		dm.q_lower[i] = model.q_lower[i], dm.q_upper[i] = model.q_upper[i], dm.effort[i] = model.effort[i];
	}
	for (int k = 0; k < 3; k++) dm.gravity[k] = model.gravity[k];
}

}  // namespace sai2b
