/*
 * sai2b_detfk.h — bit-reproducible pose of a control frame from the joint positions.
 *
 * Why: a MotionForceTask's goals and its internal OTG start from the CURRENT pose of the control frame
 * (MotionForceTask::reInitializeTask, MotionForceTask.cpp:204-245; enableInternalOtg*, :511-523; the force / motion
 * space re-parametrisation, :830-890), and ruckig's synchronisation tests (calculator_target.hpp:123-200) compare
 * quantities against 4 * DBL_EPSILON: two correct forward kinematics that differ in the last bit start two
 * generators that can take different — equally valid — planner branches on a later re-plan. The product kernels and
 * the test oracle therefore compute THIS pose (and only this one: the torque path keeps each side's own
 * kinematics) with the same sequence of IEEE + and * and the sine / cosine of sai2b_detmath.h, no fused
 * multiply-add. Plain C, also valid HIP device code; contraction is switched off inside the functions, so the
 * translation unit's own setting does not matter.
 *
 * Chain convention as everywhere in this library (sai2b_robot_model, include/sai2b.h): joint i's frame is
 * E_i (constant rotation) at xyz_i in its parent link, then a rotation about (joint_type 0) or a slide along
 * (joint_type 1) its z by q_i.
 */
#ifndef SAI2B_DETFK_H_
#define SAI2B_DETFK_H_

#include "sai2b_detmath.h"

#if defined(__clang__)
#define SAI2B_DET_NO_CONTRACT _Pragma("clang fp contract(off)")
#else
#define SAI2B_DET_NO_CONTRACT /* gcc: the oracle is built with -ffp-contract=off */
#endif

/* C (3 x 3) = A B, every entry as (a0 b0 + a1 b1) + a2 b2 */
SAI2B_DET_FN void sai2b_det_mm3(const double* A, const double* B, double* C) {
	SAI2B_DET_NO_CONTRACT
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) C[3 * i + j] = (A[3 * i] * B[j] + A[3 * i + 1] * B[3 + j]) + A[3 * i + 2] * B[6 + j];
}

/* E: n x 9 joint rotations (row-major 3 x 3 each), xyz: n x 3 joint origins, joint_type: n entries; link: the link the
 * frame is attached to (0-based); frame_pos / frame_rot: the frame in that link. Out: x (3), R (9, row-major). */
SAI2B_DET_FN void sai2b_det_frame_pose(const double* E, const double* xyz, const int* joint_type, const double* q, int link,
										const double* frame_pos, const double* frame_rot, double* x, double* R) {
	SAI2B_DET_NO_CONTRACT
	double Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
	for (int i = 0; i <= link; i++) {
		double RE[9], p[3], s, c;
		const double* o = xyz + 3 * i;
		for (int r = 0; r < 3; r++) p[r] = ((Rp[3 * r] * o[0] + Rp[3 * r + 1] * o[1]) + Rp[3 * r + 2] * o[2]) + pp[r];
		sai2b_det_mm3(Rp, E + 9 * i, RE);
		if (joint_type[i] != 0) { /* prismatic: slides along the joint frame's z */
			for (int r = 0; r < 3; r++) {
				p[r] = p[r] + RE[3 * r + 2] * q[i];
				Rp[3 * r] = RE[3 * r], Rp[3 * r + 1] = RE[3 * r + 1], Rp[3 * r + 2] = RE[3 * r + 2];
			}
		} else {
			sai2b_det_sincos(q[i], &s, &c);
			for (int r = 0; r < 3; r++) {
				Rp[3 * r] = c * RE[3 * r] + s * RE[3 * r + 1];
				Rp[3 * r + 1] = c * RE[3 * r + 1] - s * RE[3 * r];
				Rp[3 * r + 2] = RE[3 * r + 2];
			}
		}
		for (int r = 0; r < 3; r++) pp[r] = p[r];
	}
	for (int r = 0; r < 3; r++) x[r] = ((Rp[3 * r] * frame_pos[0] + Rp[3 * r + 1] * frame_pos[1]) + Rp[3 * r + 2] * frame_pos[2]) + pp[r];
	sai2b_det_mm3(Rp, frame_rot, R);
}

#endif /* SAI2B_DETFK_H_ */
