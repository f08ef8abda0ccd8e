// sai2b_sim.hip — simulation harness: rigid-body forward dynamics of the batch,
// state resident in HBM between control ticks.
//
// The reference's examples close the loop through the external sai2-simulation
// (examples/05-...cpp:215-236: setJointTorques / integrate / getJointPositions); nothing of it is in
// the reference tree, so the dynamics here are the textbook ones and the integrator is defined by
// this file: the joint torques are held over the control period dt, which is split into `substeps`
// semi-implicit Euler steps   dq += h M(q)^-1 (tau - b(q, dq)),   q += h dq,
// with b = C(q, dq) dq (+ g(q) when asked) from a recursive Newton-Euler pass in world coordinates.
// One lane per robot, as everywhere in this library; FK and the mass matrix are the tick kernel's own.
#include <hip/hip_runtime.h>

#include "sai2b_device.hpp"
#include "sai2b_fast.hpp"
#include "sai2b_launch.h"

namespace sai2b {

// b(q, dq) = C dq (+ g): Newton-Euler with zero joint accelerations. World-frame angular velocity w,
// angular acceleration al, linear acceleration a of each joint-frame origin; forces F and moments Nn
// about the link COMs; backward accumulation to the joint axes.
template <class MD>
DI void bias_forces(const MD& md, const Frames& Fr, const real* dq, bool with_gravity, real* b) {
	real F[N][3], Nn[N][3], rc[N][3];
	real w[3] = {0, 0, 0}, al[3] = {0, 0, 0}, a[3] = {0, 0, 0}, o[3] = {0, 0, 0};
	if (with_gravity) {
		UNROLL for (int k = 0; k < 3; k++) a[k] = -md.gravity[k];
	}
	UNROLL for (int i = 0; i < N; i++) {
		const real* R = Fr.R[i];
		const real z[3] = {R[2], R[5], R[8]};
		real r[3], t1[3], t2[3], t3[3];
		UNROLL for (int k = 0; k < 3; k++) r[k] = Fr.p[i][k] - o[k];
		cross3(al, r, t1);
		cross3(w, r, t2);
		cross3(w, t2, t3);
		UNROLL for (int k = 0; k < 3; k++) a[k] += t1[k] + t3[k];
		real zq[3] = {z[0] * dq[i], z[1] * dq[i], z[2] * dq[i]};
		cross3(w, zq, t1);
		const bool pris = md.jtype[i] != 0;
		UNROLL for (int k = 0; k < 3; k++) {
			if (pris) {	 // sliding frame: Coriolis acceleration of its origin, no change of the angular motion
				a[k] += 2 * t1[k];
			} else {
				w[k] += zq[k];
				al[k] += t1[k];
			}
			o[k] = Fr.p[i][k];
		}
		UNROLL for (int k = 0; k < 3; k++)
			rc[i][k] = fma(R[3 * k], md.com[i][0], fma(R[3 * k + 1], md.com[i][1], R[3 * k + 2] * md.com[i][2]));
		real ac[3];
		cross3(al, rc[i], t1);
		cross3(w, rc[i], t2);
		cross3(w, t2, t3);
		UNROLL for (int k = 0; k < 3; k++) ac[k] = a[k] + t1[k] + t3[k];
		// I_world x = R I_link R^T x
		const real* li = md.inertia[i];
		real Il[9] = {li[0], li[3], li[4], li[3], li[1], li[5], li[4], li[5], li[2]};
		real u[3], Iu[3], Ial[3], Iw[3];
		mv_t<3, 3>(R, al, u);
		mv<3, 3>(Il, u, Iu);
		mv<3, 3>(R, Iu, Ial);
		mv_t<3, 3>(R, w, u);
		mv<3, 3>(Il, u, Iu);
		mv<3, 3>(R, Iu, Iw);
		cross3(w, Iw, t1);
		UNROLL for (int k = 0; k < 3; k++) {
			F[i][k] = md.mass[i] * ac[k];
			Nn[i][k] = Ial[k] + t1[k];
		}
	}
	real f[3] = {0, 0, 0}, n[3] = {0, 0, 0};
	UNROLL for (int i = N - 1; i >= 0; i--) {
		real t1[3];
		cross3(rc[i], F[i], t1);
		if (i + 1 < N) {
			real d[3] = {Fr.p[i + 1][0] - Fr.p[i][0], Fr.p[i + 1][1] - Fr.p[i][1], Fr.p[i + 1][2] - Fr.p[i][2]}, t2[3];
			cross3(d, f, t2);
			UNROLL for (int k = 0; k < 3; k++) n[k] += t2[k];
		}
		UNROLL for (int k = 0; k < 3; k++) {
			n[k] += Nn[i][k] + t1[k];
			f[k] += F[i][k];
		}
		const real* pr = (md.jtype[i] != 0) ? f : n;  // prismatic: the force along the axis
		b[i] = Fr.R[i][2] * pr[0] + Fr.R[i][5] * pr[1] + Fr.R[i][8] * pr[2];
	}
}

// q_keep != NULL: the joint positions as they are on entry are saved there first (the pose the tasks cached at
// their last torque computation: see q_pose in sai2b_host.cpp)
__global__ __launch_bounds__(64) void sim_kernel(const DevParams* __restrict__ Pp, const real* __restrict__ tau,
												 real dt, int substeps, int with_gravity, real* __restrict__ dbg_bias,
												 real* __restrict__ q_keep) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
	real q[N], dq[N], tq[N];
	UNROLL for (int i = 0; i < N; i++) {
		q[i] = ld(P.q, i, B, b);
		dq[i] = ld(P.dq, i, B, b);
		tq[i] = tau ? ld(tau, i, B, b) : 0.0;
		if (q_keep) st(q_keep, i, B, b, q[i]);
	}
	const real h = dt / substeps;
#pragma unroll 1
	for (int s = 0; s < substeps; s++) {
		Frames F;
		fk(P.model, q, F);
		real M[N * N], L[N * N], dinv[N], bias[N], x[N];
		bias_forces(P.model, F, dq, with_gravity != 0, bias);
		if (dbg_bias && s == 0) {
			UNROLL for (int i = 0; i < N; i++) st(dbg_bias, i, B, b, bias[i]);
		}
		mass_matrix(P.model, F, M);
		chol<N>(M, L, dinv);
		UNROLL for (int i = 0; i < N; i++) x[i] = tq[i] - bias[i];
		solve_lower<N>(L, dinv, x);
		solve_lower_t<N>(L, dinv, x);
		UNROLL for (int i = 0; i < N; i++) {
			dq[i] = fma(h, x[i], dq[i]);
			q[i] = fma(h, dq[i], q[i]);
		}
	}
	UNROLL for (int i = 0; i < N; i++) {
		st((real*)P.q, i, B, b, q[i]);
		st((real*)P.dq, i, B, b, dq[i]);
	}
}

// What the tasks' observers read between ticks (MotionForceTask.h:121-165: getCurrentPosition /
// Orientation, getSensedForce/MomentControlWorldFrame; MotionForceTask.cpp:540-579: getPositionError,
// getOrientationError, goalPositionReached, goalOrientationReached), computed from the state and goal
// buffers as they are now. out [26][B]: position 3, orientation 9, sensed force 3 and moment 3 in the
// world frame at the control point, sigma_position (goal - current) 3, sigma_orientation orientationError 3,
// and the two scalar norms sqrt(e^T sigma e) the goal...Reached tests compare with their tolerance.
__global__ __launch_bounds__(64) void mft_status_kernel(const DevParams* __restrict__ Pp, int task, real* __restrict__ out) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
	const DevTask& t = P.task[task];
	real q[N], dq[N];
	UNROLL for (int i = 0; i < N; i++) {
		q[i] = ld(P.q, i, B, b);
		dq[i] = ld(P.dq, i, B, b);
	}
	Frames F;
	fk(P.model, q, F);
	real x[3], R[9];
	frame_pose(t, F, x, R);
	{  // getCurrentLinearVelocity / getCurrentAngularVelocity (MotionForceTask.h:127-146, MotionForceTask.cpp:293-298)
		real J[6 * N], v[6];
		jacobian(P.model, t, F, x, J);
		mv<6, N>(J, dq, v);
		UNROLL for (int k = 0; k < 6; k++) st(out, 26 + k, B, b, v[k]);
	}
	real sf[9], sp[9], sm[9], so[9];
	sigma_pair(t, 0, t.fdim, t.faxis, R, sf, sp);
	sigma_pair(t, 1, t.mdim, t.maxis, R, sm, so);
	real gpos[3], grot[9], e[3], oe[3], se[3], soe[3];
	UNROLL for (int k = 0; k < 3; k++) gpos[k] = ld(t.goals, k, B, b);
	UNROLL for (int k = 0; k < 9; k++) grot[k] = ld(t.goals, 3 + k, B, b);
	UNROLL for (int k = 0; k < 3; k++) e[k] = gpos[k] - x[k];
	orientation_error(grot, R, oe);
	mv3(sp, e, se);
	mv3(so, oe, soe);
	// sensed wrench at the control point, world frame (MotionForceTask.cpp:805-828)
	real sfc[3], smc[3], fs_c[3], ms_c[3], tmp[3], fs_w[3], ms_w[3];
	UNROLL for (int k = 0; k < 3; k++) {
		sfc[k] = ld(t.sensed, k, B, b);
		smc[k] = ld(t.sensed, 3 + k, B, b);
	}
	mv3(t.sensor_rot, sfc, fs_c);
	mv3(t.sensor_rot, smc, ms_c);
	cross3(t.sensor_pos, fs_c, tmp);
	UNROLL for (int k = 0; k < 3; k++) ms_c[k] += tmp[k];
	mv3(R, fs_c, fs_w);
	mv3(R, ms_c, ms_w);
	UNROLL for (int k = 0; k < 3; k++) {
		st(out, k, B, b, x[k]);
		st(out, 12 + k, B, b, fs_w[k]);
		st(out, 15 + k, B, b, ms_w[k]);
		st(out, 18 + k, B, b, se[k]);
		st(out, 21 + k, B, b, soe[k]);
	}
	UNROLL for (int k = 0; k < 9; k++) {
		st(out, 3 + k, B, b, R[k]);
		// sigmaForce / sigmaPosition / sigmaMoment / sigmaOrientation (MotionForceTask.cpp:892-971)
		st(out, 32 + k, B, b, sf[k]);
		st(out, 41 + k, B, b, sp[k]);
		st(out, 50 + k, B, b, sm[k]);
		st(out, 59 + k, B, b, so[k]);
	}
	st(out, 24, B, b, sqrt(fmax(e[0] * se[0] + e[1] * se[1] + e[2] * se[2], 0.0)));
	st(out, 25, B, b, sqrt(fmax(oe[0] * soe[0] + oe[1] * soe[1] + oe[2] * soe[2], 0.0)));
}

}  // namespace sai2b

extern "C" int sai2b_launch_mft_status(const sai2b::DevParams* d_params, int B, int task, double* out, hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::mft_status_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params, task, out);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int sai2b_launch_sim(const sai2b::DevParams* d_params, int B, const double* tau, double dt, int substeps,
								int with_gravity, double* dbg_bias, double* q_keep, hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::sim_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params, tau, dt, substeps, with_gravity,
					   dbg_bias, q_keep);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}
