// sai2b_otg.hip — kernels of the tasks' internal online trajectory generation (OTG).
//
// otg_kernel runs once per torque computation, before the tick kernel(s): for every task whose OTG
// is on it does what JointTask::computeTorques / MotionForceTask::computeTorques do at
// JointTask.cpp:313-320 / MotionForceTask.cpp:394-407 — setGoal...(goal), update(), read the next
// state — for one robot per lane, and writes that next state into the task's `otg_desired` buffer,
// which has the layout of the goals and is what the control law of the tick kernels then tracks.
// The generator itself is sai2b_otg_core.hpp. This file is compiled WITHOUT floating-point
// contraction (Makefile: -ffp-contract=off) so that the planner's accept/reject tests see the same
// arithmetic as the reference's CPU build.
//
// Memory: the state of one generator is OTG_ROWS doubles per robot (SoA rows of otg_state). A robot
// whose goal is reached and unchanged touches ~30 of them and writes nothing; a moving robot reads
// all and writes back the ~70 that change every tick (the trajectory rows only when it re-planned).
#include <hip/hip_runtime.h>

#include "sai2b_device.hpp"
#include "sai2b_launch.h"
#include "sai2b_otg_core.hpp"

namespace sai2b {
namespace {

using otg::Gen;
constexpr int MD = otg::MAXD;

DI void load7(const real* S, int row0, int n, int B, int b, double (&v)[MD]) {
	UNROLL for (int d = 0; d < MD; d++) v[d] = d < n ? ld(S, row0 + d, B, b) : 0.0;
}
DI void store7(real* S, int row0, int n, int B, int b, const double (&v)[MD]) {
	UNROLL for (int d = 0; d < MD; d++)
		if (d < n) st(S, row0 + d, B, b, v[d]);
}
DI int ldflag(const real* S, int row, int B, int b) { return (int)ld(S, row, B, b); }

// flags and the wrapper's target: enough to decide whether anything happens this tick
DI void load_head(const real* S, int n, bool cart, int B, int b, Gen& g) {
	g.goal_reached = ldflag(S, OTG_GOAL_REACHED, B, b);
	g.result = ldflag(S, OTG_RESULT, B, b);
	g.target_set = ldflag(S, OTG_TARGET_SET, B, b);
	load7(S, OTG_IN + 3 * MD, n, B, b, g.in.tp);
	load7(S, OTG_IN + 4 * MD, n, B, b, g.in.tv);
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) g.goal_R[i] = ld(S, OTG_CART + 9 + i, B, b);
		UNROLL for (int i = 0; i < 3; i++) g.goal_w[i] = ld(S, OTG_CART + 18 + i, B, b);
	}
}
DI void load_body(const real* S, int n, bool cart, int B, int b, Gen& g) {
	load7(S, OTG_IN, n, B, b, g.in.cp);
	load7(S, OTG_IN + MD, n, B, b, g.in.cv);
	load7(S, OTG_IN + 2 * MD, n, B, b, g.in.ca);
	load7(S, OTG_CI, n, B, b, g.ci.cp);
	load7(S, OTG_CI + MD, n, B, b, g.ci.cv);
	load7(S, OTG_CI + 2 * MD, n, B, b, g.ci.ca);
	load7(S, OTG_CI + 3 * MD, n, B, b, g.ci.tp);
	load7(S, OTG_CI + 4 * MD, n, B, b, g.ci.tv);
	load7(S, OTG_OUT, n, B, b, g.np);
	load7(S, OTG_OUT + MD, n, B, b, g.nv);
	load7(S, OTG_OUT + 2 * MD, n, B, b, g.na);
	g.time = ld(S, OTG_TIME, B, b);
	g.traj.duration = ld(S, OTG_DURATION, B, b);
	g.ci_init = ldflag(S, OTG_CI_INIT, B, b);
	g.ci_epoch = ld(S, OTG_CI_EPOCH, B, b);
	UNROLL for (int d = 0; d < MD; d++) {
		otg::Dof& f = g.traj.dof[d];
		otg::Prof& p = g.traj.prof[d];
		if (d < n) {
			const int r = OTG_TRAJ + d * OTG_TRAJ_STRIDE;
			f.brake_t = ld(S, r, B, b), f.brake_a = ld(S, r + 1, B, b), f.brake_p = ld(S, r + 2, B, b);
			f.brake_v = ld(S, r + 3, B, b), f.p0 = ld(S, r + 4, B, b), f.v0 = ld(S, r + 5, B, b);
			p.t0 = ld(S, r + 6, B, b), p.t1 = ld(S, r + 7, B, b), p.t2 = ld(S, r + 8, B, b), p.t6 = ld(S, r + 9, B, b);
			p.a0 = ld(S, r + 10, B, b), p.a2 = ld(S, r + 11, B, b), p.a6 = ld(S, r + 12, B, b);
			p.dur = ((p.t0 + p.t1) + p.t2) + p.t6;	// t_sum.back(), as Profile::check accumulates it
		} else {
			f = otg::Dof{0, 0, 0, 0, 0, 0, 0, 0};
			p = otg::Prof{0, 0, 0, 0, 0, 0, 0, 0, 0};
		}
		f.pf = f.vf = 0;
		p.dir = 0;
	}
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) g.ref[i] = ld(S, OTG_CART + i, B, b);
	}
}
DI void store_traj(real* S, int n, int B, int b, const Gen& g) {
	st(S, OTG_DURATION, B, b, g.traj.duration);
	UNROLL for (int d = 0; d < MD; d++)
		if (d < n) {
			const otg::Dof& f = g.traj.dof[d];
			const otg::Prof& p = g.traj.prof[d];
			const int r = OTG_TRAJ + d * OTG_TRAJ_STRIDE;
			st(S, r, B, b, f.brake_t), st(S, r + 1, B, b, f.brake_a), st(S, r + 2, B, b, f.brake_p);
			st(S, r + 3, B, b, f.brake_v), st(S, r + 4, B, b, f.p0), st(S, r + 5, B, b, f.v0);
			st(S, r + 6, B, b, p.t0), st(S, r + 7, B, b, p.t1), st(S, r + 8, B, b, p.t2), st(S, r + 9, B, b, p.t6);
			st(S, r + 10, B, b, p.a0), st(S, r + 11, B, b, p.a2), st(S, r + 12, B, b, p.a6);
		}
}
DI void store_state(real* S, int n, bool cart, int B, int b, const Gen& g) {
	store7(S, OTG_IN, n, B, b, g.in.cp);
	store7(S, OTG_IN + MD, n, B, b, g.in.cv);
	store7(S, OTG_IN + 2 * MD, n, B, b, g.in.ca);
	store7(S, OTG_IN + 3 * MD, n, B, b, g.in.tp);
	store7(S, OTG_IN + 4 * MD, n, B, b, g.in.tv);
	store7(S, OTG_CI, n, B, b, g.ci.cp);
	store7(S, OTG_CI + MD, n, B, b, g.ci.cv);
	store7(S, OTG_CI + 2 * MD, n, B, b, g.ci.ca);
	store7(S, OTG_CI + 3 * MD, n, B, b, g.ci.tp);
	store7(S, OTG_CI + 4 * MD, n, B, b, g.ci.tv);
	store7(S, OTG_OUT, n, B, b, g.np);
	store7(S, OTG_OUT + MD, n, B, b, g.nv);
	store7(S, OTG_OUT + 2 * MD, n, B, b, g.na);
	st(S, OTG_TIME, B, b, g.time);
	st(S, OTG_GOAL_REACHED, B, b, (double)g.goal_reached);
	st(S, OTG_RESULT, B, b, (double)g.result);
	st(S, OTG_TARGET_SET, B, b, (double)g.target_set);
	st(S, OTG_CI_INIT, B, b, (double)g.ci_init);
	st(S, OTG_CI_EPOCH, B, b, g.ci_epoch);
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) st(S, OTG_CART + i, B, b, g.ref[i]);
		UNROLL for (int i = 0; i < 9; i++) st(S, OTG_CART + 9 + i, B, b, g.goal_R[i]);
		UNROLL for (int i = 0; i < 3; i++) st(S, OTG_CART + 18 + i, B, b, g.goal_w[i]);
	}
}

// the next state in the layout of the goals (getNext*: OTG_joints.h:142-146, OTG_6dof_cartesian.h:205-227)
DI void store_desired_joints(real* D, int k0, int B, int b, const Gen& g) {
	UNROLL for (int d = 0; d < MD; d++)
		if (d < k0) {
			st(D, d, B, b, g.np[d]);
			st(D, k0 + d, B, b, g.nv[d]);
			st(D, 2 * k0 + d, B, b, g.na[d]);
		}
}
DI void store_desired_cart(real* D, int B, int b, const Gen& g) {
	real R[9], w[3], al[3];
	otg::cart_next_orientation(g, R);
	otg::mat3_vec(g.ref, g.nv[3], g.nv[4], g.nv[5], w);
	otg::mat3_vec(g.ref, g.na[3], g.na[4], g.na[5], al);
	UNROLL for (int k = 0; k < 3; k++) {
		st(D, k, B, b, g.np[k]);
		st(D, 12 + k, B, b, g.nv[k]);
		st(D, 15 + k, B, b, w[k]);
		st(D, 18 + k, B, b, g.na[k]);
		st(D, 21 + k, B, b, al[k]);
	}
	UNROLL for (int k = 0; k < 9; k++) st(D, 3 + k, B, b, R[k]);
}

DI void limits_of(const DevTask& t, double (&vmax)[MD], double (&amax)[MD]) {
	UNROLL for (int d = 0; d < MD; d++) vmax[d] = t.otg_vmax[d], amax[d] = t.otg_amax[d];
}

// one task, one robot, one tick
DI void otg_task_tick(const DevTask& t, int B, int b) {
	real* S = t.otg_state;
	const bool cart = t.type == SAI2B_MOTION_FORCE_TASK;
	const int n = t.otg_n;
	Gen g;
	load_head(S, n, cart, B, b, g);
	double vmax[MD], amax[MD];
	limits_of(t, vmax, amax);
	if (!cart) {
		double gp[MD], gv[MD];
		load7(t.goals, 0, n, B, b, gp);
		load7(t.goals, n, n, B, b, gv);
		otg::joints_set_goal(g, n, gp, gv);
		if (g.goal_reached) return;	 // nothing moves: otg_desired already holds the final state
		load_body(S, n, false, B, b, g);
		g.replanned = 0;
		otg::joints_update(g, n, t.dt, vmax, amax, t.otg_epoch);
		store_state(S, n, false, B, b, g);
		if (g.replanned) store_traj(S, n, B, b, g);
		store_desired_joints(t.otg_desired, n, B, b, g);
	} else {
		real gp[3], gR[9], gv[3], gw[3];
		UNROLL for (int k = 0; k < 3; k++) {
			gp[k] = ld(t.goals, k, B, b);
			gv[k] = ld(t.goals, 12 + k, B, b);
			gw[k] = ld(t.goals, 15 + k, B, b);
		}
		UNROLL for (int k = 0; k < 9; k++) gR[k] = ld(t.goals, 3 + k, B, b);
		// setGoalOrientationAndAngularVelocity re-references the frame from the current output, so
		// the body is needed before it whenever a goal changes; the idle test only needs the head
		const double p7[MD] = {gp[0], gp[1], gp[2], 0, 0, 0, 0}, v7[MD] = {gv[0], gv[1], gv[2], 0, 0, 0, 0};
		const bool same_pos = (g.target_set & 1) && otg::approx_range(p7, g.in.tp, 0, 3, 1e-3) &&
							  otg::approx_range(v7, g.in.tv, 0, 3, 1e-3);
		const bool same_ori = (g.target_set & 2) && otg::approx9(g.goal_R, gR, 9, 1e-3) && otg::approx9(g.goal_w, gw, 3, 1e-3);
		if (g.goal_reached && same_pos && same_ori) return;
		load_body(S, 6, true, B, b, g);
		otg::cart_set_goal_position(g, gp, gv);
		otg::cart_set_goal_orientation(g, gR, gw);
		g.replanned = 0;
		otg::cart_update(g, t.dt, vmax, amax, t.otg_epoch);
		store_state(S, 6, true, B, b, g);
		if (g.replanned) store_traj(S, 6, B, b, g);
		store_desired_cart(t.otg_desired, B, b, g);
	}
}

}  // namespace

// every task with its OTG on, one robot per lane
__global__ __launch_bounds__(64) void otg_kernel(const DevParams* __restrict__ Pp) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
#pragma unroll 1
	for (int t = 0; t < P.n_tasks; t++) {
		const DevTask& tk = P.task[t];
		if (!tk.otg_on) continue;
		otg_task_tick(tk, B, b);
	}
}

// The OTG objects exist whether or not the OTG is enabled and are re-initialised with the task
// (JointTask.cpp:71,106; MotionForceTask.cpp:171,244).
//   mode 0: reInitializeTask of every task; runs right after reinit_kernel, which has just written
//           goals = current pose / joint positions.
//   mode 1: enableInternalOtgAccelerationLimited on task `only_task` whose OTG was off: re-initialise
//           at the current state (JointTask.cpp:374-376, MotionForceTask.cpp:514-516; the reference
//           uses the pose cached by the last torque computation, this uses the state buffers), and
//           for a JointTask zero the input acceleration (OTG_joints::disableJerkLimits, :88-91).
//   mode 2: the same call on a task whose OTG was already on: only the JointTask's zeroing.
__global__ __launch_bounds__(64) void otg_reinit_kernel(const DevParams* __restrict__ Pp, int only_task, int mode) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
	real q[N];
	Frames F;
	if (mode == 1) {
		UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
		fk(P.model, q, F);
	}
#pragma unroll 1
	for (int t = 0; t < P.n_tasks; t++) {
		const DevTask& tk = P.task[t];
		if (only_task >= 0 && t != only_task) continue;
		real* S = tk.otg_state;
		const bool cart = tk.type == SAI2B_MOTION_FORCE_TASK;
		const int n = tk.otg_n;
		if (mode == 2 && cart) continue;
		Gen g;
		load_head(S, n, cart, B, b, g);
		load_body(S, n, cart, B, b, g);
		const bool constructed = ldflag(S, OTG_CONSTRUCTED, B, b) != 0;
		if (!constructed) g.result = otg::FINISHED;	 // OTG_joints.h:166, OTG_6dof_cartesian.h:235
		if (!cart) {
			if (mode != 2) {
				double x0[MD];
				if (mode == 1) {
					real cur[N];
					mv<N, N>(tk.S, q, cur);
					UNROLL for (int d = 0; d < MD; d++) x0[d] = d < n ? cur[d] : 0.0;
				} else {
					load7(tk.goals, 0, n, B, b, x0);
				}
				otg::joints_reinitialize(g, n, x0);
			}
			if (mode != 0) {
				UNROLL for (int d = 0; d < MD; d++) g.in.ca[d] = 0;
			}
			store_state(S, n, false, B, b, g);
			store_desired_joints(tk.otg_desired, n, B, b, g);
		} else {
			real x[3], R[9];
			if (mode == 1) {
				frame_pose(tk, F, x, R);
			} else {
				UNROLL for (int k = 0; k < 3; k++) x[k] = ld(tk.goals, k, B, b);
				UNROLL for (int k = 0; k < 9; k++) R[k] = ld(tk.goals, 3 + k, B, b);
			}
			if (!constructed) {	 // OTG_6dof_cartesian.cpp:41
				UNROLL for (int i = 0; i < 9; i++) g.ref[i] = R[i];
			}
			otg::cart_reinitialize(g, x, R);
			store_state(S, 6, true, B, b, g);
			store_desired_cart(tk.otg_desired, B, b, g);
			// wrench rows are not the OTG's; keep the desired buffer a complete goal record anyway
			for (int k = MFT_MOTION_GOAL_ROWS; k < MFT_GOAL_ROWS; k++) st(tk.otg_desired, k, B, b, 0.0);
		}
		st(S, OTG_CONSTRUCTED, B, b, 1.0);
	}
}

}  // namespace sai2b

extern "C" int sai2b_launch_otg(const sai2b::DevParams* d_params, int B, hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::otg_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int sai2b_launch_otg_reinit(const sai2b::DevParams* d_params, int B, int only_task, int mode,
									   hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::otg_reinit_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params, only_task, mode);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}
