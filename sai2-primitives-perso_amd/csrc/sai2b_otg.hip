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
// Two kernels per tick. otg_kernel classifies every robot: goal reached and unchanged -> nothing to
// do (~35 rows read, nothing written); moving on its stored trajectory -> sampled right there (~180
// rows read, ~85 written: state is OTG_ROWS doubles per robot, SoA rows of otg_state); goal changed or
// input differs -> appended to a work list. otg_plan_kernel runs the planner on the compacted list,
// so a tick in which 1 % of the robots get a new goal costs 1 % of the planner's time, not one
// divergent lane in every wavefront.
#include <hip/hip_runtime.h>

#include <cstdlib>

#include "sai2b_device.hpp"
#include "sai2b_launch.h"
#if SAI2B_N > 7
#define SAI2B_OTG_MAXD SAI2B_N
#endif
#include "sai2b_otg_core.hpp"
#include "sai2b_otg_group.hpp"
#include "sai2b_otg3_core.hpp"

namespace sai2b {
namespace {

using otg::Gen;
constexpr int MD = otg::MAXD;
static_assert(MD == OTG_MD, "generator DoFs and the state row layout (sai2b_params.h) must agree");

DI void load7(const real* S, int row0, int n, int B, int b, double (&v)[MD]) {
	UNROLL for (int d = 0; d < MD; d++) v[d] = d < n ? ld(S, row0 + d, B, b) : 0.0;
}
DI void store7(real* S, int row0, int n, int B, int b, const double (&v)[MD]) {
	UNROLL for (int d = 0; d < MD; d++)
		if (d < n) st(S, row0 + d, B, b, v[d]);
}
DI int ldflag(const real* S, int row, int B, int b) { return (int)ld(S, row, B, b); }
// element j of a batch-uniform 7-vector (scalar registers) without dynamic indexing
DI double sel7(const double (&v)[MD], int j) {
	double r = v[0];
	UNROLL for (int k = 1; k < MD; k++)
		if (k == j) r = v[k];
	return r;
}

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
	if (ldflag(S, OTG_IN_SYNC, B, b) != 0) {  // see OTG_IN_SYNC: the rows just read are stale, the state is the output
		UNROLL for (int d = 0; d < MD; d++) {
			g.in.cp[d] = g.ci.cp[d] = g.np[d];
			g.in.cv[d] = g.ci.cv[d] = g.nv[d];
			g.in.ca[d] = g.ci.ca[d] = g.na[d];
			g.ci.tp[d] = g.in.tp[d];
			g.ci.tv[d] = g.in.tv[d];
		}
	}
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
	st(S, OTG_IN_SYNC, B, b, 0.0);
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

// goals of one robot as the wrappers' setGoal... calls take them
struct Goals {
	double jp[MD], jv[MD];			  // JointTask: position, velocity
	real cp[3], cR[9], cv[3], cw[3];  // MotionForceTask: position, orientation, linear/angular velocity
};
DI void load_goals(const DevTask& t, bool cart, int n, int B, int b, Goals& G) {
	if (!cart) {
		load7(t.goals, 0, n, B, b, G.jp);
		load7(t.goals, n, n, B, b, G.jv);
	} else {
		UNROLL for (int k = 0; k < 3; k++) {
			G.cp[k] = ld(t.goals, k, B, b);
			G.cv[k] = ld(t.goals, 12 + k, B, b);
			G.cw[k] = ld(t.goals, 15 + k, B, b);
		}
		UNROLL for (int k = 0; k < 9; k++) G.cR[k] = ld(t.goals, 3 + k, B, b);
	}
}

// What this tick means for one generator:
//   IDLE    goal reached and unchanged: update() returns at once, nothing to read or write;
//   SAMPLE  goal unchanged, input equal to Ruckig's stored one: advance along the stored trajectory;
//   PLAN    the goal changed (setGoal... will touch the input) or the input differs from the stored
//           one (ruckig.hpp:194): the full update with the planner, done by otg_plan_kernel.
enum { IDLE = 0, SAMPLE = 1, PLAN = 2 };
// head_loaded: the wrapper's targets and flags (load_head) are in g; a robot that only samples a trajectory that is still
// running needs none of them, and they are fetched when its trajectory ends (sample_lane)
DI int classify(const DevTask& t, bool cart, int n, int B, int b, Gen& g, Goals& G, bool goals_clean, bool& in_sync, bool& head_loaded) {
	const real* S = t.otg_state;
	in_sync = false;
	head_loaded = false;
	// The host has not touched this task's goals (nor its OTG configuration) since the previous
	// update: setGoal...() is the same no-op as last time. For a robot whose goal is reached its flag alone
	// decides (1 row instead of ~35); for a robot on its way, in sync with its stored input, the goal rows and the
	// stored targets they would be compared with are not read at all (round 3: 28 of ~150 rows of a JointTask's
	// generator, 42 of ~165 of a MotionForceTask's, per tick)
	if (goals_clean) {
		if (ldflag(S, OTG_GOAL_REACHED, B, b) != 0) return IDLE;
		if (ldflag(S, OTG_IN_SYNC, B, b) != 0 && ldflag(S, OTG_TARGET_SET, B, b) == (cart ? 3 : 1)) {
			in_sync = true;
			g.goal_reached = 0;
			g.ci_init = ldflag(S, OTG_CI_INIT, B, b);
			g.ci_epoch = ld(S, OTG_CI_EPOCH, B, b);
			return (g.ci_epoch != t.otg_epoch || !g.ci_init) ? PLAN : SAMPLE;	// inputs equal by definition
		}
	}
	head_loaded = true;
	load_head(S, n, cart, B, b, g);
	load_goals(t, cart, n, B, b, G);
	bool unchanged;
	if (!cart) {
		unchanged = g.target_set && otg::approx_range(G.jp, g.in.tp, 0, n, 1e-12) && otg::approx_range(G.jv, g.in.tv, 0, n, 1e-12);
	} else {
		const double p7[MD] = {G.cp[0], G.cp[1], G.cp[2], 0, 0, 0, 0}, v7[MD] = {G.cv[0], G.cv[1], G.cv[2], 0, 0, 0, 0};
		unchanged = (g.target_set & 1) && otg::approx_range(p7, g.in.tp, 0, 3, 1e-3) && otg::approx_range(v7, g.in.tv, 0, 3, 1e-3) &&
					(g.target_set & 2) && otg::approx9(g.goal_R, G.cR, 9, 1e-3) && otg::approx9(g.goal_w, G.cw, 3, 1e-3);
	}
	if (!unchanged) return PLAN;
	if (g.goal_reached) return IDLE;
	in_sync = ldflag(S, OTG_IN_SYNC, B, b) != 0;
	g.ci_init = ldflag(S, OTG_CI_INIT, B, b);
	g.ci_epoch = ld(S, OTG_CI_EPOCH, B, b);
	if (in_sync) return (g.ci_epoch != t.otg_epoch || !g.ci_init) ? PLAN : SAMPLE;  // inputs equal by definition
	load7(S, OTG_IN, n, B, b, g.in.cp);
	load7(S, OTG_IN + MD, n, B, b, g.in.cv);
	load7(S, OTG_IN + 2 * MD, n, B, b, g.in.ca);
	load7(S, OTG_CI, n, B, b, g.ci.cp);
	load7(S, OTG_CI + MD, n, B, b, g.ci.cv);
	load7(S, OTG_CI + 2 * MD, n, B, b, g.ci.ca);
	load7(S, OTG_CI + 3 * MD, n, B, b, g.ci.tp);
	load7(S, OTG_CI + 4 * MD, n, B, b, g.ci.tv);
	return otg::needs_plan(g, n, t.otg_epoch) ? PLAN : SAMPLE;
}

DI void load_traj(const real* S, int n, bool cart, int B, int b, Gen& g) {
	g.time = ld(S, OTG_TIME, B, b);
	g.traj.duration = ld(S, OTG_DURATION, B, b);
	UNROLL for (int d = 0; d < MD; d++) {
		otg::Dof& f = g.traj.dof[d];
		otg::Prof& p = g.traj.prof[d];
		if (d < n) {
			const int r = OTG_TRAJ + d * OTG_TRAJ_STRIDE;
			f.brake_t = ld(S, r, B, b), f.p0 = ld(S, r + 4, B, b), f.v0 = ld(S, r + 5, B, b);
			f.brake_a = f.brake_p = f.brake_v = 0.0;
			if (f.brake_t > 0.0) {	// the rest of a brake pre-trajectory is read only where there is one (at_time)
				f.brake_a = ld(S, r + 1, B, b), f.brake_p = ld(S, r + 2, B, b), f.brake_v = ld(S, r + 3, B, b);
			}
			p.t0 = ld(S, r + 6, B, b), p.t1 = ld(S, r + 7, B, b), p.t2 = ld(S, r + 8, B, b), p.t6 = ld(S, r + 9, B, b);
			p.a0 = ld(S, r + 10, B, b), p.a2 = ld(S, r + 11, B, b), p.a6 = ld(S, r + 12, B, b);
			p.dur = ((p.t0 + p.t1) + p.t2) + p.t6;
		}
	}
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) g.ref[i] = ld(S, OTG_CART + i, B, b);
	}
}

// SAMPLE: Ruckig::update without a new calculation, then the wrapper's bookkeeping. `in_sync` on entry:
// the input rows were not loaded (they equal the output, OTG_IN_SYNC).
// Ruckig::update without a new calculation for a JERK-LIMITED generator (ruckig.hpp:205-215, trajectory.hpp:65-142): of
// the stored third-order profile of each DoF only the phase the new time falls into is read (OTG3_* rows of otg3_traj)
DI int sample_jerk(const DevTask& t, bool cart, int n, int B, int b, Gen& g) {
	const real* S = t.otg_state;
	const real* T3 = t.otg3_traj;
	g.time = ld(S, OTG_TIME, B, b) + t.dt;
	const double duration = ld(S, OTG_DURATION, B, b);
	g.traj.duration = duration;
	UNROLL for (int d = 0; d < MD; d++)
		if (d < n) {
			const int r = d * OTG3_STRIDE;
			const double brake_duration = ld(T3, r + OTG3_BRAKE, B, b);
			const double brake_t0 = brake_duration > 0 ? ld(T3, r + OTG3_BRAKE + 1, B, b) : 0.0;
			double t_sum[7];
			UNROLL for (int i = 0; i < 7; i++) t_sum[i] = ld(T3, r + OTG3_TSUM + i, B, b);
			double t_in;
			const int ph = otg3::phase_at(brake_duration, brake_t0, t_sum, duration, g.time, t_in);
			double p0, v0, a0, j0;
			if (ph >= 8) {	// brake pre-trajectory, phase ph - 8
				const int k = ph - 8;
				j0 = ld(T3, r + OTG3_BRAKE + 3 + k, B, b), a0 = ld(T3, r + OTG3_BRAKE + 5 + k, B, b);
				v0 = ld(T3, r + OTG3_BRAKE + 7 + k, B, b), p0 = ld(T3, r + OTG3_BRAKE + 9 + k, B, b);
			} else {
				j0 = ph < 7 ? ld(T3, r + OTG3_J + ph, B, b) : 0.0;
				a0 = ld(T3, r + OTG3_A + ph, B, b), v0 = ld(T3, r + OTG3_V + ph, B, b), p0 = ld(T3, r + OTG3_P + ph, B, b);
			}
			otg3::integrate(t_in, p0, v0, a0, j0, g.np[d], g.nv[d], g.na[d]);
			g.ci.cp[d] = g.np[d], g.ci.cv[d] = g.nv[d], g.ci.ca[d] = g.na[d];
		}
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) g.ref[i] = ld(S, OTG_CART + i, B, b);
	}
	return g.time > duration ? otg::FINISHED : otg::WORKING;
}

// JERK: the instantiation of otg_kernel launched while some task's generator is jerk-limited; the other one compiles
// exactly as it did before that mode existed (its registers and scratch are what the all-moving case is bound by)
template <bool JERK> DI void sample_lane(const DevTask& t, bool cart, int n, int B, int b, Gen& g, bool in_sync, bool head_loaded) {
	real* S = t.otg_state;
	bool done = false;
	if constexpr (JERK) {
		if (t.otg_jerk) {
			g.result = sample_jerk(t, cart, n, B, b, g);
			done = true;
		}
	}
	if (!done) {
		load_traj(S, n, cart, B, b, g);
		g.result = otg::ruckig_sample(g, n, t.dt, otg::WORKING);
	}
	if (g.result == otg::WORKING) {
		// both pass_to_input calls: input, Ruckig's stored input and output are one state again
		if (!in_sync) st(S, OTG_IN_SYNC, B, b, 1.0);
	} else {
		// Finished: Ruckig's stored input follows the output, the wrapper's input stays where it was
		// (OTG_joints.cpp:125-135): back to explicit rows
		if (!head_loaded) {
			const int res = g.result;  // (load_head reads the stored result of the previous update)
			load_head(S, n, cart, B, b, g);
			g.result = res;
		}
		if (in_sync) {	// where it was = the previous output, still in memory
			load7(S, OTG_OUT, n, B, b, g.in.cp);
			load7(S, OTG_OUT + MD, n, B, b, g.in.cv);
			load7(S, OTG_OUT + 2 * MD, n, B, b, g.in.ca);
			UNROLL for (int d = 0; d < MD; d++) g.ci.tp[d] = g.in.tp[d], g.ci.tv[d] = g.in.tv[d];
		}
		otg::Prev none;	 // the error branch cannot be taken without a calculation
		if (cart)
			otg::cart_finish(g, none);
		else
			otg::joints_finish(g, n, none);
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
		st(S, OTG_IN_SYNC, B, b, 0.0);
		st(S, OTG_GOAL_REACHED, B, b, (double)g.goal_reached);
		st(S, OTG_TARGET_SET, B, b, (double)g.target_set);
		if (cart && !g.goal_reached) {
			UNROLL for (int i = 0; i < 9; i++) st(S, OTG_CART + i, B, b, g.ref[i]);
			UNROLL for (int i = 0; i < 9; i++) st(S, OTG_CART + 9 + i, B, b, g.goal_R[i]);
			UNROLL for (int i = 0; i < 3; i++) st(S, OTG_CART + 18 + i, B, b, g.goal_w[i]);
		}
	}
	store7(S, OTG_OUT, n, B, b, g.np);
	store7(S, OTG_OUT + MD, n, B, b, g.nv);
	store7(S, OTG_OUT + 2 * MD, n, B, b, g.na);
	st(S, OTG_TIME, B, b, g.time);
	st(S, OTG_RESULT, B, b, (double)g.result);
	if (cart)
		store_desired_cart(t.otg_desired, B, b, g);
	else if (!t.otg_out_is_desired)
		store_desired_joints(t.otg_desired, n, B, b, g);
}

// PLAN, one DoF per lane (sai2b_otg_group.hpp): the whole computeTorques-time sequence
// setGoal...(goal); update(); (JointTask.cpp:314-315, MotionForceTask.cpp:395-399) for robot b by the
// caller's group of 8 lanes
DI void plan_group(const DevTask& t, bool cart, int n, int B, int b) {
	using namespace otgg;
	real* S = t.otg_state;
	const int j = lane_j();
	const bool active = j < n;
	auto row = [&](int r0) { return active ? ld(S, r0 + j, B, b) : 0.0; };
	LaneGen g;
	g.in_cp = row(OTG_IN), g.in_cv = row(OTG_IN + MD), g.in_ca = row(OTG_IN + 2 * MD), g.in_tp = row(OTG_IN + 3 * MD),
	g.in_tv = row(OTG_IN + 4 * MD);
	g.ci_cp = row(OTG_CI), g.ci_cv = row(OTG_CI + MD), g.ci_ca = row(OTG_CI + 2 * MD), g.ci_tp = row(OTG_CI + 3 * MD),
	g.ci_tv = row(OTG_CI + 4 * MD);
	g.np = row(OTG_OUT), g.nv = row(OTG_OUT + MD), g.na = row(OTG_OUT + 2 * MD);
	if (ldflag(S, OTG_IN_SYNC, B, b) != 0) {  // see OTG_IN_SYNC
		g.in_cp = g.ci_cp = g.np, g.in_cv = g.ci_cv = g.nv, g.in_ca = g.ci_ca = g.na;
		g.ci_tp = g.in_tp, g.ci_tv = g.in_tv;
	}
	{
		const int r = OTG_TRAJ + (active ? j : 0) * OTG_TRAJ_STRIDE;
		g.f.brake_t = ld(S, r, B, b), g.f.brake_a = ld(S, r + 1, B, b), g.f.brake_p = ld(S, r + 2, B, b);
		g.f.brake_v = ld(S, r + 3, B, b), g.f.p0 = ld(S, r + 4, B, b), g.f.v0 = ld(S, r + 5, B, b);
		g.f.pf = g.f.vf = 0;
		g.p.t0 = ld(S, r + 6, B, b), g.p.t1 = ld(S, r + 7, B, b), g.p.t2 = ld(S, r + 8, B, b), g.p.t6 = ld(S, r + 9, B, b);
		g.p.a0 = ld(S, r + 10, B, b), g.p.a2 = ld(S, r + 11, B, b), g.p.a6 = ld(S, r + 12, B, b);
		g.p.dur = ((g.p.t0 + g.p.t1) + g.p.t2) + g.p.t6;
		g.p.dir = 0;
	}
	g.time = ld(S, OTG_TIME, B, b);
	g.duration = ld(S, OTG_DURATION, B, b);
	g.goal_reached = ldflag(S, OTG_GOAL_REACHED, B, b);
	g.result = ldflag(S, OTG_RESULT, B, b);
	g.target_set = ldflag(S, OTG_TARGET_SET, B, b);
	g.ci_init = ldflag(S, OTG_CI_INIT, B, b);
	g.ci_epoch = ld(S, OTG_CI_EPOCH, B, b);
	g.replanned = 0;
	const double vmax = active ? sel7(t.otg_vmax, j) : 0.0, amax = active ? sel7(t.otg_amax, j) : 0.0;
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) g.ref[i] = ld(S, OTG_CART + i, B, b), g.goal_R[i] = ld(S, OTG_CART + 9 + i, B, b);
		UNROLL for (int i = 0; i < 3; i++) g.goal_w[i] = ld(S, OTG_CART + 18 + i, B, b);
		real gR[9], gw[3];
		UNROLL for (int k = 0; k < 9; k++) gR[k] = ld(t.goals, 3 + k, B, b);
		UNROLL for (int k = 0; k < 3; k++) gw[k] = ld(t.goals, 15 + k, B, b);
		const double gp = j < 3 ? ld(t.goals, j, B, b) : 0.0, gv = j < 3 ? ld(t.goals, 12 + j, B, b) : 0.0;
		cart_set_goal_position(g, gp, gv);
		cart_set_goal_orientation(g, gR, gw);
	} else {
		const double gp = active ? ld(t.goals, j, B, b) : 0.0, gv = active ? ld(t.goals, n + j, B, b) : 0.0;
		joints_set_goal(g, active, n, gp, gv);
	}
	update(g, cart, active, n, t.dt, vmax, amax, t.otg_epoch);

	auto put = [&](int r0, double v) {
		if (active) st(S, r0 + j, B, b, v);
	};
	put(OTG_IN, g.in_cp), put(OTG_IN + MD, g.in_cv), put(OTG_IN + 2 * MD, g.in_ca), put(OTG_IN + 3 * MD, g.in_tp), put(OTG_IN + 4 * MD, g.in_tv);
	put(OTG_CI, g.ci_cp), put(OTG_CI + MD, g.ci_cv), put(OTG_CI + 2 * MD, g.ci_ca), put(OTG_CI + 3 * MD, g.ci_tp), put(OTG_CI + 4 * MD, g.ci_tv);
	put(OTG_OUT, g.np), put(OTG_OUT + MD, g.nv), put(OTG_OUT + 2 * MD, g.na);
	if (g.replanned && active) {
		const int r = OTG_TRAJ + j * OTG_TRAJ_STRIDE;
		st(S, r, B, b, g.f.brake_t), st(S, r + 1, B, b, g.f.brake_a), st(S, r + 2, B, b, g.f.brake_p);
		st(S, r + 3, B, b, g.f.brake_v), st(S, r + 4, B, b, g.f.p0), st(S, r + 5, B, b, g.f.v0);
		st(S, r + 6, B, b, g.p.t0), st(S, r + 7, B, b, g.p.t1), st(S, r + 8, B, b, g.p.t2), st(S, r + 9, B, b, g.p.t6);
		st(S, r + 10, B, b, g.p.a0), st(S, r + 11, B, b, g.p.a2), st(S, r + 12, B, b, g.p.a6);
	}
	real R[9], w[3], al[3];
	if (cart) {	 // getNextOrientation / getNextAngular* (OTG_6dof_cartesian.cpp:226-237, .h:222-227)
		real local[9];
		otg::vec_to_rot(gget(g.np, 3), gget(g.np, 4), gget(g.np, 5), local);
		otg::mat3_mul(g.ref, local, R);
		otg::mat3_vec(g.ref, gget(g.nv, 3), gget(g.nv, 4), gget(g.nv, 5), w);
		otg::mat3_vec(g.ref, gget(g.na, 3), gget(g.na, 4), gget(g.na, 5), al);
	}
	if (j == 0) {
		st(S, OTG_TIME, B, b, g.time);
		if (g.replanned) st(S, OTG_DURATION, B, b, g.duration);
		st(S, OTG_GOAL_REACHED, B, b, (double)g.goal_reached);
		st(S, OTG_RESULT, B, b, (double)g.result);
		st(S, OTG_TARGET_SET, B, b, (double)g.target_set);
		st(S, OTG_CI_INIT, B, b, (double)g.ci_init);
		st(S, OTG_CI_EPOCH, B, b, g.ci_epoch);
		st(S, OTG_IN_SYNC, B, b, 0.0);
		if (cart) {
			UNROLL for (int i = 0; i < 9; i++) st(S, OTG_CART + i, B, b, g.ref[i]), st(S, OTG_CART + 9 + i, B, b, g.goal_R[i]);
			UNROLL for (int i = 0; i < 3; i++) st(S, OTG_CART + 18 + i, B, b, g.goal_w[i]);
			UNROLL for (int k = 0; k < 9; k++) st(t.otg_desired, 3 + k, B, b, R[k]);
			UNROLL for (int k = 0; k < 3; k++) st(t.otg_desired, 15 + k, B, b, w[k]), st(t.otg_desired, 21 + k, B, b, al[k]);
		}
	}
	if (cart) {
		if (j < 3) st(t.otg_desired, j, B, b, g.np), st(t.otg_desired, 12 + j, B, b, g.nv), st(t.otg_desired, 18 + j, B, b, g.na);
	} else if (active && !t.otg_out_is_desired) {
		st(t.otg_desired, j, B, b, g.np), st(t.otg_desired, n + j, B, b, g.nv), st(t.otg_desired, 2 * n + j, B, b, g.na);
	}
}

// PLAN for a JERK-LIMITED generator, one DoF per lane (sai2b_otg_group.hpp: LaneGen3, calculate3): the counterpart of
// plan_group for ruckig's third-order interface. The stored profile of the lane's DoF travels through otg3_traj.
DI void plan_group3(const DevTask& t, bool cart, int n, int B, int b) {
	using namespace otgg;
	real* S = t.otg_state;
	real* T3 = t.otg3_traj;
	const int j = lane_j();
	const bool active = j < n;
	auto row = [&](int r0) { return active ? ld(S, r0 + j, B, b) : 0.0; };
	LaneGen3 g;
	g.in_cp = row(OTG_IN), g.in_cv = row(OTG_IN + MD), g.in_ca = row(OTG_IN + 2 * MD), g.in_tp = row(OTG_IN + 3 * MD),
	g.in_tv = row(OTG_IN + 4 * MD);
	g.ci_cp = row(OTG_CI), g.ci_cv = row(OTG_CI + MD), g.ci_ca = row(OTG_CI + 2 * MD), g.ci_tp = row(OTG_CI + 3 * MD),
	g.ci_tv = row(OTG_CI + 4 * MD);
	g.np = row(OTG_OUT), g.nv = row(OTG_OUT + MD), g.na = row(OTG_OUT + 2 * MD);
	if (ldflag(S, OTG_IN_SYNC, B, b) != 0) {  // see OTG_IN_SYNC
		g.in_cp = g.ci_cp = g.np, g.in_cv = g.ci_cv = g.nv, g.in_ca = g.ci_ca = g.na;
		g.ci_tp = g.in_tp, g.ci_tv = g.in_tv;
	}
	{	// the stored third-order profile of this lane's DoF (sampled when no new calculation is needed)
		const int r = (active ? j : 0) * OTG3_STRIDE;
		otg3::Prof& p = g.p;
		p.brake.duration = ld(T3, r + OTG3_BRAKE, B, b);
		UNROLL for (int k = 0; k < 2; k++) {
			p.brake.t[k] = ld(T3, r + OTG3_BRAKE + 1 + k, B, b), p.brake.j[k] = ld(T3, r + OTG3_BRAKE + 3 + k, B, b);
			p.brake.a[k] = ld(T3, r + OTG3_BRAKE + 5 + k, B, b), p.brake.v[k] = ld(T3, r + OTG3_BRAKE + 7 + k, B, b);
			p.brake.p[k] = ld(T3, r + OTG3_BRAKE + 9 + k, B, b);
		}
		UNROLL for (int k = 0; k < 7; k++) p.t_sum[k] = ld(T3, r + OTG3_TSUM + k, B, b), p.j[k] = ld(T3, r + OTG3_J + k, B, b), p.t[k] = 0;
		UNROLL for (int k = 0; k < 8; k++) p.a[k] = ld(T3, r + OTG3_A + k, B, b), p.v[k] = ld(T3, r + OTG3_V + k, B, b), p.p[k] = ld(T3, r + OTG3_P + k, B, b);
		p.pf = p.vf = p.af = 0, p.limits = p.direction = p.control_signs = 0;
	}
	g.time = ld(S, OTG_TIME, B, b);
	g.duration = ld(S, OTG_DURATION, B, b);
	g.goal_reached = ldflag(S, OTG_GOAL_REACHED, B, b);
	g.result = ldflag(S, OTG_RESULT, B, b);
	g.target_set = ldflag(S, OTG_TARGET_SET, B, b);
	g.ci_init = ldflag(S, OTG_CI_INIT, B, b);
	g.ci_epoch = ld(S, OTG_CI_EPOCH, B, b);
	g.replanned = 0;
	const double vmax = active ? sel7(t.otg_vmax, j) : 0.0, amax = active ? sel7(t.otg_amax, j) : 0.0;
	g.jmax = active ? sel7(t.otg_jmax, j) : 0.0;
	if (cart) {
		UNROLL for (int i = 0; i < 9; i++) g.ref[i] = ld(S, OTG_CART + i, B, b), g.goal_R[i] = ld(S, OTG_CART + 9 + i, B, b);
		UNROLL for (int i = 0; i < 3; i++) g.goal_w[i] = ld(S, OTG_CART + 18 + i, B, b);
		real gR[9], gw[3];
		UNROLL for (int k = 0; k < 9; k++) gR[k] = ld(t.goals, 3 + k, B, b);
		UNROLL for (int k = 0; k < 3; k++) gw[k] = ld(t.goals, 15 + k, B, b);
		const double gp = j < 3 ? ld(t.goals, j, B, b) : 0.0, gv = j < 3 ? ld(t.goals, 12 + j, B, b) : 0.0;
		cart_set_goal_position(g, gp, gv);
		cart_set_goal_orientation(g, gR, gw);
	} else {
		const double gp = active ? ld(t.goals, j, B, b) : 0.0, gv = active ? ld(t.goals, n + j, B, b) : 0.0;
		joints_set_goal(g, active, n, gp, gv);
	}
	update(g, cart, active, n, t.dt, vmax, amax, t.otg_epoch);

	auto put = [&](int r0, double v) {
		if (active) st(S, r0 + j, B, b, v);
	};
	put(OTG_IN, g.in_cp), put(OTG_IN + MD, g.in_cv), put(OTG_IN + 2 * MD, g.in_ca), put(OTG_IN + 3 * MD, g.in_tp), put(OTG_IN + 4 * MD, g.in_tv);
	put(OTG_CI, g.ci_cp), put(OTG_CI + MD, g.ci_cv), put(OTG_CI + 2 * MD, g.ci_ca), put(OTG_CI + 3 * MD, g.ci_tp), put(OTG_CI + 4 * MD, g.ci_tv);
	put(OTG_OUT, g.np), put(OTG_OUT + MD, g.nv), put(OTG_OUT + 2 * MD, g.na);
	if (g.replanned && active) {
		const otg3::Prof& p = g.p;
		const int r = j * OTG3_STRIDE;
		st(T3, r + OTG3_BRAKE, B, b, p.brake.duration);
		UNROLL for (int k = 0; k < 2; k++) {
			st(T3, r + OTG3_BRAKE + 1 + k, B, b, p.brake.t[k]), st(T3, r + OTG3_BRAKE + 3 + k, B, b, p.brake.j[k]);
			st(T3, r + OTG3_BRAKE + 5 + k, B, b, p.brake.a[k]), st(T3, r + OTG3_BRAKE + 7 + k, B, b, p.brake.v[k]);
			st(T3, r + OTG3_BRAKE + 9 + k, B, b, p.brake.p[k]);
		}
		UNROLL for (int k = 0; k < 7; k++) st(T3, r + OTG3_TSUM + k, B, b, p.t_sum[k]), st(T3, r + OTG3_J + k, B, b, p.j[k]);
		UNROLL for (int k = 0; k < 8; k++) st(T3, r + OTG3_A + k, B, b, p.a[k]), st(T3, r + OTG3_V + k, B, b, p.v[k]), st(T3, r + OTG3_P + k, B, b, p.p[k]);
	}
	real R[9], w[3], al[3];
	if (cart) {
		real local[9];
		otg::vec_to_rot(gget(g.np, 3), gget(g.np, 4), gget(g.np, 5), local);
		otg::mat3_mul(g.ref, local, R);
		otg::mat3_vec(g.ref, gget(g.nv, 3), gget(g.nv, 4), gget(g.nv, 5), w);
		otg::mat3_vec(g.ref, gget(g.na, 3), gget(g.na, 4), gget(g.na, 5), al);
	}
	if (j == 0) {
		st(S, OTG_TIME, B, b, g.time);
		if (g.replanned) st(S, OTG_DURATION, B, b, g.duration);
		st(S, OTG_GOAL_REACHED, B, b, (double)g.goal_reached);
		st(S, OTG_RESULT, B, b, (double)g.result);
		st(S, OTG_TARGET_SET, B, b, (double)g.target_set);
		st(S, OTG_CI_INIT, B, b, (double)g.ci_init);
		st(S, OTG_CI_EPOCH, B, b, g.ci_epoch);
		st(S, OTG_IN_SYNC, B, b, 0.0);
		if (cart) {
			UNROLL for (int i = 0; i < 9; i++) st(S, OTG_CART + i, B, b, g.ref[i]), st(S, OTG_CART + 9 + i, B, b, g.goal_R[i]);
			UNROLL for (int i = 0; i < 3; i++) st(S, OTG_CART + 18 + i, B, b, g.goal_w[i]);
			UNROLL for (int k = 0; k < 9; k++) st(t.otg_desired, 3 + k, B, b, R[k]);
			UNROLL for (int k = 0; k < 3; k++) st(t.otg_desired, 15 + k, B, b, w[k]), st(t.otg_desired, 21 + k, B, b, al[k]);
		}
	}
	if (cart) {
		if (j < 3) st(t.otg_desired, j, B, b, g.np), st(t.otg_desired, 12 + j, B, b, g.nv), st(t.otg_desired, 18 + j, B, b, g.na);
	} else if (active && !t.otg_out_is_desired) {
		st(t.otg_desired, j, B, b, g.np), st(t.otg_desired, n + j, B, b, g.nv), st(t.otg_desired, 2 * n + j, B, b, g.na);
	}
}

// PLAN for a JERK-LIMITED generator, one lane per robot (kept as the A/B partner of plan_group3: SAI2B_OTG3_ONE_LANE=1): setGoal...(goal); update(); with ruckig's third-order
// interface (sai2b_otg3_core.hpp), DoF after DoF out of this lane's scratch memory. The wrapper state travels through
// the same rows as for the acceleration-limited generator (load_head / load_body / store_state); only the stored
// trajectory differs (otg3_traj).
__device__ __noinline__ void plan_lane3(const DevTask& t, bool cart, int n, int B, int b) {
	real* S = t.otg_state;
	otg3::Gen g3;
	{
		Gen g;	// wrapper part of the state, through the common row layout
		load_head(S, n, cart, B, b, g);
		load_body(S, n, cart, B, b, g);
		g3.in = g.in, g3.ci = g.ci;
		UNROLL for (int d = 0; d < MD; d++) g3.np[d] = g.np[d], g3.nv[d] = g.nv[d], g3.na[d] = g.na[d], g3.jmax[d] = t.otg_jmax[d];
		g3.time = g.time, g3.goal_reached = g.goal_reached, g3.result = g.result, g3.target_set = g.target_set;
		g3.ci_init = g.ci_init, g3.ci_epoch = g.ci_epoch, g3.replanned = 0;
		UNROLL for (int i = 0; i < 9; i++) g3.ref[i] = g.ref[i], g3.goal_R[i] = g.goal_R[i];
		UNROLL for (int i = 0; i < 3; i++) g3.goal_w[i] = g.goal_w[i];
		g3.traj.duration = g.traj.duration;
	}
	Goals G;
	load_goals(t, cart, n, B, b, G);
	if (cart) {
		otg::cart_set_goal_position(g3, G.cp, G.cv);
		otg::cart_set_goal_orientation(g3, G.cR, G.cw);
		otg::cart_update(g3, t.dt, t.otg_vmax, t.otg_amax, t.otg_epoch);
	} else {
		otg::joints_set_goal(g3, n, G.jp, G.jv);
		otg::joints_update(g3, n, t.dt, t.otg_vmax, t.otg_amax, t.otg_epoch);
	}
	if (g3.replanned) {
		real* T3 = t.otg3_traj;
		st(S, OTG_DURATION, B, b, g3.traj.duration);
		for (int d = 0; d < n; d++) {
			const otg3::Prof& p = g3.traj.prof[d];
			const int r = d * OTG3_STRIDE;
			st(T3, r + OTG3_BRAKE, B, b, p.brake.duration);
			for (int k = 0; k < 2; k++) {
				st(T3, r + OTG3_BRAKE + 1 + k, B, b, p.brake.t[k]), st(T3, r + OTG3_BRAKE + 3 + k, B, b, p.brake.j[k]);
				st(T3, r + OTG3_BRAKE + 5 + k, B, b, p.brake.a[k]), st(T3, r + OTG3_BRAKE + 7 + k, B, b, p.brake.v[k]);
				st(T3, r + OTG3_BRAKE + 9 + k, B, b, p.brake.p[k]);
			}
			for (int k = 0; k < 7; k++) st(T3, r + OTG3_TSUM + k, B, b, p.t_sum[k]), st(T3, r + OTG3_J + k, B, b, p.j[k]);
			for (int k = 0; k < 8; k++) st(T3, r + OTG3_A + k, B, b, p.a[k]), st(T3, r + OTG3_V + k, B, b, p.v[k]), st(T3, r + OTG3_P + k, B, b, p.p[k]);
		}
	}
	{
		Gen g;
		g.in = g3.in, g.ci = g3.ci;
		UNROLL for (int d = 0; d < MD; d++) g.np[d] = g3.np[d], g.nv[d] = g3.nv[d], g.na[d] = g3.na[d];
		g.time = g3.time, g.goal_reached = g3.goal_reached, g.result = g3.result, g.target_set = g3.target_set;
		g.ci_init = g3.ci_init, g.ci_epoch = g3.ci_epoch;
		UNROLL for (int i = 0; i < 9; i++) g.ref[i] = g3.ref[i], g.goal_R[i] = g3.goal_R[i];
		UNROLL for (int i = 0; i < 3; i++) g.goal_w[i] = g3.goal_w[i];
		store_state(S, n, cart, B, b, g);
		if (cart)
			store_desired_cart(t.otg_desired, B, b, g);
		else if (!t.otg_out_is_desired)
			store_desired_joints(t.otg_desired, n, B, b, g);
	}
}

}  // namespace

// Work list of the planner: per task a counter and the robot indices that need it this tick.
// Two counter sets alternate between ticks (`parity`): the plan kernel of tick k clears the set of
// tick k+1, so no extra clearing launch is needed.
//   counts: [2][SAI2B_MAX_TASKS] ints          list: [SAI2B_MAX_TASKS][B] ints

// Every generator that is on, one robot per lane: idle and sampling robots are finished here; robots
// that need the planner are appended to the task's work list (one atomic per wavefront, lanes of a
// wavefront stay adjacent and ordered, so the plan kernel's accesses coalesce in runs).
template <bool JERK>
__global__ __launch_bounds__(64) void otg_kernel(const DevParams* __restrict__ Pp, int* __restrict__ counts,
												 int* __restrict__ list, int parity, int clean_mask, int task_mask) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	const bool live = b < B;
	bool busy = false;	// some generator of this robot is not idle (the host stops launching these kernels when none is: sai2b_host.cpp)
#pragma unroll 1
	for (int t = 0; t < P.n_tasks; t++) {
		const DevTask& tk = P.task[t];
		if (!tk.otg_on || !((task_mask >> t) & 1)) continue;
		const bool cart = tk.type == SAI2B_MOTION_FORCE_TASK;
		int cls = IDLE;
		// a JointTask with an empty range returns before it touches its generator (JointTask.cpp:302-306)
		if (live && !(tk.otg_gated && ld(tk.otg_state, OTG_ACTIVE, B, b) == 0.0)) {
			Gen g;
			Goals G;
			bool in_sync, head_loaded;
			cls = classify(tk, cart, tk.otg_n, B, b, g, G, ((clean_mask >> t) & 1) != 0, in_sync, head_loaded);
			if (cls == SAMPLE) sample_lane<JERK>(tk, cart, tk.otg_n, B, b, g, in_sync, head_loaded);
			busy = busy || cls != IDLE;
		}
		const unsigned long long mask = __ballot(cls == PLAN);
		if (mask) {
			int base = 0;
			if (threadIdx.x == 0) base = atomicAdd(&counts[parity * SAI2B_MAX_TASKS + t], __popcll(mask));
			base = __shfl(base, 0);
			if (cls == PLAN) {
				const int pos = base + __popcll(mask & ((1ull << threadIdx.x) - 1ull));
				((gint*)list)[(size_t)t * B + pos] = b;
			}
		}
	}
	{
		const unsigned long long m = __ballot(busy);
		if (m && (threadIdx.x & 63) == 0) atomicAdd(&counts[2 * SAI2B_MAX_TASKS + parity], __popcll(m));
	}
}

// The robots otg_kernel left over (goal changed / input differs), compacted: group i (8 lanes, one DoF
// per lane) of the grid takes entry i of the task's list and does the robot's full update.
__global__ __launch_bounds__(64) void otg_plan_kernel(const DevParams* __restrict__ Pp, int* __restrict__ counts,
													  const int* __restrict__ list, int parity, int task_mask) {
	const DevParams& P = *Pp;
	const int B = P.B;
	if (blockIdx.x == 0 && threadIdx.x < SAI2B_MAX_TASKS) ((gint*)counts)[(1 - parity) * SAI2B_MAX_TASKS + threadIdx.x] = 0;  // next tick's counters
	if (blockIdx.x == 0 && threadIdx.x == 0) ((gint*)counts)[2 * SAI2B_MAX_TASKS + (1 - parity)] = 0;  // and its count of non-idle robots
	constexpr int GROUPS = 64 / otgg::G;
#pragma unroll 1
	for (int t = 0; t < P.n_tasks; t++) {
		const DevTask& tk = P.task[t];
		if (!tk.otg_on || tk.otg_jerk || !((task_mask >> t) & 1)) continue;  // (jerk-limited generators: otg3_plan_kernel)
		const int cnt = ((const gint*)counts)[parity * SAI2B_MAX_TASKS + t];
		// grid-stride over the list: the grid is sized for the machine, not for the worst-case list
#pragma unroll 1
		for (int e0 = blockIdx.x * GROUPS; e0 < cnt; e0 += gridDim.x * GROUPS) {  // uniform over the wavefront
			const int e = e0 + threadIdx.x / otgg::G;
			if (e >= cnt) continue;	 // uniform over the group
			const int b = ((const gint*)list)[(size_t)t * B + e];
			plan_group(tk, tk.type == SAI2B_MOTION_FORCE_TASK, tk.otg_n, B, b);
		}
	}
}

// The listed robots of the JERK-LIMITED generators, one lane each (plan_lane3). Launched behind otg_plan_kernel only
// while some task is jerk-limited; a small grid striding over the lists bounds the scratch memory the third-order
// planner needs per lane (~20 KB: seven DoFs of blocks, each with up to three stored profiles).
template <bool ONE_LANE>
__global__ __launch_bounds__(64) void otg3_plan_kernel(const DevParams* __restrict__ Pp, const int* __restrict__ counts,
													   const int* __restrict__ list, int parity, int task_mask) {
	const DevParams& P = *Pp;
	const int B = P.B;
	constexpr int GROUPS = 64 / otgg::G;
#pragma unroll 1
	for (int t = 0; t < P.n_tasks; t++) {
		const DevTask& tk = P.task[t];
		if (!tk.otg_on || !tk.otg_jerk || !((task_mask >> t) & 1)) continue;
		const int cnt = ((const gint*)counts)[parity * SAI2B_MAX_TASKS + t];
		if constexpr (ONE_LANE) {
#pragma unroll 1
			for (int e = blockIdx.x * 64 + threadIdx.x; e < cnt; e += gridDim.x * 64) {
				const int b = ((const gint*)list)[(size_t)t * B + e];
				plan_lane3(tk, tk.type == SAI2B_MOTION_FORCE_TASK, tk.otg_n, B, b);
			}
		} else {
#pragma unroll 1
			for (int e0 = blockIdx.x * GROUPS; e0 < cnt; e0 += gridDim.x * GROUPS) {  // uniform over the wavefront
				const int e = e0 + threadIdx.x / otgg::G;
				if (e >= cnt) continue;	 // uniform over the group
				const int b = ((const gint*)list)[(size_t)t * B + e];
				plan_group3(tk, tk.type == SAI2B_MOTION_FORCE_TASK, tk.otg_n, B, b);
			}
		}
	}
}

// The OTG objects exist whether or not the OTG is enabled and are re-initialised with the task
// (JointTask.cpp:71,106; MotionForceTask.cpp:171,244).
//   mode 0: reInitializeTask of every task; runs right after reinit_kernel, which has just written
//           goals = current pose / joint positions.
//   mode 1: enableInternalOtgAccelerationLimited on task `only_task` whose OTG was off: re-initialise
//           at the task's _current_position / _current_orientation (JointTask.cpp:374-376,
//           MotionForceTask.cpp:514-516), i.e. the pose of the last torque computation or
//           re-initialisation: q_pose holds the joint positions of that moment (the host keeps them
//           when the state buffers move on), and for a JointTask zero the input acceleration
//           (OTG_joints::disableJerkLimits, :88-91).
//   mode 2: the same call on a task whose OTG was already on: only the JointTask's zeroing.
__global__ __launch_bounds__(64) void otg_reinit_kernel(const DevParams* __restrict__ Pp, int only_task, int mode,
														const double* __restrict__ q_pose) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
	real q[N];
	if (mode == 1) {
		UNROLL for (int i = 0; i < N; i++) q[i] = ld(q_pose, i, B, b);
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
					UNROLL for (int d = 0; d < MD; d++) x0[d] = (d < n && d < N) ? cur[d < N ? d : 0] : 0.0;
				} else {
					load7(tk.goals, 0, n, B, b, x0);
				}
				otg::joints_reinitialize(g, n, x0);
			}
			if (mode != 0) {
				UNROLL for (int d = 0; d < MD; d++) g.in.ca[d] = 0;
			}
			store_state(S, n, false, B, b, g);
			if (!tk.otg_out_is_desired) store_desired_joints(tk.otg_desired, n, B, b, g);
		} else {
			real x[3], R[9];
			if (mode == 1) {
				det_frame_pose(P.model, tk, q, x, R);
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

// Run-time re-parametrisation of a MotionForceTask's force / motion spaces (parametrizeForceMotionSpaces,
// parametrizeMomentRotMotionSpaces: MotionForceTask.cpp:830-890) and closed-loop switches (:973-986), the
// per-robot part. flags: 1 = linear half (goal position := current position, goal linear velocity /
// acceleration := 0, generator re-initialised there: reInitializeLinear), 2 = angular half, 4 / 8 = reset
// the linear (position, force) / angular (orientation, moment) integrators. "Current" is the task's cached
// pose: q_pose, the joint positions of the last torque computation or re-initialisation.
__global__ __launch_bounds__(64) void mft_reparam_kernel(const DevParams* __restrict__ Pp, int task, int flags,
														 const double* __restrict__ q_pose) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
	const DevTask& tk = P.task[task];
	if (flags & 3) {
		real q[N], x[3], R[9];
		UNROLL for (int i = 0; i < N; i++) q[i] = ld(q_pose, i, B, b);
		det_frame_pose(P.model, tk, q, x, R);
		real* S = tk.otg_state;
		Gen g;
		load_head(S, 6, true, B, b, g);
		load_body(S, 6, true, B, b, g);
		if (flags & 1) {
			UNROLL for (int k = 0; k < 3; k++) {
				st(tk.goals, k, B, b, x[k]);
				st(tk.goals, 12 + k, B, b, 0.0);
				st(tk.goals, 18 + k, B, b, 0.0);
			}
			otg::cart_reinitialize_linear(g, x);
		}
		if (flags & 2) {
			UNROLL for (int k = 0; k < 9; k++) st(tk.goals, 3 + k, B, b, R[k]);
			UNROLL for (int k = 0; k < 3; k++) {
				st(tk.goals, 15 + k, B, b, 0.0);
				st(tk.goals, 21 + k, B, b, 0.0);
			}
			otg::cart_reinitialize_angular(g, R);
		}
		store_state(S, 6, true, B, b, g);
		store_desired_cart(tk.otg_desired, B, b, g);
	}
	UNROLL for (int k = 0; k < 3; k++) {
		if (flags & 4) st(tk.state, k, B, b, 0.0), st(tk.state, 6 + k, B, b, 0.0);
		if (flags & 8) st(tk.state, 3 + k, B, b, 0.0), st(tk.state, 9 + k, B, b, 0.0);
	}
}

}  // namespace sai2b

extern "C" int sai2b_launch_mft_reparam(const sai2b::DevParams* d_params, int B, int task, int flags, const double* q_pose,
										hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::mft_reparam_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params, task, flags, q_pose);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

// counts: [2][SAI2B_MAX_TASKS] ints, zero before the first call; list: [SAI2B_MAX_TASKS][B] ints;
// parity alternates 0/1 between consecutive calls
// clean_mask bit t: the host has not written task t's goals / OTG settings since the previous call
// jerk_mask: tasks whose generator is jerk-limited (host knowledge: DevTask::otg_jerk) — their lists get a third launch
extern "C" int sai2b_launch_otg(const sai2b::DevParams* d_params, int B, int* counts, int* list, int parity, int clean_mask,
								int task_mask, int jerk_mask, hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	if (jerk_mask & task_mask)
		hipLaunchKernelGGL(sai2b::otg_kernel<true>, grid, block, 0, stream, d_params, counts, list, parity, clean_mask, task_mask);
	else
		hipLaunchKernelGGL(sai2b::otg_kernel<false>, grid, block, 0, stream, d_params, counts, list, parity, clean_mask, task_mask);
	const int plan_blocks = (B + 7) / 8 < 2048 ? (B + 7) / 8 : 2048;
	hipLaunchKernelGGL(sai2b::otg_plan_kernel, dim3(plan_blocks), block, 0, stream, d_params, counts, (const int*)list, parity, task_mask);
	if (jerk_mask & task_mask) {
		static const bool one_lane = std::getenv("SAI2B_OTG3_ONE_LANE") != nullptr;	 // A/B: round 3's first planner, one lane per robot
		if (one_lane) {
			const int blocks3 = (B + 63) / 64 < 256 ? (B + 63) / 64 : 256;
			hipLaunchKernelGGL(sai2b::otg3_plan_kernel<true>, dim3(blocks3), block, 0, stream, d_params, (const int*)counts, (const int*)list,
							   parity, task_mask);
		} else {  // one DoF per lane; the grid strides over the list and bounds the scratch (one block of profiles per lane)
			const int blocks3 = (B + 7) / 8 < 1024 ? (B + 7) / 8 : 1024;
			hipLaunchKernelGGL(sai2b::otg3_plan_kernel<false>, dim3(blocks3), block, 0, stream, d_params, (const int*)counts, (const int*)list,
							   parity, task_mask);
		}
	}
	return hipGetLastError() == hipSuccess ? 0 : 1;
}

extern "C" int sai2b_launch_otg_reinit(const sai2b::DevParams* d_params, int B, int only_task, int mode,
									   const double* q_pose, hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::otg_reinit_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params, only_task, mode, q_pose);
	return hipGetLastError() == hipSuccess ? 0 : 1;
}
