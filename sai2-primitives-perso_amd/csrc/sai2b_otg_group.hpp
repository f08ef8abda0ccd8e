// sai2b_otg_group.hpp — the full OTG update of one robot spread over a group of 8 lanes, one DoF per
// lane (device only; included by sai2b_otg.hip).
//
// otg_kernel leaves the robots whose goal changed (or whose input differs from Ruckig's stored one)
// on a compacted work list. Planning them one robot per lane means seven DoFs in sequence out of
// per-lane scratch: ~300 us of single-wavefront latency. Here lane j of a group owns DoF j: brake,
// step 1, the phase / step 2 profile and the sampling of the seven DoFs run side by side out of
// registers; what couples the DoFs — the wrappers' isApprox norms, Ruckig's input comparison, the
// synchronisation of calculator_target.hpp:120-222, the collinearity test — is done with 8-lane
// shuffles and ballots. Per-DoF arithmetic is sai2b_otg_core.hpp's (the host-tested code); the sums
// over DoFs are taken in DoF order, like the sequential form.
//
// A wavefront holds 8 groups. Conditions that are uniform over a group may diverge between groups;
// every shuffle below reads lanes of the caller's own group, which take the same branch.
#pragma once
#include "sai2b_device.hpp"
#include "sai2b_otg_core.hpp"
#include "sai2b_otg3_core.hpp"

namespace sai2b {
namespace otgg {

using otg::Block;
using otg::Dof;
using otg::Prof;
constexpr int G = 8;  // lanes per group; DoF j < n <= 8 active

DI int lane_j() { return threadIdx.x & (G - 1); }
DI int group_base() { return threadIdx.x & ~(G - 1); }
DI double gget(double v, int k) { return __shfl(v, group_base() + k); }
DI int ggeti(int v, int k) { return __shfl(v, group_base() + k); }
DI unsigned gbits(bool pred) { return (unsigned)((__ballot(pred) >> group_base()) & 0xffull); }
DI bool gany(bool pred) { return gbits(pred) != 0; }
// sum over the group's lanes in lane (= DoF) order
DI double gsum(double v) {
	double s = 0;
	UNROLL for (int k = 0; k < G; k++) s += gget(v, k);
	return s;
}
DI int gor(int v) {
	v |= __shfl_xor(v, 1);
	v |= __shfl_xor(v, 2);
	v |= __shfl_xor(v, 4);
	return v;
}
// Eigen isApprox over the lanes [lo, hi): |a-b|^2 <= prec^2 min(|a|^2, |b|^2)
DI bool gapprox(double a, double b, int lo, int hi, double prec) {
	const int j = lane_j();
	const bool on = j >= lo && j < hi;
	const double dd = gsum(on ? (a - b) * (a - b) : 0.0), na = gsum(on ? a * a : 0.0), nb = gsum(on ? b * b : 0.0);
	return dd <= prec * prec * (na < nb ? na : nb);
}

// one lane's share of a generator (struct Gen of sai2b_otg_core.hpp, one DoF) + the group-uniform part
struct LaneGen {
	double in_cp, in_cv, in_ca, in_tp, in_tv;
	double ci_cp, ci_cv, ci_ca, ci_tp, ci_tv;
	double np, nv, na;
	Dof f;
	Prof p;
	// uniform over the group
	double time, duration, ci_epoch;
	int goal_reached, result, target_set, ci_init, replanned;
	double ref[9], goal_R[9], goal_w[3];  // Cartesian wrapper
};

// TargetCalculator::calculate (calculator_target.hpp:249-532), one DoF per lane. Returns the Result;
// on WORKING f, p and duration hold the new trajectory.
DI int calculate(bool active, int n, double cp, double cv, double ca, double tp, double tv, double vmax, double amax, Dof& f,
				 Prof& p, double& duration) {
	const int j = lane_j();
	Block bl;
	bl.a = bl.b = false;
	bl.tmin = 0;
	bool ok1 = true;
	if (active) {
		const double vMax = vmax, vMin = -vmax, aMax = amax, aMin = -amax;
		f.brake_t = 0.0, f.brake_a = 0.0, f.brake_p = 0.0, f.brake_v = 0.0;	 // brake.cpp:79-99, brake.hpp:66-75
		if (!(aMax == 0.0 || aMin == 0.0)) {
			if (cv > vMax) {
				f.brake_a = aMin;
				f.brake_t = (vMax - cv) / aMin + 2.2e-14;
			} else if (cv < vMin) {
				f.brake_a = aMax;
				f.brake_t = (vMin - cv) / aMax + 2.2e-14;
			}
		}
		f.p0 = cp, f.v0 = cv, f.pf = tp, f.vf = tv;
		if (f.brake_t > 0.0) {
			const double t = f.brake_t, ps = f.p0, vs = f.v0, ab = f.brake_a;
			f.brake_p = ps, f.brake_v = vs;
			f.p0 = ps + t * (vs + t * (ab / 2 + t * 0.0 / 6));
			f.v0 = vs + t * (ab + t * 0.0 / 2);
		}
		ok1 = otg::step1(f, bl, vMax, aMax);
	}
	if (gany(active && !ok1)) return otg::ERR_EXECUTION_TIME;
	if (n == 1) {
		duration = gget(bl.tmin, 0);
		if (j == 0) p = bl.pmin;
		return otg::WORKING;
	}

	// synchronize (calculator_target.hpp:120-222): every lane sees all candidates (t_min, ends of the
	// blocked intervals) and whether any DoF blocks them; the choice is then made redundantly
	const double c0 = active ? bl.tmin : INFINITY, c1 = (active && bl.a) ? bl.aright : INFINITY,
				 c2 = (active && bl.b) ? bl.bright : INFINITY;
	const bool any_interval = gany(active && (bl.a || bl.b));
	double cand[3][G];
	UNROLL for (int d = 0; d < G; d++) {
		cand[0][d] = gget(c0, d);
		cand[1][d] = gget(c1, d);
		cand[2][d] = gget(c2, d);
	}
	int blocked = 0;
	UNROLL for (int q = 0; q < 3; q++)
		UNROLL for (int d = 0; d < G; d++)
			if (active && otg::is_blocked(bl, cand[q][d])) blocked |= 1 << (q * G + d);
	blocked = gor(blocked);
	int best_rank = 1 << 20, best_q = -1, best_d = -1;
	UNROLL for (int q = 0; q < 3; q++)
		UNROLL for (int d = 0; d < G; d++) {
			if (d >= n || (!any_interval && q > 0)) continue;
			const double t = cand[q][d];
			int rank = 0;  // position in the reference's stably sorted index array (index = q*n + d)
			UNROLL for (int oq = 0; oq < 3; oq++)
				UNROLL for (int od = 0; od < G; od++) {
					if (od >= n || (!any_interval && oq > 0)) continue;
					const bool before = (oq < q) || (oq == q && od < d);
					if (cand[oq][od] < t || (cand[oq][od] == t && before)) rank++;
				}
			if (rank < n - 1 || rank >= best_rank) continue;
			if (((blocked >> (q * G + d)) & 1) || t < 0.0 || isinf(t)) continue;
			best_rank = rank, best_q = q, best_d = d;
		}
	if (best_d < 0) return otg::ERR_SYNCHRONIZATION;
	const int limiting = best_d;
	{
		double t = 0;
		UNROLL for (int q = 0; q < 3; q++)
			UNROLL for (int d = 0; d < G; d++)
				if (q == best_q && d == best_d) t = cand[q][d];
		duration = t;
	}
	if (j == limiting) p = best_q == 0 ? bl.pmin : best_q == 1 ? bl.aprof : bl.bprof;
	if (duration > 7.6e3) return otg::ERR_TRAJECTORY_DURATION;
	if (duration == 0.0) {
		if (active) p = bl.pmin;
		return otg::WORKING;
	}

	// phase synchronisation (calculator_target.hpp:398-467), collinearity test :46-118
	{
		const double pl_t0 = gget(p.t0, limiting), pl_t1 = gget(p.t1, limiting), pl_t2 = gget(p.t2, limiting),
					 pl_t6 = gget(p.t6, limiting);
		const int pl_dir = ggeti(p.dir, limiting);
		const double pd = tp - cp;
		int which = -1;
		if (active) {
			if (fabs(pd) > otg::EPS)
				which = 0;
			else if (fabs(cv) > otg::EPS)
				which = 1;
			else if (fabs(ca) > otg::EPS)
				which = 2;
			else if (fabs(tv) > otg::EPS)
				which = 3;
		}
		const unsigned has = gbits(which >= 0);
		if (has) {
			const int scale_dof = __ffs((int)has) - 1;
			const int w = ggeti(which, scale_dof);
			const double sv = w == 0 ? pd : w == 1 ? cv : w == 2 ? ca : tv;
			const double scale = gget(sv, scale_dof);
			const double pd_scale = gget(pd, scale_dof) / scale, v0_scale = gget(cv, scale_dof) / scale,
						 vf_scale = gget(tv, scale_dof) / scale, a0_scale = gget(ca, scale_dof) / scale, af_scale = 0.0 / scale;
			const double scale_limiting = gget(sv, limiting);
			const double amax_lim = gget(amax, limiting);
			const double control_limiting = (pl_dir == 0) ? amax_lim : -amax_lim;
			const bool off = active && (fabs(pd - pd_scale * sv) > otg::EPS || fabs(cv - v0_scale * sv) > otg::EPS ||
										fabs(ca - a0_scale * sv) > otg::EPS || fabs(tv - vf_scale * sv) > otg::EPS ||
										fabs(0.0 - af_scale * sv) > otg::EPS);
			if (!gany(off)) {
				const double npc = control_limiting * sv / scale_limiting;
				bool lane_ok = true;
				if (active && j != limiting) {
					p.t0 = pl_t0, p.t1 = pl_t1, p.t2 = pl_t2, p.t6 = pl_t6;
					const double aUp = npc, aDown = -npc, aMax = amax, aMin = -amax;
					const bool within = (aMin - 1e-12 < aUp) && (aUp < aMax + 1e-12) && (aMin - 1e-12 < aDown) && (aDown < aMax + 1e-12);
					lane_ok = within && otg::check(p, f, aUp, aDown, vmax, -vmax);
				}
				if (!gany(!lane_ok)) return otg::WORKING;
			}
		}
	}

	// time synchronisation (calculator_target.hpp:469-529)
	bool bad = false;
	if (active && j != limiting) {
		const double t_profile = duration - otg::brake_duration(f) - 0.0;
		if (fabs(t_profile - bl.tmin) < 2 * otg::EPS) {
			p = bl.pmin;
		} else if (bl.a && fabs(t_profile - bl.aright) < 2 * otg::EPS) {
			p = bl.aprof;
		} else if (bl.b && fabs(t_profile - bl.bright) < 2 * otg::EPS) {
			p = bl.bprof;
		} else if (!otg::step2(p, f, t_profile, vmax, amax)) {
			bad = true;
		}
	}
	return gany(bad) ? otg::ERR_SYNCHRONIZATION : otg::WORKING;
}

// ---- the jerk-limited generator, one DoF per lane (round 3): the per-DoF planner is sai2b_otg3_core.hpp's (brake, step 1,
// step 2, check, sampling: the code whose host build is bit-equal to the reference's ruckig), the couplings between the
// DoFs are the same shuffles and ballots as above on the third-order profile ----
struct LaneGen3 {
	double in_cp, in_cv, in_ca, in_tp, in_tv;
	double ci_cp, ci_cv, ci_ca, ci_tp, ci_tv;
	double np, nv, na;
	otg3::Prof p;
	double jmax;
	// uniform over the group
	double time, duration, ci_epoch;
	int goal_reached, result, target_set, ci_init, replanned;
	double ref[9], goal_R[9], goal_w[3];
};

// TargetCalculator::calculate (calculator_target.hpp:236-532) for finite max_jerk, one DoF per lane
__device__ __noinline__ int calculate3(bool active, int n, double cp, double cv, double ca, double tp, double tv, double vmax, double amax,
									   double jmax, otg3::Prof& p, double& duration) {
	const int j = lane_j();
	otg3::Block bl;
	bl.a = bl.b = false;
	bl.tmin = 0;
	bool ok1 = true;
	if (active) {
		otg3::position_brake(p.brake, cv, ca, vmax, -vmax, amax, -amax, jmax);
		p.p[0] = cp, p.v[0] = cv, p.a[0] = ca, p.pf = tp, p.vf = tv, p.af = 0.0;
		otg3::brake_finalize(p.brake, p.p[0], p.v[0], p.a[0]);
		ok1 = otg3::step1(p, bl, vmax, -vmax, amax, -amax, jmax);
	}
	if (gany(active && !ok1)) return gany(active && !ok1 && (amax == 0.0 || jmax == 0.0)) ? otg::ERR_ZERO_LIMITS : otg::ERR_EXECUTION_TIME;
	if (n == 1) {
		duration = gget(bl.tmin, 0);
		if (j == 0) p = bl.pmin;
		return otg::WORKING;
	}
	// synchronize (calculator_target.hpp:120-222), as in calculate() above
	const double c0 = active ? bl.tmin : INFINITY, c1 = (active && bl.a) ? bl.aright : INFINITY, c2 = (active && bl.b) ? bl.bright : INFINITY;
	const bool any_interval = gany(active && (bl.a || bl.b));
	double cand[3][G];
	UNROLL for (int d = 0; d < G; d++) {
		cand[0][d] = gget(c0, d);
		cand[1][d] = gget(c1, d);
		cand[2][d] = gget(c2, d);
	}
	int blocked = 0;
	UNROLL for (int q = 0; q < 3; q++)
		UNROLL for (int d = 0; d < G; d++)
			if (active && otg3::is_blocked(bl, cand[q][d])) blocked |= 1 << (q * G + d);
	blocked = gor(blocked);
	int best_rank = 1 << 20, best_q = -1, best_d = -1;
	UNROLL for (int q = 0; q < 3; q++)
		UNROLL for (int d = 0; d < G; d++) {
			if (d >= n || (!any_interval && q > 0)) continue;
			const double t = cand[q][d];
			int rank = 0;
			UNROLL for (int oq = 0; oq < 3; oq++)
				UNROLL for (int od = 0; od < G; od++) {
					if (od >= n || (!any_interval && oq > 0)) continue;
					const bool before = (oq < q) || (oq == q && od < d);
					if (cand[oq][od] < t || (cand[oq][od] == t && before)) rank++;
				}
			if (rank < n - 1 || rank >= best_rank) continue;
			if (((blocked >> (q * G + d)) & 1) || t < 0.0 || isinf(t)) continue;
			best_rank = rank, best_q = q, best_d = d;
		}
	if (best_d < 0) return otg::ERR_SYNCHRONIZATION;
	const int limiting = best_d;
	{
		double t = 0;
		UNROLL for (int q = 0; q < 3; q++)
			UNROLL for (int d = 0; d < G; d++)
				if (q == best_q && d == best_d) t = cand[q][d];
		duration = t;
	}
	if (j == limiting) p = best_q == 0 ? bl.pmin : best_q == 1 ? bl.aprof : bl.bprof;
	if (duration > 7.6e3) return otg::ERR_TRAJECTORY_DURATION;
	if (duration == 0.0) {
		if (active) p = bl.pmin;
		return otg::WORKING;
	}
	// phase synchronisation (calculator_target.hpp:398-467), collinearity test :46-118 with the jerk as control
	{
		double plt[7];
		UNROLL for (int i = 0; i < 7; i++) plt[i] = gget(p.t[i], limiting);
		const int pl_dir = ggeti(p.direction, limiting), pl_cs = ggeti(p.control_signs, limiting), pl_lim = ggeti(p.limits, limiting);
		const double pd = tp - cp;
		int which = -1;
		if (active) {
			if (fabs(pd) > otg::EPS)
				which = 0;
			else if (fabs(cv) > otg::EPS)
				which = 1;
			else if (fabs(ca) > otg::EPS)
				which = 2;
			else if (fabs(tv) > otg::EPS)
				which = 3;
		}
		const unsigned has = gbits(which >= 0);
		if (has) {
			const int scale_dof = __ffs((int)has) - 1;
			const int w = ggeti(which, scale_dof);
			const double sv = w == 0 ? pd : w == 1 ? cv : w == 2 ? ca : tv;
			const double scale = gget(sv, scale_dof);
			const double pd_scale = gget(pd, scale_dof) / scale, v0_scale = gget(cv, scale_dof) / scale,
						 vf_scale = gget(tv, scale_dof) / scale, a0_scale = gget(ca, scale_dof) / scale, af_scale = 0.0 / scale;
			const double scale_limiting = gget(sv, limiting);
			const double jmax_lim = gget(jmax, limiting);
			const double control_limiting = (pl_dir == otg3::UP) ? jmax_lim : -jmax_lim;
			const bool off = active && (fabs(pd - pd_scale * sv) > otg::EPS || fabs(cv - v0_scale * sv) > otg::EPS ||
										fabs(ca - a0_scale * sv) > otg::EPS || fabs(tv - vf_scale * sv) > otg::EPS ||
										fabs(0.0 - af_scale * sv) > otg::EPS);
			if (!gany(off)) {
				const double npc = control_limiting * sv / scale_limiting;
				bool lane_ok = true;
				if (active && j != limiting) {
					const double t_profile = duration - p.brake.duration - 0.0;
					UNROLL for (int i = 0; i < 7; i++) p.t[i] = plt[i];
					p.control_signs = pl_cs;
					if (pl_cs == otg3::UDDU)
						lane_ok = otg3::check_tj<otg3::UDDU, otg3::L_NONE>(p, t_profile, npc, vmax, -vmax, amax, -amax, jmax);
					else
						lane_ok = otg3::check_tj<otg3::UDUD, otg3::L_NONE>(p, t_profile, npc, vmax, -vmax, amax, -amax, jmax);
					p.limits = pl_lim;
				}
				if (!gany(!lane_ok)) return otg::WORKING;
			}
		}
	}
	// time synchronisation (calculator_target.hpp:469-529)
	bool bad = false;
	if (active && j != limiting) {
		const double t_profile = duration - p.brake.duration - 0.0;
		if (fabs(t_profile - bl.tmin) < 2 * otg::EPS) {
			p = bl.pmin;
		} else if (bl.a && fabs(t_profile - bl.aright) < 2 * otg::EPS) {
			p = bl.aprof;
		} else if (bl.b && fabs(t_profile - bl.bright) < 2 * otg::EPS) {
			p = bl.bprof;
		} else if (!otg3::step2(p, t_profile, vmax, -vmax, amax, -amax, jmax)) {
			bad = true;
		}
	}
	return gany(bad) ? otg::ERR_SYNCHRONIZATION : otg::WORKING;
}

// Ruckig::update (ruckig.hpp:180-216), jerk-limited
DI int ruckig_update(LaneGen3& g, bool active, int n, double dt, double vmax, double amax, double epoch) {
	int result = otg::WORKING;
	g.replanned = 0;
	const bool differs = gany(active && !(g.in_cp == g.ci_cp && g.in_cv == g.ci_cv && g.in_ca == g.ci_ca && g.in_tp == g.ci_tp &&
										  g.in_tv == g.ci_tv)) ||
						 g.ci_epoch != epoch || !g.ci_init;
	if (differs) {
		const bool invalid = active && (isnan(g.jmax) || g.jmax < 0.0 || isnan(amax) || amax < 0.0 || isnan(vmax) || vmax < 0.0 || isnan(g.in_ca) ||
										isnan(g.in_cv) || isnan(g.in_tv) || isnan(g.in_cp) || isnan(g.in_tp) || g.in_tv > vmax || g.in_tv < -vmax);
		if (gany(invalid)) return otg::ERR_INVALID_INPUT;
		// (a failed calculation leaves the stored rows alone: the wrapper then zeroes the input velocity / acceleration
		// (OTG_joints.cpp:141-149), so the next update plans again before anything is sampled)
		otg3::Prof p = g.p;
		double duration = g.duration;
		result = calculate3(active, n, g.in_cp, g.in_cv, g.in_ca, g.in_tp, g.in_tv, vmax, amax, g.jmax, p, duration);
		if (result != otg::WORKING) return result;
		g.p = p, g.duration = duration;
		g.replanned = 1;
		g.ci_cp = g.in_cp, g.ci_cv = g.in_cv, g.ci_ca = g.in_ca, g.ci_tp = g.in_tp, g.ci_tv = g.in_tv;
		g.ci_epoch = epoch;
		g.ci_init = 1;
		g.time = 0.0;
	}
	g.time += dt;
	if (active) {
		otg3::at_time(g.p, g.duration, g.time, g.np, g.nv, g.na);
		g.ci_cp = g.np, g.ci_cv = g.nv, g.ci_ca = g.na;
	}
	if (g.time > g.duration) return otg::FINISHED;
	return result;
}

// Ruckig::update (ruckig.hpp:180-216)
DI int ruckig_update(LaneGen& g, bool active, int n, double dt, double vmax, double amax, double epoch) {
	int result = otg::WORKING;
	g.replanned = 0;
	const bool differs = gany(active && !(g.in_cp == g.ci_cp && g.in_cv == g.ci_cv && g.in_ca == g.ci_ca && g.in_tp == g.ci_tp &&
										  g.in_tv == g.ci_tv)) ||
						 g.ci_epoch != epoch || !g.ci_init;
	if (differs) {
		const bool invalid = active && (isnan(amax) || amax < 0.0 || isnan(vmax) || vmax < 0.0 || isnan(g.in_ca) || isnan(g.in_cv) ||
										isnan(g.in_tv) || isnan(g.in_cp) || isnan(g.in_tp) || g.in_tv > vmax || g.in_tv < -vmax);
		if (gany(invalid)) return otg::ERR_INVALID_INPUT;
		Dof f = g.f;
		Prof p = g.p;
		double duration = 0;
		result = calculate(active, n, g.in_cp, g.in_cv, g.in_ca, g.in_tp, g.in_tv, vmax, amax, f, p, duration);
		if (result != otg::WORKING) return result;
		g.f = f, g.p = p, g.duration = duration;
		g.ci_cp = g.in_cp, g.ci_cv = g.in_cv, g.ci_ca = g.in_ca, g.ci_tp = g.in_tp, g.ci_tv = g.in_tv;
		g.ci_epoch = epoch;
		g.ci_init = 1;
		g.time = 0.0;
		g.replanned = 1;
	}
	g.time += dt;
	if (active) {
		otg::at_time(g.f, g.p, g.duration, g.time, g.np, g.nv, g.na);
		g.ci_cp = g.np, g.ci_cv = g.nv, g.ci_ca = g.na;
	}
	if (g.time > g.duration) return otg::FINISHED;
	return result;
}

// OTG_joints::setGoalPositionAndVelocity (OTG_joints.cpp:98-116)
template <class LG> DI void joints_set_goal(LG& g, bool active, int n, double gp, double gv) {
	if (g.target_set && gapprox(gp, g.in_tp, 0, n, 1e-12) && gapprox(gv, g.in_tv, 0, n, 1e-12)) return;
	g.goal_reached = 0;
	g.target_set = 1;
	if (active) g.in_tp = gp, g.in_tv = gv;
}

// OTG_6dof_cartesian::setGoalPositionAndLinearVelocity (OTG_6dof_cartesian.cpp:140-149): lanes 0-2
template <class LG> DI void cart_set_goal_position(LG& g, double gp, double gv) {
	if ((g.target_set & 1) && gapprox(gp, g.in_tp, 0, 3, 1e-3) && gapprox(gv, g.in_tv, 0, 3, 1e-3)) return;
	g.goal_reached = 0;
	g.target_set |= 1;
	if (lane_j() < 3) g.in_tp = gp, g.in_tv = gv;
}
// OTG_6dof_cartesian::setGoalOrientationAndAngularVelocity (OTG_6dof_cartesian.cpp:151-185); the
// rotation algebra is done redundantly by every lane, lanes 3-5 keep their component
template <class LG> DI void cart_set_goal_orientation(LG& g, const double* gR, const double* gw) {
	if ((g.target_set & 2) && otg::approx9(g.goal_R, gR, 9, 1e-3) && otg::approx9(g.goal_w, gw, 3, 1e-3)) return;
	const int j = lane_j();
	g.goal_reached = 0;
	g.target_set |= 2;
	double local[9], new_ref[9], R_n2p[9], tmp[3], ref_to_goal[9];
	otg::vec_to_rot(gget(g.np, 3), gget(g.np, 4), gget(g.np, 5), local);
	otg::mat3_mul(g.ref, local, new_ref);
	otg::mat3_tmul(new_ref, g.ref, R_n2p);
	UNROLL for (int i = 0; i < 9; i++) g.ref[i] = new_ref[i], g.goal_R[i] = gR[i];
	UNROLL for (int i = 0; i < 3; i++) g.goal_w[i] = gw[i];
	otg::mat3_vec(R_n2p, gget(g.nv, 3), gget(g.nv, 4), gget(g.nv, 5), tmp);
	if (j >= 3 && j < 6) g.np = 0, g.nv = j == 3 ? tmp[0] : j == 4 ? tmp[1] : tmp[2];
	otg::mat3_vec(R_n2p, gget(g.na, 3), gget(g.na, 4), gget(g.na, 5), tmp);
	if (j >= 3 && j < 6) g.na = j == 3 ? tmp[0] : j == 4 ? tmp[1] : tmp[2];
	if (j < 6) g.in_cp = g.np, g.in_cv = g.nv, g.in_ca = g.na;	 // _output.pass_to_input(_input)
	otg::mat3_tmul(g.ref, g.goal_R, ref_to_goal);
	otg::rot_to_vec(ref_to_goal, tmp);
	if (j >= 3 && j < 6) g.in_tp = j == 3 ? tmp[0] : j == 4 ? tmp[1] : tmp[2];
	otg::mat3_tvec(g.ref, g.goal_w, tmp);
	if (j >= 3 && j < 6) g.in_tv = j == 3 ? tmp[0] : j == 4 ? tmp[1] : tmp[2];
}

// update() of both wrappers after the goal was set (OTG_joints.cpp:118-150,
// OTG_6dof_cartesian.cpp:187-224); for the JointTask's Finished-with-velocity branch see
// sai2b_otg_core.hpp: joints_finish
template <class LG> DI void update(LG& g, bool cart, bool active, int n, double dt, double vmax, double amax, double epoch) {
	if (g.goal_reached) return;
	const double pp = g.np, pv = g.nv, pa = g.na;
	g.result = ruckig_update(g, active, n, dt, vmax, amax, epoch);
	if (g.result == otg::FINISHED) {
		const double vn = sqrt(gsum(active ? g.nv * g.nv : 0.0));
		if (vn < 1e-3) {
			g.goal_reached = 1;
		} else if (cart) {
			const double zeros[3] = {0, 0, 0};
			double gR[9];
			UNROLL for (int i = 0; i < 9; i++) gR[i] = g.goal_R[i];
			cart_set_goal_position(g, g.in_tp, 0.0);
			cart_set_goal_orientation(g, gR, zeros);
		} else {
			joints_set_goal(g, active, n, g.in_tp, 0.0);
		}
	} else if (g.result == otg::WORKING) {
		if (active) g.in_cp = g.np, g.in_cv = g.nv, g.in_ca = g.na;
	} else if (active) {
		g.np = pp, g.nv = pv, g.na = pa, g.in_cv = 0, g.in_ca = 0;
	}
}

}  // namespace otgg
}  // namespace sai2b
