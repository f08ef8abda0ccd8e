// sai2b_otg_core.hpp — per-robot internal online trajectory generation (OTG), device code.
//
// What the reference does per task and tick when its internal OTG is on (the default:
// JointTask.h:38, MotionForceTask.h:67): OTG_joints / OTG_6dof_cartesian
// (src/helper_modules/OTG_joints.cpp, OTG_6dof_cartesian.cpp) feed ruckig 0.10.1's
// acceleration-limited position interface (max_jerk = inf: ruckig/src/ruckig/brake.cpp:79-99,
// position-second-step1.cpp, position-second-step2.cpp, ruckig/include/ruckig/block.hpp,
// calculator_target.hpp, trajectory.hpp, ruckig.hpp:180-216) and the control law tracks the
// generator's next state instead of the goal (JointTask.cpp:313-320, MotionForceTask.cpp:394-407).
//
// Here one lane owns one robot. A DoF's profile is kept in compact form: for the second-order
// interface only the phases 0, 1, 2 and 6 of ruckig's seven can have a non-zero duration and only
// 0, 2 and 6 a non-zero acceleration, so (t0 t1 t2 t6, a0 a2 a6) plus the start state determine it;
// positions and velocities at the phase boundaries are re-derived with ruckig's own recurrence
// (profile.hpp:337-340), so sampling reproduces its values.
//
// The header is plain C++ apart from SAI2B_HD: tests/cpp/otg_core_test.cpp compiles the same code
// for the host to check the logic against the oracle on a machine without a GPU. The product only
// ever runs it inside otg_kernel (sai2b_otg.hip).
#pragma once
#include <math.h>

#include "../../include/sai2b_detmath.h"

#ifndef SAI2B_HD
#ifdef __HIPCC__
#define SAI2B_HD __host__ __device__ __forceinline__
#else
#define SAI2B_HD inline
#endif
#endif

namespace sai2b {
namespace otg {

#ifndef SAI2B_OTG_MAXD
#define SAI2B_OTG_MAXD 7  // DoFs of the largest generator; the 8-joint build of the library sets 8 (sai2b_params.h: OTG_MD)
#endif
constexpr int MAXD = SAI2B_OTG_MAXD;
constexpr int WORKING = 0, FINISHED = 1, ERR_INVALID_INPUT = -100, ERR_TRAJECTORY_DURATION = -101, ERR_ZERO_LIMITS = -104,
			  ERR_EXECUTION_TIME = -110, ERR_SYNCHRONIZATION = -111;
constexpr double EPS = 2.220446049250313e-16;

// start state of a DoF's main profile (after the brake pre-trajectory) and its target
struct Dof {
	double brake_t, brake_a, brake_p, brake_v;	// BrakeProfile, second order (brake.hpp:25-28)
	double p0, v0, pf, vf;
};

struct Prof {
	double t0, t1, t2, t6, a0, a2, a6;
	double dur;	 // t_sum.back()
	int dir;	 // 0 UP, 1 DOWN
};

SAI2B_HD double brake_duration(const Dof& d) { return d.brake_t > 0.0 ? d.brake_t : 0.0; }

// Profile::check_for_second_order<UDDU, .> (profile.hpp:307-350) on the compact form
SAI2B_HD bool check(Prof& pr, const Dof& d, double aUp, double aDown, double vMax, double vMin) {
	if (pr.t0 < 0) return false;
	if (pr.t1 < 0) return false;
	const double ts1 = pr.t0 + pr.t1;
	if (pr.t2 < 0) return false;
	const double ts2 = ts1 + pr.t2;
	if (pr.t6 < 0) return false;
	const double ts6 = ts2 + pr.t6;
	if (ts6 > 1e12) return false;
	pr.dur = ts6;
	pr.a0 = pr.t0 > 0 ? aUp : 0;
	pr.a2 = pr.t2 > 0 ? aDown : 0;
	pr.a6 = pr.t6 > 0 ? aUp : 0;
	pr.dir = (vMax > 0) ? 0 : 1;
	const double vUppLim = (pr.dir == 0 ? vMax : vMin) + 1e-12;
	const double vLowLim = (pr.dir == 0 ? vMin : vMax) - 1e-12;
	const double v1 = d.v0 + pr.t0 * pr.a0;
	const double p1 = d.p0 + pr.t0 * (d.v0 + pr.t0 * pr.a0 / 2);
	const double p2 = p1 + pr.t1 * v1;
	const double v3 = v1 + pr.t2 * pr.a2;
	const double p3 = p2 + pr.t2 * (v1 + pr.t2 * pr.a2 / 2);
	const double v7 = v3 + pr.t6 * pr.a6;
	const double p7 = p3 + pr.t6 * (v3 + pr.t6 * pr.a6 / 2);
	return fabs(p7 - d.pf) < 1e-8 && fabs(v7 - d.vf) < 1e-8 && v1 <= vUppLim && v3 <= vUppLim &&
		   v1 >= vLowLim && v3 >= vLowLim;
}

// Block (block.hpp): fastest profile plus up to two blocked duration intervals
struct Block {
	Prof pmin, aprof, bprof;
	double tmin, aleft, aright, bleft, bright;
	bool a, b;
};

SAI2B_HD void set_min(Block& bl, const Prof& p, double bd) {
	bl.pmin = p;
	bl.tmin = p.dur + bd + 0.0;
	bl.a = bl.b = false;
}
SAI2B_HD void interval(double& left, double& right, Prof& prof, const Prof& pl, const Prof& pr, double bd) {
	const double ld = pl.dur + bd + 0.0, rd = pr.dur + bd + 0.0;
	if (ld < rd) {
		left = ld, right = rd, prof = pr;
	} else {
		left = rd, right = ld, prof = pl;
	}
}
SAI2B_HD bool is_blocked(const Block& b, double t) {
	return (t < b.tmin) || (b.a && b.aleft < t && t < b.aright) || (b.b && b.bleft < t && t < b.bright);
}

// element access without dynamic register indexing
template <int K> SAI2B_HD Prof pick(const Prof (&v)[K], int i) {
	Prof r = v[0];
#pragma unroll
	for (int k = 1; k < K; k++)
		if (k == i) r = v[k];
	return r;
}

// Block::calculate_block<N, true> (block.hpp:61-134); v holds `count` accepted profiles
SAI2B_HD bool calculate_block(Block& bl, Prof (&v)[6], int count, double bd) {
	if (count == 1) {
		set_min(bl, v[0], bd);
		return true;
	} else if (count == 2) {
		if (fabs(v[0].dur - v[1].dur) < 8 * EPS) {
			set_min(bl, v[0], bd);
			return true;
		}
		const int imin = (v[0].dur < v[1].dur) ? 0 : 1;
		set_min(bl, imin == 0 ? v[0] : v[1], bd);
		bl.a = true;
		if (imin == 0)
			interval(bl.aleft, bl.aright, bl.aprof, v[0], v[1], bd);
		else
			interval(bl.aleft, bl.aright, bl.aprof, v[1], v[0], bd);
		return true;
	} else if (count == 4) {
		if (fabs(v[0].dur - v[1].dur) < 32 * EPS && v[0].dir != v[1].dir) {
			v[1] = v[2], v[2] = v[3];
		} else if (fabs(v[2].dur - v[3].dur) < 256 * EPS && v[2].dir != v[3].dir) {
		} else if (fabs(v[0].dur - v[3].dur) < 256 * EPS && v[0].dir != v[3].dir) {
		} else {
			return false;
		}
		count = 3;
	} else if (count % 2 == 0) {
		return false;
	}
	int imin = 0;
#pragma unroll
	for (int i = 1; i < 5; i++)
		if (i < count && v[i].dur < pick(v, imin).dur) imin = i;
	set_min(bl, pick(v, imin), bd);
	if (count == 3) {
		bl.a = true;
		interval(bl.aleft, bl.aright, bl.aprof, pick(v, (imin + 1) % 3), pick(v, (imin + 2) % 3), bd);
		return true;
	} else if (count == 5) {
		const Prof e1 = pick(v, (imin + 1) % 5), e2 = pick(v, (imin + 2) % 5), e3 = pick(v, (imin + 3) % 5),
				   e4 = pick(v, (imin + 4) % 5);
		bl.a = bl.b = true;
		if (e1.dir == e2.dir) {
			interval(bl.aleft, bl.aright, bl.aprof, e1, e2, bd);
			interval(bl.bleft, bl.bright, bl.bprof, e3, e4, bd);
		} else {
			interval(bl.aleft, bl.aright, bl.aprof, e1, e4, bd);
			interval(bl.bleft, bl.bright, bl.bprof, e2, e3, bd);
		}
		return true;
	}
	return false;
}

// PositionSecondOrderStep1 (position-second-step1.cpp); the profile being tried lives in `cur`
struct Step1 {
	Prof v[6];
	int count;
};
SAI2B_HD void s1_accept(Step1& s, const Prof& cur) {
#pragma unroll
	for (int k = 0; k < 6; k++)
		if (k == s.count) s.v[k] = cur;
	s.count++;
}
SAI2B_HD void s1_time_acc0(Step1& s, const Dof& d, double pd, double vMax, double vMin, double aMax, double aMin) {
	Prof cur;
	cur.t0 = (-d.v0 + vMax) / aMax;
	cur.t1 = (aMin * d.v0 * d.v0 - aMax * d.vf * d.vf) / (2 * aMax * aMin * vMax) +
			 vMax * (aMax - aMin) / (2 * aMax * aMin) + pd / vMax;
	cur.t2 = (d.vf - vMax) / aMin;
	cur.t6 = 0;
	if (check(cur, d, aMax, aMin, vMax, vMin)) s1_accept(s, cur);
}
SAI2B_HD void s1_time_none(Step1& s, const Dof& d, double pd, double vMax, double vMin, double aMax, double aMin,
						   bool return_after_found) {
	double h1 = (aMax * d.vf * d.vf - aMin * d.v0 * d.v0 - 2 * aMax * aMin * pd) / (aMax - aMin);
	if (h1 >= 0.0) {
		h1 = sqrt(h1);
		Prof cur;
		cur.t6 = 0;
		cur.t0 = -(d.v0 + h1) / aMax;
		cur.t1 = 0;
		cur.t2 = (d.vf + h1) / aMin;
		if (check(cur, d, aMax, aMin, vMax, vMin)) {
			s1_accept(s, cur);
			if (return_after_found) return;
		}
		cur.t0 = (-d.v0 + h1) / aMax;
		cur.t1 = 0;
		cur.t2 = (d.vf - h1) / aMin;
		if (check(cur, d, aMax, aMin, vMax, vMin)) s1_accept(s, cur);
	}
}
// get_profile (position-second-step1.cpp:100-136); limits are > 0 (checked on the host)
SAI2B_HD bool step1(const Dof& d, Block& bl, double vMaxIn, double aMaxIn) {
	Step1 s;
	s.count = 0;
	const double vMinIn = -vMaxIn, aMinIn = -aMaxIn;
	const double pd = d.pf - d.p0;
	if (fabs(d.vf) < EPS) {
		const double vMax = (pd >= 0) ? vMaxIn : vMinIn, vMin = (pd >= 0) ? vMinIn : vMaxIn;
		const double aMax = (pd >= 0) ? aMaxIn : aMinIn, aMin = (pd >= 0) ? aMinIn : aMaxIn;
		s1_time_none(s, d, pd, vMax, vMin, aMax, aMin, true);
		if (s.count == 0) s1_time_acc0(s, d, pd, vMax, vMin, aMax, aMin);
		if (s.count == 0) s1_time_none(s, d, pd, vMin, vMax, aMin, aMax, true);
		if (s.count == 0) s1_time_acc0(s, d, pd, vMin, vMax, aMin, aMax);
	} else {
		s1_time_none(s, d, pd, vMaxIn, vMinIn, aMaxIn, aMinIn, false);
		s1_time_none(s, d, pd, vMinIn, vMaxIn, aMinIn, aMaxIn, false);
		s1_time_acc0(s, d, pd, vMaxIn, vMinIn, aMaxIn, aMinIn);
		s1_time_acc0(s, d, pd, vMinIn, vMaxIn, aMinIn, aMaxIn);
	}
	return calculate_block(bl, s.v, s.count, brake_duration(d));
}

// PositionSecondOrderStep2 (position-second-step2.cpp)
SAI2B_HD bool s2_time_acc0(Prof& pr, const Dof& d, double tf, double pd, double vd, double vMax, double vMin,
						   double aMax, double aMin) {
	{
		const double h1 =
			sqrt((2 * aMax * (pd - tf * d.vf) - 2 * aMin * (pd - tf * d.v0) + vd * vd) / (aMax * aMin) + tf * tf);
		pr.t0 = (aMax * vd - aMax * aMin * (tf - h1)) / (aMax * (aMax - aMin));
		pr.t1 = h1;
		pr.t2 = tf - (pr.t0 + h1);
		pr.t6 = 0;
		if (check(pr, d, aMax, aMin, vMax, vMin)) return true;
	}
	{
		const double h1 = (-vd + aMax * tf);
		pr.t0 = -vd * vd / (2 * aMax * h1) + (pd - d.v0 * tf) / h1;
		pr.t1 = -vd / aMax + tf;
		pr.t2 = 0;
		pr.t6 = tf - (pr.t0 + pr.t1);
		if (check(pr, d, aMax, aMin, vMax, vMin)) return true;
	}
	{
		pr.t0 = 0;
		pr.t1 = -vd / aMax + tf;
		pr.t2 = 0;
		pr.t6 = vd / aMax;
		if (check(pr, d, aMax, aMin, vMax, vMin)) return true;
	}
	return false;
}
SAI2B_HD bool s2_time_none(Prof& pr, const Dof& d, double tf, double pd, double vd, double vMax, double vMin,
						   double aMax, double aMin) {
	if (fabs(d.v0) < EPS && fabs(d.vf) < EPS && fabs(pd) < EPS) {
		pr.t0 = 0;
		pr.t1 = tf;
		pr.t2 = 0;
		pr.t6 = 0;
		if (check(pr, d, aMax, aMin, vMax, vMin)) return true;
	}
	{
		const double h1 = 2 * (d.vf * tf - pd);
		pr.t0 = h1 / vd;
		pr.t1 = tf - pr.t0;
		pr.t2 = 0;
		pr.t6 = 0;
		const double af = vd * vd / h1;
		if ((aMin - 1e-12 < af) && (af < aMax + 1e-12) && check(pr, d, af, -af, vMax, vMin)) return true;
	}
	return false;
}
SAI2B_HD bool s2_check_all(Prof& pr, const Dof& d, double tf, double pd, double vd, double vMax, double vMin,
						   double aMax, double aMin) {
	return s2_time_acc0(pr, d, tf, pd, vd, vMax, vMin, aMax, aMin) ||
		   s2_time_none(pr, d, tf, pd, vd, vMax, vMin, aMax, aMin);
}
SAI2B_HD bool step2(Prof& pr, const Dof& d, double tf, double vMax, double aMax) {
	const double pd = d.pf - d.p0, vd = d.vf - d.v0;
	const double vMin = -vMax, aMin = -aMax;
	if (pd > 0)
		return s2_check_all(pr, d, tf, pd, vd, vMax, vMin, aMax, aMin) ||
			   s2_check_all(pr, d, tf, pd, vd, vMin, vMax, aMin, aMax);
	return s2_check_all(pr, d, tf, pd, vd, vMin, vMax, aMin, aMax) ||
		   s2_check_all(pr, d, tf, pd, vd, vMax, vMin, aMax, aMin);
}

// InputParameter subset (input_parameter.hpp:73-104): current and target state; the limits are
// batch-uniform and passed alongside
struct Input {
	double cp[MAXD], cv[MAXD], ca[MAXD], tp[MAXD], tv[MAXD];
};
struct Traj {
	Dof dof[MAXD];
	Prof prof[MAXD];
	double duration;
};

SAI2B_HD double sel(const double (&v)[MAXD], int i) {
	double r = v[0];
#pragma unroll
	for (int k = 1; k < MAXD; k++)
		if (k == i) r = v[k];
	return r;
}

// InputParameter::validate(false, true) (input_parameter.hpp:153-330) with max_jerk = inf
SAI2B_HD bool validate(const Input& in, int n, const double (&vmax)[MAXD], const double (&amax)[MAXD]) {
	bool ok = true;
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) {
			if (isnan(amax[d]) || amax[d] < 0.0 || isnan(vmax[d]) || vmax[d] < 0.0) ok = false;
			if (isnan(in.ca[d]) || isnan(in.cv[d]) || isnan(in.tv[d]) || isnan(in.cp[d]) || isnan(in.tp[d]))
				ok = false;
			if (in.tv[d] > vmax[d] || in.tv[d] < -vmax[d]) ok = false;
		}
	return ok;
}

// is_input_collinear (calculator_target.hpp:46-118), every DoF phase-synchronised, max_jerk = inf
SAI2B_HD bool collinear(const Input& in, int n, const double (&amax)[MAXD], int limiting_direction,
						int limiting_dof, double (&npc)[MAXD]) {
	double pd[MAXD], sv[MAXD];
	int which = -1, scale_dof = -1;	 // which vector scales: 0 pd, 1 cv, 2 ca, 3 tv
#pragma unroll 1
	for (int d = 0; d < n; d++) {
		pd[d] = in.tp[d] - in.cp[d];
		if (scale_dof < 0) {
			if (fabs(pd[d]) > EPS)
				which = 0, scale_dof = d;
			else if (fabs(in.cv[d]) > EPS)
				which = 1, scale_dof = d;
			else if (fabs(in.ca[d]) > EPS)
				which = 2, scale_dof = d;
			else if (fabs(in.tv[d]) > EPS)
				which = 3, scale_dof = d;
		}
	}
	if (scale_dof < 0) return false;
#pragma unroll 1
	for (int d = 0; d < n; d++) sv[d] = which == 0 ? pd[d] : which == 1 ? in.cv[d] : which == 2 ? in.ca[d] : in.tv[d];
	const double scale = sv[scale_dof];
	const double pd_scale = pd[scale_dof] / scale, v0_scale = in.cv[scale_dof] / scale,
				 vf_scale = in.tv[scale_dof] / scale, a0_scale = in.ca[scale_dof] / scale, af_scale = 0.0 / scale;
	const double scale_limiting = sv[limiting_dof];
	const double control_limiting = (limiting_direction == 0) ? amax[limiting_dof] : -amax[limiting_dof];
#pragma unroll 1
	for (int d = 0; d < n; d++) {
		const double cs = sv[d];
		if (fabs(pd[d] - pd_scale * cs) > EPS || fabs(in.cv[d] - v0_scale * cs) > EPS ||
			fabs(in.ca[d] - a0_scale * cs) > EPS || fabs(in.tv[d] - vf_scale * cs) > EPS ||
			fabs(0.0 - af_scale * cs) > EPS)
			return false;
		npc[d] = control_limiting * cs / scale_limiting;
	}
	return true;
}

// TargetCalculator::calculate (calculator_target.hpp:249-532): acceleration-limited position
// interface, synchronisation Phase (with Time as its fallback), continuous durations
SAI2B_HD int calculate(const Input& in, int n, const double (&vmax)[MAXD], const double (&amax)[MAXD], Traj& tr) {
	Block bl[MAXD];
	bool failed = false;
	// per-DoF loops stay rolled: the planner runs only when a goal changes, and compact code (dynamic
	// indexing into per-lane scratch) beats seven inlined copies of step 1 / step 2
#pragma unroll 1
	for (int d = 0; d < n; d++) {
		{
			Dof& f = tr.dof[d];
			const double vMax = vmax[d], vMin = -vmax[d], aMax = amax[d], aMin = -amax[d];
			// brake.cpp:79-99, brake.hpp:66-75
			f.brake_t = 0.0, f.brake_a = 0.0;
			if (!(aMax == 0.0 || aMin == 0.0)) {
				if (in.cv[d] > vMax) {
					f.brake_a = aMin;
					f.brake_t = (vMax - in.cv[d]) / aMin + 2.2e-14;
				} else if (in.cv[d] < vMin) {
					f.brake_a = aMax;
					f.brake_t = (vMin - in.cv[d]) / aMax + 2.2e-14;
				}
			}
			f.p0 = in.cp[d], f.v0 = in.cv[d], f.pf = in.tp[d], f.vf = in.tv[d];
			if (f.brake_t > 0.0) {
				const double t = f.brake_t, ps = f.p0, vs = f.v0, ab = f.brake_a;
				f.brake_p = ps, f.brake_v = vs;
				f.p0 = ps + t * (vs + t * (ab / 2 + t * 0.0 / 6));
				f.v0 = vs + t * (ab + t * 0.0 / 2);
			}
			if (!step1(f, bl[d], vMax, aMax)) failed = true;
		}
	}
	if (failed) return ERR_EXECUTION_TIME;

	if (n == 1) {
		tr.duration = bl[0].tmin;
		tr.prof[0] = bl[0].pmin;
		return WORKING;
	}

	// synchronize (calculator_target.hpp:120-222): candidate durations (index q*n + d: t_min of every
	// DoF, then the right ends of its blocked intervals) in stable ascending order, tried from
	// position n-1; the first one no DoF blocks wins
	int limiting = -1;
	{
		double cand[3 * MAXD];
		bool any_interval = false;
#pragma unroll 1
		for (int d = 0; d < n; d++) {
			cand[d] = bl[d].tmin;
			cand[n + d] = bl[d].a ? bl[d].aright : INFINITY;
			cand[2 * n + d] = bl[d].b ? bl[d].bright : INFINITY;
			any_interval |= bl[d].a || bl[d].b;
		}
		const int total = any_interval ? 3 * n : n;	 // without intervals only the n t_min values are ordered
		int best_rank = 1 << 20, best_c = -1;
#pragma unroll 1
		for (int c = 0; c < total; c++) {
			const double t = cand[c];
			int rank = 0;
#pragma unroll 1
			for (int o = 0; o < total; o++)
				if (cand[o] < t || (cand[o] == t && o < c)) rank++;
			if (rank < n - 1 || rank >= best_rank) continue;
			bool blocked = false;
#pragma unroll 1
			for (int d = 0; d < n; d++)
				if (is_blocked(bl[d], t)) blocked = true;
			if (blocked || t < 0.0 || isinf(t)) continue;
			best_rank = rank, best_c = c;
		}
		if (best_c < 0) return ERR_SYNCHRONIZATION;
		tr.duration = cand[best_c];
		limiting = best_c % n;
		const int quot = best_c / n;
		tr.prof[limiting] = quot == 0 ? bl[limiting].pmin : quot == 1 ? bl[limiting].aprof : bl[limiting].bprof;
	}

	if (tr.duration > 7.6e3) return ERR_TRAJECTORY_DURATION;
	if (tr.duration == 0.0) {
#pragma unroll 1
		for (int d = 0; d < n; d++) tr.prof[d] = bl[d].pmin;
		return WORKING;
	}

	// phase synchronisation (calculator_target.hpp:398-467)
	{
		const Prof pl = tr.prof[limiting];
		double npc[MAXD];
		if (collinear(in, n, amax, pl.dir, limiting, npc)) {
			bool found = true;
#pragma unroll 1
			for (int d = 0; d < n; d++)
				if (d != limiting) {
					Prof& p = tr.prof[d];
					p.t0 = pl.t0, p.t1 = pl.t1, p.t2 = pl.t2, p.t6 = pl.t6;
					const double aUp = npc[d], aDown = -npc[d], aMax = amax[d], aMin = -amax[d];
					const bool within = (aMin - 1e-12 < aUp) && (aUp < aMax + 1e-12) && (aMin - 1e-12 < aDown) &&
										(aDown < aMax + 1e-12);
					found &= within && check(p, tr.dof[d], aUp, aDown, vmax[d], -vmax[d]);
				}
			if (found) return WORKING;
		}
	}

	// time synchronisation (calculator_target.hpp:469-529)
	bool bad = false;
#pragma unroll 1
	for (int d = 0; d < n; d++)
		if (d != limiting) {
			Prof& p = tr.prof[d];
			const double t_profile = tr.duration - brake_duration(tr.dof[d]) - 0.0;
			if (fabs(t_profile - bl[d].tmin) < 2 * EPS) {
				p = bl[d].pmin;
			} else if (bl[d].a && fabs(t_profile - bl[d].aright) < 2 * EPS) {
				p = bl[d].aprof;
			} else if (bl[d].b && fabs(t_profile - bl[d].bright) < 2 * EPS) {
				p = bl[d].bprof;
			} else if (!step2(p, tr.dof[d], t_profile, vmax[d], amax[d])) {
				bad = true;
			}
		}
	return bad ? ERR_SYNCHRONIZATION : WORKING;
}

SAI2B_HD void integrate0(double t, double p0, double v0, double a0, double& p, double& v, double& a) {
	p = p0 + t * (v0 + t * (a0 / 2 + t * 0.0 / 6));
	v = v0 + t * (a0 + t * 0.0 / 2);
	a = a0 + t * 0.0;
}

// Trajectory::at_time for one DoF (trajectory.hpp:65-142)
SAI2B_HD void at_time(const Dof& d, const Prof& pr, double duration, double time, double& p, double& v, double& a) {
	// boundary states by ruckig's recurrence (profile.hpp:337-340)
	const double v1 = d.v0 + pr.t0 * pr.a0;
	const double p1 = d.p0 + pr.t0 * (d.v0 + pr.t0 * pr.a0 / 2);
	const double p2 = p1 + pr.t1 * v1;
	const double v3 = v1 + pr.t2 * pr.a2;
	const double p3 = p2 + pr.t2 * (v1 + pr.t2 * pr.a2 / 2);
	const double v7 = v3 + pr.t6 * pr.a6;
	const double p7 = p3 + pr.t6 * (v3 + pr.t6 * pr.a6 / 2);
	const double ts0 = pr.t0, ts1 = ts0 + pr.t1, ts2 = ts1 + pr.t2, ts6 = pr.dur;
	const double bd = brake_duration(d);
	if (time >= duration) {
		integrate0(time - (bd + ts6), p7, v7, 0.0, p, v, a);
		return;
	}
	double t = time;
	if (bd > 0) {
		if (t < bd) {
			integrate0(t, d.brake_p, d.brake_v, d.brake_a, p, v, a);
			return;
		}
		t -= bd;
	}
	if (t >= ts6) {
		integrate0(t - ts6, p7, v7, 0.0, p, v, a);
	} else if (ts0 > t) {
		integrate0(t, d.p0, d.v0, pr.a0, p, v, a);
	} else if (ts1 > t) {
		integrate0(t - ts0, p1, v1, 0.0, p, v, a);
	} else if (ts2 > t) {
		integrate0(t - ts1, p2, v1, pr.a2, p, v, a);
	} else {
		integrate0(t - ts2, p3, v3, pr.a6, p, v, a);
	}
}

// ---------------------------------------------------------------------------------------------
// Ruckig<> + the sai2 wrappers. One Gen is the state of one OTG_joints / OTG_6dof_cartesian
// object: `in` = the wrapper's _input, `ci` = Ruckig's current_input, np/nv/na/time/traj = _output.
struct Gen {
	Input in, ci;
	double np[MAXD], nv[MAXD], na[MAXD];
	double time;
	Traj traj;
	int goal_reached, result, target_set, ci_init;
	int replanned;	  // the last ruckig_update computed a new trajectory (OutputParameter::new_calculation)
	double ci_epoch;  // limits generation the stored current_input was planned with
	// OTG_6dof_cartesian only
	double ref[9], goal_R[9], goal_w[3];
};

SAI2B_HD int plan(Gen& g, int n, const double (&vmax)[MAXD], const double (&amax)[MAXD]) {
	if (!validate(g.in, n, vmax, amax)) return ERR_INVALID_INPUT;
	Traj tr;
	const int result = calculate(g.in, n, vmax, amax, tr);
	if (result == WORKING) g.traj = tr;
	return result;
}

template <class G> SAI2B_HD bool input_differs(const G& g, int n, double epoch) {
	bool diff = g.ci_epoch != epoch;
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n)
			diff |= !(g.in.cp[d] == g.ci.cp[d] && g.in.cv[d] == g.ci.cv[d] && g.in.ca[d] == g.ci.ca[d] &&
					  g.in.tp[d] == g.ci.tp[d] && g.in.tv[d] == g.ci.tv[d]);
	return diff;
}

// The wrappers below are written once for both generators: Gen (acceleration-limited, this file) and otg3::Gen
// (jerk-limited, sai2b_otg3_core.hpp). What differs is reached through two overloaded hooks found by argument type:
// sample_dof(g, d) = Trajectory::at_time of DoF d at g.time; plan(g, n, vmax, amax) = validate + calculate into g.traj.
SAI2B_HD void sample_dof(Gen& g, int d) { at_time(g.traj.dof[d], g.traj.prof[d], g.traj.duration, g.time, g.np[d], g.nv[d], g.na[d]); }

// Ruckig::update, second half (ruckig.hpp:205-215): advance along the stored trajectory
template <class G> SAI2B_HD int ruckig_sample(G& g, int n, double dt, int result) {
	g.time += dt;
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) {
			sample_dof(g, d);
			g.ci.cp[d] = g.np[d], g.ci.cv[d] = g.nv[d], g.ci.ca[d] = g.na[d];
		}
	if (g.time > g.traj.duration) return FINISHED;
	return result;
}
// does Ruckig::update have to calculate a new trajectory (ruckig.hpp:194)?
template <class G> SAI2B_HD bool needs_plan(const G& g, int n, double epoch) { return input_differs(g, n, epoch) || !g.ci_init; }

// Ruckig::update (ruckig.hpp:180-216)
template <class G> SAI2B_HD int ruckig_update(G& g, int n, double dt, const double (&vmax)[MAXD], const double (&amax)[MAXD],
						   double epoch) {
	int result = WORKING;
	g.replanned = 0;
	if (needs_plan(g, n, epoch)) {
		result = plan(g, n, vmax, amax);
		if (result != WORKING) return result;  // the stored trajectory stays (the wrapper restores it)
		g.ci = g.in;
		g.ci_epoch = epoch;
		g.ci_init = 1;
		g.time = 0.0;
		g.replanned = 1;
	}
	return ruckig_sample(g, n, dt, result);
}

template <class G> SAI2B_HD void pass_to_input(G& g, int n) {
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) g.in.cp[d] = g.np[d], g.in.cv[d] = g.nv[d], g.in.ca[d] = g.na[d];
}

// Eigen isApprox on the DoF range [lo, hi): |a-b|^2 <= prec^2 min(|a|^2, |b|^2)
SAI2B_HD bool approx_range(const double (&a)[MAXD], const double (&b)[MAXD], int lo, int hi, double prec) {
	double dd = 0, na = 0, nb = 0;
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d >= lo && d < hi) {
			dd += (a[d] - b[d]) * (a[d] - b[d]);
			na += a[d] * a[d];
			nb += b[d] * b[d];
		}
	return dd <= prec * prec * (na < nb ? na : nb);
}

// ---- OTG_joints (OTG_joints.cpp) ----
template <class G> SAI2B_HD void joints_set_goal(G& g, int n, const double (&gp)[MAXD], const double (&gv)[MAXD]) {
	if (g.target_set && approx_range(gp, g.in.tp, 0, n, 1e-12) && approx_range(gv, g.in.tv, 0, n, 1e-12)) return;
	g.goal_reached = 0;
	g.target_set = 1;
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) g.in.tp[d] = gp[d], g.in.tv[d] = gv[d];
}
template <class G> SAI2B_HD void joints_reinitialize(G& g, int n, const double (&x0)[MAXD]) {
	const double zeros[MAXD] = {0, 0, 0, 0, 0, 0, 0};
	joints_set_goal(g, n, x0, zeros);
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) g.np[d] = x0[d], g.nv[d] = 0, g.na[d] = 0;
	pass_to_input(g, n);
}
// `previous_output` of the wrappers' update() (OTG_joints.cpp:123, OTG_6dof_cartesian.cpp:192)
struct Prev {
	double p[MAXD], v[MAXD], a[MAXD];
};
template <class G> SAI2B_HD void save_prev(const G& g, Prev& pv) {
#pragma unroll
	for (int d = 0; d < MAXD; d++) pv.p[d] = g.np[d], pv.v[d] = g.nv[d], pv.a[d] = g.na[d];
}
template <class G> SAI2B_HD void on_error(G& g, int n, const Prev& pv) {	 // OTG_joints.cpp:141-149, OTG_6dof_cartesian.cpp:215-223
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) g.np[d] = pv.p[d], g.nv[d] = pv.v[d], g.na[d] = pv.a[d], g.in.cv[d] = 0, g.in.ca[d] = 0;
}
template <class G> SAI2B_HD double velocity_norm(const G& g, int n) {
	double nrm = 0;
#pragma unroll
	for (int d = 0; d < MAXD; d++)
		if (d < n) nrm += g.nv[d] * g.nv[d];
	return sqrt(nrm);
}
// OTG_joints::update after _otg->update() returned g.result (OTG_joints.cpp:125-149). The
// Finished-with-velocity branch calls setGoalPosition with a member that is never assigned (:129),
// which throws in the reference; it does what the Cartesian wrapper does there (keep the target
// position, zero the target velocity).
template <class G> SAI2B_HD void joints_finish(G& g, int n, const Prev& pv) {
	if (g.result == FINISHED) {
		if (velocity_norm(g, n) < 1e-3) {
			g.goal_reached = 1;
		} else {
			const double zeros[MAXD] = {0, 0, 0, 0, 0, 0, 0};
			double tp[MAXD];
#pragma unroll
			for (int d = 0; d < MAXD; d++) tp[d] = g.in.tp[d];
			joints_set_goal(g, n, tp, zeros);
		}
	} else if (g.result == WORKING) {
		pass_to_input(g, n);
	} else {
		on_error(g, n, pv);
	}
}
// OTG_joints::update (OTG_joints.cpp:118-150)
template <class G> SAI2B_HD void joints_update(G& g, int n, double dt, const double (&vmax)[MAXD], const double (&amax)[MAXD],
							double epoch) {
	if (g.goal_reached) return;
	Prev pv;
	save_prev(g, pv);
	g.result = ruckig_update(g, n, dt, vmax, amax, epoch);
	joints_finish(g, n, pv);
}

// ---- rotations (Eigen AngleAxisd semantics, from the published algorithms) ----
SAI2B_HD void mat3_mul(const double* A, const double* B, double* C) {
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i * 3] * B[j] + A[i * 3 + 1] * B[3 + j] + A[i * 3 + 2] * B[6 + j];
}
SAI2B_HD void mat3_tmul(const double* A, const double* B, double* C) {
#pragma unroll
	for (int i = 0; i < 3; i++)
#pragma unroll
		for (int j = 0; j < 3; j++) C[i * 3 + j] = A[i] * B[j] + A[3 + i] * B[3 + j] + A[6 + i] * B[6 + j];
}
SAI2B_HD void mat3_vec(const double* A, double x0, double x1, double x2, double* y) {
#pragma unroll
	for (int i = 0; i < 3; i++) y[i] = A[i * 3] * x0 + A[i * 3 + 1] * x1 + A[i * 3 + 2] * x2;
}
SAI2B_HD void mat3_tvec(const double* A, const double* x, double* y) {
#pragma unroll
	for (int i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}
// rotation matrix -> angle * axis, through the quaternion (Shepperd's branches)
SAI2B_HD void rot_to_vec(const double* R, double* out) {
	double qx, qy, qz, qw;
	double t = R[0] + R[4] + R[8];
	if (t > 0) {
		t = sqrt(t + 1.0);
		qw = 0.5 * t;
		t = 0.5 / t;
		qx = (R[7] - R[5]) * t, qy = (R[2] - R[6]) * t, qz = (R[3] - R[1]) * t;
	} else if (R[0] >= R[4] && R[0] >= R[8]) {	 // i = 0
		t = sqrt(R[0] - R[4] - R[8] + 1.0);
		qx = 0.5 * t;
		t = 0.5 / t;
		qw = (R[7] - R[5]) * t, qy = (R[3] + R[1]) * t, qz = (R[6] + R[2]) * t;
	} else if (R[4] > R[0] && R[4] >= R[8]) {  // i = 1
		t = sqrt(R[4] - R[8] - R[0] + 1.0);
		qy = 0.5 * t;
		t = 0.5 / t;
		qw = (R[2] - R[6]) * t, qz = (R[7] + R[5]) * t, qx = (R[1] + R[3]) * t;
	} else {  // i = 2
		t = sqrt(R[8] - R[0] - R[4] + 1.0);
		qz = 0.5 * t;
		t = 0.5 / t;
		qw = (R[3] - R[1]) * t, qx = (R[2] + R[6]) * t, qy = (R[5] + R[7]) * t;
	}
	double nrm = sqrt(qx * qx + qy * qy + qz * qz);
	if (nrm != 0) {
		#ifdef SAI2B_OTG_LIBM_TRIG  // A/B builds only: the library's own functions, as before round 2
		const double angle = 2 * atan2(nrm, fabs(qw));
#else
		const double angle = 2 * sai2b_det_atan2_pos(nrm, fabs(qw));
#endif
		if (qw < 0) nrm = -nrm;
		out[0] = angle * (qx / nrm), out[1] = angle * (qy / nrm), out[2] = angle * (qz / nrm);
	} else {
		out[0] = out[1] = out[2] = 0;
	}
}
// OTG_6dof_cartesian::getNextOrientation's local rotation (OTG_6dof_cartesian.cpp:226-237)
SAI2B_HD void vec_to_rot(double x, double y, double z, double* R) {
	const double nrm = sqrt(x * x + y * y + z * z);
	if (nrm < 1e-3) {
		R[0] = R[4] = R[8] = 1;
		R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0;
		return;
	}
	const double ax = x / nrm, ay = y / nrm, az = z / nrm;
	double s, c;
#ifdef SAI2B_OTG_LIBM_TRIG
	s = sin(nrm), c = cos(nrm);
#else
	sai2b_det_sincos(nrm, &s, &c);
#endif
	const double sx = s * ax, sy = s * ay, sz = s * az;
	const double cx = (1 - c) * ax, cy = (1 - c) * ay, cz = (1 - c) * az;
	double tmp = cx * ay;
	R[1] = tmp - sz, R[3] = tmp + sz;
	tmp = cx * az;
	R[2] = tmp + sy, R[6] = tmp - sy;
	tmp = cy * az;
	R[5] = tmp - sx, R[7] = tmp + sx;
	R[0] = cx * ax + c, R[4] = cy * ay + c, R[8] = cz * az + c;
}

// ---- OTG_6dof_cartesian (OTG_6dof_cartesian.cpp); DoF 0-2 position, 3-5 rotation vector in
// the reference frame; target_set bit 0 = position target set, bit 1 = orientation goal set ----
template <class G> SAI2B_HD void cart_next_orientation(const G& g, double* rot) {
	double local[9];
	vec_to_rot(g.np[3], g.np[4], g.np[5], local);
	mat3_mul(g.ref, local, rot);
}
SAI2B_HD bool approx9(const double* a, const double* b, int n, double prec) {
	double dd = 0, na = 0, nb = 0;
#pragma unroll
	for (int i = 0; i < 9; i++)
		if (i < n) {
			dd += (a[i] - b[i]) * (a[i] - b[i]);
			na += a[i] * a[i];
			nb += b[i] * b[i];
		}
	return dd <= prec * prec * (na < nb ? na : nb);
}
template <class G> SAI2B_HD void cart_set_goal_position(G& g, const double* gp, const double* gv) {
	const double p7[MAXD] = {gp[0], gp[1], gp[2], 0, 0, 0, 0}, v7[MAXD] = {gv[0], gv[1], gv[2], 0, 0, 0, 0};
	if ((g.target_set & 1) && approx_range(p7, g.in.tp, 0, 3, 1e-3) && approx_range(v7, g.in.tv, 0, 3, 1e-3)) return;
	g.goal_reached = 0;
	g.target_set |= 1;
#pragma unroll
	for (int i = 0; i < 3; i++) g.in.tp[i] = gp[i], g.in.tv[i] = gv[i];
}
template <class G> SAI2B_HD void cart_set_goal_orientation(G& g, const double* gR, const double* gw) {
	if ((g.target_set & 2) && approx9(g.goal_R, gR, 9, 1e-3) && approx9(g.goal_w, gw, 3, 1e-3)) return;
	g.goal_reached = 0;
	g.target_set |= 2;
	double new_ref[9], R_new_to_prev[9], tmp[3], ref_to_goal[9];
	cart_next_orientation(g, new_ref);
	mat3_tmul(new_ref, g.ref, R_new_to_prev);
#pragma unroll
	for (int i = 0; i < 9; i++) g.ref[i] = new_ref[i], g.goal_R[i] = gR[i];
#pragma unroll
	for (int i = 0; i < 3; i++) g.goal_w[i] = gw[i];
	g.np[3] = g.np[4] = g.np[5] = 0;
	mat3_vec(R_new_to_prev, g.nv[3], g.nv[4], g.nv[5], tmp);
	g.nv[3] = tmp[0], g.nv[4] = tmp[1], g.nv[5] = tmp[2];
	mat3_vec(R_new_to_prev, g.na[3], g.na[4], g.na[5], tmp);
	g.na[3] = tmp[0], g.na[4] = tmp[1], g.na[5] = tmp[2];
	pass_to_input(g, 6);
	mat3_tmul(g.ref, g.goal_R, ref_to_goal);
	rot_to_vec(ref_to_goal, tmp);
	g.in.tp[3] = tmp[0], g.in.tp[4] = tmp[1], g.in.tp[5] = tmp[2];
	mat3_tvec(g.ref, g.goal_w, tmp);
	g.in.tv[3] = tmp[0], g.in.tv[4] = tmp[1], g.in.tv[5] = tmp[2];
}
template <class G> SAI2B_HD void cart_reinitialize(G& g, const double* pos, const double* rot) {
	const double zeros[3] = {0, 0, 0};
	cart_set_goal_position(g, pos, zeros);
	cart_set_goal_orientation(g, rot, zeros);
#pragma unroll
	for (int i = 0; i < 6; i++) {
		g.in.cp[i] = g.in.tp[i], g.in.cv[i] = 0, g.in.ca[i] = 0;
		g.np[i] = g.in.tp[i], g.nv[i] = 0, g.na[i] = 0;
	}
}
// reInitializeLinear / reInitializeAngular (OTG_6dof_cartesian.cpp:60-83)
template <class G> SAI2B_HD void cart_reinitialize_linear(G& g, const double* pos) {
	const double zeros[3] = {0, 0, 0};
	cart_set_goal_position(g, pos, zeros);
#pragma unroll
	for (int i = 0; i < 3; i++) {
		g.in.cp[i] = g.in.tp[i], g.in.cv[i] = 0, g.in.ca[i] = 0;
		g.np[i] = g.in.tp[i], g.nv[i] = 0, g.na[i] = 0;
	}
}
template <class G> SAI2B_HD void cart_reinitialize_angular(G& g, const double* rot) {
	const double zeros[3] = {0, 0, 0};
	cart_set_goal_orientation(g, rot, zeros);
#pragma unroll
	for (int i = 3; i < 6; i++) {
		g.in.cp[i] = g.in.tp[i], g.in.cv[i] = 0, g.in.ca[i] = 0;
		g.np[i] = g.in.tp[i], g.nv[i] = 0, g.na[i] = 0;
	}
}
// OTG_6dof_cartesian::update after _otg->update() returned g.result (OTG_6dof_cartesian.cpp:194-223)
template <class G> SAI2B_HD void cart_finish(G& g, const Prev& pv) {
	if (g.result == FINISHED) {
		if (velocity_norm(g, 6) < 1e-3) {
			g.goal_reached = 1;
		} else {
			const double zeros[3] = {0, 0, 0};
			const double tp[3] = {g.in.tp[0], g.in.tp[1], g.in.tp[2]};
			double gR[9];
#pragma unroll
			for (int i = 0; i < 9; i++) gR[i] = g.goal_R[i];
			cart_set_goal_position(g, tp, zeros);
			cart_set_goal_orientation(g, gR, zeros);
		}
	} else if (g.result == WORKING) {
		pass_to_input(g, 6);
	} else {
		on_error(g, 6, pv);
	}
}
// OTG_6dof_cartesian::update (OTG_6dof_cartesian.cpp:187-224)
template <class G> SAI2B_HD void cart_update(G& g, double dt, const double (&vmax)[MAXD], const double (&amax)[MAXD], double epoch) {
	if (g.goal_reached) return;
	Prev pv;
	save_prev(g, pv);
	g.result = ruckig_update(g, 6, dt, vmax, amax, epoch);
	cart_finish(g, pv);
}

}  // namespace otg
}  // namespace sai2b
