// sai2b_otg3_core.hpp — the JERK-LIMITED internal trajectory generator (third-order position interface), device + host.
//
// What the reference does when a task's internal OTG is switched to jerk limitation
// (JointTask::enableInternalOtgJerkLimited, JointTask.h:295-310, JointTask.cpp:383-406;
// MotionForceTask::enableInternalOtgJerkLimited, MotionForceTask.h:416, MotionForceTask.cpp:525-538; called by
// examples/02-joint_control_internal_otg.cpp:175 and examples/03-cartesian_motion_control.cpp:179): the wrappers set a
// finite max_jerk and ruckig 0.10.1 plans with its third-order position interface — brake pre-trajectory
// (ruckig/src/ruckig/brake.cpp:14-77), extremal profiles (position-third-step1.cpp), blocked intervals (block.hpp),
// synchronisation (calculator_target.hpp), time synchronisation (position-third-step2.cpp), seven-phase sampling
// (trajectory.hpp:65-142), with the root finders of roots.hpp. This header restates that planner for one generator
// DoF at a time plus the sequential synchronisation over the DoFs; the wrappers around it (goal handling, Ruckig::update,
// the Finished / error bookkeeping) are the order-independent ones of sai2b_otg_core.hpp, instantiated for this
// trajectory type.
//
// The closed forms are the reference's (same operands, same order of operations) because the planner ACCEPTS or
// REJECTS candidate profiles by comparing re-integrated end states against 1e-8 / 1e-10 / 1e-12 thresholds
// (profile.hpp:175-272): another but algebraically equal expression takes another branch on marginal inputs. The file is
// compiled without floating-point contraction. Host build (tests/cpp/otg_core_test.cpp): bit-for-bit equal to the
// reference's own ruckig (oracle/_ref) on ruckig's known answers and on random inputs. Device build: cbrt / acos / cos /
// sin / atan of the root finders are the device library's, which differ from glibc's in the last bit now and then, so
// against the oracle the device planner is held to a tolerance, not to bits (tests/test_gpu_otg3.py).
//
// Execution model: the planner runs only when a goal (or a limit) changes, for the robots otg_kernel put on its work
// list; one lane plans one robot, DoF after DoF, out of per-lane scratch memory (sai2b_otg.hip: plan_lane3). Sampling a
// stored trajectory — the every-tick part — reads one phase (start state + jerk) per DoF.
#pragma once
#include "sai2b_otg_core.hpp"

// The planner's big functions are real calls on the device (not inlined): each is entered from several places (both
// directions, both planners), and inlining every copy made the translation unit take six minutes to compile for nothing —
// the planner runs out of scratch memory either way.
#ifdef __HIPCC__
#define SAI2B_HDN __host__ __device__ __attribute__((noinline))
#else
#define SAI2B_HDN inline
#endif

namespace sai2b {
namespace otg3 {

using otg::EPS;
using otg::Input;
using otg::MAXD;

// Profile::ReachedLimits / ControlSigns / Direction (profile.hpp:55-57)
enum { L_ACC0_ACC1_VEL = 0, L_VEL, L_ACC0, L_ACC1, L_ACC0_ACC1, L_ACC0_VEL, L_ACC1_VEL, L_NONE };
enum { UDDU = 0, UDUD = 1 };
enum { UP = 0, DOWN = 1 };

// std::min / std::max as the reference uses them (they return the FIRST argument when the comparison is false, also
// for NaNs — fmin / fmax do not)
SAI2B_HD double smin(double a, double b) { return (b < a) ? b : a; }
SAI2B_HD double smax(double a, double b) { return (a < b) ? b : a; }
SAI2B_HD double pow2(double v) { return v * v; }

// BrakeProfile (brake.hpp:16-77): at most two phases
struct Brake {
	double duration;
	double t[2], j[2], a[2], v[2], p[2];
};
// Profile (profile.hpp:33-57), third-order members
struct Prof {
	double t[7], t_sum[7], j[7];
	double a[8], v[8], p[8];
	Brake brake;
	double pf, vf, af;
	int limits, direction, control_signs;
};

SAI2B_HD void integrate(double t, double p0, double v0, double a0, double j, double& p, double& v, double& a) {  // utils.hpp:43-49
	p = p0 + t * (v0 + t * (a0 / 2 + t * j / 6));
	v = v0 + t * (a0 + t * j / 2);
	a = a0 + t * j;
}

// Profile::set_boundary(const Profile&) (profile.hpp:285-294)
SAI2B_HD void set_boundary(Prof& pr, const Prof& from) {
	pr.a[0] = from.a[0], pr.v[0] = from.v[0], pr.p[0] = from.p[0];
	pr.af = from.af, pr.vf = from.vf, pr.pf = from.pf;
	pr.brake = from.brake;
}

// Profile::check<control_signs, limits, set_limits> (profile.hpp:175-272)
template <int CS, int LIM, bool SET_LIMITS = false>
SAI2B_HD bool check(Prof& pr, double jf, double vMax, double vMin, double aMax, double aMin) {
	if (pr.t[0] < 0) return false;
	pr.t_sum[0] = pr.t[0];
#pragma unroll
	for (int i = 0; i < 6; ++i) {
		if (pr.t[i + 1] < 0) return false;
		pr.t_sum[i + 1] = pr.t_sum[i] + pr.t[i + 1];
	}
	if (LIM == L_ACC0_ACC1_VEL || LIM == L_ACC0_VEL || LIM == L_ACC1_VEL || LIM == L_VEL) {
		if (pr.t[3] < EPS) return false;
	}
	if (LIM == L_ACC0 || LIM == L_ACC0_ACC1) {
		if (pr.t[1] < EPS) return false;
	}
	if (LIM == L_ACC1 || LIM == L_ACC0_ACC1) {
		if (pr.t[5] < EPS) return false;
	}
	if (pr.t_sum[6] > 1e12) return false;

	pr.j[0] = (pr.t[0] > 0 ? jf : 0), pr.j[1] = 0, pr.j[2] = (pr.t[2] > 0 ? -jf : 0), pr.j[3] = 0, pr.j[5] = 0;
	if (CS == UDDU) {
		pr.j[4] = (pr.t[4] > 0 ? -jf : 0), pr.j[6] = (pr.t[6] > 0 ? jf : 0);
	} else {
		pr.j[4] = (pr.t[4] > 0 ? jf : 0), pr.j[6] = (pr.t[6] > 0 ? -jf : 0);
	}

	pr.direction = (vMax > 0) ? UP : DOWN;
	const double vUppLim = (pr.direction == UP ? vMax : vMin) + 1e-12;
	const double vLowLim = (pr.direction == UP ? vMin : vMax) - 1e-12;

#pragma unroll
	for (int i = 0; i < 7; ++i) {
		pr.a[i + 1] = pr.a[i] + pr.t[i] * pr.j[i];
		pr.v[i + 1] = pr.v[i] + pr.t[i] * (pr.a[i] + pr.t[i] * pr.j[i] / 2);
		pr.p[i + 1] = pr.p[i] + pr.t[i] * (pr.v[i] + pr.t[i] * (pr.a[i] / 2 + pr.t[i] * pr.j[i] / 6));

		if (LIM == L_ACC0_ACC1_VEL || LIM == L_ACC0_ACC1 || LIM == L_ACC0_VEL || LIM == L_ACC1_VEL || LIM == L_VEL) {
			if (i == 2) pr.a[3] = 0.0;
		}
		if (SET_LIMITS) {
			if (LIM == L_ACC1) {
				if (i == 2) pr.a[3] = aMin;
			}
			if (LIM == L_ACC0_ACC1) {
				if (i == 0) pr.a[1] = aMax;
				if (i == 4) pr.a[5] = aMin;
			}
		}
		if (i > 1 && pr.a[i + 1] * pr.a[i] < -EPS) {
			const double v_a_zero = pr.v[i] - (pr.a[i] * pr.a[i]) / (2 * pr.j[i]);
			if (v_a_zero > vUppLim || v_a_zero < vLowLim) return false;
		}
	}

	pr.control_signs = CS;
	pr.limits = LIM;

	const double aUppLim = (pr.direction == UP ? aMax : aMin) + 1e-12;
	const double aLowLim = (pr.direction == UP ? aMin : aMax) - 1e-12;

	return fabs(pr.p[7] - pr.pf) < 1e-8 && fabs(pr.v[7] - pr.vf) < 1e-8 && fabs(pr.a[7] - pr.af) < 1e-10 && pr.a[1] >= aLowLim &&
		   pr.a[3] >= aLowLim && pr.a[5] >= aLowLim && pr.a[1] <= aUppLim && pr.a[3] <= aUppLim && pr.a[5] <= aUppLim &&
		   pr.v[3] <= vUppLim && pr.v[4] <= vUppLim && pr.v[5] <= vUppLim && pr.v[6] <= vUppLim && pr.v[3] >= vLowLim &&
		   pr.v[4] >= vLowLim && pr.v[5] >= vLowLim && pr.v[6] >= vLowLim;
}
// check_with_timing (profile.hpp:274-283): the duration needs no check (every profile has a "tf - ..." equation)
template <int CS, int LIM> SAI2B_HD bool check_t(Prof& pr, double, double jf, double vMax, double vMin, double aMax, double aMin) {
	return check<CS, LIM>(pr, jf, vMax, vMin, aMax, aMin);
}
template <int CS, int LIM>
SAI2B_HD bool check_tj(Prof& pr, double tf, double jf, double vMax, double vMin, double aMax, double aMin, double jMax) {
	return (fabs(jf) < fabs(jMax) + 1e-12) && check_t<CS, LIM>(pr, tf, jf, vMax, vMin, aMax, aMin);
}

// ---------------------------------------------------------------------------------------------------------------------
// roots.hpp: real roots of cubics / monic quartics, polynomial helpers, the safe Newton iteration
// ---------------------------------------------------------------------------------------------------------------------
// PositiveSet<double, N> (roots.hpp:20-56): keeps values >= 0, hands them out in ascending order
struct Roots {
	double v[4];
	int n;
};
SAI2B_HD void rinsert(Roots& r, double value) {
	if (value >= 0) r.v[r.n++] = value;
}
SAI2B_HD void rsort(Roots& r) {
	for (int i = 1; i < r.n; i++) {
		const double x = r.v[i];
		int k = i - 1;
		while (k >= 0 && r.v[k] > x) {
			r.v[k + 1] = r.v[k];
			k--;
		}
		r.v[k + 1] = x;
	}
}

// solveCub (roots.hpp:60-149): a x^3 + b x^2 + c x + d = 0
SAI2B_HDN Roots solve_cub(double a, double b, double c, double d) {
	Roots roots;
	roots.n = 0;
	if (fabs(d) < EPS) {
		rinsert(roots, 0.0);
		d = c;
		c = b;
		b = a;
		a = 0.0;
	}
	if (fabs(a) < EPS) {
		if (fabs(b) < EPS) {
			if (fabs(c) > EPS) rinsert(roots, -d / c);
		} else {
			const double discriminant = c * c - 4 * b * d;
			if (discriminant >= 0) {
				const double inv2b = 1.0 / (2 * b);
				const double y = sqrt(discriminant);
				rinsert(roots, (-c + y) * inv2b);
				rinsert(roots, (-c - y) * inv2b);
			}
		}
	} else {
		const double inva = 1.0 / a;
		const double invaa = inva * inva;
		const double bb = b * b;
		const double bover3a = b * inva / 3;
		const double p = (a * c - bb / 3) * invaa;
		const double halfq = (2 * bb * b - 9 * a * b * c + 27 * a * a * d) / 54 * invaa * inva;
		const double yy = p * p * p / 27 + halfq * halfq;
		const double cos120 = -0.50;
		const double sin120 = 0.866025403784438646764;
		if (yy > EPS) {
			const double y = sqrt(yy);
			const double uuu = -halfq + y;
			const double vvv = -halfq - y;
			const double www = fabs(uuu) > fabs(vvv) ? uuu : vvv;
			const double w = cbrt(www);
			rinsert(roots, w - p / (3 * w) - bover3a);
		} else if (yy < -EPS) {
			const double x = -halfq;
			const double y = sqrt(-yy);
			double theta;
			double r;
			if (fabs(x) > EPS) {
				theta = (x > 0.0) ? atan(y / x) : (atan(y / x) + M_PI);
				r = sqrt(x * x - yy);
			} else {
				theta = M_PI / 2;
				r = y;
			}
			theta /= 3;
			r = 2 * cbrt(r);
			const double ux = cos(theta) * r;
			const double uyi = sin(theta) * r;
			rinsert(roots, ux - bover3a);
			rinsert(roots, ux * cos120 - uyi * sin120 - bover3a);
			rinsert(roots, ux * cos120 + uyi * sin120 - bover3a);
		} else {
			const double www = -halfq;
			const double w = 2 * cbrt(www);
			rinsert(roots, w - bover3a);
			rinsert(roots, w * cos120 - bover3a);
		}
	}
	rsort(roots);
	return roots;
}

// solveResolvent (roots.hpp:154-195)
SAI2B_HD int solve_resolvent(double (&x)[3], double a, double b, double c) {
	const double cos120 = -0.50;
	const double sin120 = 0.866025403784438646764;
	a /= 3;
	const double a2 = a * a;
	double q = a2 - b / 3;
	const double r = (a * (2 * a2 - b) + c) / 2;
	const double r2 = r * r;
	const double q3 = q * q * q;
	if (r2 < q3) {
		const double qsqrt = sqrt(q);
		const double t = smin(smax(r / (q * qsqrt), -1.0), 1.0);
		q = -2 * qsqrt;
		const double theta = acos(t) / 3;
		const double ux = cos(theta) * q;
		const double uyi = sin(theta) * q;
		x[0] = ux - a;
		x[1] = ux * cos120 - uyi * sin120 - a;
		x[2] = ux * cos120 + uyi * sin120 - a;
		return 3;
	} else {
		double A = -cbrt(fabs(r) + sqrt(r2 - q3));
		if (r < 0.0) A = -A;
		const double B = (0.0 == A ? 0.0 : q / A);
		x[0] = (A + B) - a;
		x[1] = -(A + B) / 2 - a;
		x[2] = sqrt(3.0) * (A - B) / 2;
		if (fabs(x[2]) < EPS) {
			x[2] = x[1];
			return 2;
		}
		return 1;
	}
}

// solveQuartMonic (roots.hpp:198-283): x^4 + a x^3 + b x^2 + c x + d = 0
SAI2B_HDN Roots solve_quart_monic(double a, double b, double c, double d) {
	Roots roots;
	roots.n = 0;
	if (fabs(d) < EPS) {
		if (fabs(c) < EPS) {
			rinsert(roots, 0.0);
			const double D = a * a - 4 * b;
			if (fabs(D) < EPS) {
				rinsert(roots, -a / 2);
			} else if (D > 0.0) {
				const double sqrtD = sqrt(D);
				rinsert(roots, (-a - sqrtD) / 2);
				rinsert(roots, (-a + sqrtD) / 2);
			}
			rsort(roots);
			return roots;
		}
		if (fabs(a) < EPS && fabs(b) < EPS) {
			rinsert(roots, 0.0);
			rinsert(roots, -cbrt(c));
			rsort(roots);
			return roots;
		}
	}
	const double a3 = -b;
	const double b3 = a * c - 4 * d;
	const double c3 = -a * a * d - c * c + 4 * b * d;
	double x3[3];
	const int number_zeroes = solve_resolvent(x3, a3, b3, c3);
	double y = x3[0];
	if (number_zeroes != 1) {
		if (fabs(x3[1]) > fabs(y)) y = x3[1];
		if (fabs(x3[2]) > fabs(y)) y = x3[2];
	}
	double q1, q2, p1, p2;
	double D = y * y - 4 * d;
	if (fabs(D) < EPS) {
		q1 = q2 = y / 2;
		D = a * a - 4 * (b - y);
		if (fabs(D) < EPS) {
			p1 = p2 = a / 2;
		} else {
			const double sqrtD = sqrt(D);
			p1 = (a + sqrtD) / 2;
			p2 = (a - sqrtD) / 2;
		}
	} else {
		const double sqrtD = sqrt(D);
		q1 = (y + sqrtD) / 2;
		q2 = (y - sqrtD) / 2;
		p1 = (a * q1 - c) / (q1 - q2);
		p2 = (c - a * q2) / (q1 - q2);
	}
	const double eps16 = 16 * EPS;
	D = p1 * p1 - 4 * q1;
	if (fabs(D) < eps16) {
		rinsert(roots, -p1 / 2);
	} else if (D > 0.0) {
		const double sqrtD = sqrt(D);
		rinsert(roots, (-p1 - sqrtD) / 2);
		rinsert(roots, (-p1 + sqrtD) / 2);
	}
	D = p2 * p2 - 4 * q2;
	if (fabs(D) < eps16) {
		rinsert(roots, -p2 / 2);
	} else if (D > 0.0) {
		const double sqrtD = sqrt(D);
		rinsert(roots, (-p2 - sqrtD) / 2);
		rinsert(roots, (-p2 + sqrtD) / 2);
	}
	rsort(roots);
	return roots;
}
SAI2B_HD Roots solve_quart_monic(const double (&p)[4]) { return solve_quart_monic(p[0], p[1], p[2], p[3]); }

// polyEval<N>, polyDeri<N>, polyMonicDeri<N>, shrinkInterval<N> (roots.hpp:292-395); N = number of coefficients
template <int N> SAI2B_HD double poly_eval(const double (&p)[N], double x) {
	double ret = 0.0;
	if (fabs(x) < EPS) {
		ret = p[N - 1];
	} else if (x == 1.0) {
		for (int i = N - 1; i >= 0; i--) ret += p[i];
	} else {
		double xn = 1.0;
		for (int i = N - 1; i >= 0; i--) {
			ret += p[i] * xn;
			xn *= x;
		}
	}
	return ret;
}
template <int N> SAI2B_HD void poly_deri(const double (&c)[N], double (&d)[N - 1]) {
	for (int i = 0; i < N - 1; ++i) d[i] = (double)(N - 1 - i) * c[i];
}
template <int N> SAI2B_HD void poly_monic_deri(const double (&c)[N], double (&d)[N - 1]) {
	d[0] = 1.0;
	for (int i = 1; i < N - 1; ++i) d[i] = (double)(N - 1 - i) * c[i] / (double)(N - 1);
}
template <int N> SAI2B_HD double shrink_interval(const double (&p)[N], double l, double h) {
	const double fl = poly_eval(p, l);
	const double fh = poly_eval(p, h);
	if (fl == 0.0) return l;
	if (fh == 0.0) return h;
	if (fl > 0.0) {
		const double tmp = l;
		l = h;
		h = tmp;
	}
	double rts = (l + h) / 2;
	double dxold = fabs(h - l);
	double dx = dxold;
	double deriv[N - 1];
	poly_deri(p, deriv);
	double f = poly_eval(p, rts);
	double df = poly_eval(deriv, rts);
	double temp;
	for (int j = 0; j < 128; j++) {
		if ((((rts - h) * df - f) * ((rts - l) * df - f) > 0.0) || (fabs(2 * f) > fabs(dxold * df))) {
			dxold = dx;
			dx = (h - l) / 2;
			rts = l + dx;
			if (l == rts) break;
		} else {
			dxold = dx;
			dx = f / df;
			temp = rts;
			rts -= dx;
			if (temp == rts) break;
		}
		if (fabs(dx) < 1e-14) break;
		f = poly_eval(p, rts);
		df = poly_eval(deriv, rts);
		if (f < 0.0) {
			l = rts;
		} else {
			h = rts;
		}
	}
	return rts;
}

// ---------------------------------------------------------------------------------------------------------------------
// brake.cpp:14-77, brake.hpp:42-62: pre-trajectory that brings a state outside the limits back inside
// ---------------------------------------------------------------------------------------------------------------------
SAI2B_HD double v_at_t(double v0, double a0, double j, double t) { return v0 + t * (a0 + j * t / 2); }
SAI2B_HD double v_at_a_zero(double v0, double a0, double j) { return v0 + (a0 * a0) / (2 * j); }

SAI2B_HD void velocity_brake(Brake& b, double v0, double a0, double vMax, double vMin, double, double aMin, double jMax) {
	const double eps = 2.2e-14;
	b.j[0] = -jMax;
	const double t_to_a_min = (a0 - aMin) / jMax;
	const double t_to_v_max = a0 / jMax + sqrt(a0 * a0 + 2 * jMax * (v0 - vMax)) / fabs(jMax);
	const double t_to_v_min = a0 / jMax + sqrt(a0 * a0 / 2 + jMax * (v0 - vMin)) / fabs(jMax);
	const double t_min_to_v_max = smin(t_to_v_max, t_to_v_min);
	if (t_to_a_min < t_min_to_v_max) {
		const double v_at_a_min = v_at_t(v0, a0, -jMax, t_to_a_min);
		const double t_to_v_max_with_constant = -(v_at_a_min - vMax) / aMin;
		const double t_to_v_min_with_constant = aMin / (2 * jMax) - (v_at_a_min - vMin) / aMin;
		b.t[0] = smax(t_to_a_min - eps, 0.0);
		b.t[1] = smax(smin(t_to_v_max_with_constant, t_to_v_min_with_constant), 0.0);
	} else {
		b.t[0] = smax(t_min_to_v_max - eps, 0.0);
	}
}
SAI2B_HD void acceleration_brake(Brake& b, double v0, double a0, double vMax, double vMin, double aMax, double aMin, double jMax) {
	const double eps = 2.2e-14;
	b.j[0] = -jMax;
	const double t_to_a_max = (a0 - aMax) / jMax;
	const double t_to_a_zero = a0 / jMax;
	const double v_at_a_max = v_at_t(v0, a0, -jMax, t_to_a_max);
	const double v_at_a_zero_ = v_at_t(v0, a0, -jMax, t_to_a_zero);
	if ((v_at_a_zero_ > vMax && jMax > 0) || (v_at_a_zero_ < vMax && jMax < 0)) {
		velocity_brake(b, v0, a0, vMax, vMin, aMax, aMin, jMax);
	} else if ((v_at_a_max < vMin && jMax > 0) || (v_at_a_max > vMin && jMax < 0)) {
		const double t_to_v_min = -(v_at_a_max - vMin) / aMax;
		const double t_to_v_max = -aMax / (2 * jMax) - (v_at_a_max - vMax) / aMax;
		b.t[0] = t_to_a_max + eps;
		b.t[1] = smax(smin(t_to_v_min, t_to_v_max - eps), 0.0);
	} else {
		b.t[0] = t_to_a_max + eps;
	}
}
// get_position_brake_trajectory (brake.cpp:56-77)
SAI2B_HD void position_brake(Brake& b, double v0, double a0, double vMax, double vMin, double aMax, double aMin, double jMax) {
	b.t[0] = 0.0, b.t[1] = 0.0, b.j[0] = 0.0, b.j[1] = 0.0;
	if (jMax == 0.0 || aMax == 0.0 || aMin == 0.0) return;
	if (a0 > aMax) {
		acceleration_brake(b, v0, a0, vMax, vMin, aMax, aMin, jMax);
	} else if (a0 < aMin) {
		acceleration_brake(b, v0, a0, vMin, vMax, aMin, aMax, -jMax);
	} else if ((v0 > vMax && v_at_a_zero(v0, a0, -jMax) > vMin) || (a0 > 0 && v_at_a_zero(v0, a0, jMax) > vMax)) {
		velocity_brake(b, v0, a0, vMax, vMin, aMax, aMin, jMax);
	} else if ((v0 < vMin && v_at_a_zero(v0, a0, jMax) < vMax) || (a0 < 0 && v_at_a_zero(v0, a0, -jMax) < vMin)) {
		velocity_brake(b, v0, a0, vMin, vMax, aMin, aMax, -jMax);
	}
}
// BrakeProfile::finalize (brake.hpp:42-62)
SAI2B_HD void brake_finalize(Brake& b, double& ps, double& vs, double& as) {
	b.a[0] = b.a[1] = b.v[0] = b.v[1] = b.p[0] = b.p[1] = 0.0;
	if (b.t[0] <= 0.0 && b.t[1] <= 0.0) {
		b.duration = 0.0;
		return;
	}
	b.duration = b.t[0];
	b.p[0] = ps, b.v[0] = vs, b.a[0] = as;
	integrate(b.t[0], ps, vs, as, b.j[0], ps, vs, as);
	if (b.t[1] > 0.0) {
		b.duration += b.t[1];
		b.p[1] = ps, b.v[1] = vs, b.a[1] = as;
		integrate(b.t[1], ps, vs, as, b.j[1], ps, vs, as);
	}
}

// ---------------------------------------------------------------------------------------------------------------------
// Block (block.hpp): the fastest profile and up to two blocked duration intervals
// ---------------------------------------------------------------------------------------------------------------------
struct Block {
	Prof pmin, aprof, bprof;
	double tmin, aleft, aright, bleft, bright;
	bool a, b;
};
SAI2B_HD double total(const Prof& p) { return p.t_sum[6] + p.brake.duration + 0.0; }  // (+ accel.duration, never set)
SAI2B_HD void set_min(Block& bl, const Prof& p) {
	bl.pmin = p;
	bl.tmin = total(p);
	bl.a = bl.b = false;
}
SAI2B_HD void interval(double& left, double& right, Prof& prof, const Prof& pl, const Prof& pr) {
	const double ld = total(pl), rd = total(pr);
	if (ld < rd) {
		left = ld, right = rd, prof = pr;
	} else {
		left = rd, right = ld, prof = pl;
	}
}
SAI2B_HD bool is_blocked(const Block& b, double t) {
	return (t < b.tmin) || (b.a && b.aleft < t && t < b.aright) || (b.b && b.bleft < t && t < b.bright);
}
// Block::calculate_block<6, true> (block.hpp:61-134)
SAI2B_HDN bool calculate_block(Block& bl, Prof* v, int count) {
	if (count == 1) {
		set_min(bl, v[0]);
		return true;
	} else if (count == 2) {
		if (fabs(v[0].t_sum[6] - v[1].t_sum[6]) < 8 * EPS) {
			set_min(bl, v[0]);
			return true;
		}
		const int imin = (v[0].t_sum[6] < v[1].t_sum[6]) ? 0 : 1;
		set_min(bl, v[imin]);
		bl.a = true;
		interval(bl.aleft, bl.aright, bl.aprof, v[imin], v[(imin + 1) % 2]);
		return true;
	} else if (count == 4) {
		int drop;
		if (fabs(v[0].t_sum[6] - v[1].t_sum[6]) < 32 * EPS && v[0].direction != v[1].direction) {
			drop = 1;
		} else if (fabs(v[2].t_sum[6] - v[3].t_sum[6]) < 256 * EPS && v[2].direction != v[3].direction) {
			drop = 3;
		} else if (fabs(v[0].t_sum[6] - v[3].t_sum[6]) < 256 * EPS && v[0].direction != v[3].direction) {
			drop = 3;
		} else {
			return false;
		}
		for (int i = drop; i < count - 1; ++i) v[i] = v[i + 1];
		count -= 1;
	} else if (count % 2 == 0) {
		return false;
	}
	int imin = 0;
	for (int i = 1; i < count; i++)
		if (v[i].t_sum[6] < v[imin].t_sum[6]) imin = i;
	set_min(bl, v[imin]);
	if (count == 3) {
		bl.a = true;
		interval(bl.aleft, bl.aright, bl.aprof, v[(imin + 1) % 3], v[(imin + 2) % 3]);
		return true;
	} else if (count == 5) {
		const int e1 = (imin + 1) % 5, e2 = (imin + 2) % 5, e3 = (imin + 3) % 5, e4 = (imin + 4) % 5;
		bl.a = bl.b = true;
		if (v[e1].direction == v[e2].direction) {
			interval(bl.aleft, bl.aright, bl.aprof, v[e1], v[e2]);
			interval(bl.bleft, bl.bright, bl.bprof, v[e3], v[e4]);
		} else {
			interval(bl.aleft, bl.aright, bl.aprof, v[e1], v[e4]);
			interval(bl.bleft, bl.bright, bl.bprof, v[e2], v[e3]);
		}
		return true;
	}
	return false;
}

#include "sai2b_otg3_step1.inc"
#include "sai2b_otg3_step2.inc"
#include "sai2b_otg3_calc.inc"

// One jerk-limited generator: the state of one OTG_joints / OTG_6dof_cartesian object whose max_jerk is finite. Same
// members as otg::Gen (the wrappers of sai2b_otg_core.hpp are templates over the two), the third-order trajectory and
// the jerk limits the planner hook needs.
struct Gen {
	Input in, ci;
	double np[MAXD], nv[MAXD], na[MAXD];
	double time;
	Traj traj;
	int goal_reached, result, target_set, ci_init;
	int replanned;
	double ci_epoch;
	double ref[9], goal_R[9], goal_w[3];  // OTG_6dof_cartesian only
	double jmax[MAXD];
};
// the hooks of the wrappers (sai2b_otg_core.hpp: ruckig_sample / ruckig_update), found by argument type
SAI2B_HD void sample_dof(Gen& g, int d) { at_time(g.traj.prof[d], g.traj.duration, g.time, g.np[d], g.nv[d], g.na[d]); }
// (Ruckig::calculate writes into output.trajectory in place, also when it then fails: ruckig.hpp:150-175)
SAI2B_HD int plan(Gen& g, int n, const double (&vmax)[MAXD], const double (&amax)[MAXD]) {
	if (!validate(g.in, n, vmax, amax, g.jmax)) return otg::ERR_INVALID_INPUT;
	return calculate(g.in, n, vmax, amax, g.jmax, g.traj);
}

}  // namespace otg3
}  // namespace sai2b
