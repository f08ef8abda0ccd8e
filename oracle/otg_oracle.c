/*
 * otg_oracle.c — see otg_oracle.h. TEST INFRASTRUCTURE ONLY.
 *
 * Part 1 restates the acceleration-limited position interface of ruckig 0.10.1 as vendored by the
 * reference (file:line cited per function); part 2 restates the sai2 wrappers.
 * Documented definitions where the reference leaves behaviour open:
 *  - step 1 "solution 2" after an accepted "solution 1" reads t[3..6] of a fresh, never-initialised
 *    Profile (position-second-step1.cpp:42-66, position.hpp:107-108); they are taken as 0 here;
 *  - the candidate synchronisation times are ordered by a stable sort (ties keep DoF order);
 *    std::sort (calculator_target.hpp:169) is stable for the <= 16 elements of the common case;
 *  - OTG_joints::update calls setGoalPosition(_goal_position_eigen) with a member that is never
 *    assigned (OTG_joints.cpp:129, OTG_joints.h:172), which throws; here it does what the Cartesian
 *    wrapper does in the same place (OTG_6dof_cartesian.cpp:206): keep the target position, zero
 *    the target velocity.
 */
#include "otg_oracle.h"

#include <float.h>
#include <math.h>

#include "../include/sai2b_detmath.h"
#include <string.h>
#include <dlfcn.h>
#include <stdio.h>
#include <stdlib.h>

#define EPS DBL_EPSILON
enum { LIM_ACC0 = 2, LIM_NONE = 7 };
enum { DIR_UP = 0, DIR_DOWN = 1 };
enum { CS_UDDU = 0 };

static const double v_eps = 1e-12, a_eps = 1e-12, p_precision = 1e-8, v_precision = 1e-8,
					t_max_profile = 1e12;
static const double brake_eps = 2.2e-14;

/* ------------------------------------------------------------------ Profile */

/* Profile::check_for_second_order<UDDU, limits> (profile.hpp:307-350) */
static int check_second_order(otg_profile* pr, int limits, double aUp, double aDown, double vMax,
							  double vMin) {
	int i;
	if (pr->t[0] < 0) return 0;
	pr->t_sum[0] = pr->t[0];
	for (i = 0; i < 6; i++) {
		if (pr->t[i + 1] < 0) return 0;
		pr->t_sum[i + 1] = pr->t_sum[i] + pr->t[i + 1];
	}
	if (pr->t_sum[6] > t_max_profile) return 0;

	pr->a[0] = pr->t[0] > 0 ? aUp : 0;
	pr->a[1] = 0;
	pr->a[2] = pr->t[2] > 0 ? aDown : 0;
	pr->a[3] = 0;
	pr->a[4] = pr->t[4] > 0 ? aDown : 0;
	pr->a[5] = 0;
	pr->a[6] = pr->t[6] > 0 ? aUp : 0;
	pr->a[7] = pr->af;

	pr->direction = (vMax > 0) ? DIR_UP : DIR_DOWN;
	const double vUppLim = (pr->direction == DIR_UP ? vMax : vMin) + v_eps;
	const double vLowLim = (pr->direction == DIR_UP ? vMin : vMax) - v_eps;

	for (i = 0; i < 7; i++) {
		pr->v[i + 1] = pr->v[i] + pr->t[i] * pr->a[i];
		pr->p[i + 1] = pr->p[i] + pr->t[i] * (pr->v[i] + pr->t[i] * pr->a[i] / 2);
	}
	pr->control_signs = CS_UDDU;
	pr->limits = limits;

	return fabs(pr->p[7] - pr->pf) < p_precision && fabs(pr->v[7] - pr->vf) < v_precision &&
		   pr->v[2] <= vUppLim && pr->v[3] <= vUppLim && pr->v[4] <= vUppLim &&
		   pr->v[5] <= vUppLim && pr->v[6] <= vUppLim && pr->v[2] >= vLowLim &&
		   pr->v[3] >= vLowLim && pr->v[4] >= vLowLim && pr->v[5] >= vLowLim &&
		   pr->v[6] >= vLowLim;
}

/* check_for_second_order_with_timing, 7-argument form (profile.hpp:358-361) */
static int check_second_order_timing_limits(otg_profile* pr, int limits, double aUp, double aDown,
											double vMax, double vMin, double aMax, double aMin) {
	return (aMin - a_eps < aUp) && (aUp < aMax + a_eps) && (aMin - a_eps < aDown) &&
		   (aDown < aMax + a_eps) && check_second_order(pr, limits, aUp, aDown, vMax, vMin);
}

/* Profile::set_boundary(const Profile&) (profile.hpp:283-292) */
static void set_boundary_from(otg_profile* dst, const otg_profile* src) {
	dst->a[0] = src->a[0];
	dst->v[0] = src->v[0];
	dst->p[0] = src->p[0];
	dst->af = src->af;
	dst->vf = src->vf;
	dst->pf = src->pf;
	dst->brake = src->brake;
}

static double profile_total(const otg_profile* p) { return p->t_sum[6] + p->brake.duration + 0.0; }

/* ------------------------------------------------------------------ Block (block.hpp) */

typedef struct {
	int valid;
	double left, right;
	otg_profile profile;
} otg_interval;

typedef struct {
	otg_profile p_min;
	double t_min;
	otg_interval a, b;
} otg_block;

/* Block::Interval(profile_left, profile_right) (block.hpp:32-44) */
static void interval_from(otg_interval* iv, const otg_profile* pl, const otg_profile* pr) {
	const double ld = profile_total(pl), rd = profile_total(pr);
	iv->valid = 1;
	if (ld < rd) {
		iv->left = ld;
		iv->right = rd;
		iv->profile = *pr;
	} else {
		iv->left = rd;
		iv->right = ld;
		iv->profile = *pl;
	}
}

/* Block::set_min_profile (block.hpp:47-52) */
static void set_min_profile(otg_block* b, const otg_profile* p) {
	b->p_min = *p;
	b->t_min = profile_total(p);
	b->a.valid = 0;
	b->b.valid = 0;
}

static void remove_profile(otg_profile* v, int* count, int index) {
	int i;
	for (i = index; i < *count - 1; i++) v[i] = v[i + 1];
	*count -= 1;
}

/* Block::calculate_block<N, numerical_robust = true> (block.hpp:61-134) */
static int calculate_block(otg_block* block, otg_profile* v, int count) {
	int i;
	if (count == 1) {
		set_min_profile(block, &v[0]);
		return 1;
	} else if (count == 2) {
		if (fabs(v[0].t_sum[6] - v[1].t_sum[6]) < 8 * EPS) {
			set_min_profile(block, &v[0]);
			return 1;
		}
		const int idx_min = (v[0].t_sum[6] < v[1].t_sum[6]) ? 0 : 1;
		const int idx_else = (idx_min + 1) % 2;
		set_min_profile(block, &v[idx_min]);
		interval_from(&block->a, &v[idx_min], &v[idx_else]);
		return 1;
	} else if (count == 4) {
		if (fabs(v[0].t_sum[6] - v[1].t_sum[6]) < 32 * EPS && v[0].direction != v[1].direction) {
			remove_profile(v, &count, 1);
		} else if (fabs(v[2].t_sum[6] - v[3].t_sum[6]) < 256 * EPS &&
				   v[2].direction != v[3].direction) {
			remove_profile(v, &count, 3);
		} else if (fabs(v[0].t_sum[6] - v[3].t_sum[6]) < 256 * EPS &&
				   v[0].direction != v[3].direction) {
			remove_profile(v, &count, 3);
		} else {
			return 0;
		}
	} else if (count % 2 == 0) {
		return 0;
	}

	int idx_min = 0;
	for (i = 1; i < count; i++)
		if (v[i].t_sum[6] < v[idx_min].t_sum[6]) idx_min = i;
	set_min_profile(block, &v[idx_min]);

	if (count == 3) {
		interval_from(&block->a, &v[(idx_min + 1) % 3], &v[(idx_min + 2) % 3]);
		return 1;
	} else if (count == 5) {
		const int e1 = (idx_min + 1) % 5, e2 = (idx_min + 2) % 5, e3 = (idx_min + 3) % 5,
				  e4 = (idx_min + 4) % 5;
		if (v[e1].direction == v[e2].direction) {
			interval_from(&block->a, &v[e1], &v[e2]);
			interval_from(&block->b, &v[e3], &v[e4]);
		} else {
			interval_from(&block->a, &v[e1], &v[e4]);
			interval_from(&block->b, &v[e2], &v[e3]);
		}
		return 1;
	}
	return 0;
}

/* Block::is_blocked (block.hpp:136-138) */
static int is_blocked(const otg_block* b, double t) {
	return (t < b->t_min) || (b->a.valid && b->a.left < t && t < b->a.right) ||
		   (b->b.valid && b->b.left < t && t < b->b.right);
}

/* ------------------------------------------------------------------ Step 1
 * PositionSecondOrderStep1 (position-second-step1.cpp) */

typedef struct {
	double v0, vf, vMax, vMin, aMax, aMin, pd;
	otg_profile valid[8];
	int count; /* index of the slot being tried = number of accepted profiles */
} step1_t;

static void s1_add_profile(step1_t* s) {
	const otg_profile* prev = &s->valid[s->count];
	s->count++;
	set_boundary_from(&s->valid[s->count], prev);
}

/* time_acc0 (position-second-step1.cpp:11-24) */
static void s1_time_acc0(step1_t* s, double vMax, double vMin, double aMax, double aMin) {
	otg_profile* pr = &s->valid[s->count];
	const double v0 = s->v0, vf = s->vf, pd = s->pd;
	pr->t[0] = (-v0 + vMax) / aMax;
	pr->t[1] = (aMin * v0 * v0 - aMax * vf * vf) / (2 * aMax * aMin * vMax) +
			   vMax * (aMax - aMin) / (2 * aMax * aMin) + pd / vMax;
	pr->t[2] = (vf - vMax) / aMin;
	pr->t[3] = 0;
	pr->t[4] = 0;
	pr->t[5] = 0;
	pr->t[6] = 0;
	if (check_second_order(pr, LIM_ACC0, aMax, aMin, vMax, vMin)) s1_add_profile(s);
}

/* time_none (position-second-step1.cpp:26-66) */
static void s1_time_none(step1_t* s, double vMax, double vMin, double aMax, double aMin,
						 int return_after_found) {
	const double v0 = s->v0, vf = s->vf, pd = s->pd;
	double h1 = (aMax * vf * vf - aMin * v0 * v0 - 2 * aMax * aMin * pd) / (aMax - aMin);
	if (h1 >= 0.0) {
		h1 = sqrt(h1);
		otg_profile* pr = &s->valid[s->count];
		pr->t[3] = 0;
		pr->t[4] = 0;
		pr->t[5] = 0;
		pr->t[6] = 0;
		/* solution 1 */
		pr->t[0] = -(v0 + h1) / aMax;
		pr->t[1] = 0;
		pr->t[2] = (vf + h1) / aMin;
		if (check_second_order(pr, LIM_NONE, aMax, aMin, vMax, vMin)) {
			s1_add_profile(s);
			if (return_after_found) return;
			pr = &s->valid[s->count];
			/* fresh slot: t[3..6] defined as 0 (see file header) */
			pr->t[3] = 0;
			pr->t[4] = 0;
			pr->t[5] = 0;
			pr->t[6] = 0;
		}
		/* solution 2 */
		pr->t[0] = (-v0 + h1) / aMax;
		pr->t[1] = 0;
		pr->t[2] = (vf - h1) / aMin;
		if (check_second_order(pr, LIM_NONE, aMax, aMin, vMax, vMin)) s1_add_profile(s);
	}
}

/* get_profile (position-second-step1.cpp:100-136); the zero-limits special case (:101-114) cannot
 * occur: the wrappers reject limits <= 0 (OTG_joints.cpp:50-53, OTG_6dof_cartesian.cpp:86-90) */
static int step1_get_profile(const otg_profile* input, otg_block* block, double vMaxIn,
							 double vMinIn, double aMaxIn, double aMinIn) {
	step1_t s;
	s.v0 = input->v[0];
	s.vf = input->vf;
	s.vMax = vMaxIn;
	s.vMin = vMinIn;
	s.aMax = aMaxIn;
	s.aMin = aMinIn;
	s.pd = input->pf - input->p[0];
	s.count = 0;
	memset(s.valid, 0, sizeof(s.valid));
	set_boundary_from(&s.valid[0], input);

	if (fabs(s.vf) < DBL_EPSILON) {
		const double vMax = (s.pd >= 0) ? s.vMax : s.vMin;
		const double vMin = (s.pd >= 0) ? s.vMin : s.vMax;
		const double aMax = (s.pd >= 0) ? s.aMax : s.aMin;
		const double aMin = (s.pd >= 0) ? s.aMin : s.aMax;
		s1_time_none(&s, vMax, vMin, aMax, aMin, 1);
		if (s.count > 0) goto return_block;
		s1_time_acc0(&s, vMax, vMin, aMax, aMin);
		if (s.count > 0) goto return_block;
		s1_time_none(&s, vMin, vMax, aMin, aMax, 1);
		if (s.count > 0) goto return_block;
		s1_time_acc0(&s, vMin, vMax, aMin, aMax);
	} else {
		s1_time_none(&s, s.vMax, s.vMin, s.aMax, s.aMin, 0);
		s1_time_none(&s, s.vMin, s.vMax, s.aMin, s.aMax, 0);
		s1_time_acc0(&s, s.vMax, s.vMin, s.aMax, s.aMin);
		s1_time_acc0(&s, s.vMin, s.vMax, s.aMin, s.aMax);
	}
return_block:
	return calculate_block(block, s.valid, s.count);
}

/* ------------------------------------------------------------------ Step 2
 * PositionSecondOrderStep2 (position-second-step2.cpp) */

typedef struct {
	double v0, tf, vf, vMax, vMin, aMax, aMin, pd, vd;
} step2_t;

/* time_acc0 (position-second-step2.cpp:14-68) */
static int s2_time_acc0(const step2_t* s, otg_profile* pr, double vMax, double vMin, double aMax,
						double aMin) {
	const double v0 = s->v0, tf = s->tf, vf = s->vf, pd = s->pd, vd = s->vd;
	{ /* UD */
		const double h1 = sqrt(
			(2 * aMax * (pd - tf * vf) - 2 * aMin * (pd - tf * v0) + vd * vd) / (aMax * aMin) +
			tf * tf);
		pr->t[0] = (aMax * vd - aMax * aMin * (tf - h1)) / (aMax * (aMax - aMin));
		pr->t[1] = h1;
		pr->t[2] = tf - (pr->t[0] + h1);
		pr->t[3] = 0;
		pr->t[4] = 0;
		pr->t[5] = 0;
		pr->t[6] = 0;
		if (check_second_order(pr, LIM_ACC0, aMax, aMin, vMax, vMin)) {
			pr->pf = pr->p[7];
			return 1;
		}
	}
	{ /* UU */
		const double h1 = (-vd + aMax * tf);
		pr->t[0] = -vd * vd / (2 * aMax * h1) + (pd - v0 * tf) / h1;
		pr->t[1] = -vd / aMax + tf;
		pr->t[2] = 0;
		pr->t[3] = 0;
		pr->t[4] = 0;
		pr->t[5] = 0;
		pr->t[6] = tf - (pr->t[0] + pr->t[1]);
		if (check_second_order(pr, LIM_ACC0, aMax, aMin, vMax, vMin)) {
			pr->pf = pr->p[7];
			return 1;
		}
	}
	{ /* UU, two steps */
		pr->t[0] = 0;
		pr->t[1] = -vd / aMax + tf;
		pr->t[2] = 0;
		pr->t[3] = 0;
		pr->t[4] = 0;
		pr->t[5] = 0;
		pr->t[6] = vd / aMax;
		if (check_second_order(pr, LIM_ACC0, aMax, aMin, vMax, vMin)) {
			pr->pf = pr->p[7];
			return 1;
		}
	}
	return 0;
}

/* time_none (position-second-step2.cpp:70-107) */
static int s2_time_none(const step2_t* s, otg_profile* pr, double vMax, double vMin, double aMax,
						double aMin) {
	const double v0 = s->v0, tf = s->tf, vf = s->vf, pd = s->pd, vd = s->vd;
	if (fabs(v0) < DBL_EPSILON && fabs(vf) < DBL_EPSILON && fabs(pd) < DBL_EPSILON) {
		pr->t[0] = 0;
		pr->t[1] = tf;
		pr->t[2] = 0;
		pr->t[3] = 0;
		pr->t[4] = 0;
		pr->t[5] = 0;
		pr->t[6] = 0;
		if (check_second_order(pr, LIM_NONE, aMax, aMin, vMax, vMin)) {
			pr->pf = pr->p[7];
			return 1;
		}
	}
	{
		const double h1 = 2 * (vf * tf - pd);
		pr->t[0] = h1 / vd;
		pr->t[1] = tf - pr->t[0];
		pr->t[2] = 0;
		pr->t[3] = 0;
		pr->t[4] = 0;
		pr->t[5] = 0;
		pr->t[6] = 0;
		const double af = vd * vd / h1;
		if ((aMin - 1e-12 < af) && (af < aMax + 1e-12) &&
			check_second_order(pr, LIM_NONE, af, -af, vMax, vMin)) {
			pr->pf = pr->p[7];
			return 1;
		}
	}
	return 0;
}

static int s2_check_all(const step2_t* s, otg_profile* pr, double vMax, double vMin, double aMax,
						double aMin) {
	return s2_time_acc0(s, pr, vMax, vMin, aMax, aMin) || s2_time_none(s, pr, vMax, vMin, aMax, aMin);
}

/* get_profile (position-second-step2.cpp:109-117) */
static int step2_get_profile(otg_profile* pr, double tf, double vMax, double vMin, double aMax,
							 double aMin) {
	step2_t s;
	s.v0 = pr->v[0];
	s.tf = tf;
	s.vf = pr->vf;
	s.vMax = vMax;
	s.vMin = vMin;
	s.aMax = aMax;
	s.aMin = aMin;
	s.pd = pr->pf - pr->p[0];
	s.vd = pr->vf - pr->v[0];
	if (s.pd > 0)
		return s2_check_all(&s, pr, vMax, vMin, aMax, aMin) ||
			   s2_check_all(&s, pr, vMin, vMax, aMin, aMax);
	return s2_check_all(&s, pr, vMin, vMax, aMin, aMax) ||
		   s2_check_all(&s, pr, vMax, vMin, aMax, aMin);
}

/* ------------------------------------------------------------------ TargetCalculator */

/* is_input_collinear (calculator_target.hpp:46-118), max_jerk = inf on every DoF */
static int is_input_collinear(const otg_input* inp, int limiting_direction, int limiting_dof,
							  double* new_phase_control) {
	const int n = inp->n;
	double pd[OTG_MAX_DOF];
	int dof, scale_dof = -1;
	const double* scale_vector = 0;
	const double zeros[OTG_MAX_DOF] = {0};
	for (dof = 0; dof < n; dof++) pd[dof] = inp->tp[dof] - inp->cp[dof];
	for (dof = 0; dof < n; dof++) {
		if (fabs(pd[dof]) > EPS) {
			scale_vector = pd;
			scale_dof = dof;
			break;
		} else if (fabs(inp->cv[dof]) > EPS) {
			scale_vector = inp->cv;
			scale_dof = dof;
			break;
		} else if (fabs(inp->ca[dof]) > EPS) {
			scale_vector = inp->ca;
			scale_dof = dof;
			break;
		} else if (fabs(inp->tv[dof]) > EPS) {
			scale_vector = inp->tv;
			scale_dof = dof;
			break;
		}
		/* target acceleration is identically 0 */
	}
	if (scale_dof < 0) return 0;

	const double scale = scale_vector[scale_dof];
	const double pd_scale = pd[scale_dof] / scale;
	const double v0_scale = inp->cv[scale_dof] / scale;
	const double vf_scale = inp->tv[scale_dof] / scale;
	const double a0_scale = inp->ca[scale_dof] / scale;
	const double af_scale = zeros[scale_dof] / scale;
	const double scale_limiting = scale_vector[limiting_dof];
	const double control_limiting =
		(limiting_direction == DIR_UP) ? inp->amax[limiting_dof] : -inp->amax[limiting_dof];

	for (dof = 0; dof < n; dof++) {
		const double current_scale = scale_vector[dof];
		if (fabs(pd[dof] - pd_scale * current_scale) > EPS ||
			fabs(inp->cv[dof] - v0_scale * current_scale) > EPS ||
			fabs(inp->ca[dof] - a0_scale * current_scale) > EPS ||
			fabs(inp->tv[dof] - vf_scale * current_scale) > EPS ||
			fabs(0.0 - af_scale * current_scale) > EPS) {
			return 0;
		}
		new_phase_control[dof] = control_limiting * current_scale / scale_limiting;
	}
	return 1;
}

/* synchronize (calculator_target.hpp:120-222), t_min = nullopt, continuous durations */
static int synchronize(const otg_block* blocks, int n, double* t_sync, int* limiting_dof,
					   otg_profile* profiles) {
	double cand[3 * OTG_MAX_DOF + 1];
	int idx[3 * OTG_MAX_DOF + 1];
	int dof, i, j, any_interval = 0;
	for (dof = 0; dof < n; dof++) {
		cand[dof] = blocks[dof].t_min;
		cand[n + dof] = blocks[dof].a.valid ? blocks[dof].a.right : INFINITY;
		cand[2 * n + dof] = blocks[dof].b.valid ? blocks[dof].b.right : INFINITY;
		any_interval |= blocks[dof].a.valid || blocks[dof].b.valid;
	}
	cand[3 * n] = INFINITY;
	const int n_idx = any_interval ? 3 * n + 1 : n;
	for (i = 0; i < n_idx; i++) idx[i] = i;
	for (i = 1; i < n_idx; i++) { /* stable insertion sort */
		const int k = idx[i];
		for (j = i; j > 0 && cand[k] < cand[idx[j - 1]]; j--) idx[j] = idx[j - 1];
		idx[j] = k;
	}
	for (i = n - 1; i < n_idx; i++) {
		const double t = cand[idx[i]];
		int blocked = 0;
		for (dof = 0; dof < n; dof++)
			if (is_blocked(&blocks[dof], t)) {
				blocked = 1;
				break;
			}
		if (blocked || t < 0.0 || isinf(t)) continue;
		*t_sync = t;
		if (idx[i] == 3 * n) {
			*limiting_dof = -1;
			return 1;
		}
		const int quot = idx[i] / n, rem = idx[i] % n;
		*limiting_dof = rem;
		if (quot == 0)
			profiles[rem] = blocks[rem].p_min;
		else if (quot == 1)
			profiles[rem] = blocks[rem].a.profile;
		else
			profiles[rem] = blocks[rem].b.profile;
		return 1;
	}
	return 0;
}

/* InputParameter::validate(false, true) (input_parameter.hpp:153-330), max_jerk = inf */
static int validate_input(const otg_input* inp) {
	int dof;
	for (dof = 0; dof < inp->n; dof++) {
		const double aMax = inp->amax[dof], vMax = inp->vmax[dof];
		if (isnan(aMax) || aMax < 0.0) return 0;
		if (isnan(inp->ca[dof])) return 0;
		if (isnan(inp->cv[dof]) || isnan(inp->tv[dof])) return 0;
		if (isnan(inp->cp[dof]) || isnan(inp->tp[dof])) return 0;
		if (isnan(vMax) || vMax < 0.0) return 0;
		if (inp->tv[dof] > vMax) return 0;
		if (inp->tv[dof] < -vMax) return 0;
	}
	return 1;
}

/* TargetCalculator::calculate (calculator_target.hpp:249-532) */
int otg_calculate(const otg_input* inp, otg_traj* traj) {
	const int n = inp->n;
	otg_block blocks[OTG_MAX_DOF];
	double new_phase_control[OTG_MAX_DOF];
	int dof;

	for (dof = 0; dof < n; dof++) {
		otg_profile* p = &traj->prof[dof];
		const double vMax = inp->vmax[dof], vMin = -inp->vmax[dof];
		const double aMax = inp->amax[dof], aMin = -inp->amax[dof];

		/* BrakeProfile::get_second_order_position_brake_trajectory (brake.cpp:79-99) */
		p->brake.t = 0.0;
		p->brake.a = 0.0;
		if (!(aMax == 0.0 || aMin == 0.0)) {
			if (inp->cv[dof] > vMax) {
				p->brake.a = aMin;
				p->brake.t = (vMax - inp->cv[dof]) / aMin + brake_eps;
			} else if (inp->cv[dof] < vMin) {
				p->brake.a = aMax;
				p->brake.t = (vMin - inp->cv[dof]) / aMax + brake_eps;
			}
		}
		/* Profile::set_boundary (profile.hpp:294-301) */
		p->a[0] = inp->ca[dof];
		p->v[0] = inp->cv[dof];
		p->p[0] = inp->cp[dof];
		p->af = 0.0;
		p->vf = inp->tv[dof];
		p->pf = inp->tp[dof];
		/* BrakeProfile::finalize_second_order (brake.hpp:66-75) */
		if (p->brake.t <= 0.0) {
			p->brake.duration = 0.0;
		} else {
			const double t = p->brake.t, ps = p->p[0], vs = p->v[0], ab = p->brake.a;
			p->brake.duration = t;
			p->brake.p = ps;
			p->brake.v = vs;
			p->p[0] = ps + t * (vs + t * (ab / 2 + t * 0.0 / 6));
			p->v[0] = vs + t * (ab + t * 0.0 / 2);
			p->a[0] = ab + t * 0.0;
		}
		if (!step1_get_profile(p, &blocks[dof], vMax, vMin, aMax, aMin))
			return OTG_ERROR_EXECUTION_TIME_CALCULATION;
	}

	if (n == 1) {
		traj->duration = blocks[0].t_min;
		traj->prof[0] = blocks[0].p_min;
		return OTG_WORKING;
	}

	int limiting_dof = -1;
	if (!synchronize(blocks, n, &traj->duration, &limiting_dof, traj->prof))
		return OTG_ERROR_SYNCHRONIZATION_CALCULATION;

	if (traj->duration > 7.6e3) return OTG_ERROR_TRAJECTORY_DURATION;

	if (traj->duration == 0.0) {
		for (dof = 0; dof < n; dof++) traj->prof[dof] = blocks[dof].p_min;
		return OTG_WORKING;
	}

	/* phase synchronisation (calculator_target.hpp:398-467) */
	if (limiting_dof >= 0 && inp->synchronization == OTG_SYNC_PHASE) {
		const otg_profile* pl = &traj->prof[limiting_dof];
		if (is_input_collinear(inp, pl->direction, limiting_dof, new_phase_control)) {
			int found = 1;
			for (dof = 0; dof < n; dof++) {
				if (dof == limiting_dof) continue;
				otg_profile* p = &traj->prof[dof];
				/* t_profile is unused by the check (profile.hpp:353-356) */
				memcpy(p->t, pl->t, sizeof(p->t));
				p->control_signs = pl->control_signs;
				found &= check_second_order_timing_limits(
					p, LIM_NONE, new_phase_control[dof], -new_phase_control[dof], inp->vmax[dof],
					-inp->vmax[dof], inp->amax[dof], -inp->amax[dof]);
				p->limits = pl->limits;
			}
			if (found) return OTG_WORKING;
		}
	}

	/* time synchronisation (calculator_target.hpp:469-529) */
	for (dof = 0; dof < n; dof++) {
		if (dof == limiting_dof) continue;
		otg_profile* p = &traj->prof[dof];
		const double t_profile = traj->duration - p->brake.duration - 0.0;
		if (fabs(t_profile - blocks[dof].t_min) < 2 * EPS) {
			*p = blocks[dof].p_min;
			continue;
		} else if (blocks[dof].a.valid && fabs(t_profile - blocks[dof].a.right) < 2 * EPS) {
			*p = blocks[dof].a.profile;
			continue;
		} else if (blocks[dof].b.valid && fabs(t_profile - blocks[dof].b.right) < 2 * EPS) {
			*p = blocks[dof].b.profile;
			continue;
		}
		if (!step2_get_profile(p, t_profile, inp->vmax[dof], -inp->vmax[dof], inp->amax[dof],
							   -inp->amax[dof]))
			return OTG_ERROR_SYNCHRONIZATION_CALCULATION;
	}
	return OTG_WORKING;
}

/* utils.hpp:43-49 with j = 0 */
static void integrate0(double t, double p0, double v0, double a0, double* p, double* v, double* a) {
	*p = p0 + t * (v0 + t * (a0 / 2 + t * 0.0 / 6));
	*v = v0 + t * (a0 + t * 0.0 / 2);
	*a = a0 + t * 0.0;
}

/* Trajectory::state_to_integrate_from / at_time (trajectory.hpp:65-142,182-193), one section */
void otg_at_time(const otg_traj* traj, int n, double time, double* p, double* v, double* a) {
	int dof, i;
	if (time >= traj->duration) {
		for (dof = 0; dof < n; dof++) {
			const otg_profile* pr = &traj->prof[dof];
			const double t_pre = pr->brake.duration;
			const double t_diff = time - (t_pre + pr->t_sum[6]);
			integrate0(t_diff, pr->p[7], pr->v[7], pr->a[7], &p[dof], &v[dof], &a[dof]);
		}
		return;
	}
	for (dof = 0; dof < n; dof++) {
		const otg_profile* pr = &traj->prof[dof];
		double t_diff = time;
		if (pr->brake.duration > 0) {
			if (t_diff < pr->brake.duration) {
				/* second order: brake.t[1] = 0, so the index is 0 when t_diff < brake.t[0]; the
				 * branch index 1 (t_diff >= t[0] but < duration = t[0]) cannot be taken */
				integrate0(t_diff, pr->brake.p, pr->brake.v, pr->brake.a, &p[dof], &v[dof], &a[dof]);
				continue;
			} else {
				t_diff -= pr->brake.duration;
			}
		}
		if (t_diff >= pr->t_sum[6]) {
			integrate0(t_diff - pr->t_sum[6], pr->p[7], pr->v[7], pr->a[7], &p[dof], &v[dof],
					   &a[dof]);
			continue;
		}
		/* std::upper_bound(t_sum, t_diff): first index with t_sum[i] > t_diff */
		int index = 7;
		for (i = 0; i < 7; i++)
			if (pr->t_sum[i] > t_diff) {
				index = i;
				break;
			}
		if (index > 0) t_diff -= pr->t_sum[index - 1];
		integrate0(t_diff, pr->p[index], pr->v[index], pr->a[index], &p[dof], &v[dof], &a[dof]);
	}
}

/* ---- jerk-limited planning: the reference's own ruckig (oracle/_ref), see otg_oracle.h (3) ---- */
typedef int (*plan_jerk_fn)(int, int, const double*, const double*, const double*, const double*, const double*, const double*,
							const double*, const double*, double*, double*);
static plan_jerk_fn g_plan_jerk;
static int g_plan_jerk_tried;
static plan_jerk_fn jerk_planner(void) {
#pragma omp critical(otg_jerk_planner)
	if (!g_plan_jerk_tried) {
		char path[4096];
		const char* env = getenv("SAI2B_RUCKIG_REF");
		void* h = NULL;
		g_plan_jerk_tried = 1;
		if (env && *env) h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
		if (!h) { /* next to this library: <dir>/_ref/libruckig_ref.so */
			Dl_info info;
			if (dladdr((void*)&jerk_planner, &info) && info.dli_fname) {
				const char* slash = strrchr(info.dli_fname, '/');
				const int len = slash ? (int)(slash - info.dli_fname) : 0;
				snprintf(path, sizeof path, "%.*s%s_ref/libruckig_ref.so", len, info.dli_fname, slash ? "/" : "");
				h = dlopen(path, RTLD_NOW | RTLD_LOCAL);
			}
		}
		if (h) g_plan_jerk = (plan_jerk_fn)dlsym(h, "rref_plan_jerk");
	}
	return g_plan_jerk;
}
int otg_jerk_planner_available(void) { return jerk_planner() != NULL; }
static int jerk_limited(const otg_input* inp) {
	int i;
	for (i = 0; i < inp->n; i++)
		if (!isinf(inp->jmax[i])) return 1;
	return 0;
}
static void integrate3(double t, double p0, double v0, double a0, double j, double* p, double* v, double* a) { /* utils.hpp:43-49 */
	*p = p0 + t * (v0 + t * (a0 / 2 + t * j / 6));
	*v = v0 + t * (a0 + t * j / 2);
	*a = a0 + t * j;
}
/* Trajectory::at_time (trajectory.hpp:65-142) on the numbers of rref_plan_jerk */
static void at_time3(const otg_traj* traj, int n, double time, double* p, double* v, double* a) {
	int dof, i;
	for (dof = 0; dof < n; dof++) {
		const double* o = traj->prof3[dof];
		const double brake_duration = o[0], *bt = o + 1, *bj = o + 3, *ba = o + 5, *bv = o + 7, *bp = o + 9;
		const double *t_sum = o + 11, *j = o + 18, *pa = o + 25, *pv = o + 33, *pp = o + 41;
		double t_diff = time;
		if (time >= traj->duration) {
			integrate3(time - (brake_duration + t_sum[6]), pp[7], pv[7], pa[7], 0.0, &p[dof], &v[dof], &a[dof]);
			continue;
		}
		if (brake_duration > 0) {
			if (t_diff < brake_duration) {
				const int index = (t_diff < bt[0]) ? 0 : 1;
				if (index > 0) t_diff -= bt[index - 1];
				integrate3(t_diff, bp[index], bv[index], ba[index], bj[index], &p[dof], &v[dof], &a[dof]);
				continue;
			}
			t_diff -= brake_duration;
		}
		if (t_diff >= t_sum[6]) {
			integrate3(t_diff - t_sum[6], pp[7], pv[7], pa[7], 0.0, &p[dof], &v[dof], &a[dof]);
			continue;
		}
		int index = 7;
		for (i = 0; i < 7; i++)
			if (t_sum[i] > t_diff) {
				index = i;
				break;
			}
		if (index > 0) t_diff -= t_sum[index - 1];
		integrate3(t_diff, pp[index], pv[index], pa[index], j[index], &p[dof], &v[dof], &a[dof]);
	}
}

static int input_differs(const otg_input* x, const otg_input* y) {
	int i;
	if (x->n != y->n || x->synchronization != y->synchronization) return 1;
	for (i = 0; i < x->n; i++) {
		if (!(x->cp[i] == y->cp[i] && x->cv[i] == y->cv[i] && x->ca[i] == y->ca[i] &&
			  x->tp[i] == y->tp[i] && x->tv[i] == y->tv[i] && x->vmax[i] == y->vmax[i] &&
			  x->amax[i] == y->amax[i] && x->jmax[i] == y->jmax[i]))
			return 1;
	}
	return 0;
}

/* Ruckig::update (ruckig.hpp:180-216) */
int otg_update(otg_ruckig* otg, const otg_input* inp, otg_output* out) {
	int i, result = OTG_WORKING;
	out->new_calculation = 0;
	if (input_differs(inp, &otg->current_input) || !otg->current_input_initialized) {
		/* Ruckig::calculate (ruckig.hpp:171-177) */
		if (jerk_limited(inp)) { /* the reference's own planner (validation included) */
			plan_jerk_fn plan = jerk_planner();
			if (!plan) return OTG_ERROR_NO_REFERENCE_PLANNER;
			result = plan(inp->n, inp->synchronization, inp->cp, inp->cv, inp->ca, inp->tp, inp->tv, inp->vmax, inp->amax, inp->jmax,
						  &out->traj.duration, &out->traj.prof3[0][0]);
			if (result != OTG_WORKING) return result;
			out->traj.third_order = 1;
		} else {
			if (!validate_input(inp)) return OTG_ERROR_INVALID_INPUT;
			result = otg_calculate(inp, &out->traj);
			if (result != OTG_WORKING) return result;
			out->traj.third_order = 0;
		}
		otg->current_input = *inp;
		otg->current_input_initialized = 1;
		out->time = 0.0;
		out->new_calculation = 1;
	}
	out->time += otg->delta_time;
	if (out->traj.third_order)
		at_time3(&out->traj, inp->n, out->time, out->np, out->nv, out->na);
	else
		otg_at_time(&out->traj, inp->n, out->time, out->np, out->nv, out->na);
	/* output.pass_to_input(current_input) (ruckig.hpp:209) */
	for (i = 0; i < inp->n; i++) {
		otg->current_input.cp[i] = out->np[i];
		otg->current_input.cv[i] = out->nv[i];
		otg->current_input.ca[i] = out->na[i];
	}
	if (out->time > out->traj.duration) return OTG_FINISHED;
	return result;
}

/* ================================================================== sai2 wrappers */

/* Eigen DenseBase::isApprox: |a-b|^2 <= prec^2 min(|a|^2, |b|^2) */
static int is_approx(const double* a, const double* b, int n, double prec) {
	double d = 0, na = 0, nb = 0;
	int i;
	for (i = 0; i < n; i++) {
		d += (a[i] - b[i]) * (a[i] - b[i]);
		na += a[i] * a[i];
		nb += b[i] * b[i];
	}
	return d <= prec * prec * (na < nb ? na : nb);
}

static void pass_to_input(const otg_output* out, otg_input* in) {
	int i;
	for (i = 0; i < in->n; i++) {
		in->cp[i] = out->np[i];
		in->cv[i] = out->nv[i];
		in->ca[i] = out->na[i];
	}
}

/* ---- OTG_joints ---- */

/* OTG_joints::setGoalPositionAndVelocity (OTG_joints.cpp:98-116) */
void otg_joints_set_goal(otg_joints* o, const double* gp, const double* gv) {
	int i;
	if (o->target_set && is_approx(gp, o->input.tp, o->dim, 1e-12) &&
		is_approx(gv, o->input.tv, o->dim, 1e-12))
		return;
	o->goal_reached = 0;
	o->target_set = 1;
	for (i = 0; i < o->dim; i++) {
		o->input.tp[i] = gp[i];
		o->input.tv[i] = gv[i];
	}
}

/* OTG_joints::reInitialize (OTG_joints.cpp:28-41) */
void otg_joints_reinitialize(otg_joints* o, const double* x0) {
	const double zeros[OTG_MAX_DOF] = {0};
	int i;
	otg_joints_set_goal(o, x0, zeros);
	for (i = 0; i < o->dim; i++) {
		o->output.np[i] = x0[i];
		o->output.nv[i] = 0;
		o->output.na[i] = 0;
	}
	pass_to_input(&o->output, &o->input);
}

/* OTG_joints::OTG_joints (OTG_joints.cpp:17-26) */
void otg_joints_init(otg_joints* o, int dim, const double* x0, double loop_time) {
	int i;
	memset(o, 0, sizeof(*o));
	o->dim = dim;
	o->otg.delta_time = loop_time;
	o->input.n = dim;
	o->input.synchronization = OTG_SYNC_PHASE;
	o->result_value = OTG_FINISHED;
	for (i = 0; i < dim; i++) o->input.amax[i] = o->input.jmax[i] = INFINITY;
	otg_joints_reinitialize(o, x0);
}

/* setMaxVelocity / setMaxAcceleration (OTG_joints.cpp:43-72) */
void otg_joints_set_limits(otg_joints* o, const double* vmax, const double* amax) {
	int i;
	for (i = 0; i < o->dim; i++) {
		o->input.vmax[i] = vmax[i];
		o->input.amax[i] = amax[i];
	}
}

/* OTG_joints::disableJerkLimits (OTG_joints.cpp:88-91) */
void otg_joints_disable_jerk_limits(otg_joints* o) {
	int i;
	for (i = 0; i < o->dim; i++) o->input.ca[i] = 0, o->input.jmax[i] = INFINITY;
}
/* OTG_joints::setMaxJerk (OTG_joints.cpp:73-86) */
void otg_joints_set_max_jerk(otg_joints* o, const double* max_jerk) {
	int i;
	for (i = 0; i < o->dim; i++) o->input.jmax[i] = max_jerk[i];
}

/* OTG_joints::update (OTG_joints.cpp:118-150) */
void otg_joints_update(otg_joints* o) {
	int i;
	if (o->goal_reached) return;
	const otg_output previous_output = o->output;
	o->result_value = otg_update(&o->otg, &o->input, &o->output);

	if (o->result_value == OTG_FINISHED) {
		double nrm = 0;
		for (i = 0; i < o->dim; i++) nrm += o->output.nv[i] * o->output.nv[i];
		if (sqrt(nrm) < 1e-3) {
			o->goal_reached = 1;
		} else {
			/* see file header: defined as the Cartesian wrapper's behaviour */
			const double zeros[OTG_MAX_DOF] = {0};
			double tp[OTG_MAX_DOF];
			memcpy(tp, o->input.tp, sizeof(tp));
			otg_joints_set_goal(o, tp, zeros);
		}
		return;
	}
	if (o->result_value == OTG_WORKING) {
		pass_to_input(&o->output, &o->input);
		return;
	}
	/* error: keep the previous output, zero the input velocity and acceleration */
	o->output = previous_output;
	for (i = 0; i < o->dim; i++) {
		o->input.cv[i] = 0;
		o->input.ca[i] = 0;
	}
}

/* ---- rotation helpers ---- */

/* AngleAxisd(Matrix3d): rotation matrix -> quaternion (Shepperd's branches) -> angle-axis;
 * returns angle * axis (used at OTG_6dof_cartesian.cpp:179-182) */
void otg_rot_to_angle_axis_vec(const double* R, double* out) {
	double q[4]; /* x y z w */
	double t = R[0] + R[4] + R[8];
	if (t > 0) {
		t = sqrt(t + 1.0);
		q[3] = 0.5 * t;
		t = 0.5 / t;
		q[0] = (R[7] - R[5]) * t;
		q[1] = (R[2] - R[6]) * t;
		q[2] = (R[3] - R[1]) * t;
	} else {
		int i = 0;
		if (R[4] > R[0]) i = 1;
		if (R[8] > R[i * 3 + i]) i = 2;
		const int j = (i + 1) % 3, k = (j + 1) % 3;
		t = sqrt(R[i * 3 + i] - R[j * 3 + j] - R[k * 3 + k] + 1.0);
		q[i] = 0.5 * t;
		t = 0.5 / t;
		q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
		q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
		q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
	}
	double nrm = sqrt(q[0] * q[0] + q[1] * q[1] + q[2] * q[2]);
	if (nrm != 0) {
		const double angle = 2 * sai2b_det_atan2_pos(nrm, fabs(q[3])); /* bit-reproducible: include/sai2b_detmath.h */
		if (q[3] < 0) nrm = -nrm;
		out[0] = angle * (q[0] / nrm);
		out[1] = angle * (q[1] / nrm);
		out[2] = angle * (q[2] / nrm);
	} else {
		out[0] = out[1] = out[2] = 0; /* angle 0, axis (1,0,0) */
	}
}

/* getNextOrientation's local rotation (OTG_6dof_cartesian.cpp:226-237): identity below 1e-3 rad,
 * else AngleAxisd(|v|, v/|v|).toRotationMatrix() (Rodrigues) */
void otg_angle_axis_vec_to_rot(const double* v, double* R) {
	const double nrm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
	if (nrm < 1e-3) {
		R[0] = R[4] = R[8] = 1;
		R[1] = R[2] = R[3] = R[5] = R[6] = R[7] = 0;
		return;
	}
	const double ax = v[0] / nrm, ay = v[1] / nrm, az = v[2] / nrm;
	double s, c;
	sai2b_det_sincos(nrm, &s, &c); /* bit-reproducible: include/sai2b_detmath.h */
	const double sx = s * ax, sy = s * ay, sz = s * az;
	const double cx = (1 - c) * ax, cy = (1 - c) * ay, cz = (1 - c) * az;
	double tmp;
	tmp = cx * ay;
	R[1] = tmp - sz;
	R[3] = tmp + sz;
	tmp = cx * az;
	R[2] = tmp + sy;
	R[6] = tmp - sy;
	tmp = cy * az;
	R[5] = tmp - sx;
	R[7] = tmp + sx;
	R[0] = cx * ax + c;
	R[4] = cy * ay + c;
	R[8] = cz * az + c;
}

static void mat3_mul(const double* A, const double* B, double* C) {
	int i, j, k;
	for (i = 0; i < 3; i++)
		for (j = 0; j < 3; j++) {
			double s = 0;
			for (k = 0; k < 3; k++) s += A[i * 3 + k] * B[k * 3 + j];
			C[i * 3 + j] = s;
		}
}
static void mat3_tmul(const double* A, const double* B, double* C) { /* A^T B */
	int i, j, k;
	for (i = 0; i < 3; i++)
		for (j = 0; j < 3; j++) {
			double s = 0;
			for (k = 0; k < 3; k++) s += A[k * 3 + i] * B[k * 3 + j];
			C[i * 3 + j] = s;
		}
}
static void mat3_vec(const double* A, const double* x, double* y) {
	int i;
	for (i = 0; i < 3; i++) y[i] = A[i * 3] * x[0] + A[i * 3 + 1] * x[1] + A[i * 3 + 2] * x[2];
}
static void mat3_tvec(const double* A, const double* x, double* y) {
	int i;
	for (i = 0; i < 3; i++) y[i] = A[i] * x[0] + A[3 + i] * x[1] + A[6 + i] * x[2];
}

/* ---- OTG_6dof_cartesian ---- */

/* getNextOrientation (OTG_6dof_cartesian.cpp:226-237) */
void otg_cartesian_next_orientation(const otg_cartesian* o, double* rot) {
	double local[9];
	otg_angle_axis_vec_to_rot(&o->output.np[3], local);
	mat3_mul(o->reference_frame, local, rot);
}

/* getNextAngularVelocity / getNextAngularAcceleration (OTG_6dof_cartesian.h:222-227) */
void otg_cartesian_next_angular(const otg_cartesian* o, double* w, double* dw) {
	mat3_vec(o->reference_frame, &o->output.nv[3], w);
	mat3_vec(o->reference_frame, &o->output.na[3], dw);
}

/* setGoalPositionAndLinearVelocity (OTG_6dof_cartesian.cpp:140-149) */
void otg_cartesian_set_goal_position(otg_cartesian* o, const double* gp, const double* gv) {
	int i;
	if (o->target_pos_set && is_approx(gp, o->input.tp, 3, 1e-3) &&
		is_approx(gv, o->input.tv, 3, 1e-3))
		return;
	o->goal_reached = 0;
	o->target_pos_set = 1;
	for (i = 0; i < 3; i++) {
		o->input.tp[i] = gp[i];
		o->input.tv[i] = gv[i];
	}
}

/* setGoalOrientationAndAngularVelocity (OTG_6dof_cartesian.cpp:151-185); the isValidRotation
 * throw (:153-157) is the caller's responsibility in the batched setting */
void otg_cartesian_set_goal_orientation(otg_cartesian* o, const double* gR, const double* gw) {
	double new_ref[9], R_new_to_prev[9], tmp[3], ref_to_goal[9];
	int i;
	if (o->goal_ori_set && is_approx(o->goal_orientation, gR, 9, 1e-3) &&
		is_approx(o->goal_angular_velocity, gw, 3, 1e-3))
		return;
	o->goal_reached = 0;
	o->goal_ori_set = 1;
	otg_cartesian_next_orientation(o, new_ref);
	mat3_tmul(new_ref, o->reference_frame, R_new_to_prev);
	memcpy(o->reference_frame, new_ref, sizeof(new_ref));
	memcpy(o->goal_orientation, gR, 9 * sizeof(double));
	memcpy(o->goal_angular_velocity, gw, 3 * sizeof(double));

	o->output.np[3] = o->output.np[4] = o->output.np[5] = 0;
	mat3_vec(R_new_to_prev, &o->output.nv[3], tmp);
	for (i = 0; i < 3; i++) o->output.nv[3 + i] = tmp[i];
	mat3_vec(R_new_to_prev, &o->output.na[3], tmp);
	for (i = 0; i < 3; i++) o->output.na[3 + i] = tmp[i];
	pass_to_input(&o->output, &o->input);

	mat3_tmul(o->reference_frame, o->goal_orientation, ref_to_goal);
	otg_rot_to_angle_axis_vec(ref_to_goal, &o->input.tp[3]);
	mat3_tvec(o->reference_frame, o->goal_angular_velocity, &o->input.tv[3]);
}

/* reInitialize (OTG_6dof_cartesian.cpp:46-58) */
void otg_cartesian_reinitialize(otg_cartesian* o, const double* pos, const double* rot) {
	const double zeros[3] = {0, 0, 0};
	int i;
	otg_cartesian_set_goal_position(o, pos, zeros);
	otg_cartesian_set_goal_orientation(o, rot, zeros);
	for (i = 0; i < 6; i++) {
		o->input.cp[i] = o->input.tp[i];
		o->input.cv[i] = 0;
		o->input.ca[i] = 0;
		o->output.np[i] = o->input.tp[i];
		o->output.nv[i] = 0;
		o->output.na[i] = 0;
	}
}

/* reInitializeLinear (OTG_6dof_cartesian.cpp:60-70) */
void otg_cartesian_reinitialize_linear(otg_cartesian* o, const double* pos) {
	const double zeros[3] = {0, 0, 0};
	int i;
	otg_cartesian_set_goal_position(o, pos, zeros);
	for (i = 0; i < 3; i++) {
		o->input.cp[i] = o->input.tp[i];
		o->input.cv[i] = 0;
		o->input.ca[i] = 0;
		o->output.np[i] = o->input.tp[i];
		o->output.nv[i] = 0;
		o->output.na[i] = 0;
	}
}

/* reInitializeAngular (OTG_6dof_cartesian.cpp:72-83) */
void otg_cartesian_reinitialize_angular(otg_cartesian* o, const double* rot) {
	const double zeros[3] = {0, 0, 0};
	int i;
	otg_cartesian_set_goal_orientation(o, rot, zeros);
	for (i = 3; i < 6; i++) {
		o->input.cp[i] = o->input.tp[i];
		o->input.cv[i] = 0;
		o->input.ca[i] = 0;
		o->output.np[i] = o->input.tp[i];
		o->output.nv[i] = 0;
		o->output.na[i] = 0;
	}
}

/* OTG_6dof_cartesian::OTG_6dof_cartesian (OTG_6dof_cartesian.cpp:29-44) */
void otg_cartesian_init(otg_cartesian* o, const double* pos, const double* rot, double loop_time) {
	int i;
	memset(o, 0, sizeof(*o));
	o->otg.delta_time = loop_time;
	o->input.n = 6;
	o->input.synchronization = OTG_SYNC_PHASE;
	o->result_value = OTG_FINISHED;
	for (i = 0; i < 6; i++) o->input.amax[i] = o->input.jmax[i] = INFINITY;
	memcpy(o->reference_frame, rot, 9 * sizeof(double));
	otg_cartesian_reinitialize(o, pos, rot);
}

/* setMaxLinear/AngularVelocity/Acceleration (OTG_6dof_cartesian.cpp:85-124) */
void otg_cartesian_set_limits(otg_cartesian* o, double lv, double la, double av, double aa) {
	int i;
	for (i = 0; i < 3; i++) {
		o->input.vmax[i] = lv;
		o->input.amax[i] = la;
		o->input.vmax[3 + i] = av;
		o->input.amax[3 + i] = aa;
	}
}

void otg_cartesian_set_max_jerk(otg_cartesian* o, double lj, double aj) {
	int i;
	for (i = 0; i < 3; i++) {
		o->input.jmax[i] = (lj > 0 && !isinf(lj)) ? lj : INFINITY;
		o->input.jmax[3 + i] = (aj > 0 && !isinf(aj)) ? aj : INFINITY;
	}
}

/* update (OTG_6dof_cartesian.cpp:187-224) */
void otg_cartesian_update(otg_cartesian* o) {
	int i;
	if (o->goal_reached) return;
	const otg_output previous_output = o->output;
	o->result_value = otg_update(&o->otg, &o->input, &o->output);

	if (o->result_value == OTG_FINISHED) {
		double nrm = 0;
		for (i = 0; i < 6; i++) nrm += o->output.nv[i] * o->output.nv[i];
		if (sqrt(nrm) < 1e-3) {
			o->goal_reached = 1;
		} else {
			const double zeros[3] = {0, 0, 0};
			double tp[3], gR[9];
			memcpy(tp, o->input.tp, sizeof(tp));
			memcpy(gR, o->goal_orientation, sizeof(gR));
			otg_cartesian_set_goal_position(o, tp, zeros);
			otg_cartesian_set_goal_orientation(o, gR, zeros);
		}
		return;
	}
	if (o->result_value == OTG_WORKING) {
		pass_to_input(&o->output, &o->input);
		return;
	}
	o->output = previous_output;
	for (i = 0; i < 6; i++) {
		o->input.cv[i] = 0;
		o->input.ca[i] = 0;
	}
}

/* ================================================================== flat entry points for tests
 * (same shapes as oracle/ruckig_ref_harness.cpp, so one test drives both) */

int otg_test_calculate_and_sample(int dofs, int sync, const double* cp, const double* cv,
								  const double* ca, const double* tp, const double* tv,
								  const double* vmax, const double* amax, double* duration,
								  int n_times, const double* times, double* out_p, double* out_v,
								  double* out_a) {
	otg_input in;
	otg_traj traj;
	int i, k;
	memset(&in, 0, sizeof(in));
	memset(&traj, 0, sizeof(traj));
	in.n = dofs;
	in.synchronization = sync;
	for (i = 0; i < dofs; i++) {
		in.cp[i] = cp[i];
		in.cv[i] = cv[i];
		in.ca[i] = ca[i];
		in.tp[i] = tp[i];
		in.tv[i] = tv[i];
		in.vmax[i] = vmax[i];
		in.amax[i] = amax[i];
	}
	*duration = 0.0;
	if (!validate_input(&in)) return OTG_ERROR_INVALID_INPUT;
	const int r = otg_calculate(&in, &traj);
	if (r != OTG_WORKING) return r;
	*duration = traj.duration;
	for (k = 0; k < n_times; k++)
		otg_at_time(&traj, dofs, times[k], out_p + k * dofs, out_v + k * dofs, out_a + k * dofs);
	return r;
}

typedef struct {
	otg_ruckig otg;
	otg_input input;
	otg_output output;
} otg_test_handle;

#include <stdlib.h>
void* otg_test_create(int dofs, double dt) {
	otg_test_handle* h = (otg_test_handle*)calloc(1, sizeof(otg_test_handle));
	int i;
	h->otg.delta_time = dt;
	h->input.n = dofs;
	for (i = 0; i < dofs; i++) h->input.vmax[i] = h->input.amax[i] = 1.0, h->input.jmax[i] = INFINITY;
	return h;
}
void otg_test_destroy(void* h) { free(h); }
void otg_test_set_synchronization(void* h, int s) { ((otg_test_handle*)h)->input.synchronization = s; }
void otg_test_set_jerk(void* hh, const double* jmax) { /* finite: the jerk-limited interface (the reference's planner) */
	otg_test_handle* h = (otg_test_handle*)hh;
	int i;
	for (i = 0; i < h->input.n; i++) h->input.jmax[i] = jmax[i];
}
void otg_test_set_limits(void* hh, const double* vmax, const double* amax) {
	otg_test_handle* h = (otg_test_handle*)hh;
	int i;
	for (i = 0; i < h->input.n; i++) {
		h->input.vmax[i] = vmax[i];
		h->input.amax[i] = amax[i];
	}
}
void otg_test_set_current(void* hh, const double* p, const double* v, const double* a) {
	otg_test_handle* h = (otg_test_handle*)hh;
	int i;
	for (i = 0; i < h->input.n; i++) {
		h->input.cp[i] = p[i];
		h->input.cv[i] = v[i];
		h->input.ca[i] = a[i];
	}
}
void otg_test_set_target(void* hh, const double* p, const double* v) {
	otg_test_handle* h = (otg_test_handle*)hh;
	int i;
	for (i = 0; i < h->input.n; i++) {
		h->input.tp[i] = p[i];
		h->input.tv[i] = v[i];
	}
}
int otg_test_update(void* hh) {
	otg_test_handle* h = (otg_test_handle*)hh;
	return otg_update(&h->otg, &h->input, &h->output);
}
void otg_test_pass_to_input(void* hh) {
	otg_test_handle* h = (otg_test_handle*)hh;
	pass_to_input(&h->output, &h->input);
}
void otg_test_get_output(void* hh, double* p, double* v, double* a, double* time, double* duration,
						 int* new_calculation) {
	otg_test_handle* h = (otg_test_handle*)hh;
	int i;
	for (i = 0; i < h->input.n; i++) {
		p[i] = h->output.np[i];
		v[i] = h->output.nv[i];
		a[i] = h->output.na[i];
	}
	*time = h->output.time;
	*duration = h->output.traj.duration;
	*new_calculation = h->output.new_calculation;
}

/* wrapper objects behind opaque handles */
void* otg_test_joints_create(int dim, const double* x0, double dt) {
	otg_joints* o = (otg_joints*)calloc(1, sizeof(otg_joints));
	otg_joints_init(o, dim, x0, dt);
	return o;
}
void otg_test_joints_get(const void* h, double* p, double* v, double* a, int* goal_reached, int* result) {
	const otg_joints* o = (const otg_joints*)h;
	int i;
	for (i = 0; i < o->dim; i++) {
		p[i] = o->output.np[i];
		v[i] = o->output.nv[i];
		a[i] = o->output.na[i];
	}
	*goal_reached = o->goal_reached;
	*result = o->result_value;
}
void* otg_test_cartesian_create(const double* pos, const double* rot, double dt) {
	otg_cartesian* o = (otg_cartesian*)calloc(1, sizeof(otg_cartesian));
	otg_cartesian_init(o, pos, rot, dt);
	return o;
}
/* position 3, orientation 9, linear/angular velocity, linear/angular acceleration */
void otg_test_cartesian_get(const void* h, double* pos, double* rot, double* v, double* w, double* a,
							double* al, int* goal_reached, int* result) {
	const otg_cartesian* o = (const otg_cartesian*)h;
	int i;
	for (i = 0; i < 3; i++) {
		pos[i] = o->output.np[i];
		v[i] = o->output.nv[i];
		a[i] = o->output.na[i];
	}
	otg_cartesian_next_orientation(o, rot);
	otg_cartesian_next_angular(o, w, al);
	*goal_reached = o->goal_reached;
	*result = o->result_value;
}
