/*
 * ruckig_ref_harness.cpp — C entry points around the REAL ruckig core of the reference.
 *
 * TEST INFRASTRUCTURE ONLY (oracle/). The vendored ruckig 0.10.1 core
 * (/root/reference/ruckig/{include,src}) compiles without Eigen when its vectors are
 * std::vector (ruckig/include/ruckig/utils.hpp:18-19), so this part of the reference CAN be built
 * in this container: `make -C oracle ref` compiles the reference's own sources where they lie and
 * links them with this file into oracle/_ref/libruckig_ref.so. It pins the oracle's restatement of
 * the acceleration-limited trajectory generator (oracle/otg_oracle.c) against the reference's own
 * code (tests/test_otg_oracle.py, tests/golden/make_otg_golden.py).
 *
 * The sai2 wrappers around ruckig (src/helper_modules/OTG_joints.cpp, OTG_6dof_cartesian.cpp)
 * need Eigen and cannot be built; they are restated in the oracle.
 */
#include <ruckig/ruckig.hpp>

#include <limits>
#include <vector>

using namespace ruckig;

namespace {
struct Handle {
	Ruckig<DynamicDOFs> otg;
	InputParameter<DynamicDOFs> input;
	OutputParameter<DynamicDOFs> output;
	Handle(size_t n, double dt) : otg(n, dt), input(n), output(n) {
		for (size_t i = 0; i < n; ++i) {
			input.current_position[i] = 0.0;
			input.target_position[i] = 0.0;
			input.max_velocity[i] = 1.0;
			input.max_acceleration[i] = 1.0;
			input.max_jerk[i] = std::numeric_limits<double>::infinity();
			output.new_position[i] = output.new_velocity[i] = output.new_acceleration[i] = 0.0;
		}
	}
};
}  // namespace

extern "C" {

void* rref_create(int dofs, double dt) { return new Handle((size_t)dofs, dt); }
void rref_destroy(void* h) { delete (Handle*)h; }

/* synchronization: 0 Time, 1 TimeIfNecessary, 2 Phase, 3 None (the wrappers use Phase,
 * OTG_joints.cpp:23, OTG_6dof_cartesian.cpp:35) */
void rref_set_synchronization(void* h, int s) {
	((Handle*)h)->input.synchronization = (Synchronization)s;
}

void rref_set_limits(void* hh, const double* vmax, const double* amax) {
	Handle* h = (Handle*)hh;
	for (size_t i = 0; i < h->input.degrees_of_freedom; ++i) {
		h->input.max_velocity[i] = vmax[i];
		h->input.max_acceleration[i] = amax[i];
	}
}

/* a finite max_jerk selects ruckig's third-order (jerk-limited) position interface */
void rref_set_jerk(void* hh, const double* jmax) {
	Handle* h = (Handle*)hh;
	for (size_t i = 0; i < h->input.degrees_of_freedom; ++i) h->input.max_jerk[i] = jmax[i];
}

void rref_set_current(void* hh, const double* p, const double* v, const double* a) {
	Handle* h = (Handle*)hh;
	for (size_t i = 0; i < h->input.degrees_of_freedom; ++i) {
		h->input.current_position[i] = p[i];
		h->input.current_velocity[i] = v[i];
		h->input.current_acceleration[i] = a[i];
	}
}

void rref_set_target(void* hh, const double* p, const double* v) {
	Handle* h = (Handle*)hh;
	for (size_t i = 0; i < h->input.degrees_of_freedom; ++i) {
		h->input.target_position[i] = p[i];
		h->input.target_velocity[i] = v[i];
	}
}

/* Ruckig::update (ruckig.hpp:180-216); returns the Result code (0 Working, 1 Finished, <0 error) */
int rref_update(void* hh) {
	Handle* h = (Handle*)hh;
	return (int)h->otg.update(h->input, h->output);
}

/* OutputParameter::pass_to_input (output_parameter.hpp:70-79) */
void rref_pass_to_input(void* hh) {
	Handle* h = (Handle*)hh;
	h->output.pass_to_input(h->input);
}

void rref_get_output(void* hh, double* p, double* v, double* a, double* time, double* duration,
					 int* new_calculation) {
	Handle* h = (Handle*)hh;
	for (size_t i = 0; i < h->input.degrees_of_freedom; ++i) {
		p[i] = h->output.new_position[i];
		v[i] = h->output.new_velocity[i];
		a[i] = h->output.new_acceleration[i];
	}
	*time = h->output.time;
	*duration = h->output.trajectory.get_duration();
	*new_calculation = h->output.new_calculation ? 1 : 0;
}

/* one-shot: trajectory for the given input, sampled at the given times; returns the Result of
 * Ruckig::calculate (calculator_target.hpp:249) */
int rref_calculate_and_sample(int dofs, int sync, const double* cp, const double* cv,
							  const double* ca, const double* tp, const double* tv,
							  const double* vmax, const double* amax, double* duration,
							  int n_times, const double* times, double* out_p, double* out_v,
							  double* out_a) {
	const size_t n = (size_t)dofs;
	Ruckig<DynamicDOFs> otg(n, 0.001);
	InputParameter<DynamicDOFs> in(n);
	Trajectory<DynamicDOFs> traj(n);
	in.synchronization = (Synchronization)sync;
	for (size_t i = 0; i < n; ++i) {
		in.current_position[i] = cp[i];
		in.current_velocity[i] = cv[i];
		in.current_acceleration[i] = ca[i];
		in.target_position[i] = tp[i];
		in.target_velocity[i] = tv[i];
		in.max_velocity[i] = vmax[i];
		in.max_acceleration[i] = amax[i];
		in.max_jerk[i] = std::numeric_limits<double>::infinity();
	}
	const Result r = otg.calculate(in, traj);
	if (r != Result::Working) {
		*duration = 0.0;
		return (int)r;
	}
	*duration = traj.get_duration();
	std::vector<double> p(n), v(n), a(n);
	for (int k = 0; k < n_times; ++k) {
		traj.at_time(times[k], p, v, a);
		for (size_t i = 0; i < n; ++i) {
			out_p[k * n + i] = p[i];
			out_v[k * n + i] = v[i];
			out_a[k * n + i] = a[i];
		}
	}
	return (int)r;
}

/* the same one-shot calculation with a max_jerk vector (the jerk-limited interface the wrappers select with
 * OTG_joints::setMaxJerk, OTG_joints.cpp:73-86) */
int rref_calculate_and_sample_jerk(int dofs, int sync, const double* cp, const double* cv, const double* ca, const double* tp,
								   const double* tv, const double* vmax, const double* amax, const double* jmax,
								   double* duration, int n_times, const double* times, double* out_p, double* out_v,
								   double* out_a) {
	const size_t n = (size_t)dofs;
	Ruckig<DynamicDOFs> otg(n, 0.001);
	InputParameter<DynamicDOFs> in(n);
	Trajectory<DynamicDOFs> traj(n);
	in.synchronization = (Synchronization)sync;
	for (size_t i = 0; i < n; ++i) {
		in.current_position[i] = cp[i];
		in.current_velocity[i] = cv[i];
		in.current_acceleration[i] = ca[i];
		in.target_position[i] = tp[i];
		in.target_velocity[i] = tv[i];
		in.max_velocity[i] = vmax[i];
		in.max_acceleration[i] = amax[i];
		in.max_jerk[i] = jmax[i];
	}
	const Result r = otg.calculate(in, traj);
	if (r != Result::Working) {
		*duration = 0.0;
		return (int)r;
	}
	*duration = traj.get_duration();
	std::vector<double> p(n), v(n), a(n);
	for (int k = 0; k < n_times; ++k) {
		traj.at_time(times[k], p, v, a);
		for (size_t i = 0; i < n; ++i) {
			out_p[k * n + i] = p[i];
			out_v[k * n + i] = v[i];
			out_a[k * n + i] = a[i];
		}
	}
	return (int)r;
}

/* The reference's jerk-limited planner as the ORACLE's planner: Ruckig::calculate for one input, the resulting
 * trajectory handed back as plain numbers, 49 per DoF: brake.duration, brake.t[2], j[2], a[2], v[2], p[2], then
 * t_sum[7], j[7], a[8], v[8], p[8] of the profile (profile.hpp:46-50) — what Trajectory::at_time needs. oracle/
 * otg_oracle.c samples them itself (its wrappers and Ruckig::update are restated there; the third-order planner is
 * not: for it the oracle IS the reference's code). Returns the Result. */
int rref_plan_jerk(int dofs, int sync, const double* cp, const double* cv, const double* ca, const double* tp, const double* tv,
				   const double* vmax, const double* amax, const double* jmax, double* duration, double* prof) {
	const size_t n = (size_t)dofs;
	Ruckig<DynamicDOFs> otg(n, 0.001);
	InputParameter<DynamicDOFs> in(n);
	Trajectory<DynamicDOFs> traj(n);
	in.synchronization = (Synchronization)sync;
	for (size_t i = 0; i < n; ++i) {
		in.current_position[i] = cp[i];
		in.current_velocity[i] = cv[i];
		in.current_acceleration[i] = ca[i];
		in.target_position[i] = tp[i];
		in.target_velocity[i] = tv[i];
		in.max_velocity[i] = vmax[i];
		in.max_acceleration[i] = amax[i];
		in.max_jerk[i] = jmax[i];
	}
	const Result r = otg.calculate(in, traj);
	*duration = 0.0;
	if (r != Result::Working) return (int)r;
	*duration = traj.get_duration();
	const auto profiles = traj.get_profiles();
	for (size_t d = 0; d < n; ++d) {
		const Profile& p = profiles[0][d];
		double* o = prof + 49 * d;
		o[0] = p.brake.duration;
		for (int k = 0; k < 2; k++) o[1 + k] = p.brake.t[k], o[3 + k] = p.brake.j[k], o[5 + k] = p.brake.a[k], o[7 + k] = p.brake.v[k], o[9 + k] = p.brake.p[k];
		for (int k = 0; k < 7; k++) o[11 + k] = p.t_sum[k], o[18 + k] = p.j[k];
		for (int k = 0; k < 8; k++) o[25 + k] = p.a[k], o[33 + k] = p.v[k], o[41 + k] = p.p[k];
	}
	return (int)r;
}
}
