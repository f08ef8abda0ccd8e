// Host build of the product's OTG device code (sai2-primitives-perso_amd/csrc/sai2b_otg_core.hpp),
// for tests only: the same entry-point names as the oracle's flat test API (oracle/otg_oracle.c),
// so tests/test_otg_core.py can drive this library and the oracle with the same adapters and
// compare them to the reference-generated fixtures without a GPU. The product never loads this.
#include <cstdlib>
#include <cstring>

#include "sai2b_otg_core.hpp"
#include "sai2b_otg3_core.hpp"

using namespace sai2b::otg;

namespace {
struct Handle {
	Gen g;
	int n;
	double dt, epoch;
	double vmax[MAXD], amax[MAXD];
};
Handle* make(int n, double dt) {
	Handle* h = (Handle*)calloc(1, sizeof(Handle));
	h->n = n;
	h->dt = dt;
	h->g.result = FINISHED;
	for (int i = 0; i < MAXD; i++) h->vmax[i] = 0.0, h->amax[i] = INFINITY;
	return h;
}
void pad(const double* x, int n, double (&out)[MAXD]) {
	for (int i = 0; i < MAXD; i++) out[i] = i < n ? x[i] : 0.0;
}
}  // namespace

extern "C" {

int otg_test_calculate_and_sample(int dofs, int sync, const double* cp, const double* cv, const double* ca,
								  const double* tp, const double* tv, const double* vmax, const double* amax,
								  double* duration, int n_times, const double* times, double* out_p,
								  double* out_v, double* out_a) {
	(void)sync;	 // Phase with Time as fallback; for Time-only inputs see the test
	Input in;
	Traj tr;
	double vm[MAXD], am[MAXD];
	pad(cp, dofs, in.cp), pad(cv, dofs, in.cv), pad(ca, dofs, in.ca), pad(tp, dofs, in.tp), pad(tv, dofs, in.tv);
	pad(vmax, dofs, vm), pad(amax, dofs, am);
	*duration = 0.0;
	if (!validate(in, dofs, vm, am)) return ERR_INVALID_INPUT;
	const int r = calculate(in, dofs, vm, am, tr);
	if (r != WORKING) return r;
	*duration = tr.duration;
	for (int k = 0; k < n_times; k++)
		for (int d = 0; d < dofs; d++)
			at_time(tr.dof[d], tr.prof[d], tr.duration, times[k], out_p[k * dofs + d], out_v[k * dofs + d],
					out_a[k * dofs + d]);
	return r;
}

void* otg_test_joints_create(int dim, const double* x0, double dt) {
	Handle* h = make(dim, dt);
	double x[MAXD];
	pad(x0, dim, x);
	joints_reinitialize(h->g, dim, x);
	return h;
}
void otg_joints_set_limits(void* hh, const double* vmax, const double* amax) {
	Handle* h = (Handle*)hh;
	pad(vmax, h->n, h->vmax), pad(amax, h->n, h->amax);
	h->epoch += 1.0;
}
void otg_joints_disable_jerk_limits(void* hh) {
	Handle* h = (Handle*)hh;
	for (int i = 0; i < h->n; i++) h->g.in.ca[i] = 0;
}
void otg_joints_reinitialize(void* hh, const double* x0) {
	Handle* h = (Handle*)hh;
	double x[MAXD];
	pad(x0, h->n, x);
	joints_reinitialize(h->g, h->n, x);
}
void otg_joints_set_goal(void* hh, const double* gp, const double* gv) {
	Handle* h = (Handle*)hh;
	double p[MAXD], v[MAXD];
	pad(gp, h->n, p), pad(gv, h->n, v);
	joints_set_goal(h->g, h->n, p, v);
}
void otg_joints_update(void* hh) {
	Handle* h = (Handle*)hh;
	joints_update(h->g, h->n, h->dt, h->vmax, h->amax, h->epoch);
}
void otg_test_joints_get(const void* hh, double* p, double* v, double* a, int* goal_reached, int* result) {
	const Handle* h = (const Handle*)hh;
	for (int i = 0; i < h->n; i++) p[i] = h->g.np[i], v[i] = h->g.nv[i], a[i] = h->g.na[i];
	*goal_reached = h->g.goal_reached;
	*result = h->g.result;
}

void* otg_test_cartesian_create(const double* pos, const double* rot, double dt) {
	Handle* h = make(6, dt);
	memcpy(h->g.ref, rot, 9 * sizeof(double));
	cart_reinitialize(h->g, pos, rot);
	return h;
}
void otg_cartesian_set_limits(void* hh, double lv, double la, double av, double aa) {
	Handle* h = (Handle*)hh;
	for (int i = 0; i < 3; i++) h->vmax[i] = lv, h->amax[i] = la, h->vmax[3 + i] = av, h->amax[3 + i] = aa;
	h->epoch += 1.0;
}
void otg_cartesian_reinitialize(void* hh, const double* pos, const double* rot) {
	cart_reinitialize(((Handle*)hh)->g, pos, rot);
}
void otg_cartesian_set_goal_position(void* hh, const double* p, const double* v) {
	cart_set_goal_position(((Handle*)hh)->g, p, v);
}
void otg_cartesian_set_goal_orientation(void* hh, const double* R, const double* w) {
	cart_set_goal_orientation(((Handle*)hh)->g, R, w);
}
void otg_cartesian_update(void* hh) {
	Handle* h = (Handle*)hh;
	cart_update(h->g, h->dt, h->vmax, h->amax, h->epoch);
}
void otg_test_cartesian_get(const void* hh, double* pos, double* rot, double* v, double* w, double* a, double* al,
							int* goal_reached, int* result) {
	const Handle* h = (const Handle*)hh;
	for (int i = 0; i < 3; i++) pos[i] = h->g.np[i], v[i] = h->g.nv[i], a[i] = h->g.na[i];
	cart_next_orientation(h->g, rot);
	mat3_vec(h->g.ref, h->g.nv[3], h->g.nv[4], h->g.nv[5], w);
	mat3_vec(h->g.ref, h->g.na[3], h->g.na[4], h->g.na[5], al);
	*goal_reached = h->g.goal_reached;
	*result = h->g.result;
}

// ---- jerk-limited generator (sai2b_otg3_core.hpp): same shape of test API, with a max_jerk vector ----
int otg3_test_calculate_and_sample(int dofs, int sync, const double* cp, const double* cv, const double* ca, const double* tp,
								   const double* tv, const double* vmax, const double* amax, const double* jmax, double* duration,
								   int n_times, const double* times, double* out_p, double* out_v, double* out_a) {
	(void)sync;
	namespace o3 = sai2b::otg3;
	Input in;
	o3::Traj* tr = (o3::Traj*)calloc(1, sizeof(o3::Traj));
	double vm[MAXD], am[MAXD], jm[MAXD];
	pad(cp, dofs, in.cp), pad(cv, dofs, in.cv), pad(ca, dofs, in.ca), pad(tp, dofs, in.tp), pad(tv, dofs, in.tv);
	pad(vmax, dofs, vm), pad(amax, dofs, am), pad(jmax, dofs, jm);
	*duration = 0.0;
	int r = ERR_INVALID_INPUT;
	if (o3::validate(in, dofs, vm, am, jm)) {
		r = o3::calculate(in, dofs, vm, am, jm, *tr);
		if (r == WORKING) {
			*duration = tr->duration;
			for (int k = 0; k < n_times; k++)
				for (int d = 0; d < dofs; d++)
					o3::at_time(tr->prof[d], tr->duration, times[k], out_p[k * dofs + d], out_v[k * dofs + d], out_a[k * dofs + d]);
		}
	}
	free(tr);
	return r;
}

struct Handle3 {
	sai2b::otg3::Gen g;
	int n;
	double dt, epoch;
	double vmax[MAXD], amax[MAXD];
};
void* otg3_test_joints_create(int dim, const double* x0, double dt) {
	Handle3* h = (Handle3*)calloc(1, sizeof(Handle3));
	h->n = dim, h->dt = dt;
	h->g.result = FINISHED;
	double x[MAXD];
	pad(x0, dim, x);
	joints_reinitialize(h->g, dim, x);
	return h;
}
void otg3_joints_set_limits(void* hh, const double* vmax, const double* amax, const double* jmax) {
	Handle3* h = (Handle3*)hh;
	pad(vmax, h->n, h->vmax), pad(amax, h->n, h->amax), pad(jmax, h->n, h->g.jmax);
	h->epoch += 1.0;
}
void otg3_joints_set_goal(void* hh, const double* gp, const double* gv) {
	Handle3* h = (Handle3*)hh;
	double p[MAXD], v[MAXD];
	pad(gp, h->n, p), pad(gv, h->n, v);
	joints_set_goal(h->g, h->n, p, v);
}
void otg3_joints_update(void* hh) {
	Handle3* h = (Handle3*)hh;
	joints_update(h->g, h->n, h->dt, h->vmax, h->amax, h->epoch);
}
void otg3_test_joints_get(const void* hh, double* p, double* v, double* a, int* goal_reached, int* result) {
	const Handle3* h = (const Handle3*)hh;
	for (int i = 0; i < h->n; i++) p[i] = h->g.np[i], v[i] = h->g.nv[i], a[i] = h->g.na[i];
	*goal_reached = h->g.goal_reached;
	*result = h->g.result;
}
}
