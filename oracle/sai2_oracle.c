/*
 * sai2_oracle.c — CPU oracle (TEST INFRASTRUCTURE ONLY; see sai2_oracle.h for the rules and the
 * "parity unpinned" statement).
 *
 * One robot at a time, FP64, dynamic sizes, no fast-math: the reference's execution model
 * (one single-threaded control loop per robot, examples/05-using_robot_controller.cpp:138-196)
 * restated in plain C. Every block cites the reference lines it follows.
 */
#include "sai2_oracle.h"
#include "otg_oracle.h"
#include "../include/sai2b_detfk.h"

#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

/* joints of the robots this build of the oracle serves (oracle/Makefile builds one library per size: 4, 6, 7, 8) */
#ifndef N7
#define N7 SAI2B_DOF
#endif
#define NN (N7 * N7) /* n x n */
#define J6 (6 * N7)	 /* 6 x n */
#define J3 (3 * N7)	 /* 3 x n */
#define MAXD (N7 > 7 ? N7 : 7) /* largest matrix dimension anywhere on the path */
#define POPC_RING 1024 /* window ring capacity (the reference queue is unbounded; same cap as the product) */

static char g_err[512] = "";
const char* oracle_last_error(void) { return g_err; }
static int fail(const char* msg) {
	snprintf(g_err, sizeof(g_err), "%s", msg);
	return SAI2B_INVALID_ARGUMENT;
}

/* ------------------------------------------------------------------------------------------
 * small dense helpers (row-major)
 * ------------------------------------------------------------------------------------------ */
static void mm(int m, int k, int n, const double* A, const double* B, double* C) {
	for (int i = 0; i < m; i++)
		for (int j = 0; j < n; j++) {
			double s = 0;
			for (int l = 0; l < k; l++) s += A[i * k + l] * B[l * n + j];
			C[i * n + j] = s;
		}
}
/* C(m x n) = A^T B with A (k x m), B (k x n) */
static void mm_tn(int m, int k, int n, const double* A, const double* B, double* C) {
	for (int i = 0; i < m; i++)
		for (int j = 0; j < n; j++) {
			double s = 0;
			for (int l = 0; l < k; l++) s += A[l * m + i] * B[l * n + j];
			C[i * n + j] = s;
		}
}
/* C(m x n) = A B^T with A (m x k), B (n x k) */
static void mm_nt(int m, int k, int n, const double* A, const double* B, double* C) {
	for (int i = 0; i < m; i++)
		for (int j = 0; j < n; j++) {
			double s = 0;
			for (int l = 0; l < k; l++) s += A[i * k + l] * B[j * k + l];
			C[i * n + j] = s;
		}
}
static void eye(int n, double* A) {
	for (int i = 0; i < n * n; i++) A[i] = 0;
	for (int i = 0; i < n; i++) A[i * n + i] = 1;
}
static void cross3(const double* a, const double* b, double* c) {
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}

/* General inverse by Gauss-Jordan with partial pivoting: stands in for Eigen's
 * MatrixXd::inverse() (SingularityHandler.cpp:120,182,190,201,212; JointTask.cpp:260-265) and for
 * sai2-model's M^-1. Returns nonzero when a pivot is exactly zero (result then holds inf/nan like
 * Eigen's would). */
int oracle_inverse(int n, const double* A, double* Ainv) {
	double a[MAXD * MAXD], b[MAXD * MAXD];
	int rc = 0;
	memcpy(a, A, sizeof(double) * n * n);
	eye(n, b);
	for (int c = 0; c < n; c++) {
		int p = c;
		for (int r = c + 1; r < n; r++)
			if (fabs(a[r * n + c]) > fabs(a[p * n + c])) p = r;
		if (p != c)
			for (int j = 0; j < n; j++) {
				double t = a[c * n + j];
				a[c * n + j] = a[p * n + j];
				a[p * n + j] = t;
				t = b[c * n + j];
				b[c * n + j] = b[p * n + j];
				b[p * n + j] = t;
			}
		double piv = a[c * n + c];
		if (piv == 0) rc = 1;
		double ip = 1.0 / piv;
		for (int j = 0; j < n; j++) {
			a[c * n + j] *= ip;
			b[c * n + j] *= ip;
		}
		for (int r = 0; r < n; r++) {
			if (r == c) continue;
			double f = a[r * n + c];
			if (f == 0) continue;
			for (int j = 0; j < n; j++) {
				a[r * n + j] -= f * a[c * n + j];
				b[r * n + j] -= f * b[c * n + j];
			}
		}
	}
	memcpy(Ainv, b, sizeof(double) * n * n);
	return rc;
}

/* One-sided (Hestenes) Jacobi: orthogonalise the `cols` columns of X (rows x cols, in place) by
 * plane rotations accumulated in W (cols x cols): X_in W = X_out. */
static void hestenes(int rows, int cols, double* X, double* W) {
	eye(cols, W);
	for (int sweep = 0; sweep < 60; sweep++) {
		int rotated = 0;
		for (int i = 0; i < cols - 1; i++)
			for (int j = i + 1; j < cols; j++) {
				double al = 0, be = 0, ga = 0;
				for (int r = 0; r < rows; r++) {
					double xi = X[r * cols + i], xj = X[r * cols + j];
					al += xi * xi;
					be += xj * xj;
					ga += xi * xj;
				}
				if (ga == 0 || fabs(ga) <= 1e-15 * sqrt(al * be)) continue;
				rotated = 1;
				double zeta = (be - al) / (2 * ga);
				double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1 + zeta * zeta));
				double c = 1 / sqrt(1 + t * t), s = c * t;
				for (int r = 0; r < rows; r++) {
					double xi = X[r * cols + i], xj = X[r * cols + j];
					X[r * cols + i] = c * xi - s * xj;
					X[r * cols + j] = s * xi + c * xj;
				}
				for (int r = 0; r < cols; r++) {
					double wi = W[r * cols + i], wj = W[r * cols + j];
					W[r * cols + i] = c * wi - s * wj;
					W[r * cols + j] = s * wi + c * wj;
				}
			}
		if (!rotated) break;
	}
}

/* Thin SVD A (m x n) = U diag(s) V^T with p = min(m,n): U m x p, s p (descending), V n x p.
 * Stands in for Eigen::JacobiSVD(ComputeThinU|ComputeThinV) (SingularityHandler.cpp:78-81).
 * Singular vectors of distinct non-zero singular values are unique up to sign; every use on the
 * path but ONE is sign-invariant: classifySingularity perturbs q along V_s[:, i] (sh_classify below,
 * enum sai2b_singular_vector_sign). Columns belonging to s == 0 on the normalised side are set to zero. */
void oracle_svd(int m, int n, const double* A, double* U, double* s, double* V) {
	double X[MAXD * MAXD], W[MAXD * MAXD];
	int p = m < n ? m : n;
	int rows, cols;
	if (m < n) { /* work on A^T (n x m): A^T W = Q, U = W, V = Q/s */
		rows = n;
		cols = m;
		for (int i = 0; i < m; i++)
			for (int j = 0; j < n; j++) X[j * m + i] = A[i * n + j];
	} else { /* A W = Q, U = Q/s, V = W */
		rows = m;
		cols = n;
		memcpy(X, A, sizeof(double) * m * n);
	}
	hestenes(rows, cols, X, W);
	double sv[MAXD];
	int order[MAXD];
	for (int j = 0; j < cols; j++) {
		double a = 0;
		for (int r = 0; r < rows; r++) a += X[r * cols + j] * X[r * cols + j];
		sv[j] = sqrt(a);
		order[j] = j;
	}
	for (int i = 0; i < cols; i++) /* selection sort, descending, stable */
		for (int j = i + 1; j < cols; j++)
			if (sv[order[j]] > sv[order[i]]) {
				int t = order[i];
				order[i] = order[j];
				order[j] = t;
			}
	for (int k = 0; k < p; k++) {
		int j = order[k];
		s[k] = sv[j];
		double inv = sv[j] > 0 ? 1.0 / sv[j] : 0.0;
		if (m < n) {
			for (int r = 0; r < m; r++) U[r * p + k] = W[r * cols + j];
			for (int r = 0; r < n; r++) V[r * p + k] = X[r * cols + j] * inv;
		} else {
			for (int r = 0; r < m; r++) U[r * p + k] = X[r * cols + j] * inv;
			for (int r = 0; r < n; r++) V[r * p + k] = W[r * cols + j];
		}
		/* sign convention (DEFINED here; Eigen's is unknown): the right singular vector is oriented so
		 * that its largest-magnitude component is positive. Only the FK-perturbation classification
		 * (SingularityHandler.cpp:253-273) depends on the sign. */
		int big = 0;
		for (int r = 1; r < n; r++)
			if (fabs(V[r * p + k]) > fabs(V[big * p + k])) big = r;
		if (V[big * p + k] < 0) {
			for (int r = 0; r < n; r++) V[r * p + k] = -V[r * p + k];
			for (int r = 0; r < m; r++) U[r * p + k] = -U[r * p + k];
		}
	}
}

/* Sai2Model::matrixRangeBasis(A, tol = 1e-3) as DEFINED in SURVEY.md App. D: left singular vectors
 * whose sigma_i/sigma_0 >= tol; identity when full row rank; "zero" (return 0 columns) when
 * sigma_0 < tol. R is m x cols, returns cols. Call sites: JointTask.cpp:233,
 * MotionForceTask.cpp:66,78,147,149. */
int oracle_range_basis(int m, int n, const double* A, double tol, double* R) {
	double U[MAXD * MAXD], s[MAXD], V[MAXD * MAXD];
	int p = m < n ? m : n;
	oracle_svd(m, n, A, U, s, V);
	if (s[0] < tol) return 0;
	int dof = p;
	for (int i = p - 1; i > 0; i--) {
		if (s[i] / s[0] < tol)
			dof--;
		else
			break;
	}
	if (dof == m) {
		eye(m, R);
		return m;
	}
	for (int r = 0; r < m; r++)
		for (int c = 0; c < dof; c++) R[r * dof + c] = U[r * p + c];
	return dof;
}

/* pseudo-inverse of a symmetric matrix (stands in for
 * completeOrthogonalDecomposition().pseudoInverse(), SingularityHandler.cpp:96-98) */
static void sym_pinv(int n, const double* A, double* P) {
	double U[MAXD * MAXD], s[MAXD], V[MAXD * MAXD], T[MAXD * MAXD];
	oracle_svd(n, n, A, U, s, V);
	for (int i = 0; i < n; i++)
		for (int j = 0; j < n; j++)
			T[i * n + j] = (s[j] > n * 2.220446049250313e-16 * s[0] && s[j] > 0) ? V[i * n + j] / s[j] : 0.0;
	mm_nt(n, n, n, T, U, P);
}

/* Sai2Model::orientationError(desired, current) = -1/2 sum_i current[:,i] x desired[:,i]
 * (call sites MotionForceTask.cpp:292,443, SingularityHandler.cpp:260) */
static void orientation_error(const double* Rd, const double* Rc, double* e) {
	e[0] = e[1] = e[2] = 0;
	for (int i = 0; i < 3; i++) {
		double c[3] = {Rc[0 + i], Rc[3 + i], Rc[6 + i]};
		double d[3] = {Rd[0 + i], Rd[3 + i], Rd[6 + i]};
		double x[3];
		cross3(c, d, x);
		for (int k = 0; k < 3; k++) e[k] -= 0.5 * x[k];
	}
}

/* Sai2Model::computePseudoInverse restricted to what the path applies it to: diagonal gain
 * matrices (JointTask.cpp:328, MotionForceTask.cpp:417,450). DEFINED: entries below 1e-6 map to 0. */
static double gain_pinv(double k) { return fabs(k) > 1e-6 ? 1.0 / k : 0.0; }

static void rpy_to_rot(const double* rpy, double* R) {
	double cr = cos(rpy[0]), sr = sin(rpy[0]);
	double cp = cos(rpy[1]), sp = sin(rpy[1]);
	double cy = cos(rpy[2]), sy = sin(rpy[2]);
	R[0] = cy * cp;
	R[1] = cy * sp * sr - sy * cr;
	R[2] = cy * sp * cr + sy * sr;
	R[3] = sy * cp;
	R[4] = sy * sp * sr + cy * cr;
	R[5] = sy * sp * cr - cy * sr;
	R[6] = -sp;
	R[7] = cp * sr;
	R[8] = cp * cr;
}

/* ------------------------------------------------------------------------------------------
 * host-side config helpers (independent restatement)
 * ------------------------------------------------------------------------------------------ */
int oracle_model_merge_fixed_body(sai2b_robot_model* md, int link, const double xyz[3],
								  const double rpy[3], double mass, const double com[3],
								  const double inertia[6]) {
	if (!md || link < 0 || link >= N7) return fail("merge_fixed_body: bad link");
	double Rf[9], cb[3], Ib[9], Iw[9], T[9];
	rpy_to_rot(rpy, Rf);
	for (int i = 0; i < 3; i++)
		cb[i] = xyz[i] + Rf[3 * i] * com[0] + Rf[3 * i + 1] * com[1] + Rf[3 * i + 2] * com[2];
	Ib[0] = inertia[0];
	Ib[4] = inertia[1];
	Ib[8] = inertia[2];
	Ib[1] = Ib[3] = inertia[3];
	Ib[2] = Ib[6] = inertia[4];
	Ib[5] = Ib[7] = inertia[5];
	mm(3, 3, 3, Rf, Ib, T);
	mm_nt(3, 3, 3, T, Rf, Iw); /* child inertia about its COM in parent-link axes */
	double ma = md->link_mass[link], mt = ma + mass;
	double ca[3] = {md->link_com[link][0], md->link_com[link][1], md->link_com[link][2]};
	double cn[3];
	for (int i = 0; i < 3; i++) cn[i] = (ma * ca[i] + mass * cb[i]) / mt;
	double Ia[9];
	const double* li = md->link_inertia[link];
	Ia[0] = li[0];
	Ia[4] = li[1];
	Ia[8] = li[2];
	Ia[1] = Ia[3] = li[3];
	Ia[2] = Ia[6] = li[4];
	Ia[5] = Ia[7] = li[5];
	double In[9];
	for (int i = 0; i < 9; i++) In[i] = Ia[i] + Iw[i];
	/* parallel-axis terms m (|d|^2 I - d d^T) for both bodies about the new COM */
	for (int body = 0; body < 2; body++) {
		double m = body ? mass : ma;
		const double* c = body ? cb : ca;
		double d[3] = {c[0] - cn[0], c[1] - cn[1], c[2] - cn[2]};
		double d2 = d[0] * d[0] + d[1] * d[1] + d[2] * d[2];
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) In[3 * i + j] += m * ((i == j ? d2 : 0.0) - d[i] * d[j]);
	}
	md->link_mass[link] = mt;
	for (int i = 0; i < 3; i++) md->link_com[link][i] = cn[i];
	double* lo = md->link_inertia[link];
	lo[0] = In[0];
	lo[1] = In[4];
	lo[2] = In[8];
	lo[3] = In[1];
	lo[4] = In[2];
	lo[5] = In[5];
	return 0;
}

/* Panda constants: examples/15-haptic_control_impedance_type/panda_arm.urdf:17-116 (inertials),
 * :118-178 (joints), :179-183 (fixed end-effector joint) */
int oracle_panda_model(sai2b_robot_model* md) {
	if (!md) return fail("panda_model: null");
	memset(md, 0, sizeof(*md));
	md->dof = N7;
	static const double xyz[7][3] = {{0, 0, 0.333},		  {0, 0, 0},	 {0, -0.316, 0}, {0.0825, 0, 0},
									 {-0.0825, 0.384, 0}, {0, 0, 0},	 {0.088, 0, 0}};
	static const double roll[7] = {0,			   -1.57079632679, 1.57079632679, 1.57079632679,
								   -1.57079632679, 1.57079632679,  1.57079632679};
	static const double mass[7] = {3, 3, 2, 2, 2, 1.5, 1.8};
	static const double com[7][3] = {{0, 0, -0.07}, {0, -0.1, 0},  {0.04, 0, -0.05}, {-0.04, 0.05, 0},
									 {0, 0, -0.15}, {0.06, 0, 0},  {0, 0, 0.17}};
	static const double idiag[7][3] = {{0.3, 0.3, 0.3}, {0.3, 0.3, 0.3}, {0.2, 0.2, 0.2},
									   {0.2, 0.2, 0.2}, {0.2, 0.2, 0.2}, {0.1, 0.1, 0.1},
									   {0.09, 0.05, 0.07}};
	static const double lo[7] = {-2.8973, -1.7628, -2.8973, -3.0718, -2.8973, -0.0175, -2.8973};
	static const double hi[7] = {2.8973, 1.7628, 2.8973, -0.0698, 2.8973, 3.7525, 2.8973};
	static const double eff[7] = {87, 87, 87, 87, 12, 12, 12};
	for (int i = 0; i < 7; i++) {
		for (int k = 0; k < 3; k++) {
			md->joint_xyz[i][k] = xyz[i][k];
			md->link_com[i][k] = com[i][k];
			md->link_inertia[i][k] = idiag[i][k];
		}
		md->joint_rpy[i][0] = roll[i];
		md->link_mass[i] = mass[i];
		md->q_lower[i] = lo[i];
		md->q_upper[i] = hi[i];
		md->effort[i] = eff[i];
	}
	md->gravity[2] = -9.81;
	const double exyz[3] = {0, 0, 0.15}, erpy[3] = {0, 0, 0}, ecom[3] = {0, 0, 0};
	const double ein[6] = {0.01, 0.01, 0.01, 0, 0, 0};
	return oracle_model_merge_fixed_body(md, 6, exyz, erpy, 0.2, ecom, ein);
}

static void sh_defaults(sai2b_task_config* c) {
	/* SingularityHandler.cpp:10-20, MotionForceTask.cpp:197 */
	c->s_min = 6e-3;
	c->s_max = 6e-2;
	c->s_abs_tol = 1e-3;
	c->type_1_tol = 0.5;
	c->type_2_torque_ratio = 1e-2;
	c->type_2_angle_threshold = 5 * M_PI / 180;
	c->perturb_step_size = 5;
	c->sh_buffer_size = 200;
	c->kp_type_1 = 50;
	c->kv_type_1 = 14;
	c->kv_type_2 = 5;
	c->enforce_type_1_strategy = 0;
	c->enforce_handling_strategy = 1;
}

int oracle_default_joint_task(sai2b_task_config* c, const char* name, int task_dof,
							  const double* selection) {
	if (!c) return fail("null config");
	memset(c, 0, sizeof(*c));
	c->type = SAI2B_JOINT_TASK;
	snprintf(c->name, sizeof(c->name), "%s", name ? name : "joint_task");
	c->loop_timestep = 0.001;								 /* JointTask.h:58 */
	c->dynamic_decoupling_type = SAI2B_BOUNDED_INERTIA_ESTIMATES; /* JointTask.h:35-37 */
	c->bie_threshold = 0.1;
	if (!selection) {
		c->task_dof = N7; /* JointTask.cpp:17-19 */
		eye(N7, c->joint_selection);
	} else {
		if (task_dof < 1 || task_dof > N7)
			return fail("joint selection matrix size not consistent with robot dof in JointTask constructor");
		/* JointTask.cpp:34-39: FullPivLU rank must equal the number of rows */
		double U[NN], s[N7], V[NN];
		oracle_svd(task_dof, N7, selection, U, s, V);
		for (int i = 0; i < task_dof; i++)
			if (!(s[i] > 1e-12 * s[0]) || s[0] == 0)
				return fail("joint selection matrix is not full rank in JointTask constructor");
		c->task_dof = task_dof;
		memcpy(c->joint_selection, selection, sizeof(double) * task_dof * N7);
	}
	for (int i = 0; i < N7; i++) { /* JointTask.h:32-34,44 */
		c->kp[i] = 50.0;
		c->kv[i] = 14.0;
		c->ki[i] = 0.0;
		c->saturation_velocity[i] = M_PI / 3.0;
		c->otg_max_velocity[i] = M_PI / 3.0;	 /* JointTask.h:40 */
		c->otg_max_acceleration[i] = 2.0 * M_PI; /* JointTask.h:41 */
	}
	c->use_velocity_saturation = 0;
	c->use_internal_otg = 1; /* JointTask.h:38-39 */
	c->internal_otg_jerk_limited = 0;
	for (int i = 0; i < SAI2B_MAX_DOF; i++) c->otg_max_jerk[i] = 10.0 * M_PI; /* JointTask.h:42 */
	c->robot_dof = N7;
	return 0;
}

int oracle_default_motion_force_task(sai2b_task_config* c, const char* name, int link,
									 const double frame_pos[3], const double* frame_rot,
									 int n_trans, const double* dirs_trans, int n_rot,
									 const double* dirs_rot) {
	if (!c) return fail("null config");
	memset(c, 0, sizeof(*c));
	c->type = SAI2B_MOTION_FORCE_TASK;
	int partial = !(n_trans < 0 && n_rot < 0);
	snprintf(c->name, sizeof(c->name), "%s",
			 name ? name : (partial ? "partial_motion_force_task" : "motion_force_task"));
	c->loop_timestep = 0.001;
	if (link < 0 || link >= N7) return fail("MotionForceTask: bad link");
	c->link = link;
	for (int i = 0; i < 3; i++) c->frame_pos[i] = frame_pos ? frame_pos[i] : 0.0;
	if (frame_rot)
		memcpy(c->frame_rot, frame_rot, sizeof(double) * 9);
	else
		eye(3, c->frame_rot);
	if (!partial) {
		eye(6, c->partial_projection); /* MotionForceTask.cpp:28 */
		c->pos_range = c->ori_range = 3;
	} else {
		/* MotionForceTask.cpp:47-87 */
		if (n_trans < 0) n_trans = 0;
		if (n_rot < 0) n_rot = 0;
		if (n_trans == 0 && n_rot == 0)
			return fail(
				"controlled_directions_translation and controlled_directions_rotation cannot both be "
				"empty in MotionForceTask::MotionForceTask");
		for (int blk = 0; blk < 2; blk++) {
			int nd = blk ? n_rot : n_trans;
			const double* d = blk ? dirs_rot : dirs_trans;
			int cols = 0;
			double B[9];
			if (nd > 0) {
				if (nd > 7) return fail("too many controlled directions");
				double A[3 * 7];
				for (int j = 0; j < nd; j++)
					for (int r = 0; r < 3; r++) A[r * nd + j] = d[3 * j + r];
				cols = oracle_range_basis(3, nd, A, 1e-3, B);
			}
			/* P_blk = B B^T; then MotionForceTask.cpp:146-152 re-derives the ranks from P */
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) {
					double s = 0;
					for (int k = 0; k < cols; k++) s += B[i * cols + k] * B[j * cols + k];
					c->partial_projection[(3 * blk + i) * 6 + 3 * blk + j] = s;
				}
			if (blk)
				c->ori_range = cols;
			else
				c->pos_range = cols;
		}
		if (c->pos_range + c->ori_range == 0)
			return fail(
				"controlled_directions_translation and controlled_directions_rotation cannot both be "
				"empty in MotionForceTask::MotionForceTask");
	}
	c->parametrization_in_compliant_frame = 0;
	c->dynamic_decoupling_type = SAI2B_BOUNDED_INERTIA_ESTIMATES; /* MotionForceTask.h:41-43 */
	c->bie_threshold = 0.1;
	for (int i = 0; i < 3; i++) { /* MotionForceTask.h:44-55 */
		c->kp_pos[i] = 100.0;
		c->kv_pos[i] = 20.0;
		c->ki_pos[i] = 0.0;
		c->kp_ori[i] = 200.0;
		c->kv_ori[i] = 28.3;
		c->ki_ori[i] = 0.0;
		c->kp_force[i] = 0.7;
		c->kv_force[i] = 10.0;
		c->ki_force[i] = 1.3;
		c->kp_moment[i] = 0.7;
		c->kv_moment[i] = 10.0;
		c->ki_moment[i] = 1.3;
	}
	c->kff_force = c->kff_moment = 0.95; /* MotionForceTask.h:56-57 */
	c->max_force_feedback = 20.0;		 /* :58-59 */
	c->max_moment_feedback = 10.0;
	c->closed_loop_force = c->closed_loop_moment = 0;
	c->force_space_dimension = c->moment_space_dimension = 0;
	c->force_axis[2] = 1; /* MotionForceTask.h: _force_or_motion_axis default unit z */
	c->moment_axis[2] = 1;
	c->use_velocity_saturation = 0;
	c->linear_saturation_velocity = 0.3;
	c->angular_saturation_velocity = M_PI / 3;
	eye(3, c->sensor_rot); /* MotionForceTask.cpp:94 */
	sh_defaults(c);
	c->use_internal_otg = 1; /* MotionForceTask.h:67-72 */
	c->internal_otg_jerk_limited = 0;
	c->otg_max_linear_jerk = 10.0, c->otg_max_angular_jerk = 10.0 * M_PI; /* MotionForceTask.h:73-74 */
	c->otg_max_linear_velocity = 0.3;
	c->otg_max_linear_acceleration = 2.0;
	c->otg_max_angular_velocity = M_PI / 3;
	c->otg_max_angular_acceleration = 2.0 * M_PI;
	c->robot_dof = N7;
	return 0;
}

/* ------------------------------------------------------------------------------------------
 * per-robot objects
 * ------------------------------------------------------------------------------------------ */
typedef struct {
	double q[N7], dq[N7];
	double Rl[N7][9], pl[N7][3]; /* link frames in world */
	double M[NN], Minv[NN];
	int model_valid;
	const int* jtype; /* the ctx model's joint types (enum sai2b_joint_type) */
} robot_t;

typedef struct {
	/* model */
	double N_prec[NN], Jp[NN], R[NN], M_partial[NN], M_partial_mod[NN], N[NN];
	int k; /* columns of the range basis, 0 = "zero range" */
	/* goals / state */
	double goal_q[N7], goal_dq[N7], goal_ddq[N7], integ[N7];
	double cur_pos[N7];						/* _current_position (JointTask.cpp:299) */
	double des_q[N7], des_dq[N7], des_ddq[N7]; /* _desired_* (JointTask.cpp:308-320) */
	double tau[N7];
} jt_t;

typedef struct {
	double N_prec[NN], J[J6], Jp[J6], N[NN];
	/* goals */
	double g_pos[3], g_rot[9], g_v[3], g_w[3], g_a[3], g_al[3], g_f[3], g_m[3];
	double sens_f[3], sens_m[3];
	double integ_pos[3], integ_ori[3], integ_f[3], integ_m[3];
	/* SingularityHandler members */
	double U[36], s[6], V[J6];
	int ns, sc; /* columns of _task_range_ns / _task_range_s (0 = zero placeholder) */
	double U_ns[36], J_ns[J6], L_ns[36], Jbar_ns[J6], N_ns[NN];
	double U_s[36], V_s[J6], J_s[J6], L_s[36];
	double J_post[J6], L_joint[36];
	double L_ns_mod[36], L_s_mod[36], L_joint_mod[36];
	double alpha;
	int n_types, types[6];
	unsigned char hist[SAI2B_SH_HISTORY + 1];
	int hist_head, hist_size, c1, c2;
	double q_prior[N7], dq_prior[N7], t2dir[N7];
	double tau[N7];
	double Fu[6], Ff[6];
	/* POPCExplicitForceControl members (POPCExplicitForceControl.h:42-50) */
	double popc_po, popc_ecorr, popc_vsum, popc_Rc;
	int popc_counter, popc_head, popc_size;
	double popc_ring[POPC_RING];
	double cur_pos[3], cur_rot[9]; /* _current_position/_orientation (MotionForceTask.cpp:286-289) */
	/* _desired_* (MotionForceTask.cpp:386-407) */
	double des_pos[3], des_rot[9], des_v[3], des_w[3], des_a[3], des_al[3];
} mft_t;

struct oracle_ctx {
	int B, T, threads, gravity_comp;
	sai2b_robot_model model;
	double E[N7][9]; /* constant joint rotations from rpy */
	sai2b_task_config cfg[SAI2B_MAX_TASKS];
	robot_t* robots;
	jt_t* jt[SAI2B_MAX_TASKS];
	mft_t* mft[SAI2B_MAX_TASKS];
	/* internal OTG objects: they exist (and are re-initialised) whether or not the OTG is enabled,
	 * as in the reference (JointTask.cpp:71,106; MotionForceTask.cpp:171,244) */
	otg_joints* jotg[SAI2B_MAX_TASKS];
	otg_cartesian* cotg[SAI2B_MAX_TASKS];
};

/* ---- model (the subset of sai2-model the path calls) ---- */
static void fk(const oracle_ctx* c, const double* q, double Rl[N7][9], double pl[N7][3]) {
	double Rp[9], pp[3] = {0, 0, 0};
	eye(3, Rp);
	for (int i = 0; i < N7; i++) {
		const double* x = c->model.joint_xyz[i];
		for (int r = 0; r < 3; r++) pl[i][r] = pp[r] + Rp[3 * r] * x[0] + Rp[3 * r + 1] * x[1] + Rp[3 * r + 2] * x[2];
		double RE[9], Rz[9] = {cos(q[i]), -sin(q[i]), 0, sin(q[i]), cos(q[i]), 0, 0, 0, 1};
		mm(3, 3, 3, Rp, c->E[i], RE);
		if (c->model.joint_type[i] == SAI2B_PRISMATIC) { /* slides along the joint frame's z by q */
			memcpy(Rl[i], RE, sizeof(RE));
			for (int r = 0; r < 3; r++) pl[i][r] += RE[3 * r + 2] * q[i];
		} else {
			mm(3, 3, 3, RE, Rz, Rl[i]);
		}
		memcpy(Rp, Rl[i], sizeof(Rp));
		memcpy(pp, pl[i], sizeof(pp));
	}
}
/* JWorldFrame(link, pos_in_link): 6 x 7, linear rows first */
static void jacobian(const robot_t* r, int link, const double* pos_in_link, double* J) {
	double p[3];
	for (int k = 0; k < 3; k++)
		p[k] = r->pl[link][k] + r->Rl[link][3 * k] * pos_in_link[0] + r->Rl[link][3 * k + 1] * pos_in_link[1] +
			   r->Rl[link][3 * k + 2] * pos_in_link[2];
	for (int i = 0; i < J6; i++) J[i] = 0;
	for (int i = 0; i <= link; i++) {
		double z[3] = {r->Rl[i][2], r->Rl[i][5], r->Rl[i][8]};
		double d[3] = {p[0] - r->pl[i][0], p[1] - r->pl[i][1], p[2] - r->pl[i][2]};
		double v[3];
		cross3(z, d, v);
		const int prismatic = r->jtype && r->jtype[i] == SAI2B_PRISMATIC; /* column (z, 0) instead of (z x d, z) */
		for (int k = 0; k < 3; k++) {
			J[k * N7 + i] = prismatic ? z[k] : v[k];
			J[(3 + k) * N7 + i] = prismatic ? 0.0 : z[k];
		}
	}
}
static void update_model(const oracle_ctx* c, robot_t* r) {
	fk(c, r->q, r->Rl, r->pl);
	/* M = sum_k m_k Jv_k^T Jv_k + Jw_k^T (R_k I_k R_k^T) Jw_k   (Jacobians at the link COMs) */
	for (int i = 0; i < NN; i++) r->M[i] = 0;
	for (int k = 0; k < N7; k++) {
		double J[J6];
		jacobian(r, k, c->model.link_com[k], J);
		const double* li = c->model.link_inertia[k];
		double Il[9] = {li[0], li[3], li[4], li[3], li[1], li[5], li[4], li[5], li[2]};
		double T[9], Iw[9];
		mm(3, 3, 3, r->Rl[k], Il, T);
		mm_nt(3, 3, 3, T, r->Rl[k], Iw);
		double IJw[J3];
		mm(3, 3, N7, Iw, J + J3, IJw);
		for (int a = 0; a < N7; a++)
			for (int b = 0; b < N7; b++) {
				double s = 0;
				for (int d = 0; d < 3; d++)
					s += c->model.link_mass[k] * J[d * N7 + a] * J[d * N7 + b] + J[(3 + d) * N7 + a] * IJw[d * N7 + b];
				r->M[a * N7 + b] += s;
			}
	}
	oracle_inverse(N7, r->M, r->Minv);
	r->model_valid = 1;
}
static void gravity_vector(const oracle_ctx* c, const robot_t* r, double* g) {
	for (int i = 0; i < N7; i++) g[i] = 0;
	for (int k = 0; k < N7; k++) {
		double J[J6];
		jacobian(r, k, c->model.link_com[k], J);
		for (int i = 0; i < N7; i++)
			for (int d = 0; d < 3; d++) g[i] -= c->model.link_mass[k] * J[d * N7 + i] * c->model.gravity[d];
	}
}
/* ---- simulation harness: rigid-body forward dynamics. The reference's examples get
 * this from the external sai2-simulation (examples/05-...cpp:215-236); nothing of it is in the
 * reference tree, so these are textbook definitions of ours. ----
 * b(q, dq) = C(q, dq) dq (+ g(q)): recursive Newton-Euler with qdd = 0, world-frame quantities. */
static void bias_vector(const oracle_ctx* c, const robot_t* r, int with_gravity, double* b) {
	double w[N7][3], al[N7][3], F[N7][3], Nn[N7][3], cw[N7][3];
	double w_prev[3] = {0, 0, 0}, al_prev[3] = {0, 0, 0}, a_prev[3] = {0, 0, 0}, o_prev[3] = {0, 0, 0};
	if (with_gravity)
		for (int k = 0; k < 3; k++) a_prev[k] = -c->model.gravity[k];
	for (int i = 0; i < N7; i++) {
		const double* R = r->Rl[i];
		const double z[3] = {R[2], R[5], R[8]};
		double rr[3], t1[3], t2[3], a_i[3], zq[3];
		for (int k = 0; k < 3; k++) rr[k] = r->pl[i][k] - o_prev[k];
		cross3(al_prev, rr, t1);
		cross3(w_prev, rr, t2);
		double t3[3];
		cross3(w_prev, t2, t3);
		for (int k = 0; k < 3; k++) a_i[k] = a_prev[k] + t1[k] + t3[k];
		for (int k = 0; k < 3; k++) zq[k] = z[k] * r->dq[i];
		cross3(w_prev, zq, t1);
		const int prismatic = r->jtype && r->jtype[i] == SAI2B_PRISMATIC;
		for (int k = 0; k < 3; k++) {
			if (prismatic) { /* sliding frame: Coriolis acceleration 2 w x (z dq) of its origin, angular motion unchanged */
				w[i][k] = w_prev[k];
				al[i][k] = al_prev[k];
				a_i[k] += 2 * t1[k];
			} else {
				w[i][k] = w_prev[k] + zq[k];
				al[i][k] = al_prev[k] + t1[k];
			}
		}
		double rc[3];
		for (int k = 0; k < 3; k++) {
			rc[k] = R[3 * k] * c->model.link_com[i][0] + R[3 * k + 1] * c->model.link_com[i][1] +
					R[3 * k + 2] * c->model.link_com[i][2];
			cw[i][k] = r->pl[i][k] + rc[k];
		}
		double a_c[3];
		cross3(al[i], rc, t1);
		cross3(w[i], rc, t2);
		cross3(w[i], t2, t3);
		for (int k = 0; k < 3; k++) a_c[k] = a_i[k] + t1[k] + t3[k];
		const double* li = c->model.link_inertia[i];
		double Il[9] = {li[0], li[3], li[4], li[3], li[1], li[5], li[4], li[5], li[2]}, T[9], Iw[9];
		mm(3, 3, 3, R, Il, T);
		mm_nt(3, 3, 3, T, R, Iw);
		double Ial[3], Iom[3];
		mm(3, 3, 1, Iw, al[i], Ial);
		mm(3, 3, 1, Iw, w[i], Iom);
		cross3(w[i], Iom, t1);
		for (int k = 0; k < 3; k++) {
			F[i][k] = c->model.link_mass[i] * a_c[k];
			Nn[i][k] = Ial[k] + t1[k];
			w_prev[k] = w[i][k];
			al_prev[k] = al[i][k];
			a_prev[k] = a_i[k];
			o_prev[k] = r->pl[i][k];
		}
	}
	double f_next[3] = {0, 0, 0}, n_next[3] = {0, 0, 0};
	for (int i = N7 - 1; i >= 0; i--) {
		double f[3], n[3], d[3], t1[3];
		for (int k = 0; k < 3; k++) d[k] = cw[i][k] - r->pl[i][k];
		cross3(d, F[i], t1);
		for (int k = 0; k < 3; k++) {
			f[k] = F[i][k] + f_next[k];
			n[k] = Nn[i][k] + n_next[k] + t1[k];
		}
		if (i + 1 < N7) {
			for (int k = 0; k < 3; k++) d[k] = r->pl[i + 1][k] - r->pl[i][k];
			cross3(d, f_next, t1);
			for (int k = 0; k < 3; k++) n[k] += t1[k];
		}
		const double* pr = (r->jtype && r->jtype[i] == SAI2B_PRISMATIC) ? f : n; /* prismatic: the force along the axis */
		b[i] = r->Rl[i][2] * pr[0] + r->Rl[i][5] * pr[1] + r->Rl[i][8] * pr[2];
		for (int k = 0; k < 3; k++) {
			f_next[k] = f[k];
			n_next[k] = n[k];
		}
	}
}

static void frame_pose(const sai2b_task_config* t, const double Rl[N7][9], const double pl[N7][3], double* x,
					   double* R) {
	for (int k = 0; k < 3; k++)
		x[k] = pl[t->link][k] + Rl[t->link][3 * k] * t->frame_pos[0] + Rl[t->link][3 * k + 1] * t->frame_pos[1] +
			   Rl[t->link][3 * k + 2] * t->frame_pos[2];
	mm(3, 3, 3, Rl[t->link], t->frame_rot, R);
}

/* The pose a MotionForceTask's goals and internal OTG START from (re-initialisation, OTG enable, force / motion
 * space re-parametrisation): the bit-reproducible forward kinematics shared with the product's initialisation
 * kernels (include/sai2b_detfk.h), so that both sides' generators see identical bits and take the same planner
 * branches. Everything on the torque path uses fk() / frame_pose() above, this oracle's own. */
static void det_pose(const oracle_ctx* c, const sai2b_task_config* t, const double* q, double* x, double* R) {
	sai2b_det_frame_pose(&c->E[0][0], &c->model.joint_xyz[0][0], c->model.joint_type, q, t->link, t->frame_pos, t->frame_rot, x, R);
}

/* Sai2Model::operationalSpaceMatrices(J) as DEFINED in SURVEY App. D:
 * Lambda = (J M^-1 J^T)^-1, Jbar = M^-1 J^T Lambda, N = I - Jbar J */
static void opspace(const robot_t* r, int m, const double* J, double* L, double* Jbar, double* N) {
	double A[NN], T[NN], JT[NN];
	mm(m, N7, N7, J, r->Minv, T);
	mm_nt(m, N7, m, T, J, A);
	oracle_inverse(m, A, L);
	mm_tn(N7, m, m, T, L, JT); /* (J Minv)^T L = Minv J^T L */
	if (Jbar) memcpy(Jbar, JT, sizeof(double) * N7 * m);
	mm(N7, m, N7, JT, J, T);
	eye(N7, N);
	for (int i = 0; i < NN; i++) N[i] -= T[i];
}
static void bie_minv(const robot_t* r, double thr, double* MinvB) {
	/* SingularityHandler.cpp:176-182, JointTask.cpp:254-260 */
	double MB[NN];
	memcpy(MB, r->M, sizeof(MB));
	for (int i = 0; i < N7; i++)
		if (MB[i * N7 + i] < thr) MB[i * N7 + i] = thr;
	oracle_inverse(N7, MB, MinvB);
}
/* (J Minv J^T)^-1 for an m x 7 J */
static void lambda_of(int m, const double* J, const double* Minv, double* L) {
	double T[NN], A[NN];
	mm(m, N7, N7, J, Minv, T);
	mm_nt(m, N7, m, T, J, A);
	oracle_inverse(m, A, L);
}

/* ---- JointTask ---- */
static void jt_reinit(const sai2b_task_config* t, const robot_t* r, jt_t* s, otg_joints* o) {
	/* JointTask.cpp:91-107 */
	mm(t->task_dof, N7, 1, t->joint_selection, r->q, s->goal_q);
	for (int i = 0; i < N7; i++) s->goal_dq[i] = s->goal_ddq[i] = s->integ[i] = 0;
	for (int i = 0; i < t->task_dof; i++) {
		s->cur_pos[i] = s->des_q[i] = s->goal_q[i];
		s->des_dq[i] = s->des_ddq[i] = 0;
	}
	otg_joints_reinitialize(o, s->cur_pos);
}
/* JointTask::enableInternalOtgAccelerationLimited / enableInternalOtgJerkLimited (JointTask.cpp:360-406): the
 * generator restarts at the task's current position when it was off or when the kind of limitation changes */
static void jt_otg_enable(const sai2b_task_config* t, const jt_t* s, otg_joints* o, int was_enabled, int was_jerk) {
	const int jerk = t->internal_otg_jerk_limited != 0;
	if (!was_enabled || was_jerk != jerk) otg_joints_reinitialize(o, s->cur_pos);
	otg_joints_set_limits(o, t->otg_max_velocity, t->otg_max_acceleration);
	if (jerk)
		otg_joints_set_max_jerk(o, t->otg_max_jerk);
	else
		otg_joints_disable_jerk_limits(o);
}
/* JointTask::initialSetup, OTG part + reInitializeTask (JointTask.cpp:50,70-89) */
static void jt_construct(const sai2b_task_config* t, const robot_t* r, jt_t* s, otg_joints* o) {
	mm(t->task_dof, N7, 1, t->joint_selection, r->q, s->cur_pos);
	otg_joints_init(o, t->task_dof, s->cur_pos, t->loop_timestep);
	if (t->use_internal_otg) jt_otg_enable(t, s, o, 0, 0);
	jt_reinit(t, r, s, o);
}
static void jt_update(const sai2b_task_config* t, const robot_t* r, jt_t* s, const double* N_prec) {
	/* JointTask.cpp:218-283 */
	int k0 = t->task_dof;
	memcpy(s->N_prec, N_prec, sizeof(double) * NN);
	mm(k0, N7, N7, t->joint_selection, N_prec, s->Jp);
	s->k = oracle_range_basis(k0, N7, s->Jp, 1e-3, s->R);
	if (s->k == 0) { /* :234-239 */
		eye(N7, s->N);
		return;
	}
	double Jr[NN];
	mm_tn(s->k, k0, N7, s->R, s->Jp, Jr);
	opspace(r, s->k, Jr, s->M_partial, NULL, s->N);
	switch (t->dynamic_decoupling_type) {
		case SAI2B_FULL_DYNAMIC_DECOUPLING:
			memcpy(s->M_partial_mod, s->M_partial, sizeof(double) * s->k * s->k);
			break;
		case SAI2B_BOUNDED_INERTIA_ESTIMATES: {
			double MinvB[NN];
			bie_minv(r, t->bie_threshold, MinvB);
			lambda_of(s->k, Jr, MinvB, s->M_partial_mod);
			break;
		}
		default:
			eye(s->k, s->M_partial_mod);
	}
}
static void jt_torques(const sai2b_task_config* t, const robot_t* r, jt_t* s, otg_joints* o, double* tau) {
	/* JointTask.cpp:294-356 */
	int k0 = t->task_dof, k = s->k;
	double cur[N7], vel[N7], f[N7];
	for (int i = 0; i < N7; i++) tau[i] = 0;
	mm(k0, N7, N7, t->joint_selection, s->N_prec, s->Jp);
	mm(k0, N7, 1, t->joint_selection, r->q, cur);
	mm(k0, N7, 1, t->joint_selection, r->dq, vel);
	memcpy(s->cur_pos, cur, sizeof(double) * k0);
	if (k == 0) return;
	for (int i = 0; i < k0; i++) {
		s->des_q[i] = s->goal_q[i];
		s->des_dq[i] = s->goal_dq[i];
		s->des_ddq[i] = s->goal_ddq[i];
	}
	if (t->use_internal_otg) { /* :313-320 */
		otg_joints_set_goal(o, s->goal_q, s->goal_dq);
		otg_joints_update(o);
		for (int i = 0; i < k0; i++) {
			s->des_q[i] = o->output.np[i];
			s->des_dq[i] = o->output.nv[i];
			s->des_ddq[i] = o->output.na[i];
		}
	}
	const double *des_q = s->des_q, *des_ddq = s->des_ddq;
	double des_dq[N7];
	for (int i = 0; i < k0; i++) des_dq[i] = s->des_dq[i];
	for (int i = 0; i < k0; i++) s->integ[i] += (cur[i] - des_q[i]) * t->loop_timestep;
	if (t->use_velocity_saturation) { /* :327-340 (loop bound fixed to task dof: SURVEY App. B-7) */
		for (int i = 0; i < k0; i++) {
			double kvi = gain_pinv(t->kv[i]);
			des_dq[i] = -t->kp[i] * kvi * (cur[i] - des_q[i]) - t->ki[i] * kvi * s->integ[i];
			if (des_dq[i] > t->saturation_velocity[i]) des_dq[i] = t->saturation_velocity[i];
			if (des_dq[i] < -t->saturation_velocity[i]) des_dq[i] = -t->saturation_velocity[i];
			f[i] = -t->kv[i] * (vel[i] - des_dq[i]);
		}
	} else {
		for (int i = 0; i < k0; i++)
			f[i] = -t->kp[i] * (cur[i] - des_q[i]) - t->kv[i] * (vel[i] - des_dq[i]) - t->ki[i] * s->integ[i];
	}
	double ra[N7], rf[N7], x1[N7], x2[N7], y[N7];
	mm_tn(k, k0, 1, s->R, des_ddq, ra);
	mm_tn(k, k0, 1, s->R, f, rf);
	mm(k, k, 1, s->M_partial, ra, x1);
	mm(k, k, 1, s->M_partial_mod, rf, x2);
	for (int i = 0; i < k; i++) x1[i] += x2[i];
	mm(k0, k, 1, s->R, x1, y);
	mm_tn(N7, k0, 1, s->Jp, y, tau);
}
static void jt_compensation(const sai2b_task_config* t, const robot_t* r, const jt_t* s, const double* tau_prec,
							double* comp) {
	/* JointTask.cpp:285-292: Jp^T R M_partial R^T S M^-1 tau_prec */
	int k0 = t->task_dof, k = s->k;
	for (int i = 0; i < N7; i++) comp[i] = 0;
	if (k == 0) return;
	double a[N7], b[N7], c[N7], d[N7], e[N7];
	mm(N7, N7, 1, r->Minv, tau_prec, a);
	mm(k0, N7, 1, t->joint_selection, a, b);
	mm_tn(k, k0, 1, s->R, b, c);
	mm(k, k, 1, s->M_partial, c, d);
	mm(k0, k, 1, s->R, d, e);
	mm_tn(N7, k0, 1, s->Jp, e, comp);
}

/* ---- MotionForceTask + SingularityHandler ---- */
static void mft_reinit(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s, otg_cartesian* o) {
	/* MotionForceTask.cpp:204-245; SingularityHandler.cpp:53-63 */
	det_pose(c, t, r->q, s->g_pos, s->g_rot);
	for (int i = 0; i < 3; i++) {
		s->g_v[i] = s->g_w[i] = s->g_a[i] = s->g_al[i] = s->g_f[i] = s->g_m[i] = 0;
		s->sens_f[i] = s->sens_m[i] = 0;
		s->integ_pos[i] = s->integ_ori[i] = s->integ_f[i] = s->integ_m[i] = 0;
		s->des_v[i] = s->des_w[i] = s->des_a[i] = s->des_al[i] = 0;
	}
	memcpy(s->cur_pos, s->g_pos, sizeof(s->cur_pos));
	memcpy(s->cur_rot, s->g_rot, sizeof(s->cur_rot));
	memcpy(s->des_pos, s->g_pos, sizeof(s->des_pos));
	memcpy(s->des_rot, s->g_rot, sizeof(s->des_rot));
	otg_cartesian_reinitialize(o, s->cur_pos, s->cur_rot);
}
/* MotionForceTask::enableInternalOtgAccelerationLimited / enableInternalOtgJerkLimited (MotionForceTask.cpp:511-538) */
static void mft_otg_enable(const sai2b_task_config* t, const mft_t* s, otg_cartesian* o, int was_enabled, int was_jerk) {
	const int jerk = t->internal_otg_jerk_limited != 0;
	if (!was_enabled || was_jerk != jerk) otg_cartesian_reinitialize(o, s->cur_pos, s->cur_rot);
	otg_cartesian_set_limits(o, t->otg_max_linear_velocity, t->otg_max_linear_acceleration,
							 t->otg_max_angular_velocity, t->otg_max_angular_acceleration);
	otg_cartesian_set_max_jerk(o, jerk ? t->otg_max_linear_jerk : INFINITY, jerk ? t->otg_max_angular_jerk : INFINITY);
}
/* MotionForceTask::initialSetup, OTG part + reInitializeTask (MotionForceTask.cpp:100-103,170-201) */
static void mft_construct(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s, otg_cartesian* o) {
	det_pose(c, t, r->q, s->cur_pos, s->cur_rot);
	otg_cartesian_init(o, s->cur_pos, s->cur_rot, t->loop_timestep);
	if (t->use_internal_otg) mft_otg_enable(t, s, o, 0, 0);
	mft_reinit(c, t, r, s, o);
}
static void popc_init(mft_t* s) { /* POPCExplicitForceControl.cpp:10-22 */
	s->popc_po = s->popc_ecorr = s->popc_vsum = 0;
	s->popc_Rc = 1.0;
	s->popc_counter = 50;
	s->popc_head = s->popc_size = 0;
}
/* POPCExplicitForceControl::computePassivitySaturatedForce, observer enabled (POPCExplicitForceControl.cpp:37-95) */
static void popc_force(const sai2b_task_config* t, mft_t* s, const double* fd, const double* fs, const double* vcl,
					   const double* vr, double* out) {
	double fcmd[3], vc2 = 0, p = 0;
	for (int k = 0; k < 3; k++) {
		fcmd[k] = t->kff_force * fd[k] + s->popc_Rc * vcl[k] - t->kv_force[k] * vr[k];
		vc2 += vcl[k] * vcl[k];
		p += (fs[k] - fd[k]) * vcl[k] - fcmd[k] * vr[k];
	}
	p *= t->loop_timestep;
	s->popc_po += p;
	if (s->popc_size == POPC_RING) { /* bounded stand-in for the unbounded std::queue */
		double front = s->popc_ring[s->popc_head];
		if (front > 0) s->popc_po -= front;
		s->popc_head = (s->popc_head + 1) % POPC_RING;
		s->popc_size--;
	}
	s->popc_ring[(s->popc_head + s->popc_size) % POPC_RING] = p;
	s->popc_size++;
	if (s->popc_po + s->popc_ecorr > 0) {
		while (s->popc_size > 250) {
			double front = s->popc_ring[s->popc_head];
			if (s->popc_po + s->popc_ecorr > front) {
				if (front > 0) s->popc_po -= front;
				s->popc_head = (s->popc_head + 1) % POPC_RING;
				s->popc_size--;
			} else
				break;
		}
	}
	if (s->popc_counter <= 0) {
		s->popc_counter = 50;
		double old = s->popc_Rc;
		if (s->popc_po + s->popc_ecorr < 0) {
			s->popc_Rc = 1 + (s->popc_po + s->popc_ecorr) / (s->popc_vsum * t->loop_timestep);
			if (s->popc_Rc > 1) s->popc_Rc = 1;
			if (s->popc_Rc < 0) s->popc_Rc = 0;
		} else {
			s->popc_Rc = (1 + (0.1 * 50 - 1) * s->popc_Rc) / (0.1 * 50);
		}
		s->popc_ecorr += (1 - old) * s->popc_vsum * t->loop_timestep;
		s->popc_vsum = 0;
	}
	s->popc_counter--;
	s->popc_vsum += vc2;
	for (int k = 0; k < 3; k++) out[k] = s->popc_Rc * vcl[k] - t->kv_force[k] * vr[k];
}
static void sh_init(const oracle_ctx* c, mft_t* s) {
	s->n_types = 0;
	s->hist_head = s->hist_size = s->c1 = s->c2 = 0;
	for (int i = 0; i < N7; i++) {
		s->q_prior[i] = 0.5 * (c->model.q_lower[i] + c->model.q_upper[i]);
		s->dq_prior[i] = 0;
		s->t2dir[i] = 1;
	}
}
static void sh_classify(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s) {
	/* SingularityHandler.cpp:230-295 */
	if (s->n_types == 0 || s->c2 > s->c1) {
		memcpy(s->q_prior, r->q, sizeof(double) * N7);
		memcpy(s->dq_prior, r->dq, sizeof(double) * N7);
	}
	if (s->sc == 0) {
		s->n_types = 0;
		s->hist_head = s->hist_size = s->c1 = s->c2 = 0;
		return;
	}
	s->n_types = s->sc;
	double x0[3], R0[9];
	frame_pose(t, r->Rl, r->pl, x0, R0);
	int any1 = 0;
	for (int i = 0; i < s->sc; i++) {
		/* The perturbation direction is V_s[:, i] as Eigen's JacobiSVD leaves it in the reference (:78-81, :254); its
		 * sign is not specified there, so it is a setting here (enum sai2b_singular_vector_sign; oracle_svd orients
		 * every right singular vector "largest-magnitude component positive"): that orientation, the opposite one, or
		 * the two sign-free rules "either" / "both". */
		int moved[2] = {0, 0};
		for (int pass = 0; pass < 2; pass++) {
			const double step = pass ? -t->perturb_step_size : t->perturb_step_size;
			double qp[N7], Rl[N7][9], pl[N7][3], x1[3], R1[9], d[6];
			for (int j = 0; j < N7; j++) qp[j] = r->q[j] + step * s->V_s[j * s->sc + i];
			fk(c, qp, Rl, pl);
			frame_pose(t, Rl, pl, x1, R1);
			for (int k = 0; k < 3; k++) d[k] = x1[k] - x0[k];
			orientation_error(R1, R0, d + 3);
			double m = 0;
			for (int k = 0; k < 6; k++) m += d[k] * s->U_s[k * s->sc + i];
			moved[pass] = fabs(m) > t->type_1_tol;
		}
		int type1;
		switch (t->singular_vector_sign) {
			case SAI2B_SV_SIGN_V_MAX_NEGATIVE: type1 = moved[1]; break;
			case SAI2B_SV_SIGN_EITHER: type1 = moved[0] || moved[1]; break;
			case SAI2B_SV_SIGN_BOTH: type1 = moved[0] && moved[1]; break;
			default: type1 = moved[0];
		}
		s->types[i] = type1 ? 1 : 2;
		if (type1) any1 = 1;
	}
	int cap = t->sh_buffer_size;
	if (cap > SAI2B_SH_HISTORY) cap = SAI2B_SH_HISTORY;
	int ring = SAI2B_SH_HISTORY + 1;
	s->hist[(s->hist_head + s->hist_size) % ring] = any1 ? 1 : 2;
	s->hist_size++;
	if (any1)
		s->c1++;
	else
		s->c2++;
	if (s->hist_size > cap) {
		if (s->hist[s->hist_head] == 1)
			s->c1--;
		else
			s->c2--;
		s->hist_head = (s->hist_head + 1) % ring;
		s->hist_size--;
	}
}
static void sh_update(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s,
					  const double* N_prec) {
	/* SingularityHandler.cpp:75-228 */
	int rank = t->pos_range + t->ori_range;
	const int SVP = N7 < 6 ? N7 : 6; /* the thin SVD of the 6 x n Jacobian has min(6, n) columns */
	for (int i = 0; i < 6; i++) s->s[i] = 0;
	oracle_svd(6, N7, s->Jp, s->U, s->s, s->V);
	if (rank > SVP) rank = SVP; /* a task asking for more directions than the robot has joints: the missing ones are singular by construction */
	s->ns = s->sc = 0;
	int split = -1; /* number of non-singular columns */
	if (s->s[0] < t->s_abs_tol) { /* :83-98 fully singular */
		s->alpha = 0;
		split = 0;
	} else {
		split = rank; /* also covers rank == 1 */
		s->alpha = 1;
		for (int i = 1; i < rank; i++) {
			double icn = s->s[i] / s->s[0];
			if (icn < t->s_max) { /* :103-121 */
				double a = (icn - t->s_min) / (t->s_max - t->s_min);
				s->alpha = a < 0 ? 0 : (a > 1 ? 1 : a);
				split = i;
				break;
			}
		}
	}
	s->ns = split;
	s->sc = rank - split;
	if (s->ns > 0) {
		for (int rr = 0; rr < 6; rr++)
			for (int cc = 0; cc < s->ns; cc++) s->U_ns[rr * s->ns + cc] = s->U[rr * SVP + cc];
		mm_tn(s->ns, 6, N7, s->U_ns, s->Jp, s->J_ns);
		opspace(r, s->ns, s->J_ns, s->L_ns, s->Jbar_ns, s->N_ns);
	}
	if (s->sc > 0) {
		for (int rr = 0; rr < 6; rr++)
			for (int cc = 0; cc < s->sc; cc++) s->U_s[rr * s->sc + cc] = s->U[rr * SVP + split + cc];
		for (int rr = 0; rr < N7; rr++)
			for (int cc = 0; cc < s->sc; cc++) s->V_s[rr * s->sc + cc] = s->V[rr * SVP + split + cc];
		mm_tn(s->sc, 6, N7, s->U_s, s->Jp, s->J_s);
		double T[NN], A[NN];
		mm(s->sc, N7, N7, s->J_s, r->Minv, T);
		mm_nt(s->sc, N7, s->sc, T, s->J_s, A);
		if (s->ns == 0)
			sym_pinv(s->sc, A, s->L_s); /* :96-98 */
		else
			oracle_inverse(s->sc, A, s->L_s); /* :120 */
	}
	/* :146-158 nullspace selection. Cases the reference leaves to stale members are defined
	 * explicitly: fully singular -> N = N_prec. */
	int have_post = 0;
	if (s->ns == 0) {
		memcpy(s->N, N_prec, sizeof(double) * NN);
	} else if (s->sc == 0 || !t->enforce_handling_strategy) {
		memcpy(s->N, s->N_ns, sizeof(double) * NN);
	} else {
		double T[NN], Np[NN];
		mm(N7, N7, N7, s->N_ns, N_prec, T);
		mm_tn(s->sc, N7, N7, s->V_s, T, s->J_post);
		opspace(r, s->sc, s->J_post, s->L_joint, NULL, Np);
		mm(N7, N7, N7, Np, s->N_ns, s->N);
		have_post = 1;
	}
	/* :160-225 dynamic decoupling */
	switch (t->dynamic_decoupling_type) {
		case SAI2B_IMPEDANCE:
			if (s->ns) eye(s->ns, s->L_ns_mod);
			if (s->sc) eye(s->sc, s->L_s_mod);
			if (have_post) eye(s->sc, s->L_joint_mod);
			break;
		case SAI2B_BOUNDED_INERTIA_ESTIMATES: {
			double MinvB[NN];
			bie_minv(r, t->bie_threshold, MinvB);
			if (s->ns) lambda_of(s->ns, s->J_ns, MinvB, s->L_ns_mod);
			if (s->sc) lambda_of(s->sc, s->J_s, MinvB, s->L_s_mod);
			if (have_post) lambda_of(s->sc, s->J_post, MinvB, s->L_joint_mod);
			break;
		}
		default:
			if (s->ns) memcpy(s->L_ns_mod, s->L_ns, sizeof(double) * s->ns * s->ns);
			if (s->sc) memcpy(s->L_s_mod, s->L_s, sizeof(double) * s->sc * s->sc);
			if (have_post) memcpy(s->L_joint_mod, s->L_joint, sizeof(double) * s->sc * s->sc);
	}
	sh_classify(c, t, r, s);
}
static void mft_update(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s,
					   const double* N_prec) {
	/* MotionForceTask.cpp:247-268 */
	double Jw[J6];
	memcpy(s->N_prec, N_prec, sizeof(double) * NN);
	jacobian(r, t->link, t->frame_pos, Jw);
	mm(6, 6, N7, t->partial_projection, Jw, s->J);
	mm(6, N7, N7, s->J, N_prec, s->Jp);
	sh_update(c, t, r, s, N_prec);
}
/* tau = J_x^T ( L_x_mod U_x^T Fu + U_x^T Ff ) for x in {ns, s} */
static void range_torque(int cols, const double* U, const double* J, const double* Lmod, const double* Fu,
						 const double* Ff, double* tau) {
	double a[6], b[6], c[6];
	mm_tn(cols, 6, 1, U, Fu, a);
	mm_tn(cols, 6, 1, U, Ff, b);
	mm(cols, cols, 1, Lmod, a, c);
	for (int i = 0; i < cols; i++) c[i] += b[i];
	mm_tn(N7, cols, 1, J, c, tau);
}
static void sh_torques(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s,
					   const double* Fu, const double* Ff, double* tau) {
	/* SingularityHandler.cpp:297-368 */
	for (int i = 0; i < N7; i++) tau[i] = 0;
	if (s->n_types == 0) { /* :307-309 */
		if (s->ns) range_torque(s->ns, s->U_ns, s->J_ns, s->L_ns_mod, Fu, Ff, tau);
		return;
	}
	if (t->dynamic_decoupling_type == SAI2B_IMPEDANCE) { /* :310-312 */
		if (s->ns) {
			double I6[36];
			eye(s->ns, I6);
			range_torque(s->ns, s->U_ns, s->J_ns, I6, Fu, Ff, tau);
		}
		return;
	}
	if (s->ns == 0) return; /* :317-318 */
	double tau_ns[N7], tau_s[N7], tau_j[N7];
	range_torque(s->ns, s->U_ns, s->J_ns, s->L_ns_mod, Fu, Ff, tau_ns);
	if (!t->enforce_handling_strategy) { /* :322-324 */
		memcpy(tau, tau_ns, sizeof(tau_ns));
		return;
	}
	int sc = s->sc;
	double ut[N7], a[6], b[6];
	if (s->c1 > s->c2 || t->enforce_type_1_strategy) { /* :327-331 */
		for (int i = 0; i < N7; i++) ut[i] = -t->kp_type_1 * (r->q[i] - s->q_prior[i]) - t->kv_type_1 * r->dq[i];
		mm_tn(sc, N7, 1, s->V_s, ut, a);
		mm(sc, sc, 1, s->L_joint_mod, a, b);
		mm_tn(N7, sc, 1, s->J_post, b, tau_j);
	} else { /* :332-351 */
		for (int i = 0; i < N7; i++) {
			if (s->V_s[i * sc + 0] != 0) {
				if (fabs(r->q[i] - c->model.q_upper[i]) < t->type_2_angle_threshold)
					s->t2dir[i] = -1;
				else if (fabs(r->q[i] - c->model.q_lower[i]) < t->type_2_angle_threshold)
					s->t2dir[i] = 1;
			}
		}
		double F[6], nrm = 0, fTd = 0;
		for (int i = 0; i < 6; i++) {
			F[i] = Fu[i] + Ff[i];
			nrm += F[i] * F[i];
		}
		nrm = sqrt(nrm);
		for (int i = 0; i < 6; i++) fTd += (nrm > 0 ? F[i] / nrm : F[i]) * s->U_s[i * sc + 0];
		/* _type_2_torque_vector = ratio * effort (intent of SingularityHandler.cpp:48; App. B-4) */
		for (int i = 0; i < N7; i++) ut[i] = s->t2dir[i] * fabs(fTd) * t->type_2_torque_ratio * c->model.effort[i];
		double t1[N7], t2[N7];
		mm_tn(sc, N7, 1, s->V_s, ut, a);
		mm_tn(N7, sc, 1, s->J_post, a, t1);
		for (int i = 0; i < N7; i++) ut[i] = -t->kv_type_2 * r->dq[i];
		mm_tn(sc, N7, 1, s->V_s, ut, a);
		mm(sc, sc, 1, s->L_joint_mod, a, b);
		mm_tn(N7, sc, 1, s->J_post, b, t2);
		for (int i = 0; i < N7; i++) tau_j[i] = t1[i] + t2[i];
	}
	range_torque(sc, s->U_s, s->J_s, s->L_s_mod, Fu, Ff, tau_s); /* :354-355 */
	for (int i = 0; i < N7; i++) {								 /* :357-365 */
		if (isnan(tau_s[i]))
			tau_s[i] = 0;
		else if (tau_s[i] > c->model.effort[i])
			tau_s[i] = c->model.effort[i];
		else if (tau_s[i] < -c->model.effort[i])
			tau_s[i] = -c->model.effort[i];
	}
	for (int i = 0; i < N7; i++) tau[i] = tau_ns[i] + s->alpha * tau_s[i] + (1 - s->alpha) * tau_j[i];
}
/* sigma matrices, MotionForceTask.cpp:892-971 */
static void sigma_pair(const double* P6, int blk, int dim, const double* axis, const double* Rw, int in_frame,
					   double* sig_f, double* sig_p) {
	double P[9], A[9], T[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) P[3 * i + j] = P6[(3 * blk + i) * 6 + 3 * blk + j];
	double a[3];
	for (int i = 0; i < 3; i++)
		a[i] = in_frame ? Rw[3 * i] * axis[0] + Rw[3 * i + 1] * axis[1] + Rw[3 * i + 2] * axis[2] : axis[i];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double aa = a[i] * a[j];
			A[3 * i + j] = dim == 0 ? 0.0 : dim == 1 ? aa : dim == 2 ? ((i == j) - aa) : (double)(i == j);
		}
	if (dim == 0) {
		for (int i = 0; i < 9; i++) sig_f[i] = 0;
	} else if (dim == 3) {
		memcpy(sig_f, P, sizeof(P));
	} else {
		mm(3, 3, 3, P, A, T);
		mm_nt(3, 3, 3, T, P, sig_f);
	}
	for (int i = 0; i < 9; i++) A[i] = ((i % 4) == 0) - sig_f[i];
	mm(3, 3, 3, P, A, T);
	mm_nt(3, 3, 3, T, P, sig_p);
}
static void mv3(const double* A, const double* x, double* y) {
	for (int i = 0; i < 3; i++) y[i] = A[3 * i] * x[0] + A[3 * i + 1] * x[1] + A[3 * i + 2] * x[2];
}
static void mft_torques(const oracle_ctx* c, const sai2b_task_config* t, const robot_t* r, mft_t* s,
						otg_cartesian* o, double* tau) {
	/* MotionForceTask.cpp:278-509 */
	double Jw[J6], x[3], R[9], v[3], w[3];
	for (int i = 0; i < N7; i++) tau[i] = 0;
	jacobian(r, t->link, t->frame_pos, Jw);
	mm(6, 6, N7, t->partial_projection, Jw, s->J);
	mm(6, N7, N7, s->J, s->N_prec, s->Jp);
	frame_pose(t, r->Rl, r->pl, x, R);
	mm(3, N7, 1, s->J, r->dq, v);
	mm(3, N7, 1, s->J + J3, r->dq, w);
	det_pose(c, t, r->q, s->cur_pos, s->cur_rot); /* the cached pose later (re)initialisations start from */
	if (t->pos_range + t->ori_range == 0) return;
	double sf[9], sp[9], sm[9], so[9];
	sigma_pair(t->partial_projection, 0, t->force_space_dimension, t->force_axis, R,
			   t->parametrization_in_compliant_frame, sf, sp);
	sigma_pair(t->partial_projection, 1, t->moment_space_dimension, t->moment_axis, R,
			   t->parametrization_in_compliant_frame, sm, so);
	double gf[3], gm[3]; /* getGoalForce/Moment :755-769 */
	if (t->parametrization_in_compliant_frame) {
		mv3(R, s->g_f, gf);
		mv3(R, s->g_m, gm);
	} else {
		memcpy(gf, s->g_f, sizeof(gf));
		memcpy(gm, s->g_m, sizeof(gm));
	}
	/* sensed wrench resolved to the world frame with the current pose (:805-828) */
	double fs_c[3], ms_c[3], tmp[3], fs_w[3], ms_w[3];
	mv3(t->sensor_rot, s->sens_f, fs_c);
	mv3(t->sensor_rot, s->sens_m, ms_c);
	cross3(t->sensor_pos, fs_c, tmp);
	for (int i = 0; i < 3; i++) ms_c[i] += tmp[i];
	mv3(R, fs_c, fs_w);
	mv3(R, ms_c, ms_w);

	double f_force[3], f_moment[3], f_pos[3], f_ori[3], e[3], y[3];
	const double dt = t->loop_timestep;
	/* force (:327-354) */
	if (t->closed_loop_force) {
		for (int i = 0; i < 3; i++) e[i] = fs_w[i] - gf[i];
		mv3(sf, e, y);
		for (int i = 0; i < 3; i++) s->integ_f[i] += y[i] * dt;
		for (int i = 0; i < 3; i++) e[i] = -t->kp_force[i] * (fs_w[i] - gf[i]) - t->ki_force[i] * s->integ_f[i];
		double fb[3];
		mv3(sf, e, fb);
		double n = sqrt(fb[0] * fb[0] + fb[1] * fb[1] + fb[2] * fb[2]);
		if (n > t->max_force_feedback)
			for (int i = 0; i < 3; i++) fb[i] *= t->max_force_feedback / n;
		/* POPC disabled: vcl - kv_force vr (POPCExplicitForceControl.cpp:33-35) */
		double vcl[3], vr[3];
		mv3(sf, fb, vcl);
		mv3(sf, v, vr);
		if (t->passivity_enabled) {
			double fd[3], fsf[3];
			mv3(sf, gf, fd);
			mv3(sf, fs_w, fsf);
			popc_force(t, s, fd, fsf, vcl, vr, f_force);
		} else {
			for (int i = 0; i < 3; i++) f_force[i] = vcl[i] - t->kv_force[i] * vr[i];
		}
	} else {
		for (int i = 0; i < 3; i++) e[i] = -t->kv_force[i] * v[i];
		mv3(sf, e, f_force);
	}
	/* moment (:356-383) */
	if (t->closed_loop_moment) {
		for (int i = 0; i < 3; i++) e[i] = ms_w[i] - gm[i];
		mv3(sm, e, y);
		for (int i = 0; i < 3; i++) s->integ_m[i] += y[i] * dt;
		for (int i = 0; i < 3; i++) e[i] = -t->kp_moment[i] * (ms_w[i] - gm[i]) - t->ki_moment[i] * s->integ_m[i];
		double fb[3];
		mv3(sm, e, fb);
		double n = sqrt(fb[0] * fb[0] + fb[1] * fb[1] + fb[2] * fb[2]);
		if (n > t->max_moment_feedback)
			for (int i = 0; i < 3; i++) fb[i] *= t->max_moment_feedback / n;
		for (int i = 0; i < 3; i++) e[i] = fb[i] - t->kv_moment[i] * w[i];
		mv3(sm, e, f_moment);
	} else {
		for (int i = 0; i < 3; i++) e[i] = -t->kv_moment[i] * w[i];
		mv3(sm, e, f_moment);
	}
	/* desired state: the goal, or the next OTG state (:385-407) */
	memcpy(s->des_pos, s->g_pos, sizeof(s->des_pos));
	memcpy(s->des_rot, s->g_rot, sizeof(s->des_rot));
	memcpy(s->des_v, s->g_v, sizeof(s->des_v));
	memcpy(s->des_w, s->g_w, sizeof(s->des_w));
	memcpy(s->des_a, s->g_a, sizeof(s->des_a));
	memcpy(s->des_al, s->g_al, sizeof(s->des_al));
	if (t->use_internal_otg) {
		otg_cartesian_set_goal_position(o, s->g_pos, s->g_v);
		otg_cartesian_set_goal_orientation(o, s->g_rot, s->g_w);
		otg_cartesian_update(o);
		for (int i = 0; i < 3; i++) {
			s->des_pos[i] = o->output.np[i];
			s->des_v[i] = o->output.nv[i];
			s->des_a[i] = o->output.na[i];
		}
		otg_cartesian_next_orientation(o, s->des_rot);
		otg_cartesian_next_angular(o, s->des_w, s->des_al);
	}
	/* linear motion (:409-437) */
	double des_v[3], des_w[3];
	memcpy(des_v, s->des_v, sizeof(des_v));
	memcpy(des_w, s->des_w, sizeof(des_w));
	for (int i = 0; i < 3; i++) e[i] = x[i] - s->des_pos[i];
	mv3(sp, e, y);
	for (int i = 0; i < 3; i++) s->integ_pos[i] += y[i] * dt;
	if (t->use_velocity_saturation) {
		for (int i = 0; i < 3; i++) {
			double kvi = gain_pinv(t->kv_pos[i]);
			des_v[i] = -t->kp_pos[i] * kvi * y[i] - t->ki_pos[i] * kvi * s->integ_pos[i];
		}
		double n = sqrt(des_v[0] * des_v[0] + des_v[1] * des_v[1] + des_v[2] * des_v[2]);
		if (n > t->linear_saturation_velocity)
			for (int i = 0; i < 3; i++) des_v[i] *= t->linear_saturation_velocity / n;
		for (int i = 0; i < 3; i++) e[i] = s->des_a[i] - t->kv_pos[i] * (v[i] - des_v[i]);
	} else {
		for (int i = 0; i < 3; i++)
			e[i] = s->des_a[i] - t->kp_pos[i] * (x[i] - s->des_pos[i]) - t->kv_pos[i] * (v[i] - des_v[i]) -
				   t->ki_pos[i] * s->integ_pos[i];
	}
	mv3(sp, e, f_pos);
	/* angular motion (:439-468) */
	double oe[3], step[3];
	orientation_error(s->des_rot, R, oe);
	mv3(so, oe, step);
	for (int i = 0; i < 3; i++) s->integ_ori[i] += step[i] * dt;
	if (t->use_velocity_saturation) {
		for (int i = 0; i < 3; i++) {
			double kvi = gain_pinv(t->kv_ori[i]);
			des_w[i] = -t->kp_ori[i] * kvi * step[i] - t->ki_ori[i] * kvi * s->integ_ori[i];
		}
		double n = sqrt(des_w[0] * des_w[0] + des_w[1] * des_w[1] + des_w[2] * des_w[2]);
		if (n > t->angular_saturation_velocity)
			for (int i = 0; i < 3; i++) des_w[i] *= t->angular_saturation_velocity / n;
		for (int i = 0; i < 3; i++) e[i] = s->des_al[i] - t->kv_ori[i] * (w[i] - des_w[i]);
	} else {
		for (int i = 0; i < 3; i++)
			e[i] = s->des_al[i] - t->kp_ori[i] * step[i] - t->kv_ori[i] * (w[i] - des_w[i]) -
				   t->ki_ori[i] * s->integ_ori[i];
	}
	mv3(so, e, f_ori);
	/* task force (:470-506) */
	double Fu[6], Ff[6], ff[6];
	for (int i = 0; i < 3; i++) {
		Fu[i] = f_pos[i];
		Fu[3 + i] = f_ori[i];
	}
	mv3(sf, gf, ff);
	mv3(sm, gm, ff + 3);
	if (t->closed_loop_force) /* sic: one flag scales both (:484-487) */
		for (int i = 0; i < 3; i++) {
			ff[i] *= t->kff_force;
			ff[3 + i] *= t->kff_moment;
		}
	for (int i = 0; i < 3; i++) {
		Ff[i] = f_force[i] + ff[i];
		Ff[3 + i] = f_moment[i] + ff[3 + i];
	}
	memcpy(s->Fu, Fu, sizeof(Fu));
	memcpy(s->Ff, Ff, sizeof(Ff));
	sh_torques(c, t, r, s, Fu, Ff, tau);
}

/* ------------------------------------------------------------------------------------------
 * batch API
 * ------------------------------------------------------------------------------------------ */
oracle_ctx* oracle_create(const sai2b_robot_model* model, const sai2b_task_config* tasks, int n_tasks, int batch) {
	if (!model || !tasks || n_tasks < 1 || n_tasks > SAI2B_MAX_TASKS || batch < 1 || model->dof != N7) {
		fail("oracle_create: bad arguments");
		return NULL;
	}
	/* RobotController.cpp:8-51 */
	int closed = 0;
	for (int i = 0; i < n_tasks; i++) {
		if (tasks[i].loop_timestep != tasks[0].loop_timestep) {
			fail("All tasks must have the same loop timestep in RobotController");
			return NULL;
		}
		for (int j = 0; j < i; j++)
			if (strncmp(tasks[i].name, tasks[j].name, sizeof(tasks[i].name)) == 0) {
				fail("Tasks in RobotController must have unique names");
				return NULL;
			}
		if (closed) {
			fail("task cannot be added to the controller because it is in the nullspace of a full joint task");
			return NULL;
		}
		if (tasks[i].type == SAI2B_JOINT_TASK && tasks[i].task_dof == N7) closed = 1;
	}
	oracle_ctx* c = (oracle_ctx*)calloc(1, sizeof(oracle_ctx));
	c->B = batch;
	c->T = n_tasks;
	c->threads = 1;
	c->model = *model;
	for (int i = 0; i < N7; i++) rpy_to_rot(model->joint_rpy[i], c->E[i]);
	c->robots = (robot_t*)calloc(batch, sizeof(robot_t));
	for (int b = 0; b < batch; b++) c->robots[b].jtype = c->model.joint_type;
	for (int i = 0; i < n_tasks; i++) {
		c->cfg[i] = tasks[i];
		if (tasks[i].type == SAI2B_JOINT_TASK) {
			c->jt[i] = (jt_t*)calloc(batch, sizeof(jt_t));
			c->jotg[i] = (otg_joints*)calloc(batch, sizeof(otg_joints));
		} else {
			c->mft[i] = (mft_t*)calloc(batch, sizeof(mft_t));
			c->cotg[i] = (otg_cartesian*)calloc(batch, sizeof(otg_cartesian));
		}
	}
	/* the reference constructs tasks from the model's current state; here q = 0 until set */
	for (int b = 0; b < batch; b++) {
		update_model(c, &c->robots[b]);
		for (int i = 0; i < n_tasks; i++) {
			if (c->jt[i]) {
				jt_construct(&c->cfg[i], &c->robots[b], &c->jt[i][b], &c->jotg[i][b]);
				eye(N7, c->jt[i][b].N_prec);
				c->jt[i][b].k = 0;
			} else {
				mft_construct(c, &c->cfg[i], &c->robots[b], &c->mft[i][b], &c->cotg[i][b]);
				sh_init(c, &c->mft[i][b]);
				popc_init(&c->mft[i][b]);
				eye(N7, c->mft[i][b].N_prec);
			}
		}
	}
	return c;
}
void oracle_destroy(oracle_ctx* c) {
	if (!c) return;
	for (int i = 0; i < SAI2B_MAX_TASKS; i++) {
		free(c->jt[i]);
		free(c->mft[i]);
		free(c->jotg[i]);
		free(c->cotg[i]);
	}
	free(c->robots);
	free(c);
}
void oracle_set_threads(oracle_ctx* c, int n) { c->threads = n < 1 ? 1 : n; }
/* `reset` of parametrizeForceMotionSpaces (MotionForceTask.cpp:838-848): dimension changed, or, for
 * dimension 1 or 2, the normalised axis is not isApprox (1e-12) the one in use */
static int space_changed(int dim, const double* axis, int old_dim, const double* old_axis) {
	if (dim != old_dim) return 1;
	if (dim != 1 && dim != 2) return 0;
	double a[3], b[3], na = 0, nb = 0, d2 = 0, a2 = 0, b2 = 0;
	for (int i = 0; i < 3; i++) na += axis[i] * axis[i], nb += old_axis[i] * old_axis[i];
	na = sqrt(na), nb = sqrt(nb);
	for (int i = 0; i < 3; i++) {
		a[i] = axis[i] / na, b[i] = old_axis[i] / nb;
		d2 += (a[i] - b[i]) * (a[i] - b[i]), a2 += a[i] * a[i], b2 += b[i] * b[i];
	}
	return !(d2 <= 1e-24 * (a2 < b2 ? a2 : b2));
}
int oracle_update_task_config(oracle_ctx* c, int task, const sai2b_task_config* cfg) {
	if (!c || task < 0 || task >= c->T || !cfg || cfg->type != c->cfg[task].type ||
		cfg->task_dof != c->cfg[task].task_dof)
		return fail("update_task_config: bad arguments");
	if (c->mft[task] && (cfg->passivity_enabled != 0) != (c->cfg[task].passivity_enabled != 0))
		for (int b = 0; b < c->B; b++) popc_init(&c->mft[task][b]); /* enable()/disable() */
	if (cfg->internal_otg_jerk_limited && cfg->use_internal_otg && !otg_jerk_planner_available())
		return fail("update_task_config: the jerk-limited OTG needs oracle/_ref/libruckig_ref.so (make -C oracle ref)");
	/* enableInternalOtgAccelerationLimited() / ...JerkLimited() is applied when the OTG fields change */
	const sai2b_task_config* old = &c->cfg[task];
	const int jerk = cfg->internal_otg_jerk_limited != 0, was_jerk = old->internal_otg_jerk_limited != 0;
	int otg_changed = (cfg->use_internal_otg != 0) != (old->use_internal_otg != 0) || jerk != was_jerk;
	if (c->jt[task]) {
		for (int i = 0; i < cfg->task_dof; i++)
			otg_changed |= cfg->otg_max_velocity[i] != old->otg_max_velocity[i] ||
						   cfg->otg_max_acceleration[i] != old->otg_max_acceleration[i] ||
						   (jerk && cfg->otg_max_jerk[i] != old->otg_max_jerk[i]);
	} else {
		otg_changed |= cfg->otg_max_linear_velocity != old->otg_max_linear_velocity ||
					   cfg->otg_max_linear_acceleration != old->otg_max_linear_acceleration ||
					   cfg->otg_max_angular_velocity != old->otg_max_angular_velocity ||
					   cfg->otg_max_angular_acceleration != old->otg_max_angular_acceleration ||
					   (jerk && (cfg->otg_max_linear_jerk != old->otg_max_linear_jerk || cfg->otg_max_angular_jerk != old->otg_max_angular_jerk));
	}
	if (cfg->use_internal_otg && otg_changed)
		for (int b = 0; b < c->B; b++) {
			if (c->jt[task])
				jt_otg_enable(cfg, &c->jt[task][b], &c->jotg[task][b], old->use_internal_otg != 0, was_jerk);
			else
				mft_otg_enable(cfg, &c->mft[task][b], &c->cotg[task][b], old->use_internal_otg != 0, was_jerk);
		}
	if (c->mft[task]) {
		/* parametrizeForceMotionSpaces / parametrizeMomentRotMotionSpaces (MotionForceTask.cpp:830-890): a new
		 * dimension, or a new axis for dimension 1 or 2, moves the goal to the current pose, re-initialises that
		 * half of the generator there and resets that half of the integrators; setClosedLoopForceControl /
		 * setClosedLoopMomentControl (:973-986) reset them when the mode changes */
		const int lin = space_changed(cfg->force_space_dimension, cfg->force_axis, old->force_space_dimension, old->force_axis);
		const int ang = space_changed(cfg->moment_space_dimension, cfg->moment_axis, old->moment_space_dimension, old->moment_axis);
		const int cl_f = (cfg->closed_loop_force != 0) != (old->closed_loop_force != 0);
		const int cl_m = (cfg->closed_loop_moment != 0) != (old->closed_loop_moment != 0);
		for (int b = 0; b < c->B && (lin || ang || cl_f || cl_m); b++) {
			mft_t* s = &c->mft[task][b];
			if (lin) {
				memcpy(s->g_pos, s->cur_pos, sizeof(s->g_pos));
				for (int i = 0; i < 3; i++) s->g_v[i] = s->g_a[i] = 0;
				otg_cartesian_reinitialize_linear(&c->cotg[task][b], s->cur_pos);
			}
			if (ang) {
				memcpy(s->g_rot, s->cur_rot, sizeof(s->g_rot));
				for (int i = 0; i < 3; i++) s->g_w[i] = s->g_al[i] = 0;
				otg_cartesian_reinitialize_angular(&c->cotg[task][b], s->cur_rot);
			}
			for (int i = 0; i < 3; i++) {
				if (lin || cl_f) s->integ_pos[i] = s->integ_f[i] = 0;
				if (ang || cl_m) s->integ_ori[i] = s->integ_m[i] = 0;
			}
		}
	}
	c->cfg[task] = *cfg;
	return 0;
}
int oracle_enable_gravity_compensation(oracle_ctx* c, int e) {
	c->gravity_comp = e;
	return 0;
}
int oracle_set_state(oracle_ctx* c, const double* q, const double* dq) {
	for (int b = 0; b < c->B; b++) {
		for (int i = 0; i < N7; i++) {
			if (q) c->robots[b].q[i] = q[i * c->B + b];
			if (dq) c->robots[b].dq[i] = dq[i * c->B + b];
		}
		c->robots[b].model_valid = 0;
	}
	return 0;
}
#define COPY_IN(dst, src, n)                                                  \
	if (src)                                                                  \
		for (int b = 0; b < c->B; b++)                                        \
			for (int i = 0; i < (n); i++) s[b].dst[i] = (src)[i * c->B + b];
int oracle_set_mft_goals(oracle_ctx* c, int task, const double* pos, const double* rot, const double* lv,
						 const double* av, const double* la, const double* aa) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	mft_t* s = c->mft[task];
	COPY_IN(g_pos, pos, 3) COPY_IN(g_rot, rot, 9) COPY_IN(g_v, lv, 3) COPY_IN(g_w, av, 3) COPY_IN(g_a, la, 3)
		COPY_IN(g_al, aa, 3) return 0;
}
int oracle_set_mft_goal_wrench(oracle_ctx* c, int task, const double* f, const double* m) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	mft_t* s = c->mft[task];
	COPY_IN(g_f, f, 3) COPY_IN(g_m, m, 3) return 0;
}
int oracle_set_mft_sensed_wrench(oracle_ctx* c, int task, const double* f, const double* m) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	mft_t* s = c->mft[task];
	COPY_IN(sens_f, f, 3) COPY_IN(sens_m, m, 3) return 0;
}
int oracle_set_jt_goals(oracle_ctx* c, int task, const double* qg, const double* dqg, const double* ddqg) {
	if (task < 0 || task >= c->T || !c->jt[task]) return fail("not a JointTask");
	jt_t* s = c->jt[task];
	int k0 = c->cfg[task].task_dof;
	COPY_IN(goal_q, qg, k0) COPY_IN(goal_dq, dqg, k0) COPY_IN(goal_ddq, ddqg, k0) return 0;
}
static void ensure_model(oracle_ctx* c, int b) {
	if (!c->robots[b].model_valid) update_model(c, &c->robots[b]);
}
int oracle_reinitialize(oracle_ctx* c) {
	for (int b = 0; b < c->B; b++) {
		ensure_model(c, b);
		for (int i = 0; i < c->T; i++) {
			if (c->jt[i])
				jt_reinit(&c->cfg[i], &c->robots[b], &c->jt[i][b], &c->jotg[i][b]);
			else {
				mft_reinit(c, &c->cfg[i], &c->robots[b], &c->mft[i][b], &c->cotg[i][b]);
				sh_init(c, &c->mft[i][b]);
			}
		}
	}
	return 0;
}
static void robot_update_models(oracle_ctx* c, int b) {
	/* RobotController.cpp:53-60 */
	robot_t* r = &c->robots[b];
	ensure_model(c, b);
	double N_prec[NN], T[NN];
	eye(N7, N_prec);
	for (int i = 0; i < c->T; i++) {
		const double* N;
		if (c->jt[i]) {
			jt_update(&c->cfg[i], r, &c->jt[i][b], N_prec);
			N = c->jt[i][b].N;
		} else {
			mft_update(c, &c->cfg[i], r, &c->mft[i][b], N_prec);
			N = c->mft[i][b].N;
		}
		mm(N7, N7, N7, N, N_prec, T); /* getTaskAndPreviousNullspace = N * N_prec */
		memcpy(N_prec, T, sizeof(T));
	}
}
static void robot_torques(oracle_ctx* c, int b, double* tau, int with_comp) {
	/* RobotController.cpp:62-74 */
	robot_t* r = &c->robots[b];
	ensure_model(c, b);
	for (int i = 0; i < N7; i++) tau[i] = 0;
	for (int i = 0; i < c->T; i++) {
		double tt[N7], comp[N7];
		if (c->jt[i]) {
			jt_torques(&c->cfg[i], r, &c->jt[i][b], &c->jotg[i][b], tt);
			if (with_comp) {
				jt_compensation(&c->cfg[i], r, &c->jt[i][b], tau, comp);
				for (int k = 0; k < N7; k++) tt[k] -= comp[k];
			}
			memcpy(c->jt[i][b].tau, tt, sizeof(tt));
		} else {
			/* MotionForceTask.cpp:270-276: the compensation term multiplies the never-assigned zero
			 * _Lambda => identically zero */
			mft_torques(c, &c->cfg[i], r, &c->mft[i][b], &c->cotg[i][b], tt);
			memcpy(c->mft[i][b].tau, tt, sizeof(tt));
		}
		for (int k = 0; k < N7; k++) tau[k] += tt[k];
	}
	if (c->gravity_comp) {
		double g[N7];
		gravity_vector(c, r, g);
		for (int k = 0; k < N7; k++) tau[k] += g[k];
	}
}
int oracle_update_task_models(oracle_ctx* c) {
#pragma omp parallel for num_threads(c->threads) schedule(static)
	for (int b = 0; b < c->B; b++) robot_update_models(c, b);
	return 0;
}
int oracle_compute_control_torques(oracle_ctx* c, double* tau, int with_comp) {
#pragma omp parallel for num_threads(c->threads) schedule(static)
	for (int b = 0; b < c->B; b++) {
		double t[N7];
		robot_torques(c, b, t, with_comp);
		if (tau)
			for (int i = 0; i < N7; i++) tau[i * c->B + b] = t[i];
	}
	return 0;
}
int oracle_tick(oracle_ctx* c, double* tau) {
#pragma omp parallel for num_threads(c->threads) schedule(static)
	for (int b = 0; b < c->B; b++) {
		double t[N7];
		c->robots[b].model_valid = 0; /* robot->updateModel() every tick (examples/05:143-145) */
		robot_update_models(c, b);
		robot_torques(c, b, t, 1);
		if (tau)
			for (int i = 0; i < N7; i++) tau[i * c->B + b] = t[i];
	}
	return 0;
}

/* ---- task-level plugin interface (TemplateTask.h:42-88): one task driven on its own, the caller chaining
 * the nullspaces (examples/04-task_and_redundancy.cpp:141-150,188-189; examples/01-joint_control.cpp:131-191) ---- */
int oracle_task_update_model(oracle_ctx* c, int task, const double* N_prec) {
	if (task < 0 || task >= c->T) return fail("bad task");
#pragma omp parallel for num_threads(c->threads) schedule(static)
	for (int b = 0; b < c->B; b++) {
		double Np[NN];
		if (N_prec)
			for (int i = 0; i < NN; i++) Np[i] = N_prec[i * c->B + b];
		else
			eye(N7, Np);
		ensure_model(c, b);
		if (c->jt[task])
			jt_update(&c->cfg[task], &c->robots[b], &c->jt[task][b], Np); /* JointTask.cpp:218-283 */
		else
			mft_update(c, &c->cfg[task], &c->robots[b], &c->mft[task][b], Np); /* MotionForceTask.cpp:247-268 */
	}
	return 0;
}
/* computeTorques() (tau_prec == NULL) / computeTorques(tau_prec) with the task's cached model */
int oracle_task_compute_torques(oracle_ctx* c, int task, const double* tau_prec, double* tau) {
	if (task < 0 || task >= c->T) return fail("bad task");
#pragma omp parallel for num_threads(c->threads) schedule(static)
	for (int b = 0; b < c->B; b++) {
		robot_t* r = &c->robots[b];
		double tt[N7], comp[N7], tp[N7];
		ensure_model(c, b);
		if (c->jt[task]) {
			jt_torques(&c->cfg[task], r, &c->jt[task][b], &c->jotg[task][b], tt);
			if (tau_prec) { /* JointTask.cpp:285-292 */
				for (int k = 0; k < N7; k++) tp[k] = tau_prec[k * c->B + b];
				jt_compensation(&c->cfg[task], r, &c->jt[task][b], tp, comp);
				for (int k = 0; k < N7; k++) tt[k] -= comp[k];
			}
			memcpy(c->jt[task][b].tau, tt, sizeof(tt));
		} else { /* MotionForceTask.cpp:270-276: the compensation term is identically zero (App. B-1) */
			mft_torques(c, &c->cfg[task], r, &c->mft[task][b], &c->cotg[task][b], tt);
			memcpy(c->mft[task][b].tau, tt, sizeof(tt));
		}
		if (tau)
			for (int k = 0; k < N7; k++) tau[k * c->B + b] = tt[k];
	}
	return 0;
}
int oracle_task_reinitialize(oracle_ctx* c, int task) {
	if (task < 0 || task >= c->T) return fail("bad task");
	for (int b = 0; b < c->B; b++) {
		ensure_model(c, b);
		if (c->jt[task])
			jt_reinit(&c->cfg[task], &c->robots[b], &c->jt[task][b], &c->jotg[task][b]);
		else {
			mft_reinit(c, &c->cfg[task], &c->robots[b], &c->mft[task][b], &c->cotg[task][b]);
			sh_init(c, &c->mft[task][b]);
		}
	}
	return 0;
}
int oracle_task_get_nullspaces(oracle_ctx* c, int task, double* N, double* N_prec, double* N_total) {
	if (task < 0 || task >= c->T) return fail("bad task");
	for (int b = 0; b < c->B; b++) {
		const double* n = c->jt[task] ? c->jt[task][b].N : c->mft[task][b].N;
		const double* np = c->jt[task] ? c->jt[task][b].N_prec : c->mft[task][b].N_prec;
		double T[NN];
		mm(N7, N7, N7, n, np, T);
		for (int i = 0; i < NN; i++) {
			if (N) N[i * c->B + b] = n[i];
			if (N_prec) N_prec[i * c->B + b] = np[i];
			if (N_total) N_total[i * c->B + b] = T[i];
		}
	}
	return 0;
}

/* ---- getters ---- */
int oracle_get_task_nullspace(oracle_ctx* c, int task, double* out) {
	if (task < 0 || task >= c->T) return fail("bad task");
	for (int b = 0; b < c->B; b++) {
		double T[NN];
		if (c->jt[task])
			mm(N7, N7, N7, c->jt[task][b].N, c->jt[task][b].N_prec, T);
		else
			mm(N7, N7, N7, c->mft[task][b].N, c->mft[task][b].N_prec, T);
		for (int i = 0; i < NN; i++) out[i * c->B + b] = T[i];
	}
	return 0;
}
int oracle_get_task_torques(oracle_ctx* c, int task, double* out) {
	if (task < 0 || task >= c->T) return fail("bad task");
	for (int b = 0; b < c->B; b++)
		for (int i = 0; i < N7; i++) out[i * c->B + b] = c->jt[task] ? c->jt[task][b].tau[i] : c->mft[task][b].tau[i];
	return 0;
}
int oracle_get_mft_singularity(oracle_ctx* c, int task, double* sigma, double* alpha, double* ns_rank) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	for (int b = 0; b < c->B; b++) {
		const mft_t* s = &c->mft[task][b];
		if (sigma)
			for (int i = 0; i < 6; i++) sigma[i * c->B + b] = s->s[i];
		if (alpha) alpha[b] = s->alpha;
		if (ns_rank) ns_rank[b] = s->ns;
	}
	return 0;
}
int oracle_get_model(oracle_ctx* c, int task, double* M, double* J, double* pos, double* rot) {
	for (int b = 0; b < c->B; b++) {
		ensure_model(c, b);
		const robot_t* r = &c->robots[b];
		if (M)
			for (int i = 0; i < NN; i++) M[i * c->B + b] = r->M[i];
		if (J || pos || rot) {
			if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
			double Jw[J6], x[3], R[9];
			jacobian(r, c->cfg[task].link, c->cfg[task].frame_pos, Jw);
			frame_pose(&c->cfg[task], r->Rl, r->pl, x, R);
			if (J)
				for (int i = 0; i < J6; i++) J[i * c->B + b] = Jw[i];
			if (pos)
				for (int i = 0; i < 3; i++) pos[i * c->B + b] = x[i];
			if (rot)
				for (int i = 0; i < 9; i++) rot[i * c->B + b] = R[i];
		}
	}
	return 0;
}
int oracle_get_minv(oracle_ctx* c, double* out) {
	for (int b = 0; b < c->B; b++) {
		ensure_model(c, b);
		for (int i = 0; i < NN; i++) out[i * c->B + b] = c->robots[b].Minv[i];
	}
	return 0;
}
int oracle_get_gravity(oracle_ctx* c, double* out) {
	for (int b = 0; b < c->B; b++) {
		double g[N7];
		ensure_model(c, b);
		gravity_vector(c, &c->robots[b], g);
		for (int i = 0; i < N7; i++) out[i * c->B + b] = g[i];
	}
	return 0;
}
static void embed(int rows, int cols, const double* U, const double* L, double* out) { /* U L U^T */
	double T[NN];
	mm(rows, cols, cols, U, L, T);
	mm_nt(rows, cols, rows, T, U, out);
}
int oracle_get_mft_lambda(oracle_ctx* c, int task, double* L_full, double* Lmod_full) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	for (int b = 0; b < c->B; b++) {
		const mft_t* s = &c->mft[task][b];
		double A[36] = {0}, Bm[36] = {0};
		if (s->ns) {
			embed(6, s->ns, s->U_ns, s->L_ns, A);
			embed(6, s->ns, s->U_ns, s->L_ns_mod, Bm);
		}
		for (int i = 0; i < 36; i++) {
			if (L_full) L_full[i * c->B + b] = A[i];
			if (Lmod_full) Lmod_full[i * c->B + b] = Bm[i];
		}
	}
	return 0;
}
int oracle_get_mft_task_forces(oracle_ctx* c, int task, double* Fu, double* Ff) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	for (int b = 0; b < c->B; b++)
		for (int i = 0; i < 6; i++) {
			if (Fu) Fu[i * c->B + b] = c->mft[task][b].Fu[i];
			if (Ff) Ff[i * c->B + b] = c->mft[task][b].Ff[i];
		}
	return 0;
}
/* one control period of the simulated robots: tau ([7][B], NULL = zero) held over dt, `substeps`
 * semi-implicit Euler steps dq += h M^-1 (tau - b), q += h dq */
int oracle_sim_step(oracle_ctx* c, const double* tau, double dt, int substeps, int with_gravity) {
	if (!c || substeps < 1 || !(dt > 0)) return fail("oracle_sim_step: bad arguments");
	const double h = dt / substeps;
#pragma omp parallel for num_threads(c->threads) schedule(static)
	for (int b = 0; b < c->B; b++) {
		robot_t* r = &c->robots[b];
		for (int s = 0; s < substeps; s++) {
			update_model(c, r);
			double bias[N7], rhs[N7], qdd[N7];
			bias_vector(c, r, with_gravity, bias);
			for (int i = 0; i < N7; i++) rhs[i] = (tau ? tau[i * c->B + b] : 0.0) - bias[i];
			mm(N7, N7, 1, r->Minv, rhs, qdd);
			for (int i = 0; i < N7; i++) {
				r->dq[i] += h * qdd[i];
				r->q[i] += h * r->dq[i];
			}
		}
		r->model_valid = 0;
	}
	return 0;
}
int oracle_get_state(oracle_ctx* c, double* q, double* dq) {
	for (int b = 0; b < c->B; b++)
		for (int i = 0; i < N7; i++) {
			if (q) q[i * c->B + b] = c->robots[b].q[i];
			if (dq) dq[i * c->B + b] = c->robots[b].dq[i];
		}
	return 0;
}
/* bias vector C dq (+ g) [7][B] at the current state */
int oracle_get_bias(oracle_ctx* c, int with_gravity, double* out) {
	for (int b = 0; b < c->B; b++) {
		ensure_model(c, b);
		double g[N7];
		bias_vector(c, &c->robots[b], with_gravity, g);
		for (int i = 0; i < N7; i++) out[i * c->B + b] = g[i];
	}
	return 0;
}
/* observers between ticks (MotionForceTask.h:121-165, MotionForceTask.cpp:540-579), from the current
 * state and goals; out arrays [rows][B], any NULL */
int oracle_get_mft_status(oracle_ctx* c, int task, double* pos, double* rot, double* fw, double* mw, double* pe,
						  double* oe_out, double* pn, double* on) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	const sai2b_task_config* t = &c->cfg[task];
	for (int b = 0; b < c->B; b++) {
		ensure_model(c, b);
		const robot_t* r = &c->robots[b];
		const mft_t* s = &c->mft[task][b];
		double x[3], R[9], sf[9], sp[9], sm[9], so[9], e[3], oe[3], se[3], soe[3];
		frame_pose(t, r->Rl, r->pl, x, R);
		sigma_pair(t->partial_projection, 0, t->force_space_dimension, t->force_axis, R, t->parametrization_in_compliant_frame, sf, sp);
		sigma_pair(t->partial_projection, 1, t->moment_space_dimension, t->moment_axis, R, t->parametrization_in_compliant_frame, sm, so);
		for (int i = 0; i < 3; i++) e[i] = s->g_pos[i] - x[i];
		orientation_error(s->g_rot, R, oe);
		mv3(sp, e, se);
		mv3(so, oe, soe);
		double fs_c[3], ms_c[3], tmp[3], fs_w[3], ms_w[3];
		mv3(t->sensor_rot, s->sens_f, fs_c);
		mv3(t->sensor_rot, s->sens_m, ms_c);
		cross3(t->sensor_pos, fs_c, tmp);
		for (int i = 0; i < 3; i++) ms_c[i] += tmp[i];
		mv3(R, fs_c, fs_w);
		mv3(R, ms_c, ms_w);
		for (int i = 0; i < 3; i++) {
			if (pos) pos[i * c->B + b] = x[i];
			if (fw) fw[i * c->B + b] = fs_w[i];
			if (mw) mw[i * c->B + b] = ms_w[i];
			if (pe) pe[i * c->B + b] = se[i];
			if (oe_out) oe_out[i * c->B + b] = soe[i];
		}
		if (rot)
			for (int i = 0; i < 9; i++) rot[i * c->B + b] = R[i];
		double a = e[0] * se[0] + e[1] * se[1] + e[2] * se[2], d = oe[0] * soe[0] + oe[1] * soe[1] + oe[2] * soe[2];
		if (pn) pn[b] = sqrt(a > 0 ? a : 0);
		if (on) on[b] = sqrt(d > 0 ? d : 0);
	}
	return 0;
}
/* MotionForceTask::resetIntegrators* (MotionForceTask.cpp:988-1001), JointTask::resetIntegrators */
int oracle_reset_integrators(oracle_ctx* c, int task, int which) {
	if (task < 0 || task >= c->T) return fail("bad task");
	for (int b = 0; b < c->B; b++) {
		if (c->jt[task]) {
			for (int i = 0; i < N7; i++) c->jt[task][b].integ[i] = 0;
		} else {
			mft_t* s = &c->mft[task][b];
			for (int i = 0; i < 3; i++) {
				if (which == 0 || which == 1) s->integ_pos[i] = s->integ_f[i] = 0;
				if (which == 0 || which == 2) s->integ_ori[i] = s->integ_m[i] = 0;
			}
		}
	}
	return 0;
}
/* test observer: the four integrators of a MotionForceTask, [12][B]: position, orientation, force, moment */
int oracle_get_mft_integrators(oracle_ctx* c, int task, double* out) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	for (int b = 0; b < c->B; b++) {
		const mft_t* s = &c->mft[task][b];
		for (int i = 0; i < 3; i++) {
			out[(0 + i) * c->B + b] = s->integ_pos[i];
			out[(3 + i) * c->B + b] = s->integ_ori[i];
			out[(6 + i) * c->B + b] = s->integ_f[i];
			out[(9 + i) * c->B + b] = s->integ_m[i];
		}
	}
	return 0;
}
int oracle_get_jt_desired(oracle_ctx* c, int task, double* q, double* dq, double* ddq) {
	if (task < 0 || task >= c->T || !c->jt[task]) return fail("not a JointTask");
	const int k0 = c->cfg[task].task_dof;
	for (int b = 0; b < c->B; b++)
		for (int i = 0; i < k0; i++) {
			if (q) q[i * c->B + b] = c->jt[task][b].des_q[i];
			if (dq) dq[i * c->B + b] = c->jt[task][b].des_dq[i];
			if (ddq) ddq[i * c->B + b] = c->jt[task][b].des_ddq[i];
		}
	return 0;
}
int oracle_get_mft_desired(oracle_ctx* c, int task, double* pos, double* rot, double* v, double* w, double* a,
						   double* al) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	for (int b = 0; b < c->B; b++) {
		const mft_t* s = &c->mft[task][b];
		for (int i = 0; i < 3; i++) {
			if (pos) pos[i * c->B + b] = s->des_pos[i];
			if (v) v[i * c->B + b] = s->des_v[i];
			if (w) w[i * c->B + b] = s->des_w[i];
			if (a) a[i * c->B + b] = s->des_a[i];
			if (al) al[i * c->B + b] = s->des_al[i];
		}
		if (rot)
			for (int i = 0; i < 9; i++) rot[i * c->B + b] = s->des_rot[i];
	}
	return 0;
}
/* OTG flags per robot: goal reached, last ruckig result (as doubles) */
int oracle_get_otg_status(oracle_ctx* c, int task, double* goal_reached, double* result) {
	if (task < 0 || task >= c->T) return fail("bad task");
	for (int b = 0; b < c->B; b++) {
		if (c->jt[task]) {
			goal_reached[b] = c->jotg[task][b].goal_reached;
			result[b] = c->jotg[task][b].result_value;
		} else {
			goal_reached[b] = c->cotg[task][b].goal_reached;
			result[b] = c->cotg[task][b].result_value;
		}
	}
	return 0;
}
int oracle_get_mft_sh_state(oracle_ctx* c, int task, double* first_type, double* c1, double* c2) {
	if (task < 0 || task >= c->T || !c->mft[task]) return fail("not a MotionForceTask");
	for (int b = 0; b < c->B; b++) {
		const mft_t* s = &c->mft[task][b];
		if (first_type) first_type[b] = s->n_types ? s->types[0] : 0;
		if (c1) c1[b] = s->c1;
		if (c2) c2[b] = s->c2;
	}
	return 0;
}
int oracle_get_jt_inertia(oracle_ctx* c, int task, double* Mp, double* Mpm) {
	if (task < 0 || task >= c->T || !c->jt[task]) return fail("not a JointTask");
	int k0 = c->cfg[task].task_dof;
	for (int b = 0; b < c->B; b++) {
		const jt_t* s = &c->jt[task][b];
		double A[NN] = {0}, Bm[NN] = {0};
		if (s->k) {
			embed(k0, s->k, s->R, s->M_partial, A);
			embed(k0, s->k, s->R, s->M_partial_mod, Bm);
		}
		for (int i = 0; i < k0 * k0; i++) {
			if (Mp) Mp[i * c->B + b] = A[i];
			if (Mpm) Mpm[i * c->B + b] = Bm[i];
		}
	}
	return 0;
}
