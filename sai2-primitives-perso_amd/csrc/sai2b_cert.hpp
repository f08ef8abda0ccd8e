// sai2b_cert.hpp — SVD-free tick for ANY hierarchy of MotionForceTasks and JointTasks, one lane per robot.
//
// sai2b_fast.hpp covers [full MFT(, full JT)] on a 7-joint arm; this is the same idea for every other hierarchy
// (partial tasks, several levels, 4 / 6 / 7 / 8 joints, prismatic joints): a robot runs here as long as every task
// carries the certificate that the reference's decision is "full rank, fully non-singular"
// (SingularityHandler.cpp:100-141, the 1e-3 range rule of JointTask.cpp:233); the first task that cannot be
// vouched for sends the robot — with nothing stored yet — to the work list of the generic kernel.
//
// Mathematics. With M = L L^T every dynamically consistent nullspace of the cascade (RobotController.cpp:53-63,
// MotionForceTask.h:207-209) is   N_prec = L^-T Q L^T   with Q an ORTHOGONAL projector of R^n:
//     Jp = Jr N_prec  (Jr: the task's rows, PU^T J or S)      Y := L^-1 Jp^T = Q (L^-1 Jr^T)
//     Jp M^-1 Jp^T = Y^T Y =: R^T R  (Y = Z R, Z orthonormal)   Lambda = R^-1 R^-T
//     N = I - M^-1 Jp^T Lambda Jp = L^-T (I - Z Z^T) L^T        Q <- Q - Z Z^T
//     torques Jp^T F = (L Y) F
// so a level costs m triangular solves, m symmetric products, one Gram-Schmidt and a rank-m downdate instead
// of the 6 x 6 / 7 x 7 sandwiches, projector pseudo-inverses and nullspace products of the projector form
// (sai2b_device.hpp, sai2b_group_tick.hpp) — the results are the same functions of (J, M), equal up to rounding.
// Reduced coordinates: a partial MotionForceTask works on the rows PU^T J (PU: basis of range(P)); forces enter as
// PU^T F. A full JointTask behind other tasks (Jp = N_prec, JointTask.cpp:218-283) needs a basis C of range(Q)
// (pivoted Cholesky of the projector, d = n - rows consumed so far columns):
//     A = L^-T C, K = A^T A:  R M_partial R^T = L^-T C K^-2 C^T L^-1,  torques  L C K^-1 A^T v
// and the bounded-inertia / impedance variants below; for d = 1 these are the rank-one formulas of sai2b_fast.hpp.
//
// State (integrators) is written only once the robot is known to finish here: the values wait in LDS
// (PEND_SLOTS doubles per lane).
#pragma once
#include "sai2b_device.hpp"
#include "sai2b_fast.hpp"

namespace sai2b {
namespace cert {

constexpr int MM = N > 6 ? N : 6;  // most rows one task brings (MotionForceTask: 6, JointTask: N)
constexpr int DM = N - 1;		   // largest nullspace a full JointTask behind another task can see
constexpr int PEND_SLOTS = 48;	   // deferred stores per robot: gravity N, MotionForceTask 12, JointTask k0

struct Fact {
	real L[N * N], dL[N];	// M = L L^T (lower), reciprocal diagonal
	real LB[N * N], dB[N];	// the same for the bounded inertia estimate (SingularityHandler.cpp:176-182)
};

DI real symat(const real* Q, int i, int j) { return i >= j ? Q[i * N + j] : Q[j * N + i]; }  // lower triangle kept

// Cholesky / triangular solves on the leading m x m block (m is the same for every robot: scalar branches)
template <int n>
DI void chol_m(const real* A, int m, real* L, real* dinv) {
	UNROLL for (int j = 0; j < n; j++) {
		if (j < m) {
			real s = A[j * n + j];
			UNROLL for (int k = 0; k < j; k++) s = fma(-L[j * n + k], L[j * n + k], s);
			const real r = rsqrt(s);
			dinv[j] = r;
			L[j * n + j] = s * r;
			UNROLL for (int i = j + 1; i < n; i++) {
				if (i < m) {
					real t = A[i * n + j];
					UNROLL for (int k = 0; k < j; k++) t = fma(-L[i * n + k], L[j * n + k], t);
					L[i * n + j] = t * r;
				}
			}
		}
	}
}
// x <- (L L^T)^-1 x; entries of L and x beyond m are zero
template <int n>
DI void spd_solve_m(const real* L, const real* dinv, int m, real* x) {
	UNROLL for (int i = 0; i < n; i++) {
		if (i < m) {
			real t = x[i];
			UNROLL for (int k = 0; k < i; k++) t = fma(-L[i * n + k], x[k], t);
			x[i] = t * dinv[i];
		}
	}
	UNROLL for (int i = n - 1; i >= 0; i--) {
		if (i < m) {
			real t = x[i];
			UNROLL for (int k = i + 1; k < n; k++) t = fma(-L[k * n + i], x[k], t);
			x[i] = t * dinv[i];
		}
	}
}

// certify_gram (sai2b_device.hpp) on the leading m x m block of a Gram matrix whose other rows and columns are
// zero: lambda_max <= ub := tr(G^8)^(1/8) <= m^(1/8) lambda_max, positive LDL^T pivots of G - rel2 ub I.
template <int n>
DI bool certify_gram_m(const real* G, int m, real abs2, real rel2) {
	real G2[n * n], G4[n * n];
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		if (i < m) {
			UNROLL for (int l = 0; l < n; l++) s = fma(G[i * n + l], G[j * n + l], s);
		}
		G2[i * n + j] = G2[j * n + i] = s;
	}
	real t8 = 0;
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		if (i < m) {
			real s = 0;
			UNROLL for (int l = 0; l < n; l++) s = fma(G2[i * n + l], G2[j * n + l], s);
			t8 = fma(s, (i == j) ? s : 2 * s, t8);
		}
	}
	const real ub = sqrt(sqrt(sqrt(t8)));
	bool ok = ub > 1.30 * abs2;	 // lambda_max >= ub / m^(1/8), 8^(1/8) = 1.2968
	const real c = rel2 * ub * (1.0 + 1e-9);
	const real floor_ = 1e-5 * c;
	real Lm[n * n], d[n];
	UNROLL for (int i = 0; i < n * n; i++) Lm[i] = 0;
	UNROLL for (int i = 0; i < n; i++) d[i] = 0;
	UNROLL for (int j = 0; j < n; j++) {
		if (j < m) {
			real s = G[j * n + j] - c;
			UNROLL for (int k = 0; k < j; k++) s = fma(-Lm[j * n + k] * Lm[j * n + k], d[k], s);
			d[j] = s;
			ok = ok && (s > floor_);
			const real inv = 1.0 / s;
			UNROLL for (int i = j + 1; i < n; i++) {
				if (i < m) {
					real t = G[i * n + j];
					UNROLL for (int k = 0; k < j; k++) t = fma(-Lm[i * n + k] * Lm[j * n + k], d[k], t);
					Lm[i * n + j] = t * inv;
				}
			}
		}
	}
	return ok;
}

// One level of the cascade in reduced coordinates. Jr: the task's m rows (Jr[c * N + i], rows >= m ignored).
// Task forces: Lambda va + Lambda_mod vf + vd with Lambda_mod by the decoupling type (SingularityHandler.cpp:
// 165-206, JointTask.cpp:240-270). Adds the level's torques to tau, takes its directions out of Q. Returns the
// certificate (true when do_cert is false: a first-level selection).
DI bool level(const Fact& f, const real* Jr, int m, bool first, bool last, bool do_cert, real abs2, real rel2, int decoupling,
			  bool has_va, const real* va, const real* vf, const real* vd, real* Q, real* tau) {
	real Y[MM * N], JP[MM * N];
	UNROLL for (int c = 0; c < MM; c++) {
		if (c < m) {
			real col[N];
			UNROLL for (int i = 0; i < N; i++) col[i] = Jr[c * N + i];
			solve_lower<N>(f.L, f.dL, col);
			if (first) {
				UNROLL for (int i = 0; i < N; i++) JP[c * N + i] = Jr[c * N + i];
			} else {
				real y[N];
				UNROLL for (int i = 0; i < N; i++) {
					real s = 0;
					UNROLL for (int j = 0; j < N; j++) s = fma(symat(Q, i, j), col[j], s);
					y[i] = s;
				}
				UNROLL for (int i = 0; i < N; i++) col[i] = y[i];
				UNROLL for (int i = 0; i < N; i++) {  // row c of Jp = (L Y)^T
					real s = 0;
					UNROLL for (int k = 0; k <= i; k++) s = fma(f.L[i * N + k], col[k], s);
					JP[c * N + i] = s;
				}
			}
			UNROLL for (int i = 0; i < N; i++) Y[c * N + i] = col[i];
		} else {
			UNROLL for (int i = 0; i < N; i++) Y[c * N + i] = JP[c * N + i] = 0;
		}
	}
	bool ok = true;
	if (do_cert) {
		real G[MM * MM];
		UNROLL for (int i = 0; i < MM; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			if (i < m) {
				UNROLL for (int l = 0; l < N; l++) s = fma(JP[i * N + l], JP[j * N + l], s);
			}
			G[i * MM + j] = G[j * MM + i] = s;
		}
		ok = certify_gram_m<MM>(G, m, abs2, rel2);
	}
	SAI2B_PHASE();
	// Y = Z R by modified Gram-Schmidt (Z overwrites Y); R upper triangular, rinv its reciprocal diagonal
	real R[MM * MM], rinv[MM];
	UNROLL for (int i = 0; i < MM * MM; i++) R[i] = 0;
	UNROLL for (int i = 0; i < MM; i++) rinv[i] = 0;
	UNROLL for (int j = 0; j < MM; j++) {
		if (j < m) {
			real nn = 0;
			UNROLL for (int i = 0; i < N; i++) nn = fma(Y[j * N + i], Y[j * N + i], nn);
			const real r = rsqrt(nn);
			rinv[j] = r;
			UNROLL for (int i = 0; i < N; i++) Y[j * N + i] *= r;
			UNROLL for (int k = j + 1; k < MM; k++) {
				if (k < m) {
					real s = 0;
					UNROLL for (int i = 0; i < N; i++) s = fma(Y[j * N + i], Y[k * N + i], s);
					R[j * MM + k] = s;
					UNROLL for (int i = 0; i < N; i++) Y[k * N + i] = fma(-s, Y[j * N + i], Y[k * N + i]);
				}
			}
		}
	}
	SAI2B_PHASE();
	// task forces in reduced coordinates
	const bool full = decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	real z[MM];
	UNROLL for (int c = 0; c < MM; c++) z[c] = 0;
	if (has_va || full) {  // Lambda x = R^-1 R^-T x
		UNROLL for (int c = 0; c < MM; c++) z[c] = (has_va ? va[c] : 0.0) + (full ? vf[c] : 0.0);
		UNROLL for (int j = 0; j < MM; j++) {
			if (j < m) {
				real t = z[j];
				UNROLL for (int i = 0; i < j; i++) t = fma(-R[i * MM + j], z[i], t);
				z[j] = t * rinv[j];
			}
		}
		UNROLL for (int j = MM - 1; j >= 0; j--) {
			if (j < m) {
				real t = z[j];
				UNROLL for (int k = j + 1; k < MM; k++) t = fma(-R[j * MM + k], z[k], t);
				z[j] = t * rinv[j];
			}
		}
	}
	if (decoupling == SAI2B_IMPEDANCE) {
		UNROLL for (int c = 0; c < MM; c++) z[c] += vf[c];
	} else if (decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
		// Lambda_mod = (Jp M_BIE^-1 Jp^T)^-1 = (YB^T YB)^-1, YB = LB^-1 Jp^T
		real AB[MM * MM], LA[MM * MM], dA[MM], YB[MM * N];
		UNROLL for (int i = 0; i < MM * MM; i++) LA[i] = 0;
		UNROLL for (int c = 0; c < MM; c++) {
			real col[N];
			UNROLL for (int i = 0; i < N; i++) col[i] = JP[c * N + i];
			if (c < m) solve_lower<N>(f.LB, f.dB, col);
			UNROLL for (int i = 0; i < N; i++) YB[c * N + i] = col[i];
		}
		UNROLL for (int i = 0; i < MM; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			if (i < m) {
				UNROLL for (int l = 0; l < N; l++) s = fma(YB[i * N + l], YB[j * N + l], s);
			}
			AB[i * MM + j] = s;
		}
		chol_m<MM>(AB, m, LA, dA);
		real y[MM];
		UNROLL for (int c = 0; c < MM; c++) y[c] = vf[c];
		spd_solve_m<MM>(LA, dA, m, y);
		UNROLL for (int c = 0; c < MM; c++) z[c] += y[c];
	}
	UNROLL for (int c = 0; c < MM; c++) z[c] += vd[c];
	UNROLL for (int c = 0; c < MM; c++) {
		if (c < m) {
			UNROLL for (int i = 0; i < N; i++) tau[i] = fma(JP[c * N + i], z[c], tau[i]);  // Jp^T F
		}
	}
	if (!last) {
		UNROLL for (int c = 0; c < MM; c++) {
			if (c < m) {
				UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-Y[c * N + i], Y[c * N + j], Q[i * N + j]);
			}
		}
	}
	return ok;
}

// A full JointTask behind other tasks: Jp = N_prec = L^-T Q L^T with rank d = n - (rows of the certified tasks
// above) >= 1 (JointTask.cpp:218-283; the reference's 1e-3 range rule sees singular values >= 1 and ~1e-16).
// va: goal acceleration minus the compensation of the tasks above, vf: the PD(+I) unit torques.
DI void full_joint_task_behind(const Fact& f, int d, int decoupling, const real* va, const real* vf, real* Q, real* tau) {
	// basis of range(Q): pivoted Cholesky of the projector, Q = C C^T with orthonormal columns
	real C[DM * N];
	UNROLL for (int i = 0; i < DM * N; i++) C[i] = 0;
	UNROLL for (int s = 0; s < DM; s++) {
		if (s < d) {
			real best = -1.0;
			int ks = 0;
			UNROLL for (int i = 0; i < N; i++) {
				const bool take = Q[i * N + i] > best;
				best = take ? Q[i * N + i] : best;
				ks = take ? i : ks;
			}
			const real rn = rsqrt(best);
			UNROLL for (int i = 0; i < N; i++) {
				real v = symat(Q, i, 0);
				UNROLL for (int j = 1; j < N; j++) v = (ks == j) ? symat(Q, i, j) : v;
				C[s * N + i] = v * rn;
			}
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-C[s * N + i], C[s * N + j], Q[i * N + j]);
		}
	}
	SAI2B_PHASE();
	real A[DM * N], K[DM * DM], LK[DM * DM], dK[DM];
	UNROLL for (int i = 0; i < DM * DM; i++) LK[i] = 0;
	UNROLL for (int s = 0; s < DM; s++) {
		real col[N];
		UNROLL for (int i = 0; i < N; i++) col[i] = C[s * N + i];
		if (s < d) solve_lower_t<N>(f.L, f.dL, col);  // A = L^-T C
		UNROLL for (int i = 0; i < N; i++) A[s * N + i] = col[i];
	}
	UNROLL for (int i = 0; i < DM; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		if (i < d) {
			UNROLL for (int l = 0; l < N; l++) s = fma(A[i * N + l], A[j * N + l], s);
		}
		K[i * DM + j] = s;
	}
	chol_m<DM>(K, d, LK, dK);
	const bool full = decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	real h[DM];
	UNROLL for (int s = 0; s < DM; s++) {  // A^T (va (+ vf))
		real a = 0;
		UNROLL for (int i = 0; i < N; i++) a = fma(A[s * N + i], va[i] + (full ? vf[i] : 0.0), a);
		h[s] = a;
	}
	spd_solve_m<DM>(LK, dK, d, h);
	if (decoupling == SAI2B_IMPEDANCE) {  // R R^T f  ->  C^T L^-1 f
		real u[N];
		UNROLL for (int i = 0; i < N; i++) u[i] = vf[i];
		solve_lower<N>(f.L, f.dL, u);
		UNROLL for (int s = 0; s < DM; s++) {
			real a = 0;
			UNROLL for (int i = 0; i < N; i++) a = fma(C[s * N + i], u[i], a);
			h[s] += a;
		}
	} else if (decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {	 // Kb^-1 K^-1 A^T f, Kb = (LB^-1 L C)^T (LB^-1 L C)
		real g[DM], Kb[DM * DM], LKb[DM * DM], dKb[DM], Wb[DM * N];
		UNROLL for (int i = 0; i < DM * DM; i++) LKb[i] = 0;
		UNROLL for (int s = 0; s < DM; s++) {
			real a = 0;
			UNROLL for (int i = 0; i < N; i++) a = fma(A[s * N + i], vf[i], a);
			g[s] = a;
		}
		spd_solve_m<DM>(LK, dK, d, g);
		UNROLL for (int s = 0; s < DM; s++) {
			real col[N];
			UNROLL for (int i = 0; i < N; i++) {
				real a = 0;
				UNROLL for (int k = 0; k <= i; k++) a = fma(f.L[i * N + k], C[s * N + k], a);
				col[i] = a;
			}
			if (s < d) solve_lower<N>(f.LB, f.dB, col);
			UNROLL for (int i = 0; i < N; i++) Wb[s * N + i] = col[i];
		}
		UNROLL for (int i = 0; i < DM; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			if (i < d) {
				UNROLL for (int l = 0; l < N; l++) s = fma(Wb[i * N + l], Wb[j * N + l], s);
			}
			Kb[i * DM + j] = s;
		}
		chol_m<DM>(Kb, d, LKb, dKb);
		spd_solve_m<DM>(LKb, dKb, d, g);
		UNROLL for (int s = 0; s < DM; s++) h[s] += g[s];
	}
	real w[N];
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int s = 0; s < DM; s++) a = fma(C[s * N + i], h[s], a);
		w[i] = a;
	}
	UNROLL for (int i = 0; i < N; i++) {  // Jp^T x = L Q L^-1 x = L C (...)
		real a = 0;
		UNROLL for (int k = 0; k <= i; k++) a = fma(f.L[i * N + k], w[k], a);
		tau[i] += a;
	}
}

// y = M x = L (L^T x)
DI void mul_llt(const real* L, const real* x, real* y) {
	real u[N];
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int k = i; k < N; k++) a = fma(L[k * N + i], x[k], a);
		u[i] = a;
	}
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int k = 0; k <= i; k++) a = fma(L[i * N + k], u[k], a);
		y[i] = a;
	}
}

// The whole tick of one robot. pend: this lane's column of the deferred-store buffer (stride 64 doubles).
// Returns whether every task was certified (and no MotionForceTask is inside / leaving a singular region): only
// then may the caller flush the deferred stores and write tau.
DI bool tick(const DevParams& P, int B, int b, bool with_comp, real* pend, real* tau) {
	Fact f;
	bool ok = true;
	int np = N;	 // deferred-store slots used so far: 0..N-1 hold the gravity torques
	{
		// Sai2Model::updateModel(): kinematics, M (CRBA) and its factor (examples/05-using_robot_controller.cpp:143-145)
		real q[N];
		UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
		Frames F;
		fk(P.model, q, F);
		real M[N * N];
		mass_matrix(P.model, F, M);
		if (P.gravity_comp) {
			real g[N];
			gravity_vector(P.model, F, g);
			UNROLL for (int i = 0; i < N; i++) pend[i * 64] = g[i];
		} else {
			UNROLL for (int i = 0; i < N; i++) pend[i * 64] = 0.0;
		}
		SAI2B_PHASE();
		chol<N>(M, f.L, f.dL);
		// bounded inertia estimate, shared by the tasks that ask for it (host checks the thresholds agree)
		bool any_bie = false;
		real thr = 0;
		for (int t = 0; t < P.n_tasks; t++)
			if (P.task[t].decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
				any_bie = true;
				thr = P.task[t].bie_threshold;
			}
		if (any_bie) {
			UNROLL for (int i = 0; i < N; i++) M[i * N + i] = fmax(M[i * N + i], thr);
			chol<N>(M, f.LB, f.dB);
		} else {
			UNROLL for (int i = 0; i < N * N; i++) f.LB[i] = f.L[i];
			UNROLL for (int i = 0; i < N; i++) f.dB[i] = f.dL[i];
		}
	}
	SAI2B_PHASE();
	real Q[N * N];
	UNROLL for (int i = 0; i < N * N; i++) Q[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
	UNROLL for (int i = 0; i < N; i++) tau[i] = 0;
	int wrows = 0;	// rows of the certified tasks so far (batch-uniform)
#pragma unroll 1
	for (int ti = 0; ti < P.n_tasks; ti++) {
		const DevTask& t = P.task[ti];
		const bool first = (ti == 0), last = (ti == P.n_tasks - 1);
		real Jr[MM * N], va[MM], vf[MM], vd[MM];
		UNROLL for (int i = 0; i < MM; i++) va[i] = vf[i] = vd[i] = 0;
		int m;
		bool do_cert, has_va;
		real abs2, rel2;
		if (t.type == SAI2B_MOTION_FORCE_TASK) {
			// MotionForceTask::updateTaskModel / computeTorques (MotionForceTask.cpp:247-509) in the fully
			// non-singular branch of the SingularityHandler (SingularityHandler.cpp:100-141,307-309)
			ok = ok && (ldi(t.istate, IS_NTYPES, B, b) == 0);
			real Jw[6 * N], x[3], R[9], Fu[6], Ff[6];
			{
				real q[N];
				UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
				Frames F;
				fk(P.model, q, F);
				frame_pose(t, F, x, R);
				jacobian(P.model, t, F, x, Jw);
			}
			{
				real dq[N], vw0[6], vw[6];
				UNROLL for (int i = 0; i < N; i++) dq[i] = ld(P.dq, i, B, b);
				mv<6, N>(Jw, dq, vw0);
				if (t.full_projection) {
					UNROLL for (int i = 0; i < 6; i++) vw[i] = vw0[i];
				} else {
					mv<6, 6>(t.P, vw0, vw);
				}
				MftIn in;
				mft_load(t, B, b, in);
				mft_law_vw(t, vw, vw + 3, x, R, in, Fu, Ff);
				// mft_store_integrators, deferred
				UNROLL for (int k = 0; k < 12; k++) pend[(np + k) * 64] = in.integ[k];
				np += 12;
			}
			m = t.rank;
			if (t.full_projection || t.p_lead < 6) {  // range(P) = the leading coordinates
				UNROLL for (int c = 0; c < MM; c++) UNROLL for (int i = 0; i < N; i++) Jr[c * N + i] = (c < 6) ? Jw[(c < 6 ? c : 0) * N + i] : 0.0;
				UNROLL for (int c = 0; c < 6; c++) {
					vf[c] = Fu[c];
					vd[c] = Ff[c];
				}
			} else {  // rows PU^T J, forces PU^T F
				UNROLL for (int c = 0; c < MM; c++) UNROLL for (int i = 0; i < N; i++) {
					real s = 0;
					if (c < 6) {
						UNROLL for (int k = 0; k < 6; k++) s = fma(t.PU[k * 6 + (c < 6 ? c : 0)], Jw[k * N + i], s);
					}
					Jr[c * N + i] = s;
				}
				UNROLL for (int c = 0; c < 6; c++) {
					real a = 0, bb = 0;
					UNROLL for (int k = 0; k < 6; k++) {
						a = fma(t.PU[k * 6 + c], Fu[k], a);
						bb = fma(t.PU[k * 6 + c], Ff[k], bb);
					}
					vf[c] = a;
					vd[c] = bb;
				}
			}
			UNROLL for (int c = 0; c < MM; c++) {  // coordinates outside range(P) carry no force
				if (c >= m) vf[c] = vd[c] = 0;
			}
			do_cert = true;
			has_va = false;
			abs2 = t.s_abs_tol * t.s_abs_tol;
			rel2 = t.s_max * t.s_max;
		} else {
			// JointTask::updateTaskModel / computeTorques (JointTask.cpp:218-356)
			const int k0 = t.k0;
			real cur[N], vel[N];
			{
				real q[N], dq[N];
				UNROLL for (int i = 0; i < N; i++) {
					q[i] = ld(P.q, i, B, b);
					dq[i] = ld(P.dq, i, B, b);
				}
				if (t.full_selection) {
					UNROLL for (int i = 0; i < N; i++) cur[i] = q[i], vel[i] = dq[i];
				} else {
					mv<N, N>(t.S, q, cur);
					mv<N, N>(t.S, dq, vel);
				}
			}
			real comp[N];
			UNROLL for (int i = 0; i < N; i++) comp[i] = 0;
			if (with_comp && !first) {	// JointTask.cpp:285-292: S M^-1 tau_prec
				real u[N];
				UNROLL for (int i = 0; i < N; i++) u[i] = tau[i];
				solve_lower<N>(f.L, f.dL, u);
				solve_lower_t<N>(f.L, f.dL, u);
				if (t.full_selection) {
					UNROLL for (int i = 0; i < N; i++) comp[i] = u[i];
				} else {
					mv<N, N>(t.S, u, comp);
				}
			}
			UNROLL for (int i = 0; i < N; i++) {
				if (i < k0) {  // PD(+I) law of task coordinate i (JointTask.cpp:299-345)
					const real* G = t.law_goals;
					const real qd = ld(G, i, B, b), dqd = ld(G, k0 + i, B, b), ddq_d = ld(G, 2 * k0 + i, B, b);
					const real integ = fma(cur[i] - qd, t.dt, ld(t.state, i, B, b));
					pend[(np + i) * 64] = integ;
					real fi;
					if (t.use_vsat) {
						const real kvi = gain_pinv(t.kv[i]);
						real dv = -t.kp[i] * kvi * (cur[i] - qd) - t.ki[i] * kvi * integ;
						dv = fmin(fmax(dv, -t.vsat[i]), t.vsat[i]);
						fi = -t.kv[i] * (vel[i] - dv);
					} else {
						fi = -t.kp[i] * (cur[i] - qd) - t.kv[i] * (vel[i] - dqd) - t.ki[i] * integ;
					}
					vf[i] = fi;
					va[i] = ddq_d - comp[i];
				}
			}
			np += k0;
			if (t.full_selection) {
				if (first) {
					// Jp = I: M_partial = M, and with the bounded estimate M_BIE (JointTask.cpp:247-265)
					real y[N], x1[N];
					const bool full = t.decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
					UNROLL for (int i = 0; i < N; i++) x1[i] = va[i] + (full ? vf[i] : 0.0);
					mul_llt(f.L, x1, y);
					UNROLL for (int i = 0; i < N; i++) tau[i] += y[i];
					if (t.decoupling == SAI2B_IMPEDANCE) {
						UNROLL for (int i = 0; i < N; i++) tau[i] += vf[i];
					} else if (!full) {
						UNROLL for (int i = 0; i < N; i++) x1[i] = vf[i];
						mul_llt(f.LB, x1, y);
						UNROLL for (int i = 0; i < N; i++) tau[i] += y[i];
					}
				} else if (wrows < N) {
					full_joint_task_behind(f, N - wrows, t.decoupling, va, vf, Q, tau);
				}
				// a full JointTask leaves nothing to the tasks below it
				UNROLL for (int i = 0; i < N * N; i++) Q[i] = 0;
				wrows = N;
				continue;
			}
			UNROLL for (int c = 0; c < MM; c++) UNROLL for (int i = 0; i < N; i++) Jr[c * N + i] = (c < N) ? t.S[(c < N ? c : 0) * N + i] : 0.0;
			m = k0;
			do_cert = !first;  // a first-level selection has full row rank by construction (validated on the host)
			has_va = true;
			abs2 = 1e-6;  // s_0 >= 1e-3 and s_i / s_0 >= 1e-3: the range rule keeps every row
			rel2 = 1e-6;
		}
		if (wrows + m > N) ok = false;	// more task rows than joints left: never full rank
		const bool c_ok = level(f, Jr, m, first, last, do_cert, abs2, rel2, t.decoupling, has_va, va, vf, vd, Q, tau);
		ok = ok && c_ok;
		wrows += m;
		if (wrows > N) wrows = N;
	}
	return ok;
}

// write the deferred state of a robot that finished here (same task walk as tick())
DI void flush(const DevParams& P, int B, int b, const real* pend) {
	int np = N;
#pragma unroll 1
	for (int ti = 0; ti < P.n_tasks; ti++) {
		const DevTask& t = P.task[ti];
		if (t.type == SAI2B_MOTION_FORCE_TASK) {
			for (int k = 0; k < 6; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
			if (t.cl_force)
				for (int k = 6; k < 9; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
			if (t.cl_moment)
				for (int k = 9; k < 12; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
			np += 12;
		} else {
			for (int k = 0; k < t.k0; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
			np += t.k0;
		}
	}
}

}  // namespace cert
}  // namespace sai2b
