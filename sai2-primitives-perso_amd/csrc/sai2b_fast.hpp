// sai2b_fast.hpp — SVD-free fast path of the tick for the hierarchies
//     [full 6-DOF MotionForceTask]                        (BASELINE config 2)
//     [full 6-DOF MotionForceTask, full JointTask]        (BASELINE configs 3 and 5)
// taken by a robot only when it carries a certificate that the SingularityHandler's decision is "fully
// non-singular" (SingularityHandler.cpp:100-141). In that
// branch the reference's result does not depend on the singular vectors at all:
//     tau_mft = J^T ( (J Mb^-1 J^T)^-1 F_unit + F_force )            (U_ns cancels, :307-309)
//     N       = I - M^-1 J^T (J M^-1 J^T)^-1 J  =  L^-T (w w^T) L^T   with M = L L^T, w ⟂ range(L^-1 J^T)
// so with a = L^-T w, b = L w the nullspace projector is the rank-one matrix a b^T, and the whole
// second-level JointTask (JointTask.cpp:218-356) collapses to a handful of dot products:
//     Jp = N,  range(Jp) = span(a),  R M_partial R^T = a a^T / |a|^4,
//     R (R^T Jp Mb^-1 Jp^T R)^-1 R^T = a a^T / (beta |a|^4),   beta = b^T Mb^-1 b.
// The singular values themselves are needed only for the branch decision, which is replaced by a
// certificate on G = J J^T:  lambda_max(G) <= ub := tr(G^8)^(1/8) <= 6^(1/8) lambda_max(G), and
// G - s_max^2 ub I positive definite  =>  s_5/s_0 >= s_max (and every s_i/s_0 with it).
// A robot the certificate cannot vouch for (s_5/s_0 below ~0.067, or anything singular) touches no state here
// and is appended to a work list (one atomic per wavefront that has such robots); the generic kernel launched
// right behind runs over that compacted list, so the results are the reference's in all cases.
#pragma once
#include "sai2b_device.hpp"

namespace sai2b {

// Cholesky factor of an SPD n x n matrix: lower L (row-major, upper part untouched) and the
// reciprocals of its diagonal.
template <int n>
DI void chol(const real* A, real* L, real* dinv) {
	UNROLL for (int j = 0; j < n; j++) {
		real s = A[j * n + j];
		UNROLL for (int k = 0; k < j; k++) s = fma(-L[j * n + k], L[j * n + k], s);
		real r = rsqrt(s);
		dinv[j] = r;
		L[j * n + j] = s * r;
		UNROLL for (int i = j + 1; i < n; i++) {
			real t = A[i * n + j];
			UNROLL for (int k = 0; k < j; k++) t = fma(-L[i * n + k], L[j * n + k], t);
			L[i * n + j] = t * r;
		}
	}
}
// x <- L^-1 x (forward substitution)
template <int n>
DI void solve_lower(const real* L, const real* dinv, real* x) {
	UNROLL for (int i = 0; i < n; i++) {
		real t = x[i];
		UNROLL for (int k = 0; k < i; k++) t = fma(-L[i * n + k], x[k], t);
		x[i] = t * dinv[i];
	}
}
// x <- L^-T x (back substitution)
template <int n>
DI void solve_lower_t(const real* L, const real* dinv, real* x) {
	UNROLL for (int i = n - 1; i >= 0; i--) {
		real t = x[i];
		UNROLL for (int k = i + 1; k < n; k++) t = fma(-L[k * n + i], x[k], t);
		x[i] = t * dinv[i];
	}
}

// Certificate for "s_0 >= s_abs_tol and s_5 / s_0 >= s_max" on the 6 x 7 Jacobian (see header).
DI bool certify_nonsingular(const real* J, real s_abs_tol, real s_max) {
	real G[36];
	mm_nt_sym<6, N>(J, J, G);
	return certify_gram<6>(G, nullptr, s_abs_tol * s_abs_tol, s_max * s_max);
}

// Inputs of the second-level JointTask law after the early part (fast_jt_early): the PD(+I) unit
// torques f, the goal accelerations, and the advanced integrators (stored once the wavefront is
// committed to the fast path).
struct JtEarly {
	real f[N], ddq[N], integ[N];
};
DI void fast_jt_early(const DevTask& t1, const RobotCtx& rc, int B, int b, JtEarly& e) {
	const real* G = t1.law_goals;
	UNROLL for (int i = 0; i < N; i++) {  // JointTask.cpp:299-345, S = I
		const real qd = ld(G, i, B, b), dqd = ld(G, N + i, B, b);
		e.ddq[i] = ld(G, 2 * N + i, B, b);
		const real integ = fma(rc.q[i] - qd, t1.dt, ld(t1.state, i, B, b));
		e.integ[i] = integ;
		if (t1.use_vsat) {
			const real kvi = gain_pinv(t1.kv[i]);
			real dv = -t1.kp[i] * kvi * (rc.q[i] - qd) - t1.ki[i] * kvi * integ;
			dv = fmin(fmax(dv, -t1.vsat[i]), t1.vsat[i]);
			e.f[i] = -t1.kv[i] * (rc.dq[i] - dv);
		} else {
			e.f[i] = -t1.kp[i] * (rc.q[i] - qd) - t1.kv[i] * (rc.dq[i] - dqd) - t1.ki[i] * integ;
		}
	}
}

// Scheduling fence between phases (and a marker in the ISA for per-phase inspection): the kernel is
// one huge basic block, and without fences the scheduler interleaves phases and inflates the live set.
#define SAI2B_PHASE() do { __builtin_amdgcn_sched_barrier(0); asm volatile("; SAI2B_PHASE_MARK"); __builtin_amdgcn_sched_barrier(0); } while (0)

// The Cholesky part of the fast tick. J and M are the Jacobian and mass matrix at rc.q, Fu/Ff the task
// forces of the MotionForceTask law, jt the JointTask law; HAS_JT selects the 2-level form. Ordered to
// keep few matrices alive at once: bounded-inertia side -> Y = L^-1 J^T -> nullspace vectors -> torques.
template <bool HAS_JT>
DI void fast_tick(const DevParams& P, const real* J, const real* M, const real* Fu, const real* Ff, int B, int b,
				  bool with_comp, const JtEarly& jt, real* tau) {
	const DevTask& t0 = P.task[0];

	// bounded inertia estimate shared by the tasks that ask for it (host checks thresholds agree)
	const bool mft_bie = t0.decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES;
	const bool mft_full = t0.decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	bool any_bie = mft_bie;
	real thr = t0.bie_threshold;
	if (HAS_JT && P.task[1].decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
		any_bie = true;
		thr = P.task[1].bie_threshold;
	}
	// Phase A: bounded-inertia side (LB, YB = LB^-1 J^T, AB = YB^T YB, z = AB^-1 F_unit); YB dies here
	real LB[N * N], dB[N], z[6];
	UNROLL for (int i = 0; i < 6; i++) z[i] = Fu[i];
	if (any_bie) {
		real MB[N * N];
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) MB[i * N + j] = M[i * N + j];
		UNROLL for (int i = 0; i < N; i++) MB[i * N + i] = fmax(MB[i * N + i], thr);
		chol<N>(MB, LB, dB);
	}
	if (mft_bie) {	// Lambda_mod = (J Mb^-1 J^T)^-1
		real YB[N * 6], AB[36], LAB[36], dAB[6];
		UNROLL for (int c = 0; c < 6; c++) {
			real colb[N];
			UNROLL for (int i = 0; i < N; i++) colb[i] = J[c * N + i];
			solve_lower<N>(LB, dB, colb);
			UNROLL for (int i = 0; i < N; i++) YB[i * 6 + c] = colb[i];
		}
		UNROLL for (int i = 0; i < 6; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			UNROLL for (int l = 0; l < N; l++) s = fma(YB[l * 6 + i], YB[l * 6 + j], s);
			AB[i * 6 + j] = s;
			AB[j * 6 + i] = s;
		}
		chol<6>(AB, LAB, dAB);
		solve_lower<6>(LAB, dAB, z);
		solve_lower_t<6>(LAB, dAB, z);
	}
	SAI2B_PHASE();
	// Phase B: M = L L^T, Y = L^-1 J^T; from here J and M are dead: tau_mft = J^T z = L (Y z)
	real L[N * N], dL[N], Y[N * 6];
	chol<N>(M, L, dL);
	UNROLL for (int c = 0; c < 6; c++) {
		real col[N];
		UNROLL for (int i = 0; i < N; i++) col[i] = J[c * N + i];
		solve_lower<N>(L, dL, col);
		UNROLL for (int i = 0; i < N; i++) Y[i * 6 + c] = col[i];
	}
	SAI2B_PHASE();
	// ---- A = J M^-1 J^T = Y^T Y and its factor: Lambda for FULL decoupling, nullspace for the JT
	real a[N], bb[N], na2 = 0;
	if (HAS_JT || mft_full) {
		real A[36], LA[36], dA[6];
		UNROLL for (int i = 0; i < 6; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			UNROLL for (int l = 0; l < N; l++) s = fma(Y[l * 6 + i], Y[l * 6 + j], s);
			A[i * 6 + j] = s;
			A[j * 6 + i] = s;
		}
		chol<6>(A, LA, dA);
		if (mft_full) {	 // Lambda_mod = Lambda = A^-1
			solve_lower<6>(LA, dA, z);
			solve_lower_t<6>(LA, dA, z);
		}
		if (HAS_JT) {
			// w = unit vector orthogonal to range(Y): the row k of the complementary projector
			// I - Y A^-1 Y^T with the largest diagonal, normalised
			real tk[6], best = -1;
			int ks = 0;
			UNROLL for (int c = 0; c < 6; c++) tk[c] = 0;
			UNROLL for (int i = 0; i < N; i++) {
				real t[6];
				UNROLL for (int c = 0; c < 6; c++) t[c] = Y[i * 6 + c];
				solve_lower<6>(LA, dA, t);
				real pd = 1.0;
				UNROLL for (int c = 0; c < 6; c++) pd = fma(-t[c], t[c], pd);
				const bool take = pd > best;
				best = take ? pd : best;
				ks = take ? i : ks;
				UNROLL for (int c = 0; c < 6; c++) tk[c] = take ? t[c] : tk[c];
			}
			solve_lower_t<6>(LA, dA, tk);  // A^-1 y_k
			real w[N];
			const real wn = rsqrt(best);
			UNROLL for (int i = 0; i < N; i++) {
				real s = (ks == i) ? 1.0 : 0.0;
				UNROLL for (int c = 0; c < 6; c++) s = fma(-Y[i * 6 + c], tk[c], s);
				w[i] = s * wn;
			}
			{  // one re-orthogonalisation pass: w <- w - Y A^-1 Y^T w, renormalise
				real cw[6];
				UNROLL for (int c = 0; c < 6; c++) {
					real s = 0;
					UNROLL for (int i = 0; i < N; i++) s = fma(Y[i * 6 + c], w[i], s);
					cw[c] = s;
				}
				solve_lower<6>(LA, dA, cw);
				solve_lower_t<6>(LA, dA, cw);
				real nn = 0;
				UNROLL for (int i = 0; i < N; i++) {
					real s = w[i];
					UNROLL for (int c = 0; c < 6; c++) s = fma(-Y[i * 6 + c], cw[c], s);
					w[i] = s;
					nn = fma(s, s, nn);
				}
				const real rn = rsqrt(nn);
				UNROLL for (int i = 0; i < N; i++) w[i] *= rn;
			}
			UNROLL for (int i = 0; i < N; i++) a[i] = w[i];
			solve_lower_t<N>(L, dL, a);	 // a = L^-T w
			UNROLL for (int i = 0; i < N; i++) {
				real s = 0;
				UNROLL for (int k = 0; k <= i; k++) s = fma(L[i * N + k], w[k], s);
				bb[i] = s;	// b = L w
			}
			UNROLL for (int i = 0; i < N; i++) na2 = fma(a[i], a[i], na2);
		}
	}
	SAI2B_PHASE();
	// ---- MotionForceTask torques J^T (Lambda_mod F_unit + F_force) = L (Y z)   (SingularityHandler.cpp:307-309)
	UNROLL for (int i = 0; i < 6; i++) z[i] += Ff[i];
	real yz[N], tau_mft[N];
	mv<N, 6>(Y, z, yz);
	UNROLL for (int i = 0; i < N; i++) {
		real s = 0;
		UNROLL for (int k = 0; k <= i; k++) s = fma(L[i * N + k], yz[k], s);
		tau_mft[i] = s;
	}
	UNROLL for (int i = 0; i < N; i++) tau[i] = tau_mft[i];
	if (!HAS_JT) return;

	// ---- JointTask: the law was evaluated up front (fast_jt_early); project it
	const DevTask& t1 = P.task[1];
	real af = 0, aacc = 0;
	UNROLL for (int i = 0; i < N; i++) {
		af = fma(a[i], jt.f[i], af);
		aacc = fma(a[i], jt.ddq[i], aacc);
	}
	if (with_comp) {  // JointTask.cpp:285-292: - Jp^T R M_partial R^T S M^-1 tau_prec;  M^-1 tau_mft = L^-T (Y z)
		real u[N];
		UNROLL for (int i = 0; i < N; i++) u[i] = yz[i];
		solve_lower_t<N>(L, dL, u);
		UNROLL for (int i = 0; i < N; i++) aacc = fma(-a[i], u[i], aacc);
	}
	real coef = aacc / na2;	 // a^T (a a^T / |a|^4) v = (a.v) / |a|^2
	if (t1.decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING) {
		coef += af / na2;
	} else if (t1.decoupling == SAI2B_IMPEDANCE) {
		coef += af;	 // a^T (a a^T / |a|^2) f
	} else {
		real y[N];
		UNROLL for (int i = 0; i < N; i++) y[i] = bb[i];
		solve_lower<N>(LB, dB, y);
		real beta = 0;
		UNROLL for (int i = 0; i < N; i++) beta = fma(y[i], y[i], beta);
		coef += af / (beta * na2);
	}
	UNROLL for (int i = 0; i < N; i++) tau[i] = fma(bb[i], coef, tau[i]);  // Jp^T x = b (a^T x)
}

}  // namespace sai2b
