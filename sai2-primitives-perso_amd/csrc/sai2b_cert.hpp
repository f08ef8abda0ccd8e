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
// (PEND_SLOTS doubles per lane). Partial tasks of up to 6 rows (the kernel's instantiations: MCAP = 3 and 6);
// the host routes anything else to the generic kernel (sai2b_host.cpp: cert_kind).
#pragma once
#include "sai2b_device.hpp"
#include "sai2b_fast.hpp"

namespace sai2b {
namespace cert {

// -DSAI2B_CERT_STAMP (diagnostic build, scripts/micro/cert_stamps.py): lane 0 of workgroup 0 records (mark, cycle
// counter) at every CSTAMP into g_cstamps: where one wavefront's cycles go
#ifdef SAI2B_CERT_STAMP
__device__ unsigned long long g_cstamps[1024];
__device__ int g_cstamp_n;
#define CSTAMP(id)                                                        \
	do {                                                                  \
		__builtin_amdgcn_sched_barrier(0);                                \
		if (blockIdx.x == 0 && threadIdx.x == 0) {                        \
			const int k_ = g_cstamp_n;                                    \
			if (k_ < 500) {                                               \
				g_cstamps[2 * k_] = (unsigned long long)(id);             \
				g_cstamps[2 * k_ + 1] = __builtin_readcyclecounter();     \
				g_cstamp_n = k_ + 1;                                      \
			}                                                             \
		}                                                                 \
		__builtin_amdgcn_sched_barrier(0);                                \
	} while (0)
// inside a divergent branch: the first active lane of workgroup 0 records
#define CSTAMP_ANY(id)                                                                              \
	do {                                                                                            \
		__builtin_amdgcn_sched_barrier(0);                                                          \
		if (blockIdx.x == 0 && (int)threadIdx.x == __ffsll((unsigned long long)__ballot(1)) - 1) { \
			const int k_ = g_cstamp_n;                                                              \
			if (k_ < 500) {                                                                         \
				g_cstamps[2 * k_] = (unsigned long long)(id);                                       \
				g_cstamps[2 * k_ + 1] = __builtin_readcyclecounter();                               \
				g_cstamp_n = k_ + 1;                                                                \
			}                                                                                       \
		}                                                                                           \
		__builtin_amdgcn_sched_barrier(0);                                                          \
	} while (0)
#else
#define CSTAMP(id) do { } while (0)
#define CSTAMP_ANY(id) asm volatile("; SAI2B_SING_" #id)
#endif

constexpr int DM = N - 1;		   // largest nullspace a full JointTask behind another task can see
constexpr int PEND_SLOTS = 36;	   // deferred stores per robot: gravity N, MotionForceTask 12, JointTask k0
constexpr int LB_SLOTS = N * (N + 1) / 2 + N;  // factor of the bounded inertia estimate, parked in LDS between its uses
constexpr int LDS_SLOTS = PEND_SLOTS + LB_SLOTS;  // doubles per lane: 4 workgroups of 64 lanes fit the 160 KB of a CU
static_assert(LDS_SLOTS * 64 * 8 * 4 <= 160 * 1024, "four wavefronts per CU");
// what is left of a lane's column (7 joints: 9 doubles) holds the control frame's pose for the in-lane singular branch
#ifdef SAI2B_SING_NO_LDS_POSE  // A/B (scripts/micro/cert_variants.sh)
constexpr int POSE_SLOTS = 0;
#else
constexpr int POSE_SLOTS = ((LDS_SLOTS + 9) * 64 * 8 * 4 <= 160 * 1024) ? 9 : 0;
#endif

struct Fact {
	real L[N * N], dL[N];	// M = L L^T (lower), reciprocal diagonal
	const real* lb;			// this lane's column of the LDS copy of LB, dB: M_BIE = LB LB^T (SingularityHandler.cpp:176-182)
};
// the factor of the bounded inertia estimate lives in LDS (stride 64 doubles) and comes into registers where a
// level needs it: it is the one large object nothing needs most of the time
DI void store_lb(real* lb, const real* LB, const real* dB) {
	UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) lb[(i * (i + 1) / 2 + j) * 64] = LB[i * N + j];
	UNROLL for (int i = 0; i < N; i++) lb[(N * (N + 1) / 2 + i) * 64] = dB[i];
}
DI void load_lb(const real* lb, real* LB, real* dB) {
	UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) LB[i * N + j] = lb[(i * (i + 1) / 2 + j) * 64];
	UNROLL for (int i = 0; i < N; i++) dB[i] = lb[(N * (N + 1) / 2 + i) * 64];
}

DI real symat(const real* Q, int i, int j) { return i >= j ? Q[i * N + j] : Q[j * N + i]; }  // lower triangle kept

// certify_gram of sai2b_device.hpp with the live set of a triangle: G (lower triangle, row-major n x n storage),
// lambda_max <= ub := tr(G^8)^(1/8) <= n^(1/8) lambda_max and positive LDL^T pivots of G - rel2 ub I
template <int n>
DI real lower(const real* A, int i, int j) { return i >= j ? A[i * n + j] : A[j * n + i]; }
template <int n>
DI bool certify_gram_lower(const real* G, real abs2, real rel2) {
	real G2[n * n];
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < n; l++) s = fma(lower<n>(G, i, l), lower<n>(G, j, l), s);
		G2[i * n + j] = s;
	}
	real t8 = 0;
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < n; l++) s = fma(lower<n>(G2, i, l), lower<n>(G2, j, l), s);
		t8 = fma(s, (i == j) ? s : 2 * s, t8);
	}
	const real ub = sqrt(sqrt(sqrt(t8)));
	bool ok = ub > 1.30 * abs2;	 // lambda_max >= ub / n^(1/8), 8^(1/8) = 1.2968
	const real c = rel2 * ub * (1.0 + 1e-9);
	const real floor_ = 1e-5 * c;
	real Lm[n * n], d[n];
	UNROLL for (int j = 0; j < n; j++) {
		real s = G[j * n + j] - c;
		UNROLL for (int k = 0; k < j; k++) s = fma(-Lm[j * n + k] * Lm[j * n + k], d[k], s);
		d[j] = s;
		ok = ok && (s > floor_);
		const real inv = 1.0 / s;
		UNROLL for (int i = j + 1; i < n; i++) {
			real t = G[i * n + j];
			UNROLL for (int k = 0; k < j; k++) t = fma(-Lm[i * n + k] * Lm[j * n + k], d[k], t);
			Lm[i * n + j] = t * inv;
		}
	}
	return ok;
}

// ---- The singular branch of the SingularityHandler in whitened coordinates (round 3) --------------------------------
// A MotionForceTask level whose certificate fails used to send the robot to the generic kernel, which redid the whole
// tick in projector form (the "work-list pass": half of a C4 step for 5.5 % of the robots). Everything the handler does
// inside and around a blending region (SingularityHandler.cpp:76-160, 230-295, 313-367) has a whitened form too. With the
// thin SVD Jp = U S V^T and Y' = L^-1 Jp^T U:
//     non-singular columns:  Y'_ns = Z_ns R_ns:  Lambda_ns = (R^T R)^-1,  N_ns = L^-T (I - Z_ns Z_ns^T) L^T
//     singular column:       Lambda_s = 1 / (y'_s . y'_s)
//     posture task  Jpost = v_s^T N_ns N_prec:   L^-1 Jpost^T = Q' L^-1 v_s =: y_p,  Q' = Q - Z_ns Z_ns^T,  Lambda_joint_s = 1 / (y_p . y_p)
//     N N_prec = L^-T (Q' - y_p y_p^T / |y_p|^2) L^T
// so the level still takes `rank` directions out of Q — Z_ns and y_p instead of the Z of Y — and the cascade below goes on
// in whitened form. ONE singular direction (singular_streamed below says how it is found without an SVD); what is not
// handled goes to the work list as before: two or more small singular values, a fully singular task (s_0 < s_abs_tol) and
// enforce_handling_strategy = false (they consume fewer directions than `rank`: wrows would stop being batch-uniform),
// a second singular MotionForceTask of the same robot.

// singularity bookkeeping of the one MotionForceTask that went through the branch, kept in registers until the robot is
// known to finish in this kernel (flush_singular)
struct SingPend {
	// set by the kernel: is this call one that commits the once-per-model-update bookkeeping (the fused tick, updateTaskModel:
	// classifySingularity runs in SingularityHandler::updateTaskModel), and one that computes torques (the type-2 direction
	// memory changes in computeTorques, :339-345)?
	int commit, store_t2;
	int took;  // out: the robot went through the singular branch (the host's choice of kernel looks at how many do)
	int task;  // -1: none
	int clear, write_prior, ring, ntypes, idx, word, count, size, c1, c2;
	int t2mask;	 // bit 2 i: store MFT_T2DIR + i, bit 2 i + 1: the value is +1 (else -1)
};
struct SingArgs {
	const DevParams* P;
	const DevTask* t;
	int ti, B, b, enabled;
	real fnorm;		 // |unit_mass_force + force_related_terms| over all six coordinates (SingularityHandler.cpp:349)
	const real* pu;	 // M columns of the basis of range(P) in the six task coordinates (6 x 6 row-major), or NULL: the leading ones
	const real* pose;  // this lane's LDS column behind the bounded-inertia factor (POSE_SLOTS doubles, stride 64), see singular_tail
	SingPend* sp;
};

// pose of the control frame alone (one running frame: classifySingularity's perturbed kinematics, :253-258)
template <class MD>
DI void pose_only(const MD& md, const DevTask& t, const real* q, real* x, real* R) {
	real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
	UNROLL for (int k = 0; k < 3; k++) x[k] = 0;
	UNROLL for (int k = 0; k < 9; k++) R[k] = 0;
	UNROLL for (int i = 0; i < N; i++) {
		real RE[9], pi[3];
		UNROLL for (int k = 0; k < 3; k++)
			pi[k] = fma(Rp[3 * k], md.xyz[i][0], fma(Rp[3 * k + 1], md.xyz[i][1], fma(Rp[3 * k + 2], md.xyz[i][2], pp[k])));
		mm<3, 3, 3>(Rp, md.E[i], RE);
		real s, c;
		sincos_joint(q[i], &s, &c);
		const bool pris = md.jtype[i] != 0;
		if (pris) s = 0, c = 1;
		UNROLL for (int k = 0; k < 3; k++) {
			Rp[3 * k + 0] = fma(c, RE[3 * k], s * RE[3 * k + 1]);
			Rp[3 * k + 1] = fma(c, RE[3 * k + 1], -s * RE[3 * k]);
			Rp[3 * k + 2] = RE[3 * k + 2];
			if (pris) pi[k] = fma(q[i], RE[3 * k + 2], pi[k]);
			pp[k] = pi[k];
		}
		if (t.link == i) {
			UNROLL for (int k = 0; k < 3; k++)
				x[k] = fma(Rp[3 * k], t.frame_pos[0], fma(Rp[3 * k + 1], t.frame_pos[1], fma(Rp[3 * k + 2], t.frame_pos[2], pp[k])));
			mm<3, 3, 3>(Rp, t.frame_rot, R);
		}
	}
}

// Gram-Schmidt of the columns of Y flagged `on` (Z overwrites them; the others become zero columns with rinv = 0, so that
// solves with R leave zeros in their places). Returns the smallest squared norm met (a collapsed column: caller declines).
template <int M>
DI real masked_gram_schmidt(real* Y, const bool* on, real* R, real* rinv) {
	real least = 1e300;
	UNROLL for (int j = 0; j < M; j++) {
		real nn = 0;
		UNROLL for (int i = 0; i < N; i++) nn = fma(Y[j * N + i], Y[j * N + i], nn);
		least = on[j] ? fmin(least, nn) : least;
		const real r = on[j] ? rsqrt(nn) : 0.0;
		rinv[j] = r;
		UNROLL for (int i = 0; i < N; i++) Y[j * N + i] = on[j] ? Y[j * N + i] * r : 0.0;
		UNROLL for (int k = j + 1; k < M; k++) {
			real s = 0;
			UNROLL for (int i = 0; i < N; i++) s = fma(Y[j * N + i], Y[k * N + i], s);
			R[j * M + k] = s;
			UNROLL for (int i = 0; i < N; i++) Y[k * N + i] = fma(-s, Y[j * N + i], Y[k * N + i]);
		}
	}
	return least;
}
// (A^T A restricted to the flagged columns)^-1 a, A's columns given as A[j * N + i]; unflagged places return 0
template <int M>
DI void masked_gram_solve(const real* A, const bool* on, real* a) {
	real G[M * M], LG[M * M], dG[M];
	UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < N; l++) s = fma(A[i * N + l], A[j * N + l], s);
		G[i * M + j] = (on[i] && on[j]) ? s : ((i == j) ? 1.0 : 0.0);
	}
	chol<M>(G, LG, dG);
	UNROLL for (int j = 0; j < M; j++) a[j] = on[j] ? a[j] : 0.0;
	solve_lower<M>(LG, dG, a);
	solve_lower_t<M>(LG, dG, a);
}
// A <- LB^-1 A for the M columns of A (A[j * N + i]): the factor of the bounded inertia estimate comes from LDS a row at a
// time (each row serves all the columns) and is never whole in registers
template <int M>
DI void solve_lb_columns(const real* lb, real* A) {
	UNROLL for (int i = 0; i < N; i++) {
		real row[N];
		UNROLL for (int k = 0; k < i; k++) row[k] = lb[(i * (i + 1) / 2 + k) * 64];
		const real di = lb[(N * (N + 1) / 2 + i) * 64];
		UNROLL for (int c = 0; c < M; c++) {
			real t = A[c * N + i];
			UNROLL for (int k = 0; k < i; k++) t = fma(-row[k], A[c * N + k], t);
			A[c * N + i] = t * di;
		}
	}
}
// tau += L w
DI void add_l_times(const real* L, const real* w, real* tau) {
	UNROLL for (int i = 0; i < N; i++) {
		real s = 0;
		UNROLL for (int k = 0; k <= i; k++) s = fma(L[i * N + k], w[k], s);
		tau[i] += s;
	}
}

// The second half of the singular branch (singular_streamed): bookkeeping (classification by perturbed kinematics, history ring), the posture task in the
// singular joint direction, the joint strategy, the blend, and the posture direction out of Q. xs: the singular column of
// Xs = Jp^T U (= sigma v), ws: the singular column of U in the task's reduced coordinates, tau_s: the sanitised and
// clamped singular-direction torques; Q: already without the regular block's directions (Q').
template <int M>
DI bool singular_tail(const Fact& f, const SingArgs& sa, bool reg, real least, const real* xs, const real* ws, real s_last, real alpha,
					  const real* tau_s, int decoupling, const real* fu, const real* ff, real* Q, real* tau) {
	const DevParams& P = *sa.P;
	const DevTask& t = *sa.t;
	const int B = sa.B, b = sa.b;
	SingPend& sp = *sa.sp;
	const bool bie = decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES, impedance = decoupling == SAI2B_IMPEDANCE;
	// ---- bookkeeping (classifySingularity, :230-295)
	const int prev_types = ldi(t.istate, IS_NTYPES, B, b);
	sp.task = sa.ti;
	sp.clear = sp.write_prior = sp.t2mask = sp.ring = 0;
	sp.ntypes = sp.idx = sp.word = sp.count = sp.size = sp.c1 = sp.c2 = 0;
	if (reg) {	// the SVD says: not singular after all
		sp.clear = sp.commit && prev_types != 0;
		if (!sp.clear) sp.task = -1;
		return least > 1e-280;
	}
	int c1 = ldi(t.istate, IS_C1, B, b), c2 = ldi(t.istate, IS_C2, B, b);
	const int ring_count = ldi(t.istate, IS_COUNT, B, b), ring_size = ldi(t.istate, IS_SIZE, B, b);
	real q[N];
	UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
	const int ring_word = ldi(t.istate, (ring_count % t.sh_cap) >> 5, B, b);
	sp.write_prior = sp.commit && (prev_types == 0 || c2 > c1);
	// V_s = Xs / sigma with the sign convention shared with the oracle (largest-magnitude component positive)
	real v[N], us[M];
	{
		real big = 0, bigabs = -1;
		UNROLL for (int i = 0; i < N; i++) {
			const bool take = fabs(xs[i]) > bigabs;
			bigabs = take ? fabs(xs[i]) : bigabs;
			big = take ? xs[i] : big;
		}
		const real inv = s_last > 0 ? 1.0 / s_last : 0.0;
		const real vs = big < 0 ? -inv : inv, sgn = big < 0 ? -1.0 : 1.0;
		UNROLL for (int i = 0; i < N; i++) v[i] = vs * xs[i];
		UNROLL for (int r = 0; r < M; r++) us[r] = sgn * ws[r];
	}
	bool any1 = false;
	if (sp.commit) {  // (computeTorques behind updateTaskModel of the same state classifies nothing)
		// the pose at q: parked in LDS by the Jacobian sweep where the lane's column has room for it (position and two
		// columns of the rotation), recomputed otherwise
		real x0[3], R0[9];
		if (POSE_SLOTS == 9 && sa.pose && t.frame_rigid) {
			UNROLL for (int k = 0; k < 3; k++) {
				x0[k] = sa.pose[k * 64];
				R0[3 * k] = sa.pose[(3 + k) * 64];
				R0[3 * k + 1] = sa.pose[(6 + k) * 64];
			}
			R0[2] = R0[3] * R0[7] - R0[6] * R0[4];
			R0[5] = R0[6] * R0[1] - R0[0] * R0[7];
			R0[8] = R0[0] * R0[4] - R0[3] * R0[1];
		} else {
			pose_only(P.model, t, q, x0, R0);
		}
		real u6[6];
		UNROLL for (int k = 0; k < 6; k++) {
			real s = 0;
			if (sa.pu) {
				UNROLL for (int r = 0; r < M; r++) s = fma(sa.pu[k * 6 + r], us[r], s);
			} else {
				UNROLL for (int r = 0; r < M; r++) s = (k == r) ? us[r] : s;
			}
			u6[k] = s;
		}
		// classification by FK perturbation (:253-273) along +v, -v or both (enum sai2b_singular_vector_sign)
		const int pass0 = t.sv_sign == SAI2B_SV_SIGN_V_MAX_NEGATIVE ? 1 : 0;
		const int pass1 = t.sv_sign == SAI2B_SV_SIGN_V_MAX_POSITIVE ? 0 : 1;
		bool moved[2] = {false, false};
#pragma unroll 1
		for (int pass = pass0; pass <= pass1; pass++) {
			const real step = pass ? -t.perturb : t.perturb;
			real qp[N], x1[3], R1[9], d[6];
			UNROLL for (int i = 0; i < N; i++) qp[i] = fma(step, v[i], q[i]);
			pose_only(P.model, t, qp, x1, R1);
			UNROLL for (int k = 0; k < 3; k++) d[k] = x1[k] - x0[k];
			orientation_error(R1, R0, d + 3);
			real m = 0;
			UNROLL for (int k = 0; k < 6; k++) m = fma(d[k], u6[k], m);
			if (pass)
				moved[1] = fabs(m) > t.type_1_tol;
			else
				moved[0] = fabs(m) > t.type_1_tol;
		}
		any1 = t.sv_sign == SAI2B_SV_SIGN_BOTH ? (moved[0] && moved[1]) : (moved[0] || moved[1]);
	}
	CSTAMP_ANY(55);
	if (sp.commit) {  // history ring (:276-293), stored by flush_singular
		int count = ring_count, size = ring_size;
		const int cap = t.sh_cap;
		const int idx = count % cap;
		int word = ring_word;
		const int bit = 1 << (idx & 31);
		if (size == cap) {
			if (word & bit)
				c1--;
			else
				c2--;
		} else {
			size++;
		}
		if (any1) {
			word |= bit;
			c1++;
		} else {
			word &= ~bit;
			c2++;
		}
		sp.idx = idx >> 5, sp.word = word, sp.count = (count + 1) % (cap * 32768), sp.size = size, sp.c1 = c1, sp.c2 = c2, sp.ntypes = 1, sp.ring = 1;
	}
	CSTAMP_ANY(56);
	// ---- posture task in the singular joint direction (:152-157): yp = Q' L^-1 v, and the joint strategy (:327-351)
	real yp[N];
	{
		real col[N];
		UNROLL for (int i = 0; i < N; i++) col[i] = v[i];
		solve_lower<N>(f.L, f.dL, col);
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int k = 0; k < N; k++) s = fma(symat(Q, i, k), col[k], s);
			yp[i] = s;
		}
	}
	real npp = 0;
	UNROLL for (int i = 0; i < N; i++) npp = fma(yp[i], yp[i], npp);
	CSTAMP_ANY(57);
	if (!impedance) {
		// V_s^T of the unit torques: what goes through Lambda_joint_s_modified (hl) and what goes in directly (hd); both
		// strategies without branches
		real hl = 0, hd = 0;
		{
			const bool type1 = c1 > c2 || t.enforce_t1;	 // joint holding to the entering conditions, else open-loop torques
			real fTd = 0;
			const real finv = sa.fnorm > 0 ? 1.0 / sa.fnorm : 1.0;	// normalized() of a zero vector is the vector
			UNROLL for (int r = 0; r < M; r++) fTd = fma((fu[r] + ff[r]) * finv, us[r], fTd);
			const real mag = fabs(fTd) * t.t2_ratio;
			int mask = 0;
			UNROLL for (int i = 0; i < N; i++) {
				const real dqi = ld(P.dq, i, B, b);
				const real qpi = sp.write_prior ? q[i] : ld(t.state, MFT_QPRIOR + i, B, b);
				const real t2i = ld(t.state, MFT_T2DIR + i, B, b);
				const bool has = v[i] != 0;
				const bool up = has && fabs(q[i] - P.model.q_upper[i]) < t.t2_angle;
				const bool lo = has && !up && fabs(q[i] - P.model.q_lower[i]) < t.t2_angle;
				const real dir = up ? -1.0 : (lo ? 1.0 : t2i);
				mask |= (up ? 1 : (lo ? 3 : 0)) << (2 * i);
				const real um = type1 ? 0.0 : dir * mag * P.model.effort[i];
				const real ut = type1 ? -t.kp1 * (q[i] - qpi) - t.kv1 * dqi : -t.kv2 * dqi;
				hl = fma(v[i], ut, hl);
				hd = fma(v[i], um, hd);
			}
			sp.t2mask = (type1 || !sp.store_t2) ? 0 : mask;
		}
		real lam = 1.0 / npp;  // Lambda_joint_s
		if (bie) {			   // Lambda_joint_s_modified = (Jpost M_BIE^-1 Jpost^T)^-1, Jpost^T = L yp (:202-205)
			real yb[N];
			UNROLL for (int i = 0; i < N; i++) {
				real s = 0;
				UNROLL for (int k = 0; k <= i; k++) s = fma(f.L[i * N + k], yp[k], s);
				yb[i] = s;
			}
			solve_lb_columns<1>(f.lb, yb);
			real g = 0;
			UNROLL for (int i = 0; i < N; i++) g = fma(yb[i], yb[i], g);
			lam = 1.0 / g;
		}
		const real coef = fma(lam, hl, hd);
		real w[N], tj[N];
		UNROLL for (int i = 0; i < N; i++) w[i] = yp[i] * coef, tj[i] = 0;
		add_l_times(f.L, w, tj);
		UNROLL for (int i = 0; i < N; i++) tau[i] += alpha * tau_s[i] + (1 - alpha) * tj[i];  // :366
	}
	CSTAMP_ANY(58);
	// N = N_posture N_ns: Q'' = Q' - z_p z_p^T
	{
		const real r2 = 1.0 / npp;
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-yp[i] * r2, yp[j], Q[i * N + j]);
	}
	CSTAMP_ANY(59);
	return least > 1e-280 && npp > 1e-280;
}

// The cascade alone through a singular level (the range pass ahead of the trajectory generators, cert::range_tick): the
// directions the level takes out of Q — the regular block's and the posture task's — without forces, torques or
// bookkeeping. Same scope and same decisions as singular_streamed (the decomposition is the one-sided Jacobi here).
template <int M>
DI bool singular_range(const Fact& f, const SingArgs& sa, const real* Y, const real* JP, real* Q) {
	constexpr int K = M - 1;
	const DevTask& t = *sa.t;
	real X[N * M], W[M * M];
	UNROLL for (int c = 0; c < M; c++) UNROLL for (int i = 0; i < N; i++) X[i * M + c] = JP[c * N + i];
	hestenes<N, M>(X, W);
	real sv[M];
	UNROLL for (int j = 0; j < M; j++) {
		real a = 0;
		UNROLL for (int r = 0; r < N; r++) a = fma(X[r * M + j], X[r * M + j], a);
		sv[j] = sqrt(a);
	}
	real s0 = 0, s_last = 0, s_prev = 0;
	int js = 0;
	UNROLL for (int j = 0; j < M; j++) {
		int p = 0;
		UNROLL for (int k = 0; k < M; k++) p += (sv[k] > sv[j] || (sv[k] == sv[j] && k < j)) ? 1 : 0;
		s0 = fmax(s0, sv[j]);
		js = (p == M - 1) ? j : js;
		s_last = (p == M - 1) ? sv[j] : s_last;
		s_prev = (p == M - 2) ? sv[j] : s_prev;
	}
	if (s0 < t.s_abs_tol) return false;
	if (M > 2 && s_prev / s0 < t.s_max) return false;
	const bool reg = !(s_last / s0 < t.s_max);
	if (!reg && !t.enforce) return false;
	// Y' = Y U in the order [regular block | js], Gram-Schmidt of the block (and of the last column when it is regular)
	real Yc[M * N], v[N];
	UNROLL for (int k = 0; k < M; k++) UNROLL for (int i = 0; i < N; i++) {
		real y = 0;
		UNROLL for (int r = 0; r < M; r++) {
			real u = 0;
			if (k < K) {
				u = (js <= k) ? W[r * M + (k + 1 < M ? k + 1 : k)] : W[r * M + k];
			} else {
				UNROLL for (int j = 0; j < M; j++) u = (js == j) ? W[r * M + j] : u;
			}
			y = fma(Y[r * N + i], u, y);
		}
		Yc[k * N + i] = y;
	}
	UNROLL for (int i = 0; i < N; i++) {
		real x = 0;
		UNROLL for (int j = 0; j < M; j++) x = (js == j) ? X[i * M + j] : x;
		v[i] = x;  // sigma v_s: the scale does not matter to a direction
	}
	bool on[M];
	UNROLL for (int k = 0; k < M; k++) on[k] = (k < K) ? true : reg;
	real R[M * M], rinv[M];
	real least = masked_gram_schmidt<M>(Yc, on, R, rinv);
	UNROLL for (int c = 0; c < M; c++)
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-Yc[c * N + i], Yc[c * N + j], Q[i * N + j]);
	if (reg) return least > 1e-280;
	// the posture direction y_p = Q' L^-1 v_s
	real yp[N];
	solve_lower<N>(f.L, f.dL, v);
	real npp = 0;
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int k = 0; k < N; k++) a = fma(symat(Q, i, k), v[k], a);
		yp[i] = a;
		npp = fma(a, a, npp);
	}
	const real r2 = 1.0 / npp;
	UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-yp[i] * r2, yp[j], Q[i * N + j]);
	return least > 1e-280 && npp > 1e-280;
}

// the deferred bookkeeping of the singular branch, for a robot that finishes in this kernel
DI void flush_singular(const DevParams& P, int B, int b, const SingPend& sp) {
	if (sp.task < 0) return;
	const DevTask& t = P.task[sp.task];
	int* IS = t.istate;
	if (sp.clear) {	 // leaving the singular region (:239-245)
		sti(IS, IS_NTYPES, B, b, 0);
		sti(IS, IS_COUNT, B, b, 0);
		sti(IS, IS_SIZE, B, b, 0);
		sti(IS, IS_C1, B, b, 0);
		sti(IS, IS_C2, B, b, 0);
		return;
	}
	if (sp.write_prior) {  // entering conditions (:233-236)
		for (int i = 0; i < N; i++) {
			st(t.state, MFT_QPRIOR + i, B, b, ld(P.q, i, B, b));
			st(t.state, MFT_DQPRIOR + i, B, b, ld(P.dq, i, B, b));
		}
	}
	for (int i = 0; i < N; i++)
		if (sp.t2mask & (1 << (2 * i))) st(t.state, MFT_T2DIR + i, B, b, (sp.t2mask & (2 << (2 * i))) ? 1.0 : -1.0);
	if (!sp.ring) return;
	sti(IS, sp.idx, B, b, sp.word);
	sti(IS, IS_COUNT, B, b, sp.count);
	sti(IS, IS_SIZE, B, b, sp.size);
	sti(IS, IS_C1, B, b, sp.c1);
	sti(IS, IS_C2, B, b, sp.c2);
	sti(IS, IS_NTYPES, B, b, sp.ntypes);
}

// ---- The singular branch, torques (every task size: level() for 2 and 3 rows, level_streamed() for 4 to 6) -------------
// Column rotations of Jp^T with U beside it (a one-sided Jacobi: round 3's first form for 2- and 3-row tasks) are 78 numbers
// for a 6-row task and do not fit beside L, Q and Jp (1.9 KB of scratch per lane, the branch 250 us). With ONE singular
// direction the handler needs less than an SVD:
// the smallest singular triplet (sigma_s, u_s, v_s = Jp^T u_s / sigma_s), the largest and the second smallest singular values
// for its decisions, and for everything regular only the SUBSPACE orthogonal to u_s — a regular level does not care about
// an orthogonal change of its task coordinates. So:
//   * eigenvalues of the M x M Gram matrix G = Jp Jp^T by cyclic two-sided Jacobi, values only (21 numbers in registers);
//     absolute accuracy eps |G|, i.e. 1e-12 relative on the smallest one inside a blending region — it only feeds decisions
//     with thresholds (s_max) and the blending weight alpha;
//   * u_s by inverse iteration on G - mu I (mu just below the smallest eigenvalue: four iterations), sigma_s = |Jp^T u_s|
//     from Jp itself;
//   * a Householder reflector H with H e_M = +-u_s: the rows of H^T Jp are [an orthonormal mix of the regular directions |
//     +-sigma_s v_s^T], the singular one LAST at compile time: the regular block is level_streamed's Gram-Schmidt through the
//     running projector over M - 1 rows, the singular direction is one row and scalar Lambdas, the rest is singular_tail.
// 1 / x and 1 / sqrt(x) from the hardware estimates with two Newton steps (a rotation angle does not need the IEEE division)
DI real recip_nr(real x) {
	real r = __builtin_amdgcn_rcp(x);
	r = fma(fma(-x, r, 1.0), r, r);
	return fma(fma(-x, r, 1.0), r, r);
}
DI real rsqrt_nr(real x) {
	real r = __builtin_amdgcn_rsq(x);
	const real h = 0.5 * x;
	r = fma(fma(-h * r, r, 0.5), r, r);
	return fma(fma(-h * r, r, 0.5), r, r);
}
template <int M>
DI void sym_eigenvalues(real* G, real* lam) {  // G: lower triangle in M x M row-major storage, destroyed
#pragma unroll 1
	for (int sweep = 0; sweep < 30; sweep++) {
		real off = 0, dia = 0;
		UNROLL for (int i = 0; i < M; i++) {
			dia = fma(G[i * M + i], G[i * M + i], dia);
			UNROLL for (int j = 0; j < i; j++) off = fma(G[i * M + j], G[i * M + j], off);
		}
		if (!(off > 1e-26 * dia)) break;	// off-diagonal norm below 1e-13 of the diagonal's: the eigenvalues move by its square
		UNROLL for (int p = 0; p < M - 1; p++) UNROLL for (int q = p + 1; q < M; q++) {
			const real apq = G[q * M + p];
			const real app = G[p * M + p], aqq = G[q * M + q];
			// rotation that zeroes (p, q): t = sign(theta) / (|theta| + sqrt(theta^2 + 1)), theta = (aqq - app) / (2 apq);
			// written without the division by apq (a zero off-diagonal gives t = 0)
			const real d = aqq - app, g2 = 2 * apq;
			const real h2 = fma(d, d, g2 * g2);
			const real den = fabs(d) + (h2 > 0 ? h2 * rsqrt_nr(h2) : 0.0);
			const real t = den > 0 ? (d < 0 ? -g2 : g2) * recip_nr(den) : 0.0;
			const real c = rsqrt_nr(fma(t, t, 1.0)), sn = c * t;
			G[p * M + p] = fma(-t, apq, app);
			G[q * M + q] = fma(t, apq, aqq);
			G[q * M + p] = 0;
			UNROLL for (int k = 0; k < M; k++) {
				if (k != p && k != q) {
					real& gkp = (k > p) ? G[k * M + p] : G[p * M + k];
					real& gkq = (k > q) ? G[k * M + q] : G[q * M + k];
					const real a = gkp, b2 = gkq;
					gkp = fma(c, a, -sn * b2);
					gkq = fma(sn, a, c * b2);
				}
			}
		}
	}
	UNROLL for (int i = 0; i < M; i++) lam[i] = G[i * M + i];
}

template <int M>
DI bool singular_streamed(const Fact& f, const SingArgs& sa, real* JP, int decoupling, const real* fu_in, const real* ff_in, real* Q,
						  real* tau) {
	const DevTask& t = *sa.t;
	const DevParams& P = *sa.P;
	SingPend& sp = *sa.sp;
	sp.took = 1;
	if (sp.task >= 0) return false;	 // one per robot
	real us[M];
	real s0, s_last, s_prev;
	{
		real G[M * M], lam[M];
		UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
			real a = 0;
			UNROLL for (int l = 0; l < N; l++) a = fma(JP[i * N + l], JP[j * N + l], a);
			G[i * M + j] = a;
		}
		sym_eigenvalues<M>(G, lam);
		real l0 = 0, l1 = 1e300, l2 = 1e300;  // largest, smallest, second smallest
		UNROLL for (int i = 0; i < M; i++) {
			const real v = fmax(lam[i], 0.0);
			l0 = fmax(l0, v);
			const bool lt1 = v < l1;
			l2 = lt1 ? l1 : fmin(l2, v);
			l1 = lt1 ? v : l1;
		}
		s0 = sqrt(l0), s_prev = sqrt(l2);
		if (s0 < t.s_abs_tol) return false;		  // fully singular: the task is passed through
		if (s_prev / s0 < t.s_max) return false;  // two or more singular directions
		// u_s: inverse iteration on G - mu I (Cholesky: the shifted matrix is positive definite by 1e-6 of its smallest eigenvalue,
		// or by 1e-14 of the largest; the error shrinks by <= 1e-5 per iteration while the next eigenvalue is 1.2 times the smallest)
		const real mu = l1 * (1.0 - 1e-6) - 1e-14 * l0;
		real A[M * M], LA[M * M], dA[M];
		UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
			real a = (i == j) ? -mu : 0.0;
			UNROLL for (int l = 0; l < N; l++) a = fma(JP[i * N + l], JP[j * N + l], a);
			A[i * M + j] = a;
		}
		chol<M>(A, LA, dA);
		UNROLL for (int i = 0; i < M; i++) us[i] = 1.0 + 0.37 * i;	// (any vector with a component along u_s)
#pragma unroll 1
		for (int it = 0; it < 4; it++) {
			solve_lower<M>(LA, dA, us);
			solve_lower_t<M>(LA, dA, us);
			real nn = 0;
			UNROLL for (int i = 0; i < M; i++) nn = fma(us[i], us[i], nn);
			const real r = rsqrt(nn);
			UNROLL for (int i = 0; i < M; i++) us[i] *= r;
		}
	}
	// sigma_s v_s = Jp^T u_s
	real xs[N];
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int c = 0; c < M; c++) a = fma(JP[c * N + i], us[c], a);
		xs[i] = a;
	}
	{
		real a = 0;
		UNROLL for (int i = 0; i < N; i++) a = fma(xs[i], xs[i], a);
		s_last = sqrt(a);
	}
	const real icn = s_last / s0;
	const bool reg = !(icn < t.s_max);	// not singular after all
	const real alpha = reg ? 1.0 : fmin(fmax((icn - t.s_min) / (t.s_max - t.s_min), 0.0), 1.0);
	if (!reg && !t.enforce) return false;
	const bool bie = decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES, impedance = decoupling == SAI2B_IMPEDANCE;
	const bool full = decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	// Householder H = I - 2 w w^T with H e_M = sg u_s (sg = -sign(u_s[M-1]): no cancellation in u_s - sg e_M)
	real fu[M], ff[M];
	{
		const real sg = us[M - 1] > 0 ? -1.0 : 1.0;
		real w[M];
		UNROLL for (int i = 0; i < M; i++) w[i] = us[i] * sg - ((i == M - 1) ? 1.0 : 0.0);	// sg u_s - e_M
		real nn = 0;
		UNROLL for (int i = 0; i < M; i++) nn = fma(w[i], w[i], nn);
		const real beta = 2.0 / nn;
		// rows: JP <- H^T JP = JP - beta w (w^T JP);  forces: H^T F
		UNROLL for (int l = 0; l < N; l++) {
			real a = 0;
			UNROLL for (int c = 0; c < M; c++) a = fma(w[c], JP[c * N + l], a);
			a *= beta;
			UNROLL for (int c = 0; c < M; c++) JP[c * N + l] = fma(-w[c], a, JP[c * N + l]);
		}
		real au = 0, af = 0;
		UNROLL for (int c = 0; c < M; c++) {
			au = fma(w[c], fu_in[c], au);
			af = fma(w[c], ff_in[c], af);
		}
		au *= beta, af *= beta;
		UNROLL for (int c = 0; c < M; c++) {
			fu[c] = fma(-w[c], au, fu_in[c]);
			ff[c] = fma(-w[c], af, ff_in[c]);
		}
		// the last row is now sg sigma_s v_s^T, the last force coordinates sg u_s^T F: take the sign out
		UNROLL for (int l = 0; l < N; l++) JP[(M - 1) * N + l] = xs[l];
		fu[M - 1] *= sg, ff[M - 1] *= sg;
	}
	// ---- regular block: direct terms, then the Gram-Schmidt through the running projector (level_streamed, pass 2)
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int k = 0; k < M; k++) a = fma(JP[k * N + i], (k < M - 1 || reg) ? ff[k] + (impedance ? fu[k] : 0.0) : 0.0, a);
		tau[i] += a;
	}
	real least = 1e300, gs = 0;
	{
		real w[N];
		UNROLL for (int i = 0; i < N; i++) w[i] = 0;
		UNROLL for (int k = 0; k < M; k++) {
			const bool on = (k < M - 1) || reg;
			real y[N], z[N];
			UNROLL for (int i = 0; i < N; i++) y[i] = JP[k * N + i];
			solve_lower<N>(f.L, f.dL, y);
			real wy = 0, yy = 0;
			UNROLL for (int i = 0; i < N; i++) {
				wy = fma(w[i], y[i], wy);
				yy = fma(y[i], y[i], yy);
			}
			if (k == M - 1) gs = yy;  // Lambda_s = 1 / (y_s . y_s) (:121)
			real nn = 0;
			UNROLL for (int i = 0; i < N; i++) {
				real a = 0;
				UNROLL for (int j = 0; j < N; j++) a = fma(symat(Q, i, j), y[j], a);
				z[i] = a;
				nn = fma(a, a, nn);
			}
			least = on ? fmin(least, nn) : least;
			const real r = on ? rsqrt(nn) : 0.0;
			UNROLL for (int i = 0; i < N; i++) z[i] = on ? z[i] * r : 0.0;
			if (full) {
				const real u = (fu[k] - wy) * r;
				UNROLL for (int i = 0; i < N; i++) w[i] = fma(z[i], u, w[i]);
			}
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-z[i], z[j], Q[i * N + j]);
			__builtin_amdgcn_sched_barrier(0);
		}
		if (full) add_l_times(f.L, w, tau);
	}
	// ---- Lambda_s_modified U_s^T Fu (scalar); with bounded inertia the regular block's Lambda_ns_modified too (:184-206)
	real zs = fu[M - 1] / gs;
	if (bie) {
		solve_lb_columns<M>(f.lb, JP);	// YB = LB^-1 Xs overwrites the rows
		real g = 0;
		UNROLL for (int i = 0; i < N; i++) g = fma(JP[(M - 1) * N + i], JP[(M - 1) * N + i], g);
		zs = fu[M - 1] / g;
		bool on[M];
		UNROLL for (int k = 0; k < M; k++) on[k] = (k < M - 1) || reg;
		real zn[M];
		UNROLL for (int k = 0; k < M; k++) zn[k] = fu[k];
		masked_gram_solve<M>(JP, on, zn);
		real w[N];
		UNROLL for (int i = 0; i < N; i++) {
			real a = 0;
			UNROLL for (int k = 0; k < M; k++) a = fma(JP[k * N + i], zn[k], a);  // (zn is zero outside the block)
			w[i] = a;
		}
		UNROLL for (int i = 0; i < N; i++) {  // Jp_ns^T x = LB (YB x)
			real a = 0;
			UNROLL for (int k = 0; k <= i; k++) a = fma(f.lb[(i * (i + 1) / 2 + k) * 64], w[k], a);
			tau[i] += a;
		}
	}
	// ---- singular-direction torques, sanitised and clamped (:354-365)
	real tau_s[N];
	UNROLL for (int i = 0; i < N; i++) {
		real a = xs[i] * (zs + ff[M - 1]);
		a = (a != a) ? 0.0 : fmin(fmax(a, -P.model.effort[i]), P.model.effort[i]);
		tau_s[i] = reg ? 0.0 : a;
	}
	// (singular_tail's force terms — the type-2 magnitude (F . u_s) / |F| — in the original task coordinates)
	return singular_tail<M>(f, sa, reg, least, xs, us, s_last, alpha, tau_s, decoupling, fu_in, ff_in, Q, tau);
}

// The same level for tasks of 4 to 6 rows, where Y (42 doubles) and R beside Jp, L and Q do not fit the register file
// (round 2: 1.1 KB of scratch per lane in the 6-row instantiation): Y is never stored. Pass 1 builds the rows of Jp
// (Jp^T = L Q L^-1 Jr^T) for the certificate and the torque terms that need Jp; pass 2 is the Gram-Schmidt one column at a
// time with the running projector doing the orthogonalisation: y'_c = Q_c (L^-1 jp_c) is already orthogonal to the
// columns before it once their directions are out of Q (Q_c = Q - sum_k<c z_k z_k^T), so z_c = y'_c / |y'_c| with
// r_cc = |y'_c|, and the Lambda term L Z R^-T a needs no stored Z or R either: with w = sum_k<c z_k u_k,
// u_c = (a_c - w . y_c) / r_cc  (sum_k R_kc u_k = w . y_c, and w . y_c = w . L^-1 jp_c because w lies in range(Q)).
// The bounded-inertia term comes last and overwrites Jp. One triangular solve per row more than level(), 80 live doubles fewer.
template <int M, bool TORQUE = true, bool SING = false>
DI bool level_streamed(const Fact& f, const real* Jr, bool first, bool last, bool do_cert, real abs2, real rel2, int decoupling,
					   bool has_va, const real* va, const real* vf, const real* vd, real* Q, real* tau, const SingArgs* sa = nullptr) {
	real JP[M * N];
	UNROLL for (int c = 0; c < M; c++) {
		if (first) {
			UNROLL for (int i = 0; i < N; i++) JP[c * N + i] = Jr[c * N + i];
		} else {
			real col[N], y[N];
			UNROLL for (int i = 0; i < N; i++) col[i] = Jr[c * N + i];
			solve_lower<N>(f.L, f.dL, col);
			UNROLL for (int i = 0; i < N; i++) {
				real s = 0;
				UNROLL for (int j = 0; j < N; j++) s = fma(symat(Q, i, j), col[j], s);
				y[i] = s;
			}
			UNROLL for (int i = 0; i < N; i++) {  // row c of Jp = (L Y)^T
				real s = 0;
				UNROLL for (int k = 0; k <= i; k++) s = fma(f.L[i * N + k], y[k], s);
				JP[c * N + i] = s;
			}
		}
		__builtin_amdgcn_sched_barrier(0);	// one column at a time
	}
	CSTAMP(20);
	bool ok = true;
	if (do_cert) {
		real G[M * M];
		UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			UNROLL for (int l = 0; l < N; l++) s = fma(JP[i * N + l], JP[j * N + l], s);
			G[i * M + j] = s;
		}
		ok = certify_gram_lower<M>(G, abs2, rel2);
	}
	if constexpr (SING && TORQUE) {
		if (sa->enabled && !ok) return singular_streamed<M>(f, *sa, JP, decoupling, vf, vd, Q, tau);
	}
	SAI2B_PHASE();
	CSTAMP(21);
	const bool full = decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	if constexpr (TORQUE) {	 // direct terms: Jp^T (vd (+ vf with IMPEDANCE))
		const bool imp = decoupling == SAI2B_IMPEDANCE;
		UNROLL for (int c = 0; c < M; c++) {
			const real zc = vd[c] + (imp ? vf[c] : 0.0);
			UNROLL for (int i = 0; i < N; i++) tau[i] = fma(JP[c * N + i], zc, tau[i]);
		}
	}
	SAI2B_PHASE();
	// Gram-Schmidt through the running projector, the Lambda term and the downdate of Q, a column at a time
	{
		real w[N];
		UNROLL for (int i = 0; i < N; i++) w[i] = 0;
		const bool lam = TORQUE && (has_va || full);
		UNROLL for (int c = 0; c < M; c++) {
			real y[N], z[N];
			UNROLL for (int i = 0; i < N; i++) y[i] = JP[c * N + i];
			solve_lower<N>(f.L, f.dL, y);  // y_c = L^-1 jp_c (in range(Q) as it was when the level began)
			real wy = 0;
			UNROLL for (int i = 0; i < N; i++) wy = fma(w[i], y[i], wy);
			real nn = 0;
			UNROLL for (int i = 0; i < N; i++) {
				real s = 0;
				UNROLL for (int j = 0; j < N; j++) s = fma(symat(Q, i, j), y[j], s);
				z[i] = s;
				nn = fma(s, s, nn);
			}
			const real r = rsqrt(nn);
			UNROLL for (int i = 0; i < N; i++) z[i] *= r;
			if (lam) {
				const real u = ((has_va ? va[c] : 0.0) + (full ? vf[c] : 0.0) - wy) * r;
				UNROLL for (int i = 0; i < N; i++) w[i] = fma(z[i], u, w[i]);
			}
			// (the running projector is needed by the next column even at the last level; the last column's downdate is not)
			if (!last || c + 1 < M) {
				UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-z[i], z[j], Q[i * N + j]);
			}
			__builtin_amdgcn_sched_barrier(0);
		}
		if (lam) add_l_times(f.L, w, tau);	// Jp^T Lambda a = L Z R^-T a
	}
	CSTAMP(23);
	if (TORQUE && decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
		// Lambda_mod = (YB^T YB)^-1, YB = LB^-1 Jp^T (overwrites Jp); torques Jp^T x = LB (YB x)
		solve_lb_columns<M>(f.lb, JP);
		real AB[M * M], LA[M * M], dA[M], y[M];
		UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			UNROLL for (int l = 0; l < N; l++) s = fma(JP[i * N + l], JP[j * N + l], s);
			AB[i * M + j] = s;
		}
		chol<M>(AB, LA, dA);
		UNROLL for (int c = 0; c < M; c++) y[c] = vf[c];
		solve_lower<M>(LA, dA, y);
		solve_lower_t<M>(LA, dA, y);
		real w[N];
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int c = 0; c < M; c++) s = fma(JP[c * N + i], y[c], s);
			w[i] = s;
		}
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int k = 0; k <= i; k++) s = fma(f.lb[(i * (i + 1) / 2 + k) * 64], w[k], s);
			tau[i] += s;
		}
	}
	CSTAMP(24);
	return ok;
}

// One level of the cascade in reduced coordinates, M rows exactly (an instantiation per task size: the arrays of
// a 3-row task are 3 columns wide, nothing is guarded). Jr: the task's rows (Jr[c * N + i]).
// Task forces: Lambda va + Lambda_mod vf + vd with Lambda_mod by the decoupling type (SingularityHandler.cpp:
// 165-206, JointTask.cpp:240-270). Adds the level's torques to tau, takes its directions out of Q. Returns the
// certificate (true when do_cert is false: a first-level selection).
// Order of the phases keeps few matrices alive: Y, Jp -> certificate -> the torque terms that need Jp (direct and
// bounded-inertia ones: Jp^T x = LB (YB x), YB = LB^-1 Jp^T overwrites Jp) -> Gram-Schmidt of Y ->
// Lambda term as L (Z R^-T a) -> downdate of Q.
// TORQUE = false: the cascade alone (certificate and Q), for the range pass ahead of the trajectory generators.
// SING: a MotionForceTask level whose certificate fails goes through singular_streamed (torques) / singular_range (the range pass).
template <int M, bool TORQUE = true, bool SING = false>
DI bool level(const Fact& f, const real* Jr, bool first, bool last, bool do_cert, real abs2, real rel2, int decoupling,
			  bool has_va, const real* va, const real* vf, const real* vd, real* Q, real* tau, const SingArgs* sa = nullptr) {
	real Y[M * N], JP[M * N];
	UNROLL for (int c = 0; c < M; c++) {
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
		__builtin_amdgcn_sched_barrier(0);	// one column at a time: interleaving them only inflates the live set
	}
	CSTAMP(20);
	bool ok = true;
	if (do_cert) {
		real G[M * M];
		UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			UNROLL for (int l = 0; l < N; l++) s = fma(JP[i * N + l], JP[j * N + l], s);
			G[i * M + j] = s;
		}
		ok = certify_gram_lower<M>(G, abs2, rel2);
	}
	if constexpr (SING && M >= 2) {	 // (a one-row task has no blending region: it is regular or fully singular)
		if (sa->enabled && !ok) {
			// (the form written for 4- to 6-row tasks — Gram eigenvalues, inverse iteration, Householder — is also the faster
			// one for 2 and 3 rows: 74.5 against 76.0 us on C4 in a same-box A/B with round 3's first form, column rotations
			// of Jp^T with the columns re-ordered afterwards; same 5e-14 against the oracle)
			if constexpr (TORQUE)
				return singular_streamed<M>(f, *sa, JP, decoupling, vf, vd, Q, tau);
			else
				return singular_range<M>(f, *sa, Y, JP, Q);
		}
	}
	SAI2B_PHASE();
	CSTAMP(21);
	const bool full = decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	if constexpr (TORQUE) {	 // direct terms: Jp^T (vd (+ vf with IMPEDANCE: Lambda_mod = projector onto the range))
		const bool imp = decoupling == SAI2B_IMPEDANCE;
		UNROLL for (int c = 0; c < M; c++) {
			const real zc = vd[c] + (imp ? vf[c] : 0.0);
			UNROLL for (int i = 0; i < N; i++) tau[i] = fma(JP[c * N + i], zc, tau[i]);
		}
	}
	if (TORQUE && decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
		// Lambda_mod = (Jp M_BIE^-1 Jp^T)^-1 = (YB^T YB)^-1, YB = LB^-1 Jp^T; torques Jp^T x = LB (YB x)
		// LB comes from LDS a row at a time (each row serves all M columns): it is never whole in registers
		UNROLL for (int i = 0; i < N; i++) {
			real row[N];
			UNROLL for (int k = 0; k < i; k++) row[k] = f.lb[(i * (i + 1) / 2 + k) * 64];
			const real di = f.lb[(N * (N + 1) / 2 + i) * 64];
			UNROLL for (int c = 0; c < M; c++) {
				real t = JP[c * N + i];
				UNROLL for (int k = 0; k < i; k++) t = fma(-row[k], JP[c * N + k], t);
				JP[c * N + i] = t * di;
			}
		}
		real AB[M * M], LA[M * M], dA[M], y[M];
		UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
			real s = 0;
			UNROLL for (int l = 0; l < N; l++) s = fma(JP[i * N + l], JP[j * N + l], s);
			AB[i * M + j] = s;
		}
		chol<M>(AB, LA, dA);
		UNROLL for (int c = 0; c < M; c++) y[c] = vf[c];
		solve_lower<M>(LA, dA, y);
		solve_lower_t<M>(LA, dA, y);
		real w[N];
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int c = 0; c < M; c++) s = fma(JP[c * N + i], y[c], s);
			w[i] = s;
		}
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int k = 0; k <= i; k++) s = fma(f.lb[(i * (i + 1) / 2 + k) * 64], w[k], s);
			tau[i] += s;
		}
	}
	SAI2B_PHASE();
	CSTAMP(22);
	// Y = Z R by modified Gram-Schmidt (Z overwrites Y); R upper triangular, rinv its reciprocal diagonal
	real R[M * M], rinv[M];
	UNROLL for (int j = 0; j < M; j++) {
		real nn = 0;
		UNROLL for (int i = 0; i < N; i++) nn = fma(Y[j * N + i], Y[j * N + i], nn);
		const real r = rsqrt(nn);
		rinv[j] = r;
		UNROLL for (int i = 0; i < N; i++) Y[j * N + i] *= r;
		UNROLL for (int k = j + 1; k < M; k++) {
			real s = 0;
			UNROLL for (int i = 0; i < N; i++) s = fma(Y[j * N + i], Y[k * N + i], s);
			R[j * M + k] = s;
			UNROLL for (int i = 0; i < N; i++) Y[k * N + i] = fma(-s, Y[j * N + i], Y[k * N + i]);
		}
	}
	if (TORQUE && (has_va || full)) {  // Jp^T Lambda a = L Y R^-1 R^-T a = L Z (R^-T a)
		real u[M], w[N];
		UNROLL for (int j = 0; j < M; j++) {
			real t = (has_va ? va[j] : 0.0) + (full ? vf[j] : 0.0);
			UNROLL for (int i = 0; i < j; i++) t = fma(-R[i * M + j], u[i], t);
			u[j] = t * rinv[j];
		}
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int c = 0; c < M; c++) s = fma(Y[c * N + i], u[c], s);
			w[i] = s;
		}
		UNROLL for (int i = 0; i < N; i++) {
			real s = 0;
			UNROLL for (int k = 0; k <= i; k++) s = fma(f.L[i * N + k], w[k], s);
			tau[i] += s;
		}
	}
	CSTAMP(23);
	if (!last) {
		UNROLL for (int c = 0; c < M; c++)
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-Y[c * N + i], Y[c * N + j], Q[i * N + j]);
	}
	CSTAMP(24);
	return ok;
}
// run-time row count (the same for every robot) -> the instantiation
template <int M, bool TORQUE = true, bool SING = false>
DI bool level_any(int m, const Fact& f, const real* Jr, bool first, bool last, bool do_cert, real abs2, real rel2, int decoupling,
				  bool has_va, const real* va, const real* vf, const real* vd, real* Q, real* tau, const SingArgs* sa = nullptr) {
	if constexpr (M > 1) {
		if (m < M) return level_any<M - 1, TORQUE, SING>(m, f, Jr, first, last, do_cert, abs2, rel2, decoupling, has_va, va, vf, vd, Q, tau, sa);
	}
#ifndef SAI2B_NO_STREAMED_LEVEL	 // A/B (scripts/micro/cert_variants.sh)
	if constexpr (M > 3) return level_streamed<M, TORQUE, SING>(f, Jr, first, last, do_cert, abs2, rel2, decoupling, has_va, va, vf, vd, Q, tau, sa);
#endif
	return level<M, TORQUE, SING>(f, Jr, first, last, do_cert, abs2, rel2, decoupling, has_va, va, vf, vd, Q, tau, sa);
}

// A full JointTask behind other tasks: Jp = N_prec = L^-T Q L^T with rank D = n - (rows of the certified tasks
// above) >= 1 (JointTask.cpp:218-283; the reference's 1e-3 range rule sees singular values >= 1 and ~1e-16).
// va: goal acceleration minus the compensation of the tasks above, vf: the PD(+I) unit torques.
template <int D>
DI void full_joint_task_behind(const Fact& f, int decoupling, const real* va, const real* vf, real* Q, real* tau) {
	// basis of range(Q): pivoted Cholesky of the projector, Q = C C^T with orthonormal columns
	real C[D * N];
	UNROLL for (int s = 0; s < D; s++) {
		real best = -1.0;
		int ks = 0;
		UNROLL for (int i = 0; i < N; i++) {
			const bool take = Q[i * N + i] > best;
			best = take ? Q[i * N + i] : best;
			ks = take ? i : ks;
		}
		// column ks of Q as a product with the unit vector (selects between array elements would turn into
		// run-time indexing and send Q to scratch memory)
		const real rn = rsqrt(best);
		real e[N];
		UNROLL for (int j = 0; j < N; j++) e[j] = (ks == j) ? rn : 0.0;
		UNROLL for (int i = 0; i < N; i++) {
			real v = 0;
			UNROLL for (int j = 0; j < N; j++) v = fma(symat(Q, i, j), e[j], v);
			C[s * N + i] = v;
		}
		if (s + 1 < D) {
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = fma(-C[s * N + i], C[s * N + j], Q[i * N + j]);
		}
	}
	SAI2B_PHASE();
	real A[D * N], K[D * D], LK[D * D], dK[D];
	UNROLL for (int s = 0; s < D; s++) {
		UNROLL for (int i = 0; i < N; i++) A[s * N + i] = C[s * N + i];
		solve_lower_t<N>(f.L, f.dL, A + s * N);	 // A = L^-T C
	}
	mm_nt_sym<D, N>(A, A, K);
	chol<D>(K, LK, dK);
	const bool full = decoupling == SAI2B_FULL_DYNAMIC_DECOUPLING;
	real h[D];
	UNROLL for (int s = 0; s < D; s++) {  // A^T (va (+ vf))
		real a = 0;
		UNROLL for (int i = 0; i < N; i++) a = fma(A[s * N + i], va[i] + (full ? vf[i] : 0.0), a);
		h[s] = a;
	}
	solve_lower<D>(LK, dK, h);
	solve_lower_t<D>(LK, dK, h);
	if (decoupling == SAI2B_IMPEDANCE) {  // R R^T f  ->  C^T L^-1 f
		real u[N];
		UNROLL for (int i = 0; i < N; i++) u[i] = vf[i];
		solve_lower<N>(f.L, f.dL, u);
		UNROLL for (int s = 0; s < D; s++) {
			real a = 0;
			UNROLL for (int i = 0; i < N; i++) a = fma(C[s * N + i], u[i], a);
			h[s] += a;
		}
	} else if (decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {	 // Kb^-1 K^-1 A^T f, Kb = (LB^-1 L C)^T (LB^-1 L C)
		real g[D], Kb[D * D], LKb[D * D], dKb[D], LB[N * N], dB[N];
		load_lb(f.lb, LB, dB);
		UNROLL for (int s = 0; s < D; s++) {
			real a = 0;
			UNROLL for (int i = 0; i < N; i++) a = fma(A[s * N + i], vf[i], a);
			g[s] = a;
		}
		solve_lower<D>(LK, dK, g);
		solve_lower_t<D>(LK, dK, g);
		UNROLL for (int s = 0; s < D; s++) {  // A is dead: reuse it for LB^-1 L C
			UNROLL for (int i = 0; i < N; i++) {
				real a = 0;
				UNROLL for (int k = 0; k <= i; k++) a = fma(f.L[i * N + k], C[s * N + k], a);
				A[s * N + i] = a;
			}
			solve_lower<N>(LB, dB, A + s * N);
		}
		mm_nt_sym<D, N>(A, A, Kb);
		chol<D>(Kb, LKb, dKb);
		solve_lower<D>(LKb, dKb, g);
		solve_lower_t<D>(LKb, dKb, g);
		UNROLL for (int s = 0; s < D; s++) h[s] += g[s];
	}
	real w[N];
	UNROLL for (int i = 0; i < N; i++) {
		real a = 0;
		UNROLL for (int s = 0; s < D; s++) a = fma(C[s * N + i], h[s], a);
		w[i] = a;
	}
	UNROLL for (int i = 0; i < N; i++) {  // Jp^T x = L Q L^-1 x = L C (...)
		real a = 0;
		UNROLL for (int k = 0; k <= i; k++) a = fma(f.L[i * N + k], w[k], a);
		tau[i] += a;
	}
}
template <int D>
DI void full_joint_task_behind_any(int d, const Fact& f, int decoupling, const real* va, const real* vf, real* Q, real* tau) {
	if constexpr (D > 1) {
		if (d < D) return full_joint_task_behind_any<D - 1>(d, f, decoupling, va, vf, Q, tau);
	}
	full_joint_task_behind<D>(f, decoupling, va, vf, Q, tau);
}

// JWorldFrame(link, pos) and the pose of the control frame from the joint positions (fk + frame_pose + jacobian of
// sai2b_device.hpp in one sweep that keeps only the joint axes and origins, not every link frame)
template <class MD>
DI void jacobian_and_pose(const MD& md, const DevTask& t, const real* q, const real* sc, real* J, real* x, real* R) {
	real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
	real z[N][3], o[N][3];
	UNROLL for (int k = 0; k < 3; k++) x[k] = 0;
	UNROLL for (int k = 0; k < 9; k++) R[k] = 0;
	UNROLL for (int i = 0; i < N; i++) {
		real RE[9], pi[3];
		UNROLL for (int k = 0; k < 3; k++)
			pi[k] = fma(Rp[3 * k], md.xyz[i][0], fma(Rp[3 * k + 1], md.xyz[i][1], fma(Rp[3 * k + 2], md.xyz[i][2], pp[k])));
		mm<3, 3, 3>(Rp, md.E[i], RE);
		const real s = sc[2 * i], c = sc[2 * i + 1];
		const bool pris = md.jtype[i] != 0;
		UNROLL for (int k = 0; k < 3; k++) {
			Rp[3 * k + 0] = fma(c, RE[3 * k], s * RE[3 * k + 1]);
			Rp[3 * k + 1] = fma(c, RE[3 * k + 1], -s * RE[3 * k]);
			Rp[3 * k + 2] = RE[3 * k + 2];
			if (pris) pi[k] = fma(q[i], RE[3 * k + 2], pi[k]);
			pp[k] = pi[k];
			z[i][k] = RE[3 * k + 2];
			o[i][k] = pi[k];
		}
		if (t.link == i) {	// batch-uniform
			UNROLL for (int k = 0; k < 3; k++)
				x[k] = fma(Rp[3 * k], t.frame_pos[0], fma(Rp[3 * k + 1], t.frame_pos[1], fma(Rp[3 * k + 2], t.frame_pos[2], pp[k])));
			mm<3, 3, 3>(Rp, t.frame_rot, R);
		}
	}
	UNROLL for (int i = 0; i < N; i++) {
		const real d[3] = {x[0] - o[i][0], x[1] - o[i][1], x[2] - o[i][2]};
		real v[3];
		cross3(z[i], d, v);
		const bool on = i <= t.link, pris = md.jtype[i] != 0;
		UNROLL for (int k = 0; k < 3; k++) {
			J[k * N + i] = on ? (pris ? z[i][k] : v[k]) : 0.0;
			J[(3 + k) * N + i] = (on && !pris) ? z[i][k] : 0.0;
		}
	}
}

// Pose of the control frame and its velocity J dq in one sweep over the joints, without the Jacobian itself
// (v = omega x x_frame - sum dq_i z_i x o_i + sliding terms): the control law runs on this, and the Jacobian's 42
// numbers only come to life afterwards (jacobian_and_pose), when the law's own ~100 are gone.
// sc: sines and cosines of the joints, kept for the second sweep.
template <class MD>
DI void pose_and_velocity(const MD& md, const DevTask& t, const real* q, const real* dq, real* x, real* R, real* vw, real* sc) {
	real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
	real om[3] = {0, 0, 0}, a[3] = {0, 0, 0}, vs[3] = {0, 0, 0};
	UNROLL for (int k = 0; k < 3; k++) x[k] = 0;
	UNROLL for (int k = 0; k < 9; k++) R[k] = 0;
	UNROLL for (int i = 0; i < N; i++) {
		real RE[9], pi[3];
		UNROLL for (int k = 0; k < 3; k++)
			pi[k] = fma(Rp[3 * k], md.xyz[i][0], fma(Rp[3 * k + 1], md.xyz[i][1], fma(Rp[3 * k + 2], md.xyz[i][2], pp[k])));
		mm<3, 3, 3>(Rp, md.E[i], RE);
		real s, c;
		sincos_joint(q[i], &s, &c);
		const bool pris = md.jtype[i] != 0;
		if (pris) s = 0, c = 1;
		sc[2 * i] = s, sc[2 * i + 1] = c;
		UNROLL for (int k = 0; k < 3; k++) {
			Rp[3 * k + 0] = fma(c, RE[3 * k], s * RE[3 * k + 1]);
			Rp[3 * k + 1] = fma(c, RE[3 * k + 1], -s * RE[3 * k]);
			Rp[3 * k + 2] = RE[3 * k + 2];
			if (pris) pi[k] = fma(q[i], RE[3 * k + 2], pi[k]);
			pp[k] = pi[k];
		}
		if (i <= t.link) {	// batch-uniform
			const real z[3] = {RE[2], RE[5], RE[8]};
			real zo[3];
			cross3(z, pi, zo);
			UNROLL for (int k = 0; k < 3; k++) {
				if (pris) {
					vs[k] = fma(dq[i], z[k], vs[k]);
				} else {
					om[k] = fma(dq[i], z[k], om[k]);
					a[k] = fma(dq[i], zo[k], a[k]);
				}
			}
		}
		if (t.link == i) {
			UNROLL for (int k = 0; k < 3; k++)
				x[k] = fma(Rp[3 * k], t.frame_pos[0], fma(Rp[3 * k + 1], t.frame_pos[1], fma(Rp[3 * k + 2], t.frame_pos[2], pp[k])));
			mm<3, 3, 3>(Rp, t.frame_rot, R);
		}
	}
	real ox[3];
	cross3(om, x, ox);
	UNROLL for (int k = 0; k < 3; k++) {
		vw[k] = ox[k] - a[k] + vs[k];
		vw[3 + k] = om[k];
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
// MD: where the robot constants come from (the parameter block, or the compile-time Panda of sai2b_baked_panda.h)
// TASK = true: the TemplateTask calls on ONE task under a caller-supplied N_prec (TemplateTask.h:42-88; round 3). The
// cascade needs N_prec in whitened form, N_prec = L^-T Q L^T with Q an orthogonal projector: Q = L^T N_prec L^-T is
// computed and CHECKED (symmetric and idempotent to 1e-9, integer trace) — every dynamically consistent nullspace a
// chain of these calls produces passes, anything else sends the robot to the generic kernel like an uncertified level.
// Outputs besides the torques: the task's nullspace N = L^-T (I - Q0 + Q1) L^T and N N_prec = L^-T Q1 L^T.
struct TaskArgs {
	int task;
	const real* Nprec;	   // [N * N][B], NULL: the identity a task is constructed with (JointTask.cpp:62, MotionForceTask.cpp:138)
	const real* tau_prec;  // [N][B] or NULL (the no-argument computeTorques)
	real* N_out;		   // [N * N][B] or NULL
	real* Ntot_out;		   // [N * N][B] or NULL
	real* q0;			   // this lane's column of the LDS area (stride 64); slot of entry k: q0_slot(k): Q before the level
	int write_active;	   // model update of a gated JointTask: OTG_ACTIVE (does the task have a range this tick?) for otg_kernel
};

// where the N (N + 1) / 2 entries of Q0 wait in the lane's LDS column: the slots a one-task call leaves unused — the
// gravity torques' [0, N) and what lies behind the one task's deferred stores [N + 12, PEND_SLOTS) — then TASK_EXTRA
// slots behind the factor of the bounded inertia estimate (4 for 7 joints: four workgroups per CU still fit)
constexpr int TASK_FREE = N + (PEND_SLOTS - N - 12);
constexpr int TASK_EXTRA = (N * (N + 1) / 2 > TASK_FREE) ? N * (N + 1) / 2 - TASK_FREE : 0;
DI int q0_slot(int k) { return k < N ? k : (k < TASK_FREE ? k + 12 : LDS_SLOTS + (k - TASK_FREE)); }

// sp: the bookkeeping of a MotionForceTask that went through the singular branch (the caller flushes it with the rest);
// NULL: such robots go to the work list (MCAP = 6, the task-level calls, SAI2B_NO_INLANE_SINGULAR)
// S6: the instantiation whose 4- to 6-row MotionForceTasks have the singular branch too (singular_streamed). A kernel of its
// own, chosen by the host while many robots are inside a blending region (sai2b_host.cpp: launch_tick): the branch's
// registers cost the REGULAR path of the 6-row kernel 45 -> 75 us, inlined, and more as a call.
template <int MCAP, int DCAP, class MD, bool TASK = false, bool S6 = false>
DI bool tick(const DevParams& P, const MD& md, int B, int b, bool with_comp, real* pend, real* tau, const TaskArgs* io = nullptr,
			 SingPend* sp = nullptr) {
	CSTAMP(0);
	Fact f;
	bool ok = true;
	int np = N;	 // deferred-store slots used so far: 0..N-1 hold the gravity torques
	{
		// Sai2Model::updateModel(): kinematics, M (CRBA) and its factor (examples/05-using_robot_controller.cpp:143-145)
		real q[N];
		UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
		Frames F;
		CSTAMP(1);
		fk(md, q, F);
		CSTAMP(2);
		real M[N * N];
		mass_matrix(md, F, M);
		CSTAMP(3);
		if (P.gravity_comp) {
			real g[N];
			gravity_vector(md, F, g);
			UNROLL for (int i = 0; i < N; i++) pend[i * 64] = g[i];
		} else {
			UNROLL for (int i = 0; i < N; i++) pend[i * 64] = 0.0;
		}
		SAI2B_PHASE();
		chol<N>(M, f.L, f.dL);
		// bounded inertia estimate, shared by the tasks that ask for it (host: one threshold, upload_params)
		const bool any_bie = P.any_bie != 0;
		const real thr = P.bie_thr;
		if (any_bie) {
			real LB[N * N], dB[N];
			UNROLL for (int i = 0; i < N; i++) M[i * N + i] = fmax(M[i * N + i], thr);
			chol<N>(M, LB, dB);
			store_lb(pend + PEND_SLOTS * 64, LB, dB);
		}
		f.lb = pend + PEND_SLOTS * 64;
	}
	SAI2B_PHASE();
	CSTAMP(4);
	real Q[N * N];
	UNROLL for (int i = 0; i < N * N; i++) Q[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
	UNROLL for (int i = 0; i < N; i++) tau[i] = 0;
	int wrows = 0;	// rows of the certified tasks so far (batch-uniform)
	bool task_first = true;	 // TASK: N_prec is the identity
	if constexpr (TASK) {
		if (io->tau_prec) {
			UNROLL for (int i = 0; i < N; i++) tau[i] = ld(io->tau_prec, i, B, b);
		}
		if (io->Nprec) {
			// Q = L^T N_prec L^-T, row by row: A = L^T N_prec, then L Q[i, :]^T = A[i, :]^T
			real Np[N * N], Qf[N * N];
			UNROLL for (int i = 0; i < N * N; i++) Np[i] = ld(io->Nprec, i, B, b);
			UNROLL for (int i = 0; i < N; i++) {
				real row[N];
				UNROLL for (int j = 0; j < N; j++) {
					real s = 0;
					UNROLL for (int k = i; k < N; k++) s = fma(f.L[k * N + i], Np[k * N + j], s);
					row[j] = s;
				}
				solve_lower<N>(f.L, f.dL, row);
				UNROLL for (int j = 0; j < N; j++) Qf[i * N + j] = row[j];
			}
			real asym = 0, tr = 0;
			UNROLL for (int i = 0; i < N; i++) {
				tr += Qf[i * N + i];
				UNROLL for (int j = 0; j < i; j++) {
					asym = fmax(asym, fabs(Qf[i * N + j] - Qf[j * N + i]));
					Qf[i * N + j] = Qf[j * N + i] = 0.5 * (Qf[i * N + j] + Qf[j * N + i]);
				}
			}
			real idem = 0;
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) {
				real s = -Qf[i * N + j];
				UNROLL for (int k = 0; k < N; k++) s = fma(Qf[i * N + k], Qf[k * N + j], s);
				idem = fmax(idem, fabs(s));
			}
			const int rk = (int)(tr + 0.5);
			ok = ok && asym < 1e-9 && idem < 1e-9 && fabs(tr - (real)rk) < 1e-6 && rk >= 0 && rk <= N;	 // (NaNs fail too)
			wrows = (rk >= 0 && rk <= N) ? N - rk : 0;
			task_first = wrows == 0;  // an N_prec that IS the identity
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) Q[i * N + j] = Qf[i * N + j];
		}
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) io->q0[q0_slot(i * (i + 1) / 2 + j) * 64] = Q[i * N + j];
	}
#pragma unroll 1
	for (int ti = (TASK ? io->task : 0); ti < (TASK ? io->task + 1 : P.n_tasks); ti++) {
		const DevTask& t = P.task[ti];
		const bool first = TASK ? task_first : (ti == 0), last = TASK ? false : (ti == P.n_tasks - 1);
		if (t.type == SAI2B_MOTION_FORCE_TASK) {
			// MotionForceTask::updateTaskModel / computeTorques (MotionForceTask.cpp:247-509) in the fully
			// non-singular branch of the SingularityHandler (SingularityHandler.cpp:100-141,307-309)
			CSTAMP(10);
			constexpr bool SING = MCAP <= 3 || S6;
			const bool inlane = SING && sp != nullptr;
			const int prev_types = ldi(t.istate, IS_NTYPES, B, b);
			if (!inlane) ok = ok && (prev_types == 0);
			real Jw[6 * N], Fu[6], Ff[6];
			{
				real q[N], sc[2 * N], x[3], R[9];
				UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
				{
					real dq[N], vw0[6], vw[6];
					UNROLL for (int i = 0; i < N; i++) dq[i] = ld(P.dq, i, B, b);
					MftIn in;
					mft_load(t, B, b, in);	// issued ahead of the sweep: it arrives while the kinematics run
					CSTAMP(11);
					pose_and_velocity(md, t, q, dq, x, R, vw0, sc);
					CSTAMP(12);
					if (t.full_projection) {
						UNROLL for (int i = 0; i < 6; i++) vw[i] = vw0[i];
					} else {
						mv<6, 6>(t.P, vw0, vw);
					}
					mft_law_vw(t, vw, vw + 3, x, R, in, Fu, Ff);
					// mft_store_integrators, deferred
					UNROLL for (int k = 0; k < 12; k++) pend[(np + k) * 64] = in.integ[k];
					np += 12;
				}
				SAI2B_PHASE();
				CSTAMP(13);
				jacobian_and_pose(md, t, q, sc, Jw, x, R);
				if constexpr (!TASK && MCAP <= 3 && POSE_SLOTS == 9) {
					UNROLL for (int k = 0; k < 3; k++) {
						pend[(LDS_SLOTS + k) * 64] = x[k];
						pend[(LDS_SLOTS + 3 + k) * 64] = R[3 * k];
						pend[(LDS_SLOTS + 6 + k) * 64] = R[3 * k + 1];
					}
				}
				CSTAMP(14);
			}
			const int m = t.rank;
			if (wrows + m > N) ok = false;	// more task rows than joints left: never full rank
			const real abs2 = t.s_abs_tol * t.s_abs_tol, rel2 = t.s_max * t.s_max;
			const real zero[6] = {0, 0, 0, 0, 0, 0};
			SingArgs sa;
			SingPend none;
			none.task = 0;	// "taken": the singular branch declines
			sa.P = &P, sa.t = &t, sa.ti = ti, sa.B = B, sa.b = b, sa.enabled = inlane, sa.pu = nullptr, sa.sp = inlane ? sp : &none, sa.pose = (TASK || MCAP > 3) ? nullptr : pend + LDS_SLOTS * 64;
			const int task_before = sa.sp->task;
			{
				real nn = 0;
				UNROLL for (int i = 0; i < 6; i++) nn = fma(Fu[i] + Ff[i], Fu[i] + Ff[i], nn);
				sa.fnorm = sqrt(nn);
			}
			bool c_ok;
			if (t.full_projection || t.p_lead < 6) {  // range(P) = the leading coordinates: rows and forces as they are
				c_ok = level_any<(MCAP < 6 ? MCAP : 6), true, SING>(m, f, Jw, first, last, true, abs2, rel2, t.decoupling, false, zero, Fu, Ff, Q, tau, &sa);
			} else {  // rows PU^T J, forces PU^T F
				real Jr[6 * N], fu[6], ff[6];
				mm_tn<6, 6, N>(t.PU, Jw, Jr);
				mv_t<6, 6>(t.PU, Fu, fu);
				mv_t<6, 6>(t.PU, Ff, ff);
				sa.pu = t.PU;
				c_ok = level_any<(MCAP < 6 ? MCAP : 6), true, SING>(m, f, Jr, first, last, true, abs2, rel2, t.decoupling, false, zero, fu, ff, Q, tau, &sa);
			}
			ok = ok && c_ok;
			if (inlane && prev_types != 0 && sp->task == task_before) {
				// certified, with singularity history: the robot has left the region and the history goes (:239-245)
				if (sp->task >= 0) {
					ok = false;
				} else if (sp->commit) {
					sp->task = ti, sp->clear = 1;
				}
			}
			wrows += m;
		} else {
			// JointTask::updateTaskModel / computeTorques (JointTask.cpp:218-356)
			CSTAMP(30);
			const int k0 = t.k0;
			if constexpr (TASK) {
				// what the generic kernel's model pass tells the generator kernels (sai2b_device.hpp: jt_task): a full
				// JointTask has a range while N_prec leaves a direction, a certified partial one keeps all its rows (an
				// uncertified one is decided — and this row overwritten — by the generic pass behind)
				if (t.otg_gated && io->write_active) st(t.otg_state, OTG_ACTIVE, B, b, (!t.full_selection || first || wrows < N) ? 1.0 : 0.0);
			}
			real va[N], vf[N];
			{
				// goals and integrators first, in one burst (rows clamped into the task's own: the loads must not sit
				// behind the per-coordinate branches), then the state; they arrive behind the compensation solves
				real gq[N], gdq[N], gddq[N], integ0[N];
				UNROLL for (int i = 0; i < N; i++) {
					const int r = i < k0 ? i : k0 - 1;
					gq[i] = ld(t.law_goals, r, B, b);
					gdq[i] = ld(t.law_goals, k0 + r, B, b);
					gddq[i] = ld(t.law_goals, 2 * k0 + r, B, b);
					integ0[i] = ld(t.state, r, B, b);
				}
				real cur[N], vel[N], comp[N];
				{
					real q[N], dq[N];
					UNROLL for (int i = 0; i < N; i++) {
						q[i] = ld(P.q, i, B, b);
						dq[i] = ld(P.dq, i, B, b);
					}
					UNROLL for (int i = 0; i < N; i++) comp[i] = 0;
					if (with_comp && (!first || TASK)) {	// JointTask.cpp:285-292: S M^-1 tau_prec
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
					if (t.full_selection) {
						UNROLL for (int i = 0; i < N; i++) cur[i] = q[i], vel[i] = dq[i];
					} else {
						mv<N, N>(t.S, q, cur);
						mv<N, N>(t.S, dq, vel);
					}
				}
				UNROLL for (int i = 0; i < N; i++) {
					va[i] = vf[i] = 0;
					if (i < k0) {  // PD(+I) law of task coordinate i (JointTask.cpp:299-345)
						const real qd = gq[i], dqd = gdq[i], ddq_d = gddq[i];
						const real integ = fma(cur[i] - qd, t.dt, integ0[i]);
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
			}
			CSTAMP(31);
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
						real LB[N * N], dB[N];
						load_lb(f.lb, LB, dB);
						UNROLL for (int i = 0; i < N; i++) x1[i] = vf[i];
						mul_llt(LB, x1, y);
						UNROLL for (int i = 0; i < N; i++) tau[i] += y[i];
					}
				} else if (wrows < N) {
					full_joint_task_behind_any<(DCAP < DM ? DCAP : DM)>(N - wrows, f, t.decoupling, va, vf, Q, tau);
				}
				// a full JointTask leaves nothing to the tasks below it
				UNROLL for (int i = 0; i < N * N; i++) Q[i] = 0;
				wrows = N;
			} else {
				if (wrows + k0 > N) ok = false;
				const real zero[N] = {};
				// a first-level selection has full row rank by construction (validated on the host); behind other
				// tasks: s_0 >= 1e-3 and s_i / s_0 >= 1e-3, the range rule keeps every row
				const bool c_ok = level_any<(MCAP < N ? MCAP : N)>(k0, f, t.S, first, last, !first, 1e-6, 1e-6, t.decoupling, true, va, vf, zero, Q, tau);
				ok = ok && c_ok;
				wrows += k0;
			}
		}
		if (wrows > N) wrows = N;
		CSTAMP(40);
	}
	if constexpr (TASK) {
		// getTaskNullspace / getTaskAndPreviousNullspace (TemplateTask.h:73-88) of a robot that finishes here: X = L^-T S L^T
		// for S = I - Q0 + Q1 and S = Q1, column by column (T[:, j] = S L^T e_j, then L^T X[:, j] = T[:, j])
		if (ok) {
			UNROLL for (int which = 0; which < 2; which++) {
				real* out = which ? io->Ntot_out : io->N_out;
				if (!out) continue;
				real S[N * N];
				UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j <= i; j++) {
					const real q1 = Q[i * N + j];
					S[i * N + j] = which ? q1 : ((i == j ? 1.0 : 0.0) - io->q0[q0_slot(i * (i + 1) / 2 + j) * 64] + q1);
				}
				UNROLL for (int j = 0; j < N; j++) {
					real col[N];
					UNROLL for (int i = 0; i < N; i++) {
						real s = 0;
						UNROLL for (int k = 0; k <= j; k++) s = fma(symat(S, i, k), f.L[j * N + k], s);
						col[i] = s;
					}
					solve_lower_t<N>(f.L, f.dL, col);
					UNROLL for (int i = 0; i < N; i++) st(out, i * N + j, B, b, col[i]);
				}
			}
		}
	}
	return ok;
}

// TASK: write the deferred integrators of the one task that ran (slots from N on, as tick() filled them)
DI void flush_task(const DevParams& P, int task, int B, int b, const real* pend) {
	const DevTask& t = P.task[task];
	const int np = N;
	if (t.type == SAI2B_MOTION_FORCE_TASK) {
		for (int k = 0; k < 6; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
		if (t.cl_force)
			for (int k = 6; k < 9; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
		if (t.cl_moment)
			for (int k = 9; k < 12; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
	} else {
		for (int k = 0; k < t.k0; k++) st(t.state, k, B, b, pend[(np + k) * 64]);
	}
}

// The range pass ahead of the trajectory generators for the robots this kernel family can vouch for: a JointTask whose
// range can come out empty leaves its generator alone on such a tick (JointTask.cpp:233-239, 302-306), so the generator
// kernels need "active this tick" per robot before the tick itself runs (DevTask::otg_gated, row OTG_ACTIVE). The model
// and the cascade of tick() without laws, torques or stores, down to the last gated task: a partial JointTask whose
// level carries its certificate has all its rows in its range (active); a full one at the bottom is active while
// rows are left. Returns false when some level could not be certified: the caller hands the robot to the range pass
// of the generic kernel, which decides (and overwrites) with the reference's own rule.
// inlane: robots inside a blending region of a 2- or 3-row MotionForceTask stay (singular_range), as in tick()
template <int MCAP, class MD>
DI bool range_tick(const DevParams& P, const MD& md, int B, int b, bool inlane = false) {
	Fact f;
	f.lb = nullptr;
	real q[N];
	UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
	{
		Frames F;
		fk(md, q, F);
		real M[N * N];
		mass_matrix(md, F, M);
		SAI2B_PHASE();
		chol<N>(M, f.L, f.dL);
	}
	SAI2B_PHASE();
	real Q[N * N], tau[N];
	UNROLL for (int i = 0; i < N * N; i++) Q[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
	UNROLL for (int i = 0; i < N; i++) tau[i] = 0;
	int n_run = 0;
	for (int ti = 0; ti < P.n_tasks; ti++)
		if (P.task[ti].otg_gated) n_run = ti + 1;
	bool ok = true;
	int wrows = 0;
	const real zero[N] = {};
#pragma unroll 1
	for (int ti = 0; ti < n_run; ti++) {
		const DevTask& t = P.task[ti];
		const bool first = (ti == 0), last = (ti == n_run - 1);
		if (t.type == SAI2B_MOTION_FORCE_TASK) {
			const int n_types = ldi(t.istate, IS_NTYPES, B, b);
			real Jw[6 * N];
			{
				real sc[2 * N], x[3], R[9];
				UNROLL for (int i = 0; i < N; i++) {
					real s_, c_;
					sincos_joint(q[i], &s_, &c_);
					const bool pris = md.jtype[i] != 0;
					sc[2 * i] = pris ? 0.0 : s_, sc[2 * i + 1] = pris ? 1.0 : c_;
				}
				jacobian_and_pose(md, t, q, sc, Jw, x, R);
			}
			const int m = t.rank;
			if (wrows + m > N) ok = false;
			const real abs2 = t.s_abs_tol * t.s_abs_tol, rel2 = t.s_max * t.s_max;
			bool c_ok;
			constexpr bool SING = MCAP <= 3;
			SingArgs sa;
			sa.P = &P, sa.t = &t, sa.ti = ti, sa.B = B, sa.b = b, sa.enabled = SING && inlane, sa.fnorm = 0, sa.pu = nullptr, sa.pose = nullptr, sa.sp = nullptr;
			if (t.full_projection || t.p_lead < 6) {
				c_ok = level_any<(MCAP < 6 ? MCAP : 6), false, SING>(m, f, Jw, first, last, true, abs2, rel2, t.decoupling, false, zero, zero, zero, Q, tau, &sa);
			} else {
				real Jr[6 * N];
				mm_tn<6, 6, N>(t.PU, Jw, Jr);
				c_ok = level_any<(MCAP < 6 ? MCAP : 6), false, SING>(m, f, Jr, first, last, true, abs2, rel2, t.decoupling, false, zero, zero, zero, Q, tau, &sa);
			}
			// (singularity history does not enter a range decision; without the in-lane branch such a robot is the generic pass's)
			ok = ok & c_ok & (sa.enabled || n_types == 0);
			wrows += m;
		} else if (t.full_selection) {
			if (t.otg_gated) st(t.otg_state, OTG_ACTIVE, B, b, wrows < N ? 1.0 : 0.0);
			UNROLL for (int i = 0; i < N * N; i++) Q[i] = 0;
			wrows = N;
		} else {
			const int k0 = t.k0;
			if (wrows + k0 > N) ok = false;
			const bool c_ok = level_any<(MCAP < N ? MCAP : N), false>(k0, f, t.S, first, last, !first, 1e-6, 1e-6, t.decoupling, false, zero, zero, zero, Q, tau);
			ok = ok & c_ok;
			if (t.otg_gated) st(t.otg_state, OTG_ACTIVE, B, b, 1.0);
			wrows += k0;
		}
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
