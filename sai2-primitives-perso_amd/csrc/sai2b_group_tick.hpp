// sai2b_group_tick.hpp — the generic tick (any hierarchy, full SingularityHandler) with a robot spread over a
// group of G lanes (sai2b_group.hpp). Same mathematics, projector form and reference citations as the
// one-lane-per-robot functions of sai2b_device.hpp (mft_task / jt_task), re-expressed row-distributed:
//   * joint l lives in lane l: FK is a 3-step scan of link transforms over the lanes, lane l holds link l's
//     world frame, Jacobian column, spatial inertia (CRBA composite inertias = a suffix scan);
//   * an n x n (or 6 x n) matrix has row r in lane r; products are the fused broadcast-FMA blocks;
//   * the hierarchy carries N_prec TRANSPOSED (lane r holds column r of N_prec), because every product that needs
//     it has it on the left of a transpose:  Jp^T = N_prec^T J^T,  N_prec'^T = N_prec^T N^T;
//   * per-robot branches (certified / Jacobi SVD / singular handling) are uniform inside a group, so a wavefront
//     diverges between robots, never inside one — DPP reads never touch a masked-off lane of the own group.
// Register footprint per lane is ~1/8 of the one-lane-per-robot kernel's: no scratch, several wavefronts per SIMD.
#pragma once
#include "sai2b_group.hpp"

namespace sai2b {
namespace grp {

DI real kd(int r, int j) { return r == j ? 1.0 : 0.0; }
// `on ? v : 0` where v is a memory operand: the load itself must not be conditional (the compiler cannot speculate
// it and would wrap every single one in an exec-mask branch); indices are clamped by the caller
DI real sel0(bool on, real v) { return on ? v : 0.0; }
#define SAI2B_TASK_FN DI
// scheduling fence + a named comment in the ISA (static per-phase instruction counts: scripts/count_group_phases.py).
// -DSAI2B_GROUP_STAMP: lane 0 of workgroup 0 also records (id, s_memtime) at every mark into g_stamps (read back
// by scripts/micro/group_stamps.py): where one robot's cycles go.
#ifdef SAI2B_GROUP_STAMP
__device__ unsigned long long g_stamps[512];
__device__ int g_stamp_n;
#define GMARK(id, name)                                                                        \
	do {                                                                                       \
		__builtin_amdgcn_sched_barrier(0);                                                     \
		if (blockIdx.x == 0 && threadIdx.x == 0) {                                             \
			const int k_ = g_stamp_n;                                                          \
			if (k_ < 255) {                                                                    \
				g_stamps[2 * k_] = (unsigned long long)(id);                                   \
				g_stamps[2 * k_ + 1] = __builtin_readcyclecounter();                           \
				g_stamp_n = k_ + 1;                                                            \
			}                                                                                  \
		}                                                                                      \
		__builtin_amdgcn_sched_barrier(0);                                                     \
	} while (0)
#elif defined(SAI2B_GROUP_MARKS)
#define GMARK(id, name) do { __builtin_amdgcn_sched_barrier(0); asm volatile("; GMARK " name); __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define GMARK(id, name) do { } while (0)
#endif

// LDS scratch of a group: the largest transposition is max(N, 6) square, rows padded to an odd length
constexpr int PAD_N = N > 6 ? N : 6;
constexpr int PAD_DOUBLES = PAD_N * (PAD_N | 1);

// per-lane view of one robot
struct Rob {
	int r;			 // lane within the group
	int B, b;		 // batch size, robot index
	real q, dq;		 // this lane's joint (0 beyond the last joint)
	real FR[9], Fp[3];	// world frame of this lane's link
	real minv[N], minvB[N];	 // row r of M^-1 and of the bounded-inertia M_BIE^-1
	real* pad;		 // the group's LDS scratch (PAD_DOUBLES)
};

// ------------------------------------------------------------------ forward kinematics: scan over the lanes
template <int G, int K>
DI void fk_scan_step(real* R, real* p) {
	real PR[9], Pp[3], NR[9], np[3];
	UNROLL for (int i = 0; i < 9; i++) PR[i] = shift_up<G, K>(R[i], (i % 4 == 0) ? 1.0 : 0.0);
	UNROLL for (int i = 0; i < 3; i++) Pp[i] = shift_up<G, K>(p[i], 0.0);
	mm<3, 3, 3>(PR, R, NR);
	UNROLL for (int k = 0; k < 3; k++) np[k] = fma(PR[3 * k], p[0], fma(PR[3 * k + 1], p[1], fma(PR[3 * k + 2], p[2], Pp[k])));
	UNROLL for (int i = 0; i < 9; i++) R[i] = NR[i];
	UNROLL for (int i = 0; i < 3; i++) p[i] = np[i];
}
// lane l: world frame of link l for joint positions q (this lane's q); same chain as fk() of sai2b_device.hpp
template <int G>
DI void fk_scan(const DevModel& md, int r, real q, real* R, real* p) {
	const bool act = r < N;
	const int rr = act ? r : 0;
	real s, c;
	sincos_joint(q, &s, &c);
	const bool pris = md.jtype[rr] != 0;  // prismatic: slides along the joint frame's z
	if (pris) s = 0, c = 1;
	UNROLL for (int k = 0; k < 3; k++) {
		const real e0 = md.E[rr][3 * k], e1 = md.E[rr][3 * k + 1], e2 = md.E[rr][3 * k + 2];
		R[3 * k + 0] = act ? fma(c, e0, s * e1) : kd(k, 0);
		R[3 * k + 1] = act ? fma(c, e1, -s * e0) : kd(k, 1);
		R[3 * k + 2] = act ? e2 : kd(k, 2);
		p[k] = sel0(act, fma(pris ? q : 0.0, e2, md.xyz[rr][k]));
	}
	fk_scan_step<G, 1>(R, p);
	fk_scan_step<G, 2>(R, p);
	fk_scan_step<G, 4>(R, p);
}
// pose of the compliant frame (MotionForceTask.cpp:286-289): computed in every lane for its own link, then taken
// from the lane of the task's link
template <int G>
DI void frame_pose_g(const DevTask& t, const real* FR, const real* Fp, real* x, real* R) {
	real xl[3], Rl[9];
	UNROLL for (int k = 0; k < 3; k++)
		xl[k] = fma(FR[3 * k], t.frame_pos[0], fma(FR[3 * k + 1], t.frame_pos[1], fma(FR[3 * k + 2], t.frame_pos[2], Fp[k])));
	mm<3, 3, 3>(FR, t.frame_rot, Rl);
	UNROLL for (int k = 0; k < 3; k++) x[k] = gather<G>(xl[k], t.link);
	UNROLL for (int k = 0; k < 9; k++) R[k] = gather<G>(Rl[k], t.link);
}

// ------------------------------------------------------------------ joint-space inertia (CRBA) and gravity
// Mrow: row r of M; g: this lane's component of the gravity vector (only when want_g)
template <int G>
DI void crba_g(const DevModel& md, int r, const real* R, const real* p, real* Mrow, bool want_g, real* g) {
	const bool act = r < N;
	const int rr = act ? r : 0;
	real c[3], comp[10];  // composite: mass, first moment (3), inertia about the world origin xx yy zz xy xz yz
	{
		UNROLL for (int a = 0; a < 3; a++)
			c[a] = fma(R[3 * a], md.com[rr][0], fma(R[3 * a + 1], md.com[rr][1], fma(R[3 * a + 2], md.com[rr][2], p[a])));
		const real* li = md.inertia[rr];
		const real l0 = li[0], l1 = li[1], l2 = li[2], l3 = li[3], l4 = li[4], l5 = li[5];
		real Il[9] = {l0, l3, l4, l3, l1, l5, l4, l5, l2}, T[9];
		mm<3, 3, 3>(R, Il, T);
		const real m = sel0(act, md.mass[rr]);
		const real c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
		const int ia[6] = {0, 1, 2, 0, 0, 1}, ib[6] = {0, 1, 2, 1, 2, 2};
		UNROLL for (int e = 0; e < 6; e++) {
			real s = 0;
			UNROLL for (int l = 0; l < 3; l++) s = fma(T[3 * ia[e] + l], R[3 * ib[e] + l], s);
			comp[4 + e] = act ? s + m * ((ia[e] == ib[e] ? c2 : 0.0) - c[ia[e]] * c[ib[e]]) : 0.0;
		}
		comp[0] = m;
		UNROLL for (int a = 0; a < 3; a++) comp[1 + a] = m * c[a];
	}
	// composite rigid bodies: suffix sums over the lanes (link k and everything outboard of it)
	UNROLL for (int i = 0; i < 10; i++) comp[i] += shift_down<G, 1>(comp[i], 0.0);
	UNROLL for (int i = 0; i < 10; i++) comp[i] += shift_down<G, 2>(comp[i], 0.0);
	UNROLL for (int i = 0; i < 10; i++) comp[i] += shift_down<G, 4>(comp[i], 0.0);
	const real mt = comp[0];
	const real* h = comp + 1;
	const real* IO = comp + 4;
	real S[6], W[6];  // joint twist (z, p x z) and the composite body's wrench under unit acceleration of this joint
	{
		const real ax[3] = {act ? R[2] : 0.0, act ? R[5] : 0.0, act ? R[8] : 0.0};
		const bool pris = md.jtype[rr] != 0;
		real pxz[3];
		cross3(p, ax, pxz);
		UNROLL for (int a = 0; a < 3; a++) {
			S[a] = pris ? 0.0 : ax[a];
			S[3 + a] = pris ? ax[a] : pxz[a];
		}
	}
	{
		real hv[3], hz[3];
		cross3(h, S + 3, hv);
		cross3(h, S, hz);
		W[0] = IO[0] * S[0] + IO[3] * S[1] + IO[4] * S[2] + hv[0];
		W[1] = IO[3] * S[0] + IO[1] * S[1] + IO[5] * S[2] + hv[1];
		W[2] = IO[4] * S[0] + IO[5] * S[1] + IO[2] * S[2] + hv[2];
		UNROLL for (int a = 0; a < 3; a++) W[3 + a] = mt * S[3 + a] - hz[a];
	}
	// M[k][j] = S_j . W_k for j <= k; the other triangle by symmetry = S_k . W_j
	real m1[N], m2[N];
	mm_rt<G, 6, N>(W, S, m1);
	mm_rt<G, 6, N>(S, W, m2);
	UNROLL for (int j = 0; j < N; j++) Mrow[j] = (j <= r) ? m1[j] : m2[j];
	if (want_g) {  // Sai2Model::jointGravityVector (RobotController.cpp:71)
		// generalised gravity force on the composite body: -(omega . (h x g) + v . (m g)) with the twist (omega, v)
		real hg[3];
		cross3(h, md.gravity, hg);
		real acc = 0;
		UNROLL for (int a = 0; a < 3; a++) acc = fma(S[a], hg[a], fma(S[3 + a], mt * md.gravity[a], acc));
		*g = -acc;
	}
}

// out[i] = the value `x` of lane i, i < n, in every lane
template <int G, int i, int n>
DI void bcast_all(real x, real* out) {
	if constexpr (i < n) {
		out[i] = bcast<G, i>(x);
		bcast_all<G, i + 1, n>(x, out);
	}
}

template <int G, int n, int j>
DI void certify_step(real* e, int r, real floor_, bool& ok) {
	if constexpr (j < n) {
		const real d = bcast<G, j>(e[j]);
		ok = ok && (d > floor_);
		const real nf = (r > j && r < n) ? -e[j] * recip(d) : 0.0;
		selffma<G, j, n>(e, nf);  // rows below the pivot: e[k] -= (e[j] / d) * pivot_row[k]
		certify_step<G, n, j + 1>(e, r, floor_, ok);
	}
}

// ------------------------------------------------------------------ certificate, projected pseudo-inverse
// certify_gram (sai2b_device.hpp) on a Gram matrix whose row r is in lane r < n: lambda_max <= ub := tr(G^4)^(1/4)
// <= n^(1/4) lambda_max and positive LDL^T pivots of G + ub Pc - rel2 ub I. pc: rows of the complement projector,
// or NULL. The verdict is the same in every lane of the group.
template <int G, int n>
DI bool certify_rows(const real* g, const real* pc, real abs2, real rel2) {
	const int r = lane<G>();
	real g2[n];
	mm_rr<G, n, n>(g, g, g2);  // G symmetric: G G = G^2
	real ss = 0;
	UNROLL for (int k = 0; k < n; k++) ss = fma(g2[k], g2[k], ss);
	const real t4 = allsum<G, n>(ss);
	const real ub = sqrt_nr(sqrt_nr(t4));
	bool ok = ub > 1.63 * abs2;	 // lambda_max >= ub / n^(1/4), 7^(1/4) = 1.6266
	const real c = rel2 * ub * (1.0 + 1e-9);
	const real floor_ = 1e-5 * c;
	real e[n];
	UNROLL for (int j = 0; j < n; j++) e[j] = (r < n) ? g[j] - ((j == r) ? c : 0.0) + (pc ? ub * pc[j] : 0.0) : 0.0;
	certify_step<G, n, 0>(e, r, floor_, ok);
	return ok;
}

// U_x (U_x^T A U_x)^-1 U_x^T for the orthogonal projector Pi (rows in lanes), A symmetric: pinv_proj of sai2b_device.hpp
// kmax < n: Pi = diag(1 x kmax, 0 ...) for every robot of the launch (see spd_inverse_rows)
template <int G, int n>
DI void pinv_proj_rows(const real* A, const real* Pi, real* out, int kmax = n) {
	const int r = lane<G>();
	real T[n];
	mm_rr<G, n, n>(Pi, A, T);
	mm_rr<G, n, n>(T, Pi, out);
	UNROLL for (int j = 0; j < n; j++) out[j] += (r < n) ? kd(r, j) - Pi[j] : 0.0;
	spd_inverse_rows<G, n>(out, kmax);
	UNROLL for (int j = 0; j < n; j++) out[j] -= (r < n) ? kd(r, j) - Pi[j] : 0.0;
}
// (Jp A Jp^T): rows of Jp (m x N) in lanes, A symmetric N x N rows in lanes; T1 = Jp A is returned too
template <int G, int m>
DI void sandwich_rows(const real* jp, const real* A, real* t1, real* out) {
	mm_rr<G, N, N>(jp, A, t1);
	mm_rt<G, N, m>(t1, jp, out);
}

// ------------------------------------------------------------------ one-sided (Hestenes) Jacobi, columns across registers
// X (N x C): row r in lane r; W (C x C, rotations accumulated, starts as I): row i in lane i < C. Same cyclic order,
// threshold and rotation formula as hestenes() of sai2b_device.hpp / the oracle; the column norms are carried
// (refreshed every sweep) instead of recomputed for every pair.
// ncols (the same for every robot of the launch): columns >= ncols of X are exactly zero and are left alone.
template <int G, int C>
DI void jacobi_rows(real* x, real* w, int ncols = C) {
	const int r = lane<G>();
	UNROLL for (int j = 0; j < C; j++) w[j] = (r < C) ? kd(r, j) : 0.0;
#pragma unroll 1
	for (int sweep = 0; sweep < 60; sweep++) {
		real d[C];
		UNROLL for (int j = 0; j < C; j++) {
			d[j] = 0;
			if (j < ncols) d[j] = allsum<G, N>(x[j] * x[j]);
		}
		bool rotated = false;
		UNROLL for (int i = 0; i < C - 1; i++) UNROLL for (int j = i + 1; j < C; j++) {
			if (j >= ncols) continue;  // uniform over the wavefront: a scalar branch around the pair
			const real ga = allsum<G, N>(x[i] * x[j]);
			const real al = d[i], be = d[j];
			// |ga| > 1e-15 sqrt(al be), without the square root (the same for every lane of the robot's group)
			const bool rot = ga * ga > 1e-30 * (al * be);
			if (rot) {	// no cross-lane operation inside: robots of the wavefront that are done with this pair idle
				rotated = true;
				const real zeta = (be - al) * (0.5 * recip(ga));
				const real t = copysign(1.0, zeta) * recip(fabs(zeta) + sqrt_nr(fma(zeta, zeta, 1.0)));
				const real c = rsqrt_nr(fma(t, t, 1.0)), s = c * t;
				const real xi = x[i], xj = x[j], wi = w[i], wj = w[j];
				x[i] = c * xi - s * xj;
				x[j] = s * xi + c * xj;
				w[i] = c * wi - s * wj;
				w[j] = s * wi + c * wj;
				d[i] = fma(-t, ga, al);
				d[j] = fma(t, ga, be);
			}
		}
		if (!rotated) break;
	}
}

// Row space accumulated over the certified tasks (Chain of sai2b_device.hpp), rows in lanes
struct ChainG {
	bool ok;
	int wrows;
	real W[N];
};
template <int G>
DI void chain_append_g(ChainG& ch, const real* rows, int nrows) {
	const int r = lane<G>();
	const bool hit = r >= ch.wrows && r < ch.wrows + nrows;
	const int src = hit ? r - ch.wrows : 0;
	UNROLL for (int j = 0; j < N; j++) {
		const real v = gather<G>(rows[j], src);
		ch.W[j] = hit ? v : ch.W[j];
	}
	ch.wrows += nrows;
}

// rows of an n-row matrix (row j in lane j) picked by src_of(slot) into the lanes after the chain's rows: lane
// wrows + m receives the row of lane `src` where `src` was chosen by the caller for slot m = r - wrows
template <int G>
DI void chain_append_from(ChainG& ch, const real* rows, int src, int count) {
	const int r = lane<G>();
	const bool hit = r >= ch.wrows && r < ch.wrows + count;
	UNROLL for (int j = 0; j < N; j++) {
		const real v = gather<G>(rows[j], hit ? src : 0);
		ch.W[j] = hit ? v : ch.W[j];
	}
	ch.wrows += count;
	if (ch.wrows > N) ch.ok = false;
}
// one replicated row (the same values in every lane) appended to the chain
template <int G>
DI void chain_append_row(ChainG& ch, const real* row) {
	const int r = lane<G>();
	UNROLL for (int j = 0; j < N; j++) ch.W[j] = (r == ch.wrows) ? row[j] : ch.W[j];
	ch.wrows += 1;
	if (ch.wrows > N) ch.ok = false;
}

// N_prec^T <- N_prec^T N^T  (RobotController.cpp:58, MotionForceTask.h:207-209)
template <int G>
DI void nprec_update(bool first, const real* ntaskT, real* nprecT) {
	if (first) {
		UNROLL for (int j = 0; j < N; j++) nprecT[j] = ntaskT[j];
	} else {
		real T[N];
		mm_rr<G, N, N>(nprecT, ntaskT, T);
		UNROLL for (int j = 0; j < N; j++) nprecT[j] = T[j];
	}
}
// N^T = I - Jp^T (L T1) for N = I - M^-1 Jp^T L Jp, T1 = Jp M^-1 (rows in lanes), L symmetric m x m
template <int G, int m>
DI void nullspace_T(const real* jpT, const real* L, const real* t1, int r, real* nT) {
	real w2[N], acc[N];
	mm_rr<G, m, N>(L, t1, w2);
	mm_rr<G, m, N>(jpT, w2, acc);
	UNROLL for (int j = 0; j < N; j++) nT[j] = (r < N ? kd(r, j) : 0.0) - acc[j];
}

// ------------------------------------------------------------------ MotionForceTask (MotionForceTask.cpp:247-509,
// SingularityHandler.cpp:75-368): mft_task of sai2b_device.hpp, rows in lanes. tau: this lane's component.
template <int G>
SAI2B_TASK_FN void mft_task_g(const DevParams& P, const DevTask& t, const Rob& rb, bool first, bool last, bool commit_sh, bool do_torque,
				   real* nprecT, real& tau_total, ChainG& chain, real* Ntask_out = nullptr) {
	const int r = rb.r, B = rb.B, b = rb.b;
	const bool r6 = r < 6, rN = r < N;
	real x[3], R[9];
	GMARK(0, "mft_begin");
	frame_pose_g<G>(t, rb.FR, rb.Fp, x, R);
	// Jacobian column of this lane's joint (JWorldFrame, linear rows first), projected: J = P Jw
	real jT[6];
	{
		const real z[3] = {rb.FR[2], rb.FR[5], rb.FR[8]};
		const real d[3] = {x[0] - rb.Fp[0], x[1] - rb.Fp[1], x[2] - rb.Fp[2]};
		real v[3];
		cross3(z, d, v);
		const bool on = rN && r <= t.link, pris = P.model.jtype[rN ? r : 0] != 0;
		real jw[6];
		UNROLL for (int k = 0; k < 3; k++) {
			jw[k] = on ? (pris ? z[k] : v[k]) : 0.0;
			jw[3 + k] = (on && !pris) ? z[k] : 0.0;
		}
		if (t.full_projection) {
			UNROLL for (int i = 0; i < 6; i++) jT[i] = jw[i];
		} else {
			sai2b::mv<6, 6>(t.P, jw, jT);
		}
	}
	real jpT[6], jp[N];
	if (first) {
		UNROLL for (int i = 0; i < 6; i++) jpT[i] = jT[i];
	} else {
		mm_rr<G, N, 6>(nprecT, jT, jpT);
	}
	transpose_lds<G, N, 6>(rb.pad, jpT, jp);
	GMARK(1, "mft_jp_done");
	// ---- branch decision (SingularityHandler.cpp:83-143): SVD-free certificate first, per robot
	const int rank = t.rank;
	real g6[6], pns[6], ps[6], Prow[6];
	mm_rt<G, N, 6>(jp, jp, g6);
	const int r5 = r6 ? r : 0;
	UNROLL for (int k = 0; k < 6; k++) Prow[k] = sel0(r6, t.P[r5 * 6 + k]);
	bool certified;
	{
		real pc[6];
		UNROLL for (int k = 0; k < 6; k++) pc[k] = r6 ? kd(r, k) - Prow[k] : 0.0;
		certified = certify_rows<G, 6>(g6, t.full_projection ? nullptr : pc, t.s_abs_tol * t.s_abs_tol, t.s_max * t.s_max);
	}
	GMARK(2, "mft_cert_done");
	real x6[6], w6[6], sv[6], alpha = 1;
	int pos[6], split = rank;
	UNROLL for (int i = 0; i < 6; i++) {
		x6[i] = w6[i] = sv[i] = 0;
		pos[i] = i;
		pns[i] = Prow[i];
		ps[i] = 0;
	}
	if (certified) {
		if (chain.ok) {	 // row space of this task for a later full JointTask: rows PU^T Jp (rank of them)
			real rows[N];
			if (t.full_projection) {
				UNROLL for (int j = 0; j < N; j++) rows[j] = jp[j];
			} else {
				real pu[6];
				UNROLL for (int k = 0; k < 6; k++) pu[k] = sel0(r6, t.PU[k * 6 + r5]);
				mm_rr<G, 6, N>(pu, jp, rows);
			}
			chain_append_g<G>(chain, rows, rank);
		}
	} else {
		// ---- thin SVD of Jp via one-sided Jacobi on Jp^T (N x 6): Jp^T W = Q, U = W, V = Q / s. A partial task is
		// rotated into the basis PU of range(P) first: Jp^T PU has exactly `rank` non-zero columns, the pairs of the
		// others are skipped, and U = PU W'
		if (t.full_projection) {
			UNROLL for (int i = 0; i < 6; i++) x6[i] = jpT[i];
			jacobi_rows<G, 6>(x6, w6);
		} else {
			mv_t<6, 6>(t.PU, jpT, x6);	// x6[c] = sum_k jpT[k] PU[k][c]
			real wp[6], pu[6];
			jacobi_rows<G, 6>(x6, wp, rank);
			UNROLL for (int k = 0; k < 6; k++) pu[k] = sel0(r6, t.PU[r5 * 6 + k]);	 // row r of PU
			mm_rr<G, 6, 6>(pu, wp, w6);
		}
		UNROLL for (int j = 0; j < 6; j++) {
			sv[j] = 0;
			if (j < rank) sv[j] = sqrt_nr(allsum<G, N>(x6[j] * x6[j]));
		}
		real ss[6];
		UNROLL for (int j = 0; j < 6; j++) {
			int p = 0;
			UNROLL for (int k = 0; k < 6; k++) p += (sv[k] > sv[j] || (sv[k] == sv[j] && k < j)) ? 1 : 0;
			pos[j] = p;
		}
		UNROLL for (int p = 0; p < 6; p++) {
			real s = 0;
			UNROLL for (int j = 0; j < 6; j++) s = (pos[j] == p) ? sv[j] : s;
			ss[p] = s;
		}
		if (ss[0] < t.s_abs_tol) {
			split = 0;
			alpha = 0;
		} else {
			split = rank;
			alpha = 1;
			bool found = false;
			const real inv0 = recip(ss[0]);
			UNROLL for (int i = 1; i < 6; i++) {
				const real icn = ss[i] * inv0;
				if (i < rank && !found && icn < t.s_max) {
					alpha = fmin(fmax((icn - t.s_min) / (t.s_max - t.s_min), 0.0), 1.0);
					split = i;
					found = true;
				}
			}
		}
		if (split != rank) {  // fully non-singular: U_ns spans range(P) and the projector is P itself
			real wa[6], wb[6];
			UNROLL for (int j = 0; j < 6; j++) {
				wa[j] = (pos[j] < split) ? w6[j] : 0.0;
				wb[j] = (pos[j] >= split && pos[j] < rank) ? w6[j] : 0.0;
			}
			mm_rt<G, 6, 6>(wa, w6, pns);
			mm_rt<G, 6, 6>(wb, w6, ps);
		}
		if (chain.ok && split > 0) {
			// row space of the non-singular part for a later full JointTask: U_ns^T Jp, the columns ranked < split
			real uT[6], rows6[N];
			transpose_lds<G, 6, 6>(rb.pad, w6, uT);
			mm_rr<G, 6, N>(uT, jp, rows6);	// lane j: (U^T Jp)[j][:]
			const int slot = r - chain.wrows;
			int src = 0;
			UNROLL for (int j = 0; j < 6; j++) src = (pos[j] == slot) ? j : src;
			chain_append_from<G>(chain, rows6, src, split);
		}
	}
	GMARK(3, "mft_branch_done");
	const int sc = rank - split;
	const bool bie = t.decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES;
	const bool impedance = t.decoupling == SAI2B_IMPEDANCE;
	// ---- non-singular part: Lambda_ns (embedded), N_ns (SingularityHandler.cpp:110-114,130-134)
	real A[6], AB[6], t1[N], lns[6], lnsMod[6], nnsT[N];
	const int kmax6 = (split == rank) ? t.p_lead : 6;  // pns == P == diag(1 x p_lead, 0 ...)
	sandwich_rows<G, 6>(jp, rb.minv, t1, A);
	pinv_proj_rows<G, 6>(A, pns, lns, kmax6);
	if (bie) {
		real t1b[N];
		sandwich_rows<G, 6>(jp, rb.minvB, t1b, AB);
		pinv_proj_rows<G, 6>(AB, pns, lnsMod, kmax6);
	} else {
		UNROLL for (int i = 0; i < 6; i++) {
			AB[i] = A[i];
			lnsMod[i] = impedance ? pns[i] : lns[i];
		}
	}
	nullspace_T<G, 6>(jpT, lns, t1, r, nnsT);
	GMARK(4, "mft_lambda_done");
	// ---- control law (replicated in every lane)
	real Fu[6], Ff[6];
	{
		real vw[6];
		UNROLL for (int i = 0; i < 6; i++) vw[i] = allsum<G, N>(jT[i] * rb.dq);
		MftIn in;
		mft_load(t, B, b, in);
		mft_law_vw(t, vw, vw + 3, x, R, in, Fu, Ff, B, b, do_torque);
		if (do_torque) mft_store_integrators(t, B, b, in);
	}
	real tau;
	{
		real a6 = 0;
		UNROLL for (int k = 0; k < 6; k++) a6 = fma(lnsMod[k], Fu[k], fma(pns[k], Ff[k], a6));
		tau = mv<G, 6>(jpT, a6);  // SingularityHandler.cpp:307-309
	}
	GMARK(5, "mft_law_done");
	// ---- singularity bookkeeping (SingularityHandler.cpp:230-295) and blended torques (:313-367)
	int* IS = t.istate;
	real* S = t.state;
	const int prev_types = ldi(IS, IS_NTYPES, B, b);
	real ntaskT[N];
	UNROLL for (int j = 0; j < N; j++) ntaskT[j] = nnsT[j];
	const int rs = rN ? r : 0;	// state rows of this lane's joint
	if (sc == 0) {
		if (prev_types != 0 && commit_sh) {	 // leaving the singular region: clear history (:239-245)
			sti(IS, IS_NTYPES, B, b, 0);
			sti(IS, IS_COUNT, B, b, 0);
			sti(IS, IS_SIZE, B, b, 0);
			sti(IS, IS_C1, B, b, 0);
			sti(IS, IS_C2, B, b, 0);
		}
	} else {
		int c1 = ldi(IS, IS_C1, B, b), c2 = ldi(IS, IS_C2, B, b);
		real qprior;  // this lane's component
		if (commit_sh && (prev_types == 0 || c2 > c1)) {
			if (rN) {
				st(S, MFT_QPRIOR + rs, B, b, rb.q);
				st(S, MFT_DQPRIOR + rs, B, b, rb.dq);
			}
			qprior = rb.q;
		} else {
			qprior = ld(S, MFT_QPRIOR + rs, B, b);
		}
		real us0[6], vs0 = 0, pv[N];
		UNROLL for (int j = 0; j < N; j++) pv[j] = 0;
		UNROLL for (int i = 0; i < 6; i++) us0[i] = 0;
		bool any1 = false;
#pragma unroll 1
		for (int p = split; p < rank; p++) {
			real wsel = 0, v = 0, s = 0;
			UNROLL for (int j = 0; j < 6; j++) {
				const bool hit = pos[j] == p;
				s = hit ? sv[j] : s;
				wsel = hit ? w6[j] : wsel;
				v = hit ? x6[j] : v;
			}
			real u[6];
			bcast_all<G, 0, 6>(wsel, u);
			real inv = s > 0 ? 1.0 / s : 0.0;
			{  // sign convention shared with the oracle: largest-magnitude component of v positive
				real big = 0, bigabs = -1;
				real vi[N];
				bcast_all<G, 0, N>(v, vi);
				UNROLL for (int i = 0; i < N; i++) {
					const bool take = fabs(vi[i]) > bigabs;
					bigabs = take ? fabs(vi[i]) : bigabs;
					big = take ? vi[i] : big;
				}
				if (big < 0) {
					inv = -inv;
					UNROLL for (int i = 0; i < 6; i++) u[i] = -u[i];
				}
			}
			v = rN ? v * inv : 0.0;
			if (p == split) {
				UNROLL for (int i = 0; i < 6; i++) us0[i] = u[i];
				vs0 = v;
			}
			lanefma<G, N>(pv, v, v);  // PV += v v^T
			// classification by FK perturbation (:253-273) along +v, -v or both (enum sai2b_singular_vector_sign; the
			// setting is batch-uniform, so the passes are uniform over the wavefront)
			bool moved0 = false, moved1 = false;
			const int pass0 = t.sv_sign == SAI2B_SV_SIGN_V_MAX_NEGATIVE ? 1 : 0;
			const int pass1 = t.sv_sign == SAI2B_SV_SIGN_V_MAX_POSITIVE ? 0 : 1;
#pragma unroll 1
			for (int pass = pass0; pass <= pass1; pass++) {
				real R1f[9], p1f[3], x1[3], R1[9], dd[6];
				fk_scan<G>(P.model, r, fma(pass ? -t.perturb : t.perturb, v, rb.q), R1f, p1f);
				frame_pose_g<G>(t, R1f, p1f, x1, R1);
				UNROLL for (int k = 0; k < 3; k++) dd[k] = x1[k] - x[k];
				orientation_error(R1, R, dd + 3);
				real m = 0;
				UNROLL for (int k = 0; k < 6; k++) m = fma(dd[k], u[k], m);
				const bool mv = fabs(m) > t.type_1_tol;
				moved0 = pass ? moved0 : mv;
				moved1 = pass ? mv : moved1;
			}
			any1 = any1 || (t.sv_sign == SAI2B_SV_SIGN_BOTH ? (moved0 && moved1) : (moved0 || moved1));
		}
		if (commit_sh) {  // history ring (:276-293); every lane of the group computes and stores the same values
			int count = ldi(IS, IS_COUNT, B, b), size = ldi(IS, IS_SIZE, B, b);
			const int cap = t.sh_cap;
			const int idx = count % cap;
			int word = ldi(IS, idx >> 5, B, b);
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
			sti(IS, idx >> 5, B, b, word);
			sti(IS, IS_COUNT, B, b, (count + 1) % (cap * 32768));
			sti(IS, IS_SIZE, B, b, size);
			sti(IS, IS_C1, B, b, c1);
			sti(IS, IS_C2, B, b, c2);
			sti(IS, IS_NTYPES, B, b, sc);
		}
		if (split == 0) {
			// fully singular: pass the task through (:149-150, :317-318)
			tau = 0;
			UNROLL for (int j = 0; j < N; j++) ntaskT[j] = rN ? kd(r, j) : 0.0;
		}
		if (split != 0 && t.enforce) {
			// posture task in the singular joint directions (:152-157): Bm = N_ns N_prec
			real bmT[N], bm[N];
			if (first) {
				UNROLL for (int j = 0; j < N; j++) bmT[j] = nnsT[j];
			} else {
				mm_rr<G, N, N>(nprecT, nnsT, bmT);
			}
			transpose_lds<G, N, N>(rb.pad, bmT, bm);
			if (chain.ok) {
				// ... and of the posture task: the rows V_s^T Bm, one per singular column
#pragma unroll 1
				for (int p = split; p < rank; p++) {
					real v = 0, s_ = 0, row[N];
					UNROLL for (int j = 0; j < 6; j++) {
						const bool hit = pos[j] == p;
						s_ = hit ? sv[j] : s_;
						v = hit ? x6[j] : v;
					}
					v = (rN && s_ > 0) ? v / s_ : 0.0;
					UNROLL for (int c = 0; c < N; c++) row[c] = allsum<G, N>(v * bm[c]);
					chain_append_row<G>(chain, row);
				}
			}
			real C[N], tb[N], lj[N], ljMod[N];
			sandwich_rows<G, N>(bm, rb.minv, tb, C);
			pinv_proj_rows<G, N>(C, pv, lj);
			{
				real npT[N], T[N];
				nullspace_T<G, N>(bmT, lj, tb, r, npT);
				mm_rr<G, N, N>(nnsT, npT, T);  // N = N_posture N_ns  ->  N^T = N_ns^T N_posture^T
				UNROLL for (int j = 0; j < N; j++) ntaskT[j] = T[j];
			}
			if (!impedance) {
				if (bie) {
					real tbb[N];
					sandwich_rows<G, N>(bm, rb.minvB, tbb, C);
					pinv_proj_rows<G, N>(C, pv, ljMod);
				} else {
					UNROLL for (int j = 0; j < N; j++) ljMod[j] = lj[j];
				}
				real lsMod[6];
				pinv_proj_rows<G, 6>(bie ? AB : A, ps, lsMod);
				// joint strategy (:327-351)
				real tau_j;
				if (c1 > c2 || t.enforce_t1) {
					const real ut = rN ? -t.kp1 * (rb.q - qprior) - t.kv1 * rb.dq : 0.0;
					const real y7 = mv<G, N>(ljMod, ut);
					tau_j = mv<G, N>(bmT, y7);
				} else {
					real dir = ld(S, MFT_T2DIR + rs, B, b);
					if (rN && vs0 != 0) {
						if (fabs(rb.q - P.model.q_upper[rs]) < t.t2_angle) {
							dir = -1;
							if (do_torque) st(S, MFT_T2DIR + rs, B, b, dir);
						} else if (fabs(rb.q - P.model.q_lower[rs]) < t.t2_angle) {
							dir = 1;
							if (do_torque) st(S, MFT_T2DIR + rs, B, b, dir);
						}
					}
					real Fs[6], nrm = 0, fTd = 0;
					UNROLL for (int i = 0; i < 6; i++) {
						Fs[i] = Fu[i] + Ff[i];
						nrm = fma(Fs[i], Fs[i], nrm);
					}
					nrm = sqrt(nrm);
					UNROLL for (int i = 0; i < 6; i++) fTd = fma(nrm > 0 ? Fs[i] / nrm : Fs[i], us0[i], fTd);
					const real ut = rN ? dir * fabs(fTd) * t.t2_ratio * P.model.effort[rs] : 0.0;
					const real a7 = mv<G, N>(pv, ut);
					const real ut2 = rN ? -t.kv2 * rb.dq : 0.0;
					const real b7 = mv<G, N>(ljMod, ut2);
					tau_j = mv<G, N>(bmT, a7 + b7);
				}
				// singular-direction torques, sanitised and clamped (:354-365)
				real a6 = 0;
				UNROLL for (int k = 0; k < 6; k++) a6 = fma(lsMod[k], Fu[k], fma(ps[k], Ff[k], a6));
				real v = mv<G, 6>(jpT, a6);
				const real eff = P.model.effort[rs];
				v = (v != v) ? 0.0 : fmin(fmax(v, -eff), eff);
				tau = tau + alpha * v + (1 - alpha) * tau_j;  // :366
			}
		}
	}
	tau_total += rN ? tau : 0.0;
	GMARK(6, "mft_sing_done");
	if (Ntask_out && rN) {	// getTaskNullspace (TemplateTask.h:73-77), task-level calls only: lane r holds column r
		UNROLL for (int j = 0; j < N; j++) st(Ntask_out, j * N + r, B, b, ntaskT[j]);
	}
	if (!last) nprec_update<G>(first, ntaskT, nprecT);
	GMARK(7, "mft_end");
}

// ------------------------------------------------------------------ JointTask (JointTask.cpp:218-356): jt_task of
// sai2b_device.hpp, rows in lanes
template <int G, bool RANGE_ONLY = false>
SAI2B_TASK_FN void jt_task_g(const DevParams& P, const DevTask& t, const Rob& rb, bool first, bool last, bool with_comp, bool do_torque,
				  real* nprecT, real& tau_total, ChainG& chain, real* Ntask_out = nullptr) {
	const int r = rb.r, B = rb.B, b = rb.b;
	const bool rN = r < N, rk = r < t.k0;
	const int rs = rk ? r : 0;
	GMARK(8, "jt_begin");
	// Jp = S N_prec: its transpose column by column is local (S is batch-uniform); rows through the pad
	real jpT[N], jp[N], srow[N];
	UNROLL for (int l = 0; l < N; l++) srow[l] = sel0(rk, t.S[rs * N + l]);  // row r of S
	if (first) {
		const int rr = rN ? r : 0;
		UNROLL for (int i = 0; i < N; i++) jpT[i] = sel0(rN, t.S[i * N + rr]);
		UNROLL for (int l = 0; l < N; l++) jp[l] = srow[l];
	} else {
		if (t.full_selection) {
			UNROLL for (int i = 0; i < N; i++) jpT[i] = nprecT[i];
		} else {
			sai2b::mv<N, N>(t.S, nprecT, jpT);	 // (S N_prec)[i][r] = sum_l S[i][l] N_prec[l][r]
		}
		transpose_lds<G, N, N>(rb.pad, jpT, jp);
	}
	GMARK(9, "jt_jp_done");
	// range projector of Jp (Sai2Model::matrixRangeBasis, tolerance 1e-3: SURVEY App. D), rows in lanes
	real PR[N];
	bool zero_range = false, need_svd = false;
	int pr_lead = N;  // PR == diag(1 x pr_lead, 0 ...): the 7 x 7 eliminations stop there
	if (first) {
		UNROLL for (int j = 0; j < N; j++) PR[j] = rk ? kd(r, j) : 0.0;
		pr_lead = t.k0;
		if (chain.ok) chain_append_g<G>(chain, jp, t.k0);
	} else if (t.full_selection && chain.ok) {
		// Jp = N_prec behind certified tasks: range(N_prec) = null(W)
		zero_range = chain.wrows >= N;
		real wwt[N], X[N], wT[N], acc[N];
		mm_rt<G, N, N>(chain.W, chain.W, wwt);
		UNROLL for (int j = 0; j < N; j++) wwt[j] += (rN && r >= chain.wrows) ? kd(r, j) : 0.0;
		spd_inverse_rows<G, N>(wwt);
		mm_rr<G, N, N>(wwt, chain.W, X);
		transpose_lds<G, N, N>(rb.pad, chain.W, wT);
		mm_rr<G, N, N>(wT, X, acc);
		UNROLL for (int j = 0; j < N; j++) PR[j] = (rN ? kd(r, j) : 0.0) - acc[j];
		chain.wrows = N;  // a full JointTask closes the hierarchy
	} else if (!t.full_selection) {
		// partial task: certified full row rank (all k0 singular values above the 1e-3 rule) -> R = I_k0
		real c0[N], pc[N];
		mm_rt<G, N, N>(jp, jp, c0);
		UNROLL for (int j = 0; j < N; j++) pc[j] = (rN && !rk) ? kd(r, j) : 0.0;
		if (certify_rows<G, N>(c0, pc, 1e-6, 1e-6)) {
			UNROLL for (int j = 0; j < N; j++) PR[j] = rk ? kd(r, j) : 0.0;
			pr_lead = t.k0;
			if (chain.ok) chain_append_g<G>(chain, jp, t.k0);
		} else {
			need_svd = true;
		}
	} else {
		need_svd = true;
	}
	if (need_svd) {
		// left singular vectors of Jp = the rotations of the one-sided Jacobi on Jp^T (N x N, columns >= k0 zero)
		real X[N], W[N], sv[N];
		UNROLL for (int j = 0; j < N; j++) X[j] = jpT[j];
		jacobi_rows<G, N>(X, W, t.k0);
		real s0 = 0;
		UNROLL for (int j = 0; j < N; j++) {
			sv[j] = 0;
			if (j < t.k0) sv[j] = sqrt_nr(allsum<G, N>(X[j] * X[j]));
			s0 = fmax(s0, sv[j]);
		}
		zero_range = s0 < 1e-3;
		int dof = 0;
		real wa[N];
		bool keep[N];
		UNROLL for (int j = 0; j < N; j++) {
			keep[j] = !zero_range && sv[j] / s0 >= 1e-3;
			dof += keep[j] ? 1 : 0;
			wa[j] = keep[j] ? W[j] : 0.0;
		}
		mm_rt<G, N, N>(wa, W, PR);
		if (dof == t.k0) {	// full row rank -> identity basis
			UNROLL for (int j = 0; j < N; j++) PR[j] = rk ? kd(r, j) : 0.0;
			pr_lead = t.k0;
		}
		if (chain.ok && dof > 0) {	// row space of the task: R^T Jp, the kept columns in order
			real uT[N], rowsN[N];
			transpose_lds<G, N, N>(rb.pad, W, uT);
			mm_rr<G, N, N>(uT, jp, rowsN);	// lane j: (W^T Jp)[j][:]
			const int slot = r - chain.wrows;
			int src = 0, cnt = 0;
			UNROLL for (int j = 0; j < N; j++) {
				src = (keep[j] && cnt == slot) ? j : src;
				cnt += keep[j] ? 1 : 0;
			}
			chain_append_from<G>(chain, rowsN, src, dof);
		}
	}
	GMARK(10, "jt_range_done");
	if (t.otg_gated && !do_torque) st(t.otg_state, OTG_ACTIVE, B, b, zero_range ? 0.0 : 1.0);	 // read by otg_kernel
	if constexpr (RANGE_ONLY) return;
	real tau = 0, ntaskT[N];
	UNROLL for (int j = 0; j < N; j++) ntaskT[j] = rN ? kd(r, j) : 0.0;
	if (!zero_range) {
		// controller state and PD(+I) law of this lane's task coordinate (JointTask.cpp:299-345)
		real cur, vel;
		if (t.full_selection) {
			cur = rb.q, vel = rb.dq;
		} else {
			cur = mv<G, N>(srow, rb.q);
			vel = mv<G, N>(srow, rb.dq);
		}
		real f = 0, ddq_d = 0;
		if (rk) {
			const real* Gl = t.law_goals;
			const real qd = ld(Gl, rs, B, b), dqd = ld(Gl, t.k0 + rs, B, b);
			ddq_d = ld(Gl, 2 * t.k0 + rs, B, b);
			const real integ = fma(cur - qd, t.dt, ld(t.state, rs, B, b));
			if (do_torque) st(t.state, rs, B, b, integ);
			const real kp = t.kp[rs], kv = t.kv[rs], ki = t.ki[rs];
			if (t.use_vsat) {
				const real kvi = gain_pinv(kv);
				real dv = -kp * kvi * (cur - qd) - ki * kvi * integ;
				const real vs = t.vsat[rs];
				dv = fmin(fmax(dv, -vs), vs);
				f = -kv * (vel - dv);
			} else {
				f = -kp * (cur - qd) - kv * (vel - dqd) - ki * integ;
			}
		}
		real C[N], t1[N], L[N], LMod[N];
		sandwich_rows<G, N>(jp, rb.minv, t1, C);
		pinv_proj_rows<G, N>(C, PR, L, pr_lead);
		if (t.decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
			real t1b[N];
			sandwich_rows<G, N>(jp, rb.minvB, t1b, C);
			pinv_proj_rows<G, N>(C, PR, LMod, pr_lead);
		} else {
			UNROLL for (int j = 0; j < N; j++) LMod[j] = (t.decoupling == SAI2B_IMPEDANCE) ? PR[j] : L[j];
		}
		// x = M_partial R^T ddq_d + M_partial_mod R^T f (JointTask.cpp:348-351), embedded
		real xa = mv<G, N>(L, ddq_d) + mv<G, N>(LMod, f);
		if (with_comp && !first) {	// JointTask.cpp:285-292
			const real a7 = mv<G, N>(rb.minv, tau_total);
			const real b7 = t.full_selection ? a7 : mv<G, N>(srow, a7);
			xa -= mv<G, N>(L, b7);
		}
		tau = mv<G, N>(jpT, xa);
		GMARK(11, "jt_torque_done");
		if (!last) nullspace_T<G, N>(jpT, L, t1, r, ntaskT);
	}
	tau_total += rN ? tau : 0.0;
	if (Ntask_out && rN) {
		UNROLL for (int j = 0; j < N; j++) st(Ntask_out, j * N + r, B, b, ntaskT[j]);
	}
	if (!last) nprec_update<G>(first, ntaskT, nprecT);
	GMARK(12, "jt_end");
}

// ------------------------------------------------------------------ one robot's tick
// flags as tick_kernel (sai2b_kernels.hip). RANGE: only the range decisions of gated JointTasks (DevTask::otg_gated).
template <int G, bool RANGE>
DI void tick_robot(const DevParams& P, int b, real* pad, int commit_sh, int with_comp, int do_torque) {
	const int B = P.B;
	Rob rb;
	rb.r = lane<G>();
	rb.B = B;
	rb.b = b;
	rb.pad = pad;
	const bool rN = rb.r < N;
	const int rs = rN ? rb.r : 0;
	rb.q = sel0(rN, ld(P.q, rs, B, b));
	rb.dq = sel0(rN, ld(P.dq, rs, B, b));
	real g = 0;
	GMARK(13, "model_begin");
	{
		// Sai2Model::updateModel(): kinematics, M (CRBA), M^-1 (examples/05-using_robot_controller.cpp:143-145)
		fk_scan<G>(P.model, rb.r, rb.q, rb.FR, rb.Fp);
		GMARK(14, "model_fk_done");
		real Mrow[N];
		crba_g<G>(P.model, rb.r, rb.FR, rb.Fp, Mrow, P.gravity_comp != 0, &g);
		GMARK(15, "model_crba_done");
		UNROLL for (int j = 0; j < N; j++) rb.minv[j] = Mrow[j];
		spd_inverse_rows<G, N>(rb.minv);
		GMARK(16, "model_minv_done");
		// bounded inertia estimate (SingularityHandler.cpp:176-182, JointTask.cpp:254-260), shared
		const bool any_bie = P.any_bie != 0;  // (host: upload_params)
		const real thr = P.bie_thr;
		if (any_bie) {
			UNROLL for (int j = 0; j < N; j++) rb.minvB[j] = (j == rb.r) ? fmax(Mrow[j], thr) : Mrow[j];
			spd_inverse_rows<G, N>(rb.minvB);
		} else {
			UNROLL for (int j = 0; j < N; j++) rb.minvB[j] = rb.minv[j];
		}
	}
	GMARK(17, "model_end");
	real nprecT[N], tau = 0;
	UNROLL for (int j = 0; j < N; j++) nprecT[j] = rN ? kd(rb.r, j) : 0.0;
	ChainG chain;
	chain.ok = true;
	chain.wrows = 0;
	UNROLL for (int j = 0; j < N; j++) chain.W[j] = 0;
	int n_run = P.n_tasks;
	if constexpr (RANGE) {
		n_run = 0;
		for (int t = 0; t < P.n_tasks; t++)
			if (P.task[t].otg_gated) n_run = t + 1;
	}
#pragma unroll 1
	for (int t = 0; t < n_run; t++) {
		const DevTask& tk = P.task[t];
		const bool first = (t == 0), last = (t == P.n_tasks - 1);
		if (tk.type == SAI2B_MOTION_FORCE_TASK)
			mft_task_g<G>(P, tk, rb, first, last, commit_sh != 0, do_torque != 0, nprecT, tau, chain);
		else if (RANGE && t == n_run - 1)
			jt_task_g<G, true>(P, tk, rb, first, last, with_comp != 0, do_torque != 0, nprecT, tau, chain);
		else
			jt_task_g<G>(P, tk, rb, first, last, with_comp != 0, do_torque != 0, nprecT, tau, chain);
	}
	if (do_torque && rN) st(P.tau, rs, B, b, tau + g);	// RobotController.cpp:70-72
}


// The TemplateTask calls on ONE task under a caller-supplied N_prec (TemplateTask.h:42-88), a robot spread over G lanes: the
// lanes-per-robot form of task_kernel (sai2b_kernels.hip), which it replaces behind task_cert_kernel's work list and for
// the hierarchies that kernel does not take (round 3: the last user-reachable kernel with kilobytes of scratch per
// lane is gone). Nothing is known about the tasks above: the range decisions take the Jacobi SVD (chain.ok = false), and a
// JointTask applies the compensation term also behind an identity N_prec (first = false), as task_kernel does.
template <int G>
DI void task_robot(const DevParams& P, int task, int b, real* pad, const real* Nprec_in, const real* tau_prec, real* tau_out, real* N_out,
				   real* Ntot_out, int commit_sh, int do_torque) {
	const int B = P.B;
	Rob rb;
	rb.r = lane<G>();
	rb.B = B;
	rb.b = b;
	rb.pad = pad;
	const bool rN = rb.r < N;
	const int rs = rN ? rb.r : 0;
	rb.q = sel0(rN, ld(P.q, rs, B, b));
	rb.dq = sel0(rN, ld(P.dq, rs, B, b));
	real g = 0;
	GMARK(13, "model_begin");
	{
		// Sai2Model::updateModel(): kinematics, M (CRBA), M^-1 (examples/05-using_robot_controller.cpp:143-145)
		fk_scan<G>(P.model, rb.r, rb.q, rb.FR, rb.Fp);
		GMARK(14, "model_fk_done");
		real Mrow[N];
		crba_g<G>(P.model, rb.r, rb.FR, rb.Fp, Mrow, false, &g);
		GMARK(15, "model_crba_done");
		UNROLL for (int j = 0; j < N; j++) rb.minv[j] = Mrow[j];
		spd_inverse_rows<G, N>(rb.minv);
		GMARK(16, "model_minv_done");
		// bounded inertia estimate (SingularityHandler.cpp:176-182, JointTask.cpp:254-260), shared
		const bool any_bie = P.any_bie != 0;  // (host: upload_params)
		const real thr = P.bie_thr;
		if (any_bie) {
			UNROLL for (int j = 0; j < N; j++) rb.minvB[j] = (j == rb.r) ? fmax(Mrow[j], thr) : Mrow[j];
			spd_inverse_rows<G, N>(rb.minvB);
		} else {
			UNROLL for (int j = 0; j < N; j++) rb.minvB[j] = rb.minv[j];
		}
	}
	GMARK(17, "model_end");
	real nprecT[N];
	UNROLL for (int j = 0; j < N; j++) nprecT[j] = rN ? (Nprec_in ? ld(Nprec_in, j * N + rs, B, b) : kd(rb.r, j)) : 0.0;
	const real tp = (rN && tau_prec) ? ld(tau_prec, rs, B, b) : 0.0;
	real tau = tp;
	ChainG chain;
	chain.ok = false;
	chain.wrows = 0;
	UNROLL for (int j = 0; j < N; j++) chain.W[j] = 0;
	const DevTask& tk = P.task[task];
	if (tk.type == SAI2B_MOTION_FORCE_TASK)
		mft_task_g<G>(P, tk, rb, false, false, commit_sh != 0, do_torque != 0, nprecT, tau, chain, N_out);
	else
		jt_task_g<G>(P, tk, rb, false, false, tau_prec != nullptr, do_torque != 0, nprecT, tau, chain, N_out);
	if (Ntot_out && rN) {
		UNROLL for (int j = 0; j < N; j++) st(Ntot_out, j * N + rs, B, b, nprecT[j]);
	}
	if (do_torque && tau_out && rN) st(tau_out, rs, B, b, tau - tp);
}

}  // namespace grp
}  // namespace sai2b
