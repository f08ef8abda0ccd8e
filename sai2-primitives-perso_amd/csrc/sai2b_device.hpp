// sai2b_device.hpp — per-robot device code of the batched operational-space controller.
//
// Execution model: ONE LANE PER ROBOT (64 robots per wavefront). Every matrix of a robot lives in
// that lane's registers as a fully unrolled fixed-size array, so there is no cross-lane traffic at
// all and global loads/stores are coalesced over the batch axis (SoA, batch-minor). See DESIGN.md
// for why this beats the "one wavefront per robot" layout on CDNA4.
//
// The mathematics follows the reference (file:line cited per function) but is re-expressed with
// range PROJECTORS instead of explicit reduced bases, so that the shapes are static while the
// per-robot rank split of the SingularityHandler is data:
//     U_x (U_x^T A U_x)^-1 U_x^T  ==  (Pi A Pi + I - Pi)^-1 - (I - Pi),   Pi = U_x U_x^T
// Everything the reference computes from a reduced basis is invariant under the choice of the
// orthonormal basis of that range, so the results are identical up to rounding.
#pragma once
#include <hip/hip_runtime.h>

#include "sai2b_params.h"
#include "../../include/sai2b_detfk.h"

namespace sai2b {

#define DI __device__ __forceinline__
#define UNROLL _Pragma("unroll")
typedef double real;

// ------------------------------------------------------------------ tiny dense helpers
template <int M, int K, int Nn>
DI void mm(const real* A, const real* B, real* C) {  // C = A(MxK) B(KxN)
	UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j < Nn; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < K; l++) s = fma(A[i * K + l], B[l * Nn + j], s);
		C[i * Nn + j] = s;
	}
}
template <int M, int K, int Nn>
DI void mm_tn(const real* A, const real* B, real* C) {	// C(MxN) = A^T B, A is KxM
	UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j < Nn; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < K; l++) s = fma(A[l * M + i], B[l * Nn + j], s);
		C[i * Nn + j] = s;
	}
}
template <int M, int K, int Nn>
DI void mm_nt(const real* A, const real* B, real* C) {	// C(MxN) = A B^T, B is NxK
	UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j < Nn; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < K; l++) s = fma(A[i * K + l], B[j * K + l], s);
		C[i * Nn + j] = s;
	}
}
// symmetric product C(MxM) = A B^T where the result is known to be symmetric (computes the lower
// triangle and mirrors it)
template <int M, int K>
DI void mm_nt_sym(const real* A, const real* B, real* C) {
	UNROLL for (int i = 0; i < M; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < K; l++) s = fma(A[i * K + l], B[j * K + l], s);
		C[i * M + j] = s;
		C[j * M + i] = s;
	}
}
template <int M, int K>
DI void mv(const real* A, const real* x, real* y) {	 // y = A x
	UNROLL for (int i = 0; i < M; i++) {
		real s = 0;
		UNROLL for (int l = 0; l < K; l++) s = fma(A[i * K + l], x[l], s);
		y[i] = s;
	}
}
template <int M, int K>
DI void mv_t(const real* A, const real* x, real* y) {  // y(K) = A^T x, A is MxK
	UNROLL for (int j = 0; j < K; j++) {
		real s = 0;
		UNROLL for (int l = 0; l < M; l++) s = fma(A[l * K + j], x[l], s);
		y[j] = s;
	}
}
DI void cross3(const real* a, const real* b, real* c) {
	c[0] = a[1] * b[2] - a[2] * b[1];
	c[1] = a[2] * b[0] - a[0] * b[2];
	c[2] = a[0] * b[1] - a[1] * b[0];
}
DI void mv3(const real* A, const real* x, real* y) { mv<3, 3>(A, x, y); }

// Inverse of a symmetric positive definite n x n matrix through its Cholesky factor. Stands in for
// Eigen's .inverse() on the SPD matrices of the path (SingularityHandler.cpp:120,182,190,201,212,
// JointTask.cpp:260-265, sai2-model M^-1). A non-positive pivot yields NaN/inf like a singular LU.
template <int n>
DI void spd_inverse(const real* A, real* Ai) {
	real L[n * n], d[n];
	UNROLL for (int j = 0; j < n; j++) {
		real s = A[j * n + j];
		UNROLL for (int k = 0; k < j; k++) s = fma(-L[j * n + k], L[j * n + k], s);
		real r = rsqrt(s);
		d[j] = r;
		L[j * n + j] = s * r;
		UNROLL for (int i = j + 1; i < n; i++) {
			real t = A[i * n + j];
			UNROLL for (int k = 0; k < j; k++) t = fma(-L[i * n + k], L[j * n + k], t);
			L[i * n + j] = t * r;
		}
	}
	real Li[n * n];	 // inverse of L (lower)
	UNROLL for (int j = 0; j < n; j++) {
		Li[j * n + j] = d[j];
		UNROLL for (int i = j + 1; i < n; i++) {
			real t = 0;
			UNROLL for (int k = j; k < i; k++) t = fma(L[i * n + k], Li[k * n + j], t);
			Li[i * n + j] = -t * d[i];
		}
	}
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = 0;
		UNROLL for (int k = i; k < n; k++) s = fma(Li[k * n + i], Li[k * n + j], s);
		Ai[i * n + j] = s;
		Ai[j * n + i] = s;
	}
}

// U_x (U_x^T A U_x)^-1 U_x^T for the orthogonal projector Pi = U_x U_x^T (n x n, symmetric A)
template <int n>
DI void pinv_proj(const real* A, const real* Pi, real* out) {
	real T[n * n], X[n * n];
	mm<n, n, n>(Pi, A, T);
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		real s = (i == j ? 1.0 : 0.0) - Pi[i * n + j];
		UNROLL for (int l = 0; l < n; l++) s = fma(T[i * n + l], Pi[l * n + j], s);
		X[i * n + j] = s;
		X[j * n + i] = s;
	}
	spd_inverse<n>(X, out);
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j < n; j++)
		out[i * n + j] -= (i == j ? 1.0 : 0.0) - Pi[i * n + j];
}

// One-sided (Hestenes) Jacobi: X (ROWS x COLS) W = orthogonal columns; W accumulates the rotations.
// Same cyclic order, threshold and rotation formula as the oracle (oracle/sai2_oracle.c:hestenes);
// stands in for Eigen::JacobiSVD (SingularityHandler.cpp:78-81) and the SVD inside
// Sai2Model::matrixRangeBasis (JointTask.cpp:233).
template <int ROWS, int COLS>
DI void hestenes(real* X, real* W) {
	UNROLL for (int i = 0; i < COLS; i++) UNROLL for (int j = 0; j < COLS; j++) W[i * COLS + j] = (i == j) ? 1.0 : 0.0;
#pragma unroll 1
	for (int sweep = 0; sweep < 60; sweep++) {
		bool rotated = false;
		UNROLL for (int i = 0; i < COLS - 1; i++) UNROLL for (int j = i + 1; j < COLS; j++) {
			real al = 0, be = 0, ga = 0;
			UNROLL for (int r = 0; r < ROWS; r++) {
				real xi = X[r * COLS + i], xj = X[r * COLS + j];
				al = fma(xi, xi, al);
				be = fma(xj, xj, be);
				ga = fma(xi, xj, ga);
			}
			if (fabs(ga) > 1e-15 * sqrt(al * be)) {
				rotated = true;
				real zeta = (be - al) / (2 * ga);
				real t = copysign(1.0, zeta) / (fabs(zeta) + sqrt(fma(zeta, zeta, 1.0)));
				real c = rsqrt(fma(t, t, 1.0)), s = c * t;
				UNROLL for (int r = 0; r < ROWS; r++) {
					real xi = X[r * COLS + i], xj = X[r * COLS + j];
					X[r * COLS + i] = c * xi - s * xj;
					X[r * COLS + j] = s * xi + c * xj;
				}
				UNROLL for (int r = 0; r < COLS; r++) {
					real wi = W[r * COLS + i], wj = W[r * COLS + j];
					W[r * COLS + i] = c * wi - s * wj;
					W[r * COLS + j] = s * wi + c * wj;
				}
			}
		}
		if (!rotated) break;
	}
}

// Sai2Model::orientationError(desired, current)
DI void orientation_error(const real* Rd, const real* Rc, real* e) {
	e[0] = e[1] = e[2] = 0;
	UNROLL for (int i = 0; i < 3; i++) {
		real c[3] = {Rc[i], Rc[3 + i], Rc[6 + i]}, d[3] = {Rd[i], Rd[3 + i], Rd[6 + i]}, x[3];
		cross3(c, d, x);
		UNROLL for (int k = 0; k < 3; k++) e[k] = fma(-0.5, x[k], e[k]);
	}
}
DI real gain_pinv(real k) { return fabs(k) > 1e-6 ? 1.0 / k : 0.0; }

// sin and cos of a joint angle. Joint angles are bounded (|q| of a few pi at most), so the argument
// reduction is a two-term Cody-Waite step with FMAs (k * PIO2_HI is exact inside the fma) followed by
// the classic minimax kernels on [-pi/4, pi/4] (coefficients as in fdlibm's __kernel_sin/__kernel_cos);
// results agree with libm to 1 ulp without OCML's large-argument (Payne-Hanek) machinery.
DI void sincos_joint(real x, real* sn, real* cs) {
	const real kf = rint(x * 0.63661977236758134308);  // 2/pi
	real r = fma(-kf, 1.57079632679489655800e+00, x);
	r = fma(-kf, 6.12323399573676603587e-17, r);
	const real z = r * r;
	const real ps = fma(z, fma(z, fma(z, fma(z, fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08),
										 2.75573137070700676789e-06), -1.98412698298579493134e-04), 8.33333333332248946124e-03), -1.66666666666666324348e-01);
	const real s0 = fma(r * z, ps, r);
	const real pc = fma(z, fma(z, fma(z, fma(z, fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09),
										 -2.75573143513906633035e-07), 2.48015872894767294178e-05), -1.38888888888741095749e-03), 4.16666666666666019037e-02);
	const real c0 = fma(z * z, pc, fma(-0.5, z, 1.0));
	const int k = (int)kf;
	const real s1 = (k & 1) ? c0 : s0, c1 = (k & 1) ? s0 : c0;
	*sn = (k & 2) ? -s1 : s1;
	*cs = ((k + 1) & 2) ? -c1 : c1;
}

// ------------------------------------------------------------------ model (sai2-model subset)
struct Frames {
	real R[N][9];
	real p[N][3];
};
template <class MD>
DI void fk(const MD& md, const real* q, Frames& F) {
	real Rp[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, pp[3] = {0, 0, 0};
	UNROLL for (int i = 0; i < N; i++) {
		real RE[9];
		if (i == 0) {
			UNROLL for (int k = 0; k < 3; k++) F.p[0][k] = md.xyz[0][k];
			UNROLL for (int k = 0; k < 9; k++) RE[k] = md.E[0][k];
		} else {
			UNROLL for (int k = 0; k < 3; k++)
				F.p[i][k] = fma(Rp[3 * k], md.xyz[i][0], fma(Rp[3 * k + 1], md.xyz[i][1], fma(Rp[3 * k + 2], md.xyz[i][2], pp[k])));
			mm<3, 3, 3>(Rp, md.E[i], RE);
		}
		real s, c;
		sincos_joint(q[i], &s, &c);
		const bool pris = md.jtype[i] != 0;	 // prismatic: the frame slides along its z instead of turning about it
		if (pris) s = 0, c = 1;
		UNROLL for (int k = 0; k < 3; k++) {
			F.R[i][3 * k + 0] = fma(c, RE[3 * k], s * RE[3 * k + 1]);
			F.R[i][3 * k + 1] = fma(c, RE[3 * k + 1], -s * RE[3 * k]);
			F.R[i][3 * k + 2] = RE[3 * k + 2];
			if (pris) F.p[i][k] = fma(q[i], RE[3 * k + 2], F.p[i][k]);
		}
		UNROLL for (int k = 0; k < 9; k++) Rp[k] = F.R[i][k];
		UNROLL for (int k = 0; k < 3; k++) pp[k] = F.p[i][k];
	}
}
// positionInWorld / rotationInWorld of the compliant frame (MotionForceTask.cpp:286-289)
DI void frame_pose(const DevTask& t, const Frames& F, real* x, real* R) {
	// link is batch-uniform; select the link frame without dynamic register indexing
	real Rl[9], pl[3];
	UNROLL for (int k = 0; k < 9; k++) Rl[k] = F.R[N - 1][k];
	UNROLL for (int k = 0; k < 3; k++) pl[k] = F.p[N - 1][k];
	UNROLL for (int i = 0; i < N - 1; i++)
		if (t.link == i) {
			UNROLL for (int k = 0; k < 9; k++) Rl[k] = F.R[i][k];
			UNROLL for (int k = 0; k < 3; k++) pl[k] = F.p[i][k];
		}
	UNROLL for (int k = 0; k < 3; k++)
		x[k] = fma(Rl[3 * k], t.frame_pos[0], fma(Rl[3 * k + 1], t.frame_pos[1], fma(Rl[3 * k + 2], t.frame_pos[2], pl[k])));
	mm<3, 3, 3>(Rl, t.frame_rot, R);
}
// The pose goals and internal generators START from (reInitializeTask, enableInternalOtg*, re-parametrisation of
// the force / motion spaces): the bit-reproducible restatement shared with the test oracle (include/sai2b_detfk.h),
// so that both sides' generators see identical bits. The torque path keeps fk() / frame_pose() above.
DI void det_frame_pose(const DevModel& md, const DevTask& t, const real* q, real* x, real* R) {
	sai2b_det_frame_pose(&md.E[0][0], &md.xyz[0][0], md.jtype, q, t.link, t.frame_pos, t.frame_rot, x, R);
}
// Sai2Model::JWorldFrame(link, pos): 6 x n, linear rows first; a prismatic joint's column is (z, 0)
template <class MD>
DI void jacobian(const MD& md, const DevTask& t, const Frames& F, const real* x, real* J) {
	UNROLL for (int i = 0; i < N; i++) {
		real z[3] = {F.R[i][2], F.R[i][5], F.R[i][8]};
		real d[3] = {x[0] - F.p[i][0], x[1] - F.p[i][1], x[2] - F.p[i][2]}, v[3];
		cross3(z, d, v);
		const bool on = i <= t.link, pris = md.jtype[i] != 0;
		UNROLL for (int k = 0; k < 3; k++) {
			J[k * N + i] = on ? (pris ? z[k] : v[k]) : 0.0;
			J[(3 + k) * N + i] = (on && !pris) ? z[k] : 0.0;
		}
	}
}
// Joint-space inertia matrix by the composite-rigid-body algorithm with spatial inertias expressed
// about the world origin (what Sai2Model::updateModel() obtains from RBDL's CRBA).
template <class MD>
DI void mass_matrix(const MD& md, const Frames& F, real* M) {
	real z[N][3], v[N][3];	// joint twists about the world origin: (z_i, p_i x z_i) revolute, (0, z_i) prismatic
	UNROLL for (int i = 0; i < N; i++) {
		const real ax[3] = {F.R[i][2], F.R[i][5], F.R[i][8]};
		const bool pris = md.jtype[i] != 0;
		real pxz[3];
		cross3(F.p[i], ax, pxz);
		UNROLL for (int k = 0; k < 3; k++) {
			z[i][k] = pris ? 0.0 : ax[k];
			v[i][k] = pris ? ax[k] : pxz[k];
		}
	}
	real mt = 0, h[3] = {0, 0, 0}, IO[6] = {0, 0, 0, 0, 0, 0};	// xx yy zz xy xz yz
	UNROLL for (int k = N - 1; k >= 0; k--) {
		const real* R = F.R[k];
		real c[3];
		UNROLL for (int a = 0; a < 3; a++)
			c[a] = fma(R[3 * a], md.com[k][0], fma(R[3 * a + 1], md.com[k][1], fma(R[3 * a + 2], md.com[k][2], F.p[k][a])));
		const real* li = md.inertia[k];
		real Il[9] = {li[0], li[3], li[4], li[3], li[1], li[5], li[4], li[5], li[2]}, T[9];
		mm<3, 3, 3>(R, Il, T);
		const real m = md.mass[k];
		const real c2 = c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
		const int ia[6] = {0, 1, 2, 0, 0, 1}, ib[6] = {0, 1, 2, 1, 2, 2};
		UNROLL for (int e = 0; e < 6; e++) {
			real s = 0;
			UNROLL for (int l = 0; l < 3; l++) s = fma(T[3 * ia[e] + l], R[3 * ib[e] + l], s);
			IO[e] += s + m * ((ia[e] == ib[e] ? c2 : 0.0) - c[ia[e]] * c[ib[e]]);
		}
		mt += m;
		UNROLL for (int a = 0; a < 3; a++) h[a] = fma(m, c[a], h[a]);
		// wrench of the composite body under unit acceleration of joint k
		real n[3], f[3], hv[3], hz[3];
		cross3(h, v[k], hv);
		cross3(h, z[k], hz);
		n[0] = IO[0] * z[k][0] + IO[3] * z[k][1] + IO[4] * z[k][2] + hv[0];
		n[1] = IO[3] * z[k][0] + IO[1] * z[k][1] + IO[5] * z[k][2] + hv[1];
		n[2] = IO[4] * z[k][0] + IO[5] * z[k][1] + IO[2] * z[k][2] + hv[2];
		UNROLL for (int a = 0; a < 3; a++) f[a] = mt * v[k][a] - hz[a];
		UNROLL for (int j = 0; j <= k; j++) {
			real s = z[j][0] * n[0] + z[j][1] * n[1] + z[j][2] * n[2] + v[j][0] * f[0] + v[j][1] * f[1] + v[j][2] * f[2];
			M[k * N + j] = s;
			M[j * N + k] = s;
		}
	}
}
// Sai2Model::jointGravityVector (RobotController.cpp:71): g_i = -sum_k m_k (z_i x (c_k - p_i)) . gravity
template <class MD>
DI void gravity_vector(const MD& md, const Frames& F, real* g) {
	real mt = 0, h[3] = {0, 0, 0};
	UNROLL for (int k = N - 1; k >= 0; k--) {
		const real* R = F.R[k];
		real c[3];
		UNROLL for (int a = 0; a < 3; a++)
			c[a] = fma(R[3 * a], md.com[k][0], fma(R[3 * a + 1], md.com[k][1], fma(R[3 * a + 2], md.com[k][2], F.p[k][a])));
		mt += md.mass[k];
		UNROLL for (int a = 0; a < 3; a++) h[a] = fma(md.mass[k], c[a], h[a]);
		real d[3] = {h[0] - mt * F.p[k][0], h[1] - mt * F.p[k][1], h[2] - mt * F.p[k][2]};
		real zk[3] = {R[2], R[5], R[8]}, x[3];
		cross3(zk, d, x);
		if (md.jtype[k] != 0) {	 // prismatic: the weight of everything outboard along the axis
			UNROLL for (int a = 0; a < 3; a++) x[a] = mt * zk[a];
		}
		g[k] = -(x[0] * md.gravity[0] + x[1] * md.gravity[1] + x[2] * md.gravity[2]);
	}
}

// ------------------------------------------------------------------ per-robot tick
struct RobotCtx {
	real q[N], dq[N];
	real Minv[N * N], MinvB[N * N];
};

// Batched arrays are device (global) memory: say so, or pointers read from the parameter block are
// generic and every access becomes a flat_load/flat_store that also ties up lgkmcnt.
typedef __attribute__((address_space(1))) real greal;
typedef __attribute__((address_space(1))) int gint;
DI real ld(const real* p, int row, int B, int b) { return ((const greal*)p)[(size_t)row * B + b]; }
DI void st(real* p, int row, int B, int b, real v) { ((greal*)p)[(size_t)row * B + b] = v; }
DI int ldi(const int* p, int row, int B, int b) { return ((const gint*)p)[(size_t)row * B + b]; }
DI void sti(int* p, int row, int B, int b, int v) { ((gint*)p)[(size_t)row * B + b] = v; }


// sigma matrices (MotionForceTask.cpp:892-971): sf = sigmaForce / sigmaMoment, sp = sigmaPosition /
// sigmaOrientation for the 3x3 block `blk` of the partial-task projection
DI void sigma_pair(const DevTask& t, int blk, int dim, const real* axis, const real* Rw, real* sf, real* sp) {
	real Pb[9], A[9], T[9], a[3];
	UNROLL for (int i = 0; i < 3; i++) UNROLL for (int j = 0; j < 3; j++) Pb[3 * i + j] = t.P[(3 * blk + i) * 6 + 3 * blk + j];
	if (t.in_frame)
		mv3(Rw, axis, a);
	else {
		UNROLL for (int i = 0; i < 3; i++) a[i] = axis[i];
	}
	UNROLL for (int i = 0; i < 3; i++) UNROLL for (int j = 0; j < 3; j++) {
		real aa = a[i] * a[j], id = (i == j) ? 1.0 : 0.0;
		A[3 * i + j] = dim == 1 ? aa : (dim == 2 ? id - aa : (dim == 3 ? id : 0.0));
	}
	mm<3, 3, 3>(Pb, A, T);
	mm_nt<3, 3, 3>(T, Pb, sf);
	if (dim == 3) {
		UNROLL for (int i = 0; i < 9; i++) sf[i] = Pb[i];
	}
	UNROLL for (int i = 0; i < 3; i++) UNROLL for (int j = 0; j < 3; j++) A[3 * i + j] = ((i == j) ? 1.0 : 0.0) - sf[3 * i + j];
	mm<3, 3, 3>(Pb, A, T);
	mm_nt<3, 3, 3>(T, Pb, sp);
}

// Per-robot inputs of the MotionForceTask law: goals, sensed wrench, integrator state. Loaded in one
// burst (mft_load) so a caller can issue the loads long before the law consumes them.
struct MftIn {
	real g_pos[3], g_rot[9], g_v[3], g_w[3], g_a[3], g_al[3], g_f[3], g_m[3], s_f[3], s_m[3];
	real integ[12];	 // pos 3, ori 3, force 3, moment 3
};
DI void mft_load(const DevTask& t, int B, int b, MftIn& in) {
	const real* G = t.law_goals;  // goal, or the internal OTG's next state (MotionForceTask.cpp:386-407)
	UNROLL for (int k = 0; k < 3; k++) {
		in.g_pos[k] = ld(G, k, B, b);
		in.g_v[k] = ld(G, 12 + k, B, b);
		in.g_w[k] = ld(G, 15 + k, B, b);
		in.g_a[k] = ld(G, 18 + k, B, b);
		in.g_al[k] = ld(G, 21 + k, B, b);
		in.g_f[k] = in.g_m[k] = in.s_f[k] = in.s_m[k] = 0;
	}
	UNROLL for (int k = 0; k < 9; k++) in.g_rot[k] = ld(G, 3 + k, B, b);
	UNROLL for (int k = 0; k < 6; k++) in.integ[k] = ld(t.state, k, B, b);
	UNROLL for (int k = 6; k < 12; k++) in.integ[k] = 0;
	if ((t.fdim | t.mdim) != 0) {  // batch-uniform
		UNROLL for (int k = 0; k < 3; k++) {
			in.g_f[k] = ld(t.goals, 24 + k, B, b);
			in.g_m[k] = ld(t.goals, 27 + k, B, b);
		}
		if (t.cl_force || t.cl_moment) {
			UNROLL for (int k = 0; k < 3; k++) {
				in.s_f[k] = ld(t.sensed, k, B, b);
				in.s_m[k] = ld(t.sensed, 3 + k, B, b);
			}
		}
	}
	// read whenever they are written back (mft_store_integrators), also with no force / moment space
	// parametrised (sigma = 0: they do not move, and keep their values for when a space comes back)
	if (t.cl_force || t.cl_moment) {
		UNROLL for (int k = 6; k < 12; k++) in.integ[k] = ld(t.state, k, B, b);
	}
}

// integrators advance on every torque computation, even with ki = 0 (MotionForceTask.cpp:411-413,446);
// the force/moment ones only in closed-loop mode (:329-331,359-361)
DI void mft_store_integrators(const DevTask& t, int B, int b, const MftIn& in) {
	UNROLL for (int k = 0; k < 6; k++) st(t.state, k, B, b, in.integ[k]);
	if (t.cl_force) {
		UNROLL for (int k = 6; k < 9; k++) st(t.state, k, B, b, in.integ[k]);
	}
	if (t.cl_moment) {
		UNROLL for (int k = 9; k < 12; k++) st(t.state, k, B, b, in.integ[k]);
	}
}

// MotionForceTask::computeTorques() control law up to the task forces (MotionForceTask.cpp:278-503).
// Produces F_unit (unit-mass motion force) and F_force (force-related terms + feed-forward).
// The integrators in `in.integ` are advanced in place; mft_store_integrators() writes them back.
// POPCExplicitForceControl::computePassivitySaturatedForce with the observer ENABLED
// (POPCExplicitForceControl.cpp:37-95): windowed passivity observer on the force loop and the
// controller scaling Rc re-evaluated every 50 ticks. State per robot in t.popc_*; committed only when
// `commit`. The reference's std::queue is unbounded; the ring holds POPC_RING samples and, if a robot
// stays active longer than that, retires the oldest sample as the reference's pop would.
DI void popc_force(const DevTask& t, int B, int b, bool commit, const real* fd, const real* fs, const real* vcl,
				   const real* vr, real* out) {
	real po = ld(t.popc_f, 0, B, b), ecorr = ld(t.popc_f, 1, B, b), vsum = ld(t.popc_f, 2, B, b), Rc = ld(t.popc_f, 3, B, b);
	int counter = ldi(t.popc_i, 0, B, b), head = ldi(t.popc_i, 1, B, b), size = ldi(t.popc_i, 2, B, b);
	real fcmd[3], vc2 = 0, p_in = 0;
	UNROLL for (int k = 0; k < 3; k++) {
		fcmd[k] = t.kff_f * fd[k] + Rc * vcl[k] - t.kv_f[k] * vr[k];
		vc2 = fma(vcl[k], vcl[k], vc2);
		p_in += (fs[k] - fd[k]) * vcl[k] - fcmd[k] * vr[k];
	}
	p_in *= t.dt;
	po += p_in;
	if (size == POPC_RING) {  // overflow: retire the oldest sample
		const real front = ld(t.popc_q, head, B, b);
		if (front > 0) po -= front;
		head = (head + 1) % POPC_RING;
		size--;
	}
	if (commit) st(t.popc_q, (head + size) % POPC_RING, B, b, p_in);
	real newest = p_in;	 // value at the back of the window (not yet visible in memory when !commit)
	size++;
	if (po + ecorr > 0) {
		while (size > POPC_WINDOW) {
			const real front = (size == 1) ? newest : ld(t.popc_q, head, B, b);
			if (po + ecorr > front) {
				if (front > 0) po -= front;
				head = (head + 1) % POPC_RING;
				size--;
			} else {
				break;
			}
		}
	}
	if (counter <= 0) {
		counter = POPC_MAX_COUNTER;
		const real old_Rc = Rc;
		if (po + ecorr < 0) {
			Rc = 1 + (po + ecorr) / (vsum * t.dt);
			Rc = (Rc > 1) ? 1.0 : ((Rc < 0) ? 0.0 : Rc);
		} else {
			Rc = (1 + (0.1 * POPC_MAX_COUNTER - 1) * Rc) / (0.1 * POPC_MAX_COUNTER);
		}
		ecorr += (1 - old_Rc) * vsum * t.dt;
		vsum = 0;
	}
	counter--;
	vsum += vc2;
	UNROLL for (int k = 0; k < 3; k++) out[k] = Rc * vcl[k] - t.kv_f[k] * vr[k];
	if (commit) {
		st(t.popc_f, 0, B, b, po);
		st(t.popc_f, 1, B, b, ecorr);
		st(t.popc_f, 2, B, b, vsum);
		st(t.popc_f, 3, B, b, Rc);
		sti(t.popc_i, 0, B, b, counter);
		sti(t.popc_i, 1, B, b, head);
		sti(t.popc_i, 2, B, b, size);
	}
}

// v, w: linear and angular velocity of the control frame (J dq, MotionForceTask.cpp:293-298)
DI void mft_law_vw(const DevTask& t, const real* v, const real* w, const real* x, const real* R, MftIn& in, real* Fu, real* Ff,
				   int B = 0, int b = 0, bool commit = false) {
	if (t.plain_motion) {
		// full task, no force space, world-frame gains, no velocity saturation: sigma_position =
		// sigma_orientation = I, sigma_force = sigma_moment = 0 (MotionForceTask.cpp:431-436,463-467)
		real oe[3];
		orientation_error(in.g_rot, R, oe);
		UNROLL for (int k = 0; k < 3; k++) {
			const real ex = x[k] - in.g_pos[k];
			in.integ[k] = fma(ex, t.dt, in.integ[k]);
			in.integ[3 + k] = fma(oe[k], t.dt, in.integ[3 + k]);
			Fu[k] = in.g_a[k] - t.kp_pos[k] * ex - t.kv_pos[k] * (v[k] - in.g_v[k]) - t.ki_pos[k] * in.integ[k];
			Fu[3 + k] = in.g_al[k] - t.kp_ori[k] * oe[k] - t.kv_ori[k] * (w[k] - in.g_w[k]) - t.ki_ori[k] * in.integ[3 + k];
			Ff[k] = Ff[3 + k] = 0;
		}
		return;
	}
	real sf[9], sp[9], sm[9], so[9];
	if (t.in_frame) {
		sigma_pair(t, 0, t.fdim, t.faxis, R, sf, sp);
		sigma_pair(t, 1, t.mdim, t.maxis, R, sm, so);
	} else {  // world-frame parametrisation: the four selection matrices are batch-uniform (host-made)
		UNROLL for (int i = 0; i < 9; i++) {
			sf[i] = t.sig[0][i];
			sp[i] = t.sig[1][i];
			sm[i] = t.sig[2][i];
			so[i] = t.sig[3][i];
		}
	}
	const real *g_pos = in.g_pos, *g_rot = in.g_rot, *g_v = in.g_v, *g_w = in.g_w, *g_a = in.g_a, *g_al = in.g_al;
	const bool uses_force = (t.fdim | t.mdim) != 0;	 // batch-uniform
	real gf[3] = {0, 0, 0}, gm[3] = {0, 0, 0}, fs_w[3] = {0, 0, 0}, ms_w[3] = {0, 0, 0};
	if (uses_force) {
		if (t.in_frame) {  // getGoalForce / getGoalMoment (MotionForceTask.cpp:755-769)
			mv3(R, in.g_f, gf);
			mv3(R, in.g_m, gm);
		} else {
			UNROLL for (int k = 0; k < 3; k++) {
				gf[k] = in.g_f[k];
				gm[k] = in.g_m[k];
			}
		}
		if (t.cl_force || t.cl_moment) {  // updateSensedForceAndMoment (MotionForceTask.cpp:805-828)
			real fc[3], mc[3], tmp[3];
			mv3(t.sensor_rot, in.s_f, fc);
			mv3(t.sensor_rot, in.s_m, mc);
			cross3(t.sensor_pos, fc, tmp);
			UNROLL for (int k = 0; k < 3; k++) mc[k] += tmp[k];
			mv3(R, fc, fs_w);
			mv3(R, mc, ms_w);
		}
	}
	const real dt = t.dt;
	real f_force[3], f_moment[3], e[3], y[3];
	// force (MotionForceTask.cpp:327-354); POPC disabled -> vcl - kv vr (POPCExplicitForceControl.cpp:33-35)
	if (t.cl_force) {
		real integ[3], fb[3], vcl[3], vr[3];
		UNROLL for (int k = 0; k < 3; k++) e[k] = fs_w[k] - gf[k];
		mv3(sf, e, y);
		UNROLL for (int k = 0; k < 3; k++) {
			integ[k] = fma(y[k], dt, in.integ[6 + k]);
			in.integ[6 + k] = integ[k];
			e[k] = -t.kp_f[k] * (fs_w[k] - gf[k]) - t.ki_f[k] * integ[k];
		}
		mv3(sf, e, fb);
		real n = sqrt(fb[0] * fb[0] + fb[1] * fb[1] + fb[2] * fb[2]);
		if (n > t.max_f) {
			UNROLL for (int k = 0; k < 3; k++) fb[k] *= t.max_f / n;
		}
		mv3(sf, fb, vcl);
		mv3(sf, v, vr);
		if (t.passivity) {
			real fd[3], fs[3];
			mv3(sf, gf, fd);
			mv3(sf, fs_w, fs);
			popc_force(t, B, b, commit, fd, fs, vcl, vr, f_force);
		} else {
			UNROLL for (int k = 0; k < 3; k++) f_force[k] = vcl[k] - t.kv_f[k] * vr[k];
		}
	} else {
		UNROLL for (int k = 0; k < 3; k++) e[k] = -t.kv_f[k] * v[k];
		mv3(sf, e, f_force);
	}
	// moment (MotionForceTask.cpp:356-383)
	if (t.cl_moment) {
		real integ[3], fb[3];
		UNROLL for (int k = 0; k < 3; k++) e[k] = ms_w[k] - gm[k];
		mv3(sm, e, y);
		UNROLL for (int k = 0; k < 3; k++) {
			integ[k] = fma(y[k], dt, in.integ[9 + k]);
			in.integ[9 + k] = integ[k];
			e[k] = -t.kp_m[k] * (ms_w[k] - gm[k]) - t.ki_m[k] * integ[k];
		}
		mv3(sm, e, fb);
		real n = sqrt(fb[0] * fb[0] + fb[1] * fb[1] + fb[2] * fb[2]);
		if (n > t.max_m) {
			UNROLL for (int k = 0; k < 3; k++) fb[k] *= t.max_m / n;
		}
		UNROLL for (int k = 0; k < 3; k++) e[k] = fb[k] - t.kv_m[k] * w[k];
		mv3(sm, e, f_moment);
	} else {
		UNROLL for (int k = 0; k < 3; k++) e[k] = -t.kv_m[k] * w[k];
		mv3(sm, e, f_moment);
	}
	// linear motion (MotionForceTask.cpp:385-437); desired = goal (internal OTG: "next" row)
	real f_pos[3], f_ori[3], ip[3], io[3], des[3];
	UNROLL for (int k = 0; k < 3; k++) e[k] = x[k] - g_pos[k];
	mv3(sp, e, y);
	UNROLL for (int k = 0; k < 3; k++) {
		ip[k] = fma(y[k], dt, in.integ[k]);
		in.integ[k] = ip[k];
	}
	if (t.use_vsat) {
		UNROLL for (int k = 0; k < 3; k++) {
			real kvi = gain_pinv(t.kv_pos[k]);
			des[k] = -t.kp_pos[k] * kvi * y[k] - t.ki_pos[k] * kvi * ip[k];
		}
		real n = sqrt(des[0] * des[0] + des[1] * des[1] + des[2] * des[2]);
		if (n > t.lin_vsat) {
			UNROLL for (int k = 0; k < 3; k++) des[k] *= t.lin_vsat / n;
		}
		UNROLL for (int k = 0; k < 3; k++) e[k] = g_a[k] - t.kv_pos[k] * (v[k] - des[k]);
	} else {
		UNROLL for (int k = 0; k < 3; k++)
			e[k] = g_a[k] - t.kp_pos[k] * (x[k] - g_pos[k]) - t.kv_pos[k] * (v[k] - g_v[k]) - t.ki_pos[k] * ip[k];
	}
	mv3(sp, e, f_pos);
	// angular motion (MotionForceTask.cpp:439-468)
	real oe[3], step[3];
	orientation_error(g_rot, R, oe);
	mv3(so, oe, step);
	UNROLL for (int k = 0; k < 3; k++) {
		io[k] = fma(step[k], dt, in.integ[3 + k]);
		in.integ[3 + k] = io[k];
	}
	if (t.use_vsat) {
		UNROLL for (int k = 0; k < 3; k++) {
			real kvi = gain_pinv(t.kv_ori[k]);
			des[k] = -t.kp_ori[k] * kvi * step[k] - t.ki_ori[k] * kvi * io[k];
		}
		real n = sqrt(des[0] * des[0] + des[1] * des[1] + des[2] * des[2]);
		if (n > t.ang_vsat) {
			UNROLL for (int k = 0; k < 3; k++) des[k] *= t.ang_vsat / n;
		}
		UNROLL for (int k = 0; k < 3; k++) e[k] = g_al[k] - t.kv_ori[k] * (w[k] - des[k]);
	} else {
		UNROLL for (int k = 0; k < 3; k++)
			e[k] = g_al[k] - t.kp_ori[k] * step[k] - t.kv_ori[k] * (w[k] - g_w[k]) - t.ki_ori[k] * io[k];
	}
	mv3(so, e, f_ori);
	// task force (MotionForceTask.cpp:470-506)
	real ff[6];
	mv3(sf, gf, ff);
	mv3(sm, gm, ff + 3);
	if (t.cl_force) {  // sic: one flag scales both (MotionForceTask.cpp:484-487)
		UNROLL for (int k = 0; k < 3; k++) {
			ff[k] *= t.kff_f;
			ff[3 + k] *= t.kff_m;
		}
	}
	UNROLL for (int k = 0; k < 3; k++) {
		Fu[k] = f_pos[k];
		Fu[3 + k] = f_ori[k];
		Ff[k] = f_force[k] + ff[k];
		Ff[3 + k] = f_moment[k] + ff[3 + k];
	}
}
DI void mft_law(const DevTask& t, const RobotCtx& rc, const real* J, const real* x, const real* R, MftIn& in, real* Fu,
				real* Ff, int B = 0, int b = 0, bool commit = false) {
	real v[3], w[3];
	mv<3, N>(J, rc.dq, v);
	mv<3, N>(J + 3 * N, rc.dq, w);
	mft_law_vw(t, v, w, x, R, in, Fu, Ff, B, b, commit);
}

// (Jp A Jp^T) for a 6x7 Jp and symmetric 7x7 A
DI void sandwich6(const real* Jp, const real* A, real* out) {
	real T[6 * N];
	mm<6, N, N>(Jp, A, T);
	mm_nt_sym<6, N>(T, Jp, out);
}
DI void sandwich7(const real* Jp, const real* A, real* out) {
	real T[N * N];
	mm<N, N, N>(Jp, A, T);
	mm_nt_sym<N, N>(T, Jp, out);
}

// ------------------------------------------------------------------ SVD-free certificates
// Certificate that a symmetric PSD Gram matrix G (n x n) restricted to the range it lives in has
// lambda_max >= abs2 and lambda_min >= rel2 * lambda_max, i.e. for G = Jp Jp^T that s_0 >= sqrt(abs2)
// and s_min / s_0 >= sqrt(rel2): lambda_max(G) <= ub := tr(G^8)^(1/8) <= n^(1/8) lambda_max(G), and
// positive LDL^T pivots of G + ub Pc - rel2 ub I (Pc = projector onto the complement of the range, or
// NULL when the range is everything) imply the bound. Sufficient, never necessary: whoever fails it
// takes the Jacobi-SVD path, so decisions are the reference's in all cases.
template <int n>
DI bool certify_gram(const real* G, const real* Pc, real abs2, real rel2) {
	real G2[n * n], G4[n * n];
	mm_nt_sym<n, n>(G, G, G2);	// G symmetric: G G^T = G^2
	mm_nt_sym<n, n>(G2, G2, G4);
	real t8 = 0;
	UNROLL for (int i = 0; i < n; i++) UNROLL for (int j = 0; j <= i; j++) {
		real v = G4[i * n + j] * G4[i * n + j];
		t8 += (i == j) ? v : 2 * v;
	}
	const real ub = sqrt(sqrt(sqrt(t8)));
	bool ok = ub > 1.30 * abs2;	 // lambda_max >= ub / n^(1/8), 8^(1/8) = 1.2968
	const real c = rel2 * ub * (1.0 + 1e-9);
	const real floor_ = 1e-5 * c;
	real Lm[n * n], d[n];
	UNROLL for (int j = 0; j < n; j++) {
		real s = G[j * n + j] - c + (Pc ? ub * Pc[j * n + j] : 0.0);
		UNROLL for (int k = 0; k < j; k++) s = fma(-Lm[j * n + k] * Lm[j * n + k], d[k], s);
		d[j] = s;
		ok = ok && (s > floor_);
		const real inv = 1.0 / s;
		UNROLL for (int i = j + 1; i < n; i++) {
			real t = G[i * n + j] + (Pc ? ub * Pc[i * n + j] : 0.0);
			UNROLL for (int k = 0; k < j; k++) t = fma(-Lm[i * n + k] * Lm[j * n + k], d[k], t);
			Lm[i * n + j] = t * inv;
		}
	}
	return ok;
}

// Row space accumulated over the certified tasks of the hierarchy: W stacks the (full-row-rank)
// projected Jacobians, so range(N_prec) = null(W) and a full JointTask that follows gets its range
// projector as I - W^T (W W^T)^-1 W instead of an SVD of N_prec (JointTask.cpp:233).
struct Chain {
	bool ok;	// every task so far was certified by the whole wavefront
	int wrows;	// batch-uniform
	real W[N * N];
};
DI void chain_append(Chain& ch, const real* rows, int nrows) {
	UNROLL for (int R = 0; R < N; R++) UNROLL for (int i = 0; i < N; i++) {
		const bool hit = (i < nrows) && (R == ch.wrows + i);
		UNROLL for (int j = 0; j < N; j++) ch.W[R * N + j] = hit ? rows[i * N + j] : ch.W[R * N + j];
	}
	ch.wrows += nrows;
}

// MotionForceTask::updateTaskModel + SingularityHandler::updateTaskModel/classifySingularity +
// MotionForceTask::computeTorques + SingularityHandler::computeTorques for one robot
// (MotionForceTask.cpp:247-509, SingularityHandler.cpp:75-368).
template <bool DEBUG>
DI void mft_task(const DevParams& P, const DevTask& t, const RobotCtx& rc, int B, int b, bool first, bool last,
				 bool commit_sh, bool do_torque, real* Nprec, real* tau_total, Chain& chain, real* Ntask_out = nullptr) {
	Frames F;
	fk(P.model, rc.q, F);
	real x[3], R[9], Jw[6 * N], J[6 * N], Jp[6 * N];
	frame_pose(t, F, x, R);
	jacobian(P.model, t, F, x, Jw);
	if (t.full_projection) {
		UNROLL for (int i = 0; i < 6 * N; i++) J[i] = Jw[i];
	} else {
		mm<6, 6, N>(t.P, Jw, J);
	}
	if (first) {
		UNROLL for (int i = 0; i < 6 * N; i++) Jp[i] = J[i];
	} else {
		mm<6, N, N>(J, Nprec, Jp);
	}
	// ---- branch decision (SingularityHandler.cpp:83-143). First an SVD-free certificate that the whole
	// wavefront is in the "fully non-singular" branch (then U_ns spans range(P) and only the projector
	// P is needed); the introspection build always runs the SVD because it reports singular values.
	const int rank = t.rank;
	bool certified = false;
	if (!DEBUG) {
		real G[36], Pc[36];
		mm_nt_sym<6, N>(Jp, Jp, G);
		UNROLL for (int i = 0; i < 6; i++) UNROLL for (int j = 0; j < 6; j++) Pc[i * 6 + j] = ((i == j) ? 1.0 : 0.0) - t.P[i * 6 + j];
		certified = __all(certify_gram<6>(G, t.full_projection ? nullptr : Pc, t.s_abs_tol * t.s_abs_tol, t.s_max * t.s_max));
	}
	real Q[N * 6], W[36], sv[6], ss[6], Pns[36], Ps[36], alpha = 1;
	int pos[6], split = rank;
	if (certified) {
		UNROLL for (int i = 0; i < 36; i++) {
			Pns[i] = t.P[i];
			Ps[i] = 0;
			W[i] = 0;
		}
		UNROLL for (int i = 0; i < N * 6; i++) Q[i] = 0;
		UNROLL for (int i = 0; i < 6; i++) {
			sv[i] = ss[i] = 0;
			pos[i] = i;
		}
		if (chain.ok) {	 // row space of this task for a later full JointTask
			real rows[N * N];
			UNROLL for (int i = 0; i < N * N; i++) rows[i] = 0;
			mm_tn<6, 6, N>(t.PU, Jp, rows);	 // rows >= rank are zero (PU has `rank` non-zero columns)
			chain_append(chain, rows, rank);
		}
	} else {
		chain.ok = false;
		// ---- thin SVD of Jp via one-sided Jacobi on Jp^T (7x6): Jp^T W = Q, U = W, V = Q / s
		UNROLL for (int i = 0; i < 6; i++) UNROLL for (int j = 0; j < N; j++) Q[j * 6 + i] = Jp[i * N + j];
		hestenes<N, 6>(Q, W);
		UNROLL for (int j = 0; j < 6; j++) {
			real a = 0;
			UNROLL for (int r = 0; r < N; r++) a = fma(Q[r * 6 + j], Q[r * 6 + j], a);
			sv[j] = sqrt(a);
		}
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
		// ---- range split (SingularityHandler.cpp:83-143)
		if (ss[0] < t.s_abs_tol) {
			split = 0;
			alpha = 0;
		} else {
			split = rank;
			alpha = 1;
			bool found = false;
			UNROLL for (int i = 1; i < 6; i++) {
				real icn = ss[i] / ss[0];
				if (i < rank && !found && icn < t.s_max) {
					alpha = fmin(fmax((icn - t.s_min) / (t.s_max - t.s_min), 0.0), 1.0);
					split = i;
					found = true;
				}
			}
		}
		UNROLL for (int i = 0; i < 6; i++) UNROLL for (int k = 0; k <= i; k++) {
			real a = 0, c = 0;
			UNROLL for (int j = 0; j < 6; j++) {
				real uu = W[i * 6 + j] * W[k * 6 + j];
				a += (pos[j] < split) ? uu : 0.0;
				c += (pos[j] >= split && pos[j] < rank) ? uu : 0.0;
			}
			Pns[i * 6 + k] = Pns[k * 6 + i] = a;
			Ps[i * 6 + k] = Ps[k * 6 + i] = c;
		}
	}
	const int sc = rank - split;
	const bool bie = t.decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES;
	const bool impedance = t.decoupling == SAI2B_IMPEDANCE;
	// ---- non-singular part: Lambda_ns (embedded), N_ns (SingularityHandler.cpp:110-114,130-134)
	real A[36], AB[36], Lns[36], LnsMod[36], Nns[N * N];
	sandwich6(Jp, rc.Minv, A);
	if (bie) sandwich6(Jp, rc.MinvB, AB);
	pinv_proj<6>(A, Pns, Lns);
	if (bie)
		pinv_proj<6>(AB, Pns, LnsMod);
	else {
		UNROLL for (int i = 0; i < 36; i++) LnsMod[i] = impedance ? Pns[i] : Lns[i];
	}
	{
		real T1[6 * N], T2[6 * N];
		mm<6, N, N>(Jp, rc.Minv, T1);  // (Minv Jp^T)^T
		mm<6, 6, N>(Lns, Jp, T2);
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j < N; j++) {
			real s = (i == j) ? 1.0 : 0.0;
			UNROLL for (int l = 0; l < 6; l++) s = fma(-T1[l * N + i], T2[l * N + j], s);
			Nns[i * N + j] = s;
		}
	}
	// ---- control law
	real Fu[6], Ff[6];
	{
		MftIn in;
		mft_load(t, B, b, in);
		mft_law(t, rc, J, x, R, in, Fu, Ff, B, b, do_torque);
		if (do_torque) mft_store_integrators(t, B, b, in);
	}
	real tau[N];
	{
		real a6[6], b6[6];
		mv<6, 6>(LnsMod, Fu, a6);
		mv<6, 6>(Pns, Ff, b6);
		UNROLL for (int i = 0; i < 6; i++) a6[i] += b6[i];
		mv_t<6, N>(Jp, a6, tau);  // SingularityHandler.cpp:307-309
	}
	// ---- singularity bookkeeping (SingularityHandler.cpp:230-295) and blended torques (:313-367)
	int* IS = t.istate;
	real* S = t.state;
	const int prev_types = ldi(IS, IS_NTYPES, B, b);
	real Ntask[N * N];
	UNROLL for (int i = 0; i < N * N; i++) Ntask[i] = Nns[i];
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
		// entering conditions (:233-236); lazily kept only while singular — equivalent to the
		// reference's every-tick refresh because the first singular tick always overwrites them
		real qprior[N], t2dir[N];
		if (commit_sh && (prev_types == 0 || c2 > c1)) {
			UNROLL for (int i = 0; i < N; i++) {
				st(S, MFT_QPRIOR + i, B, b, rc.q[i]);
				st(S, MFT_DQPRIOR + i, B, b, rc.dq[i]);
				qprior[i] = rc.q[i];
			}
		} else {
			UNROLL for (int i = 0; i < N; i++) qprior[i] = ld(S, MFT_QPRIOR + i, B, b);
		}
		// first singular column (descending order) and the singular joint-space range
		real us0[6], vs0[N], PV[N * N];
		UNROLL for (int i = 0; i < N * N; i++) PV[i] = 0;
		UNROLL for (int i = 0; i < 6; i++) us0[i] = 0;
		UNROLL for (int i = 0; i < N; i++) vs0[i] = 0;
		bool any1 = false;
#pragma unroll 1
		for (int p = split; p < rank; p++) {
			real u[6], v[N], s = 0;
			UNROLL for (int i = 0; i < 6; i++) u[i] = 0;
			UNROLL for (int i = 0; i < N; i++) v[i] = 0;
			UNROLL for (int j = 0; j < 6; j++)
				if (pos[j] == p) {
					s = sv[j];
					UNROLL for (int i = 0; i < 6; i++) u[i] = W[i * 6 + j];
					UNROLL for (int i = 0; i < N; i++) v[i] = Q[i * 6 + j];
				}
			real inv = s > 0 ? 1.0 / s : 0.0;
			{  // sign convention shared with the oracle: largest-magnitude component of v positive
				real big = 0, bigabs = -1;
				UNROLL for (int i = 0; i < N; i++) {
					const bool take = fabs(v[i]) > bigabs;
					bigabs = take ? fabs(v[i]) : bigabs;
					big = take ? v[i] : big;
				}
				if (big < 0) {
					inv = -inv;
					UNROLL for (int i = 0; i < 6; i++) u[i] = -u[i];
				}
			}
			UNROLL for (int i = 0; i < N; i++) v[i] *= inv;
			if (p == split) {
				UNROLL for (int i = 0; i < 6; i++) us0[i] = u[i];
				UNROLL for (int i = 0; i < N; i++) vs0[i] = v[i];
			}
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int k = 0; k < N; k++) PV[i * N + k] = fma(v[i], v[k], PV[i * N + k]);
			// classification by FK perturbation (:253-273) along +v, -v or both (enum sai2b_singular_vector_sign: the
			// reference perturbs along V_s[:, i] as Eigen left it, a sign it does not specify)
			bool moved[2] = {false, false};
			const int pass0 = t.sv_sign == SAI2B_SV_SIGN_V_MAX_NEGATIVE ? 1 : 0;
			const int pass1 = t.sv_sign == SAI2B_SV_SIGN_V_MAX_POSITIVE ? 0 : 1;
#pragma unroll 1
			for (int pass = pass0; pass <= pass1; pass++) {
				const real step = pass ? -t.perturb : t.perturb;
				real qp[N], x1[3], R1[9], d[6];
				UNROLL for (int i = 0; i < N; i++) qp[i] = fma(step, v[i], rc.q[i]);
				Frames F1;
				fk(P.model, qp, F1);
				frame_pose(t, F1, x1, R1);
				UNROLL for (int k = 0; k < 3; k++) d[k] = x1[k] - x[k];
				orientation_error(R1, R, d + 3);
				real m = 0;
				UNROLL for (int k = 0; k < 6; k++) m = fma(d[k], u[k], m);
				if (pass)
					moved[1] = fabs(m) > t.type_1_tol;
				else
					moved[0] = fabs(m) > t.type_1_tol;
			}
			const bool type1 = t.sv_sign == SAI2B_SV_SIGN_BOTH ? (moved[0] && moved[1]) : (moved[0] || moved[1]);
			any1 = any1 || type1;
		}
		if (commit_sh) {  // history ring (:276-293)
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
			UNROLL for (int i = 0; i < N; i++) tau[i] = 0;
			UNROLL for (int i = 0; i < N * N; i++) Ntask[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;  // N_total = N_prec
		} else if (impedance) {
			// :310-312 — tau already holds J_ns^T (U_ns^T Fu + U_ns^T Ff); N = posture nullspace below
		}
		if (split != 0 && t.enforce) {
			// posture task in the singular joint directions (:152-157)
			real Bm[N * N];
			if (first) {
				UNROLL for (int i = 0; i < N * N; i++) Bm[i] = Nns[i];
			} else {
				mm<N, N, N>(Nns, Nprec, Bm);
			}
			real C[N * N], Lj[N * N], LjMod[N * N];
			sandwich7(Bm, rc.Minv, C);
			pinv_proj<N>(C, PV, Lj);
			{
				real T1[N * N], T2[N * N], Np[N * N];
				mm<N, N, N>(Bm, rc.Minv, T1);
				mm<N, N, N>(Lj, Bm, T2);
				UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j < N; j++) {
					real s = (i == j) ? 1.0 : 0.0;
					UNROLL for (int l = 0; l < N; l++) s = fma(-T1[l * N + i], T2[l * N + j], s);
					Np[i * N + j] = s;
				}
				mm<N, N, N>(Np, Nns, Ntask);
			}
			if (!impedance) {
				if (bie) {
					sandwich7(Bm, rc.MinvB, C);
					pinv_proj<N>(C, PV, LjMod);
				} else {
					UNROLL for (int i = 0; i < N * N; i++) LjMod[i] = Lj[i];
				}
				real LsMod[36];
				if (bie)
					pinv_proj<6>(AB, Ps, LsMod);
				else
					pinv_proj<6>(A, Ps, LsMod);
				// joint strategy (:327-351)
				real ut[N], y7[N], tau_j[N];
				if (c1 > c2 || t.enforce_t1) {
					UNROLL for (int i = 0; i < N; i++) ut[i] = -t.kp1 * (rc.q[i] - qprior[i]) - t.kv1 * rc.dq[i];
					mv<N, N>(LjMod, ut, y7);
					mv_t<N, N>(Bm, y7, tau_j);
				} else {
					UNROLL for (int i = 0; i < N; i++) {
						real dir = ld(S, MFT_T2DIR + i, B, b);
						if (vs0[i] != 0) {
							if (fabs(rc.q[i] - P.model.q_upper[i]) < t.t2_angle) {
								dir = -1;
								if (do_torque) st(S, MFT_T2DIR + i, B, b, dir);
							} else if (fabs(rc.q[i] - P.model.q_lower[i]) < t.t2_angle) {
								dir = 1;
								if (do_torque) st(S, MFT_T2DIR + i, B, b, dir);
							}
						}
						t2dir[i] = dir;
					}
					real Fs[6], nrm = 0, fTd = 0;
					UNROLL for (int i = 0; i < 6; i++) {
						Fs[i] = Fu[i] + Ff[i];
						nrm = fma(Fs[i], Fs[i], nrm);
					}
					nrm = sqrt(nrm);
					UNROLL for (int i = 0; i < 6; i++) fTd = fma(nrm > 0 ? Fs[i] / nrm : Fs[i], us0[i], fTd);
					UNROLL for (int i = 0; i < N; i++) ut[i] = t2dir[i] * fabs(fTd) * t.t2_ratio * P.model.effort[i];
					real a7[N], b7[N], c7[N];
					mv<N, N>(PV, ut, a7);
					UNROLL for (int i = 0; i < N; i++) ut[i] = -t.kv2 * rc.dq[i];
					mv<N, N>(LjMod, ut, b7);
					UNROLL for (int i = 0; i < N; i++) c7[i] = a7[i] + b7[i];
					mv_t<N, N>(Bm, c7, tau_j);
				}
				// singular-direction torques, sanitised and clamped (:354-365)
				real a6[6], b6[6], tau_s[N];
				mv<6, 6>(LsMod, Fu, a6);
				mv<6, 6>(Ps, Ff, b6);
				UNROLL for (int i = 0; i < 6; i++) a6[i] += b6[i];
				mv_t<6, N>(Jp, a6, tau_s);
				UNROLL for (int i = 0; i < N; i++) {
					real v = tau_s[i];
					v = (v != v) ? 0.0 : fmin(fmax(v, -P.model.effort[i]), P.model.effort[i]);
					tau[i] = tau[i] + alpha * v + (1 - alpha) * tau_j[i];  // :366
				}
			}
		}
	}
	UNROLL for (int i = 0; i < N; i++) tau_total[i] += tau[i];
	if (DEBUG) {
		if (t.dbg_tau) {
			UNROLL for (int i = 0; i < N; i++) st(t.dbg_tau, i, B, b, tau[i]);
		}
		if (t.dbg_sigma) {
			UNROLL for (int i = 0; i < 6; i++) st(t.dbg_sigma, i, B, b, ss[i]);
			st(t.dbg_sigma, 6, B, b, alpha);
			st(t.dbg_sigma, 7, B, b, (real)split);
		}
		if (t.dbg_J) {
			UNROLL for (int i = 0; i < 6 * N; i++) st(t.dbg_J, i, B, b, Jw[i]);
		}
		if (t.dbg_F) {
			UNROLL for (int i = 0; i < 6; i++) {
				st(t.dbg_F, i, B, b, Fu[i]);
				st(t.dbg_F, 6 + i, B, b, Ff[i]);
			}
		}
		if (t.dbg_pose) {
			UNROLL for (int i = 0; i < 3; i++) st(t.dbg_pose, i, B, b, x[i]);
			UNROLL for (int i = 0; i < 9; i++) st(t.dbg_pose, 3 + i, B, b, R[i]);
		}
	}
	if (Ntask_out) {  // TemplateTask::getTaskNullspace (MotionForceTask.h:193), task-level calls only
		UNROLL for (int i = 0; i < N * N; i++) st(Ntask_out, i, B, b, Ntask[i]);
	}
	// N_prec <- N N_prec (RobotController.cpp:58, MotionForceTask.h:207-209)
	if (!last || DEBUG) {
		if (first) {
			UNROLL for (int i = 0; i < N * N; i++) Nprec[i] = Ntask[i];
		} else {
			real T[N * N];
			mm<N, N, N>(Ntask, Nprec, T);
			UNROLL for (int i = 0; i < N * N; i++) Nprec[i] = T[i];
		}
		if (DEBUG && t.dbg_N) {
			UNROLL for (int i = 0; i < N * N; i++) st(t.dbg_N, i, B, b, Nprec[i]);
		}
	}
}

// JointTask::updateTaskModel + computeTorques(tau_prec) for one robot (JointTask.cpp:218-356)
template <bool DEBUG, bool RANGE_ONLY = false>
DI void jt_task(const DevParams& P, const DevTask& t, const RobotCtx& rc, int B, int b, bool first, bool last,
				bool with_comp, bool do_torque, real* Nprec, real* tau_total, Chain& chain, real* Ntask_out = nullptr) {
	real Jp[N * N];
	if (first) {
		UNROLL for (int i = 0; i < N * N; i++) Jp[i] = t.S[i];
	} else if (t.full_selection) {
		UNROLL for (int i = 0; i < N * N; i++) Jp[i] = Nprec[i];
	} else {
		mm<N, N, N>(t.S, Nprec, Jp);
	}
	// range projector of Jp (Sai2Model::matrixRangeBasis, tolerance 1e-3: SURVEY App. D)
	real PR[N * N];
	bool zero_range = false;
	bool need_svd = false;
	if (first) {
		// Jp = S, full row rank by construction of the task (JointTask.cpp:34-39): R = I_k0
		UNROLL for (int i = 0; i < N * N; i++) PR[i] = 0;
		UNROLL for (int i = 0; i < N; i++) PR[i * N + i] = (i < t.k0) ? 1.0 : 0.0;
		if (chain.ok) chain_append(chain, Jp, t.k0);
	} else if (!DEBUG && t.full_selection && chain.ok) {
		// Jp = N_prec behind certified tasks: range(N_prec) = null(W), no SVD needed
		zero_range = chain.wrows >= N;
		real WWt[N * N], X[N * N], T[N * N];
		mm_nt_sym<N, N>(chain.W, chain.W, WWt);
		UNROLL for (int i = 0; i < N; i++) WWt[i * N + i] += (i < chain.wrows) ? 0.0 : 1.0;
		spd_inverse<N>(WWt, X);
		mm<N, N, N>(X, chain.W, T);
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int k = 0; k <= i; k++) {
			real a = (i == k) ? 1.0 : 0.0;
			UNROLL for (int l = 0; l < N; l++) a = fma(-chain.W[l * N + i], T[l * N + k], a);
			PR[i * N + k] = PR[k * N + i] = a;
		}
		chain.wrows = N;  // a full JointTask closes the hierarchy
	} else if (!DEBUG && !t.full_selection) {
		// partial task: certified full row rank (all k0 singular values above the 1e-3 rule) -> R = I_k0
		real C0[N * N], Pc[N * N];
		mm_nt_sym<N, N>(Jp, Jp, C0);
		UNROLL for (int i = 0; i < N * N; i++) Pc[i] = 0;
		UNROLL for (int i = 0; i < N; i++) Pc[i * N + i] = (i < t.k0) ? 0.0 : 1.0;
		if (__all(certify_gram<N>(C0, Pc, 1e-6, 1e-6))) {
			UNROLL for (int i = 0; i < N * N; i++) PR[i] = 0;
			UNROLL for (int i = 0; i < N; i++) PR[i * N + i] = (i < t.k0) ? 1.0 : 0.0;
			if (chain.ok) chain_append(chain, Jp, t.k0);
		} else {
			need_svd = true;
		}
	} else {
		need_svd = true;
	}
	if (need_svd) {
		chain.ok = false;
		real X[N * N], W[N * N], sv[N];
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j < N; j++) X[j * N + i] = Jp[i * N + j];
		hestenes<N, N>(X, W);
		real s0 = 0;
		UNROLL for (int j = 0; j < N; j++) {
			real a = 0;
			UNROLL for (int r = 0; r < N; r++) a = fma(X[r * N + j], X[r * N + j], a);
			sv[j] = sqrt(a);
			s0 = fmax(s0, sv[j]);
		}
		zero_range = s0 < 1e-3;
		int dof = 0;
		UNROLL for (int j = 0; j < N; j++) dof += (sv[j] / s0 >= 1e-3) ? 1 : 0;
		UNROLL for (int i = 0; i < N; i++) UNROLL for (int k = 0; k <= i; k++) {
			real a = 0;
			UNROLL for (int j = 0; j < N; j++) a += (sv[j] / s0 >= 1e-3) ? W[i * N + j] * W[k * N + j] : 0.0;
			if (dof == t.k0) a = (i == k && i < t.k0) ? 1.0 : 0.0;	// full row rank -> identity basis
			PR[i * N + k] = PR[k * N + i] = a;
		}
	}
	if (t.otg_gated && !do_torque) st(t.otg_state, OTG_ACTIVE, B, b, zero_range ? 0.0 : 1.0);  // read by otg_kernel
	if constexpr (RANGE_ONLY) return;  // last gated task of the range pass: its inertias and nullspace are not needed
	real tau[N];
	UNROLL for (int i = 0; i < N; i++) tau[i] = 0;
	real Ntask[N * N];
	UNROLL for (int i = 0; i < N * N; i++) Ntask[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
	real* S = t.state;
	// controller state and PD(+I) law (JointTask.cpp:299-345); integrator advances every call
	real cur[N], vel[N], f[N], ddq_d[N];
	if (t.full_selection) {
		UNROLL for (int i = 0; i < N; i++) {
			cur[i] = rc.q[i];
			vel[i] = rc.dq[i];
		}
	} else {
		mv<N, N>(t.S, rc.q, cur);
		mv<N, N>(t.S, rc.dq, vel);
	}
	if (!zero_range) {
		const real* G = t.law_goals;  // goal, or the internal OTG's next state (JointTask.cpp:308-320)
		UNROLL for (int i = 0; i < N; i++) {
			f[i] = 0;
			ddq_d[i] = 0;
			if (i < t.k0) {
				real qd = ld(G, i, B, b), dqd = ld(G, t.k0 + i, B, b);
				ddq_d[i] = ld(G, 2 * t.k0 + i, B, b);
				real integ = fma(cur[i] - qd, t.dt, ld(S, i, B, b));
				if (do_torque) st(S, i, B, b, integ);
				if (t.use_vsat) {
					real kvi = gain_pinv(t.kv[i]);
					real dv = -t.kp[i] * kvi * (cur[i] - qd) - t.ki[i] * kvi * integ;
					dv = fmin(fmax(dv, -t.vsat[i]), t.vsat[i]);
					f[i] = -t.kv[i] * (vel[i] - dv);
				} else {
					f[i] = -t.kp[i] * (cur[i] - qd) - t.kv[i] * (vel[i] - dqd) - t.ki[i] * integ;
				}
			}
		}
		real C[N * N], L[N * N], LMod[N * N];
		if (first && t.full_selection) {
			// Jp = I: Lambda = M (A-KA 1); keep the generic formula (inverse of M^-1) for parity
		}
		sandwich7(Jp, rc.Minv, C);
		pinv_proj<N>(C, PR, L);
		if (t.decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
			sandwich7(Jp, rc.MinvB, C);
			pinv_proj<N>(C, PR, LMod);
		} else {
			UNROLL for (int i = 0; i < N * N; i++) LMod[i] = (t.decoupling == SAI2B_IMPEDANCE) ? PR[i] : L[i];
		}
		// x = M_partial R^T ddq_d + M_partial_mod R^T f (JointTask.cpp:348-351), embedded
		real xa[N], xb[N];
		mv<N, N>(L, ddq_d, xa);
		mv<N, N>(LMod, f, xb);
		if (with_comp && !first) {	// JointTask.cpp:285-292
			real a7[N], b7[N], c7[N];
			mv<N, N>(rc.Minv, tau_total, a7);
			if (t.full_selection) {
				UNROLL for (int i = 0; i < N; i++) b7[i] = a7[i];
			} else {
				mv<N, N>(t.S, a7, b7);
			}
			mv<N, N>(L, b7, c7);
			UNROLL for (int i = 0; i < N; i++) xa[i] -= c7[i];
		}
		UNROLL for (int i = 0; i < N; i++) xa[i] += xb[i];
		mv_t<N, N>(Jp, xa, tau);
		if (!last || DEBUG) {
			real T1[N * N], T2[N * N];
			mm<N, N, N>(Jp, rc.Minv, T1);
			mm<N, N, N>(L, Jp, T2);
			UNROLL for (int i = 0; i < N; i++) UNROLL for (int j = 0; j < N; j++) {
				real s = (i == j) ? 1.0 : 0.0;
				UNROLL for (int l = 0; l < N; l++) s = fma(-T1[l * N + i], T2[l * N + j], s);
				Ntask[i * N + j] = s;
			}
		}
	}
	UNROLL for (int i = 0; i < N; i++) tau_total[i] += tau[i];
	if (DEBUG && t.dbg_tau) {
		UNROLL for (int i = 0; i < N; i++) st(t.dbg_tau, i, B, b, tau[i]);
	}
	if (Ntask_out) {  // TemplateTask::getTaskNullspace (JointTask.h:207), task-level calls only
		UNROLL for (int i = 0; i < N * N; i++) st(Ntask_out, i, B, b, Ntask[i]);
	}
	if (!last || DEBUG) {
		if (first) {
			UNROLL for (int i = 0; i < N * N; i++) Nprec[i] = Ntask[i];
		} else {
			real T[N * N];
			mm<N, N, N>(Ntask, Nprec, T);
			UNROLL for (int i = 0; i < N * N; i++) Nprec[i] = T[i];
		}
		if (DEBUG && t.dbg_N) {
			UNROLL for (int i = 0; i < N * N; i++) st(t.dbg_N, i, B, b, Nprec[i]);
		}
	}
}

}  // namespace sai2b
