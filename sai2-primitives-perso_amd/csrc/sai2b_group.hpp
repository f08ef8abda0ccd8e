// sai2b_group.hpp — primitives of the LANES-PER-ROBOT kernels (sai2b_group_tick.hpp).
//
// A robot is spread over a group of G lanes (G = 16: one DPP row, lowest latency; G = 8: two robots per
// row, fewer idle lanes). Matrices are distributed BY ROWS: lane r of the group holds row r (n <= 8 doubles in
// registers); lanes >= n idle along (they compute on zeros, never branch away: a DPP read of a lane that
// is masked off returns nothing). The work-horse is `acc += bcast_L(src) * a` in ONE instruction
// (v_fmac_f64_dpp row_newbcast:L, the one DPP control the FP64 pipe has; measured at the rate of a plain
// v_fmac_f64, scripts/micro/dpp_f64.hip), from which the three products below are built:
//     mm_rr : C = A B      rows of A local, rows of B in the lanes       c[j] += a[l] * bcast_l(b[j])
//     mm_rt : C = A B^T    rows of A local, rows of B in the lanes       c[j] += a[l] * bcast_j(b[l])
//     mv    : y = A x      rows of A local, x spread over the lanes      y    += a[l] * bcast_l(x)
// A^T B cannot be formed that way (it would need lane-dependent register indices), so a matrix is
// transposed through a small per-group LDS pad when its columns are needed (transpose_lds).
#pragma once
#include <hip/hip_runtime.h>

#include "sai2b_device.hpp"
#include "sai2b_dpp_blocks.h"

namespace sai2b {
namespace grp {

// ---------------------------------------------------------------- lane moves
template <int G>
DI int lane() {
	return threadIdx.x & (G - 1);
}
template <int G>
DI int group() {
	return threadIdx.x / G;
}
DI real from_halves(int lo, int hi) { return __hiloint2double(hi, lo); }

// value of lane L of the group, in every lane (compiler-visible DPP moves: hazards handled by hipcc)
template <int G, int L>
DI real bcast(real x) {
	const long long v = __double_as_longlong(x);
	if constexpr (G == 16) {
		return __longlong_as_double(__builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, false));
	} else {
		// two full-row broadcasts and a select: bank-masked DP-DPP writes must not be followed at once by a reader
		// of the same register (tools/gen_dpp_blocks.py), which the compiler does not know
		const long long lo = __builtin_amdgcn_update_dpp(v, v, 0x150 + L, 0xf, 0xf, false);
		const long long hi = __builtin_amdgcn_update_dpp(v, v, 0x150 + L + 8, 0xf, 0xf, false);
		return __longlong_as_double((threadIdx.x & 8) ? hi : lo);
	}
}
// value of lane `src` (0 <= src < G, any run-time value, may differ per lane) of the group
template <int G>
DI real gather(real x, int src) {
	const int base = (threadIdx.x & 63) & ~(G - 1);
	return __shfl(x, base + src, 64);
}
// lane r receives the value of lane r - K of its group; lanes r < K receive `fill`
template <int G, int K>
DI real shift_up(real x, real fill) {
	int lo = __double2loint(x), hi = __double2hiint(x);
	const int flo = __double2loint(fill), fhi = __double2hiint(fill);
	lo = __builtin_amdgcn_update_dpp(flo, lo, 0x110 + K, 0xf, 0xf, false);	// row_shr:K, invalid source -> keeps `old`
	hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x110 + K, 0xf, 0xf, false);
	real v = from_halves(lo, hi);
	if constexpr (G == 8) v = (lane<G>() < K) ? fill : v;  // lanes 8..8+K-1 of the row read the other robot
	return v;
}
// lane r receives the value of lane r + K of its group; lanes r + K >= G receive `fill`
template <int G, int K>
DI real shift_down(real x, real fill) {
	int lo = __double2loint(x), hi = __double2hiint(x);
	const int flo = __double2loint(fill), fhi = __double2hiint(fill);
	lo = __builtin_amdgcn_update_dpp(flo, lo, 0x100 + K, 0xf, 0xf, false);	// row_shl:K
	hi = __builtin_amdgcn_update_dpp(fhi, hi, 0x100 + K, 0xf, 0xf, false);
	real v = from_halves(lo, hi);
	if constexpr (G == 8) v = (lane<G>() + K >= G) ? fill : v;
	return v;
}

// ---------------------------------------------------------------- fused broadcast-FMA blocks
template <int G, int L, int K>
DI void rowfma(real* c, const real* b, real a) {
	if constexpr (G == 16)
		rowfma16<L, K>(c, b, a);
	else
		rowfma8<L, K>(c, b, a);
}
template <int G, int K>
DI void lanefma(real* c, real b, real a) {
	if constexpr (G == 16)
		lanefma16<K>(c, b, a);
	else
		lanefma8<K>(c, b, a);
}
template <int G, int K>
DI void mvfma(real& acc, real x, const real* a) {
	if constexpr (G == 16)
		mvfma16<K>(acc, x, a);
	else
		mvfma8<K>(acc, x, a);
}

template <int G, int L, int K>
DI void selffma(real* c, real a) {
	if constexpr (G == 16)
		selffma16<L, K>(c, a);
	else
		selffma8<L, K>(c, a);
}

// c[0..NC) (+)= sum_{l < KB} a[l] * (row l of B)[0..NC),  B's row l = b[] of lane l.   C = A B
template <int G, int l, int KB, int NC>
DI void mm_rr_step(const real* a, const real* b, real* c) {
	if constexpr (l < KB) {
		rowfma<G, l, NC>(c, b, a[l]);
		mm_rr_step<G, l + 1, KB, NC>(a, b, c);
	}
}
template <int G, int KB, int NC, bool ACC = false>
DI void mm_rr(const real* a, const real* b, real* c) {
	if (!ACC) {
		UNROLL for (int j = 0; j < NC; j++) c[j] = 0;
	}
	mm_rr_step<G, 0, KB, NC>(a, b, c);
}
// c[0..NR) (+)= sum_{l < K} a[l] * (row j of B)[l],  j < NR.   C = A B^T
template <int G, int K, int NR, bool ACC = false>
DI void mm_rt(const real* a, const real* b, real* c) {
	if (!ACC) {
		UNROLL for (int j = 0; j < NR; j++) c[j] = 0;
	}
	UNROLL for (int l = 0; l < K; l++) lanefma<G, NR>(c, b[l], a[l]);
}
// sum_{l < K} a[l] * x_l with x_l the value `x` of lane l.   y = A x
template <int G, int K>
DI real mv(const real* a, real x) {
	real s = 0;
	mvfma<G, K>(s, x, a);
	return s;
}
// sum over lanes 0..K-1 of x, in every lane
template <int G, int K>
DI real allsum(real x) {
	real ones[K];
	UNROLL for (int l = 0; l < K; l++) ones[l] = 1.0;
	real s = 0;
	mvfma<G, K>(s, x, ones);
	return s;
}

// ---------------------------------------------------------------- transposition through LDS
// pad: this group's scratch of at least NR * (NC | 1) doubles. in: row r of an NR x NC matrix in lane r (< NR);
// out: row r of its transpose (NC x NR) in lane r (< NC); lanes beyond get zeros.
template <int G, int NR, int NC>
DI void transpose_lds(real* pad, const real* in, real* out) {
	constexpr int LD = NC | 1;	// odd leading dimension: column reads of the 8-byte words do not collide in the banks
	const int r = lane<G>();
	if (r < NR) {
		UNROLL for (int j = 0; j < NC; j++) pad[r * LD + j] = in[j];
	}
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
	__builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
	UNROLL for (int i = 0; i < NR; i++) out[i] = (r < NC) ? pad[i * LD + r] : 0.0;
	__builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
	__builtin_amdgcn_wave_barrier();
}

// ---------------------------------------------------------------- small scalar helpers (replicated per lane)
// 1 / x to double precision: hardware estimate + two Newton steps (the 15-instruction IEEE division is not needed:
// every use is a pivot or a norm that is far from the ends of the exponent range)
DI real recip(real x) {
	real r = __builtin_amdgcn_rcp(x);
	r = fma(fma(-x, r, 1.0), r, r);
	r = fma(fma(-x, r, 1.0), r, r);
	return r;
}
// 1 / sqrt(x), x > 0
DI real rsqrt_nr(real x) {
	real r = __builtin_amdgcn_rsq(x);
	const real h = 0.5 * x;
	r = fma(fma(-h * r, r, 0.5), r, r);
	r = fma(fma(-h * r, r, 0.5), r, r);
	return r;
}
DI real sqrt_nr(real x) {
	if (!(x > 0)) return 0.0;
	const real r = rsqrt_nr(x);
	real s = x * r;
	s = fma(fma(-s, s, x), 0.5 * r, s);	 // one correction of the product
	return s;
}

// ---------------------------------------------------------------- in-place inverse of an SPD matrix, rows in lanes
// Gauss-Jordan without pivoting (stable for SPD: it is the LDL^T elimination order). a[0..n): row r of A in lane
// r < n; lanes >= n must hold zero rows and come out as zero rows. Stands in for Eigen's .inverse() on the SPD
// matrices of the path (SingularityHandler.cpp:120,182,190,201,212, JointTask.cpp:260-265, sai2-model M^-1).
template <int G, int n, int k>
DI void spd_inverse_step(real* a, int r, int kmax) {
	if constexpr (k < n) {
		if (k < kmax) {	 // uniform over the wavefront
		const real d = recip(bcast<G, k>(a[k]));
		const bool me = (r == k);
		// the pivot row is scaled in its own lane; the others subtract f times it
		const real sc = me ? d : 1.0;
		const real nf = me ? 0.0 : -a[k];
		UNROLL for (int j = 0; j < n; j++) a[j] *= sc;
		selffma<G, k, n>(a, nf);  // a[j] += bcast_k(a[j]) * nf   (column k comes out as 0 in the other rows)
		a[k] = me ? d : nf * d;	  // ... and is replaced by the column of the inverse being built
		}
		spd_inverse_step<G, n, k + 1>(a, r, kmax);
	}
}
// kmax < n (the same for every robot of the launch): rows and columns >= kmax of A are those of the identity (A is
// block diagonal), whose elimination steps change nothing and are skipped
template <int G, int n>
DI void spd_inverse_rows(real* a, int kmax = n) {
	spd_inverse_step<G, n, 0>(a, lane<G>(), kmax);
}

}  // namespace grp
}  // namespace sai2b
