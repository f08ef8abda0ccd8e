/*
 * sai2b_detmath.h — bit-reproducible sine / cosine / arctangent for the internal OTG's Cartesian wrapper.
 *
 * Why: OTG_6dof_cartesian (reference src/helper_modules/OTG_6dof_cartesian.cpp:138-239) turns rotations into
 * rotation vectors and back (Eigen AngleAxisd: atan2, sin, cos) and hands the result to ruckig, whose
 * phase-synchronisation test (ruckig/include/ruckig/calculator_target.hpp:123-200) compares quantities that are
 * collinear up to rounding against 4 * DBL_EPSILON. Two correct libm's (glibc on the host, OCML on the GPU) differ
 * in the last bit of a sine often enough to send the planner down the other — equally valid — branch. These
 * versions use only IEEE + - * / and comparisons, no fused multiply-add, no table lookups that depend on a
 * library: compiled with -ffp-contract=off they return identical bits on the host and on the GPU. Accuracy is
 * that of the classic fdlibm kernels they restate (< 1 ulp for sin / cos on the reduced range, < 2 ulp overall
 * for |x| up to ~1e5; < 1 ulp for atan).
 *
 * Used by sai2-primitives-perso_amd/csrc/sai2b_otg_core.hpp (device and host) and by oracle/otg_oracle.c.
 */
#ifndef SAI2B_DETMATH_H_
#define SAI2B_DETMATH_H_

#include <math.h>

#if defined(__HIPCC__)
#define SAI2B_DET_FN __host__ __device__ static inline
#else
#define SAI2B_DET_FN static inline
#endif
/* clang (hipcc): no contraction inside these functions whatever the translation unit's setting; gcc builds (the
 * oracle) pass -ffp-contract=off */
#if defined(__clang__)
#define SAI2B_DET_PRAGMA _Pragma("clang fp contract(off)")
#else
#define SAI2B_DET_PRAGMA
#endif

/* sin and cos of x (|x| < ~1e5). Argument reduction by pi/2 in three exactly representable pieces (fdlibm's
 * pio2_1 .. pio2_3: k * piece is exact for |k| < 2^20), then the minimax kernels on [-pi/4, pi/4]. */
SAI2B_DET_FN void sai2b_det_sincos(double x, double* sn, double* cs) {
	SAI2B_DET_PRAGMA
	const double invpio2 = 6.36619772367581382433e-01;
	const double pio2_1 = 1.57079632673412561417e+00, pio2_2 = 6.07710050630396597660e-11, pio2_3 = 2.02226624871116645580e-21;
	const double pio2_3t = 8.47842766036889956997e-32;
	const double kf = floor(x * invpio2 + 0.5); /* floor is exact on both sides */
	double r = x - kf * pio2_1;
	r = r - kf * pio2_2;
	r = r - kf * pio2_3;
	r = r - kf * pio2_3t;
	const double z = r * r;
	/* __kernel_sin: r + r z (S1 + z (S2 + ... z S6)) */
	const double ps = -1.66666666666666324348e-01 + z * (8.33333333332248946124e-03 + z * (-1.98412698298579493134e-04 +
					  z * (2.75573137070700676789e-06 + z * (-2.50507602534068634195e-08 + z * 1.58969099521155010221e-10))));
	const double s0 = r + (r * z) * ps;
	/* __kernel_cos: 1 - z/2 + z z (C1 + z (C2 + ... z C6)) */
	const double pc = 4.16666666666666019037e-02 + z * (-1.38888888888741095749e-03 + z * (2.48015872894767294178e-05 +
					  z * (-2.75573143513906633035e-07 + z * (2.08757232129817482790e-09 + z * -1.13596475577881948265e-11))));
	const double c0 = (1.0 - 0.5 * z) + (z * z) * pc;
	const long long k = (long long)kf;
	const double s1 = (k & 1) ? c0 : s0, c1 = (k & 1) ? s0 : c0;
	*sn = (k & 2) ? -s1 : s1;
	*cs = ((k + 1) & 2) ? -c1 : c1;
}

/* atan(t) for t >= 0: fdlibm's argument reduction at 7/16, 11/16, 19/16, 39/16 and its odd / even polynomial */
SAI2B_DET_FN double sai2b_det_atan_pos(double t) {
	SAI2B_DET_PRAGMA
	const double pio2_hi = 1.57079632679489655800e+00, pio2_lo = 6.12323399573676603587e-17;
	double x = t, hi = 0.0, lo = 0.0;
	int reduced = 1;
	if (t > 1e300) return pio2_hi + pio2_lo;
	if (x < 0.4375) {
		reduced = 0;
	} else if (x < 0.6875) {
		hi = 4.63647609000806093515e-01, lo = 2.26987774529616870924e-17; /* atan(0.5) */
		x = (2.0 * x - 1.0) / (2.0 + x);
	} else if (x < 1.1875) {
		hi = 7.85398163397448278999e-01, lo = 3.06161699786838301793e-17; /* atan(1) */
		x = (x - 1.0) / (x + 1.0);
	} else if (x < 2.4375) {
		hi = 9.82793723247329054082e-01, lo = 1.39033110312309984516e-17; /* atan(1.5) */
		x = (x - 1.5) / (1.0 + 1.5 * x);
	} else {
		hi = pio2_hi, lo = pio2_lo;
		x = -1.0 / x;
	}
	const double z = x * x, w = z * z;
	const double s1 = z * (3.33333333333329318027e-01 + w * (1.42857142725034663711e-01 + w * (9.09088713343650656196e-02 +
					  w * (6.66107313738753120669e-02 + w * (4.97687799461593236017e-02 + w * 1.62858201153657823623e-02)))));
	const double s2 = w * (-1.99999999998764832476e-01 + w * (-1.11111104054623557880e-01 + w * (-7.69187620504482999495e-02 +
					  w * (-5.83357013379057348645e-02 + w * -3.65315727442169155270e-02))));
	if (!reduced) return x - x * (s1 + s2);
	return hi - ((x * (s1 + s2) - lo) - x);
}

/* atan2(y, x) for y >= 0 and x >= 0 (the only quadrant the wrapper needs: atan2(|q_xyz|, |q_w|)) */
SAI2B_DET_FN double sai2b_det_atan2_pos(double y, double x) {
	SAI2B_DET_PRAGMA
	if (x == 0.0) return (y == 0.0) ? 0.0 : 1.57079632679489655800e+00 + 6.12323399573676603587e-17;
	return sai2b_det_atan_pos(y / x);
}

#endif /* SAI2B_DETMATH_H_ */
