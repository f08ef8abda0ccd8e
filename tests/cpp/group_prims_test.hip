// Unit test of the lanes-per-robot primitives (sai2-primitives-perso_amd/csrc/sai2b_group.hpp) on the GPU: every
// lane move and fused broadcast-FMA block for G = 16 and G = 8 against a host computation, one wavefront.
// Build: hipcc --offload-arch=gfx950 -O2 -I sai2-primitives-perso_amd/csrc tests/cpp/group_prims_test.hip -o group_prims_test
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <vector>

#include "sai2b_group.hpp"

using namespace sai2b;
using namespace sai2b::grp;

// out rows: 0 bcast<3>, 1 shift_up<2> fill -1, 2 shift_down<4> fill -2, 3..9 rowfma<2,7> (c[j]), 10..16 lanefma<7>, 17 mvfma<7>,
// 18..24 selffma<5,7>, 25..30 transpose (row r of the 6 x 7 -> 7 x 6), 31 gather(x, (r+3)%G), 32 allsum<7>, 33..39 spd inverse row
template <int G>
__global__ void prims(const double* in, double* out) {
	__shared__ double pads[64 / G][N * (N | 1)];
	const int t = threadIdx.x, r = lane<G>();
	double a[7], b[7], c[7];
	for (int j = 0; j < 7; j++) {
		a[j] = in[j * 64 + t];
		b[j] = in[(7 + j) * 64 + t];
		c[j] = in[(14 + j) * 64 + t];
	}
	out[0 * 64 + t] = bcast<G, 3>(a[0]);
	out[1 * 64 + t] = shift_up<G, 2>(a[0], -1.0);
	out[2 * 64 + t] = shift_down<G, 4>(a[0], -2.0);
	{
		double cc[7];
		for (int j = 0; j < 7; j++) cc[j] = c[j];
		rowfma<G, 2, 7>(cc, b, a[1]);
		for (int j = 0; j < 7; j++) out[(3 + j) * 64 + t] = cc[j];
	}
	{
		double cc[7];
		for (int j = 0; j < 7; j++) cc[j] = c[j];
		lanefma<G, 7>(cc, b[2], a[2]);
		for (int j = 0; j < 7; j++) out[(10 + j) * 64 + t] = cc[j];
	}
	{
		double acc = c[0];
		mvfma<G, 7>(acc, b[3], a);
		out[17 * 64 + t] = acc;
	}
	{
		double cc[7];
		for (int j = 0; j < 7; j++) cc[j] = c[j];
		selffma<G, 5, 7>(cc, a[3]);
		for (int j = 0; j < 7; j++) out[(18 + j) * 64 + t] = cc[j];
	}
	{
		double o6[6];
		transpose_lds<G, 6, 7>(pads[group<G>()], a, o6);  // a: row r (< 6) of a 6 x 7 matrix
		for (int j = 0; j < 6; j++) out[(25 + j) * 64 + t] = o6[j];
	}
	out[31 * 64 + t] = gather<G>(a[0], (r + 3) % G);
	out[32 * 64 + t] = allsum<G, 7>(a[1]);
	{
		// SPD matrix rows: S = X X^T + 7 I with X rows = a (lanes < 7)
		double s[7];
		for (int j = 0; j < 7; j++) s[j] = 0;
		double az[7];
		for (int j = 0; j < 7; j++) az[j] = (r < 7) ? a[j] : 0.0;
		mm_rt<G, 7, 7>(az, az, s);
		for (int j = 0; j < 7; j++) s[j] += (r < 7 && j == r) ? 7.0 : 0.0;
		spd_inverse_rows<G, 7>(s);
		for (int j = 0; j < 7; j++) out[(33 + j) * 64 + t] = s[j];
	}
}

template <int G>
static int run() {
	std::vector<double> in(21 * 64), out(40 * 64);
	for (size_t i = 0; i < in.size(); i++) in[i] = std::sin(0.37 * i + 0.1) + 0.01 * (i % 13);
	double *d_in, *d_out;
	hipMalloc(&d_in, in.size() * 8);
	hipMalloc(&d_out, out.size() * 8);
	hipMemcpy(d_in, in.data(), in.size() * 8, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(prims<G>, dim3(1), dim3(64), 0, 0, d_in, d_out);
	hipMemcpy(out.data(), d_out, out.size() * 8, hipMemcpyDeviceToHost);
	auto A = [&](int j, int t) { return in[j * 64 + t]; };
	auto Bm = [&](int j, int t) { return in[(7 + j) * 64 + t]; };
	auto Cm = [&](int j, int t) { return in[(14 + j) * 64 + t]; };
	int fails = 0;
	auto chk = [&](const char* what, int row, int t, double ref, double tol = 1e-13) {
		const double got = out[row * 64 + t];
		if (!(std::fabs(got - ref) <= tol * (1 + std::fabs(ref)))) {
			if (fails < 12) std::printf("G=%d %s row %d lane %d: got %.15g want %.15g\n", G, what, row, t, got, ref);
			fails++;
		}
	};
	for (int t = 0; t < 64; t++) {
		const int g0 = t & ~(G - 1), r = t & (G - 1);
		chk("bcast", 0, t, A(0, g0 + 3));
		chk("shift_up", 1, t, r >= 2 ? A(0, t - 2) : -1.0);
		chk("shift_down", 2, t, r + 4 < G ? A(0, t + 4) : -2.0);
		for (int j = 0; j < 7; j++) chk("rowfma", 3 + j, t, Cm(j, t) + Bm(j, g0 + 2) * A(1, t));
		for (int j = 0; j < 7; j++) chk("lanefma", 10 + j, t, Cm(j, t) + Bm(2, g0 + j) * A(2, t));
		{
			double s = Cm(0, t);
			for (int l = 0; l < 7; l++) s += Bm(3, g0 + l) * A(l, t);
			chk("mvfma", 17, t, s);
		}
		for (int j = 0; j < 7; j++) chk("selffma", 18 + j, t, Cm(j, t) + Cm(j, g0 + 5) * A(3, t));
		for (int j = 0; j < 6; j++) chk("transpose", 25 + j, t, r < 7 ? A(r, g0 + j) : 0.0);
		chk("gather", 31, t, A(0, g0 + (r + 3) % G));
		{
			double s = 0;
			for (int l = 0; l < 7; l++) s += A(1, g0 + l);
			chk("allsum", 32, t, s);
		}
	}
	// inverse: S * Sinv = I per group
	for (int g0 = 0; g0 < 64; g0 += G) {
		double S[49], Si[49];
		for (int i = 0; i < 7; i++)
			for (int j = 0; j < 7; j++) {
				double s = (i == j) ? 7.0 : 0.0;
				for (int l = 0; l < 7; l++) s += A(l, g0 + i) * A(l, g0 + j);
				S[i * 7 + j] = s;
				Si[i * 7 + j] = out[(33 + j) * 64 + g0 + i];
			}
		for (int i = 0; i < 7; i++)
			for (int j = 0; j < 7; j++) {
				double s = 0;
				for (int l = 0; l < 7; l++) s += S[i * 7 + l] * Si[l * 7 + j];
				if (std::fabs(s - (i == j)) > 1e-12) {
					if (fails < 12) std::printf("G=%d inverse group %d (%d,%d): %.3e\n", G, g0 / G, i, j, s - (i == j));
					fails++;
				}
			}
	}
	std::printf("G=%d: %s (%d mismatches)\n", G, fails ? "FAIL" : "ok", fails);
	return fails;
}

int main() { return (run<16>() + run<8>()) ? 1 : 0; }
