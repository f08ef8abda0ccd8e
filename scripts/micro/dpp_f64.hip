// DP-ALU DPP microbenchmark for the 16-lanes-per-robot kernels: does
// `v_fmac_f64_dpp ... row_newbcast:L` (gfx90a+: the one DPP control the FP64 pipe accepts) run at the
// rate of a plain v_fmac_f64, and does it read lane L of each 16-lane row?  Also times the two ways of
// moving a double between lanes that are not broadcasts (v_mov_b32_dpp row_shr, ds_swizzle).
// Build + run: hipcc --offload-arch=gfx950 -O3 dpp_f64.hip -o dpp_f64 && ./dpp_f64
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define FMAC_B(acc, bsrc, a, L) \
	asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:" #L " row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(bsrc), "v"(a))

template <int MODE>
__global__ __launch_bounds__(256) void fma_kernel(double* out, int iters, double a, double b) {
	double x[8], y = threadIdx.x * 1e-3;
#pragma unroll
	for (int i = 0; i < 8; i++) x[i] = threadIdx.x * 1e-3 + i;
	for (int it = 0; it < iters; it++) {
		if (MODE == 0) {
#pragma unroll
			for (int i = 0; i < 8; i++) x[i] = fma(y, a, x[i]);
		} else if (MODE == 1) {
			FMAC_B(x[0], y, a, 0); FMAC_B(x[1], y, a, 1); FMAC_B(x[2], y, a, 2); FMAC_B(x[3], y, a, 3);
			FMAC_B(x[4], y, a, 4); FMAC_B(x[5], y, a, 5); FMAC_B(x[6], y, a, 6); FMAC_B(x[7], y, a, 7);
		} else if (MODE == 2) {  // 64-bit move through two 32-bit DPP moves + fma
#pragma unroll
			for (int i = 0; i < 8; i++) {
				int lo = __double2loint(x[i]), hi = __double2hiint(x[i]);
				lo = __builtin_amdgcn_update_dpp(lo, lo, 0x111, 0xf, 0xf, false);  // row_shr:1
				hi = __builtin_amdgcn_update_dpp(hi, hi, 0x111, 0xf, 0xf, false);
				x[i] = fma(__hiloint2double(hi, lo), a, b);
			}
		} else {  // ds_swizzle xor 4 (two per double) + fma
#pragma unroll
			for (int i = 0; i < 8; i++) {
				int lo = __double2loint(x[i]), hi = __double2hiint(x[i]);
				lo = __builtin_amdgcn_ds_swizzle(lo, 0x101f);  // and 0x1f, or 0, xor 4
				hi = __builtin_amdgcn_ds_swizzle(hi, 0x101f);
				x[i] = fma(__hiloint2double(hi, lo), a, b);
			}
		}
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < 8; i++) s += x[i];
	if (s == 12345.678) out[0] = s;
}

template <int MODE>
double run(int waves_per_simd, double* d_out) {
	const int iters = 20000;
	const int blocks = 256 * waves_per_simd;
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(fma_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 0.999, 0.001);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(fma_kernel<MODE>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999, 0.001);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	// wave-instructions of the measured kind per second per SIMD -> cycles per instruction at 2.4 GHz
	const double instr = 8.0 * iters * blocks * 4;	// per wave: 8 per iteration; 4 waves per block
	const double per_simd_per_s = instr / (ms * 1e-3) / 1024.0;
	return 2.4e9 / per_simd_per_s;
}

__global__ void semantics(const double* in, double* out) {
	const int t = threadIdx.x;
	double x = in[t], acc3 = 0, acc12 = 0, one = 1.0;
	FMAC_B(acc3, x, one, 3);
	FMAC_B(acc12, x, one, 12);
	out[t] = acc3;
	out[64 + t] = acc12;
}

int main() {
	double* d_out;
	hipMalloc(&d_out, 8 * 256);
	std::printf("cycles per wave-instruction per SIMD (2.4 GHz assumed), 8 independent accumulators\n");
	const char* names[4] = {"v_fmac_f64 (plain)            ", "v_fmac_f64_dpp row_newbcast   ", "2x v_mov_b32_dpp row_shr + fma",
							"2x ds_swizzle + fma           "};
	for (int w : {1, 2, 4, 8}) {
		std::printf("  %d waves/SIMD: %s %5.2f | %s %5.2f | %s %5.2f (per double moved) | %s %5.2f\n", w, names[0], run<0>(w, d_out),
					names[1], run<1>(w, d_out), names[2], run<2>(w, d_out), names[3], run<3>(w, d_out));
	}
	std::vector<double> h(64), r(128);
	for (int i = 0; i < 64; i++) h[i] = 100 + i;
	double* d_in;
	hipMalloc(&d_in, 8 * 64);
	hipMemcpy(d_in, h.data(), 8 * 64, hipMemcpyHostToDevice);
	hipLaunchKernelGGL(semantics, dim3(1), dim3(64), 0, 0, d_in, d_out);
	hipMemcpy(r.data(), d_out, 8 * 128, hipMemcpyDeviceToHost);
	bool ok = true;
	for (int t = 0; t < 64; t++) ok = ok && r[t] == 100 + (t / 16) * 16 + 3 && r[64 + t] == 100 + (t / 16) * 16 + 12;
	std::printf("row_newbcast:L reads lane L of the lane's own 16-lane row: %s (lane 37 got %.0f and %.0f)\n", ok ? "yes" : "NO", r[37],
				r[64 + 37]);
	return ok ? 0 : 1;
}
