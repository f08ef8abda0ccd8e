// FP64 vector-FMA peak microbenchmark for the roofline of DESIGN.md ("confirm the
// datasheet value with a microbenchmark"): independent v_fma_f64 chains per lane, no memory traffic.
// Build + run: hipcc --offload-arch=gfx950 -O3 fp64_peak.hip -o fp64_peak && ./fp64_peak
#include <hip/hip_runtime.h>

#include <cstdio>

template <int CHAINS>
__global__ __launch_bounds__(256) void fma_kernel(double* out, int iters, double a, double b) {
	double x[CHAINS];
#pragma unroll
	for (int i = 0; i < CHAINS; i++) x[i] = threadIdx.x * 1e-3 + i;
	for (int it = 0; it < iters; it++) {
#pragma unroll
		for (int i = 0; i < CHAINS; i++) x[i] = fma(x[i], a, b);
	}
	double s = 0;
#pragma unroll
	for (int i = 0; i < CHAINS; i++) s += x[i];
	if (s == 12345.678) out[0] = s;	 // keeps the chains alive
}

template <int CHAINS>
double run(int waves_per_simd, double* d_out) {
	const int iters = 20000;
	const int blocks = 256 * waves_per_simd;  // 256 CUs x 4 SIMDs: one 256-thread block = 4 waves = one per SIMD of a CU
	hipEvent_t e0, e1;
	hipEventCreate(&e0), hipEventCreate(&e1);
	hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, d_out, 100, 0.999, 0.001);
	hipDeviceSynchronize();
	hipEventRecord(e0, 0);
	hipLaunchKernelGGL(fma_kernel<CHAINS>, dim3(blocks), dim3(256), 0, 0, d_out, iters, 0.999, 0.001);
	hipEventRecord(e1, 0);
	hipEventSynchronize(e1);
	float ms = 0;
	hipEventElapsedTime(&ms, e0, e1);
	const double flops = 2.0 * CHAINS * (double)iters * blocks * 256;
	return flops / (ms * 1e-3) / 1e12;
}

int main() {
	double* d_out;
	hipMalloc(&d_out, 8);
	std::printf("FP64 vector FMA, TFLOP/s (chains per lane x waves per SIMD)\n");
	std::printf("  1 chain  x 1 wave : %6.1f   (dependent-chain latency bound)\n", run<1>(1, d_out));
	std::printf("  2 chains x 1 wave : %6.1f\n", run<2>(1, d_out));
	std::printf("  4 chains x 1 wave : %6.1f\n", run<4>(1, d_out));
	std::printf("  8 chains x 1 wave : %6.1f\n", run<8>(1, d_out));
	std::printf("  8 chains x 2 waves: %6.1f\n", run<8>(2, d_out));
	std::printf("  4 chains x 2 waves: %6.1f\n", run<4>(2, d_out));
	std::printf("  1 chain  x 4 waves: %6.1f\n", run<1>(4, d_out));
	std::printf("  8 chains x 4 waves: %6.1f\n", run<8>(4, d_out));
	std::printf("  8 chains x 8 waves: %6.1f   (peak)\n", run<8>(8, d_out));
	return 0;
}
