// What does the (usually empty) work-list pass behind the SVD-free tick kernel cost, and what would replace it?
// DESIGN.md §10.6 / VERDICT r2 item 6b. A = a stand-in for tick_fast_kernel (1 024 workgroups of one wavefront, one
// per SIMD through a 40 KB LDS allocation, ~25 us of dependent FMAs); B = a kernel that reads one counter and exits.
// Per-step time of 300 steps between ONE event pair:
//   1. A alone, back to back
//   2. A ; B with 1 024 / 16 / 1 workgroups (today's launch sequence)
//   3. A ; hipStreamWaitValue32 on a word the LAST workgroup of A writes (stream memory operation instead of a kernel)
//   4. A with T tail workgroups in the SAME launch that spin until the 1 024 are done (what a fused work-list pass
//      would cost when the list is empty)
//   5. A on the main stream, B on a side stream behind an event, the next A not waiting for B
// Build + run: hipcc --offload-arch=gfx950 -O3 launch_gap.hip -o launch_gap && ./launch_gap
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHECK(x)                                                                        \
	do {                                                                                \
		hipError_t e_ = (x);                                                            \
		if (e_ != hipSuccess) {                                                         \
			std::printf("%s failed: %s (line %d)\n", #x, hipGetErrorString(e_), __LINE__); \
			std::exit(1);                                                               \
		}                                                                               \
	} while (0)

constexpr int NWG = 1024;

// tail > 0: workgroups >= NWG wait for the first NWG to finish; signal != nullptr: the last of the NWG writes step
__global__ __launch_bounds__(64) void kernel_a(double* out, int iters, int* finished, unsigned* signal, unsigned step, int* count,
												 double* bulk) {
	extern __shared__ double lds[];
	if (blockIdx.x >= NWG) {  // tail workgroup: would run the work list; spins until the list is complete
		if (threadIdx.x == 0) {
			while (__hip_atomic_load(finished, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (int)(NWG * step)) __builtin_amdgcn_s_sleep(2);
		}
		__syncthreads();
		if (*(volatile int*)count > (int)threadIdx.x + 64 * ((int)blockIdx.x - NWG)) out[1] = 1.0;  // (never: count == 0)
		return;
	}
	double x = threadIdx.x * 1e-3, y = x + 1, z = x + 2, w = x + 3;
	for (int it = 0; it < iters; it++) {
		x = fma(x, 0.999, 0.001);
		y = fma(y, 0.999, 0.001);
		z = fma(z, 0.999, 0.001);
		w = fma(w, 0.999, 0.001);
	}
	lds[threadIdx.x] = x + y + z + w;
	if (lds[threadIdx.x] == 12345.678) out[0] = x;
	if (bulk) {	 // what the real kernel leaves behind: 27 rows of doubles per robot (torques, integrators), dirty in the L2s
		for (int r = 0; r < 27; r++) bulk[(size_t)r * NWG * 64 + blockIdx.x * 64 + threadIdx.x] = lds[threadIdx.x] + r;
	}
	if (finished && threadIdx.x == 0) {
		const int n = __hip_atomic_fetch_add(finished, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
		if (signal && n == (int)(NWG * step) - 1) __hip_atomic_store(signal, step, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
	}
}

__global__ __launch_bounds__(64) void kernel_b(const int* count, double* out) {
	const int i = blockIdx.x * 64 + threadIdx.x;
	if (i >= *(volatile const int*)count) return;
	out[2] = 1.0;
}

int main() {
	double* d_out;
	int *d_fin, *d_count;
	CHECK(hipMalloc(&d_out, 64));
	CHECK(hipMalloc(&d_fin, 4));
	CHECK(hipMalloc(&d_count, 4));
	CHECK(hipMemset(d_count, 0, 4));
	unsigned* d_sig = nullptr;
	int can_wait = 0;
	CHECK(hipDeviceGetAttribute(&can_wait, hipDeviceAttributeCanUseStreamWaitValue, 0));
	if (can_wait && hipExtMallocWithFlags((void**)&d_sig, 8, hipMallocSignalMemory) != hipSuccess) d_sig = nullptr;
	hipStream_t s, s2;
	CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
	CHECK(hipStreamCreateWithFlags(&s2, hipStreamNonBlocking));
	hipEvent_t e0, e1, dep;
	CHECK(hipEventCreate(&e0));
	CHECK(hipEventCreate(&e1));
	CHECK(hipEventCreateWithFlags(&dep, hipEventDisableTiming));
	const int steps = 300, lds = 40 * 1024;
	int iters = 700;
	auto timed = [&](const char* what, auto body) {
		for (int rep = 0; rep < 2; rep++) {	 // first repetition warms up
			CHECK(hipMemsetAsync(d_fin, 0, 4, s));
			if (d_sig) CHECK(hipMemsetAsync(d_sig, 0, 8, s));
			CHECK(hipStreamSynchronize(s));
			CHECK(hipEventRecord(e0, s));
			for (int k = 1; k <= steps; k++) body((unsigned)k);
			CHECK(hipEventRecord(e1, s));
			CHECK(hipStreamSynchronize(s));
			CHECK(hipStreamSynchronize(s2));
			float ms = 0;
			CHECK(hipEventElapsedTime(&ms, e0, e1));
			if (rep) std::printf("%-64s %7.2f us/step\n", what, ms * 1e3 / steps);
		}
	};
	double* d_bulk = nullptr;
	CHECK(hipMalloc(&d_bulk, sizeof(double) * 27 * NWG * 64));
	double* bulk = nullptr;
	auto A = [&](int tail, int* fin, unsigned* sig, unsigned k) {
		hipLaunchKernelGGL(kernel_a, dim3(NWG + tail), dim3(64), lds, s, d_out, iters, fin, sig, k, d_count, bulk);
	};
	for (int pass = 0; pass < 2; pass++) {
	bulk = pass ? d_bulk : nullptr;
	std::printf("---- A %s\n", pass ? "writes 14 MB per launch (dirty lines in the L2s at its end)" : "writes nothing");
	timed("1. A alone", [&](unsigned k) { A(0, nullptr, nullptr, k); });
	timed("1b. A alone + finished counter", [&](unsigned k) { A(0, d_fin, nullptr, k); });
	for (int g : {1024, 16, 1}) {
		char name[96];
		std::snprintf(name, sizeof name, "2. A ; B over an empty list, %d workgroups", g);
		timed(name, [&](unsigned k) {
			A(0, nullptr, nullptr, k);
			hipLaunchKernelGGL(kernel_b, dim3(g), dim3(64), 0, s, d_count, d_out);
		});
	}
	if (d_sig) {
		timed("3. A ; hipStreamWaitValue32(word written by A's last workgroup)", [&](unsigned k) {
			A(0, d_fin, d_sig, k);
			CHECK(hipStreamWaitValue32(s, d_sig, k, hipStreamWaitValueGte, 0xffffffffu));
		});
	} else {
		std::printf("3. hipStreamWaitValue32: not supported here (attribute %d)\n", can_wait);
	}
	for (int t : {1, 4, 16}) {
		char name[96];
		std::snprintf(name, sizeof name, "4. A with %d tail workgroups spinning in the same launch", t);
		timed(name, [&](unsigned k) { A(t, d_fin, nullptr, k); });
	}
	for (int g : {1024, 1}) {
		char name[96];
		std::snprintf(name, sizeof name, "5. A ; B(%d) on a side stream behind an event, next A not waiting", g);
		timed(name, [&](unsigned k) {
			A(0, nullptr, nullptr, k);
			CHECK(hipEventRecord(dep, s));
			CHECK(hipStreamWaitEvent(s2, dep, 0));
			hipLaunchKernelGGL(kernel_b, dim3(g), dim3(64), 0, s2, d_count, d_out);
		});
	}
	}
	return 0;
}
