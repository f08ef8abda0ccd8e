// sai2b_group.hip — the generic tick with a robot spread over G = 16 or 8 lanes (sai2b_group_tick.hpp): what runs
// for hierarchies outside the SVD-free path, and for the robots that path hands over (its work list).
#include <hip/hip_runtime.h>

#include "sai2b_group_tick.hpp"
#include "sai2b_launch.h"

#ifndef SAI2B_GROUP_WAVES
#define SAI2B_GROUP_WAVES 1
#endif
#define SAI2B_GROUP_OCC __attribute__((amdgpu_waves_per_eu(SAI2B_GROUP_WAVES, SAI2B_GROUP_WAVES)))

namespace sai2b {

// fb_count == NULL: group i of the grid takes robot i. Otherwise the pass behind tick_fast_kernel: the groups
// stride over the compacted work list fb_list[0 .. *fb_count) (the grid is sized for the machine, not for the list).
template <int G, bool RANGE>
__global__ __launch_bounds__(64) SAI2B_GROUP_OCC void tick_group_kernel(const DevParams* __restrict__ Pp, int commit_sh, int with_comp,
														   int do_torque, const int* __restrict__ fb_count,
														   const int* __restrict__ fb_list) {
	constexpr int GPB = 64 / G;	 // robots per workgroup (one wavefront)
	__shared__ real pads[GPB][grp::PAD_DOUBLES];
	const DevParams& P = *Pp;
	const int gi = grp::group<G>();
	if (fb_count) {
		const int cnt = *(const gint*)fb_count;
#pragma unroll 1
		// entry e -> workgroup e % gridDim.x: a short list gives every robot a wavefront of its own (robots of one
		// wavefront that take different branches run one after the other)
		for (int e = blockIdx.x + gi * gridDim.x; e < cnt; e += gridDim.x * GPB)
			grp::tick_robot<G, RANGE>(P, ((const gint*)fb_list)[e], pads[gi], commit_sh, with_comp, do_torque);
		return;
	}
	const int b = blockIdx.x * GPB + gi;
	if (b >= P.B) return;
	grp::tick_robot<G, RANGE>(P, b, pads[gi], commit_sh, with_comp, do_torque);
}

// the TemplateTask calls, lanes per robot (grp::task_robot); tk_count / tk_list as above
template <int G>
__global__ __launch_bounds__(64) SAI2B_GROUP_OCC void task_group_kernel(const DevParams* __restrict__ Pp, int task, const double* __restrict__ Nprec_in,
														   const double* __restrict__ tau_prec, double* __restrict__ tau_out,
														   double* __restrict__ N_out, double* __restrict__ Ntot_out, int commit_sh, int do_torque,
														   const int* __restrict__ tk_count, const int* __restrict__ tk_list) {
	constexpr int GPB = 64 / G;
	__shared__ real pads[GPB][grp::PAD_DOUBLES];
	const DevParams& P = *Pp;
	const int gi = grp::group<G>();
	if (tk_count) {
		const int cnt = *(const gint*)tk_count;
#pragma unroll 1
		for (int e = blockIdx.x + gi * gridDim.x; e < cnt; e += gridDim.x * GPB)
			grp::task_robot<G>(P, task, ((const gint*)tk_list)[e], pads[gi], Nprec_in, tau_prec, tau_out, N_out, Ntot_out, commit_sh, do_torque);
		return;
	}
	const int b = blockIdx.x * GPB + gi;
	if (b >= P.B) return;
	grp::task_robot<G>(P, task, b, pads[gi], Nprec_in, tau_prec, tau_out, N_out, Ntot_out, commit_sh, do_torque);
}

}  // namespace sai2b

#ifdef SAI2B_GROUP_STAMP
// diagnostic build only (scripts/micro/group_stamps.py)
extern "C" void sai2b_debug_reset_stamps() {
	int zero = 0;
	(void)hipMemcpyToSymbol(HIP_SYMBOL(sai2b::grp::g_stamp_n), &zero, sizeof(int));
}
extern "C" int sai2b_debug_read_stamps(unsigned long long* out, int cap) {
	int n = 0;
	(void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(sai2b::grp::g_stamp_n), sizeof(int));
	if (2 * n > cap) n = cap / 2;
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(sai2b::grp::g_stamps), sizeof(unsigned long long) * 2 * n);
	return n;
}
#endif

// lanes: 16 or 8. range_only: the pass ahead of the generator kernels (gated JointTasks)
extern "C" int sai2b_launch_tick_group(const sai2b::DevParams* d_params, int B, int lanes, int range_only, int commit_sh, int with_comp,
									   int do_torque, const int* fb_count, const int* fb_list, hipStream_t stream) {
	const int gpb = 64 / lanes;
	int blocks = (B + gpb - 1) / gpb;
	// the pass over a work list strides over it: one wavefront per SIMD is all this kernel can have resident (512
	// registers), so more workgroups than SIMDs only add launch cost when the list is short, which it usually is
	if (fb_count && blocks > 1024) blocks = 1024;
	const dim3 grid(blocks), block(64);
	if (lanes == 16) {
		if (range_only)
			hipLaunchKernelGGL((sai2b::tick_group_kernel<16, true>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque, fb_count, fb_list);
		else
			hipLaunchKernelGGL((sai2b::tick_group_kernel<16, false>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque, fb_count, fb_list);
	} else {
		if (range_only)
			hipLaunchKernelGGL((sai2b::tick_group_kernel<8, true>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque, fb_count, fb_list);
		else
			hipLaunchKernelGGL((sai2b::tick_group_kernel<8, false>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque, fb_count, fb_list);
	}
	return (int)hipGetLastError();
}

// the TemplateTask calls with a robot spread over `lanes` (16: a short work list, latency; 8: a whole batch, throughput)
extern "C" int sai2b_launch_task_group(const sai2b::DevParams* d_params, int B, int lanes, int task, const double* Nprec_in, const double* tau_prec,
									   double* tau_out, double* N_out, double* Ntot_out, int commit_sh, int do_torque, const int* tk_count,
									   const int* tk_list, hipStream_t stream) {
	const int gpb = 64 / lanes;
	int blocks = (B + gpb - 1) / gpb;
	if (tk_count && blocks > 1024) blocks = 1024;
	const dim3 grid(blocks), block(64);
	if (lanes == 16)
		hipLaunchKernelGGL((sai2b::task_group_kernel<16>), grid, block, 0, stream, d_params, task, Nprec_in, tau_prec, tau_out, N_out, Ntot_out,
						   commit_sh, do_torque, tk_count, tk_list);
	else
		hipLaunchKernelGGL((sai2b::task_group_kernel<8>), grid, block, 0, stream, d_params, task, Nprec_in, tau_prec, tau_out, N_out, Ntot_out,
						   commit_sh, do_torque, tk_count, tk_list);
	return (int)hipGetLastError();
}
