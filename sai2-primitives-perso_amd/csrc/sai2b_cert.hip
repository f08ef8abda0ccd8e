// sai2b_cert.hip — the SVD-free tick for general hierarchies (sai2b_cert.hpp), one lane per robot: what runs
// first for every hierarchy outside [full MFT(, full JT)]; the robots it declines go to the lanes-per-robot
// generic kernel (sai2b_group.hip) through the same work list as behind tick_fast_kernel.
#include <hip/hip_runtime.h>

#include "sai2b_cert.hpp"
#include "sai2b_launch.h"

namespace sai2b {

// fb_counts / fb_list / parity: as tick_fast_kernel (sai2b_kernels.hip). MCAP: most rows a partial task of the
// hierarchy brings; the instantiation for small tasks (position-only MotionForceTask, a few selected joints) does
// not carry the register footprint of a 6- or 7-row level.
// (The robot constants are read from the parameter block: compiling the stock Panda in, as tick_fast_kernel does,
// measured 9 % SLOWER here — 53.2 vs 49.0 us on the same box — the literals cost registers this kernel does not have.)
// fb_counts[2 + parity]: how many robots went through the in-lane singular branch (cleared and filled like the work list's count)
template <int MCAP, bool S6 = false>
__global__ __launch_bounds__(64) void tick_cert_kernel(const DevParams* __restrict__ Pp, int with_comp, int* __restrict__ fb_counts,
													  int* __restrict__ fb_list, int parity) {
	__shared__ real pend_lds[(cert::LDS_SLOTS + (MCAP <= 3 ? cert::POSE_SLOTS : 0)) * 64];
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (blockIdx.x == 0 && threadIdx.x == 0) ((gint*)fb_counts)[1 - parity] = 0, ((gint*)fb_counts)[2 + (1 - parity)] = 0;
	if (b >= B) return;
	real* pend = pend_lds + threadIdx.x;
	real tau[N];
	// with_comp: bit 0 = JointTask compensation of the tasks above, bit 1 = singular MotionForceTasks go to the work list
	// instead of through cert::singular_part (SAI2B_NO_INLANE_SINGULAR=1, the A/B switch)
	cert::SingPend sp;
	sp.task = -1, sp.commit = 1, sp.store_t2 = 1, sp.took = 0;
	const bool mine = cert::tick<MCAP, cert::DM, DevModel, false, S6>(P, P.model, B, b, (with_comp & 1) != 0, pend, tau, nullptr, (with_comp & 2) ? nullptr : &sp);
	{
		const unsigned long long took = __ballot(sp.took != 0);
		if (took && threadIdx.x == 0) atomicAdd(&fb_counts[2 + parity], __popcll(took));  // lane 0 is always in range
	}
	const unsigned long long declined = __ballot(!mine);
	if (declined) {
		int base = 0;
		if (threadIdx.x == 0) base = atomicAdd(&fb_counts[parity], __popcll(declined));	 // lane 0 is always in range
		base = __shfl(base, 0);
		if (!mine) {
			((gint*)fb_list)[base + __popcll(declined & ((1ull << threadIdx.x) - 1ull))] = b;
			return;
		}
	}
	cert::flush(P, B, b, pend);
	cert::flush_singular(P, B, b, sp);
	UNROLL for (int i = 0; i < N; i++) st(P.tau, i, B, b, tau[i] + pend[i * 64]);  // RobotController.cpp:70-72
}

// The TemplateTask calls on one task (updateTaskModel(N_prec), computeTorques(), computeTorques(tau_prec), the nullspace
// getters: TemplateTask.h:42-88) through the same whitened cascade (cert::tick<.., TASK = true>): what examples 01 / 04 /
// 18 / 19 do by hand per period. Robots whose N_prec is not a whitened projector, whose level is not certified or that
// carry singularity history go — nothing stored — to a work list for the generic task_kernel (sai2b_kernels.hip).
// tk_counts: two counters alternating between launches (`parity`), as behind tick_fast_kernel: this launch fills
// [parity] and clears [1 - parity] for the next one.
template <int MCAP>
__global__ __launch_bounds__(64) void task_cert_kernel(const DevParams* __restrict__ Pp, int task, const double* __restrict__ Nprec_in,
													  const double* __restrict__ tau_prec, double* __restrict__ tau_out, double* __restrict__ N_out,
													  double* __restrict__ Ntot_out, int call_bits, int* __restrict__ tk_counts,
													  int* __restrict__ tk_list, int parity) {
	// call_bits: bit 0 = torques (computeTorques), bit 1 = the call commits the once-per-model-update singularity bookkeeping
	// (updateTaskModel, or a computeTorques that has to update the model itself), bit 2 = singular MotionForceTasks go to the
	// work list (SAI2B_NO_INLANE_SINGULAR)
	const int do_torque = call_bits & 1;
	__shared__ real pend_lds[(cert::LDS_SLOTS + cert::TASK_EXTRA) * 64];
	static_assert((cert::LDS_SLOTS + cert::TASK_EXTRA) * 64 * 8 * 4 <= 160 * 1024 || N > 7, "four wavefronts per CU (robots of up to 7 joints)");
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (blockIdx.x == 0 && threadIdx.x == 0) ((gint*)tk_counts)[1 - parity] = 0;
	int* tk_count = tk_counts + parity;
	if (b >= B) return;
	real* pend = pend_lds + threadIdx.x;
	cert::TaskArgs io;
	io.task = task, io.Nprec = Nprec_in, io.tau_prec = tau_prec, io.N_out = N_out, io.Ntot_out = Ntot_out, io.q0 = pend;
	io.write_active = do_torque ? 0 : 1;
	real tau[N];
	cert::SingPend sp;
	sp.task = -1, sp.commit = (call_bits >> 1) & 1, sp.store_t2 = do_torque, sp.took = 0;
	const bool mine = cert::tick<MCAP, cert::DM, DevModel, true>(P, P.model, B, b, tau_prec != nullptr, pend, tau, &io, (call_bits & 4) ? nullptr : &sp);
	const unsigned long long declined = __ballot(!mine);
	if (declined) {
		int base = 0;
		if (threadIdx.x == 0) base = atomicAdd(tk_count, __popcll(declined));  // lane 0 is always in range
		base = __shfl(base, 0);
		if (!mine) {
			((gint*)tk_list)[base + __popcll(declined & ((1ull << threadIdx.x) - 1ull))] = b;
			return;
		}
	}
	cert::flush_singular(P, B, b, sp);
	if (do_torque) {
		cert::flush_task(P, task, B, b, pend);
		if (tau_out) {
			UNROLL for (int i = 0; i < N; i++) st(tau_out, i, B, b, tau[i] - (tau_prec ? ld(tau_prec, i, B, b) : 0.0));
		}
	}
}

// The range pass ahead of the trajectory generators (cert::range_tick): OTG_ACTIVE of the gated JointTasks for the
// robots whose levels are all certified; the others go to a work list for the generic kernel's range pass, with
// the same two alternating counters protocol as above.
template <int MCAP>
__global__ __launch_bounds__(64) void range_cert_kernel(const DevParams* __restrict__ Pp, int* __restrict__ rg_counts, int* __restrict__ rg_list,
													   int parity, int inlane) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (blockIdx.x == 0 && threadIdx.x == 0) ((gint*)rg_counts)[1 - parity] = 0;
	if (b >= B) return;
	const bool mine = cert::range_tick<MCAP>(P, P.model, B, b, inlane != 0);
	const unsigned long long declined = __ballot(!mine);
	if (declined) {
		int base = 0;
		if (threadIdx.x == 0) base = atomicAdd(&rg_counts[parity], __popcll(declined));  // lane 0 is always in range
		base = __shfl(base, 0);
		if (!mine) ((gint*)rg_list)[base + __popcll(declined & ((1ull << threadIdx.x) - 1ull))] = b;
	}
}

}  // namespace sai2b

#ifdef SAI2B_CERT_STAMP
// diagnostic build only (scripts/micro/cert_stamps.py)
extern "C" void sai2b_debug_reset_cstamps() {
	int zero = 0;
	(void)hipMemcpyToSymbol(HIP_SYMBOL(sai2b::cert::g_cstamp_n), &zero, sizeof(int));
}
extern "C" int sai2b_debug_read_cstamps(unsigned long long* out, int cap) {
	int n = 0;
	(void)hipMemcpyFromSymbol(&n, HIP_SYMBOL(sai2b::cert::g_cstamp_n), sizeof(int));
	if (2 * n > cap) n = cap / 2;
	(void)hipMemcpyFromSymbol(out, HIP_SYMBOL(sai2b::cert::g_cstamps), sizeof(unsigned long long) * 2 * n);
	return n;
}
#endif

// inlane: robots inside a blending region of a 2- or 3-row MotionForceTask stay in the kernel (cert::singular_range)
extern "C" int sai2b_launch_range_cert(const sai2b::DevParams* d_params, int B, int max_rows, int* rg_counts, int* rg_list, int parity, int inlane,
									   hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	if (max_rows <= 3)
		hipLaunchKernelGGL(sai2b::range_cert_kernel<3>, grid, block, 0, stream, d_params, rg_counts, rg_list, parity, inlane);
	else
		hipLaunchKernelGGL(sai2b::range_cert_kernel<6>, grid, block, 0, stream, d_params, rg_counts, rg_list, parity, 0);
	return (int)hipGetLastError();
}

extern "C" int sai2b_launch_task_cert(const sai2b::DevParams* d_params, int B, int task, int max_rows, const double* Nprec_in,
									  const double* tau_prec, double* tau_out, double* N_out, double* Ntot_out, int do_torque, int* tk_counts,
									  int* tk_list, int parity, hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	if (max_rows <= 3)
		hipLaunchKernelGGL(sai2b::task_cert_kernel<3>, grid, block, 0, stream, d_params, task, Nprec_in, tau_prec, tau_out, N_out, Ntot_out,
						   do_torque, tk_counts, tk_list, parity);
	else
		hipLaunchKernelGGL(sai2b::task_cert_kernel<6>, grid, block, 0, stream, d_params, task, Nprec_in, tau_prec, tau_out, N_out, Ntot_out,
						   do_torque, tk_counts, tk_list, parity);
	return (int)hipGetLastError();
}

extern "C" int sai2b_launch_tick_cert(const sai2b::DevParams* d_params, int B, int max_rows, int with_comp, int* fb_counts, int* fb_list,
									  int parity, hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	// with_comp bit 2: the 6-row instantiation with the singular branch in the lane (cert::tick<.., S6>)
	if (max_rows <= 3)
		hipLaunchKernelGGL((sai2b::tick_cert_kernel<3>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
	else if (with_comp & 4)
		hipLaunchKernelGGL((sai2b::tick_cert_kernel<6, true>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
	else
		hipLaunchKernelGGL((sai2b::tick_cert_kernel<6>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
	return (int)hipGetLastError();
}
