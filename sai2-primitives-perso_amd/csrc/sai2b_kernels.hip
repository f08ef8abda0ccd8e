// sai2b_kernels.hip — gfx950 kernels of the batched operational-space controller.
//
// tick_kernel: one launch = RobotController::updateControllerTaskModels() +
// computeControlTorques() (reference src/RobotController.cpp:53-74) for B robots, preceded by the
// model update the reference gets from Sai2Model::updateModel(). One lane per robot, 64-thread
// workgroups (one wavefront each) so that B/64 workgroups spread over all 1024 SIMDs.
#include <hip/hip_runtime.h>

#include "sai2b_device.hpp"
#include "sai2b_fast.hpp"
#include "sai2b_baked_panda.h"
#include "sai2b_launch.h"

namespace sai2b {

// Generic tick: Jacobi-SVD based, any hierarchy (the reference's control flow, projector form).
// fb_count != NULL: the fallback pass behind tick_fast_kernel — lane i of the grid takes robot
// fb_list[i] for i < *fb_count (the robots the SVD-free kernel declined, compacted), the rest exits.
// RANGE: the pass ahead of the generator kernels that only decides which robots' gated JointTasks have a
// range this tick (DevTask::otg_gated): nothing after the last such task is needed. A separate
// instantiation, so that the tick proper compiles as it did without it.
template <bool DEBUG, bool RANGE = false>
__global__ __launch_bounds__(64) void tick_kernel(const DevParams* __restrict__ Pp, int commit_sh, int with_comp,
													 int do_torque, const int* __restrict__ fb_count,
													 const int* __restrict__ fb_list) {
	const DevParams& P = *Pp;
	const int B = P.B;
	int b = blockIdx.x * 64 + threadIdx.x;
	if (fb_count) {
		if (b >= *(const gint*)fb_count) return;
		b = ((const gint*)fb_list)[b];
	}
	if (b >= B) return;
	RobotCtx rc;
	UNROLL for (int i = 0; i < N; i++) {
		rc.q[i] = ld(P.q, i, B, b);
		rc.dq[i] = ld(P.dq, i, B, b);
	}
	real g[N];
	{
		// Sai2Model::updateModel(): kinematics, M (CRBA), M^-1 (examples/05-using_robot_controller.cpp:143-145)
		Frames F;
		fk(P.model, rc.q, F);
		real M[N * N];
		mass_matrix(P.model, F, M);
		spd_inverse<N>(M, rc.Minv);
		if (DEBUG && P.dbg_M) {
			UNROLL for (int i = 0; i < N * N; i++) st(P.dbg_M, i, B, b, M[i]);
		}
		// bounded inertia estimate (SingularityHandler.cpp:176-182, JointTask.cpp:254-260), shared by
		// every task with the same threshold (the reference recomputes it per task: SURVEY App. B-8)
		bool any_bie = false;
		real thr = 0;
		for (int t = 0; t < P.n_tasks; t++)
			if (P.task[t].decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
				any_bie = true;
				thr = P.task[t].bie_threshold;
			}
		if (any_bie) {
			UNROLL for (int i = 0; i < N; i++) M[i * N + i] = fmax(M[i * N + i], thr);
			spd_inverse<N>(M, rc.MinvB);
		} else {
			UNROLL for (int i = 0; i < N * N; i++) rc.MinvB[i] = rc.Minv[i];
		}
		if (P.gravity_comp)
			gravity_vector(P.model, F, g);
		else {
			UNROLL for (int i = 0; i < N; i++) g[i] = 0;
		}
	}
	real Nprec[N * N], tau[N];
	UNROLL for (int i = 0; i < N * N; i++) Nprec[i] = (i % (N + 1) == 0) ? 1.0 : 0.0;
	UNROLL for (int i = 0; i < N; i++) tau[i] = 0;
	Chain chain;
	chain.ok = !DEBUG;
	chain.wrows = 0;
	UNROLL for (int i = 0; i < N * N; i++) chain.W[i] = 0;
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
			mft_task<DEBUG>(P, tk, rc, B, b, first, last, commit_sh != 0, do_torque != 0, Nprec, tau, chain);
		else if (RANGE && t == n_run - 1)
			jt_task<DEBUG, true>(P, tk, rc, B, b, first, last, with_comp != 0, do_torque != 0, Nprec, tau, chain);
		else
			jt_task<DEBUG>(P, tk, rc, B, b, first, last, with_comp != 0, do_torque != 0, Nprec, tau, chain);
	}
	if (do_torque) {
		UNROLL for (int i = 0; i < N; i++) st(P.tau, i, B, b, tau[i] + g[i]);  // RobotController.cpp:70-72
	}
}

#if SAI2B_N == 7  // the SVD-free kernels are for 7-joint robots (6-DOF task + a one-dimensional nullspace)
// FAST = 1: hierarchy [full MFT]; FAST = 2: [full MFT, full JT] — the SVD-free path of
// sai2b_fast.hpp. A robot takes it only when it is certified non-singular (and is not leaving a
// singular region); otherwise its lane touches no state and appends the robot to the work list of
// the generic kernel launched right behind (one atomic per wavefront that has such robots).
// fb_counts: two counters alternating between ticks (`parity`): this launch fills [parity] and clears
// [1 - parity] for the next one (the generic pass that read it finished before this kernel started).
// BAKED selects where the robot constants come from: false = the ctx's parameter block (any robot),
// true = the compile-time Panda literals of sai2b_baked_panda.h (chosen by the host only when the ctx
// model is bit-equal to them): no scalar loads from the parameter block for the model phase.
template <int FAST, bool BAKED>
__global__ __launch_bounds__(64) void tick_fast_kernel(const DevParams* __restrict__ Pp, int with_comp,
														  int* __restrict__ fb_counts, int* __restrict__ fb_list,
														  int parity) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (blockIdx.x == 0 && threadIdx.x == 0) ((gint*)fb_counts)[1 - parity] = 0;
	if (b >= B) return;
	RobotCtx rc;
	UNROLL for (int i = 0; i < N; i++) {
		rc.q[i] = ld(P.q, i, B, b);
		rc.dq[i] = ld(P.dq, i, B, b);
	}
	const DevTask& t0 = P.task[0];
	const bool clean = ldi(t0.istate, IS_NTYPES, B, b) == 0;
	// Phase order keeps the live set small (only 256 of the 512 registers are VGPRs the VALU can
	// address; the rest are AGPRs the compiler uses as spill space) and gives every load a phase of
	// arithmetic to hide behind: JT law inputs -> FK/Jacobian -> MFT law -> CRBA -> certificate.
	JtEarly jt;
	if (FAST == 2) fast_jt_early(P.task[1], rc, B, b, jt);
	MftIn in0;
	mft_load(t0, B, b, in0);
	SAI2B_PHASE();
	real J[6 * N], M[N * N], g[N], Fu[6], Ff[6];
	{
		Frames F;
		if constexpr (BAKED)
			fk(PandaBaked{}, rc.q, F);
		else
			fk(P.model, rc.q, F);
		real x[3], R[9];
		frame_pose(t0, F, x, R);
		if constexpr (BAKED)
			jacobian(PandaBaked{}, t0, F, x, J);
		else
			jacobian(P.model, t0, F, x, J);
		SAI2B_PHASE();
		mft_law(t0, rc, J, x, R, in0, Fu, Ff);	// MotionForceTask.cpp:278-503 (integrators not yet stored)
		SAI2B_PHASE();
		if constexpr (BAKED)
			mass_matrix(PandaBaked{}, F, M);
		else
			mass_matrix(P.model, F, M);
		if (P.gravity_comp) {
			if constexpr (BAKED)
				gravity_vector(PandaBaked{}, F, g);
			else
				gravity_vector(P.model, F, g);
		}
		else {
			UNROLL for (int i = 0; i < N; i++) g[i] = 0;
		}
	}
	SAI2B_PHASE();
	const bool ok = certify_nonsingular(J, t0.s_abs_tol, t0.s_max);
	const bool mine = ok && clean;
	const unsigned long long declined = __ballot(!mine);
	if (declined) {
		int base = 0;
		if (threadIdx.x == 0) base = atomicAdd(&fb_counts[parity], __popcll(declined));  // lane 0 is always in range
		base = __shfl(base, 0);
		if (!mine) {
			((gint*)fb_list)[base + __popcll(declined & ((1ull << threadIdx.x) - 1ull))] = b;
			return;
		}
	}
	// committed to the fast path: integrators can go out now
	mft_store_integrators(t0, B, b, in0);
	if (FAST == 2) {
		UNROLL for (int i = 0; i < N; i++) st(P.task[1].state, i, B, b, jt.integ[i]);
	}
	SAI2B_PHASE();
	real tau[N];
	fast_tick<FAST == 2>(P, J, M, Fu, Ff, B, b, with_comp != 0, jt, tau);
	UNROLL for (int i = 0; i < N; i++) st(P.tau, i, B, b, tau[i] + g[i]);
}

#endif	// SAI2B_N == 7

// One task driven on its own, the reference's plugin interface (TemplateTask.h:42-88):
//   updateTaskModel(N_prec)            -> do_torque = 0: the task's nullspace N and N * N_prec; the once-per-
//                                         update singularity bookkeeping when commit_sh
//   computeTorques() / (tau_prec)      -> do_torque = 1: the task's own torques (tau_prec == NULL: the
//                                         no-argument form, no compensation of the previous tasks)
// Nprec_in == NULL is the identity (a task constructed and never given one, JointTask.cpp:62,
// MotionForceTask.cpp:138). Nothing is known about the tasks above, so the range decisions always take
// the Jacobi SVD (the introspection instantiation of the task functions), and a JointTask applies the
// compensation term also behind an identity N_prec, as JointTask::computeTorques(tau_prec) does
// (JointTask.cpp:285-292). Used for examples 01 / 04 / 18-style manual hierarchies; the batched hot path is
// the fused tick above.
// tk_count != NULL: the pass behind task_cert_kernel (sai2b_cert.hip) over the robots it declined, compacted.
__global__ __launch_bounds__(64) void task_kernel(const DevParams* __restrict__ Pp, int task, const double* __restrict__ Nprec_in,
													 const double* __restrict__ tau_prec, double* __restrict__ tau_out,
													 double* __restrict__ N_out, double* __restrict__ Ntot_out, int commit_sh,
													 int do_torque, const int* __restrict__ tk_count, const int* __restrict__ tk_list) {
	const DevParams& P = *Pp;
	const int B = P.B;
	int b = blockIdx.x * 64 + threadIdx.x;
	if (tk_count) {
		if (b >= *(const gint*)tk_count) return;
		b = ((const gint*)tk_list)[b];
	}
	if (b >= B) return;
	RobotCtx rc;
	UNROLL for (int i = 0; i < N; i++) {
		rc.q[i] = ld(P.q, i, B, b);
		rc.dq[i] = ld(P.dq, i, B, b);
	}
	const DevTask& tk = P.task[task];
	{
		Frames F;
		fk(P.model, rc.q, F);
		real M[N * N];
		mass_matrix(P.model, F, M);
		spd_inverse<N>(M, rc.Minv);
		if (tk.decoupling == SAI2B_BOUNDED_INERTIA_ESTIMATES) {
			UNROLL for (int i = 0; i < N; i++) M[i * N + i] = fmax(M[i * N + i], tk.bie_threshold);
			spd_inverse<N>(M, rc.MinvB);
		} else {
			UNROLL for (int i = 0; i < N * N; i++) rc.MinvB[i] = rc.Minv[i];
		}
	}
	real Nprec[N * N], tau[N], tp[N];
	UNROLL for (int i = 0; i < N * N; i++) Nprec[i] = Nprec_in ? ld(Nprec_in, i, B, b) : ((i % (N + 1) == 0) ? 1.0 : 0.0);
	UNROLL for (int i = 0; i < N; i++) tau[i] = tp[i] = tau_prec ? ld(tau_prec, i, B, b) : 0.0;
	Chain chain;
	chain.ok = false;
	chain.wrows = 0;
	UNROLL for (int i = 0; i < N * N; i++) chain.W[i] = 0;
	if (tk.type == SAI2B_MOTION_FORCE_TASK)
		mft_task<true>(P, tk, rc, B, b, false, false, commit_sh != 0, do_torque != 0, Nprec, tau, chain, N_out);
	else
		jt_task<true>(P, tk, rc, B, b, false, false, tau_prec != nullptr, do_torque != 0, Nprec, tau, chain, N_out);
	if (Ntot_out) {
		UNROLL for (int i = 0; i < N * N; i++) st(Ntot_out, i, B, b, Nprec[i]);
	}
	if (do_torque && tau_out) {
		UNROLL for (int i = 0; i < N; i++) st(tau_out, i, B, b, tau[i] - tp[i]);
	}
}

// RobotController::reinitializeTasks (RobotController.cpp:76-80): MotionForceTask::reInitializeTask
// (MotionForceTask.cpp:204-245), SingularityHandler ctor state (SingularityHandler.cpp:53-63),
// JointTask::reInitializeTask (JointTask.cpp:91-107)
__global__ __launch_bounds__(64) void reinit_kernel(const DevParams* __restrict__ Pp, int only_task) {
	const DevParams& P = *Pp;
	const int B = P.B;
	const int b = blockIdx.x * 64 + threadIdx.x;
	if (b >= B) return;
	real q[N];
	UNROLL for (int i = 0; i < N; i++) q[i] = ld(P.q, i, B, b);
#pragma unroll 1
	for (int t = 0; t < P.n_tasks; t++) {
		const DevTask& tk = P.task[t];
		if (only_task >= 0 && t != only_task) continue;	 // TemplateTask::reInitializeTask of one task
		if (tk.type == SAI2B_MOTION_FORCE_TASK) {
			real x[3], R[9];
			det_frame_pose(P.model, tk, q, x, R);
			UNROLL for (int k = 0; k < 3; k++) st(tk.goals, k, B, b, x[k]);
			UNROLL for (int k = 0; k < 9; k++) st(tk.goals, 3 + k, B, b, R[k]);
			for (int k = 12; k < MFT_GOAL_ROWS; k++) st(tk.goals, k, B, b, 0.0);
			for (int k = 0; k < 6; k++) st(tk.sensed, k, B, b, 0.0);
			for (int k = 0; k < 12; k++) st(tk.state, k, B, b, 0.0);
			UNROLL for (int i = 0; i < N; i++) {
				st(tk.state, MFT_QPRIOR + i, B, b, 0.5 * (P.model.q_lower[i] + P.model.q_upper[i]));
				st(tk.state, MFT_DQPRIOR + i, B, b, 0.0);
				st(tk.state, MFT_T2DIR + i, B, b, 1.0);
			}
			for (int k = 0; k < MFT_ISTATE_ROWS; k++) sti(tk.istate, k, B, b, 0);
		} else {
			real cur[N];
			mv<N, N>(tk.S, q, cur);
			UNROLL for (int i = 0; i < N; i++)
				if (i < tk.k0) {
					st(tk.goals, i, B, b, cur[i]);
					st(tk.goals, tk.k0 + i, B, b, 0.0);
					st(tk.goals, 2 * tk.k0 + i, B, b, 0.0);
					st(tk.state, i, B, b, 0.0);
				}
		}
	}
}

}  // namespace sai2b

static void launch_fast(int fast, int baked, dim3 grid, dim3 block, hipStream_t stream, const sai2b::DevParams* d_params,
						int with_comp, int* fb_counts, int* fb_list, int parity) {
	if (fast >= 3) {  // general hierarchies (sai2b_cert.hip); fast - 3 = most rows of a partial task
		sai2b_launch_tick_cert(d_params, (int)grid.x * 64, fast - 3, with_comp, fb_counts, fb_list, parity, stream);
		return;
	}
#if SAI2B_N == 7
	if (fast == 2 && baked)
		hipLaunchKernelGGL((sai2b::tick_fast_kernel<2, true>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
	else if (fast == 2)
		hipLaunchKernelGGL((sai2b::tick_fast_kernel<2, false>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
	else if (baked)
		hipLaunchKernelGGL((sai2b::tick_fast_kernel<1, true>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
	else
		hipLaunchKernelGGL((sai2b::tick_fast_kernel<1, false>), grid, block, 0, stream, d_params, with_comp, fb_counts, fb_list, parity);
#endif
}

// with_comp: bit 0 = JointTask compensation, bit 1 (tick_cert_kernel only) = no in-lane singular handling
extern "C" int sai2b_launch_tick(const sai2b::DevParams* d_params, int B, int debug, int fast, int baked, int commit_sh,
								 int with_comp_bits, int do_torque, int* fb_counts, int* fb_list, int parity, int group, hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	const int with_comp = with_comp_bits & 1;
	// the fast path produces torques only: introspection and model-only passes use the generic kernel
	if (debug) {
		hipLaunchKernelGGL((sai2b::tick_kernel<true>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque, nullptr, nullptr);
	} else if (fast != 0 && do_torque && commit_sh) {
		launch_fast(fast, baked, grid, block, stream, d_params, fast >= 3 ? with_comp_bits : with_comp, fb_counts, fb_list, parity);
		if (group)
			return sai2b_launch_tick_group(d_params, B, group, 0, commit_sh, with_comp, do_torque, (const int*)(fb_counts + parity),
										   (const int*)fb_list, stream);
		hipLaunchKernelGGL((sai2b::tick_kernel<false>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque,
						   (const int*)(fb_counts + parity), (const int*)fb_list);
	} else {
		if (group) return sai2b_launch_tick_group(d_params, B, group, 0, commit_sh, with_comp, do_torque, nullptr, nullptr, stream);
		hipLaunchKernelGGL((sai2b::tick_kernel<false>), grid, block, 0, stream, d_params, commit_sh, with_comp, do_torque, nullptr, nullptr);
	}
	return (int)hipGetLastError();
}

// the range pass of hierarchies with gated generators (same DEBUG variant as the tick that follows, so that
// both take the same range decisions)
extern "C" int sai2b_launch_range_pass(const sai2b::DevParams* d_params, int B, int debug, int with_comp, int group, hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	if (!debug && group) return sai2b_launch_tick_group(d_params, B, group, 1, 0, with_comp, 0, nullptr, nullptr, stream);
	if (debug)
		hipLaunchKernelGGL((sai2b::tick_kernel<true, true>), grid, block, 0, stream, d_params, 0, with_comp, 0, nullptr, nullptr);
	else
		hipLaunchKernelGGL((sai2b::tick_kernel<false, true>), grid, block, 0, stream, d_params, 0, with_comp, 0, nullptr, nullptr);
	return (int)hipGetLastError();
}

extern "C" int sai2b_launch_tick_part(const sai2b::DevParams* d_params, int B, int debug, int fast, int baked, int part, int with_comp_bits,
									  int* fb_counts, int* fb_list, int parity, int group, hipStream_t stream) {
	const dim3 grid((B + 63) / 64), block(64);
	if (debug)
		hipLaunchKernelGGL((sai2b::tick_kernel<true>), grid, block, 0, stream, d_params, 1, 1, 1, nullptr, nullptr);
	else if (fast == 0 && group)
		return sai2b_launch_tick_group(d_params, B, group, 0, 1, 1, 1, nullptr, nullptr, stream);
	else if (fast == 0)
		hipLaunchKernelGGL((sai2b::tick_kernel<false>), grid, block, 0, stream, d_params, 1, 1, 1, nullptr, nullptr);
	else if (part == 1 && group)
		return sai2b_launch_tick_group(d_params, B, group, 0, 1, 1, 1, (const int*)(fb_counts + parity), (const int*)fb_list, stream);
	else if (part == 1)
		hipLaunchKernelGGL((sai2b::tick_kernel<false>), grid, block, 0, stream, d_params, 1, 1, 1, (const int*)(fb_counts + parity),
						   (const int*)fb_list);
	else
		launch_fast(fast, baked, grid, block, stream, d_params, fast >= 3 ? with_comp_bits : 1, fb_counts, fb_list, parity);
	return (int)hipGetLastError();
}

extern "C" int sai2b_launch_reinit(const sai2b::DevParams* d_params, int B, int only_task, hipStream_t stream) {
	const int blocks = (B + 63) / 64;
	hipLaunchKernelGGL(sai2b::reinit_kernel, dim3(blocks), dim3(64), 0, stream, d_params, only_task);
	return (int)hipGetLastError();
}

extern "C" int sai2b_launch_task(const sai2b::DevParams* d_params, int B, int task, const double* Nprec_in, const double* tau_prec,
								 double* tau_out, double* N_out, double* Ntot_out, int commit_sh, int do_torque, const int* tk_count,
								 const int* tk_list, hipStream_t stream) {
	hipLaunchKernelGGL(sai2b::task_kernel, dim3((B + 63) / 64), dim3(64), 0, stream, d_params, task, Nprec_in, tau_prec, tau_out, N_out,
					   Ntot_out, commit_sh, do_torque, tk_count, tk_list);
	return (int)hipGetLastError();
}
