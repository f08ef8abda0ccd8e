// sai2b_launch.h — launch entry points shared by sai2b_kernels.hip and sai2b_host.cpp
#pragma once
#include <hip/hip_runtime.h>

#include "sai2b_params.h"

// group: lanes per robot of the generic kernel (16 / 8), or 0 = the one-lane-per-robot generic kernel
extern "C" int sai2b_launch_tick(const sai2b::DevParams* d_params, int B, int debug, int fast, int baked, int commit_sh,
								 int with_comp, int do_torque, int* fb_counts, int* fb_list, int parity, int group, hipStream_t stream);
// generic tick with a robot spread over `lanes` = 16 or 8 lanes (sai2b_group.hip); fb_count / fb_list as tick_kernel
extern "C" int sai2b_launch_tick_group(const sai2b::DevParams* d_params, int B, int lanes, int range_only, int commit_sh, int with_comp,
									   int do_torque, const int* fb_count, const int* fb_list, hipStream_t stream);
// the SVD-free tick for general hierarchies (sai2b_cert.hip): fills the work list like the fast kernels.
// max_rows: most rows of a partial task of the hierarchy (selects the instantiation)
extern "C" int sai2b_launch_tick_cert(const sai2b::DevParams* d_params, int B, int max_rows, int with_comp, int* fb_counts, int* fb_list,
									  int parity, hipStream_t stream);
// the range pass ahead of the trajectory generators for certified robots (sai2b_cert.hip: range_cert_kernel); the rest
// of the batch goes to rg_list for sai2b_launch_tick_group(..., range_only = 1, ...)
extern "C" int sai2b_launch_range_cert(const sai2b::DevParams* d_params, int B, int max_rows, int* rg_counts, int* rg_list, int parity, int inlane,
									   hipStream_t stream);
extern "C" int sai2b_launch_range_pass(const sai2b::DevParams* d_params, int B, int debug, int with_comp, int group, hipStream_t stream);
// only_task < 0: every task (RobotController::reinitializeTasks); else TemplateTask::reInitializeTask of that one
extern "C" int sai2b_launch_reinit(const sai2b::DevParams* d_params, int B, int only_task, hipStream_t stream);
// one task on its own (TemplateTask.h:42-88): model update (do_torque = 0) or the task's torques (do_torque = 1) under
// a caller-supplied N_prec ([49][B], NULL = identity) and tau_prec ([7][B], NULL = the no-argument computeTorques());
// N_out / Ntot_out [49][B]: the task's nullspace and N * N_prec; tau_out [7][B]
extern "C" int sai2b_launch_task_group(const sai2b::DevParams* d_params, int B, int lanes, int task, const double* Nprec_in, const double* tau_prec,
									   double* tau_out, double* N_out, double* Ntot_out, int commit_sh, int do_torque, const int* tk_count,
									   const int* tk_list, hipStream_t stream);
extern "C" int sai2b_launch_task(const sai2b::DevParams* d_params, int B, int task, const double* Nprec_in, const double* tau_prec,
								 double* tau_out, double* N_out, double* Ntot_out, int commit_sh, int do_torque, const int* tk_count,
								 const int* tk_list, hipStream_t stream);
// the same calls through the whitened cascade (sai2b_cert.hip: task_cert_kernel); robots it declines are appended to
// tk_list (tk_counts: two counters, zero before the first launch, `parity` alternating between launches) for
// sai2b_launch_task(..., tk_counts + parity, tk_list) behind it. max_rows: rows of the task (<= 3: the small instantiation)
extern "C" int sai2b_launch_task_cert(const sai2b::DevParams* d_params, int B, int task, int max_rows, const double* Nprec_in,
									  const double* tau_prec, double* tau_out, double* N_out, double* Ntot_out, int do_torque, int* tk_counts,
									  int* tk_list, int parity, hipStream_t stream);
// one kernel of a (fast) tick on its own, for per-kernel timing: part 0 = first kernel, part 1 = the
// generic kernel over the work list of the SVD-free one. fb_counts: 2 ints, zero before the first
// launch; fb_list: B ints; parity alternates 0/1 between consecutive launches of the SVD-free kernel
extern "C" int sai2b_launch_tick_part(const sai2b::DevParams* d_params, int B, int debug, int fast, int baked, int part, int with_comp_bits,
									  int* fb_counts, int* fb_list, int parity, int group, hipStream_t stream);
// internal OTG (sai2b_otg.hip): one update of every enabled generator; (re)initialisation (modes in the kernel's comment)
// task_mask bit t: advance task t's generator (all enabled ones: ~0)
extern "C" int sai2b_launch_otg(const sai2b::DevParams* d_params, int B, int* counts, int* list, int parity, int clean_mask,
								int task_mask, int jerk_mask, hipStream_t stream);
// q_pose: [7][B] joint positions the tasks' cached poses correspond to (read in mode 1 only)
extern "C" int sai2b_launch_otg_reinit(const sai2b::DevParams* d_params, int B, int only_task, int mode, const double* q_pose,
									   hipStream_t stream);
// force / motion space re-parametrisation of MotionForceTask `task` at run time (flags in the kernel's comment)
extern "C" int sai2b_launch_mft_reparam(const sai2b::DevParams* d_params, int B, int task, int flags, const double* q_pose,
										hipStream_t stream);
// simulation harness (sai2b_sim.hip): one control period of rigid-body dynamics, state updated in place
extern "C" int sai2b_launch_sim(const sai2b::DevParams* d_params, int B, const double* tau, double dt, int substeps,
								int with_gravity, double* dbg_bias, double* q_keep, hipStream_t stream);
// observers of a MotionForceTask between ticks: out [68][B] (rows in sai2b_sim.hip: mft_status_kernel)
extern "C" int sai2b_launch_mft_status(const sai2b::DevParams* d_params, int B, int task, double* out, hipStream_t stream);
