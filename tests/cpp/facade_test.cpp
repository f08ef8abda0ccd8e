// Exercises include/Sai2PrimitivesBatched.h the way examples/05-using_robot_controller.cpp:103-196
// uses the reference classes. Modes:
//   facade_test validate        (no GPU) argument checks throw std::invalid_argument as the reference's
//   facade_test tick <B> <in>   (GPU)    reads q,dq,goals (raw doubles) from <in>, prints torques
//   facade_test example04 <B> <in> <ticks> / example01 <B> <in> <ticks>   (GPU) the reference's examples 04 and 01,
//                               tasks driven through the TemplateTask virtuals with no RobotController
//   facade_test example02 <B> <in> <ticks>   (GPU) example 02: JointTask with the acceleration-limited internal OTG
//   facade_test example03 <B> <in> <ticks>   (GPU) example 03: MotionForceTask with the Cartesian internal OTG
//   facade_test example05 <B> <in> <ticks>   (GPU) example 05: MotionForceTask + JointTask through a RobotController
//   facade_test example07 <B> <in> <ticks>   (GPU) example 07: surface alignment, force + moment control in the compliant frame
//   facade_test example08 <B> <in> <ticks>   (GPU) example 08: partial MotionForceTask (y, z, rotation about x) + JointTask
//   facade_test example10 <B> <in> <ticks>   (GPU) example 10: orientation-only MotionForceTask + JointTask
//   facade_test example09 <B> <in> <ticks>   (GPU) example 09: position control until contact, then force control with POPC
//   facade_test example18 <B> <in> <ticks>   (GPU) example 18: the Panda driven into its singularities
//   facade_test example19 <B> <urdf> <in> <ticks>   (GPU) example 19: a 6R arm started in its wrist singularity
//   facade_test example11 <B> <urdf> <in> <ticks>   (GPU) example 11: the planar 4R from its URDF, RobotController
//   facade_test example06 <B> <urdf> <in> <ticks>   (GPU) example 06: the 8-joint sliding-base Panda from its URDF
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "Sai2PrimitivesBatched.h"

using namespace Sai2Primitives;

template <class F>
static bool throws_invalid(F f, const char* needle) {
	try {
		f();
	} catch (const std::invalid_argument& e) {
		return std::strstr(e.what(), needle) != nullptr;
	} catch (...) {
		return false;
	}
	return false;
}

static int validate() {
	auto robot = std::make_shared<BatchedRobotModel>(4);
	const double pos[3] = {0, 0, 0.22};
	int fails = 0;
	auto expect = [&](bool ok, const char* what) {
		if (!ok) {
			std::printf("FAIL %s\n", what);
			fails++;
		}
	};
	expect(throws_invalid([&] {
			   std::vector<std::shared_ptr<TemplateTask>> tasks;
			   RobotController c(robot, tasks);
		   }, "at least one task"), "empty task list");
	expect(throws_invalid([&] {
			   std::vector<std::shared_ptr<TemplateTask>> tasks = {std::make_shared<JointTask>(robot, "a"), std::make_shared<JointTask>(robot, "b")};
			   RobotController c(robot, tasks);
		   }, "nullspace of a full joint task"), "task after full joint task");
	expect(throws_invalid([&] {
			   std::vector<std::shared_ptr<TemplateTask>> tasks = {std::make_shared<MotionForceTask>(robot, 6, pos, nullptr, "a"),
																   std::make_shared<JointTask>(robot, "a")};
			   RobotController c(robot, tasks);
		   }, "unique names"), "duplicate names");
	expect(throws_invalid([&] {
			   auto other = std::make_shared<BatchedRobotModel>(4);
			   std::vector<std::shared_ptr<TemplateTask>> tasks = {std::make_shared<JointTask>(other, "a")};
			   RobotController c(robot, tasks);
		   }, "same robot model"), "different robot model");
	expect(throws_invalid([&] {
			   std::vector<double> sel(14, 0.0);
			   sel[0] = 1;
			   sel[7] = 2;	// rows are parallel
			   JointTask t(robot, sel, 2);
		   }, "not full rank"), "rank-deficient selection");
	expect(throws_invalid([&] { MotionForceTask t(robot, 6, std::vector<double>{}, std::vector<double>{}, pos); }, "cannot both be empty"),
		   "no controlled directions");
	expect(throws_invalid([&] {
			   JointTask t(robot);
			   t.setGains(-1, 1);
		   }, "positive or zero"), "negative gains");
	expect(throws_invalid([&] {
			   JointTask t(robot);
			   t.setGoalPosition(Batch(3, 0.0));
		   }, "size not consistent"), "goal size");
	{	// accessors of the reference that need no device (JointTask.h:120,225-257,333-354; MotionForceTask.h:437-440,653-659)
		std::vector<double> sel(14, 0.0);
		sel[1] = 1, sel[7 + 4] = 1;
		JointTask t(robot, sel, 2);
		t.setGains(30.0, 8.0, 1.0);
		expect(t.getGains().size() == 1 && t.getGains()[0].kp == 30.0 && t.getGains()[0].ki == 1.0, "isotropic gains come back as one entry");
		t.setGains(std::vector<double>{10, 20}, std::vector<double>{1, 2});
		expect(t.getGains().size() == 2 && t.getGains()[1].kp == 20.0 && t.getGains()[1].ki == 0.0, "one gain per task coordinate");
		expect(throws_invalid([&] { t.setGains(std::vector<double>{1, 2, 3}, std::vector<double>{1, 2, 3}); }, "inconsistent with number of task dofs"), "gain vector size");
		expect(throws_invalid([&] { t.setGains(std::vector<double>{-1, -2}, std::vector<double>{1, 2}); }, "positive or zero"), "all-negative gain vector");
		t.setGainsUnsafe(std::vector<double>{-1, 5}, std::vector<double>{1, 2}, std::vector<double>{0, 0});
		expect(t.getGains()[0].kp == -1.0, "unsafe gains skip the sign check");
		expect(t.getJointSelectionMatrix() == sel, "joint selection matrix");
		t.enableVelocitySaturation(std::vector<double>{0.3, 0.6});
		expect(t.getVelocitySaturationMaxVelocity() == std::vector<double>({0.3, 0.6}), "per-coordinate saturation velocities");
		expect(throws_invalid([&] { t.enableVelocitySaturation(std::vector<double>{0.3, 0.0}); }, "must be positive"), "zero saturation velocity");
		expect(throws_invalid([&] { t.enableInternalOtgAccelerationLimited(std::vector<double>{1.0}, std::vector<double>{1.0, 1.0}); }, "does not match task size"), "otg limit vector size");
		MotionForceTask m(robot, 6, std::vector<double>{1, 0, 0, 0, 1, 0}, std::vector<double>{0, 0, 1}, pos);
		const std::vector<double> pp = m.posSelectionProjector(), po = m.oriSelectionProjector();
		expect(pp[0] == 1 && pp[4] == 1 && pp[8] == 0 && po[8] == 1 && po[0] == 0, "selection projectors of a partial task");
		{	// sensor frame given in the link (MotionForceTask.cpp:794-803): T_control_to_sensor = compliant_frame^-1 * T
			const double c = std::cos(0.3), sn = std::sin(0.3);
			const double frot[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1}, fpos[3] = {0.01, 0.02, 0.2};
			MotionForceTask ms(robot, 6, fpos, frot);
			const double spos[3] = {0.03, -0.01, 0.25};
			ms.setForceSensorFrame(6, spos, nullptr);
			const sai2b_task_config& k = ms.config();
			const double d[3] = {0.02, -0.03, 0.05};
			bool ok = std::fabs(k.sensor_pos[0] - (c * d[0] + sn * d[1])) < 1e-15 && std::fabs(k.sensor_pos[1] - (-sn * d[0] + c * d[1])) < 1e-15 &&
					  std::fabs(k.sensor_pos[2] - d[2]) < 1e-15 && std::fabs(k.sensor_rot[0] - c) < 1e-15 && std::fabs(k.sensor_rot[1] - sn) < 1e-15 &&
					  std::fabs(k.sensor_rot[3] + sn) < 1e-15;
			expect(ok, "sensor frame relative to the control frame");
			expect(throws_invalid([&] { ms.setForceSensorFrame(5, spos, nullptr); }, "same as the link"), "sensor on another link");
		}
		m.enableVelocitySaturation(0.25, 0.5);
		expect(m.getLinearSaturationVelocity() == 0.25 && m.getAngularSaturationVelocity() == 0.5, "saturation velocities");
	}
	{	// Sai2Model::setTRobotBase (examples/05-...cpp:69): part of the model the kernels see; a second call replaces the first
		auto r2 = std::make_shared<BatchedRobotModel>(2);
		const double c = std::cos(0.3), sn = std::sin(0.3);
		const double bp[3] = {0.4, -0.2, 0.35}, bR[9] = {c, -sn, 0, sn, c, 0, 0, 0, 1};
		const double z0 = r2->model().joint_xyz[0][2];
		r2->setTRobotBase(bp, bR);
		r2->setTRobotBase(bp, bR);
		const sai2b_robot_model& mm = r2->model();
		expect(std::fabs(mm.joint_xyz[0][0] - 0.4) < 1e-15 && std::fabs(mm.joint_xyz[0][2] - (0.35 + z0)) < 1e-15 &&
				   std::fabs(mm.joint_rpy[0][2] - 0.3) < 1e-15 && r2->TRobotBasePosition()[1] == -0.2 && r2->TRobotBaseRotation()[1] == -sn,
			   "base pose folded into the first joint");
		const double mirror[9] = {1, 0, 0, 0, 1, 0, 0, 0, -1};
		expect(throws_invalid([&] { r2->setTRobotBase(bp, mirror); }, "not a rotation matrix"), "reflection as base orientation");
	}
	std::printf(fails ? "validate: %d failures\n" : "validate: ok\n", fails);
	return fails;
}

static int tick(int B, const char* path) {
	std::ifstream f(path, std::ios::binary);
	auto rd = [&](size_t rows) {
		Batch b(rows * (size_t)B);
		f.read((char*)b.data(), b.size() * sizeof(double));
		return b;
	};
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(rd(7));
	robot->setDq(rd(7));
	robot->updateModel();
	const double pos[3] = {0, 0, 0.22};
	auto mft = std::make_shared<MotionForceTask>(robot, 6, pos);
	mft->disableInternalOtg();
	auto jt = std::make_shared<JointTask>(robot);
	jt->disableInternalOtg();
	std::vector<std::shared_ptr<TemplateTask>> tasks = {mft, jt};
	RobotController ctl(robot, tasks);
	mft->setGoalPosition(rd(3));
	mft->setGoalOrientation(rd(9));
	mft->setGoalLinearVelocity(rd(3));
	mft->setGoalAngularVelocity(rd(3));
	mft->setGoalLinearAcceleration(rd(3));
	mft->setGoalAngularAcceleration(rd(3));
	jt->setGoalPosition(rd(7));
	ctl.updateControllerTaskModels();
	Batch tau = ctl.computeControlTorques();
	std::fwrite(tau.data(), sizeof(double), tau.size(), stdout);
	// the examples' simulation step (examples/05-...cpp:215-236), state staying on the device
	BatchedSimulation sim(ctl, 0.001, 2);
	sim.integrate();
	Batch q1 = sim.getJointPositions(), dq1 = sim.getJointVelocities();
	std::fwrite(q1.data(), sizeof(double), q1.size(), stdout);
	std::fwrite(dq1.data(), sizeof(double), dq1.size(), stdout);
	return 0;
}

// The same workload through ShardedRobotController: `shards` contexts on device 0, each with a host thread of its own
// (on an 8-GPU node: devices = {} gives one shard per GPU). Prints the torques of two ticks.
static int sharded(int B, const char* path, int shards) {
	std::ifstream f(path, std::ios::binary);
	auto rd = [&](size_t rows) {
		Batch b(rows * (size_t)B);
		f.read((char*)b.data(), b.size() * sizeof(double));
		return b;
	};
	const Batch q = rd(7), dq = rd(7);
	// task configurations as the reference's constructors + setters leave them
	auto robot = std::make_shared<BatchedRobotModel>(B);
	const double pos[3] = {0, 0, 0.22};
	auto mft = std::make_shared<MotionForceTask>(robot, 6, pos);
	mft->disableInternalOtg();
	auto jt = std::make_shared<JointTask>(robot);
	jt->disableInternalOtg();
	ShardedRobotController ctl(robot->model(), {mft->config(), jt->config()}, B, std::vector<int>((size_t)shards, 0));
	if (ctl.shards() != shards || ctl.batch() != B) return 3;
	int covered = 0;
	for (int s = 0; s < shards; s++) {
		if (ctl.shardBounds(s).first != covered) return 4;
		covered = ctl.shardBounds(s).second;
	}
	if (covered != B) return 5;
	ctl.setState(q, dq);
	ctl.reinitializeTasks();
	const Batch gp = rd(3), gr = rd(9), gv = rd(3), gw = rd(3), ga = rd(3), gal = rd(3);
	ctl.setMotionForceTaskGoals(0, gp, gr, gv, gw, ga, gal);
	ctl.setJointTaskGoals(1, rd(7));
	ctl.updateControllerTaskModels();
	Batch tau = ctl.computeControlTorques();
	std::fwrite(tau.data(), sizeof(double), tau.size(), stdout);
	tau = ctl.tick();
	std::fwrite(tau.data(), sizeof(double), tau.size(), stdout);
	bool threw = false;
	try {
		ctl.setState(Batch(7 * (size_t)B - 1), {});
	} catch (const std::invalid_argument&) {
		threw = true;
	}
	return threw ? 0 : 6;
}

static Batch add(const Batch& a, const Batch& b) {
	Batch c(a.size());
	for (size_t i = 0; i < a.size(); i++) c[i] = a[i] + b[i];
	return c;
}

// examples/04-task_and_redundancy/04-task_and_redundancy.cpp:101-206 call for call (no RobotController: the two
// tasks are driven through the TemplateTask virtuals and the nullspace is chained by hand); the trajectory
// and the cycle at which the joint task wakes up are compressed so that `ticks` periods cover them.
// Prints the control torques of every period, then the final joint positions.
static int example04(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :104
	const double pos_in_link[3] = {0.0, 0.0, 0.22};	 // "end-effector" (0, 0, 0.07) seen from link7: + 0.15 (:111-113)
	auto motion_force_task = std::make_unique<MotionForceTask>(robot, 6, pos_in_link);	// :114-115
	motion_force_task->disableInternalOtg();											// :117
	const Batch initial_orientation = motion_force_task->getCurrentOrientation();		// :120
	const Batch initial_position = motion_force_task->getCurrentPosition();				// :121
	auto joint_task = std::make_unique<JointTask>(robot);								// :125 (default gains, OTG on)
	const Batch initial_q = robot->q();													// :128
	BatchedSimulation sim(*motion_force_task, 0.001, 1);
	const int wake = ticks / 3;
	for (int cycle = 0; cycle < ticks; cycle++) {
		const double time = 0.001 * cycle;
		robot->setQ(sim.getJointPositions());  // :139-141
		robot->setDq(sim.getJointVelocities());
		robot->updateModel();
		motion_force_task->updateTaskModel();							 // :144-146 N_prec = identity
		const Batch N_prec = motion_force_task->getTaskAndPreviousNullspace();	// :147
		joint_task->updateTaskModel(N_prec);							 // :152
		// orientation: oscillation around Y (:157-172), position: circle in the y-z plane (:175-185)
		const double w_ori = 2 * M_PI * 0.2, amp = M_PI / 8, ang = amp * std::sin(w_ori * time);
		const double c = std::cos(ang), s_ = std::sin(ang);
		const double Rt[9] = {c, 0, -s_, 0, 1, 0, s_, 0, c};  // R^T for a rotation by ang about Y
		Batch Rg(9 * (size_t)B), wg(3 * (size_t)B, 0.0), ag(3 * (size_t)B, 0.0), pg(3 * (size_t)B), vg(3 * (size_t)B), lg(3 * (size_t)B);
		const double r = 0.05, wc = 2 * M_PI * 0.33;
		const double dp[3] = {0.0, std::sin(wc * time), 1 - std::cos(wc * time)}, dv[3] = {0.0, std::cos(wc * time), std::sin(wc * time)},
					 da[3] = {0.0, -std::sin(wc * time), std::cos(wc * time)};
		for (int b = 0; b < B; b++) {
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += Rt[3 * i + k] * initial_orientation[(size_t)(3 * k + j) * B + b];
					Rg[(size_t)(3 * i + j) * B + b] = v;
				}
			wg[(size_t)1 * B + b] = amp * w_ori * std::cos(w_ori * time);
			ag[(size_t)1 * B + b] = amp * w_ori * w_ori * -std::sin(w_ori * time);
			for (int i = 0; i < 3; i++) {
				pg[(size_t)i * B + b] = initial_position[(size_t)i * B + b] + r * dp[i];
				vg[(size_t)i * B + b] = r * wc * dv[i];
				lg[(size_t)i * B + b] = r * wc * wc * da[i];
			}
		}
		motion_force_task->setGoalOrientation(Rg);
		motion_force_task->setGoalAngularVelocity(wg);
		motion_force_task->setGoalAngularAcceleration(ag);
		motion_force_task->setGoalPosition(pg);
		motion_force_task->setGoalLinearVelocity(vg);
		motion_force_task->setGoalLinearAcceleration(lg);
		Batch motion_force_task_torques = motion_force_task->computeTorques();	// :188
		Batch joint_task_torques = joint_task->computeTorques();				// :189
		if (cycle < wake) std::fill(joint_task_torques.begin(), joint_task_torques.end(), 0.0);	 // :193-195
		if (cycle == wake) {													// :196-201
			joint_task->reInitializeTask();
			Batch goal_joint_pos = initial_q;
			for (int b = 0; b < B; b++) goal_joint_pos[b] += 1.5;
			joint_task->setGoalPosition(goal_joint_pos);
		}
		const Batch control_torques = add(motion_force_task_torques, joint_task_torques);  // :206
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.setJointTorques(control_torques);
		sim.integrate();
	}
	const Batch q1 = sim.getJointPositions();
	std::fwrite(q1.data(), sizeof(double), q1.size(), stdout);
	return 0;
}

// examples/01-joint_control/01-joint_control.cpp:123-191 call for call (BASELINE config 1: one JointTask driven on
// its own), the schedule of goal steps / gain changes / velocity saturation compressed into `ticks` periods.
static int example01(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();											// :126
	auto joint_task = std::make_shared<JointTask>(robot);			// :131
	joint_task->setGains(100, 20);									// :133
	Batch goal_position = joint_task->getGoalPosition();			// :134
	joint_task->disableInternalOtg();								// :136
	BatchedSimulation sim(*joint_task, 0.001, 1);
	Batch N_prec(49 * (size_t)B, 0.0);
	for (int cycle = 0; cycle < ticks; cycle++) {
		robot->setQ(sim.getJointPositions());  // :148-150
		robot->setDq(sim.getJointVelocities());
		robot->updateModel();
		for (int i = 0; i < 7; i++)
			for (int b = 0; b < B; b++) N_prec[(size_t)(8 * i) * B + b] = 1.0;	 // :153 setIdentity
		joint_task->updateTaskModel(N_prec);									 // :154
		if (cycle % 30 == 5)													 // :158-161
			for (int b = 0; b < B; b++) goal_position[(size_t)2 * B + b] += 0.4, goal_position[(size_t)3 * B + b] -= 0.6;
		if (cycle % 30 == 20)													 // :162-165
			for (int b = 0; b < B; b++) goal_position[(size_t)2 * B + b] -= 0.4, goal_position[(size_t)3 * B + b] += 0.6;
		joint_task->setGoalPosition(goal_position);								 // :166
		if (cycle == 35) joint_task->setGains(100, 10);							 // :169-174
		if (cycle == 45) joint_task->enableVelocitySaturation(M_PI / 4);		 // :176-181
		if (cycle == 55) joint_task->setGains(100, 20);							 // :184-189
		const Batch joint_task_torques = joint_task->computeTorques();			 // :191
		std::fwrite(joint_task_torques.data(), sizeof(double), joint_task_torques.size(), stdout);
		sim.setJointTorques(joint_task_torques);
		sim.integrate();
	}
	const Batch q1 = sim.getJointPositions();
	std::fwrite(q1.data(), sizeof(double), q1.size(), stdout);
	return 0;
}

// examples/06-partial_joint_task/06-partial_joint_task.cpp:99-176 call for call: the 8-joint Panda on its prismatic
// base read from a URDF file, a partial JointTask on the slider and the last joint above a MotionForceTask, driven by a
// RobotController; the schedule of the slider's goal steps compressed into `ticks` periods. (The partial task keeps
// the reference's default internal OTG, whose next state the example feeds to the motion task.)
static int example06(int B, const char* urdf, const char* path, int ticks) {
	auto robot = std::make_shared<BatchedRobotModel>(std::string(urdf), B);
	const int dof = robot->dof();
	std::ifstream f(path, std::ios::binary);
	Batch q0((size_t)dof * B), dq0((size_t)dof * B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :102
	std::vector<double> joint_selection(2 * (size_t)dof, 0.0);	// :107-109
	joint_selection[0] = 1;
	joint_selection[(size_t)dof + 7] = 1;
	auto partial_joint_task = std::make_shared<JointTask>(robot, joint_selection, 2);  // :110-111
	const double pos_in_link[3] = {0.0, 0.0, 0.07};										// :116-117
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, std::string("end-effector"), pos_in_link);  // :118-119
	motion_force_task->disableInternalOtg();																	   // :120
	std::vector<std::shared_ptr<TemplateTask>> task_list = {partial_joint_task, motion_force_task};			   // :123-124
	RobotController robot_controller(robot, task_list);															   // :125-126
	Batch joint_goal_pos = partial_joint_task->getGoalPosition();  // :112 getCurrentPosition() at construction = the goal after it
	const Batch initial_position = motion_force_task->getCurrentPosition();	 // :121
	BatchedSimulation sim(robot_controller, 0.001, 2);						 // the example's 2 kHz simulation thread
	for (int cycle = 0; cycle < ticks; cycle++) {
		robot->setQ(sim.getJointPositions());  // :137-139
		robot->setDq(sim.getJointVelocities());
		robot->updateModel();
		robot_controller.updateControllerTaskModels();	// :142
		if (cycle % 40 == 10)							// :146-150
			for (int b = 0; b < B; b++) joint_goal_pos[b] -= 1.0;
		else if (cycle % 40 == 30)
			for (int b = 0; b < B; b++) joint_goal_pos[b] += 1.0;
		partial_joint_task->setGoalPosition(joint_goal_pos);  // :151
		const double time = 0.001 * cycle, w = 2.0 * M_PI * 0.3;
		const Batch dp = partial_joint_task->getDesiredPosition(), dv = partial_joint_task->getDesiredVelocity(),
					da = partial_joint_task->getDesiredAcceleration();
		Batch gp(3 * (size_t)B), gv(3 * (size_t)B), ga(3 * (size_t)B);
		for (int b = 0; b < B; b++) {  // :154-166
			gp[b] = initial_position[b] + 0.1 * std::sin(w * time);
			gp[(size_t)B + b] = dp[b];
			gp[2 * (size_t)B + b] = initial_position[2 * (size_t)B + b] + 0.1 * (1 - std::cos(w * time));
			gv[b] = 0.1 * w * std::cos(w * time), gv[(size_t)B + b] = dv[b], gv[2 * (size_t)B + b] = 0.1 * w * std::sin(w * time);
			ga[b] = -0.1 * w * w * std::sin(w * time), ga[(size_t)B + b] = da[b], ga[2 * (size_t)B + b] = 0.1 * w * w * std::cos(w * time);
		}
		motion_force_task->setGoalPosition(gp);	 // :167-169
		motion_force_task->setGoalLinearVelocity(gv);
		motion_force_task->setGoalLinearAcceleration(ga);
		const Batch control_torques = robot_controller.computeControlTorques();	 // :174
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	const Batch q1 = sim.getJointPositions();
	std::fwrite(q1.data(), sizeof(double), q1.size(), stdout);
	return 0;
}

// examples/18-panda_singularity/18-panda_singularity.cpp:104-228 call for call: a full MotionForceTask with velocity
// saturation and a JointTask behind it, nullspaces chained by hand, position goals 2 m outside the workspace in
// +x, +y, -y, +z (back to the start in between), so that the arm stretches into its elbow / wrist singularities and
// the SingularityHandler blends the type-1 / type-2 strategies in and out. The waits between goals are compressed
// to `ticks / 8` periods. Prints, per period: the state read from the simulation (q, dq) and the control torques.
static int example18(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :107
	const double pos_in_link[3] = {0.0, 0.0, 0.22};	 // "end-effector" (0, 0, 0.07) seen from link7 (:112-114)
	auto motion_force_task = std::make_unique<MotionForceTask>(robot, 6, pos_in_link);	// :117-118
	motion_force_task->disableInternalOtg();											// :129
	motion_force_task->enableVelocitySaturation();										// :130
	const Batch initial_position = motion_force_task->getCurrentPosition();				// :135
	auto joint_task = std::make_unique<JointTask>(robot);								// :139
	joint_task->setGains(100, 20);														// :140
	const Batch initial_q = robot->q();													// :143
	joint_task->setGoalPosition(initial_q);												// :144
	const double desired_offsets[8][3] = {{2, 0, 0}, {0, 0, 0}, {0, 2, 0}, {0, 0, 0}, {0, -2, 0}, {0, 0, 0}, {0, 0, 2}, {0, 0, 0}};  // :147-150
	const int wait = ticks / 8, max_cnt = 8;
	int cnt = 0, prev = -wait;
	BatchedSimulation sim(*motion_force_task, 0.001, 2);  // the example's 2 kHz simulation thread (:236-241)
	const Batch zero3(3 * (size_t)B, 0.0);
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :176-178
		robot->setDq(dq);
		robot->updateModel();
		motion_force_task->updateTaskModel();									// :181-185 N_prec = identity
		const Batch N_prec = motion_force_task->getTaskAndPreviousNullspace();	// :186
		joint_task->updateTaskModel(N_prec);									// :191
		if (cycle - prev >= wait) {												// :195-201
			Batch goal(3 * (size_t)B);
			for (int i = 0; i < 3; i++)
				for (int b = 0; b < B; b++) goal[(size_t)i * B + b] = initial_position[(size_t)i * B + b] + desired_offsets[cnt][i];
			motion_force_task->setGoalPosition(goal);
			cnt++;
			prev = cycle;
			if (cnt == max_cnt) cnt = max_cnt - 1;
		}
		motion_force_task->setGoalLinearVelocity(zero3);	  // :202-203
		motion_force_task->setGoalLinearAcceleration(zero3);
		const Batch motion_force_task_torques = motion_force_task->computeTorques();  // :206
		const Batch joint_task_torques = joint_task->computeTorques();				   // :207
		const Batch control_torques = add(motion_force_task_torques, joint_task_torques);  // :212
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.setJointTorques(control_torques);
		sim.integrate();
	}
	return 0;
}

// examples/02-joint_control_internal_otg/02-joint_control_internal_otg.cpp:118-179 call for call: one JointTask with
// the acceleration-limited internal OTG, goal steps every "second", limits raised after "5 seconds", jerk limits added
// after "10 seconds" and two more goal steps under them; with u = ticks/12 for the example's 1000 cycles: steps at u and
// 3u of every 4u, new limits at 5u, jerk limits at 10u, end at 12u. Prints, per period, the state read and the torques.
static int example02(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();													 // :118
	auto joint_task = std::make_shared<JointTask>(robot);					 // :123
	joint_task->setGains(100, 20);											 // :125
	Batch goal_position = joint_task->getGoalPosition();					 // :126
	joint_task->enableInternalOtgAccelerationLimited(M_PI / 3, M_PI);		 // :129
	BatchedSimulation sim(*joint_task, 0.001, 1);
	Batch N_prec(49 * (size_t)B, 0.0);
	const int u = ticks / 12, period = 4 * u;
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :141-143
		robot->setDq(dq);
		robot->updateModel();
		for (int i = 0; i < 7; i++)
			for (int b = 0; b < B; b++) N_prec[(size_t)(8 * i) * B + b] = 1.0;	 // :146
		joint_task->updateTaskModel(N_prec);									 // :147
		if (cycle % period == u)												 // :151-155
			for (int b = 0; b < B; b++) goal_position[(size_t)1 * B + b] -= 0.2, goal_position[(size_t)2 * B + b] += 0.4, goal_position[(size_t)3 * B + b] -= 0.6;
		if (cycle % period == 3 * u)											 // :156-160
			for (int b = 0; b < B; b++) goal_position[(size_t)1 * B + b] += 0.2, goal_position[(size_t)2 * B + b] -= 0.4, goal_position[(size_t)3 * B + b] += 0.6;
		joint_task->setGoalPosition(goal_position);								 // :161
		if (cycle == 5 * u) joint_task->enableInternalOtgAccelerationLimited(M_PI, 3 * M_PI);	 // :164-169
		if (cycle == 10 * u) joint_task->enableInternalOtgJerkLimited(M_PI, 3 * M_PI, 3 * M_PI);  // :171-176
		const Batch joint_task_torques = joint_task->computeTorques();	// :178
		std::fwrite(joint_task_torques.data(), sizeof(double), joint_task_torques.size(), stdout);
		sim.setJointTorques(joint_task_torques);
		sim.integrate();
	}
	return 0;
}

// examples/03-cartesian_motion_control/03-cartesian_motion_control.cpp:109-183 call for call: one MotionForceTask with the
// reference's default (acceleration-limited, Cartesian) internal OTG, goal steps of 0.1 m in z with a 45 degree turn
// about z, the generator switched off later on and switched back on with jerk limits; with u = ticks/30 for the example's
// 500 cycles: steps at u and 4u of every 6u, generator off at 13u, jerk-limited at 25u (the cycle of a goal step, as in
// the example: 12500 % 3000 == 500), one more step at 28u, end at 30u. Prints, per period, the state read and the torques.
static int example03(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();										// :112
	const double pos_in_link[3] = {0.07, 0.0, 0.15};			// "end-effector" (0.07, 0, 0) seen from link7 (:117-122)
	auto motion_force_task = std::make_unique<MotionForceTask>(robot, 6, pos_in_link);	// :123-125
	motion_force_task->setPosControlGains(100.0, 20.0);		// :128-129
	motion_force_task->setOriControlGains(100.0, 20.0);
	Batch goal_orientation = motion_force_task->getCurrentOrientation();  // :132-135
	Batch goal_position = motion_force_task->getCurrentPosition();
	BatchedSimulation sim(*motion_force_task, 0.001, 1);
	const double th = M_PI / 4.0, R[9] = {std::cos(th), std::sin(th), 0, -std::sin(th), std::cos(th), 0, 0, 0, 1};	// :156-158
	const int u = ticks / 30, period = 6 * u;
	auto turn = [&](bool transpose) {  // goal_orientation = R (or R^T) * goal_orientation
		Batch out(goal_orientation.size());
		for (int b = 0; b < B; b++)
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += (transpose ? R[3 * k + i] : R[3 * i + k]) * goal_orientation[(size_t)(3 * k + j) * B + b];
					out[(size_t)(3 * i + j) * B + b] = v;
				}
		goal_orientation = out;
	};
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :146-148
		robot->setDq(dq);
		robot->updateModel();
		motion_force_task->updateTaskModel();  // :151-152 N_prec = identity
		if (cycle % period == 4 * u) {	// :159-162
			for (int b = 0; b < B; b++) goal_position[(size_t)2 * B + b] += 0.1;
			turn(false);
		} else if (cycle % period == u) {  // :164-167
			for (int b = 0; b < B; b++) goal_position[(size_t)2 * B + b] -= 0.1;
			turn(true);
		}
		motion_force_task->setGoalPosition(goal_position);		 // :168-169
		motion_force_task->setGoalOrientation(goal_orientation);
		if (cycle == 13 * u) motion_force_task->disableInternalOtg();														// :172-174
		if (cycle == 25 * u) motion_force_task->enableInternalOtgJerkLimited(0.3, 1.0, 3.0, M_PI / 3, M_PI, 3 * M_PI);	// :177-180
		const Batch motion_force_task_torques = motion_force_task->computeTorques();  // :182-183
		std::fwrite(motion_force_task_torques.data(), sizeof(double), motion_force_task_torques.size(), stdout);
		sim.setJointTorques(motion_force_task_torques);
		sim.integrate();
	}
	return 0;
}

// examples/09-3d_position_force_controller/09-3d_position_force_controller.cpp:106-206 call for call on the Panda: a
// translation-only MotionForceTask + JointTask in a RobotController, the goal moving in x / y and sinking in z until
// the force sensor reports contact, then force control along z (goal -5 N, closed loop, passivity observer on). Two
// things differ from the example and are part of what a batch means: (1) the contact is a virtual spring under each
// robot's start height (2 kN/m, sensed in the control frame; the simulation here has no contact), (2) a task's
// configuration is one per controller, so the switch to force control happens when EVERY robot reports contact.
// Cycle numbers 1000 / 2000 / 3000 are compressed to ticks/6, ticks/3, ticks/2. Prints, per period: q, dq, the
// sensor reading handed to the task (3 rows), the force-control flag (1 row) and the torques.
static int example09(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :109
	const double pos_in_link[3] = {0.0, 0.0, 0.15};
	const std::vector<double> controlled_directions_translation = {1, 0, 0, 0, 1, 0, 0, 0, 1};	// :113-116
	const std::vector<double> controlled_directions_rotation = {};								// :117
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, 6, controlled_directions_translation, controlled_directions_rotation,
																pos_in_link);  // :118-120
	bool force_control = false;													   // :121
	const Batch initial_position = motion_force_task->getCurrentPosition();		   // :124
	Batch goal_position = initial_position;
	auto joint_task = std::make_shared<JointTask>(robot);										   // :128
	std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};	   // :131-132
	auto robot_controller = std::make_unique<RobotController>(robot, task_list);				   // :133-134
	BatchedSimulation sim(*robot_controller, 0.001, 1);
	const int k2000 = ticks / 3, k1000 = ticks / 6, k3000 = ticks / 2;
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :144-146
		robot->setDq(dq);
		robot->updateModel();
		robot_controller->updateControllerTaskModels();	 // :149
		// the force sensor: a spring 4 mm under the start height, read in the control frame
		const Batch x = motion_force_task->getCurrentPosition(), R = motion_force_task->getCurrentOrientation();
		Batch sensed_force(3 * (size_t)B), zero(3 * (size_t)B, 0.0);
		for (int b = 0; b < B; b++) {
			const double pen = (initial_position[(size_t)2 * B + b] - 0.004) - x[(size_t)2 * B + b];
			const double fz = pen > 0 ? -2000.0 * pen : 0.0;
			for (int i = 0; i < 3; i++) sensed_force[(size_t)i * B + b] = R[(size_t)(6 + i) * B + b] * fz;	 // R^T (0, 0, fz)
		}
		motion_force_task->updateSensedForceAndMoment(sensed_force, zero);	// :152-156
		std::fwrite(sensed_force.data(), sizeof(double), sensed_force.size(), stdout);
		if (cycle % k2000 == 0)	 // :162-168
			for (int b = 0; b < B; b++) goal_position[b] -= 0.07, goal_position[(size_t)B + b] -= 0.07;
		else if (cycle % k2000 == k1000)
			for (int b = 0; b < B; b++) goal_position[b] += 0.07, goal_position[(size_t)B + b] += 0.07;
		if (cycle > k2000 && cycle < k3000)	 // :169-171
			for (int b = 0; b < B; b++) goal_position[(size_t)2 * B + b] -= 0.00015;
		motion_force_task->setGoalPosition(goal_position);	// :172
		if (!force_control) {								// :174-183
			const Batch fw = motion_force_task->getSensedForceControlWorldFrame();
			bool all = true;
			for (int b = 0; b < B; b++) all = all && fw[(size_t)2 * B + b] <= -1.0;
			if (all) {
				force_control = true;
				const double unit_z[3] = {0, 0, 1};
				motion_force_task->parametrizeForceMotionSpaces(1, unit_z);
				Batch gf(3 * (size_t)B, 0.0);
				for (int b = 0; b < B; b++) gf[(size_t)2 * B + b] = -5.0;
				motion_force_task->setGoalForce(gf);
				motion_force_task->setClosedLoopForceControl();
				motion_force_task->enablePassivity();
			}
		}
		const Batch flag((size_t)B, force_control ? 1.0 : 0.0);
		std::fwrite(flag.data(), sizeof(double), flag.size(), stdout);
		const Batch control_torques = robot_controller->computeControlTorques();  // :201-204
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	return force_control ? 0 : 4;  // the scenario must have reached the force-control phase
}

// examples/05-using_robot_controller/05-using_robot_controller.cpp:103-196 call for call: the example04 trajectories
// driven through a RobotController (updateControllerTaskModels / computeControlTorques: the two calls run the fused
// tick, i.e. the SVD-free kernel of the headline benchmark), the joint goal stepped at cycle 5000 = ticks/2. Prints,
// per period, the state read and the torques.
static int example05(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :106
	const double pos_in_link[3] = {0.0, 0.0, 0.22};	 // "end-effector" (0, 0, 0.07) seen from link7 (:111-113)
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, 6, pos_in_link);	// :114-115
	motion_force_task->disableInternalOtg();											// :117
	const Batch initial_orientation = motion_force_task->getCurrentOrientation();		// :120-122
	const Batch initial_position = motion_force_task->getCurrentPosition();
	const Batch initial_q = robot->q();													// :123
	auto joint_task = std::make_shared<JointTask>(robot);								// :126
	std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};	 // :129-130
	auto robot_controller = std::make_unique<RobotController>(robot, task_list);			 // :131-132
	BatchedSimulation sim(*robot_controller, 0.001, 1);
	for (int cycle = 0; cycle < ticks; cycle++) {
		const double time = 0.001 * cycle;
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :143-145
		robot->setDq(dq);
		robot->updateModel();
		robot_controller->updateControllerTaskModels();	 // :148
		const double w_ori = 2 * M_PI * 0.2, amp = M_PI / 8, ang = amp * std::sin(w_ori * time);  // :153-160
		const double c = std::cos(ang), s_ = std::sin(ang);
		const double Rt[9] = {c, 0, -s_, 0, 1, 0, s_, 0, c};  // R^T for a rotation by ang about Y
		Batch Rg(9 * (size_t)B), wg(3 * (size_t)B, 0.0), ag(3 * (size_t)B, 0.0), pg(3 * (size_t)B), vg(3 * (size_t)B), lg(3 * (size_t)B);
		const double r = 0.05, wc = 2 * M_PI * 0.33;
		const double dp[3] = {0.0, std::sin(wc * time), 1 - std::cos(wc * time)}, dv[3] = {0.0, std::cos(wc * time), std::sin(wc * time)},
					 da[3] = {0.0, -std::sin(wc * time), std::cos(wc * time)};
		for (int b = 0; b < B; b++) {
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += Rt[3 * i + k] * initial_orientation[(size_t)(3 * k + j) * B + b];
					Rg[(size_t)(3 * i + j) * B + b] = v;
				}
			wg[(size_t)1 * B + b] = amp * w_ori * std::cos(w_ori * time);
			ag[(size_t)1 * B + b] = amp * w_ori * w_ori * -std::sin(w_ori * time);
			for (int i = 0; i < 3; i++) {
				pg[(size_t)i * B + b] = initial_position[(size_t)i * B + b] + r * dp[i];
				vg[(size_t)i * B + b] = r * wc * dv[i];
				lg[(size_t)i * B + b] = r * wc * wc * da[i];
			}
		}
		motion_force_task->setGoalOrientation(Rg);			 // :164-169
		motion_force_task->setGoalAngularVelocity(wg);
		motion_force_task->setGoalAngularAcceleration(ag);
		motion_force_task->setGoalPosition(pg);				 // :174-183
		motion_force_task->setGoalLinearVelocity(vg);
		motion_force_task->setGoalLinearAcceleration(lg);
		if (cycle == ticks / 2) {							 // :185-190
			Batch goal_joint_pos = initial_q;
			for (int b = 0; b < B; b++) goal_joint_pos[b] += 1.5;
			joint_task->setGoalPosition(goal_joint_pos);
		}
		const Batch control_torques = robot_controller->computeControlTorques();  // :195
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	return 0;
}

// examples/07-surface_surface_contact/07-surface_surface_contact.cpp:124-228 call for call on the Panda: one 6-DOF
// MotionForceTask parametrised in its COMPLIANT frame, passivity observer on, force sensor at the link origin
// (setForceSensorFrame(link, identity)), sinking until contact, then force control along the frame's z and moment
// control about its x and y (closed loop both, gains 0.7 / 5 / 1.5 and 0.7 / 4 / 1.5). As in example09 the contact
// is virtual (a stiff spring 0.5 mm under the start height plus a torsional spring that wants the frame's z vertical,
// read in the sensor frame) and the switch happens when every robot reports contact. Prints, per period: q, dq, the
// sensor readings (force 3, moment 3), the contact-control flag (1 row) and the torques.
static int example07(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :130
	const double pos_in_link[3] = {0.0, 0.0, 0.22};
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, 6, pos_in_link, nullptr, "surface_alignment_task", true);  // :134-136
	motion_force_task->enablePassivity();		// :137
	motion_force_task->disableInternalOtg();	// :138
	const double origin[3] = {0, 0, 0};
	motion_force_task->setForceSensorFrame(6, origin, nullptr);	 // :141
	const Batch initial_position = motion_force_task->getCurrentPosition();	 // :145
	Batch goal_position = initial_position;
	std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task};		 // :150-151
	auto robot_controller = std::make_unique<RobotController>(robot, task_list);	 // :152-153
	BatchedSimulation sim(*robot_controller, 0.001, 1);
	bool contact_control = false;  // GO_TO_CONTACT / CONTACT_CONTROL (:126)
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :164-166
		robot->setDq(dq);
		robot->updateModel();
		const Batch x = motion_force_task->getCurrentPosition(), R = motion_force_task->getCurrentOrientation();
		Batch sensed_force(3 * (size_t)B), sensed_moment(3 * (size_t)B);
		for (int b = 0; b < B; b++) {
			const double pen = (initial_position[(size_t)2 * B + b] - 0.0005) - x[(size_t)2 * B + b];
			const double fz = pen > 0 ? -20000.0 * pen : 0.0;
			// torsional spring: 2 Nm/rad towards the frame's z being vertical, only in contact; z_f = third column of R
			const double zf[3] = {R[(size_t)2 * B + b], R[(size_t)5 * B + b], R[(size_t)8 * B + b]};
			const double sgn = zf[2] < 0 ? -1.0 : 1.0, k = pen > 0 ? 2.0 : 0.0;
			const double mw[3] = {k * sgn * zf[1], -k * sgn * zf[0], 0.0};	// k (z_f x (+-z_world))
			for (int i = 0; i < 3; i++) {  // into the sensor frame (the link's axes = the control frame's): R^T w
				sensed_force[(size_t)i * B + b] = R[(size_t)(6 + i) * B + b] * fz;
				sensed_moment[(size_t)i * B + b] = R[(size_t)i * B + b] * mw[0] + R[(size_t)(3 + i) * B + b] * mw[1];
			}
		}
		motion_force_task->updateSensedForceAndMoment(sensed_force, sensed_moment);	 // :170-171
		std::fwrite(sensed_force.data(), sizeof(double), sensed_force.size(), stdout);
		std::fwrite(sensed_moment.data(), sizeof(double), sensed_moment.size(), stdout);
		robot_controller->updateControllerTaskModels();	 // :174
		if (!contact_control) {							 // :178-204
			for (int b = 0; b < B; b++) goal_position[(size_t)2 * B + b] -= 0.00003;
			motion_force_task->setGoalPosition(goal_position);
			const Batch fw = motion_force_task->getSensedForceControlWorldFrame();
			bool all = true;
			for (int b = 0; b < B; b++) all = all && fw[(size_t)2 * B + b] <= -1.0;
			if (all) {
				const double unit_z[3] = {0, 0, 1};
				motion_force_task->parametrizeForceMotionSpaces(1, unit_z);
				motion_force_task->parametrizeMomentRotMotionSpaces(2, unit_z);
				motion_force_task->setClosedLoopForceControl();
				motion_force_task->setClosedLoopMomentControl();
				Batch gf(3 * (size_t)B, 0.0), gm(3 * (size_t)B, 0.0);
				for (int b = 0; b < B; b++) gf[(size_t)2 * B + b] = 10.0;
				motion_force_task->setGoalForce(gf);
				motion_force_task->setGoalMoment(gm);
				motion_force_task->setForceControlGains(0.7, 5.0, 1.5);
				motion_force_task->setMomentControlGains(0.7, 4.0, 1.5);
				contact_control = true;
			}
		}
		const Batch flag((size_t)B, contact_control ? 1.0 : 0.0);
		std::fwrite(flag.data(), sizeof(double), flag.size(), stdout);
		const Batch control_torques = robot_controller->computeControlTorques();  // :225
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	return contact_control ? 0 : 4;
}

// examples/08-partial_motion_force_task/08-partial_motion_force_task.cpp:106-200 call for call: a partial
// MotionForceTask (translation along y and z, rotation about x: a projection that is NOT a leading block) without
// internal OTG or velocity saturation + a JointTask (internal OTG on) in a RobotController; goal steps the task cannot
// follow (x, rotation about z) and steps it can, then the first joint moved in the nullspace. Cycles 1000 / 2000 / 3000 /
// 4000 / 8000 are compressed to k ticks/9. Prints, per period, the state read and the torques.
static int example08(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :108
	const double pos_in_link[3] = {0.07, 0.0, 0.15};									 // "end-effector" (0.07, 0, 0) seen from link7 (:112-114)
	const std::vector<double> controlled_directions_translation = {0, 1, 0, 0, 0, 1};	 // :115-117
	const std::vector<double> controlled_directions_rotation = {1, 0, 0};				 // :118-119
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, 6, controlled_directions_translation, controlled_directions_rotation,
																pos_in_link);  // :120-122
	motion_force_task->disableInternalOtg();			 // :124
	motion_force_task->disableVelocitySaturation();		 // :125
	motion_force_task->setPosControlGains(100.0, 20.0);	 // :128-129
	motion_force_task->setOriControlGains(100.0, 20.0);
	const Batch initial_orientation = motion_force_task->getCurrentOrientation();  // :132-135
	const Batch initial_position = motion_force_task->getCurrentPosition();
	Batch goal_position = initial_position, goal_orientation = initial_orientation;
	auto joint_task = std::make_shared<JointTask>(robot);									   // :138
	std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};   // :141-142
	auto robot_controller = std::make_unique<RobotController>(robot, task_list);			   // :143-144
	joint_task->setGains(100.0, 20.0);														   // :145
	BatchedSimulation sim(*robot_controller, 0.001, 1);
	auto rotated = [&](int axis, double angle) {  // AngleAxis(angle, unit axis) * initial_orientation
		const double c = std::cos(angle), s_ = std::sin(angle);
		double A[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
		const int i = (axis + 1) % 3, j = (axis + 2) % 3;
		A[3 * i + i] = c, A[3 * i + j] = -s_, A[3 * j + i] = s_, A[3 * j + j] = c;
		Batch out(initial_orientation.size());
		for (int b = 0; b < B; b++)
			for (int r = 0; r < 3; r++)
				for (int col = 0; col < 3; col++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += A[3 * r + k] * initial_orientation[(size_t)(3 * k + col) * B + b];
					out[(size_t)(3 * r + col) * B + b] = v;
				}
		return out;
	};
	const int u = ticks / 9;
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :155-157
		robot->setDq(dq);
		robot->updateModel();
		robot_controller->updateControllerTaskModels();	 // :160
		if (cycle == 1 * u)								 // :167-169: x, not controlled
			for (int b = 0; b < B; b++) goal_position[b] += 0.1;
		if (cycle == 2 * u)								 // :171-173: y and z
			for (int b = 0; b < B; b++) goal_position[(size_t)B + b] += 0.1, goal_position[(size_t)2 * B + b] += 0.1;
		if (cycle == 3 * u) goal_orientation = rotated(2, M_PI / 6);  // :176-179: about z, not controlled
		if (cycle == 4 * u) goal_orientation = rotated(0, M_PI / 6);  // :181-184: about x
		if (cycle == 8 * u) {										  // :187-191
			Batch q_des = robot->q();
			for (int b = 0; b < B; b++) q_des[b] += 0.5;
			joint_task->setGoalPosition(q_des);
		}
		motion_force_task->setGoalPosition(goal_position);		 // :193-194
		motion_force_task->setGoalOrientation(goal_orientation);
		const Batch control_torques = robot_controller->computeControlTorques();  // :199
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	return 0;
}

// examples/10-3d_orientation_controller/10-3d_orientation_controller.cpp:101-166 call for call on the Panda: an
// orientation-only MotionForceTask (three rotation directions, no translation; internal OTG on) + JointTask in a
// RobotController, the goal orientation stepped between three attitudes; cycles 0 / 2000 / 4000 of 6000 are compressed
// to 0 / ticks/3 / 2 ticks/3. Prints, per period, the state read and the torques.
static int example10(int B, const char* path, int ticks) {
	std::ifstream f(path, std::ios::binary);
	Batch q0(7 * (size_t)B), dq0(7 * (size_t)B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	auto robot = std::make_shared<BatchedRobotModel>(B);
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :104
	const double pos_in_link[3] = {0.0, 0.0, 0.0};
	const std::vector<double> controlled_directions_translation = {};					  // :108
	const std::vector<double> controlled_directions_rotation = {1, 0, 0, 0, 1, 0, 0, 0, 1};	  // :109-112
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, 6, controlled_directions_translation, controlled_directions_rotation,
																pos_in_link);  // :113-115
	const Batch initial_orientation = motion_force_task->getCurrentOrientation();  // :118
	Batch goal_orientation = initial_orientation;
	auto joint_task = std::make_shared<JointTask>(robot);									   // :122
	std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};   // :123-124
	auto robot_controller = std::make_unique<RobotController>(robot, task_list);			   // :125-126
	BatchedSimulation sim(*robot_controller, 0.001, 1);
	auto premultiplied = [&](const double A[9]) {
		Batch out(initial_orientation.size());
		for (int b = 0; b < B; b++)
			for (int r = 0; r < 3; r++)
				for (int col = 0; col < 3; col++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += A[3 * r + k] * initial_orientation[(size_t)(3 * k + col) * B + b];
					out[(size_t)(3 * r + col) * B + b] = v;
				}
		return out;
	};
	const double cx = std::cos(M_PI / 3), sx = std::sin(M_PI / 3), cy = std::cos(M_PI / 4), sy = std::sin(M_PI / 4);
	const double Rx[9] = {1, 0, 0, 0, cx, -sx, 0, sx, cx};
	const double RyRx[9] = {cy, sy * sx, sy * cx, 0, cx, -sx, -sy, cy * sx, cy * cx};  // Ry(pi/4) Rx(pi/3)
	const int period = ticks;
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :136-138
		robot->setDq(dq);
		robot->updateModel();
		robot_controller->updateControllerTaskModels();	 // :141
		if (cycle % period == 0)						 // :147-158
			goal_orientation = initial_orientation;
		else if (cycle % period == period / 3)
			goal_orientation = premultiplied(Rx);
		else if (cycle % period == 2 * period / 3)
			goal_orientation = premultiplied(RyRx);
		motion_force_task->setGoalOrientation(goal_orientation);				  // :160
		const Batch control_torques = robot_controller->computeControlTorques();  // :165
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	return 0;
}

// examples/11-planar_robot_controller/11-planar_robot_controller.cpp:99-166 call for call: the planar 4R read from its
// URDF, a partial MotionForceTask (x, y, rotation about z) on "link4" given by name and a JointTask behind it in a
// RobotController, both with the reference's default internal OTG left on; the goal steps of cycles 0 / 2000 of
// 4000 compressed to 0 / ticks/2. Prints, per period, the state read (q, dq) and the control torques.
static int example11(int B, const char* urdf, const char* path, int ticks) {
	auto robot = std::make_shared<BatchedRobotModel>(std::string(urdf), B);
	const int dof = robot->dof();
	std::ifstream f(path, std::ios::binary);
	Batch q0((size_t)dof * B), dq0((size_t)dof * B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :101
	const double frame_pos[3] = {0.5, 0.0, 0.0};								  // :106-107
	const std::vector<double> controlled_directions_translation = {1, 0, 0, 0, 1, 0};	  // :108-110
	const std::vector<double> controlled_directions_rotation = {0, 0, 1};				  // :111-112
	auto motion_force_task = std::make_shared<MotionForceTask>(robot, std::string("link4"), controlled_directions_translation,
																controlled_directions_rotation, frame_pos);	 // :113-115
	const Batch initial_orientation = motion_force_task->getCurrentOrientation();  // :118-120
	const Batch initial_position = motion_force_task->getCurrentPosition();
	Batch goal_position = initial_position, goal_orientation = initial_orientation;
	auto joint_task = std::make_shared<JointTask>(robot);													 // :125
	std::vector<std::shared_ptr<TemplateTask>> task_list = {motion_force_task, joint_task};				 // :126-127
	auto robot_controller = std::make_unique<RobotController>(robot, task_list);							 // :128-129
	BatchedSimulation sim(*robot_controller, 0.001, 2);
	const double c = std::cos(-M_PI / 4), s_ = std::sin(-M_PI / 4);
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :139-141
		robot->setDq(dq);
		robot->updateModel();
		robot_controller->updateControllerTaskModels();	 // :144
		if (cycle % ticks == 0) {						 // :147-149
			goal_position = initial_position;
			goal_orientation = initial_orientation;
		} else if (cycle % ticks == ticks / 2) {  // :150-154: back by (0.25, 0.25, 0), turned by -pi/4 about z
			for (int b = 0; b < B; b++) {
				goal_position[(size_t)0 * B + b] = initial_position[(size_t)0 * B + b] - 0.25;
				goal_position[(size_t)1 * B + b] = initial_position[(size_t)1 * B + b] - 0.25;
				for (int j = 0; j < 3; j++) {
					const double r0 = initial_orientation[(size_t)j * B + b], r1 = initial_orientation[(size_t)(3 + j) * B + b];
					goal_orientation[(size_t)j * B + b] = c * r0 - s_ * r1;
					goal_orientation[(size_t)(3 + j) * B + b] = s_ * r0 + c * r1;
				}
			}
		}
		motion_force_task->setGoalPosition(goal_position);		 // :156-157
		motion_force_task->setGoalOrientation(goal_orientation);
		const Batch control_torques = robot_controller->computeControlTorques();  // :162
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.integrate();
	}
	return 0;
}

// examples/19-puma_singularity/19-puma_singularity.cpp:129-270 call for call (the wrist-lock variant) on a 6R arm read
// from a URDF (tests/robots.py: six_r, PUMA-like; the PUMA's own URDF is not part of the reference tree): a 6-DOF
// MotionForceTask (internal OTG on, singularity handling gains 50 / 20 / 20) started IN the wrist singularity and a
// JointTask behind it, nullspaces chained by hand; the state machine sends the frame 0.2 m down with a 90 degree turn
// about its z and back, every "5 seconds" = ticks/6 periods. Prints, per period, the state read and the torques.
static int example19(int B, const char* urdf, const char* path, int ticks) {
	auto robot = std::make_shared<BatchedRobotModel>(std::string(urdf), B);
	const int dof = robot->dof();
	std::ifstream f(path, std::ios::binary);
	Batch q0((size_t)dof * B), dq0((size_t)dof * B, 0.0);
	f.read((char*)q0.data(), q0.size() * sizeof(double));
	f.read((char*)dq0.data(), dq0.size() * sizeof(double));
	robot->setQ(q0);
	robot->setDq(dq0);
	robot->updateModel();  // :132
	const double pos_in_link[3] = {0.0, 0.0, 0.0};	// :139-140
	auto motion_force_task = std::make_unique<MotionForceTask>(robot, std::string("link6"), pos_in_link);  // :143-144
	motion_force_task->setSingularityHandlingGains(50, 20, 20);										   // :145
	auto joint_task = std::make_unique<JointTask>(robot);												   // :178
	joint_task->setGains(100, 20);																		   // :179
	const Batch initial_q = robot->q();																	   // :182
	joint_task->setGoalPosition(initial_q);																   // :183
	enum { GO_TO_SINGULARITY, EXIT_SINGULARITY } state = GO_TO_SINGULARITY;								   // :185
	int start = 0, cnt = 0;
	const int five_seconds = ticks / 6;
	Batch starting_ee_pos, starting_ee_ori;
	BatchedSimulation sim(*motion_force_task, 0.001, 1);
	auto away = [&]() {	 // starting pose + (0, 0, -0.2), turned by 90 degrees about its own z (:214-217)
		Batch p = starting_ee_pos, R(starting_ee_ori.size());
		for (int b = 0; b < B; b++) {
			p[(size_t)2 * B + b] -= 0.2;
			for (int i = 0; i < 3; i++) {  // R * Rz(90): columns (y, -x, z) of R
				R[(size_t)(3 * i + 0) * B + b] = starting_ee_ori[(size_t)(3 * i + 1) * B + b];
				R[(size_t)(3 * i + 1) * B + b] = -starting_ee_ori[(size_t)(3 * i + 0) * B + b];
				R[(size_t)(3 * i + 2) * B + b] = starting_ee_ori[(size_t)(3 * i + 2) * B + b];
			}
		}
		motion_force_task->setGoalPosition(p);
		motion_force_task->setGoalOrientation(R);
	};
	for (int cycle = 0; cycle < ticks; cycle++) {
		const Batch q = sim.getJointPositions(), dq = sim.getJointVelocities();
		std::fwrite(q.data(), sizeof(double), q.size(), stdout);
		std::fwrite(dq.data(), sizeof(double), dq.size(), stdout);
		robot->setQ(q);	 // :199-201
		robot->setDq(dq);
		robot->updateModel();
		if (cycle - start > five_seconds && state == GO_TO_SINGULARITY) {  // :209-221
			starting_ee_pos = motion_force_task->getCurrentPosition();	   // (robot->position / rotation of the frame, :204-206)
			starting_ee_ori = motion_force_task->getCurrentOrientation();
			away();
			state = EXIT_SINGULARITY;
			start = cycle;
		}
		if (cycle - start > five_seconds && state == EXIT_SINGULARITY) {  // :223-244
			if (cnt == 0 || cnt == 2) {
				motion_force_task->setGoalPosition(starting_ee_pos);
				motion_force_task->setGoalOrientation(starting_ee_ori);
			} else {
				away();
			}
			start = cycle;
			cnt = (cnt + 1) % 4;
		}
		motion_force_task->updateTaskModel();									// :250-255 N_prec = identity
		const Batch N_prec = motion_force_task->getTaskAndPreviousNullspace();	// :256
		joint_task->updateTaskModel(N_prec);									// :261
		const Batch motion_force_task_torques = motion_force_task->computeTorques();  // :264
		const Batch joint_task_torques = joint_task->computeTorques();				   // :265
		const Batch control_torques = add(motion_force_task_torques, joint_task_torques);  // :270
		std::fwrite(control_torques.data(), sizeof(double), control_torques.size(), stdout);
		sim.setJointTorques(control_torques);
		sim.integrate();
	}
	return 0;
}

int main(int argc, char** argv) {
	try {
		if (argc >= 6 && std::strcmp(argv[1], "example06") == 0) return example06(std::atoi(argv[2]), argv[3], argv[4], std::atoi(argv[5]));
		if (argc >= 5 && std::strcmp(argv[1], "example04") == 0) return example04(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 6 && std::strcmp(argv[1], "example19") == 0) return example19(std::atoi(argv[2]), argv[3], argv[4], std::atoi(argv[5]));
		if (argc >= 6 && std::strcmp(argv[1], "example11") == 0) return example11(std::atoi(argv[2]), argv[3], argv[4], std::atoi(argv[5]));
		if (argc >= 5 && std::strcmp(argv[1], "example02") == 0) return example02(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example03") == 0) return example03(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example05") == 0) return example05(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example07") == 0) return example07(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example08") == 0) return example08(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example10") == 0) return example10(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example09") == 0) return example09(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example18") == 0) return example18(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 5 && std::strcmp(argv[1], "example01") == 0) return example01(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
		if (argc >= 2 && std::strcmp(argv[1], "validate") == 0) return validate();
		if (argc >= 4 && std::strcmp(argv[1], "tick") == 0) return tick(std::atoi(argv[2]), argv[3]);
		if (argc >= 5 && std::strcmp(argv[1], "sharded") == 0) return sharded(std::atoi(argv[2]), argv[3], std::atoi(argv[4]));
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 2;
	}
	std::fprintf(stderr, "usage: facade_test validate | tick <B> <input>\n");
	return 64;
}
