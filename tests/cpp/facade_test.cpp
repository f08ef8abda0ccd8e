// Exercises include/Sai2PrimitivesBatched.h the way examples/05-using_robot_controller.cpp:103-196
// uses the reference classes. Modes:
//   facade_test validate        (no GPU) argument checks throw std::invalid_argument as the reference's
//   facade_test tick <B> <in>   (GPU)    reads q,dq,goals (raw doubles) from <in>, prints torques
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

int main(int argc, char** argv) {
	try {
		if (argc >= 2 && std::strcmp(argv[1], "validate") == 0) return validate();
		if (argc >= 4 && std::strcmp(argv[1], "tick") == 0) return tick(std::atoi(argv[2]), argv[3]);
	} catch (const std::exception& e) {
		std::fprintf(stderr, "exception: %s\n", e.what());
		return 2;
	}
	std::fprintf(stderr, "usage: facade_test validate | tick <B> <input>\n");
	return 64;
}
