// The reference's example 05 (examples/05-using_robot_controller/05-using_robot_controller.cpp:103-196) with ITS OWN
// types and calls — shared_ptr<Sai2Model::Sai2Model>, Affine3d, Vector3d, Matrix3d, VectorXd, make_shared<Sai2Primitives::
// MotionForceTask>(robot, link_name, compliant_frame) ... — compiled against include/Sai2PrimitivesEigen.h: what a program
// written for the reference looks like when it only swaps the include. (Eigen itself is not in this image: the test
// compiles against tests/cpp/mini_eigen, a test double of the handful of Eigen operations used here; with the real
// Eigen on the include path nothing else changes.) The simulation of the example is Sai2PrimitivesBatched's.
//   eigen_adapter_test <urdf> <q0 file> <ticks> [manual|base]   prints, per period: q, dq read from the simulation and
//   the torques; `manual`: example 04's hand-chained hierarchy instead of the RobotController; `base`: example 05 with a
//   robot base away from the world's origin (setTRobotBase)
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>

// the reference example's own include lines (05-using_robot_controller.cpp:8-13), resolved by include/sai2_compat
#include "RobotController.h"
#include "Sai2Model.h"
#include "tasks/JointTask.h"
#include "tasks/MotionForceTask.h"

using namespace std;
using namespace Eigen;

int main(int argc, char** argv) {
	if (argc < 4) return 2;
	const string robot_file = argv[1];
	const int ticks = atoi(argv[3]);
	try {
		auto robot = make_shared<Sai2Model::Sai2Model>(robot_file, false);	// :96
		const int dof = robot->dof();
		VectorXd q0(dof);
		{
			ifstream f(argv[2], ios::binary);
			vector<double> buf((size_t)dof);
			f.read((char*)buf.data(), buf.size() * sizeof(double));
			for (int i = 0; i < dof; i++) q0(i) = buf[(size_t)i];
		}
		const bool with_base = argc > 4 && string(argv[4]) == "base";
		if (with_base) {
			// :69 robot->setTRobotBase(sim->getRobotBaseTransform(robot_name)) — here with a base that is NOT at the
			// world's origin: goals, poses and Jacobians of the MotionForceTask are world quantities
			Affine3d T_world_base = Affine3d(Translation3d(Vector3d(0.4, -0.2, 0.35)));
			T_world_base.linear() = AngleAxisd(0.6, Vector3d(1.0, 2.0, 3.0) * (1.0 / sqrt(14.0))).toRotationMatrix();
			robot->setTRobotBase(T_world_base);
		}
		robot->setQ(q0);	   // :97 (from the simulation's initial state)
		robot->updateModel();  // :106

		// Position plus orientation task (:109-117)
		string link_name = "end-effector";
		Vector3d pos_in_link = Vector3d(0.0, 0.0, 0.07);
		Affine3d compliant_frame = Affine3d(Translation3d(pos_in_link));
		auto motion_force_task = make_shared<Sai2Primitives::MotionForceTask>(robot, link_name, compliant_frame);
		motion_force_task->disableInternalOtg();

		// no gains setting here, using the default task values (:119-123)
		const Matrix3d initial_orientation = robot->rotationInWorld(link_name);
		const Vector3d initial_position = robot->positionInWorld(link_name, pos_in_link);
		const VectorXd initial_q = robot->q();

		if (with_base) {  // the plain getters stay in the base frame, the ...InWorld ones include TRobotBase()
			const Affine3d T = robot->TRobotBase();
			const Vector3d p_world = T.linear() * robot->position(link_name, pos_in_link) + T.translation();
			if ((p_world - initial_position).norm() > 1e-12) return 7;
			const Matrix3d R_world = T.linear() * robot->rotation(link_name);
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++)
					if (fabs(R_world(i, j) - initial_orientation(i, j)) > 1e-12) return 7;
			if ((robot->transformInWorld(link_name).translation() - robot->positionInWorld(link_name)).norm() > 0) return 7;
		}

		// joint task in the nullspace of the motion-force task (:125-126)
		auto joint_task = make_shared<Sai2Primitives::JointTask>(robot);

		if (argc > 4 && string(argv[4]) == "manual") {
			// examples/04-task_and_redundancy.cpp:141-150,188-206: no controller, the nullspace handed from task to task as
			// a MatrixXd (a non-symmetric matrix: the adapter's row- / column-major conversions are on this path)
			Sai2PrimitivesBatched::BatchedSimulation sim(*motion_force_task->batched(), 0.001, 1);
			MatrixXd N_prec = MatrixXd::Identity(dof, dof);
			for (int cycle = 0; cycle < ticks; cycle++) {
				const vector<double> qs = sim.getJointPositions(), dqs = sim.getJointVelocities();
				fwrite(qs.data(), sizeof(double), qs.size(), stdout);
				fwrite(dqs.data(), sizeof(double), dqs.size(), stdout);
				VectorXd q(dof), dq(dof);
				for (int i = 0; i < dof; i++) q(i) = qs[(size_t)i], dq(i) = dqs[(size_t)i];
				robot->setQ(q);
				robot->setDq(dq);
				robot->updateModel();
				N_prec = MatrixXd::Identity(dof, dof);
				motion_force_task->updateTaskModel(N_prec);
				N_prec = motion_force_task->getTaskAndPreviousNullspace();
				joint_task->updateTaskModel(N_prec);
				motion_force_task->setGoalPosition(initial_position + Vector3d(0.0, 0.05, -0.03));
				VectorXd motion_force_task_torques = motion_force_task->computeTorques();
				VectorXd joint_task_torques = joint_task->computeTorques();
				VectorXd control_torques = motion_force_task_torques + joint_task_torques;
				vector<double> out((size_t)dof);
				for (int i = 0; i < dof; i++) out[(size_t)i] = control_torques(i);
				fwrite(out.data(), sizeof(double), out.size(), stdout);
				sim.setJointTorques(out);
				sim.integrate();
			}
			return 0;
		}

		// robot controller (:128-132)
		vector<shared_ptr<Sai2Primitives::TemplateTask>> task_list = {motion_force_task, joint_task};
		auto robot_controller = make_unique<Sai2Primitives::RobotController>(robot, task_list);

		Sai2PrimitivesBatched::BatchedSimulation sim(robot_controller->batched(), 0.001, 1);
		for (int cycle = 0; cycle < ticks; cycle++) {
			const double time = 0.001 * cycle;
			// read joint positions, velocities, update robot model (:143-145)
			const vector<double> qs = sim.getJointPositions(), dqs = sim.getJointVelocities();
			fwrite(qs.data(), sizeof(double), qs.size(), stdout);
			fwrite(dqs.data(), sizeof(double), dqs.size(), stdout);
			VectorXd q(dof), dq(dof);
			for (int i = 0; i < dof; i++) q(i) = qs[(size_t)i], dq(i) = dqs[(size_t)i];
			robot->setQ(q);
			robot->setDq(dq);
			robot->updateModel();

			robot_controller->updateControllerTaskModels();	 // :148

			// orientation: oscillation around Y (:153-169)
			double w_ori_traj = 2 * M_PI * 0.2;
			double amp_ori_traj = M_PI / 8;
			double angle_ori_traj = amp_ori_traj * sin(w_ori_traj * time);
			double ang_vel_traj = amp_ori_traj * w_ori_traj * cos(w_ori_traj * time);
			double ang_accel_traj = amp_ori_traj * w_ori_traj * w_ori_traj * -sin(w_ori_traj * time);
			Matrix3d R = AngleAxisd(angle_ori_traj, Vector3d::UnitY()).toRotationMatrix();
			motion_force_task->setGoalOrientation(R.transpose() * initial_orientation);
			motion_force_task->setGoalAngularVelocity(ang_vel_traj * Vector3d::UnitY());
			motion_force_task->setGoalAngularAcceleration(ang_accel_traj * Vector3d::UnitY());

			// position: circle in the y-z plane (:171-183)
			double radius_circle_pos = 0.05;
			double w_circle_pos = 2 * M_PI * 0.33;
			motion_force_task->setGoalPosition(initial_position +
											   radius_circle_pos * Vector3d(0.0, sin(w_circle_pos * time), 1 - cos(w_circle_pos * time)));
			motion_force_task->setGoalLinearVelocity(radius_circle_pos * w_circle_pos *
													 Vector3d(0.0, cos(w_circle_pos * time), sin(w_circle_pos * time)));
			motion_force_task->setGoalLinearAcceleration(radius_circle_pos * w_circle_pos * w_circle_pos *
														 Vector3d(0.0, -sin(w_circle_pos * time), cos(w_circle_pos * time)));

			if (cycle == ticks / 2) {  // :185-190 (cycle 5000 of the example)
				VectorXd goal_joint_pos = initial_q;
				goal_joint_pos(0) += 1.5;
				joint_task->setGoalPosition(goal_joint_pos);
			}

			VectorXd control_torques = robot_controller->computeControlTorques();  // :195
			vector<double> out((size_t)dof);
			for (int i = 0; i < dof; i++) out[(size_t)i] = control_torques(i);
			fwrite(out.data(), sizeof(double), out.size(), stdout);
			sim.integrate();
		}
		// Sai2Model's Jacobians and point velocities (host side of the adapter) against what the device path reports for
		// the control frame after the last period (MotionForceTask.h:127-133), and against a finite difference of the pose
		{
			// (the task's getters look at the state buffers as they are now, one simulation step past the last period)
			const vector<double> qs = sim.getJointPositions(), dqs = sim.getJointVelocities();
			VectorXd q_end(dof), dq_end(dof);
			for (int i = 0; i < dof; i++) q_end(i) = qs[(size_t)i], dq_end(i) = dqs[(size_t)i];
			robot->setQ(q_end);
			robot->setDq(dq_end);
			robot->updateModel();
			const Vector3d v_host = robot->linearVelocityInWorld(link_name, pos_in_link), w_host = robot->angularVelocityInWorld(link_name);
			if ((v_host - motion_force_task->getCurrentLinearVelocity()).norm() > 1e-10 ||
				(w_host - motion_force_task->getCurrentAngularVelocity()).norm() > 1e-10)
				return 9;
			const MatrixXd Jh = robot->JWorldFrame(link_name, pos_in_link);
			if (Jh.rows() != 6 || Jh.cols() != dof || robot->Jv(link_name, pos_in_link).rows() != 3 || robot->JwWorldFrame(link_name).cols() != dof) return 9;
			const VectorXd q_now = robot->q();
			const double h = 1e-6;
			for (int j = 0; j < dof; j++) {
				VectorXd qp = q_now, qm = q_now;
				qp(j) += h, qm(j) -= h;
				robot->setQ(qp);
				const Vector3d xp = robot->positionInWorld(link_name, pos_in_link);
				robot->setQ(qm);
				const Vector3d xm = robot->positionInWorld(link_name, pos_in_link);
				for (int a = 0; a < 3; a++)
					if (fabs((xp(a) - xm(a)) / (2 * h) - Jh(a, j)) > 1e-8) return 9;
			}
			robot->setQ(q_now);
		}
		// Sai2PrimitivesCommonDefinitions.h:25-27
		const VectorXd kp = Sai2Primitives::extractKpFromGainVector(joint_task->getGains());
		if (kp.size() != 1 || kp(0) != 50.0 || Sai2Primitives::extractKvFromGainVector(joint_task->getGains())(0) != 14.0 ||
			Sai2Primitives::extractKiFromGainVector(joint_task->getGains())(0) != 0.0)
			return 8;
		// a few of the reference's getters, Eigen-typed
		if ((motion_force_task->getGoalPosition() - motion_force_task->getCurrentPosition()).norm() > 0.2) return 5;
		if (joint_task->getGoalPosition().size() != dof || motion_force_task->getTaskAndPreviousNullspace().rows() != dof) return 6;
	} catch (const std::exception& e) {
		cerr << "exception: " << e.what() << endl;
		return 1;
	}
	return 0;
}
