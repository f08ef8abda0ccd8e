/*
 * Sai2PrimitivesEigen.h — the reference's OWN signatures (Eigen types, one robot) on top of the batched library:
 * `Sai2Primitives::RobotController / TemplateTask / JointTask / MotionForceTask` as declared in the reference's
 * src/RobotController.h:25-40, src/tasks/TemplateTask.h:25-123, src/tasks/JointTask.h:56-384 and
 * src/tasks/MotionForceTask.h:96-753, for a batch of ONE robot, so that a control program written against the reference
 * (examples/05-using_robot_controller.cpp:103-196 and the others) compiles and runs with
 *     #include "Sai2PrimitivesEigen.h"            // instead of "Sai2Primitives.h"
 * and nothing else changed. SURVEY.md §8 f-4 ("header-compatible adapter taking Eigen types when they are installed").
 *
 * Needs Eigen 3 (`<Eigen/Dense>`: VectorXd, MatrixXd, Vector3d, Matrix3d, Affine3d and their element access — nothing
 * else of Eigen is used). sai2-model is NOT needed: `Sai2Model::Sai2Model` below is the subset of it the path touches
 * (state, dof, updateModel, position / rotation of a frame), built from the same URDF file. Define
 * SAI2B_EXTERNAL_SAI2_MODEL before including to keep the real class out of the way of this one (then construct the
 * tasks from a `Sai2PrimitivesEigenRobot`, see below).
 *
 * A batch of one robot uses one lane of a GPU: this header is for running existing programs unchanged and for
 * porting them step by step; the throughput is in Sai2PrimitivesBatched.h.
 */
#ifndef SAI2_PRIMITIVES_EIGEN_H_
#define SAI2_PRIMITIVES_EIGEN_H_

#define SAI2B_FACADE_NAMESPACE Sai2PrimitivesBatched
#include <Eigen/Dense>

#include "Sai2PrimitivesBatched.h"

namespace Sai2PrimitivesEigenDetail {
using Sai2PrimitivesBatched::Batch;
using Eigen::Affine3d;
using Eigen::Matrix3d;
using Eigen::MatrixXd;
using Eigen::Vector3d;
using Eigen::VectorXd;

// element-wise conversions only (operator() and sizes): a batch of one is [C][1], i.e. the plain vector; matrices are
// row-major inside the component index on the batched side, whatever Eigen's storage order is
template <class V>
inline Batch batch_of(const V& v) {
	Batch b((size_t)v.size());
	for (int i = 0; i < (int)v.size(); i++) b[i] = v(i);
	return b;
}
template <class M>
inline Batch batch_of_matrix(const M& m) {
	Batch b((size_t)m.rows() * m.cols());
	for (int i = 0; i < (int)m.rows(); i++)
		for (int j = 0; j < (int)m.cols(); j++) b[(size_t)i * m.cols() + j] = m(i, j);
	return b;
}
inline VectorXd vector_of(const Batch& b) {
	VectorXd v((int)b.size());
	for (int i = 0; i < (int)b.size(); i++) v(i) = b[i];
	return v;
}
inline Vector3d vector3_of(const Batch& b) {
	Vector3d v;
	for (int i = 0; i < 3; i++) v(i) = b[i];
	return v;
}
inline MatrixXd matrix_of(const Batch& b, int rows, int cols) {
	MatrixXd m(rows, cols);
	for (int i = 0; i < rows; i++)
		for (int j = 0; j < cols; j++) m(i, j) = b[(size_t)i * cols + j];
	return m;
}
inline Matrix3d matrix3_of(const Batch& b) {
	Matrix3d m;
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) m(i, j) = b[3 * i + j];
	return m;
}
inline void frame_of(const Affine3d& T, double pos[3], double rot[9]) {
	for (int i = 0; i < 3; i++) {
		pos[i] = T.translation()(i);
		for (int j = 0; j < 3; j++) rot[3 * i + j] = T.linear()(i, j);
	}
}
inline std::vector<double> directions_of(const std::vector<Vector3d>& dirs) {
	std::vector<double> d;
	for (const Vector3d& v : dirs)
		for (int i = 0; i < 3; i++) d.push_back(v(i));
	return d;
}
}  // namespace Sai2PrimitivesEigenDetail

// ---- the subset of Sai2Model::Sai2Model the controller path touches (sai2-model is not a dependency of this library)
#ifndef SAI2B_EXTERNAL_SAI2_MODEL
namespace Sai2Model {
class Sai2Model {
#else
class Sai2PrimitivesEigenRobot {
#endif
public:
	// Sai2Model::Sai2Model(path_to_model_file, verbose): a serial chain of 4, 6, 7 or 8 revolute / prismatic joints
#ifndef SAI2B_EXTERNAL_SAI2_MODEL
	explicit Sai2Model(const std::string& path_to_model_file, bool /*verbose*/ = false, int device = 0)
#else
	explicit Sai2PrimitivesEigenRobot(const std::string& path_to_model_file, bool /*verbose*/ = false, int device = 0)
#endif
		: _impl(std::make_shared<Sai2PrimitivesBatched::BatchedRobotModel>(path_to_model_file, 1, device)) {
	}
	int dof() const { return _impl->dof(); }
	int qSize() const { return _impl->dof(); }
	Eigen::VectorXd q() const { return Sai2PrimitivesEigenDetail::vector_of(_impl->q()); }
	Eigen::VectorXd dq() const { return Sai2PrimitivesEigenDetail::vector_of(_impl->dq()); }
	void setQ(const Eigen::VectorXd& q) { _impl->setQ(Sai2PrimitivesEigenDetail::batch_of(q)); }
	void setDq(const Eigen::VectorXd& dq) { _impl->setDq(Sai2PrimitivesEigenDetail::batch_of(dq)); }
	void updateModel() { _impl->updateModel(); }  // kinematics and dynamics are recomputed on the device every tick
	// Sai2Model::setTRobotBase / TRobotBase: the pose of the robot base in the world (examples/05-...cpp:69); the tasks
	// work in the world frame, so this goes into the model the kernels see — before tasks are built on the robot
	void setTRobotBase(const Eigen::Affine3d& T) {
		double p[3], R[9];
		for (int i = 0; i < 3; i++) {
			p[i] = T.translation()(i);
			for (int j = 0; j < 3; j++) R[3 * i + j] = T.linear()(i, j);
		}
		_impl->setTRobotBase(p, R);
	}
	Eigen::Affine3d TRobotBase() const {
		Eigen::Affine3d T = Eigen::Affine3d::Identity();
		for (int i = 0; i < 3; i++) {
			T.translation()(i) = _impl->TRobotBasePosition()[i];
			for (int j = 0; j < 3; j++) T.linear()(i, j) = _impl->TRobotBaseRotation()[3 * i + j];
		}
		return T;
	}
	// pose of a frame attached to a link: ...InWorld in the world frame, the plain ones in the robot's base frame
	Eigen::Vector3d positionInWorld(const std::string& link_name, const Eigen::Vector3d& pos_in_link = Eigen::Vector3d::Zero()) const {
		Eigen::Affine3d T = transformInWorld(link_name);
		Eigen::Vector3d p;
		for (int i = 0; i < 3; i++) {
			p(i) = T.translation()(i);
			for (int j = 0; j < 3; j++) p(i) += T.linear()(i, j) * pos_in_link(j);
		}
		return p;
	}
	Eigen::Vector3d position(const std::string& link_name, const Eigen::Vector3d& pos_in_link = Eigen::Vector3d::Zero()) const {
		const Eigen::Vector3d pw = positionInWorld(link_name, pos_in_link);
		const double *pb = _impl->TRobotBasePosition(), *Rb = _impl->TRobotBaseRotation();
		Eigen::Vector3d p;
		for (int i = 0; i < 3; i++) {
			p(i) = 0;
			for (int k = 0; k < 3; k++) p(i) += Rb[3 * k + i] * (pw(k) - pb[k]);
		}
		return p;
	}
	Eigen::Matrix3d rotationInWorld(const std::string& link_name, const Eigen::Matrix3d& rot_in_link = Eigen::Matrix3d::Identity()) const {
		Eigen::Affine3d T = transformInWorld(link_name);
		Eigen::Matrix3d R;
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				R(i, j) = 0;
				for (int k = 0; k < 3; k++) R(i, j) += T.linear()(i, k) * rot_in_link(k, j);
			}
		return R;
	}
	Eigen::Matrix3d rotation(const std::string& link_name, const Eigen::Matrix3d& rot_in_link = Eigen::Matrix3d::Identity()) const {
		const Eigen::Matrix3d Rw = rotationInWorld(link_name, rot_in_link);
		const double* Rb = _impl->TRobotBaseRotation();
		Eigen::Matrix3d R;
		for (int i = 0; i < 3; i++)
			for (int j = 0; j < 3; j++) {
				R(i, j) = 0;
				for (int k = 0; k < 3; k++) R(i, j) += Rb[3 * k + i] * Rw(k, j);
			}
		return R;
	}
	Eigen::Affine3d transform(const std::string& link_name) const {
		Eigen::Affine3d T = Eigen::Affine3d::Identity();
		T.translation() = position(link_name);
		T.linear() = rotation(link_name);
		return T;
	}
	// world transform of a (possibly fixed-attached) link by its URDF name: forward kinematics on the host, same chain
	// convention as the library (include/sai2b.h: sai2b_robot_model; the base pose is part of the first joint's origin)
	Eigen::Affine3d transformInWorld(const std::string& link_name) const {
		double axes[SAI2B_MAX_DOF][3], origins[SAI2B_MAX_DOF][3];
		int link;
		return sweep(link_name, link, axes, origins);
	}
	// Sai2Model::JWorldFrame(link_name, pos_in_link): 6 x dof, linear rows first (what MotionForceTask.cpp:262 takes), and
	// its halves / base-frame forms; the velocities of the point are J dq
	Eigen::MatrixXd JWorldFrame(const std::string& link_name, const Eigen::Vector3d& pos_in_link = Eigen::Vector3d::Zero()) const {
		double z[SAI2B_MAX_DOF][3], o[SAI2B_MAX_DOF][3];
		int link;
		const Eigen::Affine3d T = sweep(link_name, link, z, o);
		double x[3];
		for (int a = 0; a < 3; a++) {
			x[a] = T.translation()(a);
			for (int b = 0; b < 3; b++) x[a] += T.linear()(a, b) * pos_in_link(b);
		}
		const sai2b_robot_model& m = _impl->model();
		Eigen::MatrixXd J = Eigen::MatrixXd::Zero(6, m.dof);
		for (int i = 0; i <= link; i++) {
			const double d[3] = {x[0] - o[i][0], x[1] - o[i][1], x[2] - o[i][2]};
			const double v[3] = {z[i][1] * d[2] - z[i][2] * d[1], z[i][2] * d[0] - z[i][0] * d[2], z[i][0] * d[1] - z[i][1] * d[0]};
			const bool prismatic = m.joint_type[i] != 0;
			for (int a = 0; a < 3; a++) {
				J(a, i) = prismatic ? z[i][a] : v[a];
				J(3 + a, i) = prismatic ? 0.0 : z[i][a];
			}
		}
		return J;
	}
	Eigen::MatrixXd J(const std::string& link_name, const Eigen::Vector3d& pos_in_link = Eigen::Vector3d::Zero()) const {
		const Eigen::MatrixXd Jw_ = JWorldFrame(link_name, pos_in_link);
		const double* Rb = _impl->TRobotBaseRotation();
		Eigen::MatrixXd Jb = Eigen::MatrixXd::Zero(6, Jw_.cols());
		for (int h = 0; h < 2; h++)
			for (int a = 0; a < 3; a++)
				for (int j = 0; j < (int)Jw_.cols(); j++)
					for (int k = 0; k < 3; k++) Jb(3 * h + a, j) += Rb[3 * k + a] * Jw_(3 * h + k, j);
		return Jb;
	}
	Eigen::MatrixXd JvWorldFrame(const std::string& l, const Eigen::Vector3d& p = Eigen::Vector3d::Zero()) const { return half(JWorldFrame(l, p), 0); }
	Eigen::MatrixXd JwWorldFrame(const std::string& l) const { return half(JWorldFrame(l), 1); }
	Eigen::MatrixXd Jv(const std::string& l, const Eigen::Vector3d& p = Eigen::Vector3d::Zero()) const { return half(J(l, p), 0); }
	Eigen::MatrixXd Jw(const std::string& l) const { return half(J(l), 1); }
	Eigen::Vector3d linearVelocityInWorld(const std::string& l, const Eigen::Vector3d& p = Eigen::Vector3d::Zero()) const { return times_dq(JWorldFrame(l, p), 0); }
	Eigen::Vector3d angularVelocityInWorld(const std::string& l) const { return times_dq(JWorldFrame(l), 1); }
	Eigen::Vector3d linearVelocity(const std::string& l, const Eigen::Vector3d& p = Eigen::Vector3d::Zero()) const { return times_dq(J(l, p), 0); }
	Eigen::Vector3d angularVelocity(const std::string& l) const { return times_dq(J(l), 1); }
	std::shared_ptr<Sai2PrimitivesBatched::BatchedRobotModel>& batched() { return _impl; }

private:
	static Eigen::MatrixXd half(const Eigen::MatrixXd& J6, int which) {
		Eigen::MatrixXd H = Eigen::MatrixXd::Zero(3, J6.cols());
		for (int a = 0; a < 3; a++)
			for (int j = 0; j < (int)J6.cols(); j++) H(a, j) = J6(3 * which + a, j);
		return H;
	}
	Eigen::Vector3d times_dq(const Eigen::MatrixXd& J6, int which) const {
		Eigen::Vector3d v = Eigen::Vector3d::Zero();
		for (int a = 0; a < 3; a++)
			for (int j = 0; j < (int)J6.cols(); j++) v(a) += J6(3 * which + a, j) * _impl->dq()[j];
		return v;
	}
	// forward kinematics on the host down to the moving link `link_name` hangs on: world transform of the named link,
	// and for every joint up to there its axis and a point on it, in the world
	Eigen::Affine3d sweep(const std::string& link_name, int& link, double axes[][3], double origins[][3]) const {
		const double zero[3] = {0, 0, 0};
		double fp[3], fr[9];
		link = _impl->resolveLink(link_name, zero, nullptr, fp, fr);
		const sai2b_robot_model& m = _impl->model();
		double R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, p[3] = {0, 0, 0};
		for (int i = 0; i <= link; i++) {
			const double cr = std::cos(m.joint_rpy[i][0]), sr = std::sin(m.joint_rpy[i][0]), cp = std::cos(m.joint_rpy[i][1]),
						 sp = std::sin(m.joint_rpy[i][1]), cy = std::cos(m.joint_rpy[i][2]), sy = std::sin(m.joint_rpy[i][2]);
			const double E[9] = {cy * cp, cy * sp * sr - sy * cr, cy * sp * cr + sy * sr, sy * cp, sy * sp * sr + cy * cr,
								 sy * sp * cr - cy * sr, -sp, cp * sr, cp * cr};
			const double qi = _impl->q()[i];
			const bool prismatic = m.joint_type[i] != 0;
			const double c = prismatic ? 1.0 : std::cos(qi), s = prismatic ? 0.0 : std::sin(qi);
			double RE[9], Rn[9], pn[3];
			for (int a = 0; a < 3; a++) {
				pn[a] = p[a] + R[3 * a] * m.joint_xyz[i][0] + R[3 * a + 1] * m.joint_xyz[i][1] + R[3 * a + 2] * m.joint_xyz[i][2];
				for (int b = 0; b < 3; b++) RE[3 * a + b] = R[3 * a] * E[b] + R[3 * a + 1] * E[3 + b] + R[3 * a + 2] * E[6 + b];
			}
			for (int a = 0; a < 3; a++) {
				Rn[3 * a] = c * RE[3 * a] + s * RE[3 * a + 1];
				Rn[3 * a + 1] = c * RE[3 * a + 1] - s * RE[3 * a];
				Rn[3 * a + 2] = RE[3 * a + 2];
				if (prismatic) pn[a] += RE[3 * a + 2] * qi;
			}
			for (int a = 0; a < 9; a++) R[a] = Rn[a];
			for (int a = 0; a < 3; a++) p[a] = pn[a], axes[i][a] = RE[3 * a + 2], origins[i][a] = pn[a];
		}
		Eigen::Affine3d T = Eigen::Affine3d::Identity();
		for (int a = 0; a < 3; a++) {
			T.translation()(a) = p[a] + R[3 * a] * fp[0] + R[3 * a + 1] * fp[1] + R[3 * a + 2] * fp[2];
			for (int b = 0; b < 3; b++) T.linear()(a, b) = R[3 * a] * fr[b] + R[3 * a + 1] * fr[3 + b] + R[3 * a + 2] * fr[6 + b];
		}
		return T;
	}
	std::shared_ptr<Sai2PrimitivesBatched::BatchedRobotModel> _impl;
};
#ifndef SAI2B_EXTERNAL_SAI2_MODEL
}  // namespace Sai2Model
#endif

namespace Sai2Primitives {
using namespace Eigen;
using std::shared_ptr;
using std::string;
using std::vector;
namespace B_ = Sai2PrimitivesBatched;
namespace D_ = Sai2PrimitivesEigenDetail;
#ifndef SAI2B_EXTERNAL_SAI2_MODEL
typedef Sai2Model::Sai2Model RobotModel_;
#else
typedef Sai2PrimitivesEigenRobot RobotModel_;
#endif

using B_::BOUNDED_INERTIA_ESTIMATES;
using B_::DynamicDecouplingType;
using B_::FULL_DYNAMIC_DECOUPLING;
using B_::IMPEDANCE;
using B_::JOINT_TASK;
using B_::MOTION_FORCE_TASK;
using B_::PIDGains;
using B_::TaskType;
using B_::UNDEFINED;

// reference src/helper_modules/Sai2PrimitivesCommonDefinitions.h:25-27: one field of a gain list as a vector
template <class Field>
inline VectorXd gainFieldAsVector_(const vector<PIDGains>& gains, Field field) {
	VectorXd out((int)gains.size());
	for (size_t i = 0; i < gains.size(); i++) out((int)i) = field(gains[i]);
	return out;
}
inline VectorXd extractKpFromGainVector(const vector<PIDGains>& gains) { return gainFieldAsVector_(gains, [](const PIDGains& g) { return g.kp; }); }
inline VectorXd extractKvFromGainVector(const vector<PIDGains>& gains) { return gainFieldAsVector_(gains, [](const PIDGains& g) { return g.kv; }); }
inline VectorXd extractKiFromGainVector(const vector<PIDGains>& gains) { return gainFieldAsVector_(gains, [](const PIDGains& g) { return g.ki; }); }

// reference src/tasks/TemplateTask.h:25-123
class TemplateTask {
public:
	virtual ~TemplateTask() = default;
	void updateTaskModel(const MatrixXd& N_prec) {	// TemplateTask.h:42
		const int n = _robot->dof();
		if (N_prec.rows() != N_prec.cols()) throw std::invalid_argument("N_prec matrix not square in TemplateTask::updateTaskModel\n");
		if (N_prec.rows() != n) throw std::invalid_argument("N_prec matrix size not consistent with robot dof in TemplateTask::updateTaskModel\n");
		base().updateTaskModel(D_::batch_of_matrix(N_prec));
	}
	VectorXd computeTorques() { return D_::vector_of(base().computeTorques()); }	 // TemplateTask.h:49
	VectorXd computeTorques(const VectorXd& tau_prec) { return D_::vector_of(base().computeTorques(D_::batch_of(tau_prec))); }	// :58
	void reInitializeTask() { base().reInitializeTask(); }																			// :65
	MatrixXd getTaskNullspace() const { return D_::matrix_of(base().getTaskNullspace(), _robot->dof(), _robot->dof()); }				// :73
	MatrixXd getPreviousTasksNullspace() const { return D_::matrix_of(base().getPreviousTasksNullspace(), _robot->dof(), _robot->dof()); }
	MatrixXd getTaskAndPreviousNullspace() const { return D_::matrix_of(base().getTaskAndPreviousNullspace(), _robot->dof(), _robot->dof()); }
	const shared_ptr<RobotModel_>& getConstRobotModel() const { return _robot; }
	double getLoopTimestep() const { return base().getLoopTimestep(); }
	TaskType getTaskType() const { return base().getTaskType(); }
	string getTaskName() const { return base().getTaskName(); }
	void setDynamicDecouplingType(const DynamicDecouplingType type) { base().setDynamicDecouplingType(type); }
	void setBoundedInertiaEstimateThreshold(const double threshold) { base().setBoundedInertiaEstimateThreshold(threshold); }
	virtual shared_ptr<B_::TemplateTask> batched() const = 0;

protected:
	explicit TemplateTask(shared_ptr<RobotModel_>& robot) : _robot(robot) {}
	B_::TemplateTask& base() const { return *batched(); }
	shared_ptr<RobotModel_> _robot;
};

// reference src/tasks/JointTask.h:56-384
class JointTask : public TemplateTask {
public:
	JointTask(shared_ptr<RobotModel_>& robot, const string& task_name = "joint_task", const double loop_timestep = 0.001)
		: TemplateTask(robot), _t(std::make_shared<B_::JointTask>(robot->batched(), task_name, loop_timestep)) {}
	JointTask(shared_ptr<RobotModel_>& robot, const MatrixXd& joint_selection_matrix, const string& task_name = "partial_joint_task",
			  const double loop_timestep = 0.001)
		: TemplateTask(robot) {
		if (joint_selection_matrix.cols() != robot->dof())
			throw std::invalid_argument("joint selection matrix size not consistent with robot dof in JointTask constructor\n");
		_t = std::make_shared<B_::JointTask>(robot->batched(), D_::batch_of_matrix(joint_selection_matrix), (int)joint_selection_matrix.rows(),
											 task_name, loop_timestep);
	}
	shared_ptr<B_::TemplateTask> batched() const override { return _t; }
	MatrixXd getJointSelectionMatrix() const { return D_::matrix_of(_t->getJointSelectionMatrix(), _t->getTaskDof(), _robot->dof()); }
	int getTaskDof() const { return _t->getTaskDof(); }
	bool isFullJointTask() const { return _t->isFullJointTask(); }
	VectorXd getCurrentPosition() { return D_::vector_of(_t->getCurrentPosition()); }
	VectorXd getCurrentVelocity() { return D_::vector_of(_t->getCurrentVelocity()); }
	void setGoalPosition(const VectorXd& v) { _t->setGoalPosition(D_::batch_of(v)); }
	void setGoalVelocity(const VectorXd& v) { _t->setGoalVelocity(D_::batch_of(v)); }
	void setGoalAcceleration(const VectorXd& v) { _t->setGoalAcceleration(D_::batch_of(v)); }
	VectorXd getGoalPosition() const { return D_::vector_of(_t->getGoalPosition()); }
	VectorXd getGoalVelocity() const { return D_::vector_of(_t->getGoalVelocity()); }
	VectorXd getGoalAcceleration() const { return D_::vector_of(_t->getGoalAcceleration()); }
	VectorXd getDesiredPosition() const { return D_::vector_of(_t->getDesiredPosition()); }
	VectorXd getDesiredVelocity() const { return D_::vector_of(_t->getDesiredVelocity()); }
	VectorXd getDesiredAcceleration() const { return D_::vector_of(_t->getDesiredAcceleration()); }
	void setGains(const VectorXd& kp, const VectorXd& kv, const VectorXd& ki) { _t->setGains(D_::batch_of(kp), D_::batch_of(kv), D_::batch_of(ki)); }
	void setGains(const VectorXd& kp, const VectorXd& kv) { _t->setGains(D_::batch_of(kp), D_::batch_of(kv)); }
	void setGains(const double kp, const double kv, const double ki = 0) { _t->setGains(kp, kv, ki); }
	void setGainsUnsafe(const VectorXd& kp, const VectorXd& kv, const VectorXd& ki) { _t->setGainsUnsafe(D_::batch_of(kp), D_::batch_of(kv), D_::batch_of(ki)); }
	vector<PIDGains> getGains() const { return _t->getGains(); }
	void enableInternalOtgAccelerationLimited(const VectorXd& max_velocity, const VectorXd& max_acceleration) {
		_t->enableInternalOtgAccelerationLimited(D_::batch_of(max_velocity), D_::batch_of(max_acceleration));
	}
	void enableInternalOtgAccelerationLimited(const double max_velocity, const double max_acceleration) {
		_t->enableInternalOtgAccelerationLimited(max_velocity, max_acceleration);
	}
	void enableInternalOtgJerkLimited(const double max_velocity, const double max_acceleration, const double max_jerk) {
		_t->enableInternalOtgJerkLimited(max_velocity, max_acceleration, max_jerk);
	}
	void enableInternalOtgJerkLimited(const VectorXd& max_velocity, const VectorXd& max_acceleration, const VectorXd& max_jerk) {
		_t->enableInternalOtgJerkLimited(D_::batch_of(max_velocity), D_::batch_of(max_acceleration), D_::batch_of(max_jerk));
	}
	void disableInternalOtg() { _t->disableInternalOtg(); }
	bool getInternalOtgEnabled() const { return _t->getInternalOtgEnabled(); }
	void enableVelocitySaturation(const VectorXd& saturation_velocity) { _t->enableVelocitySaturation(std::vector<double>(D_::batch_of(saturation_velocity))); }
	void enableVelocitySaturation(const double saturation_velocity) { _t->enableVelocitySaturation(saturation_velocity); }
	void disableVelocitySaturation() { _t->disableVelocitySaturation(); }
	bool getVelocitySaturationEnabled() const { return _t->getVelocitySaturationEnabled(); }
	VectorXd getVelocitySaturationMaxVelocity() const { return D_::vector_of(_t->getVelocitySaturationMaxVelocity()); }
	double getBoundedInertiaEstimateThreshold() const { return _t->getBoundedInertiaEstimateThreshold(); }
	void resetIntegrators() { _t->resetIntegrators(); }
	bool goalPositionReached(const double tolerance) {	// JointTask.cpp:437
		double n2 = 0;
		const VectorXd e = getGoalPosition(), c = getCurrentPosition();
		for (int i = 0; i < (int)e.size(); i++) n2 += (e(i) - c(i)) * (e(i) - c(i));
		return std::sqrt(n2) < tolerance;
	}

private:
	shared_ptr<B_::JointTask> _t;
};

// reference src/tasks/MotionForceTask.h:96-753
class MotionForceTask : public TemplateTask {
public:
	MotionForceTask(shared_ptr<RobotModel_>& robot, const string& link_name, const Affine3d& compliant_frame = Affine3d::Identity(),
					const string& task_name = "motion_force_task", const bool is_force_motion_parametrization_in_compliant_frame = false,
					const double loop_timestep = 0.001)
		: TemplateTask(robot), _link_name(link_name) {
		double pos[3], rot[9];
		D_::frame_of(compliant_frame, pos, rot);
		_t = std::make_shared<B_::MotionForceTask>(robot->batched(), link_name, pos, rot, task_name, is_force_motion_parametrization_in_compliant_frame,
												   loop_timestep);
	}
	MotionForceTask(shared_ptr<RobotModel_>& robot, const string& link_name, const vector<Vector3d>& controlled_directions_translation,
					const vector<Vector3d>& controlled_directions_rotation, const Affine3d& compliant_frame = Affine3d::Identity(),
					const string& task_name = "partial_motion_force_task", const bool is_force_motion_parametrization_in_compliant_frame = false,
					const double loop_timestep = 0.001)
		: TemplateTask(robot), _link_name(link_name) {
		double pos[3], rot[9];
		D_::frame_of(compliant_frame, pos, rot);
		_t = std::make_shared<B_::MotionForceTask>(robot->batched(), link_name, D_::directions_of(controlled_directions_translation),
												   D_::directions_of(controlled_directions_rotation), pos, rot, task_name,
												   is_force_motion_parametrization_in_compliant_frame, loop_timestep);
	}
	shared_ptr<B_::TemplateTask> batched() const override { return _t; }
	// current state (MotionForceTask.h:121-165)
	Vector3d getCurrentPosition() const { return D_::vector3_of(_t->getCurrentPosition()); }
	Vector3d getCurrentLinearVelocity() const { return D_::vector3_of(_t->getCurrentLinearVelocity()); }
	Matrix3d getCurrentOrientation() const { return D_::matrix3_of(_t->getCurrentOrientation()); }
	Vector3d getCurrentAngularVelocity() const { return D_::vector3_of(_t->getCurrentAngularVelocity()); }
	Vector3d getSensedForceControlWorldFrame() const { return D_::vector3_of(_t->getSensedForceControlWorldFrame()); }
	Vector3d getSensedMomentControlWorldFrame() const { return D_::vector3_of(_t->getSensedMomentControlWorldFrame()); }
	Vector3d getSensedForceSensor() const { return D_::vector3_of(_t->getSensedForceSensor()); }
	Vector3d getSensedMomentSensor() const { return D_::vector3_of(_t->getSensedMomentSensor()); }
	// goals (MotionForceTask.h:211-247)
	void setGoalPosition(const Vector3d& v) { _t->setGoalPosition(D_::batch_of(v)); }
	void setGoalOrientation(const Matrix3d& R) { _t->setGoalOrientation(D_::batch_of_matrix(R)); }
	void setGoalLinearVelocity(const Vector3d& v) { _t->setGoalLinearVelocity(D_::batch_of(v)); }
	void setGoalAngularVelocity(const Vector3d& v) { _t->setGoalAngularVelocity(D_::batch_of(v)); }
	void setGoalLinearAcceleration(const Vector3d& v) { _t->setGoalLinearAcceleration(D_::batch_of(v)); }
	void setGoalAngularAcceleration(const Vector3d& v) { _t->setGoalAngularAcceleration(D_::batch_of(v)); }
	Vector3d getGoalPosition() const { return D_::vector3_of(_t->getGoalPosition()); }
	Matrix3d getGoalOrientation() const { return D_::matrix3_of(_t->getGoalOrientation()); }
	Vector3d getGoalLinearVelocity() const { return D_::vector3_of(_t->getGoalLinearVelocity()); }
	Vector3d getGoalAngularVelocity() const { return D_::vector3_of(_t->getGoalAngularVelocity()); }
	Vector3d getGoalLinearAcceleration() const { return D_::vector3_of(_t->getGoalLinearAcceleration()); }
	Vector3d getGoalAngularAcceleration() const { return D_::vector3_of(_t->getGoalAngularAcceleration()); }
	Vector3d getDesiredPosition() const { return D_::vector3_of(_t->getDesiredPosition()); }
	Matrix3d getDesiredOrientation() const { return D_::matrix3_of(_t->getDesiredOrientation()); }
	Vector3d getDesiredLinearVelocity() const { return D_::vector3_of(_t->getDesiredLinearVelocity()); }
	Vector3d getDesiredAngularVelocity() const { return D_::vector3_of(_t->getDesiredAngularVelocity()); }
	Vector3d getDesiredLinearAcceleration() const { return D_::vector3_of(_t->getDesiredLinearAcceleration()); }
	Vector3d getDesiredAngularAcceleration() const { return D_::vector3_of(_t->getDesiredAngularAcceleration()); }
	Vector3d getPositionError() const { return D_::vector3_of(_t->getPositionError()); }
	Vector3d getOrientationError() const { return D_::vector3_of(_t->getOrientationError()); }
	bool goalPositionReached(const double tolerance) const { return _t->goalPositionReached(tolerance)[0]; }
	bool goalOrientationReached(const double tolerance) const { return _t->goalOrientationReached(tolerance)[0]; }
	VectorXd getUnitMassForce() const { return D_::vector_of(_t->getUnitMassForce()); }
	// gains (MotionForceTask.h:272-328)
	void setPosControlGains(double kp, double kv, double ki = 0) { _t->setPosControlGains(kp, kv, ki); }
	void setPosControlGains(const Vector3d& kp, const Vector3d& kv, const Vector3d& ki = Vector3d::Zero()) { gains3(&B_::MotionForceTask::setPosControlGains, kp, kv, ki); }
	void setPosControlGainsUnsafe(const Vector3d& kp, const Vector3d& kv, const Vector3d& ki = Vector3d::Zero()) { gains3(&B_::MotionForceTask::setPosControlGainsUnsafe, kp, kv, ki); }
	void setOriControlGains(double kp, double kv, double ki = 0) { _t->setOriControlGains(kp, kv, ki); }
	void setOriControlGains(const Vector3d& kp, const Vector3d& kv, const Vector3d& ki = Vector3d::Zero()) { gains3(&B_::MotionForceTask::setOriControlGains, kp, kv, ki); }
	void setOriControlGainsUnsafe(const Vector3d& kp, const Vector3d& kv, const Vector3d& ki = Vector3d::Zero()) { gains3(&B_::MotionForceTask::setOriControlGainsUnsafe, kp, kv, ki); }
	void setForceControlGains(double kp, double kv, double ki) { _t->setForceControlGains(kp, kv, ki); }
	void setMomentControlGains(double kp, double kv, double ki) { _t->setMomentControlGains(kp, kv, ki); }
	vector<PIDGains> getPosControlGains() const { return _t->getPosControlGains(); }
	vector<PIDGains> getOriControlGains() const { return _t->getOriControlGains(); }
	vector<PIDGains> getForceControlGains() const { return _t->getForceControlGains(); }
	vector<PIDGains> getMomentControlGains() const { return _t->getMomentControlGains(); }
	void setFeedforwardForceGain(const double k) { _t->setFeedforwardForceGain(k); }
	double getFeedforwardForceGain() const { return _t->getFeedforwardForceGain(); }
	void setFeedforwardmomentGain(const double k) { _t->setFeedforwardmomentGain(k); }
	double getFeedforwardmomentGain() const { return _t->getFeedforwardmomentGain(); }
	void setMaxForceControlFeedbackOutput(const double v) { _t->setMaxForceControlFeedbackOutput(v); }
	double getMaxForceControlFeedbackOutput() const { return _t->getMaxForceControlFeedbackOutput(); }
	void setMaxMomentControlFeedbackOutput(const double v) { _t->setMaxMomentControlFeedbackOutput(v); }
	double getMaxMomentControlFeedbackOutput() const { return _t->getMaxMomentControlFeedbackOutput(); }
	// force / motion spaces and sensing (MotionForceTask.h:360-385,576-660; MotionForceTask.cpp:794-890)
	void setGoalForce(const Vector3d& v) { _t->setGoalForce(D_::batch_of(v)); }
	void setGoalMoment(const Vector3d& v) { _t->setGoalMoment(D_::batch_of(v)); }
	Vector3d getGoalForce() const { return D_::vector3_of(_t->getGoalForce()); }
	Vector3d getGoalMoment() const { return D_::vector3_of(_t->getGoalMoment()); }
	void setForceSensorFrame(const string link_name, const Affine3d transformation_in_link) {
		double pos[3], rot[9];
		D_::frame_of(transformation_in_link, pos, rot);
		_t->setForceSensorFrame(link_name, pos, rot);
	}
	void updateSensedForceAndMoment(const Vector3d sensed_force_sensor_frame, const Vector3d sensed_moment_sensor_frame) {
		_t->updateSensedForceAndMoment(D_::batch_of(sensed_force_sensor_frame), D_::batch_of(sensed_moment_sensor_frame));
	}
	bool parametrizeForceMotionSpaces(const int force_space_dimension, const Vector3d& force_or_motion_single_axis = Vector3d::Zero()) {
		const double a[3] = {force_or_motion_single_axis(0), force_or_motion_single_axis(1), force_or_motion_single_axis(2)};
		return _t->parametrizeForceMotionSpaces(force_space_dimension, a);
	}
	bool parametrizeMomentRotMotionSpaces(const int moment_space_dimension, const Vector3d& moment_or_rot_motion_single_axis = Vector3d::Zero()) {
		const double a[3] = {moment_or_rot_motion_single_axis(0), moment_or_rot_motion_single_axis(1), moment_or_rot_motion_single_axis(2)};
		return _t->parametrizeMomentRotMotionSpaces(moment_space_dimension, a);
	}
	int getForceSpaceDimension() const { return _t->getForceSpaceDimension(); }
	int getMomentSpaceDimension() const { return _t->getMomentSpaceDimension(); }
	Vector3d getForceMotionSingleAxis() const { return D_::vector3_of(_t->getForceMotionSingleAxis()); }
	Vector3d getMomentRotMotionSingleAxis() const { return D_::vector3_of(_t->getMomentRotMotionSingleAxis()); }
	Matrix3d sigmaForce() const { return D_::matrix3_of(_t->sigmaForce()); }
	Matrix3d sigmaPosition() const { return D_::matrix3_of(_t->sigmaPosition()); }
	Matrix3d sigmaMoment() const { return D_::matrix3_of(_t->sigmaMoment()); }
	Matrix3d sigmaOrientation() const { return D_::matrix3_of(_t->sigmaOrientation()); }
	Matrix3d posSelectionProjector() const { return D_::matrix3_of(_t->posSelectionProjector()); }
	Matrix3d oriSelectionProjector() const { return D_::matrix3_of(_t->oriSelectionProjector()); }
	void setClosedLoopForceControl() { _t->setClosedLoopForceControl(true); }
	void setOpenLoopForceControl() { _t->setClosedLoopForceControl(false); }
	void setClosedLoopMomentControl() { _t->setClosedLoopMomentControl(true); }
	void setOpenLoopMomentControl() { _t->setClosedLoopMomentControl(false); }
	void enablePassivity() { _t->enablePassivity(); }
	void disablePassivity() { _t->disablePassivity(); }
	// velocity saturation, internal OTG (MotionForceTask.h:387-440)
	void enableVelocitySaturation(const double linear_vel_sat = 0.3, const double angular_vel_sat = M_PI / 3) { _t->enableVelocitySaturation(linear_vel_sat, angular_vel_sat); }
	void disableVelocitySaturation() { _t->disableVelocitySaturation(); }
	bool getVelocitySaturationEnabled() const { return _t->getVelocitySaturationEnabled(); }
	double getLinearSaturationVelocity() const { return _t->getLinearSaturationVelocity(); }
	double getAngularSaturationVelocity() const { return _t->getAngularSaturationVelocity(); }
	void enableInternalOtgAccelerationLimited(const double max_linear_velelocity, const double max_linear_acceleration,
											  const double max_angular_velocity, const double max_angular_acceleration) {
		_t->enableInternalOtgAccelerationLimited(max_linear_velelocity, max_linear_acceleration, max_angular_velocity, max_angular_acceleration);
	}
	void enableInternalOtgJerkLimited(const double a, const double b, const double c, const double d, const double e, const double f) {
		_t->enableInternalOtgJerkLimited(a, b, c, d, e, f);
	}
	void disableInternalOtg() { _t->disableInternalOtg(); }
	bool getInternalOtgEnabled() const { return _t->getInternalOtgEnabled(); }
	double getBoundedInertiaEstimateThreshold() const { return _t->getBoundedInertiaEstimateThreshold(); }
	void resetIntegrators() { _t->resetIntegrators(); }
	void resetIntegratorsLinear() { _t->resetIntegratorsLinear(); }
	void resetIntegratorsAngular() { _t->resetIntegratorsAngular(); }
	// singularity handling (MotionForceTask.h:669-753)
	void handleAllSingularitiesAsType1(const bool flag) { _t->handleAllSingularitiesAsType1(flag); }
	// not in the reference: see Sai2PrimitivesBatched.h (the sign Eigen's JacobiSVD leaves on V_s is not reproduced)
	void setSingularVectorSign(const int convention) { _t->setSingularVectorSign(convention); }
	void setType1Posture(const VectorXd& q_des) { _t->setType1Posture(D_::batch_of(q_des)); }
	void enableSingularityHandling() { _t->enableSingularityHandling(true); }
	void disableSingularityHandling() { _t->disableSingularityHandling(); }
	void setSingularityHandlingBounds(const double& s_min, const double& s_max) { _t->setSingularityHandlingBounds(s_min, s_max); }
	void setSingularityHandlingGains(const double& kp_type_1, const double& kv_type_1, const double& kv_type_2) {
		_t->setSingularityHandlingGains(kp_type_1, kv_type_1, kv_type_2);
	}

private:
	typedef void (B_::MotionForceTask::*Gains3)(const double*, const double*, const double*);
	void gains3(Gains3 f, const Vector3d& kp, const Vector3d& kv, const Vector3d& ki) {
		const double p[3] = {kp(0), kp(1), kp(2)}, v[3] = {kv(0), kv(1), kv(2)}, i[3] = {ki(0), ki(1), ki(2)};
		((*_t).*f)(p, v, i);
	}
	shared_ptr<B_::MotionForceTask> _t;
	string _link_name;
};

// reference src/RobotController.h:25-40
class RobotController {
public:
	RobotController(shared_ptr<RobotModel_>& robot, vector<shared_ptr<TemplateTask>>& tasks) : _robot(robot), _tasks(tasks) {
		vector<shared_ptr<B_::TemplateTask>> inner;
		for (auto& t : tasks) inner.push_back(t->batched());
		_c = std::make_unique<B_::RobotController>(robot->batched(), inner);
		// every getter of the reference works at any time (nullspaces, singular values, unit-mass force of the last tick):
		// for one robot the per-robot debug outputs cost nothing that matters
		_c->enableIntrospection(true);
	}
	void updateControllerTaskModels() { _c->updateControllerTaskModels(); }					   // RobotController.cpp:53-60
	VectorXd computeControlTorques() { return D_::vector_of(_c->computeControlTorques()); }  // RobotController.cpp:62-74
	void enableGravityCompensation(const bool enable) { _c->enableGravityCompensation(enable); }
	void reinitializeTasks() { _c->reinitializeTasks(); }
	vector<string> getTaskNames() const { return _c->getTaskNames(); }
	shared_ptr<JointTask> getJointTaskByName(const string& task_name) {
		for (auto& t : _tasks)
			if (t->getTaskName() == task_name) {
				auto j = std::dynamic_pointer_cast<JointTask>(t);
				if (!j) throw std::invalid_argument("Task " + task_name + " is not a JointTask in RobotController::getJointTaskByName\n");
				return j;
			}
		throw std::invalid_argument("Task " + task_name + " not found in RobotController::getJointTaskByName\n");
	}
	shared_ptr<MotionForceTask> getMotionForceTaskByName(const string& task_name) {
		for (auto& t : _tasks)
			if (t->getTaskName() == task_name) {
				auto m = std::dynamic_pointer_cast<MotionForceTask>(t);
				if (!m) throw std::invalid_argument("Task " + task_name + " is not a MotionForceTask in RobotController::getMotionForceTaskByName\n");
				return m;
			}
		throw std::invalid_argument("Task " + task_name + " not found in RobotController::getMotionForceTaskByName\n");
	}
	B_::RobotController& batched() { return *_c; }

private:
	shared_ptr<RobotModel_> _robot;
	vector<shared_ptr<TemplateTask>> _tasks;
	std::unique_ptr<B_::RobotController> _c;
};

}  // namespace Sai2Primitives

#endif /* SAI2_PRIMITIVES_EIGEN_H_ */
