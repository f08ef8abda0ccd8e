// sai2b_urdf.cpp — host-side URDF -> constant-table converter.
//
// The reference loads its robots from URDF through sai2-model / RBDL's urdfreader (e.g.
// examples/05-using_robot_controller/05-using_robot_controller.cpp:45-47,96-97 with
// examples/15-haptic_control_impedance_type/panda_arm.urdf). This file reads the same files into the
// plain sai2b_robot_model the kernels take, and reports, for every URDF link, the moving link it is
// rigidly attached to and its fixed transform there — what a task needs to resolve a link *name* plus
// a position in that link (MotionForceTask.h:96-101: link_name "end-effector", a body on a fixed joint
// of link7) into (moving link index, compliant frame).
//
// Scope: one serial chain of 4, 6, 7 or 8 revolute / continuous / prismatic joints (the sizes the library is built
// for) with any <axis>, any number of fixed joints anywhere (bodies behind them are merged into the moving link
// they hang on, as RBDL does), <inertial> origins with rotation. The kernels move every joint about / along the z
// axis of its joint frame: a joint with another axis gets its frame rotated by A (A z = axis) and everything
// that hangs on the child link is re-expressed in the rotated link frame (sai2b.h: sai2b_robot_model).
// Anything else is refused with a message. No XML library: URDF needs elements, attributes and comments only.
#include <cctype>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <map>
#include <sstream>
#include <string>
#include <vector>

#include "../../include/sai2b.h"
#include "sai2b_model_host.h"

extern "C" int sai2b_set_global_error(int code, const char* msg);  // sai2b_host.cpp

namespace {

struct Element {
	std::string name;
	std::map<std::string, std::string> attr;
	int parent = -1;
	std::vector<int> children;
};

struct Parser {
	const std::string& s;
	size_t i = 0;
	std::vector<Element> nodes;
	std::string error;
	explicit Parser(const std::string& text) : s(text) {}

	void skip_ws() {
		while (i < s.size() && std::isspace((unsigned char)s[i])) i++;
	}
	bool starts(const char* lit) const { return s.compare(i, std::strlen(lit), lit) == 0; }
	std::string ident() {
		size_t j = i;
		while (j < s.size() && (std::isalnum((unsigned char)s[j]) || s[j] == '_' || s[j] == '-' || s[j] == ':' || s[j] == '.')) j++;
		std::string r = s.substr(i, j - i);
		i = j;
		return r;
	}
	bool parse() {
		std::vector<int> stack;
		while (i < s.size()) {
			if (s[i] != '<') {	// character data: ignored
				i++;
				continue;
			}
			if (starts("<!--")) {
				const size_t e = s.find("-->", i + 4);
				if (e == std::string::npos) return fail("unterminated comment");
				i = e + 3;
				continue;
			}
			if (starts("<?")) {
				const size_t e = s.find("?>", i + 2);
				if (e == std::string::npos) return fail("unterminated processing instruction");
				i = e + 2;
				continue;
			}
			if (starts("<!")) {	 // DOCTYPE etc.
				const size_t e = s.find('>', i);
				if (e == std::string::npos) return fail("unterminated declaration");
				i = e + 1;
				continue;
			}
			if (starts("</")) {
				i += 2;
				const std::string name = ident();
				skip_ws();
				if (i >= s.size() || s[i] != '>') return fail("malformed closing tag </" + name);
				i++;
				if (stack.empty() || nodes[stack.back()].name != name) return fail("mismatched closing tag </" + name + ">");
				stack.pop_back();
				continue;
			}
			i++;  // '<'
			Element el;
			el.name = ident();
			if (el.name.empty()) return fail("malformed tag");
			el.parent = stack.empty() ? -1 : stack.back();
			bool self_closing = false;
			for (;;) {
				skip_ws();
				if (i >= s.size()) return fail("unterminated tag <" + el.name);
				if (s[i] == '>') {
					i++;
					break;
				}
				if (starts("/>")) {
					i += 2;
					self_closing = true;
					break;
				}
				const std::string key = ident();
				skip_ws();
				if (key.empty() || i >= s.size() || s[i] != '=') return fail("malformed attribute in <" + el.name + ">");
				i++;
				skip_ws();
				if (i >= s.size() || (s[i] != '"' && s[i] != '\'')) return fail("unquoted attribute value in <" + el.name + ">");
				const char quote = s[i++];
				const size_t e = s.find(quote, i);
				if (e == std::string::npos) return fail("unterminated attribute value in <" + el.name + ">");
				el.attr[key] = s.substr(i, e - i);
				i = e + 1;
			}
			const int id = (int)nodes.size();
			nodes.push_back(el);
			if (el.parent >= 0) nodes[el.parent].children.push_back(id);
			if (!self_closing) stack.push_back(id);
		}
		if (!stack.empty()) return fail("unclosed element <" + nodes[stack.back()].name + ">");
		return true;
	}
	bool fail(const std::string& m) {
		error = "URDF: " + m;
		return false;
	}
	int child(int node, const char* name) const {
		for (int c : nodes[node].children)
			if (nodes[c].name == name) return c;
		return -1;
	}
};

// n whitespace-separated numbers. Each token is read like std::stod does (a numeric prefix, the rest ignored):
// examples/06-partial_joint_task/panda_arm_sliding_base.urdf:172 has xyz="0 0 0.-75", which the reference's
// reader takes as 0 0 0.
bool numbers(const std::string& text, int n, double* out) {
	std::istringstream ss(text);
	std::string tok;
	int k = 0;
	while (ss >> tok) {
		if (k >= n) return false;
		char* end = nullptr;
		out[k] = std::strtod(tok.c_str(), &end);
		// (strtod also reads "nan", "inf" and overflows to HUGE_VAL: a model with such an entry would put NaNs into every
		// robot's torques — found by tests/test_sanitized_host.py)
		if (end == tok.c_str() || !std::isfinite(out[k])) return false;
		k++;
	}
	return k == n;
}

struct Inertial {
	bool present = false;
	double mass = 0, com[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}, I6[6] = {0, 0, 0, 0, 0, 0};  // ixx iyy izz ixy ixz iyz
};
struct Link {
	std::string name;
	Inertial in;
};
struct Joint {
	std::string name, type, parent, child;
	double xyz[3] = {0, 0, 0}, rpy[3] = {0, 0, 0}, axis[3] = {1, 0, 0};	 // URDF default axis is x
	bool has_limit = false;
	double lower = 0, upper = 0, effort = 0;
};

void mat_mul(const double* A, const double* B, double* C) {
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double v = 0;
			for (int k = 0; k < 3; k++) v += A[3 * i + k] * B[3 * k + j];
			C[3 * i + j] = v;
		}
}

// inertia given in a frame rotated by rpy, expressed in the link axes: R I R^T (6-vector in, 6 out)
void rotate_inertia(const double* I6, const double* rpy, double* out6) {
	if (rpy[0] == 0 && rpy[1] == 0 && rpy[2] == 0) {
		std::memcpy(out6, I6, 6 * sizeof(double));
		return;
	}
	double R[9], I[9], T[9], W[9];
	sai2b::rot_from_rpy(rpy, R);
	sai2b::sym3_from6(I6, I);
	mat_mul(R, I, T);
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) {
			double v = 0;
			for (int k = 0; k < 3; k++) v += T[3 * i + k] * R[3 * j + k];
			W[3 * i + j] = v;
		}
	out6[0] = W[0], out6[1] = W[4], out6[2] = W[8], out6[3] = W[1], out6[4] = W[2], out6[5] = W[5];
}

int fail(const std::string& m) { return sai2b_set_global_error(SAI2B_INVALID_ARGUMENT, m.c_str()); }

// rotation A with A z = a (a unit): identity for a = z, a half turn about x for a = -z, else the rotation about z x a
void axis_frame(const double* a, double* A) {
	const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	std::memcpy(A, I, sizeof(I));
	if (a[0] == 0 && a[1] == 0 && a[2] == 1) return;
	if (a[0] == 0 && a[1] == 0 && a[2] == -1) {
		A[4] = A[8] = -1;
		return;
	}
	const double v[3] = {-a[1], a[0], 0.0}, c = a[2], s2 = v[0] * v[0] + v[1] * v[1];  // v = z x a
	const double K[9] = {0, -v[2], v[1], v[2], 0, -v[0], -v[1], v[0], 0};
	double K2[9];
	mat_mul(K, K, K2);
	for (int i = 0; i < 9; i++) A[i] = I[i] + K[i] + K2[i] * (1 - c) / s2;
}
// rpy with rot_from_rpy(rpy) == R (R = Rz(y) Ry(p) Rx(r))
void rpy_from_rot(const double* R, double* rpy) {
	const double cp = std::sqrt(R[0] * R[0] + R[3] * R[3]);
	rpy[1] = std::atan2(-R[6], cp);
	if (cp < 1e-9) {
		// pitch = +-90 degrees (an x axis folded into z, or two folded axes in a row): roll and yaw turn about the
		// same line and only their difference / sum is defined — all of it goes to the roll
		rpy[2] = 0.0;
		rpy[0] = std::atan2(R[6] < 0 ? R[1] : -R[1], R[4]);
		return;
	}
	rpy[2] = std::atan2(R[3], R[0]);
	rpy[0] = std::atan2(R[7], R[8]);
}
bool is_identity(const double* R) {
	for (int i = 0; i < 9; i++)
		if (R[i] != (i % 4 == 0 ? 1.0 : 0.0)) return false;
	return true;
}

}  // namespace

extern "C" int sai2b_model_from_urdf(const char* urdf, int is_file, sai2b_robot_model* model, sai2b_urdf_links* links) {
	constexpr int N = SAI2B_MAX_DOF;
	if (!urdf || !model) return fail("sai2b_model_from_urdf: null argument");
	std::string text;
	if (is_file) {
		std::ifstream f(urdf);
		if (!f) return fail(std::string("URDF: cannot open ") + urdf);
		std::stringstream ss;
		ss << f.rdbuf();
		text = ss.str();
	} else {
		text = urdf;
	}
	Parser px(text);
	if (!px.parse()) return fail(px.error);
	int robot = -1;
	for (size_t k = 0; k < px.nodes.size(); k++)
		if (px.nodes[k].name == "robot" && px.nodes[k].parent < 0) robot = (int)k;
	if (robot < 0) return fail("URDF: no <robot> element");

	std::vector<Link> lk;
	std::vector<Joint> jt;
	for (int c : px.nodes[robot].children) {
		const Element& e = px.nodes[c];
		if (e.name == "link") {
			Link l;
			if (!e.attr.count("name")) return fail("URDF: <link> without a name");
			l.name = e.attr.at("name");
			// (sai2b_urdf_links holds 63 characters + NUL per name: a longer one would be cut and could then collide)
			if (l.name.empty() || l.name.size() >= sizeof(sai2b_urdf_links{}.name[0])) return fail("URDF: link name empty or longer than 63 characters");
			for (const Link& other : lk)
				if (other.name == l.name) return fail("URDF: two links named " + l.name);
			const int in = px.child(c, "inertial");
			if (in >= 0) {
				l.in.present = true;
				const int o = px.child(in, "origin"), m = px.child(in, "mass"), I = px.child(in, "inertia");
				if (o >= 0) {
					const auto& a = px.nodes[o].attr;
					if (a.count("xyz") && !numbers(a.at("xyz"), 3, l.in.com)) return fail("URDF: bad inertial origin xyz of link " + l.name);
					if (a.count("rpy") && !numbers(a.at("rpy"), 3, l.in.rpy)) return fail("URDF: bad inertial origin rpy of link " + l.name);
				}
				if (m < 0 || !px.nodes[m].attr.count("value") || !numbers(px.nodes[m].attr.at("value"), 1, &l.in.mass))
					return fail("URDF: link " + l.name + " has an <inertial> without a mass");
				if (I >= 0) {
					const char* keys[6] = {"ixx", "iyy", "izz", "ixy", "ixz", "iyz"};
					for (int k = 0; k < 6; k++) {
						const auto& a = px.nodes[I].attr;
						if (a.count(keys[k]) && !numbers(a.at(keys[k]), 1, &l.in.I6[k])) return fail("URDF: bad inertia of link " + l.name);
					}
				}
			}
			lk.push_back(l);
		} else if (e.name == "joint") {
			Joint j;
			if (!e.attr.count("name") || !e.attr.count("type")) return fail("URDF: <joint> without name or type");
			j.name = e.attr.at("name"), j.type = e.attr.at("type");
			const int p = px.child(c, "parent"), ch = px.child(c, "child"), o = px.child(c, "origin"), ax = px.child(c, "axis"),
					  lim = px.child(c, "limit");
			if (p < 0 || ch < 0 || !px.nodes[p].attr.count("link") || !px.nodes[ch].attr.count("link"))
				return fail("URDF: joint " + j.name + " needs <parent link> and <child link>");
			j.parent = px.nodes[p].attr.at("link"), j.child = px.nodes[ch].attr.at("link");
			if (o >= 0) {
				const auto& a = px.nodes[o].attr;
				if (a.count("xyz") && !numbers(a.at("xyz"), 3, j.xyz)) return fail("URDF: bad origin xyz of joint " + j.name);
				if (a.count("rpy") && !numbers(a.at("rpy"), 3, j.rpy)) return fail("URDF: bad origin rpy of joint " + j.name);
			}
			if (ax >= 0 && px.nodes[ax].attr.count("xyz") && !numbers(px.nodes[ax].attr.at("xyz"), 3, j.axis))
				return fail("URDF: bad axis of joint " + j.name);
			if (lim >= 0) {
				const auto& a = px.nodes[lim].attr;
				j.has_limit = true;
				for (const auto& kv : {std::make_pair("lower", &j.lower), std::make_pair("upper", &j.upper), std::make_pair("effort", &j.effort)})
					if (a.count(kv.first) && !numbers(a.at(kv.first), 1, kv.second)) return fail("URDF: bad limit of joint " + j.name);
			}
			jt.push_back(j);
		}
	}
	auto link_index = [&](const std::string& name) {
		for (size_t k = 0; k < lk.size(); k++)
			if (lk[k].name == name) return (int)k;
		return -1;
	};
	std::vector<int> parent_joint(lk.size(), -1);
	for (size_t k = 0; k < jt.size(); k++) {
		const int c = link_index(jt[k].child), p = link_index(jt[k].parent);
		if (c < 0 || p < 0) return fail("URDF: joint " + jt[k].name + " refers to an unknown link");
		if (parent_joint[c] >= 0) return fail("URDF: link " + jt[k].child + " has two parent joints");
		parent_joint[c] = (int)k;
	}
	int root = -1;
	for (size_t k = 0; k < lk.size(); k++)
		if (parent_joint[k] < 0) {
			if (root >= 0) return fail("URDF: more than one root link (" + lk[root].name + ", " + lk[k].name + ")");
			root = (int)k;
		}
	if (root < 0) return fail("URDF: no root link");

	// walk the tree from the root: every link gets (moving link it is rigidly attached to, fixed
	// transform there); a revolute joint starts a new moving link
	std::memset(model, 0, sizeof(*model));
	model->gravity[2] = -9.81;	// Sai2Model's default world gravity
	struct Placed {
		int moving;	 // -1: fixed to the world
		double R[9], p[3];
		int depth;	 // fixed joints between the moving link and this one
		double rpy1[3];	 // the rpy of that joint when depth == 1 (exact re-use of the model's own merge)
		bool exact = false;	 // depth == 1 on a moving link whose own frame is not re-oriented: R == rot(rpy1), p == xyz bit for bit
	};
	std::vector<Placed> placed(lk.size());
	std::vector<int> order = {root};
	placed[root] = Placed{-1, {1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}, 0, {0, 0, 0}, false};
	int n_moving = 0;
	for (size_t head = 0; head < order.size(); head++) {
		const int cur = order[head];
		for (size_t k = 0; k < jt.size(); k++) {
			if (link_index(jt[k].parent) != cur) continue;
			const Joint& j = jt[k];
			const int c = link_index(j.child);
			if (j.type == "fixed") {
				Placed pl = placed[cur];
				double Rj[9], Rn[9];
				sai2b::rot_from_rpy(j.rpy, Rj);
				mat_mul(placed[cur].R, Rj, Rn);
				for (int a = 0; a < 3; a++)
					pl.p[a] = placed[cur].p[a] + placed[cur].R[3 * a] * j.xyz[0] + placed[cur].R[3 * a + 1] * j.xyz[1] + placed[cur].R[3 * a + 2] * j.xyz[2];
				std::memcpy(pl.R, Rn, sizeof(Rn));
				pl.depth = placed[cur].depth + 1;
				pl.exact = placed[cur].depth == 0 && is_identity(placed[cur].R) && placed[cur].p[0] == 0 && placed[cur].p[1] == 0 && placed[cur].p[2] == 0;
				std::memcpy(pl.rpy1, j.rpy, sizeof(pl.rpy1));
				placed[c] = pl;
			} else if (j.type == "revolute" || j.type == "continuous" || j.type == "prismatic") {
				if (placed[cur].moving != n_moving - 1)
					return fail("URDF: the moving joints do not form one serial chain (branch at joint " + j.name + ")");
				if (n_moving >= N) return fail("URDF: more than " + std::to_string(N) + " moving joints (the largest robot this library is built for)");
				double an = std::sqrt(j.axis[0] * j.axis[0] + j.axis[1] * j.axis[1] + j.axis[2] * j.axis[2]);
				if (!(an > 1e-9)) return fail("URDF: joint " + j.name + " has a zero axis");
				const double axis[3] = {j.axis[0] / an, j.axis[1] / an, j.axis[2] / an};
				double A[9];
				axis_frame(axis, A);
				const int i = n_moving++;
				const Placed& par = placed[cur];
				const bool plain = par.depth == 0 && is_identity(par.R) && par.p[0] == 0 && par.p[1] == 0 && par.p[2] == 0 && is_identity(A);
				if (plain) {  // the file's own numbers, bit for bit
					for (int a = 0; a < 3; a++) model->joint_xyz[i][a] = j.xyz[a], model->joint_rpy[i][a] = j.rpy[a];
				} else {  // joint frame in the (possibly re-oriented) parent link frame, turned so that the axis is its z
					double Rj[9], T[9], E[9];
					sai2b::rot_from_rpy(j.rpy, Rj);
					mat_mul(par.R, Rj, T);
					mat_mul(T, A, E);
					rpy_from_rot(E, model->joint_rpy[i]);
					for (int a = 0; a < 3; a++)
						model->joint_xyz[i][a] = par.p[a] + par.R[3 * a] * j.xyz[0] + par.R[3 * a + 1] * j.xyz[1] + par.R[3 * a + 2] * j.xyz[2];
				}
				model->joint_type[i] = j.type == "prismatic" ? SAI2B_PRISMATIC : SAI2B_REVOLUTE;
				model->q_lower[i] = j.type == "continuous" ? -1e30 : j.lower;
				model->q_upper[i] = j.type == "continuous" ? 1e30 : j.upper;
				model->effort[i] = j.effort;
				// the child link's own frame in the model's link frame: A^T (identity for a z-axis joint)
				Placed me{i, {A[0], A[3], A[6], A[1], A[4], A[7], A[2], A[5], A[8]}, {0, 0, 0}, 0, {0, 0, 0}, false};
				if (lk[c].in.present) {
					model->link_mass[i] = lk[c].in.mass;
					if (is_identity(A)) {
						for (int a = 0; a < 3; a++) model->link_com[i][a] = lk[c].in.com[a];
						rotate_inertia(lk[c].in.I6, lk[c].in.rpy, model->link_inertia[i]);
					} else {
						double I6[6], I[9], T[9], W[9];
						rotate_inertia(lk[c].in.I6, lk[c].in.rpy, I6);
						sai2b::sym3_from6(I6, I);
						mat_mul(me.R, I, T);
						for (int a = 0; a < 3; a++)
							for (int b = 0; b < 3; b++) {
								double v = 0;
								for (int k2 = 0; k2 < 3; k2++) v += T[3 * a + k2] * me.R[3 * b + k2];
								W[3 * a + b] = v;
							}
						double* o = model->link_inertia[i];
						o[0] = W[0], o[1] = W[4], o[2] = W[8], o[3] = W[1], o[4] = W[2], o[5] = W[5];
						for (int a = 0; a < 3; a++)
							model->link_com[i][a] = me.R[3 * a] * lk[c].in.com[0] + me.R[3 * a + 1] * lk[c].in.com[1] + me.R[3 * a + 2] * lk[c].in.com[2];
					}
				}
				placed[c] = me;
			} else {
				return fail("URDF: joint " + j.name + " has type \"" + j.type + "\" (revolute, continuous, prismatic and fixed are supported)");
			}
			order.push_back(c);
		}
	}
	if (order.size() != lk.size()) return fail("URDF: some links are not connected to the root");
	if (n_moving != 4 && n_moving != 6 && n_moving != 7 && n_moving != 8)
		return fail("URDF: " + std::to_string(n_moving) + " moving joints; this library is built for robots with 4, 6, 7 or 8");
	model->dof = n_moving;
	// bodies behind fixed joints are merged into the moving link they hang on (in file order of the
	// tree walk), as RBDL's urdfreader does; bodies fixed to the world carry no dynamics
	for (size_t h = 1; h < order.size(); h++) {
		const int l = order[h];
		const Placed& pl = placed[l];
		if (pl.depth == 0 || pl.moving < 0 || !lk[l].in.present) continue;
		double I6[6];
		rotate_inertia(lk[l].in.I6, lk[l].in.rpy, I6);
		if (pl.depth == 1 && pl.exact) {
			sai2b::host_merge_fixed_body(model, pl.moving, pl.p, pl.rpy1, lk[l].in.mass, lk[l].in.com, I6);
		} else {
			// deeper chains: express the body in the frame of the first fixed link (identity rpy) by
			// rotating its COM and inertia with the accumulated rotation
			double I[9], T[9], W[9], com[3];
			sai2b::sym3_from6(I6, I);
			mat_mul(pl.R, I, T);
			for (int i = 0; i < 3; i++)
				for (int j = 0; j < 3; j++) {
					double v = 0;
					for (int k = 0; k < 3; k++) v += T[3 * i + k] * pl.R[3 * j + k];
					W[3 * i + j] = v;
				}
			for (int a = 0; a < 3; a++) com[a] = pl.R[3 * a] * lk[l].in.com[0] + pl.R[3 * a + 1] * lk[l].in.com[1] + pl.R[3 * a + 2] * lk[l].in.com[2];
			const double zero[3] = {0, 0, 0}, Iw6[6] = {W[0], W[4], W[8], W[1], W[2], W[5]};
			sai2b::host_merge_fixed_body(model, pl.moving, pl.p, zero, lk[l].in.mass, com, Iw6);
		}
	}
	if (links) {
		std::memset(links, 0, sizeof(*links));
		if (lk.size() > SAI2B_URDF_MAX_LINKS) return fail("URDF: more links than SAI2B_URDF_MAX_LINKS");
		links->n_links = (int)lk.size();
		for (size_t k = 0; k < lk.size(); k++) {
			std::snprintf(links->name[k], sizeof(links->name[k]), "%s", lk[k].name.c_str());
			links->moving_link[k] = placed[k].moving;
			for (int a = 0; a < 3; a++) links->pos[k][a] = placed[k].p[a];
			for (int a = 0; a < 9; a++) links->rot[k][a] = placed[k].R[a];
		}
	}
	return SAI2B_OK;
}

// Sai2Model::setTRobotBase: T_world_base folded into the first joint's origin (the chain then starts in the world
// frame, where the tasks work and where the model's gravity vector lives)
extern "C" int sai2b_model_set_base_transform(sai2b_robot_model* model, const double pos[3], const double* rot) {
	if (!model || !pos) return fail("sai2b_model_set_base_transform: null argument");
	if (model->dof < 1 || model->dof > SAI2B_MAX_DOF) return fail("sai2b_model_set_base_transform: the model has no joints");
	const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
	const double* Rb = rot ? rot : I;
	double RtR[9];
	for (int i = 0; i < 3; i++)
		for (int j = 0; j < 3; j++) RtR[3 * i + j] = Rb[i] * Rb[j] + Rb[3 + i] * Rb[3 + j] + Rb[6 + i] * Rb[6 + j];
	const double det = Rb[0] * (Rb[4] * Rb[8] - Rb[5] * Rb[7]) - Rb[1] * (Rb[3] * Rb[8] - Rb[5] * Rb[6]) + Rb[2] * (Rb[3] * Rb[7] - Rb[4] * Rb[6]);
	for (int i = 0; i < 9; i++)
		if (!(std::fabs(RtR[i] - I[i]) < 1e-9) || !(det > 0) || !std::isfinite(pos[i % 3]))
			return fail("sai2b_model_set_base_transform: the base orientation is not a rotation matrix (or the position is not finite)");
	double R0[9], R1[9];
	sai2b::rot_from_rpy(model->joint_rpy[0], R0);
	mat_mul(Rb, R0, R1);
	const double* x = model->joint_xyz[0];
	const double p1[3] = {pos[0] + Rb[0] * x[0] + Rb[1] * x[1] + Rb[2] * x[2], pos[1] + Rb[3] * x[0] + Rb[4] * x[1] + Rb[5] * x[2],
						  pos[2] + Rb[6] * x[0] + Rb[7] * x[1] + Rb[8] * x[2]};
	for (int a = 0; a < 3; a++) model->joint_xyz[0][a] = p1[a];
	rpy_from_rot(R1, model->joint_rpy[0]);
	return SAI2B_OK;
}

// link name + position/orientation in that link -> moving link index + compliant frame in it
// (MotionForceTask.h:96-101 takes a link name and an Affine3d compliant frame)
extern "C" int sai2b_urdf_resolve_frame(const sai2b_urdf_links* links, const char* link_name, const double pos_in_link[3],
										const double* rot_in_link, int* moving_link, double frame_pos[3], double frame_rot[9]) {
	if (!links || !link_name || !moving_link || !frame_pos) return fail("sai2b_urdf_resolve_frame: null argument");
	for (int k = 0; k < links->n_links; k++) {
		if (std::strcmp(links->name[k], link_name) != 0) continue;
		if (links->moving_link[k] < 0) return fail(std::string("link ") + link_name + " is fixed to the world: no task can control it");
		*moving_link = links->moving_link[k];
		const double* R = links->rot[k];
		const double zero[3] = {0, 0, 0};
		const double* p = pos_in_link ? pos_in_link : zero;
		for (int a = 0; a < 3; a++) frame_pos[a] = links->pos[k][a] + R[3 * a] * p[0] + R[3 * a + 1] * p[1] + R[3 * a + 2] * p[2];
		if (frame_rot) {
			const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
			mat_mul(R, rot_in_link ? rot_in_link : I, frame_rot);
		}
		return SAI2B_OK;
	}
	return fail(std::string("link ") + link_name + " not found in the robot description");
}
