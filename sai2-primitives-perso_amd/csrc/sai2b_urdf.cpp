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
// Scope of this build: one serial chain of exactly SAI2B_DOF revolute joints about their local z axis
// ("0 0 1": the convention of every robot file in the reference's examples), any number of fixed
// joints (bodies behind them are merged into their parent, as RBDL does), <inertial> origins with
// rotation. Anything else is refused with a message. No XML library: URDF needs elements, attributes
// and comments only.
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

bool numbers(const std::string& text, int n, double* out) {
	const char* p = text.c_str();
	for (int k = 0; k < n; k++) {
		char* end = nullptr;
		out[k] = std::strtod(p, &end);
		if (end == p) return false;
		p = end;
	}
	while (*p && std::isspace((unsigned char)*p)) p++;
	return *p == 0;
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

}  // namespace

extern "C" int sai2b_model_from_urdf(const char* urdf, int is_file, sai2b_robot_model* model, sai2b_urdf_links* links) {
	constexpr int N = SAI2B_DOF;
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
				if (a.count("lower")) numbers(a.at("lower"), 1, &j.lower);
				if (a.count("upper")) numbers(a.at("upper"), 1, &j.upper);
				if (a.count("effort")) numbers(a.at("effort"), 1, &j.effort);
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
	model->dof = N;
	model->gravity[2] = -9.81;	// Sai2Model's default world gravity
	struct Placed {
		int moving;	 // -1: fixed to the world
		double R[9], p[3];
		int depth;	 // fixed joints between the moving link and this one
		double rpy1[3];	 // the rpy of that joint when depth == 1 (exact re-use of the model's own merge)
	};
	std::vector<Placed> placed(lk.size());
	std::vector<int> order = {root};
	placed[root] = Placed{-1, {1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}, 0, {0, 0, 0}};
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
				std::memcpy(pl.rpy1, j.rpy, sizeof(pl.rpy1));
				placed[c] = pl;
			} else if (j.type == "revolute" || j.type == "continuous") {
				if (!(j.axis[0] == 0 && j.axis[1] == 0 && j.axis[2] == 1))
					return fail("URDF: joint " + j.name + " does not turn about its local z axis (only \"0 0 1\" is supported in this build)");
				if (placed[cur].depth != 0)
					return fail("URDF: joint " + j.name + " hangs on a link behind a fixed joint (not supported in this build)");
				if (placed[cur].moving != n_moving - 1)
					return fail("URDF: the moving joints do not form one serial chain (branch at joint " + j.name + ")");
				if (n_moving >= N) return fail("URDF: more than " + std::to_string(N) + " moving joints (this build is compiled for 7)");
				const int i = n_moving++;
				for (int a = 0; a < 3; a++) model->joint_xyz[i][a] = j.xyz[a], model->joint_rpy[i][a] = j.rpy[a];
				model->q_lower[i] = j.type == "continuous" ? -1e30 : j.lower;
				model->q_upper[i] = j.type == "continuous" ? 1e30 : j.upper;
				model->effort[i] = j.effort;
				if (lk[c].in.present) {
					model->link_mass[i] = lk[c].in.mass;
					for (int a = 0; a < 3; a++) model->link_com[i][a] = lk[c].in.com[a];
					rotate_inertia(lk[c].in.I6, lk[c].in.rpy, model->link_inertia[i]);
				}
				placed[c] = Placed{i, {1, 0, 0, 0, 1, 0, 0, 0, 1}, {0, 0, 0}, 0, {0, 0, 0}};
			} else {
				return fail("URDF: joint " + j.name + " has type \"" + j.type + "\" (only revolute, continuous and fixed are supported in this build)");
			}
			order.push_back(c);
		}
	}
	if (order.size() != lk.size()) return fail("URDF: some links are not connected to the root");
	if (n_moving != N) return fail("URDF: " + std::to_string(n_moving) + " moving joints; this build is compiled for exactly " + std::to_string(N));
	// bodies behind fixed joints are merged into the moving link they hang on (in file order of the
	// tree walk), as RBDL's urdfreader does; bodies fixed to the world carry no dynamics
	for (size_t h = 1; h < order.size(); h++) {
		const int l = order[h];
		const Placed& pl = placed[l];
		if (pl.depth == 0 || pl.moving < 0 || !lk[l].in.present) continue;
		double I6[6];
		rotate_inertia(lk[l].in.I6, lk[l].in.rpy, I6);
		if (pl.depth == 1) {
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
