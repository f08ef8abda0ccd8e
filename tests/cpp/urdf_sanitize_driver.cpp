// Test-only driver (tests/test_sanitized_host.py): feeds URDF files — the reference's, ours and malformed variants of
// both — to the product's URDF loader (csrc/sai2b_urdf.cpp, compiled with -fsanitize=address,undefined next to this
// file) as a file name and as XML text. A loader that reads files a user supplies must refuse bad input with a reason,
// never read out of bounds or overflow. Prints one line per input: "<rc> <dof> <message>"; exits 0 unless a load
// both "succeeds" and returns an unusable model.
#include <cmath>
#include <cstdio>
#include <fstream>
#include <sstream>
#include <string>

#include "../../include/sai2b.h"

static std::string g_msg;
extern "C" int sai2b_set_global_error(int code, const char* msg) {  // lives in sai2b_dispatch.cpp in the library
	g_msg = msg ? msg : "";
	return code;
}

static bool finite_model(const sai2b_robot_model& m) {
	if (m.dof != 4 && m.dof != 6 && m.dof != 7 && m.dof != 8) return false;
	for (int i = 0; i < m.dof; i++) {
		for (int k = 0; k < 3; k++)
			if (!std::isfinite(m.joint_xyz[i][k]) || !std::isfinite(m.joint_rpy[i][k]) || !std::isfinite(m.link_com[i][k])) return false;
		for (int k = 0; k < 6; k++)
			if (!std::isfinite(m.link_inertia[i][k])) return false;
		if (!std::isfinite(m.link_mass[i])) return false;
	}
	return true;
}

int main(int argc, char** argv) {
	int bad = 0;
	for (int a = 1; a < argc; a++) {
		std::ifstream f(argv[a]);
		std::stringstream ss;
		ss << f.rdbuf();
		const std::string text = ss.str();
		for (int is_file = 0; is_file < 2; is_file++) {
			sai2b_robot_model model;
			sai2b_urdf_links links;
			g_msg.clear();
			const int rc = sai2b_model_from_urdf(is_file ? argv[a] : text.c_str(), is_file, &model, &links);
			const std::string msg = g_msg;
			if (rc == SAI2B_OK) {
				if (!finite_model(model) || links.n_links < 1 || links.n_links > SAI2B_URDF_MAX_LINKS) bad++;
				// the calls a task constructor makes on a loaded robot
				int ml = -1;
				double fp[3], fr[9];
				const double zero[3] = {0, 0, 0};
				(void)sai2b_urdf_resolve_frame(&links, links.name[links.n_links - 1], zero, nullptr, &ml, fp, fr);
				(void)sai2b_urdf_resolve_frame(&links, "no-such-link", zero, nullptr, &ml, fp, fr);
				const double pos[3] = {0.1, -0.2, 0.3}, rot[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
				(void)sai2b_model_set_base_transform(&model, pos, rot);
			}
			std::printf("%d %d %s\n", rc, rc == SAI2B_OK ? model.dof : -1, msg.c_str());
		}
	}
	return bad ? 1 : 0;
}
