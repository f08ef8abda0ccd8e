"""The C++ facade (include/Sai2PrimitivesBatched.h) compiled with g++ against the C ABI: argument
checks on CPU, one tick against the oracle on the GPU."""
import os
import subprocess

import numpy as np
import pytest

import sai2_primitives_perso_amd as pkg

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "sai2-primitives-perso_amd", "csrc")


@pytest.fixture(scope="module")
def facade_bin(tmp_path_factory):
    pkg._abi.load_library()  # make sure it exists
    out = str(tmp_path_factory.mktemp("cpp") / "facade_test")
    subprocess.run(
        ["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "facade_test.cpp"),
         "-o", out, "-L", CSRC, "-lsai2b", f"-Wl,-rpath,{CSRC}", "-Wl,-rpath,/opt/rocm/lib"],
        check=True,
    )
    return out


def test_cpp_facade_argument_checks(facade_bin):
    r = subprocess.run([facade_bin, "validate"], capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "validate: ok" in r.stdout


@pytest.mark.gpu
def test_cpp_facade_tick_matches_oracle(facade_bin, tmp_path):
    import oracle_lib as ol

    B = 256
    inp = pkg.workloads.make_inputs(3, B=B, seed=99)
    g = inp["mft0"]
    blob = np.concatenate([inp["q"].ravel(), inp["dq"].ravel(), g["pos"].ravel(), g["rot"].ravel(), g["v"].ravel(), g["w"].ravel(),
                           g["a"].ravel(), g["alpha"].ravel(), inp["jt1"]["q"].ravel()])
    path = tmp_path / "in.bin"
    blob.astype(np.float64).tofile(path)
    r = subprocess.run([facade_bin, "tick", str(B), str(path)], capture_output=True)
    assert r.returncode == 0, r.stderr.decode()
    out = np.frombuffer(r.stdout, dtype=np.float64).reshape(3, 7, B)
    tau, q1, dq1 = out
    o = ol.Oracle(ol.panda_model(), ol.task_configs(inp["tasks"]), B)
    ol.load_inputs(o, inp)
    ref = o.tick()
    assert (np.abs(tau - ref).max(axis=0) / np.maximum(np.abs(ref).max(axis=0), 1)).max() < 1e-10
    # BatchedSimulation::integrate() consumed those torques on the device (2 sub-steps of 0.5 ms)
    o.sim_step(ref, 0.001, substeps=2)
    qo, vo = o.get_state()
    assert np.abs(q1 - qo).max() < 1e-12 and np.abs(dq1 - vo).max() < 1e-10
