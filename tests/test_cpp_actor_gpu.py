"""The C++ host surface without Python in the loop: tests/cpp/actor_parity.cpp drives nbody::OctreeSearchActor exactly as
the reference's Blueprints drive AOctreeSearch and compares every frame with the oracle, all in C++."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "actor_parity")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "actor_parity.cpp"), "-o", exe,
                           "-L", os.path.join(ROOT, "parallelnbody_amd"), "-lnbody_amd",
                           "-L", os.path.join(ROOT, "oracle"), "-lnbody_oracle",
                           "-Wl,-rpath," + os.path.join(ROOT, "parallelnbody_amd"), "-Wl,-rpath," + os.path.join(ROOT, "oracle")])
    return exe


def test_cpp_actor_program_links_against_the_c_abi(nb, oracle, tmp_path):
    # CPU half: the header-only actor + the C-ABI library + the oracle link into a plain g++ program
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    if nb.device_count() == 0:
        assert out.returncode == 2 and "no HIP device" in out.stdout


@pytest.mark.gpu
def test_cpp_actor_parity_program(nb, oracle, tmp_path):
    exe = _build(tmp_path)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "actor parity: ok" in out.stdout


@pytest.mark.gpu
def test_plain_c_host_program(nb, tmp_path):
    # the C shim (nbody_actor.h) from a C11 program: the host INTEGRATION.md sketches, sixty frames at theta = 1 and at 0
    exe = str(tmp_path / "actor_demo")
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "actor_demo.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "parallelnbody_amd"), "-lnbody_amd",
                           "-Wl,-rpath," + os.path.join(ROOT, "parallelnbody_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "actor demo: ok" in out.stdout
