"""nbody_create_multi — one context over several GPUs driven by one caller thread (RCCL between the devices, csrc/multi.hip).
The test box has one GPU: the communicator then has a single rank, every collective still runs, and the results must equal
an ordinary context's bit for bit.  tests/cpp/multi_parity.c uses every device it finds."""
import os
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def test_plain_c_host_over_every_visible_gpu(nb, tmp_path):
    exe = str(tmp_path / "multi_parity")
    subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "multi_parity.c"), "-o", exe,
                           "-L", os.path.join(ROOT, "parallelnbody_amd"), "-lnbody_amd", "-lm",
                           "-Wl,-rpath," + os.path.join(ROOT, "parallelnbody_amd")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=600)
    print(out.stdout)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "multi parity: ok" in out.stdout


@pytest.mark.parametrize("n,precision", [(40000, "f32"), (32768, "f32_kahan"), (8192, "f32")])
def test_one_device_multi_context_equals_a_plain_context(nb, n, precision):
    posm, vel = nb.ic_plummer(n, seed=9)
    with nb.NBodyEngine(n, precision=precision) as a, nb.NBodyEngine(n, precision=precision, devices=[0]) as b:
        assert a.launch_config() == b.launch_config()
        for e in (a, b):
            e.set_state(posm, vel)
            e.compute_forces()                     # accelerations only
            e.step(0.01, 3)
        for x, y in zip(a.state(), b.state()):
            np.testing.assert_array_equal(x, y)
        np.testing.assert_array_equal(a.positions(), b.positions())
        np.testing.assert_array_equal(a.particles(), b.particles())
        assert a.bounds() == b.bounds()
        assert a.energy() == b.energy()              # fixed-order reduction: the diagnostic is reproducible in every bit
        assert a.energy() == a.energy()
        assert a.steps_done() == b.steps_done() == 3


def test_multi_context_checkpoint_is_the_single_context_file(nb, tmp_path):
    n = 40000
    posm, vel = nb.ic_plummer(n, seed=11)
    path = str(tmp_path / "multi.ckpt")
    with nb.NBodyEngine(n, devices=[0]) as m, nb.NBodyEngine(n) as s:
        m.set_state(posm, vel); s.set_state(posm, vel)
        m.step(0.01, 2); s.step(0.01, 5)
        m.save_checkpoint(path)
        with nb.NBodyEngine(n) as r:                 # a plain context resumes the multi-device file ...
            assert r.load_checkpoint(path) == 2
            r.step(0.01, 3)
            for x, y in zip(r.state(), s.state()):
                np.testing.assert_array_equal(x, y)
        with nb.NBodyEngine(n, devices=[0]) as r:    # ... and so does a multi-device one
            assert r.load_checkpoint(path) == 2
            r.step(0.01, 3)
            for x, y in zip(r.state(), s.state()):
                np.testing.assert_array_equal(x, y)


def test_multi_context_argument_errors(nb):
    with pytest.raises(nb.NBodyError):
        nb.NBodyEngine(1000, devices=[0, 0])        # a device listed twice
    with pytest.raises(nb.NBodyError):
        nb.NBodyEngine(1000, devices=[0], theta=1.0, precision="f32_kahan")   # the reference's tree walk is plain fp32
    with pytest.raises(nb.NBodyError):
        nb.NBodyEngine(1001, devices=[0, 1])        # equal slices only (and there is one GPU here anyway)


@pytest.mark.parametrize("theta", [0.0, 1.0])
def test_actor_over_a_device_list(nb, theta):
    # the AOctreeSearch mirror on nbody_create_multi (real RCCL, one rank): same frames as on one device — also at the actor's
    # default, the reference's shipped opening angle (every device builds the whole tree and walks its slice: csrc/multi.hip)
    a, b = nb.OctreeSearch(), nb.OctreeSearch()
    b.set_devices([0])
    for act in (a, b):
        act.set_theta(theta)
        act.set_seed(5)
        act.CreateSpacePoints(2000, 1000.0)
        for _ in range(5):
            act.Tick(1.0 / 60)
        assert act.LastStatus == 0
    np.testing.assert_array_equal(a.Particles, b.Particles)
    assert a.Size == b.Size
