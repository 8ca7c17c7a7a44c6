"""BASELINE config #1 through the PRODUCT: `restir_app --example 4` = the 04_ao kernelMain (examples/04_ao/04_ao.cu:31-88)
as a host C++ loop (cedec_2024_rt_amd/csrc/host_path.h), cornellbox1.obj, 256x256, default camera — byte for byte the
reference's own kernel run on the host (tests/golden/ref_ao04_256.npz, made by tests/golden/make_golden.py from oracle/_ref).
No GPU call and nothing from oracle/: the test only reads the golden file. Runs in the CPU suite; the same function runs
again under the `gpu` marker (tests/test_host_checks_on_the_gpu_box.py) on the GPU box's own cores and libm."""
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
APP = os.path.join(ROOT, "app", "restir_app")


def run_config1(tmp_path, golden_dir, extra=()):
    g = np.load(os.path.join(golden_dir, "ref_ao04_256.npz"))
    out = os.path.join(str(tmp_path), "ao04.raw")
    cmd = [APP, "--example", "4", "--obj", os.path.join(golden_dir, "assets", "cornellbox1.obj"), "--rgba", out, *extra]
    env = dict(os.environ, HIP_VISIBLE_DEVICES="", ROCR_VISIBLE_DEVICES="")  # a host loop: it must not need a device
    p = subprocess.run(cmd, capture_output=True, text=True, timeout=600, env=env)
    assert p.returncode == 0, p.stderr
    assert "triangles: 36" in p.stdout and "04_ao 256x256" in p.stdout, p.stdout
    px = np.fromfile(out, np.uint8).reshape(256, 256, 4)
    assert np.array_equal(px, g["pixels"]), f"{int((px != g['pixels']).any(axis=2).sum())} pixels differ from the reference kernel"
    return px


def test_config1_04_ao_product_host_loop(tmp_path, golden_dir):
    px = run_config1(tmp_path, golden_dir)
    assert (px[..., 0] != 32).mean() > 0.1 and (px[..., 3] == 255).all()
    # the result does not depend on how the rows are dealt out to threads
    run_config1(tmp_path, golden_dir, extra=("--threads", "1") if (os.cpu_count() or 1) >= 4 else ())
    # explicit size / camera arguments = the defaults of 04_ao.cpp (common/misc.hpp:217-218)
    run_config1(tmp_path, golden_dir, extra=("--size", "256", "256", "--eye", "8", "8", "8", "--lookat", "0", "0", "0", "--threads", "3"))


def test_config1_sources_do_not_touch_the_oracle():
    for f in ("app/restir_main.cpp", "cedec_2024_rt_amd/csrc/host_path.h"):
        txt = open(os.path.join(ROOT, f)).read()
        includes = [l for l in txt.splitlines() if l.lstrip().startswith("#include")]
        assert not any("oracle" in l for l in includes), includes
        assert "restir_oracle" not in txt and "liboracle" not in txt and "o_ao_04" not in txt
