"""The CPU checks whose result depends on the machine they run on, once more under the `gpu` marker.

The driver runs `-m gpu` on the MI355X box and `-m "not gpu"` in the build container; a test carries one marker or
the other. Three host-side links of the parity chain must also hold on the GPU box's own EPYC cores and libm (VERDICT
r02 item 2c): BASELINE config #1 (04_ao, 256x256, a host loop by definition), the tinyobj parity of both OBJ readers,
and the portable-math-vs-libm gate on two full 1920x1080 frames of the benchmark workload. They need no GPU; the
functions are the CPU suite's own, called here unchanged.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def test_config1_04_ao_256x256_on_this_host(oracle, golden_dir):
    import os

    from tests import test_oracle_golden as t

    oracle.set_math_mode(oracle.MATH_LIBM)
    try:
        t.test_config1_04_ao_at_256x256(oracle, golden_dir, np.load(os.path.join(golden_dir, "scenes.npz")))
    finally:
        oracle.set_math_mode(oracle.MATH_PORTABLE)


def test_config1_04_ao_product_host_loop_on_this_host(tmp_path, golden_dir):
    """the PRODUCT's config #1 (restir_app --example 4, host_path.h) on the GPU box's cores and libm"""
    from tests import test_app_config1 as t

    t.test_config1_04_ao_product_host_loop(tmp_path, golden_dir)


def test_obj_readers_match_tinyobj_fixture_on_this_host(tmp_path):
    from tests import test_scenes as t

    t.test_obj_readers_match_tinyobj_fixture(tmp_path)


def test_portable_vs_libm_gate_1080p_on_this_host(oracle):
    from tests import test_portable_math as t

    t.test_portable_vs_libm_frame_within_the_north_star_tolerance(oracle)
