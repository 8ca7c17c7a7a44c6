"""`python bench.py --gpus N` as README.md documents it: no launcher, no RANK / WORLD_SIZE.

The reference has one device only (examples/10_restir_di/10_restir_di.cpp:35-46: `deviceIdx = 0`), so everything about
N > 1 is this build's own contract: bench.py starts N FRESH child processes itself before it makes any GPU call, relays
rank 0's ONE JSON line, and exits non-zero with a `"value": null` line if a child fails.

* CPU (`-m "not gpu"`): on a box without a GPU both children fail at start-up: exactly one JSON line, value null, exit code 1.
* GPU: two ranks on this box's one GPU with the exact host-staged SHM transport (BENCH_DEV_SHM=1): one JSON line whose
  assembled two-strip frame equals a single context's, bit for bit.
* GPU, auto-enabled when the node shows >= 2 GPUs (skips on the one-GPU boxes of this pool): the same command with NO
  development switch = real RCCL send/recv between two GPUs, and `restir_app --ranks 2` (the C++ host over rt_mg_*).
"""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BENCH = os.path.join(ROOT, "bench.py")


def _run_bench(extra_env, *argv, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(extra_env)
    p = subprocess.run([sys.executable, BENCH] + list(argv), env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=timeout, cwd=ROOT)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    return p.returncode, lines, p.stderr.decode()


def _gpu_count():
    import torch

    return torch.cuda.device_count()  # counting devices does not initialise the GPU on this image


def test_plain_launch_without_a_gpu_prints_one_null_line():
    """no GPU here: both children die at start-up; the parent still prints exactly one JSON line and says why"""
    if _gpu_count() > 0:
        pytest.skip("this box has a GPU: the failing-child path is covered by the watchdog self-test below")
    rc, lines, err = _run_bench({"BENCH_CHILD_GRACE_S": "20"}, "--gpus", "2", "--steps", "1", "--warmup", "0", timeout=300)
    assert rc == 1, err
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["value"] is None and d["n_gpus"] == 2 and d["steps"] == 1 and d["warmup"] == 0
    assert d["child_return_codes"] == [2, 2] and "GPU" in d["error"], d
    assert "fresh child processes" in d["launched_by"]


def test_world_size_mismatch_is_one_null_line():
    """`--gpus 4` inside a 1-rank environment a launcher made: a message and a null line, not a hang or a traceback"""
    if _gpu_count() > 0:
        pytest.skip("CPU-box check")
    env = {k: v for k, v in os.environ.items()}
    env.update(RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, BENCH, "--gpus", "4", "--steps", "1", "--warmup", "0"], env=env, stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE, timeout=300, cwd=ROOT)
    lines = [ln for ln in p.stdout.decode().splitlines() if ln.strip()]
    assert p.returncode == 2 and len(lines) == 1, (p.returncode, lines, p.stderr.decode())
    d = json.loads(lines[0])
    assert d["value"] is None and "WORLD_SIZE=1" in d["error"]


@pytest.mark.gpu
def test_plain_launch_two_ranks_shm_is_verified():
    """the documented command, two ranks on ONE GPU with the exact SHM transport: one line, two-strip frame == single context"""
    rc, lines, err = _run_bench({"BENCH_DEV_SHM": "1", "BENCH_NO_4K": "1"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["verified_vs_single_context"] is True, d
    assert len(d["config"]["strips"]) == 2 and "dev_shm" in d
    one = d["single_gpu_on_this_node"]  # the like-with-like reference the line carries itself
    assert one["ms_per_frame_unpipelined"] > 0 and one["ms_per_frame_pipelined"] > 0 and one["speedup_vs_pipelined"] > 0
    assert "fresh child processes" in d["launched_by"]


@pytest.mark.gpu
def test_plain_launch_eight_ranks_shm_is_verified():
    """VERDICT r05 item 8, first-contact insurance for the driver's N = 8 run: the documented command with EIGHT ranks — eight fresh
    children, the file-store rendezvous, the id broadcast, eight strip contexts with two neighbours each (six of them), the
    gather of the verification — end to end on this box's one GPU over the exact SHM transport, at a reduced size (640 x 720:
    eight 90-row strips, the thinnest the 87-row halo allows). The assembled eight-strip frame == a single context's, bit for bit."""
    rc, lines, err = _run_bench({"BENCH_DEV_SHM": "1", "BENCH_NO_4K": "1", "BENCH_LAUNCH_TIMEOUT_S": "800"},
                                "--gpus", "8", "--steps", "2", "--warmup", "1", "--width", "640", "--height", "720", timeout=900)
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 8 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0 and d["scaling"] == "strong"
    assert d["verified_vs_single_context"] is True, d
    assert [b - a for a, b in d["config"]["strips"]] == [90] * 8
    assert len(d["strip_driver"]) == 8 and sorted(x["rank"] for x in d["strip_driver"]) == list(range(8))
    assert "fresh child processes" in d["launched_by"]


@pytest.mark.gpu
def test_plain_launch_reports_a_stalled_rank():
    """rank 1 stops in the second timed frame: its watchdog ends it, rank 0's ends rank 0 with the diagnostic line, the parent
    relays ONE null line and exits 1"""
    rc, lines, err = _run_bench({"BENCH_DEV_MIRROR": "1", "BENCH_NO_4K": "1", "BENCH_TEST_STALL": "1:3", "BENCH_WATCHDOG_S": "15",
                                 "BENCH_CHILD_GRACE_S": "60", "BENCH_VERIFY": "0"}, "--gpus", "2", "--steps", "3", "--warmup", "1")
    assert rc == 1, err[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["value"] is None and "error" in d and d["n_gpus"] == 2, d


@pytest.mark.gpu
def test_plain_launch_two_gpus_real_rccl():
    """auto-enabled on a node with >= 2 GPUs: `python bench.py --gpus 2` with no development switch = the native strip driver
    over real RCCL send/recv; the assembled frame must equal a single context's (BENCH_VERIFY defaults to on there)"""
    if _gpu_count() < 2:
        pytest.skip("one GPU on this box: RCCL refuses two ranks on one device")
    rc, lines, err = _run_bench({"BENCH_NO_4K": "1"}, "--gpus", "2", "--steps", "5", "--warmup", "2")
    assert rc == 0, err[-3000:]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["verified_vs_single_context"] is True, d
    assert "RCCL send/recv (native driver)" in d["config"]["parallelism"] and "FALLBACK" not in d["config"]["parallelism"], d


@pytest.mark.gpu
def test_host_app_two_gpus_real_rccl(tmp_path):
    """auto-enabled on a node with >= 2 GPUs: `restir_app --ranks 2` = rt_mg_* with the RCCL transport between two devices (the
    C++ host, no Python in the ranks), byte-identical to the single-process image — the test of tests/test_mg_native.py's
    --shm run with the real transport"""
    if _gpu_count() < 2:
        pytest.skip("one GPU on this box: RCCL refuses two ranks on one device")
    from cedec_2024_rt_amd import scenes

    app = os.path.join(ROOT, "app", "restir_app")
    assert os.path.exists(app), "app/restir_app not built: run __graft_entry__.build()"
    tris = scenes.make_quad_room()
    tpath = str(tmp_path / "scene.tris")
    tris.tofile(tpath)
    common = ["--tris", tpath, "--size", "160", "330", "--frames", "5", "--eye", "0.5", "2.5", "6.0", "--lookat", "0.0", "1.5", "-1.0"]
    one, two = str(tmp_path / "one.pfm"), str(tmp_path / "two.pfm")
    subprocess.check_call([app] + common + ["--pfm", one], stdout=subprocess.DEVNULL, timeout=120)
    out = subprocess.check_output([app] + common + ["--pfm", two, "--ranks", "2"], timeout=300).decode()
    assert "2 ranks:" in out, out
    a, b = open(one, "rb").read(), open(two, "rb").read()
    assert len(a) == len(b) and a == b
