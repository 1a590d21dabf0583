"""bench.py contract on the GPU box: the JSON line of a small single-rank run, and the N>1 code path
(rank sharding, the detections all-gather every step, barrier + max-over-ranks timing, rank-0-only
reporting) rehearsed with two ranks that share the one GPU over gloo (MASKLAB_BENCH_REHEARSAL=1; the
driver's real multi-GPU runs use RCCL and one GPU per rank).  -m gpu."""
import json
import os
import subprocess
import sys

import pytest

torch = pytest.importorskip("torch")
pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module", autouse=True)
def _need_gpu():
    if not torch.cuda.is_available():
        pytest.skip("no GPU")


def _json_line(text):
    lines = [l for l in text.splitlines() if l.startswith("{")]
    assert len(lines) == 1, text[-2000:]
    return json.loads(lines[0])


def test_bench_json_contract_single_rank():
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    # (--quick-cpu-baseline: one timed oracle forward per way instead of the repeated / per-thread-count legs, which at
    # this size only cost minutes; the driver's default run takes the full legs)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--workload", "resnext50_full_b2_256",
                        "--steps", "2", "--warmup", "1", "--quick-cpu-baseline"], capture_output=True, text=True, env=env,
                       timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _json_line(r.stdout)
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
                "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline", "parity"):
        assert key in d, key
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["value"] > 0
    assert d["scaling"] == "weak" and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    assert d["config"]["workload"] == "resnext50_full_b2_256"
    for key in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert key in d["roofline"], key
    assert 0 < d["roofline"]["frac"] < 1
    for key in ("value", "unit", "cores", "kind", "sample"):
        assert key in d["cpu_baseline"], key
    assert d["cpu_baseline"]["kind"] in ("port", "port (torch-CPU convs)")        # the oracle, NumPy-only or with oneDNN convs
    assert any("torch-CPU convs" in k for k in d["cpu_baseline"]["images_per_sec"])   # both ways are timed and reported
    assert d["parity"]["timed_forward"]["loc_pred_and_seg_pred_bit_identical_to_the_checked_forward"] is True
    assert d["library"]["path"].endswith("masklab_hip/libmasklab_hip.so") and d["library"]["abi_version"] == 7
    for key in ("cpu_model", "blas_threads", "images_per_sec"):
        assert key in d["cpu_baseline"], key
    assert any("torch" not in k for k in d["cpu_baseline"]["images_per_sec"])     # the NumPy-only way
    assert d["parity"]["ok"] is True and d["parity"]["detection_fmeasure"] > 0.999
    assert d["parity"]["rows_exact"] is True and d["parity"]["order_exact"] is True
    assert all("hbm_frac" in v or "mfma_frac" in v for k, v in d["kernels"].items()
               if k.startswith(("conv_mfma", "groupnorm", "gconv3x3")))
    # the extra leg under the split-operand conv math: beside `value`, same protocol, its own parity block
    x3 = d["f32x3"]
    assert x3["value"] > 0 and x3["steps"] == 2 and x3["warmup"] == 1 and x3["unit"] == "images/sec"
    assert x3["parity"]["ok"] is True and x3["parity"]["order_exact"] is True and x3["parity"]["rows_exact"] is True
    assert d["dtype"] == "f32" and "f16 MFMA" in x3["dtype"]


def test_bench_refuses_more_gpus_than_visible():
    """`python bench.py --gpus 8` started bare on a box with fewer GPUs must fail, not measure one GPU and say so
    in small print.  The check runs before the process touches the device."""
    import torch
    env = dict(os.environ)
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASKLAB_BENCH_REHEARSAL"):
        env.pop(k, None)
    want = torch.cuda.device_count() + 1
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(want), "--steps", "1",
                        "--warmup", "0"], capture_output=True, text=True, env=env, timeout=300)
    assert r.returncode != 0
    assert "visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]
    # a launcher-provided WORLD_SIZE that disagrees with --gpus is refused too
    env2 = dict(env, WORLD_SIZE="2", RANK="0", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4", "--steps", "1"],
                       capture_output=True, text=True, env=env2, timeout=300)
    assert r.returncode != 0 and "WORLD_SIZE" in r.stderr


def test_bench_bare_gpus2_spawns_its_own_ranks():
    """bare `--gpus 2` launches two ranks itself (rehearsal mode: both on cuda:0 over gloo) and relays ONE JSON line."""
    env = dict(os.environ, MASKLAB_BENCH_REHEARSAL="1")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--workload",
                        "resnext50_full_b2_256", "--steps", "2", "--warmup", "1"],
                       capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _json_line(r.stdout)
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4
    assert "merged batch 4" in d["config"]["collective"]


def test_bench_two_ranks_rehearsal():
    env = dict(os.environ, MASKLAB_BENCH_REHEARSAL="1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", "29533", os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--workload", "resnext50_full_b2_256", "--steps", "2", "--warmup", "1"]
    r = subprocess.run(cmd, capture_output=True, text=True, env=env, timeout=900)
    assert r.returncode == 0, (r.stdout[-1500:], r.stderr[-3000:])
    d = _json_line(r.stdout)                       # exactly one line: rank 0 reports
    assert d["n_gpus"] == 2 and d["config"]["global_batch"] == 4 and d["config"]["parallelism"] == "dp2"
    assert d["value"] > 0 and d["cpu_baseline"] is None
    assert "merged batch 4" in d["config"]["collective"]            # the gather really merged both ranks' images


def test_rccl_single_rank_async_gather():
    """RCCL itself (torch backend 'nccl'), in the one form a one-GPU box allows -- a one-rank communicator: the
    detection all-gather launched asynchronously on AsyncDetectionGather's stream, three collectives in flight, data
    back bit-identical (scripts/rccl_single_rank_check.py).  The two-rank paths are covered by gloo tests
    (tests/test_parallel_cpu.py, test_bench_two_ranks_rehearsal); the driver's 8-GPU run is the first real xGMI exchange."""
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MASTER_PORT="29631", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "scripts", "rccl_single_rank_check.py")],
                       capture_output=True, text=True, env=env, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "rccl single-rank gather: OK" in r.stdout
