"""bench.py's output contract on the small plumbing configuration (GPU): one JSON line with the fields the driver reads,
the executed-flops roofline, the K1 roofline and the CPU baseline; and the same run through a one-rank RCCL communicator
(file rendezvous + nls_comm_init_rank + barrier / max collectives, exactly what an N-rank launch does per rank)."""
from __future__ import annotations

import json
import os
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
pytestmark = pytest.mark.gpu


def run_bench(extra_env=None, args=()):
    env = dict(os.environ, **(extra_env or {}))
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    r = subprocess.run([sys.executable, str(ROOT / "bench.py"), "--config", "c0", "--steps", "2", "--warmup", "1", *args], env=env,
                       capture_output=True, text=True, timeout=900)  # fmt: skip
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, f"stdout must be exactly one line, got {len(lines)}: {r.stdout[:500]}"
    return json.loads(lines[0])


def test_bench_line_contract():
    d = run_bench()
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype",
              "data", "config", "roofline", "cpu_baseline"):  # fmt: skip
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["dtype"] == "f64" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None and d["higher_is_better"] is True and "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - 1e3 / d["ms_per_step"]) < 1e-6 * d["value"]
    ro = d["roofline"]
    assert ro["bound"] == "mfma" and ro["unit"] == "TFLOP/s" and ro["peak"] == 78.6 and abs(ro["frac"] - ro["achieved"] / ro["peak"]) < 1e-12
    assert 0 < ro["frac"] < 1.0, "frac is the matrix-pipe utilisation: executed flops / time / peak"
    assert 1.0 < ro["algorithmic_gain"] < 1.5
    # what the run cannot measure itself (counter passes under profiles/) is labelled as such, apart from the live fields
    assert "NOT measured by this run" in ro["from_profiles"]["what"] and "mfma_busy" in ro["from_profiles"] and "mfma_busy" not in ro
    k1 = d["roofline_k1"]
    assert k1["bound"] == "hbm" and k1["unit"] == "GB/s" and k1["peak"] == 8000.0 and 0 < k1["frac"] < 1
    dp = k1["datapath"]  # the bound that binds K1: matrix pipe + vector issue on the shared fp64 datapath
    assert dp["valu_instructions_per_feature"] > 30 and abs(dp["bound_ms_per_launch"] - dp["matrix_ms_per_launch"] - dp["valu_ms_per_launch"]) < 1e-9 and 0 < dp["frac"] < 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "fits/s" and cb["cores"] >= 1 and cb["host_cpus"] >= cb["cores"] and "mode" in cb and "sample" in cb
    assert cb["value"] > 0 and "seconds_mode_R" in cb and "seconds_mode_S" in cb  # c0 fits host RAM: both schedules at full size
    assert d["value_pcie_inclusive"] and d["value_pcie_inclusive"] <= d["value"] * 1.2


def test_bench_through_a_one_rank_rccl_communicator(tmp_path):
    d = run_bench({"NLS_BENCH_FORCE_COMM": "1", "NLS_RENDEZVOUS_DIR": str(tmp_path)}, args=("--no-cpu-baseline",))
    assert d["n_gpus"] == 1 and d["cpu_baseline"] is None and d["value"] > 0
    assert d["stage_ms_per_step"]["allreduce"] > 0.0  # the collectives really ran (RCCL on the device buffers)
    assert not list(tmp_path.iterdir()), "rank 0 removes the rendezvous file"


def test_bench_virtual_rank_line(tmp_path):
    """``--as-rank r --of W``: one rank's share of a W-GPU sharded fit on one GPU (own rows, own eigenvector columns, every exchange through a
    one-rank communicator on the real librccl); the peers' blocks are replayed from a captured complete fit, so the results must be that fit's."""
    d = run_bench({"NLS_RENDEZVOUS_DIR": str(tmp_path)}, args=("--no-cpu-baseline", "--no-end-to-end", "--as-rank", "1", "--of", "4"))
    v = d["virtual_rank"]
    assert (v["rank"], v["of"], v["global_n"]) == (1, 4, 20_000) and d["n_gpus"] == 1 and d["config"]["rows_per_gpu"] == 5_000
    assert "VIRTUAL rank 1 of 4" in d["config"]["parallelism"] and "not run" in d["config"]["parallelism"]
    eq = v["equals_complete_fit"]
    assert eq["argmin_equal"] and eq["beta_max_rel_diff"] < 1e-12 and eq["loo_errors_max_rel_diff"] < 1e-12
    assert d["stage_ms_per_step"]["allreduce"] > 0.0 and d["cpu_baseline"] is None
    d0 = run_bench({"NLS_RENDEZVOUS_DIR": str(tmp_path)}, args=("--no-cpu-baseline", "--no-end-to-end", "--as-rank", "0", "--of", "4"))
    assert "run (rank 0)" in d0["config"]["parallelism"] and d0["virtual_rank"]["equals_complete_fit"]["beta_max_rel_diff"] < 1e-12
