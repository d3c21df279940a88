"""bench.py's output contract, checked on CPU against the committed lines of the final binary
(profiles/r05_bench_cfg*.json and earlier rounds') and on the pieces of bench.py that run without a GPU: the JSON keys the driver
parses, the roofline / cpu_baseline objects, the per-configuration PMC traffic lookup (never a number measured
on another configuration), BASELINE.json's metric name."""
import importlib.util
import json
import os

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PROF = os.path.join(ROOT, "profiles")


def _bench():
    spec = importlib.util.spec_from_file_location("bench_mod", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)          # main() is guarded by __name__
    return m


def _line(name):
    with open(os.path.join(PROF, name)) as f:
        return json.loads(f.read().strip().splitlines()[-1])


@pytest.mark.parametrize("name", ["r06_bench_cfg3.json", "r06_bench_cfg2.json", "r06_bench_cfg4.json", "r06_bench_cfg5.json",
                                  "r05_bench_cfg3.json", "r05_bench_cfg2.json", "r05_bench_cfg4.json", "r05_bench_cfg5.json",
                                  "r04h_bench_cfg3.json", "r04h_bench_cfg2.json", "r04h_bench_cfg4.json",
                                  "r04h_bench_cfg5.json", "r03f_bench_cfg3.json"])
def test_committed_bench_lines_carry_the_contract(name):
    d = _line(name)
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    # (the committed lines predate the verbatim metric string: they carry it without the " at 1/2/4/8 MI355X" tail)
    assert base["metric"].startswith(d["metric"]) and d["unit"] == "samples/s"
    assert _bench().METRIC == base["metric"]
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["data"] == "synthetic"
    assert d["vs_baseline"] is None                       # BASELINE.md holds no published number for this metric
    assert "workload" in d["config"] and "model" not in d["config"]
    assert abs(d["value"] - d["n_gpus"] * d["config"]["per_gpu_batch"] / (d["ms_per_step"] * 1e-3)) < 0.02 * d["value"]
    r = d["roofline"]
    assert r["bound"] == "mfma" and r["unit"] == "TFLOP/s" and r["peak"] == 2500.0
    assert 0.2 < r["frac"] < 1.0
    if "fp8_gemms" not in r:                              # one peak: frac = achieved / peak
        assert abs(r["frac"] - r["achieved"] / r["peak"]) < 2e-3
    assert r["launches_per_step"] > 500 and r["gemm_ms_per_step"] < d["ms_per_step"]
    c = d["cpu_baseline"]
    assert c["kind"] == "port" and c["cores"] >= 1 and c["unit"] == "samples/s" and c["value"] > 0 and c["sample"]


def test_traffic_is_looked_up_per_configuration_and_null_without_a_profile():
    b = _bench()
    t3, src3 = b.pmc_gemm_traffic(3)
    t2, src2 = b.pmc_gemm_traffic(2)
    t4, src4 = b.pmc_gemm_traffic(4)
    t5, src5 = b.pmc_gemm_traffic(5)
    # (newest committed profile of each configuration: round 6 for all four)
    assert src3 == "r06_step_traffic_pmc.csv" and src2 == "r06_step_traffic_pmc_cfg2.csv"
    assert src4 == "r06_step_traffic_pmc_cfg4.csv" and src5 == "r06_step_traffic_pmc_cfg5.csv"
    assert len({t3, t2, t4, t5}) == 4 and all(3e8 < t < 2e9 for t in (t3, t2, t4, t5))     # bytes per launch
    assert b.pmc_gemm_traffic(1) == (None, None)          # no PMC profile of cfg 1 (the CPU plumbing case) exists
    # (a committed line carries the figure of the newest profile that existed when it was printed)
    assert _line("r04h_bench_cfg3.json")["roofline"]["traffic"] in (t3, 738185118, 738929220)
    assert _line("r03f_bench_cfg5.json")["roofline"]["traffic"] is None      # round 3: the 13B PMC run aborted


def test_configs_match_baseline_json():
    b = _bench()
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))
    assert sorted(b.CONFIGS) == [2, 3, 4, 5]
    assert len(base["configs"]) >= 5
    assert b.CONFIGS[3]["batch"] == 32 and b.CONFIGS[3]["seq"] == 144
    assert b.CONFIGS[4]["seq"] == 2048 and b.CONFIGS[5]["model"] == "real_13b" and b.CONFIGS[5]["fp8"]


def test_bare_gpus_n_forms_the_torchrun_command_of_train_sh():
    """`python bench.py --gpus 8 --steps K --warmup W` without a launcher re-executes itself as
    `python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port P bench.py ...`
    (the driver's own N > 1 form; /root/reference/train.sh:13), arguments forwarded verbatim."""
    import sys
    b = _bench()
    cmd = b.self_launch_command(8, ["--gpus", "8", "--steps", "20", "--warmup", "3"], port=29400)
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and "--nproc-per-node=8" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29400"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "8", "--steps", "20", "--warmup", "3"]
    p1, p2 = b._free_port(), b._free_port()
    assert 1024 < p1 < 65536 and 1024 < p2 < 65536
    # the prediction the first measured curve is read against travels in the line (comm.predicted_ms)
    assert sorted(b.PREDICTED_CFG3) == [1, 2, 4, 8]
    assert b.PREDICTED_CFG3[2]["speedup"][1] < 1.5 and "one xGMI link" in b.PREDICTED_CFG3[2]["note"]
    assert all(len(v["step_ms"]) == 2 for v in b.PREDICTED_CFG3.values())


def test_bare_gpus_n_without_a_gpu_fails_in_the_ranks_not_in_the_launcher(tmp_path):
    """here (no GPU): the bare N = 2 call must really spawn the ranks -- each exits with the 'needs an MI355X' message
    -- and hand their non-zero exit code back"""
    import subprocess
    import sys
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present: covered by tests/test_train_gpu.py")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"],
                       env=env, capture_output=True, text=True, timeout=240, cwd=str(tmp_path))
    assert r.returncode != 0
    assert "needs an MI355X" in r.stderr and "torch.distributed" in r.stderr       # the message came from a spawned rank
