"""bench.py's driver-facing contract: the LAST stdout line is a compact JSON object (target <= 4 KB, never
above 8 KB — round 3's 23 KB line left BENCH_r03.json unparsed), every detail goes to bench_full.json / an earlier line, and
`--gpus N` means N ranks (self-launch under torch.distributed.run; loud failure when the devices are not there)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

REQUIRED = ["metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
            "dtype", "data", "config", "roofline", "cpu_baseline"]


def _full_record():
    """a real full record (round 3's 23 KB line) — the worst case seen so far"""
    return json.load(open(os.path.join(ROOT, "profiles", "r03_bench_10m_full.json")))


def test_final_line_of_a_real_record_is_small_and_complete():
    res = _full_record()
    line = bench.final_line(res)
    assert len(line) <= bench.LINE_TARGET, len(line)
    assert "\n" not in line
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert c["value"] == pytest.approx(res["value"], rel=1e-5)
    assert c["config"]["workload"] and "model" not in c["config"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
        assert k in c["roofline"], k
    assert c["roofline"]["frac"] == pytest.approx(res["roofline"]["achieved"] / res["roofline"]["peak"], rel=1e-4)
    for k in ("value", "unit", "cores", "kind"):
        assert k in c["cpu_baseline"], k
    # the operating point and the FLAT legs are first-class, one short object each
    assert c["op"]["recall_at_10"] >= 0.98 and 0 < c["op"]["frac"] < 1 and c["op"]["ef"] == res["operating_point"]["ef"]
    for leg in ("c1", "c2", "c3"):
        assert set(c[leg]) >= {"value", "ms", "frac"}, leg
    assert "trimmed" not in c


def test_final_line_of_this_rounds_record_carries_every_leg():
    """the round-4 record (all legs incl. the product quantiser and the f8 store): still under the target, nothing trimmed"""
    res = json.load(open(os.path.join(ROOT, "profiles", "r04aa_bench_10m_full.json")))
    line = bench.final_line(res)
    assert len(line) <= bench.LINE_TARGET, len(line)
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert "trimmed" not in c
    for leg in ("op", "c1", "c2", "c3", "c3f8", "pq", "f3", "h1"):
        assert leg in c, leg
    assert c["pq"]["equals_oracle"] is True and 0 < c["pq"]["frac"] < 1
    assert c["roofline"]["frac"] == pytest.approx(res["roofline"]["achieved"] / res["roofline"]["peak"], rel=1e-4)
    # the committed last line of that run is what final_line() produces from the committed full record
    committed = json.load(open(os.path.join(ROOT, "profiles", "r04aa_bench_10m_line.json")))
    now = {kk: v for kk, v in c["roofline"].items() if kk != "traffic_source"}   # (round 5 carries the PMC file's name in the compact line as well)
    assert committed["value"] == c["value"] and committed["roofline"] == now and committed["op"] == c["op"]
    assert c["roofline"]["traffic_source"].startswith("profiles/")


def test_final_line_is_bounded_whatever_the_legs_hold():
    """pathological strings everywhere: the line is trimmed leg by leg, the contract's fields survive"""
    res = _full_record()
    res["config"]["workload"] = "w" * 300
    res["roofline"]["kernel"] = "k" * 400
    res["cpu_baseline"]["sample_short"] = "s" * 500
    res["secondary"]["pq"] = {"value": 1.0, "ms_per_batch_kernels": 2.0, "roofline": {"frac": 0.5}, "equals_oracle": True}
    res["secondary"]["shard"] = {"value": 1.0, "exchange": "rccl", "world": 8, "shard_rows": 5, "n_total": 40}
    res["secondary"]["f3"]["lists"]["every_10th"].update({f"batch_{i}_mfma": {"kernels_ms": 0.123456, "frac_of_hbm_peak": 0.2, "equals_exact_mode": True} for i in range(100, 400)})
    line = bench.final_line(res)
    assert len(line) <= bench.LINE_HARD
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert "f3" in c["trimmed"]


def test_an_error_in_a_leg_stays_short():
    res = _full_record()
    res["operating_point"] = {"error": "x" * 5000}
    res["secondary"]["c3"] = {"error": "y" * 5000}
    c = json.loads(bench.final_line(res))
    assert len(c["op"]["error"]) <= 160 and len(c["c3"]["error"]) <= 160


def test_gpus_n_without_devices_fails_loudly():
    """no launcher around it and no GPUs: `--gpus 2` must not silently run world 1"""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], env=env,
                       capture_output=True, text=True, timeout=300)
    assert p.returncode != 0
    assert "HIP device(s) visible" in (p.stderr + p.stdout)


def test_gpus_n_under_a_launcher_of_another_size_is_refused():
    env = dict(os.environ, WORLD_SIZE="1", RANK="0", LOCAL_RANK="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "4"], env=env, capture_output=True, text=True, timeout=300)
    assert p.returncode != 0 and "WORLD_SIZE=1" in (p.stderr + p.stdout)


@pytest.mark.gpu
def test_gpus_2_launches_two_ranks_on_one_device(gpu):
    """the command shape the driver uses for SCALE (`python bench.py --gpus N`), two ranks sharing cuda:0: n_gpus == 2, the sharded
    layout exchanged through shared memory, answers equal to the same shards held by ONE process."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["COLTT_BENCH_EXCHANGE"] = "shm"
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--share-device", "--backend", "gloo", "--n", "20000",
                        "--dim", "64", "--queries", "256", "--legs", "none", "--no-cpu-baseline", "--steps", "2", "--warmup", "1",
                        "--build-batch", "1024"], env=env, capture_output=True, text=True, timeout=900)
    assert p.returncode == 0, p.stderr[-3000:]
    last = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
    assert len(last) <= bench.LINE_TARGET
    c = json.loads(last)
    assert c["n_gpus"] == 2 and c["config"]["mode"] == "replica"
    assert c["shard"]["exchange"] == "shm" and c["shard"]["world"] == 2
    assert c["shard"]["equals_single_process_group"] is True
    full = json.load(open(os.path.join(ROOT, "bench_full.json")))
    assert full["secondary"]["shard"]["n_total"] == 20000


def test_final_line_of_round_5_carries_the_pq_walk_and_the_traffic_source():
    """the round-5 record (call I): every leg, the product-quantised walk next to the operating point, the PMC file's name, full-size parity of C3 / c3f8"""
    res = json.load(open(os.path.join(ROOT, "profiles", "r05i_bench_10m_full.json")))
    line = bench.final_line(res)
    assert len(line) <= bench.LINE_TARGET, len(line)
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert "trimmed" not in c
    for leg in ("op", "c1", "c2", "c3", "c3f8", "pq", "f3", "h1"):
        assert leg in c, leg
    assert c["roofline"]["traffic_source"].startswith("profiles/") and 0.7 < c["roofline"]["frac"] < 1
    pq = c["op"]["pq"]
    assert pq["recall_at_10"] >= 0.98 and pq["gpu_equals_oracle"] is True and pq["over_plain_walk"] > 1.0 and pq["value"] > c["op"]["value"]
    assert c["op"]["recall_at_10"] >= 0.98 and c["op"]["gpu_equals_oracle"] is True
    for leg in ("c3", "c3f8"):     # the parity sample of these legs is full size now (cpu_baseline.full_size_oracle_sample)
        assert c[leg]["gpu_equals_oracle"] is True and res["secondary"][leg]["cpu_baseline"]["full_size_oracle_sample"]["rows"] == 10_000_000
    committed = json.load(open(os.path.join(ROOT, "profiles", "r05i_bench_10m_line.json")))
    assert committed["value"] == c["value"] and committed["op"]["pq"]["value"] == pq["value"]


def test_last_record_of_round_5_is_the_last_library():
    """the round's last full run of the driver's command (call V, the shipped library): every leg incl. the 8-member pipeline; the product-quantised
    walk answers the recall >= 0.98 point at more than twice the plain walk's rate; the line is what final_line() makes of the full record"""
    res = json.load(open(os.path.join(ROOT, "profiles", "r05v_bench_10m_full.json")))
    line = bench.final_line(res)
    assert len(line) <= bench.LINE_TARGET, len(line)
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert "trimmed" not in c
    for leg in ("op", "c1", "c2", "c3", "c3f8", "pq", "f3", "h1", "g8"):
        assert leg in c, leg
    committed = json.load(open(os.path.join(ROOT, "profiles", "r05v_bench_10m_line.json")))
    assert committed["value"] == c["value"] and committed["roofline"]["frac"] == c["roofline"]["frac"]
    assert c["cpu_baseline"]["gpu_equals_oracle_on_sample"] is True and c["cpu_baseline"]["counters_equal"] is True
    op = c["op"]; pq = op["pq"]
    assert op["recall_at_10"] >= 0.98 and pq["recall_at_10"] >= 0.98 and op["gpu_equals_oracle"] is True and pq["gpu_equals_oracle"] is True
    assert pq["value"] > 2.0 * op["value"] and committed["op"]["pq"]["value"] == pq["value"]
    g8 = c["g8"]
    assert g8["streamed_equals_serial"] is True and g8["exposed_frac_of_a_streamed_batch"] < 0.05


def test_final_line_of_round_6_record_fits_and_carries_the_new_legs():
    """round 6's final record (call AX: op.pq, op.pq_ref, op.diverse — marked as not reference behaviour —, the one-launch PQ scan beside the segment chain): under the
    4 KB target, nothing trimmed, the fields the round's claims rest on present"""
    res = json.load(open(os.path.join(ROOT, "profiles", "r06ax_bench_10m_full.json")))
    line = bench.final_line(res)
    assert len(line) <= bench.LINE_TARGET, len(line)
    c = json.loads(line)
    for k in REQUIRED:
        assert k in c, k
    assert "trimmed" not in c
    op = c["op"]
    assert op["recall_at_10"] >= 0.98 and op["pq"]["recall_at_10"] >= 0.98 and op["pq"]["value"] > 4e5 and op["pq"]["gpu_equals_oracle"] is True
    assert op["pq_ref"]["m"] == 32 and op["pq_ref"]["centroids"] == 256
    assert op["diverse"]["reference_behaviour"] is False and op["diverse"]["recall_at_10"] >= 0.98 and op["diverse"]["ef"] < op["ef"]
    assert c["pq"]["eq_chain"] is True and c["pq"]["ms"] < c["pq"]["segment_chain_ms"] and 0 < c["pq"]["search_frac"] < c["pq"]["frac"] < 1
    assert c["roofline"]["frac"] == pytest.approx(res["roofline"]["achieved"] / res["roofline"]["peak"], rel=1e-4)
