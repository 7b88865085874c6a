"""The line bench.py prints must stay readable by the driver: BENCH_r05.json came back `parsed: null` because the one JSON line had grown to 22.7 KB
and the driver keeps a bounded tail of stdout (8 081 characters).  compact_line() is a pure function of the full record, so it is tested here on the
stored round-5 record (profiles/r05_bench_all.json = what the line used to carry) and on an inflated 8-rank record.
Reference style: the compact per-layer table of demo/common/inferenceProcessor.cpp:143-199."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

import bench  # noqa: E402

DRIVER_TAIL = 8081


def _detail():
    return json.load(open(os.path.join(ROOT, "profiles", "r05_bench_all.json")))


def test_line_from_the_stored_round5_record_is_small_and_round_trips():
    d = _detail()
    assert len(json.dumps(d)) > 20000  # the record that broke the driver's parser
    text = bench.compact_line(d)
    assert "\n" not in text and len(text) < 8192 and len(text) <= bench.LINE_BUDGET < DRIVER_TAIL
    line = json.loads(text)
    assert json.loads(json.dumps(line)) == line
    # the driver's standard keys, unchanged in meaning
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in line, k
    assert line["metric"] == d["metric"] and line["steps"] == d["steps"] and line["warmup"] == d["warmup"] and line["n_gpus"] == d["n_gpus"]
    assert abs(line["value"] - d["value"]) <= 1e-4 * d["value"] and abs(line["ms_per_step"] - d["ms_per_step"]) <= 1e-4 * d["ms_per_step"]
    assert line["config"]["workload"] == d["config"]["workload"] and "model" not in line["config"]
    rf = line["roofline"]
    for k in ("bound", "achieved", "peak", "unit", "frac", "traffic", "kernel", "whole_step_frac"):
        assert k in rf, k
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-3
    assert [c["id"] for c in rf["other_configs"]] == ["c1", "c3", "c4", "c5"]
    cb = line["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] >= 1 and cb["value"] > 0 and cb["unit"] == "images/s" and len(cb["sample"]) <= 200
    assert line["parity"]["ok"] is True
    # what moved to the file
    for k in ("kernels", "layer_table", "wait_semantics", "configs"):
        assert k not in line and k in d
    assert line["detail"] == "bench_detail.json"


def test_line_of_an_eight_rank_record_with_a_census_stays_under_the_cap():
    d = _detail()
    d["n_gpus"] = 8
    d["config"]["rccl_ranks"] = 8
    d["config"]["collective_ranks_seen"] = 8
    d["config"]["backend"] = "nccl"
    d["config"]["ranks"] = [{"rank": r, "device": r, "pci": "0000:%02x:00" % (5 + 16 * r)} for r in range(8)]
    d["ms_per_step_of_each_rank"] = [0.1116 + 1e-4 * r for r in range(8)]
    text = bench.compact_line(d)
    assert len(text) <= bench.LINE_BUDGET
    line = json.loads(text)
    assert line["config"]["rccl_ranks"] == 8 and len(line["config"]["ranks"]) == 8 and len(line["ms_per_step_of_each_rank"]) == 8
    assert "truncated" not in line


def test_an_oversized_record_loses_optional_rows_never_the_headline():
    d = _detail()
    row = copy.deepcopy(d["roofline"]["other_configs"][0])
    d["roofline"]["other_configs"] = [dict(row, id="x%d" % i, dominant_kernel="k" * 120) for i in range(40)]
    text = bench.compact_line(d)
    assert len(text) <= bench.LINE_BUDGET
    line = json.loads(text)
    assert "roofline.other_configs" in line["truncated"] and "other_configs" not in line["roofline"]
    assert line["value"] > 0 and line["roofline"]["frac"] > 0 and line["cpu_baseline"]["value"] > 0


def test_write_detail_keeps_the_full_record(tmp_path):
    d = _detail()
    p = str(tmp_path / "bench_detail.json")
    assert bench.write_detail(d, p) == [p]
    assert json.load(open(p)) == d
