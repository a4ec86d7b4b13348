"""bench.py's output contract on a small graph: ONE JSON line on stdout with the fields the driver reads, the `roofline`
and `cpu_baseline` objects, for one GPU and for the multi-rank launch (two ranks sharing the GPU over gloo)."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
        "dtype", "data", "config", "roofline")


def _one_line(out):
    lines = [l for l in out.splitlines() if l.strip()]
    assert len(lines) == 1, "stdout must hold ONE line, got %d:\n%s" % (len(lines), out[-2000:])
    return json.loads(lines[0])


def _env():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONPATH=ROOT)
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    return env


@pytest.mark.gpu
def test_single_gpu_line():
    r = subprocess.run([sys.executable, "bench.py", "--workload", "medium", "--steps", "6", "--warmup", "2", "--cpu-seconds", "1",
                        "--hbm-leg", "off", "--epoch-leg", "off"], cwd=ROOT, env=_env(), capture_output=True, text=True,
                       timeout=600)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    for k in KEYS + ("cpu_baseline",):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2 and d["dtype"] == "f32" and d["higher_is_better"] is True
    assert d["unit"] == "triples/s" and d["value"] > 0 and d["vs_baseline"] is None and "workload" in d["config"]
    assert abs(d["value"] - d["config"]["batch"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    rf = d["roofline"]
    # (the "medium" graph's panel is cache resident: the bound is the cache hierarchy's gather rate, and the line says so)
    assert rf["bound"] == "l2+infinity-cache gather" and rf["cache_resident"] is True
    assert rf["unit"] == "GB/s" and rf["peak"] == 8000.0
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and rf["us_per_launch"] > 0
    assert rf["model_estimate"] > 0 and abs(rf["frac_of_model_estimate"] - rf["achieved"] / rf["model_estimate"]) < 1e-9
    cb = d["cpu_baseline"]
    assert cb["kind"] in ("port", "reference") and cb["value"] > 0 and cb["cores"] >= 1 and cb["sample"]


@pytest.mark.gpu
def test_headline_carries_the_other_configs_steady_state_and_in_run_traffic():
    """The default headline's extra fields on a cut-down run (no synth-10M legs): `configs` = one measured leg for each of
    BASELINE configs[0], [2], [3] with its own roofline object and CPU leg; `steady_state` (50-step windows); the step as
    ONE library call with the host's busy time; and `roofline.traffic` MEASURED IN THIS RUN by rocprofv3 children (when the
    profiler is on PATH), the file-based figure kept beside it."""
    import shutil

    r = subprocess.run([sys.executable, "bench.py", "--steps", "20", "--warmup", "5", "--cpu-seconds", "1", "--hbm-leg", "off",
                        "--scale-point", "off", "--epoch-leg", "off", "--configs-leg", "on", "--pmc", "on", "--steady-steps", "200"],
                       cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and d["steps"] == 20 and "yelp2018" in d["config"]["workload"]
    assert d["step_call"]["library_calls_per_step"] == 1 and 0 < d["host_busy_ms_per_step"] <= d["host_issue_ms_per_step"] + 1e-9
    st = d["steady_state"]
    assert st["windows"] == 4 and 0 < st["ms_per_step_min"] <= st["ms_per_step_median"] <= st["ms_per_step_max"]
    cfg = {c["baseline_config"]: c for c in d["configs"]}
    assert sorted(cfg) == [0, 2, 3], d["configs"]
    for i, model in ((0, "MFBPR"), (2, "LightGCN"), (3, "SimGCL")):
        c = cfg[i]
        assert "error" not in c, c
        assert model in c["metric"] and c["ms_per_step"] > 0 and c["dominant_kernel_us"] > 0 and c["cpu_baseline"]["value"] > 0
        assert abs(c["value"] - (2048 if i != 2 else 1024) * 1e3 / c["ms_per_step"]) <= 1e-6 * c["value"]
        assert c["roofline"]["frac"] > 0 and c["loss_first_last"][1] < c["loss_first_last"][0]
    assert "amazon-book" in cfg[2]["workload"] and "amazon-book" in cfg[3]["workload"] and "yelp2018" in cfg[0]["workload"]
    if shutil.which("rocprofv3"):
        t = d["roofline"]["traffic_in_run"]
        assert "yelp2018" in t["graphs"] and "amazon-book" in t["graphs"], t
        assert d["roofline"]["traffic_source"].startswith("measured in this run")
        # bytes at the L2s' memory side of a cache-resident gather: between the compulsory bytes and the gather bytes
        assert d["roofline"]["bytes_min"] * 0.5 < d["roofline"]["traffic"] < d["roofline"]["bytes_gather"]
        assert cfg[2]["roofline"]["traffic_source"].startswith("measured in this run")


@pytest.mark.gpu
def test_two_rank_line():
    """The launch the driver uses for N > 1 (torch.distributed.run, one process per rank), rehearsed with two ranks on
    one GPU over gloo: the user-row-sharded step, exchanged row counts reported, item table coherent across ranks."""
    r = subprocess.run([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                        "127.0.0.1", "--master-port", "29633", "bench.py", "--gpus", "2", "--backend", "gloo", "--parallel", "shard",
                        "--workload", "medium", "--steps", "5", "--warmup", "2", "--no-cpu-baseline"], cwd=ROOT, env=_env(),
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 2 and d["steps"] == 5 and d["scaling"] == "strong" and d["item_table_coherent"] is True
    assert abs(d["value"] - d["config"]["batch"] * 1e3 / d["ms_per_step"]) <= 1e-6 * d["value"]
    assert d["roofline"]["exchange_rows"]["items"] == 3000 and d["config"]["parallelism"] == "user-row shard x2"
    # O(I + B d) — the flag vector, the guest rows, the batch's item rows: nothing of size [I, d] is replicated work
    assert 0 < d["replicated_bytes_per_step_per_rank"] <= 8 * 3000 + 4 * 64 * 16 * 1024
    # the same workload on ONE GPU, measured in the same run (rank 0, after the timed region)
    assert d["single_gpu_reference"]["ms_per_step"] > 0
    assert abs(d["speedup_vs_1gpu"] - d["single_gpu_reference"]["ms_per_step"] / d["ms_per_step"]) < 1e-9
    _assert_parity(d)


def _assert_parity(d, steps=3):
    """the N-rank line proves its own correctness — the first steps, from the fresh tables, repeated on one
    device in the same run: loss, FIN at the batch rows and sampled table rows within 1e-4."""
    par = d["parity_vs_1gpu"]
    assert par["ok"] is True and par["tol"] == 1e-4 and par["steps"] == steps and par["initial_tables_equal"] is True, par
    for key in ("loss_rel_err", "final_rows_rel_err", "table_rel_err"):
        assert 0 <= par[key] <= 1e-4, (key, par[key])
    assert par["update_rel_err"] <= 1e-2, par  # (the tables as UPDATES since the initial ones: the steps did move them alike)
    assert len(par["loss_n_ranks"]) == steps and "error" not in d


TIMELINE_TAGS = {"F1.panel", "flags", "F2.touched", "F3.items", "guest_rows", "B1.touched", "B2.panel", "B3.reduce_scatter",
                 "item_table.all_gather"}


@pytest.mark.gpu
def test_eight_rank_line():
    """The HEADLINE multi-GPU launch — eight ranks, the item panel in eight slices, as `bench.py --gpus 8` runs on a real
    node — rehearsed with the eight ranks SHARING the one GPU over gloo (host-staged collectives: the structure, not the
    numbers): one JSON line with n_gpus 8, every collective of the step under its timeline tag, the launch order of the
    instrumented steps as DESIGN.md §7's overlap model assumes on every rank, the item table coherent — inside the
    watchdog's deadline (the 8-rank configuration had never executed a step in any form)."""
    env = dict(_env(), IDG_BENCH_TIMEOUT="900")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--backend", "gloo", "--parallel", "shard", "--workload",
                        "synth-1M", "--dim", "64", "--item-slices", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--scale-point", "on"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    for k in KEYS:
        assert k in d, k
    assert d["n_gpus"] == 8 and d["steps"] == 3 and d["scaling"] == "strong" and d["item_table_coherent"] is True
    assert "retried" not in d and d["config"]["parallelism"] == "user-row shard x8" and d["config"]["item_panel_slices"] == 8
    t = d["timeline"]
    assert t["instrumented_steps"] == 3 and TIMELINE_TAGS <= set(t["collectives"]), sorted(t["collectives"])
    assert t["host_sync_ms"] == 0.0
    assert t["collectives"]["F1.panel"]["calls_per_step"] == 8 and t["collectives"]["item_table.all_gather"]["calls_per_step"] == 8
    o = t["issue_order"]
    assert o["steps"] == 3 and o["slices"] == 8 and o["violations"] == [] and o["ranks_with_violations"] == 0
    # every collective's achieved bus rate against what one GPU can send over its seven xGMI links
    assert t["xgmi_peak_gbs"] == 7 * 153.0
    for tag in ("F1.panel", "B2.panel", "B3.reduce_scatter", "item_table.all_gather"):
        c = t["collectives"][tag]
        assert c["bus_gbs"] > 0 and abs(c["bus_frac_of_xgmi_peak"] - c["bus_gbs"] / (7 * 153.0)) < 1e-12, c
    # ... and the line's own correctness evidence: the eight ranks' first steps against ONE device, in this run
    _assert_parity(d)
    assert d["single_gpu_reference"]["ms_per_step"] > 0 and d["speedup_vs_1gpu"] > 0


@pytest.mark.gpu
def test_eight_rank_line_with_the_24_bit_exchange():
    """--panel-bits 24 (opt-in): the eight-rank rehearsal with the panel exchanges as 24-bit rows summed in rank
    order — 3/4 of the fp32 exchange's bytes on the wire in the line, the halves of every exchange issued in the pipelined
    order, and the run's own parity check against the single-GPU result inside 1e-4."""
    env = dict(_env(), IDG_BENCH_TIMEOUT="900")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "8", "--backend", "gloo", "--parallel", "shard", "--workload",
                        "synth-1M", "--dim", "64", "--item-slices", "8", "--steps", "3", "--warmup", "1", "--no-cpu-baseline",
                        "--scale-point", "on", "--panel-bits", "24"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 8 and d["item_table_coherent"] is True and "24-bit panel exchange" in d["config"]["comm"]
    px = d["panel_exchange"]
    assert px["bits"] == 24 and abs(px["ratio"] - 0.75) < 1e-9 and px["order_violations"] == [] and px["exchanges_per_step"] >= 4 * 8
    assert px["packed_bytes_per_step_per_rank"] > 0.9 * 0.75 * px["fp32_bytes_they_replace"]
    assert px["bytes_on_the_wire_per_step_per_rank"] < 0.8 * d["roofline"]["exchange_bytes_per_step_per_rank"]
    assert "rank order" in px["reduction_order"]
    t = d["timeline"]
    assert t["issue_order"]["violations"] == [] and t["issue_order"]["ranks_with_violations"] == 0
    par = d["parity_vs_1gpu"]
    assert par["ok"] is True and 0 < par["table_rel_err"] <= 1e-4 and 0 < par["final_rows_rel_err"] <= 1e-4 and par["loss_rel_err"] <= 1e-4, par


@pytest.mark.gpu
def test_two_rank_line_with_the_rank_ordered_fp32_exchange():
    """--reduce-order rank: the panel reductions as an explicit fp32 exchange summed in rank order (no packing): the bytes of
    the RCCL form on the wire, the parity check at fp32 rounding."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--parallel", "shard", "--workload",
                        "synth-1M", "--dim", "64", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--scale-point", "on",
                        "--reduce-order", "rank"], cwd=ROOT, env=_env(), capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    px = d["panel_exchange"]
    assert px["bits"] == 32 and abs(px["ratio"] - 1.0) < 1e-9 and px["order_violations"] == [] and "idg_reduce_blocks_f32" in px["reduction_order"]
    assert "explicit fp32 panel exchange" in d["config"]["comm"] and d["item_table_coherent"] is True
    _assert_parity(d)


@pytest.mark.gpu
def test_a_wrong_multi_rank_result_fails_the_run():
    """`parity_vs_1gpu.ok` false: the line is printed WITH an "error" field and the launcher leaves with status 3 — no
    retry (a wrong result is not a hang).  IDG_BENCH_TEST_BREAK_PARITY stands in for a broken exchange: rank 1 scales the
    user rows it owns by 1 + 1e-3 after the parity steps' first one."""
    env = dict(_env(), IDG_BENCH_TEST_BREAK_PARITY="1")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--parallel", "shard", "--workload",
                        "medium", "--steps", "3", "--warmup", "1", "--no-cpu-baseline", "--scale-point", "on"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 3, (r.returncode, r.stderr[-3000:])
    d = _one_line(r.stdout)
    par = d["parity_vs_1gpu"]
    assert par["ok"] is False and "parity_vs_1gpu FAILED" in d["error"] and "retried" not in d
    assert max(par["final_rows_rel_err"], par["table_rel_err"]) > 1e-4, par
    assert d["n_gpus"] == 2 and d["value"] > 0  # (the measurement is in the line; the status says it must not be used)


@pytest.mark.gpu
def test_world_one_over_rccl_carries_the_parity_check():
    """The sharded path at world size 1 over backend nccl (RCCL, the library's communicator) with the single-GPU reference
    in the run: the code path of the driver's N > 1 launch on the one device there is."""
    env = dict(_env(), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT="29641")
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "1", "--force-sharded", "--parallel", "shard", "--workload", "medium",
                        "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--scale-point", "on"], cwd=ROOT, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 1 and d["config"]["parallelism"] == "user-row shard x1"
    _assert_parity(d)


@pytest.mark.gpu
def test_gpus_flag_launches_its_own_ranks():
    """`python bench.py --gpus 2` WITHOUT a launcher (the form of the driver's N = 1 command with another N): the ranks are
    started as child processes before anything touches the GPU, and rank 0's line comes back on stdout."""
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--parallel", "shard", "--workload",
                        "medium", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--scale-point", "off"], cwd=ROOT,
                       env=_env(), capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["steps"] == 4 and d["item_table_coherent"] is True and "speedup_vs_1gpu" not in d


def test_hung_ranks_end_in_a_json_error_line():
    """A multi-GPU run whose ranks never come back (IDG_BENCH_TEST_HANG stands in for a rank stuck in ncclCommInitRank or
    a collective — here every rank stops at its first phase, before anything touches a GPU, so this runs anywhere): each
    rank's supervisor ends its worker at the deadline, the run is tried once more with --comm torch in fresh processes,
    and when that hangs too rank 0 prints ONE JSON line with "error", both attempts and the phase every rank had reached,
    and the launcher exits non-zero — within the timeout, instead of leaving the driver to its 1800 s kill."""
    import time

    env = dict(_env(), IDG_BENCH_TEST_HANG="*:start", IDG_BENCH_TIMEOUT="16")
    t0 = time.time()
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--workload", "medium", "--steps", "2",
                        "--warmup", "1", "--no-cpu-baseline"], cwd=ROOT, env=env, capture_output=True, text=True, timeout=300)
    took = time.time() - t0
    assert r.returncode != 0
    d = _one_line(r.stdout)
    assert d["value"] is None and d["n_gpus"] == 2 and "both attempts failed" in d["error"]
    assert [a["attempt"] for a in d["attempts"]] == [1, 2] and [a["comm"] for a in d["attempts"]] == ["auto", "torch"]
    for a in d["attempts"]:
        assert all("deadline" in v or "another rank" in v for v in a["verdicts"]), a
        assert a["phase_reached"] == {"0": "start", "1": "start"}
    assert took < 120, took


@pytest.mark.gpu
def test_a_hung_first_attempt_is_retried_over_torch_distributed():
    """Rank 1 hangs in the warm-up of the FIRST attempt only (two ranks sharing the GPU over gloo): the watchdog ends the
    attempt, the retry in fresh processes with --comm torch measures, and the line says it was a retry and why."""
    env = dict(_env(), IDG_BENCH_TEST_HANG="1:warmup:1", IDG_BENCH_TIMEOUT="200")  # 120 s for the hung attempt, >= 60 s for the retry
    r = subprocess.run([sys.executable, "bench.py", "--gpus", "2", "--backend", "gloo", "--parallel", "shard", "--workload",
                        "medium", "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--scale-point", "off"], cwd=ROOT,
                       env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _one_line(r.stdout)
    assert d["n_gpus"] == 2 and d["value"] > 0 and d["item_table_coherent"] is True
    why = d["retried"]["why"]
    assert why["attempt"] == 1 and why["phase_reached"]["1"] == "warmup" and any("deadline" in v for v in why["verdicts"])
    t = d["timeline"]
    assert t["instrumented_steps"] == 3 and "F1.panel" in t["collectives"] and t["host_sync_ms"] == 0.0


def test_counter_csv_parsing_tells_the_graphs_apart(tmp_path):
    """bench.py --pmc reads rocprofv3's counter_collection.csv: only dispatches of the plain dense product count, and the
    grid size (tiles x 256) says which graph a dispatch belongs to (the same kernel instantiation serves every graph)."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("idg_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    d = tmp_path / "FETCH_SIZE" / "host"
    d.mkdir(parents=True)
    head = '"Correlation_Id","Dispatch_Id","Agent_Id","Queue_Id","Process_Id","Thread_Id","Grid_Size","Kernel_Id","Kernel_Name",' \
           '"Workgroup_Size","LDS_Block_Size","Scratch_Size","VGPR_Count","Accum_VGPR_Count","SGPR_Count","Counter_Name",' \
           '"Counter_Value","Start_Timestamp","End_Timestamp"\n'
    dense = "void (anonymous namespace)::spmm_tile_kernel<16, 1, 8, true, 1, 0, true>((anonymous namespace)::Tile const*, long const*)"
    adam = "void (anonymous namespace)::spmm_tile_kernel<16, 1, 8, true, 1, 2, true>((anonymous namespace)::Tile const*, long const*)"
    wide = "void (anonymous namespace)::spmm_tile_kernel<64, 2, 8, true, 8, 0, false>((anonymous namespace)::Tile const*, long const*)"

    def row(grid, name, ctr, val):
        return '1,1,"Agent 2",1,10,10,%d,8,"%s",256,0,0,64,0,32,"%s",%f,1,2\n' % (grid, name, ctr, val)

    (d / "p_counter_collection.csv").write_text(head + row(5321 * 256, dense, "FETCH_SIZE", 100.0) + row(5321 * 256, dense, "FETCH_SIZE", 102.0)
                                                + row(5321 * 256, adam, "FETCH_SIZE", 999.0)        # another epilogue: not the plain product
                                                + row(10242 * 256, dense, "FETCH_SIZE", 300.0)
                                                + row(836557 * 256, wide, "FETCH_SIZE", 7.0)          # the d = 256 instantiation counts too
                                                + row(777 * 256, dense, "FETCH_SIZE", 1.0)            # a grid nobody named (the clock ramp)
                                                + row(512, "__amd_rocclr_copyBuffer", "FETCH_SIZE", 5.0))
    got = bench.collect_dense_counters(str(tmp_path), {5321 * 256: "yelp2018", 10242 * 256: "amazon-book", 836557 * 256: "synth-10M"}, {})
    assert got == {"yelp2018": {"FETCH_SIZE": [100.0, 102.0]}, "amazon-book": {"FETCH_SIZE": [300.0]}, "synth-10M": {"FETCH_SIZE": [7.0]}}


def test_north_star_figures_are_top_level_scalars_of_roofline():
    """the driver's record keeps top-level scalars of `roofline` only — the HBM fractions the north star is
    stated on (nested under hbm_bound / hbm_reuse_free) and the evaluation's bf16 MFMA fraction are flattened there."""
    import importlib.util

    spec = importlib.util.spec_from_file_location("idg_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    out = {"roofline": {"frac": 1.79, "hbm_bound": {"frac": 0.95, "frac_traffic": 0.88, "us_per_launch": 14330.0,
                                                    "traffic_over_bytes_gather": 0.93, "traffic_l2_hit_rate": 0.11},
                        "hbm_reuse_free": {"frac": 0.78, "frac_traffic": 0.80, "us_per_launch": 17477.0, "traffic_l2_hit_rate": 0.05,
                                           "access_pattern": "random", "strided": {"frac": 0.80, "us_per_launch": 17044.0}}},
           "eval_scale_point": {"bf16_mfma_frac": 0.38, "fp32_equivalent_tflops": 808.0},
           "scale_point": {"ms_per_step": 209.0, "batch_2p20": {"value": 3.08e6, "ms_per_step": 340.0}}}
    bench.flatten_north_star(out)
    rf = out["roofline"]
    assert (rf["hbm_frac_gather"], rf["hbm_frac_traffic"], rf["hbm_us_per_launch"]) == (0.95, 0.88, 14330.0)
    assert (rf["hbm_reuse_free_frac"], rf["hbm_reuse_free_frac_traffic"], rf["hbm_reuse_free_pattern"]) == (0.78, 0.80, "random")
    assert rf["hbm_reuse_free_strided_frac"] == 0.80 and rf["eval_bf16_mfma_frac"] == 0.38 and isinstance(rf["hbm_workload"], str)
    assert out["scale_point_ms_per_step"] == 209.0 and out["scale_point_batch_2p20_triples_per_s"] == 3.08e6
    assert all(not isinstance(v, (dict, list)) for k, v in rf.items() if k.startswith(("hbm_frac", "hbm_us", "hbm_reuse_free_", "eval_")))
    bench.flatten_north_star({"roofline": {"hbm_bound": {"error": "x"}}})  # (legs that failed leave nothing behind)
