"""CPU tests (no GPU needed): the C-ABI library loads and exports every symbol include/evg.h declares, its
default tables equal the oracle's independent restatement, evg_create fails loudly without a device (no CPU
fallback), and the Python host logic mirrors the reference's interface."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from conftest import ROOT


@pytest.fixture(scope="module")
def evg():
    import everglades_amd
    return everglades_amd


def test_every_declared_symbol_is_exported(evg):
    lib = evg.load_library()
    header = open(os.path.join(ROOT, "include", "evg.h")).read()
    declared = set(re.findall(r"\b(evg_[a-z_]+)\s*\(", header))
    assert {"evg_create", "evg_step", "evg_reset", "evg_rollout_random", "evg_get_state", "evg_set_state"} <= declared
    for name in sorted(declared):
        assert hasattr(lib, name), "include/evg.h declares %s but libevg.so does not export it" % name
    assert declared == set(evg._lib.EXPORTS), (declared ^ set(evg._lib.EXPORTS))
    assert lib.evg_abi_version() == evg._lib.ABI_VERSION


def test_default_tables_match_oracle_restatement(evg, oracle_mod):
    a, b = evg.default_tables(), oracle_mod.demo_tables()
    assert C.sizeof(a) == C.sizeof(b)
    assert bytes(a) == bytes(b)
    assert a.node_dist[3][6] == 3 and a.node_dist[6][3] == 3 and a.node_dist[1][3] == 0 and a.node_defense[3] == 1.75
    assert [a.p1_node_map[i] for i in range(12)] == [0, 11, 8, 9, 10, 5, 6, 7, 2, 3, 4, 1]
    assert [a.group_type[0][g] for g in range(12)] == [1, 2, 0] * 4 and a.group_size[1][11] == 12


def _has_gpu():
    import torch
    return torch.cuda.is_available()


def test_create_fails_loudly_without_device(evg):
    if _has_gpu():
        pytest.skip("a HIP device is present")
    lib = evg.load_library()
    cfg = evg._lib.EvgConfig()
    cfg.struct_size, cfg.abi_version, cfg.num_envs, cfg.device_id = C.sizeof(evg._lib.EvgConfig), evg._lib.ABI_VERSION, 4, 0
    cfg.tables = evg.default_tables()
    h = C.c_void_p()
    rc = lib.evg_create(C.byref(cfg), C.byref(h))
    assert rc == -2 and not h.value                              # EVG_ERR_NO_DEVICE
    assert b"no CPU path" in lib.evg_last_error() or b"device" in lib.evg_last_error()
    with pytest.raises(evg.EvgError):
        evg.EvergladesVecEnv(4)                                   # the Python layer refuses too: no silent fallback
    with pytest.raises(evg.EvgError):
        evg.EvergladesEnv(seed=1).reset(players={0: None, 1: None})


def test_config_abi_guard(evg):
    lib = evg.load_library()
    cfg = evg._lib.EvgConfig()
    cfg.struct_size, cfg.abi_version, cfg.num_envs = 12, evg._lib.ABI_VERSION, 1
    h = C.c_void_p()
    assert lib.evg_create(C.byref(cfg), C.byref(h)) == -1        # EVG_ERR_INVALID before any device work
    assert b"mismatch" in lib.evg_last_error()


def test_tables_from_reference_schema_json(evg, tmp_path):
    """A map/unit file pair in the reference's JSON schema (config/DemoMap.json, UnitDefinitions.json) parses to
    the same tables as the built-in defaults."""
    t = evg.default_tables()
    nodes = []
    for i in range(1, 12):
        res = [name for name, bit in (("DEFENSE", 1), ("OBSERVE", 2)) if t.node_resource[i] & bit]
        nodes.append({"ID": i, "Radius": 1.0, "Resource": res, "StructureDefense": t.node_defense[i],
                      "ControlPoints": t.node_control_points[i], "TeamStart": t.node_team_start[i],
                      "Connections": [{"ConnectedID": j, "Distance": t.node_dist[i][j]} for j in range(1, 12) if t.node_dist[i][j]]})
    (tmp_path / "Map.json").write_text(json.dumps({"MapName": "Default", "nodes": nodes}))
    units = [{"Name": n, "Health": t.unit_health[u], "Damage": t.unit_damage[u], "Speed": t.unit_speed[u],
              "Control": t.unit_control[u], "Cost": t.unit_cost[u]} for u, n in enumerate(["Tank", "Controller", "Striker"])]
    (tmp_path / "Units.json").write_text(json.dumps({"units": units}))
    t2 = evg.tables_from_json("Map.json", "Units.json", config_dir=str(tmp_path) + "/")
    assert bytes(t2) == bytes(t)
    with pytest.raises(FileNotFoundError):
        evg.tables_from_json("nope.json", None, config_dir=str(tmp_path))


def test_canonical_actions_follow_the_server(evg):
    ca = evg.canonical_actions
    a = np.array([[3.9, 2.2], [1, 0], [11, 11.99], [0, 5], [5, 5], [6, 6], [7, 7], [8, 8], [9, 9]])
    out = ca(a)
    assert out.dtype == np.int32 and out.shape == (7, 2)
    assert out.tolist() == [[3, 2], [1, 0], [11, 11], [0, 5], [5, 5], [6, 6], [7, 7]]        # first 7 rows, truncation
    assert ca(np.zeros((3, 2))).tolist() == [[0, 0]] * 7                                      # padded with the invalid order
    assert ca(np.array([[12, 1], [-1, 3], [3, 40], [-13, -12]])).tolist()[:4] == [[99, 1], [-1, 3], [3, 99], [99, -12]]   # Python-list domain
    with pytest.raises(AssertionError):
        ca(np.zeros((7, 3)))


def test_single_env_constants_and_spaces(evg):
    env = evg.EvergladesEnv(seed=3)
    assert (env.num_turns, env.num_units, env.num_groups, env.num_nodes, env.num_actions_per_turn) == (150, 100, 12, 11, 7)
    assert env.unit_classes == ["controller", "striker", "tank"]
    assert env.observation_space.shape == (105,)
    assert env.observation_space.low[0] == 1 and env.observation_space.high[0] == 151 and env.observation_space.high[4] == 100
    v = evg.EvergladesVecEnv
    assert (v.num_actions_per_turn, v.obs_len, v.num_groups) == (7, 105, 12)


def test_shard_range_partitions(evg):
    for total, world in [(524288, 8), (65536, 1), (10, 3), (7, 8)]:
        spans = [evg.shard_range(total, world, r) for r in range(world)]
        assert spans[0][0] == 0 and sum(c for _, c in spans) == total
        for (s0, c0), (s1, _) in zip(spans, spans[1:]):
            assert s0 + c0 == s1
    assert evg.shard_range(524288, 8, 3) == (3 * 65536, 65536)


def test_plain_c_client_builds_and_fails_loudly_without_a_device():
    """examples/c_client.c compiles with gcc against include/evg.h alone; on a box without a gfx950 device the library
    refuses to create a handle (no CPU fallback) and the client reports the ABI's error text."""
    import subprocess
    import __graft_entry__ as g
    g.build_c_client()
    exe = os.path.join(g.ROOT, "examples", "c_client")
    try:
        import torch
        if torch.cuda.is_available():
            pytest.skip("a GPU is present: covered by the gpu tests")
    except ImportError:
        pass
    out = subprocess.run([exe, "8", "2", "1"], capture_output=True, text=True, timeout=60)
    assert out.returncode != 0 and "no HIP device" in out.stderr


def test_normal_confidence_interval_matches_scipy(evg):
    """evaluate.py prints statsmodels' proportion_confint(..., 'normal'): p +- z_{1-alpha/2} sqrt(p(1-p)/n), clipped."""
    from scipy.stats import norm
    for count, n, alpha in ((37, 100, 0.05), (0, 50, 0.05), (999, 1000, 0.05), (512, 1024, 0.1), (3, 7, 0.01)):
        q = count / n
        half = norm.ppf(1 - alpha / 2) * np.sqrt(q * (1 - q) / n)
        lo, hi = evg.proportion_confint_normal(count, n, alpha)
        assert abs(lo - max(0.0, q - half)) < 1e-8 and abs(hi - min(1.0, q + half)) < 1e-8


def test_ctypes_mirror_has_the_c_layout(evg, tmp_path):
    """The ctypes structures of _lib.py against the compiler's view of include/evg.h: sizes and the offsets that matter."""
    import subprocess
    src = tmp_path / "layout.c"
    src.write_text('#include <stdio.h>\n#include <stddef.h>\n#include "evg.h"\nint main(void) { printf("%zu %zu %zu %zu %zu %zu %d\\n", sizeof(evg_config), '
                   'sizeof(evg_tables), offsetof(evg_config, tables), offsetof(evg_config, rng_mode), offsetof(evg_tables, node_defense), '
                   'offsetof(evg_tables, max_turns), EVG_ABI_VERSION); return 0; }\n')
    exe = tmp_path / "layout"
    subprocess.check_call(["gcc", "-I" + os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(x) for x in subprocess.check_output([str(exe)]).split()]
    L = evg._lib
    want = [C.sizeof(L.EvgConfig), C.sizeof(L.EvgTables), L.EvgConfig.tables.offset, L.EvgConfig.rng_mode.offset, L.EvgTables.node_defense.offset,
            L.EvgTables.max_turns.offset, L.ABI_VERSION]
    assert got == want


def test_render_is_a_warning_noop(evg):
    """evaluate.py calls env.render() on every step by default: the drop-in must not break that loop."""
    import warnings
    env = evg.EvergladesEnv()
    with warnings.catch_warnings(record=True) as w:
        warnings.simplefilter("always")
        assert env.render() is None and env.render(mode="human") is None
    assert len(w) == 1 and "not part of the accelerated path" in str(w[0].message)
    env.close()


def test_product_library_has_no_diagnostic_hooks(evg):
    """The product library reads no environment variable and exports no diagnostic entry point; the kernel's ablation
    branches, the helper-lane variant of the two-lane kernel (16 envs per wave, 32 idle lanes) and the forced-division switch
    exist only in libevg_diag.so (make diag)."""
    import subprocess
    lib_path = evg._lib.LIB_PATH
    undefined = subprocess.check_output(["nm", "-D", "--undefined-only", lib_path], text=True)
    assert "getenv" not in undefined, "libevg.so must not read the environment"
    defined = subprocess.check_output(["nm", "-D", "--defined-only", lib_path], text=True)
    assert "evg_diag_configure" not in defined and "evg_debug_read_stamps" not in defined
    assert "evg_step_kernelIfLi32" not in defined                  # no helper-lane instantiation of the two-lane step kernel
    src = open(os.path.join(ROOT, "everglades-ai-wargame_amd", "_lib.py")).read() + open(os.path.join(ROOT, "everglades-ai-wargame_amd", "csrc",
                                                                                                      "evg_abi.hip")).read()
    assert "getenv" not in src and "os.environ" not in src
    if os.path.exists(evg._lib.DIAG_LIB_PATH):
        assert "evg_diag_configure" in subprocess.check_output(["nm", "-D", "--defined-only", evg._lib.DIAG_LIB_PATH], text=True)


def test_library_exports_exactly_the_abi(evg):
    """`nm -D` of the product library lists the entry points of include/evg.h and NOTHING else (-fvisibility=hidden + the version script
    csrc/evg.map: no C++ launcher, no kernel stub, no host-side kernel handle); the diagnostic library adds only its configure call."""
    import subprocess

    def exported(path):
        out = subprocess.check_output(["nm", "-D", "--defined-only", path], text=True)
        return sorted(line.split()[-1] for line in out.splitlines() if line.strip())

    assert exported(evg._lib.LIB_PATH) == sorted(evg._lib.EXPORTS)
    if os.path.exists(evg._lib.DIAG_LIB_PATH):
        assert exported(evg._lib.DIAG_LIB_PATH) == sorted(evg._lib.EXPORTS + ["evg_diag_configure"])


def test_create_rejects_bad_arguments_before_touching_a_device(evg):
    """Argument validation of evg_create that comes before device discovery: wrong struct size / ABI version, no envs, global env
    ids beyond the 32-bit key space of the random streams."""
    lib = evg.load_library()

    def create(**kw):
        cfg = evg._lib.EvgConfig()
        cfg.struct_size, cfg.abi_version, cfg.num_envs, cfg.device_id = C.sizeof(evg._lib.EvgConfig), evg._lib.ABI_VERSION, 4, 0
        cfg.tables = evg.default_tables()
        for k, v in kw.items():
            setattr(cfg, k, v)
        h = C.c_void_p()
        rc = lib.evg_create(C.byref(cfg), C.byref(h))
        assert (rc == 0) == bool(h.value)
        if h.value:
            lib.evg_destroy(h)
        return rc, lib.evg_last_error().decode()

    assert create(struct_size=8)[0] == -1 and "mismatch" in create(struct_size=8)[1]
    assert create(abi_version=evg._lib.ABI_VERSION + 1)[0] == -1
    assert create(num_envs=0)[0] == -1
    rc, msg = create(env_id_base=2 ** 32 - 2)
    assert rc == -1 and "2^32" in msg
    assert create(env_id_base=2 ** 40)[0] == -1
    assert create(env_id_base=2 ** 32 - 4)[0] in (0, -2)          # fits exactly: accepted (or no device here)
    assert lib.evg_create(None, None) == -1


def test_reward_by_reciprocal_equals_the_division_in_float32():
    """The step kernel computes the non-terminal reward scores[p] / 3700 (everglades_env.py:63-64) as a float64 product with
    the rounded reciprocal and stores float32: identical to the float32 rounding of the float64 quotient for every score."""
    s = np.arange(0, 1 << 22, dtype=np.float64)
    assert np.array_equal((s / 3700.0).astype(np.float32), (s * np.float64(1.0 / 3700.0)).astype(np.float32))


def test_smart_state_quotients_by_reciprocal_equal_the_division_in_float32():
    """csrc/evg_kernels.hip (evg_smart_state_kernel) takes float32(n / d) of the reference's float64 features (DQNAgent.py:280-294) as float32(n * (1 / d)):
    equal for EVERY numerator the features can have -- turn / 150, control / 100, units / 100, idle groups / 12, average health x alive / 1000."""
    for lo, hi, den in ((0, 256, 150.0), (-511, 512, 100.0), (0, 201, 100.0), (0, 13, 12.0), (0, 128 * 13, 1000.0)):
        n = np.arange(lo, hi).astype(np.float64)
        assert np.array_equal((n / den).astype(np.float32), (n * (1.0 / den)).astype(np.float32)), den


def test_bench_window_helpers():
    """bench.py's desynchronising pre-roll: the episode phase of a global env id is the same function on the device (torch) and
    in the CPU replay (numpy), covers 0..149 about uniformly and is unrelated between neighbouring envs; the hash that ties
    committed counter passes to a build depends on the kernel sources only."""
    import importlib.util
    import torch
    spec = importlib.util.spec_from_file_location("evg_bench", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    ids = np.arange(70000, 70000 + 65536, dtype=np.int64)
    ph = bench.episode_phase(ids)
    assert np.array_equal(ph, bench.episode_phase(torch.arange(70000, 70000 + 65536, dtype=torch.int64)).numpy())
    assert ph.min() == 0 and ph.max() == bench.PHASES - 1
    cnt = np.bincount(ph, minlength=bench.PHASES)
    assert cnt.min() > 0.8 * 65536 / bench.PHASES and cnt.max() < 1.2 * 65536 / bench.PHASES
    per_wave = ph.reshape(-1, 32)                                   # the 32 envs of a wavefront: a spread of phases, not a window
    assert (per_wave.max(axis=1) - per_wave.min(axis=1) > 75).mean() > 0.99
    h = bench.kernel_source_hash()
    assert len(h) == 16 and h == bench.kernel_source_hash()
    sys_path_tools = os.path.join(ROOT, "tools")
    spec2 = importlib.util.spec_from_file_location("evg_prof", os.path.join(sys_path_tools, "_prof.py"))
    prof = importlib.util.module_from_spec(spec2)
    spec2.loader.exec_module(prof)
    assert prof.kernel_source_hash() == h                           # the profile summaries are keyed by the same hash
    for kind in ("pmc_traffic", "sq_counters"):                     # whatever is committed for THIS build parses and has what bench.py reads
        d = bench.committed_counters(kind, 65536, "random", "float32")
        if d is not None:
            if kind == "pmc_traffic":
                for form in ("persistent", "one_launch_per_turn"):
                    f = d["forms"][form]
                    assert 971 <= f["bytes_per_env_step_steady"] < 4530 and f["state_round_trip_bytes_per_env"] in (0, 320)
            else:
                assert 1000 < d["kernels"]["persistent"]["valu_insts_per_wave_turn"] < 10000


def test_bench_times_the_region_several_times_and_reports_the_median():
    """bench.py times the exact K-step region R times and reports the median region (round 5: a 0.4 ms region is one draw from a +-8 % distribution).  The rule
    for R depends on K only -- every rank of a multi-GPU job must time the same number of regions -- and the median is the lower middle one."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("evg_bench3", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    assert bench.repeats_for(20) == 9 and bench.repeats_for(450) == 9 and bench.repeats_for(2499) == 9 and bench.repeats_for(2500) == 1 and bench.repeats_for(100000) == 1
    assert bench.repeats_for(20, forced=1) == 1 and bench.repeats_for(100000, forced=3) == 3
    assert bench.median_regions([5.0]) == [0]
    assert bench.median_regions([0.41, 0.37, 0.36, 0.39, 0.36, 0.35, 0.40, 0.38, 0.42]) == [7]      # 0.38: four regions below, four above
    assert bench.median_regions([3.0, 1.0, 2.0, 4.0]) == [2, 0]                                        # even count: BOTH middle ones, 2.0 and 3.0 ...
    s = bench.summarise_regions([3.0, 1.0, 2.0, 4.0], [30.0, 10.0, 20.0, 40.0], [None] * 4, steps=10, total=100)
    assert s["seconds"] == 2.5 and s["kernel_ms_sum"] == 25.0 and s["collective_ms"] is None         # ... and the line reports their mean (a true median)
    assert s["timing"]["reported"] == "mean of the two middle regions" and s["timing"]["min_value"] == 250.0 and s["timing"]["max_value"] == 1000.0
    s = bench.summarise_regions([0.3, 0.1, 0.2], [3.0, 1.0, 2.0], [0.03, 0.01, 0.02], steps=20, total=65536)
    assert s["median_indices"] == [2] and s["seconds"] == 0.2 and s["kernel_ms_sum"] == 2.0 and s["collective_ms"] == 0.02


def test_bench_stdout_line_is_compact_and_complete():
    """bench.py prints the COMPACT form of its result: a driver that keeps only the tail of the output must still see one whole JSON
    object.  Every full line committed under profiles/ for this round compacts to less than 3.6 KB (the size of a line a driver is known
    to have captured whole; round 5 added the timing block, survey_8d_frac / hbm_proper_frac and one more leg within the same size) and keeps every field of the contract: the metric block, config.workload, roofline {bound, achieved, peak,
    unit, frac, traffic}, cpu_baseline {value, unit, cores, kind, sample} and the numbers of both per-turn legs."""
    import glob
    import importlib.util
    import json
    spec = importlib.util.spec_from_file_location("evg_bench2", os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    files = sorted(f for f in glob.glob(os.path.join(ROOT, "profiles", "r06_e_bench_*.json")) if "compact" not in f)
    assert len(files) >= 7
    for f in files:
        full = json.loads(open(f).read())
        line = json.dumps(bench.compact_line(full), separators=(",", ":"))
        # (a one-rank REHEARSAL carries the single-GPU legs and the distributed block at once; a real N > 1 line has no extra legs)
        assert len(line) < (4200 if "rehearse_rccl" in f else 3600), (f, len(line))
        c = json.loads(line)
        for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                  "config", "roofline"):
            assert k in c, (f, k)
        assert abs(c["value"] / full["value"] - 1) < 1e-5 and c["metric"] == full["metric"] and "workload" in c["config"]
        for k in ("bound", "achieved", "peak", "unit", "frac", "traffic"):
            assert k in c["roofline"], (f, k)
        assert abs(c["roofline"]["frac"] / full["roofline"]["frac"] - 1) < 1e-3
        # round 5: the line says what the bytes are -- "fabric" when a round's working set is cache-resident (then with the HBM figure proper beside it) -- and
        # carries SURVEY 8(d)'s byte model priced at the same kernel time (above 1: not applicable to a design that keeps the state on chip), and how the
        # region was timed
        assert c["roofline"]["bound"] in ("fabric", "hbm") and c["roofline"]["survey_8d_frac"] > 0
        if "bound_detail" in c["roofline"]:          # round 6: `bound` is always the contract's value, the qualifier is a field of its own
            assert c["roofline"]["bound"] == "hbm" and c["roofline"]["bound_detail"] == "fabric" and "bound_detail_is" in c["roofline"]
        if full["config"]["envs_per_gpu"] == 65536 and full["config"]["workload"].startswith("65536 concurrent DemoMap games per GPU, random") and "float32" in full["config"]["workload"]:
            assert 0.3 < c["roofline"]["hbm_proper_frac"] < 0.9 and "cycled" in full["roofline"]["hbm_proper_source"]
            if "rehearse" not in f:                                   # (two gloo ranks share one GPU: that line's own fraction says nothing)
                assert c["roofline"]["hbm_proper_frac"] < c["roofline"]["frac"] + 0.05      # leaving the cache does not make the kernel faster
        t = c["timing"]
        assert t["repeats"] >= 1 and t["min_ms_per_step"] <= c["ms_per_step"] <= t["max_ms_per_step"] and t["min_value"] <= c["value"] <= t["max_value"]
        if "cpu_baseline" in full:
            for k in ("value", "unit", "cores", "kind", "sample"):
                assert k in c["cpu_baseline"]
            assert c["cpu_baseline"]["reference_python_env_steps_per_s"] == [529, 554]       # the reference's own rate, as numbers, in the line
        # round 6: the cold figure at top level beside `value`, one long region (`sustained`) beside the short ones, frac priced at ms_per_step
        if full["n_gpus"] == 1 and "distributed" not in full:
            assert c["value_cold"] and c["value_cold"] < c["value"] * 1.02
            assert c["sustained"]["turns"] == 450 and 2.0 < c["sustained"]["region_ms"] < 12.0 and c["sustained"]["kernel_ms"] <= c["sustained"]["ms_per_step"]
        assert c["roofline"]["frac"] <= c["roofline"]["frac_kernel_events"] * 1.001
        assert abs(c["roofline"]["frac"] - c["roofline"]["bytes_per_env_step"] * full["config"]["envs_per_gpu"] / (c["ms_per_step"] * 1e-3) / 1e9 / 8000.0) < 2e-3 * c["roofline"]["frac"] or "distributed" in full
        for leg in ("one_launch_per_turn", "caller_actions_per_turn", "learner_vs_bot_per_turn"):
            if full["config"].get(leg):
                assert c["config"][leg]["roofline"]["frac"] > 0 and c["config"][leg]["kernel_ms"] <= c["config"][leg]["ms_per_step"]
        if full["config"].get("pipelined_halves_per_turn"):
            leg = c["config"]["pipelined_halves_per_turn"]
            assert leg["parts"] == 2 and leg["kernel_ms"] <= leg["ms_per_step"]
        if "distributed" in full:
            d = c["distributed"]
            assert d["collective_us"] > 0 and d["gathered_wins_equal_sum_of_per_rank_counts"] is True and "expected" in d and 0 < d["step_share_of_region"] < 1
    # what a first multi-GPU line is read against: N x the committed one-rank RCCL rehearsal of the same shape, next to the one-GPU line of that shape
    exp = bench.expected_if_wire_free(8, 20)
    assert exp and "r06_e_bench_rehearse_rccl_1rank.json" in exp["from"] and abs(exp["value_if_wire_free"] / (8 * exp["per_gpu"]) - 1) < 1e-12
    # step launches / (step launches + collective path), one run
    assert 0.9 < exp["weak_scaling_efficiency_if_wire_free"] < 1.0 and exp["collective_us_1rank"] > 0
    assert bench.expected_if_wire_free(8, 12345) is None


def test_a_missing_rccl_is_reported_as_evg_err_comm(tmp_path):
    """include/evg.h promises EVG_ERR_COMM from evg_comm_unique_id / evg_comm_init / evg_gather_returns on a host without RCCL.  Every box here has RCCL, so
    the dlopen is made to fail: a preloaded shim turns every dlopen of a name containing "rccl" into a dlopen of a file that does not exist (which fails
    AND sets dlerror(), the case the library's message is built from -- dlerror() hands its text out once).  Fresh process, no torch: nothing has loaded
    librccl before."""
    import subprocess
    import sys
    shim_src = tmp_path / "no_rccl.c"
    shim_src.write_text('#define _GNU_SOURCE\n#include <dlfcn.h>\n#include <string.h>\n'
                        'void* dlopen(const char* name, int flags) {\n'
                        '    static void* (*real)(const char*, int);\n'
                        '    if (!real) real = (void* (*)(const char*, int))dlsym(RTLD_NEXT, "dlopen");\n'
                        '    if (name && strstr(name, "rccl")) return real("/nonexistent/evg-test/librccl.so.1", flags);\n'
                        '    return real(name, flags);\n}\n')
    shim = tmp_path / "no_rccl.so"
    subprocess.check_call(["gcc", "-shared", "-fPIC", "-O1", str(shim_src), "-o", str(shim), "-ldl"])
    lib_path = os.path.join(ROOT, "everglades-ai-wargame_amd", "libevg.so")
    code = ("import ctypes as C\n"
            "lib = C.CDLL(%r)\n"
            "lib.evg_last_error.restype = C.c_char_p\n"
            "buf = C.create_string_buffer(128)\n"
            "rc = lib.evg_comm_unique_id(buf)\n"
            "print(rc, lib.evg_last_error().decode())\n" % lib_path)
    out = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, LD_PRELOAD=str(shim)), capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    rc, msg = out.stdout.strip().split(" ", 1)
    assert int(rc) == -6 and "RCCL is not available" in msg and "librccl.so.1 not found (" in msg and "/nonexistent/evg-test" in msg, out.stdout


def test_gym_registration_runs_with_a_gym_package():
    """INTEGRATION.md section 1 offers `gym.make('everglades-v0')` (the reference: gym_everglades/__init__.py:3-6).  gym is in neither image, so the
    registration line is executed here against a stand-in `gym` whose register() records its arguments and whose make() resolves the
    'module:Class' entry point the way gym does: the id is registered once, the entry point imports, and the object has the reference's interface.
    A failure inside register() must propagate (only a missing gym is tolerated)."""
    import subprocess
    import sys
    code = '''
import importlib, sys, types
calls = []
gym = types.ModuleType("gym"); envs = types.ModuleType("gym.envs"); reg = types.ModuleType("gym.envs.registration")
def register(**kw): calls.append(kw)
def make(id):
    spec = [c for c in calls if c["id"] == id][0]["entry_point"]
    mod, cls = spec.split(":")
    return getattr(importlib.import_module(mod), cls)()
reg.register = register; envs.registration = reg; gym.envs = envs; gym.make = make
sys.modules.update({"gym": gym, "gym.envs": envs, "gym.envs.registration": reg})
sys.path.insert(0, %r)
import everglades_amd
assert everglades_amd.GYM_REGISTERED is True
assert calls == [dict(id="everglades-v0", entry_point="everglades_amd:EvergladesEnv")], calls
env = gym.make("everglades-v0")
assert type(env) is everglades_amd.EvergladesEnv and env.num_actions_per_turn == 7 and env.observation_space.shape == (105,)
for name in ("reset", "step", "render", "close"):
    assert callable(getattr(env, name))
# a registration that fails is an error, not a silent skip
def broken(**kw): raise RuntimeError("id already registered")
reg.register = broken
for m in [m for m in sys.modules if m == "everglades_amd" or m.startswith("everglades_amd.")]:
    del sys.modules[m]
try:
    import everglades_amd
except RuntimeError as e:
    assert "already registered" in str(e)
else:
    raise AssertionError("a failing register() was swallowed")
print("ok")
''' % ROOT
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
    import everglades_amd
    assert everglades_amd.GYM_REGISTERED is False          # this image has no gym: nothing was registered, nothing failed


def test_tables_from_json_refuses_what_the_reference_would_play_differently(evg, tmp_path):
    """tables_from_json's domain is what tests/golden/custom_*.npz pin against the reference.  Files the reference would play DIFFERENTLY from
    the integer tables are refused instead of being approximated: nodes out of ID order (the reference's fog mask mixes list positions with IDs,
    server.py:409-418), a 'DEFEND' resource (switches the fortress bonus on, server.py:595), fractional unit stats / distances / control points
    (the reference would compute with the float), a map without exactly one start node per player."""
    import copy
    import custom_configs as cc

    def write(name, obj):
        path = tmp_path / name
        path.write_text(json.dumps(obj))
        return str(path)

    good_map, good_units = write("m.json", cc.MAP_A), write("u.json", cc.UNITS_A)
    t = evg.tables_from_json(good_map, good_units)
    assert t.node_dist[2][3] == 3 and t.node_dist[3][2] == 4 and t.node_dist[6][5] == 2 and t.node_dist[5][6] == 0      # directed, one-way
    assert t.node_resource[5] == 3 and t.node_resource[9] == 0 and t.node_defense[2] == 2.1 and t.node_control_points[6] == 511
    assert [t.group_type[0][g] for g in range(3)] == [2, 0, 1] and t.unit_speed[0] == 3                                 # ids follow the file order

    def bad_map(edit):
        m = copy.deepcopy(cc.MAP_A)
        edit(m)
        with pytest.raises(ValueError):
            evg.tables_from_json(write("bad.json", m), good_units)

    bad_map(lambda m: m["nodes"].reverse())
    bad_map(lambda m: m["nodes"][3]["Resource"].append("DEFEND"))
    bad_map(lambda m: m["nodes"][0]["Connections"][0].update(Distance=2.5))
    bad_map(lambda m: m["nodes"][4].update(ControlPoints=100.5))
    bad_map(lambda m: m["nodes"][4].update(TeamStart=0))
    bad_map(lambda m: m["nodes"][0].update(TeamStart=-1))
    bad_map(lambda m: m["nodes"][0]["Connections"][0].update(ConnectedID=12))
    u = copy.deepcopy(cc.UNITS_A)
    u["units"][1]["Health"] = 2.5
    with pytest.raises(ValueError):
        evg.tables_from_json(good_map, write("bad_u.json", u))
    u = copy.deepcopy(cc.UNITS_B)
    u["units"].append(dict(u["units"][0], Name="Fifth"))
    with pytest.raises(ValueError):
        evg.tables_from_json(good_map, write("bad_u5.json", u))


def _load_bench(name):
    import importlib.util
    spec = importlib.util.spec_from_file_location(name, os.path.join(ROOT, "bench.py"))
    bench = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(bench)
    return bench


def test_bench_roofline_pricing_from_a_saved_counter_pass():
    """bench.py's hbm_roofline is a pure function of a committed counter summary and the measured times: priced here against the saved pass
    profiles/r05_d_pmc_traffic.json with hand-computed expectations, incl. the re-scaling of the state round trip to the turns per launch that were timed,
    the two time bases (`frac` from the ms_per_step of the region `value` comes from, `frac_kernel_events` from the launches' own HIP-event time) and the
    fallback to the unavoidable output bytes when profiles/ holds no pass of the running build (a lower bound, and the object says so)."""
    bench = _load_bench("evg_bench_roof")
    pmc = json.load(open(os.path.join(ROOT, "profiles", "r05_d_pmc_traffic.json")))
    pmc["_file"] = "profiles/r05_d_pmc_traffic.json"
    form = pmc["forms"]["persistent"]
    n = 65536
    for tpl_timed in (150.0, 20.0):
        bpe = form["bytes_per_env_step_steady"] + form["state_round_trip_bytes_per_env"] / tpl_timed
        r = bench.hbm_roofline(pmc, "persistent", 0.0150, tpl_timed, n, "float32", "random", "abc", region_ms_per_step=0.0160)
        assert r["bound"] == "hbm" and r["peak"] == 8000.0 and r["unit"] == "GB/s"
        assert abs(r["bytes_per_env_step"] - bpe) < 1e-9 and r["bytes_source"] == "profiles/r05_d_pmc_traffic.json [persistent]"
        assert abs(r["achieved"] - bpe * n / 0.0160e-3 / 1e9) < 1e-6 and abs(r["frac"] - r["achieved"] / 8000.0) < 1e-12
        assert abs(r["achieved_kernel_events"] - bpe * n / 0.0150e-3 / 1e9) < 1e-6 and r["frac_kernel_events"] > r["frac"]
        assert abs(r["traffic"] - bpe * n * tpl_timed) < 1e-3 and r["traffic_unit"] == "bytes per launch"
        assert 1300 < r["algorithmic_bytes_per_env_step"] < 1400 and 1.0 < r["traffic_over_algorithmic"] < 1.15      # 971 + 2 x rows written
        assert abs(r["frac_algorithmic"] - r["frac"] / r["traffic_over_algorithmic"]) < 1e-12
    assert 1400 < bench.hbm_roofline(pmc, "persistent", 0.015, 150.0, n, "float32")["bytes_per_env_step"] < 1450         # the r05 figure: 1 425 B
    # a per-turn leg prices its stream time: both pairs are the same figure
    r = bench.hbm_roofline(pmc, "learner_vs_bot_per_turn", 0.030, 1, n, "float32")
    assert r["frac"] == r["frac_kernel_events"] and r["mandatory_output_bytes_per_env_step"] == 420 + 112 + 19
    # no pass of this build: the mandatory outputs, a lower bound, traffic null
    r = bench.hbm_roofline(None, "persistent", 0.0150, 150.0, n, "float32", "random", "deadbeef", region_ms_per_step=0.0160)
    assert r["traffic"] is None and r["bytes_per_env_step"] == 971 and "lower bound" in r["bytes_source"] and "deadbeef" in r["bytes_source"]
    assert abs(r["frac"] - 971 * n / 0.0160e-3 / 1e9 / 8000.0) < 1e-12
    # committed_counters() refuses a pass of another build (hash mismatch) -- what makes the fallback above happen
    assert bench.committed_counters("pmc_traffic", 65536, "random", "float32") is None or \
        bench.committed_counters("pmc_traffic", 65536, "random", "float32")["kernel_source_hash"] == bench.kernel_source_hash()
    # cache-resident working set: `bound` stays the contract's "hbm", the qualifier is a field of its own
    roof = bench.mark_cache_resident({"bound": "hbm"}, pmc, n, "float32")
    assert roof["bound"] == "hbm" and roof["bound_detail"] == "fabric"
    assert "bound_detail" not in bench.mark_cache_resident({"bound": "hbm"}, None, n, "float32")
    assert bench.round_working_set_bytes(262144, "float32") == bench.round_working_set_bytes(65536, "float32") < (256 << 20)
    cyc = json.load(open(os.path.join(ROOT, "profiles", "r05_d_262144envs_cycled_pmc_traffic.json")))
    cyc["_file"] = "profiles/r05_d_262144envs_cycled_pmc_traffic.json"
    bm, fields = bench.beyond_the_cache(pmc, None, cyc, None, "float32")
    assert 0.4 < fields["hbm_proper_frac"] < 0.7 and "cycled" in fields["hbm_proper_source"] and bm["diag_library_chunked_over_262144_envs"]["working_set_MB"] > 600
    assert bench.beyond_the_cache(None, None, cyc, None, "float32") == (None, {})
    sq = json.load(open(os.path.join(ROOT, "profiles", "r05_d_sq_counters.json")))
    sq["_file"] = "profiles/r05_d_sq_counters.json"
    v = bench.valu_roofline(sq, "persistent", n, 0.0141)
    assert 0.6 < v["frac"] < 0.85 and v["valu_insts_per_wave_turn"] > 2500 and bench.valu_roofline(None, "persistent", n, 0.0141) is None


def test_bench_line_says_where_every_rank_sat_and_what_the_reference_does():
    """The first multi-GPU line must prove itself (N ranks on N distinct GPUs), and a reader of the compact line must see the long region, the cold figure
    and the Python reference's own rate: compact_line() of a synthetic full object keeps all of them within the size a driver is known to capture."""
    bench = _load_bench("evg_bench_ident")
    ids = [dict(local_rank=r, device_index=r, name="AMD Instinct MI355X", pci="0000:%02x:00" % (5 + r), uuid="GPU-%04d" % r) for r in range(8)]
    assert bench.check_distinct_devices(ids, "nccl", False) == {"distinct_devices": True, "identified_by": "uuid", "devices_seen": 8}
    no_uuid = [dict(i, uuid=None) for i in ids]
    assert bench.check_distinct_devices(no_uuid, "nccl", False)["identified_by"] == "pci"
    placeholder = [dict(i, uuid="00000000-0000-0000-0000-000000000000") for i in ids]                      # one uuid for every card, PCI addresses differ: still N GPUs
    assert bench.check_distinct_devices(placeholder, "nccl", False) == {"distinct_devices": True, "identified_by": "pci", "devices_seen": 8}
    twice = [ids[0], dict(ids[0], local_rank=1)]
    with pytest.raises(SystemExit):
        bench.check_distinct_devices(twice, "nccl", False)                     # two RCCL ranks on one GPU: not a one-rank-per-GPU run
    assert bench.check_distinct_devices(twice, "gloo", True) == {"distinct_devices": False, "identified_by": "uuid", "devices_seen": 1}   # a rehearsal records it
    full = json.load(open(os.path.join(ROOT, "profiles", "r05_d_bench_rehearse_rccl_1rank.json")))
    full["value_cold"], full["value_protocol"] = 3.2e9, "..."
    full["sustained"] = dict(turns=450, launches=3, turns_per_launch=150, region_ms=6.41, ms_per_step=0.01424, value=4.6e9, kernel_ms=0.0141, what="...")
    full["cpu_baseline"] = dict(full.get("cpu_baseline") or dict(value=9.6e6, unit="env-steps/s", cores=16, kind="port", obs_dtype="float64", sample="8192 envs x 1 turns (random vs random"),
                                reference_python_env_steps_per_s=list(bench.REFERENCE_PYTHON_ENV_STEPS_PER_S), reference_python_source="BASELINE.md section 2 ...")
    d = full["distributed"]
    d["per_rank"] = [dict(p, **ids[i]) for i, p in enumerate(d["per_rank"])]
    d.update(bench.check_distinct_devices(ids[:len(d["per_rank"])], "nccl", True), rccl_ranks_seen=1, evg_comm_ranks=None)
    c = json.loads(json.dumps(bench.compact_line(full), separators=(",", ":")))
    assert c["value_cold"] == 3200000000 and c["sustained"]["turns"] == 450 and c["sustained"]["region_ms"] == 6.41 and c["sustained"]["value"] == 4600000000
    assert c["cpu_baseline"]["reference_python_env_steps_per_s"] == [529, 554] and c["cpu_baseline"]["kind"] == "port"
    pr = c["distributed"]["per_rank"]
    assert pr["local_rank"] == [0] and pr["device_index"] == [0] and pr["pci"] == ["0000:05:00"] and c["distributed"]["device_names"] == ["AMD Instinct MI355X"]
    assert c["distributed"]["distinct_devices"] is True and c["distributed"]["rccl_ranks_seen"] == 1
    # eight ranks still fit
    d["per_rank"] = [dict(d["per_rank"][0], **ids[i], rank=i) for i in range(8)]
    assert len(json.dumps(bench.compact_line(full), separators=(",", ":"))) < 5200


def test_both_lane_mappings_cite_the_same_reference_lines():
    """csrc/evg_step4.inc (four lanes per env) is a hand-maintained second copy of the turn next to the two-lane fragments (step_orders / step_combat /
    step_move_capture / step_outputs .inc).  The parity suite keeps their RESULTS equal on the GPU; this table-driven check keeps their STRUCTURE comparable on
    the CPU: both walk through the phase markers PHASE(1) .. PHASE(13) in the same order, and between the same two markers both cite the same anchor ranges of
    the reference (server.py / everglades_env.py line ranges in the comments) -- so a rule that moves or is re-derived in one copy shows up as a mismatch here."""
    csrc = os.path.join(ROOT, "everglades-ai-wargame_amd", "csrc")
    two = "".join(open(os.path.join(csrc, f)).read() for f in ("step_orders.inc", "step_combat.inc", "step_move_capture.inc", "step_outputs.inc"))
    four = open(os.path.join(csrc, "evg_step4.inc")).read()
    four = four[four.index("if (MULTI) { PHASE(0); }"):]                  # the turn loop (the prologue's STAMP(0) / PHASE(0) belong to step_kernel.inc there)

    def phases(src):
        parts = re.split(r"PHASE\((\d+)\);", src)
        order, text, prev = [], {}, parts[0]
        for i in range(1, len(parts), 2):
            k = int(parts[i])
            order.append(k)
            text[k] = prev
            prev = parts[i + 1]
        return order, text

    def cited(text):
        out = set()
        for line in text.split("\n"):
            if "//" in line:
                for m in re.finditer(r"(?:\.py)?:(\d{2,3})(?:-(\d{2,3}))?\b", line.split("//", 1)[1]):
                    a, b = int(m.group(1)), int(m.group(2) or m.group(1))
                    if b >= a:
                        out.add((a, b))
        return out

    # phase k = the code in front of PHASE(k): what it implements in the reference
    anchors = {2: [(218, 271)],                                    # order application, server.py:218-271
               3: [(503, 654)],                                    # combat (stage 0: who fights), server.py:503-654
               5: [(549, 566)],                                    # the draws and the infliction table, :549-566
               6: [(573, 644), (592, 597), (601, 601), (609, 609)],    # damage application: node defence, the quotient, the subtraction
               7: [(656, 706)],                                    # movement
               8: [(708, 767), (291, 317), (321, 328)],            # capture, node / unit scores, status precedence (game_end)
               9: [(37, 61), (133, 209)],                          # rewards (everglades_env.py:37-61), auto-reset = game_init (server.py:133-209)
               10: [(382, 455), (457, 501), (158, 171)]}           # board_state, player_state, observation assembly (everglades_env.py:158-171)
    o2, t2 = phases(two)
    o4, t4 = phases(four)
    assert o2 == list(range(1, 14)) and o4[-13:] == list(range(1, 14)), (o2, o4)
    for k, want in anchors.items():
        c2, c4 = cited(t2[k]), cited(t4[k])
        for rng in want:
            assert rng in c2, ("two-lane fragments, phase %d: no citation of lines %d-%d" % ((k,) + rng))
            assert rng in c4, ("evg_step4.inc, phase %d: no citation of lines %d-%d" % ((k,) + rng))


def test_kernel_source_hash_is_about_code_not_comments(evg, tmp_path):
    """The hash that ties a counter pass under profiles/ to a build covers the kernel CODE: comments and white space do not change it (a documentation change
    in include/evg.h must not invalidate the evidence), any token does; string and character literals are kept verbatim."""
    L = evg._lib
    a = 'int f(int x) { return x + 1; }  // adds one\n/* block\n comment */ const char* s = "// kept /* kept */";\n'
    b = 'int   f(int x)\n{\n    return x + 1;   /* another comment */ }\nconst char* s = "// kept /* kept */";'
    assert L._code_only(a) == L._code_only(b)
    assert L._code_only(a) != L._code_only(a.replace("x + 1", "x + 2")) and L._code_only(a) != L._code_only(a.replace("// kept", "// Kept"))
    assert L._code_only("char c = '\\\"'; // q") == "char c = '\\\"';"
    h = L.kernel_source_hash()
    assert len(h) == 16 and h == L.kernel_source_hash()


def test_no_cpp_exception_can_cross_the_abi():
    """include/evg.h promises that no C++ exception crosses the ABI.  The host side allocates std::vector / std::string temporaries (state exchange, launch-plan
    text, the RCCL loader), so every multi-line int-returning entry point of csrc/evg_abi.hip is a function-try-block ending in on_exception(), which maps
    std::bad_alloc to EVG_ERR_ALLOC; the one-liners (evg_abi_version, evg_num_envs) allocate nothing."""
    src = open(os.path.join(ROOT, "everglades-ai-wargame_amd", "csrc", "evg_abi.hip")).read().split("\n")
    defs, guarded = [], []
    for i, line in enumerate(src):
        m = re.match(r"^(EVG_API )?int (evg_\w+)\(", line)
        if not m or line.rstrip().endswith(";") or line.rstrip().endswith("}"):
            continue
        j = i
        while not (src[j].rstrip().endswith(") {") or src[j].rstrip().endswith(") try {")):
            j += 1
        defs.append(m.group(2))
        if src[j].rstrip().endswith(") try {"):
            k = j + 1
            while not src[k].startswith("}"):
                k += 1
            assert src[k] == "} catch (...) { return on_exception(); }", (m.group(2), src[k])
            guarded.append(m.group(2))
    assert len(defs) >= 40 and defs == guarded, sorted(set(defs) - set(guarded))
    text = "\n".join(src)
    assert "catch (const std::bad_alloc&) { return fail(EVG_ERR_ALLOC" in text
