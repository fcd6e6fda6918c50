"""bench.py: the pieces that need no GPU (roofline bookkeeping, defaults), and on the GPU the
JSON line itself -- the keys the driver reads, and that the numbers in it agree with each other."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402

LINE_KEYS = ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better",
             "scaling", "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline")
ROOFLINE_KEYS = ("bound", "achieved", "peak", "unit", "frac", "traffic")
CPU_KEYS = ("value", "unit", "cores", "kind", "sample")


def _timings(**ms):
    return {name: {"ms": ms.get(name, 0.0) * 4, "launches": 4 if name in ms else 0} for name in bench.BYTES}


def test_defaults_finish_in_minutes_on_one_gpu(monkeypatch):
    monkeypatch.setattr(sys, "argv", ["bench.py"])
    a = bench.parse_args()
    assert (a.gpus, a.steps, a.warmup) == (1, 100, 20)
    assert a.particles_log2 == 24 and a.grid == "" and a.scene == "ellipsoid" and a.mode == "chaos"


def test_roofline_names_the_longest_launch_and_pairs_the_p2g_passes():
    n, cells = 1 << 24, 256 ** 3
    roof, per = bench.roofline_of(_timings(clearGrid=0.05, p2g1=0.14, p2g2=0.59, updateGrid=0.07, g2p=0.63), n, cells)
    assert roof["kernel"] == "g2p" and roof["bound"] == "hbm" and roof["unit"] == "GB/s"
    want = (96 * n + 16 * cells) / 0.63e-3 / 1e9
    assert abs(roof["achieved"] - want) < 0.1 and abs(roof["frac"] - want / 8000.0) < 1e-4
    assert abs(per["p2g"]["avg_ms"] - 0.73) < 1e-9
    # when one of the two p2g passes is the longest launch the pair is judged: 116 N + 32 G bytes
    roof, per = bench.roofline_of(_timings(clearGrid=0.05, p2g1=0.14, p2g2=0.70, updateGrid=0.07, g2p=0.63), n, cells)
    assert roof["kernel"] == "p2g"
    want = (116 * n + 32 * cells) / 0.84e-3 / 1e9
    assert abs(roof["achieved"] - want) < 0.1
    assert set(ROOFLINE_KEYS) <= set(roof)


def test_algorithmic_bytes_per_step_add_up():
    per_particle = sum(b[0] for b in bench.BYTES.values())
    per_cell = sum(b[1] for b in bench.BYTES.values())
    assert (per_particle, per_cell) == (212, 112)


def test_measured_traffic_comes_from_the_committed_counter_passes():
    key = "ellipsoid_%d_256x256x256" % (1 << 24)
    for kernel in ("p2g", "g2p", "updateGrid"):
        assert bench.measured_traffic(kernel, key) > 0
    assert bench.measured_traffic("clearGrid", key) is None      # flowmpm_step launches no clearGrid
    assert bench.measured_traffic("g2p", "some other workload") is None


def test_scene_uniforms_fix_the_rest_density_to_the_fill():
    u = bench.scene_uniforms(1 << 20, 128 ** 3, "ellipsoid")
    assert u["dt"] == 0.1 and u["noise"] == 0.0
    assert abs(u["rest_density"] - (1 << 20) / (0.268 * 128 ** 3)) < 1e-9
    assert bench.scene_uniforms(1 << 20, 128 ** 3, "ellipsoid", noise=1.0)["noise"] == 1.0


def test_a_plain_multi_gpu_command_starts_the_ranks_as_a_child(monkeypatch, capsys):
    """`python bench.py --gpus 2` with no WORLD_SIZE: torch.distributed.run is started as a child
    process (never exec), with the same arguments, before torch is imported; rank 0's line and the
    child's status are relayed."""
    seen = {}

    class Done:
        returncode = 0
        stdout = "some log line\n" + json.dumps({"n_gpus": 2, "value": 1.0}) + "\n"

    def fake_run(cmd, **kw):
        seen["cmd"], seen["kw"] = cmd, kw
        return Done()

    monkeypatch.delenv("WORLD_SIZE", raising=False)
    monkeypatch.setattr(sys, "argv", ["bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1"])
    monkeypatch.setattr(bench.subprocess, "run", fake_run)
    with pytest.raises(SystemExit) as exit_info:
        bench.main()
    assert exit_info.value.code == 0
    cmd = seen["cmd"]
    assert cmd[:3] == [sys.executable, "-m", "torch.distributed.run"]
    assert cmd[cmd.index("--nproc-per-node") + 1] == "2" and "--nnodes=1" in cmd
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1"
    script = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[script + 1:] == ["--gpus", "2", "--steps", "3", "--warmup", "1"]
    out = capsys.readouterr()
    assert json.loads(out.out.strip()) == {"n_gpus": 2, "value": 1.0}      # ONE line on stdout
    assert "some log line" in out.err

    class Failed(Done):
        returncode = 7
        stdout = ""
    monkeypatch.setattr(bench.subprocess, "run", lambda cmd, **kw: Failed())
    with pytest.raises(SystemExit) as exit_info:
        bench.main()
    assert exit_info.value.code == 7


def _last_json_line(text):
    lines = [l for l in text.strip().splitlines() if l.startswith("{")]
    assert lines, text[-2000:]
    return json.loads(lines[-1])


@pytest.mark.gpu
def test_the_single_gpu_line():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "6", "--warmup", "2",
                        "--particles-log2", "18", "--grid", "64,64,64", "--cpu-steps", "1", "--timing-stride", "2",
                        "--steady-windows", "2", "--steady-steps", "20"],
                       capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-3000:]
    d = _last_json_line(r.stdout)
    assert set(LINE_KEYS) <= set(d)
    assert d["metric"] == "particle-substeps/sec" and d["n_gpus"] == 1 and d["steps"] == 6 and d["warmup"] == 2
    assert d["higher_is_better"] is True and d["scaling"] == "weak" and d["vs_baseline"] is None
    assert d["data"] == "synthetic" and "workload" in d["config"] and "model" not in d["config"]
    n = d["config"]["particles"]
    assert n == 1 << 18
    assert abs(d["value"] - n / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    roof = d["roofline"]
    assert set(ROOFLINE_KEYS) <= set(roof) and roof["bound"] == "hbm" and roof["peak"] == 8000.0
    assert abs(roof["frac"] - roof["achieved"] / roof["peak"]) < 1e-3
    assert roof["avg_launch_ms"] > 0 and roof["kernel"] in ("p2g", "g2p", "clearGrid", "updateGrid")
    cpu = d["cpu_baseline"]
    assert set(CPU_KEYS) <= set(cpu) and cpu["kind"] == "port" and cpu["cores"] >= 1 and cpu["value"] > 0
    assert d["value"] > cpu["value"]
    # the steady windows: the step with its share of re-bins, whatever --steps was
    steady = d["steady"]
    assert steady["steps"] == 2 * 20 and len(steady["windows"]) == 2 and steady["rebins"] >= 0
    assert steady["min"] <= steady["value"] <= max(steady["windows"])
    assert d["rebin_in_window"] == (d["rebin"]["performed"] > 0)
    assert d["kernels"]["p2g_with_sort"]["window"] == "steady"
    # the line's roofline is the steady one; the timed window's figure rides along
    assert roof["window"].startswith("steady") and d["value_steady"] == steady["value"]
    assert set(roof["first_window"]) == {"kernel", "achieved", "frac", "avg_launch_ms"}
    assert abs(roof["avg_launch_ms"] - sum(steady["kernel_avg_ms"][k] for k in
                                           (("p2g1", "p2g2") if roof["kernel"] == "p2g" else (roof["kernel"],)))) < 2e-4
    assert "frac" not in d["kernels"]["updateGrid"] and "frac" in d["kernels"]["g2p"]
    assert "clearGrid" not in d["kernels"]        # no launch of its own: updateGrid zeroes the older grid set
    assert roof["traffic"] is None and roof["traffic_source"] is None       # this workload was never profiled


@pytest.mark.gpu
def test_the_two_rank_line_over_gloo():
    """Two ranks on the one GPU, gloo transport: everything bench.py does for N > 1 except RCCL."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "bench.py"),
           "--gpus", "2", "--dist-backend", "gloo", "--particles-log2", "18", "--slab-yz", "64",
           "--slab-x-per-gpu", "32", "--steps", "6", "--warmup", "2", "--timing-stride", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    d = _last_json_line(r.stdout)
    assert set(LINE_KEYS) <= set(d)
    assert d["n_gpus"] == 2 and d["scaling"] == "weak" and d["cpu_baseline"] is None
    assert d["config"]["particles"] == 2 << 18 and d["config"]["grid"] == [64, 64, 64]
    assert sum(d["config"]["resident_per_rank_end"]) == 2 << 18
    assert abs(d["value"] - (2 << 18) / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    assert set(ROOFLINE_KEYS) <= set(d["roofline"])
    assert len(d["config"]["kernel_ms_per_step_per_rank"]) == 2
    assert d["config"]["kernel_time_imbalance_max_over_mean"] >= 1.0
    assert len(d["config"]["rebins_per_rank"]) == 2 and min(d["config"]["rebins_per_rank"]) >= 0
    assert len(d["config"]["particles_sent_away_per_step_per_rank"]) == 2


@pytest.mark.gpu
def test_the_plain_two_gpu_command():
    """`python bench.py --gpus 2 ...` exactly as a driver would type it for one GPU: no launcher, no
    WORLD_SIZE.  On a one-GPU box the ranks share the GPU and fall back to the gloo transport."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--particles-log2", "18",
                        "--slab-yz", "64", "--slab-x-per-gpu", "32", "--steps", "6", "--warmup", "2",
                        "--timing-stride", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and set(LINE_KEYS) <= set(d)
    assert sum(d["config"]["resident_per_rank_end"]) == 2 << 18


@pytest.mark.gpu
def test_the_one_slab_line():
    """`--one-slab`: one rank's share of the weak-scaling problem through the multi-GPU code path, no
    process group, exactly one line on stdout."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--one-slab", "--particles-log2", "18",
                        "--slab-yz", "64", "--slab-x-per-gpu", "32", "--steps", "6", "--warmup", "2",
                        "--timing-stride", "2"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    d = json.loads(lines[0])
    assert d["n_gpus"] == 1 and set(LINE_KEYS) <= set(d)
    assert d["config"]["grid"] == [32, 64, 64] and d["config"]["resident_per_rank_end"] == [1 << 18]
    assert d["config"]["parallelism"] == "slab1" and d["config"]["transport"] == "none"
