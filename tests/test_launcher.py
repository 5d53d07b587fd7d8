"""`python bench.py --gpus N` must start N ranks by itself and must never print an N = 1 line under --gpus N.
CPU only: the launcher is exercised with a stand-in rank script; bench.py's own guard runs before anything touches the GPU."""
import json
import os
import subprocess
import sys
import textwrap

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_run_ranks_starts_n_processes_and_relays_rank0_json(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text(textwrap.dedent("""
        import json, os, sys
        import torch.distributed as dist
        dist.init_process_group("gloo")
        w = [None] * dist.get_world_size()
        dist.all_gather_object(w, (int(os.environ["RANK"]), int(os.environ["LOCAL_RANK"])))
        print("noise from rank %d" % dist.get_rank())
        if dist.get_rank() == 0:
            print(json.dumps({"n_gpus": dist.get_world_size(), "ranks": w, "argv": sys.argv[1:]}))
        dist.destroy_process_group()
    """))
    code = "import sys; sys.path.insert(0, %r); from devicekmc_amd import launch; sys.exit(launch.run_ranks(%r, ['--x', '1'], 2, timeout=240))" % (ROOT, str(script))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    out = json.loads(lines[-1])                       # the JSON line is the LAST line of stdout
    assert len(lines) == 1                            # ... and the only one: everything else went to stderr
    assert out["n_gpus"] == 2 and sorted(map(tuple, out["ranks"])) == [(0, 0), (1, 1)] and out["argv"] == ["--x", "1"]
    assert "noise from rank" in r.stderr


def test_run_ranks_reports_a_failed_rank(tmp_path):
    script = tmp_path / "rank.py"
    script.write_text("import os, sys\nsys.exit(7 if os.environ['RANK'] == '1' else 0)\n")
    code = "import sys; sys.path.insert(0, %r); from devicekmc_amd import launch; sys.exit(launch.run_ranks(%r, [], 2, timeout=240))" % (ROOT, str(script))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=300)
    assert r.returncode != 0
    assert json.loads(r.stdout.strip().splitlines()[-1])["value"] is None


def test_bench_refuses_a_world_that_is_not_gpus():
    # --gpus 2 inside a one-rank environment: error JSON + non-zero exit, before the GPU is touched (so this runs on CPU)
    env = dict(os.environ, RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2"], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0
    out = json.loads(r.stdout.strip().splitlines()[-1])
    assert out["value"] is None and "WORLD_SIZE=1" in out["error"]
    # and the reverse: a 2-rank environment under the default --gpus 1
    env = dict(os.environ, RANK="0", WORLD_SIZE="2", LOCAL_RANK="0")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")], capture_output=True, text=True, timeout=120, env=env)
    assert r.returncode != 0 and json.loads(r.stdout.strip().splitlines()[-1])["value"] is None


def test_run_ranks_timeout_ends_the_whole_rank_group(tmp_path):
    """A rank group that outlives its time-out is ended as a GROUP (the ranks are grandchildren of the launcher's child): no orphan is left
    holding the device while the caller reports exit code 3."""
    import time
    script = tmp_path / "rank.py"
    pidfile = tmp_path / "pids"
    script.write_text("import os, time\nopen(%r, 'a').write('%%d\\n' %% os.getpid())\ntime.sleep(600)\n" % str(pidfile))
    code = "import sys; sys.path.insert(0, %r); from devicekmc_amd import launch; sys.exit(launch.run_ranks(%r, [], 2, timeout=20))" % (ROOT, str(script))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, timeout=200)
    assert r.returncode == 3, (r.returncode, r.stderr[-1000:])
    assert json.loads(r.stdout.strip().splitlines()[-1])["value"] is None
    pids = [int(x) for x in pidfile.read_text().split()]
    assert len(pids) == 2
    time.sleep(1.0)
    for pid in pids:
        alive = True
        try:
            os.kill(pid, 0)
        except ProcessLookupError:
            alive = False
        assert not alive, pid
