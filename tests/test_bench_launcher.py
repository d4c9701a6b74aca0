"""bench.py --gpus N without a launcher starts its ranks as child processes; the parent must not have touched a GPU
(a re-exec from a GPU-initialised process takes this pool's machines down): it neither imports torch nor maps the HIP
runtime before the spawn."""
import json
import os
import subprocess
import sys

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

PROBE = r"""
import sys, json
sys.argv = ["bench.py", "--gpus", "2"]
sys.path.insert(0, %r)
import bench
n = bench.count_gpus_sysfs()
cores = bench.effective_cores()
maps = open("/proc/self/maps").read()
print(json.dumps({"gpus": n, "cores": cores, "torch": "torch" in sys.modules,
                  "hip": ("libamdhip64" in maps) or ("libhsa-runtime" in maps)}))
"""


def test_the_launcher_parent_stays_off_the_gpu():
    out = subprocess.run([sys.executable, "-c", PROBE % REPO], capture_output=True, text=True, timeout=120, cwd=REPO)
    assert out.returncode == 0, out.stderr[-2000:]
    res = json.loads(out.stdout.strip().splitlines()[-1])
    assert res["torch"] is False and res["hip"] is False, res
    assert res["gpus"] is None or res["gpus"] >= 0
    assert res["cores"] >= 1


def test_spawn_refuses_more_ranks_than_gpus_without_touching_torch(monkeypatch):
    """with a readable topology that holds fewer GPUs than asked for, the parent answers with an error line and starts
    nothing; torch stays unimported"""
    sys.path.insert(0, REPO)
    import bench
    monkeypatch.setattr(bench, "count_gpus_sysfs", lambda: 1)
    started = []
    monkeypatch.setattr(bench.subprocess, "call", lambda *a, **k: started.append(a) or 0)

    class A:
        gpus = 4
    was = "torch" in sys.modules
    rc = bench.spawn_ranks(A())
    assert rc == 2 and not started
    assert ("torch" in sys.modules) == was
