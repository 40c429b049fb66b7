"""Which tasks of the zoo (tests/ref_env_zoo.py: the 45 registered tasks that need no download) may have their control step replayed as a HIP graph?  For each id
the verdict of maniskill_amd.fused_step.accelerate(env, graph="watch") on the CPU checker (tests/ref_fused_step.py graph_safe:<id>): the level it reaches (task
plugin / fused controller + the reference's own step / nothing), and what the watch of two consecutive steps found (waits, state handed over through fresh tensors,
host data).  No GPU needed.    python tools/zoo_graph_survey.py [workers=3] [ids...]   -> profiles/r06_zoo_graph_survey.log on stdout"""
import json, os, subprocess, sys
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ref_env_zoo


def one(eid):
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "ref_fused_step.py"), "oracle", "graph_safe:" + eid, "3"], cwd=os.path.join(ROOT, "tests"),
                       capture_output=True, text=True, timeout=1800)
    line = [ln for ln in r.stdout.splitlines() if ln.startswith("FUSED ")]
    if r.returncode != 0 or not line:
        tail = (r.stderr.strip().splitlines() or ["?"])[-1]
        return eid, dict(level="error", error=tail[:300])
    return eid, json.loads(line[-1][6:])


def main():
    workers = int(sys.argv[1]) if len(sys.argv) > 1 else 3
    ids = sys.argv[2:] or ref_env_zoo.ENV_IDS
    ok = 0
    with ThreadPoolExecutor(workers) as ex:
        for eid, res in ex.map(one, ids):
            clean = res.get("level") not in ("error", None) and not res.get("sync") and not res.get("flow")      # (host_data: served from the device, listed only)
            ok += clean
            why = "" if clean else "  <- " + (res.get("error") or "; ".join(f"{k}: {str(res.get(k))[:160]}" for k in ("sync", "flow") if res.get(k)))
            print(f"{eid:36s} {'graph' if clean else 'eager':5s} level {res.get('level')}{why}", flush=True)
    print(f"{ok} of {len(ids)} tasks pass the watch")


if __name__ == "__main__":
    main()
