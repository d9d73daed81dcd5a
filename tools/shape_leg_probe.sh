# the tutorial-shape leg of bench.py in shortened bench runs: does the slow mode (1.2 instead of 0.45 ms) follow the legs before it?
show() { python - "$1" <<'PY'
import json, sys
d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
t = d["shapes"]["tutorial_61x101x101_k5"]
print(sys.argv[1], "tutorial resident %.3f ms, numpy in %.3f ms; C2 serial %.3f" % (t["latency_ms_per_call"], t["latency_ms_numpy_in"], d["latency_ms_per_call"]))
PY
}
python bench.py --steps 20 --sustained-seconds 0 --no-cpu-baseline --no-numpy-in --no-strong > /tmp/a.json 2>/tmp/a.err; show /tmp/a.json
python bench.py --steps 20 --sustained-seconds 3 --no-cpu-baseline --no-numpy-in --no-strong > /tmp/b.json 2>/tmp/b.err; show /tmp/b.json
python bench.py --steps 200 --sustained-seconds 10 --no-cpu-baseline --no-numpy-in --no-strong > /tmp/c.json 2>/tmp/c.err; show /tmp/c.json
python tools/sync_probe.py 2>&1 | grep "61x101" | head -3
