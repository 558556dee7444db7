"""A/B timing of builds of the library on one workload: tools/ab.py WORKLOAD LIB [LIB ...]
Runs tools/run_workload.py for each library in turn, several rounds, and prints per library the minimum and
the median launch time over all rounds (first two launches of every run dropped: clocks ramp up)."""
import os, re, statistics, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1]
libs = sys.argv[2:]
rounds, launches = int(os.environ.get("AB_ROUNDS", "3")), int(os.environ.get("AB_LAUNCHES", "10"))
variant = os.environ.get("AB_VARIANT", "4" if wl.startswith("c5") else "2")
times = {l: [] for l in libs}
for _ in range(rounds):
    for l in libs:
        env = dict(os.environ, RTMI_LIBRARY=os.path.join(ROOT, l))
        out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "run_workload.py"), wl, variant, str(launches)],
                             env=env, capture_output=True, text=True, timeout=280)
        if out.returncode != 0:
            print(l, "FAILED", out.stderr[-500:])
            sys.exit(1)
        ms = [float(v) for v in re.findall(r"'([0-9.]+)'", out.stdout)]
        times[l] += ms[2:]
for l in libs:
    t = times[l]
    print("%-40s min %.3f  median %.3f  (n=%d)" % (l, min(t), statistics.median(t), len(t)), flush=True)
