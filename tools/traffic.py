"""HBM-side bytes per launch of a workload, from the L2's memory-side counters: one rocprofv3 --pmc pass for
FETCH_SIZE and one for WRITE_SIZE (they do not fit one pass), per-kernel means over the launches, written to
profiles/traffic_<workload>.json (what bench.py reports as roofline.traffic).

    python tools/traffic.py WORKLOAD [--lib LIB] [--variant V] [--frames N] [--tag TAG]

FETCH_SIZE is in KiB and counts 64 bytes per request; on gfx950 a wide coalesced read is one 128-byte request, so
the figure is doubled for the streaming kernels (an upper bound where reads are narrower); for the hierarchy walk
(scattered 16-byte gathers, 64-byte requests) it is taken as it is.  Infinity-Cache hits are included either way.
This process never touches the GPU; the profiled program is tools/run_workload.py."""
import argparse
import csv
import glob
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def one_pass(counter, out_dir, lib, workload, variant, frames):
    env = dict(os.environ, TMPDIR="/tmp", RTMI_LIBRARY=lib)
    cmd = ["timeout", "-s", "KILL", "170", "rocprofv3", "--pmc", counter, "-d", out_dir, "-o", counter.lower(),
           "--output-format", "csv", "--", "python3", os.path.join(ROOT, "tools", "run_workload.py"), workload, str(variant), "3",
           str(frames)]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True)
    if r.returncode != 0:
        sys.exit("rocprofv3 %s failed: %s" % (counter, r.stderr[-800:]))
    per_kernel = {}
    for f in glob.glob(out_dir + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                per_kernel.setdefault(row["Kernel_Name"], []).append(float(row["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in per_kernel.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("workload")
    ap.add_argument("--lib", default=os.path.join(ROOT, "raytracer_amd", "librtmi.so"))
    ap.add_argument("--variant", type=int, default=None)
    ap.add_argument("--frames", type=int, default=None)
    ap.add_argument("--tag", default="")
    a = ap.parse_args()
    variant = a.variant if a.variant is not None else (4 if a.workload.startswith("c5") else 2)
    frames = a.frames or {"c4_mesh_4k": 512, "c1_plumbing": 64, "c5_stress": 16}.get(a.workload, 256)
    out = os.path.join(ROOT, "gpurun_out", "traffic_%s%s" % (a.workload, a.tag))
    fetch = one_pass("FETCH_SIZE", out + "/fetch", a.lib, a.workload, variant, frames)
    write = one_pass("WRITE_SIZE", out + "/write", a.lib, a.workload, variant, frames)
    streaming = variant != 4
    kernels = {}
    total = 0.0
    for name in sorted(set(fetch) | set(write)):
        m = re.search(r"\b(k_[a-z_0-9]+)", name)
        if not m:
            continue    # fills and copies of the runtime
        short = m.group(1)
        f_kib, w_kib = fetch.get(name, 0.0), write.get(name, 0.0)
        kernels.setdefault(short, {"FETCH_SIZE_KiB": 0.0, "WRITE_SIZE_KiB": 0.0})
        kernels[short]["FETCH_SIZE_KiB"] += f_kib
        kernels[short]["WRITE_SIZE_KiB"] += w_kib
        total += (f_kib * (2.0 if streaming else 1.0) + w_kib) * 1024.0
    res = {
        "workload": "%s, %d frames per launch, variant %d" % (a.workload, frames, variant),
        "frames": frames, "variant": variant, "library": os.path.relpath(a.lib, ROOT),
        "source": "tools/traffic.py: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, means over 3 launches",
        "units": "KiB; FETCH_SIZE %s; Infinity-Cache hits included" % (
            "doubled in hbm_bytes_per_launch (gfx950 tallies a 128-byte request as 64 bytes; an upper bound for narrower reads)"
            if streaming else "taken as it is (scattered 16-byte gathers are 64-byte requests)"),
        "kernels": kernels, "hbm_bytes_per_launch": int(total),
    }
    path = os.path.join(ROOT, "profiles", "traffic_%s%s.json" % (a.workload, a.tag))
    with open(path, "w") as f:
        json.dump(res, f, indent=2)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
