"""Per-kernel HBM bytes per launch from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE).
Units (MI355X_MICROARCH.md): counter value x 1024 = bytes... FETCH_SIZE derives from TCC_EA0_RDREQ (64 B tallies);
on gfx950 it reports HALF the bytes of a wide coalesced read, so read bytes are reported both raw and x2."""
import csv, collections, glob, json, os, sys
root = sys.argv[1]
res = collections.defaultdict(dict)
for cname in ("FETCH_SIZE", "WRITE_SIZE"):
    files = glob.glob(os.path.join(root, cname, "**", "*counter_collection.csv"), recursive=True)
    if not files:
        continue
    tot = collections.defaultdict(float); disp = collections.defaultdict(set)
    for r in csv.DictReader(open(files[0])):
        if r["Counter_Name"] != cname:
            continue
        kn = r["Kernel_Name"].split("(")[0].replace("rpw::", "").replace("void ", "").split("<")[0]
        tot[kn] += float(r["Counter_Value"]); disp[kn].add(r["Dispatch_Id"])
    for kn, v in tot.items():
        res[kn][cname + "_per_launch"] = v / max(1, len(disp[kn]))
        res[kn]["launches_" + cname] = len(disp[kn])
out = {}
for kn, d in res.items():
    if "kernel" not in kn:
        continue
    f = d.get("FETCH_SIZE_per_launch", 0.0) * 1024.0
    w = d.get("WRITE_SIZE_per_launch", 0.0) * 1024.0
    out[kn] = {"read_bytes_raw": f, "read_bytes_x2_gfx950": 2 * f, "write_bytes": w,
               "hbm_bytes_per_launch": 2 * f + w, "launches": d.get("launches_FETCH_SIZE")}
try:     # the batch the counters belong to (bench.py only quotes them for the same batch size)
    b = json.loads(open(os.path.join(root, "FETCH_SIZE.bench.json")).read().strip().splitlines()[-1])
    out["_meta"] = {"frames_per_launch": b["config"]["frames_per_gpu_per_step"], "points_per_frame": b["config"]["points_per_frame"],
                    "command": "tools/collect_traffic.sh (bench.py --steps 4 --warmup 1 --regions 1 --no-cpu-baseline --no-host-path)"}
except Exception:
    try:
        b = json.loads(open(os.path.join(root, "FETCH_SIZE.probe.json")).read().strip().splitlines()[-1])
        out["_meta"] = {"frames_per_launch": b["frames"], "points_per_launch": b["points"],
                        "command": "PROBE=1 tools/collect_traffic.sh (tools/binning_probe.py, MAX_ITER=1)"}
    except Exception:
        pass
print(json.dumps(out, indent=1))
