"""Average kernel durations of a rocprofv3 --kernel-trace csv of `bench.py`, split by phase: all launches (what
--stats prints), the timed region (launches warmup .. warmup+steps-1, what bench.py's HIP events cover) and the
solo launches after it. Usage: trace_phase_avg.py <kernel_trace.csv> <warmup> <steps>"""
import collections, csv, json, sys
rows = list(csv.DictReader(open(sys.argv[1]))); W, K = int(sys.argv[2]), int(sys.argv[3])
d = collections.defaultdict(list)
for r in rows:
    d[r["Kernel_Name"].split("(")[0].replace("void ", "").replace("rpw::", "").split("<")[0]].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
out = {"warmup": W, "steps": K}
for k, v in d.items():
    if "kernel" not in k or "rocclr" in k or "native" in k:
        continue
    v.sort(); dur = [(e - s) / 1e6 for s, e in v]
    out[k] = {"launches": len(dur), "avg_ms_all": round(sum(dur) / len(dur), 4),
              "avg_ms_timed_region": round(sum(dur[W:W + K]) / max(1, len(dur[W:W + K])), 4),
              "avg_ms_solo_after": round(sum(dur[W + K:]) / max(1, len(dur[W + K:])), 4)}
print(json.dumps(out, indent=1))
