"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes into per-kernel HBM bytes per launch.
argv: <fetch_dir> <write_dir> <cal_fetch_dir> <cal_write_dir>.  Prints JSON."""
import collections, csv, glob, json, sys

GIB = float(1 << 30)


def per_kernel(d, counter):
    agg, cnt = collections.defaultdict(float), collections.Counter()
    for fn in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"] != counter:
                continue
            k = r["Kernel_Name"]
            agg[k] += float(r["Counter_Value"]); cnt[k] += 1
    return {k: (agg[k] / cnt[k], cnt[k]) for k in agg}


def short(k):
    k = k.replace("void ", "").replace("emer::", "")
    return k.split("(")[0][:90]


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
cal_f, cal_w = per_kernel(sys.argv[3], "FETCH_SIZE"), per_kernel(sys.argv[4], "WRITE_SIZE")
# rocprofv3 reports FETCH_SIZE / WRITE_SIZE in KiB
cal = {}
for k, (v, n) in cal_f.items():
    if "layout_transpose" in k or "elementwise" in k or "copy" in k.lower():
        cal[short(k)] = {"launches": n, "known_read_GiB": 1.0, "FETCH_SIZE_GiB": v * 1024 / GIB}
for k, (v, n) in cal_w.items():
    if short(k) in cal:
        cal[short(k)].update({"known_write_GiB": 1.0, "WRITE_SIZE_GiB": v * 1024 / GIB})
tr = [c for k, c in cal.items() if "layout_transpose" in k]
rf = (1.0 / tr[0]["FETCH_SIZE_GiB"]) if tr and tr[0].get("FETCH_SIZE_GiB") else 2.0
wf = (1.0 / tr[0]["WRITE_SIZE_GiB"]) if tr and tr[0].get("WRITE_SIZE_GiB") else 1.0
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import source_hash  # noqa: E402  (what the summary is valid for; bench.py refuses a summary of other sources)
out = {"unit": "bytes per launch", "source_sha16": source_hash(), "read_correction": rf, "write_correction": wf,
       "correction_note": "factors = known bytes / counter on tools/pmc_calibrate.py's layout_transpose dispatch (same 8 B/lane "
                          "access width as the grid kernels); MI355X_MICROARCH.md documents the x2 on reads for gfx950",
       "calibration": cal, "kernels": {}}
for k in sorted(set(fetch) | set(write)):
    if "emer" not in k:
        continue
    f, nf = fetch.get(k, (0.0, 0)); w, nw = write.get(k, (0.0, 0))
    out["kernels"][short(k)] = {"launches": max(nf, nw), "fetch_raw": f * 1024, "write_raw": w * 1024,
                                "hbm_bytes": f * 1024 * rf + w * 1024 * wf}
print(json.dumps(out, indent=1))
