"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as
MI355X_MICROARCH.md prescribes) of `bench.py --steps 1 --warmup 1` into HBM bytes per launch of the
igemm kernels.  gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) streams ->
doubled; WRITE_SIZE is exact (checked on a conv whose output size is known: 130560 KB reported = written).

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <model> > profiles/r01_pmc_traffic_<model>.json
"""
import collections, csv, glob, json, re, sys

fetch_dir, write_dir, model = sys.argv[1:4]
LAY = {"0": "fwd", "1": "dgrad", "2": "wgrad"}
out = {"model": model, "unit": "bytes per launch (HBM/fabric side, FETCH_SIZE x2 + WRITE_SIZE)", "kernels": {}}
acc = collections.defaultdict(lambda: {"launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == name and "igemm" in r["Kernel_Name"]:
            k = LAY[re.search(r"igemm_f32_kernel<(\d)", r["Kernel_Name"]).group(1)]
            acc[k]["fetch_kb" if name == "FETCH_SIZE" else "write_kb"] += float(r["Counter_Value"])
            n[k] += 1
    for k, v in n.items():
        acc[k]["launches"] = v
for k, v in acc.items():
    out["kernels"][k] = {"launches": v["launches"], "fetch_size_kb_raw_per_launch": v["fetch_kb"] / v["launches"],
                         "write_size_kb_per_launch": v["write_kb"] / v["launches"],
                         "hbm_bytes_per_launch": (2 * v["fetch_kb"] + v["write_kb"]) * 1024 / v["launches"]}
print(json.dumps(out, indent=1))
