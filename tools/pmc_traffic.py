"""Turn two rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE, collected separately as MI355X_MICROARCH.md prescribes) of
`bench.py --steps 1 --warmup 1 --no-roofline --no-cpu-baseline` (= 2 training steps) into HBM bytes per step of the three
convolution operations.  A group = every kernel the C-ABI call launches: the implicit GEMM of that layout, and for
backward-weight also the direct small-channel kernel, the slab reductions and the bias column sums.
gfx950 correction: FETCH_SIZE counts 128-B requests as 64 B for wide (16 B/lane) streams -> doubled; WRITE_SIZE is exact
(checked on a conv whose output size is known: 130560 KB reported = written).  bench.py divides by its own count of calls.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <model> [steps=2] > profiles/r01_pmc_traffic_<model>.json
"""
import collections, csv, glob, json, re, sys

fetch_dir, write_dir, model = sys.argv[1:4]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
LAY = {"0": "fwd", "1": "dgrad", "2": "wgrad"}


def group(name):
    m = re.search(r"igemm_f32_kernel<(\d)", name)
    if m:
        return LAY[m.group(1)]
    if "hf_fwd_kernel" in name or "hf_bwd_kernel" in name or "hf_reduce_kernel" in name:
        return "head"           # the class heads fused with the BatchNorm in front of them (round 6, csrc/headfuse.h): forward, both backward passes
    if "p1t_kernel" in name or "p1t_reduce" in name:
        return "wgrad_p1"       # pointwise backward-weight + its slab sum (round 5, csrc/pconv1.hip)
    if "p1_kernel" in name:
        return "p1"             # pointwise forward AND backward-data
    if "p1_prep" in name or "p1_amax" in name:
        return "d3_prep"
    if "dwgrad3_pl" in name:
        return "wgrad_d3p"      # backward-weight on producer-written planes + its slab reduction (round 4)
    if "dconv3_pl_kernel" in name:
        return "d3p"            # forward AND backward-data on producer-written planes (round 4)
    if "planes_from_f32" in name:
        return "d3_prep"
    if "dwgrad3_h2_" in name:
        return "wgrad_d3h"
    if "dconv3_h2_kernel" in name or "dconv3_h2_spec_kernel" in name:
        return "d3h"            # f16x2 direct kernels, forward AND backward-data
    if "dconv3_h2_" in name:
        return "d3_prep"
    if "igemm_h2t_kernel" in name or "h2_reduce_slabs" in name:
        return "wgrad_h2"
    if "igemm_h2w_kernel" in name or "igemm_h2w8_kernel" in name:
        return "h2w"            # f16x2 forward AND backward-data (round 6: igemm_h2w8_kernel, two waves per SIMD)
    if "split2h" in name or "amax_kernel" in name:
        return "split3"
    if "igemm_b3w_kernel" in name or "igemm_b3_kernel" in name:
        return "b3w"            # bf16x3 forward AND backward-data (one kernel serves both)
    if "igemm_b3t_kernel" in name or "b3_reduce_slabs" in name:
        return "wgrad_b3"
    if "dconv3_b3" in name:
        return "d3"             # direct 3x3 forward / backward-data of the HRNet trunk (dconv3_b3_kernel, dconv3_b3_spec_kernel)
    if "dwgrad3" in name:
        return "wgrad_d3"       # direct backward-weight + its slab reduction
    if "dconv3_prep" in name:
        return "d3_prep"
    if "split3" in name:
        return "split3"
    if "wgrad_direct_kernel" in name or "reduce_slabs" in name or "colsum_" in name:
        return "wgrad"
    return None


out = {"model": model, "steps_profiled": steps,
       "unit": "bytes per training step per operation group (HBM/fabric side, FETCH_SIZE x2 + WRITE_SIZE)", "kernels": {}}
acc = collections.defaultdict(lambda: {"kernel_launches": 0, "fetch_kb": 0.0, "write_kb": 0.0})
for name, d in (("FETCH_SIZE", fetch_dir), ("WRITE_SIZE", write_dir)):
    f = glob.glob(d + "/*/*counter_collection.csv")[0]
    n = collections.Counter()
    for r in csv.DictReader(open(f)):
        k = group(r["Kernel_Name"])
        if r["Counter_Name"] == name and k:
            acc[k]["fetch_kb" if name == "FETCH_SIZE" else "write_kb"] += float(r["Counter_Value"])
            n[k] += 1
    for k, v in n.items():
        acc[k]["kernel_launches"] = v
for k, v in acc.items():
    out["kernels"][k] = {"kernel_launches_per_step": v["kernel_launches"] / steps,
                         "fetch_size_kb_raw_per_step": v["fetch_kb"] / steps, "write_size_kb_per_step": v["write_kb"] / steps,
                         "hbm_bytes_per_step": (2 * v["fetch_kb"] + v["write_kb"]) * 1024 / steps}
print(json.dumps(out, indent=1))
