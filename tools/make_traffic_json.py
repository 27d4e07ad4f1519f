#!/usr/bin/env python3
"""profiles/latest_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over `bench.py --no-secondary --no-cpu-baseline`:
HBM bytes per launch of the evaluation pass's kernels, keyed as bench.py keys its families, with the SHA-256 of the kernel sources the profile was taken from
(bench.py reports roofline.traffic only while that hash matches the tree).
usage: make_traffic_json.py FETCH_DIR WRITE_DIR SOURCE_LABEL [SQ_DIR SQ_LABEL] > profiles/latest_traffic.json
SQ_DIR (optional): the second SQ counter pass of tools/pmc_sq.sh (SQ_INSTS_MFMA, SQ_INSTS_VALU, SQ_VALU_MFMA_BUSY_CYCLES per dispatch) -> "sq": what the kernels EXECUTE
(bench.py roofline.executed).
Units (MI355X_MICROARCH.md, HBM / rocprofv3 section): FETCH_SIZE and WRITE_SIZE count kilobytes (x 1024); FETCH_SIZE is NOT doubled here — these kernels read 8-byte
strided values, not the 16 B/lane streams the guide's gfx950 x2 correction was calibrated on."""
import csv, glob, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import kernel_sources_sha   # noqa: E402

KEYS = {"k_imu_own": "imu", "SurfAccT<false>": "surfel", "k_reproj_cross": "reproj_cross", "k_reproj_jac": "reproj_jac", "RepSideAcc<1": "reproj_obs", "RepSideAcc<0": "reproj_ref",
        "k_reproj_lmrows": "reproj_lmrows", "k_clear": "clear", "k_reproj_asm": "reproj_asm"}


def mean_per_kernel(d, counter):
    acc = {}
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for pat, key in KEYS.items():
                if pat in r["Kernel_Name"]:
                    acc.setdefault(key, []).append(float(r["Counter_Value"]))
    # dispatches without normal equations (cost-only passes) move far less: keep the upper half
    out = {}
    for k, v in acc.items():
        v = sorted(v)[len(v) // 2:]
        out[k] = 1024.0 * sum(v) / len(v)
    return out


fetch, write = mean_per_kernel(sys.argv[1], "FETCH_SIZE"), mean_per_kernel(sys.argv[2], "WRITE_SIZE")
kern = {k: {"fetch_bytes": fetch.get(k, 0.0), "write_bytes": write.get(k, 0.0)} for k in sorted(set(fetch) | set(write))}
kern["reproj"] = {"fetch_bytes": sum(v["fetch_bytes"] for k, v in kern.items() if k.startswith("reproj_")), "write_bytes": sum(v["write_bytes"] for k, v in kern.items() if k.startswith("reproj_"))}
doc = {"sources_sha256": kernel_sources_sha(), "source": sys.argv[3], "kernels": kern}
if len(sys.argv) > 5:
    cnt = {c: mean_per_kernel(sys.argv[4], c) for c in ("SQ_INSTS_MFMA", "SQ_INSTS_VALU", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS")}
    keys = sorted(set().union(*[set(v) for v in cnt.values()]))
    # mean_per_kernel scales by 1024 (the HBM counters are in KiB); instruction counters are plain counts
    doc["sq"] = {k: {"insts_mfma": cnt["SQ_INSTS_MFMA"].get(k, 0.0) / 1024.0, "insts_valu": cnt["SQ_INSTS_VALU"].get(k, 0.0) / 1024.0,
                     "mfma_busy_cycles": cnt["SQ_VALU_MFMA_BUSY_CYCLES"].get(k, 0.0) / 1024.0, "insts_lds": cnt["SQ_INSTS_LDS"].get(k, 0.0) / 1024.0} for k in keys}
    doc["sq_source"] = sys.argv[5]
print(json.dumps(doc, indent=1))
