"""Per-kernel averages of the rocprofv3 PMC passes written by scripts/pmc_passes.sh.
usage: pmc_summary.py gpurun_out/pmc/<tag>  -> prints a table and writes <dir>/summary.json

Units (MI355X_MICROARCH.md): SQ_WAVE_CYCLES / SQ_WAIT_* / SQ_ACTIVE_INST_* count quad-cycles summed over waves;
SQ_VALU_MFMA_BUSY_CYCLES and SQ_BUSY_CU_CYCLES count cycles; FETCH_SIZE / WRITE_SIZE are KiB, FETCH_SIZE reports half
of the bytes of a coalesced streaming read on gfx950 (doubled here; the 4-B-per-lane shape is checked with
scripts/probe/fetch_calib when gpurun_out/pmc/calib exists)."""
import collections
import csv
import glob
import json
import os
import sys

root = sys.argv[1].rstrip("/")


def load(passdir):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    meta = {}
    for f in glob.glob(os.path.join(passdir, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            meta[k] = {"workgroup_size": int(r["Workgroup_Size"]), "lds_bytes": int(r["LDS_Block_Size"]),
                       "vgpr": int(r["VGPR_Count"]), "agpr": int(r["Accum_VGPR_Count"]), "sgpr": int(r["SGPR_Count"]),
                       "scratch": int(r["Scratch_Size"]), "grid": int(r["Grid_Size"])}
    return agg, meta


def durations(passdir):
    d = collections.defaultdict(list)
    for f in glob.glob(os.path.join(passdir, "**", "*kernel_trace.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            d[r["Kernel_Name"]].append(float(r["End_Timestamp"]) - float(r["Start_Timestamp"]))
    return d


def calib():
    out = {"fetch_scale": 2.0, "write_scale": 1.0, "source": "guide: FETCH_SIZE x2 on gfx950, WRITE_SIZE exact"}
    c = os.path.join(os.path.dirname(root), "calib")
    try:
        fa, _ = load(os.path.join(c, "fetch"))
        wa, _ = load(os.path.join(c, "write"))
        rf = [v for k, cs in fa.items() if "k_read_dword" in k for v in cs["FETCH_SIZE"]]
        ww = [v for k, cs in wa.items() if "k_write_dword" in k for v in cs["WRITE_SIZE"]]
        known = float(1 << 30)
        out = {"fetch_scale": known / (sum(rf) / len(rf) * 1024.0), "write_scale": known / (sum(ww) / len(ww) * 1024.0),
               "source": "scripts/probe/fetch_calib: 1 GiB coalesced 4-B-per-lane read / write in this run"}
    except Exception:
        pass
    return out


cal = calib()
passes = {p: load(os.path.join(root, p)) for p in ("sq_a", "sq_b", "fetch", "write")}
dur = durations(os.path.join(root, "sq_a"))
kernels = sorted({k for agg, _ in passes.values() for k in agg})
import hashlib
_lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "relearn_amd", "librelearn_hip.so")
summary = {"calibration": cal, "kernels": {},
           # the build these counters belong to: bench.py reports them only as applying when it runs the same library
           "lib_sha16": hashlib.sha256(open(_lib, "rb").read()).hexdigest()[:16] if os.path.exists(_lib) else None}
for k in kernels:
    if k.startswith("__amd") or "fillBuffer" in k:
        continue
    row = {}
    for p, (agg, meta) in passes.items():
        for c, vals in agg.get(k, {}).items():
            row[c] = sum(vals) / len(vals)
            row["launches"] = len(vals)
            if c in ("FETCH_SIZE", "WRITE_SIZE") and len(vals) > 1:
                # the first launch of a kernel writes memory nobody has touched yet (round 3's rollout figure, 1.39 x
                # its record, was such a launch): the traffic figures are those of the later launches
                row[c + "_first_launch"] = vals[0]
                row[c] = sum(vals[1:]) / (len(vals) - 1)
        if k in meta:
            row.update(meta[k])
    if k in dur:
        row["avg_duration_us_under_pmc"] = sum(dur[k]) / len(dur[k]) / 1e3
    wc = row.get("SQ_WAVE_CYCLES")
    if wc:
        for name, c in (("valu_active_frac_of_wave_cycles", "SQ_ACTIVE_INST_VALU"),
                        ("lds_active_frac_of_wave_cycles", "SQ_ACTIVE_INST_LDS"),
                        ("any_active_frac_of_wave_cycles", "SQ_ACTIVE_INST_ANY"),
                        ("wait_any_frac_of_wave_cycles", "SQ_WAIT_ANY"),
                        ("wait_inst_any_frac_of_wave_cycles", "SQ_WAIT_INST_ANY")):
            if c in row:
                row[name] = row[c] / wc
    if "SQ_VALU_MFMA_BUSY_CYCLES" in row and row.get("SQ_BUSY_CU_CYCLES"):
        # matrix-pipe busy cycles summed over SIMDs against CU-busy cycles summed over CUs x 4 SIMDs
        row["mfma_busy_frac"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (4.0 * row["SQ_BUSY_CU_CYCLES"])
    if "SQ_ACTIVE_INST_VALU" in row and row.get("SQ_BUSY_CU_CYCLES"):
        # VALU-issuing quad-cycles (x4 = cycles) summed over waves against the SIMD cycles available
        row["valu_busy_frac"] = 4.0 * row["SQ_ACTIVE_INST_VALU"] / (4.0 * row["SQ_BUSY_CU_CYCLES"])
    if "FETCH_SIZE" in row:
        row["fetch_bytes"] = row["FETCH_SIZE"] * 1024.0 * cal["fetch_scale"]
    if "WRITE_SIZE" in row:
        row["write_bytes"] = row["WRITE_SIZE"] * 1024.0 * cal["write_scale"]
    if "fetch_bytes" in row and "write_bytes" in row:
        row["hbm_bytes_per_launch"] = row["fetch_bytes"] + row["write_bytes"]
    summary["kernels"][k] = row
json.dump(summary, open(os.path.join(root, "summary.json"), "w"), indent=1)
print("calibration:", cal)
for k, row in summary["kernels"].items():
    print(k[:100])
    for c in sorted(row):
        v = row[c]
        print("    %-36s %s" % (c, ("%.4g" % v) if isinstance(v, float) else v))
