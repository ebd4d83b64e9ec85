import csv, glob, json


def avg(root, needle, counter):
    vals = []
    for f in glob.glob(root + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if needle in r["Kernel_Name"] and r["Counter_Name"] == counter:
                vals.append(float(r["Counter_Value"]))
    return sum(vals) / len(vals), len(vals)


known = float(1 << 30)
rf, _ = avg("gpurun_out/traffic/cal_f", "k_read_dword", "FETCH_SIZE")
ww, _ = avg("gpurun_out/traffic/cal_w", "k_write_dword", "WRITE_SIZE")
# counters are in KiB; calibration factor = true bytes / reported bytes for this access shape
cal_f = known / (rf * 1024.0)
cal_w = known / (ww * 1024.0)
kf, nf = avg("gpurun_out/traffic/k_f", "k_critic_step_mfma", "FETCH_SIZE")
kw, nw = avg("gpurun_out/traffic/k_w", "k_critic_step_mfma", "WRITE_SIZE")
res = {"k_critic_step_mfma": {
    "fetch_size_kib_raw": kf, "write_size_kib_raw": kw, "launches": nf,
    "calibration": {"fetch_true_over_reported": cal_f, "write_true_over_reported": cal_w,
                    "fetch_raw_kib_for_1GiB": rf, "write_raw_kib_for_1GiB": ww,
                    "shape": "4 B per lane coalesced, 1 GiB stream (scripts/probe/fetch_calib.hip)"},
    "hbm_bytes_per_launch": kf * 1024.0 * cal_f + kw * 1024.0 * cal_w,
    "algorithmic_bytes_per_launch": 24.0 * 65536 * 128,
    "workload": "65,536 envs x T=128 (B = 8,388,608 samples), scripts/critic_only.py"}}
json.dump(res, open("gpurun_out/traffic/r01_traffic.json", "w"), indent=1)
print(json.dumps(res, indent=1))
