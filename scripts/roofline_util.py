"""Shared by the per-configuration measurement scripts: counter evidence from a committed PMC summary (profiles/), used
only when it was collected from the library that is running, and the `roofline` object of the bench contract."""
import hashlib
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBS = 8000.0      # MI355X_MICROARCH.md: HBM3E 8 TB/s
BF16_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
F32_PEAK_TFLOPS = 157.3    # f32 vector = f32 MFMA


def lib_sha16():
    import relearn_amd as ra
    return hashlib.sha256(open(ra.LIB_PATH, "rb").read()).hexdigest()[:16]


def pmc_kernels(summary_file):
    """{kernel name: row} of profiles/<summary_file> with `applies` = collected from this very library"""
    path = os.path.join(ROOT, "profiles", summary_file)
    src = {"file": "profiles/" + summary_file, "kind": "committed rocprofv3 PMC summary, not measured in this run",
           "applies": False}
    if not os.path.exists(path):
        src["why_not"] = "no such summary"
        return {}, src
    summary = json.load(open(path))
    src["collected_from_lib_sha16"] = summary.get("lib_sha16")
    src["this_lib_sha16"] = lib_sha16()
    src["applies"] = summary.get("lib_sha16") == src["this_lib_sha16"]
    return summary.get("kernels", {}), src


def traffic_of(kernels, prefixes):
    """HBM bytes per launch (FETCH_SIZE + WRITE_SIZE, corrected as the guide prescribes: scripts/pmc_summary.py) summed
    over ONE launch of each kernel whose name starts with one of `prefixes`, weighted by `weights` launches"""
    total, found = 0.0, []
    for name, row in kernels.items():
        short = name.replace("void ", "").replace("(anonymous namespace)::", "")
        for p, w in prefixes.items():
            if short.startswith(p) and row.get("hbm_bytes_per_launch") is not None:
                total += w * row["hbm_bytes_per_launch"]
                found.append(p)
    return (total if found else None), sorted(set(found))
