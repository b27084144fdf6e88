#!/usr/bin/env python3
"""profiles/r06_pmc_traffic.json from the summaries tools/rocprof_collect.sh wrote: HBM bytes per working launch of the DigitBinningPass and of the
histogram sweep = FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 B: MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, in KiB per dispatch, with the
calibration of the same runs (init_random writes 2^30 B; the histogram sweep reads 2^30 B) and the hash of the kernel sources they were measured on
(bench.py hands the numbers on only for that source state).  usage: tools/make_pmc_json.py IN_DIR COMMIT"""
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def counters(path):
    """kernel name -> mean over the working dispatches (last column of rocprof_summary's PMC table)"""
    out = {}
    for ln in open(path):
        m = re.match(r"(.+?)\s*\|\s*(FETCH_SIZE|WRITE_SIZE)\s*\|\s*(\d+)\s*\|\s*([0-9.]+)\s*\|\s*(\d+)\s*\|\s*([0-9.]+)", ln)
        if m:
            out[m.group(1).strip()] = (float(m.group(6)), int(m.group(5)))
    return out


def pick(tab, *needles):
    for k, v in tab.items():
        if all(n in k for n in needles):
            return k, v
    return None, (0.0, 0)


def main():
    d, commit = sys.argv[1], sys.argv[2]
    res = {"round": 6, "commit": commit, "kernel_sources_sha256": bench.kernel_sources_sha256(), "kernel_sources": list(bench.KERNEL_SOURCES),
           "collected_by": "tools/rocprof_collect.sh: rocprofv3 --kernel-trace --pmc FETCH_SIZE / WRITE_SIZE in separate passes of `python bench.py --steps 3 "
                           "--warmup 1 --no-cpu-baseline --no-more [--pairs N]`, 2^28 elements, summaries under profiles/r06_rocprof/",
           "correction": "FETCH_SIZE x 2 on gfx950 (128-byte requests tallied at 64 B); KiB per dispatch; mean over the WORKING dispatches of a kernel "
                         "(a sort also enqueues launches that exit on their flag word)"}
    for cfg, vb in (("keys", 0), ("pairs4", 4), ("pairs8", 8)):
        f, w = counters(os.path.join(d, f"{cfg}_fetch.txt")), counters(os.path.join(d, f"{cfg}_write.txt"))
        n = 1 << 28
        entry = {}
        passk = ("digit_binning_dual_kernel<0, false>",) if not vb else ("digit_binning_persist_kernel",)
        for name, needles, alg in (("pass", passk, (8 + 2 * vb) * n), ("histogram", ("hy_histogram_kernel",), 4 * n),
                                   ("local_sort", ("hy_local_sort",), (8 + 2 * vb) * n)):
            kf, (fv, fc) = pick(f, *needles)
            kw, (wv, wc) = pick(w, *needles)
            if kf is None:
                continue
            traffic = int(fv * 2048 + wv * 1024)
            entry[name] = {"kernel": kf, "fetch_bytes": int(fv * 2048), "write_bytes": int(wv * 1024), "FETCH_SIZE_KiB_per_launch": fv,
                           "WRITE_SIZE_KiB_per_launch": wv, "working_dispatches": [fc, wc], "traffic_bytes_per_launch": traffic,
                           "algorithmic_bytes_per_launch": alg, "ratio": round(traffic / alg, 4)}
        _, (gen_w, _) = pick(w, "init_random_kernel")
        entry["calibration"] = {"init_random_WRITE_SIZE_KiB": gen_w, "expected_KiB": (4 + vb) * n / 1024,
                                "histogram_FETCH_SIZE_KiB": entry.get("histogram", {}).get("FETCH_SIZE_KiB_per_launch"), "expected_half_KiB": n * 4 / 2048}
        res[cfg] = entry
    json.dump(res, open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles", "r06_pmc_traffic.json"), "w"), indent=1)
    print(json.dumps({k: {kk: vv.get("ratio") for kk, vv in v.items() if isinstance(vv, dict) and "ratio" in vv} for k, v in res.items() if isinstance(v, dict)}))


if __name__ == "__main__":
    main()
