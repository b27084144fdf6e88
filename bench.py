#!/usr/bin/env python3
"""bench.py — BASELINE.json's metric on MI355X: GKeys/s of the uint32 OneSweep sort.

  python bench.py --gpus 1 --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

A "step" is one complete sort of one batch of synthetic keys already resident
in HBM: N=1 sorts BASELINE configs[1] (2^28 uniform uint32 keys, keys-only,
reference protocol GPUSortingCUDA/GPUSortingCUDA.cu:22 — InitRandom seed 10+i,
entropy preset 1); N>1 sorts configs[3]'s shape at 2^28 keys PER GPU (weak
scaling): MSD split + RCCL all-to-all-v + per-GPU OneSweep.  Every step gets
its own freshly generated input buffer (generated before the timed region:
288 GB of HBM holds them all), so no step ever re-sorts sorted data.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel =
digit_binning_kernel, HIP events on the sort's stream) and `cpu_baseline`
(the oracle's std::sort on this box's cores, bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--log2-keys", type=int, default=28, help="keys per GPU = 2^this (default: BASELINE 2^28)")
    ap.add_argument("--pairs", type=int, default=0, choices=(0, 4, 8), help="value bytes (configs[2]/[4]); default keys-only")
    ap.add_argument("--entropy", type=int, default=0, help="ENTROPY_PRESET index 0..4")
    ap.add_argument("--shape", type=str, default="", help="tile shape TxK (tuning)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-more", action="store_true", help="skip the 'more' block (configs[2], configs[4], entropy rows) of the N=1 run")
    ap.add_argument("--more-steps", type=int, default=5, help="timed sorts per entry of the 'more' block")
    ap.add_argument("--dry-backend", type=str, default="", help="rehearsal of the N>1 code path on a one-GPU box: 'gloo' = "
                    "every rank on cuda:0, collectives over gloo (host-staged); never used for reported numbers")
    ap.add_argument("--strong", action="store_true", help="N > 1: strong scaling — 2^log2-keys keys IN TOTAL, split evenly over the GPUs "
                    "(SURVEY.md 8d cfg 4; default: weak scaling, 2^log2-keys per GPU)")
    ap.add_argument("--cpu-log2", type=int, default=28, help="keys of the host-sort baseline sample (default: the workload itself)")
    return ap.parse_args()


def baseline_metric() -> str:
    """BASELINE.json's metric name, verbatim (the file travels with the repo)."""
    try:
        return json.load(open(os.path.join(ROOT, "BASELINE.json")))["metric"]
    except Exception:  # noqa: BLE001
        return "GKeys/s uint32 OneSweep at 1/2/4/8 MI355X; % HBM-read roofline"


def cpu_baseline(log2n: int):
    """The oracle's host std::sort on the SAME generator's keys (bounded sample).  Checker/baseline only."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import oracle_lib
    o = oracle_lib.load()
    n = 1 << log2n
    threads = o.hardware_threads()
    keys = o.init_random(n, 10, 0)
    t0 = time.perf_counter()
    out = o.std_sort_parallel(keys, threads)
    dt = time.perf_counter() - t0
    assert o.validate(out) == 0
    used = 1
    while used * 2 <= threads:
        used *= 2
    n1 = 1 << min(log2n, 26)
    k1 = o.init_random(n1, 10, 0)
    t0 = time.perf_counter()
    o.std_sort(k1)
    dt1 = time.perf_counter() - t0
    return {
        "value": n / dt / 1e9, "unit": "GKeys/s", "cores": used if threads >= 2 else 1, "kind": "port",
        "stated_baseline": "value / cores above = host std::sort on the box's own cores (north_star): chunked std::sort + merge tree on "
                           "`cores` threads; single_thread_std_sort below = one plain std::sort call on one core",
        "sample": f"2^{log2n} uint32 keys (InitRandom seed 10, preset 1), chunked std::sort + merge tree on {used} "
                  f"of {threads} hw threads, {dt:.2f} s",
        "single_thread_std_sort": {"value": n1 / dt1 / 1e9, "unit": "GKeys/s", "sample": f"2^{min(log2n, 26)} keys, {dt1:.2f} s"},
    }


KERNEL_SOURCES = ("gpusorting_amd/csrc/onesweep_kernels.hpp", "gpusorting_amd/csrc/hybrid_kernels.hpp", "gpusorting_amd/csrc/gpusort_capi.hip")


def kernel_sources_sha256() -> str:
    """One hash over the files the kernels and their launches are compiled from: what a committed counter measurement is valid for."""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), "rb") as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic_file():
    """The newest profiles/rNN_pmc_traffic.json (tools/make_pmc_json.py writes one per round)."""
    import glob
    files = sorted(glob.glob(os.path.join(ROOT, "profiles", "r[0-9][0-9]_pmc_traffic.json")))
    return files[-1] if files else None


def pmc_traffic(log2_keys, vb, entropy, shape, tile_keys, kernel="pass"):
    """HBM bytes per launch of `kernel` from the PMC counters (FETCH_SIZE x 2 + WRITE_SIZE, collected in separate rocprofv3 --pmc
    passes of this same command and committed under profiles/) — a BORROWED number: measured by the builder's rocprofv3 runs, not
    by this run.  The file carries the hash of the kernel sources it was collected with (the GPU box has no .git, so a commit id
    could not be checked there); provenance["sources_match"] says whether this tree still hashes to it — a number from another source
    state is handed on as well, flagged stale (round 5 dropped it instead and the driver's record then had no traffic at all).
    Returns (bytes or None, provenance dict)."""
    path = pmc_traffic_file()
    if path is None:
        return None, {"reason": "no profiles/rNN_pmc_traffic.json"}
    rel = os.path.relpath(path, ROOT)
    if log2_keys != 28 or entropy or shape:
        return None, {"reason": "the committed counters are for 2^28 keys, entropy preset 1, the library's own tile shapes"}
    d = json.load(open(path))
    now = kernel_sources_sha256()
    e = d.get("keys" if not vb else f"pairs{vb}", {}).get(kernel)
    if not e:
        return None, {"reason": f"no counters committed for value bytes {vb} / {kernel}"}
    match = d.get("kernel_sources_sha256") == now
    return e["traffic_bytes_per_launch"], {
        "source": f"{rel} ({d.get('collected_by')}; commit {d.get('commit')}); builder-run rocprofv3 --pmc passes of the same command, NOT measured by this run",
        "sources_match": match, "measured_on_kernel_sources": d.get("kernel_sources_sha256"), "this_tree": now,
        "kernel": e.get("kernel"), "fetch_bytes": e.get("fetch_bytes"), "write_bytes": e.get("write_bytes"), "ratio_to_algorithmic": e.get("ratio")}


def box_floor(n):
    """What THIS box streams, measured in this process with the tuning build's calibration kernels (libgpusort_tuning.so; never
    part of the product path): a read-only 16-byte sweep of n keys (the GlobalHistogram's floor) and the DigitBinningPass's own
    access shape without ranking and look-back — wave-striped dword loads of a 16 384-key tile, one LDS round trip, coalesced
    dword stores, SEQUENTIAL output (the pass's floor; steady state = a copy that follows a copy).  floor = read + 4 x copy."""
    from gpusorting_amd import _lib
    try:
        lib = _lib.load_tuning()
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)}
    a = torch.empty(n, dtype=torch.int32, device="cuda")
    b = torch.empty(n, dtype=torch.int32, device="cuda")
    a.random_()
    sp = int(torch.cuda.current_stream().cuda_stream)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(8)]

    def seq():
        ev[0].record()
        lib.gs_debug_copy_floor(a.data_ptr(), b.data_ptr(), n, 0, 3, sp)   # read-only sweep
        ev[1].record()
        src, dst = a, b
        for i in range(5):
            lib.gs_debug_copy_floor(src.data_ptr(), dst.data_ptr(), n, 512, 32, sp)  # tile-shaped copy
            ev[2 + i].record()
            src, dst = dst, src
        torch.cuda.synchronize()
        return [ev[i].elapsed_time(ev[i + 1]) for i in range(6)]

    def best_of(code, reps=4):
        """one calibration kernel of the tuning build (threads = 0, kpt = code), best of `reps`"""
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        best = 1e30
        for _ in range(reps + 1):
            e0.record()
            if lib.gs_debug_copy_floor(a.data_ptr(), b.data_ptr(), n, 0, code, sp) != 0:
                return None
            e1.record()
            torch.cuda.synchronize()
            best = min(best, e0.elapsed_time(e1))
        return best

    seq()
    runs = [seq() for _ in range(4)]
    # what the box streams at BEST (the floor as a fact, against the guide's 6.29 TB/s copy): read with four nt loads
    # in flight, copy as a one-shot grid with four nt loads in flight, copy with four plain loads in flight at one workgroup per CU,
    # hipMemcpyAsync D2D
    best = {"read_x4_nt_2wg_per_cu_ms": best_of(4), "copy_x4_nt_one_shot_ms": best_of(5), "copy_x4_1wg_per_cu_ms": best_of(6),
            "hipMemcpyAsync_d2d_ms": best_of(7)}
    copies = [v for k, v in best.items() if k != "read_x4_nt_2wg_per_cu_ms" and v]
    if copies:
        best["best_copy_GBps"] = 8.0 * n / min(copies) / 1e6
        best["best_copy_vs_guide_6290_GBps"] = best["best_copy_GBps"] / 6290.0
    if best["read_x4_nt_2wg_per_cu_ms"]:
        best["best_read_GBps"] = 4.0 * n / best["read_x4_nt_2wg_per_cu_ms"] / 1e6
    read_ms = min(r[0] for r in runs)
    first_copy = min(r[1] for r in runs)
    steady = sorted(sum(r[3:6]) / 3.0 for r in runs)[len(runs) // 2]
    floor_ms = read_ms + first_copy + 3.0 * steady
    return {
        "read_only_sweep_ms": read_ms, "read_GBps": 4.0 * n / read_ms / 1e6,
        "tile_copy_ms_first_after_read": first_copy, "tile_copy_ms_steady": steady, "tile_copy_GBps_steady": 8.0 * n / steady / 1e6,
        "floor_ms": floor_ms, "floor_GKeys_per_s": n / floor_ms / 1e6,
        "streaming_best": best,
        "how": "libgpusort_tuning.so gs_debug_copy_floor: 16-byte grid-stride read sweep; 512x32 tile copy (wave-striped dword loads, LDS "
               "round trip, coalesced dword stores, sequential output); floor = read + first copy + 3 x steady-state copy, HIP events, "
               "best / median of 4 sequences, this process, this box",
    }


def roofline_block(n, vb, prof, log2_keys, entropy, shape, tile_keys=None, rank_mode=None, floor=None, two_level=False):
    """The dominant kernel (one DigitBinningPass launch) against the HBM peak: algorithmic bytes per launch =
    (4 + 4 key bytes + 2 x value bytes) x n, divided by the launch's average duration from HIP
    events recorded on the sort's own stream.  two_level: the device ran the two-level plan (hybrid_kernels.hpp) — profile slots
    pass0 / pass1 are its two DigitBinningPasses (top byte; byte 2 on 256 chains), pass2 the bucket-local sort (with the exit of
    one idle launch), pass3 the exit of another; the sort then MOVES 28 + 6 x value bytes per key, while the metric's algorithmic
    bytes stay those of the reference's algorithm (36 B/key for uint32 keys), which is how whole_sort.frac_of_8000 is
    defined — both are in the block."""
    bpk_pass = 8 + 2 * vb
    bpk_sort = 4 + 4 * bpk_pass
    passes = ("pass0", "pass1") if two_level else ("pass0", "pass1", "pass2", "pass3")
    pass_ms = sum(prof[p] for p in passes) / len(passes)
    achieved = bpk_pass * n / (pass_ms * 1e-3) / 1e9
    whole = bpk_sort * n / (prof["total"] * 1e-3) / 1e9
    bpk_moved = 4 + 3 * bpk_pass if two_level else bpk_sort
    traffic, traffic_src = pmc_traffic(log2_keys, vb, entropy, shape, tile_keys, "pass")
    hist_traffic, hist_src = pmc_traffic(log2_keys, vb, entropy, shape, tile_keys, "histogram")

    def kern(name, ms, bpk, what):
        gbs = bpk * n / (ms * 1e-3) / 1e9
        return {"kernel": name, "what": what, "algorithmic_bytes": bpk * n, "ms": ms, "achieved_GBs": gbs, "frac": gbs / HBM_PEAK_GBS}

    pass_kernel = "digit_binning_dual_kernel" if not vb else f"digit_binning_persist_kernel<{'1024,16,4' if vb == 4 else '512,32,8'}>"
    if two_level:
        kernels = [kern("hy_histogram_kernel + hy_reduce_kernel", prof["global_histogram"], 4, "one read of the keys: 16-bit-prefix histogram (+ digit-0 counts)"),
                   kern(pass_kernel + " (pass A)", prof["pass0"], bpk_pass, "DigitBinningPass on the top byte, 16 position chains"),
                   kern(pass_kernel + " (pass B)", prof["pass1"], bpk_pass, "DigitBinningPass on byte 2 inside the top-byte buckets, 256 chains"),
                   kern("hy_local_sort_kernel" if not vb else "hy_local_sort_pairs_kernel", prof["pass2"], bpk_pass,
                        "one workgroup per 16-bit-prefix bucket: low 16 bits sorted in LDS, in place (LDS-instruction-bound; the slot includes the exit of "
                        "LSD pass 2's idle launch)" + ("" if not vb else "; the values move once, behind the keys"))]
    else:
        kernels = [kern("global_histogram_kernel + hist_reduce_kernel", prof["global_histogram"], 4, "one read of the keys: four joint histograms")] + \
                  [kern(f"DigitBinningPass {p}", prof[f"pass{p}"], bpk_pass, "one 8-bit LSD pass") for p in range(4)]
    return {
        "bound": "hbm",
        "kernel": ("digit_binning_dual_kernel (one 8-bit DigitBinningPass per launch; persistent workgroups; plain form for even keys, "
                   "position-chain form when the device plans PF_POS)") if not vb else
                  (pass_kernel + " (one 8-bit DigitBinningPass per launch; persistent workgroups)" if two_level else "digit_binning_kernel (one 8-bit DigitBinningPass)"),
        "plan": ("two-level: histogram of the top 16 bits, DigitBinningPass on byte 3, DigitBinningPass on byte 2 (256 chains), bucket-local LDS sort "
                 "of the low 16 bits — chosen on the device") if two_level else "GlobalHistogram + Scan + four LSD DigitBinningPasses",
        "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
        "traffic": traffic, "traffic_provenance": traffic_src,
        "algorithmic_bytes_per_launch": bpk_pass * n, "avg_launch_ms": pass_ms, "launches_averaged": list(passes),
        "frac_of_measured_copy_6290": achieved / 6290.0,
        "whole_sort": {
            "bytes_per_key": bpk_sort, "bytes_per_key_definition": "the reference algorithm's algorithmic bytes (SURVEY.md 8d: histogram read + 4 x (read + write))",
            "ms": prof["total"], "achieved_GBs": whole, "frac_of_8000": whole / HBM_PEAK_GBS,
            "bytes_per_key_moved_by_this_plan": bpk_moved, "moved_GBs": bpk_moved * n / (prof["total"] * 1e-3) / 1e9,
            "moved_frac_of_8000": bpk_moved * n / (prof["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            # BASELINE.json words the target as an "HBM-read roofline": the read half alone
            "read_bytes_per_key": 4 + 4 * (4 + vb),
            "read_only_frac_of_8000": (4 + 4 * (4 + vb)) * n / (prof["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
        },
        "per_kernel_ms": prof,
        "per_kernel": kernels,
        "tile_keys": tile_keys, "rank_mode": rank_mode,
        # the kernel furthest below its roofline (4 B/key read): the histogram sweep against the same peak
        "global_histogram": {"algorithmic_bytes": 4 * n, "ms": prof["global_histogram"],
                             "achieved_GBs": 4 * n / (prof["global_histogram"] * 1e-3) / 1e9,
                             "frac": 4 * n / (prof["global_histogram"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                             "traffic": hist_traffic, "traffic_provenance": hist_src},
        **({"frac_of_box_floor": {"pass": floor["tile_copy_ms_steady"] / pass_ms,
                                  "whole_sort": (floor["read_only_sweep_ms"] + floor["tile_copy_ms_first_after_read"] + 2.0 * floor["tile_copy_ms_steady"]
                                                 if two_level else floor["floor_ms"]) / prof["total"],
                                  "whole_sort_floor_definition": "read sweep + 3 tile-shaped copies" if two_level else "read sweep + 4 tile-shaped copies",
                                  "global_histogram": floor["read_only_sweep_ms"] / prof["global_histogram"]}}
           if floor and "floor_ms" in floor and not vb else {}),
    }


def measure_single(g, n, vb, entropy, steps, warm=1, prof_reps=3):
    """One 1-GPU configuration outside the headline region: `steps` back-to-back sorts of distinct pre-generated
    inputs between device synchronisations (wall clock, same protocol as the headline), then `prof_reps` profiled
    sorts for the per-kernel HIP-event times.  Returns (GKeys/s, ms per sort, profile, sorted?)."""
    import torch
    pairs = vb != 0
    vdt = torch.int32 if vb == 4 else torch.int64
    sorter = g.OneSweep(n, mode=g.MODE_PAIRS if pairs else g.MODE_KEYS_ONLY, value_bytes=vb)
    nb = max(steps, warm)
    ks = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
    vs = [torch.empty(n, dtype=vdt, device="cuda") for _ in range(nb)] if pairs else [None] * nb
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    valt = torch.empty(n, dtype=vdt, device="cuda") if pairs else None

    def gen(seed0):
        for i in range(nb):
            g.init_random(ks[i], seed0 + i, entropy, vs[i])
        torch.cuda.synchronize()

    gen(5000)
    for i in range(warm):
        sorter.sort(ks[i], vs[i], alt_keys=alt, alt_values=valt)
    gen(10)  # reference seeds 10 + i (GPUSortingCUDA.cu:22)
    t0 = time.perf_counter()
    for i in range(steps):
        sorter.sort(ks[i], vs[i], alt_keys=alt, alt_values=valt)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    sorter.check()
    ok = g.validate(ks[steps - 1], vs[steps - 1] if vb == 4 else None) == 0
    sorter.set_profiling(True)
    acc = {}
    for r in range(prof_reps):
        g.init_random(ks[0], 777 + r, entropy, vs[0])
        torch.cuda.synchronize()
        sorter.sort(ks[0], vs[0], alt_keys=alt, alt_values=valt)
        for k_, v_ in sorter.get_profile().items():
            acc[k_] = acc.get(k_, 0.0) + v_ / prof_reps
    sorter.set_profiling(False)
    tile = sorter.partition_size
    rank = sorter.rank_mode
    two_level = sorter.last_plan()["two_level"]
    sorter.close()
    return n * steps / dt / 1e9, dt / steps * 1e3, acc, ok, (tile, rank, two_level)


def more_block(g, n, log2, steps):
    """Driver-visible numbers for the other single-GPU configurations of BASELINE.json — configs[2] (u32 values),
    configs[4] (u64 values, Thearling-Smith entropy sweep) — and the keys-only entropy row, each with the
    roofline of its own dominant kernel.  Reference protocol: GPUSortingCUDA/GPUSortingCUDA.cu:36-39,
    GPUSortingD3D12/Tests.h:383-387,406-410.  Runs after the headline's timed region; never part of `value`."""
    ent_bits = (1.0, 0.811, 0.544, 0.337, 0.201)  # OneSweepDispatcher.cuh:201
    out = {"note": "measured after the headline region, same process; steps per entry = %d" % steps}
    for vb, name, cfg in ((4, "pairs_u32", 2), (8, "pairs_u64", 4)):
        gk, ms, prof, ok, (tile, rk, tl) = measure_single(g, n, vb, 0, steps)
        out[name] = {
            "workload": f"2^{log2} (uint32 key, uint{8 * vb} value) pairs OneSweep, 1 MI355X (BASELINE configs[{cfg}])",
            "value": gk, "unit": "GKeys/s", "ms_per_sort": ms, "steps": steps, "tile_keys": tile, "verified_sorted": bool(ok),
            "dtype": f"u32 keys + u{8 * vb} values",
            "roofline": roofline_block(n, vb, prof, log2, 0, "", tile, rk, None, tl),
        }
    rows = {}
    for vb, name in ((0, "keys"), (4, "pairs_u32"), (8, "pairs_u64")):
        row = []
        for preset in range(5):
            gk, ms, prof, ok, (tile, rk, tl) = measure_single(g, n, vb, preset, max(2, steps // 2), prof_reps=2)
            pass_ms = (prof["pass0"] + prof["pass1"]) / 2.0 if tl else sum(prof[f"pass{p}"] for p in range(4)) / 4.0
            row.append({"preset": preset + 1, "entropy_bits": ent_bits[preset], "value": gk, "unit": "GKeys/s", "ms_per_sort": ms,
                        "verified_sorted": bool(ok), "plan": "two-level" if tl else "LSD passes", "global_histogram_ms": prof["global_histogram"], "avg_pass_ms": pass_ms,
                        "tile_keys": tile, "rank_mode": rk,
                        "global_histogram_frac_of_8000": 4 * n / (prof["global_histogram"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                        "pass_frac_of_8000": (8 + 2 * vb) * n / (pass_ms * 1e-3) / 1e9 / HBM_PEAK_GBS})
        rows[name] = row
    out["entropy_sweep"] = rows
    out["entropy_sweep_note"] = ("keys-only, preset 1: the two-level plan (decided on the device from the 16-bit-prefix histogram); presets 2-5 (keys-only and "
                                 "pairs): the four LSD passes on position chains (the histogram kernel finds the prefixes / digit groups uneven); their counting "
                                 "passes run on 16 384-key tiles with packed counters (keys-only) and on 12 288-key tiles (pairs)")
    out["other_inputs"] = other_inputs(g, n, max(2, steps // 2))
    # ---- 64-bit keys: eight passes over 8-byte elements, planned by ONE histogram sweep + Scan ----
    n64 = min(n, 1 << 27)
    k64 = [torch.empty(n64, dtype=torch.int64, device="cuda") for _ in range(3)]
    a64 = torch.empty(n64, dtype=torch.int64, device="cuda")
    s64 = g.OneSweep(n64, key_type=g.KEY_UINT64)
    for t in k64:
        t.random_(-(1 << 62), 1 << 62)
    s64.sort(k64[0], alt_keys=a64)
    for t in k64:
        t.random_(-(1 << 62), 1 << 62)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for t in k64:
        s64.sort(t, alt_keys=a64)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / len(k64)
    s64.check()
    ok64 = g.validate(k64[-1], key_type=g.KEY_UINT64) == 0
    s64.close()
    out["keys64"] = {"workload": f"2^{n64.bit_length() - 1} uniform uint64 keys, keys-only", "value": n64 / dt / 1e9, "unit": "GKeys/s",
                     "ms_per_sort": dt * 1e3, "steps": len(k64), "verified_sorted": bool(ok64),
                     "bytes_per_key": 8 + 8 * 16, "achieved_GBs": (8 + 8 * 16) * n64 / dt / 1e9,
                     "frac_of_8000": (8 + 8 * 16) * n64 / dt / 1e9 / HBM_PEAK_GBS,
                     "structure": "one GlobalHistogram sweep (eight joint tables) + Scan + 8 passes on 8192-element tiles; bytes_per_key = 8 + 8 x 16"}
    del k64, a64
    torch.cuda.empty_cache()
    # ---- size sweep (reference: GPUSortingD3D12/Tests.h:392-393,415-416: 2^10 .. 2^27) ----
    out["size_sweep"] = size_sweep(g)
    return out


def other_inputs(g, n, steps):
    """Inputs the entropy presets do not cover: already sorted, reverse-sorted and block-clustered keys (every 2^20
    consecutive positions share their top byte) at the headline size, keys-only.  GKeys/s, the plan the device chose, sorted?"""
    import torch
    base = torch.empty(n, dtype=torch.int32, device="cuda")
    alt = torch.empty(n, dtype=torch.int32, device="cuda")
    work = torch.empty(n, dtype=torch.int32, device="cuda")
    s = g.OneSweep(n)
    rows = []
    for kind in ("sorted", "reverse_sorted", "block_clustered", "uniform"):
        g.init_random(base, 4242, 0)
        if kind in ("sorted", "reverse_sorted"):
            s.sort(base, alt_keys=alt)
            if kind == "reverse_sorted":
                base = torch.flip(base, dims=(0,)).contiguous()
        elif kind == "block_clustered":
            idx = torch.arange(n, dtype=torch.int32, device="cuda")
            base = ((base & 0x00FFFFFF) | (((idx >> 20) * 37 & 0xFF) << 24)).contiguous()
            del idx
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        tot = 0.0
        for i in range(steps + 1):
            work.copy_(base)
            torch.cuda.synchronize()
            a.record()
            s.sort(work, alt_keys=alt)
            b.record()
            b.synchronize()
            if i:
                tot += a.elapsed_time(b)
        s.check()
        ms = tot / steps
        rows.append({"input": kind, "ms_per_sort": ms, "value": n / ms / 1e6, "unit": "GKeys/s", "plan": "two-level" if s.last_plan()["two_level"] else "LSD passes",
                     "verified_sorted": bool(g.validate(work) == 0)})
    s.close()
    return rows


def comparator_block():
    """ROCm's own device radix sort (rocprim::radix_sort_keys / _pairs) beside this library: sizes 2^16 .. 2^28 and the five entropy presets,
    keys-only and (u32, u32) pairs, same box, same inputs (tools/rocprim_compare.cpp `sweep`; the reference does the same against CUB,
    GPUSortingCUDA/Sort/CubDispatcher.cuh:105-404).  Never part of `value`."""
    import subprocess
    exe = os.path.join(ROOT, "build", "rocprim_compare")
    if not os.path.exists(exe):
        return {"error": "build/rocprim_compare is missing (make tools)"}
    try:
        p = subprocess.run([exe, "sweep", "5"], capture_output=True, text=True, timeout=600)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")][-1]
        return json.loads(line)
    except Exception as e:  # noqa: BLE001
        return {"error": str(e)[:200]}


def size_sweep(g, sorts=10):
    """2^10 .. 2^27 keys, keys-only and (u32, u32) pairs, `sorts` back-to-back sorts of distinct pre-generated inputs per point
    (HIP events around the batch): microseconds per sort and GKeys/s."""
    rows = {"keys": [], "pairs_u32": [], "sorts_per_point": sorts,
            "routes": "n <= 8192: one workgroup, one launch; <= 2^20 (keys-only 2^23, u32 values 2^22): two launches (MSD pass + LDS "
                      "bucket sorts); above: GlobalHistogram + Scan + 4 DigitBinningPass; keys from 3 x 2^24, pairs from 2^25 + 1 elements: the two-level plan"}
    for vb, name in ((0, "keys"), (4, "pairs_u32")):
        for lg in range(10, 28):
            n = 1 << lg
            nb = max(1, min(sorts, (1 << 29) // (n * 4)))
            ks = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)]
            vs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nb)] if vb else [None] * nb
            alt = torch.empty(n, dtype=torch.int32, device="cuda")
            valt = torch.empty(n, dtype=torch.int32, device="cuda") if vb else None
            s = g.OneSweep(n, mode=g.MODE_PAIRS if vb else g.MODE_KEYS_ONLY, value_bytes=vb)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            total, done, ok = 0.0, 0, True
            for rnd in range(-1, (sorts + nb - 1) // nb):   # round -1 = warm-up
                for i in range(nb):
                    g.init_random(ks[i], 10 + done + i, 0, vs[i])
                torch.cuda.synchronize()
                a.record()
                for i in range(nb):
                    s.sort(ks[i], vs[i], alt_keys=alt, alt_values=valt)
                b.record()
                b.synchronize()
                if rnd >= 0:
                    total += a.elapsed_time(b)
                    done += nb
            s.check()
            ok = g.validate(ks[-1], vs[-1]) == 0
            us = total / done * 1e3
            row = {"log2_keys": lg, "us_per_sort": round(us, 2), "GKeys_per_s": round(n / us / 1e3, 3), "sorted": bool(ok)}
            if lg <= 20:
                # the same batch captured ONCE into a HIP graph and replayed: what the sorts cost when the host's launch calls are out of
                # the picture (a sort is pure stream work: nothing synchronous, nothing read back)
                try:
                    graph = torch.cuda.CUDAGraph()
                    with torch.cuda.graph(graph):
                        for i in range(nb):
                            s.sort(ks[i], vs[i], alt_keys=alt, alt_values=valt)
                    tg, dg = 0.0, 0
                    for rnd in range(3):
                        for i in range(nb):
                            g.init_random(ks[i], 40 + rnd * nb + i, 0, vs[i])
                        torch.cuda.synchronize()
                        a.record()
                        graph.replay()
                        b.record()
                        b.synchronize()
                        if rnd:
                            tg += a.elapsed_time(b)
                            dg += nb
                    s.check()
                    row["us_per_sort_graph_replay"] = round(tg / dg * 1e3, 2)
                    row["sorted_graph_replay"] = bool(g.validate(ks[-1], vs[-1]) == 0)
                    del graph
                except Exception as e:  # (a capture failure must not take the bench line with it)
                    row["graph_replay_error"] = str(e)[:120]
            s.close()
            rows[name].append(row)
    return rows


HEADLINE_MAX_BYTES = 4096  # the driver keeps a bounded tail of stdout: round 5's 28 KB line was cut and recorded as unparsed


def _r(x, nd=4):
    return round(x, nd) if isinstance(x, float) else x


def headline_line(out: dict) -> str:
    """The ONE JSON line the driver parses, from the full record: the contract's fields, `roofline` (dominant kernel, with the counter
    traffic) and `cpu_baseline`, plus a numbers-only digest of the `more` block — <= HEADLINE_MAX_BYTES, printed LAST.  Everything else
    (per-kernel tables, box floor, entropy / size sweeps, the rocPRIM comparator) goes to gpurun_out/bench_full.json and to '# bench_full'
    lines printed BEFORE this one."""
    rf = out["roofline"]
    ws = rf.get("whole_sort", {})
    prov = rf.get("traffic_provenance") or {}
    roof = {
        "bound": rf["bound"], "kernel": rf["kernel"].split(" (")[0], "achieved": _r(rf["achieved"], 1), "peak": rf["peak"], "unit": rf["unit"],
        "frac": _r(rf["frac"]), "traffic": rf.get("traffic"),
        "traffic_src": (f"{prov['source'].split(' (')[0]}, rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, sources_match={prov.get('sources_match')}"
                        if "source" in prov else prov.get("reason")),
        "algorithmic_bytes_per_launch": rf["algorithmic_bytes_per_launch"], "avg_launch_ms": _r(rf["avg_launch_ms"]),
        "launches_averaged": rf["launches_averaged"], "plan": rf["plan"].split(":")[0],
        "per_kernel_ms": {k: _r(v) for k, v in rf["per_kernel_ms"].items()},
        "whole_sort": {"bytes_per_key_metric": ws.get("bytes_per_key"), "frac_of_8000": _r(ws.get("frac_of_8000")),
                       "bytes_per_key_moved": ws.get("bytes_per_key_moved_by_this_plan"), "moved_frac_of_8000": _r(ws.get("moved_frac_of_8000"))},
    }
    if "frac_of_box_floor" in rf:
        roof["frac_of_box_tile_copy"] = _r(rf["frac_of_box_floor"]["pass"])
    cfg = out["config"]
    line = {k: out[k] for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline",
                                "dtype", "data")}
    line["config"] = {k: cfg[k] for k in ("workload", "keys_per_gpu", "entropy_preset", "generator", "tile_keys", "verified_sorted") if k in cfg}
    line["roofline"] = roof
    cb = out.get("cpu_baseline")
    line["cpu_baseline"] = None if not cb else {
        "value": _r(cb["value"], 5), "unit": cb["unit"], "cores": cb["cores"], "kind": cb["kind"], "sample": cb["sample"],
        "single_thread_std_sort": _r(cb.get("single_thread_std_sort", {}).get("value"), 5)}
    mg = out.get("multi_gpu")
    if mg:
        line["multi_gpu"] = {"phase_ms_max_over_ranks": {k: _r(v) for k, v in mg["phase_ms_max_over_ranks"].items()},
                             "exchange_GBps_per_link": _r(mg["exchange_GBps_per_link"], 1), "frac_of_link_peak": _r(mg["frac_of_link_peak"]),
                             "pipeline": mg["pipeline"].split(":")[0][:80], "bucket_layout": str(mg.get("bucket_layout", "")).split(" in ")[0].split(":")[0][:24],
                             "exchange_call_ab": mg.get("exchange_call_ab")}
    more = out.get("more")
    if more:
        dg = {}
        for k in ("pairs_u32", "pairs_u64", "keys64"):
            if k in more and "value" in more[k]:
                dg[k] = _r(more[k]["value"], 1)
        for k, rows in more.get("entropy_sweep", {}).items():
            dg["entropy_" + k] = [_r(r["value"], 1) for r in rows]
        if "other_inputs" in more:
            dg["other_inputs"] = {r["input"]: _r(r["value"], 1) for r in more["other_inputs"]}
        sw = more.get("size_sweep", {})
        for k in ("keys", "pairs_u32"):
            if k in sw:
                dg["size_sweep_" + k + "_log2_20_27"] = [_r(r["GKeys_per_s"], 1) for r in sw[k] if 20 <= r["log2_keys"] <= 27]
        rows = (more.get("comparator") or {}).get("rows") or []
        big = [r for r in rows if r.get("log2_keys") == 28 and r.get("entropy_preset") == 1]
        if big:
            dg["rocprim_2pow28"] = {r["mode"]: [_r(r["rocprim_GKeys_per_s"], 1), _r(r["gpusort_GKeys_per_s"], 1)] for r in big}
        dg["unit"] = "GKeys/s; measured after the timed region, never part of value"
        line["more_digest"] = dg
    line["full_record"] = out.get("full_record", "stdout lines starting '# bench_full' above this one")
    text = json.dumps(line)
    if len(text) > HEADLINE_MAX_BYTES:  # never let diagnostics take the record with them
        line.pop("more_digest", None)
        roof.pop("per_kernel_ms", None)
        text = json.dumps(line)
    return text


def emit(out: dict) -> None:
    """Full record -> gpurun_out/bench_full.json + '# bench_full <block> <json>' lines; then the compact line, LAST."""
    path = os.path.join(ROOT, "gpurun_out", "bench_full.json")
    try:
        os.makedirs(os.path.dirname(path), exist_ok=True)
        with open(path, "w") as f:
            json.dump(out, f)
        out["full_record"] = "gpurun_out/bench_full.json; stdout lines starting '# bench_full' above"
    except OSError:
        pass
    for block in ("roofline", "box_floor", "multi_gpu", "cpu_baseline"):
        if out.get(block) is not None:
            print(f"# bench_full {block} " + json.dumps(out[block]))
    for k, v in (out.get("more") or {}).items():
        print(f"# bench_full more.{k} " + json.dumps(v))
    sys.stdout.flush()
    print(headline_line(out))
    sys.stdout.flush()


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with torch.distributed.run --nproc-per-node N for --gpus N")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (the product path has no CPU fallback)")
    dry = args.dry_backend == "gloo"
    torch.cuda.set_device(0 if dry else local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if dry:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    coll_dev = "cpu" if dry else "cuda"  # where the few scalar collectives of this script live

    import gpusorting_amd as g
    from gpusorting_amd.sharded import ShardedOneSweep

    n = 1 << args.log2_keys
    strong = args.strong and world > 1
    if strong:
        n = max(n // world, 1)  # keys per GPU: the total stays 2^log2-keys
    K, W = args.steps, args.warmup
    pairs = args.pairs != 0
    vdt = torch.int32 if args.pairs == 4 else torch.int64
    mode = g.MODE_PAIRS if pairs else g.MODE_KEYS_ONLY

    # ---- inputs resident in HBM before the timed region: one buffer per step ----
    nbuf = max(K, W, 1)
    bufs = [torch.empty(n, dtype=torch.int32, device="cuda") for _ in range(nbuf)]
    vbufs = [torch.empty(n, dtype=vdt, device="cuda") for _ in range(nbuf)] if pairs else [None] * nbuf

    def regenerate(base_seed):
        for i in range(nbuf):
            g.init_random(bufs[i], base_seed + i + 1000 * rank, args.entropy, vbufs[i])
        torch.cuda.synchronize()

    if world == 1:
        sorter = g.OneSweep(n, mode=mode, value_bytes=args.pairs)
        if args.shape:
            t, k = (int(x) for x in args.shape.split("x"))
            sorter.set_shape(t, k)
        alt = torch.empty(n, dtype=torch.int32, device="cuda")
        valt = torch.empty(n, dtype=vdt, device="cuda") if pairs else None

        def step(i):
            sorter.sort(bufs[i], vbufs[i], alt_keys=alt, alt_values=valt)
            return bufs[i], vbufs[i], n
    else:
        # the product pipeline (C++ over RCCL behind the C-ABI); if ANY rank cannot bring it up, every rank runs the same
        # steps driven from Python over torch.distributed instead, and the JSON line says so
        pipeline = "gs_onesweep_sort_sharded (C++ over RCCL)"
        try:
            sharded = ShardedOneSweep(n, pairs=pairs, value_bytes=args.pairs or 4)
            up = 1
        except Exception as e:  # noqa: BLE001
            sharded, up = None, 0
            print(f"rank {rank}: C-ABI multi-GPU pipeline unavailable: {e}", file=sys.stderr)
        flag = torch.tensor([up], dtype=torch.int32, device=coll_dev)
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        if int(flag.item()) == 0:
            from gpusorting_amd.sharded import HipLocalEngine
            from gpusorting_amd import _lib as _gl
            if sharded is not None:
                sharded.close()
            cap = min(int(n * 1.25) + 256, _gl.GS_MAX_KEYS)
            sharded = ShardedOneSweep(n, engine=HipLocalEngine(cap, pairs, args.pairs or 4), pairs=pairs, value_bytes=args.pairs or 4)
            pipeline = "FALLBACK: the same steps driven from Python over torch.distributed (the C-ABI pipeline did not come up on every rank)"
        sorter = sharded.engine.sorter

        def step(i):
            return sharded.sort(bufs[i], values=vbufs[i])

    def fence():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
            torch.cuda.synchronize()

    # ---- warmup (untimed) ----
    regenerate(5000)
    for i in range(W):
        step(i)
    fence()

    # ---- timed: exactly K steps ----
    regenerate(10)  # reference seeds: 10 + i
    fence()
    t0 = time.perf_counter()
    last = None
    for i in range(K):
        last = step(i)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- N > 1, C-ABI pipeline over RCCL: the same steps with the OTHER bucket-exchange call, timed the same way, so that one run
    # records both (north_star names ncclAllToAllv; the default is grouped ncclSend / ncclRecv) ----
    exchange_ab = None
    if dist is not None and not dry and sharded._ctx and "FALLBACK" not in pipeline:
        try:
            reps = min(K, 5)
            sharded.set_alltoallv(True)
            for i in range(min(W, 1)):
                step(i)
            fence()
            t1 = time.perf_counter()
            for i in range(reps):
                step(i)
            fence()
            e2 = time.perf_counter() - t1
            t = torch.tensor([e2], dtype=torch.float64, device=coll_dev)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            exchange_ab = {"send_recv_ms_per_step": elapsed / K * 1e3, "alltoallv_ms_per_step": float(t.item()) / reps * 1e3,
                           "alltoallv_steps": reps, "timed_line_uses": "send_recv"}
            sharded.set_alltoallv(False)
            last = step(0)  # (the correctness check below looks at a default-mode step)
            fence()
        except Exception as e:  # noqa: BLE001
            exchange_ab = {"error": str(e)}

    # ---- correctness of the last step (outside the timed region) ----
    sorter.check()
    out_k, out_v, out_n = last
    sorted_ok = out_n == 0 or g.validate(out_k, out_v if args.pairs == 4 else None, n=out_n) == 0
    total_ok = True
    if dist is not None:
        lo = int(out_k[0].item()) & 0xFFFFFFFF if out_n else 0xFFFFFFFF
        hi = int(out_k[out_n - 1].item()) & 0xFFFFFFFF if out_n else 0
        info = torch.tensor([out_n, lo, hi], dtype=torch.int64, device=coll_dev)
        allinfo = [torch.empty_like(info) for _ in range(world)]
        dist.all_gather(allinfo, info)
        rows = [x.tolist() for x in allinfo]
        total_ok = sum(r[0] for r in rows) == n * world
        nonempty = [r for r in rows if r[0]]
        total_ok = total_ok and all(a[2] <= b[1] for a, b in zip(nonempty[:-1], nonempty[1:]))

    # ---- multi-GPU: phase times of the last step on every rank (HIP events inside gs_onesweep_sort_sharded),
    # bytes exchanged, and the exchange rate per xGMI link against its peak ----
    mgpu = None
    if dist is not None:
        p = sharded.profile() if sharded._ctx else {"split_ms": 0.0, "exchange_ms": 0.0, "local_sort_ms": 0.0, "total_ms": 0.0,
                                                    "bytes_sent": 0, "bytes_received": 0}
        mine = torch.tensor([p["split_ms"], p["exchange_ms"], p["local_sort_ms"], p["total_ms"], float(p["bytes_sent"]),
                             float(p["bytes_received"]), float(out_n)], dtype=torch.float64, device=coll_dev)
        allp = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(allp, mine)
        rows = [x.tolist() for x in allp]
        mx = [max(r[i] for r in rows) for i in range(4)]
        sent = [r[4] for r in rows]
        link_peak = 153.0  # GB/s per xGMI link, 7 links per GPU (task statement / BASELINE.md multi-GPU anchor)
        ex_s = mx[1] * 1e-3
        per_rank_gbs = max(sent) / ex_s / 1e9 if ex_s > 0 else 0.0
        mgpu = {
            "pipeline": pipeline + ": histogram + all-gather + plan + partition pass | bucket exchange | local OneSweep (two-level plan or four LSD passes, as the device decides)",
            "phase_ms_max_over_ranks": {"split": mx[0], "exchange": mx[1], "local_sort": mx[2], "total": mx[3]},
            "phase_ms_rank0": {"split": rows[0][0], "exchange": rows[0][1], "local_sort": rows[0][2], "total": rows[0][3]},
            "bytes_sent_off_rank": {"max": max(sent), "min": min(sent), "sum": sum(sent)},
            "bucket_keys": {"max": max(r[6] for r in rows), "min": min(r[6] for r in rows)},
            "exchange_GBps_per_rank": per_rank_gbs,
            "exchange_GBps_per_link": per_rank_gbs / max(world - 1, 1),
            "xgmi_link_peak_GBps": link_peak, "links_used_per_rank": world - 1,
            "frac_of_link_peak": per_rank_gbs / max(world - 1, 1) / link_peak,
            "split": sharded.last_split,
            "bucket_layout": ("bin-major in the local sort's alternate buffer (one message per (peer, top byte)): the local sort starts at the two-level "
                              "plan's second pass — the sender's split pass was its first") if getattr(sharded, "last_bin_major", False) else
                             "source-major in the output buffer: full local sort",
            "exchange_call_ab": exchange_ab,
        }

    # ---- per-kernel HIP-event profile of the local 4-pass sort (dominant kernel roofline) ----
    prof_sorter = sorter
    prof = None
    two_level = False
    if rank == 0:
        prof_sorter.set_profiling(True)
        acc = {}
        reps = min(K, 10)
        palt = torch.empty(n, dtype=torch.int32, device="cuda") if world > 1 else alt
        pvalt = (torch.empty(n, dtype=vdt, device="cuda") if world > 1 else valt) if pairs else None
        g.init_random(bufs[0], 777, args.entropy, vbufs[0])
        for r in range(reps):
            g.init_random(bufs[0], 777 + r, args.entropy, vbufs[0])
            torch.cuda.synchronize()
            if world == 1:
                prof_sorter.sort(bufs[0], vbufs[0], alt_keys=palt, alt_values=pvalt)
            else:
                prof_sorter.sort(bufs[0], vbufs[0], n=n, alt_keys=palt, alt_values=pvalt)
            p = prof_sorter.get_profile()
            for k_, v_ in p.items():
                acc[k_] = acc.get(k_, 0.0) + v_
        prof = {k_: v_ / reps for k_, v_ in acc.items()}
        prof_sorter.set_profiling(False)
        two_level = prof_sorter.last_plan()["two_level"]
    if dist is not None:
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    floor = None
    if world == 1 and not args.no_more:
        bufs.clear()      # (the calibration buffers need room)
        vbufs.clear()
        last = out_k = out_v = None
        torch.cuda.empty_cache()
        floor = box_floor(n)
    total_keys = n * world * K
    value = total_keys / elapsed / 1e9
    ms_per_step = elapsed / K * 1e3
    out = {
        "metric": baseline_metric(), "value": value, "unit": "GKeys/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "strong" if strong else "weak", "vs_baseline": None,
        "dtype": "u32" if not pairs else f"u32 keys + u{8 * args.pairs} values", "data": "synthetic" if not dry else "synthetic (REHEARSAL: ranks share one GPU, gloo; not a measurement)",
        "config": {
            "workload": (f"2^{args.log2_keys} uniform-random uint32 {'pairs' if pairs else 'keys-only'} OneSweep, 1 MI355X "
                         f"(BASELINE configs[{2 if args.pairs == 4 else 4 if args.pairs == 8 else 1}])") if world == 1 else
                        (f"2^{args.log2_keys} uint32 keys per GPU x {world} GPUs: MSD split + RCCL bucket exchange + per-GPU "
                         f"OneSweep (BASELINE configs[3] shape, weak scaling)") if not strong else
                        (f"2^{args.log2_keys} uint32 keys in total over {world} GPUs ({n} per GPU): MSD split + RCCL bucket exchange + "
                         f"per-GPU OneSweep (SURVEY 8d cfg 4, strong scaling)"),
            "timed_region": "whole sort per step: histogram sweep (incl. the state clear) + Scan + every launch of the plan the device chooses "
                            "(two-level: 2 DigitBinningPass + bucket-local sort; otherwise 4 DigitBinningPass)"
                            + ("" if world == 1 else ", after the top-byte split + bucket exchange of the step (all inside the timed region)"),
            "keys_per_gpu": n, "entropy_preset": args.entropy + 1, "generator": "InitRandom seed 10+i (+1000*rank)",
            "tile_keys": sorter.partition_size, "verified_sorted": bool(sorted_ok and total_ok),
        },
        "roofline": roofline_block(n, args.pairs, prof, args.log2_keys, args.entropy, args.shape, sorter.partition_size, sorter.rank_mode, floor, two_level),
    }
    if floor is not None:
        out["box_floor"] = floor
    if mgpu is not None:
        out["multi_gpu"] = mgpu
    if world == 1 and not args.pairs and not args.entropy and not args.shape and not args.no_more:
        # free the headline's buffers first: the block allocates its own
        bufs.clear()
        vbufs.clear()
        last = out_k = out_v = None
        sorter.close()
        torch.cuda.empty_cache()
        out["more"] = more_block(g, n, args.log2_keys, args.more_steps)
        out["more"]["comparator"] = comparator_block()
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_log2)
    else:
        out["cpu_baseline"] = None
    emit(out)
    if dist is not None:
        dist.destroy_process_group()
    if not (sorted_ok and total_ok):
        raise SystemExit("bench: output of the last step is NOT sorted")


if __name__ == "__main__":
    main()
