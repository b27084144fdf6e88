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
    ap.add_argument("--dry-backend", type=str, default="", help="rehearsal of the N>1 code path on a one-GPU box: 'gloo' = "
                    "every rank on cuda:0, collectives over gloo (host-staged); never used for reported numbers")
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
        "sample": f"2^{log2n} uint32 keys (InitRandom seed 10, preset 1), chunked std::sort + merge tree on {used} "
                  f"of {threads} hw threads, {dt:.2f} s",
        "single_thread_std_sort": {"value": n1 / dt1 / 1e9, "unit": "GKeys/s", "sample": f"2^{min(log2n, 26)} keys, {dt1:.2f} s"},
    }


def pmc_traffic(args, sorter):
    """HBM bytes per DigitBinningPass launch from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, collected in
    separate rocprofv3 --pmc passes of this same command and committed under profiles/); None if the committed
    measurement is for another workload/tile shape."""
    path = os.path.join(ROOT, "profiles", "r01_pmc_traffic.json")
    if args.log2_keys != 28 or args.entropy or args.shape or not os.path.exists(path):
        return None
    d = json.load(open(path))
    if args.pairs:
        d = d.get(f"pairs{args.pairs}")
        return d["traffic_bytes_per_launch"] if d else None
    return d["traffic_bytes_per_launch"] if f"<{sorter.partition_size // 32},32,0,0," in d["kernel"] else None


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
        sharded = ShardedOneSweep(n, pairs=pairs, value_bytes=args.pairs or 4)
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

    # ---- per-kernel HIP-event profile of the local 4-pass sort (dominant kernel roofline) ----
    prof_sorter = sorter
    prof = None
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
    if dist is not None:
        dist.barrier()

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    bytes_per_key_pass = 8 + 2 * args.pairs            # read + write of key (+ value) per DigitBinningPass
    bytes_per_key_sort = 4 + 4 * bytes_per_key_pass    # + one histogram read (36 / 68 / 100)
    pass_ms = sum(prof[f"pass{p}"] for p in range(4)) / 4.0
    achieved = bytes_per_key_pass * n / (pass_ms * 1e-3) / 1e9
    total_keys = n * world * K
    value = total_keys / elapsed / 1e9
    ms_per_step = elapsed / K * 1e3
    out = {
        "metric": baseline_metric(), "value": value, "unit": "GKeys/s", "n_gpus": world, "steps": K, "warmup": W, "ms_per_step": ms_per_step,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u32" if not pairs else f"u32 keys + u{8 * args.pairs} values", "data": "synthetic" if not dry else "synthetic (REHEARSAL: ranks share one GPU, gloo; not a measurement)",
        "config": {
            "workload": (f"2^{args.log2_keys} uniform-random uint32 {'pairs' if pairs else 'keys-only'} OneSweep, 1 MI355X "
                         f"(BASELINE configs[{2 if args.pairs == 4 else 4 if args.pairs == 8 else 1}])") if world == 1 else
                        (f"2^{args.log2_keys} uint32 keys per GPU x {world} GPUs: MSD split + RCCL all-to-all-v + per-GPU "
                         f"OneSweep (BASELINE configs[3] shape, weak scaling)"),
            "timed_region": "whole sort per step: GlobalHistogram (incl. the state clear) + Scan + 4 DigitBinningPass"
                            + ("" if world == 1 else ", after the top-byte split + all-to-all-v exchange of the step"),
            "keys_per_gpu": n, "entropy_preset": args.entropy + 1, "generator": "InitRandom seed 10+i (+1000*rank)",
            "tile_keys": sorter.partition_size, "verified_sorted": bool(sorted_ok and total_ok),
        },
        "roofline": {
            "bound": "hbm", "kernel": "digit_binning_kernel (one 8-bit DigitBinningPass)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": pmc_traffic(args, sorter),
            "algorithmic_bytes_per_launch": bytes_per_key_pass * n, "avg_launch_ms": pass_ms,
            "frac_of_measured_copy_6290": achieved / 6290.0,
            "whole_sort": {
                "bytes_per_key": bytes_per_key_sort, "ms": prof["total"],
                "achieved_GBs": bytes_per_key_sort * n / (prof["total"] * 1e-3) / 1e9,
                "frac_of_8000": bytes_per_key_sort * n / (prof["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
                # BASELINE.json words the target as an "HBM-read roofline": the read half alone
                "read_bytes_per_key": 4 + 4 * (4 + args.pairs),
                "read_only_frac_of_8000": (4 + 4 * (4 + args.pairs)) * n / (prof["total"] * 1e-3) / 1e9 / HBM_PEAK_GBS,
            },
            "per_kernel_ms": prof,
        },
    }
    if world == 1 and not args.no_cpu_baseline:
        out["cpu_baseline"] = cpu_baseline(args.cpu_log2)
    else:
        out["cpu_baseline"] = None
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()
    if not (sorted_ok and total_ok):
        raise SystemExit("bench: output of the last step is NOT sorted")


if __name__ == "__main__":
    main()
