"""Per-slot timings of the two-level plan against the LSD passes (HIP events on the sort's stream; gs_onesweep_set_profiling).
  [GPUSORT_LIB=gpusorting_amd/lib/libgpusort_tuning.so] python tools/hy_perf.py [log2n ...] [--flags 0,1,2] [--preset P]
Slots of a two-level sort: global_histogram = histogram sweep + slice sum, scan = both Scan kernels, pass0 = pass A (top byte),
pass1 = pass B (byte 2, 256 chains), pass2 = bucket-local sort (+ the exit of LSD pass 2's launch), pass3 = the exit of LSD pass 3's."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402


def main():
    args = sys.argv[1:]
    flags, preset, sizes, inp, check = [0], 0, [], "generator", True
    i = 0
    while i < len(args):
        if args[i] == "--flags":
            flags = [int(x, 0) for x in args[i + 1].split(",")]
            i += 2
        elif args[i] == "--input":   # generator | sorted | reverse | clustered (every 2^20 positions share their top byte)
            inp = args[i + 1]
            i += 2
        elif args[i] == "--no-check":  # ablation builds (GS_EXP & 256: no look-back wait): the result is not sorted
            check = False
            i += 1
        elif args[i] == "--preset":
            preset = int(args[i + 1])
            i += 2
        else:
            sizes.append(int(args[i]))
            i += 1
    for log2n in sizes or [28]:
        n = 1 << log2n
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        for plan, fl in [(1, 0)] + [(2, f) for f in flags]:
            s = g.OneSweep(n, plan=plan, position_chains_min_log2=20, debug_flags=fl)
            s.set_profiling(True)
            runs = []
            for it in range(9):
                g.init_random(dk, 10 + it, preset)
                if inp in ("sorted", "reverse"):
                    dk = torch.sort(dk.to(torch.int64) & 0xFFFFFFFF, descending=inp == "reverse").values.to(torch.int32)
                elif inp == "clustered":
                    idx = torch.arange(n, dtype=torch.int32, device="cuda")
                    dk = ((dk & 0x00FFFFFF) | (((idx >> 20) * 37 & 0xFF) << 24)).contiguous()
                want = torch.sort(dk.to(torch.int64) & 0xFFFFFFFF).values if it == 8 else None  # (keys-only u32: the sorted array is unique)
                s.sort(dk)
                p = s.get_profile()
                if it >= 1:
                    runs.append(p)
                if it == 8 and check:
                    assert g.validate(dk) == 0, "not sorted"
                    assert bool(((dk.to(torch.int64) & 0xFFFFFFFF) == want).all().item()), "differs from torch.sort"
                    del want
            runs.sort(key=lambda r: r["total"])
            med = runs[len(runs) // 2]
            print(f"2^{log2n} {inp} preset {preset + 1} plan={plan} flags={fl:#x}: median " + " ".join(f"{k}={v:.4f}" for k, v in med.items()) +
                  f" -> {n / med['total'] / 1e6:.1f} GKeys/s (best {n / runs[0]['total'] / 1e6:.1f}) {s.last_plan()}", flush=True)
            s.close()


if __name__ == "__main__":
    main()
