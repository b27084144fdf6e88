"""Per-slot HIP-event times of (u32, u32) / (u32, u64) pair sorts at 2^log2n, LSD passes (plan 1) against the default routing.
  python tools/pairs_perf.py [log2n=28] [value_bytes ...]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402


def main():
    log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
    vbs = [int(x) for x in sys.argv[2:]] or [4, 8]
    n = 1 << log2n
    for vb in vbs:
        dk = torch.empty(n, dtype=torch.int32, device="cuda")
        dv = torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
        for plan in (1, 0):
            s = g.OneSweep(n, mode=g.MODE_PAIRS, value_bytes=vb, plan=plan)
            s.set_profiling(True)
            runs = []
            for it in range(7):
                g.init_random(dk, 10 + it, 0, dv)
                s.sort(dk, dv)
                if it:
                    runs.append(s.get_profile())
            assert g.validate(dk, dv if vb == 4 else None) == 0
            runs.sort(key=lambda r: r["total"])
            m = runs[len(runs) // 2]
            print(f"2^{log2n} vb={vb} plan={plan}: " + " ".join(f"{k}={v:.4f}" for k, v in m.items()) + f" -> {n / m['total'] / 1e6:.1f} GKeys/s {s.last_plan()}", flush=True)
            s.close()


if __name__ == "__main__":
    main()
