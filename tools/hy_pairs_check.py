"""Bring-up / timing aid of the two-level plan for (key, value) pairs: value = original index, checked against numpy's stable argsort
(descending: its exact reverse), then per-slot timings of the LSD passes (plan 1) against the two-level plan (plan 2).
  [GPUSORT_LIB=...] python tools/hy_pairs_check.py [log2n ...] [--time 28]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402


def check(log2n, extra, vb, preset=0):
    n = (1 << log2n) + extra
    ok = True
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    g.init_random(dk, 10 + log2n, preset)
    k0 = dk.cpu().numpy().view(np.uint32).copy()
    perm = np.argsort(k0, kind="stable")
    for order in (0, 1):
        s = g.OneSweep(n, order=order, mode=g.MODE_PAIRS, value_bytes=vb, plan=2, position_chains_min_log2=20, small_path=0, mid_path=0)
        d = dk.clone()
        dv = torch.arange(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
        s.sort(d, dv)
        s.check()
        lp = s.last_plan()
        want = perm if order == 0 else perm[::-1]
        gk, gv = d.cpu().numpy().view(np.uint32), dv.cpu().numpy().astype(np.int64)
        ek, ev = int((gk != k0[want]).sum()), int((gv != want).sum())
        print(f"vb={vb} n=2^{log2n}+{extra} preset {preset + 1} order={order}: plan={lp} key mismatches={ek} value mismatches={ev} state={s.check_state()['keys_per_pass']}", flush=True)
        if ek or ev:
            i = int(np.nonzero((gk != k0[want]) | (gv != want))[0][0])
            print(f"   first mismatch at {i}: key {gk[i]:08x} want {k0[want][i]:08x}, value {gv[i]} want {want[i]}; keys sorted? {bool((np.diff(gk.astype(np.int64)) * (1 - 2 * order) >= 0).all())}")
        ok = ok and ek == 0 and ev == 0 and lp["two_level"] == (preset == 0)
        s.close()
    return ok


def timing(log2n, vb):
    n = 1 << log2n
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    dv = torch.empty(n, dtype=torch.int32 if vb == 4 else torch.int64, device="cuda")
    for plan in (1, 2):
        s = g.OneSweep(n, mode=g.MODE_PAIRS, value_bytes=vb, plan=plan, position_chains_min_log2=20)
        s.set_profiling(True)
        runs = []
        for it in range(7):
            g.init_random(dk, 10 + it, 0, dv)
            s.sort(dk, dv)
            p = s.get_profile()
            if it >= 1:
                runs.append(p)
        assert g.validate(dk, dv if vb == 4 else None) == 0
        runs.sort(key=lambda r: r["total"])
        med = runs[len(runs) // 2]
        print(f"vb={vb} 2^{log2n} plan={plan}: median " + " ".join(f"{k}={v:.4f}" for k, v in med.items()) + f" -> {n / med['total'] / 1e6:.1f} GKeys/s {s.last_plan()}", flush=True)
        s.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    tsz = None
    if "--time" in args:
        tsz = int(args[args.index("--time") + 1])
        args = args[:args.index("--time")]
    allok = True
    for l in [int(a) for a in args]:
        for vb in (4, 8):
            allok = check(l, 12345, vb) and allok
    if args:
        allok = check(int(args[0]), 777, 8, preset=2) and allok
    if tsz:
        for vb in (4, 8):
            timing(tsz, vb)
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
