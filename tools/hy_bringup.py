"""Bring-up / regression aid of the two-level plan (hybrid_kernels.hpp): stage-by-stage checks against numpy and per-slot timings.
  python tools/hy_bringup.py [log2n ...]
For every size: (1) the plan forced (plan=2) with the bucket-local sort skipped (debug flag): alt must hold the input stably
partitioned by the top byte (pass A), keys the input stably sorted by its top 16 bits (pass B); (2) the complete sort, ascending and
descending, against numpy; (3) skewed keys (preset 3), which must fall back to the LSD passes; (4) per-slot times of plan 1 (LSD only)
and plan 0 / 2 at that size."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import gpusorting_amd as g  # noqa: E402


def stable_by(k, shift, bits):
    return k[np.argsort((k >> np.uint32(shift)) & np.uint32((1 << bits) - 1), kind="stable")]


def run(log2n, extra=0):
    n = (1 << log2n) + extra
    dk = torch.empty(n, dtype=torch.int32, device="cuda")
    g.init_random(dk, 10 + log2n, 0)
    k0 = dk.cpu().numpy().view(np.uint32).copy()
    ok = True
    if n <= (1 << 26) + 100:
        s = g.OneSweep(n, plan=2, position_chains_min_log2=20, small_path=0, mid_path=0, debug_flags=0x40000000)
        alt = torch.zeros(n, dtype=torch.int32, device="cuda")
        d = dk.clone()
        s.sort(d, alt_keys=alt)
        s.check()
        lp = s.last_plan()
        a = alt.cpu().numpy().view(np.uint32)
        b = d.cpu().numpy().view(np.uint32)
        wa = stable_by(k0, 24, 8)
        wb = stable_by(k0, 16, 16)
        ea, eb = int((a != wa).sum()), int((b != wb).sum())
        print(f"n=2^{log2n}+{extra}: plan={lp} passA mismatches={ea} passB mismatches={eb}", flush=True)
        if ea:
            i = int(np.nonzero(a != wa)[0][0])
            print(f"   passA first mismatch at {i}: got {a[i]:08x} want {wa[i]:08x}; sorted-by-top-byte? {bool((np.diff((a >> 24).astype(np.int64)) >= 0).all())}; multiset ok? {bool((np.sort(a) == np.sort(k0)).all())}")
        if eb:
            i = int(np.nonzero(b != wb)[0][0])
            print(f"   passB first mismatch at {i}: got {b[i]:08x} want {wb[i]:08x}; sorted-by-top16? {bool((np.diff((b >> 16).astype(np.int64)) >= 0).all())}; multiset ok? {bool((np.sort(b) == np.sort(k0)).all())}")
        ok = ok and ea == 0 and eb == 0 and lp["two_level"]
        s.close()
    want = np.sort(k0)
    for order in (0, 1):
        s = g.OneSweep(n, order=order, plan=2, position_chains_min_log2=20, small_path=0, mid_path=0)
        d = dk.clone()
        s.sort(d)
        s.check()
        lp = s.last_plan()
        got = d.cpu().numpy().view(np.uint32)
        w = want if order == 0 else want[::-1]
        e = int((got != w).sum())
        print(f"n=2^{log2n}+{extra} order={order}: plan={lp} mismatches={e} state={s.check_state()}", flush=True)
        if e:
            i = int(np.nonzero(got != w)[0][0])
            print(f"   first mismatch at {i}: got {got[i]:08x} want {w[i]:08x}; multiset ok? {bool((np.sort(got) == want).all())}")
        ok = ok and e == 0 and lp["two_level"]
        s.close()
    # skewed keys: the device must choose the LSD passes
    g.init_random(dk, 99, 2)
    k1 = dk.cpu().numpy().view(np.uint32).copy()
    s = g.OneSweep(n, plan=2, position_chains_min_log2=20, small_path=0, mid_path=0)
    s.sort(dk)
    s.check()
    lp = s.last_plan()
    e = int((dk.cpu().numpy().view(np.uint32) != np.sort(k1)).sum())
    print(f"n=2^{log2n}+{extra} preset 3: plan={lp} mismatches={e}", flush=True)
    ok = ok and e == 0 and not lp["two_level"]
    s.close()
    # timings
    for plan in (1, 2):
        s = g.OneSweep(n, plan=plan, position_chains_min_log2=20)
        s.set_profiling(True)
        best = None
        for it in range(6):
            g.init_random(dk, 10 + it, 0)
            s.sort(dk)
            p = s.get_profile()
            if it >= 1 and (best is None or p["total"] < best["total"]):
                best = p
        print(f"n=2^{log2n}+{extra} plan={plan}: " + " ".join(f"{k}={v:.4f}" for k, v in best.items()) + f"  -> {n / best['total'] / 1e6:.1f} GKeys/s ({s.last_plan()})", flush=True)
        s.close()
    return ok


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [22, 24, 26, 28]
    allok = True
    for l in sizes:
        allok = run(l, 12345 if l < 28 else 0) and allok
    print("ALL OK" if allok else "FAILURES")
    sys.exit(0 if allok else 1)
