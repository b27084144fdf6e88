"""Round 4: the local-sort plan against torch.sort and against the GlobalHistogram / Scan / 4-pass pipeline.
usage: python tools/r04_ls_check.py [check] [time LOG2 REPS]"""
import sys, time
import torch
sys.path.insert(0, ".")
import gpusorting_amd as ga
from gpusorting_amd import onesweep as osw

dev = torch.device("cuda")

def gen(n, kind, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    if kind.startswith("and"):
        k = torch.full((n,), -1, dtype=torch.int32, device=dev)
        for _ in range(int(kind[3:]) + 1):
            k &= torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
        return k
    if kind == "const": return torch.full((n,), 0x12345678, dtype=torch.int32, device=dev)
    if kind == "low16": return torch.randint(0, 65536, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    if kind == "sorted": return torch.arange(n, dtype=torch.int64, device=dev).mul_(7).to(torch.int32)
    if kind == "blocks":  # long runs of one low byte: oversize units
        return (torch.arange(n, dtype=torch.int64, device=dev) // 300000 % 256 + (torch.randint(0, 2**23, (n,), dtype=torch.int64, device=dev, generator=g) << 8)).to(torch.int32)
    if kind == "rare":   # one low byte value per ~16384 keys
        r = torch.randint(0, 2**31, (n,), dtype=torch.int64, device=dev, generator=g)
        return torch.where(r % 16384 == 0, r | 0xff, r & ~0xff).to(torch.int32)
    raise ValueError(kind)

def expect(keys, key_type, order):
    if key_type == osw.KEY_UINT32: v = keys.to(torch.int64) & 0xffffffff
    elif key_type == osw.KEY_INT32: v = keys.to(torch.int64)
    else:
        b = keys.to(torch.int64) & 0xffffffff
        v = torch.where(b >> 31 == 1, b ^ 0xffffffff, b | 0x80000000)
    idx = torch.sort(v, stable=True, descending=(order == 1)).indices
    return keys[idx]

def check():
    bad = 0
    cases = []
    for lg, extra in ((25, 1), (25, 12345), (26, 0), (26, -7777), (27, 3)):
        cases.append((2**lg + extra, "and0"))
    for kind in ("and1", "and2", "and4", "const", "low16", "sorted", "blocks", "rare"):
        cases.append((2**25 + 4099, kind))
    for n, kind in cases:
        for kt, order in ((osw.KEY_UINT32, 0), (osw.KEY_INT32, 1), (osw.KEY_FLOAT32, 0)) if kind == "and0" and n < 2**26 else ((osw.KEY_UINT32, 0), (osw.KEY_UINT32, 1)):
            k0 = gen(n, kind, n & 0xffff)
            if kt == osw.KEY_FLOAT32: k0 = torch.where((k0 & 0x7f800000) == 0x7f800000, k0 & ~0x00800000, k0)  # no NaN/inf patterns for the torch reference
            want = expect(k0, kt, order)
            s = osw.OneSweep(n, order=order, key_type=kt)
            for plan in (1, 0):
                s.set_plan(bool(plan))
                k = k0.clone()
                s.sort(k); s.check()
                ok = bool(torch.equal(k, want))
                bad += not ok
                print(f"n={n} kind={kind} kt={kt} order={order} plan={plan}: {'ok' if ok else 'MISMATCH'}", flush=True)
                if not ok:
                    d = (k != want).nonzero().flatten()
                    print("   first diffs at", d[:8].tolist(), "of", d.numel(), " got", k[d[:4]].tolist(), "want", want[d[:4]].tolist())
            s.close()
    print("CHECK", "PASSED" if bad == 0 else f"FAILED ({bad})")
    return bad

def timeit(lg, reps):
    n = 2**lg
    for kind in ("and0", "and1", "and2", "and4"):
        k0 = gen(n, kind, 10)
        s = osw.OneSweep(n)
        for plan in (0, 1, 0, 1):
            s.set_plan(bool(plan)); s.set_profiling(True)
            acc = {}
            wall = []
            for r in range(reps + 1):
                k = k0.clone(); torch.cuda.synchronize()
                t0 = time.perf_counter(); s.sort(k); torch.cuda.synchronize(); t1 = time.perf_counter()
                pr = s.get_profile()
                if r:
                    wall.append(t1 - t0)
                    for kk, v in pr.items(): acc[kk] = acc.get(kk, 0.0) + v / reps
            print(f"2^{lg} {kind} plan={plan}: total {acc['total']:.4f} ms = {n / acc['total'] * 1e-6:.1f} GKeys/s | " +
                  " ".join(f"{kk}={v:.4f}" for kk, v in acc.items() if kk != 'total'), flush=True)
        s.close()

if __name__ == "__main__":
    a = sys.argv[1:]
    rc = 0
    if not a or "check" in a: rc = check()
    if "quick" in a:  # u32 only (works with a -DGS_MINIMAL build)
        for n, kind in ((2**25 + 12345, "and0"), (2**26, "and2"), (2**25 + 77, "blocks")):
            k0 = gen(n, kind, 7); want = expect(k0, osw.KEY_UINT32, 0)
            s_ = osw.OneSweep(n); s_.set_plan(True); k = k0.clone(); s_.sort(k); s_.check()
            ok = bool(torch.equal(k, want)); rc += not ok
            print(f"quick n={n} {kind}: {'ok' if ok else 'MISMATCH'}", flush=True); s_.close()
    if "time" in a:
        i = a.index("time"); timeit(int(a[i + 1]), int(a[i + 2]))
    sys.exit(1 if rc else 0)
