"""Concurrent phase of the n1_dual step (decode + latent passes -> System-1 of the System-2 envs on the main stream || System-1 of the other
envs on the side stream) with the SIDE stream restricted to a subset of the CUs (hipExtStreamCreateWithCUMask): does the decode chain, which
waits for CU slots behind the System-1 workgroups, end earlier when some CUs are never offered to the side stream?

    python tools/cu_mask_probe.py            # sweep: no mask, then side stream on 240 / 224 / 208 / 192 / 160 CUs, both bit layouts
"""
import ctypes as C
import sys
import time
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402

hip = C.CDLL("libamdhip64.so")


def masked_stream(bits):
    """bits: list of 256 0/1 -> torch ExternalStream on a HIP stream created with that CU mask."""
    words = (C.c_uint32 * 8)()
    for i, b in enumerate(bits):
        if b:
            words[i // 32] |= 1 << (i % 32)
    s = C.c_void_p()
    rc = hip.hipExtStreamCreateWithCUMask(C.byref(s), 8, words)
    assert rc == 0, f"hipExtStreamCreateWithCUMask -> {rc}"
    return torch.cuda.ExternalStream(s.value)


def layouts(n_on):
    """two guesses at how mask bits map to CUs: 'low' = the first n bits (whole XCDs / SEs drop out if the numbering is contiguous),
    'strided' = the same number of CUs kept in every block of 32 bits."""
    low = [1 if i < n_on else 0 for i in range(256)]
    per = n_on // 8
    strided = [1 if (i % 32) < per else 0 for i in range(256)]
    inter = [1 if (i // 8) < per else 0 for i in range(256)]      # bit i -> XCD i % 8, CU i // 8 (round-robin numbering)
    return {"low": low, "per32": strided, "rr8": inter}


a = bench.default_args()
dev = torch.device("cuda:0")
wl = bench.N1Dual(a, dev, 0)
wl.capture()
m = max(wl.mb)
nA = 64 - m
main = torch.cuda.current_stream()


def timeline(side, reps=4):
    out = []
    for _ in range(reps):
        ev = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "dec", "s1b", "s1a")}
        torch.cuda.synchronize()
        ev["start"].record(main)
        side.wait_stream(main)
        with torch.cuda.stream(side):
            wl.gA[nA]()
            ev["s1a"].record(side)
        wl.gD[m]()
        ev["dec"].record(main)
        wl.gB[m]()
        ev["s1b"].record(main)
        main.wait_stream(side)
        torch.cuda.synchronize()
        out.append((ev["start"].elapsed_time(ev["dec"]), ev["start"].elapsed_time(ev["s1b"]), ev["start"].elapsed_time(ev["s1a"])))
    out = out[1:]
    return tuple(sum(x[k] for x in out) / len(out) for k in range(3))


def alone(side, reps=3):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        with torch.cuda.stream(side):
            wl.gA[nA]()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps * 1e3


print("side stream CUs | layout | S1(%d) alone ms | decode ends | S1 small ends (main) | S1 side ends | phase" % nA)
d, b, s = timeline(wl.side)
print(f"256 | torch stream | {alone(wl.side):.1f} | {d:.1f} | {b:.1f} | {s:.1f} | {max(b, s):.1f}", flush=True)
for n_on in (240, 224, 208, 192, 160):
    for name, bits in layouts(n_on).items():
        st = masked_stream(bits)
        al = alone(st)
        d, b, s = timeline(st)
        print(f"{n_on} | {name} | {al:.1f} | {d:.1f} | {b:.1f} | {s:.1f} | {max(b, s):.1f}", flush=True)
# the other way round: the decode chain + S1 small on a masked stream of its own (strict partition: side on the complement)
for n_dec in (32, 64):
    for name, bits in layouts(n_dec).items():
        comp = [1 - x for x in bits]
        sd, ss = masked_stream(bits), masked_stream(comp)
        ev = {k: torch.cuda.Event(enable_timing=True) for k in ("start", "dec", "s1b", "s1a")}
        res = []
        for _ in range(3):
            torch.cuda.synchronize()
            ev["start"].record(main)
            sd.wait_stream(main)
            ss.wait_stream(main)
            with torch.cuda.stream(ss):
                wl.gA[nA]()
                ev["s1a"].record(ss)
            with torch.cuda.stream(sd):
                wl.gD[m]()
                ev["dec"].record(sd)
            main.wait_stream(sd)
            wl.gB[m]()
            ev["s1b"].record(main)
            main.wait_stream(ss)
            torch.cuda.synchronize()
            res = (ev["start"].elapsed_time(ev["dec"]), ev["start"].elapsed_time(ev["s1b"]), ev["start"].elapsed_time(ev["s1a"]))
        print(f"partition: decode on {n_dec} CUs ({name}), side on the other {256 - n_dec}: decode ends {res[0]:.1f} | S1 small ends {res[1]:.1f} | S1 side ends {res[2]:.1f} | phase {max(res[1], res[2]):.1f}", flush=True)
