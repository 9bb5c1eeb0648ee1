#!/usr/bin/env python3
"""Achievable streaming rate on this box ("re-measure an achievable-peak copy/triad and report both").

fp64 copy (y = x: 16 B per element) and triad (a = b + s c: 24 B per element) over a range of footprints, timed with events
on torch's stream.  Small footprints sit partly in the 256 MB Infinity Cache, which is also where the C3 SpMV (277 MB per
launch) lives; the large ones are the plain HBM3E rate.

    python tools/hbm_peak.py
"""
import json

import torch


def timed(fn, reps):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def main():
    dev = torch.device("cuda", 0)
    rows = []
    for mb in (64, 128, 256, 512, 1024, 4096):
        n = mb * (1 << 20) // 16            # copy footprint = mb MiB (read + write)
        x = torch.randn(n, dtype=torch.float64, device=dev)
        y = torch.empty_like(x)
        z = torch.randn(n, dtype=torch.float64, device=dev)
        reps = max(20, 20000 // mb)
        t_copy = timed(lambda: y.copy_(x), reps)
        t_triad = timed(lambda: torch.add(x, z, alpha=0.5, out=y), reps)
        rows.append({"copy_footprint_MiB": mb, "copy_GBps": 16.0 * n / t_copy / 1e6, "triad_GBps": 24.0 * n / t_triad / 1e6})
        del x, y, z
    print(json.dumps({"device": torch.cuda.get_device_name(0), "rates": rows}, indent=1))


if __name__ == "__main__":
    main()
