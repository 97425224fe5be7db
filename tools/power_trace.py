#!/usr/bin/env python3
"""Sample the GPU's power and clocks while a command runs (kernel-development evidence for DESIGN.md §8's power-limit paragraph).

    python tools/power_trace.py OUT.json -- <command ...>

Reads the amdgpu hwmon files of card 0 every ~50 ms (power1_average / power1_input, power1_cap, freq1_input = sclk, freq2_input = mclk,
temp*_input) and takes a `rocm-smi --showpower --showclocks --showuse --json` snapshot every 2 s; writes the samples, their summary over
the middle 80 % of the run (the command's start-up and tear-down excluded) and the command's own stdout tail."""
import glob
import json
import os
import subprocess
import sys
import time


def hwmon_files():
    out = {}
    for d in sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*")):
        for f in ("power1_average", "power1_input", "power1_cap", "freq1_input", "freq2_input", "temp1_input", "temp2_input", "temp3_input"):
            p = os.path.join(d, f)
            if os.path.exists(p):
                out.setdefault(d, {})[f] = p
    return out


def read_int(p):
    try:
        return int(open(p).read().strip())
    except Exception:
        return None


def smi():
    for cmd in (["rocm-smi", "--showpower", "--showclocks", "--showuse", "--showperflevel", "--json"], ["amd-smi", "metric", "--power", "--clock", "--usage", "--json"]):
        try:
            r = subprocess.run(cmd, capture_output=True, text=True, timeout=10)
            if r.returncode == 0 and r.stdout.strip():
                try:
                    return {"cmd": " ".join(cmd), "out": json.loads(r.stdout)}
                except Exception:
                    return {"cmd": " ".join(cmd), "out": r.stdout[-4000:]}
        except Exception:
            continue
    return None


def main():
    out_path = sys.argv[1]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    files = hwmon_files()
    dev = sorted(files)[0] if files else None
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    t0 = time.time()
    samples, snaps, last_snap = [], [], -10.0
    while proc.poll() is None:
        t = time.time() - t0
        if dev:
            samples.append({"t": round(t, 3), **{k: read_int(p) for k, p in files[dev].items()}})
        if t - last_snap >= 2.0:
            s = smi()
            if s:
                snaps.append({"t": round(t, 3), **s})
            last_snap = time.time() - t0
        time.sleep(0.05)
    tail = proc.stdout.read()[-3000:]
    dur = time.time() - t0
    mid = [s for s in samples if 0.1 * dur <= s["t"] <= 0.9 * dur]
    summ = {}
    for k in ("power1_average", "power1_input", "freq1_input", "freq2_input", "temp1_input", "temp2_input"):
        v = [s[k] for s in mid if s.get(k) is not None]
        if v:
            v.sort()
            summ[k] = {"mean": sum(v) / len(v), "p10": v[len(v) // 10], "median": v[len(v) // 2], "p90": v[len(v) * 9 // 10], "max": v[-1], "n": len(v)}
    cap = next((s.get("power1_cap") for s in samples if s.get("power1_cap")), None)
    json.dump({"command": cmd, "seconds": dur, "rc": proc.returncode, "hwmon": dev, "power_cap_uW": cap, "summary_mid80": summ, "stdout_tail": tail, "smi_snapshots": snaps, "samples": samples[:: max(1, len(samples) // 400)]}, open(out_path, "w"), indent=1)
    print(out_path, "rc", proc.returncode, "s %.1f" % dur, "cap", cap, {k: round(v["median"]) for k, v in summ.items()})


if __name__ == "__main__":
    main()
