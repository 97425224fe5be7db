"""pmc_csv_summary.py's text -> the markdown table kept under profiles/ (MfmaUtil, wait share, FETCH/WRITE bytes, L2 hit rate).
    python tools/pmc_table.py gpurun_out/pmcfull_summary.txt "Round 1, run J" [profiles/r01_j_bench_bf16_kernel_stats.md] > profiles/r01_j_pmc_bf16.md
With the kernel-stats table of the same build (rocprofv3 --kernel-trace --stats) a last column gives HBM GB/s = (FETCH x2 + WRITE) / the
kernel's average duration there (counters and durations come from separate passes of the same command, as the guide prescribes)."""
import re
import sys


def main():
    txt = open(sys.argv[1]).read().splitlines()
    title = sys.argv[2] if len(sys.argv) > 2 else "PMC"
    avg_us = {}
    if len(sys.argv) > 3:
        for ln in open(sys.argv[3]):
            m = re.match(r"\| `([^`]+)` \| (\d+) \| ([0-9.]+) \| ([0-9.]+) \|", ln)
            if m:
                avg_us[m.group(1)[:36]] = float(m.group(4))
    kern, cur = {}, None
    for ln in txt:
        if not ln.startswith(" "):
            cur = ln.strip()
            kern[cur] = {}
        else:
            m = re.match(r"\s+(\S+)\s+n=\s*(\d+)\s+mean=(\S+)", ln)
            if m and cur:
                kern[cur][m.group(1)] = (int(m.group(2)), float(m.group(3)))
    rows = []
    for k, c in kern.items():
        if "SQ_VALU_MFMA_BUSY_CYCLES" not in c or "GRBM_GUI_ACTIVE" not in c:
            continue
        g = lambda n: c.get(n, (0, 0.0))[1]
        util = g("SQ_VALU_MFMA_BUSY_CYCLES") / (g("GRBM_GUI_ACTIVE") / 8 * 1024) if g("GRBM_GUI_ACTIVE") else 0.0
        wait = g("SQ_WAIT_ANY") / g("SQ_WAVE_CYCLES") if g("SQ_WAVE_CYCLES") else 0.0
        hit = g("TCC_HIT_sum") / (g("TCC_HIT_sum") + g("TCC_MISS_sum")) if g("TCC_HIT_sum") + g("TCC_MISS_sum") else 0.0
        name = k.replace("void vtts::", "").replace("vtts::", "")
        fg, wg = g("FETCH_SIZE") * 1024 * 2 / 1e9, g("WRITE_SIZE") * 1024 / 1e9
        us = avg_us.get(name[:36])
        rows.append((util, name[:54], c["SQ_VALU_MFMA_BUSY_CYCLES"][0], wait, g("FETCH_SIZE"), fg, wg, hit, g("SQ_LDS_BANK_CONFLICT"),
                     (fg + wg) / (us * 1e-6) if us else None))
    rows.sort(reverse=True)
    print(f"# {title} — PMC passes (`rocprofv3 --kernel-trace --pmc ...`, one counter set per pass, CSV) on `bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-rtf --no-f32` (bf16, B=64 x T=1024)\n")
    print("MfmaUtil = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE/8 x 1024 SIMDs); FETCH_SIZE / WRITE_SIZE in KiB per launch as reported (`x2`: the gfx950 correction for 16-byte-per-lane streams, MI355X_MICROARCH.md §HBM).\n")
    print("| kernel | launches | MfmaUtil | wait_any/wave_cycles | FETCH_SIZE KiB (raw) | x2 -> GB | WRITE_SIZE -> GB | L2 hit rate | LDS bank-conflict cycles | HBM GB/s |")
    print("|---|---:|---:|---:|---:|---:|---:|---:|---:|---:|")
    for u, n, l, w, f, fg, wg, h, bc, bw in rows:
        print(f"| `{n}` | {l} | {u:.3f} | {w:.2f} | {f:.0f} | {fg:.2f} | {wg:.2f} | {h:.2f} | {bc:.0f} | {bw:.0f} |" if bw else
              f"| `{n}` | {l} | {u:.3f} | {w:.2f} | {f:.0f} | {fg:.2f} | {wg:.2f} | {h:.2f} | {bc:.0f} | |")


if __name__ == "__main__":
    main()
