#!/usr/bin/env python3
"""Matrix-pipe / VALU occupancy per kernel from one SQ counter pass (tools/gpu_profile.sh <tag> <config> sq).
usage: pmc_busy.py <prof_summary csv of the SQ pass> <out txt> [top N]
Per kernel (sorted by GPU time = GRBM_GUI_ACTIVE): launches, share of the pass's GPU time, and
  mfma_busy  = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs)   (kernel cycles = GRBM_GUI_ACTIVE / 8 XCDs; the counter
               counts cycles a SIMD's matrix pipe is busy: 32 per v_mfma_f32_32x32x16_bf16)
  valu/mfma  = SQ_INSTS_VALU / SQ_INSTS_MFMA (SQ_INSTS_VALU includes the MFMAs)
  wait_inst  = SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES   (issue stalls)      wait_any = SQ_WAIT_ANY / SQ_WAVE_CYCLES (s_waitcnt / barrier)"""
import csv
import sys


def main():
    src, dst = sys.argv[1], sys.argv[2]
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 16
    rows = list(csv.DictReader(open(src)))

    def f(r, k):
        return float(r.get(k + "_total", 0) or 0)
    rows = [r for r in rows if f(r, "GRBM_GUI_ACTIVE") > 0]
    tot = sum(f(r, "GRBM_GUI_ACTIVE") for r in rows)
    rows.sort(key=lambda r: -f(r, "GRBM_GUI_ACTIVE"))
    with open(dst, "w") as out:
        out.write("%-58s %8s %7s %10s %10s %10s %9s\n" % ("kernel", "launches", "time %", "mfma_busy", "valu/mfma", "wait_inst", "wait_any"))
        for r in rows[:top]:
            cyc = f(r, "GRBM_GUI_ACTIVE") / 8.0
            mf = f(r, "SQ_VALU_MFMA_BUSY_CYCLES") / (cyc * 1024.0) if cyc else 0.0
            nm = f(r, "SQ_INSTS_MFMA")
            wc = f(r, "SQ_WAVE_CYCLES")
            out.write("%-58s %8d %7.2f %10.3f %10s %10.3f %9.3f\n" % (
                r["kernel"][:58], int(float(r["launches"])), 100.0 * f(r, "GRBM_GUI_ACTIVE") / tot, mf,
                ("%.1f" % (f(r, "SQ_INSTS_VALU") / nm)) if nm else "-", f(r, "SQ_WAIT_INST_ANY") / wc if wc else 0.0,
                f(r, "SQ_WAIT_ANY") / wc if wc else 0.0))
    print(open(dst).read())


if __name__ == "__main__":
    main()
