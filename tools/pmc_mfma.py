"""MFMA (matrix-core) utilisation per kernel from one rocprofv3 counter pass (SQ_VALU_MFMA_BUSY_CYCLES, SQ_BUSY_CYCLES,
GRBM_GUI_ACTIVE) of the bench command:

    python tools/pmc_mfma.py <counter_collection.csv> profiles/rNN_mfma_util.csv profiles/rNN_mfma_util.json

mfma_util = SQ_VALU_MFMA_BUSY_CYCLES / (kernel cycles x 1024 SIMDs), kernel cycles = GRBM_GUI_ACTIVE / 8 (rocprofv3 reports
the sum over the 8 XCDs; MI355X_MICROARCH.md, DVFS note).  SQ_VALU_MFMA_BUSY_CYCLES counts cycles of every SIMD's matrix
pipe (32 per v_mfma_f32_32x32x16_bf16), summed over the chip.  Kernels are grouped by name (hipBLASLt: the Cijk_ tile
name up to the macro-tile field; own kernels: xfm::<name><template args>)."""
import collections
import csv
import json
import re
import sys


def key_of(name):
    m = re.search(r"xfm::(\w+(?:<[^>]*>)?)", name)
    if m:
        return m.group(1)
    m = re.match(r"(Cijk_\w+?_MT\d+x\d+x\d+)", name)
    if m:
        return m.group(1)
    return None


def main():
    path, out_csv, out_json = sys.argv[1:4]
    acc = collections.defaultdict(collections.Counter)
    n = collections.Counter()
    for row in csv.DictReader(open(path)):
        k = key_of(row["Kernel_Name"])
        if k is None:
            continue
        acc[k][row["Counter_Name"]] += float(row["Counter_Value"])
        if row["Counter_Name"] == "GRBM_GUI_ACTIVE":
            n[k] += 1
    rows = []
    for k, c in acc.items():
        cyc = c["GRBM_GUI_ACTIVE"] / 8.0
        if cyc <= 0:
            continue
        rows.append((c["SQ_VALU_MFMA_BUSY_CYCLES"] / (cyc * 1024.0), k, n[k], cyc / max(n[k], 1), c["SQ_VALU_MFMA_BUSY_CYCLES"] / max(n[k], 1)))
    rows.sort(key=lambda r: -r[3] * r[2])
    js = {}
    with open(out_csv, "w") as f:
        f.write("kernel,launches,avg_kernel_cycles,avg_mfma_busy_cycles_chip,mfma_util\n")
        for u, k, cnt, cyc, busy in rows:
            if busy <= 0:
                continue
            f.write(f"\"{k}\",{cnt},{cyc:.0f},{busy:.0f},{u:.4f}\n")
            js[k] = {"launches": cnt, "avg_kernel_cycles": round(cyc), "mfma_util": round(u, 4)}
    json.dump(js, open(out_json, "w"), indent=1)
    for u, k, cnt, cyc, busy in rows[:25]:
        if busy > 0:
            print(f"{k[:70]:70s} launches {cnt:5d}  avg cycles {cyc:9.0f}  mfma_util {u:.3f}")


if __name__ == "__main__":
    main()
