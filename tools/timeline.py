"""Prints the kernels of the LAST suffix-array build in a rocprofv3 kernel trace in launch order, with the idle gap in
front of each one: where the build's wall time goes besides its kernels (host synchronisations, launch latency).
usage: timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, os, re, sys
f = glob.glob(os.path.join(sys.argv[1], "**", "*kernel_trace.csv"), recursive=True)[0]
rows = []
for r in csv.DictReader(open(f)):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# last build: from the last sa_bytecount to the first q_ kernel after it
starts = [i for i, r in enumerate(rows) if "sa_bytecount" in r[2]]
i0 = starts[-1]
i1 = next((i for i in range(i0, len(rows)) if re.search(r"\bq_", rows[i][2])), len(rows))
prev_end = rows[i0][0]
tot_k = tot_gap = 0
for s, e, name in rows[i0:i1]:
    short = re.sub(r"<.*", "", name.replace("cdb::", "").replace("(anonymous namespace)::", ""))
    m = re.search(r"cdb::(TextGen|NoGen), ([\w: ]+?)(?:, cdb::(\w+))?>", name)
    if m: short += f"[{m.group(1)},{m.group(2).replace('unsigned ', 'u').replace('cdb::', '')},{m.group(3)}]"
    gap = (s - prev_end) / 1e3
    print(f"{gap:9.1f} us gap  {(e - s) / 1e3:9.1f} us  {short}")
    tot_k += e - s; tot_gap += max(0, s - prev_end); prev_end = max(prev_end, e)
print(f"kernels {tot_k / 1e6:.3f} ms, gaps {tot_gap / 1e6:.3f} ms, span {(prev_end - rows[i0][0]) / 1e6:.3f} ms")
