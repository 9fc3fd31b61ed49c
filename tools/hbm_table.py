"""HBM GB/s per kernel of the bench step: counter traffic per launch (profiles/<tag>_pmc_traffic.json: FETCH_SIZE x 2 + WRITE_SIZE from two
separate rocprofv3 --pmc passes) over the average launch duration of the serial kernel-trace pass (profiles/<tag>_final_bench_serial_kernel_stats.csv).
usage: python tools/hbm_table.py [tag] > profiles/<tag>_hbm_gbps_per_kernel.txt"""
import csv, json, re, sys, os
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r2"
traffic = json.load(open(os.path.join(R, "profiles", f"{tag}_pmc_traffic.json")))
stats = list(csv.DictReader(open(os.path.join(R, "profiles", f"{tag}_final_bench_serial_kernel_stats.csv"))))
sq = {r["kernel"]: r for r in json.load(open(os.path.join(R, "profiles", f"{tag}_pmc_sq_mfma.json")))}

def short(n):
    """kernel tag: conv3* as bench.py prints them, other kernels with every template argument"""
    m = re.match(r"_ZN2pu\d+([a-z0-9_]+)I(DF16_|NS_4bf16E|f)((?:L[ib]\d+E)*)", n)
    if not m: return n
    dt = {"DF16_": "f16", "NS_4bf16E": "bf16", "f": "f32"}[m.group(2)]
    toks = re.findall(r"L([ib])(\d+)E", m.group(3))
    if m.group(1).startswith("conv3"):          # bench.py's profiling tags: integers up to the bool, ",1" for the GroupNorm-backward epilogue
        nums = []
        for kind, v in toks:
            if kind == "b":
                if v == "1": nums.append("1")
                break
            nums.append(v)
    else:
        nums = [v for _, v in toks]             # every template argument (several instantiations of one GroupNorm kernel stay apart)
    return f"{m.group(1)}<{','.join([dt] + nums)}>"
dom = next(int(r["Calls"]) for r in stats if "conv3_kernel" in r["Name"] and "Li3ELi4ELi32ELi1ELi4ELi16" in r["Name"])
steps = dom / 80.0
rows = []
for r in stats:
    k = short(r["Name"])
    t = traffic.get(k)
    if not t:
        continue
    us = float(r["AverageNs"]) / 1e3
    mb = t["hbm_bytes_per_launch"] / 1e6
    rows.append((float(r["TotalDurationNs"]) / steps / 1e6, k, int(r["Calls"]) / steps, us, mb, mb / us * 1e3 / 1e3, sq.get(k, {}).get("mfma_util")))
rows.sort(reverse=True)
print(f"# {tag}: per kernel of one bench step (serial pass): ms per step, launches per step, average us, HBM MB per launch (counters), TB/s, share of the 8 TB/s peak, MFMA-busy")
print(f"{'kernel':58s} {'ms/step':>8s} {'n/step':>7s} {'us':>8s} {'MB':>8s} {'TB/s':>6s} {'of 8':>5s} {'MFMA':>5s}")
for ms, k, n, us, mb, tbs, mf in rows:
    if ms < 0.02:
        continue
    # rows above ~6.5 TB/s are not evidence of HBM rate: the blanket FETCH_SIZE x 2 correction over-counts narrow / Infinity-Cache-resident reads
    note = "  (> 6.5 TB/s: counter correction over-applies, not evidence)" if tbs > 6.5 else ""
    print(f"{k[:58]:58s} {ms:8.3f} {n:7.1f} {us:8.1f} {mb:8.1f} {tbs:6.2f} {tbs / 8:5.2f} {('%.2f' % mf) if mf is not None else '   -'}{note}")
