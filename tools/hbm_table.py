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
    """the tag tools/pmc_summary.py gives the same kernel: _ZN2pu12conv3_kernelIDF16_Li3E...E -> conv3_kernel<f16,3,...>; others verbatim"""
    m = re.match(r"_ZN2pu\d+([a-z0-9_]+)I(DF16_|NS_4bf16E|f)((?:Li\d+E)*)", n)
    if not m:
        return n
    dt = {"DF16_": "f16", "NS_4bf16E": "bf16", "f": "f32"}[m.group(2)]
    nums = re.findall(r"Li(\d+)E", m.group(3))
    return m.group(1) + "<" + ",".join([dt] + nums) + ">"

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
    print(f"{k[:58]:58s} {ms:8.3f} {n:7.1f} {us:8.1f} {mb:8.1f} {tbs:6.2f} {tbs / 8:5.2f} {('%.2f' % mf) if mf is not None else '   -'}")
