"""profiles/r1_pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.
usage: python tools/pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <out.json>"""
import csv, collections, json, re, sys

def agg(path):
    acc = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][0] += 1; acc[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return acc

def tag(mangled):
    """kernel tag: conv3* as bench.py prints them, other kernels with every template argument"""
    m = re.match(r"_ZN2pu\d+([a-z0-9_]+)I(DF16_|NS_4bf16E|f)((?:L[ib]\d+E)*)", mangled)
    if not m: return mangled
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
f, w = agg(sys.argv[1]), agg(sys.argv[2])
out = {}
for k in f:
    n = f[k][0]
    fetch_kib, write_kib = f[k][1] / n, (w[k][1] / w[k][0]) if k in w else 0.0
    out[tag(k)] = dict(launches=n, fetch_size_kib_per_launch=round(fetch_kib, 1), write_size_kib_per_launch=round(write_kib, 1),
                       hbm_bytes_per_launch=round((2 * fetch_kib + write_kib) * 1024),
                       note="FETCH_SIZE doubled (gfx950 reports half of a wide coalesced stream, MI355X_MICROARCH.md); separate --pmc passes")
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
print(len(out), "kernels")
