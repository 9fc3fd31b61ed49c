"""profiles/r1_pmc_sq_mfma.json from one rocprofv3 --pmc pass of SQ counters over bench.py (PU_NO_SIDE_STREAM=1):
  SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT
  SQ_LDS_IDX_ACTIVE GRBM_GUI_ACTIVE
usage: python tools/pmc_sq_summary.py <counter_collection.csv> <kernel_trace.csv> <out.json>
MFMA utilisation = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) (MI355X_MICROARCH.md: the counter
counts 32 cycles per 32x32x16 MFMA; GRBM_GUI_ACTIVE is summed over the 8 XCDs)."""
import csv, collections, re, json, sys
acc = collections.defaultdict(lambda: collections.defaultdict(float)); cnt = collections.Counter(); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r['Kernel_Name']; acc[k][r['Counter_Name']] += float(r['Counter_Value'])
    if r['Dispatch_Id'] not in seen:
        seen.add(r['Dispatch_Id']); cnt[k] += 1
dur = collections.defaultdict(float)
for r in csv.DictReader(open(sys.argv[2])):
    dur[r['Kernel_Name']] += int(r['End_Timestamp']) - int(r['Start_Timestamp'])
def tag(m):
    mm = re.match(r"_ZN2pu\d+([a-z0-9_]+)I(DF16_|NS_4bf16E|f)((?:Li\d+E)*)((?:Lb[01]E)?)", m)
    if not mm: return m[:60]
    nums = re.findall(r"Li(\d+)E", mm.group(3))
    return mm.group(1) + "<" + ",".join(["f16"] + nums + (["1"] if mm.group(4) == "Lb1E" else [])) + ">"
rows = []
for k, c in acc.items():
    if c.get('SQ_VALU_MFMA_BUSY_CYCLES', 0) <= 0: continue
    gui = c['GRBM_GUI_ACTIVE'] / 8.0; wc = c['SQ_WAVE_CYCLES']
    rows.append(dict(kernel=tag(k), launches=cnt[k], ms=round(dur[k] / 1e6, 3), mfma_util=round(c['SQ_VALU_MFMA_BUSY_CYCLES'] / (gui * 1024), 4),
                     wait_any=round(c['SQ_WAIT_ANY'] / wc, 3), wait_inst_any=round(c['SQ_WAIT_INST_ANY'] / wc, 3),
                     active_inst_any=round(c['SQ_ACTIVE_INST_ANY'] / wc, 3), wait_inst_lds=round(c['SQ_WAIT_INST_LDS'] / wc, 3),
                     lds_conflict_frac=round(c['SQ_LDS_BANK_CONFLICT'] / max(c['SQ_LDS_IDX_ACTIVE'], 1), 4),
                     eff_clock_ghz=round(gui / (dur[k] / 1e9) / 1e9, 3) if dur[k] else None))
    # GRBM_GUI_ACTIVE also counts cycles outside the kernel's start / end timestamps: for short kernels the derived clock exceeds the 2.4 GHz
    # part and the utilisation of that row is normalised wrongly - flagged, not evidence
    rows[-1]['normalisation_ok'] = bool(rows[-1]['eff_clock_ghz'] is not None and rows[-1]['eff_clock_ghz'] <= 2.45)
rows.sort(key=lambda r: -r['ms'])
json.dump(rows, open(sys.argv[3], 'w'), indent=1)
for r in rows[:14]: print(r)
