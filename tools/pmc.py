"""Aggregate rocprofv3 --pmc counter_collection.csv files per kernel (mean over dispatches).
usage: python tools/pmc.py gpurun_out/<tag>   (expects pmc_sq/, pmc_fetch/, pmc_write/ below it)"""
import csv, glob, sys, collections
root = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for sub in ("pmc_sq", "pmc_fetch", "pmc_write"):
    for f in glob.glob(f"{root}/{sub}/*/*_counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
            if sub == "pmc_sq" and r["Counter_Name"] == "SQ_WAVES":
                agg[k]["dur_us"].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
                agg[k]["vgpr"].append(float(r["VGPR_Count"])); agg[k]["lds"].append(float(r["LDS_Block_Size"]))
def m(v): return sum(v) / len(v) if v else float("nan")
rows = []
for k, c in agg.items():
    if "dur_us" not in c: continue
    rows.append((m(c["dur_us"]) * len(c["dur_us"]), k, c))
rows.sort(reverse=True)
print(f"{'kernel':58s} {'n':>4s} {'us':>7s} {'vgpr':>4s} {'ldsKB':>5s} {'waves':>7s} {'mfma%':>6s} {'wait%':>6s} {'wInst%':>6s} {'act%':>5s} {'ldsCf%':>6s} {'rdMB':>7s} {'wrMB':>7s} {'GB/s':>6s}")
for tot, k, c in rows[:int(sys.argv[2]) if len(sys.argv) > 2 else 24]:
    d = m(c["dur_us"]); wc = m(c["SQ_WAVE_CYCLES"]); gui = m(c["GRBM_GUI_ACTIVE"])
    # SQ_* cycle counters count quad-cycles summed over waves; MFMA busy counts cycles summed over SIMDs
    mfma = m(c["SQ_VALU_MFMA_BUSY_CYCLES"]) / (gui / 8 * 1024) * 100 if gui else float("nan")   # 1024 SIMDs; gui summed over 8 XCDs
    rd = m(c.get("FETCH_SIZE", [])) * 1024 * 2 / 1e6   # KB -> bytes; x2 gfx950 correction for wide streaming reads
    wr = m(c.get("WRITE_SIZE", [])) * 1024 / 1e6
    print(f"{k.replace('(anonymous namespace)::','')[:58]:58s} {len(c['dur_us']):4d} {d:7.1f} {m(c['vgpr']):4.0f} {m(c['lds'])/1024:5.1f} {m(c['SQ_WAVES']):7.0f} {mfma:6.1f} "
          f"{100*m(c['SQ_WAIT_ANY'])/wc:6.1f} {100*m(c['SQ_WAIT_INST_ANY'])/wc:6.1f} {100*m(c['SQ_ACTIVE_INST_ANY'])/wc:5.1f} {100*m(c['SQ_LDS_BANK_CONFLICT'])/max(m(c['SQ_BUSY_CYCLES']),1):6.1f} {rd:7.1f} {wr:7.1f} {(rd+wr)/d*1e3/1e3:6.0f}")
