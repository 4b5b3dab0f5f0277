"""Per-kernel matrix-pipe utilisation from one rocprofv3 --pmc pass (SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES
GRBM_GUI_ACTIVE).  Units (MI355X_MICROARCH.md, cycle constants): SQ_VALU_MFMA_BUSY_CYCLES counts cycles (16 per
v_mfma_f32_16x16x32_bf16, 32 per v_mfma_f32_16x16x4_f32), summed over the chip's SIMDs; GRBM_GUI_ACTIVE is summed over the 8
XCDs.  mfma_util = MFMA_BUSY / (GUI_ACTIVE / 8 * 1024 SIMDs): the share of SIMD-cycles in which the matrix pipe was busy."""
import collections, csv, glob, sys

path = sys.argv[1]
agg = collections.defaultdict(lambda: collections.defaultdict(float))
calls = collections.Counter()
for f in glob.glob(path + '/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        agg[k][r['Counter_Name']] += float(r['Counter_Value'])
        if r['Counter_Name'] == 'GRBM_GUI_ACTIVE':
            calls[k] += 1
print(f"{'kernel':64s} {'calls':>6s} {'mfma_busy/launch':>17s} {'gui_active/8/launch':>20s} {'mfma_util':>10s}")
rows = []
for k, v in agg.items():
    n = max(1, calls[k])
    busy, gui = v.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), v.get('GRBM_GUI_ACTIVE', 0.0) / 8.0
    util = busy / (gui * 1024.0) if gui else 0.0
    rows.append((busy, k, n, busy / n, gui / n, util))
for busy, k, n, b, g, u in sorted(rows, reverse=True)[:30]:
    print(f"{k[:64]:64s} {n:6d} {b:17.4g} {g:20.4g} {u:10.4f}")
