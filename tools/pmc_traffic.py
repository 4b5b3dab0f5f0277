"""Turn two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: separate runs, the TCC block cannot hold both) into a
per-kernel HBM-traffic summary.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports exactly half of the
bytes of wide coalesced streaming reads -> doubled here; WRITE_SIZE is exact for 16-B stores.  Counter unit: KiB."""
import collections, csv, glob, json, sys

def agg(path, counter):
    f = glob.glob(path + '/*/*counter_collection.csv')[0]
    d = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(f)):
        if r['Counter_Name'] != counter:
            continue
        k = r['Kernel_Name'].replace('void ', '').replace('(anonymous namespace)::', '').split('(')[0]
        d[k][0] += 1
        d[k][1] += float(r['Counter_Value'])
    return d

fetch_dir, write_dir, out = sys.argv[1], sys.argv[2], sys.argv[3]
fe, wr = agg(fetch_dir, 'FETCH_SIZE'), agg(write_dir, 'WRITE_SIZE')
res = {}
for k, (n, v) in fe.items():
    nw, w = wr.get(k, [n, 0.0])
    res[k] = {"launches_in_trace": n, "fetch_bytes_per_launch_raw": v / n * 1024,
              "fetch_bytes_per_launch_x2": 2 * v / n * 1024, "write_bytes_per_launch": w / max(nw, 1) * 1024,
              "hbm_bytes_per_launch": (2 * v / n + w / max(nw, 1)) * 1024}
json.dump({"note": "FETCH_SIZE doubled (gfx950 wide-read correction), WRITE_SIZE as is; KiB -> bytes",
           "command": "rocprofv3 --kernel-trace --pmc FETCH_SIZE|WRITE_SIZE -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-roofline",
           "kernels": res}, open(out, 'w'), indent=1, sort_keys=True)
print("wrote", out, len(res), "kernels")
