"""Turns two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over scratch/one_frame.py into per-launch HBM
traffic per kernel.  FETCH_SIZE / WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE tallies 128-byte requests as 64 bytes,
so it is doubled (calibrated for wide coalesced reads; this kernel's 16-byte gathers are not separately calibrated)."""
import sys, csv, glob, json, collections

def per_kernel(d, counter):
    path = glob.glob(d + '/**/*counter_collection.csv', recursive=True)[0]
    tot = collections.defaultdict(float); n = collections.defaultdict(int)
    for r in csv.DictReader(open(path)):
        if r['Counter_Name'] != counter: continue
        k = r['Kernel_Name'].split('(')[0]
        tot[k] += float(r['Counter_Value']); n[k] += 1
    return tot, n

fetch, nf = per_kernel(sys.argv[1], 'FETCH_SIZE')
write, nw = per_kernel(sys.argv[2], 'WRITE_SIZE')
views = int(sys.argv[5]) if len(sys.argv) > 5 else 1
out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) over 3 " + ("passes of %d frames" % views if views > 1 else "frames (one at a time)") +
                 " of the bench.py workload " + (sys.argv[4] if len(sys.argv) > 4 else "reflective_bunnies_1080p"),
       "views": views,
       "correction": "FETCH_SIZE KiB x 1024 x 2 (gfx950 tallies 128-B requests as 64 B); WRITE_SIZE KiB x 1024", "kernels": {}}
for k in fetch:
    fb = fetch[k] * 1024 * 2 / nf[k]
    wb = write.get(k, 0.0) * 1024 / max(nw.get(k, 1), 1)
    out["kernels"][k] = {"launches": nf[k], "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb, "hbm_bytes_per_launch": fb + wb}
if "rt_traverse" in out["kernels"]:
    out["kernel"] = "rt_traverse"
    out["hbm_bytes_per_launch"] = out["kernels"]["rt_traverse"]["hbm_bytes_per_launch"]
json.dump(out, open(sys.argv[3], 'w'), indent=1)
print(json.dumps(out, indent=1))
