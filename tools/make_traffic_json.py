"""profiles/<tag>_pmc_fetch_write_summary.csv -> profiles/<round>_pmc_traffic.json (what bench.py reports as roofline.traffic).
HBM bytes per launch = 2 x FETCH_SIZE + WRITE_SIZE: on gfx950 FETCH_SIZE tallies 64 B per 128-B request of wide coalesced reads
(MI355X_MICROARCH.md, HBM section), WRITE_SIZE is exact; both counters come from separate --pmc passes (tools/collect_profiles.sh).
usage: make_traffic_json.py <summary.csv> <commit> [out.json]"""
import csv, json, sys
rows = [r for r in csv.reader(l for l in open(sys.argv[1]) if not l.startswith("#"))]
hdr, rows = rows[0], rows[1:]
fi, wi = hdr.index("mean_FETCH_SIZE"), hdr.index("mean_WRITE_SIZE")
# (kernel name prefixes: the tile kernel gained a third template argument in round 3)
names = {"edgeblock_bwd_kernel<0, 8": "edgeblock_bwd_conv4", "edgeblock_bwd_kernel<0, 4": "edgeblock_bwd_conv3",
         "edgeblock_bwd_kernel<0, 2": "edgeblock_bwd_conv2", "edgeblock_fwd_kernel<2, false>": "edgeblock_fwd_conv4",
         "mfma_tn_tern_kernel<5, true>": "edgeblock_wgrad", "mfma_tn_aff2_kernel": "edgeblock_wgrad_conv4", "edgeblock_bwd_gather_kernel<3, false, false>": "edgeblock_gather_conv4"}
def key_of(kernel):
    for pre, key in names.items():
        if kernel.startswith(pre):
            return key
    return None
out = {"commit": sys.argv[2], "source": sys.argv[1], "unit": "bytes per launch (mean)", "formula": "2*FETCH_SIZE + WRITE_SIZE (KB -> B)", "kernels": {}}
for r in rows:
    if key_of(r[0]) and r[fi] and r[wi]:
        f, w = float(r[fi]) * 1e3, float(r[wi]) * 1e3
        out["kernels"][key_of(r[0])] = {"kernel": r[0], "launches": int(r[1]), "fetch_size_bytes": f, "write_size_bytes": w, "traffic_bytes": 2 * f + w}
json.dump(out, open(sys.argv[3] if len(sys.argv) > 3 else "profiles/r05_pmc_traffic.json", "w"), indent=1)
print(json.dumps(out["kernels"].get("edgeblock_bwd_conv4")))
