"""profiles/pmc_traffic.json (what bench.py quotes as roofline.traffic) from the summary of the two separate --pmc passes of an evidence run
(tools/gpu_r5_profile.sh: rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE of `bench.py --eager --steps 1`, tools/rocpd_pmc.py -> <tag>_pmc.txt).
Per launch, dispatch-weighted over the first-sweep and steady instantiations of the elimination kernel; FETCH_SIZE doubled as MI355X_MICROARCH.md
prescribes for gfx950 (calibrated in round 1 on k_candidate_cost), WRITE_SIZE as reported, both in KB.
    python tools/make_pmc_traffic.py profiles/round5_v7_pmc.txt round5_v7"""
import json
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
path, tag = sys.argv[1], sys.argv[2]
cur, vals = None, {}
for line in open(path):
    m = re.match(r"\s+(FETCH_SIZE|WRITE_SIZE)\s+avg/dispatch\s+([0-9.]+)\s+dispatches\s+(\d+)", line)
    if m:
        vals.setdefault(cur, {})[m.group(1)] = (float(m.group(2)), int(m.group(3)))
    elif line.strip():
        cur = line.strip()


def weighted(pred, counter):
    num = den = 0.0
    for k, v in vals.items():
        if k and pred(k) and counter in v:
            num += v[counter][0] * v[counter][1]; den += v[counter][1]
    return num / den if den else None


elim = lambda k: "k_eliminate_grouped" in k
bs = lambda k: k.endswith("k_backsub")
f, w = weighted(elim, "FETCH_SIZE"), weighted(elim, "WRITE_SIZE")
fb, wb = weighted(bs, "FETCH_SIZE"), weighted(bs, "WRITE_SIZE")
old = json.load(open(os.path.join(ROOT, "profiles", "pmc_traffic.json")))
out = {"windows": 1024, "lines": 2000, "lba_elimination": 4,
       "kernel": "k_eliminate_grouped (first-sweep and steady launches of the profiled solves, dispatch-weighted)", "source": tag,
       "fetch_size_kb_per_launch": round(f, 1), "write_size_kb_per_launch": round(w, 1), "hbm_bytes_per_launch": (2.0 * f + w) * 1024.0,
       "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes (profiles/%s_pmc.txt), FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 reports half the bytes of wide coalesced reads; calibrated in round 1 on k_candidate_cost), WRITE_SIZE as reported; tools/make_pmc_traffic.py" % tag,
       "note": old.get("note", ""),
       "k_backsub": {"fetch_size_kb_per_launch": round(fb, 1), "write_size_kb_per_launch": round(wb, 1), "hbm_bytes_per_launch": (2.0 * fb + wb) * 1024.0}}
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_traffic.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
