"""Summarises a rocprofv3 --pmc counter_collection.csv: per kernel name, mean of each counter over the
dispatches.  With --json FILE the means are also merged into FILE ({kernel: {counter: mean}}), which
tools/pmc_to_json.py turns into profiles/traffic.json and profiles/valu.json."""
import collections
import csv
import json
import os
import sys

path = sys.argv[1]
jpath = sys.argv[sys.argv.index("--json") + 1] if "--json" in sys.argv else None
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(path)):
    name = r['Kernel_Name']
    if 'voc::' not in name:
        continue
    short = name.split('(')[0].replace('void voc::', '')
    acc[short][r['Counter_Name']].append(float(r['Counter_Value']))
merged = json.load(open(jpath)) if jpath and os.path.exists(jpath) else {}
for k, d in acc.items():
    print(k)
    # a counter is only meaningful against the clock of its OWN pass: every counter of this pass also keeps
    # the pass' GRBM_GUI_ACTIVE beside it (<counter>@GRBM_GUI_ACTIVE), since later passes overwrite the plain one
    gui = sum(d["GRBM_GUI_ACTIVE"]) / len(d["GRBM_GUI_ACTIVE"]) if "GRBM_GUI_ACTIVE" in d else None
    for cn, vals in d.items():
        print(f'   {cn:28s} mean {sum(vals)/len(vals):.4g}  (n={len(vals)})')
        merged.setdefault(k, {})[cn] = sum(vals) / len(vals)
        merged[k]["_dispatches_" + cn] = len(vals)
        if gui is not None and cn != "GRBM_GUI_ACTIVE":
            merged[k][cn + "@GRBM_GUI_ACTIVE"] = gui
if jpath:
    json.dump(merged, open(jpath, "w"), indent=1, sort_keys=True)
