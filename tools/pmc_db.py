#!/usr/bin/env python3
"""Average PMC counter values per kernel from a rocprofv3 results database (`rocprofv3 --pmc ... -d DIR -o NAME` writes
DIR/**/NAME_results.db).  usage: pmc_db.py DB_OR_DIR [kernel-name substring] [--json]"""
import glob
import json
import os
import sqlite3
import sys


def read(path, sub=""):
    if os.path.isdir(path):
        path = sorted(glob.glob(os.path.join(path, "**", "*_results.db"), recursive=True))[0]
    db = sqlite3.connect(path)
    cur = db.cursor()
    tabs = [r[0] for r in cur.execute("select name from sqlite_master where type='table'")]
    t = lambda n: [x for x in tabs if x.startswith(n)][0]  # noqa: E731
    q = (f"select s.kernel_name, i.name, avg(p.value), count(*) from {t('rocpd_pmc_event')} p join {t('rocpd_info_pmc')} i "
         f"on p.pmc_id = i.id join {t('rocpd_kernel_dispatch')} k on p.event_id = k.event_id join "
         f"{t('rocpd_info_kernel_symbol')} s on k.kernel_id = s.id where s.kernel_name like ? group by s.kernel_name, i.name")
    out = {}
    for kname, cname, avg, n in cur.execute(q, (f"%{sub}%",)):
        out.setdefault(kname, {})[cname] = {"avg": avg, "samples": n}
    return out


if __name__ == "__main__":
    res = read(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 and not sys.argv[2].startswith("--") else "")
    if "--json" in sys.argv:
        print(json.dumps(res, indent=1))
    else:
        for k, v in res.items():
            print(k[:110])
            for c, r in sorted(v.items()):
                print(f"    {c:32s} {r['avg']:16.0f}  (n={r['samples']})")
