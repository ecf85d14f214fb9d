#!/usr/bin/env python3
"""Turn rocprofv3 rocpd databases (ROCm 7 default output) into the small CSV summaries kept under profiles/.

usage: prof_summary.py <trace.db> <fetch.db> <write.db> <out_prefix> <bench.json> [<sq.db>]
  <out_prefix>_kernel_stats.csv : per kernel  calls, total/avg/min/max duration (us)   (rocprofv3 --kernel-trace --stats)
  <out_prefix>_pmc_fetch.csv    : per kernel  mean FETCH_SIZE (KB) per dispatch       (rocprofv3 --pmc FETCH_SIZE, own pass)
  <out_prefix>_pmc_write.csv    : per kernel  mean WRITE_SIZE (KB) per dispatch       (rocprofv3 --pmc WRITE_SIZE, own pass)
  <out_prefix>_traffic.json     : HBM bytes per launch of every kernel = (2 x FETCH_SIZE + WRITE_SIZE) x 1024
       (gfx950: FETCH_SIZE tallies 128-byte requests at 64 bytes -> doubled, MI355X_MICROARCH.md "HBM / rocprofv3")
  <out_prefix>_sq.csv           : per kernel  mean SQ counters per dispatch (own pass): wave cycles, instructions by kind, waits
"""
import csv, json, sqlite3, sys


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select name, count(*), sum(duration), avg(duration), min(duration), max(duration) from kernels "
                       "group by name order by sum(duration) desc").fetchall()
    return [(n, c, t / 1e3, a / 1e3, mi / 1e3, ma / 1e3) for n, c, t, a, mi, ma in rows]


def pmc(db, counter):
    cur = sqlite3.connect(db).cursor()
    rows = cur.execute("select kernel_name, count(*), avg(value), min(value), max(value) from counters_collection "
                       "where counter_name = ? group by kernel_name order by avg(value) desc", (counter,)).fetchall()
    return rows


def short(name):
    name = name.replace("void ", "")
    depth = 0
    for i in range(len(name) - 1, -1, -1):            # drop the trailing argument list
        if name[i] == ")": depth += 1
        elif name[i] == "(":
            depth -= 1
            if depth == 0: return name[:i]
    return name


def main():
    trace, fetch, write, prefix, bench = sys.argv[1:6]
    ks = kernel_stats(trace)
    with open(prefix + "_kernel_stats.csv", "w", newline="") as f:
        w = csv.writer(f); w.writerow(["kernel", "calls", "total_us", "avg_us", "min_us", "max_us"])
        for r in ks: w.writerow([r[0], r[1]] + [f"{x:.3f}" for x in r[2:]])
    fe = pmc(fetch, "FETCH_SIZE"); wr = pmc(write, "WRITE_SIZE")
    for rows, tag, cname in ((fe, "_pmc_fetch.csv", "FETCH_SIZE_KB"), (wr, "_pmc_write.csv", "WRITE_SIZE_KB")):
        with open(prefix + tag, "w", newline="") as f:
            w = csv.writer(f); w.writerow(["kernel", "dispatches", "avg_" + cname, "min_" + cname, "max_" + cname])
            for r in rows: w.writerow([r[0], r[1]] + [f"{x:.3f}" for x in r[2:]])
    cfg = json.load(open(bench))["config"]
    fmap = {r[0]: r[2] for r in fe}; wmap = {r[0]: r[2] for r in wr}
    out = {}
    for name in fmap:
        if "Mic" not in name: continue
        out[short(name).replace(", ", ",")] = {
            "hbm_bytes_per_launch": int((2.0 * fmap[name] + wmap.get(name, 0.0)) * 1024),
            "fetch_size_kb": fmap[name], "write_size_kb": wmap.get(name, 0.0),
            "frames_per_gpu": cfg["frames_per_gpu"], "width": cfg["width"], "height": cfg["height"],
            "depth": cfg["max_value"].bit_length(),
            "note": "2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes; gfx950 FETCH_SIZE counts 128-byte reads as 64"}
    json.dump(out, open(prefix + "_traffic.json", "w"), indent=1)
    for r in ks[:12]: print(f"{r[3]:10.1f} us avg  x{r[1]:3d}  {r[0][:90]}")
    if len(sys.argv) > 6:
        cur = sqlite3.connect(sys.argv[6]).cursor()
        rows = cur.execute("select kernel_name, counter_name, count(*), avg(value) from counters_collection "
                           "group by kernel_name, counter_name").fetchall()
        names = sorted({r[1] for r in rows})
        table = {}
        for k, c, n, v in rows: table.setdefault(k, {})[c] = (n, v)
        with open(prefix + "_sq.csv", "w", newline="") as f:
            w = csv.writer(f); w.writerow(["kernel", "dispatches"] + names)
            for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", (0, 0))[1]):
                w.writerow([short(k).replace(", ", ","), max(v[0] for v in table[k].values())] + [f"{table[k].get(c, (0, 0))[1]:.0f}" for c in names])


if __name__ == "__main__":
    main()
