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


# A kernel name is launched with and without work in one step (decode classes that hold no unit of the batch, the second launch of a
# two-class kernel): an average over ALL dispatches of a name halves the figures of a kernel whose sibling launch is empty (round 3's
# traffic table did).  Every summary here is over the dispatches that did work: those within a factor 100 of the name's largest.
def _working(vals):
    top = max(vals) if vals else 0
    return [v for v in vals if v * 100 >= top] or vals


def kernel_stats(db):
    cur = sqlite3.connect(db).cursor()
    per = {}
    for n, d in cur.execute("select name, duration from kernels"):
        per.setdefault(n, []).append(d)
    out = []
    for n, ds in per.items():
        w = _working(ds)
        out.append((n, len(w), sum(w) / 1e3, sum(w) / len(w) / 1e3, min(w) / 1e3, max(w) / 1e3, len(ds) - len(w)))
    return sorted(out, key=lambda r: -r[2])


def pmc(db, counter):
    cur = sqlite3.connect(db).cursor()
    per = {}
    for n, v in cur.execute("select kernel_name, value from counters_collection where counter_name = ?", (counter,)):
        per.setdefault(n, []).append(v)
    out = []
    for n, vs in per.items():
        w = _working(vs)
        out.append((n, len(w), sum(w) / len(w), min(w), max(w), len(vs) - len(w)))
    return sorted(out, key=lambda r: -r[2])


def timeline(db, path, last=160):
    """the last `last` kernels in launch order: start (us, from the first of them), duration, gap to the kernel before -- where a
    step's time goes that is in no kernel (host round trips, empty launches)"""
    con = sqlite3.connect(db); cur = con.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    s_col = next((c for c in ("start", "start_timestamp", "begin") if c in cols), None)
    e_col = next((c for c in ("end", "end_timestamp") if c in cols), None)
    if not s_col or not e_col:
        return
    rows = cur.execute(f"select name, {s_col}, {e_col} from kernels order by {s_col}").fetchall()[-last:]
    if not rows:
        return
    t0 = rows[0][1]; prev_end = None; busy = 0
    with open(path, "w", newline="") as f:
        w = csv.writer(f); w.writerow(["kernel", "start_us", "duration_us", "gap_before_us"])
        for n, st, en in rows:
            w.writerow([short(n).replace(", ", ","), f"{(st - t0) / 1e3:.1f}", f"{(en - st) / 1e3:.1f}", f"{(st - prev_end) / 1e3:.1f}" if prev_end is not None else ""])
            prev_end = en if prev_end is None else max(prev_end, en); busy += en - st
        w.writerow(["(span, busy, not in a kernel)", f"{(prev_end - t0) / 1e3:.1f}", f"{busy / 1e3:.1f}", f"{(prev_end - t0 - busy) / 1e3:.1f}"])


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
        w = csv.writer(f); w.writerow(["kernel", "calls_with_work", "total_us", "avg_us", "min_us", "max_us", "empty_calls_left_out"])
        for r in ks: w.writerow([r[0], r[1]] + [f"{x:.3f}" for x in r[2:6]] + [r[6]])
    fe = pmc(fetch, "FETCH_SIZE"); wr = pmc(write, "WRITE_SIZE")
    for rows, tag, cname in ((fe, "_pmc_fetch.csv", "FETCH_SIZE_KB"), (wr, "_pmc_write.csv", "WRITE_SIZE_KB")):
        with open(prefix + tag, "w", newline="") as f:
            w = csv.writer(f); w.writerow(["kernel", "dispatches_with_work", "avg_" + cname, "min_" + cname, "max_" + cname, "empty_dispatches_left_out"])
            for r in rows: w.writerow([r[0], r[1]] + [f"{x:.3f}" for x in r[2:5]] + [r[5]])
    cfg = json.load(open(bench))["config"]
    fmap = {r[0]: r[2] for r in fe}; wmap = {r[0]: r[2] for r in wr}
    fmax = {r[0]: r[4] for r in fe}; wmax = {r[0]: r[4] for r in wr}
    timeline(trace, prefix + "_timeline.csv")
    out = {}
    for name in fmap:
        if "Mic" not in name: continue
        out[short(name).replace(", ", ",")] = {
            "hbm_bytes_per_launch": int((2.0 * fmap[name] + wmap.get(name, 0.0)) * 1024),
            "hbm_bytes_per_launch_max": int((2.0 * fmax[name] + wmax.get(name, 0.0)) * 1024),
            "fetch_size_kb": fmap[name], "write_size_kb": wmap.get(name, 0.0),
            "frames_per_gpu": cfg["frames_per_gpu"], "width": cfg["width"], "height": cfg["height"],
            "depth": cfg["max_value"].bit_length(),
            "note": "2 x FETCH_SIZE + WRITE_SIZE, KB -> bytes, mean over the dispatches that did work; gfx950 FETCH_SIZE counts 128-byte reads as 64"}
    # the build these counters belong to: bench.py looks a kernel's traffic up here only while csrc/ still hashes to this
    import hashlib, os
    d = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "medical-image-codec_amd", "csrc")
    hsh = hashlib.sha256()
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".h")):
            hsh.update(name.encode()); hsh.update(open(os.path.join(d, name), "rb").read())
    out["_meta"] = {"csrc_sha16": hsh.hexdigest()[:16]}
    json.dump(out, open(prefix + "_traffic.json", "w"), indent=1)
    for r in ks[:14]: print(f"{r[3]:10.1f} us avg  x{r[1]:3d}  {r[0][:90]}")
    if len(sys.argv) > 6:
        cur = sqlite3.connect(sys.argv[6]).cursor()
        per = {}
        for k, c, v in cur.execute("select kernel_name, counter_name, value from counters_collection"):
            per.setdefault((k, c), []).append(v)
        names = sorted({c for _, c in per})
        table = {}
        for (k, c), vs in per.items():
            w_ = _working(vs)
            table.setdefault(k, {})[c] = (len(w_), sum(w_) / len(w_))
        with open(prefix + "_sq.csv", "w", newline="") as f:
            w = csv.writer(f); w.writerow(["kernel", "dispatches"] + names)
            for k in sorted(table, key=lambda k: -table[k].get("SQ_WAVE_CYCLES", (0, 0))[1]):
                w.writerow([short(k).replace(", ", ","), max(v[0] for v in table[k].values())] + [f"{table[k].get(c, (0, 0))[1]:.0f}" for c in names])


if __name__ == "__main__":
    main()
