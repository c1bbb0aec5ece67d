"""Condenses rocprofv3 rocpd databases (gpurun_out/prof*/..._results.db) into the text summaries kept under profiles/.

    python scripts/prof_summary.py <stats_db> [<pmc_db> ...] > profiles/rNN_rocprof_summary.txt
"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    m = re.match(r"(?:void )?([A-Za-z_0-9:]+)(<.*)?", name)
    base = m.group(1) if m else name
    targs = ""
    if m and m.group(2):
        t = m.group(2)
        t = t[: t.find(">(") + 1] if ">(" in t else t
        targs = t if len(t) < 60 else t[:57] + "...>"
    return (base + targs)[:100]


def stats(db):
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, total_calls, total_duration, average, percentage from top_kernels"))
    print(f"# kernel-trace --stats  ({db})")
    print(f"{'kernel':100s} {'calls':>7s} {'total_ms':>10s} {'avg_us':>10s} {'%':>6s}")
    for name, calls, tot, avg, pct in rows[:16]:
        print(f"{short(name):100s} {calls:7d} {tot / 1e3:10.2f} {avg:10.2f} {pct:6.2f}")
    print("\n# dispatch geometry (first dispatch of each ahip kernel)")
    seen = set()
    q = ("select name, grid_x, grid_y, grid_z, workgroup_x, lds_size, vgpr_count, accum_vgpr_count, sgpr_count, "
         "scratch_size from kernels where name like '%ahip%' order by start")
    for r in c.execute(q):
        k = short(r[0])
        if k in seen:
            continue
        seen.add(k)
        print(f"{k:100s} grid=({r[1]},{r[2]},{r[3]}) wg={r[4]} lds={r[5]} vgpr={r[6]} agpr={r[7]} sgpr={r[8]} scratch={r[9]}")


def by_grid(db, patterns=("sweep", "syrk_batch", "csr_axpy", "csr_tmul")):
    """Launches of one template differ by orders of magnitude in work (a full-design sweep and a 200-column one share a name):
    per (kernel, grid) class the call count and average duration, so that a per-launch roofline figure can be recomputed from
    this file for the class it was quoted on (config 4's sweep could not be; r4: every sweep-type template —
    sweep_kernel, sweep_snp_lut_kernel, csc_*sweep*, multi_sweep* — and the batched block builds, whose launches carry 1..16 blocks)."""
    c = sqlite3.connect(db)
    print("\n# sweep-type kernels by launch geometry (grid in threads; the largest grid of a template is the full design)")
    print(f"{'kernel':100s} {'grid':>22s} {'calls':>7s} {'avg_us':>10s} {'total_ms':>10s}")
    cond = " or ".join(f"name like '%{p}%'" for p in patterns)
    q = (f"select name, grid_x, grid_y, grid_z, count(*), avg(end - start), sum(end - start) from kernels where {cond} "
         "group by name, grid_x, grid_y, grid_z order by name, sum(end - start) desc")
    shown = {}
    for name, gx, gy, gz, n, av, tot in c.execute(q):
        k = short(name)
        shown[k] = shown.get(k, 0) + 1
        if shown[k] > 8:
            continue
        print(f"{k:100s} {f'({gx},{gy},{gz})':>22s} {n:7d} {av / 1e3:10.2f} {tot / 1e6:10.2f}")


def pmc(db):
    c = sqlite3.connect(db)
    print(f"\n# PMC pass ({db}): per-dispatch counter, averaged per kernel (value unit as reported by rocprofv3: KB)")
    q = ("select kernel_name, counter_name, count(*), avg(value), min(value), max(value), avg(duration) "
         "from counters_collection where kernel_name like '%ahip%' group by kernel_name, counter_name "
         "order by avg(value)*count(*) desc")
    print(f"{'kernel':100s} {'counter':>11s} {'n':>5s} {'avg_KB':>14s} {'min_KB':>14s} {'max_KB':>14s} {'avg_us':>10s}")
    for name, cn, n, av, mn, mx, dur in c.execute(q):
        print(f"{short(name):100s} {cn:>11s} {n:5d} {av:14.1f} {mn:14.1f} {mx:14.1f} {dur / 1e3:10.1f}")


if __name__ == "__main__":
    stats(sys.argv[1])
    try:
        by_grid(sys.argv[1])
    except sqlite3.Error as e:  # (older rocpd schemas)
        print(f"# by_grid: {e}")
    for db in sys.argv[2:]:
        pmc(db)
