"""Summarise a rocprofv3 (rocpd sqlite) kernel trace into a small markdown table for profiles/.
usage: python tools/rocpd_stats.py <results.db> <out.md> "<command that was profiled>" """
import sqlite3
import sys

db, out, cmd = sys.argv[1], sys.argv[2], sys.argv[3]
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute(
    "select name, count(*), sum(duration), avg(duration), min(duration), max(duration), "
    "max(vgpr_count), max(accum_vgpr_count), max(sgpr_count), max(lds_size), max(grid_x), max(workgroup_x) "
    "from kernels group by name order by sum(duration) desc"))
tot = sum(r[2] for r in rows) or 1
with open(out, "w") as f:
    f.write(f"# rocprofv3 --kernel-trace --stats summary\n\ncommand: `{cmd}`\n\n")
    f.write("| kernel | calls | total ms | avg us | min us | max us | % | vgpr | agpr | sgpr | lds B | grid_x | wg_x |\n")
    f.write("|---|---|---|---|---|---|---|---|---|---|---|---|---|\n")
    for r in rows:
        name = r[0] if len(r[0]) < 90 else r[0][:87] + "..."
        f.write(f"| `{name}` | {r[1]} | {r[2]/1e6:.3f} | {r[3]/1e3:.2f} | {r[4]/1e3:.2f} | {r[5]/1e3:.2f} | "
                f"{100*r[2]/tot:.2f} | {r[6]} | {r[7]} | {r[8]} | {r[9]} | {r[10]} | {r[11]} |\n")
print(open(out).read())
