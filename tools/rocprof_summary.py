#!/usr/bin/env python
"""Summarise a rocprofv3 result database (rocprofv3 --kernel-trace --stats -d DIR -o NAME) as text for profiles/."""
import glob
import sqlite3
import sys


def main(path, out):
    db = glob.glob(path + "/*.db")[0] if not path.endswith(".db") else path
    cur = sqlite3.connect(db).cursor()
    lines = [f"# rocprofv3 kernel-trace summary of {db.split('/')[-1]}", "", "## top kernels (us)", "",
             "| kernel | calls | total_us | avg_us | % |", "|---|---|---|---|---|"]
    for name, calls, total, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        short = name if len(name) < 110 else name[:60] + " ... " + name[-40:]
        lines.append(f"| `{short}` | {calls} | {total:.1f} | {avg:.1f} | {pct:.2f} |")
    lines += ["", "## dispatches of pk:: kernels", "", "| kernel | duration_us | grid | wg | lds | scratch | vgpr | agpr | sgpr |", "|---|---|---|---|---|---|---|---|---|"]
    q = ("select name,duration,grid_x,workgroup_x,lds_size,scratch_size,vgpr_count,accum_vgpr_count,sgpr_count from kernels "
         "where name like '%pk::advect%' or name like '%pk::sort_key%' order by start")
    for r in cur.execute(q):
        lines.append(f"| `{r[0][:80]}` | {r[1] / 1e3:.1f} | {r[2]} | {r[3]} | {r[4]} | {r[5]} | {r[6]} | {r[7]} | {r[8]} |")
    open(out, "w").write("\n".join(lines) + "\n")
    print("\n".join(lines[:14]))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
