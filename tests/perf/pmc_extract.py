"""rocprofv3 (ROCm 7.2 default: rocpd sqlite) -> CSV rows of the PMC counters of one kernel.
usage: pmc_extract.py results.db [kernel-name-substring] > out.csv"""
import csv
import sqlite3
import sys

db = sys.argv[1]
pat = sys.argv[2] if len(sys.argv) > 2 else ""
con = sqlite3.connect(db)
cur = con.cursor()
w = csv.writer(sys.stdout)
w.writerow(["Dispatch_Id", "Kernel_Name", "Grid_Size", "Workgroup_Size", "LDS_Block_Size", "Scratch_Size",
            "VGPR_Count", "Accum_VGPR_Count", "SGPR_Count", "Counter_Name", "Counter_Value",
            "Start_Timestamp", "End_Timestamp", "Duration_ns"])
q = ("select dispatch_id, kernel_name, grid_size, workgroup_size, lds_block_size, scratch_size, vgpr_count, "
     "accum_vgpr_count, sgpr_count, counter_name, value, start, end from counters_collection "
     "where kernel_name like ? order by dispatch_id, counter_name")
for r in cur.execute(q, ("%" + pat + "%",)):
    w.writerow(list(r) + [r[12] - r[11]])
