"""Print the kernel timeline of the middle of a rocprofv3 kernel trace (tools/trace_only.sh output)."""
import csv, glob, os, sys
d = sys.argv[1]
n_show = int(sys.argv[2]) if len(sys.argv) > 2 else 40
f = sorted(glob.glob(d + '/trace/*/*kernel_trace.csv'), key=os.path.getmtime)[-1]
rows = [r for r in csv.DictReader(open(f)) if 'nsk::' in r['Kernel_Name'] and 'stream_copy' not in r['Kernel_Name']]
rows.sort(key=lambda r: int(r['Start_Timestamp']))
sel = rows[len(rows) // 2:len(rows) // 2 + n_show]
t0 = int(sel[0]['Start_Timestamp'])
for r in sel:
    nm = r['Kernel_Name'].split('(')[0].replace('void nsk::', '')[:44]
    s, e = int(r['Start_Timestamp']) - t0, int(r['End_Timestamp']) - t0
    print("%-46s start %8.1f end %8.1f dur %7.1f grid %s" % (nm, s / 1e3, e / 1e3, (e - s) / 1e3, r.get('Grid_Size_X', '')))
