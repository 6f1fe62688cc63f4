import csv, glob, collections, os, sys
d = sys.argv[1]
f = max(glob.glob(os.path.join(d, '**', '*_kernel_trace.csv'), recursive=True), key=os.path.getmtime)
rows = list(csv.DictReader(open(f)))
g = collections.defaultdict(list)
for r in rows:
    n = r['Kernel_Name'].split('(')[0][-40:]
    grid = (int(r['Grid_Size_X']) // int(r['Workgroup_Size_X']), int(r['Grid_Size_Y']), int(r['Grid_Size_Z']))
    g[(n, grid, r['VGPR_Count'], r['LDS_Block_Size'])].append(int(r['End_Timestamp']) - int(r['Start_Timestamp']))
tot = sum(sum(v) for v in g.values())
for k, v in sorted(g.items(), key=lambda kv: -sum(kv[1]))[:int(sys.argv[2]) if len(sys.argv) > 2 else 16]:
    v.sort()
    print('%-42s grid %-16s vgpr %-4s lds %-6s n %-6d avg %8.0f med %8d  %5.1f%%' % (k[0], k[1], k[2], k[3], len(v), sum(v) / len(v), v[len(v) // 2], 100 * sum(v) / tot))
