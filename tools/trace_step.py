"""Summarise one training step from a rocprofv3 kernel trace CSV (per dispatch)."""
import csv, re, sys
path = sys.argv[1]
thr = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
rows = list(csv.DictReader(open(path)))
rows.sort(key=lambda r: int(r['Start_Timestamp']))
def short(name):
    n = name.replace('(anonymous namespace)::', '').replace('void ', '')
    m = re.search(r'([A-Za-z_0-9]+_kernel)(<[^>]*>)?', n)
    if m: return m.group(1) + (m.group(2) or '')
    m = re.search(r'\d+([a-z_0-9]+_kernel)(ILi(\d+)E)?', n)
    if m: return m.group(1) + (f"<{m.group(3)}>" if m.group(3) else '')
    return n[:40]
idx = [i for i, r in enumerate(rows) if 'adamw' in r['Kernel_Name']]
step = rows[idx[-3] + 1: idx[-1] + 1]
t0 = int(step[0]['Start_Timestamp'])
for r in step:
    d = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    if d >= thr:
        print(f"{(int(r['Start_Timestamp'])-t0)/1e6:8.2f} ms {d:8.1f} us  {short(r['Kernel_Name']):30s} grid {int(r['Grid_Size_X'])//int(r['Workgroup_Size_X'])}x{r['Grid_Size_Y']}x{r['Grid_Size_Z']} vgpr {r['VGPR_Count']} lds {r['LDS_Block_Size']}")
