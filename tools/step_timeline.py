"""Launch timeline of one generation step from a rocprofv3 --kernel-trace of tools/run_steps.py: start (us from the
first launch of the step), queue, kernel, grid and duration; consecutive launches of the same kernel and grid on the same
queue are shown once.    python tools/step_timeline.py <trace dir>"""
import csv, sys, glob
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = list(csv.DictReader(open(f)))
for r in rows:
    r['s'] = int(r['Start_Timestamp']); r['e'] = int(r['End_Timestamp'])
rows.sort(key=lambda r: r['s'])
# find the last mano_lbs kernel -> end of a step; take the step before it
manos = [i for i, r in enumerate(rows) if 'mano_lbs' in r['Kernel_Name']]
# steps end with the call-site mano (every 2nd mano launch)
end = manos[-1]; begin = manos[-3] + 1
t0 = rows[begin]['s']
print("step kernels:", end - begin + 1, "span ms", (rows[end]['e'] - t0) / 1e6)
def short(n): return n.replace('void dvq::', '').split('(')[0][:60]
last = None
for r in rows[begin:end + 1]:
    name = short(r['Kernel_Name'])
    gx = int(r['Grid_Size_X']) * int(r['Grid_Size_Y']) * int(r['Grid_Size_Z'])
    key = (name, r['Queue_Id'], gx)
    if key != last:
        print(f"{(r['s'] - t0) / 1e3:9.1f} us  q{r['Queue_Id']} {name} grid {gx} dur {(r['e'] - r['s']) / 1e3:.1f}")
        last = key
