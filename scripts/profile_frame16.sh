cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d gpurun_out/r01k_frame16 -o fr --output-format csv -- python3 scripts/frame_bench.py --streams 16 --frames 40 --only on > gpurun_out/r01k_frame16.txt 2>/dev/null
tail -1 gpurun_out/r01k_frame16.txt
python3 - <<'PY'
import csv
rows=list(csv.DictReader(open("gpurun_out/r01k_frame16/fr_kernel_stats.csv")))
tot=sum(float(r["TotalDurationNs"]) for r in rows)
print("sum of kernel durations %.1f ms" % (tot/1e6))
for r in rows[:34]:
    print(f'{r["Name"][:48]:48s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us total {float(r["TotalDurationNs"])/1e6:9.1f} ms')
PY
python3 - <<'PY'
# wall-clock span and GPU busy union from the kernel trace
import csv
rows=list(csv.DictReader(open("gpurun_out/r01k_frame16/fr_kernel_trace.csv")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"])) for r in rows)
# skip warm-up: last 70 % of the span
t0,t1=ev[0][0],max(e for _,e in ev)
lo=t0+(t1-t0)*0.4
busy=0; cur_s=None; cur_e=None
for s,e in ev:
    if e<lo: continue
    s=max(s,lo)
    if cur_e is None or s>cur_e:
        if cur_e is not None: busy+=cur_e-cur_s
        cur_s,cur_e=s,e
    else: cur_e=max(cur_e,e)
busy+=cur_e-cur_s
print("span %.1f ms, union busy %.1f ms (%.2f)" % ((t1-lo)/1e6,busy/1e6,busy/(t1-lo)))
PY
