cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
for S in ${STREAMS:-8}; do
rocprofv3 --kernel-trace --stats -d gpurun_out/r01k_ent$S -o e --output-format csv -- python3 scripts/entropy_only_bench.py --streams $S --reps 50 > gpurun_out/r01k_ent$S.txt 2>/dev/null
tail -1 gpurun_out/r01k_ent$S.txt
python3 - $S <<'PY'
import csv, sys, collections
S=sys.argv[1]
rows=list(csv.DictReader(open(f"gpurun_out/r01k_ent{S}/e_kernel_stats.csv")))
for r in rows[:20]:
    print(f'{r["Name"][:48]:48s} calls {int(r["Calls"]):6d} avg {float(r["AverageNs"])/1e3:9.1f} us')
tr=list(csv.DictReader(open(f"gpurun_out/r01k_ent{S}/e_kernel_trace.csv")))
ev=sorted((int(r["Start_Timestamp"]),int(r["End_Timestamp"]),r["Kernel_Name"][:30],r["Queue_Id"]) for r in tr)
t1=max(e[1] for e in ev); t0=ev[0][0]; lo=t0+(t1-t0)*0.5
busy=0; cs=ce=None; tot=0
perq=collections.defaultdict(int)
for s,e,_,q in ev:
    if e<lo: continue
    s=max(s,lo); tot+=e-s; perq[q]+=e-s
    if ce is None or s>ce:
        if ce is not None: busy+=ce-cs
        cs,ce=s,e
    else: ce=max(ce,e)
busy+=ce-cs
print("span %.1f ms, union busy %.1f ms, sum of durations %.1f ms (mean concurrency %.2f), queues %d" % ((t1-lo)/1e6,busy/1e6,tot/1e6,tot/(t1-lo),len(perq)))
print("busy fraction per queue:", " ".join("%.2f"%(v/(t1-lo)) for v in perq.values()))
PY
done
