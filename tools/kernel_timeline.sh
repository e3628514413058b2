# kernel_timeline.sh - start / end of every kernel of one 8-channel decode (rocprofv3 --kernel-trace); run on the GPU box from the repo root
cd /tmp && export TMPDIR=/tmp
timeout -k 10 300 rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o out -- python3 $GRAFT_REPO_ROOT/tools/ab_bench.py --depth 24 --channels 8 --packets 16384 --rounds 1 --steps 2 $GRAFT_REPO_ROOT/saprobe-alac_amd/csrc/libalacgpu.so > /tmp/tr.log 2>&1
f=$(find /tmp/tr -name "*kernel_trace.csv" | head -1)
[ -n "$f" ] || { echo no trace; tail -5 /tmp/tr.log; exit 1; }
python3 - "$f" <<'PY'
import csv,sys
rows=[r for r in csv.DictReader(open(sys.argv[1])) if 'alack' in r['Kernel_Name']]
rows.sort(key=lambda r:int(r['Start_Timestamp']))
# last decode: take last 14 kernels
t0=None
for r in rows[-16:]:
    s,e=int(r['Start_Timestamp']),int(r['End_Timestamp'])
    if t0 is None: t0=s
    print("%-22s start %9.1f us  end %9.1f us  dur %8.1f  queue %s" % (r['Kernel_Name'].split('(')[0][7:], (s-t0)/1e3,(e-t0)/1e3,(e-s)/1e3, r.get('Queue_Id')))
PY
